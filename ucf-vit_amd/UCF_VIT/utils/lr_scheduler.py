"""Linear warm-up + cosine annealing, stepped once per iteration.

Same schedule as the reference's src/UCF_VIT/utils/lr_scheduler.py:12-94 (used at training_scripts/train_class_simple.py:357),
written in closed form: lr(t) depends only on t, not on the previous lr, so resuming and hipGraph-captured optimizer steps
cannot drift.  tests/test_oracle_golden.py compares it with a sequence produced by the reference class.
"""
import math

from torch.optim.lr_scheduler import LRScheduler


class LinearWarmupCosineAnnealingLR(LRScheduler):
    def __init__(self, optimizer, warmup_epochs, max_epochs, warmup_start_lr=0.0, eta_min=0.0, last_epoch=-1):
        self.warmup_epochs = warmup_epochs
        self.max_epochs = max_epochs
        self.warmup_start_lr = warmup_start_lr
        self.eta_min = eta_min
        super().__init__(optimizer, last_epoch)

    def _lr_at(self, t, base_lr):
        w, T = self.warmup_epochs, self.max_epochs
        if t < w:
            return self.warmup_start_lr + t * (base_lr - self.warmup_start_lr) / max(1, w - 1)
        return self.eta_min + 0.5 * (base_lr - self.eta_min) * (1.0 + math.cos(math.pi * (t - w) / (T - w)))

    def get_lr(self):
        return [self._lr_at(self.last_epoch, b) for b in self.base_lrs]

    _get_closed_form_lr = get_lr

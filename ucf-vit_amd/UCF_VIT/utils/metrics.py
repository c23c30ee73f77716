"""Losses on the hot path (reference: src/UCF_VIT/utils/metrics.py:11-17 masked_mse; nn.MSELoss / nn.CrossEntropyLoss
call sites in training_scripts/train_masked_simple.py:43-47 and train_class_simple.py:24-30), running on the HIP kernels."""
import torch

from .._hip import functional as HF


def masked_mse(pred, y, mask):
    """mean over the patch dim of (pred-y)^2, then sum(loss*mask)/sum(mask).  `y` is the patchified target
    [B, L, P]; prefer patch_mse_loss(), which reads the image directly and fuses forward+backward."""
    loss = ((pred.float() - y.float()) ** 2).mean(dim=-1)
    return (loss * mask).sum() / mask.sum()


def patch_mse_loss(pred, data, patch_size, mask=None):
    """MSE(pred, patchify(data)) without materialising the target; mask=None -> plain MSE over all patches
    (config loss_fn "MSE"), mask given -> masked_mse semantics ("maskMSE")."""
    return HF.patch_mse(pred, data, patch_size, mask)


def seq_mse_loss(pred, seq, mask=None):
    """MSE(pred, rearrange(seq, 'b c s p -> b s (p c)')) for adaptively patched input (reference train_masked_simple.py:24-32),
    target never materialised; mask given -> masked_mse semantics."""
    return HF.patch_mse(pred, seq, None, mask)


def cross_entropy_loss(logits, labels):
    return HF.cross_entropy(logits, labels)

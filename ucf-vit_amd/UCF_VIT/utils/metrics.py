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


class DiceBLoss(torch.nn.Module):
    """Dice + binary cross-entropy over the non-background channels of a segmentation map (reference utils/metrics.py:95-121, the
    loss of train_sap_simple.py:44-46): sigmoid, channels 1.. flattened, loss = w * BCE + (1 - w) * (1 - dice).  Loss arithmetic on
    the [B, classes, ...] output map runs on torch ops in fp32 (outside the token path)."""

    def __init__(self, weight=0.5, num_class=2, size_average=True):
        super().__init__()
        self.weight, self.num_class = weight, num_class

    def forward(self, inputs, targets, smooth=1, act=True):
        inputs = inputs.float()
        if act:
            inputs = torch.sigmoid(inputs)
        pred = torch.flatten(inputs[:, 1:])
        true = torch.flatten(targets[:, 1:].float())
        inter = (pred * true).sum()
        dice_loss = 1 - (2. * inter + smooth) / (pred.sum() + true.sum() + smooth)
        bce = torch.nn.functional.binary_cross_entropy(pred, true, reduction='mean')
        return self.weight * bce + (1 - self.weight) * dice_loss

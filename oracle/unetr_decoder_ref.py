"""TEST INFRASTRUCTURE — CPU / plain-torch restatement of the UNETR decoder's normalisation chain and of its loss, the checker for
csrc/unetr_decoder.hip (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import oracle/).

PARITY UNPINNED: the arithmetic lives in the un-vendored dependency monai >= 1.4.0 (pyproject.toml:20; `==1.4.0` in Docker/Dockerfile:6),
which is not installed here and for which the reference ships no fixtures.  The functions restate monai's published formulas at the
reference's call sites:
  * UnetResBlock (monai.networks.blocks.dynunet_block, used by UnetrBasicBlock / UnetrPrUpBlock / UnetrUpBlock at
    src/UCF_VIT/simple/arch.py:808-940): conv -> InstanceNorm -> LeakyReLU(0.01) -> conv -> InstanceNorm; residual = 1x1 conv -> InstanceNorm
    when the channel count or stride changes, else the input; out = LeakyReLU(out + residual).  InstanceNorm: affine-free, biased variance,
    eps 1e-5 (torch.nn.InstanceNorm3d defaults).
  * DiceCELoss(to_onehot_y=True, softmax=True, squared_pred=True) (training_scripts/train_unetr_simple.py:38): DiceLoss with
    smooth_nr = smooth_dr = 1e-5, include_background, mean over batch and classes, + nn.CrossEntropyLoss (mean), both with weight 1.
"""
import torch
import torch.nn.functional as F


def inst_norm_act(x, res=None, eps=1e-5, slope=0.01):
    """leaky_relu(instance_norm(x) [+ res], slope); slope 1.0 = no activation"""
    y = F.instance_norm(x, eps=eps)
    if res is not None:
        y = y + res
    return y if slope == 1.0 else F.leaky_relu(y, slope)


def res_block(inp, w1, w2, w3=None, slope=0.01, eps=1e-5):
    """UnetResBlock with kernel-3 / kernel-1 bias-free convolutions given as weights (2-D or 3-D by the weights' rank)"""
    conv = F.conv3d if w1.dim() == 5 else F.conv2d
    out = inst_norm_act(conv(inp, w1, padding=1), None, eps, slope)
    out = conv(out, w2, padding=1)
    residual = inst_norm_act(conv(inp, w3), None, eps, 1.0) if w3 is not None else inp
    return inst_norm_act(out, residual, eps, slope)


def dice_ce_loss(logits, label, smooth_nr=1e-5, smooth_dr=1e-5):
    """logits [B, n, *spatial], label int64 [B, *spatial]"""
    n = logits.shape[1]
    prob = logits.float().softmax(dim=1)
    onehot = F.one_hot(label, n).movedim(-1, 1).float()
    dims = tuple(range(2, logits.dim()))
    inter = (prob * onehot).sum(dims)
    denom = (prob ** 2).sum(dims) + (onehot ** 2).sum(dims)
    dice = 1.0 - (2.0 * inter + smooth_nr) / (denom + smooth_dr)
    return dice.mean() + F.cross_entropy(logits.float(), label)

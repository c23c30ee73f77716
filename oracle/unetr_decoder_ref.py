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


# ---- the whole skip-connection decoder (reference wiring: src/UCF_VIT/simple/arch.py:951-993 unetr_head / proj_feat, blocks :808-940) ----
def proj_feat(tokens, feat_size, hidden):
    """arch.py:951-958: [B, N, D] -> [B, D, *feat_size]"""
    x = tokens.float().view(tokens.size(0), *feat_size, hidden)
    return x.permute(0, len(feat_size) + 1, *range(1, len(feat_size) + 1)).contiguous()


def _res(p, prefix, inp):
    w3 = p.get(prefix + ".conv3.conv.weight")
    return res_block(inp, p[prefix + ".conv1.conv.weight"], p[prefix + ".conv2.conv.weight"], w3)


def _pr_up(p, prefix, x, num_layer):
    """monai UnetrPrUpBlock: transposed conv (k 2, s 2), then num_layer x [transposed conv, UnetResBlock]"""
    x = F.conv_transpose3d(x, p[prefix + ".transp_conv_init.conv.weight"], stride=2)
    for i in range(num_layer):
        x = F.conv_transpose3d(x, p[f"{prefix}.blocks.{i}.0.conv.weight"], stride=2)
        x = _res(p, f"{prefix}.blocks.{i}.1", x)
    return x


def _up(p, prefix, inp, skip):
    """monai UnetrUpBlock: transposed conv, concatenate the skip, UnetResBlock"""
    out = F.conv_transpose3d(inp, p[prefix + ".transp_conv.conv.weight"], stride=2)
    return _res(p, prefix + ".conv_block", torch.cat((out, skip), dim=1))


def unetr_head(p, img, feats, taps, feat_size, hidden):
    """arch.py:981-993 (skip_connection=True, full-resolution grid): p = the model's state_dict (fp32), img [B, C, X, Y, Z], feats = the
    final-normed tokens, taps = the three intermediate token maps (blocks depth/4, 2 depth/4, 3 depth/4) -> logits [B, classes, X, Y, Z]"""
    enc1 = _res(p, "encoder1.layer", img.float())
    dec4 = proj_feat(feats, feat_size, hidden)
    dec3 = _up(p, "decoder5", dec4, _pr_up(p, "encoder4", proj_feat(taps[2], feat_size, hidden), 0))
    dec2 = _up(p, "decoder4", dec3, _pr_up(p, "encoder3", proj_feat(taps[1], feat_size, hidden), 1))
    dec1 = _up(p, "decoder3", dec2, _pr_up(p, "encoder2", proj_feat(taps[0], feat_size, hidden), 2))
    out = _up(p, "decoder2", dec1, enc1)
    return F.conv3d(out, p["out.conv.conv.weight"], p["out.conv.conv.bias"])

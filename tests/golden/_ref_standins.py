"""Test-only stand-ins for the third-party helper symbols the reference imports
(timm / monai are not installed in the build container).

Only glue is provided (tuple helpers, identity-at-config DropPath / LayerScale,
torch's own trunc_normal_, a module walker).  Every arithmetic op on the hot path
(conv, linear, LayerNorm, GELU, softmax, gather, argsort) stays the reference's own
code calling torch.  Used ONLY by make_golden.py inside the build container; it is
never imported by the product, the tests on the GPU box, or bench.py.

Symbols mirrored: see /root/reference/src/UCF_VIT/simple/building_blocks.py:14-26 and
simple/arch.py:20,33-34.
"""
import sys
import types
from itertools import repeat
import collections.abc

import torch
import torch.nn as nn


def _ntuple(n):
    def parse(x):
        if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
            return tuple(x)
        return tuple(repeat(x, n))
    return parse


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        assert self.drop_prob == 0.0 or not self.training, "stand-in supports drop_path=0 only"
        return x


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5, inplace=False):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _Unused(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("stand-in: not on the hot path")


def named_apply(fn, module, name='', depth_first=True, include_root=False):
    if not depth_first and include_root:
        fn(module=module, name=name)
    for child_name, child in module.named_children():
        child_name = '.'.join((name, child_name)) if name else child_name
        named_apply(fn=fn, module=child, name=child_name, depth_first=depth_first, include_root=True)
    if depth_first and include_root:
        fn(module=module, name=name)
    return module


def install():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    timm = mod('timm')
    layers = mod('timm.layers')
    helpers = mod('timm.layers.helpers')
    trace = mod('timm.layers.trace_utils')
    models = mod('timm.models')
    vt = mod('timm.models.vision_transformer')
    manip = mod('timm.models._manipulate')
    timm.layers, timm.models = layers, models
    layers.helpers, layers.trace_utils = helpers, trace
    models.vision_transformer, models._manipulate = vt, manip

    helpers.to_2tuple = _ntuple(2)
    helpers.to_3tuple = _ntuple(3)
    trace._assert = lambda cond, msg: (_ for _ in ()).throw(AssertionError(msg)) if not cond else None
    layers.DropPath = DropPath
    layers.AttentionPoolLatent = _Unused
    layers.PatchDropout = _Unused
    layers.trunc_normal_ = nn.init.trunc_normal_
    layers.resample_patch_embed = None
    layers.resample_abs_pos_embed = None
    layers.get_act_layer = lambda x: x
    layers.get_norm_layer = lambda x: x
    layers.LayerType = object
    layers.use_fused_attn = lambda: True
    vt.LayerScale = LayerScale
    manip.named_apply = named_apply
    manip.checkpoint_seq = None

    monai = mod('monai')
    nets = mod('monai.networks')
    blocks = mod('monai.networks.blocks')
    dyn = mod('monai.networks.blocks.dynunet_block')
    monai.networks, nets.blocks, blocks.dynunet_block = nets, blocks, dyn
    blocks.UnetrBasicBlock = blocks.UnetrPrUpBlock = blocks.UnetrUpBlock = _Unused
    dyn.UnetOutBlock = _Unused
    dyn.get_conv_layer = None

    sys.path.insert(0, '/root/reference/src')

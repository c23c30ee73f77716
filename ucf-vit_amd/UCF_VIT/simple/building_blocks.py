"""Operator layer of the MI355X build — drop-in for the reference's src/UCF_VIT/simple/building_blocks.py
(PatchEmbed:30, Mlp:94, Attention:131, Block:194, MyUnetBlock:241, EmbeddingDenseLayer:286,
VariableMapping_Attention:301): same class names, constructor arguments, forward signatures and state_dict keys.

The modules only HOLD parameters (ordinary nn.Linear / nn.Conv / nn.LayerNorm children, so checkpoints interchange);
their forward/backward run exclusively on the hand-written gfx950 kernels of libucfvit_hip.so through
UCF_VIT._hip.functional.  There is no PyTorch fallback: CPU tensors or a missing library raise.

compute_dtype (torch.float32 | torch.bfloat16) selects the arithmetic: float32 = the reference's `simple` mode
(exact-fp32 MFMA), bfloat16 = the reference's fsdp MixedPrecision(bf16) semantics with fp32 master weights.
Set it with set_compute_dtype(model, dtype).
"""
from functools import partial
from itertools import repeat
from typing import Callable, Optional
import collections.abc

import torch
import torch.nn as nn

from UCF_VIT.utils.fused_attn import FusedAttn
from UCF_VIT._hip import functional as HF


def _ntuple(n):
    def parse(x):
        if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
            return tuple(x)
        return tuple(repeat(x, n))
    return parse


to_2tuple, to_3tuple = _ntuple(2), _ntuple(3)
LayerType = object


def _assert(cond, msg):
    if not cond:
        raise AssertionError(msg)


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def get_act_layer(x):
    return x


def get_norm_layer(x):
    return x


def set_compute_dtype(module: nn.Module, dtype: torch.dtype):
    """Select fp32 (reference `simple` mode) or bf16 (reference fsdp MixedPrecision semantics) arithmetic for a model."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    for m in module.modules():
        m.compute_dtype = dtype
    return module


def _cd(m):
    return getattr(m, "compute_dtype", torch.float32)


def _no_dropout(p, training, what):
    if p > 0.0 and training:
        raise NotImplementedError(f"{what}={p}: the HIP hot path implements the reference configs (all drop rates 0.0)")


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm parameters, HIP kernel arithmetic (ucfvit_layernorm_fwd/bwd)."""

    def forward(self, x):
        return HF.LayerNormFn.apply(x, self.weight, self.bias, self.eps, _cd(self))


class Linear(nn.Linear):
    """nn.Linear parameters, HIP MFMA GEMM arithmetic (ucfvit_gemm)."""

    def forward(self, x):
        return HF.LinearFn.apply(x, self.weight, self.bias, _cd(self))


class DropPath(nn.Module):
    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        _no_dropout(self.drop_prob, self.training, "drop_path")
        return x


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5, inplace=False):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        raise NotImplementedError("LayerScale (init_values) is not on the HIP hot path; reference configs use init_values=None")


class PatchDropout(nn.Module):
    def __init__(self, prob=0.0, num_prefix_tokens=1):
        super().__init__()
        self.prob = prob

    def forward(self, x):
        _no_dropout(self.prob, self.training, "patch_drop_rate")
        return x


AttentionPoolLatent = None
resample_patch_embed = None
resample_abs_pos_embed = None


class PatchEmbed(nn.Module):
    """2-D / 3-D image -> patch tokens [B, L, D]; `proj` is a Conv2d/Conv3d parameter holder (weight [D,C,p,p(,p)])."""

    def __init__(self, img_size: Optional[int] = 224, patch_size: int = 16, in_chans: int = 3, embed_dim: int = 768,
                 twoD: Optional[bool] = True, norm_layer: Optional[Callable] = None, bias: bool = True,
                 sqrt_len_method: bool = False):
        super().__init__()
        self.twoD = twoD
        self.sqrt_len_method = sqrt_len_method
        nd = 2 if twoD else 3
        self.patch_size = _ntuple(nd)(patch_size)
        if img_size is None:
            self.img_size = self.grid_size = self.num_patches = None
        else:
            self.img_size = _ntuple(nd)(img_size)
            self.grid_size = tuple(s // p for s, p in zip(self.img_size, self.patch_size))
            n = 1
            for g in self.grid_size:
                n *= g
            self.num_patches = n
        conv = nn.Conv2d if twoD else nn.Conv3d
        self.proj = conv(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        if self.img_size is not None and not self.sqrt_len_method:
            for i, name in enumerate(("height", "width", "depth")[: len(self.img_size)]):
                _assert(x.shape[2 + i] == self.img_size[i], f"Input {name} ({x.shape[2 + i]}) doesn't match model ({self.img_size[i]}).")
        p = self.patch_size[0]
        _assert(all(q == p for q in self.patch_size), "HIP patch embedding needs a cubic/square patch")
        x = HF.PatchEmbedFn.apply(x, self.proj.weight, self.proj.bias, p, _cd(self))
        return self.norm(x)


class Mlp(nn.Module):
    """fc1 -> erf-GELU -> fc2 as two MFMA GEMMs with fused bias/activation epilogues."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, norm_layer=None, bias=True,
                 drop=0.0, use_conv=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        bias = to_2tuple(bias)
        drop_probs = to_2tuple(drop)
        if use_conv:
            raise NotImplementedError("Mlp(use_conv=True) is not used by the reference models")
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias[0])
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop_probs[0])
        self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias[1])
        self.drop2 = nn.Dropout(drop_probs[1])

    def _fusable(self):
        return (isinstance(self.act, nn.GELU) and getattr(self.act, "approximate", "none") == "none"
                and isinstance(self.norm, nn.Identity))

    def forward(self, x):
        _no_dropout(self.drop1.p, self.training, "Mlp.drop")
        _no_dropout(self.drop2.p, self.training, "Mlp.drop")
        if not self._fusable():
            raise NotImplementedError("HIP Mlp implements fc1 -> exact GELU -> fc2 (the reference configuration)")
        return HF.MlpFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, _cd(self))


class Attention(nn.Module):
    """Multi-head self-attention; every FusedAttn member runs the fused gfx950 kernel (see utils/fused_attn.py)."""

    def __init__(self, dim: int, fused_attn: FusedAttn = FusedAttn.NONE, num_heads: int = 8, qkv_bias: bool = False,
                 qk_norm: bool = False, attn_drop: float = 0.0, proj_drop: float = 0.0, norm_layer: nn.Module = nn.LayerNorm) -> None:
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.fused_attn = fused_attn
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def _fusable(self):
        return isinstance(self.q_norm, nn.Identity) and isinstance(self.k_norm, nn.Identity)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _no_dropout(self.attn_drop.p, self.training, "attn_drop")
        _no_dropout(self.proj_drop.p, self.training, "proj_drop")
        if not self._fusable():
            raise NotImplementedError("qk_norm=True is not on the HIP hot path; reference configs use qk_norm=False")
        return HF.AttentionFn.apply(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, self.num_heads, _cd(self))


class Block(nn.Module):
    """Pre-LN transformer block.  With the reference configuration (no LayerScale, drop_path 0, standard Attention/Mlp)
    the whole block is ONE autograd node running a fixed sequence of fused HIP launches (HF.BlockFn)."""

    def __init__(self, dim: int, num_heads: int, fused_attn: FusedAttn = FusedAttn.NONE, mlp_ratio: float = 4.0,
                 qkv_bias: bool = False, qk_norm: bool = False, proj_drop: float = 0.0, attn_drop: float = 0.0,
                 init_values: Optional[float] = None, drop_path: float = 0.0, act_layer: nn.Module = nn.GELU,
                 norm_layer: nn.Module = LayerNorm, mlp_layer: nn.Module = Mlp) -> None:
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, fused_attn=fused_attn, num_heads=num_heads, qkv_bias=qkv_bias, qk_norm=qk_norm,
                              attn_drop=attn_drop, proj_drop=proj_drop, norm_layer=norm_layer)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = mlp_layer(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=proj_drop)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path2 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.activation_checkpointing = False      # set by apply_activation_checkpointing(): keep only the Block's input for backward

    def _fusable(self):
        return (isinstance(self.norm1, nn.LayerNorm) and isinstance(self.norm2, nn.LayerNorm)
                and self.norm1.elementwise_affine and self.norm2.elementwise_affine and self.norm1.eps == self.norm2.eps
                and isinstance(self.ls1, nn.Identity) and isinstance(self.ls2, nn.Identity)
                and type(self.attn) is Attention and self.attn._fusable()
                and type(self.mlp) is Mlp and self.mlp._fusable()
                and self.mlp.fc1.out_features % 8 == 0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._fusable() and x.dim() == 3:
            for d in (self.drop_path1, self.drop_path2):
                if isinstance(d, DropPath):
                    _no_dropout(d.drop_prob, self.training, "drop_path")
            a, m = self.attn, self.mlp
            _no_dropout(a.attn_drop.p, self.training, "attn_drop")
            _no_dropout(a.proj_drop.p, self.training, "proj_drop")
            _no_dropout(m.drop1.p, self.training, "Mlp.drop")
            return HF.BlockFn.apply(x, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                                    self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                    a.num_heads, self.norm1.eps, _cd(self), None, self.activation_checkpointing and torch.is_grad_enabled())
        x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
        x = x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))
        return x


class MyUnetBlock(nn.Module):
    """Transposed-conv upsampling stage of the skip-less UNETR decoder (reference :241-284; monai get_conv_layer(conv_only,
    is_transposed) == a bias-free ConvTranspose).  Not covered by the HIP convolution kernels: a model that uses it must be built with
    UNETR(allow_torch_decoder=True) (torch's own convolutions; UNETR.forward raises otherwise)."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, upsample_kernel_size: int, res_block: bool = False) -> None:
        super().__init__()
        conv = nn.ConvTranspose3d if spatial_dims == 3 else nn.ConvTranspose2d
        self.transp_conv = nn.Sequential()
        self.transp_conv.add_module("conv", conv(in_channels, out_channels, kernel_size=upsample_kernel_size,
                                                 stride=upsample_kernel_size, bias=False))

    def forward(self, inp):
        return self.transp_conv(inp.float())


class EmbeddingDenseLayer(nn.Module):
    """time-embedding MLP of DiffusionVIT (reference :286-299): linear -> ReLU -> dropout -> linear"""

    def __init__(self, c_in: int, c_out: int, dropout_prob: float):
        super().__init__()
        self.linear1 = nn.Linear(c_in, c_out)
        self.linear2 = nn.Linear(c_out, c_out)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(p=dropout_prob)

    def forward(self, x):
        return self.linear2(self.dropout(self.relu(self.linear1(x))))


class VariableMapping_Attention(nn.Module):
    """Cross-attention that aggregates the per-variable tokens of every position with a learnt query (reference :301-373).
    forward(var_query [1, N_a = 1, D], x [V, R, D]) -> [R, D]: the q / kv / proj Linears are MFMA GEMMs, the softmax over the V
    variables and the weighted sum are one HBM-bound kernel (csrc/varagg.hip).  One aggregated variable (the reference's
    `aggregated_variables = 1`); x is laid out variable-major instead of [R, V, D] so that no permuted copy is needed."""

    def __init__(self, dim: int, fused_attn: FusedAttn = FusedAttn.NONE, num_heads: int = 8, qkv_bias: bool = False,
                 qk_norm: bool = False, proj_bias: bool = True, attn_drop: float = 0.0, proj_drop: float = 0.0,
                 norm_layer: nn.Module = nn.LayerNorm) -> None:
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.fused_attn = fused_attn
        self.q = Linear(dim, dim, bias=qkv_bias)
        self.kv = Linear(dim, dim * 2, bias=qkv_bias)
        self.q_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = Linear(dim, dim, bias=proj_bias)
        self.proj_drop = nn.Dropout(proj_drop)
        self.qk_norm = qk_norm

    def forward(self, var_query: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        if self.qk_norm or (self.training and (self.attn_drop.p > 0.0 or self.proj_drop.p > 0.0)):
            raise NotImplementedError("qk_norm / dropout inside the variable aggregation are not on the HIP hot path")
        if var_query.shape[1] != 1:
            raise NotImplementedError("aggregated_variables > 1 is not on the HIP hot path")
        V, R, D = x.shape
        q = self.q(var_query.reshape(1, D))                       # [1, D], the same query for every token
        kv = self.kv(x.reshape(V * R, D))                         # [V * R, 2 D]
        out = HF.VarAggFn.apply(kv, q, V, self.head_dim, self.scale)
        return self.proj(out)


def apply_activation_checkpointing(model, check_fn=None):
    """the reference's `apply_activation_checkpointing(model, checkpoint_wrapper_fn=checkpoint_wrapper, check_fn=lambda m: isinstance(m, Block))`
    (training_scripts/train_masked_fsdp.py:393-396) for the HIP Blocks: every Block (or every module check_fn accepts) keeps only its input
    and re-runs its forward launches inside backward (HF.BlockFn recompute).  Returns the number of Blocks switched."""
    n = 0
    for mod in model.modules():
        if isinstance(mod, Block) and (check_fn is None or check_fn(mod)):
            mod.activation_checkpointing = True
            n += 1
    return n

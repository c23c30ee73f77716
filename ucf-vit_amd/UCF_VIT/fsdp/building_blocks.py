"""Hybrid-OP (tensor-parallel) operator layer — drop-in for the reference's src/UCF_VIT/fsdp/building_blocks.py
(Mlp:98, Attention:146, Block:221): same constructor arguments (tensor_par_size, tensor_par_group) and parameter shapes —
column-parallel qkv [3D/tp, D] (= H/tp heads) and fc1 [4D/tp, D], row-parallel proj [D, D/tp] and fc2 [D, 4D/tp] — with the
local arithmetic on the gfx950 kernels and the entry/exit collectives on RCCL (torch.distributed "nccl").

Collectives per Block: forward 2 SUM all-reduces of [B,N,D] (after proj and after fc2), backward 2 (gradients entering the
qkv and fc1 GEMMs); see UCF_VIT._hip.functional (_attn_fwd/_attn_bwd/_mlp_fwd/_mlp_bwd).  Like the reference, every rank adds
the full proj/fc2 bias before the SUM (bias counted tensor_par_size times — SURVEY.md §0 known defect, kept for parity).
"""
from typing import Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from UCF_VIT.simple.building_blocks import (  # noqa: F401  (re-exported operator set)
    PatchEmbed, LayerNorm, Linear, DropPath, LayerScale, PatchDropout, MyUnetBlock, EmbeddingDenseLayer,
    VariableMapping_Attention, to_2tuple, to_3tuple, trunc_normal_, get_act_layer, get_norm_layer, LayerType,
    set_compute_dtype, _cd, _no_dropout, _assert)
from UCF_VIT.simple import building_blocks as _S
from UCF_VIT.utils.fused_attn import FusedAttn
from UCF_VIT._hip import functional as HF


def _tp(size, group):
    return HF.TP(group, size) if size > 1 else None


class Mlp(_S.Mlp):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, norm_layer=None, bias=True,
                 drop=0.0, use_conv=False, tensor_par_size: int = 1, tensor_par_group: Optional[dist.ProcessGroup] = None):
        hidden_features = hidden_features or in_features
        assert hidden_features % tensor_par_size == 0
        super().__init__(in_features, hidden_features // tensor_par_size, out_features or in_features, act_layer, None, bias, drop, use_conv)
        if norm_layer is not None:
            self.norm = norm_layer(hidden_features)
        self.tensor_par_size, self.tensor_par_group = tensor_par_size, tensor_par_group

    def forward(self, x):
        _no_dropout(self.drop1.p, self.training, "Mlp.drop")
        if not self._fusable():
            raise NotImplementedError("HIP Mlp implements fc1 -> exact GELU -> fc2 (the reference configuration)")
        return HF.MlpFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, _cd(self),
                              _tp(self.tensor_par_size, self.tensor_par_group))


class Attention(nn.Module):
    def __init__(self, dim: int, fused_attn: FusedAttn = FusedAttn.NONE, num_heads: int = 8, qkv_bias: bool = False,
                 qk_norm: bool = False, attn_drop: float = 0.0, proj_drop: float = 0.0, norm_layer: nn.Module = nn.LayerNorm,
                 tensor_par_size: int = 1, tensor_par_group: Optional[dist.ProcessGroup] = None) -> None:
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'
        assert num_heads % tensor_par_size == 0, 'num_heads should be divisible by tensor_par_size'
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.fused_attn = fused_attn
        self.tensor_par_size, self.tensor_par_group = tensor_par_size, tensor_par_group
        self.qkv = nn.Linear(dim, dim * 3 // tensor_par_size, bias=qkv_bias)
        self.q_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim // tensor_par_size, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def _fusable(self):
        return isinstance(self.q_norm, nn.Identity) and isinstance(self.k_norm, nn.Identity)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _no_dropout(self.attn_drop.p, self.training, "attn_drop")
        _no_dropout(self.proj_drop.p, self.training, "proj_drop")
        if not self._fusable():
            raise NotImplementedError("qk_norm=True is not on the HIP hot path; reference configs use qk_norm=False")
        return HF.AttentionFn.apply(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, self.num_heads, _cd(self),
                                    _tp(self.tensor_par_size, self.tensor_par_group))


class Block(nn.Module):
    def __init__(self, dim: int, num_heads: int, fused_attn: FusedAttn = FusedAttn.NONE, mlp_ratio: float = 4.0,
                 qkv_bias: bool = False, qk_norm: bool = False, proj_drop: float = 0.0, attn_drop: float = 0.0,
                 init_values: Optional[float] = None, drop_path: float = 0.0, act_layer: nn.Module = nn.GELU,
                 norm_layer: nn.Module = LayerNorm, mlp_layer: nn.Module = Mlp, tensor_par_size: int = 1,
                 tensor_par_group: Optional[dist.ProcessGroup] = None) -> None:
        super().__init__()
        self.tensor_par_size, self.tensor_par_group = tensor_par_size, tensor_par_group
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, fused_attn=fused_attn, num_heads=num_heads, qkv_bias=qkv_bias, qk_norm=qk_norm, attn_drop=attn_drop,
                              proj_drop=proj_drop, norm_layer=norm_layer, tensor_par_size=tensor_par_size, tensor_par_group=tensor_par_group)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = mlp_layer(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=proj_drop,
                             tensor_par_size=tensor_par_size, tensor_par_group=tensor_par_group)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path2 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def _fusable(self):
        return (isinstance(self.norm1, nn.LayerNorm) and isinstance(self.norm2, nn.LayerNorm) and self.norm1.eps == self.norm2.eps
                and isinstance(self.ls1, nn.Identity) and isinstance(self.ls2, nn.Identity)
                and type(self.attn) is Attention and self.attn._fusable() and type(self.mlp) is Mlp and self.mlp._fusable()
                and self.mlp.fc1.out_features % 8 == 0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._fusable() and x.dim() == 3:
            a, m = self.attn, self.mlp
            for d in (self.drop_path1, self.drop_path2):
                if isinstance(d, DropPath):
                    _no_dropout(d.drop_prob, self.training, "drop_path")
            _no_dropout(a.attn_drop.p, self.training, "attn_drop")
            _no_dropout(a.proj_drop.p, self.training, "proj_drop")
            _no_dropout(m.drop1.p, self.training, "Mlp.drop")
            return HF.BlockFn.apply(x, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                                    self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                    a.num_heads, self.norm1.eps, _cd(self), _tp(self.tensor_par_size, self.tensor_par_group),
                                    getattr(self, "activation_checkpointing", False) and torch.is_grad_enabled())
        x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
        return x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))

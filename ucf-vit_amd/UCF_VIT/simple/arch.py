"""Model layer of the MI355X build — drop-in for the reference's src/UCF_VIT/simple/arch.py
(VIT:101, SAP:491, MAE:538, UNETR:757, DiffusionVIT:1115): same class names, constructor keywords, forward signatures
and state_dict layout (incl. the `token_embeds` alias of `patch_embed`), so the reference's training scripts and
checkpoints interchange.  All token-path arithmetic (patch embedding, cls/pos assembly, transformer blocks, final norm,
head, MAE masking/gather/unshuffle) runs on libucfvit_hip.so; nothing here falls back to PyTorch math.

Round-1 scope: image input with adaptive_patching=False and use_varemb=False — what every BASELINE
config uses.  The adaptive-patching and variable-aggregation front ends keep their parameters (checkpoint compatible)
but raise NotImplementedError in forward (§8f "next" rows 1 and 4).
"""
from functools import partial
from typing import Callable, List, Optional, Tuple, Type, Union

try:
    from typing import Literal
except ImportError:  # pragma: no cover
    from typing_extensions import Literal

import numpy as np
import os

import torch
import torch.nn as nn

from .building_blocks import (Block, PatchEmbed, Mlp, LayerNorm, Linear, DropPath, PatchDropout, trunc_normal_, get_act_layer,
                              get_norm_layer, LayerType, MyUnetBlock, EmbeddingDenseLayer, VariableMapping_Attention,
                              set_compute_dtype, _cd)
from UCF_VIT.utils.pos_embed import (get_1d_sincos_pos_embed_from_grid, get_2d_sincos_pos_embed, get_3d_sincos_pos_embed,
                                     SinusoidalEmbeddings)
from UCF_VIT.utils.fused_attn import FusedAttn
from UCF_VIT._hip import functional as HF
from UCF_VIT._hip.params import ensure_store


def named_apply(fn: Callable, module: nn.Module, name: str = '', depth_first: bool = True, include_root: bool = False) -> nn.Module:
    if not depth_first and include_root:
        fn(module=module, name=name)
    for child_name, child in module.named_children():
        full = f"{name}.{child_name}" if name else child_name
        named_apply(fn=fn, module=child, name=full, depth_first=depth_first, include_root=True)
    if depth_first and include_root:
        fn(module=module, name=name)
    return module


def feature_take_indices(num_features: int, indices: Optional[Union[int, List[int]]] = None, as_set: bool = False):
    """absolute block indices to tap (None: all, int n: last n, list: as given, negatives from the end) and their max"""
    if indices is None:
        indices = num_features
    if isinstance(indices, int):
        assert 0 < indices <= num_features, f'last-n ({indices}) is out of range (1 to {num_features})'
        take = [num_features - indices + i for i in range(indices)]
    else:
        take = []
        for i in indices:
            idx = num_features + i if i < 0 else i
            assert 0 <= idx < num_features, f'feature index {idx} is out of range (0 to {num_features - 1})'
            take.append(idx)
    return (set(take) if as_set else take), max(take)


def init_weights_vit_timm(module: nn.Module, name: str = '') -> None:
    """timm ViT init: Linear weights trunc_normal(std .02), biases zero"""
    if isinstance(module, nn.Linear):
        trunc_normal_(module.weight, std=.02)
        if module.bias is not None:
            nn.init.zeros_(module.bias)
    elif hasattr(module, 'init_weights'):
        module.init_weights()


def get_init_weights_vit(head_bias: float = 0.0) -> Callable:
    return init_weights_vit_timm


def global_pool_nlc(x: torch.Tensor, num_prefix_tokens: int = 1):
    return x[:, 0] if num_prefix_tokens == 1 else x[:, num_prefix_tokens:]


class VIT(nn.Module):
    def __init__(
            self,
            img_size: Union[int, Tuple[int, int], Tuple[int, int, int]] = 224,
            patch_size: Union[int, Tuple[int, int], Tuple[int, int, int]] = 16,
            in_chans: int = 3,
            num_classes: Optional[int] = None,
            embed_dim: int = 768,
            depth: int = 12,
            num_heads: int = 12,
            mlp_ratio: float = 4.,
            qkv_bias: bool = True,
            qk_norm: bool = False,
            init_values: Optional[float] = None,
            class_token: bool = True,
            pos_embed: str = 'learn',
            drop_rate: float = 0.,
            pos_drop_rate: float = 0.,
            patch_drop_rate: float = 0.,
            proj_drop_rate: float = 0.,
            attn_drop_rate: float = 0.,
            drop_path_rate: float = 0.,
            weight_init: Literal['skip', ''] = '',
            embed_layer: Callable = PatchEmbed,
            norm_layer: Optional[LayerType] = None,
            act_layer: Optional[LayerType] = None,
            block_fn: Type[nn.Module] = Block,
            mlp_layer: Type[nn.Module] = Mlp,
            twoD: Optional[bool] = True,
            adaptive_patching: Optional[bool] = False,
            fixed_length: Optional[int] = 4096,
            default_vars: List = None,
            single_channel: bool = False,
            use_varemb: bool = False,
            FusedAttn_option=FusedAttn.NONE,
            use_adaptive_pos_emb: bool = False,
            sqrt_len_method: bool = False,
    ) -> None:
        super().__init__()
        assert pos_embed in ('', 'none', 'learn')
        norm_layer = get_norm_layer(norm_layer) or partial(LayerNorm, eps=1e-6)
        act_layer = get_act_layer(act_layer) or nn.GELU
        self.norm_layer, self.act_layer, self.mlp_layer, self.block_fn = norm_layer, act_layer, mlp_layer, block_fn
        self.num_classes, self.embed_dim = num_classes, embed_dim
        self.num_prefix_tokens = 1 if class_token else 0
        self.in_chans, self.patch_size, self.twoD = in_chans, patch_size, twoD
        self.qkv_bias, self.qk_norm = qkv_bias, qk_norm
        self.drop_path_rate, self.proj_drop_rate, self.attn_drop_rate = drop_path_rate, proj_drop_rate, attn_drop_rate
        self.init_values, self.img_size, self.num_heads, self.depth = init_values, img_size, num_heads, depth
        self.adaptive_patching, self.fixed_length = adaptive_patching, fixed_length
        self.default_vars, self.single_channel, self.use_varemb = default_vars, single_channel, use_varemb
        self.aggregated_variables = 1
        self.class_token, self.FusedAttn_option = class_token, FusedAttn_option
        self.use_adaptive_pos_emb, self.sqrt_len_method = use_adaptive_pos_emb, sqrt_len_method
        self.compute_dtype = torch.float32

        tokens_from_patches = not (adaptive_patching and not sqrt_len_method)
        if tokens_from_patches:
            self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=1 if use_varemb else in_chans,
                                           embed_dim=embed_dim, twoD=twoD, sqrt_len_method=sqrt_len_method)
            num_patches = self.patch_embed.num_patches
            self.grid_size = self.patch_embed.grid_size
        else:
            num_patches = fixed_length
        self.num_patches = num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if class_token else None
        self.embed_len = num_patches + self.num_prefix_tokens
        if not pos_embed or pos_embed == 'none':
            self.pos_embed = None
        else:
            self.pos_embed = nn.Parameter(torch.randn(1, self.embed_len, embed_dim) * .02)
        self.pos_drop = nn.Dropout(p=pos_drop_rate)
        self.patch_drop = PatchDropout(patch_drop_rate, num_prefix_tokens=self.num_prefix_tokens) if patch_drop_rate > 0 else nn.Identity()

        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.Sequential(*[
            block_fn(dim=embed_dim, num_heads=num_heads, fused_attn=FusedAttn_option, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                     qk_norm=qk_norm, init_values=init_values, proj_drop=proj_drop_rate, attn_drop=attn_drop_rate,
                     drop_path=dpr[i], norm_layer=norm_layer, act_layer=act_layer, mlp_layer=mlp_layer)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head_drop = nn.Dropout(drop_rate)
        if num_classes is not None:
            self.head = Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        else:
            self.head = None

        nd = 2 if twoD else 3
        self.patch_dim = in_chans * patch_size ** nd
        self.patch_dim_woc = patch_size ** nd

        if tokens_from_patches:
            if use_varemb:
                self.token_embeds = nn.ModuleList([self.patch_embed for _ in range(len(default_vars))])
            else:
                self.token_embeds = self.patch_embed   # alias: state_dict lists both names
        else:
            def seq(k):
                return nn.Sequential(LayerNorm(k), Linear(k, embed_dim), LayerNorm(embed_dim))
            if use_varemb:
                self.token_embeds = nn.ModuleList([seq(self.patch_dim_woc) for _ in range(len(default_vars))])
            else:
                self.token_embeds = seq(self.patch_dim)

        if use_varemb:
            self.var_embed, self.var_map = self.create_var_embedding(embed_dim)
            if single_channel or len(default_vars) == 1:
                self.var_query = None
                self.var_agg = None
            else:
                self.var_query = nn.Parameter(torch.zeros(1, self.aggregated_variables, embed_dim), requires_grad=True)
                self.var_agg = VariableMapping_Attention(embed_dim, fused_attn=FusedAttn_option, num_heads=num_heads, qkv_bias=False)

        if use_adaptive_pos_emb:
            self.adaptive_pos_dep_emb = nn.Sequential(Linear(in_features=3 if twoD else 4, out_features=embed_dim), nn.GELU())

        if weight_init != 'skip':
            self.init_weights('')

    # ------------------------------------------------------------------ init
    def _grid(self):
        return [int(s / self.patch_size) for s in self.img_size]

    def _sincos_table(self, dim, cls_token):
        g = self._grid()
        if self.twoD:
            return get_2d_sincos_pos_embed(dim, g[0], g[1], cls_token=cls_token)
        return get_3d_sincos_pos_embed(dim, g[0], g[1], g[2], cls_token=cls_token)

    def _init_common(self):
        if self.cls_token is not None:
            nn.init.normal_(self.cls_token, std=1e-6)
        if not self.adaptive_patching:
            embeds = list(self.token_embeds) if self.use_varemb else [self.token_embeds]
            for te in embeds:
                w = te.proj.weight.data
                trunc_normal_(w.view([w.shape[0], -1]), std=0.02)
        if self.use_varemb:
            table = get_1d_sincos_pos_embed_from_grid(self.var_embed.shape[-1], np.arange(len(self.default_vars)))
            self.var_embed.data.copy_(torch.from_numpy(table).float().unsqueeze(0))
        named_apply(get_init_weights_vit(0.), self)

    def init_weights(self, mode: str = '') -> None:
        if (not self.adaptive_patching or self.sqrt_len_method) and self.pos_embed is not None:
            table = self._sincos_table(self.pos_embed.shape[-1], self.class_token)
            self.pos_embed.data.copy_(torch.from_numpy(table).float().unsqueeze(0))
        self._init_common()

    # ------------------------------------------------------------------ helpers
    def set_compute_dtype(self, dtype):
        return set_compute_dtype(self, dtype)

    def _prepare(self):
        """flat master/grad/shadow buffers (re)built if the parameters moved; bf16 shadow refreshed if weights changed"""
        st = ensure_store(self)
        if _cd(self) == torch.bfloat16:
            st.refresh_shadow()
        return st

    def create_var_embedding(self, dim):
        var_map = {v: i for i, v in enumerate(self.default_vars)}
        return nn.Parameter(torch.zeros(1, len(self.default_vars), dim), requires_grad=True), var_map

    def _embed_tokens(self, x, variables):
        if self.use_varemb:
            return self._embed_variables(x, variables)
        if self.adaptive_patching and not self.sqrt_len_method:
            # reference :465-467: x [B, C, S, P] arrives already cut into S resized patches -> rows (p c) -> LN, Linear, LN
            if x.dim() != 4 or x.shape[1] * x.shape[3] != self.patch_dim:
                raise ValueError(f"adaptive_patching expects x [B, C={self.in_chans}, S, P={self.patch_dim_woc}], got {tuple(x.shape)}")
            return self.token_embeds(HF.SeqPatchesFn.apply(x, _cd(self)))
        return self.token_embeds(x)

    def get_var_ids(self, variables):
        return [self.var_map[v] for v in variables]

    def _embed_variables(self, x, variables):
        """use_varemb front end (reference :434-462): every input channel is tokenised on its own by the token embedding of ITS variable,
        gets that variable's embedding and — unless there is a single channel — the V embeddings of a token are aggregated by the
        cross-attention with the learnt query (aggregate_variables, :414-432).  The per-variable embeddings run on the HIP kernels;
        stacking them (variable-major, so nothing is permuted) and adding the variable embedding are two element-wise torch ops."""
        ids = self.get_var_ids(tuple(variables))
        adaptive = self.adaptive_patching and not self.sqrt_len_method
        if x.shape[1] < (1 if self.single_channel else len(ids)):
            raise ValueError(f"use_varemb: {len(ids)} variables but input has {x.shape[1]} channels")
        toks = []
        for i, vid in enumerate(ids):
            xi = x[:, i] if adaptive else x[:, i:i + 1]           # [B, S, P] pre-cut patches of this channel | a 1-channel image
            toks.append(self.token_embeds[vid](xi))               # [B, L, D]
            if self.single_channel:
                break
        ve = self.var_embed[0, ids[:len(toks)]].to(toks[0].dtype)  # [V, D]
        if self.single_channel or self.var_agg is None:
            return toks[0] + ve[0]
        B, L, D = toks[0].shape
        xs = torch.stack(toks, dim=0) + ve.view(-1, 1, 1, D)      # [V, B, L, D]
        out = self.var_agg(self.var_query, xs.view(len(toks), B * L, D))
        return out.view(B, L, D)

    def _pos_embed(self, x: torch.Tensor, seq_ps) -> torch.Tensor:
        if self.pos_embed is None:
            return x.view(x.shape[0], -1, x.shape[-1])      # reference :367-368 returns before the class token is attached
        if self.pos_drop.p > 0.0 and self.training:
            raise NotImplementedError("pos_drop_rate > 0 is not on the HIP hot path")
        if self.use_adaptive_pos_emb:
            if seq_ps is None:
                raise ValueError("use_adaptive_pos_emb needs seq_ps [B, S, 3|4] (position and size of every token)")
            lin = self.adaptive_pos_dep_emb[0]
            return HF.AdaptivePosFn.apply(x, seq_ps, lin.weight, lin.bias, self.cls_token, _cd(self))
        if self.pos_drop.p > 0.0 and self.training:
            raise NotImplementedError("pos_drop_rate > 0 is not on the HIP hot path")
        return HF.TokensFn.apply(x, self.cls_token, self.pos_embed, _cd(self))

    # ------------------------------------------------------------------ forward
    def forward_features(self, x: torch.Tensor, variables, seq_ps) -> torch.Tensor:
        self._prepare()
        x = self._embed_tokens(x, variables)
        x = self._pos_embed(x, seq_ps)
        x = self.patch_drop(x)
        x = self.blocks(x)
        return self.norm(x)

    def pool(self, x: torch.Tensor) -> torch.Tensor:
        return global_pool_nlc(x, num_prefix_tokens=self.num_prefix_tokens)

    def forward_head(self, x: torch.Tensor) -> torch.Tensor:
        x = self.pool(x)
        if self.head_drop.p > 0.0 and self.training:
            raise NotImplementedError("drop_rate > 0 is not on the HIP hot path")
        return self.head(x)

    def forward(self, x: torch.Tensor, variables, seq_ps=None) -> torch.Tensor:
        return self.forward_head(self.forward_features(x, variables, seq_ps))


class SAP(VIT):
    """Segmentation head on the ViT encoder (reference :491-536): ConvTranspose neck + 1x1 conv (MIOpen; SURVEY §2 out of scope)."""

    def __init__(self, *args, **kwargs):
        self.sqrt_len = kwargs.pop('sqrt_len', '')
        super().__init__(*args, **kwargs)
        self.head = None
        p = self.patch_size
        if self.twoD:
            self.neck = nn.Sequential(nn.ConvTranspose2d(self.embed_dim, 256, kernel_size=(p, p), stride=(p, p), bias=False))
            self.mask_header = nn.Sequential(nn.Conv2d(256, self.num_classes, 1))
        else:
            self.neck = nn.Sequential(nn.ConvTranspose3d(self.embed_dim, 256, kernel_size=(p, p, p), stride=(p, p, p), bias=False))
            self.mask_header = nn.Sequential(nn.Conv3d(256, self.num_classes, 1))
        self.init_weights('')

    def mask_head(self, x: torch.Tensor):
        s = self.sqrt_len
        B, _, C = x.shape
        x = x.float()
        if self.twoD:
            x = x.reshape(B, s, s, C).permute(0, 3, 1, 2)
        else:
            x = x.reshape(B, s, s, s, C).permute(0, 4, 1, 2, 3)
        return self.mask_header(self.neck(x))

    def forward_head(self, x: torch.Tensor) -> torch.Tensor:
        return self.mask_head(self.pool(x))


class MAE(VIT):
    """Masked auto-encoder (reference :538-755).  forward(x, variables, seq_ps=None) -> (pred [B,L,p^nd*C], mask [B,L])."""

    def __init__(self, *args, **kwargs):
        self.mask_ratio = kwargs.pop('mask_ratio', '')
        self.linear_decoder = kwargs.pop('linear_decoder', '')
        self.decoder_depth = kwargs.pop('decoder_depth', '')
        self.decoder_embed_dim = kwargs.pop('decoder_embed_dim', '')
        self.decoder_num_heads = kwargs.pop('decoder_num_heads', '')
        self.mlp_ratio_decoder = kwargs.pop('mlp_ratio_decoder', '')
        super().__init__(*args, **kwargs)
        self.head = None
        dd = self.embed_dim if self.linear_decoder else self.decoder_embed_dim
        self.decoder_pred = Linear(dd, self.patch_dim)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, dd))
        if not self.linear_decoder:
            self.decoder_embed = Linear(self.embed_dim, dd)
            self.decoder_norm = LayerNorm(dd)   # default eps 1e-5, like the reference's nn.LayerNorm(decoder_embed_dim)
            if self.use_adaptive_pos_emb:
                self.decoder_pos_embed = None
            elif self.adaptive_patching:
                self.decoder_pos_embed = nn.Parameter(torch.randn(1, self.num_patches, dd) * .02)
            else:
                self.decoder_pos_embed = nn.Parameter(torch.zeros(1, self.num_patches, dd))
            dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.decoder_depth)]
            self.decoder_blocks = nn.Sequential(*[
                self.block_fn(dim=dd, num_heads=self.decoder_num_heads, fused_attn=self.FusedAttn_option,
                              mlp_ratio=self.mlp_ratio_decoder, qkv_bias=self.qkv_bias, qk_norm=self.qk_norm,
                              init_values=self.init_values, proj_drop=self.proj_drop_rate, attn_drop=self.attn_drop_rate,
                              drop_path=dpr[i], norm_layer=self.norm_layer, act_layer=self.act_layer, mlp_layer=self.mlp_layer)
                for i in range(self.decoder_depth)])
            if self.use_adaptive_pos_emb:
                self.decoder_adaptive_pos_dep_emb = nn.Sequential(Linear(in_features=3 if self.twoD else 4, out_features=dd), nn.GELU())
        else:
            self.decoder_pos_embed = None
        self.init_weights('')

    def init_weights(self, mode: str = '') -> None:
        if not self.adaptive_patching:
            if self.pos_embed is not None:
                self.pos_embed.data.copy_(torch.from_numpy(self._sincos_table(self.pos_embed.shape[-1], False)).float().unsqueeze(0))
            if getattr(self, 'decoder_pos_embed', None) is not None:
                t = self._sincos_table(self.decoder_pos_embed.shape[-1], False)
                self.decoder_pos_embed.data.copy_(torch.from_numpy(t).float().unsqueeze(0))
        self._init_common()

    def random_masking(self, sequence, noise=None):
        """per-sample random masking by argsort of uniform noise; returns (kept tokens, mask [B,L] (1 = masked), ids_restore).
        Index math, gather and mask are one HIP kernel pair (ucfvit_mae_mask + ucfvit_gather_rows), bit-exact w.r.t. argsort."""
        if sequence.dim() != 3:
            raise NotImplementedError("aggregated_variables > 1 is not on the HIP hot path")
        B, L, _ = sequence.shape
        len_keep = int(L * (1 - self.mask_ratio))
        if noise is None:
            noise = torch.rand(B, L, device=sequence.device)
        return HF.RandomMaskFn.apply(sequence, noise, len_keep)

    def mask_head(self, x: torch.Tensor, ids_restore, seq_ps):
        if not self.linear_decoder:
            x = self.decoder_embed(x)
        pos = None if (self.linear_decoder or self.use_adaptive_pos_emb) else self.decoder_pos_embed
        x = HF.UnshuffleFn.apply(x, self.mask_token, ids_restore, pos, _cd(self))
        if not self.linear_decoder:
            if self.use_adaptive_pos_emb:           # reference :693-697: decoder positions from seq_ps as well (no class token here)
                if seq_ps is None:
                    raise ValueError("use_adaptive_pos_emb needs seq_ps [B, S, 3|4] (position and size of every token)")
                lin = self.decoder_adaptive_pos_dep_emb[0]
                x = HF.AdaptivePosFn.apply(x, seq_ps, lin.weight, lin.bias, None, _cd(self))
            x = self.decoder_norm(self.decoder_blocks(x))
        return self.decoder_pred(x)

    def forward_features(self, x: torch.Tensor, variables, seq_ps, noise=None):
        self._prepare()
        x = self._embed_tokens(x, variables)
        x = self._pos_embed(x, seq_ps)
        x, mask, ids_restore = self.random_masking(x, noise)
        x = self.patch_drop(x)
        x = self.blocks(x)
        return self.norm(x), mask, ids_restore

    def forward_head(self, x: torch.Tensor, ids_restore, seq_ps):
        return self.mask_head(self.pool(x), ids_restore, seq_ps)

    def forward(self, x: torch.Tensor, variables, seq_ps=None, noise=None):
        """`noise` ([B, L] uniform) is an extension for reproducible masks / parity tests; None = torch.rand like the reference."""
        x, mask, ids_restore = self.forward_features(x, variables, seq_ps, noise)
        return self.forward_head(x, ids_restore, seq_ps), mask


class UNETR(VIT):
    """UNETR (reference :757-1113): ViT encoder on the HIP kernels with taps after blocks depth/4, 2*depth/4, 3*depth/4
    (raw block outputs, forward_intermediates :995-1086) and the convolutional skip-connection decoder on the HIP convolution kernels
    (csrc/conv3d.hip, csrc/unetr_decoder.hip through unetr_blocks.forward_cl; layer semantics = monai's published blocks, parity unpinned).
    forward(x, variables, seq_ps=None, x_seq=None) -> [B, num_classes, *img_size]

    ONE backend: a configuration the HIP decoder kernels do not cover (2-D, feature sizes other than 16 / 32 k, the pooling decoder
    without skip connections, a patch size other than 16) raises in forward() unless the caller asked for torch's own convolutions
    explicitly with the constructor argument allow_torch_decoder=True (an extension of the reference's signature; default False)."""

    def __init__(self, *args, **kwargs):
        self.linear_decoder = kwargs.pop('linear_decoder', '')
        self.feature_size = kwargs.pop('feature_size', '')
        self.skip_connection = kwargs.pop('skip_connection', '')
        self.sqrt_len = kwargs.pop('sqrt_len', '')
        self.allow_torch_decoder = bool(kwargs.pop('allow_torch_decoder', False))
        self.force_torch_decoder = False      # tests: the same weights through torch's convolutions (needs allow_torch_decoder)
        super().__init__(*args, **kwargs)
        self.head = None
        from .unetr_blocks import UnetrBasicBlock, UnetrPrUpBlock, UnetrUpBlock, UnetOutBlock
        nd = 2 if self.twoD else 3
        if self.adaptive_patching:
            self.feat_size = (self.sqrt_len,) * nd
        else:
            self.feat_size = tuple(int(self.img_size[i] / self.patch_size) for i in range(nd))
        fs, D = self.feature_size, self.embed_dim
        if not self.linear_decoder:
            if self.skip_connection:
                self.skip_indices = [(i + 1) * (self.depth // 4) for i in range(3)]
                kw = dict(kernel_size=3, stride=1, upsample_kernel_size=2, norm_name="instance", conv_block=True, res_block=True)
                self.encoder1 = UnetrBasicBlock(nd, self.in_chans, fs, kernel_size=3, stride=1, norm_name="instance", res_block=True)
                self.encoder2 = UnetrPrUpBlock(nd, D, fs * 2, num_layer=2, **kw)
                self.encoder3 = UnetrPrUpBlock(nd, D, fs * 4, num_layer=1, **kw)
                self.encoder4 = UnetrPrUpBlock(nd, D, fs * 8, num_layer=0, **kw)
                up = dict(kernel_size=3, upsample_kernel_size=2, norm_name="instance", res_block=True)
                self.decoder5 = UnetrUpBlock(nd, D, fs * 8, **up)
                self.decoder4 = UnetrUpBlock(nd, fs * 8, fs * 4, **up)
                self.decoder3 = UnetrUpBlock(nd, fs * 4, fs * 2, **up)
                full = self.feat_size[0] * 16 == self.img_size[0]
                self.decoder2 = UnetrUpBlock(nd, fs * 2, fs, kernel_size=3, upsample_kernel_size=2 if full else 1, norm_name="instance", res_block=True)
            else:
                self.decoder5 = MyUnetBlock(nd, D, fs * 8, upsample_kernel_size=2, res_block=True)
                self.decoder4 = MyUnetBlock(nd, fs * 8, fs * 4, upsample_kernel_size=2, res_block=True)
                self.decoder3 = MyUnetBlock(nd, fs * 4, fs * 2, upsample_kernel_size=2, res_block=True)
                self.decoder2 = MyUnetBlock(nd, fs * 2, fs, upsample_kernel_size=2, res_block=True)
            self.out = UnetOutBlock(nd, fs, self.num_classes)
            if self.feat_size[0] * 16 != self.img_size[0]:
                self.upsample = nn.Upsample(size=self.img_size, mode='trilinear' if not self.twoD else 'bilinear', align_corners=True)
        else:
            self.mlp_head = Linear(D, self.num_classes)
            self.upsample = nn.Upsample(scale_factor=self.patch_size, mode='trilinear' if not self.twoD else 'bilinear', align_corners=True)
        self.init_weights('')

    def proj_feat(self, x, hidden_size, feat_size):
        x = x.float().view(x.size(0), *feat_size, hidden_size)
        return x.permute(0, len(feat_size) + 1, *range(1, len(feat_size) + 1)).contiguous()

    def forward_intermediates(self, x, variables, seq_ps, indices=None, return_prefix_tokens=False, norm=False, stop_early=False,
                              intermediates_only=False):
        take, max_index = feature_take_indices(len(self.blocks), indices)
        self._prepare()
        x = self._embed_tokens(x, variables)
        x = self._pos_embed(x, seq_ps)
        x = self.patch_drop(x)
        intermediates = []
        blocks = self.blocks if not stop_early else self.blocks[:max_index + 1]
        for i, blk in enumerate(blocks):
            x = blk(x)
            if i in take:
                intermediates.append(self.norm(x) if norm else x)
        if self.num_prefix_tokens:
            prefix = [y[:, 0:self.num_prefix_tokens] for y in intermediates]
            intermediates = [y[:, self.num_prefix_tokens:] for y in intermediates]
            if return_prefix_tokens:
                intermediates = list(zip(intermediates, prefix))
        if intermediates_only:
            return intermediates
        return self.norm(x), intermediates

    def unetr_head(self, x, intermediates, enc1):
        D, fz = self.embed_dim, self.feat_size
        if not self.skip_connection:
            if self.linear_decoder:
                x = self.mlp_head(x).float()
                x = x.view(x.size(0), *self.grid_size, -1).permute(0, len(fz) + 1, *range(1, len(fz) + 1))
                return self.upsample(x)
            out = self.decoder2(self.decoder3(self.decoder4(self.decoder5(self.proj_feat(x, D, fz)))))
            if fz[0] * 16 != self.img_size[0]:
                out = self.upsample(out)
            return self.out(out)
        n = len(intermediates)
        dec4 = self.proj_feat(x, D, fz)
        dec3 = self.decoder5(dec4, self.encoder4(self.proj_feat(intermediates[n - 1], D, fz)))
        dec2 = self.decoder4(dec3, self.encoder3(self.proj_feat(intermediates[n - 2], D, fz)))
        dec1 = self.decoder3(dec2, self.encoder2(self.proj_feat(intermediates[n - 3], D, fz)))
        if fz[0] * 16 != self.img_size[0]:
            dec1 = self.upsample(dec1)
        return self.out(self.decoder2(dec1, enc1))

    def forward_head(self, x, intermediates, enc1):
        return self.unetr_head(self.pool(x), intermediates, enc1)

    def hip_decoder(self):
        """True when the convolutional decoder runs on the HIP kernels end to end (unetr_blocks.hip_decoder_supported)"""
        from .unetr_blocks import hip_decoder_supported
        if self.force_torch_decoder or self.linear_decoder or not self.skip_connection:
            return False
        full = all(self.feat_size[i] * 16 == self.img_size[i] for i in range(len(self.feat_size)))
        return full and hip_decoder_supported(2 if self.twoD else 3, self.in_chans, self.embed_dim, self.feature_size)

    def _tokens_cl(self, t):
        """token matrix [B, N, D] -> channels-last feature map [B, *feat_size, D] in bf16: a view, no permute (the decoder kernels read a
        voxel's channels contiguously, which is what a token row is)"""
        return t.to(torch.bfloat16).reshape(t.size(0), *self.feat_size, self.embed_dim)

    def _unetr_head_cl(self, x, intermediates, enc1):
        n = len(intermediates)
        dec3 = self.decoder5.forward_cl(self._tokens_cl(x), self.encoder4.forward_cl(self._tokens_cl(intermediates[n - 1])))
        dec2 = self.decoder4.forward_cl(dec3, self.encoder3.forward_cl(self._tokens_cl(intermediates[n - 2])))
        dec1 = self.decoder3.forward_cl(dec2, self.encoder2.forward_cl(self._tokens_cl(intermediates[n - 3])))
        logits = self.out.forward_cl(self.decoder2.forward_cl(dec1, enc1))            # [B, X, Y, Z, classes] fp32
        return logits.permute(0, 4, 1, 2, 3)                                          # the reference's [B, classes, X, Y, Z] as a view

    def forward(self, x, variables, seq_ps=None, x_seq=None):
        tokens_in = x_seq if self.adaptive_patching else x
        if self.skip_connection:
            if self.hip_decoder():
                if not x.is_cuda:
                    raise RuntimeError("UNETR: the convolutional decoder runs on the MI355X only; there is no CPU path")
                from UCF_VIT._hip import ops as _ops
                enc1 = self.encoder1.forward_cl(_ops.pad_channels8(x.float().contiguous()))
                feats, intermediates = self.forward_intermediates(tokens_in, variables, seq_ps, indices=self.skip_indices)
                return self._unetr_head_cl(self.pool(feats), intermediates, enc1)
            self._require_torch_decoder_opt_in()
            enc1 = self.encoder1(x)
            feats, intermediates = self.forward_intermediates(tokens_in, variables, seq_ps, indices=self.skip_indices)
            return self.forward_head(feats, intermediates, enc1)
        if not self.linear_decoder:
            self._require_torch_decoder_opt_in()
        return self.forward_head(self.forward_features(tokens_in, variables, seq_ps), None, None)

    def _require_torch_decoder_opt_in(self):
        if not self.allow_torch_decoder:
            raise RuntimeError(
                "UNETR: this configuration's convolutional decoder is not covered by the HIP convolution kernels (they need 3-D volumes, patch "
                "size 16, the skip-connection decoder, feature_size 16 or a power-of-two multiple of 32, 1-8 input channels, an embed_dim of "
                "8, 16 or a multiple of 32). Construct the model with allow_torch_decoder=True to run its convolutions on torch instead.")


class DiffusionVIT(VIT):
    """Noise-prediction ViT (reference :1115-1283).  API surface and state_dict layout only: the reference's own forward path is
    broken upstream (forward_features calls self._pos_embed(x) without seq_ps -> TypeError) and diffusion training
    is out of scope.  forward(x, t, variables) here follows the evident intent: tokens + time embedding -> blocks."""

    def __init__(self, *args, **kwargs):
        self.linear_decoder = kwargs.pop('linear_decoder', '')
        self.decoder_depth = kwargs.pop('decoder_depth', '')
        self.decoder_embed_dim = kwargs.pop('decoder_embed_dim', '')
        self.decoder_num_heads = kwargs.pop('decoder_num_heads', '')
        self.mlp_ratio_decoder = kwargs.pop('mlp_ratio_decoder', '')
        self.time_steps = kwargs.pop('time_steps', '')
        super().__init__(*args, **kwargs)
        self.head = None
        self.temporalEmbeddings = SinusoidalEmbeddings(time_steps=self.time_steps, embed_dim=self.embed_dim)
        self.timeEmbeddingMap = EmbeddingDenseLayer(self.embed_dim, self.embed_dim, 0.5)
        dd = self.embed_dim if self.linear_decoder else self.decoder_embed_dim
        self.decoder_pred = Linear(dd, self.patch_dim)
        if not self.linear_decoder:
            self.decoder_embed = Linear(self.embed_dim, dd)
            self.decoder_norm = LayerNorm(dd)
            if self.adaptive_patching:
                self.decoder_pos_embed = nn.Parameter(torch.randn(1, self.num_patches, dd) * .02)
            else:
                self.decoder_pos_embed = nn.Parameter(torch.zeros(1, self.num_patches, dd))
            dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.decoder_depth)]
            self.decoder_blocks = nn.Sequential(*[
                self.block_fn(dim=dd, num_heads=self.decoder_num_heads, fused_attn=self.FusedAttn_option, mlp_ratio=self.mlp_ratio_decoder,
                              qkv_bias=self.qkv_bias, qk_norm=self.qk_norm, init_values=self.init_values, proj_drop=self.proj_drop_rate,
                              attn_drop=self.attn_drop_rate, drop_path=dpr[i], norm_layer=self.norm_layer, act_layer=self.act_layer,
                              mlp_layer=self.mlp_layer) for i in range(self.decoder_depth)])
        else:
            self.decoder_pos_embed = None
        self.init_weights('')

    init_weights = MAE.init_weights

    def forward_features(self, x, t, variables):
        self._prepare()
        x = self._embed_tokens(x, variables)
        x = self._pos_embed(x, None)
        time_emb = self.timeEmbeddingMap(self.temporalEmbeddings(x, t).float())[:, None, :]
        x = x + time_emb.to(x.dtype)
        return self.norm(self.blocks(x))

    def forward_head(self, x):
        x = self.pool(x)
        if not self.linear_decoder:
            x = self.decoder_embed(x)
            x = x + self.decoder_pos_embed.to(x.dtype)
            x = self.decoder_norm(self.decoder_blocks(x))
        return self.decoder_pred(x)

    def forward(self, x, t, variables):
        return self.forward_head(self.forward_features(x, t.to('cpu'), variables))

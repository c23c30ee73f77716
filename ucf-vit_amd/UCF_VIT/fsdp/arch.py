"""Hybrid-OP model layer — drop-in for the reference's src/UCF_VIT/fsdp/arch.py: the `simple` models plus
tensor_par_size / tensor_par_group (reference :137-138), tensor-parallel Blocks, the activation broadcast that makes all
ranks of a TP group work on the TP-source rank's tokens (reference :476-486: dist.broadcast before `blocks`,
F_Identity_B_Broadcast after `norm`) and the TP-synchronised MAE mask noise (:682-689).

Deviations from the reference's fsdp copy, which carries known defects: the image path follows the `simple`
condition (`adaptive_patching and not sqrt_len_method`), and VIT.forward_head applies `head`.
"""
from functools import partial

import torch
import torch.distributed as dist

from UCF_VIT.simple import arch as _S
from UCF_VIT.simple.arch import (  # noqa: F401
    feature_take_indices, init_weights_vit_timm, get_init_weights_vit, global_pool_nlc, named_apply)
from UCF_VIT.utils.dist_functions import F_Identity_B_Broadcast
from UCF_VIT.utils.fused_attn import FusedAttn
from .building_blocks import Block, Mlp


class _TPMixin:
    def _tp_setup(self, kwargs):
        self.tensor_par_size = kwargs.pop('tensor_par_size', 1)
        self.tensor_par_group = kwargs.pop('tensor_par_group', None)
        kwargs.setdefault('block_fn', partial(Block, tensor_par_size=self.tensor_par_size, tensor_par_group=self.tensor_par_group))
        kwargs.setdefault('mlp_layer', Mlp)
        return kwargs

    def _tp_src(self):
        return dist.get_rank() - dist.get_rank(group=self.tensor_par_group)

    def _tp_enter(self, x):
        """all ranks of the TP group continue with the source rank's tokens (C5, reference :476-479)"""
        if self.tensor_par_size > 1:
            x = x.contiguous()
            dist.broadcast(x, self._tp_src(), group=self.tensor_par_group)
        return x

    def _tp_exit(self, x):
        if self.tensor_par_size > 1:
            x = F_Identity_B_Broadcast(x, self._tp_src(), group=self.tensor_par_group)
        return x


class VIT(_TPMixin, _S.VIT):
    def __init__(self, *args, **kwargs):
        kwargs = self._tp_setup(kwargs)
        super().__init__(*args, **kwargs)

    def forward_features(self, x, variables, seq_ps):
        self._prepare()
        x = self._embed_tokens(x, variables)
        x = self._pos_embed(x, seq_ps)
        x = self.patch_drop(x)
        x = self._tp_enter(x)
        x = self.norm(self.blocks(x))
        return self._tp_exit(x)


class MAE(_TPMixin, _S.MAE):
    def __init__(self, *args, **kwargs):
        kwargs = self._tp_setup(kwargs)
        super().__init__(*args, **kwargs)

    def random_masking(self, sequence, noise=None):
        if noise is None and self.tensor_par_size > 1:
            noise = torch.rand(sequence.shape[0], sequence.shape[1], device=sequence.device)
            dist.broadcast(noise, src=self._tp_src(), group=self.tensor_par_group)    # C6: same mask on every TP rank
        return super().random_masking(sequence, noise)

    def forward_features(self, x, variables, seq_ps, noise=None):
        self._prepare()
        x = self._embed_tokens(x, variables)
        x = self._pos_embed(x, seq_ps)
        x, mask, ids_restore = self.random_masking(x, noise)
        x = self.patch_drop(x)
        x = self._tp_enter(x)
        x = self.norm(self.blocks(x))
        return self._tp_exit(x), mask, ids_restore

    def mask_head(self, x, ids_restore, seq_ps):
        if self.tensor_par_size <= 1 or self.linear_decoder:
            return super().mask_head(x, ids_restore, seq_ps)
        from UCF_VIT._hip import functional as HF
        from UCF_VIT.simple.building_blocks import _cd
        x = self.decoder_embed(x)
        x = HF.UnshuffleFn.apply(x, self.mask_token, ids_restore, self.decoder_pos_embed, _cd(self))
        x = self._tp_enter(x)
        x = self.decoder_norm(self.decoder_blocks(x))
        x = self._tp_exit(x)
        return self.decoder_pred(x)

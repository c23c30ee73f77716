"""Sequence parallelism for the long-sequence encoder (UNETR 512x512x128: N = 8192 tokens) — NEW capability: the reference only
constructs seq_par_group and asserts seq_par_size == 1 everywhere (training_scripts/train_masked_fsdp.py:220, utils/misc.py:147-160,
README.md:993), so the parity target is the unsharded operator (tests/test_sp.py).

Scheme (Ulysses-style, chosen for the MI355X node: 8 GPUs fully connected by xGMI, all-to-all uses all 7 links of a GPU at once):
tokens are sharded contiguously over the P ranks of the group ([B, N/P, D] per rank).  LayerNorm, the qkv/proj/fc GEMMs and the
MLP are token-local and need no communication.  Around the attention core two all-to-alls swap the sharded axis:
    qkv  [B, N/P, 3, H, dh]  --all-to-all-->  [B, N, 3, H/P, dh]   (every rank: all tokens, its own heads)
    out  [B, N, H/P * dh]    --all-to-all-->  [B, N/P, H * dh]
and the fused gfx950 attention kernel runs unchanged on H/P heads.  Backward applies the same two exchanges in reverse.
Requirements: N % P == 0 and H % P == 0.  Parameter gradients are partial sums over the local tokens: reduce them over the
sequence-parallel group together with the data-parallel reduction (HipDataParallel over the dp x sp group).
The pack/unpack permutes around the collective are plain strided copies (torch) — data movement, no arithmetic.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from UCF_VIT._hip import functional as HF
from UCF_VIT._hip import ops
from UCF_VIT._hip.params import compute_param
from UCF_VIT.simple.building_blocks import _cd


def _all_to_all(t, group):
    """t: [P, ...] contiguous; chunk r goes to rank r; returns [P, ...] with chunk r received from rank r"""
    out = torch.empty_like(t)
    if t.is_cuda and dist.get_backend(group) == "gloo":       # test transport (ranks sharing one GPU): stage through the host
        h_in, h_out = t.float().cpu(), torch.empty(t.shape, dtype=torch.float32)
        dist.all_to_all_single(h_out, h_in, group=group)
        out.copy_(h_out.to(t.dtype))
        return out
    dist.all_to_all_single(out, t, group=group)               # RCCL over xGMI
    return out


def _seq_to_heads(qkv, B, Nl, H, dh, P, group):
    """[B*Nl, 3*H*dh] (local tokens, all heads) -> [B*N, 3*(H/P)*dh] (all tokens, local heads)"""
    Hl = H // P
    send = qkv.view(B, Nl, 3, P, Hl, dh).permute(3, 0, 1, 2, 4, 5).contiguous()          # [P, B, Nl, 3, Hl, dh]
    recv = _all_to_all(send, group)                                                       # chunk r = tokens of rank r
    return recv.permute(1, 0, 2, 3, 4, 5).contiguous().view(B * P * Nl, 3 * Hl * dh)


def _heads_to_seq(qkv_h, B, Nl, H, dh, P, group):
    """adjoint of _seq_to_heads"""
    Hl = H // P
    send = qkv_h.view(B, P, Nl, 3, Hl, dh).permute(1, 0, 2, 3, 4, 5).contiguous()        # [P, B, Nl, 3, Hl, dh]
    recv = _all_to_all(send, group)                                                       # chunk r = head group of rank r
    return recv.permute(1, 2, 3, 0, 4, 5).contiguous().view(B * Nl, 3 * H * dh)


def _out_heads_to_seq(o_h, B, Nl, H, dh, P, group):
    """[B*N, (H/P)*dh] -> [B*Nl, H*dh]"""
    Hl = H // P
    send = o_h.view(B, P, Nl, Hl, dh).permute(1, 0, 2, 3, 4).contiguous()                # [P, B, Nl, Hl, dh]
    recv = _all_to_all(send, group)
    return recv.permute(1, 2, 0, 3, 4).contiguous().view(B * Nl, H * dh)


def _out_seq_to_heads(do, B, Nl, H, dh, P, group):
    Hl = H // P
    send = do.view(B, Nl, P, Hl, dh).permute(2, 0, 1, 3, 4).contiguous()
    recv = _all_to_all(send, group)
    return recv.permute(1, 0, 2, 3, 4).contiguous().view(B * P * Nl, Hl * dh)


class SeqParallelAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, qkvw, qkvb, projw, projb, num_heads, cdtype, group):
        P = dist.get_world_size(group)
        xin = HF._as(x, cdtype)
        B, Nl, D = xin.shape
        H, dh = num_heads, D // num_heads
        assert H % P == 0, "num_heads must be divisible by the sequence-parallel size"
        x2 = xin.view(B * Nl, D)
        c = lambda p: compute_param(p, cdtype)
        qkv = ops.linear_fwd(x2, c(qkvw), c(qkvb))
        qkv_h = _seq_to_heads(qkv, B, Nl, H, dh, P, group)
        o_h, lse = ops.attention_fwd(qkv_h, B, P * Nl, H // P, dh, dh ** -0.5)
        o = _out_heads_to_seq(o_h, B, Nl, H, dh, P, group)
        y = ops.linear_fwd(o, c(projw), c(projb))
        ctx.save_for_backward(x2, qkv_h, o_h, lse, o, qkvw, qkvb, projw, projb)
        ctx.meta = (B, Nl, H, dh, P, cdtype, x.dtype, group)
        return y.view(B, Nl, D)

    @staticmethod
    def backward(ctx, dy):
        x2, qkv_h, o_h, lse, o, qkvw, qkvb, projw, projb = ctx.saved_tensors
        B, Nl, H, dh, P, cdtype, in_dtype, group = ctx.meta
        c = lambda p: compute_param(p, cdtype)
        dy2 = HF._as(dy, cdtype).reshape(x2.shape)
        need = ctx.needs_input_grad
        g_projw = HF._wgrad(projw, dy2, o) if need[3] else None
        g_projb = HF._bgrad(projb, dy2) if (projb is not None and need[4]) else None
        do = HF._dgrad(dy2, projw, c(projw))
        do_h = _out_seq_to_heads(do, B, Nl, H, dh, P, group)
        dqkv_h = ops.attention_bwd(qkv_h, o_h, do_h, lse, B, P * Nl, H // P, dh, dh ** -0.5)
        dqkv = _heads_to_seq(dqkv_h, B, Nl, H, dh, P, group)
        g_qkvw = HF._wgrad(qkvw, dqkv, x2) if need[1] else None
        g_qkvb = HF._bgrad(qkvb, dqkv) if (qkvb is not None and need[2]) else None
        dx = HF._dgrad(dqkv, qkvw, c(qkvw))
        return HF._ret_grad(dx.view(B, Nl, -1), in_dtype), g_qkvw, g_qkvb, g_projw, g_projb, None, None, None


class SeqParallelBlock(nn.Module):
    """Runs an existing (unsharded-parameter) Block on a token shard: x [B, N/P, D] -> [B, N/P, D]."""

    def __init__(self, block, seq_par_group):
        super().__init__()
        self.block, self.group = block, seq_par_group

    def forward(self, x):
        b = self.block
        a = b.attn
        h = SeqParallelAttentionFn.apply(b.norm1(x), a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias, a.num_heads, _cd(b), self.group)
        x = x + h if x.dtype == h.dtype else x.to(h.dtype) + h
        return x + b.mlp(b.norm2(x))

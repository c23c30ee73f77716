"""Sequence parallelism for the long-sequence encoder (UNETR 512x512x128: N = 8192 tokens) — NEW capability: the reference only
constructs seq_par_group and asserts seq_par_size == 1 everywhere (training_scripts/train_masked_fsdp.py:220, utils/misc.py:147-160,
README.md:993), so the parity target is the unsharded operator (tests/test_sp.py).

Tokens are sharded contiguously over the P ranks of the sequence-parallel group ([B, N/P, D] per rank).  LayerNorm, the qkv / proj / fc
GEMMs and the MLP are token-local and need no communication.  Around the attention core the group is used as a 2-D grid
P = P_r x P_u (sp rank = r * P_u + u), chosen for the MI355X node — 8 GPUs fully connected by xGMI, 7 links per GPU:

  * P_u ranks (same r): Ulysses exchange.  Two all-to-alls swap the sharded axis,
        qkv  [B, N/P, 3, H, dh]   --all-to-all-->  [B, N/P_r, 3, H/P_u, dh]   (the tokens of ring block r, this rank's heads)
        out  [B, N/P_r, H/P_u*dh] --all-to-all-->  [B, N/P, H*dh]
    an all-to-all drives all links of a GPU at once; it needs H % P_u == 0.
  * P_r ranks (same u): ring exchange of the K / V blocks (P_r - 1 steps of [B, N/P_r, 2, H/P_u, dh]); every step is one launch of the
    cross-block attention kernel (ucfvit_attention_cross_fwd) whose partial result is folded into the running one by the log-sum-exp
    merge (ucfvit_attention_merge).  Backward sends each K / V block round the ring once more together with its fp32 gradient
    accumulator (ucfvit_attention_cross_bwd adds this rank's terms), so no gradient is ever summed in bf16.
    A ring needs no divisibility of H; on xGMI one ring step moves 3 MB per rank over ONE link (~20 us at 153 GB/s) against ~4 us for the
    all-to-all of the same bytes over seven, so P_u is taken as large as H allows: H = 12 on 8 ranks -> P_u = 4, P_r = 2.

Parameter gradients are partial sums over the local tokens: reduce them over the sequence-parallel group together with the
data-parallel reduction — HipDataParallel over the dp x sp ranks, which takes a MEAN.  Two cases: (i) encoder only, every rank's loss
a mean over its LOCAL tokens: the mean over ranks of the local gradients is the gradient of the global mean; (ii) the whole UNETR, whose
loss every rank of the group computes identically behind gather_tokens_autograd: that function scales the gradient slices it hands
back by P, so the same mean over ranks yields the SUM of the shard contributions for the encoder and leaves the replicated decoder's
rank-identical gradients as they are (GatherTokensFn).  The pack / unpack permutes around the all-to-all are plain strided copies (torch) — data movement,
no arithmetic; with B = 1 (the volume workload) the gathered side needs none: [P_u][1][N/P] IS the token order.
"""
import math

import torch
import torch.distributed as dist
import torch.nn as nn

from UCF_VIT._hip import functional as HF
from UCF_VIT._hip import ops
from UCF_VIT._hip.params import compute_param
from UCF_VIT.simple.building_blocks import _cd


# ---------------------------------------------------------------------------------------------------------------- groups
class SeqParallelGroups:
    """the 2-D view of one sequence-parallel group.  `ulysses_group`: the P_u ranks with this rank's r; `ring_ranks`: global ranks of the
    P_r ranks with this rank's u, in ring order.  Built by make_seq_parallel_groups (dist.new_group is collective over the whole world)."""

    def __init__(self, sp_group, size, rank, ulysses_size, ulysses_group, ring_group, ring_ranks):
        self.sp_group, self.size, self.rank = sp_group, size, rank
        self.pu, self.pr = ulysses_size, size // ulysses_size
        self.r, self.u = rank // ulysses_size, rank % ulysses_size
        self.ulysses_group, self.ring_group, self.ring_ranks = ulysses_group, ring_group, ring_ranks

    @property
    def ring_next(self):
        return self.ring_ranks[(self.r + 1) % self.pr]

    @property
    def ring_prev(self):
        return self.ring_ranks[(self.r - 1) % self.pr]


def default_ulysses_size(num_heads, sp_size):
    """largest P_u that divides both the head count and the group size (all-to-all wherever the heads allow it)"""
    return math.gcd(num_heads, sp_size)


def make_seq_parallel_groups(sp_rank_lists, num_heads, ulysses_size=None):
    """EVERY rank of the world calls this with the same arguments.  sp_rank_lists: the global ranks of every sequence-parallel group
    (e.g. [[0..7]] for one group on an 8-GPU node, utils/misc.init_par_groups' seq-parallel layout otherwise).
    Returns this rank's SeqParallelGroups (None if it is in no list)."""
    me = dist.get_rank()
    mine = None
    for ranks in sp_rank_lists:
        P = len(ranks)
        pu = ulysses_size or default_ulysses_size(num_heads, P)
        if P % pu or num_heads % pu:
            raise ValueError(f"ulysses size {pu} must divide the group size {P} and the head count {num_heads}")
        pr = P // pu
        sp_group = dist.new_group(ranks)
        ug, rg, rr = {}, {}, {}
        for r in range(pr):
            ug[r] = dist.new_group([ranks[r * pu + u] for u in range(pu)])
        for u in range(pu):
            rr[u] = [ranks[r * pu + u] for r in range(pr)]
            rg[u] = dist.new_group(rr[u])
        if me in ranks:
            k = ranks.index(me)
            mine = SeqParallelGroups(sp_group, P, k, pu, ug[k // pu], rg[k % pu], rr[k % pu])
    return mine


# ---------------------------------------------------------------------------------------------------------------- transport
def _host_staged(group):
    return dist.get_backend(group) == "gloo"       # test transport (ranks sharing one GPU): RCCL refuses that, stage through the host


def _all_to_all(t, group):
    """t: [P, ...] contiguous; chunk r goes to rank r; returns [P, ...] with chunk r received from rank r"""
    out = torch.empty_like(t)
    if t.is_cuda and _host_staged(group):
        h_in, h_out = t.float().cpu(), torch.empty(t.shape, dtype=torch.float32)
        dist.all_to_all_single(h_out, h_in, group=group)
        out.copy_(h_out.to(t.dtype))
        return out
    dist.all_to_all_single(out, t, group=group)               # RCCL over xGMI
    return out


def _ring_shift(tensors, spg):
    """send every tensor of `tensors` to the next rank of the ring, receive the previous rank's: returns the received list (async on RCCL:
    the returned work handles are waited for by the caller just before the data is used)"""
    recv = [torch.empty_like(t) for t in tensors]
    if tensors[0].is_cuda and _host_staged(spg.ring_group):
        hs = [t.detach().to("cpu", torch.float32) for t in tensors]
        hr = [torch.empty_like(h) for h in hs]
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, h, spg.ring_next, group=spg.ring_group) for h in hs] +
                                      [dist.P2POp(dist.irecv, h, spg.ring_prev, group=spg.ring_group) for h in hr])
        for q in reqs:
            q.wait()
        for r_, h in zip(recv, hr):
            r_.copy_(h.to(r_.dtype))
        return recv, []
    reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, t, spg.ring_next, group=spg.ring_group) for t in tensors] +
                                  [dist.P2POp(dist.irecv, r_, spg.ring_prev, group=spg.ring_group) for r_ in recv])
    return recv, reqs


def _wait(reqs):
    for q in reqs:
        q.wait()          # stream-ordered on the nccl backend: the compute stream waits, the host does not


# ---------------------------------------------------------------------------------------------------------------- Ulysses packing
def _seq_to_heads(qkv, B, Nl, H, dh, P, group):
    """[B*Nl, 3*H*dh] (local tokens, all heads) -> [B*P*Nl, 3*(H/P)*dh] (the group's tokens, local heads)"""
    if P == 1:
        return qkv
    Hl = H // P
    send = qkv.view(B, Nl, 3, P, Hl, dh).permute(3, 0, 1, 2, 4, 5).contiguous()          # [P, B, Nl, 3, Hl, dh]
    recv = _all_to_all(send, group)                                                       # chunk r = tokens of rank r
    if B == 1:
        return recv.view(P * Nl, 3 * Hl * dh)                                             # [P][1][Nl] is already the token order
    return recv.permute(1, 0, 2, 3, 4, 5).contiguous().view(B * P * Nl, 3 * Hl * dh)


def _heads_to_seq(qkv_h, B, Nl, H, dh, P, group):
    """adjoint of _seq_to_heads"""
    if P == 1:
        return qkv_h
    Hl = H // P
    send = qkv_h.view(P, 1, Nl, 3, Hl, dh) if B == 1 else qkv_h.view(B, P, Nl, 3, Hl, dh).permute(1, 0, 2, 3, 4, 5).contiguous()
    recv = _all_to_all(send.contiguous(), group)                                          # chunk r = head group of rank r
    return recv.view(P, B, Nl, 3, Hl, dh).permute(1, 2, 3, 0, 4, 5).contiguous().view(B * Nl, 3 * H * dh)


def _out_heads_to_seq(o_h, B, Nl, H, dh, P, group):
    """[B*P*Nl, (H/P)*dh] -> [B*Nl, H*dh]"""
    if P == 1:
        return o_h
    Hl = H // P
    send = o_h.view(P, 1, Nl, Hl, dh) if B == 1 else o_h.view(B, P, Nl, Hl, dh).permute(1, 0, 2, 3, 4).contiguous()
    recv = _all_to_all(send.contiguous(), group)
    return recv.view(P, B, Nl, Hl, dh).permute(1, 2, 0, 3, 4).contiguous().view(B * Nl, H * dh)


def _out_seq_to_heads(do, B, Nl, H, dh, P, group):
    if P == 1:
        return do
    Hl = H // P
    send = do.view(B, Nl, P, Hl, dh).permute(2, 0, 1, 3, 4).contiguous()
    recv = _all_to_all(send, group)
    if B == 1:
        return recv.view(P * Nl, Hl * dh)
    return recv.permute(1, 0, 2, 3, 4).contiguous().view(B * P * Nl, Hl * dh)


# ---------------------------------------------------------------------------------------------------------------- ring attention core
def _ring_attention_fwd(qkv_h, B, Nr, Hl, dh, spg):
    """qkv_h [B*Nr, 3*Hl*dh]: this ring block's tokens (q, and the K / V block this rank owns) -> (out [B*Nr, Hl*dh], lse [B, Hl, Nr])
    of the softmax over ALL P_r key blocks"""
    W = Hl * dh
    scale = dh ** -0.5
    q = qkv_h[:, :W]
    kv = qkv_h[:, W:]                                    # [B*Nr, 2W] view (row stride 3W)
    o_acc = torch.empty((B * Nr, W), dtype=torch.float32, device=qkv_h.device)
    lse = torch.empty((B, Hl, Nr), dtype=torch.float32, device=qkv_h.device)
    cur = kv
    send = kv.contiguous() if spg.pr > 1 else None       # the travelling copy
    for s in range(spg.pr):
        nxt, reqs = (None, [])
        if s + 1 < spg.pr:
            (nxt,), reqs = _ring_shift([send], spg)      # in flight while this step computes
        o_s, lse_s = ops.attention_cross_fwd(q, cur[:, :W], cur[:, W:], B, Nr, Nr, Hl, dh, scale)
        ops.attention_merge(o_acc, lse, o_s, lse_s, B, Nr, Hl, dh, first=(s == 0))
        if nxt is not None:
            _wait(reqs)
            cur = send = nxt
    out = ops.cast(o_acc, torch.empty((B * Nr, W), dtype=qkv_h.dtype, device=qkv_h.device))
    return out, lse


def _ring_attention_bwd(qkv_h, out, dout, lse, B, Nr, Hl, dh, spg):
    """gradient of _ring_attention_fwd w.r.t. qkv_h.  Each K / V block travels round the ring with its fp32 gradient accumulator; after
    P_r steps both are home again."""
    W = Hl * dh
    scale = dh ** -0.5
    dev = qkv_h.device
    q = qkv_h[:, :W]
    dq = torch.empty((B * Nr, W), dtype=torch.float32, device=dev)
    cur_kv = qkv_h[:, W:]
    travel_kv = cur_kv.contiguous() if spg.pr > 1 else None
    acc = torch.empty((2, B * Nr, W), dtype=torch.float32, device=dev)          # dK, dV of the block in hand
    for s in range(spg.pr):
        ops.attention_cross_bwd(q, cur_kv[:, :W], cur_kv[:, W:], out, dout, lse, dq, acc[0], acc[1], B, Nr, Nr, Hl, dh, scale,
                                accumulate=(s > 0))     # step 0: the accumulators are born here (dq too: += from step 1 on)
        if spg.pr > 1:
            (travel_kv, acc), reqs = _ring_shift([travel_kv, acc], spg)
            _wait(reqs)
            cur_kv = travel_kv
    # pack [dq | dk | dv] in the compute dtype (three casts into a contiguous scratch, then one strided copy: no arithmetic)
    packed = torch.empty((3, B * Nr, W), dtype=qkv_h.dtype, device=dev)
    ops.cast(dq, packed[0])
    ops.cast(acc[0], packed[1])
    ops.cast(acc[1], packed[2])
    dqkv = torch.empty((B * Nr, 3, W), dtype=qkv_h.dtype, device=dev)
    dqkv.copy_(packed.permute(1, 0, 2))
    return dqkv.view(B * Nr, 3 * W)


class SeqParallelAttentionFn(torch.autograd.Function):
    """Attention.forward (building_blocks.py:157-192) on a token shard: x [B, N/P, D] -> [B, N/P, D]"""

    @staticmethod
    def forward(ctx, x, qkvw, qkvb, projw, projb, num_heads, cdtype, spg):
        xin = HF._as(x, cdtype)
        B, Nl, D = xin.shape
        H, dh = num_heads, D // num_heads
        assert H % spg.pu == 0, "num_heads must be divisible by the Ulysses size of the sequence-parallel group"
        Hl, Nr = H // spg.pu, spg.pu * Nl
        x2 = xin.view(B * Nl, D)
        c = lambda p: compute_param(p, cdtype)
        qkv = ops.linear_fwd(x2, c(qkvw), c(qkvb))
        qkv_h = _seq_to_heads(qkv, B, Nl, H, dh, spg.pu, spg.ulysses_group)
        if spg.pr == 1:
            o_h, lse = ops.attention_fwd(qkv_h, B, Nr, Hl, dh, dh ** -0.5)
        else:
            o_h, lse = _ring_attention_fwd(qkv_h, B, Nr, Hl, dh, spg)
        o = _out_heads_to_seq(o_h, B, Nl, H, dh, spg.pu, spg.ulysses_group)
        y = ops.linear_fwd(o, c(projw), c(projb))
        ctx.save_for_backward(x2, qkv_h, o_h, lse, o, qkvw, qkvb, projw, projb)
        ctx.meta = (B, Nl, H, dh, cdtype, x.dtype, spg)
        return y.view(B, Nl, D)

    @staticmethod
    def backward(ctx, dy):
        x2, qkv_h, o_h, lse, o, qkvw, qkvb, projw, projb = ctx.saved_tensors
        B, Nl, H, dh, cdtype, in_dtype, spg = ctx.meta
        Hl, Nr = H // spg.pu, spg.pu * Nl
        c = lambda p: compute_param(p, cdtype)
        dy2 = HF._as(dy, cdtype).reshape(x2.shape)
        need = ctx.needs_input_grad
        g_projw = HF._wgrad(projw, dy2, o) if need[3] else None
        g_projb = HF._bgrad(projb, dy2) if (projb is not None and need[4]) else None
        do = HF._dgrad(dy2, projw, c(projw))
        do_h = _out_seq_to_heads(do, B, Nl, H, dh, spg.pu, spg.ulysses_group)
        if spg.pr == 1:
            dqkv_h = ops.attention_bwd(qkv_h, o_h, do_h, lse, B, Nr, Hl, dh, dh ** -0.5)
        else:
            dqkv_h = _ring_attention_bwd(qkv_h, o_h, do_h.contiguous(), lse, B, Nr, Hl, dh, spg)
        dqkv = _heads_to_seq(dqkv_h, B, Nl, H, dh, spg.pu, spg.ulysses_group)
        g_qkvw = HF._wgrad(qkvw, dqkv, x2) if need[1] else None
        g_qkvb = HF._bgrad(qkvb, dqkv) if (qkvb is not None and need[2]) else None
        dx = HF._dgrad(dqkv, qkvw, c(qkvw))
        return HF._ret_grad(dx.view(B, Nl, -1), in_dtype), g_qkvw, g_qkvb, g_projw, g_projb, None, None, None


class _PlainGroup:
    """a torch process group used as a 1-D Ulysses group (the round-1 interface of SeqParallelBlock)"""

    def __init__(self, group):
        self.sp_group = self.ulysses_group = group
        self.size = self.pu = dist.get_world_size(group)
        self.rank = self.u = dist.get_rank(group)
        self.pr, self.r, self.ring_group, self.ring_ranks = 1, 0, None, []


class SeqParallelBlock(nn.Module):
    """Runs an existing (unsharded-parameter) Block on a token shard: x [B, N/P, D] -> [B, N/P, D].  seq_par: a SeqParallelGroups, or a
    plain process group (pure Ulysses: needs num_heads % P == 0)."""

    def __init__(self, block, seq_par):
        super().__init__()
        self.block = block
        self.spg = seq_par if isinstance(seq_par, SeqParallelGroups) else _PlainGroup(seq_par)

    def forward(self, x):
        b = self.block
        a = b.attn
        h = SeqParallelAttentionFn.apply(b.norm1(x), a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias, a.num_heads, _cd(b), self.spg)
        x = x + h if x.dtype == h.dtype else x.to(h.dtype) + h
        return x + b.mlp(b.norm2(x))


def shard_tokens(x, spg):
    """[B, N, D] -> this rank's contiguous token shard [B, N/P, D]"""
    N = x.shape[1]
    assert N % spg.size == 0, "the token count must be divisible by the sequence-parallel size"
    n = N // spg.size
    return x[:, spg.rank * n:(spg.rank + 1) * n]


def gather_tokens(x_local, spg):
    """[B, N/P, D] on every rank -> [B, N, D] (test / decoder hand-over helper; not on the encoder's hot path)"""
    parts = [torch.empty_like(x_local) for _ in range(spg.size)]
    if x_local.is_cuda and _host_staged(spg.sp_group):
        hp = [torch.empty(x_local.shape, dtype=torch.float32) for _ in range(spg.size)]
        dist.all_gather(hp, x_local.detach().float().cpu(), group=spg.sp_group)
        parts = [h.to(x_local.device, x_local.dtype) for h in hp]
    else:
        dist.all_gather(parts, x_local.contiguous(), group=spg.sp_group)
    return torch.cat(parts, dim=1)


class GatherTokensFn(torch.autograd.Function):
    """differentiable gather_tokens for a consumer that every rank of the group runs IDENTICALLY on the gathered sequence (the UNETR
    convolutional decoder, replicated): forward all-gathers the shards, backward hands each rank the slice of the (rank-identical) gradient
    that belongs to its shard, SCALED BY THE GROUP SIZE.  The replicas count as one consumer, so the parameter gradients of the sharded
    encoder are partial sums that need a SUM over the group while those of the replicated decoder are rank-identical and need a MEAN;
    with the factor P in here one MEAN over the group — what HipDataParallel does over the dp x sp ranks — is right for both:
    mean_r(P * partial_r) = sum_r partial_r for the encoder, mean_r(g) = g for the decoder (tests/test_sp.py:
    test_unetr_sequence_parallel_under_data_parallel_steps_like_the_unsharded_model)."""

    @staticmethod
    def forward(ctx, x_local, spg):
        ctx.spg = spg
        ctx.n = x_local.shape[1]
        return gather_tokens(x_local, spg)

    @staticmethod
    def backward(ctx, g):
        lo = ctx.spg.rank * ctx.n
        return g[:, lo:lo + ctx.n] * float(ctx.spg.size), None


def gather_tokens_autograd(x_local, spg):
    return GatherTokensFn.apply(x_local, spg)

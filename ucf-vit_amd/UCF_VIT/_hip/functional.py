"""torch.autograd.Function wrappers that drive the HIP kernels for the UCF-VIT operator layer.

Each Function's forward/backward is a fixed sequence of libucfvit_hip.so launches (no torch math on the hot path).
Reference semantics: src/UCF_VIT/simple/building_blocks.py (PatchEmbed:78-92, Mlp:122-129, Attention:157-192,
Block:236-239) and src/UCF_VIT/simple/arch.py (_pos_embed:367-393, random_masking:663-681, mask_head:683-702).

Parameter gradients are written by the kernels straight into the flat fp32 gradient buffer of the model's
HipParamStore when there is one (see params.py); the view is handed to autograd so hooks (e.g. a DDP reducer) still fire.
"""
import os
import weakref

import torch

from . import ops
from .lib import ACT_GELU, ACT_GELU_SAVE_DERIV, ACT_NONE
from .params import compute_param, compute_param_t, grad_target


def _as(x, dtype):
    """x in the compute dtype, contiguous (kernel-side cast, no torch math)."""
    if x.dtype == dtype:
        return x if x.is_contiguous() else x.contiguous()
    x = x if x.is_contiguous() else x.contiguous()
    return ops.cast(x, torch.empty(x.shape, dtype=dtype, device=x.device))


def _ret_grad(g, like):
    """gradient handed back to autograd must carry the dtype of the forward input"""
    if g is None or g.dtype == like:
        return g
    return ops.cast(g, torch.empty(g.shape, dtype=like, device=g.device))


# ---------------------------------------------------------------------------------------------- param-grad helpers
_QUEUES = weakref.WeakSet()      # every live queue (one per HipParamStore + the stand-alone one): flush_wgrads() walks them


class WgradQueue:
    """Per-model backward context (one per HipParamStore: two models whose backward passes interleave — GAN-style training, tensor-
    parallel threads — never see each other's pending gradients or column sums).

    Weight gradients are off the critical path of backward: nothing reads them before the optimizer step (or a data-parallel
    all-reduce).  They are collected and issued as grouped launches (ucfvit_gemm_grouped: one persistent 256x256 ping-pong kernel
    over the union of the output tiles, no split-K partial sums).  One ViT-L Block has 192 tiles = 75 % of one round of 256 CUs,
    four Blocks have 768 = three whole rounds, so the queue keeps collecting ACROSS Blocks until the tile count fills whole
    rounds (>= 95 %) or 32 problems are pending, and the rest goes out when backward ends (autograd engine callback).

    Deferral across Blocks is only used for gradients that land in the flat gradient buffer (params.grad_target gives a view):
    autograd receives that view before the kernel has run, which is safe because AccumulateGrad adopts it without reading it and
    every reader of the buffer (optimizer, HipDataParallel bucket launch, grad clipping) comes after flush_wgrads()."""

    MAX_PROBLEMS = 32       # GROUP_MAX of csrc/gemm2.hip
    CUS = 256

    def __init__(self):
        self.items, self.owners = [], []
        self.tiles = 0
        self.deferrable = True
        self.callback_armed = False
        self.listeners = []          # called after every flush (HipDataParallel launches the buckets that were waiting for it)
        # one-slot hand-over of a residual-stream gradient's column sums between two autograd Functions of the SAME model (see
        # publish_stream_colsum below): (dx tensor, fp32 [D] column sums)
        self.stream_colsum = None
        self.stream_colsum_armed = False
        # Blocks per backward pass (learnt from the first pass): with a data-parallel listener the LAST group of four Blocks is flushed in
        # pairs, so that only two Blocks' gradients (ViT-L: 50 MB of bf16) are still to be all-reduced when backward ends, not four
        self.blocks_seen = 0
        self.blocks_per_pass = 0
        _QUEUES.add(self)

    def add(self, weight, dy2, x2):
        out, acc = grad_target(weight)
        o2 = out.view(out.shape[0], -1) if out is not None else None
        if o2 is None:
            o2 = torch.empty((dy2.shape[1], x2.shape[1]), dtype=torch.float32, device=dy2.device)
            ret = o2.view(weight.shape)
            self.deferrable = False          # autograd will READ this tensor (clone / accumulate): it must be complete on return
        else:
            ret = None if acc else out
        self.items.append((dy2, x2, o2, acc))
        # (o2 is a separate view object: holding `out` itself would raise its use count and make AccumulateGrad clone it)
        self.owners.append((weight, o2) if (out is not None and not acc) else None)
        self.tiles += ((dy2.shape[1] + 255) // 256) * ((x2.shape[1] + 255) // 256)
        return ret

    def end_block(self):
        """called when a Block's backward has queued its gradients"""
        self.blocks_seen += 1
        if not self.callback_armed:             # one callback per backward pass: flushes the rest and closes the Block count
            self.callback_armed = True
            self._arm()
        if not self.items:
            return
        rounds = -(-self.tiles // self.CUS)
        full = self.tiles >= 0.95 * rounds * self.CUS
        left = self.blocks_per_pass - self.blocks_seen           # Blocks of this pass still to come (unknown in the first pass: < 0)
        tail = bool(self.listeners) and self.blocks_per_pass > 0 and 0 <= left < 4 and left % 2 == 0
        if not (_DEFER_WGRAD and self.deferrable) or full or tail or len(self.items) + 4 > self.MAX_PROBLEMS:
            self.flush()

    def _arm(self):
        torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)

    def _end_of_backward(self):
        self.callback_armed = False
        if self.blocks_seen:
            self.blocks_per_pass = self.blocks_seen
        self.blocks_seen = 0
        self.flush()

    def flush(self):
        items, owners = self.items, self.owners
        self.items, self.owners, self.tiles, self.deferrable = [], [], 0, True
        for i in range(0, len(items), self.MAX_PROBLEMS):
            ops.wgrad_grouped(items[i:i + self.MAX_PROBLEMS])
        for ow in owners:
            # a gradient handed to autograd before it was computed: if AccumulateGrad took a copy instead of the view, refresh it
            if ow is not None and ow[0].grad is not None and ow[0].grad.data_ptr() != ow[1].data_ptr():
                ow[0].grad.copy_(ow[1].view(ow[0].shape))
        for ref in list(self.listeners):
            fn = ref()
            if fn is None:
                self.listeners.remove(ref)       # its HipDataParallel wrapper is gone
            else:
                fn()


# Measured (round 1, ViT-L B=166): the four weight gradients of a Block as ONE grouped 256x256 ping-pong launch take 786 us
# (1.05 PFLOP/s on 192 of the 256 CUs) against 1142 us as four 128x128 split-K launches, +9.6 % images/s on the whole step.
# (Before the KS fragment reads moved to inline asm the grouped launch ran at 0.45-0.54 PFLOP/s: hipcc drained the LDS-DMA
# prefetch in front of every ds_read_tr builtin.)  UCFVIT_WGRAD_GROUPED=0 restores the per-GEMM launches,
# UCFVIT_WGRAD_DEFER=0 keeps one launch per Block.
_GROUP_WGRAD = os.environ.get("UCFVIT_WGRAD_GROUPED", "1") != "0"
_DEFER_WGRAD = os.environ.get("UCFVIT_WGRAD_DEFER", "1") != "0"
_STANDALONE_WQ = WgradQueue()     # modules used without a flat parameter store (single operators in tests)


def queue_of_store(store):
    q = getattr(store, "_ucf_wq", None)
    if q is None:
        q = store._ucf_wq = WgradQueue()
    return q


def _queue_for(p):
    """the backward context of the model that owns parameter `p`"""
    slot = getattr(p, "_ucf_slot", None)
    if slot is not None and slot[0].owns(p, slot[1]):
        return queue_of_store(slot[0])
    return _STANDALONE_WQ


def flush_wgrads(store=None):
    """issue every pending weight-gradient launch (of one model's store, or of every live model)"""
    if store is not None:
        queue_of_store(store).flush()
        return
    for q in list(_QUEUES):
        q.flush()


def wgrads_pending(store=None):
    if store is not None:
        return bool(queue_of_store(store).items)
    return any(q.items for q in list(_QUEUES))


def add_wgrad_flush_listener(bound_method, store=None):
    q = queue_of_store(store) if store is not None else _STANDALONE_WQ
    q.listeners.append(weakref.WeakMethod(bound_method))


def _wgrad(weight, dy2, x2, queue=None):
    if _GROUP_WGRAD and queue is not None and dy2.dtype == torch.bfloat16:
        return queue.add(weight, dy2, x2)
    out, acc = grad_target(weight)
    o2 = out.view(out.shape[0], -1) if out is not None else None
    r = ops.linear_wgrad(dy2, x2, out=o2, accumulate=acc)
    if acc:
        return None
    return out if out is not None else r.view(weight.shape)


def _bgrad(bias, dy2):
    out, acc = grad_target(bias)
    r = ops.colsum(dy2, out=out, accumulate=acc)
    return None if acc else r


def _ln_bwd(dy2, x2, gamma_c, mean, rstd, weight, bias, dres=None, dx_colsum=None, dx_colsum_accumulate=False):
    ow, aw = grad_target(weight)
    ob, ab = grad_target(bias)
    if aw != ab or (ow is None) != (ob is None):  # mixed states: take the simple route
        ow = ob = None
        aw = ab = False
    dx, dg, db = ops.layernorm_bwd(dy2, x2, gamma_c, mean, rstd, dres=dres, dgamma=ow, dbeta=ob, accumulate=aw, dx_colsum=dx_colsum,
                                   dx_colsum_accumulate=dx_colsum_accumulate)
    return dx, (None if aw else dg), (None if ab else db)


# The gradient of the residual stream leaves a LayerNorm backward (dx + dres) and is the output gradient of the Linear that wrote
# into the stream before that norm: proj (inside the same Block) or fc2 (the Block before / the model's final norm).  Its column
# sums = that Linear's bias gradient are taken by the LayerNorm-backward kernel; across autograd Functions they travel in a
# one-slot cache of the model's backward context (WgradQueue.stream_colsum), keyed by the tensor autograd hands on (held strongly
# until consumed or until backward ends, so its address cannot be reused by another tensor in between).
def _publish_stream_colsum(q, dx, cs):
    q.stream_colsum = (dx, cs)
    if not q.stream_colsum_armed:
        q.stream_colsum_armed = True

        def clear():
            q.stream_colsum = None
            q.stream_colsum_armed = False
        torch.autograd.Variable._execution_engine.queue_callback(clear)


def _take_stream_colsum(q, dy2):
    """column sums of dy2 if the LayerNorm backward that produced exactly this tensor published them, else None"""
    ent = q.stream_colsum
    q.stream_colsum = None
    if ent is None or not _BIAS_FROM_EPILOGUE:
        return None
    dx, cs = ent
    if dx.data_ptr() == dy2.data_ptr() and dx.numel() == dy2.numel() and dx.dtype == dy2.dtype and cs.numel() == dy2.shape[-1]:
        return cs
    return None


def _bgrad_from(bias, cs):
    """bias gradient from ready column sums (fp32 [D]): written / accumulated into the gradient target without touching dy"""
    out, acc = grad_target(bias)
    if out is None:
        return cs
    ops.reduce_rows(cs.view(1, -1), out, accumulate=acc)
    return None if acc else out


def _dgrad(dy2, p_w, w, aux=None, aux_is_deriv=False, c_colsum=None, c_colsum_accumulate=False, out=None):
    """dx = dy·W (optionally x gelu'(aux), or x aux when aux already is the derivative); uses the transposed weight shadow
    when the flat store keeps one.  c_colsum: fp32 [K] that receives the column sums of dx (bias gradient of the layer before)."""
    wT = compute_param_t(p_w, dy2.dtype)
    if wT is not None:
        return ops.linear_dgrad_t(dy2, wT, act_grad_aux=aux, aux_is_deriv=aux_is_deriv, c_colsum=c_colsum,
                                  c_colsum_accumulate=c_colsum_accumulate, out=out)
    return ops.linear_dgrad(dy2, w, act_grad_aux=aux, aux_is_deriv=aux_is_deriv, c_colsum=c_colsum, c_colsum_accumulate=c_colsum_accumulate,
                            out=out)


_BIAS_FROM_EPILOGUE = os.environ.get("UCFVIT_BIAS_FROM_EPILOGUE", "1") != "0"   # A/B switch
_GELU_SAVE_DERIV = os.environ.get("UCFVIT_GELU_SAVE_DERIV", "1") != "0"     # A/B switch


def _saves_gelu_deriv(dtype):
    """bf16: the fc1 epilogue evaluates gelu' next to gelu (shared erfc) and saves it instead of the pre-activation, so the fc2
    data-gradient epilogue is a multiply; fp32 keeps the pre-activation (one formulation with the oracle, parity 1e-6)"""
    return dtype == torch.bfloat16 and _GELU_SAVE_DERIV


class TP:
    """tensor-parallel (Hybrid-OP) handle: process group, size, rank in group.  group "local" = no collective (the
    single-GPU shard-equivalence tests sum the partial results themselves)."""

    def __init__(self, group, size):
        self.group, self.size = group, size
        if hasattr(group, "tp_rank"):          # in-process stand-in used by the single-GPU equivalence tests
            self.rank = group.tp_rank
        elif group is None or isinstance(group, str):
            self.rank = 0
        else:
            self.rank = torch.distributed.get_rank(group)

    def all_reduce(self, t):
        if self.size <= 1 or self.group is None or isinstance(self.group, str):
            return t
        if hasattr(self.group, "all_reduce_sum"):
            return self.group.all_reduce_sum(t)
        if t.is_cuda and torch.distributed.get_backend(self.group) == "gloo":
            # test-only transport (several ranks sharing one GPU cannot use RCCL): stage through host memory
            h = t.detach().float().cpu()
            torch.distributed.all_reduce(h, op=torch.distributed.ReduceOp.SUM, group=self.group)
            t.copy_(h.to(t.dtype))
            return t
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=self.group)   # RCCL, own stream; stream-ordered
        return t


# ---------------------------------------------------------------------------------------------- raw fused sequences
# With a TP handle the weights are this rank's shards (qkv: 3*D/tp rows = H/tp heads, proj: D/tp columns, fc1: 4D/tp rows,
# fc2: 4D/tp columns; reference fsdp/building_blocks.py:123-142,169-217): the exit GEMM output is SUM-all-reduced (forward) and
# the entry gradient is SUM-all-reduced (backward).  The residual is fused on TP rank 0 only, so the sum adds it exactly once;
# the row-parallel biases are added on every rank like the reference does (bias counted tp times).
def _attn_fwd(x2, B, N, H, wqkv, bqkv, wproj, bproj, residual, tp=None):
    dh = wproj.shape[0] // H
    Hl = H // tp.size if tp else H
    qkv = ops.linear_fwd(x2, wqkv, bqkv)                               # K4: qkv GEMM + bias
    o, lse = ops.attention_fwd(qkv, B, N, Hl, dh, dh ** -0.5)          # K5: fused softmax(QKᵀ)V over the local heads
    if tp and tp.rank != 0:
        residual = None
    y = ops.linear_fwd(o, wproj, bproj, residual=residual)             # K6: proj GEMM + bias (+ residual)
    if tp:
        tp.all_reduce(y)                                               # C3: exit all-reduce
    return y, (qkv, o, lse)


def _attn_bwd(dy2, x2, saved, B, N, H, wqkv, wproj, p_qkvw, p_qkvb, p_projw, p_projb, needs, tp=None, wq=None, projb_done=None):
    """projb_done: (grad,) when the producer of dy2 (LayerNorm backward) has already written the proj bias gradient"""
    qkv, o, lse = saved
    dh = wproj.shape[0] // H
    H = H // tp.size if tp else H
    g_projw = _wgrad(p_projw, dy2, o, wq) if needs[2] else None
    if projb_done is not None:
        g_projb = projb_done[0]
    else:
        g_projb = _bgrad(p_projb, dy2) if (p_projb is not None and needs[3]) else None
    want_b = p_qkvb is not None and needs[1]
    if want_b and ops.attention_bwd_colsum_supported(B, N, H, dh, qkv.dtype):
        # the qkv bias gradient without a second read of dqkv (include/ucfvit_hip.h, ucfvit_attention_bwd_colsum): Q third = the per-image
        # column sums the backward kernel hands out, summed over the images; K third = 0 exactly; V third = the column sums of dO, taken in
        # the epilogue of the GEMM that produces dO
        Dl = H * dh
        gb, acc = grad_target(p_qkvb)
        if gb is None:
            gb, acc = torch.empty(3 * Dl, dtype=torch.float32, device=dy2.device), False
        do = _dgrad(dy2, p_projw, wproj, c_colsum=gb[2 * Dl:], c_colsum_accumulate=acc)
        dqkv, part = ops.attention_bwd(qkv, o, do, lse, B, N, H, dh, dh ** -0.5, want_colsum=True)
        ops.reduce_rows(part, gb[:2 * Dl], accumulate=acc)
        g_qkvb = None if acc else gb
    else:
        do = _dgrad(dy2, p_projw, wproj)
        dqkv = ops.attention_bwd(qkv, o, do, lse, B, N, H, dh, dh ** -0.5)
        g_qkvb = _bgrad(p_qkvb, dqkv) if want_b else None
    g_qkvw = _wgrad(p_qkvw, dqkv, x2, wq) if needs[0] else None
    dx = _dgrad(dqkv, p_qkvw, wqkv)
    if tp:
        tp.all_reduce(dx)                                              # C2: entry gradient all-reduce
    return dx, (g_qkvw, g_qkvb, g_projw, g_projb)


def _mlp_fwd(x2, w1, b1, w2, b2, residual, tp=None):
    h = torch.empty((x2.shape[0], w1.shape[0]), dtype=x2.dtype, device=x2.device)
    act = ACT_GELU_SAVE_DERIV if _saves_gelu_deriv(x2.dtype) else ACT_GELU
    a = ops.linear_fwd(x2, w1, b1, act=act, aux_out=h)                 # K7: fc1 GEMM + bias + erf-GELU (h: pre-activation, or gelu' of it)
    if tp and tp.rank != 0:
        residual = None
    y = ops.linear_fwd(a, w2, b2, residual=residual)                   # K7: fc2 GEMM + bias (+ residual)
    if tp:
        tp.all_reduce(y)                                               # C4: exit all-reduce
    return y, (h, a)


def _mlp_bwd(dy2, x2, saved, w1, w2, p_w1, p_b1, p_w2, p_b2, needs, tp=None, wq=None, dy2_colsum=None):
    """dy2_colsum: column sums of dy2 when its producer (a LayerNorm backward) published them: the fc2 bias gradient"""
    h, a = saved
    g_w2 = _wgrad(p_w2, dy2, a, wq) if needs[2] else None
    if p_b2 is not None and needs[3]:
        g_b2 = _bgrad_from(p_b2, dy2_colsum) if dy2_colsum is not None else _bgrad(p_b2, dy2)
    else:
        g_b2 = None
    # dgrad fused with gelu'(pre-activation); the fc1 bias gradient = column sums of dh comes out of the same epilogue
    need_b1 = p_b1 is not None and needs[1]
    if need_b1 and _BIAS_FROM_EPILOGUE:
        bout, bacc = grad_target(p_b1)
        if bout is None:
            bout = torch.empty(p_b1.shape, dtype=torch.float32, device=dy2.device)
        dh = _dgrad(dy2, p_w2, w2, aux=h, aux_is_deriv=_saves_gelu_deriv(h.dtype), c_colsum=bout, c_colsum_accumulate=bacc)
        g_b1 = None if bacc else bout
    else:
        dh = _dgrad(dy2, p_w2, w2, aux=h, aux_is_deriv=_saves_gelu_deriv(h.dtype))
        g_b1 = _bgrad(p_b1, dh) if need_b1 else None
    g_w1 = _wgrad(p_w1, dh, x2, wq) if needs[0] else None
    dx = _dgrad(dh, p_w1, w1)
    if tp:
        tp.all_reduce(dx)                                              # C4: entry gradient all-reduce
    return dx, (g_w1, g_b1, g_w2, g_b2)


# ---------------------------------------------------------------------------------------------- Functions
class LinearFn(torch.autograd.Function):
    """y = x·Wᵀ + b over the last dim (nn.Linear)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cdtype):
        xin = _as(x, cdtype)
        x2 = xin.reshape(-1, xin.shape[-1])
        w, b = compute_param(weight, cdtype), compute_param(bias, cdtype)
        y = ops.linear_fwd(x2, w, b)
        ctx.save_for_backward(x2, weight, bias)
        ctx.cdtype, ctx.in_dtype, ctx.in_shape = cdtype, x.dtype, x.shape
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias = ctx.saved_tensors
        dy2 = _as(dy, ctx.cdtype).reshape(-1, dy.shape[-1])
        w = compute_param(weight, ctx.cdtype)
        gw = _wgrad(weight, dy2, x2) if ctx.needs_input_grad[1] else None
        gb = _bgrad(bias, dy2) if (bias is not None and ctx.needs_input_grad[2]) else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _ret_grad(_dgrad(dy2, weight, w).view(ctx.in_shape), ctx.in_dtype)
        return dx, gw, gb, None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, cdtype):
        xin = _as(x, cdtype)
        x2 = xin.reshape(-1, xin.shape[-1])
        g, b = compute_param(weight, cdtype), compute_param(bias, cdtype)
        y, mean, rstd = ops.layernorm_fwd(x2, g, b, eps)
        ctx.save_for_backward(x2, mean, rstd, weight, bias)
        ctx.cdtype, ctx.in_dtype, ctx.in_shape = cdtype, x.dtype, x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, weight, bias = ctx.saved_tensors
        dy2 = _as(dy, ctx.cdtype).reshape(x2.shape)
        cs = torch.empty(x2.shape[1], dtype=torch.float32, device=dy2.device) if (_BIAS_FROM_EPILOGUE and ctx.in_dtype == ctx.cdtype) else None
        dx, dg, db = _ln_bwd(dy2, x2, compute_param(weight, ctx.cdtype), mean, rstd, weight, bias, dx_colsum=cs)
        if cs is not None:
            _publish_stream_colsum(_queue_for(weight), dx, cs)      # e.g. the model's final norm: its dx is the last Block's output gradient
        return _ret_grad(dx.view(ctx.in_shape), ctx.in_dtype), dg, db, None, None


class AttentionFn(torch.autograd.Function):
    """Attention.forward (building_blocks.py:157-192): qkv Linear -> fused SDPA -> proj Linear."""

    @staticmethod
    def forward(ctx, x, qkvw, qkvb, projw, projb, num_heads, cdtype, tp=None):
        xin = _as(x, cdtype)
        B, N, D = xin.shape
        x2 = xin.view(B * N, D)
        c = lambda p: compute_param(p, cdtype)
        y, saved = _attn_fwd(x2, B, N, num_heads, c(qkvw), c(qkvb), c(projw), c(projb), None, tp)
        ctx.save_for_backward(x2, *saved, qkvw, qkvb, projw, projb)
        ctx.meta = (B, N, num_heads, cdtype, x.dtype, tp)
        return y.view(B, N, D)

    @staticmethod
    def backward(ctx, dy):
        x2, qkv, o, lse, qkvw, qkvb, projw, projb = ctx.saved_tensors
        B, N, H, cdtype, in_dtype, tp = ctx.meta
        dy2 = _as(dy, cdtype).reshape(x2.shape)
        c = lambda p: compute_param(p, cdtype)
        dx, g = _attn_bwd(dy2, x2, (qkv, o, lse), B, N, H, c(qkvw), c(projw), qkvw, qkvb, projw, projb, ctx.needs_input_grad[1:5], tp)
        return (_ret_grad(dx.view(B, N, -1), in_dtype),) + g + (None, None, None)


class MlpFn(torch.autograd.Function):
    """Mlp.forward (building_blocks.py:122-129): fc1 -> erf-GELU -> fc2 (drops are p=0, norm=Identity)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, cdtype, tp=None):
        xin = _as(x, cdtype)
        x2 = xin.reshape(-1, xin.shape[-1])
        c = lambda p: compute_param(p, cdtype)
        y, saved = _mlp_fwd(x2, c(w1), c(b1), c(w2), c(b2), None, tp)
        ctx.save_for_backward(x2, *saved, w1, b1, w2, b2)
        ctx.meta = (cdtype, x.dtype, x.shape, tp)
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        x2, h, a, w1, b1, w2, b2 = ctx.saved_tensors
        cdtype, in_dtype, in_shape, tp = ctx.meta
        dy2 = _as(dy, cdtype).reshape(-1, dy.shape[-1])
        c = lambda p: compute_param(p, cdtype)
        dx, g = _mlp_bwd(dy2, x2, (h, a), c(w1), c(w2), w1, b1, w2, b2, ctx.needs_input_grad[1:5], tp)
        return (_ret_grad(dx.view(in_shape), in_dtype),) + g + (None, None)


class BlockFn(torch.autograd.Function):
    """Block.forward (building_blocks.py:236-239) with LayerScale/DropPath = Identity:
    x1 = x + proj(attn(norm1(x))) ; y = x1 + fc2(gelu(fc1(norm2(x1)))).  7 forward launches, residual adds and the
    activation fused into GEMM epilogues; the residual-branch gradients are fused into the LayerNorm backward."""

    @staticmethod
    def _run_forward(x2, B, N, num_heads, eps, c, params, tp):
        n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, f1w, f1b, f2w, f2b = params
        ln1, mean1, rstd1 = ops.layernorm_fwd(x2, c(n1w), c(n1b), eps)
        x1, sa = _attn_fwd(ln1, B, N, num_heads, c(qkvw), c(qkvb), c(projw), c(projb), x2, tp)
        ln2, mean2, rstd2 = ops.layernorm_fwd(x1, c(n2w), c(n2b), eps)
        y, sm = _mlp_fwd(ln2, c(f1w), c(f1b), c(f2w), c(f2b), x1, tp)
        return y, (mean1, rstd1, ln1, *sa, x1, mean2, rstd2, ln2, *sm)

    @staticmethod
    def forward(ctx, x, n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, f1w, f1b, f2w, f2b, num_heads, eps, cdtype, tp=None, recompute=False):
        """recompute = activation checkpointing of the Block (reference: apply_activation_checkpointing(Block),
        training_scripts/train_masked_fsdp.py:393-396): only the Block's INPUT is kept; backward re-runs the seven forward launches to
        rebuild ln1, qkv, o, lse, x1, ln2, h, a (15 of the 16 token matrices a Block otherwise keeps), then proceeds as usual.  The
        rebuilt tensors are bit-identical (no atomics, fixed reduction orders), so gradients equal the non-checkpointed ones exactly."""
        xin = _as(x, cdtype)
        B, N, D = xin.shape
        x2 = xin.view(B * N, D)
        c = lambda p: compute_param(p, cdtype)
        params = (n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, f1w, f1b, f2w, f2b)
        y, saved = BlockFn._run_forward(x2, B, N, num_heads, eps, c, params, tp)
        if recompute:
            ctx.save_for_backward(x2, *params)
        else:
            ctx.save_for_backward(x2, *saved, *params)
        ctx.meta = (B, N, num_heads, cdtype, x.dtype, tp)
        ctx.recompute, ctx.eps = recompute, eps
        return y.view(B, N, D)

    @staticmethod
    def backward(ctx, dy):
        B, N, H, cdtype, in_dtype, tp = ctx.meta
        c = lambda p: compute_param(p, cdtype)
        if ctx.recompute:
            x2, *params = ctx.saved_tensors
            _, saved = BlockFn._run_forward(x2, B, N, H, ctx.eps, c, tuple(params), tp)
            (mean1, rstd1, ln1, qkv, o, lse, x1, mean2, rstd2, ln2, h, a) = saved
            n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, f1w, f1b, f2w, f2b = params
        else:
            (x2, mean1, rstd1, ln1, qkv, o, lse, x1, mean2, rstd2, ln2, h, a,
             n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, f1w, f1b, f2w, f2b) = ctx.saved_tensors
        need = ctx.needs_input_grad
        dy2 = _as(dy, cdtype).reshape(x2.shape)
        wq = _queue_for(qkvw)
        dln2, gm = _mlp_bwd(dy2, ln2, (h, a), c(f1w), c(f2w), f1w, f1b, f2w, f2b, need[9:13], tp, wq, dy2_colsum=_take_stream_colsum(wq, dy2))
        # LayerNorm backward also sums its output (the residual-stream gradient) over the tokens: proj's bias gradient here ...
        projb_done = None
        pb_out, pb_acc = None, False
        if _BIAS_FROM_EPILOGUE and projb is not None and need[6]:
            pb_out, pb_acc = grad_target(projb)
            if pb_out is None:
                pb_out = torch.empty(projb.shape, dtype=torch.float32, device=dy2.device)
            projb_done = (None if pb_acc else pb_out,)
        dx1, g_n2w, g_n2b = _ln_bwd(dln2, x1, c(n2w), mean2, rstd2, n2w, n2b, dres=dy2, dx_colsum=pb_out,       # + residual branch
                                    dx_colsum_accumulate=pb_acc)
        dln1, ga = _attn_bwd(dx1, ln1, (qkv, o, lse), B, N, H, c(qkvw), c(projw), qkvw, qkvb, projw, projb, need[3:7], tp, wq,
                             projb_done=projb_done)
        # ... and, published for whoever receives dx, the fc2 bias gradient of the Block before this one
        cs0 = torch.empty(x2.shape[1], dtype=torch.float32, device=dy2.device) if _BIAS_FROM_EPILOGUE else None
        dx, g_n1w, g_n1b = _ln_bwd(dln1, x2, c(n1w), mean1, rstd1, n1w, n1b, dres=dx1, dx_colsum=cs0)
        if cs0 is not None and in_dtype == cdtype:
            _publish_stream_colsum(wq, dx, cs0)
        wq.end_block()                                   # the Block's 4 weight gradients: grouped launch now, or with the next Blocks'
        return (_ret_grad(dx.view(B, N, -1), in_dtype), g_n1w, g_n1b) + ga + (g_n2w, g_n2b) + gm + (None, None, None, None, None)


class PatchEmbedFn(torch.autograd.Function):
    """PatchEmbed.forward (building_blocks.py:78-92): conv(k=s=p) == im2col + GEMM + bias -> [B, L, D]."""

    @staticmethod
    def forward(ctx, img, weight, bias, patch, cdtype):
        if img.dtype != torch.float32:
            img = img.float()
        img = img if img.is_contiguous() else img.contiguous()
        cols = ops.im2col(img, patch, cdtype)
        w = compute_param(weight, cdtype)
        y = ops.linear_fwd(cols, w.view(w.shape[0], -1), compute_param(bias, cdtype))
        ctx.save_for_backward(cols, weight, bias)
        ctx.cdtype = cdtype
        B = img.shape[0]
        L = 1
        for d_ in img.shape[2:]:
            L *= d_ // patch
        return y.view(B, L, w.shape[0])          # (explicit L: an empty batch has no elements to infer it from)

    @staticmethod
    def backward(ctx, dy):
        cols, weight, bias = ctx.saved_tensors
        dy2 = _as(dy, ctx.cdtype).reshape(cols.shape[0], -1)
        gw = _wgrad(weight, dy2, cols) if ctx.needs_input_grad[1] else None
        gb = _bgrad(bias, dy2) if (bias is not None and ctx.needs_input_grad[2]) else None
        return None, gw, gb, None, None   # no gradient w.r.t. the image (leaf input of the training step)


class TokensFn(torch.autograd.Function):
    """VIT._pos_embed (arch.py:367-393): cat(cls, x) + pos_embed."""

    @staticmethod
    def forward(ctx, x, cls, pos, cdtype):
        xin = _as(x, cdtype)
        B, Lp, D = xin.shape
        out = ops.tokens_fwd(xin.view(B * Lp, D), compute_param(cls, cdtype), compute_param(pos, cdtype), B, Lp, D)
        ctx.save_for_backward(cls, pos)
        ctx.meta = (B, Lp, D, cdtype, x.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        cls, pos = ctx.saved_tensors
        B, Lp, D, cdtype, in_dtype = ctx.meta
        d = _as(dout, cdtype)
        want_pos = pos is not None and ctx.needs_input_grad[2]
        has_cls = cls is not None
        op = oc = None
        ap = ac = False
        if want_pos:
            op, ap = grad_target(pos)
        if has_cls:
            oc, ac = grad_target(cls)
        if (want_pos and has_cls and ap != ac) or (want_pos and has_cls and ((op is None) != (oc is None))):
            op = oc = None
            ap = ac = False
        acc = ap or ac
        dpatches, dpos, dcls = ops.tokens_bwd(d, B, Lp, D, has_cls, want_pos,
                                              dpos=op.view(-1, D) if op is not None else None,
                                              dcls=oc.view(-1) if oc is not None else None, accumulate=acc,
                                              want_patches=ctx.needs_input_grad[0])
        g_pos = None if (not want_pos or acc) else (op if op is not None else dpos.view(pos.shape))
        g_cls = None if (not has_cls or acc) else (oc if oc is not None else dcls.view(cls.shape))
        gx = _ret_grad(dpatches.view(B, Lp, D), in_dtype) if dpatches is not None else None
        return gx, g_cls, g_pos, None


class SeqPatchesFn(torch.autograd.Function):
    """einops 'b c s p -> b s (p c)' on the adaptively patched input (arch.py:466); the input is data, no gradient."""

    @staticmethod
    def forward(ctx, x, cdtype):
        xin = x if x.dtype == torch.float32 else x.float()
        xin = xin if xin.is_contiguous() else xin.contiguous()
        B, C, S, P = xin.shape
        return ops.seq_patches(xin, cdtype).view(B, S, P * C)

    @staticmethod
    def backward(ctx, dy):
        return None, None


class AdaptivePosFn(torch.autograd.Function):
    """VIT._pos_embed with use_adaptive_pos_emb (arch.py:366-393): cat(cls, x) + cat(0, GELU(Linear(seq_ps))).  The K = 3 | 4 Linear,
    the GELU, the concatenation and the add are one kernel; backward recomputes the pre-activation from seq_ps."""

    @staticmethod
    def forward(ctx, x, seq_ps, w, bias, cls, cdtype):
        xin = _as(x, cdtype)
        B, S, D = xin.shape
        sp = seq_ps if seq_ps.dtype == torch.float32 else seq_ps.float()
        sp = sp if sp.is_contiguous() else sp.contiguous()
        out = ops.adaptive_pos_fwd(xin.view(B * S, D), sp, compute_param(w, cdtype), compute_param(bias, cdtype),
                                   compute_param(cls, cdtype).reshape(-1) if cls is not None else None, B, S, D)
        ctx.save_for_backward(sp, w, bias, cls)
        ctx.meta = (B, S, D, cdtype, x.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        sp, w, bias, cls = ctx.saved_tensors
        B, S, D, cdtype, in_dtype = ctx.meta
        d = _as(dout, cdtype)
        has_cls = cls is not None
        tw, aw = grad_target(w)
        tb, ab = grad_target(bias)
        tc, ac = grad_target(cls) if has_cls else (None, False)
        bits = (1 if aw else 0) | (2 if ab else 0) | (4 if ac else 0)
        dx, dw, db, dc = ops.adaptive_pos_bwd(d, sp, compute_param(w, cdtype), compute_param(bias, cdtype), B, S, D, has_cls,
                                              want_dx=ctx.needs_input_grad[0], dw=tw, dbias=tb,
                                              dcls=tc.view(-1) if tc is not None else None, acc_bits=bits)
        gx = _ret_grad(dx.view(B, S, D), in_dtype) if dx is not None else None
        g_w = None if aw else (tw if tw is not None else dw)
        g_b = None if ab else (tb if tb is not None else db)
        g_c = None
        if has_cls and not ac:
            g_c = tc if tc is not None else dc.view(cls.shape)
        return gx, None, g_w, g_b, g_c, None


class VarAggFn(torch.autograd.Function):
    """The attention of VariableMapping_Attention (building_blocks.py:336-366) for one aggregated variable: kv rows ordered (v, r),
    q = the projected learnt query [D] (fp32) -> [R, D]."""

    @staticmethod
    def forward(ctx, kv, q, V, head_dim, scale):
        kv = kv if kv.is_contiguous() else kv.contiguous()
        q32 = (q if q.dtype == torch.float32 else q.float()).reshape(-1).contiguous()
        D = q32.numel()
        R = kv.shape[0] // V
        out, lse = ops.varagg_fwd(kv, q32, V, R, D, head_dim, scale)
        ctx.save_for_backward(kv, q32, out, lse)
        ctx.meta = (V, R, D, head_dim, scale, q.dtype, q.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        kv, q32, out, lse = ctx.saved_tensors
        V, R, D, head_dim, scale, q_dtype, q_shape = ctx.meta
        d = _as(dout, kv.dtype)
        dkv, dq = ops.varagg_bwd(kv, q32, out, lse, d, V, R, D, head_dim, scale)
        return dkv, dq.to(q_dtype).view(q_shape), None, None, None


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss()(logits, labels), mean reduction (train_class_simple.py:24-30); fp32 loss scalar."""

    @staticmethod
    def forward(ctx, logits, labels):
        l2 = logits if logits.is_contiguous() else logits.contiguous()
        loss, dl, _ = ops.cross_entropy(l2, labels, 1.0, want_grad=True)
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g.to(dl.dtype), None


class RandomMaskFn(torch.autograd.Function):
    """MAE.random_masking (arch.py:663-681) for a given noise tensor: returns (kept tokens, mask, ids_restore)."""

    @staticmethod
    def forward(ctx, x, noise, len_keep):
        B, Lp, D = x.shape
        xin = x if x.is_contiguous() else x.contiguous()
        ids_shuffle, ids_restore, mask = ops.mae_mask(noise.contiguous().float(), len_keep)
        kept = ops.gather_rows(xin, ids_shuffle, len_keep, Lp)
        ctx.save_for_backward(ids_shuffle)
        ctx.meta = (Lp, len_keep)
        ctx.mark_non_differentiable(mask, ids_restore)
        return kept, mask, ids_restore

    @staticmethod
    def backward(ctx, dkept, _dm, _di):
        (ids_shuffle,) = ctx.saved_tensors
        Lp, len_keep = ctx.meta
        d = dkept if dkept.is_contiguous() else dkept.contiguous()
        return ops.scatter_rows(d, ids_shuffle, Lp, Lp), None, None


class UnshuffleFn(torch.autograd.Function):
    """MAE.mask_head's un-shuffle (arch.py:687-697): gather(cat(x, mask_tokens), ids_restore) + decoder_pos_embed."""

    @staticmethod
    def forward(ctx, x, mask_token, ids_restore, pos, cdtype):
        xin = _as(x, cdtype)
        out = ops.unshuffle_fwd(xin, compute_param(mask_token, cdtype).reshape(-1), ids_restore, compute_param(pos, cdtype))
        ctx.save_for_backward(ids_restore, mask_token, pos)
        ctx.meta = (xin.shape[1], cdtype, x.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        ids_restore, mask_token, pos = ctx.saved_tensors
        R, cdtype, in_dtype = ctx.meta
        d = _as(dout, cdtype)
        want_pos = pos is not None and ctx.needs_input_grad[3]
        om, am = grad_target(mask_token)
        op, ap = grad_target(pos) if want_pos else (None, False)
        if want_pos and (am != ap or ((om is None) != (op is None))):
            om = op = None
            am = ap = False
        dx, dmask, dpos = ops.unshuffle_bwd(d, ids_restore, R, want_pos, dmask=om.view(-1) if om is not None else None,
                                            dpos=op.view(-1, d.shape[-1]) if op is not None else None, accumulate=am)
        g_m = None if am else (om if om is not None else dmask.view(mask_token.shape))
        g_p = None if (not want_pos or ap) else (op if op is not None else dpos.view(pos.shape))
        return _ret_grad(dx, in_dtype), g_m, None, g_p, None


class PatchMSEFn(torch.autograd.Function):
    """MSE between pred and patchify(img) (misc.py:14-33 + train_masked_simple.py:43-47), optional mask (metrics.py:11-17)."""

    @staticmethod
    def forward(ctx, pred, img, mask, patch):
        p = pred if pred.is_contiguous() else pred.contiguous()
        im = img if img.is_contiguous() else img.contiguous()
        loss, dpred = ops.patch_mse(p, im.float(), patch, mask=mask, grad_scale=1.0, want_grad=True)
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g.to(dpred.dtype), None, None, None


class InstNormActFn(torch.autograd.Function):
    """leaky_relu(instance_norm(x) [+ res], slope) — the normalisation / residual / activation chain of monai's UnetResBlock (reference
    simple/arch.py:808-940) as two HBM passes forward (statistics, apply) and two backward."""

    @staticmethod
    def forward(ctx, x, res, eps, slope):
        xin = x if x.is_contiguous() else x.contiguous()
        rin = None if res is None else (res if res.is_contiguous() else res.contiguous())
        y, mean, rstd = ops.instnorm_fwd(xin, rin, eps, slope)
        ctx.save_for_backward(xin, y, mean, rstd)
        ctx.slope, ctx.has_res = slope, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd = ctx.saved_tensors
        d = dy if dy.is_contiguous() else dy.contiguous()
        dx, dres = ops.instnorm_bwd(d.to(x.dtype), y, x, mean, rstd, ctx.slope, ctx.has_res and ctx.needs_input_grad[1])
        return dx, dres, None, None


class DiceCEFn(torch.autograd.Function):
    """monai DiceCELoss(to_onehot_y=True, softmax=True, squared_pred=True) (reference train_unetr_simple.py:38), forward + backward fused"""

    @staticmethod
    def forward(ctx, logits, labels, smooth_nr, smooth_dr):
        lb = labels if labels.is_contiguous() else labels.contiguous()
        if logits.is_contiguous() or _is_channels_last_rows(logits):       # the HIP decoder's logits: read in place, no N C D H W copy
            lg = logits
        else:
            lg = logits.contiguous()
        loss, dl = ops.dice_ce(lg, lb, smooth_nr, smooth_dr, 1.0, want_grad=True)
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        if dl.is_contiguous():
            return dl * g.to(dl.dtype), None, None, None
        flat = dl.as_strided((dl.shape[0] * dl.stride(0),), (1,))           # the padded channels-last buffer behind the view
        return (flat * g.to(dl.dtype)).as_strided(dl.shape, dl.stride()), None, None, None


def _is_channels_last_rows(t):
    """[B, n, *spatial] view over a dense [B, *spatial, ld >= n] buffer"""
    if t.dim() < 3 or t.stride(1) != 1 or t.storage_offset() != 0:
        return False
    ld = t.stride(-1)
    exp = ld
    for d in range(t.dim() - 1, 1, -1):
        if t.stride(d) != exp:
            return False
        exp *= t.shape[d]
    return ld >= t.shape[1] and t.stride(0) == exp


# ---------------------------------------------------------------------------------------------- public helpers
def instnorm_act(x, res=None, eps=1e-5, slope=0.01):
    return InstNormActFn.apply(x, res, eps, slope)


def dice_ce(logits, labels, smooth_nr=1e-5, smooth_dr=1e-5):
    return DiceCEFn.apply(logits, labels, smooth_nr, smooth_dr)


def cross_entropy(logits, labels):
    return CrossEntropyFn.apply(logits, labels)


def patch_mse(pred, img, patch_size, mask=None):
    return PatchMSEFn.apply(pred, img, mask, patch_size)

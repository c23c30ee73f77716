"""Data-parallel gradient reduction for the flat-buffer models (replaces the torch DDP wrap of the reference's
training_scripts/train_class_simple.py:230, train_masked_simple.py:263, train_unetr_simple.py:273).

One process per GPU; gradients already live in ONE flat fp32 buffer (HipParamStore.flat_g), so a bucket is just a
slice of it: no flatten / unflatten copies.  Each bucket is all-reduced (mean) with torch.distributed — backend "nccl"
is RCCL on ROCm, which runs the collective on its own HIP stream — as soon as the last gradient of the bucket has been
written by the backward kernels, so communication overlaps the remaining backward GEMMs.  An autograd end-of-backward
callback makes the compute stream wait for the outstanding collectives (no host synchronisation), so the reference loop
`loss.backward(); optimizer.step(); optimizer.zero_grad()` works unchanged.

MI355X note (8 GPUs fully connected by xGMI, ~153 GB/s per link): bucket size is a trade between per-collective launch
latency and overlap; 32 MiB fp32 buckets keep ~40 collectives per ViT-L step, each far above the latency-bound regime.

Gradient transport dtype: fp32 in fp32 mode (torch DDP of the train_*_simple.py scripts); bf16 when the model computes in bf16,
mirroring the reference's bf16 policy `MixedPrecision(param_dtype=reduce_dtype=buffer_dtype=bfloat16)`
(training_scripts/train_masked_fsdp.py:375-381): the bucket is cast into a persistent bf16 staging buffer, all-reduced (mean)
there and cast back into the fp32 gradient buffer at the end of backward.  Half the xGMI bytes (ViT-L: 0.61 instead of 1.22 GB
per step and rank) and half the time RCCL's workgroups hold CUs that the persistent GEMM grids want.  `reduce_dtype=` or
UCFVIT_DDP_REDUCE_DTYPE=fp32|bf16 override the default.

Requirements: call optimizer.zero_grad() (set_to_none or not) every step, as the reference loop does.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import functional as _HF
from . import ops as _ops
from .params import ensure_store


class _FlatGrads:
    """flat gradient buffer for a plain (non-HIP) module: used by the CPU/gloo tests of the reducer logic"""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()
        self.total = off
        dev = self.params[0].device
        self.flat_g = torch.zeros(off, dtype=self.params[0].dtype, device=dev)
        self.decay_end = off

    def attach(self):
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat_g[o:o + p.numel()].view(p.shape)   # autograd then accumulates in place


class HipDataParallel(nn.Module):
    def __init__(self, module, process_group=None, bucket_mb=32, broadcast_from=0, reduce_dtype=None, algorithm=None):
        super().__init__()
        self.module = module
        # gradient transport of a bucket: "all_reduce" = one RCCL all-reduce (RCCL picks ring / tree / its own direct forms), "direct" =
        # reduce-scatter + all-gather written out for a fully connected xGMI node: an all-to-all sends chunk j of the
        # bucket to rank j over all 7 links at once (S/8 per link instead of a ring's 2 (P-1)/P S over one), every rank sums the P chunks
        # it received in a fixed order (fp32) and an all-gather hands the means round the same way.  Same result as the all-reduce up to
        # the summation order (fixed here).  UNMEASURED on hardware (no multi-GPU box in the build): opt-in, UCFVIT_DDP_ALGO=direct.
        algo = algorithm or os.environ.get("UCFVIT_DDP_ALGO", "all_reduce")
        if algo not in ("all_reduce", "direct"):
            raise ValueError("HipDataParallel: algorithm must be 'all_reduce' or 'direct'")
        self.algorithm = algo
        self._cstream = None
        rd = reduce_dtype if reduce_dtype is not None else os.environ.get("UCFVIT_DDP_REDUCE_DTYPE")
        if isinstance(rd, str):
            rd = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}[rd.lower()]
        if rd not in (None, torch.float32, torch.bfloat16):
            raise ValueError("HipDataParallel: reduce_dtype must be torch.float32 or torch.bfloat16")
        self._reduce_dtype = rd          # None: follow the module's compute dtype (looked up when the first bucket is launched)
        self._comm = None
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        first = next(module.parameters())
        self._hip = first.is_cuda
        if self._hip:
            st = ensure_store(module)
            self.store, self.flat_g = st, st.flat_g
            params, offsets = st.params, st.offsets
            boundaries = [st.decay_end]      # the no-decay tail (pos_embed ...) gets its gradient last: own bucket
            flat_p = st.flat_p
        else:
            fg = _FlatGrads(module)
            self.store, self.flat_g = fg, fg.flat_g
            params, offsets = fg.params, fg.offsets
            boundaries = []
            flat_p = None
        # --- parameter broadcast at wrap time (DDP semantics)
        with torch.no_grad():
            if flat_p is not None:
                dist.broadcast(flat_p, src=dist.get_global_rank(process_group, broadcast_from) if process_group else broadcast_from,
                               group=process_group)
                if getattr(st, "flat_s", None) is not None:
                    st.refresh_shadow(force=True)
            else:
                for p in params:
                    dist.broadcast(p.data, src=broadcast_from, group=process_group)
        # --- buckets: contiguous [lo, hi) slices of the flat buffer, walking from the end (backward order)
        limit = int(bucket_mb * (1 << 20) // self.flat_g.element_size())
        self.buckets = []           # (lo, hi, n_params)
        self.param_bucket = {}
        cur_hi, cur_n, cur_lo = None, 0, None
        ends = [o + ((p.numel() + 63) // 64 * 64 if self._hip else p.numel()) for p, o in zip(params, offsets)]
        for idx in range(len(params) - 1, -1, -1):
            lo, hi = offsets[idx], ends[idx]
            if cur_hi is None:
                cur_hi, cur_n = hi, 0
            cur_lo = lo
            cur_n += 1
            self.param_bucket[params[idx]] = len(self.buckets)
            if (cur_hi - cur_lo) >= limit or lo in boundaries or idx == 0:
                self.buckets.append((cur_lo, cur_hi, cur_n))
                cur_hi = None
        self._pending = [0] * len(self.buckets)
        self._works = []
        self._ready = []
        self._callback_queued = False
        if self._hip:
            _HF.add_wgrad_flush_listener(self._after_wgrad_flush, self.store)
            if self.world > 1:
                _ops.set_dynamic_tile_schedule(True)     # RCCL's workgroups will hold CUs while the persistent GEMM grids run
        for p in params:
            if p.requires_grad:
                p.register_post_accumulate_grad_hook(self._on_grad)
        if not self._hip:
            fg.attach()

    # ------------------------------------------------------------------ hooks
    def _on_grad(self, p):
        if not self._callback_queued:
            self._callback_queued = True
            self._pending = [0] * len(self.buckets)
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
        if self._hip:
            # a gradient torch autograd produced itself (a parameter used by a torch op: variable embedding, torch/MIOpen decoder)
            # is not in the flat buffer the buckets are slices of: move it there before its bucket can be reduced
            slot = getattr(p, "_ucf_slot", None)
            if slot is not None and p.grad is not None and slot[0].owns(p, slot[1]):
                view = slot[0].grad_view(p, slot[1], slot[2])
                if p.grad.data_ptr() != view.data_ptr() and p.grad.dtype == view.dtype:
                    view.copy_(p.grad)
                    p.grad = view
        b = self.param_bucket[p]
        self._pending[b] += 1
        if self._pending[b] == self.buckets[b][2]:
            if self._hip and _HF.wgrads_pending(self.store):
                # weight gradients are queued for grouped launches that span several Blocks (functional.WgradQueue): this slice is
                # complete for autograd but not yet on the stream; reduce it when the queue has flushed
                self._ready.append(b)
            else:
                self._launch(b)

    def _after_wgrad_flush(self):
        ready, self._ready = self._ready, []
        for b in ready:
            if self._pending[b] >= 0:
                self._launch(b)

    def reduce_dtype(self):
        if not self._hip:
            return torch.float32        # plain-module (CPU test) path
        if self._reduce_dtype is not None:
            return self._reduce_dtype
        return torch.bfloat16 if getattr(self.module, "compute_dtype", torch.float32) == torch.bfloat16 else torch.float32

    def _staging(self, lo, hi):
        if self._comm is None:
            self._comm = torch.empty(self.flat_g.numel(), dtype=torch.bfloat16, device=self.flat_g.device)
        return self._comm[lo:hi]

    def _launch_direct(self, b):
        """reduce-scatter by all-to-all + fixed-order local sum, then all-gather (see __init__); runs on a side stream on the GPU"""
        lo, hi, _ = self.buckets[b]
        view = self.flat_g[lo:hi]
        P, n = self.world, hi - lo
        m = -(-n // P)
        rdt = self.reduce_dtype()
        host = (not self._hip) or dist.get_backend(self.pg) == "gloo"
        dev = torch.device("cpu") if host else view.device

        def run():
            send = torch.zeros(P * m, dtype=rdt, device=dev)
            send[:n].copy_(view)                                             # (cast to the transport dtype; the tail pad stays zero)
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=self.pg)                # chunk j of every rank's bucket lands on rank j
            mine = recv.view(P, m).sum(dim=0, dtype=torch.float32).mul_(1.0 / P).to(rdt)     # fixed order: rank 0's chunk first
            out = torch.empty(P * m, dtype=rdt, device=dev)
            dist.all_gather_into_tensor(out, mine, group=self.pg)
            view.copy_(out[:n])                                              # the mean, back in the fp32 gradient buffer

        if host:
            run()
        else:
            if self._cstream is None:
                self._cstream = torch.cuda.Stream(device=view.device)
            self._cstream.wait_stream(torch.cuda.current_stream())           # the bucket's gradients are complete on the compute stream
            with torch.cuda.stream(self._cstream):
                run()                                                        # RCCL work is stream-ordered behind / ahead of the glue kernels
        self._pending[b] = -(1 << 30)

    def _launch(self, b):
        if self.algorithm == "direct" and self.world > 1:
            return self._launch_direct(b)
        lo, hi, _ = self.buckets[b]
        view = self.flat_g[lo:hi]
        bf16 = self.reduce_dtype() == torch.bfloat16
        if self._hip and dist.get_backend(self.pg) != "gloo":
            if bf16:
                c = _ops.cast(view, self._staging(lo, hi))                                      # compute stream; RCCL waits for it
                w = dist.all_reduce(c, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
                self._works.append((w, lo, hi))
            else:
                w = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)   # RCCL, its own HIP stream
                self._works.append((w, None, None))
        elif self._hip:
            # test transport (several ranks sharing one GPU cannot use RCCL): host-staged, synchronous
            h = view.cpu()
            if bf16:
                h = h.bfloat16().float()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
            h.div_(self.world)
            view.copy_(h.bfloat16().float() if bf16 else h)
        else:
            view.div_(self.world)
            w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._works.append((w, None, None))
        self._pending[b] = -(1 << 30)

    def _finish(self):
        if self._hip:
            _HF.flush_wgrads(self.store)       # (also launches the buckets that were waiting for it)
        # parameters that received no gradient this step (unused) leave their bucket incomplete: reduce it anyway so
        # every rank issues the same collectives in the same order
        for b in range(len(self.buckets)):
            if self._pending[b] >= 0:
                self._launch(b)
        for w, lo, hi in self._works:
            w.wait()          # compute stream waits for the RCCL stream; no host sync on the nccl backend
            if lo is not None:
                _ops.cast(self._comm[lo:hi], self.flat_g[lo:hi])      # bf16 mean back into the fp32 gradient buffer
        self._works = []
        if self._cstream is not None:
            torch.cuda.current_stream().wait_stream(self._cstream)      # "direct" transport: the side stream's last copy-back
        self._callback_queued = False

    # ------------------------------------------------------------------ module API
    def forward(self, *args, **kwargs):
        if not self._hip:
            # plain-module path keeps gradients inside the flat buffer even after zero_grad(set_to_none=True)
            for p, o in zip(self.store.params, self.store.offsets):
                if p.grad is None:
                    self.flat_g[o:o + p.numel()].zero_()
                    p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
        return self.module(*args, **kwargs)

    def state_dict(self, *args, **kwargs):
        return super().state_dict(*args, **kwargs)   # keys carry the 'module.' prefix like torch DDP checkpoints

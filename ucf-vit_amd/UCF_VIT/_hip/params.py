"""Flat parameter store for a model running on the HIP kernels.

MI355X-first memory layout: every parameter of a model lives in ONE flat fp32 buffer (master weights), with a
same-shaped flat fp32 gradient buffer and (bf16 mode) a flat bf16 "shadow" used as the GEMM operand.  This gives
 - one fused AdamW launch per hyper-parameter group (the launch also refreshes the bf16 shadow in the same pass),
 - contiguous gradient buckets for the RCCL all-reduce (no flatten/unflatten copies),
 - one cast launch when something else (load_state_dict, a torch optimizer) changed the master weights.
nn.Parameters stay ordinary Parameters (views into the flat buffer), so state_dict keys and shapes are unchanged
(reference layout: src/UCF_VIT/simple/arch.py:232-271, verified in SURVEY.md §8b).

Parameter order = the two AdamW groups of utils/misc.py:58-84: weight-decayed parameters first, then the
no-decay ones (names containing var_embed / pos_embed / time_pos_embed).
"""
import torch

from . import ops

_ALIGN = 64  # elements; keeps every parameter (fp32 and bf16 view) 16-byte aligned for vector loads


def is_no_decay(name):
    return ("var_embed" in name) or ("pos_embed" in name) or ("time_pos_embed" in name)


class HipParamStore:
    def __init__(self, module):
        named = list(module.named_parameters())
        if not named:
            raise RuntimeError("HipParamStore: module has no parameters")
        dev = named[0][1].device
        for n, p in named:
            if p.device != dev or not p.is_cuda:
                raise RuntimeError(f"HipParamStore: parameter {n} is on {p.device}; all parameters must be on one cuda device")
            if p.dtype != torch.float32:
                raise RuntimeError(f"HipParamStore: parameter {n} is {p.dtype}; master weights are float32")
        named.sort(key=lambda kv: 1 if is_no_decay(kv[0]) else 0)  # stable: decay group first
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = off
        self.n_decay = sum(0 if is_no_decay(n) else 1 for n in self.names)
        self.decay_end = self.offsets[self.n_decay] if self.n_decay < len(self.params) else self.total
        self.device = dev
        self.flat_p = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.flat_s = None  # bf16 shadow, allocated on first bf16 use
        self.flat_t = None  # bf16 transposed shadow of the 2-D (nn.Linear) weights, for the data-gradient GEMMs
        self._shadow_sig = None
        self._t_fresh = False
        # transposable weights: plain 2-D matrices with both extents >= 128 and multiples of 8
        rows = []
        tiles = 0
        for p, o in zip(self.params, self.offsets):
            if p.dim() == 2 and p.shape[0] % 8 == 0 and p.shape[1] % 8 == 0 and min(p.shape) >= 128:
                r, c = p.shape
                rows.append([o, o, r, c, tiles])
                tiles += ((r + 63) // 64) * ((c + 63) // 64)
                p._ucf_has_t = True
        self._t_tiles = tiles
        self._t_table = torch.tensor(rows, dtype=torch.int64, device=dev) if rows else None
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                if p.grad is not None:
                    self.flat_g[o:o + n].copy_(p.grad.reshape(-1))
                    p.grad = self.flat_g[o:o + n].view(p.shape)
                p._ucf_slot = (self, o, n)
        self._base = self.flat_p.data_ptr()

    # -- validity: .to(device) / load of a foreign model replaces p.data; then the store must be rebuilt
    def valid(self):
        b = self._base
        for p, o in zip(self.params, self.offsets):
            if p.data_ptr() != b + 4 * o:
                return False
        return True

    def owns(self, p, o):
        return p.data_ptr() == self._base + 4 * o

    def grad_view(self, p, o, n):
        return self.flat_g[o:o + n].view(p.shape)

    def _sig(self):
        return sum(p._version for p in self.params)

    def refresh_shadow(self, force=False):
        """(Re)cast master fp32 -> bf16 shadow if any parameter was modified by something other than HipAdamW."""
        if self.flat_s is None:
            self.flat_s = torch.empty(self.total, dtype=torch.bfloat16, device=self.device)
            force = True
        sig = self._sig()
        if force or sig != self._shadow_sig:
            ops.cast(self.flat_p, self.flat_s)
            self._shadow_sig = sig
            self._t_fresh = False
        if not self._t_fresh and self._t_table is not None:
            if self.flat_t is None:
                self.flat_t = torch.empty(self.total, dtype=torch.bfloat16, device=self.device)
            ops.transpose_batched(self.flat_s, self.flat_t, self._t_table, self._t_table.shape[0], self._t_tiles)
            self._t_fresh = True

    def shadow_view(self, p, o, n):
        return self.flat_s[o:o + n].view(p.shape)

    def note_shadow_fresh(self):
        """called by HipAdamW after a fused update that also wrote the shadow (the transposed copy follows on the next forward)"""
        self._shadow_sig = self._sig()
        self._t_fresh = False

    def shadow_t_view(self, p, o, n):
        return self.flat_t[o:o + n].view(p.shape[1], p.shape[0])


def ensure_store(module):
    """Return the module's flat store, (re)building it when the parameters are not (or no longer) views of it."""
    st = getattr(module, "_ucf_store", None)
    if st is None or not st.valid():
        st = HipParamStore(module)
        object.__setattr__(module, "_ucf_store", st)
    return st


def compute_param(p, dtype):
    """The tensor the kernels read for parameter `p` in compute dtype `dtype` (fp32 master or bf16 shadow)."""
    if p is None:
        return None
    if dtype == torch.float32:
        t = p.detach()
        if not t.is_cuda:
            raise RuntimeError("UCF_VIT: parameters must be on the MI355X (cuda) device; there is no CPU path")
        return t if t.is_contiguous() else t.contiguous()
    slot = getattr(p, "_ucf_slot", None)
    if slot is not None and slot[0].owns(p, slot[1]) and slot[0].flat_s is not None:
        return slot[0].shadow_view(p, slot[1], slot[2])
    # stand-alone module (no flat store): per-parameter shadow keyed by version
    if not p.is_cuda:
        raise RuntimeError("UCF_VIT: parameters must be on the MI355X (cuda) device; there is no CPU path")
    ent = getattr(p, "_ucf_shadow", None)  # lives and dies with the Parameter object
    if ent is None or ent[0] != p._version or ent[1] != p.data_ptr():
        sh = torch.empty(p.shape, dtype=torch.bfloat16, device=p.device)
        ops.cast(p.detach().contiguous(), sh)
        ent = (p._version, p.data_ptr(), sh)
        p._ucf_shadow = ent
    return ent[2]


def compute_param_t(p, dtype):
    """Transposed bf16 shadow [K, N] of a 2-D weight [N, K] if the flat store keeps one (bf16 mode), else None."""
    if dtype != torch.bfloat16 or not getattr(p, "_ucf_has_t", False):
        return None
    slot = getattr(p, "_ucf_slot", None)
    if slot is None or not slot[0].owns(p, slot[1]) or slot[0].flat_t is None or not slot[0]._t_fresh:
        return None
    return slot[0].shadow_t_view(p, slot[1], slot[2])


def grad_target(p):
    """Where a backward kernel should write d(loss)/dp.

    returns (out, accumulate): out = fp32 tensor to write into (None: allocate a fresh one and hand it to autograd),
    accumulate = True when `out` already holds this step's partial gradient (then nothing is returned to autograd).
    """
    slot = getattr(p, "_ucf_slot", None)
    if slot is None:
        return None, False
    st, o, n = slot
    if not st.owns(p, o):
        return None, False
    view = st.grad_view(p, o, n)
    if p.grad is None:
        return view, False
    if p.grad.data_ptr() == view.data_ptr():
        return view, True
    return None, False

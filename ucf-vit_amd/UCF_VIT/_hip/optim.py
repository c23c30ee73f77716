"""Fused AdamW on the HIP kernel (ucfvit_adamw), a torch.optim.Optimizer so schedulers / state_dict keep working.

Semantics = torch.optim.AdamW (reference: utils/misc.py:58-84, two param groups).  When the parameters of a group
are consecutive slots of one HipParamStore and every one of them received its gradient in the store's flat gradient
buffer, the whole group is ONE launch over the flat segment (which also rewrites the bf16 shadow weights in the
same pass); otherwise it falls back to one launch per parameter.
"""
import torch

from . import ops


class HipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.grad_scale = grad_scale  # multiplies gradients inside the kernel (e.g. 1/world_size, or 1/loss_scale)
        # flat moment buffers of the one-launch path, keyed (id(store), lo, hi).  They live on the optimizer object, NOT in self.state:
        # state_dict() then serialises only the per-parameter {step, exp_avg, exp_avg_sq} entries (views into these buffers) — the
        # torch.optim.AdamW layout the reference's checkpoints hold (train_class_simple.py:364-388) — and nothing keyed by an id()
        self._flat = {}

    def load_state_dict(self, state_dict):
        """accepts a torch.optim.AdamW / HipAdamW state_dict; the flat buffers are rebuilt from the per-parameter entries on the next
        step (the loaded tensors are copied in, so a checkpoint mapped to the CPU ends up in device memory before any kernel sees it)"""
        super().load_state_dict(state_dict)
        self._flat = {}

    @staticmethod
    def _adopt_foreign_grads(group):
        """a gradient that torch autograd produced itself (a parameter used by a torch op: the variable embedding, a torch/MIOpen
        decoder) lives outside the flat buffer: copy it into its slot and re-point p.grad, so the group keeps its one-launch update"""
        for p in group["params"]:
            s = getattr(p, "_ucf_slot", None)
            if s is None or p.grad is None or not s[0].owns(p, s[1]):
                continue
            view = s[0].grad_view(p, s[1], s[2])
            if p.grad.data_ptr() != view.data_ptr() and p.grad.dtype == view.dtype and p.grad.device == view.device:
                view.copy_(p.grad)
                p.grad = view

    @staticmethod
    def _flat_run(group):
        """(store, start, end) if the group's params are one contiguous run of a store with grads in the flat buffer."""
        ps = group["params"]
        slot0 = getattr(ps[0], "_ucf_slot", None)
        if slot0 is None:
            return None
        st = slot0[0]
        expect = slot0[1]
        for p in ps:
            s = getattr(p, "_ucf_slot", None)
            if s is None or s[0] is not st or s[1] != expect or not st.owns(p, s[1]):
                return None
            if p.grad is None or p.grad.data_ptr() != st.flat_g.data_ptr() + 4 * s[1]:
                return None
            expect = s[1] + (s[2] + 63) // 64 * 64
        return st, slot0[1], expect

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            if not group["params"]:
                continue
            b1, b2 = group["betas"]
            lr, eps, wd = group["lr"], group["eps"], group["weight_decay"]
            self._adopt_foreign_grads(group)
            run = self._flat_run(group)
            if run is not None:
                st, lo, hi = run
                fs = self._flat
                key = (id(st), lo, hi)
                ent = fs.get(key)
                if ent is None:
                    ent = fs[key] = dict(step=0, m=torch.zeros(hi - lo, dtype=torch.float32, device=st.device),
                                         v=torch.zeros(hi - lo, dtype=torch.float32, device=st.device))
                    # adopt per-parameter state if the slow path ran before
                    for p in group["params"]:
                        s = self.state.get(p)
                        if s and "exp_avg" in s:
                            o = p._ucf_slot[1] - lo
                            ent["m"][o:o + p.numel()].copy_(s["exp_avg"].reshape(-1).to(ent["m"].device, torch.float32))
                            ent["v"][o:o + p.numel()].copy_(s["exp_avg_sq"].reshape(-1).to(ent["v"].device, torch.float32))
                            ent["step"] = int(s["step"])
                    for p in group["params"]:
                        o = p._ucf_slot[1] - lo
                        self.state[p] = dict(step=torch.tensor(float(ent["step"])),
                                             exp_avg=ent["m"][o:o + p.numel()].view(p.shape),
                                             exp_avg_sq=ent["v"][o:o + p.numel()].view(p.shape))
                ent["step"] += 1
                shadow = st.flat_s[lo:hi] if st.flat_s is not None else None
                ops.adamw(st.flat_p[lo:hi], st.flat_g[lo:hi], ent["m"], ent["v"], shadow, lr, b1, b2, eps, wd, ent["step"],
                          self.grad_scale)
                for p in group["params"]:
                    self.state[p]["step"] += 1
                if shadow is not None:
                    st.note_shadow_fresh()
                continue
            for p in group["params"]:
                if p.grad is None:
                    continue
                s = self.state.setdefault(p, {})
                if "exp_avg" not in s:
                    s["step"] = torch.tensor(0.0)
                    s["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                s["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adamw(p.data, g, s["exp_avg"], s["exp_avg_sq"], None, lr, b1, b2, eps, wd, int(s["step"]), self.grad_scale)
                p._version  # (kernel wrote p in place; bf16 shadows are re-cast lazily via the store signature)
                slot = getattr(p, "_ucf_slot", None)
                if slot is not None:
                    slot[0]._shadow_sig = None
        return loss

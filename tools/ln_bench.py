#!/usr/bin/env python3
"""LayerNorm forward / backward kernel times at the ViT-L token matrix (the grids are fixed in csrc/norm.hip: ln_grid_cap)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
from UCF_VIT._hip import ops
M, D = 166 * 197, 1024
x = torch.randn(M, D, device="cuda").bfloat16(); dy = torch.randn(M, D, device="cuda").bfloat16(); dres = torch.randn(M, D, device="cuda").bfloat16()
g = torch.ones(D, device="cuda").bfloat16(); b = torch.zeros(D, device="cuda").bfloat16()
cs = torch.empty(D, device="cuda")
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / reps * 1e3
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
tf = t(lambda: ops.layernorm_fwd(x, g, b, 1e-6))
tb = t(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dx_colsum=cs))
print(f"LN fwd {tf:.1f} us ({2*M*D*2/tf/1e6:.2f} TB/s) | LN bwd(+dres,+colsum, incl. reduce) {tb:.1f} us ({4*M*D*2/tb/1e6:.2f} TB/s)")

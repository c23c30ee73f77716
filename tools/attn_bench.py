#!/usr/bin/env python3
"""attention micro-benchmark (GPU box): python tools/attn_bench.py [B] [N] [H] [dh]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
from UCF_VIT._hip import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 166
N = int(sys.argv[2]) if len(sys.argv) > 2 else 197
H = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dh = int(sys.argv[4]) if len(sys.argv) > 4 else 64
qkv = torch.randn(B * N, 3 * H * dh, device="cuda").bfloat16()
do = torch.randn(B * N, H * dh, device="cuda").bfloat16()
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / reps
o, lse = ops.attention_fwd(qkv, B, N, H, dh, dh ** -0.5)
q, k, v = qkv.view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4).float().unbind(0)
ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B * N, H * dh)
print("fwd err", ((o.float() - ref).abs().max() / ref.abs().max()).item())
fl = 4.0 * B * H * N * N * dh
tf = t(lambda: ops.attention_fwd(qkv, B, N, H, dh, dh ** -0.5))
tb = t(lambda: ops.attention_bwd(qkv, o, do, lse, B, N, H, dh, dh ** -0.5))
print(f"B={B} N={N} H={H} dh={dh}: fwd {tf*1e3:.1f} us {fl/tf/1e9:.1f} TF | bwd {tb*1e3:.1f} us {2.5*fl/tb/1e9:.1f} TF (2.5x fwd flops)")

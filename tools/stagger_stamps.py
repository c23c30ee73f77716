#!/usr/bin/env python3
"""Diagnostic: barrier-to-barrier shader-clock intervals of workgroup 0 of the staggered GEMM (csrc/gemm_stagger.hip built with -DS5_STAMP
into a scratch library; the product library carries no stamps).

    python tools/stagger_stamps.py [fc1|qkv|proj|fc2d] [B=665]
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ucf-vit_amd")
out_dir = os.path.join(ROOT, "gpurun_out", "stamps")
os.makedirs(out_dir, exist_ok=True)
lib = os.path.join(out_dir, "libucfvit_stamp.so")
objs = [os.path.join(PKG, "build", f) for f in os.listdir(os.path.join(PKG, "build")) if f.endswith(".o") and f != "gemm_stagger.o"]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-DS5_STAMP", *__import__("shlex").split(os.environ.get("S5_EXTRA", "")), "-c",
                os.path.join(PKG, "csrc", "gemm_stagger.hip"), "-o", os.path.join(out_dir, "gemm_stagger_stamp.o")], check=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, os.path.join(out_dir, "gemm_stagger_stamp.o")] + objs, check=True)
os.environ["UCFVIT_HIP_LIB"] = lib
sys.path.insert(0, PKG)
import torch
from UCF_VIT._hip import ops, lib as L
from UCF_VIT._hip.lib import ACT_GELU_SAVE_DERIV
which = sys.argv[1] if len(sys.argv) > 1 else "qkv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 665
M, D = B * 197, 1024
g = torch.Generator().manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).bfloat16().cuda()
x = rnd(M, D)
if which == "qkv":
    w, b = rnd(3 * D, D, sc=0.03), rnd(3 * D)
    fn = lambda: ops.linear_fwd(x, w, b)
elif which == "proj":
    w, b, res = rnd(D, D, sc=0.03), rnd(D), rnd(M, D)
    fn = lambda: ops.linear_fwd(x, w, b, residual=res)
elif which == "fc1":
    w, b, aux = rnd(4 * D, D, sc=0.03), rnd(4 * D), torch.empty(M, 4 * D, dtype=torch.bfloat16, device="cuda")
    fn = lambda: ops.linear_fwd(x, w, b, act=ACT_GELU_SAVE_DERIV, aux_out=aux)
else:
    wT, gp, cs = rnd(4 * D, D, sc=0.03), rnd(M, 4 * D), torch.empty(4 * D, dtype=torch.float32, device="cuda")
    fn = lambda: ops.linear_dgrad_t(x, wT, act_grad_aux=gp, aux_is_deriv=True, c_colsum=cs)
h = L.load()
for _ in range(3):
    fn()
torch.cuda.synchronize()
h.ucfvit_debug_stagger_stamps(None, None, 1)
fn()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
cnt = (ctypes.c_int * 2)()
h.ucfvit_debug_stagger_stamps(buf, cnt, 1)
E = int(os.environ.get("UCFVIT_GEMM_STAGGER", "0")) or (4 if which in ("fc1", "fc2d") else 2)
nk = D // 64
print(f"{which}: stamps G0 {cnt[0]} G1 {cnt[1]}; E = {E}, nk = {nk}, intervals per tile cycle = {4 * (nk + E)}")
for gi in range(2):
    t = [buf[gi * 2048 + i] for i in range(cnt[gi])]
    d = [t[i + 1] - t[i] for i in range(len(t) - 1)]
    per = 4 * (nk + E)
    off = 1 + (1 if gi == 1 else 0)      # prologue barrier (+ G1's skew barrier): stamp index of the first step's first barrier
    # second tile cycle (steady state)
    base = off + per
    seg = d[base:base + per]
    print(f"G{gi} second cycle, intervals (shader clocks) by step:")
    for st in range(nk + E):
        print(f"   step {st:2d}: " + " ".join(f"{v:5d}" for v in seg[4 * st:4 * st + 4]) + f"   = {sum(seg[4 * st:4 * st + 4])}")
    print(f"   cycle total {sum(seg)} clocks")

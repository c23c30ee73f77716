#!/usr/bin/env python3
"""Diagnostic: K-loop / epilogue timing of the 128x128 GEMM kernel via s_memrealtime stamps (100 MHz).
usage: python tools/wgrad_stamps.py wgrad|fwd      (fwd: run with UCFVIT_GEMM_SMALL=1 to force the 128x128 KC x KC kernel)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
NB = 1024
dbg = torch.zeros(NB * 8 * 4, dtype=torch.int64, device="cuda")
os.environ["UCFVIT_GEMM_DBG"] = str(dbg.data_ptr())
from UCF_VIT._hip import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
M = 166 * 197
for name, N, K in [("fc1", 4096, 1024), ("qkv", 3072, 1024), ("proj", 1024, 1024)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); dy = torch.randn(M, N, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    fn = (lambda: ops.linear_wgrad(dy, x)) if mode == "wgrad" else (lambda: ops.linear_fwd(x, w))
    for _ in range(3): fn()
    torch.cuda.synchronize(); dbg.zero_(); torch.cuda.synchronize()
    fn(); torch.cuda.synchronize()
    d = dbg.cpu().view(NB, 8, 4).double()
    t0 = d[:, 0, 0][d[:, 0, 0] > 0].min()
    for r in range(4):
        v = d[:, r, :]
        ok = v[:, 3] > 0
        if ok.sum() == 0: break
        v = v[ok]
        ml = (v[:, 1] - v[:, 0]) / 100
        print(f"{mode} {name} round {r}: n={int(ok.sum())} start {((v[:,0]-t0).mean()/100):7.2f} us | K loop {ml.mean():7.2f} (min {ml.min():6.2f} max {ml.max():6.2f}) | barrier {((v[:,2]-v[:,1]).mean()/100):6.2f} | epilogue {((v[:,3]-v[:,2]).mean()/100):6.2f} | end max {((v[:,3]-t0).max()/100):7.2f}")

#!/usr/bin/env python3
"""Diagnostic: per-tile phase timing of the persistent GEMM via s_memrealtime stamps (100 MHz)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
dbg = torch.zeros(256 * 8 * 4, dtype=torch.int64, device="cuda")
os.environ["UCFVIT_GEMM_DBG"] = str(dbg.data_ptr())
from UCF_VIT._hip import ops
M = 128 * 197
for name, N, K in [("proj", 1024, 1024), ("qkv", 3072, 1024), ("fc2", 1024, 4096)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    for _ in range(3): ops.linear_fwd(x, w, b)
    torch.cuda.synchronize(); dbg.zero_(); torch.cuda.synchronize()
    ops.linear_fwd(x, w, b); torch.cuda.synchronize()
    d = dbg.cpu().view(256, 8, 4).double()
    t0 = d[:, 0, 0].min()
    for r in range(3):
        v = d[:, r, :]
        ok = v[:, 3] > 0
        if ok.sum() == 0: break
        v = v[ok]
        print(f"{name} round {r}: n={int(ok.sum())} start {((v[:,0]-t0).mean()/100):7.2f} us | mainloop {((v[:,1]-v[:,0]).mean()/100):7.2f} (min {((v[:,1]-v[:,0]).min()/100):6.2f} max {((v[:,1]-v[:,0]).max()/100):6.2f}) | barrier {((v[:,2]-v[:,1]).mean()/100):6.2f} | epilogue {((v[:,3]-v[:,2]).mean()/100):6.2f} | end {((v[:,3]-t0).mean()/100):7.2f} max {((v[:,3]-t0).max()/100):7.2f}")

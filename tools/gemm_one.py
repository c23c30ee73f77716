#!/usr/bin/env python3
"""run one GEMM shape a few times (for rocprofv3 --pmc): python tools/gemm_one.py M N K [mode]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
from UCF_VIT._hip import ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "fwd"
x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
dy = torch.randn(M, N, device="cuda").bfloat16()
for _ in range(5):
    if mode == "fwd": ops.linear_fwd(x, w, b)
    elif mode == "dgrad": ops.linear_dgrad(dy, w)
    else: ops.linear_wgrad(dy, x)
torch.cuda.synchronize()

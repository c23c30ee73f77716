#!/usr/bin/env python3
"""One GEMM shape launched repeatedly (for rocprofv3 --pmc passes): python tools/gemm_one.py [name] [reps]; name in qkv|proj|fc1|fc2"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
from UCF_VIT._hip import ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "fc1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
M, D = 166 * 197, 1024
N, K = {"qkv": (3 * D, D), "proj": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D)}[name]
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
b = torch.randn(N, device="cuda").bfloat16()
for _ in range(reps):
    ops.linear_fwd(x, w, b)
torch.cuda.synchronize()
print(name, M, N, K, "done")

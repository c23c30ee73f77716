#!/usr/bin/env python3
"""GEMM micro-benchmark on the ViT shapes (run on the GPU box): correctness vs torch.matmul + HIP-event timing.
usage: python tools/gemm_bench.py [B] [D] [reps]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
from UCF_VIT._hip import ops  # noqa: E402
from UCF_VIT._hip.lib import ACT_GELU  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
M = B * 197
dev = "cuda"
torch.manual_seed(0)


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max()).item()


print(f"M={M} D={D}")
for name, N, K in [("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)]:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    dy = torch.randn(M, N, device=dev).bfloat16()
    flops = 2.0 * M * N * K
    # fwd
    y = ops.linear_fwd(x, w, b)
    ref = torch.nn.functional.linear(x, w, b)
    t = timeit(lambda: ops.linear_fwd(x, w, b), reps)
    t_ref = timeit(lambda: torch.nn.functional.linear(x, w, b), reps)
    print(f"{name:5s} fwd   N={N:5d} K={K:5d}: hip {t*1e3:8.1f} us {flops/t/1e9:7.1f} TF | torch {t_ref*1e3:8.1f} us {flops/t_ref/1e9:7.1f} TF | err {rel(y, ref):.1e}")
    # dgrad
    dx = ops.linear_dgrad(dy, w)
    ref = dy @ w
    t = timeit(lambda: ops.linear_dgrad(dy, w), reps)
    t_ref = timeit(lambda: dy @ w, reps)
    print(f"{name:5s} dgrad                  : hip {t*1e3:8.1f} us {flops/t/1e9:7.1f} TF | torch {t_ref*1e3:8.1f} us {flops/t_ref/1e9:7.1f} TF | err {rel(dx, ref):.1e}")
    # wgrad
    dw = ops.linear_wgrad(dy, x)
    ref = (dy.float().T @ x.float())
    t = timeit(lambda: ops.linear_wgrad(dy, x), reps)
    t_ref = timeit(lambda: dy.T @ x, reps)
    print(f"{name:5s} wgrad                  : hip {t*1e3:8.1f} us {flops/t/1e9:7.1f} TF | torch {t_ref*1e3:8.1f} us {flops/t_ref/1e9:7.1f} TF | err {rel(dw, ref):.1e}")

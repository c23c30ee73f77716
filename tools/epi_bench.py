#!/usr/bin/env python3
"""Epilogue cost of the forward / data-gradient GEMM on the ViT-L shapes (run on the GPU box).
usage: python tools/epi_bench.py [B] [reps]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
from UCF_VIT._hip import ops  # noqa: E402
from UCF_VIT._hip.lib import ACT_GELU, ACT_GELU_GRAD, ACT_GELU_SAVE_DERIV, ACT_MUL_AUX, ACT_NONE, LAYOUT_KC  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 166
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = B * 197
dev = "cuda"
torch.manual_seed(0)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def run(name, N, K, **kw):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    args = {}
    if kw.get("bias"):
        args["bias"] = torch.randn(N, device=dev).bfloat16()
    if kw.get("residual"):
        args["residual"] = torch.randn(M, N, device=dev).bfloat16()
    if kw.get("aux_in"):
        args["aux_in"] = torch.randn(M, N, device=dev).bfloat16()
    if kw.get("aux_out"):
        args["aux_out"] = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    act = kw.get("act", ACT_NONE)
    t = timeit(lambda: ops.gemm(x, w, M, N, K, LAYOUT_KC, LAYOUT_KC, out=out, act=act, **args))
    print(f"{name:34s} N={N:5d} K={K:5d}: {t:8.1f} us {2.0*M*N*K/t/1e6:7.1f} TF", flush=True)


print(f"M={M}")
run("proj  plain+bias", 1024, 1024, bias=True)
run("proj  bias+residual", 1024, 1024, bias=True, residual=True)
run("fc2   bias", 1024, 4096, bias=True)
run("fc2   bias+residual", 1024, 4096, bias=True, residual=True)
run("fc1   bias", 4096, 1024, bias=True)
run("fc1   bias+gelu", 4096, 1024, bias=True, act=ACT_GELU)
run("fc1   bias+gelu+aux_out", 4096, 1024, bias=True, act=ACT_GELU, aux_out=True)
run("fc2dg plain", 4096, 1024)
run("fc2dg gelu_grad(aux_in)", 4096, 1024, act=ACT_GELU_GRAD, aux_in=True)
run("fc1   bias+gelu+save deriv", 4096, 1024, bias=True, act=ACT_GELU_SAVE_DERIV, aux_out=True)
run("fc2dg mul_aux(aux_in)", 4096, 1024, act=ACT_MUL_AUX, aux_in=True)
run("qkv   bias", 3072, 1024, bias=True)

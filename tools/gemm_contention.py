#!/usr/bin/env python3
"""Static vs dynamic tile schedule of the persistent GEMM while another kernel holds some of the CUs (what an RCCL all-reduce that overlaps
backward does on a multi-GPU node; on this one-GPU box ucfvit_occupy stands in for it: `held` workgroups that each keep one CU for 2 ms).

    python tools/gemm_contention.py [held=32]

For each of {static, dynamic} x {free chip, `held` CUs taken}: the fc1-shaped forward GEMM of ViT-L at B = 166 (2048 tiles = 8 rounds)
and the grouped weight gradient of 4 Blocks (768 tiles), HIP-event time of the GEMM alone (the occupying kernel runs on its own stream
and is started first)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
from UCF_VIT._hip import lib, ops  # noqa: E402

held = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
L = lib.load()
g = torch.Generator().manual_seed(0)
M, D = 166 * 197, 1024
x = torch.randn(M, D, generator=g).bfloat16().to(dev)
w = (torch.randn(4 * D, D, generator=g) * 0.03).bfloat16().to(dev)
b = torch.randn(4 * D, generator=g).bfloat16().to(dev)
dy4 = torch.randn(M, 4 * D, generator=g).bfloat16().to(dev)
dy = torch.randn(M, D, generator=g).bfloat16().to(dev)
out = torch.empty(M, 4 * D, dtype=torch.bfloat16, device=dev)
dws = [torch.empty(4 * D, D, dtype=torch.float32, device=dev) for _ in range(4)]
sink = torch.zeros(1024, device=dev)
side = torch.cuda.Stream()


def fwd():
    ops.linear_fwd(x, w, b, out=out)


def wgrad():
    ops.wgrad_grouped([(dy4, x, dws[i], False) for i in range(4)])


def timed(fn, contended, reps=5):
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        if contended:
            with torch.cuda.stream(side):
                lib.check(L.ucfvit_occupy(held, 2000, sink.data_ptr(), side.cuda_stream), "ucfvit_occupy")
            torch.cuda._sleep(200000)            # ~0.1 ms on the compute stream: the occupying workgroups are resident before the GEMM starts
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts[1:])[len(ts[1:]) // 2]


print(f"{held} CUs held by another kernel; median of 5, us")
print(f"{'schedule':10s} {'fc1 fwd free':>14s} {'fc1 fwd held':>14s} {'wgrad x4 free':>14s} {'wgrad x4 held':>14s}")
for name, dyn in (("static", False), ("dynamic", True)):
    ops._dynamic_sched = dyn
    r = [timed(fwd, False), timed(fwd, True), timed(wgrad, False), timed(wgrad, True)]
    print(f"{name:10s} " + " ".join(f"{v:14.1f}" for v in r))

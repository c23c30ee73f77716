#!/usr/bin/env python3
"""the four weight gradients of a ViT-L Block: separate 128x128 split-K launches vs ONE grouped 256x256 ping-pong launch.
usage: python tools/wgrad_group_bench.py [B] [D] [reps]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
from UCF_VIT._hip import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 166
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
M = B * 197
dev = "cuda"
torch.manual_seed(0)
shapes = [(3 * D, D), (D, D), (4 * D, D), (D, 4 * D)]
dys = [torch.randn(M, n, device=dev).bfloat16() for n, _ in shapes]
xs = [torch.randn(M, k, device=dev).bfloat16() for _, k in shapes]
outs = [torch.empty(n, k, device=dev) for n, k in shapes]
flops = sum(2.0 * M * n * k for n, k in shapes)


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


def separate():
    for dy, x, o in zip(dys, xs, outs):
        ops.linear_wgrad(dy, x, out=o)


def grouped():
    ops.wgrad_grouped([(dy, x, o, False) for dy, x, o in zip(dys, xs, outs)])


separate()
ref = [o.clone() for o in outs]
grouped()
err = max(((o - r).abs().max() / r.abs().max()).item() for o, r in zip(outs, ref))
t1, t2 = timeit(separate), timeit(grouped)
print(f"M={M} D={D}: separate {t1:8.1f} us {flops/t1/1e6:7.1f} TF | grouped {t2:8.1f} us {flops/t2/1e6:7.1f} TF | max rel diff {err:.1e}")

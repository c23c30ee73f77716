#!/usr/bin/env python3
"""Does the row stride of a K = 4096 activation operand matter?  fc2-shaped forward GEMM (M x 1024 x 4096) with lda = 4096 vs padded.
usage: python tools/stride_bench.py [B] [reps]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
from UCF_VIT._hip import ops  # noqa: E402
from UCF_VIT._hip.lib import LAYOUT_KC  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 166
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
M, N, K = B * 197, 1024, 4096
dev = "cuda"


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


w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for pad in (0, 8, 64, 128, 256, 512):
    buf = torch.randn(M, K + pad, device=dev).bfloat16()
    x = buf[:, :K]
    t = timeit(lambda: ops.gemm(x, w, M, N, K, LAYOUT_KC, LAYOUT_KC, out=out))
    print(f"lda = {K + pad:5d} ({(K + pad) * 2} B): {t:7.1f} us {2.0 * M * N * K / t / 1e6:7.1f} TF", flush=True)
# and the output side: C with N = 4096 columns, K = 1024 (fc1 shape), ldc padded
w2 = (torch.randn(4096, 1024, device=dev) * 0.05).bfloat16()
x2 = torch.randn(M, 1024, device=dev).bfloat16()
for pad in (0, 64, 256):
    obuf = torch.empty(M, 4096 + pad, device=dev, dtype=torch.bfloat16)
    o = obuf[:, :4096]
    t = timeit(lambda: ops.gemm(x2, w2, M, 4096, 1024, LAYOUT_KC, LAYOUT_KC, out=o))
    print(f"ldc = {4096 + pad:5d}: {t:7.1f} us {2.0 * M * 4096 * 1024 / t / 1e6:7.1f} TF", flush=True)
print("--- K = 1024 operand (qkv forward shape, N = 3072)")
w3 = (torch.randn(3072, 1024, device=dev) * 0.05).bfloat16()
o3 = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
for pad in (0, 8, 64, 128):
    buf = torch.randn(M, 1024 + pad, device=dev).bfloat16()
    xx = buf[:, :1024]
    t = timeit(lambda: ops.gemm(xx, w3, M, 3072, 1024, LAYOUT_KC, LAYOUT_KC, out=o3))
    print(f"lda = {1024 + pad:5d}: {t:7.1f} us {2.0 * M * 3072 * 1024 / t / 1e6:7.1f} TF", flush=True)
print("--- K = 3072 operand (qkv data-gradient shape, N = 1024)")
w4 = (torch.randn(1024, 3072, device=dev) * 0.05).bfloat16()
o4 = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
for pad in (0, 64, 128):
    buf = torch.randn(M, 3072 + pad, device=dev).bfloat16()
    xx = buf[:, :3072]
    t = timeit(lambda: ops.gemm(xx, w4, M, 1024, 3072, LAYOUT_KC, LAYOUT_KC, out=o4))
    print(f"lda = {3072 + pad:5d}: {t:7.1f} us {2.0 * M * 1024 * 3072 / t / 1e6:7.1f} TF", flush=True)
print("--- weight-gradient (KS x KS) fc2 shape: dW[1024,4096] = dy[M,1024]^T a[M,4096], a padded")
from UCF_VIT._hip.lib import LAYOUT_KS  # noqa: E402
dy = torch.randn(M, 1024, device=dev).bfloat16()
dw = torch.empty(1024, 4096, device=dev)
for pad in (0, 64):
    buf = torch.randn(M, 4096 + pad, device=dev).bfloat16()
    aa = buf[:, :4096]
    t = timeit(lambda: ops.gemm(dy, aa, 1024, 4096, M, LAYOUT_KS, LAYOUT_KS, out=dw, out_dtype=torch.float32))
    print(f"ld(a) = {4096 + pad:5d}: {t:7.1f} us {2.0 * M * 1024 * 4096 / t / 1e6:7.1f} TF", flush=True)
print("--- weight operand B [1024, 4096] (fc2 forward), A padded to 4160, ldb varied")
bufa = torch.randn(M, 4096 + 64, device=dev).bfloat16()
xa = bufa[:, :4096]
for pad in (0, 8, 64, 128):
    wb = (torch.randn(1024, 4096 + pad, device=dev) * 0.05).bfloat16()
    ww = wb[:, :4096]
    t = timeit(lambda: ops.gemm(xa, ww, M, 1024, 4096, LAYOUT_KC, LAYOUT_KC, out=out))
    print(f"ldb = {4096 + pad:5d}: {t:7.1f} us {2.0 * M * 1024 * 4096 / t / 1e6:7.1f} TF", flush=True)

#!/usr/bin/env python3
"""GPU quadtree patcher vs the CPU oracle (the reference's per-image Python loop restated): images/s at 256^2, fixed_length 196, p 16."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
sys.path.insert(0, ROOT)
from UCF_VIT.dataloaders.transform import Patchify  # noqa: E402
from oracle import quadtree_ref as QR  # noqa: E402

B, H, L, p = 166, 256, 196, 16      # a power of two: every leaf stays square (the reference asserts that, quadtree.py:157)
rng = np.random.Generator(np.random.PCG64(0))
yy, xx = np.ogrid[:H, :H]
maps = []
for b in range(B):                                   # a few circle outlines per image, like object contours
    m = np.zeros((H, H), dtype=bool)
    for _ in range(6):
        cy, cx, r = rng.integers(0, H), rng.integers(0, H), rng.integers(8, 80)
        m |= np.abs(np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2) - r) < 1.0
    maps.append(m.astype(np.uint8) * 255)
maps = np.stack(maps)
imgs = rng.random((B, H, H, 3)).astype(np.float32) * 255
e, x = torch.from_numpy(maps).cuda(), torch.from_numpy(imgs).cuda()
pt = Patchify(L, p, 3)
for _ in range(3):
    out = pt(x, e)
torch.cuda.synchronize()
s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
reps = 20
for _ in range(reps):
    out = pt(x, e)
t.record()
torch.cuda.synchronize()
ms = s.elapsed_time(t) / reps
print(f"HIP  : batch {B} x {H}^2 -> {L} tokens of {p}^2: {ms:.3f} ms per batch = {B / ms * 1e3:.0f} images/s")
t0 = time.perf_counter()
n = 8
for b in range(n):
    nodes, _ = QR.build_tree(maps[b], L)
    QR.serialize(imgs[b], nodes, L, p)
dt = time.perf_counter() - t0
print(f"CPU oracle (1 thread Python, {n} images): {n / dt:.1f} images/s")

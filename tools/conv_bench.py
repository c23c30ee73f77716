#!/usr/bin/env python3
"""The 3x3x3 convolution launches of the UNETR decoder at BASELINE config 5 sizes (B = 2), one by one: HIP-event timing of forward / data
gradient / weight gradient, algorithmic TFLOP/s and GB/s.   python tools/conv_bench.py [reps=5]      (UCFVIT_CONV_STRIP=0: one tile per workgroup)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
from UCF_VIT._hip import conv, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = 2
LAYERS = [  # X, Y, Z, Cin, Cout, label
    (512, 512, 128, 8, 16, "encoder1.conv1 (input volume, 1 -> 16)"),
    (512, 512, 128, 16, 16, "encoder1.conv2 / decoder2.conv2"),
    (512, 512, 128, 32, 16, "decoder2.conv1"),
    (512, 512, 128, 16, 32, "decoder2.conv1 data gradient"),
    (256, 256, 64, 32, 32, "encoder2 / decoder3 32 -> 32"),
    (256, 256, 64, 64, 32, "decoder3.conv1"),
    (256, 256, 64, 32, 64, "decoder3.conv1 data gradient"),
    (128, 128, 32, 64, 64, "encoder3 / decoder4 64 -> 64"),
    (128, 128, 32, 128, 64, "decoder4.conv1"),
    (64, 64, 16, 256, 128, "decoder5.conv1"),
]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


print(f"{'layer':46s} {'voxels':>9s} {'fwd ms':>8s} {'TF/s':>7s} {'GB/s':>7s} | {'wgrad ms':>8s} {'TF/s':>7s}")
for X, Y, Z, cin, cout, label in LAYERS:
    V = B * X * Y * Z
    x = torch.randn(B, X, Y, Z, cin, device="cuda").bfloat16()
    dy = torch.randn(B, X, Y, Z, cout, device="cuda").bfloat16()
    wp = conv.pack_conv_weight(torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05)
    flops = 2.0 * 27 * V * cin * cout
    t_f = timed(lambda: ops.conv3d_fwd(x, wp, cout))
    t_w = timed(lambda: ops.conv3d_wgrad(x, dy))
    gb = V * (cin + cout) * 2 / 1e9
    print(f"{label:46s} {V:9d} {t_f:8.3f} {flops / t_f / 1e9:7.1f} {gb / t_f * 1e3:7.0f} | {t_w:8.3f} {flops / t_w / 1e9:7.1f}")
    del x, dy

#!/usr/bin/env python3
"""Diagnostic: shader-clock stamps of one workgroup (middle of the grid) of the fused short-sequence attention backward
(csrc/attention_short.hip built with -DAG_STAMP into a scratch library; the product library carries no stamps).

    python tools/attn_stamps.py [B=665] [N=197] [H=16] [dh=64]
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ucf-vit_amd")
out_dir = os.path.join(ROOT, "gpurun_out", "stamps")
os.makedirs(out_dir, exist_ok=True)
lib = os.path.join(out_dir, "libucfvit_agstamp.so")
objs = [os.path.join(PKG, "build", f) for f in os.listdir(os.path.join(PKG, "build")) if f.endswith(".o") and f != "attention_short.o"]
obj = os.path.join(out_dir, "attention_short_stamp.o")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-DAG_STAMP", "-c",
                os.path.join(PKG, "csrc", "attention_short.hip"), "-o", obj], check=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + objs, check=True)
os.environ["UCFVIT_HIP_LIB"] = lib
sys.path.insert(0, PKG)
import torch
from UCF_VIT._hip import ops, lib as L
B, N, H, dh = [int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 665), (2, 197), (3, 16), (4, 64))]
qkv = torch.randn(B * N, 3 * H * dh, device="cuda").bfloat16()
do = torch.randn(B * N, H * dh, device="cuda").bfloat16()
o, lse = ops.attention_fwd(qkv, B, N, H, dh, dh ** -0.5)
for _ in range(3):
    ops.attention_bwd(qkv, o, do, lse, B, N, H, dh, dh ** -0.5)
torch.cuda.synchronize()
h = L.load()
buf = (ctypes.c_ulonglong * 128)()
h.ucfvit_debug_attn_stamps(buf)
names = {0: "start", 1: "K,V images stored", 2: "first barrier passed", 8: "phase A done", 9: "barrier", 10: "Q,dO stored + barrier", 16: "end"}
for w in range(4):
    t = [buf[w * 32 + k] for k in range(32)]
    t0 = t[0]
    line = []
    order = [0, 1, 2, 17, 18, 19, 20, 21, 22, 23, 3, 24, 25, 26, 27, 28, 29, 30, 4, 8, 9, 10, 11, 12, 16]
    names.update({3: "A1 done", 4: "A2 done", 11: "B1 done", 12: "B2 done"})
    names.update({17 + i: f"A1c{i}" for i in range(7)})
    names.update({24 + i: f"A2c{i}" for i in range(7)})
    for k in order:
        if t[k]:
            line.append(f"{names.get(k, 'pass')}@{t[k] - t0}")
    print(f"wave {w}: " + "  ".join(line))
print("(s_memtime: shader clocks)")

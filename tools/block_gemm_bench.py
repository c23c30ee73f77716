#!/usr/bin/env python3
"""The GEMM launches of one transformer Block exactly as the training step issues them (fused epilogues included), against torch.matmul
(hipBLASLt) on the same operands: HIP-event timing, interleaved rounds in one process, median over rounds, random data.

    python tools/block_gemm_bench.py [B=166] [D=1024] [rounds=7] [reps=6]          (UCFVIT_HIP_LIB selects another build for an A/B)

forward:  qkv  [M,D]x[3D,D]^T + bias                     proj [M,D]x[D,D]^T + bias + residual
          fc1  [M,D]x[4D,D]^T + bias, GELU, saves gelu'   fc2  [M,4D]x[D,4D]^T + bias + residual
backward: the four data gradients through the transposed weight shadow (KC x KC); fc2's multiplies by the saved gelu' and takes the
          column sums (fc1 bias gradient); the four weight gradients as ONE grouped launch (KS x KS, fp32 out)."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
from UCF_VIT._hip import ops  # noqa: E402
from UCF_VIT._hip.lib import ACT_GELU_SAVE_DERIV  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 166
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
N_TOK = int(os.environ.get("N_TOK", "197"))
M = B * N_TOK
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(0)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).bfloat16().to(dev)


x, x4 = rnd(M, D), rnd(M, 4 * D)
res = rnd(M, D)
dy, dy3, dy4 = rnd(M, D), rnd(M, 3 * D), rnd(M, 4 * D)
w = {"qkv": rnd(3 * D, D, scale=0.03), "proj": rnd(D, D, scale=0.03), "fc1": rnd(4 * D, D, scale=0.03), "fc2": rnd(D, 4 * D, scale=0.03)}
wT = {k: v.T.contiguous() for k, v in w.items()}
b = {k: rnd(v.shape[0]) for k, v in w.items()}
aux = torch.empty(M, 4 * D, dtype=torch.bfloat16, device=dev)
gp = rnd(M, 4 * D)
cs = torch.empty(4 * D, dtype=torch.float32, device=dev)
o_qkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
o_d = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
o_4d = torch.empty(M, 4 * D, dtype=torch.bfloat16, device=dev)
dws = [torch.empty(v.shape, dtype=torch.float32, device=dev) for v in w.values()]

CASES = [
    # name, flops, hip launch, torch equivalent (plain matmul: what hipBLASLt does WITHOUT the fused epilogue traffic)
    ("qkv  fwd  N=3D K=D  +bias", 2.0 * M * 3 * D * D, lambda: ops.linear_fwd(x, w["qkv"], b["qkv"], out=o_qkv), lambda: torch.matmul(x, wT["qkv"], out=o_qkv)),
    ("proj fwd  N=D  K=D  +bias+res", 2.0 * M * D * D, lambda: ops.linear_fwd(x, w["proj"], b["proj"], residual=res, out=o_d), lambda: torch.matmul(x, wT["proj"], out=o_d)),
    ("fc1  fwd  N=4D K=D  +bias,gelu,gelu'", 2.0 * M * 4 * D * D, lambda: ops.linear_fwd(x, w["fc1"], b["fc1"], act=ACT_GELU_SAVE_DERIV, aux_out=aux, out=o_4d),
     lambda: torch.matmul(x, wT["fc1"], out=o_4d)),
    ("fc2  fwd  N=D  K=4D +bias+res", 2.0 * M * 4 * D * D, lambda: ops.linear_fwd(x4, w["fc2"], b["fc2"], residual=res, out=o_d), lambda: torch.matmul(x4, wT["fc2"], out=o_d)),
    ("fc2  dgrad N=4D K=D  *gelu',colsum", 2.0 * M * 4 * D * D,
     lambda: ops.linear_dgrad_t(dy, wT["fc2"], act_grad_aux=gp, aux_is_deriv=True, c_colsum=cs, out=o_4d), lambda: torch.matmul(dy, w["fc2"], out=o_4d)),
    ("fc1  dgrad N=D  K=4D", 2.0 * M * 4 * D * D, lambda: ops.linear_dgrad_t(dy4, wT["fc1"], out=o_d), lambda: torch.matmul(dy4, w["fc1"], out=o_d)),
    ("proj dgrad N=D  K=D", 2.0 * M * D * D, lambda: ops.linear_dgrad_t(dy, wT["proj"], out=o_d), lambda: torch.matmul(dy, w["proj"], out=o_d)),
    ("qkv  dgrad N=D  K=3D", 2.0 * M * 3 * D * D, lambda: ops.linear_dgrad_t(dy3, wT["qkv"], out=o_d), lambda: torch.matmul(dy3, w["qkv"], out=o_d)),
    ("wgrad x4 grouped (fp32 out)", 2.0 * M * 12 * D * D,
     lambda: ops.wgrad_grouped([(dy3, x, dws[0], False), (dy, x, dws[1], False), (dy4, x, dws[2], False), (dy, x4, dws[3], False)]),
     lambda: (torch.matmul(dy3.T, x), torch.matmul(dy.T, x), torch.matmul(dy4.T, x), torch.matmul(dy.T, x4))),
]


def check():
    y = ops.linear_fwd(x, w["proj"], b["proj"], residual=res)
    ref = x.float() @ w["proj"].float().T + b["proj"].float() + res.float()
    e1 = ((y.float() - ref).abs().max() / ref.abs().max()).item()
    y = ops.linear_dgrad_t(dy, wT["fc2"], act_grad_aux=gp, aux_is_deriv=True, c_colsum=cs)
    ref = (dy.float() @ w["fc2"].float()) * gp.float()
    e2 = ((y.float() - ref).abs().max() / ref.abs().max()).item()
    e3 = ((cs - y.float().sum(0)).abs().max() / y.float().sum(0).abs().max()).item()
    print(f"check: proj fwd rel err {e1:.2e}, fc2 dgrad {e2:.2e}, its column sums {e3:.2e}")
    assert e1 < 1e-2 and e2 < 1e-2 and e3 < 1e-2


def time_once(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


if not os.environ.get("SKIP_CHECK"):
    check()
for _, _, f_hip, f_ref in CASES:        # warm-up (hipBLASLt heuristics, code objects)
    for _ in range(2):
        f_hip()
        f_ref()
torch.cuda.synchronize()
th = {c[0]: [] for c in CASES}
tr = {c[0]: [] for c in CASES}
for _ in range(rounds):
    for name, _, f_hip, f_ref in CASES:
        th[name].append(time_once(f_hip))
        tr[name].append(time_once(f_ref))
print(f"lib: {os.environ.get('UCFVIT_HIP_LIB', 'in-tree default')}    M = {B} x {N_TOK} = {M}, D = {D}; median of {rounds} rounds x {reps} launches")
print(f"{'launch':40s} {'hip us':>9s} {'TFLOP/s':>8s} | {'torch us':>9s} {'TFLOP/s':>8s} | hip/torch time")
tot_h = tot_r = tot_f = 0.0
for name, fl, _, _ in CASES:
    a, r = statistics.median(th[name]), statistics.median(tr[name])
    tot_h, tot_r, tot_f = tot_h + a, tot_r + r, tot_f + fl
    print(f"{name:40s} {a * 1e3:9.1f} {fl / a / 1e9:8.1f} | {r * 1e3:9.1f} {fl / r / 1e9:8.1f} | {a / r:5.2f}")
print(f"{'Block total':40s} {tot_h * 1e3:9.1f} {tot_f / tot_h / 1e9:8.1f} | {tot_r * 1e3:9.1f} {tot_f / tot_r / 1e9:8.1f} | {tot_h / tot_r:5.2f}")

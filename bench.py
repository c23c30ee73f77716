#!/usr/bin/env python3
"""Benchmark of the ViT training hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward + cross-entropy + backward (+ data-parallel gradient all-reduce) + fused AdamW + zero_grad +
scheduler step on one synthetic batch that is already resident in HBM.  Rank 0 prints ONE JSON line.

metric : images/sec (whole job) for ViT-L/16 224^2, bf16 compute with fp32 master weights (BASELINE.json).
roofline: the dominant kernel is the bf16 MFMA GEMM (gemm3_kernel<...>, csrc/gemm2.hip); `achieved` = algorithmic FLOPs of all
          GEMM launches in the timed region (forward, data gradient, grouped weight gradient) / their HIP-event-measured
          durations (events recorded on the launch stream).
cpu_baseline: the CPU oracle's training loop (oracle/ucf_vit_ref.py, kind "port") on this box's host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "ucf-vit_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (img, patch, dim, depth, heads, classes, default per-GPU batch)
    "vit_l16_224": dict(img=224, patch=16, dim=1024, depth=24, heads=16, classes=1000, batch=166),   # 166*197 = 32702 rows -> 128 M-tiles of 256: every GEMM fills whole rounds of 256 CUs
    "vit_b16_224": dict(img=224, patch=16, dim=768, depth=12, heads=12, classes=1000, batch=332),   # 65404 rows -> 256 M-tiles: 3 N-tiles of 256 fill whole rounds
    "vit_tiny16_256": dict(img=256, patch=16, dim=192, depth=12, heads=3, classes=2, batch=256),
    # SURVEY §8f row 1 (not the headline metric): the reference's imagenet config shape, adaptive_patching with fixed_length 196 and
    # use_adaptive_pos_emb (configs/imagenet/classification/base_config.yaml:46-49); input = token sequences [B, 3, 196, 256] + seq_ps
    "vit_l16_adaptive196": dict(img=224, patch=16, dim=1024, depth=24, heads=16, classes=1000, batch=166, adaptive=196),
}
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_F32_TFLOPS = 157.3


def train_flops_per_image(w):
    """algorithmic FLOPs of one training step per image = 3 x forward"""
    n = (w["img"] // w["patch"]) ** 2 + 1
    d = w["dim"]
    layer = 24 * n * d * d + 4 * n * n * d
    patch = 2 * (n - 1) * (3 * w["patch"] ** 2) * d
    head = 2 * d * w["classes"]
    return 3 * (w["depth"] * layer + patch + head)


class GemmProfiler:
    """HIP-event timing of every ucfvit_gemm / ucfvit_gemm_grouped launch in the timed region (events on the launch stream)."""

    def __init__(self):
        self.records = []   # (start_evt, end_evt, flops, is_mfma_path)
        self.enabled = False

    def install(self):
        from UCF_VIT._hip import ops
        orig = ops.gemm
        prof = self

        def timed_gemm(A, B, M, N, K, *a, **kw):
            if not prof.enabled:
                return orig(A, B, M, N, K, *a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig(A, B, M, N, K, *a, **kw)
            e.record()
            prof.records.append((s, e, 2.0 * M * N * K))
            return out
        ops.gemm = timed_gemm
        orig_grouped = ops.wgrad_grouped

        def timed_grouped(items):
            if not prof.enabled:
                return orig_grouped(items)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig_grouped(items)
            e.record()
            prof.records.append((s, e, sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _, _ in items)))
            return out
        ops.wgrad_grouped = timed_grouped

    def summary(self):
        tot_ms, tot_flops = 0.0, 0.0
        for s, e, f in self.records:
            tot_ms += s.elapsed_time(e)
            tot_flops += f
        n = len(self.records)
        return n, tot_ms, tot_flops


def pmc_traffic(workload, dtype, batch):
    """HBM bytes per launch of the dominant kernel (average over every gemm3_kernel launch of the step) from the committed rocprofv3
    PMC passes of this same command (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, newest profiles/*pmc_traffic*.json whose
    workload / dtype / batch match) — counters cannot be read from inside the process; null if no matching profile."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            d = json.load(open(f))
            if d.get("workload") == workload and d.get("dtype") == dtype and d.get("per_gpu_batch") == batch:
                k = d["kernels"]
                if "ALL gemm3_kernel launches" in k:
                    return round(k["ALL gemm3_kernel launches"]["hbm_bytes_per_launch_corrected"])
        except Exception:
            pass
    return None


def cpu_baseline(wname, w, steps=3, batch=8):
    """reference training loop (train_class_simple.py:344-357) restated on CPU fp32: oracle, timed on the host cores"""
    from oracle import ucf_vit_ref as R
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))      # the GPU box grants a 16-CPU share per GPU; more threads only oversubscribe it
    torch.set_num_threads(cores)
    if w.get("adaptive"):
        return None          # the headline baseline is the image-input loop; the adaptive oracle is timed by nothing
    m = R.VIT([w["img"], w["img"]], patch_size=w["patch"], in_chans=3, num_classes=w["classes"], embed_dim=w["dim"], depth=w["depth"],
              num_heads=w["heads"], sdpa=True)
    opt = R.configure_optimizer(m, 1e-4, 0.9, 0.95, 1e-5)
    sch = R.WarmupCosineLR(opt, 1000, 20000, 1e-8, 1e-8)
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 256, (batch, 3, w["img"], w["img"]), generator=g).float()
    y = torch.randint(0, w["classes"], (batch,), generator=g)
    t0 = time.perf_counter()
    R.train_step_class(m, opt, sch, x, y)     # warm-up
    warm = time.perf_counter() - t0
    print(f"[bench] cpu_baseline warm-up step {warm:.1f} s on {cores} threads", file=sys.stderr, flush=True)
    steps = max(1, min(steps, int(25.0 / max(warm, 1e-3))))     # keep the whole leg to ~10-30 s of CPU work
    t0 = time.perf_counter()
    for i in range(steps):
        R.train_step_class(m, opt, sch, x, y)
        print(f"[bench] cpu_baseline step {i + 1}/{steps}", file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    return {"value": batch * steps / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{wname} fp32 CPU oracle train step, batch {batch}, {steps} timed steps after 1 warm-up ({dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="vit_l16_224", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: workload default)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    # Rehearsal switches (never set by the driver): UCFVIT_BENCH_BACKEND=gloo + UCFVIT_BENCH_ONE_GPU=1 run the N > 1 code path with
    # every rank on GPU 0 and host-staged gradient reduction, because RCCL refuses two ranks on one device (1-GPU development box).
    backend = os.environ.get("UCFVIT_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("UCFVIT_BENCH_ONE_GPU") else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.fused_attn import FusedAttn
    from UCF_VIT.utils.metrics import cross_entropy_loss
    from UCF_VIT.utils.misc import configure_optimizer, configure_scheduler
    from UCF_VIT._hip import lib
    lib.load()   # fail loudly if the HIP library is missing

    w = WORKLOADS[args.workload]
    B = args.batch or w["batch"]
    cdtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    adaptive = w.get("adaptive", 0)
    akw = dict(adaptive_patching=True, fixed_length=adaptive, use_adaptive_pos_emb=True) if adaptive else {}
    model = VIT(img_size=[w["img"], w["img"]], patch_size=w["patch"], in_chans=3, num_classes=w["classes"], embed_dim=w["dim"],
                depth=w["depth"], num_heads=w["heads"], mlp_ratio=4.0, FusedAttn_option=FusedAttn.HIP, **akw).to(dev)
    model.set_compute_dtype(cdtype)
    net = model
    if world > 1:
        from UCF_VIT._hip.ddp import HipDataParallel
        net = HipDataParallel(model)
    opt = configure_optimizer(model, 1e-4, 0.9, 0.95, 1e-5)        # configs/*/base_config.yaml: lr, betas, wd
    sch = configure_scheduler(opt, 1000, 20000, 1e-8, 1e-8)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)      # per-rank synthetic shard
    seq_ps = None
    if adaptive:     # pre-cut, resized patches as the reference's quadtree dataloader emits them, (size, x, y) per token
        x = torch.randint(0, 256, (B, 3, adaptive, w["patch"] ** 2), generator=g).float().to(dev)
        seq_ps = torch.cat([2.0 ** torch.randint(2, 7, (B, adaptive, 1), generator=g).float(),
                            torch.randint(0, w["img"], (B, adaptive, 2), generator=g).float()], dim=-1).to(dev)
    else:
        x = torch.randint(0, 256, (B, 3, w["img"], w["img"]), generator=g).float().to(dev)   # un-normalised pixels, resident in HBM
    y = torch.randint(0, w["classes"], (B,), generator=g).to(dev)
    variables = ["red", "green", "blue"]

    prof = GemmProfiler()
    prof.install()

    def step():
        out = net(x, variables, seq_ps)
        loss = cross_entropy_loss(out, y)
        loss.backward()
        opt.step()
        opt.zero_grad()
        sch.step()
        return loss

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    log(f"model built: {args.workload} {args.dtype} per-GPU batch {B}, world {world}")
    for i in range(args.warmup):
        loss = step()
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    torch.cuda.synchronize()
    log("warm-up done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    prof.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof.enabled = False
    log(f"timed region: {args.steps} steps in {dt:.3f} s")
    final_loss = float(loss.item())
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # measured MFMA ceiling of THIS device (outside the timed region): a pure bf16 MFMA stream, ~0.3 s so the clock settles the way
    # it does inside the step (the chip lowers its clock under matrix load; the nominal 2.5 PFLOP/s assumes 2.4 GHz)
    mfma_stream = None
    if rank == 0 and args.dtype == "bf16":
        from UCF_VIT._hip import ops as _ops
        for _ in range(20):
            _ops.mfma_probe()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fl = 0
        for _ in range(60):
            fl += _ops.mfma_probe()
        e1.record()
        torch.cuda.synchronize()
        mfma_stream = fl / (e0.elapsed_time(e1) * 1e-3) / 1e12

    if rank == 0:
        imgs = world * B * args.steps
        value = imgs / dt
        n_g, g_ms, g_flops = prof.summary()
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        achieved = (g_flops / (g_ms * 1e-3)) / 1e12 if g_ms > 0 else 0.0
        step_tflops = value / world * train_flops_per_image(w) / 1e12
        res = {
            "metric": "images/sec/node ViT-L/16 224^2 bf16 train step" if args.workload == "vit_l16_224" and args.dtype == "bf16"
            else f"images/sec/node {args.workload} {args.dtype} train step",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload} train step (fwd+bwd+AdamW), synthetic U{{0..255}} images resident in HBM",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}", "final_loss": round(final_loss, 4),
                       "grad_all_reduce": (str(net.reduce_dtype()).replace("torch.", "") + " mean, " + ("RCCL" if backend == "nccl" else backend + " (rehearsal)") + ", overlapped with backward") if world > 1 else "none (1 rank)"},
            "roofline": {"bound": "mfma", "kernel": "gemm3_kernel<%s> 256x256x64 persistent ping-pong MFMA GEMM (every forward, data-gradient and grouped "
                                             "weight-gradient launch of the timed region)" % args.dtype,
                         "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": pmc_traffic(args.workload, args.dtype, B), "launches": n_g, "avg_launch_ms": round(g_ms / max(n_g, 1), 4),
                         "avg_launch_gflop": round(g_flops / max(n_g, 1) / 1e9, 2),
                         "gemm_time_share_of_step": round(g_ms * 1e-3 / dt, 3),
                         "whole_step_tflops_per_gpu": round(step_tflops, 1), "whole_step_frac": round(step_tflops / peak, 4)},
        }
        if mfma_stream:
            # context, not the contract's `peak`: what a register-only MFMA loop sustains on this device under its power management
            res["roofline"]["mfma_stream_measured"] = round(mfma_stream, 1)
            res["roofline"]["frac_of_mfma_stream"] = round(achieved / mfma_stream, 4)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.workload, w, steps=args.cpu_steps)
            if cb is not None:
                res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the ViT training hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...                      (starts N ranks itself, one per GPU, over RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward + loss + backward (+ data-parallel gradient all-reduce) + fused AdamW + zero_grad + scheduler step on one
synthetic batch that is already resident in HBM.  Rank 0 prints ONE JSON line.

workloads (BASELINE.json configs):
  vit_l16_224            headline: ViT-L/16 224^2 classification step (train_class_simple.py:344-357), configs[2]
  vit_b16_224            configs[1]
  mae_vit_l16_224        configs[3]: MAE ViT-L/16 mask ratio 0.75, decoder 8 x 512 / 16 heads, MSE over all patches
                         (train_masked_simple.py:35-49, configs/imagenet/mae/base_config.yaml:44-46)
  unetr_enc_512x512x128  configs[4], encoder: 512x512x128 volumes, p 16 -> 8192 tokens, D 768 / 12 / 12, taps after blocks 3, 6, 9
                         (simple/arch.py:995-1086); synthetic quadratic objective on the taps + features (the conv decoder and the
                         Dice/CE loss are SURVEY §8f row 2, outside this figure: BASELINE.md §4 "conv decoder excluded")
  unetr_512x512x128      configs[4], whole model: the encoder above + the skip-connection convolutional decoder (UnetrBasicBlock / UnetrPrUpBlock /
                         UnetrUpBlock, feature_size 16, simple/arch.py:808-940) on the HIP convolution kernels + Dice/CE loss
                         (train_unetr_simple.py:38) on 4 classes
roofline: the dominant kernel family of the workload — the bf16 MFMA GEMM (gemm3_kernel in csrc/gemm2.hip and its staggered form gemm5_kernel in csrc/gemm_stagger.hip) for the 224^2 workloads, the
          streaming attention kernels (csrc/attention.hip) for the 8192-token encoder; `achieved` = algorithmic FLOPs of that family's
          launches in the timed region / their HIP-event-measured durations (events recorded on the launch stream).
cpu_baseline: the CPU oracle's training loop (oracle/ucf_vit_ref.py, kind "port") on this box's host cores, rank 0, N=1 only.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "ucf-vit_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

WORKLOADS = {
    # Per-GPU batches are chosen so that the token rows fill a multiple of 64 M-tiles of 256 rows: every GEMM of the step (3 ... 16 N-tiles) is then
    # a whole number of rounds of the 256 CUs (one M-tile more costs a whole extra round: MAE 334 -> 336 images loses 14 %).  Among those batches the
    # large ones win: the per-step constants (AdamW over 304 M parameters = 1.8 ms, weight-shadow transposes, launch tails) are paid once per step and
    # a data-parallel gradient all-reduce has 4x the backward time to hide in — 288 GB of HBM hold them easily (ViT-L at 665 images: see DESIGN §6a).
    # 665*197 = 131005 rows -> 512 M-tiles (166: 128 M-tiles, 2393 images/s; 332: 2491; 498: 2536; 665: 2554 on the same box)
    "vit_l16_224": dict(kind="vit", img=224, patch=16, dim=1024, depth=24, heads=16, classes=1000, batch=665),
    "vit_b16_224": dict(kind="vit", img=224, patch=16, dim=768, depth=12, heads=12, classes=1000, batch=1330),   # 262010 rows -> 1024 M-tiles
    "vit_tiny16_256": dict(kind="vit", img=256, patch=16, dim=192, depth=12, heads=3, classes=2, batch=256),
    # SURVEY §8f row 1 (not the headline metric): the reference's imagenet config shape, adaptive_patching with fixed_length 196 and
    # use_adaptive_pos_emb (configs/imagenet/classification/base_config.yaml:46-49); input = token sequences [B, 3, 196, 256] + seq_ps
    "vit_l16_adaptive196": dict(kind="vit", img=224, patch=16, dim=1024, depth=24, heads=16, classes=1000, batch=665, adaptive=196),
    # 1002 * 49 = 49098 encoder rows -> 192 M-tiles of 256, 1002 * 196 = 196392 decoder rows -> 768 M-tiles (batch 256: 49 / 196 M-tiles, 77 % of
    # the last round idle: 5442 images/s; 334: 6086; 336 — one tile more — 5233; 668: 6435; 1002: 6561)
    "mae_vit_l16_224": dict(kind="mae", img=224, patch=16, dim=1024, depth=24, heads=16, batch=1002, mask_ratio=0.75,
                            dec_dim=512, dec_depth=8, dec_heads=16),
    # per-GPU batch 2 = the reference's basic_ct batch size (configs/basic_ct/unetr/base_config.yaml:82)
    "unetr_enc_512x512x128": dict(kind="unetr", vol=(512, 512, 128), patch=16, dim=768, depth=12, heads=12, batch=2),
    "unetr_512x512x128": dict(kind="unetr", vol=(512, 512, 128), patch=16, dim=768, depth=12, heads=12, batch=2, decoder=True, fs=16, classes=4),
}
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_F32_TFLOPS = 157.3


def _layer_flops(n, d):
    return 24 * n * d * d + 4 * n * n * d


def train_flops_per_unit(w):
    """algorithmic FLOPs of one training step per image / volume = 3 x forward"""
    if w["kind"] == "vit":
        n = (w["img"] // w["patch"]) ** 2 + 1
        d = w["dim"]
        patch = 2 * (n - 1) * (3 * w["patch"] ** 2) * d
        return 3 * (w["depth"] * _layer_flops(n, d) + patch + 2 * d * w["classes"])
    if w["kind"] == "mae":
        L = (w["img"] // w["patch"]) ** 2
        keep = int(L * (1 - w["mask_ratio"]))
        d, dd, P = w["dim"], w["dec_dim"], 3 * w["patch"] ** 2
        fwd = (w["depth"] * _layer_flops(keep, d) + 2 * L * P * d + 2 * keep * d * dd + w["dec_depth"] * _layer_flops(L, dd) + 2 * L * dd * P)
        return 3 * fwd
    n = 1
    for s in w["vol"]:
        n *= s // w["patch"]
    fwd = w["depth"] * _layer_flops(n, w["dim"]) + 2 * n * w["patch"] ** 3 * w["dim"]
    if w.get("decoder"):
        fwd += unetr_decoder_fwd_flops(n, w["dim"], w["fs"], w["classes"])
    return 3 * fwd


def unetr_decoder_fwd_flops(n, D, fs, classes, in_chans=1):
    """forward FLOPs of the skip-connection decoder on a grid of n tokens (monai block structure, simple/unetr_blocks.py): a residual block
    Cin -> Cout at V voxels = 2 V (27 Cin Cout + 27 Cout^2 + [Cin != Cout] Cin Cout); a transposed 2x2x2 convolution Cin -> Cout FROM V
    voxels = 2 V Cin 8 Cout"""
    def res(v, ci, co):
        return 2 * v * (27 * ci * co + 27 * co * co + (ci * co if ci != co else 0))

    def up(v, ci, co):
        return 2 * v * ci * 8 * co
    v1, v2, v3, v4 = n * 8, n * 64, n * 512, n * 4096            # voxels after 1..4 upsamplings
    f = res(v4, in_chans, fs)                                                                     # encoder1 (full resolution)
    f += up(n, D, 2 * fs) + up(v1, 2 * fs, 2 * fs) + res(v2, 2 * fs, 2 * fs) + up(v2, 2 * fs, 2 * fs) + res(v3, 2 * fs, 2 * fs)   # encoder2
    f += up(n, D, 4 * fs) + up(v1, 4 * fs, 4 * fs) + res(v2, 4 * fs, 4 * fs)                      # encoder3
    f += up(n, D, 8 * fs)                                                                         # encoder4
    f += up(n, D, 8 * fs) + res(v1, 16 * fs, 8 * fs)                                              # decoder5
    f += up(v1, 8 * fs, 4 * fs) + res(v2, 8 * fs, 4 * fs)                                         # decoder4
    f += up(v2, 4 * fs, 2 * fs) + res(v3, 4 * fs, 2 * fs)                                         # decoder3
    f += up(v3, 2 * fs, fs) + res(v4, 2 * fs, fs)                                                 # decoder2
    return f + 2 * v4 * fs * classes                                                              # 1x1 output convolution


def source_hash():
    """hash of the kernel sources + C ABI header of the running build: a committed PMC profile counts only for the code it measured"""
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "include", "ucfvit_hip.h")]
    cs = os.path.join(ROOT, "ucf-vit_amd", "csrc")
    files += sorted(os.path.join(cs, f) for f in os.listdir(cs) if f.endswith((".hip", ".h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, dtype, batch, family, profiles_dir=None):
    """HBM bytes per launch of the dominant kernel family from the committed rocprofv3 PMC passes of this same command (separate
    FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 gfx950 correction: tools/pmc_summarize.py) — counters cannot be read from inside the
    process.  A profile is used only if it was taken from THIS build: its recorded `src_hash` must equal the hash of the kernel
    sources now in the tree; otherwise (stale or no profile) the figure is null and `traffic_note` says why."""
    import glob
    want = source_hash()
    note = "no committed PMC profile for this workload"
    for f in sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "*pmc_traffic*.json")), reverse=True):
        try:
            with open(f) as fh:
                d = json.load(fh)
        except Exception:
            continue
        if d.get("workload") != workload or d.get("dtype") != dtype or d.get("per_gpu_batch") != batch:
            continue
        if d.get("src_hash") != want:
            note = f"stale: {os.path.basename(f)} measured source {d.get('src_hash')}, this build is {want}"
            continue
        ent = d.get("families", {}).get(family)
        if ent:
            return round(ent["hbm_bytes_per_launch_corrected"]), os.path.basename(f)
    return None, note


# --------------------------------------------------------------------------------------------------------------- parent: start N ranks
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, poll_s=0.2):
    """`python bench.py --gpus N` without a launcher: start one child per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, like
    torch.distributed.run does) BEFORE this process has touched the GPU, relay rank 0's JSON line.  The parent never initialises HIP and
    never replaces itself (children are ordinary, freshly started subprocesses).  Every child is polled: the first one that exits non-zero
    (out of memory, RCCL initialisation failure, GPU fault) gets its siblings terminated — they would otherwise sit in a collective until
    the process-group timeout, holding their GPUs — and the parent exits non-zero naming (rank, exit code)."""
    import threading
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))   # stderr of every rank is inherited
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)    # rank 0's pipe must be drained while we poll
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc is not None and rc != 0:
                failed = (r, rc)
                break
        else:
            time.sleep(poll_s)
    if failed is None:
        failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:                       # exactly the children started above, by handle (never by pattern)
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        raise SystemExit(f"bench.py: rank {failed[0]} exited with code {failed[1]}; the other ranks were stopped "
                         f"(exit codes {[p.returncode for p in procs]})")
    reader.join(timeout=10.0)
    if out0 and out0[0]:
        sys.stdout.write(out0[0])
        sys.stdout.flush()


# --------------------------------------------------------------------------------------------------------------- in-process profiler
class KernelProfiler:
    """HIP-event timing of the launches of the two MFMA kernel families in the timed region (events on the launch stream = torch's
    current stream): `gemm` = every ucfvit_gemm / ucfvit_gemm_grouped call, `attention` = every ucfvit_attention_fwd / _bwd call."""

    def __init__(self):
        self.records = {"gemm": [], "attention": [], "conv": []}   # (start_evt, end_evt, algorithmic flops)
        self.enabled = False

    def _timed(self, family, fn, flops_of):
        import torch
        prof = self

        def wrapper(*a, **kw):
            if not prof.enabled:
                return fn(*a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(*a, **kw)
            e.record()
            prof.records[family].append((s, e, flops_of(*a, **kw)))
            return out
        return wrapper

    def install(self):
        from UCF_VIT._hip import ops
        ops.gemm = self._timed("gemm", ops.gemm, lambda A, B, M, N, K, *a, **kw: 2.0 * M * N * K)
        ops.wgrad_grouped = self._timed("gemm", ops.wgrad_grouped,
                                        lambda items: sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _, _ in items))
        # algorithmic attention FLOPs: forward 4 B H N^2 dh (QK^T and PV), backward twice that (train = 3 x forward);
        # the recompute of S and dP inside the backward kernels is not credited
        ops.attention_fwd = self._timed("attention", ops.attention_fwd, lambda qkv, B, N, H, dh, scale: 4.0 * B * H * N * N * dh)
        ops.attention_bwd = self._timed("attention", ops.attention_bwd,
                                        lambda qkv, out, dout, lse, B, N, H, dh, scale, **kw: 8.0 * B * H * N * N * dh)

        # 3x3x3 and 1x1x1 convolutions (csrc/conv3d.hip): 2 x taps x voxels x Cin x Cout per forward / data-gradient / weight-gradient
        # launch, with the operand widths as launched (the zero channels of the 8-channel input operand are 1 % of the total)
        def conv_flops(x, w_packed, cout, ksize=3, bias=None, cout_store=None, out_dtype=None, accumulate_into=None, stats_eps=None):
            return 2.0 * ksize ** 3 * x.numel() * (cout_store or cout)
        ops.conv3d_fwd = self._timed("conv", ops.conv3d_fwd, conv_flops)
        ops.conv3d_wgrad = self._timed("conv", ops.conv3d_wgrad, lambda x, dy, ksize=3: 2.0 * ksize ** 3 * x.numel() * dy.shape[-1])

    def summary(self, family):
        tot_ms, tot_flops = 0.0, 0.0
        for s, e, f in self.records[family]:
            tot_ms += s.elapsed_time(e)
            tot_flops += f
        return len(self.records[family]), tot_ms, tot_flops


# --------------------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(wname, w, steps=5):
    """reference training loop (train_class_simple.py:344-357 / train_masked_simple.py:35-49) restated on CPU fp32: the oracle, timed
    on the host cores; 3 warm-up steps (fewer only when a step is so long that the leg would exceed ~40 s) + 5 timed steps"""
    import torch
    from oracle import ucf_vit_ref as R
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))      # the GPU box grants a 16-CPU share per GPU; more threads only oversubscribe it
    torch.set_num_threads(cores)
    if w.get("adaptive"):
        return None          # the headline baseline is the image-input loop; the adaptive oracle is timed by nothing
    g = torch.Generator().manual_seed(0)
    if w["kind"] == "vit":
        batch, unit = 8, "images/sec"
        m = R.VIT([w["img"], w["img"]], patch_size=w["patch"], in_chans=3, num_classes=w["classes"], embed_dim=w["dim"], depth=w["depth"],
                  num_heads=w["heads"], sdpa=True)
        x = torch.randint(0, 256, (batch, 3, w["img"], w["img"]), generator=g).float()
        y = torch.randint(0, w["classes"], (batch,), generator=g)
        opt = R.configure_optimizer(m, 1e-4, 0.9, 0.95, 1e-5)
        sch = R.WarmupCosineLR(opt, 1000, 20000, 1e-8, 1e-8)
        what = f"{wname} fp32 CPU oracle train step, batch {batch}"

        def one():
            R.train_step_class(m, opt, sch, x, y)
    elif w["kind"] == "mae":
        batch, unit = 8, "images/sec"
        m = R.MAE([w["img"], w["img"]], patch_size=w["patch"], in_chans=3, num_classes=None, embed_dim=w["dim"], depth=w["depth"],
                  num_heads=w["heads"], class_token=False, mask_ratio=w["mask_ratio"], decoder_depth=w["dec_depth"],
                  decoder_embed_dim=w["dec_dim"], decoder_num_heads=w["dec_heads"], sdpa=True)
        x = torch.rand(batch, 3, w["img"], w["img"], generator=g)
        opt = R.configure_optimizer(m, 1e-4, 0.9, 0.95, 0.05)
        sch = R.WarmupCosineLR(opt, 1000, 20000, 1e-8, 1e-8)
        what = f"{wname} fp32 CPU oracle MAE train step (MSE over all patches), batch {batch}"

        def one():
            R.train_step_mae(m, opt, sch, x)
    else:
        # bounded sample of the volume workload: ONE 128 x 128 x 128 crop (512 tokens) of the same encoder and objective; the 8192-token
        # volume would take minutes per step on the host (attention cost grows with N^2, so the per-volume rate is extrapolated by
        # FLOPs in `sample`, not by voxels)
        batch, unit = 1, "volumes/sec"
        crop = (128, 128, 128)
        m = R.VIT(list(crop), patch_size=w["patch"], in_chans=1, num_classes=None, embed_dim=w["dim"], depth=w["depth"], num_heads=w["heads"],
                  class_token=False, twoD=False, sdpa=True)
        x = torch.rand(batch, 1, *crop, generator=g)
        opt = R.configure_optimizer(m, 1e-4, 0.9, 0.95, 1e-5)
        sch = R.WarmupCosineLR(opt, 1000, 20000, 1e-8, 1e-8)
        taps_at = [(i + 1) * (w["depth"] // 4) for i in range(3)]
        crop_w = dict(w, vol=crop)
        scale = train_flops_per_unit(crop_w) / train_flops_per_unit(w)
        what = (f"{wname} fp32 CPU oracle encoder step on a 128^3 crop (512 tokens), scaled to whole volumes by algorithmic FLOPs "
                f"(x{scale:.5f})")

        if w.get("decoder"):
            # whole model: the decoder restated in oracle/unetr_decoder_ref.py over parameters of the shapes the model declares
            from oracle import unetr_decoder_ref as D
            from UCF_VIT.simple.arch import UNETR
            shapes = UNETR(img_size=list(crop), patch_size=w["patch"], in_chans=1, embed_dim=w["dim"], depth=1, num_heads=w["heads"], class_token=False,
                           twoD=False, num_classes=w["classes"], linear_decoder=False, feature_size=w["fs"], skip_connection=True).state_dict()
            dec = {k: torch.nn.Parameter(v.detach().float().clone()) for k, v in shapes.items() if k.startswith(("encoder", "decoder", "out."))}
            opt = torch.optim.AdamW(list(m.parameters()) + list(dec.values()), lr=1e-4, betas=(0.9, 0.95), weight_decay=1e-5)
            sch = R.WarmupCosineLR(opt, 1000, 20000, 1e-8, 1e-8)
            labels = torch.randint(0, w["classes"], (batch, *crop), generator=g)
            fz = tuple(c // w["patch"] for c in crop)
            what = (f"{wname} fp32 CPU oracle whole-model step (encoder + conv decoder + Dice/CE) on a 128^3 crop, scaled to whole volumes by "
                    f"algorithmic FLOPs (x{scale:.5f})")

            def one():
                feats, taps = R.vit_forward_intermediates(m, x, taps_at)
                loss = D.dice_ce_loss(D.unetr_head(dec, x, feats, taps, fz, w["dim"]), labels)
                loss.backward()
                opt.step()
                opt.zero_grad()
                sch.step()
        else:
            def one():
                feats, taps = R.vit_forward_intermediates(m, x, taps_at)
                loss = feats.square().mean() + sum(t.square().mean() for t in taps)
                loss.backward()
                opt.step()
                opt.zero_grad()
                sch.step()
    t0 = time.perf_counter()
    one()     # first warm-up step: also sizes the leg
    warm = time.perf_counter() - t0
    print(f"[bench] cpu_baseline warm-up step {warm:.1f} s on {cores} threads", file=sys.stderr, flush=True)
    n_warm = 1
    while n_warm < 3 and (n_warm + 1 + steps) * warm <= 40.0:    # >= 3 warm-up steps — whenever they and the timed steps fit ~30-40 s
        one()
        n_warm += 1
    steps = max(1, min(steps, int(30.0 / max(warm, 1e-3))))     # keep the whole leg to ~10-30 s of CPU work
    t0 = time.perf_counter()
    for i in range(steps):
        one()
        print(f"[bench] cpu_baseline step {i + 1}/{steps}", file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    value = batch * steps / dt
    if w["kind"] == "unetr":
        value *= scale
    return {"value": value, "unit": unit, "cores": cores, "kind": "port", "sample": f"{what}, {steps} timed steps after {n_warm} warm-up step(s) ({dt:.1f} s)"}


# --------------------------------------------------------------------------------------------------------------- workloads
def setup_parallel_groups(args, w, world, rank):
    """--sp / --tp: the process groups of the reference's Hybrid-OP layout (utils/misc.init_par_groups = reference utils/misc.py:129-238;
    the entry scripts build them at training_scripts/train_masked_fsdp.py:219-223) and, for sequence parallelism, the 2-D Ulysses x ring
    view of every sequence-parallel group.  Collective over the whole world: every rank calls it."""
    sp, tp = args.sp, args.tp
    if sp == 1 and tp == 1:
        return None
    from UCF_VIT.utils.misc import init_par_groups
    dp = world // (sp * tp)
    seq_g, ddp_g, tp_g, _, _, _ = init_par_groups(rank, dp, tp, sp, dp, 1)
    par = dict(sp=sp, tp=tp, dp=dp, dp_index=rank // (sp * tp), ddp_group=ddp_g, tp_group=tp_g, spg=None, grid="")
    if sp > 1:
        from UCF_VIT.fsdp.seq_parallel import make_seq_parallel_groups
        lists = [[i * tp * sp + s_ * tp + t for s_ in range(sp)] for i in range(dp) for t in range(tp)]      # init_par_groups' "sp" lists
        par["spg"] = make_seq_parallel_groups(lists, w["heads"], ulysses_size=args.ulysses or None)
        par["grid"] = f"({par['spg'].pr}x{par['spg'].pu})"
    return par


def build_workload(args, w, dev, rank, par=None):
    """-> (net-able model, step closure factory inputs): returns (model, make_step) where make_step(net, opt, sch) -> step()"""
    import torch
    from UCF_VIT.utils.fused_attn import FusedAttn
    B = args.batch or w["batch"]
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)      # per-rank synthetic shard
    variables = ["red", "green", "blue"]
    if w["kind"] == "vit":
        from UCF_VIT.simple.arch import VIT
        from UCF_VIT.utils.metrics import cross_entropy_loss
        adaptive = w.get("adaptive", 0)
        akw = dict(adaptive_patching=True, fixed_length=adaptive, use_adaptive_pos_emb=True) if adaptive else {}
        model = VIT(img_size=[w["img"], w["img"]], patch_size=w["patch"], in_chans=3, num_classes=w["classes"], embed_dim=w["dim"],
                    depth=w["depth"], num_heads=w["heads"], mlp_ratio=4.0, FusedAttn_option=FusedAttn.HIP, **akw).to(dev)
        seq_ps = None
        if adaptive:     # pre-cut, resized patches as the reference's quadtree dataloader emits them, (size, x, y) per token
            x = torch.randint(0, 256, (B, 3, adaptive, w["patch"] ** 2), generator=g).float().to(dev)
            seq_ps = torch.cat([2.0 ** torch.randint(2, 7, (B, adaptive, 1), generator=g).float(),
                                torch.randint(0, w["img"], (B, adaptive, 2), generator=g).float()], dim=-1).to(dev)
        else:
            x = torch.randint(0, 256, (B, 3, w["img"], w["img"]), generator=g).float().to(dev)   # un-normalised pixels, resident in HBM
        y = torch.randint(0, w["classes"], (B,), generator=g).to(dev)

        def make_step(net, opt, sch):
            def step():
                loss = cross_entropy_loss(net(x, variables, seq_ps), y)
                loss.backward()
                opt.step()
                opt.zero_grad()
                sch.step()
                return loss
            return step
        desc = "train step (fwd+bwd+AdamW), synthetic U{0..255} images resident in HBM"
        return model, make_step, B, desc, 1e-5
    if w["kind"] == "mae":
        from UCF_VIT.simple.arch import MAE
        from UCF_VIT.utils.metrics import patch_mse_loss
        model = MAE(img_size=[w["img"], w["img"]], patch_size=w["patch"], in_chans=3, embed_dim=w["dim"], depth=w["depth"],
                    num_heads=w["heads"], class_token=False, weight_init='skip', mask_ratio=w["mask_ratio"], linear_decoder=False,
                    decoder_depth=w["dec_depth"], decoder_embed_dim=w["dec_dim"], decoder_num_heads=w["dec_heads"], mlp_ratio_decoder=4.0,
                    FusedAttn_option=FusedAttn.HIP).to(dev)
        x = torch.rand(B, 3, w["img"], w["img"], generator=g).to(dev)         # min-max normalised images (dataloaders/dataset.py:76)
        L = (w["img"] // w["patch"]) ** 2
        noise = torch.rand(B, L, generator=g).to(dev)                          # fresh mask noise would be torch.rand on the device: same cost

        def make_step(net, opt, sch):
            def step():
                pred, mask = net(x, variables, None, noise=noise)
                loss = patch_mse_loss(pred, x, w["patch"])                     # loss_fn "MSE": over ALL patches (base_config.yaml:29)
                loss.backward()
                opt.step()
                opt.zero_grad()
                sch.step()
                return loss
            return step
        desc = "MAE train step (mask 0.75 + gather, 24-block encoder on 49 tokens, 8 x 512 decoder on 196, MSE; fwd+bwd+AdamW), synthetic images resident in HBM"
        return model, make_step, B, desc, 0.05
    vol = list(w["vol"])
    ukw = dict(img_size=vol, patch_size=w["patch"], in_chans=1, embed_dim=w["dim"], depth=w["depth"], num_heads=w["heads"],
               class_token=False, twoD=False, num_classes=4, linear_decoder=False, feature_size=16, skip_connection=True,
               FusedAttn_option=FusedAttn.HIP)
    if par is None:
        from UCF_VIT.simple.arch import UNETR
        model = UNETR(**ukw)
    else:
        from UCF_VIT.fsdp.arch import UNETR
        model = UNETR(seq_par_size=par["sp"], seq_par_group=par["spg"], tensor_par_size=par["tp"], tensor_par_group=par["tp_group"], **ukw)
        # every rank of a sequence- / tensor-parallel group works on the SAME volumes: the data stream is seeded per data-parallel replica
        g = torch.Generator().manual_seed(1234 + par["dp_index"])
    if w.get("decoder"):
        from UCF_VIT._hip import functional as HF
        model = model.to(dev)
        if not model.hip_decoder():
            raise SystemExit("bench.py: the UNETR decoder of this configuration is not on the HIP convolution kernels")
        x = torch.rand(B, 1, *vol, generator=g).to(dev)
        labels = torch.randint(0, 4, (B, *vol), generator=g).to(dev)          # segmentation classes of every voxel, resident in HBM

        sharded = par is not None and par["sp"] > 1 and model.shard_decoder()
        if sharded:
            # the decoder runs on X-slabs (fsdp/sharded_decoder.py): every rank holds its slab of the logits and of the labels
            from UCF_VIT.fsdp import sharded_decoder as SD
            labels_l = SD.local_slab(labels, par["spg"], 1).contiguous()

        def make_step(net, opt, sch):
            def step():
                if sharded:
                    loss = SD.sharded_dice_ce(net(x, None), labels_l, par["spg"])
                else:
                    loss = HF.dice_ce(net(x, None), labels)                   # DiceCELoss(to_onehot_y, softmax, squared_pred), train_unetr_simple.py:38
                loss.backward()
                opt.step()
                opt.zero_grad()
                sch.step()
                return loss
            return step
        desc = ("UNETR train step (12-Block encoder on 8192 tokens with taps 3/6/9, skip-connection conv decoder feature_size 16 up to 512x512x128, "
                "Dice+CE on 4 classes; fwd+bwd+AdamW), synthetic volumes and labels resident in HBM")
        if sharded:
            desc += "; encoder token shards AND decoder X-slabs sharded over the sequence-parallel group (halo exchange per 3x3x3 layer)"
        return model, make_step, B, desc, 1e-5
    for n_, p_ in model.named_parameters():        # encoder workload: the conv decoder takes no part
        if not n_.startswith(("blocks.", "patch_embed.", "token_embeds.", "norm.", "pos_embed")):
            p_.requires_grad_(False)
    model = model.to(dev)
    x = torch.rand(B, 1, *vol, generator=g).to(dev)                            # basic_ct volumes are min-max normalised
    taps_at = model.skip_indices

    def make_step(net, opt, sch):
        mod = net.module if hasattr(net, "module") else net

        def step():
            feats, taps = mod.forward_intermediates(x, None, None, indices=taps_at)
            # synthetic objective on what the decoder consumes (final features + the three taps): four tiny element-wise reductions
            loss = feats.float().square().mean() + sum(t.float().square().mean() for t in taps)
            loss.backward()
            opt.step()
            opt.zero_grad()
            sch.step()
            return loss
        return step
    desc = "UNETR encoder train step (3-D patch embedding, 12 Blocks on 8192 tokens, taps 3/6/9; fwd+bwd+AdamW), synthetic volumes resident in HBM"
    return model, make_step, B, desc, 1e-5


def dry_run(args, world, rank):
    """UCFVIT_BENCH_DRY=1 (tests on a box without a GPU): everything of the N-rank protocol except the GPU work — rendezvous over gloo,
    the barrier-bracketed timed region, MAX over ranks, one JSON line from rank 0.  The line is marked "dry_run" and has no value."""
    import torch
    import torch.distributed as dist
    if os.environ.get("UCFVIT_BENCH_DRY_FAIL_RANK") == str(rank):      # rehearsal of a rank that dies before the rendezvous
        raise SystemExit(3)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (rank + 1))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"dry_run": True, "metric": "none (launcher rehearsal)", "value": None, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * dt / max(args.steps, 1), 3), "workload": args.workload}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="vit_l16_224", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: workload default)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=5)
    # BASELINE config 5 ("hybrid-op sequence-parallel across 8 GPUs"): --gpus N = dp x sp (or dp x tp) ranks laid out like the reference's
    # init_par_groups (utils/misc.py:129-238): rank = dp_index * (tp * sp) + sp_index * tp + tp_index.  UNETR workloads only.
    ap.add_argument("--sp", type=int, default=1, help="sequence-parallel group size (UNETR workloads; the token sequence is sharded)")
    ap.add_argument("--ulysses", type=int, default=0, help="all-to-all factor P_u of the 2-D Ulysses x ring layout (default gcd(heads, sp))")
    ap.add_argument("--tp", type=int, default=1, help="tensor-parallel (Hybrid-OP head / hidden shard) group size (UNETR workloads)")
    args = ap.parse_args()
    if args.sp < 1 or args.tp < 1 or (args.sp > 1 and args.tp > 1) or args.gpus % (args.sp * args.tp):
        raise SystemExit("bench.py: --gpus must be a multiple of --sp or --tp (sequence and tensor parallelism are not combined inside one Block)")
    if (args.sp > 1 or args.tp > 1) and WORKLOADS[args.workload]["kind"] != "unetr":
        raise SystemExit("bench.py: --sp / --tp apply to the UNETR workloads (BASELINE config 5)")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:])      # nothing below has run: this process has not touched the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: launch one rank per GPU (or let bench.py start them: no WORLD_SIZE)")

    import torch
    import torch.distributed as dist
    if os.environ.get("UCFVIT_BENCH_DRY"):
        return dry_run(args, world, rank)
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

    from UCF_VIT.utils.misc import configure_optimizer, configure_scheduler
    from UCF_VIT._hip import lib
    lib.load()   # fail loudly if the HIP library is missing

    w = WORKLOADS[args.workload]
    cdtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    par = setup_parallel_groups(args, w, world, rank)          # None, or the Hybrid-OP / sequence-parallel groups of this rank
    model, make_step, B, desc, wd = build_workload(args, w, dev, rank, par)
    model.set_compute_dtype(cdtype)
    net = model
    if world > 1:
        from UCF_VIT._hip.ddp import HipDataParallel
        # sequence parallelism: one MEAN over all dp x sp ranks (seq_parallel.GatherTokensFn carries the factor P); tensor parallelism:
        # every rank owns its weight shards, only the data-parallel replicas of a lane reduce (reference: DDP over ddp_group)
        net = HipDataParallel(model, process_group=par["ddp_group"] if par and par["tp"] > 1 else None)
    opt = configure_optimizer(model, 1e-4, 0.9, 0.95, wd)        # configs/*/base_config.yaml: lr, betas, wd
    sch = configure_scheduler(opt, 1000, 20000, 1e-8, 1e-8)
    step = make_step(net, opt, sch)

    prof = KernelProfiler()
    prof.install()

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
        dp = world // (args.sp * args.tp)                    # data-parallel replicas: a sequence- / tensor-parallel group works on ONE batch
        units = dp * B * args.steps
        value = units / dt
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        fam = {}
        for name in ("gemm", "attention", "conv"):
            n_l, ms, fl = prof.summary(name)
            fam[name] = dict(launches=n_l, ms=ms, flops=fl, tflops=(fl / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0, share=ms * 1e-3 / dt)
        dom = max(fam, key=lambda k: fam[k]["ms"])
        other = max((k for k in fam if k != dom), key=lambda k: fam[k]["ms"])
        kernel_names = {
            "gemm": "gemm3_kernel<%s> / gemm5_kernel 256x256x64 persistent ping-pong MFMA GEMM (ping-pong: weight gradients, residual / "
                    "column-sum epilogues; staggered ping-pong with the epilogue under the partner group's K loop: the other forward / "
                    "data-gradient launches) — every ucfvit_gemm / ucfvit_gemm_grouped launch of the timed region" % args.dtype,
            "attention": "fused attention kernels (attn_fwd / attn_bwd_dq / attn_bwd_dkv streaming for N > 208, attn_s3_fwd / attn_g_bwd "
                         "resident below): every forward and backward launch of the timed region, algorithmic FLOPs 4 / 8 B H N^2 dh",
            "conv": "conv3_fwd_kernel / conv3_wgrad_kernel (csrc/conv3d.hip): 3x3x3 implicit-GEMM convolutions on MFMA, channels-last bf16 — every "
                    "forward, data-gradient and weight-gradient launch of the timed region, algorithmic FLOPs 2 x 27 x voxels x Cin x Cout",
        }
        step_tflops = value / world * train_flops_per_unit(w) / 1e12       # per GPU: the group's volumes spread over its ranks
        traffic, traffic_src = pmc_traffic(args.workload, args.dtype, B, dom)
        d = fam[dom]
        unit = "volumes/sec" if w["kind"] == "unetr" else "images/sec"
        headline = args.workload == "vit_l16_224" and args.dtype == "bf16"
        parallelism = f"dp{dp}"
        if args.sp > 1:
            parallelism = (f"dp{dp} x " if dp > 1 else "") + f"sp{args.sp}{par['grid']}"      # (ring x Ulysses), e.g. sp8(2x4) for 12 heads on 8 GPUs
        elif args.tp > 1:
            parallelism = (f"dp{dp} x " if dp > 1 else "") + f"tp{args.tp}"
        res = {
            "metric": "images/sec/node ViT-L/16 224^2 bf16 train step" if headline else f"{unit.replace('/sec', '')}/sec/node {args.workload} {args.dtype} train step",
            "value": round(value, 3 if w["kind"] == "unetr" else 2), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload} {desc}",
                       "per_gpu_batch": B, "global_batch": B * dp, "parallelism": parallelism, "final_loss": round(final_loss, 4),
                       "per_group_batch_note": None if par is None else "per_gpu_batch is the batch of one sequence- / tensor-parallel group (its ranks share it)",
                       "grad_all_reduce": (str(net.reduce_dtype()).replace("torch.", "") + " mean, " + ("RCCL" if backend == "nccl" else backend + " (rehearsal)") + ", overlapped with backward" + (", direct reduce-scatter + all-gather (UCFVIT_DDP_ALGO=direct)" if getattr(net, "algorithm", "") == "direct" else "")) if world > 1 else "none (1 rank)"},
            "roofline": {"bound": "mfma", "kernel": kernel_names[dom],
                         "achieved": round(d["tflops"], 1), "peak": peak, "unit": "TFLOP/s", "frac": round(d["tflops"] / peak, 4),
                         "traffic": traffic, "traffic_note": traffic_src, "launches": d["launches"],
                         "avg_launch_ms": round(d["ms"] / max(d["launches"], 1), 4),
                         "avg_launch_gflop": round(d["flops"] / max(d["launches"], 1) / 1e9, 2),
                         "time_share_of_step": round(d["share"], 3),
                         "other_family": {"kernel": other, "achieved": round(fam[other]["tflops"], 1), "launches": fam[other]["launches"],
                                          "time_share_of_step": round(fam[other]["share"], 3)},
                         "whole_step_tflops_per_gpu": round(step_tflops, 1), "whole_step_frac": round(step_tflops / peak, 4),
                         "src_hash": source_hash()},
        }
        if mfma_stream:
            # context, not the contract's `peak`: what a register-only MFMA loop sustains on this device under its power management
            res["roofline"]["mfma_stream_measured"] = round(mfma_stream, 1)
            res["roofline"]["frac_of_mfma_stream"] = round(d["tflops"] / mfma_stream, 4)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.workload, w, steps=args.cpu_steps)
            if cb is not None:
                res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

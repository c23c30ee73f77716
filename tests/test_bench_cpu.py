"""bench.py on a box without a GPU: the N-rank launcher protocol (UCFVIT_BENCH_DRY rehearsal), the work figures of the workloads against
BASELINE.md §4, and the rule that a committed PMC traffic profile counts only for the kernel sources it was measured on."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout=180):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_starts_two_ranks_by_itself_and_relays_one_json_line():
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns rank 0 and rank 1 (RANK / WORLD_SIZE / MASTER_* set),
    they rendezvous (gloo here), time a barrier-bracketed region, take the MAX over ranks, rank 0 prints ONE JSON line"""
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1"], {"UCFVIT_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["steps"] == 4
    assert d["ms_per_step"] >= 2.0          # rank 1 sleeps 2 ms per step, rank 0 one: the MAX over ranks is reported


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "4"], {"UCFVIT_BENCH_DRY": "1", "WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in r.stderr


def test_parent_reports_a_failed_rank():
    """a rank that dies (here: no GPU in this container and no dry switch) must fail the whole command, not hang or print a line"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0 and "exited with code" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_parent_stops_the_other_ranks_when_one_dies():
    """rank 1 exits before the rendezvous, rank 0 would wait for it in the process-group initialisation (30 minutes by default): the
    parent polls every child, stops rank 0, names the failing (rank, exit code) and returns within seconds, without a JSON line"""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "0"], {"UCFVIT_BENCH_DRY": "1", "UCFVIT_BENCH_DRY_FAIL_RANK": "1"}, timeout=120)
    dt = time.time() - t0
    assert r.returncode != 0 and "rank 1 exited with code 3" in r.stderr, r.stderr[-1500:]
    assert dt < 60, dt
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_work_figures_match_baseline_md():
    """BASELINE.md §4 / SURVEY §8d: 369.4 GF per image (ViT-L), 105.4 (ViT-B), 122.5 (MAE ViT-L r = 0.75), 11.8 TF per volume (UNETR encoder)"""
    import bench
    f = lambda n: bench.train_flops_per_unit(bench.WORKLOADS[n])
    assert abs(f("vit_l16_224") / 1e9 - 369.4) < 0.5
    assert abs(f("vit_b16_224") / 1e9 - 105.4) < 0.3
    assert abs(f("mae_vit_l16_224") / 1e9 - 122.5) < 0.3
    assert abs(f("unetr_enc_512x512x128") / 1e12 - 11.8) < 0.1
    # MAE batch: encoder rows (49 per image) and decoder rows (196 per image) both land just under a multiple of 64 M-tiles of 256 rows, so that
    # every GEMM of the step (4 ... 16 N-tiles) is a whole number of rounds of the 256 CUs
    b = bench.WORKLOADS["mae_vit_l16_224"]["batch"]
    for rows in (49 * b, 196 * b, 197 * bench.WORKLOADS["vit_l16_224"]["batch"], 197 * bench.WORKLOADS["vit_b16_224"]["batch"]):
        tiles = -(-rows // 256)
        assert tiles % 64 == 0 and tiles * 256 - rows < 256
    assert abs(f("unetr_512x512x128") / 1e12 - 23.9) < 0.1                     # whole UNETR: encoder + 4.04 TF forward of decoder convolutions


def test_stale_pmc_profile_is_refused(tmp_path):
    import bench
    rec = {"workload": "vit_l16_224", "dtype": "bf16", "per_gpu_batch": 166, "src_hash": "0" * 16,
           "families": {"gemm": {"hbm_bytes_per_launch_corrected": 123.0}}}
    p = tmp_path / "r99_pmc_traffic_x.json"
    p.write_text(json.dumps(rec))
    val, note = bench.pmc_traffic("vit_l16_224", "bf16", 166, "gemm", profiles_dir=str(tmp_path))
    assert val is None and note.startswith("stale")
    rec["src_hash"] = bench.source_hash()
    p.write_text(json.dumps(rec))
    val, note = bench.pmc_traffic("vit_l16_224", "bf16", 166, "gemm", profiles_dir=str(tmp_path))
    assert val == 123 and note == p.name
    val, note = bench.pmc_traffic("vit_l16_224", "bf16", 128, "gemm", profiles_dir=str(tmp_path))
    assert val is None


def test_committed_pmc_profiles_of_this_round_carry_a_source_hash():
    import glob
    for f in glob.glob(os.path.join(ROOT, "profiles", "r0[2-9]*pmc_traffic*.json")):
        d = json.load(open(f))
        assert len(d.get("src_hash", "")) == 16 and "families" in d, f


def test_wgrad_queue_flushes_the_last_group_in_pairs_under_data_parallel(monkeypatch):
    """the weight-gradient queue groups four Blocks per launch (768 tiles = 3 whole rounds of 256 CUs); with a data-parallel listener the
    last four Blocks of a backward pass go out two by two, so the final all-reduce that nothing can overlap carries two Blocks' gradients
    instead of four.  Policy only: the launches are faked."""
    import torch
    from UCF_VIT._hip import functional as HF
    launches = []
    monkeypatch.setattr(HF.ops, "wgrad_grouped", lambda items: launches.append(len(items)))
    monkeypatch.setattr(HF.WgradQueue, "_arm", lambda self: None)       # (the autograd engine takes callbacks only inside a backward pass)

    class Owner:
        def poke(self):
            pass
    owner = Owner()

    def one_pass(q, blocks):
        for _ in range(blocks):
            q.items += [(None, None, None, False)] * 4          # the four Linear layers of a Block
            q.owners += [None] * 4
            q.tiles += 192
            q.end_block()
        q._end_of_backward()

    q = HF.WgradQueue()
    one_pass(q, 24)
    assert launches == [16] * 6                                   # single GPU: six launches of four Blocks
    launches.clear()
    import weakref
    q.listeners.append(weakref.WeakMethod(owner.poke))
    one_pass(q, 24)                                               # Block count known from the first pass
    assert launches == [16] * 5 + [8, 8]
    launches.clear()
    q2 = HF.WgradQueue()
    q2.listeners.append(weakref.WeakMethod(owner.poke))
    one_pass(q2, 24)
    assert launches == [16] * 6                                   # first pass of a fresh queue: the count is not known yet


def test_unetr_decoder_flops_and_hip_decoder_rule():
    """host logic of the whole-UNETR workload: the decoder's forward FLOPs at 512 x 512 x 128 / feature_size 16 (4.04 TF: 3.85 TF of 3x3x3
    convolutions, the rest transposed / pointwise layers), and the rule that sends a configuration to the HIP convolution kernels"""
    import bench
    from UCF_VIT.simple.unetr_blocks import hip_decoder_supported
    f = bench.unetr_decoder_fwd_flops(8192, 768, 16, 4)
    assert abs(f / 1e12 - 4.035) < 0.01
    # doubling the feature size quadruples the convolution work (channels in x channels out) of all but the input layer
    assert 3.5 < bench.unetr_decoder_fwd_flops(8192, 768, 32, 4) / f < 4.0
    assert hip_decoder_supported(3, 1, 768, 16) and hip_decoder_supported(3, 3, 768, 32) and hip_decoder_supported(3, 8, 96, 64)
    assert not hip_decoder_supported(2, 1, 768, 16)            # 2-D models keep the torch/MIOpen convolutions (+ the fused N C H W norm kernels)
    assert not hip_decoder_supported(3, 9, 768, 16)            # more input channels than the zero-padded 8-channel operand
    assert not hip_decoder_supported(3, 1, 768, 48)            # 48, 96, 192 ... channels: not powers of two (the channels-last norm kernels)
    assert not hip_decoder_supported(3, 1, 768, 16, kernel_size=5)

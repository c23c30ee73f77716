"""bench.py end to end on the GPU box: the self-launched 2-rank run (rehearsal switches: both ranks on GPU 0, gradients staged through
gloo because RCCL refuses two ranks on one device) and the contract keys of every BASELINE workload's line."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
BENCH = os.path.join(ROOT, "bench.py")
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline")


def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _check(d, n_gpus, unit):
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["unit"] == unit and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(d["value"] - d["config"]["global_batch"] * 1000.0 / d["ms_per_step"]) < 0.01 * d["value"]


def test_bench_gpus_2_self_launch_end_to_end():
    d = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "vit_tiny16_256", "--batch", "32"],
               {"UCFVIT_BENCH_BACKEND": "gloo", "UCFVIT_BENCH_ONE_GPU": "1"})
    _check(d, 2, "images/sec")
    assert d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2" and "gloo" in d["config"]["grad_all_reduce"]


def test_bench_config5_sequence_parallel_layout_self_launch():
    """BASELINE config 5's layout from the command line: `--gpus 2 --sp 2` builds fsdp.arch.UNETR over init_par_groups' sequence-parallel
    group (2-D Ulysses x ring view), HipDataParallel takes the mean over the dp x sp ranks; one group works on ONE batch, so the value
    counts the group's volumes once.  (Rehearsal transport: both ranks on GPU 0, gloo.)"""
    d = _bench(["--gpus", "2", "--sp", "2", "--steps", "2", "--warmup", "1", "--workload", "unetr_512x512x128", "--batch", "1", "--no-cpu-baseline"],
               {"UCFVIT_BENCH_BACKEND": "gloo", "UCFVIT_BENCH_ONE_GPU": "1"}, timeout=1200)
    _check(d, 2, "volumes/sec")
    assert d["config"]["parallelism"] == "sp2(1x2)" and d["config"]["global_batch"] == 1 and "X-slabs" in d["config"]["workload"]
    assert 0.3 < d["config"]["final_loss"] < 5.0


def test_bench_config5_tensor_parallel_layout_self_launch():
    d = _bench(["--gpus", "2", "--tp", "2", "--steps", "2", "--warmup", "1", "--workload", "unetr_enc_512x512x128", "--batch", "1", "--no-cpu-baseline"],
               {"UCFVIT_BENCH_BACKEND": "gloo", "UCFVIT_BENCH_ONE_GPU": "1"}, timeout=1200)
    _check(d, 2, "volumes/sec")
    assert d["config"]["parallelism"] == "tp2" and d["config"]["global_batch"] == 1


def test_bench_mae_workload_line():
    d = _bench(["--steps", "2", "--warmup", "1", "--workload", "mae_vit_l16_224", "--batch", "64", "--cpu-steps", "1"])
    _check(d, 1, "images/sec")
    assert "MAE" in d["config"]["workload"] and "gemm3_kernel" in d["roofline"]["kernel"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["unit"] == "images/sec"


def test_bench_unetr_volume_workload_line():
    d = _bench(["--steps", "2", "--warmup", "1", "--workload", "unetr_enc_512x512x128", "--batch", "1", "--cpu-steps", "1"])
    _check(d, 1, "volumes/sec")
    assert "attention" in d["roofline"]["kernel"] or "gemm3_kernel" in d["roofline"]["kernel"]
    assert d["roofline"]["other_family"]["launches"] > 0
    assert d["cpu_baseline"]["unit"] == "volumes/sec" and "crop" in d["cpu_baseline"]["sample"]


def test_bench_unetr_whole_model_line():
    """configs[4] complete (encoder + conv decoder + Dice/CE): the dominant family is the convolution kernels; no CPU leg here (the whole-model
    oracle step takes ~10 s per step on the host; tests/test_unetr_decoder_model.py checks that oracle against the HIP model)"""
    d = _bench(["--steps", "2", "--warmup", "1", "--workload", "unetr_512x512x128", "--batch", "1", "--no-cpu-baseline"])
    _check(d, 1, "volumes/sec")
    assert "conv" in d["roofline"]["kernel"] and d["roofline"]["launches"] > 40
    assert "decoder" in d["config"]["workload"]

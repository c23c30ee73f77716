"""The C-ABI library loads and exports exactly the symbols include/ucfvit_hip.h declares (no compute, no GPU)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "ucfvit_hip.h")


def header_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ucfvit_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_entry_points():
    syms = header_symbols()
    for must in ("ucfvit_gemm", "ucfvit_layernorm_fwd", "ucfvit_layernorm_bwd", "ucfvit_attention_fwd", "ucfvit_attention_bwd",
                 "ucfvit_im2col", "ucfvit_mae_mask", "ucfvit_gather_rows", "ucfvit_adamw", "ucfvit_last_error"):
        assert must in syms


def test_binding_covers_header():
    from UCF_VIT._hip import lib
    assert sorted(lib.SIGNATURES) == header_symbols()


def test_library_loads_and_exports_every_symbol():
    from UCF_VIT._hip import lib
    assert os.path.exists(lib.LIB_PATH), "build libucfvit_hip.so first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = lib.load()
    assert handle.ucfvit_abi_version() == lib.ABI_VERSION
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (ucfvit_[a-z0-9_]+)", out))
    assert set(header_symbols()) <= exported


def test_no_torch_types_in_abi():
    txt = open(HEADER).read()
    assert "torch" not in txt.lower().replace("pytorch", "") or "at::" not in txt
    assert "at::Tensor" not in txt and "c10::" not in txt


def test_error_reporting_without_gpu():
    """argument validation happens before any HIP call, so it works on a box without a GPU"""
    import ctypes
    from UCF_VIT._hip import lib
    L = lib.load()
    rc = L.ucfvit_gemm(None, None)
    assert rc == -1 and b"null descriptor" in L.ucfvit_last_error()
    rc = L.ucfvit_attention_fwd(1, 1, 1, 1, 1, 1, 48, ctypes.c_float(1.0), 0, None)
    assert rc == -1 and b"head dim" in L.ucfvit_last_error()


def test_committed_bench_line_has_the_contract_keys():
    """the newest committed bench line of the headline workload (profiles/r*_bench_vitl16_b<batch>.json, written by `python bench.py` on an MI355X)
    carries every key of the driver's contract plus the roofline / cpu_baseline objects, with consistent arithmetic; from round 2 on its
    roofline.traffic is a measured number (a PMC profile of the same source hash was in the tree when the line was produced)"""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench_vitl16_b[0-9]*.json")))
    assert files, "no committed bench line"
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    if not os.path.basename(files[-1]).startswith("r01_"):
        assert isinstance(d["roofline"]["traffic"], int) and d["roofline"]["traffic"] > 0
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "images/sec" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(d["value"] - d["config"]["global_batch"] * 1000.0 / d["ms_per_step"]) < 0.01 * d["value"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference")

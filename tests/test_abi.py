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

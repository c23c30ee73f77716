"""Arity of every ctypes signature equals the number of parameters in the header declaration (CPU)."""
import os
import re

from conftest import ROOT


def test_signature_arity_matches_header():
    from UCF_VIT._hip import lib
    txt = open(os.path.join(ROOT, "include", "ucfvit_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for m in re.finditer(r"\b(ucfvit_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        name, params = m.group(1), m.group(2).strip()
        n = 0 if params in ("", "void") else len(params.split(","))
        assert len(lib.SIGNATURES[name][1]) == n, name

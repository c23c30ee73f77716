import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ucf-vit_amd"), ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def rel_err(a, b):
    """max |a-b| / max|b| — the '1e-3 rel fp32' metric of BASELINE.json, robust to entries near zero"""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    denom = b.abs().max().clamp_min(1e-30)
    return ((a - b).abs().max() / denom).item()

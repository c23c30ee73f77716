"""Deterministic, platform-independent weights/inputs for the parity tests (numpy PCG64, no torch RNG involved).
Shared by make_golden.py (which loads them into the reference models) and the tests (which load the same values
into the oracle and the HIP modules)."""
import numpy as np
import torch


def det_tensor(shape, seed, scale=1.0, offset=0.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((rng.standard_normal(tuple(shape)) * scale + offset).astype(np.float32))


def det_state_dict(model, seed, keep=("pos_embed", "decoder_pos_embed")):
    """New values for every entry of model.state_dict(): weights N(0, 0.02..0.05), norm weights ~1, biases small;
    entries whose name is in `keep` (the sincos tables) stay as initialised.  Aliased entries get identical values."""
    out = {}
    by_ptr = {}
    for i, (k, v) in enumerate(model.state_dict().items()):
        if k.split(".")[-1] in keep or k in keep:
            out[k] = v.detach().clone()
            continue
        ptr = v.data_ptr()
        if ptr in by_ptr:
            out[k] = out[by_ptr[ptr]].clone()
            continue
        by_ptr[ptr] = k
        if "norm" in k and k.endswith("weight"):
            t = det_tensor(v.shape, seed * 1000 + i, 0.1, 1.0)
        elif k.endswith("bias"):
            t = det_tensor(v.shape, seed * 1000 + i, 0.02)
        else:
            t = det_tensor(v.shape, seed * 1000 + i, 0.05)
        out[k] = t
    return out


def proj_vector(shape, seed):
    """fixed random direction used to compress a large gradient into one scalar <grad, r>"""
    return det_tensor(shape, 777000 + seed, 1.0)

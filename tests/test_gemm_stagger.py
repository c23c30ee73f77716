"""The staggered ping-pong GEMM (csrc/gemm_stagger.hip: the epilogue of one wave group hidden under the partner group's K loop) against fp32
products of the same bf16 operands, for every epilogue it implements and the shapes that stress its stream bookkeeping: ragged M, a
ragged last N tile, tile counts below / above / not a multiple of the 256 workgroups, short K loops (fewer epilogue steps), long K loops.
By default the library takes that kernel only for K >= 1024 (where it is faster): test_stagger_forced_variants re-runs this file in
subprocesses with UCFVIT_GEMM_STAGGER = 1 / 2 / 4, which forces it — with that many epilogue steps — for every shape it can run.
Reference call sites of these launches: /root/reference/src/UCF_VIT/simple/building_blocks.py:115-128,150-159,189-191."""
import os
import subprocess
import sys

import pytest
import torch

from UCF_VIT._hip import ops
from UCF_VIT._hip.lib import ACT_GELU_SAVE_DERIV

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _rnd(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).bfloat16().to(DEV)


def _rel(a, ref):
    return ((a.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def _gelu_ref(h):
    cdf = 0.5 * (1.0 + torch.erf(h * 0.7071067811865476))
    pdf = 0.3989422804014327 * torch.exp(-0.5 * h * h)
    return h * cdf, cdf + h * pdf


# (M, N, K): tiles = ceil(M/256) * ceil(N/256) must reach 192 for the 256x256 persistent kernel
SHAPES = [
    (32702, 3072, 1024),    # ViT-L qkv, B = 166: ragged M (190 rows in the last tile), 1536 tiles = 6 whole rounds
    (32702, 1024, 4096),    # fc2: 64 K-steps per tile
    (12608, 1024, 1024),    # 200 tiles < 256 workgroups: one tile each
    (19300, 1024, 512),     # 304 tiles: 48 workgroups own two tiles, the rest one; 8 K-steps
    (16384, 1000, 256),     # ragged last N tile (232 columns), 4 K-steps
    (13000, 1024, 128),     # 2 K-steps: one epilogue step
    (12544, 2304, 768),     # ViT-B qkv shape: 9 N-tiles, 12 K-steps
]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_stagger_plain_and_residual(M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b, res = _rnd(g, M, K), _rnd(g, N, K, scale=0.05), _rnd(g, N), _rnd(g, M, N)
    ref = x.float() @ w.float().T
    y = ops.linear_fwd(x, w, b)
    assert _rel(y, ref + b.float()) < 1e-2
    y0 = ops.linear_fwd(x, w, None)
    assert _rel(y0, ref) < 1e-2
    yr = ops.linear_fwd(x, w, b, residual=res)
    assert _rel(yr, ref + b.float() + res.float()) < 1e-2
    # bitwise reproducible (fixed accumulation order, no atomics)
    assert torch.equal(yr, ops.linear_fwd(x, w, b, residual=res))
    assert torch.equal(y, ops.linear_fwd(x, w, b))


@pytest.mark.parametrize("M,N,K", [(32702, 4096, 1024), (12608, 1024, 1024), (16384, 1000, 256), (13000, 1024, 128)])
def test_stagger_gelu_save_deriv(M, N, K):
    g = torch.Generator().manual_seed(7 * M + N + K)
    x, w, b = _rnd(g, M, K), _rnd(g, N, K, scale=0.05), _rnd(g, N)
    aux = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    y = ops.linear_fwd(x, w, b, act=ACT_GELU_SAVE_DERIV, aux_out=aux)
    h = x.float() @ w.float().T + b.float()
    gr, dr = _gelu_ref(h)
    assert _rel(y, gr) < 1e-2
    assert _rel(aux, dr) < 1e-2
    aux2 = torch.empty_like(aux)
    assert torch.equal(y, ops.linear_fwd(x, w, b, act=ACT_GELU_SAVE_DERIV, aux_out=aux2)) and torch.equal(aux, aux2)


@pytest.mark.parametrize("M,N,K", [(32702, 4096, 1024), (12608, 1024, 1024), (19300, 1024, 512), (16384, 1000, 256)])
def test_stagger_mul_aux_and_column_sums(M, N, K):
    g = torch.Generator().manual_seed(3 * M + N + K)
    dy, wT, gp = _rnd(g, M, K), _rnd(g, N, K, scale=0.05), _rnd(g, M, N)
    cs = torch.full((N,), float("nan"), dtype=torch.float32, device=DEV)
    y = ops.linear_dgrad_t(dy, wT, act_grad_aux=gp, aux_is_deriv=True, c_colsum=cs)
    ref = (dy.float() @ wT.float().T) * gp.float()
    assert _rel(y, ref) < 1e-2
    # the column sums are taken in fp32 before the rounding of the output
    assert _rel(cs, ref.sum(0)) < 2e-3
    # without the column sums the launch may take another kernel (another fixed accumulation order): same values to rounding, and each
    # variant bitwise reproducible
    y2 = ops.linear_dgrad_t(dy, wT, act_grad_aux=gp, aux_is_deriv=True)
    assert _rel(y2, ref) < 1e-2
    assert torch.equal(y2, ops.linear_dgrad_t(dy, wT, act_grad_aux=gp, aux_is_deriv=True))
    cs2 = torch.empty_like(cs)
    y3 = ops.linear_dgrad_t(dy, wT, act_grad_aux=gp, aux_is_deriv=True, c_colsum=cs2)
    assert torch.equal(cs, cs2) and torch.equal(y, y3)


def test_stagger_strided_output_and_inputs():
    """row-padded C / residual (leading dimensions larger than N): the byte offsets of the buffer stores follow ldc, not N"""
    M, N, K = 12608, 1024, 1024
    g = torch.Generator().manual_seed(5)
    x, w, b = _rnd(g, M, K), _rnd(g, N, K, scale=0.05), _rnd(g, N)
    res_full = _rnd(g, M, N + 64)
    out_full = torch.zeros(M, N + 128, dtype=torch.bfloat16, device=DEV)
    out = out_full[:, :N]
    ops.linear_fwd(x, w, b, residual=res_full[:, :N], out=out)
    ref = x.float() @ w.float().T + b.float() + res_full[:, :N].float()
    assert _rel(out, ref) < 1e-2
    assert (out_full[:, N:] == 0).all()            # nothing written beyond the N columns of a row


def test_stagger_random_shapes():
    """seeded random shapes (M ragged, N any multiple of 8, K any multiple of 64, tile counts around the 192-tile threshold of the 256x256
    kernels and around multiples of the 256 workgroups), every epilogue, against fp32 products of the same bf16 operands"""
    import random
    rng = random.Random(20251005)
    g = torch.Generator().manual_seed(99)
    for _ in range(10):
        tn = rng.choice([1, 2, 3, 4, 5, 9])
        N = tn * 256 - rng.choice([0, 0, 8, 128, 248])
        tm = max(2, rng.choice([192, 200, 256, 300, 520]) // tn + rng.choice([0, 1]))
        M = tm * 256 - rng.randrange(0, 255)
        K = 64 * rng.choice([2, 3, 5, 8, 16, 17, 24, 40])
        x, w, b = _rnd(g, M, K), _rnd(g, N, K, scale=0.05), _rnd(g, N)
        ref = x.float() @ w.float().T + b.float()
        tag = f"M={M} N={N} K={K}"
        assert _rel(ops.linear_fwd(x, w, b), ref) < 1e-2, tag
        res = _rnd(g, M, N)
        assert _rel(ops.linear_fwd(x, w, b, residual=res), ref + res.float()) < 1e-2, tag
        aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        y = ops.linear_fwd(x, w, b, act=ACT_GELU_SAVE_DERIV, aux_out=aux)
        gr, dr = _gelu_ref(ref)
        assert _rel(y, gr) < 1e-2 and _rel(aux, dr) < 1e-2, tag
        yd = ops.linear_dgrad_t(x, w, act_grad_aux=res, aux_is_deriv=True)
        assert _rel(yd, (x.float() @ w.float().T) * res.float()) < 1e-2, tag
        del x, w, b, ref, res, aux, y, yd, gr, dr


@pytest.mark.parametrize("steps", ["1", "2", "4"])
def test_stagger_forced_variants(steps):
    """every test of this file with the staggered kernel forced (also for the short-K shapes the default leaves to the ping-pong kernel) and
    with 1 / 2 / 4 epilogue steps: the switch is read once per process, hence the subprocess"""
    if os.environ.get("UCFVIT_GEMM_STAGGER"):
        pytest.skip("already inside a forced run")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "not forced_variants"],
                       env=dict(os.environ, UCFVIT_GEMM_STAGGER=steps), capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]

"""Size-independent properties of the HIP path at the FULL sizes of the headline workload (ViT-L/16 224^2: N = 197 tokens, D = 1024,
16 heads of 64, MLP 4096), where the CPU oracle would take minutes per operator — at the per-GPU batch of round 1 (166 images -> 32702
token rows) AND at the batch bench.py quotes the headline on (665 images -> 131005 rows, 537 M-element fc1 outputs).
Each property holds exactly (or to one rounding of the storage type) for the reference's arithmetic, whatever the size."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
N, H, DH = 197, 16, 64
D = H * DH
BATCHES = [166, 665]        # 665 = bench.py's per-GPU batch (bench.WORKLOADS["vit_l16_224"]["batch"], checked below)


def test_the_bench_batch_is_covered():
    import bench
    assert bench.WORKLOADS["vit_l16_224"]["batch"] in BATCHES


def _ulp_bf16(x):
    return x.abs().clamp_min(2.0 ** -126) * 2.0 ** -7


@pytest.mark.parametrize("B", BATCHES)
def test_attention_rows_are_convex_combinations_full_size(B):
    M = B * N
    """softmax rows sum to one: with V = const the output is that constant for every query, head and batch element (forward);
    the backward of sum(O) with such a V gives dQ = dK = 0 (the scores do not influence O) and dV = column sums of P = exactly
    the number of queries per (batch, head) when summed over keys."""
    from UCF_VIT._hip import ops
    gen = torch.Generator().manual_seed(1)
    qkv = torch.randn(M, 3 * D, generator=gen).bfloat16().to(DEV)
    qkv.view(B, N, 3, H, DH)[:, :, 2] = 0.75                               # V = 0.75 everywhere (exact in bf16)
    o, lse = ops.attention_fwd(qkv, B, N, H, DH, DH ** -0.5)
    assert float((o.float() - 0.75).abs().max()) <= 0.75 * 2.0 ** -7       # one bf16 rounding of a sum of weights that is 1
    assert torch.isfinite(lse).all()
    do = torch.ones_like(o)
    dqkv = ops.attention_bwd(qkv, o, do, lse, B, N, H, DH, DH ** -0.5).view(B, N, 3, H, DH).float()
    # dP = dO V^T = 64 * 0.75 for every (q, k): dS = P (dP - sum_k P dP) = 0  ->  dQ = dK = 0 up to rounding of P in bf16
    assert float(dqkv[:, :, 0].abs().max()) < 2e-2 and float(dqkv[:, :, 1].abs().max()) < 2e-2
    # dV[k, d] = sum_q P[q, k]; summed over k that is the number of queries
    tot = dqkv[:, :, 2].sum(dim=1)                                          # [B, H, DH]
    assert float((tot - N).abs().max()) < 0.02 * N


@pytest.mark.parametrize("B", BATCHES)
def test_attention_is_equivariant_to_batch_and_head_permutations_full_size(B):
    M = B * N
    """(batch, head) pairs are independent problems: permuting them permutes the outputs bit for bit"""
    from UCF_VIT._hip import ops
    gen = torch.Generator().manual_seed(2)
    qkv = torch.randn(M, 3 * D, generator=gen).bfloat16().to(DEV)
    o1, lse1 = ops.attention_fwd(qkv, B, N, H, DH, DH ** -0.5)
    pb = torch.randperm(B, generator=gen).to(DEV)
    ph = torch.randperm(H, generator=gen).to(DEV)
    q2 = qkv.view(B, N, 3, H, DH)[pb][:, :, :, ph].contiguous().view(M, 3 * D)
    o2, lse2 = ops.attention_fwd(q2, B, N, H, DH, DH ** -0.5)
    assert torch.equal(o2.view(B, N, H, DH), o1.view(B, N, H, DH)[pb][:, :, ph])
    assert torch.equal(lse2, lse1[pb][:, ph])


@pytest.mark.parametrize("B", BATCHES)
def test_layernorm_rows_are_standardised_full_size(B):
    M = B * N
    from UCF_VIT._hip import ops
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn(M, D, generator=gen) * 3.0 + 1.5).bfloat16().to(DEV)
    one, zero = torch.ones(D, dtype=torch.bfloat16, device=DEV), torch.zeros(D, dtype=torch.bfloat16, device=DEV)
    y, mean, rstd = ops.layernorm_fwd(x, one, zero, 1e-6)
    yf = y.float()
    assert float(yf.mean(dim=1).abs().max()) < 2e-3                          # bf16 rounding of 1024 standardised values
    assert float((yf.var(dim=1, unbiased=False) - 1.0).abs().max()) < 1e-2
    xf = x.float()
    assert float((mean - xf.mean(dim=1)).abs().max()) < 1e-4
    # backward of sum(y * w) with constant w: the gradient of a standardised row w.r.t. a constant direction vanishes
    dy = torch.full_like(x, 0.5)
    dx, dg, db = ops.layernorm_bwd(dy, x, one, mean, rstd)
    assert float(dx.float().abs().max()) < 2e-2
    assert float((db - 0.5 * M).abs().max()) < 1e-3 * M


@pytest.mark.parametrize("B", BATCHES)
def test_linear_layers_are_linear_full_size(B):
    M = B * N
    """fc1-shaped GEMM (32702 x 4096 x 1024, the 256x256 ping-pong kernel): f(2x) - f(0) = 2 (f(x) - f(0)) up to the bf16 rounding of
    the three outputs, a zero input returns the bias exactly, and the data / weight gradients agree with <dy, f(x)> = <W^T dy, x>."""
    from UCF_VIT._hip import ops
    gen = torch.Generator().manual_seed(4)
    x1 = torch.randn(M, D, generator=gen).bfloat16().to(DEV)
    w = (torch.randn(4 * D, D, generator=gen) * 0.03).bfloat16().to(DEV)
    bias = torch.randn(4 * D, generator=gen).bfloat16().to(DEV)
    x2 = (x1.float() * 2.0).bfloat16()                                       # exact in bf16 (power-of-two scaling)
    y1, y2 = ops.linear_fwd(x1, w, bias), ops.linear_fwd(x2, w, bias)
    y0 = ops.linear_fwd(torch.zeros_like(x1), w, bias)
    err = ((y2.float() - y0.float()) - 2.0 * (y1.float() - y0.float())).abs()      # f(2x) - f(0) = 2 (f(x) - f(0))
    tol = 3 * _ulp_bf16(y0.float()) + 2 * _ulp_bf16(y1.float()) + _ulp_bf16(y2.float()) + 1e-3   # roundings of the three outputs
    assert bool((err <= tol).all()), float((err - tol).max())
    assert torch.equal(y0, bias.unsqueeze(0).expand(M, -1))
    # adjoint identity in fp32: sum(dy * (x W^T)) == sum((dy W) * x) == sum(dW * W)
    dy = torch.randn(M, 4 * D, generator=gen).bfloat16().to(DEV)
    y_nb = ops.linear_fwd(x1, w, None)
    dx = ops.linear_dgrad(dy, w)
    dw = ops.linear_wgrad(dy, x1)
    a = float((dy.double() * y_nb.double()).sum())
    b_ = float((dx.double() * x1.double()).sum())
    c = float((dw.double() * w.double()).sum())
    # the three sums are random walks of 1.3e8 terms: compare them on the scale of the bf16 rounding noise of such a walk
    noise = float((dy.double() * y_nb.double()).pow(2).sum().sqrt()) * 2.0 ** -8
    assert abs(a - b_) < 4 * noise and abs(a - c) < 4 * noise, (a, b_, c, noise)


@pytest.mark.parametrize("Bs", [32, 665])
def test_training_step_is_bitwise_reproducible_full_size(Bs):
    """two identical ViT-L training steps from the same state give bit-identical logits, loss and parameters (no atomics anywhere:
    split-K, grouped weight gradients, LayerNorm / bias reductions and attention gradients are all summed in a fixed order)"""
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    from UCF_VIT.utils.misc import configure_optimizer
    torch.manual_seed(0)
    x = torch.rand(Bs, 3, 224, 224, device=DEV) * 255.0
    y = torch.randint(0, 1000, (Bs,), device=DEV)

    def run():
        torch.manual_seed(7)
        m = VIT(img_size=[224, 224], patch_size=16, in_chans=3, num_classes=1000, embed_dim=1024, depth=24, num_heads=16).to(DEV)
        m.set_compute_dtype(torch.bfloat16)
        opt = configure_optimizer(m, 1e-4, 0.9, 0.95, 1e-5)
        out = m(x, ["red", "green", "blue"])
        loss = cross_entropy_loss(out, y)
        loss.backward()
        opt.step()
        return out.detach().clone(), loss.item(), m._ucf_store.flat_p.detach().clone()

    o1, l1, p1 = run()
    torch.cuda.empty_cache()
    o2, l2, p2 = run()
    assert torch.equal(o1, o2) and l1 == l2 and torch.equal(p1, p2)
    assert torch.isfinite(p1).all()
    assert abs(l1 - 6.9078) < 0.35          # ln 1000 at random initialisation

"""-m gpu tests of the BASELINE.json configurations that round 1 never ran through the HIP path, and of the kernels only they use:

  config 2  ViT-B/16 224^2 bf16 (D 768, 12 heads, N 197): GEMM forward / data gradient / weight gradient at the Block's shapes against an
            fp32 product of the SAME bf16 operands, and a bitwise-reproducible training step;
  config 4  MAE ViT-L/16 mask ratio 0.75 with the 8 x 512 / 16-head decoder (train_masked_simple.py:35-49): one full-size step, the mask
            bit-exact against torch.argsort of the injected noise (arch.py:663-681);
  config 5  UNETR 3-D 512x512x128, p 16 (N = 8192 tokens, D 768, 12 heads): encoder taps (arch.py:995-1086) at full size through
            size-independent properties, and a reduced 128x128x64 volume (N = 512) against the CPU oracle;
  kernel    attn_fwd_kernel<bf16> + attn_bwd_dq / attn_bwd_dkv (streaming attention, every N > 256 and head dim 128): against fp32
            softmax(QK^T / sqrt(dh)) V of the same bf16 inputs (building_blocks.py:175-187).

Reference arithmetic for the comparisons at sizes the CPU oracle cannot finish in seconds is plain fp32 torch math on the GPU of the
box (test infrastructure only; the product path never calls it)."""
import math

import pytest
import torch

from conftest import rel_err
from det_weights import det_state_dict, det_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda"
VARS = ["red", "green", "blue"]
LOG2E = 1.4426950408889634


def _attn_ref(qkv, do, B, N, H, dh):
    """fp32 softmax(QK^T / sqrt(dh)) V and its gradients on the same bf16 inputs, one (batch, head) at a time (N^2 fp32 scores)"""
    x = qkv.float().view(B, N, 3, H, dh)
    o = torch.empty(B, N, H, dh, device=qkv.device)
    d = torch.empty_like(x)
    lse = torch.empty(B, H, N, device=qkv.device)
    dof = do.float().view(B, N, H, dh)
    for b in range(B):
        for h in range(H):
            q, k, v = (x[b, :, i, h].clone().requires_grad_(True) for i in range(3))
            s = (q @ k.T) * dh ** -0.5
            p = torch.softmax(s, dim=-1)
            oh = p @ v
            oh.backward(dof[b, :, h])
            o[b, :, h] = oh.detach()
            lse[b, h] = torch.logsumexp(s.detach(), dim=-1) * LOG2E        # the kernels keep the log-sum-exp in log2
            d[b, :, 0, h], d[b, :, 1, h], d[b, :, 2, h] = q.grad, k.grad, v.grad
    return o.view(B * N, H * dh), d.view(B * N, 3 * H * dh), lse


@pytest.mark.parametrize("N,dh,B,H", [(257, 64, 2, 3), (257, 32, 2, 2), (257, 128, 1, 2), (512, 64, 2, 2), (512, 32, 1, 3), (512, 128, 1, 2),
                                      (1000, 64, 1, 2), (2048, 64, 1, 2), (2048, 32, 1, 2), (2048, 128, 1, 1), (8192, 64, 1, 2), (8192, 32, 1, 1)])
def test_streaming_attention_bf16_vs_fp32_math(N, dh, B, H):
    """every dispatch of the streaming kernels in bf16 (N > 256: UNETR volumes, the vit_tiny16_256 workload; head dim 128 at any N):
    forward, log-sum-exp and all three gradients.  Tolerance 2e-2 of the largest reference magnitude (bf16 probabilities / outputs),
    the same bound the resident kernels are held to."""
    from UCF_VIT._hip import ops
    gen = torch.Generator().manual_seed(N * 131 + dh)
    qkv = torch.randn(B * N, 3 * H * dh, generator=gen).bfloat16().to(DEV)
    do = torch.randn(B * N, H * dh, generator=gen).bfloat16().to(DEV)
    o, lse = ops.attention_fwd(qkv, B, N, H, dh, dh ** -0.5)
    dqkv = ops.attention_bwd(qkv, o, do, lse, B, N, H, dh, dh ** -0.5)
    ref_o, ref_d, ref_lse = _attn_ref(qkv, do, B, N, H, dh)
    assert rel_err(o.float(), ref_o) < 2e-2
    assert float((lse - ref_lse).abs().max()) < 2e-2
    d = dqkv.float().view(B * N, 3, H * dh)
    r = ref_d.view(B * N, 3, H * dh)
    for i, name in enumerate("qkv"):
        assert rel_err(d[:, i], r[:, i]) < 2e-2, f"d{name}"


def test_streaming_attention_forced_max_jump():
    """a rare, data-dependent branch needs its own input: one key whose score towers over the rest arrives in a LATE key tile, so the
    running maximum of the online softmax jumps there and every earlier partial sum must be rescaled"""
    from UCF_VIT._hip import ops
    B, N, H, dh = 1, 1024, 1, 64
    gen = torch.Generator().manual_seed(99)
    x = torch.randn(B * N, 3, H * dh, generator=gen) * 0.5
    x[900, 1] = x[17, 0] * 12.0            # key 900 aligned with query 17: score ~ 12 * |q|^2 / 8
    x[300, 1] = x[650, 0] * -9.0           # and an anti-aligned one
    qkv = x.view(B * N, 3 * H * dh).bfloat16().to(DEV)
    do = torch.randn(B * N, H * dh, generator=gen).bfloat16().to(DEV)
    o, lse = ops.attention_fwd(qkv, B, N, H, dh, dh ** -0.5)
    dqkv = ops.attention_bwd(qkv, o, do, lse, B, N, H, dh, dh ** -0.5)
    ref_o, ref_d, ref_lse = _attn_ref(qkv, do, B, N, H, dh)
    assert torch.isfinite(o.float()).all() and torch.isfinite(dqkv.float()).all()
    assert rel_err(o.float(), ref_o) < 2e-2
    assert float((lse - ref_lse).abs().max()) < 3e-2
    assert rel_err(dqkv.float(), ref_d) < 2e-2


def test_streaming_attention_properties_unetr_full_size():
    """N = 8192, 12 heads of 64 (the UNETR 512x512x128 / p16 encoder, B = 1): softmax rows are convex combinations (V = const gives
    O = const, dQ = dK = 0, sum_k dV = N) and heads are independent problems (a head permutation permutes the outputs bit for bit)"""
    from UCF_VIT._hip import ops
    B, N, H, dh = 1, 8192, 12, 64
    gen = torch.Generator().manual_seed(5)
    qkv = torch.randn(B * N, 3 * H * dh, generator=gen).bfloat16().to(DEV)
    o1, lse1 = ops.attention_fwd(qkv, B, N, H, dh, dh ** -0.5)
    ph = torch.randperm(H, generator=gen).to(DEV)
    q2 = qkv.view(B, N, 3, H, dh)[:, :, :, ph].contiguous().view(B * N, 3 * H * dh)
    o2, lse2 = ops.attention_fwd(q2, B, N, H, dh, dh ** -0.5)
    assert torch.equal(o2.view(B, N, H, dh), o1.view(B, N, H, dh)[:, :, ph])
    assert torch.equal(lse2, lse1[:, ph])
    qkv.view(B, N, 3, H, dh)[:, :, 2] = 0.75
    o, lse = ops.attention_fwd(qkv, B, N, H, dh, dh ** -0.5)
    assert float((o.float() - 0.75).abs().max()) <= 2 * 0.75 * 2.0 ** -7
    assert torch.isfinite(lse).all()
    dqkv = ops.attention_bwd(qkv, o, torch.ones_like(o), lse, B, N, H, dh, dh ** -0.5).view(B, N, 3, H, dh).float()
    assert float(dqkv[:, :, 0].abs().max()) < 2e-2 and float(dqkv[:, :, 1].abs().max()) < 2e-2
    assert float((dqkv[:, :, 2].sum(dim=1) - N).abs().max()) < 0.02 * N


# ---------------------------------------------------------------------------------------------- config 2: ViT-B/16
VITB = dict(B=64, N=197, D=768, H=12)


@pytest.mark.parametrize("which", ["qkv", "proj", "fc1", "fc2"])
def test_vit_b16_block_gemm_shapes(which):
    """the four Linear layers of a ViT-B Block at B = 64 (12608 token rows): forward with its fused epilogue, data gradient (through the
    transposed weight shadow's layout: KC x KC) and weight gradient, against fp32 products of the same bf16 operands"""
    from UCF_VIT._hip import ops
    from UCF_VIT._hip.lib import ACT_GELU
    M, D = VITB["B"] * VITB["N"], VITB["D"]
    n_out, k_in = {"qkv": (3 * D, D), "proj": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D)}[which]
    gen = torch.Generator().manual_seed(len(which) * 7 + n_out)
    x = torch.randn(M, k_in, generator=gen).bfloat16().to(DEV)
    w = (torch.randn(n_out, k_in, generator=gen) * 0.04).bfloat16().to(DEV)
    b = torch.randn(n_out, generator=gen).bfloat16().to(DEV)
    dy = torch.randn(M, n_out, generator=gen).bfloat16().to(DEV)
    res = torch.randn(M, n_out, generator=gen).bfloat16().to(DEV)
    pre = x.float() @ w.float().T + b.float()
    if which == "fc1":
        y = ops.linear_fwd(x, w, b, act=ACT_GELU)
        ref = torch.nn.functional.gelu(pre)
    elif which in ("proj", "fc2"):
        y = ops.linear_fwd(x, w, b, residual=res)
        ref = pre + res.float()
    else:
        y = ops.linear_fwd(x, w, b)
        ref = pre
    assert rel_err(y.float(), ref) < 1e-2
    dx = ops.linear_dgrad_t(dy, w.T.contiguous())
    assert rel_err(dx.float(), dy.float() @ w.float()) < 1e-2
    dx2 = ops.linear_dgrad(dy, w)
    assert rel_err(dx2.float(), dy.float() @ w.float()) < 1e-2
    dw = ops.linear_wgrad(dy, x)
    assert rel_err(dw, dy.float().T @ x.float()) < 2e-3


def test_vit_b16_training_step_is_bitwise_reproducible_and_decreases_loss():
    """BASELINE config 2 on one GPU: ViT-B/16 224^2 bf16, batch 64; two identical steps are bit-identical (no atomics anywhere) and five
    steps on one batch reduce the loss"""
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    from UCF_VIT.utils.misc import configure_optimizer
    Bs = VITB["B"]
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 256, (Bs, 3, 224, 224), generator=g).float().to(DEV)
    y = torch.randint(0, 1000, (Bs,), generator=g).to(DEV)

    def run(steps):
        torch.manual_seed(11)
        m = VIT(img_size=[224, 224], patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12).to(DEV)
        m.set_compute_dtype(torch.bfloat16)
        opt = configure_optimizer(m, 1e-4, 0.9, 0.95, 1e-5)
        losses = []
        for _ in range(steps):
            out = m(x, VARS)
            loss = cross_entropy_loss(out, y)
            loss.backward()
            opt.step()
            opt.zero_grad()
            losses.append(loss.item())
        return out.detach().clone(), losses, m._ucf_store.flat_p.detach().clone()

    o1, l1, p1 = run(1)
    o2, l2, p2 = run(1)
    assert torch.equal(o1, o2) and l1 == l2 and torch.equal(p1, p2)
    _, l5, p5 = run(5)
    assert all(math.isfinite(v) for v in l5) and torch.isfinite(p5).all()
    assert l5[0] == l1[0] and l5[-1] < l5[0]
    assert abs(l5[0] - math.log(1000.0)) < 1.0            # random init: the loss starts near ln(classes)


# ---------------------------------------------------------------------------------------------- config 4: MAE ViT-L/16, r = 0.75
def _mae_l():
    from UCF_VIT.simple.arch import MAE
    torch.manual_seed(3)
    m = MAE(img_size=[224, 224], patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16, class_token=False, weight_init='skip',
            mask_ratio=0.75, linear_decoder=False, decoder_depth=8, decoder_embed_dim=512, decoder_num_heads=16, mlp_ratio_decoder=4.0).to(DEV)
    m.set_compute_dtype(torch.bfloat16)
    return m


def test_mae_vit_l16_full_size_step():
    """train_masked_simple.py:35-49 at the size of BASELINE config 4 (encoder on 49 kept tokens of 196, 8 x 512 / 16-head decoder on all
    196, attention shapes (49, 64) and (196, 32)): the mask is bit-exact against torch.argsort of the injected noise, the kept tokens
    are exactly the gathered rows, the step is bitwise reproducible and the loss decreases."""
    from UCF_VIT.utils.metrics import patch_mse_loss
    from UCF_VIT.utils.misc import configure_optimizer
    Bs, L = 48, 196
    g = torch.Generator().manual_seed(21)
    x = torch.rand(Bs, 3, 224, 224, generator=g).to(DEV)
    noise = torch.rand(Bs, L, generator=g)
    # the reference's index arithmetic (arch.py:671-679) on the CPU: argsort twice, mask gathered through ids_restore
    ids_shuffle = torch.argsort(noise, dim=1)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    mask_ref = torch.ones(Bs, L)
    mask_ref[:, :49] = 0
    mask_ref = torch.gather(mask_ref, 1, ids_restore)

    def run(steps):
        m = _mae_l()
        opt = configure_optimizer(m, 1e-4, 0.9, 0.95, 0.05)
        out = []
        for _ in range(steps):
            pred, mask = m(x, VARS, None, noise=noise.to(DEV))
            loss = patch_mse_loss(pred, x, 16, mask)
            loss.backward()
            opt.step()
            opt.zero_grad()
            out.append(loss.item())
        return pred.detach().clone(), mask.detach().clone(), out, m

    pred1, mask1, l1, m1 = run(1)
    assert tuple(pred1.shape) == (Bs, L, 768) and tuple(mask1.shape) == (Bs, L)
    assert torch.equal(mask1.cpu(), mask_ref)
    assert float(mask1.sum(dim=1).min()) == 147.0 and float(mask1.sum(dim=1).max()) == 147.0
    pred2, mask2, l2, m2 = run(1)
    assert torch.equal(pred1, pred2) and l1 == l2 and torch.equal(m1._ucf_store.flat_p, m2._ucf_store.flat_p)
    # kept tokens = gathered rows of the position-embedded sequence, bit for bit (random_masking on a known sequence)
    seq = torch.randn(Bs, L, 1024, generator=g).bfloat16().to(DEV)
    kept, mask3, ids3 = m1.random_masking(seq, noise.to(DEV))
    assert torch.equal(ids3.cpu(), ids_restore)
    assert torch.equal(kept.cpu(), torch.gather(seq.cpu(), 1, ids_shuffle[:, :49].unsqueeze(-1).expand(-1, -1, 1024)))
    _, _, l4, m4 = run(4)
    assert all(math.isfinite(v) for v in l4) and l4[-1] < l4[0]
    assert torch.isfinite(m4._ucf_store.flat_p).all()


# ---------------------------------------------------------------------------------------------- config 5: UNETR 512 x 512 x 128
UNETR_KW = dict(patch_size=16, in_chans=1, embed_dim=768, depth=12, num_heads=12, class_token=False, twoD=False, num_classes=4,
                linear_decoder=False, feature_size=16, skip_connection=True)


def test_unetr_encoder_reduced_volume_vs_oracle():
    """reduced volume 128 x 128 x 128, p 16 -> 8 x 8 x 8 = 512 tokens (the N = 512 case of SURVEY §8e) at the full width D 768 / 12 heads /
    12 blocks, i.e. the streaming attention kernels inside the model, against the CPU oracle's forward_intermediates
    (oracle.vit_forward_intermediates, arch.py:995-1086): fp32 mode 1e-3, bf16 mode 5e-2."""
    from UCF_VIT.simple.arch import UNETR
    from oracle import ucf_vit_ref as R
    img = [128, 128, 128]
    ref = R.VIT(img, patch_size=16, in_chans=1, num_classes=None, embed_dim=768, depth=12, num_heads=12, class_token=False, twoD=False)
    sd = det_state_dict(ref, 61)
    ref.load_state_dict(sd)
    x = det_tensor((1, 1, *img), 62)
    with torch.no_grad():
        feats_ref, taps_ref = R.vit_forward_intermediates(ref, x, [3, 6, 9])
    for dtype, tol in ((torch.float32, 1e-3), (torch.bfloat16, 5e-2)):
        m = UNETR(img_size=img, **UNETR_KW)
        m.load_state_dict(sd, strict=False)
        m = m.to(DEV)
        m.set_compute_dtype(dtype)
        assert m.skip_indices == [3, 6, 9] and m.num_patches == 512
        with torch.no_grad():
            feats, taps = m.forward_intermediates(x.to(DEV), None, None, indices=m.skip_indices)
        assert rel_err(feats.float(), feats_ref) < tol
        assert len(taps) == 3
        for a, b in zip(taps, taps_ref):
            assert tuple(a.shape) == (1, 512, 768)
            assert rel_err(a.float(), b) < tol
        del m


def test_unetr_encoder_full_size_512x512x128():
    """BASELINE config 5's encoder on one GPU: [1, 1, 512, 512, 128] -> 8192 tokens of 768 through 12 Blocks with taps after blocks
    3, 6, 9.  Properties: finite taps of the right shape; the final-normed features are standardised rows scaled by the norm's weight
    and bias; two runs are bit-identical; the backward pass gives finite gradients for every encoder parameter."""
    from UCF_VIT.simple.arch import UNETR
    img = [512, 512, 128]
    torch.manual_seed(8)
    m = UNETR(img_size=img, **UNETR_KW).to(DEV)
    m.set_compute_dtype(torch.bfloat16)
    assert m.num_patches == 8192 and m.skip_indices == [3, 6, 9]
    g = torch.Generator().manual_seed(9)
    x = torch.rand(1, 1, *img, generator=g).to(DEV)
    feats, taps = m.forward_intermediates(x, None, None, indices=m.skip_indices)
    assert tuple(feats.shape) == (1, 8192, 768) and len(taps) == 3
    for t in taps:
        assert tuple(t.shape) == (1, 8192, 768) and torch.isfinite(t.float()).all()
    f = feats.detach().float()
    w, b = m.norm.weight.float(), m.norm.bias.float()
    z = (f - b) / w                                      # undo the affine part: rows of zero mean and unit variance
    assert float(z.mean(dim=-1).abs().max()) < 2e-2 and float((z.var(dim=-1, unbiased=False) - 1).abs().max()) < 5e-2
    with torch.no_grad():
        feats2, taps2 = m.forward_intermediates(x, None, None, indices=m.skip_indices)
    assert torch.equal(feats, feats2) and all(torch.equal(a, b_) for a, b_ in zip(taps, taps2))
    (feats.float().square().mean() + sum(t.float().square().mean() for t in taps)).backward()
    for n, p in m.named_parameters():
        if n.startswith(("blocks.", "patch_embed.", "norm.", "pos_embed")):
            assert p.grad is not None and torch.isfinite(p.grad).all(), n


def test_unetr_whole_model_full_size_512x512x128():
    """BASELINE config 5 complete: encoder + skip-connection convolutional decoder (HIP convolution kernels, channels-last bf16) + Dice/CE at
    512 x 512 x 128 on one GPU.  Size-independent properties: logits of the reference's shape [B, classes, X, Y, Z], finite; at random
    initialisation the loss sits near ln(classes) + mean dice of a near-uniform prediction; two runs are bit-identical; every parameter of
    the model receives a finite gradient; the loss gradient of the logits sums to ~0 over the classes of a voxel (softmax + dice both do)."""
    from UCF_VIT.simple.arch import UNETR
    from UCF_VIT._hip import functional as HF
    img = [512, 512, 128]
    torch.manual_seed(11)
    m = UNETR(img_size=img, **UNETR_KW).to(DEV)
    m.set_compute_dtype(torch.bfloat16)
    assert m.hip_decoder()
    g = torch.Generator().manual_seed(12)
    x = torch.rand(1, 1, *img, generator=g).to(DEV)
    lab = torch.randint(0, 4, (1, *img), generator=g).to(DEV)
    logits = m(x, None)
    assert tuple(logits.shape) == (1, 4, *img) and logits.dtype == torch.float32
    assert torch.isfinite(logits).all()
    lg = logits.detach().requires_grad_(True)
    loss_probe = HF.dice_ce(lg, lab)
    loss_probe.backward()
    assert float(lg.grad.sum(dim=1).abs().max()) < 1e-6 * max(1.0, float(lg.grad.abs().max()) * 1e6)
    loss = HF.dice_ce(logits, lab)
    assert 1.0 < loss.item() < 4.0                        # ln 4 = 1.39 for a uniform prediction, + a dice term in (0, 1); random init is not far off
    with torch.no_grad():
        logits2 = m(x, None)
    assert torch.equal(logits, logits2)
    loss.backward()
    from UCF_VIT._hip.functional import flush_wgrads
    flush_wgrads()
    n_dec = 0
    for n, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        n_dec += n.startswith(("encoder", "decoder", "out."))
    assert n_dec == 33                                     # every convolution of the decoder: 12 blocks + the output head
    assert torch.cuda.max_memory_allocated() < 120 * 2 ** 30

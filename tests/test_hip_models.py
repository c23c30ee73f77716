"""GPU parity tests of whole models on the HIP path against reference-generated golden vectors: ViT (small and the
ViT-Tiny/16 catsdogs config of BASELINE.json configs[0]), MAE, and a 5-step AdamW training trajectory."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from det_weights import det_state_dict, proj_vector

pytestmark = pytest.mark.gpu
DEV = "cuda"
VARS = ["red", "green", "blue"]


def build(cls, kw, seed, dtype=torch.float32):
    m = cls(**kw)
    m.load_state_dict(det_state_dict(m, seed))
    m = m.to(DEV)
    m.set_compute_dtype(dtype)
    return m


VIT_KW = dict(img_size=[32, 32], patch_size=8, in_chans=3, num_classes=5, embed_dim=64, depth=2, num_heads=2)
MAE_KW = dict(img_size=[32, 32], patch_size=8, in_chans=3, embed_dim=64, depth=2, num_heads=2, class_token=False, weight_init='skip',
              mask_ratio=0.75, linear_decoder=False, decoder_depth=1, decoder_embed_dim=32, decoder_num_heads=1, mlp_ratio_decoder=4.0)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
def test_vit_small_vs_reference(dtype, tol):
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    g = load_golden("model_vit_small.npz")
    m = build(VIT, VIT_KW, 21, dtype)
    out = m(g["x"].to(DEV), VARS)
    loss = cross_entropy_loss(out, g["labels"].to(DEV))
    loss.backward()
    assert out.dtype == dtype
    assert rel_err(out.float(), g["logits"]) < tol
    assert abs(loss.item() - g["loss"].item()) < tol * max(1.0, abs(g["loss"].item()))
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        assert rel_err(p.grad, g["g." + k]) < tol, k
    # gradients were written by the kernels straight into the flat buffer (no copies)
    st = m._ucf_store
    for p, o in zip(st.params, st.offsets):
        assert p.grad.data_ptr() == st.flat_g.data_ptr() + 4 * o


ADAPTIVE_CASES = [("model_vit_adaptive.npz", dict(img_size=[32, 32], patch_size=8, in_chans=3, twoD=True, use_adaptive_pos_emb=True), 52),
                  ("model_vit_adaptive_learnpos.npz", dict(img_size=[32, 32], patch_size=8, in_chans=3, twoD=True, use_adaptive_pos_emb=False), 53),
                  ("model_vit_adaptive_3d.npz", dict(img_size=[16, 16, 16], patch_size=4, in_chans=1, twoD=False, use_adaptive_pos_emb=True), 56)]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
@pytest.mark.parametrize("name,kw,seed", ADAPTIVE_CASES)
def test_vit_adaptive_patching_vs_reference(name, kw, seed, dtype, tol):
    """VIT(adaptive_patching=True) (arch.py:282-289, :311-321, :366-393, :465-467): x [B, C, S, P] pre-cut patches + seq_ps [B, S, 3|4];
    fixtures generated from the reference by tests/golden/make_golden_adaptive.py"""
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    g = load_golden(name)
    m = VIT(num_classes=5, embed_dim=64, depth=2, num_heads=2, adaptive_patching=True, fixed_length=12, **kw)
    m.load_state_dict(det_state_dict(m, seed, keep=()))
    m = m.to(DEV)
    m.set_compute_dtype(dtype)
    out = m(g["x"].to(DEV), VARS, g["seq_ps"].to(DEV))
    loss = cross_entropy_loss(out, g["labels"].to(DEV))
    loss.backward()
    assert out.dtype == dtype
    assert rel_err(out.float(), g["logits"]) < tol
    assert abs(loss.item() - g["loss"].item()) < tol * max(1.0, abs(g["loss"].item()))
    for k, p in m.named_parameters():
        ref = g["g." + k]
        if float(ref.abs().max()) == 0.0:                    # pos_embed is unused when positions come from seq_ps
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert p.grad is not None, k
            assert rel_err(p.grad, ref) < tol, k


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
@pytest.mark.parametrize("name,adaptive_pos,seed", [("model_mae_adaptive.npz", True, 58), ("model_mae_adaptive_learnpos.npz", False, 59)])
def test_mae_adaptive_patching_vs_reference(name, adaptive_pos, seed, dtype, tol):
    """MAE(adaptive_patching=True) (arch.py:538-755, train_masked_simple.py:24-32): the mask is bit-exact, prediction / loss / every
    gradient within the fp32 (bf16) tolerance; the target 'b c s p -> b s (p c)' is read in place by the loss kernel"""
    from UCF_VIT.simple.arch import MAE
    from UCF_VIT.utils.metrics import seq_mse_loss
    g = load_golden(name)
    m = MAE(img_size=[32, 32], patch_size=8, in_chans=3, embed_dim=64, depth=2, num_heads=2, adaptive_patching=True, fixed_length=12,
            class_token=False, weight_init='skip', mask_ratio=0.5, linear_decoder=False, decoder_depth=1, decoder_embed_dim=32,
            decoder_num_heads=1, mlp_ratio_decoder=4.0, use_adaptive_pos_emb=adaptive_pos)
    m.load_state_dict(det_state_dict(m, seed, keep=()))
    m = m.to(DEV)
    m.set_compute_dtype(dtype)
    x, sp = g["x"].to(DEV), g["seq_ps"].to(DEV)
    pred, mask = m(x, VARS, sp, noise=g["noise"].to(DEV))
    assert torch.equal(mask.cpu(), g["mask"])
    loss = seq_mse_loss(pred, x)
    loss.backward()
    assert rel_err(pred.float(), g["pred"]) < tol
    assert abs(loss.item() - g["loss"].item()) < tol * max(1.0, abs(g["loss"].item()))
    with torch.no_grad():
        lm = seq_mse_loss(pred.detach(), x, mask)
    assert abs(lm.item() - g["loss_masked"].item()) < tol * max(1.0, abs(g["loss_masked"].item()))
    for k, p in m.named_parameters():
        ref = g["g." + k]
        if float(ref.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert p.grad is not None, k
            assert rel_err(p.grad, ref) < tol, k


def _check_grads_vs_golden(m, g, tol):
    for k, p in m.named_parameters():
        ref = g["g." + k]
        if float(ref.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert p.grad is not None, k
            assert rel_err(p.grad, ref) < tol, k


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
def test_vit_sqrt_len_method_3d_vs_reference(dtype, tol):
    """adaptive patching as train_unetr_simple.py:43-49 / train_sap_simple.py:28-43 use it (sqrt_len_method=True): the token sequence
    reshaped into a pseudo volume goes through the patch-embedding convolution (im2col + GEMM), positions come from seq_ps [B, S, 4]"""
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    g = load_golden("model_vit_sqrtlen_3d.npz")
    m = VIT(img_size=[16, 16, 16], patch_size=4, in_chans=1, num_classes=5, embed_dim=96, depth=2, num_heads=3, adaptive_patching=True,
            fixed_length=8, twoD=False, use_adaptive_pos_emb=True, sqrt_len_method=True, class_token=False)
    m.load_state_dict(det_state_dict(m, 62, keep=()))
    m = m.to(DEV)
    m.set_compute_dtype(dtype)
    out = m(g["x"].to(DEV), VARS, g["seq_ps"].to(DEV))
    assert tuple(out.shape) == (2, 8, 5)
    loss = cross_entropy_loss(out.flatten(0, 1), g["labels"].to(DEV))
    loss.backward()
    assert rel_err(out.float(), g["logits"]) < tol
    assert abs(loss.item() - g["loss"].item()) < tol * max(1.0, abs(g["loss"].item()))
    _check_grads_vs_golden(m, g, tol)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
def test_sap_adaptive_vs_reference(dtype, tol):
    """SAP as train_sap_simple.py:231-250 builds it for adaptively patched input; encoder on the HIP kernels, the transposed-convolution
    neck and the 1x1 header on torch / MIOpen"""
    from UCF_VIT.simple.arch import SAP
    g = load_golden("model_sap_adaptive.npz")
    m = SAP(img_size=[64, 64], patch_size=8, in_chans=3, num_classes=3, embed_dim=64, depth=2, num_heads=2, adaptive_patching=True,
            fixed_length=16, sqrt_len=4, twoD=True, use_adaptive_pos_emb=True, sqrt_len_method=True, class_token=False, weight_init='skip')
    m.load_state_dict(det_state_dict(m, 65, keep=()))
    m = m.to(DEV)
    m.set_compute_dtype(dtype)
    out = m(g["x"].to(DEV), VARS, g["seq_ps"].to(DEV))
    loss = torch.nn.MSELoss()(out.float(), g["target"].to(DEV))
    loss.backward()
    assert rel_err(out.float(), g["out"]) < tol
    assert abs(loss.item() - g["loss"].item()) < tol * max(1.0, abs(g["loss"].item()))
    _check_grads_vs_golden(m, g, tol)


def test_unetr_adaptive_sqrt_len_encoder_vs_oracle_and_trains():
    """UNETR on adaptively patched 3-D input (basic_ct/unetr config: adaptive_patching, use_adaptive_pos_emb, single channel):
    forward(x, variables, seq_ps, x_seq) with x_seq = the pseudo volume; encoder output and taps against the CPU oracle
    (SqrtLenVIT, pinned by model_vit_sqrtlen_3d.npz), then a backward pass through the conv decoder (torch / MIOpen)"""
    from oracle import ucf_vit_ref as R
    from UCF_VIT.simple.arch import UNETR
    from det_weights import det_tensor
    kw = dict(img_size=[32, 32, 32], patch_size=4, in_chans=1, embed_dim=96, depth=4, num_heads=3, class_token=False, twoD=False)
    ref = R.SqrtLenVIT(kw["img_size"], patch_size=4, in_chans=1, num_classes=None, embed_dim=96, depth=4, num_heads=3, class_token=False, twoD=False)
    sd = det_state_dict(ref, 71, keep=())
    ref.load_state_dict(sd)
    m = UNETR(num_classes=3, linear_decoder=False, feature_size=4, skip_connection=True, allow_torch_decoder=True, adaptive_patching=True, fixed_length=8, sqrt_len=2,
              use_adaptive_pos_emb=True, sqrt_len_method=True, **kw)
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    m = m.to(DEV)
    x_seq, sp = det_tensor((2, 1, 8, 8, 8), 72), torch.rand(2, 8, 4, generator=torch.Generator().manual_seed(3)) * 16
    feats, taps = m.forward_intermediates(x_seq.to(DEV), VARS, sp.to(DEV), indices=m.skip_indices)
    with torch.no_grad():
        t = ref._pos_embed(ref.token_embeds(x_seq), sp)
        want_taps = []
        for i, blk in enumerate(ref.blocks):
            t = blk(t)
            if i in m.skip_indices:
                want_taps.append(t)
        want = ref.norm(t)
    assert rel_err(feats.float().cpu(), want) < 1e-3
    for a, b in zip(taps, want_taps):
        assert rel_err(a.float().cpu(), b) < 1e-3
    # the decoder needs grids of 16 x feat_size = img_size or an upsample: here 2 * 16 = 32 = img_size
    out = m(det_tensor((2, 1, 32, 32, 32), 73).to(DEV), VARS, sp.to(DEV), x_seq.to(DEV))
    assert tuple(out.shape) == (2, 3, 32, 32, 32)
    out.float().square().mean().backward()
    gsum = sum(float(p.grad.abs().sum()) for p in m.blocks.parameters())
    assert math.isfinite(gsum) and gsum > 0 and m.adaptive_pos_dep_emb[0].weight.grad is not None


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
@pytest.mark.parametrize("name,single,variables,seed,chans", [("model_vit_varemb.npz", False, ["v", "q", "u"], 81, 3),
                                                              ("model_vit_varemb_single.npz", True, ["t"], 83, 1)])
def test_vit_variable_aggregation_vs_reference(name, single, variables, seed, chans, dtype, tol):
    """VIT(use_varemb=True) (README "Variable Aggregation", arch.py:395-462, building_blocks.py:301-373) on adaptively patched input,
    3 of 4 default variables passed out of order: logits, loss and every gradient (the token embeddings of the unused variable stay 0)"""
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    g = load_golden(name)
    m = VIT(img_size=[32, 32], patch_size=8, in_chans=chans, num_classes=5, embed_dim=64, depth=2, num_heads=2, use_varemb=True,
            default_vars=["u", "v", "t", "q"], single_channel=single, adaptive_patching=True, fixed_length=12, use_adaptive_pos_emb=True)
    m.load_state_dict(det_state_dict(m, seed, keep=()))
    m = m.to(DEV)
    m.set_compute_dtype(dtype)
    out = m(g["x"].to(DEV), variables, g["seq_ps"].to(DEV))
    loss = cross_entropy_loss(out, g["labels"].to(DEV))
    loss.backward()
    assert rel_err(out.float(), g["logits"]) < tol
    assert abs(loss.item() - g["loss"].item()) < tol * max(1.0, abs(g["loss"].item()))
    _check_grads_vs_golden(m, g, tol)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
def test_mae_variable_aggregation_vs_reference(dtype, tol):
    """MAE(use_varemb=True) on adaptively patched input (the same front end through MAE.forward_features, arch.py:704-745): prediction,
    bit-exact mask, loss against the rearranged sequence, every gradient; fixture generated from the reference"""
    from UCF_VIT.simple.arch import MAE
    from UCF_VIT.utils.metrics import seq_mse_loss
    g = load_golden("model_mae_varemb.npz")
    m = MAE(img_size=[32, 32], patch_size=8, in_chans=3, embed_dim=64, depth=2, num_heads=2, adaptive_patching=True, fixed_length=12,
            class_token=False, weight_init='skip', mask_ratio=0.5, linear_decoder=False, decoder_depth=1, decoder_embed_dim=32,
            decoder_num_heads=1, mlp_ratio_decoder=4.0, use_varemb=True, default_vars=["u", "v", "t", "q"], single_channel=False,
            use_adaptive_pos_emb=True)
    m.load_state_dict(det_state_dict(m, 87, keep=()))
    m = m.to(DEV)
    m.set_compute_dtype(dtype)
    x = g["x"].to(DEV)
    pred, mask = m(x, ["q", "u", "t"], g["seq_ps"].to(DEV), noise=g["noise"].to(DEV))
    assert torch.equal(mask.cpu(), g["mask"])
    loss = seq_mse_loss(pred, x)
    loss.backward()
    assert rel_err(pred.float(), g["pred"]) < tol
    assert abs(loss.item() - g["loss"].item()) < tol * max(1.0, abs(g["loss"].item()))
    _check_grads_vs_golden(m, g, tol)


def test_vit_adaptive_patching_trains_like_the_oracle():
    """10 AdamW steps on one adaptive batch (bf16 HIP path against the fp32 CPU oracle): same loss curve, loss goes down"""
    from oracle import ucf_vit_ref as R
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    from UCF_VIT.utils.misc import configure_optimizer
    g = load_golden("model_vit_adaptive.npz")
    kw = ADAPTIVE_CASES[0][1]
    m = VIT(num_classes=5, embed_dim=64, depth=2, num_heads=2, adaptive_patching=True, fixed_length=12, **kw)
    sd = det_state_dict(m, 52, keep=())
    m.load_state_dict(sd)
    m = m.to(DEV)
    m.set_compute_dtype(torch.bfloat16)
    ref = R.AdaptiveVIT(patch_size=8, in_chans=3, num_classes=5, embed_dim=64, depth=2, num_heads=2, fixed_length=12)
    ref.load_state_dict(sd)
    opt = configure_optimizer(m, 2e-3, 0.9, 0.95, 1e-5)
    ropt = R.configure_optimizer(ref, 2e-3, 0.9, 0.95, 1e-5)
    x, sp, y = g["x"], g["seq_ps"], g["labels"]
    got, want = [], []
    for _ in range(10):
        loss = cross_entropy_loss(m(x.to(DEV), VARS, sp.to(DEV)), y.to(DEV))
        loss.backward()
        opt.step()
        opt.zero_grad()
        got.append(loss.item())
        rl = torch.nn.CrossEntropyLoss()(ref(x, None, sp), y)
        ropt.zero_grad()
        rl.backward()
        ropt.step()
        want.append(rl.item())
    for i, (a, b) in enumerate(zip(got, want)):
        assert abs(a - b) <= 0.05 * abs(b) + 0.02, (i, a, b)
    assert got[-1] < 0.9 * got[0] and want[-1] < 0.9 * want[0], (got, want)


def test_vit_tiny_config_T_vs_reference():
    """BASELINE.json configs[0]: ViT-Tiny/16 on catsdogs-shaped input (256x256, pixels 0..255, 2 classes), fp32"""
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.fused_attn import FusedAttn
    from UCF_VIT.utils.metrics import cross_entropy_loss
    g = load_golden("model_vit_tiny_catsdogs.npz")
    m = build(VIT, dict(img_size=[256, 256], patch_size=16, in_chans=3, num_classes=2, embed_dim=192, depth=12, num_heads=3,
                        FusedAttn_option=FusedAttn.DEFAULT), 23)
    out = m(g["x"].to(DEV), VARS)
    loss = cross_entropy_loss(out, g["labels"].to(DEV))
    loss.backward()
    assert rel_err(out, g["logits"]) < 1e-3
    assert abs(loss.item() - g["loss"].item()) < 1e-3 * max(1.0, abs(g["loss"].item()))
    for i, (k, p) in enumerate(m.named_parameters()):
        gn_ref = g["gn." + k].item()
        gn = p.grad.double().norm().item()
        assert abs(gn - gn_ref) <= 1e-3 * max(gn_ref, 1e-12), k
        gp = (p.grad.double().cpu() * proj_vector(p.shape, i).double()).sum().item()
        assert abs(gp - g["gp." + k].item()) <= 1e-3 * max(gn_ref * math.sqrt(p.numel()), 1e-12), k


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
def test_mae_small_vs_reference(dtype, tol):
    from UCF_VIT.simple.arch import MAE
    from UCF_VIT.utils.metrics import patch_mse_loss
    g = load_golden("model_mae_small.npz")
    m = build(MAE, MAE_KW, 26, dtype)
    x = g["x"].to(DEV)
    pred, mask = m(x, VARS, noise=g["noise"].to(DEV))
    assert torch.equal(mask.cpu(), g["mask"])                      # bit-exact mask
    assert rel_err(pred.float(), g["pred"]) < tol
    loss = patch_mse_loss(pred, x, 8)
    assert abs(loss.item() - g["loss"].item()) < tol * abs(g["loss"].item())
    lm = patch_mse_loss(pred.detach(), x, 8, mask)
    assert abs(lm.item() - g["loss_masked"].item()) < tol * abs(g["loss_masked"].item())
    loss.backward()
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        assert rel_err(p.grad, g["g." + k]) < tol, k


def test_adamw_trajectory_vs_reference():
    """5 training steps in the reference loop order (train_class_simple.py:344-357) with the fused AdamW + closed-form schedule"""
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    from UCF_VIT.utils.misc import configure_optimizer, configure_scheduler
    g = load_golden("traj_vit_small.npz")
    m = build(VIT, VIT_KW, 31)
    opt = configure_optimizer(m, 1e-3, 0.9, 0.95, 1e-2)
    sch = configure_scheduler(opt, 2, 10, 1e-5, 1e-6)
    for i in range(5):
        out = m(g["x%d" % i].to(DEV), VARS)
        loss = cross_entropy_loss(out, g["labels"][i].to(DEV))
        assert abs(loss.item() - g["losses"][i].item()) < 1e-3 * max(1.0, g["losses"][i].item()), i
        loss.backward()
        opt.step()
        opt.zero_grad()
        sch.step()
    assert bool(opt._flat), "fused flat-segment AdamW path was not taken"
    for k, v in m.state_dict().items():
        if k.startswith("token_embeds"):
            continue
        ref = g["final." + k]
        if k.endswith("attn.qkv.bias"):
            # d(loss)/d(k-bias) is exactly zero in exact arithmetic (softmax shift invariance): its computed gradient is pure
            # rounding noise which Adam normalises to +-lr steps, in the reference as well -> compare the q and v thirds only
            D = ref.numel() // 3
            v, ref = torch.cat([v[:D], v[2 * D:]]), torch.cat([ref[:D], ref[2 * D:]])
        assert rel_err(v, ref) < 1e-3, k


def test_bf16_loss_curve_tracks_cpu_reference():
    """BASELINE north_star: 'loss-curve-equivalent to the CPU reference' (50-100 steps, every loss within 2 % rel and the
    same trend).  60 optimiser steps of the classification loop (reference
    order, train_class_simple.py:344-357) on a fixed cycle of 4 synthetic batches: the bf16 HIP path (fast-GELU epilogues, saved
    gelu', fused attention, grouped weight gradients, by-product bias gradients, fused AdamW) against the fp32 CPU oracle from the
    same deterministic initialisation.  Tolerance: every loss within 2 % of the oracle's (measured: < 0.1 %),
    and both curves must go down (random labels: final loss below 97 % of the first)."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ucf_vit_ref as R
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    from UCF_VIT.utils.misc import configure_optimizer, configure_scheduler
    kw = dict(img_size=[32, 32], patch_size=8, in_chans=3, num_classes=5, embed_dim=64, depth=3, num_heads=2)
    m = build(VIT, kw, 77, torch.bfloat16)
    ref = R.VIT(kw["img_size"], patch_size=8, in_chans=3, num_classes=5, embed_dim=64, depth=3, num_heads=2, sdpa=True)
    ref.load_state_dict(det_state_dict(ref, 77))
    opt = configure_optimizer(m, 2e-3, 0.9, 0.95, 1e-2)
    sch = configure_scheduler(opt, 5, 60, 1e-5, 1e-6)
    ropt = R.configure_optimizer(ref, 2e-3, 0.9, 0.95, 1e-2)
    rsch = R.WarmupCosineLR(ropt, 5, 60, 1e-5, 1e-6)
    gen = torch.Generator().manual_seed(123)
    xs = [torch.rand(16, 3, 32, 32, generator=gen) for _ in range(4)]
    ys = [torch.randint(0, 5, (16,), generator=gen) for _ in range(4)]
    got, want = [], []
    for i in range(60):
        x, y = xs[i % 4], ys[i % 4]
        loss = cross_entropy_loss(m(x.to(DEV), VARS), y.to(DEV))
        loss.backward()
        opt.step()
        opt.zero_grad()
        sch.step()
        got.append(loss.item())
        rl, _ = R.train_step_class(ref, ropt, rsch, x, y)
        want.append(rl.item())
    for i, (a, b) in enumerate(zip(got, want)):
        assert abs(a - b) <= 0.02 * abs(b), (i, a, b)
    assert got[-1] < 0.97 * got[0] and want[-1] < 0.97 * want[0], (got[0], got[-1], want[0], want[-1])


def test_deferred_weight_gradients_accumulate_and_serve_autograd_grad():
    """the grouped weight-gradient launches are deferred to the end of backward (functional.WgradQueue): (i) two backward passes
    without zero_grad accumulate exactly twice the gradient into the flat buffer, (ii) torch.autograd.grad() sees finished
    gradients, (iii) a model WITHOUT a flat store (plain parameters) gets the same numbers"""
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    g = load_golden("model_vit_small.npz")
    x, y = g["x"].to(DEV), g["labels"].to(DEV)
    m = build(VIT, VIT_KW, 21)
    cross_entropy_loss(m(x, VARS), y).backward()
    once = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    for k, p in m.named_parameters():
        assert rel_err(p.grad, g["g." + k]) < 1e-3, k
    cross_entropy_loss(m(x, VARS), y).backward()              # accumulate
    for k, p in m.named_parameters():
        assert rel_err(p.grad, 2 * once[k]) < 1e-5, k
    m2 = build(VIT, VIT_KW, 21)
    params = [p for p in m2.parameters()]
    grads = torch.autograd.grad(cross_entropy_loss(m2(x, VARS), y), params)
    torch.cuda.synchronize()
    for (k, _), gr in zip(m2.named_parameters(), grads):
        assert rel_err(gr, once[k]) < 1e-5, k


def test_bf16_shadow_follows_master():
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    from UCF_VIT.utils.misc import configure_optimizer
    g = load_golden("model_vit_small.npz")
    m = build(VIT, VIT_KW, 21, torch.bfloat16)
    opt = configure_optimizer(m, 1e-2, 0.9, 0.95, 0.0)
    x, y = g["x"].to(DEV), g["labels"].to(DEV)
    for _ in range(2):
        cross_entropy_loss(m(x, VARS), y).backward()
        opt.step()
        opt.zero_grad()
    st = m._ucf_store
    assert torch.equal(st.flat_s.float(), st.flat_p.to(torch.bfloat16).float())
    with torch.no_grad():
        m.head.weight.mul_(0.5)          # an outside modification bumps the version -> shadow re-cast on next forward
    m(x, VARS)
    assert torch.equal(st.flat_s.float(), st.flat_p.to(torch.bfloat16).float())


def test_state_dict_layout_matches_reference_listing():
    from UCF_VIT.simple.arch import MAE, VIT
    keys = list(VIT(**VIT_KW).state_dict().keys())
    assert keys[:4] == ["cls_token", "pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias"]
    assert "token_embeds.proj.weight" in keys and "head.weight" in keys and "blocks.1.mlp.fc2.bias" in keys
    mk = set(MAE(**MAE_KW).state_dict().keys())
    for k in ("mask_token", "decoder_pos_embed", "decoder_embed.weight", "decoder_norm.bias", "decoder_pred.weight",
              "decoder_blocks.0.attn.qkv.weight"):
        assert k in mk


@pytest.mark.parametrize("reduce_dtype", [None, torch.bfloat16])
def test_hip_data_parallel_world1_nccl(reduce_dtype):
    """the RCCL reducer path (flat-buffer buckets, AVG all-reduce on the nccl backend, end-of-backward stream wait) on one rank;
    with bf16 gradient transport (staging buffer, cast back at the end of backward) the gradients are the bf16-rounded ones"""
    import os
    import torch.distributed as dist
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    from UCF_VIT.utils.misc import configure_optimizer
    from UCF_VIT._hip.ddp import HipDataParallel
    g = load_golden("model_vit_small.npz")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29533" if reduce_dtype is None else "29534"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        m = build(VIT, VIT_KW, 21)
        ddp = HipDataParallel(m, bucket_mb=0.05, reduce_dtype=reduce_dtype)
        assert len(ddp.buckets) >= 3
        assert ddp.reduce_dtype() == (reduce_dtype or torch.float32)
        opt = configure_optimizer(m, 1e-3, 0.9, 0.95, 1e-2)
        for _ in range(2):
            out = ddp(g["x"].to(DEV), VARS, None)
            loss = cross_entropy_loss(out, g["labels"].to(DEV))
            loss.backward()
            if _ == 0:
                for k, p in m.named_parameters():
                    assert rel_err(p.grad, g["g." + k]) < (1e-3 if reduce_dtype is None else 8e-3), k
                    if reduce_dtype is not None:
                        assert torch.equal(p.grad, p.grad.bfloat16().float()), k
            opt.step()
            opt.zero_grad()
        assert bool(opt._flat)
        assert all(k.startswith("module.") for k in ddp.state_dict())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
def test_unetr_encoder_3d_vs_oracle(dtype, tol):
    """3-D patch embedding + 3-D sincos table + encoder taps after blocks d/4, 2d/4, 3d/4 (raw outputs)"""
    from UCF_VIT.simple.arch import UNETR
    from oracle import ucf_vit_ref as R
    from det_weights import det_tensor
    kw = dict(img_size=[32, 32, 16], patch_size=8, in_chans=1, embed_dim=96, depth=4, num_heads=3, class_token=False, twoD=False)
    ref = R.VIT(kw["img_size"], patch_size=8, in_chans=1, num_classes=None, embed_dim=96, depth=4, num_heads=3, class_token=False, twoD=False)
    sd = det_state_dict(ref, 51)
    ref.load_state_dict(sd)
    m = UNETR(num_classes=4, linear_decoder=False, feature_size=4, skip_connection=True, allow_torch_decoder=True, **kw)
    missing = m.load_state_dict(sd, strict=False)
    assert all(k.split(".")[0] in ("encoder1", "encoder2", "encoder3", "encoder4", "decoder2", "decoder3", "decoder4", "decoder5", "out")
               for k in missing.missing_keys) and not missing.unexpected_keys
    m = m.to(DEV)
    m.set_compute_dtype(dtype)
    assert m.skip_indices == [1, 2, 3]
    x = det_tensor((2, 1, 32, 32, 16), 52)
    feats_ref, taps_ref = R.vit_forward_intermediates(ref, x, m.skip_indices)
    feats, taps = m.forward_intermediates(x.to(DEV), None, None, indices=m.skip_indices)
    assert rel_err(feats.float(), feats_ref) < tol
    assert len(taps) == 3
    for a, b in zip(taps, taps_ref):
        assert rel_err(a.float(), b) < tol
    # whole model (conv decoder on MIOpen, parity unpinned): shape + finite gradients through the HIP encoder
    y = m(x.to(DEV), None)
    assert tuple(y.shape) == (2, 4, 32, 32, 16)
    y.float().square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in m.named_parameters() if n.startswith("blocks."))


def test_sap_and_diffusion_api_surface():
    from UCF_VIT.simple.arch import SAP, DiffusionVIT
    from det_weights import det_tensor
    s = SAP(img_size=[32, 32], patch_size=8, in_chans=3, num_classes=3, embed_dim=64, depth=1, num_heads=2, class_token=False, sqrt_len=4).to(DEV)
    y = s(det_tensor((2, 3, 32, 32), 1).to(DEV), None)
    assert tuple(y.shape) == (2, 3, 32, 32)
    d = DiffusionVIT(img_size=[32, 32], patch_size=8, in_chans=3, embed_dim=64, depth=1, num_heads=2, class_token=False, linear_decoder=False,
                     decoder_depth=1, decoder_embed_dim=32, decoder_num_heads=1, mlp_ratio_decoder=4.0, time_steps=10).to(DEV).eval()
    out = d(det_tensor((2, 3, 32, 32), 2).to(DEV), torch.tensor([1, 7]), None)
    assert tuple(out.shape) == (2, 16, 192)


def test_train_class_simple_entry_point_runs(tmp_path):
    """BASELINE.json configs[0] plumbing: the reference-compatible entry script runs 2 epochs of ViT-Tiny/16 (catsdogs shape,
    batch 8, synthetic data) on one GPU through RCCL world_size 1 and writes the even/odd checkpoints"""
    import os
    import subprocess
    import sys
    import yaml
    from conftest import ROOT
    cfg = yaml.safe_load(open(os.path.join(ROOT, "ucf-vit_amd", "configs", "catsdogs_vit_tiny_smoke.yaml")))
    cfg["trainer"]["checkpoint_path"] = str(tmp_path)
    p = tmp_path / "cfg.yaml"
    p.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, MASTER_PORT="29577")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "ucf-vit_amd", "training_scripts", "train_class_simple.py"), str(p)],
                         capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "epoch: 1" in out.stdout
    ck = torch.load(tmp_path / "multi_last_odd.ckpt", map_location="cpu", weights_only=True)
    assert ck["epoch"] == 1 and "module.blocks.11.mlp.fc2.weight" in ck["model_state_dict"]


def _run_entry(script, cfg, tmp_path, port):
    import os
    import subprocess
    import sys
    import yaml
    from conftest import ROOT
    cfg["trainer"]["checkpoint_path"] = str(tmp_path)
    p = tmp_path / "cfg.yaml"
    p.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "ucf-vit_amd", "training_scripts", script), str(p)],
                         capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def _smoke_cfg():
    import os
    import yaml
    from conftest import ROOT
    return yaml.safe_load(open(os.path.join(ROOT, "ucf-vit_amd", "configs", "catsdogs_vit_tiny_smoke.yaml")))


def test_train_masked_simple_entry_point_runs(tmp_path):
    """train_masked_simple.py-compatible entry script (BASELINE configs[3] path): MAE with random masking / gather, decoder,
    masked patch-MSE, HipDataParallel (RCCL world 1), bf16; 2 epochs on synthetic data; the loss must be finite and go down"""
    cfg = _smoke_cfg()
    cfg["trainer"]["data_type"] = "bfloat16"
    cfg["trainer"]["loss_fn"] = "maskMSE"
    a = cfg["model"]["net"]["init_args"]
    a.update(tile_size=[64, 64], patch_size=8, embed_dim=128, depth=3, num_heads=2, mask_ratio=0.75, linear_decoder=False,
             decoder_depth=2, decoder_embed_dim=64, decoder_num_heads=2, mlp_ratio_decoder=4.0)
    cfg["model"]["lr"] = 1e-3
    cfg["model"]["warmup_steps"] = 2
    cfg["load_balancing"]["batches_per_rank_epoch"]["catsdogs"] = 8
    out = _run_entry("train_masked_simple.py", cfg, tmp_path, 29578)
    losses = [float(l.split("epoch_loss")[1].split()[0]) for l in out.splitlines() if "epoch_loss" in l]
    assert len(losses) == 2 and all(math.isfinite(v) for v in losses) and losses[1] < losses[0], out


def test_train_scripts_run_the_adaptive_patching_configuration(tmp_path):
    """the reference's imagenet / basic_ct configs set adaptive_patching: True, fixed_length: 196, use_adaptive_pos_emb: True
    (configs/imagenet/classification/base_config.yaml:46-49): both entry scripts train on token sequences [B, C, S, P] + seq_ps"""
    cfg = _smoke_cfg()
    cfg["trainer"]["data_type"] = "bfloat16"
    a = cfg["model"]["net"]["init_args"]
    a.update(tile_size=[64, 64], patch_size=8, embed_dim=128, depth=3, num_heads=2, adaptive_patching=True, fixed_length=49, use_adaptive_pos_emb=True)
    cfg["model"]["lr"] = 1e-3
    cfg["model"]["warmup_steps"] = 2
    cfg["load_balancing"]["batches_per_rank_epoch"]["catsdogs"] = 8
    out = _run_entry("train_class_simple.py", cfg, tmp_path, 29580)
    losses = [float(l.split("epoch_loss")[1].split()[0]) for l in out.splitlines() if "epoch_loss" in l]
    assert len(losses) == 2 and all(math.isfinite(v) for v in losses), out
    a.update(mask_ratio=0.75, linear_decoder=False, decoder_depth=2, decoder_embed_dim=64, decoder_num_heads=2, mlp_ratio_decoder=4.0)
    out = _run_entry("train_masked_simple.py", cfg, tmp_path, 29581)
    losses = [float(l.split("epoch_loss")[1].split()[0]) for l in out.splitlines() if "epoch_loss" in l]
    assert len(losses) == 2 and all(math.isfinite(v) for v in losses) and losses[1] < losses[0], out


def test_train_class_script_runs_variable_aggregation(tmp_path):
    """use_varemb: True in the reference-schema config: three channels tokenised separately and aggregated, on token sequences from
    the GPU patcher; bf16"""
    cfg = _smoke_cfg()
    cfg["trainer"]["data_type"] = "bfloat16"
    a = cfg["model"]["net"]["init_args"]
    a.update(tile_size=[64, 64], patch_size=8, embed_dim=128, depth=2, num_heads=2, adaptive_patching=True, fixed_length=49, use_adaptive_pos_emb=True,
             use_varemb=True)
    cfg["model"]["lr"] = 1e-3
    cfg["model"]["warmup_steps"] = 2
    cfg["load_balancing"]["batches_per_rank_epoch"]["catsdogs"] = 6
    out = _run_entry("train_class_simple.py", cfg, tmp_path, 29584)
    losses = [float(l.split("epoch_loss")[1].split()[0]) for l in out.splitlines() if "epoch_loss" in l]
    assert len(losses) == 2 and all(math.isfinite(v) for v in losses), out


def test_train_sap_and_unetr_scripts_run_adaptive_configs(tmp_path):
    """basic_ct/sap and basic_ct/unetr set adaptive_patching + use_adaptive_pos_emb: SAP on 2-D pseudo images (fixed_length 16 = 3n+1
    and a square), UNETR on 3-D pseudo volumes (fixed_length 8 = 7n+1 and a cube) with the full volume feeding the first conv encoder"""
    cfg = _smoke_cfg()
    a = cfg["model"]["net"]["init_args"]
    a.update(tile_size=[64, 64], patch_size=8, embed_dim=96, depth=2, num_heads=3, adaptive_patching=True, fixed_length=16, use_adaptive_pos_emb=True)
    cfg["data"]["num_classes"] = 3
    cfg["data"]["batch_size"] = 4
    cfg["model"]["lr"] = 1e-3
    cfg["model"]["warmup_steps"] = 2
    cfg["load_balancing"]["batches_per_rank_epoch"]["catsdogs"] = 6
    out = _run_entry("train_sap_simple.py", cfg, tmp_path, 29582)
    losses = [float(l.split("epoch_loss")[1].split()[0]) for l in out.splitlines() if "epoch_loss" in l]
    assert len(losses) == 2 and all(math.isfinite(v) for v in losses) and losses[1] < losses[0], out
    a.update(tile_size=[32, 32, 32], patch_size=4, twoD=False, fixed_length=8, feature_size=4, depth=4, allow_torch_decoder=True)
    cfg["data"]["single_channel"] = True
    cfg["data"]["batch_size"] = 2
    cfg["load_balancing"]["batches_per_rank_epoch"]["catsdogs"] = 3
    out = _run_entry("train_unetr_simple.py", cfg, tmp_path, 29583)
    losses = [float(l.split("epoch_loss")[1].split()[0]) for l in out.splitlines() if "epoch_loss" in l]
    assert len(losses) == 2 and all(math.isfinite(v) for v in losses), out


def test_train_unetr_simple_entry_point_runs(tmp_path):
    """train_unetr_simple.py-compatible entry script (BASELINE configs[4] path, one GPU): 3-D volumes, tapped ViT encoder on the HIP
    kernels + conv decoder, Dice+CE loss; 2 epochs on synthetic data"""
    cfg = _smoke_cfg()
    a = cfg["model"]["net"]["init_args"]
    a.update(tile_size=[32, 32, 32], patch_size=8, embed_dim=96, depth=4, num_heads=3, twoD=False, feature_size=8, allow_torch_decoder=True)
    cfg["data"]["num_classes"] = 3
    cfg["data"]["batch_size"] = 2
    cfg["data"]["single_channel"] = True
    cfg["load_balancing"]["batches_per_rank_epoch"]["catsdogs"] = 3
    out = _run_entry("train_unetr_simple.py", cfg, tmp_path, 29579)
    losses = [float(l.split("epoch_loss")[1].split()[0]) for l in out.splitlines() if "epoch_loss" in l]
    assert len(losses) == 2 and all(math.isfinite(v) for v in losses), out


def test_empty_batch_forward_returns_empty_logits():
    """edge case: a batch of zero images goes through every forward kernel entry point (im2col, GEMMs, LayerNorm, attention,
    token assembly, adaptive front end) without a launch and yields [0, classes] like the torch reference does"""
    from UCF_VIT.simple.arch import VIT
    for dtype in (torch.float32, torch.bfloat16):
        m = build(VIT, VIT_KW, 21, dtype).eval()
        with torch.no_grad():
            out = m(torch.empty(0, 3, 32, 32, device=DEV), VARS)
        assert tuple(out.shape) == (0, 5) and out.dtype == dtype
        ma = VIT(num_classes=5, embed_dim=64, depth=2, num_heads=2, adaptive_patching=True, fixed_length=12, img_size=[32, 32], patch_size=8,
                 in_chans=3, use_adaptive_pos_emb=True).to(DEV).eval()
        ma.set_compute_dtype(dtype)
        with torch.no_grad():
            out = ma(torch.empty(0, 3, 12, 64, device=DEV), VARS, torch.empty(0, 12, 3, device=DEV))
        assert tuple(out.shape) == (0, 5)


# ---------------------------------------------------------------------------------------------- checkpoint / resume
def _resume_conf(tmp_path, load=None):
    return {"trainer": {"checkpoint_path": str(tmp_path), "checkpoint_filename": "ck", "resume_from_checkpoint": load is not None,
                        "checkpoint_filename_for_loading": load}}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_save_resume_continues_bit_exactly(tmp_path, dtype):
    """train 3 steps, save_checkpoint (reference dictionary layout, train_class_simple.py:364-388), build a FRESH model + optimizer +
    scheduler, maybe_resume (weights-only loader), and the next 2 steps are bit-equal to the uninterrupted 5-step run.  The saved
    optimizer state is the torch.optim.AdamW layout: per-parameter {step, exp_avg, exp_avg_sq}, nothing else."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ucf-vit_amd", "training_scripts"))
    import torch.distributed as dist
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    from UCF_VIT.utils.misc import configure_optimizer, configure_scheduler
    from _common import maybe_resume, save_checkpoint
    own_pg = not dist.is_initialized()
    if own_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29631")
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        g = torch.Generator().manual_seed(5)
        xs = [torch.randn(4, 3, 32, 32, generator=g).to(DEV) for _ in range(5)]
        ys = [torch.randint(0, 5, (4,), generator=g).to(DEV) for _ in range(5)]

        def fresh():
            m = build(VIT, VIT_KW, 31, dtype)
            opt = configure_optimizer(m, 1e-3, 0.9, 0.95, 1e-2)
            sch = configure_scheduler(opt, 2, 10, 1e-8, 1e-8)
            return m, opt, sch

        def steps(m, opt, sch, lo, hi):
            out = []
            for i in range(lo, hi):
                loss = cross_entropy_loss(m(xs[i], VARS), ys[i])
                loss.backward()
                opt.step()
                opt.zero_grad()
                sch.step()
                out.append(loss.item())
            return out

        m, opt, sch = fresh()
        la = steps(m, opt, sch, 0, 5)
        pa = m._ucf_store.flat_p.detach().clone()

        m, opt, sch = fresh()
        lb = steps(m, opt, sch, 0, 3)
        sd = opt.state_dict()
        assert set(sd["state"].keys()) == set(range(len(list(m.parameters()))))       # integer parameter indices only: no '_flat'
        for ent in sd["state"].values():
            assert set(ent.keys()) == {"step", "exp_avg", "exp_avg_sq"}
        save_checkpoint(_resume_conf(tmp_path), 0, m, opt, sch, lb, 0)
        ck = os.path.join(str(tmp_path), "ck_even.ckpt")
        size_one = os.path.getsize(ck)
        m2, opt2, sch2 = fresh()
        start, ll = maybe_resume(_resume_conf(tmp_path, "ck_even"), m2, opt2, sch2)
        assert start == 1 and ll == lb
        lb2 = steps(m2, opt2, sch2, 3, 5)
        assert lb + lb2 == la
        assert torch.equal(m2._ucf_store.flat_p, pa)
        # a second save after the resume does not grow (no dead copies of the moment buffers ride along)
        save_checkpoint(_resume_conf(tmp_path), 0, m2, opt2, sch2, la, 0)
        assert os.path.getsize(ck) < 1.05 * size_one
    finally:
        if own_pg:
            dist.destroy_process_group()


def test_optimizer_loads_a_torch_adamw_state_dict():
    """the reference's checkpoints hold torch.optim.AdamW state (utils/misc.py:67-82): it loads into the fused optimizer and the next
    step equals torch's own"""
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.utils.metrics import cross_entropy_loss
    from UCF_VIT.utils.misc import configure_optimizer
    from UCF_VIT._hip.params import is_no_decay
    g = torch.Generator().manual_seed(6)
    x, y = torch.randn(4, 3, 32, 32, generator=g).to(DEV), torch.randint(0, 5, (4,), generator=g).to(DEV)
    m = build(VIT, VIT_KW, 32)
    named = list(m.named_parameters())
    groups = [dict(params=[p for n, p in named if not is_no_decay(n)], weight_decay=1e-2),
              dict(params=[p for n, p in named if is_no_decay(n)], weight_decay=0.0)]
    topt = torch.optim.AdamW(groups, lr=1e-3, betas=(0.9, 0.95))
    for _ in range(2):
        cross_entropy_loss(m(x, VARS), y).backward()
        topt.step()
        topt.zero_grad()
    import copy
    sd = copy.deepcopy(topt.state_dict())            # (a .cpu() of the host-resident `step` tensors would alias them)
    sd["state"] = {i: {k: v.cpu() for k, v in e.items()} for i, e in sd["state"].items()}
    w_before = {n: p.detach().clone() for n, p in named}
    grads = None
    # torch's third step
    cross_entropy_loss(m(x, VARS), y).backward()
    grads = {n: p.grad.detach().clone() for n, p in named}
    topt.step()
    want = {n: p.detach().clone() for n, p in named}
    # rewind the weights, load the 2-step state into the fused optimizer, take the same third step
    with torch.no_grad():
        for n, p in named:
            p.copy_(w_before[n])
    hopt = configure_optimizer(m, 1e-3, 0.9, 0.95, 1e-2)
    hopt.load_state_dict(sd)
    for n, p in named:
        p.grad = None
    cross_entropy_loss(m(x, VARS), y).backward()
    hopt.step()
    for n, p in named:
        assert rel_err(p.detach(), want[n]) < 1e-6, n


# ---------------------------------------------------------------------------------------------- SURVEY §8f row 4 remainder
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_activation_checkpointing_of_blocks_is_exact_and_saves_memory(dtype):
    """apply_activation_checkpointing (the reference wraps every Block, train_masked_fsdp.py:393-396): the Blocks keep only their input and
    re-run their forward launches in backward; logits, loss and EVERY gradient are bit-identical to the plain run, and the activations
    held between forward and backward shrink."""
    from UCF_VIT.simple.arch import VIT
    from UCF_VIT.simple.building_blocks import apply_activation_checkpointing
    from UCF_VIT.utils.metrics import cross_entropy_loss
    kw = dict(img_size=[64, 64], patch_size=8, in_chans=3, num_classes=7, embed_dim=128, depth=6, num_heads=4)
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(16, 3, 64, 64, generator=g).to(DEV), torch.randint(0, 7, (16,), generator=g).to(DEV)

    def run(ckpt):
        m = build(VIT, kw, 41, dtype)
        if ckpt:
            assert apply_activation_checkpointing(m) == 6
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        out = m(x, VARS)
        loss = cross_entropy_loss(out, y)
        torch.cuda.synchronize()
        held = torch.cuda.memory_allocated() - base
        loss.backward()
        return out.detach(), loss.item(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, held

    o0, l0, g0, held0 = run(False)
    o1, l1, g1, held1 = run(True)
    assert torch.equal(o0, o1) and l0 == l1
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    assert held1 < 0.45 * held0, (held0, held1)


def test_mae_encoder_transfers_into_unetr_and_tp_checkpoint_names(tmp_path):
    """train_unetr_simple.py:328-340: the pretrained MAE's encoder entries (no 'decoder', no 'mask_token') overwrite the UNETR's, its conv
    decoder keeps its initialisation; train_masked_fsdp.py:624-644: one checkpoint per tensor-parallel rank, `_even_rank_<r>.ckpt`"""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ucf-vit_amd", "training_scripts"))
    from UCF_VIT.simple.arch import MAE, UNETR
    from UCF_VIT.utils.misc import configure_optimizer, configure_scheduler
    from _common import load_pretrained_mae_encoder, save_checkpoint_tp
    enc = dict(img_size=[32, 32, 16], patch_size=8, in_chans=1, embed_dim=96, depth=4, num_heads=3, class_token=False, twoD=False)
    mae = MAE(weight_init='skip', mask_ratio=0.75, linear_decoder=False, decoder_depth=1, decoder_embed_dim=48, decoder_num_heads=3,
              mlp_ratio_decoder=4.0, **enc)
    mae.load_state_dict(det_state_dict(mae, 91))
    un = UNETR(num_classes=4, linear_decoder=False, feature_size=4, skip_connection=True, allow_torch_decoder=True, **enc)
    before = {k: v.clone() for k, v in un.state_dict().items()}
    sd = {"module." + k: v for k, v in mae.state_dict().items()}             # a DDP-saved checkpoint
    copied = load_pretrained_mae_encoder(un, sd)
    after = un.state_dict()
    assert copied and all("decoder" not in k and "mask_token" not in k for k in copied)
    for k in copied:
        assert torch.equal(after[k], mae.state_dict()[k]), k
    for k in after:
        if k not in copied:
            assert torch.equal(after[k], before[k]), k                       # conv decoder untouched
    assert any(k.startswith("blocks.3.") for k in copied) and any(k.startswith("decoder5.") for k in after if k not in copied)
    un = un.to(DEV)
    opt = configure_optimizer(un, 1e-3, 0.9, 0.95, 0.0)
    sch = configure_scheduler(opt, 2, 10, 1e-8, 1e-8)
    conf = {"trainer": {"checkpoint_path": str(tmp_path), "checkpoint_filename": "hy"}}
    for r in range(3):
        save_checkpoint_tp(conf, 2, un, opt, sch, [0.5], world_rank=r, tensor_par_size=2)
    assert sorted(os.listdir(tmp_path)) == ["hy_even_rank_0.ckpt", "hy_even_rank_1.ckpt"]
    ck = torch.load(os.path.join(tmp_path, "hy_even_rank_1.ckpt"), map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "loss_list"} and ck["epoch"] == 2

"""Generates the adaptive-patching fixtures (tests/golden/model_vit_adaptive*.npz) from the REFERENCE implementation; run only in
the build container:

    cd /root/repo/tests/golden && python make_golden_adaptive.py

Same recipe as make_golden.py (reference imported with the _ref_standins stand-ins, deterministic PCG64 weights): the reference's
VIT(adaptive_patching=True) is fed x [B, C, S, P] (S already cut, resized patches) and seq_ps [B, S, 3|4] (position and size per
token) and its logits, loss and every parameter gradient are recorded.  Data only; no reference source text is stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_standins  # noqa: E402

_ref_standins.install()
from det_weights import det_state_dict, det_tensor  # noqa: E402

from UCF_VIT.simple.arch import MAE, SAP, VIT  # noqa: E402  (reference)
from UCF_VIT.utils.fused_attn import FusedAttn  # noqa: E402

torch.set_num_threads(4)
torch.manual_seed(0)
labels = torch.tensor([1, 3])


def seq_ps_of(B, S, kin, img, seed):
    """(x, y[, z], size) of each token as the quadtree patcher emits them: integer pixel positions and power-of-two sizes"""
    rng = np.random.Generator(np.random.PCG64(seed))
    pos = rng.integers(0, img, (B, S, kin - 1)).astype(np.float32)
    size = (2 ** rng.integers(1, 5, (B, S, 1))).astype(np.float32)
    return torch.from_numpy(np.concatenate([pos, size], axis=2))


def case(name, kw, x, seq_ps, seed):
    model = VIT(**kw)
    model.load_state_dict(det_state_dict(model, seed, keep=()))          # pos_embed is a random table here: make it deterministic too
    model.train()
    out = model(x, ["red", "green", "blue"], seq_ps)
    lab = labels
    if out.dim() == 3:                    # class_token=False: the head sees every token (arch.py:90-99); per-token labels
        lab = torch.arange(out.shape[0] * out.shape[1]) % out.shape[2]
        loss = torch.nn.CrossEntropyLoss()(out.flatten(0, 1), lab)
    else:
        loss = torch.nn.CrossEntropyLoss()(out, lab)
    loss.backward()
    rec = dict(x=x, seq_ps=seq_ps, logits=out, loss=loss, labels=lab)
    for k, p in model.named_parameters():
        rec["g." + k] = p.grad if p.grad is not None else torch.zeros_like(p)
    out_np = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in rec.items()}
    np.savez_compressed(os.path.join(HERE, name), **out_np)
    print(name, {k: v.shape for k, v in out_np.items() if not k.startswith("g.")}, float(loss))


B, S = 2, 12
kw2 = dict(img_size=[32, 32], patch_size=8, in_chans=3, num_classes=5, embed_dim=64, depth=2, num_heads=2, adaptive_patching=True,
           fixed_length=S, FusedAttn_option=FusedAttn.NONE)
x2 = det_tensor((B, 3, S, 64), 50)
case("model_vit_adaptive.npz", dict(kw2, use_adaptive_pos_emb=True), x2, seq_ps_of(B, S, 3, 32, 51), 52)
case("model_vit_adaptive_learnpos.npz", dict(kw2, use_adaptive_pos_emb=False), x2, seq_ps_of(B, S, 3, 32, 51), 53)
# 3-D volume, one channel, p = 4 -> P = 64, seq_ps = (x, y, z, size)
kw3 = dict(img_size=[16, 16, 16], patch_size=4, in_chans=1, num_classes=5, embed_dim=64, depth=2, num_heads=2, adaptive_patching=True,
           fixed_length=S, twoD=False, use_adaptive_pos_emb=True, FusedAttn_option=FusedAttn.NONE)
case("model_vit_adaptive_3d.npz", kw3, det_tensor((B, 1, S, 64), 54), seq_ps_of(B, S, 4, 16, 55), 56)


def mae_case(name, kw, x, seq_ps, noise, seed):
    model = MAE(**kw)
    model.load_state_dict(det_state_dict(model, seed, keep=()))
    model.train()
    orig = model.random_masking          # MAE.forward_features calls random_masking(x) without noise (arch.py:741): inject ours
    model.random_masking = lambda s, noise_=None: orig(s, noise)
    pred, mask = model(x, ["red", "green", "blue"], seq_ps)
    target = x.permute(0, 2, 3, 1).flatten(2)                                # rearrange 'b c s p -> b s (p c)' (train_masked_simple.py:29)
    loss = torch.nn.MSELoss()(pred, target)
    loss.backward()
    rec = dict(x=x, seq_ps=seq_ps, noise=noise, pred=pred, mask=mask, loss=loss,
               loss_masked=(((pred - target) ** 2).mean(-1) * mask).sum() / mask.sum())
    for k, p in model.named_parameters():
        rec["g." + k] = p.grad if p.grad is not None else torch.zeros_like(p)
    out_np = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in rec.items()}
    np.savez_compressed(os.path.join(HERE, name), **out_np)
    print(name, {k: v.shape for k, v in out_np.items() if not k.startswith("g.")}, loss.item())


mae_kw = dict(img_size=[32, 32], patch_size=8, in_chans=3, embed_dim=64, depth=2, num_heads=2, adaptive_patching=True, fixed_length=S,
              class_token=False, weight_init='skip', mask_ratio=0.5, linear_decoder=False, decoder_depth=1, decoder_embed_dim=32,
              decoder_num_heads=1, mlp_ratio_decoder=4.0, FusedAttn_option=FusedAttn.NONE)
noise = torch.from_numpy(np.random.Generator(np.random.PCG64(57)).random((B, S)).astype(np.float32))
mae_case("model_mae_adaptive.npz", dict(mae_kw, use_adaptive_pos_emb=True), x2, seq_ps_of(B, S, 3, 32, 51), noise, 58)
mae_case("model_mae_adaptive_learnpos.npz", dict(mae_kw, use_adaptive_pos_emb=False), x2, seq_ps_of(B, S, 3, 32, 51), noise, 59)

# ---- sqrt_len_method (what train_unetr_simple.py / train_sap_simple.py use with adaptive patching): the token sequence is reshaped
# into a pseudo image of (sqrt_len * p)^nd pixels and goes through the ordinary patch-embedding convolution; positions from seq_ps.
kw_sq = dict(img_size=[16, 16, 16], patch_size=4, in_chans=1, num_classes=5, embed_dim=96, depth=2, num_heads=3, adaptive_patching=True,
             fixed_length=8, twoD=False, use_adaptive_pos_emb=True, sqrt_len_method=True, class_token=False, FusedAttn_option=FusedAttn.NONE)
case("model_vit_sqrtlen_3d.npz", kw_sq, det_tensor((B, 1, 8, 8, 8), 60), seq_ps_of(B, 8, 4, 16, 61), 62)


def sap_case(name, kw, x, seq_ps, seed):
    model = SAP(**kw)
    model.load_state_dict(det_state_dict(model, seed, keep=()))
    model.train()
    out = model(x, ["red", "green", "blue"], seq_ps)
    target = det_tensor(out.shape, seed + 1)
    loss = torch.nn.MSELoss()(out, target)
    loss.backward()
    rec = dict(x=x, seq_ps=seq_ps, out=out, target=target, loss=loss)
    for k, p in model.named_parameters():
        rec["g." + k] = p.grad if p.grad is not None else torch.zeros_like(p)
    out_np = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in rec.items()}
    np.savez_compressed(os.path.join(HERE, name), **out_np)
    print(name, {k: v.shape for k, v in out_np.items() if not k.startswith("g.")}, loss.item())


# SAP as train_sap_simple.py builds it for adaptive input (:231-250): class_token=False, weight_init='skip', sqrt_len_method=True
kw_sap = dict(img_size=[64, 64], patch_size=8, in_chans=3, num_classes=3, embed_dim=64, depth=2, num_heads=2, adaptive_patching=True,
              fixed_length=16, sqrt_len=4, twoD=True, use_adaptive_pos_emb=True, sqrt_len_method=True, class_token=False, weight_init='skip',
              FusedAttn_option=FusedAttn.NONE)
sap_case("model_sap_adaptive.npz", kw_sap, det_tensor((B, 3, 32, 32), 63), seq_ps_of(B, 16, 3, 64, 64), 65)
print("done")

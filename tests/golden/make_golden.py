"""Generates tests/golden/*.npz from the REFERENCE implementation (run only in the build container):

    cd /root/repo/tests/golden && python make_golden.py

Imports /root/reference/src/UCF_VIT with stand-ins for the absent timm / monai helper symbols (_ref_standins.py;
SURVEY.md §8c) and records inputs + outputs + gradients of the hot-path operators and of small end-to-end models.
The fixtures are data only; no reference source text is stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_standins  # noqa: E402

_ref_standins.install()
from det_weights import det_state_dict, det_tensor, proj_vector  # noqa: E402

from UCF_VIT.simple.arch import VIT, MAE  # noqa: E402  (reference)
from UCF_VIT.simple.building_blocks import Attention, Block, Mlp, PatchEmbed  # noqa: E402
from UCF_VIT.utils.fused_attn import FusedAttn  # noqa: E402
from UCF_VIT.utils.pos_embed import get_2d_sincos_pos_embed, get_3d_sincos_pos_embed  # noqa: E402
from UCF_VIT.utils.lr_scheduler import LinearWarmupCosineAnnealingLR  # noqa: E402

torch.set_num_threads(4)
torch.manual_seed(0)


def npz(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, {k: v.shape for k, v in out.items()})


def grads_of(module):
    return {"g." + k: p.grad for k, p in module.named_parameters()}


# ------------------------------------------------------------------ G1: operators
def op_case(name, mod, x, seed):
    mod.load_state_dict(det_state_dict(mod, seed))
    x = x.clone().requires_grad_(True)
    y = mod(x)
    gy = det_tensor(y.shape, seed + 50)
    y.backward(gy)
    rec = {"x": x, "y": y, "gy": gy, "gx": x.grad}
    rec.update({"w." + k: v for k, v in mod.state_dict().items()})
    rec.update(grads_of(mod))
    npz(name, **rec)


B, N, D, H = 2, 17, 64, 2
op_case("op_mlp.npz", Mlp(in_features=D, hidden_features=4 * D), det_tensor((B, N, D), 1), 11)
op_case("op_attn_none.npz", Attention(D, fused_attn=FusedAttn.NONE, num_heads=H, qkv_bias=True), det_tensor((B, N, D), 2), 12)
op_case("op_attn_default.npz", Attention(D, fused_attn=FusedAttn.DEFAULT, num_heads=H, qkv_bias=True), det_tensor((B, N, D), 2), 12)
op_case("op_attn_n197_dh64.npz", Attention(128, fused_attn=FusedAttn.NONE, num_heads=2, qkv_bias=True), det_tensor((1, 197, 128), 3), 13)
from functools import partial  # noqa: E402
op_case("op_block.npz", Block(D, H, fused_attn=FusedAttn.NONE, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6)),
        det_tensor((B, N, D), 4), 14)
op_case("op_layernorm.npz", torch.nn.LayerNorm(D, eps=1e-6), det_tensor((B, N, D), 5, 2.0, 0.5), 15)

pe2 = PatchEmbed(img_size=[32, 32], patch_size=8, in_chans=3, embed_dim=D, twoD=True)
pe2.load_state_dict(det_state_dict(pe2, 16))
x = det_tensor((B, 3, 32, 32), 6)
y = pe2(x)
gy = det_tensor(y.shape, 66)
y.backward(gy)
npz("op_patch2d.npz", x=x, y=y, gy=gy, **{"w." + k: v for k, v in pe2.state_dict().items()}, **grads_of(pe2))

pe3 = PatchEmbed(img_size=[16, 16, 8], patch_size=4, in_chans=1, embed_dim=48, twoD=False)
pe3.load_state_dict(det_state_dict(pe3, 17))
x = det_tensor((B, 1, 16, 16, 8), 7)
y = pe3(x)
gy = det_tensor(y.shape, 67)
y.backward(gy)
npz("op_patch3d.npz", x=x, y=y, gy=gy, **{"w." + k: v for k, v in pe3.state_dict().items()}, **grads_of(pe3))

# ------------------------------------------------------------------ G2: position tables
npz("pos_tables.npz",
    t2d_32_3x5_cls=get_2d_sincos_pos_embed(32, 3, 5, cls_token=True),
    t2d_64_4x4=get_2d_sincos_pos_embed(64, 4, 4, cls_token=False),
    t3d_48_2x3x4=get_3d_sincos_pos_embed(48, 2, 3, 4),
    vitl_sum=np.array([get_2d_sincos_pos_embed(1024, 14, 14, cls_token=True).sum(),
                       np.abs(get_2d_sincos_pos_embed(1024, 14, 14, cls_token=True)).sum()]))

# ------------------------------------------------------------------ G3: MAE random masking (bit-exact)
mae_kw = dict(img_size=[32, 32], patch_size=8, in_chans=3, embed_dim=D, depth=2, num_heads=H, class_token=False, weight_init='skip',
              mask_ratio=0.75, linear_decoder=False, decoder_depth=1, decoder_embed_dim=32, decoder_num_heads=1, mlp_ratio_decoder=4.0,
              FusedAttn_option=FusedAttn.NONE)
mae = MAE(**mae_kw)
seq = det_tensor((2, 196, 32), 8)
noise = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).random((2, 196)).astype(np.float32))
kept, mask, ids_restore = mae.random_masking(seq, noise)
npz("mae_masking.npz", seq=seq, noise=noise, kept=kept, mask=mask, ids_restore=ids_restore)

# ------------------------------------------------------------------ G4: models
def model_case(name, model, x, fwd, seed, full_grads):
    model.load_state_dict(det_state_dict(model, seed))
    model.train()
    outs = fwd(model, x)
    loss = outs["loss"]
    loss.backward()
    rec = dict(x=x, **{k: v for k, v in outs.items()})
    for i, (k, p) in enumerate(model.named_parameters()):
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        if full_grads:
            rec["g." + k] = g
        else:
            rec["gn." + k] = g.double().norm()
            rec["gp." + k] = (g.double() * proj_vector(g.shape, i).double()).sum()
    npz(name, **rec)


labels2 = torch.tensor([1, 3])


def fwd_vit(m, x, labels=labels2):
    out = m(x, ["red", "green", "blue"])
    return dict(logits=out, loss=torch.nn.CrossEntropyLoss()(out, labels), labels=labels)


vit_kw = dict(img_size=[32, 32], patch_size=8, in_chans=3, num_classes=5, embed_dim=D, depth=2, num_heads=H, FusedAttn_option=FusedAttn.NONE)
model_case("model_vit_small.npz", VIT(**vit_kw), det_tensor((2, 3, 32, 32), 20), fwd_vit, 21, True)

# config T (BASELINE configs[0]): ViT-Tiny/16, catsdogs shape 256x256, 2 classes, B=2 for the fixture
vit_t = VIT(img_size=[256, 256], patch_size=16, in_chans=3, num_classes=2, embed_dim=192, depth=12, num_heads=3, FusedAttn_option=FusedAttn.DEFAULT)
xT = torch.from_numpy(np.random.Generator(np.random.PCG64(22)).integers(0, 256, (2, 3, 256, 256)).astype(np.float32))
model_case("model_vit_tiny_catsdogs.npz", vit_t, xT, lambda m, x: fwd_vit(m, x, torch.tensor([0, 1])), 23, False)


def patchify(data, p):
    B_, C_ = data.shape[0], data.shape[1]
    g = data.shape[2] // p
    return torch.einsum("nchpwq->nhwpqc", data.reshape(B_, C_, g, p, g, p)).reshape(B_, g * g, p * p * C_)


noise_m = torch.from_numpy(np.random.Generator(np.random.PCG64(24)).random((2, 16)).astype(np.float32))


def fwd_mae(m, x):
    # MAE.forward_features calls random_masking(x) without noise (arch.py:741): inject ours through the method
    orig = m.random_masking
    m.random_masking = lambda s, noise=None: orig(s, noise_m)
    pred, mask = m(x, ["red", "green", "blue"])
    m.random_masking = orig
    tgt = patchify(x, 8)
    return dict(pred=pred, mask=mask, noise=noise_m, loss=torch.nn.MSELoss()(pred, tgt),
                loss_masked=(((pred - tgt) ** 2).mean(-1) * mask).sum() / mask.sum())


model_case("model_mae_small.npz", MAE(**mae_kw), det_tensor((2, 3, 32, 32), 25), fwd_mae, 26, True)

# ------------------------------------------------------------------ G5: lr schedule
p = torch.nn.Parameter(torch.zeros(1))
opt = torch.optim.AdamW([p], lr=1e-4)
sch = LinearWarmupCosineAnnealingLR(opt, 5, 20, 1e-8, 1e-8)
lrs = []
for _ in range(30):
    lrs.append(opt.param_groups[0]["lr"])
    opt.step()
    sch.step()
opt2 = torch.optim.AdamW([p], lr=1e-4)
sch2 = LinearWarmupCosineAnnealingLR(opt2, 1000, 20000, 1e-8, 1e-8)
lrs2 = []
for _ in range(40):
    lrs2.append(opt2.param_groups[0]["lr"])
    opt2.step()
    sch2.step()
npz("lr_schedule.npz", lrs_5_20=np.array(lrs, dtype=np.float64), lrs_1000_20000=np.array(lrs2, dtype=np.float64))

# ------------------------------------------------------------------ G6: 5-step AdamW trajectory (config of configs/catsdogs)
model = VIT(**vit_kw)
model.load_state_dict(det_state_dict(model, 31))
decay, no_decay = [], []
for n_, p_ in model.named_parameters():
    (no_decay if ("var_embed" in n_ or "pos_embed" in n_ or "time_pos_embed" in n_) else decay).append(p_)
opt = torch.optim.AdamW([{"params": decay, "lr": 1e-3, "betas": (0.9, 0.95), "weight_decay": 1e-2},
                         {"params": no_decay, "lr": 1e-3, "betas": (0.9, 0.95), "weight_decay": 0}])
sch = LinearWarmupCosineAnnealingLR(opt, 2, 10, 1e-5, 1e-6)
losses = []
xs = [det_tensor((2, 3, 32, 32), 40 + i) for i in range(5)]
ys = [torch.tensor([i % 5, (i + 2) % 5]) for i in range(5)]
for i in range(5):
    out = model(xs[i], ["red", "green", "blue"])
    loss = torch.nn.CrossEntropyLoss()(out, ys[i])
    losses.append(loss.item())
    loss.backward()
    opt.step()
    opt.zero_grad()
    sch.step()
npz("traj_vit_small.npz", losses=np.array(losses), labels=torch.stack(ys), **{"x%d" % i: xs[i] for i in range(5)},
    **{"final." + k: v for k, v in model.state_dict().items() if not k.startswith("token_embeds")})
print("done")

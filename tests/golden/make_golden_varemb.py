"""Generates tests/golden/model_vit_varemb*.npz from the REFERENCE implementation (run only in the build container):

    cd /root/repo/tests/golden && python make_golden_varemb.py

VIT(use_varemb=True): every input channel is tokenised on its own (one shared 1-channel patch embedding), gets its variable embedding,
and the channels of a token are aggregated by a cross-attention with a learnt query (VariableMapping_Attention).  Same recipe as
make_golden.py (reference imported with the _ref_standins stand-ins, deterministic PCG64 weights).  Data only.
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

from UCF_VIT.simple.arch import MAE, VIT  # noqa: E402  (reference)
from UCF_VIT.utils.fused_attn import FusedAttn  # noqa: E402

torch.set_num_threads(4)
torch.manual_seed(0)
labels = torch.tensor([1, 3])
DEFAULT_VARS = ["u", "v", "t", "q"]


def case(name, kw, x, variables, seq_ps, seed):
    model = VIT(**kw)
    model.load_state_dict(det_state_dict(model, seed, keep=()))
    model.train()
    out = model(x, variables, seq_ps)
    loss = torch.nn.CrossEntropyLoss()(out, labels)
    loss.backward()
    rec = dict(x=x, seq_ps=seq_ps, logits=out, loss=loss, labels=labels)
    for k, p in model.named_parameters():
        rec["g." + k] = p.grad if p.grad is not None else torch.zeros_like(p)
    out_np = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in rec.items()}
    np.savez_compressed(os.path.join(HERE, name), **out_np)
    print(name, variables, {k: v.shape for k, v in out_np.items() if not k.startswith("g.")}, loss.item())
    print("   params:", [k for k, _ in model.named_parameters() if "blocks" not in k])


# NOTE: with adaptive_patching=False the reference cannot construct a use_varemb model at all (arch.py:211-218 passes the misspelt keyword
# `sqrt_len_meth` to PatchEmbed -> TypeError), so variable aggregation is pinned on the path that runs: adaptively patched input, one
# LayerNorm-Linear-LayerNorm token embedding PER VARIABLE (arch.py:282-286), variable embedding, VariableMapping_Attention.
B, S = 2, 12
rng = np.random.Generator(np.random.PCG64(84))
seq_ps = torch.from_numpy(np.concatenate([rng.integers(0, 32, (B, S, 2)), 2 ** rng.integers(1, 5, (B, S, 1))], axis=2).astype(np.float32))
kw = dict(img_size=[32, 32], patch_size=8, in_chans=3, num_classes=5, embed_dim=64, depth=2, num_heads=2, use_varemb=True,
          default_vars=DEFAULT_VARS, single_channel=False, adaptive_patching=True, fixed_length=S, use_adaptive_pos_emb=True,
          FusedAttn_option=FusedAttn.NONE)
case("model_vit_varemb.npz", kw, det_tensor((B, 3, S, 64), 80), ["v", "q", "u"], seq_ps, 81)     # 3 of the 4 variables, out of order
case("model_vit_varemb_single.npz", dict(kw, in_chans=1, single_channel=True), det_tensor((B, 1, S, 64), 82), ["t"], seq_ps, 83)


def mae_case(name, kw, x, variables, seq_ps, noise, seed):
    model = MAE(**kw)
    model.load_state_dict(det_state_dict(model, seed, keep=()))
    model.train()
    orig = model.random_masking
    model.random_masking = lambda s_, noise_=None: orig(s_, noise)
    pred, mask = model(x, variables, seq_ps)
    # the reference's adaptive MAE target (train_masked_simple.py:29) has in_chans * p^2 entries per token like the prediction
    target = x.permute(0, 2, 3, 1).flatten(2)
    loss = torch.nn.MSELoss()(pred, target)
    loss.backward()
    rec = dict(x=x, seq_ps=seq_ps, noise=noise, pred=pred, mask=mask, loss=loss)
    for k, p in model.named_parameters():
        rec["g." + k] = p.grad if p.grad is not None else torch.zeros_like(p)
    out_np = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in rec.items()}
    np.savez_compressed(os.path.join(HERE, name), **out_np)
    print(name, variables, {k: v.shape for k, v in out_np.items() if not k.startswith("g.")}, loss.item())


mae_kw = dict(img_size=[32, 32], patch_size=8, in_chans=3, embed_dim=64, depth=2, num_heads=2, adaptive_patching=True, fixed_length=S,
              class_token=False, weight_init='skip', mask_ratio=0.5, linear_decoder=False, decoder_depth=1, decoder_embed_dim=32,
              decoder_num_heads=1, mlp_ratio_decoder=4.0, use_varemb=True, default_vars=DEFAULT_VARS, single_channel=False,
              use_adaptive_pos_emb=True, FusedAttn_option=FusedAttn.NONE)
noise = torch.from_numpy(np.random.Generator(np.random.PCG64(85)).random((B, S)).astype(np.float32))
mae_case("model_mae_varemb.npz", mae_kw, det_tensor((B, 3, S, 64), 86), ["q", "u", "t"], seq_ps, noise, 87)
print("done")

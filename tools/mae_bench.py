#!/usr/bin/env python3
"""MAE ViT-L/16 (mask ratio 0.75, 8-block 512-wide decoder) training-step throughput on one GPU (BASELINE.json configs[3] shape,
single GPU): python tools/mae_bench.py [B] [steps]"""
import os
import sys
import time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
from UCF_VIT.simple.arch import MAE  # noqa: E402
from UCF_VIT.utils.metrics import patch_mse_loss  # noqa: E402
from UCF_VIT.utils.misc import configure_optimizer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda"
torch.manual_seed(0)
m = MAE(img_size=[224, 224], patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16, class_token=False, weight_init='skip',
        mask_ratio=0.75, linear_decoder=False, decoder_depth=8, decoder_embed_dim=512, decoder_num_heads=16, mlp_ratio_decoder=4.0).to(dev)
m.set_compute_dtype(torch.bfloat16)
opt = configure_optimizer(m, 1e-4, 0.9, 0.95, 0.05)
x = torch.rand(B, 3, 224, 224, device=dev)
var = ["red", "green", "blue"]


def step():
    out, mask = m(x, var, None)
    loss = patch_mse_loss(out, x, 16, mask)
    loss.backward()
    opt.step()
    opt.zero_grad()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f"MAE ViT-L/16 mask 0.75 bf16 B={B}: {dt*1e3:.1f} ms/step, {B/dt:.0f} images/s, loss {loss.item():.4f}")

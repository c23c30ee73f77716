#!/usr/bin/env python3
"""SAP segmentation training on adaptively patched input — entry point compatible with the reference's
training_scripts/train_sap_simple.py:  python train_sap_simple.py <config.yaml> [MPI]
The ViT encoder (patch-embedding convolution on the pseudo image, position embedding from seq_ps, Blocks) runs on the HIP kernels;
the transposed-convolution neck, the 1x1 header and the Dice+BCE loss stay on torch / MIOpen."""
import sys

import torch

from _common import (SyntheticSeqLoader, StepTimer, init_distributed, iters_per_epoch, load_config, maybe_resume, model_args, save_checkpoint,
                     sqrt_len_of)
from UCF_VIT.simple.arch import SAP
from UCF_VIT.utils.fused_attn import FusedAttn
from UCF_VIT.utils.metrics import DiceBLoss
from UCF_VIT.utils.misc import configure_optimizer, configure_scheduler
from UCF_VIT._hip.ddp import HipDataParallel


def training_step_adaptive(seq, seq_label, variables, net, patch_size, twoD, num_classes, sqrt_len, seq_ps, in_chans):
    """reference :28-46 (plain reshape of the token sequence into the pseudo image, as there)"""
    side = (patch_size * sqrt_len,) * (2 if twoD else 3)
    seq = torch.reshape(seq, (-1, in_chans) + side)
    seq_label = torch.reshape(seq_label, (-1, num_classes) + side)
    output = net(seq, variables, seq_ps)
    return DiceBLoss(num_class=num_classes)(output, seq_label)


def main(device, local_rank, rank, world):
    conf = load_config(sys.argv[1])
    margs, a, d = model_args(conf)
    if not margs["adaptive_patching"]:
        raise ValueError("train_sap_simple.py trains on adaptively patched input (the reference script has no other path): set adaptive_patching True")
    m = conf["model"]
    sqrt_len = sqrt_len_of(margs["fixed_length"], margs["twoD"])
    nc = d["num_classes"]
    model = SAP(num_classes=nc, sqrt_len=sqrt_len, sqrt_len_method=True, class_token=False, weight_init='skip', FusedAttn_option=FusedAttn.HIP,
                **margs).to(device)
    model.set_compute_dtype(torch.bfloat16 if conf["trainer"].get("data_type", "float32") == "bfloat16" else torch.float32)
    net = HipDataParallel(model)
    optimizer = configure_optimizer(model, float(m["lr"]), float(m["beta_1"]), float(m["beta_2"]), float(m["weight_decay"]))
    scheduler = configure_scheduler(optimizer, int(m["warmup_steps"]), int(m["max_steps"]), float(m["warmup_start_lr"]), float(m["eta_min"]))
    epoch_start, loss_list = maybe_resume(conf, net, optimizer, scheduler)
    variables = d["dict_in_variables"][d["dataset"]]
    loader = SyntheticSeqLoader(d["batch_size"], margs["in_chans"], margs["img_size"], margs["patch_size"], margs["fixed_length"], nc,
                                iters_per_epoch(conf), device, 1234 + rank)
    for epoch in range(epoch_start, conf["trainer"]["max_epochs"]):
        model.train()
        epoch_loss = torch.zeros((), device=device)
        timer = StepTimer()
        for seq, seq_ps, _ in loader:
            # synthetic per-pixel labels from the data itself (brightness bands), one-hot over the classes, in sequence layout
            band = torch.clamp((seq.mean(dim=1, keepdim=True) / 256.0 * nc).long(), 0, nc - 1)
            seq_label = torch.zeros(seq.shape[0], nc, *seq.shape[2:], device=device).scatter_(1, band, 1.0)
            loss = training_step_adaptive(seq / 255.0, seq_label, variables, net, margs["patch_size"], margs["twoD"], nc, sqrt_len, seq_ps,
                                          margs["in_chans"])
            epoch_loss += loss.detach()
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            scheduler.step()
            timer.tick(seq.shape[0] * world)
        loss_list.append(epoch_loss)
        if rank == 0:
            print(f"epoch: {epoch} epoch_loss {epoch_loss.item():.4f} images/s {timer.rate():.1f}", flush=True)
        save_checkpoint(conf, epoch, net, optimizer, scheduler, loss_list, rank)


if __name__ == "__main__":
    dev, lr_, r, w = init_distributed(sys.argv[2] if len(sys.argv) > 2 else None)
    main(dev, lr_, r, w)
    torch.distributed.destroy_process_group()

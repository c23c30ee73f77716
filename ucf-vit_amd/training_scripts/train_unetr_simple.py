#!/usr/bin/env python3
"""UNETR segmentation training — entry point compatible with the reference's training_scripts/train_unetr_simple.py.
The ViT encoder, the convolutional decoder (3x3x3 / transposed / 1x1x1 convolutions on csrc/conv3d.hip, channels-last instance-norm /
LeakyReLU / residual chains on csrc/unetr_decoder.hip) and the Dice+CE loss (monai DiceCELoss(to_onehot_y, softmax, squared_pred) in the
reference, :38; parity unpinned: monai is not vendored) all run on the HIP kernels.  A configuration those kernels do not cover (2-D,
feature sizes other than 16 / 32 k, ...) needs `allow_torch_decoder: True` in model.net.init_args, else UNETR.forward raises."""
import sys

import torch
import torch.nn.functional as F

from _common import (SyntheticLoader, SyntheticSeqLoader, StepTimer, init_distributed, iters_per_epoch, load_config, maybe_resume, model_args,
                     save_checkpoint, seq_to_pseudo_image, sqrt_len_of)
from UCF_VIT.simple.arch import UNETR
from UCF_VIT.utils.fused_attn import FusedAttn
from UCF_VIT.utils.misc import configure_optimizer, configure_scheduler
from UCF_VIT._hip.ddp import HipDataParallel
from UCF_VIT._hip import functional as HF


def dice_ce_loss(logits, label, smooth=1e-5):
    """monai DiceCELoss(to_onehot_y, softmax, squared_pred) (reference :38): DiceLoss over softmax probabilities with the squared-prediction
    denominator, mean over batch and classes, + CrossEntropy — one fused HIP kernel pair (ucfvit_dice_ce: forward and gradient)"""
    return HF.dice_ce(logits.float(), label.view(label.shape[0], *label.shape[2:]) if label.dim() == logits.dim() else label, smooth, smooth)


def training_step(data, variables, label, net):
    output = net(data, variables, None, None)
    return dice_ce_loss(output, label), output


def training_step_adaptive(data, seq, label, variables, net, patch_size, twoD, seq_ps, sqrt_len):
    """reference :43-52: the full-resolution volume feeds the first conv encoder, the token sequence (as a pseudo volume) the ViT"""
    output = net(data, variables, seq_ps, seq_to_pseudo_image(seq, sqrt_len, patch_size, twoD))
    return dice_ce_loss(output, label), output


def main(device, local_rank, rank, world):
    conf = load_config(sys.argv[1])
    margs, a, d = model_args(conf)
    adaptive = margs["adaptive_patching"]
    sqrt_len = sqrt_len_of(margs["fixed_length"], margs["twoD"]) if adaptive else None
    m = conf["model"]
    model = UNETR(num_classes=d["num_classes"], class_token=False, weight_init='', linear_decoder=a.get("linear_decoder", False),
                  feature_size=a.get("feature_size", 16), skip_connection=a.get("skip_connection", True), sqrt_len=sqrt_len,
                  allow_torch_decoder=bool(a.get("allow_torch_decoder", False)),   # explicit opt-in for configurations the HIP decoder does not cover
                  sqrt_len_method=adaptive, FusedAttn_option=FusedAttn.HIP, **margs).to(device)
    model.set_compute_dtype(torch.bfloat16 if conf["trainer"].get("data_type", "float32") == "bfloat16" else torch.float32)
    net = HipDataParallel(model)
    optimizer = configure_optimizer(model, float(m["lr"]), float(m["beta_1"]), float(m["beta_2"]), float(m["weight_decay"]))   # PyYAML reads "1e-5" as str
    scheduler = configure_scheduler(optimizer, int(m["warmup_steps"]), int(m["max_steps"]), float(m["warmup_start_lr"]), float(m["eta_min"]))
    epoch_start, loss_list = maybe_resume(conf, net, optimizer, scheduler)
    variables = d["dict_in_variables"][d["dataset"]]
    loader = SyntheticLoader(d["batch_size"], margs["in_chans"], margs["img_size"], d["num_classes"], iters_per_epoch(conf), device,
                             1234 + rank, volumetric_labels=True)
    seq_loader = SyntheticSeqLoader(d["batch_size"], margs["in_chans"], margs["img_size"], margs["patch_size"], margs["fixed_length"], 0,
                                    iters_per_epoch(conf), device, 4321 + rank) if adaptive else None
    for epoch in range(epoch_start, conf["trainer"]["max_epochs"]):
        model.train()
        epoch_loss = torch.zeros((), device=device)
        timer = StepTimer()
        seqs = iter(seq_loader) if adaptive else None
        for data, label in loader:
            if adaptive:
                seq, seq_ps, _ = next(seqs)
                loss, _ = training_step_adaptive(data / 255.0, seq / 255.0, label, variables, net, margs["patch_size"], margs["twoD"], seq_ps, sqrt_len)
            else:
                loss, _ = training_step(data / 255.0, variables, label, net)       # basic_ct volumes are min-max normalised
            epoch_loss += loss.detach()
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            scheduler.step()
            timer.tick(data.shape[0] * world)
        loss_list.append(epoch_loss)
        if rank == 0:
            print(f"epoch: {epoch} epoch_loss {epoch_loss.item():.4f} volumes/s {timer.rate():.2f}", flush=True)
        save_checkpoint(conf, epoch, net, optimizer, scheduler, loss_list, rank)


if __name__ == "__main__":
    dev, lr_, r, w = init_distributed(sys.argv[2] if len(sys.argv) > 2 else None)
    main(dev, lr_, r, w)
    torch.distributed.destroy_process_group()

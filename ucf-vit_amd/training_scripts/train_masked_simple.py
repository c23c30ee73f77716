#!/usr/bin/env python3
"""MAE pre-training on the MI355X hot path — entry point compatible with the reference's
training_scripts/train_masked_simple.py:  python train_masked_simple.py <config.yaml> [MPI]"""
import sys

import torch

from _common import (SyntheticLoader, SyntheticSeqLoader, StepTimer, init_distributed, iters_per_epoch, load_config, maybe_resume, model_args,
                     save_checkpoint)
from UCF_VIT.simple.arch import MAE
from UCF_VIT.utils.fused_attn import FusedAttn
from UCF_VIT.utils.metrics import patch_mse_loss, seq_mse_loss
from UCF_VIT.utils.misc import configure_optimizer, configure_scheduler
from UCF_VIT._hip.ddp import HipDataParallel


def training_step(data, variables, net, patch_size, twoD, loss_fn):
    output, mask = net(data, variables, None)
    # target = patchify(data) is never materialised: the loss kernel reads the image in place (misc.py:14-33 layout)
    return patch_mse_loss(output, data, patch_size, mask if loss_fn == "maskMSE" else None), output, mask


def training_step_adaptive(seq, variables, net, loss_fn, seq_ps):
    """reference :24-32: target = rearrange(seq, 'b c s p -> b s (p c)'), read in place by the loss kernel"""
    output, mask = net(seq, variables, seq_ps)
    return seq_mse_loss(output, seq, mask if loss_fn == "maskMSE" else None), output, mask


def main(device, local_rank, rank, world):
    conf = load_config(sys.argv[1])
    margs, a, d = model_args(conf)
    m = conf["model"]
    model = MAE(class_token=False, weight_init='skip', mask_ratio=a["mask_ratio"], linear_decoder=a["linear_decoder"],
                decoder_depth=a["decoder_depth"], decoder_embed_dim=a["decoder_embed_dim"], decoder_num_heads=a["decoder_num_heads"],
                mlp_ratio_decoder=a["mlp_ratio_decoder"], FusedAttn_option=FusedAttn.HIP, **margs).to(device)
    model.set_compute_dtype(torch.bfloat16 if conf["trainer"].get("data_type", "float32") == "bfloat16" else torch.float32)
    net = HipDataParallel(model)
    optimizer = configure_optimizer(model, float(m["lr"]), float(m["beta_1"]), float(m["beta_2"]), float(m["weight_decay"]))   # PyYAML reads "1e-5" as str
    scheduler = configure_scheduler(optimizer, int(m["warmup_steps"]), int(m["max_steps"]), float(m["warmup_start_lr"]), float(m["eta_min"]))
    epoch_start, loss_list = maybe_resume(conf, net, optimizer, scheduler)
    variables = d["dict_in_variables"][d["dataset"]]
    loss_fn = conf["trainer"].get("loss_fn", a.get("loss_fn", "MSE"))
    if margs["adaptive_patching"]:
        seq_loader = SyntheticSeqLoader(d["batch_size"], margs["in_chans"], margs["img_size"], margs["patch_size"], margs["fixed_length"], 0,
                                        iters_per_epoch(conf), device, 1234 + rank)
        loader = lambda: ((seq, seq_ps) for seq, seq_ps, _ in seq_loader)
    else:
        img_loader = SyntheticLoader(d["batch_size"], margs["in_chans"], margs["img_size"], 0, iters_per_epoch(conf), device, 1234 + rank)
        loader = lambda: ((data, None) for data, _ in img_loader)
    for epoch in range(epoch_start, conf["trainer"]["max_epochs"]):
        model.train()
        epoch_loss = torch.zeros((), device=device)
        timer = StepTimer()
        for data, seq_ps in loader():
            if margs["adaptive_patching"]:
                loss, _, _ = training_step_adaptive(data, variables, net, loss_fn, seq_ps)
            else:
                loss, _, _ = training_step(data, variables, net, margs["patch_size"], margs["twoD"], loss_fn)
            epoch_loss += loss.detach()
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            scheduler.step()
            timer.tick(data.shape[0] * world)
        loss_list.append(epoch_loss)
        if rank == 0:
            print(f"epoch: {epoch} epoch_loss {epoch_loss.item():.4f} images/s {timer.rate():.1f}", flush=True)
        save_checkpoint(conf, epoch, net, optimizer, scheduler, loss_list, rank)


if __name__ == "__main__":
    dev, lr_, r, w = init_distributed(sys.argv[2] if len(sys.argv) > 2 else None)
    main(dev, lr_, r, w)
    torch.distributed.destroy_process_group()

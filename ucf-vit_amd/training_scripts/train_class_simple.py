#!/usr/bin/env python3
"""ViT classification training on the MI355X hot path — entry point compatible with the reference's
training_scripts/train_class_simple.py:  python train_class_simple.py <config.yaml> [MPI]
(same YAML schema; data is synthetic unless a loader is plugged in; trainer.data_type may be float32 or bfloat16)."""
import sys

import torch

from _common import (SyntheticLoader, SyntheticSeqLoader, StepTimer, init_distributed, iters_per_epoch, load_config, maybe_resume, model_args,
                     save_checkpoint)
from UCF_VIT.simple.arch import VIT
from UCF_VIT.utils.fused_attn import FusedAttn
from UCF_VIT.utils.metrics import cross_entropy_loss
from UCF_VIT.utils.misc import configure_optimizer, configure_scheduler
from UCF_VIT._hip.ddp import HipDataParallel


def training_step(data, variables, label, net, seq_ps):
    output = net(data, variables, seq_ps)
    return cross_entropy_loss(output, label), output


def main(device, local_rank, rank, world):
    conf = load_config(sys.argv[1])
    margs, a, d = model_args(conf)
    m = conf["model"]
    model = VIT(num_classes=d["num_classes"], drop_rate=a.get("drop_rate", 0.0), weight_init='', FusedAttn_option=FusedAttn.HIP, **margs).to(device)
    model.set_compute_dtype(torch.bfloat16 if conf["trainer"].get("data_type", "float32") == "bfloat16" else torch.float32)
    net = HipDataParallel(model)                                             # reference: DDP(model, find_unused_parameters=True)
    optimizer = configure_optimizer(model, float(m["lr"]), float(m["beta_1"]), float(m["beta_2"]), float(m["weight_decay"]))   # PyYAML reads "1e-5" as str
    scheduler = configure_scheduler(optimizer, int(m["warmup_steps"]), int(m["max_steps"]), float(m["warmup_start_lr"]), float(m["eta_min"]))
    epoch_start, loss_list = maybe_resume(conf, net, optimizer, scheduler)
    variables = d["dict_in_variables"][d["dataset"]]
    if margs["adaptive_patching"]:     # reference :322-337: the batch carries the token sequence and (size, position) per token
        seq_loader = SyntheticSeqLoader(d["batch_size"], margs["in_chans"], margs["img_size"], margs["patch_size"], margs["fixed_length"],
                                        d["num_classes"], iters_per_epoch(conf), device, 1234 + rank)
        loader = lambda: ((seq, label, seq_ps) for seq, seq_ps, label in seq_loader)
    else:
        img_loader = SyntheticLoader(d["batch_size"], margs["in_chans"], margs["img_size"], d["num_classes"], iters_per_epoch(conf), device, 1234 + rank)
        loader = lambda: ((data, label, None) for data, label in img_loader)
    for epoch in range(epoch_start, conf["trainer"]["max_epochs"]):
        model.train()
        epoch_loss = torch.zeros((), device=device)
        epoch_acc = torch.zeros((), device=device)
        timer = StepTimer()
        for batch_idx, (data, label, seq_ps) in enumerate(loader()):
            loss, output = training_step(data, variables, label, net, seq_ps)
            epoch_acc += (output.argmax(dim=1) == label).float().mean()
            epoch_loss += loss.detach()
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            scheduler.step()
            timer.tick(data.shape[0] * world)
        loss_list.append(epoch_loss)
        if rank == 0:   # one host sync per epoch (the reference prints every iteration, a sync per step)
            print(f"epoch: {epoch} epoch_loss {epoch_loss.item():.4f} epoch_accuracy {epoch_acc.item():.4f} images/s {timer.rate():.1f}", flush=True)
        save_checkpoint(conf, epoch, net, optimizer, scheduler, loss_list, rank)


if __name__ == "__main__":
    dev, lr_, r, w = init_distributed(sys.argv[2] if len(sys.argv) > 2 else None)
    main(dev, lr_, r, w)
    torch.distributed.destroy_process_group()

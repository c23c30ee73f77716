"""Hot-path helpers of the reference's utils/misc.py, restated for the HIP build:
patchify / unpatchify (misc.py:14-56, MAE target layout), configure_optimizer (:58-84), configure_scheduler (:86-96),
init_par_groups (:129-238).  File-system / dataloader helpers of the reference module are out of scope.
"""
import torch
import torch.distributed as dist

from .._hip.optim import HipAdamW
from .._hip.params import is_no_decay
from .lr_scheduler import LinearWarmupCosineAnnealingLR


def patchify(data, patch_size, twoD):
    """[B,C,H,W(,Z)] -> [B, L, p^nd * C] with per-patch element order (ph, pw[, pd], c) — channel LAST, unlike the
    conv weight's (c, ph, pw).  Pure index shuffle (torch view/permute); the training loss does not call this: the
    patch-MSE kernel reads the image in place (ucfvit_patch_mse)."""
    B, C = data.shape[0], data.shape[1]
    p = patch_size
    if twoD:
        gh, gw = data.shape[2] // p, data.shape[3] // p
        return data.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 3, 5, 1).reshape(B, gh * gw, p * p * C)
    gh, gw, gz = data.shape[2] // p, data.shape[3] // p, data.shape[4] // p
    return data.reshape(B, C, gh, p, gw, p, gz, p).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(B, gh * gw * gz, p ** 3 * C)


def unpatchify(patchified_pixel_values, data, patch_size, twoD):
    """inverse of patchify; `data` only supplies the target shape"""
    B, C = patchified_pixel_values.shape[0], data.shape[1]
    p = patch_size
    if twoD:
        gh, gw = data.shape[2] // p, data.shape[3] // p
        return patchified_pixel_values.reshape(B, gh, gw, p, p, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, gh * p, gw * p)
    gh, gw, gz = data.shape[2] // p, data.shape[3] // p, data.shape[4] // p
    x = patchified_pixel_values.reshape(B, gh, gw, gz, p, p, p, C).permute(0, 7, 1, 4, 2, 5, 3, 6)
    return x.reshape(B, C, gh * p, gw * p, gz * p)


def configure_optimizer(model, lr, beta_1, beta_2, weight_decay):
    """AdamW with the reference's two groups: everything weight-decayed except names containing
    var_embed / pos_embed / time_pos_embed.  Returns the fused HIP AdamW (a torch.optim.Optimizer)."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        (no_decay if is_no_decay(name) else decay).append(p)
    return HipAdamW(
        [dict(params=decay, lr=lr, betas=(beta_1, beta_2), weight_decay=weight_decay),
         dict(params=no_decay, lr=lr, betas=(beta_1, beta_2), weight_decay=0)])


def configure_scheduler(optimizer, warmup_steps, max_steps, warmup_start_lr, eta_min):
    return LinearWarmupCosineAnnealingLR(optimizer, warmup_steps, max_steps, warmup_start_lr, eta_min)


def init_par_groups(world_rank, data_par_size, tensor_par_size, seq_par_size, fsdp_size, simple_ddp_size):
    """Process groups of the Hybrid-OP layout (reference misc.py:129-238): world = DP x TP(x SP); TP ranks are
    contiguous [i*tp, (i+1)*tp).  Every rank must call this (dist.new_group is collective).
    Returns (seq_par_group, ddp_group, tensor_par_group, data_seq_ort_group, fsdp_group, simple_ddp_group)."""
    world = dist.get_world_size()
    tp, sp = tensor_par_size, seq_par_size
    assert world == data_par_size * tp * sp, "world size must equal data_par_size * tensor_par_size * seq_par_size"
    assert data_par_size == fsdp_size * simple_ddp_size
    mine = dict(tp=None, sp=None, ddp=None, ort=None, fsdp=None, sddp=None)

    def make(key, ranks):
        g = dist.new_group(ranks)
        if world_rank in ranks:
            mine[key] = g

    for i in range(world // tp):                       # tensor-parallel: contiguous ranks
        make("tp", list(range(i * tp, (i + 1) * tp)))
    for i in range(data_par_size):                     # sequence-parallel inside one DP replica, per TP lane
        for t in range(tp):
            make("sp", [i * tp * sp + s * tp + t for s in range(sp)])
    for k in range(tp * sp):                           # data-parallel: same (sp, tp) lane across replicas
        make("ddp", [i * tp * sp + k for i in range(data_par_size)])
    for t in range(tp):                                # same TP lane across every replica and sequence shard
        make("ort", [t + tp * j for j in range(data_par_size * sp)])
    for k in range(tp * sp):                           # FSDP shards x plain-DDP replicas of the DP dimension
        for j in range(simple_ddp_size):
            make("fsdp", [(j * fsdp_size + f) * tp * sp + k for f in range(fsdp_size)])
        for f in range(fsdp_size):
            make("sddp", [(j * fsdp_size + f) * tp * sp + k for j in range(simple_ddp_size)])
    return mine["sp"], mine["ddp"], mine["tp"], mine["ort"], mine["fsdp"], mine["sddp"]

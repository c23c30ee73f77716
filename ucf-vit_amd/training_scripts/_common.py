"""Shared plumbing of the train_*_simple.py entry points (reference: training_scripts/train_class_simple.py:33-446 and siblings):
YAML config in the reference schema, one process per GPU over RCCL, even/odd checkpoints with the reference's dictionary
layout, and a synthetic data source (the reference dataloaders are out of scope: SURVEY.md §2)."""
import os
import sys
import time

import torch
import torch.distributed as dist
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
if PKG not in sys.path:
    sys.path.insert(0, PKG)


def init_distributed(launcher=None):
    """rank / world from torchrun env, SLURM env (reference :426-437) or a single process; backend nccl (= RCCL)"""
    env = os.environ
    if "RANK" in env:
        rank, world, local = int(env["RANK"]), int(env["WORLD_SIZE"]), int(env.get("LOCAL_RANK", 0))
    elif "SLURM_PROCID" in env:
        rank, world = int(env["SLURM_PROCID"]), int(env["SLURM_NTASKS"])
        local = int(env.get("SLURM_LOCALID", rank % max(1, torch.cuda.device_count())))
    else:
        rank, world, local = 0, 1, 0
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", "29500")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    return device, local, rank, world


def load_config(path):
    with open(path) as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def model_args(conf):
    """the subset of `model.net.init_args` + `data` the hot-path models use"""
    a, d = conf["model"]["net"]["init_args"], conf["data"]
    dataset = d["dataset"]
    chans = 1 if d.get("single_channel", False) else max(1, d["num_channels_used"][dataset])
    adaptive = bool(a.get("adaptive_patching", False))
    return dict(img_size=a["tile_size"], patch_size=a["patch_size"], in_chans=chans, embed_dim=a["embed_dim"], depth=a["depth"],
                num_heads=a["num_heads"], mlp_ratio=a["mlp_ratio"], drop_path_rate=a.get("drop_path", 0.0), twoD=a["twoD"],
                default_vars=a["default_vars"], single_channel=d.get("single_channel", False), use_varemb=bool(a.get("use_varemb", False)),
                adaptive_patching=adaptive, fixed_length=a.get("fixed_length", 4096),
                use_adaptive_pos_emb=bool(a.get("use_adaptive_pos_emb", False)) and adaptive), a, d


class SyntheticLoader:
    """per-rank synthetic batches shaped like the reference dataloader's output: (data fp32 un-normalised, label, variables)"""

    def __init__(self, batch_size, in_chans, img_size, num_classes, iters, device, seed, volumetric_labels=False):
        self.shape = (batch_size, in_chans, *img_size)
        self.num_classes, self.iters, self.device, self.seed = num_classes, iters, device, seed
        self.volumetric_labels = volumetric_labels

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.iters):
            data = torch.randint(0, 256, self.shape, generator=g).float()
            if self.volumetric_labels:
                label = torch.randint(0, self.num_classes, (self.shape[0], 1, *self.shape[2:]), generator=g)
            else:
                label = torch.randint(0, max(self.num_classes, 1), (self.shape[0],), generator=g)
            yield data.to(self.device, non_blocking=True), label.to(self.device, non_blocking=True)


def sqrt_len_of(fixed_length, twoD):
    """side of the token grid the reference's UNETR / SAP scripts reshape an adaptive sequence into (train_unetr_simple.py:202-220)"""
    if twoD:
        s = int(round(fixed_length ** 0.5))
        assert s * s == fixed_length, "sqrt of fixed length needs to be a whole number"
        assert fixed_length % 3 == 1, "Quadtree fixed length needs to be 3n+1, where n is some integer"
    else:
        s = int(round(fixed_length ** (1.0 / 3.0)))
        assert s ** 3 == fixed_length, "cube root of fixed length needs to be a whole number"
        assert fixed_length % 7 == 1, "Octtree fixed length needs to be 7n+1, where n is some integer"
    return s


def seq_to_pseudo_image(seq, sqrt_len, patch_size, twoD):
    """[B, C, S, P] -> [B, C, (s1 p1), (s2 p2)[, (s3 p3)]] as train_unetr_simple.py:43-47 does with einops"""
    import einops
    if twoD:
        return einops.rearrange(seq, 'b c (s1 s2) (ps1 ps2) -> b c (s1 ps1) (s2 ps2)', s1=sqrt_len, s2=sqrt_len, ps1=patch_size,
                                ps2=patch_size).contiguous()
    return einops.rearrange(seq, 'b c (s1 s2 s3) (ps1 ps2 ps3) -> b c (s1 ps1) (s2 ps2) (s3 ps3)', s1=sqrt_len, s2=sqrt_len, s3=sqrt_len,
                            ps1=patch_size, ps2=patch_size, ps3=patch_size).contiguous()


class SyntheticSeqLoader:
    """per-rank synthetic batches shaped like the reference's adaptive-patching dataloader output (train_class_simple.py:322-337):
    seq fp32 [B, C, S, P] (S = fixed_length resized patches of P = p^nd pixels), seq_ps fp32 [B, S, 3|4] = (size, position...),
    label.  When fixed_length fits the tree rule (3n+1 / 7n+1) and the images are square, synthetic images and synthetic edge maps
    go through the GPU quadtree / octree patcher (UCF_VIT.dataloaders.transform); otherwise the sequences are drawn directly."""

    def __init__(self, batch_size, in_chans, img_size, patch_size, fixed_length, num_classes, iters, device, seed):
        nd = len(img_size)
        self.shape = (batch_size, in_chans, fixed_length, patch_size ** nd)
        self.nd, self.img = nd, min(img_size)
        self.square = len(set(img_size)) == 1          # the quadtree / octree patchers work on square images / cubic volumes
        self.num_classes, self.iters, self.device, self.seed = num_classes, iters, device, seed

    def _patchified(self, g):
        """synthetic images + synthetic edge maps through the GPU patcher (UCF_VIT.dataloaders.transform, the device-side counterpart
        of the reference's Patchify transform): the same tensors the reference dataloader hands to the training step"""
        from UCF_VIT.dataloaders.transform import Patchify, Patchify_3D
        B, C, S, P = self.shape
        n, p = self.img, round(P ** (1.0 / self.nd))
        dev = self.device
        img = torch.randint(0, 256, (B,) + (n,) * self.nd + (C,), generator=g).float().to(dev)
        edges = ((torch.rand((B,) + (n,) * self.nd, generator=g) < 0.02).to(torch.uint8) * 255).to(dev)
        if self.nd == 2:
            seq, size, pos, _, _ = Patchify(S, p, C)(img, edges)
        else:
            seq, size, pos, _, _ = Patchify_3D(S, p, C)(img, edges)
        return seq, torch.cat([size.unsqueeze(-1), pos], dim=-1)

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        B, _, S, _ = self.shape
        tree_ok = S % (3 if self.nd == 2 else 7) == 1 and (self.nd == 2 or self.img <= 256)
        for _ in range(self.iters):
            if tree_ok and self.square:
                seq, seq_ps = self._patchified(g)
                label = torch.randint(0, max(self.num_classes, 1), (B,), generator=g)
                yield seq, seq_ps, label.to(self.device, non_blocking=True)
                continue
            seq = torch.randint(0, 256, self.shape, generator=g).float()
            size = 2.0 ** torch.randint(1, 6, (B, S, 1), generator=g).float()
            pos = torch.randint(0, self.img, (B, S, self.nd), generator=g).float()
            label = torch.randint(0, max(self.num_classes, 1), (B,), generator=g)
            dev = self.device
            yield seq.to(dev, non_blocking=True), torch.cat([size, pos], dim=-1).to(dev, non_blocking=True), label.to(dev, non_blocking=True)


def save_checkpoint(conf, epoch, model, optimizer, scheduler, loss_list, rank):
    """alternating <name>_even.ckpt / _odd.ckpt on rank 0 (reference :364-388), same dictionary keys"""
    t = conf["trainer"]
    if rank == 0:
        os.makedirs(t["checkpoint_path"], exist_ok=True)
        torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                    "scheduler_state_dict": scheduler.state_dict(), "loss_list": loss_list},
                   os.path.join(t["checkpoint_path"], f"{t['checkpoint_filename']}_{'even' if epoch % 2 == 0 else 'odd'}.ckpt"))
    dist.barrier()


def save_checkpoint_tp(conf, epoch, model, optimizer, scheduler, loss_list, world_rank, tensor_par_size):
    """Hybrid-OP checkpoints (reference train_masked_fsdp.py:624-644): ONE file per tensor-parallel rank of the first TP group,
    `<name>_<even|odd>_rank_<r>.ckpt`, same dictionary keys as the simple scripts.  (The reference stores epoch_start + max_epochs under
    'epoch'; this build stores the epoch that was just finished, which is what its own resume path reads back.)"""
    t = conf["trainer"]
    if world_rank < tensor_par_size:
        os.makedirs(t["checkpoint_path"], exist_ok=True)
        torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                    "scheduler_state_dict": scheduler.state_dict(), "loss_list": loss_list},
                   os.path.join(t["checkpoint_path"], f"{t['checkpoint_filename']}_{'even' if epoch % 2 == 0 else 'odd'}_rank_{world_rank}.ckpt"))
    if dist.is_initialized():
        dist.barrier()


def load_pretrained_mae_encoder(model, mae_state_dict):
    """MAE -> UNETR / SAP encoder transfer (reference train_unetr_simple.py:328-340): every entry of the pretrained MAE's state_dict whose
    name contains neither 'decoder' nor 'mask_token' overwrites the same-named entry of `model`; everything else of `model` (conv decoder,
    heads) keeps its initialisation.  `mae_state_dict` may carry the DDP 'module.' prefix on either side.  Returns the names copied."""
    strip = lambda k: k[len("module."):] if k.startswith("module.") else k
    enc = {strip(k): v for k, v in mae_state_dict.items() if "decoder" not in k and "mask_token" not in k}
    own = model.state_dict()
    prefixed = any(k.startswith("module.") for k in own)
    merged, copied = dict(own), []
    for k, v in enc.items():
        kk = ("module." + k) if prefixed else k
        if kk not in own:
            raise KeyError(f"pretrained encoder entry {k} has no counterpart in the model")
        if tuple(own[kk].shape) != tuple(v.shape):
            raise ValueError(f"pretrained encoder entry {k}: shape {tuple(v.shape)} != model {tuple(own[kk].shape)}")
        merged[kk] = v
        copied.append(k)
    model.load_state_dict(merged)
    return copied


def maybe_resume(conf, model, optimizer, scheduler):
    t = conf["trainer"]
    if not t.get("resume_from_checkpoint", False):
        return 0, []
    ck = torch.load(os.path.join(t["checkpoint_path"], t["checkpoint_filename_for_loading"] + ".ckpt"), map_location="cpu", weights_only=True)
    model.load_state_dict(ck["model_state_dict"])
    optimizer.load_state_dict(ck["optimizer_state_dict"])
    scheduler.load_state_dict(ck["scheduler_state_dict"])
    return ck["epoch"] + 1, ck["loss_list"]


def iters_per_epoch(conf):
    return max(conf["load_balancing"]["batches_per_rank_epoch"].values())


class StepTimer:
    def __init__(self):
        self.t0, self.n = time.perf_counter(), 0

    def tick(self, imgs):
        self.n += imgs

    def rate(self):
        torch.cuda.synchronize()
        return self.n / (time.perf_counter() - self.t0)

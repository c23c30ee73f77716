"""Hybrid-OP (tensor parallel) tests.
CPU: autograd pairing of the collectives in UCF_VIT.utils.dist_functions with 2 gloo ranks.
GPU: TP=2 Block / Attention / Mlp on ONE MI355X — two shard models run in two threads whose 'process group' is an in-process
stand-in that sums their tensors; result must equal the unsharded operator with the mapped weights."""
import os
import sys
import threading
from functools import partial

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, rel_err


def _df_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "ucf-vit_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from UCF_VIT.utils import dist_functions as DF
    ok = True
    x = torch.full((3,), float(rank + 1), requires_grad=True)
    y = DF.F_Identity_B_AllReduce(x)                       # fwd identity, bwd all-reduce(SUM)
    (y * (rank + 1)).sum().backward()
    ok &= torch.equal(y.detach(), x.detach()) and torch.equal(x.grad, torch.full((3,), 3.0))
    x = torch.full((3,), float(rank + 1), requires_grad=True)
    y = DF.F_AllReduce_B_Identity(x)                       # fwd all-reduce(SUM), bwd identity
    (y * (rank + 2)).sum().backward()
    ok &= torch.equal(y.detach(), torch.full((3,), 3.0)) and torch.equal(x.grad, torch.full((3,), float(rank + 2)))
    x = torch.full((3,), float(rank + 1), requires_grad=True)
    y = DF.F_Identity_B_Broadcast(x, 0)                    # bwd: gradient of rank 0 everywhere
    (y * (rank + 5)).sum().backward()
    ok &= torch.equal(x.grad, torch.full((3,), 5.0))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_dist_functions_autograd_pairs_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_df_worker, args=(r, 2, 29544, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_init_par_groups_layout_single_process():
    """rank lists of the Hybrid-OP groups (reference utils/misc.py:129-238), checked without a process group"""
    import UCF_VIT.utils.misc as misc
    made = []

    class FakeDist:
        @staticmethod
        def get_world_size():
            return 8

        @staticmethod
        def new_group(ranks):
            made.append(list(ranks))
            return tuple(ranks)
    real = misc.dist
    misc.dist = FakeDist
    try:
        sp, ddp, tp, ort, fsdp, sddp = misc.init_par_groups(5, data_par_size=2, tensor_par_size=2, seq_par_size=2, fsdp_size=2, simple_ddp_size=1)
    finally:
        misc.dist = real
    assert tp == (4, 5) and sp == (5, 7) and ddp == (1, 5) and ort == (1, 3, 5, 7) and fsdp == (1, 5) and sddp == (5,)


# ---------------------------------------------------------------------------------------------------------- GPU
def _shard_block_state(full, r, tp, H):
    """weights of TP rank r from an unsharded Block state_dict (qkv rows are laid out [3][H][dh])"""
    D = full["norm1.weight"].numel()
    dh, Hl = D // H, H // tp
    out = {}
    for k, v in full.items():
        if k == "attn.qkv.weight":
            out[k] = v.view(3, H, dh, D)[:, r * Hl:(r + 1) * Hl].reshape(3 * Hl * dh, D)
        elif k == "attn.qkv.bias":
            out[k] = v.view(3, H, dh)[:, r * Hl:(r + 1) * Hl].reshape(-1)
        elif k == "attn.proj.weight":
            out[k] = v[:, r * Hl * dh:(r + 1) * Hl * dh]
        elif k in ("mlp.fc1.weight", "mlp.fc1.bias"):
            n = v.shape[0] // tp
            out[k] = v[r * n:(r + 1) * n]
        elif k == "mlp.fc2.weight":
            n = v.shape[1] // tp
            out[k] = v[:, r * n:(r + 1) * n]
        else:
            out[k] = v
        out[k] = out[k].contiguous().clone()
    return out


def _tp_worker(rank, world, port, dtype_name, tol, q):
    for p in (os.path.join(ROOT, "ucf-vit_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from UCF_VIT.simple import building_blocks as SB
        from UCF_VIT.fsdp import building_blocks as FB
        dtype = getattr(torch, dtype_name)
        g = load_golden("op_block.npz")
        H, tp = 2, world
        full_sd = {k[2:]: v for k, v in g.items() if k.startswith("w.")}
        sd = _shard_block_state(full_sd, rank, tp, H)
        for k in ("attn.proj.bias", "mlp.fc2.bias"):
            sd[k] = sd[k] / tp     # every rank adds its bias before the SUM (reference behaviour): split it so the sum matches
        blk = FB.Block(64, H, qkv_bias=True, norm_layer=partial(SB.LayerNorm, eps=1e-6), tensor_par_size=tp,
                       tensor_par_group=dist.group.WORLD)
        blk.load_state_dict(sd)
        blk = blk.to("cuda:0")
        SB.set_compute_dtype(blk, dtype)
        x = g["x"].to("cuda:0").requires_grad_(True)
        y = blk(x)
        y.backward(g["gy"].to("cuda:0", y.dtype))
        msgs = []
        if rel_err(y.float(), g["y"]) >= tol:
            msgs.append("y")
        if rel_err(x.grad.float(), g["gx"]) >= tol:
            msgs.append("gx")
        exp = _shard_block_state({k[2:]: v for k, v in g.items() if k.startswith("g.")}, rank, tp, H)
        for k, p in blk.named_parameters():
            if rel_err(p.grad, exp[k]) >= tol:
                msgs.append(k)
        q.put((rank, msgs))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name,tol", [("float32", 1e-3), ("bfloat16", 5e-2)])
def test_tp2_block_equals_unsharded(dtype_name, tol):
    """two TP ranks (two processes sharing the one GPU, gloo transport for the collectives) == unsharded reference Block"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29550 + (0 if dtype_name == "float32" else 1)
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, port, dtype_name, tol, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, msgs in res:
        assert not msgs, f"TP rank {rank}: mismatch in {msgs}"

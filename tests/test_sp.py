"""Sequence parallelism (new capability; no reference implementation exists — SURVEY.md F4): P=2 ranks, each holding half of the
tokens, must reproduce the unsharded Block: outputs / input gradients are the token shards, parameter gradients sum to the
unsharded ones.  Two processes share the one GPU; the all-to-alls go through gloo (host-staged) in the test."""
import os
import sys
from functools import partial

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, rel_err


def _sp_worker(rank, world, port, dtype_name, tol, q):
    for p in (os.path.join(ROOT, "ucf-vit_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from UCF_VIT.simple import building_blocks as SB
        from UCF_VIT.fsdp.seq_parallel import SeqParallelBlock
        from oracle import ucf_vit_ref as R
        from det_weights import det_state_dict, det_tensor
        dtype = getattr(torch, dtype_name)
        D, H, B, N = 128, 4, 2, 96
        ref = R.Block(D, H, 4.0, True, partial(torch.nn.LayerNorm, eps=1e-6))
        sd = det_state_dict(ref, 61)
        ref.load_state_dict(sd)
        x = det_tensor((B, N, D), 62)
        gy = det_tensor((B, N, D), 63)
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)
        yr.backward(gy)
        blk = SB.Block(D, H, qkv_bias=True, norm_layer=partial(SB.LayerNorm, eps=1e-6))
        blk.load_state_dict(sd)
        blk = blk.to("cuda:0")
        SB.set_compute_dtype(blk, dtype)
        sp = SeqParallelBlock(blk, dist.group.WORLD)
        Nl = N // world
        sl = slice(rank * Nl, (rank + 1) * Nl)
        xl = x[:, sl].contiguous().to("cuda:0").requires_grad_(True)
        y = sp(xl)
        y.backward(gy[:, sl].contiguous().to("cuda:0", y.dtype))
        bad = []
        if rel_err(y.float(), yr[:, sl]) >= tol:
            bad.append("y")
        if rel_err(xl.grad.float(), xr.grad[:, sl]) >= tol:
            bad.append("gx")
        for (k, p), (_, pr) in zip(blk.named_parameters(), ref.named_parameters()):
            g = p.grad.detach().float().cpu()
            dist.all_reduce(g)                                  # partial sums over the token shards
            if rel_err(g, pr.grad) >= tol:
                bad.append(k)
        q.put((rank, bad))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name,tol", [("float32", 1e-3), ("bfloat16", 5e-2)])
def test_seq_parallel_block_equals_unsharded(dtype_name, tol):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29560 + (0 if dtype_name == "float32" else 1)
    procs = [ctx.Process(target=_sp_worker, args=(r, 2, port, dtype_name, tol, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, bad in res:
        assert not bad, f"SP rank {rank}: mismatch in {bad}"

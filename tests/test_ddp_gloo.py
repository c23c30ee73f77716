"""Data-parallel reducer logic (UCF_VIT._hip.ddp.HipDataParallel) with 2 gloo ranks on CPU: parameter broadcast at wrap time,
bucketed mean all-reduce overlapped with backward, identical results to the mean of per-rank gradients.  The HIP model path
uses the same bucket/hook code with the flat gradient buffer of HipParamStore (GPU tests cover that with world_size 1)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, bucket_mb, q):
    for p in (os.path.join(ROOT, "ucf-vit_amd"), ROOT, os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import ucf_vit_ref as R
    from det_weights import det_state_dict, det_tensor
    from UCF_VIT._hip.ddp import HipDataParallel
    kw = dict(img_size=[32, 32], patch_size=8, in_chans=3, num_classes=5, embed_dim=64, depth=2, num_heads=2)
    model = R.VIT(**kw)
    model.load_state_dict(det_state_dict(model, 100 + rank))       # ranks start DIFFERENT: wrap must broadcast rank 0's weights
    ddp = HipDataParallel(model, bucket_mb=bucket_mb)
    ref = R.VIT(**kw)
    ref.load_state_dict(det_state_dict(ref, 100))
    same = all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), ref.state_dict().values()))
    # two steps, each rank its own batch; expected grad = mean over ranks of the single-rank grads
    ok = same
    for step in range(2):
        xs = [det_tensor((2, 3, 32, 32), 10 * step + r) for r in range(world)]
        ys = [torch.tensor([(r + step) % 5, (2 * r + 1) % 5]) for r in range(world)]
        exp = None
        for r in range(world):
            ref.zero_grad()
            torch.nn.CrossEntropyLoss()(ref(xs[r]), ys[r]).backward()
            g = [p.grad.clone() for p in ref.parameters()]
            exp = g if exp is None else [a + b for a, b in zip(exp, g)]
        exp = [e / world for e in exp]
        ddp.zero_grad(set_to_none=(step == 0))
        loss = torch.nn.CrossEntropyLoss()(ddp(xs[rank]), ys[rank])
        loss.backward()
        for p, e in zip(model.parameters(), exp):
            ok = ok and torch.allclose(p.grad, e, rtol=1e-5, atol=1e-7)
    keys_ok = all(k.startswith("module.") for k in ddp.state_dict().keys())
    q.put((rank, bool(ok), bool(keys_ok), len(ddp.buckets)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb,min_buckets", [(32, 1), (0.05, 3)])
def test_reducer_two_ranks_gloo(bucket_mb, min_buckets):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29511 + int(bucket_mb * 100) % 97
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_mb, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, keys_ok, nb in res:
        assert ok, f"rank {rank}: gradients differ from the mean of per-rank gradients"
        assert keys_ok and nb >= min_buckets

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


def _worker(rank, world, port, bucket_mb, q, algorithm="all_reduce"):
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
    ddp = HipDataParallel(model, bucket_mb=bucket_mb, algorithm=algorithm)
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


@pytest.mark.parametrize("world", [2, 3])
def test_reducer_direct_reduce_scatter_all_gather_gloo(world):
    """algorithm="direct": the bucket's mean by all-to-all (chunk j to rank j) + fixed-order local sum + all-gather — the transport SURVEY §5
    recommends for the fully connected xGMI node — equals the mean of the per-rank gradients; 3 ranks: bucket sizes that do not divide by
    the world size (zero-padded chunks)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, 29541 + world, 0.05, q, "direct")) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, keys_ok, nb in res:
        assert ok, f"rank {rank}: gradients differ from the mean of per-rank gradients"
        assert keys_ok and nb >= 3


def _hip_dp_worker(rank, world, port, q, reduce_dtype=None, algorithm=None):
    for p in (os.path.join(ROOT, "ucf-vit_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from UCF_VIT.simple.arch import VIT
        from UCF_VIT.utils.metrics import cross_entropy_loss
        from UCF_VIT.utils.misc import configure_optimizer
        from UCF_VIT._hip.ddp import HipDataParallel
        from oracle import ucf_vit_ref as R
        from det_weights import det_state_dict, det_tensor
        kw = dict(img_size=[32, 32], patch_size=8, in_chans=3, num_classes=5, embed_dim=64, depth=2, num_heads=2)
        ref = R.VIT(**kw)
        ref.load_state_dict(det_state_dict(ref, 100))
        m = VIT(**kw)
        m.load_state_dict(det_state_dict(m, 100 + rank))        # different per rank: the wrap broadcasts rank 0's weights
        m = m.to("cuda:0")
        ddp = HipDataParallel(m, bucket_mb=0.05, reduce_dtype=reduce_dtype, algorithm=algorithm)
        opt = configure_optimizer(m, 1e-3, 0.9, 0.95, 0.0)
        xs = [det_tensor((2, 3, 32, 32), 10 + r) for r in range(world)]
        ys = [torch.tensor([r % 5, (2 * r + 1) % 5]) for r in range(world)]
        exp = None
        for r in range(world):
            ref.zero_grad()
            torch.nn.CrossEntropyLoss()(ref(xs[r]), ys[r]).backward()
            g = [p.grad.clone() for p in ref.parameters()]
            exp = g if exp is None else [a + b for a, b in zip(exp, g)]
        out = ddp(xs[rank].to("cuda:0"), None, None)
        cross_entropy_loss(out, ys[rank].to("cuda:0")).backward()
        torch.cuda.synchronize()
        from conftest import rel_err
        tol = 1e-3 if reduce_dtype is None else 1e-2           # bf16 transport: each rank's gradient and the mean are rounded to 8 bits
        bad = [k for (k, p), e in zip(m.named_parameters(), exp) if rel_err(p.grad, e / world) > tol]
        if reduce_dtype == "bf16":
            bad += [k + " (not a bf16 value)" for k, p in m.named_parameters() if not torch.equal(p.grad, p.grad.bfloat16().float())]
        opt.step()
        opt.zero_grad()
        # after an identical update every rank must hold identical weights
        w = m._ucf_store.flat_p.detach().cpu()
        w0 = w.clone()
        dist.broadcast(w0, 0)
        q.put((rank, bad, bool(torch.equal(w, w0)), bool(opt._flat)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_hip_data_parallel_bf16_gradient_transport_two_ranks():
    """reduce_dtype = bf16 (the reference's MixedPrecision(reduce_dtype=bfloat16) policy, train_masked_fsdp.py:375-381): the mean
    gradient arrives as bf16 values within 1e-2 of the fp32 mean, and the ranks stay bit-identical after the optimizer step"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hip_dp_worker, args=(r, 2, 29573, q, "bf16")) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, bad, same, flat in res:
        assert not bad, f"rank {rank}: gradient mismatch in {bad}"
        assert same and flat


def _hip_dp_foreign_worker(rank, world, port, q):
    """SAP: encoder gradients are written by the HIP kernels into the flat buffer, the transposed-convolution neck and the 1x1 header are
    torch modules whose gradients autograd allocates itself: both kinds must come out as the mean over the ranks"""
    for p in (os.path.join(ROOT, "ucf-vit_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from UCF_VIT.simple.arch import SAP
        from UCF_VIT._hip.ddp import HipDataParallel
        from UCF_VIT.utils.misc import configure_optimizer
        from det_weights import det_state_dict, det_tensor
        kw = dict(img_size=[32, 32], patch_size=8, in_chans=3, num_classes=3, embed_dim=64, depth=1, num_heads=2, sqrt_len=4, class_token=False,
                  weight_init='skip')
        torch.manual_seed(5)
        m = SAP(**kw)
        m.load_state_dict(det_state_dict(m, 300))
        m = m.to("cuda:0")
        xs = [det_tensor((2, 3, 32, 32), 310 + r) for r in range(world)]
        # expected: mean over the ranks of the single-rank gradients (computed with an unwrapped copy of the same weights)
        ref = SAP(**kw)
        ref.load_state_dict(det_state_dict(ref, 300))
        ref = ref.to("cuda:0")
        exp = None
        for r in range(world):
            ref.zero_grad()
            ref(xs[r].to("cuda:0"), None).float().square().mean().backward()
            g = [p.grad.detach().clone() for p in ref.parameters()]
            exp = g if exp is None else [a + b for a, b in zip(exp, g)]
        ddp = HipDataParallel(m, bucket_mb=0.05)
        opt = configure_optimizer(m, 1e-3, 0.9, 0.95, 0.0)
        ddp(xs[rank].to("cuda:0"), None).float().square().mean().backward()
        torch.cuda.synchronize()
        from conftest import rel_err
        bad = [k for (k, p), e in zip(m.named_parameters(), exp) if p.grad is None or rel_err(p.grad, e / world) > 1e-3]
        inflat = all(p.grad.data_ptr() == m._ucf_store.flat_g.data_ptr() + 4 * o for p, o in zip(m._ucf_store.params, m._ucf_store.offsets)
                     if p.grad is not None)
        opt.step()
        q.put((rank, bad, inflat))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_hip_data_parallel_reduces_torch_produced_gradients_too():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hip_dp_foreign_worker, args=(r, 2, 29574, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, bad, inflat in res:
        assert not bad, f"rank {rank}: gradient mismatch in {bad}"
        assert inflat


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm,reduce_dtype", [(None, None), ("direct", None), ("direct", "bf16")])
def test_hip_data_parallel_two_ranks_share_one_gpu(algorithm, reduce_dtype):
    """the HIP model + flat-buffer reducer + fused AdamW with world_size 2 (both ranks on the one GPU, gloo transport):
    gradients = mean over ranks, weights stay identical after the step — with the default all-reduce and with the written-out
    reduce-scatter + all-gather ("direct": all-to-all, fixed-order local sum, all-gather), fp32 and bf16 transport"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29571 + (10 if algorithm else 0) + (1 if reduce_dtype else 0)
    procs = [ctx.Process(target=_hip_dp_worker, args=(r, 2, port, q, reduce_dtype, algorithm)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, bad, same, flat in res:
        assert not bad, f"rank {rank}: gradient mismatch in {bad}"
        assert same and flat


def _hip_dp_unetr_worker(rank, world, port, q):
    """whole UNETR with the convolutional decoder on the HIP kernels: the decoder's weight gradients come back from autograd functions
    (UCF_VIT/_hip/conv.py) as freshly allocated tensors, the encoder's are written into the flat buffer by the kernels: both kinds must come
    out as the mean over the ranks, inside the flat buffer the optimizer and the buckets use"""
    for p in (os.path.join(ROOT, "ucf-vit_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from UCF_VIT.simple.arch import UNETR
        from UCF_VIT._hip import functional as HF
        from UCF_VIT._hip.ddp import HipDataParallel
        from UCF_VIT.utils.misc import configure_optimizer
        from det_weights import det_state_dict, det_tensor
        img = [32, 32, 32]
        kw = dict(img_size=img, patch_size=16, in_chans=1, embed_dim=96, depth=4, num_heads=3, class_token=False, twoD=False, num_classes=4,
                  linear_decoder=False, feature_size=16, skip_connection=True)
        m = UNETR(**kw)
        sd = det_state_dict(m, 400)
        m.load_state_dict(sd)
        m = m.to("cuda:0")
        assert m.hip_decoder()
        xs = [det_tensor((1, 1, *img), 410 + r).abs() for r in range(world)]
        labs = [(x[:, 0] * 3).long().clamp_(0, 3) for x in xs]
        ref = UNETR(**kw)
        ref.load_state_dict(sd)
        ref = ref.to("cuda:0")
        exp = None
        for r in range(world):
            ref.zero_grad()
            HF.dice_ce(ref(xs[r].to("cuda:0"), None), labs[r].to("cuda:0")).backward()
            HF.flush_wgrads()
            g = [p.grad.detach().clone() for p in ref.parameters()]
            exp = g if exp is None else [a + b for a, b in zip(exp, g)]
        ddp = HipDataParallel(m, bucket_mb=0.05)
        opt = configure_optimizer(m, 1e-3, 0.9, 0.95, 0.0)
        HF.dice_ce(ddp(xs[rank].to("cuda:0"), None), labs[rank].to("cuda:0")).backward()
        torch.cuda.synchronize()
        from conftest import rel_err
        bad = [k for (k, p), e in zip(m.named_parameters(), exp) if p.grad is None or rel_err(p.grad, e / world) > 1e-3]
        inflat = all(p.grad.data_ptr() == m._ucf_store.flat_g.data_ptr() + 4 * o for p, o in zip(m._ucf_store.params, m._ucf_store.offsets)
                     if p.grad is not None)
        n_dec = sum(k.startswith(("encoder", "decoder", "out.")) for k, _ in m.named_parameters())
        opt.step()
        w = m._ucf_store.flat_p.detach().cpu()
        w0 = w.clone()
        dist.broadcast(w0, 0)
        q.put((rank, bad, inflat, n_dec, bool(torch.equal(w, w0))))
        dist.barrier()
    except Exception as e:          # report instead of letting the parent sit out its queue timeout on a GPU box
        import traceback
        q.put((rank, ["EXCEPTION " + repr(e) + " " + traceback.format_exc()[-1200:]], False, -1, False))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_hip_data_parallel_whole_unetr_with_hip_decoder_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hip_dp_unetr_worker, args=(r, 2, 29576, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    for _ in range(2):
        res.append(q.get(timeout=240))
        if res[-1][3] == -1:                       # a rank raised: the other one hangs in a collective
            for p in procs:
                p.kill()
            raise AssertionError(f"rank {res[-1][0]}: {res[-1][1]}")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, bad, inflat, n_dec, same in res:
        assert not bad, f"rank {rank}: gradient mismatch in {bad}"
        assert inflat and n_dec == 33 and same

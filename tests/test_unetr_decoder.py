"""csrc/unetr_decoder.hip (the HBM-bound part of the UNETR conv decoder) against oracle/unetr_decoder_ref.py — the
plain-torch restatement of monai's UnetResBlock normalisation chain and DiceCELoss (PARITY UNPINNED against monai itself: not vendored,
no reference fixtures).  Tolerances: fp32 1e-4 rel (statistics over up to 10^6 voxels), bf16 3e-2."""
import pytest
import torch

from conftest import rel_err

DEV = "cuda"


def test_oracle_dice_ce_matches_its_definition_cpu():
    """(CPU) the restated loss on a case that can be checked by hand: perfect one-hot logits give dice -> 0 and CE -> 0"""
    from oracle import unetr_decoder_ref as R
    lab = torch.randint(0, 4, (2, 6, 5, 4), generator=torch.Generator().manual_seed(0))
    logits = torch.nn.functional.one_hot(lab, 4).movedim(-1, 1).float() * 60.0
    assert float(R.dice_ce_loss(logits, lab)) < 1e-4
    uniform = torch.zeros(2, 4, 6, 5, 4)
    l0 = float(R.dice_ce_loss(uniform, lab))
    assert abs(l0 - (torch.log(torch.tensor(4.0)).item() + 1.0 - float((2 * (0.25 * torch.nn.functional.one_hot(lab, 4).movedim(-1, 1).float()).sum((2, 3, 4)) + 1e-5)
                                                                        .div((0.0625 * 120 + torch.nn.functional.one_hot(lab, 4).movedim(-1, 1).float().sum((2, 3, 4))) + 1e-5).mean()))) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("shape,with_res,slope", [((2, 5, 16, 12, 8), False, 0.01), ((1, 3, 40, 40, 32), True, 0.01), ((3, 4, 24, 20), True, 0.01),
                                                  ((2, 6, 8, 8, 8), False, 1.0)])
def test_instance_norm_activation_chain_fwd_bwd(dtype, tol, shape, with_res, slope):
    from UCF_VIT._hip import functional as HF
    from oracle import unetr_decoder_ref as R
    gen = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=gen) * 2.0 + 0.7).to(dtype)
    res = torch.randn(*shape, generator=gen).to(dtype) if with_res else None
    gy = torch.randn(*shape, generator=gen).to(dtype)
    xr = x.float().clone().requires_grad_(True)
    rr = res.float().clone().requires_grad_(True) if with_res else None
    yr = R.inst_norm_act(xr, rr, 1e-5, slope)
    yr.backward(gy.float())
    xd = x.to(DEV).requires_grad_(True)
    rd = res.to(DEV).requires_grad_(True) if with_res else None
    y = HF.instnorm_act(xd, rd, 1e-5, slope)
    y.backward(gy.to(DEV))
    assert rel_err(y.float(), yr.detach()) < tol
    assert rel_err(xd.grad.float(), xr.grad) < (tol if dtype == torch.float32 else 6e-2)
    if with_res:
        assert rel_err(rd.grad.float(), rr.grad) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("n,shape", [(4, (2, 16, 12, 8)), (2, (3, 30, 30)), (8, (1, 40, 40, 21))])
def test_dice_ce_loss_fwd_bwd(n, shape):
    from UCF_VIT._hip import functional as HF
    from oracle import unetr_decoder_ref as R
    gen = torch.Generator().manual_seed(n)
    B = shape[0]
    logits = torch.randn(B, n, *shape[1:], generator=gen) * 2.0
    lab = torch.randint(0, n, shape, generator=gen)
    lr = logits.clone().requires_grad_(True)
    ref = R.dice_ce_loss(lr, lab)
    (ref * 1.7).backward()
    ld = logits.to(DEV).requires_grad_(True)
    loss = HF.dice_ce(ld, lab.to(DEV))
    (loss * 1.7).backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
    assert rel_err(ld.grad, lr.grad) < 1e-4
    # bf16 logits
    lb = logits.bfloat16().to(DEV).requires_grad_(True)
    loss_b = HF.dice_ce(lb, lab.to(DEV))
    loss_b.backward()
    assert abs(loss_b.item() - ref.item()) < 2e-2 * abs(ref.item())
    assert rel_err(lb.grad.float(), lr.grad / 1.7) < 3e-2


@pytest.mark.gpu
def test_unet_res_block_module_vs_torch_restatement():
    """the module the UNETR decoder is built from (fused HIP normalisation chains around MIOpen convolutions) against the all-torch
    restatement with the same weights: output, input gradient and every weight gradient"""
    from UCF_VIT.simple.unetr_blocks import UnetResBlock
    from oracle import unetr_decoder_ref as R
    gen = torch.Generator().manual_seed(5)
    for cin, cout in ((6, 4), (4, 4)):
        blk = UnetResBlock(3, cin, cout, 3, 1).to(DEV)
        x = torch.randn(2, cin, 12, 12, 8, generator=gen).to(DEV)
        gy = torch.randn(2, cout, 12, 12, 8, generator=gen).to(DEV)
        xd = x.clone().requires_grad_(True)
        y = blk(xd)
        y.backward(gy)
        w1, w2 = blk.conv1.conv.weight.detach().clone().requires_grad_(True), blk.conv2.conv.weight.detach().clone().requires_grad_(True)
        w3 = blk.conv3.conv.weight.detach().clone().requires_grad_(True) if blk.downsample else None
        xr = x.clone().requires_grad_(True)
        yr = R.res_block(xr, w1, w2, w3)
        yr.backward(gy)
        assert rel_err(y, yr.detach()) < 1e-4
        assert rel_err(xd.grad, xr.grad) < 1e-3
        assert rel_err(blk.conv1.conv.weight.grad, w1.grad) < 1e-3 and rel_err(blk.conv2.conv.weight.grad, w2.grad) < 1e-3
        if w3 is not None:
            assert rel_err(blk.conv3.conv.weight.grad, w3.grad) < 1e-3


@pytest.mark.gpu
def test_instance_norm_properties_at_volume_size():
    """512 x 512 x 128 voxels per row (the BASELINE config 5 volume; 4 channels = 537 MB fp32): rows come out standardised, a constant
    upstream gradient vanishes through the normalisation, and the residual gradient is the activation-masked upstream gradient"""
    from UCF_VIT._hip import ops
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(1, 4, 512, 512, 128, device=DEV, generator=gen) * 3.0 + 5.0
    y, mean, rstd = ops.instnorm_fwd(x, None, 1e-5, 1.0)
    assert float(y.mean(dim=(2, 3, 4)).abs().max()) < 1e-4 and float((y.var(dim=(2, 3, 4), unbiased=False) - 1).abs().max()) < 1e-3
    assert float((mean - x.mean(dim=(2, 3, 4)).view(-1)).abs().max()) < 1e-3
    dx, _ = ops.instnorm_bwd(torch.ones_like(x), y, x, mean, rstd, 1.0, False)
    assert float(dx.abs().max()) < 1e-4
    del dx
    yl, mean, rstd = ops.instnorm_fwd(x, x, 1e-5, 0.01)
    dy = torch.full_like(x, 2.0)
    dx, dres = ops.instnorm_bwd(dy, yl, x, mean, rstd, 0.01, True)
    assert torch.equal(dres, torch.where(yl > 0, dy, dy * 0.01))

"""The UNETR convolutional decoder SHARDED over the ranks of a sequence-parallel group — new capability, like sequence parallelism itself
(the reference constructs seq_par_group and asserts seq_par_size == 1: training_scripts/train_masked_fsdp.py:220; the decoder being
sharded is src/UCF_VIT/simple/arch.py:808-940,960-993, the loss training_scripts/train_unetr_simple.py:38).

The token shards of fsdp/seq_parallel.py are contiguous in (x, y, z) row-major order, i.e. X-SLABS of the token grid, and every decoder
stage up-samples by 2 per axis: rank r owns the slab x in [r X / P, (r + 1) X / P) of EVERY feature map, at every resolution.  Per layer:

  * transposed 2x2x2 convolutions, 1x1x1 convolutions, channel concatenation, LeakyReLU: voxel-local, no communication.
  * 3x3x3 convolution: needs ONE plane of the neighbours' slabs on either side.  HaloPadFn exchanges the two boundary planes
    ([B, 1, Y, Z, C] each, point to point with the two neighbours) and returns the slab extended by a plane on both sides (zeros at the
    volume's own faces: the convolution's zero padding); the existing convolution kernel runs on the extended slab and the two outer
    output planes are dropped.  Backward: the gradient of the extended slab's outer planes belongs to the neighbours' boundary planes
    and is sent back and added there (the reverse exchange).  Per rank and 3x3x3 layer: 2 x B x Y x Z x C x 2 bytes each way — at
    512 x 512 x 128 / 16 channels 2 x 2.1 MB, forward and backward (DESIGN.md §7 lists every layer).
  * instance norm: the statistics run over the whole volume.  Slabs are equal, so the global mean / variance follow from the ranks'
    (mean, variance) by one all-reduce of 2 floats per (sample, channel); the backward pass needs the two global means of dy' and
    dy' xhat the same way (ucfvit_instnorm_cl_bwd_sums / _bwd_apply).
  * Dice + CE: a function of per-(sample, class) sums over voxels: one all-reduce of B x 25 floats (ucfvit_dice_ce_stats / _from_stats).

Parameter gradients of a sharded layer are partial sums over the rank's slab.  ShardedDiceCEFn multiplies the gradient it starts
backward with by P, so the MEAN over the dp x sp ranks that HipDataParallel takes is the sum over the slabs (the same convention as
seq_parallel.GatherTokensFn, which the replicated-decoder path uses).

The blocks below run the 3-D residual block as a chain of per-layer autograd functions (halo exchange, convolution, normalisation);
the single-node fusions of _hip/conv.py:UnetResBlockFn (statistics from the convolution epilogue, dual-normalisation tail) are not
applied across the exchanges — per-rank work is a little higher than 1 / P of the fused unsharded decoder, communication is per layer.
"""
import torch
import torch.distributed as dist

from UCF_VIT._hip import conv as HC
from UCF_VIT._hip import ops


def _host_staged(group):
    return dist.get_backend(group) == "gloo"       # test transport (ranks sharing one GPU): RCCL refuses that, stage through the host


def _neighbours(spg):
    """global ranks of the slab below / above this rank's (None at the volume's faces)"""
    g = spg.sp_group
    lo = dist.get_global_rank(g, spg.rank - 1) if spg.rank > 0 else None
    hi = dist.get_global_rank(g, spg.rank + 1) if spg.rank + 1 < spg.size else None
    return lo, hi


def _exchange(send_lo, send_hi, spg):
    """send_lo goes to the rank below, send_hi to the rank above; returns (received from below, received from above), each None at a face.
    The planes are small (a few MB at full resolution): one batched isend / irecv pair per neighbour."""
    g = spg.sp_group
    lo, hi = _neighbours(spg)
    staged = send_lo.is_cuda and _host_staged(g)
    conv = (lambda t: t.float().cpu()) if staged else (lambda t: t.contiguous())
    s_lo, s_hi = conv(send_lo), conv(send_hi)
    r_lo = torch.empty_like(s_lo) if lo is not None else None
    r_hi = torch.empty_like(s_hi) if hi is not None else None
    reqs = []
    if lo is not None:
        reqs += [dist.P2POp(dist.isend, s_lo, lo, g), dist.P2POp(dist.irecv, r_lo, lo, g)]
    if hi is not None:
        reqs += [dist.P2POp(dist.isend, s_hi, hi, g), dist.P2POp(dist.irecv, r_hi, hi, g)]
    if reqs:
        for w in dist.batch_isend_irecv(reqs):
            w.wait()
    if staged:
        r_lo = None if r_lo is None else r_lo.to(send_lo.device, send_lo.dtype)
        r_hi = None if r_hi is None else r_hi.to(send_hi.device, send_hi.dtype)
    return r_lo, r_hi


def _mean_over_group(t, spg):
    """in-place mean of a small fp32 tensor over the group"""
    g = spg.sp_group
    if t.is_cuda and _host_staged(g):
        h = t.detach().cpu()
        dist.all_reduce(h, group=g)
        t.copy_(h.to(t.device))
    else:
        dist.all_reduce(t, group=g)
    t.div_(spg.size)
    return t


def _sum_over_group(t, spg):
    g = spg.sp_group
    if t.is_cuda and _host_staged(g):
        h = t.detach().cpu()
        dist.all_reduce(h, group=g)
        t.copy_(h.to(t.device))
    else:
        dist.all_reduce(t, group=g)
    return t


class HaloPadFn(torch.autograd.Function):
    """x [B, Xl, Y, Z, C] -> [B, Xl + 2, Y, Z, C]: the slab with the neighbours' boundary planes (zeros at the volume's faces)"""

    @staticmethod
    def forward(ctx, x, spg):
        ctx.spg = spg
        B, Xl, Y, Z, C = x.shape
        out = torch.empty((B, Xl + 2, Y, Z, C), dtype=x.dtype, device=x.device)
        out[:, 1:Xl + 1] = x
        from_lo, from_hi = _exchange(x[:, :1], x[:, Xl - 1:], spg)          # my first plane goes down, my last plane goes up
        if from_lo is None:
            out[:, :1].zero_()
        else:
            out[:, :1] = from_lo
        if from_hi is None:
            out[:, Xl + 1:].zero_()
        else:
            out[:, Xl + 1:] = from_hi
        return out

    @staticmethod
    def backward(ctx, g):
        spg = ctx.spg
        Xl = g.shape[1] - 2
        dx = g[:, 1:Xl + 1].clone()
        # the gradient of my two outer planes belongs to the neighbours' boundary planes: plane 0 is the lower neighbour's last plane
        from_lo, from_hi = _exchange(g[:, :1], g[:, Xl + 1:], spg)
        if from_lo is not None:
            dx[:, :1] += from_lo              # what the lower neighbour computed for ITS upper halo = my first plane
        if from_hi is not None:
            dx[:, Xl - 1:] += from_hi
        return dx, None


def halo_pad(x, spg):
    return HaloPadFn.apply(x, spg)


def _interior(y):
    """[B, Xl + 2, ...] -> the Xl inner planes as a dense tensor (a view when B == 1)"""
    v = y[:, 1:-1]
    return v if v.is_contiguous() else v.contiguous()


class ShardedInstNormActFn(torch.autograd.Function):
    """y = lrelu(instance_norm(x) [+ res], slope) with the statistics of the WHOLE volume (x is this rank's slab)"""

    @staticmethod
    def forward(ctx, x, res, eps, slope, spg):
        mean_l, rstd_l = ops.instnorm_cl_stats(x, eps)
        var_l = rstd_l.pow(-2) - eps                          # the slab's biased variance
        st = torch.stack((mean_l, var_l + mean_l * mean_l))   # equal slabs: E[x], E[x^2] of the volume = the averages over the ranks
        _mean_over_group(st, spg)
        mean = st[0].contiguous()
        rstd = (st[1] - mean * mean).clamp_min_(0.0).add_(eps).rsqrt_().contiguous()
        y = ops.instnorm_cl_apply(x, mean, rstd, res, slope)
        ctx.save_for_backward(x, y, mean, rstd)
        ctx.slope, ctx.has_res, ctx.spg = slope, res is not None, spg
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd = ctx.saved_tensors
        m1, m2, dy = ops.instnorm_cl_bwd_sums(dy, y, x, mean, rstd, ctx.slope, ctx.has_res)
        st = torch.stack((m1, m2))
        _mean_over_group(st, ctx.spg)
        dx, dres = ops.instnorm_cl_bwd_apply(dy, y, x, mean, rstd, st[0].contiguous(), st[1].contiguous(), ctx.slope,
                                             ctx.has_res and ctx.needs_input_grad[1], ctx.has_res)
        return dx, dres, None, None, None


def sharded_instnorm_act(x, res, eps, slope, spg):
    return ShardedInstNormActFn.apply(x, res, eps, slope, spg)


def sharded_res_block(block, inp, spg):
    """monai UnetResBlock (simple/unetr_blocks.py:UnetResBlock over the same parameters) on this rank's slab [B, Xl, Y, Z, Cin]"""
    eps, slope = block.norm1.eps, block.lrelu.negative_slope
    w1, w2 = block.conv1.conv.weight, block.conv2.conv.weight
    c1 = _interior(HC.conv3x3x3(halo_pad(inp, spg), w1))
    y1 = sharded_instnorm_act(c1, None, eps, slope, spg)
    c2 = _interior(HC.conv3x3x3(halo_pad(y1, spg), w2))
    if block.downsample:
        c3 = HC.conv1x1x1(inp, block.conv3.conv.weight)
        res = sharded_instnorm_act(c3, None, block.norm3.eps, 1.0, spg)          # slope 1: the plain normalisation of the projection branch
    else:
        res = inp
    return sharded_instnorm_act(c2, res, block.norm2.eps, slope, spg)


def sharded_basic_block(blk, inp, spg):        # UnetrBasicBlock
    return sharded_res_block(blk.layer, inp, spg)


def sharded_prup_block(blk, x, spg):           # UnetrPrUpBlock
    x = HC.tconv2x2x2(x, blk.transp_conv_init.conv.weight)
    for b in blk.blocks:
        x = sharded_res_block(b[1], HC.tconv2x2x2(x, b[0].conv.weight), spg)
    return x


def sharded_up_block(blk, inp, skip, spg):     # UnetrUpBlock
    return sharded_res_block(blk.conv_block, HC.tconv2x2x2(inp, blk.transp_conv.conv.weight, skip), spg)


class ShardedDiceCEFn(torch.autograd.Function):
    """Dice + CE of the WHOLE volume from this rank's slab of the logits and labels; identical value on every rank of the group.  The
    gradient carries the factor P (see the module docstring)."""

    @staticmethod
    def forward(ctx, logits, labels, smooth_nr, smooth_dr, spg):
        lb = labels if labels.is_contiguous() else labels.contiguous()
        B, n = logits.shape[0], logits.shape[1]
        S = logits.numel() // (B * n)
        stats = _sum_over_group(ops.dice_ce_stats(logits, lb), spg)
        loss, dl = ops.dice_ce_from_stats(logits, lb, stats, S * spg.size, smooth_nr, smooth_dr, float(spg.size), want_grad=True)
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        if dl.is_contiguous():
            return dl * g.to(dl.dtype), None, None, None, None
        flat = dl.as_strided((dl.shape[0] * dl.stride(0),), (1,))           # the padded channels-last buffer behind the view
        return (flat * g.to(dl.dtype)).as_strided(dl.shape, dl.stride()), None, None, None, None


def sharded_dice_ce(logits_local, labels_local, spg, smooth_nr=1e-5, smooth_dr=1e-5):
    return ShardedDiceCEFn.apply(logits_local, labels_local, smooth_nr, smooth_dr, spg)


def local_slab(t, spg, dim):
    """this rank's X-slab of a whole-volume tensor (labels, inputs): dim = the X axis"""
    n = t.shape[dim] // spg.size
    return t.narrow(dim, spg.rank * n, n)

"""Convolutions of the UNETR decoder on the HIP kernels of csrc/conv3d.hip: autograd functions over CHANNELS-LAST bf16 activations
[B, X, Y, Z, C] with fp32 master weights in torch's own parameter layout (so state_dicts stay those of nn.Conv3d / nn.ConvTranspose3d).

Reference call sites: src/UCF_VIT/simple/arch.py:808-940 (monai UnetrBasicBlock / UnetrPrUpBlock / UnetrUpBlock / UnetOutBlock).  monai is
absent from the build container: PARITY UNPINNED against it; tests/test_conv3d.py checks every function here against torch.nn.functional.

  conv3x3x3(x, w)        Conv3d(k=3, s=1, p=1, bias=False): forward and data gradient by ucfvit_conv3d_fwd (implicit GEMM, MFMA), weight
                         gradient by ucfvit_conv3d_wgrad (deterministic two-stage sum)
  tconv2x2x2(x, w)       ConvTranspose3d(k=2, s=2, bias=False): GEMM [V, Cin] x [Cin, 8 Cout] + depth-to-space
  conv1x1x1(x, w, b)     pointwise Conv3d: the GEMM of nn.Linear over the voxel rows
  instnorm_act_cl        InstanceNorm3d (+ residual) + LeakyReLU on the channels-last layout
"""
import torch

from . import ops

_DIRECT_CIN = (8, 16)


def conv3_cin_supported(cin):
    return cin in _DIRECT_CIN or (cin > 0 and cin % 32 == 0)


def pack_conv3_weight(w):
    """w [Cout, Cin, 3, 3, 3] (any float dtype) -> bf16 [Cin / CPC, NTS, Cout, 32] as ucfvit_conv3d_fwd reads it (include/ucfvit_hip.h):
    CPC = min(Cin, 32) channels per contraction chunk, TPS = 32 / CPC taps per 32-wide step, NTS = ceil(27 / TPS) steps per chunk."""
    cout, cin = w.shape[0], w.shape[1]
    if not conv3_cin_supported(cin):
        raise ValueError(f"conv3x3x3: Cin must be 8, 16 or a multiple of 32, got {cin}")
    cpc = min(cin, 32)
    tps = 32 // cpc
    nts = -(-27 // tps)
    wt = w.reshape(cout, cin, 27).permute(2, 0, 1)                     # [27, Cout, Cin]
    if nts * tps > 27:
        wt = torch.cat((wt, wt.new_zeros((nts * tps - 27, cout, cin))), 0)
    wt = wt.reshape(nts, tps, cout, cin // cpc, cpc).permute(3, 0, 2, 1, 4)   # [chunk, step, Cout, tap in step, channel in chunk]
    return wt.reshape(cin // cpc, nts, cout, 32).to(torch.bfloat16).contiguous()


def pack_conv3_weight_dgrad(w):
    """the data gradient of a stride-1 'same' convolution is the convolution of dy with the flipped taps and swapped channel roles"""
    return pack_conv3_weight(w.transpose(0, 1).flip(2, 3, 4))


def unpack_conv3_wgrad(packed, cin, cout):
    """packed fp32 from ucfvit_conv3d_wgrad -> [Cout, Cin, 3, 3, 3]"""
    cpc = min(cin, 32)
    mb16 = 32 if cout % 32 == 0 else 16
    nbk16 = max(cpc, 16)
    t = packed.view(cout // mb16, cin // cpc, 27, mb16, nbk16)[..., :cpc]
    return t.permute(0, 3, 1, 4, 2).reshape(cout, cin, 3, 3, 3)


class Conv3x3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        cout, cin = w.shape[0], w.shape[1]
        if tuple(w.shape[2:]) != (3, 3, 3):
            raise ValueError("conv3x3x3: weight must be [Cout, Cin, 3, 3, 3]")
        if cout % 16:
            raise ValueError(f"conv3x3x3: Cout must be a multiple of 16, got {cout}")
        cin_x = x.shape[-1]
        if cin_x != cin:
            if not (cin < cin_x and cin_x == 8):
                raise ValueError(f"conv3x3x3: input has {cin_x} channels, weight expects {cin}")
            wk = torch.cat((w, w.new_zeros((cout, cin_x - cin, 3, 3, 3))), 1)   # zero-padded input channels (ops.pad_channels8)
        else:
            wk = w
        y = ops.conv3d_fwd(x, pack_conv3_weight(wk.detach()), cout)
        ctx.save_for_backward(x, w)
        ctx.cin_x = cin_x
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        cout, cin = w.shape[0], w.shape[1]
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if cin % 16 or ctx.cin_x != cin:
                raise RuntimeError(f"conv3x3x3: no data gradient for a {cin}-channel input (the kernel writes multiples of 16 channels)")
            dx = ops.conv3d_fwd(dy, pack_conv3_weight_dgrad(w.detach()), cin)
        if ctx.needs_input_grad[1]:
            dw = unpack_conv3_wgrad(ops.conv3d_wgrad(x, dy), ctx.cin_x, cout)[:, :cin].contiguous()
        return dx, dw


class TConv2x2x2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        cin, cout = w.shape[0], w.shape[1]
        if tuple(w.shape[2:]) != (2, 2, 2) or x.shape[-1] != cin:
            raise ValueError("tconv2x2x2: weight must be [Cin, Cout, 2, 2, 2] with Cin = the input's channels")
        B, X, Y, Z, _ = x.shape
        w2 = w.detach().permute(2, 3, 4, 1, 0).reshape(8 * cout, cin).to(torch.bfloat16).contiguous()      # rows (dx, dy, dz, co)
        cols = ops.linear_fwd(x.reshape(-1, cin), w2)
        ctx.save_for_backward(x, w2)
        ctx.wshape = tuple(w.shape)
        return ops.depth_to_space2(cols, B, X, Y, Z, cout)

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        cin, cout = ctx.wshape[0], ctx.wshape[1]
        dcols = ops.space_to_depth2(dy.contiguous())
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_dgrad(dcols, w2).view(x.shape)
        if ctx.needs_input_grad[1]:
            dw2 = ops.linear_wgrad(dcols, x.reshape(-1, cin))                                                # [8 Cout, Cin] fp32
            dw = dw2.view(2, 2, 2, cout, cin).permute(4, 3, 0, 1, 2).contiguous()
        return dx, dw


class Conv1x1x1Fn(torch.autograd.Function):
    """pointwise convolution = nn.Linear over the voxel rows; `pad_to` output columns (zero weights) keep every GEMM operand 16-byte
    aligned when Cout is small (the 1x1 output head)"""

    @staticmethod
    def forward(ctx, x, w, b, out_fp32):
        cout, cin = w.shape[0], w.shape[1]
        cin_x = x.shape[-1]
        w2 = w.detach().reshape(cout, cin).to(torch.bfloat16)
        if cin_x != cin:
            w2 = torch.cat((w2, w2.new_zeros((cout, cin_x - cin))), 1)
        npad = -(-cout // 8) * 8
        if npad != cout:
            w2 = torch.cat((w2, w2.new_zeros((npad - cout, cin_x))), 0)
        w2 = w2.contiguous()
        bias = None
        if b is not None:
            bias = b.detach().to(torch.bfloat16)
            if npad != cout:
                bias = torch.cat((bias, bias.new_zeros(npad - cout)))
        x2 = x.reshape(-1, cin_x)
        y = ops.gemm(x2, w2, x2.shape[0], npad, cin_x, ops.LAYOUT_KC, ops.LAYOUT_KC, bias=bias,
                     out_dtype=torch.float32 if out_fp32 else torch.bfloat16)
        ctx.save_for_backward(x, w2)
        ctx.dims = (cout, cin, npad, b is not None)
        return y.view(tuple(x.shape[:-1]) + (npad,))[..., :cout]

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        cout, cin, npad, has_b = ctx.dims
        cin_x = x.shape[-1]
        if npad != cout:
            d2 = torch.zeros((dy.numel() // cout, npad), dtype=torch.bfloat16, device=dy.device)
            d2[:, :cout] = dy.reshape(-1, cout)
        else:
            d2 = dy.to(torch.bfloat16).reshape(-1, cout).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_dgrad(d2, w2).view(x.shape)
        if ctx.needs_input_grad[1]:
            dw = ops.linear_wgrad(d2, x.reshape(-1, cin_x))[:cout, :cin].reshape(cout, cin, 1, 1, 1).contiguous()
        if has_b and ctx.needs_input_grad[2]:
            db = ops.colsum(d2)[:cout].contiguous()
        return dx, dw, db, None


class InstNormActCLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, eps, slope):
        y, mean, rstd = ops.instnorm_cl_fwd(x, res, eps, slope)
        ctx.save_for_backward(x, y, mean, rstd)
        ctx.slope = slope
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd = ctx.saved_tensors
        dx, dres = ops.instnorm_cl_bwd(dy.contiguous(), y, x, mean, rstd, ctx.slope, ctx.has_res and ctx.needs_input_grad[1])
        return dx, dres, None, None


def conv3x3x3(x, w):
    return Conv3x3x3Fn.apply(x, w)


def tconv2x2x2(x, w):
    return TConv2x2x2Fn.apply(x, w)


def conv1x1x1(x, w, b=None, out_fp32=False):
    return Conv1x1x1Fn.apply(x, w, b, out_fp32)


def instnorm_act_cl(x, res=None, eps=1e-5, slope=0.01):
    return InstNormActCLFn.apply(x, res, eps, slope)

"""Convolutions of the UNETR decoder on the HIP kernels of csrc/conv3d.hip: autograd functions over CHANNELS-LAST bf16 activations
[B, X, Y, Z, C] with fp32 master weights in torch's own parameter layout (so state_dicts stay those of nn.Conv3d / nn.ConvTranspose3d).

Reference call sites: src/UCF_VIT/simple/arch.py:808-940 (monai UnetrBasicBlock / UnetrPrUpBlock / UnetrUpBlock / UnetOutBlock).  monai is
absent from the build container: PARITY UNPINNED against it; tests/test_conv3d.py checks every function here against torch.nn.functional.

  conv3x3x3(x, w)        Conv3d(k=3, s=1, p=1, bias=False): forward and data gradient by ucfvit_conv3d_fwd (implicit GEMM, MFMA), weight
                         gradient by ucfvit_conv3d_wgrad (deterministic two-stage sum)
  tconv2x2x2(x, w)       ConvTranspose3d(k=2, s=2, bias=False): pointwise layer [V, Cin] -> [V, 8 Cout] (GEMM or 1x1x1 kernel) + depth-to-space
  conv1x1x1(x, w, b)     pointwise Conv3d: the 1x1x1 instance of the convolution kernels (tall-skinny voxel rows), the tiled GEMM of
                         nn.Linear when both channel counts reach 128
  instnorm_act_cl        InstanceNorm3d (+ residual) + LeakyReLU on the channels-last layout
"""
import torch

from . import ops

_DIRECT_CIN = (8, 16)
GEMM_MIN = 128      # a pointwise / transposed layer goes to the tiled GEMM only when both channel counts reach its 128-wide tiles


def conv3_cin_supported(cin):
    return cin in _DIRECT_CIN or (cin > 0 and cin % 32 == 0)


def _ceil_to(n, m):
    return -(-n // m) * m


def _pad_cin(c):
    """smallest channel count >= c the convolution kernels accept as an input operand"""
    return 8 if c <= 8 else 16 if c <= 16 else _ceil_to(c, 32)


def pack_conv_weight(w):
    """w [Cout, Cin, k, k, k] (k = 3 or 1, any float dtype) -> bf16 [Cin / CPC, NTS, Cout, 32] as ucfvit_conv3d_fwd reads it
    (include/ucfvit_hip.h): CPC = min(Cin, 32) channels per contraction chunk, TPS = 32 / CPC taps per 32-wide step, NTS = ceil(k^3 / TPS)."""
    cout, cin = w.shape[0], w.shape[1]
    nt = w.shape[2] * w.shape[3] * w.shape[4]
    if not conv3_cin_supported(cin):
        raise ValueError(f"convolution kernels: Cin must be 8, 16 or a multiple of 32, got {cin}")
    if nt not in (1, 27):
        raise ValueError("convolution kernels: kernel size must be 1 or 3")
    cpc = min(cin, 32)
    tps = 32 // cpc
    nts = -(-nt // tps)
    wt = w.reshape(cout, cin, nt).permute(2, 0, 1)                     # [taps, Cout, Cin]
    if nts * tps > nt:
        wt = torch.cat((wt, wt.new_zeros((nts * tps - nt, cout, cin))), 0)
    wt = wt.reshape(nts, tps, cout, cin // cpc, cpc).permute(3, 0, 2, 1, 4)   # [chunk, step, Cout, tap in step, channel in chunk]
    return wt.reshape(cin // cpc, nts, cout, 32).to(torch.bfloat16).contiguous()


pack_conv3_weight = pack_conv_weight


def pack_conv3_weight_dgrad(w):
    """the data gradient of a stride-1 'same' convolution is the convolution of dy with the flipped taps and swapped channel roles"""
    return pack_conv_weight(w.transpose(0, 1).flip(2, 3, 4))


def unpack_conv_wgrad(packed, cin, cout, ksize=3):
    """packed fp32 from ucfvit_conv3d_wgrad -> [Cout, Cin, k, k, k]"""
    cpc = min(cin, 32)
    mb16 = 32 if cout % 32 == 0 else 16
    nbk16 = max(cpc, 16)
    nt = ksize ** 3
    t = packed.view(cout // mb16, cin // cpc, nt, mb16, nbk16)[..., :cpc]
    return t.permute(0, 3, 1, 4, 2).reshape(cout, cin, ksize, ksize, ksize)


unpack_conv3_wgrad = unpack_conv_wgrad


def _pad_last(t, c):
    """[..., c0] -> contiguous bf16 [..., c] with zero columns appended"""
    if t.shape[-1] == c and t.dtype == torch.bfloat16:
        return t.contiguous()
    if c == 8 and t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.stride(-1) == 1:
        try:
            return ops.pad_rows8(t)                    # one pass (the output head's logits gradient: 67 M voxel rows at 512 x 512 x 128)
        except ValueError:
            pass                                       # rows without a common stride: the general path below
    out = torch.zeros(tuple(t.shape[:-1]) + (c,), dtype=torch.bfloat16, device=t.device)
    out[..., :t.shape[-1]] = t
    return out


def _colsum_narrow(d2):
    """column sums of a tall matrix with few columns: fold rows into the column axis first so that every lane of ucfvit_colsum has a column
    (its wave covers 512 bf16 columns), then add the folds"""
    V, C = d2.shape
    f = 1
    while C * f < 512 and V % (2 * f) == 0:
        f *= 2
    return ops.colsum(d2.view(V // f, C * f)).view(f, C).sum(0)


def _pointwise_fwd(x, w2, bias, cout_store, out_dtype=torch.bfloat16, accumulate_into=None, stats_eps=None):
    """x [B, X, Y, Z, K] bf16 (K one of the kernels' input widths), w2 [N, K] float -> [B, X, Y, Z, cout_store] through the 1x1x1 kernel"""
    n16 = _ceil_to(w2.shape[0], 16)
    if n16 != w2.shape[0]:
        w2 = torch.cat((w2, w2.new_zeros((n16 - w2.shape[0], w2.shape[1]))), 0)
        if bias is not None:
            bias = torch.cat((bias, bias.new_zeros(n16 - bias.numel())))
    packed = pack_conv_weight(w2.reshape(n16, w2.shape[1], 1, 1, 1))
    return ops.conv3d_fwd(x, packed, n16, ksize=1, bias=None if bias is None else bias.float().contiguous(), cout_store=cout_store,
                          out_dtype=out_dtype, accumulate_into=accumulate_into, stats_eps=stats_eps)


def _pointwise_wgrad(x, dy):
    """x [.., K], dy [.., N] channels-last bf16 (kernel input widths) -> fp32 [N, K] = sum over voxels of dy^T x"""
    K, N = x.shape[-1], dy.shape[-1]
    if N % 16 == 0:
        return unpack_conv_wgrad(ops.conv3d_wgrad(x, dy, ksize=1), K, N, 1).reshape(N, K)
    if K % 16:
        raise ValueError(f"pointwise weight gradient: one of the channel counts ({K}, {N}) must be a multiple of 16")
    return unpack_conv_wgrad(ops.conv3d_wgrad(dy, x, ksize=1), N, K, 1).reshape(K, N).t()      # roles swapped: [K, N] = x^T dy


def conv3_wgrad(x, dy, cin_x, cout):
    """weight gradient [Cout, cin_x, 3, 3, 3] of a 3x3x3 convolution from its input x [.., cin_x] and output gradient dy [.., Cout].
    For a 16-channel output under a wider input the operands swap roles: dW[tap][co][ci] = sum_v dy[v][co] x[v + tap][ci]
    = sum_u x[u][ci] dy[u - tap][co], i.e. the weight gradient of the convolution dy -> x with the taps mirrored and the channel roles
    exchanged.  The kernel shifts its INPUT operand per tap and loads its output-gradient operand once per voxel row, so the shifted
    operand should be the narrow one: 18 instead of 30 transposed LDS reads per 14 MFMAs for 32 -> 16 channels."""
    if cout == 16 and cin_x >= 32:
        t = unpack_conv_wgrad(ops.conv3d_wgrad(dy, x), cout, cin_x)          # [cin_x, Cout, 3, 3, 3] of the mirrored problem
        return t.flip(2, 3, 4).transpose(0, 1)
    return unpack_conv_wgrad(ops.conv3d_wgrad(x, dy), cin_x, cout)


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
        y = ops.conv3d_fwd(x, pack_conv_weight(wk.detach()), cout)
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
            dw = conv3_wgrad(x, dy, ctx.cin_x, cout)[:, :cin].contiguous()
        return dx, dw


class TConv2x2x2Fn(torch.autograd.Function):
    """ConvTranspose3d(k=2, s=2): out[2v + d][co] = sum_ci x[v][ci] w[ci][co][d] = a pointwise layer to 8 Cout channels + depth-to-space.
    Large channel counts (the 768-wide token maps) use the tiled GEMM; small ones the 1x1x1 convolution kernel.
    With `skip` ([B, 2X, 2Y, 2Z, Cs]) the result is the channel concatenation (out, skip) of monai's UnetrUpBlock: the depth-to-space pass
    writes the up-sampled map straight into the first Cout channels of the concatenation, and the backward pass reads the two halves of the
    concatenation's gradient in place (no torch.cat / slice copies of the largest tensors of the decoder)."""

    @staticmethod
    def forward(ctx, x, w, skip):
        cin, cout = w.shape[0], w.shape[1]
        if tuple(w.shape[2:]) != (2, 2, 2) or x.shape[-1] != cin:
            raise ValueError("tconv2x2x2: weight must be [Cin, Cout, 2, 2, 2] with Cin = the input's channels")
        B, X, Y, Z, _ = x.shape
        w2 = w.detach().permute(2, 3, 4, 1, 0).reshape(8 * cout, cin)                                        # rows (dx, dy, dz, co)
        ctx.small = cin < GEMM_MIN
        if ctx.small:
            if not conv3_cin_supported(cin) or (8 * cout) % 32:
                raise ValueError(f"tconv2x2x2: unsupported channel counts {cin} -> {cout}")
            cols = _pointwise_fwd(x, w2, None, 8 * cout)
        else:
            w2 = w2.to(torch.bfloat16).contiguous()
            cols = ops.linear_fwd(x.reshape(-1, cin), w2)
        ctx.save_for_backward(x, w2)
        ctx.wshape = tuple(w.shape)
        ctx.cs = 0
        if skip is None:
            return ops.depth_to_space2(cols.view(-1, 8 * cout), B, X, Y, Z, cout)
        if tuple(skip.shape[:4]) != (B, 2 * X, 2 * Y, 2 * Z) or skip.dtype != torch.bfloat16 or cout % 8 or skip.shape[-1] % 8:
            raise ValueError("tconv2x2x2: skip must be a channels-last bf16 map of the up-sampled extent")
        ctx.cs = skip.shape[-1]
        cat = torch.empty((B, 2 * X, 2 * Y, 2 * Z, cout + ctx.cs), dtype=torch.bfloat16, device=x.device)
        ops.depth_to_space2(cols.view(-1, 8 * cout), B, X, Y, Z, cout, out=cat[..., :cout], skip=skip.contiguous())   # whole rows in one pass
        return cat

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        cin, cout = ctx.wshape[0], ctx.wshape[1]
        dskip = None
        if ctx.cs:
            if ctx.needs_input_grad[2]:
                dskip = dy[..., cout:]                      # a view: the consumer (normalisation backward) reads the slice in place
            dy = dy[..., :cout]
        dcols = ops.space_to_depth2(dy)
        dx = dw = None
        if ctx.small:
            dcols = dcols.view(tuple(x.shape[:-1]) + (8 * cout,))
            if ctx.needs_input_grad[0]:
                dx = _pointwise_fwd(dcols, w2.t(), None, cin)
            if ctx.needs_input_grad[1]:
                dw2 = _pointwise_wgrad(x, dcols)
        else:
            if ctx.needs_input_grad[0]:
                dx = ops.linear_dgrad(dcols, w2).view(x.shape)
            if ctx.needs_input_grad[1]:
                dw2 = ops.linear_wgrad(dcols, x.reshape(-1, cin))                                            # [8 Cout, Cin] fp32
        if ctx.needs_input_grad[1]:
            dw = dw2.reshape(2, 2, 2, cout, cin).permute(4, 3, 0, 1, 2).contiguous()
        return dx, dw, dskip


class Conv1x1x1Fn(torch.autograd.Function):
    """pointwise convolution over channels-last voxel rows.  The input may carry more channels than the weight (the zero-padded 8-channel
    input volume); the output has exactly Cout channels (fp32 on request: the logits of the output head)."""

    @staticmethod
    def forward(ctx, x, w, b, out_fp32):
        cout, cin = w.shape[0], w.shape[1]
        cin_x = x.shape[-1]
        w2 = w.detach().reshape(cout, cin)
        if cin_x != cin:
            w2 = torch.cat((w2, w2.new_zeros((cout, cin_x - cin))), 1)
        odt = torch.float32 if out_fp32 else torch.bfloat16
        ctx.small = cin_x < GEMM_MIN or cout < GEMM_MIN
        if ctx.small:
            if not conv3_cin_supported(cin_x):
                raise ValueError(f"conv1x1x1: unsupported input channel count {cin_x}")
            y = _pointwise_fwd(x, w2, None if b is None else b.detach(), cout, odt)
        else:
            w2 = w2.to(torch.bfloat16).contiguous()
            x2 = x.reshape(-1, cin_x)
            y = ops.gemm(x2, w2, x2.shape[0], cout, cin_x, ops.LAYOUT_KC, ops.LAYOUT_KC, bias=None if b is None else b.detach().to(torch.bfloat16),
                         out_dtype=odt).view(tuple(x.shape[:-1]) + (cout,))
        ctx.save_for_backward(x, w2)
        ctx.dims = (cout, cin, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        cout, cin, has_b = ctx.dims
        cin_x = x.shape[-1]
        dx = dw = db = None
        if ctx.small:
            dyp = _pad_last(dy, _pad_cin(cout))                       # bf16, channel count the kernels take as an input operand
            if ctx.needs_input_grad[0]:
                wt = torch.cat((w2.t(), w2.new_zeros((cin_x, dyp.shape[-1] - cout))), 1) if dyp.shape[-1] != cout else w2.t()
                dx = _pointwise_fwd(dyp, wt, None, cin_x)
            if ctx.needs_input_grad[1]:
                dw = _pointwise_wgrad(x, dyp)[:cout, :cin].reshape(cout, cin, 1, 1, 1).contiguous()
            if has_b and ctx.needs_input_grad[2]:
                db = _colsum_narrow(dyp.view(-1, dyp.shape[-1]))[:cout].contiguous()
        else:
            d2 = dy.to(torch.bfloat16).reshape(-1, cout).contiguous()
            if ctx.needs_input_grad[0]:
                dx = ops.linear_dgrad(d2, w2).view(x.shape)
            if ctx.needs_input_grad[1]:
                dw = ops.linear_wgrad(d2, x.reshape(-1, cin_x))[:, :cin].reshape(cout, cin, 1, 1, 1).contiguous()
            if has_b and ctx.needs_input_grad[2]:
                db = ops.colsum(d2)
        return dx, dw, db, None


class UnetResBlockFn(torch.autograd.Function):
    """monai UnetResBlock (kernel 3, stride 1) as ONE autograd node over the channels-last kernels:
        out = lrelu(norm(conv3(lrelu(norm(conv3(inp, w1))), w2)) + res),   res = norm(conv1(inp, w3)) if w3 is given else inp
    Against the chain of per-layer functions: the two data gradients that meet in `inp` are summed in the second kernel's epilogue (no
    torch add over the largest tensors), the normalised 1x1x1 branch is not kept for the backward pass, and the backward of the
    residual-free normalisations never reads their outputs."""

    @staticmethod
    def forward(ctx, inp, w1, w2, w3, eps, slope):
        cout, cin = w1.shape[0], w1.shape[1]
        cin_x = inp.shape[-1]
        if cout % 16 or tuple(w1.shape[2:]) != (3, 3, 3) or tuple(w2.shape) != (cout, cout, 3, 3, 3):
            raise ValueError("UnetResBlock: 3x3x3 weights [Cout, Cin, 3, 3, 3] / [Cout, Cout, 3, 3, 3] with Cout a multiple of 16")
        if cin_x != cin and not (cin < cin_x == 8):
            raise ValueError(f"UnetResBlock: input has {cin_x} channels, weight expects {cin}")
        if w3 is None and cin_x != cout:
            raise ValueError("UnetResBlock: an identity residual needs Cin == Cout")
        w1k = w1.detach() if cin_x == cin else torch.cat((w1.detach(), w1.new_zeros((cout, cin_x - cin, 3, 3, 3))), 1)
        # every convolution hands over the instance-norm statistics of its output (epilogue by-product where the kernel has one)
        c1, m1, r1 = ops.conv3d_fwd(inp, pack_conv_weight(w1k), cout, stats_eps=eps)
        y1 = ops.instnorm_cl_apply(c1, m1, r1, None, slope)
        c2, m2, r2 = ops.conv3d_fwd(y1, pack_conv_weight(w2.detach()), cout, stats_eps=eps)
        c3 = m3 = r3 = w3k = None
        if w3 is not None:
            w3k = w3.detach().reshape(cout, cin)
            if cin_x != cin:
                w3k = torch.cat((w3k, w3k.new_zeros((cout, cin_x - cin))), 1)
            if cout % 16:
                raise ValueError("UnetResBlock: Cout must be a multiple of 16")
            c3, m3, r3 = _pointwise_fwd(inp, w3k, None, cout, stats_eps=eps)
            out = ops.instnorm_cl_apply2(c2, m2, r2, c3, m3, r3, slope)        # lrelu(norm(c2) + norm(c3)): the normalised branch is never written
        else:
            out = ops.instnorm_cl_apply(c2, m2, r2, inp, slope)
        ctx.save_for_backward(inp, w1, w2, w3, c1, y1, c2, c3, out, m1, r1, m2, r2, m3, r3)
        ctx.slope = slope
        return out

    @staticmethod
    def backward(ctx, dout):
        inp, w1, w2, w3, c1, y1, c2, c3, out, m1, r1, m2, r2, m3, r3 = ctx.saved_tensors
        cout, cin = w1.shape[0], w1.shape[1]
        cin_x = inp.shape[-1]
        slope = ctx.slope
        need_dinp = ctx.needs_input_grad[0]
        if w3 is not None:
            dc2, dc3 = ops.instnorm_cl_bwd2(dout, out, c2, m2, r2, c3, m3, r3, slope)                   # both normalisations in one pair of passes
            dres = None
        else:
            dc2, dres = ops.instnorm_cl_bwd(dout, out, c2, m2, r2, slope, want_dres=need_dinp, had_res=True)
        dw2 = conv3_wgrad(y1, dc2, cout, cout).contiguous() if ctx.needs_input_grad[2] else None
        dy1 = ops.conv3d_fwd(dc2, pack_conv3_weight_dgrad(w2.detach()), cout)
        del dc2
        dc1, _ = ops.instnorm_cl_bwd(dy1, c1, c1, m1, r1, slope, want_dres=False, had_res=False)       # no residual: the output is not read
        del dy1
        dw1 = conv3_wgrad(inp, dc1, cin_x, cout)[:, :cin].contiguous() if ctx.needs_input_grad[1] else None
        dw3 = dinp = None
        if w3 is not None:
            if ctx.needs_input_grad[3]:
                dw3 = _pointwise_wgrad(inp, dc3)[:cout, :cin].reshape(cout, cin, 1, 1, 1).contiguous()
            if need_dinp:
                w3k = w3.detach().reshape(cout, cin)
                dinp = _pointwise_fwd(dc3, w3k.t(), None, cin_x)
        elif need_dinp:
            dinp = dres
        if need_dinp:
            if cin % 16 or cin_x != cin:
                raise RuntimeError(f"UnetResBlock: no data gradient for a {cin}-channel input")
            ops.conv3d_fwd(dc1, pack_conv3_weight_dgrad(w1.detach()), cin, accumulate_into=dinp)            # dinp += the 3x3x3 branch
        return dinp, dw1, dw2, dw3, None, None


def unet_res_block(inp, w1, w2, w3=None, eps=1e-5, slope=0.01):
    return UnetResBlockFn.apply(inp, w1, w2, w3, eps, slope)


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
        dx, dres = ops.instnorm_cl_bwd(dy, y, x, mean, rstd, ctx.slope, ctx.has_res and ctx.needs_input_grad[1], had_res=ctx.has_res)
        return dx, dres, None, None


def conv3x3x3(x, w):
    return Conv3x3x3Fn.apply(x, w)


def tconv2x2x2(x, w, skip=None):
    """skip given: returns the channel concatenation (transposed convolution, skip) — see TConv2x2x2Fn"""
    return TConv2x2x2Fn.apply(x, w, skip)


def conv1x1x1(x, w, b=None, out_fp32=False):
    return Conv1x1x1Fn.apply(x, w, b, out_fp32)


def instnorm_act_cl(x, res=None, eps=1e-5, slope=0.01):
    return InstNormActCLFn.apply(x, res, eps, slope)

"""Tensor-level wrappers over the C ABI (no autograd here).  Every function enqueues HIP kernels on torch's
current stream and returns torch tensors that own the outputs.  PyTorch is used for device memory and streams only.
"""
import ctypes
import os
import math

import torch

from . import lib as _l
from .lib import ACT_GELU, ACT_GELU_GRAD, ACT_GELU_SAVE_DERIV, ACT_MUL_AUX, ACT_NONE, BF16, F32, LAYOUT_KC, LAYOUT_KS

_DT = {torch.float32: F32, torch.bfloat16: BF16}
_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}


def dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"UCF_VIT HIP ops support float32 and bfloat16 tensors, got {t.dtype}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, rows_ok=False):
    """rows_ok: a 2-D matrix whose rows are contiguous but padded (stride(0) >= shape[1]) is accepted: the C ABI takes leading dimensions"""
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the MI355X (cuda) device, got {t.device}. "
                           "UCF_VIT operators run only through libucfvit_hip.so; there is no CPU path.")
    if not t.is_contiguous():
        if not (rows_ok and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]):
            raise RuntimeError(f"{name}: tensor must be contiguous")
    return t


def _p(t):
    return None if t is None else t.data_ptr()


_workspaces = {}


def workspace(nbytes, device):
    """fp32 scratch, cached per (device, stream): used and consumed inside one ABI call sequence on that stream."""
    key = (device.index, _stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(max(int(nbytes) // 4 + 1, 1 << 16), dtype=torch.float32, device=device)
        _workspaces[key] = ws
    return ws


# Dynamic tile schedule of the persistent GEMM (desc.sched_state, include/ucfvit_hip.h): off for a single process (every CU is the GEMM's,
# the static order costs nothing), switched on by HipDataParallel / the tensor-parallel blocks, whose collectives hold CUs while GEMMs run.
# UCFVIT_GEMM_DYNAMIC=1|0 forces it.  One state block per (device, stream): launches of one stream are ordered, so they can share it.
_dynamic_sched = os.environ.get("UCFVIT_GEMM_DYNAMIC", "") == "1"
_sched_states = {}


def set_dynamic_tile_schedule(on):
    global _dynamic_sched
    if os.environ.get("UCFVIT_GEMM_DYNAMIC", "") in ("0", "1"):
        return _dynamic_sched                      # forced by the environment
    _dynamic_sched = bool(on)
    return _dynamic_sched


def _sched_state(device):
    if not _dynamic_sched:
        return None
    key = (device.index, _stream())
    st = _sched_states.get(key)
    if st is None:
        st = torch.zeros(_l.GEMM_SCHED_BYTES // 4, dtype=torch.int32, device=device)     # zeroed once; every launch leaves it zeroed
        _sched_states[key] = st
    return st.data_ptr()


def mfma_probe(iters=10000):
    """diagnostic: one launch of a pure bf16 MFMA stream on the current stream; returns the FLOPs it executes"""
    L = _l.load()
    sink = workspace(1024, torch.device("cuda", torch.cuda.current_device()))
    flops = L.ucfvit_mfma_probe(sink.data_ptr(), int(iters), _stream())
    if flops < 0:
        _l.check(int(flops), "ucfvit_mfma_probe")
    return flops


# ------------------------------------------------------------------------------------------------ GEMM
def gemm(A, B, M, N, K, a_layout, b_layout, out=None, out_dtype=None, bias=None, residual=None, act=ACT_NONE,
         aux_in=None, aux_out=None, accumulate=False, alpha=1.0, c_colsum=None, c_colsum_accumulate=False):
    """C[M,N] = epilogue(alpha * op(A)·op(B)); A, B are 2-D row-major tensors (see include/ucfvit_hip.h).
    c_colsum: optional fp32 [N] tensor that receives (+)= the column sums of C — as a by-product of the epilogue where the library
    has one (desc.c_colsum_partial + ucfvit_reduce_rows), else by a separate ucfvit_colsum pass over C."""
    L = _l.load()
    _chk(A, "gemm.A", rows_ok=True), _chk(B, "gemm.B", rows_ok=True)
    if A.dtype != B.dtype:
        raise TypeError("gemm: A and B must have the same dtype")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or A.dtype, device=A.device)
    d = _l.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
    d.bias, d.residual, d.aux_in, d.aux_out = _p(bias), _p(residual), _p(aux_in), _p(aux_out)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = A.stride(0), B.stride(0), out.stride(0)
    d.ldr = residual.stride(0) if residual is not None else 0
    aux = aux_in if aux_in is not None else aux_out
    d.ldaux = aux.stride(0) if aux is not None else 0
    d.a_layout, d.b_layout = a_layout, b_layout
    d.dtype, d.out_dtype = dt(A), dt(out)
    d.act, d.accumulate, d.alpha = act, 1 if accumulate else 0, alpha
    d.workspace, d.workspace_bytes = None, 0
    d.c_colsum_partial = None
    d.sched_state = _sched_state(A.device)
    cs_rows, cs_part = 0, None
    if c_colsum is not None:
        _chk(c_colsum, "gemm.c_colsum")
        if c_colsum.dtype != torch.float32 or c_colsum.numel() != N or not c_colsum.is_contiguous():
            raise TypeError("gemm: c_colsum must be a contiguous fp32 tensor of N elements")
        cs_rows = L.ucfvit_gemm_colsum_rows(ctypes.byref(d))
        if cs_rows > 0:
            cs_part = workspace(cs_rows * N * 4, A.device)
            d.c_colsum_partial = cs_part.data_ptr()
    if act == ACT_NONE and bias is None and residual is None:      # only epilogue-free GEMMs (weight gradients) split K
        need = L.ucfvit_gemm_workspace(ctypes.byref(d))
        if need > 0:
            ws = workspace(need, A.device)
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    _l.check(L.ucfvit_gemm(ctypes.byref(d), _stream()), "ucfvit_gemm")
    if c_colsum is not None:
        if cs_rows > 0:
            _l.check(L.ucfvit_reduce_rows(cs_part.data_ptr(), c_colsum.data_ptr(), cs_rows, N, 1 if c_colsum_accumulate else 0, _stream()),
                     "ucfvit_reduce_rows")
        else:
            colsum(out, out=c_colsum, accumulate=c_colsum_accumulate)
    return out


def wgrad_grouped(items):
    """items: list of (dy2 [M,N], x2 [M,K], out fp32 [N,K] or None, accumulate) -> list of dW tensors; ONE launch when groupable"""
    L = _l.load()
    n = len(items)
    outs = []
    arr = (_l.GemmDesc * n)()
    for i, (dy2, x2, out, acc) in enumerate(items):
        _chk(dy2, "wgrad.dy", rows_ok=True), _chk(x2, "wgrad.x", rows_ok=True)
        Mtok, N = dy2.shape
        K = x2.shape[1]
        if out is None:
            out = torch.empty((N, K), dtype=torch.float32, device=dy2.device)
        outs.append(out)
        d = arr[i]
        d.A, d.B, d.C = dy2.data_ptr(), x2.data_ptr(), out.data_ptr()
        d.bias = d.residual = d.aux_in = d.aux_out = None
        d.M, d.N, d.K = N, K, Mtok
        d.lda, d.ldb, d.ldc, d.ldr, d.ldaux = dy2.stride(0), x2.stride(0), out.stride(0), 0, 0
        d.a_layout, d.b_layout = LAYOUT_KS, LAYOUT_KS
        d.dtype, d.out_dtype = dt(dy2), F32
        d.act, d.accumulate, d.alpha = ACT_NONE, 1 if acc else 0, 1.0
        d.workspace, d.workspace_bytes = None, 0
        d.c_colsum_partial = None
        d.sched_state = _sched_state(dy2.device)
    _l.check(L.ucfvit_gemm_grouped(arr, n, _stream()), "ucfvit_gemm_grouped")
    return outs


def linear_fwd(x2, w, b=None, act=ACT_NONE, residual=None, aux_out=None, out=None):
    """y[M,N] = act(x2[M,K]·w[N,K]ᵀ + b) + residual   (nn.Linear forward, building_blocks.py:123,127,159,190)"""
    M, K = x2.shape
    N = w.shape[0]
    return gemm(x2, w, M, N, K, LAYOUT_KC, LAYOUT_KC, out=out, bias=b, residual=residual, act=act,
                aux_out=aux_out)


def _dgrad_act(act_grad_aux, aux_is_deriv):
    if act_grad_aux is None:
        return ACT_NONE
    return ACT_MUL_AUX if aux_is_deriv else ACT_GELU_GRAD


def linear_dgrad(dy2, w, act_grad_aux=None, out=None, aux_is_deriv=False, c_colsum=None, c_colsum_accumulate=False):
    """dx[M,K] = dy2[M,N]·w[N,K], optionally times gelu'(aux) (aux = the fc1 pre-activation) or times aux itself
    (aux_is_deriv: aux = gelu' saved by the forward epilogue, ACT_GELU_SAVE_DERIV)"""
    M, N = dy2.shape
    K = w.shape[1]
    return gemm(dy2, w, M, K, N, LAYOUT_KC, LAYOUT_KS, out=out, act=_dgrad_act(act_grad_aux, aux_is_deriv), aux_in=act_grad_aux,
                c_colsum=c_colsum, c_colsum_accumulate=c_colsum_accumulate)


def linear_wgrad(dy2, x2, out=None, accumulate=False):
    """dW[N,K] (fp32) = dy2[M,N]ᵀ·x2[M,K]"""
    M, N = dy2.shape
    K = x2.shape[1]
    return gemm(dy2, x2, N, K, M, LAYOUT_KS, LAYOUT_KS, out=out, out_dtype=torch.float32, accumulate=accumulate)


def reduce_rows(partial, out, accumulate=False):
    """out[N] (+)= sum over the rows of partial [R, N] (fp32)"""
    L = _l.load()
    R, N = partial.shape
    _l.check(L.ucfvit_reduce_rows(partial.data_ptr(), out.data_ptr(), R, N, 1 if accumulate else 0, _stream()), "ucfvit_reduce_rows")
    return out


def colsum(x2, out=None, accumulate=False):
    L = _l.load()
    _chk(x2, "colsum.x", rows_ok=True)
    M, N = x2.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x2.device)
    ws = workspace(L.ucfvit_colsum_workspace(M, N), x2.device)
    _l.check(L.ucfvit_colsum(x2.data_ptr(), out.data_ptr(), M, N, x2.stride(0), 1 if accumulate else 0, ws.data_ptr(), dt(x2),
                             _stream()), "ucfvit_colsum")
    return out


# ------------------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x2, gamma, beta, eps):
    L = _l.load()
    _chk(x2, "layernorm.x")
    rows, D = x2.shape
    y = torch.empty_like(x2)
    mean = torch.empty(rows, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2.device)
    _l.check(L.ucfvit_layernorm_fwd(x2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                    rows, D, eps, dt(x2), _stream()), "ucfvit_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy2, x2, gamma, mean, rstd, dres=None, dgamma=None, dbeta=None, accumulate=False, dx_colsum=None,
                  dx_colsum_accumulate=False):
    """-> dx, dgamma, dbeta; dx_colsum: optional fp32 [D] that receives (+)= the column sums of dx (a bias gradient, see the header)"""
    L = _l.load()
    _chk(dy2, "layernorm_bwd.dy")
    rows, D = x2.shape
    dx = torch.empty_like(x2)
    if dgamma is None:
        dgamma = torch.empty(D, dtype=torch.float32, device=x2.device)
    if dbeta is None:
        dbeta = torch.empty(D, dtype=torch.float32, device=x2.device)
    ws = workspace(L.ucfvit_layernorm_bwd_workspace(rows, D), x2.device)
    _l.check(L.ucfvit_layernorm_bwd(dy2.data_ptr(), x2.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(dres),
                                    dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), rows, D, 1 if accumulate else 0,
                                    _p(dx_colsum), 1 if dx_colsum_accumulate else 0, ws.data_ptr(), dt(x2), _stream()),
             "ucfvit_layernorm_bwd")
    return dx, dgamma, dbeta


# ------------------------------------------------------------------------------------------------ attention
def attention_fwd(qkv, B, N, H, dh, scale):
    """qkv: [B*N, 3*H*dh] (the qkv Linear's output) -> out [B*N, H*dh], lse [B,H,N]"""
    L = _l.load()
    _chk(qkv, "attention.qkv")
    out = torch.empty((B * N, H * dh), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    _l.check(L.ucfvit_attention_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, H, dh, scale, dt(qkv), _stream()),
             "ucfvit_attention_fwd")
    return out, lse


def attention_bwd_colsum_supported(B, N, H, dh, dtype):
    return bool(_l.load().ucfvit_attention_bwd_colsum_supported(B, N, H, dh, _DT[dtype]))


def attention_bwd(qkv, out, dout, lse, B, N, H, dh, scale, want_colsum=False):
    """-> dqkv, or (dqkv, partial) with want_colsum: partial fp32 [B, 2 H dh] = the column sums of dQ over each batch element's tokens and
    zeros for dK (include/ucfvit_hip.h: the K third of the qkv bias gradient is identically 0, the V third is the column sum of dout), or
    None where the shape runs the streaming kernels (the caller sums dqkv itself)"""
    L = _l.load()
    _chk(dout, "attention_bwd.dout")
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    if want_colsum and L.ucfvit_attention_bwd_colsum_supported(B, N, H, dh, dt(qkv)):
        part = torch.empty((B, 2 * H * dh), dtype=torch.float32, device=qkv.device)
        _l.check(L.ucfvit_attention_bwd_colsum(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), delta.data_ptr(),
                                               part.data_ptr(), B, N, H, dh, scale, dt(qkv), _stream()), "ucfvit_attention_bwd_colsum")
        return dqkv, part
    _l.check(L.ucfvit_attention_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), delta.data_ptr(),
                                    B, N, H, dh, scale, dt(qkv), _stream()), "ucfvit_attention_bwd")
    return (dqkv, None) if want_colsum else dqkv


def attention_cross_fwd(q, k, v, B, Nq, Nk, H, dh, scale):
    """q [B*Nq, ldq], k / v [B*Nk, ldkv] (2-D views, head h at columns h*dh..) -> (out [B*Nq, H*dh], lse [B, H, Nq] in log2 units)"""
    L = _l.load()
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, "attention_cross_fwd." + n, rows_ok=True)
    if k.stride(0) != v.stride(0) or k.dtype != q.dtype or v.dtype != q.dtype:
        raise ValueError("attention_cross_fwd: k and v must share their row stride, and all operands their dtype")
    out = torch.empty((B * Nq, H * dh), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
    _l.check(L.ucfvit_attention_cross_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), B, Nq, Nk, H, dh, q.stride(0),
                                          k.stride(0), scale, dt(q), _stream()), "ucfvit_attention_cross_fwd")
    return out, lse


def attention_cross_bwd(q, k, v, out, dout, lse, dq, dk, dv, B, Nq, Nk, H, dh, scale, accumulate):
    """this (query block, key block) pair's gradient terms, (+)= into fp32 dq [B*Nq, H*dh], dk / dv [B*Nk, H*dh]; lse / out: of the FULL softmax"""
    L = _l.load()
    _chk(dout, "attention_cross_bwd.dout"), _chk(out, "attention_cross_bwd.out")
    for t in (dq, dk, dv):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("attention_cross_bwd: dq / dk / dv must be contiguous fp32")
    delta = workspace(B * H * Nq * 4, q.device)
    _l.check(L.ucfvit_attention_cross_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dq.data_ptr(),
                                          dk.data_ptr(), dv.data_ptr(), delta.data_ptr(), B, Nq, Nk, H, dh, q.stride(0), k.stride(0), scale,
                                          1 if accumulate else 0, dt(q), _stream()), "ucfvit_attention_cross_bwd")


def attention_merge(o_acc, lse_acc, o_part, lse_part, B, Nq, H, dh, first):
    L = _l.load()
    _l.check(L.ucfvit_attention_merge(o_acc.data_ptr(), lse_acc.data_ptr(), o_part.data_ptr(), lse_part.data_ptr(), B, Nq, H, dh, 1 if first else 0,
                                      dt(o_part), _stream()), "ucfvit_attention_merge")


# ------------------------------------------------------------------------------------------------ front end
def im2col(img, p, out_dtype):
    """img fp32 [B,C,H,W] or [B,C,H,W,Z] -> [B*L, C*p^nd] in out_dtype, K-order (c, ph, pw[, pd])"""
    L = _l.load()
    _chk(img, "im2col.img")
    if img.dtype != torch.float32:
        raise TypeError("im2col: image must be float32 (as the reference dataloaders deliver it)")
    B, C = img.shape[0], img.shape[1]
    sp = list(img.shape[2:])
    nd = len(sp)
    Lp = 1
    for s in sp:
        Lp *= s // p
    cols = torch.empty((B * Lp, C * p ** nd), dtype=out_dtype, device=img.device)
    dims = (ctypes.c_int64 * 3)(*(sp + [1] * (3 - nd)))
    _l.check(L.ucfvit_im2col(img.data_ptr(), cols.data_ptr(), B, C, dims, nd, p, _DT[out_dtype], _stream()), "ucfvit_im2col")
    return cols


def tokens_fwd(patches, cls, pos, B, Lp, D):
    L = _l.load()
    _chk(patches, "tokens.patches")
    pre = 0 if cls is None else 1
    out = torch.empty((B, Lp + pre, D), dtype=patches.dtype, device=patches.device)
    _l.check(L.ucfvit_tokens_fwd(patches.data_ptr(), _p(cls), _p(pos), out.data_ptr(), B, Lp, D, pre, dt(patches), _stream()),
             "ucfvit_tokens_fwd")
    return out


def tokens_bwd(dout, B, Lp, D, has_cls, want_pos, dpos=None, dcls=None, accumulate=False, want_patches=True):
    L = _l.load()
    _chk(dout, "tokens_bwd.dout")
    dev = dout.device
    dpatches = torch.empty((B * Lp, D), dtype=dout.dtype, device=dev) if want_patches else None
    pre = 1 if has_cls else 0
    if want_pos and dpos is None:
        dpos = torch.empty((Lp + pre, D), dtype=torch.float32, device=dev)
    if has_cls and dcls is None:
        dcls = torch.empty(D, dtype=torch.float32, device=dev)
    _l.check(L.ucfvit_tokens_bwd(dout.data_ptr(), _p(dpatches), _p(dpos) if want_pos else None, _p(dcls) if has_cls else None, B, Lp, D,
                                 pre, 1 if accumulate else 0, dt(dout), _stream()), "ucfvit_tokens_bwd")
    return dpatches, dpos, dcls


# ------------------------------------------------------------------------------------------------ adaptive patching
def seq_patches(x, dtype):
    """x fp32 [B, C, S, P] -> rows [B*S, P*C] in `dtype` (einops 'b c s p -> b s (p c)', arch.py:466)"""
    L = _l.load()
    _chk(x, "seq_patches.x")
    if x.dtype != torch.float32 or x.dim() != 4:
        raise TypeError("seq_patches: x must be fp32 [B, C, S, P]")
    B, C, S, P = x.shape
    out = torch.empty((B * S, P * C), dtype=dtype, device=x.device)
    _l.check(L.ucfvit_seq_patches(x.data_ptr(), out.data_ptr(), B, C, S, P, _DT[dtype], _stream()), "ucfvit_seq_patches")
    return out


def _chk_seq_ps(seq_ps, B, S, name):
    _chk(seq_ps, name)
    if seq_ps.dtype != torch.float32 or seq_ps.dim() != 3 or seq_ps.shape[0] != B or seq_ps.shape[1] != S or seq_ps.shape[2] not in (3, 4):
        raise ValueError(f"{name}: seq_ps must be fp32 [B={B}, S={S}, 3|4], got {tuple(seq_ps.shape)} {seq_ps.dtype}")
    return seq_ps.shape[2]


def adaptive_pos_fwd(x2, seq_ps, w, bias, cls, B, S, D):
    """out [B, S+pre, D] = cat(cls, x) + cat(0, GELU(seq_ps w^T + bias))"""
    L = _l.load()
    _chk(x2, "adaptive_pos.x"), _chk(w, "adaptive_pos.w"), _chk(bias, "adaptive_pos.bias")
    kin = _chk_seq_ps(seq_ps, B, S, "adaptive_pos.seq_ps")
    if tuple(x2.shape) != (B * S, D) or tuple(w.shape) != (D, kin) or bias.numel() != D or w.dtype != x2.dtype or bias.dtype != x2.dtype:
        raise ValueError("adaptive_pos_fwd: shape / dtype mismatch")
    pre = 0 if cls is None else 1
    if cls is not None and (cls.numel() != D or cls.dtype != x2.dtype):
        raise ValueError("adaptive_pos_fwd: cls must hold D values of the compute dtype")
    out = torch.empty((B, S + pre, D), dtype=x2.dtype, device=x2.device)
    _l.check(L.ucfvit_adaptive_pos_fwd(x2.data_ptr(), seq_ps.data_ptr(), w.data_ptr(), bias.data_ptr(), _p(cls), out.data_ptr(), B, S, D, kin,
                                       pre, dt(x2), _stream()), "ucfvit_adaptive_pos_fwd")
    return out


def adaptive_pos_bwd(dout, seq_ps, w, bias, B, S, D, has_cls, want_dx=True, dw=None, dbias=None, dcls=None, acc_bits=0):
    """returns (dx [B*S, D] or None, dw fp32 [D, kin], dbias fp32 [D], dcls fp32 [D] or None); given outputs are written in place,
    acc_bits (1 dw, 2 dbias, 4 dcls) selects += for them"""
    L = _l.load()
    _chk(dout, "adaptive_pos_bwd.dout")
    kin = _chk_seq_ps(seq_ps, B, S, "adaptive_pos_bwd.seq_ps")
    pre = 1 if has_cls else 0
    if tuple(dout.shape) != (B, S + pre, D) or tuple(w.shape) != (D, kin) or w.dtype != dout.dtype:
        raise ValueError("adaptive_pos_bwd: shape / dtype mismatch")
    dev = dout.device
    dx = torch.empty((B * S, D), dtype=dout.dtype, device=dev) if want_dx else None
    if dw is None:
        dw = torch.empty((D, kin), dtype=torch.float32, device=dev)
    if dbias is None:
        dbias = torch.empty(D, dtype=torch.float32, device=dev)
    if has_cls and dcls is None:
        dcls = torch.empty(D, dtype=torch.float32, device=dev)
    for t, n in ((dw, D * kin), (dbias, D), (dcls, D)):
        if t is not None and (t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous()):
            raise ValueError("adaptive_pos_bwd: gradient outputs must be contiguous fp32 of the parameter's size")
    ws = workspace(L.ucfvit_adaptive_pos_bwd_workspace(B, S, D, kin, pre, dt(dout)), dev)
    _l.check(L.ucfvit_adaptive_pos_bwd(dout.data_ptr(), seq_ps.data_ptr(), w.data_ptr(), bias.data_ptr(), _p(dx), dw.data_ptr(), dbias.data_ptr(),
                                       _p(dcls) if has_cls else None, B, S, D, kin, pre, acc_bits, ws.data_ptr(), dt(dout), _stream()),
             "ucfvit_adaptive_pos_bwd")
    return dx, dw, dbias, (dcls if has_cls else None)


# ------------------------------------------------------------------------------------------------ variable aggregation
def varagg_fwd(kv, q, V, R, D, head_dim, scale):
    """kv [V*R, 2D] (rows ordered (v, r)), q fp32 [D] -> (out [R, D], lse fp32 [R, H])"""
    L = _l.load()
    _chk(kv, "varagg.kv"), _chk(q, "varagg.q")
    if tuple(kv.shape) != (V * R, 2 * D) or q.dtype != torch.float32 or q.numel() != D:
        raise ValueError("varagg_fwd: kv must be [V*R, 2D] and q fp32 [D]")
    out = torch.empty((R, D), dtype=kv.dtype, device=kv.device)
    lse = torch.empty((R, D // head_dim), dtype=torch.float32, device=kv.device)
    _l.check(L.ucfvit_varagg_fwd(kv.data_ptr(), q.data_ptr(), out.data_ptr(), lse.data_ptr(), R, V, D, head_dim, scale, dt(kv), _stream()),
             "ucfvit_varagg_fwd")
    return out, lse


def varagg_bwd(kv, q, out, lse, dout, V, R, D, head_dim, scale):
    """-> (dkv [V*R, 2D], dq fp32 [D])"""
    L = _l.load()
    _chk(dout, "varagg_bwd.dout")
    dkv = torch.empty_like(kv)
    dq_rows = torch.empty((R, D), dtype=torch.float32, device=kv.device)
    _l.check(L.ucfvit_varagg_bwd(kv.data_ptr(), q.data_ptr(), out.data_ptr(), lse.data_ptr(), dout.data_ptr(), dkv.data_ptr(), dq_rows.data_ptr(),
                                 R, V, D, head_dim, scale, dt(kv), _stream()), "ucfvit_varagg_bwd")
    return dkv, colsum(dq_rows)


# ------------------------------------------------------------------------------------------------ quadtree patcher
def quadtree_build(edges, fixed_length):
    """edges uint8 [B, H, W] (0 / 255) -> (nodes int32 [B, L, 4] = (x1, x2, y1, y2), values int32 [B, L], count int32 [B],
    seq_ps fp32 [B, L, 3] = (size, centre x, centre y))"""
    L = _l.load()
    _chk(edges, "quadtree_build.edges")
    if edges.dtype != torch.uint8 or edges.dim() != 3:
        raise TypeError("quadtree_build: edges must be uint8 [B, H, W]")
    B, H, W = edges.shape
    dev = edges.device
    nodes = torch.empty((B, fixed_length, 4), dtype=torch.int32, device=dev)
    values = torch.empty((B, fixed_length), dtype=torch.int32, device=dev)
    count = torch.empty(B, dtype=torch.int32, device=dev)
    seq_ps = torch.empty((B, fixed_length, 3), dtype=torch.float32, device=dev)
    ws = workspace(L.ucfvit_quadtree_workspace(B, H, W), dev)
    _l.check(L.ucfvit_quadtree_build(edges.data_ptr(), nodes.data_ptr(), values.data_ptr(), count.data_ptr(), seq_ps.data_ptr(), B, H, W,
                                     fixed_length, ws.data_ptr(), _stream()), "ucfvit_quadtree_build")
    return nodes, values, count, seq_ps


def quadtree_serialize(img, nodes, count, patch):
    """img fp32 [B, H, W, C], nodes / count from quadtree_build -> x [B, C, L, patch*patch] (the reference's plain reshape of the
    [L, p, p, C] patch list, transform.py:42-48) = the model input of adaptive_patching=True"""
    L = _l.load()
    _chk(img, "quadtree_serialize.img"), _chk(nodes, "quadtree_serialize.nodes"), _chk(count, "quadtree_serialize.count")
    if img.dtype != torch.float32 or img.dim() != 4 or nodes.dtype != torch.int32 or count.dtype != torch.int32:
        raise TypeError("quadtree_serialize: img fp32 [B, H, W, C], nodes int32 [B, L, 4], count int32 [B]")
    B, H, W, C = img.shape
    S = nodes.shape[1]
    seq = torch.empty((B, S, patch, patch, C), dtype=torch.float32, device=img.device)
    _l.check(L.ucfvit_quadtree_serialize(img.data_ptr(), nodes.data_ptr(), count.data_ptr(), seq.data_ptr(), B, H, W, C, S, patch, _stream()),
             "ucfvit_quadtree_serialize")
    return seq.view(B, C, S, patch * patch)


def octree_build(domain, fixed_length, norm_factor=255):
    """domain uint8 [B, N, N, N] ([z][y][x]) -> (nodes int32 [B, L, 6], values int32 [B, L], count int32 [B], seq_ps fp32 [B, L, 4])"""
    L = _l.load()
    _chk(domain, "octree_build.domain")
    if domain.dtype != torch.uint8 or domain.dim() != 4 or not (domain.shape[1] == domain.shape[2] == domain.shape[3]):
        raise TypeError("octree_build: domain must be a batch of cubic uint8 volumes [B, N, N, N]")
    B, N = domain.shape[0], domain.shape[1]
    dev = domain.device
    nodes = torch.empty((B, fixed_length, 6), dtype=torch.int32, device=dev)
    values = torch.empty((B, fixed_length), dtype=torch.int32, device=dev)
    count = torch.empty(B, dtype=torch.int32, device=dev)
    seq_ps = torch.empty((B, fixed_length, 4), dtype=torch.float32, device=dev)
    ws = workspace(L.ucfvit_octree_workspace(B, N), dev)
    _l.check(L.ucfvit_octree_build(domain.data_ptr(), nodes.data_ptr(), values.data_ptr(), count.data_ptr(), seq_ps.data_ptr(), B, N,
                                   fixed_length, int(norm_factor), ws.data_ptr(), _stream()), "ucfvit_octree_build")
    return nodes, values, count, seq_ps


def octree_serialize(img, nodes, count, patch, flat=True):
    """img fp32 [B, N, N, N, C] -> x [B, C, L, patch**3] (Patchify_3D's plain reshape, transform.py:123-126) or, flat=False, the patch list
    [B, L, p, p, p, C]"""
    L = _l.load()
    _chk(img, "octree_serialize.img"), _chk(nodes, "octree_serialize.nodes"), _chk(count, "octree_serialize.count")
    if img.dtype != torch.float32 or img.dim() != 5 or nodes.dtype != torch.int32 or count.dtype != torch.int32:
        raise TypeError("octree_serialize: img fp32 [B, N, N, N, C], nodes int32 [B, L, 6], count int32 [B]")
    B, N, _, _, C = img.shape
    S = nodes.shape[1]
    seq = torch.empty((B, S, patch, patch, patch, C), dtype=torch.float32, device=img.device)
    _l.check(L.ucfvit_octree_serialize(img.data_ptr(), nodes.data_ptr(), count.data_ptr(), seq.data_ptr(), B, N, C, S, patch, _stream()),
             "ucfvit_octree_serialize")
    return seq.view(B, C, S, patch ** 3) if flat else seq


def cross_entropy(logits, labels, grad_scale=1.0, want_grad=True):
    """returns (loss fp32 scalar tensor, dlogits or None, row_loss)"""
    L = _l.load()
    _chk(logits, "cross_entropy.logits"), _chk(labels, "cross_entropy.labels")
    if labels.dtype != torch.int64:
        raise TypeError("cross_entropy: labels must be int64")
    B, C = logits.shape
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    rows = torch.empty(B, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    _l.check(L.ucfvit_cross_entropy(logits.data_ptr(), labels.data_ptr(), loss.data_ptr(), rows.data_ptr(), _p(dl), B, C, grad_scale,
                                    dt(logits), _stream()), "ucfvit_cross_entropy")
    return loss, dl, rows


# ------------------------------------------------------------------------------------------------ UNETR decoder (HBM-bound part)
def _rows_view(x):
    """N C (D) H W tensor -> (rows = N*C, S = voxels per row); contiguous"""
    _chk(x, "instnorm.x")
    if x.dim() < 3:
        raise ValueError("instnorm: expected [N, C, *spatial]")
    rows = x.shape[0] * x.shape[1]
    S = x.numel() // max(rows, 1)
    V = 16 // x.element_size()
    if S % V:
        raise ValueError(f"instnorm: the spatial size {S} must be a multiple of {V} elements (16-byte rows); there is no slow path")
    return rows, S


def instnorm_fwd(x, res=None, eps=1e-5, slope=0.01):
    """y = leaky_relu(instance_norm(x) [+ res], slope) over every (n, c) row (slope 1.0: no activation) -> (y, mean, rstd)"""
    L = _l.load()
    rows, S = _rows_view(x)
    if res is not None:
        _chk(res, "instnorm.res")
        if res.shape != x.shape or res.dtype != x.dtype:
            raise ValueError("instnorm: res must have the shape and dtype of x")
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    ws = workspace(L.ucfvit_instnorm_workspace(rows, S), x.device)
    _l.check(L.ucfvit_instnorm_fwd(x.data_ptr(), _p(res), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, S, eps, slope, ws.data_ptr(), dt(x),
                                   _stream()), "ucfvit_instnorm_fwd")
    return y, mean, rstd


def instnorm_bwd(dy, y, x, mean, rstd, slope, want_dres):
    L = _l.load()
    rows, S = _rows_view(x)
    _chk(dy, "instnorm_bwd.dy")
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    ws = workspace(L.ucfvit_instnorm_workspace(rows, S), x.device)
    _l.check(L.ucfvit_instnorm_bwd(dy.data_ptr(), y.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _p(dres), rows, S,
                                   slope, ws.data_ptr(), dt(x), _stream()), "ucfvit_instnorm_bwd")
    return dx, dres


def dice_ce(logits, labels, smooth_nr=1e-5, smooth_dr=1e-5, grad_scale=1.0, want_grad=True):
    """logits [B, n, *spatial] (2 <= n <= 8), labels int64 [B, *spatial] (or [B, 1, *spatial]) -> (loss fp32 scalar, dlogits or None).
    logits may be contiguous (N C D H W) or a channels-last view of a [B, *spatial, ld >= n] buffer (what the HIP decoder returns)."""
    L = _l.load()
    _chk(labels, "dice_ce.labels")
    if not logits.is_cuda:
        _chk(logits, "dice_ce.logits")
    if labels.dtype != torch.int64:
        raise TypeError("dice_ce: labels must be int64")
    B, n = logits.shape[0], logits.shape[1]
    S = logits.numel() // (B * n)
    if labels.numel() != B * S:
        raise ValueError("dice_ce: labels must hold one class index per voxel")
    if logits.is_contiguous():
        sb, sc, ss = n * S, S, 1
        dl = torch.empty_like(logits) if want_grad else None
    else:
        cl = logits.movedim(1, -1)                           # [B, *spatial, n]
        ld = cl.stride(-2)
        ok = cl.stride(-1) == 1 and ld >= n and cl.stride(0) == S * ld
        exp = ld
        for d in range(cl.dim() - 2, 0, -1):
            ok = ok and cl.stride(d) == exp
            exp *= cl.shape[d]
        if not ok:
            raise RuntimeError("dice_ce: logits must be contiguous or a channels-last view with contiguous voxel rows")
        sb, sc, ss = S * ld, 1, ld
        mk = torch.empty if ld == n else torch.zeros                 # padding columns of the rows, if any, read as zero gradient
        dl = mk(B * S * ld, dtype=logits.dtype, device=logits.device).as_strided(logits.shape, logits.stride()) if want_grad else None
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    ws = workspace(L.ucfvit_dice_ce_workspace(B, S), logits.device)
    _l.check(L.ucfvit_dice_ce_strided(logits.data_ptr(), labels.data_ptr(), loss.data_ptr(), _p(dl), B, n, S, sb, sc, ss, smooth_nr, smooth_dr,
                                      grad_scale, ws.data_ptr(), dt(logits), _stream()), "ucfvit_dice_ce")
    return loss, dl


def _dice_strides(logits, B, n, S):
    """(stride_b, stride_c, stride_s) of contiguous N C (D) H W logits or of a channels-last view with contiguous voxel rows"""
    if logits.is_contiguous():
        return n * S, S, 1
    cl = logits.movedim(1, -1)                           # [B, *spatial, n]
    ld = cl.stride(-2)
    ok = cl.stride(-1) == 1 and ld >= n and cl.stride(0) == S * ld
    exp = ld
    for d in range(cl.dim() - 2, 0, -1):
        ok = ok and cl.stride(d) == exp
        exp *= cl.shape[d]
    if not ok:
        raise RuntimeError("dice_ce: logits must be contiguous or a channels-last view with contiguous voxel rows")
    return S * ld, 1, ld


def dice_ce_stats(logits, labels):
    """this rank's per-(batch, class) sums of the Dice + CE loss over its slab of a sharded volume: fp32 [B, ucfvit_dice_ce_stats_floats()]
    (to be summed over the group and handed to dice_ce_from_stats)"""
    L = _l.load()
    _chk(labels, "dice_ce_stats.labels")
    if not logits.is_cuda:
        _chk(logits, "dice_ce_stats.logits")
    if labels.dtype != torch.int64:
        raise TypeError("dice_ce_stats: labels must be int64")
    B, n = logits.shape[0], logits.shape[1]
    S = logits.numel() // (B * n)
    if labels.numel() != B * S:
        raise ValueError("dice_ce_stats: labels must hold one class index per local voxel")
    sb, sc, ss = _dice_strides(logits, B, n, S)
    stats = torch.empty((B, L.ucfvit_dice_ce_stats_floats()), dtype=torch.float32, device=logits.device)
    ws = workspace(L.ucfvit_dice_ce_workspace(B, S), logits.device)
    _l.check(L.ucfvit_dice_ce_stats(logits.data_ptr(), labels.data_ptr(), stats.data_ptr(), B, n, S, sb, sc, ss, ws.data_ptr(), dt(logits), _stream()),
             "ucfvit_dice_ce_stats")
    return stats


def dice_ce_from_stats(logits, labels, stats, S_total, smooth_nr=1e-5, smooth_dr=1e-5, grad_scale=1.0, want_grad=True):
    """stats: the per-(batch, class) sums of the WHOLE volume (dice_ce_stats summed over the group) -> (loss of the whole volume, gradient of the
    LOCAL logits or None).  S_total: voxels of the whole volume per batch element."""
    L = _l.load()
    B, n = logits.shape[0], logits.shape[1]
    S = logits.numel() // (B * n)
    sb, sc, ss = _dice_strides(logits, B, n, S)
    if want_grad:
        if logits.is_contiguous():
            dl = torch.empty_like(logits)
        else:
            mk = torch.empty if ss == n else torch.zeros             # padding columns of the rows, if any, read as zero gradient
            dl = mk(B * sb, dtype=logits.dtype, device=logits.device).as_strided(logits.shape, logits.stride())
    else:
        dl = None
    stats = stats.clone()                                # rewritten in place by the fold
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    _l.check(L.ucfvit_dice_ce_from_stats(logits.data_ptr(), labels.data_ptr(), stats.data_ptr(), loss.data_ptr(), _p(dl), B, n, S, int(S_total), sb, sc, ss,
                                         smooth_nr, smooth_dr, grad_scale, dt(logits), _stream()), "ucfvit_dice_ce_from_stats")
    return loss, dl


# ------------------------------------------------------------------------------------------------ UNETR decoder, channels-last bf16
def _chk_cl(t, name, C=None):
    _chk(t, name)
    if t.dtype != torch.bfloat16 or t.dim() != 5:
        raise TypeError(f"{name}: expected a channels-last bf16 tensor [B, X, Y, Z, C], got {tuple(t.shape)} {t.dtype}")
    if C is not None and t.shape[-1] != C:
        raise ValueError(f"{name}: expected {C} channels, got {t.shape[-1]}")
    return t


def conv_packed_numel(cin, cout, ksize):
    cpc = min(cin, 32)
    tps = 32 // cpc
    return (cin // cpc) * (-(-(ksize ** 3) // tps)) * cout * 32


def conv3d_fwd(x, w_packed, cout, ksize=3, bias=None, cout_store=None, out_dtype=torch.bfloat16, accumulate_into=None, stats_eps=None):
    """x [B, X, Y, Z, Cin] bf16, w_packed from conv.pack_conv_weight (cout rows, a multiple of 16) -> y [B, X, Y, Z, cout_store] (ksize 3:
    stride 1, zero padding 1; ksize 1: pointwise).  bias: fp32 [cout] or None."""
    L = _l.load()
    _chk_cl(x, "conv3d.x"), _chk(w_packed, "conv3d.w_packed")
    B, X, Y, Z, cin = x.shape
    if w_packed.dtype != torch.bfloat16 or w_packed.numel() != conv_packed_numel(cin, cout, ksize):
        raise ValueError("conv3d: w_packed does not match (Cin, Cout, ksize)")
    if bias is not None:
        _chk(bias, "conv3d.bias")
        if bias.dtype != torch.float32 or bias.numel() != cout:
            raise ValueError("conv3d: bias must be fp32 [Cout]")
    cs = cout if cout_store is None else cout_store
    if accumulate_into is not None:                          # y += result: the second data gradient of an input two layers consume
        y = _chk(accumulate_into, "conv3d.accumulate_into")
        if tuple(y.shape) != (B, X, Y, Z, cs) or y.dtype != out_dtype:
            raise ValueError("conv3d: accumulate_into must have the output's shape and dtype")
    else:
        y = torch.empty((B, X, Y, Z, cs), dtype=out_dtype, device=x.device)
    part, rows = None, 0
    if stats_eps is not None:
        # instance-norm statistics of the output as a by-product of the epilogue where the serving kernel has one, else one pass over y
        if accumulate_into is not None or cs != cout or out_dtype != torch.bfloat16:
            raise ValueError("conv3d: statistics need a dense bf16 output without accumulation")
        rows = L.ucfvit_conv3d_fwd_stats_rows(B, X, Y, Z, cin, cout, ksize, 0 if bias is None else 1)
        if rows:
            buf = workspace((B * rows * 3 * cout + B * 256 * 3 * cout) * 4, x.device)       # partial rows (count, mean, M2), then the first fold stage's scratch
            part, fold_ws = buf, buf[B * rows * 3 * cout:]
    _l.check(L.ucfvit_conv3d_fwd(x.data_ptr(), w_packed.data_ptr(), _p(bias), y.data_ptr(), B, X, Y, Z, cin, cout, ksize, cs, cs, dt(y),
                                 0 if accumulate_into is None else 1, _p(part), _stream()), "ucfvit_conv3d_fwd")
    if stats_eps is None:
        return y
    if not rows:
        return (y,) + instnorm_cl_stats(y, stats_eps)
    mean = torch.empty((B, cout), dtype=torch.float32, device=x.device)
    rstd = torch.empty((B, cout), dtype=torch.float32, device=x.device)
    _l.check(L.ucfvit_instnorm_cl_stats_fold(part.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, X * Y * Z, cout, rows, stats_eps, fold_ws.data_ptr(),
                                             _stream()), "ucfvit_instnorm_cl_stats_fold")
    return y, mean, rstd


def conv3d_wgrad(x, dy, ksize=3):
    """-> packed fp32 weight gradient (conv.unpack_conv_wgrad turns it into [Cout, Cin, k, k, k])"""
    L = _l.load()
    _chk_cl(x, "conv3d_wgrad.x"), _chk_cl(dy, "conv3d_wgrad.dy")
    B, X, Y, Z, cin = x.shape
    cout = dy.shape[-1]
    if dy.shape[:4] != x.shape[:4]:
        raise ValueError("conv3d_wgrad: x and dy must cover the same voxels")
    n = L.ucfvit_conv3d_wgrad_size(cin, cout, ksize)
    nbytes = L.ucfvit_conv3d_wgrad_workspace(B, X, Y, Z, cin, cout, ksize)
    if n <= 0 or nbytes <= 0:
        raise ValueError(f"conv3d_wgrad: unsupported channel counts Cin={cin} Cout={cout}")
    dw = torch.empty(n, dtype=torch.float32, device=x.device)
    ws = workspace(nbytes, x.device)
    _l.check(L.ucfvit_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, X, Y, Z, cin, cout, ksize, _stream()),
             "ucfvit_conv3d_wgrad")
    return dw


def cl_row_stride(t):
    """voxel-row stride (elements) of a channels-last bf16 tensor [B, X, Y, Z, C] that is dense or a channel slice of a dense
    [B, X, Y, Z, ld] buffer (what torch.cat's backward and a pre-allocated concatenation hand around); None if it is neither"""
    if t.dim() != 5 or t.dtype != torch.bfloat16 or not t.is_cuda or t.stride(-1) != 1:
        return None
    ld = t.stride(-2)
    if ld < t.shape[-1] or ld % 8 or t.data_ptr() % 16:
        return None
    exp = ld
    for d in (3, 2, 1):
        if t.stride(d) != exp and t.shape[d] != 1:
            return None
        exp *= t.shape[d]
    if t.stride(0) != exp and t.shape[0] != 1:
        return None
    return ld


def depth_to_space2(cols, B, Xi, Yi, Zi, C, out=None, skip=None):
    """cols [B Xi Yi Zi, 8 C] (column blocks (dx, dy, dz)) -> [B, 2Xi, 2Yi, 2Zi, C]; `out` may be a channel slice of a wider channels-last
    buffer; with `skip` (dense [B, 2Xi, 2Yi, 2Zi, Cs]) the buffer behind `out` must be C + Cs wide and receives the skip behind the C channels"""
    L = _l.load()
    _chk(cols, "depth_to_space2.cols")
    if cols.dtype != torch.bfloat16 or cols.numel() != B * Xi * Yi * Zi * 8 * C:
        raise ValueError("depth_to_space2: bad operand")
    if out is None:
        out = torch.empty((B, 2 * Xi, 2 * Yi, 2 * Zi, C), dtype=torch.bfloat16, device=cols.device)
    ld = cl_row_stride(out)
    if ld is None or tuple(out.shape) != (B, 2 * Xi, 2 * Yi, 2 * Zi, C):
        raise ValueError("depth_to_space2: out must be [B, 2Xi, 2Yi, 2Zi, C], dense or a channel slice of a dense channels-last buffer")
    cs = 0
    if skip is not None:
        _chk_cl(skip, "depth_to_space2.skip")
        cs = skip.shape[-1]
        if tuple(skip.shape[:4]) != tuple(out.shape[:4]) or ld < C + cs:
            raise ValueError("depth_to_space2: skip must cover the output's voxels and fit behind it in the buffer")
    _l.check(L.ucfvit_depth_to_space2(cols.data_ptr(), out.data_ptr(), B, Xi, Yi, Zi, C, ld, 1, _p(skip), cs, _stream()), "ucfvit_depth_to_space2")
    return out


def space_to_depth2(y):
    """[B, 2Xi, 2Yi, 2Zi, C] (dense or a channel slice of a dense channels-last buffer) -> [B Xi Yi Zi, 8 C]"""
    L = _l.load()
    ld = cl_row_stride(y)
    if ld is None:
        y = _chk_cl(y.contiguous(), "space_to_depth2.y")
        ld = y.shape[-1]
    B, X2, Y2, Z2, C = y.shape
    if X2 % 2 or Y2 % 2 or Z2 % 2:
        raise ValueError("space_to_depth2: extents must be even")
    Xi, Yi, Zi = X2 // 2, Y2 // 2, Z2 // 2
    cols = torch.empty((B * Xi * Yi * Zi, 8 * C), dtype=torch.bfloat16, device=y.device)
    _l.check(L.ucfvit_depth_to_space2(y.data_ptr(), cols.data_ptr(), B, Xi, Yi, Zi, C, ld, 0, None, 0, _stream()), "ucfvit_depth_to_space2")
    return cols


def pad_channels8(vol):
    """fp32 [B, C, X, Y, Z] with C <= 8 -> channels-last bf16 [B, X, Y, Z, 8] (channels C..7 zero)"""
    L = _l.load()
    _chk(vol, "pad_channels8.vol")
    if vol.dtype != torch.float32 or vol.dim() != 5 or vol.shape[1] > 8:
        raise TypeError("pad_channels8: expected an fp32 [B, C <= 8, X, Y, Z] volume")
    B, C = vol.shape[0], vol.shape[1]
    out = torch.empty((B,) + tuple(vol.shape[2:]) + (8,), dtype=torch.bfloat16, device=vol.device)
    _l.check(L.ucfvit_pad_channels8(vol.data_ptr(), out.data_ptr(), B, C, vol.numel() // (B * C), _stream()), "ucfvit_pad_channels8")
    return out


def pad_rows8(t):
    """[..., C <= 8] fp32 or bf16 with contiguous voxel rows (element stride 1 in the last axis, one row stride ld >= C for all the others:
    a dense tensor or a channel slice of a wider channels-last buffer) -> dense bf16 [..., 8], columns C..7 zero"""
    L = _l.load()
    if not t.is_cuda or t.dtype not in (torch.float32, torch.bfloat16) or t.shape[-1] > 8 or t.stride(-1) != 1:
        raise TypeError("pad_rows8: expected a CUDA fp32 / bf16 tensor with at most 8 contiguous channels in the last axis")
    ld = t.stride(-2) if t.dim() > 1 else t.shape[-1]
    exp = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            raise ValueError("pad_rows8: the leading axes must form one run of rows with a common stride")
        exp *= t.shape[d]
    V = t.numel() // t.shape[-1]
    out = torch.empty(tuple(t.shape[:-1]) + (8,), dtype=torch.bfloat16, device=t.device)
    _l.check(L.ucfvit_pad_rows8(t.data_ptr(), dt(t), out.data_ptr(), V, t.shape[-1], ld, _stream()), "ucfvit_pad_rows8")
    return out


def instnorm_cl_fwd(x, res=None, eps=1e-5, slope=0.01):
    """channels-last: y = leaky_relu(instance_norm(x) [+ res], slope) per (batch, channel) -> (y, mean [B, C], rstd [B, C])"""
    L = _l.load()
    _chk_cl(x, "instnorm_cl.x")
    if res is not None:
        _chk_cl(res, "instnorm_cl.res")
        if res.shape != x.shape:
            raise ValueError("instnorm_cl: res must have the shape of x")
    B, C = x.shape[0], x.shape[-1]
    S = x.numel() // (B * C)
    y = torch.empty_like(x)
    mean = torch.empty((B, C), dtype=torch.float32, device=x.device)
    rstd = torch.empty((B, C), dtype=torch.float32, device=x.device)
    ws = workspace(L.ucfvit_instnorm_cl_workspace(B, S, C), x.device)
    _l.check(L.ucfvit_instnorm_cl_fwd(x.data_ptr(), _p(res), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, S, C, eps, slope, ws.data_ptr(),
                                      _stream()), "ucfvit_instnorm_cl_fwd")
    return y, mean, rstd


def instnorm_cl_stats(x, eps=1e-5):
    """-> (mean [B, C], rstd [B, C]) of a channels-last bf16 map"""
    L = _l.load()
    _chk_cl(x, "instnorm_cl_stats.x")
    B, C = x.shape[0], x.shape[-1]
    S = x.numel() // (B * C)
    mean = torch.empty((B, C), dtype=torch.float32, device=x.device)
    rstd = torch.empty((B, C), dtype=torch.float32, device=x.device)
    ws = workspace(L.ucfvit_instnorm_cl_workspace(B, S, C), x.device)
    _l.check(L.ucfvit_instnorm_cl_stats(x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, S, C, eps, ws.data_ptr(), _stream()), "ucfvit_instnorm_cl_stats")
    return mean, rstd


def instnorm_cl_apply(x, mean, rstd, res=None, slope=0.01):
    """y = lrelu((x - mean) rstd [+ res], slope) with the statistics given"""
    L = _l.load()
    _chk_cl(x, "instnorm_cl_apply.x")
    if res is not None:
        _chk_cl(res, "instnorm_cl_apply.res")
        if res.shape != x.shape:
            raise ValueError("instnorm_cl_apply: res must have the shape of x")
    B, C = x.shape[0], x.shape[-1]
    S = x.numel() // (B * C)
    y = torch.empty_like(x)
    _l.check(L.ucfvit_instnorm_cl_apply(x.data_ptr(), _p(res), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, S, C, slope, _stream()),
             "ucfvit_instnorm_cl_apply")
    return y


def instnorm_cl_apply2(x, mean, rstd, x2, mean2, rstd2, slope):
    """y = lrelu(norm(x) + norm(x2)) with both statistics given"""
    L = _l.load()
    _chk_cl(x, "instnorm_cl_apply2.x"), _chk_cl(x2, "instnorm_cl_apply2.x2")
    if x2.shape != x.shape:
        raise ValueError("instnorm_cl_apply2: the two branches must have one shape")
    B, C = x.shape[0], x.shape[-1]
    S = x.numel() // (B * C)
    y = torch.empty_like(x)
    _l.check(L.ucfvit_instnorm_cl_apply2(x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), x2.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(), y.data_ptr(),
                                         B, S, C, slope, _stream()), "ucfvit_instnorm_cl_apply2")
    return y


def instnorm_cl_bwd2(dy, y, x, mean, rstd, x2, mean2, rstd2, slope):
    """backward of instnorm_cl_apply2 -> (dx, dx2); dy may be a channel slice of a wider channels-last gradient"""
    L = _l.load()
    _chk_cl(x, "instnorm_cl_bwd2.x")
    ld = cl_row_stride(dy)
    if ld is None or dy.shape != x.shape:
        dy = _chk_cl(dy.contiguous(), "instnorm_cl_bwd2.dy")
        ld = dy.shape[-1]
    B, C = x.shape[0], x.shape[-1]
    S = x.numel() // (B * C)
    dx, dx2 = torch.empty_like(x), torch.empty_like(x)
    ws = workspace(L.ucfvit_instnorm_cl_bwd2_workspace(B, S, C), x.device)
    _l.check(L.ucfvit_instnorm_cl_bwd2(dy.data_ptr(), y.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), x2.data_ptr(), mean2.data_ptr(),
                                       rstd2.data_ptr(), dx.data_ptr(), dx2.data_ptr(), B, S, C, ld, slope, ws.data_ptr(), _stream()),
             "ucfvit_instnorm_cl_bwd2")
    return dx, dx2


def instnorm_cl_bwd(dy, y, x, mean, rstd, slope, want_dres, had_res=None):
    """dy may be a channel slice of a wider channels-last gradient (the skip half of a concatenation's gradient): read in place"""
    L = _l.load()
    _chk_cl(x, "instnorm_cl_bwd.x")
    ld = cl_row_stride(dy)
    if ld is None or dy.shape != x.shape:
        dy = _chk_cl(dy.contiguous(), "instnorm_cl_bwd.dy")
        ld = dy.shape[-1]
    B, C = x.shape[0], x.shape[-1]
    S = x.numel() // (B * C)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    ws = workspace(L.ucfvit_instnorm_cl_workspace(B, S, C), x.device)
    had_res = want_dres if had_res is None else had_res
    _l.check(L.ucfvit_instnorm_cl_bwd(dy.data_ptr(), y.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _p(dres), B, S, C,
                                      ld, slope, 1 if had_res else 0, ws.data_ptr(), _stream()), "ucfvit_instnorm_cl_bwd")
    return dx, dres


def instnorm_cl_bwd_sums(dy, y, x, mean, rstd, slope, had_res):
    """first half of instnorm_cl_bwd for a sharded volume: (m1, m2) [B, C] = the means over the LOCAL voxels of dy' and dy' xhat"""
    L = _l.load()
    _chk_cl(x, "instnorm_cl_bwd_sums.x")
    ld = cl_row_stride(dy)
    if ld is None or dy.shape != x.shape:
        dy = _chk_cl(dy.contiguous(), "instnorm_cl_bwd_sums.dy")
        ld = dy.shape[-1]
    B, C = x.shape[0], x.shape[-1]
    S = x.numel() // (B * C)
    m1 = torch.empty((B, C), dtype=torch.float32, device=x.device)
    m2 = torch.empty((B, C), dtype=torch.float32, device=x.device)
    ws = workspace(L.ucfvit_instnorm_cl_workspace(B, S, C), x.device)
    _l.check(L.ucfvit_instnorm_cl_bwd_sums(dy.data_ptr(), y.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), m1.data_ptr(), m2.data_ptr(),
                                           B, S, C, ld, slope, 1 if had_res else 0, ws.data_ptr(), _stream()), "ucfvit_instnorm_cl_bwd_sums")
    return m1, m2, dy


def instnorm_cl_bwd_apply(dy, y, x, mean, rstd, m1, m2, slope, want_dres, had_res):
    L = _l.load()
    ld = cl_row_stride(dy)
    if ld is None or dy.shape != x.shape:
        dy = _chk_cl(dy.contiguous(), "instnorm_cl_bwd_apply.dy")
        ld = dy.shape[-1]
    B, C = x.shape[0], x.shape[-1]
    S = x.numel() // (B * C)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    _l.check(L.ucfvit_instnorm_cl_bwd_apply(dy.data_ptr(), y.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), m1.data_ptr(), m2.data_ptr(),
                                            dx.data_ptr(), _p(dres), B, S, C, ld, slope, 1 if had_res else 0, _stream()), "ucfvit_instnorm_cl_bwd_apply")
    return dx, dres


# ------------------------------------------------------------------------------------------------ MAE
def mae_mask(noise, len_keep):
    L = _l.load()
    _chk(noise, "mae_mask.noise")
    if noise.dtype != torch.float32:
        raise TypeError("mae_mask: noise must be float32")
    B, Lp = noise.shape
    ids_shuffle = torch.empty((B, Lp), dtype=torch.int64, device=noise.device)
    ids_restore = torch.empty((B, Lp), dtype=torch.int64, device=noise.device)
    mask = torch.empty((B, Lp), dtype=torch.float32, device=noise.device)
    _l.check(L.ucfvit_mae_mask(noise.data_ptr(), ids_shuffle.data_ptr(), ids_restore.data_ptr(), mask.data_ptr(), B, Lp, len_keep,
                               _stream()), "ucfvit_mae_mask")
    return ids_shuffle, ids_restore, mask


def gather_rows(src, idx, R, idx_stride):
    """src [B,L,D]; idx int64 with row stride idx_stride; -> [B,R,D]"""
    L = _l.load()
    _chk(src, "gather_rows.src")
    B, Lp, D = src.shape
    out = torch.empty((B, R, D), dtype=src.dtype, device=src.device)
    _l.check(L.ucfvit_gather_rows(src.data_ptr(), idx.data_ptr(), out.data_ptr(), B, Lp, R, D, idx_stride, dt(src), _stream()),
             "ucfvit_gather_rows")
    return out


def scatter_rows(dout, idx, Lp, idx_stride):
    L = _l.load()
    _chk(dout, "scatter_rows.dout")
    B, R, D = dout.shape
    dsrc = torch.empty((B, Lp, D), dtype=dout.dtype, device=dout.device)
    _l.check(L.ucfvit_scatter_rows(dout.data_ptr(), idx.data_ptr(), dsrc.data_ptr(), B, Lp, R, D, idx_stride, dt(dout), _stream()),
             "ucfvit_scatter_rows")
    return dsrc


def unshuffle_fwd(x, mask_token, ids_restore, pos):
    """x [B,R,D], mask_token [D], ids_restore [B,L] int64, pos [L,D] or None -> [B,L,D]"""
    L = _l.load()
    _chk(x, "unshuffle.x")
    B, R, D = x.shape
    Lp = ids_restore.shape[1]
    out = torch.empty((B, Lp, D), dtype=x.dtype, device=x.device)
    _l.check(L.ucfvit_unshuffle_fwd(x.data_ptr(), mask_token.data_ptr(), ids_restore.data_ptr(), _p(pos), out.data_ptr(), B, Lp, R, D,
                                    dt(x), _stream()), "ucfvit_unshuffle_fwd")
    return out


def unshuffle_bwd(dout, ids_restore, R, want_pos, dmask=None, dpos=None, accumulate=False):
    L = _l.load()
    _chk(dout, "unshuffle_bwd.dout")
    B, Lp, D = dout.shape
    dev = dout.device
    dx = torch.empty((B, R, D), dtype=dout.dtype, device=dev)
    if dmask is None:
        dmask = torch.empty(D, dtype=torch.float32, device=dev)
    if want_pos and dpos is None:
        dpos = torch.empty((Lp, D), dtype=torch.float32, device=dev)
    ws = workspace(L.ucfvit_unshuffle_bwd_workspace(B, D), dev)
    _l.check(L.ucfvit_unshuffle_bwd(dout.data_ptr(), ids_restore.data_ptr(), dx.data_ptr(), dmask.data_ptr(), _p(dpos) if want_pos else None,
                                    B, Lp, R, D, 1 if accumulate else 0, ws.data_ptr(), dt(dout), _stream()), "ucfvit_unshuffle_bwd")
    return dx, dmask, dpos


def patch_mse(pred, img, p, mask=None, grad_scale=1.0, want_grad=True):
    """pred [B,L,P]; img fp32 NCHW(D) (p = patch size) or the adaptive token sequence [B,C,S,P] (p = None); mask fp32 [B,L] or None
    -> (loss scalar fp32, dpred or None)"""
    L = _l.load()
    _chk(pred, "patch_mse.pred"), _chk(img, "patch_mse.img")
    B, C = img.shape[0], img.shape[1]
    sp = list(img.shape[2:])
    nd = len(sp)
    if p is None:       # adaptive token sequence x [B, C, S, P]: target 'b c s p -> b s (p c)'
        if nd != 2 or tuple(pred.shape) != (B, sp[0], C * sp[1]):
            raise ValueError(f"patch_mse: sequence target [B,C,S,P]={tuple(img.shape)} does not match pred {tuple(pred.shape)}")
        nd, p, sp = 1, sp[1], [sp[0]]
    dims = (ctypes.c_int64 * 3)(*(sp + [1] * (3 - nd)))
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    ws = workspace(4 * 2049, pred.device)
    _l.check(L.ucfvit_patch_mse(pred.data_ptr(), img.data_ptr(), _p(mask), loss.data_ptr(), _p(dpred), B, C, dims, nd, p, grad_scale,
                                ws.data_ptr(), dt(pred), _stream()), "ucfvit_patch_mse")
    return loss, dpred


# ------------------------------------------------------------------------------------------------ optimizer / casts
def adamw(p, g, m, v, shadow, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    L = _l.load()
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    _l.check(L.ucfvit_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _p(shadow), p.numel(), lr, beta1, beta2, eps,
                            weight_decay, bc1, bc2, grad_scale, dt(g), _stream()), "ucfvit_adamw")


def cast(src, dst, scale=1.0):
    L = _l.load()
    _l.check(L.ucfvit_cast(src.data_ptr(), dst.data_ptr(), src.numel(), dt(src), dt(dst), scale, _stream()), "ucfvit_cast")
    return dst


def transpose_batched(src_flat, dst_flat, table, n_mats, total_tiles):
    L = _l.load()
    _l.check(L.ucfvit_transpose_batched(src_flat.data_ptr(), dst_flat.data_ptr(), table.data_ptr(), n_mats, total_tiles, _stream()),
             "ucfvit_transpose_batched")


def linear_dgrad_t(dy2, wT, act_grad_aux=None, out=None, aux_is_deriv=False, c_colsum=None, c_colsum_accumulate=False):
    """dx[M,K] = dy2[M,N]·W with W given TRANSPOSED (wT [K,N]): both operands contraction-contiguous (fast path)"""
    M, N = dy2.shape
    K = wT.shape[0]
    return gemm(dy2, wT, M, K, N, LAYOUT_KC, LAYOUT_KC, out=out, act=_dgrad_act(act_grad_aux, aux_is_deriv), aux_in=act_grad_aux,
                c_colsum=c_colsum, c_colsum_accumulate=c_colsum_accumulate)

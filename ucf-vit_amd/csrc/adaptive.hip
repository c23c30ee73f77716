// Adaptive-patching front end for gfx950 (VIT.forward_features / VIT._pos_embed with adaptive_patching=True,
// reference src/UCF_VIT/simple/arch.py:465-467 and :366-393).  The token sequence arrives already cut and resized by the
// data loader as x[B][C][S][P] (S tokens of P = p^nd pixels per channel) with seq_ps[B][S][KIN] = (position..., size).
//   seq_patches  : 'b c s p -> b s (p c)' rows for LayerNorm(p^nd C) -> Linear -> LayerNorm(D)
//   adaptive_pos : out = cat(cls, x) + cat(0, GELU(seq_ps W^T + b))      (K = 3 or 4: FMA work, not a GEMM)
// All three kernels are HBM-bound streams; the parameter gradients of the position Linear are reduced deterministically
// (per-chunk partial sums, then one ordered pass), no float atomics.
#include "common.h"

namespace {

#define DTYPE_OK(d) ((d) == UCFVIT_F32 || (d) == UCFVIT_BF16)

// one workgroup per (b, s): C runs of P contiguous floats -> LDS -> one contiguous output row of P*C elements
template <typename T>
__global__ __launch_bounds__(256) void seq_patches_kernel(const float* __restrict__ x, T* __restrict__ out, int C, int64_t S, int P) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);   // [C][P]
    const int64_t bs = blockIdx.x;
    const int64_t b = bs / S, s = bs - b * S;
    const int n = C * P;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int c = i / P, p = i - c * P;
        tile[i] = x[((b * C + c) * S + s) * P + p];
    }
    __syncthreads();
    T* o = out + bs * n;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const int p = j / C, c = j - p * C;
        o[j] = from_f32<T>(tile[c * P + p]);
    }
}

// Thread = one 16-byte column vector of D, looping over a chunk of output rows (b, t), t in [0, S + pre).
template <typename T, int KIN>
__global__ __launch_bounds__(256) void adaptive_pos_fwd_kernel(const T* __restrict__ x, const float* __restrict__ seq_ps,
                                                               const T* __restrict__ w, const T* __restrict__ bias,
                                                               const T* __restrict__ cls, T* __restrict__ out, int64_t rows, int S, int D,
                                                               int pre, int rows_per_chunk) {
    constexpr int EPV = Vec16<T>::N;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v * EPV >= D) return;
    const int d0 = v * EPV;
    float wr[KIN][EPV], br[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        br[e] = to_f32<T>(bias[d0 + e]);
#pragma unroll
        for (int k = 0; k < KIN; ++k) wr[k][e] = to_f32<T>(w[(int64_t)(d0 + e) * KIN + k]);
    }
    Vec16<T> cv;
    if (pre) cv = *reinterpret_cast<const Vec16<T>*>(cls + d0);
    const int N = S + pre;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    for (int64_t r = r0; r < r1; ++r) {
        const int64_t b = r / N;
        const int t = (int)(r - b * N);
        Vec16<T> o;
        if (t < pre) {
            o = cv;                                             // cls + 0 (arch.py:381-385: the class token gets a zero position)
        } else {
            const int64_t tok = b * S + (t - pre);
            const Vec16<T> xv = *reinterpret_cast<const Vec16<T>*>(x + tok * D + d0);
            float sp[KIN];
#pragma unroll
            for (int k = 0; k < KIN; ++k) sp[k] = seq_ps[tok * KIN + k];
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float h = br[e];
#pragma unroll
                for (int k = 0; k < KIN; ++k) h = fmaf(sp[k], wr[k][e], h);
                o.set(e, xv.get(e) + gelu_f(h));
            }
        }
        *reinterpret_cast<Vec16<T>*>(out + r * D + d0) = o;
    }
}

// partial[chunk][j][D], j < KIN: sum_r dh * seq_ps[k]; j = KIN: sum_r dh; j = KIN+1: sum_b dout[b][0] (cls); dh = dout * gelu'(h)
template <typename T, int KIN>
__global__ __launch_bounds__(256) void adaptive_pos_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ seq_ps,
                                                               const T* __restrict__ w, const T* __restrict__ bias, T* __restrict__ dx,
                                                               float* __restrict__ partial, int64_t rows, int S, int D, int pre,
                                                               int rows_per_chunk) {
    constexpr int EPV = Vec16<T>::N;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v * EPV >= D) return;
    const int d0 = v * EPV;
    float wr[KIN][EPV], br[EPV], acc[KIN + 2][EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        br[e] = to_f32<T>(bias[d0 + e]);
#pragma unroll
        for (int k = 0; k < KIN; ++k) wr[k][e] = to_f32<T>(w[(int64_t)(d0 + e) * KIN + k]);
#pragma unroll
        for (int j = 0; j < KIN + 2; ++j) acc[j][e] = 0.f;
    }
    const int N = S + pre;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    for (int64_t r = r0; r < r1; ++r) {
        const int64_t b = r / N;
        const int t = (int)(r - b * N);
        const Vec16<T> dv = *reinterpret_cast<const Vec16<T>*>(dout + r * D + d0);
        if (t < pre) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) acc[KIN + 1][e] += dv.get(e);
            continue;
        }
        const int64_t tok = b * S + (t - pre);
        if (dx) *reinterpret_cast<Vec16<T>*>(dx + tok * D + d0) = dv;
        float sp[KIN];
#pragma unroll
        for (int k = 0; k < KIN; ++k) sp[k] = seq_ps[tok * KIN + k];
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            float h = br[e];
#pragma unroll
            for (int k = 0; k < KIN; ++k) h = fmaf(sp[k], wr[k][e], h);
            const float dh = dv.get(e) * gelu_grad_f(h);
            acc[KIN][e] += dh;
#pragma unroll
            for (int k = 0; k < KIN; ++k) acc[k][e] = fmaf(dh, sp[k], acc[k][e]);
        }
    }
    float* pp = partial + (int64_t)blockIdx.y * (KIN + 2) * D + d0;
#pragma unroll
    for (int j = 0; j < KIN + 2; ++j)
#pragma unroll
        for (int e = 0; e < EPV; ++e) pp[(int64_t)j * D + e] = acc[j][e];
}

// ordered sum over the chunks; dW is written in the parameter's [D][KIN] layout.  acc bits: 1 = dW, 2 = dbias, 4 = dcls
template <int KIN>
__global__ __launch_bounds__(256) void adaptive_pos_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                  float* __restrict__ dbias, float* __restrict__ dcls, int chunks, int D,
                                                                  int acc_bits) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    float s[KIN + 2];
#pragma unroll
    for (int j = 0; j < KIN + 2; ++j) s[j] = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const float* pp = partial + (int64_t)c * (KIN + 2) * D + d;
#pragma unroll
        for (int j = 0; j < KIN + 2; ++j) s[j] += pp[(int64_t)j * D];
    }
    if (dw) {
#pragma unroll
        for (int k = 0; k < KIN; ++k) {
            float* o = dw + (int64_t)d * KIN + k;
            *o = (acc_bits & 1) ? *o + s[k] : s[k];
        }
    }
    if (dbias) dbias[d] = (acc_bits & 2) ? dbias[d] + s[KIN] : s[KIN];
    if (dcls) dcls[d] = (acc_bits & 4) ? dcls[d] + s[KIN + 1] : s[KIN + 1];
}

struct PosPlan {
    int threads, gx, chunks, rpc;
};

inline PosPlan pos_plan(int64_t rows, int64_t D, int epv) {
    PosPlan p;
    const int nvec = (int)(D / epv);
    p.threads = nvec >= 256 ? 256 : ((nvec + 63) / 64) * 64;
    p.gx = (nvec + p.threads - 1) / p.threads;
    int64_t chunks = (rows + 63) / 64;            // >= 64 rows per workgroup so the W/bias preload is amortised
    if (chunks > 1024) chunks = 1024;
    if (chunks < 1) chunks = 1;
    p.rpc = (int)((rows + chunks - 1) / chunks);
    p.chunks = (int)((rows + p.rpc - 1) / p.rpc);
    return p;
}

}  // namespace

extern "C" int ucfvit_seq_patches(const float* x, void* rows_out, int64_t B, int64_t C, int64_t S, int64_t P, int dtype, void* stream) {
    UCF_CHECK_ARG(x && rows_out, "ucfvit_seq_patches: null pointer");
    UCF_CHECK_ARG(DTYPE_OK(dtype), "ucfvit_seq_patches: bad dtype %d", dtype);
    UCF_CHECK_ARG(B >= 0 && C > 0 && S > 0 && P > 0, "ucfvit_seq_patches: bad shape B=%lld C=%lld S=%lld P=%lld", (long long)B, (long long)C,
                  (long long)S, (long long)P);
    UCF_CHECK_ARG(C * P * 4 <= 64 * 1024, "ucfvit_seq_patches: C*P=%lld floats do not fit the 64 KiB staging tile", (long long)(C * P));
    UCF_CHECK_ARG(B * S < (1ll << 31), "ucfvit_seq_patches: B*S too large");
    if (B == 0) return UCFVIT_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = (size_t)(C * P) * sizeof(float);
    if (dtype == UCFVIT_F32)
        hipLaunchKernelGGL(seq_patches_kernel<float>, dim3((unsigned)(B * S)), dim3(256), smem, s, x, (float*)rows_out, (int)C, S, (int)P);
    else
        hipLaunchKernelGGL(seq_patches_kernel<bf16>, dim3((unsigned)(B * S)), dim3(256), smem, s, x, (bf16*)rows_out, (int)C, S, (int)P);
    UCF_LAUNCH_CHECK("ucfvit_seq_patches");
    return UCFVIT_OK;
}

extern "C" int ucfvit_adaptive_pos_fwd(const void* x, const float* seq_ps, const void* w, const void* bias, const void* cls, void* out,
                                       int64_t B, int64_t S, int64_t D, int kin, int has_cls, int dtype, void* stream) {
    UCF_CHECK_ARG(x && seq_ps && w && bias && out, "ucfvit_adaptive_pos_fwd: null pointer");
    UCF_CHECK_ARG(!has_cls || cls, "ucfvit_adaptive_pos_fwd: has_cls without cls pointer");
    UCF_CHECK_ARG(DTYPE_OK(dtype), "ucfvit_adaptive_pos_fwd: bad dtype %d", dtype);
    UCF_CHECK_ARG(kin == 3 || kin == 4, "ucfvit_adaptive_pos_fwd: kin=%d (3 for 2-D, 4 for 3-D input)", kin);
    const int epv = dtype == UCFVIT_F32 ? 4 : 8;
    UCF_CHECK_ARG(B >= 0 && S > 0 && D > 0 && D % epv == 0, "ucfvit_adaptive_pos_fwd: D=%lld must be a multiple of %d", (long long)D, epv);
    UCF_CHECK_ARG(ucf_is_aligned16(x) && ucf_is_aligned16(out) && ucf_is_aligned16(cls), "ucfvit_adaptive_pos_fwd: pointers must be 16-byte aligned");
    if (B == 0) return UCFVIT_OK;
    const int pre = has_cls ? 1 : 0;
    const int64_t rows = B * (S + pre);
    const PosPlan p = pos_plan(rows, D, epv);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(p.gx, p.chunks), block(p.threads);
#define POS_FWD(T, K)                                                                                                               \
    hipLaunchKernelGGL((adaptive_pos_fwd_kernel<T, K>), grid, block, 0, s, (const T*)x, seq_ps, (const T*)w, (const T*)bias, (const T*)cls, \
                       (T*)out, rows, (int)S, (int)D, pre, p.rpc)
    if (dtype == UCFVIT_F32) {
        if (kin == 3) POS_FWD(float, 3); else POS_FWD(float, 4);
    } else {
        if (kin == 3) POS_FWD(bf16, 3); else POS_FWD(bf16, 4);
    }
#undef POS_FWD
    UCF_LAUNCH_CHECK("ucfvit_adaptive_pos_fwd");
    return UCFVIT_OK;
}

extern "C" int64_t ucfvit_adaptive_pos_bwd_workspace(int64_t B, int64_t S, int64_t D, int kin, int has_cls, int dtype) {
    if (B <= 0 || S <= 0 || D <= 0 || (kin != 3 && kin != 4)) return 0;
    const PosPlan p = pos_plan(B * (S + (has_cls ? 1 : 0)), D, dtype == UCFVIT_F32 ? 4 : 8);
    return (int64_t)p.chunks * (kin + 2) * D * (int64_t)sizeof(float);
}

extern "C" int ucfvit_adaptive_pos_bwd(const void* dout, const float* seq_ps, const void* w, const void* bias, void* dx, float* dw, float* dbias,
                                       float* dcls, int64_t B, int64_t S, int64_t D, int kin, int has_cls, int accumulate, void* workspace,
                                       int dtype, void* stream) {
    UCF_CHECK_ARG(dout && seq_ps && w && bias && workspace, "ucfvit_adaptive_pos_bwd: null pointer");
    UCF_CHECK_ARG(DTYPE_OK(dtype), "ucfvit_adaptive_pos_bwd: bad dtype %d", dtype);
    UCF_CHECK_ARG(kin == 3 || kin == 4, "ucfvit_adaptive_pos_bwd: kin=%d (3 for 2-D, 4 for 3-D input)", kin);
    UCF_CHECK_ARG(accumulate >= 0 && accumulate <= 7, "ucfvit_adaptive_pos_bwd: accumulate is a bit mask (1 dW, 2 dbias, 4 dcls)");
    const int epv = dtype == UCFVIT_F32 ? 4 : 8;
    UCF_CHECK_ARG(B >= 0 && S > 0 && D > 0 && D % epv == 0, "ucfvit_adaptive_pos_bwd: D=%lld must be a multiple of %d", (long long)D, epv);
    UCF_CHECK_ARG(ucf_is_aligned16(dout) && ucf_is_aligned16(dx) && ucf_is_aligned16(workspace), "ucfvit_adaptive_pos_bwd: pointers must be 16-byte aligned");
    if (B == 0) return UCFVIT_OK;
    const int pre = has_cls ? 1 : 0;
    const int64_t rows = B * (S + pre);
    const PosPlan p = pos_plan(rows, D, epv);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(p.gx, p.chunks), block(p.threads);
    float* part = (float*)workspace;
#define POS_BWD(T, K)                                                                                                                     \
    hipLaunchKernelGGL((adaptive_pos_bwd_kernel<T, K>), grid, block, 0, s, (const T*)dout, seq_ps, (const T*)w, (const T*)bias, (T*)dx, part, \
                       rows, (int)S, (int)D, pre, p.rpc)
    if (dtype == UCFVIT_F32) {
        if (kin == 3) POS_BWD(float, 3); else POS_BWD(float, 4);
    } else {
        if (kin == 3) POS_BWD(bf16, 3); else POS_BWD(bf16, 4);
    }
#undef POS_BWD
    UCF_LAUNCH_CHECK("ucfvit_adaptive_pos_bwd");
    const dim3 rgrid((unsigned)((D + 255) / 256));
    if (kin == 3)
        hipLaunchKernelGGL(adaptive_pos_reduce_kernel<3>, rgrid, dim3(256), 0, s, part, dw, dbias, pre ? dcls : nullptr, p.chunks, (int)D, accumulate);
    else
        hipLaunchKernelGGL(adaptive_pos_reduce_kernel<4>, rgrid, dim3(256), 0, s, part, dw, dbias, pre ? dcls : nullptr, p.chunks, (int)D, accumulate);
    UCF_LAUNCH_CHECK("ucfvit_adaptive_pos_bwd(reduce)");
    return UCFVIT_OK;
}

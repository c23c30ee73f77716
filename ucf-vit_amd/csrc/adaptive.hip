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

// One workgroup per (b, group of TS consecutive tokens): for a fixed channel those tokens are one contiguous run of TS*P floats.
// float4 loads into an LDS tile [C][TS*P], then each thread assembles 16-byte pieces of the output rows (order (p, c)).
constexpr int SP_TS = 4;
template <typename T>
__global__ __launch_bounds__(256) void seq_patches_kernel(const float* __restrict__ x, T* __restrict__ out, int C, int64_t S, int P,
                                                          int groups_per_b) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);   // [C][SP_TS * P]
    constexpr int EPV = Vec16<T>::N;
    const int64_t b = blockIdx.x / groups_per_b;
    const int64_t s0 = (int64_t)(blockIdx.x - b * groups_per_b) * SP_TS;
    const int ts = (int)(S - s0 < SP_TS ? S - s0 : SP_TS);
    const int run = ts * P;                              // floats per channel for this group
    const int RUN = SP_TS * P;
    if ((P & 3) == 0) {
        const int run4 = run >> 2;
        for (int i = threadIdx.x; i < C * run4; i += blockDim.x) {
            const int c = i / run4, q = i - c * run4;
            *reinterpret_cast<f32x4*>(tile + c * RUN + 4 * q) = *reinterpret_cast<const f32x4*>(x + ((b * C + c) * S + s0) * P + 4 * q);
        }
    } else {
        for (int i = threadIdx.x; i < C * run; i += blockDim.x) {
            const int c = i / run, q = i - c * run;
            tile[c * RUN + q] = x[((b * C + c) * S + s0) * P + q];
        }
    }
    __syncthreads();
    const int n = C * P;                                 // elements per output row
    T* o = out + (b * S + s0) * n;
    if (n % EPV == 0) {
        const int nv = n / EPV;
        for (int i = threadIdx.x; i < ts * nv; i += blockDim.x) {
            const int t = i / nv, j0 = (i - t * nv) * EPV;
            Vec16<T> v;
            int p = j0 / C, c = j0 - p * C;              // (p, c) advance incrementally: one division per 16-byte piece
            const float* tt = tile + t * P;
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                v.set(e, tt[c * RUN + p]);
                if (++c == C) {
                    c = 0;
                    ++p;
                }
            }
            *reinterpret_cast<Vec16<T>*>(o + (int64_t)t * n + j0) = v;
        }
    } else {
        for (int i = threadIdx.x; i < ts * n; i += blockDim.x) {
            const int t = i / n, j = i - t * n, p = j / C, c = j - p * C;
            o[(int64_t)t * n + j] = from_f32<T>(tile[c * RUN + t * P + p]);
        }
    }
}

// Block = 256 threads = LANES row lanes x CPB column vectors (CPB = power of two >= D/EPV, <= 256).  A block covers
// rows_per_chunk output rows (b, t), t in [0, S + pre); each row lane walks its rows four at a time (independent loads in flight).
template <typename T, int KIN>
__global__ __launch_bounds__(256) void adaptive_pos_fwd_kernel(const T* __restrict__ x, const float* __restrict__ seq_ps,
                                                               const T* __restrict__ w, const T* __restrict__ bias,
                                                               const T* __restrict__ cls, T* __restrict__ out, int64_t rows, int S, int D,
                                                               int pre, int rows_per_chunk, int cpb_log2) {
    constexpr int EPV = Vec16<T>::N;
    const int cpb = 1 << cpb_log2, lanes = 256 >> cpb_log2;
    const int v = blockIdx.x * cpb + (threadIdx.x & (cpb - 1));
    const int rl = threadIdx.x >> cpb_log2;
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
    const int64_t b0 = r0 / N;
    const unsigned t0 = (unsigned)(r0 - b0 * N);
    constexpr int U = 4;
    for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)U * lanes) {
        Vec16<T> xv[U];
        float sp[U][KIN];
        int kind[U];            // 0: past the end, 1: class-token row, 2: token row
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = rb + (int64_t)u * lanes;
            kind[u] = 0;
            if (r < r1) {
                const unsigned loc = (unsigned)(r - r0) + t0, bq = loc / (unsigned)N;      // 32-bit: r - r0 < rows_per_chunk
                const int64_t b = b0 + bq;
                const int t = (int)(loc - bq * (unsigned)N);
                kind[u] = t < pre ? 1 : 2;
                if (t >= pre) {
                    const int64_t tok = b * S + (t - pre);
                    xv[u] = *reinterpret_cast<const Vec16<T>*>(x + tok * D + d0);
#pragma unroll
                    for (int k = 0; k < KIN; ++k) sp[u][k] = seq_ps[tok * KIN + k];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kind[u] == 0) continue;
            Vec16<T> o;
            if (kind[u] == 1) {
                o = cv;                                         // cls + 0 (arch.py:381-385: the class token gets a zero position)
            } else {
                float h[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    h[e] = br[e];
#pragma unroll
                    for (int k = 0; k < KIN; ++k) h[e] = fmaf(sp[u][k], wr[k][e], h[e]);
                }
                if constexpr (EPV == 8) {
                    gelu_fast8(h);          // bf16 output: rational erfc, |err| < 1e-6 (common.h), as in the GEMM epilogues
                } else {
#pragma unroll
                    for (int e = 0; e < EPV; ++e) h[e] = gelu_f(h[e]);
                }
#pragma unroll
                for (int e = 0; e < EPV; ++e) o.set(e, xv[u].get(e) + h[e]);
            }
            *reinterpret_cast<Vec16<T>*>(out + (rb + (int64_t)u * lanes) * D + d0) = o;
        }
    }
}

// partial[chunk][j][D], j < KIN: sum_r dh * seq_ps[k]; j = KIN: sum_r dh; j = KIN+1: sum_b dout[b][0] (cls); dh = dout * gelu'(h).
// The row lanes of a block are summed through LDS in lane order (fixed order: deterministic).
template <typename T, int KIN>
__global__ __launch_bounds__(256) void adaptive_pos_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ seq_ps,
                                                               const T* __restrict__ w, const T* __restrict__ bias, T* __restrict__ dx,
                                                               float* __restrict__ partial, int64_t rows, int S, int D, int pre,
                                                               int rows_per_chunk, int cpb_log2) {
    constexpr int EPV = Vec16<T>::N;
    constexpr int NJ = KIN + 2;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);     // [lanes - 1][cpb][NJ * EPV]
    const int cpb = 1 << cpb_log2, lanes = 256 >> cpb_log2;
    const int cv = threadIdx.x & (cpb - 1);
    const int v = blockIdx.x * cpb + cv;
    const int rl = threadIdx.x >> cpb_log2;
    const bool active = v * EPV < D;
    const int d0 = active ? v * EPV : 0;
    float wr[KIN][EPV], br[EPV], acc[NJ][EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        br[e] = to_f32<T>(bias[d0 + e]);
#pragma unroll
        for (int k = 0; k < KIN; ++k) wr[k][e] = to_f32<T>(w[(int64_t)(d0 + e) * KIN + k]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j][e] = 0.f;
    }
    const int N = S + pre;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    const int64_t b0 = r0 / N;
    const unsigned t0 = (unsigned)(r0 - b0 * N);
    constexpr int U = 4;
    if (active) {
        for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)U * lanes) {
            Vec16<T> dv[U];
            float sp[U][KIN];
            int64_t tok[U];
            int kind[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = rb + (int64_t)u * lanes;
                kind[u] = 0;
                if (r < r1) {
                    const unsigned loc = (unsigned)(r - r0) + t0, bq = loc / (unsigned)N;  // 32-bit: r - r0 < rows_per_chunk
                    const int64_t b = b0 + bq;
                    const int t = (int)(loc - bq * (unsigned)N);
                    kind[u] = t < pre ? 1 : 2;
                    dv[u] = *reinterpret_cast<const Vec16<T>*>(dout + r * D + d0);
                    if (t >= pre) {
                        tok[u] = b * S + (t - pre);
#pragma unroll
                        for (int k = 0; k < KIN; ++k) sp[u][k] = seq_ps[tok[u] * KIN + k];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (kind[u] == 0) continue;
                if (kind[u] == 1) {
#pragma unroll
                    for (int e = 0; e < EPV; ++e) acc[KIN + 1][e] += dv[u].get(e);
                    continue;
                }
                if (dx) *reinterpret_cast<Vec16<T>*>(dx + tok[u] * D + d0) = dv[u];
                float h[EPV], dhv[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    h[e] = br[e];
#pragma unroll
                    for (int k = 0; k < KIN; ++k) h[e] = fmaf(sp[u][k], wr[k][e], h[e]);
                    dhv[e] = dv[u].get(e);
                }
                if constexpr (EPV == 8) {
                    gelu_grad_fast8(dhv, h);
                } else {
#pragma unroll
                    for (int e = 0; e < EPV; ++e) dhv[e] *= gelu_grad_f(h[e]);
                }
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    acc[KIN][e] += dhv[e];
#pragma unroll
                    for (int k = 0; k < KIN; ++k) acc[k][e] = fmaf(dhv[e], sp[u][k], acc[k][e]);
                }
            }
        }
    }
    if (rl > 0) {
        float* dst = red + ((int64_t)(rl - 1) * cpb + cv) * (NJ * EPV);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < EPV; ++e) dst[j * EPV + e] = acc[j][e];
    }
    __syncthreads();
    if (rl == 0 && active) {
        for (int l = 1; l < lanes; ++l) {
            const float* src = red + ((int64_t)(l - 1) * cpb + cv) * (NJ * EPV);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < EPV; ++e) acc[j][e] += src[j * EPV + e];
        }
        float* pp = partial + (int64_t)blockIdx.y * NJ * D + d0;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < EPV; ++e) pp[(int64_t)j * D + e] = acc[j][e];
    }
}

// ordered sum over the chunks: block = 64 columns x 4 chunk lanes for one j (grid.y = KIN + 2); lane l sums chunks l, l+4, ... and the
// four lane sums are added in lane order.  dW is written in the parameter's [D][KIN] layout.  acc bits: 1 = dW, 2 = dbias, 4 = dcls
template <int KIN>
__global__ __launch_bounds__(256) void adaptive_pos_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                  float* __restrict__ dbias, float* __restrict__ dcls, int chunks, int D,
                                                                  int acc_bits) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, l = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + col;
    const int j = blockIdx.y;
    float s = 0.f;
    if (d < D) {
#pragma unroll 8
        for (int c = l; c < chunks; c += 4) s += partial[((int64_t)c * (KIN + 2) + j) * D + d];
    }
    red[l][col] = s;
    __syncthreads();
    if (l != 0 || d >= D) return;
    s = ((red[0][col] + red[1][col]) + red[2][col]) + red[3][col];
    if (j < KIN) {
        if (dw) {
            float* o = dw + (int64_t)d * KIN + j;
            *o = (acc_bits & 1) ? *o + s : s;
        }
    } else if (j == KIN) {
        if (dbias) dbias[d] = (acc_bits & 2) ? dbias[d] + s : s;
    } else if (dcls) {
        dcls[d] = (acc_bits & 4) ? dcls[d] + s : s;
    }
}

struct PosPlan {
    int cpb_log2, gx, chunks, rpc, lanes;
};

inline PosPlan pos_plan(int64_t rows, int64_t D, int epv, int rows_per_lane) {
    PosPlan p;
    const int nvec = (int)(D / epv);
    p.cpb_log2 = 0;
    while ((1 << p.cpb_log2) < nvec && p.cpb_log2 < 8) ++p.cpb_log2;
    const int cpb = 1 << p.cpb_log2;
    p.lanes = 256 / cpb;
    p.gx = (nvec + cpb - 1) / cpb;
    const int64_t min_rows = rows_per_lane * (int64_t)p.lanes;   // the W/bias preload (kin + 1 values per column) is amortised over these
    int64_t chunks = (rows + min_rows - 1) / min_rows;
    if (chunks > 4096) chunks = 4096;
    if (chunks < 1) chunks = 1;
    p.rpc = (int)((rows + chunks - 1) / chunks);
    p.chunks = (int)((rows + p.rpc - 1) / p.rpc);
    return p;
}

}  // namespace

extern "C" int ucfvit_seq_patches(const float* x, void* rows_out, int64_t B, int64_t C, int64_t S, int64_t P, int dtype, void* stream) {
    if (B == 0) return UCFVIT_OK;                      // empty batch (pointers may be NULL)
    UCF_CHECK_ARG(x && rows_out, "ucfvit_seq_patches: null pointer");
    UCF_CHECK_ARG(DTYPE_OK(dtype), "ucfvit_seq_patches: bad dtype %d", dtype);
    UCF_CHECK_ARG(B >= 0 && C > 0 && S > 0 && P > 0, "ucfvit_seq_patches: bad shape B=%lld C=%lld S=%lld P=%lld", (long long)B, (long long)C,
                  (long long)S, (long long)P);
    UCF_CHECK_ARG(C * P * SP_TS * 4 <= 64 * 1024, "ucfvit_seq_patches: C*P=%lld floats do not fit the staging tile (4 tokens, 64 KiB)", (long long)(C * P));
    UCF_CHECK_ARG(ucf_is_aligned16(x) && ucf_is_aligned16(rows_out), "ucfvit_seq_patches: pointers must be 16-byte aligned");
    UCF_CHECK_ARG(B * S < (1ll << 31), "ucfvit_seq_patches: B*S too large");
    if (B == 0) return UCFVIT_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = (size_t)(C * P) * SP_TS * sizeof(float);
    const int gpb = (int)((S + SP_TS - 1) / SP_TS);
    const unsigned grid = (unsigned)(B * gpb);
    if (dtype == UCFVIT_F32)
        hipLaunchKernelGGL(seq_patches_kernel<float>, dim3(grid), dim3(256), smem, s, x, (float*)rows_out, (int)C, S, (int)P, gpb);
    else
        hipLaunchKernelGGL(seq_patches_kernel<bf16>, dim3(grid), dim3(256), smem, s, x, (bf16*)rows_out, (int)C, S, (int)P, gpb);
    UCF_LAUNCH_CHECK("ucfvit_seq_patches");
    return UCFVIT_OK;
}

extern "C" int ucfvit_adaptive_pos_fwd(const void* x, const float* seq_ps, const void* w, const void* bias, const void* cls, void* out,
                                       int64_t B, int64_t S, int64_t D, int kin, int has_cls, int dtype, void* stream) {
    if (B == 0) return UCFVIT_OK;                      // empty batch (pointers may be NULL)
    UCF_CHECK_ARG(x && seq_ps && w && bias && out, "ucfvit_adaptive_pos_fwd: null pointer");
    UCF_CHECK_ARG(!has_cls || cls, "ucfvit_adaptive_pos_fwd: has_cls without cls pointer");
    UCF_CHECK_ARG(DTYPE_OK(dtype), "ucfvit_adaptive_pos_fwd: bad dtype %d", dtype);
    UCF_CHECK_ARG(kin == 3 || kin == 4, "ucfvit_adaptive_pos_fwd: kin=%d (3 for 2-D, 4 for 3-D input)", kin);
    const int epv = dtype == UCFVIT_F32 ? 4 : 8;
    UCF_CHECK_ARG(B >= 0 && S > 0 && D > 0 && D % epv == 0, "ucfvit_adaptive_pos_fwd: D=%lld must be a multiple of %d", (long long)D, epv);
    UCF_CHECK_ARG(B * (S + 1) < (1ll << 30), "ucfvit_adaptive_pos_fwd: B*(S+1) too large");
    UCF_CHECK_ARG(ucf_is_aligned16(x) && ucf_is_aligned16(out) && ucf_is_aligned16(cls), "ucfvit_adaptive_pos_fwd: pointers must be 16-byte aligned");
    if (B == 0) return UCFVIT_OK;
    const int pre = has_cls ? 1 : 0;
    const int64_t rows = B * (S + pre);
    const PosPlan p = pos_plan(rows, D, epv, 8);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(p.gx, p.chunks), block(256);
#define POS_FWD(T, K)                                                                                                               \
    hipLaunchKernelGGL((adaptive_pos_fwd_kernel<T, K>), grid, block, 0, s, (const T*)x, seq_ps, (const T*)w, (const T*)bias, (const T*)cls, \
                       (T*)out, rows, (int)S, (int)D, pre, p.rpc, p.cpb_log2)
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
    const PosPlan p = pos_plan(B * (S + (has_cls ? 1 : 0)), D, dtype == UCFVIT_F32 ? 4 : 8, 32);
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
    UCF_CHECK_ARG(B * (S + 1) < (1ll << 30), "ucfvit_adaptive_pos_bwd: B*(S+1) too large");
    UCF_CHECK_ARG(ucf_is_aligned16(dout) && ucf_is_aligned16(dx) && ucf_is_aligned16(workspace), "ucfvit_adaptive_pos_bwd: pointers must be 16-byte aligned");
    if (B == 0) return UCFVIT_OK;
    const int pre = has_cls ? 1 : 0;
    const int64_t rows = B * (S + pre);
    const PosPlan p = pos_plan(rows, D, epv, 32);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(p.gx, p.chunks), block(256);
    float* part = (float*)workspace;
    const size_t smem = (size_t)(p.lanes - 1) * (256 / p.lanes) * (kin + 2) * epv * sizeof(float);
#define POS_BWD(T, K)                                                                                                                     \
    hipLaunchKernelGGL((adaptive_pos_bwd_kernel<T, K>), grid, block, smem, s, (const T*)dout, seq_ps, (const T*)w, (const T*)bias, (T*)dx, part, \
                       rows, (int)S, (int)D, pre, p.rpc, p.cpb_log2)
    if (dtype == UCFVIT_F32) {
        if (kin == 3) POS_BWD(float, 3); else POS_BWD(float, 4);
    } else {
        if (kin == 3) POS_BWD(bf16, 3); else POS_BWD(bf16, 4);
    }
#undef POS_BWD
    UCF_LAUNCH_CHECK("ucfvit_adaptive_pos_bwd");
    const dim3 rgrid((unsigned)((D + 63) / 64), (unsigned)(kin + 2));
    if (kin == 3)
        hipLaunchKernelGGL(adaptive_pos_reduce_kernel<3>, rgrid, dim3(256), 0, s, part, dw, dbias, pre ? dcls : nullptr, p.chunks, (int)D, accumulate);
    else
        hipLaunchKernelGGL(adaptive_pos_reduce_kernel<4>, rgrid, dim3(256), 0, s, part, dw, dbias, pre ? dcls : nullptr, p.chunks, (int)D, accumulate);
    UCF_LAUNCH_CHECK("ucfvit_adaptive_pos_bwd(reduce)");
    return UCFVIT_OK;
}

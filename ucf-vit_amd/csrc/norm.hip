// LayerNorm forward/backward and column sums (bias gradients) for gfx950.  HBM-bound kernels:
// one wave (64 lanes) per row, 16-byte vector loads, fp32 statistics, wavefront shuffle reductions,
// deterministic two-stage reductions for the parameter gradients (no atomics).
#include "common.h"

namespace {

constexpr int LN_THREADS = 256;  // 4 waves = 4 rows in flight per workgroup

// ---------------------------------------------------------------------------------------------------
// forward: y = (x - mean) * rstd * gamma + beta ; two-pass (mean, then centred variance) on register-cached rows
template <typename T, int NV>  // NV = 16-B vectors per lane (row length <= 64*NV*EPV)
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                                                            const T* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                            int D, float eps) {
    constexpr int EPV = Vec16<T>::N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = D / EPV;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const T* xr = x + row * D;
        Vec16<T> xv[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < nvec) {
                xv[i] = *reinterpret_cast<const Vec16<T>*>(xr + v * EPV);
#pragma unroll
                for (int e = 0; e < EPV; ++e) s += xv[i].get(e);
            }
        }
        const float mu = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < nvec) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const float d = xv[i].get(e) - mu;
                    q += d * d;
                }
            }
        }
        const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
        T* yr = y + row * D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < nvec) {
                const Vec16<T> gv = *reinterpret_cast<const Vec16<T>*>(gamma + v * EPV);
                const Vec16<T> bv = *reinterpret_cast<const Vec16<T>*>(beta + v * EPV);
                Vec16<T> o;
#pragma unroll
                for (int e = 0; e < EPV; ++e) o.set(e, (xv[i].get(e) - mu) * rs * gv.get(e) + bv.get(e));
                *reinterpret_cast<Vec16<T>*>(yr + v * EPV) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward: dx = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*gamma ; per-workgroup partial dgamma/dbeta
// DXS: also the column sums of dx (+ dres) = the bias gradient of the Linear layer that produced LayerNorm's input's residual
// partner (proj / fc2 of the Block before): third row of the per-workgroup partials
template <typename T, int NV, bool DXS>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const T* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const T* __restrict__ dres,
                                                            T* __restrict__ dx, float* __restrict__ partial, int64_t rows, int D) {
    constexpr int EPV = Vec16<T>::N;
    __shared__ float red[4][64 * 8 + 8];  // cross-wave reduction staging, one 16-B vector slot (as fp32 x EPV) at a time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = D / EPV;
    float dg[NV][EPV], db[NV][EPV], dxs[DXS ? NV : 1][EPV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < EPV; ++e) dg[i][e] = db[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < (DXS ? NV : 1); ++i)
#pragma unroll
        for (int e = 0; e < EPV; ++e) dxs[i][e] = 0.f;
    Vec16<T> gv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
        if (v < nvec) gv[i] = *reinterpret_cast<const Vec16<T>*>(gamma + v * EPV);
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        Vec16<T> xv[NV], dv[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < nvec) {
                xv[i] = *reinterpret_cast<const Vec16<T>*>(x + row * D + v * EPV);
                dv[i] = *reinterpret_cast<const Vec16<T>*>(dy + row * D + v * EPV);
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const float xh = (xv[i].get(e) - mu) * rs;
                    const float d = dv[i].get(e);
                    const float g = d * gv[i].get(e);
                    s1 += g;
                    s2 += g * xh;
                    dg[i][e] += d * xh;
                    db[i][e] += d;
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            if (v < nvec) {
                Vec16<T> o, rv;
                if (dres) rv = *reinterpret_cast<const Vec16<T>*>(dres + row * D + v * EPV);
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const float xh = (xv[i].get(e) - mu) * rs;
                    const float g = dv[i].get(e) * gv[i].get(e);
                    float r = rs * (g - c1 - xh * c2);
                    if (dres) r += rv.get(e);  // fused residual-branch gradient
                    o.set(e, r);
                    if constexpr (DXS) dxs[i][e] += r;
                }
                *reinterpret_cast<Vec16<T>*>(dx + row * D + v * EPV) = o;
            }
        }
    }
    // reduce the 4 waves' column partials through LDS, then one row of `partial` per workgroup: [grid][2 or 3][D]
    constexpr int NOUT = DXS ? 3 : 2;
    float* pg = partial + (int64_t)blockIdx.x * NOUT * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
#pragma unroll
        for (int which = 0; which < NOUT; ++which) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < EPV; ++e) red[wave][lane * EPV + e] = which == 0 ? dg[i][e] : (which == 1 ? db[i][e] : dxs[DXS ? i : 0][e]);
            __syncthreads();
            if (wave == 0 && v < nvec) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const float t = red[0][lane * EPV + e] + red[1][lane * EPV + e] + red[2][lane * EPV + e] + red[3][lane * EPV + e];
                    pg[which * D + v * EPV + e] = t;
                }
            }
        }
    }
}

// out[j] (+)= sum_b partial[b][j], j in [0, W).  Workgroup = 16 columns x 16 row groups, four independent loads in flight per
// thread (fixed summation order).  The grid is W/16 workgroups: with <= 512 partial rows this kernel is pure latency, so it wants
// many short threads rather than few long ones (measured 13.6 us per launch at 32 x 8, 147 launches per ViT-L step).
constexpr int RP_COLS = 16, RP_GROUPS = 16;
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out0,
                                                              float* __restrict__ out1, int nblocks, int W0, int accumulate,
                                                              float* __restrict__ out2 = nullptr, int accumulate2 = 0) {
    // partial rows are [1, 2 or 3][W0]: first W0 -> out0, next W0 -> out1, last W0 -> out2 (NULL outputs end the row)
    __shared__ float red[RP_GROUPS][RP_COLS + 1];
    const int cl = threadIdx.x % RP_COLS, rg = threadIdx.x / RP_COLS;
    const int j = blockIdx.x * RP_COLS + cl;
    const int W = out2 ? 3 * W0 : (out1 ? 2 * W0 : W0);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < W) {
        const float* p = partial + j;
        int b = rg;
        for (; b + 3 * RP_GROUPS < nblocks; b += 4 * RP_GROUPS) {
            s0 += p[(int64_t)b * W];
            s1 += p[(int64_t)(b + RP_GROUPS) * W];
            s2 += p[(int64_t)(b + 2 * RP_GROUPS) * W];
            s3 += p[(int64_t)(b + 3 * RP_GROUPS) * W];
        }
        for (; b < nblocks; b += RP_GROUPS) s0 += p[(int64_t)b * W];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && j < W) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < RP_GROUPS; ++k) t += red[k][cl];
        float* o = (j < W0) ? (out0 + j) : (j < 2 * W0 ? out1 + (j - W0) : out2 + (j - 2 * W0));
        const int acc = j < 2 * W0 ? accumulate : accumulate2;
        *o = acc ? (*o + t) : t;
    }
}

// ---------------------------------------------------------------------------------------------------
// column sums: a wave covers 64 x 16 B = 1 KiB of contiguous columns of one row (full-line coalescing), the 4 waves of a
// workgroup take rows r, r+1, r+2, r+3 ...; grid.y row chunks -> partial[chunk][N], reduced in a fixed order afterwards
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ partial, int64_t M, int N,
                                                     int64_t ldx, int rows_per_chunk) {
    constexpr int EPV = Vec16<T>::N;
    __shared__ float red[4][64 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * EPV;
    const int64_t r_begin = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r_end = min(M, r_begin + rows_per_chunk);
    float acc[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
    if (col < N) {
        int64_t r = r_begin + wave;
        for (; r + 12 < r_end; r += 16) {   // 4 independent loads in flight per lane
            const Vec16<T> v0 = *reinterpret_cast<const Vec16<T>*>(x + r * ldx + col);
            const Vec16<T> v1 = *reinterpret_cast<const Vec16<T>*>(x + (r + 4) * ldx + col);
            const Vec16<T> v2 = *reinterpret_cast<const Vec16<T>*>(x + (r + 8) * ldx + col);
            const Vec16<T> v3 = *reinterpret_cast<const Vec16<T>*>(x + (r + 12) * ldx + col);
#pragma unroll
            for (int e = 0; e < EPV; ++e) acc[e] += (v0.get(e) + v1.get(e)) + (v2.get(e) + v3.get(e));
        }
        for (; r < r_end; r += 4) {
            const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(x + r * ldx + col);
#pragma unroll
            for (int e = 0; e < EPV; ++e) acc[e] += v.get(e);
        }
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) red[wave][lane * EPV + e] = acc[e];
    __syncthreads();
    if (wave == 0 && col < N) {
#pragma unroll
        for (int e = 0; e < EPV; ++e)
            partial[(int64_t)blockIdx.y * N + col + e] = (red[0][lane * EPV + e] + red[1][lane * EPV + e]) + (red[2][lane * EPV + e] + red[3][lane * EPV + e]);
    }
}

template <typename T>
__global__ void colsum_scalar_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t M, int N, int64_t ldx,
                                     int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int64_t r = 0; r < M; ++r) s += to_f32<T>(x[r * ldx + n]);
    out[n] = accumulate ? out[n] + s : s;
}

inline int colsum_chunks(int64_t M) {
    int64_t c = (M + 127) / 128;
    if (c > 256) c = 256;
    if (c < 1) c = 1;
    return (int)c;
}

// workgroups of 4 waves (one row per wave at a time).  Forward: 2048 workgroups = 32 waves per CU (bytes in flight = resident
// waves x one row; +0.9 % on the ViT-L step over 512).  Backward: 768 = the three workgroups per CU its registers allow (61 us at
// 512, 54 us at 768, 68 us at 832 where a second partial round starts; tools/ln_bench.py).
inline int ln_grid_cap(bool backward) { return backward ? 768 : 2048; }
inline int ln_grid(int64_t rows, bool backward = true) {
    int64_t g = (rows + 3) / 4;
    const int cap = ln_grid_cap(backward);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

template <typename T> int ln_nv(int64_t D) {
    constexpr int EPV = 16 / sizeof(T);
    if (D % EPV) return -1;
    const int64_t nvec = D / EPV;
    if (nvec <= 64) return 1;
    if (nvec <= 128) return 2;
    if (nvec <= 256) return 4;
    if (nvec <= 512) return 8;
    return -1;
}

template <typename T>
int ln_fwd_t(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int64_t rows, int64_t D,
             float eps, hipStream_t s) {
    const int nv = ln_nv<T>(D);
    UCF_CHECK_ARG(nv > 0, "ucfvit_layernorm_fwd: D=%lld must be a multiple of %d and <= %d", (long long)D, (int)(16 / sizeof(T)),
                  (int)(512 * 16 / sizeof(T)));
    const dim3 grid(ln_grid(rows, false)), block(LN_THREADS);
#define LN_FWD(NVV)                                                                                                        \
    hipLaunchKernelGGL((ln_fwd_kernel<T, NVV>), grid, block, 0, s, (const T*)x, (const T*)gamma, (const T*)beta, (T*)y, mean, \
                       rstd, rows, (int)D, eps)
    switch (nv) {
        case 1: LN_FWD(1); break;
        case 2: LN_FWD(2); break;
        case 4: LN_FWD(4); break;
        default: LN_FWD(8); break;
    }
#undef LN_FWD
    UCF_LAUNCH_CHECK("ucfvit_layernorm_fwd");
    return UCFVIT_OK;
}

template <typename T>
int ln_bwd_t(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd, const void* dres, void* dx, float* dgamma,
             float* dbeta, int64_t rows, int64_t D, int accumulate, float* dx_colsum, int dx_colsum_accumulate, void* ws, hipStream_t s) {
    const int nv = ln_nv<T>(D);
    UCF_CHECK_ARG(nv > 0 && nv <= 4, "ucfvit_layernorm_bwd: D=%lld must be a multiple of %d and <= %d", (long long)D,
                  (int)(16 / sizeof(T)), (int)(256 * 16 / sizeof(T)));
    const int g = ln_grid(rows);
    const dim3 grid(g), block(LN_THREADS);
#define LN_BWD(NVV)                                                                                                            \
    do {                                                                                                                       \
        if (dx_colsum)                                                                                                         \
            hipLaunchKernelGGL((ln_bwd_kernel<T, NVV, true>), grid, block, 0, s, (const T*)dy, (const T*)x, (const T*)gamma, mean, rstd, \
                               (const T*)dres, (T*)dx, (float*)ws, rows, (int)D);                                             \
        else                                                                                                                   \
            hipLaunchKernelGGL((ln_bwd_kernel<T, NVV, false>), grid, block, 0, s, (const T*)dy, (const T*)x, (const T*)gamma, mean, rstd, \
                               (const T*)dres, (T*)dx, (float*)ws, rows, (int)D);                                             \
    } while (0)
    switch (nv) {
        case 1: LN_BWD(1); break;
        case 2: LN_BWD(2); break;
        default: LN_BWD(4); break;
    }
#undef LN_BWD
    UCF_LAUNCH_CHECK("ucfvit_layernorm_bwd");
    const int W = (dx_colsum ? 3 : 2) * (int)D;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((W + RP_COLS - 1) / RP_COLS), dim3(256), 0, s, (const float*)ws, dgamma, dbeta, g, (int)D,
                       accumulate, dx_colsum, dx_colsum_accumulate);
    UCF_LAUNCH_CHECK("ucfvit_layernorm_bwd(reduce)");
    return UCFVIT_OK;
}

}  // namespace

extern "C" int ucfvit_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                                    int64_t rows, int64_t D, float eps, int dtype, void* stream) {
    if (rows == 0) return UCFVIT_OK;                   // empty batch (pointers may be NULL)
    UCF_CHECK_ARG(x && gamma && beta && y && mean && rstd, "ucfvit_layernorm_fwd: null pointer");
    UCF_CHECK_ARG(rows >= 0 && D > 0, "ucfvit_layernorm_fwd: bad shape rows=%lld D=%lld", (long long)rows, (long long)D);
    UCF_CHECK_ARG(ucf_is_aligned16(x) && ucf_is_aligned16(y) && ucf_is_aligned16(gamma) && ucf_is_aligned16(beta),
                  "ucfvit_layernorm_fwd: pointers must be 16-byte aligned");
    if (rows == 0) return UCFVIT_OK;
    if (dtype == UCFVIT_F32) return ln_fwd_t<float>(x, gamma, beta, y, mean, rstd, rows, D, eps, (hipStream_t)stream);
    if (dtype == UCFVIT_BF16) return ln_fwd_t<bf16>(x, gamma, beta, y, mean, rstd, rows, D, eps, (hipStream_t)stream);
    ucfvit_set_error("ucfvit_layernorm_fwd: bad dtype %d", dtype);
    return UCFVIT_ERR_UNSUPPORTED;
}

extern "C" int64_t ucfvit_layernorm_bwd_workspace(int64_t rows, int64_t D) {
    return (int64_t)ln_grid(rows) * 3 * D * (int64_t)sizeof(float);      // dgamma, dbeta and (optional) dx column-sum partials
}

extern "C" int ucfvit_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                                    const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t D, int accumulate,
                                    float* dx_colsum, int dx_colsum_accumulate, void* workspace, int dtype, void* stream) {
    UCF_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && workspace, "ucfvit_layernorm_bwd: null pointer");
    UCF_CHECK_ARG(rows > 0 && D > 0, "ucfvit_layernorm_bwd: bad shape");
    UCF_CHECK_ARG(ucf_is_aligned16(x) && ucf_is_aligned16(dy) && ucf_is_aligned16(dx) && ucf_is_aligned16(gamma) && ucf_is_aligned16(dres),
                  "ucfvit_layernorm_bwd: pointers must be 16-byte aligned");
    if (dtype == UCFVIT_F32)
        return ln_bwd_t<float>(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, D, accumulate, dx_colsum, dx_colsum_accumulate, workspace,
                               (hipStream_t)stream);
    if (dtype == UCFVIT_BF16)
        return ln_bwd_t<bf16>(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, D, accumulate, dx_colsum, dx_colsum_accumulate, workspace,
                              (hipStream_t)stream);
    ucfvit_set_error("ucfvit_layernorm_bwd: bad dtype %d", dtype);
    return UCFVIT_ERR_UNSUPPORTED;
}

extern "C" int64_t ucfvit_colsum_workspace(int64_t M, int64_t N) { return (int64_t)colsum_chunks(M) * N * (int64_t)sizeof(float); }

extern "C" int ucfvit_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ldx, int accumulate, void* workspace,
                             int dtype, void* stream) {
    UCF_CHECK_ARG(x && out, "ucfvit_colsum: null pointer");
    UCF_CHECK_ARG(M >= 0 && N > 0 && ldx >= N, "ucfvit_colsum: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int epv = dtype == UCFVIT_F32 ? 4 : 8;
    UCF_CHECK_ARG(dtype == UCFVIT_F32 || dtype == UCFVIT_BF16, "ucfvit_colsum: bad dtype %d", dtype);
    const bool vec = ucf_is_aligned16(x) && (N % epv == 0) && (ldx % epv == 0) && workspace && M > 0;
    if (!vec) {
        if (dtype == UCFVIT_F32)
            hipLaunchKernelGGL(colsum_scalar_kernel<float>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, (const float*)x, out,
                               M, (int)N, ldx, accumulate);
        else
            hipLaunchKernelGGL(colsum_scalar_kernel<bf16>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, (const bf16*)x, out, M,
                               (int)N, ldx, accumulate);
        UCF_LAUNCH_CHECK("ucfvit_colsum(scalar)");
        return UCFVIT_OK;
    }
    const int chunks = colsum_chunks(M);
    const int rpc = (int)((M + chunks - 1) / chunks);
    const dim3 grid((unsigned)((N / epv + 63) / 64), chunks);
    if (dtype == UCFVIT_F32)
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (float*)workspace, M, (int)N, ldx, rpc);
    else
        hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)x, (float*)workspace, M, (int)N, ldx, rpc);
    UCF_LAUNCH_CHECK("ucfvit_colsum");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((N + RP_COLS - 1) / RP_COLS)), dim3(256), 0, s, (const float*)workspace, out,
                       (float*)nullptr, chunks, (int)N, accumulate);
    UCF_LAUNCH_CHECK("ucfvit_colsum(reduce)");
    return UCFVIT_OK;
}

extern "C" int ucfvit_reduce_rows(const float* partial, float* out, int64_t rows, int64_t N, int accumulate, void* stream) {
    UCF_CHECK_ARG(partial && out, "ucfvit_reduce_rows: null pointer");
    UCF_CHECK_ARG(rows > 0 && N > 0 && rows < (1ll << 31) && N < (1ll << 31), "ucfvit_reduce_rows: bad shape");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((N + RP_COLS - 1) / RP_COLS)), dim3(256), 0, (hipStream_t)stream, partial, out,
                       (float*)nullptr, (int)rows, (int)N, accumulate);
    UCF_LAUNCH_CHECK("ucfvit_reduce_rows");
    return UCFVIT_OK;
}

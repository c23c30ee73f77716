// bf16 MFMA GEMM v2 for gfx950: large tiles, direct global->LDS DMA, double-buffered LDS, optional split-K.
//
//   tile      : 256x256x64 (8 waves as 2x4, 128x64 per wave, 1 workgroup/CU, 128 KiB LDS)  — fwd / dgrad (M = B*N tokens is huge)
//               128x128x64 (4 waves as 2x2,  64x64 per wave, 2 workgroups/CU, 64 KiB LDS)  — small outputs (weight gradients), + split-K
//   staging   : __builtin_amdgcn_global_load_lds (16 B/lane, 1 KiB per wave-instruction) straight into the LDS image of the
//               NEXT K-tile while the MFMAs of the current one run; the LDS image is lane-linear, so the bank-conflict
//               swizzle is applied to the per-lane SOURCE address and undone by the same XOR on the fragment reads.
//   operands  : KC (contraction contiguous) images [rows][128 B] read by ds_read_b128; KS (contraction strided) images
//               [k][rows] read by ds_read_b64_tr_b16 (hardware transpose) — weight-gradient and data-gradient GEMMs need
//               no transposed copies of activations or weights in HBM.
//   edges     : rows/cols beyond M/N are CLAMPED on load (duplicates of valid data) and masked in the epilogue;
//               the contraction extent must be a multiple of 64 (host-checked, else gemm.hip's v1 kernel runs).
//   split-K   : gridDim.y K-slices write fp32 partial matrices to a caller workspace; a second kernel sums them in a fixed
//               order (deterministic) and applies accumulate.  Used when the output has too few tiles to fill 256 CUs.
#include "common.h"

namespace {

constexpr int BK2 = 64;

__device__ __forceinline__ int kc_off2(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }
__device__ __forceinline__ int ks_swz2(int krow) { return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 5; }

template <int BR>  // BR = tile extent along the operand's row/col index (128 or 256)
__device__ __forceinline__ int ks_off2(int krow, int colbyte) { return krow * (BR * 2) + (colbyte ^ ks_swz2(krow)); }

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Issue the DMA loads of one operand K-tile.  Every wave issues NPW wave-instructions of 1 KiB.
template <int LAYOUT, int BR, int NWAVES>
__device__ __forceinline__ void issue_tile(const bf16* __restrict__ base, int64_t ld, int r0, int k0, int R, char* lds, int wave, int lane) {
    constexpr int NPIECES = BR / 8;  // 1-KiB pieces per K-tile (tile bytes = BR * 128)
    constexpr int NPW = NPIECES / NWAVES;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int idx = wave * NPW + i;
        const bf16* src;
        if (LAYOUT == UCFVIT_LAYOUT_KC) {
            const int row = idx * 8 + (lane >> 3);
            const int gslot = (lane & 7) ^ (row & 7);
            int gr = r0 + row;
            gr = gr < R ? gr : R - 1;
            src = base + (int64_t)gr * ld + k0 + gslot * 8;
        } else {
            constexpr int RB = BR * 2;          // bytes per k-row
            constexpr int KPP = 1024 / RB;      // k-rows per piece
            const int krow = idx * KPP + (lane * 16) / RB;
            const int pbyte = (lane * 16) % RB;
            const int col = (pbyte ^ ks_swz2(krow)) >> 1;
            int gc = r0 + col;
            gc = gc <= R - 8 ? gc : R - 8;
            src = base + (int64_t)(k0 + krow) * ld + gc;
        }
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + idx * 1024), 16, 0, 0);
    }
}

template <int LAYOUT, int BR>
__device__ __forceinline__ bf16x8 load_frag2(const char* lds, int rbase, int c, int lane) {
    const int g = lane >> 4, i = lane & 15;
    if constexpr (LAYOUT == UCFVIT_LAYOUT_KC) {
        return *reinterpret_cast<const bf16x8*>(lds + kc_off2(rbase + i, 4 * c + g));
    } else {
        const int kb = 32 * c + 8 * g, q = i >> 2, p = i & 3;
        const int colbyte = (rbase + 4 * p) * 2;
        const char* a0 = lds + ks_off2<BR>(kb + q, colbyte);
        const char* a1 = lds + ks_off2<BR>(kb + 4 + q, colbyte);
        short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, a0));
        short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, a1));
        short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, r);
    }
}

struct Epi2 {
    const bf16* bias;
    const bf16* residual;
    const bf16* aux_in;
    bf16* aux_out;
    int64_t ldc, ldr, ldaux;
    int act, accumulate;
    float alpha;
    float* slab;       // split-K: fp32 partial matrices [splits][M][N] (ld = N); NULL when gridDim.y == 1
};

template <int LA, int LB, int BM, int BN, int WM, int WN, typename OutT>
__global__ __launch_bounds__(WM* WN * 64) void gemm2_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, OutT* __restrict__ C,
                                                             int M, int N, int K, int64_t lda, int64_t ldb, Epi2 ep, int tiles_m,
                                                             int tiles_n, int k_per_split) {
    constexpr int NWAVES = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nwg = tiles_m * tiles_n;
    const int t = xcd_remap(blockIdx.x, nwg);
    constexpr int BAND = 8;
    const int band_tiles = BAND * tiles_m;
    const int band = t / band_tiles;
    const int band_w = min(BAND, tiles_n - band * BAND);
    const int in_band = t - band * band_tiles;
    const int tm = in_band / band_w;
    const int tn = band * BAND + in_band % band_w;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / WN) * TM, wn = (wave % WN) * TN;

    const int k_begin = blockIdx.y * k_per_split;
    const int k_end = min(K, k_begin + k_per_split);
    const int nk = (k_end - k_begin) / BK2;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue_tile<LA, BM, NWAVES>(A, lda, m0, k_begin, M, smem, wave, lane);
    issue_tile<LB, BN, NWAVES>(B, ldb, n0, k_begin, N, smem + A_BYTES, wave, lane);

    for (int kt = 0; kt < nk; ++kt) {
        // K-tile kt has landed (vmcnt(0) of every wave, then the barrier); everyone is done reading the other buffer
        __syncthreads();
        const char* bufA = smem + (kt & 1) * BUF;
        const char* bufB = bufA + A_BYTES;
        if (kt + 1 < nk) {
            char* nb = smem + ((kt + 1) & 1) * BUF;
            issue_tile<LA, BM, NWAVES>(A, lda, m0, k_begin + (kt + 1) * BK2, M, nb, wave, lane);
            issue_tile<LB, BN, NWAVES>(B, ldb, n0, k_begin + (kt + 1) * BK2, N, nb + A_BYTES, wave, lane);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x8 fb[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = load_frag2<LB, BN>(bufB, wn + 16 * j, c, lane);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const bf16x8 fa = load_frag2<LA, BM>(bufA, wm + 16 * i, c, lane);
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa, acc[i][j], 0, 0, 0);  // roles swapped: D[n][m]
            }
        }
    }

    // Epilogue through LDS so every global access is a full 128-B line:
    //   phase A: v = alpha*acc + bias (fp32) of a 32-row x TN-col strip -> wave-private LDS rows
    //   phase B: each lane owns 8 consecutive columns of a row: activation / aux / residual / accumulate with 16-B loads, 16-B stores
    __syncthreads();   // all waves finished reading the pipeline buffers: LDS is reusable
    constexpr int PADW = TN + 4;                       // fp32 words per staged row (pad keeps 16-B alignment, spreads banks)
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * PADW);
    const int g = lane >> 4, li = lane & 15;
    constexpr int LPR = TN / 8;                        // lanes per staged row
    constexpr int RPI = 64 / LPR;                      // rows per wave-instruction
    const int prow = lane / LPR, pcol = (lane % LPR) * 8;
    float* slab = ep.slab ? ep.slab + (int64_t)blockIdx.y * M * N : nullptr;
#pragma unroll
    for (int p = 0; p < FM / 2; ++p) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * p + ii;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                f32x4 v = acc[i][j];
                if (!slab) {
                    const int n = n0 + wn + 16 * j + 4 * g;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= ep.alpha;
                    if (ep.bias && n < N) {
                        const Vec4<bf16> b = *reinterpret_cast<const Vec4<bf16>*>(ep.bias + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += b.get(r);
                    }
                }
                *reinterpret_cast<f32x4*>(stage + (16 * ii + li) * PADW + 16 * j + 4 * g) = v;
            }
        }
        // wave-private region: the compiler's lgkmcnt wait orders the ds_writes above before the ds_reads below
#pragma unroll
        for (int rr = 0; rr < 32; rr += RPI) {
            const int row = rr + prow;
            const int m = m0 + wm + 32 * p + row;
            const int n = n0 + wn + pcol;
            float v[8];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * PADW + pcol);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * PADW + pcol + 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = lo[r];
                v[4 + r] = hi[r];
            }
            if (m >= M || n >= N) continue;     // N % 8 == 0 on this path: the 8 columns are all in or all out
            if (slab) {
                *reinterpret_cast<f32x4*>(slab + (int64_t)m * N + n) = lo;
                *reinterpret_cast<f32x4*>(slab + (int64_t)m * N + n + 4) = hi;
                continue;
            }
            if (ep.act == UCFVIT_ACT_GELU) {
                if (ep.aux_out) {
                    Vec16<bf16> o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o.set(r, v[r]);
                    *reinterpret_cast<Vec16<bf16>*>(ep.aux_out + (int64_t)m * ep.ldaux + n) = o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = o.get(r);   // activation sees the stored (rounded) pre-activation
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = gelu_f(v[r]);
            } else if (ep.act == UCFVIT_ACT_GELU_GRAD) {
                const Vec16<bf16> h = *reinterpret_cast<const Vec16<bf16>*>(ep.aux_in + (int64_t)m * ep.ldaux + n);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] *= gelu_grad_f(h.get(r));
            }
            if (ep.residual) {
                const Vec16<bf16> rv = *reinterpret_cast<const Vec16<bf16>*>(ep.residual + (int64_t)m * ep.ldr + n);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += rv.get(r);
            }
            OutT* cp = C + (int64_t)m * ep.ldc + n;
            if constexpr (sizeof(OutT) == 2) {
                if (ep.accumulate) {
                    const Vec16<bf16> old = *reinterpret_cast<const Vec16<bf16>*>(cp);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += old.get(r);
                }
                Vec16<bf16> o;
#pragma unroll
                for (int r = 0; r < 8; ++r) o.set(r, v[r]);
                *reinterpret_cast<Vec16<bf16>*>(cp) = o;
            } else {
                f32x4 o0, o1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o0[r] = v[r];
                    o1[r] = v[4 + r];
                }
                if (ep.accumulate) {
                    o0 += *reinterpret_cast<const f32x4*>(cp);
                    o1 += *reinterpret_cast<const f32x4*>(cp + 4);
                }
                *reinterpret_cast<f32x4*>(cp) = o0;
                *reinterpret_cast<f32x4*>(cp + 4) = o1;
            }
        }
    }
}

// C[m][n] = alpha * sum_s slab[s][m][n] (+ C_old)   — fixed order, 16-B vectors
template <typename OutT>
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, OutT* __restrict__ C, int64_t M, int64_t N, int64_t ldc, int splits,
                                     float alpha, int accumulate) {
    const int64_t nv = N / 4;
    const int64_t total = M * nv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / nv, n = (i % nv) * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(slab + ((int64_t)k * M + m) * N + n);
        OutT* cp = C + m * ldc + n;
        Vec4<OutT> o;
        if (accumulate) {
            const Vec4<OutT> old = *reinterpret_cast<const Vec4<OutT>*>(cp);
#pragma unroll
            for (int r = 0; r < 4; ++r) o.set(r, alpha * s[r] + old.get(r));
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) o.set(r, alpha * s[r]);
        }
        *reinterpret_cast<Vec4<OutT>*>(cp) = o;
    }
}

struct Plan2 {
    int big;     // 1: 256x256 tile, 0: 128x128
    int splits;
    int k_per_split;
};

inline bool plan2(const ucfvit_gemm_desc* d, Plan2* p) {
    if (d->dtype != UCFVIT_BF16) return false;
    if (d->K < 128 || d->K % BK2 != 0) return false;
    if (d->M < 128 || d->N < 128) return false;
    const int64_t t256 = ((d->M + 255) / 256) * ((d->N + 255) / 256);
    const int64_t t128 = ((d->M + 127) / 128) * ((d->N + 127) / 128);
    const bool plain_epi = !d->bias && !d->residual && !d->aux_in && !d->aux_out && d->act == UCFVIT_ACT_NONE;
    p->splits = 1;
    if (t256 >= 192) {
        p->big = 1;
    } else {
        p->big = 0;
        if (plain_epi && t128 < 384) {
            int s = (int)((512 + t128 - 1) / t128);          // aim at ~512 workgroups (2 per CU)
            const int kmax = (int)(d->K / (8 * BK2));        // at least 8 K-tiles per slice
            if (s > kmax) s = kmax;
            if (s > 16) s = 16;
            if (s < 1) s = 1;
            p->splits = s;
        }
    }
    const int64_t ktiles = d->K / BK2;
    p->k_per_split = (int)(((ktiles + p->splits - 1) / p->splits) * BK2);
    p->splits = (int)((d->K + p->k_per_split - 1) / p->k_per_split);
    return true;
}

template <int LA, int LB, int BM, int BN, int WM, int WN, typename OutT>
int launch2(const ucfvit_gemm_desc* d, const Plan2& p, Epi2 ep, hipStream_t s) {
    const int tiles_m = (int)((d->M + BM - 1) / BM), tiles_n = (int)((d->N + BN - 1) / BN);
    constexpr size_t smem = 2 * (size_t)(BM + BN) * 128;
    auto kern = gemm2_kernel<LA, LB, BM, BN, WM, WN, OutT>;
    if (smem > 64 * 1024) {
        static bool done = false;  // per instantiation
        if (!done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) {
                ucfvit_set_error("ucfvit_gemm: cannot raise dynamic LDS to %zu bytes: %s", smem, hipGetErrorString(e));
                return UCFVIT_ERR_HIP;
            }
            done = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, p.splits), dim3(WM * WN * 64), smem, s, (const bf16*)d->A, (const bf16*)d->B, (OutT*)d->C,
                       (int)d->M, (int)d->N, (int)d->K, d->lda, d->ldb, ep, tiles_m, tiles_n, p.k_per_split);
    UCF_LAUNCH_CHECK("ucfvit_gemm(v2)");
    if (p.splits > 1) {
        const int64_t work = d->M * (d->N / 4);
        int64_t grid = (work + 255) / 256;
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL((splitk_reduce_kernel<OutT>), dim3((unsigned)grid), dim3(256), 0, s, (const float*)ep.slab, (OutT*)d->C, d->M, d->N,
                           d->ldc, p.splits, d->alpha, d->accumulate);
        UCF_LAUNCH_CHECK("ucfvit_gemm(v2 split-K reduce)");
    }
    return UCFVIT_OK;
}

template <int LA, int LB, typename OutT>
int dispatch_tile(const ucfvit_gemm_desc* d, const Plan2& p, const Epi2& ep, hipStream_t s) {
    if (p.big) return launch2<LA, LB, 256, 256, 2, 4, OutT>(d, p, ep, s);
    return launch2<LA, LB, 128, 128, 2, 2, OutT>(d, p, ep, s);
}

template <typename OutT>
int dispatch_layout2(const ucfvit_gemm_desc* d, const Plan2& p, const Epi2& ep, hipStream_t s) {
    const int la = d->a_layout, lb = d->b_layout;
    if (la == 0 && lb == 0) return dispatch_tile<0, 0, OutT>(d, p, ep, s);
    if (la == 0 && lb == 1) return dispatch_tile<0, 1, OutT>(d, p, ep, s);
    if (la == 1 && lb == 1) return dispatch_tile<1, 1, OutT>(d, p, ep, s);
    return dispatch_tile<1, 0, OutT>(d, p, ep, s);
}

}  // namespace

// bytes of fp32 workspace ucfvit_gemm wants for this problem (0 if none)
extern "C" int64_t ucfvit_gemm_workspace(const ucfvit_gemm_desc* d) {
    Plan2 p;
    if (!d || !plan2(d, &p) || p.splits <= 1) return 0;
    return (int64_t)p.splits * d->M * d->N * (int64_t)sizeof(float);
}

// returns 1 if the v2 kernel handled the problem, 0 if the caller should use the v1 path, <0 on error
int ucfvit_gemm_v2_try(const ucfvit_gemm_desc* d, hipStream_t s) {
    Plan2 p;
    if (!plan2(d, &p)) return 0;
    // alignment / shape requirements of the DMA + vector-epilogue path
    const int64_t a_contig = (d->a_layout == UCFVIT_LAYOUT_KC) ? d->K : d->M;
    const int64_t b_contig = (d->b_layout == UCFVIT_LAYOUT_KC) ? d->K : d->N;
    const size_t osz = d->out_dtype == UCFVIT_F32 ? 4 : 2;
    bool ok = ucf_is_aligned16(d->A) && ucf_is_aligned16(d->B) && d->lda % 8 == 0 && d->ldb % 8 == 0 && a_contig % 8 == 0 &&
              b_contig % 8 == 0 && d->N % 8 == 0 && d->ldc % 8 == 0 && ((uintptr_t)d->C) % 16 == 0 && d->M < (1ll << 31) &&
              d->N < (1ll << 31) && d->K < (1ll << 31);
    if (d->bias) ok = ok && ((uintptr_t)d->bias) % 8 == 0;
    if (d->residual) ok = ok && ((uintptr_t)d->residual) % 16 == 0 && d->ldr % 8 == 0;
    if (d->aux_in) ok = ok && ((uintptr_t)d->aux_in) % 16 == 0 && d->ldaux % 8 == 0;
    if (d->aux_out) ok = ok && ((uintptr_t)d->aux_out) % 16 == 0 && d->ldaux % 8 == 0;
    if (!ok) return 0;
    Epi2 ep;
    ep.bias = (const bf16*)d->bias;
    ep.residual = (const bf16*)d->residual;
    ep.aux_in = (const bf16*)d->aux_in;
    ep.aux_out = (bf16*)d->aux_out;
    ep.ldc = d->ldc;
    ep.ldr = d->ldr;
    ep.ldaux = d->ldaux;
    ep.act = d->act;
    ep.accumulate = d->accumulate;
    ep.alpha = d->alpha;
    ep.slab = nullptr;
    if (p.splits > 1) {
        const int64_t need = (int64_t)p.splits * d->M * d->N * (int64_t)sizeof(float);
        if (!d->workspace || d->workspace_bytes < need || !ucf_is_aligned16(d->workspace)) {
            // no (or too small a) workspace: run un-split
            p.splits = 1;
            p.k_per_split = (int)d->K;
        } else {
            ep.slab = (float*)d->workspace;
        }
    }
    int rc;
    if (d->out_dtype == UCFVIT_BF16)
        rc = dispatch_layout2<bf16>(d, p, ep, s);
    else if (d->out_dtype == UCFVIT_F32)
        rc = dispatch_layout2<float>(d, p, ep, s);
    else
        return 0;
    return rc == UCFVIT_OK ? 1 : rc;
}

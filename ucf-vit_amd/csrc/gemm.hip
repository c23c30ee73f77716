// GEMM with fused epilogue for gfx950: C[M,N] = epi(alpha * op(A)·op(B)).
//
// v1 structure ("register-staged, one LDS image per operand"):
//   128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA 16x16 fragments),
//   K-tile = 128 bytes of the contraction index (64 bf16 / 32 fp32).  Global->VGPR loads of K-tile t+1 are issued
//   before the MFMA phase of tile t and written to LDS after the barrier (issue-early / write-late).
//   MFMA: v_mfma_f32_16x16x32_bf16 (bf16) or v_mfma_f32_16x16x4_f32 (exact fp32).
//   Operands are staged in their natural memory layout with 16-byte coalesced loads:
//     KC (contraction contiguous): LDS image [128 rows][128 B], 16-B slots XOR-swizzled by (row&7); fragments by ds_read_b128
//     KS (contraction strided)   : LDS image [k][128 rows], 32-B groups XOR-swizzled; fragments by ds_read_b64_tr_b16
//                                  (bf16, hardware transpose) or 4 ds_read_b32 (fp32)
//   The MFMA is issued with the operand roles swapped (D = Btile·Atileᵀ) so each lane ends up holding 4 CONSECUTIVE
//   output columns of one output row: bias/residual/aux are read and C is written with 8/16-byte vectors.
//
// Requirements of the MFMA path (checked on the host; anything else goes to the scalar fallback kernel):
//   16-byte aligned base pointers, leading dimensions and the contiguous extent of each operand multiples of 16 bytes.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int TILE_BYTES = 16384;  // one operand K-tile: 128 rows x 128 B (KC) or BK k-rows x (128 elems) (KS)

template <typename T> struct MmaT;
template <> struct MmaT<bf16> {
    static constexpr int BK = 64;      // contraction elements per K-tile
    static constexpr int EPV = 8;      // elements per 16 bytes
    static constexpr int KCHUNK = 32;  // contraction elements consumed per fragment pair
    typedef bf16x8 frag_t;
    static __device__ __forceinline__ void mma(f32x4& acc, const frag_t& a, const frag_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
};
template <> struct MmaT<float> {
    static constexpr int BK = 32;
    static constexpr int EPV = 4;
    static constexpr int KCHUNK = 16;
    typedef f32x4 frag_t;
    static __device__ __forceinline__ void mma(f32x4& acc, const frag_t& a, const frag_t& b) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
    }
};

// ---- LDS image addressing -------------------------------------------------------------------------
// KC image: byte offset of 16-B slot `slot` (0..7) of row `row` (0..127)
__device__ __forceinline__ int kc_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

// KS image, bf16: k-row = 128 elems = 256 B; 32-B groups swizzled so the 8 k-rows a half-wave's transposed read
// touches land on 8 distinct 32-B bank groups.
__device__ __forceinline__ int ks_swz_bf16(int krow) { return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 5; }
__device__ __forceinline__ int ks_off_bf16(int krow, int colbyte) { return krow * 256 + (colbyte ^ ks_swz_bf16(krow)); }
// KS image, fp32: k-row = 128 elems = 512 B
__device__ __forceinline__ int ks_off_f32(int krow, int colbyte) { return krow * 512 + (colbyte ^ (((krow >> 2) & 1) << 6)); }

template <typename T> struct Stage {  // 4 x 16 B per thread per operand K-tile
    u32x4 v[4];
};

// Global -> registers for one operand K-tile.  R = extent of the operand's row/col index, K = contraction extent.
template <typename T, int LAYOUT>
__device__ __forceinline__ void stage_load(Stage<T>& st, const T* __restrict__ base, int64_t ld, int r0, int k0, int R, int K,
                                           int tid) {
    constexpr int EPV = MmaT<T>::EPV;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + i * NTHREADS;
        int64_t off;
        bool ok;
        if (LAYOUT == UCFVIT_LAYOUT_KC) {
            const int row = p >> 3, slot = p & 7;
            const int k = k0 + slot * EPV;
            ok = (r0 + row < R) && (k < K);
            off = (int64_t)(r0 + row) * ld + k;
        } else {
            constexpr int PPR = 128 / EPV;  // 16-B pieces per k-row
            const int krow = p / PPR, piece = p % PPR;
            const int r = r0 + piece * EPV;
            ok = (k0 + krow < K) && (r < R);
            off = (int64_t)(k0 + krow) * ld + r;
        }
        u32x4 z = {0u, 0u, 0u, 0u};
        st.v[i] = ok ? *reinterpret_cast<const u32x4*>(base + off) : z;
    }
}

template <typename T, int LAYOUT>
__device__ __forceinline__ void stage_store(const Stage<T>& st, char* lds, int tid) {
    constexpr int EPV = MmaT<T>::EPV;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + i * NTHREADS;
        int off;
        if (LAYOUT == UCFVIT_LAYOUT_KC) {
            off = kc_off(p >> 3, p & 7);
        } else {
            constexpr int PPR = 128 / EPV;
            const int krow = p / PPR, piece = p % PPR;
            off = (sizeof(T) == 2) ? ks_off_bf16(krow, piece * 16) : ks_off_f32(krow, piece * 16);
        }
        *reinterpret_cast<u32x4*>(lds + off) = st.v[i];
    }
}

// Fragment of the 16 rows [rbase, rbase+16) for k-chunk `c` of the K-tile.
template <typename T, int LAYOUT>
__device__ __forceinline__ typename MmaT<T>::frag_t load_frag(const char* lds, int rbase, int c, int lane) {
    typedef typename MmaT<T>::frag_t frag_t;
    const int g = lane >> 4, i = lane & 15;
    if constexpr (LAYOUT == UCFVIT_LAYOUT_KC) {
        // lane holds 16 B = k-elements [KCHUNK*c + EPV*g, +EPV) of row rbase+i (both dtypes: chunk = 64 B = 4 slots)
        const int row = rbase + i;
        return *reinterpret_cast<const frag_t*>(lds + kc_off(row, 4 * c + g));
    } else if constexpr (sizeof(T) == 2) {
        // two hardware-transposed reads: k-rows kb+{0..3} and kb+4+{0..3}, columns rbase..rbase+15;
        // lane 4q+p of each 16-lane group supplies the address of k-row q, columns 4p..4p+3.
        const int kb = 32 * c + 8 * g, q = i >> 2, p = i & 3;
        const int colbyte = (rbase + 4 * p) * 2;
        short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, lds + ks_off_bf16(kb + q, colbyte)));
        short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, lds + ks_off_bf16(kb + 4 + q, colbyte)));
        short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(frag_t, r);
    } else {
        const int kb = 16 * c + 4 * g;
        const int colbyte = (rbase + i) * 4;
        f32x4 r;
#pragma unroll
        for (int s = 0; s < 4; ++s) r[s] = *reinterpret_cast<const float*>(lds + ks_off_f32(kb + s, colbyte));
        return __builtin_bit_cast(frag_t, r);
    }
}

struct EpiArgs {
    const void* bias;
    const void* residual;
    const void* aux_in;
    void* aux_out;
    int64_t ldc, ldr, ldaux;
    int act, accumulate;
    float alpha;
};

template <typename T, typename OutT, int LA, int LB>
__global__ __launch_bounds__(NTHREADS) void gemm_mfma_kernel(const T* __restrict__ A, const T* __restrict__ B,
                                                              OutT* __restrict__ C, int M, int N, int K, int64_t lda,
                                                              int64_t ldb, EpiArgs ep, int tiles_m, int tiles_n) {
    typedef MmaT<T> MM;
    typedef typename MM::frag_t frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsA = smem;
    char* ldsB = smem + TILE_BYTES;

    // XCD-aware tile order: each XCD label gets a contiguous run of tiles; inside a run, tiles walk down M within a
    // band of 8 N-tiles so neighbouring workgroups share the B panel and successive ones reuse the A panel from L2.
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

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (K + MM::BK - 1) / MM::BK;
    Stage<T> sa, sb;
    stage_load<T, LA>(sa, A, lda, m0, 0, M, K, tid);
    stage_load<T, LB>(sb, B, ldb, n0, 0, N, K, tid);

    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // everyone finished reading the previous tile's LDS image
        stage_store<T, LA>(sa, ldsA, tid);
        stage_store<T, LB>(sb, ldsB, tid);
        __syncthreads();
        if (kt + 1 < nk) {  // next tile's loads fly under this tile's MFMAs
            stage_load<T, LA>(sa, A, lda, m0, (kt + 1) * MM::BK, M, K, tid);
            stage_load<T, LB>(sb, B, ldb, n0, (kt + 1) * MM::BK, N, K, tid);
        }
#pragma unroll
        for (int c = 0; c < MM::BK / MM::KCHUNK; ++c) {
            frag_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = load_frag<T, LA>(ldsA, wm + 16 * i, c, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = load_frag<T, LB>(ldsB, wn + 16 * j, c, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) MM::mma(acc[i][j], fb[j], fa[i]);  // roles swapped: D[n][m]
        }
    }

    // Epilogue.  acc[i][j][r] = C[m0+wm+16i+(lane&15)][n0+wn+16j+4*(lane>>4)+r]
    const int g = lane >> 4, li = lane & 15;
    const T* bias = (const T*)ep.bias;
    const T* res = (const T*)ep.residual;
    const T* auxi = (const T*)ep.aux_in;
    T* auxo = (T*)ep.aux_out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm + 16 * i + li;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn + 16 * j + 4 * g;
            if (n >= N) continue;  // N % 4 == 0 on this path, so the 4 columns are all in range
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = ep.alpha * acc[i][j][r];
            if (bias) {
                Vec4<T> b = *reinterpret_cast<const Vec4<T>*>(bias + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += b.get(r);
            }
            if (ep.act == UCFVIT_ACT_GELU) {
                if (auxo) {
                    Vec4<T> o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o.set(r, v[r]);
                    *reinterpret_cast<Vec4<T>*>(auxo + (int64_t)m * ep.ldaux + n) = o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = o.get(r);  // activation sees the stored (rounded) pre-activation
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
            } else if (ep.act == UCFVIT_ACT_GELU_GRAD) {
                Vec4<T> h = *reinterpret_cast<const Vec4<T>*>(auxi + (int64_t)m * ep.ldaux + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= gelu_grad_f(h.get(r));
            } else if (ep.act == UCFVIT_ACT_GELU_SAVE_DERIV) {
                Vec4<T> o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o.set(r, gelu_grad_f(v[r]));
                    v[r] = gelu_f(v[r]);
                }
                *reinterpret_cast<Vec4<T>*>(auxo + (int64_t)m * ep.ldaux + n) = o;
            } else if (ep.act == UCFVIT_ACT_MUL_AUX) {
                Vec4<T> h = *reinterpret_cast<const Vec4<T>*>(auxi + (int64_t)m * ep.ldaux + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= h.get(r);
            }
            if (res) {
                Vec4<T> rr = *reinterpret_cast<const Vec4<T>*>(res + (int64_t)m * ep.ldr + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rr.get(r);
            }
            OutT* cp = C + (int64_t)m * ep.ldc + n;
            if (ep.accumulate) {
                Vec4<OutT> old = *reinterpret_cast<const Vec4<OutT>*>(cp);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += old.get(r);
            }
            Vec4<OutT> o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o.set(r, v[r]);
            *reinterpret_cast<Vec4<OutT>*>(cp) = o;
        }
    }
}

// Scalar fallback for tiny / unaligned shapes (e.g. a 2-class head): one thread per output element.
template <typename T, typename OutT>
__global__ void gemm_scalar_kernel(const T* __restrict__ A, const T* __restrict__ B, OutT* __restrict__ C, int64_t M,
                                   int64_t N, int64_t K, int64_t lda, int64_t ldb, int la, int lb, EpiArgs ep) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * N) return;
    const int64_t m = idx / N, n = idx % N;
    float acc = 0.f;
    for (int64_t k = 0; k < K; ++k) {
        const float a = to_f32<T>(la == UCFVIT_LAYOUT_KC ? A[m * lda + k] : A[k * lda + m]);
        const float b = to_f32<T>(lb == UCFVIT_LAYOUT_KC ? B[n * ldb + k] : B[k * ldb + n]);
        acc = fmaf(a, b, acc);
    }
    float v = ep.alpha * acc;
    if (ep.bias) v += to_f32<T>(((const T*)ep.bias)[n]);
    if (ep.act == UCFVIT_ACT_GELU) {
        if (ep.aux_out) {
            T h = from_f32<T>(v);
            ((T*)ep.aux_out)[m * ep.ldaux + n] = h;
            v = to_f32<T>(h);
        }
        v = gelu_f(v);
    } else if (ep.act == UCFVIT_ACT_GELU_GRAD) {
        v *= gelu_grad_f(to_f32<T>(((const T*)ep.aux_in)[m * ep.ldaux + n]));
    } else if (ep.act == UCFVIT_ACT_GELU_SAVE_DERIV) {
        ((T*)ep.aux_out)[m * ep.ldaux + n] = from_f32<T>(gelu_grad_f(v));
        v = gelu_f(v);
    } else if (ep.act == UCFVIT_ACT_MUL_AUX) {
        v *= to_f32<T>(((const T*)ep.aux_in)[m * ep.ldaux + n]);
    }
    if (ep.residual) v += to_f32<T>(((const T*)ep.residual)[m * ep.ldr + n]);
    OutT* cp = C + m * ep.ldc + n;
    if (ep.accumulate) v += to_f32<OutT>(*cp);
    *cp = from_f32<OutT>(v);
}

template <typename T, typename OutT, int LA, int LB>
int launch_mfma(const ucfvit_gemm_desc* d, const EpiArgs& ep, hipStream_t s) {
    const int tiles_m = (int)((d->M + BM - 1) / BM), tiles_n = (int)((d->N + BN - 1) / BN);
    const int nwg = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_mfma_kernel<T, OutT, LA, LB>), dim3(nwg), dim3(NTHREADS), 2 * TILE_BYTES, s, (const T*)d->A,
                       (const T*)d->B, (OutT*)d->C, (int)d->M, (int)d->N, (int)d->K, d->lda, d->ldb, ep, tiles_m, tiles_n);
    UCF_LAUNCH_CHECK("ucfvit_gemm(mfma)");
    return UCFVIT_OK;
}

template <typename T, typename OutT>
int dispatch_layout(const ucfvit_gemm_desc* d, const EpiArgs& ep, hipStream_t s) {
    const int la = d->a_layout, lb = d->b_layout;
    constexpr int64_t EPV = 16 / sizeof(T);
    constexpr int64_t OPV = 4;  // output vector = 4 elements
    const int64_t a_contig = (la == UCFVIT_LAYOUT_KC) ? d->K : d->M;
    const int64_t b_contig = (lb == UCFVIT_LAYOUT_KC) ? d->K : d->N;
    bool ok = ucf_is_aligned16(d->A) && ucf_is_aligned16(d->B) && (d->lda % EPV == 0) && (d->ldb % EPV == 0) &&
              (a_contig % EPV == 0) && (b_contig % EPV == 0) && (d->N % OPV == 0) && (d->ldc % OPV == 0) &&
              (((uintptr_t)d->C) % (OPV * sizeof(OutT)) == 0) && d->M < (1ll << 31) && d->N < (1ll << 31) && d->K < (1ll << 31);
    const size_t t4 = 4 * sizeof(T);
    if (d->bias) ok = ok && (((uintptr_t)d->bias) % t4 == 0);
    if (d->residual) ok = ok && (((uintptr_t)d->residual) % t4 == 0) && (d->ldr % 4 == 0);
    if (d->aux_in) ok = ok && (((uintptr_t)d->aux_in) % t4 == 0) && (d->ldaux % 4 == 0);
    if (d->aux_out) ok = ok && (((uintptr_t)d->aux_out) % t4 == 0) && (d->ldaux % 4 == 0);
    if (ok && d->M * d->N >= 256) {
        if (la == UCFVIT_LAYOUT_KC && lb == UCFVIT_LAYOUT_KC) return launch_mfma<T, OutT, 0, 0>(d, ep, s);
        if (la == UCFVIT_LAYOUT_KC && lb == UCFVIT_LAYOUT_KS) return launch_mfma<T, OutT, 0, 1>(d, ep, s);
        if (la == UCFVIT_LAYOUT_KS && lb == UCFVIT_LAYOUT_KS) return launch_mfma<T, OutT, 1, 1>(d, ep, s);
        if (la == UCFVIT_LAYOUT_KS && lb == UCFVIT_LAYOUT_KC) return launch_mfma<T, OutT, 1, 0>(d, ep, s);
    }
    const int64_t total = d->M * d->N;
    const int64_t blocks = (total + 255) / 256;
    UCF_CHECK_ARG(blocks < (1ll << 31), "ucfvit_gemm: scalar fallback grid too large (M=%lld N=%lld)", (long long)d->M,
                  (long long)d->N);
    hipLaunchKernelGGL((gemm_scalar_kernel<T, OutT>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)d->A, (const T*)d->B,
                       (OutT*)d->C, d->M, d->N, d->K, d->lda, d->ldb, la, lb, ep);
    UCF_LAUNCH_CHECK("ucfvit_gemm(scalar)");
    return UCFVIT_OK;
}

}  // namespace

int ucfvit_gemm_v2_try(const ucfvit_gemm_desc* d, hipStream_t s);  // gemm2.hip


extern "C" int ucfvit_gemm(const ucfvit_gemm_desc* d, void* stream) {
    UCF_CHECK_ARG(d != nullptr, "ucfvit_gemm: null descriptor");
    if (d->M == 0 || d->N == 0) return UCFVIT_OK;      // empty batch / empty output: nothing to write (pointers may be NULL)
    UCF_CHECK_ARG(d->A && d->B && d->C, "ucfvit_gemm: null operand pointer");
    UCF_CHECK_ARG(d->M >= 0 && d->N >= 0 && d->K >= 0, "ucfvit_gemm: negative size");
    UCF_CHECK_ARG(d->a_layout == 0 || d->a_layout == 1, "ucfvit_gemm: bad a_layout %d", d->a_layout);
    UCF_CHECK_ARG(d->b_layout == 0 || d->b_layout == 1, "ucfvit_gemm: bad b_layout %d", d->b_layout);
    UCF_CHECK_ARG(d->act >= 0 && d->act <= 4, "ucfvit_gemm: bad act %d", d->act);
    UCF_CHECK_ARG((d->act != UCFVIT_ACT_GELU_GRAD && d->act != UCFVIT_ACT_MUL_AUX) || d->aux_in, "ucfvit_gemm: act %d needs aux_in", d->act);
    UCF_CHECK_ARG(d->act != UCFVIT_ACT_GELU_SAVE_DERIV || d->aux_out, "ucfvit_gemm: ACT_GELU_SAVE_DERIV needs aux_out");
    UCF_CHECK_ARG(d->lda >= ((d->a_layout == 0) ? d->K : d->M), "ucfvit_gemm: lda too small");
    UCF_CHECK_ARG(d->ldb >= ((d->b_layout == 0) ? d->K : d->N), "ucfvit_gemm: ldb too small");
    UCF_CHECK_ARG(d->ldc >= d->N, "ucfvit_gemm: ldc too small");
    if (d->M == 0 || d->N == 0) return UCFVIT_OK;
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == UCFVIT_BF16) {
        const int rc = ucfvit_gemm_v2_try(d, s);   // large-tile DMA-pipelined kernel when the shape qualifies
        if (rc == 1) return UCFVIT_OK;
        if (rc < 0) return rc;
    }
    if (d->c_colsum_partial) {
        ucfvit_set_error("ucfvit_gemm: c_colsum_partial is not available for this problem (ask ucfvit_gemm_colsum_rows first)");
        return UCFVIT_ERR_UNSUPPORTED;
    }
    EpiArgs ep;
    ep.bias = d->bias;
    ep.residual = d->residual;
    ep.aux_in = d->aux_in;
    ep.aux_out = d->aux_out;
    ep.ldc = d->ldc;
    ep.ldr = d->ldr;
    ep.ldaux = d->ldaux;
    ep.act = d->act;
    ep.accumulate = d->accumulate;
    ep.alpha = d->alpha;
    if (d->dtype == UCFVIT_F32) {
        UCF_CHECK_ARG(d->out_dtype == UCFVIT_F32, "ucfvit_gemm: f32 inputs require f32 output");
        return dispatch_layout<float, float>(d, ep, s);
    } else if (d->dtype == UCFVIT_BF16) {
        if (d->out_dtype == UCFVIT_BF16) return dispatch_layout<bf16, bf16>(d, ep, s);
        if (d->out_dtype == UCFVIT_F32) return dispatch_layout<bf16, float>(d, ep, s);
    }
    ucfvit_set_error("ucfvit_gemm: unsupported dtype combination (%d -> %d)", d->dtype, d->out_dtype);
    return UCFVIT_ERR_UNSUPPORTED;
}

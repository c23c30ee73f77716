// Short-sequence fused attention for gfx950 (N <= 256 tokens: ViT 224^2/16 has N = 197, MAE encoder 49, MAE decoder 196).
//
// One workgroup per (batch, head): the whole K and V (forward, backward phase A) or Q and dO (backward phase B) of that head live
// in LDS — they are read from HBM/L2 once instead of once per 64-query tile — and the softmax is single
// pass (all scores of a query row are in registers at once: no running max, no rescaling of the output accumulator).
// Work is cut in 16-row MFMA blocks, so N = 197 costs 13 x 13 blocks (208^2) instead of the 4 x 4 tiles of 64 (256^2) of the
// streaming kernel.  Same "softmax index on the lane" layout as attention.hip: Sᵀ[key][q] = K·Qᵀ, accumulators feed the
// next MFMA as operands, Vᵀ / Kᵀ / Qᵀ / dOᵀ come out of the row-major LDS images through ds_read_b64_tr_b16.
// NB = number of 16-row blocks (compile time); rows >= N are zero-filled and masked.
#include "common.h"

namespace {

constexpr int AS_THREADS = 256;

template <typename T> struct MmaS;
template <> struct MmaS<bf16> {
    typedef bf16x8 frag_t;
    static __device__ __forceinline__ void mma(f32x4& acc, const frag_t& a, const frag_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
};
template <> struct MmaS<float> {
    typedef f32x4 frag_t;
    static __device__ __forceinline__ void mma(f32x4& acc, const frag_t& a, const frag_t& b) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
    }
};

template <typename T, int DH, int NB> struct AS {
    static constexpr int EPV = 16 / sizeof(T);
    static constexpr int RB = DH * sizeof(T);       // image row bytes
    static constexpr int SPR = RB / 16;             // 16-B slots per row
    static constexpr int NCH = DH / (4 * EPV);      // head-dim chunks (contraction over d)
    static constexpr int NDB = DH / 16;             // 16-wide head-dim blocks (outputs)
    static constexpr int BPC = (4 * EPV) / 16;      // 16-row blocks per row chunk (bf16: 2, fp32: 1)
    static constexpr int NRC = (NB + BPC - 1) / BPC;  // row chunks (contraction over keys / queries)
    static constexpr int ROWS = NRC * BPC * 16;     // rows held in an LDS image (padded to whole chunks)
    static constexpr int IMG = ROWS * RB;
    typedef typename MmaS<T>::frag_t frag_t;
};

template <int SPR> __device__ __forceinline__ int swz_s(int row) {
    if (SPR == 4) return (0x1230 >> (((row >> 2) & 3) * 4)) & 3;
    if (SPR == 8) return row & 7;
    return row & 15;
}
template <typename T, int DH> __device__ __forceinline__ int img_off(int row, int slot) {
    constexpr int RB = DH * sizeof(T), SPR = RB / 16;
    return row * RB + ((slot ^ swz_s<SPR>(row)) << 4);
}

// global rows [0, R) of a [.., row_stride] matrix -> LDS image of ROWS rows (rows >= R zero)
template <typename T, int DH, int ROWS>
__device__ __forceinline__ void load_image(char* lds, const T* __restrict__ base, int64_t row_stride, int R, int tid) {
    constexpr int EPV = 16 / sizeof(T), SPR = DH * sizeof(T) / 16;
    constexpr int NIT = (ROWS * SPR + AS_THREADS - 1) / AS_THREADS;
    u32x4 v[NIT];
    // all loads in flight first (one HBM/L2 round trip for the whole image), then the LDS stores
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int p = tid + i * AS_THREADS;
        const int row = p / SPR, slot = p % SPR;
        v[i] = u32x4{0u, 0u, 0u, 0u};
        if (p < ROWS * SPR && row < R) v[i] = *reinterpret_cast<const u32x4*>(base + (int64_t)row * row_stride + slot * EPV);
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int p = tid + i * AS_THREADS;
        if (p < ROWS * SPR) *reinterpret_cast<u32x4*>(lds + img_off<T, DH>(p / SPR, p % SPR)) = v[i];
    }
}

template <typename T, int DH>
__device__ __forceinline__ typename MmaS<T>::frag_t s_frag_row(const char* lds, int rb, int c, int lane) {
    const int row = rb * 16 + (lane & 15), g = lane >> 4;
    return *reinterpret_cast<const typename MmaS<T>::frag_t*>(lds + img_off<T, DH>(row, 4 * c + g));
}
template <typename T, int DH>
__device__ __forceinline__ typename MmaS<T>::frag_t s_frag_tr(const char* lds, int rc, int db, int lane) {
    typedef typename MmaS<T>::frag_t frag_t;
    const int g = lane >> 4, i = lane & 15;
    if constexpr (sizeof(T) == 2) {
        const int q = i >> 2, p = i & 3;
        const int slot = 2 * db + (p >> 1), sub = (p & 1) * 8;
        const int r_lo = 32 * rc + 4 * g + q;
        const char* a_lo = lds + img_off<T, DH>(r_lo, slot) + sub;
        const char* a_hi = lds + img_off<T, DH>(r_lo + 16, slot) + sub;
        short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, a_lo));
        short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, a_hi));
        short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(frag_t, r);
    } else {
        const int slot = 4 * db + (i >> 2), sub = (i & 3) * 4;
        f32x4 r;
#pragma unroll
        for (int s = 0; s < 4; ++s) r[s] = *reinterpret_cast<const float*>(lds + img_off<T, DH>(16 * rc + 4 * g + s, slot) + sub);
        return __builtin_bit_cast(frag_t, r);
    }
}
// accumulator blocks -> operand fragment of row chunk rc; blocks >= nb contribute zeros
template <typename T, int NBLK> __device__ __forceinline__ typename MmaS<T>::frag_t s_frag_acc(const f32x4 (&acc)[NBLK], int rc) {
    typedef typename MmaS<T>::frag_t frag_t;
    if constexpr (sizeof(T) == 2) {
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[j] = (bf16)acc[2 * rc][j];
            r[4 + j] = (2 * rc + 1 < NBLK) ? (bf16)acc[(2 * rc + 1 < NBLK) ? 2 * rc + 1 : 0][j] : (bf16)0.f;
        }
        return __builtin_bit_cast(frag_t, r);
    } else {
        return __builtin_bit_cast(frag_t, acc[rc]);
    }
}
template <typename T, int DH>
__device__ __forceinline__ typename MmaS<T>::frag_t s_frag_global(const T* __restrict__ base, int64_t row_stride, int row, int R, int c, int lane) {
    typedef typename MmaS<T>::frag_t frag_t;
    const int g = lane >> 4;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < R) v = *reinterpret_cast<const u32x4*>(base + (int64_t)row * row_stride + (4 * c + g) * (16 / (int)sizeof(T)));
    return __builtin_bit_cast(frag_t, v);
}
__device__ __forceinline__ float gmax(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float gsum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// =====================================================================================================================
// forward: wave handles 16-query blocks qb = wave, wave+4, ...
// =====================================================================================================================
template <typename T, int DH, int NB, bool EXACT>
__global__ __launch_bounds__(AS_THREADS) void attn_s_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse, int N,
                                                                 int H, float scale_log2e) {
    typedef AS<T, DH, NB> A;
    typedef typename A::frag_t frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsK = smem;
    char* ldsV = smem + A::IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int h = blockIdx.x % H;
    const int64_t b = blockIdx.x / H;
    const int D = H * DH;
    const int64_t rs = 3 * (int64_t)D;
    const T* qbase = qkv + b * N * rs + h * DH;
    load_image<T, DH, A::ROWS>(ldsK, qbase + D, rs, N, tid);
    load_image<T, DH, A::ROWS>(ldsV, qbase + 2 * D, rs, N, tid);
    __syncthreads();
    const int nqb = (N + 15) / 16;
    for (int qb = wave; qb < nqb; qb += 4) {
        const int q = qb * 16 + li;
        frag_t qf[A::NCH];
#pragma unroll
        for (int c = 0; c < A::NCH; ++c) qf[c] = s_frag_global<T, DH>(qbase, rs, q, N, c, lane);
        f32x4 s[NB];
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            s[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < A::NCH; ++c) MmaS<T>::mma(s[kb], s_frag_row<T, DH>(ldsK, kb, c, lane), qf[c]);
            if (EXACT ? (kb == NB - 1) : (kb * 16 + 16 > N)) {   // only the ragged / padding key blocks pay for masking
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kb][r] = (kb * 16 + 4 * g + r < N) ? s[kb][r] : -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][r]);
        }
        // softmax in the log2 domain with the scale folded into one fma per score: p = 2^(s*c - max*c)   (c > 0)
        const float m = gmax(mx) * scale_log2e;
        float l = 0.f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], scale_log2e, -m));
                s[kb][r] = p;
                l += p;
            }
        const float lt = gsum(l);
        f32x4 o[A::NDB];
#pragma unroll
        for (int d = 0; d < A::NDB; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rc = 0; rc < A::NRC; ++rc) {
            const frag_t pf = s_frag_acc<T, NB>(s, rc);
#pragma unroll
            for (int d = 0; d < A::NDB; ++d) MmaS<T>::mma(o[d], s_frag_tr<T, DH>(ldsV, rc, d, lane), pf);
        }
        if (q < N) {
            const float inv = 1.f / lt;
            T* op = out + (b * N + q) * D + h * DH;
#pragma unroll
            for (int d = 0; d < A::NDB; ++d) {
                Vec4<T> v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v.set(r, o[d][r] * inv);
                *reinterpret_cast<Vec4<T>*>(op + d * 16 + 4 * g) = v;
            }
            if (g == 0) lse[(b * H + h) * N + q] = m + log2f(lt);
        }
    }
}

template <typename K> int big_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return UCFVIT_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        ucfvit_set_error("attention(short): cannot raise dynamic LDS to %zu bytes: %s", bytes, hipGetErrorString(e));
        return UCFVIT_ERR_HIP;
    }
    return UCFVIT_OK;
}

template <typename T, int DH, int NB>
int launch_fwd(const void* qkv, void* out, float* lse, int64_t B, int64_t N, int64_t H, float scale, hipStream_t s) {
    constexpr size_t smem = 2 * AS<T, DH, NB>::IMG;
    const float sl2 = scale * 1.44269504088896340736f;
    const dim3 grid((unsigned)(B * H)), block(AS_THREADS);
    if ((N + 15) / 16 == NB) {
        if (int rc = big_lds(attn_s_fwd_kernel<T, DH, NB, true>, smem)) return rc;
        hipLaunchKernelGGL((attn_s_fwd_kernel<T, DH, NB, true>), grid, block, smem, s, (const T*)qkv, (T*)out, lse, (int)N, (int)H, sl2);
    } else {
        if (int rc = big_lds(attn_s_fwd_kernel<T, DH, NB, false>, smem)) return rc;
        hipLaunchKernelGGL((attn_s_fwd_kernel<T, DH, NB, false>), grid, block, smem, s, (const T*)qkv, (T*)out, lse, (int)N, (int)H, sl2);
    }
    UCF_LAUNCH_CHECK("ucfvit_attention_fwd(short)");
    return UCFVIT_OK;
}
// =====================================================================================================================
// backward, fused (bf16, head dim 64 / 32, N <= 256): ONE launch per attention layer instead of delta + dQ + dK/dV.
// Phase A: each wave takes query blocks and produces dQ with K and V resident in LDS; phase B: each wave takes key blocks and
// produces dK, dV with Q and dO resident — the two formulations of the two-kernel backward, without the 64-row tile padding
// (13 x 13 blocks of 16 instead of 16 x 16 for N = 197) and without the re-reads of Q / dO per key tile.
//
// A wave works on TWO 16-row blocks at a time where the sequence has more than 8 blocks (every K / V fragment read from LDS in phase A,
// every Q / dO fragment in phase B feeds two MFMAs: 1372 KiB of LDS reads per head at N = 197 instead of 2444), and the score tiles are
// made and consumed per 32-row chunk instead of being kept for a whole row — no register array grows with the sequence, so N <= 256 fits
// (16 blocks; the first form stopped at 13).  For that delta = rowsum(dO o O) comes from the forward output (read once in phase A, handed
// to phase B through LDS) instead of from a whole row of P and dP.
// What bounds the kernel is not its arithmetic (measured at ViT-L B = 665, 16 heads, N = 197: with 6/7 of the chunk loop removed it still
// took 711 of 888 us) but the workgroup's chain of global-memory round trips, see AgOpsA below and profiles/r03_f_attention_bwd.txt.
// Every LDS address is one of six per-lane bases plus an immediate (row images: base ^ 64 c + 2048 block; transposed reads:
// base ^ 32 d + 4096 chunk); sched_barriers keep hipcc from hoisting all fragment reads of a phase.
// =====================================================================================================================
template <int RB> __device__ __forceinline__ bf16x8 af_row(const char* img, int base_c, int blk) {   // rows 16 blk .. + 15, k-chunk c
    return *reinterpret_cast<const bf16x8*>(img + base_c + blk * (16 * RB));
}
template <int RB> __device__ __forceinline__ bf16x8 af_tr(const char* img, int base_d, int rc) {     // rows 32 rc .. + 31 transposed, head-dim block d
    const char* a = img + base_d + rc * (32 * RB);
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, a));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, a + 16 * RB));
    short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}
// per-lane address bases of the row images (head dim 64: 128-byte rows, 8 slots, swizzle row & 7; head dim 32: 64-byte rows, 4 slots,
// swizzle by row group, see swz_s): the fragment of k-chunk c of row block blk is at rowb[c] + blk * 16 RB, the transposed fragment of
// head-dim block d of row chunk rc at (trb ^ 32 d) + rc * 32 RB
template <int DH> struct AfBases {
    int rowb[DH / 32], trb;
    __device__ __forceinline__ AfBases(int lane) {
        constexpr int RB = DH * 2, SPR = RB / 16;
        const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
        const int sw_row = swz_s<SPR>(li);                 // blocks start at multiples of 16 rows: the swizzle sees li only
#pragma unroll
        for (int c = 0; c < DH / 32; ++c) rowb[c] = li * RB + (((4 * c + g) ^ sw_row) << 4);
        const int sw_tr = swz_s<SPR>(4 * g + tq);          // chunks start at multiples of 32 rows
        trb = (4 * g + tq) * RB + ((((tp >> 1) ^ sw_tr)) << 4) + (tp & 1) * 8;
    }
};

// Only TWO images are resident at a time (K, V for phase A, then Q, dO for phase B), so two workgroups of 4 waves fit a CU
// (58 KiB of LDS each) and one workgroup's loads overlap the other's MFMA/softmax work (all four images in one 8-wave workgroup
// measured 273 us per layer at ViT-L B=166, this form 222 us, the streaming pair + delta kernel 413 us); the Q / dO images of
// phase B are fetched into registers BEFORE phase A starts and stored to LDS after it.  The blocks a wave owns in a phase come
// straight from global memory as MFMA fragments (the lines were just read for the images: L2 hits).  delta = rowsum(dO o O) is
// taken in phase A as sum_k P dP from values the wave holds anyway, so no delta kernel and no read of O.
constexpr int AG_THREADS = 256, AG_WAVES = 4;

template <int ROWS, int DH> struct AgStage {
    static constexpr int SPR = DH / 8;       // 16-byte slots per row
    static constexpr int NIT = (ROWS * SPR + AG_THREADS - 1) / AG_THREADS;
    u32x4 v[NIT];
};
template <int ROWS, int DH>
__device__ __forceinline__ void ag_fetch(AgStage<ROWS, DH>& st, const bf16* __restrict__ base, int64_t row_stride, int R, int tid) {
    constexpr int SPR = DH / 8;
#pragma unroll
    for (int i = 0; i < AgStage<ROWS, DH>::NIT; ++i) {
        const int p = tid + i * AG_THREADS;
        const int row = p / SPR, slot = p % SPR;
        st.v[i] = u32x4{0u, 0u, 0u, 0u};
        if (p < ROWS * SPR && row < R) st.v[i] = *reinterpret_cast<const u32x4*>(base + (int64_t)row * row_stride + slot * 8);
    }
}
// All global accesses of the fused backward are RAW BUFFER loads / stores whose resource covers rows [0, N) of one (batch, head): rows >= N
// are out of range for the hardware (loads return 0, stores are dropped), so there is no branch around any of them and hipcc's
// s_waitcnt bookkeeping stays exact — a load requested a pass ahead is waited for with vmcnt(number of younger stores), not vmcnt(0).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ag_rsrc(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
template <int ROWS, int DH>
__device__ __forceinline__ void ag_fetch(AgStage<ROWS, DH>& st, __amdgpu_buffer_rsrc_t r, int byte0, int row_bytes, int tid) {
    constexpr int SPR = DH / 8;
#pragma unroll
    for (int i = 0; i < AgStage<ROWS, DH>::NIT; ++i) {
        const int p = tid + i * AG_THREADS;
        st.v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, byte0 + (p / SPR) * row_bytes + (p % SPR) * 16, 0, 0);
    }
}
template <int ROWS, int DH> __device__ __forceinline__ void ag_store(const AgStage<ROWS, DH>& st, char* lds, int tid) {
    constexpr int SPR = DH / 8;
#pragma unroll
    for (int i = 0; i < AgStage<ROWS, DH>::NIT; ++i) {
        const int p = tid + i * AG_THREADS;
        if (p < ROWS * SPR) *reinterpret_cast<u32x4*>(lds + img_off<bf16, DH>(p / SPR, p % SPR)) = st.v[i];
    }
}

__device__ __forceinline__ bf16x8 ag_pack(const f32x4& lo, const f32x4& hi) {    // two 16-row accumulator blocks -> one 32-deep operand
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = (bf16)lo[j];
        r[4 + j] = (bf16)hi[j];
    }
    return r;
}

#ifdef AG_STAMP
// diagnostic build (tools/attn_stamps.py; never part of the product library): every wave of one workgroup in the middle of the grid
// records the shader clock at the phase boundaries
__device__ unsigned long long g_ag_stamps[4][32];
#define AG_STAMP_HERE(k)                                                                                        \
    do {                                                                                                        \
        if (blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0) g_ag_stamps[threadIdx.x >> 6][k] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define AG_STAMP_HERE(k) do { } while (0)
#endif

template <int DH, int NB, bool EXACT> struct AgCtx {
    static constexpr int RB = DH * 2, NCH = DH / 32, NDB = DH / 16, NRC = (NB + 1) / 2;
    const char *slot0, *slot1;
    const float *ldsLse, *ldsDelta;
    int rowb[NCH], trb, N, g, li, lane;
    float scale, sl2;
};

// Operands a wave holds in registers for one pass: NJ 16-row blocks of (Q, dO, O) in phase A, of (K, V) in phase B.  A pass's operands are
// requested one pass AHEAD (the first phase-A pass with the K / V images at kernel start, the first phase-B pass from the K / V images in
// LDS before Q / dO replace them), so a workgroup's global-memory round trips are the images and nothing else: with the operand fetch at
// the top of every pass the kernel ran at 711 us per ViT-L layer (B = 665) even with 6/7 of its arithmetic removed, against 420 us for
// the same bytes moved by a kernel that only loads and stores (tools/attn_mem_pattern.hip).
template <int NJ, int NCH> struct AgOpsA {
    bf16x8 qf[NJ][NCH], dof[NJ][NCH], of[NJ][NCH];
};
template <int NJ, int NCH> struct AgOpsB {
    bf16x8 kf[NJ][NCH], vf[NJ][NCH];
};

template <int DH> __device__ __forceinline__ bf16x8 ag_frag(__amdgpu_buffer_rsrc_t r, int byte0, int row, int row_bytes, int c, int lane) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, byte0 + row * row_bytes + (4 * c + (lane >> 4)) * 16, 0, 0));
}
template <int NJ, int DH>
__device__ __forceinline__ void ag_a_fetch(AgOpsA<NJ, DH / 32>& o, int blk0, __amdgpu_buffer_rsrc_t rQ, int rs_bytes, __amdgpu_buffer_rsrc_t rDO,
                                           __amdgpu_buffer_rsrc_t rO, int d_bytes, int li, int lane) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = (blk0 + j) * 16 + li;
#pragma unroll
        for (int c = 0; c < DH / 32; ++c) {
            o.qf[j][c] = ag_frag<DH>(rQ, 0, q, rs_bytes, c, lane);
            o.dof[j][c] = ag_frag<DH>(rDO, 0, q, d_bytes, c, lane);
            o.of[j][c] = ag_frag<DH>(rO, 0, q, d_bytes, c, lane);
        }
    }
}
template <int NJ, int DH>
__device__ __forceinline__ void ag_b_fetch(AgOpsB<NJ, DH / 32>& o, int blk0, __amdgpu_buffer_rsrc_t rQ, int rs_bytes, int d_bytes, int li, int lane) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int key = (blk0 + j) * 16 + li;
#pragma unroll
        for (int c = 0; c < DH / 32; ++c) {
            o.kf[j][c] = ag_frag<DH>(rQ, d_bytes, key, rs_bytes, c, lane);
            o.vf[j][c] = ag_frag<DH>(rQ, 2 * d_bytes, key, rs_bytes, c, lane);
        }
    }
}
__device__ __forceinline__ void ag_store4(__amdgpu_buffer_rsrc_t r, int byte_off, const f32x4& v, float mul) {
    Vec4<bf16> o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o.set(i, v[i] * mul);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), r, byte_off, 0, 0);
}

// sum over the 16 lanes of a DPP row (= one lane group g), in every lane of the row: four v_add_f32 with DPP operands (quad swaps, then the
// half-row and the row mirrored), no LDS crossbar traffic (hipcc turns __shfl_xor by 4 and 8 into ds_bpermute_b32)
__device__ __forceinline__ float ag_row16_sum(float t) {
#define AG_DPP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true))
    t += AG_DPP(t, 0xB1);        // quad_perm [1, 0, 3, 2]
    t += AG_DPP(t, 0x4E);        // quad_perm [2, 3, 0, 1]
    t += AG_DPP(t, 0x141);       // row_half_mirror
    t += AG_DPP(t, 0x140);       // row_mirror
#undef AG_DPP
    return t;
}

// column sums of the dQ tiles for the qkv bias gradient: v[d][r] holds, for head-dim index 16 d + 4 g + r, this lane's column (one
// query or key); the 16 lanes of a group are summed and lane li = 0 adds the group's total into the wave's own LDS row (no atomics: index
// 16 d + 4 g + r belongs to exactly one lane of the wave, passes follow each other)
template <int NDB> __device__ __forceinline__ void ag_colsum_add(float (&v)[NDB][4], float* __restrict__ row, int g, int li, float mul) {
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = ag_row16_sum(v[d][r]);
            if (li == 0) row[d * 16 + 4 * g + r] += t * mul;
        }
}

// phase A for the query blocks blk0 .. blk0 + NJ - 1: K in slot 0, V in slot 1.  -delta is the initial value of the dP accumulator, so
// dS = P o (dP - delta) is one multiply.
template <int NJ, int DH, int NB, bool EXACT>
__device__ __forceinline__ void ag_dq_pass(const AgCtx<DH, NB, EXACT>& cx, int blk0, const AgOpsA<NJ, DH / 32>& o, float* __restrict__ ldsDeltaW,
                                           __amdgpu_buffer_rsrc_t rDQ, int rs_bytes, float* __restrict__ cs_row) {
    typedef bf16 T;
    typedef AgCtx<DH, NB, EXACT> C;
    constexpr int RB = C::RB, NCH = C::NCH, NDB = C::NDB, NRC = C::NRC;
    const int N = cx.N, g = cx.g, li = cx.li;
    float neg_lse[NJ], neg_delta[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = (blk0 + j) * 16 + li;
        float dl = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) dl = fmaf((float)o.dof[j][c][i], (float)o.of[j][c][i], dl);
        neg_delta[j] = -gsum(dl);                      // delta[q] = sum_d dO[q][d] O[q][d]; the lanes li, li + 16, .. hold the row's slots
        neg_lse[j] = -cx.ldsLse[q];
        if (g == 0) ldsDeltaW[q] = neg_delta[j];       // phase B reads -delta (after the barrier between the phases)
    }
    f32x4 dq[NJ][NDB];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int d = 0; d < NDB; ++d) dq[j][d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rc = 0; rc < NRC; ++rc) {
        f32x4 dst[NJ][2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int kb = 2 * rc + h2;
            if (kb < NB) {
                bf16x8 kfr[NCH], vfr[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    kfr[c] = af_row<RB>(cx.slot0, cx.rowb[c], kb);
                    vfr[c] = af_row<RB>(cx.slot1, cx.rowb[c], kb);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dp = {neg_delta[j], neg_delta[j], neg_delta[j], neg_delta[j]};
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        MmaS<T>::mma(sacc, kfr[c], o.qf[j][c]);
                        MmaS<T>::mma(dp, vfr[c], o.dof[j][c]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p = __builtin_amdgcn_exp2f(fmaf(sacc[r], cx.sl2, neg_lse[j]));
                        if (EXACT ? (kb == NB - 1) : (kb * 16 + 16 > N)) p = (kb * 16 + 4 * g + r < N) ? p : 0.f;     // ragged key block only
                        dst[j][h2][r] = p * dp[r];
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NJ; ++j) dst[j][h2] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        bf16x8 f[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) f[j] = ag_pack(dst[j][0], dst[j][1]);
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            const bf16x8 ktr = af_tr<RB>(cx.slot0, cx.trb ^ (d << 5), rc);
#pragma unroll
            for (int j = 0; j < NJ; ++j) MmaS<T>::mma(dq[j][d], ktr, f[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef AG_STAMP
        if (blk0 < AG_WAVES * NJ) AG_STAMP_HERE(17 + rc);
        else AG_STAMP_HERE(24 + rc);
#endif
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = (blk0 + j) * 16 + li;
#pragma unroll
        for (int d = 0; d < NDB; ++d) ag_store4(rDQ, q * rs_bytes + d * 32 + 8 * g, dq[j][d], cx.scale);       // rows >= N: dropped by the range check
    }
    if (cs_row) {                                      // padding queries carry dQ = 0 (their P is 0)
        float v[NDB][4];
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[d][r] = dq[0][d][r];
#pragma unroll
                for (int j = 1; j < NJ; ++j) v[d][r] += dq[j][d][r];
            }
        ag_colsum_add<NDB>(v, cs_row, g, li, cx.scale);
    }
}

// phase B for the key blocks blk0 .. blk0 + NJ - 1: Q in slot 0, dO in slot 1
template <int NJ, int DH, int NB, bool EXACT>
__device__ __forceinline__ void ag_dkv_pass(const AgCtx<DH, NB, EXACT>& cx, int blk0, const AgOpsB<NJ, DH / 32>& o, __amdgpu_buffer_rsrc_t rDQ, int rs_bytes,
                                            int d_bytes) {
    typedef bf16 T;
    typedef AgCtx<DH, NB, EXACT> C;
    constexpr int RB = C::RB, NCH = C::NCH, NDB = C::NDB, NRC = C::NRC;
    const int N = cx.N, g = cx.g, li = cx.li;
    f32x4 dk[NJ][NDB], dv[NJ][NDB];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            dk[j][d] = f32x4{0.f, 0.f, 0.f, 0.f};
            dv[j][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int rc = 0; rc < NRC; ++rc) {
        f32x4 pt[NJ][2], st[NJ][2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int qb = 2 * rc + h2;
            if (qb < NB) {
                bf16x8 qfr[NCH], dofr[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    qfr[c] = af_row<RB>(cx.slot0, cx.rowb[c], qb);
                    dofr[c] = af_row<RB>(cx.slot1, cx.rowb[c], qb);
                }
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(cx.ldsLse + qb * 16 + 4 * g);         // +inf on padding queries: P = 0
                const f32x4 nd4 = *reinterpret_cast<const f32x4*>(cx.ldsDelta + qb * 16 + 4 * g);      // -delta of the rows 4 g .. 4 g + 3
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dp = nd4;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        MmaS<T>::mma(sacc, qfr[c], o.kf[j][c]);
                        MmaS<T>::mma(dp, dofr[c], o.vf[j][c]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[r], cx.sl2, -l4[r]));
                        pt[j][h2][r] = p;
                        st[j][h2][r] = p * dp[r];
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    pt[j][h2] = f32x4{0.f, 0.f, 0.f, 0.f};
                    st[j][h2] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        bf16x8 fp[NJ], fs[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            fp[j] = ag_pack(pt[j][0], pt[j][1]);
            fs[j] = ag_pack(st[j][0], st[j][1]);
        }
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            const bf16x8 dotr = af_tr<RB>(cx.slot1, cx.trb ^ (d << 5), rc);
            const bf16x8 qtr = af_tr<RB>(cx.slot0, cx.trb ^ (d << 5), rc);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                MmaS<T>::mma(dv[j][d], dotr, fp[j]);
                MmaS<T>::mma(dk[j][d], qtr, fs[j]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int key = (blk0 + j) * 16 + li;
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            ag_store4(rDQ, d_bytes + key * rs_bytes + d * 32 + 8 * g, dk[j][d], cx.scale);
            ag_store4(rDQ, 2 * d_bytes + key * rs_bytes + d * 32 + 8 * g, dv[j][d], 1.f);
        }
    }
}

// NJ blocks per pass: 2 where a wave has more than one block per phase anyway (N > 128), 1 below that (every wave keeps a block)
template <int DH, int NB, bool EXACT>
__global__ __launch_bounds__(AG_THREADS, 2) void attn_g_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                                 const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                                 bf16* __restrict__ dqkv, float* __restrict__ cs_partial, int N, int H, float scale,
                                                                 float scale_log2e) {
    typedef bf16 T;
    constexpr int RB = DH * 2, NCH = DH / 32, NRC = (NB + 1) / 2, ROWS = NRC * 32, IMG = ROWS * RB, NJ = NB > 8 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* slot0 = smem;                 // K, then Q
    char* slot1 = smem + IMG;           // V, then dO
    float* ldsLse = reinterpret_cast<float*>(smem + 2 * IMG);   // [ROWS]
    float* ldsDelta = ldsLse + ROWS;
    float* ldsCs = ldsDelta + ROWS;                              // [4 waves][DH] column sums of dQ (cs_partial only)
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x % H;
    const int64_t b = blockIdx.x / H;
    const int D = H * DH;
    const int64_t rs = 3 * (int64_t)D;
    const T* qbase = qkv + b * N * rs + h * DH;
    const T* dobase = dout + b * N * (int64_t)D + h * DH;
    const T* obase = out + b * N * (int64_t)D + h * DH;
    T* dqbase = dqkv + b * N * rs + h * DH;
    const float* lse_bh = lse + (b * H + h) * N;
    const int nqb = (N + 15) / 16;
    constexpr int STEP = AG_WAVES * NJ;                 // blocks between a wave's two passes (N <= 256: never more than two per phase)
    const int rs_bytes = (int)rs * 2, d_bytes = D * 2;
    const __amdgpu_buffer_rsrc_t rQ = ag_rsrc(qbase, N * rs_bytes), rDO = ag_rsrc(dobase, N * d_bytes), rO = ag_rsrc(obase, N * d_bytes),
                                 rDQ = ag_rsrc(dqbase, N * rs_bytes);
    AG_STAMP_HERE(0);
    AgOpsA<NJ, NCH> oa, oa2;
    ag_a_fetch<NJ, DH>(oa, wave * NJ, rQ, rs_bytes, rDO, rO, d_bytes, li, lane);     // first phase-A pass: requested with the images
    {
        AgStage<ROWS, DH> sk, sv;
        ag_fetch<ROWS, DH>(sk, rQ, d_bytes, rs_bytes, tid);
        ag_fetch<ROWS, DH>(sv, rQ, 2 * d_bytes, rs_bytes, tid);
        ag_store<ROWS, DH>(sk, slot0, tid);
        ag_store<ROWS, DH>(sv, slot1, tid);
    }
    AG_STAMP_HERE(1);
    for (int i = tid; i < ROWS; i += AG_THREADS) {
        ldsLse[i] = i < N ? lse_bh[i] : INFINITY;    // +inf -> P = 0 for padding queries
        ldsDelta[i] = 0.f;                           // phase A fills the blocks that exist (with -delta)
    }
    if (cs_partial)
        for (int i = tid; i < AG_WAVES * DH; i += AG_THREADS) ldsCs[i] = 0.f;
    float* cs_row = cs_partial ? ldsCs + wave * DH : nullptr;
    AgStage<ROWS, DH> sq, sdo;                           // phase B's images travel while phase A computes
    ag_fetch<ROWS, DH>(sq, rQ, 0, rs_bytes, tid);
    ag_fetch<ROWS, DH>(sdo, rDO, 0, d_bytes, tid);
    ag_a_fetch<NJ, DH>(oa2, wave * NJ + STEP, rQ, rs_bytes, rDO, rO, d_bytes, li, lane);   // second pass (out of range where there is none)
    __syncthreads();
    AG_STAMP_HERE(2);

    const AfBases<DH> ab(lane);
    AgCtx<DH, NB, EXACT> cx;
    cx.slot0 = slot0;
    cx.slot1 = slot1;
    cx.ldsLse = ldsLse;
    cx.ldsDelta = ldsDelta;
#pragma unroll
    for (int c = 0; c < NCH; ++c) cx.rowb[c] = ab.rowb[c];
    cx.trb = ab.trb;
    cx.N = N;
    cx.g = lane >> 4;
    cx.li = li;
    cx.lane = lane;
    cx.scale = scale;
    cx.sl2 = scale_log2e;

    // ---------------- phase A: dQ (K in slot 0, V in slot 1) ----------------------------------------------------------
    ag_dq_pass<NJ>(cx, wave * NJ, oa, ldsDelta, rDQ, rs_bytes, cs_row);
    AG_STAMP_HERE(3);
    if (wave * NJ + STEP < nqb) ag_dq_pass<NJ>(cx, wave * NJ + STEP, oa2, ldsDelta, rDQ, rs_bytes, cs_row);
    AG_STAMP_HERE(4);
    // the first phase-B pass's K / V blocks come out of the images before Q / dO replace them (reverse wave order in phase B)
    const int bw = AG_WAVES - 1 - wave;
    AgOpsB<NJ, NCH> ob, ob2;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            ob.kf[j][c] = *reinterpret_cast<const bf16x8*>(slot0 + cx.rowb[c] + (bw * NJ + j) * (16 * RB));
            ob.vf[j][c] = *reinterpret_cast<const bf16x8*>(slot1 + cx.rowb[c] + (bw * NJ + j) * (16 * RB));
        }
    AG_STAMP_HERE(8);
    __syncthreads();                     // every wave is done with K and V
    AG_STAMP_HERE(9);
    ag_store<ROWS, DH>(sq, slot0, tid);
    ag_store<ROWS, DH>(sdo, slot1, tid);
    ag_b_fetch<NJ, DH>(ob2, bw * NJ + STEP, rQ, rs_bytes, d_bytes, li, lane);          // second phase-B pass: in flight under the first
    __syncthreads();
    AG_STAMP_HERE(10);

    // ---------------- phase B: dK, dV (Q in slot 0, dO in slot 1), reverse wave order ----------------------------------
    ag_dkv_pass<NJ>(cx, bw * NJ, ob, rDQ, rs_bytes, d_bytes);
    AG_STAMP_HERE(11);
    if (bw * NJ + STEP < nqb) ag_dkv_pass<NJ>(cx, bw * NJ + STEP, ob2, rDQ, rs_bytes, d_bytes);
    AG_STAMP_HERE(16);
    if (cs_partial) {
        // this (batch, head)'s column sums of dQ over its tokens, the four waves added in a fixed order: row b of [B][2][H][DH].  (Phase A is
        // long finished; the barrier between the phases ordered the waves' LDS rows.)
        for (int i = tid; i < 2 * DH; i += AG_THREADS) {
            const int d = i % DH;
            const float t = (ldsCs[d] + ldsCs[DH + d]) + (ldsCs[2 * DH + d] + ldsCs[3 * DH + d]);
            cs_partial[b * (2 * D) + (i / DH) * D + h * DH + d] = i < DH ? t : 0.f;      // [B][2][H][DH]: dQ's sums, then dK's (identically 0)
        }
    }
}


// =====================================================================================================================
// forward, bf16 head dim 64, three workgroups per CU: the resident forward above with (i) images of exactly 16 NB rows, V FIRST
// and K behind it — the transposed V reads of the last 32-row chunk run up to 16 rows past V's end, i.e. into K's first rows,
// which are finite numbers multiplied by P = 0 — so a workgroup needs 2 x 26 KiB of LDS and three fit a CU (159,744 of 163,840 B);
// (ii) the six-base addressing and sched_barriers of the fused backward, which bring it under the 170 VGPRs three waves per SIMD
// allow.  One more workgroup per CU = one more head's loads in flight behind the MFMA / softmax work of the other two.
// =====================================================================================================================
template <int DH, int NB, bool EXACT>
__global__ __launch_bounds__(AG_THREADS, 3) void attn_s3_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, float* __restrict__ lse,
                                                                     int N, int H, float scale_log2e) {
    typedef bf16 T;
    constexpr int RB = DH * 2, NCH = DH / 32, NDB = DH / 16, NRC = (NB + 1) / 2, ROWS = NB * 16, IMG = ROWS * RB;
    static_assert(NRC * 32 - ROWS <= 16, "the last transposed chunk may only run into K's first 16 rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsV = smem;
    char* ldsK = smem + IMG;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x % H;
    const int64_t b = blockIdx.x / H;
    const int D = H * DH;
    const int64_t rs = 3 * (int64_t)D;
    const T* qbase = qkv + b * N * rs + h * DH;
    // the query blocks of ALL the wave's passes are requested with the images (range-checked buffer loads: rows >= N read as 0): the
    // workgroup has one global-memory round trip instead of one per pass
    constexpr int NPASS = (NB + AG_WAVES - 1) / AG_WAVES;
    const int rs_bytes = (int)rs * 2;
    const __amdgpu_buffer_rsrc_t rQ = ag_rsrc(qbase, N * rs_bytes);
    bf16x8 qf_all[NPASS][NCH];
#pragma unroll
    for (int i = 0; i < NPASS; ++i)
#pragma unroll
        for (int c = 0; c < NCH; ++c) qf_all[i][c] = ag_frag<DH>(rQ, 0, (wave + i * AG_WAVES) * 16 + li, rs_bytes, c, lane);
    {
        AgStage<ROWS, DH> sk, sv;
        ag_fetch<ROWS, DH>(sk, rQ, D * 2, rs_bytes, tid);
        ag_fetch<ROWS, DH>(sv, rQ, 2 * D * 2, rs_bytes, tid);
        ag_store<ROWS, DH>(sk, ldsK, tid);
        ag_store<ROWS, DH>(sv, ldsV, tid);
    }
    __syncthreads();
    const AfBases<DH> ab(lane);
    const int trb = ab.trb;
    constexpr int nb_ = NB;
    const int nqb = (N + 15) / 16;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int qb = wave + ps * AG_WAVES;
        if (qb >= nqb) break;
        const int q = qb * 16 + li;
        bf16x8 qf[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) qf[c] = qf_all[ps][c];
        f32x4 s[NB];
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            s[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NCH; ++c) MmaS<T>::mma(s[kb], af_row<RB>(ldsK, ab.rowb[c], kb), qf[c]);
            if (EXACT ? (kb == NB - 1) : (kb * 16 + 16 > N)) {   // only the ragged / padding key blocks pay for masking
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kb][r] = (kb * 16 + 4 * g + r < N) ? s[kb][r] : -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][r]);
            if ((kb & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        const float m = gmax(mx) * scale_log2e;
        float l = 0.f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], scale_log2e, -m));
                s[kb][r] = p;
                l += p;
            }
        const float lt = gsum(l);
        f32x4 o[NDB];
#pragma unroll
        for (int d = 0; d < NDB; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rc = 0; rc < NRC; ++rc) {
            const bf16x8 pf = s_frag_acc<T, nb_>(s, rc);
#pragma unroll
            for (int d = 0; d < NDB; ++d) MmaS<T>::mma(o[d], af_tr<RB>(ldsV, trb ^ (d << 5), rc), pf);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q < N) {
            const float inv = 1.f / lt;
            T* op = out + (b * N + q) * D + h * DH;
#pragma unroll
            for (int d = 0; d < NDB; ++d) {
                Vec4<T> v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v.set(r, o[d][r] * inv);
                *reinterpret_cast<Vec4<T>*>(op + d * 16 + 4 * g) = v;
            }
            if (g == 0) lse[(b * H + h) * N + q] = m + log2f(lt);
        }
    }
}

template <int DH, int NB>
int launch_s3_fwd(const void* qkv, void* out, float* lse, int64_t B, int64_t N, int64_t H, float scale, hipStream_t s) {
    constexpr size_t smem = 2 * (size_t)NB * 16 * (DH * 2);
    const float sl2 = scale * 1.44269504088896340736f;
    const dim3 grid((unsigned)(B * H)), block(AG_THREADS);
    if ((N + 15) / 16 == NB) {
        if (int rc = big_lds(attn_s3_fwd_kernel<DH, NB, true>, smem)) return rc;
        hipLaunchKernelGGL((attn_s3_fwd_kernel<DH, NB, true>), grid, block, smem, s, (const bf16*)qkv, (bf16*)out, lse, (int)N, (int)H, sl2);
    } else {
        if (int rc = big_lds(attn_s3_fwd_kernel<DH, NB, false>, smem)) return rc;
        hipLaunchKernelGGL((attn_s3_fwd_kernel<DH, NB, false>), grid, block, smem, s, (const bf16*)qkv, (bf16*)out, lse, (int)N, (int)H, sl2);
    }
    UCF_LAUNCH_CHECK("ucfvit_attention_fwd(short, 3 per CU)");
    return UCFVIT_OK;
}

template <int DH, int NB>
int launch_fused_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* cs_partial, int64_t B, int64_t N, int64_t H,
                     float scale, hipStream_t s) {
    constexpr int ROWS = ((NB + 1) / 2) * 32;
    constexpr size_t smem = 2 * (size_t)ROWS * (DH * 2) + 2 * ROWS * sizeof(float) + AG_WAVES * DH * sizeof(float);
    const float sl2 = scale * 1.44269504088896340736f;
    const dim3 grid((unsigned)(B * H)), block(AG_THREADS);
    if ((N + 15) / 16 == NB) {
        if (int rc = big_lds(attn_g_bwd_kernel<DH, NB, true>, smem)) return rc;
        hipLaunchKernelGGL((attn_g_bwd_kernel<DH, NB, true>), grid, block, smem, s, (const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, cs_partial, (int)N,
                           (int)H, scale, sl2);
    } else {
        if (int rc = big_lds(attn_g_bwd_kernel<DH, NB, false>, smem)) return rc;
        hipLaunchKernelGGL((attn_g_bwd_kernel<DH, NB, false>), grid, block, smem, s, (const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, cs_partial, (int)N,
                           (int)H, scale, sl2);
    }
    UCF_LAUNCH_CHECK("ucfvit_attention_bwd(fused)");
    return UCFVIT_OK;
}

}  // namespace

#ifdef AG_STAMP
extern "C" int ucfvit_debug_attn_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ag_stamps), sizeof(unsigned long long) * 4 * 32);
}
#endif

// returns 1 when handled, 0 when the shape is outside the short-sequence kernels (caller streams), <0 on error
#define AS_PICK(FN, T, DH, ...)                          \
    do {                                                 \
        if (nb <= 4) return FN<T, DH, 4>(__VA_ARGS__) == UCFVIT_OK ? 1 : UCFVIT_ERR_HIP;   \
        if (nb <= 8) return FN<T, DH, 8>(__VA_ARGS__) == UCFVIT_OK ? 1 : UCFVIT_ERR_HIP;   \
        return FN<T, DH, 16>(__VA_ARGS__) == UCFVIT_OK ? 1 : UCFVIT_ERR_HIP;               \
    } while (0)

int ucfvit_attention_short_fwd(const void* qkv, void* out, float* lse, int64_t B, int64_t N, int64_t H, int64_t dh, float scale, int dtype,
                               hipStream_t s) {
    // bf16 only: the fp32 instantiations exceed the register file (the exact-fp32 parity mode keeps the streaming kernels)
    if (dtype != UCFVIT_BF16 || N > 256 || (dh != 32 && dh != 64) || B * H >= (1ll << 31)) return 0;
    const int nb = (int)((N + 15) / 16);
    if (nb > 8 && nb <= 13) {          // 129 .. 208 tokens (N = 197): K and V resident as LDS images, three workgroups per CU
        const int rc = dh == 64 ? launch_s3_fwd<64, 13>(qkv, out, lse, B, N, H, scale, s) : launch_s3_fwd<32, 13>(qkv, out, lse, B, N, H, scale, s);
        return rc == UCFVIT_OK ? 1 : UCFVIT_ERR_HIP;
    }
    if (dh == 64) AS_PICK(launch_fwd, bf16, 64, qkv, out, lse, B, N, H, scale, s);
    AS_PICK(launch_fwd, bf16, 32, qkv, out, lse, B, N, H, scale, s);
}

int ucfvit_attention_fused_bwd_applies(int64_t B, int64_t N, int64_t H, int64_t dh, int dtype) {
    return (dtype == UCFVIT_BF16 && N <= 256 && (dh == 64 || dh == 32) && B * H < (1ll << 31)) ? 1 : 0;
}

// fused backward (bf16, head dim 64 or 32, N <= 256; needs no delta workspace): 1 = handled, 0 = not applicable, <0 = error.
// cs_partial (may be null): fp32 [B][2][H][dh], row b = the column sums of dQ over batch element b's tokens, then zeros for dK (see
// ucfvit_attention_bwd_colsum)
int ucfvit_attention_fused_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* cs_partial, int64_t B, int64_t N,
                               int64_t H, int64_t dh, float scale, int dtype, hipStream_t s) {
    if (!ucfvit_attention_fused_bwd_applies(B, N, H, dh, dtype)) return 0;
    const int nb = (int)((N + 15) / 16);
    int rc;
#define AF_BWD(DH_)                                                                                       \
    (nb <= 4 ? launch_fused_bwd<DH_, 4>(qkv, out, dout, lse, dqkv, cs_partial, B, N, H, scale, s)                     \
             : nb <= 8 ? launch_fused_bwd<DH_, 8>(qkv, out, dout, lse, dqkv, cs_partial, B, N, H, scale, s)           \
                       : nb <= 13 ? launch_fused_bwd<DH_, 13>(qkv, out, dout, lse, dqkv, cs_partial, B, N, H, scale, s) \
                                  : launch_fused_bwd<DH_, 16>(qkv, out, dout, lse, dqkv, cs_partial, B, N, H, scale, s))
    rc = dh == 64 ? AF_BWD(64) : AF_BWD(32);
#undef AF_BWD
    return rc == UCFVIT_OK ? 1 : rc;
}

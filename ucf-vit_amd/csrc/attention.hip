// Fused multi-head self-attention (flash style) for gfx950, forward + backward, dh in {32, 64, 128},
// bf16 (v_mfma_f32_16x16x32_bf16) or exact fp32 (v_mfma_f32_16x16x4_f32).
//
// Reads q/k/v straight out of the packed qkv GEMM output [B][N][3][H][dh] and writes [B][N][H*dh] — the reference's
// reshape/permute/transpose copies (building_blocks.py:159,180,189) never materialise.
//
// Layout idea ("the softmax index lives on the lane"): scores are computed TRANSPOSED, Sᵀ[key][q] = K·Qᵀ, so a lane
// (q = lane & 15) owns one query column: the row max / row sum are in-register reductions plus two cross-group
// shuffles (wavefront-reduced softmax), and the 16x16 accumulator of Sᵀ is already the B operand of the next MFMA
// (Oᵀ[d][q] += Vᵀ[d][key]·Pᵀ[key][q]) with no LDS round trip.  Vᵀ / Kᵀ / Qᵀ / dOᵀ operands come from the row-major
// LDS tiles through ds_read_b64_tr_b16 (hardware transpose).  The backward pass uses the same trick with the key on
// the lane for dK/dV and the query on the lane for dQ, recomputing P from the saved log-sum-exp; no atomics, so
// gradients are bitwise reproducible.
#include <stdlib.h>

#include "common.h"

// attention_short.hip: whole-sequence-in-LDS kernels for N <= 256 (return 1 = handled, 0 = not applicable, <0 = error)
int ucfvit_attention_short_fwd(const void* qkv, void* out, float* lse, int64_t B, int64_t N, int64_t H, int64_t dh, float scale, int dtype,
                               hipStream_t s);
int ucfvit_attention_fused_bwd(const void* qkv, const void* dout, const float* lse, void* dqkv, int64_t B, int64_t N, int64_t H, int64_t dh,
                               float scale, int dtype, hipStream_t s);

static bool short_enabled() {
    static int flag = -1;
    if (flag < 0) {
        const char* e = getenv("UCFVIT_ATTN_STREAM");
        flag = (e && e[0] == '1') ? 0 : 1;
    }
    return flag == 1;
}

namespace {

constexpr int AT_THREADS = 256;  // 4 waves x 16 lane-columns = 64 queries (or keys) per workgroup
constexpr int KT = 64;           // rows per LDS tile

template <typename T> struct Mma16;
template <> struct Mma16<bf16> {
    typedef bf16x8 frag_t;
    static __device__ __forceinline__ void mma(f32x4& acc, const frag_t& a, const frag_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
};
template <> struct Mma16<float> {
    typedef f32x4 frag_t;
    static __device__ __forceinline__ void mma(f32x4& acc, const frag_t& a, const frag_t& b) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
    }
};

template <typename T, int DH> struct AT {
    static constexpr int EPV = 16 / sizeof(T);
    static constexpr int RB = DH * sizeof(T);    // tile row bytes
    static constexpr int SPR = RB / 16;          // 16-B slots per row
    static constexpr int NCH = DH / (4 * EPV);   // chunks of the head dim (contraction over d)
    static constexpr int NDB = DH / 16;          // 16-wide blocks of the head dim (outputs)
    static constexpr int NRC = KT / (4 * EPV);   // chunks of the tile rows (contraction over keys / queries)
    static constexpr int ACC_PER_RC = (4 * EPV) / 16;  // 16-row accumulator blocks per row chunk (bf16: 2, fp32: 1)
    static constexpr int PPT = KT * SPR / AT_THREADS;  // 16-B pieces per thread per tile
    static constexpr int TILE_BYTES = KT * RB;
    typedef typename Mma16<T>::frag_t frag_t;
};

template <int SPR> __device__ __forceinline__ int slot_swz(int row) {
    if (SPR == 4) return (0x1230 >> (((row >> 2) & 3) * 4)) & 3;  // {0,3,2,1}[(row>>2)&3]
    if (SPR == 8) return row & 7;
    return row & 15;
}
template <typename T, int DH> __device__ __forceinline__ int tile_off(int row, int slot) {
    return row * AT<T, DH>::RB + ((slot ^ slot_swz<AT<T, DH>::SPR>(row)) << 4);
}

template <typename T, int DH> struct TileStage {
    u32x4 v[AT<T, DH>::PPT];
};

// rows [r0, r0+64) of a [.., row_stride] matrix (DH contiguous elements per row) -> registers; rows >= R read as zero
template <typename T, int DH>
__device__ __forceinline__ void tile_load(TileStage<T, DH>& st, const T* __restrict__ base, int64_t row_stride, int r0, int R, int tid) {
    typedef AT<T, DH> A;
#pragma unroll
    for (int i = 0; i < A::PPT; ++i) {
        const int p = tid + i * AT_THREADS;
        const int row = p / A::SPR, slot = p % A::SPR;
        u32x4 z = {0u, 0u, 0u, 0u};
        st.v[i] = (r0 + row < R) ? *reinterpret_cast<const u32x4*>(base + (int64_t)(r0 + row) * row_stride + slot * A::EPV) : z;
    }
}
template <typename T, int DH> __device__ __forceinline__ void tile_store(const TileStage<T, DH>& st, char* lds, int tid) {
    typedef AT<T, DH> A;
#pragma unroll
    for (int i = 0; i < A::PPT; ++i) {
        const int p = tid + i * AT_THREADS;
        *reinterpret_cast<u32x4*>(lds + tile_off<T, DH>(p / A::SPR, p % A::SPR)) = st.v[i];
    }
}

// "row" fragment: 16 B of row (rb*16 + lane&15) at head-dim chunk c: elements d = (4c + g)*EPV .. +EPV
template <typename T, int DH>
__device__ __forceinline__ typename AT<T, DH>::frag_t frag_row(const char* lds, int rb, int c, int lane) {
    const int row = rb * 16 + (lane & 15), g = lane >> 4;
    return *reinterpret_cast<const typename AT<T, DH>::frag_t*>(lds + tile_off<T, DH>(row, 4 * c + g));
}
// "transposed" fragment for a contraction over tile rows: lane (lane&15 = i) gets column d = db*16 + i of the rows
// of row-chunk rc in ACCUMULATOR order: bf16: rows 32rc + 16*(j>>2) + 4g + (j&3), j=0..7 ; fp32: rows 16rc + 4g + s.
template <typename T, int DH>
__device__ __forceinline__ typename AT<T, DH>::frag_t frag_tr(const char* lds, int rc, int db, int lane) {
    typedef typename AT<T, DH>::frag_t frag_t;
    const int g = lane >> 4, i = lane & 15;
    if constexpr (sizeof(T) == 2) {
        const int q = i >> 2, p = i & 3;
        const int slot = 2 * db + (p >> 1), sub = (p & 1) * 8;
        const int r_lo = 32 * rc + 4 * g + q, r_hi = r_lo + 16;
        const char* a_lo = lds + tile_off<T, DH>(r_lo, slot) + sub;
        const char* a_hi = lds + tile_off<T, DH>(r_hi, slot) + sub;
        short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, a_lo));
        short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, a_hi));
        short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(frag_t, r);
    } else {
        const int slot = 4 * db + (i >> 2), sub = (i & 3) * 4;
        f32x4 r;
#pragma unroll
        for (int s = 0; s < 4; ++s) r[s] = *reinterpret_cast<const float*>(lds + tile_off<T, DH>(16 * rc + 4 * g + s, slot) + sub);
        return __builtin_bit_cast(frag_t, r);
    }
}
// accumulator blocks -> operand fragment for row-chunk rc (bf16: two 16-row blocks packed; fp32: one block as is)
template <typename T> __device__ __forceinline__ typename Mma16<T>::frag_t frag_from_acc(const f32x4* acc, int rc) {
    typedef typename Mma16<T>::frag_t frag_t;
    if constexpr (sizeof(T) == 2) {
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[j] = (bf16)acc[2 * rc][j];
            r[4 + j] = (bf16)acc[2 * rc + 1][j];
        }
        return __builtin_bit_cast(frag_t, r);
    } else {
        return __builtin_bit_cast(frag_t, acc[rc]);
    }
}
// 16 B of a global row as an operand fragment (zero beyond R)
template <typename T, int DH>
__device__ __forceinline__ typename AT<T, DH>::frag_t frag_global(const T* __restrict__ base, int64_t row_stride, int row, int R, int c, int lane) {
    typedef typename AT<T, DH>::frag_t frag_t;
    const int g = lane >> 4;
    u32x4 z = {0u, 0u, 0u, 0u};
    u32x4 v = (row < R) ? *reinterpret_cast<const u32x4*>(base + (int64_t)row * row_stride + (4 * c + g) * AT<T, DH>::EPV) : z;
    return __builtin_bit_cast(frag_t, v);
}

__device__ __forceinline__ float group_max(float v) {  // over the 4 lane groups that share lane&15
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ===================================================================================================
// forward
// ===================================================================================================
template <typename T, int DH>
__global__ __launch_bounds__(AT_THREADS) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse,
                                                               int N, int H, float scale_log2e) {
    typedef AT<T, DH> A;
    typedef typename A::frag_t frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsK = smem;
    char* ldsV = smem + A::TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int D = H * DH;
    const int64_t rs = 3 * (int64_t)D;  // token stride inside qkv
    const T* qbase = qkv + b * N * rs + h * DH;
    const T* kbase = qbase + D;
    const T* vbase = qbase + 2 * D;
    const int q = blockIdx.x * 64 + wave * 16 + li;

    frag_t qf[A::NCH];
#pragma unroll
    for (int c = 0; c < A::NCH; ++c) qf[c] = frag_global<T, DH>(qbase, rs, q, N, c, lane);

    f32x4 o[A::NDB];
#pragma unroll
    for (int d = 0; d < A::NDB; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;

    const int ntiles = (N + KT - 1) / KT;
    TileStage<T, DH> sk, sv;
    tile_load<T, DH>(sk, kbase, rs, 0, N, tid);
    tile_load<T, DH>(sv, vbase, rs, 0, N, tid);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        tile_store<T, DH>(sk, ldsK, tid);
        tile_store<T, DH>(sv, ldsV, tid);
        __syncthreads();
        if (kt + 1 < ntiles) {
            tile_load<T, DH>(sk, kbase, rs, (kt + 1) * KT, N, tid);
            tile_load<T, DH>(sv, vbase, rs, (kt + 1) * KT, N, tid);
        }
        // Sᵀ[key][q] for the tile's 4 key blocks
        f32x4 s[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            s[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < A::NCH; ++c) Mma16<T>::mma(s[kb], frag_row<T, DH>(ldsK, kb, c, lane), qf[c]);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * KT + kb * 16 + 4 * g + r;
                const float t = key < N ? s[kb][r] * scale_log2e : -INFINITY;
                s[kb][r] = t;
                mx = fmaxf(mx, t);
            }
        const float m_new = fmaxf(m, group_max(mx));  // finite: every tile holds at least one valid key
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[kb][r] - m_new);
                s[kb][r] = p;
                psum += p;
            }
        l = l * alpha + psum;
        m = m_new;
#pragma unroll
        for (int d = 0; d < A::NDB; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[d][r] *= alpha;
        // Oᵀ[d][q] += Vᵀ[d][key] · Pᵀ[key][q]
#pragma unroll
        for (int rc = 0; rc < A::NRC; ++rc) {
            const frag_t pf = frag_from_acc<T>(s, rc);
#pragma unroll
            for (int d = 0; d < A::NDB; ++d) Mma16<T>::mma(o[d], frag_tr<T, DH>(ldsV, rc, d, lane), pf);
        }
    }
    const float lt = group_sum(l);
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

// delta[b][h][q] = sum_d dO·O : DH/EPV consecutive lanes share one (token, head) row segment (coalesced 16-B loads)
template <typename T, int DH>
__global__ void attn_delta_kernel(const T* __restrict__ out, const T* __restrict__ dout, float* __restrict__ delta, int64_t B, int N, int H) {
    constexpr int EPV = 16 / sizeof(T);
    constexpr int LPR = DH / EPV;  // lanes per (token, head): 4, 8, 16 or 32
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over (b, q, h, piece)
    const int64_t total = B * N * H * LPR;
    float s = 0.f;
    if (i < total) {
        const Vec16<T> a = *reinterpret_cast<const Vec16<T>*>(out + i * EPV);
        const Vec16<T> c = *reinterpret_cast<const Vec16<T>*>(dout + i * EPV);
#pragma unroll
        for (int e = 0; e < EPV; ++e) s += a.get(e) * c.get(e);
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (i < total && (threadIdx.x & (LPR - 1)) == 0) {
        const int64_t bqh = i / LPR;
        const int h = bqh % H;
        const int64_t bq = bqh / H;
        const int q = bq % N;
        const int64_t b = bq / N;
        delta[(b * H + h) * N + q] = s;
    }
}

// ===================================================================================================
// backward, dQ: query on the lane, loop over key tiles
// ===================================================================================================
template <typename T, int DH>
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  T* __restrict__ dqkv, int N, int H, float scale, float scale_log2e) {
    typedef AT<T, DH> A;
    typedef typename A::frag_t frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsK = smem;
    char* ldsV = smem + A::TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int D = H * DH;
    const int64_t rs = 3 * (int64_t)D;
    const T* qbase = qkv + b * N * rs + h * DH;
    const T* kbase = qbase + D;
    const T* vbase = qbase + 2 * D;
    const T* dobase = dout + b * N * (int64_t)D + h * DH;
    const int q = blockIdx.x * 64 + wave * 16 + li;

    frag_t qf[A::NCH], dof[A::NCH];
#pragma unroll
    for (int c = 0; c < A::NCH; ++c) {
        qf[c] = frag_global<T, DH>(qbase, rs, q, N, c, lane);
        dof[c] = frag_global<T, DH>(dobase, D, q, N, c, lane);
    }
    const float my_lse = q < N ? lse[(b * H + h) * N + q] : 0.f;
    const float my_delta = q < N ? delta[(b * H + h) * N + q] : 0.f;

    f32x4 dq[A::NDB];
#pragma unroll
    for (int d = 0; d < A::NDB; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntiles = (N + KT - 1) / KT;
    TileStage<T, DH> sk, sv;
    tile_load<T, DH>(sk, kbase, rs, 0, N, tid);
    tile_load<T, DH>(sv, vbase, rs, 0, N, tid);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        tile_store<T, DH>(sk, ldsK, tid);
        tile_store<T, DH>(sv, ldsV, tid);
        __syncthreads();
        if (kt + 1 < ntiles) {
            tile_load<T, DH>(sk, kbase, rs, (kt + 1) * KT, N, tid);
            tile_load<T, DH>(sv, vbase, rs, (kt + 1) * KT, N, tid);
        }
        f32x4 ds[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < A::NCH; ++c) {
                Mma16<T>::mma(s, frag_row<T, DH>(ldsK, kb, c, lane), qf[c]);    // Sᵀ[key][q]
                Mma16<T>::mma(dp, frag_row<T, DH>(ldsV, kb, c, lane), dof[c]);  // dPᵀ[key][q]
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * KT + kb * 16 + 4 * g + r;
                const float p = key < N ? __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -my_lse)) : 0.f;
                ds[kb][r] = p * (dp[r] - my_delta);
            }
        }
        // dQᵀ[d][q] += Kᵀ[d][key] · dSᵀ[key][q]
#pragma unroll
        for (int rc = 0; rc < A::NRC; ++rc) {
            const frag_t f = frag_from_acc<T>(ds, rc);
#pragma unroll
            for (int d = 0; d < A::NDB; ++d) Mma16<T>::mma(dq[d], frag_tr<T, DH>(ldsK, rc, d, lane), f);
        }
    }
    if (q < N) {
        T* op = dqkv + (b * N + q) * rs + h * DH;
#pragma unroll
        for (int d = 0; d < A::NDB; ++d) {
            Vec4<T> v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v.set(r, dq[d][r] * scale);
            *reinterpret_cast<Vec4<T>*>(op + d * 16 + 4 * g) = v;
        }
    }
}

// ===================================================================================================
// backward, dK / dV: key on the lane, loop over query tiles
// ===================================================================================================
template <typename T, int DH>
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                                   T* __restrict__ dqkv, int N, int H, float scale, float scale_log2e) {
    typedef AT<T, DH> A;
    typedef typename A::frag_t frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsQ = smem;
    char* ldsDO = smem + A::TILE_BYTES;
    float* ldsLse = reinterpret_cast<float*>(smem + 2 * A::TILE_BYTES);  // [64]
    float* ldsDelta = ldsLse + KT;                                       // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int D = H * DH;
    const int64_t rs = 3 * (int64_t)D;
    const T* qbase = qkv + b * N * rs + h * DH;
    const T* kbase = qbase + D;
    const T* vbase = qbase + 2 * D;
    const T* dobase = dout + b * N * (int64_t)D + h * DH;
    const float* lse_bh = lse + (b * H + h) * N;
    const float* delta_bh = delta + (b * H + h) * N;
    const int key = blockIdx.x * 64 + wave * 16 + li;

    frag_t kf[A::NCH], vf[A::NCH];
#pragma unroll
    for (int c = 0; c < A::NCH; ++c) {
        kf[c] = frag_global<T, DH>(kbase, rs, key, N, c, lane);
        vf[c] = frag_global<T, DH>(vbase, rs, key, N, c, lane);
    }
    f32x4 dk[A::NDB], dv[A::NDB];
#pragma unroll
    for (int d = 0; d < A::NDB; ++d) {
        dk[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int ntiles = (N + KT - 1) / KT;
    TileStage<T, DH> sq, sdo;
    tile_load<T, DH>(sq, qbase, rs, 0, N, tid);
    tile_load<T, DH>(sdo, dobase, D, 0, N, tid);
    for (int qt = 0; qt < ntiles; ++qt) {
        __syncthreads();
        tile_store<T, DH>(sq, ldsQ, tid);
        tile_store<T, DH>(sdo, ldsDO, tid);
        if (tid < KT) {
            const int qq = qt * KT + tid;
            ldsLse[tid] = qq < N ? lse_bh[qq] : INFINITY;  // +inf -> P = 0 for padding queries
            ldsDelta[tid] = qq < N ? delta_bh[qq] : 0.f;
        }
        __syncthreads();
        if (qt + 1 < ntiles) {
            tile_load<T, DH>(sq, qbase, rs, (qt + 1) * KT, N, tid);
            tile_load<T, DH>(sdo, dobase, D, (qt + 1) * KT, N, tid);
        }
        f32x4 pm[4], ds[4];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < A::NCH; ++c) {
                Mma16<T>::mma(s, frag_row<T, DH>(ldsQ, qb, c, lane), kf[c]);     // S[q][key]
                Mma16<T>::mma(dp, frag_row<T, DH>(ldsDO, qb, c, lane), vf[c]);   // dP[q][key]
            }
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(ldsLse + qb * 16 + 4 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(ldsDelta + qb * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -l4[r]));
                pm[qb][r] = p;
                ds[qb][r] = p * (dp[r] - d4[r]);
            }
        }
#pragma unroll
        for (int rc = 0; rc < A::NRC; ++rc) {
            const frag_t fp = frag_from_acc<T>(pm, rc);
            const frag_t fs = frag_from_acc<T>(ds, rc);
#pragma unroll
            for (int d = 0; d < A::NDB; ++d) {
                Mma16<T>::mma(dv[d], frag_tr<T, DH>(ldsDO, rc, d, lane), fp);  // dVᵀ[d][key] += dOᵀ[d][q]·P[q][key]
                Mma16<T>::mma(dk[d], frag_tr<T, DH>(ldsQ, rc, d, lane), fs);   // dKᵀ[d][key] += Qᵀ[d][q]·dS[q][key]
            }
        }
    }
    if (key < N) {
        T* kp = dqkv + (b * N + key) * rs + D + h * DH;
        T* vp = kp + D;
#pragma unroll
        for (int d = 0; d < A::NDB; ++d) {
            Vec4<T> a, c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a.set(r, dk[d][r] * scale);
                c.set(r, dv[d][r]);
            }
            *reinterpret_cast<Vec4<T>*>(kp + d * 16 + 4 * g) = a;
            *reinterpret_cast<Vec4<T>*>(vp + d * 16 + 4 * g) = c;
        }
    }
}

template <typename K> int allow_big_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return UCFVIT_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        ucfvit_set_error("attention: cannot raise dynamic LDS to %zu bytes: %s", bytes, hipGetErrorString(e));
        return UCFVIT_ERR_HIP;
    }
    return UCFVIT_OK;
}

template <typename T, int DH>
int attn_fwd_launch(const void* qkv, void* out, float* lse, int64_t B, int64_t N, int64_t H, float scale, hipStream_t s) {
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)H, (unsigned)B);
    constexpr size_t smem_fwd = 2 * AT<T, DH>::TILE_BYTES;
    if (int rc = allow_big_lds(attn_fwd_kernel<T, DH>, smem_fwd)) return rc;
    hipLaunchKernelGGL((attn_fwd_kernel<T, DH>), grid, dim3(AT_THREADS), smem_fwd, s, (const T*)qkv, (T*)out, lse, (int)N,
                       (int)H, scale * 1.44269504088896340736f);
    UCF_LAUNCH_CHECK("ucfvit_attention_fwd");
    return UCFVIT_OK;
}

template <typename T, int DH>
int attn_bwd_launch(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta, int64_t B, int64_t N,
                    int64_t H, float scale, hipStream_t s) {
    // the FUSED backward (one launch, operands read once, delta taken from P and dP inside) is the default where it applies
    static const bool fused_bwd = [] { const char* e = getenv("UCFVIT_ATTN_FUSED_BWD"); return !(e && e[0] == '0'); }();
    if (short_enabled() && fused_bwd) {
        const int rc = ucfvit_attention_fused_bwd(qkv, dout, lse, dqkv, B, N, H, DH, scale, sizeof(T) == 2 ? UCFVIT_BF16 : UCFVIT_F32, s);
        if (rc == 1) return UCFVIT_OK;
        if (rc < 0) return rc;
    }
    const int64_t nd = B * N * H * (DH / (16 / (int64_t)sizeof(T)));
    hipLaunchKernelGGL((attn_delta_kernel<T, DH>), dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, (const T*)out, (const T*)dout, delta, B,
                       (int)N, (int)H);
    UCF_LAUNCH_CHECK("ucfvit_attention_bwd(delta)");
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)H, (unsigned)B);
    const float sl2 = scale * 1.44269504088896340736f;
    constexpr size_t smem_dq = 2 * AT<T, DH>::TILE_BYTES;
    constexpr size_t smem_dkv = smem_dq + 2 * KT * sizeof(float);
    if (int rc = allow_big_lds(attn_bwd_dq_kernel<T, DH>, smem_dq)) return rc;
    if (int rc = allow_big_lds(attn_bwd_dkv_kernel<T, DH>, smem_dkv)) return rc;
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DH>), grid, dim3(AT_THREADS), smem_dq, s, (const T*)qkv, (const T*)dout, lse,
                       (const float*)delta, (T*)dqkv, (int)N, (int)H, scale, sl2);
    UCF_LAUNCH_CHECK("ucfvit_attention_bwd(dq)");
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DH>), grid, dim3(AT_THREADS), smem_dkv, s,
                       (const T*)qkv, (const T*)dout, lse, (const float*)delta, (T*)dqkv, (int)N, (int)H, scale, sl2);
    UCF_LAUNCH_CHECK("ucfvit_attention_bwd(dkv)");
    return UCFVIT_OK;
}

int check_attn_args(const char* name, int64_t B, int64_t N, int64_t H, int64_t dh, int dtype) {
    UCF_CHECK_ARG(B > 0 && N > 0 && H > 0, "%s: bad shape B=%lld N=%lld H=%lld", name, (long long)B, (long long)N, (long long)H);
    UCF_CHECK_ARG(dh == 32 || dh == 64 || dh == 128, "%s: head dim %lld not in {32,64,128}", name, (long long)dh);
    UCF_CHECK_ARG(dtype == UCFVIT_F32 || dtype == UCFVIT_BF16, "%s: bad dtype %d", name, dtype);
    UCF_CHECK_ARG(H < 65536 && B < 65536 && N < (1ll << 30), "%s: grid too large", name);
    return UCFVIT_OK;
}

}  // namespace

#define ATTN_DISPATCH(FN, ...)                                                                 \
    do {                                                                                       \
        if (dtype == UCFVIT_BF16) {                                                            \
            if (dh == 32) return FN<bf16, 32>(__VA_ARGS__);                                    \
            if (dh == 64) return FN<bf16, 64>(__VA_ARGS__);                                    \
            return FN<bf16, 128>(__VA_ARGS__);                                                 \
        } else {                                                                               \
            if (dh == 32) return FN<float, 32>(__VA_ARGS__);                                   \
            if (dh == 64) return FN<float, 64>(__VA_ARGS__);                                   \
            return FN<float, 128>(__VA_ARGS__);                                                \
        }                                                                                      \
    } while (0)

extern "C" int ucfvit_attention_fwd(const void* qkv, void* out, float* lse, int64_t B, int64_t N, int64_t H, int64_t dh, float scale,
                                    int dtype, void* stream) {
    if (B == 0) return UCFVIT_OK;                      // empty batch (pointers may be NULL)
    UCF_CHECK_ARG(qkv && out && lse, "ucfvit_attention_fwd: null pointer");
    int rc = check_attn_args("ucfvit_attention_fwd", B, N, H, dh, dtype);
    if (rc) return rc;
    UCF_CHECK_ARG(ucf_is_aligned16(qkv) && ucf_is_aligned16(out), "ucfvit_attention_fwd: pointers must be 16-byte aligned");
    if (short_enabled()) {
        rc = ucfvit_attention_short_fwd(qkv, out, lse, B, N, H, dh, scale, dtype, (hipStream_t)stream);
        if (rc == 1) return UCFVIT_OK;
        if (rc < 0) return rc;
    }
    ATTN_DISPATCH(attn_fwd_launch, qkv, out, lse, B, N, H, scale, (hipStream_t)stream);
}

extern "C" int ucfvit_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta_ws,
                                    int64_t B, int64_t N, int64_t H, int64_t dh, float scale, int dtype, void* stream) {
    UCF_CHECK_ARG(qkv && out && dout && lse && dqkv && delta_ws, "ucfvit_attention_bwd: null pointer");
    int rc = check_attn_args("ucfvit_attention_bwd", B, N, H, dh, dtype);
    if (rc) return rc;
    UCF_CHECK_ARG(ucf_is_aligned16(qkv) && ucf_is_aligned16(out) && ucf_is_aligned16(dout) && ucf_is_aligned16(dqkv),
                  "ucfvit_attention_bwd: pointers must be 16-byte aligned");
    ATTN_DISPATCH(attn_bwd_launch, qkv, out, dout, lse, dqkv, delta_ws, B, N, H, scale, (hipStream_t)stream);
}

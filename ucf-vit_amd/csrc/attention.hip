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
#include <string.h>

#include <type_traits>

#include "common.h"

// attention_short.hip: whole-sequence-in-LDS kernels for N <= 256 (return 1 = handled, 0 = not applicable, <0 = error)
int ucfvit_attention_short_fwd(const void* qkv, void* out, float* lse, int64_t B, int64_t N, int64_t H, int64_t dh, float scale, int dtype,
                               hipStream_t s);
int ucfvit_attention_fused_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* cs_partial, int64_t B, int64_t N,
                               int64_t H, int64_t dh, float scale, int dtype, hipStream_t s);
int ucfvit_attention_fused_bwd_applies(int64_t B, int64_t N, int64_t H, int64_t dh, int dtype);


namespace {

constexpr int AT_THREADS = 256;  // 4 waves x 16 lane-columns = 64 queries (or keys) per workgroup
constexpr int KT = 64;           // rows per LDS tile

template <typename T> struct Mma16;
template <> struct Mma16<bf16> {
    typedef bf16x8 frag_t;
    static __device__ __forceinline__ void mma(f32x4& acc, const frag_t& a, const frag_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ frag_t ones() {
        frag_t r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (bf16)1.0f;
        return r;
    }
};
template <> struct Mma16<float> {
    typedef f32x4 frag_t;
    static __device__ __forceinline__ void mma(f32x4& acc, const frag_t& a, const frag_t& b) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
    }
    static __device__ __forceinline__ frag_t ones() { return f32x4{1.f, 1.f, 1.f, 1.f}; }
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

// max(a, b, c) as ONE instruction: hipcc puts a canonicalising v_max_f32 x, x, x in front of every fmaxf of an MFMA result
// (32 extra VALU instructions per key tile of the forward kernel, which is bound by its VALU work)
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float group_max(float v) {  // over the 4 lane groups that share lane&15
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// Operands of the streaming kernels: every matrix is [batch][token][head][DH] with its own token / batch stride (in elements), so the same
// kernels serve self-attention on the packed qkv GEMM output (q = qkv, k = qkv + D, v = qkv + 2D, token stride 3D) and attention of a
// query block against ANOTHER token block's keys / values (ring sequence parallelism: UCF_VIT/fsdp/seq_parallel.py): Nq queries, Nk keys.
template <typename T> struct AttnArgs {
    const T* q;
    const T* k;
    const T* v;
    const T* dout;        // backward
    T* out;               // forward
    void* dq;             // backward outputs: T, or float when the kernel is instantiated with OUTF (+= when accumulate)
    void* dk;
    void* dv;
    float* lse;           // [B][H][Nq], log2 units (written by forward, read by backward)
    const float* delta;   // [B][H][Nq] rowsum(dO * O) (backward)
    int64_t sq, skv, so, sdo, sdq, sdkv;      // token strides
    int64_t bq, bkv, bo, bdo, bdq, bdkv;      // batch strides
    int Nq, Nk, H;
    int accumulate;
};

// ===================================================================================================
// Streaming kernels.  Template parameters shared by all three:
//   QB   = 16-row blocks of the lane-resident index per wave (queries for forward / dQ, keys for dK/dV).  Every K / V (Q / dO) fragment
//          read from LDS feeds QB MFMAs, so QB = 2 halves the LDS bytes per MFMA of the QB = 1 form (which was bound by them).
//   NBUF = LDS buffers per tile stream.  2: the next tile is written into the other buffer while this one is being read: ONE barrier per
//          tile; 1: store / barrier / compute / barrier (fp32 at head dim 128: 64 KB per tile pair).
// A workgroup is 4 waves = 64 QB rows of the resident index; the streamed index advances in tiles of KT = 64 rows.
// ===================================================================================================
template <typename T, int DH, int NBUF> struct Stream {
    static constexpr int PAIR = 2 * AT<T, DH>::TILE_BYTES;      // the two tiles of one step (K+V or Q+dO)
    // top of step t: registers hold tile t+1 (NBUF 2) or tile t (NBUF 1)
    template <typename LoadNext>
    static __device__ __forceinline__ char* begin(char* smem, int t, int ntiles, TileStage<T, DH>& s0, TileStage<T, DH>& s1, int tid, LoadNext load) {
        if constexpr (NBUF == 2) {
            char* cur = smem + (t & 1) * PAIR;
            if (t + 1 < ntiles) {
                char* nxt = smem + ((t + 1) & 1) * PAIR;            // last read in step t-1: every wave has passed that step's barrier
                tile_store<T, DH>(s0, nxt, tid);
                tile_store<T, DH>(s1, nxt + AT<T, DH>::TILE_BYTES, tid);
                if (t + 2 < ntiles) load(t + 2);                    // in flight during this step's MFMAs
            }
            return cur;
        } else {
            __syncthreads();
            tile_store<T, DH>(s0, smem, tid);
            tile_store<T, DH>(s1, smem + AT<T, DH>::TILE_BYTES, tid);
            __syncthreads();
            if (t + 1 < ntiles) load(t + 1);
            return smem;
        }
    }
    static __device__ __forceinline__ void end() {
        if constexpr (NBUF == 2) __syncthreads();
    }
    // before the loop: tile 0 into LDS (NBUF 2) and the first register tile loaded
    template <typename LoadNext>
    static __device__ __forceinline__ void prime(char* smem, int ntiles, TileStage<T, DH>& s0, TileStage<T, DH>& s1, int tid, LoadNext load) {
        load(0);
        if constexpr (NBUF == 2) {
            tile_store<T, DH>(s0, smem, tid);
            tile_store<T, DH>(s1, smem + AT<T, DH>::TILE_BYTES, tid);
            if (ntiles > 1) load(1);
            __syncthreads();
        }
    }
};

// ===================================================================================================
// forward
// ===================================================================================================
template <typename T, int DH, int QB, int NBUF>
__global__ __launch_bounds__(AT_THREADS) void attn_fwd_kernel(const AttnArgs<T> a, float scale_log2e) {
    typedef AT<T, DH> A;
    typedef typename A::frag_t frag_t;
    typedef Stream<T, DH, NBUF> ST;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int N = a.Nk, Nq = a.Nq, H = a.H;       // N: keys streamed; Nq: queries resident on the lanes
    const int64_t rs = a.skv;
    const T* qbase = a.q + b * a.bq + h * DH;
    const T* kbase = a.k + b * a.bkv + h * DH;
    const T* vbase = a.v + b * a.bkv + h * DH;
    T* __restrict__ out = a.out;
    float* __restrict__ lse = a.lse;
    const int q0 = blockIdx.x * (64 * QB) + wave * (16 * QB) + li;     // query of block qb: q0 + 16 qb

    frag_t qf[QB][A::NCH];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int c = 0; c < A::NCH; ++c) qf[qb][c] = frag_global<T, DH>(qbase, a.sq, q0 + 16 * qb, Nq, c, lane);

    f32x4 o[QB][A::NDB];
    // the row sums of P ride on the matrix pipe: an all-ones A fragment times Pᵀ gives, in EVERY accumulator row, the sum over the tile's
    // keys of the (rounded) probabilities that also multiply V — 2 QB extra MFMAs per tile (the pipe has slack) instead of 16 QB v_add
    // and the final cross-group sum
    f32x4 lacc[QB];
    float m[QB];
    const frag_t ones = Mma16<T>::ones();
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int d = 0; d < A::NDB; ++d) o[qb][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        lacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
        m[qb] = -INFINITY;
    }

    const int ntiles = (N + KT - 1) / KT;
    TileStage<T, DH> sk, sv;
    auto load = [&](int t) {
        tile_load<T, DH>(sk, kbase, rs, t * KT, N, tid);
        tile_load<T, DH>(sv, vbase, rs, t * KT, N, tid);
    };
    ST::prime(smem, ntiles, sk, sv, tid, load);
    for (int kt = 0; kt < ntiles; ++kt) {
        const char* ldsK = ST::begin(smem, kt, ntiles, sk, sv, tid, load);
        const char* ldsV = ldsK + A::TILE_BYTES;
        // Sᵀ[key][q] for the tile's 4 key blocks: each K fragment feeds the QB query blocks
        f32x4 s[QB][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) s[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < A::NCH; ++c) {
                const frag_t kf = frag_row<T, DH>(ldsK, kb, c, lane);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) Mma16<T>::mma(s[qb][kb], kf, qf[qb][c]);
            }
        }
        const bool ragged = (kt + 1) * KT > N;          // only the last tile can hold keys beyond N
        frag_t pf[QB][A::NRC];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (__builtin_expect(ragged, 0)) {
                asm volatile("" ::: "memory");          // keep this a branch: as selects it costs 64 VALU instructions in EVERY tile
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kt * KT + kb * 16 + 4 * g + r >= N) s[qb][kb][r] = -INFINITY;
            }
            // running maximum in units of log2: the scale is positive, so max(scale * s) = scale * max(s)
            float mx = max3f(s[qb][0][0], s[qb][0][1], s[qb][0][2]);
            mx = max3f(mx, s[qb][0][3], s[qb][1][0]);
            mx = max3f(mx, s[qb][1][1], s[qb][1][2]);
            mx = max3f(mx, s[qb][1][3], s[qb][2][0]);
            mx = max3f(mx, s[qb][2][1], s[qb][2][2]);
            mx = max3f(mx, s[qb][2][3], s[qb][3][0]);
            mx = max3f(mx, s[qb][3][1], s[qb][3][2]);
            mx = fmaxf(mx, s[qb][3][3]);
            const float m_new = fmaxf(m[qb], group_max(mx) * scale_log2e);  // finite: every tile holds at least one valid key
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[qb][kb][r] = __builtin_amdgcn_exp2f(fmaf(s[qb][kb][r], scale_log2e, -m_new));
            if (__any(m_new > m[qb])) {                 // exact: where no row's maximum moved the factor is exp2(0) = 1
                const float alpha = __builtin_amdgcn_exp2f(m[qb] - m_new);
#pragma unroll
                for (int r = 0; r < 4; ++r) lacc[qb][r] *= alpha;
#pragma unroll
                for (int d = 0; d < A::NDB; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qb][d][r] *= alpha;
                m[qb] = m_new;
            }
#pragma unroll
            for (int rc = 0; rc < A::NRC; ++rc) {
                pf[qb][rc] = frag_from_acc<T>(s[qb], rc);
                Mma16<T>::mma(lacc[qb], ones, pf[qb][rc]);
            }
        }
        // Oᵀ[d][q] += Vᵀ[d][key] · Pᵀ[key][q]: each Vᵀ fragment feeds the QB query blocks
#pragma unroll
        for (int rc = 0; rc < A::NRC; ++rc)
#pragma unroll
            for (int d = 0; d < A::NDB; ++d) {
                const frag_t vf = frag_tr<T, DH>(ldsV, rc, d, lane);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) Mma16<T>::mma(o[qb][d], vf, pf[qb][rc]);
            }
        ST::end();
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float lt = lacc[qb][0];          // every accumulator row holds the full row sum of its query column
        const int q = q0 + 16 * qb;
        if (q < Nq) {
            const float inv = 1.f / lt;
            T* op = out + b * a.bo + q * a.so + h * DH;
#pragma unroll
            for (int d = 0; d < A::NDB; ++d) {
                Vec4<T> v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v.set(r, o[qb][d][r] * inv);
                *reinterpret_cast<Vec4<T>*>(op + d * 16 + 4 * g) = v;
            }
            if (g == 0) lse[(b * H + h) * (int64_t)Nq + q] = m[qb] + log2f(lt);
        }
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
template <typename T, int DH, int QB, int NBUF, bool OUTF>
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dq_kernel(const AttnArgs<T> a, float scale, float scale_log2e) {
    typedef AT<T, DH> A;
    typedef typename A::frag_t frag_t;
    typedef Stream<T, DH, NBUF> ST;
    typedef typename std::conditional<OUTF, float, T>::type OT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int N = a.Nk, Nq = a.Nq, H = a.H;
    const int64_t rs = a.skv;
    const T* qbase = a.q + b * a.bq + h * DH;
    const T* kbase = a.k + b * a.bkv + h * DH;
    const T* vbase = a.v + b * a.bkv + h * DH;
    const T* dobase = a.dout + b * a.bdo + h * DH;
    const float* __restrict__ lse = a.lse;
    const float* __restrict__ delta = a.delta;
    const int q0 = blockIdx.x * (64 * QB) + wave * (16 * QB) + li;

    frag_t qf[QB][A::NCH], dof[QB][A::NCH];
    float my_lse[QB], my_delta[QB];
    f32x4 dq[QB][A::NDB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q0 + 16 * qb;
#pragma unroll
        for (int c = 0; c < A::NCH; ++c) {
            qf[qb][c] = frag_global<T, DH>(qbase, a.sq, q, Nq, c, lane);
            dof[qb][c] = frag_global<T, DH>(dobase, a.sdo, q, Nq, c, lane);
        }
        my_lse[qb] = q < Nq ? lse[(b * H + h) * (int64_t)Nq + q] : 0.f;
        my_delta[qb] = q < Nq ? delta[(b * H + h) * (int64_t)Nq + q] : 0.f;
#pragma unroll
        for (int d = 0; d < A::NDB; ++d) dq[qb][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int ntiles = (N + KT - 1) / KT;
    TileStage<T, DH> sk, sv;
    auto load = [&](int t) {
        tile_load<T, DH>(sk, kbase, rs, t * KT, N, tid);
        tile_load<T, DH>(sv, vbase, rs, t * KT, N, tid);
    };
    ST::prime(smem, ntiles, sk, sv, tid, load);
    for (int kt = 0; kt < ntiles; ++kt) {
        const char* ldsK = ST::begin(smem, kt, ntiles, sk, sv, tid, load);
        const char* ldsV = ldsK + A::TILE_BYTES;
        f32x4 ds[QB][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 s[QB], dp[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                s[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
                dp[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int c = 0; c < A::NCH; ++c) {
                const frag_t kf = frag_row<T, DH>(ldsK, kb, c, lane), vf = frag_row<T, DH>(ldsV, kb, c, lane);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    Mma16<T>::mma(s[qb], kf, qf[qb][c]);     // Sᵀ[key][q]
                    Mma16<T>::mma(dp[qb], vf, dof[qb][c]);   // dPᵀ[key][q]
                }
            }
            // keys beyond N need no mask here: their K rows were zero-filled by tile_load, so whatever dS they get multiplies zeros in
            // dQ += dS K (and it is finite: S = dP = 0 for them)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[qb][r], scale_log2e, -my_lse[qb]));
                    ds[qb][kb][r] = p * (dp[qb][r] - my_delta[qb]);
                }
        }
        // dQᵀ[d][q] += Kᵀ[d][key] · dSᵀ[key][q]
        frag_t f[QB][A::NRC];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int rc = 0; rc < A::NRC; ++rc) f[qb][rc] = frag_from_acc<T>(ds[qb], rc);
#pragma unroll
        for (int rc = 0; rc < A::NRC; ++rc)
#pragma unroll
            for (int d = 0; d < A::NDB; ++d) {
                const frag_t kt_f = frag_tr<T, DH>(ldsK, rc, d, lane);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) Mma16<T>::mma(dq[qb][d], kt_f, f[qb][rc]);
            }
        ST::end();
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q0 + 16 * qb;
        if (q < Nq) {
            OT* op = reinterpret_cast<OT*>(a.dq) + b * a.bdq + q * a.sdq + h * DH;
#pragma unroll
            for (int d = 0; d < A::NDB; ++d) {
                Vec4<OT> v;
                if (a.accumulate) v = *reinterpret_cast<const Vec4<OT>*>(op + d * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) v.set(r, dq[qb][d][r] * scale + (a.accumulate ? v.get(r) : 0.f));
                *reinterpret_cast<Vec4<OT>*>(op + d * 16 + 4 * g) = v;
            }
        }
    }
}

// ===================================================================================================
// backward, dK / dV: key on the lane, loop over query tiles
// ===================================================================================================
template <typename T, int DH, int QB, int NBUF, bool OUTF>
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dkv_kernel(const AttnArgs<T> a, float scale, float scale_log2e) {
    typedef AT<T, DH> A;
    typedef typename A::frag_t frag_t;
    typedef Stream<T, DH, NBUF> ST;
    typedef typename std::conditional<OUTF, float, T>::type OT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // per-query constants of the streamed tile, one [2][64] float pair per pipeline buffer, behind the tile buffers
    float* ldsRow = reinterpret_cast<float*>(smem + NBUF * ST::PAIR);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int N = a.Nq, Nk = a.Nk, H = a.H;        // N: queries streamed; Nk: keys resident on the lanes
    const int64_t rs = a.sq;
    const int64_t D = a.sdo;                       // token stride of dO
    const T* qbase = a.q + b * a.bq + h * DH;
    const T* kbase = a.k + b * a.bkv + h * DH;
    const T* vbase = a.v + b * a.bkv + h * DH;
    const T* dobase = a.dout + b * a.bdo + h * DH;
    const float* lse_bh = a.lse + (b * H + h) * (int64_t)N;
    const float* delta_bh = a.delta + (b * H + h) * (int64_t)N;
    const int key0 = blockIdx.x * (64 * QB) + wave * (16 * QB) + li;

    frag_t kf[QB][A::NCH], vf[QB][A::NCH];
    f32x4 dk[QB][A::NDB], dv[QB][A::NDB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int c = 0; c < A::NCH; ++c) {
            kf[qb][c] = frag_global<T, DH>(kbase, a.skv, key0 + 16 * qb, Nk, c, lane);
            vf[qb][c] = frag_global<T, DH>(vbase, a.skv, key0 + 16 * qb, Nk, c, lane);
        }
#pragma unroll
        for (int d = 0; d < A::NDB; ++d) {
            dk[qb][d] = f32x4{0.f, 0.f, 0.f, 0.f};
            dv[qb][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int ntiles = (N + KT - 1) / KT;
    TileStage<T, DH> sq, sdo;
    float r_lse = 0.f, r_delta = 0.f;       // threads 0-63: constants of the register tile's queries
    auto load = [&](int t) {
        tile_load<T, DH>(sq, qbase, rs, t * KT, N, tid);
        tile_load<T, DH>(sdo, dobase, D, t * KT, N, tid);
        if (tid < KT) {
            const int qq = t * KT + tid;
            r_lse = qq < N ? lse_bh[qq] : INFINITY;      // +inf -> P = 0 for padding queries
            r_delta = qq < N ? delta_bh[qq] : 0.f;
        }
    };
    // the row constants travel with their tile: written to LDS where the tile is written
    if constexpr (NBUF == 2) {
        load(0);
        tile_store<T, DH>(sq, smem, tid);
        tile_store<T, DH>(sdo, smem + A::TILE_BYTES, tid);
        if (tid < KT) {
            ldsRow[tid] = r_lse;
            ldsRow[KT + tid] = r_delta;
        }
        if (ntiles > 1) load(1);
        __syncthreads();
    } else {
        load(0);
    }
    for (int qt = 0; qt < ntiles; ++qt) {
        const char* ldsQ;
        const float* rowc;
        if constexpr (NBUF == 2) {
            ldsQ = smem + (qt & 1) * ST::PAIR;
            rowc = ldsRow + (qt & 1) * 2 * KT;
            if (qt + 1 < ntiles) {
                char* nxt = smem + ((qt + 1) & 1) * ST::PAIR;
                tile_store<T, DH>(sq, nxt, tid);
                tile_store<T, DH>(sdo, nxt + A::TILE_BYTES, tid);
                if (tid < KT) {
                    float* rn = ldsRow + ((qt + 1) & 1) * 2 * KT;
                    rn[tid] = r_lse;
                    rn[KT + tid] = r_delta;
                }
                if (qt + 2 < ntiles) load(qt + 2);
            }
        } else {
            __syncthreads();
            tile_store<T, DH>(sq, smem, tid);
            tile_store<T, DH>(sdo, smem + A::TILE_BYTES, tid);
            if (tid < KT) {
                ldsRow[tid] = r_lse;
                ldsRow[KT + tid] = r_delta;
            }
            __syncthreads();
            if (qt + 1 < ntiles) load(qt + 1);
            ldsQ = smem;
            rowc = ldsRow;
        }
        const char* ldsDO = ldsQ + A::TILE_BYTES;
        f32x4 pm[QB][4], ds[QB][4];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {      // 16-query blocks of the streamed tile
            f32x4 s[QB], dp[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                s[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
                dp[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int c = 0; c < A::NCH; ++c) {
                const frag_t qfr = frag_row<T, DH>(ldsQ, tb, c, lane), dofr = frag_row<T, DH>(ldsDO, tb, c, lane);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    Mma16<T>::mma(s[qb], qfr, kf[qb][c]);      // S[q][key]
                    Mma16<T>::mma(dp[qb], dofr, vf[qb][c]);    // dP[q][key]
                }
            }
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(rowc + tb * 16 + 4 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(rowc + KT + tb * 16 + 4 * g);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[qb][r], scale_log2e, -l4[r]));
                    pm[qb][tb][r] = p;
                    ds[qb][tb][r] = p * (dp[qb][r] - d4[r]);
                }
        }
        frag_t fp[QB][A::NRC], fs[QB][A::NRC];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int rc = 0; rc < A::NRC; ++rc) {
                fp[qb][rc] = frag_from_acc<T>(pm[qb], rc);
                fs[qb][rc] = frag_from_acc<T>(ds[qb], rc);
            }
#pragma unroll
        for (int rc = 0; rc < A::NRC; ++rc)
#pragma unroll
            for (int d = 0; d < A::NDB; ++d) {
                const frag_t dot = frag_tr<T, DH>(ldsDO, rc, d, lane), qt_f = frag_tr<T, DH>(ldsQ, rc, d, lane);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    Mma16<T>::mma(dv[qb][d], dot, fp[qb][rc]);   // dVᵀ[d][key] += dOᵀ[d][q]·P[q][key]
                    Mma16<T>::mma(dk[qb][d], qt_f, fs[qb][rc]);  // dKᵀ[d][key] += Qᵀ[d][q]·dS[q][key]
                }
            }
        ST::end();
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int key = key0 + 16 * qb;
        if (key < Nk) {
            OT* kp = reinterpret_cast<OT*>(a.dk) + b * a.bdkv + key * a.sdkv + h * DH;
            OT* vp = reinterpret_cast<OT*>(a.dv) + b * a.bdkv + key * a.sdkv + h * DH;
#pragma unroll
            for (int d = 0; d < A::NDB; ++d) {
                Vec4<OT> ka, va;
                if (a.accumulate) {
                    ka = *reinterpret_cast<const Vec4<OT>*>(kp + d * 16 + 4 * g);
                    va = *reinterpret_cast<const Vec4<OT>*>(vp + d * 16 + 4 * g);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ka.set(r, dk[qb][d][r] * scale + (a.accumulate ? ka.get(r) : 0.f));
                    va.set(r, dv[qb][d][r] + (a.accumulate ? va.get(r) : 0.f));
                }
                *reinterpret_cast<Vec4<OT>*>(kp + d * 16 + 4 * g) = ka;
                *reinterpret_cast<Vec4<OT>*>(vp + d * 16 + 4 * g) = va;
            }
        }
    }
}

// geometry of the streaming kernels per element type: bf16 takes two 16-row blocks per wave and two LDS buffers per stream
template <typename T, int DH> struct Geo {
    static constexpr int QB = sizeof(T) == 2 ? 2 : 1;
    static constexpr int NBUF = sizeof(T) == 2 ? 2 : 1;
    static constexpr size_t SMEM = (size_t)NBUF * 2 * AT<T, DH>::TILE_BYTES;
    static constexpr size_t SMEM_DKV = SMEM + (size_t)NBUF * 2 * KT * sizeof(float);
};

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
int attn_fwd_launch_args(const AttnArgs<T>& a, int64_t B, float scale, hipStream_t s) {
    typedef Geo<T, DH> G;
    const dim3 grid((unsigned)((a.Nq + 64 * G::QB - 1) / (64 * G::QB)), (unsigned)a.H, (unsigned)B);
    auto kern = attn_fwd_kernel<T, DH, G::QB, G::NBUF>;
    if (int rc = allow_big_lds(kern, G::SMEM)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(AT_THREADS), G::SMEM, s, a, scale * 1.44269504088896340736f);
    UCF_LAUNCH_CHECK("ucfvit_attention_fwd");
    return UCFVIT_OK;
}

// delta + dQ + dK/dV launches; `out` / `a.dout` are contiguous [B][Nq][H][DH]
template <typename T, int DH, bool OUTF>
int attn_bwd_launch_args(AttnArgs<T> a, const void* out, float* delta, int64_t B, float scale, hipStream_t s) {
    typedef Geo<T, DH> G;
    const int64_t nd = B * a.Nq * a.H * (DH / (16 / (int64_t)sizeof(T)));
    hipLaunchKernelGGL((attn_delta_kernel<T, DH>), dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, (const T*)out, a.dout, delta, B, a.Nq,
                       a.H);
    UCF_LAUNCH_CHECK("ucfvit_attention_bwd(delta)");
    a.delta = delta;
    const float sl2 = scale * 1.44269504088896340736f;
    auto k_dq = attn_bwd_dq_kernel<T, DH, G::QB, G::NBUF, OUTF>;
    auto k_dkv = attn_bwd_dkv_kernel<T, DH, G::QB, G::NBUF, OUTF>;
    if (int rc = allow_big_lds(k_dq, G::SMEM)) return rc;
    if (int rc = allow_big_lds(k_dkv, G::SMEM_DKV)) return rc;
    const dim3 grid_q((unsigned)((a.Nq + 64 * G::QB - 1) / (64 * G::QB)), (unsigned)a.H, (unsigned)B);
    const dim3 grid_k((unsigned)((a.Nk + 64 * G::QB - 1) / (64 * G::QB)), (unsigned)a.H, (unsigned)B);
    hipLaunchKernelGGL(k_dq, grid_q, dim3(AT_THREADS), G::SMEM, s, a, scale, sl2);
    UCF_LAUNCH_CHECK("ucfvit_attention_bwd(dq)");
    hipLaunchKernelGGL(k_dkv, grid_k, dim3(AT_THREADS), G::SMEM_DKV, s, a, scale, sl2);
    UCF_LAUNCH_CHECK("ucfvit_attention_bwd(dkv)");
    return UCFVIT_OK;
}

// self-attention on the packed qkv GEMM output [B][N][3][H][DH]
template <typename T> AttnArgs<T> self_args(const void* qkv, int64_t N, int64_t H, int64_t DH) {
    AttnArgs<T> a;
    memset(&a, 0, sizeof(a));
    const int64_t D = H * DH;
    a.q = (const T*)qkv;
    a.k = a.q + D;
    a.v = a.q + 2 * D;
    a.sq = a.skv = 3 * D;
    a.bq = a.bkv = N * 3 * D;
    a.so = a.sdo = D;
    a.bo = a.bdo = N * D;
    a.Nq = a.Nk = (int)N;
    a.H = (int)H;
    return a;
}

template <typename T, int DH>
int attn_fwd_launch(const void* qkv, void* out, float* lse, int64_t B, int64_t N, int64_t H, float scale, hipStream_t s) {
    AttnArgs<T> a = self_args<T>(qkv, N, H, DH);
    a.out = (T*)out;
    a.lse = lse;
    return attn_fwd_launch_args<T, DH>(a, B, scale, s);
}

template <typename T, int DH>
int attn_bwd_launch(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta, float* cs_partial, int64_t B,
                    int64_t N, int64_t H, float scale, hipStream_t s) {
    // the FUSED backward (one launch, operands read once, delta = rowsum(dO o O) taken inside) is the default where it applies
    {
        const int rc = ucfvit_attention_fused_bwd(qkv, out, dout, lse, dqkv, cs_partial, B, N, H, DH, scale, sizeof(T) == 2 ? UCFVIT_BF16 : UCFVIT_F32, s);
        if (rc == 1) return UCFVIT_OK;
        if (rc < 0) return rc;
    }
    if (cs_partial) {
        ucfvit_set_error("ucfvit_attention_bwd_colsum: this shape runs the streaming kernels, which produce no column sums (ask "
                         "ucfvit_attention_bwd_colsum_supported first)");
        return UCFVIT_ERR_UNSUPPORTED;
    }
    AttnArgs<T> a = self_args<T>(qkv, N, H, DH);
    const int64_t D = H * DH;
    a.dout = (const T*)dout;
    a.lse = const_cast<float*>(lse);
    a.dq = dqkv;
    a.dk = (T*)dqkv + D;
    a.dv = (T*)dqkv + 2 * D;
    a.sdq = a.sdkv = 3 * D;
    a.bdq = a.bdkv = N * 3 * D;
    return attn_bwd_launch_args<T, DH, false>(a, out, delta, B, scale, s);
}

// ---- attention of a query block against another block's keys / values (ring sequence parallelism) ------------------------------------
template <typename T, int DH>
int attn_cross_fwd_launch(const void* q, const void* k, const void* v, void* out, float* lse, int64_t B, int64_t Nq, int64_t Nk, int64_t H,
                          int64_t ldq, int64_t ldkv, float scale, hipStream_t s) {
    AttnArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.q = (const T*)q;
    a.k = (const T*)k;
    a.v = (const T*)v;
    a.out = (T*)out;
    a.lse = lse;
    a.sq = ldq;
    a.skv = ldkv;
    a.bq = Nq * ldq;
    a.bkv = Nk * ldkv;
    a.so = H * DH;
    a.bo = Nq * H * DH;
    a.Nq = (int)Nq;
    a.Nk = (int)Nk;
    a.H = (int)H;
    return attn_fwd_launch_args<T, DH>(a, B, scale, s);
}

template <typename T, int DH>
int attn_cross_bwd_launch(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse, float* dq, float* dk,
                          float* dv, float* delta, int64_t B, int64_t Nq, int64_t Nk, int64_t H, int64_t ldq, int64_t ldkv, float scale,
                          int accumulate, hipStream_t s) {
    AttnArgs<T> a;
    memset(&a, 0, sizeof(a));
    const int64_t D = H * DH;
    a.q = (const T*)q;
    a.k = (const T*)k;
    a.v = (const T*)v;
    a.dout = (const T*)dout;
    a.lse = const_cast<float*>(lse);
    a.dq = dq;
    a.dk = dk;
    a.dv = dv;
    a.sq = ldq;
    a.skv = ldkv;
    a.bq = Nq * ldq;
    a.bkv = Nk * ldkv;
    a.so = a.sdo = a.sdq = a.sdkv = D;
    a.bo = a.bdo = a.bdq = Nq * D;
    a.bdkv = Nk * D;
    a.Nq = (int)Nq;
    a.Nk = (int)Nk;
    a.H = (int)H;
    a.accumulate = accumulate;
    return attn_bwd_launch_args<T, DH, true>(a, out, delta, B, scale, s);
}

// online merge of two partial attention results over disjoint key sets (log2-domain log-sum-exp):
//   lse' = log2(2^lse_acc + 2^lse_part),  o' = o_acc 2^(lse_acc - lse') + o_part 2^(lse_part - lse');  first: plain copy
template <typename T, int DH>
__global__ void attn_merge_kernel(float* __restrict__ o_acc, float* __restrict__ lse_acc, const T* __restrict__ o_part,
                                  const float* __restrict__ lse_part, int64_t B, int Nq, int H, int first) {
    constexpr int V = DH / 4;                                   // 4-element pieces per (token, head)
    const int64_t i_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over (b, q, h, piece)
    const bool valid = i_raw < B * Nq * H * V;                  // (a predicate, not an early return: every thread reaches the barrier below)
    const int64_t i = valid ? i_raw : 0;
    const int piece = (int)(i % V);
    const int64_t bqh = i / V;
    const int h = (int)(bqh % H);
    const int64_t bq = bqh / H;
    const int q = (int)(bq % Nq);
    const int64_t b = bq / Nq;
    const int64_t li = (b * H + h) * (int64_t)Nq + q;
    const float lp = lse_part[li];
    const Vec4<T> p = *reinterpret_cast<const Vec4<T>*>(o_part + i * 4);
    f32x4 r;
    float ln = lp;
    if (first) {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = p.get(e);
    } else {
        const float la = lse_acc[li];
        const float mx = fmaxf(la, lp);
        ln = mx + log2f(__builtin_amdgcn_exp2f(la - mx) + __builtin_amdgcn_exp2f(lp - mx));
        const float wa = __builtin_amdgcn_exp2f(la - ln), wp = __builtin_amdgcn_exp2f(lp - ln);
        const f32x4 o = *reinterpret_cast<const f32x4*>(o_acc + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = o[e] * wa + p.get(e) * wp;
    }
    if (valid) *reinterpret_cast<f32x4*>(o_acc + i * 4) = r;
    __syncthreads();      // every piece of a (token, head) sits in one workgroup (V divides 256): all have read lse_acc before it changes
    if (valid && piece == 0) lse_acc[li] = ln;
}

template <typename T, int DH>
int attn_merge_launch(float* o_acc, float* lse_acc, const void* o_part, const float* lse_part, int64_t B, int64_t Nq, int64_t H, int first,
                      hipStream_t s) {
    const int64_t n = B * Nq * H * (DH / 4);
    hipLaunchKernelGGL((attn_merge_kernel<T, DH>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, o_acc, lse_acc, (const T*)o_part, lse_part, B,
                       (int)Nq, (int)H, first);
    UCF_LAUNCH_CHECK("ucfvit_attention_merge");
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
    {
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
    ATTN_DISPATCH(attn_bwd_launch, qkv, out, dout, lse, dqkv, delta_ws, (float*)nullptr, B, N, H, scale, (hipStream_t)stream);
}

extern "C" int ucfvit_attention_bwd_colsum_supported(int64_t B, int64_t N, int64_t H, int64_t dh, int dtype) {
    return ucfvit_attention_fused_bwd_applies(B, N, H, dh, dtype);
}

extern "C" int ucfvit_attention_bwd_colsum(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta_ws,
                                           float* colsum_partial, int64_t B, int64_t N, int64_t H, int64_t dh, float scale, int dtype, void* stream) {
    UCF_CHECK_ARG(qkv && out && dout && lse && dqkv && delta_ws && colsum_partial, "ucfvit_attention_bwd_colsum: null pointer");
    int rc = check_attn_args("ucfvit_attention_bwd_colsum", B, N, H, dh, dtype);
    if (rc) return rc;
    UCF_CHECK_ARG(ucf_is_aligned16(qkv) && ucf_is_aligned16(out) && ucf_is_aligned16(dout) && ucf_is_aligned16(dqkv),
                  "ucfvit_attention_bwd_colsum: pointers must be 16-byte aligned");
    ATTN_DISPATCH(attn_bwd_launch, qkv, out, dout, lse, dqkv, delta_ws, colsum_partial, B, N, H, scale, (hipStream_t)stream);
}

extern "C" int ucfvit_attention_cross_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int64_t B, int64_t Nq, int64_t Nk,
                                          int64_t H, int64_t dh, int64_t ldq, int64_t ldkv, float scale, int dtype, void* stream) {
    if (B == 0) return UCFVIT_OK;
    UCF_CHECK_ARG(q && k && v && out && lse, "ucfvit_attention_cross_fwd: null pointer");
    int rc = check_attn_args("ucfvit_attention_cross_fwd", B, Nq, H, dh, dtype);
    if (rc) return rc;
    UCF_CHECK_ARG(Nk > 0 && Nk < (1ll << 30) && ldq >= H * dh && ldkv >= H * dh && ldq % 8 == 0 && ldkv % 8 == 0,
                  "ucfvit_attention_cross_fwd: need Nk > 0 and row strides >= H*dh, multiples of 8 elements");
    UCF_CHECK_ARG(ucf_is_aligned16(q) && ucf_is_aligned16(k) && ucf_is_aligned16(v) && ucf_is_aligned16(out),
                  "ucfvit_attention_cross_fwd: pointers must be 16-byte aligned");
    ATTN_DISPATCH(attn_cross_fwd_launch, q, k, v, out, lse, B, Nq, Nk, H, ldq, ldkv, scale, (hipStream_t)stream);
}

extern "C" int ucfvit_attention_cross_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                                          float* dq, float* dk, float* dv, float* delta_ws, int64_t B, int64_t Nq, int64_t Nk, int64_t H,
                                          int64_t dh, int64_t ldq, int64_t ldkv, float scale, int accumulate, int dtype, void* stream) {
    if (B == 0) return UCFVIT_OK;
    UCF_CHECK_ARG(q && k && v && out && dout && lse && dq && dk && dv && delta_ws, "ucfvit_attention_cross_bwd: null pointer");
    int rc = check_attn_args("ucfvit_attention_cross_bwd", B, Nq, H, dh, dtype);
    if (rc) return rc;
    UCF_CHECK_ARG(Nk > 0 && Nk < (1ll << 30) && ldq >= H * dh && ldkv >= H * dh && ldq % 8 == 0 && ldkv % 8 == 0,
                  "ucfvit_attention_cross_bwd: need Nk > 0 and row strides >= H*dh, multiples of 8 elements");
    UCF_CHECK_ARG(ucf_is_aligned16(q) && ucf_is_aligned16(k) && ucf_is_aligned16(v) && ucf_is_aligned16(out) && ucf_is_aligned16(dout) &&
                      ucf_is_aligned16(dq) && ucf_is_aligned16(dk) && ucf_is_aligned16(dv),
                  "ucfvit_attention_cross_bwd: pointers must be 16-byte aligned");
    ATTN_DISPATCH(attn_cross_bwd_launch, q, k, v, out, dout, lse, dq, dk, dv, delta_ws, B, Nq, Nk, H, ldq, ldkv, scale, accumulate,
                  (hipStream_t)stream);
}

extern "C" int ucfvit_attention_merge(float* o_acc, float* lse_acc, const void* o_part, const float* lse_part, int64_t B, int64_t Nq, int64_t H,
                                      int64_t dh, int first, int dtype, void* stream) {
    if (B == 0) return UCFVIT_OK;
    UCF_CHECK_ARG(o_acc && lse_acc && o_part && lse_part, "ucfvit_attention_merge: null pointer");
    int rc = check_attn_args("ucfvit_attention_merge", B, Nq, H, dh, dtype);
    if (rc) return rc;
    UCF_CHECK_ARG(ucf_is_aligned16(o_acc) && (((uintptr_t)o_part) & 7) == 0, "ucfvit_attention_merge: o_acc 16-byte, o_part 8-byte aligned");
    ATTN_DISPATCH(attn_merge_launch, o_acc, lse_acc, o_part, lse_part, B, Nq, H, first, (hipStream_t)stream);
}

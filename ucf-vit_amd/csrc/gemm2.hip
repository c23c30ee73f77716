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
//               a ragged last K-tile (K % 64 != 0, K % 8 == 0) is zero-filled by pointing the out-of-range DMA lanes at a zero page.
//   split-K   : gridDim.y K-slices write fp32 partial matrices to a caller workspace; a second kernel sums them in a fixed
//               order (deterministic) and applies accumulate.  Used when the output has too few tiles to fill 256 CUs.
#include <stdlib.h>

#include "common.h"

int ucfvit_gemm_stagger_try(const ucfvit_gemm_desc* d, hipStream_t s);   // gemm_stagger.hip

namespace {

// C / aux tiles are written once and not read again by this kernel: stored with the non-temporal hint, the dirty lines leave L2 as the
// K loop's fills need the ways instead of in one write-back burst one L2 turnover later (4 MB of C per XCD and tile round).
#ifndef UCFVIT_GEMM_STORE_TEMPORAL
__device__ __forceinline__ void store_out16(bf16* p, const Vec16<bf16>& o) { __builtin_nontemporal_store(o.v, reinterpret_cast<bf16x8*>(p)); }
#else
__device__ __forceinline__ void store_out16(bf16* p, const Vec16<bf16>& o) { *reinterpret_cast<Vec16<bf16>*>(p) = o; }
#endif


constexpr int BK2 = 64;

__device__ __forceinline__ int kc_off2(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }
__device__ __forceinline__ int ks_swz2(int krow) { return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 5; }

template <int BR>  // BR = tile extent along the operand's row/col index (128 or 256)
__device__ __forceinline__ int ks_off2(int krow, int colbyte) { return krow * (BR * 2) + (colbyte ^ ks_swz2(krow)); }

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Source of the zero fill for the ragged last K-tile: lanes whose contraction index is >= K read this page instead of the
// operand (a DMA lane cannot be predicated off without leaving stale LDS bytes).
__device__ __attribute__((aligned(16))) unsigned char g_zero_page[1024];

// Issue the DMA loads of one operand K-tile.  Every wave issues NPW wave-instructions of 1 KiB.
template <int LAYOUT, int BR, int NWAVES>
__device__ __forceinline__ void issue_tile(const bf16* __restrict__ base, int64_t ld, int r0, int k0, int R, char* lds, int wave, int lane,
                                           int krem = BK2) {
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
            if (gslot * 8 >= krem) src = reinterpret_cast<const bf16*>(g_zero_page) + lane * 8;
        } else {
            constexpr int RB = BR * 2;          // bytes per k-row
            constexpr int KPP = 1024 / RB;      // k-rows per piece
            const int krow = idx * KPP + (lane * 16) / RB;
            const int pbyte = (lane * 16) % RB;
            const int col = (pbyte ^ ks_swz2(krow)) >> 1;
            int gc = r0 + col;
            gc = gc <= R - 8 ? gc : R - 8;
            src = base + (int64_t)(k0 + krow) * ld + gc;
            if (krow >= krem) src = reinterpret_cast<const bf16*>(g_zero_page) + lane * 8;
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

// KS fragment reads with ONE per-lane address register: for a fragment whose first row/col is rbase (a multiple of 16) the byte
// offset inside the K-tile image is  krow * 2BR + ((2 rbase + 8p) ^ swz(krow)),  krow = 32c + 8g + q (+4),  q = (lane & 15) >> 2,
// p = lane & 3.  swz depends on the lane only (krow & 3 = q, (krow >> 3) & 1 = g & 1) and occupies bits 5-7, 8p bits 3-4, 2 rbase
// bits 5-8 and krow * 2BR bits >= 8 (BR = 128) or 9 (BR = 256), so
//     offset = ks_lane_base(lane) ^ (2 rbase)  +  c * 64 BR  (+ 8 BR for the second half),
// i.e. one v_xor per fragment and immediates for the rest.  Computing every fragment's address separately costs 12 loop-invariant
// VGPRs in the 256x256 kernel, which is what made its KS x KS instantiation spill inside the K loop.
template <int BR> __device__ __forceinline__ int ks_lane_base(int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    return (8 * g + q) * (BR * 2) + ((8 * p) | ((q | ((g & 1) << 2)) << 5));
}
template <int BR> __device__ __forceinline__ bf16x8 load_frag_ks(const char* lds, int lane_base, int rbase, int c) {
    const char* a0 = lds + (lane_base ^ (2 * rbase)) + c * (64 * BR);
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, a0));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, a0 + 8 * BR));
    short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

// The same read issued from inline asm.  hipcc treats the ds_read_tr builtin as a possible reader of every LDS-DMA in flight and
// puts `s_waitcnt vmcnt(0)` in front of it: with a KS operand each K-step then waits for the prefetch it has just issued (the
// K loops of every KS instantiation had that wait; the KC ones, whose fragment reads are plain loads, did not).  asm reads are
// invisible to that bookkeeping; their own completion is waited for by ks_wait() (lgkmcnt) before the first MFMA that uses them.
struct KsFrag {
    u32x2 lo, hi;
};
template <int BR, int C> __device__ __forceinline__ void load_frag_ks_asm(KsFrag& f, unsigned img, int lane_base, int rbase) {
    const unsigned a = img + (unsigned)(lane_base ^ (2 * rbase));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(a), "i"(C * 64 * BR));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(a), "i"(C * 64 * BR + 8 * BR));
}
__device__ __forceinline__ bf16x8 ks_frag_value(const KsFrag& f) {
    const u32x4 r = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(uintptr_t)LDS_PTR(const char, p); }

struct Epi2 {
    const bf16* bias;
    const bf16* residual;
    const bf16* aux_in;
    bf16* aux_out;
    int64_t ldc, ldr, ldaux;
    int act, accumulate;
    float alpha;
    float* slab;       // split-K: fp32 partial matrices [splits][M][N] (ld = N); NULL when gridDim.y == 1
    float* cs_partial; // column sums of the output per 128-row block: [2 * tiles_m][N] (CS instantiations only), or NULL
    unsigned* sched;   // dynamic tile schedule of the persistent 256x256 kernel (desc->sched_state), or NULL: static round-robin
};

// ---- dynamic tile schedule (gemm3_kernel) ------------------------------------------------------------------------------------------
// A persistent grid with a STATIC tile list assumes every workgroup is resident from the start.  When another kernel holds some CUs —
// an RCCL all-reduce overlapping backward — the workgroups that do not fit start only when a running one has finished its whole list,
// and the launch takes twice as long.  With desc->sched_state the tiles are handed out by device-scope atomic counters instead: a
// workgroup that starts late finds the list (nearly) empty and leaves; the resident ones have shared its tiles.  One counter per XCD
// label (blockIdx.x & 7) and the same round / chunk geometry as the static order (xcd_remap), so the tiles an XCD works on at one time
// still share operand panels.  A workgroup owns at most two tiles (the running one and the next, whose first K-tile it prefetches):
// the draw for tile t + 1 is issued at the start of tile t - 1's epilogue and read at its end (nothing waits inside the K loop).
// No workgroup ever waits for another one (no spins: nothing to deadlock).  The last workgroup to leave zeroes the counters, so the
// state is ready for the next launch of the same stream; the caller zeroes it once, when it allocates it.
constexpr int SCHED_XCD_STRIDE = 16;      // 32-bit words between the XCD counters (64 B: no two counters share a line)
constexpr int SCHED_EXIT_WORD = 8 * SCHED_XCD_STRIDE;
// tile-list position of the k-th draw of XCD label x (G = workgroups of the launch = tiles per round); -1: list exhausted
__device__ __forceinline__ int sched_tile(unsigned k, int x, int G, int total) {
    const int nf = (G - x + 7) >> 3;                 // positions of this label in a full round
    const int full = total / G;
    if (nf > 0 && k < (unsigned)(full * nf)) {
        const int r = (int)k / nf, i = (int)k - r * nf;
        return r * G + xcd_remap(i * 8 + x, G);
    }
    const int k2 = (int)k - full * nf, cnt = total - full * G;
    if (k2 >= 0 && k2 < ((cnt - x + 7) >> 3)) return full * G + xcd_remap(k2 * 8 + x, cnt);
    return -1;
}
__device__ __forceinline__ void sched_leave(unsigned* sched, int tid) {
    if (tid == 0) {
        const unsigned before = atomicAdd(sched + SCHED_EXIT_WORD, 1u);
        if (before == gridDim.x - 1) {               // everybody else has left: nobody draws any more
#pragma unroll
            for (int x = 0; x < 8; ++x) sched[x * SCHED_XCD_STRIDE] = 0u;
            sched[SCHED_EXIT_WORD] = 0u;
        }
    }
}

// logical tile index -> (m0, n0): bands of 8 N-tiles, walking down M inside a band (neighbouring tiles share operand panels)
__device__ __forceinline__ void tile_origin(int t, int tiles_m, int tiles_n, int BM, int BN, int& m0, int& n0) {
    // 12 N-tiles (the qkv projection: N = 3072) as three bands of 4 rather than 8 + 4: every XCD block is 8 x 4 tiles
    const int BAND = (tiles_n > 8 && tiles_n % 8 != 0 && tiles_n % 4 == 0) ? 4 : 8;
    const int band_tiles = BAND * tiles_m;
    const int band = t / band_tiles;
    const int band_w = min(BAND, tiles_n - band * BAND);
    const int in_band = t - band * band_tiles;
    m0 = (in_band / band_w) * BM;
    n0 = (band * BAND + in_band % band_w) * BN;
}

// XCD-blocked work schedule.  Hardware hands workgroup w to XCD w % 8 and every XCD has a private L2, so operand panels are
// only shared between tiles that run on the SAME XCD at the same time.  The output tile grid of one K-slice is cut into compact
// blocks of bm x bn tiles (<= 64 = the XCD's resident workgroups, 2 per CU); block-slice bs = block * splits + slice is processed
// by XCD bs % 8 in round bs / 8, workgroup slot (w / 8) taking tile `slot` of the block.  A block touches bm + bn operand panels
// instead of 2 * bm * bn, and no other XCD touches them during that round (measured before this schedule, wgrad at B=166:
// FETCH_SIZE 2.7x the algorithmic bytes because each XCD walked 3 x 8 tiles of EVERY slice).
struct Sched2 {
    int bm, bn;        // block shape in tiles
    int nbn;           // blocks along N
    int nblocks;       // blocks per K-slice
    int splits;        // K-slices
    int k_per_split;   // contraction extent of a slice (multiple of BK2)
};

struct Work2 {
    int m0, n0, k_begin, nk, klast, slice;
};

// first round >= r in which workgroup (xcd, slot) owns a tile; -1 when there is none
__device__ __forceinline__ int sched2_locate(const Sched2& sc, int tiles_m, int tiles_n, int BM, int BN, int K, int xcd, int slot, int r,
                                             Work2& w) {
    const int total = sc.nblocks * sc.splits;
    for (int bs = xcd + 8 * r; bs < total; bs += 8, ++r) {
        const int block = bs / sc.splits;
        const int bi = block / sc.nbn, bj = block - bi * sc.nbn;
        const int rows = min(sc.bm, tiles_m - bi * sc.bm), cols = min(sc.bn, tiles_n - bj * sc.bn);
        if (slot >= rows * cols) continue;
        w.slice = bs - block * sc.splits;
        w.m0 = (bi * sc.bm + slot / cols) * BM;
        w.n0 = (bj * sc.bn + slot % cols) * BN;
        w.k_begin = w.slice * sc.k_per_split;
        const int k_end = min(K, w.k_begin + sc.k_per_split);
        w.nk = (k_end - w.k_begin + BK2 - 1) / BK2;
        w.klast = k_end - w.k_begin - (w.nk - 1) * BK2;      // valid contraction extent of the last K-tile (1..64)
        return r;
    }
    return -1;
}

// PERSISTENT kernel: 8 x slots workgroups (<= resident capacity) walk their XCD's block-slices round by round.  The first K-tile
// of the NEXT output tile is DMA-prefetched during the last K-step of the current one, so neither the prologue latency nor the
// epilogue (LDS-staged, full-line stores) leaves the MFMA pipe idle for a whole HBM round trip.
template <int LA, int LB, int BM, int BN, int WM, int WN, typename OutT>
__global__ __launch_bounds__(WM* WN * 64) void gemm2_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, OutT* __restrict__ C,
                                                             int M, int N, int K, int64_t lda, int64_t ldb, Epi2 ep, int tiles_m,
                                                             int tiles_n, Sched2 sc) {
    constexpr int NWAVES = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / WN) * TM, wn = (wave % WN) * TN;
    const int g = lane >> 4, li = lane & 15;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;

    // epilogue staging geometry (wave-private LDS rows inside the just-consumed pipeline buffer)
    constexpr int PADW = TN + 4;                       // fp32 words per staged row
    constexpr int LPR = TN / 8;                        // lanes per staged row (8 columns each)
    constexpr int RPI = 64 / LPR;                      // rows per wave-instruction
    const int prow = lane / LPR, pcol = (lane % LPR) * 8;
    static_assert(NWAVES * 16 * PADW * 4 <= BUF, "epilogue staging must fit one pipeline buffer");

    int it = 0;        // running K-tile counter: K-tile `it` lives in pipeline buffer it & 1
    Work2 cur, nxt;
    int sched_round = sched2_locate(sc, tiles_m, tiles_n, BM, BN, K, xcd, slot, 0, cur);
    if (sched_round < 0) return;
    issue_tile<LA, BM, NWAVES>(A, lda, cur.m0, cur.k_begin, M, smem, wave, lane, cur.nk == 1 ? cur.klast : BK2);
    issue_tile<LB, BN, NWAVES>(B, ldb, cur.n0, cur.k_begin, N, smem + A_BYTES, wave, lane, cur.nk == 1 ? cur.klast : BK2);

    for (int round = 0;; ++round) {
        // next tile of this workgroup (if any)
        const int next_round = sched2_locate(sc, tiles_m, tiles_n, BM, BN, K, xcd, slot, sched_round + 1, nxt);
        const bool has_next = next_round >= 0;
        const int m0 = cur.m0, n0 = cur.n0, k_begin = cur.k_begin, nk = cur.nk, klast = cur.klast;
        float* slab = ep.slab ? ep.slab + (int64_t)cur.slice * M * N : nullptr;
        f32x4 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nk; ++kt, ++it) {
            // K-tile `it` has landed (vmcnt(0) of every wave, then the barrier); everyone is done with the other buffer
            __syncthreads();
            const char* bufA = smem + (it & 1) * BUF;
            const char* bufB = bufA + A_BYTES;
            char* nb = smem + ((it + 1) & 1) * BUF;
            if (kt + 1 < nk) {
                const int kr = (kt + 2 == nk) ? klast : BK2;
                issue_tile<LA, BM, NWAVES>(A, lda, m0, k_begin + (kt + 1) * BK2, M, nb, wave, lane, kr);
                issue_tile<LB, BN, NWAVES>(B, ldb, n0, k_begin + (kt + 1) * BK2, N, nb + A_BYTES, wave, lane, kr);
            } else if (has_next) {
                const int kr = nxt.nk == 1 ? nxt.klast : BK2;
                issue_tile<LA, BM, NWAVES>(A, lda, nxt.m0, nxt.k_begin, M, nb, wave, lane, kr);
                issue_tile<LB, BN, NWAVES>(B, ldb, nxt.n0, nxt.k_begin, N, nb + A_BYTES, wave, lane, kr);
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

        // ---- epilogue through LDS so every global access is a full 128-B line ---------------------------------------
        //   phase A: v = alpha*acc + bias (fp32) of a 16-row x TN-col strip -> wave-private LDS rows
        //   phase B: each lane owns 8 consecutive columns of a row: activation / aux / residual / accumulate, 16-B loads+stores
        // The staging area is the pipeline buffer of the K-tile just consumed; the other buffer is receiving the next tile's
        // first K-tile, so only LDS reads (not the DMA) are waited for here: raw barrier, no vmcnt.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        float* stage = reinterpret_cast<float*>(smem + ((it - 1) & 1) * BUF) + wave * (16 * PADW);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
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
                *reinterpret_cast<f32x4*>(stage + li * PADW + 16 * j + 4 * g) = v;
            }
            // wave-private region: LDS executes one wave's DS instructions in order (writes above, reads below, next pass's writes after)
#pragma unroll
            for (int rr = 0; rr < 16; rr += RPI) {
                const int row = rr + prow;
                const int m = m0 + wm + 16 * i + row;
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
                        store_out16(ep.aux_out + (int64_t)m * ep.ldaux + n, o);
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = o.get(r);   // activation sees the stored (rounded) pre-activation
                    }
                    gelu_fast8(v);
                } else if (ep.act == UCFVIT_ACT_GELU_GRAD) {
                    const Vec16<bf16> h = *reinterpret_cast<const Vec16<bf16>*>(ep.aux_in + (int64_t)m * ep.ldaux + n);
                    float hf[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) hf[r] = h.get(r);
                    gelu_grad_fast8(v, hf);
                } else if (ep.act == UCFVIT_ACT_GELU_SAVE_DERIV) {
                    float df[8];
                    gelu_and_grad_fast8(v, df);
                    Vec16<bf16> o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o.set(r, df[r]);
                    store_out16(ep.aux_out + (int64_t)m * ep.ldaux + n, o);
                } else if (ep.act == UCFVIT_ACT_MUL_AUX) {
                    const Vec16<bf16> h = *reinterpret_cast<const Vec16<bf16>*>(ep.aux_in + (int64_t)m * ep.ldaux + n);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] *= h.get(r);
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
                    store_out16(cp, o);
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
        if (!has_next) break;
        cur = nxt;
        sched_round = next_round;
    }
}

// =====================================================================================================================
// PING-PONG variant of the persistent 256x256x64 kernel (8 waves).  The two wave groups G0 = waves 0-3 (rows 0-127 of the
// tile) and G1 = waves 4-7 (rows 128-255) — wave w and w+4 share a SIMD — run the SAME program skewed by one barrier
// interval: while one group is in a MEM interval (DMA issue for the next K-tile + ds_read of a 32-deep fragment set), the
// other is in a COMPUTE interval (32 MFMAs out of registers), so each SIMD's matrix pipe always has one wave feeding it.
//
//   interval 4t+0: G0 MEM(kt,c0) [+ issue DMA of its half of tile kt+1] | G1 COMPUTE(kt-1,c1)
//   interval 4t+1: G0 COMPUTE(kt,c0)                                    | G1 MEM(kt,c0) [+ issue DMA of its half of tile kt+1]
//   interval 4t+2: G0 MEM(kt,c1)                                        | G1 COMPUTE(kt,c0)
//   interval 4t+3: G0 COMPUTE(kt,c1) ; vmcnt(0)                         | G1 MEM(kt,c1) ; vmcnt(0)
//
// Every interval ends with one raw s_barrier executed by all 8 waves.  DMA data of tile kt+1 is waited for (vmcnt(0) by the
// issuing waves, then the barrier) at the end of interval 4t+3, i.e. 3-4 intervals after it was issued and right before its
// first reader (G0 in interval 4t+4).  A pipeline buffer is overwritten only >= 1 interval after its last ds_read.
// =====================================================================================================================
// Per-lane 32-bit byte offsets (row clamp + source-side swizzle folded in) of this wave's 4 DMA pieces of an operand K-tile;
// the K position is a wave-uniform byte offset added to the (SGPR) base pointer at issue time.
template <int LAYOUT, int BR>
__device__ __forceinline__ void half_offsets(unsigned (&off)[4], int64_t ld, int r0, int R, int grp, int w4, int lane) {
    constexpr int NP = BR / 8, PER = NP / 8;
    static_assert(PER == 4, "256-wide operand tiles only");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int idx = grp * (NP / 2) + w4 * PER + i;
        if (LAYOUT == UCFVIT_LAYOUT_KC) {
            const int row = idx * 8 + (lane >> 3);
            const int gslot = (lane & 7) ^ (row & 7);
            int gr = r0 + row;
            gr = gr < R ? gr : R - 1;
            off[i] = (unsigned)(((int64_t)gr * ld + gslot * 8) * 2);
        } else {
            constexpr int RB = BR * 2, KPP = 1024 / RB;
            const int krow = idx * KPP + (lane * 16) / RB;
            const int pbyte = (lane * 16) % RB;
            const int col = (pbyte ^ ks_swz2(krow)) >> 1;
            int gc = r0 + col;
            gc = gc <= R - 8 ? gc : R - 8;
            off[i] = (unsigned)(((int64_t)krow * ld + gc) * 2);
        }
    }
}
template <int BR>
__device__ __forceinline__ void issue_half(const char* __restrict__ base_k, const unsigned (&off)[4], char* lds, int grp, int w4) {
    constexpr int NP = BR / 8, PER = NP / 8;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int idx = grp * (NP / 2) + w4 * PER + i;
        __builtin_amdgcn_global_load_lds((gptr_t)(base_k + off[i]), (lptr_t)(lds + idx * 1024), 16, 0, 0);
    }
}
// ragged last K-tile: lanes beyond the valid contraction extent `krem` read the zero page
template <int LAYOUT, int BR>
__device__ __forceinline__ void issue_half_tail(const char* __restrict__ base_k, const unsigned (&off)[4], char* lds, int grp, int w4, int lane,
                                                int krem) {
    constexpr int NP = BR / 8, PER = NP / 8;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int idx = grp * (NP / 2) + w4 * PER + i;
        bool ok;
        if (LAYOUT == UCFVIT_LAYOUT_KC) {
            const int row = idx * 8 + (lane >> 3);
            ok = (((lane & 7) ^ (row & 7)) * 8) < krem;
        } else {
            constexpr int RB = BR * 2, KPP = 1024 / RB;
            ok = (idx * KPP + (lane * 16) / RB) < krem;
        }
        const char* src = ok ? base_k + off[i] : reinterpret_cast<const char*>(g_zero_page) + lane * 16;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + idx * 1024), 16, 0, 0);
    }
}
// wave-uniform byte offset of K position k0 for an operand
template <int LAYOUT> __device__ __forceinline__ int64_t k_byte_off(int k0, int64_t ld) {
    return LAYOUT == UCFVIT_LAYOUT_KC ? (int64_t)k0 * 2 : (int64_t)k0 * ld * 2;
}

#define PP_BARRIER()                              \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        asm volatile("" ::: "memory");            \
        __builtin_amdgcn_s_barrier();             \
        asm volatile("" ::: "memory");            \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)
#define PP_WAIT_B() /* all but the 4 youngest (the A pieces just issued) */ \
    do {                                                   \
        __builtin_amdgcn_sched_barrier(0);                 \
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)
#define PP_WAIT_DMA()                                      \
    do {                                                   \
        __builtin_amdgcn_sched_barrier(0);                 \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)

// up to GROUP_MAX GEMMs of one launch (same K, layouts and output type): the weight gradients of one or several transformer blocks
constexpr int GROUP_MAX = 32;   // 32 x 72-byte problem records stay inside the 4-KiB kernel-argument segment
struct Problem3 {
    const void* A;
    const void* B;
    void* C;
    int64_t lda, ldb, ldc;
    int M, N, tiles_m, tiles_n, tile_start, accumulate;
};
// NP = problem slots of the kernel-argument record: a single GEMM carries 1 (a 2.3-KiB argument segment per launch measurably
// lengthens the gap in front of the kernel), a grouped launch GROUP_MAX
template <int NP> struct GroupsT {
    Problem3 p[NP];
    int n, total_tiles;
};

// Epilogue specialisations of the forward / data-gradient GEMM (KC x KC, bf16 out, one problem, no split-K, no accumulate).  The
// generic epilogue decides everything at run time; under the accumulators' register pressure that code spills, and hipcc, which
// cannot see the asm waits that retire the LDS-DMA, drains vmcnt(0) — loads AND the strips' stores, which share the counter — at
// every use of a loaded value (124 drains per tile).  A specialised epilogue is straight-line: its one C-shaped input is fetched
// PD strips ahead with counted waits, its stores are never waited for.  (A cache-warming DMA touch of that input during the K loop
// was worth 5 % before this and nothing after it; removed.)
enum { EPI_GENERIC = 0, EPI_PLAIN = 1, EPI_RESIDUAL = 2, EPI_GELU = 3, EPI_GELU_GRAD = 4, EPI_GELU_SAVE_DERIV = 5, EPI_MUL_AUX = 6 };

typedef GroupsT<GROUP_MAX> Groups3;

template <int LA, int LB, typename OutT, int EPI = EPI_GENERIC, bool CS = false, int NP = GROUP_MAX>
__global__ __launch_bounds__(512) void gemm3_kernel(GroupsT<NP> gt, int K, Epi2 ep, int k_per_split) {
    constexpr int BM = 256, BN = 256, WN = 4, TM = 128, TN = 64, FM = 8, FN = 4;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nwg = gt.total_tiles;
    const int G = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wm = grp * TM, wn = w4 * TN;
    const int g = lane >> 4, li = lane & 15;

    const int k_begin = blockIdx.y * k_per_split;
    const int k_end = min(K, k_begin + k_per_split);
    const int nk = (k_end - k_begin + BK2 - 1) / BK2;
    const int klast = k_end - k_begin - (nk - 1) * BK2;      // valid contraction extent of the last K-tile (1..64)
    constexpr int PADW = TN + 4, LPR = TN / 8, RPI = 64 / LPR;
    const int prow = lane / LPR, pcol = (lane % LPR) * 8;

    int it = 0;
    int m0, n0;
    // current problem of this workgroup (grouped launches carry up to 4 GEMMs that share K and the layouts)
    const char *Ab, *Bb;
    OutT* C;
    int M, N, accum;
    int64_t lda, ldb, ldc;
#define PP_SELECT(t_, Ab_, Bb_, C_, M_, N_, lda_, ldb_, ldc_, acc_, m0_, n0_)                 \
    do {                                                                                       \
        int gi = 0;                                                                            \
        for (int q = 1; q < gt.n; ++q) if ((t_) >= gt.p[q].tile_start) gi = q;                 \
        const Problem3& P = gt.p[gi];                                                          \
        Ab_ = reinterpret_cast<const char*>(P.A);                                              \
        Bb_ = reinterpret_cast<const char*>(P.B);                                              \
        C_ = reinterpret_cast<OutT*>(P.C);                                                     \
        M_ = P.M; N_ = P.N; lda_ = P.lda; ldb_ = P.ldb; ldc_ = P.ldc; acc_ = P.accumulate;     \
        tile_origin((t_) - P.tile_start, P.tiles_m, P.tiles_n, BM, BN, m0_, n0_);              \
    } while (0)
    const bool dyn = ep.sched != nullptr;                         // (launched with gridDim.y == 1 only)
    const int xcd = blockIdx.x & 7;
    int* sched_lds = reinterpret_cast<int*>(smem + 2 * BUF);      // 2 words behind the pipeline buffers (dynamic schedule only)
    int t_next = -1;                                              // dynamic: tile-list position of the NEXT tile (-1: none)
    {
        int t0;
        if (dyn) {
            if (tid == 0) {
                unsigned* ctr = ep.sched + xcd * SCHED_XCD_STRIDE;
                const int a = sched_tile(atomicAdd(ctr, 1u), xcd, G, nwg);
                sched_lds[0] = a;
                sched_lds[1] = a < 0 ? -1 : sched_tile(atomicAdd(ctr, 1u), xcd, G, nwg);
            }
            __syncthreads();
            t0 = sched_lds[0];
            t_next = sched_lds[1];
            if (t0 < 0) {                                         // started late: the resident workgroups have taken every tile
                sched_leave(ep.sched, tid);
                return;
            }
        } else {
            const int first = min(G, nwg);
            if ((int)blockIdx.x >= first) return;
            t0 = xcd_remap(blockIdx.x, first);
        }
        PP_SELECT(t0, Ab, Bb, C, M, N, lda, ldb, ldc, accum, m0, n0);
    }
    float* slab = ep.slab ? ep.slab + (int64_t)blockIdx.y * M * N : nullptr;
    // prologue: both groups issue their halves of the first K-tile, wait, barrier
    unsigned offA[4], offB[4];
    half_offsets<LA, BM>(offA, lda, m0, M, grp, w4, lane);
    half_offsets<LB, BN>(offB, ldb, n0, N, grp, w4, lane);
    if (nk == 1 && klast < BK2) {
        issue_half_tail<LA, BM>(Ab + k_byte_off<LA>(k_begin, lda), offA, smem, grp, w4, lane, klast);
        issue_half_tail<LB, BN>(Bb + k_byte_off<LB>(k_begin, ldb), offB, smem + A_BYTES, grp, w4, lane, klast);
    } else {
        issue_half<BM>(Ab + k_byte_off<LA>(k_begin, lda), offA, smem, grp, w4);
        issue_half<BN>(Bb + k_byte_off<LB>(k_begin, ldb), offB, smem + A_BYTES, grp, w4);
    }
    PP_WAIT_DMA();
    PP_BARRIER();

    bf16x8 fa[FM], fb[FN];   // one 32-deep fragment set (48 VGPRs)
    KsFrag ka[FM], kb[FN];   // the same registers while asm reads of a KS operand are in flight
#define KS_TIE1(f_) "+v"(f_.lo), "+v"(f_.hi)
#define KS_TIE4(a_) KS_TIE1(a_[0]), KS_TIE1(a_[1]), KS_TIE1(a_[2]), KS_TIE1(a_[3])
#define KS_TIE8(a_) KS_TIE4(a_), KS_TIE1(a_[4]), KS_TIE1(a_[5]), KS_TIE1(a_[6]), KS_TIE1(a_[7])
    const int ksA0 = ks_lane_base<BM>(lane), ksB0 = ks_lane_base<BN>(lane);

    for (int round = 0;; ++round) {
        int nm0 = 0, nn0 = 0;
        if (!dyn) {
            const int next_base = (round + 1) * G;
            const int next_cnt = min(G, nwg - next_base);
            t_next = (int)blockIdx.x < next_cnt ? next_base + xcd_remap(blockIdx.x, next_cnt) : -1;
        }
        const bool has_next = t_next >= 0;
        const char *nAb = Ab, *nBb = Bb;
        OutT* nC = C;
        int nM = M, nN = N, naccum = accum;
        int64_t nlda = lda, nldb = ldb, nldc = ldc;
        if (has_next) {
            PP_SELECT(t_next, nAb, nBb, nC, nM, nN, nlda, nldb, nldc, naccum, nm0, nn0);
        }

        f32x4 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define PP_READ(bufA_, bufB_, c_)                                                              \
    do {                                                                                       \
        _Pragma("unroll") for (int j = 0; j < FN; ++j) {                                       \
            if constexpr (LB == UCFVIT_LAYOUT_KS) load_frag_ks_asm<BN, c_>(kb[j], lds_addr(bufB_), ksB, wn + 16 * j); \
            else fb[j] = load_frag2<LB, BN>(bufB_, wn + 16 * j, c_, lane);                     \
        }                                                                                      \
        _Pragma("unroll") for (int i = 0; i < FM; ++i) {                                       \
            if constexpr (LA == UCFVIT_LAYOUT_KS) load_frag_ks_asm<BM, c_>(ka[i], lds_addr(bufA_), ksA, wm + 16 * i); \
            else fa[i] = load_frag2<LA, BM>(bufA_, wm + 16 * i, c_, lane);                     \
        }                                                                                      \
        if constexpr (LA == UCFVIT_LAYOUT_KS || LB == UCFVIT_LAYOUT_KS) {                      \
            /* the asm reads have landed before this group leaves its MEM interval; then they become ordinary values */ \
            if constexpr (LA == UCFVIT_LAYOUT_KS && LB == UCFVIT_LAYOUT_KS)                    \
                asm volatile("s_waitcnt lgkmcnt(0)" : KS_TIE8(ka), KS_TIE4(kb));               \
            else if constexpr (LA == UCFVIT_LAYOUT_KS)                                         \
                asm volatile("s_waitcnt lgkmcnt(0)" : KS_TIE8(ka));                            \
            else                                                                               \
                asm volatile("s_waitcnt lgkmcnt(0)" : KS_TIE4(kb));                            \
            if constexpr (LA == UCFVIT_LAYOUT_KS) { _Pragma("unroll") for (int i = 0; i < FM; ++i) fa[i] = ks_frag_value(ka[i]); } \
            if constexpr (LB == UCFVIT_LAYOUT_KS) { _Pragma("unroll") for (int j = 0; j < FN; ++j) fb[j] = ks_frag_value(kb[j]); } \
        }                                                                                      \
    } while (0)
#define PP_COMPUTE()                                                                           \
    do {                                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                         \
        _Pragma("unroll") for (int i = 0; i < FM; ++i)                                         \
            _Pragma("unroll") for (int j = 0; j < FN; ++j)                                     \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                         \
    } while (0)
#define PP_ISSUE_ONE(LAY_, BR_, base_, ld_, off_, nbase_, nld_, r0n_, Rn_, ldsoff_, kt_)         \
    do {                                                                                       \
        char* nb = smem + ((it + 1) & 1) * BUF + (ldsoff_);                                    \
        if ((kt_) + 1 < nk) {                                                                  \
            if ((kt_) + 2 == nk && klast < BK2)                                                \
                issue_half_tail<LAY_, BR_>(base_ + k_byte_off<LAY_>(k_begin + ((kt_) + 1) * BK2, ld_), off_, nb, grp, w4, lane, klast); \
            else                                                                               \
                issue_half<BR_>(base_ + k_byte_off<LAY_>(k_begin + ((kt_) + 1) * BK2, ld_), off_, nb, grp, w4); \
        } else if (has_next) {                                                                 \
            half_offsets<LAY_, BR_>(off_, nld_, r0n_, Rn_, grp, w4, lane);  /* offsets now belong to the next tile */ \
            if (nk == 1 && klast < BK2)                                                        \
                issue_half_tail<LAY_, BR_>(nbase_ + k_byte_off<LAY_>(k_begin, nld_), off_, nb, grp, w4, lane, klast); \
            else                                                                               \
                issue_half<BR_>(nbase_ + k_byte_off<LAY_>(k_begin, nld_), off_, nb, grp, w4);  \
        }                                                                                      \
    } while (0)
#define PP_ISSUE_B(kt_) PP_ISSUE_ONE(LB, BN, Bb, ldb, offB, nBb, nldb, nn0, nN, A_BYTES, kt_)
#define PP_ISSUE_A(kt_) PP_ISSUE_ONE(LA, BM, Ab, lda, offA, nAb, nlda, nm0, nM, 0, kt_)

        // One program for both groups; G1 runs it one barrier interval behind G0 (extra barrier before / after the loop).
        if (grp == 1) PP_BARRIER();
#pragma clang loop unroll(disable)
        for (int kt = 0; kt < nk; ++kt, ++it) {
            const char* bufA = smem + (it & 1) * BUF;
            const char* bufB = bufA + A_BYTES;
            int ksA = ksA0, ksB = ksB0;
            if constexpr (LA == UCFVIT_LAYOUT_KS) asm volatile("" : "+v"(ksA));   // opaque per iteration: the per-fragment XORs are
            if constexpr (LB == UCFVIT_LAYOUT_KS) asm volatile("" : "+v"(ksB));   // recomputed, not hoisted into 12 live VGPRs
            PP_ISSUE_B(kt);                 // MEM(c0): DMA of this group's half of the next B tile (read by BOTH groups) ...
            PP_ISSUE_A(kt);                 // ... and of its OWN rows of the next A tile (that region was last read in MEM(c1) of the tile before)
            PP_READ(bufA, bufB, 0);
            PP_BARRIER();
            PP_COMPUTE();                   // COMPUTE(c0)
            PP_BARRIER();
            PP_READ(bufA, bufB, 1);         // MEM(c1): fragment reads only
            if (grp == 1) PP_WAIT_B();      // G1's B half (issued 2 intervals ago) must be visible before G0's next MEM(c0); its A pieces stay in flight
            PP_BARRIER();
            PP_COMPUTE();                   // COMPUTE(c1)
            PP_WAIT_DMA();                  // everything this wave issued for the next K-tile has landed
            PP_BARRIER();
        }
        if (grp == 0) PP_BARRIER();
#undef PP_READ
#undef PP_COMPUTE
#undef PP_SELECT
#undef PP_ISSUE_ONE
#undef PP_ISSUE_A
#undef PP_ISSUE_B

        // ---- epilogue: staging area = the pipeline buffer of the K-tile just consumed ------------------------------------------
        float* stage = reinterpret_cast<float*>(smem + ((it - 1) & 1) * BUF) + wave * (16 * PADW);
        // dynamic schedule: the draw for the tile after the next one goes out at the start of the epilogue and is read when it is done
        unsigned drawn = 0u;
        const bool draw = dyn && has_next && tid == 0;
        if constexpr (EPI == EPI_GENERIC) {
            if (draw) drawn = atomicAdd(ep.sched + xcd * SCHED_XCD_STRIDE, 1u);
        }
        if constexpr (EPI != EPI_GENERIC) {
            static_assert(sizeof(OutT) == 2, "specialised epilogues write bf16");
            constexpr int NPASS = 16 / RPI;
            constexpr bool HAS_IN = EPI == EPI_RESIDUAL || EPI == EPI_GELU_GRAD || EPI == EPI_MUL_AUX;
            constexpr int PD = 3;                      // strips of the C-shaped input in flight ahead of their use
            // every DMA piece has been waited for by the asm waits above; a wait hipcc can see resets its bookkeeping (free: the
            // queue is empty), so the loads below get counted waits instead of vmcnt(0)
            __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0) expcnt(7) lgkmcnt(15)
            if (draw) drawn = atomicAdd(ep.sched + xcd * SCHED_XCD_STRIDE, 1u);
            Vec4<bf16> bias_v[FN];
            const bool has_bias = ep.bias != nullptr;
            if (has_bias) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int n = n0 + wn + 16 * j + 4 * g;
                    bias_v[j] = *reinterpret_cast<const Vec4<bf16>*>(ep.bias + (n < N ? n : 0));
                }
            }
            const bf16* in_base = EPI == EPI_RESIDUAL ? ep.residual : ep.aux_in;
            const int64_t in_ld = EPI == EPI_RESIDUAL ? ep.ldr : ep.ldaux;
            Vec16<bf16> pre[FM][NPASS];
            const int ncl = min(n0 + wn + pcol, N - 8);                 // clamped: the loads are unconditional, the stores masked
#define EPI_LOAD(i_)                                                                                                   \
    do {                                                                                                               \
        _Pragma("unroll") for (int ps = 0; ps < NPASS; ++ps) {                                                         \
            const int m_ = min(m0 + wm + 16 * (i_) + ps * RPI + prow, M - 1);                                          \
            pre[i_][ps] = *reinterpret_cast<const Vec16<bf16>*>(in_base + (int64_t)m_ * in_ld + ncl);                  \
        }                                                                                                              \
    } while (0)
            if constexpr (HAS_IN) {
#pragma unroll
                for (int i = 0; i < PD; ++i) EPI_LOAD(i);
            }
            float csum[8];          // CS: this lane's 8 columns summed over its 16 rows of the wave tile
#pragma unroll
            for (int r = 0; r < 8; ++r) csum[r] = 0.f;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    f32x4 v = acc[i][j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= ep.alpha;
                    if (has_bias) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += bias_v[j].get(r);
                    }
                    *reinterpret_cast<f32x4*>(stage + li * PADW + 16 * j + 4 * g) = v;
                }
                if constexpr (HAS_IN) {
                    if (i + PD < FM) EPI_LOAD(i + PD);      // ahead of this strip's stores: its wait will not include them
                }
#pragma unroll
                for (int rr = 0; rr < 16; rr += RPI) {
                    const int row = rr + prow;
                    const int m = m0 + wm + 16 * i + row;
                    const int n = n0 + wn + pcol;
                    const bool inside = m < M && n < N;
                    float v[8];
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * PADW + pcol);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * PADW + pcol + 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = lo[r];
                        v[4 + r] = hi[r];
                    }
                    if constexpr (EPI == EPI_GELU) {
                        if (ep.aux_out) {
                            Vec16<bf16> o;
#pragma unroll
                            for (int r = 0; r < 8; ++r) o.set(r, v[r]);
                            if (inside) store_out16(ep.aux_out + (int64_t)m * ep.ldaux + n, o);
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] = o.get(r);   // activation sees the stored (rounded) pre-activation
                        }
                        gelu_fast8(v);
                    } else if constexpr (EPI == EPI_GELU_SAVE_DERIV) {
                        float df[8];
                        gelu_and_grad_fast8(v, df);
                        Vec16<bf16> o;
#pragma unroll
                        for (int r = 0; r < 8; ++r) o.set(r, df[r]);
                        if (inside) store_out16(ep.aux_out + (int64_t)m * ep.ldaux + n, o);
                    } else if constexpr (EPI == EPI_GELU_GRAD) {
                        float hf[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) hf[r] = pre[i][rr / RPI].get(r);
                        gelu_grad_fast8(v, hf);
                    } else if constexpr (EPI == EPI_MUL_AUX) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] *= pre[i][rr / RPI].get(r);
                    } else if constexpr (EPI == EPI_RESIDUAL) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += pre[i][rr / RPI].get(r);
                    }
                    Vec16<bf16> o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o.set(r, v[r]);
                    if (inside) store_out16(C + (int64_t)m * ldc + n, o);
                    if constexpr (CS) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) csum[r] += inside ? v[r] : 0.f;
                    }
                }
            }
            if constexpr (CS) {
                // lanes l, l+8, ..., l+56 hold the same 8 columns for different rows: xor-shuffle over lane bits 3..5, then lanes 0-7
                // write the 64 column sums of this wave's 128 x 64 sub-tile (every [row block][column] entry is written exactly once)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float t = csum[r];
                    t += __shfl_xor(t, 8, 64);
                    t += __shfl_xor(t, 16, 64);
                    t += __shfl_xor(t, 32, 64);
                    csum[r] = t;
                }
                const int n = n0 + wn + pcol;
                if (prow == 0 && n < N) {
                    float* cp = ep.cs_partial + (int64_t)((m0 >> 7) + grp) * N + n;
                    *reinterpret_cast<f32x4*>(cp) = f32x4{csum[0], csum[1], csum[2], csum[3]};
                    *reinterpret_cast<f32x4*>(cp + 4) = f32x4{csum[4], csum[5], csum[6], csum[7]};
                }
            }
#undef EPI_LOAD
        } else {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
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
                *reinterpret_cast<f32x4*>(stage + li * PADW + 16 * j + 4 * g) = v;
            }
#pragma unroll
            for (int rr = 0; rr < 16; rr += RPI) {
                const int row = rr + prow;
                const int m = m0 + wm + 16 * i + row;
                const int n = n0 + wn + pcol;
                float v[8];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * PADW + pcol);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * PADW + pcol + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = lo[r];
                    v[4 + r] = hi[r];
                }
                if (m >= M || n >= N) continue;
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
                        store_out16(ep.aux_out + (int64_t)m * ep.ldaux + n, o);
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = o.get(r);
                    }
                    gelu_fast8(v);
                } else if (ep.act == UCFVIT_ACT_GELU_GRAD) {
                    const Vec16<bf16> h = *reinterpret_cast<const Vec16<bf16>*>(ep.aux_in + (int64_t)m * ep.ldaux + n);
                    float hf[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) hf[r] = h.get(r);
                    gelu_grad_fast8(v, hf);
                } else if (ep.act == UCFVIT_ACT_GELU_SAVE_DERIV) {
                    float df[8];
                    gelu_and_grad_fast8(v, df);
                    Vec16<bf16> o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o.set(r, df[r]);
                    store_out16(ep.aux_out + (int64_t)m * ep.ldaux + n, o);
                } else if (ep.act == UCFVIT_ACT_MUL_AUX) {
                    const Vec16<bf16> h = *reinterpret_cast<const Vec16<bf16>*>(ep.aux_in + (int64_t)m * ep.ldaux + n);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] *= h.get(r);
                }
                if (ep.residual) {
                    const Vec16<bf16> rv = *reinterpret_cast<const Vec16<bf16>*>(ep.residual + (int64_t)m * ep.ldr + n);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += rv.get(r);
                }
                OutT* cp = C + (int64_t)m * ldc + n;
                if constexpr (sizeof(OutT) == 2) {
                    if (accum) {
                        const Vec16<bf16> old = *reinterpret_cast<const Vec16<bf16>*>(cp);
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += old.get(r);
                    }
                    Vec16<bf16> o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o.set(r, v[r]);
                    store_out16(cp, o);
                } else {
                    f32x4 o0, o1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o0[r] = v[r];
                        o1[r] = v[4 + r];
                    }
                    if (accum) {
                        o0 += *reinterpret_cast<const f32x4*>(cp);
                        o1 += *reinterpret_cast<const f32x4*>(cp + 4);
                    }
                    *reinterpret_cast<f32x4*>(cp) = o0;
                    *reinterpret_cast<f32x4*>(cp + 4) = o1;
                }
            }
        }
        }   // EPI_GENERIC
        if (!has_next) break;
        m0 = nm0;
        n0 = nn0;
        Ab = nAb; Bb = nBb; C = nC; M = nM; N = nN; accum = naccum; lda = nlda; ldb = nldb; ldc = nldc;
        if (draw) sched_lds[0] = sched_tile(drawn, xcd, G, nwg);     // (one lane waits for its atomic here; the epilogue hid its latency)
        // the next tile's first K-tile was issued during the last K-tile and waited for (vmcnt(0) + barrier) at its end;
        // the epilogue's LDS staging of every wave must be finished before that buffer's partner is refilled:
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (dyn) t_next = sched_lds[0];      // written before the barrier; rewritten only behind the next tile's end-of-tile barrier
    }
    if (dyn) sched_leave(ep.sched, tid);
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

static bool generic_epilogue_only();
static bool pp_enabled();

struct Plan2 {
    int big;     // 1: 256x256 tile, 0: 128x128
    int splits;
    int k_per_split;
};

inline bool plan2(const ucfvit_gemm_desc* d, Plan2* p) {
    if (d->dtype != UCFVIT_BF16) return false;
    if (d->K < 128) return false;
    // ragged last K-tile is zero-filled by the DMA issue; 16-byte vectors along K need K % 8 == 0 only for KC operands
    if ((d->a_layout == UCFVIT_LAYOUT_KC || d->b_layout == UCFVIT_LAYOUT_KC) && d->K % 8 != 0) return false;
    if (d->M < 128 || d->N < 128) return false;
    const int64_t t256 = ((d->M + 255) / 256) * ((d->N + 255) / 256);
    const int64_t t128 = ((d->M + 127) / 128) * ((d->N + 127) / 128);
    const int64_t ktiles_all = (d->K + BK2 - 1) / BK2;
    const bool plain_epi = !d->bias && !d->residual && !d->aux_in && !d->aux_out && d->act == UCFVIT_ACT_NONE;
    p->splits = 1;
    if (t256 >= 192) {
        p->big = 1;
    } else {
        p->big = 0;
        if (plain_epi && t128 < 384) {
            constexpr int target = 384;
            // split-K so that every XCD owns whole 8 x 8-tile blocks (64 resident workgroups = 2 per CU) of ONE K-slice and
            // all 8 XCDs are busy in every round: nb64 * s block-slices must be a multiple of 8 (see Sched2)
            const int64_t tm = (d->M + 127) / 128, tn = (d->N + 127) / 128;
            const int nb64 = (int)(((tm + 7) / 8) * ((tn + 7) / 8));
            int g8 = 8;
            while (nb64 % g8) g8 >>= 1;                     // gcd(nb64, 8)
            int s = 8 / g8;
            if (nb64 * s > target / 8) s = 1;                // too many rounds of slab traffic: plain persistent walk
            const int kmax = (int)(ktiles_all / 8);          // at least 8 K-tiles per slice
            if (s > kmax) s = kmax;
            if (s < 1) s = 1;
            p->splits = s;
        }
    }
    const int64_t ktiles = (d->K + BK2 - 1) / BK2;
    p->k_per_split = (int)(((ktiles + p->splits - 1) / p->splits) * BK2);
    p->splits = (int)((d->K + p->k_per_split - 1) / p->k_per_split);
    return true;
}

// block shape for the XCD-blocked schedule: `cap` = resident workgroups of one XCD (32 CUs x 1 or 2)
inline Sched2 make_sched2(int tiles_m, int tiles_n, int splits, int k_per_split, int cap) {
    const int64_t work = (int64_t)tiles_m * tiles_n * splits;
    int T = (int)((work + 7) / 8);                        // tile-slices per XCD if everything fits one round
    if (T > cap || splits > 1) T = cap;                   // split-K plans are made for full blocks (plan2)
    if (T < 1) T = 1;
    int bm = cap == 64 ? 8 : 4;
    if (bm > tiles_m) bm = tiles_m;
    if (bm > T) bm = T;
    int bn = (T + bm - 1) / bm;
    if (bn > tiles_n) {                                   // narrow problem: spend the rest of the block on rows
        bn = tiles_n;
        bm = (T + bn - 1) / bn;
        if (bm > tiles_m) bm = tiles_m;
    }
    while (bm * bn > cap) {
        if (bn > 1) --bn;
        else --bm;
    }
    Sched2 sc;
    sc.bm = bm;
    sc.bn = bn;
    sc.nbn = (tiles_n + bn - 1) / bn;
    sc.nblocks = ((tiles_m + bm - 1) / bm) * sc.nbn;
    sc.splits = splits;
    sc.k_per_split = k_per_split;
    return sc;
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
    // persistent grid: 8 XCDs x `slots` workgroups; one workgroup per CU for the 128-KiB-LDS tile, two for the 64-KiB one
    const Sched2 sc = make_sched2(tiles_m, tiles_n, p.splits, p.k_per_split, BM == 256 ? 32 : 64);
    const int gx = 8 * sc.bm * sc.bn;
    hipLaunchKernelGGL(kern, dim3(gx), dim3(WM * WN * 64), smem, s, (const bf16*)d->A, (const bf16*)d->B, (OutT*)d->C,
                       (int)d->M, (int)d->N, (int)d->K, d->lda, d->ldb, ep, tiles_m, tiles_n, sc);
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

template <int LA, int LB, typename OutT, int EPI = EPI_GENERIC, bool CS = false, int NP = GROUP_MAX>
int launch3g(const GroupsT<NP>& gt, int K, const Epi2& ep, int splits, int k_per_split, hipStream_t s) {
    constexpr size_t smem = 2 * (size_t)(256 + 256) * 128 + 16;      // + the two schedule words
    auto kern = gemm3_kernel<LA, LB, OutT, EPI, CS, NP>;
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            ucfvit_set_error("ucfvit_gemm: cannot raise dynamic LDS to %zu bytes: %s", smem, hipGetErrorString(e));
            return UCFVIT_ERR_HIP;
        }
        done = true;
    }
    // persistent grid = the CUs this launch may count on.  One 512-thread workgroup fills a CU's register file, so a CU that hosts a
    // wave of another kernel (an RCCL all-reduce overlapping backward) cannot take one: a grid larger than the free CUs leaves
    // workgroups waiting for a whole tile list.  UCFVIT_GEMM_CUS (default 256) lets a multi-GPU job reserve the CUs RCCL uses.
    static const int cus = [] {                 // read once (thread-safe static: forward and autograd threads both launch GEMMs)
        const char* e = getenv("UCFVIT_GEMM_CUS");
        const int v = e ? atoi(e) : 256;
        return (v < 8 || v > 256) ? 256 : v;
    }();
    int cap = cus / splits;
    if (cap < 1) cap = 1;
    const int gx = gt.total_tiles < cap ? gt.total_tiles : cap;
    Epi2 epl = ep;
    if (splits != 1) epl.sched = nullptr;       // K-slices walk their own tile lists: static order
    hipLaunchKernelGGL(kern, dim3(gx, splits), dim3(512), smem, s, gt, K, epl, k_per_split);
    UCF_LAUNCH_CHECK("ucfvit_gemm(v3 ping-pong)");
    return UCFVIT_OK;
}

inline void fill_problem(Problem3& P, const ucfvit_gemm_desc* d, int tile_start) {
    P.A = d->A;
    P.B = d->B;
    P.C = d->C;
    P.lda = d->lda;
    P.ldb = d->ldb;
    P.ldc = d->ldc;
    P.M = (int)d->M;
    P.N = (int)d->N;
    P.tiles_m = (int)((d->M + 255) / 256);
    P.tiles_n = (int)((d->N + 255) / 256);
    P.tile_start = tile_start;
    P.accumulate = d->accumulate;
}

template <int LA, int LB, typename OutT>
int launch3(const ucfvit_gemm_desc* d, const Plan2& p, Epi2 ep, hipStream_t s) {
    GroupsT<1> gt;
    memset(&gt, 0, sizeof(gt));
    fill_problem(gt.p[0], d, 0);
    gt.n = 1;
    gt.total_tiles = gt.p[0].tiles_m * gt.p[0].tiles_n;
    if constexpr (LA == UCFVIT_LAYOUT_KC && LB == UCFVIT_LAYOUT_KC && sizeof(OutT) == 2) {
        // straight-line epilogues for the shapes of the training step (see EPI_* above); everything else is generic
        if (!generic_epilogue_only() && p.splits == 1 && !ep.slab && !d->accumulate && d->N >= 8) {
            const int K_ = (int)d->K;
            // the epilogue hidden under the partner group's K loop (gemm_stagger.hip) where that kernel applies
            const int rs = ucfvit_gemm_stagger_try(d, s);
            if (rs == 1) return UCFVIT_OK;
            if (rs < 0) return rs;
            if (ep.act == UCFVIT_ACT_NONE && !ep.residual && !ep.aux_out) {
                if (ep.cs_partial) return launch3g<LA, LB, OutT, EPI_PLAIN, true, 1>(gt, K_, ep, 1, p.k_per_split, s);
                return launch3g<LA, LB, OutT, EPI_PLAIN, false, 1>(gt, K_, ep, 1, p.k_per_split, s);
            }
            if (ep.act == UCFVIT_ACT_NONE && ep.residual && !ep.aux_out)
                return launch3g<LA, LB, OutT, EPI_RESIDUAL, false, 1>(gt, K_, ep, 1, p.k_per_split, s);
            if (ep.act == UCFVIT_ACT_GELU && !ep.residual)
                return launch3g<LA, LB, OutT, EPI_GELU, false, 1>(gt, K_, ep, 1, p.k_per_split, s);
            if (ep.act == UCFVIT_ACT_GELU_GRAD && !ep.residual && !ep.aux_out)
                return launch3g<LA, LB, OutT, EPI_GELU_GRAD, false, 1>(gt, K_, ep, 1, p.k_per_split, s);
            if (ep.act == UCFVIT_ACT_GELU_SAVE_DERIV && !ep.residual)
                return launch3g<LA, LB, OutT, EPI_GELU_SAVE_DERIV, false, 1>(gt, K_, ep, 1, p.k_per_split, s);
            if (ep.act == UCFVIT_ACT_MUL_AUX && !ep.residual && !ep.aux_out) {
                if (ep.cs_partial) return launch3g<LA, LB, OutT, EPI_MUL_AUX, true, 1>(gt, K_, ep, 1, p.k_per_split, s);
                return launch3g<LA, LB, OutT, EPI_MUL_AUX, false, 1>(gt, K_, ep, 1, p.k_per_split, s);
            }
        }
    }
    return launch3g<LA, LB, OutT, EPI_GENERIC, false, 1>(gt, (int)d->K, ep, p.splits, p.k_per_split, s);
}

static bool generic_epilogue_only() { return false; }
static bool pp_enabled() { return true; }

template <int LA, int LB, typename OutT>
int dispatch_tile(const ucfvit_gemm_desc* d, const Plan2& p, const Epi2& ep, hipStream_t s) {
    const int64_t a_bytes = ((d->a_layout == UCFVIT_LAYOUT_KC ? d->M : d->K) * d->lda) * 2;
    const int64_t b_bytes = ((d->b_layout == UCFVIT_LAYOUT_KC ? d->N : d->K) * d->ldb) * 2;
    if (p.big && pp_enabled() && a_bytes < (1ll << 32) && b_bytes < (1ll << 32)) return launch3<LA, LB, OutT>(d, p, ep, s);
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

// shape / alignment requirements of the DMA + vector-epilogue path (shared by ucfvit_gemm_v2_try and the by-product predicates)
static bool v2_operands_ok(const ucfvit_gemm_desc* d) {
    const int64_t a_contig = (d->a_layout == UCFVIT_LAYOUT_KC) ? d->K : d->M;
    const int64_t b_contig = (d->b_layout == UCFVIT_LAYOUT_KC) ? d->K : d->N;
    bool ok = ucf_is_aligned16(d->A) && ucf_is_aligned16(d->B) && d->lda % 8 == 0 && d->ldb % 8 == 0 && a_contig % 8 == 0 &&
              b_contig % 8 == 0 && d->N % 8 == 0 && d->ldc % 8 == 0 && ((uintptr_t)d->C) % 16 == 0 && d->M < (1ll << 31) &&
              d->N < (1ll << 31) && d->K < (1ll << 31);
    if (d->bias) ok = ok && ((uintptr_t)d->bias) % 8 == 0;
    if (d->residual) ok = ok && ((uintptr_t)d->residual) % 16 == 0 && d->ldr % 8 == 0;
    if (d->aux_in) ok = ok && ((uintptr_t)d->aux_in) % 16 == 0 && d->ldaux % 8 == 0;
    if (d->aux_out) ok = ok && ((uintptr_t)d->aux_out) % 16 == 0 && d->ldaux % 8 == 0;
    return ok;
}

// the output column sums (desc->c_colsum_partial) exist in the specialised MUL_AUX and plain epilogues of the 256x256 ping-pong kernel: the
// data-gradient GEMM through the activation (C = dh of the MLP) and the plain data gradients (C = dO of the attention projection: the V
// third of the qkv bias gradient), two 128-row blocks per output tile row
static int64_t colsum_rows_for(const ucfvit_gemm_desc* d) {
    Plan2 p;
    if (!d || !plan2(d, &p) || !v2_operands_ok(d)) return 0;
    const int64_t a_bytes = d->M * d->lda * 2, b_bytes = d->N * d->ldb * 2;
    const bool path = p.big && p.splits == 1 && pp_enabled() && !generic_epilogue_only() && a_bytes < (1ll << 32) && b_bytes < (1ll << 32) &&
                      d->a_layout == UCFVIT_LAYOUT_KC && d->b_layout == UCFVIT_LAYOUT_KC && d->out_dtype == UCFVIT_BF16 && !d->accumulate &&
                      d->N >= 8 && !d->residual && !d->aux_out &&
                      (d->act == UCFVIT_ACT_MUL_AUX || (d->act == UCFVIT_ACT_NONE && !d->aux_in));
    return path ? 2 * ((d->M + 255) / 256) : 0;
}

extern "C" int64_t ucfvit_gemm_colsum_rows(const ucfvit_gemm_desc* d) { return colsum_rows_for(d); }

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
    if (!v2_operands_ok(d)) return 0;
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
    ep.cs_partial = nullptr;
    ep.sched = (unsigned*)d->sched_state;
    if (d->c_colsum_partial) {
        if (colsum_rows_for(d) == 0 || !ucf_is_aligned16(d->c_colsum_partial)) {
            ucfvit_set_error("ucfvit_gemm: c_colsum_partial is not available for this problem (ask ucfvit_gemm_colsum_rows first)");
            return UCFVIT_ERR_UNSUPPORTED;
        }
        ep.cs_partial = d->c_colsum_partial;
    }
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


// Grouped launch: n <= 32 epilogue-free GEMMs with identical K, layouts and dtypes (the weight gradients of 1..8 Blocks) run
// as ONE persistent ping-pong launch over the union of their 256x256 tiles — no split-K, no partial-sum slabs.
extern "C" int ucfvit_gemm_grouped(const ucfvit_gemm_desc* descs, int64_t n, void* stream) {
    UCF_CHECK_ARG(descs && n >= 1 && n <= GROUP_MAX, "ucfvit_gemm_grouped: need 1..%d descriptors", GROUP_MAX);
    const ucfvit_gemm_desc& d0 = descs[0];
    bool fast = d0.dtype == UCFVIT_BF16 && pp_enabled() && d0.K >= 128;
    for (int64_t i = 0; i < n && fast; ++i) {
        const ucfvit_gemm_desc& d = descs[i];
        const int64_t a_contig = (d.a_layout == UCFVIT_LAYOUT_KC) ? d.K : d.M;
        const int64_t b_contig = (d.b_layout == UCFVIT_LAYOUT_KC) ? d.K : d.N;
        const int64_t a_bytes = ((d.a_layout == UCFVIT_LAYOUT_KC ? d.M : d.K) * d.lda) * 2;
        const int64_t b_bytes = ((d.b_layout == UCFVIT_LAYOUT_KC ? d.N : d.K) * d.ldb) * 2;
        fast = d.K == d0.K && d.dtype == d0.dtype && d.out_dtype == d0.out_dtype && d.a_layout == d0.a_layout && d.b_layout == d0.b_layout &&
               !d.bias && !d.residual && !d.aux_in && !d.aux_out && d.act == UCFVIT_ACT_NONE && d.alpha == 1.0f && d.A && d.B && d.C &&
               d.M >= 128 && d.N >= 128 && ucf_is_aligned16(d.A) && ucf_is_aligned16(d.B) && ucf_is_aligned16(d.C) && d.lda % 8 == 0 &&
               d.ldb % 8 == 0 && d.ldc % 8 == 0 && a_contig % 8 == 0 && b_contig % 8 == 0 && d.N % 8 == 0 && a_bytes < (1ll << 32) &&
               b_bytes < (1ll << 32) &&
               !((d.a_layout == UCFVIT_LAYOUT_KC || d.b_layout == UCFVIT_LAYOUT_KC) && d.K % 8 != 0);
    }
    hipStream_t s = (hipStream_t)stream;
    if (!fast) {   // not groupable: run them one by one through the ordinary dispatcher
        for (int64_t i = 0; i < n; ++i) {
            const int rc = ucfvit_gemm(&descs[i], stream);
            if (rc) return rc;
        }
        return UCFVIT_OK;
    }
    Groups3 gt;
    memset(&gt, 0, sizeof(gt));
    int tiles = 0;
    for (int64_t i = 0; i < n; ++i) {
        fill_problem(gt.p[i], &descs[i], tiles);
        tiles += gt.p[i].tiles_m * gt.p[i].tiles_n;
    }
    gt.n = (int)n;
    gt.total_tiles = tiles;
    Epi2 ep;
    memset(&ep, 0, sizeof(ep));
    ep.alpha = 1.0f;
    ep.sched = (unsigned*)d0.sched_state;
    const int K = (int)d0.K;
    const int kps = ((K + BK2 - 1) / BK2) * BK2;
    const int la = d0.a_layout, lb = d0.b_layout;
#define G3(LA_, LB_)                                                                                            \
    (d0.out_dtype == UCFVIT_F32 ? launch3g<LA_, LB_, float>(gt, K, ep, 1, kps, s) : launch3g<LA_, LB_, bf16>(gt, K, ep, 1, kps, s))
    if (la == 0 && lb == 0) return G3(0, 0);
    if (la == 0 && lb == 1) return G3(0, 1);
    if (la == 1 && lb == 1) return G3(1, 1);
    return G3(1, 0);
#undef G3
}

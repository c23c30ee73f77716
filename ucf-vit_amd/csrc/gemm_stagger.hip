// Staggered ping-pong bf16 GEMM for gfx950: the forward / data-gradient launches of the training step (KC x KC operands, bf16 out,
// one problem, K a multiple of 64) with the tile EPILOGUE HIDDEN UNDER THE PARTNER WAVE GROUP'S K LOOP.
//
// gemm3_kernel (gemm2.hip) runs both wave groups of a 256x256 tile through the K loop together and then through the epilogue together:
// for the time of the epilogue (5-8 us of a 30-us tile at K = 1024; 17 us with the GELU + gelu' arithmetic) no MFMA issues on the CU.
// Here the two groups keep the SAME persistent tile list, the SAME shared B stream in LDS and the SAME four barriers per K-step, but
// G1 (waves 4-7, rows 128-255 of every tile) runs E K-steps behind G0 (waves 0-3, rows 0-127):
//
//   stream step p of tile c :   0 .. E-1            E .. nk-1        nk .. nk+E-1
//   B K-tile in LDS         :   p                   p                p - nk            (the panel is streamed cyclically: E tiles twice)
//   G0                      :   K-step k = p        K-step k = p     epilogue of tile c, slice by slice
//   G1                      :   epilogue of c - 1   K-step k = p     K-step k = p - nk
//
// so every epilogue slice (LDS transpose of a 16 x 64 strip, bias / activation / residual arithmetic, full-line stores) of one group sits
// between two barriers of a K-step of the other group, whose MFMAs keep the matrix pipe busy.  G1 accumulates its rows in the rotated
// order k = E..nk-1, 0..E-1 (a fixed order: results are bitwise reproducible).  Nothing needs a second accumulator set: a group in its
// epilogue holds its finished tile in the accumulator registers and has no K-step of its own.
//
// (An asm store of more than 64 bits ends with `s_nop 1`: hipcc does not pad the hazard between an asm VMEM store and the next VALU
// write of its data registers.)
// Memory operations of an epilogue slice are inline asm (buffer loads / stores with hardware range checking, ds_write / ds_read of the
// wave-private staging rows) with hand-counted s_waitcnt: hipcc drains vmcnt(0) at every use of an ordinary load while an LDS-DMA is in
// flight, and here the B stream's DMA is ALWAYS in flight.  The sequence of vector-memory operations of an epilogue phase is fixed at
// compile time (EpiLog below: stores and loads are unconditional, out-of-range lanes are dropped by the buffer range check; DMA pieces
// that nobody needs at the very end of the tile list are issued anyway), so every wait is an exact count.
#include <stdlib.h>

#include "common.h"

#ifndef S5_ST_MOD
#define S5_ST_MOD " nt"      // C / aux are written once: non-temporal, so the dirty lines leave L2 gradually (see gemm2.hip:store_out16)
#endif

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ int s5_kc_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

// per-lane byte offsets of this wave's 4 DMA pieces (1 KiB each: 8 rows x 128 B) of its group's half of a 256-row KC operand K-tile
__device__ __forceinline__ void s5_offsets(unsigned (&off)[4], int64_t ld, int r0, int R, int grp, int w4, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = grp * 16 + w4 * 4 + i;
        const int row = idx * 8 + (lane >> 3);
        const int gslot = (lane & 7) ^ (row & 7);
        int gr = r0 + row;
        gr = gr < R ? gr : R - 1;
        off[i] = (unsigned)(((int64_t)gr * ld + gslot * 8) * 2);
    }
}
__device__ __forceinline__ void s5_issue(const char* __restrict__ base_k, const unsigned (&off)[4], char* lds, int grp, int w4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = grp * 16 + w4 * 4 + i;
        __builtin_amdgcn_global_load_lds((gptr_t)(base_k + off[i]), (lptr_t)(lds + idx * 1024), 16, 0, 0);
    }
}
__device__ __forceinline__ bf16x8 s5_frag(const char* lds, int rbase, int c, int lane) {
    const int g = lane >> 4, i = lane & 15;
    return *reinterpret_cast<const bf16x8*>(lds + s5_kc_off(rbase + i, 4 * c + g));
}

__device__ __forceinline__ i32x4 s5_rsrc(const void* p, unsigned bytes) {
    const uint64_t a = (uint64_t)p;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffff));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
constexpr unsigned S5_OOB = 0x80000000u;    // a byte offset beyond every descriptor's num_records (< 2^31): the lane is dropped

template <int N> __device__ __forceinline__ void s5_wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit count");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
#ifdef S5_STAMP
// diagnostic build (tools/stagger_stamps.py; never part of the product library): workgroup 0 records the shader clock behind every barrier
__device__ unsigned long long g_s5_stamps[2][2048];
__device__ int g_s5_stamp_n[2];
#define S5_STAMP_HERE()                                                                              \
    do {                                                                                             \
        if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) {                                           \
            const int gi_ = threadIdx.x >> 8;                                                        \
            const int n_ = g_s5_stamp_n[gi_];                                                        \
            if (n_ < 2048) {                                                                         \
                g_s5_stamps[gi_][n_] = __builtin_amdgcn_s_memtime();                                 \
                g_s5_stamp_n[gi_] = n_ + 1;                                                          \
            }                                                                                        \
        }                                                                                            \
    } while (0)
#else
#define S5_STAMP_HERE() do { } while (0)
#endif
#define s5_barrier()                          \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        asm volatile("" ::: "memory");        \
        __builtin_amdgcn_s_barrier();         \
        asm volatile("" ::: "memory");        \
        S5_STAMP_HERE();                      \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

template <int I> struct IC {
    static constexpr int value = I;
};
template <int B, int E_, typename F> __device__ __forceinline__ void s5_for(F&& f) {
    if constexpr (B < E_) {
        f(IC<B>{});
        s5_for<B + 1, E_>(f);
    }
}

enum { S5_PLAIN = 1, S5_RESIDUAL = 2, S5_GELU_SAVE_DERIV = 5, S5_MUL_AUX = 6 };

// The vector-memory operations one wave issues during an epilogue phase of E K-steps (4 E barrier intervals), in order, and the number
// issued so far at every point that waits for one of them.  Interval u = 4 j + r:
//   r == 0 : (OWN_B, see EpiLog) the 4 DMA pieces of this group's half of the next B K-tile; in the last step the 4 A pieces of the
//            group's next tile
//   passes q of this interval (16 passes of 8 rows x 64 columns per wave tile; pass q runs in interval q * 4E / 16):
//            input piece of pass q + PD, WAIT for the input of pass q (q >= PD), NS stores
//   r == 2 : (OWN_B, G1) WAIT for the B pieces of this step     r == 3 : WAIT for every DMA piece of this step
// The bias and the inputs of passes 0 .. PD-1 are loaded to registers at the start of the group's LAST K-step of the tile and have landed
// behind that step's closing vmcnt(0): the phase opens without a memory round trip.
template <int EPI, int E> struct EpiLog {
    static constexpr int U = 4 * E;
    static constexpr bool HAS_IN = EPI == S5_RESIDUAL || EPI == S5_MUL_AUX;
    static constexpr int NS = EPI == S5_GELU_SAVE_DERIV ? 2 : 1;
    static constexpr int PD = 2;
    // Who feeds the draining group's half of the B stream?  OWN_B: the draining group itself, 4 pieces in the first interval of every one
    // of its steps.  Its wait for the pieces of step j sits behind its stores of step j - 1 (vmcnt retires in order): harmless when those
    // are old by then — one step (E == 1: no older stores at all) or long arithmetic slices (the GELU epilogue: a slice per interval, four
    // intervals before the wait) — and a stall for the short slices of the other epilogues at E > 1, where the partner group, which is in
    // its K loop, issues both halves instead.
    static constexpr bool OWN_B = E == 1 || EPI == S5_GELU_SAVE_DERIV;
    int idxL[16] = {}, idxDmaB[E] = {}, idxDmaAll[E] = {};
    int nWL[16] = {}, nWB[E] = {}, nWD[E] = {};
    static constexpr int slot_of(int q) { return q * U / 16; }
    constexpr EpiLog() {
        int n = 0;
        for (int u = 0; u < U; ++u) {
            const int j = u / 4, r = u % 4;
            if (r == 0) {                // first interval of a step: (the B pieces,) in the last step the 4 A pieces of the group's next tile
                if (OWN_B) n += 4;
                idxDmaB[j] = n;
                if (j == E - 1) n += 4;
                idxDmaAll[j] = n;
            }
            for (int q = 0; q < 16; ++q) {
                if (slot_of(q) != u) continue;
                if (HAS_IN && q + PD < 16) {
                    n += 1;
                    idxL[q + PD] = n;
                }
                nWL[q] = n;
                n += NS;
            }
            if (r == 2) nWB[j] = n;
            if (r == 3) nWD[j] = n;
        }
    }
    constexpr int wait_in(int q) const { return nWL[q] - idxL[q]; }          // q >= PD only
    constexpr int wait_b(int j) const { return nWB[j] - idxDmaB[j]; }        // OWN_B, G1: its B pieces one interval before everything else
    constexpr int wait_all(int j) const { return nWD[j] - idxDmaAll[j]; }    // every DMA piece of step j (OWN_B: all steps; else the last)
};

template <int EPI, int E> inline constexpr EpiLog<EPI, E> kLog{};

struct StaggerArgs {
    const char* A;
    const char* B;
    bf16* C;
    const bf16* bias;      // [N], or the zero page (bias_bytes says how much of it may be read)
    const bf16* in;        // residual (S5_RESIDUAL) / aux_in (S5_MUL_AUX), [M][ldin]
    bf16* aux_out;         // gelu' (S5_GELU_SAVE_DERIV), [M][ldaux]
    float* cs_partial;     // [2 tiles_m][N] column sums (CS)
    int M, N, K;
    int64_t lda, ldb;
    int ldc, ldin, ldaux;
    unsigned bias_bytes;
    float alpha;
    int tiles_m, tiles_n, total_tiles;
};

__device__ __attribute__((aligned(16))) unsigned char g_s5_zero_page[1024];

// logical tile index -> (m0, n0): bands of 8 (or 4) N-tiles, walking down M inside a band — the order of gemm3_kernel
__device__ __forceinline__ void s5_tile_origin(int t, int tiles_m, int tiles_n, int& m0, int& n0) {
    const int BAND = (tiles_n > 8 && tiles_n % 8 != 0 && tiles_n % 4 == 0) ? 4 : 8;
    const int band_tiles = BAND * tiles_m;
    const int band = t / band_tiles;
    const int band_w = min(BAND, tiles_n - band * BAND);
    const int in_band = t - band * band_tiles;
    m0 = (in_band / band_w) * 256;
    n0 = (band * BAND + in_band % band_w) * 256;
}

// first half of an epilogue pass (see gemm5_kernel: `pass`): stage strip Q / 2 (even Q), read rows (Q & 1) * 8 + prow back transposed.
// (A free function, not a lambda: hipcc does not capture the operands of an asm statement inside a generic lambda that is instantiated
// from within another generic lambda.)
template <int Q>
__device__ __forceinline__ void s5_pass_fetch(unsigned st_w, unsigned st_r, const f32x4 (&acc)[8][4], f32x4 (&plo)[2], f32x4 (&phi)[2]) {
    constexpr int i = Q >> 1, ps = Q & 1, PADW = 68;
    if constexpr (ps == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(st_w), "v"(acc[i][j]), "i"(64 * j) : "memory");
    }
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(plo[Q & 1]) : "v"(st_r), "i"(ps * 8 * PADW * 4) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(phi[Q & 1]) : "v"(st_r), "i"(ps * 8 * PADW * 4 + 16) : "memory");
}

template <int EPI, bool CS, int E>
__global__ __launch_bounds__(512) void gemm5_kernel(StaggerArgs a) {
    constexpr int BM = 256, BN = 256, TM = 128, TN = 64, FM = 8, FN = 4;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
    constexpr int PADW = TN + 4, STAGE = 16 * PADW * 4;
    constexpr bool HAS_IN = EpiLog<EPI, E>::HAS_IN;
    constexpr int PD = EpiLog<EPI, E>::PD;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wm = grp * TM, wn = w4 * TN;
    const int g = lane >> 4, li = lane & 15;
    const int prow = lane >> 3, pcol = (lane & 7) * 8;
    const int nk = a.K >> 6;
    const int G = gridDim.x, nwg = a.total_tiles, bid = blockIdx.x;
    const int T = (nwg - bid + G - 1) / G;                     // tiles of this workgroup (the launch keeps G <= nwg)
    auto tile_mn = [&](int c, int& m0_, int& n0_) {
        const int cnt = min(G, nwg - c * G);
        s5_tile_origin(c * G + xcd_remap(bid, cnt), a.tiles_m, a.tiles_n, m0_, n0_);
    };

    // epilogue staging rows of this wave (wave w4 of whichever group is draining: the two groups never drain at the same time)
    const unsigned stage = (unsigned)(uintptr_t)LDS_PTR(char, smem) + 2 * BUF + w4 * STAGE;
    const unsigned st_w = stage + (li * PADW + 4 * g) * 4;       // + 64 j       : accumulator block j of a strip
    const unsigned st_r = stage + (prow * PADW + pcol) * 4;      // + ps 8 PADW 4: 8 columns of row ps * 8 + prow
    const i32x4 rC = s5_rsrc(a.C, (unsigned)((int64_t)a.M * a.ldc * 2));
    const i32x4 rBias = s5_rsrc(a.bias, a.bias_bytes);
    // The C-shaped input (residual / saved gelu'): passes 0 .. PD-1 of a tile get theirs in registers (loaded one K-step before the phase
    // opens, see EpiLog), the others by LDS-DMA into a wave-private ring of PD + 1 pieces of 1 KiB — lane l's 16 bytes of a piece are the 8
    // columns lane l works on (8 rows x 128 B per piece = the rows of one pass).  A loaded REGISTER must not stay un-waited across the
    // step loop's back edge (hipcc may copy it before the data has landed); LDS has no such hazard.
    const i32x4 rInR = s5_rsrc(HAS_IN ? (const void*)a.in : (const void*)a.C, HAS_IN ? (unsigned)((int64_t)a.M * a.ldin * 2) : 0u);
    const __amdgpu_buffer_rsrc_t rIn = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(HAS_IN ? a.in : (const bf16*)a.C), 0, HAS_IN ? (int)((int64_t)a.M * a.ldin * 2) : 0, 0x00020000);
    char* const inring = smem + 2 * BUF + 4 * STAGE + w4 * ((PD + 1) * 1024);
    const unsigned in_r = (unsigned)(uintptr_t)LDS_PTR(char, inring) + lane * 16;
    const i32x4 rAux = s5_rsrc(EPI == S5_GELU_SAVE_DERIV ? (const void*)a.aux_out : (const void*)a.C,
                               EPI == S5_GELU_SAVE_DERIV ? (unsigned)((int64_t)a.M * a.ldaux * 2) : 0u);

    unsigned offA[4], offB[4], offB2[4];      // offB2: the partner group's half of the B K-tile (issued while the partner drains)
    int m0, n0;
    tile_mn(0, m0, n0);
    s5_offsets(offB, a.ldb, n0, a.N, grp, w4, lane);
    s5_offsets(offB2, a.ldb, n0, a.N, grp ^ 1, w4, lane);
    s5_offsets(offA, a.lda, m0, a.M, grp, w4, lane);
    // prologue: stream step 0 = B K-tile 0 of the first panel (both halves), G0's A rows of K-tile 0 (G1's first K-step is step E)
    s5_issue(a.B, offB, smem + A_BYTES, grp, w4);
    if (grp == 0) s5_issue(a.A, offA, smem, grp, w4);
    s5_wait_vm<0>();
    s5_barrier();

    f32x4 acc[FM][FN];
    bf16x8 fa[FM], fb[FN];
    // state of the epilogue phase (set up in the group's last K-step of the tile)
    unsigned voC = 0, voIn = 0, voAux = 0;
    int em = 0;                      // this lane's output row of pass 0
    int etm0 = 0, etn0 = 0;          // origin of the tile being drained
    bool nok = false;
    u32x4 biasw = {0u, 0u, 0u, 0u};  // the 8 bias values of this lane's columns
    u32x4 early[PD];                 // the inputs of passes 0 .. PD-1
#pragma unroll
    for (int i = 0; i < PD; ++i) early[i] = u32x4{0u, 0u, 0u, 0u};
    float csum[8];

#define S5_READ(bufA_, bufB_, c_)                                                                  \
    do {                                                                                           \
        _Pragma("unroll") for (int j = 0; j < FN; ++j) fb[j] = s5_frag(bufB_, wn + 16 * j, c_, lane); \
        _Pragma("unroll") for (int i = 0; i < FM; ++i) fa[i] = s5_frag(bufA_, wm + 16 * i, c_, lane); \
    } while (0)
#define S5_COMPUTE()                                                                               \
    do {                                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                             \
        _Pragma("unroll") for (int i = 0; i < FM; ++i)                                             \
            _Pragma("unroll") for (int j = 0; j < FN; ++j)                                         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                             \
    } while (0)

    // One pass of the epilogue: rows ps * 8 + prow of strip i, 8 columns per lane.  Two halves so that the LDS round trip of pass q + 1
    // (stage the strip, read it back transposed) is in flight under the arithmetic of pass q:
    //   pass_fetch<q>: [even q: the 4 ds_write_b128 of strip q / 2]  the input piece of pass q + PD  2 ds_read_b128 -> plo / phi[q & 1]
    //   pass_math<q, AHEAD>: wait for those reads leaving the AHEAD younger LDS operations (the fetch of pass q + 1) in flight, arithmetic, stores
    f32x4 plo[2], phi[2];
    auto pass = [&](auto qc, auto aheadc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value, i = q >> 1, ps = q & 1, AHEAD = decltype(aheadc)::value;
        if constexpr (HAS_IN && q + PD < 16) {
            constexpr int qn = q + PD;
            const int mrow = em + 16 * (qn >> 1) + 8 * (qn & 1);
            const unsigned vo = (mrow < a.M && nok) ? voIn + (unsigned)((16 * (qn >> 1) + 8 * (qn & 1)) * a.ldin * 2) : S5_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rIn, (lptr_t)(inring + (qn % (PD + 1)) * 1024), 16, vo, 0, 0, 0);
        }
        u32x4 hin = {0u, 0u, 0u, 0u};
        if constexpr (HAS_IN && q >= PD) {
            constexpr int WAIT_IN = kLog<EPI, E>.wait_in(q);
            static_assert(WAIT_IN >= 0 && WAIT_IN <= 63, "vmcnt is a 6-bit count");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_IN) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(hin) : "v"(in_r), "i"((q % (PD + 1)) * 1024) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(plo[q & 1]), "+v"(phi[q & 1]), "+v"(hin)::"memory");
        } else {
            // LDS operations of one wave complete in order: AHEAD counts the ones issued after this pass's two reads (compiler-issued scalar
            // loads in between could only make the wait stricter)
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(plo[q & 1]), "+v"(phi[q & 1]) : "n"(AHEAD) : "memory");
            if constexpr (HAS_IN) hin = early[q];
        }
        const f32x4 lo = plo[q & 1], hi = phi[q & 1];
        float v[8];
        if constexpr (EPI == S5_MUL_AUX) {       // (data gradients carry no bias)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = lo[r] * a.alpha;
                v[4 + r] = hi[r] * a.alpha;
            }
        } else {
            const bf16x8 bb = __builtin_bit_cast(bf16x8, biasw);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = lo[r] * a.alpha + (float)bb[r];
                v[4 + r] = hi[r] * a.alpha + (float)bb[4 + r];
            }
        }
        const int m = em + 16 * i + 8 * ps;
        const bool inside = m < a.M && nok;
        const unsigned rowoff = (unsigned)(16 * i + 8 * ps);
        if constexpr (EPI == S5_GELU_SAVE_DERIV) {
            float df[8];
            gelu_and_grad_fast8(v, df);
            Vec16<bf16> o;
#pragma unroll
            for (int r = 0; r < 8; ++r) o.set(r, df[r]);
            const unsigned vo = inside ? voAux + rowoff * (unsigned)(a.ldaux * 2) : S5_OOB;
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" S5_ST_MOD "\n\ts_nop 1" ::"v"(o.v), "v"(vo), "s"(rAux) : "memory");
        } else if constexpr (EPI == S5_MUL_AUX || EPI == S5_RESIDUAL) {
            const bf16x8 h = __builtin_bit_cast(bf16x8, hin);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if constexpr (EPI == S5_MUL_AUX) v[r] *= (float)h[r];
                else v[r] += (float)h[r];
            }
        }
        Vec16<bf16> o;
#pragma unroll
        for (int r = 0; r < 8; ++r) o.set(r, v[r]);
        const unsigned vo = inside ? voC + rowoff * (unsigned)(a.ldc * 2) : S5_OOB;
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" S5_ST_MOD "\n\ts_nop 1" ::"v"(o.v), "v"(vo), "s"(rC) : "memory");
        if constexpr (CS) {
            const float keep = inside ? 1.0f : 0.0f;
#pragma unroll
            for (int r = 0; r < 8; ++r) csum[r] += keep * v[r];
        }
    };

    // epilogue step J (4 barrier intervals); the DMA of the next stream step goes first
    auto estep = [&](auto jc, const char* bnext, const char* anext, char* nb) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value;
        s5_for<0, 4>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value, u = 4 * J + r;
            if constexpr (r == 0) {
                if constexpr (EpiLog<EPI, E>::OWN_B) s5_issue(bnext, offB, nb + A_BYTES, grp, w4);
                if constexpr (J == E - 1) s5_issue(anext, offA, nb, grp, w4);
            }
            if constexpr (CS && u == 0) {
#pragma unroll
                for (int r2 = 0; r2 < 8; ++r2) csum[r2] = 0.f;
            }
            s5_for<0, 16>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                if constexpr (EpiLog<EPI, E>::slot_of(q) == u) {
                    constexpr bool first = q == 0 || EpiLog<EPI, E>::slot_of(q - 1) != u;
                    constexpr bool more = q + 1 < 16 && EpiLog<EPI, E>::slot_of(q + 1) == u;
                    if constexpr (first) s5_pass_fetch<q>(st_w, st_r, acc, plo, phi);
                    if constexpr (more) s5_pass_fetch<q + 1>(st_w, st_r, acc, plo, phi);   // in flight under this pass's arithmetic
                    pass(qc, IC<(more ? (((q + 1) & 1) == 0 ? 6 : 2) : 0)>{});
                }
            });
            if constexpr (EpiLog<EPI, E>::OWN_B && r == 2) {
                if (grp == 1) s5_wait_vm<kLog<EPI, E>.wait_b(J)>();
            }
            if constexpr (r == 3 && (EpiLog<EPI, E>::OWN_B || J == E - 1)) s5_wait_vm<kLog<EPI, E>.wait_all(J)>();
            if constexpr (CS && u == 4 * E - 1) {
                // lanes l, l + 8, ..., l + 56 hold the same 8 columns for different rows
#pragma unroll
                for (int r2 = 0; r2 < 8; ++r2) {
                    float t = csum[r2];
                    t += __shfl_xor(t, 8, 64);
                    t += __shfl_xor(t, 16, 64);
                    t += __shfl_xor(t, 32, 64);
                    csum[r2] = t;
                }
                const int n = etn0 + wn + pcol;
                if (prow == 0 && n < a.N) {
                    float* cp = a.cs_partial + (int64_t)((etm0 >> 7) + grp) * a.N + n;
                    *reinterpret_cast<f32x4*>(cp) = f32x4{csum[0], csum[1], csum[2], csum[3]};
                    *reinterpret_cast<f32x4*>(cp + 4) = f32x4{csum[4], csum[5], csum[6], csum[7]};
                }
            }
            s5_barrier();
        });
    };

    // last K-step of a group's tile: set the epilogue phase up and fetch what its first passes need
    auto epi_setup = [&]() __attribute__((always_inline)) {
        etm0 = m0;
        etn0 = n0;
        em = m0 + wm + prow;
        const int ncol = n0 + wn + pcol;
        nok = ncol < a.N;
        voC = (unsigned)(((int64_t)em * a.ldc + ncol) * 2);
        voIn = (unsigned)(((int64_t)em * a.ldin + ncol) * 2);
        voAux = (unsigned)(((int64_t)em * a.ldaux + ncol) * 2);
        if constexpr (EPI != S5_MUL_AUX) {
            const unsigned vb = nok ? (unsigned)(ncol * 2) : S5_OOB;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(biasw) : "v"(vb), "s"(rBias) : "memory");
        }
        if constexpr (HAS_IN) {
#pragma unroll
            for (int qn = 0; qn < PD; ++qn) {
                const int mrow = em + 16 * (qn >> 1) + 8 * (qn & 1);
                const unsigned vo = (mrow < a.M && nok) ? voIn + (unsigned)((16 * (qn >> 1) + 8 * (qn & 1)) * a.ldin * 2) : S5_OOB;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(early[qn]) : "v"(vo), "s"(rInR) : "memory");
            }
        }
    };

    int s = 0;                       // stream step counter: step s lives in pipeline buffer s & 1
    if (grp == 1) s5_barrier();      // G1 runs one barrier interval behind G0 (ping-pong of the paired K-steps)
    for (int c = 0;; ++c) {
        const bool tile_ok = c < T;
        const bool next_ok = c + 1 < T;
        const int plen = tile_ok ? nk + E : E;
        int nm0 = m0, nn0 = n0;
        if (next_ok) tile_mn(c + 1, nm0, nn0);
#pragma clang loop unroll(disable)
        for (int p = 0; p < plen; ++p, ++s) {
            if (p == E && tile_ok && nk > E) {
                // ---- the bulk of the tile: steps E .. nk-1, both groups in their K loop (G0 on k = p, G1 on the same stream step of
                // its own rows), nothing changes but the K position; in the last of them G0 prepares its drain and fetches no A.
                // (Kept apart from the general step below: its mode / operand bookkeeping per step is scalar work in front of
                // G0's DMA issue, on the critical path of the paired K-steps.)
                if (grp == 1) {
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma clang loop unroll(disable)
                for (; p < nk; ++p, ++s) {
                    const char* bufA = smem + (s & 1) * BUF;
                    const char* bufB = bufA + A_BYTES;
                    char* nb = smem + ((s + 1) & 1) * BUF;
                    const bool g0_last = grp == 0 && p == nk - 1;
                    const int kn = p == nk - 1 ? 0 : p + 1;            // the B panel is streamed cyclically
                    if (g0_last) epi_setup();
                    s5_issue(a.B + kn * 128, offB, nb + A_BYTES, grp, w4);
                    if (!g0_last) s5_issue(a.A + kn * 128, offA, nb, grp, w4);
                    S5_READ(bufA, bufB, 0);
                    s5_barrier();
                    S5_COMPUTE();
                    s5_barrier();
                    S5_READ(bufA, bufB, 1);
                    if (grp == 1) s5_wait_vm<4>();
                    s5_barrier();
                    S5_COMPUTE();
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (PD == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(biasw), "+v"(early[0]), "+v"(early[1])::"memory");
                    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(biasw), "+v"(early[0])::"memory");
                    __builtin_amdgcn_sched_barrier(0);
                    s5_barrier();
                }
            }
            const bool wrap = p + 1 == plen;
            const int pn = wrap ? 0 : p + 1;
            const bool nxt_tile = wrap ? next_ok : tile_ok;              // the next stream step belongs to an existing tile
            const char* bufA = smem + (s & 1) * BUF;
            const char* bufB = bufA + A_BYTES;
            char* nb = smem + ((s + 1) & 1) * BUF;
            // ---- modes
            const bool kmode = grp == 0 ? (tile_ok && p < nk) : (p >= E);
            const bool emode = grp == 0 ? (tile_ok && p >= nk) : (p < E && c > 0);
            const bool knext = nxt_tile && (grp == 0 ? pn < nk : pn >= E);
            // ---- operands of the next stream step
            if (wrap && next_ok) {
                s5_offsets(offB, a.ldb, nn0, a.N, grp, w4, lane);
                s5_offsets(offB2, a.ldb, nn0, a.N, grp ^ 1, w4, lane);
            }
            const int kb = pn < nk ? pn : pn - nk;
            const char* bnext = a.B + (nxt_tile ? kb * 128 : 0);
            if (knext && (grp == 0 ? pn == 0 : pn == E)) s5_offsets(offA, a.lda, wrap ? nm0 : m0, a.M, grp, w4, lane);
            const char* anext = a.A + (knext ? kb * 128 : 0);
            if (kmode) {
                if (grp == 0 ? p == 0 : (p == E && !(nk > E))) {
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (grp == 0 ? p == nk - 1 : p == nk + E - 1) epi_setup();
                s5_issue(bnext, offB, nb + A_BYTES, grp, w4);
                if constexpr (!EpiLog<EPI, E>::OWN_B) {
                    if (grp == 0 ? p < E : p >= nk) s5_issue(bnext, offB2, nb + A_BYTES, grp ^ 1, w4);     // the partner drains: its half too
                } else {
                    if (grp == 0 && c == 0 && p < E) s5_issue(bnext, offB2, nb + A_BYTES, grp ^ 1, w4);    // (G1 has not started yet)
                }
                if (knext) s5_issue(anext, offA, nb, grp, w4);
                S5_READ(bufA, bufB, 0);
                s5_barrier();
                S5_COMPUTE();
                s5_barrier();
                S5_READ(bufA, bufB, 1);
                if (grp == 1) {
                    if (knext) s5_wait_vm<4>();
                    else s5_wait_vm<0>();
                }
                s5_barrier();
                S5_COMPUTE();
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (PD == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(biasw), "+v"(early[0]), "+v"(early[1])::"memory");
                else asm volatile("s_waitcnt vmcnt(0)" : "+v"(biasw), "+v"(early[0])::"memory");
                __builtin_amdgcn_sched_barrier(0);
                s5_barrier();
            } else if (emode) {
                const int j = grp == 0 ? p - nk : p;
                switch (j) {
                    case 0: estep(IC<0>{}, bnext, anext, nb); break;
                    case 1: if constexpr (E > 1) estep(IC<1>{}, bnext, anext, nb); break;
                    case 2: if constexpr (E > 2) estep(IC<2>{}, bnext, anext, nb); break;
                    case 3: if constexpr (E > 3) estep(IC<3>{}, bnext, anext, nb); break;
                    case 4: if constexpr (E > 4) estep(IC<4>{}, bnext, anext, nb); break;
                    case 5: if constexpr (E > 5) estep(IC<5>{}, bnext, anext, nb); break;
                    case 6: if constexpr (E > 6) estep(IC<6>{}, bnext, anext, nb); break;
                    default: if constexpr (E > 7) estep(IC<7>{}, bnext, anext, nb); break;
                }
            } else if (grp == 1) {
                // G1 before its first tile (G0 feeds the whole B stream): in the last of these steps it fetches its first A K-tile
                if (knext) s5_issue(anext, offA, nb, grp, w4);
                s5_barrier();
                s5_barrier();
                s5_barrier();
                s5_wait_vm<0>();
                s5_barrier();
            } else {
                // G0 behind its last tile: G1 drains the last tile
                s5_barrier();
                s5_barrier();
                s5_barrier();
                s5_barrier();
            }
        }
        if (!tile_ok) break;
        m0 = nm0;
        n0 = nn0;
    }
    if (grp == 0) s5_barrier();
    s5_wait_vm<0>();                 // the DMA pieces nobody reads must still have landed before the LDS is given back
#undef S5_READ
#undef S5_COMPUTE
}

template <int EPI, bool CS, int E> int s5_launch(const StaggerArgs& a, hipStream_t s) {
    constexpr size_t smem = 2 * (size_t)(256 + 256) * 128 + 4 * 16 * 68 * 4 + 4 * 3 * 1024;
    static_assert(EpiLog<EPI, E>::PD == 2, "the input ring is sized for a prefetch distance of 2 passes");
    auto kern = gemm5_kernel<EPI, CS, E>;
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            ucfvit_set_error("ucfvit_gemm: cannot raise dynamic LDS to %zu bytes: %s", smem, hipGetErrorString(e));
            return UCFVIT_ERR_HIP;
        }
        done = true;
    }
    const int gx = a.total_tiles < 256 ? a.total_tiles : 256;
    hipLaunchKernelGGL(kern, dim3(gx), dim3(512), smem, s, a);
    UCF_LAUNCH_CHECK("ucfvit_gemm(staggered ping-pong)");
    return UCFVIT_OK;
}

// UCFVIT_GEMM_STAGGER — a TEST / measurement hook: 0 keeps every launch on gemm3_kernel (tests/test_hip_ops.py compares the dynamic tile
// schedule with the static order of the SAME kernel; tools/block_gemm_bench.py A/B), 1 / 2 / 4 / 8 force the number of epilogue steps
// (tests/test_gemm_stagger.py runs every variant).  Read once (thread-safe static).
int s5_steps_override() {
    static const int v = [] {
        const char* e = getenv("UCFVIT_GEMM_STAGGER");
        return e ? atoi(e) : -1;
    }();
    return v;
}

}  // namespace

#ifdef S5_STAMP
extern "C" int ucfvit_debug_stagger_stamps(unsigned long long* out, int* counts, int reset) {
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s5_stamps), sizeof(unsigned long long) * 2 * 2048);
    if (counts) (void)hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_s5_stamp_n), sizeof(int) * 2);
    if (reset) {
        const int z[2] = {0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_s5_stamp_n), z, sizeof(z));
    }
    return 0;
}
#endif

// returns 1 when the staggered kernel ran the problem, 0 when the caller should use gemm3_kernel, < 0 on error.
// The caller (gemm2.hip: launch3) has already checked the DMA path's alignment rules and that the 256x256 tile applies.
int ucfvit_gemm_stagger_try(const ucfvit_gemm_desc* d, hipStream_t s) {
    const int ov = s5_steps_override();
    if (ov == 0) return 0;
    if (d->dtype != UCFVIT_BF16 || d->out_dtype != UCFVIT_BF16 || d->a_layout != UCFVIT_LAYOUT_KC || d->b_layout != UCFVIT_LAYOUT_KC) return 0;
    if (d->accumulate || d->sched_state || d->K % 64 != 0 || d->N % 8 != 0) return 0;
    const int64_t lim = (1ll << 31) - 1;
    if (d->M * d->ldc * 2 > lim || d->M * d->lda * 2 >= (1ll << 32) || d->N * d->ldb * 2 >= (1ll << 32)) return 0;
    int epi;
    if (d->act == UCFVIT_ACT_NONE && !d->residual && !d->aux_out && !d->aux_in) epi = S5_PLAIN;
    else if (d->act == UCFVIT_ACT_NONE && d->residual && !d->aux_out && !d->aux_in) epi = S5_RESIDUAL;
    else if (d->act == UCFVIT_ACT_GELU_SAVE_DERIV && !d->residual && d->aux_out) epi = S5_GELU_SAVE_DERIV;
    else if (d->act == UCFVIT_ACT_MUL_AUX && !d->residual && !d->aux_out && d->aux_in) epi = S5_MUL_AUX;
    else return 0;
    if (d->c_colsum_partial && epi != S5_MUL_AUX) return 0;
    if (epi == S5_MUL_AUX && d->bias) return 0;             // (data gradients carry no bias: that epilogue has no bias registers)
    if (epi == S5_RESIDUAL && d->M * d->ldr * 2 > lim) return 0;
    if ((epi == S5_MUL_AUX || epi == S5_GELU_SAVE_DERIV) && d->M * d->ldaux * 2 > lim) return 0;
    const int nk = (int)(d->K / 64);
    StaggerArgs a;
    a.A = (const char*)d->A;
    a.B = (const char*)d->B;
    a.C = (bf16*)d->C;
    if (d->bias) {
        a.bias = (const bf16*)d->bias;
        a.bias_bytes = (unsigned)(d->N * 2);
    } else {
        static void* const zp = [] {               // looked up once (thread-safe static); nullptr: this launch stays on gemm3_kernel
            void* q = nullptr;
            return hipGetSymbolAddress(&q, HIP_SYMBOL(g_s5_zero_page)) == hipSuccess ? q : nullptr;
        }();
        if (!zp) return 0;
        a.bias = (const bf16*)zp;
        a.bias_bytes = 0;                                   // every lane out of range: the bias reads as zero
    }
    a.in = (const bf16*)(epi == S5_RESIDUAL ? d->residual : d->aux_in);
    a.aux_out = (bf16*)d->aux_out;
    a.cs_partial = d->c_colsum_partial;
    a.M = (int)d->M;
    a.N = (int)d->N;
    a.K = (int)d->K;
    a.lda = d->lda;
    a.ldb = d->ldb;
    a.ldc = (int)d->ldc;
    a.ldin = (int)(epi == S5_RESIDUAL ? d->ldr : d->ldaux);
    a.ldaux = (int)d->ldaux;
    a.alpha = d->alpha;
    a.tiles_m = (int)((d->M + 255) / 256);
    a.tiles_n = (int)((d->N + 255) / 256);
    a.total_tiles = a.tiles_m * a.tiles_n;
    const bool cs = d->c_colsum_partial != nullptr;
    // Epilogue steps.  Measured (tools/block_gemm_bench.py, ViT-L shapes at M = 131005; profiles/r03_a_*): a K-step with only ONE group
    // computing costs about what a paired K-step costs (the step is paced by the DMA round trip, not by the MFMAs), so the fewer such steps
    // the better: E = 1 wins everywhere it applies, larger E loses.  The residual epilogue only pays for itself behind a long K loop, and
    // the column-sum variant of the multiply epilogue does not fit the register budget at E = 1: both stay on gemm3_kernel.
    // Which launches: per-shape A/B at the shapes of all five workloads (profiles/r03_c_stagger_shapes.txt): the staggered kernel wins
    // 3-7 % at K >= 1536, 0-6 % at K = 1024 and LOSES 4-10 % at K = 768 / 512 (the cyclic re-read of a B K-tile and the two unpaired steps
    // per tile weigh 1 / nk): K >= 1024 only.
    int E = 1;                                       // (fc1 forward with the GELU epilogue, E = 1 / 2 / 4 / 8: 1251 / 1262 / 1317 / 1652 us)
    if (ov > 0) E = ov;
    else if (nk < 16) return 0;
    else if (epi == S5_RESIDUAL && nk < 32) return 0;
    else if (cs) return 0;
    if (nk < 2 * E) E = nk >= 8 ? 4 : (nk >= 4 ? 2 : (nk >= 2 ? 1 : 0));
    if (E == 0 || (cs && E == 1)) return 0;
    int rc;
#define S5_GO(EPI_, CS_)                                                               \
    (E == 1 ? s5_launch<EPI_, CS_, 1>(a, s) : E == 2 ? s5_launch<EPI_, CS_, 2>(a, s) : \
     E == 4 ? s5_launch<EPI_, CS_, 4>(a, s) : s5_launch<EPI_, CS_, 8>(a, s))
    if (epi == S5_PLAIN) rc = S5_GO(S5_PLAIN, false);
    else if (epi == S5_RESIDUAL) rc = S5_GO(S5_RESIDUAL, false);
    else if (epi == S5_GELU_SAVE_DERIV) rc = S5_GO(S5_GELU_SAVE_DERIV, false);
    else rc = cs ? S5_GO(S5_MUL_AUX, true) : S5_GO(S5_MUL_AUX, false);
#undef S5_GO
    return rc == UCFVIT_OK ? 1 : rc;
}

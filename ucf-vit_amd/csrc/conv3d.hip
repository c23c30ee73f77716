// Convolutions of the UNETR decoder for gfx950: 3x3x3 (stride 1, zero padding 1) and 1x1x1 as implicit GEMMs on
// v_mfma_f32_16x16x32_bf16, plus the data movers of the 2x2x2 stride-2 transposed convolution (which itself is a 1x1x1 convolution to
// 8 Cout channels, or a plain GEMM when the channel counts are large).  The 1x1x1 case exists because the decoder's pointwise layers are
// tall and skinny (67 M voxels x 8..32 channels): a 128 x 128-tiled GEMM wastes 8..16x of its MFMA work on them and a weight gradient
// with one output tile has no parallelism at all; here the voxel axis is the tiled one.
//
//   reference call sites: src/UCF_VIT/simple/arch.py:808-940 — monai UnetrBasicBlock / UnetrPrUpBlock / UnetrUpBlock are chains of
//   Conv3d(k=3, s=1, p=1, bias=False) and ConvTranspose3d(k=2, s=2, bias=False).  monai is absent from the build container (PARITY UNPINNED
//   against it); the oracle is torch.nn.functional.conv3d / conv_transpose3d on the same bf16-rounded operands (tests/test_conv3d.py).
//
// Layout: activations are CHANNELS-LAST bf16, [B][X][Y][Z][C] — a voxel's channel vector is contiguous, so the encoder's token matrices
// [B, N, D] ARE decoder inputs without a permute, an MFMA operand fragment (8 channels of one voxel) is one 16-byte load, and an output
// block of 16 voxels x 16 channels is one contiguous 512-byte run.  Weights arrive pre-packed per 32-wide contraction step (see
// UCF_VIT/_hip/conv.py:pack_conv3_weight): step s of channel chunk cc holds, for every output channel, 32 values over (tap, ci).
//
// forward / data gradient (same kernel; the data gradient is the convolution of dy with the flipped, transposed weights):
//   one workgroup = TX x TY x 16 output voxels x 16 NB output channels.  The (TX+2)(TY+2)(18) halo of the input chunk (<= 32 channels)
//   is staged in LDS once and read 27 times (once per tap) as the B operand; the weight slab of the chunk is staged next to it (A operand).
//   D[co][voxel] -> each lane holds 4 consecutive output channels of one voxel: 8-byte stores, 512 B contiguous per 16 x 16 block.
//   algorithmic HBM bytes per output voxel: 2 Cin (read once; the halo re-reads are L2 hits) + 2 Cout.
// weight gradient: dW[tap][co][ci] = sum_v dy[v][co] x[v + tap][ci]: contraction over voxels, so both operands come out of their
//   voxel-major LDS images through ds_read_b64_tr_b16 (hardware transpose).  A workgroup walks a range of voxel tiles with the 27 x
//   (Cout/16) x (Cin/16) accumulator blocks spread over its 4 waves (7 taps each), and writes ONE partial per workgroup; the partials are
//   folded by ucfvit_reduce_rows in a fixed order (deterministic, no atomics).
#include "common.h"

namespace {

constexpr int CT = 256;
typedef bf16x8 frag_t;

__device__ __forceinline__ f32x4 mma(const frag_t& a, const frag_t& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int CPC, int KS> struct CG {                // CPC = channels per contraction chunk (8, 16 or 32), KS = kernel size (3 or 1)
    static constexpr int NT = KS * KS * KS;           // taps
    static constexpr int PAD = KS / 2;
    static constexpr int TPS = 32 / CPC;              // taps folded into one 32-wide MFMA step
    static constexpr int NTS = (NT + TPS - 1) / TPS;  // MFMA steps per chunk (KS 3: 27, 14, 7; KS 1: 1)
    static constexpr int VS = CPC * 2;                // bytes per voxel in the LDS image
    static constexpr int PPV = VS / 16;               // 16-byte pieces per voxel
};

struct ConvGeo {
    int B, X, Y, Z, Cin, Cout;
    int tx, ty, tz;      // tile counts along x, y, z
    int tiles;           // B tx ty tz
    int ldy, cout_store; // forward: output row stride (elements) and number of channels written (<= Cout)
    int accumulate;      // forward: y += result (the second data gradient of an input two layers consume)
};

// 16-byte load of 8 channels of voxel (b, gx, gy, gz), zero outside the volume (= the convolution's zero padding)
__device__ __forceinline__ u32x4 load_voxel(const bf16* __restrict__ x, const ConvGeo& g, int b, int gx, int gy, int gz, int C, int ch) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if ((unsigned)gx < (unsigned)g.X && (unsigned)gy < (unsigned)g.Y && (unsigned)gz < (unsigned)g.Z)
        v = *reinterpret_cast<const u32x4*>(x + ((((int64_t)b * g.X + gx) * g.Y + gy) * g.Z + gz) * C + ch);
    return v;
}

template <int CPC, int PAD, int HX, int HY, int HZ>
__device__ __forceinline__ void stage_halo(char* halo, const bf16* __restrict__ x, const ConvGeo& g, int b, int x0, int y0, int z0, int ch0,
                                           int tid) {
    constexpr int VS = CPC * 2, PPV = VS / 16;
    constexpr int NP = HX * HY * HZ * PPV;
    for (int p = tid; p < NP; p += CT) {
        const int hv = p / PPV, piece = p % PPV;
        const int hz = hv % HZ, hy = (hv / HZ) % HY, hx = hv / (HZ * HY);
        *reinterpret_cast<u32x4*>(halo + hv * VS + piece * 16) =
            load_voxel(x, g, b, x0 + hx - PAD, y0 + hy - PAD, z0 + hz - PAD, g.Cin, ch0 + piece * 8);
    }
}

__device__ __forceinline__ void decode_tile(const ConvGeo& g, int t, int& b, int& ix, int& iy, int& iz) {
    iz = t % g.tz;
    t /= g.tz;
    iy = t % g.ty;
    t /= g.ty;
    ix = t % g.tx;
    b = t / g.tx;
}

// One partial row of the statistics epilogue ([column * 4 + wave][3][Cout] fp32): per channel the COUNT of outputs this wave stored, their
// MEAN and M2 = sum (q - mean)^2.  The sums are accumulated relative to a per-(wave, channel) shift — the wave's first computed output of
// that channel — so a channel whose |mean| is far larger than its spread keeps its variance bits (E[q^2] - mean^2 in fp32 does not);
// ucfvit_instnorm_cl_stats_fold combines the rows with the parallel-variance formula in double.
__device__ __forceinline__ void stats_row_write(float* __restrict__ sp, int Cout, int c, float s1, float s2, float shift, float n, bool writer) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {                          // over the 16 voxel lanes of a channel group
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if (writer) {
        const float inv = n > 0.f ? 1.f / n : 0.f;
        sp[c] = n;
        sp[Cout + c] = n > 0.f ? shift + s1 * inv : 0.f;
        sp[2 * Cout + c] = fmaxf(s2 - s1 * s1 * inv, 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------------- forward / data gradient
template <int CPC, int NB, int TX, int TY, int KS, typename OutT>
__global__ __launch_bounds__(CT) void conv_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wp, const float* __restrict__ bias,
                                                      OutT* __restrict__ y, ConvGeo g) {
    typedef CG<CPC, KS> G;
    constexpr int PAD = G::PAD, HX = TX + 2 * PAD, HY = TY + 2 * PAD, HZ = 16 + 2 * PAD;
    constexpr int HALO_BYTES = HX * HY * HZ * G::VS;
    constexpr int CB = 16 * NB;                     // output channels per workgroup
    constexpr int RPW = TX * TY / 4;                // 16-voxel rows per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* wl = smem + HALO_BYTES;                   // [NTS][CB][32] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    int b, ix, iy, iz;
    decode_tile(g, xcd_remap(blockIdx.x, gridDim.x), b, ix, iy, iz);
    const int x0 = ix * TX, y0 = iy * TY, z0 = iz * 16;
    const int co0 = blockIdx.y * CB;
    f32x4 acc[RPW][NB];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[r][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nchunks = g.Cin / CPC;
    for (int cc = 0; cc < nchunks; ++cc) {
        if (cc) __syncthreads();
        stage_halo<CPC, PAD, HX, HY, HZ>(halo, x, g, b, x0, y0, z0, cc * CPC, tid);
        for (int p = tid; p < G::NTS * CB * 4; p += CT) {
            const int piece = p & 3, row = (p >> 2) % CB, ts = (p >> 2) / CB;
            *reinterpret_cast<u32x4*>(wl + p * 16) =
                *reinterpret_cast<const u32x4*>(wp + ((int64_t)(cc * G::NTS + ts) * g.Cout + co0 + row) * 32 + piece * 8);
        }
        __syncthreads();
#pragma unroll
        for (int ts = 0; ts < G::NTS; ++ts) {
            frag_t a[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) a[nb] = *reinterpret_cast<const frag_t*>(wl + ((ts * CB + nb * 16 + li) * 64 + lg * 16));
            int tap, chb;                                       // the tap and channel byte offset this lane's 8 contraction values cover
            if (CPC == 32) {
                tap = ts;
                chb = lg * 16;
            } else if (CPC == 16) {
                tap = 2 * ts + (lg >> 1);
                chb = (lg & 1) * 16;
            } else {
                tap = 4 * ts + lg;
                chb = 0;
            }
            if (tap > G::NT - 1) tap = G::NT - 1;               // padding step: its weights are zero, read any staged voxel
            const int dx = KS == 3 ? tap / 9 : 0, dy = KS == 3 ? (tap / 3) % 3 : 0, dz = KS == 3 ? tap % 3 : 0;
            const int boff = ((dx * HY + dy) * HZ + dz + li) * G::VS + chb;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = wave * RPW + r, xl = row / TY, yl = row % TY;
                const frag_t bf = *reinterpret_cast<const frag_t*>(halo + boff + (xl * HY + yl) * HZ * G::VS);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[r][nb] = mma(a[nb], bf, acc[r][nb]);
            }
        }
    }
    // D[co = 4 lg + e][voxel = li]: 4 consecutive channels of one voxel per lane
    const bool vec_ok = (g.ldy & 3) == 0;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = wave * RPW + r, gx = x0 + row / TY, gy = y0 + row % TY, gz = z0 + li;
        if (gx < g.X && gy < g.Y && gz < g.Z) {
            OutT* yp = y + ((((int64_t)b * g.X + gx) * g.Y + gy) * g.Z + gz) * g.ldy;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int c = co0 + nb * 16 + lg * 4;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[r][nb][e] + (bias ? bias[c + e] : 0.f);
                if (c + 4 <= g.cout_store && vec_ok) {
                    Vec4<OutT> o;
                    if (g.accumulate) {
                        o = *reinterpret_cast<const Vec4<OutT>*>(yp + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += o.get(e);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.set(e, v[e]);
                    *reinterpret_cast<Vec4<OutT>*>(yp + c) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < g.cout_store) yp[c + e] = from_f32<OutT>(v[e] + (g.accumulate ? to_f32<OutT>(yp[c + e]) : 0.f));
                }
            }
        }
    }
}

// ---- the same computation for a single-chunk input (Cin <= 32: every layer at the two finest resolutions), software-pipelined: a workgroup
// walks the whole z extent of its (x, y) tile column.  The weight slab is staged once; the halo of tile z + 1 is fetched into registers while
// tile z is multiplied out of LDS, so the global-load latency (the dominant cost of the one-tile-per-workgroup kernel at these sizes:
// ~60..100 MFMAs per wave per tile) is hidden behind the MFMA work.
// FAST (picked by the launcher when the output rows are 4-channel aligned and every tensor of a batch element is below 4 GB): the halo loads,
// the loads of the values to accumulate onto and the output stores are range-checked RAW BUFFER operations whose offset is pushed out of range
// for voxels outside the volume — no branch around any memory operation, so hipcc counts them exactly (vmcnt(n) instead of vmcnt(0) at the
// halo store) and DEPTH = 2 halo tiles can be in flight per workgroup instead of one (the kernel waits on global memory, see
// profiles/r03_n_decoder_strip_kernel.txt).
template <int CPC, int NB, int TX, int TY, int KS, typename OutT, bool FAST = false, int DEPTH = 1>
__global__ __launch_bounds__(CT) void conv_fwd_strip_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wp, const float* __restrict__ bias,
                                                            OutT* __restrict__ y, ConvGeo g, float* __restrict__ stats) {
    static_assert(FAST || DEPTH == 1, "two tiles in flight need the branch-free memory operations");
    typedef CG<CPC, KS> G;
    constexpr int PAD = G::PAD, HX = TX + 2 * PAD, HY = TY + 2 * PAD, HZ = 16 + 2 * PAD;
    constexpr int HALO_BYTES = HX * HY * HZ * G::VS;
    constexpr int CB = 16 * NB;
    constexpr int RPW = TX * TY / 4;
    constexpr int NP = HX * HY * HZ * G::PPV;          // 16-byte pieces of one halo
    constexpr int NPT = (NP + CT - 1) / CT;            // per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* wl = smem + HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    int t = xcd_remap(blockIdx.x, gridDim.x);          // (b, ix, iy) column
    const int iy = t % g.ty;
    t /= g.ty;
    const int ix = t % g.tx, b = t / g.tx;
    const int x0 = ix * TX, y0 = iy * TY;
    const int co0 = blockIdx.y * CB;
    // SHARE (3x3x3, 16 input channels): a wave's RPW voxel rows are consecutive in y at one x, and the B fragment of halo row hy is the
    // same for (output row yl, tap dy) whenever yl + dy = hy — read once, it feeds up to three MFMAs (the kernel was LDS-read bound: one
    // ds_read_b128 per MFMA at 16 output channels).  For that the taps folded into one 32-wide step must not mix dy: the step types are the
    // (dx, dz) pairs {00,01} {10,11} {20,21} {02,12} {22,-} for each dy (15 steps instead of 14), re-assembled from the packed weights while
    // they are staged.
    // (Measured at 512 x 512 x 128, B = 2: the 16 -> 16 full-resolution layer 1567 -> 1440 us, 16 -> 32 3154 -> 3022; with 32 input channels
    // — two MFMAs per fragment already — 544 -> 569 us, so those keep one fragment per (row, tap).)
    constexpr bool SHARE = KS == 3 && CPC == 16 && RPW <= TY && TY % RPW == 0;
    constexpr int NST = CPC == 16 ? 5 : 9;              // step types (dy-free), SHARE only
    if constexpr (SHARE && CPC == 16) {
        for (int p = tid; p < NST * 3 * CB * 4; p += CT) {
            const int piece = p & 3, row = (p >> 2) % CB, slot = (p >> 2) / CB, st = slot / 3, dy = slot % 3, second = piece >> 1;
            // (dx, dz) of the two taps of step type st: 0x(dx << 2 | dz) per half
            const int code = second ? ((0x1 << 0) | (0x5 << 4) | (0x9 << 8) | (0x6 << 12) | (0xF << 16)) : ((0x0 << 0) | (0x4 << 4) | (0x8 << 8) | (0x2 << 12) | (0xA << 16));
            const int dxdz = (code >> (4 * st)) & 0xF;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (dxdz != 0xF) {
                const int tap = 9 * (dxdz >> 2) + 3 * dy + (dxdz & 3);
                v = *reinterpret_cast<const u32x4*>(wp + ((int64_t)(tap >> 1) * g.Cout + co0 + row) * 32 + (tap & 1) * 16 + (piece & 1) * 8);
            }
            *reinterpret_cast<u32x4*>(wl + p * 16) = v;
        }
    } else {
        for (int p = tid; p < G::NTS * CB * 4; p += CT) {
            const int piece = p & 3, row = (p >> 2) % CB, ts = (p >> 2) / CB;
            *reinterpret_cast<u32x4*>(wl + p * 16) = *reinterpret_cast<const u32x4*>(wp + ((int64_t)ts * g.Cout + co0 + row) * 32 + piece * 8);
        }
    }
    // this thread's pieces of a halo: fixed (hx, hy, hz, piece) for every z tile
    int hoff[NPT], hxy[NPT], hzz[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int p = tid + i * CT;
        const int hv = p / G::PPV, piece = p % G::PPV;
        const int hz = hv % HZ, hy = (hv / HZ) % HY, hx = hv / (HZ * HY);
        const int gx = x0 + hx - PAD, gy = y0 + hy - PAD;
        const bool ok = p < NP && (unsigned)gx < (unsigned)g.X && (unsigned)gy < (unsigned)g.Y;
        hoff[i] = hv * G::VS + piece * 16;
        hxy[i] = ok ? (gx * g.Y + gy) : -1;             // row of voxels along z, -1: outside (zero padding) or no piece
        hzz[i] = ((hz - PAD) & 0xffff) | (piece << 16);
    }
    const bf16* xb = x + (int64_t)b * g.X * g.Y * g.Z * g.Cin;
    OutT* yb = y + (int64_t)b * g.X * g.Y * g.Z * g.ldy;
    constexpr unsigned OOB = 0xFFFFFFF0u;              // beyond every resource below: loads return 0, stores are dropped
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(xb), 0, FAST ? (int)((unsigned)g.X * g.Y * g.Z * g.Cin * 2u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(yb, 0, FAST ? (int)((unsigned)g.X * g.Y * g.Z * g.ldy * (unsigned)sizeof(OutT)) : 0, 0x00020000);
    u32x4 pre[DEPTH][NPT];
    auto fetch = [&](int z0, u32x4 (&dst)[NPT]) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int gz = z0 + (int)(short)(hzz[i] & 0xffff), piece = hzz[i] >> 16;
            if constexpr (FAST) {
                const bool ok = hxy[i] >= 0 && (unsigned)gz < (unsigned)g.Z;
                const unsigned off = (((unsigned)hxy[i] * (unsigned)g.Z + (unsigned)gz) * (unsigned)g.Cin + (unsigned)piece * 8u) * 2u;
                dst[i] = __builtin_amdgcn_raw_buffer_load_b128(rX, (int)(ok ? off : OOB), 0, 0);
            } else {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (hxy[i] >= 0 && (unsigned)gz < (unsigned)g.Z)
                    v = *reinterpret_cast<const u32x4*>(xb + ((int64_t)hxy[i] * g.Z + gz) * g.Cin + piece * 8);
                dst[i] = v;
            }
        }
    };
    fetch(0, pre[0]);
    if constexpr (DEPTH == 2) fetch(16, pre[1]);       // (beyond the last tile: out of range, no traffic)
    const bool vec_ok = (g.ldy & 3) == 0;
    float bv[NB][4];                                    // this lane's bias values (FAST: fetched once, no load under a branch in the tile loop)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[nb][e] = (FAST && bias) ? bias[co0 + nb * 16 + lg * 4 + e] : 0.f;
    // stats (optional): per-channel sum and sum of squares of the ROUNDED outputs of this column, per wave — the instance-norm statistics of the
    // layer's output without a pass over it ([column * 4 + wave][2][Cout] partials, folded by ucfvit_instnorm_cl_stats_fold)
    float st1[NB][4], st2[NB][4], shf[NB][4];
    float cnt = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) st1[nb][e] = st2[nb][e] = shf[nb][e] = 0.f;
    // one z tile: its halo (in `buf`) into LDS, the registers of `buf` re-used for the tile DEPTH ahead, MFMAs, epilogue
    auto tile = [&](int iz, u32x4 (&buf)[NPT]) __attribute__((always_inline)) {
        const int z0 = iz * 16;
#pragma unroll
        for (int i = 0; i < NPT; ++i)
            if (tid + i * CT < NP) *reinterpret_cast<u32x4*>(halo + hoff[i]) = buf[i];
        __syncthreads();
        Vec4<OutT> prev[RPW][NB];                       // accumulate: the values to add to, fetched now and used in the epilogue
        if constexpr (!FAST) {
            if (iz + 1 < g.tz) fetch(z0 + 16, buf);     // in flight during the MFMA work below
        }
        if constexpr (FAST) {
            // (requested BEFORE the halo tiles: the epilogue's wait for them must not include the younger halo loads — vmcnt retires in order)
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = wave * RPW + r, gx = x0 + row / TY, gy = y0 + row % TY, gz = z0 + li;
                const bool valid = g.accumulate && gx < g.X && gy < g.Y && gz < g.Z;
                const unsigned vox = ((unsigned)gx * (unsigned)g.Y + (unsigned)gy) * (unsigned)g.Z + (unsigned)gz;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int c = co0 + nb * 16 + lg * 4;
                    const unsigned off = (vox * (unsigned)g.ldy + (unsigned)c) * (unsigned)sizeof(OutT);
                    const int o = (int)((valid && c + 4 <= g.cout_store) ? off : OOB);
                    if constexpr (sizeof(OutT) == 2) prev[r][nb] = __builtin_bit_cast(Vec4<OutT>, __builtin_amdgcn_raw_buffer_load_b64(rY, o, 0, 0));
                    else prev[r][nb] = __builtin_bit_cast(Vec4<OutT>, __builtin_amdgcn_raw_buffer_load_b128(rY, o, 0, 0));
                }
            }
            fetch(z0 + 16 * DEPTH, buf);                // DEPTH tiles ahead, into the registers the halo store above has just freed
        } else if (g.accumulate && vec_ok) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = wave * RPW + r, gx = x0 + row / TY, gy = y0 + row % TY, gz = z0 + li;
                if (gx < g.X && gy < g.Y && gz < g.Z) {
                    const OutT* yp = y + ((((int64_t)b * g.X + gx) * g.Y + gy) * g.Z + gz) * g.ldy;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int c = co0 + nb * 16 + lg * 4;
                        if (c + 4 <= g.cout_store) prev[r][nb] = *reinterpret_cast<const Vec4<OutT>*>(yp + c);
                    }
                }
            }
        }
        f32x4 acc[RPW][NB];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[r][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (SHARE) {
            const int xl = (wave * RPW) / TY, yl0 = (wave * RPW) % TY;
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                frag_t a[3][NB];
                int dx, dz, chb;
                if (CPC == 32) {
                    dx = st / 3;
                    dz = st % 3;
                    chb = lg * 16;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            a[dy][nb] = *reinterpret_cast<const frag_t*>(wl + (((9 * dx + 3 * dy + dz) * CB + nb * 16 + li) * 64 + lg * 16));
                } else {
                    // this lane's half of the step: lanes lg 0, 1 the first tap, lg 2, 3 the second (the padding half reads the first tap's voxels
                    // against zero weights)
                    const int second = lg >> 1;
                    const int dxdz = st == 0 ? (second ? 0x1 : 0x0) : st == 1 ? (second ? 0x5 : 0x4) : st == 2 ? (second ? 0x9 : 0x8)
                                   : st == 3 ? (second ? 0x6 : 0x2) : 0xA;
                    dx = dxdz >> 2;
                    dz = dxdz & 3;
                    chb = (lg & 1) * 16;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            a[dy][nb] = *reinterpret_cast<const frag_t*>(wl + (((st * 3 + dy) * CB + nb * 16 + li) * 64 + lg * 16));
                }
                const int boff = (((dx + xl) * HY + yl0) * HZ + dz + li) * G::VS + chb;
#pragma unroll
                for (int j = 0; j < RPW + 2; ++j) {
                    const frag_t bf = *reinterpret_cast<const frag_t*>(halo + boff + j * HZ * G::VS);
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int r = j - dy;
                        if (r >= 0 && r < RPW) {
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) acc[r][nb] = mma(a[dy][nb], bf, acc[r][nb]);
                        }
                    }
                }
            }
        } else {
#pragma unroll
        for (int ts = 0; ts < G::NTS; ++ts) {
            frag_t a[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) a[nb] = *reinterpret_cast<const frag_t*>(wl + ((ts * CB + nb * 16 + li) * 64 + lg * 16));
            int tap, chb;
            if (CPC == 32) {
                tap = ts;
                chb = lg * 16;
            } else if (CPC == 16) {
                tap = 2 * ts + (lg >> 1);
                chb = (lg & 1) * 16;
            } else {
                tap = 4 * ts + lg;
                chb = 0;
            }
            if (tap > G::NT - 1) tap = G::NT - 1;
            const int dx = KS == 3 ? tap / 9 : 0, dy = KS == 3 ? (tap / 3) % 3 : 0, dz = KS == 3 ? tap % 3 : 0;
            const int boff = ((dx * HY + dy) * HZ + dz + li) * G::VS + chb;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = wave * RPW + r, xl = row / TY, yl = row % TY;
                const frag_t bf = *reinterpret_cast<const frag_t*>(halo + boff + (xl * HY + yl) * HZ * G::VS);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[r][nb] = mma(a[nb], bf, acc[r][nb]);
            }
        }
        }
        if (stats && iz == 0) {
            // the shift of the statistics: this wave's first computed output per channel (lane li = 0 of each channel group), rounded like a
            // stored one; any finite value near the data serves (a voxel outside the volume computes from the zero padding)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = acc[0][nb][e] + (FAST ? bv[nb][e] : (bias ? bias[co0 + nb * 16 + lg * 4 + e] : 0.f));
                    shf[nb][e] = __shfl(to_f32<OutT>(from_f32<OutT>(v0)), lane & 48, 64);
                }
        }
        if constexpr (FAST) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = wave * RPW + r, gx = x0 + row / TY, gy = y0 + row % TY, gz = z0 + li;
                const bool valid = gx < g.X && gy < g.Y && gz < g.Z;
                const unsigned vox = ((unsigned)gx * (unsigned)g.Y + (unsigned)gy) * (unsigned)g.Z + (unsigned)gz;
                if (stats && valid) cnt += 1.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int c = co0 + nb * 16 + lg * 4;
                    Vec4<OutT> o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.set(e, acc[r][nb][e] + bv[nb][e] + prev[r][nb].get(e));      // prev: zeros unless accumulating
                    const unsigned off = (vox * (unsigned)g.ldy + (unsigned)c) * (unsigned)sizeof(OutT);
                    const int ob = (int)((valid && c + 4 <= g.cout_store) ? off : OOB);
                    if constexpr (sizeof(OutT) == 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rY, ob, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rY, ob, 0, 0);
                    if (stats && valid) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float q = o.get(e) - shf[nb][e];
                            st1[nb][e] += q;
                            st2[nb][e] = fmaf(q, q, st2[nb][e]);
                        }
                    }
                }
            }
        } else {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = wave * RPW + r, gx = x0 + row / TY, gy = y0 + row % TY, gz = z0 + li;
            if (gx < g.X && gy < g.Y && gz < g.Z) {
                if (stats) cnt += 1.f;
                OutT* yp = y + ((((int64_t)b * g.X + gx) * g.Y + gy) * g.Z + gz) * g.ldy;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int c = co0 + nb * 16 + lg * 4;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[r][nb][e] + (bias ? bias[c + e] : 0.f);
                    if (c + 4 <= g.cout_store && vec_ok) {
                        Vec4<OutT> o;
                        if (g.accumulate) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += prev[r][nb].get(e);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) o.set(e, v[e]);
                        *reinterpret_cast<Vec4<OutT>*>(yp + c) = o;
                        if (stats) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float q = o.get(e) - shf[nb][e];
                                st1[nb][e] += q;
                                st2[nb][e] = fmaf(q, q, st2[nb][e]);
                            }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c + e < g.cout_store) yp[c + e] = from_f32<OutT>(v[e] + (g.accumulate ? to_f32<OutT>(yp[c + e]) : 0.f));
                    }
                }
            }
        }
        }
        __syncthreads();                                // every wave is done with this halo before the next one is written
    };
    if constexpr (DEPTH == 2) {
        // two tiles per trip, each with its own register set (no copy between them: a copy would wait for the younger tile's loads); with an
        // odd tile count the last trip's second tile lies beyond the volume: its loads and stores are out of range, its arithmetic is wasted
        for (int iz = 0; iz < g.tz; iz += 2) {
            tile(iz, pre[0]);
            tile(iz + 1, pre[1]);
        }
    } else {
        for (int iz = 0; iz < g.tz; ++iz) tile(iz, pre[0]);
    }
    if (stats) {
        float* sp = stats + ((int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + wave) * 3 * g.Cout;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) cnt += __shfl_xor(cnt, o, 64);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 4; ++e) stats_row_write(sp, g.Cout, co0 + nb * 16 + lg * 4 + e, st1[nb][e], st2[nb][e], shf[nb][e], cnt, li == 0);
    }
}

// ---- multi-chunk inputs (Cin = 64, 128, 256: the coarser levels), 3x3x3, 32 output channels per workgroup: the same column walk with the
// channel chunks OUTSIDE the z tiles.  The accumulators of all TZT z tiles of the column stay in registers (Z = 16 TZT <= 64 at these levels),
// so a chunk's weight slab is staged once per column instead of once per tile, and the halos of the (chunk, z tile) sequence are prefetched
// through registers like in the single-chunk kernel; for TZT <= 2 the next chunk's weight slab travels through registers as well.
template <int TZT>
__global__ __launch_bounds__(CT) void conv3_fwd_mc_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wp, bf16* __restrict__ y, ConvGeo g,
                                                          float* __restrict__ stats) {
    typedef CG<32, 3> G;
    constexpr int NB = 2, TX = 2, TY = 8, HX = 4, HY = 10, HZ = 18;
    constexpr int HALO_BYTES = HX * HY * HZ * G::VS;
    constexpr int CB = 32, RPW = 4;
    constexpr int NP = HX * HY * HZ * G::PPV, NPT = (NP + CT - 1) / CT;
    constexpr int NWP = G::NTS * CB * 4, NWT = (NWP + CT - 1) / CT;       // 16-byte pieces of a weight slab, per thread
    constexpr bool WPRE = TZT <= 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* wl = smem + HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int iy = t % g.ty;
    t /= g.ty;
    const int ix = t % g.tx, b = t / g.tx;
    const int x0 = ix * TX, y0 = iy * TY;
    const int co0 = blockIdx.y * CB;
    int hoff[NPT], hxy[NPT], hzz[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int p = tid + i * CT;
        const int hv = p / G::PPV, piece = p % G::PPV;
        const int hz = hv % HZ, hy = (hv / HZ) % HY, hx = hv / (HZ * HY);
        const int gx = x0 + hx - 1, gy = y0 + hy - 1;
        const bool ok = p < NP && (unsigned)gx < (unsigned)g.X && (unsigned)gy < (unsigned)g.Y;
        hoff[i] = hv * G::VS + piece * 16;
        hxy[i] = ok ? (gx * g.Y + gy) : -1;
        hzz[i] = ((hz - 1) & 0xffff) | (piece << 16);
    }
    const bf16* xb = x + (int64_t)b * g.X * g.Y * g.Z * g.Cin;
    const int nch = g.Cin / 32, nsteps = nch * TZT;
    u32x4 pre[NPT];
    auto fetch = [&](int step) {
        const int cc = step / TZT, z0 = (step % TZT) * 16;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int gz = z0 + (int)(short)(hzz[i] & 0xffff), piece = hzz[i] >> 16;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (hxy[i] >= 0 && (unsigned)gz < (unsigned)g.Z)
                v = *reinterpret_cast<const u32x4*>(xb + ((int64_t)hxy[i] * g.Z + gz) * g.Cin + cc * 32 + piece * 8);
            pre[i] = v;
        }
    };
    u32x4 wpre[WPRE ? NWT : 1];
    auto wfetch = [&](int cc) {
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int p = tid + i * CT;
            if (p < NWP) {
                const int piece = p & 3, row = (p >> 2) % CB, ts = (p >> 2) / CB;
                const u32x4 v = *reinterpret_cast<const u32x4*>(wp + ((int64_t)(cc * G::NTS + ts) * g.Cout + co0 + row) * 32 + piece * 8);
                if (WPRE)
                    wpre[i] = v;
                else
                    *reinterpret_cast<u32x4*>(wl + p * 16) = v;
            }
        }
    };
    f32x4 acc[TZT][RPW][NB];
#pragma unroll
    for (int z = 0; z < TZT; ++z)
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[z][r][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    fetch(0);
    if (WPRE) wfetch(0);
    for (int cc = 0; cc < nch; ++cc) {
        // the previous chunk's last tile ended with a barrier: the slab may be overwritten
        if (WPRE) {
#pragma unroll
            for (int i = 0; i < NWT; ++i)
                if (tid + i * CT < NWP) *reinterpret_cast<u32x4*>(wl + (tid + i * CT) * 16) = wpre[i];
        } else {
            wfetch(cc);
        }
#pragma unroll
        for (int z = 0; z < TZT; ++z) {
            const int step = cc * TZT + z;
#pragma unroll
            for (int i = 0; i < NPT; ++i)
                if (tid + i * CT < NP) *reinterpret_cast<u32x4*>(halo + hoff[i]) = pre[i];
            __syncthreads();
            if (step + 1 < nsteps) fetch(step + 1);
            if (WPRE && z == TZT - 1 && cc + 1 < nch) wfetch(cc + 1);
#pragma unroll
            for (int ts = 0; ts < G::NTS; ++ts) {
                frag_t a[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) a[nb] = *reinterpret_cast<const frag_t*>(wl + ((ts * CB + nb * 16 + li) * 64 + lg * 16));
                const int dx = ts / 9, dy = (ts / 3) % 3, dz = ts % 3;
                const int boff = ((dx * HY + dy) * HZ + dz + li) * G::VS + lg * 16;
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int row = wave * RPW + r, xl = row / TY, yl = row % TY;
                    const frag_t bf = *reinterpret_cast<const frag_t*>(halo + boff + (xl * HY + yl) * HZ * G::VS);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[z][r][nb] = mma(a[nb], bf, acc[z][r][nb]);
                }
            }
            __syncthreads();
        }
    }
    float st1[NB][4], st2[NB][4], shf[NB][4];
    float cnt = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            st1[nb][e] = st2[nb][e] = 0.f;
            shf[nb][e] = stats ? __shfl((float)(bf16)acc[0][0][nb][e], lane & 48, 64) : 0.f;      // the statistics' shift, see stats_row_write
        }
#pragma unroll
    for (int z = 0; z < TZT; ++z)
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = wave * RPW + r, gx = x0 + row / TY, gy = y0 + row % TY, gz = z * 16 + li;
            if (gx < g.X && gy < g.Y && gz < g.Z) {
                if (stats) cnt += 1.f;
                bf16* yp = y + ((((int64_t)b * g.X + gx) * g.Y + gy) * g.Z + gz) * g.ldy + co0 + lg * 4;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    bf16x4 o;
                    if (g.accumulate) {
                        o = *reinterpret_cast<const bf16x4*>(yp + nb * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[z][r][nb][e] + (float)o[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[z][r][nb][e];
                    }
                    *reinterpret_cast<bf16x4*>(yp + nb * 16) = o;
                    if (stats) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float q = (float)o[e] - shf[nb][e];
                            st1[nb][e] += q;
                            st2[nb][e] = fmaf(q, q, st2[nb][e]);
                        }
                    }
                }
            }
        }
    if (stats) {                                                    // see conv_fwd_strip_kernel
        float* sp = stats + ((int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + wave) * 3 * g.Cout;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) cnt += __shfl_xor(cnt, o, 64);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 4; ++e) stats_row_write(sp, g.Cout, co0 + nb * 16 + lg * 4 + e, st1[nb][e], st2[nb][e], shf[nb][e], cnt, li == 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------- weight gradient
constexpr int TZW = 32;       // z extent of a weight-gradient tile = one 32-deep contraction step per (x, y) row

// KS 3: wave w owns taps w, w + 4, ... (7 accumulator sets) and walks every (x, y) row of the tile; ONE partial per workgroup.
// KS 1: there is one tap, so the waves split the rows instead (row % 4 == wave) and each writes its own partial (4 per workgroup).
template <int CPC, int MB, int TX, int TY, int KS>
__global__ __launch_bounds__(CT) void conv_wgrad_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ part,
                                                        ConvGeo g, int tiles_per_wg) {
    typedef CG<CPC, KS> G;
    constexpr int PAD = G::PAD, HX = TX + 2 * PAD, HY = TY + 2 * PAD, HZ = TZW + 2 * PAD;
    constexpr int NBK = CPC >= 16 ? CPC / 16 : 1;
    constexpr int VSD = 32 * MB;                    // bytes per voxel of the dy image (16 MB channels)
    constexpr int DY_BYTES = TX * TY * TZW * VSD;
    constexpr int WTAPS = KS == 3 ? 7 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* dyt = smem;
    char* halo = smem + DY_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4, q = li >> 2, p4 = li & 3;
    const int cin_chunks = g.Cin / CPC;
    const int cib = blockIdx.y % cin_chunks, cob = blockIdx.y / cin_chunks;
    f32x4 acc[WTAPS][MB][NBK];
#pragma unroll
    for (int k = 0; k < WTAPS; ++k)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) acc[k][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int t_lo = blockIdx.x * tiles_per_wg, t_hi = min(g.tiles, t_lo + tiles_per_wg);
    // software pipeline over the workgroup's tiles: the dy tile and the x halo of tile t + 1 are fetched into registers while tile t is
    // multiplied out of LDS.  A thread's pieces have fixed tile-local coordinates: (xl, yl, zl, piece) packed 8 bits each, -1 = no piece.
    constexpr int DPV = VSD / 16;
    constexpr int NPD = (TX * TY * TZW * DPV + CT - 1) / CT, NPH = (HX * HY * HZ * G::PPV + CT - 1) / CT;
    int dyc[NPD], hc[NPH];
#pragma unroll
    for (int i = 0; i < NPD; ++i) {
        const int p = tid + i * CT, v = p / DPV, piece = p % DPV;
        dyc[i] = p < TX * TY * TZW * DPV ? ((v / (TZW * TY)) | (((v / TZW) % TY) << 8) | ((v % TZW) << 16) | (piece << 24)) : -1;
    }
#pragma unroll
    for (int i = 0; i < NPH; ++i) {
        const int p = tid + i * CT, hv = p / G::PPV, piece = p % G::PPV;
        hc[i] = p < HX * HY * HZ * G::PPV ? ((hv / (HZ * HY)) | (((hv / HZ) % HY) << 8) | ((hv % HZ) << 16) | (piece << 24)) : -1;
    }
    u32x4 pd[NPD], ph[NPH];
    auto fetch = [&](int t) {
        int b, ix, iy, iz;
        decode_tile(g, t, b, ix, iy, iz);
        const int x0 = ix * TX, y0 = iy * TY, z0 = iz * TZW;
#pragma unroll
        for (int i = 0; i < NPD; ++i) {
            const int c = dyc[i];
            pd[i] = c < 0 ? u32x4{0u, 0u, 0u, 0u}
                          : load_voxel(dy, g, b, x0 + (c & 255), y0 + ((c >> 8) & 255), z0 + ((c >> 16) & 255), g.Cout, cob * 16 * MB + (c >> 24) * 8);
        }
#pragma unroll
        for (int i = 0; i < NPH; ++i) {
            const int c = hc[i];
            ph[i] = c < 0 ? u32x4{0u, 0u, 0u, 0u}
                          : load_voxel(x, g, b, x0 + (c & 255) - PAD, y0 + ((c >> 8) & 255) - PAD, z0 + ((c >> 16) & 255) - PAD, g.Cin,
                                       cib * CPC + (c >> 24) * 8);
        }
    };
    if (t_lo < t_hi) fetch(t_lo);
    for (int t = t_lo; t < t_hi; ++t) {
#pragma unroll
        for (int i = 0; i < NPD; ++i)
            if (dyc[i] >= 0) *reinterpret_cast<u32x4*>(dyt + (tid + i * CT) * 16) = pd[i];        // piece p lives at byte 16 p of its image
#pragma unroll
        for (int i = 0; i < NPH; ++i)
            if (hc[i] >= 0) *reinterpret_cast<u32x4*>(halo + (tid + i * CT) * 16) = ph[i];
        __syncthreads();
        if (t + 1 < t_hi) fetch(t + 1);
        for (int row = (KS == 3 ? 0 : wave); row < TX * TY; row += (KS == 3 ? 1 : 4)) {
            const int xl = row / TY, yl = row % TY;
            frag_t a[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const char* ap = dyt + (row * TZW + 8 * lg + q) * VSD + (mb * 16 + 4 * p4) * 2;
                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, ap));
                const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, ap + 4 * VSD));
                const short8v rr = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                a[mb] = __builtin_bit_cast(frag_t, rr);
            }
#pragma unroll
            for (int k = 0; k < WTAPS; ++k) {
                const int tap = KS == 3 ? wave + 4 * k : 0;
                if (tap < G::NT) {
                    const int dx = KS == 3 ? tap / 9 : 0, dyy = KS == 3 ? (tap / 3) % 3 : 0, dz = KS == 3 ? tap % 3 : 0;
                    const char* bp = halo + ((((xl + dx) * HY + yl + dyy) * HZ) + dz + 8 * lg + q) * G::VS + 4 * p4 * 2;
#pragma unroll
                    for (int nb = 0; nb < NBK; ++nb) {
                        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, bp + nb * 32));
                        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, bp + nb * 32 + 4 * G::VS));
                        const short8v rr = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        const frag_t bf = __builtin_bit_cast(frag_t, rr);
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) acc[k][mb][nb] = mma(a[mb], bf, acc[k][mb][nb]);
                    }
                }
            }
        }
        __syncthreads();                                // every wave is done with these images before the next tile overwrites them
    }
    // partial [slot][blockIdx.y][tap][16 MB][16 NBK] with slot = the workgroup (KS 3) or (workgroup, wave) (KS 1); D[co = 4 lg + e][ci = li]
    constexpr int PB = 16 * MB * 16 * NBK;
    const int slot = KS == 3 ? blockIdx.x : blockIdx.x * 4 + wave;
    float* out = part + ((int64_t)slot * gridDim.y + blockIdx.y) * G::NT * PB;
#pragma unroll
    for (int k = 0; k < WTAPS; ++k) {
        const int tap = KS == 3 ? wave + 4 * k : 0;
        if (tap < G::NT) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) out[tap * PB + (mb * 16 + 4 * lg + e) * (16 * NBK) + nb * 16 + li] = acc[k][mb][nb][e];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- data movers
// depth-to-space of the transposed convolution's GEMM output: cols [B Xi Yi Zi][8 C] with the 8 = (dx, dy, dz) -> out [B][2Xi][2Yi][2Zi][C]
// (to_space = 1), or the inverse gather for the backward pass (to_space = 0).  16 bytes per thread step; C % 8 == 0.  The space tensor may be
// a channel slice of a wider channels-last buffer (row stride lds8 sixteen-byte units): the up-sampled map is written straight into the
// first half of the concatenation the next residual block reads, and its gradient is gathered straight out of that block's input gradient.
__global__ __launch_bounds__(CT) void d2s_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int B, int Xi, int Yi, int Zi, int C,
                                                 int to_space, int64_t lds8, const bf16* __restrict__ skip, int Cs, int cvt_shift) {
    // a thread owns one 16-byte vector of a space-side voxel ROW: C / 8 vectors of the up-sampled map, then (to_space with a skip) Cs / 8 of the
    // skip connection, so a whole row of the concatenation is written by consecutive lanes (full cache lines).  The row length cvt is a power of
    // two (cvt_shift): voxel and vector index are a shift and a mask of the thread index, the voxel coordinates three 32-bit divisions.
    const int cv = C / 8, cvt = 1 << cvt_shift;
    const unsigned nvox = (unsigned)B * (2u * Xi) * (2u * Yi) * (2u * Zi);
    const unsigned vpb = CT >> cvt_shift;                                  // voxels per workgroup step
    const int c = threadIdx.x & (cvt - 1);
    for (unsigned vox = blockIdx.x * vpb + (threadIdx.x >> cvt_shift); vox < nvox; vox += gridDim.x * vpb) {
        const int64_t k = (int64_t)vox * lds8 + c;
        if (c >= cv) {                                          // skip half of the concatenation
            reinterpret_cast<u32x4*>(dst)[k] = reinterpret_cast<const u32x4*>(skip)[(int64_t)vox * (Cs / 8) + (c - cv)];
            continue;
        }
        unsigned r = vox;
        const unsigned zo = r % (2u * Zi);
        r /= 2u * Zi;
        const unsigned yo = r % (2u * Yi);
        r /= 2u * Yi;
        const unsigned xo = r % (2u * Xi), b = r / (2u * Xi);
        const int64_t vin = (((int64_t)b * Xi + (xo >> 1)) * Yi + (yo >> 1)) * Zi + (zo >> 1);
        const int64_t j = (vin * 8 + ((xo & 1) * 4 + (yo & 1) * 2 + (zo & 1))) * cv + c;
        if (to_space)
            reinterpret_cast<u32x4*>(dst)[k] = reinterpret_cast<const u32x4*>(src)[j];
        else
            reinterpret_cast<u32x4*>(dst)[j] = reinterpret_cast<const u32x4*>(src)[k];
    }
}

// fp32 [B][C][S] (C <= 8) -> bf16 [B][S][8], channels C..7 zero: the input volume as an 8-channel operand of the MFMA kernels
__global__ __launch_bounds__(CT) void pad8_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int64_t B, int C, int64_t S) {
    const int64_t V = B * S;
    for (int64_t i = (int64_t)blockIdx.x * CT + threadIdx.x; i < V; i += (int64_t)gridDim.x * CT) {
        const int64_t b = i / S, v = i % S;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(e < C ? src[(b * C + e) * S + v] : 0.f);
        reinterpret_cast<bf16x8*>(dst)[i] = o;
    }
}

// rows of C <= 8 values (fp32 or bf16, row stride ld elements) -> dense bf16 rows of 8, columns C..7 zero: a narrow channels-last gradient (the
// logits' gradient of the output head) as an operand of the MFMA kernels, in one pass instead of a fill and a strided cast copy
template <typename S> __global__ __launch_bounds__(CT) void padrow8_kernel(const S* __restrict__ src, bf16* __restrict__ dst, int64_t V, int C, int64_t ld) {
    for (int64_t i = (int64_t)blockIdx.x * CT + threadIdx.x; i < V; i += (int64_t)gridDim.x * CT) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(e < C ? (float)src[i * ld + e] : 0.f);
        reinterpret_cast<bf16x8*>(dst)[i] = o;
    }
}

int conv_check(const char* name, const void* x, const void* w, const void* y, int64_t B, int64_t X, int64_t Y, int64_t Z, int64_t Cin,
               int64_t Cout, int ksize) {
    UCF_CHECK_ARG(x && w && y, "%s: null pointer", name);
    UCF_CHECK_ARG(B > 0 && X > 0 && Y > 0 && Z > 0, "%s: empty volume", name);
    UCF_CHECK_ARG(B * X * Y * Z < (1ll << 31), "%s: more than 2^31 voxels", name);
    UCF_CHECK_ARG(ksize == 1 || ksize == 3, "%s: kernel size must be 1 or 3 (got %d)", name, ksize);
    UCF_CHECK_ARG(Cin == 8 || Cin == 16 || (Cin > 0 && Cin % 32 == 0), "%s: Cin must be 8, 16 or a multiple of 32 (got %lld)", name, (long long)Cin);
    UCF_CHECK_ARG(Cout > 0 && Cout % 16 == 0, "%s: Cout must be a multiple of 16 (got %lld)", name, (long long)Cout);
    UCF_CHECK_ARG(ucf_is_aligned16(x) && ucf_is_aligned16(w) && ucf_is_aligned16(y), "%s: operands must be 16-byte aligned", name);
    return UCFVIT_OK;
}

template <int CPC, int NB, int TX, int TY, int KS, typename OutT>
int launch_fwd(const bf16* x, const bf16* wp, const float* bias, OutT* y, ConvGeo g, hipStream_t s) {
    typedef CG<CPC, KS> G;
    g.tx = (g.X + TX - 1) / TX;
    g.ty = (g.Y + TY - 1) / TY;
    g.tz = (g.Z + 15) / 16;
    const int64_t tiles = (int64_t)g.B * g.tx * g.ty * g.tz;
    UCF_CHECK_ARG(tiles < (1ll << 31) && g.Cout / (16 * NB) < 65536, "ucfvit_conv3d_fwd: grid too large");
    g.tiles = (int)tiles;
    constexpr int SMEM = (TX + 2 * G::PAD) * (TY + 2 * G::PAD) * (16 + 2 * G::PAD) * G::VS + G::NTS * 16 * NB * 64;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_fwd_kernel<CPC, NB, TX, TY, KS, OutT>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_fwd_kernel<CPC, NB, TX, TY, KS, OutT>), dim3(g.tiles, g.Cout / (16 * NB)), dim3(CT), SMEM, s, x, wp, bias, y, g);
    UCF_LAUNCH_CHECK("ucfvit_conv3d_fwd");
    return UCFVIT_OK;
}

static int strip_mode();

template <int CPC, int NB, int TX, int TY, int KS, typename OutT, bool FAST, int DEPTH>
int launch_fwd_strip_v(const bf16* x, const bf16* wp, const float* bias, OutT* y, const ConvGeo& g, float* stats, int64_t cols, hipStream_t s) {
    typedef CG<CPC, KS> G;
    constexpr int WSLOTS = (KS == 3 && CPC == 16) ? 15 : G::NTS;            // the dy-free step arrangement of conv_fwd_strip_kernel (SHARE)
    constexpr int SMEM = (TX + 2 * G::PAD) * (TY + 2 * G::PAD) * (16 + 2 * G::PAD) * G::VS + WSLOTS * 16 * NB * 64;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_fwd_strip_kernel<CPC, NB, TX, TY, KS, OutT, FAST, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_fwd_strip_kernel<CPC, NB, TX, TY, KS, OutT, FAST, DEPTH>), dim3((unsigned)cols, g.Cout / (16 * NB)), dim3(CT), SMEM, s, x, wp,
                       bias, y, g, stats);
    UCF_LAUNCH_CHECK("ucfvit_conv3d_fwd");
    return UCFVIT_OK;
}

template <int CPC, int NB, int TX, int TY, int KS, typename OutT>
int launch_fwd_strip(const bf16* x, const bf16* wp, const float* bias, OutT* y, ConvGeo g, float* stats, hipStream_t s) {
    g.tx = (g.X + TX - 1) / TX;
    g.ty = (g.Y + TY - 1) / TY;
    g.tz = (g.Z + 15) / 16;
    const int64_t cols = (int64_t)g.B * g.tx * g.ty;
    UCF_CHECK_ARG(cols < (1ll << 31) && g.Cout / (16 * NB) < 65536, "ucfvit_conv3d_fwd: grid too large");
    g.tiles = (int)cols;
    // the branch-free form (see conv_fwd_strip_kernel): 4-channel aligned output rows, a batch element's input and output below 4 GB
    const int64_t vox = (int64_t)g.X * g.Y * g.Z, lim = (1ll << 32) - 64;
    const bool fast = (g.ldy & 3) == 0 && (g.cout_store & 3) == 0 && vox * g.Cin * 2 < lim && vox * g.ldy * (int64_t)sizeof(OutT) < lim && strip_mode() != 3;
    if (fast) {
        if constexpr (CPC <= 16) return launch_fwd_strip_v<CPC, NB, TX, TY, KS, OutT, true, 2>(x, wp, bias, y, g, stats, cols, s);
        else return launch_fwd_strip_v<CPC, NB, TX, TY, KS, OutT, true, 1>(x, wp, bias, y, g, stats, cols, s);
    }
    return launch_fwd_strip_v<CPC, NB, TX, TY, KS, OutT, false, 1>(x, wp, bias, y, g, stats, cols, s);
}

template <int TZT>
int launch_fwd_mc(const bf16* x, const bf16* wp, bf16* y, ConvGeo g, float* stats, hipStream_t s) {
    g.tx = (g.X + 1) / 2;
    g.ty = (g.Y + 7) / 8;
    g.tz = TZT;
    const int64_t cols = (int64_t)g.B * g.tx * g.ty;
    UCF_CHECK_ARG(cols < (1ll << 31) && g.Cout / 32 < 65536, "ucfvit_conv3d_fwd: grid too large");
    g.tiles = (int)cols;
    constexpr int SMEM = 4 * 10 * 18 * 64 + 27 * 32 * 64;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3_fwd_mc_kernel<TZT>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv3_fwd_mc_kernel<TZT>), dim3((unsigned)cols, g.Cout / 32), dim3(CT), SMEM, s, x, wp, y, g, stats);
    UCF_LAUNCH_CHECK("ucfvit_conv3d_fwd");
    return UCFVIT_OK;
}

// UCFVIT_CONV_STRIP — a TEST hook (tests/test_conv3d.py runs every shape through both kernel families and compares them bit for bit):
// 0 never, 1 (default) when the (x, y) columns fill the chip, 2 whenever the column kernel applies.  Read once (thread-safe static).
static int strip_mode() {
    static const int flag = [] {
        const char* e = getenv("UCFVIT_CONV_STRIP");
        return (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 1;      // 3 (test hook): column kernels with the branching memory operations
    }();
    return flag;
}

// which forward kernel serves a launch: 0 = one tile per workgroup, 1 = single-chunk column kernel, 2 = multi-chunk column kernel; TX, TY = its tile
struct FwdPlan {
    int kind, TX, TY;
};
static FwdPlan fwd_plan(const ConvGeo& g, int ksize, bool has_bias, bool out_bf16) {
    const int cpc = g.Cin < 32 ? g.Cin : 32, nb16 = g.Cout / 16;
    if (cpc == 32 && ksize == 3 && out_bf16 && g.Cin > 32 && !has_bias && g.Cout % 32 == 0 && g.cout_store == g.Cout && g.ldy == g.Cout && strip_mode() &&
        (g.Z == 16 || g.Z == 32 || g.Z == 64)) {
        const int64_t wgs = (int64_t)g.B * ((g.X + 1) / 2) * ((g.Y + 7) / 8) * (g.Cout / 32);
        if (strip_mode() == 2 || wgs >= 512) return FwdPlan{2, 2, 8};
    }
    if (g.Cin == cpc && g.Z > 16 && strip_mode()) {
        const int64_t cols = (int64_t)g.B * ((g.X + 1) / 2) * ((g.Y + 7) / 8);
        if (strip_mode() == 2 || cols * (nb16 % 4 == 0 ? nb16 / 4 : nb16 % 2 == 0 ? nb16 / 2 : nb16) >= 512) {
            if (nb16 % 4 == 0) return FwdPlan{1, 2, 4};
            if (nb16 % 2 == 0 || cpc == 32) return FwdPlan{1, 2, 8};      // (CPC 32, NB 1): the 4 x 8 tile's prefetch would not fit in registers
            return FwdPlan{1, 4, 8};
        }
    }
    return FwdPlan{0, 0, 0};
}

template <int CPC, int KS, typename OutT>
int dispatch_fwd(const bf16* x, const bf16* wp, const float* bias, OutT* y, const ConvGeo& g, float* stats, hipStream_t s) {
    const int nb16 = g.Cout / 16;
    const FwdPlan pl = fwd_plan(g, KS, bias != nullptr, sizeof(OutT) == 2);
    if constexpr (CPC == 32 && KS == 3 && sizeof(OutT) == 2) {
        if (pl.kind == 2) {
            if (g.Z == 16) return launch_fwd_mc<1>(x, wp, (bf16*)y, g, stats, s);
            if (g.Z == 32) return launch_fwd_mc<2>(x, wp, (bf16*)y, g, stats, s);
            return launch_fwd_mc<4>(x, wp, (bf16*)y, g, stats, s);
        }
    }
    if (pl.kind == 1) {
        if (nb16 % 4 == 0) return launch_fwd_strip<CPC, 4, 2, 4, KS, OutT>(x, wp, bias, y, g, stats, s);
        if (nb16 % 2 == 0) return launch_fwd_strip<CPC, 2, 2, 8, KS, OutT>(x, wp, bias, y, g, stats, s);
        if constexpr (CPC == 32)
            return launch_fwd_strip<CPC, 1, 2, 8, KS, OutT>(x, wp, bias, y, g, stats, s);
        else
            return launch_fwd_strip<CPC, 1, 4, 8, KS, OutT>(x, wp, bias, y, g, stats, s);
    }
    UCF_CHECK_ARG(!stats, "ucfvit_conv3d_fwd: this launch has no statistics epilogue (ask ucfvit_conv3d_fwd_stats_rows first)");
    if (nb16 % 4 == 0) return launch_fwd<CPC, 4, 2, 4, KS, OutT>(x, wp, bias, y, g, s);
    if (nb16 % 2 == 0) return launch_fwd<CPC, 2, 2, 8, KS, OutT>(x, wp, bias, y, g, s);
    return launch_fwd<CPC, 1, 2, 8, KS, OutT>(x, wp, bias, y, g, s);
}

template <int CPC, int MB, int TX, int TY, int KS> struct WG {
    static constexpr int NBK = CPC >= 16 ? CPC / 16 : 1;
    static constexpr int PB = 16 * MB * 16 * NBK;
    static constexpr int PAD = KS / 2;
    static constexpr int SMEM = TX * TY * TZW * 32 * MB + (TX + 2 * PAD) * (TY + 2 * PAD) * (TZW + 2 * PAD) * CPC * 2 + 64;
    static constexpr int SLOTS = KS == 3 ? 1 : 4;          // partials per workgroup
};

constexpr int64_t WGRAD_PART_FLOATS = 32ll << 20;      // cap of the partial-sum scratch (128 MiB)

// geometry shared by the workspace query and the launch
template <int CPC, int MB, int TX, int TY, int KS>
void wgrad_plan(ConvGeo& g, int& n_wg, int& tiles_per_wg, int& gy, int64_t& n_out) {
    typedef WG<CPC, MB, TX, TY, KS> W;
    g.tx = (g.X + TX - 1) / TX;
    g.ty = (g.Y + TY - 1) / TY;
    g.tz = (g.Z + TZW - 1) / TZW;
    g.tiles = g.B * g.tx * g.ty * g.tz;
    gy = (g.Cin / CPC) * (g.Cout / (16 * MB));
    n_out = (int64_t)gy * CG<CPC, KS>::NT * W::PB;
    int64_t cap = WGRAD_PART_FLOATS / (n_out * W::SLOTS);
    if (cap < 1) cap = 1;
    if (cap > 1024) cap = 1024;
    n_wg = (int)(g.tiles < cap ? g.tiles : cap);
    tiles_per_wg = (g.tiles + n_wg - 1) / n_wg;
    n_wg = (g.tiles + tiles_per_wg - 1) / tiles_per_wg;
}

template <int CPC, int MB, int TX, int TY, int KS>
int launch_wgrad(const bf16* x, const bf16* dy, float* dw, float* ws, ConvGeo g, hipStream_t s) {
    typedef WG<CPC, MB, TX, TY, KS> W;
    int n_wg, tpw, gy;
    int64_t n_out;
    wgrad_plan<CPC, MB, TX, TY, KS>(g, n_wg, tpw, gy, n_out);
    constexpr int SMEM = W::SMEM;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<CPC, MB, TX, TY, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_wgrad_kernel<CPC, MB, TX, TY, KS>), dim3(n_wg, gy), dim3(CT), SMEM, s, x, dy, ws, g, tpw);
    UCF_LAUNCH_CHECK("ucfvit_conv3d_wgrad");
    return ucfvit_reduce_rows(ws, dw, (int64_t)n_wg * W::SLOTS, n_out, 0, s);
}

// which instantiation serves (Cin, Cout):  CPC = min(Cin, 32);  forward NB = largest of 4, 2, 1 dividing Cout / 16;  weight gradient MB = 2
// when Cout % 32 == 0
#define CONV_SWITCH(CIN, KSIZE, ...)                                         \
    do {                                                                     \
        if ((KSIZE) == 3) {                                                  \
            constexpr int KS_ = 3;                                           \
            if ((CIN) == 8) { constexpr int CPC_ = 8; __VA_ARGS__ }          \
            else if ((CIN) == 16) { constexpr int CPC_ = 16; __VA_ARGS__ }   \
            else { constexpr int CPC_ = 32; __VA_ARGS__ }                    \
        } else {                                                             \
            constexpr int KS_ = 1;                                           \
            if ((CIN) == 8) { constexpr int CPC_ = 8; __VA_ARGS__ }          \
            else if ((CIN) == 16) { constexpr int CPC_ = 16; __VA_ARGS__ }   \
            else { constexpr int CPC_ = 32; __VA_ARGS__ }                    \
        }                                                                    \
    } while (0)

}  // namespace

// x [B][X][Y][Z][Cin] bf16, w_packed [Cin/CPC][NTS][Cout][32] bf16, bias fp32 [Cout] or NULL -> y[voxel * ldy + co] for co < cout_store
extern "C" int ucfvit_conv3d_fwd(const void* x, const void* w_packed, const float* bias, void* y, int64_t B, int64_t X, int64_t Y, int64_t Z,
                                 int64_t Cin, int64_t Cout, int ksize, int64_t ldy, int64_t cout_store, int out_dtype, int accumulate,
                                 float* stats_partial, void* stream) {
    if (int rc = conv_check("ucfvit_conv3d_fwd", x, w_packed, y, B, X, Y, Z, Cin, Cout, ksize)) return rc;
    UCF_CHECK_ARG(cout_store > 0 && cout_store <= Cout && ldy >= cout_store && ldy < (1ll << 31), "ucfvit_conv3d_fwd: need 0 < cout_store <= Cout, ldy >= cout_store");
    UCF_CHECK_ARG(out_dtype == UCFVIT_BF16 || out_dtype == UCFVIT_F32, "ucfvit_conv3d_fwd: bad out_dtype %d", out_dtype);
    ConvGeo g{(int)B, (int)X, (int)Y, (int)Z, (int)Cin, (int)Cout, 0, 0, 0, 0, (int)ldy, (int)cout_store, accumulate ? 1 : 0};
    hipStream_t s = (hipStream_t)stream;
    UCF_CHECK_ARG(!stats_partial || (out_dtype == UCFVIT_BF16 && !accumulate && cout_store == Cout && ldy == Cout),
                  "ucfvit_conv3d_fwd: statistics need a dense bf16 output without accumulation");
    CONV_SWITCH(Cin, ksize, {
        if (out_dtype == UCFVIT_BF16) return dispatch_fwd<CPC_, KS_, bf16>((const bf16*)x, (const bf16*)w_packed, bias, (bf16*)y, g, stats_partial, s);
        return dispatch_fwd<CPC_, KS_, float>((const bf16*)x, (const bf16*)w_packed, bias, (float*)y, g, stats_partial, s);
    });
    return UCFVIT_OK;
}

// rows per batch element of the statistics partials [B][rows][2][Cout] that ucfvit_conv3d_fwd writes for this launch (dense bf16 output, no
// accumulation), 0 when the kernel that serves it has no statistics epilogue (the caller then runs ucfvit_instnorm_cl_stats on the output)
extern "C" int64_t ucfvit_conv3d_fwd_stats_rows(int64_t B, int64_t X, int64_t Y, int64_t Z, int64_t Cin, int64_t Cout, int ksize, int has_bias) {
    if (!(Cin == 8 || Cin == 16 || (Cin > 0 && Cin % 32 == 0)) || Cout <= 0 || Cout % 16 || !(ksize == 1 || ksize == 3)) return 0;
    ConvGeo g{(int)B, (int)X, (int)Y, (int)Z, (int)Cin, (int)Cout, 0, 0, 0, 0, (int)Cout, (int)Cout, 0};
    const FwdPlan pl = fwd_plan(g, ksize, has_bias != 0, true);
    if (pl.kind == 0) return 0;
    return (int64_t)((X + pl.TX - 1) / pl.TX) * ((Y + pl.TY - 1) / pl.TY) * 4;
}

// number of fp32 values of the packed weight gradient [Cout/(16 MB)][Cin/CPC][taps][16 MB][16 NBK] and bytes of scratch for the partials
extern "C" int64_t ucfvit_conv3d_wgrad_size(int64_t Cin, int64_t Cout, int ksize) {
    const int64_t cpc = Cin < 32 ? Cin : 32;
    const int64_t nbk16 = cpc >= 16 ? cpc : 16;
    return (Cin / cpc) * (ksize == 3 ? 27 : 1) * Cout * nbk16;
}
extern "C" int64_t ucfvit_conv3d_wgrad_workspace(int64_t B, int64_t X, int64_t Y, int64_t Z, int64_t Cin, int64_t Cout, int ksize) {
    if (!(Cin == 8 || Cin == 16 || (Cin > 0 && Cin % 32 == 0)) || Cout <= 0 || Cout % 16 || !(ksize == 1 || ksize == 3)) return 0;
    ConvGeo g{(int)B, (int)X, (int)Y, (int)Z, (int)Cin, (int)Cout, 0, 0, 0, 0, 0, 0, 0};
    int n_wg = 0, tpw = 0, gy = 0;
    int64_t n_out = 0;
    CONV_SWITCH(Cin, ksize, {
        if (Cout % 32 == 0)
            wgrad_plan<CPC_, 2, 2, 4, KS_>(g, n_wg, tpw, gy, n_out);
        else
            wgrad_plan<CPC_, 1, 2, 4, KS_>(g, n_wg, tpw, gy, n_out);
    });
    return (int64_t)n_wg * (ksize == 3 ? 1 : 4) * n_out * (int64_t)sizeof(float);
}
// x [..][Cin], dy [..][Cout] bf16 -> dw_packed fp32 (layout above; UCF_VIT/_hip/conv.py:unpack_conv_wgrad turns it into [Cout][Cin][k][k][k])
extern "C" int ucfvit_conv3d_wgrad(const void* x, const void* dy, float* dw_packed, void* workspace, int64_t B, int64_t X, int64_t Y,
                                   int64_t Z, int64_t Cin, int64_t Cout, int ksize, void* stream) {
    if (int rc = conv_check("ucfvit_conv3d_wgrad", x, dy, dw_packed, B, X, Y, Z, Cin, Cout, ksize)) return rc;
    UCF_CHECK_ARG(workspace, "ucfvit_conv3d_wgrad: null workspace");
    ConvGeo g{(int)B, (int)X, (int)Y, (int)Z, (int)Cin, (int)Cout, 0, 0, 0, 0, 0, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    CONV_SWITCH(Cin, ksize, {
        if (Cout % 32 == 0) return launch_wgrad<CPC_, 2, 2, 4, KS_>((const bf16*)x, (const bf16*)dy, dw_packed, (float*)workspace, g, s);
        return launch_wgrad<CPC_, 1, 2, 4, KS_>((const bf16*)x, (const bf16*)dy, dw_packed, (float*)workspace, g, s);
    });
    return UCFVIT_OK;
}

// to_space = 1: cols [B Xi Yi Zi][8 C] -> space[voxel * ld_space + c] over [B][2Xi][2Yi][2Zi] voxels;  to_space = 0: the inverse.  bf16, C % 8 == 0,
// ld_space % 8 == 0 (ld_space = C: a dense tensor; larger: a channel slice of a wider channels-last buffer, pointer at the slice's first channel).
// skip (to_space = 1 only, may be NULL): dense [B][2Xi][2Yi][2Zi][Cs] map copied behind the C channels of every voxel row in the same pass
// (ld_space >= C + Cs): the concatenation (up-sampled, skip) of UnetrUpBlock written as whole rows.
extern "C" int ucfvit_depth_to_space2(const void* src, void* dst, int64_t B, int64_t Xi, int64_t Yi, int64_t Zi, int64_t C, int64_t ld_space,
                                      int to_space, const void* skip, int64_t Cs, void* stream) {
    UCF_CHECK_ARG(src && dst && B > 0 && Xi > 0 && Yi > 0 && Zi > 0 && C > 0 && C % 8 == 0, "ucfvit_depth_to_space2: bad arguments");
    UCF_CHECK_ARG(ld_space >= C && ld_space % 8 == 0, "ucfvit_depth_to_space2: ld_space must be a multiple of 8 and >= C");
    UCF_CHECK_ARG(ucf_is_aligned16(src) && ucf_is_aligned16(dst), "ucfvit_depth_to_space2: operands must be 16-byte aligned");
    if (skip) UCF_CHECK_ARG(to_space == 1 && Cs > 0 && Cs % 8 == 0 && ld_space >= C + Cs && ucf_is_aligned16(skip), "ucfvit_depth_to_space2: bad skip operand");
    const int64_t cvt = (C + (skip ? Cs : 0)) / 8;                           // 16-byte vectors per voxel row handled by this launch
    UCF_CHECK_ARG(cvt <= CT && (cvt & (cvt - 1)) == 0, "ucfvit_depth_to_space2: (C + Cs) / 8 must be a power of two <= 256 (got %lld)", (long long)cvt);
    UCF_CHECK_ARG(B * Xi * Yi * Zi * 8 < (1ll << 32), "ucfvit_depth_to_space2: more than 2^32 output voxels");
    int shift = 0;
    while ((1ll << shift) < cvt) ++shift;
    const int64_t total = B * Xi * Yi * Zi * 8 * cvt;                        // 16-byte pieces
    int64_t blocks = (total + CT - 1) / CT;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(d2s_kernel, dim3((unsigned)blocks), dim3(CT), 0, (hipStream_t)stream, (const bf16*)src, (bf16*)dst, (int)B, (int)Xi,
                       (int)Yi, (int)Zi, (int)C, to_space, ld_space / 8, (const bf16*)skip, (int)Cs, shift);
    UCF_LAUNCH_CHECK("ucfvit_depth_to_space2");
    return UCFVIT_OK;
}

extern "C" int ucfvit_pad_channels8(const float* src, void* dst, int64_t B, int64_t C, int64_t S, void* stream) {
    UCF_CHECK_ARG(src && dst && B > 0 && S > 0 && C >= 1 && C <= 8 && ucf_is_aligned16(dst), "ucfvit_pad_channels8: need 1 <= C <= 8");
    int64_t blocks = (B * S + CT - 1) / CT;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(pad8_kernel, dim3((unsigned)blocks), dim3(CT), 0, (hipStream_t)stream, src, (bf16*)dst, B, (int)C, S);
    UCF_LAUNCH_CHECK("ucfvit_pad_channels8");
    return UCFVIT_OK;
}

extern "C" int ucfvit_pad_rows8(const void* src, int src_dtype, void* dst, int64_t V, int64_t C, int64_t ld, void* stream) {
    UCF_CHECK_ARG(src && dst && V > 0 && C >= 1 && C <= 8 && ld >= C && ucf_is_aligned16(dst), "ucfvit_pad_rows8: need 1 <= C <= 8 and ld >= C");
    UCF_CHECK_ARG(src_dtype == UCFVIT_F32 || src_dtype == UCFVIT_BF16, "ucfvit_pad_rows8: source must be fp32 or bf16");
    int64_t blocks = (V + CT - 1) / CT;
    if (blocks > 65536) blocks = 65536;
    if (src_dtype == UCFVIT_F32)
        hipLaunchKernelGGL(padrow8_kernel<float>, dim3((unsigned)blocks), dim3(CT), 0, (hipStream_t)stream, (const float*)src, (bf16*)dst, V, (int)C, ld);
    else
        hipLaunchKernelGGL(padrow8_kernel<bf16>, dim3((unsigned)blocks), dim3(CT), 0, (hipStream_t)stream, (const bf16*)src, (bf16*)dst, V, (int)C, ld);
    UCF_LAUNCH_CHECK("ucfvit_pad_rows8");
    return UCFVIT_OK;
}

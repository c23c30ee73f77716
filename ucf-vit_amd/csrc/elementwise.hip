// HBM-bound front-end / loss / optimizer kernels for gfx950: patch im2col, token assembly (cls + pos-embed),
// softmax cross-entropy, MAE index math + row gathers, patchify-MSE, fused AdamW, dtype casts.
// All memory traffic is 16-byte vectorised where alignment allows; reductions are deterministic (no float atomics).
#include "common.h"

namespace {

// ===================================================================================================
// im2col for non-overlapping patches.  One workgroup stages a band of p image rows (contiguous in the
// innermost image dimension X) into LDS with coalesced row reads, then writes, for every patch along X,
// a run of p*p consecutive GEMM-row elements.
//   2-D: band = (b, c, h)       rows = ph, X = W ; cols run = [c*p*p, +p*p)           of GEMM row (b, h, w)
//   3-D: band = (b, c, h, w, ph) rows = pw, X = Z ; cols run = [(c*p + ph)*p*p, +p*p) of GEMM row (b, h, w, z)
// ===================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, T* __restrict__ cols, int C, int H, int W,
                                                     int Z, int p, int nd) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);  // [p][X]
    int64_t band = blockIdx.x;
    int X, gx;
    int64_t in_base, in_row_stride, out_row_base;
    int out_col;
    const int K = (nd == 2) ? C * p * p : C * p * p * p;
    if (nd == 2) {
        const int gh = H / p;
        X = W;
        gx = W / p;
        const int h = band % gh;
        band /= gh;
        const int c = band % C;
        const int64_t b = band / C;
        in_base = ((b * C + c) * H + (int64_t)h * p) * W;
        in_row_stride = W;
        out_row_base = (b * gh + h) * gx;
        out_col = c * p * p;
    } else {
        const int gh = H / p, gw = W / p;
        X = Z;
        gx = Z / p;
        const int ph = band % p;
        band /= p;
        const int w = band % gw;
        band /= gw;
        const int h = band % gh;
        band /= gh;
        const int c = band % C;
        const int64_t b = band / C;
        in_base = (((b * C + c) * H + (int64_t)h * p + ph) * W + (int64_t)w * p) * Z;
        in_row_stride = Z;
        out_row_base = ((b * gh + h) * gw + w) * gx;
        out_col = (c * p + ph) * p * p;
    }
    // coalesced read of p rows x X floats
    for (int i = threadIdx.x; i < p * X; i += blockDim.x) {
        const int r = i / X, xx = i - r * X;
        tile[i] = img[in_base + r * in_row_stride + xx];
    }
    __syncthreads();
    // write: for patch j along X, elements e = r*p + q  (r: band row, q: within-patch x)
    const int pp = p * p;
    for (int i = threadIdx.x; i < gx * pp; i += blockDim.x) {
        const int j = i / pp, e = i - j * pp;
        const int r = e / p, q = e - r * p;
        cols[(out_row_base + j) * K + out_col + e] = from_f32<T>(tile[r * X + j * p + q]);
    }
}

// ===================================================================================================
// token assembly: out[b][t] = (t < pre ? cls : patches[b][t-pre]) + pos[t]
// ===================================================================================================
template <typename T>
__global__ void tokens_fwd_kernel(const T* __restrict__ patches, const T* __restrict__ cls, const T* __restrict__ pos,
                                  T* __restrict__ out, int64_t B, int L, int D, int pre) {
    constexpr int EPV = Vec16<T>::N;
    const int nvec = D / EPV, N = L + pre;
    const int64_t total = B * N * nvec;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = i % nvec;
        const int64_t bt = i / nvec;
        const int t = bt % N;
        const int64_t b = bt / N;
        Vec16<T> a;
        if (t < pre)
            a = *reinterpret_cast<const Vec16<T>*>(cls + v * EPV);
        else
            a = *reinterpret_cast<const Vec16<T>*>(patches + ((b * L + (t - pre)) * D) + v * EPV);
        if (pos) {
            const Vec16<T> pv = *reinterpret_cast<const Vec16<T>*>(pos + (int64_t)t * D + v * EPV);
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < EPV; ++e) o.set(e, a.get(e) + pv.get(e));
            a = o;
        }
        *reinterpret_cast<Vec16<T>*>(out + bt * D + v * EPV) = a;
    }
}

// dpatches = dout[:, pre:], dpos[t] (+)= sum_b dout[b][t], dcls (+)= sum_b dout[b][0]
template <typename T>
__global__ void tokens_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dpatches, float* __restrict__ dpos,
                                  float* __restrict__ dcls, int64_t B, int L, int D, int pre, int accumulate) {
    constexpr int EPV = Vec16<T>::N;
    const int nvec = D / EPV, N = L + pre;
    const int64_t total = (int64_t)N * nvec;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int v = i % nvec, t = i / nvec;
    float acc[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
    for (int64_t b = 0; b < B; ++b) {
        const Vec16<T> d = *reinterpret_cast<const Vec16<T>*>(dout + ((b * N + t) * D) + v * EPV);
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[e] += d.get(e);
        if (t >= pre && dpatches) *reinterpret_cast<Vec16<T>*>(dpatches + ((b * L + (t - pre)) * D) + v * EPV) = d;
    }
    if (dpos) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            float* o = dpos + (int64_t)t * D + v * EPV + e;
            *o = accumulate ? *o + acc[e] : acc[e];
        }
    }
    if (dcls && pre && t == 0) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            float* o = dcls + v * EPV + e;
            *o = accumulate ? *o + acc[e] : acc[e];
        }
    }
}

// ===================================================================================================
// softmax cross-entropy: one wave per row; row losses to `row_loss`, then a single-workgroup ordered sum.
// ===================================================================================================
template <typename T>
__global__ __launch_bounds__(64) void ce_rows_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     float* __restrict__ row_loss, T* __restrict__ dlogits, int C, float gscale) {
    const int64_t row = blockIdx.x;
    const int lane = threadIdx.x;
    const T* lr = logits + row * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, to_f32<T>(lr[c]));
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(to_f32<T>(lr[c]) - mx);
    s = wave_sum(s);
    const int64_t lab = labels[row];
    const float lse = mx + logf(s);
    if (lane == 0) row_loss[row] = lse - to_f32<T>(lr[lab]);
    if (dlogits) {
        T* dr = dlogits + row * C;
        const float inv = 1.f / s;
        for (int c = lane; c < C; c += 64) {
            const float p = expf(to_f32<T>(lr[c]) - mx) * inv;
            dr[c] = from_f32<T>(gscale * (p - (c == lab ? 1.f : 0.f)));
        }
    }
}

// out = scale * sum(v[0..n)) with a fixed summation order (one workgroup)
__global__ __launch_bounds__(256) void ordered_sum_kernel(const float* __restrict__ v, float* __restrict__ out, int64_t n, float scale) {
    __shared__ float red[256];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0] * scale;
}

// ===================================================================================================
// MAE random masking index math: stable rank by counting (bit-exact with argsort for distinct keys).
// ===================================================================================================
__global__ __launch_bounds__(256) void mae_mask_kernel(const float* __restrict__ noise, int64_t* __restrict__ ids_shuffle,
                                                       int64_t* __restrict__ ids_restore, float* __restrict__ mask, int L,
                                                       int len_keep) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* nz = reinterpret_cast<float*>(smem_raw);
    const int64_t b = blockIdx.x;
    for (int i = threadIdx.x; i < L; i += blockDim.x) nz[i] = noise[b * L + i];
    __syncthreads();
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const float x = nz[i];
        int rank = 0;
        for (int j = 0; j < L; ++j) {
            const float y = nz[j];
            rank += (y < x) || (y == x && j < i);
        }
        ids_restore[b * L + i] = rank;
        ids_shuffle[b * L + rank] = i;
        mask[b * L + i] = rank >= len_keep ? 1.f : 0.f;
    }
}

// out[b][r][:] = src[b][idx[b*idx_stride + r]][:]   (16-B vector copy: bit-exact)
template <typename V>
__global__ void gather_rows_kernel(const V* __restrict__ src, const int64_t* __restrict__ idx, V* __restrict__ out, int64_t B,
                                   int L, int R, int nvec, int64_t idx_stride) {
    const int64_t total = B * R * nvec;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = i % nvec;
        const int64_t br = i / nvec;
        const int r = br % R;
        const int64_t b = br / R;
        const int64_t j = idx[b * idx_stride + r];
        out[br * nvec + v] = src[(b * L + j) * nvec + v];
    }
}
template <typename V>
__global__ void scatter_rows_kernel(const V* __restrict__ dout, const int64_t* __restrict__ idx, V* __restrict__ dsrc, int64_t B,
                                    int L, int R, int nvec, int64_t idx_stride) {
    const int64_t total = B * R * nvec;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = i % nvec;
        const int64_t br = i / nvec;
        const int r = br % R;
        const int64_t b = br / R;
        const int64_t j = idx[b * idx_stride + r];
        dsrc[(b * L + j) * nvec + v] = dout[br * nvec + v];
    }
}

// out[b][i] = (j = ids_restore[b][i]) < R ? x[b][j] : mask_token ; + pos[i]
template <typename T>
__global__ void unshuffle_fwd_kernel(const T* __restrict__ x, const T* __restrict__ mask_token,
                                     const int64_t* __restrict__ ids_restore, const T* __restrict__ pos, T* __restrict__ out,
                                     int64_t B, int L, int R, int D) {
    constexpr int EPV = Vec16<T>::N;
    const int nvec = D / EPV;
    const int64_t total = B * L * nvec;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = i % nvec;
        const int64_t bi = i / nvec;
        const int t = bi % L;
        const int64_t b = bi / L;
        const int64_t j = ids_restore[bi];
        Vec16<T> a = (j < R) ? *reinterpret_cast<const Vec16<T>*>(x + (b * R + j) * D + v * EPV)
                             : *reinterpret_cast<const Vec16<T>*>(mask_token + v * EPV);
        if (pos) {
            const Vec16<T> pv = *reinterpret_cast<const Vec16<T>*>(pos + (int64_t)t * D + v * EPV);
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < EPV; ++e) o.set(e, a.get(e) + pv.get(e));
            a = o;
        }
        *reinterpret_cast<Vec16<T>*>(out + bi * D + v * EPV) = a;
    }
}

// dx[b][j] = dout[b][i] where ids_restore[b][i] = j < R ; partial_mask[b][:] = sum over masked i of dout[b][i]
// one workgroup per b; thread <-> 16-B column vector; sequential over the L positions (deterministic)
template <typename T>
__global__ void unshuffle_bwd_kernel(const T* __restrict__ dout, const int64_t* __restrict__ ids_restore, T* __restrict__ dx,
                                     float* __restrict__ partial_mask, int L, int R, int D) {
    constexpr int EPV = Vec16<T>::N;
    const int nvec = D / EPV;
    const int64_t b = blockIdx.x;
    for (int v = threadIdx.x; v < nvec; v += blockDim.x) {
        float acc[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
        for (int t = 0; t < L; ++t) {
            const int64_t j = ids_restore[b * L + t];
            const Vec16<T> d = *reinterpret_cast<const Vec16<T>*>(dout + (b * L + t) * D + v * EPV);
            if (j < R) {
                *reinterpret_cast<Vec16<T>*>(dx + (b * R + j) * D + v * EPV) = d;
            } else {
#pragma unroll
                for (int e = 0; e < EPV; ++e) acc[e] += d.get(e);
            }
        }
#pragma unroll
        for (int e = 0; e < EPV; ++e) partial_mask[b * D + v * EPV + e] = acc[e];
    }
}

// dpos[t][:] (+)= sum_b dout[b][t][:]
template <typename T>
__global__ void batch_sum_kernel(const T* __restrict__ dout, float* __restrict__ dpos, int64_t B, int L, int D, int accumulate) {
    constexpr int EPV = Vec16<T>::N;
    const int nvec = D / EPV;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)L * nvec) return;
    float acc[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
    for (int64_t b = 0; b < B; ++b) {
        const Vec16<T> d = *reinterpret_cast<const Vec16<T>*>(dout + b * L * D + i * EPV);
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[e] += d.get(e);
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        float* o = dpos + i * EPV + e;
        *o = accumulate ? *o + acc[e] : acc[e];
    }
}

// ===================================================================================================
// patchify-MSE.  pred[b][l][e], e = ((ph*p + pw)[*p + pd])*C + c ; target = img[b][c][h*p+ph][w*p+pw][..]
// phase 1: per-workgroup partial sums of weighted squared error (and of the mask) ; phase 2: finalize ; phase 3: dpred
// ===================================================================================================
struct PatchGeom {
    int C, H, W, Z, p, nd, gh, gw, gz, L, P;
};
__device__ __forceinline__ float patch_target(const float* __restrict__ img, const PatchGeom& g, int64_t b, int l, int e) {
    const int c = e % g.C;
    int s = e / g.C;
    if (g.nd == 1) {  // pre-cut token sequence x[B][C][S][P]: target row l = 'b c s p -> b s (p c)' (train_masked_simple.py:29)
        return img[((b * g.C + c) * (int64_t)g.L + l) * g.p + s];
    } else if (g.nd == 2) {
        const int pw = s % g.p, ph = s / g.p;
        const int w = l % g.gw, h = l / g.gw;
        return img[((b * g.C + c) * g.H + (h * g.p + ph)) * (int64_t)g.W + (w * g.p + pw)];
    } else {
        const int pd = s % g.p;
        s /= g.p;
        const int pw = s % g.p, ph = s / g.p;
        const int z = l % g.gz;
        int t = l / g.gz;
        const int w = t % g.gw, h = t / g.gw;
        return img[(((b * g.C + c) * g.H + (h * g.p + ph)) * (int64_t)g.W + (w * g.p + pw)) * g.Z + (z * g.p + pd)];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void patch_mse_partial_kernel(const T* __restrict__ pred, const float* __restrict__ img,
                                                                const float* __restrict__ mask, float* __restrict__ ws,
                                                                PatchGeom g, int64_t B) {
    __shared__ float red[256];
    const int64_t total = B * g.L * g.P;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int e = i % g.P;
        const int64_t bl = i / g.P;
        const int l = bl % g.L;
        const int64_t b = bl / g.L;
        const float d = to_f32<T>(pred[i]) - patch_target(img, g, b, l, e);
        s += mask ? d * d * mask[bl] : d * d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) ws[blockIdx.x] = red[0];
    if (mask) {  // mask partial sums
        float m = 0.f;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B * g.L; i += (int64_t)gridDim.x * 256) m += mask[i];
        __syncthreads();
        red[threadIdx.x] = m;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) ws[1024 + blockIdx.x] = red[0];
    }
}

// ws[0..nb) err partials, ws[1024..1024+nb) mask partials -> loss ; ws[2048] = denominator used by the gradient
__global__ __launch_bounds__(64) void patch_mse_final_kernel(float* __restrict__ ws, float* __restrict__ loss, int nb, int masked,
                                                             float n_total, float P) {
    if (threadIdx.x != 0) return;
    float s = 0.f, m = 0.f;
    for (int i = 0; i < nb; ++i) s += ws[i];
    if (masked) {
        for (int i = 0; i < nb; ++i) m += ws[1024 + i];
        *loss = s / P / m;   // mean over the patch dim, then sum(loss*mask)/sum(mask)
        ws[2048] = P * m;
    } else {
        *loss = s / n_total;
        ws[2048] = n_total;
    }
}

template <typename T>
__global__ void patch_mse_grad_kernel(const T* __restrict__ pred, const float* __restrict__ img, const float* __restrict__ mask,
                                      const float* __restrict__ ws, T* __restrict__ dpred, PatchGeom g, int64_t B, float gscale) {
    const int64_t total = B * g.L * g.P;
    const float k = 2.f * gscale / ws[2048];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = i % g.P;
        const int64_t bl = i / g.P;
        const int l = bl % g.L;
        const int64_t b = bl / g.L;
        const float d = to_f32<T>(pred[i]) - patch_target(img, g, b, l, e);
        dpred[i] = from_f32<T>(mask ? k * d * mask[bl] : k * d);
    }
}

// ===================================================================================================
// fused AdamW (torch.optim.AdamW update order) + optional bf16 shadow write
// ===================================================================================================
template <typename G>
__global__ void adamw_kernel(float* __restrict__ p, const G* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             bf16* __restrict__ shadow, int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1,
                             float bc2_sqrt, float gscale) {
    const int64_t nv = n >> 2;
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
        const Vec4<G> gv = reinterpret_cast<const Vec4<G>*>(g)[i];
        bf16x4 sh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gv.get(e) * gscale;
            float pe = pv[e] * (1.f - lr * wd);
            const float me = b1 * mv[e] + (1.f - b1) * gr;
            const float ve = b2 * vv[e] + (1.f - b2) * gr * gr;
            const float denom = sqrtf(ve) / bc2_sqrt + eps;
            pe -= step_size * (me / denom);
            pv[e] = pe;
            mv[e] = me;
            vv[e] = ve;
            sh[e] = (bf16)pe;
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
        if (shadow) reinterpret_cast<bf16x4*>(shadow)[i] = sh;
    }
    // tail (n % 4)
    const int64_t t = (nv << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        const float gr = to_f32<G>(g[t]) * gscale;
        float pe = p[t] * (1.f - lr * wd);
        const float me = b1 * m[t] + (1.f - b1) * gr;
        const float ve = b2 * v[t] + (1.f - b2) * gr * gr;
        pe -= step_size * (me / (sqrtf(ve) / bc2_sqrt + eps));
        p[t] = pe;
        m[t] = me;
        v[t] = ve;
        if (shadow) shadow[t] = (bf16)pe;
    }
}

template <typename S, typename Dd>
__global__ void cast_kernel(const S* __restrict__ src, Dd* __restrict__ dst, int64_t n, float scale) {
    const int64_t nv = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        const Vec4<S> s = reinterpret_cast<const Vec4<S>*>(src)[i];
        Vec4<Dd> d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d.set(e, s.get(e) * scale);
        reinterpret_cast<Vec4<Dd>*>(dst)[i] = d;
    }
    const int64_t t = (nv << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = from_f32<Dd>(to_f32<S>(src[t]) * scale);
}

inline unsigned ew_grid(int64_t work_items, int block = 256) {
    int64_t g = (work_items + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;  // ~8 workgroups per CU, grid-stride the rest
    if (g < 1) g = 1;
    return (unsigned)g;
}

#define DTYPE_OK(d) ((d) == UCFVIT_F32 || (d) == UCFVIT_BF16)

}  // namespace

// ---------------------------------------------------------------------------------------------------
extern "C" int ucfvit_im2col(const float* img, void* cols, int64_t B, int64_t C, const int64_t* dims, int nd, int64_t p, int dtype,
                             void* stream) {
    if (B == 0) return UCFVIT_OK;                      // empty batch (pointers may be NULL)
    UCF_CHECK_ARG(img && cols && dims, "ucfvit_im2col: null pointer");
    UCF_CHECK_ARG(nd == 2 || nd == 3, "ucfvit_im2col: nd must be 2 or 3");
    UCF_CHECK_ARG(DTYPE_OK(dtype), "ucfvit_im2col: bad dtype %d", dtype);
    UCF_CHECK_ARG(p > 0 && B >= 0 && C > 0, "ucfvit_im2col: bad sizes");
    for (int i = 0; i < nd; ++i)
        UCF_CHECK_ARG(dims[i] > 0 && dims[i] % p == 0, "ucfvit_im2col: image dim %d (%lld) not a multiple of patch %lld", i,
                      (long long)dims[i], (long long)p);
    if (B == 0) return UCFVIT_OK;
    const int H = (int)dims[0], W = (int)dims[1], Z = nd == 3 ? (int)dims[2] : 1;
    const int X = nd == 2 ? W : Z;
    const int64_t bands = nd == 2 ? B * C * (H / p) : B * C * (H / p) * (W / p) * p;
    const size_t smem = (size_t)p * X * sizeof(float);
    UCF_CHECK_ARG(smem <= 64 * 1024, "ucfvit_im2col: p*X*4 = %zu exceeds 64 KiB LDS band", smem);
    UCF_CHECK_ARG(bands < (1ll << 31), "ucfvit_im2col: too many bands");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UCFVIT_F32)
        hipLaunchKernelGGL(im2col_kernel<float>, dim3((unsigned)bands), dim3(256), smem, s, img, (float*)cols, (int)C, H, W, Z, (int)p, nd);
    else
        hipLaunchKernelGGL(im2col_kernel<bf16>, dim3((unsigned)bands), dim3(256), smem, s, img, (bf16*)cols, (int)C, H, W, Z, (int)p, nd);
    UCF_LAUNCH_CHECK("ucfvit_im2col");
    return UCFVIT_OK;
}

extern "C" int ucfvit_tokens_fwd(const void* patches, const void* cls, const void* pos, void* out, int64_t B, int64_t L, int64_t D,
                                 int has_cls, int dtype, void* stream) {
    if (B == 0) return UCFVIT_OK;                      // empty batch (pointers may be NULL)
    UCF_CHECK_ARG(patches && out, "ucfvit_tokens_fwd: null pointer");
    UCF_CHECK_ARG(!has_cls || cls, "ucfvit_tokens_fwd: has_cls without cls pointer");
    UCF_CHECK_ARG(DTYPE_OK(dtype), "ucfvit_tokens_fwd: bad dtype %d", dtype);
    const int epv = dtype == UCFVIT_F32 ? 4 : 8;
    UCF_CHECK_ARG(D % epv == 0, "ucfvit_tokens_fwd: D=%lld must be a multiple of %d", (long long)D, epv);
    UCF_CHECK_ARG(ucf_is_aligned16(patches) && ucf_is_aligned16(out) && ucf_is_aligned16(cls) && ucf_is_aligned16(pos),
                  "ucfvit_tokens_fwd: pointers must be 16-byte aligned");
    if (B == 0) return UCFVIT_OK;
    const int pre = has_cls ? 1 : 0;
    const int64_t work = B * (L + pre) * (D / epv);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UCFVIT_F32)
        hipLaunchKernelGGL(tokens_fwd_kernel<float>, dim3(ew_grid(work)), dim3(256), 0, s, (const float*)patches, (const float*)cls,
                           (const float*)pos, (float*)out, B, (int)L, (int)D, pre);
    else
        hipLaunchKernelGGL(tokens_fwd_kernel<bf16>, dim3(ew_grid(work)), dim3(256), 0, s, (const bf16*)patches, (const bf16*)cls,
                           (const bf16*)pos, (bf16*)out, B, (int)L, (int)D, pre);
    UCF_LAUNCH_CHECK("ucfvit_tokens_fwd");
    return UCFVIT_OK;
}

extern "C" int ucfvit_tokens_bwd(const void* dout, void* dpatches, float* dpos, float* dcls, int64_t B, int64_t L, int64_t D,
                                 int has_cls, int accumulate, int dtype, void* stream) {
    UCF_CHECK_ARG(dout, "ucfvit_tokens_bwd: null pointer");
    UCF_CHECK_ARG(DTYPE_OK(dtype), "ucfvit_tokens_bwd: bad dtype %d", dtype);
    const int epv = dtype == UCFVIT_F32 ? 4 : 8;
    UCF_CHECK_ARG(D % epv == 0, "ucfvit_tokens_bwd: D=%lld must be a multiple of %d", (long long)D, epv);
    UCF_CHECK_ARG(ucf_is_aligned16(dout) && ucf_is_aligned16(dpatches), "ucfvit_tokens_bwd: pointers must be 16-byte aligned");
    if (B == 0) return UCFVIT_OK;
    const int pre = has_cls ? 1 : 0;
    const int64_t work = (L + pre) * (D / epv);
    const unsigned grid = (unsigned)((work + 127) / 128);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UCFVIT_F32)
        hipLaunchKernelGGL(tokens_bwd_kernel<float>, dim3(grid), dim3(128), 0, s, (const float*)dout, (float*)dpatches, dpos, dcls, B,
                           (int)L, (int)D, pre, accumulate);
    else
        hipLaunchKernelGGL(tokens_bwd_kernel<bf16>, dim3(grid), dim3(128), 0, s, (const bf16*)dout, (bf16*)dpatches, dpos, dcls, B,
                           (int)L, (int)D, pre, accumulate);
    UCF_LAUNCH_CHECK("ucfvit_tokens_bwd");
    return UCFVIT_OK;
}

extern "C" int ucfvit_cross_entropy(const void* logits, const int64_t* labels, float* loss, float* row_loss, void* dlogits,
                                    int64_t B, int64_t C, float grad_scale, int dtype, void* stream) {
    UCF_CHECK_ARG(logits && labels && loss && row_loss, "ucfvit_cross_entropy: null pointer");
    UCF_CHECK_ARG(B > 0 && C > 0 && B < (1ll << 31), "ucfvit_cross_entropy: bad shape");
    UCF_CHECK_ARG(DTYPE_OK(dtype), "ucfvit_cross_entropy: bad dtype %d", dtype);
    hipStream_t s = (hipStream_t)stream;
    const float gs = grad_scale / (float)B;
    if (dtype == UCFVIT_F32)
        hipLaunchKernelGGL(ce_rows_kernel<float>, dim3((unsigned)B), dim3(64), 0, s, (const float*)logits, labels, row_loss,
                           (float*)dlogits, (int)C, gs);
    else
        hipLaunchKernelGGL(ce_rows_kernel<bf16>, dim3((unsigned)B), dim3(64), 0, s, (const bf16*)logits, labels, row_loss,
                           (bf16*)dlogits, (int)C, gs);
    UCF_LAUNCH_CHECK("ucfvit_cross_entropy");
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, s, (const float*)row_loss, loss, B, 1.f / (float)B);
    UCF_LAUNCH_CHECK("ucfvit_cross_entropy(sum)");
    return UCFVIT_OK;
}

extern "C" int ucfvit_mae_mask(const float* noise, int64_t* ids_shuffle, int64_t* ids_restore, float* mask, int64_t B, int64_t L,
                               int64_t len_keep, void* stream) {
    UCF_CHECK_ARG(noise && ids_shuffle && ids_restore && mask, "ucfvit_mae_mask: null pointer");
    UCF_CHECK_ARG(B >= 0 && L > 0 && len_keep >= 0 && len_keep <= L, "ucfvit_mae_mask: bad shape");
    UCF_CHECK_ARG(L * 4 <= 64 * 1024, "ucfvit_mae_mask: L=%lld exceeds the 16384-token LDS row", (long long)L);
    if (B == 0) return UCFVIT_OK;
    hipLaunchKernelGGL(mae_mask_kernel, dim3((unsigned)B), dim3(256), (size_t)L * 4, (hipStream_t)stream, noise, ids_shuffle,
                       ids_restore, mask, (int)L, (int)len_keep);
    UCF_LAUNCH_CHECK("ucfvit_mae_mask");
    return UCFVIT_OK;
}

static int rows_copy(const void* a, const int64_t* idx, void* b, int64_t B, int64_t L, int64_t R, int64_t D, int64_t idx_stride,
                     int dtype, int scatter, hipStream_t s, const char* name) {
    UCF_CHECK_ARG(a && idx && b, "%s: null pointer", name);
    UCF_CHECK_ARG(DTYPE_OK(dtype), "%s: bad dtype %d", name, dtype);
    UCF_CHECK_ARG(B >= 0 && L > 0 && R >= 0 && R <= L && D > 0 && idx_stride >= R, "%s: bad shape", name);
    const int64_t row_bytes = D * (dtype == UCFVIT_F32 ? 4 : 2);
    if (scatter) {
        hipError_t e = hipMemsetAsync(b, 0, (size_t)(B * L * row_bytes), s);
        if (e != hipSuccess) {
            ucfvit_set_error("%s: memset failed: %s", name, hipGetErrorString(e));
            return UCFVIT_ERR_HIP;
        }
    }
    if (B == 0 || R == 0) return UCFVIT_OK;
    if (row_bytes % 16 == 0 && ucf_is_aligned16(a) && ucf_is_aligned16(b)) {
        const int nvec = (int)(row_bytes / 16);
        const unsigned grid = ew_grid(B * R * nvec);
        if (scatter)
            hipLaunchKernelGGL(scatter_rows_kernel<u32x4>, dim3(grid), dim3(256), 0, s, (const u32x4*)a, idx, (u32x4*)b, B, (int)L, (int)R, nvec, idx_stride);
        else
            hipLaunchKernelGGL(gather_rows_kernel<u32x4>, dim3(grid), dim3(256), 0, s, (const u32x4*)a, idx, (u32x4*)b, B, (int)L, (int)R, nvec, idx_stride);
    } else if (row_bytes % 2 == 0) {
        const int nvec = (int)(row_bytes / 2);
        const unsigned grid = ew_grid(B * R * nvec);
        if (scatter)
            hipLaunchKernelGGL(scatter_rows_kernel<unsigned short>, dim3(grid), dim3(256), 0, s, (const unsigned short*)a, idx, (unsigned short*)b, B, (int)L, (int)R, nvec, idx_stride);
        else
            hipLaunchKernelGGL(gather_rows_kernel<unsigned short>, dim3(grid), dim3(256), 0, s, (const unsigned short*)a, idx, (unsigned short*)b, B, (int)L, (int)R, nvec, idx_stride);
    }
    UCF_LAUNCH_CHECK(name);
    return UCFVIT_OK;
}

extern "C" int ucfvit_gather_rows(const void* src, const int64_t* idx, void* out, int64_t B, int64_t L, int64_t R, int64_t D,
                                  int64_t idx_stride, int dtype, void* stream) {
    return rows_copy(src, idx, out, B, L, R, D, idx_stride, dtype, 0, (hipStream_t)stream, "ucfvit_gather_rows");
}
extern "C" int ucfvit_scatter_rows(const void* dout, const int64_t* idx, void* dsrc, int64_t B, int64_t L, int64_t R, int64_t D,
                                   int64_t idx_stride, int dtype, void* stream) {
    return rows_copy(dout, idx, dsrc, B, L, R, D, idx_stride, dtype, 1, (hipStream_t)stream, "ucfvit_scatter_rows");
}

extern "C" int ucfvit_unshuffle_fwd(const void* x, const void* mask_token, const int64_t* ids_restore, const void* pos, void* out,
                                    int64_t B, int64_t L, int64_t R, int64_t D, int dtype, void* stream) {
    UCF_CHECK_ARG(x && mask_token && ids_restore && out, "ucfvit_unshuffle_fwd: null pointer");
    UCF_CHECK_ARG(DTYPE_OK(dtype), "ucfvit_unshuffle_fwd: bad dtype %d", dtype);
    const int epv = dtype == UCFVIT_F32 ? 4 : 8;
    UCF_CHECK_ARG(D % epv == 0 && R <= L, "ucfvit_unshuffle_fwd: bad shape (D=%lld must be a multiple of %d)", (long long)D, epv);
    UCF_CHECK_ARG(ucf_is_aligned16(x) && ucf_is_aligned16(mask_token) && ucf_is_aligned16(pos) && ucf_is_aligned16(out),
                  "ucfvit_unshuffle_fwd: pointers must be 16-byte aligned");
    if (B == 0) return UCFVIT_OK;
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = ew_grid(B * L * (D / epv));
    if (dtype == UCFVIT_F32)
        hipLaunchKernelGGL(unshuffle_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (const float*)mask_token,
                           ids_restore, (const float*)pos, (float*)out, B, (int)L, (int)R, (int)D);
    else
        hipLaunchKernelGGL(unshuffle_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)x, (const bf16*)mask_token,
                           ids_restore, (const bf16*)pos, (bf16*)out, B, (int)L, (int)R, (int)D);
    UCF_LAUNCH_CHECK("ucfvit_unshuffle_fwd");
    return UCFVIT_OK;
}

namespace {
__global__ void reduce_rows_kernel(const float* __restrict__ partial, float* __restrict__ out, int rows, int W, int accumulate) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= W) return;
    float s = 0.f;
    for (int b = 0; b < rows; ++b) s += partial[(int64_t)b * W + j];
    out[j] = accumulate ? out[j] + s : s;
}
}  // namespace
int ucfvit_reduce_rows_f32(const float* partial, float* out, int64_t rows, int64_t W, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, s, partial, out, (int)rows, (int)W, accumulate);
    UCF_LAUNCH_CHECK("ucfvit_reduce_rows_f32");
    return UCFVIT_OK;
}


extern "C" int64_t ucfvit_unshuffle_bwd_workspace(int64_t B, int64_t D) { return B * D * (int64_t)sizeof(float); }

extern "C" int ucfvit_unshuffle_bwd(const void* dout, const int64_t* ids_restore, void* dx, float* dmask_token, float* dpos,
                                    int64_t B, int64_t L, int64_t R, int64_t D, int accumulate, void* workspace, int dtype,
                                    void* stream) {
    UCF_CHECK_ARG(dout && ids_restore && dx && dmask_token && workspace, "ucfvit_unshuffle_bwd: null pointer");
    UCF_CHECK_ARG(DTYPE_OK(dtype), "ucfvit_unshuffle_bwd: bad dtype %d", dtype);
    const int epv = dtype == UCFVIT_F32 ? 4 : 8;
    UCF_CHECK_ARG(D % epv == 0 && R <= L && B > 0, "ucfvit_unshuffle_bwd: bad shape");
    UCF_CHECK_ARG(ucf_is_aligned16(dout) && ucf_is_aligned16(dx), "ucfvit_unshuffle_bwd: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    const int threads = (int)((D / epv + 63) / 64) * 64 > 256 ? 256 : (int)((D / epv + 63) / 64) * 64;
    if (dtype == UCFVIT_F32)
        hipLaunchKernelGGL(unshuffle_bwd_kernel<float>, dim3((unsigned)B), dim3(threads), 0, s, (const float*)dout, ids_restore,
                           (float*)dx, partial, (int)L, (int)R, (int)D);
    else
        hipLaunchKernelGGL(unshuffle_bwd_kernel<bf16>, dim3((unsigned)B), dim3(threads), 0, s, (const bf16*)dout, ids_restore,
                           (bf16*)dx, partial, (int)L, (int)R, (int)D);
    UCF_LAUNCH_CHECK("ucfvit_unshuffle_bwd");
    // dmask_token = ordered sum over b of partial[b][:]
    int rc = ucfvit_reduce_rows_f32(partial, dmask_token, B, D, accumulate, s);
    if (rc) return rc;
    if (dpos) {
        const int64_t work = L * (D / epv);
        const unsigned grid = (unsigned)((work + 127) / 128);
        if (dtype == UCFVIT_F32)
            hipLaunchKernelGGL(batch_sum_kernel<float>, dim3(grid), dim3(128), 0, s, (const float*)dout, dpos, B, (int)L, (int)D, accumulate);
        else
            hipLaunchKernelGGL(batch_sum_kernel<bf16>, dim3(grid), dim3(128), 0, s, (const bf16*)dout, dpos, B, (int)L, (int)D, accumulate);
        UCF_LAUNCH_CHECK("ucfvit_unshuffle_bwd(dpos)");
    }
    return UCFVIT_OK;
}

extern "C" int ucfvit_patch_mse(const void* pred, const float* img, const float* mask, float* loss, void* dpred, int64_t B, int64_t C,
                                const int64_t* dims, int nd, int64_t p, float grad_scale, float* workspace, int dtype, void* stream) {
    UCF_CHECK_ARG(pred && img && loss && workspace && dims, "ucfvit_patch_mse: null pointer");
    UCF_CHECK_ARG(nd == 1 || nd == 2 || nd == 3, "ucfvit_patch_mse: nd must be 1 (token sequence), 2 or 3");
    UCF_CHECK_ARG(DTYPE_OK(dtype), "ucfvit_patch_mse: bad dtype %d", dtype);
    UCF_CHECK_ARG(B > 0 && C > 0 && p > 0, "ucfvit_patch_mse: bad sizes");
    if (nd == 1)
        UCF_CHECK_ARG(dims[0] > 0 && dims[0] < (1ll << 31) && C * p < (1ll << 31), "ucfvit_patch_mse: bad sequence length");
    else
        for (int i = 0; i < nd; ++i) UCF_CHECK_ARG(dims[i] > 0 && dims[i] % p == 0, "ucfvit_patch_mse: image dim %d not a multiple of p", i);
    PatchGeom g;
    g.C = (int)C;
    g.H = (int)dims[0];
    g.W = (int)dims[1];
    g.Z = nd == 3 ? (int)dims[2] : 1;
    g.p = (int)p;
    g.nd = nd;
    g.gh = g.H / g.p;
    g.gw = g.W / g.p;
    g.gz = nd == 3 ? g.Z / g.p : 1;
    g.L = g.gh * g.gw * g.gz;
    g.P = (int)(C * p * p * (nd == 3 ? p : 1));
    if (nd == 1) {  // dims = {S}, p = pixels per token and channel
        g.H = g.W = g.Z = g.gh = g.gw = g.gz = 1;
        g.L = (int)dims[0];
        g.P = (int)(C * p);
    }
    const int64_t total = B * g.L * g.P;
    hipStream_t s = (hipStream_t)stream;
    int nb = (int)((total + 256 * 64 - 1) / (256 * 64));
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    if (dtype == UCFVIT_F32)
        hipLaunchKernelGGL(patch_mse_partial_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)pred, img, mask, workspace, g, B);
    else
        hipLaunchKernelGGL(patch_mse_partial_kernel<bf16>, dim3(nb), dim3(256), 0, s, (const bf16*)pred, img, mask, workspace, g, B);
    UCF_LAUNCH_CHECK("ucfvit_patch_mse(partial)");
    hipLaunchKernelGGL(patch_mse_final_kernel, dim3(1), dim3(64), 0, s, workspace, loss, nb, mask ? 1 : 0, (float)total, (float)g.P);
    UCF_LAUNCH_CHECK("ucfvit_patch_mse(final)");
    if (dpred) {
        const unsigned grid = ew_grid(total);
        if (dtype == UCFVIT_F32)
            hipLaunchKernelGGL(patch_mse_grad_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)pred, img, mask, workspace,
                               (float*)dpred, g, B, grad_scale);
        else
            hipLaunchKernelGGL(patch_mse_grad_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)pred, img, mask, workspace,
                               (bf16*)dpred, g, B, grad_scale);
        UCF_LAUNCH_CHECK("ucfvit_patch_mse(grad)");
    }
    return UCFVIT_OK;
}

extern "C" int ucfvit_adamw(float* p, const void* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, float grad_scale,
                            int grad_dtype, void* stream) {
    UCF_CHECK_ARG(p && g && m && v, "ucfvit_adamw: null pointer");
    UCF_CHECK_ARG(n >= 0, "ucfvit_adamw: negative size");
    UCF_CHECK_ARG(DTYPE_OK(grad_dtype), "ucfvit_adamw: bad grad dtype %d", grad_dtype);
    UCF_CHECK_ARG(ucf_is_aligned16(p) && ucf_is_aligned16(m) && ucf_is_aligned16(v) && (((uintptr_t)g) % 8 == 0) &&
                      (((uintptr_t)shadow_bf16) % 8 == 0),
                  "ucfvit_adamw: pointers must be 16-byte aligned (grad/shadow 8)");
    if (n == 0) return UCFVIT_OK;
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = ew_grid((n + 3) / 4);
    const float bc2s = sqrtf(bias_corr2);
    if (grad_dtype == UCFVIT_F32) {
        UCF_CHECK_ARG(ucf_is_aligned16(g), "ucfvit_adamw: fp32 grads must be 16-byte aligned");
        hipLaunchKernelGGL(adamw_kernel<float>, dim3(grid), dim3(256), 0, s, p, (const float*)g, m, v, (bf16*)shadow_bf16, n, lr, beta1,
                           beta2, eps, weight_decay, bias_corr1, bc2s, grad_scale);
    } else {
        hipLaunchKernelGGL(adamw_kernel<bf16>, dim3(grid), dim3(256), 0, s, p, (const bf16*)g, m, v, (bf16*)shadow_bf16, n, lr, beta1,
                           beta2, eps, weight_decay, bias_corr1, bc2s, grad_scale);
    }
    UCF_LAUNCH_CHECK("ucfvit_adamw");
    return UCFVIT_OK;
}

extern "C" int ucfvit_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, float scale, void* stream) {
    UCF_CHECK_ARG(src && dst, "ucfvit_cast: null pointer");
    UCF_CHECK_ARG(DTYPE_OK(src_dtype) && DTYPE_OK(dst_dtype), "ucfvit_cast: bad dtype");
    UCF_CHECK_ARG(n >= 0, "ucfvit_cast: negative size");
    const size_t sa = src_dtype == UCFVIT_F32 ? 16 : 8, da = dst_dtype == UCFVIT_F32 ? 16 : 8;
    UCF_CHECK_ARG(((uintptr_t)src) % sa == 0 && ((uintptr_t)dst) % da == 0, "ucfvit_cast: misaligned pointer");
    if (n == 0) return UCFVIT_OK;
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = ew_grid((n + 3) / 4);
#define CAST(S, Dd) hipLaunchKernelGGL((cast_kernel<S, Dd>), dim3(grid), dim3(256), 0, s, (const S*)src, (Dd*)dst, n, scale)
    if (src_dtype == UCFVIT_F32 && dst_dtype == UCFVIT_BF16) CAST(float, bf16);
    else if (src_dtype == UCFVIT_BF16 && dst_dtype == UCFVIT_F32) CAST(bf16, float);
    else if (src_dtype == UCFVIT_F32) CAST(float, float);
    else CAST(bf16, bf16);
#undef CAST
    UCF_LAUNCH_CHECK("ucfvit_cast");
    return UCFVIT_OK;
}

// ===================================================================================================
// Batched 2-D transpose of bf16 matrices living in one flat buffer (the transposed weight shadow used by the
// data-gradient GEMMs, so they run contraction-contiguous on both operands).  table: int64 [n][5] =
// {src_off, dst_off, rows, cols, first_tile}; 64x64 tiles through LDS, 16-byte global accesses.
// ===================================================================================================
namespace {
__global__ __launch_bounds__(256) void transpose_batched_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                                                const int64_t* __restrict__ table, int n_mats) {
    __shared__ unsigned short tile[64][64 + 8];
    int lo = 0, hi = n_mats - 1;
    const int64_t bid = blockIdx.x;
    while (lo < hi) {  // last matrix whose first_tile <= bid
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 5 + 4] <= bid) lo = mid; else hi = mid - 1;
    }
    const int64_t* t = table + lo * 5;
    const int64_t rows = t[2], cols = t[3];
    const int64_t tiles_c = (cols + 63) / 64;
    const int64_t lt = bid - t[4];
    const int64_t r0 = (lt / tiles_c) * 64, c0 = (lt % tiles_c) * 64;
    const unsigned short* s = src + t[0];
    unsigned short* d = dst + t[1];
    // load 64 rows x 64 cols: thread -> (row = tid/8 + 32*i, 8 cols)
    const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t r = r0 + tr + 32 * i, c = c0 + tc;
        if (r < rows && c < cols) {  // cols % 8 == 0 (host-checked)
            const u32x4 v = *reinterpret_cast<const u32x4*>(s + r * cols + c);
            *reinterpret_cast<u32x4*>(&tile[tr + 32 * i][tc]) = v;
        }
    }
    __syncthreads();
    // store transposed: output row = c0 + tr', 8 consecutive output cols = input rows r0 + tc' .. +7
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oc = tr + 32 * i;               // input col within tile = output row
        const int64_t orow = c0 + oc, ocol = r0 + tc;
        if (orow < cols && ocol < rows) {         // rows % 8 == 0 (host-checked)
            unsigned short v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[tc + e][oc];
            *reinterpret_cast<u32x4*>(d + orow * rows + ocol) = *reinterpret_cast<const u32x4*>(v);
        }
    }
}
}  // namespace

extern "C" int ucfvit_transpose_batched(const void* src, void* dst, const int64_t* table, int64_t n_mats, int64_t total_tiles, void* stream) {
    UCF_CHECK_ARG(src && dst && table, "ucfvit_transpose_batched: null pointer");
    UCF_CHECK_ARG(n_mats > 0 && total_tiles > 0 && total_tiles < (1ll << 31), "ucfvit_transpose_batched: bad sizes");
    UCF_CHECK_ARG(ucf_is_aligned16(src) && ucf_is_aligned16(dst), "ucfvit_transpose_batched: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(transpose_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src,
                       (unsigned short*)dst, table, (int)n_mats);
    UCF_LAUNCH_CHECK("ucfvit_transpose_batched");
    return UCFVIT_OK;
}

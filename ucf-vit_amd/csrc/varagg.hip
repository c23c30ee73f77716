// Variable aggregation for gfx950 (VariableMapping_Attention, reference src/UCF_VIT/simple/building_blocks.py:301-373, called by
// VIT.aggregate_variables, simple/arch.py:414-432): for every token position the V per-variable embeddings are compressed into one
// by a cross-attention with ONE learnt query: scores_v = scale * q_h . k_{v,h}, softmax over the V variables, out_h = sum_v p_v v_{v,h}.
// V is small (the channels of the input), so this is an HBM-bound stream over kv [V][R][2D]: one pass, online softmax per head.
//   thread = one 16-byte column vector of one token row; the D/EPV threads of a row sit side by side, the dh/EPV threads of a head
//   form an aligned power-of-two lane group reduced with shuffles.  Backward recomputes the probabilities from the saved
//   log-sum-exp and writes dkv plus one dq row per token (summed afterwards by ucfvit_colsum): no atomics.
#include "common.h"

namespace {

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float group_sum_rt(float v, int g) {
    for (int o = g >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// kv: [V][R][2D] (k = columns [0, D), v = columns [D, 2D)), q fp32 [D], out [R][D], lse fp32 [R][H]
template <typename T>
__global__ __launch_bounds__(256) void varagg_fwd_kernel(const T* __restrict__ kv, const float* __restrict__ q, T* __restrict__ out,
                                                         float* __restrict__ lse, int64_t R, int V, int D, int dh, float scale) {
    constexpr int EPV = Vec16<T>::N;
    const int tpr = D / EPV, rpb = 256 / tpr, g = dh / EPV;
    const int rl = threadIdx.x / tpr, cv = threadIdx.x - rl * tpr;
    const int64_t row = (int64_t)blockIdx.x * rpb + rl;
    const bool live = rl < rpb && row < R;
    const int col = cv * EPV;
    float q8[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) q8[e] = live ? q[col + e] * scale : 0.f;
    float m = -INFINITY, ssum = 0.f, acc[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
    const int64_t r = live ? row : 0;
    for (int v = 0; v < V; ++v) {
        const T* base = kv + ((int64_t)v * R + r) * (2 * D);
        const Vec16<T> kk = *reinterpret_cast<const Vec16<T>*>(base + col);
        const Vec16<T> vv = *reinterpret_cast<const Vec16<T>*>(base + D + col);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPV; ++e) s += q8[e] * kk.get(e);
        s = group_sum_rt(s, g);
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn), p = __expf(s - mn);
        ssum = ssum * corr + p;
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[e] = acc[e] * corr + p * vv.get(e);
        m = mn;
    }
    if (!live) return;
    const float inv = 1.f / ssum;
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < EPV; ++e) o.set(e, acc[e] * inv);
    *reinterpret_cast<Vec16<T>*>(out + row * D + col) = o;
    if ((cv % g) == 0) lse[row * (D / dh) + cv / g] = m + __logf(ssum);
}

// dkv [V][R][2D], dq_rows fp32 [R][D] (sum over rows = dq of the single query)
template <typename T>
__global__ __launch_bounds__(256) void varagg_bwd_kernel(const T* __restrict__ kv, const float* __restrict__ q, const T* __restrict__ out,
                                                         const float* __restrict__ lse, const T* __restrict__ dout, T* __restrict__ dkv,
                                                         float* __restrict__ dq_rows, int64_t R, int V, int D, int dh, float scale) {
    constexpr int EPV = Vec16<T>::N;
    const int tpr = D / EPV, rpb = 256 / tpr, g = dh / EPV;
    const int rl = threadIdx.x / tpr, cv = threadIdx.x - rl * tpr;
    const int64_t row = (int64_t)blockIdx.x * rpb + rl;
    const bool live = rl < rpb && row < R;
    const int col = cv * EPV;
    const int64_t r = live ? row : 0;
    float q8[EPV], do8[EPV], dq8[EPV];
    const Vec16<T> dov = *reinterpret_cast<const Vec16<T>*>(dout + r * D + col);
    const Vec16<T> ov = *reinterpret_cast<const Vec16<T>*>(out + r * D + col);
    float delta = 0.f;
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        q8[e] = q[col + e];
        do8[e] = dov.get(e);
        dq8[e] = 0.f;
        delta += do8[e] * ov.get(e);
    }
    delta = group_sum_rt(delta, g);                      // sum_v p_v dp_v = dout . out (per head)
    const float l = lse[r * (D / dh) + cv / g];
    for (int v = 0; v < V; ++v) {
        const int64_t off = ((int64_t)v * R + r) * (2 * D);
        const Vec16<T> kk = *reinterpret_cast<const Vec16<T>*>(kv + off + col);
        const Vec16<T> vv = *reinterpret_cast<const Vec16<T>*>(kv + off + D + col);
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            s += q8[e] * kk.get(e);
            dp += do8[e] * vv.get(e);
        }
        s = group_sum_rt(s, g) * scale;
        dp = group_sum_rt(dp, g);
        const float p = __expf(s - l);
        const float ds = p * (dp - delta) * scale;
        Vec16<T> dk, dv;
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            dk.set(e, ds * q8[e]);
            dv.set(e, p * do8[e]);
            dq8[e] += ds * kk.get(e);
        }
        if (live) {
            *reinterpret_cast<Vec16<T>*>(dkv + off + col) = dk;
            *reinterpret_cast<Vec16<T>*>(dkv + off + D + col) = dv;
        }
    }
    if (live) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) dq_rows[row * D + col + e] = dq8[e];
    }
}

inline bool varagg_shape_ok(int64_t D, int64_t dh, int epv) {
    if (D <= 0 || dh <= 0 || D % dh || D % epv || dh % epv) return false;
    const int64_t tpr = D / epv, g = dh / epv;
    return tpr <= 256 && g >= 1 && g <= 64 && (g & (g - 1)) == 0;
}

}  // namespace

extern "C" int ucfvit_varagg_fwd(const void* kv, const float* q, void* out, float* lse, int64_t R, int64_t V, int64_t D, int64_t dh,
                                 float scale, int dtype, void* stream) {
    if (R == 0) return UCFVIT_OK;
    UCF_CHECK_ARG(kv && q && out && lse, "ucfvit_varagg_fwd: null pointer");
    UCF_CHECK_ARG(dtype == UCFVIT_F32 || dtype == UCFVIT_BF16, "ucfvit_varagg_fwd: bad dtype %d", dtype);
    const int epv = dtype == UCFVIT_F32 ? 4 : 8;
    UCF_CHECK_ARG(V >= 1 && varagg_shape_ok(D, dh, epv),
                  "ucfvit_varagg_fwd: D=%lld head_dim=%lld: need head_dim | D, both multiples of %d, D/%d <= 256, head_dim/%d a power of two <= 64",
                  (long long)D, (long long)dh, epv, epv, epv);
    UCF_CHECK_ARG(ucf_is_aligned16(kv) && ucf_is_aligned16(out), "ucfvit_varagg_fwd: pointers must be 16-byte aligned");
    const int rpb = 256 / (int)(D / epv);
    const unsigned grid = (unsigned)((R + rpb - 1) / rpb);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UCFVIT_F32)
        hipLaunchKernelGGL(varagg_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)kv, q, (float*)out, lse, R, (int)V, (int)D, (int)dh, scale);
    else
        hipLaunchKernelGGL(varagg_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)kv, q, (bf16*)out, lse, R, (int)V, (int)D, (int)dh, scale);
    UCF_LAUNCH_CHECK("ucfvit_varagg_fwd");
    return UCFVIT_OK;
}

extern "C" int ucfvit_varagg_bwd(const void* kv, const float* q, const void* out, const float* lse, const void* dout, void* dkv, float* dq_rows,
                                 int64_t R, int64_t V, int64_t D, int64_t dh, float scale, int dtype, void* stream) {
    if (R == 0) return UCFVIT_OK;
    UCF_CHECK_ARG(kv && q && out && lse && dout && dkv && dq_rows, "ucfvit_varagg_bwd: null pointer");
    UCF_CHECK_ARG(dtype == UCFVIT_F32 || dtype == UCFVIT_BF16, "ucfvit_varagg_bwd: bad dtype %d", dtype);
    const int epv = dtype == UCFVIT_F32 ? 4 : 8;
    UCF_CHECK_ARG(V >= 1 && varagg_shape_ok(D, dh, epv), "ucfvit_varagg_bwd: unsupported D=%lld head_dim=%lld", (long long)D, (long long)dh);
    UCF_CHECK_ARG(ucf_is_aligned16(kv) && ucf_is_aligned16(out) && ucf_is_aligned16(dout) && ucf_is_aligned16(dkv),
                  "ucfvit_varagg_bwd: pointers must be 16-byte aligned");
    const int rpb = 256 / (int)(D / epv);
    const unsigned grid = (unsigned)((R + rpb - 1) / rpb);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UCFVIT_F32)
        hipLaunchKernelGGL(varagg_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)kv, q, (const float*)out, lse, (const float*)dout,
                           (float*)dkv, dq_rows, R, (int)V, (int)D, (int)dh, scale);
    else
        hipLaunchKernelGGL(varagg_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)kv, q, (const bf16*)out, lse, (const bf16*)dout,
                           (bf16*)dkv, dq_rows, R, (int)V, (int)D, (int)dh, scale);
    UCF_LAUNCH_CHECK("ucfvit_varagg_bwd");
    return UCFVIT_OK;
}

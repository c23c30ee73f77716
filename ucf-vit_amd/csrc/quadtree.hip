// GPU-side fixed-length quadtree patcher for gfx950: the adaptive-patching data transform of the reference
// (src/UCF_VIT/dataloaders/quadtree.py:84-174 FixedQuadTree, dataloaders/transform.py:9-55 Patchify) for a whole batch on the device.
//   quadtree_build     : per image, the reference's greedy refinement — repeatedly replace the FIRST node of maximum edge count by
//                        its four quadrants (order lt, rt, lb, rb, in place) until fixed_length nodes exist or that node is 2 pixels
//                        wide.  Integer logic, bit-exact; region sums come from a summed-area table (one pass per image).
//   quadtree_serialize : every node's region resampled to p x p with the cv2.INTER_CUBIC / torch bicubic kernel (A = -0.75, pixel
//                        centres aligned, replicated border), written in the reference's [S][p][p][C] order.
// One workgroup per image (build) / per node (serialize); the greedy loop is sequential by definition, the batch supplies the
// parallelism (166 images = 166 workgroups).  Edge detection (cv2.Canny) is not part of this file: the edge map is an input.
#include "common.h"

namespace {

struct QNode {
    short x1, x2, y1, y2;
    int v;
};

__device__ __forceinline__ unsigned sat_sum(const unsigned* __restrict__ sat, int W1, int x1, int x2, int y1, int y2) {
    return sat[y2 * W1 + x2] - sat[y1 * W1 + x2] - sat[y2 * W1 + x1] + sat[y1 * W1 + x1];
}

__global__ __launch_bounds__(256) void quadtree_build_kernel(const unsigned char* __restrict__ edges, int* __restrict__ nodes_out,
                                                             int* __restrict__ values_out, int* __restrict__ count_out,
                                                             float* __restrict__ seq_ps, unsigned* __restrict__ sat_all, int H, int W, int L) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    QNode* bufA = reinterpret_cast<QNode*>(smem_raw);
    QNode* bufB = bufA + (L + 4);
    __shared__ int red_v[4], red_i[4];
    __shared__ int s_idx, s_stop;
    __shared__ QNode s_kids[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W1 = W + 1;
    const unsigned char* e = edges + (int64_t)b * H * W;
    unsigned* sat = sat_all + (int64_t)b * (H + 1) * W1;

    // ---- summed-area table: sat[y][x] = sum of e[0..y)[0..x)  (row prefix sums, then column prefix sums)
    for (int x = tid; x < W1; x += blockDim.x) sat[x] = 0;
    for (int y = tid; y < H; y += blockDim.x) {
        unsigned run = 0;
        unsigned* row = sat + (int64_t)(y + 1) * W1;
        row[0] = 0;
        for (int x = 0; x < W; ++x) {
            run += e[(int64_t)y * W + x];
            row[x + 1] = run;
        }
    }
    __syncthreads();
    for (int x = tid; x < W1; x += blockDim.x) {
        unsigned run = 0;
        for (int y = 1; y <= H; ++y) {
            run += sat[(int64_t)y * W1 + x];
            sat[(int64_t)y * W1 + x] = run;
        }
    }
    __syncthreads();

    // ---- greedy refinement (quadtree.py:115-140)
    QNode* cur = bufA;
    QNode* nxt = bufB;
    int n = 1;
    if (tid == 0) {
        cur[0] = QNode{0, (short)W, 0, (short)H, (int)(sat_sum(sat, W1, 0, W, 0, H) / 255u)};
        s_stop = 0;
    }
    __syncthreads();
    while (n < L) {
        // first index of the maximum value: max over (v, -i)
        int bv = -1, bi = 0x7fffffff;
        for (int i = tid; i < n; i += blockDim.x) {
            const int v = cur[i].v;
            if (v > bv) {          // i ascends per thread: the first maximum of this thread's subsequence is kept
                bv = v;
                bi = i;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ov = __shfl_xor(bv, o, 64), oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            red_v[wave] = bv;
            red_i[wave] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            int v = red_v[0], i = red_i[0];
            for (int w = 1; w < 4; ++w)
                if (red_v[w] > v || (red_v[w] == v && red_i[w] < i)) {
                    v = red_v[w];
                    i = red_i[w];
                }
            s_idx = i;
            if (cur[i].x2 - cur[i].x1 == 2) s_stop = 1;            // :121-122
        }
        __syncthreads();
        if (s_stop) break;
        const int idx = s_idx;
        if (tid < 4) {
            const QNode q = cur[idx];
            const int mx = (q.x1 + q.x2) >> 1, my = (q.y1 + q.y2) >> 1;      // int((a + b) / 2) for non-negative ints
            QNode k;
            if (tid == 0) k = QNode{q.x1, (short)mx, (short)my, q.y2, 0};            // lt (:125)
            else if (tid == 1) k = QNode{(short)mx, q.x2, (short)my, q.y2, 0};       // rt
            else if (tid == 2) k = QNode{q.x1, (short)mx, q.y1, (short)my, 0};       // lb
            else k = QNode{(short)mx, q.x2, q.y1, (short)my, 0};                     // rb
            k.v = (int)(sat_sum(sat, W1, k.x1, k.x2, k.y1, k.y2) / 255u);
            s_kids[tid] = k;
        }
        __syncthreads();
        // nodes = nodes[:idx] + kids + nodes[idx+1:]   (:134), written into the other buffer
        for (int j = tid; j < n + 3; j += blockDim.x) nxt[j] = j < idx ? cur[j] : (j < idx + 4 ? s_kids[j - idx] : cur[j - 3]);
        __syncthreads();
        QNode* t = cur;
        cur = nxt;
        nxt = t;
        n += 3;
    }
    // ---- outputs: the first min(n, L) nodes (n == L whenever fixed_length = 3k + 1), padding as serialize() pads (:160-168)
    const int nv = n < L ? n : L;
    if (tid == 0) count_out[b] = nv;
    for (int i = tid; i < L; i += blockDim.x) {
        int* no = nodes_out + ((int64_t)b * L + i) * 4;
        float* sp = seq_ps + ((int64_t)b * L + i) * 3;
        if (i < nv) {
            const QNode q = cur[i];
            no[0] = q.x1;
            no[1] = q.x2;
            no[2] = q.y1;
            no[3] = q.y2;
            values_out[(int64_t)b * L + i] = q.v;
            sp[0] = (float)(q.x2 - q.x1);                       // seq_size: the width (:151)
            sp[1] = (float)(q.x2 + q.x1) * 0.5f;                // seq_pos: centre (:152, Rect.get_center)
            sp[2] = (float)(q.y2 + q.y1) * 0.5f;
        } else {
            no[0] = no[1] = no[2] = no[3] = 0;
            values_out[(int64_t)b * L + i] = 0;
            sp[0] = 0.f;
            sp[1] = -1.f;
            sp[2] = -1.f;
        }
    }
}

// cubic convolution coefficients, A = -0.75 (cv2.INTER_CUBIC, torch upsample_bicubic2d)
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
    constexpr float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__global__ __launch_bounds__(256) void quadtree_serialize_kernel(const float* __restrict__ img, const int* __restrict__ nodes,
                                                                 const int* __restrict__ count, float* __restrict__ seq, int H, int W, int C,
                                                                 int L, int p) {
    const int64_t bs = blockIdx.x;
    const int b = (int)(bs / L), s = (int)(bs - (int64_t)b * L);
    const int n_out = p * p * C;
    float* out = seq + bs * n_out;
    const int* q = nodes + bs * 4;
    const int x1 = q[0], x2 = q[1], y1 = q[2], y2 = q[3];
    const int w = x2 - x1, h = y2 - y1;
    if (s >= count[b] || w <= 0 || h <= 0) {
        for (int i = threadIdx.x; i < n_out; i += blockDim.x) out[i] = 0.f;
        return;
    }
    const float sx = (float)w / (float)p, sy = (float)h / (float)p;
    const float* src = img + (int64_t)b * H * W * C;
    for (int i = threadIdx.x; i < n_out; i += blockDim.x) {
        const int c = i % C;
        const int px = (i / C) % p, py = i / (C * p);
        const float fy = ((float)py + 0.5f) * sy - 0.5f, fx = ((float)px + 0.5f) * sx - 0.5f;
        const float fyf = floorf(fy), fxf = floorf(fx);
        const int iy = (int)fyf, ix = (int)fxf;
        float wy[4], wx[4];
        cubic_coeffs(fy - fyf, wy);
        cubic_coeffs(fx - fxf, wx);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int yy = iy - 1 + a;
            yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
            const float* rowp = src + ((int64_t)(y1 + yy) * W + x1) * C + c;
            float r = 0.f;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                int xx = ix - 1 + d;
                xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
                r += wx[d] * rowp[(int64_t)xx * C];
            }
            acc += wy[a] * r;
        }
        out[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Octree (3-D volumes): dataloaders/octree.py:66-151 FixedOctTree.  domain[z][y][x] uint8, cubic; value = sum / norm_factor;
// children n1..n8 = x fastest, then y, then z (:83-103); stop when the first maximum node is 2 wide in x; 7 nodes added per step.
// ---------------------------------------------------------------------------------------------------------------------
struct ONode {
    short x1, x2, y1, y2, z1, z2;
    int v;
};

// summed-volume table sat[z][y][x] = sum of dom[0..z)[0..y)[0..x), (N+1)^3 entries per volume, zero-initialised by the caller
__global__ void sat3_x_kernel(const unsigned char* __restrict__ dom, unsigned* __restrict__ sat, int B, int N) {
    const int64_t line = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // (b, z, y)
    if (line >= (int64_t)B * N * N) return;
    const int y = (int)(line % N), z = (int)((line / N) % N);
    const int64_t b = line / ((int64_t)N * N);
    const int N1 = N + 1;
    const unsigned char* src = dom + ((b * N + z) * N + y) * (int64_t)N;
    unsigned* dst = sat + ((b * N1 + z + 1) * N1 + y + 1) * (int64_t)N1;
    unsigned run = 0;
    for (int x = 0; x < N; ++x) {
        run += src[x];
        dst[x + 1] = run;
    }
}
__global__ void sat3_y_kernel(unsigned* __restrict__ sat, int B, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // (b, z, x')
    if (i >= (int64_t)B * N * N) return;
    const int x = (int)(i % N) + 1, z = (int)((i / N) % N) + 1;
    const int64_t b = i / ((int64_t)N * N);
    const int N1 = N + 1;
    unsigned* base = sat + ((b * N1 + z) * N1) * (int64_t)N1 + x;
    unsigned run = 0;
    for (int y = 1; y <= N; ++y) {
        run += base[(int64_t)y * N1];
        base[(int64_t)y * N1] = run;
    }
}
__global__ void sat3_z_kernel(unsigned* __restrict__ sat, int B, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // (b, y', x')
    if (i >= (int64_t)B * N * N) return;
    const int x = (int)(i % N) + 1, y = (int)((i / N) % N) + 1;
    const int64_t b = i / ((int64_t)N * N);
    const int N1 = N + 1;
    unsigned* base = sat + (b * N1 * N1 + y) * (int64_t)N1 + x;
    unsigned run = 0;
    for (int z = 1; z <= N; ++z) {
        run += base[(int64_t)z * N1 * N1];
        base[(int64_t)z * N1 * N1] = run;
    }
}

__device__ __forceinline__ unsigned sat3_sum(const unsigned* __restrict__ sat, int N1, int x1, int x2, int y1, int y2, int z1, int z2) {
    auto at = [&](int z, int y, int x) { return sat[((int64_t)z * N1 + y) * N1 + x]; };
    return at(z2, y2, x2) - at(z1, y2, x2) - at(z2, y1, x2) - at(z2, y2, x1) + at(z1, y1, x2) + at(z1, y2, x1) + at(z2, y1, x1) - at(z1, y1, x1);
}

__global__ __launch_bounds__(256) void octree_build_kernel(int* __restrict__ nodes_out, int* __restrict__ values_out, int* __restrict__ count_out,
                                                           float* __restrict__ seq_ps, const unsigned* __restrict__ sat_all, int N, int L,
                                                           unsigned norm) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ONode* cur = reinterpret_cast<ONode*>(smem_raw);
    ONode* nxt = cur + (L + 8);
    __shared__ int red_v[4], red_i[4];
    __shared__ int s_idx, s_stop;
    __shared__ ONode s_kids[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N1 = N + 1;
    const unsigned* sat = sat_all + (int64_t)b * N1 * N1 * N1;
    int n = 1;
    if (tid == 0) {
        cur[0] = ONode{0, (short)N, 0, (short)N, 0, (short)N, (int)(sat3_sum(sat, N1, 0, N, 0, N, 0, N) / norm)};
        s_stop = 0;
    }
    __syncthreads();
    while (n < L) {
        int bv = -1, bi = 0x7fffffff;
        for (int i = tid; i < n; i += blockDim.x) {
            const int v = cur[i].v;
            if (v > bv) {
                bv = v;
                bi = i;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ov = __shfl_xor(bv, o, 64), oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            red_v[wave] = bv;
            red_i[wave] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            int v = red_v[0], i = red_i[0];
            for (int w = 1; w < 4; ++w)
                if (red_v[w] > v || (red_v[w] == v && red_i[w] < i)) {
                    v = red_v[w];
                    i = red_i[w];
                }
            s_idx = i;
            if (cur[i].x2 - cur[i].x1 == 2) s_stop = 1;            // octree.py:79-80
        }
        __syncthreads();
        if (s_stop) break;
        const int idx = s_idx;
        if (tid < 8) {
            const ONode q = cur[idx];
            const int mx = (q.x1 + q.x2) >> 1, my = (q.y1 + q.y2) >> 1, mz = (q.z1 + q.z2) >> 1;
            ONode k;
            k.x1 = (tid & 1) ? (short)mx : q.x1;
            k.x2 = (tid & 1) ? q.x2 : (short)mx;
            k.y1 = (tid & 2) ? (short)my : q.y1;
            k.y2 = (tid & 2) ? q.y2 : (short)my;
            k.z1 = (tid & 4) ? (short)mz : q.z1;
            k.z2 = (tid & 4) ? q.z2 : (short)mz;
            k.v = (int)(sat3_sum(sat, N1, k.x1, k.x2, k.y1, k.y2, k.z1, k.z2) / norm);
            s_kids[tid] = k;
        }
        __syncthreads();
        for (int j = tid; j < n + 7; j += blockDim.x) nxt[j] = j < idx ? cur[j] : (j < idx + 8 ? s_kids[j - idx] : cur[j - 7]);
        __syncthreads();
        ONode* t = cur;
        cur = nxt;
        nxt = t;
        n += 7;
    }
    const int nv = n < L ? n : L;
    if (tid == 0) count_out[b] = nv;
    for (int i = tid; i < L; i += blockDim.x) {
        int* no = nodes_out + ((int64_t)b * L + i) * 6;
        float* sp = seq_ps + ((int64_t)b * L + i) * 4;
        if (i < nv) {
            const ONode q = cur[i];
            no[0] = q.x1; no[1] = q.x2; no[2] = q.y1; no[3] = q.y2; no[4] = q.z1; no[5] = q.z2;
            values_out[(int64_t)b * L + i] = q.v;
            sp[0] = (float)(q.x2 - q.x1);
            sp[1] = (float)(q.x2 + q.x1) * 0.5f;
            sp[2] = (float)(q.y2 + q.y1) * 0.5f;
            sp[3] = (float)(q.z2 + q.z1) * 0.5f;
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) no[k] = 0;
            values_out[(int64_t)b * L + i] = 0;
            sp[0] = 0.f;
            sp[1] = sp[2] = sp[3] = -1.f;
        }
    }
}

// leaf img[z1:z2, y1:y2, x1:x2, :] -> p^3 by linear interpolation with aligned corners (octree.py:120-141), out [S][p][p][p][C]
__global__ __launch_bounds__(256) void octree_serialize_kernel(const float* __restrict__ img, const int* __restrict__ nodes,
                                                               const int* __restrict__ count, float* __restrict__ seq, int N, int C, int L, int p) {
    const int64_t bs = blockIdx.x;
    const int b = (int)(bs / L), s = (int)(bs - (int64_t)b * L);
    const int n_out = p * p * p * C;
    float* out = seq + bs * n_out;
    const int* q = nodes + bs * 6;
    const int x1 = q[0], y1 = q[2], z1 = q[4];
    const int nx = q[1] - x1, ny = q[3] - y1, nz = q[5] - z1;
    if (s >= count[b] || nx <= 0 || ny <= 0 || nz <= 0) {
        for (int i = threadIdx.x; i < n_out; i += blockDim.x) out[i] = 0.f;
        return;
    }
    const float inv = p > 1 ? 1.f / (float)(p - 1) : 0.f;
    const float* src = img + (int64_t)b * N * N * N * C;
    for (int i = threadIdx.x; i < n_out; i += blockDim.x) {
        const int c = i % C;
        int r = i / C;
        const int k = r % p;
        r /= p;
        const int j = r % p, ii = r / p;
        // axis 0 of the leaf is z, axis 1 is y, axis 2 is x
        const float fz = (float)ii * (float)(nz - 1) * inv, fy = (float)j * (float)(ny - 1) * inv, fx = (float)k * (float)(nx - 1) * inv;
        int z0 = (int)fz, y0 = (int)fy, x0 = (int)fx;
        z0 = z0 > nz - 1 ? nz - 1 : z0;
        y0 = y0 > ny - 1 ? ny - 1 : y0;
        x0 = x0 > nx - 1 ? nx - 1 : x0;
        const float tz = fz - (float)z0, ty = fy - (float)y0, tx = fx - (float)x0;
        const int z1i = z0 + 1 < nz ? z0 + 1 : nz - 1, y1i = y0 + 1 < ny ? y0 + 1 : ny - 1, x1i = x0 + 1 < nx ? x0 + 1 : nx - 1;
        auto at = [&](int z, int y, int x) { return src[((((int64_t)(z1 + z) * N + (y1 + y)) * N) + (x1 + x)) * C + c]; };
        const float c00 = at(z0, y0, x0) * (1.f - tx) + at(z0, y0, x1i) * tx;
        const float c01 = at(z0, y1i, x0) * (1.f - tx) + at(z0, y1i, x1i) * tx;
        const float c10 = at(z1i, y0, x0) * (1.f - tx) + at(z1i, y0, x1i) * tx;
        const float c11 = at(z1i, y1i, x0) * (1.f - tx) + at(z1i, y1i, x1i) * tx;
        const float c0 = c00 * (1.f - ty) + c01 * ty, c1 = c10 * (1.f - ty) + c11 * ty;
        out[i] = c0 * (1.f - tz) + c1 * tz;
    }
}

}  // namespace

extern "C" int64_t ucfvit_octree_workspace(int64_t B, int64_t N) {
    if (B <= 0 || N <= 0) return 0;
    return B * (N + 1) * (N + 1) * (N + 1) * (int64_t)sizeof(unsigned);
}

extern "C" int ucfvit_octree_build(const uint8_t* domain, int32_t* nodes, int32_t* values, int32_t* count, float* seq_ps, int64_t B, int64_t N,
                                   int64_t L, int norm_factor, void* workspace, void* stream) {
    if (B == 0) return UCFVIT_OK;
    UCF_CHECK_ARG(domain && nodes && values && count && seq_ps && workspace, "ucfvit_octree_build: null pointer");
    UCF_CHECK_ARG(N > 0 && N <= 256, "ucfvit_octree_build: cubic volumes of side 1..256 (got %lld)", (long long)N);
    UCF_CHECK_ARG(L >= 1 && L % 7 == 1, "ucfvit_octree_build: fixed_length=%lld must be 7n+1 (every refinement adds seven nodes)", (long long)L);
    UCF_CHECK_ARG(norm_factor >= 1 && norm_factor <= 255, "ucfvit_octree_build: norm_factor=%d out of range", norm_factor);
    const size_t smem = 2 * (size_t)(L + 8) * sizeof(ONode);
    UCF_CHECK_ARG(smem <= 150 * 1024, "ucfvit_octree_build: fixed_length=%lld does not fit the LDS node lists", (long long)L);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, (size_t)ucfvit_octree_workspace(B, N), s) != hipSuccess) {
        ucfvit_set_error("ucfvit_octree_build: hipMemsetAsync failed");
        return UCFVIT_ERR_HIP;
    }
    const int64_t lines = B * N * N;
    const unsigned g = (unsigned)((lines + 255) / 256);
    hipLaunchKernelGGL(sat3_x_kernel, dim3(g), dim3(256), 0, s, domain, (unsigned*)workspace, (int)B, (int)N);
    hipLaunchKernelGGL(sat3_y_kernel, dim3(g), dim3(256), 0, s, (unsigned*)workspace, (int)B, (int)N);
    hipLaunchKernelGGL(sat3_z_kernel, dim3(g), dim3(256), 0, s, (unsigned*)workspace, (int)B, (int)N);
    UCF_LAUNCH_CHECK("ucfvit_octree_build(sat)");
    auto kern = octree_build_kernel;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            ucfvit_set_error("ucfvit_octree_build: cannot raise dynamic LDS to %zu bytes: %s", smem, hipGetErrorString(e));
            return UCFVIT_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(256), smem, s, nodes, values, count, seq_ps, (const unsigned*)workspace, (int)N, (int)L,
                       (unsigned)norm_factor);
    UCF_LAUNCH_CHECK("ucfvit_octree_build");
    return UCFVIT_OK;
}

extern "C" int ucfvit_octree_serialize(const float* img, const int32_t* nodes, const int32_t* count, float* seq, int64_t B, int64_t N, int64_t C,
                                       int64_t L, int64_t p, void* stream) {
    if (B == 0) return UCFVIT_OK;
    UCF_CHECK_ARG(img && nodes && count && seq, "ucfvit_octree_serialize: null pointer");
    UCF_CHECK_ARG(N > 0 && C > 0 && L > 0 && p > 0 && B * L < (1ll << 31), "ucfvit_octree_serialize: bad shape");
    hipLaunchKernelGGL(octree_serialize_kernel, dim3((unsigned)(B * L)), dim3(256), 0, (hipStream_t)stream, img, nodes, count, seq, (int)N, (int)C,
                       (int)L, (int)p);
    UCF_LAUNCH_CHECK("ucfvit_octree_serialize");
    return UCFVIT_OK;
}

extern "C" int64_t ucfvit_quadtree_workspace(int64_t B, int64_t H, int64_t W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return B * (H + 1) * (W + 1) * (int64_t)sizeof(unsigned);
}

extern "C" int ucfvit_quadtree_build(const uint8_t* edges, int32_t* nodes, int32_t* values, int32_t* count, float* seq_ps, int64_t B,
                                     int64_t H, int64_t W, int64_t L, void* workspace, void* stream) {
    if (B == 0) return UCFVIT_OK;
    UCF_CHECK_ARG(edges && nodes && values && count && seq_ps && workspace, "ucfvit_quadtree_build: null pointer");
    UCF_CHECK_ARG(H > 0 && W > 0 && H < 32768 && W < 32768 && H * W * 255 < (1ll << 32), "ucfvit_quadtree_build: image %lld x %lld out of range",
                  (long long)H, (long long)W);
    UCF_CHECK_ARG(L >= 1 && L % 3 == 1, "ucfvit_quadtree_build: fixed_length=%lld must be 3n+1 (every refinement adds three nodes)", (long long)L);
    const size_t smem = 2 * (size_t)(L + 4) * sizeof(QNode);
    UCF_CHECK_ARG(smem <= 150 * 1024, "ucfvit_quadtree_build: fixed_length=%lld does not fit the LDS node lists", (long long)L);
    auto kern = quadtree_build_kernel;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            ucfvit_set_error("ucfvit_quadtree_build: cannot raise dynamic LDS to %zu bytes: %s", smem, hipGetErrorString(e));
            return UCFVIT_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(256), smem, (hipStream_t)stream, edges, nodes, values, count, seq_ps, (unsigned*)workspace,
                       (int)H, (int)W, (int)L);
    UCF_LAUNCH_CHECK("ucfvit_quadtree_build");
    return UCFVIT_OK;
}

extern "C" int ucfvit_quadtree_serialize(const float* img, const int32_t* nodes, const int32_t* count, float* seq, int64_t B, int64_t H, int64_t W,
                                         int64_t C, int64_t L, int64_t p, void* stream) {
    if (B == 0) return UCFVIT_OK;
    UCF_CHECK_ARG(img && nodes && count && seq, "ucfvit_quadtree_serialize: null pointer");
    UCF_CHECK_ARG(H > 0 && W > 0 && C > 0 && L > 0 && p > 0 && B * L < (1ll << 31), "ucfvit_quadtree_serialize: bad shape");
    hipLaunchKernelGGL(quadtree_serialize_kernel, dim3((unsigned)(B * L)), dim3(256), 0, (hipStream_t)stream, img, nodes, count, seq, (int)H,
                       (int)W, (int)C, (int)L, (int)p);
    UCF_LAUNCH_CHECK("ucfvit_quadtree_serialize");
    return UCFVIT_OK;
}

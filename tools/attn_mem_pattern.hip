// Diagnostic (GPU box): what does the memory system deliver for the access pattern of the fused short-sequence attention backward, without
// any arithmetic?  One workgroup per (batch, head) reads the head's Q, K, V rows (128 B each at a row stride of 3 D elements) and dO, O rows
// (stride D) and writes dQ, dK, dV rows, exactly the bytes of csrc/attention_short.hip:attn_g_bwd_kernel at ViT-L B = 665.
//   hipcc -O3 --offload-arch=gfx950 tools/attn_mem_pattern.hip -o /tmp/attn_mem && /tmp/attn_mem
// Variants: workgroups resident per CU (LDS request), head-fastest or batch-fastest workgroup order, and the same bytes as one contiguous
// run per workgroup (what a streaming kernel gets).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: head-fastest (blockIdx % H), 1: batch-fastest, 2: contiguous run per workgroup
__global__ __launch_bounds__(256) void pattern_kernel(const char* __restrict__ qkv, const char* __restrict__ dout, const char* __restrict__ out,
                                                      char* __restrict__ dqkv, int B, int N, int H) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x;
    const long D2 = (long)H * 128;            // bytes of one token's heads (dh = 64 bf16)
    u32x4 acc = {0u, 0u, 0u, 0u};
    if (MODE == 2) {
        // the same byte count per workgroup, contiguous: 5 operand-rows-worth read, 3 written
        const long per = (long)N * 128;
        const char* r = qkv + (long)blockIdx.x * per * 3;
        const char* r2 = dout + (long)blockIdx.x * per;
        const char* r3 = out + (long)blockIdx.x * per;
        char* w = dqkv + (long)blockIdx.x * per * 3;
        for (long p = tid * 16; p < per * 3; p += 256 * 16) acc += *reinterpret_cast<const u32x4*>(r + p);
        for (long p = tid * 16; p < per; p += 256 * 16) acc += *reinterpret_cast<const u32x4*>(r2 + p);
        for (long p = tid * 16; p < per; p += 256 * 16) acc += *reinterpret_cast<const u32x4*>(r3 + p);
        for (long p = tid * 16; p < per * 3; p += 256 * 16) *reinterpret_cast<u32x4*>(w + p) = acc;
        return;
    }
    const int h = MODE == 0 ? blockIdx.x % H : blockIdx.x / B;
    const long b = MODE == 0 ? blockIdx.x / H : blockIdx.x % B;
    const char* qb = qkv + b * N * 3 * D2 + h * 128;
    const char* dob = dout + b * N * D2 + h * 128;
    const char* ob = out + b * N * D2 + h * 128;
    char* wq = dqkv + b * N * 3 * D2 + h * 128;
    for (int p = tid; p < N * 8; p += 256) {
        const int row = p >> 3, slot = p & 7;
        acc += *reinterpret_cast<const u32x4*>(qb + (long)row * 3 * D2 + slot * 16);
        acc += *reinterpret_cast<const u32x4*>(qb + D2 + (long)row * 3 * D2 + slot * 16);
        acc += *reinterpret_cast<const u32x4*>(qb + 2 * D2 + (long)row * 3 * D2 + slot * 16);
        acc += *reinterpret_cast<const u32x4*>(dob + (long)row * D2 + slot * 16);
        acc += *reinterpret_cast<const u32x4*>(ob + (long)row * D2 + slot * 16);
    }
    for (int p = tid; p < N * 8; p += 256) {
        const int row = p >> 3, slot = p & 7;
        *reinterpret_cast<u32x4*>(wq + (long)row * 3 * D2 + slot * 16) = acc;
        *reinterpret_cast<u32x4*>(wq + D2 + (long)row * 3 * D2 + slot * 16) = acc;
        *reinterpret_cast<u32x4*>(wq + 2 * D2 + (long)row * 3 * D2 + slot * 16) = acc;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE> float run(const char* qkv, const char* dout, const char* out, char* dqkv, int B, int N, int H, size_t smem) {
    auto k = pattern_kernel<MODE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(B * H), dim3(256), smem, 0, qkv, dout, out, dqkv, B, N, H);
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(B * H), dim3(256), smem, 0, qkv, dout, out, dqkv, B, N, H);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 665, N = 197, H = 16;
    const size_t tok = (size_t)B * N, D2 = (size_t)H * 128;
    char *qkv, *dout, *out, *dqkv;
    CK(hipMalloc(&qkv, tok * 3 * D2));
    CK(hipMalloc(&dout, tok * D2));
    CK(hipMalloc(&out, tok * D2));
    CK(hipMalloc(&dqkv, tok * 3 * D2));
    CK(hipMemset(qkv, 1, tok * 3 * D2));
    CK(hipMemset(dout, 1, tok * D2));
    CK(hipMemset(out, 1, tok * D2));
    const double gb = (double)tok * D2 * 8 / 1e9;
    printf("B = %d, N = %d, H = %d: %.2f GB moved per launch (5 operand reads + 3 gradient writes of %.0f MB)\n", B, N, H, gb, tok * D2 / 1e6);
    const size_t lds[] = {0, 16 * 1024, 40 * 1024, 58 * 1024, 96 * 1024};
    const char* occ[] = {"8 (no LDS)", "8 (16 KiB)", "4 (40 KiB)", "2 (58 KiB)", "1 (96 KiB)"};
    for (int i = 0; i < 5; ++i) {
        const float t0 = run<0>(qkv, dout, out, dqkv, B, N, H, lds[i]);
        const float t1 = run<1>(qkv, dout, out, dqkv, B, N, H, lds[i]);
        const float t2 = run<2>(qkv, dout, out, dqkv, B, N, H, lds[i]);
        printf("workgroups per CU %-11s  head-fastest %7.1f us %5.2f TB/s | batch-fastest %7.1f us %5.2f TB/s | contiguous %7.1f us %5.2f TB/s\n", occ[i], t0,
               gb / t0 * 1e3, t1, gb / t1 * 1e3, t2, gb / t2 * 1e3);
    }
    return 0;
}

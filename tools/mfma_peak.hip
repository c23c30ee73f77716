// Sustained MFMA rate of one MI355X under a pure v_mfma_f32_16x16x32_bf16 stream (no memory traffic): the clock / power limited
// ceiling that the GEMM kernels are measured against in DESIGN.md.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// the 32x32x16 shape: 32 cycles per instruction, 16 accumulator registers per tile
template <int NACC>
__global__ __launch_bounds__(256) void mfma32_loop(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x ^ e)); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x ^ e)); }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        for (int rep = 0; rep < 3; ++rep) {
            const int grid = 256 * wgs_per_cu;
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<8>, dim3(grid), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 4 * iters * 8 * (2.0 * 16 * 16 * 32);
            printf("waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flop / ms / 1e9);
        }
    }
    // shape / accumulator-count / occupancy matrix (what ONE wave per SIMD can issue matters for 4-wave GEMM designs)
    {
        struct V { const char* name; void (*k)(float*, int); double flop_per_iter_wave; };
        const V vs[] = {{"16x16x32 x4 acc", mfma_loop<4>, 4 * 2.0 * 16 * 16 * 32}, {"16x16x32 x16 acc", mfma_loop<16>, 16 * 2.0 * 16 * 16 * 32},
                        {"32x32x16 x4 acc", mfma32_loop<4>, 4 * 2.0 * 32 * 32 * 16}, {"32x32x16 x8 acc", mfma32_loop<8>, 8 * 2.0 * 32 * 32 * 16}};
        for (const V& v : vs)
            for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
                const int grid = 256 * wgs_per_cu;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, out, iters);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                printf("%-18s waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", v.name, wgs_per_cu, best, (double)grid * 4 * iters * v.flop_per_iter_wave / best / 1e9);
            }
    }
    // long run (~2 s) to see the sustained (power-limited) rate
    hipEventRecord(e0);
    for (int k = 0; k < 40; ++k) hipLaunchKernelGGL(mfma_loop<8>, dim3(512), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("sustained 40 launches: %.1f ms  %.1f TFLOP/s\n", ms, 40.0 * 512 * 4 * iters * 8 * (2.0 * 16 * 16 * 32) / ms / 1e9);
    return 0;
}

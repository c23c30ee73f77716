// Microbenchmark: how does the LDS-DMA (global_load_lds_dwordx4) delivery rate of ONE CU scale with the number of issuing waves and the
// pieces each has in flight?  Every workgroup (one per CU) re-reads its own 64-KiB-per-step window of a large bf16 matrix laid out like a GEMM
// operand (rows of `ld` bytes, 8 rows x 128 B per piece), walking along K like the GEMM K loop does.
//   hipcc -O3 --offload-arch=gfx950 tools/dma_rate.hip -o gpurun_out/dma_rate && gpurun_out/dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int P, int DEPTH>   // P pieces per wave per step; DEPTH = steps in flight (1: wait for this step's pieces; 2: wait for the previous step's)
__global__ void dma_kernel(const char* __restrict__ base, long ld, int steps, int rows_per_wg, unsigned long long* out) {
    extern __shared__ char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    // piece i of wave w: rows (w * P + i) * 8 .. + 8 of this workgroup's row window, 128 B at K offset k * 128
    unsigned off[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int row = ((wave * P + i) * 8 + (lane >> 3)) % rows_per_wg;
        off[i] = (unsigned)((long)row * ld + (lane & 7) * 16);
    }
    const char* wg = base + (long)blockIdx.x * rows_per_wg * ld;
    char* lds = smem + (wave * P) * 1024;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        const char* src = wg + (long)(s % (int)(ld / 128)) * 128;
#pragma unroll
        for (int i = 0; i < P; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(src + off[i]), (lptr_t)(lds + ((s & 1) * nw * P + i) * 1024), 16, 0, 0);
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int P, int DEPTH> void run(int W, const char* buf, long ld, int rows_per_wg, unsigned long long* dout, const char* tag) {
    const int steps = 2000;
    const size_t smem = (size_t)2 * W * P * 1024;
    if (smem > 160 * 1024) return;
    auto k = dma_kernel<P, DEPTH>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(W * 64), smem, 0, buf, ld, steps, rows_per_wg, dout);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= 256.0 * steps;
    printf("%s waves %2d pieces/wave %2d depth %d : %7.0f clk/step  %6.1f clk/piece/wave  %5.1f B/clk/CU  (%d KiB per step)\n", tag, W, P, DEPTH, avg, avg / P,
           W * P * 1024.0 / avg, W * P);
}

int main(int argc, char** argv) {
    const long ld = 2048;                       // K = 1024 bf16
    const int rows_per_wg = argc > 1 ? atoi(argv[1]) : 32;   // 32 rows x 2 KiB = 64 KiB per workgroup: 2 MiB per XCD, L2-resident
    printf("rows per workgroup window: %d (%ld KiB)\n", rows_per_wg, rows_per_wg * ld / 1024);
    char* buf;
    unsigned long long* dout;
    hipMalloc(&buf, (size_t)256 * rows_per_wg * ld);
    hipMemset(buf, 1, (size_t)256 * rows_per_wg * ld);
    hipMalloc(&dout, 256 * 8);
    for (int W : {4, 8, 12, 16}) {
        run<4, 1>(W, buf, ld, rows_per_wg, dout, "L2 window");
        run<8, 1>(W, buf, ld, rows_per_wg, dout, "L2 window");
        run<4, 2>(W, buf, ld, rows_per_wg, dout, "L2 window");
        run<8, 2>(W, buf, ld, rows_per_wg, dout, "L2 window");
    }
    run<16, 1>(4, buf, ld, rows_per_wg, dout, "L2 window");
    run<12, 1>(4, buf, ld, rows_per_wg, dout, "L2 window");
    run<2, 1>(16, buf, ld, rows_per_wg, dout, "L2 window");
    run<2, 2>(16, buf, ld, rows_per_wg, dout, "L2 window");
    return 0;
}

// ABI bookkeeping: version and the thread-local error message.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void ucfvit_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ucfvit_abi_version(void) { return UCFVIT_ABI_VERSION; }
extern "C" const char* ucfvit_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------------
// Diagnostic: a pure v_mfma_f32_16x16x32_bf16 stream (operands in registers, no memory traffic), 2 waves per SIMD on every CU,
// 16 independent accumulators per wave and 32 MFMAs per loop trip (with 8 accumulators and 8 MFMAs per trip the same loop reports
// only 1.25-1.3 PFLOP/s: the taken branch and the accumulate dependency show; tools/mfma_peak.hip has the whole matrix).
// Its rate is what the matrix pipes deliver at the clock the chip holds under a dense MFMA stream; bench.py reports it next to
// the nominal 2.5 PFLOP/s.
// ---------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void mfma_probe_kernel(float* __restrict__ sink, int iters) {
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (bf16)(0.37f * (float)((threadIdx.x * 7 + e * 3) % 13) - 2.f);
        b[e] = (bf16)(0.21f * (float)((threadIdx.x * 5 + e * 11) % 17) - 1.7f);
    }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[threadIdx.x] = s;     // keeps the loop alive; practically never true
}
}  // namespace

extern "C" int64_t ucfvit_mfma_probe(float* sink, int iters, void* stream) {
    if (!sink || iters <= 0) {
        ucfvit_set_error("ucfvit_mfma_probe: null sink or iters <= 0");
        return UCFVIT_ERR_INVALID_ARGUMENT;
    }
    const int grid = 512;                            // 2 workgroups of 4 waves per CU
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, sink, iters);
    UCF_LAUNCH_CHECK("ucfvit_mfma_probe");
    return (int64_t)grid * 4 * ((iters + 1) / 2 * 2) * 16 * (2ll * 16 * 16 * 32);
}

// ---------------------------------------------------------------------------------------------------
// Diagnostic: hold `workgroups` CUs for about `microseconds` (one 512-thread workgroup with 96 KiB of LDS each: no GEMM workgroup fits
// beside it) — what an RCCL collective does to the persistent GEMM grids when it overlaps backward.  On a one-GPU box this is how the
// dynamic tile schedule (desc->sched_state) is measured against the static one (tools/gemm_contention.py).  The wait is bounded.
// ---------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(512) void occupy_kernel(unsigned long long ticks, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char hold[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
    unsigned spins = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && spins < (1u << 26)) {
        __builtin_amdgcn_s_sleep(32);
        ++spins;
    }
    if (spins == 0xFFFFFFFFu) sink[threadIdx.x] = hold[threadIdx.x];      // never true: keeps the LDS allocation referenced
}
}  // namespace

extern "C" int ucfvit_occupy(int workgroups, int microseconds, float* sink, void* stream) {
    UCF_CHECK_ARG(workgroups >= 1 && workgroups <= 256 && microseconds >= 1 && microseconds <= 1000000 && sink,
                  "ucfvit_occupy: workgroups in 1..256, microseconds in 1..1e6, sink required");
    static bool done = false;
    constexpr int lds = 96 * 1024;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            ucfvit_set_error("ucfvit_occupy: cannot raise dynamic LDS: %s", hipGetErrorString(e));
            return UCFVIT_ERR_HIP;
        }
        done = true;
    }
    hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(512), lds, (hipStream_t)stream, (unsigned long long)microseconds * 100ull, sink);
    UCF_LAUNCH_CHECK("ucfvit_occupy");
    return UCFVIT_OK;
}

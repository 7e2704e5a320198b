// zc_ubench.hip -- libzc_ubench.so: the saturated v_mad_u64_u32 rate of THIS board, measured in the
// same process as a benchmark run (bench.py calls it after its timed region).  Measurement aid only:
// not part of the product library, not part of the ABI in include/zerocaf_hip.h.
//
// Eight workgroups of 256 threads per CU (= eight waves per SIMD, the wave-slot limit, so the
// dispatcher has no placement freedom), every lane runs 8 independent multiply-accumulate chains.
// Returns the whole-chip rate in 10^12 lane-operations per second from the HIP-event time of one
// launch (after a warm-up launch of the same length); *clock_ghz receives the effective shader
// clock (s_memtime / s_memrealtime of the median wave).  tools/ubench/occupancy.hip is the long form.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <vector>
typedef uint32_t u32;
typedef uint64_t u64;

__global__ __launch_bounds__(256) void k_mad_chains(u32* out, int iters, u32 seed, u64* stamps)
{
    u32 a = seed ^ (threadIdx.x * 2654435761u), b = (seed >> 3) | 1u;
    u64 acc[8];
    for (int c = 0; c < 8; c++) acc[c] = c;
    const u64 c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int c = 0; c < 8; c++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
    }
    const u64 c1 = clock64(), w1 = wall_clock64();
    u32 s = 0;
    for (int c = 0; c < 8; c++) s += (u32)acc[c] + (u32)(acc[c] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * wv] = c1 - c0;
        stamps[2 * wv + 1] = w1 - w0;
    }
}

extern "C" double zc_ubench_mad_u64_u32(double target_ms, double* clock_ghz)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1.0;
    const int grid = cus * 8;
    const size_t waves = (size_t)grid * 4;
    u32* out = nullptr;
    u64* stamps = nullptr;
    if (hipMalloc(&out, (size_t)grid * 256 * 4) != hipSuccess || hipMalloc(&stamps, waves * 16) != hipSuccess) return -1.0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto timed = [&](int iters) {
        float ms = 0.f;
        (void)hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(k_mad_chains, dim3(grid), dim3(256), 0, nullptr, out, iters, 12345u, stamps);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        return (double)ms;
    };
    int iters = 1024;
    (void)timed(iters);
    double ms = timed(iters);
    iters = (int)std::min(1.0e8, std::max(1024.0, iters * target_ms / std::max(ms, 1e-3)));
    (void)timed(iters);                                            // warm-up at full length
    ms = timed(iters);
    std::vector<u64> h(2 * waves);
    (void)hipMemcpy(h.data(), stamps, waves * 16, hipMemcpyDeviceToHost);
    std::vector<double> ghz(waves);
    for (size_t w = 0; w < waves; w++) ghz[w] = (double)h[2 * w] / ((double)h[2 * w + 1] * 10.0);
    std::nth_element(ghz.begin(), ghz.begin() + waves / 2, ghz.end());
    if (clock_ghz) *clock_ghz = ghz[waves / 2];
    (void)hipFree(out);
    (void)hipFree(stamps);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return (double)waves * iters * 16.0 * 64.0 / (ms * 1e-3) / 1e12;
}

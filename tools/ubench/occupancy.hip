// occupancy.hip -- issue rate of the multiplier-class VALU instructions on gfx950 with the number of
// resident waves per SIMD PINNED: grid = CUs x k workgroups of 256 threads, and each workgroup asks
// for floor(160 KB / k) of dynamic LDS (rounded so that k fit on a CU and k + 1 do not), so the
// dispatcher has to place exactly k workgroups on every CU (the plain cus x k grid of ubench.hip
// leaves the placement to the dispatcher: at k = 3 some CUs get 5 workgroups and the launch takes
// 1.7x one wave's own run time).  Every wave stamps s_memtime (shader cycles) and s_memrealtime
// (100 MHz) around its loop.  Printed per (instruction mix, k):
//   cyc/slot   shader cycles per wave-instruction per SIMD = wave cycles / (instructions x k), median wave
//   spread     slowest wave / median wave (older waves win the issue arbitration)
//   rounds     launch time / slowest wave's own run time (1.0 = all k workgroups per CU resident at once)
//   GHz        effective shader clock of the median wave during the loop
//   T lane/s   whole-chip lane-operations per second from the HIP-event time of the launch
// Mixes: "A+nB" = one A followed by n B, independent chains, to see whether 32-bit ALU ops hide
// behind the multiplier stream (co-issue from other waves) or take their own issue slots.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64;
#define CH 8

template <int MIX> __device__ __forceinline__ void body(u64 (&acc)[CH], u32 (&x)[CH], u32 a, u32 b)
{
#pragma unroll
    for (int c = 0; c < CH; c++) {
        if (MIX == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
        if (MIX == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[c]) : "v"(a));
        if (MIX == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
        if (MIX == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
        if (MIX == 4) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[c]));
        if (MIX == 5) {   // 4 mad + 1 add  (the strict loop's ratio: ~0.27 other VALU per multiplier-class op)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
            if ((c & 3) == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
        }
        if (MIX == 6) {   // 1 mad + 1 add
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
        }
        if (MIX == 7) {   // 1 mad + 2 add
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
            asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[(c + 1) % CH]) : "v"(a));
        }
        if (MIX == 8) {   // dependent chain of mads on ONE accumulator (the column-ordered multiplier's shape)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
        }
        if (MIX == 9) {   // TWO dependent chains, interleaved (two column-ordered multiplications side by side)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c & 1]) : "v"(a), "v"(b) : "vcc");
        }
        if (MIX == 10) {  // one dependent chain with the hazard s_nop hipcc puts between dependent mads
            asm volatile("s_nop 0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
        }
    }
}
template <int MIX> __global__ __launch_bounds__(256) void k(u32* out, int iters, u32 seed, u64* stamps)
{
    extern __shared__ u32 lds[];
    u32 a = seed ^ (threadIdx.x * 2654435761u), b = (seed >> 3) | 1u;
    u64 acc[CH]; u32 x[CH];
    for (int c = 0; c < CH; c++) { acc[c] = c; x[c] = c + a; }
    if (seed == 0xFFFFFFFFu) lds[threadIdx.x] = a;          // keeps the allocation
    const u64 c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) { body<MIX>(acc, x, a, b); body<MIX>(acc, x, a, b); }
    u32 s = 0;
    for (int c = 0; c < CH; c++) s += (u32)acc[c] + (u32)(acc[c] >> 32) + x[c];
    const u64 c1 = clock64(), w1 = wall_clock64();
    if (seed == 0xFFFFFFFFu) s += lds[(threadIdx.x + 1) & 255];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * wv] = c1 - c0; stamps[2 * wv + 1] = w1 - w0;
    }
}
struct Ent { const char* name; void (*fn)(u32*, int, u32, u64*); double slots; double mads; };
int main(int argc, char** argv)
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double target_ms = argc > 1 ? atof(argv[1]) : 20.0;
    const size_t lds_cu = 160 * 1024;
    printf("device %s, %d CUs, max clock %d MHz, LDS/CU %zu KB; %g ms launches after a warm-up launch of the same length\n",
           prop.name, cus, prop.clockRate / 1000, lds_cu / 1024, target_ms);
    std::vector<Ent> ents = {
        {"v_mad_u64_u32", k<0>, 1, 1}, {"v_fma_f32", k<1>, 1, 0}, {"v_add_u32", k<2>, 1, 0}, {"v_mul_lo_u32", k<3>, 1, 0},
        {"v_lshrrev_b64", k<4>, 1, 0}, {"4 mad + 1 add", k<5>, 1.25, 1}, {"1 mad + 1 add", k<6>, 2, 1}, {"1 mad + 2 alu", k<7>, 3, 1},
        {"mad, one dependent chain", k<8>, 1, 1}, {"mad, two chains interleaved", k<9>, 1, 1}, {"s_nop + dependent mad", k<10>, 1, 1}};
    const int max_waves = cus * 8 * 4;
    u32* out; hipMalloc(&out, (size_t)max_waves * 64 * 4);
    u64* stamps; hipMalloc(&stamps, (size_t)max_waves * 16);
    std::vector<u64> h(2 * (size_t)max_waves);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-28s %6s %8s %9s %7s %7s %6s %9s %12s\n", "mix", "w/SIMD", "ms", "cyc/slot", "spread", "rounds", "GHz", "cyc/mad", "T lane/s");
    for (auto& e : ents) {
        hipFuncSetAttribute((const void*)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cu);
        for (int kk : {1, 2, 3, 4, 5, 6, 8}) {
            // k workgroups fit with room to spare for the allocation granularity, k + 1 do not
            static const int kb[9] = {0, 96, 64, 48, 36, 30, 25, 0, 19};
            const size_t bytes = (size_t)kb[kk] * 1024;
            const int grid = cus * kk;
            int iters = 1024;
            auto launch = [&](int it) { hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), bytes, 0, out, it, 12345u, stamps); };
            launch(iters); hipDeviceSynchronize();
            hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (hipGetLastError() != hipSuccess) { printf("%-26s %6d launch failed (LDS %zu)\n", e.name, kk, bytes); continue; }
            iters = (int)std::min(1.0e8, std::max(1024.0, iters * target_ms / ms));
            launch(iters); hipDeviceSynchronize();
            hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            const size_t waves = (size_t)grid * 4;
            hipMemcpy(h.data(), stamps, waves * 16, hipMemcpyDeviceToHost);
            std::vector<double> cyc(waves), ghz(waves);
            for (size_t w = 0; w < waves; w++) { cyc[w] = (double)h[2 * w]; ghz[w] = (double)h[2 * w] / ((double)h[2 * w + 1] * 10.0); }
            std::sort(cyc.begin(), cyc.end());
            std::nth_element(ghz.begin(), ghz.begin() + waves / 2, ghz.end());
            const double groups = (double)iters * 2 * CH;                    // per wave: groups of `slots` instructions
            const double med = cyc[waves / 2];
            const double cyc_slot = med / (groups * e.slots * kk);
            const double lane = e.mads > 0 ? (double)waves * groups * e.mads * 64 / (ms * 1e-3) / 1e12
                                           : (double)waves * groups * e.slots * 64 / (ms * 1e-3) / 1e12;
            // launch time over the slowest wave's own run time: 1.0 = every workgroup was resident from the start
            const double rounds = ms * 1e-3 / (cyc[waves - 1] / (ghz[waves / 2] * 1e9));
            printf("%-28s %6d %8.3f %9.3f %7.3f %7.2f %6.3f %9.3f %12.2f\n", e.name, kk, ms, cyc_slot, cyc[waves - 1] / med, rounds, ghz[waves / 2],
                   e.mads > 0 ? med / (groups * e.mads * kk) : 0.0, lane);
        }
    }
    return 0;
}

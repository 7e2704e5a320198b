// ubench.hip -- instruction-rate microbenchmarks for gfx950 integer/FP64 VALU ops that
// bound the limb arithmetic ("measure v_mad_u64_u32 issue rate first").
// Each kernel runs ITERS x CHAINS x UNROLL independent-chain instructions per lane.  Every wave
// brackets its loop with s_memtime (shader cycles, clock64()) and s_memrealtime (100 MHz wall
// clock, wall_clock64()), so the cycles per wave-instruction per SIMD are read in SHADER CYCLES
// -- independent of the clock the board settles at under load -- and the effective shader clock
// (cycles / wall time) is reported next to them.  The host's HIP-event time gives the whole-chip
// rate (lane-ops/s) that bench.py prices kernels against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
typedef uint32_t u32; typedef uint64_t u64;
#ifndef CHAINS
#define CHAINS 8
#endif
#ifndef UNROLL
#define UNROLL 2
#endif

#define DEF_KERNEL(NAME, DECL, BODY, SINK)                                         \
  extern "C" __global__ __launch_bounds__(256) void NAME(u32* out, int iters, u32 seed, u64* stamps) { \
    u32 a = seed ^ (threadIdx.x * 2654435761u), b = (seed >> 3) | 1u;              \
    DECL                                                                           \
    const u64 c0 = clock64(), w0 = wall_clock64();                                 \
    for (int it = 0; it < iters; it++) {                                           \
      _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                         \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { BODY }                \
      }                                                                            \
    }                                                                              \
    u32 s = 0;                                                                     \
    _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { SINK }                    \
    const u64 c1 = clock64(), w1 = wall_clock64();                                 \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                       \
    if ((threadIdx.x & 63) == 0) {                                                 \
      const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);               \
      stamps[2 * wv] = c1 - c0; stamps[2 * wv + 1] = w1 - w0;                      \
    }                                                                              \
  }

DEF_KERNEL(k_mad_u64_u32, u64 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c;,
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");,
  s += (u32)acc[c] + (u32)(acc[c]>>32);)
DEF_KERNEL(k_mul_lo_u32, u32 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c+a;,
  asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(b));, s += acc[c];)
DEF_KERNEL(k_mul_hi_u32, u32 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c+a;,
  asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(b));, s += acc[c];)
DEF_KERNEL(k_mad_u32_u24, u32 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c+a;,
  asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));, s += acc[c];)
DEF_KERNEL(k_mul_hi_u32_u24, u32 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c+a;,
  asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(acc[c]) : "v"(b));, s += acc[c];)
DEF_KERNEL(k_add_u32, u32 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c+a;,
  asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[c]) : "v"(b));, s += acc[c];)
DEF_KERNEL(k_and_b32, u32 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c+a;,
  asm volatile("v_and_b32 %0, %0, %1" : "+v"(acc[c]) : "v"(b));, s += acc[c];)
DEF_KERNEL(k_lshl_add_u64, u64 acc[CHAINS]; u64 bb = ((u64)a<<32)|b; for (int c=0;c<CHAINS;c++) acc[c]=c;,
  asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[c]) : "v"(bb));, s += (u32)acc[c] + (u32)(acc[c]>>32);)
DEF_KERNEL(k_lshrrev_b64, u64 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=((u64)a<<32)|(b+c);,
  asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[c]));, s += (u32)acc[c] + (u32)(acc[c]>>32);)
DEF_KERNEL(k_lshlrev_b64, u64 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=((u64)a<<32)|(b+c);,
  asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(acc[c]));, s += (u32)acc[c] + (u32)(acc[c]>>32);)
DEF_KERNEL(k_addc_pair, u32 lo[CHAINS]; u32 hi[CHAINS]; for (int c=0;c<CHAINS;c++) {lo[c]=c; hi[c]=0;},
  asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo[c]), "+v"(hi[c]) : "v"(a), "v"(b) : "vcc");,
  s += lo[c] + hi[c];)
DEF_KERNEL(k_fma_f64, double acc[CHAINS]; double da = 1.0 + a*1e-12; double db = b*1e-15; for (int c=0;c<CHAINS;c++) acc[c]=c;,
  asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(da), "v"(db));, s += (u32)(long long)acc[c];)
DEF_KERNEL(k_add_f64, double acc[CHAINS]; double db = b*1e-15; for (int c=0;c<CHAINS;c++) acc[c]=c;,
  asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(db));, s += (u32)(long long)acc[c];)
DEF_KERNEL(k_fma_f32, float acc[CHAINS]; float fa = 1.0f + a*1e-12f; float fb = b*1e-15f; for (int c=0;c<CHAINS;c++) acc[c]=c;,
  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(fa), "v"(fb));, s += (u32)acc[c];)
DEF_KERNEL(k_cndmask, u32 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c+a;,
  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[c]) : "v"(b) : );, s += acc[c];)
DEF_KERNEL(k_dot2_u32_u16, u32 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c;,
  asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));, s += acc[c];)
DEF_KERNEL(k_dot4_u32_u8, u32 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c;,
  asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));, s += acc[c];)
DEF_KERNEL(k_mad_u64_u32_addc, u64 acc[CHAINS]; u32 hi[CHAINS]; for (int c=0;c<CHAINS;c++) {acc[c]=c; hi[c]=0;},
  asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[c]), "+v"(hi[c]) : "v"(a), "v"(b) : "vcc");,
  s += (u32)acc[c] + (u32)(acc[c]>>32) + hi[c];)
DEF_KERNEL(k_mad_i64_i32, u64 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c;,
  asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");,
  s += (u32)acc[c] + (u32)(acc[c]>>32);)
DEF_KERNEL(k_mul_f64, double acc[CHAINS]; double da = 1.0 + a*1e-12; for (int c=0;c<CHAINS;c++) acc[c]=c+1;,
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(da));, s += (u32)(long long)acc[c];)
DEF_KERNEL(k_pk_mul_lo_u16, u32 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c+a;,
  asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(acc[c]) : "v"(b));, s += acc[c];)
DEF_KERNEL(k_pk_mad_u16, u32 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c+a;,
  asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));, s += acc[c];)

DEF_KERNEL(k_cndmask_sgpr, u32 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c+a; unsigned long long m = __ballot((a & 1) != 0);,
  asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(b), "s"(m));, s += acc[c];)
DEF_KERNEL(k_bfi_b32, u32 acc[CHAINS]; u32 msk = (a & 1) ? 0xffffffffu : 0u; for (int c=0;c<CHAINS;c++) acc[c]=c+a;,
  asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(msk), "v"(b));, s += acc[c];)
DEF_KERNEL(k_cndmask_vcc_set, u32 acc[CHAINS]; for (int c=0;c<CHAINS;c++) acc[c]=c+a; asm volatile("v_cmp_eq_u32 vcc, 1, %0" :: "v"(a & 1) : "vcc");,
  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[c]) : "v"(b) : );, s += acc[c];)
typedef void (*kfn)(u32*, int, u32, u64*);
struct Ent { const char* name; kfn fn; };

#include <algorithm>
int main(int argc, char** argv) {
  int dev = 0; hipSetDevice(dev);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  const double target_ms = argc > 1 ? atof(argv[1]) : 25.0;     // per measurement: long enough for the board to settle
  printf("device %s CUs %d max clock %.0f MHz; each line: one launch of ~%.0f ms after a warm-up launch of the same length\n", prop.name, cus, clk/1e6, target_ms);
  std::vector<Ent> ents = {
    {"v_fma_f32", k_fma_f32}, {"v_add_u32", k_add_u32}, {"v_and_b32", k_and_b32},
    {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mad_i64_i32", k_mad_i64_i32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32},
    {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24},
    {"v_cndmask_b32 (sgpr mask)", k_cndmask_sgpr}, {"v_cndmask_b32 (vcc set once)", k_cndmask_vcc_set}, {"v_bfi_b32", k_bfi_b32},
    {"v_lshl_add_u64", k_lshl_add_u64}, {"v_lshrrev_b64", k_lshrrev_b64}, {"v_lshlrev_b64", k_lshlrev_b64},
    {"v_add_co+v_addc (pair)", k_addc_pair}, {"v_mad_u64_u32+v_addc (pair)", k_mad_u64_u32_addc},
    {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64},
    {"v_dot2_u32_u16", k_dot2_u32_u16}, {"v_dot4_u32_u8", k_dot4_u32_u8},
    {"v_pk_mul_lo_u16", k_pk_mul_lo_u16}, {"v_pk_mad_u16", k_pk_mad_u16},
  };
  const int max_waves = cus * 8 * 4;
  u32* out; hipMalloc(&out, (size_t)max_waves * 64 * 4);
  u64* stamps; hipMalloc(&stamps, (size_t)max_waves * 16);
  std::vector<u64> h(2 * (size_t)max_waves);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-30s %5s %9s %12s %10s %11s %14s\n", "instruction", "w/SIMD", "ms", "cyc/inst", "clock GHz", "cyc@events", "T lane-ops/s");
  for (int bpc : {1, 2, 3, 8}) {                                 // workgroups per CU = waves per SIMD
    for (auto& e : ents) {
      int grid = cus * bpc;
      // calibrate the trip count on a short launch
      int iters = 2048;
      hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, iters, 12345u, stamps);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, iters, 12345u, stamps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      iters = (int)std::min(2.0e8, std::max(2048.0, iters * target_ms / ms));
      hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, iters, 12345u, stamps);   // warm-up at full length
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, iters, 12345u, stamps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      const size_t waves = (size_t)grid * 4;
      hipMemcpy(h.data(), stamps, waves * 16, hipMemcpyDeviceToHost);
      std::vector<double> cyc(waves), ghz(waves);
      for (size_t w = 0; w < waves; w++) { cyc[w] = (double)h[2 * w]; ghz[w] = (double)h[2 * w] / ((double)h[2 * w + 1] * 10.0); }
      std::nth_element(cyc.begin(), cyc.begin() + waves / 2, cyc.end());
      std::nth_element(ghz.begin(), ghz.begin() + waves / 2, ghz.end());
      const double n_inst = (double)iters * CHAINS * UNROLL;     // per wave
      const double cyc_per_inst = cyc[waves / 2] / (n_inst * bpc);   // bpc waves share a SIMD
      const double rate = (double)waves * n_inst / (ms * 1e-3);  // wave-instructions per second, whole chip
      printf("%-30s %5d %9.3f %12.3f %10.3f %11.3f %14.2f\n", e.name, bpc, ms, cyc_per_inst, ghz[waves / 2],
             (double)cus * 4 * ghz[waves / 2] * 1e9 / rate, rate * 64 / 1e12);
    }
  }
  return 0;
}

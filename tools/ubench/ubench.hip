// ubench.hip -- instruction-rate microbenchmarks for gfx950 integer/FP64 VALU ops that
// bound the limb arithmetic ("measure v_mad_u64_u32 issue rate first").
// Each kernel runs ITERS x 16 independent-chain instructions per lane; the host reports
// wave-instructions per second and the implied cycles per wave-instruction per SIMD.
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
  extern "C" __global__ __launch_bounds__(256) void NAME(u32* out, int iters, u32 seed) { \
    u32 a = seed ^ (threadIdx.x * 2654435761u), b = (seed >> 3) | 1u;              \
    DECL                                                                           \
    for (int it = 0; it < iters; it++) {                                           \
      _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                         \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { BODY }                \
      }                                                                            \
    }                                                                              \
    u32 s = 0;                                                                     \
    _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { SINK }                    \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                       \
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
typedef void (*kfn)(u32*, int, u32);
struct Ent { const char* name; kfn fn; int insts_per_body; };

int main(int argc, char** argv) {
  int dev = 0; hipSetDevice(dev);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk/1e6);
  std::vector<Ent> ents = {
    {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mad_i64_i32", k_mad_i64_i32, 1}, {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
    {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1},
    {"v_add_u32", k_add_u32, 1}, {"v_and_b32", k_and_b32, 1}, {"v_cndmask_b32", k_cndmask, 1}, {"v_cndmask_b32 (sgpr mask)", k_cndmask_sgpr, 1},
    {"v_cndmask_b32 (vcc set once)", k_cndmask_vcc_set, 1}, {"v_bfi_b32", k_bfi_b32, 1},
    {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1}, {"v_lshlrev_b64", k_lshlrev_b64, 1},
    {"v_add_co+v_addc (pair)", k_addc_pair, 1}, {"v_mad_u64_u32+v_addc (pair)", k_mad_u64_u32_addc, 1},
    {"v_fma_f64", k_fma_f64, 1}, {"v_add_f64", k_add_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_fma_f32", k_fma_f32, 1},
    {"v_dot2_u32_u16", k_dot2_u32_u16, 1}, {"v_dot4_u32_u8", k_dot4_u32_u8, 1},
    {"v_pk_mul_lo_u16", k_pk_mul_lo_u16, 1}, {"v_pk_mad_u16", k_pk_mad_u16, 1},
  };
  const int iters = 4096;
  u32* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpb : {1, 2}) {   // blocks per CU multiplier: occupancy 4 or 8 waves/CU... use grid = cus*bpc
    for (auto& e : ents) {
      for (int bpc : {2, 8}) {
        if (wpb == 2 && bpc == 2) continue;
        if (wpb == 1 && bpc == 8) continue;
        int grid = cus * bpc;
        hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, 16, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, iters, 12345u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double waves = (double)grid * 4;
        double winsts = waves * (double)iters * CHAINS * UNROLL;
        double rate = winsts / (ms * 1e-3);
        double simd_cycles = (double)cus * 4 * clk;   // SIMD-cycles per second at max clock
        printf("%-30s waves/SIMD %d  %8.3f ms  %9.2f G wave-inst/s  %6.2f cyc/wave-inst/SIMD (at max clk)  %8.2f T lane-ops/s\n",
               e.name, bpc, ms, rate/1e9, simd_cycles / rate, rate*64/1e12);
      }
    }
  }
  return 0;
}

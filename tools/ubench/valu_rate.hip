// Micro-benchmark: issue rate of the integer-multiply family on gfx950 (decides the limb radix).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHAINS 8
#define ITERS 4096

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t acc[CHAINS];
  uint32_t r[CHAINS];
  double d[CHAINS];
  for (int c = 0; c < CHAINS; ++c) { acc[c] = a + c; r[c] = b + c; d[c] = (double)(a + c); }
  double da = (double)a * 1e-9, db = (double)b * 1e-9;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
      if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[c]) : "v"(a));
      if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[c]) : "v"(a));
      if (OP == 3) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[c]) : "v"(da), "v"(db));
      if (OP == 4) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(r[c]) : "v"(a), "v"(b));
      if (OP == 5) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r[c]) : "v"(a));
      if (OP == 6) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(r[c]) : "v"(a), "v"(b));
      if (OP == 7) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[c]) : "v"(a) : "vcc");
      if (OP == 8) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(acc[c]) : "v"(acc[(c + 1) % CHAINS]));
      if (OP == 9) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[c]) : "v"(a), "v"(b));
      if (OP == 10) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r[c]) : "v"(a), "v"(b));
      if (OP == 11) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "s10", "s11");
      if (OP == 12) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[c]) : "v"(a), "v"(b));
      if (OP == 13) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(r[c]) : "v"(a), "v"(b));
      if (OP == 14) asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(r[c]) : "v"(a));
      if (OP == 15) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r[c]) : "v"(a) : "vcc");
    }
  }
  uint32_t s = 0;
  for (int c = 0; c < CHAINS; ++c) s += (uint32_t)acc[c] + r[c] + (uint32_t)d[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block per CU
  uint32_t* out;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(out, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int rep = 0; rep < 5; ++rep) k<OP><<<blocks, 256>>>(out, rep);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double instr_per_simd = 5.0 * waves_per_simd * (double)ITERS * CHAINS;  // wave-instructions per SIMD
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles (at 2.4GHz) per wave64-instr per SIMD\n", name,
         waves_per_simd, ms, cyc / instr_per_simd);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_mad_u64_u32 (vcc)", w);
    run<11>("v_mad_u64_u32 (sgpr carry)", w);
    run<1>("v_mul_lo_u32", w);
    run<2>("v_mul_hi_u32", w);
    run<3>("v_fma_f64", w);
    run<12>("v_fma_f32", w);
    run<4>("v_mad_u32_u24", w);
    run<13>("v_mad_i32_i24", w);
    run<5>("v_mul_hi_u32_u24", w);
    run<6>("v_dot2_u32_u16", w);
    run<10>("v_dot4_u32_u8", w);
    run<7>("v_add_co_u32", w);
    run<15>("v_addc_co_u32", w);
    run<8>("v_lshl_add_u64", w);
    run<9>("v_add3_u32", w);
    run<14>("v_alignbit_b32", w);
  }
  return 0;
}

// Does the SGPR carry-out of v_mad_u64_u32 serialise a lone wave?  Same carry pair for every mad
// vs rotating over 4 pairs vs VCC.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t acc[8];
  for (int c = 0; c < 8; ++c) acc[c] = a + c;
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "s10", "s11");
    } else if (MODE == 1) {
      asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "s10", "s11");
      asm volatile("v_mad_u64_u32 %0, s[12:13], %1, %2, %0" : "+v"(acc[1]) : "v"(a), "v"(b) : "s12", "s13");
      asm volatile("v_mad_u64_u32 %0, s[14:15], %1, %2, %0" : "+v"(acc[2]) : "v"(a), "v"(b) : "s14", "s15");
      asm volatile("v_mad_u64_u32 %0, s[16:17], %1, %2, %0" : "+v"(acc[3]) : "v"(a), "v"(b) : "s16", "s17");
      asm volatile("v_mad_u64_u32 %0, s[18:19], %1, %2, %0" : "+v"(acc[4]) : "v"(a), "v"(b) : "s18", "s19");
      asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[5]) : "v"(a), "v"(b) : "s20", "s21");
      asm volatile("v_mad_u64_u32 %0, s[22:23], %1, %2, %0" : "+v"(acc[6]) : "v"(a), "v"(b) : "s22", "s23");
      asm volatile("v_mad_u64_u32 %0, s[24:25], %1, %2, %0" : "+v"(acc[7]) : "v"(a), "v"(b) : "s24", "s25");
    } else if (MODE == 2) {
#pragma unroll
      for (int c = 0; c < 8; ++c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
    } else if (MODE == 3) {  // 32-bit pieces: lo/hi products (no SGPR write)
      uint32_t* r = reinterpret_cast<uint32_t*>(acc);
#pragma unroll
      for (int c = 0; c < 8; ++c) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[c]) : "v"(a));
    } else if (MODE == 4) {  // dependent chain on ONE accumulator
#pragma unroll
      for (int c = 0; c < 8; ++c) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "s10", "s11");
    } else if (MODE == 5) {
      uint32_t* r = reinterpret_cast<uint32_t*>(acc);
#pragma unroll
      for (int c = 0; c < 8; ++c) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[c]) : "v"(acc[(c + 1) & 7]));
    }
  }
  uint32_t s = 0;
  for (int c = 0; c < 8; ++c) s += (uint32_t)acc[c] + (uint32_t)(acc[c] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int wps) {
  int blocks = 256 * wps; uint32_t* out; (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 1); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int r = 0; r < 5; ++r) k<MODE><<<blocks, 256>>>(out, r); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s waves/SIMD=%d  %.2f cycles@2.4GHz per wave-instr per SIMD\n", name, wps, ms * 1e-3 * 2.4e9 / (5.0 * wps * ITERS * 8));
  (void)hipFree(out);
}
int main() {
  for (int w : {1, 2}) {
    run<0>("mad_u64_u32, same carry SGPR pair", w);
    run<1>("mad_u64_u32, 8 rotating carry SGPR pairs", w);
    run<2>("mad_u64_u32, carry -> vcc", w);
    run<4>("mad_u64_u32, dependent chain (1 acc)", w);
    run<3>("v_mul_hi_u32", w);
    run<5>("v_lshl_add_u64", w);
  }
}

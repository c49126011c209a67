// Micro-benchmark: does a divsteps inversion cost the same when only one lane in eight is active
// (the fused accumulate + invert kernel) as when every lane is (the finish kernel)?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../stark-perpetual_amd/csrc inv_lanes.hip -o inv_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fp29.hpp"
using namespace sp;

template <int STRIDE, int PAD_MULS>
__global__ void __launch_bounds__(256) k(const int32_t* in, int32_t* out, int reps) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (PAD_MULS == -2 && (threadIdx.x % 64) >= (unsigned)(-STRIDE)) return;
  fe a;
  for (int i = 0; i < NL; ++i) a.l[i] = in[i] ^ (int32_t)((t * 2654435761u) & 0xfffff);
  fe acc = a;
  // optional mad-heavy prologue executed by every lane (stands in for the accumulate phase)
  for (int i = 0; i < (PAD_MULS > 0 ? PAD_MULS : 0); ++i) acc = fe_mul(acc, a);
  const bool mine = STRIDE > 0 ? (t % (STRIDE > 0 ? STRIDE : 1)) == 0 : (threadIdx.x % 64) < (unsigned)(-STRIDE);
  if (mine) {
    if (PAD_MULS >= 0) {
      for (int r = 0; r < reps; ++r) acc = fe_inv(fe_carry(fe_add(acc, a)));
    } else {  // same protocol with a chain of 110 multiplications (about one inversion's worth of instructions)
      for (int r = 0; r < reps; ++r)
        for (int i = 0; i < 110; ++i) acc = fe_mul(acc, a);
    }
  }
  for (int i = 0; i < NL; ++i) out[t * NL + i] = acc.l[i];
}

template <int STRIDE, int PAD>
void run(const char* name, int threads, int reps) {
  int32_t *in, *out;
  hipMalloc(&in, 64);
  hipMemset(in, 1, 64);
  hipMalloc(&out, (size_t)threads * NL * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<STRIDE, PAD><<<threads / 256, 256>>>(in, out, reps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<STRIDE, PAD><<<threads / 256, 256>>>(in, out, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %7d threads: %.1f us per inversion\n", name, threads, ms * 1e3 / reps);
  hipFree(in); hipFree(out);
}

int main() {
  const int reps = 8;
  run<1, 0>("all lanes active, 64 waves", 4096, reps);
  run<8, 0>("1 lane in 8 active, 512 waves", 32768, reps);
  run<1, 0>("all lanes active, 512 waves", 32768, reps);
  run<8, 64>("1 in 8 active after 64 multiplications, 512 waves", 32768, reps);
  run<1, 64>("all lanes active after 64 multiplications, 64 waves", 4096, reps);
  run<8, 0>("1 lane in 8 active, 64 waves", 4096, reps);
  run<-8, 0>("lanes 0..7 of every wave active, 512 waves", 32768, reps);
  run<-32, 0>("lanes 0..31 of every wave active, 512 waves", 32768, reps);
  run<2, 0>("every second lane active, 512 waves", 32768, reps);
  run<4, 0>("every fourth lane active, 512 waves", 32768, reps);
  run<1, -1>("110 multiplications, all lanes active, 512 waves", 32768, reps);
  run<8, -1>("110 multiplications, 1 lane in 8 active, 512 waves", 32768, reps);
  run<-8, -1>("110 multiplications, lanes 0..7 active, 512 waves", 32768, reps);
  run<-1, -1>("110 multiplications, lane 0 only, 512 waves", 32768, reps);
  run<-4, -1>("110 multiplications, lanes 0..3, 512 waves", 32768, reps);
  run<-12, -1>("110 multiplications, lanes 0..11, 512 waves", 32768, reps);
  run<-16, -1>("110 multiplications, lanes 0..15, 512 waves", 32768, reps);
  run<-1, -2>("110 multiplications, lane 0 only (others returned at entry), 512 waves", 32768, reps);
  run<-8, -2>("110 multiplications, lanes 0..7 (others returned at entry), 512 waves", 32768, reps);
  return 0;
}

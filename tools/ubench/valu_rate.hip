// Micro-benchmark: VALU issue interval on gfx950, in REAL shader cycles.
// Round 4: waves/SIMD 1..8, 8 and 16 independent chains per wave, the instructions the kernels
// consist of (v_mad_i64_i32 first), plain controls (v_add_u32, v_and_b32, v_lshlrev_b64, v_pk_fma_f32, v_fma_f32),
// and no assumed clock: every wave brackets its loop with s_memtime (shader clock) and s_memrealtime (100 MHz
// constant clock), so cycles per instruction come from the chip's own counter and the clock it held from the ratio.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define ITERS 8192

struct Stamp { unsigned long long c0, c1, r0, r1; };

template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) k(uint32_t* out, Stamp* stamps, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t acc[CHAINS];
  uint32_t r[CHAINS];
  double d[CHAINS];
  for (int c = 0; c < CHAINS; ++c) { acc[c] = a + c; r[c] = b + c; d[c] = (double)(a + c); }
  double da = (double)a * 1e-9, db = (double)b * 1e-9;
  uint64_t pk = ((uint64_t)a << 32) | b;
  if (OP == 33) asm volatile("s_mov_b32 s12, 0x55555555\n\ts_mov_b32 s13, 0x33333333" ::: "s12", "s13");
  unsigned long long c0 = __builtin_readcyclecounter();  // s_memtime
  unsigned long long r0 = wall_clock64();                // s_memrealtime
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
      if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[c]) : "v"(a));
      if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[c]) : "v"(a));
      if (OP == 3) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[c]) : "v"(da), "v"(db));
      if (OP == 4) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(r[c]) : "v"(a), "v"(b));
      if (OP == 7) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[c]) : "v"(a) : "vcc");
      if (OP == 8) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(acc[c]) : "v"(acc[(c + 1) % CHAINS]));
      if (OP == 9) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[c]) : "v"(a), "v"(b));
      if (OP == 11) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "s10", "s11");
      if (OP == 12) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[c]) : "v"(a), "v"(b));
      if (OP == 14) asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(r[c]) : "v"(a));
      if (OP == 15) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r[c]) : "v"(a) : "vcc");
      if (OP == 16) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
      if (OP == 17) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[c]) : "v"(a));
      if (OP == 18) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[c]) : "v"(a));
      if (OP == 19) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(acc[c]));
      if (OP == 20) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[c]) : "v"(pk));
      if (OP == 21) asm volatile("v_ashrrev_i64 %0, 29, %0" : "+v"(acc[c]));
      if (OP == 22) asm volatile("v_mov_b32 %0, %1" : "+v"(r[c]) : "v"(a));
      if (OP == 23) asm volatile("v_mad_i64_i32 %0, s[10:11], %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "s10", "s11");
      if (OP == 24) {  // the kernels' real mix: one product column step = mad + mad + 64-bit shift + mask
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
        asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[c]) : "v"(a));
      }
      if (OP == 25) asm volatile("v_bfe_i32 %0, %0, 0, 29" : "+v"(r[c]));
      if (OP == 26) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(r[c]) : "v"(a));
      if (OP == 27) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[c]) : "v"(a));
      if (OP == 28) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r[c]));
      if (OP == 29) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(r[c]));
      if (OP == 30) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r[c]) : "v"(a));
      if (OP == 31) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[c]) : "v"(a));
      if (OP == 32) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r[c]) : "v"(a));
      if (OP == 33) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[12:13]" : "+v"(r[c]) : "v"(a));  // mask set before the loop
      if (OP == 34) asm volatile("v_mov_b64 %0, %1" : "+v"(acc[c]) : "v"(pk));
      if (OP == 35) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[c]));
      if (OP == 36) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(r[c]));
      if (OP == 37) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[c]));
      if (OP == 38) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(r[c]) : "v"(a) : "vcc");
      if (OP == 39) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(d[c]) : "v"(a));
      if (OP == 40) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[c]));
      if (OP == 41) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[c]));
      if (OP == 42) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[c]) : "v"(da));
      if (OP == 43) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(da));
      if (OP == 44) asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(r[c]) : "v"(da));
      if (OP == 45) asm volatile("v_cmp_ne_u32 vcc, %0, %1" : : "v"(r[c]), "v"(a) : "vcc");
      if (OP == 46) {  // the reduction step of fe_reduce (round 4): and, 64-bit shift, 64-bit add, two multiply-adds
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(r[c]) : "v"((uint32_t)acc[c]), "v"(a));
        asm volatile("v_ashrrev_i64 %0, 29, %1" : "=v"(acc[c]) : "v"(acc[(c + 1) % CHAINS]));
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[(c + 2) % CHAINS]) : "v"(acc[c]));
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[(c + 3) % CHAINS]) : "v"(r[c]), "v"(b) : "vcc");
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[(c + 4) % CHAINS]) : "v"(r[c]), "v"(a) : "vcc");
      }
    }
  }
  unsigned long long c1 = __builtin_readcyclecounter();
  unsigned long long r1 = wall_clock64();
  uint32_t s = 0;
  for (int c = 0; c < CHAINS; ++c) s += (uint32_t)acc[c] + r[c] + (uint32_t)d[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    Stamp st = {c0, c1, r0, r1};
    stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st;
  }
}

template <int OP, int CHAINS>
void run(const char* name, int waves_per_simd, int per_instr = 1) {
  int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block per CU
  int nwaves = blocks * 4;
  uint32_t* out;
  Stamp* stamps;
  hipMalloc(&out, blocks * 256 * 4);
  hipMalloc(&stamps, nwaves * sizeof(Stamp));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) k<OP, CHAINS><<<blocks, 256>>>(out, stamps, 1);  // warm the clock up
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int REPS = 5;
  for (int rep = 0; rep < REPS; ++rep) k<OP, CHAINS><<<blocks, 256>>>(out, stamps, rep);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<Stamp> h(nwaves);
  hipMemcpy(h.data(), stamps, nwaves * sizeof(Stamp), hipMemcpyDeviceToHost);
  // per wave: shader cycles and 100 MHz ticks across its loop; all waves of a SIMD run side by side, so the
  // SIMD issued waves_per_simd * ITERS * CHAINS instructions in (about) one wave's span.
  std::vector<double> cyc(nwaves), mhz(nwaves);
  for (int i = 0; i < nwaves; ++i) {
    cyc[i] = (double)(h[i].c1 - h[i].c0);
    double ticks = (double)(h[i].r1 - h[i].r0);
    mhz[i] = ticks > 0 ? cyc[i] / ticks * 100.0 : 0.0;
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(mhz.begin(), mhz.end());
  double instr_per_simd_per_launch = (double)waves_per_simd * ITERS * CHAINS * per_instr;
  double med_cyc = cyc[nwaves / 2], med_mhz = mhz[nwaves / 2];
  // cross-check with events: wall time per launch * measured clock
  double ev_cyc = ms * 1e-3 / REPS * med_mhz * 1e6;
  // PRIMARY figure: launch duration from HIP events x the clock the waves measured, per instruction a SIMD issued.
  // The in-wave span is printed next to it: when it is much smaller, the waves of a SIMD did not run side by side
  // for the whole launch (oldest-first arbitration lets the first waves finish early).
  printf("%-28s chains=%2d waves/SIMD=%d  %7.3f ms/launch  clock %4.0f MHz  %5.2f cycles/instr/SIMD   "
         "(in-wave median span %5.2f, at nominal 2.4 GHz %5.2f)\n",
         name, CHAINS, waves_per_simd, ms / REPS, med_mhz, ev_cyc / instr_per_simd_per_launch,
         med_cyc / instr_per_simd_per_launch, ms * 1e-3 / REPS * 2.4e9 / instr_per_simd_per_launch);
  hipFree(out);
  hipFree(stamps);
}

template <int CHAINS>
void sweep(int w) {
  run<16, CHAINS>("v_mad_i64_i32 (vcc)", w);
  run<23, CHAINS>("v_mad_i64_i32 (sgpr carry)", w);
  run<0, CHAINS>("v_mad_u64_u32 (vcc)", w);
  run<1, CHAINS>("v_mul_lo_u32", w);
  run<2, CHAINS>("v_mul_hi_u32", w);
  run<26, CHAINS>("v_mul_i32_i24", w);
  run<4, CHAINS>("v_mad_u32_u24", w);
  run<17, CHAINS>("v_add_u32", w);
  run<18, CHAINS>("v_and_b32", w);
  run<22, CHAINS>("v_mov_b32", w);
  run<25, CHAINS>("v_bfe_i32", w);
  run<9, CHAINS>("v_add3_u32", w);
  run<14, CHAINS>("v_alignbit_b32", w);
  run<7, CHAINS>("v_add_co_u32", w);
  run<15, CHAINS>("v_addc_co_u32", w);
  run<8, CHAINS>("v_lshl_add_u64", w);
  run<19, CHAINS>("v_lshlrev_b64", w);
  run<21, CHAINS>("v_ashrrev_i64", w);
  run<12, CHAINS>("v_fma_f32", w);
  run<20, CHAINS>("v_pk_fma_f32", w);
  run<3, CHAINS>("v_fma_f64", w);
  run<24, CHAINS>("mad_i64_i32 + and_b32 pair", w, 2);
  run<46, CHAINS>("fe_reduce step (5 instr)", w, 5);
  run<27, CHAINS>("v_sub_u32", w);
  run<31, CHAINS>("v_xor_b32", w);
  run<28, CHAINS>("v_lshlrev_b32", w);
  run<29, CHAINS>("v_ashrrev_i32", w);
  run<30, CHAINS>("v_lshl_add_u32", w);
  run<32, CHAINS>("v_lshl_or_b32", w);
  run<36, CHAINS>("v_bfe_u32", w);
  run<33, CHAINS>("v_cndmask_b32 (sgpr)", w);
  run<45, CHAINS>("v_cmp_ne_u32", w);
  run<38, CHAINS>("v_sub_co_u32", w);
  run<34, CHAINS>("v_mov_b64", w);
  run<35, CHAINS>("v_lshrrev_b64", w);
  run<37, CHAINS>("v_mov_b32 dpp quad_perm", w);
  run<39, CHAINS>("v_cvt_f64_i32", w);
  run<44, CHAINS>("v_cvt_i32_f64", w);
  run<40, CHAINS>("v_rndne_f64", w);
  run<41, CHAINS>("v_rcp_f64", w);
  run<42, CHAINS>("v_mul_f64", w);
  run<43, CHAINS>("v_add_f64", w);
}

int main(int argc, char** argv) {
  int dev = 0;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, dev);
  printf("# %s, %d CUs, clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  // 16 chains: no instruction waits for its own result (8 chains measured the same from 2 waves per SIMD on)
  for (int w : {1, 2, 4, 6, 8}) sweep<16>(w);
  sweep<8>(1);
  sweep<8>(2);
  return 0;
}

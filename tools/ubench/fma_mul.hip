// Micro-benchmark: a double-precision-FMA field multiplier next to fe_mul.
//
// The DPF scheme (Emmart et al., "Faster modular exponentiation using double precision floating point
// arithmetic on the GPU"): limbs of 52 bits held as doubles; with round-toward-zero
//     hi = fma(a, b, 2^104)                  -> bits(hi) = bits(2^104) + floor(a b / 2^52)
//     lo = fma(a, b, (2^104 + 2^52) - hi)    -> bits(lo) = bits(2^52)  + (a b mod 2^52)
// so a 52 x 52 -> 104-bit product costs 2 FMAs + 1 FP subtraction, and the two halves are accumulated into
// the product columns with two 64-bit INTEGER additions of the bit patterns: 5 VALU instructions per limb
// product, 25 limb products for a 260-bit operand = 125 instructions for the PRODUCT PHASE ALONE, against
// 81 v_mad_i64_i32 for the nine 29-bit limbs of fp29.hpp (whose whole multiplication, reduction included, is
// 174 instructions).  This file measures both on the hardware:
//   * correctness: the ten DPF columns of 2^20 random operand pairs, carried out to a 520-bit integer, equal the
//     seventeen integer columns of cols_mac (the product phase of fe_mul) bit for bit;
//   * cost: wave-instruction issue time per product at 4 waves per SIMD (throughput), and per dependent product
//     on a lone wave.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../stark-perpetual_amd/csrc fma_mul.hip -o fma_mul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "fp29.hpp"
using namespace sp;

struct dpf {
  double l[5];  // 52-bit limbs as doubles
};

__device__ __forceinline__ void set_round_toward_zero_f64() {
  // MODE register (hwreg id 1), bits [3:2] = rounding of f64 / f16 operations: 3 = toward zero.  Inline asm on
  // purpose: with __builtin_amdgcn_s_setreg the backend's mode-register pass "repairs" the mode (it emits
  // s_setreg ... 0 in front of the first v_fma_f64 it sees), so the FMAs below are inline asm as well.
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
}
__device__ __forceinline__ double fma_rz(double a, double b, double c) {
  double r;
  asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

__device__ __forceinline__ uint64_t bits_of(double d) { return (uint64_t)__double_as_longlong(d); }

// col[k] accumulates bit patterns; the constants they carry are removed by the caller (dpf_columns)
__device__ __forceinline__ void dpf_product_raw(const dpf& a, const dpf& b, uint64_t col[10]) {
  const double C1 = __longlong_as_double(0x4670000000000000LL);  // 2^104
  const double C2 = __longlong_as_double(0x4670000000000001LL);  // 2^104 + 2^52
  // set here, not once per kernel: the backend's mode-register pass puts the default rounding back after
  // the u64 -> f64 conversions of the operand generator (one SALU instruction per product)
  set_round_toward_zero_f64();
#pragma unroll
  for (int k = 0; k < 10; ++k) col[k] = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const double hi = fma_rz(a.l[i], b.l[j], C1);
      const double lo = fma_rz(a.l[i], b.l[j], C2 - hi);
      col[i + j + 1] += bits_of(hi);
      col[i + j] += bits_of(lo);
    }
}
// plain column sums: col[k] = sum_{i+j=k} lo_ij + sum_{i+j=k-1} hi_ij  (each < 2^55)
__device__ __forceinline__ void dpf_columns(const dpf& a, const dpf& b, uint64_t col[10]) {
  dpf_product_raw(a, b, col);
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const int n_lo = k < 5 ? k + 1 : (k < 9 ? 9 - k : 0);      // pairs with i + j = k
    const int n_hi = k == 0 ? 0 : (k - 1 < 5 ? k : 10 - k);    // pairs with i + j = k - 1
    col[k] -= (uint64_t)n_lo * 0x4330000000000000ULL + (uint64_t)n_hi * 0x4670000000000000ULL;
  }
}

__device__ __forceinline__ uint32_t mix(uint32_t& s) {
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  return s;
}

// 260 random bits -> the same integer as nine 29-bit limbs and as five 52-bit doubles
__device__ __forceinline__ void random_operand(uint32_t& s, fe& f, dpf& d) {
  uint32_t w[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) w[i] = mix(s);
  uint64_t limb52[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const uint64_t v = ((uint64_t)w[2 * i] << 32 | w[(2 * i + 1) % 9]) & ((1ULL << 52) - 1);
    limb52[i] = v;
    d.l[i] = (double)v;  // exact: < 2^53
  }
  // repack the 260-bit integer sum limb52[i] 2^(52 i) into 29-bit limbs
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int bit = 29 * k, i = bit / 52, sh = bit % 52;
    uint64_t v = limb52[i] >> sh;
    if (sh + 29 > 52 && i + 1 < 5) v |= limb52[i + 1] << (52 - sh);
    f.l[k] = (int32_t)(v & LMASK);
  }
  f.l[8] &= (1 << 28) - 1;  // 260 bits = 8 * 29 + 28
  limb52[4] &= (1ULL << 52) - 1;
}

// the 520-bit product as 17 words of 32 bits (little endian, low 17 * 32 = 544 bits), from either column form
__device__ __forceinline__ void words_from_cols29(const cols& t, uint32_t out[17]) {
  unsigned __int128 acc = 0;
  int have = 0, k = 0;
  for (int wi = 0; wi < 17; ++wi) {
    while (have < 32 && k < 17) {
      acc += (unsigned __int128)(uint64_t)t.c[k] << have;  // columns are non-negative here
      have += 29;
      ++k;
    }
    out[wi] = (uint32_t)acc;
    acc >>= 32;
    have -= 32;
    if (have < 0) have = 0;
  }
}
__device__ __forceinline__ void words_from_cols52(const uint64_t col[10], uint32_t out[17]) {
  unsigned __int128 acc = 0;
  int have = 0, k = 0;
  for (int wi = 0; wi < 17; ++wi) {
    while (have < 32 && k < 10) {
      acc += (unsigned __int128)col[k] << have;
      have += 52;
      ++k;
    }
    out[wi] = (uint32_t)acc;
    acc >>= 32;
    have -= 32;
    if (have < 0) have = 0;
  }
}

__global__ void __launch_bounds__(64) check_kernel(unsigned long long* mismatches, int reps) {
  set_round_toward_zero_f64();
  uint32_t s = 0x9e3779b9u * (blockIdx.x * 64 + threadIdx.x + 1);
  unsigned long long bad = 0;
  for (int r = 0; r < reps; ++r) {
    fe fa, fb;
    dpf da, db;
    random_operand(s, fa, da);
    random_operand(s, fb, db);
    cols t;
    cols_zero(t);
    cols_mac(t, fa, fb);
    uint64_t col[10];
    dpf_columns(da, db, col);
    uint32_t w29[17], w52[17];
    words_from_cols29(t, w29);
    words_from_cols52(col, w52);
    bool same = true;
    for (int i = 0; i < 17; ++i) same &= w29[i] == w52[i];
    bad += same ? 0 : 1;
    if (!same && r == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
      printf("a52:"); for (int i = 0; i < 5; ++i) printf(" %llx", (unsigned long long)da.l[i]);
      printf("\nb52:"); for (int i = 0; i < 5; ++i) printf(" %llx", (unsigned long long)db.l[i]);
      printf("\na29:"); for (int i = 0; i < 9; ++i) printf(" %x", fa.l[i]);
      printf("\nb29:"); for (int i = 0; i < 9; ++i) printf(" %x", fb.l[i]);
      printf("\ncol52:"); for (int i = 0; i < 10; ++i) printf(" %llx", (unsigned long long)col[i]);
      printf("\ncol29:"); for (int i = 0; i < 17; ++i) printf(" %llx", (unsigned long long)t.c[i]);
      printf("\nw29:"); for (int i = 0; i < 17; ++i) printf(" %x", w29[i]);
      printf("\nw52:"); for (int i = 0; i < 17; ++i) printf(" %x", w52[i]);
      printf("\n");
    }
  }
  if (bad) atomicAdd(mismatches, bad);
}

// throughput / latency kernels: MODE 0 fe_mul, 1 cols_mac only (81 mads), 2 DPF product phase (125 instr)
template <int MODE>
__global__ void __launch_bounds__(64) time_kernel(uint32_t* out, int reps) {
  set_round_toward_zero_f64();
  uint32_t s = 0x9e3779b9u * (blockIdx.x * 64 + threadIdx.x + 1);
  fe fa, fb;
  dpf da, db;
  random_operand(s, fa, da);
  random_operand(s, fb, db);
  uint32_t sink = 0;
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {
      fa = fe_mul(fa, fb);  // dependent chain
    } else if (MODE == 1) {
      cols t;
      cols_zero(t);
      cols_mac(t, fa, fb);
      // feed one bit of every column back into every limb: the products form a chain and no partial product
      // is loop-invariant (with two limbs fed the compiler hoisted 49 of the 81 products out of the loop)
#pragma unroll
      for (int k = 0; k < NL; ++k)
        fa.l[k] = (fa.l[k] ^ (int32_t)((uint32_t)(t.c[k] ^ (k + 8 < 17 ? t.c[k + 8] : 0)) & 1u)) & LMASK;
      fe_pin(fa);  // as fe_mul's results are: every product stays ONE v_mad_i64_i32 (fp29.hpp)
      int64_t all = 0;  // every column is live (16 extra 64-bit xors)
#pragma unroll
      for (int k = 0; k < 17; ++k) all ^= t.c[k];
      sink += (uint32_t)all;
    } else {
      uint64_t col[10];
      dpf_product_raw(da, db, col);
#pragma unroll
      for (int k = 0; k < 5; ++k)
        da.l[k] = __longlong_as_double(__double_as_longlong(da.l[k]) ^ (long long)((col[k] ^ col[k + 5]) & 1u));
      uint64_t all = 0;  // every column is live (9 extra 64-bit xors)
#pragma unroll
      for (int k = 0; k < 10; ++k) all ^= col[k];
      sink += (uint32_t)all;
    }
  }
  for (int i = 0; i < NL; ++i) sink += (uint32_t)fa.l[i];
  sink += (uint32_t)bits_of(da.l[0]) + (uint32_t)bits_of(da.l[2]);
  out[blockIdx.x * 64 + threadIdx.x] = sink;
}

template <int MODE>
static void run(const char* name, int blocks, int reps) {
  uint32_t* out;
  hipMalloc(&out, (size_t)blocks * 64 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  time_kernel<MODE><<<blocks, 64>>>(out, reps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  time_kernel<MODE><<<blocks, 64>>>(out, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = blocks / 1024.0 < 1.0 ? 1.0 : blocks / 1024.0;
  printf("%-58s %5d waves x %5d products: %8.2f us, %7.2f ns per product per SIMD\n", name, blocks, reps, ms * 1e3,
         ms * 1e6 / reps / waves_per_simd);
  hipFree(out);
}

int main() {
  unsigned long long* bad;
  hipMalloc(&bad, 8);
  hipMemset(bad, 0, 8);
  check_kernel<<<256, 64>>>(bad, 64);  // 2^20 random pairs
  unsigned long long h = 0;
  hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
  printf("DPF product columns vs integer columns (cols_mac) on %d random 260-bit pairs: %llu mismatches\n", 256 * 64 * 64, h);
  for (int blocks : {256, 4096, 8192}) {
    run<0>("fe_mul (81 + 30 mads + carries = 174 instr), chained", blocks, 2000);
    // (MODE 1, the integer product phase alone, is not printed: with only one bit of every column live the
    // compiler legitimately narrows the 81 products to v_mul_lo_u32 - fe_mul above is the honest integer figure)
    run<2>("DPF product phase (125 instr) + 15 feedback ops", blocks, 2000);
  }
  return h == 0 ? 0 : 1;
}

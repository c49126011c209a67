// Micro-benchmark + check: batch-affine additions against the XYZZ mixed addition of the bulk
// hash kernel, as the inner loop of "one lane sums table points for K hashes at once".
//
//   XYZZ (what ped_accumulate_kernel does)   acc += q             8M + 2S per addition, no inversion until the end
//   batch-affine, K accumulators per lane    acc_k += q_k, k < K  one SHARED inversion per step: Montgomery's trick
//                                            over the K differences of the lane (3M each) x the four lanes of a DPP
//                                            quad (fe_inv_shared_quad<2>: 4M + one quad-split Lehmer inversion),
//                                            then lambda, lambda^2, x3, y3 (2M + 1S): 5M + 1S + inversion / K
// Both kernels add the SAME synthetic points in the same order (a table in L2; the HBM gathers of the real kernel
// cost the same in either design and are left out), so the result is checked: X_xyzz == x_affine * ZZ_xyzz for
// every accumulator.  Variants: accumulators in REGISTERS (K = 2, 4, 8, 16: straight-line code, the compiler decides
// what spills), in LDS (K = 2, 4, 8: 72 B per accumulator per lane - K = 4 is what 160 KB hold at two waves per SIMD) or
// in SCRATCH memory (K = 8, 16, 32: rolled loops, arrays indexed at run time).
// Reported: VGPRs / scratch / occupancy of each variant as compiled, ns per addition with the chip full at that
// occupancy, additions per second of the whole chip.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../stark-perpetual_amd/csrc batch_affine.hip -o batch_affine
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
#include <vector>
#include "quad.hpp"
using namespace sp;

// Straight-line "for k in 0..K-1" with k a compile-time constant: accumulators in registers cannot be indexed, and
// hipcc declines to unroll a `#pragma unroll` loop of this size fully (it kept the arrays in scratch memory instead).
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int TABLE = 4096;  // synthetic affine points (Montgomery N-form limbs), L2 resident
struct pt36 {
  int32_t x[NL], y[NL];
};
__device__ __forceinline__ aff table_point(const pt36* __restrict__ tab, unsigned idx) {
  const pt36& p = tab[idx & (TABLE - 1)];
  aff q;
#pragma unroll
  for (int i = 0; i < NL; ++i) { q.x.l[i] = p.x[i]; q.y.l[i] = p.y[i]; }
  return q;
}
__device__ __forceinline__ unsigned point_index(size_t hash, int step) {
  return (unsigned)(hash * 2654435761u + (unsigned)step * 40503u + (unsigned)(hash >> 7));
}

// ---- XYZZ: K hashes one after the other (K only sets the work per lane equal to the affine kernels') ----------
template <int K>
__global__ void __launch_bounds__(256, 2) xyzz_kernel(const pt36* __restrict__ tab, int steps, int32_t* __restrict__ out,
                                                       size_t lanes) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < K; ++k) {
    const size_t h = (size_t)k * lanes + t;
    xyzz acc = xyzz_from_aff(table_point(tab, point_index(h, 0)));
#pragma unroll 1
    for (int g = 1; g <= steps; ++g) acc = xyzz_madd(acc, table_point(tab, point_index(h, g)));
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      out[((size_t)(2 * i) * K + k) * lanes + t] = acc.X.l[i];
      out[((size_t)(2 * i + 1) * K + k) * lanes + t] = acc.ZZ.l[i];
    }
  }
}

// ---- batch-affine ---------------------------------------------------------------------------------------------
// LDS_ACC: the K accumulators of a lane live in LDS (int32 planes, lane-major: no bank conflicts); otherwise in
// registers (fully unrolled over k: a register file cannot be indexed).
template <int K, bool LDS_ACC>
__global__ void __launch_bounds__(256) affine_kernel(const pt36* __restrict__ tab, int steps, int32_t* __restrict__ out,
                                                     size_t lanes) {
  extern __shared__ int32_t lds[];  // LDS_ACC: [K][18][256]
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int kq = (int)(threadIdx.x & 3);
  aff acc[K];
  auto put = [&](int k, const aff& a) {
    if constexpr (LDS_ACC) {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        lds[((k * 18 + i) << 8) + threadIdx.x] = a.x.l[i];
        lds[((k * 18 + 9 + i) << 8) + threadIdx.x] = a.y.l[i];
      }
    } else {
      acc[k] = a;
    }
  };
  auto get = [&](int k) {
    if constexpr (LDS_ACC) {
      aff a;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        a.x.l[i] = lds[((k * 18 + i) << 8) + threadIdx.x];
        a.y.l[i] = lds[((k * 18 + 9 + i) << 8) + threadIdx.x];
      }
      return a;
    } else {
      return acc[k];
    }
  };
  static_for<K>([&](auto kc) { put(kc, table_point(tab, point_index((size_t)kc * lanes + t, 0))); });
#pragma unroll 1
  for (int g = 1; g <= steps; ++g) {
    // forward: differences and their running product (the table point is fetched again on the way back: the
    // real kernel would keep its 64-byte entry in registers or re-read it from L2 - 16 VGPRs per hash either way)
    fe dx[K], pre[K];
    fe run = FE_ONE_M;
    static_for<K>([&](auto kc) {
      constexpr int k = kc;
      const aff q = table_point(tab, point_index((size_t)k * lanes + t, g));
      const aff a = get(k);
      dx[k] = fe_sub(q.x, a.x);  // B = 1, signed
      pre[k] = run;
      run = k == 0 ? dx[0] : fe_mul(run, dx[k]);
    });
    // one inversion for the K differences of this lane and of its three quad neighbours
    fe inv = fe_inv_shared_quad<2, false>(fe_carry(run), kq);
    static_for<K>([&](auto kc) {
      constexpr int k = K - 1 - kc;
      const fe ik = k == 0 ? inv : fe_mul(inv, pre[k]);  // 1 / dx_k
      if (k > 0) inv = fe_mul(inv, dx[k]);
      const aff q = table_point(tab, point_index((size_t)k * lanes + t, g));
      const aff a = get(k);
      const fe lam = fe_mul(fe_carry(fe_sub(q.y, a.y)), ik);
      aff r;
      r.x = fe_carry(fe_sub(fe_sub(fe_sqr(lam), a.x), q.x));        // lambda^2 - x1 - x2
      r.y = fe_carry(fe_sub(fe_mul(lam, fe_sub(a.x, r.x)), a.y));   // lambda (x1 - x3) - y1
      put(k, r);
    });
  }
  static_for<K>([&](auto kc) {
    constexpr int k = kc;
    const aff a = get(k);
#pragma unroll
    for (int i = 0; i < NL; ++i) out[((size_t)i * K + k) * lanes + t] = a.x.l[i];
  });
}

// The third home for the accumulators: PRIVATE (scratch) memory - rolled loops over k, arrays indexed at run time, so
// acc / dx / pre live in the lane's scratch (HBM-backed, through the vector L1 and L2).  K is not bounded by the
// register file here; what it costs is the traffic: 3 x 36 B written and 4 x 36 B read per addition.
template <int K>
__global__ void __launch_bounds__(256) affine_scratch_kernel(const pt36* __restrict__ tab, int steps,
                                                             int32_t* __restrict__ out, size_t lanes) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int kq = (int)(threadIdx.x & 3);
  aff acc[K];
  fe dx[K], pre[K];
#pragma unroll 1
  for (int k = 0; k < K; ++k) acc[k] = table_point(tab, point_index((size_t)k * lanes + t, 0));
#pragma unroll 1
  for (int g = 1; g <= steps; ++g) {
    fe run = FE_ONE_M;
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const aff q = table_point(tab, point_index((size_t)k * lanes + t, g));
      dx[k] = fe_sub(q.x, acc[k].x);
      pre[k] = run;
      run = fe_mul(run, fe_carry(dx[k]));
    }
    fe inv = fe_inv_shared_quad<2, false>(run, kq);
#pragma unroll 1
    for (int k = K - 1; k >= 0; --k) {
      const fe ik = fe_mul(inv, pre[k]);
      inv = fe_mul(inv, dx[k]);
      const aff q = table_point(tab, point_index((size_t)k * lanes + t, g));
      const aff a = acc[k];
      const fe lam = fe_mul(fe_carry(fe_sub(q.y, a.y)), ik);
      aff r;
      r.x = fe_carry(fe_sub(fe_sub(fe_sqr(lam), a.x), q.x));
      r.y = fe_carry(fe_sub(fe_mul(lam, fe_sub(a.x, r.x)), a.y));
      acc[k] = r;
    }
  }
#pragma unroll 1
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int i = 0; i < NL; ++i) out[((size_t)i * K + k) * lanes + t] = acc[k].x.l[i];
}

// X_xyzz == x_affine * ZZ_xyzz (all Montgomery): one thread per accumulator
__global__ void check_kernel(const int32_t* __restrict__ xyzz_out, const int32_t* __restrict__ aff_out, size_t total,
                             int* bad) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= total) return;
  fe X, ZZ, x;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    X.l[i] = xyzz_out[(size_t)(2 * i) * total + j];
    ZZ.l[i] = xyzz_out[(size_t)(2 * i + 1) * total + j];
    x.l[i] = aff_out[(size_t)i * total + j];
  }
  if (!fe_eq(X, fe_mul(x, ZZ))) atomicAdd(bad, 1);
}

struct Result {
  double ns_per_add, adds_per_s;
};
template <typename Kern>
static Result time_kernel(Kern kern, const char* name, int K, size_t lds_bytes, int blocks, const pt36* tab, int steps,
                          int32_t* out, bool xyzz_form) {
  const size_t lanes = (size_t)blocks * 256;
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
  if (lds_bytes > 65536) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  int per_cu = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes, 0, tab, steps, out, lanes);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes, 0, tab, steps, out, lanes);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double adds = (double)lanes * K * steps;
  Result r{ms * 1e6 / ((double)K * steps), adds / (ms * 1e-3)};
  printf("%-44s K=%2d  VGPR %3d  scratch %5zu B  LDS %6zu B  blocks/CU %d  lanes %7zu : %8.3f ms, %7.1f ns per addition per lane, "
         "%.3e additions/s\n",
         name, K, fa.numRegs, (size_t)fa.localSizeBytes, lds_bytes, per_cu, lanes, ms, r.ns_per_add, r.adds_per_s);
  (void)xyzz_form;
  return r;
}

template <int K, int HOME>  // HOME 0: registers, 1: LDS, 2: scratch memory
static void run_variant(const pt36* tab, int steps, int32_t* ref_out, int32_t* out, int* bad, double base_rate) {
  // the chip full at the occupancy this variant compiles to: blocks = 256 CUs x blocks per CU (x 2 rounds)
  constexpr bool LDS_ACC = HOME == 1;
  const size_t lds_bytes = LDS_ACC ? (size_t)K * 18 * 256 * 4 : 0;
  void (*kern)(const pt36*, int, int32_t*, size_t);
  if constexpr (HOME == 2) kern = affine_scratch_kernel<K>;
  else kern = affine_kernel<K, LDS_ACC>;
  if (lds_bytes > 65536) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  int per_cu = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds_bytes);
  if (per_cu < 1) { printf("affine K=%d home %d: does not fit a CU\n", K, HOME); return; }
  const int blocks = 256 * per_cu * 2;
  const size_t lanes = (size_t)blocks * 256;
  // reference sums for exactly these hashes, then the variant, then the comparison
  hipLaunchKernelGGL((xyzz_kernel<K>), dim3(blocks), dim3(256), 0, 0, tab, steps, ref_out, lanes);
  char name[96];
  snprintf(name, sizeof name, "batch-affine, accumulators in %s", HOME == 1 ? "LDS" : HOME == 2 ? "scratch memory" : "registers");
  const Result r = time_kernel(kern, name, K, lds_bytes, blocks, tab, steps, out, false);
  hipMemset(bad, 0, 4);
  const size_t total = lanes * K;
  hipLaunchKernelGGL(check_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, ref_out, out, total, bad);
  int h = -1;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("    check: %d of %zu accumulators differ from the XYZZ sums;  rate against XYZZ: %+.1f %%\n", h, total,
         100.0 * (r.adds_per_s / base_rate - 1.0));
}

int main() {
  const int steps = 18;  // additions per hash at 26-bit windows
  std::vector<pt36> host(TABLE);
  uint64_t s = 0x243F6A8885A308D3ull;
  for (auto& p : host)
    for (int i = 0; i < NL; ++i) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      p.x[i] = (int32_t)((s >> 20) & (i == 8 ? 0x3ffff : LMASK));
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      p.y[i] = (int32_t)((s >> 20) & (i == 8 ? 0x3ffff : LMASK));
    }
  pt36* tab;
  hipMalloc(&tab, sizeof(pt36) * TABLE);
  hipMemcpy(tab, host.data(), sizeof(pt36) * TABLE, hipMemcpyHostToDevice);
  const size_t max_acc = (size_t)256 * 8 * 2 * 256 * 32;  // lanes x K upper bound
  int32_t *ref_out, *out;
  int* bad;
  hipMalloc(&ref_out, max_acc * 18 * 4);
  hipMalloc(&out, max_acc * 9 * 4);
  hipMalloc(&bad, 4);
  printf("batch-affine against XYZZ additions: %d additions per hash, synthetic table in L2, no HBM gathers in either\n", steps);
  // baseline: the XYZZ loop with the chip full at two waves per SIMD (2 x 1024 blocks: two rounds)
  const Result base = time_kernel(xyzz_kernel<4>, "XYZZ mixed addition (8M + 2S), 4 hashes in turn", 4, 0, 2048, tab, steps, ref_out, true);
  run_variant<2, 0>(tab, steps, ref_out, out, bad, base.adds_per_s);
  run_variant<4, 0>(tab, steps, ref_out, out, bad, base.adds_per_s);
  run_variant<8, 0>(tab, steps, ref_out, out, bad, base.adds_per_s);
  run_variant<16, 0>(tab, steps, ref_out, out, bad, base.adds_per_s);
  run_variant<2, 1>(tab, steps, ref_out, out, bad, base.adds_per_s);
  run_variant<4, 1>(tab, steps, ref_out, out, bad, base.adds_per_s);
  run_variant<8, 1>(tab, steps, ref_out, out, bad, base.adds_per_s);
  run_variant<8, 2>(tab, steps, ref_out, out, bad, base.adds_per_s);
  run_variant<16, 2>(tab, steps, ref_out, out, bad, base.adds_per_s);
  run_variant<32, 2>(tab, steps, ref_out, out, bad, base.adds_per_s);
  return 0;
}

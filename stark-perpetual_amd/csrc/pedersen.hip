// Batched Pedersen hash and the Merkle / hash-chain drivers built on it.
//
// Reference semantics: pedersen_hash(x, y) = x-coordinate of
//     SHIFT + sum_j x_j C[2+j] + sum_j y_j C[254+j]            (signature.py:300-318)
// computed there with ~252 affine additions (one modular inversion each).  Here one thread sums
// the window-table entries selected by the 504 bits of x || y (signed window plan, context.hip:
// 18 entries with 2^27-entry windows, 23 with the default 2^21) in XYZZ coordinates (8M + 2S per
// entry, no inversion), and a second kernel turns X/ZZ into the canonical affine x with one
// shared inversion per K hashes (Montgomery's trick, K up to 32).
//
// HBM layout
//   tables  aff_packed[entries]      : 64-byte entries, one random 64-byte gather per window
//   inputs  x[n], y[n]               : 32-byte felts, element strides given in felts
//   scratch int32[9][n] x 3          : X, ZZ and prefix products, limb-major so that a wave's
//                                      64 lanes touch 64 consecutive dwords
// Algorithmic bytes per hash: 96 (two felts in, one out).  The kernel is VALU bound
// (~1.75e3 VALU instructions per window addition, 39e3 per hash), not HBM bound - see DESIGN.md.
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "context.hpp"
#include "quad.hpp"

#ifndef SP_ACC_WAVES
#define SP_ACC_WAVES 1
#endif

namespace sp {

// A table entry as loaded (packed 256-bit words), unpacked into limbs only when it is consumed:
// prefetched entries cost 16 VGPRs instead of 18.
struct raw_aff {
  u256 x, y;
};
__device__ __forceinline__ raw_aff ld_raw(const aff_packed* e) {
  const uint4* q = reinterpret_cast<const uint4*>(e);
  const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  raw_aff r;
  r.x.w[0] = a.x; r.x.w[1] = a.y; r.x.w[2] = a.z; r.x.w[3] = a.w;
  r.x.w[4] = b.x; r.x.w[5] = b.y; r.x.w[6] = b.z; r.x.w[7] = b.w;
  r.y.w[0] = c.x; r.y.w[1] = c.y; r.y.w[2] = c.z; r.y.w[3] = c.w;
  r.y.w[4] = d.x; r.y.w[5] = d.y; r.y.w[6] = d.z; r.y.w[7] = d.w;
  return r;
}
__device__ __forceinline__ aff unpack_raw(const raw_aff& t) {
  aff r;
  r.x = fe_unpack(t.x);
  r.y = fe_unpack(t.y);
  return r;
}

// The 504-bit string x || y the window plan is laid over (context.hpp PedPlan), low bits first.
struct bits512 {
  uint32_t w[16];
};
__device__ __forceinline__ bits512 concat_xy(const u256& x, const u256& y) {  // x < 2^252
  bits512 s;
#pragma unroll
  for (int i = 0; i < 7; ++i) s.w[i] = x.w[i];
  s.w[7] = x.w[7] | (y.w[0] << 28);
#pragma unroll
  for (int i = 0; i < 7; ++i) s.w[8 + i] = (y.w[i] >> 4) | (y.w[i + 1] << 28);
  s.w[15] = y.w[7] >> 4;
  return s;
}
// Pops the low `width` (< 32) bits of s and shifts s right.
__device__ __forceinline__ uint32_t pop_bits(bits512& s, int width) {
  const uint32_t v = s.w[0] & ((1u << width) - 1u);
  bits512 n;
#pragma unroll
  for (int i = 0; i < 15; ++i) n.w[i] = __builtin_amdgcn_alignbit(s.w[i + 1], s.w[i], (uint32_t)width);
  n.w[15] = s.w[15] >> width;
  s = n;
  return v;
}
// Window g of the plan: window 0 is unsigned (w0 bits); window g >= 1 has log2e + 1 bits whose top
// bit is the sign of the entry and whose low bits (complemented for a negative entry) are its index.
struct window_ref {
  const aff_packed* entry;
  bool negative;
};
__device__ __forceinline__ window_ref window_entry(const aff_packed* __restrict__ ped, int g, uint32_t raw,
                                                    int w0, int log2e) {
  window_ref r;
  if (g == 0) {
    r.entry = ped + raw;
    r.negative = false;
  } else {
    const uint32_t mask = (1u << log2e) - 1u;
    r.negative = (raw >> log2e) == 0;
    const uint32_t idx = r.negative ? (~raw & mask) : (raw & mask);
    r.entry = ped + ((size_t)1 << w0) + ((size_t)(g - 1) << log2e) + idx;
  }
  return r;
}
__device__ __forceinline__ int window_width(int g, int w0, int log2e) { return g == 0 ? w0 : log2e + 1; }
__device__ __forceinline__ int window_start(int g, int w0, int log2e) {
  return g == 0 ? 0 : w0 + (g - 1) * (log2e + 1);
}
__device__ __forceinline__ aff signed_aff(const raw_aff& t, bool negative) {
  aff q = unpack_raw(t);
  if (negative) q.y = fe_neg(q.y);
  return q;
}

__device__ __forceinline__ void store_limbs(int32_t* base, size_t n, size_t e, const fe& v) {
#pragma unroll
  for (int k = 0; k < NL; ++k) base[(size_t)k * n + e] = v.l[k];
}
__device__ __forceinline__ fe load_limbs(const int32_t* base, size_t n, size_t e) {
  fe v;
#pragma unroll
  for (int k = 0; k < NL; ++k) v.l[k] = base[(size_t)k * n + e];
  return v;
}

// Where hash e reads its two operands.  Dense mode (src == nullptr): x + 4 e xstride, y + 4 e ystride.
// Gathered mode (sparse Merkle update): src[e] = indices of the two children in the previous level
// `x`, a negative index meaning "the empty-subtree root of this level", which `y` points at.
__device__ __forceinline__ void operand_pointers(const uint64_t* x, const uint64_t* y, size_t xstride,
                                                 size_t ystride, const int2* __restrict__ src, size_t e,
                                                 const uint64_t*& fx, const uint64_t*& fy) {
  if (src == nullptr) {
    fx = x + 4 * e * xstride;
    fy = y + 4 * e * ystride;
  } else {
    const int2 s = src[e];
    fx = s.x >= 0 ? x + 4 * (size_t)s.x : y;
    fy = s.y >= 0 ? x + 4 * (size_t)s.y : y;
  }
}

// Kernel A: one hash per thread -> projective (X, ZZ) in scratch.  `e` = the thread's hash.
__device__ __forceinline__ void
bulk_accumulate(const uint64_t* __restrict__ x, const uint64_t* __restrict__ y, size_t xstride,
                size_t ystride, size_t n, const aff_packed* __restrict__ ped, int w0, int log2e,
                int nwin, int32_t* __restrict__ sX, int32_t* __restrict__ sZZ,
                uint8_t* __restrict__ status, unsigned* __restrict__ flag,
                const int2* __restrict__ src, size_t plane, size_t e) {
  if (e >= n) return;
  const uint64_t *fx, *fy;
  operand_pointers(x, y, xstride, ystride, src, e, fx, fy);
  u256 sx = ld_u256(fx);
  u256 sy = ld_u256(fy);
  uint8_t st = SP_HASH_OK;
  if (!u256_lt(sx, U256_P) || !u256_lt(sy, U256_P)) {  // signature.py:307
    st = SP_HASH_OUT_OF_RANGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) { sx.w[i] = 0; sy.w[i] = 0; }
  }
  bits512 str = concat_xy(sx, sy);
  // The first entry initialises the accumulator; entries i+1 and i+2 are in flight (two 64-byte
  // gathers) while entry i is added, which hides the random-HBM latency behind ~3.5e3 instructions of arithmetic.
  auto next_window = [&](int g) {  // consumes the next window of the string
    return window_entry(ped, g, pop_bits(str, window_width(g, w0, log2e)), w0, log2e);
  };
  const raw_aff e0 = ld_raw(next_window(0).entry);  // window 0 is never negative
  window_ref r1 = next_window(1 < nwin ? 1 : 0);
  raw_aff n1 = ld_raw(r1.entry);
  bool neg1 = r1.negative, neg2 = false;
  raw_aff n2 = n1;
  if (nwin > 2) {
    const window_ref r2 = next_window(2);
    n2 = ld_raw(r2.entry);
    neg2 = r2.negative;
  }
  xyzz acc;
  int first = 1;
  if (nwin > 2) {
    // windows 0 and 1 are both affine: 4M + 2S (mmadd) instead of the 8M + 2S of a mixed addition
    const aff q1 = signed_aff(n1, neg1);
    n1 = n2;
    neg1 = neg2;
    if (3 < nwin) {
      const window_ref r = next_window(3);
      n2 = ld_raw(r.entry);
      neg2 = r.negative;
    }
    acc = xyzz_mmadd(unpack_raw(e0), q1);
    first = 2;
  } else {
    acc = xyzz_from_aff(unpack_raw(e0));
  }
  for (int i = first; i + 1 < nwin; ++i) {
    const aff q = signed_aff(n1, neg1);
    n1 = n2;
    neg1 = neg2;
    if (i + 2 < nwin) {
      const window_ref r = next_window(i + 2);
      n2 = ld_raw(r.entry);
      neg2 = r.negative;
    }
    acc = xyzz_madd(acc, q);
  }
  if (nwin > 1) {
    // only x = X / ZZ of the result is wanted: the last addition skips Y3 and ZZZ3 (3 of 10 multiplications)
    fe X3, ZZ3;
    xyzz_madd_x_only(acc, signed_aff(n1, neg1), X3, ZZ3);
    store_limbs(sX, plane, e, X3);
    store_limbs(sZZ, plane, e, ZZ3);
  } else {
    store_limbs(sX, plane, e, acc.X);
    store_limbs(sZZ, plane, e, acc.ZZ);
  }
  if (st != SP_HASH_OK) {
    if (status) status[e] = st;
    if (flag) atomicOr(flag, (unsigned)st);
  } else if (status) {
    status[e] = SP_HASH_OK;
  }
}

__global__ void __launch_bounds__(256, SP_ACC_WAVES)
ped_accumulate_kernel(const uint64_t* __restrict__ x, const uint64_t* __restrict__ y, size_t xstride,
                      size_t ystride, size_t n, const aff_packed* __restrict__ ped, int w0, int log2e,
                      int nwin, int32_t* __restrict__ sX, int32_t* __restrict__ sZZ,
                      uint8_t* __restrict__ status, unsigned* __restrict__ flag,
                      const int2* __restrict__ src, size_t plane) {
  bulk_accumulate(x, y, xstride, ystride, n, ped, w0, log2e, nwin, sX, sZZ, status, flag, src, plane,
                  (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// Kernel A', latency variant for small batches (upper tree levels, short chains): 2^LOG_L lanes
// share one hash.  Each lane sums its contiguous block of 2*nwin / L table entries (mmadd, then
// madd), the partial sums are combined with log2(L) butterfly rounds of wave shuffles + general
// XYZZ additions.  Latency drops from 31 dependent mixed additions (~300 field mul) to ~65 at
// L = 8, at the price of ~1.7x more total work - used only while the batch cannot fill the chip.
__device__ __forceinline__ fe shfl_xor_fe(const fe& v, int mask) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = __shfl_xor(v.l[i], mask, 64);
  return r;
}
// `width` (< 32) bits of a 256-bit operand in memory starting at `bit`; bits beyond 255 read as 0.
__device__ __forceinline__ uint32_t field_from_memory(const uint64_t* felt, int bit, int width) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(felt);
  const int wi = bit >> 5, sh = bit & 31;
  uint64_t two = (uint64_t)w[wi];
  if (wi + 1 < 8) two |= (uint64_t)w[wi + 1] << 32;
  return (uint32_t)(two >> sh) & ((1u << width) - 1u);
}
// Bits [start, start + width) of the string x || y straight from the two operands in memory.
__device__ __forceinline__ uint32_t window_from_memory(const uint64_t* fx, const uint64_t* fy, int start,
                                                       int width) {
  if (start + width <= 252) return field_from_memory(fx, start, width);
  if (start >= 252) return field_from_memory(fy, start - 252, width);
  const int lo_n = 252 - start;
  return field_from_memory(fx, start, lo_n) | (field_from_memory(fy, 0, width - lo_n) << lo_n);
}

// Partial sums of one lane group: returns X and ZZ of the full sum on every lane of the group (Y and
// ZZZ are not computed for the last combine).  Lane `sub`
// sums a contiguous run of windows (the first nwin % L lanes take one more than the others; the
// host guarantees at least two per lane).
template <int LOG_L>
__device__ __forceinline__ xyzz split_accumulate(const uint64_t* fx, const uint64_t* fy, int sub,
                                                 const aff_packed* __restrict__ ped, int w0, int log2e,
                                                 int nwin) {
  constexpr int L = 1 << LOG_L;
  const int cnt_lo = nwin / L, extra = nwin % L;
  const int cnt = cnt_lo + (sub < extra ? 1 : 0);
  const int g0 = sub * cnt_lo + (sub < extra ? sub : extra);
  auto entry = [&](int g) {
    return window_entry(ped, g, window_from_memory(fx, fy, window_start(g, w0, log2e), window_width(g, w0, log2e)),
                        w0, log2e);
  };
  const window_ref r0 = entry(g0);
  const aff q0 = signed_aff(ld_raw(r0.entry), r0.negative);
  window_ref rn = entry(g0 + 1);
  raw_aff nxt = ld_raw(rn.entry);
  xyzz acc;
  {
    const aff q1 = signed_aff(nxt, rn.negative);
    if (cnt > 2) {
      rn = entry(g0 + 2);
      nxt = ld_raw(rn.entry);
    }
    acc = xyzz_mmadd(q0, q1);
  }
  for (int j = 2; j < cnt; ++j) {
    const aff q = signed_aff(nxt, rn.negative);
    if (j + 1 < cnt) {
      rn = entry(g0 + j + 1);
      nxt = ld_raw(rn.entry);
    }
    acc = xyzz_madd(acc, q);
  }
  // NOT unrolled on purpose: one copy of the 14-multiplication general addition keeps the kernel
  // inside the instruction cache (a 58 KB straight-line body ran 2x slower than a 25 KB loop).
#pragma unroll 1
  for (int r = 0; r + 1 < LOG_L; ++r) {
    xyzz o;
    o.X = shfl_xor_fe(acc.X, 1 << r);
    o.Y = shfl_xor_fe(acc.Y, 1 << r);
    o.ZZ = shfl_xor_fe(acc.ZZ, 1 << r);
    o.ZZZ = shfl_xor_fe(acc.ZZZ, 1 << r);
    acc = xyzz_add(acc, o);
  }
  if constexpr (LOG_L > 0) {  // last round: only x = X / ZZ of the total is wanted (Y, ZZZ are left stale)
    constexpr int r = LOG_L - 1;
    xyzz o;
    o.X = shfl_xor_fe(acc.X, 1 << r);
    o.Y = shfl_xor_fe(acc.Y, 1 << r);
    o.ZZ = shfl_xor_fe(acc.ZZ, 1 << r);
    o.ZZZ = shfl_xor_fe(acc.ZZZ, 1 << r);
    fe X3, ZZ3;
    xyzz_add_x_only(acc, o, X3, ZZ3);
    acc.X = X3;
    acc.ZZ = ZZ3;
  }
  return acc;
}

// FUSED: lane 0 of each group also inverts ZZ and writes the affine x itself - no scratch planes and
// no second launch.  Used only while the whole level fits one wave per SIMD (the inversion then costs
// latency, not throughput): it saves the launch gap and the plane round trip of the small levels.
template <int LOG_L, bool FUSED>
__global__ void __launch_bounds__(256)
ped_accumulate_split_kernel(const uint64_t* __restrict__ x, const uint64_t* __restrict__ y, size_t xstride,
                            size_t ystride, size_t n, const aff_packed* __restrict__ ped, int w0, int log2e,
                            int nwin, int32_t* __restrict__ sX, int32_t* __restrict__ sZZ,
                            uint8_t* __restrict__ status, unsigned* __restrict__ flag,
                            const int2* __restrict__ src, uint64_t* __restrict__ out, size_t ostride,
                            size_t plane) {
  constexpr int L = 1 << LOG_L;
  const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t e_raw = gt >> LOG_L;
  const int sub = (int)(gt & (L - 1));
  const bool active = e_raw < n;
  const size_t e = active ? e_raw : n - 1;  // clamp: whole lane groups stay convergent for the shuffles
  const uint64_t *fx, *fy;
  operand_pointers(x, y, xstride, ystride, src, e, fx, fy);
  const xyzz acc = split_accumulate<LOG_L>(fx, fy, sub, ped, w0, log2e, nwin);
  uint8_t st = SP_HASH_OK;
  u256 xa_plain;
  if (FUSED) {
    // EVERY lane takes part in the inversion (an inversion under a 1-lane-in-8 execution mask takes 4x as
    // long as the same code with all lanes active, tools/ubench/inv_lanes.hip), and a DPP quad always
    // shares ONE quad-split divsteps run: its lanes hold one sum (L >= 4), two (L = 2) or four different
    // ones (L = 1), multiplied together first and separated afterwards (quad.hpp fe_inv_shared_quad).
    fe zz = acc.ZZ;
    // ZZ = 0: an exceptional addition happened (signature.py:313 territory).  Lanes that SHARE an inversion with
    // other sums (L < 4) must find that out first - a zero would spoil the product of the quad; where the quad
    // holds one sum the test costs nothing: the inversion answers 0 for a multiple of p, and only for one.
    if constexpr (LOG_L < 2) {
      if (fe_is_zero(zz)) {
        zz = FE_ONE_M;
        st = SP_HASH_UNHASHABLE;
      }
    }
    // the inverse without its Montgomery factor: X R x ZZ^-1 is already the plain x (no fe_from_mont pass)
    const fe zinv = fe_inv_shared_quad<(LOG_L >= 2 ? 0 : 2 - LOG_L), true>(zz, (int)(threadIdx.x & 3));
    if constexpr (LOG_L >= 2) {
      if (limbs_is_zero(zinv)) st = SP_HASH_UNHASHABLE;
    }
    xa_plain = fe_pack(fe_canon(fe_mul(acc.X, zinv)));
  }
  if (!active || sub != 0) return;
  if (!u256_lt(ld_u256(fx), U256_P) || !u256_lt(ld_u256(fy), U256_P)) st = SP_HASH_OUT_OF_RANGE;
  if (FUSED) {
    st_u256(out + 4 * e * ostride, xa_plain);
  } else {
    store_limbs(sX, plane, e, acc.X);
    store_limbs(sZZ, plane, e, acc.ZZ);
  }
  if (status) status[e] = st;
  if (st != SP_HASH_OK && flag) atomicOr(flag, (unsigned)st);
}

// Kernel A + A' in ONE launch (a level that is not a whole number of waves per SIMD): the first
// `bulk_blocks` workgroups run the one-lane-per-hash body on the first n_bulk hashes (whole rounds of 65 536),
// the others the lane-split body (unfused: X, ZZ to the same scratch planes) on the remaining `rem` hashes.
// Both populations are resident together, so the remainder's short chain fills issue slots beside the bulk
// waves instead of costing a launch of its own at one latency-bound wave per SIMD (measured for 163 840 and
// 81 920 hashes, profiles/r03_levels_forest_20_fused.txt).
template <int LOG_L>
__global__ void __launch_bounds__(256, SP_ACC_WAVES)
ped_accumulate_mixed_kernel(const uint64_t* __restrict__ x, const uint64_t* __restrict__ y, size_t xstride,
                            size_t ystride, size_t n_bulk, size_t rem, unsigned bulk_blocks,
                            const aff_packed* __restrict__ ped, int w0, int log2e, int nwin,
                            int32_t* __restrict__ sX, int32_t* __restrict__ sZZ, uint8_t* __restrict__ status,
                            unsigned* __restrict__ flag, const int2* __restrict__ src, size_t plane) {
  if (blockIdx.x < bulk_blocks) {
    bulk_accumulate(x, y, xstride, ystride, n_bulk, ped, w0, log2e, nwin, sX, sZZ, status, flag, src, plane,
                    (size_t)blockIdx.x * blockDim.x + threadIdx.x);
    return;
  }
  constexpr int L = 1 << LOG_L;
  const size_t gt = (size_t)(blockIdx.x - bulk_blocks) * blockDim.x + threadIdx.x;
  const size_t e_raw = gt >> LOG_L;
  const int sub = (int)(gt & (L - 1));
  const bool active = e_raw < rem;
  const size_t e = n_bulk + (active ? e_raw : rem - 1);  // clamp: whole lane groups stay convergent
  const uint64_t *fx, *fy;
  operand_pointers(x, y, xstride, ystride, src, e, fx, fy);
  const xyzz acc = split_accumulate<LOG_L>(fx, fy, sub, ped, w0, log2e, nwin);
  if (!active || sub != 0) return;
  uint8_t st = SP_HASH_OK;
  if (!u256_lt(ld_u256(fx), U256_P) || !u256_lt(ld_u256(fy), U256_P)) st = SP_HASH_OUT_OF_RANGE;
  store_limbs(sX, plane, e, acc.X);
  store_limbs(sZZ, plane, e, acc.ZZ);
  if (status) status[e] = st;
  if (st != SP_HASH_OK && flag) atomicOr(flag, (unsigned)st);
}

// Kernel Q, the latency path of the small levels: 4 * QUADS lanes (QUADS = 2, 4 or 8 DPP quads) share one
// hash.  Window g of the plan belongs to quad g mod QUADS.  A quad adds its first four windows pairwise
// (two affine sums at once on its two lane pairs, then one quad-parallel addition), its further windows
// one at a time (each a quad-parallel addition with an affine P2), and log2(QUADS) butterfly rounds over
// lane ^ 4, ^ 8, ^ 16 (ds_swizzle, no LDS memory) leave the total on every quad.  With 8 quads and
// 16 <= nwin <= 32 that is 3 + 4 + 4 + 4 + 3 = 18 rounds of one field multiplication instead of the ~62
// dependent multiplications of the 8-lane split kernel; 4 quads take the same 18 rounds up to 20
// windows.  Every lane then inverts ZZ (all copies are identical; a wave with few active lanes runs the
// same code 4x slower, tools/ubench/inv_lanes.hip) and lane 0 writes the affine x.  One launch per
// level, no scratch.  Requires nwin >= 2 * QUADS (every quad owns at least one pair).
//
// SPARSE (levels of a sparse multi-update, merkle.hip): a node with ONE touched child has the level's
// empty-subtree root as its other operand - a constant.  The windows that lie wholly inside that operand's half of
// x || y (the first `cx` windows for a constant left operand, the last `cy` for a constant right one) always select
// the same entries, whose sum is tabulated once per level and side (`cpts`: [0] left, [1] right;
// ped_partial_kernel).  Such a node sums 1 + nwin - cx (or - cy) points instead of nwin - 11 instead of 19 with
// 26-bit windows: a tree sum of depth 4 instead of 5, 14 instead of 18 rounds.  The window list is per lane group
// ("virtual" window v: 0 = the constant point, v >= 1 = real window cx + v - 1, resp. v - 1), the loop bound is the
// largest of the wave.  Requires 1 + nwin - max(cx, cy) >= 2 * QUADS.
// The hash of one lane group (see ped_quad_kernel): fx / fy = the two operands (global memory, or LDS through a
// generic pointer), mode = 0, or 1 / 2 when the left / right operand is the level's constant whose point is cpts[0] /
// cpts[1]; g = lane within the group.  Returns the plain affine x on EVERY lane of the group; *unhashable is set
// when the sum met an exceptional addition (signature.py:313).
template <int LOG_Q, bool SPARSE>
__device__ __forceinline__ u256 quad_hash(const uint64_t* fx, const uint64_t* fy, int mode, const aff_packed* __restrict__ cpts,
                                          int cx, int cy, const aff_packed* __restrict__ ped, int w0, int log2e,
                                          int nwin_plan, int g, bool* unhashable) {
  constexpr int QUADS = 1 << LOG_Q;
  const int q = g >> 2, k = g & 3;
  int nwin = nwin_plan, first_real = 0;
  if constexpr (SPARSE) {
    nwin = mode == 0 ? nwin_plan : 1 + nwin_plan - (mode == 1 ? cx : cy);
    first_real = mode == 1 ? cx - 1 : (mode == 2 ? -1 : 0);  // real window of virtual window v >= 1: first_real + v
  }
  auto entry = [&](int v) {
    if constexpr (SPARSE) {
      if (mode != 0 && v == 0) return unpack_raw(ld_raw(cpts + (mode - 1)));
    }
    const int w = first_real + v;
    const window_ref r =
        window_entry(ped, w, window_from_memory(fx, fy, window_start(w, w0, log2e), window_width(w, w0, log2e)), w0, log2e);
    return signed_aff(ld_raw(r.entry), r.negative);
  };
  const int cnt = nwin / QUADS + (q < nwin % QUADS ? 1 : 0);  // windows of this quad: q, q + QUADS, q + 2 QUADS, ...
  int max_cnt = (nwin + QUADS - 1) / QUADS;
  if constexpr (SPARSE) {  // the loop below must run the same number of times on every lane of the wave
    max_cnt = 0;
    if (__any(mode == 0)) max_cnt = (nwin_plan + QUADS - 1) / QUADS;
    if (__any(mode == 1)) max_cnt = max(max_cnt, (1 + nwin_plan - cx + QUADS - 1) / QUADS);
    if (__any(mode == 2)) max_cnt = max(max_cnt, (1 + nwin_plan - cy + QUADS - 1) / QUADS);
  }
  // lanes 0,1: pair A = windows (q, q + QUADS); lanes 2,3: pair B = (q + 2 QUADS, q + 3 QUADS) where they exist
  const bool upper = (k & 2) != 0;
  const bool odd = (k & 1) != 0;
  const int wa = (upper && cnt >= 3) ? q + 2 * QUADS : q;
  const int wb = (upper && cnt >= 4) ? q + 3 * QUADS : q + QUADS;
  const aff pa = entry(wa), pb = entry(wb);
  qpt p = qmmadd(pa.x, pa.y, pb.x, pb.y, k);
  if (upper && cnt == 3) {  // a lone third window: P2 = that affine point
    p.a = odd ? pa.y : pa.x;
    p.b = FE_ONE_M;
  }
  qpt s = qadd<false>(p, k);
  if (cnt == 2) {  // no pair B: the quad's sum is P1 itself
    s.a = fe_dpp<quad_perm(0, 1, 0, 1)>(p.a);
    s.b = fe_dpp<quad_perm(0, 1, 0, 1)>(p.b);
  }
  for (int j = 4; j < max_cnt; ++j) {  // further windows of the quad, one quad-parallel addition each
    const aff pj = entry(j < cnt ? q + j * QUADS : q);
    qpt t;
    t.a = fe_sel(upper, odd ? pj.y : pj.x, s.a);
    t.b = fe_sel(upper, FE_ONE_M, s.b);
    const qpt u = qadd<false>(t, k);
    if (j < cnt) s = u;
  }
  // butterfly over the quads: own sum is P1, the partner's P2; the last round yields X3 in .a and
  // ZZ3 in .b of every lane
#define SP_BUTTERFLY(XORV, LAST)                               \
  {                                                            \
    qpt t;                                                     \
    t.a = fe_sel(upper, fe_swizzle_xor<XORV>(s.a), s.a);       \
    t.b = fe_sel(upper, fe_swizzle_xor<XORV>(s.b), s.b);       \
    s = qadd<LAST>(t, k);                                      \
  }
  if constexpr (LOG_Q == 1) SP_BUTTERFLY(4, true)
  if constexpr (LOG_Q == 2) { SP_BUTTERFLY(4, false) SP_BUTTERFLY(8, true) }
  if constexpr (LOG_Q == 3) { SP_BUTTERFLY(4, false) SP_BUTTERFLY(8, false) SP_BUTTERFLY(16, true) }
#undef SP_BUTTERFLY
  // ZZ = 0 - an exceptional addition (signature.py:313 territory) - shows as a zero inverse: the inversion answers
  // 0 for a multiple of p and only for one, so no test of ZZ sits on the chain in front of it
  const fe zinv = fe_inv_quad<true>(s.b, k);  // plain-form inverse: no fe_from_mont
  *unhashable = limbs_is_zero(zinv);
  return fe_pack(fe_canon(fe_mul(s.a, zinv)));
}

template <int LOG_Q, bool SPARSE = false>
__global__ void __launch_bounds__(256)
ped_quad_kernel(const uint64_t* __restrict__ x, const uint64_t* __restrict__ y, size_t xstride, size_t ystride,
                size_t n, const aff_packed* __restrict__ ped, int w0, int log2e, int nwin_plan,
                uint8_t* __restrict__ status, unsigned* __restrict__ flag, const int2* __restrict__ src,
                uint64_t* __restrict__ out, size_t ostride, int dup, const aff_packed* __restrict__ cpts, int cx,
                int cy) {
  constexpr int QUADS = 1 << LOG_Q, LANES = 4 * QUADS;
  const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // dup: 2^dup lane groups compute the same hash, so that a wave holds ONE value while the level is small
  // enough anyway (the variable-time inversion runs as long as the slowest value of the wave)
  const size_t e_raw = gt / ((size_t)LANES << dup);
  const int g = (int)(gt % LANES);
  const bool active = e_raw < n;
  const size_t e = active ? e_raw : n - 1;  // clamp: whole groups stay convergent for the lane exchanges
  const uint64_t *fx, *fy;
  operand_pointers(x, y, xstride, ystride, src, e, fx, fy);
  int mode = 0;  // mode 1 / 2: the left / right operand is the level's constant
  if constexpr (SPARSE) {
    const int2 sc = src[e];
    mode = sc.x < 0 ? 1 : (sc.y < 0 ? 2 : 0);
  }
  bool unhashable;
  const u256 xa_plain = quad_hash<LOG_Q, SPARSE>(fx, fy, mode, cpts, cx, cy, ped, w0, log2e, nwin_plan, g, &unhashable);
  uint8_t st = unhashable ? SP_HASH_UNHASHABLE : SP_HASH_OK;
  if (!active || g != 0) return;
  if (!u256_lt(ld_u256(fx), U256_P) || !u256_lt(ld_u256(fy), U256_P)) st = SP_HASH_OUT_OF_RANGE;
  st_u256(out + 4 * e * ostride, xa_plain);
  if (status) status[e] = st;
  if (st != SP_HASH_OK && flag) atomicOr(flag, (unsigned)st);
}

// Levels of a sparse multi-update whose paths do not merge (every node has exactly one touched child: the 40-odd
// lowest levels of a height-64 update over random keys) as ONE launch: a lane group follows its path from level to
// level, the node value it has just computed - present on every lane of the group - goes to the group's LDS slot
// and is the next level's touched operand; the other operand (a sibling the lookup kernel copied from the tree's
// table, or the level's empty-subtree root) and the child lists are independent of the chain.  Every node value is
// also written to `felts` for the insertion into the table.  Saves the launch boundary and the store -> load round
// trip through HBM between two levels.  pl.val_base / pl.src_off are merkle.hip's TreeLevels fields; level
// `pl.first` is the children's level of the first hash.
struct PathLevels {
  int first, n_levels;
  int val_base[66];
  unsigned src_off[66];
};
template <int LOG_Q, bool SPARSE>
__global__ void __launch_bounds__(256)
ped_path_kernel(uint64_t* __restrict__ felts, const uint64_t* __restrict__ emp, size_t n,
                const aff_packed* __restrict__ ped, int w0, int log2e, int nwin_plan, unsigned* __restrict__ flag,
                const int2* __restrict__ src_all, PathLevels pl, int dup, const aff_packed* __restrict__ cpts_tree,
                int cx, int cy) {
  constexpr int QUADS = 1 << LOG_Q, LANES = 4 * QUADS;
  __shared__ uint64_t slot[256 / LANES][4];
  const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t e_raw = gt / ((size_t)LANES << dup);
  const int g = (int)(gt % LANES), grp = (int)(threadIdx.x / LANES);
  const bool active = e_raw < n;
  const size_t e = active ? e_raw : n - 1;
  for (int i = 0; i < pl.n_levels; ++i) {
    const int l = pl.first + i;  // the children's level; the node computed here is node e of level l + 1
    const int2 sc = src_all[pl.src_off[l] + e];
    const uint64_t* y = emp + 4 * l;
    const uint64_t* fx = sc.x >= 0 ? felts + 4 * (size_t)sc.x : y;
    const uint64_t* fy = sc.y >= 0 ? felts + 4 * (size_t)sc.y : y;
    if (i > 0) {  // the touched child is the node this group computed one level below: node e of level l
      if (sc.x == pl.val_base[l] + (int)e) fx = slot[grp];
      else fy = slot[grp];
    }
    int mode = 0;
    if constexpr (SPARSE) mode = sc.x < 0 ? 1 : (sc.y < 0 ? 2 : 0);
    bool unhashable;
    const u256 xa = quad_hash<LOG_Q, SPARSE>(fx, fy, mode, cpts_tree ? cpts_tree + 2 * l : nullptr, cx, cy, ped, w0, log2e,
                                             nwin_plan, g, &unhashable);
    uint8_t st = unhashable ? SP_HASH_UNHASHABLE : SP_HASH_OK;
    // only the first level can see a caller's value (the new leaves); above it the operands are hash outputs,
    // table values and empty-subtree roots
    if (i == 0 && g == 0 && (!u256_lt(ld_u256(fx), U256_P) || !u256_lt(ld_u256(fy), U256_P))) st = SP_HASH_OUT_OF_RANGE;
    __syncthreads();  // every lane of the block has taken its windows of this level
    if (g == 0) {
      uint32_t* sl = reinterpret_cast<uint32_t*>(slot[grp]);
#pragma unroll
      for (int w = 0; w < 8; ++w) sl[w] = xa.w[w];
      if (active) {
        st_u256(felts + 4 * ((size_t)pl.val_base[l + 1] + e), xa);
        if (st != SP_HASH_OK && flag) atomicOr(flag, (unsigned)st);
      }
    }
    __syncthreads();
  }
}

// Hash chains as ONE launch: lane group e folds its chain h <- H(h, w_j) (or H(w_j, h): h_right) over `steps`
// words, the running hash - present on every lane of the group - handed on through the group's LDS slot; the
// words are read from HBM ahead of the chain.  first + 4 e = the chain's start value, words + 4 e its first word,
// word_stride = felts between consecutive words of one chain (the word-major layout of sp_pedersen_chains_dev:
// `width`; a single chain folded from its last element down: -1).  A 4096-order batch's message hashes are three
// such steps, a program-hash chain (sp_pedersen_chain_right) thousands: the launch boundary and the HBM round trip
// of every step go.
template <int LOG_Q>
__global__ void __launch_bounds__(256)
ped_chain_kernel(const uint64_t* __restrict__ first, const uint64_t* __restrict__ words, long long word_stride, size_t n,
                 int steps, bool h_right, const aff_packed* __restrict__ ped, int w0, int log2e, int nwin_plan,
                 unsigned* __restrict__ flag, uint64_t* __restrict__ out, int dup) {
  constexpr int QUADS = 1 << LOG_Q, LANES = 4 * QUADS;
  __shared__ uint64_t slot[256 / LANES][4];
  const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t e_raw = gt / ((size_t)LANES << dup);
  const int g = (int)(gt % LANES), grp = (int)(threadIdx.x / LANES);
  const bool active = e_raw < n;
  const size_t e = active ? e_raw : n - 1;
  const uint64_t* h = first + 4 * e;
  const uint64_t* w = words + 4 * e;
  for (int j = 0; j < steps; ++j, w += 4 * word_stride) {
    const uint64_t* fx = h_right ? w : h;
    const uint64_t* fy = h_right ? h : w;
    bool unhashable;
    const u256 xa = quad_hash<LOG_Q, false>(fx, fy, 0, nullptr, 0, 0, ped, w0, log2e, nwin_plan, g, &unhashable);
    uint8_t st = unhashable ? SP_HASH_UNHASHABLE : SP_HASH_OK;
    // every word is a caller's value; the running hash is one only at the first step
    if (g == 0 && (!u256_lt(ld_u256(w), U256_P) || (j == 0 && !u256_lt(ld_u256(h), U256_P)))) st = SP_HASH_OUT_OF_RANGE;
    __syncthreads();
    if (g == 0) {
      uint32_t* sl = reinterpret_cast<uint32_t*>(slot[grp]);
#pragma unroll
      for (int k = 0; k < 8; ++k) sl[k] = xa.w[k];
      if (active) {
        if (j == steps - 1) st_u256(out + 4 * e, xa);
        if (st != SP_HASH_OK && flag) atomicOr(flag, (unsigned)st);
      }
    }
    __syncthreads();
    h = slot[grp];
  }
}

// The small levels of a DENSE forest (at most 2048 hashes per level: the eight-quad size class) as one launch per
// FOUR levels: a block of 8 lane groups (256 threads = one wave per SIMD of its CU: two blocks' waves on one SIMD
// would halve the speed of both chains) takes 2^L consecutive nodes of level j (L <= 4 levels remain inside every
// tree) and hashes them down to one node of level j + L - 8, 4, 2, 1 hashes - handing every node value on through
// LDS (two slot arrays, written and read alternately: one barrier per level) and writing it to its place in the
// level-major buffer.  At a level of c < 8 hashes the 8 / c groups that share a hash all compute it (the `dup` idiom of
// ped_quad_kernel: a wave holds ONE value for the variable-time inversion); the first of them writes.  Same chain per
// level as ped_quad_kernel<3>, without the launch boundary and the HBM round trip between levels.
__global__ void __launch_bounds__(256)
ped_top_kernel(uint64_t* __restrict__ cur, size_t n_in, int n_levels, const aff_packed* __restrict__ ped, int w0, int log2e,
               int nwin_plan, unsigned* __restrict__ flag) {
  constexpr int LANES = 32, GROUPS = 256 / LANES;  // 8
  __shared__ uint64_t slot[2][GROUPS][4];
  const int g = (int)(threadIdx.x % LANES), grp = (int)(threadIdx.x / LANES);
  const size_t b = blockIdx.x;
  const int chunk = 1 << n_levels;  // input nodes of this block (<= 16)
  uint64_t* out = cur + 4 * n_in;   // level j + 1
  size_t n_out = n_in >> 1;
  int share = 4 - n_levels;  // log2 of the groups per hash at the first level
  for (int t = 0; t < n_levels; ++t, ++share) {
    const int cnt = chunk >> (t + 1);  // hashes of this block at this level = GROUPS >> share
    const int i = grp >> share;
    const bool writer = g == 0 && (grp & ((1 << share) - 1)) == 0;
    const uint64_t *fx, *fy;
    if (t == 0) {
      fx = cur + 4 * ((size_t)chunk * b + 2 * (size_t)i);
      fy = fx + 4;
    } else {
      fx = slot[(t - 1) & 1][2 * i];
      fy = slot[(t - 1) & 1][2 * i + 1];
    }
    bool unhashable;
    const u256 xa = quad_hash<3, false>(fx, fy, 0, nullptr, 0, 0, ped, w0, log2e, nwin_plan, g, &unhashable);
    uint8_t st = unhashable ? SP_HASH_UNHASHABLE : SP_HASH_OK;
    if (t == 0 && writer && (!u256_lt(ld_u256(fx), U256_P) || !u256_lt(ld_u256(fy), U256_P))) st = SP_HASH_OUT_OF_RANGE;
    if (writer) {
      uint32_t* sl = reinterpret_cast<uint32_t*>(slot[t & 1][i]);
#pragma unroll
      for (int w = 0; w < 8; ++w) sl[w] = xa.w[w];
      st_u256(out + 4 * ((size_t)cnt * b + (size_t)i), xa);
      if (st != SP_HASH_OK && flag) atomicOr(flag, (unsigned)st);
    }
    __syncthreads();
    out += 4 * n_out;
    n_out >>= 1;
  }
}

// Kernel B: thread t owns elements t, t+T, t+2T, ...; one inversion per thread (Montgomery's trick).
// Launched with at most one wave per SIMD when it can be (finish_threads), so nothing hides a load:
// both passes are software-pipelined by hand (the operands of element j +- 1 are requested before
// element j is multiplied), the loops are NOT unrolled (a fully unrolled K = 12 body exceeds the
// instruction cache and ran 15 % slower than this loop), and the per-element zero test is replaced by
// one test of the thread's total product: a zero ZZ - an exceptional addition, signature.py:313
// territory - makes the total zero, and only then the thread re-walks its elements.
__global__ void __launch_bounds__(256)
ped_finish_kernel(const int32_t* __restrict__ sX, int32_t* __restrict__ sZZ, int32_t* __restrict__ sPre,
                  size_t n, size_t T, uint64_t* __restrict__ out, size_t ostride,
                  uint8_t* __restrict__ status, unsigned* __restrict__ flag) {
  // T is a multiple of 4 (whole DPP quads); threads t >= n own nothing and only lend their lane to the
  // quad's shared inversion
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const size_t cnt = t < n ? (n - t + T - 1) / T : 0;  // number of elements owned
  fe run = FE_ONE_M;
  if (cnt > 0) {
    fe znext = load_limbs(sZZ, n, t);
#pragma unroll 1
    for (size_t j = 0; j < cnt; ++j) {
      const size_t e = t + j * T;
      const fe z = znext;
      if (j + 1 < cnt) znext = load_limbs(sZZ, n, e + T);
      store_limbs(sPre, n, e, run);
      run = fe_mul(run, z);
    }
  }
  if (fe_is_zero(run)) {  // rare: find the zero factors, flag them, and redo the products without them
    run = FE_ONE_M;
#pragma unroll 1
    for (size_t e = t; e < n; e += T) {
      fe z = load_limbs(sZZ, n, e);
      if (fe_is_zero(z)) {
        z = FE_ONE_M;
        store_limbs(sZZ, n, e, z);
        if (status) status[e] = SP_HASH_UNHASHABLE;
        if (flag) atomicOr(flag, (unsigned)SP_HASH_UNHASHABLE);
      }
      store_limbs(sPre, n, e, run);
      run = fe_mul(run, z);
    }
  }
  // the last element's operands travel while the inversion runs
  size_t e = cnt > 0 ? t + (cnt - 1) * T : 0;
  fe zn = run, pn = run, xn = run;
  if (cnt > 0) {
    zn = load_limbs(sZZ, n, e);
    pn = load_limbs(sPre, n, e);
    xn = load_limbs(sX, n, e);
  }
  // ONE divsteps run per DPP quad: the four threads' totals are multiplied together, inverted once with the
  // quad-split inversion and separated again (quad.hpp) - 8 k instead of 13 k instructions per lane
  // ... and without the Montgomery factor: inv stays "plain" through inv * z, so zinv = inv * pre is the plain
  // 1 / ZZ and X R x zinv the plain x - one reduction pass (fe_from_mont) less per hash
  fe inv = fe_inv_shared_quad<2, true>(run, (int)(threadIdx.x & 3));
#pragma unroll 1
  for (size_t j = cnt; j-- > 0;) {
    const fe z = zn, pre = pn, xv = xn;
    const size_t cur = t + j * T;
    if (j > 0) {
      zn = load_limbs(sZZ, n, cur - T);
      pn = load_limbs(sPre, n, cur - T);
      xn = load_limbs(sX, n, cur - T);
    }
    const fe zinv = fe_mul(inv, pre);
    inv = fe_mul(inv, z);
    const fe xa = fe_mul(xv, zinv);  // X / ZZ itself (xv carries the only Montgomery factor, the product removes it)
    st_u256(out + 4 * cur * ostride, fe_pack(fe_canon(xa)));
  }
}

// Kernel B with the prefix products (and, while they fit, the ZZ values) of a thread kept in LDS instead of the
// sPre plane in HBM (round 5).  The finish launch of a big level is bound by its traffic, not by its arithmetic:
// per hash it read ZZ twice, wrote and re-read the prefix product and read X - 212 B against 921 instructions,
// 139 MB in 42 us for the 655 360 hashes of level 0 of the driver's forest (3.3 TB/s).  A block of 256 threads owns
// the whole LDS of its CU here (the launch is one wave per SIMD by design, finish_threads), so K <= 16 prefix products
// per thread fit (K x 9 KiB), and K <= 8 of prefix + ZZ: 104 B per hash are left (ZZ, X, out).  Dynamic indexing of
// LDS keeps the loops rolled (the unrolled register-resident form overflows the instruction cache, see above).
template <bool Z_IN_LDS>
__global__ void __launch_bounds__(256)
ped_finish_lds_kernel(const int32_t* __restrict__ sX, int32_t* __restrict__ sZZ, size_t n, size_t T,
                      uint64_t* __restrict__ out, size_t ostride, uint8_t* __restrict__ status,
                      unsigned* __restrict__ flag, int kmax) {
  extern __shared__ int32_t fin_lds[];  // pre: [kmax][9][256], then (Z_IN_LDS) z: [kmax][9][256]
  int32_t* lpre = fin_lds;
  int32_t* lz = fin_lds + (size_t)kmax * NL * 256;
  auto lds_put = [&](int32_t* base, size_t j, const fe& v) {
#pragma unroll
    for (int k = 0; k < NL; ++k) base[((j * NL + k) << 8) + threadIdx.x] = v.l[k];
  };
  auto lds_get = [&](const int32_t* base, size_t j) {
    fe v;
#pragma unroll
    for (int k = 0; k < NL; ++k) v.l[k] = base[((j * NL + k) << 8) + threadIdx.x];
    return v;
  };
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const size_t cnt = t < n ? (n - t + T - 1) / T : 0;  // number of elements owned (<= kmax: the host checked)
  fe run = FE_ONE_M;
  if (cnt > 0) {
    fe znext = load_limbs(sZZ, n, t);
#pragma unroll 1
    for (size_t j = 0; j < cnt; ++j) {
      const fe z = znext;
      if (j + 1 < cnt) znext = load_limbs(sZZ, n, t + (j + 1) * T);
      lds_put(lpre, j, run);
      if (Z_IN_LDS) lds_put(lz, j, z);
      run = fe_mul(run, z);
    }
  }
  if (fe_is_zero(run)) {  // rare: find the zero factors, flag them, and redo the products without them
    run = FE_ONE_M;
    size_t j = 0;
#pragma unroll 1
    for (size_t e = t; e < n; e += T, ++j) {
      fe z = load_limbs(sZZ, n, e);
      if (fe_is_zero(z)) {
        z = FE_ONE_M;
        store_limbs(sZZ, n, e, z);
        if (status) status[e] = SP_HASH_UNHASHABLE;
        if (flag) atomicOr(flag, (unsigned)SP_HASH_UNHASHABLE);
      }
      lds_put(lpre, j, run);
      if (Z_IN_LDS) lds_put(lz, j, z);
      run = fe_mul(run, z);
    }
  }
  size_t e = cnt > 0 ? t + (cnt - 1) * T : 0;
  fe zn = run, xn = run;
  if (cnt > 0) {
    if (!Z_IN_LDS) zn = load_limbs(sZZ, n, e);
    xn = load_limbs(sX, n, e);
  }
  fe inv = fe_inv_shared_quad<2, true>(run, (int)(threadIdx.x & 3));
#pragma unroll 1
  for (size_t j = cnt; j-- > 0;) {
    const fe z = Z_IN_LDS ? lds_get(lz, j) : zn, xv = xn;
    const fe pre = lds_get(lpre, j);
    const size_t cur = t + j * T;
    if (j > 0) {
      if (!Z_IN_LDS) zn = load_limbs(sZZ, n, cur - T);
      xn = load_limbs(sX, n, cur - T);
    }
    const fe zinv = fe_mul(inv, pre);
    inv = fe_mul(inv, z);
    const fe xa = fe_mul(xv, zinv);
    st_u256(out + 4 * cur * ostride, fe_pack(fe_canon(xa)));
  }
}

// Full affine point (x, y) per item - pedersen_hash_as_point (signature.py:300-318), a testing
// helper in the reference; one thread does its own inversion.
__global__ void __launch_bounds__(128)
ped_point_kernel(const uint64_t* __restrict__ x, const uint64_t* __restrict__ y, size_t n,
                 const aff_packed* __restrict__ ped, int w0, int log2e, int nwin,
                 uint64_t* __restrict__ ox, uint64_t* __restrict__ oy, uint8_t* __restrict__ status) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) e = n - 1;  // redundant copy of the last item (few active lanes are slow, see ecdsa.hip)
  u256 sx = ld_u256(x + 4 * e);
  u256 sy = ld_u256(y + 4 * e);
  if (!u256_lt(sx, U256_P) || !u256_lt(sy, U256_P)) {
    status[e] = SP_HASH_OUT_OF_RANGE;
    return;
  }
  bits512 str = concat_xy(sx, sy);
  xyzz acc = xyzz_from_aff(ld_aff(ped + pop_bits(str, w0)));
  for (int g = 1; g < nwin; ++g) {
    const window_ref r = window_entry(ped, g, pop_bits(str, log2e + 1), w0, log2e);
    acc = xyzz_madd(acc, signed_aff(ld_raw(r.entry), r.negative));
  }
  if (fe_is_zero(acc.ZZ)) {
    status[e] = SP_HASH_UNHASHABLE;
    return;
  }
  const fe izzz = fe_inv(acc.ZZZ);
  st_u256(ox + 4 * e, fe_pack(fe_from_mont(fe_mul(acc.X, fe_sqr(fe_mul(acc.ZZ, izzz))))));
  st_u256(oy + 4 * e, fe_pack(fe_from_mont(fe_mul(acc.Y, izzz))));
  status[e] = SP_HASH_OK;
}

// The constant points of the SPARSE quad kernel: for felt e of `felts` (the empty-subtree root of level e) the sum
// of the table entries its first `cx` windows select when it is the LEFT operand (out[2 e]) and of the entries its
// last `cy` windows select when it is the RIGHT operand (out[2 e + 1]) - affine, Montgomery form, packed like a
// table entry.  One thread per point, its own inversion: 2 x 64 points per empty leaf, computed once per tree.
__global__ void __launch_bounds__(64)
ped_partial_kernel(const uint64_t* __restrict__ felts, int count, const aff_packed* __restrict__ ped, int w0, int log2e,
                   int nwin, int cx, int cy, aff_packed* __restrict__ out) {
  int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (t >= 2 * count) t = 2 * count - 1;  // redundant copy of the last item (few active lanes are slow)
  const int side = t & 1;
  const uint64_t* f = felts + 4 * (size_t)(t >> 1);
  const int g0 = side == 0 ? 0 : nwin - cy, g1 = side == 0 ? cx : nwin;
  auto entry = [&](int w) {
    const window_ref r =
        window_entry(ped, w, window_from_memory(f, f, window_start(w, w0, log2e), window_width(w, w0, log2e)), w0, log2e);
    return signed_aff(ld_raw(r.entry), r.negative);
  };
  aff_packed res;
  if (g1 - g0 == 1) {
    const aff a = entry(g0);
    res.x = fe_pack(fe_canon(a.x));
    res.y = fe_pack(fe_canon(a.y));
  } else {
    xyzz acc = xyzz_mmadd(entry(g0), entry(g0 + 1));
    for (int g = g0 + 2; g < g1; ++g) acc = xyzz_madd(acc, entry(g));
    const fe izzz = fe_inv(acc.ZZZ);                       // Montgomery-form inverse
    const fe izz = fe_sqr(fe_mul(acc.ZZ, izzz));           // ZZ^2 / ZZZ^2 = 1 / ZZ
    res.x = fe_pack(fe_canon(fe_mul(acc.X, izz)));
    res.y = fe_pack(fe_canon(fe_mul(acc.Y, izzz)));
  }
  uint4* q = reinterpret_cast<uint4*>(out + t);
  q[0] = make_uint4(res.x.w[0], res.x.w[1], res.x.w[2], res.x.w[3]);
  q[1] = make_uint4(res.x.w[4], res.x.w[5], res.x.w[6], res.x.w[7]);
  q[2] = make_uint4(res.y.w[0], res.y.w[1], res.y.w[2], res.y.w[3]);
  q[3] = make_uint4(res.y.w[4], res.y.w[5], res.y.w[6], res.y.w[7]);
}

// ---- optional per-launch timing of the dominant kernel (bench.py roofline leg) -----------------
struct KernelProfile {
  bool enabled = false;
  std::vector<hipEvent_t> ev;  // pairs
  std::vector<size_t> units;
  size_t used = 0;
};
static KernelProfile g_prof;
static bool g_split_enabled = getenv("STARKPERP_NO_SPLIT") == nullptr;  // A/B switches
static bool g_fuse_enabled = getenv("STARKPERP_NO_FUSE") == nullptr;
static bool g_quad_enabled = getenv("STARKPERP_NO_QUAD") == nullptr;
static bool g_quad2_enabled = getenv("STARKPERP_NO_QUAD2") == nullptr;
static bool g_level_split = getenv("STARKPERP_NO_LEVEL_SPLIT") == nullptr;
static bool g_quad_no_dup = getenv("STARKPERP_QUAD_NO_DUP") != nullptr;
static bool g_sparse_enabled = getenv("STARKPERP_NO_SPARSE_LEVELS") == nullptr;  // constant points of sparse levels
static size_t g_quad_max = getenv("STARKPERP_QUAD_MAX") ? (size_t)atoll(getenv("STARKPERP_QUAD_MAX")) : 2048;
// ---- host-side drivers -------------------------------------------------------------------------
struct Scratch {
  int32_t *X, *ZZ, *Pre;
  unsigned* flag;
};
// Scratch is per stream, so independent calls issued on different HIP streams (e.g. several trees
// in flight) never share X / ZZ / prefix planes.  Growing a buffer reallocates it: callers that
// overlap streams should size the first call of each stream for their largest batch.
static std::map<StreamKey, DeviceBuffer> g_stream_scratch;
int get_scratch_public(size_t n, Scratch& s, hipStream_t st);
static int get_scratch(size_t n, Scratch& s, hipStream_t st) { return get_scratch_public(n, s, st); }
int get_scratch_public(size_t n, Scratch& s, hipStream_t st) {
  DeviceBuffer& buf = g_stream_scratch[stream_key(st)];
  const size_t plane = ((9 * n * sizeof(int32_t)) + 255) & ~(size_t)255;
  SP_HIP(buf.reserve(3 * plane + 256));
  char* b = (char*)buf.ptr;
  s.X = (int32_t*)b;
  s.ZZ = (int32_t*)(b + plane);
  s.Pre = (int32_t*)(b + 2 * plane);
  s.flag = (unsigned*)(b + 3 * plane);
  return SP_OK;
}

static size_t g_finish_lanes = getenv("STARKPERP_FINISH_LANES") ? (size_t)atoll(getenv("STARKPERP_FINISH_LANES")) : 65536;
// Lanes one round of the chip holds (a development knob).  The level plan cuts a level into n_bulk = a multiple of
// this many hashes for the one-lane-per-hash body, and ped_accumulate_mixed_kernel hands them out in blocks of 256:
// a value that is not a multiple of 256 would leave up to 255 hashes per level uncomputed, so the
// environment value is rounded up to one and kept inside [256, 2^24].
static size_t parse_split_lanes() {
  const char* e = getenv("STARKPERP_SPLIT_LANES");
  long long v = e ? atoll(e) : 65536;
  if (v < 256) v = 256;
  if (v > (1ll << 24)) v = 1ll << 24;
  return ((size_t)v + 255) & ~(size_t)255;
}
static size_t g_split_lanes = parse_split_lanes();
static bool g_finish_lds = getenv("STARKPERP_NO_FINISH_LDS") == nullptr;  // A/B switch
// The LDS form of the finish kernel takes up to 144 KiB of dynamic LDS: the limit is a per-function, per-DEVICE
// attribute, raised once for every device a context runs on (the caller holds the context lock).
static int finish_lds_ready() {
  static std::map<int, int> done;  // device -> SP_OK / SP_ERR_HIP
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return SP_ERR_HIP;
  auto it = done.find(dev);
  if (it != done.end()) return it->second;
  const int want = 16 * NL * 256 * (int)sizeof(int32_t);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ped_finish_lds_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, want);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(ped_finish_lds_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want);
  if (e != hipSuccess) (void)hipGetLastError();
  return done[dev] = (e == hipSuccess ? SP_OK : SP_ERR_HIP);
}
static size_t finish_threads(size_t n) {
  // K hashes share one inversion (Montgomery's trick, 3 multiplications per extra hash).  The
  // inversion is a ~14 k-instruction dependent chain of mostly 32-bit ops, and a SIMD is already
  // saturated by ~1.3 such waves: with 1 < waves/SIMD < 2 the launch takes twice as long as with one
  // wave per SIMD (measured: 81 920 inversions 68 us, 40 960 41 us).  So K is the smallest count that
  // keeps the launch within one wave per SIMD (65 536 lanes), up to 32.
  size_t K = (n + g_finish_lanes - 1) / g_finish_lanes;
  if (K < 1) K = 1;
  if (K > 32) K = 32;
  return ((n + K - 1) / K + 3) & ~(size_t)3;  // whole DPP quads: a quad shares one inversion
}

void release_pedersen_state() {
  for (auto& kv : g_stream_scratch) kv.second.release();
  g_stream_scratch.clear();
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.units.clear();
  g_prof.used = 0;
  g_prof.enabled = false;
}

// Windows of the plan that lie wholly inside the x half (the first cx) / the y half (the last cy) of x || y.
void constant_window_counts(const PedPlan& p, int& cx, int& cy) {
  cx = cy = 0;
  for (int g = 0; g < p.nwin; ++g) {
    if (p.start[g] + p.bits[g] <= 252) ++cx;
    if (p.start[g] >= 252) ++cy;
  }
}
// The constant points of a sparse tree's levels (ped_partial_kernel): out[2 l], out[2 l + 1] for felts[l].
// Returns SP_OK with *usable = false when the plan leaves no constant window on one of the sides.
int enqueue_partial_points(const uint64_t* felts, int count, aff_packed* out, hipStream_t st, bool* usable) {
  Context& c = ctx();
  int cx, cy;
  constant_window_counts(c.plan, cx, cy);
  *usable = cx >= 1 && cy >= 1 && count > 0;
  if (!*usable) return SP_OK;
  hipLaunchKernelGGL(ped_partial_kernel, dim3((unsigned)((2 * count + 63) / 64)), dim3(64), 0, st, felts, count, c.ped,
                     (int)c.plan.bits[0], c.plan.log2e, c.plan.nwin, cx, cy, out);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

static int enqueue_pedersen_impl(const uint64_t* x, size_t xs, const uint64_t* y, size_t ys, uint64_t* out,
                                 size_t os, uint8_t* status, unsigned* flag, size_t n, hipStream_t st,
                                 const Scratch& s, const int2* src, const aff_packed* cpts);
// Enqueue n hashes; x/y/out strides in felts.  `flag` (device, may be null) ORs item status.
int enqueue_pedersen(const uint64_t* x, size_t xs, const uint64_t* y, size_t ys, uint64_t* out,
                     size_t os, uint8_t* status, unsigned* flag, size_t n, hipStream_t st,
                     const Scratch& s, const int2* src) {
  return enqueue_pedersen_impl(x, xs, y, ys, out, os, status, flag, n, st, s, src, nullptr);
}
// A level of a sparse multi-update (gathered mode, src != null): `cpts` = the level's two constant points
// (enqueue_partial_points) or null.  Levels that fit the quad kernels take the SPARSE variant.
int enqueue_pedersen_sparse(const uint64_t* x, const uint64_t* y, uint64_t* out, unsigned* flag, size_t n,
                            hipStream_t st, const Scratch& s, const int2* src, const aff_packed* cpts) {
  return enqueue_pedersen_impl(x, 1, y, 1, out, 1, nullptr, flag, n, st, s, src, cpts);
}
static bool g_path_fusion = getenv("STARKPERP_NO_PATH_FUSION") == nullptr;  // A/B switch
static bool g_top_fusion = getenv("STARKPERP_NO_TOP_FUSION") == nullptr;    // A/B switch
static bool g_chain_fusion = getenv("STARKPERP_NO_CHAIN_FUSION") == nullptr;  // A/B switch
// n chains of `steps` hashes each as one launch (ped_chain_kernel); *done = false when n is not of a size class the
// quad kernels serve or there is a single step: the caller enqueues the steps one by one.
int enqueue_pedersen_chain(const uint64_t* first, const uint64_t* words, long long word_stride, size_t n, size_t steps,
                           bool h_right, uint64_t* out, unsigned* flag, hipStream_t st, bool* done) {
  *done = false;
  Context& c = ctx();
  const int w0 = c.plan.bits[0], log2e = c.plan.log2e, nwin = c.plan.nwin;
  if (!g_chain_fusion || !g_quad_enabled || nwin > 64 || n == 0 || steps < 2 || steps > 0x7fffffffull) return SP_OK;
  int log_q = 0;  // the size classes of enqueue_pedersen_impl
  if (n <= g_quad_max && nwin >= 16) log_q = 3;
  else if (n <= 2 * g_quad_max && nwin >= 8) log_q = 2;
  else if (n <= 4 * g_quad_max && nwin >= 4 && g_quad2_enabled) log_q = 1;
  if (log_q == 0) return SP_OK;
  int dup = 0;
  while ((4 << (log_q + dup)) < 64 && ((n * 4) << (log_q + dup + 1)) <= 65536) ++dup;  // up to one hash per wave
  if (g_quad_no_dup) dup = 0;
  const unsigned blocks = (unsigned)((((n * 4) << (log_q + dup)) + 255) / 256);
#define SP_LAUNCH_CHAIN(LOGQ)                                                                                      \
  hipLaunchKernelGGL((ped_chain_kernel<LOGQ>), dim3(blocks), dim3(256), 0, st, first, words, word_stride, n, (int)steps, \
                     h_right, c.ped, w0, log2e, nwin, flag, out, dup)
  if (log_q == 3) SP_LAUNCH_CHAIN(3);
  else if (log_q == 2) SP_LAUNCH_CHAIN(2);
  else SP_LAUNCH_CHAIN(1);
#undef SP_LAUNCH_CHAIN
  SP_HIP(hipGetLastError());
  *done = true;
  return SP_OK;
}
// Up to four consecutive small levels of a dense forest as one launch (ped_top_kernel): `cur` = the level-major
// buffer at a level of n_in nodes, `remaining` = levels left inside every tree.  Returns the number of levels
// enqueued (0: not this size class, the caller enqueues one level the usual way).
int enqueue_pedersen_top(uint64_t* cur, size_t n_in, unsigned remaining, unsigned* flag, hipStream_t st, int* levels_done) {
  *levels_done = 0;
  Context& c = ctx();
  const int nwin = c.plan.nwin;
  if (!g_top_fusion || !g_quad_enabled || nwin > 64 || nwin < 16 || remaining < 2 || n_in / 2 > g_quad_max || n_in < 4) return SP_OK;
  const int L = remaining < 4 ? (int)remaining : 4;
  const unsigned blocks = (unsigned)(n_in >> L);  // every tree holds a whole number of 2^L-node chunks at this level
  hipLaunchKernelGGL(ped_top_kernel, dim3(blocks), dim3(256), 0, st, cur, n_in, L, c.ped, (int)c.plan.bits[0], c.plan.log2e,
                     nwin, flag);
  SP_HIP(hipGetLastError());
  *levels_done = L;
  return SP_OK;
}
// pl.n_levels consecutive levels of a sparse multi-update with n nodes each and no merging paths, as one launch
// (ped_path_kernel).  *done = false when the levels are not of a size class the quad kernels serve (or the switch is
// off): the caller then enqueues them one by one.  cpts_tree: the tree's constant points ([2 l], [2 l + 1]) or null.
int enqueue_pedersen_path(uint64_t* felts, const uint64_t* emp, unsigned* flag, size_t n, hipStream_t st,
                          const int2* src_all, const PathLevels& pl, const aff_packed* cpts_tree, bool* done) {
  *done = false;
  Context& c = ctx();
  const int w0 = c.plan.bits[0], log2e = c.plan.log2e, nwin = c.plan.nwin;
  if (!g_path_fusion || !g_quad_enabled || nwin > 64 || n == 0 || pl.n_levels < 2) return SP_OK;
  int log_q = 0;  // the size classes of enqueue_pedersen_impl
  if (n <= g_quad_max && nwin >= 16) log_q = 3;
  else if (n <= 2 * g_quad_max && nwin >= 8) log_q = 2;
  else if (n <= 4 * g_quad_max && nwin >= 4 && g_quad2_enabled) log_q = 1;
  if (log_q == 0) return SP_OK;
  int cx = 0, cy = 0;
  bool sparse = false;
  if (g_sparse_enabled && cpts_tree != nullptr && (log_q == 1 || log_q == 2)) {
    constant_window_counts(c.plan, cx, cy);
    const int shortest = 1 + nwin - (cx > cy ? cx : cy);
    sparse = cx >= 1 && cy >= 1 && shortest >= (2 << log_q);
  }
  int dup = 0;
  while ((4 << (log_q + dup)) < 64 && ((n * 4) << (log_q + dup + 1)) <= 65536) ++dup;  // up to one hash per wave
  if (g_quad_no_dup) dup = 0;
  const unsigned blocks = (unsigned)((((n * 4) << (log_q + dup)) + 255) / 256);
#define SP_LAUNCH_PATH(LOGQ, SPARSEV)                                                                             \
  hipLaunchKernelGGL((ped_path_kernel<LOGQ, SPARSEV>), dim3(blocks), dim3(256), 0, st, felts, emp, n, c.ped, w0, log2e, \
                     nwin, flag, src_all, pl, dup, sparse ? cpts_tree : nullptr, cx, cy)
  if (sparse) {
    if (log_q == 2) SP_LAUNCH_PATH(2, true);
    else SP_LAUNCH_PATH(1, true);
  } else if (log_q == 3) SP_LAUNCH_PATH(3, false);
  else if (log_q == 2) SP_LAUNCH_PATH(2, false);
  else SP_LAUNCH_PATH(1, false);
#undef SP_LAUNCH_PATH
  SP_HIP(hipGetLastError());
  *done = true;
  return SP_OK;
}

static int enqueue_pedersen_impl(const uint64_t* x, size_t xs, const uint64_t* y, size_t ys, uint64_t* out,
                                 size_t os, uint8_t* status, unsigned* flag, size_t n, hipStream_t st,
                                 const Scratch& s, const int2* src, const aff_packed* cpts) {
  if (n == 0) return SP_OK;
  Context& c = ctx();
  // sp_profile_begin/_end time the DOMINANT kernel only (ped_accumulate_kernel: the pure one-lane-per-hash
  // launches, levels that are whole rounds of 65 536 hashes): an event pair around every small launch of a
  // tree costs ~5 us each, 7 % of a 20-tree forest
  bool prof = false;
  // lanes per hash: a SIMD issues about one VALU instruction per 5 cycles whether one wave or eight
  // live on it, so a launch takes ceil(waves / 1024) x (the dependent chain of one wave).  With
  // 1 < waves/SIMD < 2 some SIMDs get two waves and the launch takes twice the chain (measured: 81 920
  // lanes at L = 2 take 88 us, 40 960 hashes at L = 1 65 us), so the hashes are split over as many lanes
  // as still fit ONE wave per SIMD (65 536 lanes).  Every lane needs at least two windows.
  const int w0 = c.plan.bits[0], log2e = c.plan.log2e, nwin = c.plan.nwin;
  int log_l = 0;
  bool fused = false;
  if (g_split_enabled) {
    if (n * 8 <= g_split_lanes) log_l = 3;
    else if (n * 4 <= g_split_lanes) log_l = 2;
    else if (n * 2 <= g_split_lanes) log_l = 1;
    while (log_l > 0 && nwin < (2 << log_l)) --log_l;
  }
  // quad-parallel latency kernels while the level fits one wave per SIMD: 8 quads per hash up to 2048
  // hashes, 4 up to 4096 (same 18 rounds while nwin <= 20), 2 up to 8192
  int log_q = 0;
  if (g_quad_enabled && nwin <= 64) {
    if (n <= g_quad_max && nwin >= 16) log_q = 3;
    else if (n <= 2 * g_quad_max && nwin >= 8) log_q = 2;
    else if (n <= 4 * g_quad_max && nwin >= 4 && g_quad2_enabled) log_q = 1;
  }
  // sparse level: nodes with one constant operand sum 1 + nwin - cx (cy) points
  int cx = 0, cy = 0;
  bool sparse = false;
  // (levels the eight-quad kernel takes - up to 2048 nodes, the top of an update where most nodes have two touched
  // children - stay with it: 21.5 us against 22.5 for the four-quad sparse form, gpurun timeline of quick_tree_update)
  if (g_sparse_enabled && cpts != nullptr && src != nullptr && (log_q == 1 || log_q == 2)) {
    constant_window_counts(c.plan, cx, cy);
    const int shortest = 1 + nwin - (cx > cy ? cx : cy);
    sparse = cx >= 1 && cy >= 1 && shortest >= (2 << log_q);
  }
  if (log_q != 0) {
    int dup = 0;
    while ((4 << (log_q + dup)) < 64 && ((n * 4) << (log_q + dup + 1)) <= 65536) ++dup;  // up to one hash per wave
    if (g_quad_no_dup) dup = 0;  // A/B switch
    const unsigned blocks = (unsigned)((((n * 4) << (log_q + dup)) + 255) / 256);
#define SP_LAUNCH_QUAD(LOGQ, SPARSEV)                                                                           \
  hipLaunchKernelGGL((ped_quad_kernel<LOGQ, SPARSEV>), dim3(blocks), dim3(256), 0, st, x, y, xs, ys, n, c.ped, w0, \
                     log2e, nwin, status, flag, src, out, os, dup, cpts, cx, cy)
    if (sparse) {
      if (log_q == 2) SP_LAUNCH_QUAD(2, true);
      else SP_LAUNCH_QUAD(1, true);
    } else if (log_q == 3) SP_LAUNCH_QUAD(3, false);
    else if (log_q == 2) SP_LAUNCH_QUAD(2, false);
    else SP_LAUNCH_QUAD(1, false);
#undef SP_LAUNCH_QUAD
    fused = true;
  } else if (log_l == 0 && !(g_split_enabled && g_fuse_enabled && n <= 65536)) {
    // A launch takes ceil(waves / 1024) x the chain of one wave, so a level that is not a whole number of
    // waves per SIMD pays a full extra round for its last few hashes (81 920 hashes: two rounds of the
    // ~30 k-instruction one-lane-per-hash chain).  The level is cut instead into whole rounds of 65 536
    // hashes for the bulk kernel and a remainder of at most 32 768 hashes that goes to the lane-split kernel
    // (2, 4 or 8 lanes per hash, a chain of 18 k, 11 k or 9 k instructions) INSIDE THE SAME LAUNCH
    // (ped_accumulate_mixed_kernel); both write the same scratch planes and ONE finish launch serves the level.
    size_t n_bulk = n, rem = 0;
    int rem_l = 0;
    if (g_level_split && g_split_enabled && n > g_split_lanes && (n % g_split_lanes) != 0 &&
        (n % g_split_lanes) * 2 <= g_split_lanes) {
      rem = n % g_split_lanes;
      rem_l = rem * 8 <= g_split_lanes ? 3 : (rem * 4 <= g_split_lanes ? 2 : 1);
      while (rem_l > 0 && nwin < (2 << rem_l)) --rem_l;
      if (rem_l == 0) rem = 0;
      n_bulk = n - rem;
    }
    if (rem == 0) {
      prof = g_prof.enabled && g_prof.used + 2 <= g_prof.ev.size() && n > 65536;
      if (prof) (void)hipEventRecord(g_prof.ev[g_prof.used], st);
      hipLaunchKernelGGL(ped_accumulate_kernel, dim3((unsigned)((n_bulk + 255) / 256)), dim3(256), 0, st, x, y, xs, ys,
                         n_bulk, c.ped, w0, log2e, nwin, s.X, s.ZZ, status, flag, src, n);
    } else {
      const unsigned bulk_blocks = (unsigned)(n_bulk / 256);  // n_bulk is a multiple of 65 536
      const unsigned rblocks = (unsigned)(((rem << rem_l) + 255) / 256);
#define SP_LAUNCH_MIXED(LOGL)                                                                                     \
  hipLaunchKernelGGL((ped_accumulate_mixed_kernel<LOGL>), dim3(bulk_blocks + rblocks), dim3(256), 0, st, x, y, xs, \
                     ys, n_bulk, rem, bulk_blocks, c.ped, w0, log2e, nwin, s.X, s.ZZ, status, flag, src, n)
      if (rem_l == 3) SP_LAUNCH_MIXED(3);
      else if (rem_l == 2) SP_LAUNCH_MIXED(2);
      else SP_LAUNCH_MIXED(1);
#undef SP_LAUNCH_MIXED
    }
    if (prof) {
      (void)hipEventRecord(g_prof.ev[g_prof.used + 1], st);
      g_prof.units.push_back(n);
      g_prof.used += 2;
    }
  } else {
    const unsigned blocks = (unsigned)(((n << log_l) + 255) / 256);
    fused = g_fuse_enabled && (n << log_l) <= 65536;  // at most one wave per SIMD: the inversion costs latency only
#define SP_LAUNCH_SPLIT(LOGL, FUSEDV)                                                                      \
  hipLaunchKernelGGL((ped_accumulate_split_kernel<LOGL, FUSEDV>), dim3(blocks), dim3(256), 0, st, x, y, xs, \
                     ys, n, c.ped, w0, log2e, nwin, s.X, s.ZZ, status, flag, src, out, os, n)
    if (log_l == 0) SP_LAUNCH_SPLIT(0, true);  // one lane per hash, four hashes share one quad-split inversion
    else if (log_l == 3) { if (fused) SP_LAUNCH_SPLIT(3, true); else SP_LAUNCH_SPLIT(3, false); }
    else if (log_l == 2) { if (fused) SP_LAUNCH_SPLIT(2, true); else SP_LAUNCH_SPLIT(2, false); }
    else { if (fused) SP_LAUNCH_SPLIT(1, true); else SP_LAUNCH_SPLIT(1, false); }
#undef SP_LAUNCH_SPLIT
  }
  if (!fused) {
    const size_t T = finish_threads(n);
    const unsigned tpb = T >= 256 ? 256 : 64;
    const unsigned blocksB = (unsigned)((T + tpb - 1) / tpb);
    const size_t kmax = (n + T - 1) / T;  // elements per thread
    if (g_finish_lds && tpb == 256 && kmax >= 2 && kmax <= 16 && finish_lds_ready() == SP_OK) {
      const bool z_too = kmax <= 8;
      const size_t lds_bytes = kmax * NL * 256 * sizeof(int32_t) * (z_too ? 2 : 1);
      if (z_too) hipLaunchKernelGGL((ped_finish_lds_kernel<true>), dim3(blocksB), dim3(256), lds_bytes, st, s.X, s.ZZ, n, T,
                                    out, os, status, flag, (int)kmax);
      else hipLaunchKernelGGL((ped_finish_lds_kernel<false>), dim3(blocksB), dim3(256), lds_bytes, st, s.X, s.ZZ, n, T,
                              out, os, status, flag, (int)kmax);
    } else {
      hipLaunchKernelGGL(ped_finish_kernel, dim3(blocksB), dim3(tpb), 0, st, s.X, s.ZZ, s.Pre, n, T, out,
                         os, status, flag);
    }
  }
  SP_HIP(hipGetLastError());
  return SP_OK;
}

}  // namespace sp

using namespace sp;

extern "C" {

int sp_pedersen_batch_dev(const uint64_t* x, const uint64_t* y, uint64_t* out, uint8_t* status,
                          size_t n, void* stream) {
  CtxByPointer sp_ctx_sel__(x);  // the context of the device these pointers live on
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  Scratch s;
  int rc = get_scratch(n, s, (hipStream_t)stream);
  if (rc != SP_OK) return rc;
  return enqueue_pedersen(x, 1, y, 1, out, 1, status, nullptr, n, (hipStream_t)stream, s, nullptr);
}

int sp_pedersen_batch(const uint64_t* x, const uint64_t* y, uint64_t* out, uint8_t* status, size_t n) {
  if (ctx_count() > 1 && shard_context() < 0 && n >= SHARD_MIN_ITEMS) {  // one slice per device, side by side
    return shard_over_contexts(n, [&](size_t off, size_t cnt) {
      return sp_pedersen_batch(x + 4 * off, y + 4 * off, out + 4 * off, status ? status + off : nullptr, cnt);
    });
  }
  LaneScope ls;  // calls from different host threads overlap on the device(s) (context.hpp "Host lanes")
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  if (ls.open() != SP_OK) return SP_ERR_HIP;
  Context& c = ctx();
  HostLane& L = *ls.lane;
  const size_t fb = n * 32;
  SP_HIP(L.io.reserve(3 * fb + n + 64));
  uint64_t* dx = (uint64_t*)L.io.ptr;
  uint64_t* dy = (uint64_t*)((char*)L.io.ptr + fb);
  uint64_t* dout = (uint64_t*)((char*)L.io.ptr + 2 * fb);
  uint8_t* dst = (uint8_t*)L.io.ptr + 3 * fb;
  SP_HIP(hipMemcpyAsync(dx, x, fb, hipMemcpyHostToDevice, L.stream));
  SP_HIP(hipMemcpyAsync(dy, y, fb, hipMemcpyHostToDevice, L.stream));
  {
    ctx_lock lk(c.mu);  // the per-stream scratch map and the launch bookkeeping
    Scratch s;
    int rc = get_scratch(n, s, L.stream);
    if (rc != SP_OK) return rc;
    rc = enqueue_pedersen(dx, 1, dy, 1, dout, 1, dst, nullptr, n, L.stream, s, nullptr);
    if (rc != SP_OK) return rc;
  }
  SP_HIP(hipMemcpyAsync(out, dout, fb, hipMemcpyDeviceToHost, L.stream));
  if (status) SP_HIP(hipMemcpyAsync(status, dst, n, hipMemcpyDeviceToHost, L.stream));
  SP_HIP(hipStreamSynchronize(L.stream));
  return SP_OK;
}

int sp_profile_begin(size_t max_launches) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  while (g_prof.ev.size() < 2 * max_launches) {
    hipEvent_t e;
    SP_HIP(hipEventCreate(&e));
    g_prof.ev.push_back(e);
  }
  g_prof.used = 0;
  g_prof.units.clear();
  g_prof.enabled = true;
  return SP_OK;
}

int sp_profile_end(double* total_ms, uint64_t* launches, uint64_t* units) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  g_prof.enabled = false;
  double ms = 0;
  uint64_t u = 0;
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    SP_HIP(hipEventSynchronize(g_prof.ev[i + 1]));
    float f = 0;
    SP_HIP(hipEventElapsedTime(&f, g_prof.ev[i], g_prof.ev[i + 1]));
    ms += f;
    u += g_prof.units[i / 2];
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = g_prof.used / 2;
  if (units) *units = u;
  return SP_OK;
}

int sp_pedersen_point_batch(const uint64_t* x, const uint64_t* y, uint64_t* ox, uint64_t* oy,
                            uint8_t* status, size_t n) {
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  Context& c = ctx();
  ctx_lock lk(c.mu);
  const size_t fb = n * 32;
  SP_HIP(c.io.reserve(4 * fb + n + 64));
  char* b = (char*)c.io.ptr;
  uint64_t *dx = (uint64_t*)b, *dy = (uint64_t*)(b + fb), *dox = (uint64_t*)(b + 2 * fb),
           *doy = (uint64_t*)(b + 3 * fb);
  uint8_t* dst = (uint8_t*)(b + 4 * fb);
  SP_HIP(hipMemcpy(dx, x, fb, hipMemcpyHostToDevice));
  SP_HIP(hipMemcpy(dy, y, fb, hipMemcpyHostToDevice));
  SP_HIP(hipMemset(dox, 0, 2 * fb));
  hipLaunchKernelGGL(ped_point_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, 0, dx, dy, n,
                     c.ped, (int)c.plan.bits[0], c.plan.log2e, c.plan.nwin, dox, doy, dst);
  SP_HIP(hipGetLastError());
  SP_HIP(hipDeviceSynchronize());
  SP_HIP(hipMemcpy(ox, dox, fb, hipMemcpyDeviceToHost));
  SP_HIP(hipMemcpy(oy, doy, fb, hipMemcpyDeviceToHost));
  SP_HIP(hipMemcpy(status, dst, n, hipMemcpyDeviceToHost));
  return SP_OK;
}

int sp_pedersen_chains_dev(const uint64_t* elems, size_t width, size_t depth, uint64_t* out,
                           uint8_t* status, void* stream) {
  CtxByPointer sp_ctx_sel__(elems);  // the context of the device these pointers live on
  SP_REQUIRE_READY();
  if (depth < 1) { set_error("chain depth must be >= 1"); return SP_ERR_BAD_ARGUMENT; }
  if (width == 0) return SP_OK;
  Context& c = ctx();
  ctx_lock lk(c.mu);
  hipStream_t st = (hipStream_t)stream;
  Scratch s;
  int rc = get_scratch(width, s, st);
  if (rc != SP_OK) return rc;
  SP_HIP(hipMemsetAsync(s.flag, 0, sizeof(unsigned), st));
  if (depth == 1) {
    SP_HIP(hipMemcpyAsync(out, elems, width * 32, hipMemcpyDeviceToDevice, st));
  }
  // h lives in `out` and is updated in place: thread e reads x[e] and writes out[e] only after
  // kernel A of the same launch pair consumed it (A and B are separate kernels on one stream).
  bool fused = false;  // small batches of chains: every step inside one launch (ped_chain_kernel)
  if (depth >= 3) {
    rc = enqueue_pedersen_chain(elems, elems + 4 * width, (long long)width, width, depth - 1, false, out, s.flag, st, &fused);
    if (rc != SP_OK) return rc;
  }
  const uint64_t* h = elems;
  for (size_t j = 1; j < depth && !fused; ++j) {
    rc = enqueue_pedersen(h, 1, elems + 4 * j * width, 1, out, 1, nullptr, s.flag, width, st, s, nullptr);
    if (rc != SP_OK) return rc;
    h = out;
  }
  if (status) SP_HIP(hipMemcpyAsync(status, s.flag, 1, hipMemcpyDeviceToHost, st));
  return SP_OK;
}

// Host-pointer variant of sp_pedersen_chains_dev: one call for `width` chains of `depth` words.  Runs on a host
// lane (context.hpp): its own stream and staging buffer, the library lock only inside the _dev call that enqueues,
// no device-wide synchronisation - work in flight on other streams (a tree's insertion kernel, other threads'
// batches) is not waited for.
int sp_pedersen_chains(const uint64_t* elems, size_t width, size_t depth, uint64_t* out, uint8_t* status) {
  LaneScope ls(0);  // the primary context: where sp_order_batch keeps the key tables it verifies against next
  SP_REQUIRE_READY();
  if (depth < 1) { set_error("chain depth must be >= 1"); return SP_ERR_BAD_ARGUMENT; }
  if (width == 0) return SP_OK;
  if (ls.open() != SP_OK) return SP_ERR_HIP;
  HostLane& L = *ls.lane;
  SP_HIP(L.io.reserve((width * depth + width) * 32 + 64));
  uint64_t* d_el = (uint64_t*)L.io.ptr;
  uint64_t* d_out = d_el + 4 * width * depth;
  // small batches (sp_order_batch's message hashes) in and out through the lane's page-locked buffer: see PinnedBuffer
  const size_t in_bytes = width * depth * 32, out_bytes = width * 32;
  char* stage = nullptr;
  if (in_bytes + out_bytes <= PINNED_STAGE_MAX && L.hio.reserve(in_bytes + out_bytes) == hipSuccess) stage = (char*)L.hio.ptr;
  else (void)hipGetLastError();
  if (stage) std::memcpy(stage, elems, in_bytes);
  SP_HIP(hipMemcpyAsync(d_el, stage ? (const void*)stage : (const void*)elems, in_bytes, hipMemcpyHostToDevice, L.stream));
  uint8_t st8 = 0;  // written by the stream (the chains' status flag), read after the synchronisation below
  int rc = sp_pedersen_chains_dev(d_el, width, depth, d_out, &st8, L.stream);
  if (rc != SP_OK) return rc;
  SP_HIP(hipMemcpyAsync(stage ? (void*)(stage + in_bytes) : (void*)out, d_out, out_bytes, hipMemcpyDeviceToHost, L.stream));
  SP_HIP(hipStreamSynchronize(L.stream));
  if (stage) std::memcpy(out, stage + in_bytes, out_bytes);
  if (status) *status = st8;
  return SP_OK;
}

int sp_pedersen_chain(const uint64_t* elems, size_t n_elems, uint64_t* out, uint8_t* status) {
  SP_REQUIRE_READY();
  if (n_elems < 1) { set_error("chain needs at least one element"); return SP_ERR_BAD_ARGUMENT; }
  Context& c = ctx();
  ctx_lock lk(c.mu);
  SP_HIP(c.io.reserve(n_elems * 32 + 64));
  uint64_t* d_el = (uint64_t*)c.io.ptr;
  uint64_t* d_out = (uint64_t*)((char*)c.io.ptr + n_elems * 32);
  SP_HIP(hipMemcpy(d_el, elems, n_elems * 32, hipMemcpyHostToDevice));
  uint8_t st8 = 0;
  int rc = sp_pedersen_chains_dev(d_el, 1, n_elems, d_out, &st8, 0);
  if (rc != SP_OK) return rc;
  SP_HIP(hipDeviceSynchronize());
  SP_HIP(hipMemcpy(out, d_out, 32, hipMemcpyDeviceToHost));
  if (status) *status = st8;
  return SP_OK;
}

// Right fold h = H(e_i, h) from the last element down: the shape of cairo-lang's
// compute_hash_chain, which the program-hash harness calls
// (starkware/cairo/bootloaders/program_hash_test_utils.py:9).  Inherently serial: n - 1 launches.
int sp_pedersen_chain_right(const uint64_t* elems, size_t n_elems, uint64_t* out, uint8_t* status) {
  SP_REQUIRE_READY();
  if (n_elems < 1) { set_error("chain needs at least one element"); return SP_ERR_BAD_ARGUMENT; }
  Context& c = ctx();
  ctx_lock lk(c.mu);
  SP_HIP(c.io.reserve((n_elems + 2) * 32 + 64));
  uint64_t* d_el = (uint64_t*)c.io.ptr;
  uint64_t* d_a = d_el + 4 * n_elems;
  uint64_t* d_b = d_a + 4;
  SP_HIP(hipMemcpy(d_el, elems, n_elems * 32, hipMemcpyHostToDevice));
  Scratch s;
  int rc = get_scratch(1, s, 0);
  if (rc != SP_OK) return rc;
  SP_HIP(hipMemsetAsync(s.flag, 0, sizeof(unsigned), 0));
  const uint64_t* h = d_el + 4 * (n_elems - 1);
  uint64_t* nxt = d_a;
  bool fused = false;  // the whole fold inside one launch: h = H(e_i, h) for i = n - 2 .. 0 (ped_chain_kernel)
  if (n_elems >= 3) {
    rc = enqueue_pedersen_chain(h, d_el + 4 * (n_elems - 2), -1, 1, n_elems - 1, true, d_a, s.flag, 0, &fused);
    if (rc != SP_OK) return rc;
    if (fused) h = d_a;
  }
  for (size_t i = n_elems - 1; !fused && i-- > 0;) {
    rc = enqueue_pedersen(d_el + 4 * i, 1, h, 1, nxt, 1, nullptr, s.flag, 1, 0, s, nullptr);
    if (rc != SP_OK) return rc;
    h = nxt;
    nxt = (nxt == d_a) ? d_b : d_a;
  }
  SP_HIP(hipDeviceSynchronize());
  SP_HIP(hipMemcpy(out, h, 32, hipMemcpyDeviceToHost));
  if (status) {
    unsigned f = 0;
    SP_HIP(hipMemcpy(&f, s.flag, sizeof(unsigned), hipMemcpyDeviceToHost));
    *status = (uint8_t)f;
  }
  return SP_OK;
}

// Forest of n_trees independent trees (any count) of the given height, built in lockstep: the
// leaves of all trees are concatenated (tree t owns leaves [t 2^height, (t+1) 2^height)), so level k
// of the forest is one contiguous array of n_trees 2^(height - k) nodes and ONE launch pair per
// level serves every tree.  Stops after `height` levels: the last level holds the n_trees roots.
// Buffer: n_trees (2^(height+1) - 1) felts, level-major, leaves first.
int sp_merkle_forest_dev(uint64_t* levels, size_t n_trees, unsigned height, uint8_t* status,
                         void* stream) {
  CtxByPointer sp_ctx_sel__(levels);  // the context of the device these pointers live on
  SP_REQUIRE_READY();
  if (n_trees == 0) return SP_OK;
  if (height > 40 || (n_trees >> (40 - height)) != 0) { set_error("forest too large"); return SP_ERR_BAD_ARGUMENT; }
  Context& c = ctx();
  ctx_lock lk(c.mu);
  hipStream_t st = (hipStream_t)stream;
  const size_t n0 = n_trees << height;
  Scratch s;
  int rc = get_scratch(n0 / 2 + 1, s, st);
  if (rc != SP_OK) return rc;
  SP_HIP(hipMemsetAsync(s.flag, 0, sizeof(unsigned), st));
  uint64_t* cur = levels;
  size_t n = n0;
  for (unsigned k = 0; k < height;) {
    int fused = 0;  // the small levels go out four at a time (ped_top_kernel)
    rc = enqueue_pedersen_top(cur, n, height - k, s.flag, st, &fused);
    if (rc != SP_OK) return rc;
    if (fused > 0) {
      for (int j = 0; j < fused; ++j, n >>= 1) cur += 4 * n;
      k += (unsigned)fused;
      continue;
    }
    uint64_t* nxt = cur + 4 * n;
    rc = enqueue_pedersen(cur, 2, cur + 4, 2, nxt, 1, nullptr, s.flag, n / 2, st, s, nullptr);
    if (rc != SP_OK) return rc;
    cur = nxt;
    ++k;
    n >>= 1;
  }
  if (status) SP_HIP(hipMemcpyAsync(status, s.flag, 1, hipMemcpyDeviceToHost, st));
  return SP_OK;
}

int sp_merkle_build_dev(uint64_t* levels, unsigned height, uint8_t* status, void* stream) {
  return sp_merkle_forest_dev(levels, 1, height, status, stream);
}

// Commitment to a table of n_rows x n_cols felts stored column-major (column c at cols + 4 c n_rows):
// leaf r = left-fold Pedersen chain of row r (a single column commits the felts themselves), then
// the Merkle tree over the leaves.  levels: 2 n_rows - 1 felts, leaves first, root last.
int sp_commit_rows_dev(const uint64_t* cols, size_t n_rows, size_t n_cols, uint64_t* levels, uint8_t* status,
                       void* stream) {
  CtxByPointer sp_ctx_sel__(cols);  // the context of the device these pointers live on
  SP_REQUIRE_READY();
  if (n_rows == 0 || (n_rows & (n_rows - 1)) != 0 || n_cols == 0) {
    set_error("rows must be a power of two, columns at least one");
    return SP_ERR_BAD_ARGUMENT;
  }
  ctx_lock lk(ctx().mu);
  unsigned height = 0;
  while (((size_t)1 << height) < n_rows) ++height;
  uint8_t st_chain = 0, st_tree = 0;
  int rc = sp_pedersen_chains_dev(cols, n_rows, n_cols, levels, status ? &st_chain : nullptr, stream);
  if (rc != SP_OK) return rc;
  rc = sp_merkle_build_dev(levels, height, status ? &st_tree : nullptr, stream);
  if (rc != SP_OK) return rc;
  if (status) {  // the two flags arrive with the stream; fold them once it has drained
    SP_HIP(hipStreamSynchronize((hipStream_t)stream));
    *status = st_chain | st_tree;
  }
  return SP_OK;
}

int sp_merkle_root(const uint64_t* leaves, unsigned height, uint64_t* root, uint64_t* levels_out,
                   uint8_t* status) {
  SP_REQUIRE_READY();
  if (height > 30) { set_error("height too large"); return SP_ERR_BAD_ARGUMENT; }
  Context& c = ctx();
  const size_t n0 = (size_t)1 << height;
  const size_t total = 2 * n0 - 1;
  ctx_lock lk(c.mu);
  SP_HIP(c.io.reserve(total * 32));
  uint64_t* d = (uint64_t*)c.io.ptr;
  SP_HIP(hipMemcpy(d, leaves, n0 * 32, hipMemcpyHostToDevice));
  uint8_t st8 = 0;
  int rc = sp_merkle_build_dev(d, height, &st8, 0);
  if (rc != SP_OK) return rc;
  SP_HIP(hipDeviceSynchronize());
  SP_HIP(hipMemcpy(root, d + 4 * (total - 1), 32, hipMemcpyDeviceToHost));
  if (levels_out) SP_HIP(hipMemcpy(levels_out, d, total * 32, hipMemcpyDeviceToHost));
  if (status) *status = st8;
  return SP_OK;
}

}  // extern "C"

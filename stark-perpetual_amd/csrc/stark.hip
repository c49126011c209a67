// Prover-side kernels over GF(p), p = 2^251 + 17*2^192 + 1: radix-2 NTT / coset LDE staged
// through LDS, the Pedersen-step AIR (trace generation + constraint evaluation), FRI folding.
// The reference tree has no prover, so everything here is build-defined and checked against
// oracle/stark_ref.py ("parity unpinned"); the field, its generator
// (pedersen_params.json:20-21) and the AIR's step relation (signature.py:305-317,
// math_utils.py:64-67) are the reference's.
//
// Data layout in HBM: a column is a contiguous array of 32-byte felts, plain integers everywhere
// (C ABI and between passes).  No kernel converts its operands to Montgomery form: every
// multiplication in a transform, a fold or the composition has at least one CONSTANT factor (twiddle,
// coset power, 1/n, alpha, 1/Z) and the constants are kept in Montgomery form, so
// fe_mul(plain, constant R) = plain * constant - the R of the reduction cancels against the R of the
// constant (round 1 paid 13 of ~45 multiplications per composition point, 3 of 6 per fold and 2 per
// transform element for conversions).  Algorithmic bytes: an NTT pass reads and writes each felt once
// (64 B per element per pass); 2^22 points take 3 passes (two strided ones of 6 stages each and one of 10 stages on
// a contiguous 1024-felt tile in LDS; 2^20 points 2 passes with the 2048-felt tile: pick_tile_log below).
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "context.hpp"
#include "curve_consts.hpp"

namespace sp {

// Tile = 2048 felts = 72 KiB of LDS as 9 x int32 planes, two blocks of 256 threads per CU.  Round 5 measured the
// two-pass plans the round-4 verdict asked for against this one (profiles/r05_ntt_plans.txt, one box): tiles of 4096
// felts (144 KiB, ONE block of 512 threads per CU; 2^22 points = 12 local + 10 strided stages with 128-byte segments,
// -DSP_NTT_TILE_LOG=12) and 2048-felt tiles with an 11-stage strided pass of 32-byte segments
// (-DSP_NTT_STRIDED_MAX=11).  Both remove one pass over HBM and neither is faster: the single block per CU puts all
// eight waves of a CU into the same load / compute / store phase (862 + 857 us for the two big passes against
// 650 + 513 + 522), the 32-byte segments make the strided pass HBM-bound (1012 us against 513 + 522).  The
// three-pass plan stays; the macros keep the variants buildable.
#ifndef SP_NTT_TILE_LOG
#define SP_NTT_TILE_LOG 11
#endif
constexpr int TILE_LOG = SP_NTT_TILE_LOG;
constexpr int TILE = 1 << TILE_LOG;
// threads of a block = tile / 8: threads x 8 felts in registers = one radix-8 step of a tile (ntt_threads_of below)
#ifndef SP_NTT_STRIDED_MAX
#define SP_NTT_STRIDED_MAX (SP_NTT_TILE_LOG - 2)  // at least 4 adjacent columns = 128-byte segments
#endif
constexpr int NTT_STRIDED_MAX = SP_NTT_STRIDED_MAX;  // stages of a strided pass
// Round 6: a SECOND tile size for the transforms that take the same number of passes with it.  Tiles of 1024 felts
// (36 KiB of LDS, 128 threads) put FOUR independent blocks on a CU instead of two: the load / compute / store phases of
// a tile do not overlap inside a block (profiles/r06_ntt_pass_ceiling.txt: the data path alone is 0.85 of the 2.17 ms
// of a 4-column LDE), they overlap across blocks, and four blocks overlap better than two - the three forward passes
// over 2^22 points take 1.62 ms instead of 1.75.  A 2^20-point transform needs 11 + 9 stages = two passes with the big
// tile and three with the small one, so it keeps the big one: pick_tile_log() chooses per transform.
// -DSP_NTT_SMALL_TILE_LOG=0 builds the single-tile plan of rounds 3 - 5.
#ifndef SP_NTT_SMALL_TILE_LOG
#define SP_NTT_SMALL_TILE_LOG 10
#endif
constexpr int SMALL_TILE_LOG = SP_NTT_SMALL_TILE_LOG;
constexpr int ntt_threads_of(int tile_log) { return 1 << (tile_log - 3); }
constexpr size_t ntt_lds_bytes_of(int tile_log) { return (size_t)NL * ((size_t)1 << tile_log) * sizeof(int32_t); }
constexpr int ntt_strided_max_of(int tile_log) { return tile_log == TILE_LOG ? NTT_STRIDED_MAX : tile_log - 2; }

// Slot of tile element e inside a limb plane: bank bit i = e_i ^ e_(i+3).  The lanes of a wave walk the tile
// with their six index bits at e3..e8 (radix-8 group of stages 0-2), at e0..e2 + e6..e8 (stages 3-5), at e0..e5
// (stages 6+, loads, stores) or at e2..e7 (zero padding): each of these maps onto the five bank bits with rank 5,
// i.e. two lanes per bank - what a wave64 access costs anyway.  The plain layout put the first two patterns on
// 8 banks (68 % of the LDS cycles of the contiguous passes were bank conflicts, SQ_LDS_BANK_CONFLICT).
__device__ __forceinline__ int lds_slot(int e) { return e ^ ((e >> 3) & 31); }
template <int TILE_>
__device__ __forceinline__ fe lds_get(const int32_t* lds, int e) {
  const int s = lds_slot(e);
  fe v;
#pragma unroll
  for (int l = 0; l < NL; ++l) v.l[l] = lds[l * TILE_ + s];
  return v;
}
template <int TILE_>
__device__ __forceinline__ void lds_put(int32_t* lds, int e, const fe& v) {
  const int s = lds_slot(e);
#pragma unroll
  for (int l = 0; l < NL; ++l) lds[l * TILE_ + s] = v.l[l];
}
__device__ __forceinline__ fe ld_fe_packed(const uint64_t* p) { return fe_unpack(ld_u256(p)); }
// The tile kernel streams: every felt of a pass is read once and written once, 1 GB per pass of a 4-column 2^22-point
// transform against 4 MB of L2 per XCD.  SP_NTT_NT (build switch, A/B in profiles/r06_ntt_pass_ceiling.txt): bit 0 -
// non-temporal stores of the tile, bit 1 - non-temporal loads.
#ifndef SP_NTT_NT
#define SP_NTT_NT 0
#endif
typedef uint32_t ntt_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u256 ntt_ld(const uint64_t* p) {
#if SP_NTT_NT & 2
  const ntt_u32x4* q = reinterpret_cast<const ntt_u32x4*>(p);
  const ntt_u32x4 a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1);
  u256 r;
  r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w;
  r.w[4] = b.x; r.w[5] = b.y; r.w[6] = b.z; r.w[7] = b.w;
  return r;
#else
  return ld_u256(p);
#endif
}
__device__ __forceinline__ void ntt_st(uint64_t* p, const u256& v) {
#if SP_NTT_NT & 1
  ntt_u32x4* q = reinterpret_cast<ntt_u32x4*>(p);
  ntt_u32x4 a, b;
  a.x = v.w[0]; a.y = v.w[1]; a.z = v.w[2]; a.w = v.w[3];
  b.x = v.w[4]; b.y = v.w[5]; b.z = v.w[6]; b.w = v.w[7];
  __builtin_nontemporal_store(a, q);
  __builtin_nontemporal_store(b, q + 1);
#else
  st_u256(p, v);
#endif
}

// Where a tile sits in the transform and where its twiddles come from.
struct ntt_geom {
  int log_c, log_lo, log_tw;
  size_t lowb;                  // the tile's block of adjacent columns
  const uint64_t* tw;           // omega^k, k < 2^(log_tw - 1), Montgomery, packed
};
// Twiddle of the butterfly whose lower element has coupled index k, column c, in stage t of the pass
// (global span h = 2^(t + log_lo)): omega_{2h}^(index mod h).
__device__ __forceinline__ u256 ntt_twiddle(const ntt_geom& g, int t, int k, int c) {
  const size_t j = ((size_t)(k & ((1 << t) - 1)) << g.log_lo) | (g.lowb << g.log_c) | (size_t)c;
  return ld_u256(g.tw + 4 * (j << (g.log_tw - 1 - t - g.log_lo)));
}

// LOGR consecutive radix-2 stages (t_lo .. t_lo + LOGR - 1 of the pass) on 2^LOGR felts held in REGISTERS:
// one LDS round trip and one barrier per LOGR stages instead of per stage (round 2: eleven of each per tile,
// every operand through LDS as nine int32 planes), and carries only where the limb bounds ask for one.
// Bounds, in units of 2^29 per limb ("B"): tile values enter with limbs 0..7 in [0, 2^29) (B = 1); a product
// is B = 1; a sum adds the bounds of its operands, a difference of two non-negative values keeps the larger
// one; a multiplicand may be B <= 3 (27 * 2^58 < 2^63), an int32 limb holds B <= 4 strictly (4 (2^29 - 1) + a
// carry of 3 is 2^31 - 1), and fe_carry is safe up to there.  The VALUE is a separate matter:
//   DIT: a' = a + w b, b' = a - w b grow by one B and at most p per stage: three stages end at B = 4, every
//     output gets ONE fe_carry and no value reduction (11 stages of a pass add < 12 p; fe_canon at the store).
//   DIF: a' = a + b doubles, b' = (a - b) w is fresh.  Of the eight felts only x0, x1 need a carry before the
//     third stage; at the end the sums are carried and only x0 - the one felt that is a pure sum of all
//     inputs, whose value doubles per stage - gets its multiple of p taken off (fe_weak_reduce at B <= 2: its
//     limb-6 adjustment of up to 2^27 must not meet a limb near 2^31; at B = 4 it would, and four limbs of
//     2^29 - 1 each are exactly what "-small" values of a sparse polynomial look like).
// `unit_low`: the lowest stage is stage 0 of the whole transform, whose twiddle is 1 - no multiplication.
template <int LOGR, bool DIT, int U>
__device__ __forceinline__ void ntt_stage(fe (&x)[1 << LOGR], int (&B)[1 << LOGR], const u256 (&tw)[1 << LOGR],
                                          bool unit_low) {
  constexpr int R = 1 << LOGR;
  const bool unit = unit_low && U == 0;  // stage t_lo + U pairs x[i] with x[i + 2^U]
#pragma unroll
  for (int mm = 0; mm < (1 << U); ++mm) {
    fe w = FE_ZERO;
    if (!unit) w = fe_unpack(tw[(1 << U) + mm]);
#pragma unroll
    for (int hi = 0; hi < (R >> (U + 1)); ++hi) {
      const int i0 = (hi << (U + 1)) | mm, i1 = i0 + (1 << U);
      if (DIT) {
        const fe wb = unit ? x[i1] : fe_mul(x[i1], w);
        const int bw = unit ? B[i1] : 1;
        x[i1] = fe_sub(x[i0], wb);
        x[i0] = fe_add(x[i0], wb);
        B[i0] = B[i1] = B[i0] + bw;
      } else {
        if (B[i0] + B[i1] > 4) {  // resolved at compile time: only x0, x1 before the last stage of a radix-8 group
          x[i0] = fe_carry(x[i0]);
          x[i1] = fe_carry(x[i1]);
          B[i0] = B[i1] = 1;
        }
        const fe d = fe_sub(x[i0], x[i1]);  // |d| <= max(B): both operands have non-negative limbs
        x[i0] = fe_add(x[i0], x[i1]);
        x[i1] = unit ? d : fe_mul(d, w);
        const int bs = B[i0] + B[i1];
        B[i1] = unit ? (B[i0] > B[i1] ? B[i0] : B[i1]) + 1 : 1;  // a stored difference has signed limbs: carry it
        B[i0] = bs;
      }
    }
  }
}
template <int LOGR, bool DIT, int TILE_>
__device__ __forceinline__ void ntt_group(int32_t* lds, const ntt_geom& g, int grp, int t_lo, bool unit_low) {
  constexpr int R = 1 << LOGR;
  const int c = grp & ((1 << g.log_c) - 1), kk = grp >> g.log_c;
  const int k0 = ((kk >> t_lo) << (t_lo + LOGR)) | (kk & ((1 << t_lo) - 1));
  const int e0 = (k0 << g.log_c) | c, estep = 1 << (t_lo + g.log_c);
  // All 2^LOGR - 1 twiddles of the group are requested BEFORE the tile values are read from LDS: with two
  // waves per SIMD nothing else hides a table fetch, and hipcc otherwise places every load right in front of its
  // first use (seven exposed round trips per radix-8 group).  Slot 2^U + mm: stage U, pair class mm.
  u256 tw[R];
#pragma unroll
  for (int U = 0; U < LOGR; ++U) {
#pragma unroll
    for (int mm = 0; mm < (1 << U); ++mm) {
      if (!(unit_low && U == 0)) tw[(1 << U) + mm] = ntt_twiddle(g, t_lo + U, k0 + (mm << t_lo), c);
    }
  }
  fe x[R];
  int B[R];
#pragma unroll
  for (int m = 0; m < R; ++m) {
    x[m] = lds_get<TILE_>(lds, e0 + m * estep);
    B[m] = 1;
  }
  if constexpr (DIT) {
    ntt_stage<LOGR, DIT, 0>(x, B, tw, unit_low);
    if constexpr (LOGR > 1) ntt_stage<LOGR, DIT, 1>(x, B, tw, unit_low);
    if constexpr (LOGR > 2) ntt_stage<LOGR, DIT, 2>(x, B, tw, unit_low);
  } else {
    if constexpr (LOGR > 2) ntt_stage<LOGR, DIT, 2>(x, B, tw, unit_low);
    if constexpr (LOGR > 1) ntt_stage<LOGR, DIT, 1>(x, B, tw, unit_low);
    ntt_stage<LOGR, DIT, 0>(x, B, tw, unit_low);
    // x0 = the sum of all inputs: the only value that grows geometrically.  Adjust at B <= 2 only.
    if (B[0] > 2) x[0] = fe_carry(x[0]);
    x[0] = fe_weak_reduce(x[0]);
    B[0] = 1;
  }
#pragma unroll
  for (int m = 0; m < R; ++m) lds_put<TILE_>(lds, e0 + m * estep, B[m] > 1 ? fe_carry(x[m]) : x[m]);
}

// One pass = `nst` consecutive radix-2 stages on tiles of 2^log_e felts (2^log_t coupled points x
// 2^(log_e-log_t) adjacent columns), executed as radix-8 / radix-4 / radix-2 groups (ntt_group).  Stage with
// global span h = 2^(t + log_lo):
//   DIF (forward half of a natural->bit-reversed transform): a' = a + b, b' = (a - b) w
//   DIT (bit-reversed->natural):                             a' = a + w b, b' = a - w b
// with w = omega_{2h}^(index mod h) = tw[(index mod h) << (log_tw - 1 - t - log_lo)].
// use_scale: bit 0 - multiply the outputs by `scale`; bit 1 - not the last pass: store a 256-bit representative
// in (0, 3p) instead of the canonical one.
// pad_log_b > 0 (first pass of the big transform of an LDE): `in` is the coefficient array of the small
// transform (2^pad_log_n felts per column, bit-reversed order); slot e of the tile stands for
// coef[idx >> pad_log_b] * G[bitrev(idx >> pad_log_b)] when the low pad_log_b bits of idx are zero and
// for 0 otherwise (zero padding in bit-reversed positions), and the first pad_log_b DIT stages - which
// only copy that value over its 2^pad_log_b slots (a + w * 0) - are skipped: the padded column is
// never written to HBM.
template <int TL>
__global__ void __launch_bounds__(1 << (TL - 3))
ntt_tile_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int log_e, int log_t,
                int log_lo, int nst, int t_first, int dit, const uint64_t* __restrict__ tw, int log_tw,
                int use_scale, fe scale, size_t in_col_stride, size_t out_col_stride,
                const uint64_t* __restrict__ pad_G, int pad_log_b, int pad_log_n) {
  constexpr int TILE = 1 << TL, NTT_THREADS = TILE / 8;  // (shadow the file-level constants: this instantiation's tile)
  extern __shared__ int32_t lds[];  // NL planes of TILE int32 (72 KiB for the big tile: above the 64 KiB a static array may have)
  in += 4 * in_col_stride * blockIdx.y;    // grid.y = column: independent columns share one launch
  out += 4 * out_col_stride * blockIdx.y;
  const int E = 1 << log_e;
  const int log_c = log_e - log_t;
  const int C = 1 << log_c;
  const size_t low_blocks = ((size_t)1 << log_lo) >> log_c;
  const size_t high = blockIdx.x / low_blocks;
  const size_t lowb = blockIdx.x % low_blocks;
  const size_t base = (high << (log_t + log_lo)) | (lowb << log_c);
  int s_begin = 0;
  if (pad_log_b > 0) {  // contiguous tile (log_lo == 0, log_c == 0): one coefficient fills 2^pad_log_b slots
    const int B = 1 << pad_log_b;
    for (int g = threadIdx.x; g < (E >> pad_log_b); g += NTT_THREADS) {
      const size_t j = (base >> pad_log_b) + (size_t)g;
      const size_t c = pad_log_n ? (__brevll((unsigned long long)j) >> (64 - pad_log_n)) : 0;
      const fe v = fe_mul(ld_fe_packed(in + 4 * j), ld_fe_packed(pad_G + 4 * c));
      for (int q = 0; q < B; ++q) lds_put<TILE>(lds, (g << pad_log_b) + q, v);
    }
    s_begin = pad_log_b < nst ? pad_log_b : nst;
  } else {
    if (E == TILE) {  // the usual tile: all eight loads of a thread in flight before the first unpack
      constexpr int PER = TILE / NTT_THREADS;
      u256 v[PER];
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int e = threadIdx.x + q * NTT_THREADS;
        const int k = e >> log_c, c = e & (C - 1);
        v[q] = ntt_ld(in + 4 * (base | ((size_t)k << log_lo) | (size_t)c));
      }
#pragma unroll
      for (int q = 0; q < PER; ++q) lds_put<TILE>(lds, threadIdx.x + q * NTT_THREADS, fe_unpack(v[q]));
    } else {
      for (int e = threadIdx.x; e < E; e += NTT_THREADS) {
        const int k = e >> log_c, c = e & (C - 1);
        const size_t idx = base | ((size_t)k << log_lo) | (size_t)c;
        lds_put<TILE>(lds, e, fe_unpack(ntt_ld(in + 4 * idx)));  // 256-bit input: limbs 0..7 normal, top limb < 2^24
      }
    }
  }
  __syncthreads();
  const ntt_geom g = {log_c, log_lo, log_tw, lowb, tw};
  // the stages of the pass, lowest first: DIT walks them upwards, DIF downwards
  int left = nst - s_begin;
  int t_next = dit ? t_first + s_begin : t_first;  // next stage to do
  while (left > 0) {
    const int r = left == 4 ? 2 : (left >= 3 ? 3 : left);  // stages in this group: 10 = 3 + 3 + 2 + 2, never 3 + 3 + 3 + 1
    const int t_lo = dit ? t_next : t_next - r + 1;
    const bool unit_low = (t_lo + log_lo) == 0;
    const int groups = E >> r;
    for (int grp = threadIdx.x; grp < groups; grp += NTT_THREADS) {
      if (dit) {
        if (r == 3) ntt_group<3, true, TILE>(lds, g, grp, t_lo, unit_low);
        else if (r == 2) ntt_group<2, true, TILE>(lds, g, grp, t_lo, unit_low);
        else ntt_group<1, true, TILE>(lds, g, grp, t_lo, unit_low);
      } else {
        if (r == 3) ntt_group<3, false, TILE>(lds, g, grp, t_lo, unit_low);
        else if (r == 2) ntt_group<2, false, TILE>(lds, g, grp, t_lo, unit_low);
        else ntt_group<1, false, TILE>(lds, g, grp, t_lo, unit_low);
      }
    }
    __syncthreads();
    left -= r;
    t_next = dit ? t_next + r : t_next - r;
  }
  for (int e = threadIdx.x; e < E; e += NTT_THREADS) {
    const int k = e >> log_c, c = e & (C - 1);
    const size_t idx = base | ((size_t)k << log_lo) | (size_t)c;
    fe v = lds_get<TILE>(lds, e);
    if (use_scale & 1) v = fe_mul(v, scale);
    if (use_scale & 2) {
      // Between the passes of one transform the felt only has to FIT 256 bits: take floor(v / 2^251) - 1 multiples
      // of p off (value in (0, 3p), one carry chain) instead of the three chains of the canonical form.  The next
      // pass unpacks it to normal limbs and a top limb below 2^21 - all its bounds ask for.
      const int32_t q = (v.l[8] >> 19) - 1;
      v.l[0] -= q;
      v.l[6] -= q * P6;
      v.l[8] -= q * P8;
      ntt_st(out + 4 * idx, fe_pack(fe_carry(v)));
    } else {
      ntt_st(out + 4 * idx, fe_pack(fe_canon(v)));
    }
  }
}

// W[k] = base^k (Montgomery, packed) from the squarings pw[b] = base^(2^b); optional factor.
__global__ void __launch_bounds__(256)
powers_kernel(uint64_t* __restrict__ W, size_t count, const fe* __restrict__ pw, int nbits, fe factor) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  fe acc = factor;
  for (int b = 0; b < nbits; ++b) {
    if ((k >> b) & 1) acc = fe_mul(acc, pw[b]);
  }
  st_u256(W + 4 * k, fe_pack(fe_canon(fe_mul(acc, FE_ONE_M))));
}

// out[i] = in[bitrev(i)]
__global__ void __launch_bounds__(256)
bitrev_copy_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int log_n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> log_n) return;
  const size_t r = log_n ? (__brevll((unsigned long long)i) >> (64 - log_n)) : 0;
  st_u256(out + 4 * i, ld_u256(in + 4 * r));
}

// out[j] = coef[j] * G[bitrev(j)] (an LDE without blowup: only the coset shift)
__global__ void __launch_bounds__(256)
scale_copy_kernel(const uint64_t* __restrict__ coef, uint64_t* __restrict__ out, int log_n,
                  const uint64_t* __restrict__ G) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >> log_n) return;
  coef += ((size_t)4 << log_n) * blockIdx.y;
  out += ((size_t)4 << log_n) * blockIdx.y;
  const size_t c = log_n ? (__brevll((unsigned long long)j) >> (64 - log_n)) : 0;
  st_u256(out + 4 * j, fe_pack(fe_canon(fe_mul(ld_fe_packed(coef + 4 * j), ld_fe_packed(G + 4 * c)))));
}

// ---- Pedersen-step AIR --------------------------------------------------------------------------
// Trace generation (witness): one thread per hash writes 512 rows of (s, px, py, lambda).  The
// reference's affine chord rule (math_utils.py:59-68) costs one inversion per set bit; here the
// 504 partial sums are accumulated projectively (XYZZ), and the two families of inversions the
// affine rows need - 1/ZZZ of every partial sum, then 1/(px - cx) of every addition row - are each
// done with ONE divsteps inversion per hash (Montgomery's trick over the rows of that hash).
// Scratch planes are [row][limb][hash] so that the 64 hashes of a wave touch consecutive dwords.
__device__ __forceinline__ void tr_store(int32_t* plane, size_t m, int r, size_t h, const fe& v) {
#pragma unroll
  for (int l = 0; l < NL; ++l) plane[((size_t)(r * NL + l)) * m + h] = v.l[l];
}
__device__ __forceinline__ fe tr_load(const int32_t* plane, size_t m, int r, size_t h) {
  fe v;
#pragma unroll
  for (int l = 0; l < NL; ++l) v.l[l] = plane[((size_t)(r * NL + l)) * m + h];
  return v;
}

__global__ void __launch_bounds__(64)
pedersen_trace_kernel(const uint64_t* __restrict__ x, const uint64_t* __restrict__ y, size_t m,
                      const aff_packed* __restrict__ bits /* 504 per-bit points */, aff_packed shift,
                      uint64_t* __restrict__ cols /* [4][512 m] plain felts */,
                      int32_t* __restrict__ sc /* 5 planes of 512 * 9 * m int32 */) {
  const size_t hsh = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (hsh >= m) return;
  const size_t n = 512 * m;
  const size_t plane = (size_t)512 * NL * m;
  int32_t *sX = sc, *sY = sc + plane, *sZZ = sc + 2 * plane, *sZZZ = sc + 3 * plane, *sPre = sc + 4 * plane;
  uint64_t* cs = cols;
  uint64_t* cpx = cols + 4 * n;
  uint64_t* cpy = cols + 8 * n;
  uint64_t* cl = cols + 12 * n;
  // pass A: projective partial sums before every row, s column
  xyzz acc = xyzz_from_aff(ld_aff(&shift));
  for (int block = 0; block < 2; ++block) {
    u256 s = ld_u256((block ? y : x) + 4 * hsh);
    for (int j = 0; j < 256; ++j) {
      const int r = 256 * block + j;
      st_u256(cs + 4 * (512 * hsh + r), s);
      tr_store(sX, m, r, hsh, acc.X);
      tr_store(sY, m, r, hsh, acc.Y);
      tr_store(sZZ, m, r, hsh, acc.ZZ);
      tr_store(sZZZ, m, r, hsh, acc.ZZZ);
      if (j < 252) {
        if (s.w[0] & 1u) acc = xyzz_madd(acc, ld_aff(bits + 252 * block + j));
#pragma unroll
        for (int q = 0; q < 7; ++q) s.w[q] = (s.w[q] >> 1) | (s.w[q + 1] << 31);
        s.w[7] >>= 1;
      }
    }
  }
  // pass B: one inversion for all 512 ZZZ; affine px, py (plain columns; Montgomery copies kept)
  fe run = FE_ONE_M;
  for (int r = 0; r < 512; ++r) {
    tr_store(sPre, m, r, hsh, run);
    run = fe_mul(run, tr_load(sZZZ, m, r, hsh));
  }
  fe inv = fe_inv(run);
  for (int r = 511; r >= 0; --r) {
    const fe zzz = tr_load(sZZZ, m, r, hsh);
    const fe izzz = fe_mul(inv, tr_load(sPre, m, r, hsh));
    inv = fe_mul(inv, zzz);
    const fe iz = fe_mul(tr_load(sZZ, m, r, hsh), izzz);  // 1/Z
    const fe px = fe_mul(tr_load(sX, m, r, hsh), fe_sqr(iz));
    const fe py = fe_mul(tr_load(sY, m, r, hsh), izzz);
    tr_store(sX, m, r, hsh, px);
    tr_store(sY, m, r, hsh, py);
    st_u256(cpx + 4 * (512 * hsh + r), fe_pack(fe_from_mont(px)));
    st_u256(cpy + 4 * (512 * hsh + r), fe_pack(fe_from_mont(py)));
  }
  // pass C: one inversion for the chord denominators of all addition rows
  run = FE_ONE_M;
  for (int block = 0; block < 2; ++block) {
    u256 s = ld_u256((block ? y : x) + 4 * hsh);
    for (int j = 0; j < 252; ++j) {
      const int r = 256 * block + j;
      if (s.w[0] & 1u) {
        const aff c = ld_aff(bits + 252 * block + j);
        const fe dx = fe_carry(fe_sub(tr_load(sX, m, r, hsh), c.x));
        tr_store(sZZ, m, r, hsh, dx);
        tr_store(sPre, m, r, hsh, run);
        run = fe_mul(run, dx);
      }
#pragma unroll
      for (int q = 0; q < 7; ++q) s.w[q] = (s.w[q] >> 1) | (s.w[q + 1] << 31);
      s.w[7] >>= 1;
    }
  }
  inv = fe_inv(run);
  u256 zero;
#pragma unroll
  for (int q = 0; q < 8; ++q) zero.w[q] = 0;
  for (int block = 1; block >= 0; --block) {
    const u256 s0 = ld_u256((block ? y : x) + 4 * hsh);
    for (int j = 255; j >= 0; --j) {
      const int r = 256 * block + j;
      u256 lam_out = zero;
      if (j < 252 && ((s0.w[j >> 5] >> (j & 31)) & 1u)) {
        const aff c = ld_aff(bits + 252 * block + j);
        const fe dx = tr_load(sZZ, m, r, hsh);
        const fe idx = fe_mul(inv, tr_load(sPre, m, r, hsh));
        inv = fe_mul(inv, dx);
        const fe lam = fe_mul(fe_sub(tr_load(sY, m, r, hsh), c.y), idx);
        lam_out = fe_pack(fe_from_mont(lam));
      }
      st_u256(cl + 4 * (512 * hsh + r), lam_out);
    }
  }
}

// ---- EC-ladder AIR (the ECDSA builtin's building block, signature.py:176-190) --------------------
// Columns m, px, py, qx, qy, la, ld; 256 rows per mimic_ec_mult_air instance (oracle/stark_ref.py
// ec_ladder_trace).  Witness generation, one thread per ladder: the doubling chain in Jacobian
// coordinates, the partial sums in XYZZ, and four batched inversions per ladder (1/Z of the
// doubled points, 1/(2 qy) for the tangent slopes, 1/ZZZ of the partial sums, 1/(px - qx) for the
// chord slopes).  Eight scratch planes [row][limb][ladder].
__global__ void __launch_bounds__(64)
ec_ladder_trace_kernel(const uint64_t* __restrict__ pm, const uint64_t* __restrict__ pqx,
                       const uint64_t* __restrict__ pqy, size_t K, aff_packed shift,
                       uint64_t* __restrict__ cols /* [7][256 K] plain */, int32_t* __restrict__ sc) {
  const size_t h = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= K) return;
  const size_t n = 256 * K;
  const size_t plane = (size_t)256 * NL * K;
  int32_t *qX = sc, *qY = sc + plane, *qZ = sc + 2 * plane, *pX = sc + 3 * plane, *pY = sc + 4 * plane,
          *pZZ = sc + 5 * plane, *pZZZ = sc + 6 * plane, *pre = sc + 7 * plane;
  uint64_t *cm = cols, *cpx = cols + 4 * n, *cpy = cols + 8 * n, *cqx = cols + 12 * n, *cqy = cols + 16 * n,
           *cla = cols + 20 * n, *cld = cols + 24 * n;
  const size_t row0 = 256 * h;
  const u256 m0 = ld_u256(pm + 4 * h);
  u256 zero;
#pragma unroll
  for (int q = 0; q < 8; ++q) zero.w[q] = 0;
  // m column
  {
    u256 s = m0;
    for (int j = 0; j < 256; ++j) {
      st_u256(cm + 4 * (row0 + j), s);
      if (j < 255) {
#pragma unroll
        for (int q = 0; q < 7; ++q) s.w[q] = (s.w[q] >> 1) | (s.w[q + 1] << 31);
        s.w[7] >>= 1;
      }
    }
  }
  // phase 1: doubling chain, Jacobian -> affine with one inversion
  jac Q;
  Q.X = fe_to_mont(fe_unpack(ld_u256(pqx + 4 * h)));
  Q.Y = fe_to_mont(fe_unpack(ld_u256(pqy + 4 * h)));
  Q.Z = FE_ONE_M;
  fe run = FE_ONE_M;
  for (int j = 0; j < 256; ++j) {
    tr_store(qX, K, j, h, Q.X);
    tr_store(qY, K, j, h, Q.Y);
    tr_store(qZ, K, j, h, Q.Z);
    tr_store(pre, K, j, h, run);
    run = fe_mul(run, Q.Z);
    if (j < 255) Q = jac_dbl(Q, FE_ONE_M);
  }
  fe inv = fe_inv(run);
  for (int j = 255; j >= 0; --j) {
    const fe z = tr_load(qZ, K, j, h);
    const fe iz = fe_mul(inv, tr_load(pre, K, j, h));
    inv = fe_mul(inv, z);
    const fe iz2 = fe_sqr(iz);
    const fe ax = fe_mul(tr_load(qX, K, j, h), iz2);
    const fe ay = fe_mul(tr_load(qY, K, j, h), fe_mul(iz2, iz));
    tr_store(qX, K, j, h, ax);
    tr_store(qY, K, j, h, ay);
    st_u256(cqx + 4 * (row0 + j), fe_pack(fe_from_mont(ax)));
    st_u256(cqy + 4 * (row0 + j), fe_pack(fe_from_mont(ay)));
  }
  // phase 2: tangent slopes ld = (3 qx^2 + 1) / (2 qy), rows 0..254
  run = FE_ONE_M;
  for (int j = 0; j < 255; ++j) {
    tr_store(pre, K, j, h, run);
    run = fe_mul(run, fe_carry(fe_dbl(tr_load(qY, K, j, h))));
  }
  inv = fe_inv(run);
  st_u256(cld + 4 * (row0 + 255), zero);
  for (int j = 254; j >= 0; --j) {
    const fe d = fe_carry(fe_dbl(tr_load(qY, K, j, h)));
    const fe id = fe_mul(inv, tr_load(pre, K, j, h));
    inv = fe_mul(inv, d);
    const fe xx = fe_sqr(tr_load(qX, K, j, h));
    const fe num = fe_carry(fe_add(fe_carry(fe_add(fe_dbl(xx), xx)), FE_ONE_M));
    st_u256(cld + 4 * (row0 + j), fe_pack(fe_from_mont(fe_mul(num, id))));
  }
  // phase 3: partial sums in XYZZ, then affine with one inversion
  xyzz acc = xyzz_from_aff(ld_aff(&shift));
  for (int j = 0; j < 256; ++j) {
    tr_store(pX, K, j, h, acc.X);
    tr_store(pY, K, j, h, acc.Y);
    tr_store(pZZ, K, j, h, acc.ZZ);
    tr_store(pZZZ, K, j, h, acc.ZZZ);
    if (j < 251 && ((m0.w[j >> 5] >> (j & 31)) & 1u)) {
      aff q;
      q.x = tr_load(qX, K, j, h);
      q.y = tr_load(qY, K, j, h);
      acc = xyzz_madd(acc, q);
    }
  }
  run = FE_ONE_M;
  for (int j = 0; j < 256; ++j) {
    tr_store(pre, K, j, h, run);
    run = fe_mul(run, tr_load(pZZZ, K, j, h));
  }
  inv = fe_inv(run);
  for (int j = 255; j >= 0; --j) {
    const fe zzz = tr_load(pZZZ, K, j, h);
    const fe izzz = fe_mul(inv, tr_load(pre, K, j, h));
    inv = fe_mul(inv, zzz);
    const fe iz = fe_mul(tr_load(pZZ, K, j, h), izzz);
    const fe ax = fe_mul(tr_load(pX, K, j, h), fe_sqr(iz));
    const fe ay = fe_mul(tr_load(pY, K, j, h), izzz);
    tr_store(pX, K, j, h, ax);
    tr_store(pY, K, j, h, ay);
    st_u256(cpx + 4 * (row0 + j), fe_pack(fe_from_mont(ax)));
    st_u256(cpy + 4 * (row0 + j), fe_pack(fe_from_mont(ay)));
  }
  // phase 4: chord slopes la = (py - qy) / (px - qx) on the addition rows
  run = FE_ONE_M;
  for (int j = 0; j < 251; ++j) {
    if ((m0.w[j >> 5] >> (j & 31)) & 1u) {
      const fe dx = fe_carry(fe_sub(tr_load(pX, K, j, h), tr_load(qX, K, j, h)));
      tr_store(pZZ, K, j, h, dx);
      tr_store(pre, K, j, h, run);
      run = fe_mul(run, dx);
    }
  }
  inv = fe_inv(run);
  for (int j = 255; j >= 0; --j) {
    u256 out = zero;
    if (j < 251 && ((m0.w[j >> 5] >> (j & 31)) & 1u)) {
      const fe dx = tr_load(pZZ, K, j, h);
      const fe idx = fe_mul(inv, tr_load(pre, K, j, h));
      inv = fe_mul(inv, dx);
      const fe lam = fe_mul(fe_sub(tr_load(pY, K, j, h), tr_load(qY, K, j, h)), idx);
      out = fe_pack(fe_from_mont(lam));
    }
    st_u256(cla + 4 * (row0 + j), out);
  }
}

// ---- ECDSA-verification AIR (what verify() mimics, signature.py:217-260) --------------
// Ten columns m, px, py, qx, qy, la, ld, cx, cy, cr; 1024 rows per verification: three EC ladders
// (z G from -SHIFT, r Q from +SHIFT, w B from +SHIFT with B = zG + rQ) in rows 0..767, the last 256 rows
// idle (oracle/stark_ref.py ecdsa_trace is the definition).  Witness generation, one thread per
// verification: the ladder code of ec_ladder_trace_kernel run three times in a loop (the third ladder's
// base point is the sum of the first two outputs), then the two linking slopes and the carry columns.
__global__ void __launch_bounds__(64)
ecdsa_trace_kernel(const uint64_t* __restrict__ pz, const uint64_t* __restrict__ pr, const uint64_t* __restrict__ pw,
                   const uint64_t* __restrict__ pqx, const uint64_t* __restrict__ pqy, size_t K, aff_packed shift,
                   aff_packed gen, uint64_t* __restrict__ cols /* [10][1024 K] plain */, int32_t* __restrict__ sc) {
  const size_t h = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= K) return;
  const size_t n = 1024 * K;
  const size_t plane = (size_t)256 * NL * K;
  int32_t *qX = sc, *qY = sc + plane, *qZ = sc + 2 * plane, *pX = sc + 3 * plane, *pY = sc + 4 * plane,
          *pZZ = sc + 5 * plane, *pZZZ = sc + 6 * plane, *pre = sc + 7 * plane;
  uint64_t *cm = cols, *cpx = cols + 4 * n, *cpy = cols + 8 * n, *cqx = cols + 12 * n, *cqy = cols + 16 * n,
           *cla = cols + 20 * n, *cld = cols + 24 * n, *ccx = cols + 28 * n, *ccy = cols + 32 * n, *ccr = cols + 36 * n;
  u256 zero;
#pragma unroll
  for (int q = 0; q < 8; ++q) zero.w[q] = 0;
  const aff shift_m = ld_aff(&shift);
  aff out[3];  // ladder outputs (Montgomery affine)
#pragma unroll 1
  for (int blk = 0; blk < 3; ++blk) {
    const size_t row0 = 1024 * h + 256 * (size_t)blk;
    const u256 m0 = ld_u256((blk == 0 ? pz : blk == 1 ? pr : pw) + 4 * h);
    aff base, start = shift_m;
    if (blk == 0) {
      base = ld_aff(&gen);
      start.y = fe_neg(start.y);  // MINUS_SHIFT_POINT
    } else if (blk == 1) {
      base.x = fe_to_mont(fe_unpack(ld_u256(pqx + 4 * h)));
      base.y = fe_to_mont(fe_unpack(ld_u256(pqy + 4 * h)));
    } else {  // B = zG + rQ (affine chord rule; x1 == x2 cannot happen for a signature verify() accepts)
      const fe lam = fe_mul(fe_sub(out[1].y, out[0].y), fe_inv(fe_carry(fe_sub(out[1].x, out[0].x))));
      base.x = fe_carry(fe_sub(fe_sub(fe_sqr(lam), out[0].x), out[1].x));
      base.y = fe_carry(fe_sub(fe_mul(lam, fe_sub(out[0].x, base.x)), out[0].y));
      st_u256(cla + 4 * (1024 * h + 511), fe_pack(fe_from_mont(lam)));
    }
    {  // m column
      u256 s = m0;
      for (int j = 0; j < 256; ++j) {
        st_u256(cm + 4 * (row0 + j), s);
        if (j < 255) {
#pragma unroll
          for (int q = 0; q < 7; ++q) s.w[q] = (s.w[q] >> 1) | (s.w[q + 1] << 31);
          s.w[7] >>= 1;
        }
      }
    }
    // phase 1: doubling chain, Jacobian -> affine with one inversion
    jac Q;
    Q.X = base.x;
    Q.Y = base.y;
    Q.Z = FE_ONE_M;
    fe run = FE_ONE_M;
    for (int j = 0; j < 256; ++j) {
      tr_store(qX, K, j, h, Q.X);
      tr_store(qY, K, j, h, Q.Y);
      tr_store(qZ, K, j, h, Q.Z);
      tr_store(pre, K, j, h, run);
      run = fe_mul(run, Q.Z);
      if (j < 255) Q = jac_dbl(Q, FE_ONE_M);
    }
    fe inv = fe_inv(run);
    for (int j = 255; j >= 0; --j) {
      const fe z = tr_load(qZ, K, j, h);
      const fe iz = fe_mul(inv, tr_load(pre, K, j, h));
      inv = fe_mul(inv, z);
      const fe iz2 = fe_sqr(iz);
      const fe ax = fe_mul(tr_load(qX, K, j, h), iz2);
      const fe ay = fe_mul(tr_load(qY, K, j, h), fe_mul(iz2, iz));
      tr_store(qX, K, j, h, ax);
      tr_store(qY, K, j, h, ay);
      st_u256(cqx + 4 * (row0 + j), fe_pack(fe_from_mont(ax)));
      st_u256(cqy + 4 * (row0 + j), fe_pack(fe_from_mont(ay)));
    }
    // phase 2: tangent slopes ld = (3 qx^2 + 1) / (2 qy), rows 0..254
    run = FE_ONE_M;
    for (int j = 0; j < 255; ++j) {
      tr_store(pre, K, j, h, run);
      run = fe_mul(run, fe_carry(fe_dbl(tr_load(qY, K, j, h))));
    }
    inv = fe_inv(run);
    st_u256(cld + 4 * (row0 + 255), zero);
    for (int j = 254; j >= 0; --j) {
      const fe d = fe_carry(fe_dbl(tr_load(qY, K, j, h)));
      const fe id = fe_mul(inv, tr_load(pre, K, j, h));
      inv = fe_mul(inv, d);
      const fe xx = fe_sqr(tr_load(qX, K, j, h));
      const fe num = fe_carry(fe_add(fe_carry(fe_add(fe_dbl(xx), xx)), FE_ONE_M));
      st_u256(cld + 4 * (row0 + j), fe_pack(fe_from_mont(fe_mul(num, id))));
    }
    // phase 3: partial sums in XYZZ, then affine with one inversion
    xyzz acc = xyzz_from_aff(start);
    for (int j = 0; j < 256; ++j) {
      tr_store(pX, K, j, h, acc.X);
      tr_store(pY, K, j, h, acc.Y);
      tr_store(pZZ, K, j, h, acc.ZZ);
      tr_store(pZZZ, K, j, h, acc.ZZZ);
      if (j < 251 && ((m0.w[j >> 5] >> (j & 31)) & 1u)) {
        aff q;
        q.x = tr_load(qX, K, j, h);
        q.y = tr_load(qY, K, j, h);
        acc = xyzz_madd(acc, q);
      }
    }
    run = FE_ONE_M;
    for (int j = 0; j < 256; ++j) {
      tr_store(pre, K, j, h, run);
      run = fe_mul(run, tr_load(pZZZ, K, j, h));
    }
    inv = fe_inv(run);
    for (int j = 255; j >= 0; --j) {
      const fe zzz = tr_load(pZZZ, K, j, h);
      const fe izzz = fe_mul(inv, tr_load(pre, K, j, h));
      inv = fe_mul(inv, zzz);
      const fe iz = fe_mul(tr_load(pZZ, K, j, h), izzz);
      const fe ax = fe_mul(tr_load(pX, K, j, h), fe_sqr(iz));
      const fe ay = fe_mul(tr_load(pY, K, j, h), izzz);
      tr_store(pX, K, j, h, ax);
      tr_store(pY, K, j, h, ay);
      st_u256(cpx + 4 * (row0 + j), fe_pack(fe_from_mont(ax)));
      st_u256(cpy + 4 * (row0 + j), fe_pack(fe_from_mont(ay)));
      if (j == 255) { out[blk].x = ax; out[blk].y = ay; }
    }
    // phase 4: chord slopes la = (py - qy) / (px - qx) on the addition rows
    run = FE_ONE_M;
    for (int j = 0; j < 251; ++j) {
      if ((m0.w[j >> 5] >> (j & 31)) & 1u) {
        const fe dx = fe_carry(fe_sub(tr_load(pX, K, j, h), tr_load(qX, K, j, h)));
        tr_store(pZZ, K, j, h, dx);
        tr_store(pre, K, j, h, run);
        run = fe_mul(run, dx);
      }
    }
    inv = fe_inv(run);
    for (int j = 255; j >= 0; --j) {
      u256 o = zero;
      if (j < 251 && ((m0.w[j >> 5] >> (j & 31)) & 1u)) {
        const fe dx = tr_load(pZZ, K, j, h);
        const fe idx = fe_mul(inv, tr_load(pre, K, j, h));
        inv = fe_mul(inv, dx);
        const fe lam = fe_mul(fe_sub(tr_load(pY, K, j, h), tr_load(qY, K, j, h)), idx);
        o = fe_pack(fe_from_mont(lam));
      }
      if (!(blk == 1 && j == 255)) st_u256(cla + 4 * (row0 + j), o);  // row 511 receives the linking slope below
    }
  }
  // linking slope at row 767: wB + (-SHIFT), whose x must be r
  {
    const fe lam = fe_mul(fe_add(out[2].y, shift_m.y), fe_inv(fe_carry(fe_sub(out[2].x, shift_m.x))));
    st_u256(cla + 4 * (1024 * h + 767), fe_pack(fe_from_mont(lam)));
  }
  // carries and the idle block
  const u256 zgx = fe_pack(fe_from_mont(out[0].x)), zgy = fe_pack(fe_from_mont(out[0].y));
  const u256 rr = ld_u256(pr + 4 * h);
  for (int i = 0; i < 1024; ++i) {
    const size_t row = 1024 * h + i;
    const bool b1 = i >= 256 && i < 512;
    st_u256(ccx + 4 * row, b1 ? zgx : zero);
    st_u256(ccy + 4 * row, b1 ? zgy : zero);
    st_u256(ccr + 4 * row, (i >= 256 && i < 768) ? rr : zero);
    if (i >= 768) {
      st_u256(cm + 4 * row, zero); st_u256(cpx + 4 * row, zero); st_u256(cpy + 4 * row, zero);
      st_u256(cqx + 4 * row, zero); st_u256(cqy + 4 * row, zero); st_u256(cla + 4 * row, zero);
      st_u256(cld + 4 * row, zero);
    }
  }
}

struct EcdsaAirParams {
  fe alpha[26];  // Montgomery
  fe zinv[4];    // R * Montgomery form of 1 / (x^n - 1)
  fe sx, sy, gx, gy, beta;  // plain
};

// Composition of the ECDSA-verification AIR (26 constraints, oracle/stark_ref.py ecdsa_constraint_values);
// plain operands, Montgomery constants (see the file comment).  per: 12 tables of 4096 plain felts.
// SP_AIR_ECDSA_WAVES: the same occupancy switch as ecdsa.hip's SP_VERIFY_WAVES.  0 = the allocator's choice (256 VGPRs +
// 24 AGPRs, one wave per SIMD); 2 = stay within 256 registers: the composition of 4096 verifications (2^24 points)
// 8.36 -> 5.52 ms (profiles/r06_verify_occupancy.txt; same output digest).  2 is the default since round 6.
#ifndef SP_AIR_ECDSA_WAVES
#define SP_AIR_ECDSA_WAVES 2
#endif
#if SP_AIR_ECDSA_WAVES > 0
#define SP_AIR_ECDSA_OCCUPANCY __attribute__((amdgpu_waves_per_eu(SP_AIR_ECDSA_WAVES, SP_AIR_ECDSA_WAVES)))
#else
#define SP_AIR_ECDSA_OCCUPANCY
#endif
__global__ void __launch_bounds__(256) SP_AIR_ECDSA_OCCUPANCY
air_eval_ecdsa_kernel(const uint64_t* __restrict__ trace /* [10][M] plain */, const uint64_t* __restrict__ per,
                      size_t M, EcdsaAirParams prm, uint64_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const size_t in = (i + 4) & (M - 1);
  auto col = [&](int c, size_t r) { return ld_fe_packed(trace + 4 * ((size_t)c * M + r)); };
  auto pcol = [&](int c) { return ld_fe_packed(per + 4 * ((size_t)c * 4096 + (i & 4095))); };
  const fe m = col(0, i), px = col(1, i), py = col(2, i), qx = col(3, i), qy = col(4, i), la = col(5, i),
           ld = col(6, i), cx = col(7, i), cy = col(8, i), cr = col(9, i);
  const fe m_n = col(0, in), px_n = col(1, in), py_n = col(2, in), qx_n = col(3, in), qy_n = col(4, in),
           cx_n = col(7, in), cy_n = col(8, in), cr_n = col(9, in);
  const fe one = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  const fe b = fe_carry(fe_sub(m, fe_dbl(m_n)));
  const fe bM = fe_to_mont(b), laM = fe_to_mont(la), ldM = fe_to_mont(ld), qxM = fe_to_mont(qx), qyM = fe_to_mont(qy);
  const fe nbM = fe_carry(fe_sub(FE_ONE_M, bM));
  const fe qxx = fe_mul(qxM, qx);
  const fe lala = fe_mul(laM, la);
  // ladder rows (selector `step`)
  fe acc = fe_mul(prm.alpha[0], fe_mul(bM, fe_carry(fe_sub(b, one))));
  auto add = [&](int k, const fe& c) { acc = fe_weak_reduce(fe_add(acc, fe_mul(prm.alpha[k], c))); };
  add(1, fe_carry(fe_sub(fe_sub(fe_mul(ldM, fe_carry(fe_dbl(qy))), fe_carry(fe_add(fe_dbl(qxx), qxx))), one)));
  add(2, fe_carry(fe_add(fe_sub(qx_n, fe_mul(ldM, ld)), fe_dbl(qx))));
  add(3, fe_carry(fe_add(fe_sub(qy_n, fe_mul(ldM, fe_sub(qx, qx_n))), qy)));
  add(4, fe_mul(bM, fe_carry(fe_sub(fe_mul(laM, fe_sub(px, qx)), fe_sub(py, qy)))));
  add(5, fe_mul(bM, fe_carry(fe_add(fe_sub(px_n, lala), fe_add(px, qx)))));
  add(6, fe_mul(bM, fe_carry(fe_add(fe_sub(py_n, fe_mul(laM, fe_sub(px, px_n))), py))));
  add(7, fe_mul(nbM, fe_carry(fe_sub(px_n, px))));
  add(8, fe_mul(nbM, fe_carry(fe_sub(py_n, py))));
  fe total = fe_mul(pcol(0), acc);  // every term below carries the same single 1/R (absorbed by zinv)
  auto term = [&](const fe& selector, const fe& value) { total = fe_weak_reduce(fe_add(total, fe_mul(selector, value))); };
  const fe first = pcol(1);
  term(first, fe_mul(prm.alpha[9], fe_carry(fe_sub(px, prm.sx))));
  // alpha10 (first * py - start_y): start_y is plain, so alpha10 * start_y is plain and must lose one R like the rest
  term(first, fe_mul(prm.alpha[10], py));
  total = fe_weak_reduce(fe_sub(total, fe_mul(one, fe_mul(prm.alpha[10], pcol(2)))));
  term(pcol(3), fe_mul(prm.alpha[11], m));
  const fe gbase = pcol(4);
  term(gbase, fe_mul_add_mul(prm.alpha[12], fe_carry(fe_sub(qx, prm.gx)), prm.alpha[13], fe_carry(fe_sub(qy, prm.gy))));
  // on-curve: qy^2 - qx^3 - qx - beta
  term(pcol(5), fe_mul(prm.alpha[14], fe_carry(fe_sub(fe_sub(fe_mul(qyM, qy), fe_mul(qxM, qxx)), fe_add(qx, prm.beta)))));
  term(pcol(6), fe_mul_add_mul(prm.alpha[15], fe_carry(fe_sub(cx_n, px)), prm.alpha[16], fe_carry(fe_sub(cy_n, py))));
  term(pcol(7), fe_mul_add_mul(prm.alpha[17], fe_carry(fe_sub(cx_n, cx)), prm.alpha[18], fe_carry(fe_sub(cy_n, cy))));
  {
    const fe a0 = fe_carry(fe_sub(fe_mul(laM, fe_sub(px, cx)), fe_sub(py, cy)));
    const fe a1 = fe_carry(fe_add(fe_sub(qx_n, lala), fe_add(px, cx)));
    const fe a2 = fe_carry(fe_add(fe_sub(qy_n, fe_mul(laM, fe_sub(px, qx_n))), py));
    term(pcol(8), fe_mul3_add(prm.alpha[19], a0, prm.alpha[20], a1, prm.alpha[21], a2));
  }
  term(pcol(9), fe_mul(prm.alpha[22], fe_carry(fe_sub(cr, m))));
  term(pcol(10), fe_mul(prm.alpha[23], fe_carry(fe_sub(cr_n, cr))));
  {
    const fe f0 = fe_carry(fe_sub(fe_mul(laM, fe_sub(px, prm.sx)), fe_add(py, prm.sy)));
    const fe f1 = fe_carry(fe_add(fe_sub(cr, lala), fe_add(px, prm.sx)));
    term(pcol(11), fe_mul_add_mul(prm.alpha[24], f0, prm.alpha[25], f1));
  }
  st_u256(out + 4 * i, fe_pack(fe_canon(fe_mul(total, prm.zinv[i & 3]))));
}

struct EcAirParams {
  fe alpha[12];  // Montgomery
  fe zinv[4];    // R * Montgomery form of 1 / (x^n - 1) for i mod 4 (absorbs the one R the selectors leave)
  fe shift_x, shift_y;  // plain
};

// Operands are PLAIN felts.  Only what is multiplied by another variable is converted (b, la, ld, qx:
// four conversions instead of fifteen); products with one Montgomery factor are plain again, the
// alphas are Montgomery constants, and the single 1/R left by `selector * (...)` is absorbed by zinv.
__global__ void __launch_bounds__(256)
air_eval_ec_ladder_kernel(const uint64_t* __restrict__ trace /* [7][M] plain */,
                          const uint64_t* __restrict__ per /* [3][1024] plain */, size_t M, EcAirParams prm,
                          uint64_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const size_t in = (i + 4) & (M - 1);
  auto col = [&](int c, size_t r) { return ld_fe_packed(trace + 4 * ((size_t)c * M + r)); };
  auto pcol = [&](int c) { return ld_fe_packed(per + 4 * ((size_t)c * 1024 + (i & 1023))); };
  const fe m = col(0, i), px = col(1, i), py = col(2, i), qx = col(3, i), qy = col(4, i), la = col(5, i),
           ld = col(6, i);
  const fe m_n = col(0, in), px_n = col(1, in), py_n = col(2, in), qx_n = col(3, in), qy_n = col(4, in);
  const fe step = pcol(0), first = pcol(1), z251 = pcol(2);
  const fe one = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  const fe b = fe_carry(fe_sub(m, fe_dbl(m_n)));
  const fe bM = fe_to_mont(b), laM = fe_to_mont(la), ldM = fe_to_mont(ld), qxM = fe_to_mont(qx);
  const fe nbM = fe_carry(fe_sub(FE_ONE_M, bM));
  fe c[9];
  c[0] = fe_mul(bM, fe_carry(fe_sub(b, one)));
  // doubling: ld 2 qy - 3 qx^2 - 1 ; qx' - ld^2 + 2 qx ; qy' - ld (qx - qx') + qy
  const fe qxx = fe_mul(qxM, qx);
  c[1] = fe_carry(fe_sub(fe_sub(fe_mul(ldM, fe_carry(fe_dbl(qy))), fe_carry(fe_add(fe_dbl(qxx), qxx))), one));
  c[2] = fe_carry(fe_add(fe_sub(qx_n, fe_mul(ldM, ld)), fe_dbl(qx)));
  c[3] = fe_carry(fe_add(fe_sub(qy_n, fe_mul(ldM, fe_sub(qx, qx_n))), qy));
  // addition (gated by b): la (px - qx) - (py - qy) ; px' - la^2 + px + qx ; py' - la (px - px') + py
  c[4] = fe_mul(bM, fe_carry(fe_sub(fe_mul(laM, fe_sub(px, qx)), fe_sub(py, qy))));
  c[5] = fe_mul(bM, fe_carry(fe_add(fe_sub(px_n, fe_mul(laM, la)), fe_add(px, qx))));
  c[6] = fe_mul(bM, fe_carry(fe_add(fe_sub(py_n, fe_mul(laM, fe_sub(px, px_n))), py)));
  c[7] = fe_mul(nbM, fe_carry(fe_sub(px_n, px)));
  c[8] = fe_mul(nbM, fe_carry(fe_sub(py_n, py)));
  fe acc_step = fe_mul(prm.alpha[0], c[0]);
#pragma unroll
  for (int k = 1; k < 9; ++k) acc_step = fe_weak_reduce(fe_add(acc_step, fe_mul(prm.alpha[k], c[k])));
  fe acc = fe_mul(step, acc_step);
  const fe firsts = fe_mul_add_mul(prm.alpha[9], fe_carry(fe_sub(px, prm.shift_x)), prm.alpha[10],
                                   fe_carry(fe_sub(py, prm.shift_y)));
  acc = fe_weak_reduce(fe_add(acc, fe_mul(first, firsts)));
  acc = fe_weak_reduce(fe_add(acc, fe_mul(prm.alpha[11], fe_mul(z251, m))));
  st_u256(out + 4 * i, fe_pack(fe_canon(fe_mul(acc, prm.zinv[i & 3]))));
}

// ---- range-check AIR ------------------------------------------------------------------------------
// What the Cairo range-check builtin asserts (0 <= value < 2^128, the bound every amount / position id /
// nonce of the exchange messages goes through before it is packed) as a one-column AIR by bit
// decomposition: 128 rows per value, v_i = value >> i, so that b_i = v_i - 2 v_{i+1} is the i-th bit.
//   rows i mod 128 != 127 (selector `step`):  b (b - 1) = 0
//   rows i mod 128 == 127 (selector `last`):  v (v - 1) = 0
// oracle/stark_ref.py range_check_constraint_values is the definition.
__global__ void __launch_bounds__(256)
range_check_trace_kernel(const uint64_t* __restrict__ values, size_t n_values, uint64_t* __restrict__ col) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_values * 128) return;
  const uint64_t* v = values + 4 * (i >> 7);
  const unsigned sh = (unsigned)(i & 127), w = sh >> 6, b = sh & 63;
  uint64_t o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned src = k + w;
    const uint64_t lo = src < 4 ? v[src] : 0, hi = src + 1 < 4 ? v[src + 1] : 0;
    o[k] = b ? (lo >> b) | (hi << (64 - b)) : lo;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) col[4 * i + k] = o[k];
}

struct RangeCheckAirParams {
  fe alpha[2];  // Montgomery
  fe zinv[4];   // R * Montgomery form of 1 / (x^n - 1) for i mod 4 (absorbs the one R the selectors leave)
};

// Plain operands, Montgomery constants (see the file comment).  per: step, last - 2 tables of 512 plain felts.
__global__ void __launch_bounds__(256)
air_eval_range_check_kernel(const uint64_t* __restrict__ trace /* [1][M] plain */, const uint64_t* __restrict__ per,
                            size_t M, RangeCheckAirParams prm, uint64_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const fe v = ld_fe_packed(trace + 4 * i), v_n = ld_fe_packed(trace + 4 * ((i + 4) & (M - 1)));
  const fe step = ld_fe_packed(per + 4 * (i & 511)), last = ld_fe_packed(per + 4 * (512 + (i & 511)));
  const fe one = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  const fe b = fe_carry(fe_sub(v, fe_dbl(v_n)));
  const fe c0 = fe_mul(fe_to_mont(b), fe_carry(fe_sub(b, one)));
  const fe c1 = fe_mul(fe_to_mont(v), fe_carry(fe_sub(v, one)));
  const fe acc = fe_mul_add_mul(step, fe_mul(prm.alpha[0], c0), last, fe_mul(prm.alpha[1], c1));
  st_u256(out + 4 * i, fe_pack(fe_canon(fe_mul(acc, prm.zinv[i & 3]))));
}

struct AirParams {
  fe alpha[11];  // Montgomery
  fe zinv[4];    // R * Montgomery form of 1 / (x^n - 1) for i mod 4 (absorbs the one R the selectors leave)
  fe shift_x, shift_y;  // plain
};

// Composition column on the LDE coset: sum_k alpha_k C_k(x) / Z_H(x).  Reads the four trace columns
// at i and i + blowup (the next trace row) and six periodic tables at i mod 2048 - all PLAIN felts.
// Only b and lambda (the two variables that multiply other variables) are converted to Montgomery
// form: 2 conversions instead of the 13 + 1 of round 1 (see the file comment).
// Row shards (multi-GPU): the kernel evaluates `M` points whose global index starts at row0;
// the columns are `col_stride` felts apart and, when wrap == 0, carry a halo of one trace row (4 LDE
// rows) after the M points (received from the rank that owns the next rows).
// Block-cyclic row shards (log_block >= 0): the M local points are blocks of B = 2^log_block consecutive
// LDE rows, local block t being global block t * blk_mul + blk_add (blk_mul = ranks, blk_add = this rank);
// every block is stored with its own halo, B + 4 rows apart.
// SP_AIR_WAVES: occupancy switch of the Pedersen-step composition (177 VGPRs: two waves per SIMD; 3 needs <= 168).
// A/B build only (profiles/r06_verify_occupancy.txt): 0 = the allocator's choice.
#ifndef SP_AIR_WAVES
#define SP_AIR_WAVES 0
#endif
#if SP_AIR_WAVES > 0
#define SP_AIR_OCCUPANCY __attribute__((amdgpu_waves_per_eu(SP_AIR_WAVES, SP_AIR_WAVES)))
#else
#define SP_AIR_OCCUPANCY
#endif
__global__ void __launch_bounds__(256) SP_AIR_OCCUPANCY
air_eval_kernel(const uint64_t* __restrict__ trace /* [4][col_stride] plain */, const uint64_t* __restrict__ per /* [6][2048] plain */,
                size_t M, size_t col_stride, size_t row0, int wrap, AirParams prm, uint64_t* __restrict__ out /* [M] plain */,
                int log_block, size_t blk_mul, size_t blk_add) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  size_t in = wrap ? ((i + 4) & (M - 1)) : i + 4;
  size_t gi = row0 + i;  // global LDE index: periodic tables and 1 / Z_H repeat with it
  size_t ii = i;         // where the point's own row is stored
  if (log_block >= 0) {
    const size_t t = i >> log_block, off = i & (((size_t)1 << log_block) - 1);
    ii = t * (((size_t)1 << log_block) + 4) + off;
    in = ii + 4;
    gi = ((t * blk_mul + blk_add) << log_block) + off;
  }
  auto col = [&](int c, size_t r) { return ld_fe_packed(trace + 4 * ((size_t)c * col_stride + r)); };
  auto pcol = [&](int c) { return ld_fe_packed(per + 4 * ((size_t)c * 2048 + (gi & 2047))); };
  const fe s = col(0, ii), px = col(1, ii), py = col(2, ii), lam = col(3, ii);
  const fe s_n = col(0, in), px_n = col(1, in), py_n = col(2, in);
  const fe cx = pcol(0), cy = pcol(1), step = pcol(2), mid = pcol(3), end = pcol(4), z252 = pcol(5);
  const fe one = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  const fe b = fe_carry(fe_sub(s, fe_dbl(s_n)));
  const fe bM = fe_to_mont(b), lamM = fe_to_mont(lam);
  const fe nbM = fe_carry(fe_sub(FE_ONE_M, bM));
  const fe dxn = fe_carry(fe_sub(px_n, px));
  const fe dyn = fe_carry(fe_sub(py_n, py));
  // step rows: bM (alpha0 (b - 1) + alpha1 X1 + alpha2 X2 + alpha3 X3) + nbM (alpha4 dx' + alpha5 dy'): sums
  // of products share one reduction (fe_mul3_add / fe_mul_add_mul), and the common factors bM, nbM
  // are applied once - about 20 multiplication-equivalents per point instead of 27
  const fe X1 = fe_carry(fe_sub(fe_mul(lamM, fe_sub(px, cx)), fe_sub(py, cy)));
  const fe X2 = fe_carry(fe_sub(fe_sub(fe_mul(lamM, lam), px), fe_add(cx, px_n)));
  const fe X3 = fe_carry(fe_sub(fe_mul(lamM, fe_sub(px, px_n)), fe_add(py, py_n)));
  const fe in1 = fe_carry(fe_add(fe_mul(prm.alpha[0], fe_carry(fe_sub(b, one))),
                                 fe_mul3_add(prm.alpha[1], X1, prm.alpha[2], X2, prm.alpha[3], X3)));
  const fe in2 = fe_mul_add_mul(prm.alpha[4], dxn, prm.alpha[5], dyn);
  const fe acc_step = fe_mul_add_mul(bM, in1, nbM, in2);
  const fe mids = fe_mul_add_mul(prm.alpha[6], dxn, prm.alpha[7], dyn);
  const fe ends = fe_mul_add_mul(prm.alpha[8], fe_carry(fe_sub(px_n, prm.shift_x)), prm.alpha[9],
                                 fe_carry(fe_sub(py_n, prm.shift_y)));
  fe acc = fe_mul3_add(step, acc_step, mid, mids, end, ends);
  acc = fe_weak_reduce(fe_add(acc, fe_mul(prm.alpha[10], fe_mul(z252, s))));
  st_u256(out + 4 * i, fe_pack(fe_canon(fe_mul(acc, prm.zinv[gi & 3]))));
}

// FRI fold: g[i] = (f[i] + f[i+M/2]) / 2 + beta (f[i] - f[i+M/2]) / (2 x_i),  x_i = shift w_M^i.
// tw_inv holds w^{-i} (table for size 2^log_tw); c1 = 1/2, c2 = beta / (2 shift); table and constants
// in Montgomery form, the layer values plain: two multiplications and one halving per output.
// Row shards: `fa` / `fb` hold f at global indices i0 .. i0 + count and i0 + M/2 .. (the second array comes
// from the rank that owns the upper half of the layer); a whole layer is fa = f, fb = f + M/2, i0 = 0.
// Block-cyclic shards (log_block >= 0): local position i stands for the global position
// ((i >> log_block) * blk_mul + blk_add) * 2^log_block + (i mod 2^log_block) - the pair (i, i + M/2) of such a
// position lives on the same rank, so the fold needs no exchange (starkperp/sharded_prover.py).
__global__ void __launch_bounds__(256)
fri_fold_kernel(const uint64_t* __restrict__ fa, const uint64_t* __restrict__ fb, uint64_t* __restrict__ g,
                int log_m, size_t i0, size_t count, const uint64_t* __restrict__ tw_inv, int log_tw, fe c1, fe c2,
                int log_block, size_t blk_mul, size_t blk_add) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  size_t gi = i0 + i;
  if (log_block >= 0)
    gi = (((i >> log_block) * blk_mul + blk_add) << log_block) + (i & (((size_t)1 << log_block) - 1));
  const fe a = ld_fe_packed(fa + 4 * i);
  const fe b = ld_fe_packed(fb + 4 * i);
  const fe winv = ld_fe_packed(tw_inv + 4 * (gi << (log_tw - log_m)));
  const fe odd = fe_mul(fe_mul(fe_sub(a, b), winv), c2);
  const fe even = fe_half(fe_add(a, b));  // c1 = 1/2: a shift instead of a multiplication
  (void)c1;
  st_u256(g + 4 * i, fe_pack(fe_canon(fe_carry(fe_add(even, odd)))));
}

// ---- host side ----------------------------------------------------------------------------------
struct Tables {
  std::map<std::pair<int, int>, DeviceBuffer> twiddle;        // (log_n, inverse) -> N/2 powers
  std::map<std::pair<int, std::vector<uint32_t>>, DeviceBuffer> coset;  // (log_n, shift words) -> shift^c / n
  // Work buffers are per stream (like the Pedersen scratch): calls in flight on different streams
  // or from different host threads never share one.  The twiddle / coset / bits tables are
  // immutable once built (their builders synchronise before publishing them).
  std::map<hipStream_t, DeviceBuffer> work;           // coefficient column of the NTT / LDE
  std::map<hipStream_t, DeviceBuffer> trace_scratch;  // partial sums / prefix products of the witness generators
  DeviceBuffer bits;   // 504 per-bit constant points for trace generation
  bool bits_ready = false;
};
static Tables g_tab;
void release_stark_state() {
  for (auto& kv : g_tab.twiddle) kv.second.release();
  for (auto& kv : g_tab.coset) kv.second.release();
  g_tab.twiddle.clear();
  g_tab.coset.clear();
  for (auto& kv : g_tab.work) kv.second.release();
  for (auto& kv : g_tab.trace_scratch) kv.second.release();
  g_tab.work.clear();
  g_tab.trace_scratch.clear();
  g_tab.bits.release();
  g_tab.bits_ready = false;
}

static fe h_pow_u64(fe base_m, uint64_t e) {
  fe r = FE_ONE_M;
  for (int i = 63; i >= 0; --i) {
    r = fe_sqr(r);
    if ((e >> i) & 1) r = fe_mul(r, base_m);
  }
  return r;
}
// primitive 2^log_n-th root of unity: 3^((p-1) / 2^log_n), (p-1) = 2^192 (2^59 + 17)
static fe h_root_of_unity(int log_n) {
  const fe three = fe_to_mont(fe{{3, 0, 0, 0, 0, 0, 0, 0, 0}});
  fe c = h_pow_u64(three, ((uint64_t)1 << 59) + 17);
  for (int i = 0; i < 192 - log_n; ++i) c = fe_sqr(c);
  return c;
}
static fe h_inv(const fe& a) { return fe_inv(a); }

static int build_powers(DeviceBuffer& buf, size_t count, fe base_m, fe factor_m, hipStream_t st) {
  SP_HIP(buf.reserve(count * 32 + 64));
  int nbits = 0;
  while (((size_t)1 << nbits) < count) ++nbits;
  if (nbits == 0) nbits = 1;
  std::vector<fe> pw(nbits);
  fe cur = base_m;
  for (int b = 0; b < nbits; ++b) { pw[b] = fe_mul(cur, FE_ONE_M); cur = fe_sqr(cur); }
  struct DevPowers {  // freed on every exit (the error paths leaked it)
    fe* p = nullptr;
    ~DevPowers() { if (p) (void)hipFree(p); }
  } d_pw;
  SP_HIP(hipMalloc(&d_pw.p, nbits * sizeof(fe)));
  SP_HIP(hipMemcpy(d_pw.p, pw.data(), nbits * sizeof(fe), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(powers_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                     (uint64_t*)buf.ptr, count, d_pw.p, nbits, factor_m);
  SP_HIP(hipGetLastError());
  SP_HIP(hipStreamSynchronize(st));
  return SP_OK;
}

static int get_twiddles(int log_n, int inverse, const uint64_t** out, hipStream_t st) {
  auto key = std::make_pair(log_n, inverse);
  auto it = g_tab.twiddle.find(key);
  if (it == g_tab.twiddle.end()) {
    fe w = h_root_of_unity(log_n);
    if (inverse) w = h_inv(w);
    DeviceBuffer buf;
    const size_t count = log_n ? ((size_t)1 << (log_n - 1)) : 1;
    int rc = build_powers(buf, count, w, FE_ONE_M, st);
    if (rc != SP_OK) return rc;
    it = g_tab.twiddle.emplace(key, buf).first;
  }
  *out = (const uint64_t*)it->second.ptr;
  return SP_OK;
}

// G[c] = shift^c / n for c < n (Montgomery, packed): the coset scaling of an LDE, cached per (n, shift, blowup).
static int coset_table(int log_n, int log_blowup, const uint64_t* shift_host, const uint64_t** G, hipStream_t st) {
  const size_t n = (size_t)1 << log_n;
  u256 sh;
  std::memcpy(sh.w, shift_host, 32);
  std::vector<uint32_t> key_words(sh.w, sh.w + 8);
  key_words.push_back((uint32_t)log_blowup);
  auto key = std::make_pair(log_n, key_words);
  auto it = g_tab.coset.find(key);
  if (it == g_tab.coset.end()) {
    const fe shift_m = fe_to_mont(fe_unpack(sh));
    fe nm = fe_to_mont(fe{{(int32_t)(n & LMASK), (int32_t)(n >> LB), 0, 0, 0, 0, 0, 0, 0}});
    DeviceBuffer buf;
    int rc = build_powers(buf, n, shift_m, fe_inv(nm), st);
    if (rc != SP_OK) return rc;
    it = g_tab.coset.emplace(key, buf).first;
  }
  *G = (const uint64_t*)it->second.ptr;
  return SP_OK;
}

// Full transform of `ncols` columns.  dit = 0: natural -> bit-reversed (DIF), dit = 1: bit-reversed ->
// natural.  in/out may alias.  Values are plain integers throughout (see the file comment).  With
// pad_log_b > 0 (dit only) `in` is the bit-reversed coefficient array of 2^(log_n - pad_log_b) felts per
// column and the zero-padded, coset-scaled input of the transform exists only in LDS.
static const bool g_ntt_lazy_store = getenv("STARKPERP_NTT_CANON_ALL") == nullptr;  // A/B switch
// MEASUREMENT switch, never set in production: STARKPERP_NTT_PROBE=copy launches every pass with
// ZERO stages - load, unpack, LDS tile, barrier, pack, store, nothing else - so that a trace shows what the kernel's
// own data path costs per pass (its HBM ceiling).  The results are NOT a transform (tools/ntt_pass_ceiling.py only
// times them; profiles/r06_ntt_pass_ceiling.txt).
static const bool g_ntt_probe_copy = [] {
  const char* e = getenv("STARKPERP_NTT_PROBE");
  return e != nullptr && strcmp(e, "copy") == 0;
}();
// The tile needs more dynamic LDS than a kernel gets by default.  The attribute is per function AND per device, and
// sp_shutdown followed by sp_init on another device keeps this process's statics: the result is cached per device
// (as pedersen.hip's finish_lds_ready does), under its own mutex because the prover's entry points may be entered
// from several host threads.
static int ntt_lds_ready() {
  static std::mutex mu;
  static std::map<int, int> done;  // device -> SP_OK / SP_ERR_HIP
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return SP_ERR_HIP;
  std::lock_guard<std::mutex> lk(mu);
  auto it = done.find(dev);
  if (it != done.end()) return it->second;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_tile_kernel<TILE_LOG>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)ntt_lds_bytes_of(TILE_LOG));
  if (e == hipSuccess && SMALL_TILE_LOG > 0 && SMALL_TILE_LOG != TILE_LOG && ntt_lds_bytes_of(SMALL_TILE_LOG > 0 ? SMALL_TILE_LOG : TILE_LOG) > 65536)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_tile_kernel<(SMALL_TILE_LOG > 0 ? SMALL_TILE_LOG : TILE_LOG)>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)ntt_lds_bytes_of(SMALL_TILE_LOG > 0 ? SMALL_TILE_LOG : TILE_LOG));
  return done[dev] = (e == hipSuccess ? SP_OK : hip_fail(e, "hipFuncSetAttribute(ntt_tile_kernel, MaxDynamicSharedMemorySize)"));
}
// Passes a transform of 2^log_n points takes with tiles of 2^tile_log felts: one contiguous pass + strided passes.
static int ntt_passes(int log_n, int tile_log) {
  if (log_n <= tile_log) return 1;
  const int smax = ntt_strided_max_of(tile_log);
  return 1 + (log_n - tile_log + smax - 1) / smax;
}
// The small tile wherever it costs no extra pass (four blocks per CU overlap their phases better than two), the big
// one otherwise.  pad_log_b: the LDE's zero padding must fit the contiguous pass.
static int pick_tile_log(int log_n, int pad_log_b) {
  if (SMALL_TILE_LOG <= 0 || SMALL_TILE_LOG >= TILE_LOG) return TILE_LOG;
  if (log_n <= SMALL_TILE_LOG) return TILE_LOG;  // one pass either way: rounds 3 - 5's kernel
  if (pad_log_b > SMALL_TILE_LOG) return TILE_LOG;
  return ntt_passes(log_n, SMALL_TILE_LOG) <= ntt_passes(log_n, TILE_LOG) ? SMALL_TILE_LOG : TILE_LOG;
}
// One launch of the tile kernel for the chosen tile size.
#define SP_NTT_LAUNCH(TLV, GRID, ...)                                                                                  \
  do {                                                                                                                 \
    if ((TLV) == TILE_LOG)                                                                                             \
      hipLaunchKernelGGL((ntt_tile_kernel<TILE_LOG>), GRID, dim3(ntt_threads_of(TILE_LOG)), ntt_lds_bytes_of(TILE_LOG), \
                         st, __VA_ARGS__);                                                                             \
    else                                                                                                               \
      hipLaunchKernelGGL((ntt_tile_kernel<(SMALL_TILE_LOG > 0 ? SMALL_TILE_LOG : TILE_LOG)>), GRID,                    \
                         dim3(ntt_threads_of(SMALL_TILE_LOG > 0 ? SMALL_TILE_LOG : TILE_LOG)),                         \
                         ntt_lds_bytes_of(SMALL_TILE_LOG > 0 ? SMALL_TILE_LOG : TILE_LOG), st, __VA_ARGS__);           \
  } while (0)
static int ntt_column(const uint64_t* in, uint64_t* out, int log_n, int inverse, int dit, int use_scale,
                      fe scale, hipStream_t st, unsigned ncols = 1, size_t in_col_stride = 0,
                      size_t out_col_stride = 0, const uint64_t* pad_G = nullptr, int pad_log_b = 0) {
  const uint64_t* tw;
  int rc = get_twiddles(log_n, inverse, &tw, st);
  if (rc != SP_OK) return rc;
  rc = ntt_lds_ready();
  if (rc != SP_OK) return rc;
  if (log_n == 0) {
    // single point: (optionally) scale
    SP_NTT_LAUNCH(TILE_LOG, dim3(1, ncols), in, out, 0, 0, 0, 0, 0, dit, tw, 1, use_scale, scale, in_col_stride, out_col_stride,
                  (const uint64_t*)nullptr, 0, 0);
    SP_HIP(hipGetLastError());
    return SP_OK;
  }
  // pass plan: local pass covers the low min(tile, log_n) stages; the rest in strided passes of <= the tile's maximum
  struct Pass { int log_e, log_t, log_lo, nst, t_first; };
  std::vector<Pass> plan;
  const int tile_log = pick_tile_log(log_n, pad_log_b);
  const int strided_max = ntt_strided_max_of(tile_log);
  const int local = log_n < tile_log ? log_n : tile_log;
  const int rest = log_n - local;
  const int npass = (rest + strided_max - 1) / strided_max;
  std::vector<Pass> strided;
  int lo = local;
  for (int pi = 0; pi < npass; ++pi) {
    const int cnt = (rest - (lo - local) + (npass - pi) - 1) / (npass - pi);
    Pass ps;
    ps.log_e = tile_log;
    ps.log_t = cnt;
    ps.log_lo = lo;
    ps.nst = cnt;
    ps.t_first = dit ? 0 : cnt - 1;
    strided.push_back(ps);
    lo += cnt;
  }
  Pass loc{local, local, 0, local, dit ? 0 : local - 1};
  if (dit) {
    plan.push_back(loc);
    for (auto& ps : strided) plan.push_back(ps);
  } else {
    for (auto it = strided.rbegin(); it != strided.rend(); ++it) plan.push_back(*it);
    plan.push_back(loc);
  }
  if (pad_log_b > 0 && (!dit || pad_log_b > local)) { set_error("LDE padding needs a DIT transform"); return SP_ERR_BAD_ARGUMENT; }
  const uint64_t* src = in;
  size_t src_stride = in_col_stride;
  for (size_t pi = 0; pi < plan.size(); ++pi) {
    const Pass& ps = plan[pi];
    const bool first = pi == 0, last = pi + 1 == plan.size();
    const unsigned blocks = (unsigned)(((size_t)1 << log_n) >> ps.log_e);
    const int pb = first ? pad_log_b : 0;
    SP_NTT_LAUNCH(tile_log, dim3(blocks, ncols), src, out, ps.log_e, ps.log_t, ps.log_lo, g_ntt_probe_copy ? 0 : ps.nst, ps.t_first, dit,
                  tw, log_n, last ? (use_scale ? 1 : 0) : (g_ntt_lazy_store ? 2 : 0), scale, src_stride, out_col_stride,
                  pb ? pad_G : (const uint64_t*)nullptr, pb, pb ? log_n - pad_log_b : 0);
    src = out;
    src_stride = out_col_stride;
  }
  SP_HIP(hipGetLastError());
  return SP_OK;
}

}  // namespace sp

using namespace sp;

extern "C" {

int sp_ntt_dev(const uint64_t* in, uint64_t* out, unsigned log_n, int inverse, void* stream) {
  SP_REQUIRE_READY();
  if (log_n > 26) { set_error("log_n too large"); return SP_ERR_BAD_ARGUMENT; }
  ctx_lock lk(ctx().mu);
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)1 << log_n;
  DeviceBuffer& work = g_tab.work[st];
  SP_HIP(work.reserve(n * 32));
  uint64_t* tmp = (uint64_t*)work.ptr;
  // natural -> natural: DIF into tmp (bit-reversed), then permute
  fe scale = FE_ONE_M;
  if (inverse) {
    fe nm = fe_to_mont(fe{{(int32_t)(n & LMASK), (int32_t)(n >> LB), 0, 0, 0, 0, 0, 0, 0}});
    scale = fe_inv(nm);
  }
  int rc = ntt_column(in, tmp, (int)log_n, inverse, 0, inverse, scale, st);
  if (rc != SP_OK) return rc;
  hipLaunchKernelGGL(bitrev_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tmp, out,
                     (int)log_n);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_lde_dev(const uint64_t* in, uint64_t* out, unsigned ncols, unsigned log_n, unsigned log_blowup,
               const uint64_t* shift_host, void* stream) {
  SP_REQUIRE_READY();
  if (log_n + log_blowup > 26) { set_error("LDE size too large"); return SP_ERR_BAD_ARGUMENT; }
  ctx_lock lk(ctx().mu);
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)1 << log_n, m = n << log_blowup;
  const uint64_t* G;
  {
    int rc = coset_table((int)log_n, (int)log_blowup, shift_host, &G, st);
    if (rc != SP_OK) return rc;
  }
  // all columns go through each pass together (grid.y = column): 7 launches instead of 7 per column
  DeviceBuffer& work = g_tab.work[st];
  SP_HIP(work.reserve((size_t)ncols * n * 32));
  uint64_t* coef = (uint64_t*)work.ptr;
  if (ncols > 65535) { set_error("too many columns"); return SP_ERR_BAD_ARGUMENT; }
  if (ncols > 0) {
    int rc = ntt_column(in, coef, (int)log_n, 1, 0, 0, FE_ONE_M, st, ncols, n, n);  // -> bit-reversed coefficients
    if (rc != SP_OK) return rc;
    if (log_blowup > 0 && (int)log_blowup <= ((int)(log_n + log_blowup) < TILE_LOG ? (int)(log_n + log_blowup) : TILE_LOG)) {
      // coset scaling + zero padding happen inside the first pass of the big transform (LDS only)
      rc = ntt_column(coef, out, (int)(log_n + log_blowup), 0, 1, 0, FE_ONE_M, st, ncols, n, m, G, (int)log_blowup);
    } else {  // no blowup: scale while copying, then transform in place
      hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)((n + 255) / 256), ncols), dim3(256), 0, st, coef, out,
                         (int)log_n, G);
      rc = ntt_column(out, out, (int)log_n, 0, 1, 0, FE_ONE_M, st, ncols, m, m);
    }
    if (rc != SP_OK) return rc;
  }
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_pedersen_trace_dev(const uint64_t* x, const uint64_t* y, size_t n_hashes, uint64_t* cols,
                          void* stream) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  hipStream_t st = (hipStream_t)stream;
  if (!g_tab.bits_ready) {
    // per-bit points = window value 2^b of the w-bit window tables minus the offsets is awkward;
    // recompute the doubling chains on the host like sp_init does.
    std::vector<aff_packed> h(504);
    auto mk = [](const u256& px, const u256& py) {
      aff a;
      a.x = fe_to_mont(fe_unpack(px));
      a.y = fe_to_mont(fe_unpack(py));
      return a;
    };
    auto dbl = [](const aff& a) {
      const fe xx = fe_sqr(a.x);
      const fe num = fe_carry(fe_add(fe_carry(fe_add(fe_dbl(xx), xx)), FE_ONE_M));
      const fe lam = fe_mul(num, fe_inv(fe_carry(fe_dbl(a.y))));
      aff r;
      r.x = fe_mul(fe_carry(fe_sub(fe_sqr(lam), fe_dbl(a.x))), FE_ONE_M);
      r.y = fe_mul(fe_carry(fe_sub(fe_mul(lam, fe_sub(a.x, r.x)), a.y)), FE_ONE_M);
      return r;
    };
    auto pk = [](const aff& a) {
      aff_packed o;
      o.x = fe_pack(fe_canon(fe_mul(a.x, FE_ONE_M)));
      o.y = fe_pack(fe_canon(fe_mul(a.y, FE_ONE_M)));
      return o;
    };
    const aff bases[4] = {mk(PT_P0_X, PT_P0_Y), mk(PT_P1_X, PT_P1_Y), mk(PT_P2_X, PT_P2_Y), mk(PT_P3_X, PT_P3_Y)};
    for (int e2 = 0; e2 < 2; ++e2) {
      aff q = bases[2 * e2];
      for (int j = 0; j < 248; ++j) { h[252 * e2 + j] = pk(q); q = dbl(q); }
      q = bases[2 * e2 + 1];
      for (int j = 0; j < 4; ++j) { h[252 * e2 + 248 + j] = pk(q); q = dbl(q); }
    }
    SP_HIP(g_tab.bits.reserve(504 * sizeof(aff_packed)));
    SP_HIP(hipMemcpy(g_tab.bits.ptr, h.data(), 504 * sizeof(aff_packed), hipMemcpyHostToDevice));
    g_tab.bits_ready = true;
  }
  aff_packed shift;
  shift.x = fe_pack(fe_canon(fe_to_mont(fe_unpack(PT_SHIFT_X))));
  shift.y = fe_pack(fe_canon(fe_to_mont(fe_unpack(PT_SHIFT_Y))));
  DeviceBuffer& scratch = g_tab.trace_scratch[st];
  SP_HIP(scratch.reserve((size_t)5 * 512 * NL * n_hashes * sizeof(int32_t)));
  hipLaunchKernelGGL(pedersen_trace_kernel, dim3((unsigned)((n_hashes + 63) / 64)), dim3(64), 0, st, x, y,
                     n_hashes, (const aff_packed*)g_tab.bits.ptr, shift, cols, (int32_t*)scratch.ptr);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

static int air_eval_launch(const uint64_t* trace_lde, size_t col_stride, size_t n_points, size_t row0, int wrap,
                           const uint64_t* periodic_lde, unsigned log_n, const uint64_t* alphas_host,
                           const uint64_t* shift_host, uint64_t* out, void* stream, int log_block = -1,
                           size_t blk_mul = 1, size_t blk_add = 0) {
  AirParams prm;
  for (int k = 0; k < 11; ++k) {
    u256 a;
    std::memcpy(a.w, alphas_host + 4 * k, 32);
    prm.alpha[k] = fe_to_mont(fe_unpack(a));
  }
  u256 sh;
  std::memcpy(sh.w, shift_host, 32);
  const fe shift_m = fe_to_mont(fe_unpack(sh));
  // x^n on the coset = shift^n * w_4^(i mod 4)
  fe sn = shift_m;
  for (unsigned i = 0; i < log_n; ++i) sn = fe_sqr(sn);
  const fe w4 = h_root_of_unity(2);
  fe wk = FE_ONE_M;
  for (int k = 0; k < 4; ++k) {
    prm.zinv[k] = fe_mul(fe_inv(fe_carry(fe_sub(fe_mul(sn, wk), FE_ONE_M))), FE_R2);  // zinv * R^2
    wk = fe_mul(wk, w4);
  }
  prm.shift_x = fe_unpack(PT_SHIFT_X);
  prm.shift_y = fe_unpack(PT_SHIFT_Y);
  hipLaunchKernelGGL(air_eval_kernel, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     trace_lde, periodic_lde, n_points, col_stride, row0, wrap, prm, out, log_block, blk_mul, blk_add);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_air_eval_dev(const uint64_t* trace_lde, const uint64_t* periodic_lde, unsigned log_n,
                    const uint64_t* alphas_host, const uint64_t* shift_host, uint64_t* out, void* stream) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  const size_t M = (size_t)4 << log_n;
  return air_eval_launch(trace_lde, M, M, 0, 1, periodic_lde, log_n, alphas_host, shift_host, out, stream);
}

// Row shard of the composition column (multi-GPU): n_points LDE points starting at global index row0 (a
// multiple of 4); the four trace columns are col_stride felts apart and hold n_points + 4 rows (the last
// four = the halo received from the owner of the next rows).  log_n is the GLOBAL trace length.
int sp_air_eval_shard_dev(const uint64_t* trace_lde, size_t col_stride, size_t n_points, size_t row0,
                          const uint64_t* periodic_lde, unsigned log_n, const uint64_t* alphas_host,
                          const uint64_t* shift_host, uint64_t* out, void* stream) {
  SP_REQUIRE_READY();
  if ((row0 & 3) != 0 || col_stride < n_points + 4 || row0 + n_points > ((size_t)4 << log_n)) {
    set_error("bad composition shard");
    return SP_ERR_BAD_ARGUMENT;
  }
  ctx_lock lk(ctx().mu);
  return air_eval_launch(trace_lde, col_stride, n_points, row0, 0, periodic_lde, log_n, alphas_host, shift_host, out,
                         stream);
}

int sp_ec_ladder_trace_dev(const uint64_t* m, const uint64_t* qx, const uint64_t* qy, size_t n_ladders,
                           uint64_t* cols, void* stream) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  aff_packed shift;
  shift.x = fe_pack(fe_canon(fe_to_mont(fe_unpack(PT_SHIFT_X))));
  shift.y = fe_pack(fe_canon(fe_to_mont(fe_unpack(PT_SHIFT_Y))));
  DeviceBuffer& scratch = g_tab.trace_scratch[(hipStream_t)stream];
  SP_HIP(scratch.reserve((size_t)8 * 256 * NL * n_ladders * sizeof(int32_t)));
  hipLaunchKernelGGL(ec_ladder_trace_kernel, dim3((unsigned)((n_ladders + 63) / 64)), dim3(64), 0,
                     (hipStream_t)stream, m, qx, qy, n_ladders, shift, cols, (int32_t*)scratch.ptr);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_air_eval_ec_ladder_dev(const uint64_t* trace_lde, const uint64_t* periodic_lde, unsigned log_n,
                              const uint64_t* alphas_host, const uint64_t* shift_host, uint64_t* out,
                              void* stream) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  const size_t n = (size_t)1 << log_n, M = 4 * n;
  EcAirParams prm;
  for (int k = 0; k < 12; ++k) {
    u256 a;
    std::memcpy(a.w, alphas_host + 4 * k, 32);
    prm.alpha[k] = fe_to_mont(fe_unpack(a));
  }
  u256 sh;
  std::memcpy(sh.w, shift_host, 32);
  fe sn = fe_to_mont(fe_unpack(sh));
  for (unsigned i = 0; i < log_n; ++i) sn = fe_sqr(sn);
  const fe w4 = h_root_of_unity(2);
  fe wk = FE_ONE_M;
  for (int k = 0; k < 4; ++k) {
    prm.zinv[k] = fe_mul(fe_inv(fe_carry(fe_sub(fe_mul(sn, wk), FE_ONE_M))), FE_R2);  // zinv * R^2
    wk = fe_mul(wk, w4);
  }
  prm.shift_x = fe_unpack(PT_SHIFT_X);
  prm.shift_y = fe_unpack(PT_SHIFT_Y);
  hipLaunchKernelGGL(air_eval_ec_ladder_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, trace_lde, periodic_lde, M, prm, out);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_range_check_trace_dev(const uint64_t* values, size_t n_values, uint64_t* col, void* stream) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  if (n_values == 0) return SP_OK;
  hipLaunchKernelGGL(range_check_trace_kernel, dim3((unsigned)((n_values * 128 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, values, n_values, col);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_air_eval_range_check_dev(const uint64_t* trace_lde, const uint64_t* periodic_lde, unsigned log_n,
                                const uint64_t* alphas_host, const uint64_t* shift_host, uint64_t* out,
                                void* stream) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  if (log_n < 7 || log_n > 30) {
    set_error("sp_air_eval_range_check_dev: log_n out of range (one value is 128 rows)");
    return SP_ERR_BAD_ARGUMENT;
  }
  const size_t n = (size_t)1 << log_n, M = 4 * n;
  RangeCheckAirParams prm;
  for (int k = 0; k < 2; ++k) {
    u256 a;
    std::memcpy(a.w, alphas_host + 4 * k, 32);
    prm.alpha[k] = fe_to_mont(fe_unpack(a));
  }
  u256 sh;
  std::memcpy(sh.w, shift_host, 32);
  fe sn = fe_to_mont(fe_unpack(sh));
  for (unsigned i = 0; i < log_n; ++i) sn = fe_sqr(sn);
  const fe w4 = h_root_of_unity(2);
  fe wk = FE_ONE_M;
  for (int k = 0; k < 4; ++k) {
    prm.zinv[k] = fe_mul(fe_inv(fe_carry(fe_sub(fe_mul(sn, wk), FE_ONE_M))), FE_R2);  // zinv * R^2
    wk = fe_mul(wk, w4);
  }
  hipLaunchKernelGGL(air_eval_range_check_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, trace_lde, periodic_lde, M, prm, out);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_ecdsa_trace_dev(const uint64_t* z, const uint64_t* r, const uint64_t* w, const uint64_t* qx,
                       const uint64_t* qy, size_t n_sigs, uint64_t* cols, void* stream) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  aff_packed shift, gen;
  shift.x = fe_pack(fe_canon(fe_to_mont(fe_unpack(PT_SHIFT_X))));
  shift.y = fe_pack(fe_canon(fe_to_mont(fe_unpack(PT_SHIFT_Y))));
  gen.x = fe_pack(fe_canon(fe_to_mont(fe_unpack(PT_GEN_X))));
  gen.y = fe_pack(fe_canon(fe_to_mont(fe_unpack(PT_GEN_Y))));
  DeviceBuffer& scratch = g_tab.trace_scratch[(hipStream_t)stream];
  SP_HIP(scratch.reserve((size_t)8 * 256 * NL * n_sigs * sizeof(int32_t)));
  hipLaunchKernelGGL(ecdsa_trace_kernel, dim3((unsigned)((n_sigs + 63) / 64)), dim3(64), 0, (hipStream_t)stream, z, r,
                     w, qx, qy, n_sigs, shift, gen, cols, (int32_t*)scratch.ptr);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_air_eval_ecdsa_dev(const uint64_t* trace_lde, const uint64_t* periodic_lde, unsigned log_n,
                          const uint64_t* alphas_host, const uint64_t* shift_host, uint64_t* out, void* stream) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  const size_t n = (size_t)1 << log_n, M = 4 * n;
  EcdsaAirParams prm;
  for (int k = 0; k < 26; ++k) {
    u256 a;
    std::memcpy(a.w, alphas_host + 4 * k, 32);
    prm.alpha[k] = fe_to_mont(fe_unpack(a));
  }
  u256 sh;
  std::memcpy(sh.w, shift_host, 32);
  fe sn = fe_to_mont(fe_unpack(sh));
  for (unsigned i = 0; i < log_n; ++i) sn = fe_sqr(sn);
  const fe w4 = h_root_of_unity(2);
  fe wk = FE_ONE_M;
  for (int k = 0; k < 4; ++k) {
    prm.zinv[k] = fe_mul(fe_inv(fe_carry(fe_sub(fe_mul(sn, wk), FE_ONE_M))), FE_R2);  // zinv * R^2
    wk = fe_mul(wk, w4);
  }
  prm.sx = fe_unpack(PT_SHIFT_X);
  prm.sy = fe_unpack(PT_SHIFT_Y);
  prm.gx = fe_unpack(PT_GEN_X);
  prm.gy = fe_unpack(PT_GEN_Y);
  prm.beta = fe_unpack(CURVE_BETA);
  hipLaunchKernelGGL(air_eval_ecdsa_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     trace_lde, periodic_lde, M, prm, out);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

// The inverse twiddles of a layer of 2^log_m points are every 2^(log_tw - log_m)-th entry of the table of
// any larger size: a job builds ONE table (for its first, largest layer) and the kernel strides through it.
static int fri_twiddles(int log_m, const uint64_t** tw, int* log_tw, hipStream_t st) {
  int best = -1;
  for (auto& kv : g_tab.twiddle)
    if (kv.first.second == 1 && kv.first.first >= log_m && kv.first.first > best) best = kv.first.first;
  if (best < 0) best = log_m;
  *log_tw = best;
  return get_twiddles(best, 1, tw, st);
}

static int fri_fold_launch(const uint64_t* fa, const uint64_t* fb, uint64_t* out, unsigned log_m, size_t i0,
                           size_t count, const uint64_t* beta_host, const uint64_t* shift_host, void* stream,
                           int log_block = -1, size_t blk_mul = 1, size_t blk_add = 0) {
  hipStream_t st = (hipStream_t)stream;
  const uint64_t* tw;
  int log_tw = 0;
  int rc = fri_twiddles((int)log_m, &tw, &log_tw, st);
  if (rc != SP_OK) return rc;
  u256 b, s;
  std::memcpy(b.w, beta_host, 32);
  std::memcpy(s.w, shift_host, 32);
  const fe two = fe_to_mont(fe{{2, 0, 0, 0, 0, 0, 0, 0, 0}});
  const fe c1 = fe_inv(two);
  const fe c2 = fe_mul(fe_to_mont(fe_unpack(b)), fe_inv(fe_mul(two, fe_to_mont(fe_unpack(s)))));
  if (count == 0) return SP_OK;
  hipLaunchKernelGGL(fri_fold_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, fa, fb, out,
                     (int)log_m, i0, count, tw, log_tw, c1, c2, log_block, blk_mul, blk_add);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_fri_fold_dev(const uint64_t* in, uint64_t* out, unsigned log_m, const uint64_t* beta_host,
                    const uint64_t* shift_host, void* stream) {
  SP_REQUIRE_READY();
  if (log_m < 1 || log_m > 26) { set_error("bad layer size"); return SP_ERR_BAD_ARGUMENT; }
  ctx_lock lk(ctx().mu);
  const size_t half = (size_t)1 << (log_m - 1);
  return fri_fold_launch(in, in + 4 * half, out, log_m, 0, half, beta_host, shift_host, stream);
}

// Row shard of one fold (multi-GPU): out[i] = fold(fa[i], fb[i]) for the global positions i0 .. i0 + count of a
// layer of 2^log_m points; fa holds f[i0 ..], fb holds f[i0 + 2^(log_m - 1) ..].
int sp_fri_fold_shard_dev(const uint64_t* fa, const uint64_t* fb, uint64_t* out, unsigned log_m, size_t i0,
                          size_t count, const uint64_t* beta_host, const uint64_t* shift_host, void* stream) {
  SP_REQUIRE_READY();
  if (log_m < 1 || log_m > 26 || i0 + count > ((size_t)1 << (log_m - 1))) { set_error("bad fold shard"); return SP_ERR_BAD_ARGUMENT; }
  ctx_lock lk(ctx().mu);
  return fri_fold_launch(fa, fb, out, log_m, i0, count, beta_host, shift_host, stream);
}

// Block-cyclic shard of one fold (multi-GPU, starkperp/sharded_prover.py): the rank holds `count` = 2^(log_m - 1)
// / world positions of each half of the layer as blocks of 2^log_block consecutive points, local block t being
// global block t * world + rank.  fa / fb = the rank's part of the lower / upper half; both members of every
// pair are local, so no data moves.  Needs 2^(log_m - 1) >= world * 2^log_block.
int sp_fri_fold_blocks_dev(const uint64_t* fa, const uint64_t* fb, uint64_t* out, unsigned log_m, size_t count,
                           unsigned log_block, unsigned world, unsigned rank, const uint64_t* beta_host,
                           const uint64_t* shift_host, void* stream) {
  SP_REQUIRE_READY();
  const size_t half = log_m >= 1 && log_m <= 26 ? (size_t)1 << (log_m - 1) : 0;
  if (half == 0 || world == 0 || rank >= world || log_block > 26 || count * world != half ||
      (count & (((size_t)1 << log_block) - 1)) != 0) {
    set_error("bad block-cyclic fold shard");
    return SP_ERR_BAD_ARGUMENT;
  }
  ctx_lock lk(ctx().mu);
  return fri_fold_launch(fa, fb, out, log_m, 0, count, beta_host, shift_host, stream, (int)log_block, world, rank);
}

// Block-cyclic shard of the composition column: n_blocks blocks of 2^log_block LDE rows (local block t =
// global block t * world + rank), every block stored with its halo of one trace row - the four trace columns
// are col_stride felts apart and hold n_blocks * (2^log_block + 4) rows.  out: n_blocks * 2^log_block felts.
int sp_air_eval_blocks_dev(const uint64_t* trace_lde, size_t col_stride, size_t n_blocks, unsigned log_block,
                           unsigned world, unsigned rank, const uint64_t* periodic_lde, unsigned log_n,
                           const uint64_t* alphas_host, const uint64_t* shift_host, uint64_t* out, void* stream) {
  SP_REQUIRE_READY();
  const size_t B = (size_t)1 << log_block;
  if (log_block < 2 || log_block > 26 || world == 0 || rank >= world || col_stride < n_blocks * (B + 4) ||
      n_blocks * B * world != ((size_t)4 << log_n)) {
    set_error("bad block-cyclic composition shard");
    return SP_ERR_BAD_ARGUMENT;
  }
  ctx_lock lk(ctx().mu);
  return air_eval_launch(trace_lde, col_stride, n_blocks * B, 0, 0, periodic_lde, log_n, alphas_host, shift_host, out,
                         stream, (int)log_block, world, rank);
}

// The two halves of sp_lde_dev as separate calls, so that the coset transforms of one column share ONE
// interpolation (starkperp/sharded_prover.py: the four coset units of a column).
//   sp_interpolate_dev  evaluations on <w_n> (natural order) -> coefficients in BIT-REVERSED order
//   sp_coset_eval_dev   those coefficients -> evaluations on shift * <w_n> (natural order)
int sp_interpolate_dev(const uint64_t* in, uint64_t* coef, unsigned ncols, unsigned log_n, void* stream) {
  SP_REQUIRE_READY();
  if (log_n > 26 || ncols == 0 || ncols > 65535) { set_error("bad interpolation size"); return SP_ERR_BAD_ARGUMENT; }
  ctx_lock lk(ctx().mu);
  const size_t n = (size_t)1 << log_n;
  return ntt_column(in, coef, (int)log_n, 1, 0, 0, FE_ONE_M, (hipStream_t)stream, ncols, n, n);
}

int sp_coset_eval_dev(const uint64_t* coef, uint64_t* out, unsigned ncols, unsigned log_n, const uint64_t* shift_host,
                      void* stream) {
  SP_REQUIRE_READY();
  if (log_n > 26 || ncols == 0 || ncols > 65535) { set_error("bad coset size"); return SP_ERR_BAD_ARGUMENT; }
  ctx_lock lk(ctx().mu);
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)1 << log_n;
  const uint64_t* G;
  int rc = coset_table((int)log_n, 0, shift_host, &G, st);
  if (rc != SP_OK) return rc;
  hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)((n + 255) / 256), ncols), dim3(256), 0, st, coef, out,
                     (int)log_n, G);
  return ntt_column(out, out, (int)log_n, 0, 1, 0, FE_ONE_M, st, ncols, n, n);
}

}  // extern "C"

// Field arithmetic for the Stark prime p = 2^251 + 17*2^192 + 1 and the curve order N, built for
// the gfx950 VALU: nine SIGNED 29-bit limbs per element and 64-bit signed column accumulators.
//
// Why this shape (measured on MI355X, tools/ubench/valu_rate.hip, profiles/r04_valu_rate_ubench.txt): a
// v_mad_i64_i32 issues every 4.5 - 5.0 cycles per SIMD, a 64-bit add or shift every 4.2 - 4.7, a carry-writing
// add every 4.4, a plain 32-bit and / add / sub every 2.3 - 2.6.  Carry handling in 64 bits costs as much as
// multiplying, so the representation is chosen to have NO carries inside a product: 9 limbs x 29 bits give 81
// multiply-accumulates into 17 columns whose sums stay below 2^63, additions and subtractions are plain limb-wise
// 32-bit adds (lazy, signed), and the Montgomery radix R = 2^261 leaves ~9 bits of value headroom so no
// conditional subtraction is ever needed between multiplications.  p = 1 + 17*2^18 * 2^(6*29) + 2^19 * 2^(8*29)
// has only three non-zero limbs and p = 1 (mod 2^29), so a reduction step is q = c_i mod 2^29 (one v_and)
// followed by two multiply-adds with -P6, -P8 and the 64-bit carry (fe_reduce).
//
// Conventions
//   * "N-form": limbs 0..7 in [0, 2^29), limb 8 small and signed; value in (-4p, 4p).  This is
//     what fe_mul / fe_sqr / fe_carry return.
//   * fe_add / fe_sub are limb-wise and do not normalise.  A product needs
//     sum_j |a_i||b_j| < 2^63; every call site states its bound in units of 2^29 ("B=k").
//   * All elements that take part in multiplications are in Montgomery form (x * R mod p).
//
// The same header compiles for the host (g++) so the arithmetic can be unit-tested without a GPU;
// the product library only ever runs it on the device (plus one-off table seeds at sp_init).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SP_HD __host__ __device__ __forceinline__
#else
#define SP_HD inline
#endif

#if defined(SP_CHECK_BOUNDS) && !defined(__HIPCC__)
#include <stdio.h>
#include <stdlib.h>
#define SP_CHK32(expr64)                                                        \
  do {                                                                          \
    long long v__ = (long long)(expr64);                                        \
    if (v__ > 2147483647LL || v__ < -2147483648LL) {                            \
      fprintf(stderr, "limb overflow at %s:%d\n", __FILE__, __LINE__);          \
      abort();                                                                  \
    }                                                                           \
  } while (0)
#define SP_CHK64(expr128)                                                       \
  do {                                                                          \
    __int128 v__ = (expr128);                                                   \
    if (v__ > (__int128)9223372036854775807LL || v__ < -(__int128)9223372036854775807LL) { \
      fprintf(stderr, "column overflow at %s:%d\n", __FILE__, __LINE__);        \
      abort();                                                                  \
    }                                                                           \
  } while (0)
#else
#define SP_CHK32(e) ((void)0)
#define SP_CHK64(e) ((void)0)
#endif

namespace sp {

constexpr int NL = 9;
constexpr int LB = 29;
constexpr uint32_t LMASK = (1u << LB) - 1u;

struct fe {
  int32_t l[NL];
};

// ---- constants (tools/gen_consts.py prints these; tests/test_field_host.py re-derives them) ----
// p = sum P_LIMB[i] * 2^(29 i)
constexpr int32_t P6 = 0x440000;  // 17 * 2^18
constexpr int32_t P8 = 0x80000;   // 2^19
// R mod p, R^2 mod p with R = 2^261
constexpr fe FE_ONE_M = {{0x1ffffc01, 0x1fffffff, 0x1fffffff, 0x1fffffff, 0x1fffffff, 0x1fffffff,
                          0x1043ffff, 0x1ffffff7, 0x7ffff}};
constexpr fe FE_R2 = {{0x100001, 0x1fae6fc0, 0x1fffffff, 0x9987f, 0x0, 0x1ffedf00, 0x43ffff,
                       0xf004400, 0x752ad}};
constexpr fe FE_ZERO = {{0, 0, 0, 0, 0, 0, 0, 0, 0}};
constexpr fe FE_P = {{1, 0, 0, 0, 0, 0, P6, 0, P8}};

// Optimisation barrier on the limbs of a freshly produced element (device only): pins every limb
// in a 32-bit VGPR.  Without it LLVM folds sext(trunc(x) & mask) back into the 64-bit value x & mask
// and lowers each later product with such a limb as a 64x32-bit multiply (two v_mad_u64_u32 plus
// two v_mov instead of one mad) - measured +35 % instructions in the hash kernel.
SP_HD void fe_pin(fe& r) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int k = 0; k < NL; ++k) asm volatile("" : "+v"(r.l[k]));
#else
  (void)r;
#endif
}

// ---- 256-bit packed <-> limbs ----
struct u256 {
  uint32_t w[8];
};

SP_HD fe fe_unpack(const u256& a) {
  fe r;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int bit = LB * k, wi = bit >> 5, sh = bit & 31;
#if defined(__HIP_DEVICE_COMPILE__)
    // One funnel shift per limb.  Written with the builtin on purpose: the portable 64-bit form
    // below makes LLVM merge the two adjacent word loads into overlapping 64-bit loads, after which
    // it can no longer keep the eight words in registers and round-trips them through LDS / scratch.
    const uint32_t hi = wi + 1 < 8 ? a.w[wi + 1] : 0u;
    r.l[k] = (int32_t)((sh == 0 ? a.w[wi] : __builtin_amdgcn_alignbit(hi, a.w[wi], (uint32_t)sh)) & LMASK);
#else
    uint64_t two = (uint64_t)a.w[wi] | ((uint64_t)(wi + 1 < 8 ? a.w[wi + 1] : 0u) << 32);
    r.l[k] = (int32_t)((uint32_t)(two >> sh) & LMASK);
#endif
  }
  r.l[8] = (int32_t)(a.w[7] >> 8);
  fe_pin(r);
  return r;
}

// Requires canonical limbs (all in [0,2^29), value < 2^256).
SP_HD u256 fe_pack(const fe& a) {
  u256 r;
#pragma unroll
  for (int wi = 0; wi < 8; ++wi) {
    const int bit = 32 * wi, k = bit / LB, sh = bit - k * LB;  // word starts inside limb k
    uint64_t two = (uint64_t)(uint32_t)a.l[k] >> sh;
    two |= (uint64_t)(uint32_t)a.l[k + 1] << (LB - sh);
    if (k + 2 < NL) two |= (uint64_t)(uint32_t)a.l[k + 2] << (2 * LB - sh);
    r.w[wi] = (uint32_t)two;
  }
  return r;
}

// ---- lazy add / sub / small multiples ----
SP_HD fe fe_add(const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) { SP_CHK32((int64_t)a.l[i] + b.l[i]); r.l[i] = a.l[i] + b.l[i]; }
  return r;
}
SP_HD fe fe_sub(const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) { SP_CHK32((int64_t)a.l[i] - b.l[i]); r.l[i] = a.l[i] - b.l[i]; }
  return r;
}
SP_HD fe fe_neg(const fe& a) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = -a.l[i];
  return r;
}
SP_HD fe fe_dbl(const fe& a) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) { SP_CHK32((int64_t)a.l[i] * 2); r.l[i] = a.l[i] * 2; }
  return r;
}

// Signed carry pass: limbs 0..7 -> [0,2^29), limb 8 keeps the (signed) rest.  Value unchanged.
SP_HD fe fe_carry(const fe& a) {
  fe r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    SP_CHK32((int64_t)a.l[i] + c);
    int32_t v = a.l[i] + c;
    r.l[i] = (int32_t)((uint32_t)v & LMASK);
    c = v >> LB;  // arithmetic
  }
  r.l[8] = a.l[8] + c;
  return r;
}

// Weak reduction for long add/sub chains (NTT butterflies): subtracts floor(value / 2^251) * p.
// Input lazy or N-form with |limb 8| < 2^29; output N-form with value in (-p, 2p).
SP_HD fe fe_weak_reduce(const fe& a) {
  const int32_t q = a.l[8] >> 19;
  fe r = a;
  r.l[0] -= q;
  r.l[6] -= q * P6;
  r.l[8] -= q * P8;
  return fe_carry(r);
}

// value / 2 mod p for N-form input with value in (-p, 2p); output N-form with value in (-p, p].
// Adds p when the value is odd (p is odd), then shifts the 261-bit limb string right by one.
SP_HD fe fe_half(const fe& a_in) {
  fe a = fe_carry(a_in);
  const int32_t odd = a.l[0] & 1;
  a.l[0] += odd;        // + p = (1, 0, 0, 0, 0, 0, P6, 0, P8): limb 0 becomes even, may reach 2^29
  a.l[6] += odd * P6;
  a.l[8] += odd * P8;
  a = fe_carry(a);
  fe r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = (int32_t)(((uint32_t)a.l[i] >> 1) | (((uint32_t)a.l[i + 1] & 1u) << (LB - 1)));
  r.l[8] = a.l[8] >> 1;  // arithmetic: the top limb carries the sign
  return r;
}

// ---- column products ----
struct cols {
  int64_t c[17];
};

SP_HD void cols_zero(cols& t) {
#pragma unroll
  for (int i = 0; i < 17; ++i) t.c[i] = 0;
}
// t += a*b   (81 v_mad_i64_i32)
SP_HD void cols_mac(cols& t, const fe& a, const fe& b) {
#pragma unroll
  for (int i = 0; i < NL; ++i)
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      SP_CHK64((__int128)t.c[i + j] + (__int128)a.l[i] * b.l[j]);
      t.c[i + j] += (int64_t)a.l[i] * (int64_t)b.l[j];
    }
}
// t -= a*b
SP_HD void cols_msub(cols& t, const fe& a, const fe& b) {
#pragma unroll
  for (int i = 0; i < NL; ++i)
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      SP_CHK64((__int128)t.c[i + j] - (__int128)a.l[i] * b.l[j]);
      t.c[i + j] -= (int64_t)a.l[i] * (int64_t)b.l[j];
    }
}
// t += a*a   (45 multiply-accumulates)
SP_HD void cols_sqr(cols& t, const fe& a) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    SP_CHK64((__int128)t.c[2 * i] + (__int128)a.l[i] * a.l[i]);
    t.c[2 * i] += (int64_t)a.l[i] * (int64_t)a.l[i];
    SP_CHK32((int64_t)a.l[i] * 2);
    const int32_t d = a.l[i] * 2;
#pragma unroll
    for (int j = i + 1; j < NL; ++j) {
      SP_CHK64((__int128)t.c[i + j] + (__int128)d * a.l[j]);
      t.c[i + j] += (int64_t)d * (int64_t)a.l[j];
    }
  }
}

// Montgomery reduction mod p: returns (T - Q p) / 2^261 in N-form, value in (T/R - p, T/R].
// Round 4: the SUBTRACTIVE form.  p = 1 (mod 2^29), so q_i = c_i mod 2^29 is one v_and_b32 and T - q_i p 2^(29 i)
// clears limb i; what limb i hands on is (c_i - q_i) >> 29 = c_i >> 29 - no "c_i + q" to form first.  The additive
// form of rounds 1 - 3 (q = -c_i, carry = (c_i + q) >> 29) paid a v_sub and a third multiply-add per limb for the
// same thing: 174 -> 156 VALU instructions per fe_mul, 144 -> 126 per fe_sqr (tools/ubench/valu_rate.hip: a
// v_mad_i64_i32 costs 4.5 - 4.9 cycles of issue, a 32-bit and / sub 2.3 - 2.7).
SP_HD fe fe_reduce(cols& t) {
  int32_t np6 = -P6, np8 = -P8;
#if defined(__HIP_DEVICE_COMPILE__)
  // both stay SGPR multiplicands of a v_mad_i64_i32: from a visible -2^19 the compiler derives a 64-bit shift
  // and a 64-bit subtraction (two instructions at the issue cost of a multiply-add each)
  asm volatile("" : "+s"(np6), "+s"(np8));
#endif
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int32_t q = (int32_t)((uint32_t)t.c[i] & LMASK);
    t.c[i + 1] += t.c[i] >> LB;  // arithmetic shift = floor: exactly (c_i - q) / 2^29
    SP_CHK64((__int128)t.c[i + 6] - (__int128)q * P6);
    t.c[i + 6] += (int64_t)q * (int64_t)np6;
    SP_CHK64((__int128)t.c[i + 8] - ((__int128)q << 19));
    t.c[i + 8] += (int64_t)q * (int64_t)np8;  // i + 8 <= 16
  }
  fe r;
  int64_t carry = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t v = t.c[9 + k] + carry;
    r.l[k] = (int32_t)((uint32_t)v & LMASK);
    carry = v >> LB;
  }
  SP_CHK32(carry);
  r.l[8] = (int32_t)carry;
  fe_pin(r);
  return r;
}

SP_HD fe fe_mul(const fe& a, const fe& b) {
  cols t;
  cols_zero(t);
  cols_mac(t, a, b);
  return fe_reduce(t);
}
SP_HD fe fe_sqr(const fe& a) {
  cols t;
  cols_zero(t);
  cols_sqr(t, a);
  return fe_reduce(t);
}
// a*b - c*d with a single reduction
SP_HD fe fe_mul_sub_mul(const fe& a, const fe& b, const fe& c, const fe& d) {
  cols t;
  cols_zero(t);
  cols_mac(t, a, b);
  cols_msub(t, c, d);
  return fe_reduce(t);
}
// a*b + c*d with a single reduction
SP_HD fe fe_mul_add_mul(const fe& a, const fe& b, const fe& c, const fe& d) {
  cols t;
  cols_zero(t);
  cols_mac(t, a, b);
  cols_mac(t, c, d);
  return fe_reduce(t);
}

// a*b + c*d + e*f with a single reduction (all six operands N-form: 27 * 2^58 < 2^63)
SP_HD fe fe_mul3_add(const fe& a, const fe& b, const fe& c, const fe& d, const fe& e, const fe& f) {
  cols t;
  cols_zero(t);
  cols_mac(t, a, b);
  cols_mac(t, c, d);
  cols_mac(t, e, f);
  return fe_reduce(t);
}

// ---- canonical form ----
// a in N-form with value in (-p, 2p)  ->  limbs of the unique representative in [0, p).
SP_HD bool fe_geq_p_canon_limbs(const fe& a) {  // a has non-negative normalised limbs
  // compare with p = (1,0,0,0,0,0,P6,0,P8) from the top limb down - branch-free: the kernels call this once
  // per stored felt and a chain of early returns became a chain of exec-mask regions
  const bool gt8 = a.l[8] > P8, eq8 = a.l[8] == P8;
  const bool gt6 = a.l[6] > P6, eq6 = a.l[6] == P6;
  const bool low = (a.l[5] | a.l[4] | a.l[3] | a.l[2] | a.l[1]) != 0 || a.l[0] >= 1;
  return gt8 | (eq8 & ((a.l[7] != 0) | gt6 | (eq6 & low)));
}
// Canonical form [0, p) of an N-form or lazy value (|limb 8| < 2^29).  One quotient estimate from the top
// limb brings the value into (-p, 2p) (fe_weak_reduce), then ONE conditional + p and ONE conditional - p,
// both as masked limb updates (p has three non-zero limbs): three carry chains and no branch, where the
// round-1 version looped over "add p while negative, subtract p while >= p" with five.
SP_HD fe fe_canon(const fe& a_in) {
  fe a;
  {
    const int32_t q = a_in.l[8] >> 19;
    a = a_in;
    a.l[0] -= q;
    a.l[6] -= q * P6;
    a.l[8] -= q * P8;
    a = fe_carry(a);  // value in (-p, 2p), limbs 0..7 normalised
  }
  const int32_t neg = a.l[8] >> 31;  // all ones when negative
  a.l[0] += neg & 1;
  a.l[6] += neg & P6;
  a.l[8] += neg & P8;
  a = fe_carry(a);  // [0, 2p)
  const int32_t ge = fe_geq_p_canon_limbs(a) ? -1 : 0;
  a.l[0] -= ge & 1;
  a.l[6] -= ge & P6;
  a.l[8] -= ge & P8;
  return fe_carry(a);
}

// Montgomery <-> plain.  to_mont input: canonical limbs of x (< p); output N-form of x*R.
SP_HD fe fe_to_mont(const fe& a) { return fe_mul(a, FE_R2); }
// output canonical limbs of a/R mod p
SP_HD fe fe_from_mont(const fe& a) {
  cols t;
  cols_zero(t);
#pragma unroll
  for (int i = 0; i < NL; ++i) t.c[i] = a.l[i];
  return fe_canon(fe_reduce(t));
}
SP_HD bool fe_is_zero(const fe& a) {  // a in Montgomery N-form (|value| < 16p)
  fe c = fe_from_mont(a);
  int32_t acc = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) acc |= c.l[i];
  return acc == 0;
}
SP_HD bool fe_eq(const fe& a, const fe& b) { return fe_is_zero(fe_carry(fe_sub(a, b))); }

// ---- exponentiations ----
SP_HD fe fe_sqr_n(fe a, int n) {
  for (int i = 0; i < n; ++i) a = fe_sqr(a);
  return a;
}
// a^(2^59 + 16)
SP_HD fe fe_pow_2p59_plus16(const fe& a) {
  fe t = fe_sqr_n(a, 55);   // a^(2^55)
  t = fe_mul(t, a);         // a^(2^55 + 1)
  return fe_sqr_n(t, 4);    // a^(2^59 + 16)
}
// a^(p-2):  p - 2 = (2^59 + 16) * 2^192 + (2^192 - 1)
SP_HD fe fe_inv_fermat(const fe& a) {
  // a^63
  fe a2 = fe_sqr(a);
  fe a3 = fe_mul(a2, a);
  fe a7 = fe_mul(fe_sqr(a3), a);
  fe a63 = fe_mul(fe_sqr_n(a7, 3), a7);
  fe r = fe_pow_2p59_plus16(a);
  for (int i = 0; i < 32; ++i) {  // 32 windows of six one-bits
    r = fe_sqr_n(r, 6);
    r = fe_mul(r, a63);
  }
  return r;
}
// ---- inversion by divsteps ("safegcd", Bernstein-Yang 2019) ----------------------------------------
// Same structure as the constant-time modular inverse used for secp256k1 with 30-bit limbs, here
// with 29-bit limbs: 21 rounds of 29 branch-free divsteps on the low limb build a 2x2 transition
// matrix that is then applied to (f, g) exactly and to (d, e) modulo p.  21 * 29 = 609 >= 590
// divsteps suffice for 256-bit operands (half-delta variant).  ~19k VALU instructions instead of
// the ~40k of the Fermat chain above; identical for every lane, so no divergence.
struct trans2x2 {
  int32_t u, v, q, r;
};

SP_HD int32_t divsteps_29(int32_t zeta, uint32_t f0, uint32_t g0, trans2x2& t) {
  uint32_t u = 1, v = 0, q = 0, r = 1;
  uint32_t f = f0, g = g0;
  for (int i = 0; i < LB; ++i) {
    uint32_t c1 = (uint32_t)(zeta >> 31);
    const uint32_t c2 = 0u - (g & 1u);
    const uint32_t x = (f ^ c1) - c1;
    const uint32_t y = (u ^ c1) - c1;
    const uint32_t z = (v ^ c1) - c1;
    g += x & c2;
    q += y & c2;
    r += z & c2;
    c1 &= c2;
    zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
    f += g & c1;
    u += q & c1;
    v += r & c1;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  t.u = (int32_t)u;
  t.v = (int32_t)v;
  t.q = (int32_t)q;
  t.r = (int32_t)r;
  return zeta;
}

// Variable-time form of the same 29 divsteps (the "var" variant of the safegcd implementations): runs
// of zero bits of g are skipped with one count-trailing-zeros, and up to six low bits of g are
// cancelled per iteration by adding the right multiple of f (w = -g / f mod 2^k, with 1/f mod 64 from
// one Newton step on f itself, f * f = 1 mod 8).  Uses eta = -delta with delta starting at 1 (the
// classic divstep; the half-delta start of the fixed-length version buys nothing here because the
// callers stop as soon as g == 0).  Measured on random inputs (tools/sim notes in DESIGN.md): 7.5
// iterations of ~38 instructions per 29 divsteps instead of 29 x 22, 18.4 batches per inversion.
// Only for public data (hash outputs, signature verification).
SP_HD int32_t divsteps_29_var(int32_t eta, uint32_t f0, uint32_t g0, trans2x2& t) {
  uint32_t u = 1, v = 0, q = 0, r = 1;
  uint32_t f = f0, g = g0;
  int32_t i = LB;
  for (;;) {
    // sentinel at bit i: never count more zeros than divsteps left (1 <= i <= 29)
    const int32_t zeros = (int32_t)__builtin_ctz(g | (0xffffffffu << i));
    g >>= zeros;
    u <<= zeros;
    v <<= zeros;
    eta -= zeros;
    i -= zeros;
    if (i == 0) break;
    // g is odd here.  eta < 0: swap so that the divstep adds a multiple of the (new) f to g.
    const bool neg = eta < 0;
    const uint32_t nf = neg ? g : f, ng = neg ? 0u - f : g;
    const uint32_t nu = neg ? q : u, nq = neg ? 0u - u : q;
    const uint32_t nv = neg ? r : v, nr = neg ? 0u - v : r;
    f = nf; g = ng; u = nu; q = nq; v = nv; r = nr;
    eta = neg ? -eta : eta;
    // cancel min(eta + 1, i, 6) low bits of g: after that many divsteps the sign of eta would flip
    const int32_t lim = (eta + 1 < i) ? eta + 1 : i;
    const uint32_t m = (0xffffffffu >> (32 - lim)) & 63u;
    uint32_t x = f;            // f^-1 mod 8
    x *= 2u - f * x;           // f^-1 mod 64
    const uint32_t w = (g * (0u - x)) & m;
    g += f * w;
    q += u * w;
    r += v * w;
  }
  t.u = (int32_t)u;
  t.v = (int32_t)v;
  t.q = (int32_t)q;
  t.r = (int32_t)r;
  return eta;
}

// (f, g) <- t (f, g) / 2^29 (exact)
SP_HD void gcd_update_fg(fe& f, fe& g, const trans2x2& t) {
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  int64_t cf = u * f.l[0] + v * g.l[0];
  int64_t cg = q * f.l[0] + r * g.l[0];
  cf >>= LB;
  cg >>= LB;
#pragma unroll
  for (int i = 1; i < NL; ++i) {
    cf += u * f.l[i] + v * g.l[i];
    cg += q * f.l[i] + r * g.l[i];
    f.l[i - 1] = (int32_t)((uint32_t)cf & LMASK);
    g.l[i - 1] = (int32_t)((uint32_t)cg & LMASK);
    cf >>= LB;
    cg >>= LB;
  }
  f.l[NL - 1] = (int32_t)cf;
  g.l[NL - 1] = (int32_t)cg;
  fe_pin(f);
  fe_pin(g);
}

// (d, e) <- t (d, e) / 2^29 mod p, keeping d, e in (-2p, p).  p = 1 (mod 2^29).
SP_HD void gcd_update_de(fe& d, fe& e, const trans2x2& t) {
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  const int32_t sd = d.l[NL - 1] >> 31, se = e.l[NL - 1] >> 31;
  int32_t md = (t.u & sd) + (t.v & se);
  int32_t me = (t.q & sd) + (t.r & se);
  int64_t cd = u * d.l[0] + v * e.l[0];
  int64_t ce = q * d.l[0] + r * e.l[0];
  // choose md, me so that cd + p0 * md = 0 (mod 2^29) with p0 = 1
  md -= (int32_t)(((uint32_t)cd + (uint32_t)md) & LMASK);
  me -= (int32_t)(((uint32_t)ce + (uint32_t)me) & LMASK);
  cd += md;  // modulus limb 0 = 1
  ce += me;
  cd >>= LB;
  ce >>= LB;
#pragma unroll
  for (int i = 1; i < NL; ++i) {
    cd += u * d.l[i] + v * e.l[i];
    ce += q * d.l[i] + r * e.l[i];
    if (i == 6) {
      cd += (int64_t)P6 * md;
      ce += (int64_t)P6 * me;
    }
    if (i == 8) {
      cd += (int64_t)P8 * md;
      ce += (int64_t)P8 * me;
    }
    d.l[i - 1] = (int32_t)((uint32_t)cd & LMASK);
    e.l[i - 1] = (int32_t)((uint32_t)ce & LMASK);
    cd >>= LB;
    ce >>= LB;
  }
  d.l[NL - 1] = (int32_t)cd;
  e.l[NL - 1] = (int32_t)ce;
  fe_pin(d);
  fe_pin(e);
}

// Plain integer inverse: x canonical limbs in [0, p)  ->  canonical limbs of x^-1 mod p (0 -> 0).
SP_HD fe fe_inv_plain_gcd(const fe& x) {
  fe d = FE_ZERO, e = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  fe f = FE_P, g = x;
  int32_t zeta = -1;
  for (int it = 0; it < 21; ++it) {
    trans2x2 t;
    zeta = divsteps_29(zeta, (uint32_t)f.l[0], (uint32_t)g.l[0], t);
    gcd_update_de(d, e, t);
    gcd_update_fg(f, g, t);
#if defined(__HIP_DEVICE_COMPILE__)
    // Once g == 0 the remaining divsteps only halve and re-double (d, e) / (f, g) in lockstep and
    // leave d * sign(f) unchanged, so leaving early is exact.  Random inputs need 491..523 divsteps
    // (mean 507): a wave is done after 18 rounds, rarely 19, instead of the worst-case 21.  The
    // test is wave-uniform, so there is no divergence.  Field inversions only see public data
    // (hash outputs, signature verification); the mod-N inversion used by signing stays fixed-length.
    if (it >= 16) {
      int32_t nz = 0;
#pragma unroll
      for (int i = 0; i < NL; ++i) nz |= g.l[i];
      if (__all(nz == 0)) break;
    }
#endif
  }
  // f = +-1; result = d * sign(f), brought to [0, p)
  const int32_t sf = f.l[NL - 1] >> 31;  // -1 if f negative
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = (d.l[i] ^ sf) - sf;
  r = fe_carry(r);
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    if (r.l[8] < 0) r = fe_carry(fe_add(r, FE_P));
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    if (fe_geq_p_canon_limbs(r)) r = fe_carry(fe_sub(r, FE_P));
  }
  return r;
}

// Variable-time plain inverse (see divsteps_29_var): same contract as fe_inv_plain_gcd.
SP_HD fe fe_inv_plain_gcd_var(const fe& x) {
  fe d = FE_ZERO, e = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  fe f = FE_P, g = x;
  int32_t eta = -1;
  // 26 * 29 = 754 >= 741 divsteps bound the classic (delta = 1) iteration for 256-bit inputs; random
  // inputs are done after 18-19 batches.
  for (int it = 0; it < 26; ++it) {
    trans2x2 t;
    eta = divsteps_29_var(eta, (uint32_t)f.l[0], (uint32_t)g.l[0], t);
    gcd_update_de(d, e, t);
    gcd_update_fg(f, g, t);
    int32_t nz = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) nz |= g.l[i];
#if defined(__HIP_DEVICE_COMPILE__)
    if (__all(nz == 0)) break;
#else
    if (nz == 0) break;
#endif
  }
  const int32_t sf = f.l[NL - 1] >> 31;  // -1 if f negative
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = (d.l[i] ^ sf) - sf;
  r = fe_carry(r);
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    if (r.l[8] < 0) r = fe_carry(fe_add(r, FE_P));
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    if (fe_geq_p_canon_limbs(r)) r = fe_carry(fe_sub(r, FE_P));
  }
  return r;
}

// ---- inversion steered by doubles (Lehmer's gcd, nearest-integer quotients) -------------------------
// The divsteps batches above work from the LOW limb: 29 divsteps cost ~7.5 data-dependent iterations of
// ~36 instructions and take ~14 bits off f g, so an inversion is 18.4 batches (matrix applications) and
// ~140 iterations.  Working from the TOP with the FP64 unit is shorter on this chip (v_fma_f64 issues at the
// rate of an integer multiply-add): per batch A and B become doubles by Horner (a 53-bit relative
// approximation, no search for the leading limb), Euclid with nearest-integer quotients runs on the doubles
// while the divisor stays above 2^-27 of the batch's larger input - ~10 steps of ~15 instructions, every
// cofactor an exact integer below 2^29 - and the integer 2x2 matrix is applied to (A, B) exactly and to
// their Bezout cofactors (D, E).  10 batches and ~100 Euclid steps per inversion
// (tools/sim/lehmer_inverse_model.py is the executable model: counts, bounds, edge inputs).
//   * Any quotient sequence gives a unimodular matrix, so the doubles only steer: a quotient that is off
//     by one (rounding of a * rcp(b)) costs progress, never correctness.  Every remainder a - q b is exact
//     in the fma (|a - q b| <= |b| / 2 is a multiple of ulp(b)), so the doubles follow Euclid on the rounded
//     pair exactly; applied to the true integers the last remainder is off by < 2^-22 of the input.
//   * D x = A, E x = B (mod p) start as (0, 1) and need NO reduction: |E| <= 2p / |A| for remainders that
//     at least halve (A E - B D = +-p is invariant), and the limbs hold 2^260.
//   * A batch that starts with min(A, B) < 2^-27 max(A, B) would need a partial quotient no int32 matrix
//     holds (x = 1, 2, ...; probability ~2^-27 per batch on random input): lehmer_batch reports it and the
//     caller redoes the value with the divsteps inversion.
// Variable time: public data only, like divsteps_29_var.
struct lehmer_rows {
  double ua, va, ub, vb;  // new A = ua A + va B, new B = ub A + vb B
};
SP_HD uint64_t lehmer_bits(double v) {
  uint64_t u;
  __builtin_memcpy(&u, &v, 8);
  return u;
}
SP_HD double lehmer_from_bits(uint64_t u) {
  double v;
  __builtin_memcpy(&v, &u, 8);
  return v;
}
// v with its sign flipped when s is negative (one 32-bit xor on the device)
SP_HD double lehmer_flip(double v, double s) {
  return lehmer_from_bits(lehmer_bits(v) ^ (lehmer_bits(s) & 0x8000000000000000ull));
}
// signed-limb integer (limbs 0..7 in [0, 2^29), limb 8 signed) -> nearest-ish double (7 roundings)
SP_HD double lehmer_to_double(const fe& a) {
  double s = (double)a.l[NL - 1];
#pragma unroll
  for (int i = NL - 2; i >= 0; --i) s = __builtin_fma(s, 536870912.0, (double)a.l[i]);
  return s;
}
// a <- a - q b with q = nearest integer to a / b (0 when `live` is false: a no-op); the row (ua, va) follows.
// Remainders are SIGNED (|a - q b| <= |b| / 2): no absolute values and no sign flips in the loop, the matrix is
// unimodular all the same.  b != 0 when live.  The quotient comes from the bare v_rcp_f64: its error only
// matters at half-integers or for quotients above 2^20, and there it costs progress (a remainder above b / 2,
// picked up by the next step), never correctness.
SP_HD void lehmer_step(double& a, double& ua, double& va, const double b, const double ub, const double vb,
                       const bool live) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double rc = __builtin_amdgcn_rcp(b);
#elif defined(SP_LEHMER_RCP_ERROR)
  // host tests: a reciprocal that is deliberately worse than anything v_rcp_f64 delivers (relative error
  // sp_lehmer_rcp_error, sign alternating) - the quotients go wrong more often, the result must not
  extern double sp_lehmer_rcp_error;
  static thread_local int flip__ = 0;
  const double rc = (1.0 / b) * (1.0 + ((flip__++ & 1) ? sp_lehmer_rcp_error : -sp_lehmer_rcp_error));
#else
  const double rc = 1.0 / b;
#endif
  const double q = live ? __builtin_rint(a * rc) : 0.0;
  a = __builtin_fma(-q, b, a);
  ua = __builtin_fma(-q, ub, ua);
  va = __builtin_fma(-q, vb, va);
}
// a, b >= 0.  Returns false when the batch cannot be represented (see above); b == 0 gives the identity.
SP_HD bool lehmer_batch(double a, double b, lehmer_rows& m) {
  double ua = 1.0, va = 0.0, ub = 0.0, vb = 1.0;
  const double lim = __builtin_fmax(a, b) * 0x1p-27;
  const bool done = b == 0.0;
  const bool ok = done | !(__builtin_fmin(a, b) < lim);
  const double thresh = (ok & !done) ? __builtin_fmax(lim, 0.5) : __builtin_inf();
  // Two steps per trip so that a and b keep their registers; the second one is switched off (q = 0) when
  // the first already went below the threshold.  Straight-line body: one exec-mask update per trip.
  while (__builtin_fmin(__builtin_fabs(a), __builtin_fabs(b)) >= thresh) {
    lehmer_step(a, ua, va, b, ub, vb, true);
    lehmer_step(b, ub, vb, a, ua, va, __builtin_fabs(a) >= thresh);
  }
  // the last (smaller) remainder goes to the B row: "B == 0" is the end test
  const bool odd = __builtin_fabs(a) < __builtin_fabs(b);
  m.ua = odd ? ub : ua;
  m.va = odd ? vb : va;
  m.ub = odd ? ua : ub;
  m.vb = odd ? va : vb;
  return ok;
}
// cx x + cy y as signed-limb integer (no shift, no reduction); |cx|, |cy| < 2^29
SP_HD fe lehmer_row(const fe& x, const fe& y, int32_t cx, int32_t cy) {
  const int64_t a = cx, b = cy;
  int64_t c = 0;
  fe r;
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) {
    c += a * x.l[i] + b * y.l[i];
    r.l[i] = (int32_t)((uint32_t)c & LMASK);
    c >>= LB;
  }
  c += a * x.l[NL - 1] + b * y.l[NL - 1];
  SP_CHK32(c);
  r.l[NL - 1] = (int32_t)c;
  fe_pin(r);
  return r;
}
// d * sign in (-3p, 3p) -> canonical
SP_HD fe lehmer_finish(const fe& d, int32_t sf) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = (d.l[i] ^ sf) - sf;
  r = fe_carry(r);
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    if (r.l[8] < 0) r = fe_carry(fe_add(r, FE_P));
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    if (fe_geq_p_canon_limbs(r)) r = fe_carry(fe_sub(r, FE_P));
  }
  return r;
}
// Bezout cofactor of x for gcd(modulus, x), one value per lane.  x: any signed-limb integer with limbs 0..7 in
// [0, 2^29) and |x| < 16 modulus (reduced or not: A E - B D = +-modulus holds from the start, so |D| < 2 modulus
// whatever the size of x).  On return (true) D sign = x^-1 (mod modulus), sign = sf ? -1 : +1, and D = 0 when
// x is a multiple of the modulus.  false: some value of the wave needs the divsteps form.
constexpr int LEHMER_MAX_BATCHES = 24;  // ~10 on average, ~13 at most on random 252-bit input; 22 by the worst-case bound
#if defined(SP_LEHMER_RCP_ERROR)
extern int sp_lehmer_batches;      // host tests: batches the last lehmer_bezout calls ran (reset by the test)
extern int sp_lehmer_max_batches;  // host tests: a smaller budget than LEHMER_MAX_BATCHES, to run out of it on purpose
#define SP_LEHMER_BUDGET sp_lehmer_max_batches
#else
#define SP_LEHMER_BUDGET LEHMER_MAX_BATCHES
#endif
SP_HD bool lehmer_bezout(const fe& modulus, const fe& x, fe& D, int32_t& sf) {
  fe A = modulus, B = x, E = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  D = FE_ZERO;
  double ad = 0.0;
  bool ok = true;
  for (int it = 0; it < SP_LEHMER_BUDGET; ++it) {
    ad = lehmer_to_double(A);
    const double bd = lehmer_to_double(B);
    lehmer_rows m;
    ok &= lehmer_batch(__builtin_fabs(ad), __builtin_fabs(bd), m);
#if defined(__HIP_DEVICE_COMPILE__)
    if (__all(bd == 0.0 || !ok)) break;
#else
    if (bd == 0.0 || !ok) break;
#endif
    // the batch ran on (|A|, |B|) = (sa A, sb B): fold the signs into the columns
    const int32_t ua = (int32_t)lehmer_flip(m.ua, ad), va = (int32_t)lehmer_flip(m.va, bd);
    const int32_t ub = (int32_t)lehmer_flip(m.ub, ad), vb = (int32_t)lehmer_flip(m.vb, bd);
    const fe A2 = lehmer_row(A, B, ua, va), B2 = lehmer_row(A, B, ub, vb);
    const fe D2 = lehmer_row(D, E, ua, va), E2 = lehmer_row(D, E, ub, vb);
    A = A2; B = B2; D = D2; E = E2;
#if defined(SP_LEHMER_RCP_ERROR)
    ++sp_lehmer_batches;
#endif
  }
  // Out of batches with a remainder left (worst case ~22 of the 24 for 252-bit inputs, and the
  // device's bare v_rcp_f64 loses a little progress per wrong quotient): NOT converged - the caller redoes the
  // value with the divsteps form instead of reading a stale A.  A is re-read: the loop leaves `ad` one batch old.
  ok &= lehmer_to_double(B) == 0.0;
  ad = lehmer_to_double(A);
  sf = ad < 0.0 ? -1 : 0;  // A = +-1 - or +-modulus for a multiple of the modulus (reduced or not): answer 0
  if (__builtin_fabs(ad) != 1.0) D = FE_ZERO;
#if defined(__HIP_DEVICE_COMPILE__)
  return !__any(!ok);
#else
  return ok;
#endif
}
// Plain integer inverse, one value per lane: x canonical in [0, p) -> canonical x^-1 mod p (0 -> 0).
SP_HD fe fe_inv_plain_lehmer(const fe& x) {
  fe D;
  int32_t sf;
  if (!lehmer_bezout(FE_P, x, D, sf)) return fe_inv_plain_gcd_var(x);
  return lehmer_finish(D, sf);
}

// R^3 mod p: turns the plain inverse of a Montgomery value (a R)^-1 = a^-1 R^-1 into a^-1 R.
constexpr fe FE_R3 = {{0x18c6c71b, 0x1f4501b7, 0xd98e2e, 0x677ffcc, 0x3aa2b83, 0xd8c0006, 0xc2709f0,
                       0x13c0a666, 0x7bcc3}};

// Montgomery-form inverse via divsteps.  a: N-form Montgomery value, |value| < 16p.
SP_HD fe fe_inv_gcd(const fe& a) {
  const fe canon = fe_canon(fe_mul(a, FE_ONE_M));
  return fe_mul(fe_inv_plain_gcd(canon), FE_R3);
}

SP_HD fe fe_inv_gcd_var(const fe& a) {
  const fe canon = fe_canon(fe_mul(a, FE_ONE_M));
  return fe_mul(fe_inv_plain_gcd_var(canon), FE_R3);
}
// Round 3: the double-steered form on the representative as it comes (no reduction multiplication, no canonical
// form in front of the gcd), the divsteps form as its fallback.  12 us instead of 33 us on a lone wave of 64
// distinct values (tools/ubench/inv_quad.hip).
SP_HD fe fe_inv_lehmer(const fe& a) {
  fe D;
  int32_t sf;
  if (!lehmer_bezout(FE_P, fe_carry(a), D, sf)) return fe_inv_gcd_var(a);
  return fe_mul(lehmer_finish(D, sf), FE_R3);
}

// The inversion used everywhere (public data only: hash outputs, signature verification; the
// mod-N inversion used by signing, fn_inv, stays fixed-length).
#ifndef SP_INV_CONST_TIME
SP_HD fe fe_inv(const fe& a) { return fe_inv_lehmer(a); }
#else
SP_HD fe fe_inv(const fe& a) { return fe_inv_gcd(a); }
#endif

// Legendre symbol test: a^((p-1)/2) == 1, (p-1)/2 = (2^59 + 17) * 2^191.  a must be non-zero.
SP_HD bool fe_is_qr(const fe& a) {
  fe t = fe_mul(fe_pow_2p59_plus16(a), a);  // a^(2^59 + 17)
  t = fe_sqr_n(t, 191);
  return fe_eq(t, FE_ONE_M);
}

// =============================================================================================
// Arithmetic modulo the curve order N (generic Montgomery, same limb shape, R = 2^261).
// Used only for the handful of scalar operations per signature (w = 1/s, u1 = z w, u2 = r w, ...).
// =============================================================================================
constexpr int32_t N_LIMB[NL] = {0xdc64d2f, 0x1335120d, 0x19ec8c87, 0x224db95, 0x1ffffb78,
                                0x1fffffff, 0x43ffff, 0x0, 0x80000};
constexpr uint32_t N0INV = 0x8bde631;  // -N^-1 mod 2^29
constexpr fe FN_ONE_M = {{0x1491912f, 0x1eecdc54, 0x7ba6e20, 0xeb68458, 0x121b33, 0x0, 0x10440000,
                          0x1ffffff7, 0x7ffff}};
constexpr fe FN_R2 = {{0xbeb7fac, 0x25b7097, 0x15023c53, 0x9db76e9, 0xee5ec5a, 0x19ef1775,
                       0xff0ab4, 0x1199bb2f, 0x795f0}};
constexpr fe FN_N = {{0xdc64d2f, 0x1335120d, 0x19ec8c87, 0x224db95, 0x1ffffb78, 0x1fffffff,
                      0x43ffff, 0x0, 0x80000}};

SP_HD fe fn_reduce(cols& t) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const uint32_t q = ((uint32_t)t.c[i] * N0INV) & LMASK;
#pragma unroll
    for (int j = 0; j < NL; ++j) t.c[i + j] += (int64_t)q * (int64_t)N_LIMB[j];
    t.c[i + 1] += t.c[i] >> LB;
  }
  fe r;
  int64_t carry = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t v = t.c[9 + k] + carry;
    r.l[k] = (int32_t)((uint32_t)v & LMASK);
    carry = v >> LB;
  }
  r.l[8] = (int32_t)carry;
  fe_pin(r);
  return r;
}
SP_HD fe fn_mul(const fe& a, const fe& b) {
  cols t;
  cols_zero(t);
  cols_mac(t, a, b);
  return fn_reduce(t);
}
SP_HD fe fn_sqr(const fe& a) {
  cols t;
  cols_zero(t);
  cols_sqr(t, a);
  return fn_reduce(t);
}
// lexicographic a >= m on canonical non-negative limbs
SP_HD bool limbs_geq(const fe& a, const fe& m) {
#pragma unroll
  for (int i = NL - 1; i >= 0; --i) {
    if (a.l[i] != m.l[i]) return a.l[i] > m.l[i];
  }
  return true;
}
SP_HD fe fn_canon(const fe& a_in) {
  fe a = fe_carry(a_in);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    if (a.l[8] < 0) a = fe_carry(fe_add(a, FN_N));
  }
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    if (limbs_geq(a, FN_N)) a = fe_carry(fe_sub(a, FN_N));
  }
  return a;
}
SP_HD fe fn_to_mont(const fe& a) { return fn_mul(a, FN_R2); }
SP_HD fe fn_from_mont(const fe& a) {
  cols t;
  cols_zero(t);
#pragma unroll
  for (int i = 0; i < NL; ++i) t.c[i] = a.l[i];
  return fn_canon(fn_reduce(t));
}
SP_HD bool limbs_is_zero(const fe& a) {
  int32_t acc = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) acc |= a.l[i];
  return acc == 0;
}
// divsteps inversion modulo N (generic modulus limbs; N^-1 mod 2^29 = 0x174219cf)
constexpr uint32_t N_INV29 = 0x174219cf;
constexpr fe FN_R3 = {{0x7a1941b, 0x6d28493, 0x14423a93, 0xf5f04e8, 0x16829a1, 0x1fb7f694, 0x6af3bdb,
                       0x1fe1868, 0x54465}};
SP_HD void gcd_update_de_n(fe& d, fe& e, const trans2x2& t) {
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  const int32_t sd = d.l[NL - 1] >> 31, se = e.l[NL - 1] >> 31;
  int32_t md = (t.u & sd) + (t.v & se);
  int32_t me = (t.q & sd) + (t.r & se);
  int64_t cd = u * d.l[0] + v * e.l[0];
  int64_t ce = q * d.l[0] + r * e.l[0];
  md -= (int32_t)((N_INV29 * (uint32_t)cd + (uint32_t)md) & LMASK);
  me -= (int32_t)((N_INV29 * (uint32_t)ce + (uint32_t)me) & LMASK);
  cd += (int64_t)N_LIMB[0] * md;
  ce += (int64_t)N_LIMB[0] * me;
  cd >>= LB;
  ce >>= LB;
#pragma unroll
  for (int i = 1; i < NL; ++i) {
    cd += u * d.l[i] + v * e.l[i] + (int64_t)N_LIMB[i] * md;
    ce += q * d.l[i] + r * e.l[i] + (int64_t)N_LIMB[i] * me;
    d.l[i - 1] = (int32_t)((uint32_t)cd & LMASK);
    e.l[i - 1] = (int32_t)((uint32_t)ce & LMASK);
    cd >>= LB;
    ce >>= LB;
  }
  d.l[NL - 1] = (int32_t)cd;
  e.l[NL - 1] = (int32_t)ce;
  fe_pin(d);
  fe_pin(e);
}
// Montgomery-form inverse modulo N via divsteps (a: Montgomery N-form, value in (-p, 2p)).
SP_HD fe fn_inv(const fe& a) {
  const fe x = fn_canon(fn_mul(a, FN_ONE_M));
  fe d = FE_ZERO, e = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  fe f = FN_N, g = x;
  int32_t zeta = -1;
  for (int it = 0; it < 21; ++it) {
    trans2x2 t;
    zeta = divsteps_29(zeta, (uint32_t)f.l[0], (uint32_t)g.l[0], t);
    gcd_update_de_n(d, e, t);
    gcd_update_fg(f, g, t);
  }
  const int32_t sf = f.l[NL - 1] >> 31;
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = (d.l[i] ^ sf) - sf;
  r = fe_carry(r);
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    if (r.l[8] < 0) r = fe_carry(fe_add(r, FN_N));
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    if (limbs_geq(r, FN_N)) r = fe_carry(fe_sub(r, FN_N));
  }
  return fn_mul(r, FN_R3);
}

// d * sign in (-3N, 3N) -> Montgomery form of the canonical value
SP_HD fe fn_inv_finish(const fe& d, int32_t sf) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = (d.l[i] ^ sf) - sf;
  r = fe_carry(r);
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    if (r.l[8] < 0) r = fe_carry(fe_add(r, FN_N));
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    if (limbs_geq(r, FN_N)) r = fe_carry(fe_sub(r, FN_N));
  }
  return fn_mul(r, FN_R3);
}

// Variable-time twin of fn_inv for PUBLIC scalars (w = s^-1 of a signature being verified): the same
// divsteps_29_var batches as fe_inv_plain_gcd_var, transition matrices applied modulo N.
SP_HD fe fn_inv_plain_divsteps_var(const fe& x) {
  fe d = FE_ZERO, e = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  fe f = FN_N, g = x;
  int32_t eta = -1;
  for (int it = 0; it < 26; ++it) {
    trans2x2 t;
    eta = divsteps_29_var(eta, (uint32_t)f.l[0], (uint32_t)g.l[0], t);
    gcd_update_de_n(d, e, t);
    gcd_update_fg(f, g, t);
    int32_t nz = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) nz |= g.l[i];
#if defined(__HIP_DEVICE_COMPILE__)
    if (__all(nz == 0)) break;
#else
    if (nz == 0) break;
#endif
  }
  return fn_inv_finish(d, f.l[NL - 1] >> 31);
}
// Round 3: the double-steered inversion (lehmer_bezout: the modulus only enters as the start value of A, the
// cofactors need no reduction), the divsteps form above as its fallback.
SP_HD fe fn_inv_var(const fe& a) {
  const fe x = fn_canon(fn_mul(a, FN_ONE_M));
  fe D;
  int32_t sf;
  if (!lehmer_bezout(FN_N, x, D, sf)) return fn_inv_plain_divsteps_var(x);
  return fn_inv_finish(D, sf);
}

// a^(N-2) by square-and-multiply over the bits of N-2 (kept as a cross-check of fn_inv).
SP_HD fe fn_inv_fermat(const fe& a) {
  // N - 2 limbs: N_LIMB with limb0 - 2 (0xdc64d2f - 2 = 0xdc64d2d, no borrow)
  fe r = FN_ONE_M;
  for (int i = NL - 1; i >= 0; --i) {
    const uint32_t e = (uint32_t)N_LIMB[i] - (i == 0 ? 2u : 0u);
    const int top = (i == NL - 1) ? 19 : LB - 1;  // N < 2^252: limb 8 has 20 bits
    for (int b = top; b >= 0; --b) {
      r = fn_sqr(r);
      if ((e >> b) & 1u) r = fn_mul(r, a);
    }
  }
  return r;
}

}  // namespace sp

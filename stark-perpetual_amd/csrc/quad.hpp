// Quad-parallel point arithmetic: the latency path of the small tree levels.
//
// A SIMD issues about one VALU instruction per 5 cycles whether one wave or eight live on it
// (tools/ubench/lat_parts.hip), so a level that cannot fill the chip costs the LENGTH OF THE
// DEPENDENT INSTRUCTION CHAIN of one wave, and lanes are free.  A point addition in XYZZ
// coordinates is 14 field multiplications of which at most four depend on each other, so the four
// lanes of a DPP quad execute one addition together: every lane does ONE fe_mul per round (the same
// instruction stream, lane-private operands), operands move between the lanes with `v_mov_b32 ...
// quad_perm` (full-rate VALU, no LDS), and an addition is 4 rounds (~1.0 k instructions) instead of
// 14 multiplications (~2.4 k).
//
// Layout of a point on a quad ("qpt"): lane k = 2 h + e of the quad holds
//     a = e ? Y : X,   b = e ? ZZZ : ZZ        of point number h,
// i.e. lanes 0,1 carry point P1 and lanes 2,3 carry point P2 of the addition the quad is about to do.
// qadd() returns P1 + P2 replicated on both lane pairs (ready to be P1 of the next addition, or to be
// sent to the partner quad as its P2).
//
// Reference semantics are unchanged: these are the same group elements the affine chain of
// signature.py:300-318 visits; only x = X / ZZ of the total leaves the kernel.
#pragma once
#include "curve.hpp"

namespace sp {

constexpr int quad_perm(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }

template <int CTRL>
__device__ __forceinline__ fe fe_dpp(const fe& v) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    r.l[i] = __builtin_amdgcn_mov_dpp(v.l[i], CTRL, 0xF, 0xF, true);
    // Keep the move a move: hipcc (ROCm 7.2) folds two DPP moves of the SAME register with different
    // controls into one v_sub*_dpp and then mis-selects (tools/ubench/quad_check.hip: Y3 = T[1] - T[0]
    // came out 0 on the odd lanes once a v_cndmask followed).  The empty asm stops the DPP combiner.
    asm volatile("" : "+v"(r.l[i]));
  }
  return r;
}
// lane <- lane ^ XOR inside each half wave (XOR = 4, 8, 16): ds_swizzle bit mode, no LDS memory
template <int XOR>
__device__ __forceinline__ fe fe_swizzle_xor(const fe& v) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = __builtin_amdgcn_ds_swizzle(v.l[i], (XOR << 10) | 0x1F);
  return r;
}
__device__ __forceinline__ fe fe_sel(bool c, const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

struct qpt {
  fe a, b;
};

// P1 + P2 ("add-2008-s", 12M + 2S as four rounds of one multiplication per lane).  k = lane & 3.
// Exceptional inputs drive ZZ to 0 exactly like xyzz_add (curve.hpp).  With X_ONLY only X3 (in .a of
// every lane) and ZZ3 (in .b of every lane) are produced - three rounds.
template <bool X_ONLY>
__device__ __forceinline__ qpt qadd(const qpt& p, int k) {
  const bool h = (k & 2) != 0, e = (k & 1) != 0;
  // round 1: U1 = X1 ZZ2 | S1 = Y1 ZZZ2 | U2 = X2 ZZ1 | S2 = Y2 ZZZ1
  const fe T1 = fe_mul(p.a, fe_dpp<quad_perm(2, 3, 0, 1)>(p.b));
  // P = U2 - U1 | R = S2 - S1 | -P | -R      (B = 1, signed)
  const fe D = fe_sub(fe_dpp<quad_perm(2, 3, 0, 1)>(T1), T1);
  // round 2: PP = P^2 | RR = R^2 | ZZ1 ZZ2 | ZZZ1 ZZZ2
  const fe T2 = fe_mul(fe_sel(h, p.b, D), fe_sel(h, fe_dpp<quad_perm(0, 1, 0, 1)>(p.b), D));
  // round 3: PPP = P PP | Q = U1 PP | ZZ3 = ZZ1 ZZ2 PP | W = ZZZ1 ZZZ2 P
  const fe PPb = fe_dpp<quad_perm(0, 0, 0, 0)>(T2);
  const fe l3 = fe_sel(k == 0, D, fe_sel(k == 1, fe_dpp<quad_perm(0, 0, 0, 0)>(T1), T2));
  const fe r3 = fe_sel(k == 3, fe_dpp<quad_perm(0, 0, 0, 0)>(D), PPb);
  const fe T3 = fe_mul(l3, r3);
  // X3 = RR - PPP - 2 Q on every lane: limbs in (-3 * 2^29, 2^29) -> carry to N
  const fe Qb = fe_dpp<quad_perm(1, 1, 1, 1)>(T3);
  const fe X3 =
      fe_carry(fe_sub(fe_sub(fe_dpp<quad_perm(1, 1, 1, 1)>(T2), fe_dpp<quad_perm(0, 0, 0, 0)>(T3)), fe_dbl(Qb)));
  qpt r;
  if (X_ONLY) {
    r.a = X3;
    r.b = fe_dpp<quad_perm(2, 2, 2, 2)>(T3);
    return r;
  }
  // round 4: S1 PPP | R (Q - X3) | (unused) | ZZZ3 = W PP
  const fe l4 = fe_sel(k == 0, fe_dpp<quad_perm(1, 1, 1, 1)>(T1), fe_sel(k == 1, D, T3));
  const fe r4 = fe_sel(k == 1, fe_sub(T3, X3), fe_sel(k == 3, PPb, T3));
  const fe T4 = fe_mul(l4, r4);
  // Y3 = R (Q - X3) - S1 PPP: difference of two reduced products, B = 1 signed - a valid multiplicand
  const fe Y3 = fe_sub(fe_dpp<quad_perm(1, 1, 1, 1)>(T4), fe_dpp<quad_perm(0, 0, 0, 0)>(T4));
  r.a = fe_sel(e, Y3, X3);
  r.b = fe_sel(e, fe_dpp<quad_perm(3, 3, 3, 3)>(T4), fe_dpp<quad_perm(2, 2, 2, 2)>(T3));
  return r;
}

// Two sums of two affine points at once ("mmadd-2008-s", 4M + 2S as three rounds on two lanes each):
// lanes 0,1 add (x1, y1) + (x2, y2) of pair A, lanes 2,3 those of pair B (every lane passes its own
// pair).  Returns sum A as P1 and sum B as P2 of the qpt layout.
__device__ __forceinline__ qpt qmmadd(const fe& x1, const fe& y1, const fe& x2, const fe& y2, int k) {
  const bool e = (k & 1) != 0;
  // table entries may carry a negated y (signed windows): y2 - y1 can reach B = 2, whose square would
  // sit on the edge of the 64-bit column budget - carry it
  const fe P = fe_sub(x2, x1), R = fe_carry(fe_sub(y2, y1));
  const fe T1 = fe_sqr(fe_sel(e, R, P));                       // PP | RR
  const fe PPb = fe_dpp<quad_perm(0, 0, 2, 2)>(T1);
  const fe T2 = fe_mul(fe_sel(e, x1, P), PPb);                 // PPP | Q
  const fe PPPb = fe_dpp<quad_perm(0, 0, 2, 2)>(T2), Qb = fe_dpp<quad_perm(1, 1, 3, 3)>(T2);
  const fe X3 = fe_carry(fe_sub(fe_sub(fe_dpp<quad_perm(1, 1, 3, 3)>(T1), PPPb), fe_dbl(Qb)));
  const fe T3 = fe_mul(fe_sel(e, R, y1), fe_sel(e, fe_sub(Qb, X3), PPPb));  // y1 PPP | R (Q - X3)
  const fe Y3 = fe_sub(fe_dpp<quad_perm(1, 1, 3, 3)>(T3), fe_dpp<quad_perm(0, 0, 2, 2)>(T3));
  qpt r;
  r.a = fe_sel(e, Y3, X3);
  r.b = fe_sel(e, PPPb, PPb);
  return r;
}

// ---- inversion on a quad ---------------------------------------------------------------------------
// The variable-time divsteps inversion (fp29.hpp fe_inv_plain_gcd_var) spends a third of its time applying
// the 2x2 transition matrix of every 29-divstep batch to the four 9-limb vectors f, g (exactly) and
// d, e (modulo p): 72 multiply-adds and four carry chains per batch.  When the four lanes of a quad
// invert the SAME value (every lane of a quad-kernel group holds the same sum) each lane keeps ONE of
// the four vectors - lane 0: f, 1: g, 2: d, 3: e - and computes only its own row: 18 multiply-adds and
// one carry chain.  The batch of divsteps itself is computed by all four lanes from the broadcast low
// limbs of f and g.
//
// own <- (c_own * own + c_par * partner [+ m p]) / 2^29 with (c_own, c_par) = (u, v) on the f and d lanes,
// (r, q) on the g and e lanes; m makes the d / e rows divisible (p = 1 mod 2^29), as in gcd_update_de.
__device__ __forceinline__ void quad_gcd_update(fe& own, const trans2x2& t, int k) {
  const bool odd = (k & 1) != 0, modular = (k & 2) != 0;
  const fe par = fe_dpp<quad_perm(1, 0, 3, 2)>(own);
  const int32_t co = odd ? t.r : t.u, cp = odd ? t.q : t.v;
  const int64_t a = co, b = cp;
  const int32_t so = own.l[NL - 1] >> 31, sp_ = par.l[NL - 1] >> 31;
  int32_t m = (co & so) + (cp & sp_);
  int64_t c = a * own.l[0] + b * par.l[0];
  m -= (int32_t)(((uint32_t)c + (uint32_t)m) & LMASK);
  m = modular ? m : 0;  // the f, g rows divide exactly
  c += m;
  c >>= LB;
  fe r;
#pragma unroll
  for (int i = 1; i < NL; ++i) {
    c += a * own.l[i] + b * par.l[i];
    if (i == 6) c += (int64_t)P6 * m;
    if (i == 8) c += (int64_t)P8 * m;
    r.l[i - 1] = (int32_t)((uint32_t)c & LMASK);
    c >>= LB;
  }
  r.l[NL - 1] = (int32_t)c;
  fe_pin(r);
  own = r;
}

// x canonical in [0, p), identical on the four lanes of every quad -> canonical x^-1 mod p on every lane.
// (Round 3: the fallback of fe_inv_plain_quad below; until then the inversion of every small level.)
__device__ __forceinline__ fe fe_inv_plain_quad_divsteps(const fe& x, int k) {
  const fe one = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  fe own = k == 0 ? FE_P : (k == 1 ? x : (k == 2 ? FE_ZERO : one));  // f | g | d | e
  int32_t eta = -1;
  for (int it = 0; it < 26; ++it) {
    const uint32_t f0 = (uint32_t)__builtin_amdgcn_mov_dpp(own.l[0], quad_perm(0, 0, 0, 0), 0xF, 0xF, true);
    const uint32_t g0 = (uint32_t)__builtin_amdgcn_mov_dpp(own.l[0], quad_perm(1, 1, 1, 1), 0xF, 0xF, true);
    trans2x2 t;
    eta = divsteps_29_var(eta, f0, g0, t);
    quad_gcd_update(own, t, k);
    int32_t nz = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) nz |= own.l[i];
    if (__all(k != 1 || nz == 0)) break;  // g == 0 on every quad of the wave
  }
  const int32_t sf = __builtin_amdgcn_mov_dpp(own.l[NL - 1], quad_perm(0, 0, 0, 0), 0xF, 0xF, true) >> 31;
  const fe d = fe_dpp<quad_perm(2, 2, 2, 2)>(own);
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

template <int CTRL>
__device__ __forceinline__ double f64_dpp(double v) {
  const uint64_t u = lehmer_bits(v);
  int32_t lo = __builtin_amdgcn_mov_dpp((int32_t)(uint32_t)u, CTRL, 0xF, 0xF, true);
  int32_t hi = __builtin_amdgcn_mov_dpp((int32_t)(uint32_t)(u >> 32), CTRL, 0xF, 0xF, true);
  asm volatile("" : "+v"(lo), "+v"(hi));
  return lehmer_from_bits((uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)hi << 32));
}

// The same quad split for the double-steered inversion (fp29.hpp "inversion steered by doubles"): lane 0 keeps
// A, 1: B, 2: D, 3: E.  Per batch every lane turns its own vector into a double (lanes 0 and 1 broadcast
// theirs), all four run the same ~10 Euclid steps on the doubles, and each applies its own row of the integer
// matrix: 18 multiply-adds and one carry chain.  ~10 batches instead of the 18.4 of the divsteps form, ~100
// FP64 Euclid steps instead of ~140 integer divstep iterations of twice the length.
// x: any N-form integer (limbs 0..7 in [0, 2^29), |value| < 16 p), identical on the four lanes of every quad
// -> canonical x^-1 mod p on every lane (0 for a multiple of p).  The gcd does not care whether x is reduced:
// A E - B D = +-p holds from the start, so |D| < 2p whatever the size of x.
__device__ __forceinline__ fe fe_inv_plain_quad(const fe& x, int k) {
  const fe one = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  const bool odd = (k & 1) != 0;
  fe own = k == 0 ? FE_P : (k == 1 ? x : (k == 2 ? FE_ZERO : one));  // A | B | D | E
  double ad = 0.0;
  bool ok = true;
  for (int it = 0; it < LEHMER_MAX_BATCHES; ++it) {
    const double od = lehmer_to_double(own);
    ad = f64_dpp<quad_perm(0, 0, 0, 0)>(od);
    const double bd = f64_dpp<quad_perm(1, 1, 1, 1)>(od);
    lehmer_rows m;
    ok &= lehmer_batch(__builtin_fabs(ad), __builtin_fabs(bd), m);
    if (__all(bd == 0.0 || !ok)) break;  // B == 0 on every quad of the wave (or the wave falls back)
    // the batch ran on (|A|, |B|): the signs of A and B go into the columns of the matrix
    const double co = odd ? lehmer_flip(m.vb, bd) : lehmer_flip(m.ua, ad);
    const double cp = odd ? lehmer_flip(m.ub, ad) : lehmer_flip(m.va, bd);
    own = lehmer_row(own, fe_dpp<quad_perm(1, 0, 3, 2)>(own), (int32_t)co, (int32_t)cp);
  }
  {
    // out of batches with a remainder left = not converged (fallback below); A is re-read, the loop leaves `ad`
    // one batch old
    const double od = lehmer_to_double(own);
    ad = f64_dpp<quad_perm(0, 0, 0, 0)>(od);
    ok &= f64_dpp<quad_perm(1, 1, 1, 1)>(od) == 0.0;
  }
  if (__any(!ok)) {  // a partial quotient above 2^27 somewhere in the wave: the divsteps form wants [0, p)
    return fe_inv_plain_quad_divsteps(fe_canon(fe_mul(x, FE_ONE_M)), k);  // x * R / R: the value, reduced
  }
  const fe d = __builtin_fabs(ad) == 1.0 ? fe_dpp<quad_perm(2, 2, 2, 2)>(own) : FE_ZERO;  // gcd = p: x = 0 mod p
  return lehmer_finish(d, ad < 0.0 ? -1 : 0);  // A = +-1; D x = A
}

// Montgomery-form inverse; a (N-form, |value| < 16p) must be identical on the four lanes of every quad.
// The plain inverse of the representative a R is a^-1 R^-1; times R^3 (Montgomery product) gives a^-1 R.
// PLAIN = true: times R^2 instead, i.e. the inverse WITHOUT the Montgomery factor (N-form limbs of a^-1): a
// caller that only multiplies it into Montgomery-form values and then leaves the Montgomery domain
// (x = X / ZZ of a hash) gets the plain result from that product directly - fe_mul(X R, ZZ^-1) = X / ZZ - and
// saves the reduction pass of fe_from_mont per output.
template <bool PLAIN = false>
__device__ __forceinline__ fe fe_inv_quad(const fe& a, int k) {
  return fe_mul(fe_inv_plain_quad(a, k), PLAIN ? FE_R2 : FE_R3);
}


// One inversion for the four lanes of a quad that hold DIFFERENT values (Montgomery's trick across the
// lanes): the quad multiplies its values together with two DPP rounds, inverts the product once with the
// quad-split inversion above (divsteps form: 6.8 k instead of 4 x 12.9 k lane-private instructions; the
// double-steered form: ~2.8 k, tools/ubench/inv_quad.hip)
// and every lane recovers its own inverse with two more multiplications.
//   LOG_DISTINCT = 0: the four lanes hold the same value            (fe_inv_quad)
//                  1: lanes {0,1} hold one value, lanes {2,3} another
//                  2: four different values
// a: Montgomery N-form, non-zero (callers replace a zero - an exceptional addition - by one beforehand:
// a single zero would spoil the three other inverses of its quad).  All four lanes must be active.
// PLAIN as in fe_inv_quad: the shared inverse comes back without the Montgomery factor, and so does every lane's
// own inverse (plain x Montgomery-form cofactor = plain).
template <int LOG_DISTINCT, bool PLAIN = false>
__device__ __forceinline__ fe fe_inv_shared_quad(const fe& a, int k) {
  if constexpr (LOG_DISTINCT == 0) {
    return fe_inv_quad<PLAIN>(a, k);
  } else if constexpr (LOG_DISTINCT == 1) {
    const fe other = fe_dpp<quad_perm(2, 3, 0, 1)>(a);
    const fe pinv = fe_inv_quad<PLAIN>(fe_mul(a, other), k);
    return fe_mul(pinv, other);
  } else {
    const fe nb = fe_dpp<quad_perm(1, 0, 3, 2)>(a);       // the pair partner's value
    const fe pair = fe_mul(a, nb);                          // a0 a1 | a0 a1 | a2 a3 | a2 a3
    const fe opp = fe_dpp<quad_perm(2, 3, 0, 1)>(pair);    // the other pair's product
    const fe pinv = fe_inv_quad<PLAIN>(fe_mul(pair, opp), k);
    return fe_mul(pinv, fe_mul(nb, opp));
  }
}

}  // namespace sp

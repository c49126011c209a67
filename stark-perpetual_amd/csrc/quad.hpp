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

}  // namespace sp

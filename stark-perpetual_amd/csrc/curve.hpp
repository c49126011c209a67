// Group law on the Stark curve y^2 = x^3 + x + beta over GF(p) (reference: affine chord/tangent
// with one inversion per operation, starkware/crypto/signature/math_utils.py:59-88).  Here the
// same group elements are carried in inversion-free coordinates:
//   * XYZZ  (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) for sums of precomputed affine table points
//     (mixed add 8M + 2S) - Pedersen hash, fixed-base k*G;
//   * Jacobian (x = X/Z^2, y = Y/Z^3) for the variable-base ladder of ECDSA verification
//     (doubling 2M + 8S with a general `a`, mixed add 7M + 4S).
// Results are only ever compared / exported as affine x (or (x, y)), which is a function of the
// group element alone, so the output is bit-identical to the reference's affine arithmetic.
//
// Limb-bound notes use B = max|limb| / 2^29 (see fp29.hpp).  N = output of mul/sqr/carry (B = 1).
#pragma once
#include "fp29.hpp"

namespace sp {

struct aff {
  fe x, y;  // Montgomery form, N
};
struct xyzz {
  fe X, Y, ZZ, ZZZ;  // Montgomery form, N.  ZZ == 0  <=>  point at infinity
};
struct jac {
  fe X, Y, Z;
};

SP_HD xyzz xyzz_from_aff(const aff& p) {
  xyzz r;
  r.X = p.x;
  r.Y = p.y;
  r.ZZ = FE_ONE_M;
  r.ZZZ = FE_ONE_M;
  return r;
}

// acc + q, q affine ("madd-2008-s").  Exceptional inputs (acc == +-q, acc == infinity) are not
// handled here: they drive ZZ to 0, which stays 0 through later additions and is detected by the
// caller on the final ZZ.  Call sites argue why they are unreachable.
SP_HD xyzz xyzz_madd(const xyzz& a, const aff& q) {
  const fe U2 = fe_mul(q.x, a.ZZ);           // N
  const fe S2 = fe_mul(q.y, a.ZZZ);          // N
  const fe P = fe_sub(U2, a.X);              // B=1 (signed)
  const fe R = fe_sub(S2, a.Y);              // B=1
  const fe PP = fe_sqr(P);                   // cols < 9*2^58
  const fe PPP = fe_mul(P, PP);
  const fe Q = fe_mul(a.X, PP);
  xyzz r;
  // X3 = R^2 - PPP - 2Q : limbs in (-3*2^29, 2^29) -> carry to N
  r.X = fe_carry(fe_sub(fe_sub(fe_sqr(R), PPP), fe_dbl(Q)));
  // Y3 = R (Q - X3) - Y1 PPP : both products share one reduction, cols < 9*2^58 + 9*2^58
  r.Y = fe_mul_sub_mul(R, fe_sub(Q, r.X), a.Y, PPP);
  r.ZZ = fe_mul(a.ZZ, PP);
  r.ZZZ = fe_mul(a.ZZZ, PPP);
  return r;
}

// Last addition of a chain when only x = X/ZZ is wanted: skips Y3 and ZZZ3 (saves 3M).
SP_HD void xyzz_madd_x_only(const xyzz& a, const aff& q, fe& X3, fe& ZZ3) {
  const fe U2 = fe_mul(q.x, a.ZZ);
  const fe S2 = fe_mul(q.y, a.ZZZ);
  const fe P = fe_sub(U2, a.X);
  const fe R = fe_sub(S2, a.Y);
  const fe PP = fe_sqr(P);
  const fe PPP = fe_mul(P, PP);
  const fe Q = fe_mul(a.X, PP);
  X3 = fe_carry(fe_sub(fe_sub(fe_sqr(R), PPP), fe_dbl(Q)));
  ZZ3 = fe_mul(a.ZZ, PP);
}

// Sum of two affine points -> XYZZ ("mmadd-2008-s", 4M + 2S).
SP_HD xyzz xyzz_mmadd(const aff& a, const aff& b) {
  const fe P = fe_sub(b.x, a.x);
  // callers pass table entries whose y may be negated (signed windows): b.y - a.y can reach B = 2 and
  // nine maximal products of its square would exceed the 64-bit column budget - carry it to N
  const fe R = fe_carry(fe_sub(b.y, a.y));
  const fe PP = fe_sqr(P);
  const fe PPP = fe_mul(P, PP);
  const fe Q = fe_mul(a.x, PP);
  xyzz r;
  r.X = fe_carry(fe_sub(fe_sub(fe_sqr(R), PPP), fe_dbl(Q)));
  r.Y = fe_mul_sub_mul(R, fe_sub(Q, r.X), a.y, PPP);
  r.ZZ = PP;
  r.ZZZ = PPP;
  return r;
}

// General XYZZ + XYZZ ("add-2008-s", 12M + 2S); used for cross-lane combines.
SP_HD xyzz xyzz_add(const xyzz& a, const xyzz& b) {
  const fe U1 = fe_mul(a.X, b.ZZ);
  const fe U2 = fe_mul(b.X, a.ZZ);
  const fe S1 = fe_mul(a.Y, b.ZZZ);
  const fe S2 = fe_mul(b.Y, a.ZZZ);
  const fe P = fe_sub(U2, U1);
  const fe R = fe_sub(S2, S1);
  const fe PP = fe_sqr(P);
  const fe PPP = fe_mul(P, PP);
  const fe Q = fe_mul(U1, PP);
  xyzz r;
  r.X = fe_carry(fe_sub(fe_sub(fe_sqr(R), PPP), fe_dbl(Q)));
  r.Y = fe_mul_sub_mul(R, fe_sub(Q, r.X), S1, PPP);
  r.ZZ = fe_mul(fe_mul(a.ZZ, b.ZZ), PP);
  r.ZZZ = fe_mul(fe_mul(a.ZZZ, b.ZZZ), PPP);
  return r;
}

// Last combine of a butterfly when only x = X/ZZ is wanted: skips Y3 and ZZZ3 (saves 5M).
SP_HD void xyzz_add_x_only(const xyzz& a, const xyzz& b, fe& X3, fe& ZZ3) {
  const fe U1 = fe_mul(a.X, b.ZZ);
  const fe U2 = fe_mul(b.X, a.ZZ);
  const fe S1 = fe_mul(a.Y, b.ZZZ);
  const fe S2 = fe_mul(b.Y, a.ZZZ);
  const fe P = fe_sub(U2, U1);
  const fe R = fe_sub(S2, S1);
  const fe PP = fe_sqr(P);
  const fe PPP = fe_mul(P, PP);
  const fe Q = fe_mul(U1, PP);
  X3 = fe_carry(fe_sub(fe_sub(fe_sqr(R), PPP), fe_dbl(Q)));
  ZZ3 = fe_mul(fe_mul(a.ZZ, b.ZZ), PP);
}

// ---- Jacobian, general curve coefficient a (Montgomery form) ----
// "dbl-2007-bl": 2M + 8S (one M is a * ZZ^2).
SP_HD jac jac_dbl(const jac& p, const fe& a_coef) {
  const fe XX = fe_sqr(p.X);
  const fe YY = fe_sqr(p.Y);
  const fe YYYY = fe_sqr(YY);
  const fe ZZ = fe_sqr(p.Z);
  // S = 2((X + YY)^2 - XX - YYYY).  (X + YY) has B=2 and 9*2^60 exceeds the signed column
  // budget, so it is carried first; every lazy expression below stays within (-2^31, 2^31).
  const fe t0 = fe_carry(fe_add(p.X, YY));
  const fe halfS = fe_carry(fe_sub(fe_sub(fe_sqr(t0), XX), YYYY));  // limbs (-2^30, 2^29)
  const fe S = fe_carry(fe_dbl(halfS));
  // M = 3 XX + a ZZ^2
  const fe XX3 = fe_carry(fe_add(fe_dbl(XX), XX));
  const fe M = fe_carry(fe_add(XX3, fe_mul(a_coef, fe_sqr(ZZ))));
  jac r;
  r.X = fe_carry(fe_sub(fe_sqr(M), fe_dbl(S)));  // T, limbs (-2^30, 2^29)
  // Y3 = M (S - T) - 8 YYYY
  const fe y4 = fe_carry(fe_dbl(fe_dbl(YYYY)));
  r.Y = fe_carry(fe_sub(fe_mul(M, fe_sub(S, r.X)), fe_dbl(y4)));
  // Z3 = (Y + Z)^2 - YY - ZZ
  const fe t1 = fe_carry(fe_add(p.Y, p.Z));
  r.Z = fe_carry(fe_sub(fe_sub(fe_sqr(t1), YY), ZZ));
  return r;
}

// Modified Jacobian doubling (Cohen - Miyaji - Ono): the point carries W = a Z^4, so a doubling is 4M + 4S
//   S = 4 X Y^2,  U = 8 Y^4,  M = 3 X^2 + W,  X3 = M^2 - 2 S,  Y3 = M (S - X3) - U,  Z3 = 2 Y Z,  W3 = 2 U W
// and k doublings in a row cost 4k M + (4k + 2) S (W once: 1M + 2S; the last W3 is not needed) against
// k (2M + 8S) for jac_dbl: 16M + 18S instead of 8M + 32S for the four doublings of a ladder window.
struct mjac {
  fe X, Y, Z, W;
};
SP_HD mjac mjac_from(const jac& p, const fe& a_coef) {
  mjac m;
  m.X = p.X; m.Y = p.Y; m.Z = p.Z;
  m.W = fe_mul(a_coef, fe_sqr(fe_sqr(p.Z)));
  return m;
}
SP_HD void mjac_dbl(mjac& p, bool need_w) {
  const fe XX = fe_sqr(p.X);
  const fe YY = fe_sqr(p.Y);
  const fe YYYY = fe_sqr(YY);
  const fe S = fe_carry(fe_dbl(fe_carry(fe_dbl(fe_mul(p.X, YY)))));                    // 4 X YY
  const fe U = fe_carry(fe_dbl(fe_carry(fe_dbl(fe_carry(fe_dbl(YYYY))))));             // 8 YYYY
  const fe M = fe_carry(fe_add(fe_carry(fe_add(fe_dbl(XX), XX)), p.W));
  const fe X3 = fe_carry(fe_sub(fe_sqr(M), fe_dbl(S)));
  const fe Y3 = fe_carry(fe_sub(fe_mul(M, fe_carry(fe_sub(S, X3))), U));
  const fe Z3 = fe_carry(fe_dbl(fe_mul(p.Y, p.Z)));
  if (need_w) p.W = fe_carry(fe_dbl(fe_mul(U, p.W)));
  p.X = X3;
  p.Y = Y3;
  p.Z = Z3;
}

// "madd-2007-bl": Jacobian + affine, 7M + 4S.  Exceptional cases drive Z to 0 (see xyzz_madd).
SP_HD jac jac_madd(const jac& p, const aff& q) {
  const fe Z1Z1 = fe_sqr(p.Z);
  const fe U2 = fe_mul(q.x, Z1Z1);
  const fe S2 = fe_mul(fe_mul(q.y, p.Z), Z1Z1);
  const fe H = fe_sub(U2, p.X);                    // B=1
  const fe HH = fe_sqr(H);
  const fe I = fe_dbl(fe_carry(fe_dbl(HH)));       // 4 HH, B=2 (lazy)
  const fe J = fe_mul(H, I);
  const fe rr = fe_carry(fe_dbl(fe_sub(S2, p.Y)));  // 2 (S2 - Y1) -> N
  const fe V = fe_mul(p.X, I);
  jac r;
  r.X = fe_carry(fe_sub(fe_sub(fe_sqr(rr), J), fe_dbl(V)));
  // Y3 = rr (V - X3) - 2 Y1 J
  r.Y = fe_mul_sub_mul(rr, fe_sub(V, r.X), fe_carry(fe_dbl(p.Y)), J);
  // Z3 = (Z1 + H)^2 - Z1Z1 - HH
  const fe t = fe_carry(fe_add(p.Z, H));
  r.Z = fe_carry(fe_sub(fe_sub(fe_sqr(t), Z1Z1), HH));
  return r;
}

// "add-2007-bl": Jacobian + Jacobian, 11M + 5S.  Exceptional cases drive Z to 0 (see xyzz_madd).
SP_HD jac jac_add(const jac& p, const jac& q) {
  const fe Z1Z1 = fe_sqr(p.Z);
  const fe Z2Z2 = fe_sqr(q.Z);
  const fe U1 = fe_mul(p.X, Z2Z2);
  const fe U2 = fe_mul(q.X, Z1Z1);
  const fe S1 = fe_mul(fe_mul(p.Y, q.Z), Z2Z2);
  const fe S2 = fe_mul(fe_mul(q.Y, p.Z), Z1Z1);
  const fe H = fe_sub(U2, U1);                      // B=1
  const fe I = fe_dbl(fe_carry(fe_dbl(fe_sqr(H))));  // 4 H^2, B=2 (lazy)
  const fe J = fe_mul(H, I);
  const fe rr = fe_carry(fe_dbl(fe_sub(S2, S1)));   // 2 (S2 - S1) -> N
  const fe V = fe_mul(U1, I);
  jac r;
  r.X = fe_carry(fe_sub(fe_sub(fe_sqr(rr), J), fe_dbl(V)));
  // Y3 = rr (V - X3) - 2 S1 J
  r.Y = fe_mul_sub_mul(rr, fe_sub(V, r.X), fe_carry(fe_dbl(S1)), J);
  // Z3 = ((Z1 + Z2)^2 - Z1Z1 - Z2Z2) H
  const fe t = fe_carry(fe_add(p.Z, q.Z));
  r.Z = fe_mul(fe_carry(fe_sub(fe_sub(fe_sqr(t), Z1Z1), Z2Z2)), H);
  return r;
}

}  // namespace sp

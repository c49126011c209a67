// Host-side affine group law (one field inversion per operation, like the reference's
// math_utils.py:59-100).  Used once at sp_init to derive the per-bit constant points and the
// window-table offset points that seed the device table builder.  Not a compute path.
#pragma once
#include "curve.hpp"
#include "curve_consts.hpp"

namespace sp {

struct haff {
  fe x, y;  // Montgomery form, canonical
  bool inf;
};

inline fe h_canon_m(const fe& a) {  // canonical Montgomery representative
  return fe_to_mont(fe_from_mont(a));
}
inline haff h_make(const u256& x, const u256& y) {
  haff r;
  r.x = fe_canon(fe_to_mont(fe_unpack(x)));
  r.y = fe_canon(fe_to_mont(fe_unpack(y)));
  r.inf = false;
  return r;
}
inline haff h_neg(const haff& a) {
  haff r = a;
  if (!a.inf) r.y = h_canon_m(fe_carry(fe_neg(a.y)));
  return r;
}
inline haff h_dbl(const haff& a) {
  if (a.inf || fe_is_zero(a.y)) return haff{FE_ZERO, FE_ZERO, true};
  // lambda = (3x^2 + 1) / (2y)
  fe xx = fe_sqr(a.x);
  fe num = fe_carry(fe_add(fe_carry(fe_add(fe_dbl(xx), xx)), FE_ONE_M));
  fe lam = fe_mul(num, fe_inv(fe_carry(fe_dbl(a.y))));
  haff r;
  r.x = fe_carry(fe_sub(fe_sqr(lam), fe_dbl(a.x)));
  r.y = fe_carry(fe_sub(fe_mul(lam, fe_sub(a.x, r.x)), a.y));
  r.x = h_canon_m(r.x);
  r.y = h_canon_m(r.y);
  r.inf = false;
  return r;
}
inline haff h_add(const haff& a, const haff& b) {
  if (a.inf) return b;
  if (b.inf) return a;
  if (fe_eq(a.x, b.x)) {
    if (fe_eq(a.y, b.y)) return h_dbl(a);
    return haff{FE_ZERO, FE_ZERO, true};
  }
  fe lam = fe_mul(fe_sub(b.y, a.y), fe_inv(fe_carry(fe_sub(b.x, a.x))));
  haff r;
  r.x = fe_carry(fe_sub(fe_sub(fe_sqr(lam), a.x), b.x));
  r.y = fe_carry(fe_sub(fe_mul(lam, fe_sub(a.x, r.x)), a.y));
  r.x = h_canon_m(r.x);
  r.y = h_canon_m(r.y);
  r.inf = false;
  return r;
}
// k * a, k given as 4 x uint64 little-endian
inline haff h_mul(const uint64_t k[4], const haff& a) {
  haff r{FE_ZERO, FE_ZERO, true};
  for (int i = 255; i >= 0; --i) {
    r = h_dbl(r);
    if ((k[i >> 6] >> (i & 63)) & 1) r = h_add(r, a);
  }
  return r;
}

}  // namespace sp

// Host-side test shim: compiles the device arithmetic headers with g++ (bound checks on) and
// exposes plain-integer entry points so tests/test_field_host.py can compare against Python ints.
// Test infrastructure only - never loaded by the product.
#define SP_CHECK_BOUNDS 1
#include "../../stark-perpetual_amd/csrc/curve.hpp"
#include <string.h>
using namespace sp;

#if defined(SP_LEHMER_RCP_ERROR)
// second build of this shim (tests/test_field_host.py::test_lehmer_with_an_imprecise_reciprocal): the Euclid steps
// are steered by a reciprocal with a chosen relative error, and the batches of every inversion are counted
namespace sp {
double sp_lehmer_rcp_error = 0x1p-24;
int sp_lehmer_batches = 0;
int sp_lehmer_max_batches = LEHMER_MAX_BATCHES;
}
extern "C" void t_set_lehmer_budget(int n) { sp::sp_lehmer_max_batches = n; }
extern "C" void t_set_rcp_error(double e) { sp::sp_lehmer_rcp_error = e; }
extern "C" int t_take_lehmer_batches() { const int v = sp::sp_lehmer_batches; sp::sp_lehmer_batches = 0; return v; }
// 1: the double-steered form converged and answered; 0: it asked for the divsteps fallback
extern "C" int t_lehmer_bezout_ok(const uint32_t* a) {
  u256 w; memcpy(w.w, a, 32);
  fe D; int32_t sf;
  return lehmer_bezout(FE_P, fe_unpack(w), D, sf) ? 1 : 0;
}
#endif

static fe load_plain(const uint32_t* w) { u256 a; memcpy(a.w, w, 32); return fe_unpack(a); }
static void store_plain(const fe& canon, uint32_t* w) { u256 r = fe_pack(canon); memcpy(w, r.w, 32); }
static fe to_m(const uint32_t* w) { return fe_to_mont(load_plain(w)); }
static void from_m(const fe& a, uint32_t* w) { store_plain(fe_from_mont(a), w); }
static fe to_mn(const uint32_t* w) { return fn_to_mont(load_plain(w)); }
static void from_mn(const fe& a, uint32_t* w) { store_plain(fn_from_mont(a), w); }

extern "C" {
void t_roundtrip(const uint32_t* a, uint32_t* out) { store_plain(load_plain(a), out); }
void t_fe_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) { from_m(fe_mul(to_m(a), to_m(b)), out); }
void t_fe_sqr(const uint32_t* a, uint32_t* out) { from_m(fe_sqr(to_m(a)), out); }
void t_fe_inv(const uint32_t* a, uint32_t* out) { from_m(fe_inv(to_m(a)), out); }
void t_fe_inv_fermat(const uint32_t* a, uint32_t* out) { from_m(fe_inv_fermat(to_m(a)), out); }
void t_fn_inv_fermat(const uint32_t* a, uint32_t* out) { from_mn(fn_inv_fermat(to_mn(a)), out); }
void t_fe_inv_gcd(const uint32_t* a, uint32_t* out) { from_m(fe_inv_gcd(to_m(a)), out); }
void t_fe_inv_gcd_var(const uint32_t* a, uint32_t* out) { from_m(fe_inv_gcd_var(to_m(a)), out); }
void t_fe_inv_plain_gcd_var(const uint32_t* a, uint32_t* out) { store_plain(fe_inv_plain_gcd_var(load_plain(a)), out); }
void t_fe_inv_lehmer_lazy(const uint32_t* a, int k, uint32_t* out) {  // Montgomery form + k p, uncarried
  fe v = to_m(a);
  for (int i = 0; i < (k < 0 ? -k : k); ++i) v = k < 0 ? fe_sub(v, FE_P) : fe_add(v, FE_P);
  from_m(fe_inv_lehmer(v), out);
}
void t_fe_inv_plain_lehmer(const uint32_t* a, uint32_t* out) { store_plain(fe_inv_plain_lehmer(load_plain(a)), out); }
// one batch of the double-steered Euclid on (|A|, |B|): rows out, returns 1 when the batch is representable
int t_lehmer_batch(const uint32_t* a, const uint32_t* b, double* rows) {
  lehmer_rows m;
  const bool ok = lehmer_batch(__builtin_fabs(lehmer_to_double(load_plain(a))), __builtin_fabs(lehmer_to_double(load_plain(b))), m);
  rows[0] = m.ua; rows[1] = m.va; rows[2] = m.ub; rows[3] = m.vb;
  return ok ? 1 : 0;
}
void t_fe_half(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  // half of (a - b) as plain integers: exercises negative and lazy inputs
  store_plain(fe_canon(fe_half(fe_sub(load_plain(a), load_plain(b)))), out);
}
// 2^k P on y^2 = x^3 + a x + b by k modified-Jacobian doublings (mjac_dbl) and by k jac_dbl: affine x, y of both
static void jac_to_aff_plain(const fe& X, const fe& Y, const fe& Z, uint32_t* x, uint32_t* y) {
  const fe zi = fe_inv(Z), zi2 = fe_sqr(zi);
  from_m(fe_mul(X, zi2), x);
  from_m(fe_mul(Y, fe_mul(zi2, zi)), y);
}
void t_repeated_doubling(const uint32_t* px, const uint32_t* py, const uint32_t* pz, const uint32_t* a, int k,
                         uint32_t* x_m, uint32_t* y_m, uint32_t* x_j, uint32_t* y_j) {
  const fe z = to_m(pz), z2 = fe_sqr(z);
  jac p;
  p.X = fe_mul(to_m(px), z2); p.Y = fe_mul(to_m(py), fe_mul(z2, z)); p.Z = z;
  const fe am = to_m(a);
  mjac m = mjac_from(p, am);
  for (int i = 0; i < k; ++i) mjac_dbl(m, i + 1 < k);
  jac_to_aff_plain(m.X, m.Y, m.Z, x_m, y_m);
  jac q = p;
  for (int i = 0; i < k; ++i) q = jac_dbl(q, am);
  jac_to_aff_plain(q.X, q.Y, q.Z, x_j, y_j);
}
// fe_canon on raw limbs (signed, possibly lazy: the caller builds N-form / lazy patterns directly)
void t_fe_canon_limbs(const int32_t* limbs, uint32_t* out) {
  fe a;
  for (int i = 0; i < NL; ++i) a.l[i] = limbs[i];
  store_plain(fe_canon(a), out);
}
void t_fe_inv_plain_gcd(const uint32_t* a, uint32_t* out) { store_plain(fe_inv_plain_gcd(load_plain(a)), out); }
int t_fe_is_qr(const uint32_t* a) { return fe_is_qr(to_m(a)) ? 1 : 0; }
// (a - b) * (c + d) - e*f : exercises lazy add/sub feeding products
void t_fe_expr(const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d,
               const uint32_t* e, const uint32_t* f, uint32_t* out) {
  fe r = fe_mul_sub_mul(fe_sub(to_m(a), to_m(b)), fe_carry(fe_add(to_m(c), to_m(d))), to_m(e), to_m(f));
  from_m(r, out);
}
void t_fn_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) { from_mn(fn_mul(to_mn(a), to_mn(b)), out); }
void t_fn_inv(const uint32_t* a, uint32_t* out) { from_mn(fn_inv(to_mn(a)), out); }
void t_fn_inv_var(const uint32_t* a, uint32_t* out) { from_mn(fn_inv_var(to_mn(a)), out); }
void t_fn_inv_divsteps_var(const uint32_t* a, uint32_t* out) { from_mn(fn_inv_plain_divsteps_var(fn_canon(fn_mul(to_mn(a), FN_ONE_M))), out); }

static void xyzz_to_aff_plain(const xyzz& p, uint32_t* x, uint32_t* y) {
  fe izzz = fe_inv(p.ZZZ);
  fe izz = fe_mul(fe_sqr(fe_mul(p.ZZ, izzz)), FE_ONE_M);  // 1/ZZ = (ZZ/ZZZ)^2
  from_m(fe_mul(p.X, izz), x);
  from_m(fe_mul(p.Y, izzz), y);
}
// chain: ((p0 + p1) + p2) + ... with mmadd for the first pair then madd; n >= 2 affine points
void t_xyzz_chain(const uint32_t* xs, const uint32_t* ys, int n, uint32_t* x, uint32_t* y) {
  aff a{to_m(xs), to_m(ys)}, b{to_m(xs + 8), to_m(ys + 8)};
  xyzz acc = xyzz_mmadd(a, b);
  for (int i = 2; i < n; ++i) { aff q{to_m(xs + 8 * i), to_m(ys + 8 * i)}; acc = xyzz_madd(acc, q); }
  xyzz_to_aff_plain(acc, x, y);
}
void t_xyzz_add(const uint32_t* xs, const uint32_t* ys, uint32_t* x, uint32_t* y) {
  // (p0+p1) + (p2+p3) via the general add
  aff p0{to_m(xs), to_m(ys)}, p1{to_m(xs + 8), to_m(ys + 8)}, p2{to_m(xs + 16), to_m(ys + 16)},
      p3{to_m(xs + 24), to_m(ys + 24)};
  xyzz_to_aff_plain(xyzz_add(xyzz_mmadd(p0, p1), xyzz_mmadd(p2, p3)), x, y);
}
// Jacobian full addition: (2*P0 + P1') + (P2 via dbl of P2) style mix: computes a*P + b*P style sum
// via jac_add(jac(k1*P), jac(k2*P)) for small k1, k2 built by doubling/madd
void t_jac_add(const uint32_t* px, const uint32_t* py, const uint32_t* qx, const uint32_t* qy, uint32_t* x,
               uint32_t* y) {
  aff p{to_m(px), to_m(py)}, q{to_m(qx), to_m(qy)};
  jac a{p.x, p.y, FE_ONE_M}, b{q.x, q.y, FE_ONE_M};
  a = jac_dbl(a, FE_ONE_M);          // 2P  (Z != 1)
  b = jac_madd(jac_dbl(b, FE_ONE_M), q);  // 3Q  (Z != 1)
  jac r = jac_add(a, b);             // 2P + 3Q
  fe iz = fe_inv(r.Z), iz2 = fe_sqr(iz);
  from_m(fe_mul(r.X, iz2), x);
  from_m(fe_mul(r.Y, fe_mul(iz2, iz)), y);
}
// Jacobian double-and-add: k * P with plain scalar bits (MSB first), a = 1
void t_jac_mul(const uint32_t* px, const uint32_t* py, const uint32_t* k, uint32_t* x, uint32_t* y) {
  aff q{to_m(px), to_m(py)};
  jac r{q.x, q.y, FE_ONE_M};
  int top = 255;
  while (top >= 0 && !((k[top >> 5] >> (top & 31)) & 1)) --top;
  for (int i = top - 1; i >= 0; --i) {
    r = jac_dbl(r, FE_ONE_M);
    if ((k[i >> 5] >> (i & 31)) & 1) r = jac_madd(r, q);
  }
  fe iz = fe_inv(r.Z), iz2 = fe_sqr(iz);
  from_m(fe_mul(r.X, iz2), x);
  from_m(fe_mul(r.Y, fe_mul(iz2, iz)), y);
}
}

// Stark-curve ECDSA on the GPU: verify, one-attempt sign, public-key derivation.
//
// Reference semantics (signature.py:217-260): with w = s^-1 mod N the signature is valid iff
// r == x( w * (z*G + r*Q) ), evaluated there as three LSB-first 251-step affine ladders that
// "mimic the AIR" (signature.py:176-190) and return False whenever a ladder step would hit an
// exceptional case.  The group element is the same as  u1*G + u2*Q  with u1 = z w, u2 = r w mod N,
// which is what this kernel evaluates:
//   * u1*G  : sum of nwin entries of the EC_GEN window table (XYZZ mixed adds);
//   * u2*Q  : Jacobian fixed signed-window ladder (w = 4) on the per-item base point;
//   * x-only public keys (signature.py:229-238) are handled WITHOUT a modular square root (the
//     field has 2-adicity 192, Tonelli-Shanks would cost more than the rest of the verification):
//     with c = x^3 + x + beta and a formal Y, Y^2 = c, the multiples of Q = (x, Y) are (a_k, b_k Y),
//     i.e. rational points of the isomorphic curve y'^2 = x'^3 + c^2 x' + beta c^3 under
//     x' = c x, y' = c^2 b.  The ladder runs on that curve from (c x, c^2); the acceptance test
//     r in { x(A + B), x(A - B) } becomes one polynomial identity  E^2 == 4 ya^2 t^2 c  plus the
//     Legendre test "c is a square" (InvalidPublicKeyError -> False, signature.py:232-235).
// Reachable reference failure modes and how they are reproduced here:
//   z == 0                      -> False   (mimic_ec_mult_air asserts 0 < m, signature.py:181)
//   z*G + r*Q == infinity       -> False   (ec_add x-collision, signature.py:254)
//   pre-assert violations       -> SP_VERIFY_ASSERT_* codes (signature.py:219,225-227,241)
// All other assertion sites of the reference ladders (partial sum meeting the doubled point) need
// a discrete-log relation between the shift point and G or Q (DESIGN.md "failure set").
#include <map>

#include "context.hpp"
#include "curve_consts.hpp"

namespace sp {

// k * EC_GEN from the window table; k < 2^252.  Infinity (only for k == 0 mod N) shows as ZZ == 0.
__device__ __forceinline__ xyzz gen_mul(u256 k, const aff_packed* __restrict__ gen, int wbits, int nwin) {
  const size_t per = (size_t)1 << wbits;
  const uint32_t mask = (1u << wbits) - 1u;
  auto pop = [&](void) {
    const uint32_t v = k.w[0] & mask;
#pragma unroll
    for (int i = 0; i < 7; ++i) k.w[i] = (k.w[i] >> wbits) | (k.w[i + 1] << (32 - wbits));
    k.w[7] >>= wbits;
    return v;
  };
  xyzz acc = xyzz_from_aff(ld_aff(gen + pop()));
  for (int i = 1; i < nwin; ++i) acc = xyzz_madd(acc, ld_aff(gen + (size_t)i * per + pop()));
  return acc;
}

__device__ __forceinline__ u256 reduce_mod_p(u256 a) {
  // a < 2^256 < 32 p
  for (int it = 0; it < 32 && !u256_lt(a, U256_P); ++it) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint64_t d = (uint64_t)a.w[i] - U256_P.w[i] - borrow;
      a.w[i] = (uint32_t)d;
      borrow = (d >> 32) & 1u;
    }
  }
  return a;
}

__device__ __forceinline__ fe mont_of(const u256& a) { return fe_to_mont(fe_unpack(a)); }
__device__ __forceinline__ fe montn_of(const u256& a) { return fn_to_mont(fe_unpack(a)); }

// x(2A) for affine A (Montgomery form), rare path.
__device__ __noinline__ fe double_x(const fe& xa, const fe& ya) {
  const fe xx = fe_sqr(xa);
  const fe num = fe_carry(fe_add(fe_carry(fe_add(fe_dbl(xx), xx)), FE_ONE_M));
  const fe lam = fe_mul(num, fe_inv(fe_carry(fe_dbl(ya))));
  return fe_carry(fe_sub(fe_sqr(lam), fe_dbl(xa)));
}

// ---- the three stages of a verification, shared by the ladder kernel and the key-table kernel ----
struct verify_scalars {
  u256 r;       // the signature's r (plain)
  u256 u1, u2;  // z w, r w mod N (plain)
  bool z_zero;  // msg_hash == 0: reference returns False (signature.py:181 via :252) AFTER the key checks
};
constexpr uint8_t VERIFY_CONTINUE = 0xff;

// Pre-asserts of signature.py:219-227 in the reference's order; on success fills u1, u2.
__device__ __forceinline__ uint8_t verify_prepare(const uint64_t* __restrict__ pz, const uint64_t* __restrict__ pr,
                                                  const uint64_t* __restrict__ ps, size_t e,
                                                  verify_scalars& v) {
  const u256 z = ld_u256(pz + 4 * e), r = ld_u256(pr + 4 * e), s = ld_u256(ps + 4 * e);
  if (u256_is_zero(s) || !u256_lt(s, U256_N)) return SP_VERIFY_ASSERT_S;  // :219
  const fe w_m = fn_inv(montn_of(s));
  const u256 w = fe_pack(fn_from_mont(w_m));
  if (u256_is_zero(r) || !u256_lt(r, U256_2P251)) return SP_VERIFY_ASSERT_R;  // :225
  if (u256_is_zero(w) || !u256_lt(w, U256_2P251)) return SP_VERIFY_ASSERT_W;  // :226
  if (!u256_lt(z, U256_2P251)) return SP_VERIFY_ASSERT_MSG;                   // :227
  v.r = r;
  v.z_zero = u256_is_zero(z);
  v.u1 = fe_pack(fn_from_mont(fn_mul(montn_of(z), w_m)));
  v.u2 = fe_pack(fn_from_mont(fn_mul(montn_of(r), w_m)));
  return VERIFY_CONTINUE;
}

// Regular signed recoding shared by the window ladder and the comb: for odd k,
//   k = sum_{j<256} (2 E_j - 1) 2^j   with   E = (k - 1)/2 + 2^255,
// so every 4-bit window of E is an odd digit 2e - 15 and every comb column has no zero row.
// Even scalars use N - k (odd) and the caller negates the result (`flip`).
__device__ __forceinline__ u256 recode_odd(const u256& u2, bool& flip) {
  flip = (u2.w[0] & 1u) == 0;
  u256 k = u2;
  if (flip) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint64_t d = (uint64_t)U256_N.w[i] - u2.w[i] - borrow;
      k.w[i] = (uint32_t)d;
      borrow = (d >> 32) & 1u;
    }
  }
  u256 E;  // (k - 1) / 2 + 2^255   (k odd: k - 1 clears bit 0)
#pragma unroll
  for (int i = 0; i < 7; ++i) E.w[i] = (k.w[i] >> 1) | (k.w[i + 1] << 31);
  E.w[7] = (k.w[7] >> 1) | 0x80000000u;
  return E;
}

// B' = u2 * base on y^2 = x^3 + a_coef x + ...  Fixed signed window, w = 4, regular recoding:
// the digits d_i = 2 e_i - 15 (all odd, |d_i| <= 15) are read straight off the 4-bit windows e_i
// of E, and the top digit is always +1 - every lane does the same 252 doublings + 63 additions
// (no divergent branch; the binary ladder paid a mixed addition on every bit because some lane
// always needed one).  The eight odd multiples (2j+1)*base live in a per-signature table in HBM,
// limb-major (entry * 27 + limb) * n + e.
__device__ __forceinline__ jac ladder_mul(const u256& u2, const aff& base, const fe& a_coef,
                                          int32_t* __restrict__ tab, size_t n, size_t e) {
  bool flip;
  u256 E = recode_odd(u2, flip);
  auto tab_at = [&](int entry, int limb) -> int32_t* { return tab + ((size_t)(entry * 27 + limb) * n + e); };
  auto tab_store = [&](int entry, const jac& P) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      *tab_at(entry, l) = P.X.l[l];
      *tab_at(entry, 9 + l) = P.Y.l[l];
      *tab_at(entry, 18 + l) = P.Z.l[l];
    }
  };
  jac B;
  B.X = base.x; B.Y = base.y; B.Z = FE_ONE_M;
  {
    tab_store(0, B);
    const jac twoQ = jac_dbl(B, a_coef);
    jac odd = jac_madd(twoQ, base);  // 3Q
    tab_store(1, odd);
    for (int j = 2; j < 8; ++j) {
      odd = jac_add(odd, twoQ);
      tab_store(j, odd);
    }
  }
  // top window is e = 8  ->  digit +1: start from Q itself, then 63 windows
  {
#pragma unroll
    for (int i = 7; i > 0; --i) E.w[i] = (E.w[i] << 4) | (E.w[i - 1] >> 28);
    E.w[0] <<= 4;
  }
  for (int wi = 0; wi < 63; ++wi) {
    const uint32_t ew = E.w[7] >> 28;
#pragma unroll
    for (int i = 7; i > 0; --i) E.w[i] = (E.w[i] << 4) | (E.w[i - 1] >> 28);
    E.w[0] <<= 4;
    const int d = 2 * (int)ew - 15;
    const int mag = (d < 0 ? -d : d) >> 1;  // table index of |d| = 2 mag + 1
    jac T;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      T.X.l[l] = *tab_at(mag, l);
      const int32_t y = *tab_at(mag, 9 + l);
      T.Y.l[l] = d < 0 ? -y : y;
      T.Z.l[l] = *tab_at(mag, 18 + l);
    }
    B = jac_dbl(jac_dbl(B, a_coef), a_coef);
    B = jac_dbl(jac_dbl(B, a_coef), a_coef);
    B = jac_add(B, T);
  }
  if (flip) B.Y = fe_neg(B.Y);
  return B;
}

// Acceptance test for A = u1 G (XYZZ, real curve) and B = u2 Q (Jacobian; on the real curve for a
// point key, c = 1, or on the c-twisted model for an x-only key): r == x(A + B), resp.
// r in { x(A + B), x(A - B) } as one polynomial identity.
__device__ __forceinline__ uint8_t verify_finish(const xyzz& A, const jac& B, const fe& c, bool has_y,
                                                 const u256& r) {
  // one inversion for 1/ZZZ_A, 1/Z_B, 1/c
  const fe zc = fe_mul(B.Z, c);
  const fe D = fe_mul(A.ZZZ, zc);
  if (fe_is_zero(D)) return SP_VERIFY_FALSE;  // degenerate (unreachable) guard
  const fe I = fe_inv(D);
  const fe izzz = fe_mul(I, zc);
  const fe Iz3 = fe_mul(I, A.ZZZ);
  const fe iz = fe_mul(Iz3, c);
  const fe ic = fe_mul(Iz3, B.Z);
  const fe xa = fe_mul(A.X, fe_sqr(fe_mul(A.ZZ, izzz)));
  const fe ya = fe_mul(A.Y, izzz);
  const fe iz2 = fe_sqr(iz);
  const fe xB = fe_mul(fe_mul(B.X, iz2), ic);
  const fe t = fe_mul(fe_mul(B.Y, fe_mul(iz2, iz)), fe_sqr(ic));
  const fe r_m = mont_of(r);
  const fe dx = fe_carry(fe_sub(xa, xB));
  bool ok;
  if (fe_is_zero(dx)) {
    // B == +-A.  B == A: the sum is 2A;  B == -A: infinity -> False (signature.py:254).
    const bool same = has_y ? fe_eq(ya, t) : true;  // x-only: one of the two signs gives 2A
    ok = same && fe_eq(double_x(xa, ya), r_m);
  } else {
    // E = (r + xa + xB) dx^2 - ya^2 - t^2 c
    const fe sum = fe_carry(fe_add(fe_add(r_m, xa), xB));
    const fe tt = fe_sqr(t);
    const fe E = fe_carry(fe_sub(fe_mul_sub_mul(sum, fe_sqr(dx), tt, c), fe_sqr(ya)));
    const fe yat = fe_mul(ya, t);
    if (has_y) {
      ok = fe_eq(E, fe_carry(fe_neg(fe_dbl(yat))));  // x(A + B) == r
    } else {
      const fe rhs2 = fe_mul(fe_carry(fe_dbl(fe_carry(fe_dbl(fe_sqr(yat))))), c);  // 4 ya^2 t^2 c
      ok = fe_eq(fe_sqr(E), rhs2);
    }
  }
  return ok ? SP_VERIFY_TRUE : SP_VERIFY_FALSE;
}

// Public-key stage: the curve model the multiples of Q live on.  Point key: the curve itself
// (c = a = 1), off-curve -> SP_VERIFY_ASSERT_CURVE (signature.py:241).  X-only key: with
// c = x^3 + x + beta the multiples of (x, sqrt c) are rational points of
// y'^2 = x'^3 + c^2 x' + beta c^3 (x' = c x, y' = c^2 b); base = (c x, c^2);
// c a non-residue -> SP_VERIFY_FALSE (InvalidPublicKeyError, signature.py:232-235).
__device__ __forceinline__ uint8_t key_model(const uint64_t* __restrict__ pqx, const uint64_t* __restrict__ pqy,
                                             size_t e, aff& base, fe& c, fe& a_coef) {
  const fe qx = mont_of(reduce_mod_p(ld_u256(pqx + 4 * e)));
  const fe rhs = fe_carry(fe_add(fe_add(fe_mul(fe_sqr(qx), qx), qx), mont_of(CURVE_BETA)));
  if (pqy != nullptr) {
    const fe qy = mont_of(reduce_mod_p(ld_u256(pqy + 4 * e)));
    if (!fe_eq(fe_sqr(qy), rhs)) return SP_VERIFY_ASSERT_CURVE;
    c = FE_ONE_M;
    a_coef = FE_ONE_M;
    base.x = qx;
    base.y = qy;
  } else {
    c = rhs;
    if (fe_is_zero(c) || !fe_is_qr(c)) return SP_VERIFY_FALSE;
    a_coef = fe_sqr(c);
    base.x = fe_mul(c, qx);
    base.y = a_coef;
  }
  return VERIFY_CONTINUE;
}

__global__ void __launch_bounds__(128)
ecdsa_verify_kernel(const uint64_t* __restrict__ pz, const uint64_t* __restrict__ pr,
                    const uint64_t* __restrict__ ps, const uint64_t* __restrict__ pqx,
                    const uint64_t* __restrict__ pqy, uint8_t* __restrict__ result, size_t n,
                    const aff_packed* __restrict__ gen, int wbits, int nwin, int32_t* __restrict__ tab) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  verify_scalars v;
  uint8_t code = verify_prepare(pz, pr, ps, e, v);
  if (code != VERIFY_CONTINUE) { result[e] = code; return; }
  aff base;
  fe c, a_coef;
  code = key_model(pqx, pqy, e, base, c, a_coef);
  if (code != VERIFY_CONTINUE) { result[e] = code; return; }
  if (v.z_zero) { result[e] = SP_VERIFY_FALSE; return; }
  const jac B = ladder_mul(v.u2, base, a_coef, tab, n, e);
  const xyzz A = gen_mul(v.u1, gen, wbits, nwin);
  result[e] = verify_finish(A, B, c, pqy != nullptr, v.r);
}

// (qx, qy) = d * G
__global__ void __launch_bounds__(128)
public_key_kernel(const uint64_t* __restrict__ pd, uint64_t* __restrict__ ox, uint64_t* __restrict__ oy,
                  uint8_t* __restrict__ status, size_t n, const aff_packed* __restrict__ gen, int wbits,
                  int nwin) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const u256 d = ld_u256(pd + 4 * e);
  if (u256_is_zero(d) || !u256_lt(d, U256_N)) {  // signature.py:105
    if (status) status[e] = SP_SIGN_BAD_INPUT;
    return;
  }
  const xyzz A = gen_mul(d, gen, wbits, nwin);
  const fe izzz = fe_inv(A.ZZZ);
  const fe x = fe_mul(A.X, fe_sqr(fe_mul(A.ZZ, izzz)));
  const fe y = fe_mul(A.Y, izzz);
  st_u256(ox + 4 * e, fe_pack(fe_from_mont(x)));
  if (oy) st_u256(oy + 4 * e, fe_pack(fe_from_mont(y)));
  if (status) status[e] = SP_SIGN_OK;
}

// One pass of the loop body of sign() (signature.py:146-173) with a caller-supplied k.
__global__ void __launch_bounds__(128)
ecdsa_sign_kernel(const uint64_t* __restrict__ pz, const uint64_t* __restrict__ pd,
                  const uint64_t* __restrict__ pk, uint64_t* __restrict__ orr, uint64_t* __restrict__ os,
                  uint8_t* __restrict__ status, size_t n, const aff_packed* __restrict__ gen, int wbits,
                  int nwin) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const u256 z = ld_u256(pz + 4 * e), d = ld_u256(pd + 4 * e), k = ld_u256(pk + 4 * e);
  if (!u256_lt(z, U256_2P251) || u256_is_zero(d) || !u256_lt(d, U256_N) || u256_is_zero(k) ||
      !u256_lt(k, U256_N)) {
    status[e] = SP_SIGN_BAD_INPUT;
    return;
  }
  const xyzz A = gen_mul(k, gen, wbits, nwin);
  const fe izzz = fe_inv(A.ZZZ);
  const fe x = fe_mul(A.X, fe_sqr(fe_mul(A.ZZ, izzz)));
  const u256 r = fe_pack(fe_from_mont(x));
  if (u256_is_zero(r) || !u256_lt(r, U256_2P251)) { status[e] = SP_SIGN_RETRY; return; }  // :158-161
  // t = z + r d mod N
  const fe k_m = montn_of(k);
  const fe one_c = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  cols acc;
  cols_zero(acc);
  cols_mac(acc, montn_of(r), montn_of(d));
  cols_mac(acc, montn_of(z), fn_to_mont(one_c));
  const fe t_m = fn_reduce(acc);  // Montgomery form of z + r d
  if (limbs_is_zero(fn_from_mont(t_m))) { status[e] = SP_SIGN_RETRY; return; }  // :163-165
  const fe I = fn_inv(fn_mul(k_m, t_m));
  const fe w_m = fn_mul(fn_sqr(k_m), I);  // k / t
  const u256 w = fe_pack(fn_from_mont(w_m));
  if (u256_is_zero(w) || !u256_lt(w, U256_2P251)) { status[e] = SP_SIGN_RETRY; return; }  // :167-170
  const fe s_m = fn_mul(fn_sqr(t_m), I);  // t / k = w^-1
  st_u256(orr + 4 * e, r);
  st_u256(os + 4 * e, fe_pack(fn_from_mont(s_m)));
  status[e] = SP_SIGN_OK;
}

}  // namespace sp

using namespace sp;

// Per-stream, like the Pedersen scratch: verifications in flight on different streams (or issued by
// different host threads) never share a window table.
static std::map<hipStream_t, sp::DeviceBuffer> g_verify_tab;
namespace sp {
void release_ecdsa_state() {
  for (auto& kv : g_verify_tab) kv.second.release();
  g_verify_tab.clear();
}
}
static inline unsigned nblocks(size_t n, unsigned tpb) { return (unsigned)((n + tpb - 1) / tpb); }

extern "C" {

int sp_ecdsa_verify_batch_dev(const uint64_t* z, const uint64_t* r, const uint64_t* s,
                              const uint64_t* qx, const uint64_t* qy, uint8_t* result, size_t n,
                              void* stream) {
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  Context& c = ctx();
  ctx_lock lk(c.mu);
  // per-signature table of the eight odd multiples of the key: 8 x 27 limbs, limb-major
  DeviceBuffer& tab = g_verify_tab[(hipStream_t)stream];
  SP_HIP(tab.reserve(n * 8 * 27 * sizeof(int32_t)));
  hipLaunchKernelGGL(ecdsa_verify_kernel, dim3(nblocks(n, 128)), dim3(128), 0, (hipStream_t)stream, z, r,
                     s, qx, qy, result, n, c.gen, c.wbits, c.nwin, (int32_t*)tab.ptr);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

// Host staging helper: copies `count` felt arrays (each n felts; null pointers stay null) to the
// device staging buffer and returns device pointers.
static int stage_in(const uint64_t* const* host, int count, size_t n, uint64_t** dev, size_t extra,
                    char** extra_ptr) {
  Context& c = ctx();
  const size_t fb = n * 32;
  SP_HIP(c.io2.reserve((size_t)count * fb + extra + 256));
  char* base = (char*)c.io2.ptr;
  for (int i = 0; i < count; ++i) {
    if (host[i]) {
      dev[i] = (uint64_t*)(base + (size_t)i * fb);
      SP_HIP(hipMemcpy(dev[i], host[i], fb, hipMemcpyHostToDevice));
    } else {
      dev[i] = nullptr;
    }
  }
  *extra_ptr = base + (size_t)count * fb;
  return SP_OK;
}

int sp_ecdsa_verify_batch(const uint64_t* z, const uint64_t* r, const uint64_t* s,
                          const uint64_t* qx, const uint64_t* qy, uint8_t* result, size_t n) {
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  ctx_lock lk(ctx().mu);
  const uint64_t* host[5] = {z, r, s, qx, qy};
  uint64_t* dev[5];
  char* extra;
  int rc = stage_in(host, 5, n, dev, n, &extra);
  if (rc != SP_OK) return rc;
  rc = sp_ecdsa_verify_batch_dev(dev[0], dev[1], dev[2], dev[3], dev[4], (uint8_t*)extra, n, 0);
  if (rc != SP_OK) return rc;
  SP_HIP(hipDeviceSynchronize());
  SP_HIP(hipMemcpy(result, extra, n, hipMemcpyDeviceToHost));
  return SP_OK;
}

int sp_ecdsa_sign_batch(const uint64_t* z, const uint64_t* d, const uint64_t* k, uint64_t* r,
                        uint64_t* s, uint8_t* status, size_t n) {
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  Context& c = ctx();
  ctx_lock lk(c.mu);
  const uint64_t* host[3] = {z, d, k};
  uint64_t* dev[3];
  char* extra;
  const size_t fb = n * 32;
  int rc = stage_in(host, 3, n, dev, 2 * fb + n, &extra);
  if (rc != SP_OK) return rc;
  uint64_t* dr = (uint64_t*)extra;
  uint64_t* ds = (uint64_t*)(extra + fb);
  uint8_t* dst = (uint8_t*)(extra + 2 * fb);
  SP_HIP(hipMemset(dr, 0, 2 * fb));
  hipLaunchKernelGGL(ecdsa_sign_kernel, dim3(nblocks(n, 128)), dim3(128), 0, 0, dev[0], dev[1], dev[2],
                     dr, ds, dst, n, c.gen, c.wbits, c.nwin);
  SP_HIP(hipGetLastError());
  SP_HIP(hipDeviceSynchronize());
  SP_HIP(hipMemcpy(r, dr, fb, hipMemcpyDeviceToHost));
  SP_HIP(hipMemcpy(s, ds, fb, hipMemcpyDeviceToHost));
  SP_HIP(hipMemcpy(status, dst, n, hipMemcpyDeviceToHost));
  return SP_OK;
}

int sp_public_key_batch(const uint64_t* d, uint64_t* qx, uint64_t* qy, uint8_t* status, size_t n) {
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  Context& c = ctx();
  ctx_lock lk(c.mu);
  const uint64_t* host[1] = {d};
  uint64_t* dev[1];
  char* extra;
  const size_t fb = n * 32;
  int rc = stage_in(host, 1, n, dev, 2 * fb + n, &extra);
  if (rc != SP_OK) return rc;
  uint64_t* dx = (uint64_t*)extra;
  uint64_t* dy = (uint64_t*)(extra + fb);
  uint8_t* dst = (uint8_t*)(extra + 2 * fb);
  SP_HIP(hipMemset(dx, 0, 2 * fb));
  hipLaunchKernelGGL(public_key_kernel, dim3(nblocks(n, 128)), dim3(128), 0, 0, dev[0], dx, dy, dst, n,
                     c.gen, c.wbits, c.nwin);
  SP_HIP(hipGetLastError());
  SP_HIP(hipDeviceSynchronize());
  SP_HIP(hipMemcpy(qx, dx, fb, hipMemcpyDeviceToHost));
  if (qy) SP_HIP(hipMemcpy(qy, dy, fb, hipMemcpyDeviceToHost));
  if (status) SP_HIP(hipMemcpy(status, dst, n, hipMemcpyDeviceToHost));
  return SP_OK;
}

}  // extern "C"

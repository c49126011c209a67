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
//     r in { x(A + B), x(A - B) } becomes one polynomial identity  E^2 == 4 ya^2 t^2 c  (evaluated with
//     the denominators cleared, verify_finish).  "c is a square" (InvalidPublicKeyError -> False,
//     signature.py:232-235) is implied by that identity - on the twist it has no solution, see key_model -
//     so only key registration still runs a Legendre test (to label the slot).
// Reachable reference failure modes and how they are reproduced here:
//   z == 0                      -> False   (mimic_ec_mult_air asserts 0 < m, signature.py:181)
//   z*G + r*Q == infinity       -> False   (ec_add x-collision, signature.py:254)
//   pre-assert violations       -> SP_VERIFY_ASSERT_* codes (signature.py:219,225-227,241)
// All other assertion sites of the reference ladders (partial sum meeting the doubled point) need
// a discrete-log relation between the shift point and G or Q (DESIGN.md "failure set").
#include <array>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <vector>

#include "context.hpp"
#include "curve_consts.hpp"
#include "masked_walk.hpp"
#include "rfc6979.hpp"

namespace sp {

// Threads per block of the verification kernels: 256 = one wave per SIMD of a CU.  With 128-thread blocks
// the dispatcher packed two blocks onto the same SIMD pair of half the CUs for a 2^16-signature batch
// (3.1 ms; 2^15: 1.8 ms, 2^17: 3.7 ms) - the batch of BASELINE.json configs[2] ran at half speed.
#ifndef SP_VERIFY_TPB
#define SP_VERIFY_TPB 256
#endif
constexpr int VERIFY_TPB = SP_VERIFY_TPB;
// Waves per SIMD the register allocator must leave room for in the two verification kernels.
// Left to itself (0) the allocator takes 256 VGPRs + 2 (ladder) / 11 (keyed) AGPRs - a handful of registers over the
// 256 that two waves per SIMD may use - and a SIMD holds ONE wave.  With amdgpu_waves_per_eu(2, 2) it stays within 256
// and spills 12 / 60 bytes per lane.  Measured (profiles/r06_verify_occupancy.txt, same box, alternating): nothing at
// 2^16 signatures (1024 waves = one per SIMD either way), + 17 % / + 26 % at 2^18 (ladder 5.35 -> 6.27 x 10^7 /s,
// keyed 2.77 -> 3.48 x 10^8 /s), + 22 % / + 29 % at 2^20.  2 is the default since round 6.
#ifndef SP_VERIFY_WAVES
#define SP_VERIFY_WAVES 2
#endif
// The same switch for the signers and the key derivation (183 - 212 VGPRs: two waves per SIMD by themselves; 3 would
// need <= 168).  A/B build only so far (profiles/r06_verify_occupancy.txt): 0 = the allocator's choice.
#ifndef SP_SIGN_WAVES
#define SP_SIGN_WAVES 0
#endif
#if SP_SIGN_WAVES > 0
#define SP_SIGN_OCCUPANCY __attribute__((amdgpu_waves_per_eu(SP_SIGN_WAVES, SP_SIGN_WAVES)))
#else
#define SP_SIGN_OCCUPANCY
#endif
#if SP_VERIFY_WAVES > 0
#define SP_VERIFY_OCCUPANCY __attribute__((amdgpu_waves_per_eu(SP_VERIFY_WAVES, SP_VERIFY_WAVES)))
#else
#define SP_VERIFY_OCCUPANCY
#endif

// k * EC_GEN from the window table; k < 2^252.  Infinity (only for k == 0 mod N) shows as ZZ == 0.
// wbits < 0: the masked walk above (the signers and the key derivation under STARKPERP_SIGN_MASKED=1).
__device__ __forceinline__ xyzz gen_mul(u256 k, const aff_packed* __restrict__ gen, int wbits, int nwin) {
  if (wbits < 0) return gen_mul_masked(k, gen, nwin);
  const size_t per = (size_t)1 << wbits;
  const uint32_t mask = (1u << wbits) - 1u;
  auto pop = [&](void) {
    const uint32_t v = k.w[0] & mask;
#pragma unroll
    for (int i = 0; i < 7; ++i) k.w[i] = (k.w[i] >> wbits) | (k.w[i + 1] << (32 - wbits));
    k.w[7] >>= wbits;
    return v;
  };
  // the entry of window i + 1 is requested before window i is added: a lone wave per SIMD has nothing else to
  // hide a random 64-byte gather (tables of tens of GiB: a TLB miss on most of them) behind
  const aff e0 = ld_aff(gen + pop());
  aff nxt = nwin > 1 ? ld_aff(gen + per + pop()) : aff{};
  xyzz acc;
  int first = 1;
  if (nwin > 1) {  // windows 0 and 1 are both affine: mmadd (4M + 2S) instead of a mixed addition (8M + 2S)
    const aff q1 = nxt;
    if (2 < nwin) nxt = ld_aff(gen + (size_t)2 * per + pop());
    acc = xyzz_mmadd(e0, q1);
    first = 2;
  } else {
    acc = xyzz_from_aff(e0);
  }
  for (int i = first; i < nwin; ++i) {
    const aff q = nxt;
    if (i + 1 < nwin) nxt = ld_aff(gen + (size_t)(i + 1) * per + pop());
    acc = xyzz_madd(acc, q);
  }
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
__device__ __forceinline__ fe double_x(const fe& xa, const fe& ya) {
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
  const fe w_m = fn_inv_var(montn_of(s));  // s is public: variable-time divsteps
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
// limb-major (entry * 36 + limb) * n + e, and are made AFFINE before the ladder starts - one shared inversion
// of Z_1 ... Z_7 (Montgomery's trick, ~100 multiplication-equivalents) buys 63 mixed additions (7M + 4S)
// instead of 63 general ones (11M + 5S): -6 % of the ladder.  A zero among the Z's (a base point of order
// <= 15: impossible on the curve, prime order, and on its twist, no factor below 2 * 10^5) would poison the
// shared inversion; the product is tested and such a ladder returns the absorbing state Z = 0 (-> False).
__device__ __forceinline__ jac ladder_mul(const u256& u2, const aff& base, const fe& a_coef,
                                          int32_t* __restrict__ tab, size_t n, size_t e) {
  bool flip;
  u256 E = recode_odd(u2, flip);
  // planes of an entry: 0..8 X (then affine x), 9..17 Y (then affine y), 18..26 Z, 27..35 prefix product
  auto tab_at = [&](int entry, int limb) -> int32_t* { return tab + ((size_t)(entry * 36 + limb) * n + e); };
  auto tab_store = [&](int entry, int plane, const fe& v) {
#pragma unroll
    for (int l = 0; l < NL; ++l) *tab_at(entry, 9 * plane + l) = v.l[l];
  };
  auto tab_load = [&](int entry, int plane) {
    fe v;
#pragma unroll
    for (int l = 0; l < NL; ++l) v.l[l] = *tab_at(entry, 9 * plane + l);
    return v;
  };
  jac B;
  B.X = base.x; B.Y = base.y; B.Z = FE_ONE_M;
  {
    tab_store(0, 0, base.x);
    tab_store(0, 1, base.y);
    const jac twoQ = jac_dbl(B, a_coef);
    jac odd = jac_madd(twoQ, base);  // 3Q
    fe run = FE_ONE_M;
#pragma unroll 1
    for (int j = 1; j < 8; ++j) {
      if (j > 1) odd = jac_add(odd, twoQ);
      tab_store(j, 0, odd.X);
      tab_store(j, 1, odd.Y);
      tab_store(j, 2, odd.Z);
      tab_store(j, 3, run);
      run = fe_mul(run, odd.Z);
    }
    if (fe_is_zero(run)) {  // see above: unreachable for valid inputs, absorbing state otherwise
      B.Z = FE_ZERO;
      return B;
    }
    fe inv = fe_inv(run);
#pragma unroll 1
    for (int j = 7; j >= 1; --j) {
      const fe zinv = fe_mul(inv, tab_load(j, 3));
      inv = fe_mul(inv, tab_load(j, 2));
      const fe zi2 = fe_sqr(zinv);
      tab_store(j, 0, fe_mul(tab_load(j, 0), zi2));
      tab_store(j, 1, fe_mul(tab_load(j, 1), fe_mul(zi2, zinv)));
    }
  }
  // top window is e = 8  ->  digit +1: start from Q itself, then 63 windows
  {
#pragma unroll
    for (int i = 7; i > 0; --i) E.w[i] = (E.w[i] << 4) | (E.w[i - 1] >> 28);
    E.w[0] <<= 4;
  }
  // The loop body is kept to ONE doubling + ONE addition of code (~35 KB): with the four doublings unrolled the
  // body outgrows the 64 KB instruction cache and a lone wave per SIMD waits on instruction fetch.
#pragma unroll 1
  for (int wi = 0; wi < 63; ++wi) {
    const uint32_t ew = E.w[7] >> 28;
#pragma unroll
    for (int i = 7; i > 0; --i) E.w[i] = (E.w[i] << 4) | (E.w[i - 1] >> 28);
    E.w[0] <<= 4;
    const int d = 2 * (int)ew - 15;
    const int mag = (d < 0 ? -d : d) >> 1;  // table index of |d| = 2 mag + 1
    aff T;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      T.x.l[l] = *tab_at(mag, l);
      const int32_t y = *tab_at(mag, 9 + l);
      T.y.l[l] = d < 0 ? -y : y;
    }
    // four doublings in modified Jacobian form (W = a Z^4 carried along: 16M + 18S instead of 8M + 32S)
    mjac Bm = mjac_from(B, a_coef);
#pragma unroll 1
    for (int k = 0; k < 4; ++k) mjac_dbl(Bm, k < 3);
    B.X = Bm.X; B.Y = Bm.Y; B.Z = Bm.Z;
    B = jac_madd(B, T);
  }
  if (flip) B.Y = fe_neg(B.Y);
  return B;
}

// Acceptance test for A = u1 G (XYZZ, real curve) and B = u2 Q (Jacobian; on the real curve for a
// point key, c = 1, or on the c-twisted model for an x-only key): r == x(A + B), resp.
// r in { x(A + B), x(A - B) } as one polynomial identity.
//
// Affine form of the test (xa, ya the coordinates of A; xB = x(B), yB^2 = t^2 c, dx = xa - xB):
//   E = (r + xa + xB) dx^2 - ya^2 - t^2 c;   x(A + B) == r  <=>  E == -2 ya yB;   x(A +- B) == r  <=>  E^2 == 4 ya^2 t^2 c.
// verify_finish_affine evaluates it after one shared inversion (rare path: B == +-A needs the doubling);
// verify_finish clears the denominators instead - a = ZZ_A, b = c Z_B^2:
//   xa = X_A / a,  ya^2 = Y_A^2 / a^3 (ZZZ_A^2 = a^3),  xB = X_B / b,  t^2 c = Y_B^2 / b^3,  dx = Dn / (a b),
//   En = a^3 b^3 E = (r a b + X_A b + X_B a) Dn^2 - Y_A^2 b^3 - Y_B^2 a^3
//   x-only:  En^2 == 4 (Y_A^2 b^3)(Y_B^2 a^3);      point key (c = 1, b^3 = Z_B^6):  En == -2 (Y_A ZZZ_A)(Y_B Z_B b)
// - 12 multiplications + 7 squarings and no inversion (the inversion was 10 k of the 131 k instructions
// of a keyed verification).
__device__ __forceinline__ uint8_t verify_finish_affine(const xyzz& A, const jac& B, const fe& c, bool has_y,
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

__device__ __forceinline__ uint8_t verify_finish(const xyzz& A, const jac& B, const fe& c, bool has_y,
                                                 const u256& r) {
  const fe zb2 = fe_sqr(B.Z);
  const fe b = fe_mul(zb2, c);
  const fe& a = A.ZZ;
  const fe xan = fe_mul(A.X, b), xbn = fe_mul(B.X, a);
  const fe Dn = fe_carry(fe_sub(xan, xbn));
  // infinity on either side (a b == 0; unreachable for A, the absorbing state of the ladder / comb for B) and
  // B == +-A (Dn == 0) go through the affine form, which owns those cases
  const fe ab = fe_mul(a, b);
  if (fe_is_zero(ab) || fe_is_zero(Dn)) return verify_finish_affine(A, B, c, has_y, r);
  const fe sum = fe_carry(fe_add(fe_add(fe_mul(mont_of(r), ab), xan), xbn));
  const fe a3 = fe_sqr(A.ZZZ);
  const fe b3 = fe_mul(fe_sqr(b), b);
  const fe P = fe_mul(fe_sqr(A.Y), b3), Q = fe_mul(fe_sqr(B.Y), a3);
  const fe En = fe_carry(fe_sub(fe_sub(fe_mul(sum, fe_sqr(Dn)), P), Q));
  bool ok;
  if (has_y) {
    const fe rhs = fe_mul(fe_mul(A.Y, A.ZZZ), fe_mul(fe_mul(B.Y, B.Z), b));
    ok = fe_eq(En, fe_carry(fe_neg(fe_dbl(rhs))));
  } else {
    ok = fe_eq(fe_sqr(En), fe_carry(fe_dbl(fe_carry(fe_dbl(fe_mul(P, Q))))));
  }
  return ok ? SP_VERIFY_TRUE : SP_VERIFY_FALSE;
}

// Public-key stage: the curve model the multiples of Q live on.  Point key: the curve itself
// (c = a = 1), off-curve -> SP_VERIFY_ASSERT_CURVE (signature.py:241).  X-only key: with
// c = x^3 + x + beta the multiples of (x, sqrt c) are rational points of
// y'^2 = x'^3 + c^2 x' + beta c^3 (x' = c x, y' = c^2 b); base = (c x, c^2);
// c a non-residue -> SP_VERIFY_FALSE (InvalidPublicKeyError, signature.py:232-235).
//
// CHECK_QR = false (the ladder kernel): the Legendre test (~310 field multiplications) is left to
// verify_finish, whose x-only acceptance test E^2 == 4 ya^2 t^2 c can only hold for a residue c:
//   * ya != 0 and A != infinity: A = u1 G with u1 = z w != 0 mod N (z, w in [1, N), N prime) on a curve of
//     odd prime order - no point with y = 0;
//   * the model of a non-residue c is the quadratic twist, of order 2 p + 2 - N, odd as well: t = 0 only at
//     infinity, and every exceptional case of the incomplete ladder formulas (a base point of small order
//     on the twist can reach them) ends with Z = 0, which is absorbing (Z3 = Z1 Z2 H, resp. 2 Y1 Z1) and
//     answers False at the D == 0 guard;
//   * with ya t != 0 the equality gives c = (E / (2 ya t))^2, a residue.
// So a non-residue c always ends in SP_VERIFY_FALSE, which is the reference's answer for it
// (InvalidPublicKeyError -> False, signature.py:232-235); dx == 0 cannot occur either (an x that is on the
// curve and on its twist has c = 0, excluded above).
template <bool CHECK_QR = true>
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
    if (fe_is_zero(c)) return SP_VERIFY_FALSE;
    if (CHECK_QR && !fe_is_qr(c)) return SP_VERIFY_FALSE;
    a_coef = fe_sqr(c);
    base.x = fe_mul(c, qx);
    base.y = a_coef;
  }
  return VERIFY_CONTINUE;
}

__global__ void __launch_bounds__(VERIFY_TPB) SP_VERIFY_OCCUPANCY
ecdsa_verify_kernel(const uint64_t* __restrict__ pz, const uint64_t* __restrict__ pr,
                    const uint64_t* __restrict__ ps, const uint64_t* __restrict__ pqx,
                    const uint64_t* __restrict__ pqy, uint8_t* __restrict__ result, size_t n,
                    const aff_packed* __restrict__ gen, int wbits, int nwin, int32_t* __restrict__ tab) {
  // Lanes past the end redo the last item instead of idling (identical reads, identical writes): a
  // wave with <= 8 active lanes runs VALU code ~3.7x slower on this chip (tools/ubench/inv_lanes.hip),
  // which is what a scalar call (n = 1) would otherwise get.
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) e = n - 1;
  verify_scalars v;
  uint8_t code = verify_prepare(pz, pr, ps, e, v);
  if (code != VERIFY_CONTINUE) { result[e] = code; return; }
  aff base;
  fe c, a_coef;
  code = key_model<false>(pqx, pqy, e, base, c, a_coef);
  if (code != VERIFY_CONTINUE) { result[e] = code; return; }
  if (v.z_zero) { result[e] = SP_VERIFY_FALSE; return; }
  // the table slot is the LANE's (lanes past the end redo the last item: with a shared slot a wave that has already
  // made its entries affine would race one that still reads the projective ones)
  const jac B = ladder_mul(v.u2, base, a_coef, tab, (size_t)gridDim.x * blockDim.x,
                           (size_t)blockIdx.x * blockDim.x + threadIdx.x);
  const xyzz A = gen_mul(v.u1, gen, wbits, nwin);
  result[e] = verify_finish(A, B, c, pqy != nullptr, v.r);
}

// =================================================================================================
// Key tables: verification against a public key that has been seen before.
//
// The ladder above spends 252 doublings + 63 additions per signature on u2 * Q because Q is new to
// it every time.  Exchange traffic is not like that: the same accounts sign again and again
// (BASELINE.json configs[2]: 4096 orders from 1024 keys).  With 288 GB of HBM a signed comb table per
// key is cheap - COMB_TABLES x 128 affine points, 32 KiB - and turns u2 * Q into 7 doublings + 31 mixed additions
// (one table: 31 + 32):
//   rows     Q_i = 2^(32 i) Q, i = 0..7          (the 256 signed bits of the recoding, 8 rows x 32 columns)
//   entries  T[v] = Q_7 + sum_{i<7} (2 v_i - 1) Q_i,  v = 0..127   (row 7 carries the column's sign)
//   column c of E = (k-1)/2 + 2^255: bit c of word i is row i; sign = row 7, index = rows 0..6
//   (complemented when the sign is negative);  acc = 2 acc +- T[index], from column 31 down to 0.
// Tables live on the same curve model as the ladder's base point (the curve itself for a point key,
// the c-twisted model for an x-only key), so verify_finish is shared.  Unlike the window ladder, a
// chosen u2 CAN make a comb addition meet its own operand (k = N - 2|t| for a table multiple t), so
// the column addition detects Z3 == 0 and takes the doubling / infinity branch explicitly.
constexpr int COMB_ENTRIES = 128;
constexpr int COMB_ROW_POINTS = 15;  // Q_0, 2Q_0, Q_1, 2Q_1, ..., Q_6, 2Q_6, Q_7
// COMB_TABLES tables per key (round 2): table t is the same 128-entry signed comb built on 2^(t * COMB_COLS) Q and
// serves the columns t * COMB_COLS .. (t + 1) * COMB_COLS - 1 of the recoded scalar, so u2 * Q costs
// COMB_COLS - 1 = 7 doublings + 31 mixed additions instead of 31 + 32 - HBM is plentiful: 32 KiB per key.
#ifndef SP_COMB_TABLES
#define SP_COMB_TABLES 4
#endif
constexpr int COMB_TABLES = SP_COMB_TABLES;
constexpr int COMB_COLS = 32 / COMB_TABLES;
constexpr int KEY_ENTRIES = COMB_TABLES * COMB_ENTRIES;
constexpr int KEY_ROW_POINTS = COMB_TABLES * COMB_ROW_POINTS;
// A slot handle is (generation << 24) | index: sp_ecdsa_key_cache_reset starts a new generation, so a
// handle from before the reset can never name another key's table - the kernel answers
// SP_VERIFY_STALE_SLOT for it.  Generations run 1..255 and wrap (a handle would have to survive 255
// resets to alias).
constexpr uint32_t SLOT_INDEX_BITS = 24, SLOT_INDEX_MASK = (1u << SLOT_INDEX_BITS) - 1u;
constexpr uint8_t KEY_EMPTY = 0, KEY_XONLY = 1, KEY_POINT = 2, KEY_INVALID_X = 3, KEY_OFF_CURVE = 4;

__device__ __forceinline__ void st_aff(aff_packed* dst, const fe& x_m, const fe& y_m) {
  uint4* q = reinterpret_cast<uint4*>(dst);
  const u256 x = fe_pack(fe_canon(x_m)), y = fe_pack(fe_canon(y_m));
  q[0] = make_uint4(x.w[0], x.w[1], x.w[2], x.w[3]);
  q[1] = make_uint4(x.w[4], x.w[5], x.w[6], x.w[7]);
  q[2] = make_uint4(y.w[0], y.w[1], y.w[2], y.w[3]);
  q[3] = make_uint4(y.w[4], y.w[5], y.w[6], y.w[7]);
}

// Stage 1, one thread per NEW key: validity, curve model, the 15 row points (224 doublings, one
// shared inversion).  `work` holds 15 x (X, Y, Z, prefix) limb planes per key, key-minor.
__global__ void __launch_bounds__(64)
key_rows_kernel(const uint64_t* __restrict__ pqx, const uint64_t* __restrict__ pqy,
                const uint8_t* __restrict__ has_y, const uint32_t* __restrict__ slot_of, size_t n_new,
                uint64_t* __restrict__ key_c, uint8_t* __restrict__ key_flag,
                aff_packed* __restrict__ rows, int32_t* __restrict__ work) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_new) e = n_new - 1;  // redundant copy of the last key, see ecdsa_verify_kernel
  const uint32_t slot = slot_of[e];
  aff base;
  fe c, a_coef;
  const bool point_key = has_y[e] != 0;
  const uint8_t code = key_model(pqx, point_key ? pqy : nullptr, e, base, c, a_coef);
  if (code != VERIFY_CONTINUE) {
    key_flag[slot] = point_key ? KEY_OFF_CURVE : KEY_INVALID_X;
    return;
  }
  st_u256(key_c + 4 * (size_t)slot, fe_pack(fe_canon(c)));
  auto plane = [&](int point, int limb) -> int32_t* { return work + ((size_t)(point * 36 + limb) * n_new + e); };
  mjac P;  // modified Jacobian (curve.hpp): the chain is nothing but doublings, 4M + 4S each with W = a Z^4 carried along
  P.X = base.x; P.Y = base.y; P.Z = FE_ONE_M; P.W = a_coef;
  fe run = FE_ONE_M;
  auto emit = [&](int point) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      *plane(point, l) = P.X.l[l];
      *plane(point, 9 + l) = P.Y.l[l];
      *plane(point, 18 + l) = P.Z.l[l];
      *plane(point, 27 + l) = run.l[l];
    }
    run = fe_mul(run, P.Z);
  };
  // one doubling chain 2^d base, d = 0 .. 224 + (COMB_TABLES - 1) COMB_COLS: table t, row i needs
  // Q_{t,i} = 2^(32 i + t COMB_COLS) base (point 2 i of the table) and, for i < 7, its double (point 2 i + 1)
  const int last = 224 + (COMB_TABLES - 1) * COMB_COLS;
  for (int d = 0; d <= last; ++d) {
    const int i = d >> 5, off = d & 31;
    if (off % COMB_COLS == 0 && off / COMB_COLS < COMB_TABLES) emit((off / COMB_COLS) * COMB_ROW_POINTS + 2 * i);
    if (i < 7 && off % COMB_COLS == 1 && (off - 1) / COMB_COLS < COMB_TABLES)
      emit(((off - 1) / COMB_COLS) * COMB_ROW_POINTS + 2 * i + 1);
    if (d < last) mjac_dbl(P, d + 1 < last);
  }
  fe inv = fe_inv(run);
  // undo the prefix products in reverse order of emission
  for (int d = last; d >= 0; --d) {
    const int i = d >> 5, off = d & 31;
    int pts[2], np = 0;
    if (off % COMB_COLS == 0 && off / COMB_COLS < COMB_TABLES) pts[np++] = (off / COMB_COLS) * COMB_ROW_POINTS + 2 * i;
    if (i < 7 && off % COMB_COLS == 1 && (off - 1) / COMB_COLS < COMB_TABLES)
      pts[np++] = ((off - 1) / COMB_COLS) * COMB_ROW_POINTS + 2 * i + 1;
    for (int k = np - 1; k >= 0; --k) {
      const int point = pts[k];
      fe X, Y, Z, pre;
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        X.l[l] = *plane(point, l);
        Y.l[l] = *plane(point, 9 + l);
        Z.l[l] = *plane(point, 18 + l);
        pre.l[l] = *plane(point, 27 + l);
      }
      const fe zinv = fe_mul(inv, pre);
      inv = fe_mul(inv, Z);
      const fe zi2 = fe_sqr(zinv);
      st_aff(rows + e * KEY_ROW_POINTS + point, fe_mul(X, zi2), fe_mul(Y, fe_mul(zi2, zinv)));
    }
  }
  key_flag[slot] = point_key ? KEY_POINT : KEY_XONLY;
}

// Stage 2, four threads per new key: thread t owns the 32 entries whose rows 5, 6 have the signs
// given by t, walks rows 0..4 in Gray-code order (one mixed addition of +-2 Q_i per entry), then
// converts its 32 projective entries to affine with one shared inversion.
__global__ void __launch_bounds__(64)
key_table_kernel(const aff_packed* __restrict__ rows, const uint32_t* __restrict__ slot_of, size_t n_new,
                 const uint8_t* __restrict__ key_flag, aff_packed* __restrict__ key_tab,
                 int32_t* __restrict__ work) {
  size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int t = (int)(gt & 3);
  const size_t lanes = 4 * COMB_TABLES * n_new;
  if (gt >= lanes) gt = (lanes - 4) | (size_t)t;  // redundant copy, see ecdsa_verify_kernel
  const int tb = (int)((gt >> 2) % COMB_TABLES);  // which of the key's tables
  const size_t e = (gt >> 2) / COMB_TABLES;
  const uint32_t slot = slot_of[e];
  const uint8_t flag = key_flag[slot];
  if (flag != KEY_XONLY && flag != KEY_POINT) return;
  auto plane = [&](int entry, int limb) -> int32_t* { return work + ((size_t)(entry * 45 + limb) * lanes + gt); };
  auto row_point = [&](int point, bool negative) {
    aff q = ld_aff(rows + e * KEY_ROW_POINTS + tb * COMB_ROW_POINTS + point);
    if (negative) q.y = fe_neg(q.y);
    return q;
  };
  // entry (t << 5) | 0: Q_7 +- Q_6 +- Q_5 - Q_4 - Q_3 - Q_2 - Q_1 - Q_0
  xyzz acc = xyzz_mmadd(row_point(14, false), row_point(12, (t & 2) == 0));
  acc = xyzz_madd(acc, row_point(10, (t & 1) == 0));
  for (int i = 4; i >= 0; --i) acc = xyzz_madd(acc, row_point(2 * i, true));
  fe run = FE_ONE_M;
  uint32_t gray = 0;
  for (int g = 0; g < 32; ++g) {
    if (g > 0) {
      const int bit = __ffs(g) - 1;  // the row whose sign flips between Gray codes g-1 and g
      gray ^= 1u << bit;
      acc = xyzz_madd(acc, row_point(2 * bit + 1, ((gray >> bit) & 1u) == 0));  // 0 -> 1 adds +2 Q_i
    }
    const int entry = (int)gray;  // position inside this thread's block of 32
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      *plane(entry, l) = acc.X.l[l];
      *plane(entry, 9 + l) = acc.Y.l[l];
      *plane(entry, 18 + l) = acc.ZZ.l[l];
      *plane(entry, 27 + l) = acc.ZZZ.l[l];
      *plane(g, 36 + l) = run.l[l];  // prefix products are indexed by visiting order
    }
    run = fe_mul(run, acc.ZZZ);
  }
  fe inv = fe_inv(run);
  gray = 0;
  for (int g = 1; g < 32; ++g) gray ^= 1u << (__ffs(g) - 1);  // Gray code of 31
  for (int g = 31; g >= 0; --g) {
    const int entry = (int)gray;
    fe X, Y, ZZ, ZZZ, pre;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      X.l[l] = *plane(entry, l);
      Y.l[l] = *plane(entry, 9 + l);
      ZZ.l[l] = *plane(entry, 18 + l);
      ZZZ.l[l] = *plane(entry, 27 + l);
      pre.l[l] = *plane(g, 36 + l);
    }
    const fe izzz = fe_mul(inv, pre);
    inv = fe_mul(inv, ZZZ);
    st_aff(key_tab + (size_t)slot * KEY_ENTRIES + tb * COMB_ENTRIES + ((t << 5) | entry),
           fe_mul(X, fe_sqr(fe_mul(ZZ, izzz))), fe_mul(Y, izzz));
    if (g > 0) gray ^= 1u << (__ffs(g) - 1);
  }
}

// acc + q with the exceptional cases resolved (acc == infinity, acc == q, acc == -q).
__device__ __forceinline__ jac jac_madd_complete(const jac& p, const aff& q, const fe& a_coef) {
  jac r = jac_madd(p, q);
  if (fe_is_zero(r.Z)) {  // Z3 = 2 Z1 H: p is infinity or the x-coordinates agree
    jac qj;
    qj.X = q.x; qj.Y = q.y; qj.Z = FE_ONE_M;
    if (fe_is_zero(p.Z)) {
      r = qj;
    } else {
      const fe z2 = fe_sqr(p.Z);
      if (fe_eq(fe_mul(fe_mul(q.y, p.Z), z2), p.Y)) {
        r = jac_dbl(qj, a_coef);
      } else {
        r.X = FE_ONE_M; r.Y = FE_ONE_M; r.Z = FE_ZERO;
      }
    }
  }
  return r;
}

// COMPLETE = false: plain mixed additions; an exceptional column (accumulator == +-entry, possible only
// for a chosen u2 = N - 2|t| of a table multiple t) drives Z to 0 and Z stays 0 to the end, where the
// caller reruns the scalar with COMPLETE = true (the zero test costs a reduction + canonical form per
// column: 8 k of the 150 k instructions of a keyed verification when done on every column).
template <bool COMPLETE>
__device__ __forceinline__ jac comb_mul_impl(const u256& E, bool flip, const aff_packed* __restrict__ tab, const fe& a_coef) {
  auto column = [&](int col, bool& negative) -> const aff_packed* {
    uint32_t idx = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) idx |= ((E.w[i] >> col) & 1u) << i;
    negative = ((E.w[7] >> col) & 1u) == 0;
    if (negative) idx ^= 127u;
    return tab + idx;
  };
  // column c = tb * COMB_COLS + j lives in table tb: sum_j 2^j (sum_tb T_tb[column]) - Horner over j.  Steps are
  // numbered k = 0 .. 31 (j = COMB_COLS - 1 - k / COMB_TABLES, tb = COMB_TABLES - 1 - k % COMB_TABLES); the entry of
  // step k + 1 is requested before step k is added (a lone wave has nothing else to hide the gather behind).
  auto step_entry = [&](int k, bool& negative) -> const aff_packed* {
    const int j = COMB_COLS - 1 - k / COMB_TABLES, tb = COMB_TABLES - 1 - k % COMB_TABLES;
    return column(tb * COMB_COLS + j, negative) + tb * COMB_ENTRIES;
  };
  bool neg;
  aff q = ld_aff(step_entry(0, neg));  // column 31: bit 255 of E is always set, this entry is +T
  jac B;
  B.X = q.x; B.Y = q.y; B.Z = FE_ONE_M;
  aff nxt = ld_aff(step_entry(1, neg));
#pragma unroll 1
  for (int k = 1; k < 32; ++k) {
    q = nxt;
    if (neg) q.y = fe_neg(q.y);
    if (k + 1 < 32) nxt = ld_aff(step_entry(k + 1, neg));
    if (k % COMB_TABLES == 0) B = jac_dbl(B, a_coef);
    B = COMPLETE ? jac_madd_complete(B, q, a_coef) : jac_madd(B, q);
  }
  if (flip) B.Y = fe_neg(B.Y);
  return B;
}
__device__ __forceinline__ jac comb_mul(const u256& u2, const aff_packed* __restrict__ tab, const fe& a_coef) {
  bool flip;
  const u256 E = recode_odd(u2, flip);
  jac B = comb_mul_impl<false>(E, flip, tab, a_coef);
  if (__any(fe_is_zero(B.Z))) {  // wave-uniform branch: some lane met an exceptional column (or infinity)
    const jac C = comb_mul_impl<true>(E, flip, tab, a_coef);
    B = C;
  }
  return B;
}

__global__ void __launch_bounds__(VERIFY_TPB) SP_VERIFY_OCCUPANCY
ecdsa_verify_keyed_kernel(const uint64_t* __restrict__ pz, const uint64_t* __restrict__ pr,
                          const uint64_t* __restrict__ ps, const uint32_t* __restrict__ slots,
                          uint8_t* __restrict__ result, size_t n, const aff_packed* __restrict__ gen,
                          int wbits, int nwin, const aff_packed* __restrict__ key_tab,
                          const uint64_t* __restrict__ key_c, const uint8_t* __restrict__ key_flag,
                          uint32_t n_slots, uint32_t generation) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) e = n - 1;  // redundant copy of the last item, see ecdsa_verify_kernel
  verify_scalars v;
  const uint8_t code = verify_prepare(pz, pr, ps, e, v);
  if (code != VERIFY_CONTINUE) { result[e] = code; return; }
  const uint32_t handle = slots[e], slot = handle & SLOT_INDEX_MASK;
  if ((handle >> SLOT_INDEX_BITS) != generation || slot >= n_slots) {  // from before a cache reset, or never handed out
    result[e] = SP_VERIFY_STALE_SLOT;
    return;
  }
  const uint8_t flag = key_flag[slot];
  if (flag == KEY_OFF_CURVE) { result[e] = SP_VERIFY_ASSERT_CURVE; return; }  // signature.py:241
  if (flag != KEY_XONLY && flag != KEY_POINT) { result[e] = SP_VERIFY_FALSE; return; }  // :232-235
  if (v.z_zero) { result[e] = SP_VERIFY_FALSE; return; }
  const bool has_y = flag == KEY_POINT;
  fe c = FE_ONE_M, a_coef = FE_ONE_M;
  if (!has_y) {
    c = fe_unpack(ld_u256(key_c + 4 * (size_t)slot));
    a_coef = fe_sqr(c);
  }
  const jac B = comb_mul(v.u2, key_tab + (size_t)slot * KEY_ENTRIES, a_coef);
  const xyzz A = gen_mul(v.u1, gen, wbits, nwin);
  result[e] = verify_finish(A, B, c, has_y, v.r);
}

// (qx, qy) = d * G
__global__ void __launch_bounds__(128) SP_SIGN_OCCUPANCY
public_key_kernel(const uint64_t* __restrict__ pd, uint64_t* __restrict__ ox, uint64_t* __restrict__ oy,
                  uint8_t* __restrict__ status, size_t n, const aff_packed* __restrict__ gen, int wbits,
                  int nwin) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) e = n - 1;  // redundant copy of the last item, see ecdsa_verify_kernel
  const u256 d = ld_u256(pd + 4 * e);
  if (u256_is_zero(d) || !u256_lt(d, U256_N)) {  // signature.py:105
    if (status) status[e] = SP_SIGN_BAD_INPUT;
    return;
  }
  const xyzz A = gen_mul(d, gen, wbits, nwin);
  const fe izzz = fe_inv_gcd(A.ZZZ);  // fixed-length: ZZZ depends on the private key
  const fe x = fe_mul(A.X, fe_sqr(fe_mul(A.ZZ, izzz)));
  const fe y = fe_mul(A.Y, izzz);
  st_u256(ox + 4 * e, fe_pack(fe_from_mont(x)));
  if (oy) st_u256(oy + 4 * e, fe_pack(fe_from_mont(y)));
  if (status) status[e] = SP_SIGN_OK;
}

// One pass of the loop body of sign() (signature.py:146-173) for a given nonce k.
__device__ __forceinline__ uint8_t sign_attempt(const u256& z, const u256& d, const u256& k,
                                                const aff_packed* __restrict__ gen, int wbits, int nwin,
                                                u256& r_out, u256& s_out) {
  if (!u256_lt(z, U256_2P251) || u256_is_zero(d) || !u256_lt(d, U256_N) || u256_is_zero(k) ||
      !u256_lt(k, U256_N))
    return SP_SIGN_BAD_INPUT;
  const xyzz A = gen_mul(k, gen, wbits, nwin);
  const fe izzz = fe_inv_gcd(A.ZZZ);  // fixed-length: ZZZ depends on the nonce
  const fe x = fe_mul(A.X, fe_sqr(fe_mul(A.ZZ, izzz)));
  const u256 r = fe_pack(fe_from_mont(x));
  if (u256_is_zero(r) || !u256_lt(r, U256_2P251)) return SP_SIGN_RETRY;  // :158-161
  // t = z + r d mod N
  const fe k_m = montn_of(k);
  const fe one_c = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
  cols acc;
  cols_zero(acc);
  cols_mac(acc, montn_of(r), montn_of(d));
  cols_mac(acc, montn_of(z), fn_to_mont(one_c));
  const fe t_m = fn_reduce(acc);  // Montgomery form of z + r d
  if (limbs_is_zero(fn_from_mont(t_m))) return SP_SIGN_RETRY;  // :163-165
  const fe I = fn_inv(fn_mul(k_m, t_m));
  const fe w_m = fn_mul(fn_sqr(k_m), I);  // k / t
  const u256 w = fe_pack(fn_from_mont(w_m));
  if (u256_is_zero(w) || !u256_lt(w, U256_2P251)) return SP_SIGN_RETRY;  // :167-170
  const fe s_m = fn_mul(fn_sqr(t_m), I);  // t / k = w^-1
  r_out = r;
  s_out = fe_pack(fn_from_mont(s_m));
  return SP_SIGN_OK;
}

__global__ void __launch_bounds__(128) SP_SIGN_OCCUPANCY
ecdsa_sign_kernel(const uint64_t* __restrict__ pz, const uint64_t* __restrict__ pd,
                  const uint64_t* __restrict__ pk, uint64_t* __restrict__ orr, uint64_t* __restrict__ os,
                  uint8_t* __restrict__ status, size_t n, const aff_packed* __restrict__ gen, int wbits,
                  int nwin) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) e = n - 1;  // redundant copy of the last item, see ecdsa_verify_kernel
  u256 r, s;
  const uint8_t st = sign_attempt(ld_u256(pz + 4 * e), ld_u256(pd + 4 * e), ld_u256(pk + 4 * e), gen, wbits,
                                  nwin, r, s);
  if (st == SP_SIGN_OK) {
    st_u256(orr + 4 * e, r);
    st_u256(os + 4 * e, s);
  }
  status[e] = st;
}

// The whole of sign() (signature.py:137-173): RFC 6979 nonce on the device (rfc6979.hpp), the
// attempt, and - should the nonce be rejected, which takes a 2^-55 event - the reference's retry
// with the next seed (None -> 1 -> 2 ...; a seed of 0 and "no seed" give the same entropy and the
// same successor, so 0 stands for None).  After 8 rejected nonces the item is left to the caller.
__global__ void __launch_bounds__(128) SP_SIGN_OCCUPANCY
ecdsa_sign_rfc6979_kernel(const uint64_t* __restrict__ pz, const uint64_t* __restrict__ pd,
                          const uint64_t* __restrict__ pseed, uint64_t* __restrict__ orr,
                          uint64_t* __restrict__ os, uint8_t* __restrict__ status, size_t n,
                          const aff_packed* __restrict__ gen, int wbits, int nwin) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) e = n - 1;  // redundant copy of the last item, see ecdsa_verify_kernel
  const u256 z = ld_u256(pz + 4 * e), d = ld_u256(pd + 4 * e);
  uint64_t seed = pseed ? pseed[e] : 0;
  uint8_t st = SP_SIGN_BAD_INPUT;
  if (u256_lt(z, U256_2P251) && !u256_is_zero(d) && u256_lt(d, U256_N)) {
    for (int attempt = 0; attempt < 8; ++attempt) {
      const u256 k = rfc6979_nonce(z, d, seed);
      u256 r, s;
      st = sign_attempt(z, d, k, gen, wbits, nwin, r, s);
      if (st == SP_SIGN_OK) {
        st_u256(orr + 4 * e, r);
        st_u256(os + 4 * e, s);
      }
      if (st != SP_SIGN_RETRY) break;
      ++seed;
    }
  }
  status[e] = st;
}

// ---- the batch signer with compaction between RFC 6979 candidates (round 4) ------------------------------------
// Half of all candidates are rejected (a 252-bit candidate against N ~ 2^251), so in the one-kernel signer above a
// wave of 64 items runs 16 + 8 x 6.2 compressions per lane where an item needs 24 on average
// (profiles/r04_rfc6979_chain_ubench.txt).  For large batches the nonce phase is cut into rounds: every item's FIRST
// candidate in one launch, the rejected items' states (kin, kout, V: 24 words, word-major planes) appended to a
// compact list, the next launch runs one retry (8 compressions) for exactly those, and so on; a last launch loops
// the stragglers to the end.  The nonces land in a buffer and ecdsa_sign_kernel signs with them; an item whose
// attempt asks for the next seed (a 2^-55 event) is redone by the one-kernel signer, which owns that rule.
// The serial retry chain of the batch's slowest item bounds any schedule, so this pays where the batch is much larger
// than the chip: 2^20 items 5.30 -> 3.60 ms (2.0 -> 2.9 x 10^8 signatures/s), 2^18 1.71 -> 1.48 ms; at 2^16 it is
// a tie (0.84 against 0.86 ms), at 2^12 0.50 against 0.61 ms (the survivors run on lone waves); below 4096 items the
// one-kernel form stays (one launch instead of thirteen).  profiles/r04_sign_compaction.txt.
constexpr int RFC_STATE_WORDS = 24;

__device__ __forceinline__ void rfc_append(bool rejected, uint32_t item, const rfc_state& st, uint32_t* __restrict__ idx_out,
                                           uint32_t* __restrict__ state_out, size_t cap, uint32_t* __restrict__ counter) {
  const uint64_t m = __ballot(rejected);
  if (m == 0) return;
  const int lane = (int)__lane_id();
  const int leader = __ffsll((long long)m) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = (uint32_t)__shfl((int)base, leader, 64);
  if (rejected) {
    const size_t slot = (size_t)base + (size_t)__popcll(m & ((1ull << lane) - 1ull));
    idx_out[slot] = item;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      state_out[(size_t)w * cap + slot] = st.kin.h[w];
      state_out[(size_t)(8 + w) * cap + slot] = st.kout.h[w];
      state_out[(size_t)(16 + w) * cap + slot] = st.v.h[w];
    }
  }
}

__global__ void __launch_bounds__(128)
sign_nonce_first_kernel(const uint64_t* __restrict__ pz, const uint64_t* __restrict__ pd,
                        const uint64_t* __restrict__ pseed, uint64_t* __restrict__ kbuf, size_t n,
                        uint32_t* __restrict__ idx_out, uint32_t* __restrict__ state_out, uint32_t* __restrict__ counter) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = e < n;
  const size_t ee = active ? e : n - 1;
  const u256 z = ld_u256(pz + 4 * ee), d = ld_u256(pd + 4 * ee);
  const bool valid = active && u256_lt(z, U256_2P251) && !u256_is_zero(d) && u256_lt(d, U256_N);
  rfc_input in;
  rfc6979_prepare(z, d, pseed ? pseed[ee] : 0, in);
  rfc_state st;
  u256 cand;
  const bool accepted = rfc6979_run<true>(in, st, 1, cand);
  if (active && (accepted || !valid)) {
    if (!valid) {
#pragma unroll
      for (int i = 0; i < 8; ++i) cand.w[i] = 0;  // ecdsa_sign_kernel answers SP_SIGN_BAD_INPUT for such an item
    }
    st_u256(kbuf + 4 * e, cand);
  }
  rfc_append(valid && !accepted, (uint32_t)ee, st, idx_out, state_out, n, counter);
}

// One more candidate (max_rejected = 1) for the items of a compact list, or - the last launch - as many as it takes.
__global__ void __launch_bounds__(128)
sign_nonce_retry_kernel(const uint32_t* __restrict__ idx_in, const uint32_t* __restrict__ state_in,
                        const uint32_t* __restrict__ count_in, size_t cap, int max_rejected,
                        uint64_t* __restrict__ kbuf, uint32_t* __restrict__ idx_out, uint32_t* __restrict__ state_out,
                        uint32_t* __restrict__ counter) {
  const size_t count = *count_in;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const rfc_input none = {};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; (i & ~(size_t)63) < count; i += stride) {
    const bool active = i < count;
    const size_t slot = active ? i : count - 1;
    const uint32_t item = idx_in[slot];
    rfc_state st;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      st.kin.h[w] = state_in[(size_t)w * cap + slot];
      st.kout.h[w] = state_in[(size_t)(8 + w) * cap + slot];
      st.v.h[w] = state_in[(size_t)(16 + w) * cap + slot];
    }
    u256 cand;
    const bool accepted = rfc6979_run<false>(none, st, max_rejected, cand);
    const bool last = idx_out == nullptr;
    // (64 rejected candidates in a row in the last launch - a 2^-64 event - leave k = 0: the attempt then answers
    // SP_SIGN_BAD_INPUT where the one-kernel signer says SP_SIGN_RETRY; either way the item is the caller's)
    if (active && (accepted || last)) st_u256(kbuf + 4 * (size_t)item, cand);
    if (!last) rfc_append(active && !accepted, item, st, idx_out, state_out, cap, counter);
  }
}

// The one-kernel signer for exactly the items whose attempt asked for another seed (status SP_SIGN_RETRY).
__global__ void __launch_bounds__(128) SP_SIGN_OCCUPANCY
ecdsa_sign_rfc6979_redo_kernel(const uint64_t* __restrict__ pz, const uint64_t* __restrict__ pd,
                               const uint64_t* __restrict__ pseed, uint64_t* __restrict__ orr,
                               uint64_t* __restrict__ os, uint8_t* __restrict__ status, size_t n,
                               const aff_packed* __restrict__ gen, int wbits, int nwin) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n || status[e] != SP_SIGN_RETRY) return;
  const u256 z = ld_u256(pz + 4 * e), d = ld_u256(pd + 4 * e);
  uint64_t seed = pseed ? pseed[e] : 0;
  uint8_t st = SP_SIGN_RETRY;
  for (int attempt = 0; attempt < 8; ++attempt) {
    const u256 k = rfc6979_nonce(z, d, seed);
    u256 r, s;
    st = sign_attempt(z, d, k, gen, wbits, nwin, r, s);
    if (st == SP_SIGN_OK) {
      st_u256(orr + 4 * e, r);
      st_u256(os + 4 * e, s);
    }
    if (st != SP_SIGN_RETRY) break;
    ++seed;
  }
  status[e] = st;
}

}  // namespace sp

using namespace sp;

// Per-stream, like the Pedersen scratch: verifications in flight on different streams (or issued by
// different host threads) never share a window table.
static std::map<sp::StreamKey, sp::DeviceBuffer> g_verify_tab;
static std::map<sp::StreamKey, sp::DeviceBuffer> g_sign_scratch;  // nonces + compact lists of the batch signer, per stream
// Key-table cache (see "Key tables" above): slot -> 128-entry comb table, curve-model constant c and
// a flag, all in HBM; the host keeps the (qx, qy | x-only) -> slot map.
struct KeyId {
  std::array<uint64_t, 9> w;  // qx, qy (zero for an x-only key), has_y
  bool operator==(const KeyId& o) const { return w == o.w; }
};
struct KeyIdHash {
  size_t operator()(const KeyId& k) const {
    uint64_t h = 0x9e3779b97f4a7c15ull;
    for (uint64_t v : k.w) h = (h ^ v) * 0xff51afd7ed558ccdull + (h >> 29);
    return (size_t)h;
  }
};
// Slots 0 and 1 are SENTINELS that own no table: every x-only key that is not on the curve shares slot 0
// (flag KEY_INVALID_X -> the verdict of signature.py:232-235), every point key off the curve slot 1
// (KEY_OFF_CURVE -> :241); such keys are answered from the host map `invalid` and never take a slot of
// their own.  The tables are allocated lazily and grow by doubling up to `limit` slots (the first scalar
// verify does not pin 4 GiB); when the AUTO policy finds the cache full it starts a new generation (every
// table evicted, handles of earlier generations answer SP_VERIFY_STALE_SLOT) instead of falling back to
// the ladder for good.
constexpr uint32_t SENTINEL_INVALID_X = 0, SENTINEL_OFF_CURVE = 1, FIRST_KEY_SLOT = 2;
struct KeyCache {
  sp::DeviceBuffer tab, c, flag, stage;
  size_t capacity = 0, used = 0, limit = 0;  // allocated slots, slots handed out (sentinels included), ceiling
  uint32_t generation = 1;
  std::unordered_map<KeyId, uint32_t, KeyIdHash> slot_of;
  std::unordered_map<KeyId, uint32_t, KeyIdHash> invalid;  // key -> sentinel slot
  std::vector<uint32_t> free_slots;                        // slots taken back from invalid keys
  // a caller holds handles from sp_ecdsa_register_keys: the verify policy must not start a new generation
  // behind its back (its items would all come back SP_VERIFY_STALE_SLOT) - it falls back to the ladder instead
  bool external_handles = false;
};
static KeyCache g_keys;
static std::unordered_map<KeyId, uint8_t, KeyIdHash> g_seen_keys;  // unregistered keys met before (verify policy)
// value mod p of a 256-bit little-endian integer (host; at most 31 subtractions, keys are < p in practice)
static void reduce_mod_p(const uint64_t* in, uint64_t* out) {
  static const uint64_t P[4] = {1ull, 0ull, 0ull, 0x0800000000000011ull};
  for (int i = 0; i < 4; ++i) out[i] = in[i];
  for (;;) {
    bool ge = true;
    for (int i = 3; i >= 0; --i) {
      if (out[i] != P[i]) { ge = out[i] > P[i]; break; }
    }
    if (!ge) return;
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 4; ++i) {
      const unsigned __int128 d = (unsigned __int128)out[i] - P[i] - borrow;
      out[i] = (uint64_t)d;
      borrow = (d >> 64) & 1;
    }
  }
}
// Identity of a key in the cache: coordinates reduced mod p (x and x + p are the same key to every
// kernel) and an explicit point / x-only flag (no sentinel y that a caller's point key could collide with).
static KeyId key_id(const uint64_t* qx, const uint64_t* qy) {
  KeyId k;
  reduce_mod_p(qx, &k.w[0]);
  if (qy) reduce_mod_p(qy, &k.w[4]);
  else for (int i = 0; i < 4; ++i) k.w[4 + i] = 0;
  k.w[8] = qy ? 1 : 0;
  return k;
}
namespace sp {
void release_ecdsa_state() {
  for (auto& kv : g_verify_tab) kv.second.release();
  g_verify_tab.clear();
  for (auto& kv : g_sign_scratch) {  // scrubbed behind every call already; once more before the memory goes back,
    if (kv.second.ptr) {             // on the device of the context that owns the scratch
      DeviceScope on(ctx_at(kv.first.first).device);
      if (hipMemset(kv.second.ptr, 0, kv.second.bytes) != hipSuccess) {
        (void)hipGetLastError();
        fprintf(stderr, "libstarkperp: could not wipe %zu bytes of signer scratch at shutdown\n", kv.second.bytes);
      }
    }
    kv.second.release();
  }
  g_sign_scratch.clear();
  g_keys.tab.release();
  g_keys.c.release();
  g_keys.flag.release();
  g_keys.stage.release();
  g_keys.capacity = g_keys.used = g_keys.limit = 0;
  g_keys.slot_of.clear();
  g_keys.invalid.clear();
  g_keys.free_slots.clear();
  g_keys.external_handles = false;
  g_seen_keys.clear();
}
}
static inline unsigned nblocks(size_t n, unsigned tpb) { return (unsigned)((n + tpb - 1) / tpb); }

extern "C" {

int sp_ecdsa_verify_batch_dev(const uint64_t* z, const uint64_t* r, const uint64_t* s,
                              const uint64_t* qx, const uint64_t* qy, uint8_t* result, size_t n,
                              void* stream) {
  CtxByPointer sp_ctx_sel__(z);  // the context of the device these pointers live on
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  Context& c = ctx();
  ctx_lock lk(c.mu);
  // per-signature table of the eight odd multiples of the key: 8 x 36 limbs (X, Y, Z, prefix product), limb-major
  DeviceBuffer& tab = g_verify_tab[stream_key((hipStream_t)stream)];
  SP_HIP(tab.reserve((size_t)nblocks(n, VERIFY_TPB) * VERIFY_TPB * 8 * 36 * sizeof(int32_t)));  // one slot per lane
  hipLaunchKernelGGL(ecdsa_verify_kernel, dim3(nblocks(n, VERIFY_TPB)), dim3(VERIFY_TPB), 0, (hipStream_t)stream, z, r,
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

// The same on a host lane (context.hpp): the lane's own staging buffer, copies ordered on the lane's stream.
static int stage_in_lane(HostLane& L, const uint64_t* const* host, int count, size_t n, uint64_t** dev, size_t extra,
                         char** extra_ptr) {
  const size_t fb = n * 32;
  SP_HIP(L.io.reserve((size_t)count * fb + extra + 256));
  char* base = (char*)L.io.ptr;
  // small batches go through the lane's page-locked buffer (one host memcpy per input, then asynchronous DMA); a
  // failed page-locked allocation just keeps the direct copies
  char* stage = nullptr;
  if ((size_t)count * fb <= PINNED_STAGE_MAX && L.hio.reserve((size_t)count * fb) == hipSuccess) stage = (char*)L.hio.ptr;
  else (void)hipGetLastError();
  for (int i = 0; i < count; ++i) {
    if (host[i]) {
      dev[i] = (uint64_t*)(base + (size_t)i * fb);
      const void* src = host[i];
      if (stage) {
        std::memcpy(stage + (size_t)i * fb, host[i], fb);
        src = stage + (size_t)i * fb;
      }
      SP_HIP(hipMemcpyAsync(dev[i], src, fb, hipMemcpyHostToDevice, L.stream));
    } else {
      dev[i] = nullptr;
    }
  }
  *extra_ptr = base + (size_t)count * fb;
  return SP_OK;
}

static int key_sentinels() {
  const uint8_t f[2] = {KEY_INVALID_X, KEY_OFF_CURVE};
  SP_HIP(hipMemcpy(g_keys.flag.ptr, f, 2, hipMemcpyHostToDevice));
  return SP_OK;
}

// Makes room for `slots` slots: the first call allocates 4096 (128 MiB of tables), later ones double the
// allocation and carry the existing tables over, up to the limit (2^17 slots = 4 GiB by default,
// STARKPERP_KEY_CACHE_SLOTS).  hipFree of the old buffers waits for kernels still reading them.
static int key_cache_reserve(size_t slots) {
  if (g_keys.limit == 0) {
    size_t cap = (size_t)1 << 17;  // 128 Ki keys x 4 tables x 8 KiB = 4 GiB of tables
    if (const char* env = getenv("STARKPERP_KEY_CACHE_SLOTS")) {
      const long long v = atoll(env);
      if (v > 0) cap = (size_t)v + FIRST_KEY_SLOT;
    }
    if (cap > SLOT_INDEX_MASK) cap = SLOT_INDEX_MASK;  // a slot handle carries 24 index bits
    g_keys.limit = cap;
  }
  if (slots > g_keys.limit) slots = g_keys.limit;
  if (slots <= g_keys.capacity) return SP_OK;
  size_t cap = g_keys.capacity ? g_keys.capacity * 2 : 4096;
  while (cap < slots) cap *= 2;
  if (cap > g_keys.limit) cap = g_keys.limit;
  DeviceBuffer tab, c, flag;
  SP_HIP(tab.reserve(cap * KEY_ENTRIES * sizeof(aff_packed)));
  SP_HIP(c.reserve(cap * 32));
  SP_HIP(flag.reserve(cap));
  SP_HIP(hipMemset(flag.ptr, 0, cap));
  if (g_keys.used) {
    SP_HIP(hipMemcpy(tab.ptr, g_keys.tab.ptr, g_keys.used * KEY_ENTRIES * sizeof(aff_packed), hipMemcpyDeviceToDevice));
    SP_HIP(hipMemcpy(c.ptr, g_keys.c.ptr, g_keys.used * 32, hipMemcpyDeviceToDevice));
    SP_HIP(hipMemcpy(flag.ptr, g_keys.flag.ptr, g_keys.used, hipMemcpyDeviceToDevice));
  }
  g_keys.tab.release();
  g_keys.c.release();
  g_keys.flag.release();
  g_keys.tab = tab;
  g_keys.c = c;
  g_keys.flag = flag;
  const bool first = g_keys.capacity == 0;
  g_keys.capacity = cap;
  if (first) {
    g_keys.used = FIRST_KEY_SLOT;
    return key_sentinels();
  }
  return SP_OK;
}
static int key_cache_ready() { return g_keys.capacity ? SP_OK : key_cache_reserve(4096); }

// Builds the tables of keys [first, first + count) of the `fresh` list (host arrays), synchronously.
static int build_key_tables(const std::vector<uint64_t>& qx, const std::vector<uint64_t>& qy,
                            const std::vector<uint8_t>& has_y, const std::vector<uint32_t>& slots) {
  const size_t m = slots.size();
  if (m == 0) return SP_OK;
  const size_t chunk_max = (size_t)1 << 13;  // bounds the scratch: 4 x 92 KiB of work planes per key
  for (size_t first = 0; first < m; first += chunk_max) {
    const size_t cnt = m - first < chunk_max ? m - first : chunk_max;
    const size_t fb = cnt * 32;
    const size_t rows_b = cnt * KEY_ROW_POINTS * sizeof(aff_packed);
    const size_t work1 = cnt * KEY_ROW_POINTS * 36 * sizeof(int32_t);
    const size_t work2 = cnt * COMB_TABLES * 4 * 32 * 45 * sizeof(int32_t);
    const size_t work_b = work1 > work2 ? work1 : work2;
    SP_HIP(g_keys.stage.reserve(2 * fb + cnt + cnt * 4 + rows_b + work_b + 1024));
    char* b = (char*)g_keys.stage.ptr;
    aff_packed* d_rows = (aff_packed*)b;                       // 64-byte aligned first
    int32_t* d_work = (int32_t*)(b + rows_b);
    uint64_t* d_qx = (uint64_t*)(b + rows_b + work_b);
    uint64_t* d_qy = (uint64_t*)(b + rows_b + work_b + fb);
    uint32_t* d_slot = (uint32_t*)(b + rows_b + work_b + 2 * fb);
    uint8_t* d_hasy = (uint8_t*)(b + rows_b + work_b + 2 * fb + cnt * 4);
    SP_HIP(hipMemcpy(d_qx, qx.data() + 4 * first, fb, hipMemcpyHostToDevice));
    SP_HIP(hipMemcpy(d_qy, qy.data() + 4 * first, fb, hipMemcpyHostToDevice));
    SP_HIP(hipMemcpy(d_slot, slots.data() + first, cnt * 4, hipMemcpyHostToDevice));
    SP_HIP(hipMemcpy(d_hasy, has_y.data() + first, cnt, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(key_rows_kernel, dim3(nblocks(cnt, 64)), dim3(64), 0, 0, d_qx, d_qy, d_hasy, d_slot,
                       cnt, (uint64_t*)g_keys.c.ptr, (uint8_t*)g_keys.flag.ptr, d_rows, d_work);
    hipLaunchKernelGGL(key_table_kernel, dim3(nblocks(4 * COMB_TABLES * cnt, 64)), dim3(64), 0, 0, d_rows, d_slot, cnt,
                       (const uint8_t*)g_keys.flag.ptr, (aff_packed*)g_keys.tab.ptr, d_work);
    SP_HIP(hipGetLastError());
    SP_HIP(hipDeviceSynchronize());
  }
  return SP_OK;
}

// Registration proper; the caller holds the context lock (and keeps it until the launch that uses the handles
// is enqueued when it is the verify policy that registers: ADVICE r3).
static int register_keys_locked(const uint64_t* qx, const uint64_t* qy, size_t n, uint32_t* slots) {
  int rc = key_cache_ready();
  if (rc != SP_OK) return rc;
  // size cap of the "known not to be a key" map, applied BEFORE this call records anything: a clear between two
  // bad keys of one call would forget the first one and leave its handle on a slot that has been freed
  if (g_keys.invalid.size() > ((size_t)1 << 20)) g_keys.invalid.clear();
  std::vector<uint64_t> fx, fy;
  std::vector<uint8_t> fh;
  std::vector<uint32_t> fs;
  std::vector<KeyId> added;
  std::vector<size_t> pending;  // items whose slot is decided after the tables are built
  auto roll_back = [&]() {
    for (size_t j = 0; j < added.size(); ++j) {
      g_keys.slot_of.erase(added[j]);
      g_keys.free_slots.push_back(fs[j]);
    }
  };
  for (size_t i = 0; i < n; ++i) {
    const KeyId id = key_id(qx + 4 * i, qy ? qy + 4 * i : nullptr);
    auto bad = g_keys.invalid.find(id);
    if (bad != g_keys.invalid.end()) {  // known not to be a key: the shared sentinel slot, no table
      slots[i] = (g_keys.generation << SLOT_INDEX_BITS) | bad->second;
      continue;
    }
    auto it = g_keys.slot_of.find(id);
    if (it == g_keys.slot_of.end()) {
      uint32_t slot;
      if (!g_keys.free_slots.empty()) {
        slot = g_keys.free_slots.back();
        g_keys.free_slots.pop_back();
      } else {
        if (g_keys.used == g_keys.capacity && g_keys.capacity < g_keys.limit) {
          rc = key_cache_reserve(g_keys.used + 1);
          if (rc != SP_OK) { roll_back(); return rc; }
        }
        if (g_keys.used >= g_keys.capacity) {
          roll_back();
          set_error("key-table cache is full (" + std::to_string(g_keys.limit - FIRST_KEY_SLOT) +
                    " slots; STARKPERP_KEY_CACHE_SLOTS, sp_ecdsa_key_cache_reset)");
          return SP_ERR_CACHE_FULL;
        }
        slot = (uint32_t)g_keys.used++;
      }
      it = g_keys.slot_of.emplace(id, slot).first;
      added.push_back(id);
      fx.insert(fx.end(), qx + 4 * i, qx + 4 * i + 4);
      if (qy) fy.insert(fy.end(), qy + 4 * i, qy + 4 * i + 4);
      else fy.insert(fy.end(), 4, 0);
      fh.push_back(qy ? 1 : 0);
      fs.push_back(slot);
    }
    slots[i] = (g_keys.generation << SLOT_INDEX_BITS) | it->second;
    pending.push_back(i);
  }
  rc = build_key_tables(fx, fy, fh, fs);
  if (rc != SP_OK) {  // leave no slot behind whose table was never built
    roll_back();
    return rc;
  }
  // A new key that turned out not to be one (x-only and x^3 + x + beta a non-residue, or a point off the
  // curve: flagged by key_rows_kernel) gives its slot back and is remembered in the host map: an untrusted
  // key stream cannot fill the cache with keys that will never verify anything.
  if (!fs.empty()) {
    // flags of the NEW slots only: they are fs[] - mostly one contiguous run handed out by `used++`, plus recycled
    // slots - so copy the covering range of the run and the recycled ones one by one (round 3 copied all `used`
    // flag bytes per registration: 16 MiB with the largest STARKPERP_KEY_CACHE_SLOTS)
    std::vector<uint8_t> flags(fs.size());
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (uint32_t sl : fs) { lo = sl < lo ? sl : lo; hi = sl > hi ? sl : hi; }
    if ((size_t)(hi - lo) + 1 <= 4 * fs.size() + 64) {
      std::vector<uint8_t> span((size_t)(hi - lo) + 1);
      SP_HIP(hipMemcpy(span.data(), (const uint8_t*)g_keys.flag.ptr + lo, span.size(), hipMemcpyDeviceToHost));
      for (size_t j = 0; j < fs.size(); ++j) flags[j] = span[fs[j] - lo];
    } else {
      for (size_t j = 0; j < fs.size(); ++j)
        SP_HIP(hipMemcpy(&flags[j], (const uint8_t*)g_keys.flag.ptr + fs[j], 1, hipMemcpyDeviceToHost));
    }
    std::unordered_map<KeyId, uint32_t, KeyIdHash> bad_here;  // this call's bad keys -> sentinel slot
    for (size_t j = 0; j < fs.size(); ++j) {
      if (flags[j] != KEY_INVALID_X && flags[j] != KEY_OFF_CURVE) continue;
      const uint32_t sentinel = flags[j] == KEY_INVALID_X ? SENTINEL_INVALID_X : SENTINEL_OFF_CURVE;
      g_keys.slot_of.erase(added[j]);
      g_keys.invalid.emplace(added[j], sentinel);
      bad_here.emplace(added[j], sentinel);
      const uint8_t zero = KEY_EMPTY;
      SP_HIP(hipMemcpy((uint8_t*)g_keys.flag.ptr + fs[j], &zero, 1, hipMemcpyHostToDevice));
      g_keys.free_slots.push_back(fs[j]);
    }
    if (!bad_here.empty()) {
      for (size_t i : pending) {
        auto bad = bad_here.find(key_id(qx + 4 * i, qy ? qy + 4 * i : nullptr));
        if (bad != bad_here.end()) slots[i] = (g_keys.generation << SLOT_INDEX_BITS) | bad->second;
      }
    }
  }
  return rc;
}

int sp_ecdsa_register_keys(const uint64_t* qx, const uint64_t* qy, size_t n, uint32_t* slots) {
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  ctx_lock lk(ctx().mu);
  const int rc = register_keys_locked(qx, qy, n, slots);
  // Handles of this generation are in a caller's hands from here on: the verify policy must not evict behind them
  // (it serves a batch that does not fit on the ladder instead) until sp_ecdsa_key_cache_reset.  Set only when the
  // registration succeeded - a failed call hands out nothing (it used to be set before the attempt, so
  // one stray failing call switched the policy's eviction off for the rest of the process).
  if (rc == SP_OK) g_keys.external_handles = true;
  return rc;
}

int sp_ecdsa_key_cache_info(size_t* capacity, size_t* used) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  // capacity: the ceiling of the cache in keys (the allocation grows towards it); used: keys that own a table
  if (g_keys.limit == 0 && key_cache_reserve(0) != SP_OK) return SP_ERR_HIP;
  if (capacity) *capacity = g_keys.limit - FIRST_KEY_SLOT;
  if (used) *used = g_keys.capacity ? g_keys.used - FIRST_KEY_SLOT - g_keys.free_slots.size() : 0;
  return SP_OK;
}

int sp_ecdsa_key_cache_reset(void) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  SP_HIP(hipDeviceSynchronize());
  g_keys.slot_of.clear();
  g_keys.invalid.clear();
  g_keys.free_slots.clear();
  g_keys.external_handles = false;  // every handle is stale from here on, the caller that resets knows it
  g_keys.used = g_keys.capacity ? FIRST_KEY_SLOT : 0;
  g_keys.generation = g_keys.generation % 255u + 1u;
  if (g_keys.capacity) {
    SP_HIP(hipMemset(g_keys.flag.ptr, 0, g_keys.capacity));
    return key_sentinels();
  }
  return SP_OK;
}

int sp_ecdsa_verify_keyed_dev(const uint64_t* z, const uint64_t* r, const uint64_t* s,
                              const uint32_t* slots, uint8_t* result, size_t n, void* stream) {
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  Context& c = ctx();
  ctx_lock lk(c.mu);
  if (g_keys.capacity == 0) { set_error("no key has been registered"); return SP_ERR_BAD_ARGUMENT; }
  hipLaunchKernelGGL(ecdsa_verify_keyed_kernel, dim3(nblocks(n, VERIFY_TPB)), dim3(VERIFY_TPB), 0, (hipStream_t)stream, z,
                     r, s, slots, result, n, c.gen, c.wbits, c.nwin, (const aff_packed*)g_keys.tab.ptr,
                     (const uint64_t*)g_keys.c.ptr, (const uint8_t*)g_keys.flag.ptr, (uint32_t)g_keys.used,
                     g_keys.generation);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

// Policy of the host-pointer entry point.  A new key costs about 1.25 ladder verifications to
// tabulate and a tabulated verification about 0.25, so the tables pay off for keys that come back.
// A key counts as "coming back" when it is registered already, when it was seen in an earlier call
// (so a caller that verifies one signature at a time reaches the tables on the second sighting of a
// key), or when it repeats inside the batch; the tables are used when at most 40 % of the batch's
// signatures bring a key that is none of these.  STARKPERP_VERIFY_KEYED=0 / 1 forces the ladder /
// the tables.
static int initial_verify_policy() {
  const char* mode = getenv("STARKPERP_VERIFY_KEYED");
  if (mode && mode[0] == '0') return SP_VERIFY_POLICY_LADDER;
  if (mode && mode[0] == '1') return SP_VERIFY_POLICY_KEYED;
  return SP_VERIFY_POLICY_AUTO;
}
static int g_verify_policy = initial_verify_policy();  // read and written under the context lock

static bool use_key_tables(const uint64_t* qx, const uint64_t* qy, size_t n) {
  if (g_verify_policy == SP_VERIFY_POLICY_LADDER) return false;  // nothing remembered, nothing allocated
  if (key_cache_ready() != SP_OK) return false;
  std::unordered_map<KeyId, int, KeyIdHash> fresh;  // unregistered keys of this batch -> occurrences
  size_t registered = 0;                            // distinct keys of this batch that own a table already
  {
    std::unordered_map<KeyId, int, KeyIdHash> known;
    for (size_t i = 0; i < n; ++i) {
      const KeyId id = key_id(qx + 4 * i, qy ? qy + 4 * i : nullptr);
      if (g_keys.invalid.find(id) != g_keys.invalid.end()) continue;  // answered from the sentinel slots
      if (g_keys.slot_of.find(id) == g_keys.slot_of.end()) ++fresh[id];
      else if (++known[id] == 1) ++registered;
    }
  }
  size_t first_sightings = 0;  // signatures whose key is new to the library and unique in the batch
  for (const auto& kv : fresh) {
    if (kv.second == 1 && g_seen_keys.find(kv.first) == g_seen_keys.end()) ++first_sightings;
  }
  if (g_seen_keys.size() > ((size_t)1 << 20)) g_seen_keys.clear();
  for (const auto& kv : fresh) g_seen_keys.emplace(kv.first, 1);
  const bool keyed = g_verify_policy == SP_VERIFY_POLICY_KEYED || first_sightings * 5 <= n * 2;
  if (!keyed) return false;
  if (fresh.size() + registered + FIRST_KEY_SLOT > g_keys.limit) return false;  // more keys in one batch than the cache holds
  if (g_keys.used - g_keys.free_slots.size() + fresh.size() > g_keys.limit) {
    // full: the batch will need a new generation (everything evicted) rather than leave the tables to the keys
    // that came first - unless a caller holds handles of this generation (sp_ecdsa_register_keys): then the
    // ladder serves the batch.  The eviction itself is NOT done here: this function only decides;
    // verify_batch_keyed_impl evicts in the hold that registers, when registration really finds the cache full.
    if (g_keys.external_handles) return false;
  }
  return true;
}

int sp_ecdsa_set_verify_policy(int policy) {
  if (policy != SP_VERIFY_POLICY_AUTO && policy != SP_VERIFY_POLICY_LADDER && policy != SP_VERIFY_POLICY_KEYED) {
    set_error("sp_ecdsa_set_verify_policy: unknown policy");
    return SP_ERR_BAD_ARGUMENT;
  }
  ctx_lock lk(ctx().mu);
  g_verify_policy = policy;
  if (policy == SP_VERIFY_POLICY_LADDER) g_seen_keys.clear();
  return SP_OK;
}

int sp_ecdsa_get_verify_policy(void) {
  ctx_lock lk(ctx().mu);
  return g_verify_policy;
}

// Host-pointer verification through the key tables: registers the keys it has not seen, then runs
// the comb kernel.  `policy`: the call comes from sp_ecdsa_verify_batch.  Two holds of the context lock:
//   1. the DECISION (use_key_tables: reads the cache, remembers first sightings, evicts nothing), before a lane is
//      taken, so that a batch that ends on the ladder costs the primary device no lane and no stream;
//   2. EVICTION (only when registration finds the cache full, no caller holds handles and the batch fits an empty
//      cache), REGISTRATION and the LAUNCH against the tables, in one hold - a cache that another thread filled or
//      re-generated between the two holds is handled here, and one that still cannot take the batch sends it to the
//      ladder.  If no lane opens between the holds nothing has been evicted.
// *fell_back is set when the policy chose the ladder or the cache could not take the batch's keys after all, and
// nothing has been enqueued then.
static int verify_batch_keyed_impl(const uint64_t* z, const uint64_t* r, const uint64_t* s, const uint64_t* qx,
                                   const uint64_t* qy, uint8_t* result, size_t n, bool policy, bool* fell_back,
                                   const std::function<void()>* before_lock = nullptr) {
  // The key cache lives on the primary context: take a host lane of that context.  The lock covers the
  // bookkeeping (registration of new keys, the launch against the current tables); the copies and the
  // kernel run on the lane's stream and the lock is NOT held while the caller waits for them.
  // The policy speaks first, under the lock alone: a batch that ends on the ladder (fresh keys, the slices of a
  // multi-GPU caller) takes no lane and no stream of the primary device.  The lock is dropped while
  // the lane is acquired (a thread that waits for a lane must not hold the lock the lanes' owners need to
  // enqueue); registration and launch then share ONE hold, and a cache that another thread filled in between
  // answers SP_ERR_CACHE_FULL, which the policy path turns into the ladder.
  if (policy) {
    ctx_lock lk(ctx().mu);
    if (!use_key_tables(qx, qy, n)) { *fell_back = true; return SP_OK; }
  }
  LaneScope ls(0);
  if (ls.open() != SP_OK) {
    if (policy) { *fell_back = true; return SP_OK; }  // no stream for the tables: the ladder has its own lanes
    return SP_ERR_HIP;
  }
  HostLane& L = *ls.lane;
  std::vector<uint32_t> slots(n);
  uint8_t* d_res = nullptr;
  tl_mark("keyed verify: lane open");
  // z, r, s go to the lane's own buffer on the lane's own stream BEFORE the lock is taken: nothing shared is touched,
  // and three copies from pageable memory are the longest part of this function's host time (0.5 ms for 4096
  // signatures, profiles/r06_c3_timeline.txt) - round 5 made them under the lock, where sp_order_batch's tree update
  // (the critical path of that call) queued behind them.
  const uint64_t* host[3] = {z, r, s};
  uint64_t* dev[3];
  char* extra;
  int rc = stage_in_lane(L, host, 3, n, dev, n * 4 + n, &extra);
  if (rc != SP_OK) return rc;
  tl_mark("keyed verify: inputs staged");
  // sp_order_batch parks its verifier here until the tree update - the critical path of that call - has enqueued its
  // levels: both need the library lock, and the verifier used to win the race every few calls (+ 0.35 ms)
  if (before_lock) (*before_lock)();
  {
    ctx_lock lk(ctx().mu);
    tl_mark("keyed verify: context lock taken");
    // the handles never leave this call, so this is not an `external_handles` registration (sp_order_batch comes
    // through here on every batch: it must not switch the policy's eviction off)
    rc = register_keys_locked(qx, qy, n, slots.data());
    if (policy && rc == SP_ERR_CACHE_FULL && !g_keys.external_handles) {
      // full (register_keys_locked rolled its partial work back): start a new generation and register once more.
      // use_key_tables has checked that the batch's distinct keys fit an empty cache.
      if (sp_ecdsa_key_cache_reset() == SP_OK) rc = register_keys_locked(qx, qy, n, slots.data());
    }
    if (policy && rc == SP_ERR_CACHE_FULL) { *fell_back = true; lane_drain(&L); return SP_OK; }  // still no room: the ladder
    if (rc != SP_OK) { lane_drain(&L); return rc; }
    tl_mark("keyed verify: keys registered");
    uint32_t* d_slots = (uint32_t*)extra;
    d_res = (uint8_t*)(extra + n * 4);
    SP_HIP(hipMemcpyAsync(d_slots, slots.data(), n * 4, hipMemcpyHostToDevice, L.stream));
    rc = sp_ecdsa_verify_keyed_dev(dev[0], dev[1], dev[2], d_slots, d_res, n, L.stream);
    if (rc != SP_OK) return rc;
    tl_mark("keyed verify: enqueued");
  }
  // The verdicts come back OUTSIDE the lock: a copy into the caller's pageable memory makes the runtime wait for the
  // kernel in front of it, and round 5 held the library lock through that wait (the whole verification kernel,
  // 0.25 - 0.5 ms for 4096 signatures) - whoever needed the lock next, sp_order_batch's tree update for one, queued.
  SP_HIP(hipMemcpyAsync(result, d_res, n, hipMemcpyDeviceToHost, L.stream));
  SP_HIP(hipStreamSynchronize(L.stream));  // `slots` stays alive until the copy that reads it has run
  return SP_OK;
}

extern "C++" {
namespace sp {
// sp_order_batch's verifier (merkle.hip): the keyed verification with a hook that runs after the lock-free part
// (lane, staging copies) and before the library lock is taken.
int verify_batch_keyed_gated(const uint64_t* z, const uint64_t* r, const uint64_t* s, const uint64_t* qx,
                             const uint64_t* qy, uint8_t* result, size_t n, const std::function<void()>& before_lock) {
  SP_REQUIRE_READY();
  if (n == 0) { before_lock(); return SP_OK; }
  bool unused = false;
  return verify_batch_keyed_impl(z, r, s, qx, qy, result, n, false, &unused, &before_lock);
}
}  // namespace sp
}  // extern "C++"

int sp_ecdsa_verify_batch_keyed(const uint64_t* z, const uint64_t* r, const uint64_t* s,
                                const uint64_t* qx, const uint64_t* qy, uint8_t* result, size_t n) {
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  bool unused = false;
  return verify_batch_keyed_impl(z, r, s, qx, qy, result, n, false, &unused);
}

int sp_ecdsa_verify_batch(const uint64_t* z, const uint64_t* r, const uint64_t* s,
                          const uint64_t* qx, const uint64_t* qy, uint8_t* result, size_t n) {
  if (shard_context() < 0) {  // (a slice of a sharded batch: the policy has spoken for the whole batch)
    SP_REQUIRE_READY();
    if (n == 0) return SP_OK;
    bool ladder_forced;
    {
      ctx_lock lk(ctx().mu);
      ladder_forced = g_verify_policy == SP_VERIFY_POLICY_LADDER;  // nothing remembered, no lane of context 0 taken
    }
    if (!ladder_forced) {
      bool fell_back = false;
      const int rc = verify_batch_keyed_impl(z, r, s, qx, qy, result, n, true, &fell_back);
      if (rc != SP_OK || !fell_back) return rc;
    }
  }
  if (ctx_count() > 1 && shard_context() < 0 && n >= SHARD_MIN_ITEMS) {  // one slice per device, side by side
    return shard_over_contexts(n, [&](size_t off, size_t cnt) {
      return sp_ecdsa_verify_batch(z + 4 * off, r + 4 * off, s + 4 * off, qx + 4 * off, qy ? qy + 4 * off : nullptr,
                                   result + off, cnt);
    });
  }
  LaneScope ls;  // the ladder carries no shared state: calls from different host threads overlap
  SP_REQUIRE_READY();
  if (ls.open() != SP_OK) return SP_ERR_HIP;
  HostLane& L = *ls.lane;
  const uint64_t* host[5] = {z, r, s, qx, qy};
  uint64_t* dev[5];
  char* extra;
  int rc = stage_in_lane(L, host, 5, n, dev, n, &extra);
  if (rc != SP_OK) return rc;
  rc = sp_ecdsa_verify_batch_dev(dev[0], dev[1], dev[2], dev[3], dev[4], (uint8_t*)extra, n, L.stream);
  if (rc != SP_OK) return rc;
  SP_HIP(hipMemcpyAsync(result, extra, n, hipMemcpyDeviceToHost, L.stream));
  SP_HIP(hipStreamSynchronize(L.stream));
  return SP_OK;
}

// The whole of sign() for a batch on device pointers, enqueued on `st` (the context lock is held by the caller).
// Below the threshold (4096 items; STARKPERP_SIGN_COMPACT_MIN): ONE launch of the one-kernel signer.  From it on: first candidates -> rounds of one retry on
// the compacted survivors -> the stragglers -> the attempts -> the (practically empty) next-seed redo.
static size_t sign_compact_min() {
  static const size_t v = [] {
    const char* e = getenv("STARKPERP_SIGN_COMPACT_MIN");  // 0 = never compact
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)4096;
  }();
  return v;
}
// Scratch of the compacted pipeline: 232 B per item of a chunk (the nonces, two index lists, two lists of HMAC
// states).  Both lists are sized for the whole chunk - a batch of copies of one rejected item rejects everywhere,
// so n / 2 + slack would not be memory-safe - but a batch is cut into chunks of at most 2^20 items (the size from
// which the pipeline runs at its full rate, profiles/r04_sign_compaction.txt), so the scratch of a stream never
// exceeds 232 MiB + 25 % whatever n is.  STARKPERP_SIGN_CHUNK overrides the chunk size.
static size_t sign_chunk_items() {
  static const size_t v = [] {
    const char* e = getenv("STARKPERP_SIGN_CHUNK");
    const size_t c = e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)1 << 20);
    return c < 4096 ? (size_t)4096 : (c > ((size_t)1 << 30) ? ((size_t)1 << 30) : c);
  }();
  return v;
}
// Secret material staged in HBM (private keys, nonces, the HMAC state derived from them) is wiped when the scope
// that staged it ends - on EVERY path out of it (round 5 scrubbed on the success path only, and an
// early return after a failed launch or copy left the secrets in the lane's buffer).  finish() is the success
// path: the wipe is enqueued behind the last reader and the caller's own synchronize covers it.  The destructor
// is the error path: wipe, then wait for it.  A null range is skipped.
struct SecretScrub {
  void* ptr;
  size_t bytes;
  hipStream_t st;
  SecretScrub(void* p, size_t n, hipStream_t s) : ptr(p), bytes(n), st(s) {}
  hipError_t finish() {
    if (!ptr || !bytes) return hipSuccess;
    void* p = ptr;
    ptr = nullptr;
    return hipMemsetAsync(p, 0, bytes, st);
  }
  ~SecretScrub() {
    if (!ptr || !bytes) return;
    if (hipMemsetAsync(ptr, 0, bytes, st) != hipSuccess) {  // a stream in error: try the blocking form as well
      (void)hipGetLastError();
      (void)hipMemset(ptr, 0, bytes);
    }
    (void)hipStreamSynchronize(st);
  }
  SecretScrub(const SecretScrub&) = delete;
  SecretScrub& operator=(const SecretScrub&) = delete;
};

static int enqueue_sign_rfc6979(Context& c, const uint64_t* z, const uint64_t* d, const uint64_t* seeds, uint64_t* r,
                                uint64_t* s, uint8_t* status, size_t n, hipStream_t st) {
  const size_t min_n = sign_compact_min();
  auto one_kernel = [&](const uint64_t* z_, const uint64_t* d_, const uint64_t* seeds_, uint64_t* r_, uint64_t* s_,
                        uint8_t* status_, size_t n_) {
    hipLaunchKernelGGL(ecdsa_sign_rfc6979_kernel, dim3(nblocks(n_, 128)), dim3(128), 0, st, z_, d_, seeds_, r_, s_,
                       status_, n_, c.secret_gen(), c.secret_wbits(), c.secret_nwin());
  };
  if (min_n == 0 || n < min_n) {
    one_kernel(z, d, seeds, r, s, status, n);
    SP_HIP(hipGetLastError());
    return SP_OK;
  }
  constexpr int ROUNDS = 10;  // after them n / 2^11 items are left for the straggler launch
  const size_t chunk = n < sign_chunk_items() ? n : sign_chunk_items();
  DeviceBuffer& buf = g_sign_scratch[stream_key(st)];
  const size_t kb = chunk * 32, ib = chunk * 4, sb = chunk * 4 * RFC_STATE_WORDS;
  if (buf.reserve(kb + 2 * ib + 2 * sb + 256) != hipSuccess) {
    // no scratch, no compaction: the one-kernel signer needs none and computes the same signatures
    (void)hipGetLastError();
    one_kernel(z, d, seeds, r, s, status, n);
    SP_HIP(hipGetLastError());
    return SP_OK;
  }
  char* b = (char*)buf.ptr;
  uint64_t* kbuf = (uint64_t*)b;
  uint32_t* idx[2] = {(uint32_t*)(b + kb), (uint32_t*)(b + kb + ib)};
  uint32_t* state[2] = {(uint32_t*)(b + kb + 2 * ib), (uint32_t*)(b + kb + 2 * ib + sb)};
  uint32_t* counters = (uint32_t*)(b + kb + 2 * ib + 2 * sb);  // one per list: counters[j] = items that enter round j
  SecretScrub wipe(b, kb + 2 * ib + 2 * sb, st);  // error paths; every chunk also wipes behind itself below
  for (size_t off = 0; off < n; off += chunk) {
    const size_t m = n - off < chunk ? n - off : chunk;
    const uint64_t *zc = z + 4 * off, *dc = d + 4 * off, *sc = seeds ? seeds + off : nullptr;
    uint64_t *rc_ = r + 4 * off, *sc_ = s + 4 * off;
    uint8_t* stc = status + off;
    SP_HIP(hipMemsetAsync(counters, 0, 256, st));
    hipLaunchKernelGGL(sign_nonce_first_kernel, dim3(nblocks(m, 128)), dim3(128), 0, st, zc, dc, sc, kbuf, m, idx[0],
                       state[0], counters);
    for (int j = 0; j < ROUNDS; ++j) {
      const size_t expect = (m >> (j + 1)) + (m >> (j + 3)) + 8192;  // half of the previous list, + 25 % and slack; the loop in the kernel covers any count
      hipLaunchKernelGGL(sign_nonce_retry_kernel, dim3(nblocks(expect, 128)), dim3(128), 0, st, idx[j & 1], state[j & 1],
                         counters + j, m, 1, kbuf, idx[(j + 1) & 1], state[(j + 1) & 1], counters + j + 1);
    }
    hipLaunchKernelGGL(sign_nonce_retry_kernel, dim3(nblocks((m >> (ROUNDS + 1)) + 8192, 128)), dim3(128), 0, st,
                       idx[ROUNDS & 1], state[ROUNDS & 1], counters + ROUNDS, m, 64, kbuf, (uint32_t*)nullptr,
                       (uint32_t*)nullptr, (uint32_t*)nullptr);
    hipLaunchKernelGGL(ecdsa_sign_kernel, dim3(nblocks(m, 128)), dim3(128), 0, st, zc, dc, kbuf, rc_, sc_, stc, m, c.secret_gen(),
                       c.secret_wbits(), c.secret_nwin());
    hipLaunchKernelGGL(ecdsa_sign_rfc6979_redo_kernel, dim3(nblocks(m, 128)), dim3(128), 0, st, zc, dc, sc, rc_, sc_, stc,
                       m, c.secret_gen(), c.secret_wbits(), c.secret_nwin());
    // Secret material does not outlive the chunk in HBM: kbuf holds every nonce of the chunk - one
    // leaked k gives that item's private key away - and the state planes the HMAC K, V derived from d.  The
    // one-kernel signer keeps all of it in registers; here it is scrubbed on the same stream, behind its last reader.
    SP_HIP(hipMemsetAsync(b, 0, kb + 2 * ib + 2 * sb, st));
  }
  SP_HIP(hipGetLastError());
  wipe.ptr = nullptr;  // every chunk has been wiped on the stream already
  return SP_OK;
}

int sp_ecdsa_sign_batch(const uint64_t* z, const uint64_t* d, const uint64_t* k, uint64_t* r,
                        uint64_t* s, uint8_t* status, size_t n) {
  LaneScope ls;
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  if (!z || !d || !k || !r || !s || !status) { set_error("sp_ecdsa_sign_batch: null pointer"); return SP_ERR_BAD_ARGUMENT; }
  if (ls.open() != SP_OK) return SP_ERR_HIP;
  Context& c = ctx();
  HostLane& L = *ls.lane;
  const uint64_t* host[3] = {z, d, k};
  uint64_t* dev[3];
  char* extra;
  const size_t fb = n * 32;
  int rc = stage_in_lane(L, host, 3, n, dev, 2 * fb + n, &extra);
  if (rc != SP_OK) return rc;
  SecretScrub wipe(dev[1], 2 * fb, L.stream);  // the staged private keys and nonces (dev[1], dev[2] are adjacent)
  uint64_t* dr = (uint64_t*)extra;
  uint64_t* ds = (uint64_t*)(extra + fb);
  uint8_t* dst = (uint8_t*)(extra + 2 * fb);
  SP_HIP(hipMemsetAsync(dr, 0, 2 * fb, L.stream));
  hipLaunchKernelGGL(ecdsa_sign_kernel, dim3(nblocks(n, 128)), dim3(128), 0, L.stream, dev[0], dev[1], dev[2],
                     dr, ds, dst, n, c.secret_gen(), c.secret_wbits(), c.secret_nwin());
  SP_HIP(hipGetLastError());
  SP_HIP(hipMemcpyAsync(r, dr, fb, hipMemcpyDeviceToHost, L.stream));
  SP_HIP(hipMemcpyAsync(s, ds, fb, hipMemcpyDeviceToHost, L.stream));
  SP_HIP(hipMemcpyAsync(status, dst, n, hipMemcpyDeviceToHost, L.stream));
  SP_HIP(wipe.finish());
  SP_HIP(hipStreamSynchronize(L.stream));
  return SP_OK;
}

int sp_ecdsa_sign_rfc6979_batch(const uint64_t* z, const uint64_t* d, const uint64_t* seeds, uint64_t* r,
                                uint64_t* s, uint8_t* status, size_t n) {
  LaneScope ls;
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  if (!z || !d || !r || !s || !status) { set_error("sp_ecdsa_sign_rfc6979_batch: null pointer"); return SP_ERR_BAD_ARGUMENT; }
  if (ls.open() != SP_OK) return SP_ERR_HIP;
  Context& c = ctx();
  HostLane& L = *ls.lane;
  const uint64_t* host[2] = {z, d};
  uint64_t* dev[2];
  char* extra;
  const size_t fb = n * 32;
  int rc = stage_in_lane(L, host, 2, n, dev, 2 * fb + n * 8 + n + 64, &extra);
  if (rc != SP_OK) return rc;
  SecretScrub wipe(dev[1], fb, L.stream);  // the staged private keys do not stay in the lane's buffer
  uint64_t* dr = (uint64_t*)extra;
  uint64_t* ds = (uint64_t*)(extra + fb);
  uint64_t* dseed = (uint64_t*)(extra + 2 * fb);
  uint8_t* dst = (uint8_t*)(extra + 2 * fb + n * 8);
  SP_HIP(hipMemsetAsync(dr, 0, 2 * fb, L.stream));
  if (seeds) SP_HIP(hipMemcpyAsync(dseed, seeds, n * 8, hipMemcpyHostToDevice, L.stream));
  {
    ctx_lock lk(c.mu);  // the compacted pipeline keeps per-stream scratch in a shared map
    rc = enqueue_sign_rfc6979(c, dev[0], dev[1], seeds ? dseed : nullptr, dr, ds, dst, n, L.stream);
    if (rc != SP_OK) return rc;
  }
  SP_HIP(hipMemcpyAsync(r, dr, fb, hipMemcpyDeviceToHost, L.stream));
  SP_HIP(hipMemcpyAsync(s, ds, fb, hipMemcpyDeviceToHost, L.stream));
  SP_HIP(hipMemcpyAsync(status, dst, n, hipMemcpyDeviceToHost, L.stream));
  SP_HIP(wipe.finish());
  SP_HIP(hipStreamSynchronize(L.stream));
  return SP_OK;
}

// Device-pointer forms of the two signing calls (the batch signer of a device-resident pipeline: message
// hashes that sp_pedersen_chains_dev left in HBM are signed where they lie).  Nothing is staged and nothing waited
// for; r / s of an item whose status is not SP_SIGN_OK are left as they were.  sp_ecdsa_sign_batch_dev is ONE
// launch and touches no shared state.  sp_ecdsa_sign_rfc6979_batch_dev is one launch below 4096 items
// (STARKPERP_SIGN_COMPACT_MIN; 0 = always) and 15 launches + a scrub per chunk of 2^20 items from there on (the
// compacted nonce pipeline), with 232 B of per-stream scratch per item of a chunk: the first large call on a stream
// allocates it (hipMalloc: a device-wide synchronisation, not capturable into a graph - warm the stream up
// first or set STARKPERP_SIGN_COMPACT_MIN=0), and a failed allocation falls back to the one-kernel signer.
int sp_ecdsa_sign_batch_dev(const uint64_t* z, const uint64_t* d, const uint64_t* k, uint64_t* r, uint64_t* s,
                            uint8_t* status, size_t n, void* stream) {
  CtxByPointer sp_ctx_sel__(z);  // the context of the device these pointers live on
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  if (!z || !d || !k || !r || !s || !status) { set_error("sp_ecdsa_sign_batch_dev: null pointer"); return SP_ERR_BAD_ARGUMENT; }
  Context& c = ctx();
  ctx_lock lk(c.mu);
  hipLaunchKernelGGL(ecdsa_sign_kernel, dim3(nblocks(n, 128)), dim3(128), 0, (hipStream_t)stream, z, d, k, r, s, status,
                     n, c.secret_gen(), c.secret_wbits(), c.secret_nwin());
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_ecdsa_sign_rfc6979_batch_dev(const uint64_t* z, const uint64_t* d, const uint64_t* seeds, uint64_t* r,
                                    uint64_t* s, uint8_t* status, size_t n, void* stream) {
  CtxByPointer sp_ctx_sel__(z);  // the context of the device these pointers live on
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  if (!z || !d || !r || !s || !status) { set_error("sp_ecdsa_sign_rfc6979_batch_dev: null pointer"); return SP_ERR_BAD_ARGUMENT; }
  Context& c = ctx();
  ctx_lock lk(c.mu);
  return enqueue_sign_rfc6979(c, z, d, seeds, r, s, status, n, (hipStream_t)stream);
}

// (qx, qy) = d * EC_GEN on device pointers; qy and status may be null.  Outputs of a rejected item are left as they were.
int sp_public_key_batch_dev(const uint64_t* d, uint64_t* qx, uint64_t* qy, uint8_t* status, size_t n, void* stream) {
  CtxByPointer sp_ctx_sel__(d);  // the context of the device these pointers live on
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  if (!d || !qx) { set_error("sp_public_key_batch_dev: null pointer"); return SP_ERR_BAD_ARGUMENT; }
  Context& c = ctx();
  ctx_lock lk(c.mu);
  hipLaunchKernelGGL(public_key_kernel, dim3(nblocks(n, 128)), dim3(128), 0, (hipStream_t)stream, d, qx, qy, status, n,
                     c.secret_gen(), c.secret_wbits(), c.secret_nwin());
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_public_key_batch(const uint64_t* d, uint64_t* qx, uint64_t* qy, uint8_t* status, size_t n) {
  LaneScope ls;
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  if (!d || !qx) { set_error("sp_public_key_batch: null pointer"); return SP_ERR_BAD_ARGUMENT; }
  if (ls.open() != SP_OK) return SP_ERR_HIP;
  Context& c = ctx();
  HostLane& L = *ls.lane;
  const uint64_t* host[1] = {d};
  uint64_t* dev[1];
  char* extra;
  const size_t fb = n * 32;
  int rc = stage_in_lane(L, host, 1, n, dev, 2 * fb + n, &extra);
  if (rc != SP_OK) return rc;
  SecretScrub wipe(dev[0], fb, L.stream);  // the staged private keys do not stay in the lane's buffer
  uint64_t* dx = (uint64_t*)extra;
  uint64_t* dy = (uint64_t*)(extra + fb);
  uint8_t* dst = (uint8_t*)(extra + 2 * fb);
  SP_HIP(hipMemsetAsync(dx, 0, 2 * fb, L.stream));
  hipLaunchKernelGGL(public_key_kernel, dim3(nblocks(n, 128)), dim3(128), 0, L.stream, dev[0], dx, dy, dst, n,
                     c.secret_gen(), c.secret_wbits(), c.secret_nwin());
  SP_HIP(hipGetLastError());
  SP_HIP(hipMemcpyAsync(qx, dx, fb, hipMemcpyDeviceToHost, L.stream));
  if (qy) SP_HIP(hipMemcpyAsync(qy, dy, fb, hipMemcpyDeviceToHost, L.stream));
  if (status) SP_HIP(hipMemcpyAsync(status, dst, n, hipMemcpyDeviceToHost, L.stream));
  SP_HIP(wipe.finish());
  SP_HIP(hipStreamSynchronize(L.stream));
  return SP_OK;
}

}  // extern "C"

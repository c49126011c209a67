// Debug aid: quad-parallel additions (quad.hpp) against the serial formulas (curve.hpp) on the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "quad.hpp"
#include "curve_consts.hpp"
using namespace sp;

__device__ bool same(const fe& a, const fe& b) { return fe_eq(a, b); }

__global__ void k(int* res, int* dump) {
  const int lane = threadIdx.x, kq = lane & 3;
  // four affine points: small multiples built from the table seeds by serial arithmetic
  aff g{fe_to_mont(fe_unpack(PT_GEN_X)), fe_to_mont(fe_unpack(PT_GEN_Y))};
  aff s{fe_to_mont(fe_unpack(PT_SHIFT_X)), fe_to_mont(fe_unpack(PT_SHIFT_Y))};
  xyzz a1 = xyzz_mmadd(g, s);            // G + S
  xyzz a2 = xyzz_madd(a1, g);            // 2G + S  (not exceptional: a1 != g)
  xyzz a3 = xyzz_madd(a2, s);            // 2G + 2S
  // serial references
  const xyzz ref_mm = xyzz_mmadd(g, s);
  const xyzz ref_add = xyzz_add(a2, a3);
  // quad mmadd: both pairs = (g, s)
  qpt m = qmmadd(g.x, g.y, s.x, s.y, kq);
  const bool e = kq & 1;
  int bad = 0;
  if (lane < 4) {
    const fe c0 = fe_from_mont(m.a), c1 = fe_from_mont(ref_mm.X), c2 = fe_from_mont(ref_mm.Y), c3 = fe_from_mont(m.b);
    for (int i = 0; i < 9; ++i) { dump[lane * 36 + i] = c0.l[i]; dump[lane * 36 + 9 + i] = c1.l[i]; dump[lane * 36 + 18 + i] = c2.l[i]; dump[lane * 36 + 27 + i] = c3.l[i]; }
  }
  if (!same(m.a, e ? ref_mm.Y : ref_mm.X)) bad |= 1;
  if (!same(m.b, e ? ref_mm.ZZZ : ref_mm.ZZ)) bad |= 2;
  // quad add: P1 = a2 (lanes 0,1), P2 = a3 (lanes 2,3)
  qpt p;
  const bool h = kq & 2;
  p.a = h ? (e ? a3.Y : a3.X) : (e ? a2.Y : a2.X);
  p.b = h ? (e ? a3.ZZZ : a3.ZZ) : (e ? a2.ZZZ : a2.ZZ);
  qpt r = qadd<false>(p, kq);
  // compare as group elements: X3 * refZZ == refX * ZZ3 etc. (representations may differ)
  fe X3 = fe_dpp<quad_perm(0, 0, 0, 0)>(r.a), Y3 = fe_dpp<quad_perm(1, 1, 1, 1)>(r.a);
  fe ZZ3 = fe_dpp<quad_perm(0, 0, 0, 0)>(r.b), ZZZ3 = fe_dpp<quad_perm(1, 1, 1, 1)>(r.b);
  if (!same(fe_mul(X3, ref_add.ZZ), fe_mul(ref_add.X, ZZ3))) bad |= 4;
  if (!same(fe_mul(Y3, ref_add.ZZZ), fe_mul(ref_add.Y, ZZZ3))) bad |= 8;
  if (!same(X3, ref_add.X)) bad |= 16;
  if (!same(ZZ3, ref_add.ZZ)) bad |= 32;
  if (!same(Y3, ref_add.Y)) bad |= 64;
  if (!same(ZZZ3, ref_add.ZZZ)) bad |= 128;
  {  // step-by-step serial twin of qmmadd on (g, s)
    const fe P = fe_sub(s.x, g.x), R = fe_carry(fe_sub(s.y, g.y));
    const fe PP = fe_sqr(P), RR = fe_sqr(R), PPP = fe_mul(P, PP), Q = fe_mul(g.x, PP);
    const fe X3 = fe_carry(fe_sub(fe_sub(RR, PPP), fe_dbl(Q)));
    const fe A = fe_mul(R, fe_sub(Q, X3)), B = fe_mul(g.y, PPP);
    const fe Y2 = fe_sub(A, B);
    if (!same(Y2, ref_mm.Y)) bad |= 1024;      // two reductions vs one
    if (!same(X3, ref_mm.X)) bad |= 2048;
    const fe Yd = fe_sub(fe_dpp<quad_perm(1, 1, 3, 3)>(e ? A : B), fe_dpp<quad_perm(0, 0, 2, 2)>(e ? A : B));
    if (!same(Yd, ref_mm.Y)) bad |= 4096;      // the lane exchange alone
  }
  {  // qmmadd inline, every intermediate against the serial values
    const fe x1 = g.x, y1 = g.y, x2 = s.x, y2 = s.y;
    const fe P = fe_sub(x2, x1), R = fe_carry(fe_sub(y2, y1));
    const fe sPP = fe_sqr(P), sRR = fe_sqr(R), sPPP = fe_mul(P, sPP), sQ = fe_mul(x1, sPP);
    const fe sX3 = fe_carry(fe_sub(fe_sub(sRR, sPPP), fe_dbl(sQ)));
    const fe sA = fe_mul(R, fe_sub(sQ, sX3)), sB = fe_mul(y1, sPPP);
    const fe T1 = fe_sqr(fe_sel(e, R, P));
    if (!same(T1, e ? sRR : sPP)) bad |= 1 << 13;
    const fe PPb = fe_dpp<quad_perm(0, 0, 2, 2)>(T1);
    if (!same(PPb, sPP)) bad |= 1 << 14;
    const fe T2 = fe_mul(fe_sel(e, x1, P), PPb);
    if (!same(T2, e ? sQ : sPPP)) bad |= 1 << 15;
    const fe PPPb = fe_dpp<quad_perm(0, 0, 2, 2)>(T2), Qb = fe_dpp<quad_perm(1, 1, 3, 3)>(T2);
    if (!same(PPPb, sPPP)) bad |= 1 << 16;
    if (!same(Qb, sQ)) bad |= 1 << 17;
    const fe X3 = fe_carry(fe_sub(fe_sub(fe_dpp<quad_perm(1, 1, 3, 3)>(T1), PPPb), fe_dbl(Qb)));
    if (!same(X3, sX3)) bad |= 1 << 18;
    const fe rhs = fe_sel(e, fe_sub(Qb, X3), PPPb);
    if (!same(rhs, e ? fe_sub(sQ, sX3) : sPPP)) bad |= 1 << 19;
    const fe lhs = fe_sel(e, R, y1);
    if (!same(lhs, e ? R : y1)) bad |= 1 << 20;
    const fe T3 = fe_mul(lhs, rhs);
    if (!same(T3, e ? sA : sB)) bad |= 1 << 21;
    const fe Y3 = fe_sub(fe_dpp<quad_perm(1, 1, 3, 3)>(T3), fe_dpp<quad_perm(0, 0, 2, 2)>(T3));
    if (!same(Y3, ref_mm.Y)) bad |= 1 << 22;
  }
  qpt rx = qadd<true>(p, kq);
  if (!same(rx.a, ref_add.X)) bad |= 256;
  if (!same(rx.b, ref_add.ZZ)) bad |= 512;
  res[lane] = bad;
}
int main() {
  int* d; hipMalloc(&d, 64 * 4);
  int* dd; hipMalloc(&dd, 4 * 36 * 4);
  k<<<1, 64>>>(d, dd);
  int hd[4 * 36]; hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
  for (int l = 0; l < 4; ++l) for (int r = 0; r < 4; ++r) { printf("lane %d %s:", l, r == 0 ? "m.a " : r == 1 ? "refX" : r == 2 ? "refY" : "m.b "); for (int i = 0; i < 9; ++i) printf(" %08x", hd[l * 36 + r * 9 + i]); printf("\n"); }
  int h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 8; ++l) printf("lane %d: 0x%x\n", l, h[l]);
  return 0;
}

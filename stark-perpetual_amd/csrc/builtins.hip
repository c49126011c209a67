// The range-check BUILTIN as an AIR segment (build-defined, parity unpinned - the
// definition is oracle/stark_ref.py "rc16").  The Cairo program is `%builtins output pedersen range_check
// ecdsa` (services/perpetual/cairo/main.cairo:1) and its range checks sit on the order path
// (order/order.cairo:53-56).  A range-checked value is eight 16-bit limbs; every limb is a cell of column
// `a`; column `s` holds the same cells sorted, and after both are committed a challenge z is drawn and the
// SECOND-PHASE column p_i = prod_{j <= i} (z - a_j) / (z - s_j) ties them together (p_{n-1} = 1).
//
// Kernels here: witness columns (a, acc), the product column p (batched inversion + a three-pass prefix
// product over 32-byte felts), the rc16 part of the composition column (transition constraints, the two
// "every row but the last" constraints and four boundary constraints: this is the one AIR of the library
// whose composition needs the coset point x itself) and a felt addition for summing the compositions of the
// segments of a combined trace.  All columns are plain felts in HBM; arithmetic is Montgomery in registers.
#include <cstring>
#include <map>
#include <vector>

#include "context.hpp"

namespace sp {

__device__ __forceinline__ fe ld_plain_m(const uint64_t* p) { return fe_to_mont(fe_unpack(ld_u256(p))); }
__device__ __forceinline__ void st_plain(uint64_t* p, const fe& mont) { st_u256(p, fe_pack(fe_from_mont(mont))); }

// rows 8 v + k of value v: acc = value >> 16 (7 - k), a = acc mod 2^16
__global__ void __launch_bounds__(256)
rc16_trace_kernel(const uint64_t* __restrict__ values, size_t n_values, uint64_t* __restrict__ a_col,
                  uint64_t* __restrict__ acc_col) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= 8 * n_values) return;
  const size_t v = r >> 3;
  const int sh = 16 * (7 - (int)(r & 7));
  const uint64_t lo = values[4 * v], hi = values[4 * v + 1];  // values are < 2^128 (checked by the caller)
  uint64_t o0, o1;
  if (sh == 0) { o0 = lo; o1 = hi; }
  else if (sh < 64) { o0 = (lo >> sh) | (hi << (64 - sh)); o1 = hi >> sh; }
  else if (sh == 64) { o0 = hi; o1 = 0; }
  else { o0 = hi >> (sh - 64); o1 = 0; }
  acc_col[4 * r] = o0; acc_col[4 * r + 1] = o1; acc_col[4 * r + 2] = 0; acc_col[4 * r + 3] = 0;
  a_col[4 * r] = o0 & 0xffff; a_col[4 * r + 1] = 0; a_col[4 * r + 2] = 0; a_col[4 * r + 3] = 0;
}

// num = z - a, den = z - s  (plain, canonical)
__global__ void __launch_bounds__(256)
rc16_terms_kernel(const uint64_t* __restrict__ a, const uint64_t* __restrict__ s, size_t n, fe z_plain,
                  uint64_t* __restrict__ num, uint64_t* __restrict__ den) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st_u256(num + 4 * i, fe_pack(fe_canon(fe_sub(z_plain, fe_unpack(ld_u256(a + 4 * i))))));
  st_u256(den + 4 * i, fe_pack(fe_canon(fe_sub(z_plain, fe_unpack(ld_u256(s + 4 * i))))));
}

// In-place inversion of n plain felts (none zero): a thread owns K consecutive elements, Montgomery's trick
// with the prefix products parked in `tmp` (n felts), one divsteps inversion per thread.
constexpr int INV_K = 16;
__global__ void __launch_bounds__(256)
felt_batch_inverse_kernel(uint64_t* __restrict__ x, uint64_t* __restrict__ tmp, size_t n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t first = t * INV_K;
  if (first >= n) return;
  const int cnt = (int)(n - first < (size_t)INV_K ? n - first : (size_t)INV_K);
  fe run = FE_ONE_M;
  for (int j = 0; j < cnt; ++j) {
    st_u256(tmp + 4 * (first + j), fe_pack(fe_canon(run)));  // Montgomery form, canonical limbs
    run = fe_mul(run, ld_plain_m(x + 4 * (first + j)));
  }
  fe inv = fe_inv(run);
  for (int j = cnt - 1; j >= 0; --j) {
    const fe xj = ld_plain_m(x + 4 * (first + j));
    const fe pre = fe_unpack(ld_u256(tmp + 4 * (first + j)));
    st_plain(x + 4 * (first + j), fe_mul(inv, pre));
    inv = fe_mul(inv, xj);
  }
}

__global__ void __launch_bounds__(256)
felt_mul_kernel(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st_u256(out + 4 * i, fe_pack(fe_canon(fe_mul(ld_plain_m(a + 4 * i), fe_unpack(ld_u256(b + 4 * i))))));  // aR * b / R
}

__global__ void __launch_bounds__(256)
felt_add_kernel(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st_u256(out + 4 * i, fe_pack(fe_canon(fe_add(fe_unpack(ld_u256(a + 4 * i)), fe_unpack(ld_u256(b + 4 * i))))));
}

// Inclusive prefix product, three passes: (1) every thread scans its chunk of SCAN_L consecutive felts in place
// and writes the chunk's product to totals; (2) the totals are scanned (recursively, or by one thread when
// few); (3) every chunk but the first is multiplied by the scanned total of its predecessors.
constexpr int SCAN_L = 64;
__global__ void __launch_bounds__(256)
felt_scan_chunks_kernel(uint64_t* __restrict__ x, size_t n, uint64_t* __restrict__ totals) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t first = t * SCAN_L;
  if (first >= n) return;
  const int cnt = (int)(n - first < (size_t)SCAN_L ? n - first : (size_t)SCAN_L);
  fe run = FE_ONE_M;
  for (int j = 0; j < cnt; ++j) {
    run = fe_mul(run, ld_plain_m(x + 4 * (first + j)));
    st_plain(x + 4 * (first + j), run);
  }
  st_plain(totals + 4 * t, run);
}
__global__ void felt_scan_serial_kernel(uint64_t* __restrict__ x, size_t n) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  fe run = FE_ONE_M;
  for (size_t j = 0; j < n; ++j) {
    run = fe_mul(run, ld_plain_m(x + 4 * j));
    st_plain(x + 4 * j, run);
  }
}
__global__ void __launch_bounds__(256)
felt_scan_apply_kernel(uint64_t* __restrict__ x, size_t n, const uint64_t* __restrict__ scanned_totals) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t chunk = i / SCAN_L;
  if (chunk == 0) return;
  st_u256(x + 4 * i, fe_pack(fe_canon(fe_mul(ld_plain_m(scanned_totals + 4 * (chunk - 1)), fe_unpack(ld_u256(x + 4 * i))))));
}

// out[i] = x_i - c with x_i = shift * w_M^i (plain): the coset points the boundary constraints divide by
struct PowTable {
  fe pw[32];  // w_M^(2^b), Montgomery
};
__global__ void __launch_bounds__(256)
coset_minus_kernel(uint64_t* __restrict__ out, size_t M, int log_m, fe shift_m, PowTable tab, fe c_m) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  fe x = shift_m;
  for (int b = 0; b < log_m; ++b)
    if ((i >> b) & 1) x = fe_mul(x, tab.pw[b]);
  st_plain(out + 4 * i, fe_carry(fe_sub(x, c_m)));
}

struct Rc16AirParams {
  fe alpha[8];   // Montgomery
  fe zinv[4];    // Montgomery form of 1 / (x^n - 1) for i mod 4
  fe z, rc_min, rc_max, two16;  // Montgomery
};
// cols: a, acc, s (col_stride felts apart), p: the second-phase column; per: first8, step8 - 2 tables of 32 plain
// felts; d_last = x - g^(n-1), inv_last = 1 / (x - g^(n-1)), inv_first = 1 / (x - 1) at every coset point.
__global__ void __launch_bounds__(256)
air_eval_rc16_kernel(const uint64_t* __restrict__ cols, size_t col_stride, const uint64_t* __restrict__ p,
                     const uint64_t* __restrict__ per, const uint64_t* __restrict__ d_last,
                     const uint64_t* __restrict__ inv_last, const uint64_t* __restrict__ inv_first, size_t M,
                     Rc16AirParams prm, uint64_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const size_t j = (i + 4) & (M - 1);
  const fe a = ld_plain_m(cols + 4 * i), acc = ld_plain_m(cols + 4 * (col_stride + i)), s = ld_plain_m(cols + 4 * (2 * col_stride + i));
  const fe an = ld_plain_m(cols + 4 * j), accn = ld_plain_m(cols + 4 * (col_stride + j)), sn = ld_plain_m(cols + 4 * (2 * col_stride + j));
  const fe pc = ld_plain_m(p + 4 * i), pn = ld_plain_m(p + 4 * j);
  const fe first8 = ld_plain_m(per + 4 * (i & 31)), step8 = ld_plain_m(per + 4 * (32 + (i & 31)));
  const fe dl = ld_plain_m(d_last + 4 * i), il = ld_plain_m(inv_last + 4 * i), i1 = ld_plain_m(inv_first + 4 * i);
  // numerators, all Montgomery N-form
  const fe c0 = fe_mul(first8, fe_sub(acc, a));
  const fe c1 = fe_mul(step8, fe_carry(fe_sub(fe_sub(accn, fe_mul(acc, prm.two16)), an)));
  const fe d = fe_carry(fe_sub(sn, s));
  const fe c2 = fe_mul(d, fe_carry(fe_sub(d, FE_ONE_M)));
  const fe c3 = fe_mul_sub_mul(pn, fe_carry(fe_sub(prm.z, sn)), pc, fe_carry(fe_sub(prm.z, an)));
  const fe c4 = fe_carry(fe_sub(fe_mul(pc, fe_carry(fe_sub(prm.z, s))), fe_sub(prm.z, a)));
  const fe c5 = fe_carry(fe_sub(pc, FE_ONE_M));
  const fe c6 = fe_carry(fe_sub(s, prm.rc_min));
  const fe c7 = fe_carry(fe_sub(s, prm.rc_max));
  const fe t_all = fe_mul_add_mul(prm.alpha[0], c0, prm.alpha[1], c1);
  const fe t_notlast = fe_mul(fe_mul_add_mul(prm.alpha[2], c2, prm.alpha[3], c3), dl);
  const fe trans = fe_mul(fe_carry(fe_add(t_all, t_notlast)), prm.zinv[i & 3]);
  const fe firstp = fe_mul(fe_mul_add_mul(prm.alpha[4], c4, prm.alpha[6], c6), i1);
  const fe lastp = fe_mul(fe_mul_add_mul(prm.alpha[5], c5, prm.alpha[7], c7), il);
  st_plain(out + 4 * i, fe_carry(fe_add(fe_add(trans, firstp), lastp)));
}

static std::map<hipStream_t, DeviceBuffer> g_builtin_work;
void release_builtin_state() {
  for (auto& kv : g_builtin_work) kv.second.release();
  g_builtin_work.clear();
}

static fe h_pow(fe base_m, uint64_t e) {
  fe r = FE_ONE_M;
  for (int i = 63; i >= 0; --i) {
    r = fe_sqr(r);
    if ((e >> i) & 1) r = fe_mul(r, base_m);
  }
  return r;
}
static fe h_root(int log_n) {  // 3^((p - 1) / 2^log_n), p - 1 = 2^192 (2^59 + 17)
  const fe three = fe_to_mont(fe{{3, 0, 0, 0, 0, 0, 0, 0, 0}});
  fe c = h_pow(three, ((uint64_t)1 << 59) + 17);
  for (int i = 0; i < 192 - log_n; ++i) c = fe_sqr(c);
  return c;
}
static fe host_felt_m(const uint64_t* host) {
  u256 v;
  std::memcpy(v.w, host, 32);
  return fe_to_mont(fe_unpack(v));
}
static inline unsigned nb(size_t n) { return (unsigned)((n + 255) / 256); }

// inclusive prefix product of x[0 .. n) in place; `scratch` holds at least n / SCAN_L + 2 * SCAN_L felts
static int scan_mul(uint64_t* x, size_t n, uint64_t* scratch, hipStream_t st) {
  if (n <= 256) {
    hipLaunchKernelGGL(felt_scan_serial_kernel, dim3(1), dim3(64), 0, st, x, n);
    return SP_OK;
  }
  const size_t chunks = (n + SCAN_L - 1) / SCAN_L;
  hipLaunchKernelGGL(felt_scan_chunks_kernel, dim3(nb(chunks)), dim3(256), 0, st, x, n, scratch);
  int rc = scan_mul(scratch, chunks, scratch + 4 * chunks, st);
  if (rc != SP_OK) return rc;
  hipLaunchKernelGGL(felt_scan_apply_kernel, dim3(nb(n)), dim3(256), 0, st, x, n, scratch);
  return SP_OK;
}

}  // namespace sp

using namespace sp;

extern "C" {

int sp_rc16_trace_dev(const uint64_t* values, size_t n_values, uint64_t* cols, void* stream) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  if (n_values == 0) return SP_OK;
  const size_t n = 8 * n_values;
  hipLaunchKernelGGL(rc16_trace_kernel, dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, values, n_values, cols,
                     cols + 4 * n);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_rc16_product_dev(const uint64_t* a, const uint64_t* s, size_t n, const uint64_t* z_host, uint64_t* p,
                        void* stream) {
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  ctx_lock lk(ctx().mu);
  hipStream_t st = (hipStream_t)stream;
  DeviceBuffer& work = g_builtin_work[st];
  SP_HIP(work.reserve((2 * n + n / SCAN_L + 4 * SCAN_L + 64) * 32));
  uint64_t* den = (uint64_t*)work.ptr;
  uint64_t* tmp = den + 4 * n;
  u256 zv;
  std::memcpy(zv.w, z_host, 32);
  hipLaunchKernelGGL(rc16_terms_kernel, dim3(nb(n)), dim3(256), 0, st, a, s, n, fe_unpack(zv), p, den);
  hipLaunchKernelGGL(felt_batch_inverse_kernel, dim3(nb((n + INV_K - 1) / INV_K)), dim3(256), 0, st, den, tmp, n);
  hipLaunchKernelGGL(felt_mul_kernel, dim3(nb(n)), dim3(256), 0, st, p, den, p, n);  // (z - a_i) / (z - s_i)
  int rc = scan_mul(p, n, tmp, st);
  if (rc != SP_OK) return rc;
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_air_eval_rc16_dev(const uint64_t* cols, const uint64_t* p, const uint64_t* periodic_lde, unsigned log_n,
                         const uint64_t* alphas_host, const uint64_t* shift_host, const uint64_t* z_host,
                         uint64_t rc_min, uint64_t rc_max, uint64_t* out, void* stream) {
  SP_REQUIRE_READY();
  if (log_n < 3 || log_n > 24) { set_error("sp_air_eval_rc16_dev: log_n out of range (one value is 8 rows)"); return SP_ERR_BAD_ARGUMENT; }
  if (rc_min > rc_max || rc_max > 0xffff) { set_error("sp_air_eval_rc16_dev: bad limb range"); return SP_ERR_BAD_ARGUMENT; }
  ctx_lock lk(ctx().mu);
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)1 << log_n, M = 4 * n;
  const int log_m = (int)log_n + 2;
  Rc16AirParams prm;
  for (int k = 0; k < 8; ++k) prm.alpha[k] = host_felt_m(alphas_host + 4 * k);
  const fe shift_m = host_felt_m(shift_host);
  fe sn = shift_m;
  for (unsigned i = 0; i < log_n; ++i) sn = fe_sqr(sn);
  const fe w4 = h_root(2);
  fe wk = FE_ONE_M;
  for (int k = 0; k < 4; ++k) {
    prm.zinv[k] = fe_inv(fe_carry(fe_sub(fe_mul(sn, wk), FE_ONE_M)));
    wk = fe_mul(wk, w4);
  }
  prm.z = host_felt_m(z_host);
  prm.rc_min = fe_to_mont(fe{{(int32_t)rc_min, 0, 0, 0, 0, 0, 0, 0, 0}});
  prm.rc_max = fe_to_mont(fe{{(int32_t)rc_max, 0, 0, 0, 0, 0, 0, 0, 0}});
  prm.two16 = fe_to_mont(fe{{1 << 16, 0, 0, 0, 0, 0, 0, 0, 0}});
  // coset tables: x - g_last, 1 / (x - g_last), 1 / (x - 1)
  PowTable tab;
  fe w = h_root(log_m);
  for (int b = 0; b < 32; ++b) { tab.pw[b] = fe_mul(w, FE_ONE_M); w = fe_sqr(w); }
  const fe g = h_root((int)log_n);
  const fe g_last = fe_inv(g);  // g^(n - 1) = g^-1
  DeviceBuffer& work = g_builtin_work[st];
  SP_HIP(work.reserve((4 * M + 64) * 32));
  uint64_t* d_last = (uint64_t*)work.ptr;
  uint64_t* inv_last = d_last + 4 * M;
  uint64_t* inv_first = inv_last + 4 * M;
  uint64_t* tmp = inv_first + 4 * M;
  hipLaunchKernelGGL(coset_minus_kernel, dim3(nb(M)), dim3(256), 0, st, d_last, M, log_m, shift_m, tab, g_last);
  hipLaunchKernelGGL(coset_minus_kernel, dim3(nb(M)), dim3(256), 0, st, inv_last, M, log_m, shift_m, tab, g_last);
  hipLaunchKernelGGL(coset_minus_kernel, dim3(nb(M)), dim3(256), 0, st, inv_first, M, log_m, shift_m, tab, FE_ONE_M);
  hipLaunchKernelGGL(felt_batch_inverse_kernel, dim3(nb((M + INV_K - 1) / INV_K)), dim3(256), 0, st, inv_last, tmp, M);
  hipLaunchKernelGGL(felt_batch_inverse_kernel, dim3(nb((M + INV_K - 1) / INV_K)), dim3(256), 0, st, inv_first, tmp, M);
  hipLaunchKernelGGL(air_eval_rc16_kernel, dim3(nb(M)), dim3(256), 0, st, cols, M, p, periodic_lde, d_last, inv_last,
                     inv_first, M, prm, out);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

int sp_felt_add_dev(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  SP_REQUIRE_READY();
  if (n == 0) return SP_OK;
  ctx_lock lk(ctx().mu);
  hipLaunchKernelGGL(felt_add_kernel, dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  SP_HIP(hipGetLastError());
  return SP_OK;
}

}  // extern "C"

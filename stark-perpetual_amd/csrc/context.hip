// sp_init / sp_shutdown and the HBM window-table builder.
//
// Pedersen (signature.py:300-318) is  shift + sum_j x_j C[2+j] + sum_j y_j C[254+j]  over the 504
// per-bit constant points.  The device tables hold, for element e, window i and window value v,
//     T[e][i][v] = S[e][i] + sum_{b : bit b of v set} C[2 + 252 e + i w + b]
// as affine points, where the offsets S are points of unknown discrete logarithm relative to the
// hash generators (random multiples of EC_GEN) chosen so that sum_{e,i} S[e][i] = SHIFT_POINT.
// A hash is then the sum of 2*nwin table entries - no entry is the point at infinity, and a
// partial sum can only meet the next entry's x-coordinate through a non-trivial relation between
// independent generators, i.e. never for inputs anyone can compute.
// The EC_GEN table (k*G for sign / public keys / the z*G leg of verify) has the same shape with
// offsets that are multiples of P0 and sum to the point at infinity.
#include <cstring>
#include <vector>

#include "context.hpp"
#include "hostcurve.hpp"

namespace sp {

void release_pedersen_state();  // per-stream scratch, profiling events (pedersen.hip)
void release_merkle_state();    // sparse-update staging (merkle.hip)
void release_stark_state();     // twiddle / coset tables, work buffers (stark.hip)
void release_ecdsa_state();     // per-signature window tables (ecdsa.hip)

static Context g_ctx;
static std::string g_err;
static std::mutex g_err_mu;

Context& ctx() { return g_ctx; }
void set_error(const std::string& s) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  g_err = s;
}
int hip_fail(hipError_t e, const char* what) {
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return SP_ERR_HIP;
}

// One thread per table entry.
__global__ void __launch_bounds__(256)
build_table_kernel(aff_packed* tab, const aff_packed* bits, const aff_packed* offs, int wbits,
                   int nwin, int total_bits, fe beta_m, unsigned* bad) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)1 << wbits;
  if (idx >= per * (size_t)nwin) return;
  const int win = (int)(idx >> wbits);
  const uint32_t v = (uint32_t)(idx & (per - 1));
  int nb = total_bits - win * wbits;
  if (nb > wbits) nb = wbits;
  if (nb < 32 && v >= (1u << nb)) return;
  xyzz acc = xyzz_from_aff(ld_aff(offs + win));
  for (int b = 0; b < nb; ++b) {
    if ((v >> b) & 1u) acc = xyzz_madd(acc, ld_aff(bits + win * wbits + b));
  }
  const fe izzz = fe_inv(acc.ZZZ);
  const fe izz = fe_sqr(fe_mul(acc.ZZ, izzz));  // 1/ZZ = (ZZ/ZZZ)^2
  const fe x = fe_mul(acc.X, izz);
  const fe y = fe_mul(acc.Y, izzz);
  // on-curve check y^2 = x^3 + x + beta (also catches an exceptional addition: ZZ = 0 -> x = 0, y = 0)
  const fe lhs = fe_sqr(y);
  const fe rhs = fe_carry(fe_add(fe_add(fe_mul(fe_sqr(x), x), x), beta_m));
  if (!fe_eq(lhs, rhs)) atomicAdd(bad, 1u);
  aff_packed out;
  out.x = fe_pack(fe_canon(x));
  out.y = fe_pack(fe_canon(y));
  tab[idx] = out;
}

static aff_packed pack_point(const haff& p) {
  aff_packed r;
  r.x = fe_pack(fe_canon(p.x));
  r.y = fe_pack(fe_canon(p.y));
  return r;
}

// splitmix64: deterministic offset scalars (any fixed non-zero scalars work; see file header)
static uint64_t splitmix(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Builds one table group on the device: `nseg` scalars (1 for EC_GEN, 2 for Pedersen), each with
// 252 per-bit points `bit_pts[seg*252 + j]`, offsets from multiples of `off_base` that sum to
// `target` (may be infinity).
static int build_group(aff_packed* dev_tab, int nseg, const std::vector<haff>& bit_pts,
                       const haff& off_base, const haff& target, uint64_t seed, int wbits,
                       int nwin) {
  const int nofs = nseg * nwin;
  std::vector<haff> offs(nofs);
  haff sum{FE_ZERO, FE_ZERO, true};
  uint64_t st = seed;
  for (int i = 1; i < nofs; ++i) {
    uint64_t k[4] = {splitmix(st), splitmix(st), splitmix(st), splitmix(st) >> 6};
    offs[i] = h_mul(k, off_base);
    sum = h_add(sum, offs[i]);
  }
  offs[0] = h_add(target, h_neg(sum));
  if (offs[0].inf) {
    set_error("table offsets degenerate");
    return SP_ERR_TABLE_BUILD;
  }
  std::vector<aff_packed> h_bits(bit_pts.size()), h_offs(nofs);
  for (size_t i = 0; i < bit_pts.size(); ++i) h_bits[i] = pack_point(bit_pts[i]);
  for (int i = 0; i < nofs; ++i) h_offs[i] = pack_point(offs[i]);
  aff_packed *d_bits = nullptr, *d_offs = nullptr;
  unsigned* d_bad = nullptr;
  SP_HIP(hipMalloc(&d_bits, h_bits.size() * sizeof(aff_packed)));
  SP_HIP(hipMalloc(&d_offs, h_offs.size() * sizeof(aff_packed)));
  SP_HIP(hipMalloc(&d_bad, sizeof(unsigned)));
  SP_HIP(hipMemcpy(d_bits, h_bits.data(), h_bits.size() * sizeof(aff_packed), hipMemcpyHostToDevice));
  SP_HIP(hipMemcpy(d_offs, h_offs.data(), h_offs.size() * sizeof(aff_packed), hipMemcpyHostToDevice));
  SP_HIP(hipMemset(d_bad, 0, sizeof(unsigned)));
  const fe beta_m = fe_to_mont(fe_unpack(CURVE_BETA));
  const size_t per_seg = (size_t)nwin << wbits;
  for (int seg = 0; seg < nseg; ++seg) {
    const unsigned blocks = (unsigned)((per_seg + 255) / 256);
    hipLaunchKernelGGL(build_table_kernel, dim3(blocks), dim3(256), 0, 0, dev_tab + seg * per_seg,
                       d_bits + seg * 252, d_offs + seg * nwin, wbits, nwin, 252, beta_m, d_bad);
  }
  SP_HIP(hipGetLastError());
  SP_HIP(hipDeviceSynchronize());
  unsigned bad = 0;
  SP_HIP(hipMemcpy(&bad, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost));
  (void)hipFree(d_bits);
  (void)hipFree(d_offs);
  (void)hipFree(d_bad);
  if (bad != 0) {
    set_error("table build produced " + std::to_string(bad) + " off-curve entries");
    return SP_ERR_TABLE_BUILD;
  }
  return SP_OK;
}

static int init_locked(int device, int window_bits) {
  Context& c = g_ctx;
  if (c.ready) return SP_OK;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    set_error(std::string("no HIP device visible (") + hipGetErrorString(e) +
              "); libstarkperp has no CPU fallback");
    return SP_ERR_HIP;
  }
  if (device < 0 || device >= ndev) {
    set_error("device index out of range");
    return SP_ERR_BAD_ARGUMENT;
  }
  if (window_bits == 0) window_bits = 21;  // 12 windows of 21 bits cover the 252-bit scalars exactly
  if (window_bits < 4 || window_bits > 26) {
    set_error("window_bits must be in [4, 26]");
    return SP_ERR_BAD_ARGUMENT;
  }
  SP_HIP(hipSetDevice(device));
  c.device = device;
  c.wbits = window_bits;
  c.nwin = (252 + window_bits - 1) / window_bits;

  // per-bit points: C[2 + 252 e + j] (nothing_up_my_sleeve_gen.py:88-90: 248 doublings of P0/P2,
  // 4 of P1/P3) and 2^j * EC_GEN
  std::vector<haff> ped_bits(504), gen_bits(252);
  const haff bases[4] = {h_make(PT_P0_X, PT_P0_Y), h_make(PT_P1_X, PT_P1_Y),
                         h_make(PT_P2_X, PT_P2_Y), h_make(PT_P3_X, PT_P3_Y)};
  for (int e2 = 0; e2 < 2; ++e2) {
    haff q = bases[2 * e2];
    for (int j = 0; j < 248; ++j) { ped_bits[252 * e2 + j] = q; q = h_dbl(q); }
    q = bases[2 * e2 + 1];
    for (int j = 0; j < 4; ++j) { ped_bits[252 * e2 + 248 + j] = q; q = h_dbl(q); }
  }
  const haff G = h_make(PT_GEN_X, PT_GEN_Y);
  {
    haff q = G;
    for (int j = 0; j < 252; ++j) { gen_bits[j] = q; q = h_dbl(q); }
  }
  const haff shift = h_make(PT_SHIFT_X, PT_SHIFT_Y);
  const haff infinity{FE_ZERO, FE_ZERO, true};

  const size_t per_seg = (size_t)c.nwin << c.wbits;
  SP_HIP(hipMalloc(&c.ped, 2 * per_seg * sizeof(aff_packed)));
  SP_HIP(hipMalloc(&c.gen, per_seg * sizeof(aff_packed)));
  SP_HIP(hipMemset(c.ped, 0, 2 * per_seg * sizeof(aff_packed)));
  SP_HIP(hipMemset(c.gen, 0, per_seg * sizeof(aff_packed)));
  c.table_bytes = 3 * per_seg * sizeof(aff_packed);
  int rc = build_group(c.ped, 2, ped_bits, G, shift, 0x5350454445525345ull, c.wbits, c.nwin);
  if (rc != SP_OK) return rc;
  rc = build_group(c.gen, 1, gen_bits, bases[0], infinity, 0x5350474E54424C45ull, c.wbits, c.nwin);
  if (rc != SP_OK) return rc;
  c.ready = true;
  return SP_OK;
}

}  // namespace sp

using namespace sp;

extern "C" {

int sp_init(int device, int window_bits) {
  ctx_lock lk(g_ctx.mu);
  int rc = init_locked(device, window_bits);
  if (rc != SP_OK && !g_ctx.ready) {
    if (g_ctx.ped) (void)hipFree(g_ctx.ped);
    if (g_ctx.gen) (void)hipFree(g_ctx.gen);
    g_ctx.ped = g_ctx.gen = nullptr;
  }
  return rc;
}

void sp_shutdown(void) {
  ctx_lock lk(g_ctx.mu);
  (void)hipDeviceSynchronize();
  sp::release_pedersen_state();
  sp::release_merkle_state();
  sp::release_stark_state();
  sp::release_ecdsa_state();
  if (g_ctx.ped) (void)hipFree(g_ctx.ped);
  if (g_ctx.gen) (void)hipFree(g_ctx.gen);
  g_ctx.ped = g_ctx.gen = nullptr;
  g_ctx.io.release();
  g_ctx.io2.release();
  g_ctx.ready = false;
}

const char* sp_last_error(void) {
  static thread_local std::string copy;
  std::lock_guard<std::mutex> lk(g_err_mu);
  copy = g_err;
  return copy.c_str();
}

int sp_is_initialised(void) { return g_ctx.ready ? 1 : 0; }
int sp_window_bits(void) { return g_ctx.wbits; }
size_t sp_table_bytes(void) { return g_ctx.table_bytes; }

int sp_synchronize(void* stream) {
  SP_REQUIRE_READY();
  SP_HIP(hipStreamSynchronize((hipStream_t)stream));
  return SP_OK;
}

}  // extern "C"

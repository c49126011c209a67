// sp_init / sp_shutdown and the HBM window-table builder.
//
// Pedersen (signature.py:300-318) is  SHIFT + sum_j b_j C_j  over the 504 per-bit constant points
// C_j of the string b = x || y (C_j = C[2 + j] of the reference's table: doublings of P0 for bits
// 0..247, of P1 for 248..251, of P2, P3 for y).  The device tables use the SIGNED form of that sum:
// with the half points C'_j = C_j / 2 (C'_j = C_{j-1} inside a doubling chain; the four chain heads
// P_i / 2 are computed once with the scalar (N + 1) / 2) and s_j = 2 b_j - 1,
//     SHIFT + sum_j b_j C_j  =  K + sum_j s_j C'_j,      K = SHIFT + sum_j C'_j.
// A window of w + 1 bits then needs only 2^w entries: the entry for the complemented pattern is the
// negated point, so the window's top bit is read as the sign (negating y is free) and the low w bits
// (complemented when the sign is negative) index
//     T_g[v] = C'_top - sum_{b<w} C'_{s+b} + sum_{b : bit b of v set} C_{s+b}.
// Window 0 is unsigned and carries K:  T_0[v] = SHIFT + sum_{j >= w0} C'_j + sum_{b in v} C_b.
// With 2^27-entry windows (163 GB of the 288 GB) 504 = 28 + 17 * 28 bits are 18 entries per hash;
// the default 2^21 (4.3 GiB) gives 20 + 22 * 22 -> 23 entries.  No entry is the point at infinity
// (a signed window is an odd multiple of half a chain head; window 0 carries SHIFT), and a partial
// sum meets the next entry only through a relation between independent generators or between
// windows of different 2-adic weight - never.  The builder checks every entry against the curve.
// The EC_GEN table (k*G for sign / public keys / the z*G leg of verify) is the plain unsigned
// uniform-window table with offsets that are multiples of P0 and sum to the point at infinity.
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "context.hpp"
#include "hostcurve.hpp"

namespace sp {

void release_pedersen_state();  // per-stream scratch, profiling events (pedersen.hip)
void release_merkle_state();    // sparse-update staging (merkle.hip)
void release_stark_state();     // twiddle / coset tables, work buffers (stark.hip)
void release_builtin_state();   // work buffers of the builtin segments (builtins.hip)
void release_ecdsa_state();     // per-signature window tables (ecdsa.hip)
void release_tree_state();      // persistent sparse trees (merkle.hip)

std::recursive_mutex& global_mu() {
  static std::recursive_mutex mu;
  return mu;
}
static Context g_ctxs[SP_MAX_CONTEXTS];
static int g_nctx = 1;
static thread_local int tl_ctx = 0;
static std::string g_err;
static std::mutex g_err_mu;

Context& ctx() { return g_ctxs[tl_ctx]; }
Context& ctx_at(int index) { return g_ctxs[index]; }
int ctx_count() { return g_nctx; }
int ctx_current() { return tl_ctx; }
void ctx_select(int index) { tl_ctx = (index >= 0 && index < g_nctx) ? index : 0; }

CtxByPointer::CtxByPointer(const void* device_ptr) : previous(tl_ctx) {
  if (g_nctx <= 1 || device_ptr == nullptr) return;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, device_ptr) != hipSuccess) {
    (void)hipGetLastError();  // not a device pointer: the entry point will fail on its own terms
    return;
  }
  if (g_ctxs[tl_ctx].ready && g_ctxs[tl_ctx].device == attr.device) return;  // e.g. a lane's own staging buffer
  for (int i = 0; i < g_nctx; ++i) {
    if (g_ctxs[i].ready && g_ctxs[i].device == attr.device) {
      tl_ctx = i;
      return;
    }
  }
}

static HostLane g_lanes[HOST_LANES];
static std::mutex g_lane_mu;
static std::condition_variable g_lane_cv;
static int g_ctx_next = 0;  // round-robin over CONTEXTS: consecutive callers get lanes of consecutive devices

// Within a context the LOWEST free lane is taken: a single-threaded caller then stays on one stream (cycling over
// sixteen streams made every one-item call pay a hardware-queue switch: 0.09 -> 0.3 ms per scalar hash).
HostLane* lane_acquire(int* ctx_index, int want_ctx) {
  std::unique_lock<std::mutex> lk(g_lane_mu);
  int got = -1;
  g_lane_cv.wait(lk, [&] {
    for (int c = 0; c < g_nctx && got < 0; ++c) {
      const int ctx_i = want_ctx >= 0 ? want_ctx : (g_ctx_next + c) % g_nctx;
      for (int i = ctx_i; i < HOST_LANES; i += g_nctx) {  // lanes of context ctx_i: i mod g_nctx == ctx_i
        if (!g_lanes[i].busy) {
          got = i;
          break;
        }
      }
      if (want_ctx >= 0) break;
    }
    return got >= 0;
  });
  if (want_ctx < 0) g_ctx_next = (got % g_nctx + 1) % g_nctx;
  g_lanes[got].busy = true;
  *ctx_index = got % g_nctx;
  return &g_lanes[got];
}

static thread_local int tl_shard_ctx = -1;
int shard_context() { return tl_shard_ctx; }

int shard_over_contexts(size_t n, const std::function<int(size_t, size_t)>& fn) {
  const int parts = g_nctx;
  std::vector<int> rc(parts, SP_OK);
  std::vector<std::thread> workers;
  const size_t per = (n + parts - 1) / parts;
  for (int i = 0; i < parts; ++i) {
    const size_t off = (size_t)i * per;
    if (off >= n) break;
    const size_t cnt = n - off < per ? n - off : per;
    workers.emplace_back([&, i, off, cnt] {
      tl_shard_ctx = i;
      rc[i] = fn(off, cnt);
      tl_shard_ctx = -1;
    });
  }
  for (std::thread& w : workers) w.join();
  for (int v : rc) {
    if (v != SP_OK) return v;
  }
  return SP_OK;
}

void lane_drain(HostLane* lane) {
  if (lane->stream) (void)hipStreamSynchronize(lane->stream);
}

void lane_release(HostLane* lane) {
  {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    lane->busy = false;
  }
  g_lane_cv.notify_all();  // waiters may want a lane of one particular context
}

// Called with the lane's context selected and its device bound (after SP_REQUIRE_READY).
int lane_stream(HostLane* lane) {
  if (lane->stream && lane->stream_ctx == tl_ctx) {
    std::lock_guard<std::recursive_mutex> lk(global_mu());
    ++g_ctxs[tl_ctx].host_calls;
    return SP_OK;
  }
  if (lane->stream) {  // the context layout changed (sp_shutdown + another sp_init_devices)
    (void)hipStreamDestroy(lane->stream);
    lane->stream = nullptr;
    lane->io.release();
  }
  const hipError_t e = hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    lane->stream = nullptr;
    return hip_fail(e, "hipStreamCreateWithFlags (host lane)");
  }
  lane->stream_ctx = tl_ctx;
  std::lock_guard<std::recursive_mutex> lk(global_mu());
  ++g_ctxs[tl_ctx].host_calls;
  return SP_OK;
}

void release_host_lanes() {
  std::lock_guard<std::mutex> lk(g_lane_mu);
  for (HostLane& l : g_lanes) {
    if (l.stream) {
      (void)hipStreamSynchronize(l.stream);
      (void)hipStreamDestroy(l.stream);
      l.stream = nullptr;
    }
    l.io.release();
    l.hio.release();
    l.busy = false;
  }
}
void set_error(const std::string& s) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  g_err = s;
}
int hip_fail(hipError_t e, const char* what) {
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return SP_ERR_HIP;
}

struct HostTimeline {
  const char* call;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  std::mutex mu;
  std::vector<std::pair<std::string, double>> marks;
};
static std::atomic<HostTimeline*> g_timeline{nullptr};
static const bool g_timeline_on = [] {
  const char* e = getenv("STARKPERP_TIMELINE");
  return e != nullptr && e[0] == '1';
}();
void tl_mark(const char* what) {
  HostTimeline* t = g_timeline.load(std::memory_order_acquire);
  if (!t) return;
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t->t0).count();
  std::lock_guard<std::mutex> lk(t->mu);
  char tid[32];
  snprintf(tid, sizeof(tid), " [t%04x]", (unsigned)(std::hash<std::thread::id>()(std::this_thread::get_id()) & 0xffff));
  t->marks.emplace_back(std::string(what) + tid, us);
}
TimelineScope::TimelineScope(const char* call) {
  if (!g_timeline_on) return;
  HostTimeline* expected = nullptr;
  HostTimeline* t = new HostTimeline{call};
  if (g_timeline.compare_exchange_strong(expected, t)) mine = t;  // one traced call at a time; nested / concurrent calls are not traced
  else delete t;
}
TimelineScope::~TimelineScope() {
  if (!mine) return;
  tl_mark("end");
  g_timeline.store(nullptr, std::memory_order_release);
  std::string line = std::string("libstarkperp timeline ") + mine->call + ":";
  char buf[96];
  for (const auto& m : mine->marks) {
    snprintf(buf, sizeof(buf), " %.1f us ", m.second);
    line += buf + m.first + ";";
  }
  fprintf(stderr, "%s\n", line.c_str());
  // not deleted: a thread of ANOTHER call may still be inside tl_mark with this pointer (diagnostics mode only: a few
  // hundred bytes per traced call)
}

// One thread per entry of ONE window: tab[v] = off + sum_{b < nb : bit b of v set} bits[b].
__global__ void __launch_bounds__(256)
build_window_kernel(aff_packed* tab, const aff_packed* bits, const aff_packed* off, int nb, fe beta_m,
                    unsigned* bad) {
  const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >> nb) return;
  xyzz acc = xyzz_from_aff(ld_aff(off));
  for (int b = 0; b < nb; ++b) {
    if ((v >> b) & 1u) acc = xyzz_madd(acc, ld_aff(bits + b));
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
  tab[v] = out;
}

// Round 4: the doubling schedule of the table build.  tab[v | 2^b] = tab[v] + bits[b] for v < 2^b: ONE affine
// addition per new entry instead of one per set bit (13 on average at 26 bits), the division shared by the EXTEND_K
// entries of a thread (Montgomery's trick around one lane-private inversion).  Pass b reads entries
// [0, 2^b) and writes [2^b, 2^(b+1)): passes of one window run one after the other on the stream.  An exceptional
// addition (x1 = x2: impossible for these points, context.hip header) would zero the shared inverse and leave the
// thread's entries off the curve - the on-curve check below reports it like build_window_kernel does.
// 26-bit windows: 1.1 s -> 0.12 - 0.13 s per process (profiles/r04_table_build.txt).
constexpr int EXTEND_K = 8;
__global__ void __launch_bounds__(256)
extend_window_kernel(aff_packed* tab, const aff_packed* bit, int b, fe beta_m, unsigned* bad) {
  const size_t half = (size_t)1 << b;
  const size_t stride = half / EXTEND_K;  // entry k of thread t is t + k * stride: the lanes of a wave touch consecutive entries
  const size_t v0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v0 >= stride) return;
  const aff q = ld_aff(bit);
  fe prefix[EXTEND_K];
  fe run = FE_ONE_M;
#pragma unroll
  for (int k = 0; k < EXTEND_K; ++k) {
    const fe px = fe_unpack(ld_u256((const uint64_t*)&tab[v0 + k * stride].x));
    prefix[k] = run;  // product of the denominators before entry k
    run = fe_mul(run, fe_sub(q.x, px));  // B = 1 signed operand
  }
  fe inv = fe_inv(run);
#pragma unroll
  for (int k = EXTEND_K - 1; k >= 0; --k) {
    const aff p1 = ld_aff(tab + v0 + k * stride);
    const fe dx = fe_sub(q.x, p1.x);
    const fe idx = fe_mul(inv, prefix[k]);  // 1 / dx_k
    inv = fe_mul(inv, dx);
    const fe lam = fe_mul(fe_sub(q.y, p1.y), idx);
    const fe x3 = fe_carry(fe_sub(fe_sub(fe_sqr(lam), p1.x), q.x));
    const fe y3 = fe_carry(fe_sub(fe_mul(lam, fe_sub(p1.x, x3)), p1.y));
    const fe lhs = fe_sqr(y3);
    const fe rhs = fe_carry(fe_add(fe_add(fe_mul(fe_sqr(x3), x3), x3), beta_m));
    if (!fe_eq(lhs, rhs)) atomicAdd(bad, 1u);
    aff_packed out;
    out.x = fe_pack(fe_canon(x3));
    out.y = fe_pack(fe_canon(y3));
    tab[half + v0 + k * stride] = out;
  }
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

// Builds windows on the device: window g = `nb[g]` per-bit points starting at bit_pts[first[g]],
// offset point offs[g], entries written at dev_tab + base[g].
static int build_windows(aff_packed* dev_tab, const std::vector<haff>& bit_pts, const std::vector<haff>& offs,
                         const std::vector<int>& first, const std::vector<int>& nb,
                         const std::vector<uint64_t>& base) {
  const size_t nw = offs.size();
  std::vector<aff_packed> h_bits(bit_pts.size()), h_offs(nw);
  for (size_t i = 0; i < bit_pts.size(); ++i) h_bits[i] = pack_point(bit_pts[i]);
  for (size_t i = 0; i < nw; ++i) {
    if (offs[i].inf) {
      set_error("table offsets degenerate");
      return SP_ERR_TABLE_BUILD;
    }
    h_offs[i] = pack_point(offs[i]);
  }
  aff_packed *d_bits = nullptr, *d_offs = nullptr;
  unsigned* d_bad = nullptr;
  SP_HIP(hipMalloc(&d_bits, h_bits.size() * sizeof(aff_packed)));
  SP_HIP(hipMalloc(&d_offs, h_offs.size() * sizeof(aff_packed)));
  SP_HIP(hipMalloc(&d_bad, sizeof(unsigned)));
  SP_HIP(hipMemcpy(d_bits, h_bits.data(), h_bits.size() * sizeof(aff_packed), hipMemcpyHostToDevice));
  SP_HIP(hipMemcpy(d_offs, h_offs.data(), h_offs.size() * sizeof(aff_packed), hipMemcpyHostToDevice));
  SP_HIP(hipMemset(d_bad, 0, sizeof(unsigned)));
  const fe beta_m = fe_to_mont(fe_unpack(CURVE_BETA));
  // the first 2^13 entries of a window from their set bits, every further bit by one doubling pass
  // (STARKPERP_TABLE_BUILD=direct: every entry from its set bits, the round 1 - 3 build, kept for A/B and as the
  // independent construction tests/test_gpu_pedersen.py compares the tables of the two builds through)
  static const bool direct = getenv("STARKPERP_TABLE_BUILD") && !strcmp(getenv("STARKPERP_TABLE_BUILD"), "direct");
  for (size_t g = 0; g < nw; ++g) {
    const int seed_bits = direct || nb[g] < 13 ? nb[g] : 13;
    const size_t count = (size_t)1 << seed_bits;
    hipLaunchKernelGGL(build_window_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0,
                       dev_tab + base[g], d_bits + first[g], d_offs + g, seed_bits, beta_m, d_bad);
    for (int b = seed_bits; b < nb[g]; ++b) {
      const size_t threads = ((size_t)1 << b) / EXTEND_K;
      hipLaunchKernelGGL(extend_window_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0,
                         dev_tab + base[g], d_bits + first[g] + b, b, beta_m, d_bad);
    }
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

// EC_GEN table: uniform unsigned windows, offsets = random multiples of `off_base` summing to infinity.
static int build_gen_table(aff_packed* dev_tab, const std::vector<haff>& bit_pts, const haff& off_base,
                           uint64_t seed, int wbits, int nwin) {
  std::vector<haff> offs(nwin);
  haff sum{FE_ZERO, FE_ZERO, true};
  uint64_t st = seed;
  for (int i = 1; i < nwin; ++i) {
    uint64_t k[4] = {splitmix(st), splitmix(st), splitmix(st), splitmix(st) >> 6};
    offs[i] = h_mul(k, off_base);
    sum = h_add(sum, offs[i]);
  }
  offs[0] = h_neg(sum);
  std::vector<int> first(nwin), nb(nwin);
  std::vector<uint64_t> base(nwin);
  for (int i = 0; i < nwin; ++i) {
    first[i] = i * wbits;
    nb[i] = 252 - i * wbits < wbits ? 252 - i * wbits : wbits;
    base[i] = (uint64_t)i << wbits;
  }
  return build_windows(dev_tab, bit_pts, offs, first, nb, base);
}

// Window plan for 2^log2e-entry signed windows: as many (log2e + 1)-bit signed windows as fit below
// bit 504, the remaining low bits (at least one) as the unsigned window 0.
static int make_plan(int log2e, PedPlan& p) {
  const int sw = log2e + 1;
  int k = 504 / sw;
  int w0 = 504 - k * sw;
  if (w0 == 0) { --k; w0 = sw; }
  if (k + 1 > PED_MAX_WINDOWS) {
    set_error("window_bits too small for the window plan");
    return SP_ERR_BAD_ARGUMENT;
  }
  p.nwin = k + 1;
  p.log2e = log2e;
  p.start[0] = 0;
  p.bits[0] = (uint8_t)w0;
  p.base[0] = 0;
  uint64_t next = (uint64_t)1 << w0;
  for (int g = 1; g <= k; ++g) {
    p.start[g] = (uint16_t)(w0 + (g - 1) * sw);
    p.bits[g] = (uint8_t)sw;
    p.base[g] = next;
    next += (uint64_t)1 << log2e;
  }
  p.entries = next;
  return SP_OK;
}

// The HIP runtime sets something up for sizeable asynchronous copies ONCE per process, some tens of copies in: the
// ~43rd 128-KiB hipMemcpyAsync of a process stalls its caller for 6 - 7 ms (profiles/r06_copy_path_warmup_ubench.txt:
// reproduced with torch tensors alone).  An exchange's first order batches met it around their sixth call, inside
// sp_order_batch's tree update.  Initialisation already costs 0.1 s and more: run those copies here (64 x 128 KiB each
// way through a page-locked buffer, ~8 ms, best effort - every failure is ignored).  STARKPERP_NO_COPY_WARMUP=1 skips it.
static void warm_copy_path() {
  static bool done = false;  // once per process (init_locked runs under the library lock)
  if (done || getenv("STARKPERP_NO_COPY_WARMUP")) return;
  done = true;
  const size_t bytes = 128 << 10;
  void *h = nullptr, *d = nullptr;
  if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return; }
  if (hipMalloc(&d, bytes) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
  memset(h, 0, bytes);
  for (int i = 0; i < 64; ++i) {
    if (hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, 0) != hipSuccess) break;
    if (hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, 0) != hipSuccess) break;
    if ((i & 15) == 15 && hipStreamSynchronize(0) != hipSuccess) break;
  }
  (void)hipStreamSynchronize(0);
  (void)hipGetLastError();
  (void)hipFree(d);
  (void)hipHostFree(h);
}

static int init_locked(Context& c, int device, int window_bits) {
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
  if (window_bits == 0) window_bits = 21;  // 2^21-entry windows: 23 entries per hash, 4.3 GiB
  if (window_bits < 4 || window_bits > 27) {
    set_error("window_bits must be in [4, 27]");
    return SP_ERR_BAD_ARGUMENT;
  }
  SP_HIP(hipSetDevice(device));
  c.device = device;
  int rc = make_plan(window_bits, c.plan);
  if (rc != SP_OK) return rc;
  c.wbits = window_bits < 22 ? window_bits : 22;  // EC_GEN table: 12 windows are plenty for ECDSA
  c.nwin = (252 + c.wbits - 1) / c.wbits;

  // per-bit points C_j of x || y (nothing_up_my_sleeve_gen.py:88-90: 248 doublings of P0/P2, 4 of
  // P1/P3), their halves C'_j, and 2^j * EC_GEN
  std::vector<haff> ped_bits(504), half_bits(504), gen_bits(252);
  const haff bases[4] = {h_make(PT_P0_X, PT_P0_Y), h_make(PT_P1_X, PT_P1_Y),
                         h_make(PT_P2_X, PT_P2_Y), h_make(PT_P3_X, PT_P3_Y)};
  uint64_t half_n[4];
  {  // (N + 1) / 2: halving in the (odd, prime order) group
    uint64_t n[4];
    for (int i = 0; i < 4; ++i) n[i] = (uint64_t)U256_N.w[2 * i] | ((uint64_t)U256_N.w[2 * i + 1] << 32);
    unsigned carry = 1;  // N + 1
    for (int i = 0; i < 4; ++i) {
      const uint64_t v = n[i] + carry;
      carry = (carry && v == 0) ? 1 : 0;
      n[i] = v;
    }
    for (int i = 0; i < 4; ++i) half_n[i] = (n[i] >> 1) | (i < 3 ? n[i + 1] << 63 : 0);
  }
  for (int e2 = 0; e2 < 2; ++e2) {
    for (int part = 0; part < 2; ++part) {
      const int first = 252 * e2 + (part ? 248 : 0), count = part ? 4 : 248;
      const haff head = bases[2 * e2 + part];
      const haff half_head = h_mul(half_n, head);
      if (half_head.inf || !fe_eq(h_dbl(half_head).x, head.x)) {
        set_error("halving a chain head failed");
        return SP_ERR_TABLE_BUILD;
      }
      haff q = head;
      for (int j = 0; j < count; ++j) {
        ped_bits[first + j] = q;
        half_bits[first + j] = j == 0 ? half_head : ped_bits[first + j - 1];
        q = h_dbl(q);
      }
    }
  }
  const haff G = h_make(PT_GEN_X, PT_GEN_Y);
  {
    haff q = G;
    for (int j = 0; j < 252; ++j) { gen_bits[j] = q; q = h_dbl(q); }
  }
  const haff shift = h_make(PT_SHIFT_X, PT_SHIFT_Y);

  // window offsets: O_0 = SHIFT + sum_{j >= w0} C'_j;  O_g = C'_top - sum_{b < log2e} C'_{s+b}
  const PedPlan& p = c.plan;
  std::vector<haff> offs(p.nwin);
  std::vector<int> first(p.nwin), nb(p.nwin);
  std::vector<uint64_t> base(p.nwin);
  {
    haff o = shift;
    for (int j = p.bits[0]; j < 504; ++j) o = h_add(o, half_bits[j]);
    offs[0] = o;
    first[0] = 0;
    nb[0] = p.bits[0];
    base[0] = 0;
  }
  for (int g = 1; g < p.nwin; ++g) {
    const int s0 = p.start[g];
    haff o = half_bits[s0 + p.log2e];
    for (int b = 0; b < p.log2e; ++b) o = h_add(o, h_neg(half_bits[s0 + b]));
    offs[g] = o;
    first[g] = s0;
    nb[g] = p.log2e;
    base[g] = p.base[g];
  }

  const size_t gen_entries = (size_t)c.nwin << c.wbits;
  // Experiment switch: STARKPERP_CONTIGUOUS_TABLES=1 asks for physically contiguous tables
  // (hipDeviceMallocContiguous) - larger page-table fragments for the random 64-byte gathers - and falls back to
  // the ordinary allocation when the driver cannot give that much contiguous memory.
  c.ped = nullptr;
  if (const char* contig = getenv("STARKPERP_CONTIGUOUS_TABLES")) {
    if (contig[0] == '1') {
      void* ptr = nullptr;
      if (hipExtMallocWithFlags(&ptr, p.entries * sizeof(aff_packed), hipDeviceMallocContiguous) == hipSuccess) {
        c.ped = (aff_packed*)ptr;
      } else {
        (void)hipGetLastError();
        fprintf(stderr, "libstarkperp: no contiguous allocation of %zu bytes, using hipMalloc\n",
                (size_t)(p.entries * sizeof(aff_packed)));
      }
    }
  }
  if (!c.ped) SP_HIP(hipMalloc(&c.ped, p.entries * sizeof(aff_packed)));
  SP_HIP(hipMalloc(&c.gen, gen_entries * sizeof(aff_packed)));
  SP_HIP(hipMemset(c.gen, 0, gen_entries * sizeof(aff_packed)));
  SP_HIP(hipMalloc(&c.d_plan, sizeof(PedPlan)));
  SP_HIP(hipMemcpy(c.d_plan, &c.plan, sizeof(PedPlan), hipMemcpyHostToDevice));
  c.table_bytes = (p.entries + gen_entries) * sizeof(aff_packed);
  rc = build_windows(c.ped, ped_bits, offs, first, nb, base);
  if (rc != SP_OK) return rc;
  rc = build_gen_table(c.gen, gen_bits, bases[0], 0x5350474E54424C45ull, c.wbits, c.nwin);
  if (rc != SP_OK) return rc;
  if (const char* masked = getenv("STARKPERP_SIGN_MASKED")) {
    if (masked[0] == '1') {  // the small table of the masked signer: same builder, 4-bit windows, its own offsets
      c.gen_masked = nullptr;
      SP_HIP(hipMalloc(&c.gen_masked, (size_t)63 * 16 * sizeof(aff_packed)));
      rc = build_gen_table(c.gen_masked, gen_bits, bases[0], 0x4D41534B45444745ull, 4, 63);
      if (rc != SP_OK) {  // never leave a half-built masked table behind, whoever the caller is
        (void)hipFree(c.gen_masked);
        c.gen_masked = nullptr;
        return rc;
      }
      c.table_bytes += (size_t)63 * 16 * sizeof(aff_packed);
    }
  }
  warm_copy_path();
  c.ready = true;
  return SP_OK;
}

}  // namespace sp

using namespace sp;

extern "C" {

static void free_tables(Context& c) {
  if (c.ped) (void)hipFree(c.ped);
  if (c.gen) (void)hipFree(c.gen);
  if (c.gen_masked) (void)hipFree(c.gen_masked);
  if (c.d_plan) (void)hipFree(c.d_plan);
  c.ped = c.gen = c.gen_masked = nullptr;
  c.d_plan = nullptr;
  c.io.release();
  c.io2.release();
  c.ready = false;
}

int sp_init(int device, int window_bits) {
  ctx_lock lk(global_mu());
  Context& c = ctx_at(0);
  if (c.ready) return SP_OK;  // idempotent (also after sp_init_devices: the contexts stay as they are)
  int rc = init_locked(c, device, window_bits);
  if (rc != SP_OK) free_tables(c);
  return rc;
}

int sp_init_devices(int n_devices, const int* device_ids, int window_bits) {
  if (n_devices < 1 || n_devices > SP_MAX_CONTEXTS || device_ids == nullptr) {
    set_error("sp_init_devices: 1 .. 16 devices");
    return SP_ERR_BAD_ARGUMENT;
  }
  ctx_lock lk(global_mu());
  if (ctx_at(0).ready) {  // idempotent for the same layout only: tables are immutable after init
    bool same = g_nctx == n_devices;
    for (int i = 0; same && i < n_devices; ++i) same = ctx_at(i).ready && ctx_at(i).device == device_ids[i];
    if (same) return SP_OK;
    set_error("sp_init_devices: already initialised with another device layout (sp_shutdown first)");
    return SP_ERR_BAD_ARGUMENT;
  }
  int previous = -1;
  (void)hipGetDevice(&previous);
  int rc = SP_OK;
  for (int i = 0; i < n_devices && rc == SP_OK; ++i) rc = init_locked(ctx_at(i), device_ids[i], window_bits);
  if (rc != SP_OK) {
    for (int i = 0; i < n_devices; ++i) free_tables(ctx_at(i));
  } else {
    g_nctx = n_devices;
  }
  if (previous >= 0) (void)hipSetDevice(previous);
  return rc;
}

int sp_device_count(void) { return ctx_at(0).ready ? g_nctx : 0; }

int sp_context_info(int index, int* device, uint64_t* host_calls) {
  ctx_lock lk(global_mu());
  if (index < 0 || index >= g_nctx || !ctx_at(index).ready) {
    set_error("sp_context_info: no such context");
    return SP_ERR_BAD_ARGUMENT;
  }
  if (device) *device = ctx_at(index).device;
  if (host_calls) *host_calls = ctx_at(index).host_calls;
  return SP_OK;
}

void sp_shutdown(void) {
  ctx_lock lk(global_mu());
  int previous = -1;
  (void)hipGetDevice(&previous);
  for (int i = 0; i < g_nctx; ++i) {
    if (ctx_at(i).ready && hipSetDevice(ctx_at(i).device) == hipSuccess) (void)hipDeviceSynchronize();
  }
  if (previous >= 0) (void)hipSetDevice(previous);
  sp::release_pedersen_state();
  sp::release_merkle_state();
  sp::release_stark_state();
  sp::release_builtin_state();
  sp::release_ecdsa_state();
  sp::release_tree_state();
  sp::release_host_lanes();
  for (int i = 0; i < SP_MAX_CONTEXTS; ++i) {
    free_tables(ctx_at(i));
    ctx_at(i).host_calls = 0;
  }
  g_nctx = 1;
}

const char* sp_last_error(void) {
  static thread_local std::string copy;
  std::lock_guard<std::mutex> lk(g_err_mu);
  copy = g_err;
  return copy.c_str();
}

int sp_is_initialised(void) { return ctx_at(0).ready ? 1 : 0; }
int sp_window_bits(void) { return ctx_at(0).plan.log2e; }
size_t sp_table_bytes(void) {  // all contexts
  size_t total = 0;
  for (int i = 0; i < g_nctx; ++i) total += ctx_at(i).table_bytes;
  return total;
}

#ifndef SP_OFFLOAD_ARCH
#error "SP_OFFLOAD_ARCH must come from the Makefile (-DSP_OFFLOAD_ARCH='\"$(ARCH)\"'): sp_build_info reports what was compiled"
#endif
#ifndef SP_SANITIZER
#define SP_SANITIZER "none"
#endif
#define SP_STR2(x) #x
#define SP_STR(x) SP_STR2(x)
const char* sp_build_info(void) {
  return "libstarkperp; compiler: clang " __clang_version__ "; HIP " SP_STR(HIP_VERSION_MAJOR) "." SP_STR(HIP_VERSION_MINOR) "." SP_STR(
      HIP_VERSION_PATCH) "; offload-arch: " SP_OFFLOAD_ARCH "; compiled: " __DATE__ " " __TIME__ "; sanitizer: " SP_SANITIZER;
}

int sp_synchronize(void* stream) {
  SP_REQUIRE_READY();
  SP_HIP(hipStreamSynchronize((hipStream_t)stream));
  return SP_OK;
}

}  // extern "C"

// Library-wide state of libstarkperp: the selected device, the HBM-resident window tables and
// growable scratch.  One context per process (one process per GPU).
#pragma once
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <utility>
#include <string>

#include "../../include/starkperp.h"
#include "curve.hpp"

namespace sp {

// 64-byte table entry: affine point, Montgomery form, canonical 256-bit packing.
struct alignas(64) aff_packed {
  u256 x, y;
};

// Window plan of the Pedersen tables over the 504-bit string  x (252 bits) || y (252 bits).
// Window 0 is unsigned: `bits[0]` bits, 2^bits[0] entries.  Every other window is SIGNED: it covers
// log2e + 1 bits with 2^log2e entries - the top bit of the window is the sign of the whole entry
// (see context.hip).  Windows may straddle the x | y boundary.
constexpr int PED_MAX_WINDOWS = 128;
struct PedPlan {
  int nwin = 0;
  int log2e = 0;                       // entries per signed window = 2^log2e  (what sp_window_bits reports)
  uint16_t start[PED_MAX_WINDOWS];     // first bit of window g in the 504-bit string
  uint8_t bits[PED_MAX_WINDOWS];       // bits the window consumes
  uint64_t base[PED_MAX_WINDOWS];      // index of the window's first entry in the table
  uint64_t entries = 0;                // total
};

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  // Grow-only; contents are not preserved.
  hipError_t reserve(size_t need) {
    if (need <= bytes) return hipSuccess;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
    size_t want = need + need / 4;
    hipError_t e = hipMalloc(&ptr, want);
    if (e == hipSuccess) bytes = want;
    return e;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
};

// Page-locked host memory for staging SMALL host-pointer batches (round 6).  A hipMemcpyAsync from pageable memory
// makes the runtime pin / unpin the caller's pages around every copy; when two host threads of one call do that at
// the same time - sp_order_batch's tree update and its verifier thread - one of them was seen to stall for 6 - 11 ms
// (profiles/r06_c3_host_timeline.txt).  Staged copies (memcpy into this buffer, then a truly asynchronous DMA) have no
// such path.  Batches above PINNED_STAGE_MAX keep the direct copy: they are throughput-bound, and page-locking hundreds
// of megabytes per lane would cost more than it saves.
constexpr size_t PINNED_STAGE_MAX = (size_t)4 << 20;
struct PinnedBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  hipError_t reserve(size_t need) {  // grow-only; contents are not preserved
    if (need <= bytes) return hipSuccess;
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    bytes = 0;
    size_t want = need + need / 4;
    hipError_t e = hipHostMalloc(&ptr, want, hipHostMallocDefault);
    if (e == hipSuccess) bytes = want;
    return e;
  }
  void release() {
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
};

// One context per device the process drives (sp_init: one; sp_init_devices: several).  Context 0 is the
// PRIMARY: the stateful machinery - persistent trees, the ECDSA key cache, the prover's twiddle tables and
// witness scratch - lives on its device only.  The stateless batches (hash, ladder verification, signing,
// public keys, tree / forest rebuilds, row commitments) run on any context: a host-pointer call takes the
// context of the host lane it was given, a _dev call the context of the device its pointers live on.
constexpr int SP_MAX_CONTEXTS = 16;
std::recursive_mutex& global_mu();
struct Context {
  bool ready = false;
  int device = -1;
  int wbits = 16;        // EC_GEN table: window width
  int nwin = 16;         // EC_GEN table: windows per 252-bit scalar = ceil(252 / wbits)
  aff_packed* ped = nullptr;   // Pedersen window tables, laid out by `plan`
  aff_packed* gen = nullptr;   // [nwin][1 << wbits]     fixed-base EC_GEN
  // STARKPERP_SIGN_MASKED=1: a second EC_GEN table of 63 unsigned 4-bit windows (63 x 16 entries = 63 KiB) for the
  // signers and the key derivation: every window reads all 16 of its entries and keeps one by a mask, so no address
  // depends on the nonce or the private key (ecdsa.hip gen_mul_masked; include/starkperp.h "threat model")
  aff_packed* gen_masked = nullptr;
  // the fixed-base table the SECRET-scalar kernels are handed: wbits < 0 selects the masked walk
  const aff_packed* secret_gen() const { return gen_masked ? gen_masked : gen; }
  int secret_wbits() const { return gen_masked ? -4 : wbits; }
  int secret_nwin() const { return gen_masked ? 63 : nwin; }
  PedPlan plan;                // host copy
  PedPlan* d_plan = nullptr;   // device copy the kernels read (uniform loads)
  size_t table_bytes = 0;
  DeviceBuffer io;             // staging for host-pointer entry points
  DeviceBuffer io2;
  uint64_t host_calls = 0;     // host-lane calls served (sp_context_info)
  // Recursive: host-pointer entry points hold it across the staging copies AND the nested _dev
  // call, so two host threads can never interleave on the shared staging buffers.  ONE lock for all
  // contexts: it also guards the per-stream scratch maps, which are shared; it is held while work is
  // enqueued, never while the device runs a stateless batch.
  std::recursive_mutex& mu = global_mu();
};
using ctx_lock = std::lock_guard<std::recursive_mutex>;

Context& ctx();               // the context selected on this host thread (the primary unless a scope below says otherwise)
Context& ctx_at(int index);
int ctx_count();
int ctx_current();
void ctx_select(int index);
// Key of the per-stream scratch maps: the null stream exists once per device.
using StreamKey = std::pair<int, hipStream_t>;
inline StreamKey stream_key(hipStream_t st) { return StreamKey(ctx_current(), st); }

// Selects, for the duration of a _dev entry point, the context whose device owns `device_ptr`.  With one
// context (the common case: one process per GPU) it does nothing.
struct CtxByPointer {
  int previous;
  explicit CtxByPointer(const void* device_ptr);
  ~CtxByPointer() { ctx_select(previous); }
  CtxByPointer(const CtxByPointer&) = delete;
  CtxByPointer& operator=(const CtxByPointer&) = delete;
};

// Host lanes: the host-pointer entry points that carry no shared state (hash / verify / sign / public-key
// batches) each take one of HOST_LANES lanes - a non-blocking stream with its own staging buffer - so that
// calls from different host threads overlap on the device instead of queueing behind one lock (the scalar
// API of the reference is a stream of one-item calls: 0.1 - 0.4 ms of latency each, almost all of it idle
// chip).  The context lock is then held only while a kernel is enqueued.  A caller blocks while every lane is
// taken.  Lane i belongs to context i mod ctx_count(), and lanes are handed out round-robin: concurrent
// callers spread over the devices of sp_init_devices.
constexpr int HOST_LANES = 16;
struct HostLane {
  hipStream_t stream = nullptr;
  int stream_ctx = -1;          // the context the stream was created for
  DeviceBuffer io;
  PinnedBuffer hio;             // host-side staging of small batches (see PinnedBuffer)
  bool busy = false;
};
HostLane* lane_acquire(int* ctx_index, int want_ctx);  // want_ctx < 0: any context, round-robin
void lane_release(HostLane* lane);
void lane_drain(HostLane* lane);  // waits for whatever is still queued on the lane's stream
int lane_stream(HostLane* lane);  // creates the lane's stream on the current device if needed; SP_OK or SP_ERR_HIP
void release_host_lanes();        // sp_shutdown
// ecdsa.hip: keyed verification of a host batch; `before_lock` runs once, after the lock-free part and before the
// library lock is taken (also when there is nothing to verify).
int verify_batch_keyed_gated(const uint64_t* z, const uint64_t* r, const uint64_t* s, const uint64_t* qx,
                             const uint64_t* qy, uint8_t* result, size_t n, const std::function<void()>& before_lock);
// Usage in an entry point:  LaneScope ls;  SP_REQUIRE_READY();  if (ls.open() != SP_OK) return SP_ERR_HIP;
// A large host batch on several contexts: `fn(offset, count)` runs once per context on its own host thread, each
// bound to its context (so the LaneScope inside takes a lane of that context), over contiguous slices of the n
// items; returns the first non-zero code.  Used by the stateless host-pointer batches when ctx_count() > 1.
int shard_over_contexts(size_t n, const std::function<int(size_t, size_t)>& fn);
int shard_context();  // the context this host thread was bound to by shard_over_contexts, or -1
constexpr size_t SHARD_MIN_ITEMS = 16384;  // below this one device is faster than the host threads are to start

struct LaneScope {
  HostLane* lane;
  int previous_ctx;
  LaneScope() : previous_ctx(ctx_current()) {
    int index = 0;
    lane = lane_acquire(&index, shard_context());
    ctx_select(index);
  }
  explicit LaneScope(int want_ctx) : previous_ctx(ctx_current()) {  // a lane of one particular context
    int index = 0;
    lane = lane_acquire(&index, want_ctx);
    ctx_select(index);
  }
  int open() { return lane_stream(lane); }
  ~LaneScope() {
    // An entry point that leaves early (a failed HIP call) may still have copies into the CALLER's buffers
    // or kernels on the lane's staging buffer in flight: drain the stream before the lane - and, on the
    // caller's side, the buffers - can be reused.  After a normal return the stream is idle and this is free.
    lane_drain(lane);
    lane_release(lane);
    ctx_select(previous_ctx);
  }
  LaneScope(const LaneScope&) = delete;
  LaneScope& operator=(const LaneScope&) = delete;
};

void set_error(const std::string& s);

// Host-side timeline of ONE traced API call (diagnostics, STARKPERP_TIMELINE=1; tools/c3_probe.py): marks from any
// thread of the call - the caller's and sp_order_batch's verifier - printed to stderr as one line when the call
// ends.  With the variable unset tl_mark is one relaxed load and a branch.
struct HostTimeline;
void tl_mark(const char* what);
struct TimelineScope {  // opens the timeline for the duration of a call (no-op unless STARKPERP_TIMELINE=1)
  HostTimeline* mine = nullptr;
  explicit TimelineScope(const char* call);
  ~TimelineScope();
  TimelineScope(const TimelineScope&) = delete;
  TimelineScope& operator=(const TimelineScope&) = delete;
};
int hip_fail(hipError_t e, const char* what);

#define SP_HIP(call)                                   \
  do {                                                 \
    hipError_t e__ = (call);                           \
    if (e__ != hipSuccess) return hip_fail(e__, #call); \
  } while (0)

// HIP's current device is per host thread: a thread other than the one that called sp_init would
// otherwise launch on device 0.  Binds the library's device for the duration of an entry point and
// puts the caller's own choice back afterwards.
struct DeviceScope {
  int previous = -1;
  bool switched = false;
  explicit DeviceScope(int device) {
    if (hipGetDevice(&previous) == hipSuccess && previous != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(previous);
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

#define SP_REQUIRE_READY()                                                        \
  if (!ctx().ready) {                                                             \
    set_error("libstarkperp is not initialised (sp_init failed or not called; "   \
              "there is no CPU fallback)");                                       \
    return SP_ERR_NOT_INITIALISED;                                                \
  }                                                                               \
  sp::DeviceScope sp_device_scope__(ctx().device)

// ---- device helpers shared by the kernels ----
__device__ __forceinline__ u256 ld_u256(const uint64_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  u256 r;
  r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w;
  r.w[4] = b.x; r.w[5] = b.y; r.w[6] = b.z; r.w[7] = b.w;
  return r;
}
__device__ __forceinline__ void st_u256(uint64_t* p, const u256& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.w[0], v.w[1], v.w[2], v.w[3]);
  q[1] = make_uint4(v.w[4], v.w[5], v.w[6], v.w[7]);
}
__device__ __forceinline__ aff ld_aff(const aff_packed* e) {
  const uint4* q = reinterpret_cast<const uint4*>(e);
  uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  u256 x, y;
  x.w[0] = a.x; x.w[1] = a.y; x.w[2] = a.z; x.w[3] = a.w;
  x.w[4] = b.x; x.w[5] = b.y; x.w[6] = b.z; x.w[7] = b.w;
  y.w[0] = c.x; y.w[1] = c.y; y.w[2] = c.z; y.w[3] = c.w;
  y.w[4] = d.x; y.w[5] = d.y; y.w[6] = d.z; y.w[7] = d.w;
  aff r;
  r.x = fe_unpack(x);
  r.y = fe_unpack(y);
  return r;
}
// plain integer a < 2^256 compared with p / N / 2^251 on packed words
__host__ __device__ __forceinline__ bool u256_lt(const u256& a, const u256& b) {
  bool lt = false;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (a.w[i] != b.w[i]) lt = a.w[i] < b.w[i];  // highest differing word decides
  }
  return lt;
}
__host__ __device__ __forceinline__ bool u256_is_zero(const u256& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.w[i];
  return o == 0;
}
constexpr u256 U256_P = {{1u, 0u, 0u, 0u, 0u, 0u, 0x11u, 0x08000000u}};
constexpr u256 U256_N = {{0xadc64d2fu, 0x1e66a241u, 0xcae7b232u, 0xb781126du, 0xffffffffu, 0xffffffffu,
                          0x10u, 0x08000000u}};
constexpr u256 U256_2P251 = {{0u, 0u, 0u, 0u, 0u, 0u, 0u, 0x08000000u}};

// w-bit window number `win` of a 256-bit little-endian integer
__device__ __forceinline__ uint32_t window_of(const u256& a, int win, int wbits) {
  const int bit = win * wbits, wi = bit >> 5, sh = bit & 31;
  uint64_t two = (uint64_t)a.w[wi] | ((uint64_t)(wi + 1 < 8 ? a.w[wi + 1] : 0u) << 32);
  return (uint32_t)(two >> sh) & ((1u << wbits) - 1u);
}

}  // namespace sp

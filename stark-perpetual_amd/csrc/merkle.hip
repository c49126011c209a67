// Sparse Merkle multi-update: root of a height-h (h <= 64) Pedersen tree that holds the given
// leaves at the given keys and `empty_leaf` everywhere else.
//
// Reference shape: cairo-lang's merkle_multi_update as called from
// services/perpetual/cairo/state/state.cairo:155-173 (positions tree, orders tree, height 64);
// the hint-side helper starkware/python/merkle_tree.py:4-26 builds the subtree induced by the
// modified leaves level by level (parents = set(index // 2)).  Same walk here: the host does the
// integer bookkeeping of which two children feed each induced node (the merkle_tree.py part),
// the GPU does every hash: per level one pair of Pedersen launches whose accumulate kernel picks its
// operands through the host's child-index list (gathered mode of csrc/pedersen.hip).
#include <algorithm>
#include <climits>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "context.hpp"

namespace sp {

struct Scratch {
  int32_t *X, *ZZ, *Pre;
  unsigned* flag;
};
int enqueue_pedersen(const uint64_t* x, size_t xs, const uint64_t* y, size_t ys, uint64_t* out,
                     size_t os, uint8_t* status, unsigned* flag, size_t n, hipStream_t st,
                     const Scratch& s, const int2* src);
int get_scratch_public(size_t n, Scratch& s, hipStream_t st);
int enqueue_partial_points(const uint64_t* felts, int count, aff_packed* out, hipStream_t st, bool* usable);
int enqueue_pedersen_sparse(const uint64_t* x, const uint64_t* y, uint64_t* out, unsigned* flag, size_t n,
                            hipStream_t st, const Scratch& s, const int2* src, const aff_packed* cpts);
struct PathLevels {  // pedersen.hip ped_path_kernel
  int first, n_levels;
  int val_base[66];
  unsigned src_off[66];
};
int enqueue_pedersen_path(uint64_t* felts, const uint64_t* emp, unsigned* flag, size_t n, hipStream_t st,
                          const int2* src_all, const PathLevels& pl, const aff_packed* cpts_tree, bool* done);

static DeviceBuffer g_sparse_buf;
// empty-subtree roots are a pure function of the empty leaf: cached on the host per leaf value
static std::map<std::vector<uint64_t>, std::vector<uint64_t>> g_empty_cache;  // leaf -> 65 felts
void release_merkle_state() {
  g_sparse_buf.release();
  g_empty_cache.clear();
}

}  // namespace sp

using namespace sp;

// host-side range check of a caller's felt (signature.py:307: hash inputs are in [0, p))
static bool felt_below_p(const uint64_t* v) {
  static const uint64_t P[4] = {1ull, 0ull, 0ull, 0x0800000000000011ull};
  for (int i = 3; i >= 0; --i) {
    if (v[i] != P[i]) return v[i] < P[i];
  }
  return false;
}

// empty-subtree roots: empties[k+1] = H(empties[k], empties[k]); 64 sequential hashes the first
// time a given empty leaf is seen, then served from the host-side cache (65 felts per leaf value)
static int empty_roots(const uint64_t* empty_leaf, const Scratch& s, const std::vector<uint64_t>** out) {
  std::vector<uint64_t> key(empty_leaf, empty_leaf + 4);
  auto it = g_empty_cache.find(key);
  if (it == g_empty_cache.end()) {
    uint64_t* d_full = nullptr;
    SP_HIP(hipMalloc(&d_full, 65 * 32));
    SP_HIP(hipMemcpy(d_full, empty_leaf, 32, hipMemcpyHostToDevice));
    for (unsigned k2 = 0; k2 < 64; ++k2) {
      const int rc = enqueue_pedersen(d_full + 4 * k2, 1, d_full + 4 * k2, 1, d_full + 4 * (k2 + 1), 1, nullptr,
                                      s.flag, 1, 0, s, nullptr);
      if (rc != SP_OK) { (void)hipFree(d_full); return rc; }
    }
    std::vector<uint64_t> all(65 * 4);
    SP_HIP(hipDeviceSynchronize());
    SP_HIP(hipMemcpy(all.data(), d_full, 65 * 32, hipMemcpyDeviceToHost));
    (void)hipFree(d_full);
    it = g_empty_cache.emplace(key, all).first;
  }
  *out = &it->second;
  return SP_OK;
}

extern "C" int sp_merkle_sparse_root(const uint64_t* keys, const uint64_t* leaves, size_t n,
                                     unsigned height, const uint64_t* empty_leaf, uint64_t* root,
                                     uint8_t* status) {
  SP_REQUIRE_READY();
  if (height > 64) { set_error("height must be <= 64"); return SP_ERR_BAD_ARGUMENT; }
  if (!felt_below_p(empty_leaf)) { set_error("empty leaf must be a field element (< p)"); return SP_ERR_BAD_ARGUMENT; }
  for (size_t i = 0; i < n; ++i) {
    if (i > 0 && keys[i] <= keys[i - 1]) { set_error("keys must be strictly increasing"); return SP_ERR_BAD_ARGUMENT; }
    if (height < 64 && (keys[i] >> height) != 0) { set_error("key out of range for height"); return SP_ERR_BAD_ARGUMENT; }
  }
  Context& c = ctx();
  ctx_lock lk(c.mu);
  // ---- host bookkeeping: induced subtree (merkle_tree.py:18-26) ----
  std::vector<uint64_t> idx(keys, keys + n);
  std::vector<int2> src;                 // all levels, concatenated
  std::vector<size_t> level_off, level_cnt;
  for (unsigned l = 0; l < height && !idx.empty(); ++l) {
    std::vector<uint64_t> nxt;
    nxt.reserve(idx.size());
    level_off.push_back(src.size());
    const size_t m = idx.size();
    for (size_t j = 0; j < m;) {
      int2 s;
      if ((idx[j] & 1) == 0) {
        s.x = (int)j;
        if (j + 1 < m && idx[j + 1] == idx[j] + 1) { s.y = (int)(j + 1); nxt.push_back(idx[j] >> 1); j += 2; }
        else { s.y = -1; nxt.push_back(idx[j] >> 1); j += 1; }
      } else {
        s.x = -1; s.y = (int)j; nxt.push_back(idx[j] >> 1); j += 1;
      }
      src.push_back(s);
    }
    level_cnt.push_back(nxt.size());
    idx.swap(nxt);
  }
  // ---- device buffers: empties[height+1], vals ping/pong [n], src ----
  const size_t nn = n ? n : 1;
  const size_t fb = nn * 32;
  const size_t emp_bytes = ((size_t)height + 1) * 32;
  const size_t src_bytes = ((src.size() + 1) * sizeof(int2) + 63) & ~(size_t)63;
  const size_t cpt_bytes = 2 * ((size_t)height + 1) * sizeof(aff_packed);
  SP_HIP(g_sparse_buf.reserve(emp_bytes + 2 * fb + src_bytes + cpt_bytes + 1024));
  char* b = (char*)g_sparse_buf.ptr;
  uint64_t* d_emp = (uint64_t*)b;
  uint64_t* d_a = (uint64_t*)(b + emp_bytes);
  uint64_t* d_b = (uint64_t*)(b + emp_bytes + fb);
  int2* d_src = (int2*)(b + emp_bytes + 2 * fb);
  // the constant points of the levels (SPARSE quad kernel, pedersen.hip), 64-byte aligned behind the child lists
  aff_packed* d_cpts = (aff_packed*)(((uintptr_t)(b + emp_bytes + 2 * fb + src_bytes) + 63) & ~(uintptr_t)63);
  Scratch s;
  int rc = get_scratch_public(nn, s, 0);
  if (rc != SP_OK) return rc;
  SP_HIP(hipMemsetAsync(s.flag, 0, sizeof(unsigned), 0));
  {
    const std::vector<uint64_t>* all = nullptr;
    rc = empty_roots(empty_leaf, s, &all);
    if (rc != SP_OK) return rc;
    SP_HIP(hipMemcpy(d_emp, all->data(), emp_bytes, hipMemcpyHostToDevice));
  }
  if (n == 0) {
    SP_HIP(hipDeviceSynchronize());
    SP_HIP(hipMemcpy(root, d_emp + 4 * height, 32, hipMemcpyDeviceToHost));
  } else {
    SP_HIP(hipMemcpy(d_a, leaves, n * 32, hipMemcpyHostToDevice));
    if (!src.empty()) SP_HIP(hipMemcpy(d_src, src.data(), src.size() * sizeof(int2), hipMemcpyHostToDevice));
    uint64_t *cur = d_a, *nxt = d_b;
    bool have_cpts = false;
    if (height > 0) {
      rc = enqueue_partial_points(d_emp, (int)height, d_cpts, 0, &have_cpts);
      if (rc != SP_OK) return rc;
    }
    for (size_t l = 0; l < level_cnt.size(); ++l) {
      const size_t m = level_cnt[l];
      // gathered mode: operand pointers come from src (children in `cur`, or this level's
      // empty-subtree root) inside the accumulate kernel - no separate gather pass
      rc = enqueue_pedersen_sparse(cur, d_emp + 4 * l, nxt, s.flag, m, 0, s, d_src + level_off[l],
                                   have_cpts ? d_cpts + 2 * l : nullptr);
      if (rc != SP_OK) return rc;
      std::swap(cur, nxt);
    }
    SP_HIP(hipDeviceSynchronize());
    SP_HIP(hipMemcpy(root, cur, 32, hipMemcpyDeviceToHost));
  }
  if (status) {
    unsigned f = 0;
    SP_HIP(hipMemcpy(&f, s.flag, sizeof(unsigned), hipMemcpyDeviceToHost));
    *status = (uint8_t)f;
  }
  return SP_OK;
}

// =================================================================================================
// Persistent sparse trees: merkle_multi_update on a tree that already holds state.
//
// services/perpetual/cairo/state/state.cairo:155-173 updates the positions tree and the orders
// tree (height 64) once per batch: old root and new root along the union of the touched paths, the
// untouched siblings coming from the previous state (the `merkle_facts` store of main.cairo:61-64).
// sp_merkle_sparse_root only covers the first batch (everything else empty).  A tree handle keeps
// the state IN HBM: an open-addressing hash table (level, index) -> felt of every node that differs
// from the empty-subtree root of its level.  An update ships the sorted keys and the new leaves
// (40 B per leaf) and runs entirely on the device:
//   per level   tree_level_kernel: from the level's sorted node indices builds the parents' indices
//               (merkle_tree.py:18-26: parents = set(index // 2)), the child-index list of every parent
//               and looks the untouched siblings up in the table;
//               the gathered Pedersen launch of csrc/pedersen.hip hashes the level;
//   at the end  one insert kernel writes the new leaves and every new node into the table - only if no
//               input was out of range / unhashable, so a failed update leaves the tree as it was.
// The host only counts the nodes per level (a function of the keys alone) to size the launches, and
// keeps a copy of the current root.  Round 1 kept the node store in host maps: 12 ms per 4096-leaf update
// of a height-64 tree, most of it map lookups / insertions and the 8 MB of values crossing PCIe.
struct TreeSlot {       // 48 bytes
  uint64_t index;
  uint32_t level1;      // level + 1; 0 = never used
  uint32_t state;       // 0 empty, 1 being written, 2 ready
  uint64_t value[4];
};
__device__ __forceinline__ uint64_t tree_mix(uint64_t index, uint32_t level) {
  uint64_t z = index + 0x9E3779B97F4A7C15ull * (uint64_t)(level + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// value of node (level, index) or nullptr when the node is still the empty-subtree root of its level
__device__ __forceinline__ const uint64_t* tree_find(const TreeSlot* __restrict__ tab, uint64_t mask, uint32_t level,
                                                     uint64_t index) {
  for (uint64_t i = tree_mix(index, level) & mask;; i = (i + 1) & mask) {
    const TreeSlot& sl = tab[i];
    // lookups never run beside insertions (different launches of one stream): plain loads
    const uint32_t st = sl.state;
    if (st == 0) return nullptr;
    if (st == 2 && sl.level1 == level + 1 && sl.index == index) return sl.value;
  }
}
// Inserts or overwrites; the keys of one launch are distinct, so two lanes never claim the same key.
__device__ __forceinline__ bool tree_put(TreeSlot* __restrict__ tab, uint64_t mask, uint32_t level, uint64_t index,
                                         const uint64_t* value) {
  for (uint64_t i = tree_mix(index, level) & mask;; i = (i + 1) & mask) {
    TreeSlot& sl = tab[i];
    // relaxed: a slot another lane is filling right now belongs to a different key (the keys of a launch
    // are distinct), so whatever is read from it cannot match
    uint32_t st = __hip_atomic_load(&sl.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st == 0) {
      uint32_t expected = 0;
      if (__hip_atomic_compare_exchange_strong(&sl.state, &expected, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)) {
        sl.index = index;
        sl.level1 = level + 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) sl.value[q] = value[q];
        __hip_atomic_store(&sl.state, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // readers: later launches
        return true;  // a slot that was empty: the table's entry count grows
      }
      st = expected;  // somebody else claimed it: a different key (keys are distinct), keep probing
    }
    if (st == 2 && sl.level1 == level + 1 && sl.index == index) {
#pragma unroll
      for (int q = 0; q < 4; ++q) sl.value[q] = value[q];
      return false;
    }
  }
}

// Layout of one update in the work buffer (host-computed from the level counts, uploaded once).
struct TreeLevels {
  unsigned height;
  unsigned cnt[66];       // nodes per level (level 0 = the new leaves)
  int val_base[66];       // felts: where the values of level l start
  int sib_base[66];       // felts: where the siblings of level l + 1's parents go (one slot per parent)
  unsigned idx_off[66];   // idx: where the node indices of level l start
  unsigned src_off[66];   // src: where the child lists of level l + 1 start
};

// The structure of the whole update is built before any hash runs (it depends on the sorted keys only), level-
// PARALLEL: the nodes of level l are the distinct values of key >> l, in order, so
//   tree_level_nodes_kernel   block l (one workgroup per level) writes the node indices of level l with a
//                             ballot scan over the keys - no level waits for another (round 2 walked the levels
//                             one after the other in ONE workgroup: 370 us of a 3.5 ms update);
//   tree_children_kernel      one thread per node of levels 0 .. height - 1: a node that opens a new parent
//                             (merkle_tree.py:18-26) finds the parent's position by binary search in the next
//                             level's (sorted) node list and writes its child list - absolute positions in
//                             `felts`, TREE_PENDING = the sibling comes from the table (tree_lookup_kernel).
constexpr int TREE_PENDING = INT_MIN;
__global__ void __launch_bounds__(1024)
tree_level_nodes_kernel(TreeLevels lv, const uint64_t* __restrict__ keys, uint64_t* __restrict__ idx_all) {
  __shared__ unsigned wave_tot[16];
  __shared__ unsigned carry;
  const unsigned level = blockIdx.x + 1;  // 1 .. height (level 0 = the keys themselves, already in place)
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned n = lv.cnt[0];
  uint64_t* out = idx_all + lv.idx_off[level];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (unsigned tile = 0; tile < n; tile += 1024) {
    const unsigned i = tile + threadIdx.x;
    const bool live = i < n;
    const uint64_t me = live ? (level < 64 ? keys[i] >> level : 0) : 0;
    const bool head = live && (i == 0 || (level < 64 ? keys[i - 1] >> level : 0) != me);
    const unsigned long long ballot = __ballot(head);
    const unsigned before = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(ballot);
    __syncthreads();
    unsigned off = carry;
    for (unsigned w = 0; w < wave; ++w) off += wave_tot[w];
    if (head) out[off + before] = me;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned tot = carry;
      for (unsigned w = 0; w < 16; ++w) tot += wave_tot[w];
      carry = tot;
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256)
tree_children_kernel(TreeLevels lv, const uint64_t* __restrict__ idx_all, int2* __restrict__ src_all, unsigned n_nodes) {
  const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;  // node number over levels 0 .. height - 1
  if (g >= n_nodes) return;
  unsigned level = 0;
  while (level + 1 < lv.height && g >= lv.idx_off[level + 1]) ++level;  // idx_off is the running node count
  const unsigned j = g - lv.idx_off[level], cnt = lv.cnt[level];
  const uint64_t* idx = idx_all + lv.idx_off[level];
  const uint64_t me = idx[j];
  // a node opens a new parent unless its left neighbour in the array is its left sibling
  if ((me & 1) && j > 0 && idx[j - 1] == me - 1) return;
  const uint64_t* par = idx_all + lv.idx_off[level + 1];
  unsigned lo = 0, hi = lv.cnt[level + 1];
  while (lo < hi) {  // first parent >= me >> 1 (it is there)
    const unsigned mid = (lo + hi) >> 1;
    if (par[mid] < (me >> 1)) lo = mid + 1; else hi = mid;
  }
  const int val_base = lv.val_base[level];
  int2 s2;
  if ((me & 1) == 0) {
    s2.x = val_base + (int)j;
    s2.y = (j + 1 < cnt && idx[j + 1] == me + 1) ? val_base + (int)j + 1 : TREE_PENDING;
  } else {
    s2.y = val_base + (int)j;
    s2.x = TREE_PENDING;
  }
  src_all[lv.src_off[level] + lo] = s2;
}

// Every parent whose child list has a pending side: the untouched sibling from the table (copied to
// felts[sib_base + q]) or -1 = the level's empty-subtree root.  One thread per parent of the update.
__global__ void __launch_bounds__(256)
tree_lookup_kernel(TreeLevels lv, const uint64_t* __restrict__ idx_all, const TreeSlot* __restrict__ tab, uint64_t mask,
                   uint64_t* __restrict__ felts, int2* __restrict__ src_all, unsigned n_parents) {
  const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_parents) return;
  unsigned level = 0;  // children's level: src_off is the running parent count
  while (level + 1 < lv.height && g >= lv.src_off[level + 1]) ++level;
  const unsigned q = g - lv.src_off[level];
  int2 s2 = src_all[g];
  if (s2.x != TREE_PENDING && s2.y != TREE_PENDING) return;
  const uint64_t parent = idx_all[lv.idx_off[level + 1] + q];
  const bool left = s2.x == TREE_PENDING;
  const uint64_t* sib = tree_find(tab, mask, level, 2 * parent + (left ? 0 : 1));
  const int pos = sib ? lv.sib_base[level] + (int)q : -1;
  if (sib) {
#pragma unroll
    for (int w = 0; w < 4; ++w) felts[4 * (size_t)pos + w] = sib[w];
  }
  if (left) s2.x = pos; else s2.y = pos;
  src_all[g] = s2;
}

// table[(level, idx)] = value for every node of the update, all levels in one launch
__global__ void __launch_bounds__(256)
tree_insert_kernel(TreeSlot* __restrict__ tab, uint64_t mask, TreeLevels lv, const uint64_t* __restrict__ idx_all,
                   const uint64_t* __restrict__ felts, unsigned total, unsigned long long* __restrict__ fresh) {
  const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
  bool claimed = false;
  if (g < total) {
    unsigned level = 0;
    while (level < lv.height && g >= lv.idx_off[level + 1]) ++level;  // idx_off is the running node count
    const unsigned j = g - lv.idx_off[level];
    claimed = tree_put(tab, mask, level, idx_all[g], felts + 4 * (size_t)(lv.val_base[level] + (int)j));
  }
  const unsigned long long b = __ballot(claimed);  // one atomic per wave for the entry counter
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(fresh, (unsigned long long)__popcll(b));
}
// out[j] = leaf (level 0) keys[j] or the empty leaf
__global__ void __launch_bounds__(256)
tree_get_kernel(const TreeSlot* __restrict__ tab, uint64_t mask, const uint64_t* __restrict__ keys, unsigned cnt,
                const uint64_t* __restrict__ empty_leaf, uint64_t* __restrict__ out) {
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cnt) return;
  const uint64_t* v = tree_find(tab, mask, 0, keys[j]);
  if (!v) v = empty_leaf;
#pragma unroll
  for (int w = 0; w < 4; ++w) out[4 * (size_t)j + w] = v[w];
}
// rehash every ready slot of `old` into `tab`
__global__ void __launch_bounds__(256)
tree_rehash_kernel(const TreeSlot* __restrict__ old, uint64_t old_slots, TreeSlot* __restrict__ tab, uint64_t mask) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= old_slots) return;
  const TreeSlot& sl = old[i];
  if (sl.state == 2) (void)tree_put(tab, mask, sl.level1 - 1, sl.index, sl.value);
}

struct SparseTree {
  unsigned height = 0;
  uint64_t empty_leaf[4] = {0, 0, 0, 0};
  int ctx_index = 0;           // the context (device) the tree lives on
  TreeSlot* table = nullptr;   // HBM
  TreeSlot* retired = nullptr; // the table before the last growth (tree_reserve), freed at the next growth / with the tree
  uint64_t slots = 0;          // power of two
  uint64_t entries = 0;        // used slots as of the last read of d_entries
  unsigned long long* d_entries = nullptr;  // device counter of claimed slots (insert kernels add to it)
  bool has_root = false;
  uint64_t root[4] = {0, 0, 0, 0};
  // Every tree has its OWN stream and work buffer and its own mutex: an operation holds the tree's mutex from
  // start to end (updates of one tree are ordered anyway) and takes the library lock only to enqueue - the
  // copies and the ~3 ms of level launches of a height-64 update run without it, so hash / verify batches and
  // updates of OTHER trees proceed meanwhile (round 2: null stream, library lock across everything).
  hipStream_t stream = nullptr;
  DeviceBuffer buf;
  PinnedBuffer hbuf;  // host-side staging of an update's inputs (empty roots, leaves, keys): see PinnedBuffer
  // The constant points of the levels (pedersen.hip SPARSE quad kernel): for level l the sum of the table entries
  // that the level's empty-subtree root selects as a left / right operand, [2 l] and [2 l + 1]; computed on the
  // tree's stream by its first update, kept for its lifetime (the window plan cannot change under a live tree).
  DeviceBuffer cpts;
  int cpts_state = 0;  // 0: not computed, 1: ready, -1: the plan has no constant window on one side
  std::mutex mu;
  bool destroyed = false;  // set by sp_tree_destroy under `mu`: a thread that was waiting for the mutex with the
                           // old handle must not revive the freed tree
};

static std::map<int, std::shared_ptr<SparseTree>> g_trees;
static int g_next_tree = 1;

static void tree_free(SparseTree& t) {
  if (t.stream) {
    (void)hipStreamSynchronize(t.stream);
    (void)hipStreamDestroy(t.stream);
  }
  if (t.table) (void)hipFree(t.table);
  if (t.retired) (void)hipFree(t.retired);
  if (t.d_entries) (void)hipFree(t.d_entries);
  t.buf.release();
  t.hbuf.release();
  t.cpts.release();
  t.cpts_state = 0;
  t.stream = nullptr;
  t.table = t.retired = nullptr;
  t.d_entries = nullptr;
  t.slots = 0;
}

// Looks the handle up, selects the tree's context and device for this host thread and locks the tree.
struct TreeScope {
  std::shared_ptr<SparseTree> t;
  int previous_ctx = -1;
  int previous_dev = -1;
  bool switched_dev = false;
  std::unique_lock<std::mutex> held;
  int open(int handle) {
    {
      ctx_lock lk(global_mu());
      auto it = g_trees.find(handle);
      if (it == g_trees.end()) { set_error("unknown tree handle"); return SP_ERR_BAD_ARGUMENT; }
      t = it->second;
    }
    held = std::unique_lock<std::mutex>(t->mu);
    if (t->destroyed) { set_error("unknown tree handle"); return SP_ERR_BAD_ARGUMENT; }
    previous_ctx = ctx_current();
    ctx_select(t->ctx_index);
    const int dev = ctx().device;
    if (hipGetDevice(&previous_dev) == hipSuccess && previous_dev != dev) switched_dev = hipSetDevice(dev) == hipSuccess;
    if (!t->stream) SP_HIP(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    return SP_OK;
  }
  ~TreeScope() {
    if (previous_ctx >= 0) ctx_select(previous_ctx);
    if (switched_dev) (void)hipSetDevice(previous_dev);
  }
};

// Room for `extra` more entries at a load factor of at most 1/2 (tree mutex held).
// Growth policy (round 6): a table that has to grow is sized for FOUR times what is asked of it
// (load factor 1/4 straight after a growth), so that the next updates of the same size fit without another one, and
// the growth itself costs the caller no wait: allocation, clear and rehash are enqueued on the tree's stream and the
// old table is retired, not freed - hipFree synchronises the whole device, which in sp_order_batch stalled the
// verification running beside the update (calls 1 - 2 of a fresh tree cost 2.9 - 3.1 ms against 1.97 ms in round
// 5).  The retired table goes back at the next growth (its rehash has long finished: the intervening update
// synchronised the stream) or with the tree.  First allocation: 2^16 slots, or STARKPERP_TREE_INITIAL_SLOTS_LOG2.
static uint64_t tree_initial_slots() {
  static const uint64_t v = [] {
    const char* e = getenv("STARKPERP_TREE_INITIAL_SLOTS_LOG2");
    const int lg = e ? atoi(e) : 16;
    return (uint64_t)1 << (lg < 10 ? 10 : (lg > 30 ? 30 : lg));
  }();
  return v;
}
static int tree_reserve(SparseTree& t, uint64_t extra) {
  if (!t.d_entries) {
    SP_HIP(hipMalloc(&t.d_entries, sizeof(unsigned long long)));
    SP_HIP(hipMemsetAsync(t.d_entries, 0, sizeof(unsigned long long), t.stream));
  }
  if (t.slots && 2 * (t.entries + extra) > t.slots) {  // would not fit by the last known count: refresh it
    unsigned long long used = 0;
    SP_HIP(hipMemcpyAsync(&used, t.d_entries, sizeof(used), hipMemcpyDeviceToHost, t.stream));
    SP_HIP(hipStreamSynchronize(t.stream));
    t.entries = used;
  }
  if (t.slots && 2 * (t.entries + extra) <= t.slots) return SP_OK;
  uint64_t want = t.slots ? t.slots : tree_initial_slots();
  while (4 * (t.entries + extra) > want) want <<= 1;
  if (want == t.slots) return SP_OK;
  if (t.retired) {  // the table before the last one: nothing on the stream reads it any more
    SP_HIP(hipStreamSynchronize(t.stream));
    (void)hipFree(t.retired);
    t.retired = nullptr;
  }
  TreeSlot* fresh = nullptr;
  SP_HIP(hipMalloc(&fresh, want * sizeof(TreeSlot)));
  tl_mark("tree: table allocated");
  SP_HIP(hipMemsetAsync(fresh, 0, want * sizeof(TreeSlot), t.stream));
  if (t.table) {
    hipLaunchKernelGGL(tree_rehash_kernel, dim3((unsigned)((t.slots + 255) / 256)), dim3(256), 0, t.stream, t.table,
                       t.slots, fresh, want - 1);
    SP_HIP(hipGetLastError());
    t.retired = t.table;  // read by the rehash just enqueued; freed later, without a device-wide wait now
  }
  t.table = fresh;
  t.slots = want;
  return SP_OK;
}

static int tree_root_locked(SparseTree& t, uint64_t* root) {
  if (t.has_root) {
    std::memcpy(root, t.root, 32);
    return SP_OK;
  }
  ctx_lock lk(global_mu());  // the cache of empty-subtree roots and the scratch map are shared
  Scratch s;
  int rc = get_scratch_public(1, s, 0);
  if (rc != SP_OK) return rc;
  const std::vector<uint64_t>* emp = nullptr;
  rc = empty_roots(t.empty_leaf, s, &emp);
  if (rc != SP_OK) return rc;
  std::memcpy(root, emp->data() + 4 * t.height, 32);
  return SP_OK;
}

// The update itself (tree mutex held, context selected).  `may_commit`, when given, is asked once every new
// node has been hashed and before anything is written to the table: false leaves the tree as it was
// (*status = SP_TREE_NOT_COMMITTED).
static int tree_update_locked(SparseTree& t, const uint64_t* keys, const uint64_t* leaves, size_t n, uint64_t* old_root,
                              uint64_t* new_root, uint8_t* status, const std::function<bool()>* may_commit,
                              const std::function<void()>* enqueued = nullptr) {
  const unsigned height = t.height;
  for (size_t i = 0; i < n; ++i) {
    if (i > 0 && keys[i] <= keys[i - 1]) { set_error("keys must be strictly increasing"); return SP_ERR_BAD_ARGUMENT; }
    if (height < 64 && (keys[i] >> height) != 0) { set_error("key out of range for height"); return SP_ERR_BAD_ARGUMENT; }
  }
  int rc = tree_root_locked(t, old_root);
  if (rc != SP_OK) return rc;
  if (status) *status = 0;
  if (n == 0) {
    std::memcpy(new_root, old_root, 32);
    if (may_commit && !(*may_commit)() && status) *status = SP_TREE_NOT_COMMITTED;
    return SP_OK;
  }
  // ---- host: only the node COUNT of every level (merkle_tree.py:18-26 on the keys alone): level l has one node
  // per distinct key >> l, i.e. 1 + the adjacent key pairs whose highest differing bit is >= l ----
  std::vector<size_t> cnt(height + 1);
  {
    size_t hist[65] = {0};
    for (size_t i = 1; i < n; ++i) ++hist[63 - __builtin_clzll(keys[i] ^ keys[i - 1])];
    size_t above = 0;
    for (int l = 64; l >= 0; --l) {
      if (l < 64) above += hist[l];
      if ((unsigned)l <= height) cnt[l] = 1 + above;
    }
  }
  size_t total = 0;
  for (unsigned l = 0; l <= height; ++l) total += cnt[l];
  // felts: [level 0 values][siblings for level 1's parents][level 1 values][siblings ...] ...
  TreeLevels lv;
  std::memset(&lv, 0, sizeof(lv));
  lv.height = height;
  size_t felts = 0, idxs = 0, srcs = 0;
  for (unsigned l = 0; l <= height; ++l) {
    lv.cnt[l] = (unsigned)cnt[l];
    lv.val_base[l] = (int)felts;
    felts += cnt[l];
    lv.idx_off[l] = (unsigned)idxs;
    idxs += cnt[l];
    if (l < height) {
      lv.sib_base[l] = (int)felts;
      felts += cnt[l + 1];
      lv.src_off[l] = (unsigned)srcs;
      srcs += cnt[l + 1];
    }
  }
  if (felts >= (size_t)INT_MAX || total > 0x7fffffffull) { set_error("update too large"); return SP_ERR_BAD_ARGUMENT; }
  tl_mark("tree: level counts");
  rc = tree_reserve(t, total);
  if (rc != SP_OK) return rc;
  tl_mark("tree: table reserved");
  const size_t emp_bytes = ((size_t)height + 1) * 32;
  const size_t felt_bytes = felts * 32, idx_bytes = idxs * 8, src_bytes = srcs * sizeof(int2);
  SP_HIP(t.buf.reserve(emp_bytes + felt_bytes + idx_bytes + src_bytes + 1024));
  char* b = (char*)t.buf.ptr;
  uint64_t* d_emp = (uint64_t*)b;
  uint64_t* d_felts = (uint64_t*)(b + emp_bytes);
  uint64_t* d_idx = (uint64_t*)(b + emp_bytes + felt_bytes);
  int2* d_src = (int2*)(b + emp_bytes + felt_bytes + idx_bytes);
  hipStream_t st = t.stream;
  Scratch s;
  {
    // ---- enqueue under the library lock: scratch map, empty-root cache, launch bookkeeping ----
    tl_mark("tree: work buffer ready");
    ctx_lock lk(global_mu());
    tl_mark("tree: library lock taken");
    rc = get_scratch_public(n, s, st);
    if (rc != SP_OK) return rc;
    tl_mark("tree: scratch");
    SP_HIP(hipMemsetAsync(s.flag, 0, sizeof(unsigned), st));
    tl_mark("tree: flag cleared");
    const std::vector<uint64_t>* emp = nullptr;
    rc = empty_roots(t.empty_leaf, s, &emp);
    if (rc != SP_OK) return rc;
    tl_mark("tree: empty roots");
    // empty roots + leaves are adjacent in the work buffer (d_emp, then the level-0 values at the start of d_felts):
    // one copy; the keys a second one.  Small updates go through the tree's page-locked buffer (PinnedBuffer).
    const size_t in_bytes = emp_bytes + n * 32 + n * 8;
    char* stage = nullptr;
    if (in_bytes <= PINNED_STAGE_MAX && t.hbuf.reserve(in_bytes) == hipSuccess) stage = (char*)t.hbuf.ptr;
    else (void)hipGetLastError();
    if (stage) {
      // the previous update's copies out of this buffer are long finished: every update ends with a wait on `st`
      std::memcpy(stage, emp->data(), emp_bytes);
      std::memcpy(stage + emp_bytes, leaves, n * 32);
      std::memcpy(stage + emp_bytes + n * 32, keys, n * 8);
      SP_HIP(hipMemcpyAsync(d_emp, stage, emp_bytes + n * 32, hipMemcpyHostToDevice, st));
      SP_HIP(hipMemcpyAsync(d_idx, stage + emp_bytes + n * 32, n * 8, hipMemcpyHostToDevice, st));
    } else {
      SP_HIP(hipMemcpyAsync(d_emp, emp->data(), emp_bytes, hipMemcpyHostToDevice, st));
      SP_HIP(hipMemcpyAsync(d_felts, leaves, n * 32, hipMemcpyHostToDevice, st));
      SP_HIP(hipMemcpyAsync(d_idx, keys, n * 8, hipMemcpyHostToDevice, st));
    }
    tl_mark("tree: inputs copied");
    // the structure of every level and the sibling lookups, then the hashes level by level
    hipLaunchKernelGGL(tree_level_nodes_kernel, dim3(height), dim3(1024), 0, st, lv, d_idx, d_idx);
    hipLaunchKernelGGL(tree_children_kernel, dim3((unsigned)((total - cnt[height] + 255) / 256)), dim3(256), 0, st, lv,
                       d_idx, d_src, (unsigned)(total - cnt[height]));
    hipLaunchKernelGGL(tree_lookup_kernel, dim3((unsigned)((srcs + 255) / 256)), dim3(256), 0, st, lv, d_idx, t.table,
                       t.slots - 1, d_felts, d_src, (unsigned)srcs);
    SP_HIP(hipGetLastError());
    tl_mark("tree: structure kernels enqueued");
    if (t.cpts_state == 0) {  // once per tree: the constant points of its 64 levels (d_emp is on the stream already)
      SP_HIP(t.cpts.reserve(2 * (size_t)height * sizeof(aff_packed)));
      bool usable = false;
      rc = enqueue_partial_points(d_emp, (int)height, (aff_packed*)t.cpts.ptr, st, &usable);
      if (rc != SP_OK) return rc;
      t.cpts_state = usable ? 1 : -1;
    }
    const aff_packed* cpts = t.cpts_state == 1 ? (const aff_packed*)t.cpts.ptr : nullptr;
    for (unsigned l = 0; l < height;) {
      // a run of levels whose paths do not merge (as many parents as children: every node has ONE touched child)
      // goes out as one launch where the size class allows it (ped_path_kernel)
      unsigned l2 = l;
      while (l2 < height && cnt[l2 + 1] == cnt[l]) ++l2;
      if (l2 - l >= 2) {
        PathLevels pl;
        std::memset(&pl, 0, sizeof(pl));
        pl.first = (int)l;
        pl.n_levels = (int)(l2 - l);
        for (unsigned j = 0; j <= height; ++j) { pl.val_base[j] = lv.val_base[j]; pl.src_off[j] = lv.src_off[j]; }
        bool done = false;
        rc = enqueue_pedersen_path(d_felts, d_emp, s.flag, cnt[l], st, d_src, pl, cpts, &done);
        if (rc != SP_OK) return rc;
        if (done) { l = l2; continue; }
      }
      rc = enqueue_pedersen_sparse(d_felts, d_emp + 4 * l, d_felts + 4 * (size_t)lv.val_base[l + 1], s.flag, cnt[l + 1],
                                   st, s, d_src + lv.src_off[l], cpts ? cpts + 2 * l : nullptr);
      if (rc != SP_OK) return rc;
      ++l;
    }
  }
  tl_mark("tree: levels enqueued");
  if (enqueued) (*enqueued)();  // the library lock is free again: sp_order_batch lets its verifier through
  // ---- the device runs; nobody waits on the library lock for it ----
  // The status flag and the candidate root come back together: ONE wait per update.
  unsigned f = 0;
  uint64_t candidate[4];
  SP_HIP(hipMemcpyAsync(&f, s.flag, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  SP_HIP(hipMemcpyAsync(candidate, d_felts + 4 * (size_t)lv.val_base[height], 32, hipMemcpyDeviceToHost, st));
  SP_HIP(hipStreamSynchronize(st));
  tl_mark("tree: levels hashed");
  if (status) *status = (uint8_t)f;
  if (f != 0) {  // an input out of range or an unhashable pair: nothing was written, the tree is as it was
    std::memcpy(new_root, old_root, 32);
    return SP_OK;
  }
  if (may_commit && !(*may_commit)()) {
    if (status) *status = SP_TREE_NOT_COMMITTED;
    std::memcpy(new_root, old_root, 32);
    return SP_OK;
  }
  // ---- commit: every new node into the table, one launch - and nobody waits for it.  The tree's next
  // operation is ordered behind it on the tree's stream (lookups, the next update's copies into the work
  // buffer, the rehash of a growing table); a reallocation of the work buffer or sp_tree_destroy synchronise
  // (hipFree / tree_free).  ~0.09 ms of 217 k hash-table insertions leave the caller's latency.
  hipLaunchKernelGGL(tree_insert_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, t.table, t.slots - 1, lv,
                     d_idx, d_felts, (unsigned)total, t.d_entries);
  SP_HIP(hipGetLastError());
  std::memcpy(t.root, candidate, 32);
  t.has_root = true;
  t.entries += total;  // upper bound until tree_reserve reads the device counter again
  std::memcpy(new_root, t.root, 32);
  return SP_OK;
}

namespace sp {
void release_tree_state() {
  for (auto& kv : g_trees) tree_free(*kv.second);
  g_trees.clear();
}
}  // namespace sp

extern "C" {

int sp_tree_create_on(int context, unsigned height, const uint64_t* empty_leaf, int* tree) {
  SP_REQUIRE_READY();
  if (height < 1 || height > 64) { set_error("height must be in 1..64"); return SP_ERR_BAD_ARGUMENT; }
  if (context < 0 || context >= ctx_count() || !ctx_at(context).ready) { set_error("no such context"); return SP_ERR_BAD_ARGUMENT; }
  // an out-of-range empty leaf would poison the cached empty-subtree roots (their chain flags
  // SP_HASH_OUT_OF_RANGE once, later calls hit the cache and never see the flag again)
  if (!felt_below_p(empty_leaf)) { set_error("empty leaf must be a field element (< p)"); return SP_ERR_BAD_ARGUMENT; }
  ctx_lock lk(global_mu());
  auto t = std::make_shared<SparseTree>();
  t->height = height;
  t->ctx_index = context;
  std::memcpy(t->empty_leaf, empty_leaf, 32);
  const int id = g_next_tree++;
  g_trees.emplace(id, t);
  *tree = id;
  return SP_OK;
}

int sp_tree_create(unsigned height, const uint64_t* empty_leaf, int* tree) {
  return sp_tree_create_on(0, height, empty_leaf, tree);
}

int sp_tree_destroy(int tree) {
  SP_REQUIRE_READY();
  std::shared_ptr<SparseTree> keep;
  {
    TreeScope ts;
    int rc = ts.open(tree);
    if (rc != SP_OK) return rc;
    keep = ts.t;
    {
      ctx_lock lk(global_mu());
      g_trees.erase(tree);
    }
    tree_free(*ts.t);  // waits for the tree's stream
    ts.t->destroyed = true;
  }
  return SP_OK;
}

int sp_tree_root(int tree, uint64_t* root) {
  SP_REQUIRE_READY();
  TreeScope ts;
  int rc = ts.open(tree);
  if (rc != SP_OK) return rc;
  return tree_root_locked(*ts.t, root);
}

int sp_tree_get(int tree, const uint64_t* keys, size_t n, uint64_t* leaves) {
  SP_REQUIRE_READY();
  TreeScope ts;
  int rc = ts.open(tree);
  if (rc != SP_OK) return rc;
  SparseTree& t = *ts.t;
  for (size_t i = 0; i < n; ++i) {
    if (t.height < 64 && (keys[i] >> t.height) != 0) { set_error("key out of range for height"); return SP_ERR_BAD_ARGUMENT; }
  }
  if (n == 0) return SP_OK;
  if (!t.table) {
    for (size_t i = 0; i < n; ++i) std::memcpy(leaves + 4 * i, t.empty_leaf, 32);
    return SP_OK;
  }
  if (n > 0xffffffffull) { set_error("too many keys"); return SP_ERR_BAD_ARGUMENT; }
  SP_HIP(t.buf.reserve(n * 40 + 64));
  uint64_t* d_keys = (uint64_t*)t.buf.ptr;
  uint64_t* d_emp = d_keys + n;
  uint64_t* d_out = d_emp + 4;
  SP_HIP(hipMemcpyAsync(d_keys, keys, n * 8, hipMemcpyHostToDevice, t.stream));
  SP_HIP(hipMemcpyAsync(d_emp, t.empty_leaf, 32, hipMemcpyHostToDevice, t.stream));
  hipLaunchKernelGGL(tree_get_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, t.stream, t.table, t.slots - 1,
                     d_keys, (unsigned)n, d_emp, d_out);
  SP_HIP(hipGetLastError());
  SP_HIP(hipMemcpyAsync(leaves, d_out, n * 32, hipMemcpyDeviceToHost, t.stream));
  SP_HIP(hipStreamSynchronize(t.stream));
  return SP_OK;
}

int sp_tree_update(int tree, const uint64_t* keys, const uint64_t* leaves, size_t n, uint64_t* old_root,
                   uint64_t* new_root, uint8_t* status) {
  SP_REQUIRE_READY();
  TreeScope ts;
  int rc = ts.open(tree);
  if (rc != SP_OK) return rc;
  return tree_update_locked(*ts.t, keys, leaves, n, old_root, new_root, status, nullptr);
}

// BASELINE.json configs[2] in ONE call: message-hash chains -> keyed signature verification -> order ids ->
// orders-tree update, without returning to the caller between the stages (services/perpetual/cairo/order:
// limit_order.cairo:24-52 hashes the order, order.cairo:23-31 takes the order id from the top 64 bits of the
// message hash, :122-124 writes the fulfilled amount at that id, verify_ecdsa_signature checks the signature).
//   words      depth x n felts, word-major (word k of order i at words + 4 (k n + i)): the chain
//              h = H(...H(H(w0, w1), w2)..., w_{depth-1}) is the message hash z_i (written to z_out)
//   r, s, qx, qy   the signatures and public keys (qy == NULL: x-only keys), verified against z_i through the
//              key tables; verdicts[i] as sp_ecdsa_verify_batch (z_i >= 2^251: SP_VERIFY_ASSERT_MSG, nothing committed)
//   leaves     the new orders-tree leaf of order i; its key is bits [id_shift, id_shift + 64) of z_i
// The verification runs on a host lane WHILE the tree update's levels are hashed on the tree's stream; the new
// nodes are committed only if every signature verified (the batch is all-or-nothing in the Cairo program),
// otherwise *tree_status = SP_TREE_NOT_COMMITTED and the tree is unchanged.  Two orders with the same id are an
// error (the caller squashes them first, state/state.cairo:67-96).
int sp_order_batch(const uint64_t* words, size_t depth, size_t n, const uint64_t* r, const uint64_t* s,
                   const uint64_t* qx, const uint64_t* qy, int tree, const uint64_t* leaves, unsigned id_shift,
                   uint64_t* z_out, uint8_t* verdicts, uint64_t* old_root, uint64_t* new_root, uint8_t* tree_status) {
  SP_REQUIRE_READY();
  if (depth < 1 || id_shift > 192) { set_error("sp_order_batch: bad chain depth or id shift"); return SP_ERR_BAD_ARGUMENT; }
  TimelineScope timeline("sp_order_batch");
  uint8_t chain_status = 0;
  int rc = sp_pedersen_chains(words, n, depth, z_out, &chain_status);
  if (rc != SP_OK) return rc;
  tl_mark("chains done");
  if (chain_status != 0) {  // a word out of range / an unhashable pair: nothing else runs
    if (tree_status) *tree_status = chain_status;
    for (size_t i = 0; i < n; ++i) verdicts[i] = 0;
    rc = sp_tree_root(tree, old_root);  // its own code and error text (a bad handle is not a HIP failure and v.v.)
    if (rc != SP_OK) return rc;
    std::memcpy(new_root, old_root, 32);
    return SP_OK;
  }
  // A message hash of 2^251 or more is not a signed message (constants.cairo:57 SIGNED_MESSAGE_BOUND,
  // order.cairo:22; signature.py:227 asserts the same bound): its order id would not fit the 64-bit field, so
  // the batch cannot be committed - the verdicts say which orders (SP_VERIFY_ASSERT_MSG) and the tree stays.
  bool unsigned_message = false;
  for (size_t i = 0; i < n; ++i) unsigned_message |= (z_out[4 * i + 3] >> 59) != 0;
  if (unsigned_message) {
    rc = sp_ecdsa_verify_batch_keyed(z_out, r, s, qx, qy, verdicts, n);
    if (rc != SP_OK) return rc;
    if (tree_status) *tree_status = SP_TREE_NOT_COMMITTED;
    rc = sp_tree_root(tree, old_root);
    if (rc != SP_OK) return rc;
    std::memcpy(new_root, old_root, 32);
    return SP_OK;
  }
  // The verification needs nothing but z: it starts NOW, on a thread of its own, so that its host work (lane, staging
  // copies, key lookup) runs beside the sort below and its launch beside the tree's levels.  (Round 5 spawned it after
  // the sort: 150 us later, and it then queued for the library lock behind the tree's enqueue.)
  // The verifier does its lock-free part at once (lane, staging copies) and then waits at a gate until the tree
  // update has enqueued its levels: the tree is the critical path and both need the library lock.
  struct Gate {
    std::mutex m;
    std::condition_variable cv;
    bool open = false;
    void release() {
      { std::lock_guard<std::mutex> lk(m); open = true; }
      cv.notify_all();
    }
    void wait() {
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { return open; });
    }
  } gate;
  const std::function<void()> open_gate = [&] { gate.release(); };
  int vrc = SP_OK;
  std::thread verifier([&] {
    tl_mark("verifier thread runs");
    vrc = verify_batch_keyed_gated(z_out, r, s, qx, qy, verdicts, n, [&] { gate.wait(); });
    tl_mark("verifier done");
  });
  tl_mark("verifier spawned");
  struct Joiner {  // every return below opens the gate and joins the verifier first: it reads the caller's arrays
    std::thread& t;
    Gate& g;
    ~Joiner() { g.release(); if (t.joinable()) t.join(); }
  } joiner{verifier, gate};
  // order ids, sorted, with the leaves in the same order.  The ids are bits of hash outputs - uniform - so one
  // counting pass over their top bits leaves buckets of about one id each and a tiny std::sort inside every bucket
  // (any input is still sorted correctly: a skewed one only makes a bucket's sort longer).  120 us -> ~25 us for
  // 4096 orders against std::sort of (id, index) pairs.
  std::vector<std::pair<uint64_t, size_t>> ids(n);
  {
    const unsigned w = id_shift >> 6, sh = id_shift & 63;
    unsigned lg = 4;
    while (lg < 16 && ((size_t)1 << lg) < n) ++lg;
    const size_t nb = (size_t)1 << lg;
    const unsigned id_bits = 64;  // (ids of a tree lower than 64 all fall into bucket 0: one std::sort, still correct)
    std::vector<uint32_t> start(nb + 1, 0);
    std::vector<uint64_t> raw(n);
    for (size_t i = 0; i < n; ++i) {
      const uint64_t* z = z_out + 4 * i;
      uint64_t v = z[w] >> sh;
      if (sh && w + 1 < 4) v |= z[w + 1] << (64 - sh);
      raw[i] = v;
      ++start[(v >> (id_bits - lg)) + 1];
    }
    for (size_t b = 0; b < nb; ++b) start[b + 1] += start[b];
    std::vector<uint32_t> fill(start.begin(), start.end() - 1);
    for (size_t i = 0; i < n; ++i) ids[fill[raw[i] >> (id_bits - lg)]++] = {raw[i], i};
    for (size_t b = 0; b < nb; ++b)
      if (start[b + 1] - start[b] > 1) std::sort(ids.begin() + start[b], ids.begin() + start[b + 1]);
  }
  std::vector<uint64_t> keys(n), sorted_leaves(4 * n);
  for (size_t j = 0; j < n; ++j) {
    if (j > 0 && ids[j].first == ids[j - 1].first) { set_error("sp_order_batch: two orders share an order id"); return SP_ERR_BAD_ARGUMENT; }
    keys[j] = ids[j].first;
    std::memcpy(&sorted_leaves[4 * j], leaves + 4 * ids[j].second, 32);
  }
  tl_mark("ids sorted");
  bool joined = false;
  const std::function<bool()> all_verified = [&]() {
    gate.release();  // (an empty update asks before anything was enqueued)
    verifier.join();
    joined = true;
    if (vrc != SP_OK) return false;
    for (size_t i = 0; i < n; ++i)
      if (verdicts[i] != 1) return false;
    return true;
  };
  {
    TreeScope ts;
    rc = ts.open(tree);
    if (rc == SP_OK)
      rc = tree_update_locked(*ts.t, keys.data(), sorted_leaves.data(), n, old_root, new_root, tree_status, &all_verified,
                              &open_gate);
  }
  gate.release();  // (the tree failed before it enqueued anything)
  if (!joined && verifier.joinable()) verifier.join();
  // the error text is process-wide (set_error), so what the verifier thread reported is what sp_last_error says;
  // when both legs failed the tree's code wins and the text is whichever leg failed last
  if (rc != SP_OK) return rc;
  return vrc;
}

}  // extern "C"

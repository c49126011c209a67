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
#include <climits>
#include <cstring>
#include <map>
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
  const size_t src_bytes = (src.size() + 1) * sizeof(int2);
  SP_HIP(g_sparse_buf.reserve(emp_bytes + 2 * fb + src_bytes + 1024));
  char* b = (char*)g_sparse_buf.ptr;
  uint64_t* d_emp = (uint64_t*)b;
  uint64_t* d_a = (uint64_t*)(b + emp_bytes);
  uint64_t* d_b = (uint64_t*)(b + emp_bytes + fb);
  int2* d_src = (int2*)(b + emp_bytes + 2 * fb);
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
    for (size_t l = 0; l < level_cnt.size(); ++l) {
      const size_t m = level_cnt[l];
      // gathered mode: operand pointers come from src (children in `cur`, or this level's
      // empty-subtree root) inside the accumulate kernel - no separate gather pass
      rc = enqueue_pedersen(cur, 1, d_emp + 4 * l, 1, nxt, 1, nullptr, s.flag, m, 0, s, d_src + level_off[l]);
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
// the state: per level a host map  node index -> value  for every node that differs from the
// empty-subtree root of its level.  An update is ONE call: the host walks the induced subtree
// (merkle_tree.py:18-26), looks the untouched siblings up, ships leaves + siblings + child-index
// lists to the device, every level is one gathered launch (csrc/pedersen.hip), and the new node
// values come back in one copy to refresh the store.
struct FeltKey {
  uint64_t w[4];
};
// index -> felt, open addressing with linear probing (no allocation per node: an update inserts
// ~ n * height nodes, and std::unordered_map's per-node malloc dominated the call)
class NodeMap {
 public:
  const FeltKey* find(uint64_t key) const {
    if (slots_.empty()) return nullptr;
    for (size_t i = mix(key) & mask_;; i = (i + 1) & mask_) {
      const Slot& sl = slots_[i];
      if (!sl.used) return nullptr;
      if (sl.key == key) return &sl.value;
    }
  }
  void put(uint64_t key, const FeltKey& value) {
    if ((count_ + 1) * 2 > slots_.size()) grow();
    for (size_t i = mix(key) & mask_;; i = (i + 1) & mask_) {
      Slot& sl = slots_[i];
      if (!sl.used) { sl.used = true; sl.key = key; sl.value = value; ++count_; return; }
      if (sl.key == key) { sl.value = value; return; }
    }
  }
  size_t size() const { return count_; }

 private:
  struct Slot {
    uint64_t key = 0;
    FeltKey value{};
    bool used = false;
  };
  static size_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (size_t)(z ^ (z >> 31));
  }
  void grow() {
    std::vector<Slot> old;
    old.swap(slots_);
    slots_.assign(old.empty() ? 64 : old.size() * 2, Slot{});
    mask_ = slots_.size() - 1;
    count_ = 0;
    for (const Slot& sl : old)
      if (sl.used) put(sl.key, sl.value);
  }
  std::vector<Slot> slots_;
  size_t mask_ = 0, count_ = 0;
};
struct SparseTree {
  unsigned height = 0;
  uint64_t empty_leaf[4] = {0, 0, 0, 0};
  std::vector<NodeMap> nodes;  // [level][index], level 0 = leaves
};
// Runs fn(level) for level = 0..count-1 on a few host threads (levels touch disjoint node maps).
template <typename F>
static void for_levels_parallel(unsigned count, F fn) {
  const unsigned workers = count < 8 ? count : 8;
  if (workers <= 1) {
    for (unsigned l = 0; l < count; ++l) fn(l);
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve(workers);
  for (unsigned w = 0; w < workers; ++w)
    pool.emplace_back([=]() {
      for (unsigned l = w; l < count; l += workers) fn(l);
    });
  for (auto& th : pool) th.join();
}

static std::map<int, SparseTree> g_trees;
static int g_next_tree = 1;
static DeviceBuffer g_tree_buf;

namespace sp {
void release_tree_state() {
  g_trees.clear();
  g_tree_buf.release();
}
}  // namespace sp

extern "C" {

int sp_tree_create(unsigned height, const uint64_t* empty_leaf, int* tree) {
  SP_REQUIRE_READY();
  if (height < 1 || height > 64) { set_error("height must be in 1..64"); return SP_ERR_BAD_ARGUMENT; }
  // an out-of-range empty leaf would poison the cached empty-subtree roots (their chain flags
  // SP_HASH_OUT_OF_RANGE once, later calls hit the cache and never see the flag again)
  if (!felt_below_p(empty_leaf)) { set_error("empty leaf must be a field element (< p)"); return SP_ERR_BAD_ARGUMENT; }
  ctx_lock lk(ctx().mu);
  SparseTree t;
  t.height = height;
  std::memcpy(t.empty_leaf, empty_leaf, 32);
  t.nodes.resize(height + 1);
  const int id = g_next_tree++;
  g_trees.emplace(id, std::move(t));
  *tree = id;
  return SP_OK;
}

int sp_tree_destroy(int tree) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  if (g_trees.erase(tree) == 0) { set_error("unknown tree handle"); return SP_ERR_BAD_ARGUMENT; }
  return SP_OK;
}

int sp_tree_root(int tree, uint64_t* root) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  auto it = g_trees.find(tree);
  if (it == g_trees.end()) { set_error("unknown tree handle"); return SP_ERR_BAD_ARGUMENT; }
  SparseTree& t = it->second;
  if (const FeltKey* top = t.nodes[t.height].find(0)) {
    std::memcpy(root, top->w, 32);
    return SP_OK;
  }
  Scratch s;
  int rc = get_scratch_public(1, s, 0);
  if (rc != SP_OK) return rc;
  const std::vector<uint64_t>* emp = nullptr;
  rc = empty_roots(t.empty_leaf, s, &emp);
  if (rc != SP_OK) return rc;
  std::memcpy(root, emp->data() + 4 * t.height, 32);
  return SP_OK;
}

int sp_tree_get(int tree, const uint64_t* keys, size_t n, uint64_t* leaves) {
  SP_REQUIRE_READY();
  ctx_lock lk(ctx().mu);
  auto it = g_trees.find(tree);
  if (it == g_trees.end()) { set_error("unknown tree handle"); return SP_ERR_BAD_ARGUMENT; }
  const SparseTree& t = it->second;
  for (size_t i = 0; i < n; ++i) {
    if (t.height < 64 && (keys[i] >> t.height) != 0) { set_error("key out of range for height"); return SP_ERR_BAD_ARGUMENT; }
    const FeltKey* leaf = t.nodes[0].find(keys[i]);
    std::memcpy(leaves + 4 * i, leaf ? leaf->w : t.empty_leaf, 32);
  }
  return SP_OK;
}

int sp_tree_update(int tree, const uint64_t* keys, const uint64_t* leaves, size_t n, uint64_t* old_root,
                   uint64_t* new_root, uint8_t* status) {
  SP_REQUIRE_READY();
  Context& c = ctx();
  ctx_lock lk(c.mu);
  auto tit = g_trees.find(tree);
  if (tit == g_trees.end()) { set_error("unknown tree handle"); return SP_ERR_BAD_ARGUMENT; }
  SparseTree& t = tit->second;
  const unsigned height = t.height;
  for (size_t i = 0; i < n; ++i) {
    if (i > 0 && keys[i] <= keys[i - 1]) { set_error("keys must be strictly increasing"); return SP_ERR_BAD_ARGUMENT; }
    if (height < 64 && (keys[i] >> height) != 0) { set_error("key out of range for height"); return SP_ERR_BAD_ARGUMENT; }
  }
  int rc = sp_tree_root(tree, old_root);
  if (rc != SP_OK) return rc;
  if (status) *status = 0;
  if (n == 0) {
    std::memcpy(new_root, old_root, 32);
    return SP_OK;
  }
  // ---- host: induced subtree, siblings from the store ----
  // device felt buffer layout: [siblings][level 0 = new leaves][level 1][level 2]...; child indices
  // are absolute positions in that buffer, -1 = the level's empty-subtree root
  std::vector<uint64_t> sib;       // sibling values, 4 words each
  std::vector<int2> src;           // all levels
  std::vector<size_t> level_off, level_cnt;
  std::vector<std::vector<uint64_t>> level_idx(1, std::vector<uint64_t>(keys, keys + n));
  struct Pending { unsigned level; uint64_t child; bool left; size_t src_pos; };
  std::vector<Pending> want;       // sibling slots to patch once the sibling count is known
  for (unsigned l = 0; l < height; ++l) {
    const std::vector<uint64_t>& idx = level_idx[l];
    std::vector<uint64_t> nxt;
    nxt.reserve(idx.size());
    level_off.push_back(src.size());
    const size_t m = idx.size();
    for (size_t j = 0; j < m;) {
      int2 s2;
      const uint64_t parent = idx[j] >> 1;
      if ((idx[j] & 1) == 0) {
        s2.x = (int)j;  // relative to this level's array for now
        if (j + 1 < m && idx[j + 1] == idx[j] + 1) { s2.y = (int)(j + 1); j += 2; }
        else { s2.y = INT_MIN; want.push_back({l, idx[j] + 1, false, src.size()}); j += 1; }
      } else {
        s2.x = INT_MIN; s2.y = (int)j; want.push_back({l, idx[j] - 1, true, src.size()}); j += 1;
      }
      nxt.push_back(parent);
      src.push_back(s2);
    }
    level_cnt.push_back(nxt.size());
    level_idx.push_back(std::move(nxt));
  }
  // resolve siblings: stored value -> position in the sibling region, absent -> empty (-1).
  // `want` is grouped by level (it was filled level by level): the lookups of different levels go to
  // different node maps and run on a few host threads; placing the hits is serial and cheap.
  std::vector<const FeltKey*> hit(want.size(), nullptr);
  {
    std::vector<size_t> first(height + 1, want.size());
    for (size_t k = want.size(); k-- > 0;) first[want[k].level] = k;
    for (unsigned l = height; l-- > 0;)
      if (first[l] == want.size()) first[l] = first[l + 1];
    const SparseTree* tree_c = &t;
    for_levels_parallel(height, [&, tree_c](unsigned l) {
      const NodeMap& lvl = tree_c->nodes[l];
      for (size_t k = first[l]; k < first[l + 1]; ++k) hit[k] = lvl.find(want[k].child);
    });
  }
  std::vector<int> sib_pos(want.size(), -1);
  for (size_t k = 0; k < want.size(); ++k) {
    if (hit[k]) {
      sib_pos[k] = (int)(sib.size() / 4);
      sib.insert(sib.end(), hit[k]->w, hit[k]->w + 4);
    }
  }
  const size_t n_sib = sib.size() / 4;
  size_t total_nodes = n;
  for (size_t l = 0; l < level_cnt.size(); ++l) total_nodes += level_cnt[l];
  if (n_sib + total_nodes >= (size_t)INT_MAX) { set_error("update too large"); return SP_ERR_BAD_ARGUMENT; }
  // absolute positions: level l array starts at base[l]
  std::vector<size_t> base(height + 1);
  base[0] = n_sib;
  for (unsigned l = 0; l < height; ++l) base[l + 1] = base[l] + (l == 0 ? n : level_cnt[l - 1]);
  {
    size_t k = 0;
    for (unsigned l = 0; l < height; ++l) {
      for (size_t q = level_off[l]; q < level_off[l] + level_cnt[l]; ++q) {
        int2& s2 = src[q];
        if (s2.x != INT_MIN) s2.x += (int)base[l];
        if (s2.y != INT_MIN) s2.y += (int)base[l];
      }
    }
    for (k = 0; k < want.size(); ++k) {
      int2& s2 = src[want[k].src_pos];
      (want[k].left ? s2.x : s2.y) = sib_pos[k];  // -1 = empty, else index into the sibling region
    }
  }
  // ---- device ----
  const size_t emp_bytes = ((size_t)height + 1) * 32;
  const size_t felt_bytes = (n_sib + total_nodes) * 32;
  const size_t src_bytes = src.size() * sizeof(int2);
  SP_HIP(g_tree_buf.reserve(emp_bytes + felt_bytes + src_bytes + 1024));
  char* b = (char*)g_tree_buf.ptr;
  uint64_t* d_emp = (uint64_t*)b;
  uint64_t* d_felts = (uint64_t*)(b + emp_bytes);
  int2* d_src = (int2*)(b + emp_bytes + felt_bytes);
  Scratch s;
  rc = get_scratch_public(n, s, 0);
  if (rc != SP_OK) return rc;
  SP_HIP(hipMemsetAsync(s.flag, 0, sizeof(unsigned), 0));
  const std::vector<uint64_t>* emp = nullptr;
  rc = empty_roots(t.empty_leaf, s, &emp);
  if (rc != SP_OK) return rc;
  SP_HIP(hipMemcpy(d_emp, emp->data(), emp_bytes, hipMemcpyHostToDevice));
  if (n_sib) SP_HIP(hipMemcpy(d_felts, sib.data(), n_sib * 32, hipMemcpyHostToDevice));
  SP_HIP(hipMemcpy(d_felts + 4 * base[0], leaves, n * 32, hipMemcpyHostToDevice));
  SP_HIP(hipMemcpy(d_src, src.data(), src_bytes, hipMemcpyHostToDevice));
  for (unsigned l = 0; l < height; ++l) {
    rc = enqueue_pedersen(d_felts, 1, d_emp + 4 * l, 1, d_felts + 4 * base[l + 1], 1, nullptr, s.flag, level_cnt[l],
                          0, s, d_src + level_off[l]);
    if (rc != SP_OK) return rc;
  }
  SP_HIP(hipDeviceSynchronize());
  unsigned f = 0;
  SP_HIP(hipMemcpy(&f, s.flag, sizeof(unsigned), hipMemcpyDeviceToHost));
  if (status) *status = (uint8_t)f;
  if (f != 0) {  // an input out of range or an unhashable pair: the tree is left as it was
    std::memcpy(new_root, old_root, 32);
    return SP_OK;
  }
  std::vector<uint64_t> fresh((total_nodes - n) * 4);
  SP_HIP(hipMemcpy(fresh.data(), d_felts + 4 * base[1], fresh.size() * 8, hipMemcpyDeviceToHost));
  // ---- refresh the store (one node map per level: levels in parallel) ----
  std::vector<size_t> fresh_off(height + 1, 0);
  for (unsigned l = 0; l < height; ++l) fresh_off[l + 1] = fresh_off[l] + level_idx[l + 1].size();
  for_levels_parallel(height + 1, [&](unsigned lvl_no) {
    NodeMap& lvl = t.nodes[lvl_no];
    if (lvl_no == 0) {
      for (size_t i = 0; i < n; ++i) {
        FeltKey v;
        std::memcpy(v.w, leaves + 4 * i, 32);
        lvl.put(keys[i], v);
      }
      return;
    }
    const std::vector<uint64_t>& idx = level_idx[lvl_no];
    const uint64_t* vals = fresh.data() + 4 * fresh_off[lvl_no - 1];
    for (size_t q = 0; q < idx.size(); ++q) {
      FeltKey v;
      std::memcpy(v.w, vals + 4 * q, 32);
      lvl.put(idx[q], v);
    }
  });
  const size_t pos = fresh_off[height];
  std::memcpy(new_root, fresh.data() + 4 * (pos - 1), 32);
  return SP_OK;
}

}  // extern "C"

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
#include <map>
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

extern "C" int sp_merkle_sparse_root(const uint64_t* keys, const uint64_t* leaves, size_t n,
                                     unsigned height, const uint64_t* empty_leaf, uint64_t* root,
                                     uint8_t* status) {
  SP_REQUIRE_READY();
  if (height > 64) { set_error("height must be <= 64"); return SP_ERR_BAD_ARGUMENT; }
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
  // empty-subtree roots: empties[k+1] = H(empties[k], empties[k]); 64 sequential hashes the first
  // time a given empty leaf is seen, then served from the host-side cache
  {
    std::vector<uint64_t> key(empty_leaf, empty_leaf + 4);
    auto it = g_empty_cache.find(key);
    if (it == g_empty_cache.end()) {
      SP_HIP(hipMemcpy(d_emp, empty_leaf, 32, hipMemcpyHostToDevice));
      uint64_t* d_full = nullptr;
      SP_HIP(hipMalloc(&d_full, 65 * 32));
      SP_HIP(hipMemcpy(d_full, empty_leaf, 32, hipMemcpyHostToDevice));
      for (unsigned k2 = 0; k2 < 64; ++k2) {
        rc = enqueue_pedersen(d_full + 4 * k2, 1, d_full + 4 * k2, 1, d_full + 4 * (k2 + 1), 1, nullptr, s.flag, 1, 0, s, nullptr);
        if (rc != SP_OK) { (void)hipFree(d_full); return rc; }
      }
      std::vector<uint64_t> all(65 * 4);
      SP_HIP(hipDeviceSynchronize());
      SP_HIP(hipMemcpy(all.data(), d_full, 65 * 32, hipMemcpyDeviceToHost));
      (void)hipFree(d_full);
      it = g_empty_cache.emplace(key, all).first;
    }
    SP_HIP(hipMemcpy(d_emp, it->second.data(), emp_bytes, hipMemcpyHostToDevice));
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

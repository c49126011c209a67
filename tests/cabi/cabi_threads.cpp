// Host-concurrency stress of the C ABI from a native program - the target of the sanitizer builds
// (csrc/Makefile SAN=address / SAN=thread, tools/run_sanitizers.sh; SURVEY section 5 "Race detection /
// sanitizers").  The library has real host-side concurrency: sixteen host lanes, a recursive lock, the key cache
// with generations and eviction, per-stream scratch maps, persistent trees with their own mutexes and a
// std::thread inside sp_order_batch.  Phase A computes every answer on ONE thread; phase B asks the same
// questions from eight threads at once (hash batches, chains, rebuilds, ladder / AUTO / keyed verification on a
// 16-slot key cache with resets racing, persistent trees, sp_order_batch, both signers) and requires identical
// answers.  Correctness of the answers themselves is the parity suite's job; this program checks that
// concurrency changes nothing and gives ASan / UBSan / TSan something to look at.
//
//   cabi_threads [contexts=1|2] [iterations=3]      exit code 0 and "cabi_threads ok" on success
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "starkperp.h"

namespace {

struct Rng {  // splitmix64
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
};
using Felts = std::vector<uint64_t>;  // 4 little-endian words per felt
Felts random_felts(Rng& g, size_t n, int bits = 250) {
  Felts v(4 * n);
  for (size_t i = 0; i < n; ++i) {
    for (int k = 0; k < 4; ++k) {
      const int lo = 64 * k;  // keep bits [0, bits)
      v[4 * i + k] = bits >= lo + 64 ? g.next() : (bits > lo ? g.next() & ((1ull << (bits - lo)) - 1) : 0);
    }
    if ((v[4 * i] | v[4 * i + 1] | v[4 * i + 2] | v[4 * i + 3]) == 0) v[4 * i] = 1;
  }
  return v;
}
Felts slice(const Felts& v, size_t off, size_t n) { return Felts(v.begin() + 4 * off, v.begin() + 4 * (off + n)); }

std::atomic<int> g_failures{0};
void fail(const char* what, int rc) {
  std::fprintf(stderr, "cabi_threads: %s (rc=%d): %s\n", what, rc, sp_last_error());
  g_failures.fetch_add(1);
}
#define CHECK_RC(call, what)              \
  do {                                    \
    const int rc__ = (call);              \
    if (rc__ != SP_OK) { fail(what, rc__); return; } \
  } while (0)
#define CHECK_EQ(a, b, what) \
  do {                       \
    if (!((a) == (b))) { fail(what, 0); return; } \
  } while (0)

// ---- the questions ------------------------------------------------------------------------------------------
struct Data {
  size_t n_hash = 3000, n_keys = 24, n_sig = 768, n_big = 5000, n_tree = 300;
  Felts hx, hy, chain_words, leaves;               // hashing
  Felts priv, qx, qy, z, r, s, sig_qx, sig_qy;      // ECDSA (every fifth signature corrupted)
  Felts big_z, big_d;                               // the compacted signer (n >= 4096)
  Felts tree_keys[3], tree_vals[3], empty_leaf;     // three update batches of a height-64 tree
  Felts ord_z, ord_r, ord_s, ord_qx, ord_leaves;    // sp_order_batch: depth-1 chains (the words ARE the hashes)
  size_t n_ord = 256;
};
struct Answers {
  Felts hashes, chains, root;
  std::vector<uint8_t> verdicts, verdicts_point;
  Felts sig_r, sig_s, big_r, big_s;
  Felts tree_roots;  // after each of the three batches
  Felts tree_got;    // sp_tree_get of batch 0's keys after batch 2
  Felts ord_new_root;
  std::vector<uint8_t> ord_verdicts;
};

void ask_hashing(const Data& d, Answers& a) {
  a.hashes.assign(4 * d.n_hash, 0);
  std::vector<uint8_t> st(d.n_hash);
  CHECK_RC(sp_pedersen_batch(d.hx.data(), d.hy.data(), a.hashes.data(), st.data(), d.n_hash), "pedersen_batch");
  for (uint8_t v : st) CHECK_EQ(v, (uint8_t)SP_HASH_OK, "hash status");
  a.chains.assign(4 * 40, 0);
  uint8_t cst = 0;
  CHECK_RC(sp_pedersen_chains(d.chain_words.data(), 40, 5, a.chains.data(), &cst), "pedersen_chains");
  a.root.assign(4, 0);
  CHECK_RC(sp_merkle_root(d.leaves.data(), 9, a.root.data(), nullptr, &cst), "merkle_root");
}
// The signatures whose key index lies in [k0, k1): the two AUTO threads of phase B bring twelve keys each to a
// 16-slot cache, so each of them finds the cache full of the other's keys again and again (eviction = a new generation).
Data key_range(const Data& d, const Answers* ref, Answers* ref_out, size_t k0, size_t k1) {
  Data o;
  o.n_sig = 0;
  for (size_t i = 0; i < d.n_sig; ++i) {
    const size_t k = i % d.n_keys;
    if (k < k0 || k >= k1) continue;
    for (const Felts* src : {&d.z, &d.r, &d.s, &d.sig_qx, &d.sig_qy}) {
      Felts& dst = src == &d.z ? o.z : src == &d.r ? o.r : src == &d.s ? o.s : src == &d.sig_qx ? o.sig_qx : o.sig_qy;
      dst.insert(dst.end(), src->begin() + 4 * i, src->begin() + 4 * i + 4);
    }
    if (ref) ref_out->verdicts.push_back(ref->verdicts[i]);
    ++o.n_sig;
  }
  return o;
}
void ask_verify(const Data& d, Answers& a, int policy_hint) {
  // policy_hint 0: whatever the process-wide policy is (AUTO in phase B); 1: explicit keyed entry point
  a.verdicts.assign(d.n_sig, 0xEE);
  if (policy_hint == 1) {
    const int rc = sp_ecdsa_verify_batch_keyed(d.z.data(), d.r.data(), d.s.data(), d.sig_qx.data(), nullptr,
                                               a.verdicts.data(), d.n_sig);
    if (rc == SP_ERR_CACHE_FULL) {  // a full 4-slot cache is a legal answer of the explicit entry point
      a.verdicts.clear();
      return;
    }
    CHECK_RC(rc, "verify_batch_keyed");
  } else {
    CHECK_RC(sp_ecdsa_verify_batch(d.z.data(), d.r.data(), d.s.data(), d.sig_qx.data(), nullptr, a.verdicts.data(),
                                   d.n_sig), "verify_batch");
  }
  a.verdicts_point.assign(d.n_sig, 0xEE);
  CHECK_RC(sp_ecdsa_verify_batch(d.z.data(), d.r.data(), d.s.data(), d.sig_qx.data(), d.sig_qy.data(),
                                 a.verdicts_point.data(), d.n_sig), "verify_batch (point keys)");
}
void ask_sign(const Data& d, Answers& a) {
  a.sig_r.assign(4 * d.n_sig, 0);
  a.sig_s.assign(4 * d.n_sig, 0);
  std::vector<uint8_t> st(d.n_sig);
  Felts dd(4 * d.n_sig);
  for (size_t i = 0; i < d.n_sig; ++i) std::memcpy(&dd[4 * i], &d.priv[4 * (i % d.n_keys)], 32);
  CHECK_RC(sp_ecdsa_sign_rfc6979_batch(d.z.data(), dd.data(), nullptr, a.sig_r.data(), a.sig_s.data(), st.data(),
                                       d.n_sig), "sign_rfc6979");
  a.big_r.assign(4 * d.n_big, 0);
  a.big_s.assign(4 * d.n_big, 0);
  std::vector<uint8_t> bst(d.n_big);
  CHECK_RC(sp_ecdsa_sign_rfc6979_batch(d.big_z.data(), d.big_d.data(), nullptr, a.big_r.data(), a.big_s.data(),
                                       bst.data(), d.n_big), "sign_rfc6979 (compacted)");
}
void ask_tree(const Data& d, Answers& a, int context) {
  int tree = -1;
  CHECK_RC(context < 0 ? sp_tree_create(64, d.empty_leaf.data(), &tree)
                       : sp_tree_create_on(context, 64, d.empty_leaf.data(), &tree), "tree_create");
  a.tree_roots.assign(12, 0);
  Felts old(4);
  std::vector<uint8_t> st(1);
  for (int b = 0; b < 3; ++b) {
    const int rc = sp_tree_update(tree, d.tree_keys[b].data(), d.tree_vals[b].data(), d.n_tree, old.data(),
                                  &a.tree_roots[4 * b], st.data());
    if (rc != SP_OK) { fail("tree_update", rc); sp_tree_destroy(tree); return; }
  }
  a.tree_got.assign(4 * d.n_tree, 0);
  int rc = sp_tree_get(tree, d.tree_keys[0].data(), d.n_tree, a.tree_got.data());
  if (rc != SP_OK) fail("tree_get", rc);
  Felts root(4);
  rc = sp_tree_root(tree, root.data());
  if (rc != SP_OK || std::memcmp(root.data(), &a.tree_roots[8], 32) != 0) fail("tree_root", rc);
  CHECK_RC(sp_tree_destroy(tree), "tree_destroy");
}
void ask_order_batch(const Data& d, Answers& a) {
  int tree = -1;
  CHECK_RC(sp_tree_create(64, d.empty_leaf.data(), &tree), "tree_create (orders)");
  Felts zout(4 * d.n_ord), old(4);
  a.ord_new_root.assign(4, 0);
  a.ord_verdicts.assign(d.n_ord, 0xEE);
  uint8_t tst = 0xEE;
  const int rc = sp_order_batch(d.ord_z.data(), 1, d.n_ord, d.ord_r.data(), d.ord_s.data(), d.ord_qx.data(), nullptr,
                                tree, d.ord_leaves.data(), 187, zout.data(), a.ord_verdicts.data(), old.data(),
                                a.ord_new_root.data(), &tst);
  if (rc == SP_ERR_CACHE_FULL) a.ord_new_root.clear();  // the explicit keyed path does not evict: a legal answer while
                                                        // other threads hold the 16 slots (the caller resets or retries)
  else if (rc != SP_OK) fail("order_batch", rc);
  else if (tst != 0) fail("order_batch: not committed", tst);
  else if (zout != d.ord_z) fail("order_batch: z", 0);
  CHECK_RC(sp_tree_destroy(tree), "tree_destroy (orders)");
}

}  // namespace

// Proof that the instrumentation is live (tools/run_sanitizers.sh expects a finding from each): a deliberate data
// race for TSan, a deliberate heap overflow for ASan.  Neither touches the library.
int selftest(const std::string& what) {
  if (what == "selftest-race") {
    int counter = 0;  // unsynchronised on purpose
    std::thread a([&] { for (int i = 0; i < 100000; ++i) counter = counter + 1; });
    std::thread b([&] { for (int i = 0; i < 100000; ++i) counter = counter + 1; });
    a.join();
    b.join();
    std::printf("selftest-race done (%d)\n", counter);
    return 0;
  }
  volatile size_t n = 8;
  char* p = new char[n];
  p[n] = 1;  // one past the end, on purpose
  std::printf("selftest-overflow done (%d)\n", (int)p[n]);
  delete[] p;
  return 0;
}
extern "C" void __lsan_do_leak_check() __attribute__((weak));

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]).rfind("selftest", 0) == 0) return selftest(argv[1]);
  const int contexts = argc > 1 ? std::atoi(argv[1]) : 1;
  const int iterations = argc > 2 ? std::atoi(argv[2]) : 3;
  setenv("STARKPERP_KEY_CACHE_SLOTS", "16", 1);  // 24 keys fight for 16 slots: eviction, fall-backs, generations
  std::printf("%s\n", sp_build_info());
  if (contexts == 2) {
    const int ids[2] = {0, 0};  // two contexts on the one GPU: lanes go round-robin over them
    if (sp_init_devices(2, ids, 13) != SP_OK) { std::fprintf(stderr, "sp_init_devices: %s\n", sp_last_error()); return 2; }
  } else if (sp_init(0, 13) != SP_OK) {
    std::fprintf(stderr, "sp_init: %s\n", sp_last_error());
    return 2;
  }

  Rng g{2026};
  Data d;
  d.hx = random_felts(g, d.n_hash);
  d.hy = random_felts(g, d.n_hash);
  d.chain_words = random_felts(g, 40 * 5);
  d.leaves = random_felts(g, 512);
  d.priv = random_felts(g, d.n_keys, 249);
  d.qx.assign(4 * d.n_keys, 0);
  d.qy.assign(4 * d.n_keys, 0);
  {
    std::vector<uint8_t> st(d.n_keys);
    if (sp_public_key_batch(d.priv.data(), d.qx.data(), d.qy.data(), st.data(), d.n_keys) != SP_OK) return 3;
  }
  d.z = random_felts(g, d.n_sig, 250);
  d.big_z = random_felts(g, d.n_big, 250);
  d.big_d = random_felts(g, d.n_big, 249);
  d.empty_leaf.assign(4, 0);
  for (int b = 0; b < 3; ++b) {
    d.tree_keys[b].resize(d.n_tree);  // strictly increasing 64-bit leaf indices
    for (size_t i = 0; i < d.n_tree; ++i) d.tree_keys[b][i] = g.next();
    std::sort(d.tree_keys[b].begin(), d.tree_keys[b].end());
    for (size_t i = 1; i < d.n_tree; ++i)
      if (d.tree_keys[b][i] <= d.tree_keys[b][i - 1]) d.tree_keys[b][i] = d.tree_keys[b][i - 1] + 1;
    d.tree_vals[b] = random_felts(g, d.n_tree, 64);
  }

  // ---- phase A: one thread -------------------------------------------------------------------------------
  Answers ref;
  sp_ecdsa_set_verify_policy(SP_VERIFY_POLICY_LADDER);
  ask_hashing(d, ref);
  ask_sign(d, ref);
  d.r = ref.sig_r;
  d.s = ref.sig_s;
  d.sig_qx.resize(4 * d.n_sig);
  d.sig_qy.resize(4 * d.n_sig);
  for (size_t i = 0; i < d.n_sig; ++i) {
    std::memcpy(&d.sig_qx[4 * i], &d.qx[4 * (i % d.n_keys)], 32);
    std::memcpy(&d.sig_qy[4 * i], &d.qy[4 * (i % d.n_keys)], 32);
    if (i % 5 == 4) d.s[4 * i] ^= 2;  // corrupted signature: False, not an assertion
  }
  ask_verify(d, ref, 0);
  size_t n_true = 0;
  for (uint8_t v : ref.verdicts) n_true += v == SP_VERIFY_TRUE;
  if (n_true != d.n_sig - d.n_sig / 5 || ref.verdicts != ref.verdicts_point) {
    std::fprintf(stderr, "cabi_threads: reference verdicts implausible (%zu true of %zu)\n", n_true, d.n_sig);
    return 4;
  }
  ask_tree(d, ref, -1);
  // orders: valid signatures only (the batch must commit), distinct order ids = top 64 bits of z < 2^251
  d.ord_z = slice(d.z, 0, d.n_ord);
  {
    Felts dd(4 * d.n_ord);
    for (size_t i = 0; i < d.n_ord; ++i) std::memcpy(&dd[4 * i], &d.priv[4 * (i % 4)], 32);  // four accounts
    d.ord_r.assign(4 * d.n_ord, 0);
    d.ord_s.assign(4 * d.n_ord, 0);
    std::vector<uint8_t> st(d.n_ord);
    if (sp_ecdsa_sign_rfc6979_batch(d.ord_z.data(), dd.data(), nullptr, d.ord_r.data(), d.ord_s.data(), st.data(),
                                    d.n_ord) != SP_OK) return 5;
    d.ord_qx.resize(4 * d.n_ord);
    for (size_t i = 0; i < d.n_ord; ++i) std::memcpy(&d.ord_qx[4 * i], &d.qx[4 * (i % 4)], 32);
    d.ord_leaves = random_felts(g, d.n_ord, 64);
  }
  sp_ecdsa_set_verify_policy(SP_VERIFY_POLICY_AUTO);
  ask_order_batch(d, ref);
  if (g_failures.load() != 0) return 6;

  Answers lo_ref, hi_ref, few_ref;
  const Data lo_keys = key_range(d, &ref, &lo_ref, 0, 12), hi_keys = key_range(d, &ref, &hi_ref, 12, 24);
  const Data few_keys = key_range(d, &ref, &few_ref, 0, 4);
  std::atomic<int> orders_committed{0};

  // ---- phase B: eight threads ----------------------------------------------------------------------------
  auto worker = [&](int id) {
    for (int it = 0; it < iterations; ++it) {
      Answers a;
      switch (id % 8) {
        case 0:
        case 1:
          ask_hashing(d, a);
          if (a.hashes != ref.hashes || a.chains != ref.chains || a.root != ref.root) fail("hashing differs", id);
          break;
        case 2:
        case 3: {
          // AUTO: tables when the keys are known, ladder otherwise, eviction when the other thread's keys fill the cache
          const Data& part = id % 8 == 2 ? lo_keys : hi_keys;
          const Answers& want = id % 8 == 2 ? lo_ref : hi_ref;
          for (int rep = 0; rep < 3; ++rep) {
            ask_verify(part, a, 0);
            if (a.verdicts != want.verdicts || a.verdicts_point != want.verdicts) fail("AUTO verdicts differ", id);
          }
          break;
        }
        case 4: {
          ask_verify(few_keys, a, 1);  // explicit keyed entry point (4 keys) on the 16-slot cache, with resets racing
          if (!a.verdicts.empty() && a.verdicts != few_ref.verdicts) fail("keyed verdicts differ", id);
          if (it % 2 == 1) sp_ecdsa_key_cache_reset();
          uint32_t slots[4];
          const int rc = sp_ecdsa_register_keys(d.qx.data(), nullptr, 4, slots);
          if (rc != SP_OK && rc != SP_ERR_CACHE_FULL) fail("register_keys", rc);
          size_t cap = 0, used = 0;
          if (sp_ecdsa_key_cache_info(&cap, &used) != SP_OK || used > cap) fail("key_cache_info", 0);
          break;
        }
        case 5:
          ask_tree(d, a, contexts == 2 ? it % 2 : -1);
          if (a.tree_roots != ref.tree_roots || a.tree_got != ref.tree_got) fail("tree differs", id);
          break;
        case 6:
          ask_order_batch(d, a);
          if (!a.ord_new_root.empty() && (a.ord_new_root != ref.ord_new_root || a.ord_verdicts != ref.ord_verdicts))
            fail("order batch differs", id);
          if (!a.ord_new_root.empty()) orders_committed.fetch_add(1);
          break;
        case 7:
          ask_sign(d, a);
          if (a.sig_r != ref.sig_r || a.sig_s != ref.sig_s || a.big_r != ref.big_r || a.big_s != ref.big_s)
            fail("signatures differ", id);
          break;
      }
    }
  };
  std::vector<std::thread> threads;
  for (int i = 0; i < 8; ++i) threads.emplace_back(worker, i);
  for (auto& t : threads) t.join();
  int dev = -1;
  uint64_t calls = 0;
  for (int c = 0; c < sp_device_count(); ++c) {
    sp_context_info(c, &dev, &calls);
    std::printf("context %d: device %d, %llu host-lane calls\n", c, dev, (unsigned long long)calls);
  }
  sp_shutdown();
  if (g_failures.load() != 0) {
    std::fprintf(stderr, "cabi_threads: %d failure(s)\n", g_failures.load());
    return 1;
  }
  // Under AMD's ASan the HSA runtime's exit-time destructors can trip a CHECK of the sanitizer's own device
  // allocator ("dev_runtime_unloaded_", inside __cxa_finalize of libamdhip64 - no frame of this library or program):
  // run the leak check now and leave without the static destructors when asked to.
  const bool fast_exit = getenv("CABI_FAST_EXIT") != nullptr;
  if (fast_exit && __lsan_do_leak_check) __lsan_do_leak_check();
  std::printf("cabi_threads ok: 8 threads x %d iterations, %d context(s), %d of %d order batches committed (the others met "
              "a full key cache)\n", iterations, contexts, orders_committed.load(), iterations);
  std::fflush(stdout);
  if (fast_exit) _exit(0);
  return 0;
}

/*
 * libstarkperp - C ABI of the MI355X hot path for the StarkEx-Perpetual crypto builtins.
 *
 * The reference (starkware-libs/stark-perpetual) has no FFI: its hot path is the Python module
 * src/starkware/crypto/signature/signature.py.  Each entry point below names the reference
 * function (file:line, relative to /root/reference/src) whose arithmetic it replaces; the Python
 * host package (stark-perpetual_amd/starkperp) binds them with ctypes and re-exposes the
 * reference's module API unchanged.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - A field element / scalar ("felt") is 32 bytes: four little-endian uint64 limbs, plain
 *     (non-Montgomery) integer value.  Arrays are contiguous, n * 32 bytes, caller-owned.
 *   - Functions ending in _dev take DEVICE pointers (HBM resident, 32-byte aligned) and a
 *     hipStream_t passed as void* (NULL = the null stream); they enqueue work and return without
 *     synchronising.  The others take HOST pointers and are synchronous.
 *   - Return value: 0 on success, negative on library/runtime error (sp_last_error() has the
 *     text).  Data-dependent outcomes are reported per item in a uint8_t status array.
 *   - There is no CPU fallback: without a visible gfx950 device sp_init fails and every compute
 *     entry point returns SP_ERR_NOT_INITIALISED.
 *   - Ownership: the library never keeps a caller pointer after a host-pointer call returns; a _dev
 *     call's buffers must stay valid until the work enqueued on its stream has completed.
 *   - Threading: every entry point may be called from any host thread.  One recursive lock
 *     serialises the host-side bookkeeping; scratch, window tables of the verifier and NTT work
 *     columns are per stream, so _dev calls on different streams overlap on the device.  The
 *     host-pointer batches that carry no shared state - sp_pedersen_batch, sp_pedersen_chains,
 *     sp_ecdsa_verify_batch (per-signature ladder), sp_ecdsa_sign_batch, sp_ecdsa_sign_rfc6979_batch,
 *     sp_public_key_batch -
 *     run on one of 16 host lanes (own stream, own staging buffer) and hold the lock only while a
 *     kernel is enqueued: calls from different threads overlap on the device - on the devices, after
 *     sp_init_devices - (a seventeenth concurrent caller waits for a lane).  Keyed verification runs on a lane too and holds the
 *     lock to register keys and to enqueue; a persistent tree has its own stream and mutex (below).  The remaining
 *     host-pointer calls (the scalar chain calls, sp_merkle_root / _sparse_root, key registration) hold the lock for
 *     their whole staged round trip.
 *     Each entry point binds the device given to sp_init for its duration and restores the calling
 *     thread's current HIP device on return.
 */
#ifndef STARKPERP_H
#define STARKPERP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_OK 0
#define SP_ERR_NOT_INITIALISED (-1)
#define SP_ERR_HIP (-2)
#define SP_ERR_BAD_ARGUMENT (-3)
#define SP_ERR_TABLE_BUILD (-4)
#define SP_ERR_CACHE_FULL (-5) /* sp_ecdsa_register_keys: no free key-table slot */

/* per-item status of sp_pedersen_* (signature.py:300-318) */
#define SP_HASH_OK 0
#define SP_HASH_OUT_OF_RANGE 1 /* an input was not in [0, p): signature.py:307 assertion */
#define SP_HASH_UNHASHABLE 2   /* exceptional point collision: signature.py:313 ("Unhashable input.") */
#define SP_TREE_NOT_COMMITTED 0x80 /* sp_order_batch: a signature did not verify - the tree was left as it was */

/* per-item result of sp_ecdsa_verify_* (signature.py:217-260) */
#define SP_VERIFY_FALSE 0
#define SP_VERIFY_TRUE 1
#define SP_VERIFY_ASSERT_S 2     /* signature.py:219  assert 1 <= s < EC_ORDER */
#define SP_VERIFY_ASSERT_R 3     /* signature.py:225  assert 1 <= r < 2**251 */
#define SP_VERIFY_ASSERT_W 4     /* signature.py:226  assert 1 <= w < 2**251 */
#define SP_VERIFY_ASSERT_MSG 5   /* signature.py:227  assert 0 <= msg_hash < 2**251 */
#define SP_VERIFY_ASSERT_CURVE 6 /* signature.py:241  assert is_point_on_curve */
#define SP_VERIFY_STALE_SLOT 7   /* sp_ecdsa_verify_keyed_dev only: the slot handle is not from the current
                                    cache generation (handed out before sp_ecdsa_key_cache_reset) or was never
                                    handed out; no verdict was computed for this item */

/* per-item status of sp_ecdsa_sign_* (signature.py:137-173) */
#define SP_SIGN_OK 0
#define SP_SIGN_RETRY 1          /* the k was rejected (signature.py:158-170): draw the next k */
#define SP_SIGN_BAD_INPUT 2      /* msg_hash >= 2**251 (signature.py:141) or key/k out of range */

/* ---- lifecycle ---------------------------------------------------------------------------- */
/* Selects the HIP device and builds the window tables of the Pedersen constant points
 * (signature.py:43 CONSTANT_POINTS; table structure nothing_up_my_sleeve_gen.py:88-90) and of
 * EC_GEN (signature.py:56) in HBM.  window_bits = 0 picks the default (21: 2^20 + 22 x 2^21 Pedersen entries + 12 x 2^21
 * EC_GEN entries, 64 B each = 4.3 GiB of tables; 26 = 75 GiB, 19 entries per hash, ~15 % faster hashing: what the
 * benchmark runs with).  The first sp_init of a process also runs 64 asynchronous 128-KiB copies each way (~8 ms):
 * the HIP runtime takes a one-time 6 - 7 ms step at about the 43rd sizeable copy of a process, which would
 * otherwise land inside a caller's batch (STARKPERP_NO_COPY_WARMUP=1 skips it).  Idempotent. */
int sp_init(int device, int window_bits);
/* Several devices in one process (the SURVEY 8(b) shape `sp_init(n_devices, device_ids)`): one context - its own
 * window tables - per entry of device_ids; context 0 is the PRIMARY.  What runs where:
 *   - the stateless batches run on ANY context: sp_pedersen_batch, sp_ecdsa_verify_batch (ladder),
 *     sp_ecdsa_sign_batch, sp_ecdsa_sign_rfc6979_batch, sp_public_key_batch take the context of the host lane
 *     they are handed (lanes go round-robin over the contexts, so concurrent host threads spread over the
 *     devices), and one call of sp_pedersen_batch / sp_ecdsa_verify_batch with 16384 items or more is cut into
 *     one contiguous slice per context, the slices running side by side on host threads of the library; sp_pedersen_batch_dev, sp_pedersen_chains_dev, sp_merkle_build_dev, sp_merkle_forest_dev,
 *     sp_commit_rows_dev, sp_ecdsa_verify_batch_dev, sp_ecdsa_sign_batch_dev, sp_ecdsa_sign_rfc6979_batch_dev and
 *     sp_public_key_batch_dev run on the device their pointers live on (the stream
 *     must belong to that device);
 *   - everything with state - persistent trees, key tables, the prover entry points (twiddle tables, witness
 *     scratch) and the remaining host-pointer calls - stays on the primary device.
 * The reference's deployment model here is one process per GPU (torch.distributed ranks, bench.py); this
 * entry point is for a C caller that drives a node from one process.  Idempotent for the same layout;
 * another layout needs sp_shutdown first.  A device may be listed twice (two contexts on one GPU: tests). */
int sp_init_devices(int n_devices, const int* device_ids, int window_bits);
int sp_device_count(void);  /* contexts initialised (0 before sp_init) */
/* device index and host-lane calls served so far by context `index` */
int sp_context_info(int index, int* device, uint64_t* host_calls);
void sp_shutdown(void);
const char* sp_last_error(void);
int sp_is_initialised(void);
/* window width in use and bytes of HBM held by the tables */
int sp_window_bits(void);
size_t sp_table_bytes(void);
int sp_synchronize(void* stream);
/* Build provenance (no reference counterpart: the reference is interpreted Python): one static line naming the
 * compiler, the HIP version, the offload architecture, the compile date of the library and the sanitizer it was
 * instrumented with, if any - bench.py prints it next to the sha256 of the .so, so that the binary a record was
 * produced by can be identified.  Callable before sp_init. */
const char* sp_build_info(void);

/* Measurement aid: between sp_profile_begin and sp_profile_end every launch of the dominant kernel
 * (ped_accumulate_kernel) is bracketed by HIP events on the stream it is launched on; _end returns
 * the summed kernel time, the number of launches and the number of hashes they processed. */
int sp_profile_begin(size_t max_launches);
int sp_profile_end(double* total_ms, uint64_t* launches, uint64_t* units);

/* ---- Pedersen hash: pedersen_hash(x, y) signature.py:296-318 -------------------------------- */
int sp_pedersen_batch(const uint64_t* x, const uint64_t* y, uint64_t* out, uint8_t* status, size_t n);
int sp_pedersen_batch_dev(const uint64_t* x, const uint64_t* y, uint64_t* out, uint8_t* status,
                          size_t n, void* stream);
/* pedersen_hash_as_point(x, y) signature.py:300-318: the full affine point (testing helper). */
int sp_pedersen_point_batch(const uint64_t* x, const uint64_t* y, uint64_t* out_x, uint64_t* out_y,
                            uint8_t* status, size_t n);
/* left fold h = H(h, e_i) starting from h = e_0: the hash-chain shape of
 * perpetual_messages.py:279-286 and position/hash.cairo:22-43.  n_elems >= 1. */
int sp_pedersen_chain(const uint64_t* elems, size_t n_elems, uint64_t* out, uint8_t* status);
/* right fold h = H(e_i, h) starting from h = e_{n-1}: cairo-lang compute_hash_chain, the consumer
 * behind starkware/cairo/bootloaders/program_hash_test_utils.py:9. */
int sp_pedersen_chain_right(const uint64_t* elems, size_t n_elems, uint64_t* out, uint8_t* status);
/* width independent chains of equal depth, element j of chain i at elems[(j*width + i)*4]:
 * out[i] = H(...H(H(e0,e1),e2)...,e_{depth-1}) */
int sp_pedersen_chains_dev(const uint64_t* elems, size_t width, size_t depth, uint64_t* out,
                           uint8_t* status, void* stream);
/* same with host pointers (status: OR of the item status bytes) */
int sp_pedersen_chains(const uint64_t* elems, size_t width, size_t depth, uint64_t* out, uint8_t* status);

/* ---- Merkle trees with node = pedersen_hash(left, right) ------------------------------------- */
/* (merkle_multi_update call sites services/perpetual/cairo/state/state.cairo:155-173)           */
/* Full rebuild over 2^height leaves.  levels_out (optional, may be NULL) receives every level
 * bottom-up, leaves first: (2^(height+1) - 1) felts.  status is a single byte (OR of item status). */
int sp_merkle_root(const uint64_t* leaves, unsigned height, uint64_t* root, uint64_t* levels_out,
                   uint8_t* status);
/* Device version: `levels` is a device buffer of (2^(height+1) - 1) felts whose first 2^height
 * entries hold the leaves; the upper levels are written behind them, the root last. */
int sp_merkle_build_dev(uint64_t* levels, unsigned height, uint8_t* status, void* stream);
/* n_trees independent trees (any count) of the same height in lockstep: one launch pair per level
 * for all of them.  Leaves of tree t at felts [t 2^height, (t+1) 2^height); the buffer is level-major
 * with n_trees (2^(height+1) - 1) felts and ends with the n_trees roots. */
int sp_merkle_forest_dev(uint64_t* levels, size_t n_trees, unsigned height, uint8_t* status,
                         void* stream);
/* Sparse multi-update: root of the tree of the given height (<= 64) that holds new_leaves[i] at
 * keys[i] (strictly increasing) and `empty_leaf` everywhere else - the induced-subtree walk of
 * starkware/python/merkle_tree.py:4-26. */
int sp_merkle_sparse_root(const uint64_t* keys, const uint64_t* leaves, size_t n, unsigned height,
                          const uint64_t* empty_leaf, uint64_t* root, uint8_t* status);
/* Persistent sparse tree: merkle_multi_update{hash_ptr=pedersen_ptr} on a tree that already holds
 * state (services/perpetual/cairo/state/state.cairo:155-173; untouched siblings come from the previous
 * state, the role of `merkle_facts`, main.cairo:61-64).  The handle keeps, per level, the nodes that
 * differ from the empty-subtree root (an open-addressing table in HBM); sp_tree_update writes leaves[i] at keys[i]
 * (strictly increasing, < 2^height, height <= 64) in one call - one gathered launch per level - and
 * returns the root before and after.  status != 0 (SP_HASH_*) leaves the tree unchanged. */
int sp_tree_create(unsigned height, const uint64_t* empty_leaf, int* tree);
/* The same on context `context` of sp_init_devices (0 = the primary, what sp_tree_create uses): the tree's table,
 * stream and work buffer live on that device; every sp_tree_* call on the handle runs there. */
int sp_tree_create_on(int context, unsigned height, const uint64_t* empty_leaf, int* tree);
int sp_tree_update(int tree, const uint64_t* keys, const uint64_t* leaves, size_t n, uint64_t* old_root,
                   uint64_t* new_root, uint8_t* status);
int sp_tree_get(int tree, const uint64_t* keys, size_t n, uint64_t* leaves);
int sp_tree_root(int tree, uint64_t* root);
/* Threading of the sp_tree_* calls: a tree has its own HIP stream, work buffer and mutex.  An operation holds
 * the tree's mutex from start to end and the library lock only while it enqueues; the device work of an update
 * (the level launches) runs without the library lock, so other trees and the stateless batches go on meanwhile. */
/* BASELINE.json configs[2] in one call (services/perpetual/cairo/order/limit_order.cairo:24-52, order.cairo:23-31,
 * :122-124): message-hash chains of n orders (words: depth x n felts, word-major; z_out receives the hashes) ->
 * verification of (z, r, s, key) through the key tables (verdicts as sp_ecdsa_verify_batch - a z of 2^251 or more is
 * SP_VERIFY_ASSERT_MSG and nothing is committed, constants.cairo:57 SIGNED_MESSAGE_BOUND; qy == NULL:
 * x-only keys) -> order id = bits [id_shift, id_shift + 64) of z (187 for the 251-bit message: the top 64 bits) ->
 * update of the orders tree `tree` with leaves[i] at order id i.  Verification and tree hashing overlap on the
 * device; the new nodes are committed only when every signature verified, otherwise *tree_status =
 * SP_TREE_NOT_COMMITTED and new_root = old_root.  Two orders with one id: SP_ERR_BAD_ARGUMENT.
 * Host side: the verification runs on a thread of the call's own (spawned as soon as the message hashes are back,
 * parked until the tree update - the critical path - has enqueued its levels; joined before the call returns);
 * 4096 orders on a tree that holds state: 1.75 - 1.85 ms median, p90 within 7 %, against 1.51 ms of dependent
 * hashing (3 chain links + 64 tree levels; DESIGN.md 4.2, profiles/r06_c3_timeline.txt).  A tree's slot table that
 * must grow is sized for four times the need and grows without a device-wide wait; STARKPERP_TIMELINE=1 prints
 * the host-side marks of every call to stderr. */
int sp_order_batch(const uint64_t* words, size_t depth, size_t n, const uint64_t* r, const uint64_t* s,
                   const uint64_t* qx, const uint64_t* qy, int tree, const uint64_t* leaves, unsigned id_shift,
                   uint64_t* z_out, uint8_t* verdicts, uint64_t* old_root, uint64_t* new_root, uint8_t* tree_status);
int sp_tree_destroy(int tree);

/* ---- Stark-curve ECDSA ------------------------------------------------------------------------ */
/* verify(msg_hash, r, s, public_key) signature.py:217-260.  qy == NULL: public keys are x-only
 * (both y candidates are tried, signature.py:229-238). */
int sp_ecdsa_verify_batch(const uint64_t* z, const uint64_t* r, const uint64_t* s,
                          const uint64_t* qx, const uint64_t* qy, uint8_t* result, size_t n);
int sp_ecdsa_verify_batch_dev(const uint64_t* z, const uint64_t* r, const uint64_t* s,
                              const uint64_t* qx, const uint64_t* qy, uint8_t* result, size_t n,
                              void* stream);
/* Key tables - the same verify() for public keys that are seen again (an exchange's accounts):
 * a registered key owns four 128-entry signed comb tables (32 KiB, on Q, 2^8 Q, 2^16 Q, 2^24 Q) in HBM that
 * replace the 252 doublings + 63 additions of the per-signature ladder by 7 doublings + 31 mixed additions.  Results are
 * identical to sp_ecdsa_verify_batch for every input (same pre-asserts, same False cases).
 *   sp_ecdsa_register_keys  host pointers; qy == NULL registers x-only keys; equal keys share a
 *                           slot; slots[i] receives the slot of key i; keys that are not on the curve own
 *                           no table - they share two sentinel slots (their verifications return False /
 *                           SP_VERIFY_ASSERT_CURVE as before) and are remembered on the host, so an
 *                           untrusted key stream cannot fill the cache;
 *                           SP_ERR_CACHE_FULL when no slot is free (ceiling: 2^17 keys = 4 GiB, or
 *                           STARKPERP_KEY_CACHE_SLOTS; the tables are allocated lazily, 128 MiB first, doubling),
 *                           in which case nothing is registered.
 *   sp_ecdsa_verify_keyed_dev  device pointers; slots[i] names the key of signature i.
 *   sp_ecdsa_verify_batch_keyed  host pointers: registers what is new, then verifies.
 * sp_ecdsa_verify_batch itself switches to the tables when at most 40 % of a batch's signatures bring
 * a key the library has never met (not registered, not seen in an earlier call, not repeated inside the
 * batch) - so a caller verifying one signature at a time reaches the tables on the second sighting of
 * a key.  That default (SP_VERIFY_POLICY_AUTO) makes sp_ecdsa_verify_batch remember keys between calls and
 * allocate the key cache on first use; sp_ecdsa_set_verify_policy chooses otherwise for the whole process:
 *   SP_VERIFY_POLICY_LADDER  sp_ecdsa_verify_batch never looks at, fills or allocates the key cache - the
 *                            stateless-after-init function of the reference; tables only through the
 *                            explicit calls above; selecting it also forgets the keys seen so far;
 *   SP_VERIFY_POLICY_KEYED   always through the tables (the ladder only for a batch with more keys than the cache holds).
 * When the policy path (AUTO / KEYED) finds the cache full it EVICTS: a new slot generation, as after
 * sp_ecdsa_key_cache_reset - handles obtained from sp_ecdsa_register_keys before that answer SP_VERIFY_STALE_SLOT.
 * sp_ecdsa_verify_batch_keyed runs on a host lane: the library lock is held to register new keys and to enqueue,
 * not while the caller waits for the device.
 * STARKPERP_VERIFY_KEYED=0 / 1 in the environment selects LADDER / KEYED as the initial policy.  The verdicts do
 * not depend on the policy. */
#define SP_VERIFY_POLICY_AUTO 0
#define SP_VERIFY_POLICY_LADDER 1
#define SP_VERIFY_POLICY_KEYED 2
int sp_ecdsa_set_verify_policy(int policy);
int sp_ecdsa_get_verify_policy(void);
int sp_ecdsa_register_keys(const uint64_t* qx, const uint64_t* qy, size_t n, uint32_t* slots);
int sp_ecdsa_verify_keyed_dev(const uint64_t* z, const uint64_t* r, const uint64_t* s,
                              const uint32_t* slots, uint8_t* result, size_t n, void* stream);
int sp_ecdsa_verify_batch_keyed(const uint64_t* z, const uint64_t* r, const uint64_t* s,
                                const uint64_t* qx, const uint64_t* qy, uint8_t* result, size_t n);
int sp_ecdsa_key_cache_info(size_t* capacity, size_t* used);
/* Empties the cache after waiting for the device and starts a new slot generation.  A slot handle is
 * (generation << 24) | index; handles from before the reset are answered with SP_VERIFY_STALE_SLOT by
 * sp_ecdsa_verify_keyed_dev (never with another key's verdict): callers register their keys again. */
int sp_ecdsa_key_cache_reset(void);
/* ---- signing: what the batch signer is for, and its threat model ------------------------------------------
 * The batch signer exists to MAKE order batches (BASELINE configs[2]: 4096 signed limit orders) - test vectors,
 * load generators, a market maker that signs its own quotes on its own device.  It is NOT a hardened signing
 * service for keys of third parties on a device shared with untrusted tenants:
 *   data-independent   the RFC 6979 nonce derivation (HMAC-SHA256: fixed schedule, no secret-dependent branch or
 *                      address), both inversions that involve secrets (k^-1 and the (z + r d) denominator: fixed
 *                      length Bernstein-Yang divsteps, 21 x 29 branch-free steps, identical for every lane,
 *                      fp29.hpp), all field / scalar multiplications (no secret-dependent branches);
 *   NOT independent    k*G and d*G are sums of EC_GEN table entries gathered from HBM at addresses that are windows
 *                      of the secret scalar (12 random 64-byte reads per nonce at the default 21-bit windows): an
 *                      observer of the memory system of the SAME device (another process time-slicing the GPU, a
 *                      co-tenant with cache / HBM-channel contention counters) can learn bits of k; the nonce
 *                      pipeline's compaction makes the NUMBER of rejected candidates of an item visible in timing
 *                      (it depends on k's HMAC chain, not on d); the exceptional-case branches of the attempt
 *                      (r = 0, out-of-range w: 2^-55 events) are data dependent;
 *   at rest            nonces and HMAC states of the compacted pipeline are scrubbed on the stream behind their
 *                      last reader, staged private keys of the host-pointer calls likewise; r, s and the public
 *                      key are public.  Buffers the CALLER owns (d in the _dev calls) are the caller's to scrub.
 * The reference is no different in kind: signature.py:137-173 signs with Python big integers, a recursive
 * double-and-add ec_mult whose branches follow the bits of k (math_utils.py:91-100) and sympy's variable-time
 * igcdex - it is not constant-time either.
 * STARKPERP_SIGN_MASKED=1 (read by sp_init) removes the address dependence: the signers and sp_public_key_batch then
 * walk a second EC_GEN table of 63 unsigned 4-bit windows (63 KiB), reading ALL 16 entries of every window - the
 * same addresses on every lane whatever the scalar is - and keeping one by a mask; 62 mixed additions for every k.
 * Checked on the ISA, not only at source level: tests/test_masked_walk_isa.py compiles the walk (csrc/masked_walk.hpp)
 * for gfx950 and asserts that every table entry is fetched by wave-uniform scalar loads, that EXEC is never written
 * and that the only branch is the uniform loop counter's; inside the signer kernels the mask passes an opaque-value
 * barrier so that the compiler cannot re-derive a branch from it.
 * Same keys and signatures bit for bit (tests/test_gpu_ecdsa.py::test_masked_signer_...); 2^16 signatures 0.86 ->
 * 1.27 ms (7.7 -> 5.2 x 10^7 /s), d * G four times slower (profiles/r05_masked_signer.txt).  What stays data
 * dependent under the mask: the number of rejected RFC 6979 candidates (timing of the nonce phase) and the 2^-55
 * exceptional branches.  A deployment that must sign third-party keys on a shared device should still keep its HSM
 * / constant-time CPU signer and use this library for hashing and verification (which handle public data only).
 *
 * One signing attempt per item with caller-supplied nonce k (host RFC 6979, signature.py:117-134):
 * the body of the loop at signature.py:146-173. */
int sp_ecdsa_sign_batch(const uint64_t* z, const uint64_t* d, const uint64_t* k, uint64_t* r,
                        uint64_t* s, uint8_t* status, size_t n);
/* The whole of sign(msg_hash, priv_key, seed) signature.py:137-173 on the device: the RFC 6979 nonce
 * (HMAC-SHA256, python-ecdsa conventions, signature.py:117-134), the attempt, and the reference's
 * retry with the next seed.  seeds: one uint64 per item (0 = no seed; NULL = none for every item).
 * status SP_SIGN_RETRY after 8 rejected nonces in a row (a 2^-55 event each) leaves the item to the
 * caller. */
int sp_ecdsa_sign_rfc6979_batch(const uint64_t* z, const uint64_t* d, const uint64_t* seeds, uint64_t* r,
                                uint64_t* s, uint8_t* status, size_t n);
/* The two signing calls on DEVICE pointers (signature.py:137-173 per item, as above): enqueued on `stream`,
 * nothing staged and nothing waited for - the batch signer of a device-resident pipeline (message hashes left
 * in HBM by sp_pedersen_chains_dev are signed where they lie).  status is required; r / s of an item whose
 * status is not SP_SIGN_OK are left as they were.  seeds may be NULL (no seed for any item).
 * Launches and scratch: sp_ecdsa_sign_batch_dev is ONE launch with no shared state.
 * sp_ecdsa_sign_rfc6979_batch_dev is one launch below 4096 items; from 4096 items on the nonce phase runs as
 * rounds over the compacted list of rejected candidates - 15 launches and one scrubbing memset per chunk of at
 * most 2^20 items (STARKPERP_SIGN_CHUNK) - on 232 bytes of per-stream scratch per item of a chunk (at most
 * 290 MiB per stream).  The first such call on a stream allocates the scratch (hipMalloc: a device-wide
 * synchronisation, not capturable into a hipGraph): warm the stream up first, or set STARKPERP_SIGN_COMPACT_MIN=0
 * to keep the one-launch signer at every size (same signatures, 2.0 instead of 2.9 x 10^8 /s at 2^20 items).  If
 * the scratch cannot be allocated the call falls back to the one-launch signer instead of failing. */
int sp_ecdsa_sign_batch_dev(const uint64_t* z, const uint64_t* d, const uint64_t* k, uint64_t* r, uint64_t* s,
                            uint8_t* status, size_t n, void* stream);
int sp_ecdsa_sign_rfc6979_batch_dev(const uint64_t* z, const uint64_t* d, const uint64_t* seeds, uint64_t* r,
                                    uint64_t* s, uint8_t* status, size_t n, void* stream);
/* private_key_to_ec_point_on_stark_curve signature.py:104-106: (qx, qy) = d * EC_GEN.
 * status: 0 ok, 2 when d is not in (0, EC_ORDER). */
int sp_public_key_batch(const uint64_t* d, uint64_t* qx, uint64_t* qy, uint8_t* status, size_t n);
/* The same on device pointers (qy, status may be NULL; outputs of a rejected item are left as they were). */
int sp_public_key_batch_dev(const uint64_t* d, uint64_t* qx, uint64_t* qy, uint8_t* status, size_t n, void* stream);

/* ---- prover-side transforms over GF(p) (build-defined: the reference has no prover) -------------- */
/* Field and generator: pedersen_params.json:20-21 (FIELD_PRIME, FIELD_GEN = 3).  All pointers are
 * device pointers to plain felts unless named *_host.                                            */
/* Radix-2 NTT of 2^log_n felts, natural order in and out; inverse != 0 divides by n. */
int sp_ntt_dev(const uint64_t* in, uint64_t* out, unsigned log_n, int inverse, void* stream);
/* Coset low-degree extension of ncols columns (column c at in + c * 2^log_n felts) from <w_n> to
 * shift * <w_{n * 2^log_blowup}>, natural order; out holds ncols * 2^(log_n + log_blowup) felts. */
int sp_lde_dev(const uint64_t* in, uint64_t* out, unsigned ncols, unsigned log_n, unsigned log_blowup,
               const uint64_t* shift_host, void* stream);
/* Execution trace of the Pedersen-step AIR: 512 rows per hash, columns s, px, py, lambda
 * (cols = 4 columns of 512 * n_hashes felts).  Step relation: signature.py:305-317 with the chord
 * rule of math_utils.py:59-68. */
int sp_pedersen_trace_dev(const uint64_t* x, const uint64_t* y, size_t n_hashes, uint64_t* cols,
                          void* stream);
/* Composition column sum_k alpha_k C_k(x) / (x^n - 1) on the blowup-4 coset.  trace_lde: 4 columns of
 * 4 * 2^log_n felts; periodic_lde: 6 tables of 2048 felts; alphas_host: 11 felts. */
int sp_air_eval_dev(const uint64_t* trace_lde, const uint64_t* periodic_lde, unsigned log_n,
                    const uint64_t* alphas_host, const uint64_t* shift_host, uint64_t* out, void* stream);
/* Row shard of the same column for one rank of a multi-GPU job (SURVEY 8(e) "AIR eval": LDE-row shards
 * with a halo of one trace row): n_points LDE points from global index row0 (a multiple of 4); the four
 * columns are col_stride felts apart and hold n_points + 4 rows, the last four received from the owner of
 * the following rows.  log_n is the global trace length. */
int sp_air_eval_shard_dev(const uint64_t* trace_lde, size_t col_stride, size_t n_points, size_t row0,
                          const uint64_t* periodic_lde, unsigned log_n, const uint64_t* alphas_host,
                          const uint64_t* shift_host, uint64_t* out, void* stream);
/* EC-ladder AIR (one mimic_ec_mult_air instance = 256 rows, signature.py:176-190): witness columns
 * m, px, py, qx, qy, la, ld for n_ladders scalar multiplications m * (qx, qy) + SHIFT_POINT
 * (cols = 7 columns of 256 * n_ladders felts), and its composition column (12 constraints,
 * periodic_lde: 3 tables of 1024 felts). */
int sp_ec_ladder_trace_dev(const uint64_t* m, const uint64_t* qx, const uint64_t* qy, size_t n_ladders,
                           uint64_t* cols, void* stream);
int sp_air_eval_ec_ladder_dev(const uint64_t* trace_lde, const uint64_t* periodic_lde, unsigned log_n,
                              const uint64_t* alphas_host, const uint64_t* shift_host, uint64_t* out,
                              void* stream);
/* Range-check AIR (what the Cairo range-check builtin asserts for every amount, position id, nonce and
 * expiration of the exchange messages, e.g. signature_message_hashes.cairo:60-75 via assert_nn_le /
 * the 2^64, 2^32 bounds of perpetual_messages.py:226-236): 0 <= value < 2^128 by bit decomposition,
 * 128 rows of one column per value (col = 128 * n_values felts, v_i = value >> i), and its composition
 * column (2 constraints, periodic_lde: 2 tables of 512 felts).  A value >= 2^128 gives a trace whose
 * composition is not a polynomial: the proof fails at the verifier, nothing is rejected here. */
int sp_range_check_trace_dev(const uint64_t* values, size_t n_values, uint64_t* col, void* stream);
int sp_air_eval_range_check_dev(const uint64_t* trace_lde, const uint64_t* periodic_lde, unsigned log_n,
                                const uint64_t* alphas_host, const uint64_t* shift_host, uint64_t* out,
                                void* stream);
/* The range-check BUILTIN as an AIR segment (oracle/stark_ref.py "rc16"; what the Cairo program's
 * `range_check` builtin - services/perpetual/cairo/main.cairo:1, used at order/order.cairo:53-56 - asserts):
 * a value < 2^128 is eight 16-bit limbs, 8 rows of the columns a (limb), acc (running value) and s (the limb
 * column SORTED: neighbours differ by 0 or 1, first = rc_min, last = rc_max); after these are committed a
 * challenge z is drawn and the second-phase column p_i = prod_{j <= i} (z - a_j) / (z - s_j) proves that s is a
 * permutation of a.  sp_rc16_trace_dev writes a and acc (cols = 2 columns of 8 * n_values felts; the caller
 * sorts a into s); sp_rc16_product_dev writes p (n felts); sp_air_eval_rc16_dev the rc16 part of the
 * composition column (cols: a, acc, s of 4 * 2^log_n felts each; periodic_lde: 2 tables of 32 felts;
 * alphas_host: 8 felts) - transition, all-but-last-row and boundary constraints.  sp_felt_add_dev sums the
 * composition parts of the segments of a combined trace. */
int sp_rc16_trace_dev(const uint64_t* values, size_t n_values, uint64_t* cols, void* stream);
int sp_rc16_product_dev(const uint64_t* a, const uint64_t* s, size_t n, const uint64_t* z_host, uint64_t* p,
                        void* stream);
int sp_air_eval_rc16_dev(const uint64_t* cols, const uint64_t* p, const uint64_t* periodic_lde, unsigned log_n,
                         const uint64_t* alphas_host, const uint64_t* shift_host, const uint64_t* z_host,
                         uint64_t rc_min, uint64_t rc_max, uint64_t* out, void* stream);
int sp_felt_add_dev(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
/* ECDSA-verification AIR (what verify() mimics, signature.py:217-260): three linked EC ladders per
 * signature - z G from MINUS_SHIFT_POINT, r Q and w B from SHIFT_POINT with B = zG + rQ (:252-254) - and
 * x(wB - SHIFT_POINT) == r (:255); 1024 rows of ten columns m, px, py, qx, qy, la, ld, cx, cy, cr per
 * signature (cols = 10 columns of 1024 * n_sigs felts).  w = s^-1 mod N is an input, as in the reference
 * (:220: verify computes it before mimicking the AIR).  Composition: 26 constraints, periodic_lde = 12
 * tables of 4096 felts. */
int sp_ecdsa_trace_dev(const uint64_t* z, const uint64_t* r, const uint64_t* w, const uint64_t* qx,
                       const uint64_t* qy, size_t n_sigs, uint64_t* cols, void* stream);
int sp_air_eval_ecdsa_dev(const uint64_t* trace_lde, const uint64_t* periodic_lde, unsigned log_n,
                          const uint64_t* alphas_host, const uint64_t* shift_host, uint64_t* out, void* stream);
/* One FRI fold of f on shift * <w_M> (M = 2^log_m) to g on shift^2 * <w_{M/2}>:
 * g(x^2) = (f(x) + f(-x)) / 2 + beta (f(x) - f(-x)) / (2 x). */
int sp_fri_fold_dev(const uint64_t* in, uint64_t* out, unsigned log_m, const uint64_t* beta_host,
                    const uint64_t* shift_host, void* stream);
/* Row shard of one fold: out[i] = fold(fa[i], fb[i]) for the global positions i0 .. i0 + count of a layer of
 * 2^log_m points; fa = f[i0 ..], fb = f[i0 + 2^(log_m - 1) ..] (the second array comes from the rank that
 * owns the upper half of the layer). */
int sp_fri_fold_shard_dev(const uint64_t* fa, const uint64_t* fb, uint64_t* out, unsigned log_m, size_t i0,
                          size_t count, const uint64_t* beta_host, const uint64_t* shift_host, void* stream);
/* Block-cyclic row shards (one job over the GPUs of a node, starkperp/sharded_prover.py; SURVEY 8(e) rows "AIR
 * eval" and "FRI"): the LDE rows / layer positions are cut into blocks of 2^log_block consecutive indices and
 * block b belongs to rank b mod world, local block t being global block t * world + rank.  The next-row reads
 * of the AIR stay inside a block plus a halo of one trace row, and both members (i, i + M/2) of every fold pair
 * live on the same rank while M/2 >= world * 2^log_block - no data moves between folds.
 *   sp_air_eval_blocks_dev   trace_lde: 4 columns col_stride felts apart, n_blocks blocks stored 2^log_block + 4
 *                            rows apart (block + halo); out: n_blocks * 2^log_block composition values.
 *   sp_fri_fold_blocks_dev   fa / fb: the rank's `count` positions of the lower / upper half of the layer. */
int sp_air_eval_blocks_dev(const uint64_t* trace_lde, size_t col_stride, size_t n_blocks, unsigned log_block,
                           unsigned world, unsigned rank, const uint64_t* periodic_lde, unsigned log_n,
                           const uint64_t* alphas_host, const uint64_t* shift_host, uint64_t* out, void* stream);
int sp_fri_fold_blocks_dev(const uint64_t* fa, const uint64_t* fb, uint64_t* out, unsigned log_m, size_t count,
                           unsigned log_block, unsigned world, unsigned rank, const uint64_t* beta_host,
                           const uint64_t* shift_host, void* stream);
/* The two halves of sp_lde_dev (blowup 1) as separate calls, so that the coset transforms of one column share a
 * single interpolation: evaluations on <w_n> -> coefficients in BIT-REVERSED order -> evaluations on
 * shift * <w_n> (natural order).  ncols columns 2^log_n felts apart. */
int sp_interpolate_dev(const uint64_t* in, uint64_t* coef, unsigned ncols, unsigned log_n, void* stream);
int sp_coset_eval_dev(const uint64_t* coef, uint64_t* out, unsigned ncols, unsigned log_n,
                      const uint64_t* shift_host, void* stream);
/* Commit (build-defined): Pedersen-Merkle tree over the rows of a column-major table of
 * n_rows (a power of two) x n_cols felts; leaf = left-fold chain of the row's felts (one column: the
 * felt itself).  levels receives 2 n_rows - 1 felts, leaves first, root last.  Passing a status
 * pointer makes the call wait for the stream. */
int sp_commit_rows_dev(const uint64_t* cols, size_t n_rows, size_t n_cols, uint64_t* levels, uint8_t* status,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STARKPERP_H */

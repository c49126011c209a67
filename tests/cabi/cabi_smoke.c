/* Plain-C consumer of include/starkperp.h: shows the drop-in boundary needs nothing but a C
 * compiler and the shared library (no Python, no torch).  Built and run by tests/test_gpu_cabi.py.
 * Checks pedersen_hash(1, 2) and the two hash_test vectors of the reference's
 * signature_test_data.json:190-201, a 2-leaf tree, a chain, sign -> verify (ladder and key tables),
 * deterministic signing on the device, a persistent tree, two contexts in one process. */
#include <stdio.h>
#include <string.h>
#include "../../include/starkperp.h"

static int hexval(char c) { return c <= '9' ? c - '0' : (c | 32) - 'a' + 10; }
static void felt_from_hex(const char* hex, uint64_t out[4]) {
  memset(out, 0, 32);
  size_t n = strlen(hex);
  for (size_t i = 0; i < n; ++i) {
    int v = hexval(hex[n - 1 - i]);
    out[i / 16] |= (uint64_t)v << (4 * (i % 16));
  }
}
static int felt_eq_hex(const uint64_t a[4], const char* hex) {
  uint64_t b[4];
  felt_from_hex(hex, b);
  return memcmp(a, b, 32) == 0;
}

int main(void) {
  if (sp_init(0, 16) != SP_OK) { fprintf(stderr, "sp_init: %s\n", sp_last_error()); return 2; }
  uint64_t x[2][4], y[2][4], out[2][4];
  uint8_t st[2];
  felt_from_hex("3d937c035c878245caf64531a5756109c53068da139362728feb561405371cb", x[0]);
  felt_from_hex("208a0a10250e382e1e4bbe2880906c2791bf6275695e02fbbc6aeff9cd8b31a", y[0]);
  felt_from_hex("58f580910a6ca59b28927c08fe6c43e2e303ca384badc365795fc645d479d45", x[1]);
  felt_from_hex("78734f65a067be9bdb39de18434d71e79f7b6466a4b66bbd979ab9e7515fe0b", y[1]);
  if (sp_pedersen_batch(&x[0][0], &y[0][0], &out[0][0], st, 2) != SP_OK) return 3;
  if (st[0] || st[1]) return 4;
  if (!felt_eq_hex(out[0], "30e480bed5fe53fa909cc0f8c4d99b8f9f2c016be4c41e13a4848797979c662")) return 5;
  if (!felt_eq_hex(out[1], "68cc0b76cddd1dd4ed2301ada9b7c872b23875d5ff837b3a87993e0d9996b87")) return 6;

  /* 2-leaf tree root == H(leaf0, leaf1); chain of the same two words gives the same value */
  uint64_t leaves[2][4], root[4], chain_out[4];
  uint8_t s1 = 0;
  memcpy(leaves[0], x[0], 32); memcpy(leaves[1], y[0], 32);
  if (sp_merkle_root(&leaves[0][0], 1, root, NULL, &s1) != SP_OK || s1) return 7;
  if (memcmp(root, out[0], 32) != 0) return 8;
  if (sp_pedersen_chain(&leaves[0][0], 2, chain_out, &s1) != SP_OK || s1) return 9;
  if (memcmp(chain_out, out[0], 32) != 0) return 10;

  /* party_a_order: public key, one signing attempt with the RFC 6979 nonce, verification */
  uint64_t z[4], d[4], k[4], r[4], s[4], qx[4], qy[4];
  uint8_t code = 0;
  felt_from_hex("397e76d1667c4454bfb83514e120583af836f8e32a516765497823eabe16a3f", z);
  felt_from_hex("3c1e9550e66958296d11b60f8e8e7a7ad990d07fa65d5f7652c4a6c87d4e3cc", d);
  if (sp_public_key_batch(d, qx, qy, &code, 1) != SP_OK || code) return 11;
  if (!felt_eq_hex(qx, "77a3b314db07c45076d11f62b6f9e748a39790441823307743cf00d6597ea43")) return 12;
  felt_from_hex("173fd03d8b008ee7432977ac27d1e9d1a1f6c98b1a2f05fa84a21c84c44e882", r);
  felt_from_hex("4b6d75385aed025aa222f28a0adc6d58db78ff17e51c3f59e259b131cd5a1cc", s);
  if (sp_ecdsa_verify_batch(z, r, s, qx, NULL, &code, 1) != SP_OK || code != SP_VERIFY_TRUE) return 13;
  if (sp_ecdsa_verify_batch(z, r, s, qx, qy, &code, 1) != SP_OK || code != SP_VERIFY_TRUE) return 14;
  z[0] ^= 1;
  if (sp_ecdsa_verify_batch(z, r, s, qx, NULL, &code, 1) != SP_OK || code != SP_VERIFY_FALSE) return 15;
  z[0] ^= 1;
  /* any valid nonce gives a signature that verifies */
  felt_from_hex("1234567890abcdef1234567890abcdef1234567890abcdef1234567890abcd", k);
  if (sp_ecdsa_sign_batch(z, d, k, r, s, &code, 1) != SP_OK || code != SP_SIGN_OK) return 16;
  if (sp_ecdsa_verify_batch(z, r, s, qx, NULL, &code, 1) != SP_OK || code != SP_VERIFY_TRUE) return 17;

  /* the whole of sign() on the device reproduces the reference's signature of this order
   * (signature_test_data.json:17-20), and the key tables give the same verdicts as the ladder */
  if (sp_ecdsa_sign_rfc6979_batch(z, d, NULL, r, s, &code, 1) != SP_OK || code != SP_SIGN_OK) return 18;
  if (!felt_eq_hex(r, "173fd03d8b008ee7432977ac27d1e9d1a1f6c98b1a2f05fa84a21c84c44e882")) return 19;
  if (!felt_eq_hex(s, "4b6d75385aed025aa222f28a0adc6d58db78ff17e51c3f59e259b131cd5a1cc")) return 20;
  if (sp_ecdsa_verify_batch_keyed(z, r, s, qx, NULL, &code, 1) != SP_OK || code != SP_VERIFY_TRUE) return 21;
  z[0] ^= 1;
  if (sp_ecdsa_verify_batch_keyed(z, r, s, qx, NULL, &code, 1) != SP_OK || code != SP_VERIFY_FALSE) return 22;
  z[0] ^= 1;

  /* persistent tree: two updates of a height-1 tree; the second one sees the first one's leaf */
  int tree = 0;
  uint64_t zero[4] = {0, 0, 0, 0}, key0 = 0, key1 = 1, r_old[4], r_new[4], r_new2[4];
  if (sp_tree_create(1, zero, &tree) != SP_OK) return 23;
  if (sp_tree_update(tree, &key0, x[0], 1, r_old, r_new, &s1) != SP_OK || s1) return 24;
  if (sp_tree_update(tree, &key1, y[0], 1, r_old, r_new2, &s1) != SP_OK || s1) return 25;
  if (memcmp(r_old, r_new, 32) != 0 || memcmp(r_new2, out[0], 32) != 0) return 26;  /* H(x0, y0) */
  if (sp_tree_destroy(tree) != SP_OK) return 27;
  sp_shutdown();

  /* a second life with another window plan: same answers, no state left over */
  if (sp_pedersen_batch(&x[0][0], &y[0][0], &out[0][0], st, 2) != SP_ERR_NOT_INITIALISED) return 28;
  if (sp_init(0, 11) != SP_OK || sp_window_bits() != 11) return 29;
  if (sp_pedersen_batch(&x[0][0], &y[0][0], &out[0][0], st, 2) != SP_OK || st[0] || st[1]) return 30;
  if (!felt_eq_hex(out[1], "68cc0b76cddd1dd4ed2301ada9b7c872b23875d5ff837b3a87993e0d9996b87")) return 31;
  if (sp_tree_root(tree, r_old) == SP_OK) return 32; /* handles do not survive a shutdown */
  size_t cap = 0, used = 99;
  if (sp_ecdsa_verify_batch_keyed(z, r, s, qx, NULL, &code, 1) != SP_OK || code != SP_VERIFY_TRUE) return 33;
  if (sp_ecdsa_key_cache_info(&cap, &used) != SP_OK || used != 1) return 34;
  sp_shutdown();

  /* a third life with two contexts (the box has one GPU: it is listed twice) */
  {
    const int ids[2] = {0, 0};
    int dev = -1;
    uint64_t calls = 99;
    if (sp_device_count() != 0) return 35;
    if (sp_init_devices(2, ids, 10) != SP_OK || sp_device_count() != 2) return 36;
    if (sp_init_devices(1, ids, 10) == SP_OK) return 37; /* another layout needs sp_shutdown first */
    if (sp_pedersen_batch(&x[0][0], &y[0][0], &out[0][0], st, 2) != SP_OK || st[0] || st[1]) return 38;
    if (sp_pedersen_batch(&x[1][0], &y[1][0], &out[1][0], st, 1) != SP_OK || st[0]) return 39;
    if (!felt_eq_hex(out[0], "30e480bed5fe53fa909cc0f8c4d99b8f9f2c016be4c41e13a4848797979c662")) return 40;
    if (!felt_eq_hex(out[1], "68cc0b76cddd1dd4ed2301ada9b7c872b23875d5ff837b3a87993e0d9996b87")) return 41;
    if (sp_context_info(0, &dev, &calls) != SP_OK || dev != 0 || calls != 1) return 42; /* one call each: */
    if (sp_context_info(1, &dev, &calls) != SP_OK || dev != 0 || calls != 1) return 43; /* lanes alternate */
    if (sp_context_info(2, &dev, &calls) == SP_OK) return 44;
    sp_shutdown();
  }
  printf("cabi_smoke ok\n");
  return 0;
}

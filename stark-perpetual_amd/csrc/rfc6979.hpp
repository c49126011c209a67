// Deterministic ECDSA nonces on the device: RFC 6979 section 3.2 with HMAC-SHA256 and the
// conventions of python-ecdsa 0.17's `ecdsa.rfc6979.generate_k`, the third-party routine the
// reference calls at signature.py:128-134 (host twin: starkperp/rfc6979.py, pinned by the
// reference's own signatures in tests/golden).  For the Stark curve (qlen = 252, 32-byte octets):
//   h1           = msg_hash                 (the one-nibble pad of signature.py:119-121 shifts the
//                                            message left by 4 bits and bits2int shifts it back)
//   key material = int2octets(d) || int2octets(h1) || extra_entropy
//   extra_entropy = minimal big-endian bytes of the retry seed (none for seed None or 0)
//   candidate    = int(V) >> 4, accepted when 1 <= candidate < N
#pragma once
#include "context.hpp"

namespace sp {

struct sha256_state {
  uint32_t h[8];
};

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, (uint32_t)n); }

__device__ __noinline__ void sha256_compress(sha256_state& s, const uint32_t* block /* 16 big-endian words */) {
  constexpr uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
      0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
      0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
      0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
      0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
      0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
      0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = block[i];
  uint32_t a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3], e = s.h[4], f = s.h[5], g = s.h[6], h = s.h[7];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    if (i >= 16) {
      const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      const uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
      const uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
      w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
    }
    const uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
    const uint32_t ch = (e & f) ^ (~e & g);
    const uint32_t t1 = h + S1 + ch + K[i] + w[i & 15];
    const uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
    const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    const uint32_t t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s.h[0] += a; s.h[1] += b; s.h[2] += c; s.h[3] += d; s.h[4] += e; s.h[5] += f; s.h[6] += g; s.h[7] += h;
}

__device__ __forceinline__ sha256_state sha256_init() {
  return sha256_state{{0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19}};
}

// Streaming SHA-256 over big-endian bytes, byte granularity (messages here are < 200 bytes).
struct sha256_stream {
  sha256_state st;
  uint32_t block[16];
  uint32_t len;  // bytes absorbed
};
__device__ __forceinline__ void sha256_begin(sha256_stream& s, const sha256_state& start, uint32_t already) {
  s.st = start;
  s.len = already;
#pragma unroll
  for (int i = 0; i < 16; ++i) s.block[i] = 0;
}
__device__ __forceinline__ void sha256_put(sha256_stream& s, uint32_t byte) {
  const uint32_t pos = s.len & 63u;
  // dynamic word index kept cheap: select chain over 16 words
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if ((int)(pos >> 2) == i) s.block[i] |= byte << (24 - 8 * (pos & 3u));
  }
  ++s.len;
  if ((s.len & 63u) == 0) {
    sha256_compress(s.st, s.block);
#pragma unroll
    for (int i = 0; i < 16; ++i) s.block[i] = 0;
  }
}
__device__ __forceinline__ void sha256_put_words(sha256_stream& s, const uint32_t* be_words, int n_words) {
  for (int i = 0; i < n_words; ++i) {
    sha256_put(s, be_words[i] >> 24);
    sha256_put(s, (be_words[i] >> 16) & 0xff);
    sha256_put(s, (be_words[i] >> 8) & 0xff);
    sha256_put(s, be_words[i] & 0xff);
  }
}
__device__ __forceinline__ void sha256_end(sha256_stream& s, uint32_t* digest /* 8 big-endian words */) {
  const uint32_t total_bits = s.len * 8u;
  sha256_put(s, 0x80);
  while ((s.len & 63u) != 56u) sha256_put(s, 0);
  sha256_put(s, 0); sha256_put(s, 0); sha256_put(s, 0); sha256_put(s, 0);
  sha256_put(s, total_bits >> 24); sha256_put(s, (total_bits >> 16) & 0xff);
  sha256_put(s, (total_bits >> 8) & 0xff); sha256_put(s, total_bits & 0xff);
#pragma unroll
  for (int i = 0; i < 8; ++i) digest[i] = s.st.h[i];
}

// HMAC-SHA256 with a 32-byte key: midstates after the ipad / opad blocks.
struct hmac_key {
  sha256_state inner, outer;
};
__device__ __forceinline__ hmac_key hmac_prepare(const uint32_t* key /* 8 BE words */) {
  uint32_t blk[16];
  hmac_key k;
#pragma unroll
  for (int i = 0; i < 16; ++i) blk[i] = (i < 8 ? key[i] : 0u) ^ 0x36363636u;
  k.inner = sha256_init();
  sha256_compress(k.inner, blk);
#pragma unroll
  for (int i = 0; i < 16; ++i) blk[i] = (i < 8 ? key[i] : 0u) ^ 0x5c5c5c5cu;
  k.outer = sha256_init();
  sha256_compress(k.outer, blk);
  return k;
}
// One-block tail: `words` (8 big-endian words) plus, optionally, one more byte, hashed on top of a
// midstate that has absorbed 64 bytes already.
__device__ __forceinline__ void sha256_tail_block(const sha256_state& mid, const uint32_t* words, int extra_byte,
                                                  uint32_t* digest) {
  uint32_t blk[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) blk[i] = words[i];
  blk[8] = extra_byte < 0 ? 0x80000000u : (((uint32_t)extra_byte << 24) | 0x00800000u);
#pragma unroll
  for (int i = 9; i < 15; ++i) blk[i] = 0;
  blk[15] = extra_byte < 0 ? (64u + 32u) * 8u : (64u + 33u) * 8u;
  sha256_state st = mid;
  sha256_compress(st, blk);
#pragma unroll
  for (int i = 0; i < 8; ++i) digest[i] = st.h[i];
}

// out = HMAC(key, V || [sep || d || h1 || entropy])   (sep < 0: V only)
__device__ __forceinline__ void hmac_v(const hmac_key& key, const uint32_t* v, int sep, const uint32_t* d_be,
                                       const uint32_t* h1_be, uint64_t seed, bool with_material, uint32_t* out) {
  uint32_t inner[8];
  if (sep >= 0 && with_material) {  // the two long messages: byte stream
    sha256_stream s;
    sha256_begin(s, key.inner, 64);
    sha256_put_words(s, v, 8);
    sha256_put(s, (uint32_t)sep);
    sha256_put_words(s, d_be, 8);
    sha256_put_words(s, h1_be, 8);
    int nbytes = 0;
    for (uint64_t t = seed; t != 0; t >>= 8) ++nbytes;
    for (int i = nbytes - 1; i >= 0; --i) sha256_put(s, (uint32_t)((seed >> (8 * i)) & 0xff));
    sha256_end(s, inner);
  } else {  // V, or V || sep: a single block with a fixed layout
    sha256_tail_block(key.inner, v, sep, inner);
  }
  sha256_tail_block(key.outer, inner, -1, out);
}

__device__ __forceinline__ void u256_to_be_words(const u256& a, uint32_t* be) {
#pragma unroll
  for (int i = 0; i < 8; ++i) be[i] = a.w[7 - i];
}

// k = generate_k(N, d, sha256, message(z), extra_entropy(seed)); returns the first valid candidate.
__device__ __noinline__ u256 rfc6979_nonce(const u256& z, const u256& d, uint64_t seed) {
  uint32_t d_be[8], h1_be[8], v[8], kk[8], t[8];
  u256_to_be_words(d, d_be);
  u256_to_be_words(z, h1_be);
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = 0x01010101u; kk[i] = 0; }
  hmac_key key = hmac_prepare(kk);
  hmac_v(key, v, 0x00, d_be, h1_be, seed, true, kk);
  key = hmac_prepare(kk);
  hmac_v(key, v, -1, d_be, h1_be, seed, false, t);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = t[i];
  hmac_v(key, v, 0x01, d_be, h1_be, seed, true, kk);
  key = hmac_prepare(kk);
  hmac_v(key, v, -1, d_be, h1_be, seed, false, t);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = t[i];
  for (int guard = 0; guard < 64; ++guard) {
    hmac_v(key, v, -1, d_be, h1_be, seed, false, t);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = t[i];
    u256 cand;  // int(V) >> 4
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t lo = v[7 - i], hi = i < 7 ? v[6 - i] : 0u;
      cand.w[i] = (lo >> 4) | (hi << 28);
    }
    if (!u256_is_zero(cand) && u256_lt(cand, U256_N)) return cand;
    hmac_v(key, v, 0x00, d_be, h1_be, seed, false, kk);
    key = hmac_prepare(kk);
    hmac_v(key, v, -1, d_be, h1_be, seed, false, t);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = t[i];
  }
  u256 zero;
#pragma unroll
  for (int i = 0; i < 8; ++i) zero.w[i] = 0;
  return zero;  // unreachable in practice; the caller reports SP_SIGN_RETRY
}

}  // namespace sp

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

struct sha256_block {
  uint32_t w[16];  // big-endian words
};

// One compression; always inlined, and inlined exactly ONCE per kernel: rfc6979_nonce below runs its 16
// compressions as the steps of one wave-uniform loop around a single copy of these 1.8 k instructions.
// (The first version - a non-inlined compress behind a byte-stream SHA-256 whose message bytes and midstates lived
// in scratch memory: 8.4 k instructions, 1391 scratch accesses and 66 call sites - cost 0.69 ms of a 0.92 ms
// signature launch at 2^16 items; this form has no calls and no scratch: 0.62 of 0.85 ms.  What remains is the
// rejection loop, see rfc6979_nonce.)
__device__ __forceinline__ sha256_state sha256_compress(sha256_state s, sha256_block blk) {
  constexpr uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
      0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
      0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
      0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
      0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
      0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
      0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t* w = blk.w;
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
  return s;
}

__device__ __forceinline__ sha256_state sha256_init() {
  return sha256_state{{0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19}};
}

// k = generate_k(N, d, sha256, message(z), extra_entropy(seed)); returns the first valid candidate.
//
// HMAC-SHA256 with 32-byte keys throughout: a key is the pair of midstates after its ipad / opad blocks
// (kin, kout); HMAC(key, m) = compress*(kout, compress*(kin, m)).  The messages have fixed layouts:
//   V                          one block behind the ipad block: V, 0x80, zeros, bit length (64 + 32) * 8
//   V || sep                   the same with sep in front of the 0x80 and length (64 + 33) * 8
//   V || sep || d || h1 || e   97 + nb bytes (e = the nb = 0..8 minimal big-endian bytes of the seed): always TWO
//                              blocks; the stream sep || d || h1 || e || 0x80 is the word sequence X = d, h1, E
//                              (entropy left-aligned, the 0x80 marker behind it) shifted right by one byte
// so every block is assembled from whole words, and the 16 compressions of a nonce (+ 8 per rejected candidate)
// are the steps of ONE loop.  A candidate int(V) >> 4 has 252 bits and N ~ 2^251: HALF of the candidates are
// rejected, so a wave of 64 items runs 1 + ~7 rounds of the retry (the longest of 64 geometric runs): ~72
// compressions where an item needs 24 on average - measured, quick_sign.py: the nonce is 0.46 ms of a 0.59 ms
// signature launch whatever the batch size up to one wave per SIMD.  The wait is inherent to lockstep lanes
// (the retry chain K = HMAC(K, V 00), V = HMAC(K, V), V = HMAC(K, V) is serial per item and must be followed
// exactly to reproduce the reference's k); callers that hold their own nonces use sp_ecdsa_sign_batch_dev.
//    0  1  2 | 3  4 | 5  6 | 7  8  9 | 10 11 | 12 13 | 14 15 | (16 17 | 18 19 | 20 21 -> 14)
//    K = HMAC(0, V 00 d h1 e)   V = HMAC(K, V)   K = HMAC(K, V 01 d h1 e)   V = HMAC(K, V)   V = HMAC(K, V) -> k
// The all-zero key of step 0 / 2 is a constant: compress(IV, 0x36 x 64), compress(IV, 0x5c x 64).
// Round 4: the nonce generator in two pieces so that a batch signer can COMPACT between candidates
// (ecdsa.hip sign_nonce_first_kernel / sign_nonce_retry_kernel): the fixed word layouts of the key material
// (rfc6979_prepare) and a resumable run of the step loop (rfc6979_run).
struct rfc_input {
  uint32_t T[20];      // sep || d || h1 || entropy || 0x80 shifted right by one byte; the separator is OR-ed in per use
  uint32_t long_bits;  // bit length of the long messages
};
struct rfc_state {
  sha256_state kin, kout, v;  // the HMAC key as its two pad midstates, and V
};

__device__ __forceinline__ void rfc6979_prepare(const u256& z, const u256& d, uint64_t seed, rfc_input& in) {
  uint32_t X[19];  // d, h1 big-endian, entropy + marker
#pragma unroll
  for (int i = 0; i < 8; ++i) { X[i] = d.w[7 - i]; X[8 + i] = z.w[7 - i]; }
  uint32_t nb = 0;
  for (uint64_t t = seed; t != 0; t >>= 8) ++nb;
  {
    const uint64_t left = nb ? seed << (8u * (8u - nb)) : 0ull;  // entropy bytes left-aligned in 64 bits
    X[16] = (uint32_t)(left >> 32); X[17] = (uint32_t)left; X[18] = 0u;
    const uint32_t mark = 0x80u << (24u - 8u * (nb & 3u));
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if ((int)(nb >> 2) == i) X[16 + i] |= mark;
  }
  in.T[0] = X[0] >> 8;
#pragma unroll
  for (int i = 1; i < 19; ++i) in.T[i] = (X[i - 1] << 24) | (X[i] >> 8);
  in.T[19] = X[18] << 24;
  in.long_bits = (64u + 97u + nb) * 8u;
}

// Runs the step loop until a candidate is accepted (returns true, the candidate in `cand`) or `max_rejected`
// candidates were rejected (returns false; `st` is then the state right after the last rejected candidate).
// START: from step 0 with `in` (st is initialised here); otherwise from step 16 with the state a previous call
// left behind - `in` is not read (steps 0, 1, 7, 8 are the only ones that touch the key material).
template <bool START>
__device__ __forceinline__ bool rfc6979_run(const rfc_input& in, rfc_state& st, int max_rejected, u256& cand,
                                            int* rejected_out = nullptr) {
  sha256_state kin, kout, v, kk = sha256_init(), acc = sha256_init();  // V, the new key K, the running digest
  if (START) {
    kin = sha256_state{{0xf454dead, 0x9725214f, 0x90daf2a0, 0xdf1228ea, 0x64e5750f, 0xa3924181, 0x824a932b, 0xf8e04e32}};
    kout = sha256_state{{0xd385480f, 0x7abb6477, 0x37c9c538, 0x5dd82467, 0x8e043a72, 0x753434b0, 0xdeb82818, 0x361d45a6}};
#pragma unroll
    for (int i = 0; i < 8; ++i) v.h[i] = 0x01010101u;
  } else {
    kin = st.kin; kout = st.kout; v = st.v;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) cand.w[i] = 0;
  int step = START ? 0 : 16, rejected = 0;
  bool accepted = false;
  for (;;) {
    // ---- the block and the state it is compressed onto ----
    sha256_block blk;
    sha256_state cs;
    const bool long_a = START && (step == 0 || step == 7), long_b = START && (step == 1 || step == 8);
    const bool pad_in = step == 3 || step == 10 || step == 18, pad_out = step == 4 || step == 11 || step == 19;
    const bool outer = step == 2 || step == 6 || step == 9 || step == 13 || step == 15 || step == 17 || step == 21;
    if (long_a) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { blk.w[i] = v.h[i]; blk.w[8 + i] = in.T[i]; }
      blk.w[8] |= step == 7 ? 0x01000000u : 0u;
      cs = kin;
    } else if (long_b) {
#pragma unroll
      for (int i = 0; i < 12; ++i) blk.w[i] = in.T[8 + i];
      blk.w[12] = blk.w[13] = blk.w[14] = 0;
      blk.w[15] = in.long_bits;
      cs = acc;
    } else if (pad_in || pad_out) {
      const uint32_t pad = pad_in ? 0x36363636u : 0x5c5c5c5cu;
#pragma unroll
      for (int i = 0; i < 16; ++i) blk.w[i] = (i < 8 ? kk.h[i] : 0u) ^ pad;
      cs = sha256_init();
    } else {  // a 32-byte message behind a 64-byte pad block: V (or V || 00 at step 16), or an inner digest
#pragma unroll
      for (int i = 0; i < 8; ++i) blk.w[i] = outer ? acc.h[i] : v.h[i];
      blk.w[8] = step == 16 ? 0x00800000u : 0x80000000u;
#pragma unroll
      for (int i = 9; i < 15; ++i) blk.w[i] = 0;
      blk.w[15] = step == 16 ? (64u + 33u) * 8u : (64u + 32u) * 8u;
      cs = outer ? kout : kin;
    }
    acc = sha256_compress(cs, blk);
    // ---- where the result goes ----
    if (step == 2 || step == 9 || step == 17) kk = acc;                      // K = HMAC(K, ...)
    else if (pad_in) kin = acc;
    else if (pad_out) kout = acc;
    else if (step == 6 || step == 13 || step == 15 || step == 21) v = acc;   // V = HMAC(K, V)
    if (step == 15) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {  // int(V) >> 4
        const uint32_t lo = v.h[7 - i], hi = i < 7 ? v.h[6 - i] : 0u;
        cand.w[i] = (lo >> 4) | (hi << 28);
      }
      if (!u256_is_zero(cand) && u256_lt(cand, U256_N)) { accepted = true; break; }
      if (++rejected == max_rejected) break;
    }
    step = step == 21 ? 14 : step + 1;
  }
  if (!accepted) {
    st.kin = kin; st.kout = kout; st.v = v;
#pragma unroll
    for (int i = 0; i < 8; ++i) cand.w[i] = 0;
  }
  if (rejected_out) *rejected_out = rejected;
  return accepted;
}

// The whole generator in one call (the scalar path and batches below the compaction threshold).  After 64 rejected
// candidates - unreachable in practice - the nonce is 0, which sign_attempt rejects as SP_SIGN_BAD_INPUT (k out of range).
// MAX_REJECTED / rejected_out: for tools/ubench/rfc6979_chain.hip (what the retry chain costs).
template <int MAX_REJECTED = 64>
__device__ __forceinline__ u256 rfc6979_nonce(const u256& z, const u256& d, uint64_t seed, int* rejected_out = nullptr) {
  rfc_input in;
  rfc6979_prepare(z, d, seed, in);
  rfc_state st;
  u256 cand;
  (void)rfc6979_run<true>(in, st, MAX_REJECTED, cand, rejected_out);
  return cand;
}

}  // namespace sp

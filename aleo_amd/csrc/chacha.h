// chacha.h — the prover's random stream: ChaCha20 in counter mode under the proof's 32-byte seed, the same definition on the host (blinding
// scalars, hiding polynomials) and on the device (the 3|H| mask coefficients are drawn in HBM: frops.hip k_fr_random).
//
// Upstream draws every blinding value with `Fr::rand(rng)` from the caller's CSPRNG (`rand::thread_rng()` at
// /root/reference/rust/src/program/execute.rs:74, `StdRng::from_entropy()` at /root/reference/wasm/src/programs/macros.rs:80 — StdRng IS ChaCha);
// here the caller hands over 32 bytes of entropy per proof and every element is a pure function of (seed, index), so host and device agree
// without moving the stream and the restatement (oracle/varuna_ref.py random_fr) can reproduce any element.
//   block(seed, counter, nonce): D. J. Bernstein's ChaCha20, 64-bit block counter (words 12-13), 64-bit nonce (words 14-15), 20 rounds;
//   element i: counter = i, nonce = attempt 0, 1, …; a block holds two candidates (bytes 0-31, 32-63, little-endian, low 253 bits);
//   the first candidate below r is the element (rejection sampling: uniform over Fr).
#pragma once
#include <cstdint>

namespace aleo_mi355x {

#define ALEO_CHACHA_QR(a, b, c, d)                       \
  a += b; d ^= a; d = (d << 16) | (d >> 16);             \
  c += d; b ^= c; b = (b << 12) | (b >> 20);             \
  a += b; d ^= a; d = (d << 8) | (d >> 24);              \
  c += d; b ^= c; b = (b << 7) | (b >> 25);

__host__ __device__ inline void chacha20_block(uint32_t out[16], const uint32_t key[8], uint64_t counter, uint64_t nonce) {
  const uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                           (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)nonce, (uint32_t)(nonce >> 32)};
  uint32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6], x7 = in[7], x8 = in[8], x9 = in[9], x10 = in[10], x11 = in[11],
           x12 = in[12], x13 = in[13], x14 = in[14], x15 = in[15];
  for (int i = 0; i < 10; ++i) {
    ALEO_CHACHA_QR(x0, x4, x8, x12) ALEO_CHACHA_QR(x1, x5, x9, x13) ALEO_CHACHA_QR(x2, x6, x10, x14) ALEO_CHACHA_QR(x3, x7, x11, x15)
    ALEO_CHACHA_QR(x0, x5, x10, x15) ALEO_CHACHA_QR(x1, x6, x11, x12) ALEO_CHACHA_QR(x2, x7, x8, x13) ALEO_CHACHA_QR(x3, x4, x9, x14)
  }
  out[0] = x0 + in[0]; out[1] = x1 + in[1]; out[2] = x2 + in[2]; out[3] = x3 + in[3]; out[4] = x4 + in[4]; out[5] = x5 + in[5]; out[6] = x6 + in[6]; out[7] = x7 + in[7];
  out[8] = x8 + in[8]; out[9] = x9 + in[9]; out[10] = x10 + in[10]; out[11] = x11 + in[11]; out[12] = x12 + in[12]; out[13] = x13 + in[13]; out[14] = x14 + in[14]; out[15] = x15 + in[15];
}

// r as eight 32-bit words, least significant first
#define ALEO_FR_MODULUS_U32 {0x00000001u, 0x0a118000u, 0xd0000001u, 0x59aa76feu, 0x5c37b001u, 0x60b44d1eu, 0x9a2ca556u, 0x12ab655eu}

// canonical little-endian words of stream element `index`
__host__ __device__ inline void chacha_fr(uint32_t v[8], const uint32_t key[8], uint64_t index) {
  const uint32_t P[8] = ALEO_FR_MODULUS_U32;
  for (uint64_t attempt = 0;; ++attempt) {
    uint32_t blk[16]; chacha20_block(blk, key, index, attempt);
    for (int h = 0; h < 2; ++h) {
      uint32_t* c = blk + 8 * h; c[7] &= 0x1fffffffu;     // low 253 bits
      bool lt = false;
      for (int i = 7; i >= 0; --i) { if (c[i] != P[i]) { lt = c[i] < P[i]; break; } }
      if (lt) { for (int i = 0; i < 8; ++i) v[i] = c[i]; return; }
    }
  }
}

struct Seed32 { uint32_t w[8]; };                          // passed to kernels by value

}  // namespace aleo_mi355x

// poseidon.hpp — snarkVM's Poseidon on the host: parameter generation (Grain LFSR), the permutation, the duplex sponge, the console hash over Fr
// (`Network::hash_psd2/4/8`) and the prover's Fiat-Shamir sponge over Fq (`PoseidonSponge<Fq, 2, 1>` behind `AlgebraicSponge`).
//
// Replaces, for the prove path reached from /root/reference/rust/src/program/execute.rs:74 (`trace.prove_execution`), snarkVM 0.14.5 [UPSTREAM-RECALL]
//   fields/src/traits/{poseidon_grain_lfsr,poseidon_default}.rs   parameters: 8 full + 31 partial rounds, alpha = 17, Cauchy MDS from the LFSR
//   algorithms/src/crypto_hash/poseidon.rs                        PoseidonSponge: absorb native / non-native / bytes, squeeze (short) non-native
//   console/algorithms/src/poseidon/                              Poseidon<E, RATE>::hash_many (used by /root/reference/rust/src/account/encryptor.rs:37-67)
// The transcript is host work by nature (a dependent chain of ~600 Fq products per permutation, ~30 permutations per proof): one GPU lane would take
// ~0.5 ms per permutation, a host core ~20 us.  Independent of oracle/ (test infrastructure); tests/test_poseidon.py runs the reference's
// private-key-ciphertext known answer through THIS code (rates 2 and 8 over Fr) via the C ABI.
#pragma once
#include "host_field.hpp"
#include <vector>
#include <mutex>
#include <memory>

namespace aleo_mi355x { namespace host {

// ---- parameters ---------------------------------------------------------------------------------------------------------------------------
struct GrainLFSR {                                         // 80 bits: [01 | s-box 0000 | field bits (12) | width (12) | full (10) | partial (10) | thirty ones]
  bool st[80]; int head = 0;
  GrainLFSR(uint64_t field_bits, uint64_t width, uint64_t full, uint64_t partial) {
    for (bool& b : st) b = false;
    st[1] = true;
    auto put = [&](int lo, int hi, uint64_t v) { for (int i = hi; i >= lo; --i) { st[i] = v & 1; v >>= 1; } };
    put(6, 17, field_bits); put(18, 29, width); put(30, 39, full); put(40, 49, partial);
    for (int i = 50; i < 80; ++i) st[i] = true;
    for (int i = 0; i < 160; ++i) update();
  }
  bool update() {
    const int h = head;
    const bool b = st[(h + 62) % 80] ^ st[(h + 51) % 80] ^ st[(h + 38) % 80] ^ st[(h + 23) % 80] ^ st[(h + 13) % 80] ^ st[h];
    st[h] = b; head = (h + 1) % 80; return b;
  }
  bool bit() { bool b = update(); while (!b) { update(); b = update(); } return update(); }
  template <int N> void raw(uint64_t* l, int bits) {       // `bits` output bits, most significant first, as an integer
    for (int i = 0; i < N; ++i) l[i] = 0;
    for (int i = bits - 1; i >= 0; --i) if (bit()) l[i / 64] |= 1ull << (i % 64);
  }
};

template <int N> struct FieldBits;
template <> struct FieldBits<4> { static constexpr int BITS = 253; };
template <> struct FieldBits<6> { static constexpr int BITS = 377; };

static constexpr int POSEIDON_FULL = 8, POSEIDON_PARTIAL = 31, POSEIDON_ROUNDS = POSEIDON_FULL + POSEIDON_PARTIAL;   // alpha = 17

template <int N, int RATE> struct PoseidonParams {
  static constexpr int W = RATE + 1;
  HFp<N> ark[POSEIDON_ROUNDS][W], mds[W][W];               // Montgomery form
  PoseidonParams() {
    GrainLFSR g(FieldBits<N>::BITS, W, POSEIDON_FULL, POSEIDON_PARTIAL);
    for (int r = 0; r < POSEIDON_ROUNDS; ++r)
      for (int i = 0; i < W; ++i) {                        // rejection sampling
        HFp<N> v; do g.raw<N>(v.l, FieldBits<N>::BITS); while (HFp<N>::geq_p(v.l));
        ark[r][i] = HFp<N>::to_mont(v);
      }
    HFp<N> xs[W], ys[W];
    auto mod_p = [&](HFp<N>& o) { HFp<N> v; g.raw<N>(v.l, FieldBits<N>::BITS); if (HFp<N>::geq_p(v.l)) HFp<N>::sub_p(v.l); o = HFp<N>::to_mont(v); };   // 2^BITS < 2p
    for (int i = 0; i < W; ++i) mod_p(xs[i]);
    for (int i = 0; i < W; ++i) mod_p(ys[i]);
    for (int i = 0; i < W; ++i) for (int j = 0; j < W; ++j) mds[i][j] = HFp<N>::inv(HFp<N>::add(xs[i], ys[j]));
    optimise_partial_rounds();
  }
  // The partial rounds in their sparse form (the well-known optimisation of the Poseidon paper, derived here for this round order — constants, s-box,
  // matrix): a partial round is x <- M S(x + c) with S acting on coordinate 0 only.  Write a matrix B = [[b00, v], [w, B^]] as N'' N' with
  // N' = diag(1, B^) and N'' = [[b00, v B^-1], [w, I]]; N' commutes with S, so with z_j = N'_j x_j the rounds become z_{j+1} = N''_j S(z_j + N'_j c_j)
  // where B_j = N'_{j+1} M is factored from the last partial round backwards (B_last = M) and the leftover N'_0 goes into the matrix of the full round
  // before them (pre = N'_0 M).  Constants on coordinates >= 1 pass through S unchanged and are pushed into the next round's constant (at the end: into
  // the first full round after), so a partial round adds ONE constant, raises ONE element to the 17th and multiplies by a matrix with 2 W - 1 entries
  // (W = 3: 5 products instead of 9; W = 9: 17 instead of 81).  Same outputs, bit for bit (tests/test_poseidon.py against the plain restatement).
  HFp<N> pre[W][W], sp_m00[POSEIDON_PARTIAL], sp_v[POSEIDON_PARTIAL][W - 1], sp_w[POSEIDON_PARTIAL][W - 1], sp_c[POSEIDON_PARTIAL], ark_after[W];
  void optimise_partial_rounds() {
    using F = HFp<N>; constexpr int T = W - 1, H = POSEIDON_FULL / 2, RP = POSEIDON_PARTIAL;
    F B[W][W]; for (int i = 0; i < W; ++i) for (int j = 0; j < W; ++j) B[i][j] = mds[i][j];
    std::vector<std::vector<F>> hat(RP, std::vector<F>(T * T));                     // B^_j: the lower-right block of N'_j
    for (int j = RP - 1; j >= 0; --j) {
      F a[T][2 * T];                                                                 // Gauss-Jordan inverse of the lower-right block
      for (int i = 0; i < T; ++i) for (int k = 0; k < T; ++k) { a[i][k] = B[1 + i][1 + k]; a[i][T + k] = i == k ? F::one() : F::zero(); hat[j][i * T + k] = B[1 + i][1 + k]; }
      for (int col = 0; col < T; ++col) {
        int piv = col; while (piv < T && a[piv][col].is_zero()) ++piv;               // an MDS block is invertible: a pivot exists
        if (piv != col) for (int k = 0; k < 2 * T; ++k) std::swap(a[piv][k], a[col][k]);
        const F inv = F::inv(a[col][col]);
        for (int k = 0; k < 2 * T; ++k) a[col][k] = F::mul(a[col][k], inv);
        for (int r = 0; r < T; ++r) if (r != col && !a[r][col].is_zero()) { const F f = a[r][col]; for (int k = 0; k < 2 * T; ++k) a[r][k] = F::sub(a[r][k], F::mul(f, a[col][k])); }
      }
      sp_m00[j] = B[0][0];
      for (int i = 0; i < T; ++i) { F acc = F::zero(); for (int k = 0; k < T; ++k) acc = F::add(acc, F::mul(B[0][1 + k], a[k][T + i])); sp_v[j][i] = acc; sp_w[j][i] = B[1 + i][0]; }
      F nb[W][W];                                                                    // N'_j M = [[row 0 of M], [B^_j (rows 1.. of M)]]
      for (int k = 0; k < W; ++k) nb[0][k] = mds[0][k];
      for (int i = 0; i < T; ++i) for (int k = 0; k < W; ++k) { F acc = F::zero(); for (int q = 0; q < T; ++q) acc = F::add(acc, F::mul(hat[j][i * T + q], mds[1 + q][k])); nb[1 + i][k] = acc; }
      for (int i = 0; i < W; ++i) for (int k = 0; k < W; ++k) B[i][k] = nb[i][k];
    }
    for (int i = 0; i < W; ++i) for (int k = 0; k < W; ++k) pre[i][k] = B[i][k];
    F carry[W]; for (auto& v : carry) v = F::zero();
    for (int j = 0; j < RP; ++j) {
      F d[W]; d[0] = ark[H + j][0];                                                  // N'_j c_j
      for (int i = 0; i < T; ++i) { F acc = F::zero(); for (int q = 0; q < T; ++q) acc = F::add(acc, F::mul(hat[j][i * T + q], ark[H + j][1 + q])); d[1 + i] = acc; }
      for (int i = 0; i < W; ++i) d[i] = F::add(d[i], carry[i]);
      sp_c[j] = d[0];
      F c0 = F::zero(); for (int i = 0; i < T; ++i) c0 = F::add(c0, F::mul(sp_v[j][i], d[1 + i]));      // N''_j (0, d_1 ..)
      carry[0] = c0; for (int i = 0; i < T; ++i) carry[1 + i] = d[1 + i];
    }
    for (int i = 0; i < W; ++i) ark_after[i] = F::add(ark[H + RP][i], carry[i]);
  }
  static const PoseidonParams& get() { static const PoseidonParams p; return p; }        // built on first use (thread-safe static)
};

template <int N> __attribute__((always_inline)) inline HFp<N> pow17(const HFp<N>& x) {
  HFp<N> a = fsqr(x); a = fsqr(a); a = fsqr(a); a = fsqr(a); return fmul(a, x);
}

template <int N, int RATE> inline void poseidon_permute(HFp<N>* s) {
  constexpr int W = RATE + 1;
  const PoseidonParams<N, RATE>& P = PoseidonParams<N, RATE>::get();
  constexpr int H = POSEIDON_FULL / 2, RP = POSEIDON_PARTIAL;
  auto full = [&](const HFp<N> (&ark)[W], const HFp<N> (&m)[W][W]) {
    for (int i = 0; i < W; ++i) s[i] = pow17(HFp<N>::add(s[i], ark[i]));
    HFp<N> o[W];
    for (int i = 0; i < W; ++i) { Wide<N> w; w.set_mul(s[0].l, m[i][0].l); for (int j = 1; j < W; ++j) w.add_mul(s[j].l, m[i][j].l); o[i] = w.redc(); }
    for (int i = 0; i < W; ++i) s[i] = o[i];
  };
  for (int r = 0; r < H; ++r) full(P.ark[r], r == H - 1 ? P.pre : P.mds);
  for (int j = 0; j < RP; ++j) {                             // sparse form (PoseidonParams::optimise_partial_rounds)
    const HFp<N> x = pow17(HFp<N>::add(s[0], P.sp_c[j]));
    Wide<N> w; w.set_mul(x.l, P.sp_m00[j].l); for (int i = 1; i < W; ++i) w.add_mul(s[i].l, P.sp_v[j][i - 1].l);
    for (int i = 1; i < W; ++i) s[i] = HFp<N>::add(s[i], fmul(P.sp_w[j][i - 1], x));
    s[0] = w.redc();
  }
  full(P.ark_after, P.mds);
  for (int r = H + RP + 1; r < POSEIDON_ROUNDS; ++r) full(P.ark[r], P.mds);
}

// Duplex sponge; state[0] = capacity, state[1..RATE] = rate; elements in Montgomery form
template <int N, int RATE> struct PoseidonSponge {
  HFp<N> s[RATE + 1]; bool absorbing = true; int pos = 0; uint64_t permutations = 0;
  PoseidonSponge() { for (auto& v : s) v = HFp<N>::zero(); }
  void permute() { poseidon_permute<N, RATE>(s); ++permutations; }
  void absorb(const HFp<N>* e, size_t n) {
    if (!n) return;
    if (!absorbing || pos == RATE) { permute(); pos = 0; }
    absorbing = true;
    for (size_t i = 0; i < n; ++i) {
      if (pos == RATE) { permute(); pos = 0; }
      s[1 + pos] = HFp<N>::add(s[1 + pos], e[i]); ++pos;
    }
  }
  void squeeze(HFp<N>* out, size_t n) {
    if (!n) return;
    if (absorbing) { permute(); pos = 0; absorbing = false; }
    for (size_t i = 0; i < n; ++i) {
      if (pos == RATE) { permute(); pos = 0; }
      out[i] = s[1 + pos]; ++pos;
    }
  }
};

// Field::new_domain_separator = from_bytes_le_mod_order(text) for short texts (< 31 bytes over Fr)
inline HFr fr_domain_separator(const char* text) {
  HFr v = HFr::zero(); size_t n = std::strlen(text); if (n > 31) n = 31;
  std::memcpy(v.l, text, n); return HFr::to_mont(v);
}

// Poseidon<E, RATE>::hash_many over Fr: preimage [domain "AleoPoseidon{RATE}", len, 0 … RATE, inputs…]; inputs / outputs in Montgomery form
template <int RATE> inline void poseidon_hash_many_fr(const HFr* in, size_t n, HFr* out, size_t n_out) {
  static const HFr dom = fr_domain_separator(RATE == 2 ? "AleoPoseidon2" : RATE == 4 ? "AleoPoseidon4" : "AleoPoseidon8");
  PoseidonSponge<4, RATE> sp;
  std::vector<HFr> pre(RATE + n, HFr::zero());
  pre[0] = dom; pre[1] = HFr::from_u64((uint64_t)n);
  for (size_t i = 0; i < n; ++i) pre[RATE + i] = in[i];
  sp.absorb(pre.data(), pre.size()); sp.squeeze(out, n_out);
}

// ---- the prover's Fiat-Shamir sponge -------------------------------------------------------------------------------------------------------
// Fr elements enter as 5 limbs of 51 bits (find_parameters(377, 253, Weight)), most significant limb first, two neighbouring limbs packed into
// one Fq element as first * 2^53 + second (53 = 51 + the two bits of overhead upstream books for a limb with one addition); challenges are cut
// from the low 376 bits of squeezed elements, most significant bit first: 252 bits for a round challenge, 168 for an opening challenge.
struct FiatShamir {
  static constexpr int LIMBS = 5, LIMB_BITS = 51, PACK_SHIFT = 53, CAP_BITS = 376, FULL_BITS = 252, SHORT_BITS = 168;
  PoseidonSponge<6, 2> sp;
  void absorb_native(const HFq* e, size_t n) { sp.absorb(e, n); }
  // G1 affine points in snarkVM's in-memory layout (x | y Montgomery, optional infinity byte at offset 96): (x, y), infinity as (0, 1)
  void absorb_g1(const uint8_t* aff, size_t stride, size_t n) {
    std::vector<HFq> e(2 * n);
    for (size_t i = 0; i < n; ++i) {
      const uint8_t* p = aff + i * stride;
      const bool inf = stride >= 97 && p[96];
      if (inf) { e[2 * i] = HFq::zero(); e[2 * i + 1] = HFq::one(); }
      else { std::memcpy(e[2 * i].l, p, 48); std::memcpy(e[2 * i + 1].l, p + 48, 48); }
    }
    sp.absorb(e.data(), e.size());
  }
  void absorb_bytes(const uint8_t* data, size_t n) {       // bits of every byte most significant first, chunks of 376 bits read as big-endian integers
    std::vector<HFq> e;
    const size_t total = 8 * n;
    for (size_t at = 0; at < total; at += CAP_BITS) {
      const size_t len = total - at < (size_t)CAP_BITS ? total - at : (size_t)CAP_BITS;
      HFq v = HFq::zero();
      for (size_t b = 0; b < len; ++b) {                   // bit `at + b` of the stream lands at position len − 1 − b of the integer
        const size_t src = at + b;
        if ((data[src / 8] >> (7 - src % 8)) & 1) { const size_t dst = len - 1 - b; v.l[dst / 64] |= 1ull << (dst % 64); }
      }
      e.push_back(HFq::to_mont(v));
    }
    sp.absorb(e.data(), e.size());
  }
  void absorb_fr(const HFr* mont, size_t n) {              // absorb_nonnative_field_elements
    std::vector<uint64_t> limbs; limbs.reserve(LIMBS * n);
    for (size_t i = 0; i < n; ++i) {
      const HFr c = HFr::from_mont(mont[i]);
      for (int k = LIMBS - 1; k >= 0; --k) {
        const int lo = LIMB_BITS * k; uint64_t v = c.l[lo / 64] >> (lo % 64);
        if (lo % 64 + LIMB_BITS > 64 && lo / 64 + 1 < 4) v |= c.l[lo / 64 + 1] << (64 - lo % 64);
        limbs.push_back(v & ((1ull << LIMB_BITS) - 1));
      }
    }
    std::vector<HFq> e;
    for (size_t i = 0; i < limbs.size();) {
      HFq v = HFq::zero();
      if (i + 1 < limbs.size()) { const u128 w = ((u128)limbs[i] << PACK_SHIFT) + limbs[i + 1]; v.l[0] = (uint64_t)w; v.l[1] = (uint64_t)(w >> 64); i += 2; }
      else { v.l[0] = limbs[i]; i += 1; }
      e.push_back(HFq::to_mont(v));
    }
    sp.absorb(e.data(), e.size());
  }
  // n challenges of `width` bits from one get_bits call (⌈n width / 376⌉ squeezed elements); Montgomery form
  void squeeze_fr(HFr* out, size_t n, int width) {
    if (!n) return;
    const size_t total = n * (size_t)width, cnt = (total + CAP_BITS - 1) / CAP_BITS;
    std::vector<HFq> e(cnt); sp.squeeze(e.data(), cnt);
    for (auto& v : e) v = HFq::from_mont(v);
    auto bit = [&](size_t b) -> uint64_t {                 // bit b of the stream: element b / 376, its bit 375 − b % 376
      const size_t p = CAP_BITS - 1 - b % CAP_BITS; return (e[b / CAP_BITS].l[p / 64] >> (p % 64)) & 1;
    };
    for (size_t i = 0; i < n; ++i) {
      HFr v = HFr::zero();
      for (int b = 0; b < width; ++b) if (bit(i * width + b)) { const int dst = width - 1 - b; v.l[dst / 64] |= 1ull << (dst % 64); }
      out[i] = HFr::to_mont(v);                            // < 2^252 < r
    }
  }
  void squeeze_full(HFr* out, size_t n) { squeeze_fr(out, n, FULL_BITS); }
  HFr squeeze_short() { HFr v; squeeze_fr(&v, 1, SHORT_BITS); return v; }
};

}}  // namespace aleo_mi355x::host

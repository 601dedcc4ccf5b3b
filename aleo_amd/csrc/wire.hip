// wire.hip — wire formats of the prove path's results: compressed G1 points, canonical Fr bytes, bech32m
// strings and the byte layout of a Varuna proof.  Host code of the product library (no kernel here): a proof is ~900 bytes.
//
// Replaces, for the MI355X backend's results, what snarkVM 0.14.5 does when a proof leaves the prover [UPSTREAM-RECALL]:
//   utilities/src/serialize + curves/src/templates/short_weierstrass_jacobian/affine.rs   CanonicalSerialize (compressed) of G1Affine
//   fields/src/fp_256.rs                                                                  Fr::to_bytes_le / from_bytes_le (canonical, little-endian)
//   synthesizer/snark/src/proof + algorithms/src/snark/varuna/data_structures/proof.rs    Proof::to_bytes_le, Display = bech32m "proof1..."
// The reference holds one such string — the `proof` field of TRANSACTION_STRING at /root/reference/wasm/src/programs/transaction.rs:100
// (round-tripped by its test at :104-120) — and the layout below was read off its 901 payload bytes:
//   [0] version 0 | u64 #batch sizes | u64 per circuit | per instance (w, z_a, z_b) | option tag + mask_poly | g_1, h_1 | every g_a, every g_b,
//   every g_c | h_2 | evaluations: z_b per instance, g_1, every g_a, g_b, g_c | sums per circuit | openings; tests/golden/reference_proof.json
//   pins every field for one circuit (where "every g_a, g_b, g_c" and "g_a, g_b, g_c per circuit" coincide).
// Compressed G1: 48 bytes little-endian x; in the last byte bit 7 = "y is the lexicographically larger root", bit 6 = infinity.
#include "ctx.h"
#include "host_field.hpp"
#include <cstring>
#include <string>
#include <vector>

namespace aleo_mi355x { namespace host {

// ---- Fq square root: Tonelli-Shanks with the two-adicity 46 of q - 1 ------------------------------------------------
struct SqrtCtx {
  uint64_t t[6];            // (q - 1) / 2^46
  uint64_t t1h[6];          // (t + 1) / 2
  uint64_t half[6];         // (q - 1) / 2
  HFq c;                    // z^t for the smallest quadratic non-residue z: a generator of the 2^46-torsion
  SqrtCtx() {
    uint64_t qm1[6]; std::memcpy(qm1, HParams<6>::P, 48); qm1[0] -= 1;
    auto shr = [](uint64_t* o, const uint64_t* a, int s) { for (int i = 0; i < 6; ++i) o[i] = (a[i] >> s) | (i + 1 < 6 && s ? a[i + 1] << (64 - s) : 0); };
    shr(t, qm1, 46); shr(half, qm1, 1);
    uint64_t tp1[6]; std::memcpy(tp1, t, 48); for (int i = 0; i < 6 && ++tp1[i] == 0; ++i) {}
    shr(t1h, tp1, 1);
    for (uint64_t z = 2;; ++z) {
      HFq zz = HFq::from_u64(z);
      if (!(HFq::pow(zz, half, 6) == HFq::one())) { c = HFq::pow(zz, t, 6); break; }
    }
  }
};
static const SqrtCtx& sqrt_ctx() { static const SqrtCtx s; return s; }

// r = sqrt(a) if a is a square (either root); returns false otherwise
static bool fq_sqrt(HFq& r, const HFq& a) {
  if (a.is_zero()) { r = a; return true; }
  const SqrtCtx& S = sqrt_ctx();
  HFq x = HFq::pow(a, S.t1h, 6), b = HFq::pow(a, S.t, 6), c = S.c; int m = 46;
  const HFq one = HFq::one();
  while (!(b == one)) {
    int i = 0; HFq b2 = b;
    while (!(b2 == one)) { b2 = HFq::sqr(b2); if (++i >= m) return false; }      // order of b does not divide 2^(m-1): a is a non-residue
    HFq e = c; for (int k = 0; k < m - i - 1; ++k) e = HFq::sqr(e);
    x = HFq::mul(x, e); c = HFq::sqr(e); b = HFq::mul(b, c); m = i;
  }
  r = x; return true;
}
static bool canon_gt(const uint64_t* a, const uint64_t* b) {       // a > b as 6-limb integers
  for (int i = 5; i >= 0; --i) { if (a[i] > b[i]) return true; if (a[i] < b[i]) return false; }
  return false;
}
static bool y_is_larger(const HFq& y) {                             // y > q - y on canonical representatives
  HFq yc = HFq::from_mont(y), nc = HFq::from_mont(HFq::neg(y));
  return canon_gt(yc.l, nc.l);
}
static bool g1_on_curve(const HFq& x, const HFq& y) { return HFq::sqr(y) == HFq::add(HFq::mul(HFq::sqr(x), x), HFq::one()); }
static bool g1_in_subgroup(const HFq& x, const HFq& y) {            // r * P == O
  HXYZZ P; P.X = x; P.Y = y; P.ZZ = HFq::one(); P.ZZZ = HFq::one();
  HXYZZ acc = HXYZZ::infinity();
  for (int bit = 252; bit >= 0; --bit) { acc = hdouble(acc); if ((HParams<4>::P[bit >> 6] >> (bit & 63)) & 1) acc = hadd(acc, P); }
  return acc.is_inf();
}

// ---- bech32m (BIP-350) ------------------------------------------------------------------------------------------------
static const char B32[] = "qpzry9x8gf2tvdw0s3jn54khce6mua7l";
static uint32_t polymod(const std::vector<uint8_t>& v) {
  static const uint32_t gen[5] = {0x3b6a57b2u, 0x26508e6du, 0x1ea119fau, 0x3d4233ddu, 0x2a1462b3u};
  uint32_t chk = 1;
  for (uint8_t x : v) { uint32_t b = chk >> 25; chk = ((chk & 0x1ffffffu) << 5) ^ x; for (int i = 0; i < 5; ++i) if ((b >> i) & 1) chk ^= gen[i]; }
  return chk;
}
static std::vector<uint8_t> hrp_expand(const std::string& hrp) {
  std::vector<uint8_t> v; for (char ch : hrp) v.push_back((uint8_t)ch >> 5); v.push_back(0); for (char ch : hrp) v.push_back((uint8_t)ch & 31); return v;
}

}}  // namespace aleo_mi355x::host

using namespace aleo_mi355x;
using namespace aleo_mi355x::host;

extern "C" {

int32_t aleo_mi355x_g1_compress(void* out48, const void* affine104, size_t n) {
  try {
    if ((!out48 || !affine104) && n) return ALEO_MI355X_ERR_BAD_ARG;
    for (size_t i = 0; i < n; ++i) {
      const uint8_t* a = (const uint8_t*)affine104 + 104 * i; uint8_t* o = (uint8_t*)out48 + 48 * i;
      if (a[96]) { std::memset(o, 0, 48); o[47] = 0x40; continue; }
      HFq x, y; std::memcpy(x.l, a, 48); std::memcpy(y.l, a + 48, 48);
      HFq xc = HFq::from_mont(x); std::memcpy(o, xc.l, 48);
      if (y_is_larger(y)) o[47] |= 0x80;
    }
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_g1_decompress(void* out_affine104, const void* in48, size_t n, int32_t check_subgroup) {
  try {
    if ((!out_affine104 || !in48) && n) return ALEO_MI355X_ERR_BAD_ARG;
    for (size_t i = 0; i < n; ++i) {
      const uint8_t* b = (const uint8_t*)in48 + 48 * i; uint8_t* o = (uint8_t*)out_affine104 + 104 * i;
      std::memset(o, 0, 104);
      const uint8_t flags = b[47];
      if (flags & 0x40) {                                      // infinity: Affine::zero() = (0, 1, true); no other bit may be set
        bool clean = (flags & 0xbf) == 0; for (int k = 0; k < 47; ++k) clean = clean && b[k] == 0;
        if (!clean) { g_last_error = "g1_decompress: infinity flag with a non-zero x"; return ALEO_MI355X_ERR_BAD_ARG; }
        HFq one = HFq::one(); std::memcpy(o + 48, one.l, 48); o[96] = 1; continue;
      }
      HFq xc; std::memcpy(xc.l, b, 48); ((uint8_t*)xc.l)[47] &= 0x3f;
      if (HFq::geq_p(xc.l)) { g_last_error = "g1_decompress: x is not a canonical field element"; return ALEO_MI355X_ERR_BAD_ARG; }
      HFq x = HFq::to_mont(xc), y;
      if (!fq_sqrt(y, HFq::add(HFq::mul(HFq::sqr(x), x), HFq::one()))) { g_last_error = "g1_decompress: x is not on the curve"; return ALEO_MI355X_ERR_BAD_ARG; }
      if (y_is_larger(y) != ((flags & 0x80) != 0)) y = HFq::neg(y);
      if (check_subgroup && !g1_in_subgroup(x, y)) { g_last_error = "g1_decompress: point is not in the prime-order subgroup"; return ALEO_MI355X_ERR_BAD_ARG; }
      std::memcpy(o, x.l, 48); std::memcpy(o + 48, y.l, 48);
    }
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_to_bytes(void* out32, const void* fr_mont, size_t n) {
  try {
    if ((!out32 || !fr_mont) && n) return ALEO_MI355X_ERR_BAD_ARG;
    for (size_t i = 0; i < n; ++i) { HFr a; std::memcpy(a.l, (const uint8_t*)fr_mont + 32 * i, 32); a = HFr::from_mont(a); std::memcpy((uint8_t*)out32 + 32 * i, a.l, 32); }
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_from_bytes(void* out_mont, const void* in32, size_t n) {
  try {
    if ((!out_mont || !in32) && n) return ALEO_MI355X_ERR_BAD_ARG;
    for (size_t i = 0; i < n; ++i) {
      HFr a; std::memcpy(a.l, (const uint8_t*)in32 + 32 * i, 32);
      if (HFr::geq_p(a.l)) { g_last_error = "fr_from_bytes: value is not below the modulus"; return ALEO_MI355X_ERR_BAD_ARG; }
      a = HFr::to_mont(a); std::memcpy((uint8_t*)out_mont + 32 * i, a.l, 32);
    }
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

// out: NUL-terminated string "<hrp>1<data><checksum>"; returns BAD_ARG when cap is too small (needs hrp + 1 + ceil(8 len / 5) + 6 + 1)
int32_t aleo_mi355x_bech32m_encode(char* out, size_t cap, const char* hrp, const void* data, size_t len) {
  try {
    if (!out || !hrp || (!data && len)) return ALEO_MI355X_ERR_BAD_ARG;
    const std::string h(hrp);
    if (h.empty()) { g_last_error = "bech32m_encode: empty prefix"; return ALEO_MI355X_ERR_BAD_ARG; }
    std::vector<uint8_t> d; uint32_t acc = 0; int bits = 0;
    for (size_t i = 0; i < len; ++i) { acc = (acc << 8) | ((const uint8_t*)data)[i]; bits += 8; while (bits >= 5) { bits -= 5; d.push_back((acc >> bits) & 31); } }
    if (bits) d.push_back((acc << (5 - bits)) & 31);
    std::vector<uint8_t> v = hrp_expand(h); v.insert(v.end(), d.begin(), d.end()); v.insert(v.end(), 6, 0);
    const uint32_t pm = polymod(v) ^ 0x2bc830a3u;
    for (int i = 0; i < 6; ++i) d.push_back((pm >> (5 * (5 - i))) & 31);
    if (h.size() + 1 + d.size() + 1 > cap) { g_last_error = "bech32m_encode: output buffer too small"; return ALEO_MI355X_ERR_BAD_ARG; }
    std::memcpy(out, h.data(), h.size()); out[h.size()] = '1';
    for (size_t i = 0; i < d.size(); ++i) out[h.size() + 1 + i] = B32[d[i]];
    out[h.size() + 1 + d.size()] = 0;
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

// *len: in = capacity of out, out = payload bytes.  Rejects a bad checksum, mixed case and non-zero padding.
int32_t aleo_mi355x_bech32m_decode(void* out, size_t* len, char* hrp_out, size_t hrp_cap, const char* s) {
  try {
    if (!out || !len || !s) return ALEO_MI355X_ERR_BAD_ARG;
    const std::string str(s); const size_t pos = str.rfind('1');
    if (pos == std::string::npos || pos == 0 || pos + 7 > str.size()) { g_last_error = "bech32m_decode: no separator / too short"; return ALEO_MI355X_ERR_BAD_ARG; }
    const std::string hrp = str.substr(0, pos);
    std::vector<uint8_t> d;
    for (size_t i = pos + 1; i < str.size(); ++i) { const char* q = std::strchr(B32, str[i]); if (!q || !str[i]) { g_last_error = "bech32m_decode: invalid character"; return ALEO_MI355X_ERR_BAD_ARG; } d.push_back((uint8_t)(q - B32)); }
    std::vector<uint8_t> v = hrp_expand(hrp); v.insert(v.end(), d.begin(), d.end());
    if (polymod(v) != 0x2bc830a3u) { g_last_error = "bech32m_decode: bad checksum"; return ALEO_MI355X_ERR_BAD_ARG; }
    d.resize(d.size() - 6);
    std::vector<uint8_t> bytes; uint32_t acc = 0; int bits = 0;
    for (uint8_t x : d) { acc = ((acc << 5) | x) & 0xfffu; bits += 5; if (bits >= 8) { bits -= 8; bytes.push_back((acc >> bits) & 0xff); } }
    if (bits >= 5 || (acc & ((1u << bits) - 1u))) { g_last_error = "bech32m_decode: bad padding"; return ALEO_MI355X_ERR_BAD_ARG; }
    if (bytes.size() > *len || (hrp_out && hrp.size() + 1 > hrp_cap)) { g_last_error = "bech32m_decode: output buffer too small"; return ALEO_MI355X_ERR_BAD_ARG; }
    std::memcpy(out, bytes.data(), bytes.size()); *len = bytes.size();
    if (hrp_out) { std::memcpy(hrp_out, hrp.data(), hrp.size()); hrp_out[hrp.size()] = 0; }
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

// Byte layout of a Varuna proof (see the file header; field order read off the reference's own proof string, batch of one
// circuit with one instance; the multi-circuit order is [UPSTREAM-RECALL]).  *len: in = capacity, out = bytes written.
int32_t aleo_mi355x_proof_to_bytes(void* out, size_t* len, const aleo_mi355x_proof_parts* p) {
  try {
    if (!out || !len || !p || !p->batch_sizes || !p->n_circuits) return ALEO_MI355X_ERR_BAD_ARG;
    size_t instances = 0; for (size_t i = 0; i < p->n_circuits; ++i) instances += p->batch_sizes[i];
    if (!p->witness_commitments || !p->g_1 || !p->h_1 || !p->g_abc || !p->h_2 || (!p->evaluations && p->n_evaluations) || !p->sums ||
        (!p->opening_points && p->n_openings)) { g_last_error = "proof_to_bytes: missing part"; return ALEO_MI355X_ERR_BAD_ARG; }
    for (size_t i = 0; i < p->n_openings; ++i)
      if (p->opening_has_v && p->opening_has_v[i] && !p->opening_random_v) { g_last_error = "proof_to_bytes: random_v flagged but not given"; return ALEO_MI355X_ERR_BAD_ARG; }
    std::vector<uint8_t> b;
    auto u64le = [&](uint64_t v) { for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i))); };
    int32_t rc = ALEO_MI355X_OK;
    auto g1 = [&](const void* aff, size_t count) { size_t o = b.size(); b.resize(o + 48 * count); int32_t r = aleo_mi355x_g1_compress(b.data() + o, aff, count); if (r) rc = r; };
    auto fr = [&](const void* m, size_t count) { size_t o = b.size(); b.resize(o + 32 * count); int32_t r = aleo_mi355x_fr_to_bytes(b.data() + o, m, count); if (r) rc = r; };
    b.push_back(0);                                                  // version
    u64le(p->n_circuits); for (size_t i = 0; i < p->n_circuits; ++i) u64le(p->batch_sizes[i]);
    g1(p->witness_commitments, 3 * instances);
    b.push_back(p->mask_poly ? 1 : 0); if (p->mask_poly) g1(p->mask_poly, 1);
    g1(p->g_1, 1); g1(p->h_1, 1);
    // the parts list g_a, g_b, g_c (and their evaluations) circuit by circuit; upstream's Commitments / Evaluations hold one vector per matrix
    // [UPSTREAM-RECALL]: every g_a, then every g_b, then every g_c.  With one circuit — the reference's proof string — both orders coincide.
    const size_t m = p->n_circuits;
    for (size_t M = 0; M < 3; ++M) for (size_t j = 0; j < m; ++j) g1((const uint8_t*)p->g_abc + 104 * (3 * j + M), 1);
    g1(p->h_2, 1);
    if (p->n_evaluations != instances + 1 + 3 * m) { g_last_error = "proof_to_bytes: expected one z_b evaluation per instance, g_1 and three per circuit"; return ALEO_MI355X_ERR_BAD_ARG; }
    fr(p->evaluations, instances + 1);
    for (size_t M = 0; M < 3; ++M) for (size_t j = 0; j < m; ++j) fr((const uint8_t*)p->evaluations + 32 * (instances + 1 + 3 * j + M), 1);
    u64le(p->n_circuits); fr(p->sums, 3 * p->n_circuits);
    u64le(p->n_openings);
    for (size_t i = 0; i < p->n_openings; ++i) {
      g1((const uint8_t*)p->opening_points + 104 * i, 1);
      const bool has = p->opening_has_v && p->opening_has_v[i];
      b.push_back(has ? 1 : 0); if (has) fr((const uint8_t*)p->opening_random_v + 32 * i, 1);
    }
    b.push_back(0);                                                  // BatchLCProof.evaluations: None
    if (rc) return rc;
    if (b.size() > *len) { g_last_error = "proof_to_bytes: output buffer too small"; *len = b.size(); return ALEO_MI355X_ERR_BAD_ARG; }
    std::memcpy(out, b.data(), b.size()); *len = b.size();
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

}  // extern "C"

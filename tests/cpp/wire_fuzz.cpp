// Host-only sanitizer driver ("build C++ host code with -fsanitize=address,undefined in tests"): the product's parsers of
// untrusted bytes — bech32m_decode, g1_decompress, fr_from_bytes, proof_to_bytes — and the host Poseidon / Fiat-Shamir sponge / random stream,
// compiled from aleo_amd/csrc/{wire,sponge}.hip as plain C++ with AddressSanitizer + UBSan (tools/asan_host.sh) and driven with the reference's
// proof string, malformed variants of it and a seeded mutation loop.  No GPU, no HIP runtime call.  Prints SANITIZED OK.
#include "aleo_mi355x.h"
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#define CHECK(c, msg) do { if (!(c)) { std::printf("FAIL: %s (line %d) [%s]\n", msg, __LINE__, aleo_mi355x_last_error()); return 1; } } while (0)

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: wire_fuzz <proof1...>\n"); return 2; }
  const std::string proof = argv[1];
  std::vector<uint8_t> raw(4096); size_t len = raw.size(); char hrp[16];
  CHECK(aleo_mi355x_bech32m_decode(raw.data(), &len, hrp, sizeof hrp, proof.c_str()) == 0 && len == 901 && std::strcmp(hrp, "proof") == 0, "the reference's proof decodes");
  raw.resize(len);
  std::vector<char> enc(4 * len + 32);
  CHECK(aleo_mi355x_bech32m_encode(enc.data(), enc.size(), "proof", raw.data(), len) == 0 && proof == enc.data(), "and re-encodes to the same string");
  // exact-size and too-small output buffers
  { std::vector<uint8_t> exact(901); size_t l = 901; CHECK(aleo_mi355x_bech32m_decode(exact.data(), &l, nullptr, 0, proof.c_str()) == 0, "exact buffer");
    size_t l2 = 900; std::vector<uint8_t> small(900); CHECK(aleo_mi355x_bech32m_decode(small.data(), &l2, nullptr, 0, proof.c_str()) != 0, "short buffer refused");
    char h2[3]; size_t l3 = 901; CHECK(aleo_mi355x_bech32m_decode(exact.data(), &l3, h2, sizeof h2, proof.c_str()) != 0, "short prefix buffer refused");
    std::vector<char> e2(proof.size()); CHECK(aleo_mi355x_bech32m_encode(e2.data(), e2.size(), "proof", raw.data(), len) != 0, "short string buffer refused"); }
  // the twelve points: decompress (subgroup check on), recompress
  const size_t offs[12] = {17, 65, 113, 162, 210, 258, 306, 354, 402, 450, 770, 851};
  std::vector<uint8_t> aff(104 * 12), comp(48 * 12), back(48 * 12);
  for (int i = 0; i < 12; ++i) std::memcpy(&comp[48 * i], &raw[offs[i]], 48);
  CHECK(aleo_mi355x_g1_decompress(aff.data(), comp.data(), 12, 1) == 0, "reference points decompress");
  CHECK(aleo_mi355x_g1_compress(back.data(), aff.data(), 12) == 0 && back == comp, "and recompress to the same bytes");
  // mutation loop over the string: every outcome must be an error code or a clean decode, never a fault
  size_t ok = 0, bad = 0;
  for (int it = 0; it < 4000; ++it) {
    std::string s = proof; const int kind = (int)(rnd() % 6);
    if (kind == 0) s[rnd() % s.size()] = (char)(rnd() & 0x7f);
    else if (kind == 1) s.resize(rnd() % s.size());
    else if (kind == 2) s.insert(rnd() % s.size(), 1, "qpzry9x8gf2tvdw0s3jn54khce6mua7l"[rnd() % 32]);
    else if (kind == 3) s = s.substr(rnd() % 8);
    else if (kind == 4) { for (auto& ch : s) if (rnd() % 97 == 0) ch = (char)('A' + rnd() % 26); }
    else s += std::string(rnd() % 5, '1');
    for (auto& ch : s) if (!ch) ch = 'q';
    std::vector<uint8_t> out(1024); size_t l = out.size(); char h[8];
    if (aleo_mi355x_bech32m_decode(out.data(), &l, h, sizeof h, s.c_str()) == 0) ++ok; else ++bad;
  }
  CHECK(bad > 3000, "mutated strings are refused");
  // random 48-byte strings into g1_decompress, random 32-byte strings into fr_from_bytes
  size_t pts = 0;
  for (int it = 0; it < 300; ++it) {
    uint8_t c48[48], a104[104]; for (auto& b : c48) b = (uint8_t)rnd();
    if (it % 3 == 0) c48[47] &= 0x81;                                   // small x: mostly canonical
    if (aleo_mi355x_g1_decompress(a104, c48, 1, it % 2) == 0) { ++pts; uint8_t r48[48]; CHECK(aleo_mi355x_g1_compress(r48, a104, 1) == 0, "recompress"); }
    uint8_t f32[32], m32[32], b32[32]; for (auto& b : f32) b = (uint8_t)rnd();
    if (it % 2) f32[31] &= 0x0f;
    if (aleo_mi355x_fr_from_bytes(m32, f32, 1) == 0) { CHECK(aleo_mi355x_fr_to_bytes(b32, m32, 1) == 0 && std::memcmp(b32, f32, 32) == 0, "Fr bytes round trip"); }
  }
  // proof_to_bytes from parts: the reference proof reassembled, then missing parts / inconsistent counts / short buffers
  {
    std::vector<uint8_t> ev(32 * 5), sums(32 * 3), rv(32 * 2, 0);
    CHECK(aleo_mi355x_fr_from_bytes(ev.data(), &raw[498], 5) == 0 && aleo_mi355x_fr_from_bytes(sums.data(), &raw[666], 3) == 0 && aleo_mi355x_fr_from_bytes(rv.data(), &raw[819], 1) == 0, "field elements");
    aleo_mi355x_proof_parts p{}; uint64_t bs[1] = {1}; uint8_t has_v[2] = {1, 0};
    p.batch_sizes = bs; p.n_circuits = 1; p.witness_commitments = &aff[0]; p.mask_poly = &aff[104 * 3]; p.g_1 = &aff[104 * 4]; p.h_1 = &aff[104 * 5];
    p.g_abc = &aff[104 * 6]; p.h_2 = &aff[104 * 9]; p.evaluations = ev.data(); p.n_evaluations = 5; p.sums = sums.data();
    p.opening_points = &aff[104 * 10]; p.opening_random_v = rv.data(); p.opening_has_v = has_v; p.n_openings = 2;
    std::vector<uint8_t> out(901); size_t l = out.size();
    CHECK(aleo_mi355x_proof_to_bytes(out.data(), &l, &p) == 0 && l == 901 && out == raw, "the reference proof reassembled byte for byte");
    size_t l2 = 900; CHECK(aleo_mi355x_proof_to_bytes(out.data(), &l2, &p) != 0 && l2 == 901, "short buffer: size reported");
    aleo_mi355x_proof_parts q = p; q.n_evaluations = 4; l = 901; CHECK(aleo_mi355x_proof_to_bytes(out.data(), &l, &q) != 0, "wrong evaluation count refused");
    q = p; q.g_abc = nullptr; l = 901; CHECK(aleo_mi355x_proof_to_bytes(out.data(), &l, &q) != 0, "missing part refused");
    q = p; q.opening_random_v = nullptr; l = 901; CHECK(aleo_mi355x_proof_to_bytes(out.data(), &l, &q) != 0, "flagged random_v without data refused");
    q = p; q.n_circuits = 0; l = 901; CHECK(aleo_mi355x_proof_to_bytes(out.data(), &l, &q) != 0, "no circuits refused");
  }
  // Poseidon / sponge / random stream: sizes around the block boundaries
  {
    uint8_t in[32 * 20] = {0}, out[32 * 9];
    for (int i = 0; i < 20; ++i) in[32 * i] = (uint8_t)(i + 1);
    for (uint32_t rate : {2u, 4u, 8u}) for (size_t n : {0u, 1u, 2u, 7u, 8u, 9u, 20u}) CHECK(aleo_mi355x_poseidon_hash_fr(rate, in, n, out, 9) == 0, "poseidon_hash_fr");
    CHECK(aleo_mi355x_poseidon_hash_fr(3, in, 1, out, 1) != 0, "rate 3 refused");
    uint64_t h = 0; CHECK(aleo_mi355x_fs_new(&h) == 0, "fs_new");
    for (size_t n : {0u, 1u, 46u, 47u, 48u, 94u, 95u, 200u}) { std::vector<uint8_t> b(n + 1, 0xa5); CHECK(aleo_mi355x_fs_absorb_bytes(h, b.data(), n) == 0, "fs_absorb_bytes"); }
    CHECK(aleo_mi355x_fs_absorb_g1(h, aff.data(), 104, 12) == 0 && aleo_mi355x_fs_absorb_g1(h, aff.data(), 100, 1) != 0, "fs_absorb_g1");
    CHECK(aleo_mi355x_fs_absorb_fr(h, in, 20) == 0, "fs_absorb_fr");
    uint8_t ch[32 * 40];
    for (size_t n : {0u, 1u, 2u, 3u, 40u}) CHECK(aleo_mi355x_fs_squeeze_fr(h, ch, n, 0) == 0 && aleo_mi355x_fs_squeeze_fr(h, ch, n, 1) == 0, "fs_squeeze_fr");
    CHECK(aleo_mi355x_fs_free(h) == 0 && aleo_mi355x_fs_free(h) != 0 && aleo_mi355x_fs_absorb_fr(h, in, 1) != 0, "a freed sponge is refused");
    uint8_t seed[32] = {7}, fr[32 * 33];
    CHECK(aleo_mi355x_fr_random(fr, 33, seed, 0xFFFFFFFFFFFFFFF0ull) == 0 && aleo_mi355x_fr_random(fr, 1, nullptr, 0) != 0, "fr_random");
  }
  std::printf("SANITIZED OK: %zu mutated strings decoded cleanly, %zu refused; %zu random x were on the curve\n", ok, bad, pts);
  return 0;
}

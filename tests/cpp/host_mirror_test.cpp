// C++ host-mirror test (built and run by tests/test_gpu_parity.py::test_cpp_host_mirror on the GPU box; only compiled
// and linked by the CPU tier).  Checks the drop-in boundary the way a snarkVM unit test would: VariableBase::msm against
// the structured identity sum_i s_i*(i+1)G == (sum_i s_i*(i+1))G, the one-shot call against the pinned call, and
// EvaluationDomain round trips / zero padding.  Exit code 0 = all checks passed.
#include <cstdio>
#include <cstring>
#include <vector>
#include "aleo_mi355x.hpp"
using namespace aleo_mi355x;

typedef unsigned __int128 u128;
static const uint64_t R_MOD[4] = {0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL};

static uint64_t sm_state;
static uint64_t splitmix() { sm_state += 0x9E3779B97F4A7C15ULL; uint64_t z = sm_state; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static bool lt_r(const uint64_t* a) { for (int i = 3; i >= 0; --i) { if (a[i] < R_MOD[i]) return true; if (a[i] > R_MOD[i]) return false; } return false; }
static BigInteger256 rand_scalar() { BigInteger256 s; do { for (auto& l : s.l) l = splitmix(); s.l[3] &= (1ULL << 61) - 1; } while (!lt_r(s.l)); return s; }

// (acc + s * w) mod r with w < 2^32, all 256-bit: schoolbook then conditional subtractions (test helper)
static void mac_mod_r(uint64_t acc[4], const uint64_t s[4], uint64_t w) {
  uint64_t t[5] = {0, 0, 0, 0, 0}; u128 c = 0;
  for (int i = 0; i < 4; ++i) { c += (u128)s[i] * w + acc[i]; t[i] = (uint64_t)c; c >>= 64; }
  t[4] = (uint64_t)c;
  // reduce t (< 2^32 * r + r) by subtracting r << k for k = 40..0 (binary long division by comparison)
  for (int k = 40; k >= 0; --k) {
    uint64_t m[5] = {0, 0, 0, 0, 0};
    int limb = k / 64, sh = k % 64;
    for (int i = 0; i < 4; ++i) { m[i + limb] |= R_MOD[i] << sh; if (sh && i + limb + 1 < 5) m[i + limb + 1] |= R_MOD[i] >> (64 - sh); }
    bool ge = true; for (int i = 4; i >= 0; --i) { if (t[i] > m[i]) break; if (t[i] < m[i]) { ge = false; break; } }
    if (ge) { uint64_t br = 0; for (int i = 0; i < 5; ++i) { u128 d = (u128)t[i] - m[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } }
  }
  memcpy(acc, t, 32);
}

#define CHECK(cond, what) do { if (!(cond)) { printf("FAIL: %s\n", what); return 1; } else printf("ok: %s\n", what); } while (0)

int main() {
  if (aleo_mi355x_init_device(-1) != 0) { printf("FAIL: init: %s\n", aleo_mi355x_last_error()); return 2; }
  // G1 generator in Montgomery form (curves/src/bls12_377/g1.rs)
  G1Affine g{}; const uint64_t gx[6] = {0x1042a645ec301b95ULL, 0x5a990780c1060f28ULL, 0x684a8ab3a9007a5bULL, 0x1c35a184257ba63fULL, 0xb2b2abd2fea8e32eULL, 0x017df3a223fb2017ULL};
  const uint64_t gy[6] = {0xbc5a1ae8e2801ab9ULL, 0xd4f3c861cbfe13b0ULL, 0xecdd5ffc4e949f13ULL, 0x8f87199b7503667dULL, 0x0f0b1b837dc4fe1cULL, 0x004bcc7e053eaabeULL};
  memcpy(g.x, gx, 48); memcpy(g.y, gy, 48);
  const size_t n = 5000;
  auto pinned = PinnedBases::generate_multiples(g, 1, n);
  CHECK(pinned.is_ok(), "bases_generate");
  std::vector<G1Affine> bases(n);
  CHECK(aleo_mi355x_bases_download(pinned.value->handle(), 0, n, bases.data()) == 0, "bases_download");
  CHECK(memcmp(bases[0].x, gx, 48) == 0 && memcmp(bases[0].y, gy, 48) == 0, "first generated base is the generator (Montgomery constants agree)");
  sm_state = 0xA1E00002; std::vector<BigInteger256> s(n + 7);
  uint64_t k[4] = {0, 0, 0, 0};
  for (size_t i = 0; i < n + 7; ++i) { s[i] = rand_scalar(); if (i < n) mac_mod_r(k, s[i].l, i + 1); }
  auto a = VariableBase::msm(bases.data(), n, s.data(), n + 7);          // zips to the shorter slice
  auto b = VariableBase::msm(*pinned.value, s.data(), n);
  CHECK(a.is_ok() && b.is_ok(), "VariableBase::msm returns Ok");
  CHECK(memcmp(&*a.value, &*b.value, sizeof(G1Projective)) == 0, "one-shot msm == pinned msm");
  BigInteger256 kk; memcpy(kk.l, k, 32);
  auto kg = VariableBase::msm(&g, 1, &kk, 1);
  CHECK(kg.is_ok() && memcmp(&*a.value, &*kg.value, sizeof(G1Projective)) == 0, "sum_i s_i*(i+1)G == (sum_i s_i*(i+1) mod r)G");
  std::vector<BigInteger256> zeros(n, BigInteger256{{0, 0, 0, 0}});
  auto z = VariableBase::msm(bases.data(), n, zeros.data(), n);
  CHECK(z.is_ok() && z.value->is_zero(), "all-zero scalars give the identity");
  auto empty = VariableBase::msm(bases.data(), 0, s.data(), 0);
  CHECK(empty.is_ok() && empty.value->is_zero(), "empty msm is the identity");
  {                                                                       // the same MSM as three shards (all on device 0) through the sharded entry points
    CHECK(aleo_mi355x_init(0) == 0, "init(0) initialises every visible device");
    auto sh = ShardedBases::pin(bases.data(), n, {0, 0, 0}, true);
    CHECK(sh.is_ok() && sh.value->shards() == 3, "bases_pin_sharded");
    auto c3 = VariableBase::msm(*sh.value, s.data(), n + 7);
    CHECK(c3.is_ok() && memcmp(&*a.value, &*c3.value, sizeof(G1Projective)) == 0, "sharded msm == one-shot msm");
    auto gen3 = ShardedBases::generate_multiples(g, 1, n, {0, 0}, false);
    auto c2 = gen3.is_ok() ? VariableBase::msm(*gen3.value, s.data(), n) : Result<G1Projective>{std::nullopt, Error{1}};
    CHECK(c2.is_ok() && memcmp(&*a.value, &*c2.value, sizeof(G1Projective)) == 0, "sharded msm over generated shards == one-shot msm");
  }
  auto bad = VariableBase::msm(PinnedBases(), s.data(), 1);
  CHECK(!bad.is_ok() && bad.error.code == ALEO_MI355X_ERR_BAD_HANDLE, "an unknown handle is an Err (caller falls back to the CPU)");

  auto dom = EvaluationDomain::new_(1000);
  CHECK(dom && dom->size == 1024 && dom->log_size_of_group == 10, "EvaluationDomain::new(1000) -> size 1024");
  CHECK(!EvaluationDomain::new_((size_t)1 << 48).has_value(), "EvaluationDomain::new beyond the two-adicity is None");
  std::vector<Fr> x(1000); for (auto& e : x) { BigInteger256 v = rand_scalar(); memcpy(e.l, v.l, 32); }   // any canonical value is a valid Montgomery residue
  std::vector<Fr> y = x;
  CHECK(dom->fft_in_place(y).is_ok() && y.size() == 1024, "fft_in_place pads to the domain");
  CHECK(dom->ifft_in_place(y).is_ok(), "ifft_in_place");
  bool same = true; for (size_t i = 0; i < 1024; ++i) { Fr e = i < 1000 ? x[i] : Fr{{0, 0, 0, 0}}; same &= memcmp(&e, &y[i], 32) == 0; }
  CHECK(same, "ifft(fft(x)) == x (zero padded)");
  y = x; CHECK(dom->coset_fft_in_place(y).is_ok() && dom->coset_ifft_in_place(y).is_ok(), "coset round trip runs");
  same = true; for (size_t i = 0; i < 1000; ++i) same &= memcmp(&x[i], &y[i], 32) == 0;
  CHECK(same, "coset_ifft(coset_fft(x)) == x");
  std::vector<Fr> too_long(2000);
  CHECK(!dom->fft_in_place(too_long).is_ok(), "more coefficients than the domain is an Err");
  // KZG10::commit, one polynomial and one round's worth in a single call; compressed serialisation round trip
  CHECK(pinned.value->precompute() == 0, "bases_precompute");
  std::vector<Fr> p0(4096), p1(3000), p2(4096);
  for (auto* p : {&p0, &p1, &p2}) for (auto& e : *p) { BigInteger256 v = rand_scalar(); memcpy(e.l, v.l, 32); }
  for (size_t i = 4000; i < 4096; ++i) p2[i] = Fr{{0, 0, 0, 0}};                                       // trailing zeros are skipped
  auto c0 = KZG10::commit(*pinned.value, p0); auto c1 = KZG10::commit(*pinned.value, p1); auto c2 = KZG10::commit(*pinned.value, p2);
  CHECK(c0.is_ok() && c1.is_ok() && c2.is_ok(), "KZG10::commit returns Ok");
  auto cb = KZG10::commit_batch(*pinned.value, {&p0, &p1, &p2});
  CHECK(cb.is_ok() && cb.value->size() == 3, "KZG10::commit_batch returns Ok");
  CHECK(memcmp(&(*cb.value)[0], &*c0.value, 104) == 0 && memcmp(&(*cb.value)[1], &*c1.value, 104) == 0 && memcmp(&(*cb.value)[2], &*c2.value, 104) == 0,
        "batched commitments == one-by-one commitments, bit for bit");
  auto ser = serialize_compressed(*c0.value);
  CHECK(ser.is_ok(), "serialize_compressed");
  auto de = deserialize_compressed(*ser.value);
  CHECK(de.is_ok() && memcmp(&*de.value, &*c0.value, 104) == 0, "deserialize_compressed(serialize_compressed(C)) == C (subgroup checked)");
  CompressedG1 junk = *ser.value; junk.b[47] |= 0x40;
  CHECK(!deserialize_compressed(junk).is_ok(), "an infinity flag with a non-zero x is an Err");
  {                                                                       // the 4-step transform over four shards (all on device 0) == the single-device transform
    std::vector<Fr> a(1 << 12), b;
    for (auto& e : a) { BigInteger256 v = rand_scalar(); memcpy(e.l, v.l, 32); }
    b = a;
    auto d12 = EvaluationDomain::new_(1 << 12);
    CHECK(d12->coset_fft_in_place(a).is_ok() && d12->in_place_sharded(b, NTTDirection::Forward, NTTType::Coset, {0, 0, 0, 0}).is_ok(), "sharded coset_fft returns Ok");
    CHECK(memcmp(a.data(), b.data(), a.size() * 32) == 0, "sharded coset_fft == coset_fft, bit for bit");
    CHECK(d12->in_place_sharded(b, NTTDirection::Inverse, NTTType::Coset, {0, 0}).is_ok() && d12->coset_ifft_in_place(a).is_ok() && memcmp(a.data(), b.data(), a.size() * 32) == 0, "sharded coset_ifft == coset_ifft");
    CHECK(!d12->in_place_sharded(b, NTTDirection::Forward, NTTType::Standard, {0, 0, 0}).is_ok(), "three shards are refused (power of two)");
  }
  printf("ALL OK\n");
  return 0;
}

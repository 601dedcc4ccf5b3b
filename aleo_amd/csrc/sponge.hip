// sponge.hip — C ABI of the host-side Poseidon (poseidon.hpp) and of the prover's random stream on the host (chacha.h).  No kernel here: the
// transcript of a proof is a dependent chain of Fq products, host work (see poseidon.hpp); these entry points let a host side that drives the
// rounds itself (aleo_amd/varuna.py, a Rust shim) run exactly the sponge aleo_mi355x_varuna_prove* runs inside the library.
#include "ctx.h"
#include "poseidon.hpp"
#include "chacha.h"
#include <cstring>
#include <vector>
#include <mutex>
#include <unordered_map>
#include <memory>

namespace aleo_mi355x { namespace {
using host::HFr; using host::HFq; using host::FiatShamir;
std::mutex g_fs_mu; std::unordered_map<uint64_t, std::shared_ptr<FiatShamir>> g_fs; uint64_t g_fs_next = 1;
std::shared_ptr<FiatShamir> fs_find(uint64_t h) {
  std::lock_guard<std::mutex> g(g_fs_mu); auto it = g_fs.find(h);
  if (it == g_fs.end()) { g_last_error = "unknown sponge handle"; return nullptr; }
  return it->second;
}
bool fr_canonical(const void* p, size_t n, std::vector<HFr>& mont) {
  mont.resize(n);
  for (size_t i = 0; i < n; ++i) { HFr v; std::memcpy(v.l, (const uint8_t*)p + 32 * i, 32); if (HFr::geq_p(v.l)) return false; mont[i] = HFr::to_mont(v); }
  return true;
}
}}  // namespace

using namespace aleo_mi355x;

extern "C" {

int32_t aleo_mi355x_poseidon_hash_fr(uint32_t rate, const void* inputs, size_t n_inputs, void* out, size_t n_out) {
  try {
    if ((!inputs && n_inputs) || (!out && n_out) || (rate != 2 && rate != 4 && rate != 8)) { g_last_error = "poseidon_hash_fr: rate 2, 4 or 8; non-null buffers"; return ALEO_MI355X_ERR_BAD_ARG; }
    std::vector<HFr> in, o(n_out);
    if (!fr_canonical(inputs, n_inputs, in)) { g_last_error = "poseidon_hash_fr: input not canonical"; return ALEO_MI355X_ERR_BAD_ARG; }
    if (rate == 2) host::poseidon_hash_many_fr<2>(in.data(), in.size(), o.data(), n_out);
    else if (rate == 4) host::poseidon_hash_many_fr<4>(in.data(), in.size(), o.data(), n_out);
    else host::poseidon_hash_many_fr<8>(in.data(), in.size(), o.data(), n_out);
    for (size_t i = 0; i < n_out; ++i) { const HFr c = HFr::from_mont(o[i]); std::memcpy((uint8_t*)out + 32 * i, c.l, 32); }
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

// The parameters themselves (canonical 32-byte values): ark[39][rate + 1] then mds[rate + 1][rate + 1], row-major — what a circuit that constrains
// the permutation needs (aleo_amd/synth.py poseidon_chain_r1cs builds the R1CS of hash_psd2 gadgets from them).
int32_t aleo_mi355x_poseidon_parameters_fr(uint32_t rate, void* ark_out, void* mds_out) {
  try {
    if (!ark_out || !mds_out || (rate != 2 && rate != 4 && rate != 8)) { g_last_error = "poseidon_parameters_fr: rate 2, 4 or 8; non-null buffers"; return ALEO_MI355X_ERR_BAD_ARG; }
    auto emit = [&](auto& P, uint32_t W) {
      uint8_t* a = (uint8_t*)ark_out; uint8_t* m = (uint8_t*)mds_out;
      for (int r = 0; r < host::POSEIDON_ROUNDS; ++r) for (uint32_t i = 0; i < W; ++i) { const HFr c = HFr::from_mont(P.ark[r][i]); std::memcpy(a + 32 * (r * W + i), c.l, 32); }
      for (uint32_t i = 0; i < W; ++i) for (uint32_t j = 0; j < W; ++j) { const HFr c = HFr::from_mont(P.mds[i][j]); std::memcpy(m + 32 * (i * W + j), c.l, 32); }
    };
    if (rate == 2) emit(host::PoseidonParams<4, 2>::get(), 3); else if (rate == 4) emit(host::PoseidonParams<4, 4>::get(), 5); else emit(host::PoseidonParams<4, 8>::get(), 9);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fs_new(uint64_t* sponge) {
  try {
    if (!sponge) return ALEO_MI355X_ERR_BAD_ARG;
    auto fs = std::make_shared<FiatShamir>();
    std::lock_guard<std::mutex> g(g_fs_mu); *sponge = g_fs_next++; g_fs[*sponge] = std::move(fs);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fs_free(uint64_t sponge) {
  try {
    std::lock_guard<std::mutex> g(g_fs_mu);
    if (!g_fs.erase(sponge)) { g_last_error = "unknown sponge handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fs_absorb_bytes(uint64_t sponge, const void* data, size_t len) {
  try {
    if (!data && len) return ALEO_MI355X_ERR_BAD_ARG;
    auto fs = fs_find(sponge); if (!fs) return ALEO_MI355X_ERR_BAD_HANDLE;
    fs->absorb_bytes((const uint8_t*)data, len);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fs_absorb_g1(uint64_t sponge, const void* affine, size_t stride, size_t count) {
  try {
    if ((!affine && count) || (stride != 104 && stride != 96)) { g_last_error = "fs_absorb_g1: stride 104 or 96"; return ALEO_MI355X_ERR_BAD_ARG; }
    auto fs = fs_find(sponge); if (!fs) return ALEO_MI355X_ERR_BAD_HANDLE;
    for (size_t i = 0; i < count; ++i) {                    // coordinates must be reduced: the sponge adds them as field elements
      const uint64_t* p = (const uint64_t*)((const uint8_t*)affine + i * stride);
      if (HFq::geq_p(p) || HFq::geq_p(p + 6)) { g_last_error = "fs_absorb_g1: coordinate not reduced"; return ALEO_MI355X_ERR_BAD_ARG; }
    }
    fs->absorb_g1((const uint8_t*)affine, stride, count);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fs_absorb_fr(uint64_t sponge, const void* fr_canonical_, size_t count) {
  try {
    if (!fr_canonical_ && count) return ALEO_MI355X_ERR_BAD_ARG;
    auto fs = fs_find(sponge); if (!fs) return ALEO_MI355X_ERR_BAD_HANDLE;
    std::vector<HFr> m;
    if (!fr_canonical(fr_canonical_, count, m)) { g_last_error = "fs_absorb_fr: input not canonical"; return ALEO_MI355X_ERR_BAD_ARG; }
    fs->absorb_fr(m.data(), m.size());
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fs_squeeze_fr(uint64_t sponge, void* out_canonical, size_t count, int32_t short_) {
  try {
    if (!out_canonical && count) return ALEO_MI355X_ERR_BAD_ARG;
    auto fs = fs_find(sponge); if (!fs) return ALEO_MI355X_ERR_BAD_HANDLE;
    std::vector<HFr> o(count);
    fs->squeeze_fr(o.data(), count, short_ ? FiatShamir::SHORT_BITS : FiatShamir::FULL_BITS);
    for (size_t i = 0; i < count; ++i) { const HFr c = HFr::from_mont(o[i]); std::memcpy((uint8_t*)out_canonical + 32 * i, c.l, 32); }
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_random(void* out, size_t n, const uint8_t seed[32], uint64_t first_index) {
  try {
    if ((!out && n) || !seed) return ALEO_MI355X_ERR_BAD_ARG;
    uint32_t key[8]; std::memcpy(key, seed, 32);
    for (size_t i = 0; i < n; ++i) { uint32_t w[8]; chacha_fr(w, key, first_index + i); std::memcpy((uint8_t*)out + 32 * i, w, 32); }
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

}  // extern "C"

// aleo_mi355x.hpp — C++ host-side mirror of the snarkVM operator interfaces, over the C ABI in aleo_mi355x.h.
//
// The reference's host code is Rust (snarkVM 0.14.5 behind /root/reference/rust/src/program/execute.rs:74); there is no
// Rust toolchain in the build image, so the host side above the C ABI is this header.  Names, argument meaning and
// error behaviour follow the reference interfaces [UPSTREAM-RECALL]:
//   snarkvm_algorithms::msm::VariableBase::msm(bases, scalars) -> Projective        (zips to the shorter slice)
//   snarkvm_algorithms::fft::EvaluationDomain::{new, fft_in_place, ifft_in_place, coset_fft_in_place, coset_ifft_in_place}
//   snarkvm_algorithms::polycommit::kzg10::KZG10::commit(powers, polynomial) -> Commitment (G1Affine); the commitments of one
//       prover round (SonicKZG10::commit over several labelled polynomials) as ONE call; CanonicalSerialize (compressed) of it
//   snarkvm_algorithms_cuda::{msm, NTT}: Result<_, Error> — an Err means "recompute on the CPU"
//   snarkvm_synthesizer_snark::ProvingKey::prove_batch / snarkvm_algorithms::snark::varuna::{CircuitProvingKey, Proof}: a proving key bound
//       to one circuit, `prove_batch(&[assignment])` -> Proof, Proof Display as bech32m `proof1…`; `prove_batch(keys -> assignments)` over several
//       keys and `Trace::prove_execution / prove_fee` above it (rows a6 / a7 of SURVEY.md §8)
// Layouts are snarkVM's: Fr = 4 x u64 Montgomery, scalar = 4 x u64 canonical, G1Affine = 104 bytes, Projective = 144 bytes.
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <deque>
#include <future>
#include <mutex>
#include <optional>
#include <random>
#include <string>
#include <thread>
#include <utility>
#include <vector>
#include "aleo_mi355x.h"

namespace aleo_mi355x {

struct Fr { uint64_t l[4]; };                                   // Montgomery form
struct BigInteger256 { uint64_t l[4]; };                        // canonical scalar
struct G1Affine { uint64_t x[6], y[6]; uint8_t infinity; uint8_t pad[7]; };
static_assert(sizeof(G1Affine) == 104, "snarkVM Affine layout");
struct G1Projective { uint64_t x[6], y[6], z[6]; bool is_zero() const { uint64_t o = 0; for (auto v : z) o |= v; return o == 0; } };
static_assert(sizeof(G1Projective) == 144, "snarkVM Projective layout");

struct Error {
  int32_t code;
  std::string message() const { return std::string(aleo_mi355x_strerror(code)) + " [" + aleo_mi355x_last_error() + "]"; }
  bool unsatisfied() const { return code == ALEO_MI355X_ERR_UNSATISFIED; }      // a prover refused an assignment that violates its circuit
};
template <class T> struct Result {       // Ok(value) or Err(Error), like the Rust side sees it
  std::optional<T> value; Error error{0};
  bool is_ok() const { return value.has_value(); }
};

// A base set resident in HBM (Arc<Vec<G1Affine>> on the Rust side).
class PinnedBases {
 public:
  PinnedBases() = default;
  static Result<PinnedBases> pin(const G1Affine* bases, size_t n) {
    PinnedBases p; int32_t rc = aleo_mi355x_bases_pin(bases, sizeof(G1Affine), n, &p.handle_);
    if (rc) return {std::nullopt, Error{rc}};
    p.n_ = n; return {std::move(p), Error{0}};
  }
  static Result<PinnedBases> generate_multiples(const G1Affine& base, uint64_t first, size_t n) {
    PinnedBases p; int32_t rc = aleo_mi355x_bases_generate(&base, first, n, &p.handle_);
    if (rc) return {std::nullopt, Error{rc}};
    p.n_ = n; return {std::move(p), Error{0}};
  }
  PinnedBases(PinnedBases&& o) noexcept : handle_(o.handle_), n_(o.n_) { o.handle_ = 0; }
  PinnedBases& operator=(PinnedBases&& o) noexcept { release(); handle_ = o.handle_; n_ = o.n_; o.handle_ = 0; return *this; }
  PinnedBases(const PinnedBases&) = delete; PinnedBases& operator=(const PinnedBases&) = delete;
  ~PinnedBases() { release(); }
  int32_t precompute() { return aleo_mi355x_bases_precompute(handle_); }
  uint64_t handle() const { return handle_; }
  size_t len() const { return n_; }
  static PinnedBases adopt(uint64_t handle, size_t n) { PinnedBases p; p.handle_ = handle; p.n_ = n; return p; }      // takes ownership of a handle the C ABI returned
 private:
  void release() { if (handle_) { aleo_mi355x_bases_unpin(handle_); handle_ = 0; } }
  uint64_t handle_ = 0; size_t n_ = 0;
};

// One base set over several devices of this process: contiguous point shards, shard g on devices[g] (aleo_mi355x_bases_pin_sharded).
class ShardedBases {
 public:
  ShardedBases() = default;
  static Result<ShardedBases> pin(const G1Affine* bases, size_t n, const std::vector<int32_t>& devices, bool precompute = true) {
    ShardedBases p; int32_t rc = aleo_mi355x_bases_pin_sharded(bases, sizeof(G1Affine), n, devices.data(), devices.size(), precompute ? 1 : 0, &p.handle_);
    if (rc) return {std::nullopt, Error{rc}};
    p.n_ = n; p.shards_ = devices.size(); return {std::move(p), Error{0}};
  }
  static Result<ShardedBases> generate_multiples(const G1Affine& base, uint64_t first, size_t n, const std::vector<int32_t>& devices, bool precompute = true) {
    ShardedBases p; int32_t rc = aleo_mi355x_bases_generate_sharded(&base, first, n, devices.data(), devices.size(), precompute ? 1 : 0, &p.handle_);
    if (rc) return {std::nullopt, Error{rc}};
    p.n_ = n; p.shards_ = devices.size(); return {std::move(p), Error{0}};
  }
  ShardedBases(ShardedBases&& o) noexcept : handle_(o.handle_), n_(o.n_), shards_(o.shards_) { o.handle_ = 0; }
  ShardedBases& operator=(ShardedBases&& o) noexcept { release(); handle_ = o.handle_; n_ = o.n_; shards_ = o.shards_; o.handle_ = 0; return *this; }
  ShardedBases(const ShardedBases&) = delete; ShardedBases& operator=(const ShardedBases&) = delete;
  ~ShardedBases() { release(); }
  uint64_t handle() const { return handle_; }
  size_t len() const { return n_; }
  size_t shards() const { return shards_; }
 private:
  void release() { if (handle_) { aleo_mi355x_bases_unpin_sharded(handle_); handle_ = 0; } }
  uint64_t handle_ = 0; size_t n_ = 0, shards_ = 0;
};

struct VariableBase {
  // ONE msm over the devices of a sharded set: per-device Pippenger, partial sums added on the host in shard order
  static Result<G1Projective> msm(const ShardedBases& bases, const BigInteger256* scalars, size_t n_scalars) {
    G1Projective out{}; size_t n = bases.len() < n_scalars ? bases.len() : n_scalars;
    int32_t rc = aleo_mi355x_msm_g1_sharded(&out, bases.handle(), scalars, n, nullptr);
    if (rc) return {std::nullopt, Error{rc}};
    return {out, Error{0}};
  }
  // VariableBase::msm(bases, scalars): sum_i scalars[i] * bases[i] over the shorter of the two slices.
  static Result<G1Projective> msm(const G1Affine* bases, size_t n_bases, const BigInteger256* scalars, size_t n_scalars) {
    G1Projective out{}; size_t n = n_bases < n_scalars ? n_bases : n_scalars;
    int32_t rc = aleo_mi355x_msm_g1(&out, bases, sizeof(G1Affine), scalars, n);
    if (rc) return {std::nullopt, Error{rc}};
    return {out, Error{0}};
  }
  static Result<G1Projective> msm(const PinnedBases& bases, const BigInteger256* scalars, size_t n_scalars) {
    G1Projective out{}; size_t n = bases.len() < n_scalars ? bases.len() : n_scalars;
    int32_t rc = aleo_mi355x_msm_g1_pinned(&out, bases.handle(), scalars, n);
    if (rc) return {std::nullopt, Error{rc}};
    return {out, Error{0}};
  }
};

// KZG10::commit over a pinned SRS.  Polynomials are coefficient vectors in Montgomery form (DensePolynomial::coeffs); trailing zero
// coefficients are skipped like the reference's `skip_leading_zeros_and_convert_to_bigints`.
struct KZG10 {
  static size_t degree_plus_one(const std::vector<Fr>& p) { size_t n = p.size(); while (n && !(p[n - 1].l[0] | p[n - 1].l[1] | p[n - 1].l[2] | p[n - 1].l[3])) --n; return n; }
  static Result<G1Affine> commit(const PinnedBases& powers, const std::vector<Fr>& poly) {
    G1Affine out{}; int32_t rc = aleo_mi355x_kzg_commit(&out, powers.handle(), poly.data(), degree_plus_one(poly));
    if (rc) return {std::nullopt, Error{rc}};
    return {out, Error{0}};
  }
  // the commitments of one prover round: k polynomials, one call (shared launches on the device), results in the order given
  static Result<std::vector<G1Affine>> commit_batch(const PinnedBases& powers, const std::vector<const std::vector<Fr>*>& polys) {
    std::vector<const void*> ptrs; std::vector<size_t> lens;
    for (auto* p : polys) { ptrs.push_back(p->data()); lens.push_back(degree_plus_one(*p)); }
    std::vector<G1Affine> out(polys.size());
    int32_t rc = aleo_mi355x_kzg_commit_batch(out.data(), powers.handle(), ptrs.data(), lens.data(), polys.size());
    if (rc) return {std::nullopt, Error{rc}};
    return {std::move(out), Error{0}};
  }
};

// CanonicalSerialize / CanonicalDeserialize (compressed) of G1Affine: 48 bytes
struct CompressedG1 { uint8_t b[48]; };
inline Result<CompressedG1> serialize_compressed(const G1Affine& p) {
  CompressedG1 c{}; int32_t rc = aleo_mi355x_g1_compress(c.b, &p, 1);
  if (rc) return {std::nullopt, Error{rc}};
  return {c, Error{0}};
}
inline Result<G1Affine> deserialize_compressed(const CompressedG1& c, bool validate_subgroup = true) {
  G1Affine p{}; int32_t rc = aleo_mi355x_g1_decompress(&p, c.b, 1, validate_subgroup ? 1 : 0);
  if (rc) return {std::nullopt, Error{rc}};
  return {p, Error{0}};
}

enum class NTTInputOutputOrder : int32_t { NN = 0, NR = 1, RN = 2, RR = 3 };
enum class NTTDirection : int32_t { Forward = 0, Inverse = 1 };
enum class NTTType : int32_t { Standard = 0, Coset = 1 };

// snarkvm_algorithms_cuda::NTT(domain_size, data, order, direction, type)
inline Result<bool> NTT(size_t domain_size, Fr* inout, NTTInputOutputOrder order, NTTDirection dir, NTTType type) {
  uint32_t lg = 0; while (((size_t)1 << lg) < domain_size) ++lg;
  if (((size_t)1 << lg) != domain_size) return {std::nullopt, Error{ALEO_MI355X_ERR_BAD_ARG}};
  int32_t rc = aleo_mi355x_ntt_fr(inout, lg, (int32_t)order, (int32_t)dir, (int32_t)type);
  if (rc) return {std::nullopt, Error{rc}};
  return {true, Error{0}};
}

class EvaluationDomain {
 public:
  size_t size; uint32_t log_size_of_group;
  // EvaluationDomain::new(num_coeffs): None when the power of two exceeds the two-adicity of Fr (47)
  static std::optional<EvaluationDomain> new_(size_t num_coeffs) {
    size_t s = 1; uint32_t lg = 0; while (s < num_coeffs) { s <<= 1; ++lg; }
    if (lg > 47) return std::nullopt;
    return EvaluationDomain{s, lg};
  }
  // *_in_place resize the vector to the domain size with zeros first, as the reference does
  Result<bool> fft_in_place(std::vector<Fr>& x) const { return run(x, NTTDirection::Forward, NTTType::Standard); }
  Result<bool> ifft_in_place(std::vector<Fr>& x) const { return run(x, NTTDirection::Inverse, NTTType::Standard); }
  Result<bool> coset_fft_in_place(std::vector<Fr>& x) const { return run(x, NTTDirection::Forward, NTTType::Coset); }
  Result<bool> coset_ifft_in_place(std::vector<Fr>& x) const { return run(x, NTTDirection::Inverse, NTTType::Coset); }
  // the same transforms split over several devices of this process (4-step, one peer exchange): devices.size() a power of two
  // device-resident data (2^log_size_of_group elements at d_inout on the current device): the transform over several devices, no host buffer
  Result<bool> in_place_sharded_device(void* d_inout, NTTDirection d, NTTType t, const std::vector<int32_t>& devices, void* stream = nullptr) const {
    int32_t rc = aleo_mi355x_ntt_fr_sharded_device(d_inout, log_size_of_group, (int32_t)d, (int32_t)t, devices.data(), devices.size(), stream);
    if (rc) return {std::nullopt, Error{rc}};
    return {true, Error{0}};
  }
  Result<bool> in_place_sharded(std::vector<Fr>& x, NTTDirection d, NTTType t, const std::vector<int32_t>& devices) const {
    if (x.size() > size) return {std::nullopt, Error{ALEO_MI355X_ERR_BAD_ARG}};
    x.resize(size, Fr{{0, 0, 0, 0}});
    int32_t rc = aleo_mi355x_ntt_fr_sharded(x.data(), log_size_of_group, (int32_t)d, (int32_t)t, devices.data(), devices.size());
    if (rc) return {std::nullopt, Error{rc}};
    return {true, Error{0}};
  }
 private:
  Result<bool> run(std::vector<Fr>& x, NTTDirection d, NTTType t) const {
    if (x.size() > size) return {std::nullopt, Error{ALEO_MI355X_ERR_BAD_ARG}};
    x.resize(size, Fr{{0, 0, 0, 0}});
    return NTT(size, x.data(), NTTInputOutputOrder::NN, d, t);
  }
};


// ---- the prove call of one proving key ------------------------------------------------------------------------
// Mirrors what sits under `trace.prove_execution::<A, _>(…)` (/root/reference/rust/src/program/execute.rs:74) and `vm.execute(…)` (:177,
// transfer.rs:99): Trace::prove_execution -> ProvingKey::prove_batch(&[(pk, assignments)]) -> Varuna::prove_batch [UPSTREAM-RECALL], for ONE
// circuit with 1..32 instances.  CommitterKey = the universal SRS trimmed for the circuit (powers ‖ hiding powers, pinned in HBM);
// ProvingKey = the circuit's index in HBM (AHPForR1CS::index); prove_batch = one call of the C ABI.
struct R1CSMatrix { std::vector<uint32_t> row_ptr, col; std::vector<BigInteger256> val; };      // CSR over the variables, public ones first
struct R1CS { R1CSMatrix a, b, c; size_t num_constraints = 0, num_public = 0, num_private = 0; };
enum class DomainPolicy : uint32_t { Auto = 0, PerMatrix = 1, Shared = 2 };

class CommitterKey {
 public:
  // powers_of_beta_g[0..=max_degree] followed by powers_of_beta_times_gamma_g (>= 3): one resident set, fixed-base tables built once
  static Result<CommitterKey> pin(const G1Affine* powers_then_gamma_powers, size_t max_degree, size_t n_gamma) {
    auto b = PinnedBases::pin(powers_then_gamma_powers, max_degree + 1 + n_gamma);
    if (!b.is_ok()) return {std::nullopt, b.error};
    return finish(std::move(*b.value), max_degree, n_gamma);
  }
  // synthetic setup: P_i = s_i * base for canonical scalars s = (tau^i)_{i <= max_degree} ‖ (gamma tau^i)_{i < n_gamma}
  static Result<CommitterKey> from_scalars(const G1Affine& base, const BigInteger256* scalars, size_t max_degree, size_t n_gamma) {
    uint64_t h = 0; int32_t rc = aleo_mi355x_bases_from_scalars(&base, scalars, max_degree + 1 + n_gamma, &h);
    if (rc) return {std::nullopt, Error{rc}};
    return finish(PinnedBases::adopt(h, max_degree + 1 + n_gamma), max_degree, n_gamma);
  }
  uint64_t handle() const { return bases_.handle(); }
  size_t max_degree() const { return max_degree_; }
  size_t gamma_offset() const { return max_degree_ + 1; }
  size_t lagrange_offset() const { return lagrange_offset_; }      // 0: no Lagrange-basis powers pinned (commitments from coefficients)
  void set_lagrange_offset(size_t off) { lagrange_offset_ = off; }  // the set pinned holds L_i(tau) G of the circuit's domain H, then v_H(tau) G, from `off`
  // Large proofs over several devices (`north_star`: "large proofs shard MSM bases ... across the 8 GPUs"): a second copy of the key's
  // points cut into contiguous shards over `devices` (each device holds 1/G of the powers and of their window tables), owned by this key.  From then
  // on every commitment a prover makes against this key with at least min_points scalars in all is computed shard by shard — each device pulls its
  // slices of the coefficient vectors, 144 bytes per shard and result come back — and the proof bytes do not change (aleo_mi355x_bases_attach_shards).
  // transforms_from: the prover's transforms of at least that many elements take the same devices (0 = the library's default, 2^24)
  Error shard_over(const std::vector<int32_t>& devices, size_t min_points = (size_t)1 << 16, size_t transforms_from = 0) {
    unshard();
    std::vector<G1Affine> host(bases_.len());
    int32_t rc = aleo_mi355x_bases_download(bases_.handle(), 0, host.size(), host.data());
    if (rc) return Error{rc};
    auto sh = ShardedBases::pin(host.data(), host.size(), devices, true);
    if (!sh.is_ok()) return sh.error;
    rc = aleo_mi355x_bases_attach_shards(bases_.handle(), sh.value->handle(), min_points);
    if (rc) return Error{rc};
    if (transforms_from) { rc = aleo_mi355x_bases_shard_transforms(bases_.handle(), transforms_from); if (rc) return Error{rc}; }
    shards_ = std::move(*sh.value);
    return Error{0};
  }
  void unshard() { if (shards_.handle()) { aleo_mi355x_bases_attach_shards(bases_.handle(), 0, 0); shards_ = ShardedBases(); } }
  size_t shards() const { return shards_.shards(); }
  CommitterKey() = default;
  CommitterKey(CommitterKey&&) noexcept = default;
  CommitterKey& operator=(CommitterKey&& o) noexcept { if (this != &o) { unshard(); bases_ = std::move(o.bases_); shards_ = std::move(o.shards_); max_degree_ = o.max_degree_; lagrange_offset_ = o.lagrange_offset_; } return *this; }
  ~CommitterKey() { unshard(); }                                    // detach before the shards (and then the points) go
 private:
  static Result<CommitterKey> finish(PinnedBases b, size_t max_degree, size_t n_gamma) {
    if (n_gamma < 3) return {std::nullopt, Error{ALEO_MI355X_ERR_BAD_ARG}};
    int32_t rc = b.precompute(); if (rc) return {std::nullopt, Error{rc}};
    CommitterKey k; k.bases_ = std::move(b); k.max_degree_ = max_degree; return {std::move(k), Error{0}};
  }
  PinnedBases bases_; ShardedBases shards_; size_t max_degree_ = 0, lagrange_offset_ = 0;
};

// The caller's randomness for one proof: 32 bytes, the key of the prover's ChaCha20 stream.  The reference's call sites pass a CSPRNG
// (`rand::thread_rng()`, /root/reference/rust/src/program/execute.rs:74); from_entropy() is that.  A repeated seed repeats the blinding of a
// proof: from_u64 exists for reproducible tests only.
struct Seed {
  uint8_t bytes[32];
  static Seed from_entropy() { Seed s; std::random_device rd; for (int i = 0; i < 32; i += 4) { const uint32_t v = rd(); std::memcpy(s.bytes + i, &v, 4); } return s; }
  static Seed from_u64(uint64_t v) { Seed s; std::memset(s.bytes, 0, 32); for (int i = 0; i < 8; ++i) s.bytes[i] = (uint8_t)(v >> (8 * i)); return s; }
};

struct Proof {
  std::vector<uint8_t> bytes;                                   // Proof::write_le layout
  Result<std::string> to_string() const {                       // Display: bech32m, hrp "proof"
    std::string s(2 * bytes.size() + 16, '\0');
    int32_t rc = aleo_mi355x_bech32m_encode(s.data(), s.size(), "proof", bytes.data(), bytes.size());
    if (rc) return {std::nullopt, Error{rc}};
    s.resize(std::char_traits<char>::length(s.c_str())); return {std::move(s), Error{0}};
  }
};

class ProvingKey {
 public:
  // AHPForR1CS::index + the circuit's commitments (Process::synthesize_key, /root/reference/wasm/src/programs/manager/mod.rs:164-177)
  static Result<ProvingKey> index(const CommitterKey& ck, const R1CS& cs, DomainPolicy policy = DomainPolicy::Auto) {
    aleo_mi355x_r1cs_matrix m[3] = {{cs.a.row_ptr.data(), cs.a.col.data(), cs.a.val.data()}, {cs.b.row_ptr.data(), cs.b.col.data(), cs.b.val.data()},
                                    {cs.c.row_ptr.data(), cs.c.col.data(), cs.c.val.data()}};
    for (auto* q : {&cs.a, &cs.b, &cs.c}) if (q->row_ptr.size() != cs.num_constraints + 1 || q->col.size() != q->val.size()) return {std::nullopt, Error{ALEO_MI355X_ERR_BAD_ARG}};
    ProvingKey pk;
    int32_t rc = aleo_mi355x_varuna_index_build(&pk.handle_, ck.handle(), ck.max_degree(), ck.gamma_offset(), ck.lagrange_offset(), m, cs.num_constraints, cs.num_public, cs.num_private, (uint32_t)policy);
    if (rc) return {std::nullopt, Error{rc}};
    pk.num_variables_ = cs.num_public + cs.num_private;
    return {std::move(pk), Error{0}};
  }
  // ProvingKey::prove_batch: one assignment (public variables first, z_0 = 1, canonical) per instance; `seed` stands for the caller's RNG
  Result<Proof> prove_batch(const std::vector<const std::vector<BigInteger256>*>& assignments, const Seed& seed = Seed::from_entropy()) const {
    std::vector<const void*> p;
    for (auto* a : assignments) { if (!a || a->size() != num_variables_) return {std::nullopt, Error{ALEO_MI355X_ERR_BAD_ARG}}; p.push_back(a->data()); }
    Proof out; out.bytes.resize(1024 + 192 * assignments.size()); size_t len = out.bytes.size();
    int32_t rc = aleo_mi355x_varuna_prove_indexed(handle_, p.data(), p.size(), seed.bytes, out.bytes.data(), &len);
    if (rc) return {std::nullopt, Error{rc}};
    out.bytes.resize(len); return {std::move(out), Error{0}};
  }
  // what the verifier's transcript starts from: 12 compressed index commitments + the domain sizes
  Result<std::vector<uint8_t>> verifying_key_bytes() const {
    std::vector<uint8_t> v(1024); size_t len = v.size();
    int32_t rc = aleo_mi355x_varuna_index_vk(handle_, v.data(), &len);
    if (rc) return {std::nullopt, Error{rc}};
    v.resize(len); return {std::move(v), Error{0}};
  }
  ProvingKey() = default;
  ProvingKey(ProvingKey&& o) noexcept : handle_(o.handle_), num_variables_(o.num_variables_) { o.handle_ = 0; }
  ProvingKey& operator=(ProvingKey&& o) noexcept { release(); handle_ = o.handle_; num_variables_ = o.num_variables_; o.handle_ = 0; return *this; }
  ProvingKey(const ProvingKey&) = delete; ProvingKey& operator=(const ProvingKey&) = delete;
  ~ProvingKey() { release(); }
  uint64_t handle() const { return handle_; }
  size_t num_variables() const { return num_variables_; }
 private:
  void release() { if (handle_) { aleo_mi355x_varuna_index_free(handle_); handle_ = 0; } }
  uint64_t handle_ = 0; size_t num_variables_ = 0;
};

// Varuna::prove_batch(keys_to_constraints: &BTreeMap<&ProvingKey, &[Assignment]>): ONE proof for several proving keys, each with its instances, in
// the order given (upstream: the map's key order).  All keys must have been indexed against the same CommitterKey.
using KeyedAssignments = std::vector<std::pair<const ProvingKey*, std::vector<const std::vector<BigInteger256>*>>>;
inline Result<Proof> prove_batch(const KeyedAssignments& keyed, const Seed& seed = Seed::from_entropy()) {
  std::vector<uint64_t> handles; std::vector<size_t> counts; std::vector<const void*> p;
  for (auto& [pk, zs] : keyed) {
    if (!pk || zs.empty()) return {std::nullopt, Error{ALEO_MI355X_ERR_BAD_ARG}};
    handles.push_back(pk->handle()); counts.push_back(zs.size());
    for (auto* a : zs) { if (!a || a->size() != pk->num_variables()) return {std::nullopt, Error{ALEO_MI355X_ERR_BAD_ARG}}; p.push_back(a->data()); }
  }
  Proof out; out.bytes.resize(1024 + 400 * handles.size() + 192 * p.size()); size_t len = out.bytes.size();
  int32_t rc = aleo_mi355x_varuna_prove_batch_indexed(handles.data(), handles.size(), p.data(), counts.data(), seed.bytes, out.bytes.data(), &len);
  if (rc) return {std::nullopt, Error{rc}};
  out.bytes.resize(len); return {std::move(out), Error{0}};
}

// Independent proofs in lockstep (aleo_mi355x_varuna_prove_many): request i is what prove_batch(keyed[i], seeds[i]) would prove, byte for byte; the
// proofs share every round's commitment launch chain.  One Result per request: a request that fails carries its own error, the others complete.
inline std::vector<Result<Proof>> prove_many(const std::vector<KeyedAssignments>& requests, const std::vector<Seed>& seeds) {
  const size_t n = requests.size();
  std::vector<Result<Proof>> out; out.reserve(n);
  if (seeds.size() != n || n == 0) { for (size_t i = 0; i < n; ++i) out.push_back({std::nullopt, Error{ALEO_MI355X_ERR_BAD_ARG}}); return out; }
  std::vector<std::vector<uint64_t>> handles(n); std::vector<std::vector<size_t>> counts(n); std::vector<std::vector<const void*>> ptrs(n);
  std::vector<Proof> proofs(n); std::vector<aleo_mi355x_prove_request> rq(n); std::vector<int32_t> pre(n, 0);
  for (size_t i = 0; i < n; ++i) {
    for (auto& [pk, zs] : requests[i]) {
      if (!pk || zs.empty()) { pre[i] = ALEO_MI355X_ERR_BAD_ARG; break; }
      handles[i].push_back(pk->handle()); counts[i].push_back(zs.size());
      for (auto* a : zs) { if (!a || a->size() != pk->num_variables()) { pre[i] = ALEO_MI355X_ERR_BAD_ARG; break; } ptrs[i].push_back(a->data()); }
    }
    if (requests[i].empty()) pre[i] = ALEO_MI355X_ERR_BAD_ARG;
    proofs[i].bytes.resize(1024 + 400 * handles[i].size() + 192 * ptrs[i].size());
    rq[i] = aleo_mi355x_prove_request{handles[i].data(), pre[i] ? 0 : handles[i].size(), ptrs[i].data(), counts[i].data(), seeds[i].bytes, proofs[i].bytes.data(), proofs[i].bytes.size(), 0};
  }
  const int32_t rc = aleo_mi355x_varuna_prove_many(rq.data(), n);
  for (size_t i = 0; i < n; ++i) {
    const int32_t st = pre[i] ? pre[i] : (rq[i].status ? rq[i].status : rc);
    if (st) { out.push_back({std::nullopt, Error{st}}); continue; }
    proofs[i].bytes.resize(rq[i].len); out.push_back({std::move(proofs[i]), Error{0}});
  }
  return out;
}

// The front end that turns concurrent provers into lockstep calls: every proving thread of the host — the dev server proves each request on a
// tokio::task::spawn_blocking thread, /root/reference/rust/develop/src/routes.rs:119,149,229 — submits its request and waits on a future; ONE drainer
// thread takes whatever is queued (up to max_batch requests, waiting at most max_wait for company once the first one is there) and proves it with a
// single prove_many call.  A proof is byte for byte what prove_batch(keyed, seed) returns; only WHO calls the library changes.  The keyed assignments
// (proving keys and assignment vectors) must stay alive until the future is ready.  `drainers` such threads share the queue (default 2): two lockstep
// calls in flight fill each other's gaps — the small kernels between one call's commitments run beside the other's accumulations (2^15 constraints,
// 8 proofs per call: 222 proofs/s from one caller, 254 from two; profiles/r04_lockstep_pipeline_ab.jsonl).
class ProvingQueue {
 public:
  explicit ProvingQueue(size_t max_batch = 8, std::chrono::microseconds max_wait = std::chrono::microseconds(200), int32_t device = -1, size_t drainers = 2)      // device: the one the proving keys live on (-1: the drainer threads' default)
      : max_batch_(max_batch < 1 ? 1 : (max_batch > 64 ? 64 : max_batch)), max_wait_(max_wait), device_(device) {
    const size_t nd = drainers < 1 ? 1 : (drainers > 4 ? 4 : drainers);
    for (size_t i = 0; i < nd; ++i) drainers_.emplace_back([this] { run(); });
  }
  ~ProvingQueue() { { std::lock_guard<std::mutex> lk(mu_); stop_ = true; } cv_.notify_all(); for (auto& t : drainers_) t.join(); }
  ProvingQueue(const ProvingQueue&) = delete; ProvingQueue& operator=(const ProvingQueue&) = delete;
  std::future<Result<Proof>> submit(KeyedAssignments keyed, const Seed& seed = Seed::from_entropy()) {
    Item it; it.keyed = std::move(keyed); it.seed = seed; auto f = it.done.get_future();
    { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(it)); }
    cv_.notify_all(); return f;
  }
  size_t calls() const { std::lock_guard<std::mutex> lk(mu_); return calls_; }          // prove_many calls made so far (tests: fewer than requests)
 private:
  struct Item { KeyedAssignments keyed; Seed seed; std::promise<Result<Proof>> done; };
  void run() {
    (void)aleo_mi355x_init_device(device_);                 // selects the device for this thread; if it fails the requests below come back with the library's error
    for (;;) {
      std::vector<Item> batch;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
        if (q_.empty()) return;                                                          // stop requested and nothing left
        if (q_.size() < max_batch_ && !stop_) cv_.wait_for(lk, max_wait_, [&] { return stop_ || q_.size() >= max_batch_; });
        while (!q_.empty() && batch.size() < max_batch_) { batch.push_back(std::move(q_.front())); q_.pop_front(); }
        ++calls_;
      }
      std::vector<KeyedAssignments> reqs; std::vector<Seed> seeds;
      for (auto& it : batch) { reqs.push_back(it.keyed); seeds.push_back(it.seed); }
      auto out = prove_many(reqs, seeds);
      for (size_t i = 0; i < batch.size(); ++i) batch[i].done.set_value(std::move(out[i]));
    }
  }
  const size_t max_batch_; const std::chrono::microseconds max_wait_; const int32_t device_;
  mutable std::mutex mu_; std::condition_variable cv_; std::deque<Item> q_; bool stop_ = false; size_t calls_ = 0;
  std::vector<std::thread> drainers_;
};

// snarkvm_synthesizer_process::Trace as the prover sees it: the transitions of one transaction, each the proving key of its
// function and the assignment its execution produced.  prove_execution / prove_fee group the assignments per proving key (order of first
// appearance; upstream: BTreeMap order of the keys) and make ONE proof for all of them — the call under
// `trace.prove_execution::<A, _>(locator, rng)` at /root/reference/rust/src/program/execute.rs:74.  Inclusion proofs (state paths from the
// ledger) are further assignments of one more circuit and enter the same way; fetching them is the SDK's business.
class Trace {
 public:
  void insert_transition(const ProvingKey& pk, const std::vector<BigInteger256>& assignment) {
    for (auto& e : keyed_) if (e.first == &pk) { e.second.push_back(&assignment); return; }
    keyed_.push_back({&pk, {&assignment}});
  }
  size_t transitions() const { size_t n = 0; for (auto& e : keyed_) n += e.second.size(); return n; }
  Result<Proof> prove_execution(const Seed& seed = Seed::from_entropy()) const { return keyed_.empty() ? Result<Proof>{std::nullopt, Error{ALEO_MI355X_ERR_BAD_ARG}} : prove_batch(keyed_, seed); }
  Result<Proof> prove_fee(const Seed& seed = Seed::from_entropy()) const {              // a fee is exactly one transition
    return transitions() != 1 ? Result<Proof>{std::nullopt, Error{ALEO_MI355X_ERR_BAD_ARG}} : prove_batch(keyed_, seed);
  }
 private:
  KeyedAssignments keyed_;
};

}  // namespace aleo_mi355x

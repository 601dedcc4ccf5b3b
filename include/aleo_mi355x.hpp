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
// Layouts are snarkVM's: Fr = 4 x u64 Montgomery, scalar = 4 x u64 canonical, G1Affine = 104 bytes, Projective = 144 bytes.
#pragma once
#include <cstddef>
#include <cstdint>
#include <optional>
#include <string>
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
 private:
  void release() { if (handle_) { aleo_mi355x_bases_unpin(handle_); handle_ = 0; } }
  uint64_t handle_ = 0; size_t n_ = 0;
};

struct VariableBase {
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
 private:
  Result<bool> run(std::vector<Fr>& x, NTTDirection d, NTTType t) const {
    if (x.size() > size) return {std::nullopt, Error{ALEO_MI355X_ERR_BAD_ARG}};
    x.resize(size, Fr{{0, 0, 0, 0}});
    return NTT(size, x.data(), NTTInputOutputOrder::NN, d, t);
  }
};

}  // namespace aleo_mi355x

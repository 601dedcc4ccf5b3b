// varuna.hip — the host side of one proof, native: the four AHP rounds, the evaluations and the two openings of
// `Varuna::prove_batch` (up to 32 instances in any split over circuits) as ONE call of the C ABI (`aleo_mi355x_varuna_prove[_batch_indexed]`).
//
// Replaces (shape, not bytes — see DESIGN.md §4d for what differs from upstream and why) snarkVM 0.14.5
//   algorithms/src/snark/varuna/varuna.rs                      Varuna::prove_batch
//   algorithms/src/snark/varuna/ahp/prover/round_functions/*   AHPForR1CS::prover_{first,second,third,fourth}_round   [UPSTREAM-RECALL]
// reached from /root/reference/rust/src/program/execute.rs:74 (`trace.prove_execution`) and transfer.rs:99.
// Every circuit-sized step is a kernel of msm.hip / ntt.hip / frops.hip queued on the calling slot's stream; this file keeps what
// upstream keeps on the CPU between them: the Fiat-Shamir transcript (upstream's Poseidon sponge over Fq, poseidon.hpp), the
// challenge-dependent constants (host Fr arithmetic, host_field.hpp), the blinding scalars (ChaCha20 under the proof's 32-byte seed, chacha.h)
// and the O(|X|) public-input polynomial.  aleo_amd/varuna.py is the same sequence written against the public entry points; both
// must produce the bytes of the restatement in oracle/varuna_ref.py (tests/test_varuna.py).
#include "ctx.h"
#include "host_field.hpp"
#include "poseidon.hpp"
#include "chacha.h"
#include <cstring>
#include <vector>
#include <functional>
#include <utility>
#include <memory>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <algorithm>

namespace aleo_mi355x {

using host::HFr;
using host::HFq;

namespace {
// element `index` of the proof's random stream (chacha.h; the same definition as k_fr_random in frops.hip), Montgomery form
HFr random_fr(const Seed32& seed, uint64_t index) {
  uint32_t w[8]; chacha_fr(w, seed.w, index);
  HFr v; std::memcpy(v.l, w, 32); return HFr::to_mont(v);
}

inline HFr fr_u64(uint64_t v) { return HFr::from_u64(v); }
inline HFr vanish(uint64_t size, const HFr& x) { return HFr::sub(HFr::pow_u64(x, size), HFr::one()); }     // x^size − 1
inline void fr_bytes(uint8_t* out, const HFr& m) { HFr c = HFr::from_mont(m); std::memcpy(out, c.l, 32); }
inline HFr domain_gen(uint64_t size) {                    // TWO_ADIC_ROOT^(2^(47 − lg size))
  HFr g; std::memcpy(g.l, host::FR_TWO_ADIC_ROOT_CANON, 32); g = HFr::to_mont(g);
  int lg = 0; while ((1ull << lg) < size) ++lg;
  for (int i = lg; i < host::FR_TWO_ADICITY; ++i) g = HFr::sqr(g);
  return g;
}
inline HFr inv_pow2(uint32_t lg) {                          // 1 / 2^lg: lg products by 1/2 instead of a Fermat chain
  static const HFr half = HFr::inv(fr_u64(2));
  HFr r = HFr::one(); for (uint32_t i = 0; i < lg; ++i) r = HFr::mul(r, half); return r;
}
// a <- (sum_t a_t w^(t u))_u for a primitive |a|-th root w, |a| a power of two: bit-reversal + radix-2 butterflies, O(n log n) host products
inline void host_ntt(std::vector<HFr>& a, const HFr& w) {
  const size_t n = a.size();
  for (size_t i = 1, j = 0; i < n; ++i) { size_t bit = n >> 1; for (; j & bit; bit >>= 1) j ^= bit; j ^= bit; if (i < j) std::swap(a[i], a[j]); }
  std::vector<HFr> tw(n > 1 ? n / 2 : 1);
  tw[0] = HFr::one(); for (size_t i = 1; i < n / 2; ++i) tw[i] = HFr::mul(tw[i - 1], w);
  for (size_t len = 2; len <= n; len <<= 1)
    for (size_t i = 0; i < n; i += len)
      for (size_t t = 0; t < len / 2; ++t) {
        const HFr u = a[i + t], v = HFr::mul(a[i + t + len / 2], tw[t * (n / len)]);
        a[i + t] = HFr::add(u, v); a[i + t + len / 2] = HFr::sub(u, v);
      }
}
inline HFr horner(const std::vector<HFr>& p, const HFr& x) { HFr a = HFr::zero(); for (size_t i = p.size(); i-- > 0;) a = HFr::add(HFr::mul(a, x), p[i]); return a; }

struct Arena {                                             // bump allocation inside the slot's prover workspace
  char* base; size_t off = 0, cap;
  char* take(size_t elems) { char* p = base + off; off += (elems * 32 + 255) & ~(size_t)255; return off <= cap ? p : nullptr; }
};
#define TAKE(var, elems) char* var = ar.take(elems); if (!var) { g_last_error = "varuna_prove: workspace accounting"; return ALEO_MI355X_ERR_HIP; }
#define RC(call) { int32_t rc_ = (call); if (rc_) return rc_; }

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

thread_local double g_varuna_timing[8] = {};

// The prover's transforms.  With a sharded copy of the committer key attached (row e2: a proof that spans devices) a transform of >= shard_ntt_min elements (default 2^24, aleo_mi355x_bases_shard_transforms) runs over
// the devices the key is spread over (api.hip ntt_sharded_device: slabs pulled and pushed by peer copies, the coefficient vector stays on the prover's device
// between the rounds) — the "NTT coefficients" half of north_star's "large proofs shard MSM bases and NTT coefficients"; everything else, and any device list
// that is not a power of two, takes the single-device kernels.  Same values either way (tests: proofs byte-equal).
static bool ntt_routed(const PinnedBases& pb, uint32_t lg, std::vector<int>* devs) {
  if (!pb.shards || lg < 2 || ((size_t)1 << lg) < pb.shard_ntt_min) return false;
  if (sharded_devices(pb.shards, devs)) return false;
  const size_t G = devs->size(); uint32_t lg_g = 0; while (((size_t)1 << lg_g) < G) ++lg_g;
  return G >= 1 && !(G & (G - 1)) && lg / 2 >= lg_g;
}
static int32_t p_ntt(Ctx* c, const PinnedBases& pb, void* data, uint32_t lg, size_t batch, int32_t direction, int32_t type, hipStream_t s) {
  std::vector<int> devs;
  if (!ntt_routed(pb, lg, &devs)) return ntt_run(c, data, lg, batch, ALEO_NTT_ORDER_NN, direction, type, s);
  for (size_t b = 0; b < batch; ++b) { const int32_t rc = ntt_sharded_device(c, (char*)data + (b << lg) * 32, lg, direction, type, devs.data(), devs.size(), s); if (rc) return rc; }
  return ALEO_MI355X_OK;
}
static int32_t p_ntt_from(Ctx* c, const PinnedBases& pb, void* out, const void* src, size_t src_stride, size_t src_len, uint32_t lg, size_t batch, hipStream_t s) {
  std::vector<int> devs;
  if (!ntt_routed(pb, lg, &devs)) return ntt_run_from(c, out, src, src_stride, src_len, lg, batch, ALEO_NTT_FORWARD, ALEO_NTT_STANDARD, s);
  const size_t n = (size_t)1 << lg;
  for (size_t b = 0; b < batch; ++b) {                        // pad by hand (sizes where a fill and a copy are noise), then the sharded transform in place
    char* o = (char*)out + b * n * 32;
    if (src_len < n) HIPCHK(hipMemsetAsync(o + src_len * 32, 0, (n - src_len) * 32, s));
    if (src_len) HIPCHK(hipMemcpyAsync(o, (const char*)src + b * src_stride * 32, src_len * 32, hipMemcpyDeviceToDevice, s));
    const int32_t rc = ntt_sharded_device(c, o, lg, ALEO_NTT_FORWARD, ALEO_NTT_STANDARD, devs.data(), devs.size(), s); if (rc) return rc;
  }
  return ALEO_MI355X_OK;
}


// `behind`: kernels of the NEXT round that need no challenge of this one — queued behind the commitment's last kernel, so that they run while the host finishes
// the MSM's tail, compresses, hashes and derives the challenge (Ctx::tail_hook; a request that takes several launch chains runs them afterwards instead).
static int32_t commit(Ctx* c, const PinnedBases& pb, const std::vector<MsmSeg>& segs, uint32_t k, uint8_t* out104, hipStream_t s, bool sparse = false,
                      std::function<int32_t()> behind = nullptr) {
  std::vector<uint64_t> jac(18 * (size_t)k);
  MsmJob j; j.segs = segs.data(); j.nseg = (uint32_t)segs.size(); j.k = k; j.mont = true; j.sparse = sparse; j.lean = true;      // lean: no phase-timing events between the chain's kernels
  const double t0 = now_ms();
  size_t points = 0; for (const MsmSeg& g : segs) points += g.len;
  if (pb.shards && points >= pb.shard_min) {
    // a sharded copy of the committer key is attached (row e2: a proof that spans devices): the coefficient vectors stay here, every device pulls its
    // slices and runs its own Pippenger, 144 bytes per result and shard come back.  `s` is drained first (the scalars are complete), so the kernels of `behind` — which
    // only read what the commitment reads and write fresh arena blocks — are queued at once and run on this device beside its own shard.
    HIPCHK(hipStreamSynchronize(s));
    if (behind) RC(behind());
    RC(commit_sharded(c, pb.shards, segs.data(), (uint32_t)segs.size(), k, true, jac.data(), s, false));
    jacobian_rows_to_affine104(out104, jac.data(), k);
    g_varuna_timing[6] += now_ms() - t0;
    return ALEO_MI355X_OK;
  }
  {
    struct Clear { Ctx* c; ~Clear() { c->tail_hook = nullptr; } } clear{c};      // whatever happens below, no hook (it captures this proof's state) outlives the call
    c->tail_hook = std::move(behind);
    const int32_t rc = msm_batch(c, jac.data(), pb, j, s);
    std::function<int32_t()> left = std::move(c->tail_hook); c->tail_hook = nullptr;
    if (rc) return rc;
    if (left) RC(left());
  }
  jacobian_rows_to_affine104(out104, jac.data(), k);
  g_varuna_timing[6] += now_ms() - t0; g_varuna_timing[7] += c->last_msm.host;      // time inside the commitment calls / their host tails (last chain of each call)
  return ALEO_MI355X_OK;
}

// ---- the index of a circuit, built once per proving key ---------------------------------------------------------------------------------
// [UPSTREAM-RECALL: varuna/ahp/indexer — AHPForR1CS::index: matrix arithmetisation over the non-zero domain, index commitments; reached from
// Process::synthesize_key, /root/reference/wasm/src/programs/manager/mod.rs:164-177, rust/src/program/deploy.rs:142,151.]
struct VarunaIndexOwner {
  aleo_mi355x_varuna_index view{};
  std::vector<uint32_t> positions; std::vector<uint8_t> vk, vk_aff;
  std::vector<void*> dev;                                  // every device allocation the index keeps
  std::vector<void*> tmp;                                  // scratch of the build (raw columns, C's forward arrays, cursors): freed when the build's stream has drained
  std::shared_ptr<PinnedOwner> key;                        // the committer key stays pinned while the index lives
  ~VarunaIndexOwner() { free_tmp(); for (void* p : dev) if (p) (void)hipFree(p); }
  void free_tmp() { for (void* p : tmp) if (p) (void)hipFree(p); tmp.clear(); }
  int32_t alloc(void** out, size_t bytes, bool scratch = false) { void* p = nullptr; HIPCHK(hipMalloc(&p, bytes ? bytes : 32)); (scratch ? tmp : dev).push_back(p); *out = p; return ALEO_MI355X_OK; }
};
void varuna_index_delete(VarunaIndexOwner* o) { delete o; }
const aleo_mi355x_varuna_index* varuna_index_view(const VarunaIndexOwner* o) { return &o->view; }
const std::vector<uint8_t>& varuna_index_vk(const VarunaIndexOwner* o) { return o->vk; }

static uint64_t pow2_at_least(uint64_t v, uint64_t lo) { uint64_t p = lo; while (p < v) p <<= 1; return p; }

int32_t varuna_index_build(Ctx* c, const PinnedBases& pb, std::shared_ptr<PinnedOwner> key, uint64_t key_handle, uint64_t max_degree, uint64_t gamma_offset,
                           uint64_t lagrange_offset, const aleo_mi355x_r1cs_matrix* abc, size_t n_constraints, size_t n_public, size_t n_private, uint32_t domain_flags, VarunaIndexOwner** out) {
  std::unique_ptr<VarunaIndexOwner> o(new VarunaIndexOwner()); o->key = std::move(key);
  hipStream_t s = c->stream;
  if (!n_constraints || !n_public || n_constraints >= (1ull << 28)) { g_last_error = "varuna_index: bad sizes"; return ALEO_MI355X_ERR_BAD_ARG; }
  const uint64_t n_vars = n_public + n_private, n_x = pow2_at_least(n_public, 1);
  uint64_t n_h = pow2_at_least(n_constraints, 2); n_h = pow2_at_least(n_x + n_private, n_h); n_h = pow2_at_least(2 * n_x, n_h);
  uint64_t nnz[3], nnz_max = 0, nnz_sum = 0, max_row[3] = {1, 1, 1};      // max_row: the longest row of A, B and of the stacked transpose (hints for the sparse products; >= 1 = known)
  std::vector<uint32_t> col_count(n_vars, 0);
  for (int m = 0; m < 3; ++m) {
    if (!abc[m].row_ptr || abc[m].row_ptr[0] != 0) { g_last_error = "varuna_index: row_ptr must start at 0"; return ALEO_MI355X_ERR_BAD_ARG; }
    nnz[m] = abc[m].row_ptr[n_constraints]; nnz_max = nnz[m] > nnz_max ? nnz[m] : nnz_max; nnz_sum += nnz[m];
    if (nnz[m] && (!abc[m].col || !abc[m].val)) { g_last_error = "varuna_index: null matrix arrays"; return ALEO_MI355X_ERR_BAD_ARG; }
    for (uint64_t e = 0; e < nnz[m]; ++e) { if (abc[m].col[e] >= n_vars) { g_last_error = "varuna_index: column outside the variables"; return ALEO_MI355X_ERR_BAD_ARG; } ++col_count[abc[m].col[e]]; }
    for (size_t r = 0; r < n_constraints; ++r) {
      if (abc[m].row_ptr[r + 1] < abc[m].row_ptr[r]) { g_last_error = "varuna_index: row_ptr not monotone"; return ALEO_MI355X_ERR_BAD_ARG; }
      const uint64_t len = abc[m].row_ptr[r + 1] - abc[m].row_ptr[r]; if (m < 2 && len > max_row[m]) max_row[m] = len;
    }
  }
  for (uint32_t cnt : col_count) if (cnt > max_row[2]) max_row[2] = cnt;      // a row of the stacked transpose = every use of one variable in A, B and C
  uint64_t nk[3], ko[3], k_sum = 0, n_k = 0;                 // one non-zero domain per matrix; ko: elements of the earlier matrices
  for (int m = 0; m < 3; ++m) { nk[m] = pow2_at_least(nnz[m], 2); n_k = nk[m] > n_k ? nk[m] : n_k; }
  if (domain_flags == 2 || (domain_flags == 0 && n_k < (1ull << 18))) nk[0] = nk[1] = nk[2] = n_k;      // shared: latency-bound sizes (header)
  for (int m = 0; m < 3; ++m) { ko[m] = k_sum; k_sum += nk[m]; }
  (void)nnz_max;
  if (3 * n_h > max_degree + 1 || n_k > max_degree + 1 || max_degree + 1 > pb.n || gamma_offset + 3 > pb.n) { g_last_error = "varuna_index: committer key too small for this circuit"; return ALEO_MI355X_ERR_BAD_ARG; }
  // variable -> position on H: public i -> i |H|/|X|, the j-th private one -> the j-th element of H \ X
  const uint64_t ratio = n_h / n_x;
  o->positions.resize(n_vars);
  for (uint64_t v = 0; v < n_vars; ++v) { if (v < n_public) o->positions[v] = (uint32_t)(v * ratio); else { const uint64_t j = v - n_public; o->positions[v] = (uint32_t)(j + j / (ratio - 1) + 1); } }
  HFr r2; std::memcpy(r2.l, host::HParams<4>::R2, 32);
  const HFr one = HFr::one();
  aleo_mi355x_varuna_index& V = o->view;
  for (int m = 0; m < 3; ++m) V.max_row[m] = max_row[m];
  V.n_h = n_h; V.n_k_a = nk[0]; V.n_k_b = nk[1]; V.n_k_c = nk[2]; V.n_x = n_x; V.n_public = n_public; V.n_vars = n_vars; V.committer_key = key_handle; V.max_degree = max_degree; V.gamma_offset = gamma_offset; V.lagrange_offset = lagrange_offset;
  if (lagrange_offset && lagrange_offset + n_h + 1 > pb.n) { g_last_error = "varuna_index: the Lagrange powers do not fit the committer key"; return ALEO_MI355X_ERR_BAD_ARG; }
  const bool tim = std::getenv("ALEO_MI355X_INDEX_TIMING") != nullptr; double t_prev = now_ms();
  auto mark = [&](const char* what) { if (tim) { (void)hipStreamSynchronize(s); const double t = now_ms(); fprintf(stderr, "index_build %-28s %8.2f ms\n", what, t - t_prev); t_prev = t; } };
  // Index arithmetic on integers, on the device since round 3 (the host loops over the non-zeros were half of a 2^20-constraint key synthesis): per
  // matrix the rows expand into (row, column position on H) pairs and count their columns; one scan turns the counts into the transpose's row
  // pointers; a second pass drops every entry into its column's range (an atomic cursor per column: the order inside a column is whatever the
  // hardware makes it, the products M^T v are exact field sums, so every proof byte is independent of it).
  std::vector<uint32_t> rp(n_h + 1);
  auto up = [&](void** dst, const void* src, size_t bytes, bool scratch = false) -> int32_t { RC(o->alloc(dst, bytes, scratch)); if (bytes) HIPCHK(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, s)); return ALEO_MI355X_OK; };
  auto to_mont = [&](void* p, size_t n) -> int32_t { return fr_lin(c, p, n, nullptr, r2.l, p, nullptr, nullptr, s); };
  void *dpos, *kid, *kv, *dtp, *dcur, *dtcol, *dtval;
  RC(up(&dpos, o->positions.data(), n_vars * 4)); V.positions_device = dpos;
  RC(o->alloc(&kid, 2 * k_sum * 4)); RC(o->alloc(&kv, k_sum * 32, true)); RC(o->alloc(&dtp, (n_h + 1) * 4)); RC(o->alloc(&dtcol, nnz_sum * 4)); RC(o->alloc(&dtval, nnz_sum * 32));
  HIPCHK(hipMemsetAsync(kid, 0, 2 * k_sum * 4, s)); HIPCHK(hipMemsetAsync(kv, 0, k_sum * 32, s)); HIPCHK(hipMemsetAsync(dtp, 0, (n_h + 1) * 4, s));
  void *drp[3], *dcolraw[3], *dcol[3], *dval[3];
  for (int m = 0; m < 3; ++m) {                            // forward matrices with columns on H, rows padded to |H| (C only feeds the transpose and the arithmetisation)
    for (uint64_t i = 0; i <= n_h; ++i) rp[i] = i <= n_constraints ? abc[m].row_ptr[i] : (uint32_t)nnz[m];
    const bool fwd = m < 2;                                // A and B are kept as forward CSR (z_a, z_b); C's arrays, the raw columns and the cursors are scratch
    RC(up(&drp[m], rp.data(), (n_h + 1) * 4, !fwd)); RC(up(&dcolraw[m], abc[m].col, nnz[m] * 4, true)); RC(up(&dval[m], abc[m].val, nnz[m] * 32, !fwd)); RC(o->alloc(&dcol[m], nnz[m] * 4, !fwd));
    HIPCHK(hipStreamSynchronize(s));                       // rp is reused by the next matrix
    RC(index_expand_rows(c, (const uint32_t*)drp[m], (const uint32_t*)dcolraw[m], (const uint32_t*)dpos, n_constraints, (uint32_t*)kid + 2 * ko[m], (uint32_t*)kid + 2 * ko[m] + nk[m],
                         (uint32_t*)dcol[m], (uint32_t*)dtp, s));
    if (nnz[m]) HIPCHK(hipMemcpyAsync((char*)kv + ko[m] * 32, dval[m], nnz[m] * 32, hipMemcpyDeviceToDevice, s));      // canonical values: converted with the whole array below
  }
  RC(index_scan_inclusive(c, (uint32_t*)dtp, n_h + 1, s));
  RC(o->alloc(&dcur, (n_h + 1) * 4, true)); HIPCHK(hipMemcpyAsync(dcur, dtp, (n_h + 1) * 4, hipMemcpyDeviceToDevice, s));
  for (int m = 0; m < 3; ++m) RC(index_transpose_rows(c, (const uint32_t*)drp[m], (const uint32_t*)dcol[m], dval[m], n_constraints, (uint32_t)(m * n_h), (uint32_t*)dcur, (uint32_t*)dtcol, dtval, s));
  for (int m = 0; m < 2; ++m) RC(to_mont(dval[m], nnz[m]));
  V.a_row_ptr = drp[0]; V.a_col = dcol[0]; V.a_val = dval[0]; V.b_row_ptr = drp[1]; V.b_col = dcol[1]; V.b_val = dval[1];
  RC(to_mont(dtval, nnz_sum)); V.t_row_ptr = dtp; V.t_col = dtcol; V.t_val = dtval;
  mark("index arithmetic (device)");
  // 1 / v_X on H \ X (v_X(w^p) = wx^p − 1, wx = w^|X|; zeros stay zero through the batch inversion), elements of H
  void *vx, *he;
  RC(o->alloc(&vx, n_h * 32)); RC(o->alloc(&he, n_h * 32, true));
  const HFr gen_h = domain_gen(n_h), wx = HFr::pow_u64(gen_h, n_x), neg1 = HFr::neg(one);
  RC(fr_powers(c, vx, n_h, one.l, wx.l, s)); RC(fr_lin(c, vx, n_h, neg1.l, one.l, vx, nullptr, nullptr, s)); RC(fr_batch_inverse(c, vx, n_h, s));
  RC(fr_powers(c, he, n_h, one.l, gen_h.l, s));
  V.vx_inv = vx;
  mark("vx, H elements");
  // arithmetisation over K: row, col, val = M[r,c] col / |H|, row_col — padding: row = col = 1 (position 0), val = 0
  void *kev, *kpo, *k2;
  RC(o->alloc(&kev, 4 * k_sum * 32)); RC(o->alloc(&kpo, 4 * k_sum * 32)); RC(o->alloc(&k2, 8 * k_sum * 32));
  RC(to_mont(kv, k_sum));
  mark("alloc + upload K arrays");
  uint32_t lg_nh = 0; while ((1ull << lg_nh) < n_h) ++lg_nh;
  const HFr nh_inv = inv_pow2(lg_nh);
  HIPCHK(hipMemsetAsync(k2, 0, 8 * k_sum * 32, s));
  for (int m = 0; m < 3; ++m) {
    const uint64_t n = nk[m]; uint32_t lg = 0; while ((1ull << lg) < n) ++lg;
    char* e = (char*)kev + 4 * ko[m] * 32; const uint32_t* ri = (const uint32_t*)kid + 2 * ko[m]; const uint32_t* ci = ri + n;
    RC(fr_gather_mul(c, e, n, nullptr, he, ri, nullptr, nullptr, s));
    RC(fr_gather_mul(c, e + n * 32, n, nullptr, he, ci, nullptr, nullptr, s));
    RC(fr_vec_op(c, e + 2 * n * 32, (char*)kv + ko[m] * 32, e + n * 32, n, 0, s));
    RC(fr_lin(c, e + 2 * n * 32, n, nullptr, nh_inv.l, e + 2 * n * 32, nullptr, nullptr, s));
    RC(fr_vec_op(c, e + 3 * n * 32, e, e + n * 32, n, 0, s));
    char* po = (char*)kpo + 4 * ko[m] * 32; char* e2 = (char*)k2 + 8 * ko[m] * 32;
    HIPCHK(hipMemcpyAsync(po, e, 4 * n * 32, hipMemcpyDeviceToDevice, s));
    RC(p_ntt(c, pb, po, lg, 4, 1, 0, s));
    for (int j = 0; j < 4; ++j) HIPCHK(hipMemcpyAsync(e2 + (size_t)j * 2 * n * 32, po + (size_t)j * n * 32, n * 32, hipMemcpyDeviceToDevice, s));
    RC(p_ntt(c, pb, e2, lg + 1, 4, 0, 0, s));
  }
  mark("arithmetisation + transforms");
  V.k_evals = kev; V.k_idx = kid; V.k_polys = kpo; V.k2_evals = k2; V.positions = o->positions.data();
  // index commitments -> what the transcript absorbs first
  o->vk_aff.assign(12 * 104, 0); uint8_t* aff = o->vk_aff.data();
  {
    std::vector<MsmSeg> sg(12);
    for (int q = 0; q < 12; ++q) { const int m = q / 4, j = q % 4; sg[q].d_ptr = (char*)kpo + (4 * ko[m] + (size_t)j * nk[m]) * 32; sg[q].len = nk[m]; sg[q].off = 0; sg[q].out = (uint32_t)q; }
    RC(commit(c, pb, sg, 12, aff, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  o->free_tmp();                                           // nothing queued reads the scratch any more
  mark("12 commitments");
  o->vk.resize(12 * 48 + 40);
  RC(aleo_mi355x_g1_compress(o->vk.data(), aff, 12));
  const uint64_t dims[5] = {n_h, nk[0], nk[1], nk[2], n_x}; std::memcpy(&o->vk[12 * 48], dims, 40);
  V.vk_bytes = o->vk.data(); V.vk_len = o->vk.size(); V.vk_affine = o->vk_aff.data();
  *out = o.release();
  return ALEO_MI355X_OK;
}


// The state of one proof between the rounds (upstream: varuna::ahp::prover::State) and the round functions in the order upstream calls them.
// A proof covers m circuits (`keys_to_constraints: BTreeMap<&ProvingKey, &[Assignment]>`), each with its own instances: `Shared` is what they share —
// the transcript and every challenge, the mask and g_1, h_1 over the largest constraint domain H*, h_2 over the largest non-zero domain K*, the two
// openings — and one `Prover` per circuit holds that circuit's polynomials.  Circuit j enters the first sumcheck behind the selector
// s_j = v_{H*} / v_{H_j} = sum_t X^(t |H_j|): its quotient adds into h_1 as it is, its remainder block tiles over H* (fr_add_tiled); see
// oracle/varuna_ref.py prove_batch for the algebra.  With one circuit nothing is added or tiled: the circuit writes the shared buffers directly.
#define TAKE_M(var, elems) var = sh.ar.take(elems); if (!var) { g_last_error = "varuna_prove: workspace accounting"; return ALEO_MI355X_ERR_HIP; }
#define TAKE_S(var, elems) char* var = sh.ar.take(elems); if (!var) { g_last_error = "varuna_prove: workspace accounting"; return ALEO_MI355X_ERR_HIP; }
static constexpr size_t HC = 3;                            // coefficients of a hiding polynomial (hiding bound 1)
static constexpr size_t MAX_TOTAL_INSTANCES = 32, MAX_CIRCUITS = MAX_TOTAL_INSTANCES, MAX_INSTANCES = MAX_TOTAL_INSTANCES;      // one proof covers one transaction: at most 32 transitions, in any split over circuits
static constexpr size_t PIN_SUMS = 8192, PIN_FLAG = 9216, PIN_SMALL_BYTES = 12288;      // small read-backs behind the staging area: evaluations / sigma values from 0 (<= 129 x 32 bytes), the circuits' sums over H, the canonical-input flag

static int32_t lincomb_any(Ctx* c, char* dst, size_t n, const HFr& c0, std::vector<const void*>& terms, std::vector<size_t>& lens, std::vector<HFr>& co, hipStream_t s);
struct Shared {
  Ctx* c; const PinnedBases& pb; Seed32 seed;
  Shared(Ctx* c_, const PinnedBases& pb_, const uint8_t* seed32) : c(c_), pb(pb_) { std::memcpy(seed.w, seed32, 32); }
  size_t m = 0, K = 0, N = 0, n_kmax = 0, lead = 0, x_total = 0; uint64_t D = 0, gamma_offset = 0;
  hipStream_t s = nullptr; double t_mark[7] = {}; Arena ar{nullptr, 0, 0}; char* pin = nullptr; char* stage = nullptr; char* pin_small = nullptr; char* pin_small_dev = nullptr;      // pin_small_dev: the device's address of pin_small (kernels store small read-backs there)
  HFr one, neg1, r2; host::FiatShamir fs; uint64_t lay_mask = 0, lay_blind = 0, lay_blind_mask = 0;
  char *mask = nullptr, *bl = nullptr, *h1 = nullptr, *g1 = nullptr, *h2 = nullptr, *flag = nullptr, *evd = nullptr;
  std::vector<HFr> blind, comb, evals, x_mont, ch_b, ch_g; std::vector<uint8_t> wit_aff, aff3;
  uint8_t aff2[208], aff4[104], aff5[208];
  HFr alpha, eta_b, eta_c, beta, gamma, random_v;
  // staging offsets (elements of 32 bytes inside `stage`): x̂ coefficients | hiding polynomials | the opening's hiding quotient | rho
  size_t st_blind() const { return x_total; }
  size_t st_blq() const { return x_total + (3 * K + 1) * HC; }
  size_t st_rho() const { return st_blq() + HC; }
};

struct Prover {                                            // one circuit of the proof
  Shared& sh; const aleo_mi355x_varuna_index& ix; const size_t j, k, q0;      // circuit number, its instances, the number of its first instance in the proof
  Prover(Shared& sh_, const aleo_mi355x_varuna_index& ix_, size_t j_, size_t k_, size_t q0_) : sh(sh_), ix(ix_), j(j_), k(k_), q0(q0_) {}
  size_t n_h = 0, n_x = 0, L = 0, n4 = 0, nk[3] = {}, ko[3] = {}, k_sum = 0, n_k = 0, x_off = 0, pin_off = 0; uint32_t lg_h = 0, lg_km[3] = {};
  char *xp = nullptr, *wit = nullptr, *ext = nullptr, *hq = nullptr, *rq = nullptr, *f = nullptr, *evals_h = nullptr, *rho_dev = nullptr;
  char *r4_terms[3] = {}; size_t r4_lens[3] = {}; char *E = nullptr, *F = nullptr;
  std::vector<std::vector<HFr>> x_poly;
  HFr vh_alpha, vh_beta, vv, sigma[3], delta[3];
  size_t run0[3] = {}, runc[3] = {}, nrun = 0;
  bool lead() const { return sh.lead == j; }
  bool lagrange() const { return ix.lagrange_offset != 0; }

  int32_t setup();
  size_t workspace_elems() const { return n_h * (41 + 24 * k) + k_sum * 6 + n_k * 4 + 4096; }
  int32_t first_round(const void* const* assignments, std::vector<MsmSeg>& sg);      // AHPForR1CS::prover_first_round for this circuit's instances
  int32_t second_round_early();                            // its challenge-free part (operands of the sumcheck on 4|H|): queued behind round 1's commitments
  int32_t second_round();                                  // prover_second_round: t, this circuit's summand of the first sumcheck, its quotient and remainder
  int32_t third_round();                                   // prover_third_round: f_M (sigma_M, g_M follow the read-back)
  int32_t fourth_round_early();                            // its challenge-free part (f_M on the domains of size 2|K_M|): queued behind round 3's commitments
  int32_t fourth_round();                                  // prover_fourth_round: the quotients h_M of this circuit, delta-weighted, run by run
};

int32_t Prover::setup() {
  n_h = ix.n_h; n_x = ix.n_x; L = n_h + 1; n4 = 4 * n_h;
  nk[0] = ix.n_k_a; nk[1] = ix.n_k_b; nk[2] = ix.n_k_c; ko[0] = 0; ko[1] = nk[0]; ko[2] = nk[0] + nk[1]; k_sum = nk[0] + nk[1] + nk[2];
  n_k = nk[0] > nk[1] ? (nk[0] > nk[2] ? nk[0] : nk[2]) : (nk[1] > nk[2] ? nk[1] : nk[2]);      // the largest non-zero domain of this circuit
  const uint64_t D = ix.max_degree; const PinnedBases& pb = sh.pb;
  bool k_ok = true; for (int m = 0; m < 3; ++m) k_ok = k_ok && nk[m] >= 2 && !(nk[m] & (nk[m] - 1));
  if (k < 1 || k > MAX_INSTANCES || n_h < 2 || !k_ok || n_x < 1 || n_h < 2 * n_x || (n_h & (n_h - 1)) || (n_x & (n_x - 1)) ||
      ix.n_public > n_x || ix.n_vars > n_h || ix.gamma_offset + HC > pb.n || (ix.lagrange_offset && ix.lagrange_offset + n_h + 1 > pb.n) || D + 1 > pb.n || 3 * n_h > D + 1 || n_k > D + 1) {
    g_last_error = "varuna_prove: inconsistent index / key sizes"; return ALEO_MI355X_ERR_BAD_ARG;
  }
  if (!ix.a_row_ptr || !ix.a_col || !ix.a_val || !ix.b_row_ptr || !ix.b_col || !ix.b_val || !ix.t_row_ptr || !ix.t_col || !ix.t_val || !ix.vx_inv || !ix.k_evals || !ix.k_idx ||
      !ix.k_polys || !ix.k2_evals || !ix.positions || !ix.vk_bytes || ix.vk_len != 12 * 48 + 40) {
    g_last_error = "varuna_prove: the index struct has a null array (or vk_len != 616)"; return ALEO_MI355X_ERR_BAD_ARG;
  }
  lg_h = 0; lg_km[0] = lg_km[1] = lg_km[2] = 0; while ((1ull << lg_h) < n_h) ++lg_h;
  for (int m = 0; m < 3; ++m) while ((1ull << lg_km[m]) < nk[m]) ++lg_km[m];
  return ALEO_MI355X_OK;
}

int32_t Prover::first_round(const void* const* assignments, std::vector<MsmSeg>& sg) {
  Ctx* c = sh.c; hipStream_t s = sh.s; Arena& ar = sh.ar; char* pin = sh.pin + pin_off * 32; char* stage = sh.stage;
  TAKE(zH, k * n_h + 8) TAKE(ev, 3 * k * n_h) TAKE(xh, k * n_h) TAKE_M(xp, k * n_x) TAKE_M(wit, 3 * k * L)      // (+ 8: the canonical-input flag sits behind z on H, cleared by the same fill)
  x_poly.assign(k, {});
  const size_t xb0 = sh.x_mont.size(); sh.x_mont.resize(xb0 + k * n_x, HFr::zero());          // the padded public inputs: what the transcript absorbs per instance
  uint32_t lg_x = 0; while ((1ull << lg_x) < n_x) ++lg_x;
  const HFr one = sh.one, gx_inv = HFr::inv(domain_gen(n_x)), nx_inv = inv_pow2(lg_x);
  const uint32_t* pos = (const uint32_t*)ix.positions;
  const bool host_layout = ix.positions_device == nullptr;      // without the positions in HBM the host lays the assignment out on H (pinned staging)
  if (host_layout) std::memset(pin, 0, k * n_h * 32);
  else for (size_t v = 0; v < ix.n_vars; ++v) if (pos[v] >= n_h) { g_last_error = "varuna_prove: variable position outside H"; return ALEO_MI355X_ERR_BAD_ARG; }
  for (size_t i = 0; i < k; ++i) {
    const uint8_t* z = (const uint8_t*)assignments[i];
    if (host_layout)
      for (size_t v = 0; v < ix.n_vars; ++v) {
        if (pos[v] >= n_h) { g_last_error = "varuna_prove: variable position outside H"; return ALEO_MI355X_ERR_BAD_ARG; }
        if (HFr::geq_p((const uint64_t*)(z + v * 32))) { g_last_error = "varuna_prove: assignment not canonical"; return ALEO_MI355X_ERR_BAD_ARG; }
        std::memcpy(pin + (i * n_h + pos[v]) * 32, z + v * 32, 32);
      }
    std::vector<HFr> xe(n_x, HFr::zero());
    for (size_t t = 0; t < ix.n_public; ++t) { HFr v; std::memcpy(v.l, z + t * 32, 32); if (HFr::geq_p(v.l)) { g_last_error = "varuna_prove: assignment not canonical"; return ALEO_MI355X_ERR_BAD_ARG; } xe[t] = HFr::to_mont(v); sh.x_mont[xb0 + i * n_x + t] = xe[t]; }
    x_poly[i] = xe;                                        // inverse DFT over X on the host: |X| is the (padded) number of public inputs
    host_ntt(x_poly[i], gx_inv);
    for (auto& v : x_poly[i]) v = HFr::mul(v, nx_inv);
  }
  for (size_t i = 0; i < k; ++i) std::memcpy(stage + (x_off + i * n_x) * 32, x_poly[i].data(), n_x * 32);
  if (host_layout) {
    HIPCHK(hipMemcpyAsync(zH, pin, k * n_h * 32, hipMemcpyHostToDevice, s));
    RC(fr_lin(c, zH, k * n_h, nullptr, sh.r2.l, zH, nullptr, nullptr, s));                // canonical -> Montgomery
  } else {                                                                                  // upload in variable order; scatter + Montgomery form on the device
    TAKE(zraw, k * ix.n_vars)
    if (!sh.flag) sh.flag = zH + k * n_h * 32;                                               // raised by the scatter when an entry is not below r
    HIPCHK(hipMemsetAsync(zH, 0, (k * n_h + 8) * 32, s));
    for (size_t i = 0; i < k; ++i) {
      HIPCHK(hipMemcpyAsync(zraw + i * ix.n_vars * 32, assignments[i], ix.n_vars * 32, hipMemcpyHostToDevice, s));
      RC(fr_scatter_to_mont(c, zH + i * n_h * 32, zraw + i * ix.n_vars * 32, ix.positions_device, ix.n_vars, sh.flag, s));
    }
  }
  HIPCHK(hipMemcpyAsync(xp, stage + x_off * 32, k * n_x * 32, hipMemcpyHostToDevice, s));
  RC(p_ntt_from(c, sh.pb, xh, xp, n_x, n_x, lg_h, k, s));      // x̂ of every instance on H: |X| coefficients each, zero-padded by the transform's first pass
  for (size_t i = 0; i < k; ++i) {
    char* e0 = ev + 3 * i * n_h * 32; char* z_i = zH + i * n_h * 32; char* xh_i = xh + i * n_h * 32;
    RC(fr_spmv(c, e0 + n_h * 32, ix.a_row_ptr, ix.a_col, ix.a_val, z_i, n_h, s, ix.max_row[0]));
    RC(fr_spmv(c, e0 + 2 * n_h * 32, ix.b_row_ptr, ix.b_col, ix.b_val, z_i, n_h, s, ix.max_row[1]));
    RC(fr_sub_mul(c, e0, z_i, xh_i, ix.vx_inv, n_h, s));                                    // (z − x̂) / v_X off X, 0 on X
  }
  if (lagrange()) {                                        // KZG10::commit_lagrange for w, z_a, z_b: commit the evaluations (kept here) against L_i(tau) G
    evals_h = ar.take(3 * k * n_h); rho_dev = ar.take(3 * k);
    if (!evals_h || !rho_dev) { g_last_error = "varuna_prove: workspace accounting"; return ALEO_MI355X_ERR_HIP; }
    HIPCHK(hipMemcpyAsync(evals_h, ev, 3 * k * n_h * 32, hipMemcpyDeviceToDevice, s));
  }
  RC(p_ntt(c, sh.pb, ev, lg_h, 3 * k, 1, 0, s));
  {
    HFr rho[3 * MAX_INSTANCES];                                                             // rho_w, rho_a, rho_b of instance q / 3
    for (size_t q = 0; q < 3 * k; ++q) {
      rho[q] = random_fr(sh.seed, 3 * q0 + q);
      for (size_t t = 0; t < HC; ++t) sh.blind[(3 * q0 + q) * HC + t] = random_fr(sh.seed, sh.lay_blind + HC * (3 * q0 + q) + t);
    }
    for (size_t at = 0; at < 3 * k; at += 24) RC(fr_blind_rows(c, wit + at * L * 32, ev + at * n_h * 32, n_h, 3 * k - at < 24 ? 3 * k - at : 24, rho + at, s));      // + rho (X^|H| − 1), 24 polynomials per launch
    if (lagrange()) { char* st = stage + (sh.st_rho() + 3 * q0) * 32; std::memcpy(st, rho, 3 * k * 32); HIPCHK(hipMemcpyAsync(rho_dev, st, 3 * k * 32, hipMemcpyHostToDevice, s)); }
  }
  for (size_t q = 0; q < 3 * k; ++q) {
    MsmSeg a; a.out = (uint32_t)(3 * q0 + q);
    if (lagrange()) {                                                                       // sum_i evals_i L_i(tau) G + rho v_H(tau) G
      a.d_ptr = evals_h + q * n_h * 32; a.len = n_h; a.off = ix.lagrange_offset; sg.push_back(a);
      MsmSeg v; v.d_ptr = rho_dev + q * 32; v.len = 1; v.off = ix.lagrange_offset + n_h; v.out = a.out; sg.push_back(v);
    } else { a.d_ptr = wit + q * L * 32; a.len = L; a.off = 0; sg.push_back(a); }
    MsmSeg b; b.d_ptr = sh.bl + (3 * q0 + q) * HC * 32; b.len = HC; b.off = ix.gamma_offset; b.out = a.out; sg.push_back(b);
  }
  return ALEO_MI355X_OK;
}

int32_t Prover::second_round_early() {
  Ctx* c = sh.c; hipStream_t s = sh.s;
  TAKE_M(E, (2 + 3 * k) * n4)                                                               // rows 0, 1: r, t (second_round); then ẑ_i, z_a,i, z_b,i per instance
  RC(ahp_sumcheck_operands(c, E + 2 * n4 * 32, wit, xp, n_h, n_x, k, s));                    // ẑ_i = w_i (X^|X| − 1) + x̂_i, z_a,i, z_b,i — every row written in full
  return p_ntt(c, sh.pb, E + 2 * n4 * 32, lg_h + 2, 3 * k, 0, 0, s);
}

int32_t Prover::second_round() {
  Ctx* c = sh.c; hipStream_t s = sh.s; Arena& ar = sh.ar; const HFr &alpha = sh.alpha, &eta_b = sh.eta_b, &eta_c = sh.eta_c;
  vh_alpha = vanish(n_h, alpha);
  if (vh_alpha.is_zero()) { g_last_error = "varuna_prove: alpha landed in H"; return ALEO_MI355X_ERR_HIP; }
  TAKE_M(ext, 3 * n_h) TAKE(rt, 2 * n_h) TAKE(Q, n4)
  if (lead()) { hq = sh.h1; rq = sh.g1; } else { TAKE_M(hq, 2 * n_h) TAKE_M(rq, n_h) }
  {
    const HFr first = HFr::pow_u64(alpha, n_h - 1), ratio = HFr::inv(alpha);
    RC(fr_powers(c, rt, n_h, first.l, ratio.l, s));                                          // r(alpha, X) = sum_k alpha^(|H|-1-k) X^k
  }
  RC(p_ntt_from(c, sh.pb, ext, rt, n_h, n_h, lg_h, 1, s));    // v_H(alpha) / (alpha − h) on H: no inversion on the device
  { const HFr eta[2] = {eta_b, eta_c}; RC(fr_scale_rows(c, ext + n_h * 32, ext, n_h, 2, eta, s)); }      // the eta_b- and eta_c-scaled copies B^T and C^T multiply
  RC(fr_spmv(c, rt + n_h * 32, ix.t_row_ptr, ix.t_col, ix.t_val, ext, n_h, s, ix.max_row[2]));
  RC(p_ntt(c, sh.pb, rt + n_h * 32, lg_h, 1, 1, 0, s));                                       // t(X)
  RC(p_ntt_from(c, sh.pb, E, rt, n_h, n_h, lg_h + 2, 2, s));  // r, t on 4|H|: |H| coefficients each, zero-padded by the first pass (the operands of the instances are there already: second_round_early)
  for (size_t i = 0; i < k; ++i) {
    char* e_z = E + (2 + 3 * i) * n4 * 32;
    RC(ahp_first_sumcheck(c, e_z + n4 * 32, n4, E, e_z + n4 * 32, e_z + 2 * n4 * 32, E + n4 * 32, e_z, eta_b.l, eta_c.l, s));
  }
  char* q1 = E + 3 * n4 * 32;
  if (k > 1 || q0 != 0) {                                                 // sum_i c_i numerator_i (the proof's first instance has c = 1)
    std::vector<const void*> terms(k); std::vector<size_t> lens(k, n4); std::vector<HFr> co(k);
    for (size_t i = 0; i < k; ++i) { terms[i] = E + (3 + 3 * i) * n4 * 32; co[i] = sh.comb[q0 + i]; }
    RC(lincomb_any(c, Q, n4, HFr::zero(), terms, lens, co, s)); q1 = Q;                     // 29..32 instances of one circuit: more terms than one fr_lincomb launch takes
  }
  RC(p_ntt(c, sh.pb, q1, lg_h + 2, 1, 1, 0, s));
  // q (+ the mask, which rides with the largest domain) = h (X^|H| − 1) + X g, degree < 3|H|: quotient blocks p1 + p2 | p2, remainder p0 + p1 + p2; the remainder's
  // constant term — this circuit's sum over H — goes straight into pinned host memory (read with the commitments).  One launch (rounds 1-4: a copy, three vector ops, a read-back)
  RC(fr_split_quotient(c, hq, rq, q1, lead() ? sh.mask : nullptr, n_h, sh.pin_small_dev + PIN_SUMS + 32 * j, s));
  return ALEO_MI355X_OK;
}

int32_t Prover::third_round() {
  Ctx* c = sh.c; hipStream_t s = sh.s; Arena& ar = sh.ar; const HFr& beta = sh.beta;
  vh_beta = vanish(n_h, beta);
  if (vh_beta.is_zero()) { g_last_error = "varuna_prove: beta landed in H"; return ALEO_MI355X_ERR_HIP; }
  vv = HFr::mul(vh_alpha, vh_beta);
  TAKE_M(f, k_sum) TAKE(rb, n_h)                                                              // f_M at element ko[M], |K_M| values
  {
    const HFr first = HFr::pow_u64(beta, n_h - 1), ratio = HFr::inv(beta);
    RC(fr_powers(c, rb, n_h, first.l, ratio.l, s));
  }
  RC(p_ntt(c, sh.pb, rb, lg_h, 1, 0, 0, s));
  {                                                                                          // f_M = val u_H(alpha, row) u_H(beta, col) on K_M: two gathers; the three matrices in one launch
    void* dst[3]; size_t cnt[3]; const void* sc[3]; const void* i1[3]; const void* i2[3];
    for (size_t m = 0; m < 3; ++m) {
      const uint32_t* ri = (const uint32_t*)ix.k_idx + 2 * ko[m];
      dst[m] = f + ko[m] * 32; cnt[m] = nk[m]; sc[m] = (const char*)ix.k_evals + (4 * ko[m] + 2 * nk[m]) * 32; i1[m] = ri; i2[m] = ri + nk[m];
    }
    RC(fr_gather_mul3(c, dst, cnt, sc, ext, i1, rb, i2, 3, s));
  }
  // maximal runs of consecutive matrices with equal domains share batched transforms (and, in round 4, one numerator pass)
  nrun = 0;
  for (size_t m = 0; m < 3;) { size_t cnt = 1; while (m + cnt < 3 && nk[m + cnt] == nk[m]) ++cnt; run0[nrun] = m; runc[nrun++] = cnt; m += cnt; }
  for (size_t r = 0; r < nrun; ++r) RC(p_ntt(c, sh.pb, f + ko[run0[r]] * 32, lg_km[run0[r]], runc[r], 1, 0, s));
  { const void* src[3] = {f + ko[0] * 32, f + ko[1] * 32, f + ko[2] * 32}; RC(fr_pick(c, sh.pin_small_dev + 32 * (3 * j), src, 3, s)); }      // f_M(0): one launch into pinned host memory
  return ALEO_MI355X_OK;
}

int32_t Prover::fourth_round_early() {
  Ctx* c = sh.c; hipStream_t s = sh.s;
  TAKE_M(F, 2 * k_sum)                                                                        // f_M zero-padded to 2|K_M|, then its values there
  for (size_t r = 0; r < nrun; ++r)                          // the polynomials of a run are contiguous in f (|K| apart): zero-padded to 2|K| by the transform's first pass
    RC(p_ntt_from(c, sh.pb, F + 2 * ko[run0[r]] * 32, f + ko[run0[r]] * 32, nk[run0[r]], nk[run0[r]], lg_km[run0[r]] + 1, runc[r], s));
  return ALEO_MI355X_OK;
}

int32_t Prover::fourth_round() {
  Ctx* c = sh.c; hipStream_t s = sh.s; Arena& ar = sh.ar; const HFr &alpha = sh.alpha, &beta = sh.beta;
  TAKE(B, 2 * k_sum)                                                                          // per matrix on its own domain of size 2|K_M| (F: fourth_round_early)
  for (size_t r = 0; r < nrun; ++r) {
    const size_t m0 = run0[r], cnt = runc[r], n2 = 2 * nk[m0]; char* Br = B + 2 * ko[m0] * 32;
    HFr consts[7] = {HFr::zero(), HFr::zero(), HFr::zero(), HFr::mul(alpha, beta), HFr::neg(alpha), HFr::neg(beta), vv};
    const void* idx[3] = {nullptr, nullptr, nullptr}; const void* ff[3] = {nullptr, nullptr, nullptr};
    for (size_t t = 0; t < cnt; ++t) {
      const size_t m = m0 + t;
      idx[t] = (const char*)ix.k2_evals + 8 * ko[m] * 32; ff[t] = F + 2 * ko[m] * 32; consts[t] = delta[m];
    }
    RC(ahp_matrix_sumcheck(c, Br, n2, idx, n2, ff, consts, s));                                // sum over the run of delta_M (vv val_M − b_M f_M) = h (X^|K| − 1)
    RC(p_ntt(c, sh.pb, Br, lg_km[m0] + 1, 1, 1, 0, s));
    r4_terms[r] = Br + nk[m0] * 32; r4_lens[r] = nk[m0];                                     // its upper half
  }
  return ALEO_MI355X_OK;
}

// ---- the proof: rounds over all circuits, commitments and transcript in between -----------------------------------------------------------------------
// Every round is three steps: prepare (queue the round's kernels on the stream, list the commitments it needs as RoundJobs), the commitment(s)
// (run_commits below: ONE launch chain for the jobs of every proof that takes part — a single proof, or several independent proofs in lockstep,
// aleo_mi355x_varuna_prove_many), finish (absorb the commitments, squeeze the challenges).
struct RoundJob { std::vector<MsmSeg> segs; uint32_t k = 0; bool sparse = false; uint8_t* out = nullptr; };      // k results (104-byte affine) to `out`
struct Batch {
  Shared sh; std::vector<std::unique_ptr<Prover>> P;
  RoundJob job[2]; int njobs = 0; std::function<int32_t()> hook;      // this round's commitments; kernels to queue behind the last commitment chain
  size_t need_ws_bytes = 0, need_pin_bytes = 0, pin_elems = 0, stage_elems = 0;
  Batch(Ctx* c, const PinnedBases& pb, const uint8_t* seed32) : sh(c, pb, seed32) {}
  int32_t init_sponge();                                   // Varuna::init_sponge: protocol name, batch sizes, public inputs, index commitments
  int32_t setup(const aleo_mi355x_varuna_index* const* ixs, size_t m, const size_t* ks);      // checks + sizes (need_ws_bytes, need_pin_bytes)
  int32_t attach(char* ws, size_t ws_bytes, char* pin);    // the slices of the slot's device workspace and pinned staging this proof works in
  int32_t first_prepare(const void* const* assignments); int32_t first_finish();      // the 3K + 1 hiding commitments
  int32_t second_prepare(); int32_t second_finish();       // g_1, h_1
  int32_t third_prepare(); int32_t third_finish();         // sigma_{j,M}, g_{j,M}
  int32_t fourth_prepare(); int32_t fourth_finish();       // h_2
  int32_t open_evaluate();                                 // the evaluation kernels and their read-back (queued; the caller synchronises)
  int32_t open_prepare();                                  // evaluations into the transcript, the two linear combinations, both witness polynomials
  int32_t write(uint8_t* out, size_t* out_len);            // Proof::write_le
};

static void batch_inverse_vec(std::vector<HFr>& v) {        // Montgomery's trick on the host: one inversion for all (non-zero) values
  std::vector<HFr> pre(v.size()); HFr acc = HFr::one();
  for (size_t i = 0; i < v.size(); ++i) { pre[i] = acc; acc = HFr::mul(acc, v[i]); }
  acc = HFr::inv(acc);
  for (size_t i = v.size(); i-- > 0;) { const HFr t = HFr::mul(acc, pre[i]); acc = HFr::mul(acc, v[i]); v[i] = t; }
}
// dst (n values) = c0 at X^0 + sum of terms, any number of them: fr_lincomb takes 28 per launch, later launches carry dst along as a term
static int32_t lincomb_any(Ctx* c, char* dst, size_t n, const HFr& c0, std::vector<const void*>& terms, std::vector<size_t>& lens, std::vector<HFr>& co, hipStream_t s) {
  constexpr size_t LC = 28; size_t at = 0; bool first = true;
  do {
    const void* t[LC]; size_t l[LC]; HFr k[LC]; size_t nt = 0;
    if (!first) { t[nt] = dst; l[nt] = n; k[nt++] = HFr::one(); }
    while (nt < LC && at < terms.size()) { t[nt] = terms[at]; l[nt] = lens[at]; k[nt++] = co[at++]; }
    RC(fr_lincomb(c, dst, n, first ? c0.l : nullptr, t, l, k, nt, s));
    first = false;
  } while (at < terms.size());
  return ALEO_MI355X_OK;
}

int32_t Batch::setup(const aleo_mi355x_varuna_index* const* ixs, size_t m, const size_t* ks) {
  Ctx* c = sh.c;
  if (m < 1 || m > MAX_CIRCUITS) { g_last_error = "varuna_prove: 1..32 circuits per proof"; return ALEO_MI355X_ERR_BAD_ARG; }
  sh.m = m; sh.K = 0;
  for (size_t j = 0; j < m; ++j) {
    if (!ixs[j] || !ixs[j]->positions || !ixs[j]->vk_bytes) { g_last_error = "varuna_prove: null index"; return ALEO_MI355X_ERR_BAD_ARG; }
    if (ks[j] < 1 || ks[j] > MAX_INSTANCES) { g_last_error = "varuna_prove: 1..32 instances per circuit"; return ALEO_MI355X_ERR_BAD_ARG; }
    P.emplace_back(new Prover(sh, *ixs[j], j, ks[j], sh.K)); sh.K += ks[j];
    RC(P[j]->setup());
    if (ixs[j]->committer_key != ixs[0]->committer_key || ixs[j]->max_degree != ixs[0]->max_degree || ixs[j]->gamma_offset != ixs[0]->gamma_offset) {
      g_last_error = "varuna_prove: the circuits of one proof must share one committer key"; return ALEO_MI355X_ERR_BAD_ARG;
    }
  }
  if (sh.K > MAX_TOTAL_INSTANCES) { g_last_error = "varuna_prove: at most 32 instances per proof"; return ALEO_MI355X_ERR_BAD_ARG; }
  sh.D = ixs[0]->max_degree; sh.gamma_offset = ixs[0]->gamma_offset;
  sh.N = 0; sh.n_kmax = 0; sh.x_total = 0; size_t pin_elems = 0, elems = 0;
  for (size_t j = 0; j < m; ++j) {
    Prover& p = *P[j];
    if (p.n_h > sh.N) { sh.N = p.n_h; sh.lead = j; }                                        // the first circuit with the largest constraint domain carries mask, g_1, h_1
    if (p.n_k > sh.n_kmax) sh.n_kmax = p.n_k;
    p.x_off = sh.x_total; sh.x_total += p.k * p.n_x; p.pin_off = pin_elems; pin_elems += p.k * p.n_h; elems += p.workspace_elems();
  }
  sh.s = c->stream;
  sh.t_mark[0] = now_ms();
  // ---- workspace sizes (attach() places the proof in its slices) ---------------------------------------------------------------------
  elems += m > 1 ? 16 * sh.N + 4 * sh.n_kmax + 4096 : 0;                                  // the shared polynomials beside the per-circuit accounting (which already covers one circuit's)
  need_ws_bytes = elems * 32 + (64 << 10);
  this->pin_elems = pin_elems;
  stage_elems = sh.x_total + (3 * sh.K + 1) * HC + HC + 3 * sh.K;                          // x̂ coefficients, hiding polynomials, the opening's hiding quotient, rho: staged through pinned memory
  need_pin_bytes = (pin_elems + stage_elems) * 32 + PIN_SMALL_BYTES;
  sh.one = HFr::one(); sh.neg1 = HFr::neg(sh.one); std::memcpy(sh.r2.l, host::HParams<4>::R2, 32);
  // randomness layout (oracle/varuna_ref.py randomness_layout over the largest |H| and all instances)
  sh.lay_mask = 3 * sh.K; sh.lay_blind = 3 * sh.K + 3 * sh.N; sh.lay_blind_mask = sh.lay_blind + 3 * HC * sh.K;
  return ALEO_MI355X_OK;
}

int32_t Batch::attach(char* ws, size_t ws_bytes, char* pin) {
  sh.ar = Arena{ws, 0, ws_bytes};
  sh.pin = pin; sh.stage = sh.pin + pin_elems * 32; sh.pin_small = sh.stage + stage_elems * 32;      // PIN_SMALL_BYTES for small read-backs
  void* dp = nullptr; HIPCHK(hipHostGetDevicePointer(&dp, sh.pin_small, 0)); sh.pin_small_dev = (char*)dp;
  return ALEO_MI355X_OK;
}

// The slot's grow-only device workspace and pinned staging, sized for `ws_bytes` / `pin_bytes` (one proof, or the sum over the proofs of a lockstep call)
static int32_t reserve_prover_memory(Ctx* c, size_t ws_bytes, size_t pin_bytes) {
  RC(c->prover_ws.reserve(ws_bytes));
  if (c->prover_pin_cap < pin_bytes) {
    if (c->prover_pin) { HIPCHK(hipStreamSynchronize(c->stream)); (void)hipHostFree(c->prover_pin); c->prover_pin = nullptr; c->prover_pin_cap = 0; }
    HIPCHK(hipHostMalloc(&c->prover_pin, pin_bytes + pin_bytes / 8, hipHostMallocMapped)); c->prover_pin_cap = pin_bytes + pin_bytes / 8;      // mapped: kernels store small read-backs into it
  }
  return ALEO_MI355X_OK;
}

// Varuna::init_sponge [UPSTREAM-RECALL]: the protocol name; per circuit its batch size (u64 LE as bytes) and the padded public inputs of each of its
// instances (one non-native absorb per instance); then every circuit's twelve index commitments.
int32_t Batch::init_sponge() {
  static const uint8_t NAME[] = "VARUNA-2023";
  sh.fs.absorb_bytes(NAME, sizeof NAME - 1);
  for (auto& p : P) {
    const uint64_t k64 = p->k; uint8_t kb[8]; for (int i = 0; i < 8; ++i) kb[i] = (uint8_t)(k64 >> (8 * i));
    sh.fs.absorb_bytes(kb, 8);
    for (size_t i = 0; i < p->k; ++i) sh.fs.absorb_fr(&sh.x_mont[p->x_off + i * p->n_x], p->n_x);
  }
  for (auto& p : P) {
    if (p->ix.vk_affine) { sh.fs.absorb_g1((const uint8_t*)p->ix.vk_affine, 104, 12); continue; }
    uint8_t aff[12 * 104];                                  // an index struct without the affine form: decompress (twelve square roots, ~ 0.5 ms)
    RC(aleo_mi355x_g1_decompress(aff, p->ix.vk_bytes, 12, 0));
    sh.fs.absorb_g1(aff, 104, 12);
  }
  return ALEO_MI355X_OK;
}

int32_t Batch::first_prepare(const void* const* assignments) {
  Ctx* c = sh.c; hipStream_t s = sh.s; const size_t K = sh.K, N = sh.N;
  TAKE_M(sh.bl, (3 * K + 1) * HC) TAKE_M(sh.mask, 3 * N)
  sh.blind.assign((3 * K + 1) * HC, HFr::zero()); sh.x_mont.clear();
  std::vector<MsmSeg>& sg = job[0].segs; std::vector<MsmSeg>& sm = job[1].segs; sg.clear(); sm.clear();
  // the mask polynomial needs nothing from the assignments: queued FIRST, it runs while the host stages and uploads them (a pageable 1-MB copy keeps the calling thread ~50 us)
  RC(fr_random(c, sh.mask, 3 * N, (const uint8_t*)sh.seed.w, sh.lay_mask, 1, s));
  RC(fr_lin(c, sh.mask, 1, nullptr, sh.neg1.l, sh.mask + N * 32, sh.neg1.l, sh.mask + 2 * N * 32, s));   // sum over H* = |H*| (m_0 + m_|H*| + m_2|H*|) = 0
  for (auto& p : P) RC(p->first_round(assignments + p->q0, sg));
  if (sh.flag) HIPCHK(hipMemcpyAsync(sh.pin_small + PIN_FLAG, sh.flag, 4, hipMemcpyDeviceToHost, s));      // read after the round's commitments
  for (size_t t = 0; t < HC; ++t) sh.blind[3 * K * HC + t] = random_fr(sh.seed, sh.lay_blind_mask + t);
  std::memcpy(sh.stage + sh.st_blind() * 32, sh.blind.data(), sh.blind.size() * 32);
  HIPCHK(hipMemcpyAsync(sh.bl, sh.stage + sh.st_blind() * 32, sh.blind.size() * 32, hipMemcpyHostToDevice, s));
  sh.wit_aff.assign(104 * (3 * K + 1), 0);
  {
    // with the evaluations against the Lagrange powers AND a narrow-window table over [hiding powers | Lagrange powers | v_H G] the 3K witness
    // commitments are one sparse chain (their scalars are mostly 0 / 1), the mask (uniform coefficients against the monomial powers) another
    const PinnedBases& pb = sh.pb;
    bool split = pb.range.d != nullptr;
    for (auto& p : P) split = split && p->lagrange() && pb.range_off <= p->ix.gamma_offset && p->ix.lagrange_offset + p->n_h + 1 <= pb.range_off + pb.range.cover &&
                              p->ix.lagrange_offset >= pb.range_off && p->ix.gamma_offset + HC <= pb.range_off + pb.range.cover;
    std::vector<MsmSeg>& dst = split ? sm : sg;
    MsmSeg a; a.d_ptr = sh.mask; a.len = 3 * N; a.off = 0; a.out = split ? 0u : (uint32_t)(3 * K); dst.push_back(a);
    MsmSeg b; b.d_ptr = sh.bl + 3 * K * HC * 32; b.len = HC; b.off = sh.gamma_offset; b.out = a.out; dst.push_back(b);
    // needs no challenge: behind the (last) commitment chain — the operands of the sumcheck on the device, and on the host the part of the
    // transcript that precedes the first commitments (Varuna::init_sponge: ~ 20 permutations while the GPU accumulates)
    hook = [this]() -> int32_t { for (auto& p : P) RC(p->second_round_early()); return init_sponge(); };
    if (split) { njobs = 2; job[0].k = (uint32_t)(3 * K); job[0].sparse = true; job[0].out = sh.wit_aff.data(); job[1].k = 1; job[1].sparse = false; job[1].out = sh.wit_aff.data() + 104 * 3 * K; }
    else { njobs = 1; job[0].k = (uint32_t)(3 * K + 1); job[0].sparse = false; job[0].out = sh.wit_aff.data(); }
  }
  return ALEO_MI355X_OK;
}

int32_t Batch::first_finish() {
  const size_t K = sh.K;
  if (sh.flag) { uint32_t f; std::memcpy(&f, sh.pin_small + PIN_FLAG, 4); if (f) { g_last_error = "varuna_prove: assignment not canonical (an entry is not below r)"; return ALEO_MI355X_ERR_BAD_ARG; } }
  sh.fs.absorb_g1(sh.wit_aff.data(), 104, 3 * K + 1);
  // verifier_first_round [UPSTREAM-RECALL]: per circuit k_j − 1 instance combiners and (but for the first circuit) a circuit combiner in one squeeze,
  // then alpha, eta_b, eta_c in one squeeze; an instance's combiner = circuit combiner * instance combiner
  sh.comb.assign(K, sh.one);
  for (auto& p : P) {
    HFr el[MAX_INSTANCES]; const size_t cnt = p->k - 1 + (p->j ? 1 : 0);
    sh.fs.squeeze_full(el, cnt);
    const HFr cc = p->j ? el[p->k - 1] : sh.one;
    sh.comb[p->q0] = cc;
    for (size_t i = 1; i < p->k; ++i) sh.comb[p->q0 + i] = HFr::mul(cc, el[i - 1]);
  }
  { HFr el[3]; sh.fs.squeeze_full(el, 3); sh.alpha = el[0]; sh.eta_b = el[1]; sh.eta_c = el[2]; }
  sh.t_mark[1] = now_ms();
  return ALEO_MI355X_OK;
}

int32_t Batch::second_prepare() {
  Ctx* c = sh.c; hipStream_t s = sh.s; const size_t N = sh.N;
  TAKE_M(sh.h1, 2 * N) TAKE_M(sh.g1, N)
  RC(P[sh.lead]->second_round());                                                             // writes h_1, X g_1 (with the mask) in place
  for (auto& p : P) {
    if (p->lead()) continue;
    RC(p->second_round());
    RC(fr_vec_op(c, sh.h1, sh.h1, p->hq, 2 * p->n_h, 1, s));                                  // s_j h_j v_{H_j} = h_j v_{H*}
    RC(fr_add_tiled(c, sh.g1, N, p->rq, p->n_h, s));                                          // s_j (X g_j): the remainder block repeated |H*| / |H_j| times
  }
  {
    std::vector<MsmSeg>& sg = job[0].segs; sg.assign(2, MsmSeg{});
    sg[0].d_ptr = sh.g1 + 32; sg[0].len = N - 1; sg[0].off = sh.D - (N - 2); sg[0].out = 0;    // degree bound |H*| − 2: shifted powers
    sg[1].d_ptr = sh.h1; sg[1].len = 2 * N; sg[1].off = 0; sg[1].out = 1;
    njobs = 1; job[0].k = 2; job[0].sparse = false; job[0].out = sh.aff2; hook = nullptr;
  }
  return ALEO_MI355X_OK;
}

int32_t Batch::second_finish() {
  for (size_t j = 0; j < sh.m; ++j) {                                                         // the commitments returned after the stream drained: the copies have landed
    uint64_t sum[4]; std::memcpy(sum, sh.pin_small + PIN_SUMS + 32 * j, 32);
    if (sum[0] | sum[1] | sum[2] | sum[3]) { g_last_error = "varuna_prove: the assignment does not satisfy the circuit (first sumcheck: the sum over H is not zero)"; return ALEO_MI355X_ERR_UNSATISFIED; }
  }
  sh.fs.absorb_g1(sh.aff2, 104, 2);
  sh.fs.squeeze_full(&sh.beta, 1);
  sh.t_mark[2] = now_ms();
  return ALEO_MI355X_OK;
}

int32_t Batch::third_prepare() {
  const size_t m = sh.m;
  for (auto& p : P) RC(p->third_round());
  std::vector<MsmSeg>& sg = job[0].segs; sg.assign(3 * m, MsmSeg{});
  for (auto& p : P)
    for (size_t M = 0; M < 3; ++M) {
      MsmSeg& g = sg[3 * p->j + M]; g.d_ptr = p->f + (p->ko[M] + 1) * 32; g.len = p->nk[M] - 1; g.off = sh.D - (p->nk[M] - 2); g.out = (uint32_t)(3 * p->j + M);
    }
  sh.aff3.assign(312 * m, 0);
  njobs = 1; job[0].k = (uint32_t)(3 * m); job[0].sparse = false; job[0].out = sh.aff3.data();
  hook = [this]() -> int32_t { for (auto& p : P) RC(p->fourth_round_early()); return ALEO_MI355X_OK; };
  return ALEO_MI355X_OK;
}

int32_t Batch::third_finish() {
  const size_t m = sh.m;
  for (auto& p : P)                                                                           // the sums f_{j,M}(0) |K| were copied out ahead of the commitments: no stream sync of their own
    for (size_t M = 0; M < 3; ++M) { HFr v; std::memcpy(v.l, sh.pin_small + 32 * (3 * p->j + M), 32); p->sigma[M] = HFr::mul(v, fr_u64(p->nk[M])); }
  sh.fs.absorb_g1(sh.aff3.data(), 104, 3 * m);                                                  // absorb_with_msg: the commitments, then the sums circuit by circuit
  for (auto& p : P) sh.fs.absorb_fr(p->sigma, 3);
  {
    std::vector<HFr> el(3 * m); el[0] = sh.one; sh.fs.squeeze_full(el.data() + 1, 3 * m - 1);   // delta_{0,a} = 1, the rest from one squeeze
    for (auto& p : P) for (size_t M = 0; M < 3; ++M) p->delta[M] = el[3 * p->j + M];
  }
  sh.t_mark[3] = now_ms();
  return ALEO_MI355X_OK;
}

int32_t Batch::fourth_prepare() {
  Ctx* c = sh.c; hipStream_t s = sh.s;
  TAKE_M(sh.h2, sh.n_kmax)
  std::vector<const void*> terms; std::vector<size_t> lens; std::vector<HFr> co;
  for (auto& p : P) { RC(p->fourth_round()); for (size_t r = 0; r < p->nrun; ++r) { terms.push_back(p->r4_terms[r]); lens.push_back(p->r4_lens[r]); co.push_back(sh.one); } }
  RC(lincomb_any(c, sh.h2, sh.n_kmax, HFr::zero(), terms, lens, co, s));                       // h_2 = sum_{j,M} delta_{j,M} h_{j,M}
  {
    std::vector<MsmSeg>& sg = job[0].segs; sg.assign(1, MsmSeg{}); sg[0].d_ptr = sh.h2; sg[0].len = sh.n_kmax; sg[0].off = 0; sg[0].out = 0;
    njobs = 1; job[0].k = 1; job[0].sparse = false; job[0].out = sh.aff4; hook = nullptr;
  }
  return ALEO_MI355X_OK;
}

int32_t Batch::fourth_finish() {
  sh.fs.absorb_g1(sh.aff4, 104, 1);
  sh.fs.squeeze_full(&sh.gamma, 1);
  sh.t_mark[4] = now_ms();
  return ALEO_MI355X_OK;
}

int32_t Batch::open_evaluate() {
  Ctx* c = sh.c; hipStream_t s = sh.s; const size_t K = sh.K, N = sh.N, m = sh.m, ne = K + 1 + 3 * m;
  const HFr &beta = sh.beta, &gamma = sh.gamma;
  // ---- evaluations -------------------------------------------------------------------------------------------------------------------------------
  TAKE_M(sh.evd, ne + 8)
  char* evd = sh.evd;
  {
    std::vector<const void*> polys; std::vector<size_t> lens; std::vector<HFr> pts;
    for (auto& p : P) for (size_t i = 0; i < p->k; ++i) { polys.push_back(p->wit + (3 * i + 2) * p->L * 32); lens.push_back(p->L); pts.push_back(beta); }
    polys.push_back(sh.g1 + 32); lens.push_back(N - 1); pts.push_back(beta);
    for (auto& p : P) for (size_t M = 0; M < 3; ++M) { polys.push_back(p->f + (p->ko[M] + 1) * 32); lens.push_back(p->nk[M] - 1); pts.push_back(gamma); }
    for (size_t at = 0; at < ne; at += 12) { const size_t cnt = ne - at < 12 ? ne - at : 12; RC(fr_eval_batch(c, evd + at * 32, polys.data() + at, lens.data() + at, pts.data() + at, cnt, s)); }
  }
  HIPCHK(hipMemcpyAsync(sh.pin_small, evd, ne * 32, hipMemcpyDeviceToHost, s));
  return ALEO_MI355X_OK;                                                                      // the caller synchronises the stream (once for all proofs of a lockstep call)
}

int32_t Batch::open_prepare() {
  Ctx* c = sh.c; hipStream_t s = sh.s; const size_t K = sh.K, N = sh.N, m = sh.m, n_k = sh.n_kmax, ne = K + 1 + 3 * m;
  const HFr &alpha = sh.alpha, &beta = sh.beta, &gamma = sh.gamma, &eta_b = sh.eta_b, &eta_c = sh.eta_c, &one = sh.one;
  char* evd = sh.evd;
  TAKE_S(pbeta, 3 * N) TAKE_S(wq, 3 * N) TAKE_S(blq, HC) TAKE_S(pg, n_k) TAKE_S(gq, n_k)
  sh.evals.assign(ne, HFr::zero());
  for (size_t i = 0; i < ne; ++i) std::memcpy(sh.evals[i].l, sh.pin_small + 32 * i, 32);
  {
    std::vector<HFr> ser(sh.evals.begin(), sh.evals.begin() + K + 1);                          // Evaluations as serialised: z_b's, g_1, every g_a, every g_b, every g_c
    for (size_t M = 0; M < 3; ++M) for (size_t j = 0; j < m; ++j) ser.push_back(sh.evals[K + 1 + 3 * j + M]);
    sh.fs.absorb_fr(ser.data(), ser.size());
  }
  // one short challenge per polynomial of an opening [UPSTREAM-RECALL: sonic_pc combine_for_open], the point beta first:
  // beta: g_1, z_b of every instance, the lincheck combination;  gamma: g_{j,M} circuit by circuit, the matrix combination
  sh.ch_b.resize(K + 2); sh.ch_g.resize(3 * m + 1);
  for (auto& v : sh.ch_b) v = sh.fs.squeeze_short();          // (ch_g: squeezed below, behind the launch of the beta combination — the same sponge calls in the same order, ~2 permutations off the GPU's idle time)
  const HFr g1_beta = sh.evals[K];
  // one inversion for everything the openings divide by: alpha − beta, v_{H_j}(beta) (selectors), v_{K_{j,M}}(gamma)
  std::vector<HFr> inv(1 + 4 * m);
  inv[0] = HFr::sub(alpha, beta);
  for (auto& p : P) { inv[1 + p->j] = p->vh_beta; for (size_t M = 0; M < 3; ++M) inv[1 + m + 3 * p->j + M] = vanish(p->nk[M], gamma); }
  for (size_t i = 1 + m; i < inv.size(); ++i) if (inv[i].is_zero()) { g_last_error = "varuna_prove: gamma landed in K"; return ALEO_MI355X_ERR_HIP; }
  if (inv[0].is_zero()) { g_last_error = "varuna_prove: alpha equals beta"; return ALEO_MI355X_ERR_HIP; }
  batch_inverse_vec(inv);
  // ---- the linear combination of the first sumcheck, opened at beta together with g_1 and the z_b,i -----------------------------------------------
  const HFr xl = sh.ch_b[K + 1], vN_beta = vanish(N, beta);
  HFr cst = HFr::neg(HFr::mul(beta, g1_beta));
  HFr blw[3];                                                                     // blw: (bl(X) − bl(beta)) / (X − beta), uploaded below
  {
    std::vector<const void*> terms; std::vector<size_t> lens; std::vector<HFr> co;
    auto term = [&](const void* p, size_t n, const HFr& k) { terms.push_back(p); lens.push_back(n); co.push_back(k); };
    term(sh.mask, 3 * N, xl); term(sh.h1, 2 * N, HFr::neg(HFr::mul(xl, vN_beta))); term(sh.g1 + 32, N - 1, sh.ch_b[0]);
    HFr blc[3] = {HFr::zero(), HFr::zero(), HFr::zero()};
    auto axpy = [&](const HFr& coef, const HFr* src) { for (size_t t = 0; t < HC; ++t) blc[t] = HFr::add(blc[t], HFr::mul(coef, src[t])); };
    axpy(xl, &sh.blind[3 * K * HC]);
    for (auto& pp : P) {
      Prover& p = *pp;
      const HFr r_ab = HFr::mul(HFr::sub(p.vh_alpha, p.vh_beta), inv[0]);
      const HFr t_beta = HFr::add(p.sigma[0], HFr::add(HFr::mul(eta_b, p.sigma[1]), HFr::mul(eta_c, p.sigma[2])));
      const HFr sel = p.n_h == N ? one : HFr::mul(vN_beta, inv[1 + p.j]), vx_beta = vanish(p.n_x, beta);      // s_j(beta) = v_{H*}(beta) / v_{H_j}(beta)
      for (size_t i = 0; i < p.k; ++i) {
        const size_t q = p.q0 + i; const HFr& xpow = sh.ch_b[1 + q];
        const HFr x_beta = horner(p.x_poly[i], beta), zb = sh.evals[q], ci = HFr::mul(sh.comb[q], sel);
        const HFr k_za = HFr::mul(HFr::mul(xl, ci), HFr::mul(r_ab, HFr::add(one, HFr::mul(eta_c, zb))));
        const HFr k_w = HFr::neg(HFr::mul(HFr::mul(xl, ci), HFr::mul(t_beta, vx_beta)));
        cst = HFr::add(cst, HFr::mul(ci, HFr::sub(HFr::mul(HFr::mul(r_ab, eta_b), zb), HFr::mul(t_beta, x_beta))));
        term(p.wit + (3 * i + 1) * p.L * 32, p.L, k_za); term(p.wit + (3 * i) * p.L * 32, p.L, k_w); term(p.wit + (3 * i + 2) * p.L * 32, p.L, xpow);
        axpy(k_w, &sh.blind[(3 * q) * HC]); axpy(k_za, &sh.blind[(3 * q + 1) * HC]); axpy(xpow, &sh.blind[(3 * q + 2) * HC]);
      }
    }
    RC(lincomb_any(c, pbeta, 3 * N, HFr::mul(xl, cst), terms, lens, co, s));
    for (auto& v : sh.ch_g) v = sh.fs.squeeze_short();
    sh.random_v = HFr::add(blc[0], HFr::mul(beta, HFr::add(blc[1], HFr::mul(beta, blc[2]))));
    blw[1] = blc[2]; blw[0] = HFr::add(blc[1], HFr::mul(beta, blc[2])); blw[2] = HFr::zero();
    char* st = sh.stage + sh.st_blq() * 32; std::memcpy(st, blw, HC * 32);
    HIPCHK(hipMemcpyAsync(blq, st, HC * 32, hipMemcpyHostToDevice, s));
  }
  // ---- the linear combination of the second sumcheck, opened at gamma together with every g_{j,M} ------------------------------------------------------
  {
    const HFr xi3m = sh.ch_g[3 * m], vk_gamma = vanish(n_k, gamma);
    std::vector<const void*> terms; std::vector<size_t> lens; std::vector<HFr> co; HFr cg = HFr::zero();
    auto term = [&](const void* p, size_t n, const HFr& k) { terms.push_back(p); lens.push_back(n); co.push_back(k); };
    for (auto& pp : P) {
      Prover& p = *pp;
      for (size_t M = 0; M < 3; ++M) {
        const HFr fm = HFr::add(HFr::mul(gamma, sh.evals[K + 1 + 3 * p.j + M]), HFr::mul(p.sigma[M], inv_pow2(p.lg_km[M])));
        const HFr d = HFr::mul(HFr::mul(p.delta[M], xi3m), HFr::mul(vk_gamma, inv[1 + m + 3 * p.j + M]));     // selector v_{K*} / v_{K_M} at gamma
        const HFr dfm = HFr::mul(d, fm);
        const HFr cf[4] = {HFr::mul(dfm, beta), HFr::mul(dfm, alpha), HFr::mul(d, p.vv), HFr::neg(dfm)};      // row, col, val, row_col
        for (int t = 0; t < 4; ++t) term((const char*)p.ix.k_polys + (4 * p.ko[M] + (size_t)t * p.nk[M]) * 32, p.nk[M], cf[t]);
        cg = HFr::sub(cg, HFr::mul(HFr::mul(dfm, alpha), beta));
      }
    }
    term(sh.h2, n_k, HFr::neg(HFr::mul(xi3m, vk_gamma)));
    for (auto& pp : P) for (size_t M = 0; M < 3; ++M) term(pp->f + (pp->ko[M] + 1) * 32, pp->nk[M] - 1, sh.ch_g[3 * pp->j + M]);
    RC(lincomb_any(c, pg, n_k, cg, terms, lens, co, s));
  }
  {                                                                                          // both witness polynomials in the same three launches
    void* q[2] = {wq, gq}; void* ev2[2] = {evd + (ne + 1) * 32, evd + (ne + 2) * 32}; const void* pp[2] = {pbeta, pg}; const size_t nn[2] = {3 * N, n_k}; const void* zz[2] = {beta.l, gamma.l};
    RC(fr_divide_by_linear_many(c, q, ev2, pp, nn, zz, 2, s));
  }
  {
    std::vector<MsmSeg>& sg = job[0].segs; sg.assign(3, MsmSeg{});
    sg[0].d_ptr = wq; sg[0].len = 3 * N - 1; sg[0].off = 0; sg[0].out = 0;
    sg[1].d_ptr = blq; sg[1].len = HC - 1; sg[1].off = sh.gamma_offset; sg[1].out = 0;
    sg[2].d_ptr = gq; sg[2].len = n_k - 1; sg[2].off = 0; sg[2].out = 1;
    njobs = 1; job[0].k = 2; job[0].sparse = false; job[0].out = sh.aff5; hook = nullptr;      // both witness commitments in one call
  }
  return ALEO_MI355X_OK;
}

int32_t Batch::write(uint8_t* out, size_t* out_len) {
  // ---- the proof in upstream's layout ---------------------------------------------------------------------------------------------------------------
  aleo_mi355x_proof_parts parts{}; std::vector<uint64_t> batch; std::vector<HFr> sums;
  for (auto& p : P) { batch.push_back(p->k); for (size_t M = 0; M < 3; ++M) sums.push_back(p->sigma[M]); }
  uint8_t has_v[2] = {1, 0}; HFr rv[2] = {sh.random_v, HFr::zero()};
  parts.batch_sizes = batch.data(); parts.n_circuits = sh.m; parts.witness_commitments = sh.wit_aff.data(); parts.mask_poly = sh.wit_aff.data() + 104 * 3 * sh.K;
  parts.g_1 = sh.aff2; parts.h_1 = sh.aff2 + 104; parts.g_abc = sh.aff3.data(); parts.h_2 = sh.aff4;
  parts.evaluations = sh.evals.data(); parts.n_evaluations = sh.evals.size(); parts.sums = sums.data();
  parts.opening_points = sh.aff5; parts.opening_random_v = rv; parts.opening_has_v = has_v; parts.n_openings = 2;
  RC(aleo_mi355x_proof_to_bytes(out, out_len, &parts));
  for (int i = 0; i < 5; ++i) g_varuna_timing[i] = sh.t_mark[i + 1] - sh.t_mark[i];
  g_varuna_timing[5] = sh.t_mark[5] - sh.t_mark[0];
  return ALEO_MI355X_OK;
}

// The commitments of one round for every proof that is still alive: job q of all of them in ONE launch chain (the results of proof p follow those
// of proof p - 1), the hooks of all of them behind the last chain.  Proofs whose job lists differ in shape (one splits its first round into the
// sparse witness chain + the mask chain, another does not) cannot share: the caller (prove_many) falls back to one proof at a time.
static int32_t run_commits(Ctx* c, const PinnedBases& pb, std::vector<Batch*>& bs, hipStream_t s) {
  if (bs.empty()) return ALEO_MI355X_OK;
  const int nj = bs[0]->njobs;
  for (int q = 0; q < nj; ++q) {
    std::vector<MsmSeg> all; uint32_t k = 0; const bool sparse = bs[0]->job[q].sparse;
    for (Batch* b : bs) { for (MsmSeg sg : b->job[q].segs) { sg.out += k; all.push_back(sg); } k += b->job[q].k; }
    std::function<int32_t()> behind = nullptr;
    if (q == nj - 1) behind = [&bs]() -> int32_t { for (Batch* b : bs) if (b->hook) RC(b->hook()); return ALEO_MI355X_OK; };
    if (bs.size() == 1) { RC(commit(c, pb, all, k, bs[0]->job[q].out, s, sparse, std::move(behind))); continue; }
    std::vector<uint8_t> out((size_t)104 * k);
    RC(commit(c, pb, all, k, out.data(), s, sparse, std::move(behind)));
    size_t at = 0;
    for (Batch* b : bs) { std::memcpy(b->job[q].out, out.data() + 104 * at, (size_t)104 * b->job[q].k); at += b->job[q].k; }
  }
  return ALEO_MI355X_OK;
}

// assignments: the instances of circuit 0, then of circuit 1, ... (sum of ks pointers)
int32_t varuna_prove_batch(Ctx* c, const PinnedBases& pb, const aleo_mi355x_varuna_index* const* ixs, size_t m, const void* const* assignments, const size_t* ks,
                           const uint8_t* seed32, uint8_t* out, size_t* out_len) {
  g_varuna_timing[6] = g_varuna_timing[7] = 0;
  HT("prove: enter");
  Batch b(c, pb, seed32); std::vector<Batch*> one{&b}; hipStream_t s = c->stream;
  RC(b.setup(ixs, m, ks));
  RC(reserve_prover_memory(c, b.need_ws_bytes, b.need_pin_bytes));
  RC(b.attach((char*)c->prover_ws.p, c->prover_ws.cap, (char*)c->prover_pin));
  HT("prove: setup done");
  RC(b.first_prepare(assignments)); HT("r1 prepared"); RC(run_commits(c, pb, one, s)); HT("r1 committed"); RC(b.first_finish()); HT("r1 transcript");
  RC(b.second_prepare()); HT("r2 prepared"); RC(run_commits(c, pb, one, s)); HT("r2 committed"); RC(b.second_finish()); HT("r2 transcript");
  RC(b.third_prepare()); HT("r3 prepared"); RC(run_commits(c, pb, one, s)); HT("r3 committed"); RC(b.third_finish()); HT("r3 transcript");
  RC(b.fourth_prepare()); HT("r4 prepared"); RC(run_commits(c, pb, one, s)); HT("r4 committed"); RC(b.fourth_finish()); HT("r4 transcript");
  RC(b.open_evaluate()); HT("evals queued"); HIPCHK(hipStreamSynchronize(s)); HT("evals arrived"); RC(b.open_prepare()); HT("open prepared"); RC(run_commits(c, pb, one, s)); HT("open committed");
  b.sh.t_mark[5] = now_ms();
  const int32_t wrc = b.write(out, out_len);
  HT("proof written");
  if (g_host_trace_on) host_trace_mark(nullptr);
  return wrc;
}

// Several INDEPENDENT proofs in lockstep (aleo_mi355x_varuna_prove_many): every proof keeps its own transcript, challenges, randomness and workspace
// slice; what they share is every commitment launch chain (round r of all proofs is one batched MSM: its sort, slice tree, reduction, host tail and
// stream synchronisation are paid once, not once per proof).  Between the commitments the proofs are independent, so they are dealt to W worker
// threads (the caller's thread is worker 0; the others borrow helper contexts of the device: own stream, own scratch): the transcripts — host
// Poseidon, the serial part of a proof — run W at a time and the small field / NTT kernels of different proofs overlap on the card.  The workers meet
// at a barrier before and after each round's commitments, which worker 0 launches on the slot's stream behind an event of every helper stream.
// A proof that fails (unsatisfied assignment, bad argument) drops out with its status; the others go on.  Byte for byte the proofs of the
// single-proof entry points under the same seeds.

int32_t varuna_prove_many(Ctx* c, const PinnedBases& pb, std::vector<ProveRequest>& rq, int workers) {
  g_varuna_timing[6] = g_varuna_timing[7] = 0;
  const double t0 = now_ms();
  HT("many: enter");
  hipStream_t s = c->stream; const size_t n = rq.size();
  if (workers <= 0) { workers = 4; if (const char* e = std::getenv("ALEO_MI355X_LOCKSTEP_WORKERS")) { const int k = std::atoi(e); if (k >= 1 && k <= MAX_SLOTS + 1) workers = k; } }
  HelperSet hs;
  if (workers > 1 && n > 1) RC(acquire_helpers(c->dev, (int)std::min<size_t>(n, (size_t)workers) - 1, hs));
  std::vector<Ctx*> wc{c}; for (Ctx* h : hs.ctx) wc.push_back(h);
  const size_t W = wc.size();
  std::vector<std::unique_ptr<Batch>> B(n); std::vector<char> alive(n, 0);
  size_t ws_total = 0, pin_total = 0;
  for (size_t p = 0; p < n; ++p) {
    B[p].reset(new Batch(wc[p % W], pb, rq[p].seed32));
    rq[p].status = B[p]->setup(rq[p].ixs.data(), rq[p].ixs.size(), rq[p].ks);
    if (rq[p].status) { rq[p].error = g_last_error; continue; }
    alive[p] = 1; ws_total += (B[p]->need_ws_bytes + 255) & ~(size_t)255; pin_total += (B[p]->need_pin_bytes + 255) & ~(size_t)255;
  }
  RC(reserve_prover_memory(c, ws_total ? ws_total : 256, pin_total ? pin_total : 256));
  size_t ws_at = 0, pin_at = 0;
  for (size_t p = 0; p < n; ++p) {
    if (!alive[p]) continue;
    const size_t w = (B[p]->need_ws_bytes + 255) & ~(size_t)255, h = (B[p]->need_pin_bytes + 255) & ~(size_t)255;
    RC(B[p]->attach((char*)c->prover_ws.p + ws_at, w, (char*)c->prover_pin + pin_at)); ws_at += w; pin_at += h;
  }
  Barrier bar(W); int32_t fatal = ALEO_MI355X_OK; std::string fatal_error;      // fatal: written by worker 0 between two barriers, read by all after the second
  // the commitments of one round, by worker 0 while the others wait: the helper streams' events first (their kernels wrote this round's scalars)
  auto commits = [&]() -> int32_t {
    std::vector<Batch*> live; for (size_t p = 0; p < n; ++p) if (alive[p]) live.push_back(B[p].get());
    if (live.empty()) return ALEO_MI355X_OK;
    for (size_t w = 1; w < W; ++w) HIPCHK(hipStreamWaitEvent(s, wc[w]->ev[0], 0));
    bool uniform = true;
    for (Batch* b : live) { uniform = uniform && b->njobs == live[0]->njobs; for (int q = 0; uniform && q < b->njobs; ++q) uniform = b->job[q].sparse == live[0]->job[q].sparse; }
    if (uniform) return run_commits(c, pb, live, s);
    for (Batch* b : live) { std::vector<Batch*> one{b}; RC(run_commits(c, pb, one, s)); }      // mixed shapes: one proof at a time for this round
    return ALEO_MI355X_OK;
  };
  auto worker = [&](size_t w) {
    if (w && hipSetDevice(c->device) != hipSuccess) { for (size_t p = w; p < n; p += W) if (alive[p]) { alive[p] = 0; rq[p].status = ALEO_MI355X_ERR_HIP; rq[p].error = "hipSetDevice failed"; } }
    hipStream_t sw = wc[w]->stream;
    // one step of this worker's live proofs; a failure removes the proof and records its status (alive[p] is only written by p's worker, and read by
    // worker 0 behind a barrier)
    auto each = [&](const std::function<int32_t(Batch&, size_t)>& f) {
      for (size_t p = w; p < n; p += W) {
        if (!alive[p]) continue;
        int32_t rc;
        try { rc = f(*B[p], p); } catch (...) { rc = ALEO_MI355X_ERR_HIP; g_last_error = "varuna_prove_many: exception in a worker"; }      // never past the barrier protocol
        if (rc) { rq[p].status = rc; rq[p].error = g_last_error; alive[p] = 0; }
      }
    };
    auto fail_mine = [&](int32_t rc, const char* why) { for (size_t p = w; p < n; p += W) if (alive[p]) { alive[p] = 0; rq[p].status = rc; rq[p].error = why; } };
    auto round = [&](const std::function<int32_t(Batch&, size_t)>& prepare, const std::function<int32_t(Batch&, size_t)>& finish) -> bool {
      each(prepare);
      if (!w) HT("many: my proofs prepared");
      if (w && hipEventRecord(wc[w]->ev[0], sw) != hipSuccess) fail_mine(ALEO_MI355X_ERR_HIP, "hipEventRecord failed");
      bar.wait();
      if (!w) HT("many: all prepared");
      if (w == 0) { try { fatal = commits(); } catch (...) { fatal = ALEO_MI355X_ERR_HIP; g_last_error = "varuna_prove_many: exception in the commitments"; } if (fatal) fatal_error = g_last_error; }
      if (!w) HT("many: committed");
      bar.wait();
      if (fatal) return false;
      if (finish) each(finish);
      if (!w) HT("many: my transcripts");
      return true;
    };
    if (!round([&](Batch& b, size_t p) { return b.first_prepare(rq[p].assignments); }, [](Batch& b, size_t) { return b.first_finish(); })) return;
    if (!round([](Batch& b, size_t) { return b.second_prepare(); }, [](Batch& b, size_t) { return b.second_finish(); })) return;
    if (!round([](Batch& b, size_t) { return b.third_prepare(); }, [](Batch& b, size_t) { return b.third_finish(); })) return;
    if (!round([](Batch& b, size_t) { return b.fourth_prepare(); }, [](Batch& b, size_t) { return b.fourth_finish(); })) return;
    each([](Batch& b, size_t) { return b.open_evaluate(); });
    if (hipStreamSynchronize(sw) != hipSuccess) fail_mine(ALEO_MI355X_ERR_HIP, "hipStreamSynchronize failed");
    if (!round([](Batch& b, size_t) { return b.open_prepare(); }, nullptr)) return;
    each([&](Batch& b, size_t p) { b.sh.t_mark[5] = now_ms(); return b.write(rq[p].out, rq[p].out_len); });
  };
  // Helper threads wait at a gate until all of them exist: if one cannot be started, the ones that were leave at once (the barrier counts W threads) and
  // the call fails as a whole instead of unwinding past joinable threads.
  std::vector<std::thread> th; th.reserve(W); std::mutex gate_mu; std::condition_variable gate_cv; int gate = 0;      // 0 wait, 1 go, -1 leave
  bool started = true;
  for (size_t w = 1; w < W && started; ++w) {
    try { th.emplace_back([&, w] { { std::unique_lock<std::mutex> g(gate_mu); gate_cv.wait(g, [&] { return gate != 0; }); if (gate < 0) return; } worker(w); }); }
    catch (...) { started = false; }
  }
  { std::lock_guard<std::mutex> g(gate_mu); gate = started ? 1 : -1; } gate_cv.notify_all();
  HT("many: threads started");
  if (started) worker(0);
  HT("many: worker 0 done");
  for (auto& t : th) t.join();
  HT("many: joined");
  if (!started) { g_last_error = "varuna_prove_many: could not start a worker thread"; return ALEO_MI355X_ERR_HIP; }
  for (size_t w = 1; w < W; ++w) (void)hipStreamSynchronize(wc[w]->stream);      // nothing of this call is left on a helper stream when it goes back to the pool
  if (fatal) { g_last_error = fatal_error; return fatal; }
  for (int i = 0; i < 5; ++i) g_varuna_timing[i] = 0;
  g_varuna_timing[5] = now_ms() - t0;
  if (g_host_trace_on) host_trace_mark(nullptr);
  return ALEO_MI355X_OK;
}

int32_t varuna_prove(Ctx* c, const PinnedBases& pb, const aleo_mi355x_varuna_index& ix, const void* const* assignments, size_t k, const uint8_t* seed32,
                     uint8_t* out, size_t* out_len) {
  const aleo_mi355x_varuna_index* one[1] = {&ix};
  return varuna_prove_batch(c, pb, one, 1, assignments, &k, seed32, out, out_len);
}

}  // namespace aleo_mi355x

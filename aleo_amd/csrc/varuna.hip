// varuna.hip — the host side of one proof, native: the four AHP rounds, the evaluations and the two openings of
// `Varuna::prove_batch` (one circuit, up to eight instances) as ONE call of the C ABI (`aleo_mi355x_varuna_prove`).
//
// Replaces (shape, not bytes — see DESIGN.md §4d for what differs from upstream and why) snarkVM 0.14.5
//   algorithms/src/snark/varuna/varuna.rs                      Varuna::prove_batch
//   algorithms/src/snark/varuna/ahp/prover/round_functions/*   AHPForR1CS::prover_{first,second,third,fourth}_round   [UPSTREAM-RECALL]
// reached from /root/reference/rust/src/program/execute.rs:74 (`trace.prove_execution`) and transfer.rs:99.
// Every circuit-sized step is a kernel of msm.hip / ntt.hip / frops.hip queued on the calling slot's stream; this file keeps what
// upstream keeps on the CPU between them: the transcript, the challenge-dependent constants (host Fr arithmetic, host_field.hpp)
// and the O(|X|) public-input polynomial.  aleo_amd/varuna.py is the same sequence written against the public entry points; both
// must produce the bytes of the restatement in oracle/varuna_ref.py (tests/test_varuna.py).
#include "ctx.h"
#include "host_field.hpp"
#include <cstring>
#include <vector>
#include <memory>
#include <chrono>

namespace aleo_mi355x {

using host::HFr;

// ---- SHA-256 (FIPS 180-4) for the transcript -------------------------------------------------------------------------------------
namespace {
struct Sha256 {
  uint32_t h[8]; uint8_t buf[64]; uint64_t len = 0; size_t fill = 0;
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  Sha256() { static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19}; std::memcpy(h, iv, 32); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
      0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
      0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
      0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
      uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data; len += n;
    while (n) {
      size_t take = 64 - fill < n ? 64 - fill : n;
      std::memcpy(buf + fill, p, take); fill += take; p += take; n -= take;
      if (fill == 64) { block(buf); fill = 0; }
    }
  }
  void finish(uint8_t out[32]) {
    uint64_t bits = len * 8; uint8_t pad = 0x80; update(&pad, 1);
    uint8_t z = 0; while (fill != 56) update(&z, 1);
    uint8_t lb[8]; for (int i = 0; i < 8; ++i) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(lb, 8);
    for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
  }
};

const char LABEL[] = "aleo-mi355x/varuna-synthetic/v1";

// state = SHA-256(state ‖ data); a challenge is the new state read as a little-endian integer mod r
struct Transcript {
  uint8_t state[32];
  Transcript() { Sha256 s; s.update(LABEL, sizeof LABEL - 1); s.finish(state); }
  void absorb(const void* data, size_t n) { Sha256 s; s.update(state, 32); s.update(data, n); s.finish(state); }
  HFr challenge(const char* label, size_t n) {           // Montgomery form
    absorb(label, n);
    uint64_t v[4]; std::memcpy(v, state, 32);
    HFr c = HFr::reduce_lazy(v);                           // < 2^256 < 14 r: a few subtractions
    return HFr::to_mont(c);
  }
};

inline uint64_t mix64(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
// element `index` of the proof's random stream (same definition as k_fr_random in frops.hip), Montgomery form
HFr random_fr(uint64_t seed, uint64_t index) {
  for (uint64_t j = 0;; ++j) {
    uint64_t l[4];
    for (int k = 0; k < 4; ++k) l[k] = mix64(seed + (4 * index + (uint64_t)k + 1) * 0x9E3779B97F4A7C15ull + j * 0xD1B54A32D192ED03ull);
    l[3] &= (1ull << 61) - 1;
    if (!HFr::geq_p(l)) { HFr v; std::memcpy(v.l, l, 32); return HFr::to_mont(v); }
  }
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
inline void batch_inverse(HFr* v, size_t n) {               // Montgomery's trick on the host: one inversion for n non-zero values (n <= 8)
  HFr pre[8], acc = HFr::one();
  for (size_t i = 0; i < n; ++i) { pre[i] = acc; acc = HFr::mul(acc, v[i]); }
  acc = HFr::inv(acc);
  for (size_t i = n; i-- > 0;) { const HFr t = HFr::mul(acc, pre[i]); acc = HFr::mul(acc, v[i]); v[i] = t; }
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


static int32_t commit(Ctx* c, const PinnedBases& pb, const std::vector<MsmSeg>& segs, uint32_t k, uint8_t* out104, hipStream_t s, bool sparse = false) {
  std::vector<uint64_t> jac(18 * (size_t)k);
  MsmJob j; j.segs = segs.data(); j.nseg = (uint32_t)segs.size(); j.k = k; j.mont = true; j.sparse = sparse;
  const double t0 = now_ms();
  RC(msm_batch(c, jac.data(), pb, j, s));
  jacobian_rows_to_affine104(out104, jac.data(), k);
  g_varuna_timing[6] += now_ms() - t0; g_varuna_timing[7] += c->last_msm.host;      // time inside the commitment calls / their host tails (last chain of each call)
  return ALEO_MI355X_OK;
}

// ---- the index of a circuit, built once per proving key ---------------------------------------------------------------------------------
// [UPSTREAM-RECALL: varuna/ahp/indexer — AHPForR1CS::index: matrix arithmetisation over the non-zero domain, index commitments; reached from
// Process::synthesize_key, /root/reference/wasm/src/programs/manager/mod.rs:164-177, rust/src/program/deploy.rs:142,151.]
struct VarunaIndexOwner {
  aleo_mi355x_varuna_index view{};
  std::vector<uint32_t> positions; std::vector<uint8_t> vk;
  std::vector<void*> dev;                                  // every device allocation of the index
  std::shared_ptr<PinnedOwner> key;                        // the committer key stays pinned while the index lives
  ~VarunaIndexOwner() { for (void* p : dev) if (p) (void)hipFree(p); }
  int32_t alloc(void** out, size_t bytes) { void* p = nullptr; HIPCHK(hipMalloc(&p, bytes ? bytes : 32)); dev.push_back(p); *out = p; return ALEO_MI355X_OK; }
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
  uint64_t nnz[3], nnz_max = 0, nnz_sum = 0;
  for (int m = 0; m < 3; ++m) {
    if (!abc[m].row_ptr || abc[m].row_ptr[0] != 0) { g_last_error = "varuna_index: row_ptr must start at 0"; return ALEO_MI355X_ERR_BAD_ARG; }
    nnz[m] = abc[m].row_ptr[n_constraints]; nnz_max = nnz[m] > nnz_max ? nnz[m] : nnz_max; nnz_sum += nnz[m];
    if (nnz[m] && (!abc[m].col || !abc[m].val)) { g_last_error = "varuna_index: null matrix arrays"; return ALEO_MI355X_ERR_BAD_ARG; }
    for (uint64_t e = 0; e < nnz[m]; ++e) if (abc[m].col[e] >= n_vars) { g_last_error = "varuna_index: column outside the variables"; return ALEO_MI355X_ERR_BAD_ARG; }
    for (size_t r = 0; r < n_constraints; ++r) if (abc[m].row_ptr[r + 1] < abc[m].row_ptr[r]) { g_last_error = "varuna_index: row_ptr not monotone"; return ALEO_MI355X_ERR_BAD_ARG; }
  }
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
  V.n_h = n_h; V.n_k_a = nk[0]; V.n_k_b = nk[1]; V.n_k_c = nk[2]; V.n_x = n_x; V.n_public = n_public; V.n_vars = n_vars; V.committer_key = key_handle; V.max_degree = max_degree; V.gamma_offset = gamma_offset; V.lagrange_offset = lagrange_offset;
  if (lagrange_offset && lagrange_offset + n_h + 1 > pb.n) { g_last_error = "varuna_index: the Lagrange powers do not fit the committer key"; return ALEO_MI355X_ERR_BAD_ARG; }
  // host staging of everything that is index arithmetic on integers
  std::vector<uint32_t> rp(n_h + 1), tp(n_h + 1, 0), kidx(2 * k_sum, 0);
  std::vector<uint32_t> cpos[3]; std::vector<uint32_t> tcol(nnz_sum); std::vector<uint8_t> tval(nnz_sum * 32), kval(k_sum * 32, 0);
  for (int m = 0; m < 3; ++m) {
    cpos[m].resize(nnz[m]);
    for (uint64_t e = 0; e < nnz[m]; ++e) { cpos[m][e] = o->positions[abc[m].col[e]]; tp[cpos[m][e] + 1]++; }
    for (size_t r = 0; r < n_constraints; ++r)
      for (uint64_t e = abc[m].row_ptr[r]; e < abc[m].row_ptr[r + 1]; ++e) { kidx[2 * ko[m] + e] = (uint32_t)r; kidx[2 * ko[m] + nk[m] + e] = cpos[m][e]; }
    if (nnz[m]) std::memcpy(&kval[ko[m] * 32], abc[m].val, nnz[m] * 32);
  }
  for (uint64_t i = 0; i < n_h; ++i) tp[i + 1] += tp[i];
  {
    std::vector<uint32_t> cur(tp.begin(), tp.end() - 1);
    for (int m = 0; m < 3; ++m)
      for (size_t r = 0; r < n_constraints; ++r)
        for (uint64_t e = abc[m].row_ptr[r]; e < abc[m].row_ptr[r + 1]; ++e) {
          const uint32_t at = cur[cpos[m][e]]++;
          tcol[at] = (uint32_t)(m * n_h + r); std::memcpy(&tval[(size_t)at * 32], (const uint8_t*)abc[m].val + e * 32, 32);
        }
  }
  auto up = [&](void** dst, const void* src, size_t bytes) -> int32_t { RC(o->alloc(dst, bytes)); if (bytes) HIPCHK(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, s)); return ALEO_MI355X_OK; };
  auto to_mont = [&](void* p, size_t n) -> int32_t { return fr_lin(c, p, n, nullptr, r2.l, p, nullptr, nullptr, s); };
  void* d;
  for (int m = 0; m < 2; ++m) {                            // forward matrices with columns on H, rows padded to |H|
    for (uint64_t i = 0; i <= n_h; ++i) rp[i] = i <= n_constraints ? abc[m].row_ptr[i] : (uint32_t)nnz[m];
    void *drp, *dcol, *dval;
    RC(up(&drp, rp.data(), (n_h + 1) * 4)); RC(up(&dcol, cpos[m].data(), nnz[m] * 4)); RC(up(&dval, abc[m].val, nnz[m] * 32)); RC(to_mont(dval, nnz[m]));
    HIPCHK(hipStreamSynchronize(s));                       // rp is reused by the next matrix
    if (m == 0) { V.a_row_ptr = drp; V.a_col = dcol; V.a_val = dval; } else { V.b_row_ptr = drp; V.b_col = dcol; V.b_val = dval; }
  }
  RC(up(&d, tp.data(), (n_h + 1) * 4)); V.t_row_ptr = d; RC(up(&d, tcol.data(), nnz_sum * 4)); V.t_col = d;
  RC(up(&d, tval.data(), nnz_sum * 32)); RC(to_mont(d, nnz_sum)); V.t_val = d;
  // 1 / v_X on H \ X (v_X(w^p) = wx^p − 1, wx = w^|X|; zeros stay zero through the batch inversion), elements of H
  void *vx, *he;
  RC(o->alloc(&vx, n_h * 32)); RC(o->alloc(&he, n_h * 32));
  const HFr gen_h = domain_gen(n_h), wx = HFr::pow_u64(gen_h, n_x), neg1 = HFr::neg(one);
  RC(fr_powers(c, vx, n_h, one.l, wx.l, s)); RC(fr_lin(c, vx, n_h, neg1.l, one.l, vx, nullptr, nullptr, s)); RC(fr_batch_inverse(c, vx, n_h, s));
  RC(fr_powers(c, he, n_h, one.l, gen_h.l, s));
  V.vx_inv = vx;
  // arithmetisation over K: row, col, val = M[r,c] col / |H|, row_col — padding: row = col = 1 (position 0), val = 0
  void *kev, *kid, *kpo, *k2, *kv;
  RC(o->alloc(&kev, 4 * k_sum * 32)); RC(o->alloc(&kpo, 4 * k_sum * 32)); RC(o->alloc(&k2, 8 * k_sum * 32));
  RC(up(&kid, kidx.data(), 2 * k_sum * 4)); RC(up(&kv, kval.data(), k_sum * 32)); RC(to_mont(kv, k_sum));
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
    RC(ntt_run(c, po, lg, 4, 0, 1, 0, s));
    for (int j = 0; j < 4; ++j) HIPCHK(hipMemcpyAsync(e2 + (size_t)j * 2 * n * 32, po + (size_t)j * n * 32, n * 32, hipMemcpyDeviceToDevice, s));
    RC(ntt_run(c, e2, lg + 1, 4, 0, 0, 0, s));
  }
  V.k_evals = kev; V.k_idx = kid; V.k_polys = kpo; V.k2_evals = k2; V.positions = o->positions.data();
  { void* dp; RC(up(&dp, o->positions.data(), n_vars * 4)); V.positions_device = dp; }
  // index commitments -> what the transcript absorbs first
  uint8_t aff[12 * 104];
  {
    std::vector<MsmSeg> sg(12);
    for (int q = 0; q < 12; ++q) { const int m = q / 4, j = q % 4; sg[q].d_ptr = (char*)kpo + (4 * ko[m] + (size_t)j * nk[m]) * 32; sg[q].len = nk[m]; sg[q].off = 0; sg[q].out = (uint32_t)q; }
    RC(commit(c, pb, sg, 12, aff, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  o->vk.resize(12 * 48 + 40);
  RC(aleo_mi355x_g1_compress(o->vk.data(), aff, 12));
  const uint64_t dims[5] = {n_h, nk[0], nk[1], nk[2], n_x}; std::memcpy(&o->vk[12 * 48], dims, 40);
  V.vk_bytes = o->vk.data(); V.vk_len = o->vk.size();
  *out = o.release();
  return ALEO_MI355X_OK;
}


// The state of one proof between the rounds (upstream: varuna::ahp::prover::State) and the round functions in the order upstream calls them.
#define TAKE_M(var, elems) var = ar.take(elems); if (!var) { g_last_error = "varuna_prove: workspace accounting"; return ALEO_MI355X_ERR_HIP; }
struct Prover {
  static constexpr size_t HC = 3;                          // coefficients of a hiding polynomial (hiding bound 1)
  Ctx* c; const PinnedBases& pb; const aleo_mi355x_varuna_index& ix; const size_t k; const uint64_t seed;
  Prover(Ctx* c_, const PinnedBases& pb_, const aleo_mi355x_varuna_index& ix_, size_t k_, uint64_t seed_) : c(c_), pb(pb_), ix(ix_), k(k_), seed(seed_) {}
  // sizes, stream, workspace
  size_t n_h = 0, n_x = 0, L = 0, n4 = 0, nk[3] = {}, ko[3] = {}, k_sum = 0, n_k = 0; uint64_t D = 0; uint32_t lg_h = 0, lg_km[3] = {};
  hipStream_t s = nullptr; double t_mark[7] = {}; Arena ar{nullptr, 0, 0}; char* pin = nullptr; char* stage = nullptr; char* pin_small = nullptr;
  HFr one, neg1, r2; Transcript tr; uint64_t lay_mask = 0, lay_blind = 0, lay_blind_mask = 0;
  // what the rounds hand on: polynomials in HBM, commitments, challenges
  char *xp = nullptr, *wit = nullptr, *mask = nullptr, *ext = nullptr, *h1 = nullptr, *g1 = nullptr, *f = nullptr, *h2 = nullptr;
  std::vector<std::vector<HFr>> x_poly; std::vector<HFr> blind, comb, evals; std::vector<uint8_t> wit_aff, comp;
  uint8_t aff2[208], aff3[312], aff4[104], aff5[208];
  HFr alpha, eta_b, eta_c, vh_alpha, beta, vh_beta, vv, sigma[3], delta[3], gamma, random_v;
  size_t run0[3] = {}, runc[3] = {}, nrun = 0;

  int32_t setup(const void* const* assignments);
  int32_t first_round(const void* const* assignments);     // AHPForR1CS::prover_first_round + the 3k + 1 hiding commitments
  int32_t second_round();                                  // prover_second_round: t, the first sumcheck, g_1, h_1
  int32_t third_round();                                   // prover_third_round: f_M, sigma_M, g_M
  int32_t fourth_round();                                  // prover_fourth_round: h_2
  int32_t open();                                          // evaluations, the two linear combinations, both KZG openings
  int32_t write(uint8_t* out, size_t* out_len);            // Proof::write_le
};

int32_t Prover::setup(const void* const* assignments) {
  (void)assignments;
  n_h = ix.n_h; n_x = ix.n_x; L = n_h + 1; n4 = 4 * n_h;
  nk[0] = ix.n_k_a; nk[1] = ix.n_k_b; nk[2] = ix.n_k_c; ko[0] = 0; ko[1] = nk[0]; ko[2] = nk[0] + nk[1]; k_sum = nk[0] + nk[1] + nk[2];
  n_k = nk[0] > nk[1] ? (nk[0] > nk[2] ? nk[0] : nk[2]) : (nk[1] > nk[2] ? nk[1] : nk[2]);      // K: the largest non-zero domain
  D = ix.max_degree;
  bool k_ok = true; for (int m = 0; m < 3; ++m) k_ok = k_ok && nk[m] >= 2 && !(nk[m] & (nk[m] - 1));
  if (k < 1 || k > 8 || n_h < 2 || !k_ok || n_x < 1 || n_h < 2 * n_x || (n_h & (n_h - 1)) || (n_x & (n_x - 1)) ||
      ix.n_public > n_x || ix.n_vars > n_h || ix.gamma_offset + HC > pb.n || (ix.lagrange_offset && ix.lagrange_offset + n_h + 1 > pb.n) || D + 1 > pb.n || 3 * n_h > D + 1 || n_k > D + 1) {
    g_last_error = "varuna_prove: inconsistent index / key sizes"; return ALEO_MI355X_ERR_BAD_ARG;
  }
  lg_h = 0; lg_km[0] = lg_km[1] = lg_km[2] = 0; while ((1ull << lg_h) < n_h) ++lg_h;
  for (int m = 0; m < 3; ++m) while ((1ull << lg_km[m]) < nk[m]) ++lg_km[m];
  s = c->stream;
  t_mark[0] = now_ms();
  // ---- workspace ------------------------------------------------------------------------------------------------------------------
  const size_t elems = n_h * (41 + 24 * k) + k_sum * 6 + n_k * 4 + 4096;
  RC(c->prover_ws.reserve(elems * 32 + (64 << 10)));
  ar = Arena{(char*)c->prover_ws.p, 0, c->prover_ws.cap};
  const size_t stage_elems = k * n_x + (3 * k + 1) * HC + HC + 3 * k;      // x̂ coefficients, hiding polynomials, the opening's hiding quotient: staged through pinned memory
  const size_t pin_need = (k * n_h + stage_elems) * 32 + 4096;
  if (c->prover_pin_cap < pin_need) {
    if (c->prover_pin) { HIPCHK(hipStreamSynchronize(s)); (void)hipHostFree(c->prover_pin); c->prover_pin = nullptr; c->prover_pin_cap = 0; }
    HIPCHK(hipHostMalloc(&c->prover_pin, pin_need + pin_need / 8, hipHostMallocDefault)); c->prover_pin_cap = pin_need + pin_need / 8;
  }
  pin = (char*)c->prover_pin; stage = pin + k * n_h * 32; pin_small = stage + stage_elems * 32;      // 4 KB for small read-backs
  one = HFr::one(); neg1 = HFr::neg(one); std::memcpy(r2.l, host::HParams<4>::R2, 32);
  // randomness layout (oracle/varuna_ref.py randomness_layout)
  lay_mask = 3 * k; lay_blind = 3 * k + 3 * n_h; lay_blind_mask = lay_blind + 3 * HC * k;
  return ALEO_MI355X_OK;
}

int32_t Prover::first_round(const void* const* assignments) {
  // ---- round 1 ------------------------------------------------------------------------------------------------------------------------
  TAKE(zH, k * n_h) TAKE(ev, 3 * k * n_h) TAKE(xh, k * n_h) TAKE_M(xp, k * n_x) TAKE_M(wit, 3 * k * L) TAKE_M(mask, 3 * n_h) TAKE(bl, (3 * k + 1) * HC)
  x_poly.assign(k, {}); std::vector<uint8_t> x_bytes(k * n_x * 32, 0);
  uint32_t lg_x = 0; while ((1ull << lg_x) < n_x) ++lg_x;
  const HFr gx_inv = HFr::inv(domain_gen(n_x)), nx_inv = inv_pow2(lg_x);
  const uint32_t* pos = (const uint32_t*)ix.positions;
  const bool host_layout = ix.positions_device == nullptr;      // without the positions in HBM the host lays the assignment out on H (pinned staging)
  if (host_layout) std::memset(pin, 0, k * n_h * 32);
  else for (size_t v = 0; v < ix.n_vars; ++v) if (pos[v] >= n_h) { g_last_error = "varuna_prove: variable position outside H"; return ALEO_MI355X_ERR_BAD_ARG; }
  for (size_t i = 0; i < k; ++i) {
    const uint8_t* z = (const uint8_t*)assignments[i];
    if (host_layout)
      for (size_t v = 0; v < ix.n_vars; ++v) {
        if (pos[v] >= n_h) { g_last_error = "varuna_prove: variable position outside H"; return ALEO_MI355X_ERR_BAD_ARG; }
        std::memcpy(pin + (i * n_h + pos[v]) * 32, z + v * 32, 32);
      }
    std::vector<HFr> xe(n_x, HFr::zero());
    for (size_t j = 0; j < ix.n_public; ++j) { HFr v; std::memcpy(v.l, z + j * 32, 32); if (HFr::geq_p(v.l)) { g_last_error = "varuna_prove: assignment not canonical"; return ALEO_MI355X_ERR_BAD_ARG; } std::memcpy(&x_bytes[(i * n_x + j) * 32], v.l, 32); xe[j] = HFr::to_mont(v); }
    x_poly[i].assign(n_x, HFr::zero());                    // inverse DFT over X, O(|X|^2): |X| is the (padded) number of public inputs
    HFr wa = one;                                          // gx_inv^a
    for (size_t a = 0; a < n_x; ++a) {
      HFr acc = HFr::zero(), w = one;
      for (size_t j = 0; j < n_x; ++j) { acc = HFr::add(acc, HFr::mul(xe[j], w)); w = HFr::mul(w, wa); }
      x_poly[i][a] = HFr::mul(acc, nx_inv); wa = HFr::mul(wa, gx_inv);
    }
  }
  for (size_t i = 0; i < k; ++i) std::memcpy(stage + i * n_x * 32, x_poly[i].data(), n_x * 32);
  if (host_layout) {
    HIPCHK(hipMemcpyAsync(zH, pin, k * n_h * 32, hipMemcpyHostToDevice, s));
    RC(fr_lin(c, zH, k * n_h, nullptr, r2.l, zH, nullptr, nullptr, s));                   // canonical -> Montgomery
  } else {                                                                                  // upload in variable order; scatter + Montgomery form on the device
    TAKE(zraw, k * ix.n_vars)
    HIPCHK(hipMemsetAsync(zH, 0, k * n_h * 32, s));
    for (size_t i = 0; i < k; ++i) {
      HIPCHK(hipMemcpyAsync(zraw + i * ix.n_vars * 32, assignments[i], ix.n_vars * 32, hipMemcpyHostToDevice, s));
      RC(fr_scatter_to_mont(c, zH + i * n_h * 32, zraw + i * ix.n_vars * 32, ix.positions_device, ix.n_vars, s));
    }
  }
  HIPCHK(hipMemcpyAsync(xp, stage, k * n_x * 32, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemsetAsync(xh, 0, k * n_h * 32, s));
  for (size_t i = 0; i < k; ++i) {
    char* e0 = ev + 3 * i * n_h * 32; char* z_i = zH + i * n_h * 32; char* xh_i = xh + i * n_h * 32;
    RC(fr_spmv(c, e0 + n_h * 32, ix.a_row_ptr, ix.a_col, ix.a_val, z_i, n_h, s));
    RC(fr_spmv(c, e0 + 2 * n_h * 32, ix.b_row_ptr, ix.b_col, ix.b_val, z_i, n_h, s));
    HIPCHK(hipMemcpyAsync(xh_i, xp + i * n_x * 32, n_x * 32, hipMemcpyDeviceToDevice, s));
    RC(ntt_run(c, xh_i, lg_h, 1, 0, 0, 0, s));
    RC(fr_vec_op(c, e0, z_i, xh_i, n_h, 2, s));                                             // z − x̂ on H
    RC(fr_vec_op(c, e0, e0, ix.vx_inv, n_h, 0, s));                                         // / v_X off X, 0 on X
  }
  const bool lagrange = ix.lagrange_offset != 0;          // KZG10::commit_lagrange for w, z_a, z_b: commit the evaluations (kept here) against L_i(tau) G
  char* evals_h = nullptr; char* rho_dev = nullptr;
  if (lagrange) {
    evals_h = ar.take(3 * k * n_h); rho_dev = ar.take(3 * k);
    if (!evals_h || !rho_dev) { g_last_error = "varuna_prove: workspace accounting"; return ALEO_MI355X_ERR_HIP; }
    HIPCHK(hipMemcpyAsync(evals_h, ev, 3 * k * n_h * 32, hipMemcpyDeviceToDevice, s));
  }
  RC(ntt_run(c, ev, lg_h, 3 * k, 0, 1, 0, s));
  blind.assign((3 * k + 1) * HC, HFr::zero());
  {
    HFr rho[24];                                                                            // rho_w, rho_a, rho_b of instance q / 3
    for (size_t q = 0; q < 3 * k; ++q) {
      rho[q] = random_fr(seed, q);
      for (size_t j = 0; j < HC; ++j) blind[q * HC + j] = random_fr(seed, lay_blind + HC * q + j);
    }
    RC(fr_blind_rows(c, wit, ev, n_h, 3 * k, rho, s));                                      // + rho (X^|H| − 1), all 3k polynomials in one launch
    if (lagrange) { char* st = stage + (k * n_x + (3 * k + 1) * HC + HC) * 32; std::memcpy(st, rho, 3 * k * 32); HIPCHK(hipMemcpyAsync(rho_dev, st, 3 * k * 32, hipMemcpyHostToDevice, s)); }
  }
  for (size_t j = 0; j < HC; ++j) blind[3 * k * HC + j] = random_fr(seed, lay_blind_mask + j);
  RC(fr_random(c, mask, 3 * n_h, seed, lay_mask, 1, s));
  RC(fr_lin(c, mask, 1, nullptr, neg1.l, mask + n_h * 32, neg1.l, mask + 2 * n_h * 32, s));   // sum over H = |H| (m_0 + m_|H| + m_2|H|) = 0
  std::memcpy(stage + k * n_x * 32, blind.data(), blind.size() * 32);
  HIPCHK(hipMemcpyAsync(bl, stage + k * n_x * 32, blind.size() * 32, hipMemcpyHostToDevice, s));
  wit_aff.assign(104 * (3 * k + 1), 0); comp.assign(48 * 8, 0);
  {
    // with the evaluations against the Lagrange powers AND a narrow-window table over [hiding powers | Lagrange powers | v_H G] the 3k witness
    // commitments are one sparse chain (their scalars are mostly 0 / 1), the mask (uniform coefficients against the monomial powers) another
    const bool split = lagrange && pb.range.d && pb.range_off <= ix.gamma_offset && ix.lagrange_offset + n_h + 1 <= pb.range_off + pb.range.cover &&
                       ix.gamma_offset + HC <= pb.range_off + pb.range.cover;
    std::vector<MsmSeg> sg, sm;
    for (size_t q = 0; q <= 3 * k; ++q) {
      std::vector<MsmSeg>& dst = (split && q == 3 * k) ? sm : sg;
      MsmSeg a; a.out = (split && q == 3 * k) ? 0u : (uint32_t)q;
      if (q < 3 * k && lagrange) {                                                            // sum_i evals_i L_i(tau) G + rho v_H(tau) G
        a.d_ptr = evals_h + q * n_h * 32; a.len = n_h; a.off = ix.lagrange_offset; dst.push_back(a);
        MsmSeg v; v.d_ptr = rho_dev + q * 32; v.len = 1; v.off = ix.lagrange_offset + n_h; v.out = a.out; dst.push_back(v);
      } else { a.d_ptr = q < 3 * k ? wit + q * L * 32 : mask; a.len = q < 3 * k ? L : 3 * n_h; a.off = 0; dst.push_back(a); }
      MsmSeg b; b.d_ptr = bl + q * HC * 32; b.len = HC; b.off = ix.gamma_offset; b.out = a.out; dst.push_back(b);
    }
    if (split) { RC(commit(c, pb, sg, (uint32_t)(3 * k), wit_aff.data(), s, true)); RC(commit(c, pb, sm, 1, wit_aff.data() + 104 * 3 * k, s)); }
    else RC(commit(c, pb, sg, (uint32_t)(3 * k + 1), wit_aff.data(), s));
  }
  std::vector<uint8_t> c1(48 * (3 * k + 1));
  RC(aleo_mi355x_g1_compress(c1.data(), wit_aff.data(), 3 * k + 1));
  tr.absorb(ix.vk_bytes, ix.vk_len); tr.absorb(x_bytes.data(), x_bytes.size()); tr.absorb(c1.data(), c1.size());
  alpha = tr.challenge("alpha", 5); eta_b = tr.challenge("eta_b", 5); eta_c = tr.challenge("eta_c", 5);
  comb.assign(k, one);
  for (size_t i = 1; i < k; ++i) { char lab[12] = "combiner"; uint32_t ii = (uint32_t)i; std::memcpy(lab + 8, &ii, 4); comb[i] = tr.challenge(lab, 12); }
  t_mark[1] = now_ms();
  return ALEO_MI355X_OK;
}

int32_t Prover::second_round() {
  // ---- round 2: the first sumcheck --------------------------------------------------------------------------------------------------------
  vh_alpha = vanish(n_h, alpha);
  if (vh_alpha.is_zero()) { g_last_error = "varuna_prove: alpha landed in H"; return ALEO_MI355X_ERR_HIP; }
  TAKE_M(ext, 3 * n_h) TAKE(rt, 2 * n_h) TAKE(E, (2 + 3 * k) * n4) TAKE(Q, n4) TAKE_M(h1, 2 * n_h) TAKE_M(g1, n_h)
  {
    const HFr first = HFr::pow_u64(alpha, n_h - 1), ratio = HFr::inv(alpha);
    RC(fr_powers(c, rt, n_h, first.l, ratio.l, s));                                          // r(alpha, X) = sum_k alpha^(|H|-1-k) X^k
  }
  HIPCHK(hipMemcpyAsync(ext, rt, n_h * 32, hipMemcpyDeviceToDevice, s));
  RC(ntt_run(c, ext, lg_h, 1, 0, 0, 0, s));                                                 // v_H(alpha) / (alpha − h) on H: no inversion on the device
  RC(fr_lin(c, ext + n_h * 32, n_h, nullptr, eta_b.l, ext, nullptr, nullptr, s));
  RC(fr_lin(c, ext + 2 * n_h * 32, n_h, nullptr, eta_c.l, ext, nullptr, nullptr, s));
  RC(fr_spmv(c, rt + n_h * 32, ix.t_row_ptr, ix.t_col, ix.t_val, ext, n_h, s));
  RC(ntt_run(c, rt + n_h * 32, lg_h, 1, 0, 1, 0, s));                                       // t(X)
  HIPCHK(hipMemsetAsync(E, 0, 2 * n4 * 32, s));                                             // r, t: |H| coefficients each, zero padded to 4|H|
  HIPCHK(hipMemcpyAsync(E, rt, n_h * 32, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(E + n4 * 32, rt + n_h * 32, n_h * 32, hipMemcpyDeviceToDevice, s));
  RC(ahp_sumcheck_operands(c, E + 2 * n4 * 32, wit, xp, n_h, n_x, k, s));                    // ẑ_i = w_i (X^|X| − 1) + x̂_i, z_a,i, z_b,i — every row written in full
  RC(ntt_run(c, E, lg_h + 2, 2 + 3 * k, 0, 0, 0, s));
  for (size_t i = 0; i < k; ++i) {
    char* e_z = E + (2 + 3 * i) * n4 * 32;
    RC(ahp_first_sumcheck(c, e_z + n4 * 32, n4, E, e_z + n4 * 32, e_z + 2 * n4 * 32, E + n4 * 32, e_z, eta_b.l, eta_c.l, s));
  }
  char* q1 = E + 3 * n4 * 32;
  if (k > 1) {
    const void* terms[8]; size_t lens[8]; HFr co[8];
    for (size_t i = 0; i < k; ++i) { terms[i] = E + (3 + 3 * i) * n4 * 32; lens[i] = n4; co[i] = comb[i]; }
    RC(fr_lincomb(c, Q, n4, nullptr, terms, lens, co, k, s)); q1 = Q;
  }
  RC(ntt_run(c, q1, lg_h + 2, 1, 0, 1, 0, s));
  RC(fr_vec_op(c, q1, q1, mask, 3 * n_h, 1, s));                                            // q_1 = h_1 (X^|H| − 1) + X g_1, degree < 3|H|
  HIPCHK(hipMemcpyAsync(h1 + n_h * 32, q1 + 2 * n_h * 32, n_h * 32, hipMemcpyDeviceToDevice, s));   // quotient blocks: p2, p1 + p2; remainder p0 + p1 + p2
  RC(fr_vec_op(c, h1, q1 + n_h * 32, q1 + 2 * n_h * 32, n_h, 1, s));
  RC(fr_vec_op(c, g1, q1, h1, n_h, 1, s));
  HIPCHK(hipMemcpyAsync(pin_small + 3584, g1, 32, hipMemcpyDeviceToHost, s));               // the sum over H (remainder's constant term): read with the commitments
  {
    std::vector<MsmSeg> sg(2);
    sg[0].d_ptr = g1 + 32; sg[0].len = n_h - 1; sg[0].off = D - (n_h - 2); sg[0].out = 0;    // degree bound |H| − 2: shifted powers
    sg[1].d_ptr = h1; sg[1].len = 2 * n_h; sg[1].off = 0; sg[1].out = 1;
    RC(commit(c, pb, sg, 2, aff2, s));
  }
  {                                                                                         // commit() returned after the stream drained: the copy above has landed
    uint64_t sum[4]; std::memcpy(sum, pin_small + 3584, 32);
    if (sum[0] | sum[1] | sum[2] | sum[3]) { g_last_error = "varuna_prove: the assignment does not satisfy the circuit (first sumcheck: the sum over H is not zero)"; return ALEO_MI355X_ERR_UNSATISFIED; }
  }
  RC(aleo_mi355x_g1_compress(comp.data(), aff2, 2)); tr.absorb(comp.data(), 96);
  beta = tr.challenge("beta", 4);
  t_mark[2] = now_ms();
  return ALEO_MI355X_OK;
}

int32_t Prover::third_round() {
  // ---- round 3: three rational sumchecks over K ----------------------------------------------------------------------------------------------
  vh_beta = vanish(n_h, beta);
  if (vh_beta.is_zero()) { g_last_error = "varuna_prove: beta landed in H"; return ALEO_MI355X_ERR_HIP; }
  vv = HFr::mul(vh_alpha, vh_beta);
  TAKE_M(f, k_sum) TAKE(rb, n_h)                                                              // f_M at element ko[M], |K_M| values
  {
    const HFr first = HFr::pow_u64(beta, n_h - 1), ratio = HFr::inv(beta);
    RC(fr_powers(c, rb, n_h, first.l, ratio.l, s));
  }
  RC(ntt_run(c, rb, lg_h, 1, 0, 0, 0, s));
  for (size_t m = 0; m < 3; ++m) {                                                           // f_M = val u_H(alpha, row) u_H(beta, col) on K_M: two gathers
    const uint32_t* ri = (const uint32_t*)ix.k_idx + 2 * ko[m];
    RC(fr_gather_mul(c, f + ko[m] * 32, nk[m], (const char*)ix.k_evals + (4 * ko[m] + 2 * nk[m]) * 32, ext, ri, rb, ri + nk[m], s));
  }
  // maximal runs of consecutive matrices with equal domains share batched transforms (and, in round 4, one numerator pass)
  nrun = 0;
  for (size_t m = 0; m < 3;) { size_t cnt = 1; while (m + cnt < 3 && nk[m + cnt] == nk[m]) ++cnt; run0[nrun] = m; runc[nrun++] = cnt; m += cnt; }
  for (size_t r = 0; r < nrun; ++r) RC(ntt_run(c, f + ko[run0[r]] * 32, lg_km[run0[r]], runc[r], 0, 1, 0, s));
  for (size_t m = 0; m < 3; ++m) HIPCHK(hipMemcpyAsync(pin_small + 32 * m, f + ko[m] * 32, 32, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  uint8_t sig_bytes[96];
  for (size_t m = 0; m < 3; ++m) { HFr v; std::memcpy(v.l, pin_small + 32 * m, 32); sigma[m] = HFr::mul(v, fr_u64(nk[m])); fr_bytes(sig_bytes + 32 * m, sigma[m]); }
  {
    std::vector<MsmSeg> sg(3);
    for (size_t m = 0; m < 3; ++m) { sg[m].d_ptr = f + (ko[m] + 1) * 32; sg[m].len = nk[m] - 1; sg[m].off = D - (nk[m] - 2); sg[m].out = (uint32_t)m; }
    RC(commit(c, pb, sg, 3, aff3, s));
  }
  RC(aleo_mi355x_g1_compress(comp.data(), aff3, 3));
  { uint8_t b[96 + 144]; std::memcpy(b, sig_bytes, 96); std::memcpy(b + 96, comp.data(), 144); tr.absorb(b, sizeof b); }
  delta[0] = one; delta[1] = tr.challenge("delta_b", 7); delta[2] = tr.challenge("delta_c", 7);
  t_mark[3] = now_ms();
  return ALEO_MI355X_OK;
}

int32_t Prover::fourth_round() {
  // ---- round 4 ----------------------------------------------------------------------------------------------------------------------------------
  TAKE(F, 2 * k_sum) TAKE(B, 2 * k_sum) TAKE_M(h2, n_k)                                          // per matrix on its own domain of size 2|K_M|
  HIPCHK(hipMemsetAsync(F, 0, 2 * k_sum * 32, s));
  {
    const void* terms[3]; size_t lens[3]; HFr co[3];
    for (size_t r = 0; r < nrun; ++r) {
      const size_t m0 = run0[r], cnt = runc[r], n2 = 2 * nk[m0]; char* Fr0 = F + 2 * ko[m0] * 32; char* Br = B + 2 * ko[m0] * 32;
      HFr consts[7] = {HFr::zero(), HFr::zero(), HFr::zero(), HFr::mul(alpha, beta), HFr::neg(alpha), HFr::neg(beta), vv};
      const void* idx[3] = {nullptr, nullptr, nullptr}; const void* ff[3] = {nullptr, nullptr, nullptr};
      for (size_t j = 0; j < cnt; ++j) {
        const size_t m = m0 + j;
        HIPCHK(hipMemcpyAsync(F + 2 * ko[m] * 32, f + ko[m] * 32, nk[m] * 32, hipMemcpyDeviceToDevice, s));
        idx[j] = (const char*)ix.k2_evals + 8 * ko[m] * 32; ff[j] = F + 2 * ko[m] * 32; consts[j] = delta[m];
      }
      RC(ntt_run(c, Fr0, lg_km[m0] + 1, cnt, 0, 0, 0, s));
      RC(ahp_matrix_sumcheck(c, Br, n2, idx, n2, ff, consts, s));                                // sum over the run of delta_M (vv val_M − b_M f_M) = h (X^|K| − 1)
      RC(ntt_run(c, Br, lg_km[m0] + 1, 1, 0, 1, 0, s));
      terms[r] = Br + nk[m0] * 32; lens[r] = nk[m0]; co[r] = one;                              // its upper half
    }
    RC(fr_lincomb(c, h2, n_k, nullptr, terms, lens, co, nrun, s));                                 // h_2 = sum_M delta_M h_M
  }
  {
    std::vector<MsmSeg> sg(1); sg[0].d_ptr = h2; sg[0].len = n_k; sg[0].off = 0; sg[0].out = 0;
    RC(commit(c, pb, sg, 1, aff4, s));
  }
  RC(aleo_mi355x_g1_compress(comp.data(), aff4, 1)); tr.absorb(comp.data(), 48);
  gamma = tr.challenge("gamma", 5);
  t_mark[4] = now_ms();
  return ALEO_MI355X_OK;
}

int32_t Prover::open() {
  // ---- evaluations -------------------------------------------------------------------------------------------------------------------------------
  TAKE(evd, k + 8) TAKE(pbeta, 3 * n_h) TAKE(wq, 3 * n_h) TAKE(blq, HC) TAKE(pg, n_k) TAKE(gq, n_k)
  {
    const void* polys[12]; size_t lens[12]; HFr pts[12];
    for (size_t i = 0; i < k; ++i) { polys[i] = wit + (3 * i + 2) * L * 32; lens[i] = L; pts[i] = beta; }
    polys[k] = g1 + 32; lens[k] = n_h - 1; pts[k] = beta;
    for (size_t m = 0; m < 3; ++m) { polys[k + 1 + m] = f + (ko[m] + 1) * 32; lens[k + 1 + m] = nk[m] - 1; pts[k + 1 + m] = gamma; }
    RC(fr_eval_batch(c, evd, polys, lens, pts, k + 4, s));
  }
  HIPCHK(hipMemcpyAsync(pin_small, evd, (k + 4) * 32, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  evals.assign(k + 4, HFr::zero()); std::vector<uint8_t> ev_bytes((k + 4) * 32);
  for (size_t i = 0; i < k + 4; ++i) { std::memcpy(evals[i].l, pin_small + 32 * i, 32); fr_bytes(&ev_bytes[32 * i], evals[i]); }
  tr.absorb(ev_bytes.data(), ev_bytes.size());
  const HFr xi = tr.challenge("xi", 2);
  const HFr g1_beta = evals[k], ga = evals[k + 1], gb = evals[k + 2], gc = evals[k + 3];
  // ---- the linear combination of the first sumcheck, opened at beta together with g_1 and the z_b,i -----------------------------------------------
  HFr inv4[4] = {HFr::sub(alpha, beta), vanish(nk[0], gamma), vanish(nk[1], gamma), vanish(nk[2], gamma)};      // one inversion for the four the openings need
  for (int m = 1; m < 4; ++m) if (inv4[m].is_zero()) { g_last_error = "varuna_prove: gamma landed in K"; return ALEO_MI355X_ERR_HIP; }
  if (inv4[0].is_zero()) { g_last_error = "varuna_prove: alpha equals beta"; return ALEO_MI355X_ERR_HIP; }
  batch_inverse(inv4, 4);
  const HFr r_ab = HFr::mul(HFr::sub(vh_alpha, vh_beta), inv4[0]);
  const HFr t_beta = HFr::add(sigma[0], HFr::add(HFr::mul(eta_b, sigma[1]), HFr::mul(eta_c, sigma[2])));
  const HFr xl = HFr::pow_u64(xi, k + 1), vx_beta = vanish(n_x, beta);
  HFr cst = HFr::neg(HFr::mul(beta, g1_beta));
  HFr blw[3];                                                                     // blw: (bl(X) − bl(beta)) / (X − beta), uploaded below
  {
    const void* terms[28]; size_t lens[28]; HFr co[28]; size_t nt = 0;
    terms[nt] = mask; lens[nt] = 3 * n_h; co[nt++] = xl;
    terms[nt] = h1; lens[nt] = 2 * n_h; co[nt++] = HFr::neg(HFr::mul(xl, vh_beta));
    terms[nt] = g1 + 32; lens[nt] = n_h - 1; co[nt++] = one;
    HFr blc[3] = {HFr::zero(), HFr::zero(), HFr::zero()};
    auto axpy = [&](const HFr& coef, const HFr* src) { for (size_t j = 0; j < HC; ++j) blc[j] = HFr::add(blc[j], HFr::mul(coef, src[j])); };
    axpy(xl, &blind[3 * k * HC]);
    HFr xpow = xi;                                                                           // xi^(1+i)
    for (size_t i = 0; i < k; ++i) {
      const HFr x_beta = horner(x_poly[i], beta), zb = evals[i], ci = comb[i];
      const HFr k_za = HFr::mul(HFr::mul(xl, ci), HFr::mul(r_ab, HFr::add(one, HFr::mul(eta_c, zb))));
      const HFr k_w = HFr::neg(HFr::mul(HFr::mul(xl, ci), HFr::mul(t_beta, vx_beta)));
      cst = HFr::add(cst, HFr::mul(ci, HFr::sub(HFr::mul(HFr::mul(r_ab, eta_b), zb), HFr::mul(t_beta, x_beta))));
      terms[nt] = wit + (3 * i + 1) * L * 32; lens[nt] = L; co[nt++] = k_za;
      terms[nt] = wit + (3 * i) * L * 32; lens[nt] = L; co[nt++] = k_w;
      terms[nt] = wit + (3 * i + 2) * L * 32; lens[nt] = L; co[nt++] = xpow;
      axpy(k_w, &blind[(3 * i) * HC]); axpy(k_za, &blind[(3 * i + 1) * HC]); axpy(xpow, &blind[(3 * i + 2) * HC]);
      xpow = HFr::mul(xpow, xi);
    }
    const HFr c0 = HFr::mul(xl, cst);
    RC(fr_lincomb(c, pbeta, 3 * n_h, c0.l, terms, lens, co, nt, s));
    random_v = HFr::add(blc[0], HFr::mul(beta, HFr::add(blc[1], HFr::mul(beta, blc[2]))));
    blw[1] = blc[2]; blw[0] = HFr::add(blc[1], HFr::mul(beta, blc[2])); blw[2] = HFr::zero();
    char* st = stage + (k * n_x + (3 * k + 1) * HC) * 32; std::memcpy(st, blw, HC * 32);
    HIPCHK(hipMemcpyAsync(blq, st, HC * 32, hipMemcpyHostToDevice, s));
  }
  RC(fr_divide_by_linear(c, wq, evd + (k + 5) * 32, pbeta, 3 * n_h, beta.l, s));
  // ---- the linear combination of the second sumcheck, opened at gamma together with g_a, g_b, g_c --------------------------------------------------
  {
    const HFr xi2 = HFr::sqr(xi), xi3 = HFr::mul(xi2, xi), vk_gamma = vanish(n_k, gamma);
    const void* terms[20]; size_t lens[20]; HFr co[20]; size_t nt = 0; HFr cg = HFr::zero();
    const HFr gk[3] = {ga, gb, gc};
    for (size_t m = 0; m < 3; ++m) {
      const HFr fm = HFr::add(HFr::mul(gamma, gk[m]), HFr::mul(sigma[m], inv_pow2(lg_km[m])));
      const HFr d = HFr::mul(HFr::mul(delta[m], xi3), HFr::mul(vk_gamma, inv4[1 + m]));     // selector v_K / v_{K_M} at gamma
      const HFr dfm = HFr::mul(d, fm);
      const HFr cf[4] = {HFr::mul(dfm, beta), HFr::mul(dfm, alpha), HFr::mul(d, vv), HFr::neg(dfm)};      // row, col, val, row_col
      for (int j = 0; j < 4; ++j) { terms[nt] = (const char*)ix.k_polys + (4 * ko[m] + (size_t)j * nk[m]) * 32; lens[nt] = nk[m]; co[nt++] = cf[j]; }
      cg = HFr::sub(cg, HFr::mul(HFr::mul(dfm, alpha), beta));
    }
    terms[nt] = h2; lens[nt] = n_k; co[nt++] = HFr::neg(HFr::mul(xi3, vk_gamma));
    const HFr gco[3] = {one, xi, xi2};
    for (size_t m = 0; m < 3; ++m) { terms[nt] = f + (ko[m] + 1) * 32; lens[nt] = nk[m] - 1; co[nt++] = gco[m]; }
    RC(fr_lincomb(c, pg, n_k, cg.l, terms, lens, co, nt, s));
  }
  RC(fr_divide_by_linear(c, gq, evd + (k + 6) * 32, pg, n_k, gamma.l, s));
  {
    std::vector<MsmSeg> sg(3);
    sg[0].d_ptr = wq; sg[0].len = 3 * n_h - 1; sg[0].off = 0; sg[0].out = 0;
    sg[1].d_ptr = blq; sg[1].len = HC - 1; sg[1].off = ix.gamma_offset; sg[1].out = 0;
    sg[2].d_ptr = gq; sg[2].len = n_k - 1; sg[2].off = 0; sg[2].out = 1;
    RC(commit(c, pb, sg, 2, aff5, s));                                                       // both witness commitments in one call
  }
  t_mark[5] = now_ms();
  return ALEO_MI355X_OK;
}

int32_t Prover::write(uint8_t* out, size_t* out_len) {
  // ---- the proof in upstream's layout ---------------------------------------------------------------------------------------------------------------
  aleo_mi355x_proof_parts parts{}; uint64_t batch = k;
  uint8_t has_v[2] = {1, 0}; HFr rv[2] = {random_v, HFr::zero()};
  parts.batch_sizes = &batch; parts.n_circuits = 1; parts.witness_commitments = wit_aff.data(); parts.mask_poly = wit_aff.data() + 104 * 3 * k;
  parts.g_1 = aff2; parts.h_1 = aff2 + 104; parts.g_abc = aff3; parts.h_2 = aff4;
  parts.evaluations = evals.data(); parts.n_evaluations = k + 4; parts.sums = sigma;
  parts.opening_points = aff5; parts.opening_random_v = rv; parts.opening_has_v = has_v; parts.n_openings = 2;
  RC(aleo_mi355x_proof_to_bytes(out, out_len, &parts));
  for (int i = 0; i < 5; ++i) g_varuna_timing[i] = t_mark[i + 1] - t_mark[i];
  g_varuna_timing[5] = t_mark[5] - t_mark[0];
  return ALEO_MI355X_OK;
}

int32_t varuna_prove(Ctx* c, const PinnedBases& pb, const aleo_mi355x_varuna_index& ix, const void* const* assignments, size_t k, uint64_t seed,
                     uint8_t* out, size_t* out_len) {
  g_varuna_timing[6] = g_varuna_timing[7] = 0;
  Prover p(c, pb, ix, k, seed);
  RC(p.setup(assignments)); RC(p.first_round(assignments)); RC(p.second_round()); RC(p.third_round()); RC(p.fourth_round()); RC(p.open());
  return p.write(out, out_len);
}

}  // namespace aleo_mi355x

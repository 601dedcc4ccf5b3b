// api.hip — the C ABI of libaleo_mi355x.so (include/aleo_mi355x.h): argument checking, per-device context,
// host<->HBM staging, and the host-side tails.  Kernels live in msm.hip / ntt.hip.
#include "ctx.h"
#include "host_field.hpp"
#include <cstdlib>
#include <cstring>
#include <memory>

namespace aleo_mi355x {

thread_local std::string g_last_error;

static std::mutex g_ctx_mu;
static std::map<int, Ctx*> g_ctxs;

int32_t ensure_host_pinned(Ctx* c, size_t bytes) {
  if (bytes <= c->h_pinned_cap) return ALEO_MI355X_OK;
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  c->h_pinned = nullptr; c->h_pinned_cap = 0;
  size_t want = bytes < 65536 ? 65536 : bytes;
  HIPCHK(hipHostMalloc(&c->h_pinned, want, hipHostMallocDefault));
  c->h_pinned_cap = want; return ALEO_MI355X_OK;
}

static int32_t init_device(int device, Ctx** out) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { g_last_error = "no HIP device visible"; return ALEO_MI355X_ERR_NO_DEVICE; }
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) { g_last_error = "hipGetDevice failed"; return ALEO_MI355X_ERR_NO_DEVICE; } }
  if (device >= count) { g_last_error = "device index out of range"; return ALEO_MI355X_ERR_BAD_ARG; }
  auto it = g_ctxs.find(device);
  if (it != g_ctxs.end()) { *out = it->second; return ALEO_MI355X_OK; }
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_last_error = std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only";
    return ALEO_MI355X_ERR_NO_DEVICE;
  }
  std::unique_ptr<Ctx> c(new Ctx());
  c->device = device;
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  for (auto& e : c->ev) HIPCHK(hipEventCreate(&e));
  *out = c.get();
  g_ctxs[device] = c.release();
  return ALEO_MI355X_OK;
}

int32_t get_ctx(Ctx** out) {
  int device = -1;
  if (hipGetDevice(&device) != hipSuccess) { g_last_error = "no HIP device visible"; return ALEO_MI355X_ERR_NO_DEVICE; }
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto it = g_ctxs.find(device);
    if (it != g_ctxs.end()) { *out = it->second; return ALEO_MI355X_OK; }
  }
  return init_device(device, out);
}

static int32_t pin_locked(Ctx* c, const void* bases, size_t stride, size_t n, uint64_t* handle) {
  PinnedBases pb; pb.n = n;
  size_t bytes = (n ? n : 1) * 96;
  HIPCHK(hipMalloc(&pb.d_xy, bytes));
  bool any_inf = false;
  if (stride == 96) {
    HIPCHK(hipMemcpy(pb.d_xy, bases, n * 96, hipMemcpyHostToDevice));
  } else {
    // strip the infinity byte + padding of snarkVM's 104-byte Affine on the host, one pass
    std::vector<uint8_t> packed(bytes), inf(n ? n : 1, 0);
    const uint8_t* src = (const uint8_t*)bases;
    for (size_t i = 0; i < n; ++i) {
      std::memcpy(&packed[i * 96], src + i * stride, 96);
      if (src[i * stride + 96]) { inf[i] = 1; any_inf = true; }
    }
    HIPCHK(hipMemcpy(pb.d_xy, packed.data(), n * 96, hipMemcpyHostToDevice));
    if (any_inf) {
      HIPCHK(hipMalloc((void**)&pb.d_inf, n));
      HIPCHK(hipMemcpy(pb.d_inf, inf.data(), n, hipMemcpyHostToDevice));
    }
  }
  uint64_t h = c->next_handle++;
  c->bases[h] = pb; *handle = h;
  return ALEO_MI355X_OK;
}

static int32_t unpin_locked(Ctx* c, uint64_t handle) {
  auto it = c->bases.find(handle);
  if (it == c->bases.end()) { g_last_error = "unknown bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
  if (it->second.d_xy) (void)hipFree(it->second.d_xy);
  if (it->second.d_inf) (void)hipFree(it->second.d_inf);
  if (it->second.d_pre) (void)hipFree(it->second.d_pre);
  c->bases.erase(it);
  return ALEO_MI355X_OK;
}

static int32_t msm_host_scalars_locked(Ctx* c, void* out, const PinnedBases& pb, const void* scalars, size_t n, bool mont) {
  int32_t rc;
  if ((rc = c->scalars_stage.reserve((n ? n : 1) * 32))) return rc;
  if (n) HIPCHK(hipMemcpyAsync(c->scalars_stage.p, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
  return msm_run(c, (uint64_t*)out, pb, c->scalars_stage.p, n, mont, c->stream);
}

// ---- SRS cache for the one-shot entry point ----------------------------------------------------------
static uint64_t hash96(const uint8_t* p) {      // FNV-1a over the 96 coordinate bytes of one point
  uint64_t h = 1469598103934665603ull;
  for (int i = 0; i < 96; ++i) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}
static constexpr size_t SRS_SAMPLES = 256, SRS_CACHE_ENTRIES = 8, SRS_MIN_N = 1024;

static bool srs_cache_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = std::getenv("ALEO_MI355X_SRS_CACHE"); on = (e && e[0] == '0') ? 0 : 1; }
  return on == 1;
}
// Returns the cached pinned set for (bases, stride) that covers n points, or nullptr.
static SrsCacheEntry* srs_lookup(Ctx* c, const void* bases, size_t stride, size_t n) {
  for (auto& e : c->srs_cache) {
    if (e.host_ptr != bases || e.stride != stride || e.n < n) continue;
    bool ok = true; size_t checked = 0;
    for (auto& sm : e.samples) {                 // only samples inside the caller's slice may be read
      if (sm.first >= n) continue;
      ++checked;
      if (hash96((const uint8_t*)bases + sm.first * stride) != sm.second) { ok = false; break; }
    }
    if (ok && checked) return &e;
  }
  return nullptr;
}
static int32_t srs_insert(Ctx* c, const void* bases, size_t stride, size_t n, SrsCacheEntry** out) {
  // drop stale entries for the same pointer, then the least recently used one if the cache is full
  for (size_t i = 0; i < c->srs_cache.size();) {
    if (c->srs_cache[i].host_ptr == bases) { unpin_locked(c, c->srs_cache[i].handle); c->srs_cache.erase(c->srs_cache.begin() + i); } else ++i;
  }
  if (c->srs_cache.size() >= SRS_CACHE_ENTRIES) {
    size_t lru = 0; for (size_t i = 1; i < c->srs_cache.size(); ++i) if (c->srs_cache[i].last_use < c->srs_cache[lru].last_use) lru = i;
    unpin_locked(c, c->srs_cache[lru].handle); c->srs_cache.erase(c->srs_cache.begin() + lru);
  }
  SrsCacheEntry e; e.host_ptr = bases; e.stride = stride; e.n = n;
  int32_t rc = pin_locked(c, bases, stride, n, &e.handle);
  if (rc) return rc;
  // dense samples at the front (every prefix request can be checked), sparse ones over the rest
  for (size_t k = 0; k < SRS_SAMPLES; ++k) {
    size_t idx = k < 32 ? k : (size_t)((double)(k - 31) / (SRS_SAMPLES - 31) * (n - 1));
    if (idx >= n) break;
    e.samples.emplace_back(idx, hash96((const uint8_t*)bases + idx * stride));
  }
  c->srs_cache.push_back(e); *out = &c->srs_cache.back();
  return ALEO_MI355X_OK;
}

static void jacobian_to_affine104(void* out104, const uint64_t* jac18) {
  uint8_t* o = (uint8_t*)out104; std::memset(o, 0, 104);
  bool inf = true; for (int i = 12; i < 18; ++i) if (jac18[i]) inf = false;
  if (inf) { o[96] = 1; host::HFq one = host::HFq::one(); std::memcpy(o + 48, one.l, 48); return; }  // Affine::zero() = (0, 1, true)
  std::memcpy(o, jac18, 96);   // results are normalised: z == 1
}

}  // namespace aleo_mi355x

using namespace aleo_mi355x;

#define API_BEGIN Ctx* c = nullptr; { int32_t rc0 = get_ctx(&c); if (rc0) return rc0; } std::lock_guard<std::mutex> lk(c->mu); if (hipSetDevice(c->device) != hipSuccess) return ALEO_MI355X_ERR_HIP;

extern "C" {

int32_t aleo_mi355x_init(int32_t device) {
  try { Ctx* c = nullptr; return init_device(device, &c); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_pin(const void* bases, size_t base_stride, size_t n, uint64_t* handle) {
  try {
    if (!handle || (!bases && n) || (base_stride != 104 && base_stride != 96)) { g_last_error = "bases_pin: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    return pin_locked(c, bases, base_stride, n, handle);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_generate(const void* base104, uint64_t first, size_t n, uint64_t* handle) {
  try {
    if (!base104 || !handle) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    PinnedBases pb; int32_t rc = generate_multiples(c, base104, first, n, &pb);
    if (rc) return rc;
    uint64_t h = c->next_handle++; c->bases[h] = pb; *handle = h;
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_precompute(uint64_t handle) {
  try {
    API_BEGIN
    auto it = c->bases.find(handle);
    if (it == c->bases.end()) { g_last_error = "unknown bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
    return msm_precompute(c, &it->second);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_download(uint64_t handle, size_t offset, size_t n, void* out104) {
  try {
    if (!out104 && n) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    auto it = c->bases.find(handle);
    if (it == c->bases.end()) { g_last_error = "unknown bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
    if (offset + n > it->second.n) { g_last_error = "bases_download: range"; return ALEO_MI355X_ERR_BAD_ARG; }
    std::vector<uint8_t> xy(n * 96 + 1), inf(n + 1, 0);
    HIPCHK(hipMemcpy(xy.data(), (const char*)it->second.d_xy + offset * 96, n * 96, hipMemcpyDeviceToHost));
    if (it->second.d_inf) HIPCHK(hipMemcpy(inf.data(), it->second.d_inf + offset, n, hipMemcpyDeviceToHost));
    uint8_t* o = (uint8_t*)out104;
    for (size_t i = 0; i < n; ++i) { std::memcpy(o + i * 104, &xy[i * 96], 96); std::memset(o + i * 104 + 96, 0, 8); o[i * 104 + 96] = inf[i]; }
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_unpin(uint64_t handle) {
  try { API_BEGIN return unpin_locked(c, handle); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_msm_g1(void* out, const void* bases, size_t base_stride, const void* scalars, size_t n) {
  try {
    if (!out || ((!bases || !scalars) && n) || (base_stride != 104 && base_stride != 96)) { g_last_error = "msm_g1: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    if (n >= SRS_MIN_N && srs_cache_enabled()) {
      // KZG10::commit multiplies against prefixes of one SRS: keep it in HBM between calls (ALEO_MI355X_SRS_CACHE=0
      // turns this off; a caller that rewrites a base array in place between calls must do so)
      SrsCacheEntry* e = srs_lookup(c, bases, base_stride, n);
      int32_t rc = ALEO_MI355X_OK;
      if (!e) rc = srs_insert(c, bases, base_stride, n, &e);
      if (rc) return rc;
      e->last_use = ++c->srs_clock; e->hits++;
      PinnedBases& pb = c->bases[e->handle];
      if (e->hits == 3 && e->n >= (1u << 14) && !pb.d_pre) (void)msm_precompute(c, &pb);     // third use: worth the one-off table
      return msm_host_scalars_locked(c, out, pb, scalars, n, false);
    }
    uint64_t h = 0; int32_t rc = pin_locked(c, bases, base_stride, n, &h);
    if (rc) return rc;
    rc = msm_host_scalars_locked(c, out, c->bases[h], scalars, n, false);
    unpin_locked(c, h);
    return rc;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_msm_g1_pinned(void* out, uint64_t handle, const void* scalars, size_t n) {
  try {
    if (!out || (!scalars && n)) { g_last_error = "msm_g1_pinned: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    auto it = c->bases.find(handle);
    if (it == c->bases.end()) { g_last_error = "unknown bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
    return msm_host_scalars_locked(c, out, it->second, scalars, n, false);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_msm_g1_device(void* out, uint64_t handle, const void* d_scalars, size_t n, void* stream) {
  try {
    if (!out || (!d_scalars && n)) { g_last_error = "msm_g1_device: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    auto it = c->bases.find(handle);
    if (it == c->bases.end()) { g_last_error = "unknown bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
    return msm_run(c, (uint64_t*)out, it->second, d_scalars, n, false, stream ? (hipStream_t)stream : c->stream);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_g1_sum(void* out, const void* pts, size_t count) {
  try {
    if (!out || (!pts && count)) return ALEO_MI355X_ERR_BAD_ARG;
    host::HXYZZ t = host::HXYZZ::infinity();
    for (size_t i = 0; i < count; ++i) t = host::hadd(t, host::hfrom_jacobian((const uint64_t*)pts + 18 * i));
    host::hstore_jacobian_normalized((uint64_t*)out, t);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_kzg_commit(void* out104, uint64_t handle, const void* coeffs, size_t n) {
  try {
    if (!out104 || (!coeffs && n)) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    auto it = c->bases.find(handle);
    if (it == c->bases.end()) { g_last_error = "unknown bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
    uint64_t jac[18];
    int32_t rc = msm_host_scalars_locked(c, jac, it->second, coeffs, n, true);
    if (rc) return rc;
    jacobian_to_affine104(out104, jac);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_kzg_commit_device(void* out104, uint64_t handle, const void* d_coeffs, size_t n, void* stream) {
  try {
    if (!out104 || (!d_coeffs && n)) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    auto it = c->bases.find(handle);
    if (it == c->bases.end()) { g_last_error = "unknown bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
    uint64_t jac[18];
    int32_t rc = msm_run(c, jac, it->second, d_coeffs, n, true, stream ? (hipStream_t)stream : c->stream);
    if (rc) return rc;
    jacobian_to_affine104(out104, jac);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_kzg_commit_hiding(void* out104, uint64_t h_powers, const void* coeffs, size_t n, uint64_t h_gamma, const void* blind, size_t m) {
  try {
    if (!out104 || (!coeffs && n) || (!blind && m)) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    auto ip = c->bases.find(h_powers), ig = c->bases.find(h_gamma);
    if (ip == c->bases.end() || ig == c->bases.end()) { g_last_error = "unknown bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
    uint64_t parts[36];
    int32_t rc = msm_host_scalars_locked(c, parts, ip->second, coeffs, n, true);
    if (rc) return rc;
    if ((rc = msm_host_scalars_locked(c, parts + 18, ig->second, blind, m, true))) return rc;
    host::HXYZZ t = host::hadd(host::hfrom_jacobian(parts), host::hfrom_jacobian(parts + 18));
    uint64_t jac[18]; host::hstore_jacobian_normalized(jac, t);
    jacobian_to_affine104(out104, jac);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_vec_op_device(void* d_dst, const void* d_a, const void* d_b, size_t n, int32_t op, void* stream) {
  try {
    if ((!d_dst || !d_a || !d_b) && n) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return fr_vec_op(c, d_dst, d_a, d_b, n, op, stream ? (hipStream_t)stream : c->stream);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_batch_inverse_device(void* d_inout, size_t n, void* stream) {
  try {
    if (!d_inout && n) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return fr_batch_inverse(c, d_inout, n, stream ? (hipStream_t)stream : c->stream);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_ntt_fr(void* inout, uint32_t lg_n, int32_t order, int32_t direction, int32_t type) {
  try {
    if (!inout || lg_n > 30 || order < 0 || order > 3 || direction < 0 || direction > 1 || type < 0 || type > 1) { g_last_error = "ntt_fr: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    size_t bytes = ((size_t)1 << lg_n) * 32;
    int32_t rc; if ((rc = c->ntt_stage.reserve(bytes))) return rc;
    HIPCHK(hipMemcpyAsync(c->ntt_stage.p, inout, bytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = ntt_run(c, c->ntt_stage.p, lg_n, order, direction, type, c->stream))) return rc;
    HIPCHK(hipMemcpyAsync(inout, c->ntt_stage.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_ntt_fr_device(void* d_inout, uint32_t lg_n, int32_t order, int32_t direction, int32_t type, void* stream) {
  try {
    if (!d_inout || lg_n > 30 || order < 0 || order > 3 || direction < 0 || direction > 1 || type < 0 || type > 1) { g_last_error = "ntt_fr_device: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    return ntt_run(c, d_inout, lg_n, order, direction, type, stream ? (hipStream_t)stream : c->stream);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fq_mul(void* r, const void* a, const void* b, size_t n) {
  try { if ((!r || !a || !b) && n) return ALEO_MI355X_ERR_BAD_ARG; API_BEGIN return launch_fq_mul(c, r, a, b, n); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}
int32_t aleo_mi355x_fr_mul(void* r, const void* a, const void* b, size_t n) {
  try { if ((!r || !a || !b) && n) return ALEO_MI355X_ERR_BAD_ARG; API_BEGIN return launch_fr_mul(c, r, a, b, n); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_last_msm_timing(double* out_ms, int32_t cap) {
  try {
    if (!out_ms || cap <= 0) return 0;
    API_BEGIN
    double v[6] = {c->last_msm.total, c->last_msm.sort, c->last_msm.accum, c->last_msm.reduce, c->last_msm.host, c->last_msm.accum_kernel};
    int32_t k = cap < 6 ? cap : 6;
    for (int32_t i = 0; i < k; ++i) out_ms[i] = v[i];
    return k;
  } catch (...) { return 0; }
}

const char* aleo_mi355x_strerror(int32_t status) {
  switch (status) {
    case ALEO_MI355X_OK: return "ok";
    case ALEO_MI355X_ERR_NO_DEVICE: return "no gfx950 device available";
    case ALEO_MI355X_ERR_BAD_ARG: return "bad argument";
    case ALEO_MI355X_ERR_HIP: return "HIP runtime error";
    case ALEO_MI355X_ERR_BAD_HANDLE: return "unknown handle";
    case ALEO_MI355X_ERR_OOM: return "out of device memory";
    default: return "unknown status";
  }
}
const char* aleo_mi355x_last_error(void) { return g_last_error.c_str(); }
const char* aleo_mi355x_version(void) { return "aleo_mi355x 0.1.0 (gfx950)"; }

}  // extern "C"

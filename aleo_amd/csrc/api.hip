// api.hip — the C ABI of libaleo_mi355x.so (include/aleo_mi355x.h): argument checking, per-device context,
// host<->HBM staging, and the host-side tails.  Kernels live in msm.hip / ntt.hip.
#include "ctx.h"
#include "host_field.hpp"
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <thread>
#include <functional>
#include <vector>
#include <deque>

namespace aleo_mi355x {

thread_local std::string g_last_error;
thread_local MsmTiming g_last_msm;

bool g_host_trace_on = [] { const char* e = std::getenv("ALEO_MI355X_HOSTTRACE"); return e && e[0] == '1'; }();
namespace { thread_local std::vector<std::pair<const char*, double>> g_host_trace; }
void host_trace_mark(const char* label) {
  const double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
  if (label) { g_host_trace.emplace_back(label, t); return; }
  // nullptr: print and clear
  for (size_t i = 0; i < g_host_trace.size(); ++i) fprintf(stderr, "hosttrace %9.1f us  +%7.1f  %s\n", g_host_trace[i].second - g_host_trace[0].second, i ? g_host_trace[i].second - g_host_trace[i - 1].second : 0.0, g_host_trace[i].first);
  g_host_trace.clear();
}

// ---- a small persistent pool for host-side loops -------------------------------------------------------------------------------------------------------
namespace {
struct HostPool {
  static constexpr int T = 3;
  std::mutex mu; std::condition_variable cv_work, cv_done; const std::function<void(size_t)>* f = nullptr; size_t n = 0; std::atomic<size_t> next{0}; int active = 0; uint64_t gen = 0; bool started = false;
  std::mutex call_mu;                                       // one parallel loop at a time (concurrent callers fall back to their own thread)
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(size_t)>* fn; size_t cnt;
      { std::unique_lock<std::mutex> lk(mu); cv_work.wait(lk, [&] { return gen != seen; }); seen = gen; fn = f; cnt = n; }
      for (size_t i; (i = next.fetch_add(1)) < cnt;) (*fn)(i);
      { std::lock_guard<std::mutex> lk(mu); if (--active == 0) cv_done.notify_all(); }
    }
  }
};
HostPool* g_host_pool = nullptr; std::once_flag g_host_pool_once;
}  // namespace
void host_parallel_for(size_t n, const std::function<void(size_t)>& f) {
  if (n < 2) { for (size_t i = 0; i < n; ++i) f(i); return; }
  std::call_once(g_host_pool_once, [] {
    HostPool* p = new HostPool();                            // leaked on purpose: its threads sleep until the process ends
    try { for (int t = 0; t < HostPool::T; ++t) std::thread([p] { p->worker(); }).detach(); p->started = true; } catch (...) { p->started = false; }
    g_host_pool = p;
  });
  HostPool* p = g_host_pool;
  std::unique_lock<std::mutex> one(p->call_mu, std::try_to_lock);
  if (!p->started || !one.owns_lock()) { for (size_t i = 0; i < n; ++i) f(i); return; }
  { std::lock_guard<std::mutex> lk(p->mu); p->f = &f; p->n = n; p->next.store(0); p->active = HostPool::T; ++p->gen; }
  p->cv_work.notify_all();
  for (size_t i; (i = p->next.fetch_add(1)) < n;) f(i);
  { std::unique_lock<std::mutex> lk(p->mu); p->cv_done.wait(lk, [&] { return p->active == 0; }); }
}

static std::mutex g_dev_mu;
static std::map<int, Device*> g_devs;
static int32_t create_streams_in_order(Device* d);

int32_t ensure_host_pinned(Ctx* c, size_t bytes) {
  if (bytes <= c->h_pinned_cap) return ALEO_MI355X_OK;
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  c->h_pinned = nullptr; c->h_pinned_cap = 0;
  size_t want = bytes < 65536 ? 65536 : bytes;
  HIPCHK(hipHostMalloc(&c->h_pinned, want, hipHostMallocMapped));      // device-mapped: the last fold kernel of an MSM stores its result here
  c->h_pinned_cap = want; return ALEO_MI355X_OK;
}

// ---- slot scratch shared by asynchronous calls (ctx.h) --------------------------------------------------
int32_t scratch_acquire(Ctx* c, DevBuf& b, size_t bytes, hipStream_t s) {
  if (c->scratch_busy) {
    const bool grows = bytes > b.cap;
    if (c->scratch_stream && (grows || c->scratch_stream != s)) { HIPCHK(hipEventRecord(c->scratch_ev, c->scratch_stream)); c->scratch_stream = nullptr; }      // the deferred record: behind everything queued there so far
    if (grows) { HIPCHK(hipEventSynchronize(c->scratch_ev)); c->scratch_busy = false; }   // reserve() is about to free it
    else if (!c->scratch_stream) HIPCHK(hipStreamWaitEvent(s, c->scratch_ev, 0));
    // (same stream as the last user: stream order is the order)
  }
  return b.reserve(bytes);
}
int32_t scratch_release(Ctx* c, hipStream_t s) {
  c->scratch_busy = true;
  if (s && (s == c->stream || s == c->side || s == c->aux || s == c->hi)) { c->scratch_stream = s; return ALEO_MI355X_OK; }      // a stream the slot owns: it outlives the deferred record
  c->scratch_stream = nullptr;
  HIPCHK(hipEventRecord(c->scratch_ev, s)); return ALEO_MI355X_OK;      // a caller's stream may be gone by the next call: record now
}

static int32_t init_device(int device, Device** out) {
  std::lock_guard<std::mutex> lk(g_dev_mu);
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { g_last_error = "no HIP device visible"; return ALEO_MI355X_ERR_NO_DEVICE; }
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) { g_last_error = "hipGetDevice failed"; return ALEO_MI355X_ERR_NO_DEVICE; } }
  if (device >= count) { g_last_error = "device index out of range"; return ALEO_MI355X_ERR_BAD_ARG; }
  auto it = g_devs.find(device);
  if (it != g_devs.end()) { *out = it->second; (void)hipSetDevice(device); return ALEO_MI355X_OK; }      // "selects": the calling thread works on this device from here on
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_last_error = std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only";
    return ALEO_MI355X_ERR_NO_DEVICE;
  }
  std::unique_ptr<Device> d(new Device());
  d->device = device;
  if (const char* e = std::getenv("ALEO_MI355X_SLOTS")) { int k = std::atoi(e); if (k >= 1 && k <= MAX_SLOTS) d->n_slots = k; }
  for (int i = 0; i < MAX_SLOTS; ++i) { d->slots[i].dev = d.get(); d->slots[i].device = device; d->helpers[i].dev = d.get(); d->helpers[i].device = device; }
  *out = d.get();
  g_devs[device] = d.release();
  return create_streams_in_order(*out);                      // the main streams first: a hardware queue each (see "streams" below)
}

// Direct peer access between every ordered pair of initialised devices (xGMI: hipMemcpyPeerAsync then moves data link to link instead of through a
// staging buffer; the sharded transform's exchange is one such copy per pair).  "Already enabled" is fine; a refusal (no link, IOMMU) is recorded and
// the copies fall back to the runtime's staged path — never fatal.  Idempotent: pairs are tried once.
static std::map<std::pair<int, int>, bool> g_peer;          // (device, peer) -> direct access enabled; guarded by g_dev_mu
static void enable_peer_access() {
  std::lock_guard<std::mutex> lk(g_dev_mu);
  int cur = 0; if (hipGetDevice(&cur) != hipSuccess) cur = 0;
  for (const auto& a : g_devs) for (const auto& b : g_devs) {
    if (a.first == b.first || g_peer.count({a.first, b.first})) continue;
    int can = 0; bool ok = false;
    if (hipDeviceCanAccessPeer(&can, a.first, b.first) == hipSuccess && can && hipSetDevice(a.first) == hipSuccess) {
      const hipError_t e = hipDeviceEnablePeerAccess(b.first, 0);
      ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
      (void)hipGetLastError();                               // "already enabled" must not linger as the thread's last error
    }
    g_peer[{a.first, b.first}] = ok;
  }
  (void)hipSetDevice(cur);
}

// GPU_MAX_HW_QUEUES (read by the HIP runtime when it initialises; default 4) is an environment REQUIREMENT of the host, documented in the header and in
// the library never touches the process environment (setenv from a loaded library races with getenv in the host's other threads).
// The Python package and bench.py set their own default before HIP starts.

static int32_t get_device(Device** out) {
  int device = -1;
  if (hipGetDevice(&device) != hipSuccess) { g_last_error = "no HIP device visible"; return ALEO_MI355X_ERR_NO_DEVICE; }
  {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    auto it = g_devs.find(device);
    if (it != g_devs.end()) { *out = it->second; return ALEO_MI355X_OK; }
  }
  return init_device(device, out);
}

// ---- streams -------------------------------------------------------------------------------------------------------------------------------------------
// The runtime deals the streams of one priority onto at most GPU_MAX_HW_QUEUES hardware queues (to the queue with the fewest streams, so in creation order), and what shares a
// hardware queue runs in queue order.  Measured on this pool (profiles/r05_stream_order_*.txt, r05_hw_queues_ab.txt, r05_hi_priority_ab.txt):
//   * the MAIN streams are the ones that run side by side (lockstep groups and their workers, concurrent callers): created first — the slots', then as many helpers' — they get a
//     queue each; with every context creating stream | side | hi at its first use they collided: a lockstep call of 8 proofs 31.8 -> 28.0 ms, of 16 58.5 -> 54.0;
//   * more hardware queues than ~20 in all (GPU_MAX_HW_QUEUES >= 12, or a third priority level) and the same call takes 43-65 ms: the queues are oversubscribed;
//   * EIGHT high-priority streams created in a row (one per context) made a 2^20-constraint proof — whose chains' sorts run on them — take 1.1-2.5 s instead of 80 ms; at normal
//     priority the same arrangement is harmless.  So the device keeps a small POOL of high-priority streams (ALEO_MI355X_HI_POOL, default 4; chunked uploads need one per later
//     chunk) that its contexts share round-robin, slot i and helper i + 1 on different ones.
static bool hi_priority_on() { static const bool v = [] { const char* e = std::getenv("ALEO_MI355X_HI_PRIORITY"); return !(e && e[0] == '0'); }(); return v; }      // A/B: 0 = the `hi` streams at normal priority
static int hi_pool_size() { static const int v = [] { const char* e = std::getenv("ALEO_MI355X_HI_POOL"); const int k = e ? std::atoi(e) : 4; return k >= 1 && k <= MAX_SLOTS ? k : 4; }(); return v; }
static std::mutex g_stream_mu;
static int32_t create_streams_in_order(Device* d) {
  static const bool on = [] { const char* e = std::getenv("ALEO_MI355X_STREAM_ORDER"); return !(e && e[0] == '0'); }();      // A/B: 0 = every context creates its streams at its first use
  if (!on) return ALEO_MI355X_OK;
  std::lock_guard<std::mutex> lk(g_stream_mu);
  if (d->slots[0].stream) return ALEO_MI355X_OK;
  for (int i = 0; i < d->n_slots; ++i) HIPCHK(hipStreamCreateWithFlags(&d->slots[i].stream, hipStreamNonBlocking));
  for (int i = 0; i < d->n_slots; ++i) HIPCHK(hipStreamCreateWithFlags(&d->helpers[i].stream, hipStreamNonBlocking));
  return ALEO_MI355X_OK;
}
static int32_t hi_stream_for(Ctx* c, hipStream_t* out) {
  Device* d = c->dev;
  if (!d) { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); if (!hi_priority_on()) hi = 0; HIPCHK(hipStreamCreateWithPriority(out, hipStreamNonBlocking, hi)); return ALEO_MI355X_OK; }
  std::lock_guard<std::mutex> lk(g_stream_mu);
  const bool helper = c >= &d->helpers[0] && c < &d->helpers[MAX_SLOTS];
  const int idx = helper ? (int)(c - &d->helpers[0]) + 1 : (int)(c - &d->slots[0]), k = idx % hi_pool_size();      // slot i and helpers i, i + 1 (a pipeline's ring, a chunked call's later chunks) on different ones
  while (d->hi_made <= k) {
    int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); if (!hi_priority_on()) hi = 0;      // hi = the numerically lowest = highest priority
    HIPCHK(hipStreamCreateWithPriority(&d->hi_pool[d->hi_made], hipStreamNonBlocking, hi)); ++d->hi_made;
  }
  *out = d->hi_pool[k]; return ALEO_MI355X_OK;
}
static int32_t first_use(Ctx* c) {          // streams and events of a slot, created when it is first handed out (the device is current)
  if (c->ev[0]) return ALEO_MI355X_OK;
  if (c->dev) { const int32_t rc = create_streams_in_order(c->dev); if (rc) return rc; }
  if (!c->stream) HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  if (!c->side) HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  if (!c->aux) HIPCHK(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
  if (!c->hi) { const int32_t rc = hi_stream_for(c, &c->hi); if (rc) return rc; }
  for (auto& e : c->ev) HIPCHK(hipEventCreate(&e));
  HIPCHK(hipEventCreateWithFlags(&c->scratch_ev, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_hop, hipEventDisableTiming));
  return ALEO_MI355X_OK;
}

// Picks a free slot (or waits on one chosen by thread id) and locks it for the duration of the call.
static int32_t acquire_slot(Device* d, Ctx** out, std::unique_lock<std::mutex>& lk) {
  Ctx* c = nullptr;
  for (int i = 0; i < d->n_slots && !c; ++i) {
    std::unique_lock<std::mutex> t(d->slots[i].mu, std::try_to_lock);
    if (t.owns_lock()) { lk = std::move(t); c = &d->slots[i]; }
  }
  if (!c) {
    size_t i = std::hash<std::thread::id>()(std::this_thread::get_id()) % (size_t)d->n_slots;
    lk = std::unique_lock<std::mutex>(d->slots[i].mu); c = &d->slots[i];
  }
  if (hipSetDevice(d->device) != hipSuccess) { g_last_error = "hipSetDevice failed"; return ALEO_MI355X_ERR_HIP; }
  { const int32_t rc = first_use(c); if (rc) return rc; }
  *out = c; return ALEO_MI355X_OK;
}

// A context of device d for work done on behalf of a call that already holds `exclude` (a slot of the same device, or nullptr): free slots first, then
// helper contexts, never `exclude` itself — the holder is waiting for this work.  It never blocks on ONE context: the contexts of the caller's own device
// may all be held by workers of the very call this work belongs to (a lockstep call parked at its round barrier, waiting for the commitment this shard is
// part of — the round-4 form blocked on helpers[hash(tid) % 8] and could deadlock there).  `may_wait`: poll all of them until one is free (another device:
// whoever holds its contexts is an independent call and will finish); otherwise *out stays nullptr and the caller lends its own context (commit_sharded).
static int32_t acquire_other(Device* d, const Ctx* exclude, Ctx** out, std::unique_lock<std::mutex>& lk, bool may_wait) {
  Ctx* c = nullptr; *out = nullptr;
  for (;;) {
    for (int i = 0; i < d->n_slots && !c; ++i) {
      if (&d->slots[i] == exclude) continue;
      std::unique_lock<std::mutex> t(d->slots[i].mu, std::try_to_lock);
      if (t.owns_lock()) { lk = std::move(t); c = &d->slots[i]; }
    }
    for (int i = 0; i < MAX_SLOTS && !c; ++i) {
      if (&d->helpers[i] == exclude) continue;
      std::unique_lock<std::mutex> t(d->helpers[i].mu, std::try_to_lock);
      if (t.owns_lock()) { lk = std::move(t); c = &d->helpers[i]; }
    }
    if (c || !may_wait) break;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  if (!c) return ALEO_MI355X_OK;
  if (hipSetDevice(d->device) != hipSuccess) { g_last_error = "hipSetDevice failed"; return ALEO_MI355X_ERR_HIP; }
  { const int32_t rc = first_use(c); if (rc) return rc; }
  *out = c; return ALEO_MI355X_OK;
}

int32_t acquire_helpers(Device* d, int want, HelperSet& hs) {
  for (int i = 0; i < MAX_SLOTS && (int)hs.ctx.size() < want; ++i) {
    std::unique_lock<std::mutex> t(d->helpers[i].mu, std::try_to_lock);
    if (!t.owns_lock()) continue;
    { const int32_t rc = first_use(&d->helpers[i]); if (rc) return rc; }
    hs.ctx.push_back(&d->helpers[i]); hs.locks.push_back(std::move(t));
  }
  return ALEO_MI355X_OK;
}

// ---- pinned base sets (shared by all slots) ----------------------------------------------------------
static int32_t upload_bases(Ctx* c, const void* bases, size_t stride, size_t n, std::shared_ptr<PinnedOwner>* out) {
  auto o = std::make_shared<PinnedOwner>(); PinnedBases& pb = o->pb; pb.n = n;
  size_t bytes = (n ? n : 1) * 96;
  HIPCHK(hipMalloc(&pb.d_xy, bytes));
  bool any_inf = false;
  if (stride == 96) {
    HIPCHK(hipMemcpy(pb.d_xy, bases, n * 96, hipMemcpyHostToDevice));
  } else if (n) {
    // snarkVM's 104-byte Affine: the rows go up as they are (one copy at PCIe rate; a host loop that strips the infinity byte + padding first
    // cost 30 ms per 2^20 points — the one-shot call of the two-line drop-in spent most of its time there) and are unpacked on the device
    DevTmp raw, flags; int32_t rcu;
    if ((rcu = raw.alloc(n * 104)) || (rcu = flags.alloc(n + 8))) return rcu;
    HIPCHK(hipMemcpyAsync(raw.p, bases, n * 104, hipMemcpyHostToDevice, c->stream));
    if ((rcu = unpack_affine104(c, raw.p, pb.d_xy, flags.p, n, c->stream))) return rcu;      // rows -> 96-byte x | y, flag bytes, their count behind the flags
    uint32_t n_inf = 0;
    HIPCHK(hipMemcpyAsync(&n_inf, (char*)flags.p + ((n + 3) & ~(size_t)3), 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n_inf) { pb.d_inf = (uint8_t*)flags.release(); any_inf = true; }
  }
  { int32_t rc28 = make_rows28(c, &pb); if (rc28) return rc28; }
  *out = std::move(o); return ALEO_MI355X_OK;
}
// The same into the slot's own buffers, as a view nobody owns (the cold one-shot call: aleo_mi355x_msm_g1 without the SRS cache)
static int32_t cold_bases(Ctx* c, const void* bases, size_t stride, size_t n, PinnedBases* pb) {
  *pb = PinnedBases(); pb->n = n; int32_t rc;
  if ((rc = c->cold_xy.reserve((n ? n : 1) * 96)) || (rc = c->cold_xy28.reserve((n ? n : 1) * ROW28))) return rc;
  pb->d_xy = c->cold_xy.p;
  if (stride == 96) { if (n) HIPCHK(hipMemcpyAsync(pb->d_xy, bases, n * 96, hipMemcpyHostToDevice, c->stream)); }
  else if (n) {
    if ((rc = c->cold_raw.reserve(n * 104)) || (rc = c->cold_flags.reserve(n + 16))) return rc;
    HIPCHK(hipMemcpyAsync(c->cold_raw.p, bases, n * 104, hipMemcpyHostToDevice, c->stream));
    if ((rc = unpack_affine104(c, c->cold_raw.p, pb->d_xy, c->cold_flags.p, n, c->stream))) return rc;
    uint32_t n_inf = 0;
    HIPCHK(hipMemcpyAsync(&n_inf, (char*)c->cold_flags.p + ((n + 3) & ~(size_t)3), 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n_inf) pb->d_inf = (uint8_t*)c->cold_flags.p;
  }
  if ((rc = rows_to28_into(pb->d_xy, c->cold_xy28.p, n, c->stream))) return rc;
  pb->d_xy28 = c->cold_xy28.p;
  return ALEO_MI355X_OK;
}
static uint64_t register_bases(Device* d, std::shared_ptr<PinnedOwner> o) {
  std::lock_guard<std::mutex> lk(d->mu);
  uint64_t h = d->next_handle++; d->bases[h] = std::move(o); return h;
}
// The set behind a handle plus a snapshot of its fields (the table pointer may be published by another slot at any time).
static int32_t find_bases(Device* d, uint64_t handle, std::shared_ptr<PinnedOwner>* keep, PinnedBases* snap) {
  std::lock_guard<std::mutex> lk(d->mu);
  auto it = d->bases.find(handle);
  if (it == d->bases.end()) { g_last_error = "unknown bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
  *keep = it->second; *snap = it->second->pb; return ALEO_MI355X_OK;
}
static int32_t unpin(Device* d, uint64_t handle) {
  std::shared_ptr<PinnedOwner> dead;          // freed (hipFree synchronises) after the lock is dropped, once no call uses it
  std::lock_guard<std::mutex> lk(d->mu);
  auto it = d->bases.find(handle);
  if (it == d->bases.end()) { g_last_error = "unknown bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
  dead = std::move(it->second); d->bases.erase(it);
  return ALEO_MI355X_OK;
}
// Builds the fixed-base table of a set on slot c unless it exists or another slot is building it; publishes it under the lock.
static int32_t precompute_once(Ctx* c, const std::shared_ptr<PinnedOwner>& o) {
  Device* d = c->dev; PinnedBases work;
  {
    std::lock_guard<std::mutex> lk(d->mu);
    if (o->pb.tabled || o->building) return ALEO_MI355X_OK;
    o->building = true; work = o->pb;
  }
  int32_t rc = msm_precompute(c, &work);
  std::lock_guard<std::mutex> lk(d->mu);
  o->building = false;
  if (rc == ALEO_MI355X_OK) { for (int i = 0; i < 3; ++i) o->pb.tab[i] = work.tab[i]; o->pb.tabled = true; }
  return rc;
}

// Canonical host scalars whose sample is mostly 0 / 1 / short (a witness: SURVEY.md §8d "witness-like") take the set's range table when it has one
// and covers the call: 257 scalars spread over the vector, sparse = at least half of them below 2^32.
static bool looks_sparse(const void* scalars, size_t n) {
  if (n < 4096) return false;
  const uint64_t* s = (const uint64_t*)scalars; size_t small = 0; const size_t step = n / 257;
  for (size_t i = 0; i < 257; ++i) { const uint64_t* v = s + 4 * (i * step); small += (v[1] | v[2] | v[3]) == 0 && v[0] < (1ull << 32); }
  return small >= 129;
}
static int32_t msm_host_scalars(Ctx* c, void* out, const PinnedBases& pb, const void* scalars, size_t n, bool mont) {
  const bool skewed = !mont && looks_sparse(scalars, n);          // witness-like: the few huge buckets (their slice trees beside the reduction) decide, not the upload — one chain
  const bool sparse = skewed && pb.range.d && pb.range_off == 0 && n <= pb.range.cover;
  if (!n) return msm_run1(c, (uint64_t*)out, pb, nullptr, 0, mont, c->stream, false);
  return msm_run1_split(c, (uint64_t*)out, pb, nullptr, n, mont, c->stream, sparse, scalars, !skewed);      // uploads inside (whole, or in two halves that share one reduction)
}

// ---- SRS cache for the one-shot entry point ----------------------------------------------------------
static uint64_t hash96(const uint8_t* p) {      // FNV-1a over the 96 coordinate bytes of one point
  uint64_t h = 1469598103934665603ull;
  for (int i = 0; i < 96; ++i) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}
static constexpr size_t SRS_SAMPLES = 256, SRS_CACHE_ENTRIES = 8, SRS_MIN_N = 1024;

static bool srs_cache_enabled() {          // opt-in (see the header); read per call so a host can switch it around a phase
  const char* e = std::getenv("ALEO_MI355X_SRS_CACHE");
  return e && e[0] == '1';
}
// Caller holds d->mu.  Returns the cached entry for (bases, stride) that covers n points, or nullptr.
static SrsCacheEntry* srs_lookup(Device* d, const void* bases, size_t stride, size_t n) {
  for (auto& e : d->srs_cache) {
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
// The resident set for a one-shot call's base array: a cache hit, or a fresh upload that replaces stale / least recently
// used entries.  `want_table` is set on the third use of a set large enough to repay the one-off table build.
static int32_t srs_get(Ctx* c, Device* d, const void* bases, size_t stride, size_t n, std::shared_ptr<PinnedOwner>* keep, bool* want_table) {
  *want_table = false;
  {
    std::lock_guard<std::mutex> lk(d->mu);
    if (SrsCacheEntry* e = srs_lookup(d, bases, stride, n)) {
      e->last_use = ++d->srs_clock; e->hits++;
      *keep = d->bases[e->handle];
      *want_table = e->hits >= 3 && e->n >= (1u << 10) && !(*keep)->pb.tabled;
      return ALEO_MI355X_OK;
    }
  }
  std::shared_ptr<PinnedOwner> o;              // the bulk upload runs without the lock
  int32_t rc = upload_bases(c, bases, stride, n, &o);
  if (rc) return rc;
  SrsCacheEntry e; e.host_ptr = bases; e.stride = stride; e.n = n; e.hits = 1;
  // dense samples at the front (every prefix request can be checked), sparse ones over the rest
  for (size_t k = 0; k < SRS_SAMPLES; ++k) {
    size_t idx = k < 32 ? k : (size_t)((double)(k - 31) / (SRS_SAMPLES - 31) * (n - 1));
    if (idx >= n) break;
    e.samples.emplace_back(idx, hash96((const uint8_t*)bases + idx * stride));
  }
  std::vector<std::shared_ptr<PinnedOwner>> dead;
  {
    std::lock_guard<std::mutex> lk(d->mu);
    auto drop = [&](size_t i) {
      auto it = d->bases.find(d->srs_cache[i].handle);
      if (it != d->bases.end()) { dead.push_back(std::move(it->second)); d->bases.erase(it); }
      d->srs_cache.erase(d->srs_cache.begin() + i);
    };
    for (size_t i = 0; i < d->srs_cache.size();) { if (d->srs_cache[i].host_ptr == bases) drop(i); else ++i; }
    if (d->srs_cache.size() >= SRS_CACHE_ENTRIES) {
      size_t lru = 0; for (size_t i = 1; i < d->srs_cache.size(); ++i) if (d->srs_cache[i].last_use < d->srs_cache[lru].last_use) lru = i;
      drop(lru);
    }
    e.handle = d->next_handle++; e.last_use = ++d->srs_clock;
    d->bases[e.handle] = o; d->srs_cache.push_back(e);
  }
  *keep = std::move(o);
  return ALEO_MI355X_OK;
}

static void jacobian_to_affine104(void* out104, const uint64_t* jac18) {
  uint8_t* o = (uint8_t*)out104; std::memset(o, 0, 104);
  bool inf = true; for (int i = 12; i < 18; ++i) if (jac18[i]) inf = false;
  if (inf) { o[96] = 1; host::HFq one = host::HFq::one(); std::memcpy(o + 48, one.l, 48); return; }  // Affine::zero() = (0, 1, true)
  std::memcpy(o, jac18, 96);   // results are normalised: z == 1
}

void jacobian_rows_to_affine104(void* out104, const uint64_t* jac18, size_t k) {
  for (size_t q = 0; q < k; ++q) jacobian_to_affine104((uint8_t*)out104 + 104 * q, jac18 + 18 * q);
}

}  // namespace aleo_mi355x

using namespace aleo_mi355x;

// every entry point: the calling thread's device, then a slot of it locked for the duration of the call
#define API_BEGIN Device* d = nullptr; Ctx* c = nullptr; std::unique_lock<std::mutex> lk; \
  { int32_t rc0 = get_device(&d); if (rc0) return rc0; if ((rc0 = acquire_slot(d, &c, lk))) return rc0; }
#define FIND_BASES(handle) std::shared_ptr<PinnedOwner> keep; PinnedBases pb; { int32_t rcb = find_bases(d, handle, &keep, &pb); if (rcb) return rcb; }

// *_device entry points that only enqueue work.  On the caller's stream they return without synchronising (the caller orders
// its own work there).  With stream == NULL they run on the serving slot's stream, which the caller cannot order anything
// against — a following call may be served by another slot — so the work is complete when they return.
// The stream a call works on: the caller's, or the slot's own for NULL.  hipStreamLegacy is passed on as the null stream it
// names (this library is not built with a per-thread default stream): the runtime takes the special handle for launches but
// not for every event call.
// hipStreamPerThread is refused: it names a different stream on every host thread, and a slot's events may be waited on by another.
static int32_t pick_stream(Ctx* c, void* stream, hipStream_t* out) {
  if ((hipStream_t)stream == hipStreamPerThread) { g_last_error = "stream: hipStreamPerThread is not supported (pass a created stream, hipStreamLegacy or NULL)"; return ALEO_MI355X_ERR_BAD_ARG; }
  *out = !stream ? c->stream : ((hipStream_t)stream == hipStreamLegacy ? (hipStream_t)nullptr : (hipStream_t)stream);
  return ALEO_MI355X_OK;
}
#define PICK_STREAM(var) hipStream_t var = nullptr; { int32_t rcs = pick_stream(c, stream, &var); if (rcs) return rcs; }
template <class F> static int32_t run_enqueue(Ctx* c, void* stream, F&& f) {
  PICK_STREAM(s)
  int32_t rc = f(s);
  if (rc) return rc;
  if (!stream) HIPCHK(hipStreamSynchronize(s));
  return ALEO_MI355X_OK;
}

namespace {      // aleo_mi355x_selftest_host_inverse
template <int N> void inverse_selftest(uint32_t count, uint64_t seed, uint32_t* failures, double* ns) {
  using F = host::HFp<N>; using Pm = host::HParams<N>;
  std::vector<F> v;
  auto from_canon = [](const uint64_t* c) { F x; std::memcpy(x.l, c, sizeof x.l); return x; };
  { uint64_t e[N]; std::memset(e, 0, sizeof e); v.push_back(from_canon(e)); e[0] = 1; v.push_back(from_canon(e)); e[0] = 2; v.push_back(from_canon(e));
    std::memcpy(e, Pm::P, sizeof e); e[0] -= 1; v.push_back(from_canon(e)); e[0] -= 1; v.push_back(from_canon(e)); }
  uint64_t st = seed * 0x9e3779b97f4a7c15ull + N;
  auto next = [&]() { st += 0x9e3779b97f4a7c15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); };
  for (uint32_t i = 0; i < count; ++i) {
    F x; for (int k = 0; k < N; ++k) x.l[k] = next();
    x.l[N - 1] &= (1ull << (N == 4 ? 60 : 56)) - 1;        // below the modulus' top limb: a valid residue
    if (i % 7 == 3) { for (int k = 1; k < N; ++k) x.l[k] = 0; }      // short values too
    v.push_back(x);
  }
  std::vector<F> a(v.size()), b(v.size());
  auto t0 = std::chrono::steady_clock::now();
  for (size_t i = 0; i < v.size(); ++i) a[i] = F::inv(v[i]);
  auto t1 = std::chrono::steady_clock::now();
  for (size_t i = 0; i < v.size(); ++i) b[i] = F::inv_fermat(v[i]);
  auto t2 = std::chrono::steady_clock::now();
  for (size_t i = 0; i < v.size(); ++i) {
    bool ok = a[i] == b[i];
    if (ok && !v[i].is_zero()) ok = F::mul(a[i], v[i]) == F::one();      // and it IS the inverse
    ok = ok && F::sqr(v[i]) == F::mul(v[i], v[i]) && F::sqr(a[i]) == F::mul(a[i], a[i]);      // the dedicated squaring against the product
    { host::Wide<N> w; w.set_sqr(v[i].l); host::Wide<N> u; u.set_mul(v[i].l, v[i].l); ok = ok && w.redc() == u.redc(); }
    if (!ok) ++*failures;
  }
  if (ns) { ns[0] = std::chrono::duration<double, std::nano>(t1 - t0).count() / v.size(); ns[1] = std::chrono::duration<double, std::nano>(t2 - t1).count() / v.size(); }
}
}  // namespace
extern "C" {

int32_t aleo_mi355x_init_device(int32_t device) {
  try { Device* d = nullptr; return init_device(device, &d); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

// init(n_devices, 0 = all).  Initialises the first n visible devices; the calling thread's current device is left as it was.
int32_t aleo_mi355x_init(int32_t n_devices) {
  try {
    int count = 0, cur = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { g_last_error = "no HIP device visible"; return ALEO_MI355X_ERR_NO_DEVICE; }
    if (n_devices < 0 || n_devices > count) { g_last_error = "init: n_devices outside 0..visible devices (0 = all)"; return ALEO_MI355X_ERR_BAD_ARG; }
    if (hipGetDevice(&cur) != hipSuccess) cur = 0;
    const int n = n_devices ? n_devices : count;
    int32_t rc = ALEO_MI355X_OK;
    for (int i = 0; i < n && !rc; ++i) { Device* d = nullptr; rc = init_device(i, &d); }
    if (!rc) enable_peer_access();
    (void)hipSetDevice(cur);
    return rc;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_peer_info(int32_t* enabled_pairs, int32_t* refused_pairs) {
  try {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    int32_t on = 0, off = 0;
    for (const auto& kv : g_peer) (kv.second ? on : off)++;
    if (enabled_pairs) *enabled_pairs = on;
    if (refused_pairs) *refused_pairs = off;
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_device_count(int32_t* visible, int32_t* initialised) {
  try {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) count = 0;
    if (visible) *visible = count;
    if (initialised) { std::lock_guard<std::mutex> lk(g_dev_mu); *initialised = (int32_t)g_devs.size(); }
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_pin(const void* bases, size_t base_stride, size_t n, uint64_t* handle) {
  try {
    if (!handle || (!bases && n) || (base_stride != 104 && base_stride != 96)) { g_last_error = "bases_pin: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    std::shared_ptr<PinnedOwner> o; int32_t rc = upload_bases(c, bases, base_stride, n, &o);
    if (rc) return rc;
    *handle = register_bases(d, std::move(o));
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_generate(const void* base104, uint64_t first, size_t n, uint64_t* handle) {
  try {
    if (!base104 || !handle) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    auto o = std::make_shared<PinnedOwner>();
    int32_t rc = generate_multiples(c, base104, first, n, &o->pb);
    if (rc) return rc;
    *handle = register_bases(d, std::move(o));
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_from_scalars(const void* base104, const void* scalars, size_t n, uint64_t* handle) {
  try {
    if (!base104 || !handle || !scalars) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    auto o = std::make_shared<PinnedOwner>();
    int32_t rc = generate_from_scalars(c, base104, scalars, n, &o->pb);
    if (rc) return rc;
    *handle = register_bases(d, std::move(o));
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_precompute(uint64_t handle) {
  try {
    API_BEGIN
    FIND_BASES(handle)
    return precompute_once(c, keep);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_precompute_range(uint64_t handle, size_t offset, size_t n, int32_t window_bits) {
  try {
    API_BEGIN
    std::shared_ptr<PinnedOwner> keep; PinnedBases work; { int32_t rcb = find_bases(d, handle, &keep, &work); if (rcb) return rcb; }
    { std::lock_guard<std::mutex> g(d->mu); if (keep->building) { g_last_error = "bases_precompute_range: a table build is in flight"; return ALEO_MI355X_ERR_BAD_ARG; } keep->building = true; work = keep->pb; }
    int32_t rc = msm_precompute_range(c, &work, offset, n, window_bits);
    std::lock_guard<std::mutex> g(d->mu);
    keep->building = false;
    if (rc == ALEO_MI355X_OK) { keep->pb.range = work.range; keep->pb.range_off = work.range_off; }
    return rc;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_info(uint64_t handle, uint64_t* out, int32_t cap) {
  try {
    if (!out || cap <= 0) return 0;
    Device* d = nullptr; if (get_device(&d)) return 0;
    std::shared_ptr<PinnedOwner> keep; PinnedBases pb; if (find_bases(d, handle, &keep, &pb)) return 0;
    uint64_t v[8] = {pb.n, pb.n * (96 + ROW28) + (pb.d_inf ? pb.n : 0), 0, 0, 0, 0, 0, 0};
    int k = 0;
    for (const auto& t : pb.tab) if (t.d) { const uint64_t W = (254 + t.c - 1) / t.c; v[2] += W * t.cover * ROW28; v[3 + k] = (uint64_t)t.c; ++k; }
    v[6] = (uint64_t)k;
    const int32_t m = cap < 8 ? cap : 8;
    for (int32_t i = 0; i < m; ++i) out[i] = v[i];
    return m;
  } catch (...) { return 0; }
}

int32_t aleo_mi355x_bases_download(uint64_t handle, size_t offset, size_t n, void* out104) {
  try {
    if (!out104 && n) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    FIND_BASES(handle)
    if (offset + n > pb.n) { g_last_error = "bases_download: range"; return ALEO_MI355X_ERR_BAD_ARG; }
    std::vector<uint8_t> xy(n * 96 + 1), inf(n + 1, 0);
    HIPCHK(hipMemcpy(xy.data(), (const char*)pb.d_xy + offset * 96, n * 96, hipMemcpyDeviceToHost));
    if (pb.d_inf) HIPCHK(hipMemcpy(inf.data(), pb.d_inf + offset, n, hipMemcpyDeviceToHost));
    uint8_t* o = (uint8_t*)out104;
    for (size_t i = 0; i < n; ++i) { std::memcpy(o + i * 104, &xy[i * 96], 96); std::memset(o + i * 104 + 96, 0, 8); o[i * 104 + 96] = inf[i]; }
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_unpin(uint64_t handle) {
  try { API_BEGIN return unpin(d, handle); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_msm_g1(void* out, const void* bases, size_t base_stride, const void* scalars, size_t n) {
  try {
    if (!out || ((!bases || !scalars) && n) || (base_stride != 104 && base_stride != 96)) { g_last_error = "msm_g1: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    if (n >= SRS_MIN_N && srs_cache_enabled()) {
      // ALEO_MI355X_SRS_CACHE=1 (opt-in): KZG10::commit multiplies against prefixes of one SRS, so the base array is kept in
      // HBM between calls, recognised by host pointer + sampled content.  Off by default: it reads caller memory the ABI
      // otherwise does not retain, and an array rewritten in place at unsampled entries would be served stale.
      std::shared_ptr<PinnedOwner> keep; bool want_table = false;
      int32_t rc = srs_get(c, d, bases, base_stride, n, &keep, &want_table);
      if (rc) return rc;
      if (want_table) (void)precompute_once(c, keep);                       // third use: worth the one-off table
      PinnedBases pb; { std::lock_guard<std::mutex> g(d->mu); pb = keep->pb; }
      return msm_host_scalars(c, out, pb, scalars, n, false);
    }
    // Nothing cached (the literal two-line drop-in): the call's copy of the bases lives in the SLOT's grow-only buffers (rows as uploaded, x | y rows, 28-bit rows, flags:
    // 337 bytes per point, kept by the slot like its other workspaces), not in hipMalloc / hipFree pairs per call — hipFree waits for the device, and the pairs were 0.x ms of a
    // 7.6 ms call at 2^20 (ALEO_MI355X_COLD_POOL=0: a PinnedOwner per call, as before).
    static const bool cold_pool = [] { const char* e = std::getenv("ALEO_MI355X_COLD_POOL"); return !(e && e[0] == '0'); }();
    if (cold_pool) {
      PinnedBases pbc; const int32_t rc = cold_bases(c, bases, base_stride, n, &pbc);
      if (rc) return rc;
      return msm_host_scalars(c, out, pbc, scalars, n, false);
    }
    std::shared_ptr<PinnedOwner> o; int32_t rc = upload_bases(c, bases, base_stride, n, &o);
    if (rc) return rc;
    return msm_host_scalars(c, out, o->pb, scalars, n, false);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_msm_g1_pinned(void* out, uint64_t handle, const void* scalars, size_t n) {
  try {
    if (!out || (!scalars && n)) { g_last_error = "msm_g1_pinned: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    FIND_BASES(handle)
    return msm_host_scalars(c, out, pb, scalars, n, false);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_msm_g1_device(void* out, uint64_t handle, const void* d_scalars, size_t n, void* stream) {
  try {
    if (!out || (!d_scalars && n)) { g_last_error = "msm_g1_device: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    FIND_BASES(handle)
    PICK_STREAM(s)
    return msm_run1_split(c, (uint64_t*)out, pb, d_scalars, n, false, s, false, nullptr);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_msm_g1_device_sparse(void* out, uint64_t handle, const void* d_scalars, size_t n, void* stream) {
  try {
    if (!out || (!d_scalars && n)) { g_last_error = "msm_g1_device_sparse: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    FIND_BASES(handle)
    PICK_STREAM(s)
    return msm_run1(c, (uint64_t*)out, pb, d_scalars, n, false, s, true);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

// VariableBase::msm::<G2Affine>: one-shot, host pointers (G2 appears in SRS setup and verifying keys, never in the prover's loop:
// no residency handle).  bases: snarkVM G2Affine rows, stride 200 (flag byte at 192) or 192.
int32_t aleo_mi355x_msm_g2(void* out_jac288, const void* bases, size_t base_stride, const void* scalars, size_t n) {
  try {
    if (!out_jac288 || ((!bases || !scalars) && n) || (base_stride != 200 && base_stride != 192)) { g_last_error = "msm_g2: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    // the call's copy of the bases in the slot's grow-only buffers (shared with the cold G1 call: one call per slot at a time), not hipMalloc / hipFree pairs
    int32_t rc; bool any_inf = false;
    if ((rc = c->cold_xy.reserve((n ? n : 1) * 192))) return rc;
    void* xy = c->cold_xy.p;
    if (base_stride == 192) { if (n) HIPCHK(hipMemcpyAsync(xy, bases, n * 192, hipMemcpyHostToDevice, c->stream)); }
    else if (n) {
      // the 200-byte rows go up as they are and are unpacked on the device (round 4; the host loop that stripped the flag byte + padding first was
      // ~ 60 of the 91 ms of a 2^20-point call — the same finding as for G1's 104-byte rows in round 3)
      if ((rc = c->cold_raw.reserve(n * 200)) || (rc = c->cold_flags.reserve(n + 16))) return rc;
      HIPCHK(hipMemcpyAsync(c->cold_raw.p, bases, n * 200, hipMemcpyHostToDevice, c->stream));
      uint32_t* d_count = (uint32_t*)((char*)c->cold_flags.p + ((n + 3) & ~(size_t)3));
      if ((rc = g2_unpack200(c, c->cold_raw.p, xy, c->cold_flags.p, d_count, n, c->stream))) return rc;
      uint32_t n_inf = 0;
      HIPCHK(hipMemcpyAsync(&n_inf, d_count, 4, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      any_inf = n_inf != 0;
    }
    if ((rc = c->scalars_stage.reserve((n ? n : 1) * 32))) return rc;
    if (n) HIPCHK(hipMemcpyAsync(c->scalars_stage.p, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
    return msm_g2_run(c, (uint64_t*)out_jac288, xy, any_inf ? (const uint8_t*)c->cold_flags.p : nullptr, c->scalars_stage.p, n, c->stream);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

// A G2 base set kept on the device (round 5): rows, infinity flags and the 28-bit rows of the accumulation stay in HBM, so a call moves only its scalars
// (the one-shot entry point above uploads 192-200 bytes per base and rebuilds the 28-bit rows every time: 3.8 + 0.4 of its 22 ms at 2^20).  Any prefix
// of the set can be multiplied.  G2 appears in SRS setup and verifying keys, never in the prover's loop: no window tables.
static int32_t g2_upload(Ctx* c, const void* bases, size_t base_stride, size_t n, std::shared_ptr<PinnedG2>* out) {
  auto o = std::make_shared<PinnedG2>(); o->n = n; int32_t rc;
  HIPCHK(hipMalloc(&o->d_xy, (n ? n : 1) * 192)); HIPCHK(hipMalloc(&o->d_rows28, (n ? n : 1) * 224));
  if (base_stride == 192) { if (n) HIPCHK(hipMemcpyAsync(o->d_xy, bases, n * 192, hipMemcpyHostToDevice, c->stream)); }
  else if (n) {
    DevTmp raw, inf;
    if ((rc = raw.alloc(n * 200)) || (rc = inf.alloc(n + 8))) return rc;
    HIPCHK(hipMemcpyAsync(raw.p, bases, n * 200, hipMemcpyHostToDevice, c->stream));
    uint32_t* d_count = (uint32_t*)((char*)inf.p + ((n + 3) & ~(size_t)3));
    if ((rc = g2_unpack200(c, raw.p, o->d_xy, inf.p, d_count, n, c->stream))) return rc;
    uint32_t n_inf = 0;
    HIPCHK(hipMemcpyAsync(&n_inf, d_count, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n_inf) o->d_inf = (uint8_t*)inf.release();
  }
  if ((rc = g2_rows_to28(o->d_xy, o->d_rows28, n, c->stream))) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  *out = std::move(o); return ALEO_MI355X_OK;
}
int32_t aleo_mi355x_bases_g2_pin(const void* bases, size_t base_stride, size_t n, uint64_t* handle) {
  try {
    if (!handle || (!bases && n) || (base_stride != 200 && base_stride != 192) || n >= (1ull << 31)) { g_last_error = "bases_g2_pin: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    std::shared_ptr<PinnedG2> o;
    const int32_t rc = g2_upload(c, bases, base_stride, n, &o);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk2(d->mu);
    *handle = d->next_handle++; d->g2_bases[*handle] = std::move(o);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}
int32_t aleo_mi355x_bases_g2_unpin(uint64_t handle) {
  try {
    API_BEGIN
    std::shared_ptr<PinnedG2> keep;                          // freed outside the device lock (and after calls still holding it have returned)
    { std::lock_guard<std::mutex> lk2(d->mu); auto it = d->g2_bases.find(handle); if (it == d->g2_bases.end()) { g_last_error = "unknown G2 bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; } keep = std::move(it->second); d->g2_bases.erase(it); }
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}
int32_t aleo_mi355x_msm_g2_pinned(void* out_jac288, uint64_t handle, const void* scalars, size_t n) {
  try {
    if (!out_jac288 || (!scalars && n)) { g_last_error = "msm_g2_pinned: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    std::shared_ptr<PinnedG2> keep;
    { std::lock_guard<std::mutex> lk2(d->mu); auto it = d->g2_bases.find(handle); if (it == d->g2_bases.end()) { g_last_error = "unknown G2 bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; } keep = it->second; }
    if (n > keep->n) { g_last_error = "msm_g2_pinned: more scalars than pinned bases"; return ALEO_MI355X_ERR_BAD_ARG; }
    int32_t rc;
    if ((rc = c->scalars_stage.reserve((n ? n : 1) * 32))) return rc;
    if (n) HIPCHK(hipMemcpyAsync(c->scalars_stage.p, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
    return msm_g2_run(c, (uint64_t*)out_jac288, keep->d_xy, keep->d_inf, c->scalars_stage.p, n, c->stream, keep->d_rows28);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_g2_sum(void* out, const void* pts, size_t count) {
  try {
    if (!out || (!pts && count)) return ALEO_MI355X_ERR_BAD_ARG;
    return g2_sum_host((uint64_t*)out, (const uint64_t*)pts, count);
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
    FIND_BASES(handle)
    uint64_t jac[18];
    int32_t rc = msm_host_scalars(c, jac, pb, coeffs, n, true);
    if (rc) return rc;
    jacobian_to_affine104(out104, jac);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_kzg_commit_device(void* out104, uint64_t handle, const void* d_coeffs, size_t n, void* stream) {
  try {
    if (!out104 || (!d_coeffs && n)) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    FIND_BASES(handle)
    uint64_t jac[18];
    PICK_STREAM(s)
    int32_t rc = msm_run1(c, jac, pb, d_coeffs, n, true, s);
    if (rc) return rc;
    jacobian_to_affine104(out104, jac);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

// ---- one prover round's commitments in one call -------------------------------------------------------------------
static int32_t batch_args_ok(const void* out, const void* const* ptrs, const size_t* lens, size_t k) {
  if (k == 0) return ALEO_MI355X_OK;
  if (!out || !ptrs || !lens) { g_last_error = "batch: null argument"; return ALEO_MI355X_ERR_BAD_ARG; }
  for (size_t q = 0; q < k; ++q) if (!ptrs[q] && lens[q]) { g_last_error = "batch: null vector with a non-zero length"; return ALEO_MI355X_ERR_BAD_ARG; }
  return ALEO_MI355X_OK;
}
static void jac_to_affine_rows(void* out104, const uint64_t* jac, size_t k) {
  for (size_t q = 0; q < k; ++q) jacobian_to_affine104((uint8_t*)out104 + 104 * q, jac + 18 * q);
}

int32_t aleo_mi355x_msm_g1_batch_device(void* out_jac, uint64_t handle, const void* const* d_scalars, const size_t* lens, size_t k, void* stream) {
  try {
    int32_t rc = batch_args_ok(out_jac, d_scalars, lens, k); if (rc || !k) return rc;
    API_BEGIN
    FIND_BASES(handle)
    std::vector<MsmSeg> sg(k); for (size_t q = 0; q < k; ++q) { sg[q].d_ptr = d_scalars[q]; sg[q].len = lens[q]; sg[q].out = (uint32_t)q; }
    MsmJob j; j.segs = sg.data(); j.nseg = (uint32_t)k; j.k = (uint32_t)k; j.mont = false;
    if (k >= (1u << 20)) { g_last_error = "batch: too many vectors"; return ALEO_MI355X_ERR_BAD_ARG; }
    PICK_STREAM(s)
    return msm_batch(c, (uint64_t*)out_jac, pb, j, s);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_kzg_commit_batch_device(void* out104, uint64_t handle, const void* const* d_coeffs, const size_t* lens, size_t k, void* stream) {
  try {
    int32_t rc = batch_args_ok(out104, d_coeffs, lens, k); if (rc || !k) return rc;
    if (k >= (1u << 20)) { g_last_error = "batch: too many vectors"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    FIND_BASES(handle)
    std::vector<uint64_t> jac(18 * k);
    std::vector<MsmSeg> sg(k); for (size_t q = 0; q < k; ++q) { sg[q].d_ptr = d_coeffs[q]; sg[q].len = lens[q]; sg[q].out = (uint32_t)q; }
    MsmJob j; j.segs = sg.data(); j.nseg = (uint32_t)k; j.k = (uint32_t)k; j.mont = true;
    PICK_STREAM(s)
    if ((rc = msm_batch(c, jac.data(), pb, j, s))) return rc;
    jac_to_affine_rows(out104, jac.data(), k);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_kzg_commit_batch(void* out104, uint64_t handle, const void* const* coeffs, const size_t* lens, size_t k) {
  try {
    int32_t rc = batch_args_ok(out104, coeffs, lens, k); if (rc || !k) return rc;
    if (k >= (1u << 20)) { g_last_error = "batch: too many vectors"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    FIND_BASES(handle)
    size_t total = 0; for (size_t q = 0; q < k; ++q) total += lens[q];
    if ((rc = c->scalars_stage.reserve((total ? total : 1) * 32))) return rc;
    std::vector<const void*> dptr(k); size_t off = 0;
    for (size_t q = 0; q < k; ++q) {          // one staging buffer, k uploads queued back to back on the slot's stream
      dptr[q] = (const char*)c->scalars_stage.p + off * 32;
      if (lens[q]) HIPCHK(hipMemcpyAsync((char*)c->scalars_stage.p + off * 32, coeffs[q], lens[q] * 32, hipMemcpyHostToDevice, c->stream));
      off += lens[q];
    }
    std::vector<uint64_t> jac(18 * k);
    std::vector<MsmSeg> sg(k); for (size_t q = 0; q < k; ++q) { sg[q].d_ptr = dptr[q]; sg[q].len = lens[q]; sg[q].out = (uint32_t)q; }
    MsmJob j; j.segs = sg.data(); j.nseg = (uint32_t)k; j.k = (uint32_t)k; j.mont = true;
    if ((rc = msm_batch(c, jac.data(), pb, j, c->stream))) return rc;
    jac_to_affine_rows(out104, jac.data(), k);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

// SonicKZG10::commit shape: every commitment is a sum of segments (coefficient vector x base offset) over ONE pinned set.
static int32_t commit_segments(Ctx* c, Device* d, void* out104, size_t n_out, uint64_t handle, const aleo_mi355x_commit_segment* segs, size_t n_segs,
                               bool host_scalars, hipStream_t s, bool sparse = false) {
  if (!n_out) return ALEO_MI355X_OK;
  if (!out104 || (!segs && n_segs) || n_out >= (1u << 20) || n_segs >= (1u << 22)) { g_last_error = "commit_segments: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
  std::shared_ptr<PinnedOwner> keep; PinnedBases pb; { int32_t rcb = find_bases(d, handle, &keep, &pb); if (rcb) return rcb; }
  std::vector<MsmSeg> sg(n_segs); size_t total = 0; int32_t rc;
  for (size_t q = 0; q < n_segs; ++q) {
    if ((!segs[q].scalars && segs[q].len) || segs[q].output >= n_out || segs[q].base_offset + segs[q].len > pb.n) { g_last_error = "commit_segments: segment out of range"; return ALEO_MI355X_ERR_BAD_ARG; }
    sg[q].d_ptr = segs[q].scalars; sg[q].len = segs[q].len; sg[q].off = segs[q].base_offset; sg[q].out = segs[q].output; total += segs[q].len;
  }
  if (host_scalars) {          // one staging buffer, the uploads queued back to back on the slot's stream
    if ((rc = c->scalars_stage.reserve((total ? total : 1) * 32))) return rc;
    size_t off = 0;
    for (size_t q = 0; q < n_segs; ++q) {
      if (sg[q].len) HIPCHK(hipMemcpyAsync((char*)c->scalars_stage.p + off * 32, segs[q].scalars, sg[q].len * 32, hipMemcpyHostToDevice, s));
      sg[q].d_ptr = (const char*)c->scalars_stage.p + off * 32; off += sg[q].len;
    }
  }
  std::vector<uint64_t> jac(18 * n_out);
  MsmJob j; j.segs = sg.data(); j.nseg = (uint32_t)n_segs; j.k = (uint32_t)n_out; j.mont = true; j.sparse = sparse;
  if ((rc = msm_batch(c, jac.data(), pb, j, s))) return rc;
  jac_to_affine_rows(out104, jac.data(), n_out);
  return ALEO_MI355X_OK;
}

int32_t aleo_mi355x_kzg_commit_segments(void* out104, size_t n_out, uint64_t handle, const aleo_mi355x_commit_segment* segs, size_t n_segs) {
  try { API_BEGIN return commit_segments(c, d, out104, n_out, handle, segs, n_segs, true, c->stream); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}
int32_t aleo_mi355x_kzg_commit_segments_device(void* out104, size_t n_out, uint64_t handle, const aleo_mi355x_commit_segment* segs, size_t n_segs, void* stream) {
  try { API_BEGIN PICK_STREAM(s) return commit_segments(c, d, out104, n_out, handle, segs, n_segs, false, s); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_kzg_commit_segments_sparse_device(void* out104, size_t n_out, uint64_t handle, const aleo_mi355x_commit_segment* segs, size_t n_segs, void* stream) {
  try { API_BEGIN PICK_STREAM(s) return commit_segments(c, d, out104, n_out, handle, segs, n_segs, false, s, true); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_kzg_commit_hiding(void* out104, uint64_t h_powers, const void* coeffs, size_t n, uint64_t h_gamma, const void* blind, size_t m) {
  try {
    if (!out104 || (!coeffs && n) || (!blind && m)) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    std::shared_ptr<PinnedOwner> kp, kg; PinnedBases pp, pg; int32_t rc;
    if ((rc = find_bases(d, h_powers, &kp, &pp)) || (rc = find_bases(d, h_gamma, &kg, &pg))) return rc;
    uint64_t parts[36];
    if ((rc = msm_host_scalars(c, parts, pp, coeffs, n, true))) return rc;
    if ((rc = msm_host_scalars(c, parts + 18, pg, blind, m, true))) return rc;
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
    return run_enqueue(c, stream, [&](hipStream_t s) { return fr_vec_op(c, d_dst, d_a, d_b, n, op, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_lin_device(void* d_dst, size_t n, const void* c0_mont, const void* c1_mont, const void* d_a, const void* c2_mont, const void* d_b, void* stream) {
  try {
    if (!d_dst && n) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return fr_lin(c, d_dst, n, c0_mont, c1_mont, d_a, c2_mont, d_b, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_powers_device(void* d_dst, size_t n, const void* first_mont, const void* ratio_mont, void* stream) {
  try {
    if ((!d_dst && n) || !first_mont || !ratio_mont) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return fr_powers(c, d_dst, n, first_mont, ratio_mont, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_gather_mul_device(void* d_dst, size_t n, const void* d_scale, const void* d_table1, const void* d_idx1, const void* d_table2, const void* d_idx2, void* stream) {
  try {
    if (n && (!d_dst || !d_table1 || !d_idx1 || (d_table2 && !d_idx2))) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return fr_gather_mul(c, d_dst, n, d_scale, d_table1, d_idx1, d_table2, d_idx2, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_eval_batch_device(void* d_out, const void* const* d_polys, const size_t* lens, const void* z_mont, size_t k, void* stream) {
  try {
    if (k && (!d_out || !d_polys || !lens || !z_mont)) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return fr_eval_batch(c, d_out, d_polys, lens, z_mont, k, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_varuna_prove(const aleo_mi355x_varuna_index* index, const void* const* assignments, size_t n_instances, const uint8_t* seed, void* out_proof, size_t* len) {
  try {
    if (!index || !assignments || !out_proof || !len || !seed || !index->positions || !index->vk_bytes) { g_last_error = "varuna_prove: null argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    for (size_t i = 0; i < n_instances && i < 32; ++i) if (!assignments[i]) { g_last_error = "varuna_prove: null assignment"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    FIND_BASES(index->committer_key)
    return varuna_prove(c, pb, *index, assignments, n_instances, seed, (uint8_t*)out_proof, len);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

static int32_t find_varuna(Device* d, uint64_t handle, std::shared_ptr<VarunaIndexOwner>* keep) {
  std::lock_guard<std::mutex> lk(d->mu);
  auto it = d->varuna.find(handle);
  if (it == d->varuna.end()) { g_last_error = "unknown index handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
  *keep = it->second; return ALEO_MI355X_OK;
}

int32_t aleo_mi355x_varuna_index_build(uint64_t* index_handle, uint64_t committer_key, uint64_t max_degree, uint64_t gamma_offset, uint64_t lagrange_offset,
                                       const aleo_mi355x_r1cs_matrix abc[3], size_t n_constraints, size_t n_public, size_t n_private, uint32_t domain_flags) {
  try {
    if (!index_handle || !abc || domain_flags > 2) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    FIND_BASES(committer_key)
    VarunaIndexOwner* raw = nullptr;
    int32_t rc = varuna_index_build(c, pb, keep, committer_key, max_degree, gamma_offset, lagrange_offset, abc, n_constraints, n_public, n_private, domain_flags, &raw);
    if (rc) return rc;
    std::shared_ptr<VarunaIndexOwner> o(raw, varuna_index_delete);
    std::lock_guard<std::mutex> g(d->mu);
    *index_handle = d->next_varuna++; d->varuna[*index_handle] = std::move(o);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_varuna_index_export(uint64_t index_handle, aleo_mi355x_varuna_index* out) {
  try {
    if (!out) return ALEO_MI355X_ERR_BAD_ARG;
    Device* d = nullptr; int32_t rc = get_device(&d); if (rc) return rc;
    std::shared_ptr<VarunaIndexOwner> keep; if ((rc = find_varuna(d, index_handle, &keep))) return rc;
    *out = *varuna_index_view(keep.get());
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_varuna_index_vk(uint64_t index_handle, void* out, size_t* len) {
  try {
    if (!out || !len) return ALEO_MI355X_ERR_BAD_ARG;
    Device* d = nullptr; int32_t rc = get_device(&d); if (rc) return rc;
    std::shared_ptr<VarunaIndexOwner> keep; if ((rc = find_varuna(d, index_handle, &keep))) return rc;
    const std::vector<uint8_t>& vk = varuna_index_vk(keep.get());
    if (*len < vk.size()) { *len = vk.size(); g_last_error = "index_vk: output buffer too small"; return ALEO_MI355X_ERR_BAD_ARG; }
    std::memcpy(out, vk.data(), vk.size()); *len = vk.size();
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_varuna_index_free(uint64_t index_handle) {
  try {
    Device* d = nullptr; int32_t rc = get_device(&d); if (rc) return rc;
    std::shared_ptr<VarunaIndexOwner> dead;              // freed after the lock is dropped, once no proof uses it
    std::lock_guard<std::mutex> lk(d->mu);
    auto it = d->varuna.find(index_handle);
    if (it == d->varuna.end()) { g_last_error = "unknown index handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
    dead = std::move(it->second); d->varuna.erase(it);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_varuna_prove_indexed(uint64_t index_handle, const void* const* assignments, size_t n_instances, const uint8_t* seed, void* out_proof, size_t* len) {
  try {
    if (!assignments || !out_proof || !len || !seed) return ALEO_MI355X_ERR_BAD_ARG;
    for (size_t i = 0; i < n_instances && i < 32; ++i) if (!assignments[i]) { g_last_error = "varuna_prove: null assignment"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    std::shared_ptr<VarunaIndexOwner> ixk; { int32_t rci = find_varuna(d, index_handle, &ixk); if (rci) return rci; }
    const aleo_mi355x_varuna_index* ix = varuna_index_view(ixk.get());
    FIND_BASES(ix->committer_key)
    return varuna_prove(c, pb, *ix, assignments, n_instances, seed, (uint8_t*)out_proof, len);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_varuna_prove_batch_indexed(const uint64_t* index_handles, size_t n_circuits, const void* const* assignments, const size_t* n_instances, const uint8_t* seed,
                                               void* out_proof, size_t* len) {
  try {
    if (!seed || !index_handles || !assignments || !n_instances || !out_proof || !len || n_circuits < 1 || n_circuits > 32) { g_last_error = "varuna_prove_batch: null argument or circuit count outside 1..32"; return ALEO_MI355X_ERR_BAD_ARG; }
    size_t total = 0;
    for (size_t j = 0; j < n_circuits; ++j) { if (n_instances[j] < 1 || n_instances[j] > 32) { g_last_error = "varuna_prove_batch: 1..32 instances per circuit"; return ALEO_MI355X_ERR_BAD_ARG; } total += n_instances[j]; }
    for (size_t i = 0; i < total; ++i) if (!assignments[i]) { g_last_error = "varuna_prove_batch: null assignment"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    std::vector<std::shared_ptr<VarunaIndexOwner>> ixk(n_circuits); std::vector<const aleo_mi355x_varuna_index*> views(n_circuits);
    for (size_t j = 0; j < n_circuits; ++j) { int32_t rci = find_varuna(d, index_handles[j], &ixk[j]); if (rci) return rci; views[j] = varuna_index_view(ixk[j].get()); }
    FIND_BASES(views[0]->committer_key)
    return varuna_prove_batch(c, pb, views.data(), n_circuits, assignments, n_instances, seed, (uint8_t*)out_proof, len);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_varuna_prove_many(aleo_mi355x_prove_request* requests, size_t n_requests) {
  try {
    if (!requests || n_requests < 1 || n_requests > 64) { g_last_error = "varuna_prove_many: 1..64 requests"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    std::vector<ProveRequest> rq(n_requests); std::vector<std::vector<std::shared_ptr<VarunaIndexOwner>>> keep_ix(n_requests);
    uint64_t key = 0; bool have_key = false;
    for (size_t p = 0; p < n_requests; ++p) {
      aleo_mi355x_prove_request& r = requests[p]; r.status = ALEO_MI355X_OK;
      auto bad = [&](const char* why) { r.status = ALEO_MI355X_ERR_BAD_ARG; rq[p].status = r.status; rq[p].error = why; };
      if (!r.index_handles || !r.assignments || !r.n_instances || !r.seed || !r.out_proof || r.n_circuits < 1 || r.n_circuits > 32) { bad("varuna_prove_many: null argument or circuit count outside 1..32"); continue; }
      size_t total = 0; bool ok = true;
      for (size_t j = 0; j < r.n_circuits; ++j) { if (r.n_instances[j] < 1 || r.n_instances[j] > 32) ok = false; total += r.n_instances[j]; }
      for (size_t i = 0; ok && i < total; ++i) if (!r.assignments[i]) ok = false;
      if (!ok) { bad("varuna_prove_many: 1..32 instances per circuit, no null assignment"); continue; }
      keep_ix[p].resize(r.n_circuits);
      for (size_t j = 0; j < r.n_circuits && !rq[p].status; ++j) {
        const int32_t rci = find_varuna(d, r.index_handles[j], &keep_ix[p][j]);
        if (rci) { r.status = rci; rq[p].status = rci; rq[p].error = g_last_error; break; }
        rq[p].ixs.push_back(varuna_index_view(keep_ix[p][j].get()));
      }
      if (rq[p].status) continue;
      if (!have_key) { key = rq[p].ixs[0]->committer_key; have_key = true; }
      if (rq[p].ixs[0]->committer_key != key) { bad("varuna_prove_many: every request must use indexes of ONE committer key"); continue; }
      rq[p].assignments = r.assignments; rq[p].ks = r.n_instances; rq[p].seed32 = r.seed; rq[p].out = (uint8_t*)r.out_proof; rq[p].out_len = &r.len;
    }
    if (!have_key) { g_last_error = rq[0].error; return rq[0].status ? rq[0].status : ALEO_MI355X_ERR_BAD_ARG; }
    FIND_BASES(key)
    std::vector<ProveRequest> live; std::vector<size_t> where;
    for (size_t p = 0; p < n_requests; ++p) if (!rq[p].status) { live.push_back(rq[p]); where.push_back(p); }
    // The call runs as up to FOUR lockstep groups (ALEO_MI355X_LOCKSTEP_GROUPS, default 4) on as many threads (the caller's and one per further group, each on a context of
    // its own, whose main stream has a hardware queue of its own: "streams" above): while one group's commitments hold the card, the other groups' field kernels, sorts,
    // reductions and transcripts — a third of a lockstep round — run in their shadow.  Up to four proofs that is one proof per group; from five on the groups hold two or more
    // and batch their commitments.  Measured (2^15 constraints, same box; profiles/r05_lockstep_groups_ab.txt, r05_lockstep_retune*.txt, r05_group_min_ab.txt, r05_group_from_ab.txt):
    // 8 proofs 32.7 ms as one group, 27.3 as four; 2 proofs 10.4 ms in lockstep, 8.55 as two groups of one; 3: 13.5 -> 11.6; 4: 15.1 -> 14.5; 8 groups of one: 29.6.
    // ALEO_MI355X_LOCKSTEP_GROUPS=1: one group; ALEO_MI355X_LOCKSTEP_GROUP_FROM / _GROUP_MIN: fewest proofs per call that is split / per group.  Proof bytes do not depend on any of it.
    static const int groups_env = [] { const char* e = std::getenv("ALEO_MI355X_LOCKSTEP_GROUPS"); const int k = e ? std::atoi(e) : 4; return k >= 1 && k <= 4 ? k : 4; }();
    int32_t rc = ALEO_MI355X_OK; bool split_done = false;
    static const size_t group_min = [] { const char* e = std::getenv("ALEO_MI355X_LOCKSTEP_GROUP_MIN"); const int k = e ? std::atoi(e) : 1; return (size_t)(k >= 1 && k <= 64 ? k : 1); }();      // fewest proofs per group
    static const size_t group_from = [] { const char* e = std::getenv("ALEO_MI355X_LOCKSTEP_GROUP_FROM"); const int k = e ? std::atoi(e) : 2; return (size_t)(k >= 2 && k <= 64 ? k : 2); }();      // fewest proofs per call that is split into groups
    const size_t want_groups = live.size() >= group_from ? std::min<size_t>((size_t)groups_env, live.size() / group_min) : 1;
    if (want_groups >= 2) {
      // contexts for groups 1..: never waited for (whatever is free now); fewer groups if fewer are free
      std::vector<Ctx*> gc{c}; std::vector<std::unique_lock<std::mutex>> glk;
      for (size_t g = 1; g < want_groups; ++g) {
        Ctx* c2 = nullptr; std::unique_lock<std::mutex> lk2;
        if (acquire_other(d, c, &c2, lk2, false) != ALEO_MI355X_OK || !c2) break;
        gc.push_back(c2); glk.push_back(std::move(lk2));
      }
      const size_t G = gc.size();
      if (G >= 2) {
        std::vector<std::vector<ProveRequest>> grp(G); std::vector<std::vector<size_t>> at(G);
        for (size_t i = 0; i < live.size(); ++i) { const size_t g = i * G / live.size(); grp[g].push_back(live[i]); at[g].push_back(i); }      // contiguous, balanced
        std::vector<int32_t> rcs(G, ALEO_MI355X_OK); std::vector<std::string> errs(G); std::vector<std::thread> th; bool started = true;
        // inside a group ONE thread runs the group's proofs (ALEO_MI355X_GROUP_WORKERS): the groups are the call's parallelism, and four busy streams beat eight (profiles/r05_lockstep_retune.txt: 29.4 / 28.5 / 27.5 ms per 8 proofs with 4 / 2 / 1 workers per group, 54.0 / 53.9 / 52.8 per 16)
        static const int group_workers = [] { const char* e = std::getenv("ALEO_MI355X_GROUP_WORKERS"); const int k = e ? std::atoi(e) : 1; return k >= 1 && k <= MAX_SLOTS + 1 ? k : 1; }();
        for (size_t g = 1; g < G && started; ++g) {
          try {
            th.emplace_back([&, g] {
              try {
                if (hipSetDevice(d->device) != hipSuccess) { rcs[g] = ALEO_MI355X_ERR_HIP; errs[g] = "hipSetDevice failed"; return; }
                rcs[g] = varuna_prove_many(gc[g], pb, grp[g], group_workers); if (rcs[g]) errs[g] = g_last_error;
              } catch (...) { rcs[g] = ALEO_MI355X_ERR_HIP; errs[g] = "varuna_prove_many: exception in a lockstep group"; }
            });
          } catch (...) { started = false; }
        }
        // (a thread that could not be started: its group and the ones behind it run here, after group 0)
        try { rcs[0] = varuna_prove_many(c, pb, grp[0], group_workers); if (rcs[0]) errs[0] = g_last_error; } catch (...) { rcs[0] = ALEO_MI355X_ERR_HIP; errs[0] = "varuna_prove_many: exception in the first group"; }      // never unwind past the joinable threads
        for (auto& t : th) t.join();
        for (size_t g = th.size() + 1; g < G; ++g) { try { rcs[g] = varuna_prove_many(c, pb, grp[g], group_workers); if (rcs[g]) errs[g] = g_last_error; } catch (...) { rcs[g] = ALEO_MI355X_ERR_HIP; errs[g] = "varuna_prove_many: exception in a lockstep group"; } }
        for (size_t g = 0; g < G; ++g) { for (size_t i = 0; i < grp[g].size(); ++i) live[at[g][i]] = grp[g][i]; if (rcs[g] && !rc) { rc = rcs[g]; g_last_error = errs[g]; } }
        split_done = true;
      }
    }
    if (!split_done) rc = live.empty() ? ALEO_MI355X_OK : varuna_prove_many(c, pb, live);
    std::string first_error;
    for (size_t i = 0; i < live.size(); ++i) { requests[where[i]].status = live[i].status; if (live[i].status && first_error.empty()) first_error = live[i].error; }
    for (size_t p = 0; p < n_requests; ++p) if (rq[p].status && first_error.empty()) first_error = rq[p].error;
    if (!first_error.empty()) g_last_error = first_error;
    return rc;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_varuna_last_timing(double* out_ms, int32_t cap) {
  int32_t n = cap < 8 ? cap : 8;
  for (int32_t i = 0; i < n; ++i) out_ms[i] = g_varuna_timing[i];
  return n;
}

int32_t aleo_mi355x_fr_random_device(void* d_dst, size_t n, const uint8_t* seed, uint64_t first_index, int32_t montgomery, void* stream) {
  try {
    if ((!d_dst && n) || !seed) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return fr_random(c, d_dst, n, seed, first_index, montgomery, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_lincomb_device(void* d_dst, size_t n, const void* c0_mont, const void* const* d_terms, const size_t* lens, const void* coeffs_mont, size_t k, void* stream) {
  try {
    if ((!d_dst && n) || (k && (!d_terms || !lens || !coeffs_mont))) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return fr_lincomb(c, d_dst, n, c0_mont, d_terms, lens, coeffs_mont, k, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_ahp_first_sumcheck_device(void* d_dst, size_t n, const void* d_r, const void* d_za, const void* d_zb, const void* d_t, const void* d_z,
                                              const void* eta_b_mont, const void* eta_c_mont, void* stream) {
  try {
    if (n && (!d_dst || !d_r || !d_za || !d_zb || !d_t || !d_z || !eta_b_mont || !eta_c_mont)) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return ahp_first_sumcheck(c, d_dst, n, d_r, d_za, d_zb, d_t, d_z, eta_b_mont, eta_c_mont, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_ahp_matrix_sumcheck_device(void* d_dst, size_t n, const void* const* d_index, size_t index_stride, const void* const* d_f, const void* consts_mont, void* stream) {
  try {
    if (n && (!d_dst || !d_index || !d_f || !consts_mont || (d_index[0] && !d_f[0]) || (d_index[1] && !d_f[1]) || (d_index[2] && !d_f[2]))) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return ahp_matrix_sumcheck(c, d_dst, n, d_index, index_stride, d_f, consts_mont, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_blind_rows_device(void* d_dst, const void* d_src, size_t n, size_t rows, const void* rho_mont, void* stream) {
  try {
    if (rows && n && (!d_dst || !d_src || !rho_mont)) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return fr_blind_rows(c, d_dst, d_src, n, rows, rho_mont, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_ahp_sumcheck_operands_device(void* d_dst, const void* d_witness_polys, const void* d_x_polys, size_t n, size_t n_x, size_t instances, void* stream) {
  try {
    if (instances && n && (!d_dst || !d_witness_polys || (!d_x_polys && n_x))) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return ahp_sumcheck_operands(c, d_dst, d_witness_polys, d_x_polys, n, n_x, instances, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_batch_inverse_device(void* d_inout, size_t n, void* stream) {
  try {
    if (!d_inout && n) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return fr_batch_inverse(c, d_inout, n, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_divide_by_linear_device(void* d_quotient, void* d_eval, const void* d_poly, size_t n, const void* z_mont, void* stream) {
  try {
    if ((!d_poly && n) || (!d_quotient && !d_eval)) return ALEO_MI355X_ERR_BAD_ARG;
    if (!z_mont || (d_quotient && d_quotient == d_poly)) { g_last_error = "fr_divide_by_linear_device: null point, or quotient aliases the polynomial"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return fr_divide_by_linear(c, d_quotient, d_eval, d_poly, n, z_mont, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

// KZG10::open for one polynomial at one point: witness polynomial on the device (slot scratch), then its commitment.
int32_t aleo_mi355x_kzg_open_device(void* out_affine104, void* out_eval_mont, uint64_t handle, const void* d_poly_mont, size_t n, const void* z_mont, void* stream) {
  try {
    if (!out_affine104 || !z_mont || (!d_poly_mont && n)) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    FIND_BASES(handle)
    PICK_STREAM(s)
    int32_t rc; if ((rc = c->ntt_stage.reserve((n ? n : 1) * 32 + 32))) return rc;
    char* q = c->ntt_stage.as<char>(); char* ev = q + (n ? n : 1) * 32;
    if ((rc = fr_divide_by_linear(c, q, ev, d_poly_mont, n, z_mont, s))) return rc;
    if (out_eval_mont) HIPCHK(hipMemcpyAsync(out_eval_mont, ev, 32, hipMemcpyDeviceToHost, s));
    uint64_t jac[18];
    if ((rc = msm_run1(c, jac, pb, q, n ? n - 1 : 0, true, s))) return rc;      // synchronises: the evaluation has landed too
    if (n <= 1) HIPCHK(hipStreamSynchronize(s));
    jacobian_to_affine104(out_affine104, jac);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_spmv_device(void* d_y, const void* d_row_ptr, const void* d_col_idx, const void* d_vals, const void* d_x, size_t rows, void* stream) {
  try {
    if ((!d_y || !d_row_ptr) && rows) return ALEO_MI355X_ERR_BAD_ARG;
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return fr_spmv(c, d_y, d_row_ptr, d_col_idx, d_vals, d_x, rows, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_ntt_fr(void* inout, uint32_t lg_n, int32_t order, int32_t direction, int32_t type) {
  try {
    if (!inout || lg_n > 30 || order < 0 || order > 3 || direction < 0 || direction > 1 || type < 0 || type > 1) { g_last_error = "ntt_fr: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    size_t bytes = ((size_t)1 << lg_n) * 32;
    int32_t rc; if ((rc = c->ntt_stage.reserve(bytes))) return rc;
    HIPCHK(hipMemcpyAsync(c->ntt_stage.p, inout, bytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = ntt_run(c, c->ntt_stage.p, lg_n, 1, order, direction, type, c->stream))) return rc;
    HIPCHK(hipMemcpyAsync(inout, c->ntt_stage.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_ntt_fr_device(void* d_inout, uint32_t lg_n, int32_t order, int32_t direction, int32_t type, void* stream) {
  try {
    if (!d_inout || lg_n > 30 || order < 0 || order > 3 || direction < 0 || direction > 1 || type < 0 || type > 1) { g_last_error = "ntt_fr_device: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return ntt_run(c, d_inout, lg_n, 1, order, direction, type, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_ntt_fr_batch_device(void* d_inout, uint32_t lg_n, size_t batch, int32_t order, int32_t direction, int32_t type, void* stream) {
  try {
    if ((!d_inout && batch) || lg_n > 30 || order < 0 || order > 3 || direction < 0 || direction > 1 || type < 0 || type > 1) { g_last_error = "ntt_fr_batch_device: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return ntt_run(c, d_inout, lg_n, batch, order, direction, type, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_ntt_fr_from_device(void* d_out, const void* d_src, size_t src_stride, size_t src_len, uint32_t lg_n, size_t batch, int32_t direction, int32_t type, void* stream) {
  try {
    if (((!d_out || !d_src) && batch) || lg_n > 30 || direction < 0 || direction > 1 || type < 0 || type > 1 || src_len > ((size_t)1 << lg_n) || src_stride >= (1ull << 32)) {
      g_last_error = "ntt_fr_from_device: bad argument"; return ALEO_MI355X_ERR_BAD_ARG;
    }
    if (batch) {                                             // the output must not overlap the source (the first pass reads while later blocks of the same launch already write)
      const char* o0 = (const char*)d_out; const char* o1 = o0 + ((batch << lg_n) * 32);
      const char* s0 = (const char*)d_src; const char* s1 = s0 + ((batch - 1) * src_stride + src_len) * 32;
      if (src_len && o0 < s1 && s0 < o1) { g_last_error = "ntt_fr_from_device: d_out overlaps d_src"; return ALEO_MI355X_ERR_BAD_ARG; }
    }
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return ntt_run_from(c, d_out, d_src, src_stride, src_len, lg_n, batch, direction, type, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_grid_scale_device(void* d_data, uint32_t lg_n, uint64_t rows, uint64_t cols, uint64_t row0, uint64_t col0, uint64_t ld,
                                         int32_t mode, int32_t direction, void* stream) {
  try {
    if ((!d_data && rows && cols) || lg_n == 0 || lg_n > 32 || mode < 0 || mode > 1 || direction < 0 || direction > 1) { g_last_error = "fr_grid_scale_device: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return fr_grid_scale(c, d_data, lg_n, rows, cols, row0, col0, ld, mode, direction, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fr_transpose_device(void* d_dst, const void* d_src, uint64_t rows, uint64_t cols, void* stream) {
  try {
    if ((!d_dst || !d_src) && rows && cols) { g_last_error = "fr_transpose_device: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    if (d_dst == d_src && rows > 1 && cols > 1) { g_last_error = "fr_transpose_device: in place is not supported"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    return run_enqueue(c, stream, [&](hipStream_t s) { return fr_transpose(c, d_dst, d_src, rows, cols, s); });
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_fq_mul(void* r, const void* a, const void* b, size_t n) {
  try { if ((!r || !a || !b) && n) return ALEO_MI355X_ERR_BAD_ARG; API_BEGIN return launch_fq_mul(c, r, a, b, n); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}
int32_t aleo_mi355x_fr_mul(void* r, const void* a, const void* b, size_t n) {
  try { if ((!r || !a || !b) && n) return ALEO_MI355X_ERR_BAD_ARG; API_BEGIN return launch_fr_mul(c, r, a, b, n); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_selftest_madd28(uint32_t lanes, uint32_t steps, uint64_t seed, uint32_t* failures) {
  try { if (!failures) return ALEO_MI355X_ERR_BAD_ARG; API_BEGIN return selftest_madd28(c, lanes, steps, seed, failures); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}
int32_t aleo_mi355x_selftest_addquad(uint32_t ops, uint64_t seed, uint32_t* failures) {
  try { if (!failures || !ops || ops > (1u << 22)) return ALEO_MI355X_ERR_BAD_ARG; API_BEGIN return selftest_addquad(c, ops, seed, failures); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_selftest_g2pair(const void* affine192, uint32_t n_points, uint32_t n_pairs, uint32_t* failures2) {
  try { if (!affine192 || !failures2 || n_points < 3 || !n_pairs || n_pairs > (1u << 20)) return ALEO_MI355X_ERR_BAD_ARG; API_BEGIN return selftest_g2pair(c, affine192, n_points, n_pairs, failures2); } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_last_msm_timing(double* out_ms, int32_t cap) {
  try {
    if (!out_ms || cap <= 0) return 0;
    const MsmTiming& t = g_last_msm;          // of the calling thread's most recent MSM
    double v[7] = {t.total, t.sort, t.accum, t.reduce, t.host, t.accum_kernel, (double)t.accum_launches};
    int32_t k = cap < 7 ? cap : 7;
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
    case ALEO_MI355X_ERR_UNSATISFIED: return "assignment does not satisfy the circuit";
    default: return "unknown status";
  }
}
// The sizes from which the drop-in's two arms should take the GPU: measured crossovers of the COLD one-shot calls against the CPU
// path on the same box (bench.py cpu_baseline.crossover, profiles/r04_crossover.json), overridable per deployment.
static size_t env_size(const char* name, size_t dflt) { const char* e = std::getenv(name); if (!e || !*e) return dflt; char* end = nullptr; const unsigned long long v = std::strtoull(e, &end, 10); return end && *end == 0 ? (size_t)v : dflt; }
size_t aleo_mi355x_min_msm(void) { return env_size("ALEO_MI355X_MIN_MSM", (size_t)1 << 10); }
size_t aleo_mi355x_min_ntt(void) { return env_size("ALEO_MI355X_MIN_NTT", (size_t)1 << 12); }
int32_t aleo_mi355x_selftest_host_inverse(uint32_t count, uint64_t seed, uint32_t* failures, double* ns_per_inverse) {
  try {
    if (!failures) { g_last_error = "selftest_host_inverse: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    *failures = 0;
    inverse_selftest<6>(count, seed, failures, ns_per_inverse);
    inverse_selftest<4>(count, seed, failures, ns_per_inverse ? ns_per_inverse + 2 : nullptr);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}
const char* aleo_mi355x_last_error(void) { return g_last_error.c_str(); }
const char* aleo_mi355x_version(void) { return "aleo_mi355x 0.2.0 (gfx950)"; }


// ---- one MSM over several devices (BASELINE configs[4]) ------------------------------------------------------------------
// One process, G device contexts: the base set is cut into G contiguous shards, shard g pinned on devices[g]; an MSM runs the whole Pippenger
// per shard on its device (one host thread per shard: the runtime's current device is per thread) and the G partial sums — 144 bytes each —
// are added on the host in shard order, so the result's bytes do not depend on which device finished first.  No collective: inside one
// process the "all-gather" of SURVEY.md 8(e) is G stores into one host array.  (Ranks in separate processes exchange the same 144-byte
// partials over RCCL: aleo_amd/dist.py.)  A device may be listed more than once (how the tests rehearse G > 1 on one card).
}  // extern "C" (reopened below: the helpers are templates)
namespace {
struct ShardedSet { std::vector<int> devices; std::vector<uint64_t> handles; std::vector<size_t> first, count; size_t n = 0; };
std::mutex g_sh_mu; std::map<uint64_t, std::shared_ptr<ShardedSet>> g_sh; uint64_t g_sh_next = 1;

std::shared_ptr<ShardedSet> sharded_find(uint64_t h) {
  std::lock_guard<std::mutex> lk(g_sh_mu); auto it = g_sh.find(h);
  if (it == g_sh.end()) { g_last_error = "unknown sharded handle"; return nullptr; }
  return it->second;
}
// Shard work runs on LONG-LIVED worker threads, one per (device, ordinal among the shards a call lists on that device): created on first use, bound to their
// device once, parked on a condition variable between calls.  (Rounds 3-4 started G std::threads per call: every commitment of a proof against a sharded key
// paid G thread creations — on one card 8 shards cost +19 % per 2^20-constraint proof.)  The pool is never destroyed: its threads are detached and sleep until
// the process ends.  Submission is serialised (g_pool_submit_mu) so that every worker's queue holds the calls in ONE global order — two calls whose shard bodies
// meet at barriers cannot interleave into a deadlock.
struct ShardWorker {
  std::mutex mu; std::condition_variable cv; std::deque<std::function<void()>> q; int device = -1; bool started = false;
  void loop() {
    (void)hipSetDevice(device);                              // a failure shows up again in the task (it sets the device itself and reports)
    for (;;) {
      std::function<void()> job;
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !q.empty(); }); job = std::move(q.front()); q.pop_front(); }
      job();
    }
  }
};
std::mutex g_pool_mu, g_pool_submit_mu; std::map<std::pair<int, size_t>, ShardWorker*> g_pool;
ShardWorker* shard_worker(int device, size_t ordinal) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto it = g_pool.find({device, ordinal});
  if (it != g_pool.end()) return it->second;
  ShardWorker* w = new ShardWorker(); w->device = device;      // leaked on purpose: lives as long as the process
  std::thread([w] { w->loop(); }).detach();                    // (throws std::system_error if no thread can be started: caught by the entry point's try)
  w->started = true; g_pool[{device, ordinal}] = w; return w;
}
// runs f(g) for every shard on that shard's worker with its device current and waits for all of them; the first failure's code and text come back.
// on_skip(g): called for a shard whose body could not run (its device could not be selected, an exception) — a body that meets the other shards at barriers
// passes one that drops the shard from them.
template <class F> int32_t for_each_shard(const ShardedSet& S, F&& f, std::function<void(size_t)> on_skip = nullptr) {
  const size_t G = S.devices.size();
  std::vector<int32_t> rcs(G, ALEO_MI355X_OK); std::vector<std::string> errs(G);
  std::vector<ShardWorker*> ws(G); std::vector<size_t> ordinal(G, 0);
  for (size_t g = 0; g < G; ++g) for (size_t e = 0; e < g; ++e) ordinal[g] += S.devices[e] == S.devices[g];
  try { for (size_t g = 0; g < G; ++g) ws[g] = shard_worker(S.devices[g], ordinal[g]); }
  catch (...) { g_last_error = "could not start a shard's thread"; return ALEO_MI355X_ERR_HIP; }      // nothing was queued: none runs
  std::mutex done_mu; std::condition_variable done_cv; size_t done = 0;
  {
    std::lock_guard<std::mutex> order(g_pool_submit_mu);
    for (size_t g = 0; g < G; ++g) {
      auto body = [&, g]() {
        bool ran = false;
        try {
          if (hipSetDevice(S.devices[g]) != hipSuccess) { rcs[g] = ALEO_MI355X_ERR_HIP; errs[g] = "hipSetDevice failed"; }
          else { ran = true; rcs[g] = f(g); if (rcs[g]) errs[g] = g_last_error; }
        } catch (...) { rcs[g] = ALEO_MI355X_ERR_HIP; errs[g] = "exception in a shard"; ran = false; }
        if (!ran && on_skip) { try { on_skip(g); } catch (...) {} }
        { std::lock_guard<std::mutex> lk(done_mu); ++done; done_cv.notify_one(); }      // notified under the lock: the waiter cannot return (and destroy the condition variable) before the call is over
      };
      { std::lock_guard<std::mutex> lk(ws[g]->mu); ws[g]->q.emplace_back(std::move(body)); }
      ws[g]->cv.notify_one();
    }
  }
  { std::unique_lock<std::mutex> lk(done_mu); done_cv.wait(lk, [&] { return done == G; }); }
  for (size_t g = 0; g < G; ++g) if (rcs[g]) { g_last_error = "shard " + std::to_string(g) + " (device " + std::to_string(S.devices[g]) + "): " + errs[g]; return rcs[g]; }
  return ALEO_MI355X_OK;
}
int32_t sharded_layout(ShardedSet& S, size_t n, const int32_t* devices, size_t n_devices) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { g_last_error = "no HIP device visible"; return ALEO_MI355X_ERR_NO_DEVICE; }
  if (n_devices < 1 || n_devices > 64) { g_last_error = "sharded: 1..64 shards"; return ALEO_MI355X_ERR_BAD_ARG; }
  S.n = n;
  for (size_t g = 0; g < n_devices; ++g) {
    const int dev = devices ? devices[g] : (int)(g % (size_t)count);
    if (dev < 0 || dev >= count) { g_last_error = "sharded: device index out of range"; return ALEO_MI355X_ERR_BAD_ARG; }
    S.devices.push_back(dev);
    const size_t lo = n * g / n_devices, hi = n * (g + 1) / n_devices;      // the split of aleo_amd/dist.py shard_range
    S.first.push_back(lo); S.count.push_back(hi - lo);
  }
  S.handles.assign(n_devices, 0);
  return ALEO_MI355X_OK;
}
uint64_t sharded_register(std::shared_ptr<ShardedSet> S) { std::lock_guard<std::mutex> lk(g_sh_mu); const uint64_t h = g_sh_next++; g_sh[h] = std::move(S); return h; }
void sharded_release(const ShardedSet& S) {
  (void)for_each_shard(S, [&](size_t g) -> int32_t { return S.handles[g] ? aleo_mi355x_bases_unpin(S.handles[g]) : ALEO_MI355X_OK; });
}
}  // namespace
extern "C" {

int32_t aleo_mi355x_bases_pin_sharded(const void* bases, size_t base_stride, size_t n, const int32_t* devices, size_t n_devices, int32_t precompute, uint64_t* handle) {
  try {
    if (!handle || (!bases && n) || (base_stride != 104 && base_stride != 96)) { g_last_error = "bases_pin_sharded: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    auto S = std::make_shared<ShardedSet>();
    int32_t rc = sharded_layout(*S, n, devices, n_devices); if (rc) return rc;
    rc = for_each_shard(*S, [&](size_t g) -> int32_t {
      int32_t r = aleo_mi355x_bases_pin((const uint8_t*)bases + S->first[g] * base_stride, base_stride, S->count[g], &S->handles[g]);
      if (!r && precompute && S->count[g] >= 1024) r = aleo_mi355x_bases_precompute(S->handles[g]);
      return r;
    });
    if (rc) { const std::string keep = g_last_error; sharded_release(*S); g_last_error = keep; return rc; }
    *handle = sharded_register(std::move(S));
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_generate_sharded(const void* base_affine104, uint64_t first_multiple, size_t n, const int32_t* devices, size_t n_devices, int32_t precompute, uint64_t* handle) {
  try {
    if (!handle || !base_affine104) { g_last_error = "bases_generate_sharded: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    auto S = std::make_shared<ShardedSet>();
    int32_t rc = sharded_layout(*S, n, devices, n_devices); if (rc) return rc;
    rc = for_each_shard(*S, [&](size_t g) -> int32_t {
      int32_t r = aleo_mi355x_bases_generate(base_affine104, first_multiple + S->first[g], S->count[g], &S->handles[g]);
      if (!r && precompute && S->count[g] >= 1024) r = aleo_mi355x_bases_precompute(S->handles[g]);
      return r;
    });
    if (rc) { const std::string keep = g_last_error; sharded_release(*S); g_last_error = keep; return rc; }
    *handle = sharded_register(std::move(S));
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_unpin_sharded(uint64_t handle) {
  try {
    std::shared_ptr<ShardedSet> S;
    { std::lock_guard<std::mutex> lk(g_sh_mu); auto it = g_sh.find(handle); if (it == g_sh.end()) { g_last_error = "unknown sharded handle"; return ALEO_MI355X_ERR_BAD_HANDLE; } S = it->second; g_sh.erase(it); }
    sharded_release(*S);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

// shard_info: out[0] = number of shards, then per shard: device, first point, point count (as much as cap allows); returns the number of values written
int32_t aleo_mi355x_bases_sharded_info(uint64_t handle, uint64_t* out, int32_t cap) {
  try {
    auto S = sharded_find(handle); if (!S || !out) return 0;
    int32_t w = 0;
    if (w < cap) out[w++] = S->devices.size();
    for (size_t g = 0; g < S->devices.size(); ++g) { const uint64_t v[3] = {(uint64_t)S->devices[g], S->first[g], S->count[g]}; for (uint64_t x : v) if (w < cap) out[w++] = x; }
    return w;
  } catch (...) { return 0; }
}

// out: the sum as snarkVM's Projective (x, y, 1 / infinity (1, 1, 0)), 144 bytes; partials (optional, G x 144 bytes): each shard's own sum in shard order
int32_t aleo_mi355x_msm_g1_sharded(void* out_jacobian, uint64_t handle, const void* scalars, size_t n, void* partials_out) {
  try {
    if (!out_jacobian || (!scalars && n)) { g_last_error = "msm_g1_sharded: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    auto S = sharded_find(handle); if (!S) return ALEO_MI355X_ERR_BAD_HANDLE;
    if (n > S->n) { g_last_error = "msm_g1_sharded: more scalars than pinned points"; return ALEO_MI355X_ERR_BAD_ARG; }
    const size_t G = S->devices.size();
    std::vector<uint64_t> part(18 * G);
    int32_t rc = for_each_shard(*S, [&](size_t g) -> int32_t {
      const size_t lo = S->first[g] < n ? S->first[g] : n, hi = S->first[g] + S->count[g] < n ? S->first[g] + S->count[g] : n;      // a prefix of the set: shards past n contribute the identity
      return aleo_mi355x_msm_g1_pinned(&part[18 * g], S->handles[g], (const uint8_t*)scalars + lo * 32, hi - lo);
    });
    if (rc) return rc;
    if (partials_out) std::memcpy(partials_out, part.data(), part.size() * 8);
    return aleo_mi355x_g1_sum(out_jacobian, part.data(), G);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

}  // extern "C"
namespace aleo_mi355x {
// ---- commitments against a sharded committer key (row e2: a proof that spans devices) -------------------------------------------------------------------
// The scalar vectors live on ONE device (the prover's: `c`'s); the base set is cut over G devices (ShardedSet).  A segment [off, off + len) meets shard g
// in [max(off, first_g), min(off + len, first_g + count_g)): that piece of the scalars is pulled by device g (peer copy over xGMI — none when g is the
// scalars' own device) and multiplied there by the ordinary batched Pippenger against shard g's points and tables; what crosses back is k partial
// results of 144 bytes per shard, added on the host in shard order.  The sum of normalised partials is normalised again, so the bytes are those of
// the single-device commitment.
int32_t commit_sharded(Ctx* c, uint64_t sharded_handle, const MsmSeg* segs, uint32_t nseg, uint32_t k, bool mont, uint64_t* out_jac18, hipStream_t s, bool s_drain) {
  auto S = sharded_find(sharded_handle); if (!S) return ALEO_MI355X_ERR_BAD_HANDLE;
  const size_t G = S->devices.size(); const int home = c->device;
  for (uint32_t q = 0; q < nseg; ++q) if (segs[q].out >= k || segs[q].off + segs[q].len > S->n) { g_last_error = "commit_sharded: segment out of range"; return ALEO_MI355X_ERR_BAD_ARG; }
  if (s_drain) HIPCHK(hipStreamSynchronize(s));              // the scalars are complete (and whatever the caller queued before the commitment has landed)
  std::vector<uint64_t> part((size_t)18 * k * G);
  std::vector<char> busy(G, 0);                              // shards that hold a piece of some segment
  for (size_t g = 0; g < G; ++g) for (uint32_t q = 0; q < nseg && !busy[g]; ++q) {
    const size_t lo = segs[q].off > S->first[g] ? segs[q].off : S->first[g], hi = segs[q].off + segs[q].len < S->first[g] + S->count[g] ? segs[q].off + segs[q].len : S->first[g] + S->count[g];
    busy[g] = hi > lo;
  }
  std::mutex lend_mu;                                         // shards that find no free context on the caller's device take turns on the caller's own (idle while it waits here)
  const int32_t rc = for_each_shard(*S, [&](size_t g) -> int32_t {
    uint64_t* mine = &part[(size_t)18 * k * g];
    if (!busy[g]) { for (uint32_t q = 0; q < k; ++q) host::hstore_jacobian_normalized(mine + 18 * q, host::HXYZZ::infinity()); return ALEO_MI355X_OK; }
    Device* d = nullptr; Ctx* cc = nullptr; std::unique_lock<std::mutex> lk, lend;
    { int32_t r = get_device(&d); if (r) return r; if ((r = acquire_other(d, c, &cc, lk, d->device != home))) return r; }
    if (!cc) { lend = std::unique_lock<std::mutex>(lend_mu); cc = c; }      // (only on the home device: every other context there may belong to this very call)
    std::shared_ptr<PinnedOwner> keep; PinnedBases pb; { const int32_t r = find_bases(d, S->handles[g], &keep, &pb); if (r) return r; }
    std::vector<MsmSeg> sub; size_t total = 0;
    for (uint32_t q = 0; q < nseg; ++q) {
      const size_t lo = segs[q].off > S->first[g] ? segs[q].off : S->first[g], hi = segs[q].off + segs[q].len < S->first[g] + S->count[g] ? segs[q].off + segs[q].len : S->first[g] + S->count[g];
      if (hi <= lo) continue;
      MsmSeg m; m.d_ptr = (const char*)segs[q].d_ptr + (lo - segs[q].off) * 32; m.len = hi - lo; m.off = lo - S->first[g]; m.out = segs[q].out; sub.push_back(m); total += m.len;
    }
    if (d->device != home) {                                 // pull the pieces: one peer copy each, queued back to back on this shard's stream
      const int32_t r = cc->scalars_stage.reserve(total * 32); if (r) return r;
      size_t at = 0;
      for (auto& m : sub) {
        HIPCHK(hipMemcpyPeerAsync((char*)cc->scalars_stage.p + at * 32, d->device, m.d_ptr, home, m.len * 32, cc->stream));
        m.d_ptr = (const char*)cc->scalars_stage.p + at * 32; at += m.len;
      }
    }
    MsmJob j; j.segs = sub.data(); j.nseg = (uint32_t)sub.size(); j.k = k; j.mont = mont;
    return msm_batch(cc, mine, pb, j, cc->stream);
  });
  if (rc) return rc;
  std::vector<host::HXYZZ> tot(k, host::HXYZZ::infinity());
  for (size_t g = 0; g < G; ++g) { if (!busy[g]) continue; for (uint32_t q = 0; q < k; ++q) tot[q] = host::hadd(tot[q], host::hfrom_jacobian(&part[(size_t)18 * (k * g + q)])); }
  host::hstore_jacobian_normalized_batch(out_jac18, tot.data(), k);
  return ALEO_MI355X_OK;
}
}  // namespace aleo_mi355x
extern "C" {

int32_t aleo_mi355x_bases_attach_shards(uint64_t handle, uint64_t sharded_handle, size_t min_points) {
  try {
    Device* d = nullptr; { const int32_t rc = get_device(&d); if (rc) return rc; }
    size_t n_sh = 0;
    if (sharded_handle) { auto S = sharded_find(sharded_handle); if (!S) return ALEO_MI355X_ERR_BAD_HANDLE; n_sh = S->n; }
    std::lock_guard<std::mutex> lk(d->mu);
    auto it = d->bases.find(handle);
    if (it == d->bases.end()) { g_last_error = "unknown bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
    if (sharded_handle && n_sh != it->second->pb.n) { g_last_error = "bases_attach_shards: the sharded set must hold the same number of points"; return ALEO_MI355X_ERR_BAD_ARG; }
    it->second->pb.shards = sharded_handle; it->second->pb.shard_min = min_points; it->second->pb.shard_ntt_min = (size_t)1 << 24;
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_bases_shard_transforms(uint64_t handle, size_t min_elements) {
  try {
    Device* d = nullptr; { const int32_t rc = get_device(&d); if (rc) return rc; }
    std::lock_guard<std::mutex> lk(d->mu);
    auto it = d->bases.find(handle);
    if (it == d->bases.end()) { g_last_error = "unknown bases handle"; return ALEO_MI355X_ERR_BAD_HANDLE; }
    it->second->pb.shard_ntt_min = min_elements ? min_elements : (size_t)1 << 24;
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_kzg_commit_segments_sharded_device(void* out104, size_t n_out, uint64_t sharded_handle, const aleo_mi355x_commit_segment* segs, size_t n_segs, void* stream) {
  try {
    if (!n_out) return ALEO_MI355X_OK;
    if (!out104 || (!segs && n_segs) || n_out >= (1u << 20) || n_segs >= (1u << 22)) { g_last_error = "commit_segments_sharded: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    API_BEGIN
    PICK_STREAM(s)
    std::vector<MsmSeg> sg(n_segs);
    for (size_t q = 0; q < n_segs; ++q) {
      if (!segs[q].scalars && segs[q].len) { g_last_error = "commit_segments_sharded: null segment"; return ALEO_MI355X_ERR_BAD_ARG; }
      sg[q].d_ptr = segs[q].scalars; sg[q].len = segs[q].len; sg[q].off = segs[q].base_offset; sg[q].out = segs[q].output;
    }
    std::vector<uint64_t> jac(18 * n_out);
    const int32_t rc = commit_sharded(c, sharded_handle, sg.data(), (uint32_t)n_segs, (uint32_t)n_out, true, jac.data(), s, true);
    if (rc) return rc;
    jac_to_affine_rows(out104, jac.data(), n_out);
    return ALEO_MI355X_OK;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

int32_t aleo_mi355x_kzg_commit_batch_sharded_device(void* out104, uint64_t sharded_handle, const void* const* d_coeffs, const size_t* lens, size_t k, void* stream) {
  try {
    int32_t rc = batch_args_ok(out104, d_coeffs, lens, k); if (rc || !k) return rc;
    std::vector<aleo_mi355x_commit_segment> sg(k);
    for (size_t q = 0; q < k; ++q) { sg[q].scalars = d_coeffs[q]; sg[q].len = lens[q]; sg[q].base_offset = 0; sg[q].output = (uint32_t)q; }
    return aleo_mi355x_kzg_commit_segments_sharded_device(out104, k, sharded_handle, sg.data(), k, stream);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}


// ---- one transform over several devices of this process: the 4-step schedule of aleo_amd/dist.py ShardedDomain behind the C ABI --------------------
// n = R * C (R = 2^floor(lg n / 2)), G devices, natural order in and out of ONE host buffer:
//   device g uploads the coefficient COLUMNS c in [g C / G, (g + 1) C / G) of the R x C matrix x[r C + c] (a strided copy: its 1/G of the PCIe traffic),
//   transposes them, runs its C / G column transforms of length R, multiplies by w_n^(c k_r) and cuts the result into G blocks by k_r range;
//   block h goes to device h (one peer copy per pair: the all-to-all of SURVEY.md 8(e), G - 1 peers per device, one per xGMI link);
//   device h transposes what it received into rows k_r, runs its R / G row transforms of length C, transposes once more and stores X[k_c R + k_r]
//   straight into the host buffer (strided copy).  Coset shift and n^-1 as in the single-device transform (fr_grid_scale mode 1 / the batched inverse).
// Threads: one per shard and phase (the runtime's current device is per thread); phases are separated by joins, so no peer copy starts before
// every column transform has finished.  A device may be listed more than once (the tests: one card).
}  // extern "C" (helpers of the sharded transform follow)
namespace {
struct NttShard { int dev = 0; hipStream_t st = nullptr; void *a = nullptr, *b = nullptr; ShardWs* w = nullptr; };      // two buffers of n / G elements each, ping-pong (owned by the device's ShardWs)
std::mutex g_ntt_sh_mu;                                     // one sharded transform at a time: it occupies every listed device anyway
int32_t peer_copy(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t s) {
  if (dst_dev == src_dev) { HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s)); }
  else { HIPCHK(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, s)); }
  return ALEO_MI355X_OK;
}
// the k-th shard workspace of the calling thread's current device (stream created once, buffers grow-only)
int32_t shard_ws(size_t ordinal, size_t bytes, NttShard* out) {
  Device* d = nullptr; { const int32_t rc = get_device(&d); if (rc) return rc; }
  ShardWs* w = nullptr;
  {
    std::lock_guard<std::mutex> lk(d->mu);
    while (d->shard_ws.size() <= ordinal) d->shard_ws.emplace_back(new ShardWs());
    w = d->shard_ws[ordinal].get();
  }
  if (!w->st) HIPCHK(hipStreamCreateWithFlags(&w->st, hipStreamNonBlocking));
  { int32_t rc; if ((rc = w->a.reserve(bytes)) || (rc = w->b.reserve(bytes))) return rc; }
  out->dev = d->device; out->st = w->st; out->a = w->a.p; out->b = w->b.p; out->w = w;
  return ALEO_MI355X_OK;
}
// The strided moves between ONE pageable host buffer and a shard's device buffer through two pinned bounce buffers of the shard (A/B switch ALEO_MI355X_SHARD_BOUNCE=1;
// default: hipMemcpy2DAsync on the pageable buffer, which the runtime stages itself).  The shard's own thread gathers / scatters the rows with memcpy while the previous
// chunk is on the link, so G shards on G devices move their 1/G of the buffer in parallel without sharing the runtime's staging path.  rows x row_bytes, host pitch in bytes.
static constexpr size_t BOUNCE_BYTES = (size_t)8 << 20;
bool bounce_on() { static const bool v = [] { const char* e = std::getenv("ALEO_MI355X_SHARD_BOUNCE"); return e && e[0] == '1'; }(); return v; }
int32_t bounce_reserve(ShardWs* w) {
  if (w->pin_cap) return ALEO_MI355X_OK;
  for (int b = 0; b < 2; ++b) { HIPCHK(hipHostMalloc(&w->pin[b], BOUNCE_BYTES, hipHostMallocDefault)); HIPCHK(hipEventCreateWithFlags(&w->pin_ev[b], hipEventDisableTiming)); }
  w->pin_cap = BOUNCE_BYTES; return ALEO_MI355X_OK;
}
int32_t bounce_upload(ShardWs* w, char* d_dst, const char* h_src, size_t h_pitch, size_t row_bytes, size_t rows, hipStream_t st) {
  int32_t rc = bounce_reserve(w); if (rc) return rc;
  if (row_bytes > BOUNCE_BYTES) { HIPCHK(hipMemcpy2DAsync(d_dst, row_bytes, h_src, h_pitch, row_bytes, rows, hipMemcpyHostToDevice, st)); return ALEO_MI355X_OK; }
  const size_t per = BOUNCE_BYTES / row_bytes;
  for (size_t r0 = 0, i = 0; r0 < rows; r0 += per, ++i) {
    const int b = (int)(i & 1); const size_t nr = rows - r0 < per ? rows - r0 : per;
    if (i >= 2) HIPCHK(hipEventSynchronize(w->pin_ev[b]));
    for (size_t r = 0; r < nr; ++r) std::memcpy((char*)w->pin[b] + r * row_bytes, h_src + (r0 + r) * h_pitch, row_bytes);
    HIPCHK(hipMemcpyAsync(d_dst + r0 * row_bytes, w->pin[b], nr * row_bytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(w->pin_ev[b], st));
  }
  return ALEO_MI355X_OK;
}
int32_t bounce_download(ShardWs* w, char* h_dst, size_t h_pitch, const char* d_src, size_t row_bytes, size_t rows, hipStream_t st) {
  int32_t rc = bounce_reserve(w); if (rc) return rc;
  if (row_bytes > BOUNCE_BYTES) { HIPCHK(hipMemcpy2DAsync(h_dst, h_pitch, d_src, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost, st)); return ALEO_MI355X_OK; }
  const size_t per = BOUNCE_BYTES / row_bytes; size_t prev_r0 = 0, prev_nr = 0; int prev_b = -1;
  auto scatter = [&](int b, size_t r0, size_t nr) { for (size_t r = 0; r < nr; ++r) std::memcpy(h_dst + (r0 + r) * h_pitch, (const char*)w->pin[b] + r * row_bytes, row_bytes); };
  for (size_t r0 = 0, i = 0; r0 < rows; r0 += per, ++i) {
    const int b = (int)(i & 1); const size_t nr = rows - r0 < per ? rows - r0 : per;
    HIPCHK(hipMemcpyAsync(w->pin[b], d_src + r0 * row_bytes, nr * row_bytes, hipMemcpyDeviceToHost, st));      // (buffer b was scattered out two chunks ago)
    HIPCHK(hipEventRecord(w->pin_ev[b], st));
    if (prev_b >= 0) { HIPCHK(hipEventSynchronize(w->pin_ev[prev_b])); scatter(prev_b, prev_r0, prev_nr); }
    prev_b = b; prev_r0 = r0; prev_nr = nr;
  }
  if (prev_b >= 0) { HIPCHK(hipEventSynchronize(w->pin_ev[prev_b])); scatter(prev_b, prev_r0, prev_nr); }
  return ALEO_MI355X_OK;
}
}  // namespace
extern "C" {

int32_t aleo_mi355x_ntt_fr_sharded(void* inout, uint32_t lg_n, int32_t direction, int32_t type, const int32_t* devices, size_t n_devices) {
  try {
    if (!inout || lg_n < 2 || lg_n > 30 || direction < 0 || direction > 1 || type < 0 || type > 1) { g_last_error = "ntt_fr_sharded: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
    if (n_devices < 1 || n_devices > 64 || (n_devices & (n_devices - 1))) { g_last_error = "ntt_fr_sharded: the number of shards must be a power of two (1..64)"; return ALEO_MI355X_ERR_BAD_ARG; }
    uint32_t lg_g = 0; while ((1u << lg_g) < n_devices) ++lg_g;
    const uint32_t lg_r = lg_n / 2, lg_c = lg_n - lg_r;
    if (lg_r < lg_g) { g_last_error = "ntt_fr_sharded: domain too small for this many shards"; return ALEO_MI355X_ERR_BAD_ARG; }
    ShardedSet S; int32_t rc = sharded_layout(S, (size_t)1 << lg_n, devices, n_devices); if (rc) return rc;      // (only its device list is used)
    const size_t G = n_devices, R = (size_t)1 << lg_r, C = (size_t)1 << lg_c, Rg = R / G, Cg = C / G, per = R * Cg;   // per = elements per shard (= Rg * C)
    std::vector<size_t> ordinal(G, 0);                       // shard g is the ordinal[g]-th shard on its device
    for (size_t g = 0; g < G; ++g) for (size_t e = 0; e < g; ++e) ordinal[g] += S.devices[e] == S.devices[g];
    std::lock_guard<std::mutex> one(g_ntt_sh_mu);
    {                                                        // every listed device initialised (and selectable) before any shard thread starts: a thread must not drop out before the barriers
      int cur = 0; if (hipGetDevice(&cur) != hipSuccess) cur = 0;
      for (size_t g = 0; g < G; ++g) { Device* dd = nullptr; if ((rc = init_device(S.devices[g], &dd))) { (void)hipSetDevice(cur); return rc; } }
      (void)hipSetDevice(cur);
    }
    std::vector<NttShard> sh(G);
    char* host = (char*)inout;
    // One worker thread per shard for the whole call (the runtime's current device is per thread; the workers are the persistent ones of for_each_shard);
    // the three phases are separated by barriers, so no peer copy starts before every column transform has finished and no buffer is overwritten
    // before its reader is done.  A shard that fails keeps meeting the barriers (the others must not hang) and the first failure is returned; a shard
    // that cannot even start, or leaves by an exception, is dropped from the barriers (Barrier::drop).
    Barrier bar(G); std::atomic<int> failed{0};
    rc = for_each_shard(S, [&](size_t g) -> int32_t {
      NttShard& d = sh[g]; int32_t r = ALEO_MI355X_OK; std::string err;
      auto phase = [&](const std::function<int32_t()>& f) { if (!r && !failed.load()) { r = f(); if (r) { err = g_last_error; failed.store(1); } } bar.wait(); };
      // phase 1: columns in, column transforms, twiddle, blocks by destination
      phase([&]() -> int32_t {
        int32_t q = shard_ws(ordinal[g], per * 32, &d); if (q) return q;
        if (bounce_on()) { if ((q = bounce_upload(d.w, (char*)d.a, host + g * Cg * 32, C * 32, Cg * 32, R, d.st))) return q; }
        else HIPCHK(hipMemcpy2DAsync(d.a, Cg * 32, host + g * Cg * 32, C * 32, Cg * 32, R, hipMemcpyHostToDevice, d.st));          // a = [R][Cg]
        if (type == ALEO_NTT_COSET && direction == ALEO_NTT_FORWARD && (q = aleo_mi355x_fr_grid_scale_device(d.a, lg_n, R, Cg, 0, g * Cg, C, 1, 0, d.st))) return q;
        if ((q = aleo_mi355x_fr_transpose_device(d.b, d.a, R, Cg, d.st))) return q;                                                   // b = [Cg][R]
        if ((q = aleo_mi355x_ntt_fr_batch_device(d.b, lg_r, Cg, ALEO_NTT_ORDER_NN, direction, ALEO_NTT_STANDARD, d.st))) return q;     // [c][k_r] (inverse: x R^-1)
        if ((q = aleo_mi355x_fr_grid_scale_device(d.b, lg_n, Cg, R, g * Cg, 0, 0, 0, direction, d.st))) return q;                     // *= w_n^(+-c k_r)
        for (size_t h = 0; h < G; ++h)                                                                                                // a = [h][Cg][Rg]: block h = my columns, device h's k_r range
          HIPCHK(hipMemcpy2DAsync((char*)d.a + h * Cg * Rg * 32, Rg * 32, (char*)d.b + h * Rg * 32, R * 32, Rg * 32, Cg, hipMemcpyDeviceToDevice, d.st));
        HIPCHK(hipStreamSynchronize(d.st));
        return ALEO_MI355X_OK;
      });
      // phase 2: the exchange — this device pulls its block of every device e into b = [e][Cg][Rg] = [C][Rg]; own block first, then the peers starting
      // with the next device, so that at any moment every link carries one copy
      phase([&]() -> int32_t {
        for (size_t k = 0; k < G; ++k) { const size_t e = (g + k) % G; const int32_t q = peer_copy((char*)d.b + e * Cg * Rg * 32, d.dev, (char*)sh[e].a + g * Cg * Rg * 32, sh[e].dev, Cg * Rg * 32, d.st); if (q) return q; }
        HIPCHK(hipStreamSynchronize(d.st));
        return ALEO_MI355X_OK;
      });
      // phase 3: row transforms, natural order out
      phase([&]() -> int32_t {
        int32_t q;
        if ((q = aleo_mi355x_fr_transpose_device(d.a, d.b, C, Rg, d.st))) return q;                                                   // a = [Rg][C]
        if ((q = aleo_mi355x_ntt_fr_batch_device(d.a, lg_c, Rg, ALEO_NTT_ORDER_NN, direction, ALEO_NTT_STANDARD, d.st))) return q;     // [k_r][k_c] (inverse: x C^-1)
        if ((q = aleo_mi355x_fr_transpose_device(d.b, d.a, Rg, C, d.st))) return q;                                                   // b = [k_c][k_r local]: X[k_c R + k_r]
        if (type == ALEO_NTT_COSET && direction == ALEO_NTT_INVERSE && (q = aleo_mi355x_fr_grid_scale_device(d.b, lg_n, C, Rg, 0, g * Rg, R, 1, 1, d.st))) return q;
        if (bounce_on()) { if ((q = bounce_download(d.w, host + g * Rg * 32, R * 32, (const char*)d.b, Rg * 32, C, d.st))) return q; }
        else HIPCHK(hipMemcpy2DAsync(host + g * Rg * 32, R * 32, d.b, Rg * 32, Rg * 32, C, hipMemcpyDeviceToHost, d.st));
        HIPCHK(hipStreamSynchronize(d.st));
        return ALEO_MI355X_OK;
      });
      if (d.st) (void)hipStreamSynchronize(d.st);              // whatever happened, nothing of this call is left on the shard's stream
      if (r) g_last_error = err;
      return r;
    }, [&](size_t) { failed.store(1); bar.drop(); });          // a shard whose body never ran, or left by an exception, stops counting at the barriers: the others finish (and fail) instead of hanging
    return rc;
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

}  // extern "C"
namespace aleo_mi355x {
// ---- the same 4-step transform on data that is RESIDENT on the calling thread's device ("home"): row e2, a proof whose transforms span devices -----------
// n = R * C elements in natural order at d_inout on home.  Home (the caller's context and stream) transposes x[R][C] into T[C][R] in a scratch of its own
// (coset: g^j applied first, in place), so that shard g's coefficient columns are ONE contiguous slab T[g Cg .. (g + 1) Cg][R]:
//   phase 1  device g pulls its slab (hipMemcpyPeerAsync; same device: a plain copy), runs its Cg column transforms of length R, multiplies by w_n^(c k_r)
//            and cuts the result into G blocks by k_r range
//   phase 2  the exchange: device h pulls block h of every device (one peer copy per ordered pair — the all-to-all of SURVEY.md 8(e) over xGMI)
//   phase 3  device h transposes to rows k_r, runs its Rg row transforms of length C and pushes the [Rg][C] block back into home's scratch at row h Rg
// and home transposes the scratch [R][C] (k_r major) into d_inout [C][R] = X[k_c R + k_r], natural order (coset inverse: g^-o n^-1 fix-up in place).
// No host buffer anywhere.  Shard work runs on the persistent shard workers with contexts taken by acquire_other (never the caller's `c`, never blocking on one
// context: the caller may be a prover that holds `c` for the whole proof).  Blocking: the result is complete when the call returns.
int32_t ntt_sharded_device(Ctx* c, void* d_inout, uint32_t lg_n, int32_t direction, int32_t type, const int* devices, size_t n_devices, hipStream_t s) {
  if (!d_inout || lg_n < 2 || lg_n > 30 || direction < 0 || direction > 1 || type < 0 || type > 1) { g_last_error = "ntt_fr_sharded_device: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
  if (n_devices < 1 || n_devices > 64 || (n_devices & (n_devices - 1))) { g_last_error = "ntt_fr_sharded_device: the number of shards must be a power of two (1..64)"; return ALEO_MI355X_ERR_BAD_ARG; }
  uint32_t lg_g = 0; while ((1u << lg_g) < n_devices) ++lg_g;
  const uint32_t lg_r = lg_n / 2, lg_c = lg_n - lg_r;
  if (lg_r < lg_g) { g_last_error = "ntt_fr_sharded_device: domain too small for this many shards"; return ALEO_MI355X_ERR_BAD_ARG; }
  std::vector<int32_t> devs(devices, devices + n_devices);
  ShardedSet S; int32_t rc = sharded_layout(S, (size_t)1 << lg_n, devs.data(), n_devices); if (rc) return rc;
  const size_t G = n_devices, n = (size_t)1 << lg_n, R = (size_t)1 << lg_r, C = (size_t)1 << lg_c, Rg = R / G, Cg = C / G, per = R * Cg;
  const int home = c->device;
  std::lock_guard<std::mutex> one(g_ntt_sh_mu);
  {
    int cur = home;
    for (size_t g = 0; g < G; ++g) { Device* dd = nullptr; if ((rc = init_device(S.devices[g], &dd))) { (void)hipSetDevice(cur); return rc; } }
    (void)hipSetDevice(cur);
  }
  enable_peer_access();
  if ((rc = c->dev->shard_home.reserve(n * 32))) return rc;
  char* T = c->dev->shard_home.as<char>(); char* x = (char*)d_inout;
  if (type == ALEO_NTT_COSET && direction == ALEO_NTT_FORWARD && (rc = fr_grid_scale(c, x, lg_n, R, C, 0, 0, C, 1, 0, s))) return rc;      // x[j] *= g^j
  if ((rc = fr_transpose(c, T, x, R, C, s))) return rc;                                                                                      // T[c][r]
  HIPCHK(hipStreamSynchronize(s));
  std::vector<NttShard> sh(G); std::vector<size_t> ordinal(G, 0);
  for (size_t g = 0; g < G; ++g) for (size_t e = 0; e < g; ++e) ordinal[g] += S.devices[e] == S.devices[g];
  Barrier bar(G); std::atomic<int> failed{0}; std::mutex lend_mu;
  rc = for_each_shard(S, [&](size_t g) -> int32_t {
    NttShard& d = sh[g]; int32_t r = ALEO_MI355X_OK; std::string err;
    auto phase = [&](const std::function<int32_t()>& f) { if (!r && !failed.load()) { r = f(); if (r) { err = g_last_error; failed.store(1); } } bar.wait(); };
    // A context of this shard's device for the kernels of ONE phase (their scratch), given back before the barrier.  Never the caller's `c` — unless nothing else on
    // the home device is free: every other context there may be held by workers of the very call this transform belongs to (a lockstep proof), so a shard then
    // takes turns on `c`, which is idle while its owner waits here (the same rule as commit_sharded).  Every phase ends with its stream drained.
    auto with_ctx = [&](const std::function<int32_t(Ctx*)>& f) -> int32_t {
      Device* dv = nullptr; Ctx* cc = nullptr; std::unique_lock<std::mutex> lk, lend;
      int32_t q = get_device(&dv); if (q) return q;
      if ((q = acquire_other(dv, c, &cc, lk, dv->device != home))) return q;
      if (!cc) { lend = std::unique_lock<std::mutex>(lend_mu); cc = c; }
      return f(cc);
    };
    phase([&]() -> int32_t { return with_ctx([&](Ctx* cc) -> int32_t {
      int32_t q;
      if ((q = shard_ws(ordinal[g], per * 32, &d))) return q;
      if ((q = peer_copy(d.b, d.dev, T + g * Cg * R * 32, home, per * 32, d.st))) return q;                                                  // b = [Cg][R]: my columns
      if ((q = ntt_run(cc, d.b, lg_r, Cg, ALEO_NTT_ORDER_NN, direction, ALEO_NTT_STANDARD, d.st))) return q;                                 // [c][k_r] (inverse: x R^-1)
      if ((q = fr_grid_scale(cc, d.b, lg_n, Cg, R, g * Cg, 0, 0, 0, direction, d.st))) return q;                                             // *= w_n^(+-c k_r)
      for (size_t h = 0; h < G; ++h)                                                                                                        // a = [h][Cg][Rg]
        HIPCHK(hipMemcpy2DAsync((char*)d.a + h * Cg * Rg * 32, Rg * 32, (char*)d.b + h * Rg * 32, R * 32, Rg * 32, Cg, hipMemcpyDeviceToDevice, d.st));
      HIPCHK(hipStreamSynchronize(d.st));
      return ALEO_MI355X_OK;
    }); });
    phase([&]() -> int32_t {
      for (size_t k = 0; k < G; ++k) { const size_t e = (g + k) % G; const int32_t q = peer_copy((char*)d.b + e * Cg * Rg * 32, d.dev, (char*)sh[e].a + g * Cg * Rg * 32, sh[e].dev, Cg * Rg * 32, d.st); if (q) return q; }
      HIPCHK(hipStreamSynchronize(d.st));
      return ALEO_MI355X_OK;
    });
    phase([&]() -> int32_t { return with_ctx([&](Ctx* cc) -> int32_t {
      int32_t q;
      if ((q = fr_transpose(cc, d.a, d.b, C, Rg, d.st))) return q;                                                                           // a = [Rg][C]
      if ((q = ntt_run(cc, d.a, lg_c, Rg, ALEO_NTT_ORDER_NN, direction, ALEO_NTT_STANDARD, d.st))) return q;                                 // [k_r][k_c] (inverse: x C^-1)
      if ((q = peer_copy(T + g * Rg * C * 32, home, d.a, d.dev, per * 32, d.st))) return q;                                                  // home scratch [R][C], k_r major (T is dead: every slab was pulled before the first barrier)
      HIPCHK(hipStreamSynchronize(d.st));
      return ALEO_MI355X_OK;
    }); });
    if (d.st) (void)hipStreamSynchronize(d.st);
    if (r) g_last_error = err;
    return r;
  }, [&](size_t) { failed.store(1); bar.drop(); });
  if (rc) return rc;
  if (hipSetDevice(home) != hipSuccess) { g_last_error = "hipSetDevice failed"; return ALEO_MI355X_ERR_HIP; }
  if ((rc = fr_transpose(c, x, T, R, C, s))) return rc;                                                                                      // x[k_c][k_r] = X[k_c R + k_r]
  if (type == ALEO_NTT_COSET && direction == ALEO_NTT_INVERSE && (rc = fr_grid_scale(c, x, lg_n, C, R, 0, 0, R, 1, 1, s))) return rc;        // g^-o (the n^-1 of the inverse came with the two batched transforms)
  HIPCHK(hipStreamSynchronize(s));
  return ALEO_MI355X_OK;
}
// the devices of a sharded base set (the prover routes its large transforms over the devices its committer key is spread over)
int32_t sharded_devices(uint64_t sharded_handle, std::vector<int>* out) {
  auto S = sharded_find(sharded_handle); if (!S) return ALEO_MI355X_ERR_BAD_HANDLE;
  *out = S->devices; return ALEO_MI355X_OK;
}
}  // namespace aleo_mi355x
extern "C" {

int32_t aleo_mi355x_ntt_fr_sharded_device(void* d_inout, uint32_t lg_n, int32_t direction, int32_t type, const int32_t* devices, size_t n_devices, void* stream) {
  try {
    if (!devices && n_devices > 1) {                         // NULL: the first n_devices visible devices, cyclically
      int count = 0; if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { g_last_error = "no HIP device visible"; return ALEO_MI355X_ERR_NO_DEVICE; }
      std::vector<int32_t> dv(n_devices); for (size_t g = 0; g < n_devices; ++g) dv[g] = (int32_t)(g % (size_t)count);
      return aleo_mi355x_ntt_fr_sharded_device(d_inout, lg_n, direction, type, dv.data(), n_devices, stream);
    }
    int32_t one = 0; if (!devices) { if (hipGetDevice(&one) != hipSuccess) one = 0; devices = &one; }
    API_BEGIN
    PICK_STREAM(s)
    return ntt_sharded_device(c, d_inout, lg_n, direction, type, (const int*)devices, n_devices, s);
  } catch (...) { return ALEO_MI355X_ERR_HIP; }
}

}  // extern "C"

// ctx.h — per-device state of libaleo_mi355x.so.  A Device holds what calls share (pinned base sets, SRS cache, NTT
// tables) behind one short-held mutex; each call runs on one of its Ctx slots (own stream, events, grow-only HBM
// workspaces), so the rayon threads of one prover round overlap their MSMs/NTTs on the GPU instead of queueing.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <functional>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/aleo_mi355x.h"

namespace aleo_mi355x {

extern thread_local std::string g_last_error;

#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e__ = (expr);                                                                           \
    if (e__ != hipSuccess) {                                                                           \
      char buf__[512];                                                                                 \
      snprintf(buf__, sizeof buf__, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      g_last_error = buf__;                                                                            \
      return (e__ == hipErrorOutOfMemory) ? ALEO_MI355X_ERR_OOM : ALEO_MI355X_ERR_HIP;                 \
    }                                                                                                  \
  } while (0)

// A grow-only device buffer (hipMalloc is slow and synchronising; the prove path calls MSM/NTT repeatedly
// with the same few sizes, so buffers are kept at their high-water mark — there are 288 GB to spend).
struct DevBuf {
  void* p = nullptr; size_t cap = 0;
  int32_t reserve(size_t bytes) {
    if (bytes <= cap) return ALEO_MI355X_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8;
    HIPCHK(hipMalloc(&p, want));
    cap = want; return ALEO_MI355X_OK;
  }
  template <class T> T* as() const { return (T*)p; }
};

// A temporary device allocation that is freed on every return path (setup code: base generation, table builds).
struct DevTmp {
  void* p = nullptr;
  DevTmp() = default; DevTmp(const DevTmp&) = delete; DevTmp& operator=(const DevTmp&) = delete;
  ~DevTmp() { if (p) (void)hipFree(p); }
  int32_t alloc(size_t bytes) { HIPCHK(hipMalloc(&p, bytes ? bytes : 1)); return ALEO_MI355X_OK; }
  void* release() { void* q = p; p = nullptr; return q; }
};

// Stride of a point row in the 28-bit-limb form (x'[14] | y'[14] = 112 payload bytes).  112 packs the rows; 128 gives every row its
// own 128-byte line (one memory request per gathered point instead of 1.75 on average) for 14 % more HBM per table.
#ifndef ALEO_ROW28
#define ALEO_ROW28 128
#endif
static constexpr size_t ROW28 = ALEO_ROW28;
static_assert(ROW28 == 112 || ROW28 == 128, "row stride of the 28-bit tables");

struct PinnedBases {
  void* d_xy = nullptr;        // n x 96 bytes: x | y, Montgomery, canonical
  void* d_xy28 = nullptr;      // n x 112 bytes: the same points in the 28-bit-limb form the accumulation kernels compute in (fp28.h)
  uint8_t* d_inf = nullptr;    // n bytes, nullptr when no base is the point at infinity
  // optional fixed-base tables (msm_precompute): up to three tiers so that every call size against one SRS gets a window width
  // that suits it — the full set at c = 20 / 17, its first 2^17 points at c = 16, its first 2^15 points at c = 13.
  // Rows: W x cover x 112 bytes, row w holds 2^(c w) * P_i in the 28-bit-limb form (fp28.h); serves min_n <= n <= cover.
  struct PreTable { void* d = nullptr; int c = 0; size_t cover = 0, min_n = 0; };
  PreTable tab[3];
  // optional table over ONE sub-range [range_off, range_off + range.cover) with a narrow window (aleo_mi355x_bases_precompute_range): MSMs whose
  // scalars are mostly 0 / 1 (a witness in evaluation form against the Lagrange-basis powers) put too few points into the wide windows' buckets
  PreTable range; size_t range_off = 0;
  bool tabled = false;         // msm_precompute has run (a set below 2^10 points gets no table)
  // a sharded copy of the same points on several devices (aleo_mi355x_bases_attach_shards): commitments of >= shard_min points made against this set by the
  // prover / the segment entry points go to the shards (api.hip commit_sharded) — 0 = none
  uint64_t shards = 0; size_t shard_min = 0;
  size_t shard_ntt_min = (size_t)1 << 24;      // transforms of at least this many elements take the shards' devices too (aleo_mi355x_bases_shard_transforms; default 2^24: below it the 4-step split costs more in barriers and copies than a 0.1-2 ms single-device transform does)
  size_t n = 0;
};

// One-shot aleo_mi355x_msm_g1 calls keep their base array resident between calls ("device-resident SRS
// cache keyed by host pointer + length + hash"): KZG10::commit always multiplies against a prefix of the same powers.
// owner of a pinned set's HBM; calls hold a shared_ptr while they run, so an unpin from another thread cannot free it under them
struct PinnedOwner {
  PinnedBases pb; bool building = false;     // building: a table build for this set is in flight on some slot
  ~PinnedOwner() { if (pb.d_xy) (void)hipFree(pb.d_xy); if (pb.d_xy28) (void)hipFree(pb.d_xy28); if (pb.d_inf) (void)hipFree(pb.d_inf); for (auto& t : pb.tab) if (t.d) (void)hipFree(t.d); if (pb.range.d) (void)hipFree(pb.range.d); }
};

struct PinnedG2 {                        // aleo_mi355x_bases_g2_pin: x | y rows (192 B), infinity flags when any, the 28-bit rows the accumulation reads (224 B)
  void* d_xy = nullptr; uint8_t* d_inf = nullptr; void* d_rows28 = nullptr; size_t n = 0;
  ~PinnedG2() { if (d_xy) (void)hipFree(d_xy); if (d_inf) (void)hipFree(d_inf); if (d_rows28) (void)hipFree(d_rows28); }
};

struct SrsCacheEntry {
  const void* host_ptr = nullptr; size_t n = 0, stride = 0; uint64_t handle = 0, last_use = 0; uint32_t hits = 0;
  std::vector<std::pair<size_t, uint64_t>> samples;      // (point index, hash of its 96 bytes)
};

struct MsmTiming { double total = 0, sort = 0, accum = 0, reduce = 0, host = 0, accum_kernel = 0; int accum_launches = 1; };      // accum_kernel: mean duration of the accum_launches launches of k_accum28

struct NttTables;   // ntt.hip

struct Device;

struct Ctx {                           // one concurrency slot
  Device* dev = nullptr;
  int device = -1;
  hipStream_t stream = nullptr;
  std::mutex mu;                       // held for the duration of one API call
  hipEvent_t ev[8] = {};
  uint32_t meta_seq = 0;               // msm_sort_phase: sequence number of the slice metadata k_scan_top stores into h_pinned
  // The sort's zero-initialised block (hist | lists | meta | cursors | count matrix) cleared AHEAD: the table path queues the fill for the NEXT chain behind its last
  // reader (the bucket reduction), where the GPU would otherwise idle under the host tail; the next sort on the same stream within the cleared size skips its own fill.
  size_t hist_zero_max = 0;            // the largest zero-initialised block a chain on this context has needed (clear-ahead clears that much)
  size_t hist_clean = 0; hipStream_t hist_clean_stream = nullptr; void* hist_clean_ptr = nullptr;
  // MSM workspaces
  DevBuf hist, scan_local, scan_blk, sorted, part_cnt, part_items, partial, task_g, meta, vbuf, scalars_stage, out_stage;
  void* h_pinned = nullptr; size_t h_pinned_cap = 0;    // pinned host staging for small D2H results
  // Work the caller wants queued on the stream BEHIND the last kernel of a single-chain msm_batch request and BEFORE the host waits for the result (the wait is
  // then on an event recorded in between): the GPU runs it while the host does the tail of the MSM.  Consumed (reset) by whoever runs it; a request of several
  // chains leaves it alone and the caller runs it afterwards (varuna.hip commit()).
  std::function<int32_t()> tail_hook;
  DevBuf prover_ws; void* prover_pin = nullptr; size_t prover_pin_cap = 0;      // varuna.hip: one proof's device workspace, pinned staging of the assignments
  MsmTiming last_msm;
  DevBuf cold_raw, cold_xy, cold_xy28, cold_flags;      // the cold one-shot MSM (api.hip cold_bases): the call's copy of the caller's bases, grow-only like every other workspace of the slot
  DevBuf ntt_tmp, ntt_stage;
  // ntt_tmp is scratch of the *_device entry points, which enqueue on the CALLER's stream and return without synchronising;
  // the slot is then handed to the next call, possibly on another stream.  scratch_ev is recorded after the last kernel that
  // touches the scratch; the next user waits on it (scratch_acquire) before its first kernel, or synchronises on it before
  // the buffer is freed to grow.
  // The record is LAZY when the last user ran on one of the slot's own streams (scratch_stream): a later user on the same stream needs no event at all (stream
  // order), a user on another stream records it then — an event record between two kernels costs ~6 us of idle GPU, and a proof made ~17 of them.
  hipEvent_t scratch_ev = nullptr; bool scratch_busy = false; hipStream_t scratch_stream = nullptr;
  hipStream_t side = nullptr;          // second stream of the slot: small read-backs that must not wait for the kernels queued behind them
  hipStream_t hi = nullptr;            // a high-priority stream: the sort of a later chunk / the next launch chain must get its workgroups in while an accumulation fills the chip (msm_run_chunked, run_chains)
  hipStream_t aux = nullptr;           // a third normal-priority stream: run_chains_pipelined's sorts and reductions (ALEO_MI355X_PIPELINE_HI=1 puts them on `hi`)
  hipEvent_t ev_hop = nullptr;         // run_chains: a chain's sort (on hi) -> its accumulation (on the normal-priority stream)
};

int32_t scratch_acquire(Ctx* c, DevBuf& b, size_t bytes, hipStream_t s);
int32_t scratch_release(Ctx* c, hipStream_t s);

static constexpr int MAX_SLOTS = 8;

struct Barrier {                                           // reusable; C++17 has none
  std::mutex mu; std::condition_variable cv; size_t n, waiting = 0, phase = 0;
  explicit Barrier(size_t n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> lk(mu); const size_t ph = phase;
    if (++waiting >= n) { waiting = 0; ++phase; cv.notify_all(); } else cv.wait(lk, [&] { return phase != ph; });
  }
  // a party that leaves for good (it failed before or between the phases): the others stop waiting for it, now and at every later phase
  void drop() {
    std::lock_guard<std::mutex> lk(mu);
    if (n) --n;
    if (n && waiting >= n) { waiting = 0; ++phase; cv.notify_all(); }
  }
};

// What one shard of a transform sharded over devices (aleo_mi355x_ntt_fr_sharded) keeps on its device between calls: a stream and two grow-only
// buffers of n / G elements.  A device listed k times in a call uses its first k entries.
struct ShardWs { hipStream_t st = nullptr; DevBuf a, b; void* pin[2] = {nullptr, nullptr}; size_t pin_cap = 0; hipEvent_t pin_ev[2] = {nullptr, nullptr}; };      // pin: two pinned bounce buffers for the strided host <-> device moves of the host-buffer transform (ALEO_MI355X_SHARD_BOUNCE=1)

struct Device {
  int device = -1;
  int n_slots = 4;                     // ALEO_MI355X_SLOTS (1..8)
  std::mutex mu;                       // guards everything below; never held across a kernel launch or a copy of bulk data
  std::map<uint64_t, std::shared_ptr<PinnedOwner>> bases; uint64_t next_handle = 1;
  std::vector<SrsCacheEntry> srs_cache; uint64_t srs_clock = 0;
  std::map<uint64_t, std::shared_ptr<PinnedG2>> g2_bases;      // handles share next_handle's counter with `bases`
  std::map<uint64_t, NttTables*> ntt_tables;
  std::map<uint64_t, std::shared_ptr<struct VarunaIndexOwner>> varuna; uint64_t next_varuna = 1;      // circuit indices (varuna.hip)
  std::atomic<int> ntt_attr_mask{0};   // which NTT kernel instances had their LDS limit raised on THIS device
  Ctx slots[MAX_SLOTS];
  std::vector<std::unique_ptr<ShardWs>> shard_ws;     // grown under mu; used only by the one sharded transform in flight (api.hip g_ntt_sh_mu)
  DevBuf shard_home;                   // ntt_sharded_device: the home device's transposed copy of the data (n elements), same lock
  hipStream_t hi_pool[MAX_SLOTS] = {}; int hi_made = 0;      // the high-priority streams of the device, dealt to the contexts round-robin (api.hip first_use): at most ALEO_MI355X_HI_POOL of them exist
  Ctx helpers[MAX_SLOTS];              // extra streams + scratch a lockstep call borrows for its worker threads (never handed out as API slots)
};
// Borrows up to `want` idle helper contexts of the device (try-lock: none is waited for); they are released when `hs` goes out of scope.
struct HelperSet { std::vector<Ctx*> ctx; std::vector<std::unique_lock<std::mutex>> locks; };
int32_t acquire_helpers(Device* d, int want, HelperSet& hs);

extern thread_local MsmTiming g_last_msm;   // phase times of the calling thread's most recent MSM
// Host-side trace of one proof (ALEO_MI355X_HOSTTRACE=1): labelled timestamps of the calling thread, printed to stderr by varuna_prove_batch — where the
// host spends the turn-arounds between a commitment's last kernel and the next round's first (tools/proof_timeline_full.py shows the GPU's side of the same gaps).
// f(i) for i in [0, n) on the calling thread and up to 3 parked helper threads of the library (created on first use, never destroyed); returns when all are done.
// For the host tails of many-result launch chains (a lockstep round's 8 x k results: ~15 us of Horner each while every other thread of the call waits at a barrier).
// f must not throw and must not call back into this function.
void host_parallel_for(size_t n, const std::function<void(size_t)>& f);
void host_trace_mark(const char* label);
#define HT(label) do { if (::aleo_mi355x::g_host_trace_on) ::aleo_mi355x::host_trace_mark(label); } while (0)
extern bool g_host_trace_on;
int32_t ensure_host_pinned(Ctx* c, size_t bytes);

// msm.hip
// One launch chain computes job.k results ("sets"); result q is the sum over the segments with out == q of
//   sum_i scalar[i] * base[off + i],  i < len    (scalars at the DEVICE pointer d_ptr; the segment array itself is host memory).
// k > 1 needs a table tier that covers every base reached and k <= msm_max_sets(); msm_batch() groups arbitrary requests accordingly.
struct MsmSeg { const void* d_ptr = nullptr; size_t len = 0, off = 0; uint32_t out = 0; };
struct MsmJob { const MsmSeg* segs = nullptr; uint32_t nseg = 0, k = 0; bool mont = false; bool sparse = false; bool fire_tail = false; size_t tier_n = 0; bool lean = false; };      // lean: no phase-timing events on the stream (the prover's commitments: MsmTiming then only carries the host tail)      // tier_n: pick the table tier as for a reach of tier_n (a part of a split request keeps the whole request's window)      // fire_tail: this launch chain is the whole request — run Ctx::tail_hook behind its last kernel      // sparse: hint — use the set's range table when every segment lies inside it
int32_t msm_run(Ctx* c, uint64_t* out_jac18, const PinnedBases& pb, const MsmJob& job, hipStream_t s);
inline int32_t msm_run1(Ctx* c, uint64_t* out_jac18, const PinnedBases& pb, const void* d_scalars, size_t n, bool mont, hipStream_t s, bool sparse = false) {
  MsmSeg g; g.d_ptr = d_scalars; g.len = n;
  MsmJob j; j.segs = &g; j.nseg = 1; j.k = 1; j.mont = mont; j.sparse = sparse; return msm_run(c, out_jac18, pb, j, s);
}
int32_t msm_batch(Ctx* c, uint64_t* out_jac18, const PinnedBases& pb, const MsmJob& job, hipStream_t s);
// api.hip: the same request against a SHARDED copy of the base set (handle of aleo_mi355x_bases_pin_sharded): every segment is cut at the shard boundaries,
// device g pulls its pieces of the scalar vectors from the calling thread's device (peer copies; same device: none) and runs msm_batch against its shard,
// the G x k partial results are added on the host in shard order.  `s` (the stream the scalars were produced on) is synchronised first when s_drain; the results are
// normalised exactly like msm_batch's, so the bytes equal the single-device call's.  `c` is the caller's slot: shard work never waits for it.
int32_t commit_sharded(Ctx* c, uint64_t sharded_handle, const MsmSeg* segs, uint32_t nseg, uint32_t k, bool mont, uint64_t* out_jac18, hipStream_t s, bool s_drain);
// api.hip: one transform of 2^lg_n elements resident at d_inout on the caller's device, computed over the listed devices (4-step, peer copies, no host buffer); blocking
int32_t ntt_sharded_device(Ctx* c, void* d_inout, uint32_t lg_n, int32_t direction, int32_t type, const int* devices, size_t n_devices, hipStream_t s);
int32_t sharded_devices(uint64_t sharded_handle, std::vector<int>* out);      // the device list of a sharded base set
// One result over n points, scalars on the device (host_src == nullptr) or still on the host (then d_scalars is ignored and the scalars are uploaded into the
// contexts' staging buffers): host scalars from 2^19 points on go in two halves on two contexts that share one bucket reduction — msm.hip msm_run1_split
int32_t msm_run1_split(Ctx* c, uint64_t* out_jac18, const PinnedBases& pb, const void* d_scalars, size_t n, bool mont, hipStream_t s, bool sparse, const void* host_src, bool may_merge = true);
uint32_t msm_max_sets(const PinnedBases& pb, size_t n);
int32_t launch_fq_mul(Ctx* c, void* r, const void* a, const void* b, size_t n);
int32_t launch_fr_mul(Ctx* c, void* r, const void* a, const void* b, size_t n);
int32_t generate_multiples(Ctx* c, const void* base104, uint64_t first, size_t n, PinnedBases* out);
int32_t generate_from_scalars(Ctx* c, const void* base104, const void* scalars32, size_t n, PinnedBases* out);
int32_t msm_precompute(Ctx* c, PinnedBases* pb);
int32_t msm_precompute_range(Ctx* c, PinnedBases* pb, size_t off, size_t n, int window_bits);
int32_t unpack_affine104(Ctx* c, const void* d_rows104, void* d_xy96, void* d_flags, size_t n, hipStream_t s);      // d_flags: n bytes + a uint32 count at the next multiple of 4
int32_t rows_to28_into(const void* d_xy96, void* d_dst, size_t n, hipStream_t s);
int32_t make_rows28(Ctx* c, PinnedBases* pb);          // fills pb->d_xy28 from pb->d_xy
int32_t selftest_madd28(Ctx* c, uint32_t lanes, uint32_t steps, uint64_t seed, uint32_t* failures);
int32_t selftest_addquad(Ctx* c, uint32_t ops, uint64_t seed, uint32_t* failures);
// g2.hip
int32_t msm_g2_run(Ctx* c, uint64_t* out_jac36, const void* d_xy, const uint8_t* d_inf, const void* d_scalars, size_t n, hipStream_t s, const void* d_rows28 = nullptr);      // d_rows28: the set's resident 28-bit rows (a pinned G2 set), else built per call
int32_t g2_rows_to28(const void* d_xy192, void* d_dst224, size_t n, hipStream_t s);
int32_t g2_sum_host(uint64_t* out36, const uint64_t* pts36, size_t count);
int32_t selftest_g2pair(Ctx* c, const void* aff192_host, uint32_t n, uint32_t npairs, uint32_t* failures2);      // [0] pairs that disagreed, [1] OR of the failing steps
int32_t g2_unpack200(Ctx* c, const void* d_rows200, void* d_xy192, void* d_flags, uint32_t* d_count, size_t n, hipStream_t s);      // 200-byte G2Affine rows -> 192-byte rows + flag bytes + their count
// frops.hip
int32_t fr_lin(Ctx* c, void* d_dst, size_t n, const void* c0, const void* c1, const void* d_a, const void* c2, const void* d_b, hipStream_t s);
int32_t fr_powers(Ctx* c, void* d_dst, size_t n, const void* first, const void* ratio, hipStream_t s);
int32_t fr_gather_mul(Ctx* c, void* d_dst, size_t n, const void* d_scale, const void* d_t1, const void* d_idx1, const void* d_t2, const void* d_idx2, hipStream_t s);
int32_t fr_gather_mul3(Ctx* c, void* const* d_dst, const size_t* n, const void* const* d_scale, const void* d_t1, const void* const* d_idx1, const void* d_t2, const void* const* d_idx2, uint32_t count, hipStream_t s);      // the same for up to three ranges (scale and both tables present) in one launch
int32_t fr_eval_batch(Ctx* c, void* d_out, const void* const* d_polys, const size_t* lens, const void* z_mont, size_t k, hipStream_t s);
int32_t fr_random(Ctx* c, void* d_dst, size_t n, const uint8_t* seed32, uint64_t first, int32_t mont, hipStream_t s);
int32_t fr_lincomb(Ctx* c, void* d_dst, size_t n, const void* c0, const void* const* d_terms, const size_t* lens, const void* coeffs, size_t k, hipStream_t s);
int32_t fr_add_tiled(Ctx* c, void* d_dst, size_t n, const void* d_src, size_t n_src, hipStream_t s);
int32_t ahp_first_sumcheck(Ctx* c, void* d_dst, size_t n, const void* d_r, const void* d_a, const void* d_b, const void* d_t, const void* d_z, const void* eta_b, const void* eta_c, hipStream_t s);
int32_t ahp_matrix_sumcheck(Ctx* c, void* d_dst, size_t n, const void* const* d_index, size_t index_stride_elems, const void* const* d_f, const void* consts, hipStream_t s);
int32_t fr_blind_rows(Ctx* c, void* d_dst, const void* d_src, size_t n, size_t rows, const void* rho_mont, hipStream_t s);
int32_t ahp_sumcheck_operands(Ctx* c, void* d_dst, const void* d_wit, const void* d_xp, size_t n, size_t n_x, size_t instances, hipStream_t s);
int32_t fr_scatter_to_mont(Ctx* c, void* d_dst, const void* d_src, const void* d_pos, size_t n, void* d_flag, hipStream_t s);
int32_t fr_vec_op(Ctx* c, void* d_dst, const void* d_a, const void* d_b, size_t n, int32_t op, hipStream_t s);
int32_t fr_batch_inverse(Ctx* c, void* d_inout, size_t n, hipStream_t s);
int32_t fr_divide_by_linear(Ctx* c, void* d_q, void* d_eval, const void* d_p, size_t n, const void* z_mont32, hipStream_t s);
// key synthesis (varuna_index_build): CSR rows -> (row, column position) pairs + column counts; inclusive scan; column-major scatter of (row, value)
int32_t index_expand_rows(Ctx* c, const uint32_t* d_row_ptr, const uint32_t* d_col, const uint32_t* d_positions, size_t rows, uint32_t* d_k_row, uint32_t* d_k_col, uint32_t* d_cpos, uint32_t* d_count_plus1, hipStream_t s);
int32_t index_scan_inclusive(Ctx* c, uint32_t* d_a, size_t n, hipStream_t s);
int32_t index_transpose_rows(Ctx* c, const uint32_t* d_row_ptr, const uint32_t* d_cpos, const void* d_val, size_t rows, uint32_t row_base, uint32_t* d_cursor, uint32_t* d_tcol, void* d_tval, hipStream_t s);
int32_t fr_spmv(Ctx* c, void* d_y, const void* d_row_ptr, const void* d_col, const void* d_vals, const void* d_x, size_t rows, hipStream_t s, size_t max_row = 0);      // max_row: bound of the longest row (0 = unknown): skips the long / huge-row launches that cannot have work
int32_t fr_divide_by_linear_many(Ctx* c, void* const* d_q, void* const* d_eval, const void* const* d_p, const size_t* n, const void* const* z_mont32, size_t count, hipStream_t s);      // <= 4 divisions in three launches
int32_t fr_scale_rows(Ctx* c, void* d_dst, const void* d_src, size_t n, size_t rows, const void* consts_mont, hipStream_t s);
int32_t fr_split_quotient(Ctx* c, void* d_hq, void* d_rq, const void* d_q, const void* d_mask, size_t n, void* host_sum_devptr, hipStream_t s);
int32_t fr_sub_mul(Ctx* c, void* d_dst, const void* d_a, const void* d_b, const void* d_m, size_t n, hipStream_t s);
int32_t fr_pick(Ctx* c, void* dst_devptr, const void* const* d_src, size_t count, hipStream_t s);
// varuna.hip / api.hip
int32_t varuna_prove(Ctx* c, const PinnedBases& pb, const aleo_mi355x_varuna_index& ix, const void* const* assignments, size_t k, const uint8_t* seed32, uint8_t* out, size_t* out_len);
// one request of a lockstep call (varuna_prove_many): the circuits of ONE proof, its assignments, seed and output; status / error come back per request
struct ProveRequest { std::vector<const aleo_mi355x_varuna_index*> ixs; const void* const* assignments = nullptr; const size_t* ks = nullptr; const uint8_t* seed32 = nullptr;
                      uint8_t* out = nullptr; size_t* out_len = nullptr; int32_t status = 0; std::string error; };
int32_t varuna_prove_many(Ctx* c, const PinnedBases& pb, std::vector<ProveRequest>& requests, int workers = 0);      // workers: 0 = ALEO_MI355X_LOCKSTEP_WORKERS or 4
int32_t varuna_prove_batch(Ctx* c, const PinnedBases& pb, const aleo_mi355x_varuna_index* const* ixs, size_t m, const void* const* assignments, const size_t* ks,
                           const uint8_t* seed32, uint8_t* out, size_t* out_len);
extern thread_local double g_varuna_timing[8];
struct VarunaIndexOwner;
int32_t varuna_index_build(Ctx* c, const PinnedBases& pb, std::shared_ptr<PinnedOwner> key, uint64_t key_handle, uint64_t max_degree, uint64_t gamma_offset,
                           uint64_t lagrange_offset, const aleo_mi355x_r1cs_matrix* abc, size_t n_constraints, size_t n_public, size_t n_private, uint32_t domain_flags, VarunaIndexOwner** out);
void varuna_index_delete(VarunaIndexOwner* o);
const aleo_mi355x_varuna_index* varuna_index_view(const VarunaIndexOwner* o);
const std::vector<uint8_t>& varuna_index_vk(const VarunaIndexOwner* o);
void jacobian_rows_to_affine104(void* out104, const uint64_t* jac18, size_t k);
// ntt.hip
int32_t ntt_run(Ctx* c, void* d_inout, uint32_t lg_n, size_t batch, int32_t order, int32_t direction, int32_t type, hipStream_t s);
int32_t ntt_run_from(Ctx* c, void* d_out, const void* d_src, size_t src_stride_elems, size_t src_len, uint32_t lg_n, size_t batch, int32_t direction, int32_t type, hipStream_t s);      // out of place, input zero-padded from src_len to 2^lg_n

int32_t fr_transpose(Ctx* c, void* d_dst, const void* d_src, uint64_t rows, uint64_t cols, hipStream_t s);      // dst[c][r] = src[r][c], 32-byte elements
int32_t fr_grid_scale(Ctx* c, void* d_data, uint32_t lg_n, uint64_t rows, uint64_t cols, uint64_t row0, uint64_t col0, uint64_t ld, int32_t mode,
                      int32_t direction, hipStream_t s);

}  // namespace aleo_mi355x

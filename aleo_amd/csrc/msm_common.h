// msm_common.h — what the G1 and G2 multi-scalar multiplications share (msm.hip, g2.hip): the plan, the results of the sort /
// slice phase (which depends on the scalars only, not on the group) and the device helpers that read them.
#pragma once
#include "ctx.h"

namespace aleo_mi355x {

static constexpr uint32_t SCAN_TILE = 2048;     // elements per scan block (256 threads x 8)
static constexpr uint32_t SCALAR_BITS = 254;    // 253-bit scalars + 1 bit of signed-digit carry
static constexpr uint32_t MAX_SETS = 32;          // results ("sets": each owns 2^(c-1) buckets) of one launch chain
static constexpr uint32_t MAX_SEGS = 64;          // scalar vectors of one launch chain
static constexpr uint32_t SUPER_CAP = 4096;     // buckets with > 16 slices kept in their own list
static constexpr uint32_t MAX_SLICE = 512;      // longest slice pick_rule() can produce

struct MsmPlan { uint32_t c, W, B, M, S; };
// Balanced windows (msm.hip): the top W*c - 254 windows are c-1 bits wide.
inline int plan_win_width(int c, int w) { const int W = ((int)SCALAR_BITS + c - 1) / c, full = W - (W * c - (int)SCALAR_BITS); return w < full ? c : c - 1; }
MsmPlan make_plan(size_t n, int pre_c);
// A segment = one scalar vector: n scalars at ptr multiply the bases [off, off + n) of the pinned set and add into result `set`.
// Several segments may feed one set (KZG10::commit with hiding: the polynomial against the powers and the blinding polynomial against
// the gamma powers behind them; a degree-bounded polynomial is one segment at the shifted powers' offset).  col0: first column of the
// segment's blocks in its set's rows of the level-1 count matrix [set][bin][columns of the set's segments].  (kernel argument)
struct SegArgs { const char* ptr[MAX_SEGS]; uint32_t n[MAX_SEGS], off[MAX_SEGS], col0[MAX_SEGS]; uint8_t set[MAX_SEGS]; uint32_t nseg = 0, ncol = 0, tile = 2048; };      // tile: scalars per block in the level-1 passes (msm_sort_phase picks it)

// Slice sizing.  A bucket of <= single points is one slice (one lane); larger buckets are cut into slices of <= split.
struct SliceRule { uint32_t single, split; };
__device__ __forceinline__ SliceRule pick_rule(const uint32_t* total_pairs, uint32_t M) {
  // Two pulls.  Keep buckets whole where possible (every extra slice is a 14-product tree addition): single = 2 x the
  // mean bucket size.  But fill the chip: the launch wants >= 2^18 slices (2 waves per SIMD), and a lane needs ~11 us per
  // addition, so when there are few pairs (small n, sparse scalars) slices are cut down to pairs / 2^18 points even if
  // that splits ordinary buckets.  Everything in powers of two, 4 <= split <= single <= 512.
  const uint32_t pairs = total_pairs[0], mean = pairs / M, fill_shift = total_pairs[1];      // [1]: lg of the slice count the launch aims for (msm_sort_phase)
  // below 2^18 pairs (2^10..2^13-point MSMs) the accumulation is a chain of a few ~19 us additions per lane on a mostly idle chip:
  // slices of 4 instead of 8 halve it for one more slice-tree level (2^12: 0.50 -> 0.44 ms; from 2^14 up the extra level costs what it saves)
  const uint32_t fill_min = pairs < (1u << 18) ? 4u : 8u;
  uint32_t by_mean = 32u; while (by_mean < 2u * mean && by_mean < 512u) by_mean <<= 1;
  // the accumulation kernel holds two waves per SIMD (248 VGPRs): 2^17 lanes, and its slices start longest first, so the launch lasts
  // about max(longest slice, pairs / 2^17) additions: cut at exactly that many points (not the next power of two: 1.57 M pairs want
  // slices of 12, not 16), a little over 2^17 slices is harmless — the surplus are the shortest ones
  uint32_t fill = (pairs + (1u << fill_shift) - 1) >> fill_shift;
  fill = fill < fill_min ? fill_min : (fill > 256u ? 256u : fill);
  SliceRule r; r.single = by_mean; r.split = by_mean >> 1;          // plenty of pairs: whole buckets up to 2 x the mean size, larger ones cut at the mean
  // few pairs: every bucket above `fill` points is cut into slices of <= fill — no hysteresis: a bucket of 1.5 fill left whole was the
  // longest chain of the launch (2^15..2^18 points: -4..-16 % wall time; 2^20 unchanged)
  if (by_mean >= 2u * fill) { r.single = fill; r.split = fill; }
  return r;
}
__device__ __forceinline__ uint32_t slices_of(uint32_t cnt, SliceRule r) { return cnt <= r.single ? (cnt ? 1u : 0u) : (cnt + r.split - 1) / r.split; }
__device__ __forceinline__ uint2 scan_at(const uint2* local, const uint2* blk, uint32_t g) {
  uint2 a = local[g], b = blk[g / SCAN_TILE]; return make_uint2(a.x + b.x, a.y + b.y);
}
__device__ __forceinline__ uint32_t slice_len(uint32_t cnt, uint32_t m, uint32_t k) {
  return (uint32_t)(((uint64_t)(k + 1) * cnt) / m) - (uint32_t)(((uint64_t)k * cnt) / m);
}

// What msm_sort_phase leaves in the slot's workspaces (device pointers), enqueued on `s`, nothing synchronised:
//   hist[g] points of bucket g | scan_local/scan_blk: exclusive (points, slices) prefix per bucket (scan_at) | sorted: the point index
//   stream (bit 31 = negate), bucket runs contiguous | task_g[sid] bucket of slice sid | order[t] slice ids, longest first |
//   meta[0] slices, [1] most slices in one bucket, [2] pairs, [3] multi-slice buckets (listed in heavy[]), [5] super-heavy ones, [6] most slices of a bucket in heavy[] (<= 16)
struct SortPhase {
  MsmPlan P; uint32_t M = 0, digitsW = 0, slice_blocks = 0; size_t slices_max = 0, pairs_max = 0;
  uint32_t *hist = nullptr, *heavy = nullptr, *meta = nullptr; uint2 *scan_local = nullptr, *scan_blk = nullptr;
  uint32_t *sorted = nullptr, *task_g = nullptr, *order = nullptr; const uint32_t* total_pairs = nullptr; const uint32_t* super_list = nullptr;
  uint32_t meta_seq = 0;      // the sequence number k_scan_top stores behind the slice metadata in the slot's pinned buffer (msm_wait_meta polls for it)
  size_t zero_bytes = 0;      // size of the zero-initialised block at `hist` this chain used
};
struct SliceMeta { uint32_t NT = 0, max_m = 0, n_heavy = 0, n_super = 0, max_common = 0; bool super_overflow = false; };
// P: the plan (P.W windows / sets of P.B buckets).  pre: table path (digits address row w * row_stride + i of a table, all windows share
// a set's buckets).  Records ev[0] before and ev[1] after the sort unless `lean` (phase timing off: every event record between two kernels is ~6 us of
// idle GPU); the slice metadata comes back through the slot's pinned, device-mapped buffer (k_scan_top stores it, msm_wait_meta polls).
// segs: ptr / n / off / set filled in by the caller; col0, ncol are computed here.  pts = sum of the segment lengths.
int32_t msm_sort_phase(Ctx* c, SegArgs& segs, size_t pts, bool mont, const uint8_t* d_inf, uint32_t row_stride,
                       const MsmPlan& P, bool pre, hipStream_t s, SortPhase* out, bool lean = false);
int32_t msm_wait_meta(Ctx* c, const SortPhase& sp, hipStream_t s, SliceMeta* m);

}  // namespace aleo_mi355x

// msm.hip — BLS12-377 G1 Pippenger multi-scalar multiplication for MI355X (gfx950).
//
// Replaces snarkvm-algorithms 0.14.5  algorithms/src/msm/variable_base/{mod,standard,batched}.rs
// `VariableBase::msm(bases, scalars)` [UPSTREAM-RECALL; pin /root/reference/Cargo.lock:2200], reached from
// /root/reference/rust/src/program/execute.rs:74,177 and transfer.rs:99 through Varuna's KZG commitments.
// Same mathematical function (sum_i s_i * P_i); the schedule is GPU-first, not a translation:
//
//   digits    signed c-bit windows: |d| <= 2^(c-1), so half the buckets of the reference's unsigned windows.  Plain
//             schedule: c <= 16, W = ceil(254/c) windows with their own buckets.  Fixed-base schedule (pinned SRS with
//             msm_precompute's table of 2^(20w) * P_i): c = 20, 13 windows that all feed ONE set of 2^19 buckets.
//   sort      counting sort of the n*W (bucket, point) pairs in two LDS-partitioned levels, no global atomics.  The
//             sorted stream holds 4-byte point indices (bit 31 = negate), so a bucket is a contiguous run.
//   slices    bucket runs are cut into slices (pick_rule: whole buckets up to 2x the mean size, longer ones split at the
//             mean), counting-sorted by length so the lanes of a wave run equal trip counts; one lane accumulates one
//             slice with XYZZ mixed additions on 14 x 28-bit limbs (fp28.h), reading 112-byte affine rows straight from HBM
//             (table rows on the fixed-base path, the pinned set's 28-bit copy on the plain path).
//   tree      slices of multi-slice buckets are folded pairwise (short launches over the listed buckets only).
//   reduce    sum_b (b+1) * S_b: S-bucket running sums, then either a double-and-add of the chunk base + pairwise tree
//             (plain) or lg(N) masked pairwise sums folded through LDS blocks (fixed-base); every addition after the
//             accumulation kernel is shared by a lane pair (xyzz_add_pair / xyzz28_add_pair).
//   tail      plain: the W window sums go to the host for the 2^c Horner chain (~250 dependent doublings are ~0.1 ms on
//             a host core, ~4 ms on one GPU lane); fixed-base: a lg(N) + 4-point Horner.  Then affine normalisation.
//
// HBM layout: bases n x 96 B (x|y Montgomery, AoS so a gathered point is 1-2 cache lines), table W x n x 112 B; sorted
// stream n*W x 4 B; partial sums #slices x 192 B (XYZZ) or 224 B (28-bit XYZZ).  Algorithmic bytes per point: 32 (scalar)
// + 96 (base).
#include "ctx.h"
#include "ec.h"
#include "fp28.h"
#include "host_field.hpp"
#include "msm_common.h"
#include <chrono>
#include <cstdlib>
#include <atomic>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <string>

namespace aleo_mi355x {

// Window width of the fixed-base table, by size of the pinned set: the bucket reduction is latency-bound and its work
// grows with 2^(c-1), so small SRS (real Aleo circuits are 2^15..2^17) get narrower windows than the 2^20+ sets.
//   c = 20: 13 rows, 2^19 shared buckets   c = 17: 15 rows, 2^16 buckets   c = 16: 16 rows, 2^15 buckets
// (widths whose TOP window keeps >= 13 bits of the 253-bit scalar: c = 18 or 19 would leave it 1 or 6 bits, i.e. a
// handful of buckets holding n/2 points each)

MsmPlan make_plan(size_t n, int pre_c) {
  MsmPlan p;
  if (pre_c) {   // one shared bucket set: "W = 1 window of 2^(c-1) buckets" for everything after the sort
    // running-sum chunk: 2S dependent additions per lane pair vs. one more level of masked sums per halving; measured
    // best at 16 for 2^19 buckets (enough chunks to fill the chip) and 4 for 2^15..2^16 buckets (latency only)
    p.c = (uint32_t)pre_c; p.W = 1; p.B = 1u << (pre_c - 1); p.M = p.B; p.S = pre_c >= 20 ? 16 : 4;
    static const int s_env = [] { const char* e = std::getenv("ALEO_MI355X_CHUNK_S"); return e ? std::atoi(e) : 0; }();      // experiment knob (wide tables only)
    if (pre_c >= 20 && (s_env == 4 || s_env == 8 || s_env == 16 || s_env == 32)) p.S = (uint32_t)s_env;
    return p;
  }
  uint32_t lg = 0; while (((size_t)1 << (lg + 1)) <= n) ++lg;
  int c = (int)lg - 4; if (c < 2) c = 2; if (c > 16) c = 16;
  p.c = (uint32_t)c; p.W = (SCALAR_BITS + p.c - 1) / p.c; p.B = 1u << (p.c - 1); p.M = p.W * p.B;
  p.S = p.B >= 8 ? 8 : p.B;                  // buckets per running-sum chunk
  return p;
}

// ---- scalar access ----------------------------------------------------------------------------
template <bool MONT> __device__ __forceinline__ void load_scalar(const void* scalars, uint32_t i, uint32_t (&s)[8]) {
  const uint4* p = (const uint4*)scalars + 2 * (size_t)i;
  uint4 a = p[0], b = p[1];
  s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
  if constexpr (MONT) {   // KZG10::commit path: polynomial coefficients are Montgomery Fr -> canonical bigint
    Fr f; for (int k = 0; k < 8; ++k) f.v[k] = s[k];
    f = Fr::from_mont(f);
    for (int k = 0; k < 8; ++k) s[k] = f.v[k];
  }
}
// Balanced windows.  W = ceil(254 / C) windows cover exactly 254 bits: the top D = W*C - 254 windows are C-1 bits wide, so no
// window is short.  (With W uniform C-bit windows the last one keeps 254 - (W-1)*C bits — 14 of 20 at C = 20, 7 of 13 at C = 13 — and
// its few buckets receive n / 2^13 .. n / 2^6 points each: on the table path, where all windows share one bucket set, those
// buckets had to be cut into slices and folded by 3-10 extra tree launches.)  Window w starts at bit win_offset(C, w).
__host__ __device__ constexpr int win_count(int c) { return ((int)SCALAR_BITS + c - 1) / c; }
__host__ __device__ constexpr int win_full(int c) { return win_count(c) - (win_count(c) * c - (int)SCALAR_BITS); }      // windows of the full width c
__host__ __device__ constexpr int win_width(int c, int w) { return w < win_full(c) ? c : c - 1; }
__host__ __device__ constexpr int win_offset(int c, int w) { return w <= win_full(c) ? w * c : win_full(c) * c + (w - win_full(c)) * (c - 1); }

template <int C, int W_IDX> __device__ __forceinline__ uint32_t window_raw(const uint32_t (&s)[8]) {
  constexpr int bit = win_offset(C, W_IDX), width = win_width(C, W_IDX), limb = bit >> 5, off = bit & 31;
  uint32_t v = 0;
  if constexpr (limb < 8) {
    v = s[limb] >> off;
    if constexpr (off + width > 32 && limb + 1 < 8) v |= s[limb + 1] << (32 - off);
  }
  return v & ((1u << width) - 1u);
}

// Calls f(w, bucket_index_0based, negate) for every non-zero signed digit of the scalar.
template <int C, int W_IDX, class F> __device__ __forceinline__ void for_each_digit(const uint32_t (&s)[8], uint32_t carry, F&& f) {
  constexpr int W = win_count(C);
  if constexpr (W_IDX < W) {
    constexpr int width = win_width(C, W_IDX);
    constexpr uint32_t B = 1u << (width - 1);
    uint32_t d = window_raw<C, W_IDX>(s) + carry;
    uint32_t neg = d > B ? 1u : 0u;
    uint32_t mag = neg ? (1u << width) - d : d;
    if (mag) f((uint32_t)W_IDX, mag - 1u, neg);
    for_each_digit<C, W_IDX + 1>(s, neg, f);
  }
}

// ---- counting sort of the n*W (bucket, point) pairs: two LDS-partitioned levels, no global atomics ----
// (A first version drew one global atomic per pair: 1.5 ms at 2^20 uniform and 4.9 ms on witness-like scalars,
//  whose 0/1 values pile onto a few counters — profiles/r01_v1_kernel_stats.csv.)
// Level 1 splits by (window, high bucket bits) into <= 2048 coarse bins: every block histograms a tile of 2048
// scalars in LDS, an exclusive scan over the [bin][block] count matrix gives each block a private output run per
// bin, and the scatter pass ranks items with LDS atomics.  Level 2 gives one block per coarse bin: an LDS
// histogram over the low 8 bucket bits yields the final per-bucket counts and positions.
static constexpr uint32_t PART_TILE = 2048;       // scalars per block in the level-1 passes (SegArgs::tile: 2048, or 4096 / 8192 for chains of >= 2^21 / 2^22 points — a block's run in a
                                                  // coarse bin is tile * windows / bins items of 8 bytes: ~100 bytes at 2048, and the PMC write counter showed 3.3 x the bytes stored)
static constexpr uint32_t MAX_COARSE = 2048;      // coarse bins of ONE set: W * (B >> LB) at c = 16 (the LDS tables of the level-1 passes)
static constexpr uint32_t MAX_COARSE_ALL = 4096;  // coarse bins of all sets of a chain (k_bin_parts: one block, 16 bins per lane): 32 sets at c = 16, 16 at c = 17

// PRE = the base set carries precomputed window multiples 2^(c*w) * P_i (fixed-base MSM, see msm_precompute):
// every window then feeds ONE shared set of buckets, and the point of digit w of scalar i is table entry w*n + i.
// Batched calls (several scalar vectors against ONE pinned set, msm_run's `k`): blockIdx.y is the vector ("set"); every
// set owns its own 2^(c-1) buckets, so its coarse bins are [set * CB, (set + 1) * CB) and everything after the sort sees
// k * 2^(c-1) buckets.  Only the table path batches (PRE), where one set is one window's worth of buckets.
template <int C, bool PRE> struct SortGeom {
  static constexpr uint32_t W = (SCALAR_BITS + C - 1) / C, B = 1u << (C - 1);
  static constexpr uint32_t LB = (C - 1) < 8 ? (C - 1) : 8;       // low bucket bits, sorted in level 2
  static constexpr uint32_t CB = B >> LB, NCB = PRE ? CB : W * CB;      // coarse bins of ONE set
  static_assert(NCB <= MAX_COARSE, "coarse bin table too small");
  __device__ static uint32_t bin(uint32_t w, uint32_t b) { return PRE ? (b >> LB) : w * CB + (b >> LB); }
};

// Workgroups are dealt round-robin over the eight XCDs (b and b + 8 share one, with its L2).  The level-1 scatter writes, for every coarse bin, the runs
// of consecutive TILES next to each other — a run is ~100 bytes, so a 128-byte line holds pieces of two tiles: with tile = workgroup the two pieces
// come from different L2s and reach HBM as partial lines.  Dealing consecutive tiles to workgroups of ONE XCD lets its L2 merge them.
#ifdef ALEO_NO_XCD_TILE
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t) { return b; }
#else
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t g) { return (g & 7u) ? b : (b & 7u) * (g >> 3) + (b >> 3); }
#endif
template <int C, bool MONT, bool PRE>
__global__ void __launch_bounds__(256) k_part_count(SegArgs segs, const uint8_t* inf, uint32_t* __restrict__ cnt) {
  using Gm = SortGeom<C, PRE>;
  __shared__ uint32_t h[MAX_COARSE];
  const uint32_t bx = xcd_tile(blockIdx.x, gridDim.x);
  const uint32_t n = segs.n[blockIdx.y], base = bx * segs.tile;
  if (base >= n) return;                                      // the grid is as wide as the longest segment
  for (uint32_t i = threadIdx.x; i < Gm::NCB; i += 256) h[i] = 0;
  __syncthreads();
  const char* scalars = segs.ptr[blockIdx.y]; const uint32_t off = segs.off[blockIdx.y], nblk = segs.ncol;
  cnt += (size_t)segs.set[blockIdx.y] * Gm::NCB * nblk + segs.col0[blockIdx.y];
  for (uint32_t q = 0; q < segs.tile / 256; ++q) {
    uint32_t i = base + q * 256 + threadIdx.x;
    if (i < n && !(inf && inf[off + i])) {
      uint32_t s[8]; load_scalar<MONT>(scalars, i, s);
      for_each_digit<C, 0>(s, 0u, [&](uint32_t w, uint32_t b, uint32_t) { atomicAdd(&h[Gm::bin(w, b)], 1u); });
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < Gm::NCB; i += 256) cnt[(size_t)i * nblk + bx] = h[i];     // [bin][tile]
}

// plain exclusive scan of uint32 (tiles of SCAN_TILE + one top block); position(i) = local[i] + blk[i / SCAN_TILE]
__global__ void __launch_bounds__(256) k_scan32_tiles(const uint32_t* __restrict__ in, uint32_t len, uint32_t* __restrict__ local, uint32_t* __restrict__ tile_tot) {
  __shared__ uint32_t wsum[4];
  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
  uint32_t c[8], pre[8], run = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) c[k] = (base + k < len) ? in[base + k] : 0u;
#pragma unroll
  for (int k = 0; k < 8; ++k) { pre[k] = run; run += c[k]; }
  uint32_t inc = run; int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint32_t woff = 0; for (int k = 0; k < wv; ++k) woff += wsum[k];
  uint32_t excl = woff + inc - run;
#pragma unroll
  for (int k = 0; k < 8; ++k) if (base + k < len) local[base + k] = excl + pre[k];
  if (threadIdx.x == 255) tile_tot[blockIdx.x] = woff + inc;
}
__global__ void __launch_bounds__(256) k_scan32_top(const uint32_t* __restrict__ tile_tot, uint32_t ntiles, uint32_t* __restrict__ blk, uint32_t aux) {
  __shared__ uint32_t sh[256]; __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < ntiles; b0 += 256) {
    uint32_t i = b0 + threadIdx.x, v = i < ntiles ? tile_tot[i] : 0u;
    sh[threadIdx.x] = v; __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      uint32_t o = threadIdx.x >= (uint32_t)d ? sh[threadIdx.x - d] : 0u;
      __syncthreads(); sh[threadIdx.x] += o; __syncthreads();
    }
    uint32_t inc = sh[threadIdx.x], cr = carry;
    if (i < ntiles) blk[i] = cr + inc - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = cr + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) { blk[ntiles] = carry; blk[ntiles + 1] = aux; }       // grand total; aux rides along (pick_rule's fill target)
}
__device__ __forceinline__ uint32_t scan32_at(const uint32_t* local, const uint32_t* blk, size_t i) { return local[i] + blk[i / SCAN_TILE]; }

template <int C, bool MONT, bool PRE>
__global__ void __launch_bounds__(256) k_part_scatter(SegArgs segs, const uint8_t* inf, uint32_t row_stride,
                                                      const uint32_t* __restrict__ off_local, const uint32_t* __restrict__ off_blk, uint2* __restrict__ items) {
  using Gm = SortGeom<C, PRE>;
  __shared__ uint32_t cur[MAX_COARSE];
  const uint32_t bx = xcd_tile(blockIdx.x, gridDim.x);
  const uint32_t n = segs.n[blockIdx.y], base = bx * segs.tile;
  if (base >= n) return;
  const char* scalars = segs.ptr[blockIdx.y]; const uint32_t off = segs.off[blockIdx.y], nblk = segs.ncol;
  const size_t row0 = (size_t)segs.set[blockIdx.y] * Gm::NCB; const uint32_t col = segs.col0[blockIdx.y] + bx;
  for (uint32_t i = threadIdx.x; i < Gm::NCB; i += 256) cur[i] = scan32_at(off_local, off_blk, (row0 + i) * nblk + col);
  __syncthreads();
  for (uint32_t q = 0; q < segs.tile / 256; ++q) {
    uint32_t i = base + q * 256 + threadIdx.x;
    if (i < n && !(inf && inf[off + i])) {
      uint32_t s[8]; load_scalar<MONT>(scalars, i, s);
      for_each_digit<C, 0>(s, 0u, [&](uint32_t w, uint32_t b, uint32_t neg) {
        uint32_t pos = atomicAdd(&cur[Gm::bin(w, b)], 1u);
        items[pos] = make_uint2((PRE ? w * row_stride + off + i : off + i) | (neg << 31), b & ((1u << Gm::LB) - 1u));
      });
    }
  }
}

// Level 2: the low LB bucket bits.  A coarse bin is cut into parts of BIN_PART items, one block each, so a bin that
// skewed scalars overfill (a fifth of a witness vector is the constant 1: one bucket, one bin) is sorted by many blocks
// instead of one (it was 1.5 ms of a 4.1 ms MSM at 2^22), and every part is ranked and staged in LDS so that the index
// stream is written in runs per bucket rather than as scattered 4-byte stores (64-byte write granules: 3.5 GB for 54 M
// stores at 2^22).  k_bin_hist adds the parts' LDS histograms into hist[]; k_bin_scatter claims each part's range
// of a bucket with one global atomic per (part, bucket).
static constexpr uint32_t BIN_PART = 4096;

__global__ void __launch_bounds__(256) k_bin_parts(const uint32_t* __restrict__ off_local, const uint32_t* __restrict__ off_blk, uint32_t nblk, uint32_t ncb,
                                                   uint32_t cnt_tiles, uint32_t* __restrict__ part_start) {
  __shared__ uint32_t wsum[4];
  const uint32_t tid = threadIdx.x; const int lane = tid & 63, wv = tid >> 6;
  constexpr int PER = MAX_COARSE_ALL / 256;
  uint32_t pre[PER], run = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const uint32_t bin = tid * PER + k; pre[k] = run;
    if (bin < ncb) {
      const uint32_t st = scan32_at(off_local, off_blk, (size_t)bin * nblk);
      const uint32_t en = (bin + 1 < ncb) ? scan32_at(off_local, off_blk, (size_t)(bin + 1) * nblk) : off_blk[cnt_tiles];
      run += (en - st + BIN_PART - 1) / BIN_PART;
    }
  }
  uint32_t inc = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint32_t woff = 0; for (int k = 0; k < wv; ++k) woff += wsum[k];
  const uint32_t excl = woff + inc - run;
#pragma unroll
  for (int k = 0; k < PER; ++k) if (tid * PER + k < ncb) part_start[tid * PER + k] = excl + pre[k];
  if (tid == 255) part_start[ncb] = woff + inc;
}

struct BinPart { uint32_t bin, bstart, lo, hi; bool live; };
__device__ __forceinline__ BinPart locate_part(const uint32_t* __restrict__ off_local, const uint32_t* __restrict__ off_blk, uint32_t nblk, uint32_t ncb,
                                               uint32_t cnt_tiles, const uint32_t* __restrict__ part_start) {
  BinPart r; r.live = blockIdx.x < part_start[ncb];
  if (!r.live) return r;
  uint32_t lo = 0, hi = ncb;                           // largest bin with part_start[bin] <= block (empty bins share their successor's start)
  while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (part_start[mid] <= blockIdx.x) lo = mid; else hi = mid; }
  r.bin = lo;
  r.bstart = scan32_at(off_local, off_blk, (size_t)lo * nblk);
  const uint32_t bend = (lo + 1 < ncb) ? scan32_at(off_local, off_blk, (size_t)(lo + 1) * nblk) : off_blk[cnt_tiles];
  r.lo = r.bstart + (blockIdx.x - part_start[lo]) * BIN_PART;
  r.hi = r.lo + BIN_PART < bend ? r.lo + BIN_PART : bend;
  return r;
}

// Rank of each of the wave's keys in the block's LDS histogram.  The lanes that share the first lane's key go through one
// LDS atomic (the all-equal case of skewed scalars would otherwise serialise 4096 atomics on one address).
__device__ __forceinline__ uint32_t lds_rank(uint32_t* h, uint32_t key, bool valid, int lane) {
  const uint64_t vm = __ballot(valid);
  if (!vm) return 0u;
  const int first = __ffsll((unsigned long long)vm) - 1;
  const uint32_t k0 = __shfl(key, first);
  const bool grp = valid && key == k0;
  const uint64_t same = __ballot(grp);
  uint32_t base = 0;
  if (lane == first) base = atomicAdd(&h[k0], (uint32_t)__popcll(same));
  base = __shfl(base, first);
  if (grp) return base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
  return valid ? atomicAdd(&h[key], 1u) : 0u;
}

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wsum, int lane, int wv) {    // 256 threads; wsum: 4 words of LDS
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  __syncthreads();
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint32_t woff = 0; for (int k = 0; k < wv; ++k) woff += wsum[k];
  return woff + inc - v;
}

__global__ void __launch_bounds__(256) k_bin_hist(const uint2* __restrict__ items, const uint32_t* __restrict__ off_local, const uint32_t* __restrict__ off_blk,
                                                  uint32_t nblk, uint32_t ncb, uint32_t cnt_tiles, uint32_t LB, const uint32_t* __restrict__ part_start,
                                                  uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  const uint32_t tid = threadIdx.x; const int lane = tid & 63;
  const BinPart P = locate_part(off_local, off_blk, nblk, ncb, cnt_tiles, part_start);
  if (!P.live) return;
  h[tid] = 0;
  __syncthreads();
#pragma unroll 4
  for (uint32_t u = 0; u < BIN_PART / 256; ++u) {
    const uint32_t i = P.lo + u * 256 + tid; const bool valid = i < P.hi;
    const uint32_t key = valid ? items[i].y : 0u;
    (void)lds_rank(h, key, valid, lane);
  }
  __syncthreads();
  const uint32_t v = h[tid];
  if (v && tid < (1u << LB)) atomicAdd(&hist[((size_t)P.bin << LB) + tid], v);
}

__global__ void __launch_bounds__(256) k_bin_scatter(const uint2* __restrict__ items, const uint32_t* __restrict__ off_local, const uint32_t* __restrict__ off_blk,
                                                     uint32_t nblk, uint32_t ncb, uint32_t cnt_tiles, uint32_t LB, const uint32_t* __restrict__ part_start,
                                                     const uint32_t* __restrict__ hist, uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
  __shared__ uint32_t h[256], gb[256], wsum[4];
  __shared__ uint32_t l_idx[BIN_PART], l_dst[BIN_PART];
  const uint32_t tid = threadIdx.x; const int lane = tid & 63, wv = tid >> 6;
  const BinPart P = locate_part(off_local, off_blk, nblk, ncb, cnt_tiles, part_start);
  if (!P.live) return;
  h[tid] = 0;
  __syncthreads();
  constexpr int U = BIN_PART / 256;
  uint2 it[U]; uint32_t rank[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const uint32_t i = P.lo + u * 256 + tid; it[u] = i < P.hi ? items[i] : make_uint2(0u, 0xffffffffu); }
#pragma unroll
  for (int u = 0; u < U; ++u) rank[u] = lds_rank(h, it[u].y, it[u].y != 0xffffffffu, lane);
  __syncthreads();
  const uint32_t v = h[tid];                                                 // this part's count of bucket tid
  const uint32_t loff = block_excl_scan(v, wsum, lane, wv);                  // its offset inside the part's sorted tile
  const uint32_t g = tid < (1u << LB) ? hist[((size_t)P.bin << LB) + tid] : 0u;
  const uint32_t gexcl = block_excl_scan(g, wsum, lane, wv);                 // the bucket's offset inside the bin
  gb[tid] = P.bstart + gexcl + (v ? atomicAdd(&cursor[((size_t)P.bin << LB) + tid], v) : 0u);
  __syncthreads();
  h[tid] = loff;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < U; ++u) if (it[u].y != 0xffffffffu) {
    const uint32_t lp = h[it[u].y] + rank[u];
    l_idx[lp] = it[u].x; l_dst[lp] = gb[it[u].y] + rank[u];
  }
  __syncthreads();
  const uint32_t cnt = P.hi - P.lo;
  for (uint32_t q = tid; q < cnt; q += 256) sorted[l_dst[q]] = l_idx[q];
}

// ---- exclusive scan of (count, slices) over the M buckets --------------------------------------
// scan_local[g] = prefix inside the 2048-bucket tile; scan_blk[tile] = prefix of the tiles.  meta[0] = total
// slices, meta[1] = max slices of one bucket, meta[2] = total pairs.
// Slice sizing.  A bucket of <= T_SINGLE points is one slice (one lane); larger buckets are cut into slices of
// <= T_SPLIT.  One lane needs ~10-20 us per mixed addition, so the longest slice bounds the kernel from below: 128-point
// slices (tried) put a 2.7 ms floor under a 2.4 ms kernel, because the top window of a 253-bit scalar only has 13 bits
// and its 4779 buckets hold ~300 points each.  64/32 keeps the floor at about half the kernel time.
// Sparse inputs (witness-like scalars, small n) use 32/32 so that the accumulation still fills every SIMD.
__global__ void __launch_bounds__(256) k_scan_tiles(const uint32_t* hist, uint32_t M, const uint32_t* total_pairs, uint2* scan_local, uint2* tile_tot, uint32_t* meta,
                                                    uint32_t* __restrict__ heavy) {
  __shared__ uint2 wsum[4];
  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
  const SliceRule rule = pick_rule(total_pairs, M);
  uint32_t c[8]; uint32_t mx = 0, mxc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) c[k] = (base + k < M) ? hist[base + k] : 0u;
  uint2 pre[8]; uint2 run = make_uint2(0, 0);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    pre[k] = run; uint32_t m = slices_of(c[k], rule); run.x += c[k]; run.y += m; mx = mx > m ? mx : m;
    // multi-slice buckets are the only work of the slice tree; the few with > 16 slices (skewed scalars) get their own
    // list so that the launch width of the common list stays at 8 pairs per bucket
    if (m > 16) { uint32_t q = atomicAdd(&meta[5], 1u); if (q < SUPER_CAP) heavy[M + 2048 + q] = base + k; else heavy[atomicAdd(&meta[3], 1u)] = base + k; }
    else if (m > 1) { heavy[atomicAdd(&meta[3], 1u)] = base + k; mxc = mxc > m ? mxc : m; }
  }
  if (mxc > 1) atomicMax(&meta[6], mxc);                   // most slices of a common-list bucket: the depth of ITS tree (msm_run)
  // wave inclusive scan of the per-thread totals
  uint2 inc = run; int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t ox = __shfl_up(inc.x, d), oy = __shfl_up(inc.y, d);
    if (lane >= d) { inc.x += ox; inc.y += oy; }
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint2 woff = make_uint2(0, 0);
  for (int k = 0; k < wv; ++k) { woff.x += wsum[k].x; woff.y += wsum[k].y; }
  uint2 excl = make_uint2(woff.x + inc.x - run.x, woff.y + inc.y - run.y);
#pragma unroll
  for (int k = 0; k < 8; ++k) if (base + k < M) scan_local[base + k] = make_uint2(excl.x + pre[k].x, excl.y + pre[k].y);
  if (threadIdx.x == 255) tile_tot[blockIdx.x] = make_uint2(woff.x + inc.x, woff.y + inc.y);
  for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(mx, d); mx = mx > o ? mx : o; }
  if (lane == 0 && mx) atomicMax(&meta[1], mx);
}

// host_meta (device pointer of the slot's mapped pinned buffer): meta[0..7] go there followed by the call's sequence number at word 8, so the host reads the
// slice counts by polling — no copy on a side stream, no event on the launch stream (an event record between two kernels costs ~6 us of idle GPU on this runtime)
__global__ void __launch_bounds__(256) k_scan_top(const uint2* tile_tot, uint32_t ntiles, uint2* scan_blk, uint32_t* meta, volatile uint32_t* host_meta, uint32_t seq) {
  // one block; ntiles <= a few thousand: serial chunks of 256 with a running offset
  __shared__ uint2 sh[256]; __shared__ uint2 carry;
  if (threadIdx.x == 0) carry = make_uint2(0, 0);
  __syncthreads();
  for (uint32_t b0 = 0; b0 < ntiles; b0 += 256) {
    uint32_t i = b0 + threadIdx.x;
    uint2 v = i < ntiles ? tile_tot[i] : make_uint2(0, 0);
    sh[threadIdx.x] = v; __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      uint2 o = threadIdx.x >= (uint32_t)d ? sh[threadIdx.x - d] : make_uint2(0, 0);
      __syncthreads();
      sh[threadIdx.x].x += o.x; sh[threadIdx.x].y += o.y; __syncthreads();
    }
    uint2 inc = sh[threadIdx.x]; uint2 cr = carry;
    if (i < ntiles) scan_blk[i] = make_uint2(cr.x + inc.x - v.x, cr.y + inc.y - v.y);
    __syncthreads();
    if (threadIdx.x == 255) { carry.x = cr.x + inc.x; carry.y = cr.y + inc.y; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    meta[0] = carry.y; meta[2] = carry.x;
    if (host_meta) {
      host_meta[0] = carry.y; host_meta[2] = carry.x;
      for (int i : {1, 3, 4, 5, 6, 7}) host_meta[i] = meta[i];      // written by k_scan_tiles (the launch before this one)
      __threadfence_system();
      host_meta[8] = seq;
    }
  }
}

// ---- slice ordering: lanes of one wave should run the same trip count --------------------------------
// Slices are at most 512 points long; bucket sizes are Poisson, so slice lengths vary 2:1 inside a wave if
// taken in bucket order (measured: 31 % of the accumulation's lanes idle).  A counting sort by length (longest
// first) costs two tiny launches: block-local LDS histograms + a handful of global atomics per block.
// sid -> bucket (binary search over first_slice), stores task_g[sid], counts slice lengths
// FUSED (round 5, <= 512 scan tiles — every chain of a prover round): the exclusive scan of the tile totals, a single-block launch of its own until now (k_scan_top,
// ~5-8 us per chain at real-circuit sizes), runs in every block's prologue over LDS; block 0 also leaves scan_blk, the totals and the host's copy of the slice
// metadata behind for the kernels that follow (k_slice_order, the accumulation, the trees and the reduction read them from memory as before).
static constexpr uint32_t FUSED_TILES = 512;
template <bool FUSED>
__global__ void __launch_bounds__(256) k_slice_count(const uint32_t* __restrict__ hist, const uint2* __restrict__ scan_local, const uint2* scan_blk_in,
                                                     uint32_t M, const uint32_t* __restrict__ total_pairs, uint32_t* meta, uint32_t* __restrict__ task_g,
                                                     uint32_t* __restrict__ len_count, const uint2* __restrict__ tile_tot, uint32_t ntiles, uint2* scan_blk_out,
                                                     volatile uint32_t* host_meta, uint32_t seq) {
  __shared__ uint32_t h[MAX_SLICE + 1];
  __shared__ uint2 sblk[FUSED ? FUSED_TILES : 1]; __shared__ uint2 wtot[4]; __shared__ uint32_t s_total;
  for (uint32_t i = threadIdx.x; i <= MAX_SLICE; i += 256) h[i] = 0;
  uint32_t total_slices;
  if constexpr (FUSED) {
    const uint32_t tid = threadIdx.x; const int lane = tid & 63, wv = tid >> 6;
    const uint2 v0 = 2 * tid < ntiles ? tile_tot[2 * tid] : make_uint2(0u, 0u), v1 = 2 * tid + 1 < ntiles ? tile_tot[2 * tid + 1] : make_uint2(0u, 0u);
    uint2 inc = make_uint2(v0.x + v1.x, v0.y + v1.y);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t ox = __shfl_up(inc.x, d), oy = __shfl_up(inc.y, d); if (lane >= d) { inc.x += ox; inc.y += oy; } }
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    uint2 off = make_uint2(0u, 0u); for (int k = 0; k < wv; ++k) { off.x += wtot[k].x; off.y += wtot[k].y; }
    const uint2 excl = make_uint2(off.x + inc.x - v0.x - v1.x, off.y + inc.y - v0.y - v1.y);
    if (2 * tid < FUSED_TILES) sblk[2 * tid] = excl;
    if (2 * tid + 1 < FUSED_TILES) sblk[2 * tid + 1] = make_uint2(excl.x + v0.x, excl.y + v0.y);
    if (tid == 255) s_total = off.y + inc.y;
    if (blockIdx.x == 0) {
      if (2 * tid < ntiles) scan_blk_out[2 * tid] = excl;
      if (2 * tid + 1 < ntiles) scan_blk_out[2 * tid + 1] = make_uint2(excl.x + v0.x, excl.y + v0.y);
      if (tid == 255) {
        const uint32_t slices = off.y + inc.y, pairs = off.x + inc.x;
        meta[0] = slices; meta[2] = pairs;
        if (host_meta) {
          host_meta[0] = slices; host_meta[2] = pairs;
          for (int i : {1, 3, 4, 5, 6, 7}) host_meta[i] = meta[i];      // written by k_scan_tiles (the launch before this one)
          __threadfence_system();
          host_meta[8] = seq;
        }
      }
    }
    __syncthreads();
    total_slices = s_total;
  } else {
    __syncthreads();
    total_slices = meta[0];
  }
  auto at = [&](uint32_t g) -> uint2 {
    const uint2 a = scan_local[g]; uint2 b;
    if constexpr (FUSED) b = sblk[g / SCAN_TILE]; else b = scan_blk_in[g / SCAN_TILE];
    return make_uint2(a.x + b.x, a.y + b.y);
  };
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t < total_slices) {
    uint32_t lo = 0, hi = M - 1;        // largest g with first_slice(g) <= t
    while (lo < hi) {
      uint32_t mid = (lo + hi + 1) >> 1;
      if (at(mid).y <= t) lo = mid; else hi = mid - 1;
    }
    uint32_t g = lo, cnt = hist[g], m = slices_of(cnt, pick_rule(total_pairs, M)), k = t - at(g).y;
    task_g[t] = g;
    atomicAdd(&h[slice_len(cnt, m, k)], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i <= MAX_SLICE; i += 256) if (h[i]) atomicAdd(&len_count[i], h[i]);
}

// order[pos] = sid, longest slices first
// (len_start[l] = number of slices longer than l is recomputed by every block from the ~257 length counts — a single-block launch of its own, k_len_starts,
//  cost ~6 us per chain at the sizes of real circuits)
__global__ void __launch_bounds__(256) k_slice_order(const uint32_t* __restrict__ hist, const uint2* __restrict__ scan_local, const uint2* __restrict__ scan_blk,
                                                     const uint32_t* __restrict__ total_pairs, uint32_t M, const uint32_t* __restrict__ meta, const uint32_t* __restrict__ task_g,
                                                     const uint32_t* __restrict__ len_count, uint32_t* __restrict__ len_cursor, uint32_t* __restrict__ order) {
  __shared__ uint32_t h[MAX_SLICE + 1], base[MAX_SLICE + 1], len_start[MAX_SLICE + 2], wtot[4];
  {                                                        // suffix sums of len_count: lane t owns the lengths PER t .. PER t + PER - 1
    constexpr uint32_t PER = (MAX_SLICE + 1 + 255) / 256;
    const uint32_t t = threadIdx.x; uint32_t cnt[PER], tot = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) { const uint32_t l = PER * t + k; cnt[k] = l <= MAX_SLICE ? len_count[l] : 0u; tot += cnt[k]; }
    uint32_t inc = tot; const int lane = t & 63, wv = t >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_down(inc, d); if (lane + d < 64) inc += o; }      // inclusive suffix sum inside the wave
    if (lane == 0) wtot[wv] = inc;
    __syncthreads();
    uint32_t run = inc - tot; for (int k = wv + 1; k < 4; ++k) run += wtot[k];      // slices longer than this lane's last length
#pragma unroll
    for (uint32_t k = PER; k-- > 0;) { const uint32_t l = PER * t + k; if (l <= MAX_SLICE) len_start[l] = run; run += cnt[k]; }
  }
  for (uint32_t i = threadIdx.x; i <= MAX_SLICE; i += 256) h[i] = 0;
  __syncthreads();
  uint32_t t = blockIdx.x * 256 + threadIdx.x, len = 0, rank = 0;
  bool live = t < meta[0];
  if (live) {
    uint32_t g = task_g[t], cnt = hist[g], m = slices_of(cnt, pick_rule(total_pairs, M)), k = t - scan_at(scan_local, scan_blk, g).y;
    len = slice_len(cnt, m, k);
    rank = atomicAdd(&h[len], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i <= MAX_SLICE; i += 256)
    if (h[i]) base[i] = len_start[i] + atomicAdd(&len_cursor[i], h[i]);
  __syncthreads();
  if (live) order[base[len] + rank] = t;
}

// ---- bucket accumulation: one lane per slice, in the 14 x 28-bit representation (fp28.h) -----------------------------
// Point rows are 112 bytes (x'[14] | y'[14], value * 2^392 mod q as exact base-2^28 digits): table rows on the fixed-base path,
// the pinned set's own 28-bit rows on the plain path.  OUT28: the slice sum is stored as a 224-byte 28-bit XYZZ point (the
// table path's reduction continues in that form); otherwise it is converted to a 32-bit XYZZ point for the plain path's
// per-window reduction kernels.
__device__ __forceinline__ XYZZ xyzz28_to_xyzz(const XYZZ28& a) {
  XYZZ r; r.X = f28_to_fq(a.X); r.Y = f28_to_fq(a.Y); r.ZZ = f28_to_fq(a.ZZ); r.ZZZ = f28_to_fq(a.ZZZ); return r;     // all < 2q
}
__device__ __noinline__ void slice_slow_path28(const char* bases, const uint32_t* run, uint32_t j, uint32_t j1, const XYZZ28* acc28, XYZZ* acc_out, bool* inf_out) {
  XYZZ acc = xyzz28_to_xyzz(*acc28); bool inf = false;
  for (; j < j1; ++j) {
    uint32_t e = run[j];
    F28 x, y; load_affine28(bases + (size_t)(e & 0x7fffffffu) * ROW28, x, y);
    AffinePt p; p.x = Fq::reduce(f28_to_fq(x)); p.y = Fq::reduce(f28_to_fq(y));
    if (e >> 31) p.y = fq_neg_canonical(p.y);
    xyzz_madd(acc, inf, p.x, p.y);
  }
  *acc_out = acc; *inf_out = inf;
}

// The bucket sums earlier launch chains of the SAME request left behind (msm_run_chunked: one MSM whose scalars arrive in chunks, every chunk sorted and
// accumulated on its own, all chunks addressing the same buckets): newest first.  The first slice of bucket g in the current chunk starts from the newest
// earlier sum of g instead of from its own first point, so after the last chunk a bucket's total sits in the newest chunk that touched it.
struct FrontView { const char* partial; const uint32_t* hist; const uint2* scan_local; const uint2* scan_blk; };
struct FrontChain { FrontView v[3]; uint32_t n = 0; };
__device__ __forceinline__ const char* chain_sum(const FrontChain& ch, uint32_t g) {
  for (uint32_t i = 0; i < ch.n; ++i) if (ch.v[i].hist[g]) return ch.v[i].partial + (size_t)scan_at(ch.v[i].scan_local, ch.v[i].scan_blk, g).y * 224u;
  return nullptr;
}

template <bool OUT28, bool SEED = false>
__global__ void __launch_bounds__(256) k_accum28(const char* __restrict__ bases, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ hist,
                                                 const uint2* __restrict__ scan_local, const uint2* __restrict__ scan_blk, const uint32_t* __restrict__ total_pairs, uint32_t M,
                                                 const uint32_t* __restrict__ meta, const uint32_t* __restrict__ order, const uint32_t* __restrict__ task_g,
                                                 char* __restrict__ partial, FrontChain seed) {
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= meta[0]) return;
  const uint32_t sid = order[t], g = task_g[sid];
  uint2 st = scan_at(scan_local, scan_blk, g);
  uint32_t cnt = hist[g], m = slices_of(cnt, pick_rule(total_pairs, M)), k = sid - st.y;
  uint32_t j0 = (uint32_t)(((uint64_t)k * cnt) / m), j1 = (uint32_t)(((uint64_t)(k + 1) * cnt) / m);
  const uint32_t* run = sorted + st.x;
  uint32_t e_next = run[j0];
  F28 xn, yn; load_affine28(bases + (size_t)(e_next & 0x7fffffffu) * ROW28, xn, yn);      // next point's 112-byte gather in flight under the current addition
  XYZZ28 acc; bool ok = true;
  uint32_t j = j0;
  bool seeded = false;
  if constexpr (SEED) {
    // first slice of the bucket: continue from what the earlier chunks of this request summed into the same bucket (a stored point: Y may be loose,
    // class L3 < 6q — one product by R brings it to the loop's invariant, exact digits < 2q); an empty or identity sum starts the usual way
    const char* sp = k == 0 ? chain_sum(seed, g) : nullptr;
    if (sp) {
      acc.ZZ = load_f28(sp + 112);
      if (!f28_is_zero_raw(acc.ZZ)) { acc.X = load_f28(sp); acc.Y = f28_mul(load_f28(sp + 56), f28_const(ONE28)); acc.ZZZ = load_f28(sp + 168); seeded = true; }
    }
  }
  if (!seeded) {   // first point of the slice: acc = (x, +-y, 1, 1)
    uint32_t e = e_next; F28 x = xn, y = yn;
    if (j + 1 < j1) { e_next = run[j + 1]; load_affine28(bases + (size_t)(e_next & 0x7fffffffu) * ROW28, xn, yn); }
    if (e >> 31) y = f28_sub<2, 1>(f28_const(Limbs14{}), y);                            // 2q - y: limbs < 2^29
    acc.X = x; acc.Y = y; acc.ZZ = f28_const(ONE28); acc.ZZZ = f28_const(ONE28);
    ++j;
  }
  for (; j < j1; ++j) {
    uint32_t e = e_next; F28 x = xn, y = yn;
    if (j + 1 < j1) { e_next = run[j + 1]; load_affine28(bases + (size_t)(e_next & 0x7fffffffu) * ROW28, xn, yn); }
    if (e >> 31) y = f28_sub<2, 1>(f28_const(Limbs14{}), y);
    if (!xyzz28_madd_fast(acc, x, y)) { ok = false; break; }
  }
  if (ok) {
    if constexpr (OUT28) store_xyzz28(partial + (size_t)sid * 224, acc);      // X exact < 12q, Y exact < 2q, ZZ / ZZZ exact < 2q: the stored invariant of fp28.h
    else xyzz_store_normalized(partial + (size_t)sid * 192, xyzz28_to_xyzz(acc), false);   // plain path: its reduction kernels work on 32-bit points
  } else {    // P == +-acc (repeated or opposite bases): finish the slice with the general 32-bit code, out of line
    XYZZ28 tmp = acc; XYZZ out; bool inf = false;
    slice_slow_path28(bases, run, j, j1, &tmp, &out, &inf);
    if constexpr (OUT28) store_xyzz28_from32(partial + (size_t)sid * 224, out, inf);
    else xyzz_store_normalized(partial + (size_t)sid * 192, out, inf);
  }
}

// 96-byte rows (x | y, 12 x 32-bit Montgomery) -> 112-byte rows of the 28-bit table; (0, 0) marks the identity and stays 0
__global__ void __launch_bounds__(256) k_rows_to28(const char* __restrict__ src96, char* __restrict__ dst112, uint32_t n) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  AffinePt p = load_affine(src96 + (size_t)i * 96);
  F28 x, y;
  if (p.x.is_zero_raw() && p.y.is_zero_raw()) { x = f28_const(Limbs14{}); y = x; }
  else { x = f28_from_fq(p.x); y = f28_from_fq(p.y); }
  store_affine28(dst112 + (size_t)i * ROW28, x, y);
}

// Every kernel from here to the host tail is a chain of full XYZZ additions with little parallelism, so each addition
// is shared by a lane pair (ec.h xyzz_add_pair: same work, half the latency).  "op" below = pair index = thread / 2.
__device__ __forceinline__ void pair_fence() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
// Point formats of the partial sums: 32-bit XYZZ (192 B, ec.h) on the plain path, 28-bit XYZZ (224 B, fp28.h) on the table path,
// where everything after the accumulation kernel — slice tree, chunk running sums, masked sums, segment folds — stays in the
// representation that kernel computes in (its products are ~14 % cheaper and nothing is converted on the device); the host
// tail turns the lg(N) + 4 final points into its 64-bit-limb Montgomery form.
template <bool F28> struct PtFmt { static constexpr uint32_t BYTES = F28 ? 224u : 192u, WORDS = BYTES / 4; };
template <uint32_t BYTES = 192> __device__ __forceinline__ void pair_copy(const char* src, char* dst) {      // half per lane
  const uint32_t o = (threadIdx.x & 1) ? BYTES / 2 : 0;
  const uint4* s4 = (const uint4*)(src + o); uint4* d4 = (uint4*)(dst + o);
#pragma unroll
  for (int i = 0; i < (int)(BYTES / 32); ++i) d4[i] = s4[i];
}
template <uint32_t BYTES = 192> __device__ __forceinline__ void pair_zero(char* dst) {
  uint4* d4 = (uint4*)(dst + ((threadIdx.x & 1) ? BYTES / 2 : 0));
#pragma unroll
  for (int i = 0; i < (int)(BYTES / 32); ++i) d4[i] = make_uint4(0, 0, 0, 0);
}
template <bool F28> __device__ __forceinline__ void pt_add_pair(const char* pa, const char* pb, char* out) {
  if constexpr (F28) xyzz28_add_pair(pa, pb, out); else xyzz_add_pair(pa, pb, out);
}
// The same chains with FOUR lanes per addition (fp28.h xyzz28_add_quad: four product levels instead of seven, 28-bit points only): taken while a
// launch leaves the chip latency-bound (grp_lanes below), the pair form where the additions of a level already fill it (14 of a quad's 16 product slots work).
template <uint32_t LANES> __device__ __forceinline__ void pt28_add(const char* pa, const char* pb, char* out) {
  if constexpr (LANES == 4) xyzz28_add_quad(pa, pb, out); else xyzz28_add_pair(pa, pb, out);
}
template <uint32_t LANES, bool F28> __device__ __forceinline__ void pt_add_grp(const char* pa, const char* pb, char* out) {
  if constexpr (LANES == 4) { static_assert(F28, "quad additions work on 28-bit points"); xyzz28_add_quad(pa, pb, out); } else pt_add_pair<F28>(pa, pb, out);
}
template <uint32_t LANES, uint32_t BYTES> __device__ __forceinline__ void grp_copy(const char* src, char* dst) {      // BYTES / LANES per lane
  const uint32_t o = (threadIdx.x & (LANES - 1)) * (BYTES / LANES);
  const uint2* s2 = (const uint2*)(src + o); uint2* d2 = (uint2*)(dst + o);
#pragma unroll
  for (int i = 0; i < (int)(BYTES / LANES / 8); ++i) d2[i] = s2[i];
}
template <uint32_t LANES, uint32_t BYTES> __device__ __forceinline__ void grp_zero(char* dst) {
  uint2* d2 = (uint2*)(dst + (threadIdx.x & (LANES - 1)) * (BYTES / LANES));
#pragma unroll
  for (int i = 0; i < (int)(BYTES / LANES / 8); ++i) d2[i] = make_uint2(0, 0);
}
constexpr uint32_t lg_lanes(uint32_t lanes) { return lanes == 4 ? 2u : 1u; }
// lanes per addition for a launch of `ops` independent additions: quads up to two waves per SIMD (2^17 lanes), pairs beyond.  Measured: k_seg_fold 90 -> 57 us,
// k_tree_pass 15 -> 11 us, a 2^15-constraint proof 6.9 -> 6.6 ms; with the cut at 2^16 lanes the proof is at 6.8 ms.  (The 2^15-chunk kernel of the
// widest window is the exception: 2^17 quad lanes take what 2^16 pair lanes take, 330 against 321 us — it keeps the pair form.)
static inline bool quads_on() { static const bool v = [] { const char* e = std::getenv("ALEO_MI355X_QUAD_ADD"); return !(e && e[0] == '0'); }(); return v; }      // A/B switch: 0 = lane pairs everywhere
static constexpr uint32_t ASIDE_MAX = 8;  // super-heavy buckets whose slice trees may run beside the reduction (msm_run)
static inline bool aside_on() { static const bool v = [] { const char* e = std::getenv("ALEO_MI355X_ASIDE"); return !(e && e[0] == '0'); }(); return v; }      // A/B switch
static inline bool prog_on() { static const bool v = [] { const char* e = std::getenv("ALEO_MI355X_SUM_TREE"); return !(e && e[0] == '0'); }(); return v; }      // A/B switch: 0 = masked trees on the wide tables too
static inline uint32_t quad_max_lg() { static const uint32_t v = [] { const char* e = std::getenv("ALEO_MI355X_QUAD_MAX_LG"); const int k = e ? std::atoi(e) : 17; return (uint32_t)(k >= 10 && k <= 24 ? k : 17); }(); return v; }      // A/B: lg of the most quad lanes a launch may have
static inline uint32_t grp_lanes(uint64_t ops) { return quads_on() && ops * 4 <= ((uint64_t)1 << quad_max_lg()) ? 4u : 2u; }
// partial[ft + i] += partial[ft + i + half] inside every multi-slice bucket: one launch per level serves both lists of the scan —
// the common one (buckets of <= 16 slices, `pairs_a` lane pairs each) and the super-heavy one (`pairs_b` each; skewed scalars).
template <bool F28, uint32_t LANES = 2>
__global__ void __launch_bounds__(256) k_tree_pass(char* __restrict__ partial, const uint32_t* __restrict__ list_a, uint32_t len_a, uint32_t pairs_a,
                                                   const uint32_t* __restrict__ list_b, uint32_t len_b, uint32_t pairs_b, const uint2* __restrict__ scan_local,
                                                   const uint2* __restrict__ scan_blk, uint32_t M, const uint32_t* __restrict__ meta, uint32_t pass, uint32_t skip_b) {
  constexpr uint32_t PB = PtFmt<F28>::BYTES;
  uint32_t op = (blockIdx.x * 256 + threadIdx.x) >> lg_lanes(LANES);
  const uint32_t ops_a = len_a * pairs_a;
  const uint32_t* list = list_a; uint32_t max_pairs = pairs_a, list_len = len_a, skip = 0;
  if (op >= ops_a) { op -= ops_a; list = list_b; max_pairs = pairs_b; list_len = len_b; skip = skip_b; }
  if (max_pairs == 0) return;
  uint32_t h = op / max_pairs, i = op % max_pairs;
  if (h >= list_len) return;
  uint32_t g = list[h];
  uint32_t ft = scan_at(scan_local, scan_blk, g).y + skip;      // skip_b = 1: the tree of slices 1.. of a super-heavy bucket (slice 0 stays the bucket's sum for the reduction; msm_run "aside")
  uint32_t fn = (g + 1 < M) ? scan_at(scan_local, scan_blk, g + 1).y : meta[0];
  uint32_t L = fn - ft;
  for (uint32_t p = 0; p < pass; ++p) L = (L + 1) >> 1;
  if (L <= 1) return;
  uint32_t half = (L + 1) >> 1;
  if (i >= L - half) return;
  char* pa = partial + (size_t)(ft + i) * PB;
  pt_add_grp<LANES, F28>(pa, pa + (size_t)half * PB, pa);
}

// What is left of the common list's trees after the first level (at most 4 partial sums per bucket when no bucket had more than 8 slices) folded by ONE launch:
// a lane quad per multi-slice bucket adds its partials 1.. into partial 0 one after the other.  Two or three dependent additions of ~7 us inside one launch
// instead of two launches of one level each (~10 us of launch floor + its addition per level): the chains of real-circuit-sized proofs are made of such steps.
template <uint32_t LANES>
__global__ void __launch_bounds__(256) k_tree_rest(char* __restrict__ partial, const uint32_t* __restrict__ list, uint32_t list_len, const uint2* __restrict__ scan_local,
                                                   const uint2* __restrict__ scan_blk, uint32_t M, const uint32_t* __restrict__ meta) {
  constexpr uint32_t PB = PtFmt<true>::BYTES;
  const uint32_t h = (blockIdx.x * 256 + threadIdx.x) >> lg_lanes(LANES);
  if (h >= list_len) return;
  const uint32_t g = list[h];
  const uint32_t ft = scan_at(scan_local, scan_blk, g).y, fn = (g + 1 < M) ? scan_at(scan_local, scan_blk, g + 1).y : meta[0];
  const uint32_t L1 = (fn - ft + 1) >> 1;                   // partial sums the first level left at ft .. ft + L1 - 1
  char* pa = partial + (size_t)ft * PB;
  for (uint32_t i = 1; i < L1; ++i) pt_add_grp<LANES, true>(pa, pa + (size_t)i * PB, pa);
}

// The sums of slices 1.. of the super-heavy buckets (k_tree_pass with skip_b = 1 left them in slice 1) and the buckets' numbers, to the host.
__global__ void k_gather_super(const char* __restrict__ partial, const uint32_t* __restrict__ list, uint32_t len, const uint2* __restrict__ scan_local,
                               const uint2* __restrict__ scan_blk, uint32_t* __restrict__ dst) {
  constexpr uint32_t PW = PtFmt<true>::WORDS;
  const uint32_t t = blockIdx.x * 256 + threadIdx.x, h = t / (PW + 1), w = t % (PW + 1);
  if (h >= len) return;
  const uint32_t g = list[h];
  dst[h * (PW + 1) + w] = w == PW ? g : ((const uint32_t*)(partial + (size_t)(scan_at(scan_local, scan_blk, g).y + 1) * PtFmt<true>::BYTES))[w];
}

// ---- bucket reduction -----------------------------------------------------------------------------
// One lane PAIR per chunk of S consecutive buckets of one window: running sums run += S_b, acc += run (b descending)
// kept in LDS between the cooperative additions, so acc = sum_{b in chunk} (b - base + 1) * S_b and run = chunk total.
// Both go to HBM (V, Vrun); the chunk weights are applied by masked sums (fixed-base path).
static constexpr uint32_t CHUNK_PAIRS = 128;        // chunks per 256-thread block (LANES = 2; 64 with quads)
template <bool F28, uint32_t LANES = 2>
__global__ void __launch_bounds__(256) k_bucket_chunks_pair(const char* __restrict__ partial, const uint32_t* __restrict__ hist, const uint2* __restrict__ scan_local,
                                                       const uint2* __restrict__ scan_blk, uint32_t B, uint32_t S, uint32_t nchunks_total, char* __restrict__ V,
                                                       uint32_t v_set_stride, char* __restrict__ Vrun, FrontChain older) {
  constexpr uint32_t PB = PtFmt<F28>::BYTES, PW = PtFmt<F28>::WORDS;
  constexpr uint32_t CPB = 256 / LANES;                   // chunks per block
  __shared__ __attribute__((aligned(16))) uint32_t lds[2 * CPB * PW];
  const uint32_t pr = threadIdx.x >> lg_lanes(LANES), t = blockIdx.x * CPB + pr;
  if (t >= nchunks_total) return;
  char* run = (char*)(lds + pr * PW); char* acc = (char*)(lds + (CPB + pr) * PW);
  grp_zero<LANES, PB>(run); grp_zero<LANES, PB>(acc);
  pair_fence();
  const uint32_t cpw = B / S, w = t / cpw, j = t % cpw, g0 = w * B + j * S;
  // the address of bucket k + 1's sum (two dependent loads: histogram, scan) is fetched while bucket k's two additions run
  auto sum_of = [&](uint32_t g) -> const char* { return hist[g] ? partial + (size_t)scan_at(scan_local, scan_blk, g).y * PB : chain_sum(older, g); };      // older: buckets only earlier chunks of the request touched
  const char* nxt = sum_of(g0 + S - 1);
  for (uint32_t k = 0; k < S; ++k) {
    const char* cur = nxt;
    if (k + 1 < S) nxt = sum_of(g0 + S - 2 - k);
    if (cur) { pt_add_grp<LANES, F28>(run, cur, run); pair_fence(); }
    pt_add_grp<LANES, F28>(acc, run, acc); pair_fence();
  }
  grp_copy<LANES, PB>(run, Vrun + (size_t)t * PB); grp_copy<LANES, PB>(acc, V + ((size_t)w * v_set_stride + j) * PB);
}

// One lane QUAD per chunk of S consecutive buckets: running sums run_k = run_{k-1} + S_b (b descending) and acc += run_{k-1}
// are independent once run_{k-1} exists, so two lane pairs work one step apart (S + 1 dependent additions instead of 2S):
// sub-pair 0 extends the running sum (double-buffered in LDS), sub-pair 1 folds the previous one into acc.  Both make the
// SAME addition call with per-lane pointers (a branch per sub-pair would serialise them inside the wave).
// acc = sum_{b in chunk} (b - base + 1) * S_b and run = chunk total go to HBM (V, Vrun); the chunk weights are applied by
// masked sums (fixed-base path).
static constexpr uint32_t CHUNK_QUADS = 64;         // chunks per 256-thread block (LANES = 2: two lane pairs per chunk; 32 with LANES = 4: two lane quads)
template <bool F28, uint32_t LANES = 2>
__global__ void __launch_bounds__(256) k_bucket_chunks(const char* __restrict__ partial, const uint32_t* __restrict__ hist, const uint2* __restrict__ scan_local,
                                                       const uint2* __restrict__ scan_blk, uint32_t B, uint32_t S, uint32_t nchunks_total, char* __restrict__ V,
                                                       uint32_t v_set_stride, char* __restrict__ Vrun, FrontChain older) {
  constexpr uint32_t PB = PtFmt<F28>::BYTES, PW = PtFmt<F28>::WORDS;
  constexpr uint32_t CPB = 128 / LANES;                  // chunks per block: two groups of LANES lanes each
  __shared__ __attribute__((aligned(16))) uint32_t lds[(3 * CPB + 1) * PW];
  const uint32_t qd = threadIdx.x >> (lg_lanes(LANES) + 1), sp = (threadIdx.x >> lg_lanes(LANES)) & 1u, t = blockIdx.x * CPB + qd;
  char* zero = (char*)(lds + 3 * CPB * PW);
  if (threadIdx.x < LANES) grp_zero<LANES, PB>(zero);
  __syncthreads();
  if (t >= nchunks_total) return;
  char* buf0 = (char*)(lds + (3 * qd) * PW); char* buf1 = buf0 + PB; char* acc = buf1 + PB;
  grp_zero<LANES, PB>(sp ? acc : buf0); if (!sp) grp_zero<LANES, PB>(buf1);
  pair_fence();
  const uint32_t cpw = B / S, w = t / cpw, j = t % cpw, g0 = w * B + j * S;
  // the address of the next bucket's sum (two dependent loads: histogram, scan) is fetched one step ahead of the addition that uses it
  auto sum_of = [&](uint32_t g) -> const char* { if (hist[g]) return partial + (size_t)scan_at(scan_local, scan_blk, g).y * PB; const char* o = chain_sum(older, g); return o ? o : zero; };
  const char* nxt = sp ? zero : sum_of(g0 + S - 1);
  for (uint32_t k = 0; k <= S; ++k) {
    char* rprev = (k & 1) ? buf0 : buf1; char* rnext = (k & 1) ? buf1 : buf0;      // run_k lives in buf[k & 1]; run_{-1} = 0
    const char* add = k < S ? nxt : zero;
    nxt = (!sp && k + 1 < S) ? sum_of(g0 + S - 2 - k) : zero;
    const char* pa = sp ? acc : rprev; const char* pb = sp ? rprev : add; char* out = sp ? acc : rnext;
    pt_add_grp<LANES, F28>(pa, pb, out);
    pair_fence();
  }
  // after step S: buf[S & 1] holds run_{S-1} again (step S copied it forward), acc holds sum_k run_k
  grp_copy<LANES, PB>(sp ? acc : ((S & 1) ? buf1 : buf0), sp ? V + ((size_t)w * v_set_stride + j) * PB : Vrun + (size_t)t * PB);
}

// Plain path (one window set per window): one lane per chunk, V = sum_{b in chunk} (b+1) * S_b with the chunk base applied
// by double-and-add.  (The pair form loses here: the double-and-add tail is most of the chain and would idle odd lanes.)
__global__ void __launch_bounds__(256) k_bucket_chunks_plain(const char* __restrict__ partial, const uint32_t* __restrict__ hist, const uint2* __restrict__ scan_local,
                                                             const uint2* __restrict__ scan_blk, uint32_t B, uint32_t S, uint32_t nchunks_total, char* __restrict__ V) {
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nchunks_total) return;
  uint32_t cpw = B / S, w = t / cpw, j = t % cpw;
  uint32_t g0 = w * B + j * S;
  XYZZ run = xyzz_infinity(), acc = xyzz_infinity();
  // steps 2k: run += S_b (b descending); steps 2k+1: acc += run   (one inlined xyzz_add call site)
  for (uint32_t k = 0; k < 2 * S; ++k) {
    bool odd = k & 1;
    XYZZ y;
    if (!odd) {
      uint32_t g = g0 + (S - 1 - (k >> 1));
      if (hist[g]) y = load_xyzz(partial + (size_t)scan_at(scan_local, scan_blk, g).y * 192); else y = xyzz_infinity();
    } else y = run;
    XYZZ x = odd ? acc : run;
    xyzz_add(x, y);
    if (odd) acc = x; else run = x;
  }
  uint32_t base = j * S;
  if (base) {
    XYZZ r = xyzz_infinity();
    for (int bit = 31 - __clz(base); bit >= 0; --bit) {
      xyzz_double_ni(&r);
      if ((base >> bit) & 1) xyzz_add_ni(&r, &run);
    }
    xyzz_add_ni(&acc, &r);
  }
  store_xyzz(V + (size_t)t * 192, acc);
}

// V[seg*seg_len + i] += V[seg*seg_len + i + half] for i < L - half, L = current length of every segment
__global__ void __launch_bounds__(256) k_seg_tree_pass(char* __restrict__ V, uint32_t seg_len, uint32_t nseg, uint32_t L) {
  const uint32_t half = (L + 1) >> 1, pairs = L - half;
  const uint32_t op = (blockIdx.x * 256 + threadIdx.x) >> 1;
  if (op >= pairs * nseg) return;
  const uint32_t seg = op / pairs, i = op % pairs;
  char* pa = V + ((size_t)seg * seg_len + i) * 192;
  xyzz_add_pair(pa, pa + (size_t)half * 192, pa);
}

// Fixed-base path: sum_j j * run_j = sum_l 2^l * T_l with T_l = sum of run_j over the j that have bit l set.  This
// kernel does the first pairwise level of all lg(N) masked sums at once: T_l[k] = run[ins_l(2k)] + run[ins_l(2k+1)],
// ins_l(x) = x with a 1 inserted at bit l.  The remaining levels are k_seg_pair_pass / k_seg_fold; the 2^l Horner runs
// on the host.
// (These kernels only run on the table path, whose points are 224-byte 28-bit XYZZ: PB below.)
static constexpr uint32_t PB28 = 224, PW28 = 56;
// Sets (batched calls): set q reads Vrun[q * 2^lgN ...] and writes its lgN sums behind its chunk sums, at
// V[q * v_set_stride + 2^lgN ...] (v_set_stride = (lgN + 4) * 2^(lgN-2): the set's 4 + lgN segments are contiguous).
template <uint32_t LANES>
__global__ void __launch_bounds__(256) k_masked_pairs(const char* __restrict__ Vrun, uint32_t lgN, uint32_t nsets, char* __restrict__ V, uint32_t v_set_stride) {
  const uint32_t seg_len = 1u << (lgN - 2);
  const uint32_t op = (blockIdx.x * 256 + threadIdx.x) >> lg_lanes(LANES);
  if (op >= seg_len * lgN * nsets) return;
  const uint32_t q = op / (seg_len * lgN), r = op % (seg_len * lgN), l = r / seg_len, k = r % seg_len;
  auto ins = [&](uint32_t x) { return ((x >> l) << (l + 1)) | (1u << l) | (x & ((1u << l) - 1u)); };
  const char* run = Vrun + ((size_t)q << lgN) * PB28;
  pt28_add<LANES>(run + (size_t)ins(2 * k) * PB28, run + (size_t)ins(2 * k + 1) * PB28, V + ((size_t)q * v_set_stride + (1u << lgN) + r) * PB28);
}
// One block folds up to 256 consecutive points of one segment into a single point: 8 tree levels through two LDS
// buffers, 128 lane pairs — the latency floor of the chain with no launch gaps.
static constexpr uint32_t FOLD = 256;
// (the block has FOLD / 2 lane groups: 256 threads as pairs, 512 as quads — the quad form halves the latency of each of the 8 levels)
template <uint32_t LANES>
__global__ void __launch_bounds__(128 * LANES) k_seg_fold(const char* __restrict__ in, uint32_t in_stride, uint32_t L, uint32_t nseg,
                                                          char* __restrict__ out, uint32_t out_stride) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[2][(FOLD / 2) * PW28];
  const uint32_t bps = (L + FOLD - 1) / FOLD, seg = blockIdx.x / bps, blk = blockIdx.x % bps, pr = threadIdx.x >> lg_lanes(LANES);
  if (seg >= nseg) return;
  {
    const uint32_t e0 = blk * FOLD + 2 * pr;
    const char* src = in + ((size_t)seg * in_stride + e0) * PB28;
    char* dst = (char*)(lds[0] + pr * PW28);
    if (e0 + 1 < L) pt28_add<LANES>(src, src + PB28, dst);
    else if (e0 < L) grp_copy<LANES, PB28>(src, dst);
    else grp_zero<LANES, PB28>(dst);
  }
  uint32_t cur = 0;
  for (uint32_t n = FOLD / 2; n > 1; n >>= 1) {
    __syncthreads();
    if (pr < (n >> 1)) pt28_add<LANES>((const char*)(lds[cur] + (2 * pr) * PW28), (const char*)(lds[cur] + (2 * pr + 1) * PW28), (char*)(lds[cur ^ 1] + pr * PW28));
    cur ^= 1;
  }
  __syncthreads();
  if (pr == 0) grp_copy<LANES, PB28>((const char*)lds[cur], out + ((size_t)seg * out_stride + blk) * PB28);
}
// out[seg][i] = in[seg][2i] + in[seg][2i+1]: the wide (throughput-bound) levels of the segment sums
template <uint32_t LANES>
__global__ void __launch_bounds__(256) k_seg_pair_pass(const char* __restrict__ in, uint32_t in_stride, uint32_t L, uint32_t nseg,
                                                       char* __restrict__ out, uint32_t out_stride) {
  const uint32_t half = (L + 1) >> 1;
  const uint32_t op = (blockIdx.x * 256 + threadIdx.x) >> lg_lanes(LANES);
  if (op >= half * nseg) return;
  const uint32_t seg = op / half, i = op % half;
  const char* src = in + ((size_t)seg * in_stride + 2 * i) * PB28;
  char* dst = out + ((size_t)seg * out_stride + i) * PB28;
  if (2 * i + 1 < L) pt28_add<LANES>(src, src + PB28, dst); else grp_copy<LANES, PB28>(src, dst);
}
// ---- the chunk weights by ONE sum tree (wide tables, round 3) -------------------------------------------------------------------------------------
// sum_j j * run_j = sum_l 2^l T_l, T_l = the sum of run_j over the j with bit l set.  k_masked_pairs builds every T_l as its own tree over half of
// the chunks: lg(N) / 2 additions per chunk.  But T_l is also the sum of the RIGHT children of level l of the plain sum tree over the chunks — so one
// pairwise pass per level does it all: node'[i] = node[2i] + node[2i+1], the odd nodes node[2i+1] start the segment T_l, and every segment born
// earlier (T_0 .. T_{l-1}, and A = the chunks' own weighted sums acc_j) is halved the same way.  After pass s every segment is N / 2^(s+1) long:
// (s + 3) N / 2^(s+1) additions per pass, 3 N in all instead of (lg N + 4) N / 2 — which is what lets the chunks shrink (fewer dependent additions
// in k_bucket_chunks) without the weights paying for it.  Layout of a set after pass s: [node | A | T_0 | ... | T_s], contiguous.
template <uint32_t LANES>
__global__ void __launch_bounds__(256) k_prog_pass(const char* __restrict__ node, uint32_t node_set_stride, const char* __restrict__ A, uint32_t a_set_stride,
                                                   const char* __restrict__ T, uint32_t t_set_stride, uint32_t nT, uint32_t L, uint32_t nsets,
                                                   char* __restrict__ out, uint32_t out_set_stride) {
  const uint32_t half = L >> 1, per_set = (2 + nT) * half;
  const uint32_t op = (blockIdx.x * 256 + threadIdx.x) >> lg_lanes(LANES);
  if (op >= per_set * nsets) return;
  const uint32_t q = op / per_set, r = op % per_set, g = r / half, i = r % half;
  const char* src = g == 0 ? node + ((size_t)q * node_set_stride + 2 * i) * PB28
                  : g == 1 ? A + ((size_t)q * a_set_stride + 2 * i) * PB28
                           : T + ((size_t)q * t_set_stride + (size_t)(g - 2) * L + 2 * i) * PB28;
  char* dst = out + ((size_t)q * out_set_stride + (size_t)g * half + i) * PB28;
  pt28_add<LANES>(src, src + PB28, dst);
  if (g == 0) grp_copy<LANES, PB28>(src + PB28, out + ((size_t)q * out_set_stride + (size_t)(2 + nT) * half + i) * PB28);      // T_s is born: the right children of this level
}
// TWO levels in one launch while the passes are latency-bound (lane quads): an octet of lanes per four consecutive points of a segment.  Step 1: quad 0 adds points
// 0 + 1, quad 1 adds 2 + 3, into LDS; step 2: quad 0 adds the two halves (the point of the segment two levels up) while quad 1, on the node segment, adds points
// 1 + 3 (T_nT, born at the first of the two levels, already halved by the second) and copies 2 + 3 out (T_(nT+1), born at the second).  Same sums as two k_prog_pass
// launches — the grouping of the additions differs, the points they represent do not — for two dependent additions and ONE launch instead of two and two:
// ~9 us less per pair of levels (a launch boundary costs about as much as a lane-quad addition).  Output layout as after two single passes:
// [node | A | T_0 .. T_(nT-1) | T_nT | T_(nT+1)], every segment L / 4 long.
__global__ void __launch_bounds__(256) k_prog_pass2(const char* __restrict__ node, uint32_t node_set_stride, const char* __restrict__ A, uint32_t a_set_stride,
                                                    const char* __restrict__ T, uint32_t t_set_stride, uint32_t nT, uint32_t L, uint32_t nsets,
                                                    char* __restrict__ out, uint32_t out_set_stride) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[(2 * 32 + 1) * PW28];
  char* zero = (char*)(lds + 64 * PW28);
  if (threadIdx.x < 4) grp_zero<4, PB28>(zero);
  __syncthreads();
  const uint32_t quarter = L >> 2, per_set = (2 + nT) * quarter;
  const uint32_t oct = threadIdx.x >> 3, y = (threadIdx.x >> 2) & 1u, op = blockIdx.x * 32 + oct;
  if (op >= per_set * nsets) return;
  const uint32_t q = op / per_set, r = op % per_set, g = r / quarter, i = r % quarter;
  const char* src = g == 0 ? node + ((size_t)q * node_set_stride + 4 * i) * PB28
                  : g == 1 ? A + ((size_t)q * a_set_stride + 4 * i) * PB28
                           : T + ((size_t)q * t_set_stride + (size_t)(g - 2) * L + 4 * i) * PB28;
  char* slot = (char*)(lds + (2 * oct) * PW28);
  pt28_add<4>(src + (size_t)(2 * y) * PB28, src + (size_t)(2 * y + 1) * PB28, slot + y * PB28);
  pair_fence();
  char* base = out + (size_t)q * out_set_stride * PB28;
  const bool born = y && g == 0;
  const char* pa = y ? (born ? src + PB28 : zero) : slot;
  const char* pb = y ? (born ? src + 3 * PB28 : zero) : slot + PB28;
  char* dst = y ? (born ? base + ((size_t)(2 + nT) * quarter + i) * PB28 : zero) : base + ((size_t)g * quarter + i) * PB28;      // (an idle quad: 0 + 0 onto the zero point, nothing is written)
  pt28_add<4>(pa, pb, dst);
  if (born) grp_copy<4, PB28>(slot + PB28, base + ((size_t)(3 + nT) * quarter + i) * PB28);
}
// The last levels, one block per result point: A and every T_l already born are plain folds of their (<= 256-point) segments; the weights of the
// remaining lg L bits come from the node segment itself, T_(nT + b) = the sum of the nodes whose index has bit b set.  Output point o of set q:
// 0 = A, 1 + l = T_l.
template <uint32_t LANES>
__global__ void __launch_bounds__(128 * LANES) k_prog_final(const char* __restrict__ in, uint32_t in_set_stride, uint32_t L, uint32_t nT, uint32_t lgL, uint32_t nsets,
                                                            char* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[2][(FOLD / 2) * PW28];
  const uint32_t per_set = 1 + nT + lgL, q = blockIdx.x / per_set, o = blockIdx.x % per_set, pr = threadIdx.x >> lg_lanes(LANES);
  if (q >= nsets) return;
  const char* set = in + (size_t)q * in_set_stride * PB28;
  {
    char* dst = (char*)(lds[0] + pr * PW28);
    if (o <= nT) {                                         // plain fold of segment 1 (A) or 2 + (o - 1) (T_{o-1})
      const char* seg = set + (size_t)(o == 0 ? 1 : 1 + o) * L * PB28; const uint32_t e0 = 2 * pr;
      if (e0 + 1 < L) pt28_add<LANES>(seg + (size_t)e0 * PB28, seg + (size_t)(e0 + 1) * PB28, dst);
      else if (e0 < L) grp_copy<LANES, PB28>(seg + (size_t)e0 * PB28, dst);
      else grp_zero<LANES, PB28>(dst);
    } else {                                               // bit b of the node index: the L / 2 nodes that have it set, in pairs
      const uint32_t b = o - 1 - nT;
      auto ins = [&](uint32_t x) { return ((x >> b) << (b + 1)) | (1u << b) | (x & ((1u << b) - 1u)); };
      if (L >= 4 && 4 * pr + 3 < L) pt28_add<LANES>(set + (size_t)ins(2 * pr) * PB28, set + (size_t)ins(2 * pr + 1) * PB28, dst);
      else if (L == 2 && pr == 0) grp_copy<LANES, PB28>(set + PB28, dst);
      else grp_zero<LANES, PB28>(dst);
    }
  }
  uint32_t cur = 0;
  for (uint32_t n = FOLD / 2; n > 1; n >>= 1) {
    __syncthreads();
    if (pr < (n >> 1)) pt28_add<LANES>((const char*)(lds[cur] + (2 * pr) * PW28), (const char*)(lds[cur] + (2 * pr + 1) * PW28), (char*)(lds[cur ^ 1] + pr * PW28));
    cur ^= 1;
  }
  __syncthreads();
  if (pr == 0) grp_copy<LANES, PB28>((const char*)lds[cur], out + ((size_t)q * per_set + o) * PB28);
}
__global__ void k_gather_strided(const char* __restrict__ V, uint32_t stride, uint32_t count, char* __restrict__ out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * 14) return;
  uint32_t w = t / 14, q = t % 14;
  ((uint4*)out)[t] = ((const uint4*)(V + (size_t)w * stride * PB28))[q];
}

// gathers V[w*seg_len] (the window sums) into a dense array for one D2H copy
__global__ void k_gather_windows(const char* __restrict__ V, uint32_t seg_len, uint32_t W, char* __restrict__ out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= W * 12) return;
  uint32_t w = t / 12, q = t % 12;
  ((uint4*)out)[t] = ((const uint4*)(V + (size_t)w * seg_len * 192))[q];
}

// ---- dispatch on the window width ---------------------------------------------------------------
struct SortArgs { SegArgs segs; const uint8_t* inf; uint32_t nblk_x, row_stride; uint32_t* cnt; uint32_t* off_local; uint32_t* off_blk; uint2* items; };
template <int C, bool MONT, bool PRE> static void launch_sort_c(const SortArgs& a, int phase, hipStream_t s) {
  if (phase == 0) hipLaunchKernelGGL((k_part_count<C, MONT, PRE>), dim3(a.nblk_x, a.segs.nseg), dim3(256), 0, s, a.segs, a.inf, a.cnt);
  else hipLaunchKernelGGL((k_part_scatter<C, MONT, PRE>), dim3(a.nblk_x, a.segs.nseg), dim3(256), 0, s, a.segs, a.inf, a.row_stride, a.off_local, a.off_blk, a.items);
}
template <bool MONT> static void launch_sort(int c, bool pre, const SortArgs& a, int phase, hipStream_t s) {
  if (pre) {
    switch (c) {
      case 13: launch_sort_c<13, MONT, true>(a, phase, s); break;
      case 16: launch_sort_c<16, MONT, true>(a, phase, s); break;
      case 17: launch_sort_c<17, MONT, true>(a, phase, s); break;
      case 20: launch_sort_c<20, MONT, true>(a, phase, s); break;
    }
    return;
  }
  switch (c) {
#define CASE(C) case C: launch_sort_c<C, MONT, false>(a, phase, s); break;
    CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
#undef CASE
  }
}


// lg of the slice count pick_rule() aims for when it cuts buckets to fill the chip (2^17 = 2 waves per SIMD: measured equal to 2^18 on uniform input, 4 % better on witness-like scalars); ALEO_MI355X_FILL_SHIFT
// overrides it for experiments
static uint32_t fill_shift() {
  static const uint32_t v = [] { const char* e = std::getenv("ALEO_MI355X_FILL_SHIFT"); int k = e ? std::atoi(e) : 17; return (uint32_t)(k >= 14 && k <= 22 ? k : 17); }();
  return v;
}

// Upper bound of the slice count the device will compute (k_scan_tiles / pick_rule), from what the host knows: `pairs_max`
// (>= the real pair count) and the bucket count M.  Every non-empty bucket is at least one slice; a bucket cut at `split`
// adds cnt / split more.  pick_rule's split is >= 8 (4 below 2^18 pairs) always; it is >= pairs / 2^fill_shift / 1.125 while the fill rule decides and
// >= the mean bucket size while the mean rule decides, until the 256-point cap takes over.  The grids of the slice kernels
// and of the accumulation are sized by this bound, so no launch waits for the device's own count to reach the host.
static size_t slice_bound(size_t pairs_max, size_t M) {
  const size_t nonempty = M < pairs_max ? M : pairs_max;
  const size_t fill_cap = ((size_t)9 << fill_shift()) >> 3;          // pairs / fill < 1.125 * 2^fill_shift while the fill rule decides
  const size_t by_rule = M + fill_cap + pairs_max / 256;
  const size_t by_min = pairs_max / 4;          // pick_rule's shortest split (the device decides 4 or 8 from its own pair count, which may be far below pairs_max)
  return nonempty + (by_min < by_rule ? by_min : by_rule) + 1;
}

// One call = `job.k` independent MSMs ("sets") over prefixes of ONE pinned base set: set q multiplies the first lens[q] bases
// by the scalars at d_scalars + q * set_stride bytes.  k > 1 needs a fixed-base table tier that serves job.n (the longest
// set) and k <= msm_max_sets(): the sets then share every launch — one sort over k * 2^(c-1) buckets, one accumulation, one
// reduction — which is what the commitments of one prover round need (2^14..2^17 points each: alone they are latency-bound).
uint32_t msm_max_sets(const PinnedBases& pb, size_t n) {
  for (const auto& t : pb.tab) if (t.d && n >= t.min_n && n <= t.cover) {
    const uint32_t B = 1u << (t.c - 1), LB = 8, cb = B >> LB, cap = MAX_COARSE_ALL / (cb ? cb : 1);
    return cap < MAX_SETS ? (cap ? cap : 1) : MAX_SETS;
  }
  return 1;
}

int32_t msm_sort_phase(Ctx* c, SegArgs& segs, size_t pts, bool mont, const uint8_t* d_inf, uint32_t row_stride,
                       const MsmPlan& P, bool pre, hipStream_t s, SortPhase* out, bool lean) {
  SortPhase& sp = *out; sp.P = P;
  // columns of the level-1 count matrix: the blocks of a set's segments side by side; every row is as wide as the widest set
  uint32_t width[MAX_SETS] = {}, nblk_x = 0;
  static const uint32_t tile_max = [] { const char* e = std::getenv("ALEO_MI355X_PART_TILE_MAX"); const int k = e ? std::atoi(e) : 8192; return (uint32_t)(k == 2048 || k == 4096 || k == 8192 ? k : 8192); }();      // A/B switch
  uint32_t tile = pts >= ((size_t)1 << 22) ? 8192u : (pts >= ((size_t)1 << 21) ? 4096u : PART_TILE);
  segs.tile = tile < tile_max ? tile : tile_max;
  for (uint32_t q = 0; q < segs.nseg; ++q) {
    const uint32_t nb = (segs.n[q] + segs.tile - 1) / segs.tile, st = pre ? segs.set[q] : 0;
    segs.col0[q] = width[st]; width[st] += nb; nblk_x = nb > nblk_x ? nb : nblk_x;
  }
  uint32_t nblk = 1; for (uint32_t w : width) nblk = w > nblk ? w : nblk;
  segs.ncol = nblk;
  sp.digitsW = (SCALAR_BITS + P.c - 1) / P.c;
  const uint32_t M = sp.M = P.M, ntiles = (M + SCAN_TILE - 1) / SCAN_TILE;
  const size_t pairs_max = sp.pairs_max = pts * (size_t)sp.digitsW;
  if (pairs_max >= (1ull << 32)) { g_last_error = "msm: n * windows exceeds 2^32 (shard the MSM across GPUs)"; return ALEO_MI355X_ERR_BAD_ARG; }
  const size_t slices_max = sp.slices_max = slice_bound(pairs_max, M);
  sp.slice_blocks = (uint32_t)((slices_max + 255) / 256);
  int32_t rc;
  // hist | heavy list | meta | super list | level-2 cursors | the level-1 count matrix live in one allocation, zeroed by ONE fill
  const size_t hist_words = 3 * (size_t)M + 2048 + SUPER_CAP;
  if ((rc = c->scan_local.reserve((size_t)M * 8))) return rc;
  if ((rc = c->scan_blk.reserve(2 * (size_t)ntiles * 8 + 64))) return rc;
  if ((rc = c->sorted.reserve(pairs_max * 4))) return rc;
  const uint32_t LB = (P.c - 1) < 8 ? (P.c - 1) : 8, ncb = P.W * (P.B >> LB);      // coarse bins of all sets / windows
  const size_t cnt_len = (size_t)ncb * nblk;
  if (ncb > MAX_COARSE_ALL || cnt_len >= (1ull << 32)) { g_last_error = "msm: partition table too large"; return ALEO_MI355X_ERR_BAD_ARG; }
  const uint32_t cnt_tiles = (uint32_t)((cnt_len + SCAN_TILE - 1) / SCAN_TILE);
  if ((rc = c->hist.reserve((hist_words + cnt_len) * 4))) return rc;
  if ((rc = c->part_cnt.reserve((cnt_len + 2 * (size_t)cnt_tiles + 16 + MAX_COARSE_ALL) * 4))) return rc;     // off_local | tile_tot | off_blk | part_start
  if ((rc = c->part_items.reserve(pairs_max * 8))) return rc;
  if ((rc = c->task_g.reserve(2 * slices_max * 4))) return rc;     // task_g | order
  if ((rc = ensure_host_pinned(c, 64))) return rc;

  uint32_t* hist = sp.hist = c->hist.as<uint32_t>(); uint32_t* heavy = sp.heavy = hist + M; uint32_t* meta = sp.meta = heavy + M;     // heavy: <= M bucket ids
  sp.super_list = heavy + M + 2048;
  uint32_t* bin_cursor = hist + 2 * (size_t)M + 2048 + SUPER_CAP;
  uint2* scan_local = sp.scan_local = c->scan_local.as<uint2>();
  uint2* tile_tot = c->scan_blk.as<uint2>(); uint2* scan_blk = sp.scan_blk = tile_tot + ntiles;
  uint32_t* sorted = sp.sorted = c->sorted.as<uint32_t>(); uint32_t* task_g = sp.task_g = c->task_g.as<uint32_t>(); uint32_t* order = sp.order = task_g + slices_max;

  if (!lean) HIPCHK(hipEventRecord(c->ev[0], s));
  sp.zero_bytes = (hist_words + cnt_len) * 4;
  const bool cleared_ahead = c->hist_clean >= sp.zero_bytes && c->hist_clean_stream == s && c->hist_clean_ptr == (void*)hist;      // the previous chain on this stream left it clean (msm_back)
  c->hist_clean = 0;                                        // (about to be dirtied)
  if (!cleared_ahead) HIPCHK(hipMemsetAsync(hist, 0, sp.zero_bytes, s));          // (count matrix: columns no block of a set owns, and tiles past a segment's end, count zero)
  SortArgs sa;
  sa.segs = segs; sa.nblk_x = nblk_x ? nblk_x : 1;
  sa.inf = d_inf; sa.row_stride = row_stride;
  sa.cnt = hist + hist_words; sa.off_local = c->part_cnt.as<uint32_t>();
  uint32_t* cnt_tile_tot = sa.off_local + cnt_len; sa.off_blk = cnt_tile_tot + cnt_tiles;
  sa.items = c->part_items.as<uint2>();
  const uint32_t* total_pairs = sp.total_pairs = sa.off_blk + cnt_tiles;          // grand total of the level-1 scan
  uint32_t* part_start = sa.off_blk + cnt_tiles + 4;             // ncb + 1 prefix counts of the level-2 parts
  const uint32_t nparts_max = ncb + (uint32_t)(pairs_max / BIN_PART) + 1;
  if (mont) launch_sort<true>(P.c, pre, sa, 0, s); else launch_sort<false>(P.c, pre, sa, 0, s);
  hipLaunchKernelGGL(k_scan32_tiles, dim3(cnt_tiles), dim3(256), 0, s, sa.cnt, (uint32_t)cnt_len, sa.off_local, cnt_tile_tot);
  hipLaunchKernelGGL(k_scan32_top, dim3(1), dim3(256), 0, s, cnt_tile_tot, cnt_tiles, sa.off_blk, fill_shift());
  if (mont) launch_sort<true>(P.c, pre, sa, 1, s); else launch_sort<false>(P.c, pre, sa, 1, s);
  hipLaunchKernelGGL(k_bin_parts, dim3(1), dim3(256), 0, s, sa.off_local, sa.off_blk, nblk, ncb, cnt_tiles, part_start);
  hipLaunchKernelGGL(k_bin_hist, dim3(nparts_max), dim3(256), 0, s, sa.items, sa.off_local, sa.off_blk, nblk, ncb, cnt_tiles, LB, part_start, hist);
  hipLaunchKernelGGL(k_bin_scatter, dim3(nparts_max), dim3(256), 0, s, sa.items, sa.off_local, sa.off_blk, nblk, ncb, cnt_tiles, LB, part_start, hist, bin_cursor, sorted);
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(256), 0, s, hist, M, total_pairs, scan_local, tile_tot, meta, heavy);
  // The slice count, the longest bucket and the list lengths size the slice-tree launches.  k_scan_top stores them (and this call's sequence number behind them)
  // into the slot's pinned, device-mapped buffer; the slice kernels and the accumulation — launched with grids from slice_bound() — follow on `s` at once, and
  // the host polls the sequence number long before the accumulation ends (msm_wait_meta): the GPU never waits for the round trip, and nothing but kernels
  // sits on the stream (rounds 1-4 copied the words back on the side stream behind an event of `s`: a copy launch and ~6 us of idle GPU per chain).
  uint32_t* host_meta = nullptr;
  HIPCHK(hipHostGetDevicePointer((void**)&host_meta, c->h_pinned, 0));
  sp.meta_seq = ++c->meta_seq;
  static const bool fuse_top_env = [] { const char* e = std::getenv("ALEO_MI355X_FUSE_SCAN_TOP"); return !(e && e[0] == '0'); }();      // A/B switch
  const bool fuse_top = fuse_top_env && lean && ntiles <= FUSED_TILES;      // (calls that time their phases keep the sort / slice-order boundary at ev[1])
  if (!fuse_top) hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, s, tile_tot, ntiles, scan_blk, meta, (volatile uint32_t*)host_meta, sp.meta_seq);
  HIPCHK(hipGetLastError());
  if (!lean) HIPCHK(hipEventRecord(c->ev[1], s));
  uint32_t* len_count = meta + 16; uint32_t* len_cursor = len_count + MAX_SLICE + 1;   // zeroed with hist/meta
  if (fuse_top) hipLaunchKernelGGL(k_slice_count<true>, dim3(sp.slice_blocks), dim3(256), 0, s, hist, scan_local, (const uint2*)scan_blk, M, total_pairs, meta, task_g, len_count, (const uint2*)tile_tot, ntiles, scan_blk, (volatile uint32_t*)host_meta, sp.meta_seq);
  else hipLaunchKernelGGL(k_slice_count<false>, dim3(sp.slice_blocks), dim3(256), 0, s, hist, scan_local, (const uint2*)scan_blk, M, total_pairs, meta, task_g, len_count, (const uint2*)tile_tot, ntiles, scan_blk, (volatile uint32_t*)nullptr, 0u);
  hipLaunchKernelGGL(k_slice_order, dim3(sp.slice_blocks), dim3(256), 0, s, hist, scan_local, scan_blk, total_pairs, M, meta, task_g, len_count, len_cursor, order);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

int32_t msm_wait_meta(Ctx* c, const SortPhase& sp, hipStream_t s, SliceMeta* m) {
  const volatile uint32_t* h_meta = (const volatile uint32_t*)c->h_pinned;
  // poll the sequence number (it arrives ~0.15 ms after the sort was queued).  A stream that has drained or failed without delivering it is an error, not a hang.
  for (uint64_t spins = 0; h_meta[8] != sp.meta_seq; ++spins) {
    __builtin_ia32_pause();
    if ((spins & 0xfff) == 0xfff) {
      const hipError_t q = hipStreamQuery(s);
      if (q == hipErrorNotReady) continue;
      if (q == hipSuccess && h_meta[8] == sp.meta_seq) break;
      if (q == hipSuccess) { (void)hipStreamSynchronize(s); if (h_meta[8] == sp.meta_seq) break; }
      g_last_error = std::string("msm: the slice metadata never arrived (") + hipGetErrorString(q) + ")"; return ALEO_MI355X_ERR_HIP;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  m->NT = h_meta[0]; m->max_m = h_meta[1]; m->n_heavy = h_meta[3];
  m->max_common = h_meta[6] < 16u ? h_meta[6] : 16u;
  m->n_super = h_meta[5] < SUPER_CAP ? h_meta[5] : SUPER_CAP;
  m->super_overflow = h_meta[5] > SUPER_CAP;          // then the common list also holds very long buckets
  if (m->NT > sp.slices_max) {          // cannot happen (slice_bound); the kernels only touched threads below the bound
    (void)hipStreamSynchronize(s); g_last_error = "msm: internal slice count overflow"; return ALEO_MI355X_ERR_HIP;
  }
  return ALEO_MI355X_OK;
}

// One launch chain in four steps, so that several chains — the chunks of one request whose scalars are still arriving — can share buckets and one
// bucket reduction (msm_run_chunked below):
//   msm_front_sort    checks, plan, sort, slice ordering — everything queued, nothing waited for
//   msm_front_accum   the accumulation kernel (optionally behind an event, optionally seeded with the bucket sums of earlier chains)
//   msm_front_finish  the slice metadata arrives (side stream), the slice trees follow: bucket b's sum is then the first slice of b
//   msm_back          bucket reduction, host tail, phase times
namespace {
struct Front {
  const PinnedBases::PreTable* T = nullptr; MsmPlan P{}; SortPhase sp; SliceMeta sm;
  uint32_t K = 0, cpw = 0, nchunks = 0, lgN = 0, tseg = 0, fseg = 0, nseg = 0, setw = 0; size_t vpoints = 0;
  bool pre = false, masked = false, aside = false, empty = false, lean = false; const char* bases = nullptr;      // lean: MsmJob::lean of a single-chain request (no phase-timing events)
};
}

// Work queued on borrowed contexts must have finished before their locks are released, whatever way the function is left (an early HIPCHK return, an
// exception on its way to the C ABI's catch): the guard synchronises the listed streams in its destructor unless the normal path — which ends
// synchronised anyway — dismissed it.
namespace {
struct StreamDrainGuard {
  std::vector<hipStream_t> streams; bool armed = true;
  ~StreamDrainGuard() { if (!armed) return; const std::string keep = g_last_error; for (hipStream_t st : streams) if (st) (void)hipStreamSynchronize(st); g_last_error = keep; }
  void add(hipStream_t st) { streams.push_back(st); }
  void dismiss() { armed = false; }
};
}
static int32_t msm_front_sort(Ctx* c, const PinnedBases& pb, const MsmJob& job, hipStream_t s, Front& f) {
  const uint32_t K = f.K = job.k;
  size_t n = 0, pts = 0; SegArgs segs{};
  if (K > MAX_SETS || job.nseg > MAX_SEGS) { g_last_error = "msm: too many sets / segments in one call"; return ALEO_MI355X_ERR_BAD_ARG; }
  for (uint32_t q = 0; q < job.nseg; ++q) {
    const MsmSeg& g = job.segs[q];
    if (g.len == 0) continue;
    if (g.out >= K || g.off + g.len >= (1ull << 31)) { g_last_error = "msm: segment out of range"; return ALEO_MI355X_ERR_BAD_ARG; }
    const uint32_t i = segs.nseg++;
    segs.ptr[i] = (const char*)g.d_ptr; segs.n[i] = (uint32_t)g.len; segs.off[i] = (uint32_t)g.off; segs.set[i] = (uint8_t)g.out;
    n = g.off + g.len > n ? g.off + g.len : n; pts += g.len;
  }
  if (n == 0) { f.empty = true; return ALEO_MI355X_OK; }
  if (job.tier_n > n && job.tier_n <= pb.n) n = job.tier_n;
  if (n > pb.n) { g_last_error = "msm: a segment reaches past the pinned bases"; return ALEO_MI355X_ERR_BAD_ARG; }
  // the fixed-base table serves any prefix of the pinned set (row stride = pinned count) as long as the prefix still
  // puts about one point into every bucket; shorter prefixes use the plain path with its small bucket count
  const PinnedBases::PreTable* T = nullptr;
  const uint8_t* d_inf = pb.d_inf;
  bool ranged = job.sparse && pb.range.d != nullptr;       // the narrow-window table of one sub-range, when every segment lies inside it
  for (uint32_t i = 0; ranged && i < segs.nseg; ++i) ranged = segs.off[i] >= pb.range_off && (size_t)segs.off[i] + segs.n[i] <= pb.range_off + pb.range.cover;
  if (ranged) {
    T = &pb.range; n = 0;
    for (uint32_t i = 0; i < segs.nseg; ++i) { segs.off[i] -= (uint32_t)pb.range_off; n = (size_t)segs.off[i] + segs.n[i] > n ? (size_t)segs.off[i] + segs.n[i] : n; }
    if (d_inf) d_inf += pb.range_off;
  } else for (const auto& t : pb.tab) if (t.d && n >= t.min_n && n <= t.cover) { T = &t; break; }
  const bool pre = f.pre = T != nullptr; f.T = T;
  const uint32_t set_cap = ranged ? MAX_COARSE_ALL / ((1u << (pb.range.c - 1)) >> 8) : msm_max_sets(pb, n);
  if (K > 1 && (!pre || K > (set_cap < MAX_SETS ? set_cap : MAX_SETS))) { g_last_error = "msm: internal: batch without a table tier (or too many sets)"; return ALEO_MI355X_ERR_BAD_ARG; }
  MsmPlan& P = f.P; P = make_plan(pre ? n : pts, pre ? T->c : 0);
  if (pre) { P.W = K; P.M = K * P.B; }                       // after the sort a set is "a window with its own buckets"
  if (!pre && !pb.d_xy28) { g_last_error = "msm: pinned set without 28-bit rows"; return ALEO_MI355X_ERR_HIP; }
  f.bases = (const char*)(pre ? T->d : pb.d_xy28);          // 112-byte rows either way
  const uint32_t cpw = f.cpw = P.B / P.S, nchunks = f.nchunks = cpw * P.W;
  uint32_t lgN = 0; while ((1u << lgN) < cpw) ++lgN;
  f.lgN = lgN;
  const bool masked = f.masked = pre && lgN >= 2 && (1u << lgN) == cpw;          // fixed-base path: weights by masked trees
  if (pre && !masked) { g_last_error = "msm: internal: table path without masked reduction"; return ALEO_MI355X_ERR_HIP; }
  int32_t rc;
  // table path, per set: [acc of its cpw chunks | lgN masked sums of cpw/4] = (lgN + 4) segments of tseg points
  const uint32_t tseg = f.tseg = cpw / 4, fseg = f.fseg = lgN + 4, nseg = f.nseg = K * fseg, setw = f.setw = fseg * tseg;
  f.vpoints = masked ? (size_t)K * setw + nchunks + (nseg + 1) + (size_t)nseg * (tseg / 2 + tseg / 4 + 2) + 3 * (size_t)nchunks + 64 : (size_t)nchunks + P.W;      // + the two buffers of the sum-tree passes (3 cpw / 2 points per set each)
  if ((rc = ensure_host_pinned(c, 64 + (size_t)(masked ? nseg : P.W) * 224 + ASIDE_MAX * 228))) return rc;      // before the sort phase: its read-back lands in this buffer
  SortPhase& sp = f.sp;
  if ((rc = msm_sort_phase(c, segs, pts, job.mont, d_inf, (uint32_t)(pre ? T->cover : pb.n), P, pre, s, &sp, f.lean))) return rc;
  const uint32_t M = sp.M;
  if ((rc = c->partial.reserve(sp.slices_max * (pre ? 224 : 192)))) return rc;
  if ((rc = c->vbuf.reserve(f.vpoints * 224))) return rc;
  return ALEO_MI355X_OK;
}

// seed: bucket sums of the request's earlier chunks (table path only), complete once `after` has been reached
static int32_t msm_front_accum(Ctx* c, hipStream_t s, Front& f, const FrontChain* seed, hipEvent_t after) {
  const SortPhase& sp = f.sp; const uint32_t M = sp.M; const char* bases = f.bases;
  char* partial = c->partial.as<char>();
  if (after) HIPCHK(hipStreamWaitEvent(s, after, 0));
  if (!f.lean) HIPCHK(hipEventRecord(c->ev[6], s));          // ev[6]..ev[5] bracket k_accum28 alone (bench.py's roofline kernel)
  if (f.pre && seed && seed->n) hipLaunchKernelGGL((k_accum28<true, true>), dim3(sp.slice_blocks), dim3(256), 0, s, bases, sp.sorted, sp.hist, sp.scan_local, sp.scan_blk, sp.total_pairs, M, sp.meta, sp.order, sp.task_g, partial, *seed);
  else if (f.pre) hipLaunchKernelGGL(k_accum28<true>, dim3(sp.slice_blocks), dim3(256), 0, s, bases, sp.sorted, sp.hist, sp.scan_local, sp.scan_blk, sp.total_pairs, M, sp.meta, sp.order, sp.task_g, partial, FrontChain{});
  else hipLaunchKernelGGL(k_accum28<false>, dim3(sp.slice_blocks), dim3(256), 0, s, bases, sp.sorted, sp.hist, sp.scan_local, sp.scan_blk, sp.total_pairs, M, sp.meta, sp.order, sp.task_g, partial, FrontChain{});
  if (!f.lean) HIPCHK(hipEventRecord(c->ev[5], s));
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

static int32_t msm_front_finish(Ctx* c, hipStream_t s, Front& f, bool allow_aside) {
  int32_t rc;
  const SortPhase& sp = f.sp; SliceMeta& sm = f.sm; const uint32_t M = sp.M; const bool pre = f.pre;
  uint32_t* heavy = sp.heavy; uint32_t* meta = sp.meta; uint2* scan_local = sp.scan_local; uint2* scan_blk = sp.scan_blk;
  char* partial = c->partial.as<char>();
  if ((rc = msm_wait_meta(c, sp, s, &sm))) return rc;
  // A handful of super-heavy buckets (witness-like scalars: the digit-1 bucket of the lowest window holds a fifth of the points) have a slice tree of
  // 10+ dependent levels while the common list is done after 4.  Then the long trees run ASIDE, on the slot's side stream, over slices 1.. of their
  // buckets; the reduction below goes ahead with slice 0 as those buckets' sums, and the host adds (b + 1) * (sum of slices 1..) to the result.
  const bool aside = f.aside = allow_aside && f.masked && aside_on() && sm.n_super >= 1 && sm.n_super <= ASIDE_MAX && !sm.super_overflow && sm.max_m >= 64;
  uint32_t* h_aside = (uint32_t*)((char*)c->h_pinned + 64 + (size_t)f.nseg * 224);
  if (aside) {
    if (f.lean) HIPCHK(hipEventRecord(c->ev[5], s));          // (nothing has been queued behind the accumulation yet: the same position)
    HIPCHK(hipStreamWaitEvent(c->side, c->ev[5], 0));
    for (uint32_t pass = 0, L = sm.max_m - 1; L > 1; ++pass, L = (L + 1) >> 1) {
      const uint64_t ops = (uint64_t)sm.n_super * (L >> 1);
      if (grp_lanes(ops) == 4) hipLaunchKernelGGL((k_tree_pass<true, 4>), dim3((uint32_t)((4 * ops + 255) / 256)), dim3(256), 0, c->side, partial, heavy, 0u, 0u, sp.super_list, sm.n_super, L >> 1, scan_local, scan_blk, M, meta, pass, 1u);
      else hipLaunchKernelGGL(k_tree_pass<true>, dim3((uint32_t)((2 * ops + 255) / 256)), dim3(256), 0, c->side, partial, heavy, 0u, 0u, sp.super_list, sm.n_super, L >> 1, scan_local, scan_blk, M, meta, pass, 1u);
    }
    uint32_t* dst = nullptr;
    HIPCHK(hipHostGetDevicePointer((void**)&dst, h_aside, 0));
    hipLaunchKernelGGL(k_gather_super, dim3((sm.n_super * 57 + 255) / 256), dim3(256), 0, c->side, partial, sp.super_list, sm.n_super, scan_local, scan_blk, dst);
    HIPCHK(hipEventRecord(c->ev[4], c->side));
  }
  // the levels behind the first in one launch (k_tree_rest) when only the common list is left and no bucket has more than 8 slices (ALEO_MI355X_TREE_REST=0: one launch per level)
  static const bool tree_rest = [] { const char* e = std::getenv("ALEO_MI355X_TREE_REST"); return !(e && e[0] == '0'); }();
  const bool rest_ok = tree_rest && pre && quads_on() && !sm.super_overflow && (aside || sm.n_super == 0) && sm.n_heavy && sm.max_common > 2 && sm.max_common <= 8;
  if (sm.NT) {
    for (uint32_t pass = 0, L = sm.max_m, Lcm = sm.max_common; L > 1; ++pass, L = (L + 1) >> 1, Lcm = (Lcm + 1) >> 1) {
      if (rest_ok && pass == 1) {
        hipLaunchKernelGGL(k_tree_rest<4>, dim3((uint32_t)((4ull * sm.n_heavy + 255) / 256)), dim3(256), 0, s, partial, heavy, sm.n_heavy, scan_local, scan_blk, M, meta);
        break;
      }
      const uint32_t Lc = sm.super_overflow ? L : Lcm;       // longest bucket of the common list at this level
      const uint32_t len_a = (sm.n_heavy && Lc > 1) ? sm.n_heavy : 0, pairs_a = len_a ? Lc >> 1 : 0;
      const uint32_t len_b = aside ? 0 : sm.n_super, pairs_b = len_b ? L >> 1 : 0;
      const uint64_t ops = (uint64_t)len_a * pairs_a + (uint64_t)len_b * pairs_b;
      const uint32_t lanes = pre ? grp_lanes(ops) : 2u;                                                // quads while the level is latency-bound (28-bit points only)
      const uint64_t threads = lanes * ops;
      if (!threads) continue;
      if (threads >= (1ull << 32)) { (void)hipStreamSynchronize(s); g_last_error = "msm: slice tree too wide"; return ALEO_MI355X_ERR_HIP; }
      if (pre && lanes == 4) hipLaunchKernelGGL((k_tree_pass<true, 4>), dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, partial, heavy, len_a, pairs_a, sp.super_list, len_b, pairs_b, scan_local, scan_blk, M, meta, pass, 0u);
      else if (pre) hipLaunchKernelGGL(k_tree_pass<true>, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, partial, heavy, len_a, pairs_a, sp.super_list, len_b, pairs_b, scan_local, scan_blk, M, meta, pass, 0u);
      else hipLaunchKernelGGL(k_tree_pass<false>, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, partial, heavy, len_a, pairs_a, sp.super_list, len_b, pairs_b, scan_local, scan_blk, M, meta, pass, 0u);
    }
  }
  if (!f.lean) HIPCHK(hipEventRecord(c->ev[2], s));
  return ALEO_MI355X_OK;
}

// older: the earlier chunks of the same request (msm_run_chunked) — a bucket this chain did not touch keeps its sum there.
// Two halves: `collect` false = queue the bucket reduction (table path) and return with ev[3] recorded behind it; `enqueued` true = that has been done by
// an earlier call, only wait for the result and run the host tail (run_chains queues other chains' work in between).
static int32_t msm_back(Ctx* c, uint64_t* out_jac18, Front& f, hipStream_t s, bool fire_tail, const FrontChain& older, bool collect = true, bool enqueued = false) {
  using namespace host;
  const MsmPlan& P = f.P; const SortPhase& sp = f.sp; const SliceMeta& sm = f.sm;
  const uint32_t K = f.K, cpw = f.cpw, nchunks = f.nchunks, lgN = f.lgN, tseg = f.tseg, fseg = f.fseg, nseg = f.nseg, setw = f.setw;
  const bool masked = f.masked, aside = f.aside;
  uint32_t* hist = sp.hist; uint2* scan_local = sp.scan_local; uint2* scan_blk = sp.scan_blk;
  char* partial = c->partial.as<char>();
  char* V = c->vbuf.as<char>(); char* Vout = V + (size_t)nchunks * 192;
  uint32_t* h_aside = (uint32_t*)((char*)c->h_pinned + 64 + (size_t)nseg * 224);
  std::chrono::steady_clock::time_point t_host0;
  char* h_win = (char*)c->h_pinned + 64;
  auto lazy_point = [&](const char* p) {
    const uint64_t* src = (const uint64_t*)p;
    HXYZZ v; v.X = HFq::reduce_lazy(src); v.Y = HFq::reduce_lazy(src + 6); v.ZZ = HFq::reduce_lazy(src + 12); v.ZZZ = HFq::reduce_lazy(src + 18);
    return v;
  };
  // a 224-byte point of the table path: every coordinate is value * 2^392 mod q (+ a few q) as 14 x 28-bit limbs (possibly
  // loose); * 2^376 under the 2^-384 of the host Montgomery product gives value * 2^384, the HFq form
  auto lazy_point28 = [&](const char* p) {
    const uint32_t* w = (const uint32_t*)p;
    HFq c376 = HFq::zero(); c376.l[5] = 1ull << 56;
    HFq co[4]; bool inf = true;
    for (int k = 0; k < 4; ++k) {
      uint64_t big[8] = {0, 0, 0, 0, 0, 0, 0, 0};           // sum_i w_i * 2^(28 i), limbs may exceed 28 bits
      for (int i = 0; i < 14; ++i) {
        const int pos = 28 * i, j = pos >> 6, sh = pos & 63;
        const unsigned __int128 add = (unsigned __int128)w[14 * k + i] << sh;
        unsigned __int128 t = (unsigned __int128)big[j] + (uint64_t)add; big[j] = (uint64_t)t;
        t = (unsigned __int128)big[j + 1] + (uint64_t)(add >> 64) + (uint64_t)(t >> 64); big[j + 1] = (uint64_t)t;
        uint64_t cr = (uint64_t)(t >> 64);
        for (int q = j + 2; q < 8 && cr; ++q) { t = (unsigned __int128)big[q] + cr; big[q] = (uint64_t)t; cr = (uint64_t)(t >> 64); }
      }
      if (k == 2) for (int q = 0; q < 8; ++q) if (big[q]) inf = false;
      co[k] = HFq::mul(HFq::reduce_lazy(big), c376);        // value < 64q < 2^384: six limbs hold it
    }
    if (inf) return HXYZZ::infinity();
    HXYZZ v; v.X = co[0]; v.Y = co[1]; v.ZZ = co[2]; v.ZZZ = co[3];
    return v;
  };
  if (masked) {
    // per set: sum_b (b+1) S_b = sum_j acc_j + S * sum_j j * run_j ; the second sum by lg(N) masked pairwise trees
    char* Vrun = V + (size_t)K * setw * PB28; char* Tout = Vrun + (size_t)nchunks * PB28;
    // 2^19 buckets keep the chip busy with one lane pair per chunk; the small bucket sets (<= 2^16) are pure latency and take the quad form
    // (masked => pre: the partial sums are 28-bit points)
    // every launch below picks lanes per addition by its own width (grp_lanes): four while it is latency-bound, two once the additions fill the chip
    // wide tables (2^19 buckets, S = 16): one lane pair per chunk (32 dependent additions) against two pairs one step apart (17): reduce phase 0.487 -> 0.460 ms
    // at 2^20; S = 8 / 32 / 4 with either form: 0.51-0.54 / 0.48-0.55 / 0.66-0.72 ms (ALEO_MI355X_CHUNK_S, ALEO_MI355X_CHUNK_FORM=1: A/B switches)
    static const uint32_t prog_min_c = [] { const char* e = std::getenv("ALEO_MI355X_SUM_TREE_MIN_C"); const int k = e ? std::atoi(e) : 13; return (uint32_t)(k >= 13 && k <= 24 ? k : 13); }();
    const bool prog = P.c >= prog_min_c && prog_on() && cpw > FOLD;      // (a set of <= 256 chunks would go straight to the final fold: the masked form keeps those)
    uint32_t out_pts = fseg;                               // result points per set the host tail reads
    if (!enqueued) {
    static const bool wide_two_groups = [] { const char* e = std::getenv("ALEO_MI355X_CHUNK_FORM"); return !(e && e[0] == '1'); }();
    if (P.c >= 20 && !wide_two_groups) {
      if (grp_lanes(2 * (uint64_t)nchunks) == 4) hipLaunchKernelGGL((k_bucket_chunks_pair<true, 4>), dim3((nchunks + 63) / 64), dim3(256), 0, s, partial, hist, scan_local, scan_blk, P.B, P.S, nchunks, V, setw, Vrun, older);
      else hipLaunchKernelGGL(k_bucket_chunks_pair<true>, dim3((nchunks + CHUNK_PAIRS - 1) / CHUNK_PAIRS), dim3(256), 0, s, partial, hist, scan_local, scan_blk, P.B, P.S, nchunks, V, setw, Vrun, older);
    } else {
      if (grp_lanes(2 * (uint64_t)nchunks) == 4) hipLaunchKernelGGL((k_bucket_chunks<true, 4>), dim3((nchunks + 31) / 32), dim3(256), 0, s, partial, hist, scan_local, scan_blk, P.B, P.S, nchunks, V, setw, Vrun, older);
      else hipLaunchKernelGGL(k_bucket_chunks<true>, dim3((nchunks + CHUNK_QUADS - 1) / CHUNK_QUADS), dim3(256), 0, s, partial, hist, scan_local, scan_blk, P.B, P.S, nchunks, V, setw, Vrun, older);
    }
    // Measured (ALEO_MI355X_SUM_TREE=0 is the A/B switch; ALEO_MI355X_SUM_TREE_MIN_C limits it to the wider tables): reduce phase of the 2^20 MSM 0.464 -> 0.412 ms at
    // S = 16 (S = 8: 0.537 -> 0.451, S = 4: 0.720 -> 0.508: the chunk kernel is bound by its 2 additions per bucket, not by their order, so smaller chunks still lose);
    // on the small tables too: 2^15-constraint proof 6.47 -> 6.35 ms, eight instances at 2^13 7.19 -> 6.88 ms.
    if (prog) {
      char* G0 = Tout; char* G1 = G0 + (size_t)K * (3 * (cpw / 2)) * PB28;      // ping-pong: a set is at most 3 segments of cpw / 2 points after the first pass
      const char* node = Vrun; uint32_t node_ss = cpw; const char* A = V; uint32_t a_ss = setw; const char* T = Vrun; uint32_t t_ss = 0, nT = 0, L = cpw;
      char* dstbuf = G0;
      // ALEO_MI355X_PROG_PASS2=0: one level per launch throughout (A/B switch)
      static const bool pass2_on = [] { const char* e = std::getenv("ALEO_MI355X_PROG_PASS2"); return !(e && e[0] == '0'); }();
      while (L > FOLD) {
        const uint32_t half = L >> 1, out_ss = (3 + nT) * half; const uint64_t ops = (uint64_t)(2 + nT) * half * K;
        if (pass2_on && (L >> 2) >= FOLD && grp_lanes(ops) == 4) {      // two levels at once: L -> L / 4
          const uint32_t quarter = L >> 2, out2 = (4 + nT) * quarter; const uint64_t octs = (uint64_t)(2 + nT) * quarter * K;
          hipLaunchKernelGGL(k_prog_pass2, dim3((uint32_t)((octs + 31) / 32)), dim3(256), 0, s, node, node_ss, A, a_ss, T, t_ss, nT, L, K, dstbuf, out2);
          node = dstbuf; node_ss = out2; A = dstbuf + (size_t)quarter * PB28; a_ss = out2; T = dstbuf + (size_t)2 * quarter * PB28; t_ss = out2; nT += 2; L = quarter;
          dstbuf = dstbuf == G0 ? G1 : G0;
          continue;
        }
        if (grp_lanes(ops) == 4) hipLaunchKernelGGL(k_prog_pass<4>, dim3((uint32_t)((4 * ops + 255) / 256)), dim3(256), 0, s, node, node_ss, A, a_ss, T, t_ss, nT, L, K, dstbuf, out_ss);
        else hipLaunchKernelGGL(k_prog_pass<2>, dim3((uint32_t)((2 * ops + 255) / 256)), dim3(256), 0, s, node, node_ss, A, a_ss, T, t_ss, nT, L, K, dstbuf, out_ss);
        node = dstbuf; node_ss = out_ss; A = dstbuf + (size_t)half * PB28; a_ss = out_ss; T = dstbuf + (size_t)2 * half * PB28; t_ss = out_ss; ++nT; L = half;
        dstbuf = dstbuf == G0 ? G1 : G0;
      }
      uint32_t lgL = 0; while ((1u << lgL) < L) ++lgL;
      const char* fin = node; const uint32_t fin_ss = node_ss;
      out_pts = 1 + nT + lgL;                              // = 1 + lgN
      char* dst = nullptr;
      HIPCHK(hipHostGetDevicePointer((void**)&dst, h_win, 0));
      if (quads_on()) hipLaunchKernelGGL(k_prog_final<4>, dim3(K * out_pts), dim3(512), 0, s, fin, fin_ss, L, nT, lgL, K, dst);
      else hipLaunchKernelGGL(k_prog_final<2>, dim3(K * out_pts), dim3(256), 0, s, fin, fin_ss, L, nT, lgL, K, dst);
    } else {
    {
      const uint64_t ops = (uint64_t)tseg * lgN * K;
      if (grp_lanes(ops) == 4) hipLaunchKernelGGL(k_masked_pairs<4>, dim3((uint32_t)((4 * ops + 255) / 256)), dim3(256), 0, s, Vrun, lgN, K, V, setw);
      else hipLaunchKernelGGL(k_masked_pairs<2>, dim3((uint32_t)((2 * ops + 255) / 256)), dim3(256), 0, s, Vrun, lgN, K, V, setw);
    }
    // K * (lgN+4) segment sums: pairwise launches while a level still fills the chip, then ONE block per segment folds the
    // last 256 points through LDS (8 levels of lane-pair additions: the latency floor of the chain, no launch gaps)
    char* F1 = Tout + (size_t)(nseg + 1) * PB28; char* F2 = F1 + (size_t)nseg * (tseg / 2 + 1) * PB28;
    const char* cur = V; uint32_t L = tseg, stride = tseg;
    while (L > FOLD) {
      char* dst = (cur == F1) ? F2 : F1; uint32_t half = (L + 1) >> 1;
      const uint64_t ops = (uint64_t)half * nseg;
      if (grp_lanes(ops) == 4) hipLaunchKernelGGL(k_seg_pair_pass<4>, dim3((uint32_t)((4 * ops + 255) / 256)), dim3(256), 0, s, cur, stride, L, nseg, dst, half);
      else hipLaunchKernelGGL(k_seg_pair_pass<2>, dim3((uint32_t)((2 * ops + 255) / 256)), dim3(256), 0, s, cur, stride, L, nseg, dst, half);
      cur = dst; stride = half; L = half;
    }
    if (L > 1) {
      // the last fold leaves one point per segment, contiguous: it stores them straight into the slot's pinned host buffer (device-
      // mapped), so the result needs neither a gather launch nor a copy — the stream synchronisation below is all that is left
      char* dst = nullptr;
      HIPCHK(hipHostGetDevicePointer((void**)&dst, h_win, 0));
      if (quads_on()) hipLaunchKernelGGL(k_seg_fold<4>, dim3(nseg), dim3(512), 0, s, cur, stride, L, nseg, dst, 1u);
      else hipLaunchKernelGGL(k_seg_fold<2>, dim3(nseg), dim3(256), 0, s, cur, stride, L, nseg, dst, 1u);
    } else {
      hipLaunchKernelGGL(k_gather_strided, dim3((nseg * 14 + 255) / 256), dim3(256), 0, s, cur, stride, nseg, Tout);
      HIPCHK(hipMemcpyAsync(h_win, Tout, (size_t)nseg * PB28, hipMemcpyDeviceToHost, s));
    }
    }
    HIPCHK(hipEventRecord(c->ev[3], s));
    if (f.lean && older.n == 0 && !aside) {                  // single-chain prover commitments: clear the sort's block for the next chain now, under the host tail (ev[3] sits in front of it; not with trees still running aside: they read the lists in that block)
      static const bool clear_ahead = [] { const char* e = std::getenv("ALEO_MI355X_CLEAR_AHEAD"); return !(e && e[0] == '0'); }();      // A/B switch
      // (as much as the largest chain seen on this context needs, within the allocation: a proof's chains differ in size, and a chain larger than its predecessor would fill again)
      if (clear_ahead) {
        if (sp.zero_bytes > c->hist_zero_max) c->hist_zero_max = sp.zero_bytes;
        const size_t z = c->hist_zero_max <= c->hist.cap && (void*)sp.hist == c->hist.p ? c->hist_zero_max : sp.zero_bytes;
        HIPCHK(hipMemsetAsync(sp.hist, 0, z, s)); c->hist_clean = z; c->hist_clean_stream = s; c->hist_clean_ptr = (void*)sp.hist;
      }
    }
    } else if (prog) { uint32_t L = cpw, nT = 0; while (L > FOLD) { ++nT; L >>= 1; } uint32_t lgL = 0; while ((1u << lgL) < L) ++lgL; out_pts = 1 + nT + lgL; }
    if (!collect) return ALEO_MI355X_OK;
    if (fire_tail && c->tail_hook) {
      // the caller's next kernels go behind the fold; the host waits for the fold only (its result sits in pinned memory) and does the tail below while they run
      std::function<int32_t()> hook = std::move(c->tail_hook); c->tail_hook = nullptr;
      HT("msm: reduction queued");
      const int32_t hrc = hook();
      HT("msm: hook queued");
      HIPCHK(hipEventSynchronize(c->ev[3]));
      if (hrc) { (void)hipStreamSynchronize(s); return hrc; }
    } else if (enqueued) HIPCHK(hipEventSynchronize(c->ev[3]));      // (other chains' work may already be queued behind it on other streams; this chain's is all in front of ev[3])
    else HIPCHK(hipStreamSynchronize(s));
    if (aside) HIPCHK(hipEventSynchronize(c->ev[4]));
    HIPCHK(hipGetLastError());
    HT("msm: result arrived");
    t_host0 = std::chrono::steady_clock::now();
    HXYZZ totals[MAX_SETS];
    // one Horner chain per result (~15 us): from 3 results on they are spread over the library's parked helper threads (a lockstep round has 8 x k of them)
    auto horner = [&](size_t q) {
      const char* hw = h_win + q * out_pts * PB28;
      const uint32_t na = prog ? 1u : 4u;                  // points that hold sum_j acc_j: one (k_prog_final) or the four segment sums of the masked form
      HXYZZ total = HXYZZ::infinity();
      for (int l = (int)lgN - 1; l >= 0; --l) { total = hdouble(total); total = hadd(total, lazy_point28(hw + (size_t)(na + l) * PB28)); }
      for (uint32_t sft = P.S; sft > 1; sft >>= 1) total = hdouble(total);
      for (uint32_t r = 0; r < na; ++r) total = hadd(total, lazy_point28(hw + (size_t)r * PB28));
      totals[q] = total;
    };
    static const uint32_t tail_pool_min = [] { const char* e = std::getenv("ALEO_MI355X_TAIL_POOL"); const int k = e ? std::atoi(e) : 3; return (uint32_t)(k <= 0 ? 0x7fffffff : k); }();      // A/B switch: results from which the pool is used (0 = never; 2^15 proof: 5.68 / 5.65 / 5.74 ms with 8 / 3 / 2, profiles/r05_tailpool_min_ab.txt)
    if (K >= tail_pool_min) host_parallel_for(K, horner); else for (uint32_t q = 0; q < K; ++q) horner(q);
    for (uint32_t h = 0; aside && h < sm.n_super; ++h) {             // (b + 1) * (slices 1.. of super-heavy bucket b), by double-and-add
      const uint32_t* rec = h_aside + (size_t)h * 57; const uint32_t g = rec[56], q = g / P.B, wgt = g % P.B + 1;
      if (q >= K) { g_last_error = "msm: internal: super-heavy bucket outside the sets"; return ALEO_MI355X_ERR_HIP; }
      const HXYZZ T = lazy_point28((const char*)rec); HXYZZ acc = HXYZZ::infinity();
      for (int bit = 31 - __builtin_clz(wgt); bit >= 0; --bit) { acc = hdouble(acc); if ((wgt >> bit) & 1u) acc = hadd(acc, T); }
      totals[q] = hadd(totals[q], acc);
    }
    HT("msm: horner done");
    hstore_jacobian_normalized_batch(out_jac18, totals, K);          // one shared inversion for the K results
    HT("msm: normalised");
  } else {
    HXYZZ total = HXYZZ::infinity();
    hipLaunchKernelGGL(k_bucket_chunks_plain, dim3((nchunks + 255) / 256), dim3(256), 0, s, partial, hist, scan_local, scan_blk, P.B, P.S, nchunks, V);
    for (uint32_t L = cpw; L > 1; L = (L + 1) >> 1) {
      uint32_t pairs = (L - ((L + 1) >> 1)) * P.W;
      hipLaunchKernelGGL(k_seg_tree_pass, dim3((2 * pairs + 255) / 256), dim3(256), 0, s, V, cpw, P.W, L);
    }
    hipLaunchKernelGGL(k_gather_windows, dim3((P.W * 12 + 255) / 256), dim3(256), 0, s, V, cpw, P.W, Vout);
    HIPCHK(hipMemcpyAsync(h_win, Vout, (size_t)P.W * 192, hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(c->ev[3], s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    t_host0 = std::chrono::steady_clock::now();
    // host tail: total = sum_w 2^(c*w) * S_w  (Horner from the top window), then affine normalisation
    for (int w = (int)P.W - 1; w >= 0; --w) {
      for (int d = 0; d < win_width((int)P.c, w); ++d) total = hdouble(total);          // window w spans win_width bits (balanced windows)
      total = hadd(total, lazy_point(h_win + (size_t)w * 192));
    }
    hstore_jacobian_normalized(out_jac18, total);
  }
  // phase times: HIP events on the launch stream up to the result's arrival (ev[3], already complete: the stream was synchronised),
  // the host tail by the host clock — no further event round trip on the critical path of a small MSM
  const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
  float ms;
  MsmTiming tm;
  if (f.lean) { tm.host = host_ms; tm.total = host_ms; c->last_msm = tm; g_last_msm = tm; return ALEO_MI355X_OK; }      // no phase events were recorded
  HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); tm.sort = ms;
  HIPCHK(hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); tm.accum = ms;
  HIPCHK(hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); tm.reduce = ms;
  HIPCHK(hipEventElapsedTime(&ms, c->ev[6], c->ev[5])); tm.accum_kernel = ms;
  tm.host = host_ms; tm.total = tm.sort + tm.accum + tm.reduce + tm.host;
  c->last_msm = tm; g_last_msm = tm;
  return ALEO_MI355X_OK;
}

static int32_t msm_run_chunked(Ctx* c, HelperSet& hs, uint64_t* out_jac18, const PinnedBases& pb, size_t n, bool mont, hipStream_t s, const void* host_src, const MsmJob* dev_job = nullptr);
static uint32_t chunks3_min_lg();
// Requests with the scalars already on the device CAN go in chunks too (msm_run_chunked with dev_job): the sort of the later chunks then runs beside the
// accumulation of the earlier ones instead of in front of everything.  Measured and OFF by default (round 4, resident uniform scalars, whole / chunked):
// 2^20 2.82 / 3.03 ms, 2^21 5.12 / 5.24, 2^22 9.41 / 9.51 — without an upload to hide, the seeded launches (+5 % accumulation time: shorter slices, a seed
// read and a product per bucket) and the sort that crawls beside an accumulation holding every wave slot cost more than the 0.3-1.0 ms of sort they
// move out of the way; holding a later chunk's sort until the previous accumulation starts (as run_chains does for whole chains) makes it worse
// (2^20 3.07, 2^21 5.91, 2^22 9.83 ms against 2.83 / 5.04 / 9.32 whole: tools/resident_ab.py).  ALEO_MI355X_CHUNK_DEV_MIN_LG: lg of the smallest such
// request (default 0 = never; the GPU suite passes with 20).
// the cut of a device-scalar request: segment of `len` scalars up to `pc` percent, on a multiple of 256 scalars (msm_run_chunked)
static inline size_t chunk_cut(size_t len, uint32_t pc) { const size_t v = pc >= 100u ? len : ((size_t)((double)len * pc / 100.0)) & ~(size_t)255; return v < len ? v : len; }
static const uint32_t CHUNK_SHARE_DEV[4][3] = {{0, 0, 0}, {0, 0, 0}, {28, 72, 0}, {12, 28, 60}};
// every chunk of a Q-chunk device-scalar request must hold at least one scalar (a request of many short segments can leave the first chunk empty: then the
// whole-request chain runs instead)
static bool chunks_all_nonempty(const MsmJob& job, uint32_t Q) {
  uint32_t cum = 0;
  for (uint32_t k = 0; k < Q; ++k) {
    const uint32_t lo = cum, hi = k + 1 == Q ? 100u : cum + CHUNK_SHARE_DEV[Q][k]; cum = hi; bool any = false;
    for (uint32_t q = 0; q < job.nseg && !any; ++q) any = chunk_cut(job.segs[q].len, hi) > chunk_cut(job.segs[q].len, lo);
    if (!any) return false;
  }
  return true;
}
static uint32_t chunk_dev_min_lg() { static const uint32_t v = [] { const char* e = std::getenv("ALEO_MI355X_CHUNK_DEV_MIN_LG"); const int k = e ? std::atoi(e) : 0; return (uint32_t)(k >= 0 && k <= 40 ? k : 0); }(); return v; }
int32_t msm_run(Ctx* c, uint64_t* out_jac18, const PinnedBases& pb, const MsmJob& job, hipStream_t s) {
  if (job.k == 0) return ALEO_MI355X_OK;
  if (chunk_dev_min_lg() && !job.sparse && c->dev && job.k <= MAX_SETS && job.nseg <= MAX_SEGS) {
    size_t pts = 0, reach = 0; bool in_range = true;
    for (uint32_t q = 0; q < job.nseg; ++q) { const MsmSeg& g = job.segs[q]; if (!g.len) continue; pts += g.len; reach = g.off + g.len > reach ? g.off + g.len : reach; in_range = in_range && g.out < job.k; }
    if (job.tier_n > reach && job.tier_n <= pb.n) reach = job.tier_n;
    bool tiered = false; for (const auto& t : pb.tab) if (t.d && reach >= t.min_n && reach <= t.cover) { tiered = job.k <= 1 || job.k <= msm_max_sets(pb, reach); break; }
    if (in_range && tiered && pts >= ((size_t)1 << chunk_dev_min_lg())) {
      HelperSet hs; { const int32_t rc = acquire_helpers(c->dev, pts >= ((size_t)1 << chunks3_min_lg()) ? 2 : 1, hs); if (rc) return rc; }
      if (!hs.ctx.empty() && chunks_all_nonempty(job, 1 + (uint32_t)hs.ctx.size())) return msm_run_chunked(c, hs, out_jac18, pb, reach, job.mont, s, nullptr, &job);
    }
  }
  Front f; int32_t rc;
  static const bool lean_off = [] { const char* e = std::getenv("ALEO_MI355X_LEAN"); return e && e[0] == '0'; }();      // A/B switch: 0 = phase-timing events in the prover's chains too (round 4)
  f.lean = job.lean && !lean_off;
  if ((rc = msm_front_sort(c, pb, job, s, f))) return rc;
  if (f.empty) { for (uint32_t q = 0; q < job.k; ++q) host::hstore_jacobian_normalized(out_jac18 + 18 * q, host::HXYZZ::infinity()); return ALEO_MI355X_OK; }
  if ((rc = msm_front_accum(c, s, f, nullptr, nullptr))) return rc;
  if ((rc = msm_front_finish(c, s, f, true))) return rc;
  return msm_back(c, out_jac18, f, s, job.fire_tail, FrontChain{});
}

// ONE result from HOST scalars, uploaded and processed in Q chunks that share the buckets and one bucket reduction.  The upload (32 bytes per scalar at the
// link's rate: 0.6 ms of a 3.5 ms call at 2^20) cannot hide under the sort of the same scalars, but a later chunk's upload and sort can run under an
// earlier chunk's accumulation.  Chunk k goes up and through its own sort on its own context (chunk 0 on the caller's, the others on borrowed ones,
// on their high-priority streams: their sorts must get workgroups in while an accumulation fills the chip); its accumulation starts when chunk k - 1's
// bucket sums are final and is SEEDED with them (k_accum28<.., SEED>: the first slice of a bucket continues from the newest earlier sum of that bucket),
// so after the last chunk every bucket's total sits in the newest chunk that touched it and ONE reduction (+ host tail) follows, reading through the
// chain.  All chunks use the window of the whole request.  The chunks grow — each must hide its upload + sort under its predecessor's accumulation,
// which costs ~ 3x as much per point: 2 chunks of 37 / 63 % up to 2^20 points, 3 of 18 / 30 / 52 % beyond.  A copy from pageable memory keeps the calling
// thread inside the runtime until the bytes are staged, so the order of the calls below IS the schedule: copy, launches, next copy.
// Measured (2^20 points, host scalars, wall per call; one whole upload: 3.50 ms): this form 3.23 (2^21: 6.51 -> 5.49, 2^22: 12.37 -> 9.84 with three chunks);
// two halves with a merge kernel (2^19 lane-pair additions into a dense array) before the reduction 3.28; the copies from a thread of their own: no
// change; every kernel on ONE stream with only the copies beside it 3.74 — each extra sort costs ~0.17 ms of dependent ~10 us launches when nothing
// hides it.  What is left on the table: a sort queued beside an accumulation that holds every wave slot (248 VGPRs x 2 waves per SIMD, workgroups that
// live ~250 us) takes ~0.5 ms instead of 0.15, so the next accumulation starts ~0.2 ms late.  CU-masked streams do not recover it (hipExtStreamCreateWithCUMask,
// profiles/r04_chunk_cumask_ab.jsonl): sorts confined to 16-64 reserved CUs are 4-8x slower (2^20: 4.5 / 3.8 / 3.4 ms with 16 / 32 / 64 CUs against 3.26), and
// keeping the accumulations off 8-32 CUs while the sorts run anywhere changes nothing at 2^20 and costs 5-8 % beyond.
// (Round 3 ran two halves as two complete MSMs on two host threads: that paid the 0.4 ms bucket reduction twice and lost below 2^21 points.)
// dev_job != nullptr: the scalars are already on the device (any number of sets and segments: every segment is cut at the same fractions; n = the request's
// reach, its tier) — nothing is uploaded, and what the later chunks hide under the earlier chunks' accumulation is their sort alone, so the first chunk is
// smaller (28 / 72 %, 12 / 28 / 60 %); the helper streams first wait for an event on `s`, where the caller's scalars may still be in flight.
static int32_t msm_run_chunked(Ctx* c, HelperSet& hs, uint64_t* out_jac18, const PinnedBases& pb, size_t n, bool mont, hipStream_t s, const void* host_src, const MsmJob* dev_job) {
  const uint32_t Q = 1 + (uint32_t)hs.ctx.size();           // 2 or 3
  static const uint32_t share_host[4][3] = {{0, 0, 0}, {0, 0, 0}, {37, 63, 0}, {18, 30, 52}}; const uint32_t (*share_dev)[3] = CHUNK_SHARE_DEV;
  static const uint32_t share0_env = [] { const char* e = std::getenv("ALEO_MI355X_CHUNK_SHARE0"); const int k = e ? std::atoi(e) : 0; return (uint32_t)(k >= 5 && k <= 95 ? k : 0); }();      // experiment knob: first chunk's percentage of a two-chunk host-scalar request
  uint32_t share_env[4][3] = {{0, 0, 0}, {0, 0, 0}, {share0_env, 100 - share0_env, 0}, {18, 30, 52}};
  const uint32_t (*share)[3] = dev_job ? share_dev : (share0_env ? share_env : share_host);
  const uint32_t K = dev_job ? dev_job->k : 1u;
  Ctx* cx[3] = {c, Q > 1 ? hs.ctx[0] : nullptr, Q > 2 ? hs.ctx[1] : nullptr}; hipStream_t st[3] = {s, Q > 1 ? hs.ctx[0]->hi : nullptr, Q > 2 ? hs.ctx[1]->hi : nullptr};
  StreamDrainGuard guard; for (uint32_t k = 0; k < Q; ++k) { guard.add(st[k]); guard.add(cx[k]->side); }
  size_t lo[4] = {0, 0, 0, 0};
  for (uint32_t k = 0, acc = 0; k < Q; ++k) { acc += share[Q][k]; lo[k + 1] = k + 1 == Q ? n : (((size_t)((double)n * acc / 100.0)) + 255) & ~(size_t)255; if (lo[k + 1] > n) lo[k + 1] = n; }
  Front f[3]; MsmSeg seg[3]; std::vector<MsmSeg> dsegs[3]; int32_t rc; uint32_t cum[4] = {0, 0, 0, 0};
  for (uint32_t k = 0; k < Q; ++k) cum[k + 1] = k + 1 == Q ? 100u : cum[k] + share[Q][k];
  if (dev_job) {
    HIPCHK(hipEventRecord(c->ev[4], s));                     // ev[4]: free here (no slice trees aside in a chunked request)
    for (uint32_t k = 1; k < Q; ++k) HIPCHK(hipStreamWaitEvent(st[k], c->ev[4], 0));
  }
  auto drain = [&](int32_t code) { const std::string keep = g_last_error; for (uint32_t k = 0; k < Q; ++k) (void)hipStreamSynchronize(st[k]); g_last_error = keep; return code; };
  auto view = [&](uint32_t k) { return FrontView{cx[k]->partial.as<char>(), f[k].sp.hist, f[k].sp.scan_local, f[k].sp.scan_blk}; };
  auto chain_before = [&](uint32_t k) { FrontChain ch; for (uint32_t i = k; i-- > 0;) ch.v[ch.n++] = view(i); return ch; };      // newest first
  for (uint32_t k = 0; k < Q; ++k) {
    MsmJob j; j.mont = mont; j.tier_n = n; j.k = K;
    if (dev_job) {                                           // every segment cut at the same fractions (boundaries on multiples of 256 scalars)
      for (uint32_t q = 0; q < dev_job->nseg; ++q) {
        const MsmSeg& g = dev_job->segs[q]; const size_t a0 = chunk_cut(g.len, cum[k]), a1 = chunk_cut(g.len, cum[k + 1]);
        if (a1 > a0) { MsmSeg h = g; h.d_ptr = (const char*)g.d_ptr + a0 * 32; h.len = a1 - a0; h.off = g.off + a0; dsegs[k].push_back(h); }
      }
      j.segs = dsegs[k].data(); j.nseg = (uint32_t)dsegs[k].size();
    } else {
      const size_t len = lo[k + 1] - lo[k];
      if ((rc = cx[k]->scalars_stage.reserve((len ? len : 1) * 32))) return drain(rc);
      if (hipMemcpyAsync(cx[k]->scalars_stage.p, (const char*)host_src + lo[k] * 32, len * 32, hipMemcpyHostToDevice, st[k]) != hipSuccess) { g_last_error = "msm: upload of a chunk failed"; return drain(ALEO_MI355X_ERR_HIP); }
      seg[k].d_ptr = cx[k]->scalars_stage.p; seg[k].len = len; seg[k].off = lo[k];
      j.segs = &seg[k]; j.nseg = 1;
    }
    if ((rc = msm_front_sort(cx[k], pb, j, st[k], f[k]))) return drain(rc);
    if (f[k].empty || !f[k].masked || f[k].P.c != f[0].P.c || f[k].sp.M != f[0].sp.M) { g_last_error = "msm: internal: chunks without a shared table window"; return drain(ALEO_MI355X_ERR_HIP); }
    if (k) { if ((rc = msm_front_finish(cx[k - 1], st[k - 1], f[k - 1], false))) return drain(rc); }      // chunk k - 1's slice trees: its bucket sums are final at its ev[2]
    const FrontChain seed = chain_before(k);
    if ((rc = msm_front_accum(cx[k], st[k], f[k], &seed, k ? cx[k - 1]->ev[2] : nullptr))) return drain(rc);
  }
  if ((rc = msm_front_finish(cx[Q - 1], st[Q - 1], f[Q - 1], false))) return drain(rc);
  // the reduction runs where the newest sums are; it synchronises that stream, behind which (event by event) every earlier chunk has finished
  const bool fire = dev_job && dev_job->fire_tail && c->tail_hook;
  if (fire) cx[Q - 1]->tail_hook = std::move(c->tail_hook);      // the caller's next kernels (queued on ITS stream by the hook) go out when the reduction has been queued
  c->tail_hook = fire ? nullptr : c->tail_hook;
  if ((rc = msm_back(cx[Q - 1], out_jac18, f[Q - 1], st[Q - 1], fire, chain_before(Q - 1)))) { cx[Q - 1]->tail_hook = nullptr; return drain(rc); }
  cx[Q - 1]->tail_hook = nullptr;
  if (!fire) HIPCHK(hipStreamSynchronize(s));                // (with a hook the caller's stream carries the hook's kernels: the caller orders its own work behind them)
  MsmTiming tm = cx[Q - 1]->last_msm; float ms = 0, kern = 0;
  for (uint32_t k = 0; k < Q; ++k) { HIPCHK(hipEventElapsedTime(&ms, cx[k]->ev[6], cx[k]->ev[5])); kern += ms; }
  tm.accum_kernel = kern / Q; tm.accum_launches = (int)Q;
  HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); tm.sort = ms;                        // the first chunk's sort: the one nothing hides
  HIPCHK(hipEventElapsedTime(&ms, c->ev[1], cx[Q - 1]->ev[2])); tm.accum = ms;               // from there to the last chunk's final bucket sums
  tm.total = tm.sort + tm.accum + tm.reduce + tm.host;
  c->last_msm = tm; g_last_msm = tm;
  guard.dismiss();
  return ALEO_MI355X_OK;
}


// Arbitrary request: k results, each the sum of its segments.  Results are grouped by the table tier the bases they reach select
// (longest tier first) and every group goes through msm_run in chunks of msm_max_sets() results / MAX_SEGS segments; results no tier
// serves (no table, or fewer than 2^10 bases reached) run one by one.
// The launch chains of one request.  One chain: on the caller's slot and stream.  Several big ones: dealt to two host threads — the caller's on its
// slot, one more on a borrowed helper context (own stream and workspaces) — so that the sort, reduction and host tail of one chain run under the
// accumulation of the other (the accumulation is bound by VALU issue, the sort by memory: `concurrent_callers` in the bench line is the same effect
// across calls).  The helper stream waits for an event recorded on `s` first (the scalars may still be in flight there); both threads return
// with their streams drained, so the caller sees the usual synchronous call.
namespace {
struct Chain { std::vector<MsmSeg> segs; std::vector<uint32_t> results; size_t points = 0; bool sparse = false, fire_tail = false; };
inline bool chains_overlap_on() { static const bool v = [] { const char* e = std::getenv("ALEO_MI355X_CHAIN_OVERLAP"); return !(e && e[0] == '0'); }(); return v; }      // A/B switch
}
// The pipelined form (round 4; ALEO_MI355X_CHAIN_PIPELINE=0 restores the two host threads).  A 2^20-constraint proof showed what the two threads leave on
// the table (profiles/r04_varuna_2^20_timeline_two_threads.txt): both chains of a round sort first (2.6 ms with no accumulation running), then their accumulations
// share the chip, then both reductions trail — 28 of 80 ms per proof with no accumulation kernel on the card.  Here ONE host thread queues the chains so that
// the accumulations run back to back and everything else runs beside them:
//   chain i on context i mod R (R = 3: the caller's and two borrowed ones): sort + slice ordering on the context's HIGH-priority stream, the accumulation on
//   its normal-priority stream behind the previous chain's accumulation, slice trees + bucket reduction on the high-priority stream again.  Before the
//   host queues the sort of chain i + 1 it collects chain i + 1 - R (waits for its reduction; Horner, normalisation), whose context it takes over.  With
//   R = 3 that reduction ran beside accumulation i - 1, so sort i + 1 is queued when accumulation i starts and has all of it to finish; with R = 2 the
//   host would wait for reduction i - 1, which crawls beside accumulation i (an accumulation holds every wave slot: a 512-thread k_prog_final block waits
//   milliseconds for a whole CU to drain), and sort i + 1 would run exposed after it — measured: 1.3 ms gaps between the accumulations of an
//   8-instance round.
static bool chain_pipeline_on() { static const bool v = [] { const char* e = std::getenv("ALEO_MI355X_CHAIN_PIPELINE"); return !(e && e[0] == '0'); }(); return v; }
static int32_t run_chains_pipelined(Ctx* c, const std::vector<Ctx*>& helpers, uint64_t* out_jac18, const PinnedBases& pb, std::vector<Chain>& chains, bool mont, hipStream_t s) {
  static const bool on_hi = [] { const char* e = std::getenv("ALEO_MI355X_PIPELINE_HI"); return e && e[0] == '1'; }();      // A/B: the chains' sorts and reductions on the high-priority streams (rounds 3-4) instead of normal-priority ones
  auto SS = [&](Ctx* x) { return on_hi ? x->hi : x->aux; };
  const size_t n = chains.size(), R = 1 + helpers.size();   // a ring of R contexts: chain i on context i mod R
  std::vector<Ctx*> cx(R); std::vector<hipStream_t> acc_st(R);
  cx[0] = c; acc_st[0] = s; for (size_t k = 1; k < R; ++k) { cx[k] = helpers[k - 1]; acc_st[k] = helpers[k - 1]->stream; }
  StreamDrainGuard guard; for (size_t k = 0; k < R; ++k) { guard.add(SS(cx[k])); guard.add(acc_st[k]); guard.add(cx[k]->side); }      // also on an exception or an early return below
  std::vector<Front> f(n); std::vector<MsmJob> job(n); std::vector<char> live(n, 0);
  auto drain = [&](int32_t code) { return code; };           // (the guard drains)
  HIPCHK(hipEventRecord(c->ev[4], s));                       // the scalars may still be in flight on the caller's stream
  for (size_t k = 0; k < R; ++k) { HIPCHK(hipStreamWaitEvent(SS(cx[k]), c->ev[4], 0)); if (k) HIPCHK(hipStreamWaitEvent(acc_st[k], c->ev[4], 0)); }
  int32_t rc;
  auto sort_of = [&](size_t i) -> int32_t {
    Chain& ch = chains[i]; MsmJob& g = job[i];
    g.segs = ch.segs.data(); g.nseg = (uint32_t)ch.segs.size(); g.k = (uint32_t)ch.results.size(); g.mont = mont; g.sparse = ch.sparse; g.fire_tail = false;
    Ctx* cc = cx[i % R];
    const int32_t r = msm_front_sort(cc, pb, g, SS(cc), f[i]);
    if (r) return r;
    if (f[i].empty) { for (size_t q = 0; q < ch.results.size(); ++q) host::hstore_jacobian_normalized(out_jac18 + 18 * (size_t)ch.results[q], host::HXYZZ::infinity()); return ALEO_MI355X_OK; }
    if (!f[i].masked) { g_last_error = "msm: internal: a pipelined chain without a table tier"; return ALEO_MI355X_ERR_HIP; }
    live[i] = 1;
    HIPCHK(hipEventRecord(cc->ev_hop, SS(cc)));
    return ALEO_MI355X_OK;
  };
  auto collect = [&](size_t i) -> int32_t {
    if (!live[i]) return ALEO_MI355X_OK;
    uint64_t res[MAX_SETS * 18]; Ctx* cc = cx[i % R];
    const int32_t r = msm_back(cc, res, f[i], SS(cc), false, FrontChain{}, true, true);
    if (r) return r;
    for (size_t q = 0; q < chains[i].results.size(); ++q) std::memcpy(out_jac18 + 18 * (size_t)chains[i].results[q], res + 18 * q, 144);
    live[i] = 0;
    return ALEO_MI355X_OK;
  };
  if ((rc = sort_of(0))) return drain(rc);
  hipEvent_t prev_accum = nullptr; size_t collected = 0;     // chains [0, collected) are done
  for (size_t i = 0; i < n; ++i) {
    Ctx* cc = cx[i % R]; hipStream_t as = acc_st[i % R];
    if (live[i]) {
      HIPCHK(hipStreamWaitEvent(as, cc->ev_hop, 0));
      if ((rc = msm_front_accum(cc, as, f[i], nullptr, prev_accum))) return drain(rc);
      prev_accum = cc->ev[5];
    }
    if (i + 1 < n) {
      for (; collected + R <= i + 1; ++collected) if ((rc = collect(collected))) return drain(rc);       // the context of chain i + 1 must be free: chain i + 1 - R collected
      if (live[i]) HIPCHK(hipStreamWaitEvent(SS(cx[(i + 1) % R]), cc->ev[6], 0));      // not before accumulation i starts: two sorts side by side only delay the first accumulation
      if ((rc = sort_of(i + 1))) return drain(rc);
    }
    if (live[i]) {
      HIPCHK(hipStreamWaitEvent(SS(cc), cc->ev[5], 0));
      if ((rc = msm_front_finish(cc, SS(cc), f[i], true))) return drain(rc);
      if ((rc = msm_back(cc, nullptr, f[i], SS(cc), false, FrontChain{}, false, false))) return drain(rc);
    }
  }
  for (; collected < n; ++collected) if ((rc = collect(collected))) return drain(rc);
  return drain(ALEO_MI355X_OK);
}
static int32_t run_chains(Ctx* c, uint64_t* out_jac18, const PinnedBases& pb, std::vector<Chain>& chains, bool mont, hipStream_t s, bool lean = false) {
  bool lean_now = false;                                    // chains run one after another on the caller's context keep MsmJob::lean; overlapped / pipelined chains order themselves by the phase events
  auto run_one = [&](Ctx* cc, Chain& ch, hipStream_t st) -> int32_t {
    uint64_t res[MAX_SETS * 18];
    MsmJob g; g.segs = ch.segs.data(); g.nseg = (uint32_t)ch.segs.size(); g.k = (uint32_t)ch.results.size(); g.mont = mont; g.sparse = ch.sparse; g.fire_tail = ch.fire_tail; g.lean = lean_now;
    const int32_t rc = msm_run(cc, res, pb, g, st);
    if (rc) return rc;
    for (size_t i = 0; i < ch.results.size(); ++i) std::memcpy(out_jac18 + 18 * (size_t)ch.results[i], res + 18 * i, 144);
    return ALEO_MI355X_OK;
  };
  size_t total = 0; for (auto& ch : chains) total += ch.points;
  HelperSet hs;
  if (chains.size() >= 2 && total >= ((size_t)1 << 20) && chains_overlap_on() && c->dev) { const int32_t rc = acquire_helpers(c->dev, chain_pipeline_on() && chains.size() >= 3 ? 2 : 1, hs); if (rc) return rc; }
  if (hs.ctx.empty()) { lean_now = lean; for (auto& ch : chains) { const int32_t rc = run_one(c, ch, s); if (rc) return rc; } return ALEO_MI355X_OK; }
  Ctx* h = hs.ctx[0];
  if (chain_pipeline_on()) {
    bool all_tiered = true;                                   // every chain on a table tier (the grouping of msm_batch makes them so, except the tier-less singles)
    for (auto& ch : chains) {
      size_t reach = 0; for (auto& g : ch.segs) if (g.len) reach = g.off + g.len > reach ? g.off + g.len : reach;
      bool t_ok = reach == 0;                                 // (an empty chain: its results are the identity)
      if (reach) for (const auto& t : pb.tab) if (t.d && reach >= t.min_n && reach <= t.cover) { t_ok = ch.results.size() <= 1 || ch.results.size() <= msm_max_sets(pb, reach); break; }
      all_tiered = all_tiered && t_ok && ch.results.size() <= MAX_SETS;
    }
    if (all_tiered) return run_chains_pipelined(c, hs.ctx, out_jac18, pb, chains, mont, s);
  }
  HIPCHK(hipEventRecord(c->ev[4], s));                     // ev[4] is free until this chain's own msm_run (which may use it for its aside trees) starts
  HIPCHK(hipStreamWaitEvent(h->stream, c->ev[4], 0));
  std::atomic<size_t> next{0}; int32_t rc_h = ALEO_MI355X_OK; std::string err_h; MsmTiming tm_h{};
  std::thread helper([&] {
    if (hipSetDevice(c->device) != hipSuccess) { rc_h = ALEO_MI355X_ERR_HIP; err_h = "hipSetDevice failed"; return; }
    try {
      for (size_t i; (i = next.fetch_add(1)) < chains.size();) { const int32_t rc = run_one(h, chains[i], h->stream); if (rc) { rc_h = rc; err_h = g_last_error; return; } }
    } catch (...) { rc_h = ALEO_MI355X_ERR_HIP; err_h = "msm: exception on the helper thread"; }
    tm_h = h->last_msm;
  });
  int32_t rc_m = ALEO_MI355X_OK;
  try { for (size_t i; !rc_m && (i = next.fetch_add(1)) < chains.size();) rc_m = run_one(c, chains[i], s); }
  catch (...) { rc_m = ALEO_MI355X_ERR_HIP; g_last_error = "msm: exception on the calling thread"; next.store(chains.size()); }      // never unwind past the joinable helper
  helper.join();
  (void)hipStreamSynchronize(h->stream);
  if (rc_m) return rc_m;
  if (rc_h) { g_last_error = err_h; return rc_h; }
  return ALEO_MI355X_OK;
}

// One result over n points.  Scalars already on the device: one launch chain.  HOST scalars against a table tier, from 2^ALEO_MI355X_MERGE_MIN_LG points
// (default 19; 0 = never): two or three chunks on as many contexts that share the buckets and one bucket reduction (msm_run_chunked) — most of the upload
// disappears under the earlier chunks' kernels.  Smaller or table-less requests upload whole.
static uint32_t chunks3_min_lg() { static const uint32_t v = [] { const char* e = std::getenv("ALEO_MI355X_CHUNKS3_MIN_LG"); const int k = e ? std::atoi(e) : 21; return (uint32_t)(k >= 0 && k <= 40 ? k : 21); }(); return v; }      // three chunks from here on (40 = never)
static uint32_t merge_min_lg() { static const uint32_t v = [] { const char* e = std::getenv("ALEO_MI355X_MERGE_MIN_LG"); const int k = e ? std::atoi(e) : 19; return (uint32_t)(k >= 0 && k <= 30 ? k : 19); }(); return v; }
int32_t msm_run1_split(Ctx* c, uint64_t* out_jac18, const PinnedBases& pb, const void* d_scalars, size_t n, bool mont, hipStream_t s, bool sparse, const void* host_src, bool may_merge) {
  bool tiered = false;
  for (const auto& t : pb.tab) if (t.d && n >= t.min_n && n <= t.cover) tiered = true;
  HelperSet hs;
  if (host_src && may_merge && !sparse && tiered && merge_min_lg() && n >= ((size_t)1 << merge_min_lg()) && c->dev) { const int32_t rc = acquire_helpers(c->dev, n >= ((size_t)1 << chunks3_min_lg()) ? 2 : 1, hs); if (rc) return rc; }
  if (!hs.ctx.empty()) return msm_run_chunked(c, hs, out_jac18, pb, n, mont, s, host_src);
  if (host_src) {
    const int32_t rc = c->scalars_stage.reserve((n ? n : 1) * 32); if (rc) return rc;
    if (n) HIPCHK(hipMemcpyAsync(c->scalars_stage.p, host_src, n * 32, hipMemcpyHostToDevice, s));
    d_scalars = c->scalars_stage.p;
  }
  return msm_run1(c, out_jac18, pb, d_scalars, n, mont, s, sparse);
}

int32_t msm_batch(Ctx* c, uint64_t* out_jac18, const PinnedBases& pb, const MsmJob& job, hipStream_t s) {
  const uint32_t K = job.k;
  auto tier_of = [&](size_t n) { for (int t = 0; t < 3; ++t) if (pb.tab[t].d && n >= pb.tab[t].min_n && n <= pb.tab[t].cover) return t; return -1; };
  std::vector<size_t> reach(K, 0), points(K, 0); std::vector<uint32_t> nsegs(K, 0);
  for (uint32_t q = 0; q < job.nseg; ++q) {
    const MsmSeg& g = job.segs[q];
    if (g.out >= K) { g_last_error = "msm: segment names a result that does not exist"; return ALEO_MI355X_ERR_BAD_ARG; }
    if (!g.len) continue;
    reach[g.out] = g.off + g.len > reach[g.out] ? g.off + g.len : reach[g.out]; points[g.out] += g.len; nsegs[g.out]++;
  }
  // Sparse hint + a range table that holds every segment: chains of up to its set capacity, whatever the reach (the table is indexed from range_off)
  if (job.sparse && pb.range.d) {
    bool inside = true;
    for (uint32_t q = 0; q < job.nseg && inside; ++q) { const MsmSeg& g = job.segs[q]; if (g.len) inside = g.off >= pb.range_off && g.off + g.len <= pb.range_off + pb.range.cover; }
    if (inside) {
      uint32_t cap = MAX_COARSE_ALL / ((1u << (pb.range.c - 1)) >> 8); cap = cap < MAX_SETS ? cap : MAX_SETS;
      std::vector<Chain> chains;
      for (uint32_t q0 = 0; q0 < K;) {
        uint32_t take = 0, sg = 0; size_t pts = 0;
        while (q0 + take < K && take < cap && (take == 0 || (pts + points[q0 + take] <= ((size_t)1 << 26) && sg + nsegs[q0 + take] <= MAX_SEGS))) { pts += points[q0 + take]; sg += nsegs[q0 + take]; ++take; }
        if (sg > MAX_SEGS) { g_last_error = "msm: one result with more than 64 segments"; return ALEO_MI355X_ERR_BAD_ARG; }
        chains.emplace_back(); Chain& ch = chains.back(); ch.segs.reserve(sg); ch.points = pts; ch.sparse = true; ch.fire_tail = q0 == 0 && take == K;
        for (uint32_t q = 0; q < job.nseg; ++q) { const MsmSeg& g = job.segs[q]; if (g.len && g.out >= q0 && g.out < q0 + take) { MsmSeg h = g; h.out = g.out - q0; ch.segs.push_back(h); } }
        for (uint32_t i = 0; i < take; ++i) ch.results.push_back(q0 + i);
        q0 += take;
      }
      return run_chains(c, out_jac18, pb, chains, job.mont, s, job.lean);
    }
  }
  // Latency-bound requests (one prover round: a few results of <= 2^17 points each): ONE launch chain on the tier that covers the
  // longest reach beats one chain per tier — a second chain costs ~0.45 ms of dependent steps, a wider window than a short member
  // would have picked costs nothing measurable at these sizes.
  {
    size_t far = 0, pts = 0, sg = 0;
    for (uint32_t q = 0; q < K; ++q) { far = reach[q] > far ? reach[q] : far; pts += points[q]; sg += nsegs[q]; }
    if (K > 1 && K <= MAX_SETS && tier_of(far) >= 0 && K <= msm_max_sets(pb, far) && pts <= ((size_t)1 << 21) && sg <= MAX_SEGS) {
      bool split = false;
      for (uint32_t q = 0; q < K; ++q) if (points[q] && tier_of(reach[q]) != tier_of(far)) split = true;
      if (split) {
        std::vector<MsmSeg> segs; segs.reserve(sg);
        for (uint32_t q = 0; q < job.nseg; ++q) if (job.segs[q].len) segs.push_back(job.segs[q]);
        MsmJob g; g.segs = segs.data(); g.nseg = (uint32_t)segs.size(); g.k = K; g.mont = job.mont; g.fire_tail = true; g.lean = job.lean;
        return msm_run(c, out_jac18, pb, g, s);
      }
    }
  }
  std::vector<uint32_t> todo; todo.reserve(K); std::vector<Chain> chains;
  for (int t = -1; t < 3; ++t) {
    todo.clear();
    for (uint32_t q = 0; q < K; ++q) if (tier_of(reach[q]) == t) todo.push_back(q);
    size_t pos = 0;
    while (pos < todo.size()) {
      const size_t cap = t < 0 ? 1 : msm_max_sets(pb, reach[todo[pos]]);      // (chains of one set instead of two for the pipeline of run_chains: measured slower, 8 x 2^20 constraints 155 -> 160 ms)
      size_t take = 0, pts = 0, sg = 0; uint32_t local[MAX_SETS];
      // 2^32 pairs and MAX_SEGS segments per launch chain: chunks are also cut by total points and segments
      while (pos + take < todo.size() && take < cap && (take == 0 || (pts + points[todo[pos + take]] <= ((size_t)1 << 26) && sg + nsegs[todo[pos + take]] <= MAX_SEGS))) {
        pts += points[todo[pos + take]]; sg += nsegs[todo[pos + take]]; ++take;
      }
      if (sg > MAX_SEGS) { g_last_error = "msm: one result with more than 64 segments"; return ALEO_MI355X_ERR_BAD_ARG; }
      chains.emplace_back(); Chain& ch = chains.back(); ch.segs.reserve(sg); ch.points = pts; ch.fire_tail = take == K;      // fire_tail: every result of the request in this one chain
      for (size_t i = 0; i < take; ++i) { local[i] = todo[pos + i]; ch.results.push_back(local[i]); }
      for (uint32_t q = 0; q < job.nseg; ++q) {
        const MsmSeg& g = job.segs[q];
        if (!g.len) continue;
        for (size_t i = 0; i < take; ++i) if (local[i] == g.out) { MsmSeg h = g; h.out = (uint32_t)i; ch.segs.push_back(h); break; }
      }
      pos += take;
    }
  }
  return run_chains(c, out_jac18, pb, chains, job.mont, s, job.lean);
}

// ---- synthetic base sets generated in HBM: P_i = (first + i) * G --------------------------------------
// Off the hot path (setup): every group operation is an out-of-line call, code size over speed.
static constexpr uint32_t GEN_K = 64;      // consecutive points per lane
__device__ __constant__ uint32_t FQ_P_MINUS_2[12] = {0xffffffffu, 0x8508bfffu, 0x30000000u, 0x170b5d44u, 0xba094800u, 0x1ef3622fu,
                                                     0x00f5138fu, 0x1a22d9f3u, 0x6ca1493bu, 0xc63b05c0u, 0x17c510eau, 0x01ae3a46u};
__device__ __noinline__ void fq_mul_ni(Fq* r, const Fq* a, const Fq* b) { *r = Fq::mul(*a, *b); }
__device__ __noinline__ void fq_inverse_ni(Fq* io) {   // a^(q-2), a < 2q
  Fq a = *io, acc = Fq::one();
  for (int bit = 376; bit >= 0; --bit) {
    fq_mul_ni(&acc, &acc, &acc);
    if ((FQ_P_MINUS_2[bit >> 5] >> (bit & 31)) & 1u) fq_mul_ni(&acc, &acc, &a);
  }
  *io = acc;
}

__global__ void __launch_bounds__(256) k_gen_xyzz(const char* __restrict__ g_affine, uint64_t first, uint32_t n, char* __restrict__ tmp) {
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  uint64_t i0 = (uint64_t)t * GEN_K; if (i0 >= n) return;
  AffinePt g = load_affine(g_affine);
  XYZZ G; G.X = g.x; G.Y = g.y; G.ZZ = Fq::one(); G.ZZZ = Fq::one();
  uint64_t k = first + i0;
  XYZZ acc = xyzz_infinity();
  for (int bit = 63 - __clzll(k); bit >= 0; --bit) {
    xyzz_double_ni(&acc);
    if ((k >> bit) & 1ull) xyzz_add_ni(&acc, &G);
  }
  for (uint32_t j = 0; j < GEN_K && i0 + j < n; ++j) {
    store_xyzz(tmp + (i0 + j) * 192, acc);
    xyzz_add_ni(&acc, &G);
  }
}
// XYZZ -> affine with one shared inversion per lane (Montgomery's trick over the lane's GEN_K points)
__global__ void __launch_bounds__(256) k_gen_normalize(char* __restrict__ tmp, uint32_t n, char* __restrict__ prefix, char* __restrict__ out_xy) {
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  uint64_t i0 = (uint64_t)t * GEN_K; if (i0 >= n) return;
  uint32_t cnt = (uint32_t)((n - i0) < GEN_K ? (n - i0) : GEN_K);
  Fq prod = Fq::one();
  for (uint32_t j = 0; j < cnt; ++j) {
    store_fp<Fq>(prefix + (i0 + j) * 48, prod);
    Fq zzz = load_fp<Fq>(tmp + (i0 + j) * 192 + 144);
    if (zzz.is_zero_mod_lt2p()) zzz = Fq::one();      // the identity: keep it out of the shared inversion
    fq_mul_ni(&prod, &prod, &zzz);
  }
  fq_inverse_ni(&prod);
  for (uint32_t jj = cnt; jj-- > 0;) {
    const char* src = tmp + (i0 + jj) * 192;
    Fq pre = load_fp<Fq>(prefix + (i0 + jj) * 48), zzz = load_fp<Fq>(src + 144), zz = load_fp<Fq>(src + 96);
    Fq zi3, zi, zi2, x, y;
    if (zzz.is_zero_mod_lt2p()) {                       // identity -> (0, 0): never on the curve, callers skip it
      store_fp<Fq>(out_xy + (i0 + jj) * 96, Fq::zero()); store_fp<Fq>(out_xy + (i0 + jj) * 96 + 48, Fq::zero());
      continue;
    }
    fq_mul_ni(&zi3, &prod, &pre);            // 1/ZZZ_j
    fq_mul_ni(&prod, &prod, &zzz);
    fq_mul_ni(&zi, &zz, &zi3);               // 1/Z
    fq_mul_ni(&zi2, &zi, &zi);               // 1/ZZ
    Fq X = load_fp<Fq>(src), Y = load_fp<Fq>(src + 48);
    fq_mul_ni(&x, &X, &zi2); fq_mul_ni(&y, &Y, &zi3);
    store_fp<Fq>(out_xy + (i0 + jj) * 96, Fq::reduce(x));
    store_fp<Fq>(out_xy + (i0 + jj) * 96 + 48, Fq::reduce(y));
  }
}

int32_t generate_multiples(Ctx* c, const void* base104, uint64_t first, size_t n, PinnedBases* out) {
  if (n == 0 || n >= (1ull << 31) || first == 0) { g_last_error = "bases_generate: bad range"; return ALEO_MI355X_ERR_BAD_ARG; }
  DevTmp xy, g, tmp, pre; int32_t rc;           // freed on every return path; xy is handed to the caller at the end
  if ((rc = xy.alloc(n * 96)) || (rc = g.alloc(96)) || (rc = tmp.alloc(n * 192)) || (rc = pre.alloc(n * 48))) return rc;
  HIPCHK(hipMemcpyAsync(g.p, base104, 96, hipMemcpyHostToDevice, c->stream));
  uint32_t lanes = (uint32_t)((n + GEN_K - 1) / GEN_K), grid = (lanes + 255) / 256;
  hipLaunchKernelGGL(k_gen_xyzz, dim3(grid), dim3(256), 0, c->stream, (const char*)g.p, first, (uint32_t)n, (char*)tmp.p);
  hipLaunchKernelGGL(k_gen_normalize, dim3(grid), dim3(256), 0, c->stream, (char*)tmp.p, (uint32_t)n, (char*)pre.p, (char*)xy.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  PinnedBases pb; pb.n = n; pb.d_xy = xy.p;
  if ((rc = make_rows28(c, &pb))) return rc;    // xy still owned here: freed on failure
  xy.release();
  *out = pb; return ALEO_MI355X_OK;
}

// P_i = s_i * G for caller-supplied canonical scalars (SRS-shaped bases P_i = beta^i * G, whose commitment to p
// is p(beta) * G — what the opening equation of KZG10 needs).  One lane per point, plain double-and-add (setup, not timed).
__global__ void __launch_bounds__(256) k_gen_scalar_mul(const char* __restrict__ g_affine, const uint32_t* __restrict__ scalars, uint32_t n, char* __restrict__ tmp,
                                                        uint8_t* __restrict__ inf, uint32_t* __restrict__ n_inf) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  AffinePt g = load_affine(g_affine);
  XYZZ G; G.X = g.x; G.Y = g.y; G.ZZ = Fq::one(); G.ZZZ = Fq::one();
  uint32_t k[8];
  for (int l = 0; l < 8; ++l) k[l] = scalars[(size_t)i * 8 + l];
  XYZZ acc = xyzz_infinity();
  for (int bit = 255; bit >= 0; --bit) {
    xyzz_double_ni(&acc);
    if ((k[bit >> 5] >> (bit & 31)) & 1u) xyzz_add_ni(&acc, &G);
  }
  store_xyzz(tmp + (size_t)i * 192, acc);
  const bool is_inf = acc.ZZZ.is_zero_mod();                 // s_i = 0 mod r: flagged like an uploaded Affine with infinity = true
  inf[i] = is_inf ? 1 : 0;
  if (is_inf) atomicAdd(n_inf, 1u);
}

int32_t generate_from_scalars(Ctx* c, const void* base104, const void* scalars32, size_t n, PinnedBases* out) {
  if (n == 0 || n >= (1ull << 31) || !scalars32) { g_last_error = "bases_from_scalars: bad range"; return ALEO_MI355X_ERR_BAD_ARG; }
  DevTmp xy, g, tmp, pre, sc, inf, cnt; int32_t rc;
  if ((rc = xy.alloc(n * 96)) || (rc = g.alloc(96)) || (rc = tmp.alloc(n * 192)) || (rc = pre.alloc(n * 48)) || (rc = sc.alloc(n * 32)) ||
      (rc = inf.alloc(n)) || (rc = cnt.alloc(4))) return rc;
  HIPCHK(hipMemcpyAsync(g.p, base104, 96, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(sc.p, scalars32, n * 32, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(cnt.p, 0, 4, c->stream));
  hipLaunchKernelGGL(k_gen_scalar_mul, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, (const char*)g.p, (const uint32_t*)sc.p, (uint32_t)n, (char*)tmp.p,
                     (uint8_t*)inf.p, (uint32_t*)cnt.p);
  uint32_t lanes = (uint32_t)((n + GEN_K - 1) / GEN_K), grid = (lanes + 255) / 256;
  hipLaunchKernelGGL(k_gen_normalize, dim3(grid), dim3(256), 0, c->stream, (char*)tmp.p, (uint32_t)n, (char*)pre.p, (char*)xy.p);
  HIPCHK(hipGetLastError());
  uint32_t n_inf = 0;
  HIPCHK(hipMemcpyAsync(&n_inf, cnt.p, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  PinnedBases pb; pb.n = n; pb.d_xy = xy.p;
  if ((rc = make_rows28(c, &pb))) return rc;
  xy.release();
  if (n_inf) pb.d_inf = (uint8_t*)inf.release();             // only sets that hold the identity carry flags (as uploaded sets do)
  *out = pb; return ALEO_MI355X_OK;
}

// ---- fixed-base table: row w = 2^(c * w) * P_i  (setup, once per pinned base set) -------------------
// With the table every window of a scalar feeds the same 2^(c-1) buckets (c = 20 at 2^20: 13 instead of 16 additions
// per point), one bucket reduction instead of W, and no Horner tail.  Costs W x 96 bytes of HBM per point (there are
// 288 GB) and ~250 doublings per point once, at pin time — the SRS of a proving key never changes.
__global__ void __launch_bounds__(256) k_pre_init(const char* __restrict__ xy, uint32_t n, char* __restrict__ cur) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  AffinePt p = load_affine(xy + (size_t)i * 96);
  XYZZ a; a.X = p.x; a.Y = p.y; a.ZZ = Fq::one(); a.ZZZ = Fq::one();
  if (p.x.is_zero_raw() && p.y.is_zero_raw()) a = xyzz_infinity();
  store_xyzz(cur + (size_t)i * 192, a);
}
__global__ void __launch_bounds__(256) k_pre_double(char* __restrict__ cur, uint32_t n, int doublings) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  XYZZ a = load_xyzz(cur + (size_t)i * 192);
  for (int d = 0; d < doublings; ++d) xyzz_double_ni(&a);
  store_xyzz(cur + (size_t)i * 192, a);
}

// 104-byte Affine rows (x | y | infinity byte | padding) -> 96-byte rows, the flag bytes, and the number of flagged rows
__global__ void __launch_bounds__(256) k_unpack104(const char* __restrict__ src, char* __restrict__ dst, uint8_t* __restrict__ flags, uint32_t* __restrict__ count, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint2* s2 = (const uint2*)(src + i * 104); uint2* d2 = (uint2*)(dst + i * 96);
#pragma unroll
    for (int k = 0; k < 12; ++k) d2[k] = s2[k];
    const uint8_t f = (uint8_t)(s2[12].x & 0xffu) ? 1 : 0;
    flags[i] = f;
    if (f) atomicAdd(count, 1u);
  }
}
int32_t unpack_affine104(Ctx* c, const void* d_rows104, void* d_xy96, void* d_flags, size_t n, hipStream_t s) {
  (void)c;
  uint32_t* count = (uint32_t*)((char*)d_flags + ((n + 3) & ~(size_t)3));
  HIPCHK(hipMemsetAsync(count, 0, 4, s));
  const size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(k_unpack104, dim3((uint32_t)(want < 16384 ? want : 16384)), dim3(256), 0, s, (const char*)d_rows104, (char*)d_xy96, (uint8_t*)d_flags, count, n);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// the same conversion into a buffer the caller provides (the cold one-shot call's slot buffers), queued on s
int32_t rows_to28_into(const void* d_xy96, void* d_dst, size_t n, hipStream_t s) {
  if (n == 0) return ALEO_MI355X_OK;
  hipLaunchKernelGGL(k_rows_to28, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, (const char*)d_xy96, (char*)d_dst, (uint32_t)n);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}
int32_t make_rows28(Ctx* c, PinnedBases* pb) {
  if (pb->d_xy28 || pb->n == 0) return ALEO_MI355X_OK;
  DevTmp rows; int32_t rc;
  if ((rc = rows.alloc(pb->n * ROW28))) return rc;
  hipLaunchKernelGGL(k_rows_to28, dim3((uint32_t)((pb->n + 255) / 256)), dim3(256), 0, c->stream, (const char*)pb->d_xy, (char*)rows.p, (uint32_t)pb->n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  pb->d_xy28 = rows.release();
  return ALEO_MI355X_OK;
}

static int32_t build_table(Ctx* c, const PinnedBases* pb, int pre_c, size_t n, PinnedBases::PreTable* out, size_t off = 0) {
  const uint32_t W = (SCALAR_BITS + pre_c - 1) / pre_c;
  if (n * (size_t)W >= (1ull << 31)) { g_last_error = "bases_precompute: table index would exceed 31 bits"; return ALEO_MI355X_ERR_BAD_ARG; }
  DevTmp tab, cur, prefix, row; int32_t rc;     // freed on every return path; tab is handed over at the end
  if ((rc = tab.alloc(n * ROW28 * W))) return rc;                              // rows in the accumulation kernel's 28-bit format (fp28.h)
  if ((rc = cur.alloc(n * 192)) || (rc = prefix.alloc(n * 48)) || (rc = row.alloc(n * 96))) return rc;
  hipStream_t s = c->stream;
  const uint32_t g = (uint32_t)((n + 255) / 256), lanes = (uint32_t)((n + GEN_K - 1) / GEN_K), gl = (lanes + 255) / 256;
  const char* xy = (const char*)pb->d_xy + off * 96;      // the table covers points [off, off + n) of the set
  if (pb->d_xy28) HIPCHK(hipMemcpyAsync(tab.p, (const char*)pb->d_xy28 + off * ROW28, n * ROW28, hipMemcpyDeviceToDevice, s));
  else hipLaunchKernelGGL(k_rows_to28, dim3(g), dim3(256), 0, s, xy, (char*)tab.p, (uint32_t)n);
  hipLaunchKernelGGL(k_pre_init, dim3(g), dim3(256), 0, s, xy, (uint32_t)n, (char*)cur.p);
  for (uint32_t w = 1; w < W; ++w) {
    hipLaunchKernelGGL(k_pre_double, dim3(g), dim3(256), 0, s, (char*)cur.p, (uint32_t)n, win_width(pre_c, (int)w - 1));      // row w = 2^win_offset(w) * P
    hipLaunchKernelGGL(k_gen_normalize, dim3(gl), dim3(256), 0, s, (char*)cur.p, (uint32_t)n, (char*)prefix.p, (char*)row.p);
    hipLaunchKernelGGL(k_rows_to28, dim3(g), dim3(256), 0, s, (const char*)row.p, (char*)tab.p + (size_t)w * n * ROW28, (uint32_t)n);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));
  out->d = tab.release(); out->c = pre_c; out->cover = n;
  return ALEO_MI355X_OK;
}

// Tiers (measured, tools/small_probe.py): c = 20 needs >= 2^17 points per call to fill its 2^19 buckets, c = 16 wins from 2^15,
// c = 13 from 2^10 (0.5 ms against 1.1-1.4 ms on the plain path); KZG10::commit multiplies polynomials of every degree against
// prefixes of ONE SRS, so a pinned set carries a table for each range it can serve.  The two small tiers cost < 20 % extra HBM
// and build time of a 2^20-point set.
int32_t msm_precompute(Ctx* c, PinnedBases* pb) {
  if (pb->tabled || pb->n == 0) return ALEO_MI355X_OK;
  const size_t N = pb->n; int32_t rc = ALEO_MI355X_OK; int k = 0;
  auto lim = [&](size_t cap) { return N < cap ? N : cap; };
  auto tier = [&](int cbits, size_t cover, size_t min_n) {
    if (rc) return;
    if ((rc = build_table(c, pb, cbits, cover, &pb->tab[k])) == ALEO_MI355X_OK) pb->tab[k++].min_n = min_n;
  };
  // Prefix tiers reach TIER_SLACK points past their power of two: a committer key is a power of two of powers FOLLOWED by a few hiding
  // powers, and commitments that touch those (every hiding one) would otherwise fall through to the next wider window (2x the buckets per
  // result: with 25 results in a chain, as many as a batch of 8 instances commits in its first round, most of the reduction time).
  constexpr size_t TIER_SLACK = 64;
  const size_t cover16 = lim(((size_t)1 << 17) + TIER_SLACK), cover13 = lim(((size_t)1 << 15) + TIER_SLACK);
  if (N > cover16) tier(N >= (1u << 19) ? 20 : 17, N, cover16 + 1);      // up to cover16 points the c = 16 tier is faster (2^17: 0.90 vs 1.18 ms)
  if (N >= (1u << 15)) tier(16, cover16, (size_t)1 << 15);
  if (N >= (1u << 10)) tier(13, cover13, (size_t)1 << 10);
  if (rc) {                                     // a later tier failed (out of memory): give back the ones already built
    for (auto& t : pb->tab) { if (t.d) (void)hipFree(t.d); t = PinnedBases::PreTable(); }
    return rc;
  }
  pb->tabled = true;
  return ALEO_MI355X_OK;
}

int32_t msm_precompute_range(Ctx* c, PinnedBases* pb, size_t off, size_t n, int window_bits) {
  if (pb->range.d) { g_last_error = "bases_precompute_range: this set already has a range table"; return ALEO_MI355X_ERR_BAD_ARG; }
  if (!n || off + n > pb->n || (window_bits != 13 && window_bits != 16)) { g_last_error = "bases_precompute_range: bad range or window (13 or 16 bits)"; return ALEO_MI355X_ERR_BAD_ARG; }
  int32_t rc = build_table(c, pb, window_bits, n, &pb->range, off);
  if (rc) return rc;
  pb->range.min_n = 0; pb->range_off = off;
  return ALEO_MI355X_OK;
}

// ---- self-test of the 28-bit-limb mixed addition against the 32-bit formulas (test hook; tools/ubench/madd28_check.hip) ----
__device__ __forceinline__ uint32_t st_rng(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }
__device__ __forceinline__ Fq st_rnd_fq(uint64_t& s) { Fq r; for (int i = 0; i < 12; ++i) r.v[i] = st_rng(s); r.v[11] &= 0x00ffffffu; return Fq::reduce(r); }
__device__ __forceinline__ bool st_same(const Fq& a, const Fq& b) { Fq x = Fq::reduce(a), y = Fq::reduce(b); uint32_t d = 0; for (int i = 0; i < 12; ++i) d |= x.v[i] ^ y.v[i]; return d == 0; }
__global__ void __launch_bounds__(256) k_selftest_madd28(uint32_t* bad, uint32_t steps, uint64_t seed) {
  uint64_t s = seed * (blockIdx.x * 256 + threadIdx.x + 1);
  Fq z = st_rnd_fq(s);
  if (!st_same(f28_to_fq(f28_from_fq(z)), z)) { atomicAdd(bad, 1u); return; }                 // representation round trip
  XYZZ a; a.X = st_rnd_fq(s); a.Y = st_rnd_fq(s); a.ZZ = Fq::one(); a.ZZZ = Fq::one();
  XYZZ28 b; b.X = f28_from_fq(a.X); b.Y = f28_from_fq(a.Y); b.ZZ = f28_const(ONE28); b.ZZZ = f28_const(ONE28);
  for (uint32_t it = 0; it < steps; ++it) {
    Fq x = st_rnd_fq(s), y = st_rnd_fq(s);
    F28 x28 = f28_from_fq(x), y28 = f28_from_fq(y);
    if (st_rng(s) & 1) { y = fq_neg_canonical(y); y28 = f28_sub<2, 1>(f28_const(Limbs14{}), y28); }
    if (it == steps / 2) { x = Fq::reduce(a.X); x28 = f28_from_fq(x); }                       // same x as acc (ZZ == 1 only at it == 0, so usually a plain point)
    const bool ok32 = xyzz_madd_fast(a, x, y), ok28 = xyzz28_madd_fast(b, x28, y28);
    if (ok32 != ok28) { atomicAdd(bad, 1u); return; }
    if (!ok32) break;
    if (!st_same(f28_to_fq(b.X), a.X) || !st_same(f28_to_fq(b.Y), a.Y) || !st_same(f28_to_fq(b.ZZ), a.ZZ) || !st_same(f28_to_fq(b.ZZZ), a.ZZZ)) { atomicAdd(bad, 1u); return; }
  }
  XYZZ c; c.X = st_rnd_fq(s); c.Y = st_rnd_fq(s); c.ZZ = Fq::one(); c.ZZZ = Fq::one();        // P == acc must be refused by both
  XYZZ28 d; d.X = f28_from_fq(c.X); d.Y = f28_from_fq(c.Y); d.ZZ = f28_const(ONE28); d.ZZZ = f28_const(ONE28);
  if (xyzz_madd_fast(c, c.X, c.Y) || xyzz28_madd_fast(d, d.X, d.Y)) atomicAdd(bad, 1u);
}
int32_t selftest_madd28(Ctx* c, uint32_t lanes, uint32_t steps, uint64_t seed, uint32_t* failures) {
  int32_t rc; if ((rc = c->scalars_stage.reserve(64))) return rc;
  uint32_t* d = c->scalars_stage.as<uint32_t>();
  HIPCHK(hipMemsetAsync(d, 0, 4, c->stream));
  hipLaunchKernelGGL(k_selftest_madd28, dim3((lanes + 255) / 256), dim3(256), 0, c->stream, d, steps, seed | 1ull);
  HIPCHK(hipMemcpyAsync(failures, d, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// The lane-quad addition against the lane-pair one (which the MSM parity tests pin): same values mod q in all four coordinates, same
// infinity flag, over random operands and the special cases (either operand the identity, equal points, opposite points).
__global__ void __launch_bounds__(256) k_selftest_addquad(char* buf, uint32_t ops, uint64_t seed, uint32_t* bad) {
  const uint32_t op = (blockIdx.x * 256 + threadIdx.x) >> 2, q = threadIdx.x & 3u;
  if (op >= ops) return;
  char* A = buf + (size_t)op * 4 * PB28; char* B = A + PB28; char* O4 = B + PB28; char* O2 = O4 + PB28;
  if (q == 0) {
    uint64_t s = seed * (op + 1);
    XYZZ28 a, b;
    a.X = f28_from_fq(st_rnd_fq(s)); a.Y = f28_from_fq(st_rnd_fq(s)); a.ZZ = f28_from_fq(st_rnd_fq(s)); a.ZZZ = f28_from_fq(st_rnd_fq(s));
    b.X = f28_from_fq(st_rnd_fq(s)); b.Y = f28_from_fq(st_rnd_fq(s)); b.ZZ = f28_from_fq(st_rnd_fq(s)); b.ZZZ = f28_from_fq(st_rnd_fq(s));
    const uint32_t kind = op % 16;
    if (kind == 11) { a.ZZ = f28_const(Limbs14{}); }                                          // A the identity
    if (kind == 12) { b.ZZ = f28_const(Limbs14{}); }                                          // B the identity
    if (kind == 13) { a.ZZ = f28_const(Limbs14{}); b.ZZ = a.ZZ; }
    if (kind == 14) b = a;                                                                    // equal: doubling
    if (kind == 15) { b = a; b.Y = f28_normalise(f28_sub<2, 1>(f28_const(Limbs14{}), a.Y)); }   // opposite: the identity
    store_xyzz28(A, a); store_xyzz28(B, b);
  }
  pair_fence(); __syncthreads();
  xyzz28_add_quad(A, B, O4);
  if (q < 2) xyzz28_add_pair(A, B, O2);
  pair_fence(); __syncthreads();
  if (q == 0) {
    const bool i4 = f28_is_zero_raw(load_f28(O4 + 112)), i2 = f28_is_zero_raw(load_f28(O2 + 112));
    bool ok = i4 == i2;
    if (ok && !i4) for (int k = 0; k < 4; ++k) ok = ok && st_same(f28_to_fq(load_f28(O4 + 56 * k)), f28_to_fq(load_f28(O2 + 56 * k)));
    if (!ok) atomicAdd(bad, 1u);
  }
}
int32_t selftest_addquad(Ctx* c, uint32_t ops, uint64_t seed, uint32_t* failures) {
  ops = (ops + 63u) & ~63u;                                  // whole blocks: the kernel synchronises its threads
  int32_t rc; if ((rc = c->scalars_stage.reserve((size_t)ops * 4 * PB28 + 64))) return rc;
  char* buf = c->scalars_stage.as<char>() + 64; uint32_t* d = c->scalars_stage.as<uint32_t>();
  HIPCHK(hipMemsetAsync(d, 0, 4, c->stream));
  hipLaunchKernelGGL(k_selftest_addquad, dim3(ops / 64), dim3(256), 0, c->stream, buf, ops, seed | 1ull, d);
  HIPCHK(hipMemcpyAsync(failures, d, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// ---- element-wise products (parity tests pin the device arithmetic with these) -------------------
template <class F, bool SQR> __global__ void __launch_bounds__(256) k_fp_mul(char* r, const char* a, const char* b, uint32_t n) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  constexpr int bytes = F::N * 4;
  F x = load_fp<F>(a + (size_t)i * bytes), y = load_fp<F>(b + (size_t)i * bytes);
  store_fp<F>(r + (size_t)i * bytes, F::reduce(SQR ? F::sqr(x) : F::mul(x, y)));
}
template <class F> static int32_t launch_fp_mul(Ctx* c, void* r, const void* a, const void* b, size_t n) {
  constexpr size_t bytes = F::N * 4;
  if (n == 0) return ALEO_MI355X_OK;
  int32_t rc; if ((rc = c->scalars_stage.reserve(3 * n * bytes))) return rc;
  char* da = c->scalars_stage.as<char>(); char* db = da + n * bytes; char* dr = db + n * bytes;
  HIPCHK(hipMemcpyAsync(da, a, n * bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(db, b, n * bytes, hipMemcpyHostToDevice, c->stream));
  if (a == b)     // same host buffer for both operands: pin the dedicated squaring block instead of the general product
    hipLaunchKernelGGL((k_fp_mul<F, true>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, dr, da, db, (uint32_t)n);
  else
    hipLaunchKernelGGL((k_fp_mul<F, false>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, dr, da, db, (uint32_t)n);
  HIPCHK(hipMemcpyAsync(r, dr, n * bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}
int32_t launch_fq_mul(Ctx* c, void* r, const void* a, const void* b, size_t n) { return launch_fp_mul<Fq>(c, r, a, b, n); }
int32_t launch_fr_mul(Ctx* c, void* r, const void* a, const void* b, size_t n) { return launch_fp_mul<Fr>(c, r, a, b, n); }

}  // namespace aleo_mi355x

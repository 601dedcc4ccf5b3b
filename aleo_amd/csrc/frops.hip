// frops.hip — field-only vector kernels around the two operators: element-wise Fr products /
// sums / differences of evaluation vectors and batch inversion, on data that already lives in HBM between an NTT and
// the next NTT / commitment.
//
// Replaces (on the device) the loops snarkVM 0.14.5 runs on the CPU between FFTs in Varuna's rounds [UPSTREAM-RECALL]:
//   algorithms/src/fft/evaluations.rs      Evaluations::{mul_assign, add_assign, sub_assign}  (pointwise on a domain)
//   fields/src/traits/field.rs / lib.rs    snarkvm_fields::batch_inversion (Montgomery's trick; zeros stay zero)
// reached from the same call sites as the NTT: /root/reference/rust/src/program/execute.rs:74,177, transfer.rs:99.
//
// Element-wise ops are HBM-bound (96 algorithmic bytes per element against one 305-instruction product): 16-byte
// coalesced accesses, one element per lane, grid-stride.  Batch inversion: every lane owns a strided subsequence (so a
// wave's accesses stay contiguous), keeps prefix products in an HBM scratch vector and shares ONE Fermat inversion
// (a^(r-2), ~380 products) over its K elements: 3 + 380/K products per element.
#include "ctx.h"
#include "fp.h"
#include "chacha.h"
#include "host_field.hpp"
#include <cstring>

namespace aleo_mi355x {

__device__ __forceinline__ Fr fr_canonical_lt4r(const Fr& a) { return Fr::cond_sub<1>(Fr::cond_sub<2>(a)); }

template <int OP>
__global__ void __launch_bounds__(256) k_fr_vec_op(char* dst, const char* a, const char* b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    Fr x = load_fp<Fr>(a + i * 32), y = load_fp<Fr>(b + i * 32), r;
    if constexpr (OP == 0) r = Fr::mul(x, y);               // canonical inputs: < 2r
    else if constexpr (OP == 1) r = Fr::add(x, y);          // < 2r
    else r = Fr::sub<1>(x, y);                              // x + r - y < 2r
    store_fp<Fr>(dst + i * 32, Fr::cond_sub<1>(r));
  }
}

__device__ __constant__ uint32_t FR_R_MINUS_2[8] = {0xffffffffu, 0x0a117fffu, 0xd0000001u, 0x59aa76feu, 0x5c37b001u, 0x60b44d1eu, 0x9a2ca556u, 0x12ab655eu};
__device__ __noinline__ void fr_mul_ni(Fr* r, const Fr* a, const Fr* b) { *r = Fr::mul(*a, *b); }
__device__ __noinline__ void fr_inverse_ni(Fr* io) {       // a^(r-2); a < 2r, result < 2r
  Fr a = *io, acc = Fr::one();
  for (int bit = 252; bit >= 0; --bit) {
    fr_mul_ni(&acc, &acc, &acc);
    if ((FR_R_MINUS_2[bit >> 5] >> (bit & 31)) & 1u) fr_mul_ni(&acc, &acc, &a);
  }
  *io = acc;
}

// lane t owns elements t, t + T, t + 2T, ... (T = total lanes)
__global__ void __launch_bounds__(256) k_fr_batch_inverse(char* __restrict__ data, char* __restrict__ prefix, size_t n, size_t T) {
  size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  Fr prod = Fr::one();
  for (size_t i = t; i < n; i += T) {
    store_fp<Fr>(prefix + i * 32, prod);                      // product of this lane's earlier non-zero elements (< 2r)
    Fr v = load_fp<Fr>(data + i * 32);
    if (!v.is_zero_raw()) prod = Fr::mul(prod, v);
  }
  fr_inverse_ni(&prod);
  size_t cnt = (n - t + T - 1) / T;
  for (size_t k = cnt; k-- > 0;) {
    size_t i = t + k * T;
    Fr v = load_fp<Fr>(data + i * 32);
    if (v.is_zero_raw()) continue;                            // batch_inversion leaves zeros in place
    Fr pre = load_fp<Fr>(prefix + i * 32);
    Fr inv = Fr::mul(prod, pre);
    prod = Fr::mul(prod, v);
    store_fp<Fr>(data + i * 32, Fr::reduce(inv));
  }
}

int32_t fr_vec_op(Ctx* c, void* d_dst, const void* d_a, const void* d_b, size_t n, int32_t op, hipStream_t s) {
  (void)c;
  if (n == 0) return ALEO_MI355X_OK;
  size_t want = (n + 255) / 256; uint32_t grid = (uint32_t)(want < 8192 ? want : 8192);
  switch (op) {
    case 0: hipLaunchKernelGGL(k_fr_vec_op<0>, dim3(grid), dim3(256), 0, s, (char*)d_dst, (const char*)d_a, (const char*)d_b, n); break;
    case 1: hipLaunchKernelGGL(k_fr_vec_op<1>, dim3(grid), dim3(256), 0, s, (char*)d_dst, (const char*)d_a, (const char*)d_b, n); break;
    case 2: hipLaunchKernelGGL(k_fr_vec_op<2>, dim3(grid), dim3(256), 0, s, (char*)d_dst, (const char*)d_a, (const char*)d_b, n); break;
    default: g_last_error = "fr_vec_op: unknown op"; return ALEO_MI355X_ERR_BAD_ARG;
  }
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// dst[i] = c0 + c1 * a[i] + c2 * b[i] with host-side constants (Montgomery): the scalar-times-vector, shifted and blended forms
// of the AHP rounds (alpha - h_i before a batch inversion, eta-weighted sums of z_a, z_b, the linear combinations opened at
// beta and gamma) [UPSTREAM-RECALL: snark/varuna/ahp/prover/round_functions/{second,third,fourth}.rs run these as rayon maps].
struct FrK { uint32_t v[8]; };
__device__ __forceinline__ Fr fr_arg(const FrK& k) { Fr r; for (int i = 0; i < 8; ++i) r.v[i] = k.v[i]; return r; }
template <bool HAS_A, bool HAS_B>
__global__ void __launch_bounds__(256) k_fr_lin(char* dst, size_t n, FrK k0, FrK k1, const char* a, FrK k2, const char* b) {
  const Fr c0 = fr_arg(k0), c1 = fr_arg(k1), c2 = fr_arg(k2);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    Fr r = c0;                                                                                    // canonical: < r
    if constexpr (HAS_A) r = Fr::cond_sub<2>(Fr::add(r, Fr::mul(c1, load_fp<Fr>(a + i * 32))));    // < 3r -> < 2r
    if constexpr (HAS_B) r = Fr::cond_sub<2>(Fr::add(r, Fr::mul(c2, load_fp<Fr>(b + i * 32))));    // < 4r -> < 2r
    store_fp<Fr>(dst + i * 32, Fr::cond_sub<1>(r));
  }
}

int32_t fr_lin(Ctx* c, void* d_dst, size_t n, const void* c0, const void* c1, const void* d_a, const void* c2, const void* d_b, hipStream_t s) {
  (void)c;
  if (n == 0) return ALEO_MI355X_OK;
  FrK k0{}, k1{}, k2{};
  if (c0) std::memcpy(k0.v, c0, 32);
  if (d_a) { if (!c1) { g_last_error = "fr_lin: a without c1"; return ALEO_MI355X_ERR_BAD_ARG; } std::memcpy(k1.v, c1, 32); }
  if (d_b) { if (!c2) { g_last_error = "fr_lin: b without c2"; return ALEO_MI355X_ERR_BAD_ARG; } std::memcpy(k2.v, c2, 32); }
  size_t want = (n + 255) / 256; dim3 grid((uint32_t)(want < 8192 ? want : 8192)), blk(256);
  char* dst = (char*)d_dst; const char* a = (const char*)d_a; const char* b = (const char*)d_b;
  if (a && b) hipLaunchKernelGGL((k_fr_lin<true, true>), grid, blk, 0, s, dst, n, k0, k1, a, k2, b);
  else if (a) hipLaunchKernelGGL((k_fr_lin<true, false>), grid, blk, 0, s, dst, n, k0, k1, a, k2, b);
  else if (b) hipLaunchKernelGGL((k_fr_lin<false, true>), grid, blk, 0, s, dst, n, k0, k1, a, k2, b);
  else hipLaunchKernelGGL((k_fr_lin<false, false>), grid, blk, 0, s, dst, n, k0, k1, a, k2, b);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// ---- sparse matrix x vector over Fr: z_M = M * z for the R1CS matrices A, B, C ----------------
// Replaces the per-row inner products snarkVM's Varuna prover runs on rayon before committing z_a, z_b
// (`matrix row: Vec<(F, usize)>` dotted with the assignment [UPSTREAM-RECALL: snark/varuna/ahp/prover/round_functions/first.rs]).
// CSR in HBM: row_ptr u32[rows + 1], col_idx u32[nnz], vals Fr Montgomery [nnz].  Constraint rows are short (a handful
// of entries) with a few very long linear combinations, so: one lane per row for rows of <= SPMV_LANE_MAX entries; longer
// rows are queued (one global atomic each) and a second kernel gives each a whole wave, folding the 64 partial sums
// with shuffles.  68 algorithmic bytes per non-zero (32 value + 4 index + 32 gathered operand) against one product.
static constexpr uint32_t SPMV_LANE_MAX = 64;

__device__ __forceinline__ Fr spmv_term(const char* __restrict__ vals, const uint32_t* __restrict__ col, const char* __restrict__ x, uint32_t k) {
  return Fr::mul(load_fp<Fr>(vals + (size_t)k * 32), load_fp<Fr>(x + (size_t)col[k] * 32));       // < 2r
}

// Rows beyond SPMV_WAVE_MAX entries (the column of the constant 1 in a transposed constraint matrix: a fifth of all constraints of a
// compiled program touch it) would keep ONE wave busy for milliseconds: up to SPMV_HUGE_ROWS of them are spread over the whole grid instead —
// every wave sums a strided share into a partial, one block per row folds the partials.
static constexpr uint32_t SPMV_WAVE_MAX = 8192, SPMV_HUGE_ROWS = 64, SPMV_HUGE_WAVES = 4096;

__global__ void __launch_bounds__(256) k_spmv_rows(char* __restrict__ y, const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                                   const char* __restrict__ vals, const char* __restrict__ x, uint32_t rows,
                                                   uint32_t* __restrict__ long_rows, uint32_t* __restrict__ huge_rows, uint32_t* __restrict__ counters) {
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
    const uint32_t k0 = row_ptr[r], k1 = row_ptr[r + 1];
    if (k1 - k0 > SPMV_LANE_MAX) {
      if (k1 - k0 > SPMV_WAVE_MAX) { const uint32_t h = atomicAdd(counters + 1, 1u); if (h < SPMV_HUGE_ROWS) { huge_rows[h] = r; continue; } }
      long_rows[atomicAdd(counters, 1u)] = r; continue;
    }
    Fr acc = Fr::zero();
    for (uint32_t k = k0; k < k1; ++k) acc = Fr::cond_sub<2>(Fr::add(acc, spmv_term(vals, col, x, k)));     // < 4r -> < 2r
    store_fp<Fr>(y + (size_t)r * 32, Fr::cond_sub<1>(acc));
  }
}

__device__ __forceinline__ Fr spmv_wave_sum(Fr acc) {        // sum over the 64 lanes, result in every lane, < 2r
  for (int d = 32; d >= 1; d >>= 1) {
    Fr o;
#pragma unroll
    for (int l = 0; l < 8; ++l) o.v[l] = __shfl_xor(acc.v[l], d);
    acc = Fr::cond_sub<2>(Fr::add(acc, o));
  }
  return acc;
}

__global__ void __launch_bounds__(256) k_spmv_long(char* __restrict__ y, const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                                   const char* __restrict__ vals, const char* __restrict__ x,
                                                   const uint32_t* __restrict__ long_rows, const uint32_t* __restrict__ counters) {
  const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
  for (uint32_t q = wave; q < counters[0]; q += nwaves) {
    const uint32_t r = long_rows[q], k0 = row_ptr[r], k1 = row_ptr[r + 1];
    Fr acc = Fr::zero();
    for (uint32_t k = k0 + lane; k < k1; k += 64) acc = Fr::cond_sub<2>(Fr::add(acc, spmv_term(vals, col, x, k)));
    acc = spmv_wave_sum(acc);
    if (lane == 0) store_fp<Fr>(y + (size_t)r * 32, Fr::cond_sub<1>(acc));
  }
}
// grid = SPMV_HUGE_WAVES waves; wave w of the grid takes entries w*64 + lane, + SPMV_HUGE_WAVES*64, ... of every huge row
__global__ void __launch_bounds__(256) k_spmv_huge_partial(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, const char* __restrict__ vals,
                                                           const char* __restrict__ x, const uint32_t* __restrict__ huge_rows, const uint32_t* __restrict__ counters,
                                                           char* __restrict__ partial) {
  const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const uint32_t nh = counters[1] < SPMV_HUGE_ROWS ? counters[1] : SPMV_HUGE_ROWS;
  for (uint32_t q = 0; q < nh; ++q) {
    const uint32_t r = huge_rows[q], k0 = row_ptr[r], k1 = row_ptr[r + 1];
    Fr acc = Fr::zero();
    for (uint32_t k = k0 + wave * 64 + lane; k < k1; k += SPMV_HUGE_WAVES * 64) acc = Fr::cond_sub<2>(Fr::add(acc, spmv_term(vals, col, x, k)));
    acc = spmv_wave_sum(acc);
    if (lane == 0) store_fp<Fr>(partial + ((size_t)q * SPMV_HUGE_WAVES + wave) * 32, acc);      // < 2r
  }
}
// one block per huge row: 4096 partials -> y[row]
__global__ void __launch_bounds__(256) k_spmv_huge_fold(char* __restrict__ y, const uint32_t* __restrict__ huge_rows, const uint32_t* __restrict__ counters,
                                                        const char* __restrict__ partial) {
  __shared__ uint32_t l[4][8];
  const uint32_t nh = counters[1] < SPMV_HUGE_ROWS ? counters[1] : SPMV_HUGE_ROWS, q = blockIdx.x;
  if (q >= nh) return;
  Fr acc = Fr::zero();
  for (uint32_t i = threadIdx.x; i < SPMV_HUGE_WAVES; i += 256) acc = Fr::cond_sub<2>(Fr::add(acc, load_fp<Fr>(partial + ((size_t)q * SPMV_HUGE_WAVES + i) * 32)));
  acc = spmv_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) for (int k = 0; k < 8; ++k) l[threadIdx.x >> 6][k] = acc.v[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    Fr t = Fr::zero();
    for (int w = 0; w < 4; ++w) { Fr o; for (int k = 0; k < 8; ++k) o.v[k] = l[w][k]; t = Fr::cond_sub<2>(Fr::add(t, o)); }
    store_fp<Fr>(y + (size_t)huge_rows[q] * 32, Fr::cond_sub<1>(t));
  }
}

// max_row: an upper bound of the longest row when the caller knows one (the prover's index does), 0 = unknown.  The kernels for long and huge rows are
// launched only when such rows can exist — at the sizes of real circuits an empty launch still costs ~4.5 us of GPU time, and a proof runs 3k + 1 products.
int32_t fr_spmv(Ctx* c, void* d_y, const void* d_row_ptr, const void* d_col, const void* d_vals, const void* d_x, size_t rows, hipStream_t s, size_t max_row) {
  if (rows == 0) return ALEO_MI355X_OK;
  if (rows >= (1ull << 32)) { g_last_error = "fr_spmv: row count exceeds 2^32"; return ALEO_MI355X_ERR_BAD_ARG; }
  const bool may_long = max_row == 0 || max_row > SPMV_LANE_MAX, may_huge = max_row == 0 || max_row > SPMV_WAVE_MAX;
  const size_t head = ((rows + 16 + SPMV_HUGE_ROWS) * 4 + 255) & ~(size_t)255;
  int32_t rc; if ((rc = scratch_acquire(c, c->ntt_tmp, head + (size_t)SPMV_HUGE_ROWS * SPMV_HUGE_WAVES * 32, s))) return rc;
  uint32_t* counters = c->ntt_tmp.as<uint32_t>(); uint32_t* huge_rows = counters + 16; uint32_t* long_rows = huge_rows + SPMV_HUGE_ROWS;
  char* partial = c->ntt_tmp.as<char>() + head;
  if (may_long) HIPCHK(hipMemsetAsync(counters, 0, 8, s));
  const size_t want = (rows + 255) / 256; const uint32_t grid = (uint32_t)(want < 16384 ? want : 16384);
  const uint32_t* rp = (const uint32_t*)d_row_ptr; const uint32_t* cl = (const uint32_t*)d_col; const char* vl = (const char*)d_vals; const char* xx = (const char*)d_x;
  hipLaunchKernelGGL(k_spmv_rows, dim3(grid), dim3(256), 0, s, (char*)d_y, rp, cl, vl, xx, (uint32_t)rows, long_rows, huge_rows, counters);
  if (may_long) hipLaunchKernelGGL(k_spmv_long, dim3(1024), dim3(256), 0, s, (char*)d_y, rp, cl, vl, xx, long_rows, counters);
  if (may_huge) {
    hipLaunchKernelGGL(k_spmv_huge_partial, dim3(SPMV_HUGE_WAVES / 4), dim3(256), 0, s, rp, cl, vl, xx, huge_rows, counters, partial);
    hipLaunchKernelGGL(k_spmv_huge_fold, dim3(SPMV_HUGE_ROWS), dim3(256), 0, s, (char*)d_y, huge_rows, counters, (const char*)partial);
  }
  HIPCHK(hipGetLastError());
  return scratch_release(c, s);
}

// ---- division by (X - z): the witness polynomial of a KZG10 opening ------------------------------------------------------
// Replaces snarkVM 0.14.5 algorithms/src/polycommit/kzg10  KZG10::compute_witness_polynomial / open  [UPSTREAM-RECALL]:
// w(X) = (p(X) - p(z)) / (X - z), followed by the commitment to w (SURVEY.md §2c polycommit row; the two opening MSMs of row a6).
// Synthetic division is the recurrence s_j = p_j + z * s_(j+1) (s_n = 0): w_(j-1) = s_j and p(z) = s_0.  A linear recurrence
// with a constant multiplier is a suffix scan, done in three levels with the multipliers z^16, (z^16)^256, ...:
//   k_div_blocks   every lane folds its 16 coefficients (Horner), every 256-lane block folds its lanes -> E_b
//   k_div_carries  one block: carry C_b into every block = tail of the polynomial behind it, from the E_b
//   k_div_finish   every block scans its lanes' values seeded with C_b, every lane replays its 16 steps and writes w
// 96 algorithmic bytes per coefficient (p read twice, w written once); values stay lazily reduced below 4r (fp.h).
static constexpr uint32_t DIV_K = 16, DIV_B = 256, DIV_TILE = DIV_K * DIV_B;
__device__ __forceinline__ Fr fr_lt2r(const Fr& a) { return Fr::cond_sub<2>(a); }                    // < 4r -> < 2r
__device__ __forceinline__ void lds_put(uint32_t* l, uint32_t t, const Fr& a) { for (int i = 0; i < 8; ++i) l[i * DIV_B + t] = a.v[i]; }
__device__ __forceinline__ Fr lds_get(const uint32_t* l, uint32_t t) { Fr r; for (int i = 0; i < 8; ++i) r.v[i] = l[i * DIV_B + t]; return r; }
// value of the lane's 16 coefficients at z: sum_k p[lo + k] z^k (coefficients past n are zero); result < 3r
__device__ __forceinline__ Fr div_lane_fold(const char* __restrict__ p, size_t lo, size_t n, const Fr& z) {
  Fr h = Fr::zero();
  for (int k = (int)DIV_K - 1; k >= 0; --k) {
    const size_t i = lo + (size_t)k;
    Fr m = Fr::mul(h, z);                                   // < 2r
    h = i < n ? Fr::add(m, load_fp<Fr>(p + i * 32)) : m;    // canonical coefficient: < 3r
  }
  return h;
}
__device__ __forceinline__ void div_blocks_body(uint32_t* l, uint32_t blk, const char* __restrict__ p, size_t n, const FrK& zk, char* __restrict__ E) {
  const uint32_t t = threadIdx.x; const Fr z = fr_arg(zk);
  Fr x = fr_lt2r(div_lane_fold(p, ((size_t)blk * DIV_B + t) * DIV_K, n, z));
  Fr w = z; for (int i = 0; i < 4; ++i) w = Fr::sqr(w);     // Z = z^16, < 2r
  for (uint32_t d = 1; d < DIV_B; d <<= 1) {               // x_t += Z^d * x_(t+d) on the lanes that are multiples of 2d
    lds_put(l, t, x); __syncthreads();
    if ((t & (2 * d - 1)) == 0) x = fr_lt2r(Fr::add(x, Fr::mul(lds_get(l, t + d), w)));
    __syncthreads();
    w = Fr::sqr(w);
  }
  if (t == 0) store_fp<Fr>(E + (size_t)blk * 32, x);
}
__global__ void __launch_bounds__(256) k_div_blocks(const char* __restrict__ p, size_t n, FrK zk, char* __restrict__ E) {
  __shared__ uint32_t l[8 * DIV_B];
  div_blocks_body(l, blockIdx.x, p, n, zk, E);
}
// One block.  C[b] = sum_(u > b) E[u] * ZB^(u - b - 1), ZB = z^(16 * 256): lane q owns `per` consecutive blocks.
__device__ __forceinline__ void div_carries_body(uint32_t* l, const char* __restrict__ E, uint32_t nb, uint32_t per, const FrK& zk, char* __restrict__ C) {
  const uint32_t t = threadIdx.x; Fr zb = fr_arg(zk);
  for (int i = 0; i < 12; ++i) zb = Fr::sqr(zb);            // z^4096, < 2r
  const uint32_t lo = t * per, hi = lo + per < nb ? lo + per : nb;
  Fr g = Fr::zero();
  for (uint32_t b = hi; b-- > lo;) g = fr_lt2r(Fr::add(Fr::mul(g, zb), load_fp<Fr>(E + (size_t)b * 32)));
  Fr w = Fr::one();                                         // ZB^per by square-and-multiply
  for (int bit = 31 - __clz(per | 1u); bit >= 0; --bit) { w = Fr::sqr(w); if ((per >> bit) & 1u) w = Fr::mul(w, zb); }
  Fr x = g;
  for (uint32_t d = 1; d < DIV_B; d <<= 1) {               // suffix scan over the 256 lanes (Hillis-Steele): x_t += W^d * x_(t+d)
    lds_put(l, t, x); __syncthreads();
    if (t + d < DIV_B) x = fr_lt2r(Fr::add(x, Fr::mul(lds_get(l, t + d), w)));
    __syncthreads();
    w = Fr::sqr(w);
  }
  lds_put(l, t, x); __syncthreads();
  Fr c = t + 1 < DIV_B ? lds_get(l, t + 1) : Fr::zero();    // everything behind this lane's blocks
  for (uint32_t b = hi; b-- > lo;) {
    store_fp<Fr>(C + (size_t)b * 32, c);
    c = fr_lt2r(Fr::add(Fr::mul(c, zb), load_fp<Fr>(E + (size_t)b * 32)));
  }
}
__global__ void __launch_bounds__(256) k_div_carries(const char* __restrict__ E, uint32_t nb, uint32_t per, FrK zk, char* __restrict__ C) {
  __shared__ uint32_t l[8 * DIV_B];
  div_carries_body(l, E, nb, per, zk, C);
}
__device__ __forceinline__ void div_finish_body(uint32_t* l, uint32_t blk, const char* __restrict__ p, size_t n, const FrK& zk, const char* __restrict__ C, char* __restrict__ q, char* __restrict__ eval) {
  const uint32_t t = threadIdx.x; const Fr z = fr_arg(zk);
  const size_t lo = ((size_t)blk * DIV_B + t) * DIV_K;
  Fr x = fr_lt2r(div_lane_fold(p, lo, n, z));
  Fr w = z; for (int i = 0; i < 4; ++i) w = Fr::sqr(w);     // Z = z^16
  if (t == DIV_B - 1) x = fr_lt2r(Fr::add(x, Fr::mul(load_fp<Fr>(C + (size_t)blk * 32), w)));     // the block's carry enters behind its last lane
  for (uint32_t d = 1; d < DIV_B; d <<= 1) {
    lds_put(l, t, x); __syncthreads();
    if (t + d < DIV_B) x = fr_lt2r(Fr::add(x, Fr::mul(lds_get(l, t + d), w)));
    __syncthreads();
    w = Fr::sqr(w);
  }
  lds_put(l, t, x); __syncthreads();
  Fr s = t + 1 < DIV_B ? lds_get(l, t + 1) : load_fp<Fr>(C + (size_t)blk * 32);      // s_(lo + 16)
  for (int k = (int)DIV_K - 1; k >= 0; --k) {              // s_j = p_j + z s_(j+1);  w_(j-1) = s_j;  p(z) = s_0
    const size_t j = lo + (size_t)k;
    if (j >= n) continue;
    s = Fr::add(Fr::mul(s, z), load_fp<Fr>(p + j * 32));    // < 3r
    if (j) { if (q) store_fp<Fr>(q + (j - 1) * 32, Fr::reduce(s)); } else if (eval) store_fp<Fr>(eval, Fr::reduce(s));
  }
}
__global__ void __launch_bounds__(256) k_div_finish(const char* __restrict__ p, size_t n, FrK zk, const char* __restrict__ C, char* __restrict__ q, char* __restrict__ eval) {
  __shared__ uint32_t l[8 * DIV_B];
  div_finish_body(l, blockIdx.x, p, n, zk, C, q, eval);
}
// Up to DIV_MANY independent divisions in the same three launches (blockIdx.y picks the division): the two witness polynomials of a proof's openings are
// latency-bound chains of ~100 us each (the middle kernel is ONE block) that used to run one after the other.
static constexpr uint32_t DIV_MANY = 4;
struct DivJob { const char* p; size_t n; FrK z; char* q; char* eval; char* E; char* C; uint32_t nb, per; };
struct DivJobs { DivJob j[DIV_MANY]; };
__global__ void __launch_bounds__(256) k_div_blocks_many(DivJobs J) {
  __shared__ uint32_t l[8 * DIV_B];
  const DivJob& d = J.j[blockIdx.y];
  if (blockIdx.x >= d.nb) return;
  div_blocks_body(l, blockIdx.x, d.p, d.n, d.z, d.E);
}
__global__ void __launch_bounds__(256) k_div_carries_many(DivJobs J) {
  __shared__ uint32_t l[8 * DIV_B];
  const DivJob& d = J.j[blockIdx.x];
  div_carries_body(l, d.E, d.nb, d.per, d.z, d.C);
}
__global__ void __launch_bounds__(256) k_div_finish_many(DivJobs J) {
  __shared__ uint32_t l[8 * DIV_B];
  const DivJob& d = J.j[blockIdx.y];
  if (blockIdx.x >= d.nb) return;
  div_finish_body(l, blockIdx.x, d.p, d.n, d.z, d.C, d.q, d.eval);
}

int32_t fr_divide_by_linear(Ctx* c, void* d_q, void* d_eval, const void* d_p, size_t n, const void* z_mont32, hipStream_t s) {
  if (n == 0) { if (d_eval) HIPCHK(hipMemsetAsync(d_eval, 0, 32, s)); return ALEO_MI355X_OK; }
  const size_t nb = (n + DIV_TILE - 1) / DIV_TILE;
  if (nb >= (1ull << 31)) { g_last_error = "fr_divide_by_linear: polynomial too long"; return ALEO_MI355X_ERR_BAD_ARG; }
  int32_t rc; if ((rc = scratch_acquire(c, c->ntt_tmp, 2 * nb * 32, s))) return rc;
  char* E = c->ntt_tmp.as<char>(); char* C = E + nb * 32;
  FrK zk; std::memcpy(zk.v, z_mont32, 32);
  const uint32_t per = (uint32_t)((nb + DIV_B - 1) / DIV_B);
  hipLaunchKernelGGL(k_div_blocks, dim3((uint32_t)nb), dim3(256), 0, s, (const char*)d_p, n, zk, E);
  hipLaunchKernelGGL(k_div_carries, dim3(1), dim3(256), 0, s, (const char*)E, (uint32_t)nb, per, zk, C);
  hipLaunchKernelGGL(k_div_finish, dim3((uint32_t)nb), dim3(256), 0, s, (const char*)d_p, n, zk, (const char*)C, (char*)d_q, (char*)d_eval);
  HIPCHK(hipGetLastError());
  return scratch_release(c, s);
}

int32_t fr_divide_by_linear_many(Ctx* c, void* const* d_q, void* const* d_eval, const void* const* d_p, const size_t* n, const void* const* z_mont32, size_t count, hipStream_t s) {
  if (count == 0) return ALEO_MI355X_OK;
  if (count > DIV_MANY) { g_last_error = "fr_divide_by_linear_many: at most 4 divisions per call"; return ALEO_MI355X_ERR_BAD_ARG; }
  DivJobs J{}; size_t tot = 0, nb_max = 0;
  for (size_t i = 0; i < count; ++i) {
    if (n[i] == 0) { g_last_error = "fr_divide_by_linear_many: empty polynomial"; return ALEO_MI355X_ERR_BAD_ARG; }
    const size_t nb = (n[i] + DIV_TILE - 1) / DIV_TILE;
    if (nb >= (1ull << 31)) { g_last_error = "fr_divide_by_linear: polynomial too long"; return ALEO_MI355X_ERR_BAD_ARG; }
    J.j[i].nb = (uint32_t)nb; J.j[i].per = (uint32_t)((nb + DIV_B - 1) / DIV_B); tot += 2 * nb; nb_max = nb > nb_max ? nb : nb_max;
  }
  int32_t rc; if ((rc = scratch_acquire(c, c->ntt_tmp, tot * 32, s))) return rc;
  char* at = c->ntt_tmp.as<char>();
  for (size_t i = 0; i < count; ++i) {
    DivJob& d = J.j[i]; d.p = (const char*)d_p[i]; d.n = n[i]; std::memcpy(d.z.v, z_mont32[i], 32); d.q = (char*)d_q[i]; d.eval = d_eval ? (char*)d_eval[i] : nullptr;
    d.E = at; d.C = at + (size_t)d.nb * 32; at += 2 * (size_t)d.nb * 32;
  }
  hipLaunchKernelGGL(k_div_blocks_many, dim3((uint32_t)nb_max, (uint32_t)count), dim3(256), 0, s, J);
  hipLaunchKernelGGL(k_div_carries_many, dim3((uint32_t)count), dim3(256), 0, s, J);
  hipLaunchKernelGGL(k_div_finish_many, dim3((uint32_t)nb_max, (uint32_t)count), dim3(256), 0, s, J);
  HIPCHK(hipGetLastError());
  return scratch_release(c, s);
}

// ---- small fused kernels of the prover's rounds (round 5: every launch of a real-circuit-sized proof is ~4.5 us of GPU time even when it moves a few KB) ----
// dst[r][i] = k_r * src[i], r < rows <= 4 (the eta-scaled copies of u_H(alpha, .) the transpose product reads)
struct FrK4 { FrK k[4]; };
__global__ void __launch_bounds__(256) k_fr_scale_rows(char* __restrict__ dst, const char* __restrict__ src, size_t n, uint32_t rows, FrK4 K) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const Fr x = load_fp<Fr>(src + i * 32);
    for (uint32_t r = 0; r < rows; ++r) store_fp<Fr>(dst + ((size_t)r * n + i) * 32, Fr::reduce(Fr::mul(fr_arg(K.k[r]), x)));
  }
}
int32_t fr_scale_rows(Ctx* c, void* d_dst, const void* d_src, size_t n, size_t rows, const void* consts_mont, hipStream_t s) {
  (void)c;
  if (n == 0 || rows == 0) return ALEO_MI355X_OK;
  if (rows > 4) { g_last_error = "fr_scale_rows: at most 4 rows"; return ALEO_MI355X_ERR_BAD_ARG; }
  FrK4 K{}; std::memcpy(K.k, consts_mont, rows * 32);
  const size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(k_fr_scale_rows, dim3((uint32_t)(want < 8192 ? want : 8192)), dim3(256), 0, s, (char*)d_dst, (const char*)d_src, n, (uint32_t)rows, K);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}
// The quotient and remainder of q (3n coefficients, blocks p0 | p1 | p2; + mask when given) by X^n - 1:  hq = [p1 + p2 | p2],  rq = p0 + p1 + p2;
// rq[0] — the sum of the first sumcheck's summand over H — also goes to `host_sum` (a device pointer of pinned host memory) when given.
__global__ void __launch_bounds__(256) k_split_quotient(const char* __restrict__ q, const char* __restrict__ mask, size_t n, char* __restrict__ hq, char* __restrict__ rq, char* __restrict__ host_sum) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    Fr p0 = load_fp<Fr>(q + i * 32), p1 = load_fp<Fr>(q + (n + i) * 32), p2 = load_fp<Fr>(q + (2 * n + i) * 32);
    if (mask) {
      p0 = Fr::cond_sub<1>(Fr::add(p0, load_fp<Fr>(mask + i * 32))); p1 = Fr::cond_sub<1>(Fr::add(p1, load_fp<Fr>(mask + (n + i) * 32)));
      p2 = Fr::cond_sub<1>(Fr::add(p2, load_fp<Fr>(mask + (2 * n + i) * 32)));
    }
    const Fr h0 = Fr::cond_sub<1>(Fr::add(p1, p2)), r0 = Fr::cond_sub<1>(Fr::add(p0, h0));
    store_fp<Fr>(hq + (n + i) * 32, p2); store_fp<Fr>(hq + i * 32, h0); store_fp<Fr>(rq + i * 32, r0);
    if (i == 0 && host_sum) store_fp<Fr>(host_sum, r0);
  }
}
int32_t fr_split_quotient(Ctx* c, void* d_hq, void* d_rq, const void* d_q, const void* d_mask, size_t n, void* host_sum_devptr, hipStream_t s) {
  (void)c;
  if (n == 0) return ALEO_MI355X_OK;
  const size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(k_split_quotient, dim3((uint32_t)(want < 8192 ? want : 8192)), dim3(256), 0, s, (const char*)d_q, (const char*)d_mask, n, (char*)d_hq, (char*)d_rq, (char*)host_sum_devptr);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}
// dst = (a - b) * m element-wise (the witness polynomial's values: (z - x̂) / v_X off X)
__global__ void __launch_bounds__(256) k_fr_sub_mul(char* __restrict__ dst, const char* __restrict__ a, const char* __restrict__ b, const char* __restrict__ m, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const Fr d = Fr::cond_sub<1>(Fr::sub<1>(load_fp<Fr>(a + i * 32), load_fp<Fr>(b + i * 32)));
    store_fp<Fr>(dst + i * 32, Fr::cond_sub<1>(Fr::mul(d, load_fp<Fr>(m + i * 32))));
  }
}
int32_t fr_sub_mul(Ctx* c, void* d_dst, const void* d_a, const void* d_b, const void* d_m, size_t n, hipStream_t s) {
  (void)c;
  if (n == 0) return ALEO_MI355X_OK;
  const size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(k_fr_sub_mul, dim3((uint32_t)(want < 8192 ? want : 8192)), dim3(256), 0, s, (char*)d_dst, (const char*)d_a, (const char*)d_b, (const char*)d_m, n);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}
// up to 8 single elements gathered into consecutive 32-byte slots of dst (device memory, or the device pointer of pinned host memory: a read-back without a copy launch each)
struct PickArgs { const char* src[8]; };
__global__ void k_fr_pick(char* __restrict__ dst, PickArgs a, uint32_t count) {
  const uint32_t t = threadIdx.x;
  if (t < count * 8) ((uint32_t*)dst)[t] = ((const uint32_t*)a.src[t >> 3])[t & 7];
}
int32_t fr_pick(Ctx* c, void* dst_devptr, const void* const* d_src, size_t count, hipStream_t s) {
  (void)c;
  if (count == 0) return ALEO_MI355X_OK;
  if (count > 8) { g_last_error = "fr_pick: at most 8 elements"; return ALEO_MI355X_ERR_BAD_ARG; }
  PickArgs a{}; for (size_t i = 0; i < count; ++i) a.src[i] = (const char*)d_src[i];
  hipLaunchKernelGGL(k_fr_pick, dim3(1), dim3(64), 0, s, (char*)dst_devptr, a, (uint32_t)count);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// ---- geometric sequences, gathers and batched evaluation: what lets the AHP rounds run without a single field inversion on the device ----
// u_H(a, X) = (v_H(a) - v_H(X)) / (a - X) = sum_k a^(|H|-1-k) X^k, so the polynomial r(alpha, X) IS the reversed powers of alpha and its
// values v_H(alpha) / (alpha - h) over H are one NTT of them; the third round's 1 / ((alpha - row)(beta - col)) are then two gathers from
// those tables by the row / column index of every non-zero entry [UPSTREAM-RECALL: snarkVM gets the same values through batch inversions
// in round_functions/{second,third}.rs].
static constexpr uint32_t POW_K = 16;
__device__ __noinline__ void fr_pow_u64_ni(Fr* io, uint64_t e) {          // io^e, io < 2r
  Fr b = *io, acc = Fr::one();
  for (; e; e >>= 1) { if (e & 1) acc = Fr::mul(acc, b); b = Fr::sqr(b); }
  *io = acc;
}
// dst[k] = first * ratio^k: every lane raises ratio to its first index, then walks POW_K consecutive elements
__global__ void __launch_bounds__(256) k_fr_powers(char* __restrict__ dst, size_t n, FrK kfirst, FrK kratio) {
  const size_t lo = ((size_t)blockIdx.x * 256 + threadIdx.x) * POW_K;
  if (lo >= n) return;
  const Fr ratio = fr_arg(kratio);
  Fr x = ratio; fr_pow_u64_ni(&x, lo);
  x = Fr::mul(x, fr_arg(kfirst));
  for (uint32_t k = 0; k < POW_K && lo + k < n; ++k) {
    store_fp<Fr>(dst + (lo + k) * 32, Fr::reduce(x));
    x = Fr::mul(x, ratio);
  }
}
// The latency form for sequences of up to 2^26 elements (the prover's r(alpha, X), r(beta, X): 2^15 elements took 29 us as ~30 dependent products per lane for
// the lane's starting power + 16 for its walk): the squarings ratio^(4 2^i) come from the host (28 constants by value), a lane multiplies the ones its index
// selects (no squaring on the device: ~lg(n) / 2 products) and walks 4 elements.
static constexpr uint32_t POW_KS = 4, POW_SQ = 24;
struct FrSq { FrK v[POW_SQ]; };
__global__ void __launch_bounds__(256) k_fr_powers_fast(char* __restrict__ dst, uint32_t n, FrK kfirst, FrK kratio, FrSq sq) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x, lo = t * POW_KS;
  if (lo >= n) return;
  Fr x = fr_arg(kfirst);
  for (uint32_t i = 0, b = t; b; ++i, b >>= 1) if (b & 1u) x = Fr::mul(x, fr_arg(sq.v[i]));      // first * ratio^(4 t)
  const Fr ratio = fr_arg(kratio);
  for (uint32_t k = 0; k < POW_KS && lo + k < n; ++k) {
    store_fp<Fr>(dst + (size_t)(lo + k) * 32, Fr::reduce(x));
    x = Fr::mul(x, ratio);
  }
}
int32_t fr_powers(Ctx* c, void* d_dst, size_t n, const void* first, const void* ratio, hipStream_t s) {
  (void)c;
  if (n == 0) return ALEO_MI355X_OK;
  FrK kf, kr; std::memcpy(kf.v, first, 32); std::memcpy(kr.v, ratio, 32);
  if (n <= ((size_t)POW_KS << POW_SQ)) {
    FrSq sq; host::HFr r; std::memcpy(r.l, ratio, 32);
    r = host::HFr::sqr(host::HFr::sqr(r));                   // ratio^4
    for (uint32_t i = 0; i < POW_SQ; ++i) { std::memcpy(sq.v[i].v, r.l, 32); r = host::HFr::sqr(r); }
    const size_t lanes = (n + POW_KS - 1) / POW_KS;
    hipLaunchKernelGGL(k_fr_powers_fast, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, s, (char*)d_dst, (uint32_t)n, kf, kr, sq);
    HIPCHK(hipGetLastError());
    return ALEO_MI355X_OK;
  }
  const size_t lanes = (n + POW_K - 1) / POW_K;
  hipLaunchKernelGGL(k_fr_powers, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, s, (char*)d_dst, n, kf, kr);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// dst[i] = scale[i] * t1[idx1[i]] * t2[idx2[i]]   (scale and the second table optional)
template <bool HAS_SCALE, bool HAS_T2>
__global__ void __launch_bounds__(256) k_fr_gather_mul(char* __restrict__ dst, size_t n, const char* scale, const char* __restrict__ t1,
                                                       const uint32_t* __restrict__ idx1, const char* __restrict__ t2, const uint32_t* __restrict__ idx2) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    Fr r = load_fp<Fr>(t1 + (size_t)idx1[i] * 32);
    if constexpr (HAS_T2) r = Fr::mul(r, load_fp<Fr>(t2 + (size_t)idx2[i] * 32));
    if constexpr (HAS_SCALE) r = Fr::mul(r, load_fp<Fr>(scale + i * 32));
    store_fp<Fr>(dst + i * 32, Fr::reduce(r));
  }
}
int32_t fr_gather_mul(Ctx* c, void* d_dst, size_t n, const void* d_scale, const void* d_t1, const void* d_idx1, const void* d_t2, const void* d_idx2, hipStream_t s) {
  (void)c;
  if (n == 0) return ALEO_MI355X_OK;
  size_t want = (n + 255) / 256; dim3 grid((uint32_t)(want < 16384 ? want : 16384)), blk(256);
  char* dst = (char*)d_dst; const char* sc = (const char*)d_scale; const char* t1 = (const char*)d_t1; const char* t2 = (const char*)d_t2;
  const uint32_t* i1 = (const uint32_t*)d_idx1; const uint32_t* i2 = (const uint32_t*)d_idx2;
  if (sc && t2) hipLaunchKernelGGL((k_fr_gather_mul<true, true>), grid, blk, 0, s, dst, n, sc, t1, i1, t2, i2);
  else if (sc) hipLaunchKernelGGL((k_fr_gather_mul<true, false>), grid, blk, 0, s, dst, n, sc, t1, i1, t2, i2);
  else if (t2) hipLaunchKernelGGL((k_fr_gather_mul<false, true>), grid, blk, 0, s, dst, n, sc, t1, i1, t2, i2);
  else hipLaunchKernelGGL((k_fr_gather_mul<false, false>), grid, blk, 0, s, dst, n, sc, t1, i1, t2, i2);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// The same for up to three index ranges in ONE launch (the third round's f_A, f_B, f_C: three short launches were three times the ~4.5 us launch floor)
struct GatherMul3 { char* dst[3]; const char* scale[3]; const uint32_t* idx1[3]; const uint32_t* idx2[3]; uint32_t n[3]; uint32_t cnt; };
__global__ void __launch_bounds__(256) k_fr_gather_mul3(GatherMul3 a, const char* __restrict__ t1, const char* __restrict__ t2) {
  const uint32_t total = a.n[0] + a.n[1] + a.n[2];
  for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
    const uint32_t m = g < a.n[0] ? 0u : (g < a.n[0] + a.n[1] ? 1u : 2u); const size_t i = g - (m == 0 ? 0u : (m == 1 ? a.n[0] : a.n[0] + a.n[1]));
    Fr r = Fr::mul(load_fp<Fr>(t1 + (size_t)a.idx1[m][i] * 32), load_fp<Fr>(t2 + (size_t)a.idx2[m][i] * 32));
    r = Fr::mul(r, load_fp<Fr>(a.scale[m] + i * 32));
    store_fp<Fr>(a.dst[m] + i * 32, Fr::reduce(r));
  }
}
int32_t fr_gather_mul3(Ctx* c, void* const* d_dst, const size_t* n, const void* const* d_scale, const void* d_t1, const void* const* d_idx1, const void* d_t2, const void* const* d_idx2, uint32_t count, hipStream_t s) {
  (void)c;
  if (count == 0 || count > 3) { g_last_error = "fr_gather_mul3: one to three ranges"; return ALEO_MI355X_ERR_BAD_ARG; }
  GatherMul3 a{}; a.cnt = count; size_t total = 0;
  for (uint32_t m = 0; m < count; ++m) {
    if (n[m] >= (1ull << 31)) { g_last_error = "fr_gather_mul3: range too long"; return ALEO_MI355X_ERR_BAD_ARG; }
    a.dst[m] = (char*)d_dst[m]; a.scale[m] = (const char*)d_scale[m]; a.idx1[m] = (const uint32_t*)d_idx1[m]; a.idx2[m] = (const uint32_t*)d_idx2[m]; a.n[m] = (uint32_t)n[m]; total += n[m];
  }
  if (total == 0) return ALEO_MI355X_OK;
  if (total >= (1ull << 32)) { g_last_error = "fr_gather_mul3: ranges too long"; return ALEO_MI355X_ERR_BAD_ARG; }
  const size_t want = (total + 255) / 256;
  hipLaunchKernelGGL(k_fr_gather_mul3, dim3((uint32_t)(want < 16384 ? want : 16384)), dim3(256), 0, s, a, (const char*)d_t1, (const char*)d_t2);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// p_q(z_q) for up to EVAL_MAX (12) polynomials in two launches: the per-block folds of k_div_blocks for all of them at once, then one
// block per polynomial combines its folds (Horner over its lanes' runs of blocks, then a sum over the lanes weighted by powers of z^4096).
static constexpr uint32_t EVAL_MAX = 12;
struct EvalArgs { const char* p[EVAL_MAX]; size_t n[EVAL_MAX]; FrK z[EVAL_MAX]; uint32_t first_block[EVAL_MAX + 1]; uint32_t k; };
__global__ void __launch_bounds__(256) k_eval_blocks(EvalArgs a, char* __restrict__ E) {
  __shared__ uint32_t l[8 * DIV_B];
  uint32_t q = 0; while (q + 1 < a.k && blockIdx.x >= a.first_block[q + 1]) ++q;
  const uint32_t t = threadIdx.x, b = blockIdx.x - a.first_block[q]; const Fr z = fr_arg(a.z[q]);
  Fr x = fr_lt2r(div_lane_fold(a.p[q], ((size_t)b * DIV_B + t) * DIV_K, a.n[q], z));
  Fr w = z; for (int i = 0; i < 4; ++i) w = Fr::sqr(w);
  for (uint32_t d = 1; d < DIV_B; d <<= 1) {
    lds_put(l, t, x); __syncthreads();
    if ((t & (2 * d - 1)) == 0) x = fr_lt2r(Fr::add(x, Fr::mul(lds_get(l, t + d), w)));
    __syncthreads();
    w = Fr::sqr(w);
  }
  if (t == 0) store_fp<Fr>(E + (size_t)blockIdx.x * 32, x);
}
__global__ void __launch_bounds__(256) k_eval_combine(EvalArgs a, const char* __restrict__ E, char* __restrict__ out) {
  __shared__ uint32_t l[8 * DIV_B];
  const uint32_t q = blockIdx.x, t = threadIdx.x, nb = a.first_block[q + 1] - a.first_block[q];
  const char* Eq = E + (size_t)a.first_block[q] * 32;
  Fr zb = fr_arg(a.z[q]); for (int i = 0; i < 12; ++i) zb = Fr::sqr(zb);                 // z^4096
  const uint32_t per = (nb + DIV_B - 1) / DIV_B, lo = t * per, hi = lo + per < nb ? lo + per : nb;
  Fr g = Fr::zero();
  for (uint32_t b = hi; b-- > lo && hi > lo;) g = fr_lt2r(Fr::add(Fr::mul(g, zb), load_fp<Fr>(Eq + (size_t)b * 32)));
  Fr w = zb; fr_pow_u64_ni(&w, (uint64_t)lo);                                             // weight of this lane's run
  Fr x = lo < nb ? Fr::cond_sub<1>(Fr::mul(g, w)) : Fr::zero();                            // < r
  for (uint32_t d = DIV_B / 2; d >= 1; d >>= 1) {
    lds_put(l, t, x); __syncthreads();
    if (t < d) x = Fr::cond_sub<1>(Fr::add(x, lds_get(l, t + d)));                          // < 2r -> < r
    __syncthreads();
  }
  if (t == 0) store_fp<Fr>(out + (size_t)q * 32, x);
}
int32_t fr_eval_batch(Ctx* c, void* d_out, const void* const* d_polys, const size_t* lens, const void* z_mont, size_t k, hipStream_t s) {
  if (k == 0) return ALEO_MI355X_OK;
  if (k > EVAL_MAX) { g_last_error = "fr_eval_batch: more than 12 polynomials in one call"; return ALEO_MI355X_ERR_BAD_ARG; }
  EvalArgs a{}; a.k = (uint32_t)k; uint64_t total = 0;
  for (size_t q = 0; q < k; ++q) {
    if (!d_polys[q] && lens[q]) { g_last_error = "fr_eval_batch: null polynomial"; return ALEO_MI355X_ERR_BAD_ARG; }
    a.p[q] = (const char*)d_polys[q]; a.n[q] = lens[q]; std::memcpy(a.z[q].v, (const char*)z_mont + 32 * q, 32);
    a.first_block[q] = (uint32_t)total; total += (lens[q] + DIV_TILE - 1) / DIV_TILE;     // an empty polynomial owns no block: its value is 0
    if (total >= (1ull << 31)) { g_last_error = "fr_eval_batch: polynomials too long"; return ALEO_MI355X_ERR_BAD_ARG; }
  }
  a.first_block[k] = (uint32_t)total;
  int32_t rc; if ((rc = scratch_acquire(c, c->ntt_tmp, (total + 1) * 32, s))) return rc;
  char* E = c->ntt_tmp.as<char>();
  if (total) hipLaunchKernelGGL(k_eval_blocks, dim3((uint32_t)total), dim3(256), 0, s, a, E);
  hipLaunchKernelGGL(k_eval_combine, dim3((uint32_t)k), dim3(256), 0, s, a, (const char*)E, (char*)d_out);
  HIPCHK(hipGetLastError());
  return scratch_release(c, s);
}

// ---- prover randomness generated where it is used ------------------------------------------------------------------------------
// Element i of the stream under the proof's 32-byte seed: ChaCha20 in counter mode with rejection sampling below r (chacha.h).
// A counter-based definition: any element can be produced anywhere (the host draws the handful of blinding scalars it needs, the device
// the 3|H| mask coefficients) [UPSTREAM-RECALL: snarkVM draws Fr::rand from the caller's CSPRNG in prover/round_functions/first.rs].
__global__ void __launch_bounds__(256) k_fr_random(char* __restrict__ dst, size_t n, Seed32 seed, uint64_t first, int mont) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    Fr v; chacha_fr(v.v, seed.w, first + i);                    // chacha.h: one ChaCha20 block per attempt, accepted with probability 0.83
    if (mont) v = Fr::to_mont(v);
    store_fp<Fr>(dst + i * 32, v);
  }
}
int32_t fr_random(Ctx* c, void* d_dst, size_t n, const uint8_t* seed32, uint64_t first, int32_t mont, hipStream_t s) {
  (void)c;
  if (n == 0) return ALEO_MI355X_OK;
  Seed32 sd; std::memcpy(sd.w, seed32, 32);
  size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(k_fr_random, dim3((uint32_t)(want < 8192 ? want : 8192)), dim3(256), 0, s, (char*)d_dst, n, sd, first, mont ? 1 : 0);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// dst[i] = c0 [i == 0] + sum_j coeff_j * term_j[i] over up to LC_MAX ragged terms (term j ends at len_j): the linear combinations a
// proof opens at beta and gamma, and the delta-weighted sum of the fourth round, in one pass over the data.
static constexpr uint32_t LC_MAX = 28;
struct LcArgs { const char* p[LC_MAX]; size_t n[LC_MAX]; FrK k[LC_MAX]; uint32_t terms; };
__global__ void __launch_bounds__(256) k_fr_lincomb(char* dst, size_t n, LcArgs a, FrK k0) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    Fr r = i == 0 ? fr_arg(k0) : Fr::zero();
    for (uint32_t j = 0; j < a.terms; ++j)
      if (i < a.n[j]) r = Fr::cond_sub<2>(Fr::add(r, Fr::mul(fr_arg(a.k[j]), load_fp<Fr>(a.p[j] + i * 32))));      // < 4r -> < 2r
    store_fp<Fr>(dst + i * 32, Fr::cond_sub<1>(r));
  }
}
int32_t fr_lincomb(Ctx* c, void* d_dst, size_t n, const void* c0, const void* const* d_terms, const size_t* lens, const void* coeffs, size_t k, hipStream_t s) {
  (void)c;
  if (n == 0) return ALEO_MI355X_OK;
  if (k > LC_MAX) { g_last_error = "fr_lincomb: more than 28 terms in one call"; return ALEO_MI355X_ERR_BAD_ARG; }
  LcArgs a{}; a.terms = (uint32_t)k; FrK k0{};
  if (c0) std::memcpy(k0.v, c0, 32);
  for (size_t j = 0; j < k; ++j) {
    if (!d_terms[j] && lens[j]) { g_last_error = "fr_lincomb: null term"; return ALEO_MI355X_ERR_BAD_ARG; }
    a.p[j] = (const char*)d_terms[j]; a.n[j] = lens[j] < n ? lens[j] : n; std::memcpy(a.k[j].v, (const char*)coeffs + 32 * j, 32);
  }
  size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(k_fr_lincomb, dim3((uint32_t)(want < 8192 ? want : 8192)), dim3(256), 0, s, (char*)d_dst, n, a, k0);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// dst[i] += src[i mod n_src] (n_src a power of two dividing n): multiplication of a block of n_src coefficients by 1 + X^n_src + X^(2 n_src) + ...,
// the selector v_{H*} / v_H that puts a smaller circuit's remainder on the largest constraint domain of a proof (varuna.hip).
__global__ void __launch_bounds__(256) k_fr_add_tiled(char* dst, size_t n, const char* __restrict__ src, size_t mask) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    store_fp<Fr>(dst + i * 32, Fr::cond_sub<1>(Fr::add(load_fp<Fr>(dst + i * 32), load_fp<Fr>(src + (i & mask) * 32))));      // canonical inputs: < 2r
}
int32_t fr_add_tiled(Ctx* c, void* d_dst, size_t n, const void* d_src, size_t n_src, hipStream_t s) {
  (void)c;
  if (!n_src || (n_src & (n_src - 1)) || n % n_src) { g_last_error = "fr_add_tiled: block size must be a power of two dividing n"; return ALEO_MI355X_ERR_BAD_ARG; }
  if (n == 0) return ALEO_MI355X_OK;
  size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(k_fr_add_tiled, dim3((uint32_t)(want < 8192 ? want : 8192)), dim3(256), 0, s, (char*)d_dst, n, (const char*)d_src, n_src - 1);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// The two sumcheck numerators, each one pass over evaluations that already sit in HBM (values of the operands on the larger domain):
//   first  (domain 4|H|): dst = r * (a + eta_b b + eta_c a b) − t * z          [UPSTREAM-RECALL: round_functions/second.rs, the summed polynomial]
//   matrix (domain 2|K|): dst = sum_M delta_M (vv val_M − (alpha beta − beta row_M − alpha col_M + row_col_M) f_M)   [fourth.rs]
__global__ void __launch_bounds__(256) k_ahp_first_sumcheck(char* dst, size_t n, const char* r, const char* a, const char* b, const char* t,
                                                            const char* z, FrK keb, FrK kec) {
  const Fr eb = fr_arg(keb), ec = fr_arg(kec);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const Fr va = load_fp<Fr>(a + i * 32), vb = load_fp<Fr>(b + i * 32);
    Fr u = Fr::cond_sub<2>(Fr::add(va, Fr::mul(eb, vb)));                                  // < 3r -> < 2r
    u = Fr::cond_sub<2>(Fr::add(u, Fr::mul(ec, Fr::mul(va, vb))));                          // < 4r -> < 2r
    u = Fr::mul(load_fp<Fr>(r + i * 32), u);                                                // < 2r
    const Fr w = Fr::mul(load_fp<Fr>(t + i * 32), load_fp<Fr>(z + i * 32));                 // < 2r
    store_fp<Fr>(dst + i * 32, Fr::reduce(Fr::sub<2>(u, w)));                               // u + 2r − w < 4r
  }
}
int32_t ahp_first_sumcheck(Ctx* c, void* d_dst, size_t n, const void* d_r, const void* d_a, const void* d_b, const void* d_t, const void* d_z,
                           const void* eta_b, const void* eta_c, hipStream_t s) {
  (void)c;
  if (n == 0) return ALEO_MI355X_OK;
  FrK kb, kc; std::memcpy(kb.v, eta_b, 32); std::memcpy(kc.v, eta_c, 32);
  size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(k_ahp_first_sumcheck, dim3((uint32_t)(want < 16384 ? want : 16384)), dim3(256), 0, s, (char*)d_dst, n, (const char*)d_r, (const char*)d_a,
                     (const char*)d_b, (const char*)d_t, (const char*)d_z, kb, kc);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}
struct MatArgs { const char* idx[3]; const char* f[3]; FrK delta[3]; FrK ab, nalpha, nbeta, vv; size_t stride; };      // idx[m] == nullptr: matrix m is absent
__global__ void __launch_bounds__(256) k_ahp_matrix_sumcheck(char* __restrict__ dst, size_t n, MatArgs a) {
  const Fr ab = fr_arg(a.ab), na = fr_arg(a.nalpha), nb = fr_arg(a.nbeta), vv = fr_arg(a.vv);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    Fr acc = Fr::zero();
    for (int m = 0; m < 3; ++m) {
      if (!a.idx[m]) continue;                                                                // (uniform across the grid)
      const char* e = a.idx[m] + i * 32;                                                    // row, col, val, row_col: `stride` bytes apart
      Fr bq = Fr::cond_sub<2>(Fr::add(ab, Fr::mul(nb, load_fp<Fr>(e))));                     // alpha beta − beta row            < 2r
      bq = Fr::cond_sub<2>(Fr::add(bq, Fr::mul(na, load_fp<Fr>(e + a.stride))));             // − alpha col                      < 2r
      bq = Fr::cond_sub<2>(Fr::add(bq, load_fp<Fr>(e + 3 * a.stride)));                      // + row_col                        < 2r
      const Fr bf = Fr::mul(bq, load_fp<Fr>(a.f[m] + i * 32));                               // < 2r
      const Fr pm = Fr::cond_sub<2>(Fr::sub<2>(Fr::mul(vv, load_fp<Fr>(e + 2 * a.stride)), bf));      // vv val − b f: < 4r -> < 2r
      acc = Fr::cond_sub<2>(Fr::add(acc, Fr::mul(fr_arg(a.delta[m]), pm)));                  // < 2r
    }
    store_fp<Fr>(dst + i * 32, Fr::cond_sub<1>(acc));
  }
}
// consts: 7 Montgomery values on the host — delta_a, delta_b, delta_c, alpha beta, −alpha, −beta, v_H(alpha) v_H(beta)
int32_t ahp_matrix_sumcheck(Ctx* c, void* d_dst, size_t n, const void* const* d_index, size_t index_stride_elems, const void* const* d_f, const void* consts,
                            hipStream_t s) {
  (void)c;
  if (n == 0) return ALEO_MI355X_OK;
  MatArgs a{}; const char* k = (const char*)consts;
  for (int m = 0; m < 3; ++m) { a.idx[m] = (const char*)d_index[m]; a.f[m] = (const char*)d_f[m]; std::memcpy(a.delta[m].v, k + 32 * m, 32); }
  std::memcpy(a.ab.v, k + 96, 32); std::memcpy(a.nalpha.v, k + 128, 32); std::memcpy(a.nbeta.v, k + 160, 32); std::memcpy(a.vv.v, k + 192, 32);
  a.stride = index_stride_elems * 32;
  size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(k_ahp_matrix_sumcheck, dim3((uint32_t)(want < 16384 ? want : 16384)), dim3(256), 0, s, (char*)d_dst, n, a);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// Two layout steps of the prover's first two rounds, one launch each for all instances (they were 3 launches per polynomial and 5 per instance):
//   blind rows:  dst[q] (n + 1 coefficients) = src[q] (n coefficients) + rho_q (X^n − 1)        — w, z_a, z_b after the inverse transform
//   sumcheck operands of instance i on the 4n domain, zero padded:  z_i = w_i (X^|X| − 1) + x̂_i,  z_a,i,  z_b,i   (rows 3i, 3i+1, 3i+2 of dst)
static constexpr uint32_t BLIND_MAX = 24;
struct BlindArgs { FrK rho[BLIND_MAX]; };
__global__ void __launch_bounds__(256) k_blind_rows(char* __restrict__ dst, const char* __restrict__ src, size_t n, uint32_t rows, BlindArgs a) {
  const size_t L = n + 1;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < L * rows; t += (size_t)gridDim.x * 256) {
    const uint32_t q = (uint32_t)(t / L); const size_t i = t % L;
    Fr v = i < n ? load_fp<Fr>(src + ((size_t)q * n + i) * 32) : fr_arg(a.rho[q]);
    if (i == 0) v = Fr::cond_sub<1>(Fr::sub<1>(v, fr_arg(a.rho[q])));                         // v + r − rho < 2r -> < r
    store_fp<Fr>(dst + t * 32, v);
  }
}
int32_t fr_blind_rows(Ctx* c, void* d_dst, const void* d_src, size_t n, size_t rows, const void* rho_mont, hipStream_t s) {
  (void)c;
  if (!rows || !n) return ALEO_MI355X_OK;
  if (rows > BLIND_MAX) { g_last_error = "fr_blind_rows: more than 24 rows"; return ALEO_MI355X_ERR_BAD_ARG; }
  BlindArgs a{}; std::memcpy(a.rho, rho_mont, rows * 32);
  const size_t want = ((n + 1) * rows + 255) / 256;
  hipLaunchKernelGGL(k_blind_rows, dim3((uint32_t)(want < 16384 ? want : 16384)), dim3(256), 0, s, (char*)d_dst, (const char*)d_src, n, (uint32_t)rows, a);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}
// wit: 3 rows of n + 1 coefficients per instance (w, z_a, z_b); xp: |X| coefficients of x̂ per instance; dst: 3 rows of n4 per instance
__global__ void __launch_bounds__(256) k_sumcheck_operands(char* __restrict__ dst, const char* __restrict__ wit, const char* __restrict__ xp, size_t n, size_t n_x,
                                                           size_t n4, uint32_t instances) {
  const size_t L = n + 1;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n4 * 3 * instances; t += (size_t)gridDim.x * 256) {
    const size_t row = t / n4, j = t % n4, i = row / 3, kind = row % 3;
    const char* w = wit + (3 * i) * L * 32;
    Fr v = Fr::zero();
    if (kind == 0) {
      if (j >= n_x && j < L + n_x) v = load_fp<Fr>(w + (j - n_x) * 32);                       // + w X^|X|
      if (j < L) v = Fr::sub<1>(v, load_fp<Fr>(w + j * 32));                                  // − w             (< 2r)
      if (j < n_x) v = Fr::add(v, load_fp<Fr>(xp + (i * n_x + j) * 32));                      // + x̂             (< 3r)
      v = Fr::cond_sub<1>(Fr::cond_sub<2>(v));
    } else if (j < L) v = load_fp<Fr>(w + (kind * L + j) * 32);
    store_fp<Fr>(dst + t * 32, v);
  }
}
int32_t ahp_sumcheck_operands(Ctx* c, void* d_dst, const void* d_wit, const void* d_xp, size_t n, size_t n_x, size_t instances, hipStream_t s) {
  (void)c;
  if (!instances || !n) return ALEO_MI355X_OK;
  const size_t n4 = 4 * n, want = (n4 * 3 * instances + 255) / 256;
  hipLaunchKernelGGL(k_sumcheck_operands, dim3((uint32_t)(want < 32768 ? want : 32768)), dim3(256), 0, s, (char*)d_dst, (const char*)d_wit, (const char*)d_xp, n, n_x, n4, (uint32_t)instances);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// dst[pos[v]] = to_mont(src[v]): the assignment (canonical, variable order) laid out on H in Montgomery form — dst zeroed by the caller.
// An entry that is not below r (the Montgomery product assumes bounded inputs) raises *flag: the caller reads it with its next small read-back.
__global__ void __launch_bounds__(256) k_scatter_to_mont(char* __restrict__ dst, const char* __restrict__ src, const uint32_t* __restrict__ pos, size_t n, uint32_t* flag) {
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
    const Fr x = load_fp<Fr>(src + v * 32);
    bool lt = false;
    for (int k = 7; k >= 0; --k) { const uint32_t m = FrParams::P.v[k]; if (x.v[k] != m) { lt = x.v[k] < m; break; } }
    if (!lt && flag) atomicOr(flag, 1u);
    store_fp<Fr>(dst + (size_t)pos[v] * 32, Fr::to_mont(x));
  }
}
int32_t fr_scatter_to_mont(Ctx* c, void* d_dst, const void* d_src, const void* d_pos, size_t n, void* d_flag, hipStream_t s) {
  (void)c;
  if (!n) return ALEO_MI355X_OK;
  const size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(k_scatter_to_mont, dim3((uint32_t)(want < 8192 ? want : 8192)), dim3(256), 0, s, (char*)d_dst, (const char*)d_src, (const uint32_t*)d_pos, n, (uint32_t*)d_flag);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

int32_t fr_batch_inverse(Ctx* c, void* d_inout, size_t n, hipStream_t s) {
  if (n == 0) return ALEO_MI355X_OK;
  int32_t rc; if ((rc = scratch_acquire(c, c->ntt_tmp, n * 32, s))) return rc;
  // ~64 elements per lane keeps the shared inversion at ~6 products per element; small inputs use fewer per lane
  size_t T = (n + 63) / 64; if (T < 4096) T = n < 4096 ? n : 4096;
  hipLaunchKernelGGL(k_fr_batch_inverse, dim3((uint32_t)((T + 255) / 256)), dim3(256), 0, s, (char*)d_inout, c->ntt_tmp.as<char>(), n, T);
  HIPCHK(hipGetLastError());
  return scratch_release(c, s);
}

// ---- key synthesis: the integer side of the arithmetisation (varuna.hip varuna_index_build) -----------------------------------------------------
// One lane per CSR row (rows of an R1CS matrix are short; the few long linear combinations cost their lane a loop, not the launch).
__global__ void __launch_bounds__(256) k_index_expand_rows(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, const uint32_t* __restrict__ positions, uint32_t rows,
                                                           uint32_t* __restrict__ k_row, uint32_t* __restrict__ k_col, uint32_t* __restrict__ cpos, uint32_t* __restrict__ count_plus1) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  for (uint32_t e = row_ptr[r], end = row_ptr[r + 1]; e < end; ++e) {
    const uint32_t p = positions[col[e]];
    k_row[e] = r; k_col[e] = p; cpos[e] = p; atomicAdd(&count_plus1[p + 1], 1u);
  }
}
// a[i] += a[i - 1] over n entries, one block (n is a domain size: a few passes of 1024 lanes x 8 entries)
__global__ void __launch_bounds__(1024) k_index_scan_inclusive(uint32_t* __restrict__ a, uint32_t n) {
  __shared__ uint32_t wsum[16]; __shared__ uint32_t carry_s;
  const uint32_t tid = threadIdx.x; const int lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n; base += 8192) {
    uint32_t v[8], run = 0; const uint32_t i0 = base + tid * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = i0 + k < n ? a[i0 + k] : 0u; run += v[k]; v[k] = run; }
    uint32_t inc = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t off = carry_s; for (int k = 0; k < wv; ++k) off += wsum[k];
    off += inc - run;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (i0 + k < n) a[i0 + k] = off + v[k];
    __syncthreads();
    if (tid == 1023) carry_s = off + run;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_index_transpose_rows(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ cpos, const uint4* __restrict__ val, uint32_t rows, uint32_t row_base,
                                                              uint32_t* __restrict__ cursor, uint32_t* __restrict__ tcol, uint4* __restrict__ tval) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  for (uint32_t e = row_ptr[r], end = row_ptr[r + 1]; e < end; ++e) {
    const uint32_t at = atomicAdd(&cursor[cpos[e]], 1u);
    tcol[at] = row_base + r; tval[2 * (size_t)at] = val[2 * (size_t)e]; tval[2 * (size_t)at + 1] = val[2 * (size_t)e + 1];
  }
}
int32_t index_expand_rows(Ctx* c, const uint32_t* d_row_ptr, const uint32_t* d_col, const uint32_t* d_positions, size_t rows, uint32_t* d_k_row, uint32_t* d_k_col, uint32_t* d_cpos, uint32_t* d_count_plus1, hipStream_t s) {
  (void)c; if (!rows) return ALEO_MI355X_OK;
  hipLaunchKernelGGL(k_index_expand_rows, dim3((uint32_t)((rows + 255) / 256)), dim3(256), 0, s, d_row_ptr, d_col, d_positions, (uint32_t)rows, d_k_row, d_k_col, d_cpos, d_count_plus1);
  HIPCHK(hipGetLastError()); return ALEO_MI355X_OK;
}
int32_t index_scan_inclusive(Ctx* c, uint32_t* d_a, size_t n, hipStream_t s) {
  (void)c; if (!n) return ALEO_MI355X_OK;
  hipLaunchKernelGGL(k_index_scan_inclusive, dim3(1), dim3(1024), 0, s, d_a, (uint32_t)n);
  HIPCHK(hipGetLastError()); return ALEO_MI355X_OK;
}
int32_t index_transpose_rows(Ctx* c, const uint32_t* d_row_ptr, const uint32_t* d_cpos, const void* d_val, size_t rows, uint32_t row_base, uint32_t* d_cursor, uint32_t* d_tcol, void* d_tval, hipStream_t s) {
  (void)c; if (!rows) return ALEO_MI355X_OK;
  hipLaunchKernelGGL(k_index_transpose_rows, dim3((uint32_t)((rows + 255) / 256)), dim3(256), 0, s, d_row_ptr, d_cpos, (const uint4*)d_val, (uint32_t)rows, row_base, d_cursor, d_tcol, (uint4*)d_tval);
  HIPCHK(hipGetLastError()); return ALEO_MI355X_OK;
}

}  // namespace aleo_mi355x

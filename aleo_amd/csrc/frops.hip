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

namespace aleo_mi355x {

__device__ __forceinline__ Fr fr_canonical_lt4r(const Fr& a) { return Fr::cond_sub<1>(Fr::cond_sub<2>(a)); }

template <int OP>
__global__ void __launch_bounds__(256) k_fr_vec_op(char* __restrict__ dst, const char* a, const char* b, size_t n) {
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

__global__ void __launch_bounds__(256) k_spmv_rows(char* __restrict__ y, const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                                   const char* __restrict__ vals, const char* __restrict__ x, uint32_t rows,
                                                   uint32_t* __restrict__ long_rows, uint32_t* __restrict__ n_long) {
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
    const uint32_t k0 = row_ptr[r], k1 = row_ptr[r + 1];
    if (k1 - k0 > SPMV_LANE_MAX) { long_rows[atomicAdd(n_long, 1u)] = r; continue; }
    Fr acc = Fr::zero();
    for (uint32_t k = k0; k < k1; ++k) acc = Fr::cond_sub<2>(Fr::add(acc, spmv_term(vals, col, x, k)));     // < 4r -> < 2r
    store_fp<Fr>(y + (size_t)r * 32, Fr::cond_sub<1>(acc));
  }
}

__global__ void __launch_bounds__(256) k_spmv_long(char* __restrict__ y, const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                                   const char* __restrict__ vals, const char* __restrict__ x,
                                                   const uint32_t* __restrict__ long_rows, const uint32_t* __restrict__ n_long) {
  const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
  for (uint32_t q = wave; q < *n_long; q += nwaves) {
    const uint32_t r = long_rows[q], k0 = row_ptr[r], k1 = row_ptr[r + 1];
    Fr acc = Fr::zero();
    for (uint32_t k = k0 + lane; k < k1; k += 64) acc = Fr::cond_sub<2>(Fr::add(acc, spmv_term(vals, col, x, k)));
    for (int d = 32; d >= 1; d >>= 1) {
      Fr o;
#pragma unroll
      for (int l = 0; l < 8; ++l) o.v[l] = __shfl_xor(acc.v[l], d);
      acc = Fr::cond_sub<2>(Fr::add(acc, o));
    }
    if (lane == 0) store_fp<Fr>(y + (size_t)r * 32, Fr::cond_sub<1>(acc));
  }
}

int32_t fr_spmv(Ctx* c, void* d_y, const void* d_row_ptr, const void* d_col, const void* d_vals, const void* d_x, size_t rows, hipStream_t s) {
  if (rows == 0) return ALEO_MI355X_OK;
  if (rows >= (1ull << 32)) { g_last_error = "fr_spmv: row count exceeds 2^32"; return ALEO_MI355X_ERR_BAD_ARG; }
  int32_t rc; if ((rc = scratch_acquire(c, c->ntt_tmp, (rows + 16) * 4, s))) return rc;
  uint32_t* n_long = c->ntt_tmp.as<uint32_t>(); uint32_t* long_rows = n_long + 16;
  HIPCHK(hipMemsetAsync(n_long, 0, 4, s));
  const size_t want = (rows + 255) / 256; const uint32_t grid = (uint32_t)(want < 16384 ? want : 16384);
  hipLaunchKernelGGL(k_spmv_rows, dim3(grid), dim3(256), 0, s, (char*)d_y, (const uint32_t*)d_row_ptr, (const uint32_t*)d_col, (const char*)d_vals,
                     (const char*)d_x, (uint32_t)rows, long_rows, n_long);
  hipLaunchKernelGGL(k_spmv_long, dim3(1024), dim3(256), 0, s, (char*)d_y, (const uint32_t*)d_row_ptr, (const uint32_t*)d_col, (const char*)d_vals,
                     (const char*)d_x, long_rows, n_long);
  HIPCHK(hipGetLastError());
  return scratch_release(c, s);
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

}  // namespace aleo_mi355x

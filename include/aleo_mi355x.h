/*
 * aleo_mi355x.h — C ABI of libaleo_mi355x.so: the MI355X (gfx950) backend for the two operators that dominate
 * the Aleo SDK's execute/prove path:
 *
 *   a1  snarkvm_algorithms::msm::VariableBase::msm::<G1Affine>      (BLS12-377 G1 Pippenger MSM)
 *   a2  snarkvm_algorithms::fft::EvaluationDomain::<Fr>::{fft,ifft,coset_fft,coset_ifft}_in_place
 *
 * Reference interfaces this ABI replaces.  The prover is crates.io snarkVM =0.14.5 (pins:
 * /root/reference/Cargo.toml:28-53, /root/reference/Cargo.lock:2200 snarkvm-algorithms), entered from
 *   /root/reference/rust/src/program/execute.rs:74   trace.prove_execution::<A,_>(…)
 *   /root/reference/rust/src/program/execute.rs:177  vm.execute(…)
 *   /root/reference/rust/src/program/transfer.rs:99  vm.execute(… "credits.aleo","transfer_public" …)
 * and the seam is the one upstream's optional `cuda` feature uses [UPSTREAM-RECALL]:
 *   algorithms/src/msm/variable_base/mod.rs   VariableBase::msm  -> snarkvm_algorithms_cuda::msm(bases, scalars)
 *   algorithms/src/fft/domain.rs              in_order_{fft,ifft}_in_place -> snarkvm_algorithms_cuda::NTT(size, data,
 *                                             NTTInputOutputOrder, NTTDirection, NTTType)
 * with the same contract: a non-zero return means "not computed" and the Rust caller falls back to its CPU path.
 * INTEGRATION.md shows the Rust `extern "C"` block a maintainer adds on the snarkVM side.
 *
 * Data layouts are snarkVM's in-memory layouts (no conversion at the boundary):
 *   Fr      4 x u64 little-endian limbs, Montgomery form (R = 2^256)                         32 bytes
 *   scalar  BigInteger256: 4 x u64 little-endian limbs, CANONICAL (non-Montgomery), < r      32 bytes
 *   G1Affine {x: Fq, y: Fq, infinity: bool}: 6+6 x u64 Montgomery (R = 2^384) + flag byte    stride 104
 *            (stride 96 = x,y only, no flag, is also accepted)
 *   G1Projective (Jacobian) {x, y, z: Fq}                                                    144 bytes
 * MSM results are returned affine-normalised as Jacobian (x, y, 1), or (1, 1, 0) for the identity; compare
 * group elements after to_affine(), never raw Jacobian limbs.
 *
 * Threading: every entry point may be called concurrently from many host threads (snarkVM commits the
 * polynomials of one round from a rayon pool).  Each call runs on one of the device's slots (own stream and
 * workspaces; ALEO_MI355X_SLOTS = 1..8, default 4), so concurrent calls overlap on the GPU; a pinned set
 * stays alive until the calls using it return, even if another thread unpins it meanwhile.
 * Streams: the *_device entry points that return a result to HOST memory (msm, kzg_commit) complete before they return.
 * Those that only transform device data (ntt_fr*_device, fr_*_device) enqueue on the caller's `stream` and return at once:
 * the caller orders its own work on that stream, and the library orders its internal scratch between calls by events.
 * With stream == NULL they run on the serving slot's stream — which the caller cannot order against — and therefore
 * complete before returning.  hipStreamLegacy names the null stream; hipStreamPerThread is refused (BAD_ARG).
 * Ownership: the caller owns every buffer; nothing is retained after return except through bases_pin (and, when the
 * caller opts in with ALEO_MI355X_SRS_CACHE=1, the one-shot msm_g1's base-array cache described there).
 * No exceptions cross the boundary.
 */
#ifndef ALEO_MI355X_H
#define ALEO_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (0 = OK; anything else: caller recomputes on CPU) */
#define ALEO_MI355X_OK 0
#define ALEO_MI355X_ERR_NO_DEVICE 1      /* no gfx950 device / HIP runtime failure at init */
#define ALEO_MI355X_ERR_BAD_ARG 2        /* null pointer, bad stride, lg_n out of range, … */
#define ALEO_MI355X_ERR_HIP 3            /* a HIP call failed; see aleo_mi355x_last_error() */
#define ALEO_MI355X_ERR_BAD_HANDLE 4
#define ALEO_MI355X_ERR_OOM 5
#define ALEO_MI355X_ERR_UNSATISFIED 6  /* varuna_prove*: an assignment does not satisfy its circuit (upstream: the synthesiser's is_satisfied check) */

/* NTT enums: mirror snarkvm_algorithms_cuda::{NTTInputOutputOrder, NTTDirection, NTTType} */
#define ALEO_NTT_ORDER_NN 0   /* natural in, natural out (what fft_in_place exposes) */
#define ALEO_NTT_ORDER_NR 1   /* natural in, bit-reversed out */
#define ALEO_NTT_ORDER_RN 2   /* bit-reversed in, natural out */
#define ALEO_NTT_ORDER_RR 3
#define ALEO_NTT_FORWARD 0
#define ALEO_NTT_INVERSE 1    /* multiplies by size_inv, as EvaluationDomain::ifft_in_place */
#define ALEO_NTT_STANDARD 0
#define ALEO_NTT_COSET 1      /* coset shift g = Fr::multiplicative_generator() = 22 */

/* ENVIRONMENT (set by the HOST before its first HIP call; the library never edits the process environment): GPU_MAX_HW_QUEUES=8 is recommended.  The HIP
 * runtime maps a process's streams onto that many hardware queues (4 by default) and reads the variable once, when it initialises; the library runs a
 * caller's stream, a side stream, high-priority streams and the lockstep prover's worker streams at once (lockstep proofs -5 % with 8 queues, everything
 * else within noise; results never depend on it; 12 or more oversubscribe the device's queues and cost 1.5-2x on lockstep calls).  The library creates
 * the main streams of its slots and of as many helper contexts FIRST, at the device's first use, so that each gets a hardware queue of its own (a
 * lockstep call of 8 proofs: 31.8 -> 27.5 ms): initialise the device (aleo_mi355x_init_device) before the host creates many streams of its own.
 * aleo_amd/__init__.py and bench.py set the variable as their own default; INTEGRATION.md 4 shows the Rust side. */

/* initialises the first n_devices visible devices (0 = all of them).  Idempotent, thread-safe; the calling thread's current
 * HIP device is left as it was.  Every single-device entry point below works on the CALLING THREAD's current HIP device (hipSetDevice is per
 * thread) and initialises it lazily if neither init call was made.  init_device selects and initialises one device (-1 = the current one).
 * device_count: visible devices / devices initialised so far (either pointer may be NULL). */
int32_t aleo_mi355x_init(int32_t n_devices);
int32_t aleo_mi355x_init_device(int32_t device);
int32_t aleo_mi355x_device_count(int32_t* visible, int32_t* initialised);
/* aleo_mi355x_init also turns on direct peer access (xGMI) between every ordered pair of the devices initialised so far, so that the exchange of a
 * sharded transform and the slice pulls of a sharded commitment are link-to-link copies; a pair the platform refuses is left to the runtime's staged
 * copies (never an error).  peer_info: ordered pairs with direct access / refused so far (either pointer may be NULL).
 * ABI note: since 0.2.0 `init` takes a device COUNT; 0.1.x callers that meant "select device d" call init_device(d). */
int32_t aleo_mi355x_peer_info(int32_t* enabled_pairs, int32_t* refused_pairs);

/* a1 — VariableBase::msm(bases: &[G1Affine], scalars: &[BigInteger256]) -> G1Projective.
 * Host pointers.  n = min(len(bases), len(scalars)) is the caller's job (the reference zips the slices).
 * By default every call uploads and converts its base array and keeps nothing.  ALEO_MI355X_SRS_CACHE=1 (read once)
 * opts in to a cache for callers that cannot hold a bases_pin handle: arrays of >= 1024 points stay in HBM keyed by host
 * pointer + stride and are re-validated on each call by hashing 256 sampled points (32 at the front, the rest spread);
 * the third use builds the fixed-base table (~0.15 s, inside that call).  Contract when opted in: a base array must not
 * be rewritten in place (nor its address reused for different points) between calls — unsampled changes are not seen. */
int32_t aleo_mi355x_msm_g1(void* out_jacobian, const void* bases, size_t base_stride, const void* scalars, size_t n);

/* SRS residency: upload + convert a base set once per proving key (SURVEY.md §5 "device-resident SRS").
 * The handle stays valid until unpin.  bases: host pointer. */
int32_t aleo_mi355x_bases_pin(const void* bases, size_t base_stride, size_t n, uint64_t* handle);
int32_t aleo_mi355x_bases_unpin(uint64_t handle);
/* Synthetic SRS-shaped base set generated in HBM: P_i = (first_multiple + i) * base for i in [0, n)
 * ("P_i = (i+1)*G generated on device" — 2^26 points never cross PCIe).  base_affine: one
 * snarkVM Affine (104 bytes, host).  first_multiple >= 1 and first_multiple + n must stay below r. */
int32_t aleo_mi355x_bases_generate(const void* base_affine104, uint64_t first_multiple, size_t n, uint64_t* handle);
/* Synthetic SRS-shaped set: P_i = s_i * base for n canonical 32-byte scalars in host memory (P_i = beta^i * G, the
 * shape of a universal setup's powers_of_beta_g, for which a commitment to p is p(beta) * G).  A zero scalar gives the identity. */
int32_t aleo_mi355x_bases_from_scalars(const void* base_affine104, const void* scalars, size_t n, uint64_t* handle);
/* Optional fixed-base acceleration for a pinned set (an SRS never changes): builds tables of window multiples
 * 2^(c w) * P_i in HBM, stored in the 28-bit-limb form the accumulation kernel computes in (112 bytes per entry), in up to
 * three tiers so that a prefix of ANY length >= 2^10 gets a window width that suits it (KZG10::commit multiplies polynomials
 * of every degree against one SRS): the whole set at c = 20 (>= 2^19 points; 13 rows) or c = 17, its first 2^17 + 64 points at
 * c = 16, its first 2^15 + 64 points at c = 13 (the 64: a committer key is a power of two of powers followed by a few hiding powers).  MSMs over this handle then add one table entry per window into ONE shared bucket
 * set (13 instead of 16 additions per point at 2^20) and skip the Horner tail.  Same results, bit for bit after
 * normalisation.  About 0.15 s and 1.8 GB for a 2^20-point set. */
int32_t aleo_mi355x_bases_precompute(uint64_t handle);
/* What a pinned set holds in HBM: out[0] points, [1] bytes of the base rows (96-byte snarkVM rows + their 112-byte
 * 28-bit-limb copies), [2] bytes of the fixed-base tables, [3..5] window width c of each table tier (0 = none), [6] tiers.
 * Returns the number of values written (<= cap), 0 for an unknown handle. */
/* A second, narrow-window table (window_bits 13 or 16) over ONE sub-range [offset, offset + n) of a pinned set, for MSMs whose scalars are mostly
 * 0 / 1 — a witness committed in evaluation form against Lagrange-basis powers (KZG10::commit_lagrange): such vectors put hardly a point per bucket into
 * the 2^19 buckets of the wide window, and a proof commits three of them per instance.  Used by aleo_mi355x_kzg_commit_segments_sparse_device (and by
 * the prover's first round) when every segment of the call lies inside the range; up to 32 results share one launch chain.  One range per set. */
int32_t aleo_mi355x_bases_precompute_range(uint64_t handle, size_t offset, size_t n, int32_t window_bits);
int32_t aleo_mi355x_bases_info(uint64_t handle, uint64_t* out, int32_t cap);
/* Copies pinned bases [offset, offset+n) back to the host as snarkVM Affine (stride 104). */
int32_t aleo_mi355x_bases_download(uint64_t handle, size_t offset, size_t n, void* out_affine104);
/* MSM over the first n pinned bases; scalars: host pointer (pageable memory is fine: the upload runs at the link's rate either way).  Against a set with
 * window tables, from 2^19 points on (ALEO_MI355X_MERGE_MIN_LG), the scalars go up and through in 2 chunks of 37 / 63 % (3 of 18 / 30 / 52 % from 2^21
 * points, ALEO_MI355X_CHUNKS3_MIN_LG) on contexts of their own: a later chunk's upload and sort run under the previous chunk's accumulation, each
 * accumulation is SEEDED with the bucket sums of the chunk before it (no merge kernel), and ONE bucket reduction follows the last chunk; the result
 * does not depend on it. */
int32_t aleo_mi355x_msm_g1_pinned(void* out_jacobian, uint64_t handle, const void* scalars, size_t n);
/* Same, scalars already resident in device memory (hipMalloc'ed or a torch CUDA tensor's data_ptr).
 * The result (144 bytes) is written to HOST memory.  `stream` (here and in every *_device entry point) is a hipStream_t;
 * NULL = the stream of the library slot serving the call, which is NOT ordered with the caller's other streams — a caller
 * whose data was produced on the legacy default stream passes hipStreamLegacy. */
int32_t aleo_mi355x_msm_g1_device(void* out_jacobian, uint64_t handle, const void* d_scalars, size_t n, void* stream);
/* The same with the hint that the scalars are witness-like (mostly 0 / 1 / short): served from the set's range table (bases_precompute_range) when the call
 * lies inside it, otherwise exactly like msm_g1_device.  The host-scalar call msm_g1_pinned looks at 257 of its scalars and takes the hint by itself
 * when the range table starts at point 0. */
int32_t aleo_mi355x_msm_g1_device_sparse(void* out_jacobian, uint64_t handle, const void* d_scalars, size_t n, void* stream);
/* Sum of `count` Jacobian points (144 bytes each, host memory): the local group-add that follows the
 * all-gather of per-GPU partial MSM results.  Result affine-normalised as above. */
int32_t aleo_mi355x_g1_sum(void* out_jacobian, const void* jacobian_points, size_t count);
/* e — ONE MSM over several devices of this process (BASELINE configs[4]: 2^26 points over 8 GPUs).  The base set is cut into
 * n_devices contiguous shards [n g / G, n (g+1) / G); shard g is pinned on devices[g] (devices == NULL: device g mod visible; a device may be
 * listed more than once), with its fixed-base table when precompute != 0.  msm_g1_sharded multiplies scalars[0..n) (n <= the pinned count:
 * shards past n contribute the identity) — every shard runs the whole Pippenger on its device from its own host thread, the G partial sums
 * (144 bytes each; copied to partials_out in shard order when it is not NULL) are added on the host in shard order by aleo_mi355x_g1_sum, so the
 * result is the affine-normalised sum, bit for bit what a single device gives.  No collective: within one process the exchange is G stores
 * into host memory (ranks in separate processes exchange the same partials over RCCL: aleo_amd/dist.py).  generate_sharded: the synthetic set
 * P_i = (first_multiple + i) * base built shard by shard in each device's HBM.  sharded_info: [G, then per shard device, first point, count]. */
int32_t aleo_mi355x_bases_pin_sharded(const void* bases, size_t base_stride, size_t n, const int32_t* devices, size_t n_devices, int32_t precompute, uint64_t* handle);
int32_t aleo_mi355x_bases_generate_sharded(const void* base_affine104, uint64_t first_multiple, size_t n, const int32_t* devices, size_t n_devices, int32_t precompute, uint64_t* handle);
int32_t aleo_mi355x_bases_unpin_sharded(uint64_t handle);
int32_t aleo_mi355x_bases_sharded_info(uint64_t handle, uint64_t* out, int32_t cap);
int32_t aleo_mi355x_msm_g1_sharded(void* out_jacobian, uint64_t handle, const void* scalars, size_t n, void* partials_out);

/* VariableBase::msm::<G2Affine> (SRS / setup paths; the prover itself never runs one).  BLS12-377 G2 over
 * Fq2 = Fq[u]/(u^2 + 5).  bases: snarkVM G2Affine {x: Fq2 (c0, c1), y: Fq2, infinity: bool}, Montgomery limbs, stride 200 (flag byte
 * at 192) or 192; scalars as for G1; result G2Projective (Jacobian x, y, z: Fq2 = 288 bytes), affine-normalised like the G1 result:
 * (x, y, 1), or (1, 1, 0) for the identity.  Host pointers; nothing is retained.  g2_sum: the group add after an all-gather. */
int32_t aleo_mi355x_msm_g2(void* out_jacobian288, const void* bases, size_t base_stride, const void* scalars, size_t n);
int32_t aleo_mi355x_g2_sum(void* out_jacobian288, const void* jacobian_points288, size_t count);
/* A G2 base set resident in HBM (what aleo_mi355x_bases_pin is for G1; SRS powers in G2, a verifying key's elements): rows, infinity flags and the
 * accumulation's 28-bit rows stay on the device, a call uploads its scalars only and multiplies any PREFIX of the set (n <= pinned count).  Same
 * argument formats and result as aleo_mi355x_msm_g2, byte for byte; no window tables (the prover never runs a G2 MSM).  unpin frees the set once the
 * calls that hold it have returned. */
int32_t aleo_mi355x_bases_g2_pin(const void* bases, size_t base_stride, size_t n, uint64_t* handle);
int32_t aleo_mi355x_bases_g2_unpin(uint64_t handle);
int32_t aleo_mi355x_msm_g2_pinned(void* out_jacobian288, uint64_t handle, const void* scalars, size_t n);

/* a2 — EvaluationDomain NTT over Fr, in place, n = 2^lg_n elements (lg_n <= 30), host pointer. */
int32_t aleo_mi355x_ntt_fr(void* inout, uint32_t lg_n, int32_t order, int32_t direction, int32_t type);
/* Same on device-resident data (in place). */
int32_t aleo_mi355x_ntt_fr_device(void* d_inout, uint32_t lg_n, int32_t order, int32_t direction, int32_t type, void* stream);
/* `batch` independent transforms of 2^lg_n elements each, contiguous in d_inout (the local row / column transforms of a
 * transform sharded over several GPUs, SURVEY.md §8e "4-step"; also one call for the polynomials of a prover round). */
int32_t aleo_mi355x_ntt_fr_batch_device(void* d_inout, uint32_t lg_n, size_t batch, int32_t order, int32_t direction, int32_t type, void* stream);
/* The same OUT OF PLACE with a zero-padded input (natural order in and out): transform b reads element i from d_src + (b * src_stride + i) * 32 for
 * i < src_len <= 2^lg_n and takes 0 beyond, and writes its 2^lg_n values to d_out + b * 2^lg_n * 32.  EvaluationDomain::fft of a polynomial of fewer
 * coefficients than the domain (snarkVM resizes the coefficient vector with zeros first: `coeffs.resize(self.size(), F::zero())` in fft / coset_fft); the
 * padding costs no launch and no HBM traffic here — the first pass simply does not read past src_len.  d_out must not overlap the source. */
int32_t aleo_mi355x_ntt_fr_from_device(void* d_out, const void* d_src, size_t src_stride, size_t src_len, uint32_t lg_n, size_t batch, int32_t direction, int32_t type, void* stream);
/* Factors over a rows x cols block (row-major, in place) of the index space of the size-2^lg_n domain:
 *   mode 0: x[r][c] *= w^((row0 + r) * (col0 + c))   — the twiddle between the two axes of a 4-step transform
 *   mode 1: x[r][c] *= g^((row0 + r) * ld + col0 + c) — the coset shift (g = 22) of a block of the coefficient matrix
 *           (every index must stay below 2^lg_n)
 * direction 1 uses w^-1 / g^-1 (no n^-1: the inverse sub-transforms carry their own scale).  lg_n <= 32 here: the domain
 * may be larger than one GPU's share (exponents are carried in 32 bits). */
int32_t aleo_mi355x_fr_grid_scale_device(void* d_data, uint32_t lg_n, uint64_t rows, uint64_t cols, uint64_t row0, uint64_t col0, uint64_t ld,
                                         int32_t mode, int32_t direction, void* stream);
/* dst[c][r] = src[r][c] for 32-byte elements (rows x cols -> cols x rows; dst != src): the layout changes of a 4-step transform. */
int32_t aleo_mi355x_fr_transpose_device(void* d_dst, const void* d_src, uint64_t rows, uint64_t cols, void* stream);
/* e — ONE transform over several devices of this process (SURVEY.md 8(e) "NTT (if sharded): 4-step"): EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place
 * on a host buffer of 2^lg_n Montgomery elements, natural order in and out, split over n_devices (a power of two, <= 2^floor(lg_n / 2); devices == NULL:
 * device g mod visible; a device may be listed more than once).  n = R C: device g uploads the coefficient columns [g C / G, (g + 1) C / G) — its 1/G of the
 * PCIe traffic —, runs its column transforms and the twiddle, the devices exchange one block per pair (peer copies: one per xGMI link), every device
 * runs its row transforms and stores its k_r range of X[k_c R + k_r] straight into the host buffer.  Same values as aleo_mi355x_ntt_fr. */
int32_t aleo_mi355x_ntt_fr_sharded(void* inout, uint32_t lg_n, int32_t direction, int32_t type, const int32_t* devices, size_t n_devices);
/* e2 — the same transform on data RESIDENT in HBM: 2^lg_n Montgomery elements in natural order at d_inout on the calling thread's current device ("home"),
 * transformed in place over the listed devices with no host buffer: home transposes the coefficient matrix once so that every device's columns are one
 * contiguous slab, device g pulls its slab (hipMemcpyPeerAsync over xGMI; the home device itself: a plain copy), column transforms + twiddle, the all-to-all
 * as one peer copy per ordered pair, row transforms, every device pushes its rows back and home restores the natural order.  Same values as
 * aleo_mi355x_ntt_fr_device.  Blocking (the result is complete on return); `stream`: the stream the data was produced on (NULL = the slot's own).  The prover
 * routes its transforms of >= 2^24 elements (aleo_mi355x_bases_shard_transforms sets the size) through this entry when its committer key has shards attached (aleo_mi355x_bases_attach_shards): the
 * "NTT coefficients" half of "large proofs shard MSM bases and NTT coefficients across the GPUs". */
int32_t aleo_mi355x_ntt_fr_sharded_device(void* d_inout, uint32_t lg_n, int32_t direction, int32_t type, const int32_t* devices, size_t n_devices, void* stream);

/* a5 — KZG10::commit shape (commit_lagrange is the same call over the pinned Lagrange-basis powers with evaluations as
 * the scalars): coefficients in Montgomery form (as polynomials are stored), converted to canonical
 * bigints on the device, MSM over the first n pinned bases; affine result as snarkVM Affine (104 bytes, host). */
int32_t aleo_mi355x_kzg_commit(void* out_affine104, uint64_t handle, const void* coeffs_mont, size_t n);
/* Device-resident coefficients (e.g. straight out of aleo_mi355x_ntt_fr_device: no host round trip). */
int32_t aleo_mi355x_kzg_commit_device(void* out_affine104, uint64_t handle, const void* d_coeffs_mont, size_t n, void* stream);

/* The commitments of ONE prover round in one call.  Varuna commits 3*instances+1, then 2, 3 and 1 polynomials per round
 * (snarkvm-algorithms 0.14.5 snark/varuna/ahp/prover/round_functions [UPSTREAM-RECALL], reached from
 * /root/reference/rust/src/program/execute.rs:74), each by its own KZG10::commit -> VariableBase::msm from a rayon worker; at real
 * circuit sizes (2^14..2^17 coefficients) one such MSM is latency-bound on a GPU.  Here k coefficient vectors (k DEVICE
 * pointers and k lengths, both arrays in host memory) go against prefixes of ONE pinned SRS and share every launch: one sort
 * over k bucket sets, one accumulation, one reduction (pinned sets with a precomputed table; otherwise the call degrades to
 * k MSMs).  Results: k rows, in the order given — 144-byte Jacobian for msm_g1_batch_device (canonical scalars), snarkVM Affine
 * (104 bytes) for the kzg_commit_batch forms (Montgomery coefficients).  Each result is bit-identical to the single-vector call. */
int32_t aleo_mi355x_msm_g1_batch_device(void* out_jacobian, uint64_t handle, const void* const* d_scalars, const size_t* lens, size_t k, void* stream);
int32_t aleo_mi355x_kzg_commit_batch_device(void* out_affine104, uint64_t handle, const void* const* d_coeffs_mont, const size_t* lens, size_t k, void* stream);
/* Same with the k coefficient vectors in host memory (k host pointers). */
int32_t aleo_mi355x_kzg_commit_batch(void* out_affine104, uint64_t handle, const void* const* coeffs_mont, const size_t* lens, size_t k);

/* SonicKZG10::commit for the labelled polynomials of one round [UPSTREAM-RECALL: algorithms/src/polycommit/sonic_pc/mod.rs commit() ->
 * kzg10::KZG10::commit per polynomial with `powers` or, for a degree bound d, `shifted_powers` = powers_of_beta_g[max_degree - d ..], plus
 * a random polynomial against powers_of_beta_times_gamma_g when a hiding bound is set].  Every commitment is a SUM OF SEGMENTS over one
 * pinned set: a segment multiplies `len` Montgomery coefficients with the bases [base_offset, base_offset + len) and adds into commitment
 * `output`.  Plain polynomial: one segment at offset 0.  Degree bound d: one segment at offset max_degree - d.  Hiding: a second segment
 * for the blinding polynomial at the offset where the caller pinned the gamma powers behind the powers (one pinned array: powers |
 * gamma powers).  All segments of all commitments share one launch chain, like kzg_commit_batch.  out: n_outputs snarkVM Affine rows. */
typedef struct { const void* scalars; size_t len; size_t base_offset; uint32_t output; } aleo_mi355x_commit_segment;
int32_t aleo_mi355x_kzg_commit_segments(void* out_affine104, size_t n_outputs, uint64_t handle, const aleo_mi355x_commit_segment* segments, size_t n_segments);
int32_t aleo_mi355x_kzg_commit_segments_device(void* out_affine104, size_t n_outputs, uint64_t handle, const aleo_mi355x_commit_segment* segments, size_t n_segments, void* stream);
/* the same with the hint that the scalars are sparse (mostly 0 / 1): served from the set's range table (bases_precompute_range) when all segments lie inside it */
int32_t aleo_mi355x_kzg_commit_segments_sparse_device(void* out_affine104, size_t n_outputs, uint64_t handle, const aleo_mi355x_commit_segment* segments, size_t n_segments, void* stream);
/* e2 — commitments, and whole proofs, over several devices of this process (BASELINE north_star: "large proofs shard MSM bases ... across the 8 GPUs";
 * the reference's host is one process proving under spawn_blocking: /root/reference/rust/develop/src/routes.rs:119,149,229 -> rust/src/program/execute.rs:74).
 * The committer key is pinned twice: whole on the prover's device (bases_pin / bases_from_scalars — without tables it costs 224 bytes per power) and as
 * a sharded copy (bases_pin_sharded with precompute: each device holds 1/G of the powers and of the window tables, the part that needs the memory).
 *   kzg_commit_segments_sharded_device / kzg_commit_batch_sharded_device: the segment / batch commitment of kzg_commit_segments_device against the SHARDED
 *     set.  The coefficient vectors are device memory of the calling thread's current device; a segment [offset, offset + len) is cut at the shard
 *     boundaries, device g pulls its pieces (hipMemcpyPeerAsync over xGMI; no copy on the vectors' own device) and runs the ordinary batched Pippenger
 *     against its shard; n_outputs x 144 bytes per shard come back and are added on the host in shard order.  Same bytes as the single-device call.
 *   bases_attach_shards(handle, sharded_handle, min_points): from then on every commitment the PROVER makes against `handle` (varuna_index_build,
 *     varuna_prove*, the lockstep call) with at least min_points scalars in all goes through the sharded copy instead; the rounds' field work and the
 *     transcript stay on the prover's device, its transforms from a size of their own on take the shards' devices too (bases_shard_transforms below).  Same proof bytes.  sharded_handle 0 detaches. */
int32_t aleo_mi355x_kzg_commit_segments_sharded_device(void* out_affine104, size_t n_outputs, uint64_t sharded_handle, const aleo_mi355x_commit_segment* segments, size_t n_segments, void* stream);
int32_t aleo_mi355x_kzg_commit_batch_sharded_device(void* out_affine104, uint64_t sharded_handle, const void* const* d_coeffs_mont, const size_t* lens, size_t k, void* stream);
int32_t aleo_mi355x_bases_attach_shards(uint64_t handle, uint64_t sharded_handle, size_t min_points);
/* With shards attached the prover also runs its TRANSFORMS of at least min_elements elements over the shards' devices (aleo_mi355x_ntt_fr_sharded_device above;
 * the device list must be a power of two, else they stay on the prover's device).  Default (also restored by every attach, and by min_elements = 0): 2^24 —
 * below that a single-device transform takes 0.1-2 ms and the three barriers and 2 (G - 1) peer copies per device of the split cost more than they save; a
 * deployment whose vectors do not fit one card sets it lower.  Same proof bytes for every value. */
int32_t aleo_mi355x_bases_shard_transforms(uint64_t handle, size_t min_elements);

/* KZG10::commit with a hiding bound: msm(powers, coeffs) + msm(gamma_powers, blinding_coeffs), affine result.
 * Both coefficient vectors are Montgomery Fr on the host; both base sets are pinned handles. */
int32_t aleo_mi355x_kzg_commit_hiding(void* out_affine104, uint64_t h_powers, const void* coeffs_mont, size_t n,
                                      uint64_t h_gamma_powers, const void* blinding_mont, size_t m);

/* Field-only vector kernels on device-resident Montgomery Fr data (the pointwise work between
 * NTTs in Varuna's rounds — Evaluations::{mul,add,sub}_assign, snarkvm_fields::batch_inversion).
 * op: 0 = a*b, 1 = a+b, 2 = a-b; dst may alias a or b; outputs canonical.  batch_inverse leaves zeros in place. */
#define ALEO_FR_OP_MUL 0
#define ALEO_FR_OP_ADD 1
#define ALEO_FR_OP_SUB 2
int32_t aleo_mi355x_fr_vec_op_device(void* d_dst, const void* d_a, const void* d_b, size_t n, int32_t op, void* stream);
int32_t aleo_mi355x_fr_batch_inverse_device(void* d_inout, size_t n, void* stream);
/* dst[i] = c0 + c1 * a[i] + c2 * b[i]: c0, c1, c2 are 32-byte Montgomery Fr in HOST memory (c0 NULL = 0); d_a / d_b may be NULL
 * (the term drops out; its constant is then ignored); dst may alias a or b.  The scalar-times-vector and blended forms of the AHP
 * rounds: alpha - h_i ahead of a batch inversion, eta-weighted sums, the linear combinations opened at beta and gamma
 * [UPSTREAM-RECALL: snark/varuna/ahp/prover/round_functions/{second,third,fourth}.rs]. */
int32_t aleo_mi355x_fr_lin_device(void* d_dst, size_t n, const void* c0_mont, const void* c1_mont, const void* d_a, const void* c2_mont, const void* d_b, void* stream);
/* dst[k] = first * ratio^k (first, ratio: 32-byte Montgomery Fr in HOST memory).  With first = a^(n-1), ratio = 1/a this is the
 * coefficient vector of u_H(a, X) = (v_H(a) - v_H(X)) / (a - X) for |H| = n, whose NTT over H is v_H(a) / (a - h): the values the second
 * and third AHP rounds need, without a field inversion on the device. */
int32_t aleo_mi355x_fr_powers_device(void* d_dst, size_t n, const void* first_mont, const void* ratio_mont, void* stream);
/* dst[i] = scale[i] * table1[idx1[i]] * table2[idx2[i]] (uint32 indices; d_scale and the second table may be NULL): the third round's
 * val(k) / ((alpha - row(k)) (beta - col(k))) from the two tables above, by the row / column index of every non-zero entry. */
int32_t aleo_mi355x_fr_gather_mul_device(void* d_dst, size_t n, const void* d_scale, const void* d_table1, const void* d_idx1, const void* d_table2, const void* d_idx2, void* stream);
/* out[q] = p_q(z_q) for k <= 12 polynomials in two launches (d_polys, lens, z_mont: host arrays of k device pointers / lengths /
 * 32-byte Montgomery points; d_out: k x 32 bytes, device): the evaluations a proof carries (z_b, g_1 at beta; g_a, g_b, g_c at gamma). */
int32_t aleo_mi355x_fr_eval_batch_device(void* d_out, const void* const* d_polys, const size_t* lens, const void* z_mont, size_t k, void* stream);
/* Prover randomness.  `seed` is 32 bytes of caller entropy per proof (what `rand::thread_rng()` is to the reference's call sites,
 * /root/reference/rust/src/program/execute.rs:74): the key of a ChaCha20 stream (20 rounds, 64-bit block counter, 64-bit nonce).  Element i =
 * the first candidate below r among the two 32-byte halves (little-endian, low 253 bits) of block(counter = i, nonce = attempt), attempt = 0, 1, ...
 * (rejection sampling: uniform over Fr).  _device writes elements first_index .. first_index + n - 1 in HBM, canonical or (montgomery != 0)
 * Montgomery; aleo_mi355x_fr_random is the same stream on the host (out: n x 32 bytes canonical) — counter-based, so the host draws the handful
 * of blinding scalars of a proof while the device draws the 3|H| mask coefficients, and nothing is uploaded.  Never reuse a seed. */
int32_t aleo_mi355x_fr_random_device(void* d_dst, size_t n, const uint8_t seed[32], uint64_t first_index, int32_t montgomery, void* stream);
int32_t aleo_mi355x_fr_random(void* out, size_t n, const uint8_t seed[32], uint64_t first_index);
/* dst[i] = c0 [i == 0] + sum_j coeffs[j] * terms[j][i], term j contributing for i < lens[j] (k <= 28 terms; d_terms / lens / coeffs_mont
 * host arrays; c0_mont may be NULL): the linear combinations opened at beta and gamma, in one pass.  dst may be one of the terms (same offset). */
int32_t aleo_mi355x_fr_lincomb_device(void* d_dst, size_t n, const void* c0_mont, const void* const* d_terms, const size_t* lens, const void* coeffs_mont, size_t k, void* stream);
/* The numerators of the two sumchecks from evaluations already in HBM [UPSTREAM-RECALL: varuna/ahp/prover/round_functions/{second,fourth}.rs]:
 *   first:  dst = r (z_a + eta_b z_b + eta_c z_a z_b) - t z                  (n = 4|H| values each)
 *   matrix: dst = sum_M delta_M (vv val_M - (alpha beta - beta row_M - alpha col_M + row_col_M) f_M)   (n = 2|K| values; d_index[M] points at
 *           row_M, with col_M, val_M, row_col_M following at index_stride elements each; consts_mont = delta_a, delta_b, delta_c,
 *           alpha beta, -alpha, -beta, v_H(alpha) v_H(beta): 7 x 32 bytes on the host; a NULL d_index[M] leaves matrix M out — matrices whose
 *           non-zero domains differ in size are done one call each).  dst may alias an operand. */
/* Two layout steps of the first two rounds, one launch for all instances [UPSTREAM-RECALL: round_functions/{first,second}.rs]:
 *   blind_rows:         dst row q (n + 1 coefficients) = src row q (n coefficients) + rho_q (X^n - 1); rows <= 24, rho_mont: rows x 32 bytes on the host
 *   sumcheck_operands:  per instance i the rows  z_i = w_i (X^n_x - 1) + x_i,  z_a,i,  z_b,i  on 4n coefficients, zero padded (dst: 3 rows of 4n per
 *                       instance; d_witness_polys: 3 rows of n + 1 per instance in the order w, z_a, z_b; d_x_polys: n_x coefficients per instance) */
int32_t aleo_mi355x_fr_blind_rows_device(void* d_dst, const void* d_src, size_t n, size_t rows, const void* rho_mont, void* stream);
int32_t aleo_mi355x_ahp_sumcheck_operands_device(void* d_dst, const void* d_witness_polys, const void* d_x_polys, size_t n, size_t n_x, size_t instances, void* stream);
int32_t aleo_mi355x_ahp_first_sumcheck_device(void* d_dst, size_t n, const void* d_r, const void* d_za, const void* d_zb, const void* d_t, const void* d_z,
                                              const void* eta_b_mont, const void* eta_c_mont, void* stream);
int32_t aleo_mi355x_ahp_matrix_sumcheck_device(void* d_dst, size_t n, const void* const* d_index, size_t index_stride, const void* const* d_f, const void* consts_mont, void* stream);
/* Division by (X - z): quotient[j-1] = s_j with s_j = p_j + z s_(j+1) (n - 1 coefficients, canonical Montgomery form) and, when
 * d_eval != NULL, p(z) = s_0 (32 bytes, device) — KZG10's witness polynomial (p(X) - p(z)) / (X - z)
 * [UPSTREAM-RECALL: polycommit/kzg10 compute_witness_polynomial].  z_mont: 32 bytes Montgomery Fr in HOST memory.
 * d_quotient must not alias d_poly; it may be NULL when only the evaluation is wanted (d_eval then non-NULL). */
int32_t aleo_mi355x_fr_divide_by_linear_device(void* d_quotient, void* d_eval, const void* d_poly, size_t n, const void* z_mont, void* stream);
/* KZG10::open of one polynomial at one point (without hiding): the witness polynomial is built on the device and committed against
 * the first n - 1 pinned powers.  out_affine104: the opening proof's `w` (host); out_eval_mont (host, may be NULL): p(z). */
int32_t aleo_mi355x_kzg_open_device(void* out_affine104, void* out_eval_mont, uint64_t handle, const void* d_poly_mont, size_t n, const void* z_mont, void* stream);
/* y = M * x over Fr for a CSR matrix in device memory (row_ptr: uint32[rows + 1], col_idx: uint32[nnz], vals: Montgomery
 * Fr[nnz]); x and y Montgomery Fr vectors.  The z_a = A z, z_b = B z products of the R1CS matrices before round 1
 * [UPSTREAM-RECALL: snark/varuna/ahp/prover/round_functions/first.rs].  Empty rows give 0. */
int32_t aleo_mi355x_fr_spmv_device(void* d_y, const void* d_row_ptr, const void* d_col_idx, const void* d_vals, const void* d_x, size_t rows, void* stream);

/* ---- wire formats of the results (host code, no GPU needed) -------------------------------------------
 * What snarkVM does when a commitment / proof leaves the prover [UPSTREAM-RECALL: utilities/src/serialize, curves/.../affine.rs
 * CanonicalSerialize, synthesizer/snark Proof::to_bytes_le + bech32m Display]; pinned byte for byte by the `proof1...` string the
 * reference's own test holds (/root/reference/wasm/src/programs/transaction.rs:100, round-tripped at :104-120).
 * Compressed G1: 48 bytes little-endian canonical x; last byte bit 7 = y is the lexicographically larger root, bit 6 = infinity.
 * decompress rejects non-canonical x, x off the curve and (check_subgroup != 0) points outside the prime-order subgroup. */
int32_t aleo_mi355x_g1_compress(void* out48, const void* affine104, size_t n);
int32_t aleo_mi355x_g1_decompress(void* out_affine104, const void* in48, size_t n, int32_t check_subgroup);
/* Fr <-> 32 canonical little-endian bytes (Fr::to_bytes_le / from_bytes_le); from_bytes rejects values >= r. */
int32_t aleo_mi355x_fr_to_bytes(void* out32, const void* fr_mont, size_t n);
int32_t aleo_mi355x_fr_from_bytes(void* out_fr_mont, const void* in32, size_t n);
/* bech32m (BIP-350) strings: "proof1...", and every other Aleo object id.  encode writes a NUL-terminated string;
 * decode: *len in = capacity of out, out = payload bytes; hrp_out may be NULL. */
int32_t aleo_mi355x_bech32m_encode(char* out, size_t cap, const char* hrp, const void* data, size_t len);
int32_t aleo_mi355x_bech32m_decode(void* out, size_t* len, char* hrp_out, size_t hrp_cap, const char* s);
/* The byte layout of a Varuna proof (Proof::to_bytes_le) from its parts; points as snarkVM Affine (104 bytes), field elements
 * Montgomery Fr.  Field order read off the reference's own proof (one circuit, one instance).  The parts list g_a, g_b, g_c
 * and their evaluations circuit by circuit; the BYTES carry every g_a, then every g_b, then every g_c (one vector per matrix, as upstream's
 * Commitments / Evaluations structs hold them [UPSTREAM-RECALL] — unverified for more than one circuit: no such reference vector exists; with one
 * circuit both orders coincide).  n_evaluations must be instances + 1 + 3 n_circuits.  *len: in = capacity of out, out = bytes written. */
typedef struct {
  const uint64_t* batch_sizes; size_t n_circuits;      /* instances per circuit */
  const void* witness_commitments;                      /* 3 per instance: w, z_a, z_b */
  const void* mask_poly;                                /* NULL = None (non-hiding) */
  const void* g_1; const void* h_1;
  const void* g_abc;                                    /* 3 per circuit: g_a, g_b, g_c */
  const void* h_2;
  const void* evaluations; size_t n_evaluations;        /* z_b per instance, g_1, then g_a, g_b, g_c per circuit */
  const void* sums;                                     /* 3 per circuit (third-round sums) */
  const void* opening_points; const void* opening_random_v; const uint8_t* opening_has_v; size_t n_openings;   /* KZG10 proofs: w, Option<random_v> */
} aleo_mi355x_proof_parts;
int32_t aleo_mi355x_proof_to_bytes(void* out, size_t* len, const aleo_mi355x_proof_parts* parts);

/* ---- one proof in one call: the host side of Varuna::prove_batch, native (aleo_amd/csrc/varuna.hip) -------------------------------
 * Replaces the CPU work snarkVM 0.14.5 does in algorithms/src/snark/varuna/{varuna.rs, ahp/prover/round_functions} [UPSTREAM-RECALL]
 * under /root/reference/rust/src/program/execute.rs:74 and transfer.rs:99 — for one circuit with 1..32 instances, upstream's Fiat-Shamir construction (the Poseidon
 * sponge over Fq, rate 2: aleo_mi355x_fs_* below) and the committer key the caller pinned (DESIGN.md 4d lists what differs from upstream).  The index is the prover-key material of the
 * circuit, built once (aleo_amd/varuna.py CircuitIndex does it through the entry points above) and described by device pointers:
 *   positions     host, uint32[n_vars]: index on H of every variable (public inputs on the subgroup X); positions_device: a copy in HBM (optional)
 *   a_*, b_*      device CSR of A, B with columns moved to positions on H and rows padded to n_h (uint32 row_ptr[n_h+1], col[], Montgomery val[])
 *   t_*           device CSR of the stacked transpose [A^T | B^T | C^T] (rows = positions on H, columns = matrix * n_h + row)
 *   vx_inv        device Fr[n_h]: 1 / v_X on H \ X, 0 on X
 *   k_evals       device Fr, matrix after matrix: row, col, val, row_col of M on K_M (|K_M| values each; matrix M starts at element 4 * (sum of the
 *                 earlier |K|));  k_idx  device uint32, matrix after matrix: row positions, column positions (|K_M| each; M starts at 2 * sum)
 *   k_polys       the same four polynomials per matrix as coefficients (layout of k_evals);  k2_evals  their values on the domain of size 2|K_M|
 *                 (2|K_M| each; M starts at element 8 * sum)
 *   vk_bytes      host: the circuit's verifying key as bytes (12 compressed index commitments, 5 domain sizes); vk_affine: the commitments as the
 *                 transcript absorbs them (12 x 104-byte G1Affine; NULL = decompress vk_bytes per proof)
 * committer_key: a pinned set holding powers[0..max_degree] and, from gamma_offset, at least 3 hiding powers; when lagrange_offset != 0 also, from
 * lagrange_offset, the n_h Lagrange-basis powers L_i(tau) G of the domain H followed by v_H(tau) G: w, z_a, z_b are then committed from their
 * EVALUATIONS (KZG10::commit_lagrange: the same group elements; a witness of bits stays a vector of small scalars for the MSM).
 * assignments: n_instances host pointers to n_vars x 32 bytes canonical (public variables first, z_0 = 1).  seed: 32 bytes of fresh
 * entropy per proof, the key of the proof's random stream (aleo_mi355x_fr_random_device; a repeated seed repeats the blinding).  out_proof / len: Proof::to_bytes_le layout, 901 + 176 (n_instances - 1) bytes; *len in = capacity.
 * Blocking; concurrent calls from several threads run on separate slots.  aleo_mi355x_varuna_last_timing: wall ms of the calling thread's
 * last proof: rounds 1..4, openings, total, time inside the five commitment calls, their host tails. */
typedef struct {
  uint64_t n_h, n_k_a, n_k_b, n_k_c, n_x, n_public, n_vars;      /* |H|, the non-zero domains |K_A|, |K_B|, |K_C|, |X| */
  uint64_t committer_key, max_degree, gamma_offset, lagrange_offset;
  const uint32_t* positions;
  const void* positions_device;      /* the same array in HBM (may be NULL: the assignment is then laid out on H by the host) */
  const void *a_row_ptr, *a_col, *a_val, *b_row_ptr, *b_col, *b_val, *t_row_ptr, *t_col, *t_val;
  const void *vx_inv, *k_evals, *k_idx, *k_polys, *k2_evals;
  const void* vk_bytes; size_t vk_len;
  const void* vk_affine;             /* the twelve index commitments as 104-byte G1Affine (what the transcript absorbs); may be NULL: decompressed from vk_bytes per proof */
  uint64_t max_row[3];               /* hints (0 = unknown): an upper bound of the longest row of a_*, b_*, t_* — the prover skips the launches for long (> 64) / huge (> 8192) rows of a
                                      * sparse product that cannot have any (index_build fills them; a caller-built struct may leave them 0) */
} aleo_mi355x_varuna_index;
/* The index built by the library itself from the R1CS (AHPForR1CS::index shape: the arithmetisation above + the twelve index commitments)
 * and kept in HBM under a handle.  Matrices: CSR over the variables (uint32 row_ptr[n_constraints + 1], uint32 col[nnz] = variable index
 * with the n_public public variables first, val[nnz] canonical 32-byte Fr), host memory.  Domains: |X| = 2^ceil(lg n_public), |H| = the power
 * of two >= max(n_constraints, |X| + n_private, 2|X|), |K_M| = the power of two >= the non-zero count of M (>= 2), one per matrix
 * (domain_flags 1), all three the largest of them (2), or (0) shared below 2^18 — latency-bound rounds, one batched transform beats three short
 * ones — and per matrix from there on.  The
 * committer key (powers[0..max_degree], >= 3 hiding powers from gamma_offset) must hold max(3|H|, max |K_M|) powers and stays pinned while the index lives.
 * index_export fills the struct view (pointers owned by the library, valid until index_free; the transposed matrices t_* list the entries of a column in no
 * fixed order — they are written through an atomic cursor — which the exact field sums over them do not see, but two builds of one circuit may differ there);
 * index_vk copies the bytes the transcript absorbs
 * first: 12 compressed index commitments (row, col, val, row_col of A, B, C) then |H|, |K_A|, |K_B|, |K_C|, |X| as u64 LE.  prove_indexed = varuna_prove. */
typedef struct { const uint32_t* row_ptr; const uint32_t* col; const void* val; } aleo_mi355x_r1cs_matrix;
int32_t aleo_mi355x_varuna_index_build(uint64_t* index_handle, uint64_t committer_key, uint64_t max_degree, uint64_t gamma_offset, uint64_t lagrange_offset,
                                       const aleo_mi355x_r1cs_matrix abc[3], size_t n_constraints, size_t n_public, size_t n_private, uint32_t domain_flags);
int32_t aleo_mi355x_varuna_index_export(uint64_t index_handle, aleo_mi355x_varuna_index* out);
int32_t aleo_mi355x_varuna_index_vk(uint64_t index_handle, void* out, size_t* len);
int32_t aleo_mi355x_varuna_index_free(uint64_t index_handle);
int32_t aleo_mi355x_varuna_prove_indexed(uint64_t index_handle, const void* const* assignments, size_t n_instances, const uint8_t seed[32], void* out_proof, size_t* len);
int32_t aleo_mi355x_varuna_prove(const aleo_mi355x_varuna_index* index, const void* const* assignments, size_t n_instances, const uint8_t seed[32], void* out_proof, size_t* len);
/* One proof over SEVERAL circuits — upstream's `Varuna::prove_batch(keys_to_constraints: BTreeMap<&ProvingKey, &[Assignment]>)`, what
 * `Trace::prove_execution` / `prove_fee` build from the transitions of a transaction (/root/reference/rust/src/program/execute.rs:74).
 * index_handles: 1..32 indexes built against ONE committer key (same max_degree / gamma_offset), in the order the proof lists them;
 * n_instances[j]: the assignments of circuit j — at most 32 over all circuits, in any split (one proof covers one transaction: snarkVM allows 32 transitions); assignments: the pointers of circuit 0's instances, then circuit 1's, ...
 * The cap of 32 instances is this library's (the protocol has none; snarkVM's transactions carry at most 32 transitions): a request beyond it is
 * refused with ALEO_MI355X_ERR_BAD_ARG before any launch and the caller falls back (CPU prover).  One proof is one object: the library never splits
 * a request into several proofs on its own.
 * The circuits share the transcript and every challenge, one mask / g_1 / h_1 over the largest constraint domain, one h_2 over the largest non-zero
 * domain and the two openings; a smaller circuit enters behind the selector v_{H*} / v_{H_j}.  out_proof: Proof::to_bytes_le layout —
 * batch sizes, 3 witness commitments per instance, mask, g_1, h_1, every g_a, every g_b, every g_c (one vector per matrix over the circuits,
 * [UPSTREAM-RECALL]; unverifiable offline for more than one circuit — the reference's proof string has one), h_2, evaluations (z_b's, g_1,
 * then g_a / g_b / g_c likewise), sums per circuit, openings.  With one circuit
 * the bytes are those of aleo_mi355x_varuna_prove_indexed.  ALEO_MI355X_ERR_UNSATISFIED if any assignment violates its circuit. */
int32_t aleo_mi355x_varuna_prove_batch_indexed(const uint64_t* index_handles, size_t n_circuits, const void* const* assignments, const size_t* n_instances, const uint8_t seed[32],
                                               void* out_proof, size_t* len);
/* Several INDEPENDENT proofs in one call, proved in lockstep (a serving path: the reference's dev server proves concurrent requests from
 * tokio::task::spawn_blocking threads, /root/reference/rust/develop/src/routes.rs:119,149,229 — a front end that collects them calls this instead of
 * n separate proves).  Every request is one prove_batch_indexed call's worth — its own circuits, assignments, 32-byte seed, transcript and output —
 * and comes out byte for byte as that call would give it; what the requests share is every commitment launch chain: round r of all of them is ONE
 * batched MSM, so its sort, slice tree, reduction, host tail and stream synchronisation are paid once per round instead of once per proof
 * (eight 2^15-constraint proofs: see DESIGN.md 4d).  All indexes must belong to one committer key.  len: in = capacity of out_proof, out = bytes
 * written.  status (out): per request — a request that fails (unsatisfied assignment: ALEO_MI355X_ERR_UNSATISFIED; bad argument) drops out, the
 * others complete; the call itself returns non-zero only when nothing could be started or a device call failed.  1..64 requests.
 * Between the commitments the proofs run on up to ALEO_MI355X_LOCKSTEP_WORKERS (default 4, 1..9) threads of the library's own, each with a
 * borrowed stream of the device: the calling thread blocks until all proofs are written. */
typedef struct {
  const uint64_t* index_handles; size_t n_circuits; const void* const* assignments; const size_t* n_instances; const uint8_t* seed;
  void* out_proof; size_t len; int32_t status;
} aleo_mi355x_prove_request;
int32_t aleo_mi355x_varuna_prove_many(aleo_mi355x_prove_request* requests, size_t n_requests);
int32_t aleo_mi355x_varuna_last_timing(double* out_ms, int32_t cap);

/* snarkVM's Poseidon on the host (no GPU needed).  Parameters: Grain LFSR, alpha = 17, 8 full + 31 partial rounds, capacity 1 [UPSTREAM-RECALL:
 * snarkvm-fields poseidon_default; pinned over Fr by the reference's private-key ciphertext and account vectors, tests/test_poseidon.py].
 *   poseidon_hash_fr: `Network::hash_many_psd{2,4,8}` — rate = 2, 4 or 8; inputs / outputs canonical 32-byte Fr; preimage = [domain
 *   "AleoPoseidon{rate}", n_inputs, 0.., inputs] (what /root/reference/rust/src/account/encryptor.rs:37-67 calls through hash_psd2).
 *   fs_*: the prover's Fiat-Shamir sponge — PoseidonSponge<Fq, 2, 1> behind upstream's AlgebraicSponge interface: bytes (bits of every byte most
 *   significant first, 376-bit chunks), G1Affine points (x, y Montgomery, stride 104 or 96; infinity as (0, 1)), Fr elements as non-native limbs
 *   (canonical input, 5 x 51 bits packed two per Fq element), challenges (canonical output): `short_` = 0: 252 bits each from ONE squeeze of
 *   ceil(252 n / 376) elements (squeeze_nonnative_field_elements(n)); 1: 168 bits (squeeze_short_nonnative_field_elements(n)).
 *   aleo_mi355x_varuna_prove* run this sponge inside the library; the entry points exist for a host side that drives the rounds itself
 *   (aleo_amd/varuna.py) and for verifiers. */
int32_t aleo_mi355x_poseidon_hash_fr(uint32_t rate, const void* inputs, size_t n_inputs, void* out, size_t n_out);
/* the round constants ark[39][rate + 1] and the matrix mds[rate + 1][rate + 1] (row-major, canonical 32-byte Fr) of that hash: what a circuit constraining it needs */
int32_t aleo_mi355x_poseidon_parameters_fr(uint32_t rate, void* ark_out, void* mds_out);
int32_t aleo_mi355x_fs_new(uint64_t* sponge);
int32_t aleo_mi355x_fs_free(uint64_t sponge);
int32_t aleo_mi355x_fs_absorb_bytes(uint64_t sponge, const void* data, size_t len);
int32_t aleo_mi355x_fs_absorb_g1(uint64_t sponge, const void* affine, size_t stride, size_t count);
int32_t aleo_mi355x_fs_absorb_fr(uint64_t sponge, const void* fr_canonical, size_t count);
int32_t aleo_mi355x_fs_squeeze_fr(uint64_t sponge, void* out_canonical, size_t count, int32_t short_);

/* Element-wise field products on the device (host pointers): r[i] = a[i]*b[i], Montgomery form, canonical
 * output.  Used by the parity tests to pin the device arithmetic against the oracle limb for limb; when a and b
 * are the same buffer the dedicated squaring block runs instead of the general product. */
int32_t aleo_mi355x_fq_mul(void* r, const void* a, const void* b, size_t n);
int32_t aleo_mi355x_fr_mul(void* r, const void* a, const void* b, size_t n);

/* Test hook: `lanes` device lanes each run `steps` mixed additions of pseudo-random field elements through the 28-bit-limb
 * formulas of the accumulation kernel and through the 32-bit ones, comparing every intermediate as canonical residues
 * (and that both refuse P == acc).  *failures = number of lanes that disagreed. */
int32_t aleo_mi355x_selftest_madd28(uint32_t lanes, uint32_t steps, uint64_t seed, uint32_t* failures);
/* The lane-quad XYZZ addition of the reduction chains against the lane-pair one on `ops` random operand pairs and the special cases
 * (identity operands, equal points, opposite points): *failures = number of disagreements (mod q, all four coordinates). */
int32_t aleo_mi355x_selftest_addquad(uint32_t ops, uint64_t seed, uint32_t* failures);

/* The lane-pair Fq2 arithmetic of the G2 path (g2.hip: components of an Fq2 value across two lanes, 28-bit limbs) against the one-lane 32-bit code on
 * chains over n_points affine G2 points (192-byte rows, host memory, >= 3 of them, all on the curve): sums, a sum with a shared operand, a doubling, the
 * same-point case of the addition.  failures2[0] = pairs that disagreed, failures2[1] = OR of the failing steps (1 a, 2 s, 4 u, 8 2u, 16 u + u). */
int32_t aleo_mi355x_selftest_g2pair(const void* affine192, uint32_t n_points, uint32_t n_pairs, uint32_t* failures2);

/* Host arithmetic (no device needed): the inversion the host tails use — Bernstein-Yang divsteps on 62-bit limbs (csrc/host_modinv.hpp) — against the Fermat
 * chain a^(p-2) on `count` pseudo-random elements of Fq and of Fr plus the edge values 0, 1, 2, p - 1, p - 2: *failures = results that differ; ns_per_inverse
 * (optional, 4 doubles): divsteps / Fermat for Fq, divsteps / Fermat for Fr. */
int32_t aleo_mi355x_selftest_host_inverse(uint32_t count, uint64_t seed, uint32_t* failures, double* ns_per_inverse);

/* Per-call instrumentation of the calling thread's most recent MSM: milliseconds per phase
 * [0] total, [1] digit/sort, [2] bucket accumulation incl. slice tree, [3] bucket reduction, [4] host tail,
 * [5] the bucket-accumulation kernel alone (the dominant kernel bench.py prices against the roofline): mean duration of its launches,
 * [6] how many launches of it the call made (2 or 3 when host scalars went up in chunks whose accumulations are seeded from one another and share one reduction, else 1).
 * Returns the number of doubles written (<= cap). */
int32_t aleo_mi355x_last_msm_timing(double* out_ms, int32_t cap);

/* Routing thresholds for the drop-in: below these sizes the Rust arms keep the CPU path — a 2^6..2^12 host-buffer call costs two PCIe
 * copies and 20-50 us of launches against microseconds on a core.  Defaults = the crossovers measured for the cold one-shot calls against the CPU path on
 * the GPU box's host cores (bench.py cpu_baseline.crossover; profiles/r04_crossover.json); ALEO_MI355X_MIN_MSM / ALEO_MI355X_MIN_NTT override them. */
size_t aleo_mi355x_min_msm(void);
size_t aleo_mi355x_min_ntt(void);
const char* aleo_mi355x_strerror(int32_t status);
const char* aleo_mi355x_last_error(void);   /* thread-local detail string of the last failure */
const char* aleo_mi355x_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ALEO_MI355X_H */

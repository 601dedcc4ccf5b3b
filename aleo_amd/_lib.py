"""ctypes binding of libaleo_mi355x.so (include/aleo_mi355x.h).  There is no CPU fallback: if the HIP library
is missing or a call fails, an exception is raised (the Rust caller would fall back to snarkVM's CPU path; this
package never does, so a silent fallback cannot hide behind a green test)."""
from __future__ import annotations
import ctypes, os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ALEO_MI355X_LIB: another build of the same library (the probe builds of tools/ntt_phase_probe.sh, an A/B build) — never a fallback: it must exist and export every symbol
LIB_PATH = os.environ.get('ALEO_MI355X_LIB') or os.path.join(_HERE, 'lib', 'libaleo_mi355x.so')

EXPORTS = [
    'aleo_mi355x_init', 'aleo_mi355x_msm_g1', 'aleo_mi355x_bases_pin', 'aleo_mi355x_bases_unpin',
    'aleo_mi355x_bases_generate', 'aleo_mi355x_bases_from_scalars', 'aleo_mi355x_fr_lin_device', 'aleo_mi355x_msm_g1_device_sparse', 'aleo_mi355x_bases_precompute_range', 'aleo_mi355x_kzg_commit_segments_sparse_device', 'aleo_mi355x_fr_blind_rows_device', 'aleo_mi355x_ahp_sumcheck_operands_device', 'aleo_mi355x_varuna_prove', 'aleo_mi355x_varuna_index_build', 'aleo_mi355x_varuna_index_export', 'aleo_mi355x_varuna_index_vk', 'aleo_mi355x_varuna_index_free', 'aleo_mi355x_varuna_prove_indexed', 'aleo_mi355x_varuna_prove_batch_indexed', 'aleo_mi355x_varuna_last_timing', 'aleo_mi355x_fr_random_device', 'aleo_mi355x_fr_lincomb_device', 'aleo_mi355x_ahp_first_sumcheck_device', 'aleo_mi355x_ahp_matrix_sumcheck_device', 'aleo_mi355x_fr_powers_device', 'aleo_mi355x_fr_gather_mul_device', 'aleo_mi355x_fr_eval_batch_device', 'aleo_mi355x_bases_download', 'aleo_mi355x_bases_info', 'aleo_mi355x_bases_precompute',
    'aleo_mi355x_msm_g1_pinned', 'aleo_mi355x_msm_g1_device', 'aleo_mi355x_g1_sum', 'aleo_mi355x_msm_g2', 'aleo_mi355x_g2_sum', 'aleo_mi355x_bases_g2_pin', 'aleo_mi355x_bases_g2_unpin', 'aleo_mi355x_msm_g2_pinned', 'aleo_mi355x_ntt_fr',
    'aleo_mi355x_ntt_fr_device', 'aleo_mi355x_ntt_fr_batch_device', 'aleo_mi355x_ntt_fr_from_device', 'aleo_mi355x_fr_grid_scale_device', 'aleo_mi355x_kzg_commit', 'aleo_mi355x_kzg_commit_device', 'aleo_mi355x_kzg_commit_hiding',
    'aleo_mi355x_msm_g1_batch_device', 'aleo_mi355x_kzg_commit_batch_device', 'aleo_mi355x_kzg_commit_batch', 'aleo_mi355x_kzg_commit_segments', 'aleo_mi355x_kzg_commit_segments_device',
    'aleo_mi355x_fr_vec_op_device', 'aleo_mi355x_fr_batch_inverse_device', 'aleo_mi355x_fr_spmv_device', 'aleo_mi355x_fr_divide_by_linear_device', 'aleo_mi355x_kzg_open_device', 'aleo_mi355x_fq_mul',
    'aleo_mi355x_fr_mul', 'aleo_mi355x_selftest_madd28', 'aleo_mi355x_selftest_addquad', 'aleo_mi355x_selftest_g2pair', 'aleo_mi355x_last_msm_timing', 'aleo_mi355x_strerror', 'aleo_mi355x_last_error',
    'aleo_mi355x_version',
    'aleo_mi355x_g1_compress', 'aleo_mi355x_g1_decompress', 'aleo_mi355x_fr_to_bytes', 'aleo_mi355x_fr_from_bytes',
    'aleo_mi355x_bech32m_encode', 'aleo_mi355x_bech32m_decode', 'aleo_mi355x_proof_to_bytes',
    'aleo_mi355x_poseidon_hash_fr', 'aleo_mi355x_fs_new', 'aleo_mi355x_fs_free', 'aleo_mi355x_fs_absorb_bytes', 'aleo_mi355x_fs_absorb_g1',
    'aleo_mi355x_fs_absorb_fr', 'aleo_mi355x_fs_squeeze_fr', 'aleo_mi355x_fr_random', 'aleo_mi355x_poseidon_parameters_fr',
    'aleo_mi355x_init_device', 'aleo_mi355x_device_count', 'aleo_mi355x_peer_info', 'aleo_mi355x_min_msm', 'aleo_mi355x_min_ntt', 'aleo_mi355x_kzg_commit_segments_sharded_device', 'aleo_mi355x_kzg_commit_batch_sharded_device', 'aleo_mi355x_bases_attach_shards', 'aleo_mi355x_bases_shard_transforms', 'aleo_mi355x_bases_pin_sharded', 'aleo_mi355x_bases_generate_sharded', 'aleo_mi355x_bases_unpin_sharded',
    'aleo_mi355x_bases_sharded_info', 'aleo_mi355x_msm_g1_sharded', 'aleo_mi355x_fr_transpose_device', 'aleo_mi355x_ntt_fr_sharded', 'aleo_mi355x_ntt_fr_sharded_device', 'aleo_mi355x_selftest_host_inverse', 'aleo_mi355x_varuna_prove_many',
]


class AleoMi355xError(RuntimeError):
    status = None


class UnsatisfiedAssignment(AleoMi355xError):
    """ALEO_MI355X_ERR_UNSATISFIED: a prover was handed an assignment that does not satisfy its circuit."""
    status = 6


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise AleoMi355xError(f'{LIB_PATH} not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                              '(aleo_amd/csrc/build.sh).  There is no CPU fallback.')
    # PyTorch-ROCm is the plumbing for device memory / streams / torch.distributed and ships its own HIP runtime
    # (libamdhip64.so with the same SONAME as /opt/rocm's).  Load torch's first so one process never ends up with
    # /opt/rocm's runtime under torch: that combination makes torch report "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, i32, u32, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint64
    sig = {
        'aleo_mi355x_init': ([i32], i32),
        'aleo_mi355x_init_device': ([i32], i32),
        'aleo_mi355x_device_count': ([ctypes.POINTER(i32), ctypes.POINTER(i32)], i32),
        'aleo_mi355x_peer_info': ([ctypes.POINTER(i32), ctypes.POINTER(i32)], i32),
        'aleo_mi355x_min_msm': ([], sz),
        'aleo_mi355x_min_ntt': ([], sz),
        'aleo_mi355x_kzg_commit_segments_sharded_device': ([vp, sz, u64, vp, sz, vp], i32),
        'aleo_mi355x_kzg_commit_batch_sharded_device': ([vp, u64, vp, vp, sz, vp], i32),
        'aleo_mi355x_bases_attach_shards': ([u64, u64, sz], i32),
        'aleo_mi355x_bases_shard_transforms': ([u64, sz], i32),
        'aleo_mi355x_bases_pin_sharded': ([vp, sz, sz, ctypes.POINTER(i32), sz, i32, ctypes.POINTER(u64)], i32),
        'aleo_mi355x_bases_generate_sharded': ([vp, u64, sz, ctypes.POINTER(i32), sz, i32, ctypes.POINTER(u64)], i32),
        'aleo_mi355x_bases_unpin_sharded': ([u64], i32),
        'aleo_mi355x_bases_sharded_info': ([u64, ctypes.POINTER(u64), i32], i32),
        'aleo_mi355x_msm_g1_sharded': ([vp, u64, vp, sz, vp], i32),
        'aleo_mi355x_fr_transpose_device': ([vp, vp, u64, u64, vp], i32),
        'aleo_mi355x_ntt_fr_sharded': ([vp, u32, i32, i32, ctypes.POINTER(i32), sz], i32),
        'aleo_mi355x_ntt_fr_sharded_device': ([vp, u32, i32, i32, ctypes.POINTER(i32), sz, vp], i32),
        'aleo_mi355x_selftest_host_inverse': ([u32, ctypes.c_uint64, ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_double)], i32),
        'aleo_mi355x_msm_g1': ([vp, vp, sz, vp, sz], i32),
        'aleo_mi355x_bases_pin': ([vp, sz, sz, ctypes.POINTER(u64)], i32),
        'aleo_mi355x_bases_unpin': ([u64], i32),
        'aleo_mi355x_bases_generate': ([vp, u64, sz, ctypes.POINTER(u64)], i32),
        'aleo_mi355x_bases_from_scalars': ([vp, vp, sz, ctypes.POINTER(u64)], i32),
        'aleo_mi355x_fr_lin_device': ([vp, sz, vp, vp, vp, vp, vp, vp], i32),
        'aleo_mi355x_fr_powers_device': ([vp, sz, vp, vp, vp], i32),
        'aleo_mi355x_msm_g1_device_sparse': ([vp, u64, vp, sz, vp], i32),
        'aleo_mi355x_bases_precompute_range': ([u64, sz, sz, i32], i32),
        'aleo_mi355x_kzg_commit_segments_sparse_device': ([vp, sz, u64, vp, sz, vp], i32),
        'aleo_mi355x_fr_blind_rows_device': ([vp, vp, sz, sz, vp, vp], i32),
        'aleo_mi355x_ahp_sumcheck_operands_device': ([vp, vp, vp, sz, sz, sz, vp], i32),
        'aleo_mi355x_varuna_prove': ([vp, ctypes.POINTER(vp), sz, vp, vp, ctypes.POINTER(sz)], i32),
        'aleo_mi355x_varuna_index_build': ([ctypes.POINTER(u64), u64, u64, u64, u64, vp, sz, sz, sz, ctypes.c_uint32], i32),
        'aleo_mi355x_varuna_index_export': ([u64, vp], i32),
        'aleo_mi355x_varuna_index_vk': ([u64, vp, ctypes.POINTER(sz)], i32),
        'aleo_mi355x_varuna_index_free': ([u64], i32),
        'aleo_mi355x_varuna_prove_indexed': ([u64, ctypes.POINTER(vp), sz, vp, vp, ctypes.POINTER(sz)], i32),
        'aleo_mi355x_varuna_prove_batch_indexed': ([ctypes.POINTER(u64), sz, ctypes.POINTER(vp), ctypes.POINTER(sz), vp, vp, ctypes.POINTER(sz)], i32),
        'aleo_mi355x_varuna_last_timing': ([ctypes.POINTER(ctypes.c_double), i32], i32),
        'aleo_mi355x_varuna_prove_many': ([vp, sz], i32),
        'aleo_mi355x_fr_random_device': ([vp, sz, vp, u64, i32, vp], i32),
        'aleo_mi355x_fr_random': ([vp, sz, vp, u64], i32),
        'aleo_mi355x_poseidon_hash_fr': ([u32, vp, sz, vp, sz], i32),
        'aleo_mi355x_poseidon_parameters_fr': ([u32, vp, vp], i32),
        'aleo_mi355x_fs_new': ([ctypes.POINTER(u64)], i32),
        'aleo_mi355x_fs_free': ([u64], i32),
        'aleo_mi355x_fs_absorb_bytes': ([u64, vp, sz], i32),
        'aleo_mi355x_fs_absorb_g1': ([u64, vp, sz, sz], i32),
        'aleo_mi355x_fs_absorb_fr': ([u64, vp, sz], i32),
        'aleo_mi355x_fs_squeeze_fr': ([u64, vp, sz, i32], i32),
        'aleo_mi355x_fr_lincomb_device': ([vp, sz, vp, ctypes.POINTER(vp), ctypes.POINTER(sz), vp, sz, vp], i32),
        'aleo_mi355x_ahp_first_sumcheck_device': ([vp, sz, vp, vp, vp, vp, vp, vp, vp, vp], i32),
        'aleo_mi355x_ahp_matrix_sumcheck_device': ([vp, sz, ctypes.POINTER(vp), sz, ctypes.POINTER(vp), vp, vp], i32),
        'aleo_mi355x_fr_gather_mul_device': ([vp, sz, vp, vp, vp, vp, vp, vp], i32),
        'aleo_mi355x_fr_eval_batch_device': ([vp, ctypes.POINTER(vp), ctypes.POINTER(sz), vp, sz, vp], i32),
        'aleo_mi355x_bases_download': ([u64, sz, sz, vp], i32),
        'aleo_mi355x_bases_info': ([u64, ctypes.POINTER(u64), i32], i32),
        'aleo_mi355x_bases_precompute': ([u64], i32),
        'aleo_mi355x_msm_g1_pinned': ([vp, u64, vp, sz], i32),
        'aleo_mi355x_msm_g1_device': ([vp, u64, vp, sz, vp], i32),
        'aleo_mi355x_g1_sum': ([vp, vp, sz], i32),
        'aleo_mi355x_msm_g2': ([vp, vp, sz, vp, sz], i32),
        'aleo_mi355x_g2_sum': ([vp, vp, sz], i32),
        'aleo_mi355x_bases_g2_pin': ([vp, sz, sz, ctypes.POINTER(u64)], i32),
        'aleo_mi355x_bases_g2_unpin': ([u64], i32),
        'aleo_mi355x_msm_g2_pinned': ([vp, u64, vp, sz], i32),
        'aleo_mi355x_ntt_fr': ([vp, u32, i32, i32, i32], i32),
        'aleo_mi355x_ntt_fr_device': ([vp, u32, i32, i32, i32, vp], i32),
        'aleo_mi355x_ntt_fr_batch_device': ([vp, u32, sz, i32, i32, i32, vp], i32),
        'aleo_mi355x_ntt_fr_from_device': ([vp, vp, sz, sz, u32, sz, i32, i32, vp], i32),
        'aleo_mi355x_fr_grid_scale_device': ([vp, u32, u64, u64, u64, u64, u64, i32, i32, vp], i32),
        'aleo_mi355x_kzg_commit': ([vp, u64, vp, sz], i32),
        'aleo_mi355x_kzg_commit_device': ([vp, u64, vp, sz, vp], i32),
        'aleo_mi355x_kzg_commit_hiding': ([vp, u64, vp, sz, u64, vp, sz], i32),
        'aleo_mi355x_msm_g1_batch_device': ([vp, u64, vp, vp, sz, vp], i32),
        'aleo_mi355x_kzg_commit_batch_device': ([vp, u64, vp, vp, sz, vp], i32),
        'aleo_mi355x_kzg_commit_batch': ([vp, u64, vp, vp, sz], i32),
        'aleo_mi355x_kzg_commit_segments': ([vp, sz, u64, vp, sz], i32),
        'aleo_mi355x_kzg_commit_segments_device': ([vp, sz, u64, vp, sz, vp], i32),
        'aleo_mi355x_fr_vec_op_device': ([vp, vp, vp, sz, i32, vp], i32),
        'aleo_mi355x_fr_batch_inverse_device': ([vp, sz, vp], i32),
        'aleo_mi355x_fr_spmv_device': ([vp, vp, vp, vp, vp, sz, vp], i32),
        'aleo_mi355x_fr_divide_by_linear_device': ([vp, vp, vp, sz, vp, vp], i32),
        'aleo_mi355x_kzg_open_device': ([vp, vp, u64, vp, sz, vp, vp], i32),
        'aleo_mi355x_fq_mul': ([vp, vp, vp, sz], i32),
        'aleo_mi355x_fr_mul': ([vp, vp, vp, sz], i32),
        'aleo_mi355x_selftest_madd28': ([u32, u32, u64, ctypes.POINTER(u32)], i32),
        'aleo_mi355x_selftest_addquad': ([u32, u64, ctypes.POINTER(u32)], i32),
        'aleo_mi355x_selftest_g2pair': ([vp, u32, u32, ctypes.POINTER(u32)], i32),
        'aleo_mi355x_last_msm_timing': ([ctypes.POINTER(ctypes.c_double), i32], i32),
        'aleo_mi355x_strerror': ([i32], ctypes.c_char_p),
        'aleo_mi355x_last_error': ([], ctypes.c_char_p),
        'aleo_mi355x_version': ([], ctypes.c_char_p),
        'aleo_mi355x_g1_compress': ([vp, vp, sz], i32),
        'aleo_mi355x_g1_decompress': ([vp, vp, sz, i32], i32),
        'aleo_mi355x_fr_to_bytes': ([vp, vp, sz], i32),
        'aleo_mi355x_fr_from_bytes': ([vp, vp, sz], i32),
        'aleo_mi355x_bech32m_encode': ([ctypes.c_char_p, sz, ctypes.c_char_p, vp, sz], i32),
        'aleo_mi355x_bech32m_decode': ([vp, ctypes.POINTER(sz), ctypes.c_char_p, sz, ctypes.c_char_p], i32),
        'aleo_mi355x_proof_to_bytes': ([vp, ctypes.POINTER(sz), vp], i32),
    }
    for name, (args, res) in sig.items():
        f = getattr(L, name); f.argtypes = args; f.restype = res
    _LIB = L
    return L


def check(status: int, what: str):
    if status != 0:
        L = lib()
        e = (UnsatisfiedAssignment if status == UnsatisfiedAssignment.status else AleoMi355xError)(
            f'{what}: {L.aleo_mi355x_strerror(status).decode()} [{L.aleo_mi355x_last_error().decode()}]')
        e.status = status
        raise e


def seed32(seed=None):
    """A proof's 32-byte seed as a ctypes buffer.  None draws 32 bytes from the OS (what callers should do: a repeated seed repeats the blinding);
    bytes are taken as they are; an int (tests, benchmarks: reproducible proofs) is read as 32 little-endian bytes."""
    if seed is None: raw = os.urandom(32)
    elif isinstance(seed, (bytes, bytearray)):
        if len(seed) != 32: raise ValueError('a proof seed is 32 bytes')
        raw = bytes(seed)
    else: raw = int(seed).to_bytes(32, 'little')
    return (ctypes.c_uint8 * 32).from_buffer_copy(raw)

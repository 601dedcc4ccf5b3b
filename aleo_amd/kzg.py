"""Host-side mirror of snarkvm_algorithms::polycommit::kzg10::KZG10::commit (the non-hiding MSM part).

Reference (snarkVM 0.14.5 algorithms/src/polycommit/kzg10/mod.rs [UPSTREAM-RECALL]): coefficients (Montgomery Fr)
-> canonical bigints -> VariableBase::msm(powers_of_beta_g[..len], coeffs) -> to_affine().  The conversion runs on the
device, so a polynomial that is already in HBM (e.g. straight from an iNTT) is committed without a host round trip."""
from __future__ import annotations
import ctypes
import numpy as np
from ._lib import lib, check
from .msm import PinnedBases


def _p(a): return a.ctypes.data_as(ctypes.c_void_p)


class KZG10:
    @staticmethod
    def commit(powers: PinnedBases, coeffs_mont: np.ndarray) -> np.ndarray:
        c = np.ascontiguousarray(coeffs_mont, dtype=np.uint64).reshape(-1, 4)
        n = c.shape[0]
        while n and not c[n - 1].any(): n -= 1          # skip leading zeros, as the reference does
        out = np.zeros(104, dtype=np.uint8)
        check(lib().aleo_mi355x_kzg_commit(_p(out), powers.handle, _p(c), n), 'kzg_commit')
        return out

    @staticmethod
    def commit_device(powers: PinnedBases, d_coeffs_ptr: int, n: int, stream: int = 0) -> np.ndarray:
        out = np.zeros(104, dtype=np.uint8)
        check(lib().aleo_mi355x_kzg_commit_device(_p(out), powers.handle, ctypes.c_void_p(d_coeffs_ptr), n,
                                                  ctypes.c_void_p(stream)), 'kzg_commit_device')
        return out

    @staticmethod
    def commit_batch(powers: PinnedBases, polys) -> np.ndarray:
        """The commitments of one prover round (host coefficient vectors, Montgomery): one call, shared launches.  uint8[k,104]."""
        cs = []
        for c in polys:
            c = np.ascontiguousarray(c, dtype=np.uint64).reshape(-1, 4); n = c.shape[0]
            while n and not c[n - 1].any(): n -= 1      # skip leading zeros, as the reference does
            cs.append((c, n))
        k = len(cs)
        ptrs = (ctypes.c_void_p * max(k, 1))(*[c.ctypes.data for c, _ in cs]); ln = (ctypes.c_size_t * max(k, 1))(*[n for _, n in cs])
        out = np.zeros((k, 104), dtype=np.uint8)
        check(lib().aleo_mi355x_kzg_commit_batch(_p(out), powers.handle, ptrs, ln, k), 'kzg_commit_batch')
        return out

    @staticmethod
    def commit_batch_device(powers: PinnedBases, d_ptrs, lens, stream: int = 0) -> np.ndarray:
        """Same with the coefficient vectors already in HBM (k device pointers, k lengths)."""
        k = len(d_ptrs)
        ptrs = (ctypes.c_void_p * max(k, 1))(*[int(x) for x in d_ptrs]); ln = (ctypes.c_size_t * max(k, 1))(*[int(x) for x in lens])
        out = np.zeros((k, 104), dtype=np.uint8)
        check(lib().aleo_mi355x_kzg_commit_batch_device(_p(out), powers.handle, ptrs, ln, k, ctypes.c_void_p(stream)), 'kzg_commit_batch_device')
        return out

    @staticmethod
    def commit_batch_sharded_device(sharded, d_ptrs, lens, stream: int = 0) -> np.ndarray:
        """commit_batch_device against a sharded copy of the powers (aleo_mi355x_kzg_commit_batch_sharded_device)."""
        k = len(d_ptrs)
        ptrs = (ctypes.c_void_p * max(k, 1))(*[int(x) for x in d_ptrs]); ln = (ctypes.c_size_t * max(k, 1))(*[int(x) for x in lens])
        out = np.zeros((k, 104), dtype=np.uint8)
        check(lib().aleo_mi355x_kzg_commit_batch_sharded_device(_p(out), sharded.handle, ptrs, ln, k, ctypes.c_void_p(stream)), 'kzg_commit_batch_sharded_device')
        return out

    @staticmethod
    def open_device(powers: PinnedBases, d_poly_ptr: int, n: int, z_mont: np.ndarray, stream: int = 0):
        """KZG10::open (non-hiding): returns (w as snarkVM Affine uint8[104], p(z) as Montgomery uint64[4])."""
        z = np.ascontiguousarray(z_mont, dtype=np.uint64).reshape(4)
        out = np.zeros(104, dtype=np.uint8); ev = np.zeros(4, dtype=np.uint64)
        check(lib().aleo_mi355x_kzg_open_device(_p(out), _p(ev), powers.handle, ctypes.c_void_p(d_poly_ptr), n, _p(z), ctypes.c_void_p(stream)), 'kzg_open_device')
        return out, ev

    @staticmethod
    def commit_hiding(powers: PinnedBases, coeffs_mont: np.ndarray, gamma_powers: PinnedBases, blinding_mont: np.ndarray) -> np.ndarray:
        """KZG10::commit with hiding_bound: adds msm(powers_of_beta_times_gamma_g, random polynomial)."""
        c = np.ascontiguousarray(coeffs_mont, dtype=np.uint64).reshape(-1, 4)
        b = np.ascontiguousarray(blinding_mont, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(104, dtype=np.uint8)
        check(lib().aleo_mi355x_kzg_commit_hiding(_p(out), powers.handle, _p(c), c.shape[0], gamma_powers.handle, _p(b), b.shape[0]),
              'kzg_commit_hiding')
        return out


class _Segment(ctypes.Structure):
    _fields_ = [('scalars', ctypes.c_void_p), ('len', ctypes.c_size_t), ('base_offset', ctypes.c_size_t), ('output', ctypes.c_uint32)]


class CommitterKey:
    sparse_range = False; lagrange_offset = 0; lagrange_size = 0      # > 0: the set also holds the Lagrange-basis powers of one domain (varuna.synthetic_committer_key)
    """Mirror of sonic_pc::CommitterKey [UPSTREAM-RECALL]: powers_of_beta_g, powers_of_beta_times_gamma_g, max_degree.  Both power
    arrays are pinned as ONE resident set (powers | gamma powers), so a hiding commitment is a single sum of segments."""

    def __init__(self, powers_of_beta_g: np.ndarray, powers_of_beta_times_gamma_g: np.ndarray = None, precompute: bool = True):
        pw = np.ascontiguousarray(powers_of_beta_g, dtype=np.uint8).reshape(-1, 104)
        self.max_degree = pw.shape[0] - 1
        self.gamma_offset = pw.shape[0]
        if powers_of_beta_times_gamma_g is not None:
            pw = np.concatenate([pw, np.ascontiguousarray(powers_of_beta_times_gamma_g, dtype=np.uint8).reshape(-1, 104)])
        self.n_gamma = pw.shape[0] - self.gamma_offset
        self.bases = PinnedBases(pw)
        if precompute: self.bases.precompute()

    def close(self): self.bases.close()
    def __enter__(self): return self
    def __exit__(self, *a): self.close()


class SonicKZG10:
    """Mirror of sonic_pc::SonicKZG10::commit for the labelled polynomials of one round: one call, shared launches."""

    @staticmethod
    def commit_segments_device(ck: 'CommitterKey', segments, k: int, stream: int = 0, sparse: bool = False) -> np.ndarray:
        """The general form: result q = sum over its segments of <device coefficient vector, bases[offset ..]>; segments: (ptr, len, offset, q).
        Used directly where a commitment mixes bases (commit_lagrange with a blinding term: evaluations against the Lagrange powers, the
        blinding scalar against v_H(tau) G, the hiding polynomial against the gamma powers)."""
        arr = (_Segment * max(len(segments), 1))(*[_Segment(int(p_), int(n_), int(o_), int(q_)) for p_, n_, o_, q_ in segments])
        out = np.zeros((k, 104), dtype=np.uint8)
        fn = lib().aleo_mi355x_kzg_commit_segments_sparse_device if sparse else lib().aleo_mi355x_kzg_commit_segments_device      # sparse: hint, see bases_precompute_range
        check(fn(_p(out), k, ck.bases.handle, arr, len(segments), ctypes.c_void_p(stream)), 'kzg_commit_segments_device')
        return out

    @staticmethod
    def commit_segments_sharded_device(sharded, segments, k: int, stream: int = 0) -> np.ndarray:
        """commit_segments_device against a SHARDED copy of the committer key (msm.ShardedBases): the vectors stay on the calling thread's device, every
        shard's device pulls its pieces and runs its own Pippenger; 144 bytes per result and shard come back (aleo_mi355x_kzg_commit_segments_sharded_device)."""
        arr = (_Segment * max(len(segments), 1))(*[_Segment(int(p_), int(n_), int(o_), int(q_)) for p_, n_, o_, q_ in segments])
        out = np.zeros((k, 104), dtype=np.uint8)
        check(lib().aleo_mi355x_kzg_commit_segments_sharded_device(_p(out), k, sharded.handle, arr, len(segments), ctypes.c_void_p(stream)), 'kzg_commit_segments_sharded_device')
        return out

    @staticmethod
    def commit(ck: CommitterKey, polynomials, device: bool = False, stream: int = 0) -> np.ndarray:
        """polynomials: iterable of (coeffs, degree_bound, blinding) — coeffs uint64[n,4] Montgomery (or a device pointer + length
        tuple when device=True), degree_bound None or an int <= max_degree (the commitment then uses the shifted powers
        powers[max_degree - bound ..]), blinding None or uint64[m,4] Montgomery (the random polynomial of a hiding commitment, committed
        against the gamma powers).  Returns uint8[k,104]."""
        segs, keep = [], []
        for q, (coeffs, bound, blind) in enumerate(polynomials):
            if device: ptr, n = int(coeffs[0]), int(coeffs[1])
            else:
                a = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4); n = a.shape[0]
                while n and not a[n - 1].any(): n -= 1
                keep.append(a); ptr = a.ctypes.data
            off = 0
            if bound is not None:
                if bound > ck.max_degree or n > bound + 1: raise ValueError('degree bound below the degree or above max_degree')
                off = ck.max_degree - bound
            segs.append((ptr, n, off, q))
            if blind is not None:
                if device: bptr, m = int(blind[0]), int(blind[1])
                else:
                    b = np.ascontiguousarray(blind, dtype=np.uint64).reshape(-1, 4); m = b.shape[0]; keep.append(b); bptr = b.ctypes.data
                if m > ck.n_gamma: raise ValueError('blinding polynomial longer than the gamma powers')
                segs.append((bptr, m, ck.gamma_offset, q))
        k = len(polynomials) if hasattr(polynomials, '__len__') else (segs[-1][3] + 1 if segs else 0)
        arr = (_Segment * max(len(segs), 1))(*[_Segment(p_, n_, o_, q_) for p_, n_, o_, q_ in segs])
        out = np.zeros((k, 104), dtype=np.uint8)
        if device:
            check(lib().aleo_mi355x_kzg_commit_segments_device(_p(out), k, ck.bases.handle, arr, len(segs), ctypes.c_void_p(stream)), 'kzg_commit_segments_device')
        else:
            check(lib().aleo_mi355x_kzg_commit_segments(_p(out), k, ck.bases.handle, arr, len(segs)), 'kzg_commit_segments')
        return out

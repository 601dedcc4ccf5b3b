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

"""Host-side mirror of snarkvm_algorithms::msm::VariableBase for the MI355X backend.

Reference interface (snarkVM 0.14.5, algorithms/src/msm/variable_base/mod.rs [UPSTREAM-RECALL], reached from
/root/reference/rust/src/program/execute.rs:74):
    VariableBase::msm(bases: &[G1Affine], scalars: &[BigInteger256]) -> G1Projective
Arrays: bases uint8[n,104] (snarkVM Affine layout) or uint8[n,96]; scalars uint64[n,4] canonical;
result uint64[18] = Jacobian (x, y, z), affine-normalised (z = 1, or (1,1,0) for the identity).
"""
from __future__ import annotations
import ctypes
import numpy as np
from ._lib import lib, check


def _p(a): return a.ctypes.data_as(ctypes.c_void_p)


class PinnedBases:
    """A base set resident in HBM (one SRS / proving key).  Mirrors the lifetime of an Arc<Vec<G1Affine>>."""

    def __init__(self, bases: np.ndarray = None, _handle: int = 0, _n: int = 0):
        if bases is None:
            self.handle, self.n = _handle, _n
            return
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        assert bases.ndim == 2 and bases.shape[1] in (96, 104)
        self.n = bases.shape[0]
        h = ctypes.c_uint64(0)
        check(lib().aleo_mi355x_bases_pin(_p(bases), bases.shape[1], self.n, ctypes.byref(h)), 'bases_pin')
        self.handle = h.value

    @classmethod
    def generate_multiples(cls, base_affine104: np.ndarray, first_multiple: int, n: int) -> 'PinnedBases':
        """P_i = (first_multiple + i) * base, generated in HBM (synthetic SRS-shaped base set)."""
        b = np.ascontiguousarray(base_affine104, dtype=np.uint8).reshape(104)
        h = ctypes.c_uint64(0)
        check(lib().aleo_mi355x_bases_generate(_p(b), first_multiple, n, ctypes.byref(h)), 'bases_generate')
        return cls(None, h.value, n)

    @classmethod
    def from_scalars(cls, base_affine104: np.ndarray, scalars: np.ndarray) -> 'PinnedBases':
        """P_i = s_i * base for canonical scalars uint64[n,4] (e.g. s_i = beta^i: a synthetic universal-setup SRS), built in HBM."""
        b = np.ascontiguousarray(base_affine104, dtype=np.uint8).reshape(104)
        sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        h = ctypes.c_uint64(0)
        check(lib().aleo_mi355x_bases_from_scalars(_p(b), _p(sc), sc.shape[0], ctypes.byref(h)), 'bases_from_scalars')
        return cls(None, h.value, sc.shape[0])

    def precompute(self) -> 'PinnedBases':
        """Build the fixed-base window table in HBM (13 x 112 B per point at 2^20); full-length MSMs then take the fast path."""
        check(lib().aleo_mi355x_bases_precompute(self.handle), 'bases_precompute')
        return self

    def precompute_range(self, offset: int, n: int, window_bits: int = 13) -> 'PinnedBases':
        """A narrow-window table over points [offset, offset + n) for sparse (mostly 0 / 1) scalar vectors — see aleo_mi355x_bases_precompute_range."""
        check(lib().aleo_mi355x_bases_precompute_range(self.handle, offset, n, window_bits), 'bases_precompute_range')
        return self

    def info(self) -> dict:
        """Points and HBM footprint of the set: rows, fixed-base tables, window width of every table tier."""
        buf = (ctypes.c_uint64 * 8)()
        k = lib().aleo_mi355x_bases_info(self.handle, buf, 8)
        if k < 7: raise ValueError('unknown bases handle')
        return {'points': int(buf[0]), 'row_bytes': int(buf[1]), 'table_bytes': int(buf[2]), 'tier_window_bits': [int(buf[3 + i]) for i in range(int(buf[6]))]}

    def attach_shards(self, sharded: 'ShardedBases' = None, min_points: int = 0, transforms_from: int = None) -> 'PinnedBases':
        """From now on the prover's commitments against this set with >= min_points scalars go through `sharded` — the same points cut over several
        devices (aleo_mi355x_bases_attach_shards).  None detaches.  The sharded set must outlive the attachment.
        transforms_from: the prover's transforms of at least that many elements run over the same devices (aleo_mi355x_bases_shard_transforms; default 2^24)."""
        check(lib().aleo_mi355x_bases_attach_shards(self.handle, sharded.handle if sharded is not None else 0, int(min_points)), 'bases_attach_shards')
        if sharded is not None and transforms_from is not None:
            check(lib().aleo_mi355x_bases_shard_transforms(self.handle, int(transforms_from)), 'bases_shard_transforms')
        return self

    def download(self, offset: int = 0, n: int = None) -> np.ndarray:
        n = self.n - offset if n is None else n
        out = np.zeros((n, 104), dtype=np.uint8)
        check(lib().aleo_mi355x_bases_download(self.handle, offset, n, _p(out)), 'bases_download')
        return out

    def close(self):
        if self.handle:
            lib().aleo_mi355x_bases_unpin(self.handle); self.handle = 0

    def __enter__(self): return self
    def __exit__(self, *a): self.close()
    def __del__(self):
        try: self.close()
        except Exception: pass


class ShardedBases:
    """One base set cut into contiguous shards over several devices of this process (aleo_mi355x_bases_pin_sharded):
    shard g = points [n g / G, n (g+1) / G) on devices[g].  devices: a list of HIP device indices (an index may repeat), or a count G
    (device g mod the visible devices)."""

    def __init__(self, bases: np.ndarray = None, devices=1, precompute: bool = False, _handle: int = 0, _n: int = 0):
        if bases is None:
            self.handle, self.n = _handle, _n
            return
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        assert bases.ndim == 2 and bases.shape[1] in (96, 104)
        self.n = bases.shape[0]
        dv, g = self._devices(devices); h = ctypes.c_uint64(0)
        check(lib().aleo_mi355x_bases_pin_sharded(_p(bases), bases.shape[1], self.n, dv, g, 1 if precompute else 0, ctypes.byref(h)), 'bases_pin_sharded')
        self.handle = h.value

    @staticmethod
    def _devices(devices):
        if isinstance(devices, int): return None, devices
        arr = (ctypes.c_int32 * len(devices))(*[int(d) for d in devices]); return arr, len(devices)

    @classmethod
    def generate_multiples(cls, base_affine104: np.ndarray, first_multiple: int, n: int, devices=1, precompute: bool = False) -> 'ShardedBases':
        b = np.ascontiguousarray(base_affine104, dtype=np.uint8).reshape(104)
        dv, g = cls._devices(devices); h = ctypes.c_uint64(0)
        check(lib().aleo_mi355x_bases_generate_sharded(_p(b), first_multiple, n, dv, g, 1 if precompute else 0, ctypes.byref(h)), 'bases_generate_sharded')
        return cls(None, _handle=h.value, _n=n)

    def shards(self):
        """[(device, first point, point count)] in shard order."""
        buf = (ctypes.c_uint64 * 200)(); k = lib().aleo_mi355x_bases_sharded_info(self.handle, buf, 200)
        if k < 1: raise ValueError('unknown sharded handle')
        return [(int(buf[1 + 3 * g]), int(buf[2 + 3 * g]), int(buf[3 + 3 * g])) for g in range(int(buf[0]))]

    def close(self):
        if self.handle:
            lib().aleo_mi355x_bases_unpin_sharded(self.handle); self.handle = 0

    def __enter__(self): return self
    def __exit__(self, *a): self.close()
    def __del__(self):
        try: self.close()
        except Exception: pass


class VariableBase:
    @staticmethod
    def msm_sharded(bases: ShardedBases, scalars: np.ndarray, partials: bool = False):
        """ONE MSM over the devices of a ShardedBases (aleo_mi355x_msm_g1_sharded): per-device Pippenger, the partial sums added on the host in
        shard order.  Returns uint64[18] (and the partials uint64[G,18] when asked)."""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        n = min(bases.n, scalars.shape[0]); out = np.zeros(18, dtype=np.uint64)
        part = np.zeros((len(bases.shards()), 18), dtype=np.uint64) if partials else None
        check(lib().aleo_mi355x_msm_g1_sharded(_p(out), bases.handle, _p(scalars), n, _p(part) if partials else None), 'msm_g1_sharded')
        return (out, part) if partials else out

    @staticmethod
    def msm(bases, scalars: np.ndarray) -> np.ndarray:
        """sum_i scalars[i] * bases[i]; zips to the shorter length like the reference."""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(18, dtype=np.uint64)
        if isinstance(bases, ShardedBases): return VariableBase.msm_sharded(bases, scalars)
        if isinstance(bases, PinnedBases):
            n = min(bases.n, scalars.shape[0])
            check(lib().aleo_mi355x_msm_g1_pinned(_p(out), bases.handle, _p(scalars), n), 'msm_g1_pinned')
            return out
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        n = min(bases.shape[0], scalars.shape[0])
        stride = bases.shape[1] if bases.ndim == 2 and bases.shape[0] else 104
        check(lib().aleo_mi355x_msm_g1(_p(out), _p(bases), stride, _p(scalars), n), 'msm_g1')
        return out

    @staticmethod
    def msm_device(bases: PinnedBases, d_scalars_ptr: int, n: int, stream: int = 0, sparse: bool = False) -> np.ndarray:
        """Scalars already in HBM (device pointer, e.g. torch tensor .data_ptr()).  sparse: the hint that they are witness-like (range table)."""
        out = np.zeros(18, dtype=np.uint64)
        fn = lib().aleo_mi355x_msm_g1_device_sparse if sparse else lib().aleo_mi355x_msm_g1_device
        check(fn(_p(out), bases.handle, ctypes.c_void_p(d_scalars_ptr), n, ctypes.c_void_p(stream)), 'msm_g1_device')
        return out


    @staticmethod
    def msm_batch_device(bases: PinnedBases, d_ptrs, lens, stream: int = 0) -> np.ndarray:
        """k MSMs against prefixes of one pinned set in one call (the commitments of one prover round): d_ptrs[q] is the device
        pointer of lens[q] canonical scalars.  Returns uint64[k,18]."""
        k = len(d_ptrs)
        ptrs = (ctypes.c_void_p * max(k, 1))(*[int(x) for x in d_ptrs]); ln = (ctypes.c_size_t * max(k, 1))(*[int(x) for x in lens])
        out = np.zeros((k, 18), dtype=np.uint64)
        check(lib().aleo_mi355x_msm_g1_batch_device(_p(out), bases.handle, ptrs, ln, k, ctypes.c_void_p(stream)), 'msm_g1_batch_device')
        return out


def msm_g2(bases: np.ndarray, scalars: np.ndarray) -> np.ndarray:
    """VariableBase::msm over BLS12-377 G2: bases uint8[n,200] (snarkVM G2Affine) or uint8[n,192], scalars uint64[n,4] canonical;
    result uint64[36] = Jacobian (x, y, z) over Fq2, affine-normalised."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8); scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    n = min(bases.shape[0], scalars.shape[0])
    stride = bases.shape[1] if bases.ndim == 2 and bases.shape[0] else 200
    out = np.zeros(36, dtype=np.uint64)
    check(lib().aleo_mi355x_msm_g2(_p(out), _p(bases), stride, _p(scalars), n), 'msm_g2')
    return out


class PinnedG2Bases:
    """A G2 base set resident in HBM (aleo_mi355x_bases_g2_pin): `msm(scalars)` multiplies the first len(scalars) bases, result as msm_g2's."""
    def __init__(self, bases: np.ndarray):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        self.n = int(bases.shape[0]); stride = bases.shape[1] if bases.ndim == 2 and bases.shape[0] else 200
        h = ctypes.c_uint64(0)
        check(lib().aleo_mi355x_bases_g2_pin(_p(bases), stride, self.n, ctypes.byref(h)), 'bases_g2_pin')
        self.handle = h.value

    def msm(self, scalars: np.ndarray) -> np.ndarray:
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(36, dtype=np.uint64)
        check(lib().aleo_mi355x_msm_g2_pinned(_p(out), self.handle, _p(scalars), scalars.shape[0]), 'msm_g2_pinned')
        return out

    def close(self):
        if self.handle: check(lib().aleo_mi355x_bases_g2_unpin(self.handle), 'bases_g2_unpin'); self.handle = 0

    def __enter__(self): return self
    def __exit__(self, *a): self.close()


def g2_sum(points: np.ndarray) -> np.ndarray:
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 36)
    out = np.zeros(36, dtype=np.uint64)
    check(lib().aleo_mi355x_g2_sum(_p(out), _p(pts), pts.shape[0]), 'g2_sum')
    return out


def g1_sum(points: np.ndarray) -> np.ndarray:
    """Group sum of Jacobian points uint64[k,18] (the local add after the all-gather of per-GPU partials)."""
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 18)
    out = np.zeros(18, dtype=np.uint64)
    check(lib().aleo_mi355x_g1_sum(_p(out), _p(pts), pts.shape[0]), 'g1_sum')
    return out


def last_msm_timing() -> dict:
    buf = (ctypes.c_double * 7)()
    k = lib().aleo_mi355x_last_msm_timing(buf, 7)
    names = ['total_ms', 'sort_ms', 'accum_ms', 'reduce_ms', 'host_ms', 'accum_kernel_ms', 'accum_launches']
    return {names[i]: buf[i] for i in range(k)}


def fq_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 6); b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 6)
    r = np.zeros_like(a); check(lib().aleo_mi355x_fq_mul(_p(r), _p(a), _p(b), a.shape[0]), 'fq_mul'); return r


def fr_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4); b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    r = np.zeros_like(a); check(lib().aleo_mi355x_fr_mul(_p(r), _p(a), _p(b), a.shape[0]), 'fr_mul'); return r

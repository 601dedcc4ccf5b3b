"""ctypes loader for liboracle.so (oracle/oracle.c).  TEST INFRASTRUCTURE ONLY.

Numpy conventions shared with aleo_amd: Fr arrays are uint64[n,4], Fq uint64[n,6], affine bases are
uint8[n,104] (snarkVM Affine layout: x, y Montgomery + infinity byte), Jacobian results uint64[18].
"""
from __future__ import annotations
import ctypes, os, subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'oracle.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, 'liboracle.so'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        vp, sz, ci, cu = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint
        L.oracle_msm_g1.argtypes = [vp, vp, sz, vp, sz, ci, ci]; L.oracle_msm_g1.restype = ci
        L.oracle_msm_g1_naive.argtypes = [vp, vp, sz, vp, sz]; L.oracle_msm_g1_naive.restype = ci
        L.oracle_g1_to_affine.argtypes = [vp, vp]; L.oracle_g1_to_affine.restype = None
        L.oracle_g1_on_curve.argtypes = [vp]; L.oracle_g1_on_curve.restype = ci
        L.oracle_g1_mul.argtypes = [vp, vp, vp]; L.oracle_g1_mul.restype = None
        L.oracle_g1_multiples.argtypes = [vp, vp, sz]; L.oracle_g1_multiples.restype = None
        for f in ('oracle_fr_to_mont', 'oracle_fr_from_mont', 'oracle_fq_to_mont', 'oracle_fq_from_mont'):
            getattr(L, f).argtypes = [vp, sz]; getattr(L, f).restype = None
        for f in ('oracle_fr_mul', 'oracle_fq_mul'):
            getattr(L, f).argtypes = [vp, vp, vp, sz]; getattr(L, f).restype = None
        L.oracle_ntt_fr.argtypes = [vp, cu, ci, ci, ci]; L.oracle_ntt_fr.restype = ci
        L.oracle_ntt_fr_mt.argtypes = [vp, cu, ci, ci, ci, ci]; L.oracle_ntt_fr_mt.restype = ci
        L.oracle_kzg_commit.argtypes = [vp, vp, vp, sz, ci]; L.oracle_kzg_commit.restype = ci
        L.oracle_fr_batch_inverse.argtypes = [vp, sz]; L.oracle_fr_batch_inverse.restype = None
        L.oracle_fr_vec_op.argtypes = [vp, vp, vp, sz, ci]; L.oracle_fr_vec_op.restype = None
        L.oracle_fr_spmv.argtypes = [vp, vp, vp, vp, vp, sz]; L.oracle_fr_spmv.restype = None
        L.oracle_msm_g2.argtypes = [vp, vp, sz, vp, sz, ci]; L.oracle_msm_g2.restype = ci
        L.oracle_g2_to_affine.argtypes = [vp, vp]; L.oracle_g2_to_affine.restype = None
        L.oracle_g2_multiples.argtypes = [vp, vp, sz]; L.oracle_g2_multiples.restype = None
        L.oracle_fr_divide_by_linear.argtypes = [vp, vp, vp, sz, vp]; L.oracle_fr_divide_by_linear.restype = None
        _LIB = L
    return _LIB


def _p(a): return a.ctypes.data_as(ctypes.c_void_p)


# ---- int <-> limb helpers --------------------------------------------------------------------
def ints_to_limbs(vals, nl) -> np.ndarray:
    out = np.zeros((len(vals), nl), dtype=np.uint64)
    m = (1 << 64) - 1
    for i, v in enumerate(vals):
        for j in range(nl):
            out[i, j] = (v >> (64 * j)) & m
    return out


def limbs_to_ints(arr) -> list:
    arr = np.asarray(arr, dtype=np.uint64)
    if arr.ndim == 1: arr = arr[None, :]
    return [sum(int(arr[i, j]) << (64 * j) for j in range(arr.shape[1])) for i in range(arr.shape[0])]


def fr_to_mont(a): a = np.ascontiguousarray(a, dtype=np.uint64).copy(); lib().oracle_fr_to_mont(_p(a), a.shape[0]); return a
def fr_from_mont(a): a = np.ascontiguousarray(a, dtype=np.uint64).copy(); lib().oracle_fr_from_mont(_p(a), a.shape[0]); return a
def fq_to_mont(a): a = np.ascontiguousarray(a, dtype=np.uint64).copy(); lib().oracle_fq_to_mont(_p(a), a.shape[0]); return a
def fq_from_mont(a): a = np.ascontiguousarray(a, dtype=np.uint64).copy(); lib().oracle_fq_from_mont(_p(a), a.shape[0]); return a


def affine_from_ints(points) -> np.ndarray:
    """[(x,y)|None] canonical ints -> uint8[n,104] snarkVM Affine (Montgomery)."""
    n = len(points)
    out = np.zeros((n, 104), dtype=np.uint8)
    xs = [p[0] if p else 0 for p in points]; ys = [p[1] if p else 1 for p in points]
    xm = fq_to_mont(ints_to_limbs(xs, 6)); ym = fq_to_mont(ints_to_limbs(ys, 6))
    out[:, 0:48] = xm.view(np.uint8).reshape(n, 48); out[:, 48:96] = ym.view(np.uint8).reshape(n, 48)
    for i, p in enumerate(points):
        if p is None: out[i, 96] = 1
    return out


def affine_to_ints(aff) -> list:
    aff = np.ascontiguousarray(aff, dtype=np.uint8).reshape(-1, 104)
    n = aff.shape[0]
    x = fq_from_mont(np.ascontiguousarray(aff[:, 0:48]).view(np.uint64).reshape(n, 6))
    y = fq_from_mont(np.ascontiguousarray(aff[:, 48:96]).view(np.uint64).reshape(n, 6))
    xi, yi = limbs_to_ints(x), limbs_to_ints(y)
    return [None if aff[i, 96] else (xi[i], yi[i]) for i in range(n)]


def jac_to_affine(jac) -> np.ndarray:
    jac = np.ascontiguousarray(jac, dtype=np.uint64).reshape(18)
    out = np.zeros(104, dtype=np.uint8); lib().oracle_g1_to_affine(_p(out), _p(jac)); return out


def jac_to_int_point(jac):
    return affine_to_ints(jac_to_affine(jac))[0]


def msm_g1(bases, scalars, threads=1, variant=0) -> np.ndarray:
    """bases uint8[n,104|96], scalars uint64[n,4] canonical -> Jacobian uint64[18]."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8); scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = scalars.shape[0]; stride = bases.shape[1] if n else 104
    out = np.zeros(18, dtype=np.uint64)
    rc = lib().oracle_msm_g1(_p(out), _p(bases), stride, _p(scalars), n, threads, variant)
    assert rc == 0
    return out


def msm_g1_naive(bases, scalars) -> np.ndarray:
    bases = np.ascontiguousarray(bases, dtype=np.uint8); scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.zeros(18, dtype=np.uint64)
    lib().oracle_msm_g1_naive(_p(out), _p(bases), bases.shape[1] if len(bases) else 104, _p(scalars), scalars.shape[0])
    return out


def g1_multiples(base104, n) -> np.ndarray:
    out = np.zeros((n, 104), dtype=np.uint8)
    b = np.ascontiguousarray(base104, dtype=np.uint8)
    lib().oracle_g1_multiples(_p(out), _p(b), n); return out


def g1_mul(base104, scalar4) -> np.ndarray:
    out = np.zeros(104, dtype=np.uint8)
    b = np.ascontiguousarray(base104, dtype=np.uint8); s = np.ascontiguousarray(scalar4, dtype=np.uint64)
    lib().oracle_g1_mul(_p(out), _p(b), _p(s)); return out


def ntt_fr(data, order=0, direction=0, type=0, threads=1) -> np.ndarray:
    """data uint64[n,4] Montgomery -> transformed copy.  threads > 1: the rayon-style parallel transform (oracle_ntt_fr_mt), same values."""
    a = np.ascontiguousarray(data, dtype=np.uint64).copy()
    n = a.shape[0]; lg = n.bit_length() - 1; assert 1 << lg == n
    rc = lib().oracle_ntt_fr(_p(a), lg, order, direction, type) if threads <= 1 else lib().oracle_ntt_fr_mt(_p(a), lg, order, direction, type, threads)
    assert rc == 0
    return a


def _chunks(n, parts):
    return [(n * i // parts, n * (i + 1) // parts) for i in range(parts) if n * (i + 1) // parts > n * i // parts]


def parallel_rows(fn, n, threads):
    """fn(lo, hi) over `threads` contiguous row ranges on a thread pool (ctypes calls release the GIL): how the CPU baseline spreads the
    element-wise field work of a proof over the host cores, as snarkVM's rayon iterators do."""
    if threads <= 1 or n < 4096: fn(0, n); return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(threads) as ex: list(ex.map(lambda r: fn(*r), _chunks(n, threads)))


def fr_vec_op_mt(a, b, op: int, threads: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64); r = np.zeros_like(a)
    parallel_rows(lambda lo, hi: lib().oracle_fr_vec_op(_p(r[lo:hi]), _p(a[lo:hi]), _p(b[lo:hi]), hi - lo, op), a.shape[0], threads); return r


def fr_batch_inverse_mt(a, threads: int) -> np.ndarray:
    """snarkvm_fields::batch_inversion splits the slice into chunks (rayon) and inverts each with its own shared inversion: the same here."""
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    parallel_rows(lambda lo, hi: lib().oracle_fr_batch_inverse(_p(a[lo:hi]), hi - lo), a.shape[0], threads); return a


def fr_spmv_mt(row_ptr, col_idx, vals, x, threads: int) -> np.ndarray:
    rp = np.ascontiguousarray(row_ptr, dtype=np.uint32); ci = np.ascontiguousarray(col_idx, dtype=np.uint32)
    v = np.ascontiguousarray(vals, dtype=np.uint64); xx = np.ascontiguousarray(x, dtype=np.uint64)
    n = rp.shape[0] - 1; y = np.zeros((n, 4), dtype=np.uint64)
    def rows(lo, hi):
        sub = (rp[lo:hi + 1] - rp[lo]).astype(np.uint32); k0, k1 = int(rp[lo]), int(rp[hi])
        lib().oracle_fr_spmv(_p(y[lo:hi]), _p(sub), _p(np.ascontiguousarray(ci[k0:k1])), _p(np.ascontiguousarray(v[k0:k1])), _p(xx), hi - lo)
    parallel_rows(rows, n, threads); return y


def kzg_commit(bases104, coeffs_mont, threads=1) -> np.ndarray:
    out = np.zeros(104, dtype=np.uint8)
    b = np.ascontiguousarray(bases104, dtype=np.uint8); c = np.ascontiguousarray(coeffs_mont, dtype=np.uint64)
    rc = lib().oracle_kzg_commit(_p(out), _p(b), _p(c), c.shape[0], threads); assert rc == 0
    return out


def fr_batch_inverse(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).copy(); lib().oracle_fr_batch_inverse(_p(a), a.shape[0]); return a


def fr_vec_op(a, b, op: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
    r = np.zeros_like(a); lib().oracle_fr_vec_op(_p(r), _p(a), _p(b), a.shape[0], op); return r


def fr_spmv(row_ptr, col_idx, vals, x) -> np.ndarray:
    """CSR matrix (uint32 row_ptr / col_idx, uint64[nnz,4] Montgomery values) times a Montgomery Fr vector."""
    rp = np.ascontiguousarray(row_ptr, dtype=np.uint32); ci = np.ascontiguousarray(col_idx, dtype=np.uint32)
    v = np.ascontiguousarray(vals, dtype=np.uint64); xx = np.ascontiguousarray(x, dtype=np.uint64)
    y = np.zeros((rp.shape[0] - 1, 4), dtype=np.uint64)
    lib().oracle_fr_spmv(_p(y), _p(rp), _p(ci), _p(v), _p(xx), y.shape[0]); return y


def fr_divide_by_linear(poly_mont, z_mont):
    """(p(X) - p(z)) / (X - z): returns (quotient uint64[n-1,4], p(z) uint64[4]), Montgomery."""
    a = np.ascontiguousarray(poly_mont, dtype=np.uint64).reshape(-1, 4); z = np.ascontiguousarray(z_mont, dtype=np.uint64).reshape(4)
    q = np.zeros((max(a.shape[0] - 1, 0), 4), dtype=np.uint64); ev = np.zeros(4, dtype=np.uint64)
    lib().oracle_fr_divide_by_linear(_p(q) if q.size else None, _p(ev), _p(a) if a.size else None, a.shape[0], _p(z)); return q, ev


# ---- G2: uint8[n,200] snarkVM G2Affine rows (x.c0, x.c1, y.c0, y.c1 Montgomery + infinity byte at 192); Jacobian uint64[36]
def g2_affine_from_ints(points) -> np.ndarray:
    """[((x0,x1),(y0,y1))|None] canonical ints -> uint8[n,200]."""
    n = len(points); out = np.zeros((n, 200), dtype=np.uint8)
    flat = []
    for P in points: flat += [0, 0, 1, 0] if P is None else [P[0][0], P[0][1], P[1][0], P[1][1]]
    m = fq_to_mont(ints_to_limbs(flat, 6)).view(np.uint8).reshape(n, 192)
    out[:, :192] = m
    for i, P in enumerate(points):
        if P is None: out[i, 192] = 1
    return out


def g2_affine_to_ints(aff) -> list:
    aff = np.ascontiguousarray(aff, dtype=np.uint8).reshape(-1, 200); n = aff.shape[0]
    v = limbs_to_ints(fq_from_mont(np.ascontiguousarray(aff[:, :192]).view(np.uint64).reshape(4 * n, 6)))
    return [None if aff[i, 192] else ((v[4 * i], v[4 * i + 1]), (v[4 * i + 2], v[4 * i + 3])) for i in range(n)]


def g2_jac_to_int_point(jac):
    jac = np.ascontiguousarray(jac, dtype=np.uint64).reshape(36)
    out = np.zeros(200, dtype=np.uint8); lib().oracle_g2_to_affine(_p(out), _p(jac))
    return g2_affine_to_ints(out)[0]


def msm_g2(bases, scalars, threads=1) -> np.ndarray:
    bases = np.ascontiguousarray(bases, dtype=np.uint8); scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = scalars.shape[0]; stride = bases.shape[1] if n else 200
    out = np.zeros(36, dtype=np.uint64)
    rc = lib().oracle_msm_g2(_p(out), _p(bases), stride, _p(scalars), n, threads); assert rc == 0
    return out


def g2_multiples(base200, n) -> np.ndarray:
    out = np.zeros((n, 200), dtype=np.uint8); b = np.ascontiguousarray(base200, dtype=np.uint8)
    lib().oracle_g2_multiples(_p(out), _p(b), n); return out

"""Host-side mirror of the field-only vector work between NTTs (snarkVM 0.14.5 Evaluations::{mul,add,sub}_assign and
snarkvm_fields::batch_inversion [UPSTREAM-RECALL]) on device-resident Montgomery Fr vectors (torch tensors or raw pointers)."""
from __future__ import annotations
import ctypes
from ._lib import lib, check

OP_MUL, OP_ADD, OP_SUB = 0, 1, 2


def fr_vec_op_device(d_dst: int, d_a: int, d_b: int, n: int, op: int, stream: int = 0):
    check(lib().aleo_mi355x_fr_vec_op_device(ctypes.c_void_p(d_dst), ctypes.c_void_p(d_a), ctypes.c_void_p(d_b), n, op,
                                             ctypes.c_void_p(stream)), 'fr_vec_op_device')


def fr_lin_device(d_dst: int, n: int, c0, c1=None, d_a: int = 0, c2=None, d_b: int = 0, stream: int = 0):
    """dst[i] = c0 + c1 * a[i] + c2 * b[i]; c0/c1/c2: uint64[4] Montgomery on the host (None = absent term)."""
    import numpy as np
    keep = [None if c is None else np.ascontiguousarray(c, dtype=np.uint64).reshape(4) for c in (c0, c1, c2)]
    ptr = [ctypes.c_void_p(0) if k is None else k.ctypes.data_as(ctypes.c_void_p) for k in keep]
    check(lib().aleo_mi355x_fr_lin_device(ctypes.c_void_p(d_dst), n, ptr[0], ptr[1], ctypes.c_void_p(d_a), ptr[2], ctypes.c_void_p(d_b),
                                          ctypes.c_void_p(stream)), 'fr_lin_device')


def fr_powers_device(d_dst: int, n: int, first, ratio, stream: int = 0):
    """dst[k] = first * ratio^k (first, ratio: uint64[4] Montgomery on the host)."""
    import numpy as np
    f = np.ascontiguousarray(first, dtype=np.uint64).reshape(4); r = np.ascontiguousarray(ratio, dtype=np.uint64).reshape(4)
    check(lib().aleo_mi355x_fr_powers_device(ctypes.c_void_p(d_dst), n, f.ctypes.data_as(ctypes.c_void_p), r.ctypes.data_as(ctypes.c_void_p),
                                             ctypes.c_void_p(stream)), 'fr_powers_device')


def fr_gather_mul_device(d_dst: int, n: int, d_scale: int, d_table1: int, d_idx1: int, d_table2: int = 0, d_idx2: int = 0, stream: int = 0):
    """dst[i] = scale[i] * table1[idx1[i]] * table2[idx2[i]] (uint32 indices; scale / table2 may be 0)."""
    vp = ctypes.c_void_p
    check(lib().aleo_mi355x_fr_gather_mul_device(vp(d_dst), n, vp(d_scale), vp(d_table1), vp(d_idx1), vp(d_table2), vp(d_idx2), vp(stream)), 'fr_gather_mul_device')


def fr_eval_batch_device(d_out: int, d_polys, lens, points_mont, stream: int = 0):
    """out[q] = p_q(z_q) for up to 8 polynomials; points_mont: uint64[k,4] Montgomery on the host."""
    import numpy as np
    k = len(d_polys)
    z = np.ascontiguousarray(points_mont, dtype=np.uint64).reshape(k, 4)
    ptrs = (ctypes.c_void_p * max(k, 1))(*[int(x) for x in d_polys]); ln = (ctypes.c_size_t * max(k, 1))(*[int(x) for x in lens])
    check(lib().aleo_mi355x_fr_eval_batch_device(ctypes.c_void_p(d_out), ptrs, ln, z.ctypes.data_as(ctypes.c_void_p), k, ctypes.c_void_p(stream)), 'fr_eval_batch_device')


def batch_inversion_device(d_inout: int, n: int, stream: int = 0):
    check(lib().aleo_mi355x_fr_batch_inverse_device(ctypes.c_void_p(d_inout), n, ctypes.c_void_p(stream)), 'fr_batch_inverse_device')


def spmv_device(d_y: int, d_row_ptr: int, d_col_idx: int, d_vals: int, d_x: int, rows: int, stream: int = 0):
    """y = M x for a CSR matrix over Fr (uint32 row_ptr / col_idx, Montgomery values): z_a = A z, z_b = B z."""
    vp = ctypes.c_void_p
    check(lib().aleo_mi355x_fr_spmv_device(vp(d_y), vp(d_row_ptr), vp(d_col_idx), vp(d_vals), vp(d_x), rows, vp(stream)), 'fr_spmv_device')


def divide_by_linear_device(d_quotient: int, d_eval: int, d_poly: int, n: int, z_mont, stream: int = 0):
    """quotient = (p(X) - p(z)) / (X - z) (n - 1 coefficients at d_quotient), p(z) at d_eval (32 bytes, may be 0): KZG10's witness
    polynomial.  z_mont: uint64[4] Montgomery, host."""
    import numpy as np
    z = np.ascontiguousarray(z_mont, dtype=np.uint64).reshape(4)
    check(lib().aleo_mi355x_fr_divide_by_linear_device(ctypes.c_void_p(d_quotient), ctypes.c_void_p(d_eval), ctypes.c_void_p(d_poly), n,
                                                       z.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(stream)), 'fr_divide_by_linear_device')

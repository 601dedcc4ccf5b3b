"""Host-side mirror of the field-only vector work between NTTs (snarkVM 0.14.5 Evaluations::{mul,add,sub}_assign and
snarkvm_fields::batch_inversion [UPSTREAM-RECALL]) on device-resident Montgomery Fr vectors (torch tensors or raw pointers)."""
from __future__ import annotations
import ctypes
_FR_MODULUS = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001
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
    """out[q] = p_q(z_q) for any number of polynomials (12 per call of the entry point); points_mont: uint64[k,4] Montgomery on the host."""
    import numpy as np
    k = len(d_polys)
    z = np.ascontiguousarray(points_mont, dtype=np.uint64).reshape(k, 4)
    for at in range(0, max(k, 1), 12):                      # the entry point takes 12 polynomials per call
        m = min(12, k - at)
        ptrs = (ctypes.c_void_p * max(m, 1))(*[int(x) for x in d_polys[at:at + m]]); ln = (ctypes.c_size_t * max(m, 1))(*[int(x) for x in lens[at:at + m]])
        zz = np.ascontiguousarray(z[at:at + m]) if m else np.zeros((1, 4), dtype=np.uint64)
        check(lib().aleo_mi355x_fr_eval_batch_device(ctypes.c_void_p(d_out + 32 * at), ptrs, ln, zz.ctypes.data_as(ctypes.c_void_p), m, ctypes.c_void_p(stream)), 'fr_eval_batch_device')


def fr_random_device(d_dst: int, n: int, seed, first_index: int = 0, montgomery: bool = True, stream: int = 0):
    """Elements first_index.. of the proof's random stream (ChaCha20 under the 32-byte `seed`, rejection-sampled below r), written in HBM."""
    from ._lib import seed32
    check(lib().aleo_mi355x_fr_random_device(ctypes.c_void_p(d_dst), n, seed32(seed), first_index, 1 if montgomery else 0, ctypes.c_void_p(stream)), 'fr_random_device')


def random_fr(seed, index: int, n: int = 1):
    """Elements index.. of the same stream on the host (python ints; aleo_mi355x_fr_random): the blinding scalars of a proof."""
    import numpy as np
    from ._lib import seed32
    out = np.zeros((n, 4), dtype=np.uint64)
    check(lib().aleo_mi355x_fr_random(out.ctypes.data_as(ctypes.c_void_p), n, seed32(seed), index), 'fr_random')
    vals = [sum(int(out[i, l]) << (64 * l) for l in range(4)) for i in range(n)]
    return vals[0] if n == 1 else vals


def fr_lincomb_device(d_dst: int, n: int, c0, terms, stream: int = 0):
    """dst[i] = c0 [i == 0] + sum_j coeff_j term_j[i]; terms: list of (device pointer, length, coeff uint64[4] Montgomery)."""
    import numpy as np
    terms = list(terms)
    if len(terms) > 28:                                     # the entry point takes 28 terms per call: later calls carry dst along as a term (as the native host side does)
        one = np.array([((1 << 256) % _FR_MODULUS >> (64 * i)) & (2 ** 64 - 1) for i in range(4)], dtype=np.uint64)
        fr_lincomb_device(d_dst, n, c0, terms[:28], stream)
        for at in range(28, len(terms), 27): fr_lincomb_device(d_dst, n, None, [(d_dst, n, one)] + terms[at:at + 27], stream)
        return
    k = len(terms)
    ptrs = (ctypes.c_void_p * max(k, 1))(*[int(t[0]) for t in terms]); ln = (ctypes.c_size_t * max(k, 1))(*[int(t[1]) for t in terms])
    co = np.ascontiguousarray(np.stack([np.asarray(t[2], dtype=np.uint64).reshape(4) for t in terms]) if k else np.zeros((1, 4), dtype=np.uint64))
    k0 = None if c0 is None else np.ascontiguousarray(c0, dtype=np.uint64).reshape(4)
    check(lib().aleo_mi355x_fr_lincomb_device(ctypes.c_void_p(d_dst), n, ctypes.c_void_p(0) if k0 is None else k0.ctypes.data_as(ctypes.c_void_p), ptrs, ln,
                                              co.ctypes.data_as(ctypes.c_void_p), k, ctypes.c_void_p(stream)), 'fr_lincomb_device')


def ahp_first_sumcheck_device(d_dst: int, n: int, d_r: int, d_za: int, d_zb: int, d_t: int, d_z: int, eta_b, eta_c, stream: int = 0):
    import numpy as np
    vp = ctypes.c_void_p
    eb = np.ascontiguousarray(eta_b, dtype=np.uint64).reshape(4); ec = np.ascontiguousarray(eta_c, dtype=np.uint64).reshape(4)
    check(lib().aleo_mi355x_ahp_first_sumcheck_device(vp(d_dst), n, vp(d_r), vp(d_za), vp(d_zb), vp(d_t), vp(d_z), eb.ctypes.data_as(vp), ec.ctypes.data_as(vp),
                                                      vp(stream)), 'ahp_first_sumcheck_device')


def ahp_matrix_sumcheck_device(d_dst: int, n: int, d_index, index_stride: int, d_f, consts_mont, stream: int = 0):
    """consts_mont: uint64[7,4] — delta_a, delta_b, delta_c, alpha beta, −alpha, −beta, v_H(alpha) v_H(beta)."""
    import numpy as np
    vp = ctypes.c_void_p
    ix = (vp * 3)(*[int(x) for x in d_index]); ff = (vp * 3)(*[int(x) for x in d_f])
    k = np.ascontiguousarray(consts_mont, dtype=np.uint64).reshape(7, 4)
    check(lib().aleo_mi355x_ahp_matrix_sumcheck_device(vp(d_dst), n, ix, index_stride, ff, k.ctypes.data_as(vp), vp(stream)), 'ahp_matrix_sumcheck_device')


def fr_blind_rows_device(d_dst: int, d_src: int, n: int, rho_mont, stream: int = 0):
    """dst row q (n + 1 coefficients) = src row q (n coefficients) + rho_q (X^n − 1); rho_mont: uint64[rows,4] Montgomery on the host."""
    import numpy as np
    r = np.ascontiguousarray(rho_mont, dtype=np.uint64).reshape(-1, 4); vp = ctypes.c_void_p
    for at in range(0, r.shape[0], 24):                     # 24 rows per call of the entry point
        m = min(24, r.shape[0] - at); rr = np.ascontiguousarray(r[at:at + m])
        check(lib().aleo_mi355x_fr_blind_rows_device(vp(d_dst + at * (n + 1) * 32), vp(d_src + at * n * 32), n, m, rr.ctypes.data_as(vp), vp(stream)), 'fr_blind_rows_device')


def ahp_sumcheck_operands_device(d_dst: int, d_witness_polys: int, d_x_polys: int, n: int, n_x: int, instances: int, stream: int = 0):
    vp = ctypes.c_void_p
    check(lib().aleo_mi355x_ahp_sumcheck_operands_device(vp(d_dst), vp(d_witness_polys), vp(d_x_polys), n, n_x, instances, vp(stream)), 'ahp_sumcheck_operands_device')


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

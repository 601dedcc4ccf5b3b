"""AHP prover for R1CS above the MI355X operators — the shape of snarkVM 0.14.5 `Varuna::prove_batch` /
`AHPForR1CS::prover_{first,second,third,fourth}_round` [UPSTREAM-RECALL: algorithms/src/snark/varuna/{varuna.rs, ahp/prover/round_functions/*.rs}],
reached from /root/reference/rust/src/program/execute.rs:74 (`trace.prove_execution`) and transfer.rs:99.

Everything that scales with the circuit runs on the device through the C ABI (sparse matrix-vector products, NTTs, pointwise field
kernels, batch inversion, division by X − z, batched SonicKZG10 commitments and the two opening MSMs); the host keeps the transcript, the
handful of challenge-dependent constants and the O(|X|) public-input polynomial.  The protocol is Marlin's AHP for R1CS in snarkVM's
arrangement (see oracle/varuna_ref.py for the statement of every identity; that CPU restatement is what tests compare this module with,
byte for byte).  The transcript is upstream's construction — the Poseidon sponge over Fq behind the algebraic-sponge interface, run by the
library (aleo_mi355x_fs_*; absorb order [UPSTREAM-RECALL], not checkable offline) — the blinding comes from a ChaCha20 stream under the proof's
32-byte seed, the SRS is synthetic.  The proof's byte layout is upstream's (aleo_mi355x_proof_to_bytes)."""
from __future__ import annotations
import ctypes
import numpy as np
import torch
from . import synth, wire
from ._lib import lib, check, seed32, UnsatisfiedAssignment
from .fft import EvaluationDomain, FORWARD, INVERSE
from .kzg import CommitterKey, SonicKZG10, KZG10
from .msm import PinnedBases
from .poly import (fr_vec_op_device, fr_lin_device, fr_powers_device, fr_gather_mul_device, fr_eval_batch_device, fr_random_device, fr_lincomb_device,
                   ahp_first_sumcheck_device, ahp_matrix_sumcheck_device, ahp_sumcheck_operands_device, fr_blind_rows_device, batch_inversion_device, random_fr, spmv_device, divide_by_linear_device, OP_MUL, OP_ADD, OP_SUB)

R = synth.FR_MODULUS
_RM = (1 << 256) % R
_R2 = _RM * _RM % R
PROTOCOL_NAME = b'VARUNA-2023'
HIDING_COEFFS = 3
TWO_ADIC_ROOT = 8065159656716812877374967518403273466521432693661810619979959746626482506078


def _inv(a): return pow(a % R, -1, R)
def _mont(v): return synth.int_to_limbs(v % R * _RM % R, 4)                    # host constant -> Montgomery limbs
def _mont_rows(vals): return np.stack([_mont(v) for v in vals]) if len(vals) else np.zeros((0, 4), dtype=np.uint64)
def _from_mont(limbs): return synth.limbs_to_int(limbs) * _inv(_RM) % R
def _fr_bytes(v): return int(v % R).to_bytes(32, 'little')
def _gen(size): return pow(TWO_ADIC_ROOT, 1 << (47 - (size.bit_length() - 1)), R)
def _vanish(size, x): return (pow(x, size, R) - 1) % R


class _Vec:
    """n Montgomery Fr values in HBM (a torch tensor is the allocation; every operation goes through the C ABI)."""
    def __init__(self, n, data: np.ndarray = None, zero: bool = False):
        self.n = n
        if data is None: self.t = (torch.zeros if zero else torch.empty)((max(n, 1), 4), dtype=torch.int64, device='cuda')
        else: self.t = torch.from_numpy(np.ascontiguousarray(data, dtype=np.uint64).reshape(-1, 4).view(np.int64)).cuda()
    def ptr(self, off=0): return self.t.data_ptr() + 32 * off
    def host(self, off=0, n=None): return self.t[off:off + (self.n - off if n is None else n)].cpu().numpy().view(np.uint64)


def _host_ntt(a, w):
    """(sum_t a_t w^(t u))_u for a primitive len(a)-th root w (radix 2, python integers)."""
    n = len(a)
    if n == 1: return list(a)
    w2 = w * w % R
    ev, od = _host_ntt(a[0::2], w2), _host_ntt(a[1::2], w2)
    out = [0] * n; t = 1
    for i in range(n // 2):
        v = t * od[i] % R; out[i] = (ev[i] + v) % R; out[i + n // 2] = (ev[i] - v) % R; t = t * w % R
    return out


def h_positions(n_vars, n_public, n_x, n_h) -> np.ndarray:
    """Index on H of every variable: public i -> i |H|/|X|, the j-th private one -> the j-th element of H \\ X."""
    ratio = n_h // n_x
    v = np.arange(n_vars, dtype=np.int64); j = v - n_public
    return np.where(v < n_public, v * ratio, j + j // max(ratio - 1, 1) + 1).astype(np.int64)


class FiatShamir:
    """The prover's transcript: `PoseidonSponge<Fq, 2, 1>` behind upstream's AlgebraicSponge interface, run by the library (poseidon.hpp through
    aleo_mi355x_fs_*) — the same sponge aleo_mi355x_varuna_prove* runs internally."""
    def __init__(self):
        h = ctypes.c_uint64(0); check(lib().aleo_mi355x_fs_new(ctypes.byref(h)), 'fs_new'); self.h = h.value
    def __del__(self):
        try:
            if getattr(self, 'h', 0): lib().aleo_mi355x_fs_free(self.h); self.h = 0
        except Exception: pass                                                      # interpreter shutdown
    def absorb_bytes(self, data: bytes):
        buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b'\0')
        check(lib().aleo_mi355x_fs_absorb_bytes(self.h, buf, len(data)), 'fs_absorb_bytes')
    def absorb_g1(self, affine104: np.ndarray):
        a = np.ascontiguousarray(affine104, dtype=np.uint8).reshape(-1, 104)
        check(lib().aleo_mi355x_fs_absorb_g1(self.h, a.ctypes.data_as(ctypes.c_void_p), 104, a.shape[0]), 'fs_absorb_g1')
    def absorb_fr(self, values):
        a = np.stack([synth.int_to_limbs(int(v) % R, 4) for v in values]) if len(values) else np.zeros((0, 4), dtype=np.uint64)
        check(lib().aleo_mi355x_fs_absorb_fr(self.h, a.ctypes.data_as(ctypes.c_void_p), a.shape[0]), 'fs_absorb_fr')
    def squeeze(self, n: int, short: bool = False):
        out = np.zeros((max(n, 1), 4), dtype=np.uint64)
        check(lib().aleo_mi355x_fs_squeeze_fr(self.h, out.ctypes.data_as(ctypes.c_void_p), n, 1 if short else 0), 'fs_squeeze_fr')
        return [synth.limbs_to_int(out[i]) for i in range(n)]
    def squeeze_short(self) -> int: return self.squeeze(1, True)[0]


def synthetic_committer_key(tau: int, s_gamma: int, max_degree: int, n_gamma: int = HIDING_COEFFS, lagrange_size: int = 0, range_window: int = 0) -> CommitterKey:
    """powers τ^i·G (i <= max_degree), then the hiding powers s·τ^i·G, then — for lagrange_size = |H| > 0 — the Lagrange-basis powers L_i(τ)·G of
    the domain H followed by v_H(τ)·G [UPSTREAM-RECALL: CommitterKey::lagrange_bases_at_beta_g]; built in HBM (SRS-shaped bases)."""
    n_l = lagrange_size + 1 if lagrange_size else 0
    vals, a = [], 1
    for i in range(max_degree + 1):
        vals.append(a); a = a * tau % R
    a = s_gamma % R
    for i in range(n_gamma):
        vals.append(a); a = a * tau % R
    if lagrange_size:
        n = lagrange_size; w = _gen(n); vh = _vanish(n, tau); n_inv_vh = vh * _inv(n) % R
        den, x = [], 1
        for i in range(n): den.append((tau - x) % R); x = x * w % R
        pre, acc = [], 1                                   # batch inversion of tau − w^i
        for d in den: pre.append(acc); acc = acc * d % R
        inv_acc = _inv(acc); x_pows = 1; lag = [0] * n
        xs = [1] * n
        for i in range(1, n): xs[i] = xs[i - 1] * w % R
        for i in range(n - 1, -1, -1):
            lag[i] = xs[i] * n_inv_vh % R * (inv_acc * pre[i] % R) % R; inv_acc = inv_acc * den[i] % R
        vals += lag + [vh]
    sc = np.zeros((len(vals), 4), dtype=np.uint64)
    for limb in range(4): sc[:, limb] = np.array([(v >> (64 * limb)) & 0xFFFFFFFFFFFFFFFF for v in vals], dtype=np.uint64)
    ck = CommitterKey.__new__(CommitterKey)
    ck.max_degree, ck.gamma_offset, ck.n_gamma = max_degree, max_degree + 1, n_gamma
    ck.lagrange_offset = max_degree + 1 + n_gamma if lagrange_size else 0; ck.lagrange_size = lagrange_size
    ck.bases = PinnedBases.from_scalars(synth.generator_affine104(), sc).precompute()
    ck.sparse_range = False
    if lagrange_size and range_window:      # narrow-window table over [hiding powers | Lagrange powers | v_H G]: the first round's witness commitments as one sparse chain
        ck.bases.precompute_range(ck.gamma_offset, n_gamma + lagrange_size + 1, range_window); ck.sparse_range = True
    return ck


DOMAIN_FLAGS = {'auto': 0, 'per_matrix': 1, 'shared': 2}


def domain_policy(n_k_m, domains='auto'):
    """Non-zero domains of A, B, C: 'per_matrix' keeps each matrix's own power of two, 'shared' gives all three the largest, 'auto' shares
    below 2^18 (latency-bound rounds: one batched transform beats three short ones) and separates from there on (fewer points to commit)."""
    if domains not in DOMAIN_FLAGS: raise ValueError('domains: auto, per_matrix or shared')
    big = max(n_k_m)
    return [big] * 3 if domains == 'shared' or (domains == 'auto' and big < (1 << 18)) else list(n_k_m)


def _runs(n_k_m):
    """Maximal runs of consecutive matrices with equal domains: [(first, count)] — a run shares batched transforms and one numerator pass."""
    out, m = [], 0
    while m < 3:
        c = 1
        while m + c < 3 and n_k_m[m + c] == n_k_m[m]: c += 1
        out.append((m, c)); m += c
    return out


def _csr_on_h(csr, pos, n_h):
    """The matrix with its columns moved to positions on H and its rows padded to |H| (row_ptr, col, val canonical)."""
    ptr, col, val = csr
    rp = np.full(n_h + 1, ptr[-1], dtype=np.uint32); rp[:len(ptr)] = ptr
    return rp, pos[col].astype(np.uint32), val


class CircuitIndex:
    """Index of one circuit (the prover-key material): matrices in HBM, their arithmetisation over K (evaluations on K and on 2K,
    coefficient forms) and the twelve index commitments [UPSTREAM-RECALL: varuna/ahp/indexer — AHPForR1CS::index]."""

    def __init__(self, csr, n_constraints: int, n_public: int, n_private: int, ck: CommitterKey, stream: torch.cuda.Stream = None, domains: str = 'auto'):
        self.ck = ck
        self.n_constraints, self.n_public, self.n_private = n_constraints, n_public, n_private
        n_x = 1
        while n_x < n_public: n_x *= 2
        n_h = 1
        while n_h < max(n_constraints, n_x + n_private, 2 * n_x): n_h *= 2
        self.n_k_m = []                                     # one non-zero domain per matrix [UPSTREAM-RECALL: non_zero_{a,b,c}_domain]
        for m in 'abc':
            nk = 2
            while nk < int(csr[m][0][-1]): nk *= 2
            self.n_k_m.append(nk)
        n_k = max(self.n_k_m)
        self.n_k_m = domain_policy(self.n_k_m, domains)
        self.k_off = [0, self.n_k_m[0], self.n_k_m[0] + self.n_k_m[1]]; k_sum = sum(self.n_k_m)
        self.n_x, self.n_h, self.n_k = n_x, n_h, n_k
        if max(3 * n_h, n_k) > ck.max_degree + 1: raise ValueError('committer key too small for this circuit')
        self.H, self.H4 = EvaluationDomain(n_h), EvaluationDomain(4 * n_h)
        self.K_m = [EvaluationDomain(nk) for nk in self.n_k_m]; self.K2_m = [EvaluationDomain(2 * nk) for nk in self.n_k_m]
        self.pos = h_positions(n_public + n_private, n_public, n_x, n_h)
        self.stream = stream or torch.cuda.Stream()
        s = self.stream.cuda_stream
        with torch.cuda.stream(self.stream):
            one = _mont(1); r2 = synth.int_to_limbs(_R2, 4)
            def to_mont(v: _Vec): fr_lin_device(v.ptr(), v.n, None, r2, v.ptr(), stream=s); return v
            # forward matrices on H (z_a = A z, z_b = B z) and the stacked transpose (t = sum_M eta_M M^T r_alpha)
            self.fwd = {}
            trip = []
            for k, m in enumerate('abc'):
                rp, col, val = _csr_on_h(csr[m], self.pos, n_h)
                if m != 'c': self.fwd[m] = (torch.from_numpy(rp.view(np.int32)).cuda(), torch.from_numpy(col.view(np.int32)).cuda(), to_mont(_Vec(len(col), val)))
                rows = np.repeat(np.arange(n_h, dtype=np.int64), np.diff(rp.astype(np.int64)))
                trip.append((col.astype(np.int64), rows + k * n_h, val))
            tc = np.concatenate([t[0] for t in trip]); tr_ = np.concatenate([t[1] for t in trip]); tv = np.concatenate([t[2] for t in trip])
            order = np.argsort(tc, kind='stable')
            tp = np.zeros(n_h + 1, dtype=np.uint32); tp[1:] = np.cumsum(np.bincount(tc, minlength=n_h))
            self.tr = (torch.from_numpy(tp.view(np.int32)).cuda(), torch.from_numpy(tr_[order].astype(np.uint32).view(np.int32)).cuda(), to_mont(_Vec(len(order), tv[order])))
            # elements of H (NTT of the second unit vector), 1 / v_X on H \ X
            e1 = np.zeros((n_h, 4), dtype=np.uint64); e1[1] = one
            self.h_elems = _Vec(n_h, e1); self.H.ntt_device(self.h_elems.ptr(), stream=s)
            ratio = n_h // n_x; wx = pow(_gen(n_h), n_x, R)                            # v_X(w^p) = wx^p − 1 depends on p mod |H|/|X|
            self.vx_inv = _Vec(n_h)
            fr_powers_device(self.vx_inv.ptr(), n_h, one, _mont(wx), s)
            fr_lin_device(self.vx_inv.ptr(), n_h, _mont(R - 1), one, self.vx_inv.ptr(), stream=s)
            batch_inversion_device(self.vx_inv.ptr(), n_h, s)                        # zeros (the positions of X) stay zero
            # arithmetisation over K: row, col, val, row_col
            # per matrix m (|K_m| = n_k_m[m], first element at 4 k_off[m]): row, col, val, row_col, |K_m| values each
            self.k_evals = _Vec(4 * k_sum, zero=True)
            kidx = np.zeros(2 * k_sum, dtype=np.uint32)                              # per matrix: row positions, col positions on H (padding: 0, the element 1)
            host_h = self.h_elems.host()
            n_h_inv = _mont(_inv(n_h))
            for k, m in enumerate('abc'):
                nk = self.n_k_m[k]
                rp, col, val = _csr_on_h(csr[m], self.pos, n_h)
                rows = np.repeat(np.arange(n_h, dtype=np.int64), np.diff(rp.astype(np.int64)))
                nz = len(col)
                kidx[2 * self.k_off[k]:2 * self.k_off[k] + nz] = rows; kidx[2 * self.k_off[k] + nk:2 * self.k_off[k] + nk + nz] = col
                ev = np.tile(one, (4 * nk, 1)).reshape(4, nk, 4)
                ev[0, :nz] = host_h[rows]; ev[1, :nz] = host_h[col.astype(np.int64)]
                ev[2] = 0
                base = self.k_evals.ptr(4 * self.k_off[k])
                self.k_evals.t[4 * self.k_off[k]:4 * self.k_off[k] + 4 * nk] = torch.from_numpy(ev.reshape(-1, 4).view(np.int64)).cuda()
                vraw = to_mont(_Vec(nz, val))
                fr_vec_op_device(base + 2 * nk * 32, vraw.ptr(), base + 1 * nk * 32, nz, OP_MUL, s)          # v * col
                fr_lin_device(base + 2 * nk * 32, nz, None, n_h_inv, base + 2 * nk * 32, stream=s)              # / |H|
                fr_vec_op_device(base + 3 * nk * 32, base, base + nk * 32, nk, OP_MUL, s)                       # row * col
            self.k_idx = torch.from_numpy(kidx.view(np.int32)).cuda()
            self.k_polys = _Vec(4 * k_sum); self.k_polys.t.copy_(self.k_evals.t)
            self.k2_evals = _Vec(8 * k_sum, zero=True)                               # the same polynomials on the domains of size 2|K_m| (first element at 8 k_off[m])
            for k in range(3):
                nk = self.n_k_m[k]
                self.K_m[k].ntt_batch_device(self.k_polys.ptr(4 * self.k_off[k]), 4, direction=INVERSE, stream=s)
                self.k2_evals.t[8 * self.k_off[k]:8 * self.k_off[k] + 8 * nk].view(4, 2 * nk, 4)[:, :nk].copy_(self.k_polys.t[4 * self.k_off[k]:4 * self.k_off[k] + 4 * nk].view(4, nk, 4))
                self.K2_m[k].ntt_batch_device(self.k2_evals.ptr(8 * self.k_off[k]), 4, stream=s)
            self.stream.synchronize()
            self.index_commitments = KZG10.commit_batch_device(ck.bases, [self.k_polys.ptr(4 * self.k_off[k] + j * self.n_k_m[k]) for k in range(3) for j in range(4)],
                                                               [self.n_k_m[k] for k in range(3) for j in range(4)], stream=s)
        self.vk_bytes = wire.g1_compress(self.index_commitments).tobytes() + b''.join(int(v).to_bytes(8, 'little') for v in (n_h, *self.n_k_m, n_x))


MAX_INSTANCES = 32       # as the native host side: one proof covers one transaction (the wrappers in poly.py cut the evaluation / linear-combination / blinding calls into the entry points' 12 / 28 / 24 per call)


def _kp(ix, vec, m, j, mult=1):
    """Pointer to array j (row, col, val, row_col) of matrix m in one of the index's ragged buffers (mult = 2 for the 2|K| evaluations)."""
    return vec.ptr(4 * mult * ix.k_off[m] + j * mult * ix.n_k_m[m])


class _NativeIndex(ctypes.Structure):
    """aleo_mi355x_varuna_index (include/aleo_mi355x.h)."""
    _fields_ = ([(n, ctypes.c_uint64) for n in ('n_h', 'n_k_a', 'n_k_b', 'n_k_c', 'n_x', 'n_public', 'n_vars', 'committer_key', 'max_degree', 'gamma_offset', 'lagrange_offset')] +
                [(n, ctypes.c_void_p) for n in ('positions', 'positions_device', 'a_row_ptr', 'a_col', 'a_val', 'b_row_ptr', 'b_col', 'b_val', 't_row_ptr', 't_col', 't_val',
                                                'vx_inv', 'k_evals', 'k_idx', 'k_polys', 'k2_evals', 'vk_bytes')] + [('vk_len', ctypes.c_size_t), ('vk_affine', ctypes.c_void_p), ('max_row', ctypes.c_uint64 * 3)])


def native_index(ix: CircuitIndex) -> _NativeIndex:
    """The index as the C ABI takes it: sizes, the committer key's handle, device pointers (the arrays stay owned by `ix`)."""
    if getattr(ix, '_native', None) is not None: return ix._native
    n = _NativeIndex()
    n.n_h, n.n_x, n.n_public, n.n_vars = ix.n_h, ix.n_x, ix.n_public, ix.n_public + ix.n_private
    n.n_k_a, n.n_k_b, n.n_k_c = ix.n_k_m
    n.committer_key, n.max_degree, n.gamma_offset = ix.ck.bases.handle, ix.ck.max_degree, ix.ck.gamma_offset
    n.lagrange_offset = ix.ck.lagrange_offset if ix.ck.lagrange_size == ix.n_h else 0
    ix._pos32 = np.ascontiguousarray(ix.pos, dtype=np.uint32); ix._vk = np.frombuffer(ix.vk_bytes, dtype=np.uint8).copy()
    ix._pos_dev = torch.from_numpy(ix._pos32.view(np.int32)).cuda()
    n.positions = ix._pos32.ctypes.data; n.positions_device = ix._pos_dev.data_ptr(); n.vk_bytes = ix._vk.ctypes.data; n.vk_len = ix._vk.shape[0]
    ix._vk_aff = np.ascontiguousarray(ix.index_commitments, dtype=np.uint8).reshape(12, 104); n.vk_affine = ix._vk_aff.ctypes.data
    for m in 'ab':
        rp, col, val = ix.fwd[m]
        setattr(n, m + '_row_ptr', rp.data_ptr()); setattr(n, m + '_col', col.data_ptr()); setattr(n, m + '_val', val.ptr())
    n.t_row_ptr, n.t_col, n.t_val = ix.tr[0].data_ptr(), ix.tr[1].data_ptr(), ix.tr[2].ptr()
    for i, rp in enumerate((ix.fwd['a'][0], ix.fwd['b'][0], ix.tr[0])):          # hints: the longest row of each sparse product (the prover skips launches that cannot have work)
        n.max_row[i] = max(1, int((rp[1:] - rp[:-1]).max().item())) if rp.numel() > 1 else 1
    n.vx_inv, n.k_evals, n.k_idx, n.k_polys, n.k2_evals = ix.vx_inv.ptr(), ix.k_evals.ptr(), ix.k_idx.data_ptr(), ix.k_polys.ptr(), ix.k2_evals.ptr()
    ix._native = n
    return n


def prove_native(index: CircuitIndex, assignment, seed=None) -> bytes:
    """The same proof through ONE call of the C ABI (aleo_mi355x_varuna_prove: transcript, constants and all launches in C++); returns the
    proof bytes.  seed: 32 bytes of entropy (None = os.urandom; an int only for reproducible tests).  Thread-safe: concurrent calls are served
    by separate slots of the library."""
    if isinstance(assignment, np.ndarray) and assignment.ndim == 2: assignment = [assignment]
    zs = [np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4) for a in assignment]
    nv = index.n_public + index.n_private
    if any(z.shape[0] != nv for z in zs): raise ValueError('assignment length differs from the number of variables of the circuit')
    ptrs = (ctypes.c_void_p * len(zs))(*[z.ctypes.data for z in zs])
    out = np.zeros(1100 + 200 * len(zs), dtype=np.uint8); n = ctypes.c_size_t(out.shape[0])
    check(lib().aleo_mi355x_varuna_prove(ctypes.byref(native_index(index)), ptrs, len(zs), seed32(seed), out.ctypes.data_as(ctypes.c_void_p),
                                         ctypes.byref(n)), 'varuna_prove')
    return out[:n.value].tobytes()


def prove_batch_native(indexes, assignments, seed=None) -> bytes:
    """One proof over several circuits (`Varuna::prove_batch` with a map of proving keys): indexes = NativeCircuitIndex objects built against one
    committer key, assignments[j] = the list of instances (uint64[n_vars, 4] canonical) of circuit j.  One call of the C ABI
    (aleo_mi355x_varuna_prove_batch_indexed); returns the proof bytes."""
    zs, ks = [], []
    for ix, inst in zip(indexes, assignments):
        if isinstance(inst, np.ndarray) and inst.ndim == 2: inst = [inst]
        rows = [np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4) for a in inst]
        if any(z.shape[0] != ix.n_vars for z in rows): raise ValueError('assignment length differs from the number of variables of its circuit')
        zs += rows; ks.append(len(rows))
    if len(ks) != len(indexes): raise ValueError('one list of instances per circuit')
    handles = (ctypes.c_uint64 * len(indexes))(*[ix.handle for ix in indexes]); counts = (ctypes.c_size_t * len(ks))(*ks)
    ptrs = (ctypes.c_void_p * len(zs))(*[z.ctypes.data for z in zs])
    out = np.zeros(1200 + 400 * len(ks) + 200 * len(zs), dtype=np.uint8); n = ctypes.c_size_t(out.shape[0])
    check(lib().aleo_mi355x_varuna_prove_batch_indexed(handles, len(indexes), ptrs, counts, seed32(seed), out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)),
          'varuna_prove_batch')
    return out[:n.value].tobytes()


class _ProveRequest(ctypes.Structure):
    """aleo_mi355x_prove_request (include/aleo_mi355x.h)."""
    _fields_ = [('index_handles', ctypes.c_void_p), ('n_circuits', ctypes.c_size_t), ('assignments', ctypes.c_void_p), ('n_instances', ctypes.c_void_p),
                ('seed', ctypes.c_void_p), ('out_proof', ctypes.c_void_p), ('len', ctypes.c_size_t), ('status', ctypes.c_int32)]


def prove_many_native(requests):
    """Several independent proofs in lockstep (aleo_mi355x_varuna_prove_many): requests = [(indexes, assignments, seed), ...] with the arguments of
    prove_batch_native each.  Returns a list with, per request, the proof bytes — or the status code (int) of a request that dropped out (6: an
    assignment violates its circuit).  Every proof equals what prove_batch_native gives for the same arguments."""
    keep = []; arr = (_ProveRequest * len(requests))()
    for q, (indexes, assignments, seed) in enumerate(requests):
        zs, ks = [], []
        for ix, inst in zip(indexes, assignments):
            if isinstance(inst, np.ndarray) and inst.ndim == 2: inst = [inst]
            rows = [np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4) for a in inst]
            if any(z.shape[0] != ix.n_vars for z in rows): raise ValueError('assignment length differs from the number of variables of its circuit')
            zs += rows; ks.append(len(rows))
        handles = (ctypes.c_uint64 * len(indexes))(*[ix.handle for ix in indexes]); counts = (ctypes.c_size_t * len(ks))(*ks)
        ptrs = (ctypes.c_void_p * len(zs))(*[z.ctypes.data for z in zs]); sd = seed32(seed)
        out = np.zeros(1200 + 400 * len(ks) + 200 * len(zs), dtype=np.uint8)
        keep.append((zs, handles, counts, ptrs, sd, out))
        r = arr[q]; r.index_handles = ctypes.addressof(handles); r.n_circuits = len(indexes); r.assignments = ctypes.addressof(ptrs); r.n_instances = ctypes.addressof(counts)
        r.seed = ctypes.addressof(sd); r.out_proof = out.ctypes.data; r.len = out.shape[0]; r.status = 0
    check(lib().aleo_mi355x_varuna_prove_many(ctypes.byref(arr), len(requests)), 'varuna_prove_many')
    return [keep[q][5][:arr[q].len].tobytes() if arr[q].status == 0 else int(arr[q].status) for q in range(len(requests))]


class Trace:
    """snarkvm_synthesizer_process::Trace as the prover sees it (the C++ mirror is `aleo_mi355x::Trace`): the transitions of one
    transaction, each the proving key (NativeCircuitIndex) of its function and the assignment its execution produced.  prove_execution / prove_fee group
    the assignments per key in order of first appearance and make ONE proof for all of them — the call under `trace.prove_execution::<A, _>(locator, rng)`
    at /root/reference/rust/src/program/execute.rs:74."""
    def __init__(self): self._keys, self._assignments = [], []

    def insert_transition(self, index: 'NativeCircuitIndex', assignment: np.ndarray):
        for i, k in enumerate(self._keys):
            if k is index: self._assignments[i].append(assignment); return
        self._keys.append(index); self._assignments.append([assignment])

    @property
    def transitions(self) -> int: return sum(len(a) for a in self._assignments)

    def prove_execution(self, seed=None) -> bytes:
        if not self._keys: raise ValueError('Trace.prove_execution: no transitions')
        return prove_batch_native(self._keys, self._assignments, seed)

    def prove_fee(self, seed=None) -> bytes:
        if self.transitions != 1: raise ValueError('Trace.prove_fee: a fee is exactly one transition')
        return prove_batch_native(self._keys, self._assignments, seed)


class _R1csMatrix(ctypes.Structure):
    _fields_ = [('row_ptr', ctypes.c_void_p), ('col', ctypes.c_void_p), ('val', ctypes.c_void_p)]


class NativeCircuitIndex:
    """The index built and owned by the library (aleo_mi355x_varuna_index_build): the key-synthesis step of one circuit in one call.
    csr[m] = (row_ptr uint32[n+1], col uint32[nnz] variable indices, val uint64[nnz,4] canonical) for m in 'abc'."""

    def __init__(self, csr, n_constraints: int, n_public: int, n_private: int, ck: CommitterKey, domains: str = 'auto'):
        if domains not in DOMAIN_FLAGS: raise ValueError('domains: auto, per_matrix or shared')
        self.ck = ck; keep = []; mats = (_R1csMatrix * 3)()
        for i, m in enumerate('abc'):
            rp, col, val = (np.ascontiguousarray(csr[m][0], dtype=np.uint32), np.ascontiguousarray(csr[m][1], dtype=np.uint32),
                            np.ascontiguousarray(csr[m][2], dtype=np.uint64).reshape(-1, 4))
            keep += [rp, col, val]
            mats[i].row_ptr, mats[i].col, mats[i].val = rp.ctypes.data, col.ctypes.data, val.ctypes.data
        h = ctypes.c_uint64(0)
        check(lib().aleo_mi355x_varuna_index_build(ctypes.byref(h), ck.bases.handle, ck.max_degree, ck.gamma_offset, self._lagrange_offset(csr, n_constraints, n_public, n_private, ck), mats, n_constraints, n_public, n_private, DOMAIN_FLAGS[domains]),
              'varuna_index_build')
        self.handle = h.value
        view = _NativeIndex(); check(lib().aleo_mi355x_varuna_index_export(self.handle, ctypes.byref(view)), 'varuna_index_export')
        self.n_h, self.n_x, self.n_vars = view.n_h, view.n_x, view.n_vars
        self.n_k_m = [view.n_k_a, view.n_k_b, view.n_k_c]; self.n_k = max(self.n_k_m)
        buf = np.zeros(12 * 48 + 40, dtype=np.uint8); n = ctypes.c_size_t(buf.shape[0])
        check(lib().aleo_mi355x_varuna_index_vk(self.handle, buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)), 'varuna_index_vk')
        self.vk_bytes = buf[:n.value].tobytes()

    @staticmethod
    def _lagrange_offset(csr, n_constraints, n_public, n_private, ck):
        n_x = 1
        while n_x < n_public: n_x *= 2
        n_h = 1
        while n_h < max(n_constraints, n_x + n_private, 2 * n_x): n_h *= 2
        return ck.lagrange_offset if ck.lagrange_size == n_h else 0      # only the Lagrange powers of THIS circuit's domain are of use

    def prove(self, assignment, seed=None) -> bytes:
        if isinstance(assignment, np.ndarray) and assignment.ndim == 2: assignment = [assignment]
        zs = [np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4) for a in assignment]
        if any(z.shape[0] != self.n_vars for z in zs): raise ValueError('assignment length differs from the number of variables of the circuit')
        ptrs = (ctypes.c_void_p * len(zs))(*[z.ctypes.data for z in zs])
        out = np.zeros(1100 + 200 * len(zs), dtype=np.uint8); n = ctypes.c_size_t(out.shape[0])
        check(lib().aleo_mi355x_varuna_prove_indexed(self.handle, ptrs, len(zs), seed32(seed), out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)),
              'varuna_prove_indexed')
        return out[:n.value].tobytes()

    def close(self):
        if self.handle: lib().aleo_mi355x_varuna_index_free(self.handle); self.handle = 0
    def __enter__(self): return self
    def __exit__(self, *a): self.close()


def native_timing() -> dict:
    t = (ctypes.c_double * 8)(); lib().aleo_mi355x_varuna_last_timing(t, 8)
    return dict(zip(('round1', 'round2', 'round3', 'round4', 'openings', 'total', 'commit_calls', 'commit_host_tails'), t))


class Proof:
    def __init__(self, witness, commitments, evaluations, sums, openings):
        self.witness, self.commitments, self.evaluations, self.sums, self.openings = witness, commitments, evaluations, sums, openings

    def to_bytes(self) -> bytes:
        c = self.commitments
        return wire.proof_to_bytes([self.witness.shape[0] // 3], self.witness, c['mask'], c['g_1'], c['h_1'], np.stack([c['g_a'], c['g_b'], c['g_c']]), c['h_2'],
                                   _mont_rows(self.evaluations), _mont_rows(self.sums), np.stack([o[0] for o in self.openings]),
                                   [None if o[1] is None else _mont(o[1]) for o in self.openings])

    def to_string(self) -> str: return wire.proof_to_string(self.to_bytes())


def randomness_layout(n_h, k=1):
    """Positions in the proof's random stream: rho_w, rho_a, rho_b per instance, the mask's 3|H| coefficients, the hiding polynomials of
    w_i, z_a,i, z_b,i per instance and of the mask."""
    o = {'rho': [3 * i for i in range(k)], 'mask': 3 * k}
    base = 3 * k + 3 * n_h
    o['blind'] = [base + 3 * HIDING_COEFFS * i for i in range(k)]
    o['blind_mask'] = base + 3 * HIDING_COEFFS * k
    o['total'] = o['blind_mask'] + HIDING_COEFFS
    return o


class Prover:
    """State of one proof (one circuit, k instances): the round functions in the order upstream calls them."""

    def __init__(self, index: CircuitIndex, assignments, seed=None, stream: torch.cuda.Stream = None):
        """assignments: one canonical uint64[n_vars,4] array per instance (or a single array), public variables first (z_0 = 1); seed: of the
        proof's random stream (poly.random_fr): the device draws the mask polynomial from it, the host the blinding scalars."""
        self.ix = index; self.stream = stream or index.stream; self.s = self.stream.cuda_stream
        if isinstance(assignments, np.ndarray) and assignments.ndim == 2: assignments = [assignments]
        self.z = [np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4) for a in assignments]
        self.k = len(self.z)
        if not 1 <= self.k <= MAX_INSTANCES: raise ValueError('between 1 and %d instances per proof' % MAX_INSTANCES)
        self.seed = bytes(seed32(seed))
        self.lay = randomness_layout(index.n_h, self.k)
        self.fs = FiatShamir(); self.c = {}

    def _ri(self, first, n=1): v = random_fr(self.seed, first, n); return v if n > 1 else [v]

    # ---- round 1 ------------------------------------------------------------------------------------------------------------------
    def first_round(self):
        ix, s, k = self.ix, self.s, self.k; n_h, n_x = ix.n_h, ix.n_x
        r2 = synth.int_to_limbs(_R2, 4); one = _mont(1); neg1 = _mont(R - 1)
        gx_inv = _inv(_gen(n_x)); nxi = _inv(n_x)
        L = n_h + 1
        ev = _Vec(3 * k * n_h)                                                      # w_i, z_a,i, z_b,i on H, instance after instance
        self.x_evals, self.x_poly = [], []
        for i, z in enumerate(self.z):
            zh = np.zeros((n_h, 4), dtype=np.uint64); zh[ix.pos[:z.shape[0]]] = z
            xe = [synth.limbs_to_int(z[j]) if j < ix.n_public else 0 for j in range(n_x)]
            xp = [v * nxi % R for v in _host_ntt(xe, gx_inv)]                            # inverse DFT over X on the host (|X| = padded number of public inputs)
            self.x_evals.append(xe); self.x_poly.append(xp)
            zH = _Vec(n_h, zh); fr_lin_device(zH.ptr(), n_h, None, r2, zH.ptr(), stream=s)
            base = 3 * i * n_h
            for j, m in ((1, 'a'), (2, 'b')):
                rp, col, val = ix.fwd[m]
                spmv_device(ev.ptr(base + j * n_h), rp.data_ptr(), col.data_ptr(), val.ptr(), zH.ptr(), n_h, s)
            xh = _Vec(n_h, zero=True); xh.t[:n_x] = torch.from_numpy(_mont_rows(xp).view(np.int64)).cuda()
            ix.H.ntt_device(xh.ptr(), stream=s)
            fr_vec_op_device(ev.ptr(base), zH.ptr(), xh.ptr(), n_h, OP_SUB, s)
            fr_vec_op_device(ev.ptr(base), ev.ptr(base), ix.vx_inv.ptr(), n_h, OP_MUL, s)
        lagrange = ix.ck.lagrange_offset != 0 and ix.ck.lagrange_size == n_h      # KZG10::commit_lagrange for w, z_a, z_b: commit their evaluations
        if lagrange: evals_h = _Vec(3 * k * n_h); evals_h.t.copy_(ev.t)
        ix.H.ntt_batch_device(ev.ptr(), 3 * k, direction=INVERSE, stream=s)
        self.wit = _Vec(3 * k * L)                                                  # w_i, z_a,i, z_b,i as polynomials of |H| + 1 coefficients
        self.blind = []; blinds = []; rhos = []
        for q in range(3 * k):
            i, j = divmod(q, 3)
            rhos.append(self._ri(self.lay['rho'][i] + j)[0])
            bq = self._ri(self.lay['blind'][i] + HIDING_COEFFS * j, HIDING_COEFFS); self.blind.append(bq); blinds += bq
        fr_blind_rows_device(self.wit.ptr(), ev.ptr(), n_h, _mont_rows(rhos), s)            # + rho_q (X^|H| − 1), all 3k polynomials in one launch
        self.w = lambda i: self.wit.ptr((3 * i) * L); self.za = lambda i: self.wit.ptr((3 * i + 1) * L); self.zb = lambda i: self.wit.ptr((3 * i + 2) * L)
        self.mask = _Vec(3 * n_h)                                                   # drawn in HBM; its sum over H is made zero through m_0
        fr_random_device(self.mask.ptr(), 3 * n_h, self.seed, self.lay['mask'], True, s)
        fr_lin_device(self.mask.ptr(), 1, None, neg1, self.mask.ptr(n_h), neg1, self.mask.ptr(2 * n_h), stream=s)
        self.blind_mask = self._ri(self.lay['blind_mask'], HIDING_COEFFS)
        bl = _Vec((3 * k + 1) * HIDING_COEFFS, _mont_rows(blinds + self.blind_mask))
        segs = []
        if lagrange: rho_dev = _Vec(3 * k, _mont_rows(rhos))
        for q in range(3 * k):
            if lagrange: segs += [(evals_h.ptr(q * n_h), n_h, ix.ck.lagrange_offset, q), (rho_dev.ptr(q), 1, ix.ck.lagrange_offset + n_h, q)]   # sum_i evals_i L_i(tau) G + rho v_H(tau) G
            else: segs.append((self.wit.ptr(q * L), L, 0, q))
            segs.append((bl.ptr(HIDING_COEFFS * q), HIDING_COEFFS, ix.ck.gamma_offset, q))
        if lagrange and getattr(ix.ck, 'sparse_range', False):                          # the witness commitments as one sparse chain, the mask as another
            out = np.concatenate([SonicKZG10.commit_segments_device(ix.ck, segs, 3 * k, s, sparse=True),
                                  SonicKZG10.commit_segments_device(ix.ck, [(self.mask.ptr(), 3 * n_h, 0, 0), (bl.ptr(HIDING_COEFFS * 3 * k), HIDING_COEFFS, ix.ck.gamma_offset, 0)], 1, s)])
        else:
            segs += [(self.mask.ptr(), 3 * n_h, 0, 3 * k), (bl.ptr(HIDING_COEFFS * 3 * k), HIDING_COEFFS, ix.ck.gamma_offset, 3 * k)]
            out = SonicKZG10.commit_segments_device(ix.ck, segs, 3 * k + 1, s)
        self.witness_commitments = out[:3 * k]; self.c['mask'] = out[3 * k]
        fs = self.fs                                                                # Varuna::init_sponge [UPSTREAM-RECALL], then the first commitments
        fs.absorb_bytes(PROTOCOL_NAME); fs.absorb_bytes(k.to_bytes(8, 'little'))
        for xe in self.x_evals: fs.absorb_fr(xe)
        fs.absorb_g1(ix.index_commitments); fs.absorb_g1(out)
        self.comb = [1] + fs.squeeze(k - 1)                                         # one circuit: k − 1 instance combiners, then alpha, eta_b, eta_c
        self.alpha, self.eta_b, self.eta_c = fs.squeeze(3)

    # ---- round 2: the first sumcheck -----------------------------------------------------------------------------------------------
    def second_round(self):
        ix, s, k = self.ix, self.s, self.k; n_h, n_x = ix.n_h, ix.n_x
        one = _mont(1); neg1 = _mont(R - 1)
        alpha = self.alpha; vh_alpha = _vanish(n_h, alpha)
        if vh_alpha == 0: raise ArithmeticError('alpha landed in H')
        ext = self.r_alpha = _Vec(3 * n_h)                                          # r_alpha, eta_b r_alpha, eta_c r_alpha on H
        rt = _Vec(2 * n_h)                                                          # r(alpha, X) = sum_k alpha^(|H|-1-k) X^k, then t(X)
        fr_powers_device(rt.ptr(), n_h, _mont(pow(alpha, n_h - 1, R)), _mont(_inv(alpha)), s)
        ext.t[:n_h].copy_(rt.t[:n_h]); ix.H.ntt_device(ext.ptr(), stream=s)           # v_H(alpha) / (alpha − h): no inversion on the device
        fr_lin_device(ext.ptr(n_h), n_h, None, _mont(self.eta_b), ext.ptr(), stream=s)
        fr_lin_device(ext.ptr(2 * n_h), n_h, None, _mont(self.eta_c), ext.ptr(), stream=s)
        tp, tcol, tval = ix.tr
        spmv_device(rt.ptr(n_h), tp.data_ptr(), tcol.data_ptr(), tval.ptr(), ext.ptr(), n_h, s)
        ix.H.ntt_device(rt.ptr(n_h), direction=INVERSE, stream=s)                     # t(X)
        L = n_h + 1; n4 = 4 * n_h
        E = _Vec((2 + 3 * k) * n4)                                                  # r, t, then z_i, z_a,i, z_b,i on the domain of size 4|H|
        E.t[:2 * n4].zero_()
        E.t[0:n_h].copy_(rt.t[:n_h]); E.t[n4:n4 + n_h].copy_(rt.t[n_h:])
        xp = _Vec(k * n_x, _mont_rows([v for i in range(k) for v in self.x_poly[i]]))
        ahp_sumcheck_operands_device(E.ptr(2 * n4), self.wit.ptr(), xp.ptr(), n_h, n_x, k, s)   # ẑ_i = w_i (X^|X| − 1) + x̂_i, z_a,i, z_b,i: every row written in full
        ix.H4.ntt_batch_device(E.ptr(), 2 + 3 * k, stream=s)
        e_r, e_t = E.ptr(0), E.ptr(n4)
        for i in range(k):                                                          # numerator of instance i, in place over its z_a row
            e_z, e_a, e_b = (E.ptr((2 + 3 * i + j) * n4) for j in range(3))
            ahp_first_sumcheck_device(e_a, n4, e_r, e_a, e_b, e_t, e_z, _mont(self.eta_b), _mont(self.eta_c), s)
        if k == 1: q1 = E.ptr(3 * n4)
        else:                                                                       # sum_i c_i numerator_i
            Q = _Vec(n4); q1 = Q.ptr()
            fr_lincomb_device(q1, n4, None, [(E.ptr((3 + 3 * i) * n4), n4, _mont(self.comb[i])) for i in range(k)], s)
        ix.H4.ntt_device(q1, direction=INVERSE, stream=s)
        fr_vec_op_device(q1, q1, self.mask.ptr(), 3 * n_h, OP_ADD, s)                 # q_1 = h_1 (X^|H| − 1) + X g_1, degree < 3|H|
        self.h1 = _Vec(2 * n_h); self.g1 = _Vec(n_h)
        p0, p1, p2 = q1, q1 + 32 * n_h, q1 + 64 * n_h
        fr_lin_device(self.h1.ptr(n_h), n_h, None, one, p2, stream=s)                 # quotient blocks: p2, p1 + p2; remainder p0 + p1 + p2
        fr_vec_op_device(self.h1.ptr(), p1, p2, n_h, OP_ADD, s)
        fr_vec_op_device(self.g1.ptr(), p0, self.h1.ptr(), n_h, OP_ADD, s)             # remainder; its constant term is the sum over H / |H| = 0
        out = SonicKZG10.commit(ix.ck, [((self.g1.ptr(1), n_h - 1), n_h - 2, None), ((self.h1.ptr(), 2 * n_h), None, None)], device=True, stream=s)
        if self.g1.host(0, 1).any():                                                   # after the commitments (the stream has drained): 32 bytes
            raise UnsatisfiedAssignment('the assignment does not satisfy the circuit (first sumcheck: the sum over H is not zero)')
        self.c['g_1'], self.c['h_1'] = out[0], out[1]
        self.fs.absorb_g1(out)
        self.beta = self.fs.squeeze(1)[0]

    # ---- round 3: three rational sumchecks over K -----------------------------------------------------------------------------------
    def third_round(self):
        ix, s = self.ix, self.s; n_h, n_k = ix.n_h, ix.n_k
        vh_beta = _vanish(n_h, self.beta)
        if vh_beta == 0: raise ArithmeticError('beta landed in H')
        self.vv = _vanish(n_h, self.alpha) * vh_beta % R
        km, ko = ix.n_k_m, ix.k_off
        self.f = _Vec(sum(km))                                                      # f_M at offset k_off[M], |K_M| values
        self.fp = lambda m, off=0: self.f.ptr(ko[m] + off)
        rb = _Vec(n_h)                                                              # v_H(beta) / (beta − h) on H, the same way
        fr_powers_device(rb.ptr(), n_h, _mont(pow(self.beta, n_h - 1, R)), _mont(_inv(self.beta)), s)
        ix.H.ntt_device(rb.ptr(), stream=s)
        for m in range(3):                                                          # f_M = val u_H(alpha, row) u_H(beta, col) on K_M: two gathers
            fr_gather_mul_device(self.fp(m), km[m], _kp(ix, ix.k_evals, m, 2), self.r_alpha.ptr(), ix.k_idx.data_ptr() + 4 * 2 * ko[m],
                                 rb.ptr(), ix.k_idx.data_ptr() + 4 * (2 * ko[m] + km[m]), s)
        for m0, cnt in _runs(km): ix.K_m[m0].ntt_batch_device(self.fp(m0), cnt, direction=INVERSE, stream=s)
        self.stream.synchronize()
        f0 = self.f.t[ko].cpu().numpy().view(np.uint64)
        self.sigma = [_from_mont(f0[m]) * km[m] % R for m in range(3)]
        out = SonicKZG10.commit(ix.ck, [((self.fp(m, 1), km[m] - 1), km[m] - 2, None) for m in range(3)], device=True, stream=s)
        for k, name in enumerate(('g_a', 'g_b', 'g_c')): self.c[name] = out[k]
        self.fs.absorb_g1(out); self.fs.absorb_fr(self.sigma)
        self.delta = [1] + self.fs.squeeze(2)

    # ---- round 4 ----------------------------------------------------------------------------------------------------------------------
    def fourth_round(self):
        ix, s = self.ix, self.s; n_k = ix.n_k; km, ko = ix.n_k_m, ix.k_off
        F = _Vec(2 * sum(km), zero=True); B = _Vec(2 * sum(km))                     # per matrix on its own domain of size 2|K_M|
        terms = []
        for m0, cnt in _runs(km):                                                   # matrices with equal domains share the transforms and one numerator pass
            n2 = 2 * km[m0]; mem = list(range(m0, m0 + cnt))
            for m in mem: F.t[2 * ko[m]:2 * ko[m] + km[m]].copy_(self.f.t[ko[m]:ko[m] + km[m]])
            ix.K2_m[m0].ntt_batch_device(F.ptr(2 * ko[m0]), cnt, stream=s)
            pad = [0] * (3 - cnt)
            consts = _mont_rows([self.delta[m] for m in mem] + pad + [self.alpha * self.beta, -self.alpha, -self.beta, self.vv])
            ahp_matrix_sumcheck_device(B.ptr(2 * ko[m0]), n2, [_kp(ix, ix.k2_evals, m, 0, 2) for m in mem] + pad, n2, [F.ptr(2 * ko[m]) for m in mem] + pad, consts, s)
            ix.K2_m[m0].ntt_device(B.ptr(2 * ko[m0]), direction=INVERSE, stream=s)     # sum over the run of delta_M (vv val_M − b_M f_M) = h (X^|K| − 1)
            terms.append((B.ptr(2 * ko[m0] + km[m0]), km[m0], _mont(1)))             # its upper half
        self.h2 = _Vec(n_k)
        fr_lincomb_device(self.h2.ptr(), n_k, None, terms, s)                         # h_2 = sum_M delta_M h_M
        out = SonicKZG10.commit(ix.ck, [((self.h2.ptr(), n_k), None, None)], device=True, stream=s)
        self.c['h_2'] = out[0]
        self.fs.absorb_g1(out)
        self.gamma = self.fs.squeeze(1)[0]

    # ---- evaluations and openings ------------------------------------------------------------------------------------------------------
    def finish(self) -> Proof:
        ix, s, k = self.ix, self.s, self.k; n_h, n_k, n_x = ix.n_h, ix.n_k, ix.n_x
        one = _mont(1); L = n_h + 1
        self._ev = _Vec(k + 8)
        km = ix.n_k_m
        fr_eval_batch_device(self._ev.ptr(), [self.zb(i) for i in range(k)] + [self.g1.ptr(1)] + [self.fp(m, 1) for m in range(3)],
                             [L] * k + [n_h - 1] + [km[m] - 1 for m in range(3)], _mont_rows([self.beta] * (k + 1) + [self.gamma] * 3), s)
        self.stream.synchronize()
        evals = [_from_mont(v) for v in self._ev.host(0, k + 4)]
        zb_beta = evals[:k]; g1_beta, ga, gb, gc = evals[k:]
        self.fs.absorb_fr(evals)                                                    # one circuit: the serialised order is this one
        ch_b = [self.fs.squeeze_short() for _ in range(k + 2)]                      # one short challenge per opened polynomial: g_1, z_b,i, LC1 at beta;
        ch_g = [self.fs.squeeze_short() for _ in range(4)]                          # g_a, g_b, g_c, LC2 at gamma
        alpha, beta, gamma = self.alpha, self.beta, self.gamma
        # linear combination of the first sumcheck, opened at beta together with g_1 and the z_b,i
        r_ab = (_vanish(n_h, alpha) - _vanish(n_h, beta)) * _inv(alpha - beta) % R
        t_beta = (self.sigma[0] + self.eta_b * self.sigma[1] + self.eta_c * self.sigma[2]) % R
        xl = ch_b[k + 1]; const = (-beta * g1_beta) % R
        terms = [(self.mask.ptr(), 3 * n_h, _mont(xl)), (self.h1.ptr(), 2 * n_h, _mont(-xl * _vanish(n_h, beta))), (self.g1.ptr(1), n_h - 1, _mont(ch_b[0]))]
        bl = [0] * HIDING_COEFFS
        def axpy(coef, src):
            for j, v in enumerate(src): bl[j] = (bl[j] + coef * v) % R
        axpy(xl, self.blind_mask)
        for i in range(k):
            x_beta = 0
            for v in reversed(self.x_poly[i]): x_beta = (x_beta * beta + v) % R
            ci = self.comb[i]
            k_za = xl * ci % R * r_ab % R * (1 + self.eta_c * zb_beta[i]) % R; k_w = (-xl * ci % R * t_beta % R * _vanish(n_x, beta)) % R; k_zb = ch_b[1 + i]
            const = (const + ci * (r_ab * self.eta_b % R * zb_beta[i] - t_beta * x_beta)) % R
            terms += [(self.za(i), L, _mont(k_za)), (self.w(i), L, _mont(k_w)), (self.zb(i), L, _mont(k_zb))]
            axpy(k_w, self.blind[3 * i]); axpy(k_za, self.blind[3 * i + 1]); axpy(k_zb, self.blind[3 * i + 2])
        pb = _Vec(3 * n_h)
        fr_lincomb_device(pb.ptr(), 3 * n_h, _mont(xl * const), terms, s)
        random_v = 0
        for v in reversed(bl): random_v = (random_v * beta + v) % R
        blw = [0] * (HIDING_COEFFS - 1); acc = 0
        for j in range(HIDING_COEFFS - 1, 0, -1): acc = (bl[j] + beta * acc) % R; blw[j - 1] = acc
        wq = _Vec(3 * n_h); blq = _Vec(HIDING_COEFFS - 1, _mont_rows(blw))
        divide_by_linear_device(wq.ptr(), self._ev.ptr(k + 5), pb.ptr(), 3 * n_h, _mont(beta), s)
        # linear combination of the second sumcheck, opened at gamma together with g_a, g_b, g_c
        xi3 = ch_g[3]; vk_gamma = _vanish(n_k, gamma)                                # K: the largest non-zero domain
        pg = _Vec(n_k); const = 0; terms = []
        for m, gk in enumerate((ga, gb, gc)):
            fm = (gamma * gk + self.sigma[m] * _inv(km[m])) % R
            d = self.delta[m] * xi3 % R * vk_gamma % R * _inv(_vanish(km[m], gamma)) % R          # selector v_K / v_{K_M} at gamma
            for j, coef in ((2, d * self.vv), (0, d * fm % R * beta), (1, d * fm % R * alpha), (3, -d * fm)):       # val, row, col, row_col
                terms.append((_kp(ix, ix.k_polys, m, j), km[m], _mont(coef)))
            const = (const - d * fm % R * alpha % R * beta) % R
        terms.append((self.h2.ptr(), n_k, _mont(-xi3 * vk_gamma)))
        terms += [(self.fp(m, 1), km[m] - 1, _mont(ch_g[m])) for m in range(3)]
        fr_lincomb_device(pg.ptr(), n_k, _mont(const), terms, s)
        gq = _Vec(n_k)
        divide_by_linear_device(gq.ptr(), self._ev.ptr(k + 6), pg.ptr(), n_k, _mont(gamma), s)
        opn = SonicKZG10.commit(ix.ck, [((wq.ptr(), 3 * n_h - 1), None, (blq.ptr(), HIDING_COEFFS - 1)), ((gq.ptr(), n_k - 1), None, None)], device=True, stream=s)
        open_beta, open_gamma = opn[0], opn[1]                                      # both witness commitments in one call
        return Proof(self.witness_commitments, dict(self.c), evals, list(self.sigma), [(open_beta, random_v), (open_gamma, None)])


def prove(index: CircuitIndex, assignment, seed=None, stream: torch.cuda.Stream = None) -> Proof:
    """Varuna::prove_batch for one circuit with one to eight instances (one assignment array, or a list of them); `seed`: 32 bytes of entropy for the proof's
    random stream (None = os.urandom; an int only for reproducible tests).  Proofs of one index may be
    produced concurrently from several host threads, each on its own `stream` (the index is read-only while proving)."""
    import time
    with torch.cuda.stream(stream or index.stream):
        p = Prover(index, assignment, seed, stream); t = [time.perf_counter()]
        for step in (p.first_round, p.second_round, p.third_round, p.fourth_round):
            step(); t.append(time.perf_counter())                     # every round ends on its commitments: the host has them
        proof = p.finish(); t.append(time.perf_counter())
    proof.timing_ms = {k: (t[i + 1] - t[i]) * 1e3 for i, k in enumerate(('round1', 'round2', 'round3', 'round4', 'openings'))}
    return proof

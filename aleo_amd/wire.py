"""Host-side mirror of the serialisation the reference applies to prover results (snarkVM 0.14.5 CanonicalSerialize of G1Affine,
Fr::to_bytes_le, Proof::to_bytes_le and its bech32m Display [UPSTREAM-RECALL]; the reference round-trips one such string in
/root/reference/wasm/src/programs/transaction.rs:104-120).  Thin bindings over the C ABI (aleo_amd/csrc/wire.hip); host code only."""
from __future__ import annotations
import ctypes
import numpy as np
from ._lib import lib, check


def _p(a): return a.ctypes.data_as(ctypes.c_void_p)


def g1_compress(affine104: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(affine104, dtype=np.uint8).reshape(-1, 104)
    out = np.zeros((a.shape[0], 48), dtype=np.uint8)
    check(lib().aleo_mi355x_g1_compress(_p(out), _p(a), a.shape[0]), 'g1_compress')
    return out


def g1_decompress(compressed48: np.ndarray, check_subgroup: bool = True) -> np.ndarray:
    b = np.ascontiguousarray(compressed48, dtype=np.uint8).reshape(-1, 48)
    out = np.zeros((b.shape[0], 104), dtype=np.uint8)
    check(lib().aleo_mi355x_g1_decompress(_p(out), _p(b), b.shape[0], 1 if check_subgroup else 0), 'g1_decompress')
    return out


def fr_to_bytes(fr_mont: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(fr_mont, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((a.shape[0], 32), dtype=np.uint8)
    check(lib().aleo_mi355x_fr_to_bytes(_p(out), _p(a), a.shape[0]), 'fr_to_bytes')
    return out


def fr_from_bytes(b32: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(b32, dtype=np.uint8).reshape(-1, 32)
    out = np.zeros((b.shape[0], 4), dtype=np.uint64)
    check(lib().aleo_mi355x_fr_from_bytes(_p(out), _p(b), b.shape[0]), 'fr_from_bytes')
    return out


def bech32m_encode(hrp: str, data: bytes) -> str:
    cap = len(hrp) + 1 + (len(data) * 8 + 4) // 5 + 6 + 1
    buf = ctypes.create_string_buffer(cap)
    raw = np.frombuffer(bytes(data), dtype=np.uint8) if data else np.zeros(1, dtype=np.uint8)
    check(lib().aleo_mi355x_bech32m_encode(buf, cap, hrp.encode(), _p(raw), len(data)), 'bech32m_encode')
    return buf.value.decode()


def bech32m_decode(s: str):
    out = np.zeros(max(len(s), 1), dtype=np.uint8); n = ctypes.c_size_t(out.shape[0])
    hrp = ctypes.create_string_buffer(len(s) + 1)
    check(lib().aleo_mi355x_bech32m_decode(_p(out), ctypes.byref(n), hrp, len(s) + 1, s.encode()), 'bech32m_decode')
    return hrp.value.decode(), out[:n.value].tobytes()


class _ProofParts(ctypes.Structure):
    _fields_ = [('batch_sizes', ctypes.c_void_p), ('n_circuits', ctypes.c_size_t), ('witness_commitments', ctypes.c_void_p),
                ('mask_poly', ctypes.c_void_p), ('g_1', ctypes.c_void_p), ('h_1', ctypes.c_void_p), ('g_abc', ctypes.c_void_p),
                ('h_2', ctypes.c_void_p), ('evaluations', ctypes.c_void_p), ('n_evaluations', ctypes.c_size_t), ('sums', ctypes.c_void_p),
                ('opening_points', ctypes.c_void_p), ('opening_random_v', ctypes.c_void_p), ('opening_has_v', ctypes.c_void_p),
                ('n_openings', ctypes.c_size_t)]


def proof_to_bytes(batch_sizes, witness_commitments, mask_poly, g_1, h_1, g_abc, h_2, evaluations, sums, opening_points, opening_random_v) -> bytes:
    """Varuna proof bytes from its parts: commitments / opening points as snarkVM Affine rows (uint8[*,104]), field elements
    Montgomery uint64[*,4]; mask_poly None for a non-hiding proof; opening_random_v: one entry per opening, None = absent."""
    keep = []

    def arr(a, dt, w):
        a = np.ascontiguousarray(a, dtype=dt).reshape(-1, w); keep.append(a); return a
    bs = np.ascontiguousarray(batch_sizes, dtype=np.uint64); keep.append(bs)
    wc = arr(witness_commitments, np.uint8, 104); ev = arr(evaluations, np.uint64, 4); sm = arr(sums, np.uint64, 4)
    op = arr(opening_points, np.uint8, 104)
    has = np.array([0 if v is None else 1 for v in opening_random_v], dtype=np.uint8); keep.append(has)
    rv = arr([np.zeros(4, dtype=np.uint64) if v is None else np.asarray(v, dtype=np.uint64).reshape(4) for v in opening_random_v] or np.zeros((1, 4)), np.uint64, 4)
    P = _ProofParts()
    P.batch_sizes = bs.ctypes.data; P.n_circuits = bs.shape[0]; P.witness_commitments = wc.ctypes.data
    P.mask_poly = None if mask_poly is None else arr(mask_poly, np.uint8, 104).ctypes.data
    P.g_1 = arr(g_1, np.uint8, 104).ctypes.data; P.h_1 = arr(h_1, np.uint8, 104).ctypes.data
    P.g_abc = arr(g_abc, np.uint8, 104).ctypes.data; P.h_2 = arr(h_2, np.uint8, 104).ctypes.data
    P.evaluations = ev.ctypes.data; P.n_evaluations = ev.shape[0]; P.sums = sm.ctypes.data
    P.opening_points = op.ctypes.data; P.opening_random_v = rv.ctypes.data; P.opening_has_v = has.ctypes.data; P.n_openings = op.shape[0]
    out = np.zeros(64 + 48 * (wc.shape[0] + op.shape[0] + 16) + 32 * (ev.shape[0] + sm.shape[0] + op.shape[0] + 8), dtype=np.uint8)
    n = ctypes.c_size_t(out.shape[0])
    check(lib().aleo_mi355x_proof_to_bytes(_p(out), ctypes.byref(n), ctypes.byref(P)), 'proof_to_bytes')
    return out[:n.value].tobytes()


def proof_to_string(proof_bytes: bytes) -> str:
    return bech32m_encode('proof', proof_bytes)

"""Deterministic synthetic inputs shared by the tests, smoke() and bench.py.

Uses the oracle only to *generate bases* (consecutive multiples of the G1 generator) and to convert to Montgomery
form — test-side code; nothing here is imported by the product package."""
from __future__ import annotations
import numpy as np
from oracle import pyref, coracle

FR_MODULUS = pyref.FR_MODULUS
_M64 = (1 << 64) - 1


def splitmix_limbs(seed: int, count: int) -> np.ndarray:
    """count x uint64 from SplitMix64(seed), vectorised."""
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = np.uint64(seed & _M64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


_R_LIMBS = np.array([(FR_MODULUS >> (64 * i)) & _M64 for i in range(4)], dtype=np.uint64)


def _lt_r(a: np.ndarray) -> np.ndarray:
    lt = np.zeros(a.shape[0], dtype=bool); eq = np.ones(a.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        lt |= eq & (a[:, i] < _R_LIMBS[i]); eq &= a[:, i] == _R_LIMBS[i]
    return lt


def uniform_scalars(n: int, seed: int) -> np.ndarray:
    """n canonical Fr scalars uint64[n,4]: 253-bit rejection sampling (values >= r are re-drawn)."""
    out = np.zeros((n, 4), dtype=np.uint64); todo = np.arange(n); rnd = 0
    while todo.size:
        a = splitmix_limbs(seed + 7919 * rnd, 4 * todo.size).reshape(-1, 4).copy()
        a[:, 3] &= np.uint64((1 << 61) - 1)
        ok = _lt_r(a)
        out[todo[ok]] = a[ok]; todo = todo[~ok]; rnd += 1
    return out


def witness_like_scalars(n: int, seed: int) -> np.ndarray:
    """60% zero, 20% one, 10% < 2^16, 10% uniform."""
    s = uniform_scalars(n, seed)
    sel = splitmix_limbs(seed ^ 0x5151, n) % np.uint64(10)
    z = sel < 6; o = (sel >= 6) & (sel < 8); sm = sel == 8
    s[z] = 0
    s[o] = 0; s[o, 0] = 1
    s[sm, 1:] = 0; s[sm, 0] &= np.uint64(0xFFFF)
    return s


def generator_affine() -> np.ndarray:
    return coracle.affine_from_ints([pyref.G1_GENERATOR])[0]


def multiples_bases(n: int) -> np.ndarray:
    """P_i = (i+1) * G as uint8[n,104] (cheap to generate)."""
    return coracle.g1_multiples(generator_affine(), n)


def scalars_to_ints(s: np.ndarray) -> list:
    return coracle.limbs_to_ints(s)


def expected_multiples_msm(scalars: np.ndarray, n: int):
    """For bases (i+1)G the MSM is (sum_i s_i (i+1) mod r) * G — an O(n) oracle that scales to any n."""
    k = 0
    for i, v in enumerate(scalars_to_ints(scalars[:n])):
        k = (k + v * (i + 1)) % FR_MODULUS
    return pyref.g1_mul(pyref.G1_GENERATOR, k)

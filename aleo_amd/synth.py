"""Deterministic synthetic inputs for the MSM / NTT workloads: SplitMix64-seeded scalars.
No dependency on oracle/: bench.py and the product-side smoke path use this; tests/util.py re-exports it."""
from __future__ import annotations
import numpy as np

FR_MODULUS = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
FQ_MODULUS = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
FQ_R = 1 << 384
G1_GENERATOR = (
    89363714989903307245735717098563574705733591463163614225748337416674727625843187853442697973404985688481508350822,
    3702177272937190650578065972808860481433820514072818216637796320125658674906330993856598323293086021583822603349,
)
_M64 = (1 << 64) - 1
_R_LIMBS = np.array([(FR_MODULUS >> (64 * i)) & _M64 for i in range(4)], dtype=np.uint64)


def splitmix_limbs(seed: int, count: int) -> np.ndarray:
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = np.uint64(seed & _M64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _lt_r(a: np.ndarray) -> np.ndarray:
    lt = np.zeros(a.shape[0], dtype=bool); eq = np.ones(a.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        lt |= eq & (a[:, i] < _R_LIMBS[i]); eq &= a[:, i] == _R_LIMBS[i]
    return lt


def uniform_scalars(n: int, seed: int) -> np.ndarray:
    """n canonical Fr values uint64[n,4]: 253-bit rejection sampling."""
    out = np.zeros((n, 4), dtype=np.uint64); todo = np.arange(n); rnd = 0
    while todo.size:
        a = splitmix_limbs(seed + 7919 * rnd, 4 * todo.size).reshape(-1, 4).copy()
        a[:, 3] &= np.uint64((1 << 61) - 1)
        ok = _lt_r(a)
        out[todo[ok]] = a[ok]; todo = todo[~ok]; rnd += 1
    return out


def witness_like_scalars(n: int, seed: int) -> np.ndarray:
    """60% zero, 20% one, 10% < 2^16, 10% uniform — the skew of real R1CS witnesses."""
    s = uniform_scalars(n, seed)
    sel = splitmix_limbs(seed ^ 0x5151, n) % np.uint64(10)
    z = sel < 6; o = (sel >= 6) & (sel < 8); sm = sel == 8
    s[z] = 0
    s[o] = 0; s[o, 0] = 1
    s[sm, 1:] = 0; s[sm, 0] &= np.uint64(0xFFFF)
    return s


def int_to_limbs(v: int, nl: int) -> np.ndarray:
    return np.array([(v >> (64 * i)) & _M64 for i in range(nl)], dtype=np.uint64)


def limbs_to_int(a) -> int:
    return sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(a, dtype=np.uint64).reshape(-1)))


def generator_affine104() -> np.ndarray:
    """The G1 generator as a snarkVM Affine (Montgomery x, y + infinity byte)."""
    out = np.zeros(104, dtype=np.uint8)
    xm = (G1_GENERATOR[0] * FQ_R) % FQ_MODULUS; ym = (G1_GENERATOR[1] * FQ_R) % FQ_MODULUS
    out[0:48] = int_to_limbs(xm, 6).view(np.uint8); out[48:96] = int_to_limbs(ym, 6).view(np.uint8)
    return out


def weighted_scalar_sum(scalars: np.ndarray, first_multiple: int = 1) -> int:
    """sum_i s_i * (first_multiple + i) mod r — the discrete log (base G) of an MSM over bases (first+i)*G.
    Exact integer arithmetic in numpy: 16-bit pieces of the scalars times weights below 2^31, 2^15 terms per dot product."""
    s = np.ascontiguousarray(np.asarray(scalars, dtype=np.uint64).reshape(-1, 4))
    n = s.shape[0]
    if n == 0: return 0
    if first_multiple + n >= (1 << 31):      # weights too wide for the fast path
        w = np.arange(first_multiple, first_multiple + n, dtype=object); total = 0
        for limb in range(4): total += int((s[:, limb].astype(object) * w).sum()) << (64 * limb)
        return total % FR_MODULUS
    q = s.view(np.uint16).reshape(n, 16).astype(np.uint64)           # little-endian 16-bit pieces
    w = np.arange(first_multiple, first_multiple + n, dtype=np.uint64)
    total = 0
    for lo in range(0, n, 1 << 15):                                    # 2^16 * 2^31 * 2^15 = 2^62 < 2^64
        part = w[lo:lo + (1 << 15)] @ q[lo:lo + (1 << 15)]             # uint64[16], exact
        total += sum(int(v) << (16 * j) for j, v in enumerate(part))
    return total % FR_MODULUS

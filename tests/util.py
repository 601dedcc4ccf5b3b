"""Deterministic synthetic inputs shared by the tests, smoke() and bench.py.

Uses the oracle only to *generate bases* (consecutive multiples of the G1 generator) and to convert to Montgomery
form — test-side code; nothing here is imported by the product package."""
from __future__ import annotations
import numpy as np
from oracle import pyref, coracle

from aleo_amd.synth import splitmix_limbs, uniform_scalars, witness_like_scalars, weighted_scalar_sum, FR_MODULUS  # noqa: F401


def generator_affine() -> np.ndarray:
    return coracle.affine_from_ints([pyref.G1_GENERATOR])[0]


def multiples_bases(n: int) -> np.ndarray:
    """P_i = (i+1) * G as uint8[n,104] (cheap to generate)."""
    return coracle.g1_multiples(generator_affine(), n)


def scalars_to_ints(s: np.ndarray) -> list:
    return coracle.limbs_to_ints(s)


def expected_multiples_msm(scalars: np.ndarray, n: int):
    """For bases (i+1)G the MSM is (sum_i s_i (i+1) mod r) * G — an O(n) oracle that scales to any n."""
    return pyref.g1_mul(pyref.G1_GENERATOR, weighted_scalar_sum(scalars[:n], 1))


class OracleLocalOps:
    """Local steps of aleo_amd.dist.ShardedDomain on CPU tensors through the oracle (tests only): lets the exchange and
    layout logic run under gloo without a GPU.  Same interface as aleo_amd.dist.HipLocalOps."""

    def batch_ntt(self, t, lg_len, batch, direction):
        from oracle import coracle as c
        a = t.numpy().view(np.uint64).reshape(batch, 1 << lg_len, 4)
        for b in range(batch):
            a[b] = c.ntt_fr(a[b], 0, direction, 0)

    def grid_scale(self, t, lg_n, rows, cols, row0, col0, ld, mode, direction):
        from oracle import coracle as c, pyref as p
        a = t.numpy().view(np.uint64).reshape(rows * cols, 4)
        base = pow(p.FR_TWO_ADIC_ROOT, 1 << (p.FR_TWO_ADICITY - lg_n), p.FR_MODULUS) if mode == 0 else p.FR_GENERATOR
        if direction == 1: base = pow(base, -1, p.FR_MODULUS)
        vals = c.limbs_to_ints(c.fr_from_mont(a))
        out = []
        for i, v in enumerate(vals):
            r, cc = row0 + i // cols, col0 + i % cols
            e = r * cc if mode == 0 else r * ld + cc
            out.append(v * pow(base, e, p.FR_MODULUS) % p.FR_MODULUS)
        a[:] = c.fr_to_mont(c.ints_to_limbs(out, 4))

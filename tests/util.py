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

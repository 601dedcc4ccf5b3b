"""Host arithmetic of the product library that needs no device: the inversion of the MSM host tails and the prover's round constants (csrc/host_modinv.hpp,
Bernstein-Yang divsteps) against the Fermat chain it replaced, through the C ABI's self-test."""
import ctypes
import aleo_amd


def test_divsteps_inverse_equals_the_fermat_chain():
    L = aleo_amd.lib()
    bad = ctypes.c_uint32(1); ns = (ctypes.c_double * 4)()
    for seed in (1, 0xA1E00005):
        assert L.aleo_mi355x_selftest_host_inverse(5000, seed, ctypes.byref(bad), ns) == 0
        assert bad.value == 0                                      # 5005 elements of Fq and of Fr each: same bytes as a^(p-2), and a * a^-1 = 1
    assert ns[0] < ns[1] and ns[2] < ns[3], list(ns)               # and it is the faster one (Fq: ~3 us against ~30 us on a build-container core)

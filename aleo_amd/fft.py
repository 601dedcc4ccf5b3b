"""Host-side mirror of snarkvm_algorithms::fft::EvaluationDomain<Fr> for the MI355X backend.

Reference interface (snarkVM 0.14.5, algorithms/src/fft/domain.rs [UPSTREAM-RECALL]; used by Varuna's prover behind
/root/reference/rust/src/program/execute.rs:74):  EvaluationDomain::new(num_coeffs) -> Option<Self>;
fft / ifft / coset_fft / coset_ifft (and *_in_place) on slices of Fr in Montgomery form; `fft` zero-pads to `size`.
Arrays are uint64[n,4] (Montgomery limbs, as snarkVM stores Fr)."""
from __future__ import annotations
import ctypes
import numpy as np
from ._lib import lib, check

ORDER_NN, ORDER_NR, ORDER_RN, ORDER_RR = 0, 1, 2, 3
FORWARD, INVERSE = 0, 1
STANDARD, COSET = 0, 1
FR_TWO_ADICITY = 47


def _p(a): return a.ctypes.data_as(ctypes.c_void_p)


class EvaluationDomain:
    def __init__(self, num_coeffs: int):
        size = 1
        while size < num_coeffs: size *= 2
        lg = size.bit_length() - 1
        if lg > FR_TWO_ADICITY:           # reference: EvaluationDomain::new returns None
            raise ValueError('domain larger than the two-adicity of Fr')
        self.size, self.log_size_of_group = size, lg

    @classmethod
    def new(cls, num_coeffs: int):
        try: return cls(num_coeffs)
        except ValueError: return None

    def _run(self, x: np.ndarray, direction: int, type_: int, order: int = ORDER_NN) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)
        if x.shape[0] > self.size: raise ValueError('more coefficients than the domain size')
        buf = np.zeros((self.size, 4), dtype=np.uint64); buf[:x.shape[0]] = x     # resize(size, zero) as the reference
        check(lib().aleo_mi355x_ntt_fr(_p(buf), self.log_size_of_group, order, direction, type_), 'ntt_fr')
        return buf

    def fft(self, coeffs): return self._run(coeffs, FORWARD, STANDARD)
    def ifft(self, evals): return self._run(evals, INVERSE, STANDARD)
    def coset_fft(self, coeffs): return self._run(coeffs, FORWARD, COSET)
    def coset_ifft(self, evals): return self._run(evals, INVERSE, COSET)

    def _run_in_place(self, x: np.ndarray, direction: int, type_: int):
        """The reference's *_in_place shape: the caller's own buffer (full domain size, contiguous uint64[n,4]) goes to the C ABI as it
        is — one upload, the transform, one download into the same memory, no intermediate copy on the host."""
        if not (isinstance(x, np.ndarray) and x.dtype == np.uint64 and x.flags.c_contiguous and x.size == 4 * self.size):
            x[...] = self._run(x, direction, type_); return
        check(lib().aleo_mi355x_ntt_fr(_p(x), self.log_size_of_group, ORDER_NN, direction, type_), 'ntt_fr')

    def fft_in_place(self, x: np.ndarray): self._run_in_place(x, FORWARD, STANDARD)
    def ifft_in_place(self, x: np.ndarray): self._run_in_place(x, INVERSE, STANDARD)
    def coset_fft_in_place(self, x: np.ndarray): self._run_in_place(x, FORWARD, COSET)
    def coset_ifft_in_place(self, x: np.ndarray): self._run_in_place(x, INVERSE, COSET)

    def ntt_sharded_in_place(self, x: np.ndarray, devices, direction=FORWARD, type_=STANDARD):
        """The same in-place transform split over several devices of this process (aleo_mi355x_ntt_fr_sharded: 4-step, every device uploads its
        columns, one peer exchange, natural order out).  devices: a list of HIP device indices (an index may repeat) or a count; a power of two."""
        if not (isinstance(x, np.ndarray) and x.dtype == np.uint64 and x.flags.c_contiguous and x.size == 4 * self.size): raise ValueError('a contiguous uint64[size, 4] buffer')
        if isinstance(devices, int): dv, g = None, devices
        else: dv, g = (ctypes.c_int32 * len(devices))(*[int(d) for d in devices]), len(devices)
        check(lib().aleo_mi355x_ntt_fr_sharded(_p(x), self.log_size_of_group, direction, type_, dv, g), 'ntt_fr_sharded')

    def ntt(self, x, order=ORDER_NN, direction=FORWARD, type_=STANDARD):
        """The snarkvm_algorithms_cuda::NTT shape: explicit order / direction / type."""
        return self._run(x, direction, type_, order)

    def ntt_device(self, d_ptr: int, order=ORDER_NN, direction=FORWARD, type_=STANDARD, stream: int = 0):
        check(lib().aleo_mi355x_ntt_fr_device(ctypes.c_void_p(d_ptr), self.log_size_of_group, order, direction, type_,
                                              ctypes.c_void_p(stream)), 'ntt_fr_device')

    def ntt_batch_device(self, d_ptr: int, batch: int, order=ORDER_NN, direction=FORWARD, type_=STANDARD, stream: int = 0):
        """`batch` contiguous transforms of this domain's size in one call."""
        check(lib().aleo_mi355x_ntt_fr_batch_device(ctypes.c_void_p(d_ptr), self.log_size_of_group, batch, order, direction, type_,
                                                    ctypes.c_void_p(stream)), 'ntt_fr_batch_device')

    def ntt_from_device(self, d_out: int, d_src: int, src_stride: int, src_len: int, batch: int = 1, direction=FORWARD, type_=STANDARD, stream: int = 0):
        """Out of place with a zero-padded input (EvaluationDomain::fft of polynomials with fewer coefficients than the domain): transform b reads
        src_len elements at d_src + b * src_stride * 32 and writes this domain's size to d_out + b * size * 32."""
        check(lib().aleo_mi355x_ntt_fr_from_device(ctypes.c_void_p(d_out), ctypes.c_void_p(d_src), src_stride, src_len, self.log_size_of_group, batch, direction, type_,
                                                   ctypes.c_void_p(stream)), 'ntt_fr_from_device')

    def ntt_sharded_device(self, d_ptr: int, devices, direction=FORWARD, type_=STANDARD, stream: int = 0):
        """In place on device-resident data (this domain's size at d_ptr on the current device), computed over several devices by peer copies — no host buffer
        (aleo_mi355x_ntt_fr_sharded_device).  devices: a list of HIP device indices (an index may repeat) or a count; a power of two.  Blocking."""
        if isinstance(devices, int): dv, g = None, devices
        else: dv, g = (ctypes.c_int32 * len(devices))(*[int(d) for d in devices]), len(devices)
        check(lib().aleo_mi355x_ntt_fr_sharded_device(ctypes.c_void_p(d_ptr), self.log_size_of_group, direction, type_, dv, g, ctypes.c_void_p(stream)), 'ntt_fr_sharded_device')

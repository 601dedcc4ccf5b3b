"""Achieved algorithmic bandwidth of the AHP-round kernels at sizes that fill the chip (not a test): ms and GB/s per kernel."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, poly

torch.cuda.set_device(0)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 23
n = 1 << lg
def dev(seed, m=n): return torch.from_numpy(synth.uniform_scalars(m, seed).view(np.int64)).cuda()
A, B, C, D, E = (dev(i) for i in range(5))
out = torch.zeros((n, 4), dtype=torch.int64, device='cuda')
k = synth.uniform_scalars(8, 99)
idx1 = torch.from_numpy((synth.splitmix_limbs(7, n) % np.uint64(1 << 15)).astype(np.uint32).view(np.int32)).cuda()
idx2 = torch.from_numpy((synth.splitmix_limbs(8, n) % np.uint64(1 << 15)).astype(np.uint32).view(np.int32)).cuda()
T1, T2 = dev(20, 1 << 15), dev(21, 1 << 15)
IDX = torch.cat([dev(30 + i) for i in range(4)])          # row, col, val, row_col of one matrix
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
res = {}
def rec(name, ms, bytes_per_elem): res[name] = {'ms': ms, 'alg_GBps': bytes_per_elem * n / ms / 1e6}
s = torch.cuda.current_stream().cuda_stream or 0
rec('fr_lin (c0 + c1 a + c2 b)', timed(lambda: poly.fr_lin_device(out.data_ptr(), n, k[0], k[1], A.data_ptr(), k[2], B.data_ptr())), 96)
rec('fr_vec_op mul', timed(lambda: poly.fr_vec_op_device(out.data_ptr(), A.data_ptr(), B.data_ptr(), n, 0)), 96)
rec('fr_lincomb 5 terms', timed(lambda: poly.fr_lincomb_device(out.data_ptr(), n, k[0], [(t.data_ptr(), n, k[i + 1]) for i, t in enumerate((A, B, C, D, E))])), 192)
rec('ahp_first_sumcheck', timed(lambda: poly.ahp_first_sumcheck_device(out.data_ptr(), n, A.data_ptr(), B.data_ptr(), C.data_ptr(), D.data_ptr(), E.data_ptr(), k[0], k[1])), 192)
rec('ahp_matrix_sumcheck (1 matrix)', timed(lambda: poly.ahp_matrix_sumcheck_device(out.data_ptr(), n, [IDX.data_ptr(), 0, 0], n, [A.data_ptr(), 0, 0], k[:7])), 192)
rec('fr_gather_mul (2 tables of 2^15)', timed(lambda: poly.fr_gather_mul_device(out.data_ptr(), n, A.data_ptr(), T1.data_ptr(), idx1.data_ptr(), T2.data_ptr(), idx2.data_ptr())), 136)
rec('fr_powers', timed(lambda: poly.fr_powers_device(out.data_ptr(), n, k[0], k[1])), 32)
rec('fr_random', timed(lambda: poly.fr_random_device(out.data_ptr(), n, 12345, 0, True)), 32)
ev = torch.zeros((8, 4), dtype=torch.int64, device='cuda')
rec('fr_eval_batch (4 polynomials)', timed(lambda: poly.fr_eval_batch_device(ev.data_ptr(), [A.data_ptr(), B.data_ptr(), C.data_ptr(), D.data_ptr()], [n] * 4, k[:4])) / 4, 32)
q = torch.zeros((n, 4), dtype=torch.int64, device='cuda')
rec('fr_divide_by_linear', timed(lambda: poly.divide_by_linear_device(q.data_ptr(), ev.data_ptr(), A.data_ptr(), n, k[0])), 96)
print(json.dumps({'lg_n': lg, 'kernels': res}))

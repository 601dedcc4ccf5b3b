"""Where the host's time goes inside one aleo_amd.varuna.prove (not a test): wall time per wrapped call, summed over a proof."""
import os, sys, time, json, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from aleo_amd import synth, varuna, wire
from aleo_amd.kzg import SonicKZG10

TAU, S_GAMMA = 0x1F3A9C0D5E7B24681357ACE02468BDF013579BDF02468ACE1234567, 0x0FEDCBA9876543210123456789ABCDEF55AA
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 15
acc = collections.Counter(); cnt = collections.Counter()
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label or name] += time.perf_counter() - t; cnt[label or name] += 1; return r
    setattr(obj, name, g)
for n in ('fr_vec_op_device', 'fr_lin_device', 'fr_powers_device', 'fr_gather_mul_device', 'fr_eval_batch_device', 'fr_random_device', 'fr_lincomb_device',
          'ahp_first_sumcheck_device', 'ahp_matrix_sumcheck_device', 'spmv_device', 'divide_by_linear_device', '_mont', '_mont_rows', '_from_mont', 'random_fr', '_Vec'):
    wrap(varuna, n)
wrap(SonicKZG10, 'commit', 'SonicKZG10.commit'); wrap(wire, 'g1_compress'); wrap(wire, 'proof_to_bytes')
wrap(varuna.EvaluationDomain, 'ntt_device'); wrap(varuna.EvaluationDomain, 'ntt_batch_device')
wrap(torch.Tensor, 'copy_'); wrap(torch.Tensor, 'cpu'); wrap(torch.cuda.Stream, 'synchronize')
wrap(varuna.Transcript, 'absorb'); wrap(varuna.Transcript, 'challenge')
n = (1 << lg) - 64
csr, z = synth.synthetic_r1cs(n, 4, 40 + lg, long_rows=4)
zz = np.stack([synth.int_to_limbs(v, 4) for v in z])
nnz = max(int(csr[m][0][-1]) for m in 'abc'); n_k = 2
while n_k < nnz: n_k *= 2
D = 1
while D < max(3 << lg, n_k): D *= 2
ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D - 1)
ix = varuna.CircuitIndex(csr, n, 4, len(z) - 4, ck)
for rep in range(4): varuna.prove(ix, zz, rep)
acc.clear(); cnt.clear()
reps = 5; t = time.perf_counter()
for rep in range(reps): pr = varuna.prove(ix, zz, 100 + rep)
wall = (time.perf_counter() - t) / reps * 1e3
print('prove ms (instrumented)', wall, pr.timing_ms)
for k, v in acc.most_common(): print(f'{k:28s} {cnt[k] / reps:6.1f} calls  {v / reps * 1e3:8.3f} ms')
ck.close()

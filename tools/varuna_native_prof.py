"""Workload for rocprofv3: setup, then N proofs through aleo_mi355x_varuna_prove (nothing else on the GPU in the steady state)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from aleo_amd import synth, varuna
TAU, S_GAMMA = 0x1F3A9C0D5E7B24681357ACE02468BDF013579BDF02468ACE1234567, 0x0FEDCBA9876543210123456789ABCDEF55AA
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 15; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10; k = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = (1 << lg) - 64
csr, z = synth.synthetic_r1cs(n, 4, 40 + lg, long_rows=4)
zz = np.stack([synth.int_to_limbs(v, 4) for v in z])
nnz = max(int(csr[m][0][-1]) for m in 'abc'); n_k = 2
while n_k < nnz: n_k *= 2
D = 1
while D < max(3 << lg, n_k): D *= 2
ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D - 1)
ix = varuna.CircuitIndex(csr, n, 4, len(z) - 4, ck)
for rep in range(reps): varuna.prove_native(ix, [zz] * k, rep)
print(varuna.native_timing())
ck.close()

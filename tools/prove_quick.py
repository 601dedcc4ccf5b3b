"""Quick wall-time probe of the native prover (not a test): aleo_mi355x_varuna_prove at the given lg sizes (default 13 15), min / median of
`reps` proofs, the per-round split of the last one, and the host cost of one Poseidon permutation over Fq on this box."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, varuna

TAU, S_GAMMA = 0x1F3A9C0D5E7B24681357ACE02468BDF013579BDF02468ACE1234567, 0x0FEDCBA9876543210123456789ABCDEF55AA
torch.cuda.set_device(0)
fs = varuna.FiatShamir(); fs.absorb_fr([1, 2, 3]); fs.squeeze(4)
t = time.perf_counter(); fs.squeeze(1492); dt = time.perf_counter() - t          # 1492 * 252 bits = 1000 elements of 376 bits = 500 permutations
print(json.dumps({'poseidon_fq_rate2_us_per_permutation': dt / 500 * 1e6}), flush=True)
reps = int(os.environ.get('REPS', '15'))
for lg in [int(a) for a in sys.argv[1:]] or [13, 15]:
    n = (1 << lg) - 64
    csr, z = synth.synthetic_r1cs(n, 4, 40 + lg, long_rows=4)
    zz = np.stack([synth.int_to_limbs(v, 4) for v in z])
    nnz = max(int(csr[m][0][-1]) for m in 'abc'); n_k = 2
    while n_k < nnz: n_k *= 2
    D = 1
    while D < max(3 << lg, n_k): D *= 2
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D - 1)
    with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx:
        ts = []
        for rep in range(reps + 3):
            t = time.perf_counter(); nx.prove(zz, 1000 + rep); ts.append((time.perf_counter() - t) * 1e3)
        ts = ts[3:]
        t8 = []
        for rep in range(6):
            t = time.perf_counter(); nx.prove([zz] * 8, 3000 + rep); t8.append((time.perf_counter() - t) * 1e3)
        print(json.dumps({'lg_constraints': lg, 'constraints': n, 'prove_ms_min': min(ts), 'prove_ms_median': float(np.median(ts)), 'instances_8_ms': float(np.median(t8[1:])),
                          'rounds_ms': varuna.native_timing()}), flush=True)
    ck.close()

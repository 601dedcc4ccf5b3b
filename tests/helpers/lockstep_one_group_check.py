"""Run by tests/test_varuna.py in a child process with ALEO_MI355X_LOCKSTEP_GROUPS=1 (read once per process): aleo_mi355x_varuna_prove_many as ONE lockstep group — every
round's commitments of all proofs in one launch chain, the proofs dealt to worker threads on borrowed contexts — which the default (up to four groups, one thread each, one
proof per group up to four proofs) no longer takes for small calls.  Frozen cases of tests/golden/varuna_small.json: 2, 3, 5 and 8 proofs per call under different seeds
must equal the single-proof entry point byte for byte, and the frozen seed must give the frozen proof.  Prints ONE GROUP OK."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import aleo_amd
from aleo_amd import synth, varuna
import test_varuna as T
assert os.environ.get('ALEO_MI355X_LOCKSTEP_GROUPS') == '1'
tau, sg, cases = T._golden_cases()
lim = lambda q: np.stack([synth.int_to_limbs(v, 4) for v in q])
for case in cases[:3]:
    csr, zs, c = T._golden_instance(case)
    ck = varuna.synthetic_committer_key(tau, sg, case['max_degree'])
    try:
        with varuna.NativeCircuitIndex(csr, case['n_constraints'], case['n_public'], len(zs[0]) - case['n_public'], ck, domains=case['domains']) as nx:
            za = [lim(q) for q in zs]
            single = {sd: nx.prove(za, sd) for sd in range(40, 48)}
            assert nx.prove(za, case['proof_seed']).hex() == case['proof']
            for P in (2, 3, 5, 8):
                reqs = [([nx], [za], 40 + q) for q in range(P)]
                assert varuna.prove_many_native(reqs) == [single[40 + q] for q in range(P)], (case['n_constraints'], P)
            assert varuna.prove_many_native([([nx], [za], case['proof_seed'])] * 4) == [bytes.fromhex(case['proof'])] * 4
    finally: ck.close()
print('ONE GROUP OK')

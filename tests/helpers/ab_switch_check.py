"""Run by tests/test_varuna.py in a child process with the A/B switches flipped (each is read once per process: ALEO_MI355X_QUAD_ADD=0, ALEO_MI355X_NTT_WIDE_LG=0,
and the round-3 ones — SUM_TREE=0, ASIDE=0, NTT29=0, CHAIN_OVERLAP=0, CHUNK_FORM=1, LOCKSTEP_WORKERS=1): the
frozen proofs of tests/golden/varuna_small.json — single circuits and batches — must come out of the native prover with the lane-pair additions and
the register-group transform tiles as well; plus one table-path MSM and one 2^14 transform against the oracle.  Prints SWITCHES OK."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import aleo_amd
from aleo_amd import synth, varuna, msm as M
from oracle import coracle as co
import test_varuna as T
assert os.environ.get('ALEO_MI355X_QUAD_ADD') == '0' and os.environ.get('ALEO_MI355X_NTT_WIDE_LG') == '0'
tau, sg, cases = T._golden_cases(); _, batches = T._golden_batches()
lim = lambda q: np.stack([synth.int_to_limbs(v, 4) for v in q])
for case in cases:
    csr, zs, c = T._golden_instance(case)
    ck = varuna.synthetic_committer_key(tau, sg, case['max_degree'])
    try:
        with varuna.NativeCircuitIndex(csr, case['n_constraints'], case['n_public'], len(zs[0]) - case['n_public'], ck, domains=case['domains']) as nx:
            assert nx.vk_bytes.hex() == case['vk'] and nx.prove([lim(q) for q in zs], case['proof_seed']).hex() == case['proof']
    finally: ck.close()
for b in batches:
    ck = varuna.synthetic_committer_key(tau, sg, b['max_degree']); nx, za = [], []
    try:
        for j in b['members']:
            case = cases[j]; csr, zs, c = T._golden_instance(case)
            nx.append(varuna.NativeCircuitIndex(csr, case['n_constraints'], case['n_public'], len(zs[0]) - case['n_public'], ck, domains=case['domains'])); za.append([lim(q) for q in zs])
        assert varuna.prove_batch_native(nx, za, b['proof_seed']).hex() == b['proof']
    finally:
        for x in nx: x.close()
        ck.close()
co.lib()
n = 5000
with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as pb:
    pb.precompute()
    s = synth.uniform_scalars(n, 4711)
    got = M.VariableBase.msm(pb, s)
    k = synth.weighted_scalar_sum(s, 1)
    kG = M.VariableBase.msm(synth.generator_affine104().reshape(1, 104), synth.int_to_limbs(k, 4).reshape(1, 4))
    assert (np.asarray(got) == np.asarray(kG)).all()
n = 1 << 16                                                                   # witness-like scalars: a super-heavy bucket (the slice tree beside / inside the reduction), the chunk weights in either form
with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as pb:
    pb.precompute()
    s = synth.witness_like_scalars(n, 4712)
    kG = M.VariableBase.msm(synth.generator_affine104().reshape(1, 104), synth.int_to_limbs(synth.weighted_scalar_sum(s, 1), 4).reshape(1, 4))
    assert (np.asarray(M.VariableBase.msm(pb, s)) == np.asarray(kG)).all()
x = co.fr_to_mont(synth.uniform_scalars(1 << 19, 98))                         # a large-tile transform (8 x 32-bit or 9 x 29-bit limbs)
d = aleo_amd.EvaluationDomain(1 << 19)
assert (d.coset_fft(x) == co.ntt_fr(x, 0, 0, 1, threads=8)).all()
x = co.fr_to_mont(synth.uniform_scalars(1 << 14, 99))
d = aleo_amd.EvaluationDomain(1 << 14)
assert (d.fft(x) == co.ntt_fr(x, 0, 0, 0)).all() and (d.ifft(d.fft(x)) == x).all()
print('SWITCHES OK')

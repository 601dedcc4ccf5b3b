"""First on-GPU bring-up check (not a test): field products, NTT and MSM against the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np
import aleo_amd
from aleo_amd import msm as M
from oracle import coracle as c, pyref as p
import util

L = aleo_amd.lib()
print(L.aleo_mi355x_version(), 'init', L.aleo_mi355x_init_device(-1), L.aleo_mi355x_last_error())
n = 5000
a = util.uniform_scalars(n, 1); b = util.uniform_scalars(n, 2)
r = M.fr_mul(a, b); e = np.zeros_like(a); c.lib().oracle_fr_mul(c._p(e), c._p(a), c._p(b), n)
print('fr_mul ok', bool((r == e).all()))
qa = np.concatenate([a, util.uniform_scalars(n, 3)[:, :2]], axis=1); qb = np.concatenate([b, util.uniform_scalars(n, 4)[:, :2]], axis=1)
qa[:, 5] &= np.uint64((1 << 56) - 1); qb[:, 5] &= np.uint64((1 << 56) - 1)   # < q
r = M.fq_mul(qa, qb); e = np.zeros_like(qa); c.lib().oracle_fq_mul(c._p(e), c._p(qa), c._p(qb), n)
print('fq_mul ok', bool((r == e).all()))
if not (r == e).all():
    bad = np.where((r != e).any(axis=1))[0][:3]; print(bad, r[bad], e[bad])
for lg in (1, 2, 5, 10, 11, 13, 18, 19, 20):
    x = c.fr_to_mont(util.uniform_scalars(1 << lg, 10 + lg))
    d = aleo_amd.EvaluationDomain(1 << lg)
    for (direction, type_) in ((0, 0), (1, 0), (0, 1), (1, 1)):
        t = time.time(); got = d.ntt(x, 0, direction, type_); dt = time.time() - t
        exp = c.ntt_fr(x, 0, direction, type_)
        print('ntt lg', lg, 'dir', direction, 'coset', type_, 'ok', bool((got == exp).all()), '%.3fs' % dt)
for n in (1, 2, 5, 33, 100, 1000, 5000, 1 << 14, 1 << 16):
    B = util.multiples_bases(n)
    for name, s in (('uniform', util.uniform_scalars(n, 100 + n)), ('witness', util.witness_like_scalars(n, 200 + n))):
        t = time.time(); got = M.VariableBase.msm(B, s); dt = time.time() - t
        gp = c.jac_to_int_point(got)
        exp = util.expected_multiples_msm(s, n)
        print('msm n', n, name, 'ok', gp == exp, '%.3fs' % dt, M.last_msm_timing())
n = 1 << 20
B = util.multiples_bases(n); s = util.uniform_scalars(n, 999)
with M.PinnedBases(B) as pb:
    for it in range(3):
        t = time.time(); got = M.VariableBase.msm(pb, s); dt = time.time() - t
        print('msm 2^20 pinned', '%.4fs' % dt, M.last_msm_timing())
print('2^20 ok', c.jac_to_int_point(got) == util.expected_multiples_msm(s, n))

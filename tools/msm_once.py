"""Runs a few 2^lg MSMs (fixed-base table path by default) — a small target for rocprofv3 kernel traces."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, msm as M
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pre = (sys.argv[2] != 'plain') if len(sys.argv) > 2 else True
n = 1 << lg
aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_init_device(0), 'init')
pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n)
if pre: pb.precompute()
seed = int(sys.argv[3], 0) if len(sys.argv) > 3 else 5
mk = synth.witness_like_scalars if 'witness' in sys.argv else synth.uniform_scalars
s = torch.from_numpy(mk(n, seed).view(np.int64)).cuda(); torch.cuda.synchronize()
for _ in range(6): M.VariableBase.msm_device(pb, s.data_ptr(), n)
print(M.last_msm_timing())
if len(sys.argv) > 4 and sys.argv[4] == 'check':
    sc = mk(n, seed)
    got = M.VariableBase.msm_device(pb, s.data_ptr(), n)
    k = synth.weighted_scalar_sum(sc, 1)
    kG = M.VariableBase.msm(synth.generator_affine104().reshape(1, 104), synth.int_to_limbs(k, 4).reshape(1, 4))
    print('structured identity ok:', bool((got == kG).all()), 'Mpts/s', n / M.last_msm_timing()['total_ms'] / 1e3)

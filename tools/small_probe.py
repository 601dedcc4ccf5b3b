"""Prefix MSMs of every size against ONE pinned 2^20-point set (the KZG10::commit pattern): plain path vs tiered tables."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, msm as M
torch.cuda.set_device(0)
aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_init_device(0), 'init')
N = 1 << 20
S = synth.uniform_scalars(N, 71)
dS = torch.from_numpy(S.view(np.int64)).cuda(); torch.cuda.synchronize()
res = {}
for tag in ('plain', 'table'):
    pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, N)
    if tag == 'table':
        t0 = time.perf_counter(); pb.precompute(); res['precompute_s'] = time.perf_counter() - t0
    for lg in range(10, 21):
        n = 1 << lg
        r = M.VariableBase.msm_device(pb, dS.data_ptr(), n)
        k = synth.weighted_scalar_sum(S[:n], 1)
        kG = M.VariableBase.msm(synth.generator_affine104().reshape(1, 104), synth.int_to_limbs(k, 4).reshape(1, 4))
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); M.VariableBase.msm_device(pb, dS.data_ptr(), n); ts.append(time.perf_counter() - t0)
        res.setdefault(lg, {})[tag + '_ms'] = round(float(np.median(ts)) * 1e3, 3); res[lg][tag + '_ok'] = bool((r == kG).all())
    pb.close()
print(json.dumps(res))

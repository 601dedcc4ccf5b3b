"""Host-scalar MSMs against a pinned set with tables (the headline call: aleo_mi355x_msm_g1_pinned), for the sizes given: wall ms per call, the
library's phase record, and the result checked against k G in big integers (bench.py's gate).  The merged-halves path is switched by the environment
(ALEO_MI355X_MERGE_MIN_LG: 0 = off) — run the script once per setting for an A/B.  Not a test."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, msm as M
import bench
for lg in [int(a) for a in sys.argv[1:]] or [18, 19, 20, 21, 22]:
    n = 1 << lg
    pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n).precompute()
    for kind, mk in (('uniform', synth.uniform_scalars), ('witness', synth.witness_like_scalars)):
        sc = mk(n, 0xA1E00002)
        for _ in range(3): res = M.VariableBase.msm(pb, sc)
        ok = bench.result_is_multiple_of_generator(synth, res, synth.weighted_scalar_sum(sc, 1))
        torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 10
        for _ in range(reps): M.VariableBase.msm(pb, sc)
        dt = (time.perf_counter() - t0) / reps * 1e3
        print(json.dumps({'lg_n': lg, 'scalars': kind, 'merge_min_lg': os.environ.get('ALEO_MI355X_MERGE_MIN_LG', 'default'), 'ms': round(dt, 4), 'ok': ok, 'phases': {k: round(v, 4) for k, v in M.last_msm_timing().items()}}), flush=True)
    pb.close()

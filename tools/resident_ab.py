"""Resident-scalar MSMs (aleo_mi355x_msm_g1_device against a pinned set with tables), wall ms per call over 10 calls, for the sizes given; the environment
selects the variant (ALEO_MI355X_CHUNK_DEV_MIN_LG: chunked launch chains for device scalars).  Result checked against k G in big integers.  Not a test."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, msm as M
import bench
for lg in [int(a) for a in sys.argv[1:]] or [20, 21, 22]:
    n = 1 << lg
    pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n).precompute()
    for kind, mk in (('uniform', synth.uniform_scalars), ('witness', synth.witness_like_scalars)):
        sc = mk(n, 0xA1E00002); d = torch.from_numpy(sc.view(np.int64)).cuda(); torch.cuda.synchronize()
        for _ in range(3): res = M.VariableBase.msm_device(pb, d.data_ptr(), n)
        ok = bench.result_is_multiple_of_generator(synth, res, synth.weighted_scalar_sum(sc, 1))
        t0 = time.perf_counter(); reps = 10
        for _ in range(reps): M.VariableBase.msm_device(pb, d.data_ptr(), n)
        dt = (time.perf_counter() - t0) / reps * 1e3
        print(json.dumps({'lg_n': lg, 'scalars': kind, 'chunk_dev_min_lg': os.environ.get('ALEO_MI355X_CHUNK_DEV_MIN_LG', 'off'), 'ms': round(dt, 4), 'ok': ok, 'phases': {k: round(v, 4) for k, v in M.last_msm_timing().items()}}), flush=True)
    pb.close()

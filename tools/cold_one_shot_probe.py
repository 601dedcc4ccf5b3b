"""The cold two-line drop-in (aleo_mi355x_msm_g1 on host bases + host scalars, nothing cached) at the sizes given: wall ms per call, the result checked against
(sum s_i (i + 1)) G in big integers, the same request with 96-byte rows and with two infinity bases.  ALEO_MI355X_COLD_POOL=0: the call's buffers as hipMalloc / hipFree
pairs (rounds 1-4) instead of the slot's grow-only buffers — run once per setting.  Not a test."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd, bench
from aleo_amd import synth, msm as M
for lg in [int(a) for a in sys.argv[1:]] or [12, 16, 18, 19, 20, 21, 22]:
    n = 1 << lg
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as pb: hb = pb.download()
    sc = synth.uniform_scalars(n, 0xA1E00042)
    for _ in range(2): res = M.VariableBase.msm(hb, sc)
    ok = bench.result_is_multiple_of_generator(synth, res, synth.weighted_scalar_sum(sc, 1))
    reps = 6; torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): M.VariableBase.msm(hb, sc)
    dt = (time.perf_counter() - t0) / reps * 1e3
    row = {'lg_n': lg, 'cold_pool': os.environ.get('ALEO_MI355X_COLD_POOL', 'default'), 'ms': round(dt, 3), 'ok': bool(ok)}
    hb96 = np.ascontiguousarray(hb[:, :96]); row['rows96_equal'] = bool((M.VariableBase.msm(hb96, sc) == res).all())
    hi = hb.copy(); hi[n - 5] = 0; hi[n - 5, 96] = 1; hi[n // 2 + 3] = 0; hi[n // 2 + 3, 96] = 1      # two infinity bases
    sc2 = sc.copy(); sc2[n - 5] = 0; sc2[n // 2 + 3] = 0
    row['infinity_bases_equal'] = bool((M.VariableBase.msm(hi, sc) == M.VariableBase.msm(hb, sc2)).all())
    row['after_infinity_equal'] = bool((M.VariableBase.msm(hb, sc) == res).all())                      # the slot's flags buffer is not sticky
    print(json.dumps(row), flush=True)

"""G2 MSMs (aleo_mi355x_msm_g2: host bases, host scalars, nothing resident) at the sizes given: wall ms, scalar-muls/s, the result against
(sum s_i w_i) G2 in Python integers.  ALEO_MI355X_G2_PAIR28=0 selects the round-2 kernel: run once per setting.  Not a test."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import aleo_amd
from aleo_amd import synth, msm as M
for lg in [int(a) for a in sys.argv[1:]] or [12, 16, 20]:
    n = 1 << lg
    B = synth.g2_multiples_affine200(n); S = synth.uniform_scalars(n, 0xA1E00077)
    res = M.msm_g2(B, S); ok = synth.g2_result_gate(res, S)
    reps = 3 if lg >= 18 else 6; t0 = time.perf_counter()
    for _ in range(reps): M.msm_g2(B, S)
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(json.dumps({'lg_n': lg, 'ms': round(ms, 3), 'scalar_muls_per_s': round(n / ms * 1e3), 'ok': ok, 'pair28': os.environ.get('ALEO_MI355X_G2_PAIR28', '1')}), flush=True)

"""A/B of one configuration of the prover (not a test): the 2^lg bit-heavy circuit with Lagrange powers and the range window, one proof and eight instances, rounds."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from aleo_amd import synth, varuna
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, csr, z, zz, ck, D = bench._varuna_instance(synth, lg, 40 + lg, True, True)
out = {'env': {k: v for k, v in os.environ.items() if k.startswith('ALEO_MI355X_')}, 'lg': lg}
with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx:
    for k in (1, 8):
        nx.prove([zz] * k, 1); ts = []
        for rep in range(3):
            t = time.perf_counter(); nx.prove([zz] * k, 10 + rep); ts.append((time.perf_counter() - t) * 1e3)
        out['k=%d' % k] = {'ms': float(np.median(ts)), 'rounds': {a: round(b, 2) for a, b in varuna.native_timing().items()}}
ck.close()
print(json.dumps(out), flush=True)

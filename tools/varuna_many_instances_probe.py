"""constraints/s of one proof carrying many instances of one 2^lg-constraint circuit (not a test): 8 (one key), 16 and 32 (the same key listed 2 and 4 times)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from aleo_amd import synth, varuna
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 15
n, csr, z, zz, ck, D = bench._varuna_instance(synth, lg, 40 + lg)
out = {'lg': lg}
with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx:
    for m in (1, 2, 4):
        keys, za = [nx] * m, [[zz] * 8] * m
        varuna.prove_batch_native(keys, za, 1); ts = []
        for rep in range(4):
            t = time.perf_counter(); data = varuna.prove_batch_native(keys, za, 10 + rep); ts.append((time.perf_counter() - t) * 1e3)
        ms = float(np.median(ts[1:]))
        out['%d instances' % (8 * m)] = {'ms': round(ms, 2), 'constraints_per_s': round(8 * m * n / ms * 1e3), 'proof_bytes': len(data)}
ck.close()
print(json.dumps(out), flush=True)

"""Aggregate proofs/s of single 2^lg-constraint proofs with T caller threads (not a test); ALEO_MI355X_SLOTS sets how many calls the library runs at once."""
import os, sys, json, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from aleo_amd import synth, varuna
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 15
n, csr, z, zz, ck, D = bench._varuna_instance(synth, lg, 40 + lg)
out = {'slots': os.environ.get('ALEO_MI355X_SLOTS', 'default (4)'), 'lg': lg}
with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx:
    nx.prove(zz, 1)
    for T in (1, 2, 4, 6, 8):
        per = 8
        def work(k):
            for rep in range(per): nx.prove(zz, 100 * k + rep)
        for _ in range(2):
            th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
            t = time.perf_counter()
            for x in th: x.start()
            for x in th: x.join()
            dt = time.perf_counter() - t
        out['threads_%d' % T] = {'proofs_per_s': round(T * per / dt, 1), 'constraints_per_s': round(n * T * per / dt)}
ck.close()
print(json.dumps(out), flush=True)

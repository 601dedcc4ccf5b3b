import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
from aleo_amd import synth, varuna
lg = int(sys.argv[1])
n, csr, z, zz, ck, D = bench._varuna_instance(synth, lg, 40 + lg)
for rep in range(3):
    t = time.perf_counter(); nx = varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck); dt = time.perf_counter() - t
    print('index_s', round(dt, 4), flush=True); nx.close()
ck.close()

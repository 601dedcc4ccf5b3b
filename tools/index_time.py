"""Key synthesis (aleo_mi355x_varuna_index_build) timed three times at 2^lg constraints (not a test): python tools/index_time.py <lg>; ALEO_MI355X_INDEX_TIMING=1 prints the phases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from aleo_amd import synth, varuna
lg = int(sys.argv[1])
n, csr, z, zz, ck, D = bench._varuna_instance(synth, lg, 40 + lg)
for rep in range(3):
    t = time.perf_counter(); nx = varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck); dt = time.perf_counter() - t
    print('index_s', round(dt, 4), flush=True); nx.close()
ck.close()

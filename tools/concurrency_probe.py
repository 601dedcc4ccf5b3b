"""Aggregate MSM throughput with T caller threads (not a test): the rayon threads of one prover round call the
library concurrently; each call takes its own slot (stream + workspaces), so small MSMs overlap on the GPU."""
import os, sys, time, json, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, msm as M

torch.cuda.set_device(0)
aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_init_device(0), 'init')
for lg in (14, 16, 18, 20):
    n = 1 << lg
    pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n).precompute()
    ds = [torch.from_numpy(synth.uniform_scalars(n, 40 + t).view(np.int64)).cuda() for t in range(8)]
    torch.cuda.synchronize()
    ref = [M.VariableBase.msm_device(pb, d.data_ptr(), n) for d in ds]
    for T in (1, 2, 4, 8):
        reps = 16
        ok = [True] * T
        def work(t):
            torch.cuda.set_device(0)
            for _ in range(reps):
                r = M.VariableBase.msm_device(pb, ds[t].data_ptr(), n)
                if not (r == ref[t]).all(): ok[t] = False
        for warm in range(2):
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            t0 = time.perf_counter()
            for x in th: x.start()
            for x in th: x.join()
            dt = time.perf_counter() - t0
        print(json.dumps({'lg': lg, 'threads': T, 'ms_per_msm_aggregate': dt / (T * reps) * 1e3, 'Mmuls_s': T * reps * n / dt / 1e6, 'all_equal': all(ok)}), flush=True)
    pb.close()

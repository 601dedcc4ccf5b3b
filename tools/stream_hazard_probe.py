"""One process that first touches EVERY context of the device — a lockstep call of 16 proofs (four groups and their workers), four threads of chunked host-scalar MSMs
(helpers' high-priority streams) — and then times what depends on how the streams landed on hardware queues: the lockstep call again, the chunked MSM, a 2^20-constraint
proof (pipelined chains).  Not a test; run once per ALEO_MI355X_STREAM_ORDER / ALEO_MI355X_HI_POOL / ALEO_MI355X_PIPELINE_HI setting."""
import os, sys, json, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from aleo_amd import synth, varuna, msm as M
out = {k: os.environ.get(k, 'default') for k in ('ALEO_MI355X_STREAM_ORDER', 'ALEO_MI355X_HI_POOL', 'ALEO_MI355X_PIPELINE_HI', 'ALEO_MI355X_HI_PRIORITY')}
n, csr, z, zz, ck, D = bench._varuna_instance(synth, 15, 55)
def med(fn, reps):
    fn(); ts = []
    for _ in range(reps): t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
    return round(float(np.median(ts)), 3)
with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx:
    reqs16 = [([nx], [[zz]], 100 + q) for q in range(16)]; reqs8 = reqs16[:8]
    varuna.prove_many_native(reqs16)
    N = 1 << 20
    pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, N).precompute()
    sc = synth.uniform_scalars(N, 0xA1E00002)
    def work():
        for _ in range(4): M.VariableBase.msm(pb, sc)
    th = [threading.Thread(target=work) for _ in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    out['lockstep_8_ms'] = med(lambda: varuna.prove_many_native(reqs8), 8)
    out['lockstep_16_ms'] = med(lambda: varuna.prove_many_native(reqs16), 6)
    out['single_2^15_ms'] = med(lambda: nx.prove(zz, 5), 12)
    out['host_msm_2^20_ms'] = med(lambda: M.VariableBase.msm(pb, sc), 10)
    t = time.perf_counter()
    for x in (threading.Thread(target=work) for _ in range(4)): x.start(); th.append(x)
    for x in th[4:]: x.join()
    out['four_callers_ms_per_msm'] = round((time.perf_counter() - t) * 1e3 / 16, 3)
    pb.close()
ck.close()
n, csr, z, zz, ck, D = bench._varuna_instance(synth, 20, 60)
with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx:
    out['single_2^20_ms'] = med(lambda: nx.prove(zz, 7), 3)
ck.close()
print(json.dumps(out), flush=True)

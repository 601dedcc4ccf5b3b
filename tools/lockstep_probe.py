"""Independent proofs per second (not a test): one at a time, T caller threads in flight, and P proofs per aleo_mi355x_varuna_prove_many call (lockstep:
every round's commitments of the P proofs in one launch chain), also from several threads.  Usage: python tools/lockstep_probe.py [lg]
`python3 tools/lockstep_probe.py <lg> trace <P> [reps]` only repeats the P-proof call (for rocprofv3 --kernel-trace + tools/trace_busy.py)."""
import os, sys, json, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from aleo_amd import synth, varuna
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 15
n, csr, z, zz, ck, D = bench._varuna_instance(synth, lg, 40 + lg)
out = {'lg': lg, 'constraints': n}
with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx:
    nx.prove(zz, 1)
    def rate(fn, proofs, reps=4):
        fn(); t = time.perf_counter()
        for _ in range(reps): fn()
        dt = (time.perf_counter() - t) / reps
        return {'ms_per_call': round(dt * 1e3, 3), 'proofs_per_s': round(proofs / dt, 1), 'constraints_per_s': round(n * proofs / dt)}
    if len(sys.argv) > 2 and sys.argv[2] == 'trace':
        P = int(sys.argv[3]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
        reqs = [([nx], [[zz]], 100 + q) for q in range(P)]
        varuna.prove_many_native(reqs); t = time.perf_counter()
        for _ in range(reps): varuna.prove_many_native(reqs)
        print(json.dumps({'lg': lg, 'P': P, 'ms_per_call': round((time.perf_counter() - t) / reps * 1e3, 3)}), flush=True)
        ck.close(); sys.exit(0)
    out['single'] = rate(lambda: nx.prove(zz, 5), 1, 8)
    for P in (2, 4, 8, 16, 32):
        reqs = [([nx], [[zz]], 100 + q) for q in range(P)]
        got = varuna.prove_many_native(reqs)
        assert got[0] == nx.prove(zz, 100) and got[-1] == nx.prove(zz, 100 + P - 1)
        out['lockstep_%d' % P] = rate(lambda: varuna.prove_many_native(reqs), P)
    for T, P in ((2, 4), (4, 2), (2, 8), (4, 8), (2, 16)):      # several lockstep calls in flight
        per = 6 if P <= 4 else 3
        def work(k):
            reqs = [([nx], [[zz]], 1000 * k + q) for q in range(P)]
            for _ in range(per): varuna.prove_many_native(reqs)
        for _ in range(2):
            th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
            t = time.perf_counter()
            for x in th: x.start()
            for x in th: x.join()
            dt = time.perf_counter() - t
        out['threads_%d_x_lockstep_%d' % (T, P)] = {'proofs_per_s': round(T * per * P / dt, 1), 'constraints_per_s': round(n * T * per * P / dt)}
ck.close()
print(json.dumps(out), flush=True)

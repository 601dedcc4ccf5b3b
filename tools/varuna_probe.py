"""Wall time of aleo_amd.varuna.prove at several circuit sizes (not a test): per-round breakdown, constraints/s."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, varuna

TAU, S_GAMMA = 0x1F3A9C0D5E7B24681357ACE02468BDF013579BDF02468ACE1234567, 0x0FEDCBA9876543210123456789ABCDEF55AA
torch.cuda.set_device(0)
BITS = '--bits' in sys.argv; LAGRANGE = '--lagrange' in sys.argv; RANGE = -1 if '--range' in sys.argv else 0      # -1: 16-bit window from 2^18 constraints, 13 below      # bit-heavy witness; Lagrange-basis powers pinned (commit_lagrange in round 1)
for lg in [int(a) for a in sys.argv[1:] if not a.startswith('--')] or [13, 15, 16]:
    n = (1 << lg) - 64
    csr, z = synth.synthetic_r1cs_bits(n, 4, 40 + lg) if BITS else synth.synthetic_r1cs(n, 4, 40 + lg, long_rows=4)
    zz = np.stack([synth.int_to_limbs(v, 4) for v in z])
    nnz = max(int(csr[m][0][-1]) for m in 'abc'); n_k = 2
    while n_k < nnz: n_k *= 2
    D = 1
    while D < max(3 << lg, n_k): D *= 2
    t0 = time.perf_counter(); ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D - 1, lagrange_size=(1 << lg) if LAGRANGE else 0, range_window=(16 if lg >= 18 else 13) if RANGE else 0); t1 = time.perf_counter()
    ix = varuna.CircuitIndex(csr, n, 4, len(z) - 4, ck); t2 = time.perf_counter()
    ts, rounds = [], []
    for rep in range(7):
        t = time.perf_counter(); pr = varuna.prove(ix, zz, 1000 + rep); ts.append((time.perf_counter() - t) * 1e3); rounds.append(pr.timing_ms)
    med = float(np.median(ts[2:]))
    t0n = time.perf_counter(); nx = varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck); native_index_s = time.perf_counter() - t0n
    assert nx.vk_bytes == ix.vk_bytes; nx.close()
    tn = []
    for rep in range(7):
        t = time.perf_counter(); varuna.prove_native(ix, zz, 1000 + rep); tn.append((time.perf_counter() - t) * 1e3)
    native = {'index_s': native_index_s, 'prove_ms': float(np.median(tn[2:])), 'rounds_ms': varuna.native_timing()}
    tn4 = []
    for rep in range(5):
        t = time.perf_counter(); varuna.prove_native(ix, [zz] * 4, 2000 + rep); tn4.append((time.perf_counter() - t) * 1e3)
    native['instances_4_ms'] = float(np.median(tn4[1:]))
    tn8 = []
    for rep in range(5):
        t = time.perf_counter(); varuna.prove_native(ix, [zz] * 8, 3000 + rep); tn8.append((time.perf_counter() - t) * 1e3)
    native['instances_8_ms'] = float(np.median(tn8[1:])); native['instances_8_constraints_per_s'] = 8 * n / native['instances_8_ms'] * 1e3
    batch = {}
    for kb in (2, 4):                                  # instances of the same circuit proved together (here: the same assignment k times)
        tb = []
        for rep in range(5):
            t = time.perf_counter(); varuna.prove(ix, [zz] * kb, 2000 + rep); tb.append((time.perf_counter() - t) * 1e3)
        mb = float(np.median(tb[1:])); batch[str(kb)] = {'prove_ms': mb, 'constraints_per_s': kb * n / mb * 1e3}
    import threading
    conc = {}
    for T in (2, 4):
        streams = [torch.cuda.Stream() for _ in range(T)]; per = 6
        def work(k):
            for rep in range(per): varuna.prove(ix, zz, 5000 + 100 * k + rep, streams[k])
        for warm in range(2):
            th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
            t = time.perf_counter()
            for x in th: x.start()
            for x in th: x.join()
            dt = time.perf_counter() - t
        conc[str(T)] = {'proofs_per_s': T * per / dt, 'constraints_per_s': n * T * per / dt}
    print(json.dumps({'lg_constraints': lg, 'bits': BITS, 'lagrange': LAGRANGE, 'range_window': RANGE, 'native': native, 'instances': batch, 'in_flight': conc, 'constraints': n, 'n_h': ix.n_h, 'n_k': ix.n_k, 'setup_s': t1 - t0, 'index_s': t2 - t1, 'prove_ms': med,
                      'constraints_per_s': n / med * 1e3, 'rounds_ms': {k: float(np.median([r[k] for r in rounds[2:]])) for k in rounds[0]}}), flush=True)
    ck.close()

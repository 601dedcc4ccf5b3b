"""Device-side timing probe (not a test): NTT and MSM phase times at several sizes, HIP events on the launch stream."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, msm as M

dev = torch.device('cuda', 0); torch.cuda.set_device(0)
aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_init_device(0), 'init')
tstream = torch.cuda.Stream(); torch.cuda.set_stream(tstream); st = tstream.cuda_stream   # a real (non-null) stream: HIP events and the library share it
what = sys.argv[1] if len(sys.argv) > 1 else 'all'
if what in ('all', 'ntt'):
    for lg in (16, 20, 22, 24, 26):
        n = 1 << lg
        x = torch.from_numpy(synth.uniform_scalars(n, lg).view(np.int64)).to(dev)
        d = aleo_amd.EvaluationDomain(n)
        for (direction, type_) in ((0, 0), (1, 1)):
            d.ntt_device(x.data_ptr(), 0, direction, type_, st); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            reps = 10
            e0.record()
            for _ in range(reps): d.ntt_device(x.data_ptr(), 0, direction, type_, st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print(json.dumps({'ntt_lg': lg, 'dir': direction, 'coset': type_, 'ms': ms, 'GBps_alg': 64.0 * n / ms / 1e6, 'frac_hbm': 64.0 * n / ms / 1e6 / 8000}), flush=True)
if what in ('all', 'msm'):
    for lg in (16, 18, 20, 22):
        n = 1 << lg
        pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n)
        if 'pre' in sys.argv: t0 = time.perf_counter(); pb.precompute(); print('precompute s', time.perf_counter() - t0)
        for kind, mk in (('uniform', synth.uniform_scalars), ('witness', synth.witness_like_scalars)):
            s = torch.from_numpy(mk(n, 77 + lg).view(np.int64)).to(dev); torch.cuda.synchronize()
            M.VariableBase.msm_device(pb, s.data_ptr(), n)
            t0 = time.perf_counter(); reps = 5
            tms = []
            for _ in range(reps):
                M.VariableBase.msm_device(pb, s.data_ptr(), n); tms.append(M.last_msm_timing())
            dt = (time.perf_counter() - t0) / reps
            avg = {k: float(np.mean([t[k] for t in tms])) for k in tms[0]}
            print(json.dumps({'msm_lg': lg, 'scalars': kind, 'wall_ms': dt * 1e3, 'Mpts_s': n / dt / 1e6, **avg}), flush=True)
        pb.close()
if what in ('all', 'frops'):
    from aleo_amd import poly
    for lg in (22, 24):
        n = 1 << lg
        a = torch.from_numpy(synth.uniform_scalars(n, 1).view(np.int64)).to(dev); b = torch.from_numpy(synth.uniform_scalars(n, 2).view(np.int64)).to(dev)
        d = torch.empty_like(a); torch.cuda.synchronize()
        for name, fn, bytes_per in (('mul', lambda: poly.fr_vec_op_device(d.data_ptr(), a.data_ptr(), b.data_ptr(), n, 0, st), 96),
                                    ('add', lambda: poly.fr_vec_op_device(d.data_ptr(), a.data_ptr(), b.data_ptr(), n, 1, st), 96),
                                    ('batch_inverse', lambda: poly.batch_inversion_device(d.data_ptr(), n, st), 64)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(json.dumps({'frop': name, 'lg': lg, 'ms': ms, 'GBps_alg': bytes_per * n / ms / 1e6, 'frac_hbm': bytes_per * n / ms / 1e6 / 8000}), flush=True)
if what in ('all', 'spmv'):
    from aleo_amd import poly
    rng = np.random.default_rng(5)
    for lg in (20, 22):
        rows = 1 << lg; cols = rows
        lens = rng.choice([1, 2, 3, 4, 8], size=rows, p=[0.35, 0.3, 0.2, 0.1, 0.05]).astype(np.int64); lens[:8] = 20000
        row_ptr = np.zeros(rows + 1, dtype=np.uint32); row_ptr[1:] = np.cumsum(lens); nnz = int(row_ptr[-1])
        col = rng.integers(0, cols, size=nnz, dtype=np.uint32)
        dv = torch.from_numpy(synth.uniform_scalars(nnz, 3).view(np.int64)).to(dev); dx = torch.from_numpy(synth.uniform_scalars(cols, 4).view(np.int64)).to(dev)
        drp = torch.from_numpy(row_ptr.view(np.int32)).to(dev); dcol = torch.from_numpy(col.view(np.int32)).to(dev)
        dy = torch.empty((rows, 4), dtype=torch.int64, device=dev)
        fn = lambda: poly.spmv_device(dy.data_ptr(), drp.data_ptr(), dcol.data_ptr(), dv.data_ptr(), dx.data_ptr(), rows, st)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(json.dumps({'spmv_rows_lg': lg, 'nnz': nnz, 'ms': ms, 'GBps_alg': 68.0 * nnz / ms / 1e6, 'frac_hbm': 68.0 * nnz / ms / 1e6 / 8000, 'Gnnz_s': nnz / ms / 1e6}), flush=True)

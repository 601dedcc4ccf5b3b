"""One named workload, repeated, for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_workload.py <name> [reps]`
(not a test).  Names:
  msm:<lg_n>:<uniform|witness>:<table|plain>[:<lg_set>]   MSM over the first 2^lg_n points of a 2^lg_set-point pinned set
  batch:<lg_n>:<k>[:<lg_set>]                              one batched commit of k polynomials of 2^lg_n coefficients
  ntt:<lg_n>[:<batch>]                                     forward NTT, device resident
  host_msm:<lg_n>                                          msm_g1_pinned with HOST scalars (upload inside the call)
  msm_range:<lg_n>[:<window>]                              witness-like scalars on the narrow-window range table over the whole set (sparse hint)
Prints one JSON line with the wall time per repetition."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, msm as M

dev = torch.device('cuda', 0); torch.cuda.set_device(0)
aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_init_device(0), 'init')
name = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
f = name.split(':')
out = {'workload': name, 'reps': reps}
if f[0] == 'round1':
    # the first Varuna round at 2^lg constraints: w, z_a, z_b (witness-like, Montgomery) + mask_poly (3 * 2^lg, uniform) in one batched commit
    from aleo_amd import wire, kzg
    lg = int(f[1]); H = 1 << lg
    pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, 3 * H).precompute()
    wit = [torch.from_numpy(wire.fr_from_bytes(synth.witness_like_scalars(H, 2 + j).view(np.uint8).reshape(-1, 32)).view(np.int64)).to(dev) for j in range(3)]
    uni = torch.from_numpy(synth.uniform_scalars(3 * H, 9).view(np.int64)).to(dev); torch.cuda.synchronize()
    ptrs = [w.data_ptr() for w in wit] + [uni.data_ptr()]; lens = [H, H, H, 3 * H]
    run = lambda: kzg.KZG10.commit_batch_device(pb, ptrs, lens)
    run(); run()
    t0 = time.perf_counter(); tms = []
    for _ in range(reps):
        run(); tms.append(M.last_msm_timing())
    out['wall_ms'] = (time.perf_counter() - t0) / reps * 1e3
    out.update({k_: float(np.mean([t[k_] for t in tms])) for k_ in tms[0]})
elif f[0] == 'msm_range':
    lg = int(f[1]); n = 1 << lg; win = int(f[2]) if len(f) > 2 else 16
    pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n).precompute(); pb.precompute_range(0, n, win)
    ds = torch.from_numpy(synth.witness_like_scalars(n, 77 + lg).view(np.int64)).to(dev); torch.cuda.synchronize()
    run = lambda: M.VariableBase.msm_device(pb, ds.data_ptr(), n, sparse=True)
    run(); run()
    t0 = time.perf_counter(); tms = []
    for _ in range(reps):
        run(); tms.append(M.last_msm_timing())
    out['wall_ms'] = (time.perf_counter() - t0) / reps * 1e3
    out.update({k_: float(np.mean([t[k_] for t in tms])) for k_ in tms[0]})
elif f[0] in ('msm', 'host_msm', 'batch'):
    lg = int(f[1]); n = 1 << lg
    if f[0] == 'msm':
        kind, table = f[2], f[3] == 'table'; lg_set = int(f[4]) if len(f) > 4 else lg
    elif f[0] == 'batch':
        kind, table = 'uniform', True; k = int(f[2]); lg_set = int(f[3]) if len(f) > 3 else lg
    else:
        kind, table, lg_set = 'uniform', True, lg
    pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, 1 << lg_set)
    if table: pb.precompute()
    mk = synth.uniform_scalars if kind == 'uniform' else synth.witness_like_scalars
    if f[0] == 'batch':
        from aleo_amd import kzg
        hs = np.stack([mk(n, 31 + j) for j in range(k)])
        ds = torch.from_numpy(hs.view(np.int64)).to(dev); torch.cuda.synchronize()
        ptrs = [ds.data_ptr() + j * n * 32 for j in range(k)]
        run = lambda: M.VariableBase.msm_batch_device(pb, ptrs, [n] * k)
    elif f[0] == 'host_msm':
        hs = mk(n, 77 + lg)
        run = lambda: M.VariableBase.msm(pb, hs)
    else:
        ds = torch.from_numpy(mk(n, 77 + lg).view(np.int64)).to(dev); torch.cuda.synchronize()
        run = lambda: M.VariableBase.msm_device(pb, ds.data_ptr(), n)
    run(); run()
    t0 = time.perf_counter(); tms = []
    for _ in range(reps):
        run(); tms.append(M.last_msm_timing())
    out['wall_ms'] = (time.perf_counter() - t0) / reps * 1e3
    out.update({k_: float(np.mean([t[k_] for t in tms])) for k_ in tms[0]})
elif f[0] == 'ntt':
    lg = int(f[1]); n = 1 << lg; batch = int(f[2]) if len(f) > 2 else 1
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
    x = torch.from_numpy(synth.uniform_scalars(n * batch, lg).view(np.int64)).to(dev)
    d = aleo_amd.EvaluationDomain(n)
    run = (lambda: d.ntt_device(x.data_ptr(), 0, 0, 0, st)) if batch == 1 else (lambda: d.ntt_batch_device(x.data_ptr(), batch, 0, 0, 0, st))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    out['ms'] = e0.elapsed_time(e1) / reps
    out['alg_GBps'] = 64.0 * n * batch / out['ms'] / 1e6
print(json.dumps(out), flush=True)

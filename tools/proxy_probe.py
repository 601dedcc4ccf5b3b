"""Per-call phase times of the five batched commitment calls of bench.py's proof proxy at 2^lg constraints (not a test)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, wire, poly, msm as M
import bench

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 15
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_init_device(0), 'init')
H = 1 << lg
pb = aleo_amd.PinnedBases.generate_multiples(synth.generator_affine104(), 1, 3 * H).precompute()
buf = torch.from_numpy(synth.uniform_scalars(4 * H, 1).view(np.int64)).to(dev)
wit = [torch.from_numpy(wire.fr_from_bytes(synth.witness_like_scalars(H, 2 + j).view(np.uint8).reshape(-1, 32)).view(np.int64)).to(dev) for j in range(3)]
torch.cuda.synchronize()
calls = [o for o in bench.proxy_schedule(lg) if o[0] in ('commit', 'open')]
for rep in range(6):
    for o in calls:
        ptrs, lens, w = [], [], 0
        items = o[1] if o[0] == 'commit' else [('uniform', m - 1) for m in o[1]]
        for kind, m in items:
            if kind == 'witness': ptrs.append(wit[w].data_ptr()); w += 1
            else: ptrs.append(buf.data_ptr() + 32 * 64 * len(ptrs))
            lens.append(m)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        aleo_amd.KZG10.commit_batch_device(pb, ptrs, lens)
        dt = (time.perf_counter() - t0) * 1e3
        if rep >= 2: print(json.dumps({'call': [(k, m) for k, m in items], 'wall_ms': round(dt, 3), **{k: round(v, 3) for k, v in M.last_msm_timing().items()}}), flush=True)

"""Child process of test_rccl_runs_the_exchange_with_one_rank: a world of ONE rank under backend 'nccl' (= RCCL on ROCm) on the visible MI355X.  One rank
is all a one-GPU box allows (RCCL refuses two ranks on one device), but it is the real library: communicator creation, the all-gather of the 144-byte
partial (aleo_amd.dist.PartialGather — the exchange of the point-sharded MSM), the MAX all-reduce of bench.py's timing and the all-to-all of the 4-step
transform (ShardedDomain with world 1 and always_collective=True: torch.distributed.all_to_all_single under 'nccl' is really called — until round 5 a world of
one returned before it) all launch RCCL kernels on the card.  The transform's values are compared with the CPU restatement (oracle), forward and inverse, plain
and coset.  Prints one JSON line."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.distributed as dist
import aleo_amd
from aleo_amd import synth, dist as adist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', sys.argv[1] if len(sys.argv) > 1 else '29533')
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
out = {'backend': dist.get_backend(), 'world': dist.get_world_size()}
n = 1 << 12
with aleo_amd.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as pb:
    pb.precompute(); S = synth.uniform_scalars(n, 4711)
    part = aleo_amd.VariableBase.msm(pb, S)
    rows = adist.PartialGather(1, dev)(part)
    out['gathered_equals_partial'] = bool((np.asarray(rows).reshape(-1)[:18] == np.asarray(part).reshape(-1)).all())
    out['sum_equals_partial'] = bool((aleo_amd.g1_sum(rows) == part).all())
t = torch.tensor([3.25, 1.0], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); out['all_reduce_max'] = float(t[0].item())
from oracle import coracle as orc                            # the checker (CPU restatement), never the thing that runs
lg = 12; x = orc.fr_to_mont(synth.uniform_scalars(1 << lg, 99))
dom = adist.ShardedDomain(lg, 0, 1, always_collective=True)
idx = dom.evaluation_indices(); ok = True
for coset in (False, True):
    mine = torch.from_numpy(dom.coefficient_shard(x).view(np.int64).copy()).to(dev)
    ev = dom.forward(mine.clone(), coset=coset); torch.cuda.synchronize()
    want = orc.ntt_fr(x, 0, 0, 1 if coset else 0)            # natural order, forward, standard / coset
    ok = ok and bool((ev.cpu().numpy().view(np.uint64) == want[idx]).all())
    back = dom.inverse(ev, coset=coset); torch.cuda.synchronize()
    ok = ok and bool((back.cpu().numpy().view(np.uint64) == dom.coefficient_shard(x)).all())
out['sharded_ntt_matches_oracle'] = ok
out['all_to_all_single_calls'] = dom.collective_calls
dist.barrier(); dist.destroy_process_group()
print(json.dumps(out), flush=True)

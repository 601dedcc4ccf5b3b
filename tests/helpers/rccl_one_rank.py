"""Child process of test_rccl_runs_the_exchange_with_one_rank: a world of ONE rank under backend 'nccl' (= RCCL on ROCm) on the visible MI355X.  One rank
is all a one-GPU box allows (RCCL refuses two ranks on one device), but it is the real library: communicator creation, the all-gather of the 144-byte
partial (aleo_amd.dist.PartialGather — the exchange of the point-sharded MSM), the MAX all-reduce of bench.py's timing and the all-to-all of the 4-step
transform (ShardedDomain with world 1) all launch RCCL kernels on the card.  Prints one JSON line."""
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
lg = 12; x = synth.uniform_scalars(1 << lg, 99)
dom = adist.ShardedDomain(lg, 0, 1)
mine = torch.from_numpy(dom.coefficient_shard(x).view(np.int64).copy()).to(dev)
ev = dom.forward(mine.clone()); torch.cuda.synchronize()
full = torch.from_numpy(x.view(np.int64).copy()).to(dev); aleo_amd.EvaluationDomain(1 << lg).ntt_device(full.data_ptr(), 0, 0, 0, adist._torch_stream_handle()); torch.cuda.synchronize()
idx = torch.from_numpy(dom.evaluation_indices()).to(dev)
out['sharded_ntt_matches'] = bool((ev == full[idx]).all())
dist.barrier(); dist.destroy_process_group()
print(json.dumps(out), flush=True)

"""Witness-like vectors against a sub-range of a pinned set: ordinary tiers vs the range table (sparse hint). Not a test."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, msm as M
from aleo_amd.kzg import CommitterKey, SonicKZG10
torch.cuda.set_device(0)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20; K = int(sys.argv[2]) if len(sys.argv) > 2 else 3; win = int(sys.argv[3]) if len(sys.argv) > 3 else 13
n = 1 << lg; N = 4 * n + 8; off = 3 * n + 3
pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, N).precompute(); pb.precompute_range(off, n + 4, win)
ck = CommitterKey.__new__(CommitterKey); ck.bases = pb; ck.max_degree = N - 1; ck.gamma_offset = 0; ck.n_gamma = 0
d = [torch.from_numpy(synth.witness_like_scalars(n, 500 + q).view(np.int64)).cuda() for q in range(K)]      # canonical values used as Montgomery residues: same sparsity pattern? no -> convert
one = np.array([(1 << 256) % synth.FR_MODULUS >> (64 * i) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
from aleo_amd import poly
r2 = synth.int_to_limbs(((1 << 256) % synth.FR_MODULUS) ** 2 % synth.FR_MODULUS, 4)
for t in d: poly.fr_lin_device(t.data_ptr(), n, None, r2, t.data_ptr())          # to Montgomery form (the commit converts back): the sparsity is in the canonical values
segs = [(d[q].data_ptr(), n, off, q) for q in range(K)]
def run(sparse):
    SonicKZG10.commit_segments_device(ck, segs, K, sparse=sparse); ts = []
    for _ in range(5):
        t = time.perf_counter(); out = SonicKZG10.commit_segments_device(ck, segs, K, sparse=sparse); ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts)), aleo_amd.last_msm_timing(), out
a, ta, oa = run(False); b, tb, ob = run(True)
assert (oa == ob).all()
print(json.dumps({'lg': lg, 'sets': K, 'window': win, 'ordinary_ms': a, 'ordinary_last_chain': ta, 'range_ms': b, 'range_last_chain': tb}))
pb.close()

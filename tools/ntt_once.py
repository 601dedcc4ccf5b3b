"""Runs a few 2^lg NTTs on resident data — a small target for rocprofv3 kernel traces / PMC passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << lg
aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_init_device(0), 'init')
x = torch.from_numpy(synth.uniform_scalars(n, 1).view(np.int64)).cuda(); torch.cuda.synchronize()
d = aleo_amd.EvaluationDomain(n)
for _ in range(5): d.ntt_device(x.data_ptr(), 0, 0, 0)
torch.cuda.synchronize()
print('done')

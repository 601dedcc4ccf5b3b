"""Device time of small transforms (not a test): single and batched, 2^10 … 2^18, HIP events on the launch stream.  Compare ALEO_MI355X_NTT_WIDE_LG=0
(three-stage register groups, one wave per 512-element tile) with the default (one butterfly per lane for transforms of <= 2^18 elements in all)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth
import ctypes
L = aleo_amd.lib()
def ntt_fr_batch_device(ptr, lg, batch, order, direction, type_, stream):
    aleo_amd._lib.check(L.aleo_mi355x_ntt_fr_batch_device(ctypes.c_void_p(ptr), lg, batch, order, direction, type_, ctypes.c_void_p(stream)), 'ntt')
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_init_device(0), 'init')
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
out = {'wide_lg': os.environ.get('ALEO_MI355X_NTT_WIDE_LG', 'default')}
for lg in (10, 12, 13, 14, 15, 16, 17, 18):
    for batch in (1, 3, 8):
        if (batch << lg) > (1 << 20): continue
        n = 1 << lg
        x = torch.from_numpy(synth.uniform_scalars(n * batch, lg).view(np.int64)).to(dev)
        f = lambda: ntt_fr_batch_device(x.data_ptr(), lg, batch, 0, 0, 0, st)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); reps = 20
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        out['2^%d x %d' % (lg, batch)] = round(e0.elapsed_time(e1) / reps * 1e3, 1)
print(json.dumps(out))

"""Run by tests/test_gpu_parity.py in a child process with ALEO_MI355X_SLOTS=1: the FIRST call this process makes into the library is a batched
transform through a *_device entry point — once with stream == NULL (the slot's own stream; torch's default stream has handle 0), once, in a second
child, on a created stream — i.e. the first-use path of a slot's stream, events and stream-ordered scratch (scratch_acquire / scratch_release,
api.hip), which round 2 crashed in once while it was being introduced ("The host segfault of round 2").  Prints FIRST CALL OK."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import coracle as co
from aleo_amd import synth
from aleo_amd._lib import lib, check
assert os.environ.get('ALEO_MI355X_SLOTS') == '1'
mode = sys.argv[1]
lg, batch = 11, 3
x = co.fr_to_mont(synth.uniform_scalars(batch << lg, 31))
t = torch.from_numpy(x.view(np.int64).copy()).cuda(); torch.cuda.synchronize()
st = torch.cuda.Stream() if mode == 'stream' else None
handle = st.cuda_stream if st is not None else 0
check(lib().aleo_mi355x_ntt_fr_batch_device(ctypes.c_void_p(t.data_ptr()), lg, batch, 0, 0, 0, ctypes.c_void_p(handle)), 'first call: ntt_fr_batch_device')      # nothing before this
check(lib().aleo_mi355x_ntt_fr_batch_device(ctypes.c_void_p(t.data_ptr()), lg, batch, 0, 1, 0, ctypes.c_void_p(handle)), 'second call')                          # scratch handed on
if st is not None: st.synchronize()
torch.cuda.synchronize()
assert (t.cpu().numpy().view(np.uint64) == x).all(), 'round trip'
check(lib().aleo_mi355x_ntt_fr_batch_device(ctypes.c_void_p(t.data_ptr()), lg, batch, 0, 0, 0, ctypes.c_void_p(handle)), 'third call')
if st is not None: st.synchronize()
torch.cuda.synchronize()
got = t.cpu().numpy().view(np.uint64).reshape(batch, 1 << lg, 4)
for b in range(batch): assert (got[b] == co.ntt_fr(x.reshape(batch, 1 << lg, 4)[b], 0, 0, 0)).all()
print('FIRST CALL OK', mode)

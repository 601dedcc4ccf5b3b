"""Two host threads, two streams, ONE library slot (run with ALEO_MI355X_SLOTS=1): every asynchronous *_device call hands the
slot's scratch buffer to the other thread's next call while its own kernels may still be running.  Results are compared with the
oracle.  Prints 'SCRATCH OK' (tests/test_gpu_parity.py runs this as a child process)."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, poly, msm as M
from oracle import coracle as c
import util

assert os.environ.get('ALEO_MI355X_SLOTS') == '1'
aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_init_device(0), 'init')
lg = 18; n = 1 << lg; batch = 4                       # 4 x 2^18 = 2^20 elements per call
errs = []
x = [c.fr_to_mont(util.uniform_scalars(n * batch, 40 + t)) for t in range(2)]
exp_ntt = [np.concatenate([c.ntt_fr(v[b * n:(b + 1) * n], 0, 0, 0) for b in range(batch)]) for v in x]
exp_inv = [c.fr_batch_inverse(v) for v in x]
d = aleo_amd.EvaluationDomain(n)
bar = threading.Barrier(2)


def work(t):
    try:
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for rep in range(6):
                a = torch.from_numpy(x[t].view(np.int64).copy()).cuda(); b = a.clone(); st.synchronize()
                bar.wait()
                d.ntt_batch_device(a.data_ptr(), batch, 0, 0, 0, st.cuda_stream)          # returns at once; scratch in use on st
                poly.batch_inversion_device(b.data_ptr(), n * batch, st.cuda_stream)       # same slot, same scratch, same stream
                st.synchronize()
                assert (a.cpu().numpy().view(np.uint64).reshape(-1, 4) == exp_ntt[t]).all(), ('ntt', t, rep)
                assert (b.cpu().numpy().view(np.uint64).reshape(-1, 4) == exp_inv[t]).all(), ('inv', t, rep)
    except Exception as e:      # noqa: BLE001
        errs.append(repr(e)); bar.abort()


ths = [threading.Thread(target=work, args=(t,)) for t in range(2)]
for t in ths: t.start()
for t in ths: t.join()
assert not errs, errs

# NTT -> commit chain with stream == NULL from two threads: each call completes before it returns, so the chain is ordered
# whichever slot serves it
m = 1 << 14
pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, m)
B = pb.download()
ev = [c.fr_to_mont(util.uniform_scalars(m, 50 + t)) for t in range(2)]
want = [c.kzg_commit(B, c.ntt_fr(v, 0, 1, 0), threads=4) for v in ev]
dm = aleo_amd.EvaluationDomain(m)


def chain(t):
    try:
        torch.cuda.set_device(0)
        for rep in range(8):
            buf = torch.from_numpy(ev[t].view(np.int64).copy()).cuda(); torch.cuda.synchronize()
            dm.ntt_device(buf.data_ptr(), 0, 1, 0)                   # stream NULL
            got = aleo_amd.KZG10.commit_device(pb, buf.data_ptr(), m)
            assert (got == want[t]).all(), ('chain', t, rep)
    except Exception as e:      # noqa: BLE001
        errs.append(repr(e))


ths = [threading.Thread(target=chain, args=(t,)) for t in range(2)]
for t in ths: t.start()
for t in ths: t.join()
assert not errs, errs
print('SCRATCH OK')

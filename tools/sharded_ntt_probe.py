"""Host-buffer transforms through aleo_mi355x_ntt_fr (one device) and aleo_mi355x_ntt_fr_sharded (G shards; on a one-GPU box all on device 0: a
rehearsal of the path — what sharding buys is 1/G of the PCIe traffic and of the butterflies per device, which one card cannot show).  Not a test."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import aleo_amd
from aleo_amd import synth
for lg in [int(a) for a in sys.argv[1:]] or [20, 22, 24]:
    x = synth.uniform_scalars(1 << lg, 5 + lg); d = aleo_amd.EvaluationDomain(1 << lg)
    ref = x.copy(); d.fft_in_place(ref)
    out = {'lg_n': lg}
    def timed(fn, reps=4):
        fn(); t = time.perf_counter()
        for _ in range(reps): fn()
        return (time.perf_counter() - t) / reps * 1e3
    buf = x.copy(); out['single_device_ms'] = timed(lambda: d.fft_in_place(buf))
    for G in (1, 2, 4, 8):
        y = x.copy(); d.ntt_sharded_in_place(y, [0] * G, 0, 0); assert (y == ref).all()
        buf = x.copy(); out['sharded_%d_on_one_device_ms' % G] = timed(lambda: d.ntt_sharded_in_place(buf, [0] * G, 0, 0))
    print(json.dumps(out), flush=True)

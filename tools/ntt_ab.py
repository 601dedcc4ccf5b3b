"""Resident forward NTTs of the sizes given, timed by events on the launch stream (ms per transform, algorithmic GB/s at 64 B per element); the environment
selects the variant under test (ALEO_MI355X_NTT_DIRECT_MAX, ALEO_MI355X_NTT29, ALEO_MI355X_NTT_TILE, ...): run once per setting.  Checks fft -> ifft = id."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth
for lg in [int(a) for a in sys.argv[1:]] or [20, 21, 22, 24]:
    n = 1 << lg
    h = synth.uniform_scalars(n, 1)
    x = torch.from_numpy(h.view(np.int64)).cuda(); torch.cuda.synchronize()
    d = aleo_amd.EvaluationDomain(n); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): d.ntt_device(x.data_ptr(), 0, 0, 0, st.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); reps = 20
        e0.record(st)
        for _ in range(reps): d.ntt_device(x.data_ptr(), 0, 0, 0, st.cuda_stream)
        e1.record(st); st.synchronize()
        ms = e0.elapsed_time(e1) / reps
        y = torch.from_numpy(h.view(np.int64)).cuda()
        d.ntt_device(y.data_ptr(), 0, 0, 0, st.cuda_stream); d.ntt_device(y.data_ptr(), 0, 1, 0, st.cuda_stream); st.synchronize()
    ok = bool((y.cpu().numpy().view(np.uint64).reshape(-1, 4) == h).all())
    print(json.dumps({'lg_n': lg, 'ms': round(ms, 4), 'alg_GBps': round(64.0 * n / ms / 1e6, 1), 'round_trip_ok': ok,
                      'env': {k: v for k, v in os.environ.items() if k.startswith('ALEO_MI355X_NTT')}}), flush=True)

"""PCIe-inclusive timing of the host-buffer entry points (not a test): the calls a snarkVM `mi355x` feature would make
hand over pageable host memory, so these wall times include the H2D/D2H copies that bench.py's `value` leaves out."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault('ALEO_MI355X_SRS_CACHE', '1')        # the one-shot probe below measures the opt-in cache
import aleo_amd
from aleo_amd import synth, msm as M

torch.cuda.set_device(0)
aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_init_device(0), 'init')


def wall(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


for lg in (16, 20, 22):
    n = 1 << lg
    pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n).precompute()
    S = synth.uniform_scalars(n, 900 + lg)
    dS = torch.from_numpy(S.view(np.int64)).cuda(); torch.cuda.synchronize()
    ms_dev = wall(lambda: M.VariableBase.msm_device(pb, dS.data_ptr(), n))
    ms_host = wall(lambda: M.VariableBase.msm(pb, S))                     # pinned handle, pageable host scalars
    print(json.dumps({'op': 'msm_g1_pinned', 'lg': lg, 'resident_ms': ms_dev, 'host_scalars_ms': ms_host,
                      'resident_Mmuls_s': n / ms_dev / 1e3, 'host_scalars_Mmuls_s': n / ms_host / 1e3}), flush=True)
    if lg <= 20:
        B = pb.download()                                                 # 104-byte host array: the one-shot call's input
        for _ in range(4): M.VariableBase.msm(B, S)                       # warm the SRS cache (table after the 3rd hit)
        ms_one = wall(lambda: M.VariableBase.msm(B, S))
        print(json.dumps({'op': 'msm_g1 (one-shot, SRS cache warm)', 'lg': lg, 'ms': ms_one, 'Mmuls_s': n / ms_one / 1e3}), flush=True)
    pb.close()
    x = synth.uniform_scalars(n, 950 + lg)
    d = aleo_amd.EvaluationDomain(n)
    dx = torch.from_numpy(x.view(np.int64)).cuda(); torch.cuda.synchronize()
    ms_dev = wall(lambda: (d.ntt_device(dx.data_ptr(), 0, 0, 0), torch.cuda.synchronize()))
    ms_host = wall(lambda: d.fft_in_place(x))                             # the C call on the caller's own (pageable, already touched) buffer: upload + NTT + download
    ms_copy = wall(lambda: d.fft(x))                                      # the same through a fresh zero-padded numpy copy (host-side page faults included)
    print(json.dumps({'op': 'ntt_fr', 'lg': lg, 'resident_ms': ms_dev, 'host_inout_ms': ms_host, 'host_inout_with_fresh_copy_ms': ms_copy,
                      'resident_GBps_alg': 64.0 * n / ms_dev / 1e6, 'host_GBps_alg': 64.0 * n / ms_host / 1e6}), flush=True)

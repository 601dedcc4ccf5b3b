"""Circuit-density sweep of the native prover (not a test): constraints/s against the non-zeros per row of A and B, and on a
Poseidon-gadget-shaped circuit.  Per circuit: the domains |H|, |K_A|, |K_B|, |K_C|, the MSM points one proof commits per constraint, key synthesis
and proof time.  Usage: python tools/density_sweep.py [lg ...]   (default 15 18)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import aleo_amd
from aleo_amd import synth, varuna

TAU, S_GAMMA = 0x1F3A9C0D5E7B24681357ACE02468BDF013579BDF02468ACE1234567, 0x0FEDCBA9876543210123456789ABCDEF55AA


def circuits(lg):
    n = (1 << lg) - 64
    yield 'synthetic (2.0 / 1.5 / 1 non-zeros per row: the headline circuit)', lambda: synth.synthetic_r1cs(n, 4, 40 + lg, long_rows=4)
    for d in (2, 4, 8, 16):
        yield 'density %d / %d' % (d, d), (lambda d=d: synth.synthetic_r1cs_density(n, 4, 900 + d, d, d))
    yield 'poseidon-shaped, width 9 (2.8 / 4.6 / 1)', lambda: synth.synthetic_r1cs_poseidon(n, 4, 77, 9)
    yield 'poseidon-shaped, width 3 (1.8 / 2.6 / 1)', lambda: synth.synthetic_r1cs_poseidon(n, 4, 78, 3)
    yield 'hash_psd2 chain: the real Poseidon gadget, %d hashes (3.6 / 6.2 / 1)' % (n // 276), lambda: synth.poseidon_chain_r1cs(n // 276, 79)[:2]


def msm_points(n_h, km):
    """G1 points multiplied per single-instance proof: w, z_a, z_b (|H| + 1 each) + mask (3|H|) | g_1, h_1 | g_M | h_2 | the two opening witnesses."""
    return 3 * (n_h + 1) + 3 * n_h + (n_h - 1) + 2 * n_h + sum(k - 1 for k in km) + max(km) + (3 * n_h - 1) + (max(km) - 1)


def run(lg, reps=8):
    out = []
    n = (1 << lg) - 64
    for name, make in circuits(lg):
        t0 = time.perf_counter(); csr, z = make(); gen_s = time.perf_counter() - t0
        zz = np.stack([synth.int_to_limbs(v, 4) for v in z])
        n = len(csr['a'][0]) - 1; n_pub = 2 if 'hash_psd2' in name else 4
        nnz = [int(csr[m][0][-1]) for m in 'abc']; n_k = 2
        while n_k < max(nnz): n_k *= 2
        D = 1
        while D < max(3 << lg, n_k): D *= 2
        ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D - 1)
        t0 = time.perf_counter()
        with varuna.NativeCircuitIndex(csr, n, n_pub, len(z) - n_pub, ck) as nx:
            index_s = time.perf_counter() - t0
            ts = []
            for rep in range(reps + 2):
                t = time.perf_counter(); nx.prove(zz, 100 + rep); ts.append((time.perf_counter() - t) * 1e3)
            t8 = []
            for rep in range(4):
                t = time.perf_counter(); nx.prove([zz] * 8, 300 + rep); t8.append((time.perf_counter() - t) * 1e3)
            ms = float(np.median(ts[2:])); ms8 = float(np.median(t8[1:]))
            pts = msm_points(nx.n_h, nx.n_k_m)
            out.append({'circuit': name, 'constraints': n, 'nnz_per_row': [round(v / n, 2) for v in nnz], 'n_h': nx.n_h, 'n_k': nx.n_k_m, 'msm_points_per_proof': pts,
                        'msm_points_per_constraint': round(pts / n, 1), 'key_synthesis_s': index_s, 'prove_ms': ms, 'constraints_per_s': n / ms * 1e3,
                        'instances_8_ms': ms8, 'instances_8_constraints_per_s': 8 * n / ms8 * 1e3, 'generate_s': gen_s})
            print(json.dumps(out[-1]), flush=True)
        ck.close()
    return out


if __name__ == '__main__':
    torch.cuda.set_device(0)
    for lg in [int(a) for a in sys.argv[1:]] or [15, 18]: run(lg)

"""Randomised differential soak of the two operators on the GPU (marked gpu): random prefixes / batches / scalar distributions of MSMs against the
O(n) structured identity sum_i s_i (i+1) G = (sum_i s_i (i+1) mod r) G on a 3 * 2^20-point pinned set — every table tier, the range table, the
slice trees beside the reduction, multi-chain requests, host scalars in chunks over two or three contexts —, random transforms (size, variant, order,
extreme or uniform inputs) against the restatement, random G2 MSMs against the identity over Fq2 in Python integers, and random commitment requests
(segments at random offsets) against a sharded copy of the set, byte-equal to the single-device call.  ALEO_SOAK_SECONDS sets the duration (default 20 s; the round's long run: profiles/r03_soak.json)."""
import os, time, json
import numpy as np
import pytest
from oracle import coracle as c, pyref as p
from tests import util

pytestmark = pytest.mark.gpu


def _scalars(rng, n, kind, seed):
    if kind == 'uniform': return util.uniform_scalars(n, seed)
    if kind == 'witness': return util.witness_like_scalars(n, seed)
    if kind == 'equal': return np.tile(util.uniform_scalars(1, seed), (n, 1))
    if kind == 'small': S = np.zeros((n, 4), dtype=np.uint64); S[:, 0] = rng.integers(0, 1 << 16, size=n, dtype=np.uint64); return S
    if kind == 'top': S = np.tile(c.ints_to_limbs([p.FR_MODULUS - 1], 4), (n, 1)); S[:, 0] -= rng.integers(0, 4, size=n, dtype=np.uint64); return S
    if kind == 'ones': S = np.zeros((n, 4), dtype=np.uint64); S[rng.random(n) < 0.7, 0] = 1; return S
    S = np.zeros((n, 4), dtype=np.uint64); S[n // 2] = util.uniform_scalars(1, seed)[0]; return S


def test_soak_msm_and_ntt():
    import torch
    import aleo_amd
    from aleo_amd import msm as M, synth
    seconds = float(os.environ.get('ALEO_SOAK_SECONDS', '20'))
    rng = np.random.default_rng(int(os.environ.get('ALEO_SOAK_SEED', '20261004')))
    N = 3 << 20
    kinds = ['uniform', 'witness', 'equal', 'small', 'top', 'ones', 'single']
    stats = {'msm_host': 0, 'msm_device': 0, 'msm_batch_results': 0, 'msm_sparse_hint': 0, 'ntt': 0, 'points': 0, 'ntt_elements': 0, 'msm_g2': 0, 'g2_points': 0,
             'sharded_commit_results': 0, 'sharded_points': 0}
    from aleo_amd.kzg import SonicKZG10
    G2B = synth.g2_multiples_affine200(1 << 12)                  # P_i = ((i mod 4096) + 1) G2, repeated to the length asked
    t_end = time.time() + seconds
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, N) as pb:
        pb.precompute(); pb.precompute_range(0, 1 << 20, 16)
        sharded = [aleo_amd.ShardedBases(pb.download(0, 1 << 20), devices=[0] * g, precompute=True) for g in (3, 8)]      # the first 2^20 points once more, cut in 3 and in 8
        class _CK: bases = pb
        it = 0; t_say = time.time() + 30
        while time.time() < t_end:
            it += 1
            if time.time() > t_say: print('soak progress ' + json.dumps(stats), flush=True); t_say = time.time() + 30      # a long run must keep writing (run with -s)
            mode = it % 6
            if mode == 0:                                                              # host scalars (from 2^21 points on: two halves on two contexts)
                n = int(rng.choice([1, 255, 1 << 14, (1 << 20) + 3, 1 << 21, N, int(rng.integers(1, N))]))
                kind = kinds[int(rng.integers(len(kinds)))]; S = _scalars(rng, n, kind, 50000 + it)
                got = c.jac_to_int_point(M.VariableBase.msm(pb, S))
                assert got == util.expected_multiples_msm(S, n), ('host', n, kind, it)
                stats['msm_host'] += 1; stats['points'] += n
            elif mode == 1:                                                            # device scalars, with and without the sparse hint
                n = int(rng.integers(1, 1 << 20)); kind = kinds[int(rng.integers(len(kinds)))]; S = _scalars(rng, n, kind, 60000 + it)
                d = torch.from_numpy(S.view(np.int64).copy()).cuda(); torch.cuda.synchronize()
                sparse = bool(rng.integers(2))
                got = c.jac_to_int_point(M.VariableBase.msm_device(pb, d.data_ptr(), n, sparse=sparse))
                assert got == util.expected_multiples_msm(S, n), ('device', n, kind, sparse, it)
                stats['msm_device'] += 1; stats['msm_sparse_hint'] += int(sparse); stats['points'] += n
            elif mode == 2:                                                            # one request, several results of different lengths (one or several launch chains)
                k = int(rng.integers(2, 7)); lens = [int(rng.choice([int(rng.integers(1, 1 << 16)), int(rng.integers(1 << 16, 1 << 21))])) for _ in range(k)]
                Ss = [_scalars(rng, n, kinds[int(rng.integers(len(kinds)))], 70000 + 10 * it + j) for j, n in enumerate(lens)]
                ds = [torch.from_numpy(S.view(np.int64).copy()).cuda() for S in Ss]; torch.cuda.synchronize()
                out = M.VariableBase.msm_batch_device(pb, [d.data_ptr() for d in ds], lens)
                for j in range(k): assert c.jac_to_int_point(out[j]) == util.expected_multiples_msm(Ss[j], lens[j]), ('batch', lens, j, it)
                stats['msm_batch_results'] += k; stats['points'] += sum(lens)
            elif mode == 4:                                                            # G2 (lane-pair kernels): host bases + scalars, one-shot
                n = int(rng.choice([1, 2, 63, 4096, int(rng.integers(1, 1 << 16)), int(rng.integers(1 << 16, 1 << 18))]))
                kind = kinds[int(rng.integers(len(kinds)))]; S = _scalars(rng, n, kind, 90000 + it)
                B = np.tile(G2B, ((n + 4095) // 4096, 1))[:n]
                assert synth.g2_result_gate(M.msm_g2(B, S), S), ('g2', n, kind, it)
                stats['msm_g2'] += 1; stats['g2_points'] += n
            elif mode == 5:                                                            # commitments against a sharded copy of the first 2^20 points
                sb = sharded[int(rng.integers(2))]; k = int(rng.integers(1, 5)); segs = []; keep = []
                for _ in range(int(rng.integers(1, 9))):
                    m = int(rng.choice([0, 1, int(rng.integers(1, 1 << 12)), int(rng.integers(1 << 12, 1 << 19))])); off = int(rng.integers(0, (1 << 20) - m + 1))
                    x = c.fr_to_mont(_scalars(rng, max(m, 1), kinds[int(rng.integers(len(kinds)))], 95000 + 16 * it + len(segs)))
                    t = torch.from_numpy(x.view(np.int64).copy()).cuda(); keep.append(t); segs.append((t.data_ptr(), m, off, int(rng.integers(k)))); stats['sharded_points'] += m
                torch.cuda.synchronize()
                assert (SonicKZG10.commit_segments_sharded_device(sb, segs, k) == SonicKZG10.commit_segments_device(_CK, segs, k)).all(), ('sharded', segs, k, it)
                stats['sharded_commit_results'] += k
            else:                                                                      # transforms
                lg = int(rng.integers(1, 23)); n = 1 << lg
                if rng.random() < 0.3:
                    x = np.zeros((n, 4), dtype=np.uint64); x[:: int(rng.integers(1, 3))] = c.ints_to_limbs([p.FR_MODULUS - 1], 4)[0]
                else: x = c.fr_to_mont(util.uniform_scalars(n, 80000 + it))
                order = int(rng.choice([0, 0, 0, 1, 2, 3])); direction = int(rng.integers(2)); type_ = int(rng.integers(2)) if order == 0 else 0
                dom = aleo_amd.EvaluationDomain(n)
                assert (dom.ntt(x, order, direction, type_) == c.ntt_fr(x, order, direction, type_, threads=8)).all(), ('ntt', lg, order, direction, type_, it)
                stats['ntt'] += 1; stats['ntt_elements'] += n
        for sb in sharded: sb.close()
    stats.update({'seconds': seconds, 'iterations': it, 'all_equal_to_the_restatement': True})
    print('SOAK ' + json.dumps(stats))
    assert it >= 6


def test_soak_prover():
    """Random proofs through the native prover — 1 to 4 circuits of random sizes (1 … 400 constraints), public-input counts and domain policies,
    1 to 4 instances each, single calls and lockstep calls — each byte for byte the restatement's proof and accepted by its verifier.
    ALEO_SOAK_SECONDS sets the duration (default 20 s; every case costs a second or two of the plain-Python restatement)."""
    from aleo_amd import varuna, synth
    from oracle import varuna_ref as V
    from tests import test_varuna as TV
    seconds = float(os.environ.get('ALEO_SOAK_SECONDS', '20'))
    rng = np.random.default_rng(int(os.environ.get('ALEO_SOAK_SEED', '20261004')) + 1)
    lim = lambda a: np.stack([synth.int_to_limbs(v, 4) for v in a])
    stats = {'proofs': 0, 'lockstep_calls': 0, 'circuits': 0, 'instances': 0, 'constraints': 0}
    t_end = time.time() + seconds; t_say = time.time() + 30; it = 0
    while time.time() < t_end:
        it += 1
        if time.time() > t_say: print('soak progress ' + json.dumps(stats), flush=True); t_say = time.time() + 30
        m = int(rng.integers(1, 5)); domains = ['auto', 'per_matrix', 'shared'][int(rng.integers(3))]
        shapes = []
        for j in range(m):
            n = int(rng.choice([1, 2, 3, 9, int(rng.integers(4, 60)), int(rng.integers(60, 400))])); npub = int(rng.integers(1, min(n, 6) + 1))
            shapes.append((n, npub, 1000 * it + j, int(rng.integers(1, 5))))
        cs, csrs, zs, D = TV._batch_case(shapes, seed=it, domains=domains)
        setup = V.Setup(TV.TAU, TV.S_GAMMA, D); idx = [V.Index(c, setup) for c in cs]
        seed = 7000 + it
        want = V.prove_batch(list(zip(idx, zs)), setup, V.random_stream(seed, max(c.n_h for c in cs), sum(len(z) for z in zs)))[1]
        ck = varuna.synthetic_committer_key(TV.TAU, TV.S_GAMMA, D); nx = []
        try:
            for (n, npub, _, _), csr, z in zip(shapes, csrs, zs): nx.append(varuna.NativeCircuitIndex(csr, n, npub, len(z[0]) - npub, ck, domains=domains))
            za = [[lim(z) for z in zz] for zz in zs]
            got = varuna.prove_batch_native(nx, za, seed)
            assert got == want, ('single', shapes, domains, it)
            assert V.verify(idx, setup, [[z[:c.n_public] for z in zz] for c, zz in zip(cs, zs)], got), ('verify', shapes, it)
            if it % 3 == 0:                                                            # the same request twice and its first circuit alone, in lockstep
                alone = varuna.prove_batch_native(nx[:1], za[:1], seed + 1)
                out = varuna.prove_many_native([(nx, za, seed), (nx[:1], za[:1], seed + 1), (nx, za, seed)])
                assert out == [want, alone, want], ('lockstep', shapes, it)
                stats['lockstep_calls'] += 1
            stats['proofs'] += 1; stats['circuits'] += m; stats['instances'] += sum(s[3] for s in shapes); stats['constraints'] += sum(s[0] * s[3] for s in shapes)
        finally:
            for x in nx: x.close()
            ck.close()
    stats.update({'seconds': seconds, 'all_equal_to_the_restatement': True})
    print('SOAK ' + json.dumps(stats))
    assert stats['proofs'] >= 1

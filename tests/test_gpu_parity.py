"""GPU parity tests (run by the driver with `-m gpu` on a real MI355X): the HIP path, called through the C ABI,
against the oracle on the same seeded inputs, against the committed golden vectors, and — at BASELINE.json's full
sizes — through size-independent identities.  Integer work: every comparison is bit-exact."""
import json, os
import ctypes
import numpy as np
import pytest

import aleo_amd
from aleo_amd import msm as M, synth
from oracle import coracle as c, pyref as p
import util

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


def _ints(hexes): return [int(h, 16) for h in hexes]


@pytest.fixture(scope='module', autouse=True)
def device():
    L = aleo_amd.lib()
    aleo_amd._lib.check(L.aleo_mi355x_init_device(-1), 'init')     # no GPU -> hard failure, never a fallback


def _fq_random(n, seed):
    a = np.concatenate([util.uniform_scalars(n, seed), util.uniform_scalars(n, seed + 1)[:, :2]], axis=1)
    a[:, 5] &= np.uint64((1 << 56) - 1)       # < 2^376 < q
    return a


# ---- field arithmetic -----------------------------------------------------------------------------
def test_fq_fr_products_bit_exact():
    n = 20000
    a, b = _fq_random(n, 1), _fq_random(n, 3)
    a[0] = 0; b[1] = 0; a[2] = c.ints_to_limbs([p.FQ_MODULUS - 1], 6)[0]; b[2] = a[2]     # edge operands
    e = np.zeros_like(a); c.lib().oracle_fq_mul(c._p(e), c._p(a), c._p(b), n)
    assert (M.fq_mul(a, b) == e).all()
    a[3] = c.ints_to_limbs([(1 << 377) - 1], 6)[0]; a[4, :] = np.uint64(0xffffffffffffffff); a[4, 5] = np.uint64((1 << 58) - 1)   # unreduced operands: carries between the doubled limbs
    e = np.zeros_like(a); c.lib().oracle_fq_mul(c._p(e), c._p(a), c._p(a), n)
    assert (M.fq_mul(a, a) == e).all()                         # same buffer twice -> the squaring block
    x, y = util.uniform_scalars(n, 5), util.uniform_scalars(n, 6)
    x[0] = 0; x[1] = c.ints_to_limbs([p.FR_MODULUS - 1], 4)[0]; y[1] = x[1]
    e = np.zeros_like(x); c.lib().oracle_fr_mul(c._p(e), c._p(x), c._p(y), n)
    assert (M.fr_mul(x, y) == e).all()
    x[2, :] = np.uint64(0xffffffffffffffff); x[2, 3] = np.uint64((1 << 62) - 1)
    e = np.zeros_like(x); c.lib().oracle_fr_mul(c._p(e), c._p(x), c._p(x), n)
    assert (M.fr_mul(x, x) == e).all()


def test_accumulation_arithmetic_28bit_limbs_matches_32bit():
    """The bucket-accumulation kernel computes in a 14 x 28-bit-limb representation (fp28.h); its mixed addition must agree
    with the 32-bit formulas (which the product tests above pin to the oracle) on every intermediate, mod q."""
    import ctypes
    bad = ctypes.c_uint32(123)
    aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_selftest_madd28(1 << 16, 32, 0x9E3779B97F4A7C15, ctypes.byref(bad)), 'selftest')
    assert bad.value == 0


def test_lane_quad_addition_matches_lane_pair():
    """The reduction chains add XYZZ points with four lanes per addition (fp28.h xyzz28_add_quad); it must agree with the lane-pair form — which every
    MSM parity test pins — on random operands, identity operands, equal points (doubling) and opposite points."""
    import ctypes
    bad = ctypes.c_uint32(123)
    aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_selftest_addquad(1 << 16, 0x9E3779B97F4A7C15, ctypes.byref(bad)), 'selftest')
    assert bad.value == 0


# ---- NTT ----------------------------------------------------------------------------------------------
def test_ntt_golden_vectors():
    fx = json.load(open(os.path.join(G, 'ntt_small.json')))
    for case in fx['cases']:
        x = c.fr_to_mont(c.ints_to_limbs(_ints(case['input']), 4))
        d = aleo_amd.EvaluationDomain(case['n'])
        for name, fn in {'fft': d.fft, 'ifft': d.ifft, 'coset_fft': d.coset_fft, 'coset_ifft': d.coset_ifft}.items():
            if name not in case: continue
            got = c.limbs_to_ints(c.fr_from_mont(fn(x)))        # ragged inputs are zero-padded by the domain
            assert got == _ints(case[name]), (case['n'], name)


@pytest.mark.parametrize('lg', [1, 2, 3, 7, 10, 11, 12, 15, 18, 19, 20])
def test_ntt_matches_oracle_all_variants(lg):
    x = c.fr_to_mont(util.uniform_scalars(1 << lg, 100 + lg))
    d = aleo_amd.EvaluationDomain(1 << lg)
    for direction in (0, 1):
        for type_ in (0, 1):
            assert (d.ntt(x, 0, direction, type_) == c.ntt_fr(x, 0, direction, type_)).all(), (lg, direction, type_)


@pytest.mark.parametrize('lg,batch,src_len,stride', [(3, 2, 5, 7), (9, 3, 128, 128), (11, 4, 2047, 2048), (13, 3, 2048, 4096), (15, 2, 1 << 13, 1 << 13), (17, 3, 1 << 15, 1 << 15),
                                                     (17, 1, 1 << 17, 1 << 17), (16, 16, 1 << 14, 1 << 14), (19, 2, 1 << 18, 1 << 18), (20, 1, (1 << 18) + 3, 1 << 19),
                                                     (21, 1, 1 << 20, 1 << 20), (23, 1, 1 << 21, 1 << 21), (10, 2, 0, 0)])
def test_ntt_from_zero_padded_source_matches_oracle(lg, batch, src_len, stride):
    """aleo_mi355x_ntt_fr_from_device: out of place, the input zero-padded from src_len to the domain (EvaluationDomain::fft on a shorter coefficient vector),
    on every kernel family (one pass, latency tiles, 64-lane tiles, 29-bit 72 / 144 KiB tiles, three passes) against the restatement fed the padded vector."""
    import torch
    n = 1 << lg
    src = c.fr_to_mont(util.uniform_scalars(max(1, (batch - 1) * stride + src_len), 4100 + lg))
    d_src = torch.from_numpy(src.view(np.int64)).cuda()
    d_out = torch.full((batch * n, 4), -1, dtype=torch.int64, device='cuda')
    dom = aleo_amd.EvaluationDomain(n)
    for direction, type_ in ((0, 0), (0, 1), (1, 0), (1, 1)) if lg <= 17 else ((0, 0), (1, 1)):
        d_out.fill_(-1)
        dom.ntt_from_device(d_out.data_ptr(), d_src.data_ptr(), stride, src_len, batch, direction, type_, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(np.uint64)
        for b in range(batch):
            x = np.zeros((n, 4), dtype=np.uint64); x[:src_len] = src[b * stride:b * stride + src_len]
            assert (got[b * n:(b + 1) * n] == c.ntt_fr(x, 0, direction, type_)).all(), (lg, b, direction, type_)
    assert (d_src.cpu().numpy().view(np.uint64) == src).all()          # the source is only read


def test_ntt_from_rejects_overlap_and_bad_sizes():
    import torch
    buf = torch.zeros((1 << 12, 4), dtype=torch.int64, device='cuda')
    dom = aleo_amd.EvaluationDomain(1 << 10)
    with pytest.raises(aleo_amd.AleoMi355xError): dom.ntt_from_device(buf.data_ptr(), buf.data_ptr() + 32 * 512, 1024, 1024)        # output overlaps the source
    with pytest.raises(aleo_amd.AleoMi355xError): dom.ntt_from_device(buf.data_ptr(), buf.data_ptr() + 32 * 2048, 1024, 1025)       # more coefficients than the domain


@pytest.mark.parametrize('lg,G', [(4, 1), (4, 4), (10, 2), (11, 8), (16, 4), (17, 2), (20, 8)])
def test_ntt_sharded_device_resident_matches_oracle(lg, G):
    """aleo_mi355x_ntt_fr_sharded_device: the data stays in HBM on the caller's device, the 4-step transform runs over G shard contexts (the one card listed G
    times: slab pulls, the pairwise exchange and the pushes are same-device copies here) — every variant against the restatement, in place."""
    import torch
    n = 1 << lg
    x = c.fr_to_mont(util.uniform_scalars(n, 5200 + lg))
    dom = aleo_amd.EvaluationDomain(n)
    for direction, type_ in ((0, 0), (0, 1), (1, 0), (1, 1)):
        d = torch.from_numpy(x.view(np.int64).copy()).cuda()
        dom.ntt_sharded_device(d.data_ptr(), [0] * G, direction, type_, stream=torch.cuda.current_stream().cuda_stream)
        assert (d.cpu().numpy().view(np.uint64) == c.ntt_fr(x, 0, direction, type_)).all(), (lg, G, direction, type_)
    with pytest.raises(aleo_amd.AleoMi355xError): dom.ntt_sharded_device(d.data_ptr(), [0] * 3)      # not a power of two


def _extreme_fr(n, kind):
    """Raw 32-byte values that push the lazy sums of the 29-bit-limb butterflies (fr29.h) to their bounds: the largest canonical number everywhere,
    alternating with zero, or in one half only."""
    top = c.ints_to_limbs([p.FR_MODULUS - 1], 4)[0]
    x = np.zeros((n, 4), dtype=np.uint64)
    if kind == 'all': x[:] = top
    elif kind == 'alternate': x[0::2] = top
    elif kind == 'half': x[:n // 2] = top
    else: x[0] = top
    return x


@pytest.mark.parametrize('lg,batch', [(11, 512), (16, 16), (19, 1), (20, 1), (21, 1)])
def test_ntt_large_tiles_extreme_inputs(lg, batch):
    """The large-tile kernels (9 x 29-bit limbs, no conditional subtraction inside a butterfly) on inputs whose sums grow as fast as they can:
    one pass (2^11 in a batch), two passes on 72 KiB tiles (2^16 in a batch, 2^19), two passes on 144 KiB tiles (2^20, 2^21) — every variant
    against the restatement."""
    import torch
    from aleo_amd import dist as adist
    n = 1 << lg
    if batch > 1:
        ops = adist.HipLocalOps()
        kinds = ['all', 'alternate', 'half', 'one']
        x = np.concatenate([_extreme_fr(n, kinds[b % 4]) if b < 8 else c.fr_to_mont(util.uniform_scalars(n, 9100 + b)) for b in range(batch)])
        for direction in (0, 1):
            t = torch.from_numpy(x.view(np.int64).copy()).cuda()
            ops.batch_ntt(t, lg, batch, direction); torch.cuda.synchronize()
            got = t.cpu().numpy().view(np.uint64).reshape(batch, n, 4)
            for b in list(range(9)) + [batch - 1]:
                assert (got[b] == c.ntt_fr(x[b << lg:(b + 1) << lg], 0, direction, 0)).all(), (lg, b, direction)
        return
    d = aleo_amd.EvaluationDomain(n)
    for kind in ('all', 'alternate', 'half'):
        x = _extreme_fr(n, kind)
        for direction in (0, 1):
            for type_ in (0, 1):
                assert (d.ntt(x, 0, direction, type_) == c.ntt_fr(x, 0, direction, type_, threads=8)).all(), (lg, kind, direction, type_)


def test_ntt_three_passes_matches_oracle_2_23():
    """2^23: three passes on 72 KiB tiles (29-bit limbs), uniform and all-maximal inputs against the restatement (threaded)."""
    lg = 23; n = 1 << lg
    d = aleo_amd.EvaluationDomain(n)
    for x, direction, type_ in ((c.fr_to_mont(util.uniform_scalars(n, 2323)), 0, 1), (_extreme_fr(n, 'all'), 1, 0)):
        assert (d.ntt(x, 0, direction, type_) == c.ntt_fr(x, 0, direction, type_, threads=16)).all(), (direction, type_)


@pytest.mark.parametrize('lg', [3, 9, 13])
def test_ntt_orders(lg):
    x = c.fr_to_mont(util.uniform_scalars(1 << lg, 300 + lg))
    d = aleo_amd.EvaluationDomain(1 << lg)
    for order in (1, 2, 3):
        for direction in (0, 1):
            assert (d.ntt(x, order, direction, 0) == c.ntt_fr(x, order, direction, 0)).all(), (lg, order, direction)


def test_ntt_full_size_properties_2_22():
    """BASELINE config[2] size: round trips, linearity and a spot check of the definition at 2^22."""
    lg = 22; n = 1 << lg
    d = aleo_amd.EvaluationDomain(n)
    x = c.fr_to_mont(util.uniform_scalars(n, 2222)); y = c.fr_to_mont(util.uniform_scalars(n, 2223))
    fx = d.fft(x)
    assert (d.ifft(fx) == x).all()
    assert (d.coset_ifft(d.coset_fft(x)) == x).all()
    # linearity: fft(x + y) == fft(x) + fft(y), checked with python ints on 4096 sampled output positions
    fy = d.fft(y)
    idx = (util.splitmix_limbs(77, 4096) % np.uint64(n)).astype(np.int64)
    xy = _mont_add(x, y)
    fxy = d.fft(xy)
    FX = c.limbs_to_ints(fx[idx]); FY = c.limbs_to_ints(fy[idx]); FXY = c.limbs_to_ints(fxy[idx])
    assert all((a + b) % p.FR_MODULUS == s_ for a, b, s_ in zip(FX, FY, FXY))
    # definition spot check: out[k] = sum_j x[j] w^(jk) for k = 0 (plain sum) and k = n/2 (alternating sum)
    xc = c.limbs_to_ints(c.fr_from_mont(x))
    out = c.limbs_to_ints(c.fr_from_mont(fx[[0, n // 2]]))
    assert out[0] == sum(xc) % p.FR_MODULUS
    assert out[1] == (sum(xc[0::2]) - sum(xc[1::2])) % p.FR_MODULUS


def _mont_add(a, b):
    """limb-wise (a + b) mod r on uint64[n,4] arrays with numpy (test helper)."""
    r = np.array([(p.FR_MODULUS >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
    out = np.zeros_like(a); carry = np.zeros(a.shape[0], dtype=np.uint64)
    for i in range(4):
        s = a[:, i] + b[:, i]; c1 = (s < a[:, i]).astype(np.uint64)
        s2 = s + carry; c2 = (s2 < s).astype(np.uint64)
        out[:, i] = s2; carry = c1 + c2
    # conditional subtract r (sum < 2r < 2^254 so carry is 0)
    ge = np.zeros(a.shape[0], dtype=bool); eq = np.ones(a.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        ge |= eq & (out[:, i] > r[i]); eq &= out[:, i] == r[i]
    ge |= eq
    borrow = np.zeros(a.shape[0], dtype=np.uint64); res = out.copy()
    for i in range(4):
        d1 = out[:, i] - r[i]; b1 = (out[:, i] < r[i]).astype(np.uint64)
        d2 = d1 - borrow; b2 = (d1 < borrow).astype(np.uint64)
        res[:, i] = d2; borrow = b1 + b2
    out[ge] = res[ge]
    return out


# ---- MSM ----------------------------------------------------------------------------------------------
def test_msm_golden_vectors():
    fx = json.load(open(os.path.join(G, 'msm_small.json')))
    for case in fx['cases']:
        pts = [p.g1_mul(p.G1_GENERATOR, k) if k else None for k in case['base_multipliers']]
        B = c.affine_from_ints(pts); S = c.ints_to_limbs(_ints(case['scalars']), 4)
        exp = None if case['result'] is None else (int(case['result'][0], 16), int(case['result'][1], 16))
        assert c.jac_to_int_point(M.VariableBase.msm(B, S)) == exp, (case['n'], case['kind'])


@pytest.mark.parametrize('n', [1, 2, 3, 15, 31, 32, 33, 255, 1000, 4097, 1 << 14])
@pytest.mark.parametrize('kind', ['uniform', 'witness'])
def test_msm_matches_oracle(n, kind):
    B = util.multiples_bases(n)
    S = util.uniform_scalars(n, 500 + n) if kind == 'uniform' else util.witness_like_scalars(n, 600 + n)
    exp = c.jac_to_int_point(c.msm_g1(B, S, threads=8, variant=1))
    assert c.jac_to_int_point(M.VariableBase.msm(B, S)) == exp
    assert c.jac_to_int_point(M.VariableBase.msm(np.ascontiguousarray(B[:, :96]), S)) == exp      # stride 96


def test_msm_edge_cases():
    n = 600
    B = util.multiples_bases(n)
    ident = None
    assert c.jac_to_int_point(M.VariableBase.msm(B[:0], np.zeros((0, 4), dtype=np.uint64))) is ident       # empty
    assert c.jac_to_int_point(M.VariableBase.msm(B, np.zeros((n, 4), dtype=np.uint64))) is ident            # all zero
    ones = np.zeros((n, 4), dtype=np.uint64); ones[:, 0] = 1
    assert c.jac_to_int_point(M.VariableBase.msm(B, ones)) == p.g1_mul(p.G1_GENERATOR, n * (n + 1) // 2)    # all one
    rm1 = np.tile(c.ints_to_limbs([p.FR_MODULUS - 1], 4), (n, 1))
    assert c.jac_to_int_point(M.VariableBase.msm(B, rm1)) == p.g1_neg(p.g1_mul(p.G1_GENERATOR, n * (n + 1) // 2))
    eq = np.tile(util.uniform_scalars(1, 9), (n, 1))                                                          # all equal -> one bucket per window
    assert c.jac_to_int_point(M.VariableBase.msm(B, eq)) == c.jac_to_int_point(c.msm_g1(B, eq, threads=4, variant=1))
    same = np.repeat(B[:1], n, axis=0); S = util.uniform_scalars(n, 10)                                       # one base repeated: doublings
    assert c.jac_to_int_point(M.VariableBase.msm(same, S)) == p.g1_mul(p.G1_GENERATOR, sum(c.limbs_to_ints(S)) % p.FR_MODULUS)
    five = np.zeros((n, 4), dtype=np.uint64); five[:, 0] = 5
    assert c.jac_to_int_point(M.VariableBase.msm(same, five)) == p.g1_mul(p.G1_GENERATOR, 5 * n)
    neg = c.affine_from_ints([p.g1_neg(p.G1_GENERATOR)])
    pm = np.concatenate([B[:1], neg] * 50, axis=0); s77 = np.zeros((100, 4), dtype=np.uint64); s77[:, 0] = 77   # P, -P cancel
    assert c.jac_to_int_point(M.VariableBase.msm(pm, s77)) is ident
    Binf = B.copy(); Binf[5] = 0; Binf[5, 96] = 1; Binf[77] = 0; Binf[77, 96] = 1                             # infinity bases are skipped
    S = util.uniform_scalars(n, 11)
    assert c.jac_to_int_point(M.VariableBase.msm(Binf, S)) == c.jac_to_int_point(c.msm_g1(Binf, S, threads=4, variant=1))
    # zip semantics: more scalars than bases / more bases than scalars
    assert c.jac_to_int_point(M.VariableBase.msm(B[:100], S)) == c.jac_to_int_point(c.msm_g1(B[:100], S[:100], variant=1))


@pytest.mark.parametrize('lg', [19, 21])
def test_chunked_host_scalar_msm_edge_cases(lg):
    """The headline path from 2^19 points on: host scalars uploaded and processed in two (2^19) or three (2^21) chunks that share the buckets — every later
    chunk's accumulation is seeded with the earlier chunks' bucket sums — and ONE reduction.  Cases the seeding has to survive, each against the result's
    discrete logarithm in big integers: uniform scalars (also equal to the one-chain result), every scalar equal (one bucket per window: slice trees in
    every chunk), a first chunk of zeros (nothing to seed from), zeros in the LAST chunk (buckets only earlier chunks touched), one base repeated
    (the seed equals the next point: the mixed addition refuses and the slice finishes in the general code), P / -P alternating (sums that cancel
    across chunk boundaries), and a ragged length."""
    n = 1 << lg
    gen = synth.generator_affine104()
    with M.PinnedBases.generate_multiples(gen, 1, n) as pb:
        pb.precompute()
        S = util.uniform_scalars(n, 4000 + lg)
        got = M.VariableBase.msm(pb, S)
        assert M.last_msm_timing()['accum_launches'] == (2 if lg < 21 else 3)
        assert c.jac_to_int_point(got) == util.expected_multiples_msm(S, n)
        import torch
        d = torch.from_numpy(S.view(np.int64).copy()).cuda(); torch.cuda.synchronize()
        assert (M.VariableBase.msm_device(pb, d.data_ptr(), n) == got).all()                       # the one-chain path: same bytes
        eq = np.tile(util.uniform_scalars(1, 4100 + lg), (n, 1))
        assert c.jac_to_int_point(M.VariableBase.msm(pb, eq)) == util.expected_multiples_msm(eq, n)
        z0 = S.copy(); z0[: n // 2] = 0
        assert c.jac_to_int_point(M.VariableBase.msm(pb, z0)) == util.expected_multiples_msm(z0, n)
        z1 = S.copy(); z1[n // 3:] = 0
        assert c.jac_to_int_point(M.VariableBase.msm(pb, z1)) == util.expected_multiples_msm(z1, n)
        m = n - 12345
        assert c.jac_to_int_point(M.VariableBase.msm(pb, S[:m])) == util.expected_multiples_msm(S[:m], m)
    if lg > 19: return
    # repeated and opposite bases need a set pinned from host rows
    one = util.multiples_bases(1); neg = c.affine_from_ints([p.g1_neg(p.G1_GENERATOR)])
    same = np.ascontiguousarray(np.repeat(one, n, axis=0))
    with M.PinnedBases(same) as pb:
        pb.precompute()
        S = util.uniform_scalars(n, 4200)
        assert c.jac_to_int_point(M.VariableBase.msm(pb, S)) == p.g1_mul(p.G1_GENERATOR, sum(c.limbs_to_ints(S)) % p.FR_MODULUS)
    pm = np.ascontiguousarray(np.concatenate([one, neg] * (n // 2), axis=0))
    with M.PinnedBases(pm) as pb:
        pb.precompute()
        s77 = np.tile(util.uniform_scalars(1, 4300), (n, 1))
        assert c.jac_to_int_point(M.VariableBase.msm(pb, s77)) is None
        S = util.uniform_scalars(n, 4301); ints = c.limbs_to_ints(S)
        k = (sum(ints[0::2]) - sum(ints[1::2])) % p.FR_MODULUS
        assert c.jac_to_int_point(M.VariableBase.msm(pb, S)) == p.g1_mul(p.G1_GENERATOR, k)


def test_msm_heavy_buckets_witness_like_2_17():
    n = 1 << 17
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as pb:
        S = util.witness_like_scalars(n, 1717)
        got = M.VariableBase.msm(pb, S)
        assert c.jac_to_int_point(got) == util.expected_multiples_msm(S, n)
        eq = np.tile(util.uniform_scalars(1, 5), (n, 1))          # adversarial: every scalar equal
        assert c.jac_to_int_point(M.VariableBase.msm(pb, eq)) == util.expected_multiples_msm(eq, n)


def test_generated_bases_match_oracle():
    n = 1000
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as pb:
        assert (pb.download() == util.multiples_bases(n)).all()
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 12345678901, 130) as pb:
        got = c.affine_to_ints(pb.download())
        assert got[0] == p.g1_mul(p.G1_GENERATOR, 12345678901) and got[129] == p.g1_mul(p.G1_GENERATOR, 12345678901 + 129)


def test_msm_full_size_2_20_structured_identity():
    """BASELINE config[1] size.  Bases (i+1)G make sum_i s_i P_i = (sum_i s_i (i+1) mod r) G — an O(n) oracle."""
    n = 1 << 20
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as pb:
        for kind, S in (('uniform', util.uniform_scalars(n, 0xA1E00002)), ('witness', util.witness_like_scalars(n, 0xA1E00012))):
            got = M.VariableBase.msm(pb, S)
            assert c.jac_to_int_point(got) == util.expected_multiples_msm(S, n), kind
        # the device-pointer entry (scalars resident in HBM) and a prefix of the pinned set
        import torch
        S = util.uniform_scalars(n, 0xA1E00003)
        dS = torch.from_numpy(S.view(np.int64)).cuda(); torch.cuda.synchronize()
        got = M.VariableBase.msm_device(pb, dS.data_ptr(), n)
        assert c.jac_to_int_point(got) == util.expected_multiples_msm(S, n)
        got = M.VariableBase.msm_device(pb, dS.data_ptr(), 1 << 18)
        assert c.jac_to_int_point(got) == util.expected_multiples_msm(S, 1 << 18)
        # oracle on a 2^16 prefix (seconds on the host cores)
        B = pb.download(0, 1 << 16)
        assert c.jac_to_int_point(M.VariableBase.msm(pb, S[: 1 << 16])) == c.jac_to_int_point(c.msm_g1(B, S[: 1 << 16], threads=os.cpu_count(), variant=1))
        # tiered tables: every prefix length of the one pinned set, on both sides of each tier boundary (c = 13 / 16 / 20)
        pb.precompute()
        for m in (1000, 1 << 10, (1 << 15) - 1, 1 << 15, 1 << 17, (1 << 17) + 1, 3 << 18, n):
            got = M.VariableBase.msm_device(pb, dS.data_ptr(), m)
            assert c.jac_to_int_point(got) == util.expected_multiples_msm(S, m), m


def test_sharded_msm_two_shards():
    """The multi-GPU path on one device: two shards -> two partials -> g1_sum, against the unsharded result."""
    from aleo_amd.dist import shard_range
    n = 50001
    B = util.multiples_bases(n); S = util.uniform_scalars(n, 31337)
    parts = []
    for r in range(2):
        lo, hi = shard_range(n, r, 2)
        parts.append(M.VariableBase.msm(B[lo:hi], S[lo:hi]))
    total = aleo_amd.g1_sum(np.stack(parts))
    assert (total == M.VariableBase.msm(B, S)).all()
    assert c.jac_to_int_point(total) == util.expected_multiples_msm(S, n)


# ---- KZG commit shape (NTT output fed to the MSM without leaving HBM) ----------------------------------
def test_kzg_commit_matches_oracle_and_device_chain():
    import torch
    lg = 14; n = 1 << lg
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as pb:
        B = pb.download()
        evals = c.fr_to_mont(util.uniform_scalars(n, 4321))
        coeffs = c.ntt_fr(evals, 0, 1, 0)                                     # oracle iNTT
        exp = c.kzg_commit(B, coeffs, threads=8)
        d = aleo_amd.EvaluationDomain(n)
        got = aleo_amd.KZG10.commit(pb, d.ifft(evals))
        assert (got == exp).all()
        # device chain: iNTT in HBM -> commit from the same buffer (config[2] shape: no host round trip)
        dbuf = torch.from_numpy(evals.view(np.int64)).cuda(); torch.cuda.synchronize()
        d.ntt_device(dbuf.data_ptr(), 0, 1, 0)
        got2 = aleo_amd.KZG10.commit_device(pb, dbuf.data_ptr(), n)
        assert (got2 == exp).all()
        # leading zeros are skipped like the reference does
        cz = coeffs.copy(); cz[n - 100:] = 0
        assert (aleo_amd.KZG10.commit(pb, cz) == c.kzg_commit(B, cz, threads=8)).all()


def test_bad_arguments_are_rejected():
    L = aleo_amd.lib()
    assert L.aleo_mi355x_msm_g1(None, None, 104, None, 5) == 2
    out = np.zeros(18, dtype=np.uint64)
    assert L.aleo_mi355x_msm_g1(out.ctypes.data, out.ctypes.data, 100, out.ctypes.data, 1) == 2        # bad stride
    assert L.aleo_mi355x_ntt_fr(out.ctypes.data, 31, 0, 0, 0) == 2
    assert L.aleo_mi355x_msm_g1_pinned(out.ctypes.data, 987654321, out.ctypes.data, 1) == 4            # unknown handle
    import torch
    t = torch.zeros((16, 4), dtype=torch.int64, device='cuda'); torch.cuda.synchronize()
    assert L.aleo_mi355x_ntt_fr_device(t.data_ptr(), 4, 0, 0, 0, 2) == 2                                 # hipStreamPerThread is refused
    assert L.aleo_mi355x_fr_divide_by_linear_device(t.data_ptr(), None, t.data_ptr(), 16, out.ctypes.data, None) == 2      # quotient aliases the polynomial
    assert L.aleo_mi355x_msm_g2(out.ctypes.data, out.ctypes.data, 104, out.ctypes.data, 1) == 2          # bad G2 stride


def test_msm_fixed_base_table_path():
    """bases_precompute(): the 13-window fixed-base table must give the same group element as the plain path."""
    for n in (1000, 1 << 14, 1 << 16, 1 << 18):
        with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as pb:
            S = util.uniform_scalars(n, 8800 + n); Wt = util.witness_like_scalars(n, 8900 + n)
            plain_u, plain_w = M.VariableBase.msm(pb, S), M.VariableBase.msm(pb, Wt)
            pb.precompute()
            assert (M.VariableBase.msm(pb, S) == plain_u).all() and (M.VariableBase.msm(pb, Wt) == plain_w).all()
            assert c.jac_to_int_point(plain_u) == util.expected_multiples_msm(S, n)
            rm1 = np.tile(c.ints_to_limbs([p.FR_MODULUS - 1], 4), (n, 1))                  # top-window carry
            assert c.jac_to_int_point(M.VariableBase.msm(pb, rm1)) == util.expected_multiples_msm(rm1, n)
            assert c.jac_to_int_point(M.VariableBase.msm(pb, S[: n // 2])) == util.expected_multiples_msm(S, n // 2)   # prefix: plain path
    # a pinned set with an infinity base and a repeated base
    n = 500
    B = util.multiples_bases(n); B[7] = 0; B[7, 96] = 1; B[9] = B[8]
    S = util.uniform_scalars(n, 4242)
    with M.PinnedBases(B) as pb:
        exp = c.jac_to_int_point(c.msm_g1(B, S, threads=4, variant=1))
        assert c.jac_to_int_point(M.VariableBase.msm(pb, S)) == exp
        pb.precompute()
        assert c.jac_to_int_point(M.VariableBase.msm(pb, S)) == exp


def test_one_shot_msm_keeps_nothing_by_default():
    """Ownership rule of the boundary: without the opt-in the one-shot call retains nothing, so a base array
    rewritten in place at ANY index between calls is seen."""
    n = 1 << 12
    B = util.multiples_bases(n); S = util.uniform_scalars(n, 778)
    os.environ.pop('ALEO_MI355X_SRS_CACHE', None)
    for _ in range(3):
        assert c.jac_to_int_point(M.VariableBase.msm(B, S)) == util.expected_multiples_msm(S, n)
    B[2001] = B[10]                                      # an index the opt-in cache does not sample
    k = (synth.weighted_scalar_sum(S, 1) + (11 - 2002) * synth.limbs_to_int(S[2001])) % p.FR_MODULUS
    assert c.jac_to_int_point(M.VariableBase.msm(B, S)) == p.g1_mul(p.G1_GENERATOR, k)


def test_one_shot_msm_srs_cache_and_prefixes(monkeypatch):
    """With ALEO_MI355X_SRS_CACHE=1 aleo_mi355x_msm_g1 keeps a base array resident between calls (pointer + sampled-content
    key), serves prefixes of it, switches to the fixed-base table on the third use, and notices when sampled contents change."""
    monkeypatch.setenv('ALEO_MI355X_SRS_CACHE', '1')
    n = 1 << 17
    B = util.multiples_bases(n)
    S = util.uniform_scalars(n, 777)
    exp = util.expected_multiples_msm(S, n)
    for _ in range(4):                                   # miss, hit, hit (+ table build), hit on the table path
        assert c.jac_to_int_point(M.VariableBase.msm(B, S)) == exp
    for m in (n // 2, 5000, 1024):                       # prefixes of the cached array
        assert c.jac_to_int_point(M.VariableBase.msm(B[:m], S[:m])) == util.expected_multiples_msm(S, m)
    B[3] = B[10]                                         # same buffer, new contents: the sampled check must reject the entry
    got = c.jac_to_int_point(M.VariableBase.msm(B, S))
    k = (synth.weighted_scalar_sum(S, 1) + (11 - 4) * synth.limbs_to_int(S[3])) % p.FR_MODULUS
    assert got == p.g1_mul(p.G1_GENERATOR, k)


# ---- field-only vector kernels and the hiding commitment ---------------------------------
def test_fr_vector_ops_and_batch_inversion():
    import torch
    from aleo_amd import poly
    for n in (1, 7, 1000, 100003, 1 << 20):
        a = c.fr_to_mont(util.uniform_scalars(n, 31 + n)); b = c.fr_to_mont(util.uniform_scalars(n, 32 + n))
        a[0] = 0
        if n > 5: b[5] = 0; a[3] = c.fr_to_mont(c.ints_to_limbs([p.FR_MODULUS - 1], 4))[0]
        da = torch.from_numpy(a.view(np.int64)).cuda(); db = torch.from_numpy(b.view(np.int64)).cuda(); dd = torch.empty_like(da)
        torch.cuda.synchronize()
        for op in (poly.OP_MUL, poly.OP_ADD, poly.OP_SUB):
            poly.fr_vec_op_device(dd.data_ptr(), da.data_ptr(), db.data_ptr(), n, op)
            got = _sync_to_numpy(dd, n)
            assert (got == c.fr_vec_op(a, b, op)).all(), (n, op)
        poly.fr_vec_op_device(da.data_ptr(), da.data_ptr(), db.data_ptr(), n, poly.OP_MUL)             # in place: dst aliases a
        assert (_sync_to_numpy(da, n) == c.fr_vec_op(a, b, 0)).all()
        inv_in = b.copy()
        dinv = torch.from_numpy(inv_in.view(np.int64)).cuda(); torch.cuda.synchronize()
        poly.batch_inversion_device(dinv.data_ptr(), n)
        assert (_sync_to_numpy(dinv, n) == c.fr_batch_inverse(inv_in)).all(), n


def _sync_to_numpy(t, n):
    """The library works on its own stream: a 1-element MSM call is overkill, so synchronise through the API's own
    blocking entry (fr_mul on one element) before reading the tensor back."""
    M.fr_mul(np.zeros((1, 4), dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64))
    import torch
    torch.cuda.synchronize()
    return t.cpu().numpy().view(np.uint64).reshape(n, 4)


def test_kzg_commit_hiding_matches_oracle():
    n, m = 3000, 17
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as powers, \
         M.PinnedBases.generate_multiples(synth.generator_affine104(), 1000003, m) as gamma:
        coeffs = c.fr_to_mont(util.uniform_scalars(n, 91)); blind = c.fr_to_mont(util.uniform_scalars(m, 92))
        got = c.affine_to_ints(aleo_amd.KZG10.commit_hiding(powers, coeffs, gamma, blind))[0]
        a = c.affine_to_ints(c.kzg_commit(powers.download(), coeffs, threads=4))[0]
        b = c.affine_to_ints(c.kzg_commit(gamma.download(), blind, threads=1))[0]
        assert got == p.g1_add(a, b)


def test_concurrent_callers_share_the_device_safely():
    """snarkVM commits the polynomials of one round from a rayon pool: many host threads call the library at once
    (ctypes releases the GIL).  Results must match the serial ones."""
    import threading
    n = 20000
    B = util.multiples_bases(n)
    jobs = [(util.uniform_scalars(n - 100 * t, 5000 + t), n - 100 * t) for t in range(6)]
    exp = [util.expected_multiples_msm(s, m) for s, m in jobs]
    got = [None] * len(jobs); errs = []

    def work(i):
        try:
            s, m = jobs[i]
            for _ in range(3):
                got[i] = c.jac_to_int_point(M.VariableBase.msm(B[:m], s))
            x = c.fr_to_mont(util.uniform_scalars(1 << 12, 6000 + i))
            assert (aleo_amd.EvaluationDomain(1 << 12).fft(x) == c.ntt_fr(x, 0, 0, 0)).all()
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in ths: t.start()
    for t in ths: t.join()
    assert not errs, errs
    assert got == exp


def test_chunked_host_scalar_calls_from_six_threads_share_the_high_priority_streams():
    """Chunked host-scalar MSMs (the headline call's path: the later chunks run on a borrowed helper's high-priority stream) from six threads at once.  The device
    keeps a POOL of four high-priority streams (api.hip first_use: eight in a row made pipelined proofs crawl), so helpers 0 and 4, 1 and 5 share one: the calls
    must still return their own results, and mixed with three-chunk requests (2^21 points: two helpers each)."""
    import threading
    n = 1 << 21
    pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n).precompute()
    try:
        sizes = [1 << 19, (1 << 19) + 1000, 1 << 20, (1 << 20) - 256, 1 << 21, (1 << 19) + 7]
        S = [util.uniform_scalars(m, 7300 + t) for t, m in enumerate(sizes)]
        exp = [util.expected_multiples_msm(x, len(x)) for x in S]
        errs = []; started = threading.Barrier(len(sizes))

        def work(t):
            try:
                started.wait()
                for _ in range(3):
                    assert c.jac_to_int_point(M.VariableBase.msm(pb, S[t])) == exp[t], t
            except Exception as e:      # noqa: BLE001
                errs.append(e)
        ths = [threading.Thread(target=work, args=(t,)) for t in range(len(sizes))]
        for t in ths: t.start()
        for t in ths: t.join()
        assert not errs, errs
    finally:
        pb.close()


def test_concurrent_slots_share_one_pinned_set():
    """Calls from different threads run on different slots (own stream + workspaces) against ONE pinned set: the table
    build races with running MSMs, and an unpin from another thread must not free the set under a call in flight."""
    import threading, torch
    n = 1 << 16
    pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n)
    S = [util.uniform_scalars(n, 7100 + t) for t in range(4)]
    dS = [torch.from_numpy(x.view(np.int64)).cuda() for x in S]; torch.cuda.synchronize()
    exp = [util.expected_multiples_msm(x, n) for x in S]
    errs = []; started = threading.Barrier(5)

    def work(t):
        try:
            torch.cuda.set_device(0); started.wait()
            for _ in range(6):
                assert c.jac_to_int_point(M.VariableBase.msm_device(pb, dS[t].data_ptr(), n)) == exp[t]
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in ths: t.start()
    started.wait()
    pb.precompute()                      # published while the workers are mid-flight: later calls switch to the table path
    for t in ths: t.join()
    assert not errs, errs
    out = []
    th = threading.Thread(target=lambda: (torch.cuda.set_device(0), out.append(c.jac_to_int_point(M.VariableBase.msm_device(pb, dS[0].data_ptr(), n)))))
    th.start(); pb.close(); th.join()    # the unpin either lands first (clean BAD_HANDLE error in the thread) or the call keeps the set alive
    assert out == [] or out == [exp[0]]
    tm = M.last_msm_timing()             # per calling thread: this thread ran no MSM since precompute
    assert set(tm) >= {'total_ms', 'accum_kernel_ms'}


def test_cpp_host_mirror(tmp_path):
    """The C++ host side above the C ABI (include/aleo_mi355x.hpp), exercised like a snarkVM unit test."""
    import subprocess
    from test_abi import build_cpp_host_mirror
    exe = build_cpp_host_mirror(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'ALL OK' in r.stdout, r.stdout + r.stderr


def test_large_sizes_properties():
    """Beyond the 2^20 / 2^22 headline sizes: a three-pass NTT (2^24) round trip and a 2^23-point fixed-base MSM (the
    per-GPU shard of BASELINE config[4]) against the structured identity."""
    import torch
    lg = 24; n = 1 << lg
    x = util.uniform_scalars(n, 2424)                                     # canonical values double as Montgomery residues
    dx = torch.from_numpy(x.view(np.int64)).cuda(); torch.cuda.synchronize()
    d = aleo_amd.EvaluationDomain(n)
    d.ntt_device(dx.data_ptr(), 0, 0, 1)                                  # coset_fft in HBM
    mid = _sync_to_numpy(dx, n)
    assert not (mid[:1024] == x[:1024]).all()
    d.ntt_device(dx.data_ptr(), 0, 1, 1)                                  # coset_ifft
    assert (_sync_to_numpy(dx, n) == x).all()
    # definition spot check on the forward transform: out[0] = sum_j x[j] (plain fft), via the library's own add kernel? no:
    # python ints on the host (2^24 additions of 256-bit ints, ~10 s) is too slow here; 2^22 covers the definition check.
    n = 1 << 23
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as pb:
        pb.precompute()
        S = util.uniform_scalars(n, 2323)
        dS = torch.from_numpy(S.view(np.int64)).cuda(); torch.cuda.synchronize()
        got = M.VariableBase.msm_device(pb, dS.data_ptr(), n)
        assert c.jac_to_int_point(got) == util.expected_multiples_msm(S, n)           # big-integer k*G (oracle/pyref.g1_mul), not the library's own product


def test_sharded_msm_over_device_contexts_matches_single_device():
    """aleo_mi355x_bases_pin_sharded / _generate_sharded + aleo_mi355x_msm_g1_sharded (BASELINE configs[4] in one process): the set cut
    into G contiguous shards — here every shard on the one visible device, listed G times — gives, for G = 1..5, ragged sizes, a prefix of the set,
    table and plain paths, the bytes of the single-device MSM and the oracle's point; the partials are the shards' own sums; misuse is refused."""
    L = aleo_amd.lib()
    vis, ini = ctypes.c_int32(0), ctypes.c_int32(0)
    aleo_amd._lib.check(L.aleo_mi355x_init(0), 'init'); aleo_amd._lib.check(L.aleo_mi355x_device_count(ctypes.byref(vis), ctypes.byref(ini)), 'device_count')
    assert vis.value >= 1 and ini.value == vis.value and L.aleo_mi355x_init(vis.value + 1) == 2 and L.aleo_mi355x_init(-1) == 2
    n = 5000 + 37
    B = util.multiples_bases(n); S = util.uniform_scalars(n, 9191); S[::7] = 0; S[1::11] = util.uniform_scalars(1, 5)[0]
    want = c.jac_to_int_point(c.msm_g1(B, S, threads=4, variant=1))
    with M.PinnedBases(B) as pb: single = M.VariableBase.msm(pb, S)
    assert c.jac_to_int_point(single) == want
    for G, pre in ((1, False), (2, True), (3, False), (5, True)):
        with aleo_amd.ShardedBases(B, devices=[0] * G, precompute=pre) as sb:
            sh = sb.shards()
            assert len(sh) == G and sh[0][1] == 0 and sum(x[2] for x in sh) == n and all(sh[g][1] + sh[g][2] == sh[g + 1][1] for g in range(G - 1))
            got, part = M.VariableBase.msm_sharded(sb, S, partials=True)
            assert (got == single).all()
            for g, (dev, lo, cnt) in enumerate(sh):
                assert c.jac_to_int_point(part[g]) == c.jac_to_int_point(c.msm_g1(B[lo:lo + cnt], S[lo:lo + cnt], threads=2, variant=1))
            m = n // 3 + 1                                                          # a prefix: the later shards contribute the identity
            assert c.jac_to_int_point(M.VariableBase.msm_sharded(sb, S[:m])) == c.jac_to_int_point(c.msm_g1(B[:m], S[:m], threads=2, variant=1))
            assert c.jac_to_int_point(M.VariableBase.msm_sharded(sb, S[:0])) is None
            out = np.zeros(18, dtype=np.uint64); big = np.zeros((n + 1, 4), dtype=np.uint64)
            assert L.aleo_mi355x_msm_g1_sharded(out.ctypes.data_as(ctypes.c_void_p), sb.handle, big.ctypes.data_as(ctypes.c_void_p), n + 1, None) == 2      # more scalars than points
    with aleo_amd.ShardedBases.generate_multiples(synth.generator_affine104(), 1, 1 << 16, devices=4, precompute=True) as sb:      # P_i = (i + 1) G built shard by shard
        S2 = util.uniform_scalars(1 << 16, 4242)
        assert c.jac_to_int_point(M.VariableBase.msm(sb, S2)) == util.expected_multiples_msm(S2, 1 << 16)
    h = ctypes.c_uint64(0); bad = (ctypes.c_int32 * 2)(0, 99)
    assert L.aleo_mi355x_bases_pin_sharded(B.ctypes.data_as(ctypes.c_void_p), 104, n, bad, 2, 0, ctypes.byref(h)) == 2                 # no such device
    assert L.aleo_mi355x_bases_unpin_sharded(123456) == 4 and L.aleo_mi355x_msm_g1_sharded(out.ctypes.data_as(ctypes.c_void_p), 123456, None, 0, None) == 4


# ---- batched / sharded NTT -------------------------------------------------------------------------------------
def test_ntt_batch_and_grid_scale_match_oracle():
    import torch
    from aleo_amd import dist as adist
    ops = adist.HipLocalOps()
    for lg, batch in ((3, 5), (9, 7), (12, 3), (13, 2)):
        x = c.fr_to_mont(util.uniform_scalars(batch << lg, 8100 + lg))
        for direction in (0, 1):
            t = torch.from_numpy(x.view(np.int64).copy()).cuda()
            ops.batch_ntt(t, lg, batch, direction); torch.cuda.synchronize()
            got = t.cpu().numpy().view(np.uint64).reshape(batch, 1 << lg, 4)
            for b in range(batch):
                assert (got[b] == c.ntt_fr(x[b << lg:(b + 1) << lg], 0, direction, 0)).all(), (lg, b, direction)
    lg_n, rows, cols, row0, col0, ld = 20, 37, 53, 900, 77, 1024            # mode 1 needs (row0 + rows) * ld <= n: g^j is not periodic in n
    x = c.fr_to_mont(util.uniform_scalars(rows * cols, 8200))
    for mode in (0, 1):
        for direction in (0, 1):
            t = torch.from_numpy(x.view(np.int64).copy()).cuda()
            ops.grid_scale(t, lg_n, rows, cols, row0, col0, ld, mode, direction); torch.cuda.synchronize()
            e = torch.from_numpy(x.view(np.int64).copy())
            util.OracleLocalOps().grid_scale(e, lg_n, rows, cols, row0, col0, ld, mode, direction)
            assert (t.cpu() == e).all(), (mode, direction)


@pytest.mark.parametrize('lg_n,world', [(10, 4), (16, 2), (20, 4)])
def test_sharded_ntt_ranks_on_one_gpu(lg_n, world):
    """The 4-step transform of aleo_amd.dist.ShardedDomain with the HIP local steps; the `world` ranks run as threads of this
    process (each call takes its own library slot) and exchange blocks through a barrier instead of RCCL."""
    import threading, torch
    from aleo_amd import dist as adist
    x = c.fr_to_mont(util.uniform_scalars(1 << lg_n, 8300 + lg_n))
    d = aleo_amd.EvaluationDomain(1 << lg_n)
    full = {False: d.fft(x), True: d.coset_fft(x)}
    box = [None] * world; bar = threading.Barrier(world); errs = []

    def exchange(send, rank):
        box[rank] = send; bar.wait()
        recv = torch.stack([box[p][rank] for p in range(world)])
        bar.wait()
        return recv

    def work(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                dom = adist.ShardedDomain(lg_n, rank, world, exchange=exchange)
                idx = dom.evaluation_indices()
                for coset in (False, True):
                    mine = torch.from_numpy(dom.coefficient_shard(x).view(np.int64).copy()).cuda()
                    ev = dom.forward(mine.clone(), coset=coset); torch.cuda.current_stream().synchronize()
                    assert (ev.cpu().numpy().view(np.uint64) == full[coset][idx]).all(), (rank, coset)
                    back = dom.inverse(ev, coset=coset); torch.cuda.current_stream().synchronize()
                    assert (back == mine).all(), (rank, coset)
        except Exception as e:      # noqa: BLE001
            errs.append(e); bar.abort()
    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ths: t.start()
    for t in ths: t.join()
    assert not errs, errs


def test_fr_spmv_matches_oracle():
    """z_M = M z over a ragged CSR matrix: empty rows, one-entry rows, typical short rows and a few very long linear
    combinations (the wave-per-row path), coefficients mostly 1 / -1 / small like an R1CS matrix."""
    import torch
    from aleo_amd import poly
    rows, cols = 20011, 7001
    rng = np.random.default_rng(99)
    lens = rng.choice([0, 1, 2, 3, 5, 9, 64, 65], size=rows, p=[0.05, 0.3, 0.3, 0.2, 0.1, 0.03, 0.01, 0.01]).astype(np.int64)
    lens[[7, 4000, rows - 1]] = [3000, 129, 70000]                                       # long rows, also as the very last row (70000: spread over the grid)
    lens[[11, 12]] = [8192, 8193]                                                        # both sides of the wave-per-row / whole-grid switch
    lens[100:170] = 8200 + np.arange(70)                                                 # more grid-wide rows than the 64 that path takes: the rest fall back
    row_ptr = np.zeros(rows + 1, dtype=np.uint32); row_ptr[1:] = np.cumsum(lens)
    nnz = int(row_ptr[-1])
    col = rng.integers(0, cols, size=nnz, dtype=np.uint32)
    vals_c = util.uniform_scalars(nnz, 9100)
    small = rng.random(nnz) < 0.8
    vals_c[small] = 0; vals_c[small, 0] = rng.integers(1, 5, size=int(small.sum()), dtype=np.uint64)
    neg = rng.random(nnz) < 0.2
    vals_c[neg] = c.ints_to_limbs([p.FR_MODULUS - 1], 4)[0]
    vals = c.fr_to_mont(vals_c); x = c.fr_to_mont(util.uniform_scalars(cols, 9101))
    exp = c.fr_spmv(row_ptr, col, vals, x)
    d = lambda a: torch.from_numpy(a.view(np.int32 if a.dtype == np.uint32 else np.int64).copy()).cuda()
    drp, dcol, dv, dx = d(row_ptr), d(col), d(vals), d(x)
    dy = torch.zeros((rows, 4), dtype=torch.int64, device='cuda'); torch.cuda.synchronize()
    poly.spmv_device(dy.data_ptr(), drp.data_ptr(), dcol.data_ptr(), dv.data_ptr(), dx.data_ptr(), rows); torch.cuda.synchronize()
    assert (dy.cpu().numpy().view(np.uint64) == exp).all()


def test_msm_randomized_prefixes_and_distributions():
    """Differential sweep on one pinned 2^17-point set with its table: random prefix lengths (both sides of the
    table / plain switch at n = 2^(c-3)) x scalar distributions (uniform, witness-like, all-equal, small, top-heavy,
    single non-zero), each against the O(n) structured identity."""
    N = 1 << 17
    rng = np.random.default_rng(20260101)
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, N) as pb:
        pb.precompute()
        lens = [1, 2, 63, 64, 65, (1 << 14) - 1, 1 << 14, (1 << 14) + 1, N - 1, N] + [int(v) for v in rng.integers(1, N, size=14)]
        for i, n in enumerate(lens):
            kind = i % 6
            if kind == 0: S = util.uniform_scalars(n, 31000 + i)
            elif kind == 1: S = util.witness_like_scalars(n, 31000 + i)
            elif kind == 2: S = np.tile(util.uniform_scalars(1, 31000 + i), (n, 1))                       # one bucket per window
            elif kind == 3: S = np.zeros((n, 4), dtype=np.uint64); S[:, 0] = rng.integers(0, 1 << 16, size=n, dtype=np.uint64)
            elif kind == 4:                                                                                # only the top window differs
                S = np.tile(c.ints_to_limbs([p.FR_MODULUS - 1], 4), (n, 1)); S[:, 0] -= rng.integers(0, 4, size=n, dtype=np.uint64)
            else: S = np.zeros((n, 4), dtype=np.uint64); S[n // 2] = util.uniform_scalars(1, 31000 + i)[0]
            got = c.jac_to_int_point(M.VariableBase.msm(pb, S))
            assert got == util.expected_multiples_msm(S, n), (n, kind)


def test_table_path_repeated_opposite_and_infinity_bases():
    """The rare branches of the 28-bit accumulation / pair additions on the TABLE path (n above the table threshold):
    repeated bases with equal scalars (P == acc: doubling), opposite bases with equal scalars (P == -acc: cancellation to
    the identity inside a slice and in the slice tree), bases at infinity, all against the oracle."""
    n = 1 << 14
    B = util.multiples_bases(n)
    rng = np.random.default_rng(77)
    S = util.uniform_scalars(n, 9900)
    neg_y = lambda row: c.affine_from_ints([p.g1_neg(c.affine_to_ints(row.reshape(1, 104))[0])])[0]
    for k in range(0, 4000, 8):                                    # 500 groups: duplicate / opposite / infinity patterns
        kind = (k // 8) % 4
        if kind == 0: B[k + 1] = B[k]; S[k + 1] = S[k]                              # P == acc in the same buckets
        elif kind == 1: B[k + 1] = neg_y(B[k]); S[k + 1] = S[k]                     # P == -acc
        elif kind == 2: B[k + 1] = B[k]; B[k + 2] = B[k]; S[k + 1] = S[k]; S[k + 2] = S[k]      # tripled
        else: B[k + 1] = 0; B[k + 1, 96] = 1                                         # infinity base
    S[5000:5200] = S[5000]; B[5000:5200] = B[5000]                                  # 200 copies of one (point, scalar): a slice of doublings
    exp = c.jac_to_int_point(c.msm_g1(B, S, threads=8, variant=1))
    with M.PinnedBases(B) as pb:
        assert c.jac_to_int_point(M.VariableBase.msm(pb, S)) == exp                 # plain path
        pb.precompute()
        assert c.jac_to_int_point(M.VariableBase.msm(pb, S)) == exp                 # table path, 28-bit arithmetic
        Z = S.copy(); Z[:] = S[0]; Bz = B.copy()
        assert c.jac_to_int_point(M.VariableBase.msm(pb, Z)) == c.jac_to_int_point(c.msm_g1(Bz, Z, threads=8, variant=1))   # all-equal scalars
    # every contribution cancels: pairs (P, s), (P, r - s) -> the identity, through every stage of the table path
    B2 = util.multiples_bases(n); B2[1::2] = B2[0::2]
    S2 = util.uniform_scalars(n, 9901)
    ints = c.limbs_to_ints(S2[0::2]); S2[1::2] = c.ints_to_limbs([(p.FR_MODULUS - v) % p.FR_MODULUS for v in ints], 4)
    with M.PinnedBases(B2) as pb:
        pb.precompute()
        assert c.jac_to_int_point(M.VariableBase.msm(pb, S2)) is None


# ---- batched commitments: one call for the polynomials of a prover round ------------------------------------------------
@pytest.mark.parametrize('k', [1, 3, 4, 12])
def test_batched_commit_matches_oracle_per_polynomial(k):
    """k coefficient vectors against one pinned SRS in ONE call (shared sort / accumulation / reduction, k bucket sets):
    every row must equal the oracle's KZG10::commit of that polynomial, and the single-vector call, bit for bit."""
    import torch
    n = 1 << 12
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as pb:
        pb.precompute()
        B = pb.download()
        polys = [c.fr_to_mont(util.uniform_scalars(n, 12000 + 17 * k + j)) for j in range(k)]
        if k >= 3:
            polys[1] = c.fr_to_mont(util.witness_like_scalars(n, 12500 + k))          # sparse: 60 % zero coefficients
            polys[2] = polys[2][: n - 1234]                                             # ragged: lower degree
        exp = [c.kzg_commit(B[: len(f)], f, threads=8) for f in polys]
        got = aleo_amd.KZG10.commit_batch(pb, polys)
        for j in range(k):
            assert (got[j] == exp[j]).all(), j
            assert (aleo_amd.KZG10.commit(pb, polys[j]) == exp[j]).all(), j
        d = [torch.from_numpy(f.view(np.int64).copy()).cuda() for f in polys]; torch.cuda.synchronize()
        got_d = aleo_amd.KZG10.commit_batch_device(pb, [t.data_ptr() for t in d], [len(f) for f in polys])
        assert (got_d == got).all()


@pytest.mark.parametrize('G', [1, 3, 5, 8])
def test_sharded_commit_entry_points_match_the_single_device_commitments(G):
    """aleo_mi355x_kzg_commit_batch_sharded_device / _segments_sharded_device by themselves (row e2 below the prover): coefficient vectors on the calling
    thread's device, the powers cut into G contiguous shards — shard counts that are no powers of two, so the cuts fall at odd offsets — with and without
    per-shard window tables; ragged, empty, all-zero, one-element vectors and segments that straddle every cut.  Bases (i+1) G: each result against the
    O(n) identity in big integers AND byte-equal to the single-device entry point.  Then the misuse the C ABI must refuse."""
    import torch
    from aleo_amd.kzg import SonicKZG10
    N = (1 << 17) + 321
    lens = [N, 1, 0, 255, (1 << 16) + 3, N // G + 1 if G > 1 else 77, 4096]
    polys = [util.uniform_scalars(max(m, 1), 14000 + 31 * G + i)[:m] for i, m in enumerate(lens)]
    polys[6] = np.zeros((lens[6], 4), dtype=np.uint64)
    mont = [c.fr_to_mont(f) if len(f) else f for f in polys]
    d = [torch.from_numpy(np.ascontiguousarray(f).view(np.int64).copy()).cuda() if len(f) else torch.zeros((1, 4), dtype=torch.int64, device='cuda') for f in mont]
    torch.cuda.synchronize()
    ptrs = [t.data_ptr() for t in d]
    L = aleo_amd.lib()
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, N) as pb:
        pb.precompute()
        single = aleo_amd.KZG10.commit_batch_device(pb, ptrs, lens)
        as_points = c.affine_to_ints(single)
        for i, m in enumerate(lens):
            want = util.expected_multiples_msm(polys[i], m) if m else None
            assert as_points[i] == want, (i, m)
        host = pb.download()
        for pre in (True, False):
            with aleo_amd.ShardedBases(host, devices=[0] * G, precompute=pre) as sb:
                assert len(sb.shards()) == G and sum(cnt for _, _, cnt in sb.shards()) == N
                got = aleo_amd.KZG10.commit_batch_sharded_device(sb, ptrs, lens)
                assert (got == single).all(), (G, pre)
                # segments: every result a sum of pieces placed across the cuts (offsets chosen around each shard boundary)
                cuts = [first for _, first, _ in sb.shards()][1:]
                segs = [(ptrs[0], 1000, 0, 0)]
                for j, cut in enumerate(cuts[:6]): segs.append((ptrs[0] + 32 * 1000 * (j + 1), 777, cut - 300 - j, 1 + j % 2))
                segs.append((ptrs[4], lens[4], N - lens[4], 2)); segs.append((ptrs[3], 0, 5, 0))
                a = SonicKZG10.commit_segments_device(type('CK', (), {'bases': pb})(), segs, 3)
                b = SonicKZG10.commit_segments_sharded_device(sb, segs, 3)
                assert (a == b).all(), (G, pre)
                if pre:
                    out = np.zeros((2, 104), dtype=np.uint8); vp = ctypes.c_void_p
                    pa = (ctypes.c_void_p * 2)(ptrs[0], ptrs[1]); la = (ctypes.c_size_t * 2)(N + 1, 1)
                    assert L.aleo_mi355x_kzg_commit_batch_sharded_device(out.ctypes.data_as(vp), sb.handle, pa, la, 2, None) == 2            # a vector longer than the set
                    la = (ctypes.c_size_t * 2)(10, 1)
                    assert L.aleo_mi355x_kzg_commit_batch_sharded_device(out.ctypes.data_as(vp), 987654321, pa, la, 2, None) == 4           # no such sharded set
                    assert L.aleo_mi355x_kzg_commit_batch_sharded_device(None, sb.handle, pa, la, 2, None) == 2
                    assert L.aleo_mi355x_kzg_commit_batch_sharded_device(out.ctypes.data_as(vp), sb.handle, pa, la, 0, None) == 0           # nothing asked
                    assert L.aleo_mi355x_bases_attach_shards(pb.handle, 987654321, 0) == 4
                    with aleo_amd.ShardedBases(host[: N - 1], devices=[0, 0]) as other:
                        assert L.aleo_mi355x_bases_attach_shards(pb.handle, other.handle, 0) == 2                                           # not the same number of points
                    assert L.aleo_mi355x_bases_attach_shards(123456789, sb.handle, 0) == 4


def test_batched_msm_mixed_tiers_and_degenerate_sets():
    """One batched call whose vectors select different table tiers (c = 13 / 16 / 17), a vector too short for any tier
    (plain path), an empty one, an all-zero one and an all-equal one; more vectors than one launch can hold (chunking).
    Bases (i+1)G: every result against the O(n) structured identity."""
    import torch
    N = 1 << 18
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, N) as pb:
        pb.precompute()
        lens = [1 << 14, 300, 0, (1 << 15) + 77, 1 << 17, (1 << 17) + 5, 5000, N, 1 << 10, 40000] + [3000 + 11 * i for i in range(40)]
        S = [util.uniform_scalars(max(m, 1), 13000 + i)[:m] for i, m in enumerate(lens)]
        S[6] = np.zeros((lens[6], 4), dtype=np.uint64)
        S[8] = np.tile(util.uniform_scalars(1, 13999), (lens[8], 1))
        S[9] = util.witness_like_scalars(lens[9], 13998)
        d = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64).copy()).cuda() if len(x) else torch.zeros((1, 4), dtype=torch.int64, device='cuda') for x in S]
        torch.cuda.synchronize()
        got = M.VariableBase.msm_batch_device(pb, [t.data_ptr() for t in d], lens)
        for i, m in enumerate(lens):
            exp = util.expected_multiples_msm(S[i], m) if m else None
            assert c.jac_to_int_point(got[i]) == exp, (i, m)
            if i < 10 and m: assert (M.VariableBase.msm_device(pb, d[i].data_ptr(), m) == got[i]).all(), (i, m)
    # no table: the batch degrades to one MSM per vector
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, 5000) as pb:
        lens = [5000, 1234, 1]
        S = [util.uniform_scalars(m, 13100 + i) for i, m in enumerate(lens)]
        d = [torch.from_numpy(x.view(np.int64).copy()).cuda() for x in S]; torch.cuda.synchronize()
        got = M.VariableBase.msm_batch_device(pb, [t.data_ptr() for t in d], lens)
        for i, m in enumerate(lens): assert c.jac_to_int_point(got[i]) == util.expected_multiples_msm(S[i], m)


def test_big_batched_request_runs_as_a_pipeline_of_launch_chains():
    """A request of >= 2^20 points in several launch chains (different table tiers, more results than one chain holds) goes through run_chains' pipeline:
    chain i + 1 is sorted and chain i - 1 reduced beside the accumulation of chain i, on two contexts.  Results of every tier, an empty vector, an all-zero
    one, an all-equal one and witness-like ones against the O(n) identity; the same request twice (the contexts are reused); and a request with a
    tier-less member, which must take the two-thread path and still be right."""
    import torch
    N = 1 << 20
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, N) as pb:
        pb.precompute()
        lens = [N, (1 << 19) + 7, 1 << 17, (1 << 17) + 64, 0, 1 << 15, 3 << 18, N - 1, 40000, 1 << 20, 5000, (1 << 16) + 1]
        S = [util.uniform_scalars(max(m, 1), 15000 + i)[:m] for i, m in enumerate(lens)]
        S[2] = np.zeros((lens[2], 4), dtype=np.uint64)
        S[5] = np.tile(util.uniform_scalars(1, 15999), (lens[5], 1))
        S[6] = util.witness_like_scalars(lens[6], 15998); S[9] = util.witness_like_scalars(lens[9], 15997)
        d = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64).copy()).cuda() if len(x) else torch.zeros((1, 4), dtype=torch.int64, device='cuda') for x in S]
        torch.cuda.synchronize()
        exp = [util.expected_multiples_msm(S[i], m) if m else None for i, m in enumerate(lens)]
        for rep in range(2):
            got = M.VariableBase.msm_batch_device(pb, [t.data_ptr() for t in d], lens)
            for i, m in enumerate(lens): assert c.jac_to_int_point(got[i]) == exp[i], (rep, i, m)
        # the same pipeline on the narrow-window range table (sparse hint; witness-like and small scalars, every segment inside the range)
        from aleo_amd.kzg import SonicKZG10
        pb.precompute_range(0, N, 16)
        lens3 = [N, 3 << 18, (1 << 19) + 5, N - 7, 1 << 20, 1 << 18, 12345]
        S3 = [util.witness_like_scalars(m, 15600 + i) for i, m in enumerate(lens3)]
        S3[2] = np.zeros((lens3[2], 4), dtype=np.uint64); S3[2][:, 0] = np.arange(lens3[2], dtype=np.uint64) % 65536
        m3 = [c.fr_to_mont(x) for x in S3]
        d3 = [torch.from_numpy(x.view(np.int64).copy()).cuda() for x in m3]; torch.cuda.synchronize()
        segs3 = [(t.data_ptr(), m, 0, i) for i, (t, m) in enumerate(zip(d3, lens3))]
        class _CK: bases = pb
        dense = SonicKZG10.commit_segments_device(_CK, segs3, len(lens3))
        sparse = SonicKZG10.commit_segments_device(_CK, segs3, len(lens3), sparse=True)
        assert (dense == sparse).all()
        pts3 = c.affine_to_ints(sparse)
        for i, m in enumerate(lens3): assert pts3[i] == util.expected_multiples_msm(S3[i], m), (i, m)
        lens2 = lens[:4] + [300]                                                        # 300 points: below every tier -> the plain path -> not a pipeline
        S2 = S[:4] + [util.uniform_scalars(300, 15500)]
        d2 = d[:4] + [torch.from_numpy(S2[4].view(np.int64).copy()).cuda()]; torch.cuda.synchronize()
        got = M.VariableBase.msm_batch_device(pb, [t.data_ptr() for t in d2], lens2)
        for i, m in enumerate(lens2): assert c.jac_to_int_point(got[i]) == util.expected_multiples_msm(S2[i], m), (i, m)


def test_async_device_calls_order_their_scratch_across_streams(tmp_path):
    """The *_device transforms enqueue on the caller's stream and return at once; with ONE slot (ALEO_MI355X_SLOTS=1) two
    threads on two streams keep handing the same scratch buffer to each other while the previous user's kernels are still
    in flight.  Batched 2^20-element NTTs, batch inversions and an NTT -> commit chain with stream == NULL, all against the
    oracle (tests/helpers/scratch_race_check.py, run in a child process because the slot count is read once at init)."""
    import subprocess, sys
    env = dict(os.environ, ALEO_MI355X_SLOTS='1')
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers', 'scratch_race_check.py')], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and 'SCRATCH OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_sharded_ntt_over_device_contexts_matches_single_device():
    """aleo_mi355x_ntt_fr_sharded (the 4-step transform with one exchange, one process, G device contexts — here the one visible card listed
    G times): fft / ifft / coset_fft / coset_ifft on a host buffer, natural order in and out, against the oracle at small sizes and against the
    single-device transform up to 2^20, G = 1, 2, 4, 8; the transpose kernel on ragged shapes; misuse refused."""
    import torch
    L = aleo_amd.lib()
    for lg, Gs in ((2, (1, 2)), (3, (1, 2)), (6, (1, 2, 4, 8)), (9, (2, 4)), (13, (1, 4, 8)), (16, (2, 8))):
        x = c.fr_to_mont(util.uniform_scalars(1 << lg, 7000 + lg)); d = aleo_amd.EvaluationDomain(1 << lg)
        for G_ in Gs:
            for direction in (0, 1):
                for typ in (0, 1):
                    y = x.copy(); d.ntt_sharded_in_place(y, [0] * G_, direction, typ)
                    assert (y == c.ntt_fr(x, 0, direction, typ)).all(), (lg, G_, direction, typ)
    x = c.fr_to_mont(util.uniform_scalars(1 << 20, 7020)); d = aleo_amd.EvaluationDomain(1 << 20)
    want = d.coset_fft(x)
    y = x.copy(); d.ntt_sharded_in_place(y, 4, 0, 1); assert (y == want).all()            # devices given as a count: device g mod visible
    d.ntt_sharded_in_place(y, [0, 0], 1, 1); assert (y == x).all()                          # coset_ifft brings it back
    for rows, cols in ((1, 1), (3, 5), (32, 32), (33, 31), (100, 7), (1, 64)):              # dst[c][r] = src[r][c]
        a = util.uniform_scalars(rows * cols, 9).reshape(rows, cols, 4)
        src = torch.from_numpy(a.view(np.int64).copy()).cuda(); dst = torch.zeros((cols, rows, 4), dtype=torch.int64, device='cuda'); torch.cuda.synchronize()
        aleo_amd._lib.check(L.aleo_mi355x_fr_transpose_device(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), rows, cols, None), 'fr_transpose_device')
        assert (dst.cpu().numpy().view(np.uint64) == a.transpose(1, 0, 2)).all(), (rows, cols)
    buf = x[:64].copy()
    assert L.aleo_mi355x_ntt_fr_sharded(buf.ctypes.data_as(ctypes.c_void_p), 6, 0, 0, None, 3) == 2          # not a power of two
    assert L.aleo_mi355x_ntt_fr_sharded(buf.ctypes.data_as(ctypes.c_void_p), 6, 0, 0, None, 16) == 2         # more shards than rows
    assert L.aleo_mi355x_ntt_fr_sharded(None, 6, 0, 0, None, 2) == 2


def test_bench_spawns_two_ranks_on_one_card_without_a_launcher():
    """`python bench.py --gpus 2 --backend gloo ...` started plainly (no torchrun, no WORLD_SIZE): the launcher half spawns both ranks, each runs the real
    pinned MSM step on the one visible card, the 144-byte partials are all-gathered (gloo here: RCCL refuses two ranks on one device) and the result
    gate (k G in big integers) passes; one JSON line with n_gpus = 2 comes back.  Also init(0): peer access is tried between all initialised devices
    (none to pair on one card: 0 enabled, 0 refused) and the call stays idempotent."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--lg-n', '12', '--steps', '2', '--warmup', '1', '--no-variants',
                        '--no-cpu-baseline', '--varuna-lg', '0', '--sharded-ntt-lg', '0', '--config4-lg', '14'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks'] == 2 and out['rccl_ranks'] == 0 and out['backend'] == 'gloo' and out['value'] > 0
    c4 = out['config4_2^14']                                   # BASELINE configs[4] (here 2^14 points) split over the two ranks, in the same line
    assert c4['result_is_k_times_generator'] is True and c4['points_per_rank'] == 1 << 13 and c4['scaling'] == 'strong' and len(c4['per_rank_msm_ms']) == 2 and c4['scalar_muls_per_s'] > 0
    L = aleo_amd.lib()
    on, off = ctypes.c_int32(-1), ctypes.c_int32(-1)
    assert L.aleo_mi355x_init(0) == 0 and L.aleo_mi355x_init(0) == 0 and L.aleo_mi355x_peer_info(ctypes.byref(on), ctypes.byref(off)) == 0
    import torch
    g = torch.cuda.device_count()
    assert on.value + off.value == g * (g - 1)


def test_rccl_runs_the_exchange_with_one_rank():
    """backend='nccl' is RCCL on ROCm.  A one-GPU box cannot hold two RCCL ranks (one device per rank), so until round 4 every rehearsal of the N > 1 path ran
    under gloo and RCCL itself had never executed.  This runs the real library with a world of one rank: the all-gather of the 144-byte partial, the MAX
    all-reduce of the timing and the all-to-all of the sharded transform (ShardedDomain(always_collective=True): all_to_all_single is called, counted, and the
    values compared with the CPU restatement) are RCCL launches on the card (tests/helpers/rccl_one_rank.py, a child process: the process group must not leak
    into this one)."""
    import subprocess, sys, socket
    s_ = socket.socket(); s_.bind(('127.0.0.1', 0)); port = s_.getsockname()[1]; s_.close()
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers', 'rccl_one_rank.py'), str(port)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert out == {'backend': 'nccl', 'world': 1, 'gathered_equals_partial': True, 'sum_equals_partial': True, 'all_reduce_max': 3.25, 'sharded_ntt_matches_oracle': True,
                   'all_to_all_single_calls': 4}, out      # 4: forward and inverse, plain and coset — each one torch.distributed.all_to_all_single under 'nccl'


def test_config4_full_size_as_eight_shards_in_one_process():
    """BASELINE configs[4] at its full size through the multi-device entry of the C ABI: 2^26 points, P_i = (i + 1) G generated shard by shard, as
    EIGHT shards of 2^23 points with their fixed-base tables — all eight on the one visible card (no 8-GPU node here: the split, the per-shard
    Pippenger, the host merge and the sizes are the real ones, the device is shared) — against the structured identity in big integers."""
    n = 1 << 26
    with aleo_amd.ShardedBases.generate_multiples(synth.generator_affine104(), 1, n, devices=[0] * 8, precompute=True) as sb:
        sh = sb.shards()
        assert len(sh) == 8 and all(cnt == 1 << 23 for _, _, cnt in sh)
        S = util.uniform_scalars(n, 2626)
        got, part = M.VariableBase.msm_sharded(sb, S, partials=True)
        assert c.jac_to_int_point(got) == util.expected_multiples_msm(S, n)
        lo = 3 << 23                                                              # one partial against its own identity: sum_i s_i (lo + i + 1) G over shard 3
        assert c.jac_to_int_point(part[3]) == p.g1_mul(p.G1_GENERATOR, synth.weighted_scalar_sum(S[lo:lo + (1 << 23)], lo + 1))


@pytest.mark.parametrize('mode', ['null', 'stream'])
def test_first_call_of_a_process_is_a_batched_transform(mode):
    """Regression for the host segfault of round 2 (gpurun_out/r02_t1.log: inside aleo_mi355x_ntt_fr_batch_device while the stream-ordered slot
    scratch was being introduced): a fresh process with ONE slot whose first library call is a batched transform on stream NULL / on a created
    stream (tests/helpers/first_call_check.py) — the first-use path of the slot's stream, events and scratch."""
    import subprocess, sys
    env = dict(os.environ, ALEO_MI355X_SLOTS='1')
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers', 'first_call_check.py'), mode], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and 'FIRST CALL OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ---- reference-held data through the HIP path ------------------------------------------------------------------------------
def _reference_g1_points():
    """The twelve G1 points of the reference's own proof string (wasm/src/programs/transaction.rs:100): ten commitments and
    two KZG opening witnesses, decompressed by the PRODUCT's wire code from the fixture bytes."""
    from aleo_amd import wire
    fx = json.load(open(os.path.join(G, 'reference_proof.json')))
    comp = [bytes.fromhex(v['compressed']) for v in fx['commitments'].values()] + [bytes.fromhex(o['compressed']) for o in fx['openings']]
    ints = [(int(v['x'], 16), int(v['y'], 16)) for v in fx['commitments'].values()] + [(int(o['x'], 16), int(o['y'], 16)) for o in fx['openings']]
    aff = wire.g1_decompress(np.stack([np.frombuffer(b, dtype=np.uint8) for b in comp]))
    assert c.affine_to_ints(aff) == ints
    return aff, ints, comp


def test_reference_points_as_msm_bases():
    """MSMs whose BASES are the reference-held points (not multiples of the generator the oracle made up): (r-1) P == -P,
    single-scalar products against big-integer double-and-add, a 12-point MSM against the oracle, and a 4800-point MSM over
    repeated reference points against sum_j (sum of its scalars) * P_j — plain path and fixed-base table path."""
    from aleo_amd import wire
    aff, ints, comp = _reference_g1_points()
    rm1 = c.ints_to_limbs([p.FR_MODULUS - 1], 4)
    for j in range(12):
        got = c.jac_to_int_point(M.VariableBase.msm(aff[j:j + 1], rm1))
        assert got == p.g1_neg(ints[j]), j
    S = util.uniform_scalars(12, 777001)
    sv = c.limbs_to_ints(S)
    for j in (0, 5, 11):
        assert c.jac_to_int_point(M.VariableBase.msm(aff[j:j + 1], S[j:j + 1])) == p.g1_mul(ints[j], sv[j])
    exp12 = None
    for j in range(12): exp12 = p.g1_add(exp12, p.g1_mul(ints[j], sv[j]))
    assert c.jac_to_int_point(M.VariableBase.msm(aff, S)) == exp12
    assert c.jac_to_int_point(c.msm_g1(aff, S, threads=4, variant=1)) == exp12
    n = 4800
    B = np.tile(aff, (n // 12, 1)); S = util.uniform_scalars(n, 777002); S[7::97] = 0; S[11::101, 1:] = 0
    sv = c.limbs_to_ints(S)
    exp = None
    for j in range(12): exp = p.g1_add(exp, p.g1_mul(ints[j], sum(sv[j::12]) % p.FR_MODULUS))
    assert c.jac_to_int_point(c.msm_g1(B, S, threads=8, variant=1)) == exp
    with M.PinnedBases(B) as pb:
        plain = M.VariableBase.msm(pb, S)
        assert c.jac_to_int_point(plain) == exp
        pb.precompute()                                                           # tables of 2^(13 w) * P for the reference points
        assert (M.VariableBase.msm(pb, S) == plain).all()
        # KZG10::commit shape over these bases, result through the product's compressed serialisation and back
        coeffs = c.fr_to_mont(S)
        cm = aleo_amd.KZG10.commit(pb, coeffs)
        assert c.affine_to_ints(cm.reshape(1, 104))[0] == exp
        cc = wire.g1_compress(cm.reshape(1, 104))
        assert cc.tobytes() == p.g1_compress(exp)
        assert (wire.g1_decompress(cc) == cm.reshape(1, 104)).all()
        # batched: three polynomials over the same reference-point SRS
        polys = [coeffs, c.fr_to_mont(util.uniform_scalars(n, 777003)), coeffs[:2000]]
        got = aleo_amd.KZG10.commit_batch(pb, polys)
        for q, f in enumerate(polys): assert (got[q] == c.kzg_commit(B[: len(f)], f, threads=8)).all(), q


def test_reference_field_elements_through_the_device_arithmetic():
    """The eight Fr values and random_v of the reference's proof, its `…field` / `…group` literals: products and batch inversion
    on the device against big integers (Montgomery form in and out)."""
    import torch
    from aleo_amd import poly
    fx = json.load(open(os.path.join(G, 'reference_proof.json'))); lit = json.load(open(os.path.join(G, 'reference_literals.json')))
    vals = [int(v, 16) for v in fx['field_elements']['evaluations'] + fx['field_elements']['sums']] + [int(fx['openings'][0]['random_v'], 16)]
    vals += [int(f['value']) for f in lit['fields']] + [int(g['x']) for g in lit['groups']]
    r = p.FR_MODULUS
    a = c.fr_to_mont(c.ints_to_limbs(vals, 4)); b = c.fr_to_mont(c.ints_to_limbs(vals[::-1], 4))
    assert c.limbs_to_ints(c.fr_from_mont(M.fr_mul(a, b))) == [x * y % r for x, y in zip(vals, vals[::-1])]
    d = torch.from_numpy(a.view(np.int64).copy()).cuda(); torch.cuda.synchronize()
    poly.batch_inversion_device(d.data_ptr(), len(vals))
    assert c.limbs_to_ints(c.fr_from_mont(_sync_to_numpy(d, len(vals)))) == [pow(x, -1, r) for x in vals]
    # Edwards relation of the reference's group literals on the device: (1 + x^2) == y^2 (1 - 3021 x^2)
    xs = [int(g['x']) for g in lit['groups']]; ys = [int(g['y'], 16) for g in lit['groups']]
    X = c.fr_to_mont(c.ints_to_limbs(xs, 4)); Y = c.fr_to_mont(c.ints_to_limbs(ys, 4))
    D = c.fr_to_mont(c.ints_to_limbs([3021] * len(xs), 4)); one = c.fr_to_mont(c.ints_to_limbs([1] * len(xs), 4))
    XX, YY = M.fr_mul(X, X), M.fr_mul(Y, Y)
    lhs = c.fr_vec_op(one, XX, 1); rhs = M.fr_mul(YY, c.fr_vec_op(one, M.fr_mul(D, XX), 2))
    assert (lhs == rhs).all()


def test_config2_full_size_chain_known_polynomial():
    """BASELINE configs[2] at full size, checked (not only timed): 2^22 coefficients -> fft -> ifft in HBM -> kzg_commit_device
    over bases (i+1)G must give (sum_i c_i (i+1)) G; the evaluations in between must differ from the coefficients."""
    import torch
    lg = 22; n = 1 << lg
    coeffs = util.uniform_scalars(n, 0xA1E00003)                               # canonical values < r, read as Montgomery residues
    canon = c.fr_from_mont(coeffs[: 1 << 12])                                  # spot check of the convention on a prefix
    d = aleo_amd.EvaluationDomain(n)
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as pb:
        pb.precompute()
        buf = torch.from_numpy(coeffs.view(np.int64).copy()).cuda(); torch.cuda.synchronize()
        d.ntt_device(buf.data_ptr(), 0, 0, 0)                                  # evaluations
        mid = _sync_to_numpy(buf, n)
        assert not (mid[:4096] == coeffs[:4096]).all()
        d.ntt_device(buf.data_ptr(), 0, 1, 0)                                  # back to coefficients, never leaving HBM
        cm = aleo_amd.KZG10.commit_device(pb, buf.data_ptr(), n)
        # discrete log of the expected commitment: sum_i from_mont(c_i) * (i+1); from_mont is linear: R^-1 * sum_i c_i (i+1)
        k = synth.weighted_scalar_sum(coeffs, 1) * pow(1 << 256, -1, p.FR_MODULUS) % p.FR_MODULUS
        assert c.limbs_to_ints(canon)[:3] == [v * pow(1 << 256, -1, p.FR_MODULUS) % p.FR_MODULUS for v in c.limbs_to_ints(coeffs[:3])]
        kG = p.g1_mul(p.G1_GENERATOR, k)                                          # big-integer double-and-add: independent of the HIP path
        assert c.affine_to_ints(cm.reshape(1, 104))[0] == kG and cm[96] == 0


# ---- KZG10 opening: witness polynomial on the device -------------------------------------------------------------------------
@pytest.mark.parametrize('n', [1, 2, 15, 16, 17, 4095, 4096, 4097, 100003, (1 << 20) + 5, 1 << 22])
def test_divide_by_linear_matches_oracle(n):
    """(p(X) - p(z)) / (X - z) by the three-level suffix scan against the oracle's serial synthetic division: every quotient
    coefficient and p(z), bit for bit; chunk / block boundaries (16, 4096) on both sides; z in {random, 0, 1}."""
    import torch
    from aleo_amd import poly
    f = c.fr_to_mont(util.uniform_scalars(n, 14000 + n % 1000))
    d = torch.from_numpy(f.view(np.int64).copy()).cuda()
    q = torch.zeros((max(n - 1, 1), 4), dtype=torch.int64, device='cuda'); ev = torch.zeros(4, dtype=torch.int64, device='cuda')
    zs = [c.fr_to_mont(util.uniform_scalars(1, 14500 + n % 1000))[0]]
    if n <= 100003: zs += [np.zeros(4, dtype=np.uint64), c.fr_to_mont(c.ints_to_limbs([1], 4))[0]]
    for z in zs:
        torch.cuda.synchronize()
        poly.divide_by_linear_device(q.data_ptr(), ev.data_ptr(), d.data_ptr(), n, z)
        torch.cuda.synchronize()
        eq, ee = c.fr_divide_by_linear(f, z)
        assert (ev.cpu().numpy().view(np.uint64) == ee).all(), n
        if n > 1: assert (q.cpu().numpy().view(np.uint64)[: n - 1] == eq).all(), n


def test_kzg_open_matches_oracle():
    """KZG10::open shape: witness polynomial on the device, then its commitment — against oracle division + oracle commit."""
    import torch
    n = 5000
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n) as pb:
        B = pb.download()
        f = c.fr_to_mont(util.uniform_scalars(n, 15001)); z = c.fr_to_mont(util.uniform_scalars(1, 15002))[0]
        d = torch.from_numpy(f.view(np.int64).copy()).cuda(); torch.cuda.synchronize()
        eq, ee = c.fr_divide_by_linear(f, z)
        for _ in range(2):
            w, ev = aleo_amd.KZG10.open_device(pb, d.data_ptr(), n, z)
            assert (ev == ee).all()
            assert (w == c.kzg_commit(B[: n - 1], eq, threads=8)).all()
            pb.precompute()


# ---- G2 MSM ------------------------------------------------------------------------------------------------
def _g2pt(v): return None if v is None else ((int(v[0][0], 16), int(v[0][1], 16)), (int(v[1][0], 16), int(v[1][1], 16)))


def _g2_multiples(n):
    return c.g2_multiples(c.g2_affine_from_ints([p.G2_GENERATOR])[0], n)


def test_msm_g2_golden_vectors():
    fx = json.load(open(os.path.join(G, 'msm_g2_small.json')))
    mult = _g2_multiples(200)
    for case in fx['cases']:
        B = mult[np.array(case['base_multipliers']) - 1]; S = c.ints_to_limbs(_ints(case['scalars']), 4)
        assert c.g2_jac_to_int_point(M.msm_g2(B, S)) == _g2pt(case['result']), (case['n'], case['kind'])


@pytest.mark.parametrize('n', [1, 2, 3, 31, 32, 33, 255, 1000, 4097, 1 << 14])
def test_msm_g2_matches_oracle(n):
    B = _g2_multiples(n)
    for kind in ('uniform', 'witness'):
        S = util.uniform_scalars(n, 16000 + n) if kind == 'uniform' else util.witness_like_scalars(n, 16500 + n)
        got = M.msm_g2(B, S)
        k = synth.weighted_scalar_sum(S, 1)
        assert c.g2_jac_to_int_point(got) == p.g2_mul(p.G2_GENERATOR, k), (n, kind)          # structured identity in big integers
        if n <= 4097: assert c.g2_jac_to_int_point(c.msm_g2(B, S)) == c.g2_jac_to_int_point(got)      # and the oracle's standard::msm
        assert (M.msm_g2(np.ascontiguousarray(B[:, :192]), S) == got).all()                           # stride 192


def test_msm_g2_edge_cases():
    n = 300
    B = _g2_multiples(n)
    assert c.g2_jac_to_int_point(M.msm_g2(B[:0], np.zeros((0, 4), dtype=np.uint64))) is None              # empty
    assert c.g2_jac_to_int_point(M.msm_g2(B, np.zeros((n, 4), dtype=np.uint64))) is None                   # all zero
    ones = np.zeros((n, 4), dtype=np.uint64); ones[:, 0] = 1
    assert c.g2_jac_to_int_point(M.msm_g2(B, ones)) == p.g2_mul(p.G2_GENERATOR, n * (n + 1) // 2)
    rm1 = np.tile(c.ints_to_limbs([p.FR_MODULUS - 1], 4), (n, 1))
    assert c.g2_jac_to_int_point(M.msm_g2(B, rm1)) == p.g2_neg(p.g2_mul(p.G2_GENERATOR, n * (n + 1) // 2))
    same = np.repeat(B[:1], n, axis=0); S = util.uniform_scalars(n, 16900)                                  # one base repeated: doublings
    assert c.g2_jac_to_int_point(M.msm_g2(same, S)) == p.g2_mul(p.G2_GENERATOR, sum(c.limbs_to_ints(S)) % p.FR_MODULUS)
    eq = np.tile(util.uniform_scalars(1, 16901), (n, 1))                                                    # one bucket per window
    assert c.g2_jac_to_int_point(M.msm_g2(B, eq)) == p.g2_mul(p.G2_GENERATOR, synth.weighted_scalar_sum(eq, 1))
    neg = c.g2_affine_from_ints([p.g2_neg(p.G2_GENERATOR)])
    pm = np.concatenate([B[:1], neg] * 40, axis=0); s77 = np.zeros((80, 4), dtype=np.uint64); s77[:, 0] = 77  # P, -P cancel
    assert c.g2_jac_to_int_point(M.msm_g2(pm, s77)) is None
    Binf = B.copy(); Binf[5] = 0; Binf[5, 192] = 1; Binf[77] = 0; Binf[77, 192] = 1                          # infinity bases are skipped
    S = util.uniform_scalars(n, 16902)
    assert c.g2_jac_to_int_point(M.msm_g2(Binf, S)) == c.g2_jac_to_int_point(c.msm_g2(Binf, S))
    parts = np.stack([M.msm_g2(B[:100], S[:100]), M.msm_g2(B[100:], S[100:]), M.msm_g2(B[:0], S[:0])])      # shard + g2_sum (identity included)
    assert (M.g2_sum(parts) == M.msm_g2(B, S)).all()


def test_msm_g2_pinned_set_equals_the_one_shot_call():
    """aleo_mi355x_bases_g2_pin / _msm_g2_pinned: the resident set (rows, infinity flags, 28-bit rows) gives the one-shot call's bytes — on every prefix,
    with both row strides, with infinity bases in the set, for the empty and the all-zero request; more scalars than pinned bases and unknown or
    released handles are refused."""
    n = 5000
    B = _g2_multiples(n); B[17] = 0; B[17, 192] = 1; B[4096] = 0; B[4096, 192] = 1          # two infinity bases
    S = util.uniform_scalars(n, 16950); Wt = util.witness_like_scalars(n, 16951)
    with M.PinnedG2Bases(B) as pg:
        for m in (n, 4097, 4096, 1000, 33, 18, 17, 1, 0):
            got = pg.msm(S[:m])
            assert (got == M.msm_g2(B[:m], S[:m])).all(), m
            if m in (1000, 33): assert c.g2_jac_to_int_point(got) == c.g2_jac_to_int_point(c.msm_g2(B[:m], S[:m])), m      # and the oracle
        assert (pg.msm(Wt) == M.msm_g2(B, Wt)).all()
        assert c.g2_jac_to_int_point(pg.msm(np.zeros((n, 4), dtype=np.uint64))) is None
        assert (pg.msm(S) == pg.msm(S)).all()                                                  # the set is not consumed
        out = np.zeros(36, dtype=np.uint64); big = util.uniform_scalars(n + 1, 16952)
        assert aleo_amd.lib().aleo_mi355x_msm_g2_pinned(out.ctypes.data_as(ctypes.c_void_p), pg.handle, big.ctypes.data_as(ctypes.c_void_p), n + 1) == 2
        assert aleo_amd.lib().aleo_mi355x_msm_g2_pinned(None, pg.handle, big.ctypes.data_as(ctypes.c_void_p), 4) == 2
        h = pg.handle
    assert aleo_amd.lib().aleo_mi355x_msm_g2_pinned(out.ctypes.data_as(ctypes.c_void_p), h, S.ctypes.data_as(ctypes.c_void_p), 4) != 0       # released
    assert aleo_amd.lib().aleo_mi355x_bases_g2_unpin(h) != 0
    C = _g2_multiples(300)
    with M.PinnedG2Bases(np.ascontiguousarray(C[:, :192])) as pg:                              # stride 192: no flags
        assert c.g2_jac_to_int_point(pg.msm(S[:300])) == p.g2_mul(p.G2_GENERATOR, synth.weighted_scalar_sum(S[:300], 1))
    hh = ctypes.c_uint64(0)
    assert aleo_amd.lib().aleo_mi355x_bases_g2_pin(C.ctypes.data_as(ctypes.c_void_p), 104, 300, ctypes.byref(hh)) == 2                          # bad stride


def test_g2_lane_pair_arithmetic_matches_the_one_lane_code():
    """The lane-pair Fq2 arithmetic of the G2 path (components across two lanes, 28-bit limbs: full addition, doubling, the same-point case of the addition,
    the mixed addition of the accumulation loop) against the one-lane 32-bit code on chains over real curve points, every intermediate compared as a
    group element.  (Round 4's first version exchanged the lanes' zero flags behind a short-circuit `&&`: the lane whose flag was false skipped the
    exchange and its partner read an inactive lane — every affine operand, whose ZZ is (1, 0), looked like the identity to one lane of the pair.)"""
    B = _g2_multiples(200)
    rows = np.ascontiguousarray(B[:, :192]); f = (ctypes.c_uint32 * 2)()
    assert aleo_amd.lib().aleo_mi355x_selftest_g2pair(rows.ctypes.data_as(ctypes.c_void_p), 200, 198, f) == 0
    assert (f[0], f[1]) == (0, 0), 'pairs that disagreed: %d, failing steps mask %d' % (f[0], f[1])
    same = np.ascontiguousarray(np.repeat(rows[:1], 8, axis=0))            # P_i = P_j = P_k: every addition of the chain is a doubling or meets equal operands
    assert aleo_amd.lib().aleo_mi355x_selftest_g2pair(same.ctypes.data_as(ctypes.c_void_p), 8, 4, f) == 0
    assert (f[0], f[1]) == (0, 0)
    assert aleo_amd.lib().aleo_mi355x_selftest_g2pair(None, 8, 4, f) == 2


def test_msm_g2_2_16_structured_identity():
    n = 1 << 16
    B = _g2_multiples(n); S = util.uniform_scalars(n, 16999)
    assert c.g2_jac_to_int_point(M.msm_g2(B, S)) == p.g2_mul(p.G2_GENERATOR, synth.weighted_scalar_sum(S, 1))


# ---- SonicKZG10::commit shape: degree bounds (shifted powers) and hiding (gamma powers) as segments of one launch chain ------------
def test_sonic_commit_degree_bounds_and_hiding_match_oracle():
    """Labelled polynomials of one round with and without degree bounds and hiding bounds, all in ONE call over one pinned
    (powers | gamma powers) set.  Expected per polynomial: oracle KZG10::commit over the bases its segment addresses
    (powers[max_degree - bound ..] for a bounded polynomial) plus, when hiding, the oracle commitment of the blinding polynomial over the
    gamma powers — added as group elements in big integers."""
    import torch
    N = 1 << 13; m = 8
    gen = synth.generator_affine104()
    with M.PinnedBases.generate_multiples(gen, 1, N) as tmp: powers = tmp.download()
    with M.PinnedBases.generate_multiples(gen, 7_000_003, m) as tmp: gamma = tmp.download()
    for precompute in (False, True):
        with aleo_amd.CommitterKey(powers, gamma, precompute=precompute) as ck:
            assert ck.max_degree == N - 1 and ck.gamma_offset == N
            f = lambda n, seed: c.fr_to_mont(util.uniform_scalars(n, seed))
            polys = [(f(N, 17001), None, None),                               # plain, full degree
                     (f(5000, 17002), None, f(3, 17003)),                      # hiding
                     (f(4096, 17004), 4095, None),                             # degree bound = its degree: shifted powers
                     (f(3000, 17005), 6000, f(m, 17006)),                      # degree bound above the degree + hiding
                     (c.fr_to_mont(util.witness_like_scalars(N, 17007)), None, f(1, 17008)),
                     (f(1, 17009), 0, None)]                                   # constant polynomial at the very last power
            got = aleo_amd.SonicKZG10.commit(ck, polys)
            for q, (co, bound, blind) in enumerate(polys):
                off = 0 if bound is None else ck.max_degree - bound
                exp = c.affine_to_ints(c.kzg_commit(powers[off: off + len(co)], co, threads=8))[0]
                if blind is not None: exp = p.g1_add(exp, c.affine_to_ints(c.kzg_commit(gamma[: len(blind)], blind, threads=1))[0])
                assert c.affine_to_ints(got[q].reshape(1, 104))[0] == exp, (precompute, q)
            # device-resident coefficients: same commitments
            dev = [(torch.from_numpy(co.view(np.int64).copy()).cuda(), None if bl is None else torch.from_numpy(bl.view(np.int64).copy()).cuda()) for co, _, bl in polys]
            torch.cuda.synchronize()
            dpolys = [((d0.data_ptr(), len(co)), bound, None if d1 is None else (d1.data_ptr(), len(bl))) for (co, bound, bl), (d0, d1) in zip(polys, dev)]
            assert (aleo_amd.SonicKZG10.commit(ck, dpolys, device=True) == got).all()
            with pytest.raises(ValueError): aleo_amd.SonicKZG10.commit(ck, [(f(100, 1), 50, None)])          # bound below the degree


# ---- field kernels of the AHP rounds: blends, geometric sequences, gathers, batched evaluation, SRS-shaped sets ------------------
def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64).copy()).cuda()


def _mont1(v): return c.fr_to_mont(c.ints_to_limbs([v % p.FR_MODULUS], 4))[0]


@pytest.mark.parametrize('n', [1, 255, 256, 257, 70001])
def test_fr_lin_matches_bigint(n):
    """dst = c0 + c1 a + c2 b in all four shapes (terms present / absent), aliasing dst = a, against Python integers."""
    import torch
    from aleo_amd import poly
    r = p.FR_MODULUS
    A = util.uniform_scalars(n, 21000 + n); B = util.uniform_scalars(n, 22000 + n)
    a, b = c.limbs_to_ints(A), c.limbs_to_ints(B)
    c0, c1, c2 = (int(x) for x in c.limbs_to_ints(util.uniform_scalars(3, 23000 + n)))
    dA, dB = _dev(c.fr_to_mont(A)), _dev(c.fr_to_mont(B)); dst = torch.zeros((n, 4), dtype=torch.int64, device='cuda'); torch.cuda.synchronize()
    def got(): torch.cuda.synchronize(); return c.limbs_to_ints(c.fr_from_mont(dst.cpu().numpy().view(np.uint64)))
    poly.fr_lin_device(dst.data_ptr(), n, _mont1(c0), _mont1(c1), dA.data_ptr(), _mont1(c2), dB.data_ptr())
    assert got() == [(c0 + c1 * x + c2 * y) % r for x, y in zip(a, b)]
    poly.fr_lin_device(dst.data_ptr(), n, None, _mont1(r - 1), dA.data_ptr())
    assert got() == [(-x) % r for x in a]
    poly.fr_lin_device(dst.data_ptr(), n, _mont1(c0), None, 0, _mont1(c2), dB.data_ptr())
    assert got() == [(c0 + c2 * y) % r for y in b]
    poly.fr_lin_device(dst.data_ptr(), n, _mont1(c0))
    assert got() == [c0] * n
    poly.fr_lin_device(dA.data_ptr(), n, None, _mont1(1), dA.data_ptr(), _mont1(c2), dB.data_ptr())              # in place
    torch.cuda.synchronize()
    assert c.limbs_to_ints(c.fr_from_mont(dA.cpu().numpy().view(np.uint64))) == [(x + c2 * y) % r for x, y in zip(a, b)]


@pytest.mark.parametrize('n', [1, 15, 16, 17, 4096, 100001])
def test_fr_powers_and_the_inversion_free_tables(n):
    """dst[k] = first * ratio^k; and for a domain H: NTT of (a^(|H|-1-k))_k equals v_H(a) / (a - h) for every h in H."""
    import torch
    from aleo_amd import poly
    r = p.FR_MODULUS
    first, ratio = (int(x) for x in c.limbs_to_ints(util.uniform_scalars(2, 24000 + n)))
    dst = torch.zeros((n, 4), dtype=torch.int64, device='cuda'); torch.cuda.synchronize()
    poly.fr_powers_device(dst.data_ptr(), n, _mont1(first), _mont1(ratio)); torch.cuda.synchronize()
    want, acc = [], first
    for _ in range(n): want.append(acc); acc = acc * ratio % r
    assert c.limbs_to_ints(c.fr_from_mont(dst.cpu().numpy().view(np.uint64))) == want
    if n == 4096:
        a = ratio; dom = p.EvaluationDomain(n)
        poly.fr_powers_device(dst.data_ptr(), n, _mont1(pow(a, n - 1, r)), _mont1(pow(a, -1, r)))
        aleo_amd.EvaluationDomain(n).ntt_device(dst.data_ptr()); torch.cuda.synchronize()
        got = c.limbs_to_ints(c.fr_from_mont(dst.cpu().numpy().view(np.uint64)))
        vh = (pow(a, n, r) - 1) % r; h = 1
        for i in range(n):
            assert got[i] == vh * pow(a - h, -1, r) % r
            h = h * dom.group_gen % r


def test_fr_gather_mul_matches_bigint():
    import torch
    from aleo_amd import poly
    r = p.FR_MODULUS; n, m1, m2 = 50001, 777, 4100
    S, T1, T2 = util.uniform_scalars(n, 25001), util.uniform_scalars(m1, 25002), util.uniform_scalars(m2, 25003)
    i1 = (synth.splitmix_limbs(25004, n) % np.uint64(m1)).astype(np.uint32); i2 = (synth.splitmix_limbs(25005, n) % np.uint64(m2)).astype(np.uint32)
    s_, t1, t2 = c.limbs_to_ints(S), c.limbs_to_ints(T1), c.limbs_to_ints(T2)
    dS, d1, d2 = _dev(c.fr_to_mont(S)), _dev(c.fr_to_mont(T1)), _dev(c.fr_to_mont(T2))
    di1, di2 = torch.from_numpy(i1.view(np.int32)).cuda(), torch.from_numpy(i2.view(np.int32)).cuda()
    dst = torch.zeros((n, 4), dtype=torch.int64, device='cuda'); torch.cuda.synchronize()
    def got(): torch.cuda.synchronize(); return c.limbs_to_ints(c.fr_from_mont(dst.cpu().numpy().view(np.uint64)))
    poly.fr_gather_mul_device(dst.data_ptr(), n, dS.data_ptr(), d1.data_ptr(), di1.data_ptr(), d2.data_ptr(), di2.data_ptr())
    assert got() == [s_[i] * t1[i1[i]] % r * t2[i2[i]] % r for i in range(n)]
    poly.fr_gather_mul_device(dst.data_ptr(), n, 0, d1.data_ptr(), di1.data_ptr(), d2.data_ptr(), di2.data_ptr())
    assert got() == [t1[i1[i]] * t2[i2[i]] % r for i in range(n)]
    poly.fr_gather_mul_device(dst.data_ptr(), n, dS.data_ptr(), d1.data_ptr(), di1.data_ptr())
    assert got() == [s_[i] * t1[i1[i]] % r for i in range(n)]
    poly.fr_gather_mul_device(dst.data_ptr(), n, 0, d1.data_ptr(), di1.data_ptr())
    assert got() == [t1[i1[i]] for i in range(n)]


def test_fr_eval_batch_matches_oracle():
    """Up to eight polynomials of ragged lengths (empty, one coefficient, block boundaries, > 256 blocks) at their own points: each
    value equals the oracle's synthetic division remainder; a thirteenth polynomial is refused."""
    import torch
    from aleo_amd import poly
    lens = [0, 1, 16, 4095, 4096, 4097, 300001, (1 << 20) + 77]
    F = [c.fr_to_mont(util.uniform_scalars(max(n, 1), 26000 + i)) for i, n in enumerate(lens)]
    Z = c.fr_to_mont(util.uniform_scalars(len(lens), 26100)); Z[2] = 0; Z[3] = _mont1(1)
    D = [_dev(f) for f in F]
    out = torch.zeros((16, 4), dtype=torch.int64, device='cuda'); torch.cuda.synchronize()
    poly.fr_eval_batch_device(out.data_ptr(), [d.data_ptr() for d in D], lens, Z); torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    for i, n in enumerate(lens):
        want = c.fr_divide_by_linear(F[i][:n], Z[i])[1] if n else np.zeros(4, dtype=np.uint64)
        assert (got[i] == want).all(), (i, n)
    import ctypes                                                              # the entry point refuses a thirteenth polynomial; the wrapper cuts longer lists into calls of twelve
    ptrs13 = (ctypes.c_void_p * 13)(*[D[1].data_ptr()] * 13); lens13 = (ctypes.c_size_t * 13)(*[1] * 13); z13 = np.zeros((13, 4), dtype=np.uint64)
    assert aleo_amd.lib().aleo_mi355x_fr_eval_batch_device(ctypes.c_void_p(out.data_ptr()), ptrs13, lens13, z13.ctypes.data_as(ctypes.c_void_p), 13, ctypes.c_void_p(0)) == 2
    many = [D[i % 8].data_ptr() for i in range(15)]; ml = [lens[i % 8] for i in range(15)]; mz = np.stack([Z[i % 8] for i in range(15)])
    out.zero_(); poly.fr_eval_batch_device(out.data_ptr(), many, ml, mz); torch.cuda.synchronize()
    assert (out.cpu().numpy().view(np.uint64)[:15] == np.stack([got[i % 8] for i in range(15)])).all()
    ev = torch.zeros(4, dtype=torch.int64, device='cuda')                      # the division entry point with no quotient: evaluation only
    poly.divide_by_linear_device(0, ev.data_ptr(), D[6].data_ptr(), lens[6], Z[6]); torch.cuda.synchronize()
    assert (ev.cpu().numpy().view(np.uint64) == got[6]).all()


def test_bases_from_scalars_is_an_srs():
    """P_i = s_i G built in HBM: equal to the oracle's scalar multiples (incl. s = 0 -> identity, s = r - 1), and for s_i = tau^i a
    commitment is p(tau) G — the property KZG10 openings rest on."""
    r = p.FR_MODULUS; tau = 0x1234567890ABCDEF1234567890ABCDEF12345
    sc = [0, 1, 2, r - 1] + [pow(tau, i, r) for i in range(300)]
    S = c.ints_to_limbs(sc, 4)
    with M.PinnedBases.from_scalars(synth.generator_affine104(), S) as pb:
        B = pb.download()
        G = util.generator_affine()
        for i in (1, 2, 3, 4, 5, 150, 303):
            want = c.affine_to_ints(c.g1_mul(G, S[i]).reshape(1, 104))[0]
            assert c.affine_to_ints(B[i:i + 1])[0] == want
        coeffs = util.uniform_scalars(300, 27001)
        with M.PinnedBases(B[4:]) as srs:
            cm = aleo_amd.KZG10.commit(srs, c.fr_to_mont(coeffs))
        k = sum(v * pow(tau, i, r) for i, v in enumerate(c.limbs_to_ints(coeffs))) % r
        assert c.affine_to_ints(cm.reshape(1, 104))[0] == p.g1_mul(p.G1_GENERATOR, k)
        w = util.uniform_scalars(4, 27002); wi = c.limbs_to_ints(w)            # the identity as a base contributes nothing
        got = c.jac_to_int_point(M.VariableBase.msm(pb, w))
        assert got == p.g1_mul(p.G1_GENERATOR, (wi[1] + 2 * wi[2] + (r - 1) * wi[3]) % r)


def test_fr_random_stream_is_the_counter_based_definition():
    """The device's random stream equals the definition in the header element for element (oracle/varuna_ref.random_fr, and the
    host-side drawer in aleo_amd.poly), canonical and Montgomery, at an offset; every value is below r; streams differ by seed."""
    import torch
    from aleo_amd import poly
    from oracle import varuna_ref as V
    n, seed, first = 5000, 0xDEADBEEFCAFE, 12345
    dst = torch.zeros((n, 4), dtype=torch.int64, device='cuda'); torch.cuda.synchronize()
    poly.fr_random_device(dst.data_ptr(), n, seed, first, False); torch.cuda.synchronize()
    got = c.limbs_to_ints(dst.cpu().numpy().view(np.uint64))
    assert all(v < p.FR_MODULUS for v in got)
    for i in list(range(40)) + [n - 1]:
        assert got[i] == V.random_fr(seed, first + i) == poly.random_fr(seed, first + i)
    poly.fr_random_device(dst.data_ptr(), n, seed, first, True); torch.cuda.synchronize()
    assert c.limbs_to_ints(c.fr_from_mont(dst.cpu().numpy().view(np.uint64))) == got
    poly.fr_random_device(dst.data_ptr(), n, seed + 1, first, False); torch.cuda.synchronize()
    other = c.limbs_to_ints(dst.cpu().numpy().view(np.uint64))
    assert sum(a == b for a, b in zip(got, other)) == 0
    assert abs(sum(v >> 252 for v in got) / n - (p.FR_MODULUS - (1 << 252)) / p.FR_MODULUS) < 0.05        # top bit as often as uniformity says


def test_fr_lincomb_ragged_terms():
    import torch
    from aleo_amd import poly
    r = p.FR_MODULUS; n = 30011
    lens = [n, n - 1, 1, 0, 257, 4096, n + 50]                                  # a term longer than dst is cut at n
    T = [util.uniform_scalars(max(l, 1), 28000 + j) for j, l in enumerate(lens)]
    K = [int(x) for x in c.limbs_to_ints(util.uniform_scalars(len(lens) + 1, 28100))]
    D = [_dev(c.fr_to_mont(t)) for t in T]
    dst = torch.zeros((n, 4), dtype=torch.int64, device='cuda'); torch.cuda.synchronize()
    poly.fr_lincomb_device(dst.data_ptr(), n, _mont1(K[-1]), [(d.data_ptr(), l, _mont1(k)) for d, l, k in zip(D, lens, K)]); torch.cuda.synchronize()
    want = [0] * n; want[0] = K[-1]
    for t, l, k in zip(T, lens, K):
        for i, v in enumerate(c.limbs_to_ints(t[:min(l, n)])): want[i] = (want[i] + k * v) % r
    assert c.limbs_to_ints(c.fr_from_mont(dst.cpu().numpy().view(np.uint64))) == want
    # 29 terms: more than one call of the entry point takes (28) — the wrapper carries dst along as a term of the next call
    poly.fr_lincomb_device(dst.data_ptr(), n, None, [(D[0].data_ptr(), 1, _mont1(1))] * 29); torch.cuda.synchronize()
    first = c.limbs_to_ints(T[0][:1])[0]
    assert c.limbs_to_ints(c.fr_from_mont(dst.cpu().numpy().view(np.uint64)[:2])) == [29 * first % r, 0]


def test_ahp_sumcheck_numerators_match_bigint():
    import torch
    from aleo_amd import poly
    r = p.FR_MODULUS; n = 20003
    V5 = [util.uniform_scalars(n, 29000 + j) for j in range(5)]
    vr, va, vb, vt, vz = (c.limbs_to_ints(v) for v in V5)
    eb, ec = (int(x) for x in c.limbs_to_ints(util.uniform_scalars(2, 29100)))
    D = [_dev(c.fr_to_mont(v)) for v in V5]
    dst = torch.zeros((n, 4), dtype=torch.int64, device='cuda'); torch.cuda.synchronize()
    poly.ahp_first_sumcheck_device(dst.data_ptr(), n, D[0].data_ptr(), D[1].data_ptr(), D[2].data_ptr(), D[3].data_ptr(), D[4].data_ptr(), _mont1(eb), _mont1(ec))
    torch.cuda.synchronize()
    want = [(vr[i] * (va[i] + eb * vb[i] + ec * va[i] * vb[i]) - vt[i] * vz[i]) % r for i in range(n)]
    assert c.limbs_to_ints(c.fr_from_mont(dst.cpu().numpy().view(np.uint64))) == want
    poly.ahp_first_sumcheck_device(D[1].data_ptr(), n, D[0].data_ptr(), D[1].data_ptr(), D[2].data_ptr(), D[3].data_ptr(), D[4].data_ptr(), _mont1(eb), _mont1(ec))
    torch.cuda.synchronize()                                                    # in place over z_a, as the prover runs it
    assert c.limbs_to_ints(c.fr_from_mont(D[1].cpu().numpy().view(np.uint64))) == want
    # matrix sumcheck: three blocks of (row, col, val, row_col) + f
    IDX = [util.uniform_scalars(4 * n, 29200 + m) for m in range(3)]; F = [util.uniform_scalars(n, 29300 + m) for m in range(3)]
    ks = [int(x) for x in c.limbs_to_ints(util.uniform_scalars(6, 29400))]
    da, db_, dc, alpha, beta, vv = ks
    DI = [_dev(c.fr_to_mont(x)) for x in IDX]; DF = [_dev(c.fr_to_mont(x)) for x in F]
    consts = np.stack([_mont1(v) for v in (da, db_, dc, alpha * beta, -alpha, -beta, vv)])
    poly.ahp_matrix_sumcheck_device(dst.data_ptr(), n, [d.data_ptr() for d in DI], n, [d.data_ptr() for d in DF], consts); torch.cuda.synchronize()
    want = [0] * n
    for m, dl in enumerate((da, db_, dc)):
        e = c.limbs_to_ints(IDX[m]); f = c.limbs_to_ints(F[m])
        for i in range(n):
            bq = (alpha * beta - beta * e[i] - alpha * e[n + i] + e[3 * n + i]) % r
            want[i] = (want[i] + dl * (vv * e[2 * n + i] - bq * f[i])) % r
    assert c.limbs_to_ints(c.fr_from_mont(dst.cpu().numpy().view(np.uint64))) == want


def test_commit_lagrange_equals_commit_of_the_interpolant():
    """KZG10::commit_lagrange is the same MSM over the Lagrange-basis powers L_i(tau) G of the domain: committing the EVALUATIONS against
    them equals committing the interpolated coefficients against the monomial powers tau^i G (here both sets come from the synthetic
    setup's trapdoor; upstream derives the Lagrange set from the powers).  Witness-like evaluations stay witness-like scalars for the MSM."""
    import torch
    r = p.FR_MODULUS; n = 1024; tau = 0x2468ACE13579BDF02468ACE13579BDF
    dom = p.EvaluationDomain(n)
    vh = (pow(tau, n, r) - 1) % r; ninv = pow(n, -1, r)
    lag = []; w = 1
    for i in range(n):
        lag.append(w * ninv % r * vh % r * pow(tau - w, -1, r) % r); w = w * dom.group_gen % r
    mono = [pow(tau, i, r) for i in range(n)]
    evals = util.witness_like_scalars(n, 31001)
    coeffs = aleo_amd.EvaluationDomain(n).ifft(c.fr_to_mont(evals))
    with M.PinnedBases.from_scalars(synth.generator_affine104(), c.ints_to_limbs(lag, 4)) as pl, \
         M.PinnedBases.from_scalars(synth.generator_affine104(), c.ints_to_limbs(mono, 4)) as pm:
        a = aleo_amd.KZG10.commit(pl, c.fr_to_mont(evals)); b = aleo_amd.KZG10.commit(pm, coeffs)
        assert (a == b).all()
        k = sum(e * l for e, l in zip(c.limbs_to_ints(evals), lag)) % r
        assert c.affine_to_ints(a.reshape(1, 104))[0] == p.g1_mul(p.G1_GENERATOR, k)


def test_prover_layout_kernels_match_bigint():
    """fr_blind_rows (+ rho (X^n − 1) per row) and ahp_sumcheck_operands (z = w (X^|X| − 1) + x̂ and the zero-padded z_a, z_b on 4n
    coefficients, all instances in one launch) against Python integers."""
    import torch
    from aleo_amd import poly
    r = p.FR_MODULUS; n, n_x, k = 1024, 4, 3
    src = util.uniform_scalars(3 * k * n, 32001); rho = util.uniform_scalars(3 * k, 32002)
    d_src = _dev(c.fr_to_mont(src)); dst = torch.zeros((3 * k * (n + 1), 4), dtype=torch.int64, device='cuda'); torch.cuda.synchronize()
    poly.fr_blind_rows_device(dst.data_ptr(), d_src.data_ptr(), n, c.fr_to_mont(rho)); torch.cuda.synchronize()
    got = c.limbs_to_ints(c.fr_from_mont(dst.cpu().numpy().view(np.uint64))); s_, rh = c.limbs_to_ints(src), c.limbs_to_ints(rho)
    want = []
    for q in range(3 * k):
        row = s_[q * n:(q + 1) * n] + [rh[q]]; row[0] = (row[0] - rh[q]) % r; want += row
    assert got == want
    xp = util.uniform_scalars(k * n_x, 32003); d_xp = _dev(c.fr_to_mont(xp)); xs = c.limbs_to_ints(xp)
    out = torch.zeros((3 * k * 4 * n, 4), dtype=torch.int64, device='cuda'); torch.cuda.synchronize()
    poly.ahp_sumcheck_operands_device(out.data_ptr(), dst.data_ptr(), d_xp.data_ptr(), n, n_x, k); torch.cuda.synchronize()
    got2 = c.limbs_to_ints(c.fr_from_mont(out.cpu().numpy().view(np.uint64))); L = n + 1
    for i in range(k):
        w, za, zb = (want[(3 * i + j) * L:(3 * i + j + 1) * L] for j in range(3))
        z = [0] * (4 * n)
        for j, v in enumerate(w): z[j] = (z[j] - v) % r; z[j + n_x] = (z[j + n_x] + v) % r
        for j in range(n_x): z[j] = (z[j] + xs[i * n_x + j]) % r
        assert got2[(3 * i) * 4 * n:(3 * i + 1) * 4 * n] == z
        assert got2[(3 * i + 1) * 4 * n:(3 * i + 2) * 4 * n] == za + [0] * (4 * n - L) and got2[(3 * i + 2) * 4 * n:(3 * i + 3) * 4 * n] == zb + [0] * (4 * n - L)


def test_batched_msm_two_sets_on_the_widest_window():
    """Two result sets in ONE launch chain at c = 20 (2 x 2^19 buckets: the capacity the 4096 coarse bins allow) and a third that needs its own chain:
    the commitments of a 2^20-constraint proof's third round.  Against the structured identity and the single-vector calls; witness-like and
    uniform members, unequal lengths."""
    import torch
    N = 1 << 20
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, N) as pb:
        pb.precompute()
        lens = [N, (1 << 19) + 12345, N - 7]
        S = [util.uniform_scalars(lens[0], 33001), util.witness_like_scalars(lens[1], 33002), util.uniform_scalars(lens[2], 33003)]
        d = [torch.from_numpy(x.view(np.int64).copy()).cuda() for x in S]; torch.cuda.synchronize()
        got = M.VariableBase.msm_batch_device(pb, [t.data_ptr() for t in d], lens)
        for i, m in enumerate(lens):
            assert c.jac_to_int_point(got[i]) == util.expected_multiples_msm(S[i], m), (i, m)
            assert (M.VariableBase.msm_device(pb, d[i].data_ptr(), m) == got[i]).all(), (i, m)
        got2 = M.VariableBase.msm_batch_device(pb, [d[0].data_ptr(), d[2].data_ptr()], [lens[0], lens[2]])      # exactly one two-set chain
        assert (got2[0] == got[0]).all() and (got2[1] == got[2]).all()


def test_range_table_serves_sparse_commitments():
    """aleo_mi355x_bases_precompute_range + the sparse hint: results whose segments all lie inside the range come from the narrow-window table (here 9 results in
    one chain, witness-like and uniform members, a 1-element segment, unequal lengths) and equal the ordinary path's bit for bit; a call with a segment outside
    the range, and a call without the hint, are served the ordinary way; a second range on the same set is refused."""
    import torch
    from aleo_amd.kzg import CommitterKey, SonicKZG10
    N = 1 << 18; off = (1 << 17) + 3; rn = (1 << 16) + 77
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, N) as pb:
        pb.precompute(); pb.precompute_range(off, rn, 13)
        with pytest.raises(aleo_amd.AleoMi355xError): pb.precompute_range(0, 1024, 13)
        ck = CommitterKey.__new__(CommitterKey); ck.bases = pb; ck.max_degree = N - 1; ck.gamma_offset = 0; ck.n_gamma = 0
        lens = [rn - 1, 1 << 15, 40000, 5, rn - 1, 3000, 1 << 16, 1, 777]
        S = [util.witness_like_scalars(m, 34000 + i) if i % 3 else util.uniform_scalars(m, 34000 + i) for i, m in enumerate(lens)]
        d = [torch.from_numpy(c.fr_to_mont(x).view(np.int64).copy()).cuda() for x in S]; torch.cuda.synchronize()
        one = torch.from_numpy(c.fr_to_mont(util.uniform_scalars(1, 34100)).view(np.int64).copy()).cuda()
        segs = [(d[q].data_ptr(), lens[q], off, q) for q in range(9)] + [(one.data_ptr(), 1, off + rn - 1, 0), (one.data_ptr(), 1, off + rn - 1, 4)]
        want = SonicKZG10.commit_segments_device(ck, segs, 9)
        got = SonicKZG10.commit_segments_device(ck, segs, 9, sparse=True)
        assert (got == want).all()
        k0 = (synth.weighted_scalar_sum(S[3], off + 1)) % p.FR_MODULUS                    # bases are (i+1) G: result 3 is sum s_j (off + 1 + j) G
        assert c.affine_to_ints(got[3].reshape(1, 104))[0] == p.g1_mul(p.G1_GENERATOR, k0)
        outside = segs + [(one.data_ptr(), 1, 5, 2)]                                       # one segment outside the range: the whole call goes the ordinary way
        assert (SonicKZG10.commit_segments_device(ck, outside, 9, sparse=True) == SonicKZG10.commit_segments_device(ck, outside, 9)).all()


def test_witness_like_msm_takes_the_range_table():
    """A range table over the whole set (window 16) + witness-like scalars: the device call with the hint and the host-scalar call (which
    samples its input) give the ordinary result; uniform host scalars are not mistaken for a witness (same result either way, checked through
    the phase times: the wide window's reduction is the slow one)."""
    import torch
    N = 1 << 18
    with M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, N) as pb:
        pb.precompute(); pb.precompute_range(0, N, 16)
        S = util.witness_like_scalars(N, 35001); U = util.uniform_scalars(N, 35002)
        dS = torch.from_numpy(S.view(np.int64).copy()).cuda(); torch.cuda.synchronize()
        want = util.expected_multiples_msm(S, N)
        assert c.jac_to_int_point(M.VariableBase.msm_device(pb, dS.data_ptr(), N)) == want
        assert c.jac_to_int_point(M.VariableBase.msm_device(pb, dS.data_ptr(), N, sparse=True)) == want
        assert c.jac_to_int_point(M.VariableBase.msm(pb, S)) == want                       # host scalars: sampled, sparse
        assert c.jac_to_int_point(M.VariableBase.msm(pb, U)) == util.expected_multiples_msm(U, N)
        assert c.jac_to_int_point(M.VariableBase.msm(pb, S[:5000])) == util.expected_multiples_msm(S, 5000)

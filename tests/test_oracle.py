"""CPU tests: the oracle against the reference's own fixture and the committed golden vectors (not gpu)."""
import json, os
import numpy as np
import pytest
from oracle import pyref as p, coracle as c

G = os.path.join(os.path.dirname(__file__), 'golden')


def _ints(hexes): return [int(h, 16) for h in hexes]


def test_constants():
    assert pow(p.FR_GENERATOR, (p.FR_MODULUS - 1) >> 47, p.FR_MODULUS) == p.FR_TWO_ADIC_ROOT
    assert pow(p.FR_TWO_ADIC_ROOT, 1 << 46, p.FR_MODULUS) == p.FR_MODULUS - 1          # order exactly 2^47
    assert p.g1_is_on_curve(p.G1_GENERATOR) and p.g1_in_subgroup(p.G1_GENERATOR)
    assert (-pow(p.FR_MODULUS, -1, 1 << 64)) % (1 << 64) == 0x0a117fffffffffff
    assert (-pow(p.FQ_MODULUS, -1, 1 << 64)) % (1 << 64) == 0x8508bfffffffffff


def test_reference_proof_fixture_pins_curve_and_encoding():
    """The ten KZG commitments inside the reference's own `proof1…` test string must decompress to on-curve,
    r-torsion points, and recompress to the same bytes (pins q, the curve, Fq sqrt, the compressed encoding)."""
    fx = json.load(open(os.path.join(G, 'reference_proof.json')))
    hrp, raw = p.bech32m_decode(fx['proof'])
    assert hrp == 'proof' and len(raw) == fx['payload_len'] == 901
    assert raw[0] == 0 and int.from_bytes(raw[1:9], 'little') == 1 and int.from_bytes(raw[9:17], 'little') == 1
    for name, cm in fx['commitments'].items():
        buf = raw[cm['offset']:cm['offset'] + 48]
        assert buf.hex() == cm['compressed']
        pt = p.g1_decompress(buf)
        assert pt == (int(cm['x'], 16), int(cm['y'], 16))
        assert p.g1_is_on_curve(pt) and p.g1_in_subgroup(pt), name
        assert p.g1_compress(pt) == buf
        # the C oracle agrees that the point is on the curve after a Montgomery round trip
        aff = c.affine_from_ints([pt])
        assert c.lib().oracle_g1_on_curve(c._p(aff)) == 1
        assert c.affine_to_ints(aff)[0] == pt
    for name, vals in fx['field_elements'].items():
        assert all(int(v, 16) < p.FR_MODULUS for v in vals), name


def test_oracle_field_products_match_bigint():
    rng = p.SplitMix64(7)
    a = [rng.fr() for _ in range(64)]; b = [rng.fr() for _ in range(64)]
    am, bm = c.fr_to_mont(c.ints_to_limbs(a, 4)), c.fr_to_mont(c.ints_to_limbs(b, 4))
    assert c.limbs_to_ints(am) == [p.fr_to_mont(x) for x in a]
    r = np.zeros_like(am); c.lib().oracle_fr_mul(c._p(r), c._p(am), c._p(bm), 64)
    assert c.limbs_to_ints(c.fr_from_mont(r)) == [x * y % p.FR_MODULUS for x, y in zip(a, b)]
    qa = [(rng.fr() * rng.fr()) % p.FQ_MODULUS for _ in range(64)]; qb = [(rng.fr() * rng.fr() + 5) % p.FQ_MODULUS for _ in range(64)]
    qam, qbm = c.fq_to_mont(c.ints_to_limbs(qa, 6)), c.fq_to_mont(c.ints_to_limbs(qb, 6))
    r = np.zeros_like(qam); c.lib().oracle_fq_mul(c._p(r), c._p(qam), c._p(qbm), 64)
    assert c.limbs_to_ints(c.fq_from_mont(r)) == [x * y % p.FQ_MODULUS for x, y in zip(qa, qb)]


def golden_msm_cases():
    fx = json.load(open(os.path.join(G, 'msm_small.json')))
    gen = (int(fx['generator'][0], 16), int(fx['generator'][1], 16))
    assert gen == p.G1_GENERATOR
    return fx['cases']


def bases_for_case(case):
    pts = [p.g1_mul(p.G1_GENERATOR, k) if k else None for k in case['base_multipliers']]
    return c.affine_from_ints(pts)


@pytest.mark.parametrize('variant', [0, 1, 2, 3])
def test_oracle_msm_matches_golden(variant):
    for case in golden_msm_cases():
        B = bases_for_case(case); S = c.ints_to_limbs(_ints(case['scalars']), 4)
        exp = None if case['result'] is None else (int(case['result'][0], 16), int(case['result'][1], 16))
        for threads in (1, 3):
            assert c.jac_to_int_point(c.msm_g1(B, S, threads=threads, variant=variant)) == exp, (case['n'], case['kind'])
        if case['n'] <= 33:
            assert c.jac_to_int_point(c.msm_g1_naive(B, S)) == exp


def test_oracle_msm_stride96_and_edge_cases():
    rng = p.SplitMix64(11)
    G1 = c.affine_from_ints([p.G1_GENERATOR])[0]
    B = c.g1_multiples(G1, 50); S = c.ints_to_limbs([rng.fr() for _ in range(50)], 4)
    a = c.jac_to_int_point(c.msm_g1(B, S, variant=1))
    assert a == c.jac_to_int_point(c.msm_g1(np.ascontiguousarray(B[:, :96]), S, variant=1))
    assert c.jac_to_int_point(c.msm_g1(B[:0], S[:0])) is None                      # empty input -> identity
    same = np.repeat(B[:1], 40, axis=0); five = c.ints_to_limbs([5] * 40, 4)          # every pair collides: doublings
    assert c.jac_to_int_point(c.msm_g1(same, five, variant=1)) == p.g1_mul(p.G1_GENERATOR, 200)
    neg = c.affine_from_ints([p.g1_neg(p.G1_GENERATOR)])
    pm = np.concatenate([B[:1], neg] * 6, axis=0); s77 = c.ints_to_limbs([77] * 12, 4)  # P and -P cancel
    assert c.jac_to_int_point(c.msm_g1(pm, s77, variant=1)) is None
    assert c.jac_to_int_point(c.msm_g1(pm, s77, variant=0)) is None


def test_oracle_ntt_matches_golden():
    fx = json.load(open(os.path.join(G, 'ntt_small.json')))
    assert int(fx['two_adic_root'], 16) == p.FR_TWO_ADIC_ROOT and fx['generator'] == 22
    for case in fx['cases']:
        n = case['n']; x = _ints(case['input']); x = x + [0] * (n - len(x))            # fft() zero-pads to the domain
        xm = c.fr_to_mont(c.ints_to_limbs(x, 4))
        for name, (direction, type_) in {'fft': (0, 0), 'ifft': (1, 0), 'coset_fft': (0, 1), 'coset_ifft': (1, 1)}.items():
            if name not in case: continue
            got = c.limbs_to_ints(c.fr_from_mont(c.ntt_fr(xm, 0, direction, type_)))
            assert got == _ints(case[name]), (n, name)


def test_oracle_ntt_orders_and_roundtrip():
    rng = p.SplitMix64(13)
    n = 512; x = [rng.fr() for _ in range(n)]; xm = c.fr_to_mont(c.ints_to_limbs(x, 4))
    nat = c.ntt_fr(xm, 0, 0, 0)
    assert (c.ntt_fr(nat, 0, 1, 0) == xm).all()                                        # ifft(fft(x)) == x
    assert (c.ntt_fr(c.ntt_fr(xm, 0, 0, 1), 0, 1, 1) == xm).all()                      # coset round trip
    nat_i = c.limbs_to_ints(nat)
    assert c.limbs_to_ints(c.ntt_fr(xm, 1, 0, 0)) == p.bit_reverse_permute(nat_i)      # NR
    xr = c.fr_to_mont(c.ints_to_limbs(p.bit_reverse_permute(x), 4))
    assert c.limbs_to_ints(c.ntt_fr(xr, 2, 0, 0)) == nat_i                             # RN
    assert c.limbs_to_ints(c.ntt_fr(xr, 3, 0, 0)) == p.bit_reverse_permute(nat_i)      # RR
    assert c.limbs_to_ints(c.fr_from_mont(nat)) == p.fft_fast(x, p.EvaluationDomain(n).group_gen)


def test_oracle_kzg_commit_shape():
    rng = p.SplitMix64(17)
    G1 = c.affine_from_ints([p.G1_GENERATOR])[0]
    n = 40; B = c.g1_multiples(G1, n); coeff = [rng.fr() for _ in range(n)]
    cm = c.fr_to_mont(c.ints_to_limbs(coeff, 4))
    got = c.affine_to_ints(c.kzg_commit(B, cm, threads=2))[0]
    k = sum((i + 1) * v for i, v in enumerate(coeff)) % p.FR_MODULUS
    assert got == p.g1_mul(p.G1_GENERATOR, k)


def test_oracle_divide_by_linear_matches_bigint():
    """KZG10's witness polynomial: q (X - z) + p(z) == p, coefficient by coefficient, in Python integers."""
    rng = p.SplitMix64(23); r = p.FR_MODULUS
    for n in (1, 2, 3, 17, 300):
        f = [rng.fr() for _ in range(n)]; z = rng.fr()
        q, ev = c.fr_divide_by_linear(c.fr_to_mont(c.ints_to_limbs(f, 4)), c.fr_to_mont(c.ints_to_limbs([z], 4))[0])
        qi = c.limbs_to_ints(c.fr_from_mont(q)) if n > 1 else []; e = c.limbs_to_ints(c.fr_from_mont(ev.reshape(1, 4)))[0]
        assert e == sum(v * pow(z, i, r) for i, v in enumerate(f)) % r
        back = [0] * n
        for i, v in enumerate(qi): back[i + 1] = (back[i + 1] + v) % r; back[i] = (back[i] - z * v) % r
        back[0] = (back[0] + e) % r
        assert back == f


def _g2pt(v): return None if v is None else ((int(v[0][0], 16), int(v[0][1], 16)), (int(v[1][0], 16), int(v[1][1], 16)))


def test_g2_constants_and_oracle_msm_match_golden():
    """Fq2 non-residue -5, the G2 curve coefficient and generator [UPSTREAM-RECALL] are self-consistent (on the curve, r-torsion);
    the C oracle's G2 standard::msm equals the big-integer known answers."""
    assert pow(p.FQ2_NONRESIDUE, (p.FQ_MODULUS - 1) // 2, p.FQ_MODULUS) == p.FQ_MODULUS - 1
    assert p.g2_is_on_curve(p.G2_GENERATOR) and p.g2_mul_raw(p.G2_GENERATOR, p.FR_MODULUS) is None
    fx = json.load(open(os.path.join(G, 'msm_g2_small.json')))
    assert _g2pt(fx['generator']) == p.G2_GENERATOR
    gen = c.g2_affine_from_ints([p.G2_GENERATOR])[0]
    mult = c.g2_multiples(gen, 200)
    assert c.g2_affine_to_ints(mult[[0, 6, 199]]) == [p.g2_mul(p.G2_GENERATOR, k) for k in (1, 7, 200)]
    for case in fx['cases']:
        B = mult[np.array(case['base_multipliers']) - 1]; S = c.ints_to_limbs(_ints(case['scalars']), 4)
        assert c.g2_jac_to_int_point(c.msm_g2(B, S)) == _g2pt(case['result']), (case['n'], case['kind'])
        assert c.g2_jac_to_int_point(c.msm_g2(np.ascontiguousarray(B[:, :192]), S)) == _g2pt(case['result'])       # stride 192


def test_pairing_is_bilinear_and_of_order_r():
    """oracle/pairing.py (the ate pairing over Fq12 = Fq[w]/(w^12 + 5), used by the prover's verifier): non-degenerate, of order r,
    bilinear in both arguments, additive, trivial on the identity; the twist constant is 1/u."""
    from oracle import pairing as E
    q, r = p.FQ_MODULUS, p.FR_MODULUS
    assert p.G2_COEFF_B == (0, (-pow(5, -1, q)) % q) and r == E.X_PARAM ** 4 - E.X_PARAM ** 2 + 1 and (q ** 12 - 1) % r == 0
    G, H = p.G1_GENERATOR, p.G2_GENERATOR
    e = E.pairing(G, H)
    assert e != E.ONE12 and E.f12_pow(e, r) == E.ONE12
    a, b = 0x1234567890ABCDEF123, 0xFEDCBA0987654321
    assert E.pairing(p.g1_mul(G, a), p.g2_mul(H, b)) == E.f12_pow(e, a * b % r)
    assert E.pairing(p.g1_add(p.g1_mul(G, a), p.g1_mul(G, b)), H) == E.f12_mul(E.pairing(p.g1_mul(G, a), H), E.pairing(p.g1_mul(G, b), H))
    assert E.pairing(None, H) == E.ONE12 and E.pairing(G, None) == E.ONE12
    assert E.pairing_product_is_one([(p.g1_mul(G, a), H), (p.g1_neg(G), p.g2_mul(H, a))])
    assert not E.pairing_product_is_one([(p.g1_mul(G, a), H), (p.g1_neg(G), p.g2_mul(H, a + 1))])


def test_threaded_oracle_legs_equal_the_serial_ones(oracle):
    """The all-cores CPU baseline of the proof schedule (bench.py cpu_baseline.proof_proxy) runs the oracle's transforms with the butterflies of
    every stage split over threads, and the element-wise field work / batch inversion / matrix rows in per-thread chunks: same values."""
    from aleo_amd import synth
    c = oracle
    x = c.fr_to_mont(synth.uniform_scalars(1 << 12, 5))
    for order in range(4):
        for direction in (0, 1):
            for typ in (0, 1):
                assert (c.ntt_fr(x, order, direction, typ) == c.ntt_fr(x, order, direction, typ, threads=5)).all()
    for lg in (1, 2, 3):
        y = c.fr_to_mont(synth.uniform_scalars(1 << lg, 6)); assert (c.ntt_fr(y, 0, 1, 1) == c.ntt_fr(y, 0, 1, 1, threads=8)).all()
    a = c.fr_to_mont(synth.uniform_scalars(9001, 7)); b = c.fr_to_mont(synth.uniform_scalars(9001, 8)); a[17] = 0
    assert (c.fr_vec_op(a, b, 0) == c.fr_vec_op_mt(a, b, 0, 4)).all() and (c.fr_batch_inverse(a) == c.fr_batch_inverse_mt(a, 4)).all()
    rp = np.arange(0, 2 * 9001 + 1, 2, dtype=np.uint32); ci = (np.arange(2 * 9001, dtype=np.uint32) * 7) % 9001
    v = c.fr_to_mont(synth.uniform_scalars(2 * 9001, 9))
    assert (c.fr_spmv(rp, ci, v, a) == c.fr_spmv_mt(rp, ci, v, a, 3)).all()

"""CPU tests of the product's wire formats (aleo_amd/csrc/wire.hip, host code — no GPU needed) against DATA the reference's own
tests hold: the `proof1…` string of /root/reference/wasm/src/programs/transaction.rs:100 (committed with its decoding as
tests/golden/reference_proof.json) and the literals of tests/golden/reference_literals.json.  Byte-exact."""
import json, os
import numpy as np
import pytest
from aleo_amd import wire
from aleo_amd._lib import AleoMi355xError
from oracle import pyref as p, coracle as c

G = os.path.join(os.path.dirname(__file__), 'golden')
FX = json.load(open(os.path.join(G, 'reference_proof.json')))
LIT = json.load(open(os.path.join(G, 'reference_literals.json')))
ORDER = ['w', 'z_a', 'z_b', 'mask_poly', 'g_1', 'h_1', 'g_a', 'g_b', 'g_c', 'h_2']


def reference_points():
    """The twelve G1 points the reference's proof string holds: (name, compressed bytes, (x, y))."""
    pts = [(n, bytes.fromhex(FX['commitments'][n]['compressed']), (int(FX['commitments'][n]['x'], 16), int(FX['commitments'][n]['y'], 16))) for n in ORDER]
    pts += [('opening%d' % i, bytes.fromhex(o['compressed']), (int(o['x'], 16), int(o['y'], 16))) for i, o in enumerate(FX['openings'])]
    return pts


def test_g1_compression_matches_the_reference_bytes():
    pts = reference_points()
    comp = np.stack([np.frombuffer(b, dtype=np.uint8) for _, b, _ in pts])
    aff = wire.g1_decompress(comp, check_subgroup=True)                      # x -> y by Tonelli-Shanks, sign by flag, r * P == O
    assert c.affine_to_ints(aff) == [xy for _, _, xy in pts]
    assert (aff == c.affine_from_ints([xy for _, _, xy in pts])).all()         # same Montgomery limbs as the oracle's conversion
    assert (wire.g1_compress(aff) == comp).all()
    neg = c.affine_from_ints([p.g1_neg(xy) for _, _, xy in pts])              # -P: same x, other flag
    cneg = wire.g1_compress(neg)
    assert (cneg[:, :47] == comp[:, :47]).all() and ((cneg[:, 47] ^ comp[:, 47]) == 0x80).all()
    inf = np.zeros((1, 104), dtype=np.uint8); inf[0, 96] = 1
    ci = wire.g1_compress(inf); assert ci[0, 47] == 0x40 and not ci[0, :47].any()
    back = wire.g1_decompress(ci); assert back[0, 96] == 1


def test_g1_decompress_rejects_invalid_encodings():
    good = np.frombuffer(bytes.fromhex(FX['commitments']['w']['compressed']), dtype=np.uint8).copy()
    bad = good.copy(); bad[:] = 0xff; bad[47] = 0x3f                          # x >= q
    with pytest.raises(AleoMi355xError): wire.g1_decompress(bad)
    x = 1                                                                     # find a small x with x^3 + 1 a non-residue
    while p.fq_sqrt(x ** 3 + 1) is not None: x += 1
    off = np.frombuffer(x.to_bytes(48, 'little'), dtype=np.uint8)
    with pytest.raises(AleoMi355xError): wire.g1_decompress(off)
    junk = good.copy(); junk[47] |= 0x40                                      # infinity flag with an x
    with pytest.raises(AleoMi355xError): wire.g1_decompress(junk)
    # a curve point outside the prime-order subgroup (cofactor != 1): accepted without the check, rejected with it
    x = 2
    while True:
        y = p.fq_sqrt(x ** 3 + 1)
        if y is not None and not p.g1_in_subgroup((x, y)): break
        x += 1
    out = np.frombuffer(p.g1_compress((x, y)), dtype=np.uint8)
    assert c.affine_to_ints(wire.g1_decompress(out, check_subgroup=False))[0] == (x, y)
    with pytest.raises(AleoMi355xError): wire.g1_decompress(out, check_subgroup=True)


def test_fr_bytes_roundtrip_and_range():
    vals = [int(v, 16) for v in FX['field_elements']['evaluations'] + FX['field_elements']['sums']] + [int(FX['openings'][0]['random_v'], 16)]
    raw = np.stack([np.frombuffer(v.to_bytes(32, 'little'), dtype=np.uint8) for v in vals])
    m = wire.fr_from_bytes(raw)
    assert c.limbs_to_ints(m) == [p.fr_to_mont(v) for v in vals]
    assert (wire.fr_to_bytes(m) == raw).all()
    with pytest.raises(AleoMi355xError): wire.fr_from_bytes(np.frombuffer(p.FR_MODULUS.to_bytes(32, 'little'), dtype=np.uint8))
    assert c.limbs_to_ints(wire.fr_from_bytes(np.frombuffer((p.FR_MODULUS - 1).to_bytes(32, 'little'), dtype=np.uint8))) == [p.fr_to_mont(p.FR_MODULUS - 1)]


def test_proof_reassembled_from_its_parts_equals_the_reference_string():
    """Decode the reference's proof into commitments / evaluations / openings (fixture), serialise those parts with the product
    (compressed G1, canonical Fr, layout, bech32m): the result must be the reference's string, character for character."""
    aff = lambda names: c.affine_from_ints([(int(FX['commitments'][n]['x'], 16), int(FX['commitments'][n]['y'], 16)) for n in names])
    frm = lambda hexes: c.fr_to_mont(c.ints_to_limbs([int(h, 16) for h in hexes], 4))
    op = c.affine_from_ints([(int(o['x'], 16), int(o['y'], 16)) for o in FX['openings']])
    rv = [frm([FX['openings'][0]['random_v']])[0], None]
    raw = wire.proof_to_bytes(FX['layout']['batch_sizes'], aff(['w', 'z_a', 'z_b']), aff(['mask_poly']), aff(['g_1']), aff(['h_1']),
                              aff(['g_a', 'g_b', 'g_c']), aff(['h_2']), frm(FX['field_elements']['evaluations']), frm(FX['field_elements']['sums']), op, rv)
    hrp, want = p.bech32m_decode(FX['proof'])
    assert len(raw) == FX['payload_len'] == 901 and raw == want
    assert wire.proof_to_string(raw) == FX['proof']
    assert wire.bech32m_decode(FX['proof']) == ('proof', want)


def test_bech32m_against_every_string_the_reference_holds():
    for e in LIT['bech32m'] + [{'string': FX['proof'], 'hrp': 'proof', 'payload': p.bech32m_decode(FX['proof'])[1].hex()}]:
        hrp, raw = wire.bech32m_decode(e['string'])
        assert hrp == e['hrp'] and raw.hex() == e['payload'], e['source'] if 'source' in e else 'proof'
        assert wire.bech32m_encode(hrp, raw) == e['string']
        assert p.bech32m_encode(hrp, raw) == e['string']                       # the oracle's encoder agrees
        s = e['string']; flip = s[:-1] + ('q' if s[-1] != 'q' else 'p')        # any single substitution breaks the checksum
        with pytest.raises(AleoMi355xError): wire.bech32m_decode(flip)
    with pytest.raises(AleoMi355xError): wire.bech32m_decode('proof1')
    with pytest.raises(AleoMi355xError): wire.bech32m_decode('noseparator')


def test_reference_field_and_group_literals_pin_fr():
    """`…field` literals are canonical Fr values; `…group` literals and the aleo1 addresses are x-coordinates on the Edwards-BLS12
    curve -x^2 + y^2 = 1 + 3021 x^2 y^2 over Fr: y^2 = (1 + x^2) / (1 - 3021 x^2) must be a square — a known answer for the
    oracle's Fr inverse, Legendre symbol and Tonelli-Shanks (two-adicity 47), and for the C oracle's Montgomery products and
    batch inversion, on values the reference holds."""
    r = p.FR_MODULUS
    xs = [int(g['x']) for g in LIT['groups']] + [int(b['value']) for b in LIT['bech32m'] if b['hrp'] == 'aleo']
    assert len(xs) >= 7
    for f in LIT['fields'] + [b for b in LIT['bech32m'] if b['hrp'] in ('at', 'as', 'ar')]:
        assert int(f['value']) < r
    ys = []
    for x in xs:
        assert x < r
        y2 = (1 + x * x) * pow((1 - 3021 * x * x) % r, -1, r) % r
        y = p.fr_sqrt(y2); assert y is not None, x
        assert (-x * x + y * y - 1 - 3021 * x * x * y * y) % r == 0
        ys.append(y)
    for g, y in zip(LIT['groups'], ys): assert int(g['y'], 16) in (y, r - y)
    # the same relation through the C oracle's Montgomery arithmetic: den^-1 by batch inversion, products by oracle_fr_mul
    X = c.fr_to_mont(c.ints_to_limbs(xs, 4)); Y = c.fr_to_mont(c.ints_to_limbs(ys, 4))
    mul = lambda a, b: (lambda o: (c.lib().oracle_fr_mul(c._p(o), c._p(a), c._p(b), a.shape[0]), o)[1])(np.zeros_like(a))
    XX, YY = mul(X, X), mul(Y, Y)
    d = c.fr_to_mont(c.ints_to_limbs([3021] * len(xs), 4)); one = c.fr_to_mont(c.ints_to_limbs([1] * len(xs), 4))
    den = c.fr_vec_op(one, mul(d, XX), 2)                                      # 1 - d x^2
    inv = c.fr_batch_inverse(den)
    assert (mul(c.fr_vec_op(one, XX, 1), inv) == YY).all()                     # (1 + x^2) / (1 - d x^2) == y^2, limb for limb


def test_host_parsers_under_address_and_ub_sanitizers(tmp_path):
    """SURVEY.md 5 (sanitizers on the CPU build): aleo_amd/csrc/wire.hip and sponge.hip — host-only code of the product that parses untrusted bytes
    (bech32m_decode, g1_decompress, fr_from_bytes, proof_to_bytes) or hashes (Poseidon, the Fiat-Shamir sponge, the ChaCha20 stream) — compiled as
    plain C++ with -fsanitize=address,undefined and driven by tests/cpp/wire_fuzz.cpp: the reference's proof string, malformed variants, a
    4000-step mutation loop, buffer-size edge cases (tools/asan_host.sh).  Any sanitizer report fails the run."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proof = json.load(open(os.path.join(root, 'tests', 'golden', 'reference_proof.json')))['proof']
    r = subprocess.run([os.path.join(root, 'tools', 'asan_host.sh'), str(tmp_path), proof], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'SANITIZED OK' in r.stdout and 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stdout[-1500:] + r.stderr[-3000:]

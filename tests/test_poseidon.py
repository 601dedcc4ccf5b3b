"""snarkVM's Poseidon against the known answers the reference's own tests hold (tests/golden/reference_account.json, generated from
/root/reference by tests/golden/gen_golden.py) — the restatement (oracle/poseidon.py) and the PRODUCT's (aleo_amd/csrc/poseidon.hpp through the
C ABI: aleo_mi355x_poseidon_hash_fr, aleo_mi355x_fs_*, host code of libaleo_mi355x.so, no GPU needed); the product's Fiat-Shamir sponge and
random stream against the restatement's.  CPU only."""
import ctypes, json, os, random
import numpy as np
import pytest
import aleo_amd
from aleo_amd import synth
from oracle import poseidon as ps, pyref as P

HERE = os.path.dirname(os.path.abspath(__file__))
ACC = json.load(open(os.path.join(HERE, 'golden', 'reference_account.json')))


def test_private_key_ciphertext_decrypts_to_the_reference_key():
    """/root/reference/wasm/src/account/private_key_ciphertext.rs:115-128: hash_many_psd8 (randomizers) and hash_psd2 (blinding)."""
    k = ACC['ciphertext_kat']
    assert ps.decrypt_private_key(k['ciphertext'], k['secret']) == k['private_key']
    # the plaintext between the two hashes: a struct of two field literals, terminus bit and padding exactly where upstream puts them
    kind, members = ps.decrypt_symmetric(k['ciphertext'], ps.domain_separator(k['secret']))
    assert kind == 'struct' and list(members) == ['key', 'nonce'] and all(m[:2] == ('literal', ps.LITERAL_FIELD) for m in members.values())
    # the wrong secret yields garbage that does not parse (the reference asserts is_err) ...
    with pytest.raises(Exception):
        ps.decrypt_private_key(k['ciphertext'], k['wrong_secret'])
    # ... and the reference's corrupted ciphertext fails its checksum
    with pytest.raises(Exception):
        ps.ciphertext_fields(k['bad_ciphertext'])


def test_account_derivation_matches_the_reference_triples():
    """private key -> view key -> address (wasm/src/account/private_key.rs:182-198): hash_to_scalar_psd2, hash_to_scalar_psd4, Edwards-BLS12."""
    a0 = ACC['accounts'][0]
    v0, A0 = ps.view_key_scalar(a0['view_key']), ps.address_point(a0['address'])
    G = ps.ed_mul(A0, pow(v0, -1, ps.ED_SUBGROUP_ORDER))            # the account generator, recovered from the first triple
    assert ps.ed_mul(G, ps.ED_SUBGROUP_ORDER) == (0, 1)
    for a in ACC['accounts']:
        assert ps.derive_account(a['private_key'], G) == (a['view_key'], a['address']), a['source']
        assert ps.private_key_string(ps.private_key_seed(a['private_key'])) == a['private_key']


def test_parameters_shape_and_sponge_consistency():
    for mod in (P.FR_MODULUS, P.FQ_MODULUS):
        for rate in (2, 4, 8):
            ark, mds = ps.parameters(mod, rate)
            assert len(ark) == 39 and all(len(r) == rate + 1 and all(0 <= v < mod for v in r) for r in ark)
            assert len(mds) == rate + 1 and len({v for r in mds for v in r}) == (rate + 1) ** 2
    # squeezing in pieces equals squeezing at once; absorbing in pieces equals absorbing at once
    for mod in (P.FR_MODULUS, P.FQ_MODULUS):
        a, b = ps.Sponge(mod, 2), ps.Sponge(mod, 2)
        a.absorb([1, 2, 3, 4, 5]); b.absorb([1]); b.absorb([2, 3]); b.absorb([4, 5])
        x = a.squeeze(5); y = b.squeeze(1) + b.squeeze(3) + b.squeeze(1)
        assert x == y and len(set(x)) == 5
        a.absorb([7]); b.absorb([7])
        assert a.squeeze(2) == b.squeeze(2)


# ---- the product's Poseidon (C++ host code of the library) ----------------------------------------------------------------------------------
def product_hash_many(rate, inputs, n_out):
    L = aleo_amd.lib()
    a = np.stack([synth.int_to_limbs(int(v), 4) for v in inputs]) if len(inputs) else np.zeros((0, 4), dtype=np.uint64)
    out = np.zeros((max(n_out, 1), 4), dtype=np.uint64)
    aleo_amd._lib.check(L.aleo_mi355x_poseidon_hash_fr(rate, a.ctypes.data_as(ctypes.c_void_p), len(inputs), out.ctypes.data_as(ctypes.c_void_p), n_out), 'poseidon_hash_fr')
    return [synth.limbs_to_int(out[i]) for i in range(n_out)]


def test_product_poseidon_reproduces_the_reference_known_answers():
    """The same two reference-held known answers through libaleo_mi355x.so's Poseidon: rates 2, 4 and 8 over Fr."""
    k = ACC['ciphertext_kat']
    assert ps.decrypt_private_key(k['ciphertext'], k['secret'], hasher=product_hash_many) == k['private_key']
    a0 = ACC['accounts'][0]
    G = ps.ed_mul(ps.address_point(a0['address']), pow(ps.view_key_scalar(a0['view_key']), -1, ps.ED_SUBGROUP_ORDER))
    for a in ACC['accounts']:
        assert ps.derive_account(a['private_key'], G, hasher=product_hash_many) == (a['view_key'], a['address']), a['source']


def test_product_poseidon_matches_restatement_and_refuses_bad_input():
    rnd = random.Random(5)
    for rate in (2, 4, 8):
        for n in (0, 1, 2, 3, 7, 9, 17):
            ins = [rnd.randrange(P.FR_MODULUS) for _ in range(n)]
            assert product_hash_many(rate, ins, 5) == ps.hash_many(rate, ins, 5), (rate, n)
    with pytest.raises(aleo_amd.AleoMi355xError): product_hash_many(3, [1], 1)                       # no such rate
    with pytest.raises(aleo_amd.AleoMi355xError): product_hash_many(2, [P.FR_MODULUS], 1)            # not canonical


def test_product_fiat_shamir_sponge_matches_restatement(oracle):
    """Every entry point of the prover's transcript (bytes, points incl. infinity, non-native elements, full and short challenges, interleaved)."""
    from aleo_amd import varuna
    rnd = random.Random(9)
    fs, ref = varuna.FiatShamir(), ps.FiatShamir()
    for data in (b'VARUNA-2023', (3).to_bytes(8, 'little'), bytes(rnd.randrange(256) for _ in range(100)), bytes(47), bytes([255] * 48)):
        fs.absorb_bytes(data); ref.absorb_bytes(data)
    xs = [rnd.randrange(P.FR_MODULUS) for _ in range(7)] + [0, 1, P.FR_MODULUS - 1]
    fs.absorb_fr(xs); ref.absorb_nonnative(xs)
    pts = [P.g1_mul(P.G1_GENERATOR, k) for k in (1, 2, 99)] + [None]
    fs.absorb_g1(oracle.affine_from_ints(pts)); ref.absorb_points(pts)
    assert fs.squeeze(3) == ref.squeeze_nonnative(3) and fs.squeeze(0) == []
    assert fs.squeeze_short() == ref.squeeze_short_one()
    fs.absorb_fr(xs[:3]); ref.absorb_nonnative(xs[:3])
    assert fs.squeeze(5, True) == ref.squeeze_short(5) and fs.squeeze(1) == ref.squeeze_nonnative(1)
    for n in (1, 2, 4, 9): assert fs.squeeze(n) == ref.squeeze_nonnative(n)
    L = aleo_amd.lib()
    assert L.aleo_mi355x_fs_absorb_fr(12345678, None, 0) != 0                                        # unknown handle
    bad = np.full((1, 13), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)                                     # coordinates >= q
    assert L.aleo_mi355x_fs_absorb_g1(fs.h, bad.ctypes.data_as(ctypes.c_void_p), 104, 1) != 0


def test_product_random_stream_matches_restatement_and_the_rfc_vector():
    from aleo_amd import poly
    from oracle import varuna_ref as V
    # RFC 7539 section 2.3.2 (key 00..1f, counter 1, nonce 00 00 00 09 00 00 00 4a 00 00 00 00) in the 64-bit counter / nonce form
    blk = V.chacha20_block(bytes(range(32)), 1 | (0x09000000 << 32), 0x4A000000)
    assert blk.hex().startswith('10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e') and blk.hex().endswith('b5129cd1de164eb9cbd083e8a2503c4e')
    for seed in (7, b'\x01' * 32, bytes(range(32))):
        got = poly.random_fr(seed, 5, 64)
        assert got == [V.random_fr(seed, 5 + i) for i in range(64)] and all(v < P.FR_MODULUS for v in got) and len(set(got)) == 64
    assert poly.random_fr(7, 0) != poly.random_fr(8, 0)

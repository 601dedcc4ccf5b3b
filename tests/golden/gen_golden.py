#!/usr/bin/env python3
"""Generates the golden fixtures in this directory.

  reference_proof.json  DATA held by the reference's own test (the `proof1…` string inside TRANSACTION_STRING at
                        /root/reference/wasm/src/programs/transaction.rs:100), plus its decoding by oracle/pyref.py:
                        the ten KZG commitments (compressed G1) and the field evaluations.  Needs /root/reference.
  reference_literals.json  other DATA the reference's tests hold: the `...group` / `...field` decimal literals and the bech32m strings
                        (aleo1 addresses, at1 / as1 / ar1 ids, record1 ciphertexts) in transaction.rs:100, rust/src/test_utils/mod.rs:
                        132-142 and wasm/tests/offchain.rs:106, with their decoding (Edwards-BLS12 y for every group x).  Needs /root/reference.
  reference_account.json  DATA the reference's tests hold about accounts: the (private key, secret, ciphertext) triple at
                        wasm/src/account/private_key_ciphertext.rs:115-121 and the (private key, view key, address) triples at
                        wasm/src/account/private_key.rs:182-184, sdk/tests/data/account-data.ts:8-19 — known answers for Poseidon rates 2, 4 and 8
                        over Fr (oracle/poseidon.py).  Needs /root/reference.
  msm_small.json        Python big-integer MSM known answers (oracle/pyref.py msm_naive: double-and-add, no windows).
  msm_g2_small.json     the same for G2 over Fq2 (oracle/pyref.py msm_naive_g2).
  ntt_small.json        O(n^2) DFT known answers for fft / ifft / coset_fft / coset_ifft.
  varuna_small.json     proofs of the restatement of the prover (oracle/varuna_ref.py) for small synthetic circuits: seeds, sizes, the verifying-key
                        bytes and the proof bytes — what the device prover has to reproduce, frozen so that neither side can drift unnoticed.

Run from the repo root:  python tests/golden/gen_golden.py
The reference cannot be built or imported here (Rust, crates.io snarkVM 0.14.5): these vectors come from the
independent big-integer restatement, and from the reference's own data file for reference_proof.json."""
import json, os, re, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyref as p

COMMIT_OFFSETS = {'w': 17, 'z_a': 65, 'z_b': 113, 'mask_poly': 162, 'g_1': 210, 'h_1': 258, 'g_a': 306, 'g_b': 354, 'g_c': 402, 'h_2': 450}
EVAL_RANGES = {'evaluations': (498, 5), 'sums': (666, 3)}


def gen_reference_proof():
    src = '/root/reference/wasm/src/programs/transaction.rs'
    if not os.path.exists(src):
        print('skip reference_proof.json (no /root/reference)'); return
    proof = re.findall(r'proof1[0-9a-z]+', open(src).read())[0]
    hrp, raw = p.bech32m_decode(proof)
    out = {'source': 'wasm/src/programs/transaction.rs:100 (TRANSACTION_STRING, field "proof")', 'proof': proof,
           'hrp': hrp, 'payload_len': len(raw), 'commitments': {}, 'field_elements': {}}
    for name, off in COMMIT_OFFSETS.items():
        pt = p.g1_decompress(raw[off:off + 48])
        out['commitments'][name] = {'offset': off, 'compressed': raw[off:off + 48].hex(), 'x': hex(pt[0]), 'y': hex(pt[1])}
    for name, (off, cnt) in EVAL_RANGES.items():
        out['field_elements'][name] = [hex(int.from_bytes(raw[off + 32 * i: off + 32 * i + 32], 'little')) for i in range(cnt)]
    # the two KZG10 opening proofs: [762] u64 = 2; 770: w, tag 1, random_v; 851: w, tag 0; [900] = 0
    assert int.from_bytes(raw[762:770], 'little') == 2 and raw[818] == 1 and raw[899] == 0 and raw[900] == 0
    out['openings'] = []
    for off, has_v in ((770, True), (851, False)):
        pt = p.g1_decompress(raw[off:off + 48])
        o = {'offset': off, 'compressed': raw[off:off + 48].hex(), 'x': hex(pt[0]), 'y': hex(pt[1])}
        if has_v: o['random_v'] = hex(int.from_bytes(raw[off + 49:off + 81], 'little'))
        out['openings'].append(o)
    out['layout'] = {'version': raw[0], 'batch_sizes': [int.from_bytes(raw[9:17], 'little')], 'mask_poly_tag': raw[161],
                     'sums_len': int.from_bytes(raw[658:666], 'little'), 'openings_len': 2, 'trailer': raw[900]}
    json.dump(out, open(os.path.join(HERE, 'reference_proof.json'), 'w'), indent=1)


ED_D = 3021          # Edwards-BLS12: -x^2 + y^2 = 1 + 3021 x^2 y^2 over Fr (snarkvm-curves edwards_bls12 [UPSTREAM-RECALL])


def edwards_y(x):
    """A y with (x, y) on the curve, or None: y^2 = (1 + x^2) / (1 - d x^2)."""
    r = p.FR_MODULUS
    den = (1 - ED_D * x * x) % r
    if den == 0: return None
    return p.fr_sqrt((1 + x * x) * pow(den, -1, r) % r)


def gen_reference_literals():
    files = {'wasm/src/programs/transaction.rs': None, 'rust/src/test_utils/mod.rs': None, 'wasm/tests/offchain.rs': None}
    if not os.path.exists('/root/reference'):
        print('skip reference_literals.json (no /root/reference)'); return
    out = {'groups': [], 'fields': [], 'bech32m': []}
    seen = set()
    for rel in files:
        for ln, line in enumerate(open(os.path.join('/root/reference', rel)).read().split('\n'), 1):
            for m in re.finditer(r'(\d{20,})(group|field)', line):
                v, kind = int(m.group(1)), m.group(2)
                if (v, kind) in seen: continue
                seen.add((v, kind))
                if kind == 'group':
                    y = edwards_y(v)
                    out['groups'].append({'source': '%s:%d' % (rel, ln), 'x': str(v), 'y': None if y is None else hex(y)})
                else:
                    out['fields'].append({'source': '%s:%d' % (rel, ln), 'value': str(v)})
            for m in re.finditer(r'\b(aleo|at|as|ar|record)1[0-9a-z]{20,}', line):
                sv = m.group(0)
                if sv in seen: continue
                seen.add(sv)
                hrp, raw = p.bech32m_decode(sv)
                e = {'source': '%s:%d' % (rel, ln), 'string': sv, 'hrp': hrp, 'payload': raw.hex()}
                if hrp in ('aleo', 'at', 'as', 'ar'): e['value'] = str(int.from_bytes(raw, 'little'))
                out['bech32m'].append(e)
    json.dump(out, open(os.path.join(HERE, 'reference_literals.json'), 'w'), indent=1)


def gen_reference_account():
    ref = '/root/reference'
    if not os.path.exists(ref):
        print('skip reference_account.json (no /root/reference)'); return
    src = open(os.path.join(ref, 'wasm/src/account/private_key_ciphertext.rs')).read().split('\n')
    lo = next(i for i, l in enumerate(src) if 'fn test_private_key_from_string_decryption_edge_cases' in l)
    body = '\n'.join(src[lo:lo + 20])
    out = {'ciphertext_kat': {'source': 'wasm/src/account/private_key_ciphertext.rs:%d-%d' % (lo + 1, lo + 8),
                              'private_key': re.search(r'APrivateKey1[0-9A-Za-z]+', body).group(0),
                              'secret': re.search(r'decrypt_to_private_key\("([a-z]+)"\)\.unwrap', body).group(1),
                              'wrong_secret': re.search(r'decrypt_to_private_key\("([a-z]+)"\)\.is_err', body).group(1),
                              'ciphertext': re.findall(r'ciphertext1[0-9a-z]+', body)[0],
                              'bad_ciphertext': re.findall(r'ciphertext1[0-9a-z]+', body)[1]},
           'accounts': []}
    t = open(os.path.join(ref, 'wasm/src/account/private_key.rs')).read()
    out['accounts'].append({'source': 'wasm/src/account/private_key.rs:182-184', 'private_key': re.search(r'ALEO_PRIVATE_KEY: &str = "(\w+)"', t).group(1),
                            'view_key': re.search(r'ALEO_VIEW_KEY: &str = "(\w+)"', t).group(1), 'address': re.search(r'ALEO_ADDRESS: &str = "(\w+)"', t).group(1)})
    t = open(os.path.join(ref, 'sdk/tests/data/account-data.ts')).read()
    g = lambda name: re.search(name + r'\s*=\s*"(\w+)"', t).group(1)
    out['accounts'].append({'source': 'sdk/tests/data/account-data.ts:12-14', 'private_key': g('privateKeyString'), 'view_key': g('viewKeyString'), 'address': g('addressString')})
    out['accounts'].append({'source': 'sdk/tests/data/account-data.ts:8,17,19', 'private_key': g('beaconPrivateKeyString'), 'view_key': g('beaconViewKeyString'), 'address': g('beaconAddressString')})
    json.dump(out, open(os.path.join(HERE, 'reference_account.json'), 'w'), indent=1)


def gen_msm():
    rng = p.SplitMix64(0xA1E00002)
    cases = []
    r = p.FR_MODULUS

    def scal(kind, n):
        if kind == 'uniform': return [rng.fr() for _ in range(n)]
        if kind == 'zero': return [0] * n
        if kind == 'one': return [1] * n
        if kind == 'r_minus_1': return [r - 1] * n
        if kind == 'small': return [rng.next() & 0xFFFF for _ in range(n)]
        if kind == 'mixed': return [[0, 1, r - 1, rng.fr(), rng.next() & 0xFFFF][i % 5] for i in range(n)]
        raise ValueError(kind)

    for n in (1, 2, 31, 32, 33, 100):
        for kind in ('uniform', 'zero', 'one', 'r_minus_1', 'small', 'mixed'):
            if n > 33 and kind in ('zero', 'one', 'r_minus_1'): continue
            mult = [(rng.next() % 1000) + 1 for _ in range(n)]         # bases k_i * G, small k so bases repeat/collide
            bases = [p.g1_mul(p.G1_GENERATOR, k) for k in mult]
            sc = scal(kind, n)
            res = p.msm_naive(bases, sc)
            k = sum(m * s for m, s in zip(mult, sc)) % r               # cross-check: result must be k*G
            assert res == p.g1_mul(p.G1_GENERATOR, k)
            cases.append({'n': n, 'kind': kind, 'base_multipliers': mult, 'scalars': [hex(s) for s in sc],
                          'result': None if res is None else [hex(res[0]), hex(res[1])]})
    # infinity among the bases, and P / -P pairs
    mult = [3, 5, 7, 9]; bases = [p.g1_mul(p.G1_GENERATOR, k) for k in mult]; bases[1] = None
    sc = [rng.fr() for _ in range(4)]
    res = p.msm_naive([b for b in bases], sc)
    cases.append({'n': 4, 'kind': 'with_infinity_base', 'base_multipliers': [3, 0, 7, 9], 'scalars': [hex(s) for s in sc], 'result': [hex(res[0]), hex(res[1])]})
    json.dump({'generator': [hex(p.G1_GENERATOR[0]), hex(p.G1_GENERATOR[1])], 'cases': cases}, open(os.path.join(HERE, 'msm_small.json'), 'w'))


def gen_msm_g2():
    rng = p.SplitMix64(0xA1E00005); r = p.FR_MODULUS
    cases = []
    for n, kind in ((1, 'uniform'), (2, 'uniform'), (7, 'r_minus_1'), (31, 'mixed'), (33, 'uniform'), (64, 'small')):
        mult = [(rng.next() % 200) + 1 for _ in range(n)]
        bases = [p.g2_mul(p.G2_GENERATOR, k) for k in mult]
        if kind == 'uniform': sc = [rng.fr() for _ in range(n)]
        elif kind == 'r_minus_1': sc = [r - 1] * n
        elif kind == 'small': sc = [rng.next() & 0xFFFF for _ in range(n)]
        else: sc = [[0, 1, r - 1, rng.fr(), rng.next() & 0xFFFF][i % 5] for i in range(n)]
        res = p.msm_naive_g2(bases, sc)
        assert res == p.g2_mul(p.G2_GENERATOR, sum(m * v for m, v in zip(mult, sc)) % r)
        cases.append({'n': n, 'kind': kind, 'base_multipliers': mult, 'scalars': [hex(v) for v in sc],
                      'result': None if res is None else [[hex(res[0][0]), hex(res[0][1])], [hex(res[1][0]), hex(res[1][1])]]})
    g = p.G2_GENERATOR
    json.dump({'generator': [[hex(g[0][0]), hex(g[0][1])], [hex(g[1][0]), hex(g[1][1])]], 'coeff_b': [hex(v) for v in p.G2_COEFF_B], 'cases': cases},
              open(os.path.join(HERE, 'msm_g2_small.json'), 'w'))


def gen_ntt():
    rng = p.SplitMix64(0xA1E00003)
    cases = []
    for n in (1, 2, 4, 8, 64, 256):
        x = [rng.fr() for _ in range(n)]
        d = p.EvaluationDomain(n)
        cases.append({'n': n, 'input': [hex(v) for v in x], 'fft': [hex(v) for v in d.fft(x)], 'ifft': [hex(v) for v in d.ifft(x)],
                      'coset_fft': [hex(v) for v in d.coset_fft(x)], 'coset_ifft': [hex(v) for v in d.coset_ifft(x)]})
    # zero-padded input (fft of 5 coefficients over the size-8 domain), as EvaluationDomain::fft resizes
    x = [rng.fr() for _ in range(5)]; d = p.EvaluationDomain(5)
    cases.append({'n': 8, 'ragged': 5, 'input': [hex(v) for v in x], 'fft': [hex(v) for v in d.fft(x)], 'coset_fft': [hex(v) for v in d.coset_fft(x)]})
    json.dump({'two_adic_root': hex(p.FR_TWO_ADIC_ROOT), 'generator': p.FR_GENERATOR, 'cases': cases}, open(os.path.join(HERE, 'ntt_small.json'), 'w'))


def gen_varuna():
    from aleo_amd import synth
    from oracle import varuna_ref as V
    TAU, S_GAMMA = 0x1F3A9C0D5E7B24681357ACE02468BDF013579BDF02468ACE1234567, 0x0FEDCBA9876543210123456789ABCDEF55AA
    cases = []
    for (n, pub, seed, instances, domains) in [(12, 2, 101, 1, 'shared'), (60, 3, 102, 1, 'per_matrix'), (45, 4, 103, 3, 'per_matrix'), (200, 1, 104, 2, 'shared')]:
        csr, z = synth.synthetic_r1cs(n, pub, seed, long_rows=1 if n > 20 else 0)
        rows = lambda m: [[(int(csr[m][1][k]), synth.limbs_to_int(csr[m][2][k])) for k in range(csr[m][0][i], csr[m][0][i + 1])] for i in range(n)]
        c = V.Circuit(n, pub, len(z) - pub, rows('a'), rows('b'), rows('c'), domains=domains)
        D = 1
        while D < max(3 * c.n_h, c.n_k): D *= 2
        setup = V.Setup(TAU, S_GAMMA, D - 1); idx = V.Index(c, setup)
        zs = [z] + [synth.resolve_synthetic(csr, pub, [1] + [7 * i + j for j in range(1, pub)]) for i in range(1, instances)]
        proof, data = V.prove(idx, setup, zs, V.random_stream(seed + 1000, c.n_h, instances))
        assert V.verify_pairing(idx, setup.verifier_key(c), [q[:pub] for q in zs], data)
        cases.append({'n_constraints': n, 'n_public': pub, 'circuit_seed': seed, 'instances': instances, 'domains': domains, 'proof_seed': seed + 1000,
                      'other_publics': [[1] + [7 * i + j for j in range(1, pub)] for i in range(1, instances)], 'max_degree': D - 1,
                      'domain_sizes': [c.n_h, c.n_k_m['a'], c.n_k_m['b'], c.n_k_m['c'], c.n_x], 'vk': idx.vk_bytes().hex(), 'proof': data.hex()})
    # proofs over several of the circuits above (every instance of each member, in the order listed): Varuna::prove_batch with a map of proving keys
    batches = []
    for members, seed in [([1, 0, 3], 5001), ([2, 2], 5002)]:
        D = max(cases[j]['max_degree'] for j in members); setup = V.Setup(TAU, S_GAMMA, D); items = []
        for j in members:
            cj = cases[j]; n, pub = cj['n_constraints'], cj['n_public']
            csr, z = synth.synthetic_r1cs(n, pub, cj['circuit_seed'], long_rows=1 if n > 20 else 0)
            rows = lambda m: [[(int(csr[m][1][k]), synth.limbs_to_int(csr[m][2][k])) for k in range(csr[m][0][i], csr[m][0][i + 1])] for i in range(n)]
            c = V.Circuit(n, pub, len(z) - pub, rows('a'), rows('b'), rows('c'), domains=cj['domains'])
            items.append((V.Index(c, setup), [z] + [synth.resolve_synthetic(csr, pub, p_) for p_ in cj['other_publics']]))
        total = sum(len(zz) for _, zz in items)
        proof, data = V.prove_batch(items, setup, V.random_stream(seed, max(ix.circuit.n_h for ix, _ in items), total))
        assert V.verify_pairing([ix for ix, _ in items], setup.verifier_key([ix.circuit for ix, _ in items]), [[q[:ix.circuit.n_public] for q in zz] for ix, zz in items], data)
        batches.append({'members': members, 'proof_seed': seed, 'max_degree': D, 'proof': data.hex()})
    json.dump({'tau': hex(TAU), 's_gamma': hex(S_GAMMA), 'cases': cases, 'batches': batches}, open(os.path.join(HERE, 'varuna_small.json'), 'w'))


if __name__ == '__main__':
    gen_reference_proof(); gen_reference_literals(); gen_reference_account(); gen_msm(); gen_msm_g2(); gen_ntt(); gen_varuna()
    print('golden fixtures written to', HERE)

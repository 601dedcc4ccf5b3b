"""The AHP prover above the operators.  CPU: the restatement in oracle/varuna_ref.py proves, its verifier accepts,
tampering is rejected, the byte layout is the reference's.  GPU: aleo_amd.varuna produces the SAME proof bytes as the restatement from
the same circuit, assignment, randomness and setup; at sizes the restatement does not reach, the verifier accepts the device's proofs."""
import json, os
import numpy as np
import pytest
import aleo_amd
from aleo_amd import synth
from oracle import varuna_ref as V, pyref

TAU, S_GAMMA = 0x1F3A9C0D5E7B24681357ACE02468BDF013579BDF02468ACE1234567, 0x0FEDCBA9876543210123456789ABCDEF55AA


def _circuit(n_constraints, n_public, seed, long_rows=1, domains='auto'):
    csr, z = synth.synthetic_r1cs(n_constraints, n_public, seed, long_rows=long_rows)
    def rows(m):
        ptr, col, val = csr[m]
        return [[(int(col[k]), synth.limbs_to_int(val[k])) for k in range(ptr[i], ptr[i + 1])] for i in range(len(ptr) - 1)]
    c = V.Circuit(n_constraints, n_public, len(z) - n_public, rows('a'), rows('b'), rows('c'), domains=domains)
    return csr, z, c


def _max_degree(c):
    d = 1
    while d < max(3 * c.n_h, c.n_k): d *= 2
    return d - 1


def _rand(c, seed, k=1): return V.random_stream(seed, c.n_h, k)


@pytest.mark.parametrize('domains', ['shared', 'per_matrix'])
def test_restatement_proves_and_verifies(domains):
    csr, z, c = _circuit(40, 3, 11, domains=domains)
    assert len(set(c.n_k_m.values())) == (1 if domains == 'shared' else 2)
    setup = V.Setup(TAU, S_GAMMA, _max_degree(c)); idx = V.Index(c, setup)
    rand = _rand(c, 5)
    proof, data = V.prove(idx, setup, z, rand)
    assert len(data) == 901                      # the reference's proof string decodes to 901 bytes for one circuit, one instance
    assert V.verify(idx, setup, z[:3], data)
    vk = setup.verifier_key(c)                   # the verifier proper: pairing products over public G2 elements, no trapdoor
    assert V.verify_pairing(idx, vk, z[:3], data)
    for pos in (100, 520, 700, 800, 860):
        bad = bytearray(data); bad[pos] ^= 1
        assert not V.verify_pairing(idx, vk, z[:3], bytes(bad))
    assert not V.verify_pairing(idx, vk, [1, (z[1] + 1) % V.R, z[2]], data)
    # the layout is the reference's: the proof string it holds (wasm/src/programs/transaction.rs:100) parses with the same parser,
    # field for field (commitments at the offsets of tests/golden/reference_proof.json, the hiding value on the first opening only)
    ref = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'reference_proof.json')))
    hrp, payload = pyref.bech32m_decode(ref['proof'])
    assert hrp == 'proof' and len(payload) == len(data) == 901
    parsed = V.parse_proof(payload)
    found = dict(parsed['commitments'], mask_poly=parsed['commitments']['mask'], **dict(zip(('w', 'z_a', 'z_b'), parsed['witness'][0])))
    for name, ent in ref['commitments'].items():
        assert found[name].hex() == ent['compressed']
    assert [p_.hex() for p_, _ in parsed['openings']] == [o['compressed'] for o in ref['openings']]
    assert parsed['openings'][0][1] == int(ref['openings'][0]['random_v'], 16) and parsed['openings'][1][1] is None
    assert V.proof_bytes(parsed) == payload
    for pos in (20, 300, 500, 700, 800, 860):    # a commitment, evaluations, sums, both openings
        bad = bytearray(data); bad[pos] ^= 1
        assert not V.verify(idx, setup, z[:3], bytes(bad))
    assert not V.verify(idx, setup, [1, (z[1] + 1) % V.R, z[2]], data)
    z2 = list(z); z2[7] = (z2[7] + 1) % V.R     # an unsatisfied assignment cannot be proved
    with pytest.raises(AssertionError): V.prove(idx, setup, z2, rand)


def test_restatement_layout_round_trip():
    csr, z, c = _circuit(17, 2, 3, long_rows=0)
    setup = V.Setup(TAU, S_GAMMA, _max_degree(c)); idx = V.Index(c, setup)
    proof, data = V.prove(idx, setup, z, _rand(c, 9))
    back = V.parse_proof(data)
    assert back['commitments'] == proof['commitments'] and back['evaluations'] == proof['evaluations'] and back['sums'] == proof['sums']
    assert V.proof_bytes(back) == data
    assert pyref.bech32m_encode('proof', data).startswith('proof1')


def test_restatement_batch_of_instances():
    """Varuna::prove_batch shape: k instances of one circuit share the mask, g_1, h_1 and everything after; 3k + 1 first-round commitments,
    k + 4 evaluations.  The verifier needs every instance's public inputs, rejects a swapped pair and a tampered instance commitment."""
    csr, z, c = _circuit(33, 3, 4)
    zs = [z, synth.resolve_synthetic(csr, 3, [1, 5, 7]), synth.resolve_synthetic(csr, 3, [1, V.R - 2, 0])]
    assert zs[0] == synth.resolve_synthetic(csr, 3, z[:3])
    setup = V.Setup(TAU, S_GAMMA, _max_degree(c)); idx = V.Index(c, setup)
    proof, data = V.prove(idx, setup, zs, _rand(c, 31, 3))
    assert len(data) == 901 + 2 * (3 * 48 + 32) and proof['instances'] == 3
    pubs = [q[:3] for q in zs]
    assert V.verify(idx, setup, pubs, data) and V.verify_pairing(idx, setup.verifier_key(c), pubs, data)
    assert not V.verify(idx, setup, [pubs[1], pubs[0], pubs[2]], data)
    assert not V.verify(idx, setup, pubs[:2], data)
    bad = bytearray(data); bad[17 + 48 * 5 + 3] ^= 1                          # z_b of the second instance
    assert not V.verify(idx, setup, pubs, bytes(bad))
    assert V.parse_proof(data)['witness'] == proof['witness'] and V.proof_bytes(V.parse_proof(data)) == data


def _batch_case(shapes, seed=77, domains='auto'):
    """shapes: [(n_constraints, n_public, circuit_seed, instances), ...] -> (circuits, csrs, assignments per circuit, max degree)."""
    cs, csrs, zs = [], [], []
    for n, npub, cseed, k in shapes:
        csr, z, c = _circuit(n, npub, cseed, long_rows=1 if n > 8 else 0, domains=domains)
        cs.append(c); csrs.append(csr)
        zs.append([z] + [synth.resolve_synthetic(csr, npub, [1] + [(seed * 31 + 7 * i + t) % V.R for t in range(1, npub)]) for i in range(1, k)])
    d = 1
    while d < max(max(3 * c.n_h, c.n_k) for c in cs): d *= 2
    return cs, csrs, zs, d - 1


def test_restatement_batch_over_circuits():
    """Varuna::prove_batch over several circuits (different |H|, |K|, |X|, instance counts): one proof, accepted by both verifiers, refused when a
    byte, a public input or the order of the circuits changes; a one-circuit batch is prove() byte for byte."""
    shapes = [(20, 2, 5, 2), (50, 3, 6, 1), (9, 1, 7, 2)]
    cs, _, zs, D = _batch_case(shapes)
    assert len({c.n_h for c in cs}) == 3
    setup = V.Setup(TAU, S_GAMMA, D); idx = [V.Index(c, setup) for c in cs]
    rand = V.random_stream(123, max(c.n_h for c in cs), sum(len(z) for z in zs))
    proof, data = V.prove_batch(list(zip(idx, zs)), setup, rand)
    assert proof['batch_sizes'] == [2, 1, 2] and len(data) == 1 + 8 * 4 + 48 * (15 + 4 + 9) + 1 + 32 * (5 + 1 + 9) + 8 + 32 * 9 + 8 + 48 + 33 + 48 + 1 + 1
    assert V.parse_proof(data)['evaluations'] == proof['evaluations']
    pubs = [[z[:c.n_public] for z in zz] for c, zz in zip(cs, zs)]
    assert V.verify(idx, setup, pubs, data)
    assert V.verify_pairing(idx, setup.verifier_key(cs), pubs, data)
    bad = bytearray(data); bad[len(data) // 2] ^= 1
    assert not V.verify(idx, setup, pubs, bytes(bad))
    wrong = [[list(p_) for p_ in pp] for pp in pubs]; wrong[2][1][0] = 2
    assert not V.verify(idx, setup, wrong, data)
    assert not V.verify([idx[1], idx[0], idx[2]], setup, [pubs[1], pubs[0], pubs[2]], data)
    z_bad = [list(z) for z in zs[1]]; z_bad[0][9] = (z_bad[0][9] + 1) % V.R
    with pytest.raises(AssertionError): V.prove_batch([(idx[0], zs[0]), (idx[1], z_bad), (idx[2], zs[2])], setup, rand)
    # the largest circuit need not come first, and equal sizes are fine
    cs2, _, zs2, D2 = _batch_case([(30, 2, 8, 1), (30, 2, 9, 1)])
    setup2 = V.Setup(TAU, S_GAMMA, D2); idx2 = [V.Index(c, setup2) for c in cs2]
    _, d2 = V.prove_batch(list(zip(idx2, zs2)), setup2, V.random_stream(5, cs2[0].n_h, 2))
    assert V.verify(idx2, setup2, [[z[:2] for z in zz] for zz in zs2], d2)
    one, single = V.prove_batch([(idx2[0], zs2[0])], setup2, V.random_stream(6, cs2[0].n_h, 1))[1], V.prove(idx2[0], setup2, zs2[0][0], V.random_stream(6, cs2[0].n_h, 1))[1]
    assert one == single


@pytest.mark.gpu
@pytest.mark.parametrize('k', [2, 3, 4, 5, 8, 13])
def test_device_prover_batch_matches_restatement(k):
    from aleo_amd import varuna
    domains = 'per_matrix' if k % 2 else 'auto'
    csr, z, c = _circuit(150, 3, 40 + k, domains=domains)
    zs = [z] + [synth.resolve_synthetic(csr, 3, [1, 10 + i, 20 * i + 1]) for i in range(1, k)]
    D = _max_degree(c)
    setup = V.Setup(TAU, S_GAMMA, D); idx = V.Index(c, setup)
    _, want = V.prove(idx, setup, zs, _rand(c, 500 + k, k))
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    try:
        ix = varuna.CircuitIndex(csr, 150, 3, len(z) - 3, ck, domains=domains)
        got = varuna.prove(ix, [np.stack([synth.int_to_limbs(v, 4) for v in q]) for q in zs], 500 + k).to_bytes()
        assert got == want and V.verify(idx, setup, [q[:3] for q in zs], got)
        zq = [np.stack([synth.int_to_limbs(v, 4) for v in q]) for q in zs]
        assert varuna.prove_native(ix, zq, 500 + k) == want                            # the one-call C++ host side
        with varuna.NativeCircuitIndex(csr, 150, 3, len(z) - 3, ck, domains=domains) as nx: assert nx.prove(zq, 500 + k) == want
        with pytest.raises(ValueError): varuna.prove(ix, [np.zeros((len(z), 4), dtype=np.uint64)] * 33, 1)
    finally:
        ck.close()


@pytest.mark.gpu
@pytest.mark.parametrize('domains', ['shared', 'per_matrix'])
@pytest.mark.parametrize('n_constraints,n_public,seed', [(1, 1, 1), (2, 1, 2), (3, 2, 3), (8, 8, 4), (24, 3, 5), (31, 4, 9), (65, 3, 10), (100, 2, 6), (700, 5, 7), (2000, 9, 8), (900, 300, 11)])
def test_device_prover_matches_restatement(n_constraints, n_public, seed, domains):
    from aleo_amd import varuna
    csr, z, c = _circuit(n_constraints, n_public, seed, long_rows=1 if n_constraints > 8 else 0, domains=domains)
    D = _max_degree(c)
    setup = V.Setup(TAU, S_GAMMA, D); idx = V.Index(c, setup)
    want_proof, want = V.prove(idx, setup, z, _rand(c, seed + 100))
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    try:
        ix = varuna.CircuitIndex(csr, n_constraints, n_public, len(z) - n_public, ck, domains=domains)
        assert (ix.n_h, ix.n_k, ix.n_x) == (c.n_h, c.n_k, c.n_x) and ix.n_k_m == [c.n_k_m[m] for m in 'abc']
        assert ix.vk_bytes == idx.vk_bytes()
        zz = np.stack([synth.int_to_limbs(v, 4) for v in z])
        proof = varuna.prove(ix, zz, seed + 100)
        got = proof.to_bytes()
        assert got == want
        assert varuna.prove_native(ix, zz, seed + 100) == want                         # the one-call C++ host side
        with varuna.NativeCircuitIndex(csr, n_constraints, n_public, len(z) - n_public, ck, domains=domains) as nx:     # index built by the library itself
            assert (nx.n_h, nx.n_k_m, nx.n_x) == (c.n_h, [c.n_k_m[m] for m in 'abc'], c.n_x) and nx.vk_bytes == idx.vk_bytes()
            assert nx.prove(zz, seed + 100) == want
            handle = nx.handle
        with pytest.raises(aleo_amd.AleoMi355xError): aleo_amd._lib.check(aleo_amd.lib().aleo_mi355x_varuna_index_free(handle), 'index_free')   # already freed
        assert V.verify(idx, setup, z[:n_public], got)
        assert proof.to_string() == pyref.bech32m_encode('proof', want)
    finally:
        ck.close()


@pytest.mark.gpu
def test_device_prover_verifies_at_2_13():
    from aleo_amd import varuna
    n = 1 << 13
    csr, z, c = _circuit(n - 50, 4, 21, long_rows=3, domains='per_matrix')
    D = _max_degree(c)
    setup = V.Setup(TAU, S_GAMMA, D); idx = V.Index(c, setup)
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    try:
        ix = varuna.CircuitIndex(csr, n - 50, 4, len(z) - 4, ck, domains='per_matrix')
        assert ix.vk_bytes == idx.vk_bytes()
        zz = np.stack([synth.int_to_limbs(v, 4) for v in z])
        data = varuna.prove(ix, zz, 77).to_bytes()
        assert len(data) == 901 and V.verify(idx, setup, z[:4], data)
        assert V.verify_pairing(idx, setup.verifier_key(c), z[:4], data)          # and by the verifier that holds no trapdoor
        bad = bytearray(data); bad[600] ^= 4
        assert not V.verify(idx, setup, z[:4], bytes(bad))
        z2 = zz.copy(); z2[100, 0] ^= np.uint64(1)                       # a wrong witness never becomes a proof (test_unsatisfied_assignment_is_refused)
        with pytest.raises(aleo_amd.UnsatisfiedAssignment): varuna.prove(ix, z2, 77)
    finally:
        ck.close()


@pytest.mark.gpu
def test_device_prover_concurrent_streams():
    """Proofs of one index from three host threads, each on its own stream, equal the proofs produced one after the other (the index is
    read-only while proving; scratch inside the library is per call slot and stream-ordered)."""
    import threading, torch
    from aleo_amd import varuna
    csr, z, c = _circuit(900, 3, 61)
    D = _max_degree(c)
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    try:
        ix = varuna.CircuitIndex(csr, 900, 3, len(z) - 3, ck)
        zz = np.stack([synth.int_to_limbs(v, 4) for v in z])
        want = {(t, r): varuna.prove(ix, zz, 9000 + 10 * t + r).to_bytes() for t in range(3) for r in range(3)}
        got, err = {}, []
        def work(t):
            try:
                st = torch.cuda.Stream()
                for r in range(3): got[(t, r)] = varuna.prove(ix, zz, 9000 + 10 * t + r, st).to_bytes()
            except Exception as e: err.append(e)
        th = [threading.Thread(target=work, args=(t,)) for t in range(3)]
        for x in th: x.start()
        for x in th: x.join()
        assert not err and got == want
        setup = V.Setup(TAU, S_GAMMA, D); idx = V.Index(c, setup)
        assert V.verify(idx, setup, z[:3], got[(2, 2)])
    finally:
        ck.close()


@pytest.mark.gpu
def test_whole_proof_entry_points_from_plain_cpp(tmp_path):
    """tests/cpp/varuna_prove_test.cpp: key pinned from scalars, index built, two instances proved through the C ABI alone (no Python in the
    process); its proof and verifier-key bytes equal the restatement's; misuse (short buffer, freed index) is refused."""
    import struct, subprocess, sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_abi import build_cpp_host_mirror
    csr, z, c = _circuit(300, 3, 71, domains='per_matrix')
    zs = [z, synth.resolve_synthetic(csr, 3, [1, 2, 3])]
    D = _max_degree(c); ng = 3; seed = 31337
    setup = V.Setup(TAU, S_GAMMA, D); idx = V.Index(c, setup)
    _, want = V.prove(idx, setup, zs, _rand(c, seed, 2))
    sc, a = [], 1
    for i in range(D + 1): sc.append(a); a = a * TAU % V.R
    a = S_GAMMA % V.R
    for i in range(ng): sc.append(a); a = a * TAU % V.R
    blob = struct.pack('<8Q', 300, 3, len(z) - 3, D, ng, seed, 2, 1) + synth.generator_affine104().tobytes()      # last header field: domain flags (1 = per matrix)
    blob += b''.join(int(v).to_bytes(32, 'little') for v in sc)
    for m in 'abc':
        ptr, col, val = csr[m]
        blob += struct.pack('<Q', int(ptr[-1])) + np.ascontiguousarray(ptr, dtype=np.uint32).tobytes() + np.ascontiguousarray(col, dtype=np.uint32).tobytes()
        blob += np.ascontiguousarray(val, dtype=np.uint64).tobytes()
    for q in zs: blob += b''.join(int(v).to_bytes(32, 'little') for v in q)
    fin, fout = os.path.join(str(tmp_path), 'in.bin'), os.path.join(str(tmp_path), 'out.bin')
    open(fin, 'wb').write(blob)
    exe = build_cpp_host_mirror(tmp_path, 'varuna_prove_test')
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'ALL OK' in r.stdout, r.stdout + r.stderr
    out = open(fout, 'rb').read(); vk_len, plen = struct.unpack('<2Q', out[:16])
    assert out[16:16 + vk_len] == idx.vk_bytes() and out[16 + vk_len:16 + vk_len + plen] == want
    assert V.verify(idx, setup, [q[:3] for q in zs], out[16 + vk_len:16 + vk_len + plen])
    # the C++ Trace mirror: one proof for (key 1: every instance, key 2 of the same circuit: the first instance)
    at = 16 + vk_len + plen; elen, = struct.unpack('<Q', out[at:at + 8]); execution = out[at + 8:at + 8 + elen]
    n_inst = len(zs) + 1
    assert execution == V.prove_batch([(idx, zs), (idx, zs[:1])], setup, V.random_stream(seed, c.n_h, n_inst))[1]
    assert V.verify([idx, idx], setup, [[q[:3] for q in zs], [zs[0][:3]]], execution)


@pytest.mark.gpu
def test_whole_proof_entry_points_refuse_misuse():
    """Bad arguments come back as error codes (the caller falls back to its CPU prover), never as a fault: instance counts, inconsistent
    index structs, non-canonical public inputs, foreign handles, malformed matrices, a committer key that is too small."""
    import ctypes
    from aleo_amd import varuna
    L = aleo_amd.lib()
    csr, z, c = _circuit(50, 2, 81)
    D = _max_degree(c)
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    try:
        ix = varuna.CircuitIndex(csr, 50, 2, len(z) - 2, ck)
        zz = np.stack([synth.int_to_limbs(v, 4) for v in z])
        good = varuna.prove_native(ix, zz, 5)
        for k in (0, 33):
            with pytest.raises((aleo_amd.AleoMi355xError, ValueError, IndexError)): varuna.prove_native(ix, [zz] * k, 5)
        view = varuna.native_index(ix)
        out = np.zeros(2048, dtype=np.uint8); n = ctypes.c_size_t(2048); ptrs = (ctypes.c_void_p * 1)(zz.ctypes.data)
        def call(v): n.value = 2048; return L.aleo_mi355x_varuna_prove(ctypes.byref(v), ptrs, 1, aleo_amd._lib.seed32(5), out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
        def variant(**kw):
            v = varuna._NativeIndex(); ctypes.memmove(ctypes.byref(v), ctypes.byref(view), ctypes.sizeof(v))
            for a, b in kw.items(): setattr(v, a, b)
            return v
        assert call(variant()) == 0 and out[:n.value].tobytes() == good
        assert call(variant(n_h=view.n_h - 1)) == 2 and call(variant(n_vars=view.n_h + 1)) == 2 and call(variant(n_x=view.n_h)) == 2
        assert call(variant(max_degree=view.n_h)) == 2 and call(variant(gamma_offset=1 << 40)) == 2
        assert call(variant(committer_key=987654321)) == 4
        for field in ('a_row_ptr', 'b_val', 't_col', 'vx_inv', 'k_evals', 'k_idx', 'k_polys', 'k2_evals'):      # a null array is an error code, not a GPU fault
            assert call(variant(**{field: None})) == 2, field
        assert call(variant(vk_affine=None)) == 0 and out[:n.value].tobytes() == good                          # without the affine form: decompressed from vk_bytes
        assert L.aleo_mi355x_varuna_prove(ctypes.byref(variant()), ptrs, 1, None, out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)) == 2      # no seed
        bad = zz.copy(); bad[1] = synth.int_to_limbs(V.R, 4)                       # a public input that is not below r
        ptrs[0] = bad.ctypes.data; assert call(variant()) == 2; ptrs[0] = zz.ctypes.data
        def build(rows_a=None, **kw):
            mats = (varuna._R1csMatrix * 3)(); keep = []
            for i, m in enumerate('abc'):
                rp, col, val = (np.ascontiguousarray(x).copy() for x in csr[m])
                if i == 0 and rows_a: rows_a(rp, col)
                keep += [rp, col, val]; mats[i].row_ptr, mats[i].col, mats[i].val = rp.ctypes.data, col.ctypes.data, val.ctypes.data
            h = ctypes.c_uint64(0)
            a = dict(key=ck.bases.handle, max_degree=ck.max_degree, gamma_offset=ck.gamma_offset, n=50, pub=2, priv=len(z) - 2); a.update(kw)
            rc = L.aleo_mi355x_varuna_index_build(ctypes.byref(h), a['key'], a['max_degree'], a['gamma_offset'], a.get('lagrange', 0), mats, a['n'], a['pub'], a['priv'], a.get('flags', 0))
            if rc == 0: L.aleo_mi355x_varuna_index_free(h.value)
            return rc
        assert build() == 0
        def shift(rp, col): rp[0] = 1
        def wild(rp, col): col[3] = 10 ** 6
        def back(rp, col): rp[5] = rp[6] + 1
        assert build(shift) == 2 and build(wild) == 2 and build(back) == 2
        assert build(flags=3) == 2 and build(flags=1) == 0 and build(flags=2) == 0
        assert build(lagrange=ck.max_degree + 4) == 2                          # Lagrange powers announced where the key has none
        assert build(max_degree=7) == 2 and build(key=123456789) == 4 and build(pub=0) == 2 and build(priv=len(z) + 10 ** 6) == 2
    finally:
        ck.close()


@pytest.mark.gpu
@pytest.mark.parametrize('shapes,domains', [
    ([(20, 2, 5, 2), (50, 3, 6, 1), (9, 1, 7, 2)], 'auto'),                       # three sizes of H, the largest in the middle
    ([(300, 4, 8, 1), (300, 4, 9, 1)], 'per_matrix'),                              # equal domains
    ([(40, 2, 10, 3), (700, 5, 11, 2), (1, 1, 12, 1), (130, 3, 13, 8)], 'auto'),   # 14 instances: linear combinations beyond one launch (28 terms)
    ([(2000, 9, 14, 1), (24, 3, 15, 1)], 'per_matrix'),
    ([(30, 2, 16, 8), (12, 1, 17, 8), (60, 3, 18, 8), (30, 2, 19, 8)], 'auto')])   # the most a proof takes: 32 instances, 97 first-round commitments in one call
def test_device_prover_batch_over_circuits(shapes, domains):
    """aleo_mi355x_varuna_prove_batch_indexed: one proof over several circuits, byte for byte the restatement's, accepted by its verifier; an
    assignment that violates any one circuit is refused; a one-circuit batch equals the single-circuit entry point."""
    from aleo_amd import varuna
    cs, csrs, zs, D = _batch_case(shapes, domains=domains)
    setup = V.Setup(TAU, S_GAMMA, D); idx = [V.Index(c, setup) for c in cs]
    seed = 900 + len(shapes)
    rand = V.random_stream(seed, max(c.n_h for c in cs), sum(len(z) for z in zs))
    want = V.prove_batch(list(zip(idx, zs)), setup, rand)[1]
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    lim = lambda a: np.stack([synth.int_to_limbs(v, 4) for v in a])
    nx = []
    try:
        for (n, npub, _, _), csr, z in zip(shapes, csrs, zs): nx.append(varuna.NativeCircuitIndex(csr, n, npub, len(z[0]) - npub, ck, domains=domains))
        assert [x.vk_bytes for x in nx] == [i.vk_bytes() for i in idx]
        za = [[lim(z) for z in zz] for zz in zs]
        got = varuna.prove_batch_native(nx, za, seed)
        assert got == want
        assert V.verify(idx, setup, [[z[:c.n_public] for z in zz] for c, zz in zip(cs, zs)], got)
        assert varuna.prove_batch_native(nx[:1], za[:1], seed) == nx[0].prove(za[0], seed)
        bad = [list(zz) for zz in za]; t = len(shapes) - 1
        bad[t][-1] = bad[t][-1].copy(); bad[t][-1][-1, 0] ^= np.uint64(1)              # the last variable of the last instance of the last circuit
        with pytest.raises(aleo_amd.UnsatisfiedAssignment): varuna.prove_batch_native(nx, bad, seed)
        assert varuna.prove_batch_native(nx, za, seed) == want
        reps = 32 // len(za[0]) + 1
        with pytest.raises(aleo_amd.AleoMi355xError): varuna.prove_batch_native(nx[:1] * reps, za[:1] * reps, seed)      # more than 32 instances in all
        tr = varuna.Trace()                                                             # the Trace mirror groups transitions per key, in order of first appearance
        order = [(j, i) for i in range(max(len(zz) for zz in za)) for j in range(len(za)) if i < len(za[j])]      # interleaved: circuit 0, 1, 2, 0, 1, ...
        for j, i in order: tr.insert_transition(nx[j], za[j][i])
        assert tr.transitions == sum(len(zz) for zz in za) and tr.prove_execution(seed) == want
        with pytest.raises(ValueError): tr.prove_fee(seed)
        fee = varuna.Trace(); fee.insert_transition(nx[0], za[0][0])
        assert fee.prove_fee(seed) == nx[0].prove(za[0][0], seed)
    finally:
        for x in nx: x.close()
        ck.close()


@pytest.mark.gpu
def test_unsatisfied_assignment_is_refused():
    """An assignment that violates a constraint must not turn into bytes that look like a proof: both provers stop at the first sumcheck
    (sum over H != 0) — the one-call entry with ALEO_MI355X_ERR_UNSATISFIED (6), also when only one instance of a batch is bad — and the
    slot keeps working afterwards.  (Upstream refuses earlier, at synthesis: `A::is_satisfied()`; the restatement asserts the same sum.)"""
    from aleo_amd import varuna
    csr, z, c = _circuit(120, 3, 4)
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, _max_degree(c))
    try:
        ix = varuna.CircuitIndex(csr, 120, 3, len(z) - 3, ck)
        lim = lambda a: np.stack([synth.int_to_limbs(v, 4) for v in a])
        bad = list(z); bad[9] = (bad[9] + 1) % V.R
        good = varuna.prove_native(ix, lim(z), 5)
        with pytest.raises(aleo_amd.UnsatisfiedAssignment) as e: varuna.prove_native(ix, lim(bad), 5)
        assert e.value.status == 6 and 'satisfy' in str(e.value)
        with pytest.raises(aleo_amd.UnsatisfiedAssignment): varuna.prove_native(ix, [lim(z), lim(bad), lim(z)], 5)
        with pytest.raises(aleo_amd.UnsatisfiedAssignment): varuna.prove(ix, lim(bad), 5)
        assert varuna.prove_native(ix, lim(z), 5) == good and varuna.prove(ix, lim(z), 5).to_bytes() == good
    finally:
        ck.close()


def _golden_cases():
    g = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'varuna_small.json')))
    return int(g['tau'], 16), int(g['s_gamma'], 16), g['cases']


def _golden_instance(case):
    csr, z, c = _circuit(case['n_constraints'], case['n_public'], case['circuit_seed'], long_rows=1 if case['n_constraints'] > 20 else 0, domains=case['domains'])
    zs = [z] + [synth.resolve_synthetic(csr, case['n_public'], pub) for pub in case['other_publics']]
    assert [c.n_h, c.n_k_m['a'], c.n_k_m['b'], c.n_k_m['c'], c.n_x] == case['domain_sizes']
    return csr, zs, c


def test_restatement_reproduces_the_frozen_proofs():
    """tests/golden/varuna_small.json (written by gen_golden.py from the restatement): today's restatement gives the same verifying-key and
    proof bytes, and its pairing verifier accepts them — so a change of either side of the parity tests shows up here first."""
    tau, sg, cases = _golden_cases()
    assert (tau, sg) == (TAU, S_GAMMA)
    for case in cases:
        csr, zs, c = _golden_instance(case)
        setup = V.Setup(tau, sg, case['max_degree']); idx = V.Index(c, setup)
        assert idx.vk_bytes().hex() == case['vk']
        _, data = V.prove(idx, setup, zs, V.random_stream(case['proof_seed'], c.n_h, case['instances']))
        assert data.hex() == case['proof']
    csr, zs, c = _golden_instance(cases[1])
    setup = V.Setup(tau, sg, cases[1]['max_degree'])
    assert V.verify_pairing(V.Index(c, setup), setup.verifier_key(c), [q[:cases[1]['n_public']] for q in zs], bytes.fromhex(cases[1]['proof']))


def _golden_batches():
    g = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'varuna_small.json')))
    return g['cases'], g['batches']


def test_restatement_reproduces_the_frozen_batch_proofs():
    """The frozen proofs over several circuits (members = cases of the same file, every instance of each)."""
    cases, batches = _golden_batches()
    assert batches
    for b in batches:
        setup = V.Setup(TAU, S_GAMMA, b['max_degree']); items = []
        for j in b['members']:
            csr, zs, c = _golden_instance(cases[j]); items.append((V.Index(c, setup), zs))
        total = sum(len(zz) for _, zz in items)
        data = V.prove_batch(items, setup, V.random_stream(b['proof_seed'], max(ix.circuit.n_h for ix, _ in items), total))[1]
        assert data.hex() == b['proof']
        pubs = [[q[:ix.circuit.n_public] for q in zz] for ix, zz in items]
        assert V.verify([ix for ix, _ in items], setup, pubs, data)
        # a verifier that holds only the verifying keys (index commitments + domain sizes), no index
        vks = [V.VerifyingKey(bytes.fromhex(cases[j]['vk']), cases[j]['n_public']) for j in b['members']]
        assert V.verify(vks, setup, pubs, data)
        if len(set(b['members'])) > 1: assert not V.verify(vks[::-1], setup, pubs[::-1], data)
    assert V.verify_pairing(vks, setup.verifier_key([v.circuit for v in vks]), pubs, data)


@pytest.mark.gpu
def test_device_prover_reproduces_the_frozen_batch_proofs():
    from aleo_amd import varuna
    cases, batches = _golden_batches()
    for b in batches:
        ck = varuna.synthetic_committer_key(TAU, S_GAMMA, b['max_degree']); nx, za = [], []
        try:
            for j in b['members']:
                case = cases[j]; csr, zs, c = _golden_instance(case)
                nx.append(varuna.NativeCircuitIndex(csr, case['n_constraints'], case['n_public'], len(zs[0]) - case['n_public'], ck, domains=case['domains']))
                assert nx[-1].vk_bytes.hex() == case['vk']
                za.append([np.stack([synth.int_to_limbs(v, 4) for v in q]) for q in zs])
            assert varuna.prove_batch_native(nx, za, b['proof_seed']).hex() == b['proof']
        finally:
            for x in nx: x.close()
            ck.close()


@pytest.mark.gpu
def test_device_provers_reproduce_the_frozen_proofs():
    """The frozen proofs from the native index + prover, and from the step-by-step host side."""
    from aleo_amd import varuna
    tau, sg, cases = _golden_cases()
    for case in cases:
        csr, zs, c = _golden_instance(case)
        zq = [np.stack([synth.int_to_limbs(v, 4) for v in q]) for q in zs]
        ck = varuna.synthetic_committer_key(tau, sg, case['max_degree'])
        try:
            with varuna.NativeCircuitIndex(csr, case['n_constraints'], case['n_public'], len(zs[0]) - case['n_public'], ck, domains=case['domains']) as nx:
                assert nx.vk_bytes.hex() == case['vk'] and nx.prove(zq, case['proof_seed']).hex() == case['proof']
            ix = varuna.CircuitIndex(csr, case['n_constraints'], case['n_public'], len(zs[0]) - case['n_public'], ck, domains=case['domains'])
            assert ix.vk_bytes.hex() == case['vk'] and varuna.prove(ix, zq, case['proof_seed']).to_bytes().hex() == case['proof']
        finally:
            ck.close()


@pytest.mark.gpu
@pytest.mark.parametrize('k,range_window', [(1, 0), (3, 0), (2, 13), (8, 13)])
def test_commit_lagrange_in_the_first_round(k, range_window):
    """With the Lagrange-basis powers of H pinned behind the key, w, z_a, z_b are committed from their evaluations (KZG10::commit_lagrange with the
    blinding term against v_H(tau) G): the same group elements, hence the same proof bytes as the restatement's; bit-heavy circuit."""
    from aleo_amd import varuna
    csr, z = synth.synthetic_r1cs_bits(500, 3, 90 + k)
    rows = lambda m: [[(int(csr[m][1][j]), synth.limbs_to_int(csr[m][2][j])) for j in range(csr[m][0][i], csr[m][0][i + 1])] for i in range(500)]
    c = V.Circuit(500, 3, len(z) - 3, rows('a'), rows('b'), rows('c'))
    zs = [z] + [synth.resolve_synthetic(csr, 3, [1, i & 1, (i >> 1) & 1]) for i in range(1, k)]
    D = _max_degree(c); setup = V.Setup(TAU, S_GAMMA, D); idx = V.Index(c, setup)
    _, want = V.prove(idx, setup, zs, _rand(c, 640 + k, k))
    assert V.verify(idx, setup, [q[:3] for q in zs], want)
    zq = [np.stack([synth.int_to_limbs(v, 4) for v in q]) for q in zs]
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D, lagrange_size=c.n_h, range_window=range_window)      # with a window: the witness commitments as a sparse chain
    try:
        assert ck.lagrange_offset == D + 1 + 3 and ck.sparse_range == bool(range_window)
        with varuna.NativeCircuitIndex(csr, 500, 3, len(z) - 3, ck) as nx: assert nx.vk_bytes == idx.vk_bytes() and nx.prove(zq, 640 + k) == want
        ix = varuna.CircuitIndex(csr, 500, 3, len(z) - 3, ck)
        assert varuna.native_index(ix).lagrange_offset == ck.lagrange_offset
        assert varuna.prove(ix, zq, 640 + k).to_bytes() == want and varuna.prove_native(ix, zq, 640 + k) == want
    finally:
        ck.close()


@pytest.mark.gpu
@pytest.mark.parametrize('with_small,range_window', [(False, 0), (False, 13), (True, 13), (True, 0)])
def test_commit_lagrange_in_a_proof_over_circuits(with_small, range_window):
    """Several circuits with Lagrange powers pinned for |H| = 512: two bit-heavy circuits on that domain commit their witness polynomials from
    evaluations (with a range window: the sparse chain + the mask's own chain); with a smaller circuit in the proof — which has no Lagrange
    powers of its size and commits coefficients — everything goes through one mixed chain.  Bytes equal the restatement's either way."""
    from aleo_amd import varuna
    def bits(n, seed, k):
        csr, z = synth.synthetic_r1cs_bits(n, 3, seed)
        rows = lambda m: [[(int(csr[m][1][j]), synth.limbs_to_int(csr[m][2][j])) for j in range(csr[m][0][i], csr[m][0][i + 1])] for i in range(n)]
        return csr, [z] + [synth.resolve_synthetic(csr, 3, [1, i & 1, (i >> 1) & 1]) for i in range(1, k)], V.Circuit(n, 3, len(z) - 3, rows('a'), rows('b'), rows('c')), n
    parts = [bits(500, 301, 2), bits(400, 302, 1)] + ([bits(90, 303, 2)] if with_small else [])
    cs = [p[2] for p in parts]
    assert cs[0].n_h == cs[1].n_h == 512 and (not with_small or cs[2].n_h == 128)
    D = 1
    while D < max(max(3 * c.n_h, c.n_k) for c in cs): D *= 2
    D -= 1
    setup = V.Setup(TAU, S_GAMMA, D); idx = [V.Index(c, setup) for c in cs]
    zs = [p[1] for p in parts]; seed = 1200 + range_window + with_small
    want = V.prove_batch(list(zip(idx, zs)), setup, V.random_stream(seed, 512, sum(len(z) for z in zs)))[1]
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D, lagrange_size=512, range_window=range_window)
    nx = []
    try:
        for csr, z, c, n in parts: nx.append(varuna.NativeCircuitIndex(csr, n, 3, len(z[0]) - 3, ck))
        za = [[np.stack([synth.int_to_limbs(v, 4) for v in q]) for q in z] for z in zs]
        assert varuna.prove_batch_native(nx, za, seed) == want
        assert V.verify(idx, setup, [[q[:3] for q in z] for z in zs], want)
    finally:
        for x in nx: x.close()
        ck.close()


@pytest.mark.gpu
def test_device_proofs_of_large_circuits_are_verified():
    """Beyond the sizes the restatement's prover reaches: a proof over a 2^18-constraint bit-heavy circuit (two instances) and a 2^15-constraint
    one, made by the device, is accepted by the restatement's VERIFIER holding only the verifying keys the library exported (trapdoor-free pairing
    check included), and refused for a changed public input or a flipped byte.  (The single-circuit 2^20 proof of bench.py is checked the same way
    in its cpu_baseline leg.)"""
    from aleo_amd import varuna
    n1, n2 = (1 << 18) - 64, (1 << 15) - 64
    csr1, z1 = synth.synthetic_r1cs_bits(n1, 4, 501)
    csr2, z2 = synth.synthetic_r1cs(n2, 3, 502, long_rows=4)
    D = (1 << 21) - 1
    nnz = max(int(csr1[m][0][-1]) for m in 'abc'); assert nnz <= D + 1 and 3 * (1 << 18) <= D + 1
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    lim = lambda a: np.stack([synth.int_to_limbs(v, 4) for v in a])
    try:
        with varuna.NativeCircuitIndex(csr1, n1, 4, len(z1) - 4, ck) as x1, varuna.NativeCircuitIndex(csr2, n2, 3, len(z2) - 3, ck) as x2:
            zz1, zz2 = lim(z1), lim(z2)
            data = varuna.prove_batch_native([x1, x2], [[zz1, zz1], [zz2]], 77)
            vks = [V.VerifyingKey(x1.vk_bytes, 4), V.VerifyingKey(x2.vk_bytes, 3)]
            assert (vks[0].circuit.n_h, vks[1].circuit.n_h) == (1 << 18, 1 << 15)
            single = x1.prove(zz1, 78)
        setup = V.Setup(TAU, S_GAMMA, D); pubs = [[z1[:4], z1[:4]], [z2[:3]]]
        assert V.verify(vks, setup, pubs, data)
        assert V.verify_pairing(vks, setup.verifier_key([v.circuit for v in vks]), pubs, data)
        assert V.verify(vks[0], setup, z1[:4], single)
        wrong = [[list(z1[:4]), list(z1[:4])], [list(z2[:3])]]; wrong[0][1][2] ^= 1
        assert not V.verify(vks, setup, wrong, data)
        bad = bytearray(data); bad[-60] ^= 8
        assert not V.verify(vks, setup, pubs, bytes(bad))
    finally:
        ck.close()


@pytest.mark.gpu
def test_batch_proofs_from_concurrent_callers():
    """Proofs over several circuits from four host threads at once (each call on its own slot of the library, the indexes shared read-only)
    equal the proofs made one after the other; single-circuit proofs of the same indexes run in between."""
    import threading
    from aleo_amd import varuna
    cs, csrs, zs, D = _batch_case([(700, 3, 41, 2), (90, 2, 42, 1), (300, 4, 43, 1)])
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    lim = lambda a: np.stack([synth.int_to_limbs(v, 4) for v in a])
    nx = []
    try:
        for (n, npub, _, _), csr, z in zip([(700, 3, 41, 2), (90, 2, 42, 1), (300, 4, 43, 1)], csrs, zs): nx.append(varuna.NativeCircuitIndex(csr, n, npub, len(z[0]) - npub, ck))
        za = [[lim(z) for z in zz] for zz in zs]
        want = {(t, r): varuna.prove_batch_native(nx, za, 100 * t + r) for t in range(4) for r in range(3)}
        single = {t: nx[t % 3].prove(za[t % 3], 7 + t) for t in range(4)}
        got, got1, err = {}, {}, []
        def work(t):
            try:
                for r in range(3):
                    got[(t, r)] = varuna.prove_batch_native(nx, za, 100 * t + r)
                    if r == 1: got1[t] = nx[t % 3].prove(za[t % 3], 7 + t)
            except Exception as e: err.append(e)
        th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        for x in th: x.start()
        for x in th: x.join()
        assert not err, err
        assert got == want and got1 == single
    finally:
        for x in nx: x.close()
        ck.close()


@pytest.mark.gpu
def test_frozen_proofs_with_the_alternative_kernel_paths():
    """ALEO_MI355X_QUAD_ADD=0 (lane-pair additions in the reduction chains) and ALEO_MI355X_NTT_WIDE_LG=0 (register-group tiles for small transforms)
    are read once per process: a child process proves every frozen case of tests/golden/varuna_small.json with both switched off
    (tests/helpers/ab_switch_check.py), so the paths the defaults no longer take stay byte-exact."""
    import subprocess, sys
    env = dict(os.environ, ALEO_MI355X_QUAD_ADD='0', ALEO_MI355X_NTT_WIDE_LG='0', ALEO_MI355X_SUM_TREE='0', ALEO_MI355X_ASIDE='0', ALEO_MI355X_NTT29='0',
               ALEO_MI355X_CHAIN_OVERLAP='0', ALEO_MI355X_CHUNK_FORM='1', ALEO_MI355X_LOCKSTEP_WORKERS='1')      # + the round-3 switches
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers', 'ab_switch_check.py')
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and 'SWITCHES OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_lockstep_as_one_group_with_worker_threads():
    """ALEO_MI355X_LOCKSTEP_GROUPS=1 (read once per process): the whole call as ONE lockstep group with worker threads on borrowed contexts — the path small calls took
    before round 5 made one proof per group the default up to four proofs.  A child process (tests/helpers/lockstep_one_group_check.py) proves frozen cases 2, 3, 5 and 8
    at a time under different seeds and compares with the single-proof entry point and the frozen bytes."""
    import subprocess, sys
    env = dict(os.environ, ALEO_MI355X_LOCKSTEP_GROUPS='1')
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers', 'lockstep_one_group_check.py')
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and 'ONE GROUP OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_batch_entry_point_refuses_misuse():
    """aleo_mi355x_varuna_prove_batch_indexed: indexes built against different committer keys, a freed index, zero or nine instances of a circuit, a null
    assignment pointer and an output buffer that is too small all come back as error codes; the next good call is unaffected."""
    import ctypes
    from aleo_amd import varuna
    L = aleo_amd.lib()
    cs, csrs, zs, D = _batch_case([(40, 2, 51, 1), (25, 2, 52, 2)])
    ck, ck2 = varuna.synthetic_committer_key(TAU, S_GAMMA, D), varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    lim = lambda a: np.stack([synth.int_to_limbs(v, 4) for v in a])
    try:
        a = varuna.NativeCircuitIndex(csrs[0], 40, 2, len(zs[0][0]) - 2, ck); b = varuna.NativeCircuitIndex(csrs[1], 25, 2, len(zs[1][0]) - 2, ck)
        b2 = varuna.NativeCircuitIndex(csrs[1], 25, 2, len(zs[1][0]) - 2, ck2)
        za = [[lim(z) for z in zz] for zz in zs]
        good = varuna.prove_batch_native([a, b], za, 3)
        with pytest.raises(aleo_amd.AleoMi355xError): varuna.prove_batch_native([a, b2], za, 3)          # two committer keys in one proof
        flat = [z for zz in za for z in zz]
        def call(handles, counts, ptr_list, cap=4096):
            h = (ctypes.c_uint64 * len(handles))(*handles); k = (ctypes.c_size_t * len(counts))(*counts); p = (ctypes.c_void_p * len(ptr_list))(*ptr_list)
            out = np.zeros(max(cap, 8), dtype=np.uint8); n = ctypes.c_size_t(cap)
            return L.aleo_mi355x_varuna_prove_batch_indexed(h, len(handles), p, k, aleo_amd._lib.seed32(3), out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)), n.value, out
        ptrs = [z.ctypes.data for z in flat]
        rc, n, out = call([a.handle, b.handle], [1, 2], ptrs); assert rc == 0 and out[:n].tobytes() == good
        assert call([a.handle, b.handle], [0, 2], ptrs)[0] == 2 and call([a.handle, b.handle], [1, 32], ptrs * 11)[0] == 2      # 33 instances
        assert call([a.handle, b.handle], [1, 2], [ptrs[0], None, ptrs[2]])[0] == 2
        rc, n, _ = call([a.handle, b.handle], [1, 2], ptrs, cap=100); assert rc == 2 and n == len(good)      # too small: the size needed comes back
        hb = b.handle; b.close()
        assert call([a.handle, hb], [1, 2], ptrs)[0] == 4                                                    # freed index
        b = varuna.NativeCircuitIndex(csrs[1], 25, 2, len(zs[1][0]) - 2, ck)
        assert varuna.prove_batch_native([a, b], za, 3) == good
        for x in (a, b, b2): x.close()
    finally:
        ck.close(); ck2.close()


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [1, 2, 3, 4, 5, 6])
def test_device_prover_random_batches(seed):
    """A seeded sweep over the shape of a proof: 1–4 circuits of 1–400 constraints, 1–6 public inputs, 1–3 instances each, either domain policy —
    the device proof equals the restatement's and verifies."""
    import random
    from aleo_amd import varuna
    rnd = random.Random(7700 + seed)
    shapes = [(rnd.choice([1, 2, 5, 17, 33, 64, 100, 129, 255, 400]), rnd.randint(1, 6), 600 + 10 * seed + j, rnd.randint(1, 3)) for j in range(rnd.randint(1, 4))]
    domains = rnd.choice(['auto', 'per_matrix', 'shared'])
    cs, csrs, zs, D = _batch_case(shapes, seed=seed, domains=domains)
    setup = V.Setup(TAU, S_GAMMA, D); idx = [V.Index(c, setup) for c in cs]
    want = V.prove_batch(list(zip(idx, zs)), setup, V.random_stream(seed, max(c.n_h for c in cs), sum(len(z) for z in zs)))[1]
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D); nx = []
    try:
        for (n, npub, _, _), csr, z in zip(shapes, csrs, zs): nx.append(varuna.NativeCircuitIndex(csr, n, npub, len(z[0]) - npub, ck, domains=domains))
        got = varuna.prove_batch_native(nx, [[np.stack([synth.int_to_limbs(v, 4) for v in z]) for z in zz] for zz in zs], seed)
        assert got == want, shapes
        assert V.verify(idx, setup, [[z[:c.n_public] for z in zz] for c, zz in zip(cs, zs)], got)
    finally:
        for x in nx: x.close()
        ck.close()


def test_random_stream_positions_are_never_shared():
    """Every use of the proof's random stream has its own positions: rho_w / rho_a / rho_b and the three hiding polynomials of every instance (over
    all circuits of a proof — the layout is taken over the total instance count and the largest |H|), the 3|H| mask coefficients, the mask's
    hiding polynomial.  The restatement's layout and the product's (aleo_amd.varuna) are the same."""
    from aleo_amd import varuna
    for n_h, k in ((4, 1), (64, 3), (512, 8), (1 << 15, 32)):
        lay = V.randomness_layout(n_h, k)
        assert lay == varuna.randomness_layout(n_h, k)
        used = []
        for i in range(k): used += list(range(lay['rho'][i], lay['rho'][i] + 3)) + list(range(lay['blind'][i], lay['blind'][i] + 3 * V.HIDING_COEFFS))
        used += list(range(lay['mask'], lay['mask'] + 3 * n_h)) + list(range(lay['blind_mask'], lay['blind_mask'] + V.HIDING_COEFFS))
        assert len(used) == len(set(used)) == lay['total'] and max(used) == lay['total'] - 1
    # the stream itself: no short period, no collision between seeds that differ in one bit (ChaCha20 under a 32-byte key)
    a = [V.random_fr(b'\x00' * 32, i) for i in range(300)]; b = [V.random_fr(b'\x01' + b'\x00' * 31, i) for i in range(300)]
    assert len(set(a)) == 300 and len(set(b)) == 300 and not set(a) & set(b)


@pytest.mark.gpu
def test_distinct_seeds_blind_every_element_differently():
    """Two proofs of the same statement under different 32-byte seeds share nothing that is blinded: every commitment, every evaluation, both
    openings and random_v differ (the sums differ too: the challenges do); the same seed reproduces the proof; seed = None draws fresh entropy."""
    from aleo_amd import varuna
    csr, z, c = _circuit(90, 3, 61)
    D = _max_degree(c); ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    try:
        zz = np.stack([synth.int_to_limbs(v, 4) for v in z])
        with varuna.NativeCircuitIndex(csr, 90, 3, len(z) - 3, ck) as nx:
            s1, s2 = bytes(range(32)), bytes(range(1, 33))
            p1, p2 = V.parse_proof(nx.prove([zz, zz], s1)), V.parse_proof(nx.prove([zz, zz], s2))
            assert nx.prove([zz, zz], s1) == V.proof_bytes(p1)
            f1, f2 = nx.prove(zz, None), nx.prove(zz, None)                    # os.urandom seeds
            assert f1 != f2 and V.parse_proof(f1)['witness'] != V.parse_proof(f2)['witness']
        flat = lambda p: [x for t in p['witness'] for x in t] + [p['commitments'][n] for n in ('mask', 'g_1', 'h_1', 'h_2')] + p['commitments']['g_abc'] + \
            p['evaluations'] + p['sums'] + [p['openings'][0][0], p['openings'][0][1], p['openings'][1][0]]
        a, b = flat(p1), flat(p2)
        assert len(a) == len(b) and all(x != y for x, y in zip(a, b))
        w = p1['witness']; assert len({x for t in w for x in t}) == 6               # the two instances carry the same assignment, yet no two of their commitments coincide
    finally:
        ck.close()


@pytest.mark.gpu
def test_proof_of_a_real_poseidon_gadget_circuit():
    """Not a synthetic shape: the R1CS of a chain of hash_psd2 gadgets (snarkVM's Poseidon, rate 2, with the library's own round constants and matrix —
    synth.poseidon_chain_r1cs: x^17 s-boxes as five constraints each, linear layers as growing linear combinations), its public input the final hash.
    The witness satisfies it, the public root equals hash_psd2 iterated by the restatement AND by the product's host Poseidon, the device proof equals
    the restatement's byte for byte and verifies (pairing products) for that root and for no other."""
    import ctypes
    from aleo_amd import varuna
    from oracle import poseidon as ps
    k = 3
    csr, z, root = synth.poseidon_chain_r1cs(k, 4242)
    n = len(csr['a'][0]) - 1
    assert n == 276 * k
    coef = [synth.limbs_to_int(x) for x in synth.uniform_scalars(k + 1, 4242)]
    h = coef[0]; hp = coef[0]; L = aleo_amd.lib()
    for j in range(k):
        h = ps.hash_psd2([h, coef[j + 1]])
        a = np.stack([synth.int_to_limbs(hp, 4), synth.int_to_limbs(coef[j + 1], 4)]); o = np.zeros((1, 4), dtype=np.uint64)
        aleo_amd._lib.check(L.aleo_mi355x_poseidon_hash_fr(2, a.ctypes.data_as(ctypes.c_void_p), 2, o.ctypes.data_as(ctypes.c_void_p), 1), 'poseidon_hash_fr')
        hp = synth.limbs_to_int(o[0])
    assert h == hp == root == z[1]
    rows = lambda m: [[(int(csr[m][1][q]), synth.limbs_to_int(csr[m][2][q])) for q in range(csr[m][0][i], csr[m][0][i + 1])] for i in range(n)]
    c = V.Circuit(n, 2, len(z) - 2, rows('a'), rows('b'), rows('c'))
    for i in range(n):                                                       # the witness satisfies every constraint
        va, vb, vc = (sum(val * z[v] for v, val in c.m[m][i]) % V.R for m in 'abc')
        assert va * vb % V.R == vc
    D = _max_degree(c); setup = V.Setup(TAU, S_GAMMA, D); idx = V.Index(c, setup)
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    try:
        with varuna.NativeCircuitIndex(csr, n, 2, len(z) - 2, ck) as nx:
            assert nx.vk_bytes == idx.vk_bytes()
            data = nx.prove(np.stack([synth.int_to_limbs(v, 4) for v in z]), 99)
    finally:
        ck.close()
    assert data == V.prove(idx, setup, z, _rand(c, 99))[1]
    vk = setup.verifier_key(c)
    assert V.verify_pairing(idx, vk, [1, root], data) and not V.verify_pairing(idx, vk, [1, (root + 1) % V.R], data)


@pytest.mark.gpu
def test_independent_proofs_in_lockstep_equal_the_single_calls():
    """aleo_mi355x_varuna_prove_many: independent proofs — different circuits, instance counts, several circuits per proof, different seeds — proved in
    lockstep (every round's commitments of all of them in one launch chain) come out byte for byte as the single-proof entry points give them and as the
    restatement gives them; a request with an assignment that violates its circuit drops out with ALEO_MI355X_ERR_UNSATISFIED and one with an unknown
    index handle with BAD_HANDLE while the others complete; the same once more in another order."""
    from aleo_amd import varuna
    cs, csrs, zs, D = _batch_case([(60, 3, 71, 2), (25, 2, 72, 1), (130, 4, 73, 1)])
    setup = V.Setup(TAU, S_GAMMA, D); idx = [V.Index(c, setup) for c in cs]
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    lim = lambda a: np.stack([synth.int_to_limbs(v, 4) for v in a])
    try:
        nx = [varuna.NativeCircuitIndex(csrs[j], cs[j].n_constraints, cs[j].n_public, len(zs[j][0]) - cs[j].n_public, ck) for j in range(3)]
        za = [[lim(z) for z in zz] for zz in zs]
        reqs = [([nx[0]], [za[0]], 11), ([nx[1]], [za[1]], 12), ([nx[2], nx[0]], [za[2], za[0][:1]], 13), ([nx[1]], [za[1]], 14), ([nx[0], nx[1], nx[2]], [za[0], za[1], za[2]], 15)]
        single = [varuna.prove_batch_native(ix, a, sd) for ix, a, sd in reqs]
        assert single[1] != single[3]                                               # same statement, different seeds
        got = varuna.prove_many_native(reqs)
        assert got == single
        want2 = V.prove_batch([(idx[2], zs[2]), (idx[0], zs[0][:1])], setup, V.random_stream(13, max(cs[2].n_h, cs[0].n_h), 2))[1]
        assert got[2] == want2 and V.verify([idx[2], idx[0]], setup, [[z[:cs[2].n_public] for z in zs[2]], [zs[0][0][:cs[0].n_public]]], got[2])
        bad = za[1][0].copy(); bad[cs[1].n_public + 3] = synth.int_to_limbs(12345, 4)     # breaks a constraint of circuit 1
        class Ghost: handle = 987654321; n_vars = nx[1].n_vars
        mixed = [reqs[0], ([nx[1]], [[bad]], 21), reqs[2], ([Ghost()], [za[1]], 22), reqs[4]]
        out = varuna.prove_many_native(mixed)
        assert out[0] == single[0] and out[2] == single[2] and out[4] == single[4] and out[1] == 6 and out[3] == 4
        assert varuna.prove_many_native(list(reversed(reqs))) == list(reversed(single))
        assert varuna.prove_many_native(reqs[:1]) == single[:1]
        for x in nx: x.close()
    finally:
        ck.close()


@pytest.mark.gpu
def test_helper_contexts_under_contention():
    """The borrowed helper contexts (lockstep workers, the second context of a multi-chain commitment) when several calls want them at once: three
    threads run lockstep calls of different sizes while a fourth commits batches big enough to split into several launch chains (two sets per chain
    on the 2^19-bucket table) — every result equals the one computed alone."""
    import threading, torch
    from aleo_amd import varuna, msm as M, kzg
    cs, csrs, zs, D = _batch_case([(60, 3, 81, 2), (130, 4, 82, 1)])
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    lim = lambda a: np.stack([synth.int_to_limbs(v, 4) for v in a])
    n = 1 << 20
    pb = M.PinnedBases.generate_multiples(synth.generator_affine104(), 1, n).precompute()
    try:
        nx = [varuna.NativeCircuitIndex(csrs[j], cs[j].n_constraints, cs[j].n_public, len(zs[j][0]) - cs[j].n_public, ck) for j in range(2)]
        za = [[lim(z) for z in zz] for zz in zs]
        reqs = [[([nx[q % 2]], [za[q % 2]], 100 * t + q) for q in range(3 + 2 * t)] for t in range(3)]
        want = [varuna.prove_many_native(r) for r in reqs]
        polys = torch.from_numpy(synth.uniform_scalars(5 * n, 4242).view(np.int64)).cuda(); torch.cuda.synchronize()
        ptrs = [polys.data_ptr() + j * n * 32 for j in range(5)]
        commit = lambda: M.VariableBase.msm_batch_device(pb, ptrs, [n] * 5)          # five results, two per chain: three chains
        want_c = commit()
        got = [None] * 3; got_c = []; errs = []
        def prover(t):
            try:
                for _ in range(3): got[t] = varuna.prove_many_native(reqs[t])
            except Exception as e: errs.append(repr(e))
        def committer():
            try:
                for _ in range(4): got_c.append(commit())
            except Exception as e: errs.append(repr(e))
        th = [threading.Thread(target=prover, args=(t,)) for t in range(3)] + [threading.Thread(target=committer)]
        for x in th: x.start()
        for x in th: x.join()
        assert not errs, errs
        assert got == want
        assert all(np.array_equal(np.asarray(g), np.asarray(want_c)) for g in got_c)
        for x in nx: x.close()
    finally:
        pb.close(); ck.close()


@pytest.mark.gpu
def test_a_transaction_of_many_transitions_in_one_proof():
    """One proof covers a whole transaction: up to 32 transitions (snarkVM's limit per transaction) in ANY split over circuits — twelve instances of one
    circuit, and eleven circuits with one or two instances each — byte for byte the restatement's proofs and accepted by its verifier; 33 instances are
    refused."""
    from aleo_amd import varuna
    lim = lambda a: np.stack([synth.int_to_limbs(v, 4) for v in a])
    for shapes, seed in (([(40, 2, 91, 12)], 31), ([(12 + 3 * j, 1 + j % 3, 100 + j, 1 + (j % 4 == 0)) for j in range(11)], 32)):
        cs, csrs, zs, D = _batch_case(shapes)
        setup = V.Setup(TAU, S_GAMMA, D); idx = [V.Index(c, setup) for c in cs]
        ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
        try:
            nx = [varuna.NativeCircuitIndex(csrs[j], cs[j].n_constraints, cs[j].n_public, len(zs[j][0]) - cs[j].n_public, ck) for j in range(len(cs))]
            got = varuna.prove_batch_native(nx, [[lim(z) for z in zz] for zz in zs], seed)
            want = V.prove_batch(list(zip(idx, zs)), setup, V.random_stream(seed, max(c.n_h for c in cs), sum(len(z) for z in zs)))[1]
            assert got == want, shapes
            assert V.verify(idx, setup, [[z[:c.n_public] for z in zz] for c, zz in zip(cs, zs)], got)
            if len(cs) == 1:
                with pytest.raises(Exception): varuna.prove_batch_native(nx, [[lim(zs[0][0])] * 33], seed)
            for x in nx: x.close()
        finally:
            ck.close()


@pytest.mark.gpu
@pytest.mark.parametrize('k', [28, 29, 32])
def test_one_circuit_with_up_to_32_instances_native(k):
    """29..32 instances of ONE circuit: the second round's sum over c_i * numerator_i has more terms than one fr_lincomb launch takes (28) — the native
    prover chains launches (lincomb_any) as the step-by-step host side always did.  Both host sides against the restatement's bytes, and its verifier."""
    from aleo_amd import varuna
    lim = lambda a: np.stack([synth.int_to_limbs(v, 4) for v in a])
    cs, csrs, zs, D = _batch_case([(20, 2, 211, k)], seed=5 + k)
    setup = V.Setup(TAU, S_GAMMA, D); idx = V.Index(cs[0], setup)
    want = V.prove(idx, setup, zs[0], _rand(cs[0], 900 + k, k))[1]
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    try:
        with varuna.NativeCircuitIndex(csrs[0], 20, 2, len(zs[0][0]) - 2, ck) as nx:
            got = nx.prove([lim(z) for z in zs[0]], 900 + k)
            assert got == want
            assert V.verify(idx, setup, [z[:2] for z in zs[0]], got)
            many = varuna.prove_many_native([([nx], [[lim(z) for z in zs[0]]], 900 + k)])      # the lockstep entry point takes the same path
            assert many[0] == want
        ix = varuna.CircuitIndex(csrs[0], 20, 2, len(zs[0][0]) - 2, ck)
        assert varuna.prove_native(ix, [lim(z) for z in zs[0]], 900 + k) == want
    finally:
        ck.close()


def _shard_the_key(ck, devices):
    """A sharded copy (with window tables per shard) of a committer key's points, attached so that EVERY commitment and EVERY transform of the prover goes through it."""
    host = ck.bases.download()
    sb = aleo_amd.ShardedBases(host, devices=devices, precompute=True)
    ck.bases.attach_shards(sb, 0, transforms_from=1)          # and every transform over the shards' devices (power-of-two device lists)
    return sb


@pytest.mark.gpu
@pytest.mark.parametrize('G', [2, 4, 8])
def test_proofs_against_a_sharded_committer_key_equal_the_single_device_proofs(G):
    """Row e2 (a proof that spans devices): with a sharded copy of the committer key attached, index commitments and every round's commitments are cut
    at the shard boundaries, each shard's device (here the one card listed G times) pulls its slices of the coefficient vectors and runs its own
    Pippenger, and the 144-byte partials are added on the host — the frozen fixtures (single proofs and proofs over several circuits), a lockstep call
    and the verifying keys must come out byte for byte as from the single-device key."""
    from aleo_amd import varuna
    tau, sg, cases = _golden_cases()
    for case in cases:
        csr, zs, c = _golden_instance(case)
        zq = [np.stack([synth.int_to_limbs(v, 4) for v in q]) for q in zs]
        ck = varuna.synthetic_committer_key(tau, sg, case['max_degree'])
        sb = _shard_the_key(ck, [0] * G)
        try:
            with varuna.NativeCircuitIndex(csr, case['n_constraints'], case['n_public'], len(zs[0]) - case['n_public'], ck, domains=case['domains']) as nx:
                assert nx.vk_bytes.hex() == case['vk'] and nx.prove(zq, case['proof_seed']).hex() == case['proof']
                assert varuna.prove_many_native([([nx], [zq], case['proof_seed'])] * 3) == [bytes.fromhex(case['proof'])] * 3
        finally:
            ck.bases.attach_shards(None); sb.close(); ck.close()
    cases, batches = _golden_batches()
    for b in batches:
        ck = varuna.synthetic_committer_key(TAU, S_GAMMA, b['max_degree']); sb = _shard_the_key(ck, [0] * G); nx, za = [], []
        try:
            for j in b['members']:
                case = cases[j]; csr, zs, c = _golden_instance(case)
                nx.append(varuna.NativeCircuitIndex(csr, case['n_constraints'], case['n_public'], len(zs[0]) - case['n_public'], ck, domains=case['domains']))
                za.append([np.stack([synth.int_to_limbs(v, 4) for v in q]) for q in zs])
            assert varuna.prove_batch_native(nx, za, b['proof_seed']).hex() == b['proof']
        finally:
            for x in nx: x.close()
            ck.bases.attach_shards(None); sb.close(); ck.close()


@pytest.mark.gpu
def test_lockstep_with_every_context_taken_and_a_sharded_key_makes_progress():
    """Round-4 advisor finding: a shard thread of commit_sharded that found no free context on the prover's device blocked on ONE helper context, which a worker
    of the same lockstep call — parked at the round barrier, waiting for that very commitment — could hold: a deadlock with 8 lockstep workers, 8 shards on one
    device and a second caller.  Now such a shard borrows the caller's own context (idle while it waits).  Eight workers, the key as 8 shards of the one card,
    two caller threads at once: both calls must return the frozen proofs within the time limit."""
    import threading
    from aleo_amd import varuna
    tau, sg, cases = _golden_cases()
    case = cases[0]; csr, zs, c = _golden_instance(case)
    zq = [np.stack([synth.int_to_limbs(v, 4) for v in q]) for q in zs]
    ck = varuna.synthetic_committer_key(tau, sg, case['max_degree']); sb = _shard_the_key(ck, [0] * 8)
    keep = os.environ.get('ALEO_MI355X_LOCKSTEP_WORKERS'); os.environ['ALEO_MI355X_LOCKSTEP_WORKERS'] = '8'
    try:
        with varuna.NativeCircuitIndex(csr, case['n_constraints'], case['n_public'], len(zs[0]) - case['n_public'], ck, domains=case['domains']) as nx:
            got = [None, None]
            def call(i): got[i] = varuna.prove_many_native([([nx], [zq], case['proof_seed'])] * 8)
            th = [threading.Thread(target=call, args=(i,), daemon=True) for i in range(2)]
            for t in th: t.start()
            for t in th: t.join(timeout=240)
            assert not any(t.is_alive() for t in th), 'a lockstep call against a sharded key hangs when every context of the device is taken'
            assert got[0] == [bytes.fromhex(case['proof'])] * 8 and got[1] == got[0]
    finally:
        if keep is None: os.environ.pop('ALEO_MI355X_LOCKSTEP_WORKERS', None)
        else: os.environ['ALEO_MI355X_LOCKSTEP_WORKERS'] = keep
        ck.bases.attach_shards(None); sb.close(); ck.close()


@pytest.mark.gpu
@pytest.mark.parametrize('G', [2, 4, 8])
def test_a_2_18_constraint_proof_against_a_sharded_key(G):
    """The size the sharding is for: 2^18 constraints (|K_A| = 2^20: commitments of up to 2^20 points), two instances, the key's 2^21 powers as G shards
    with their own window tables — and only commitments AND TRANSFORMS of >= 2^16 elements routed to them (min_points): the small ones stay on the
    prover's device, every transform on H (2^18), 4|H| (2^20), K and 2|K| (2^20, 2^21) runs through aleo_mi355x_ntt_fr_sharded_device's path (slabs pulled and
    pushed by peer copies; here the one card listed G times).  Byte-equal to the single-device proof and verifying key; and the segment entry point itself
    against the single-device commitment (degree bound + hiding segments that straddle shard boundaries)."""
    import torch
    from aleo_amd import varuna
    from aleo_amd.kzg import SonicKZG10
    n = (1 << 18) - 64
    csr, z = synth.synthetic_r1cs(n, 4, 618, long_rows=4)
    zz = np.stack([synth.int_to_limbs(v, 4) for v in z])
    D = (1 << 21) - 1
    ck = varuna.synthetic_committer_key(TAU, S_GAMMA, D)
    try:
        with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx:
            vk0 = nx.vk_bytes; want = nx.prove([zz, zz], 4242)
        sb = aleo_amd.ShardedBases(ck.bases.download(), devices=[0] * G, precompute=True)
        try:
            # the entry point alone: three results, segments at offsets that cross shard boundaries (shard size 2^21 / G)
            m = 1 << 20
            d = torch.from_numpy(util_uniform(3 * m + 8, 77).view(np.int64)).cuda(); torch.cuda.synchronize()
            p0 = d.data_ptr()
            segs = [(p0, m, 0, 0), (p0 + 32 * m, m - 5, D - (m - 6), 1), (p0 + 64 * m, (1 << 19) + 77, (1 << 19) - 33, 2), (p0 + 96 * m, 3, ck.gamma_offset, 2), (p0 + 96 * m + 96, 1, 0, 1)]
            a = SonicKZG10.commit_segments_device(ck, segs, 3)
            b = SonicKZG10.commit_segments_sharded_device(sb, segs, 3)
            assert (a == b).all()
            ck.bases.attach_shards(sb, 1 << 16, transforms_from=1 << 16)
            with varuna.NativeCircuitIndex(csr, n, 4, len(z) - 4, ck) as nx:
                assert nx.vk_bytes == vk0
                assert nx.prove([zz, zz], 4242) == want
        finally:
            ck.bases.attach_shards(None); sb.close()
    finally:
        ck.close()


def util_uniform(n, seed):
    return synth.uniform_scalars(n, seed)

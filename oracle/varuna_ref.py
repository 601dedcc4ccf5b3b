"""TEST INFRASTRUCTURE — CPU restatement (plain Python integers) of the AHP prover for R1CS that the MI355X path runs above its
operators, and a verifier for its proofs.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module.

What it restates.  snarkVM 0.14.5 `snarkvm_algorithms::snark::varuna` (`Varuna::prove_batch`, `AHPForR1CS::prover_{first..fourth}_round`)
is a third-party dependency of the reference (Cargo.lock:2200) and absent from /root/reference; its published algorithm is Marlin
(Chiesa, Hu, Maller, Mishra, Vesely, Ward — EUROCRYPT 2020, §5 "AHP for R1CS") with the arrangement snarkVM uses [UPSTREAM-RECALL]:
  * z = (public ‖ private) laid out on the constraint domain H with the public inputs on the subgroup X ⊂ H; w = (ẑ − x̂) / v_X;
  * only ẑ_a, ẑ_b are committed (ẑ_c = ẑ_a ẑ_b inside the first sumcheck); a mask polynomial of degree 3|H| + 2b − 3 (b = 1);
  * matrices arithmetised as row / col / val / row_col over K with val = M[r,c] / u_H(col, col);
  * commitments per proof w, z_a, z_b, mask | g_1, h_1 | g_a, g_b, g_c (+ three sums) | h_2, evaluations z_b(β), g_1(β), g_a(γ), g_b(γ), g_c(γ),
    two batched KZG openings (at β with the hiding `random_v`, at γ without) — the 901-byte layout of the `proof1…` string held at
    /root/reference/wasm/src/programs/transaction.rs:100 (decoded in tests/golden/reference_proof.json);
  * SonicKZG10 commitments: hiding polynomials of degree 2 against the γ-powers, degree bounds through shifted powers.
The Fiat-Shamir transcript is upstream's construction: the Poseidon sponge over Fq (rate 2, capacity 1; oracle/poseidon.py FiatShamir) behind
the AlgebraicSponge interface — protocol name and batch sizes as bytes, public inputs / sums / evaluations as non-native limbs, commitments as their
affine coordinates, 252-bit challenges for the rounds and one 168-bit challenge per polynomial of an opening — in the order recalled from
varuna.rs / ahp/verifier [UPSTREAM-RECALL]: the Poseidon primitive itself is pinned by reference-held data over Fr (tests/test_poseidon.py), the
absorb order and the limb packing are not (no transcript value exists in /root/reference).  The prover's randomness is a ChaCha20 stream under a
32-byte seed (upstream: `Fr::rand` on the caller's CSPRNG, /root/reference/rust/src/program/execute.rs:74 — its draw order is not reproduced: the
stream layout is this module's).  The setup is synthetic (known trapdoor), upstream's is the universal SRS.  Two verifiers:
`verify_pairing` checks the openings as pairing products over public G2 elements (oracle/pairing.py); `verify` is the same check with the pairing
replaced by the equivalent G1 equation under the synthetic setup's trapdoor (fast; used where many proofs are verified).  **Parity unpinned**: the reference holds no proof this restatement could be compared with value
for value; what pins it is (i) the verifier below accepting its proofs and rejecting tampered ones, and (ii) the byte layout above."""
from __future__ import annotations
import struct
from . import pyref as P
from . import poseidon as PS

R = P.FR_MODULUS
PROTOCOL_NAME = b'VARUNA-2023'
HIDING_COEFFS = 3            # hiding bound 1 -> random polynomial of degree 2 [UPSTREAM-RECALL: kzg10 calculate_hiding_polynomial_degree]


def inv(a): return pow(a % R, -1, R)


class Domain:
    def __init__(self, size):
        assert size & (size - 1) == 0
        self.size = size; self.lg = size.bit_length() - 1
        self.gen = pow(P.FR_TWO_ADIC_ROOT, 1 << (P.FR_TWO_ADICITY - self.lg), R)
        self.size_inv = inv(size)

    def fft(self, coeffs):
        x = list(coeffs) + [0] * (self.size - len(coeffs)); assert len(x) == self.size
        return P.fft_fast(x, self.gen)

    def ifft(self, evals):
        assert len(evals) == self.size
        return [v * self.size_inv % R for v in P.fft_fast(list(evals), inv(self.gen))]

    def vanishing(self, x): return (pow(x, self.size, R) - 1) % R
    def elements(self):
        out, a = [], 1
        for _ in range(self.size): out.append(a); a = a * self.gen % R
        return out


def poly_eval(c, x):
    acc = 0
    for v in reversed(c): acc = (acc * x + v) % R
    return acc


def h_position(var, n_public, n_x, n_h):
    """Index on H of variable `var` (public variables first): public i -> i * |H|/|X|; the j-th private variable -> the j-th
    element of H \\ X [UPSTREAM-RECALL: EvaluationDomain::reindex_by_subdomain]."""
    ratio = n_h // n_x
    if var < n_public: return var * ratio
    j = var - n_public
    return j + j // (ratio - 1) + 1


def fs_start(vk_points, ks, x_evals):
    """Varuna::init_sponge [UPSTREAM-RECALL]: the protocol name; per circuit its batch size (u64 LE bytes) and the padded public inputs of each of its
    instances (non-native); then every circuit's index commitments.  vk_points: per circuit the twelve points in vk order; x_evals: per instance."""
    fs = PS.FiatShamir()
    fs.absorb_bytes(PROTOCOL_NAME)
    at = 0
    for kj in ks:
        fs.absorb_bytes(int(kj).to_bytes(8, 'little'))
        for xe in x_evals[at:at + kj]: fs.absorb_nonnative(xe)
        at += kj
    for pts in vk_points: fs.absorb_points(pts)
    return fs


def vk_points_of(vk_bytes: bytes):
    return [P.g1_decompress(vk_bytes[48 * i:48 * i + 48]) for i in range(12)]


def fr_bytes(v): return int(v % R).to_bytes(32, 'little')


class Circuit:
    """R1CS in CSR-like python form: rows of (variable, value) for A, B, C; n_public counts the leading 1."""
    def __init__(self, n_constraints, n_public, n_private, a, b, c, domains='auto'):
        self.n_constraints, self.n_public, self.n_private = n_constraints, n_public, n_private
        self.m = {'a': a, 'b': b, 'c': c}
        n_x = 1
        while n_x < n_public: n_x *= 2
        n_h = 1
        while n_h < max(n_constraints, n_x + n_private, 2 * n_x): n_h *= 2
        self.n_k_m = {}
        for name, rows in (('a', a), ('b', b), ('c', c)):                        # one non-zero domain per matrix [UPSTREAM-RECALL: non_zero_{a,b,c}_domain]
            nnz = sum(len(r) for r in rows); n_k = 2
            while n_k < nnz: n_k *= 2
            self.n_k_m[name] = n_k
        # 'per_matrix': as above; 'shared': all three use the largest; 'auto' (the provers' default): shared below 2^18, where the rounds are
        # latency-bound and one batched transform beats three short ones, per matrix from there on (fewer points to commit)
        big = max(self.n_k_m.values())
        if domains == 'shared' or (domains == 'auto' and big < (1 << 18)): self.n_k_m = {m: big for m in 'abc'}
        else: assert domains in ('auto', 'per_matrix')
        self.n_x, self.n_h, self.n_k = n_x, n_h, big


class Setup:
    """Synthetic universal setup with its trapdoor: powers τ^i·G for i <= max_degree, hiding powers s·τ^i·G."""
    def __init__(self, tau, s_gamma, max_degree):
        self.tau, self.s_gamma, self.max_degree = tau % R, s_gamma % R, max_degree

    def verifier_key(self, circuit):
        """What a verifier holds (public; derived from the trapdoor here as a ceremony would): γG, H, τH and the negative powers of τ in G2 that
        un-shift degree-bounded commitments [UPSTREAM-RECALL: sonic_pc VerifierKey — h, beta_h, prepared_neg_powers_of_beta_h, gamma_g].
        circuit: one Circuit or the list a batch proof covers (bounds: the largest |H| − 2, every |K_M| − 2)."""
        circuits = [circuit] if hasattr(circuit, 'n_h') else list(circuit)
        H = P.G2_GENERATOR; ti = inv(self.tau)
        neg = lambda bound: P.g2_mul(H, pow(ti, self.max_degree - bound, R))
        return {'gamma_g': P.g1_mul(P.G1_GENERATOR, self.s_gamma), 'h': H, 'tau_h': P.g2_mul(H, self.tau),
                'neg_h': neg(max(c.n_h for c in circuits) - 2),
                'neg_k_by_size': {n: neg(n - 2) for n in sorted({c.n_k_m[M] for c in circuits for M in 'abc'})}}


def _commit_scalar(setup, coeffs, bound=None, blind=None):
    """Discrete log (base G) of SonicKZG10::commit: τ^(D − bound)·p(τ) + s·blind(τ)."""
    v = poly_eval(coeffs, setup.tau)
    if bound is not None:
        assert len(coeffs) <= bound + 1 <= setup.max_degree + 1
        v = v * pow(setup.tau, setup.max_degree - bound, R) % R
    if blind is not None: v = (v + setup.s_gamma * poly_eval(blind, setup.tau)) % R
    return v


def _point_bytes(scalar):
    return P.g1_compress(P.g1_mul(P.G1_GENERATOR, scalar % R))


class Index:
    """The index ("proving key" material) of one circuit: the arithmetisation of A, B, C over K, and the index commitments."""
    def __init__(self, circuit: Circuit, setup: Setup):
        c = self.circuit = circuit
        self.H, self.K, self.X = Domain(c.n_h), Domain(c.n_k), Domain(c.n_x)      # K: the largest of the three non-zero domains
        self.K_m = {m: Domain(c.n_k_m[m]) for m in 'abc'}
        he = self.H.elements(); n_h_inv = self.H.size_inv
        self.entries, self.evals, self.polys = {}, {}, {}
        for name, rows in c.m.items():
            ent = [(r, h_position(v, c.n_public, c.n_x, c.n_h), val % R) for r, row in enumerate(rows) for v, val in row]
            row = [he[r] for r, _, _ in ent]; col = [he[cp] for _, cp, _ in ent]
            val = [v * he[cp] % R * n_h_inv % R for _, cp, v in ent]          # M[r,c] / u_H(col, col), u_H(x, x) = |H| / x on H
            pad = c.n_k_m[name] - len(ent)
            row += [1] * pad; col += [1] * pad; val += [0] * pad
            rc = [a * b % R for a, b in zip(row, col)]
            self.entries[name] = ent
            self.evals[name] = {'row': row, 'col': col, 'val': val, 'row_col': rc}
            self.polys[name] = {k: self.K_m[name].ifft(v) for k, v in self.evals[name].items()}
        self.commit_scalars = {(m, k): _commit_scalar(setup, self.polys[m][k]) for m in 'abc' for k in ('row', 'col', 'val', 'row_col')}
        self._points = None

    def commit_points(self):
        """The twelve index commitments as points (the circuit's verifying key)."""
        if self._points is None: self._points = {mk: P.g1_mul(P.G1_GENERATOR, v) for mk, v in self.commit_scalars.items()}
        return self._points

    def vk_bytes(self):
        out = b''
        for m in 'abc':
            for k in ('row', 'col', 'val', 'row_col'): out += _point_bytes(self.commit_scalars[(m, k)])
        c = self.circuit
        return out + b''.join(int(v).to_bytes(8, 'little') for v in (c.n_h, c.n_k_m['a'], c.n_k_m['b'], c.n_k_m['c'], c.n_x))


class VerifyingKey:
    """What a verifier holds of one circuit — nothing of the prover's index: the twelve index commitments and the five domain sizes
    (Index.vk_bytes(), 616 bytes) and the number of public inputs.  Accepted wherever verify / verify_pairing take an Index, so a proof for a
    circuit far too large for this module's prover (2^20 constraints) is still checked here in seconds."""
    def __init__(self, vk_bytes: bytes, n_public: int):
        assert len(vk_bytes) == 12 * 48 + 40
        n_h, ka, kb, kc, n_x = (int.from_bytes(vk_bytes[576 + 8 * i:584 + 8 * i], 'little') for i in range(5))
        assert 1 <= n_public <= n_x
        self._bytes = bytes(vk_bytes)
        self.circuit = type('CircuitShape', (), {'n_h': n_h, 'n_k_m': {'a': ka, 'b': kb, 'c': kc}, 'n_k': max(ka, kb, kc), 'n_x': n_x, 'n_public': n_public})()
        self.H, self.K, self.X = Domain(n_h), Domain(self.circuit.n_k), Domain(n_x)
        self.K_m = {m: Domain(v) for m, v in self.circuit.n_k_m.items()}
        self._points = None

    def commit_points(self):
        if self._points is None:
            keys = [(m, k) for m in 'abc' for k in ('row', 'col', 'val', 'row_col')]
            self._points = {mk: P.g1_decompress(self._bytes[48 * i:48 * i + 48]) for i, mk in enumerate(keys)}
        return self._points

    def vk_bytes(self): return self._bytes


_M32 = 0xFFFFFFFF


def chacha20_block(key: bytes, counter: int, nonce: int) -> bytes:
    """One 64-byte ChaCha20 block (D. J. Bernstein's original layout: words 12-13 = 64-bit block counter, words 14-15 = 64-bit nonce; with
    counter = 1 | nonce_word0 << 32 it is RFC 7539's block function, whose §2.3.2 vector tests/test_varuna.py checks)."""
    assert len(key) == 32
    st = list(struct.unpack('<4I', b'expand 32-byte k')) + list(struct.unpack('<8I', key)) + [counter & _M32, (counter >> 32) & _M32, nonce & _M32, (nonce >> 32) & _M32]
    x = st[:]
    def rot(v, n): return ((v << n) | (v >> (32 - n))) & _M32
    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & _M32; x[d] = rot(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & _M32; x[b] = rot(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & _M32; x[d] = rot(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & _M32; x[b] = rot(x[b] ^ x[c], 7)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return struct.pack('<16I', *[(a + b) & _M32 for a, b in zip(x, st)])


def seed_bytes(seed) -> bytes:
    """A proof seed is 32 bytes (the ChaCha20 key); tests pass small integers, read as 32 little-endian bytes."""
    if isinstance(seed, (bytes, bytearray)): assert len(seed) == 32; return bytes(seed)
    return int(seed).to_bytes(32, 'little')


def random_fr(seed, index: int) -> int:
    """Element `index` of the proof's random stream: ChaCha20 under key = seed, block counter = index, nonce = attempt 0, 1, …; each block holds
    two candidates (bytes 0-31, 32-63, little-endian, low 253 bits); the first candidate below r is the element (rejection sampling: uniform)."""
    key = seed_bytes(seed); attempt = 0
    while True:
        blk = chacha20_block(key, index, attempt)
        for h in (0, 1):
            v = int.from_bytes(blk[32 * h:32 * h + 32], 'little') & ((1 << 253) - 1)
            if v < R: return v
        attempt += 1


def randomness_layout(n_h, k=1):
    """Offsets into the prover's random stream (canonical Fr) for k instances: rho_w, rho_a, rho_b per instance, mask[3|H|], then the
    hiding polynomials of w_i, z_a,i, z_b,i per instance and of the mask."""
    o = {'rho': [3 * i for i in range(k)], 'mask': 3 * k}
    base = 3 * k + 3 * n_h
    o['blind'] = [base + 3 * HIDING_COEFFS * i for i in range(k)]                 # w, z_a, z_b of instance i: HIDING_COEFFS each
    o['blind_mask'] = base + 3 * HIDING_COEFFS * k
    o['total'] = o['blind_mask'] + HIDING_COEFFS
    return o


def random_stream(seed, n_h: int, k: int = 1) -> list:
    return [random_fr(seed, i) for i in range(randomness_layout(n_h, k)['total'])]


def _blinded(H, evals, rho):
    p = H.ifft(evals) + [0]
    p[0] = (p[0] - rho) % R; p[H.size] = rho % R
    return p


def _axpy(dst, k, src):
    for i, v in enumerate(src): dst[i] = (dst[i] + k * v) % R


def _challenges_after_round1(fs, ks):
    """verifier_first_round [UPSTREAM-RECALL]: per circuit k_j − 1 instance combiners and — for every circuit but the first — a circuit combiner,
    one squeeze per circuit; then alpha, eta_b, eta_c in one squeeze.  The combiner of an instance = circuit combiner * instance combiner (1 for
    the first circuit resp. the first instance of a circuit); returned per instance in proof order."""
    comb = []
    for j, kj in enumerate(ks):
        el = fs.squeeze_nonnative(kj - 1 + (1 if j else 0))
        cc = el[kj - 1] if j else 1
        comb += [cc] + [cc * e % R for e in el[:kj - 1]]
    alpha, eta_b, eta_c = fs.squeeze_nonnative(3)
    return alpha, {'a': 1, 'b': eta_b, 'c': eta_c}, comb


def _challenges_after_round3(fs, m):
    """delta_{j,M}, circuit after circuit: the first is 1, the other 3m − 1 come from one squeeze."""
    el = [1] + fs.squeeze_nonnative(3 * m - 1)
    return [dict(zip('abc', el[3 * j:3 * j + 3])) for j in range(m)]


def _matrix_major(per_circuit):
    """[a_0, b_0, c_0, a_1, …] -> [a_0, a_1, …, b_0, …, c_0, …]: upstream's Commitments / Evaluations hold g_a, g_b, g_c as one vector per matrix
    [UPSTREAM-RECALL]; with one circuit both orders coincide (the reference's proof string)."""
    return [per_circuit[3 * j + t] for t in range(3) for j in range(len(per_circuit) // 3)]


def _circuit_major(per_matrix):
    m = len(per_matrix) // 3
    return [per_matrix[t * m + j] for j in range(m) for t in range(3)]


def _serialized_evaluations(evals, k, m):
    return evals[:k + 1] + _matrix_major(evals[k + 1:])


def prove(index: Index, setup: Setup, assignments, rand, vk_bytes=None):
    """One proof for k instances of one circuit (Varuna::prove_batch with one circuit).  assignments: one list of ints per instance (or a
    single such list), public variables first (z[0] = 1).  rand: the random stream (randomness_layout(n_h, k)).
    Returns (proof dict, proof bytes in the reference's layout)."""
    if assignments and not isinstance(assignments[0], (list, tuple)): assignments = [assignments]
    return prove_batch([(index, assignments)], setup, rand, None if vk_bytes is None else [vk_bytes])


def prove_batch(items, setup: Setup, rand, vk_bytes=None):
    """One proof for several circuits, each with its own instances — `Varuna::prove_batch(keys_to_constraints: BTreeMap<&ProvingKey, &[Assignment]>)`.
    items: [(Index, [assignment, ...]), ...] in the order the proof lists them; rand: randomness_layout(max |H|, total instances).
    What the circuits share [UPSTREAM-RECALL: varuna's batching over circuits]: the transcript and every challenge; ONE mask, g_1, h_1 for the first
    sumcheck, taken over the largest constraint domain H* with the selector s_j = v_{H*} / v_{H_j} in front of circuit j's summand
    (sum over H* of s_j F_j = |H*|/|H_j| times the sum of F_j over H_j; s_j (h_j v_{H_j} + X g_j) = h_j v_{H*} + X (s_j g_j), and multiplying by
    s_j = sum_t X^(t |H_j|) tiles g_j's remainder block |H*|/|H_j| times); ONE h_2 over the largest non-zero domain K* with selectors
    v_{K*} / v_{K_{j,M}}; two openings for everything.  With one circuit this is prove() above, byte for byte."""
    m = len(items); ks = [len(a) for _, a in items]; k = sum(ks)
    N = max(ix.circuit.n_h for ix, _ in items); n_k = max(ix.circuit.n_k for ix, _ in items)
    lead = [ix.circuit.n_h for ix, _ in items].index(N)
    HN = Domain(N); KN = Domain(n_k)
    lay = randomness_layout(N, k); assert len(rand) >= lay['total']
    # ---- first round --------------------------------------------------------------------------------------------------------
    inst = []                                                                     # all instances, circuit after circuit
    for j, (index, assignments) in enumerate(items):
        c = index.circuit; H, X = index.H, index.X
        n_h, n_x = c.n_h, c.n_x; he = H.elements(); ratio = n_h // n_x
        for z_assignment in assignments:
            i = len(inst)
            zH = [0] * n_h
            for v, val in enumerate(z_assignment): zH[h_position(v, c.n_public, n_x, n_h)] = val % R
            z_m = {}
            for name in 'ab':
                out = [0] * n_h
                for r, row in enumerate(c.m[name]): out[r] = sum(val * z_assignment[v] for v, val in row) % R
                z_m[name] = out
            x_evals = [z_assignment[t] % R if t < c.n_public else 0 for t in range(n_x)]
            x_poly = X.ifft(x_evals)
            w_evals = [0] * n_h
            for p in range(n_h):
                if p % ratio == 0: continue
                w_evals[p] = (zH[p] - poly_eval(x_poly, he[p])) * inv(X.vanishing(he[p])) % R
            rho = rand[lay['rho'][i]:lay['rho'][i] + 3]
            bl = [[v % R for v in rand[lay['blind'][i] + HIDING_COEFFS * t:lay['blind'][i] + HIDING_COEFFS * (t + 1)]] for t in range(3)]
            d = {'circuit': j, 'x_evals': x_evals, 'x_poly': x_poly, 'w': _blinded(H, w_evals, rho[0]), 'za': _blinded(H, z_m['a'], rho[1]),
                 'zb': _blinded(H, z_m['b'], rho[2]), 'blind': {'w': bl[0], 'za': bl[1], 'zb': bl[2]}}
            d['cb'] = {p_: _point_bytes(_commit_scalar(setup, d[p_], blind=d['blind'][p_])) for p_ in ('w', 'za', 'zb')}
            inst.append(d)
    mask = [v % R for v in rand[lay['mask']:lay['mask'] + 3 * N]]
    mask[0] = (-(mask[N] + mask[2 * N])) % R                                      # sum over H* = |H*| (m_0 + m_|H*| + m_2|H*|) = 0
    blind_mask = [v % R for v in rand[lay['blind_mask']:lay['blind_mask'] + HIDING_COEFFS]]
    cb = {'mask': _point_bytes(_commit_scalar(setup, mask, blind=blind_mask))}
    vkp = [vk_points_of(v) for v in vk_bytes] if vk_bytes is not None else [[ix.commit_points()[(M, kd)] for M in 'abc' for kd in ('row', 'col', 'val', 'row_col')] for ix, _ in items]
    fs = fs_start(vkp, ks, [d['x_evals'] for d in inst])
    fs.absorb_points([P.g1_decompress(d['cb'][p_]) for d in inst for p_ in ('w', 'za', 'zb')] + [P.g1_decompress(cb['mask'])])
    alpha, eta, comb = _challenges_after_round1(fs, ks)
    eta_b, eta_c = eta['b'], eta['c']
    # ---- second round: first sumcheck --------------------------------------------------------------------------------------
    h1 = [0] * (2 * N); rem_all = [0] * N
    vh_alpha = []
    for j, (index, _) in enumerate(items):
        c = index.circuit; H, X = index.H, index.X; n_h, n_x = c.n_h, c.n_x; he = H.elements()
        va = H.vanishing(alpha); assert va != 0; vh_alpha.append(va)
        r_alpha = [va * inv(alpha - h) % R for h in he]                           # u_H(alpha, h) on H
        t_evals = [0] * n_h
        for name in 'abc':
            for r, cp, v in index.entries[name]: t_evals[cp] = (t_evals[cp] + eta[name] * r_alpha[r] % R * v) % R
        r_poly, t_poly = H.ifft(r_alpha), H.ifft(t_evals)
        D4 = Domain(4 * n_h)
        e_r, e_t = D4.fft(r_poly), D4.fft(t_poly)
        acc4 = [0] * (4 * n_h)
        for d, ci in zip(inst, comb):
            if d['circuit'] != j: continue
            w = d['w']
            z_poly = [0] * (n_h + 1 + n_x)                                        # ẑ = w v_X + x̂
            for i, v in enumerate(w): z_poly[i] = (z_poly[i] - v) % R; z_poly[i + n_x] = (z_poly[i + n_x] + v) % R
            for i, v in enumerate(d['x_poly']): z_poly[i] = (z_poly[i] + v) % R
            e_z, e_a, e_b = D4.fft(z_poly), D4.fft(d['za']), D4.fft(d['zb'])
            for i in range(4 * n_h):
                acc4[i] = (acc4[i] + ci * (e_r[i] * ((e_a[i] + eta_b * e_b[i] + eta_c * e_a[i] % R * e_b[i]) % R) - e_t[i] * e_z[i])) % R
        q1 = D4.ifft(acc4)
        if j == lead:
            for i, v in enumerate(mask): q1[i] = (q1[i] + v) % R
        q = [0] * (3 * n_h)                                                       # q1 = h (X^|H| − 1) + remainder
        for i in range(4 * n_h - 1, n_h - 1, -1):
            q[i - n_h] = (q1[i] + (q[i] if i < 3 * n_h else 0)) % R
        rem = [(q1[i] + q[i]) % R for i in range(n_h)]
        assert rem[0] == 0, 'first sumcheck: the sum over H is not zero (unsatisfied assignment?)'
        assert not any(q[2 * n_h:])
        _axpy(h1, 1, q[:2 * n_h])
        for i in range(N): rem_all[i] = (rem_all[i] + rem[i % n_h]) % R          # X (s_j g_j): the remainder block tiled over H*
    g1 = rem_all[1:]
    cb['g_1'], cb['h_1'] = _point_bytes(_commit_scalar(setup, g1, bound=N - 2)), _point_bytes(_commit_scalar(setup, h1))
    fs.absorb_points([P.g1_decompress(cb['g_1']), P.g1_decompress(cb['h_1'])])
    beta = fs.squeeze_nonnative(1)[0]
    # ---- third round: the rational sumchecks over the K_{j,M} --------------------------------------------------------------------
    f, sigma, g, gcb = [], [], [], []
    for j, (index, _) in enumerate(items):
        c = index.circuit
        vh_beta = index.H.vanishing(beta); assert vh_beta != 0
        fj, sj, gj, cj = {}, {}, {}, {}
        for name in 'abc':
            ev = index.evals[name]; nkm = c.n_k_m[name]
            fe = [vh_alpha[j] * vh_beta % R * ev['val'][t] % R * inv((alpha - ev['row'][t]) * (beta - ev['col'][t])) % R for t in range(nkm)]
            fj[name] = index.K_m[name].ifft(fe); sj[name] = fj[name][0] * nkm % R; gj[name] = fj[name][1:]
            cj[name] = _point_bytes(_commit_scalar(setup, gj[name], bound=nkm - 2))
        f.append(fj); sigma.append(sj); g.append(gj); gcb.append(cj)
    fs.absorb_points([P.g1_decompress(cj[M]) for cj in gcb for M in 'abc'])           # absorb_with_msg: the commitments, then the sums circuit by circuit
    for sj in sigma: fs.absorb_nonnative([sj['a'], sj['b'], sj['c']])
    delta = _challenges_after_round3(fs, m)
    # ---- fourth round ------------------------------------------------------------------------------------------------------------
    h2 = [0] * n_k                                                                # h_2 = sum_{j,M} delta_{j,M} (a − b f) / v_{K_{j,M}}, each quotient on its own domain
    for j, (index, _) in enumerate(items):
        c = index.circuit; vh_beta = index.H.vanishing(beta)
        for name in 'abc':
            nkm = c.n_k_m[name]; D2 = Domain(2 * nkm); pl = index.polys[name]
            e_row, e_col, e_val, e_rc, e_f = D2.fft(pl['row']), D2.fft(pl['col']), D2.fft(pl['val']), D2.fft(pl['row_col']), D2.fft(f[j][name])
            num = []
            for i in range(2 * nkm):
                a_ = vh_alpha[j] * vh_beta % R * e_val[i] % R
                b_ = (alpha * beta - beta * e_row[i] - alpha * e_col[i] + e_rc[i]) % R
                num.append((a_ - b_ * e_f[i]) % R)
            pc = D2.ifft(num)
            hm = pc[nkm:]                                                         # P = h_M (X^|K_M| − 1), deg P < 2|K_M|
            assert all((pc[i] + hm[i]) % R == 0 for i in range(nkm)), 'fourth round: not divisible by v_K'
            _axpy(h2, delta[j][name], hm)
    cb['h_2'] = _point_bytes(_commit_scalar(setup, h2))
    fs.absorb_points([P.g1_decompress(cb['h_2'])])
    gamma = fs.squeeze_nonnative(1)[0]
    # ---- evaluations and the two openings --------------------------------------------------------------------------------------
    zb_beta = [poly_eval(d['zb'], beta) for d in inst]
    g1_beta = poly_eval(g1, beta); g_gamma = [{M: poly_eval(gj[M], gamma) for M in 'abc'} for gj in g]
    evals = zb_beta + [g1_beta] + [gg[M] for gg in g_gamma for M in 'abc']
    fs.absorb_nonnative(_serialized_evaluations(evals, k, m))
    # one short challenge per polynomial of an opening [UPSTREAM-RECALL: sonic_pc combine_for_open], the point beta first:
    # beta: g_1, z_b of every instance, the lincheck combination;  gamma: g_{j,M} circuit by circuit, the matrix combination
    ch_b = [fs.squeeze_short_one() for _ in range(k + 2)]
    ch_g = [fs.squeeze_short_one() for _ in range(3 * m + 1)]
    circuits = [ix for ix, _ in items]
    lc1 = lincheck_coefficients(circuits, [d['circuit'] for d in inst], HN, alpha, beta, eta, comb, sigma, zb_beta, g1_beta, [poly_eval(d['x_poly'], beta) for d in inst])
    p_beta = [0] * (3 * N)
    _axpy(p_beta, lc1['mask'], mask); _axpy(p_beta, lc1['h_1'], h1)
    for d, kz, kw in zip(inst, lc1['z_a'], lc1['w']): _axpy(p_beta, kz, d['za']); _axpy(p_beta, kw, d['w'])
    p_beta[0] = (p_beta[0] + lc1['const']) % R
    assert poly_eval(p_beta, beta) == 0, 'lincheck linear combination does not vanish at beta'
    xl = ch_b[k + 1]                                                              # ch_0 g_1 + sum_i ch_(1+i) z_b,i + ch_(k+1) LC1
    p_beta = [v * xl % R for v in p_beta]
    _axpy(p_beta, ch_b[0], g1)
    for i, d in enumerate(inst): _axpy(p_beta, ch_b[1 + i], d['zb'])
    v_beta = (ch_b[0] * g1_beta + sum(ch_b[1 + i] * v for i, v in enumerate(zb_beta))) % R
    w_beta = divide_by_linear(p_beta, beta, v_beta)
    bl = [0] * HIDING_COEFFS
    _axpy(bl, xl * lc1['mask'] % R, blind_mask)
    for i, (d, kz, kw) in enumerate(zip(inst, lc1['z_a'], lc1['w'])):
        _axpy(bl, ch_b[1 + i], d['blind']['zb']); _axpy(bl, xl * kz % R, d['blind']['za']); _axpy(bl, xl * kw % R, d['blind']['w'])
    random_v = poly_eval(bl, beta)
    bl_w = divide_by_linear(bl, beta, random_v)
    open_beta = (poly_eval(w_beta, setup.tau) + setup.s_gamma * poly_eval(bl_w, setup.tau)) % R
    lc2 = matrix_coefficients(circuits, KN, alpha, beta, gamma, delta, sigma, g_gamma)
    p_gamma = [0] * n_k
    for (j, M, kind), coef in lc2['index'].items(): _axpy(p_gamma, coef, circuits[j].polys[M][kind])
    _axpy(p_gamma, lc2['h_2'], h2)
    p_gamma[0] = (p_gamma[0] + lc2['const']) % R
    assert poly_eval(p_gamma, gamma) == 0, 'matrix sumcheck linear combination does not vanish at gamma'
    p_gamma = [v * ch_g[3 * m] % R for v in p_gamma]                              # sum_{j,M} ch_(3j+M) g_{j,M} + ch_(3m) LC2
    v_gamma = 0
    for j in range(m):
        for t, M in enumerate('abc'):
            _axpy(p_gamma, ch_g[3 * j + t], g[j][M]); v_gamma = (v_gamma + ch_g[3 * j + t] * g_gamma[j][M]) % R
    open_gamma = poly_eval(divide_by_linear(p_gamma, gamma, v_gamma), setup.tau)
    proof = {'batch_sizes': ks, 'instances': k, 'witness': [(d['cb']['w'], d['cb']['za'], d['cb']['zb']) for d in inst],
             'commitments': {'mask': cb['mask'], 'g_1': cb['g_1'], 'h_1': cb['h_1'], 'h_2': cb['h_2'], 'g_abc': [cj[M] for cj in gcb for M in 'abc']},
             'evaluations': evals, 'sums': [sj[M] for sj in sigma for M in 'abc'],
             'openings': [(_point_bytes(open_beta), random_v), (_point_bytes(open_gamma), None)]}
    if m == 1: proof['commitments'].update({'g_a': gcb[0]['a'], 'g_b': gcb[0]['b'], 'g_c': gcb[0]['c']})
    return proof, proof_bytes(proof)


def divide_by_linear(p, z, value):
    """(p(X) − value) / (X − z), value = p(z)."""
    q = [0] * (len(p) - 1); s = 0
    for j in range(len(p) - 1, 0, -1):
        s = (p[j] + z * s) % R; q[j - 1] = s
    assert (p[0] + z * s) % R == value % R
    return q


def lincheck_coefficients(circuits, inst_circuit, HN, alpha, beta, eta, comb, sigma, zb_beta, g1_beta, x_beta):
    """Coefficients of the linear combination of (mask, z_a,i, w_i, h_1, 1) that must vanish at beta (first sumcheck as the verifier evaluates it):
    mask + sum_i c_i s_j(beta) [r_j(alpha, beta)(z_a,i + eta_b z_b,i(beta) + eta_c z_a,i z_b,i(beta)) − t_j(beta)(w_i v_{X_j}(beta) + x̂_i(beta))]
    − v_{H*}(beta) h_1 − beta g_1(beta), j = the circuit of instance i, s_j = v_{H*} / v_{H_j} (1 for the largest domain H*)."""
    per = []
    for ix, sg in zip(circuits, sigma):
        H = ix.H
        r_ab = (H.vanishing(alpha) - H.vanishing(beta)) * inv(alpha - beta) % R
        t_beta = (sg['a'] + eta['b'] * sg['b'] + eta['c'] * sg['c']) % R
        sel = 1 if H.size == HN.size else HN.vanishing(beta) * inv(H.vanishing(beta)) % R
        per.append((r_ab, t_beta, sel, ix.X.vanishing(beta)))
    const = (-beta * g1_beta) % R; ka, kw = [], []
    for j, ci, zb, xb in zip(inst_circuit, comb, zb_beta, x_beta):
        r_ab, t_beta, sel, vx = per[j]; cs = ci * sel % R
        const = (const + cs * (r_ab * eta['b'] % R * zb - t_beta * xb)) % R
        ka.append(cs * r_ab % R * (1 + eta['c'] * zb) % R); kw.append((-cs * t_beta % R * vx) % R)
    return {'mask': 1, 'z_a': ka, 'w': kw, 'h_1': (-HN.vanishing(beta)) % R, 'const': const}


def matrix_coefficients(circuits, KN, alpha, beta, gamma, delta, sigma, g_gamma):
    """Coefficients over (val, row, col, row_col of every circuit's A, B, C; h_2; 1) of the combination that must vanish at gamma (second sumcheck):
    sum_{j,M} delta_{j,M} s_{j,M}(gamma) (vv_j val − f_{j,M}(gamma) (alpha beta − beta row − alpha col + row_col)) − v_{K*}(gamma) h_2, with the selector
    s_{j,M} = v_{K*} / v_{K_{j,M}} (K* the largest non-zero domain), vv_j = v_{H_j}(alpha) v_{H_j}(beta) and f_{j,M}(gamma) = gamma g_{j,M}(gamma) + sigma_{j,M} / |K_{j,M}|."""
    idx = {}; const = 0
    for j, ix in enumerate(circuits):
        vv = ix.H.vanishing(alpha) * ix.H.vanishing(beta) % R
        for M in 'abc':
            Km = ix.K_m[M]
            fm = (gamma * g_gamma[j][M] + sigma[j][M] * Km.size_inv) % R          # f_M(gamma)
            d = delta[j][M] * KN.vanishing(gamma) % R * inv(Km.vanishing(gamma)) % R
            idx[(j, M, 'val')] = d * vv % R
            idx[(j, M, 'row')] = d * fm % R * beta % R
            idx[(j, M, 'col')] = d * fm % R * alpha % R
            idx[(j, M, 'row_col')] = (-d * fm) % R
            const = (const - d * fm % R * alpha % R * beta) % R
    return {'index': idx, 'h_2': (-KN.vanishing(gamma)) % R, 'const': const}


def proof_bytes(proof) -> bytes:
    """The reference's Proof::to_bytes_le layout (901 bytes for one circuit with one instance): version, the batch sizes, the
    witness commitments instance after instance, mask, g_1, h_1, every g_a, every g_b, every g_c (one vector per matrix over the circuits), h_2, the
    evaluations (z_b of every instance, g_1, then every g_a, g_b, g_c likewise), the sums circuit by circuit, the openings.  The dicts keep
    g_abc / evaluations circuit-major; only the bytes are matrix-major."""
    c = proof['commitments']; u64 = lambda v: int(v).to_bytes(8, 'little'); ks = proof['batch_sizes']
    out = b'\x00' + u64(len(ks)) + b''.join(u64(v) for v in ks) + b''.join(w + a + b for w, a, b in proof['witness'])
    k = sum(ks)
    out += b'\x01' + c['mask'] + c['g_1'] + c['h_1'] + b''.join(_matrix_major(c['g_abc'])) + c['h_2']
    out += b''.join(fr_bytes(v) for v in _serialized_evaluations(proof['evaluations'], k, len(ks))) + u64(len(ks)) + b''.join(fr_bytes(v) for v in proof['sums']) + u64(len(proof['openings']))
    for pt, rv in proof['openings']:
        out += pt + (b'\x01' + fr_bytes(rv) if rv is not None else b'\x00')
    return out + b'\x00'                                                        # BatchLCProof.evaluations: None


def parse_proof(data: bytes):
    assert data[0] == 0
    m = int.from_bytes(data[1:9], 'little'); assert 1 <= m <= 64
    ks = [int.from_bytes(data[9 + 8 * j:17 + 8 * j], 'little') for j in range(m)]; k = sum(ks); assert all(1 <= v <= 64 for v in ks)
    pos = 9 + 8 * m; c = {}
    def pt():
        nonlocal pos; v = data[pos:pos + 48]; pos += 48; assert len(v) == 48; return v
    def fr():
        nonlocal pos; v = int.from_bytes(data[pos:pos + 32], 'little'); pos += 32; assert v < R; return v
    witness = [(pt(), pt(), pt()) for _ in range(k)]
    assert data[pos] == 1; pos += 1
    for name in ('mask', 'g_1', 'h_1'): c[name] = pt()
    c['g_abc'] = _circuit_major([pt() for _ in range(3 * m)]); c['h_2'] = pt()
    if m == 1: c['g_a'], c['g_b'], c['g_c'] = c['g_abc']
    evals = [fr() for _ in range(k + 1 + 3 * m)]; evals = evals[:k + 1] + _circuit_major(evals[k + 1:])
    assert int.from_bytes(data[pos:pos + 8], 'little') == m; pos += 8
    sums = [fr() for _ in range(3 * m)]
    n_open = int.from_bytes(data[pos:pos + 8], 'little'); pos += 8
    openings = []
    for _ in range(n_open):
        p_ = pt(); tag = data[pos]; pos += 1
        openings.append((p_, fr() if tag else None))
    assert pos + 1 == len(data) and data[pos] == 0
    return {'batch_sizes': ks, 'instances': k, 'witness': witness, 'commitments': c, 'evaluations': evals, 'sums': sums, 'openings': openings}


def _as_batch(index, public_inputs):
    """(list of Index, list per circuit of lists of public inputs) from the single-circuit or the batch calling form."""
    if isinstance(index, (Index, VerifyingKey)):
        if public_inputs and not isinstance(public_inputs[0], (list, tuple)): public_inputs = [public_inputs]
        return [index], [list(public_inputs)]
    return list(index), [list(p_) for p_ in public_inputs]


def _verifier_state(index, public_inputs, data: bytes, vk_bytes=None):
    """Everything a verifier derives before the opening checks: parsed proof, challenges, the two linear-combination coefficient sets."""
    try: pr = parse_proof(data)
    except AssertionError: return None
    circuits, publics = _as_batch(index, public_inputs)
    m = len(circuits); ks = pr['batch_sizes']; k = pr['instances']
    if len(ks) != m or [len(p_) for p_ in publics] != ks: return None
    if vk_bytes is not None and isinstance(vk_bytes, (bytes, bytearray)): vk_bytes = [vk_bytes]
    try:
        cbs = pr['commitments']
        pts = {n_: P.g1_decompress(cbs[n_]) for n_ in ('mask', 'g_1', 'h_1', 'h_2')}; pts['g_abc'] = [P.g1_decompress(v) for v in cbs['g_abc']]
        opn = [P.g1_decompress(p_) for p_, _ in pr['openings']]
        wit = [tuple(P.g1_decompress(v) for v in t) for t in pr['witness']]
    except Exception: return None
    if len(opn) != 2 or pr['openings'][0][1] is None or pr['openings'][1][1] is not None: return None
    inst_circuit = [j for j, kj in enumerate(ks) for _ in range(kj)]
    x_evals = [[pub[i] % R if i < circuits[j].circuit.n_public else 0 for i in range(circuits[j].circuit.n_x)] for j, pubs in enumerate(publics) for pub in pubs]
    N = max(ix.circuit.n_h for ix in circuits); n_k = max(ix.circuit.n_k for ix in circuits)
    HN, KN = Domain(N), Domain(n_k)
    vkp = [vk_points_of(v) for v in vk_bytes] if vk_bytes is not None else [[ix.commit_points()[(M, kd)] for M in 'abc' for kd in ('row', 'col', 'val', 'row_col')] for ix in circuits]
    fs = fs_start(vkp, ks, x_evals)
    fs.absorb_points([p_ for t in wit for p_ in t] + [pts['mask']])
    alpha, eta, comb = _challenges_after_round1(fs, ks)
    fs.absorb_points([pts['g_1'], pts['h_1']]); beta = fs.squeeze_nonnative(1)[0]
    sigma = [dict(zip('abc', pr['sums'][3 * j:3 * j + 3])) for j in range(m)]
    fs.absorb_points(pts['g_abc'])
    for sj in sigma: fs.absorb_nonnative([sj['a'], sj['b'], sj['c']])
    delta = _challenges_after_round3(fs, m)
    fs.absorb_points([pts['h_2']]); gamma = fs.squeeze_nonnative(1)[0]
    evals = pr['evaluations']
    fs.absorb_nonnative(_serialized_evaluations(evals, k, m))
    ch_b = [fs.squeeze_short_one() for _ in range(k + 2)]; ch_g = [fs.squeeze_short_one() for _ in range(3 * m + 1)]
    zb_beta = evals[:k]; g1_beta = evals[k]; g_gamma = [dict(zip('abc', evals[k + 1 + 3 * j:k + 4 + 3 * j])) for j in range(m)]
    x_beta = [poly_eval(circuits[j].X.ifft(xe), beta) for j, xe in zip(inst_circuit, x_evals)]
    return {'k': k, 'm': m, 'circuits': circuits, 'n_h': N, 'pts': pts, 'opn': opn, 'wit': wit, 'beta': beta, 'gamma': gamma, 'ch_b': ch_b, 'ch_g': ch_g, 'random_v': pr['openings'][0][1],
            'lc1': lincheck_coefficients(circuits, inst_circuit, HN, alpha, beta, eta, comb, sigma, zb_beta, g1_beta, x_beta),
            'lc2': matrix_coefficients(circuits, KN, alpha, beta, gamma, delta, sigma, g_gamma),
            'v_beta': (ch_b[0] * g1_beta + sum(ch_b[1 + i] * v for i, v in enumerate(zb_beta))) % R,
            'v_gamma': sum(ch_g[3 * j + t] * g_gamma[j][M] for j in range(m) for t, M in enumerate('abc')) % R}


def _unshifted_parts(st):
    """The G1 sides of the two checks that carry no degree shift: sum_i ch_(1+i) z_b,i + ch_(k+1) LC1  and  ch'_(3m) LC2."""
    G = P.G1_GENERATOR; mul, add = P.g1_mul, P.g1_add
    pts, wit, lc1, lc2, ch_b, ch_g, k = st['pts'], st['wit'], st['lc1'], st['lc2'], st['ch_b'], st['ch_g'], st['k']
    C1 = add(add(mul(pts['mask'], lc1['mask']), mul(pts['h_1'], lc1['h_1'])), mul(G, lc1['const']))
    for (w_, a_, b_), kz, kw in zip(wit, lc1['z_a'], lc1['w']): C1 = add(C1, add(mul(a_, kz), mul(w_, kw)))
    Rb = mul(C1, ch_b[k + 1])
    for i, (w_, a_, b_) in enumerate(wit): Rb = add(Rb, mul(b_, ch_b[1 + i]))
    C2 = add(mul(G, lc2['const']), mul(pts['h_2'], lc2['h_2']))
    index_points = [ix.commit_points() for ix in st['circuits']]
    for (j, M, kind), coef in lc2['index'].items(): C2 = add(C2, mul(index_points[j][(M, kind)], coef))
    return Rb, mul(C2, ch_g[3 * st['m']])


def verify_pairing(index, vk, public_inputs, data: bytes, vk_bytes=None) -> bool:
    """The verifier proper: no trapdoor.  vk = Setup.verifier_key(circuit or list of circuits): gamma G, H, tau H and the negative powers of tau in
    G2 the degree bounds need; each circuit's verifying key = its twelve index commitments.  index / public_inputs: one Index with the public inputs of
    its instances, or lists of both (a proof over several circuits).  Both batched KZG openings are checked as pairing products
      e(shifted commitments, tau^-s H) · e(unshifted part − v G − v̄ gamma G, H) · e(−W, tau H − z H) = 1   (oracle/pairing.py)."""
    from . import pairing as E
    st = _verifier_state(index, public_inputs, data, vk_bytes)
    if st is None: return False
    G = P.G1_GENERATOR; mul, add, neg = P.g1_mul, P.g1_add, P.g1_neg
    pts, ch_b, ch_g = st['pts'], st['ch_b'], st['ch_g']
    Rb, Rg = _unshifted_parts(st)
    Rb = add(Rb, neg(add(mul(G, st['v_beta']), mul(vk['gamma_g'], st['random_v']))))
    Rg = add(Rg, neg(mul(G, st['v_gamma'])))
    zh = lambda z: P.g2_add(vk['tau_h'], P.g2_neg(P.g2_mul(vk['h'], z)))
    if not E.pairing_product_is_one([(mul(pts['g_1'], ch_b[0]), vk['neg_h']), (Rb, vk['h']), (neg(st['opn'][0]), zh(st['beta']))]): return False
    shifted = {}                                                                  # sum ch'_(3j+M) g_{j,M}, grouped by the shift of their degree bound
    for j, ix in enumerate(st['circuits']):
        for t, M in enumerate('abc'):
            key = ix.circuit.n_k_m[M]; term = mul(pts['g_abc'][3 * j + t], ch_g[3 * j + t])
            shifted[key] = (add(shifted[key][0], term), shifted[key][1]) if key in shifted else (term, vk['neg_k_by_size'][key])
    return E.pairing_product_is_one(list(shifted.values()) + [(Rg, vk['h']), (neg(st['opn'][1]), zh(st['gamma']))])


def verify(index, setup: Setup, public_inputs, data: bytes, vk_bytes=None) -> bool:
    """The same two checks without pairings, for tests that verify many proofs: with the synthetic setup's trapdoor the pairing equation
    e(C − v·G − v̄·γG, H) = e(W, (τ − z)·H) is the G1 equation C − v·G − v̄·γG = (τ − z)·W (verify_pairing is the verifier that needs no trapdoor)."""
    st = _verifier_state(index, public_inputs, data, vk_bytes)
    if st is None: return False
    G = P.G1_GENERATOR; mul, add, neg = P.g1_mul, P.g1_add, P.g1_neg
    pts, ch_b, ch_g = st['pts'], st['ch_b'], st['ch_g']
    tau_inv = inv(setup.tau); D = setup.max_degree
    def unshift(pt, bound): return mul(pt, pow(tau_inv, D - bound, R))
    Rb, Rg = _unshifted_parts(st)
    lhs = add(add(mul(unshift(pts['g_1'], st['n_h'] - 2), ch_b[0]), Rb), neg(mul(G, (st['v_beta'] + setup.s_gamma * st['random_v']) % R)))
    if lhs != mul(st['opn'][0], (setup.tau - st['beta']) % R): return False
    Sg = None
    for j, ix in enumerate(st['circuits']):
        for t, M in enumerate('abc'): Sg = add(Sg, mul(unshift(pts['g_abc'][3 * j + t], ix.circuit.n_k_m[M] - 2), ch_g[3 * j + t]))
    lhs = add(add(Sg, Rg), neg(mul(G, st['v_gamma'])))
    return lhs == mul(st['opn'][1], (setup.tau - st['gamma']) % R)

"""Deterministic synthetic inputs for the MSM / NTT workloads: SplitMix64-seeded scalars.
No dependency on oracle/: bench.py and the product-side smoke path use this; tests/util.py re-exports it."""
from __future__ import annotations
import numpy as np

FR_MODULUS = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
FQ_MODULUS = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
FQ_R = 1 << 384
G1_GENERATOR = (
    89363714989903307245735717098563574705733591463163614225748337416674727625843187853442697973404985688481508350822,
    3702177272937190650578065972808860481433820514072818216637796320125658674906330993856598323293086021583822603349,
)
_M64 = (1 << 64) - 1
_R_LIMBS = np.array([(FR_MODULUS >> (64 * i)) & _M64 for i in range(4)], dtype=np.uint64)


def splitmix_limbs(seed: int, count: int) -> np.ndarray:
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = np.uint64(seed & _M64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _lt_r(a: np.ndarray) -> np.ndarray:
    lt = np.zeros(a.shape[0], dtype=bool); eq = np.ones(a.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        lt |= eq & (a[:, i] < _R_LIMBS[i]); eq &= a[:, i] == _R_LIMBS[i]
    return lt


def uniform_scalars(n: int, seed: int) -> np.ndarray:
    """n canonical Fr values uint64[n,4]: 253-bit rejection sampling."""
    out = np.zeros((n, 4), dtype=np.uint64); todo = np.arange(n); rnd = 0
    while todo.size:
        a = splitmix_limbs(seed + 7919 * rnd, 4 * todo.size).reshape(-1, 4).copy()
        a[:, 3] &= np.uint64((1 << 61) - 1)
        ok = _lt_r(a)
        out[todo[ok]] = a[ok]; todo = todo[~ok]; rnd += 1
    return out


def witness_like_scalars(n: int, seed: int) -> np.ndarray:
    """60% zero, 20% one, 10% < 2^16, 10% uniform — the skew of real R1CS witnesses."""
    s = uniform_scalars(n, seed)
    sel = splitmix_limbs(seed ^ 0x5151, n) % np.uint64(10)
    z = sel < 6; o = (sel >= 6) & (sel < 8); sm = sel == 8
    s[z] = 0
    s[o] = 0; s[o, 0] = 1
    s[sm, 1:] = 0; s[sm, 0] &= np.uint64(0xFFFF)
    return s


def int_to_limbs(v: int, nl: int) -> np.ndarray:
    return np.array([(v >> (64 * i)) & _M64 for i in range(nl)], dtype=np.uint64)


def limbs_to_int(a) -> int:
    return sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(a, dtype=np.uint64).reshape(-1)))


def generator_affine104() -> np.ndarray:
    """The G1 generator as a snarkVM Affine (Montgomery x, y + infinity byte)."""
    out = np.zeros(104, dtype=np.uint8)
    xm = (G1_GENERATOR[0] * FQ_R) % FQ_MODULUS; ym = (G1_GENERATOR[1] * FQ_R) % FQ_MODULUS
    out[0:48] = int_to_limbs(xm, 6).view(np.uint8); out[48:96] = int_to_limbs(ym, 6).view(np.uint8)
    return out


def weighted_scalar_sum(scalars: np.ndarray, first_multiple: int = 1) -> int:
    """sum_i s_i * (first_multiple + i) mod r — the discrete log (base G) of an MSM over bases (first+i)*G.
    Exact integer arithmetic in numpy: 16-bit pieces of the scalars times weights below 2^31, 2^15 terms per dot product."""
    s = np.ascontiguousarray(np.asarray(scalars, dtype=np.uint64).reshape(-1, 4))
    n = s.shape[0]
    if n == 0: return 0
    if first_multiple + n >= (1 << 31):      # weights too wide for the fast path
        w = np.arange(first_multiple, first_multiple + n, dtype=object); total = 0
        for limb in range(4): total += int((s[:, limb].astype(object) * w).sum()) << (64 * limb)
        return total % FR_MODULUS
    q = s.view(np.uint16).reshape(n, 16).astype(np.uint64)           # little-endian 16-bit pieces
    w = np.arange(first_multiple, first_multiple + n, dtype=np.uint64)
    total = 0
    for lo in range(0, n, 1 << 15):                                    # 2^16 * 2^31 * 2^15 = 2^62 < 2^64
        part = w[lo:lo + (1 << 15)] @ q[lo:lo + (1 << 15)]             # uint64[16], exact
        total += sum(int(v) << (16 * j) for j, v in enumerate(part))
    return total % FR_MODULUS


def synthetic_r1cs(n_constraints: int, n_public: int, seed: int, long_rows: int = 4):
    """A satisfiable R1CS with its assignment, shaped like a compiled program (SURVEY.md §8d "synthetic witness vectors"): constraint
    i multiplies two short linear combinations of earlier variables into a NEW private variable (C row = that variable), except for
    `long_rows` constraints whose A side is a long linear combination (the few wide rows real circuits have).  Variables: public first
    (z_0 = 1).  Returns (csr, z): csr[m] = (row_ptr uint32[n+1], col uint32[nnz], val uint64[nnz,4] canonical) for m in 'abc',
    z = python ints."""
    r = FR_MODULUS
    rnd = splitmix_limbs(seed, 8 * n_constraints + 64)
    small = uniform_scalars(n_constraints * 2 + n_public + 8, seed ^ 0x77)
    coef = [limbs_to_int(x) for x in small]
    z = [1] + [coef[n_constraints * 2 + i] for i in range(1, n_public)]
    rows = {'a': [], 'b': [], 'c': []}
    wide = set(int(x) % n_constraints for x in rnd[-long_rows:]) if long_rows else set()
    for i in range(n_constraints):
        nv = len(z)
        def lc(k, salt):
            out = {}
            for j in range(k):
                v = int(rnd[(8 * i + 4 * salt + j) % len(rnd)] % np.uint64(nv))
                c_ = coef[(2 * i + salt + 3 * j) % len(coef)] if j else 1 + (int(rnd[8 * i + salt]) & 0xFFFF)
                out[v] = (out.get(v, 0) + c_) % r
            return sorted(out.items())
        ka = 1 + (int(rnd[8 * i + 6]) % 3); kb = 1 + (int(rnd[8 * i + 7]) % 2)
        if i in wide: ka = min(nv, 200)
        a = lc(ka, 0) if i not in wide else sorted({(int(rnd[(8 * i + j) % len(rnd)] % np.uint64(nv))): coef[(i + j) % len(coef)] for j in range(ka)}.items())
        b = lc(kb, 1)
        va = sum(c_ * z[v] for v, c_ in a) % r; vb = sum(c_ * z[v] for v, c_ in b) % r
        z.append(va * vb % r)
        rows['a'].append(a); rows['b'].append(b); rows['c'].append([(nv, 1)])
    csr = {}
    for m, rr in rows.items():
        ptr = np.zeros(n_constraints + 1, dtype=np.uint32); ptr[1:] = np.cumsum([len(x) for x in rr])
        col = np.array([v for x in rr for v, _ in x], dtype=np.uint32)
        flat = [c_ for x in rr for _, c_ in x]
        val = np.zeros((len(flat), 4), dtype=np.uint64)
        for limb in range(4): val[:, limb] = np.array([(c_ >> (64 * limb)) & _M64 for c_ in flat], dtype=np.uint64)
        csr[m] = (ptr, col, val)
    return csr, z


def resolve_synthetic(csr, n_public: int, publics) -> list:
    """Another assignment of a synthetic circuit: the same constraints solved for other public inputs (publics[0] must be 1).
    Constraint i defines variable n_public + i: it appears in row i of C (coefficient c_n), everything else in the row is known:
    z_new = ((A_i . z)(B_i . z) − sum of the known terms of C_i) / c_n."""
    r = FR_MODULUS
    z = [int(v) % r for v in publics]
    assert len(z) == n_public and z[0] == 1
    (pa, ca, va), (pb_, cb, vb), (pc, cc, vc) = csr['a'], csr['b'], csr['c']
    va_i = [limbs_to_int(x) for x in va]; vb_i = [limbs_to_int(x) for x in vb]; vc_i = [limbs_to_int(x) for x in vc]
    for i in range(len(pa) - 1):
        a = sum(va_i[k] * z[ca[k]] for k in range(pa[i], pa[i + 1])) % r
        b = sum(vb_i[k] * z[cb[k]] for k in range(pb_[i], pb_[i + 1])) % r
        known, cn = 0, None
        for k in range(pc[i], pc[i + 1]):
            if cc[k] == n_public + i: cn = vc_i[k]
            else: known += vc_i[k] * z[cc[k]]
        z.append((a * b - known) * pow(cn, -1, r) % r)
    return z


def synthetic_r1cs_bits(n_constraints: int, n_public: int, seed: int):
    """A satisfiable R1CS whose assignment looks like a compiled program's — mostly bits, evenly 0 and 1: 50 % XOR gates of two earlier bits
    ((2a) b = a + b − c), 20 % AND gates, 20 % negations ((1 − a) 1 = c), 10 % products of two short linear combinations with small coefficients
    (a field element).  Same return shape as synthetic_r1cs; resolve_synthetic solves it for other public inputs."""
    r = FR_MODULUS
    rnd = splitmix_limbs(seed, 6 * n_constraints + 64)
    z = [1] + [int(rnd[-1 - i] & np.uint64(1)) for i in range(1, n_public)]
    bits = list(range(1, n_public)) or [0]                            # variables known to hold 0 / 1
    rows = {'a': [], 'b': [], 'c': []}
    for i in range(n_constraints):
        nv = len(z); kind = int(rnd[6 * i] % np.uint64(10))
        pick = lambda j: bits[int(rnd[6 * i + j] % np.uint64(len(bits)))]
        if kind < 5:                                                   # XOR
            u, v = pick(1), pick(2); a, b, c_ = [(u, 2)], [(v, 1)], sorted({u: 1, v: 1}.items()) if u != v else [(u, 2)]
            c_ = c_ + [(nv, r - 1)]; new = (z[u] + z[v] - 2 * z[u] * z[v]) % r
        elif kind < 7:                                                 # AND
            u, v = pick(1), pick(2); a, b, c_ = [(u, 1)], [(v, 1)], [(nv, 1)]; new = z[u] * z[v] % r
        elif kind < 9:                                                 # NOT
            u = pick(1); a, b, c_ = ([(0, 1), (u, r - 1)] if u else [(0, 0)]), [(0, 1)], [(nv, 1)]; new = (1 - z[u]) % r if u else 0
        else:                                                          # (small combination) * (small combination)
            a = sorted({int(rnd[6 * i + 1] % np.uint64(nv)): 1 + int(rnd[6 * i + 3] & np.uint64(0xFF)), pick(2): 1}.items())
            b = sorted({int(rnd[6 * i + 4] % np.uint64(nv)): 1 + int(rnd[6 * i + 5] & np.uint64(0xF))}.items()); c_ = [(nv, 1)]
            new = sum(k_ * z[v] for v, k_ in a) % r * (sum(k_ * z[v] for v, k_ in b) % r) % r
        z.append(new)
        if kind < 9: bits.append(nv)
        rows['a'].append(a); rows['b'].append(b); rows['c'].append(c_)
    csr = {}
    for m, rr in rows.items():
        ptr = np.zeros(n_constraints + 1, dtype=np.uint32); ptr[1:] = np.cumsum([len(x) for x in rr])
        col = np.array([v for x in rr for v, _ in x], dtype=np.uint32)
        flat = [c_ for x in rr for _, c_ in x]
        val = np.zeros((len(flat), 4), dtype=np.uint64)
        for limb in range(4): val[:, limb] = np.array([(c_ >> (64 * limb)) & _M64 for c_ in flat], dtype=np.uint64)
        csr[m] = (ptr, col, val)
    return csr, z


def _csr_from_rows(rows, n_constraints):
    csr = {}
    for m, rr in rows.items():
        ptr = np.zeros(n_constraints + 1, dtype=np.uint32); ptr[1:] = np.cumsum([len(x) for x in rr])
        col = np.array([v for x in rr for v, _ in x], dtype=np.uint32)
        flat = [c_ for x in rr for _, c_ in x]
        val = np.zeros((len(flat), 4), dtype=np.uint64)
        for limb in range(4): val[:, limb] = np.array([(c_ >> (64 * limb)) & _M64 for c_ in flat], dtype=np.uint64)
        csr[m] = (ptr, col, val)
    return csr


def synthetic_r1cs_density(n_constraints: int, n_public: int, seed: int, nnz_a: int, nnz_b: int):
    """The circuit of synthetic_r1cs with the row density as a parameter: constraint i multiplies a combination of nnz_a earlier variables by a
    combination of nnz_b earlier variables into a new private variable (row of C = that variable).  For the density sweep of the prover: the
    non-zero domains |K_A|, |K_B| — and with them the MSM points per constraint — grow with nnz_a, nnz_b."""
    r = FR_MODULUS
    rnd = splitmix_limbs(seed, (nnz_a + nnz_b) * n_constraints + 64)
    coef = [limbs_to_int(x) for x in uniform_scalars(4096, seed ^ 0x55)]
    z = [1] + [coef[i] for i in range(1, n_public)]
    rows = {'a': [], 'b': [], 'c': []}; at = 0
    for i in range(n_constraints):
        nv = len(z)
        def lc(k):
            nonlocal at
            out = {}
            for j in range(min(k, nv)):
                v = int(rnd[at] % np.uint64(nv)); at += 1
                out[v] = (out.get(v, 0) + coef[(at + 7 * j) & 4095]) % r or 1
            return sorted(out.items())
        a, b = lc(nnz_a), lc(nnz_b)
        z.append(sum(c_ * z[v] for v, c_ in a) % r * (sum(c_ * z[v] for v, c_ in b) % r) % r)
        rows['a'].append(a); rows['b'].append(b); rows['c'].append([(nv, 1)])
    return _csr_from_rows(rows, n_constraints), z


def synthetic_r1cs_poseidon(n_constraints: int, n_public: int, seed: int, width: int = 9):
    """A circuit shaped like Poseidon gadgets (what Aleo programs are full of: hash_psd2/4/8, BHP commitments have the same flavour): a state of
    `width` elements goes through rounds; in a round every element becomes (MDS row . state + round constant)^17 — five constraints:
    L * L = t2 (both sides the dense row: width + 1 non-zeros, the constant on variable 0), t2 * t2 = t4, t4 * t4 = t8, t8 * t8 = t16,
    t16 * L = t17.  Dense MDS rows, x^17 chains; 2.8 / 4.6 non-zeros per constraint in A / B at width 9."""
    r = FR_MODULUS
    coef = [limbs_to_int(x) | 1 for x in uniform_scalars(width * width + 64, seed ^ 0x99)]
    z = [1] + [coef[-1 - i] for i in range(1, n_public)]
    while len(z) < max(n_public, 1) + width: z.append(coef[len(z) % len(coef)])      # initial state: the public inputs, then private seeds
    first_state = len(z) - width
    rows = {'a': [], 'b': [], 'c': []}
    state = list(range(first_state, first_state + width)); rnd_c = 0
    for v in range(n_public, len(z)):                                                # the private seeds are pinned by (seed) * 1 = v-style constraints later: here they are free inputs
        pass
    while len(rows['a']) + 5 <= n_constraints:
        nxt = []
        for i in range(width):
            if len(rows['a']) + 5 > n_constraints: break
            L = sorted([(0, coef[(rnd_c + i) % len(coef)])] + [(state[j], coef[i * width + j]) for j in range(width)]); rnd_c += 1
            lv = sum(c_ * z[v] for v, c_ in L) % r
            t = lv
            for step in range(4):                                                    # t2, t4, t8, t16
                nv = len(z); a = L if step == 0 else [(nv - 1, 1)]; z.append(t * t % r); t = z[-1]
                rows['a'].append(a); rows['b'].append(a); rows['c'].append([(nv, 1)])
            nv = len(z); z.append(t * lv % r)
            rows['a'].append([(nv - 1, 1)]); rows['b'].append(L); rows['c'].append([(nv, 1)]); nxt.append(nv)
        if len(nxt) == width: state = nxt
        else: break
    while len(rows['a']) < n_constraints:                                            # fill: squares of the last variable
        nv = len(z); z.append(z[-1] * z[-1] % r)
        rows['a'].append([(nv - 1, 1)]); rows['b'].append([(nv - 1, 1)]); rows['c'].append([(nv, 1)])
    return _csr_from_rows(rows, n_constraints), z


def poseidon_parameters(rate: int = 2):
    """(ark 39 x (rate + 1), mds (rate + 1) x (rate + 1)) of snarkVM's Poseidon over Fr as python ints, from the library (aleo_mi355x_poseidon_parameters_fr)."""
    import ctypes
    from ._lib import lib, check
    w = rate + 1
    ark = np.zeros((39 * w, 4), dtype=np.uint64); mds = np.zeros((w * w, 4), dtype=np.uint64)
    check(lib().aleo_mi355x_poseidon_parameters_fr(rate, ark.ctypes.data_as(ctypes.c_void_p), mds.ctypes.data_as(ctypes.c_void_p)), 'poseidon_parameters_fr')
    A = [limbs_to_int(x) for x in ark]; Mx = [limbs_to_int(x) for x in mds]
    return [A[r * w:(r + 1) * w] for r in range(39)], [Mx[i * w:(i + 1) * w] for i in range(w)]


def poseidon_chain_r1cs(n_hashes: int, seed: int):
    """The R1CS of a REAL gadget instead of a synthetic shape: a chain of `hash_psd2` (snarkVM's Poseidon, rate 2 — the `hash.psd2` opcode of Aleo
    instructions, the hash of the state-tree paths `Trace::prepare` attaches): h_0 private, h_(k+1) = hash_psd2([h_k, s_k]) with private siblings s_k,
    the final value public — a Merkle path without the left / right bits.  Per hash: the permutation over the constant-prefix state (the first
    permutation of [domain, 2] has no variable input and folds into constants), 8 full rounds x 3 + 31 partial rounds x 1 s-boxes, each x^17 as
    five constraints (L * L = x2, x2 * x2 = x4, x4 * x4 = x8, x8 * x8 = x16, x16 * L = x17) over linear combinations L that grow through the
    partial rounds exactly as in the gadget (up to ~35 terms: the dense rows of real circuits), + one constraint binding the output.
    276 constraints per hash.  Variables: [1, root] public.  Returns (csr, z, root); the witness is computed in python integers."""
    r = FR_MODULUS
    ark, mds = poseidon_parameters(2)
    dom = int.from_bytes(b'AleoPoseidon2', 'little') % r
    def permute_plain(st):
        for i in range(39):
            st = [(a + b) % r for a, b in zip(st, ark[i])]
            if 4 <= i < 35: st[0] = pow(st[0], 17, r)
            else: st = [pow(v, 17, r) for v in st]
            st = [sum(st[j] * mds[k][j] for j in range(3)) % r for k in range(3)]
        return st
    pre = permute_plain([0, dom, 2])                                  # constant: the state after absorbing [domain, length]
    coef = [limbs_to_int(x) for x in uniform_scalars(n_hashes + 1, seed)]
    z = [1, 0]                                                         # z[1] = root, filled at the end
    rows = {'a': [], 'b': [], 'c': []}
    def lc_add(a, b):
        out = dict(a)
        for v, c_ in b.items(): out[v] = (out.get(v, 0) + c_) % r
        return out
    def lc_scale(a, k): return {v: c_ * k % r for v, c_ in a.items()}
    def lc_val(a): return sum(c_ * z[v] for v, c_ in a.items()) % r
    def lc_row(a): return sorted((v, c_) for v, c_ in a.items() if c_)
    def new_var(val): z.append(val % r); return len(z) - 1
    def sbox(L):
        x = lc_val(L); cur_lc, cur = L, x
        for _ in range(4):                                            # x2, x4, x8, x16
            v = new_var(cur * cur); rows['a'].append(lc_row(cur_lc)); rows['b'].append(lc_row(cur_lc)); rows['c'].append([(v, 1)])
            cur_lc, cur = {v: 1}, z[v]
        v = new_var(cur * x); rows['a'].append(lc_row(cur_lc)); rows['b'].append(lc_row(L)); rows['c'].append([(v, 1)])
        return {v: 1}
    h = new_var(coef[0]); h_lc = {h: 1}
    for k in range(n_hashes):
        s = new_var(coef[k + 1])
        st = [{0: pre[0]}, lc_add({0: pre[1]}, h_lc), lc_add({0: pre[2]}, {s: 1})]      # absorb [h_k, s_k] into the rate part
        for i in range(39):
            st = [lc_add(st[j], {0: ark[i][j]}) for j in range(3)]
            if 4 <= i < 35: st[0] = sbox(st[0])
            else: st = [sbox(x) for x in st]
            st = [lc_add(lc_add(lc_scale(st[0], mds[kk][0]), lc_scale(st[1], mds[kk][1])), lc_scale(st[2], mds[kk][2])) for kk in range(3)]
        out_lc = st[1]                                                 # squeeze: the first rate element
        if k + 1 < n_hashes:
            v = new_var(lc_val(out_lc)); rows['a'].append(lc_row(out_lc)); rows['b'].append([(0, 1)]); rows['c'].append([(v, 1)])
            h_lc = {v: 1}
        else:
            z[1] = lc_val(out_lc); rows['a'].append(lc_row(out_lc)); rows['b'].append([(0, 1)]); rows['c'].append([(1, 1)])
    n = len(rows['a'])
    # the prover's layout wants the public variables first: they are (0 = the constant 1, 1 = root) already
    return _csr_from_rows(rows, n), z, z[1]


# ---- G2 (BLS12-377 twist over Fq2 = Fq[u] / (u^2 + 5)): synthetic base sets and an O(n) result gate for benchmarks ---------------------------------
# Plain Python integers, independent of the library and of oracle/ (bench.py may use oracle/ only in its cpu_baseline leg).
G2_GENERATOR = ((233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294,
                 140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118),
                (63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423,
                 149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491))


def _f2mul(a, b): q = FQ_MODULUS; return ((a[0] * b[0] - 5 * a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)
def _f2sub(a, b): q = FQ_MODULUS; return ((a[0] - b[0]) % q, (a[1] - b[1]) % q)
def _f2inv(a):
    q = FQ_MODULUS; n = pow((a[0] * a[0] + 5 * a[1] * a[1]) % q, -1, q)
    return (a[0] * n % q, (-a[1] * n) % q)


def g2_add_affine(P, S):
    """P + S on y^2 = x^3 + b' over Fq2 (affine tuples ((x0, x1), (y0, y1)), None = the identity)."""
    if P is None: return S
    if S is None: return P
    if P[0] == S[0]:
        if P[1] != S[1] or P[1] == (0, 0): return None
        lam = _f2mul(_f2mul((3, 0), _f2mul(P[0], P[0])), _f2inv(_f2mul((2, 0), P[1])))
    else:
        lam = _f2mul(_f2sub(S[1], P[1]), _f2inv(_f2sub(S[0], P[0])))
    x = _f2sub(_f2sub(_f2mul(lam, lam), P[0]), S[0])
    return (x, _f2sub(_f2mul(lam, _f2sub(P[0], x)), P[1]))


def g2_times(P, k: int):
    acc = None
    for bit in bin(k % FR_MODULUS)[2:] if k % FR_MODULUS else '':
        acc = g2_add_affine(acc, acc)
        if bit == '1': acc = g2_add_affine(acc, P)
    return acc


def g2_affine200(points) -> np.ndarray:
    """Affine tuples -> snarkVM G2Affine rows (x.c0 | x.c1 | y.c0 | y.c1 Montgomery, infinity byte at 192), uint8[n, 200]."""
    out = np.zeros((len(points), 200), dtype=np.uint8)
    for i, P in enumerate(points):
        if P is None: out[i, 192] = 1; continue
        for j, v in enumerate((P[0][0], P[0][1], P[1][0], P[1][1])): out[i, 48 * j: 48 * j + 48] = int_to_limbs(v * FQ_R % FQ_MODULUS, 6).view(np.uint8)
    return out


def g2_multiples_affine200(n: int, distinct: int = 1 << 12) -> np.ndarray:
    """n G2 bases P_i = ((i mod distinct) + 1) * G2 as snarkVM rows: `distinct` points by repeated affine addition (~ 40 us each), tiled to n — the
    discrete logarithm of base i is known, so sum_i s_i * ((i mod distinct) + 1) * G2 is the expected MSM result (g2_result_gate)."""
    d = min(n, distinct); pts, acc = [], None
    for _ in range(d): acc = g2_add_affine(acc, G2_GENERATOR); pts.append(acc)
    rows = g2_affine200(pts)
    return np.ascontiguousarray(np.tile(rows, ((n + d - 1) // d, 1))[:n])


def g2_result_gate(result_jac288, scalars: np.ndarray, distinct: int = 1 << 12) -> bool:
    """True iff the library's G2 result (Jacobian x, y, z: Fq2, affine-normalised) equals (sum_i s_i * ((i mod distinct) + 1)) * G2 in Python integers."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4); n = s.shape[0]
    w = (np.arange(n, dtype=np.uint64) % np.uint64(max(1, min(n, distinct)))) + np.uint64(1)
    q = s.view(np.uint16).reshape(n, 16).astype(np.uint64); k = 0      # 16-bit pieces x weights below 2^31, 2^15 terms per dot product: exact in uint64
    for lo in range(0, n, 1 << 15):
        part = w[lo:lo + (1 << 15)] @ q[lo:lo + (1 << 15)]
        k += sum(int(v) << (16 * j) for j, v in enumerate(part))
    want = g2_times(G2_GENERATOR, k % FR_MODULUS)
    r = np.asarray(result_jac288, dtype=np.uint64).reshape(36); rinv = pow(FQ_R, -1, FQ_MODULUS)
    lim = lambda a: sum(int(v) << (64 * i) for i, v in enumerate(a)) * rinv % FQ_MODULUS
    got = None if not r[24:].any() else ((lim(r[0:6]), lim(r[6:12])), (lim(r[12:18]), lim(r[18:24])))
    return got == want

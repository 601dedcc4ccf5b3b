"""Python big-integer reference for the BLS12-377 operators on the Aleo prove path.

TEST INFRASTRUCTURE ONLY.  Nothing outside tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module (see DESIGN.md "Oracle").  It is the slow, obviously-correct
restatement used to (a) generate the golden vectors under tests/golden/ and (b) pin the C oracle
(oracle/oracle.c) on small cases.

PARITY UNPINNED for MSM/NTT *values*: the arithmetic lives in crates.io snarkvm-{fields,curves,algorithms}
=0.14.5 (pinned at /root/reference/Cargo.lock:2200,2637,2652), which are absent from /root/reference and
cannot be built here (no Rust toolchain).  What pins this file instead:
  * the ten KZG commitments inside the `proof1…` string the reference's own test holds
    (wasm/src/programs/transaction.rs:100) must decompress to on-curve, r-torsion points — pins q, the
    curve equation, Fq sqrt, the compressed encoding (tests/golden/reference_proof.json);
  * the field evaluations in the same proof must be canonical (< r);
  * TWO_ADIC_ROOT_OF_UNITY = 22^((r-1)/2^47).

Algorithms restated (published behaviour of snarkVM 0.14.5, upstream-relative paths):
  fields/src/fp_256.rs, fp_384.rs            Montgomery form, R = 2^256 / 2^384, little-endian u64 limbs
  curves/src/bls12_377/{fr,fq,g1}.rs         constants
  curves/src/templates/short_weierstrass_jacobian/{affine,projective}.rs
  algorithms/src/msm/variable_base/*.rs      VariableBase::msm = sum_i s_i * P_i
  algorithms/src/fft/domain.rs               EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}
"""
from __future__ import annotations

# ----------------------------------------------------------------------------------------------
# Constants (curves/src/bls12_377/{fr,fq,g1}.rs; re-derived in SURVEY.md §8 row a3)
# ----------------------------------------------------------------------------------------------
FR_MODULUS = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
FQ_MODULUS = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
FR_BITS, FQ_BITS = 253, 377
FR_R = 1 << 256          # Montgomery radix for Fp256
FQ_R = 1 << 384          # Montgomery radix for Fp384
FR_GENERATOR = 22        # Fr multiplicative generator == coset shift of EvaluationDomain
FR_TWO_ADICITY = 47
FR_TWO_ADIC_ROOT = 8065159656716812877374967518403273466521432693661810619979959746626482506078
G1_COEFF_A, G1_COEFF_B = 0, 1
G1_GENERATOR = (
    89363714989903307245735717098563574705733591463163614225748337416674727625843187853442697973404985688481508350822,
    3702177272937190650578065972808860481433820514072818216637796320125658674906330993856598323293086021583822603349,
)
G1_COFACTOR = 0x170B5D44300000000000000000000000


def fr_to_mont(a: int) -> int: return (a * FR_R) % FR_MODULUS
def fr_from_mont(a: int) -> int: return (a * pow(FR_R, -1, FR_MODULUS)) % FR_MODULUS
def fq_to_mont(a: int) -> int: return (a * FQ_R) % FQ_MODULUS
def fq_from_mont(a: int) -> int: return (a * pow(FQ_R, -1, FQ_MODULUS)) % FQ_MODULUS


# ----------------------------------------------------------------------------------------------
# G1: y^2 = x^3 + 1 over Fq.  Affine points are (x, y) tuples; None is the point at infinity.
# ----------------------------------------------------------------------------------------------
Q = FQ_MODULUS


def g1_is_on_curve(P) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - (x * x * x + G1_COEFF_B)) % Q == 0


def g1_neg(P):
    return None if P is None else (P[0], (-P[1]) % Q)


def g1_add(P, R):
    if P is None: return R
    if R is None: return P
    x1, y1 = P; x2, y2 = R
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    y3 = (lam * (x1 - x3) - y1) % Q
    return (x3, y3)


# Jacobian arithmetic for speed in the generator (X/Z^2, Y/Z^3); (1,1,0) is infinity.
def _jac_double(P):
    X, Y, Z = P
    if Z == 0: return P
    A = X * X % Q; B = Y * Y % Q; C = B * B % Q
    D = 2 * ((X + B) * (X + B) - A - C) % Q
    E = 3 * A % Q; F = E * E % Q
    X3 = (F - 2 * D) % Q
    Y3 = (E * (D - X3) - 8 * C) % Q
    Z3 = 2 * Y * Z % Q
    return (X3, Y3, Z3)


def _jac_add_mixed(P, A):
    """P Jacobian += A affine (not infinity)."""
    X1, Y1, Z1 = P
    x2, y2 = A
    if Z1 == 0: return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % Q
    U2 = x2 * Z1Z1 % Q
    S2 = y2 * Z1 % Q * Z1Z1 % Q
    if U2 == X1:
        if S2 == Y1: return _jac_double(P)
        return (1, 1, 0)
    H = (U2 - X1) % Q; HH = H * H % Q
    I = 4 * HH % Q; J = H * I % Q
    r = 2 * (S2 - Y1) % Q; V = X1 * I % Q
    X3 = (r * r - J - 2 * V) % Q
    Y3 = (r * (V - X3) - 2 * Y1 * J) % Q
    Z3 = ((Z1 + H) * (Z1 + H) - Z1Z1 - HH) % Q
    return (X3, Y3, Z3)


def _jac_to_affine(P):
    X, Y, Z = P
    if Z == 0: return None
    zi = pow(Z, -1, Q); zi2 = zi * zi % Q
    return (X * zi2 % Q, Y * zi2 % Q * zi % Q)


def g1_mul(P, k: int):
    """k * P, double-and-add, k >= 0 (no reduction mod r: callers pass canonical scalars)."""
    if P is None or k == 0: return None
    acc = (1, 1, 0)
    for bit in bin(k)[2:]:
        acc = _jac_double(acc)
        if bit == '1':
            acc = _jac_add_mixed(acc, P)
    return _jac_to_affine(acc)


def g1_in_subgroup(P) -> bool:
    return g1_mul(P, FR_MODULUS) is None


def msm_naive(bases, scalars):
    """VariableBase::msm semantics: sum_i scalars[i] * bases[i] (scalars canonical ints < r)."""
    acc = (1, 1, 0)
    for P, s in zip(bases, scalars):
        T = g1_mul(P, s)
        if T is not None:
            acc = _jac_add_mixed(acc, T)
    return _jac_to_affine(acc)


# ----------------------------------------------------------------------------------------------
# Fq square root (Tonelli–Shanks; two-adicity 46) and the compressed G1 encoding
# (utilities/src/serialize + curves/.../short_weierstrass_jacobian/affine.rs:
#  48 bytes little-endian x; in the last byte bit 7 = "y is the lexicographically larger root",
#  bit 6 = infinity).
# ----------------------------------------------------------------------------------------------
def fq_sqrt(a: int):
    a %= Q
    if a == 0: return 0
    if pow(a, (Q - 1) // 2, Q) != 1: return None
    s, t = 0, Q - 1
    while t % 2 == 0: s += 1; t //= 2
    z = 2
    while pow(z, (Q - 1) // 2, Q) == 1: z += 1
    c = pow(z, t, Q); x = pow(a, (t + 1) // 2, Q); b = pow(a, t, Q); m = s
    while b != 1:
        i, b2 = 0, b
        while b2 != 1: b2 = b2 * b2 % Q; i += 1
        e = pow(c, 1 << (m - i - 1), Q)
        x = x * e % Q; c = e * e % Q; b = b * c % Q; m = i
    return x


def fr_sqrt(a: int):
    """Tonelli-Shanks over Fr (two-adicity 47); None for a non-residue."""
    r = FR_MODULUS; a %= r
    if a == 0: return 0
    if pow(a, (r - 1) // 2, r) != 1: return None
    s, t = FR_TWO_ADICITY, (r - 1) >> FR_TWO_ADICITY
    c = FR_TWO_ADIC_ROOT; x = pow(a, (t + 1) // 2, r); b = pow(a, t, r); m = s
    while b != 1:
        i, b2 = 0, b
        while b2 != 1: b2 = b2 * b2 % r; i += 1
        e = pow(c, 1 << (m - i - 1), r)
        x = x * e % r; c = e * e % r; b = b * c % r; m = i
    return x


def bech32m_encode(hrp: str, data: bytes) -> str:
    acc = bits = 0; d = []
    for byte in data:
        acc = (acc << 8) | byte; bits += 8
        while bits >= 5: bits -= 5; d.append((acc >> bits) & 31)
    if bits: d.append((acc << (5 - bits)) & 31)
    exp = [ord(c) >> 5 for c in hrp] + [0] + [ord(c) & 31 for c in hrp]
    pm = _polymod(exp + d + [0] * 6) ^ 0x2BC830A3
    d += [(pm >> (5 * (5 - i))) & 31 for i in range(6)]
    return hrp + '1' + ''.join(_B32[v] for v in d)


def g1_decompress(buf: bytes):
    assert len(buf) == 48
    flags = buf[47]
    if flags & 0x40: return None
    x = int.from_bytes(buf[:47] + bytes([flags & 0x3F]), 'little')
    assert x < Q
    y = fq_sqrt(x * x * x + G1_COEFF_B)
    if y is None: raise ValueError('x not on curve')
    larger = max(y, Q - y); smaller = min(y, Q - y)
    return (x, larger if (flags & 0x80) else smaller)


def g1_compress(P) -> bytes:
    if P is None:
        return bytes(47) + bytes([0x40])
    x, y = P
    b = bytearray(x.to_bytes(48, 'little'))
    if y > (Q - y) % Q: b[47] |= 0x80
    return bytes(b)


# ----------------------------------------------------------------------------------------------
# G2 over Fq2 = Fq[u]/(u^2 + 5)  (snarkvm-curves bls12_377/{fq2,g2}.rs [UPSTREAM-RECALL]; constants checked in tests/test_oracle.py:
# the generator lies on y^2 = x^3 + G2_COEFF_B and r * G2_GENERATOR = O).  Elements are (c0, c1) tuples of canonical ints.
# ----------------------------------------------------------------------------------------------
FQ2_NONRESIDUE = Q - 5
G2_COEFF_B = (0, 155198655607781456406391640216936120121836107652948796323930557600032281009004493664981332883744016074664192874906)
G2_GENERATOR = (
    (233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294,
     140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118),
    (63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423,
     149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491),
)


def fq2_add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def fq2_sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def fq2_mul(a, b): return ((a[0] * b[0] + FQ2_NONRESIDUE * a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
def fq2_inv(a):
    n = pow((a[0] * a[0] - FQ2_NONRESIDUE * a[1] * a[1]) % Q, -1, Q)
    return (a[0] * n % Q, (-a[1] * n) % Q)


def g2_is_on_curve(P) -> bool:
    if P is None: return True
    x, y = P
    return fq2_mul(y, y) == fq2_add(fq2_mul(fq2_mul(x, x), x), G2_COEFF_B)


def g2_neg(P): return None if P is None else (P[0], ((-P[1][0]) % Q, (-P[1][1]) % Q))


def g2_add(P, S):
    if P is None: return S
    if S is None: return P
    if P[0] == S[0]:
        if P[1] != S[1] or P[1] == (0, 0): return None
        lam = fq2_mul(fq2_mul((3, 0), fq2_mul(P[0], P[0])), fq2_inv(fq2_mul((2, 0), P[1])))
    else:
        lam = fq2_mul(fq2_sub(S[1], P[1]), fq2_inv(fq2_sub(S[0], P[0])))
    x = fq2_sub(fq2_sub(fq2_mul(lam, lam), P[0]), S[0])
    return (x, fq2_sub(fq2_mul(lam, fq2_sub(P[0], x)), P[1]))


def g2_mul(P, k: int):
    acc = None
    for bit in bin(k % FR_MODULUS)[2:] if k % FR_MODULUS else '':
        acc = g2_add(acc, acc)
        if bit == '1': acc = g2_add(acc, P)
    return acc


def g2_mul_raw(P, k: int):
    """k * P without reducing k mod r (subgroup check)."""
    acc = None
    for bit in bin(k)[2:]:
        acc = g2_add(acc, acc)
        if bit == '1': acc = g2_add(acc, P)
    return acc


def msm_naive_g2(bases, scalars):
    acc = None
    for P, s in zip(bases, scalars):
        if P is not None and s % FR_MODULUS: acc = g2_add(acc, g2_mul(P, s))
    return acc


# ----------------------------------------------------------------------------------------------
# bech32m (BIP-350) decode, for the `proof1…` fixture
# ----------------------------------------------------------------------------------------------
_B32 = 'qpzry9x8gf2tvdw0s3jn54khce6mua7l'


def _polymod(values):
    gen = [0x3B6A57B2, 0x26508E6D, 0x1EA119FA, 0x3D4233DD, 0x2A1462B3]
    chk = 1
    for v in values:
        b = chk >> 25
        chk = ((chk & 0x1FFFFFF) << 5) ^ v
        for i in range(5):
            chk ^= gen[i] if ((b >> i) & 1) else 0
    return chk


def bech32m_decode(s: str):
    pos = s.rfind('1')
    hrp, data = s[:pos], [_B32.index(c) for c in s[pos + 1:]]
    exp = [ord(c) >> 5 for c in hrp] + [0] + [ord(c) & 31 for c in hrp]
    assert _polymod(exp + data) == 0x2BC830A3, 'bad bech32m checksum'
    data = data[:-6]
    acc = bits = 0; out = bytearray()
    for v in data:
        acc = (acc << 5) | v; bits += 5
        while bits >= 8:
            bits -= 8; out.append((acc >> bits) & 0xFF)
    return hrp, bytes(out)


# ----------------------------------------------------------------------------------------------
# EvaluationDomain (algorithms/src/fft/domain.rs)
# ----------------------------------------------------------------------------------------------
R_ = FR_MODULUS


class EvaluationDomain:
    """size = 2^k >= num_coeffs; group_gen = TWO_ADIC_ROOT^(2^(47-k)); generator_inv = 22^-1."""

    def __init__(self, num_coeffs: int):
        size = 1
        while size < num_coeffs: size *= 2
        self.size = size
        self.log_size_of_group = size.bit_length() - 1
        assert self.log_size_of_group <= FR_TWO_ADICITY
        self.group_gen = pow(FR_TWO_ADIC_ROOT, 1 << (FR_TWO_ADICITY - self.log_size_of_group), R_)
        self.group_gen_inv = pow(self.group_gen, -1, R_)
        self.size_inv = pow(size, -1, R_)
        self.generator_inv = pow(FR_GENERATOR, -1, R_)

    def _dft(self, x, w):
        n = self.size
        x = list(x) + [0] * (n - len(x))
        # O(n^2) definition: out[i] = sum_j x[j] w^(ij); only for n <= 2^10 in the fixtures.
        pw = [1] * n
        for i in range(1, n): pw[i] = pw[i - 1] * w % R_
        return [sum(x[j] * pw[(i * j) % n] for j in range(n)) % R_ for i in range(n)]

    def fft(self, x): return self._dft(x, self.group_gen)

    def ifft(self, x): return [v * self.size_inv % R_ for v in self._dft(x, self.group_gen_inv)]

    def coset_fft(self, x):
        g, acc, y = FR_GENERATOR, 1, []
        for v in list(x) + [0] * (self.size - len(x)):
            y.append(v * acc % R_); acc = acc * g % R_
        return self.fft(y)

    def coset_ifft(self, x):
        y, acc, out = self.ifft(x), 1, []
        for v in y:
            out.append(v * acc % R_); acc = acc * self.generator_inv % R_
        return out


def fft_fast(x, w, mod=FR_MODULUS):
    """Recursive radix-2 DFT with natural ordering (same function as _dft, O(n log n))."""
    n = len(x)
    if n == 1: return list(x)
    w2 = w * w % mod
    ev, od = fft_fast(x[0::2], w2, mod), fft_fast(x[1::2], w2, mod)
    out = [0] * n; t = 1
    for i in range(n // 2):
        u = od[i] * t % mod
        out[i] = (ev[i] + u) % mod; out[i + n // 2] = (ev[i] - u) % mod
        t = t * w % mod
    return out


def bit_reverse_permute(x):
    n = len(x); k = n.bit_length() - 1
    return [x[int(format(i, '0%db' % k)[::-1], 2) if k else 0] for i in range(n)]


# ----------------------------------------------------------------------------------------------
# SplitMix64 — the deterministic input generator shared by oracle, tests and bench
# ----------------------------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed: int): self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self) -> int:
        """Uniform canonical Fr by rejection sampling on the low 253 bits of 4 limbs."""
        while True:
            v = 0
            for i in range(4): v |= self.next() << (64 * i)
            v &= (1 << 253) - 1
            if v < FR_MODULUS: return v

"""snarkVM's Poseidon in plain Python integers: parameter generation (Grain LFSR), the permutation, the console-side
hash (`hash_psd2/4/8`, `hash_many_psd8`) over Fr and the duplex sponge the Fiat-Shamir transcript of the prover runs over Fq.

TEST INFRASTRUCTURE ONLY (DESIGN.md "Oracle"): imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

The code restated lives in crates.io snarkVM 0.14.5, absent from /root/reference [UPSTREAM-RECALL]:
  fields/src/traits/poseidon_grain_lfsr.rs   PoseidonGrainLFSR (80-bit Grain LFSR in self-shrinking mode)
  fields/src/traits/poseidon_default.rs      default_poseidon_parameters::<RATE>(): ark by rejection sampling, then a Cauchy MDS
  curves/src/bls12_377/{fr,fq}.rs            PARAMS_OPT_FOR_CONSTRAINTS: (rate, alpha 17, 8 full rounds, 31 partial rounds, 0 skipped matrices)
  console/algorithms/src/poseidon/           Poseidon<E, RATE>: preimage [domain, len, 0.., input], capacity 1, state = [capacity | rate]
  console/program/src/data/ciphertext/decrypt.rs, plaintext/{from_fields,from_bits}.rs   symmetric decryption and the plaintext bit layout

PINNED by reference-held data over Fr (tests/test_poseidon.py, fixture tests/golden/reference_account.json):
  * rates 2 and 8: the reference's test at /root/reference/wasm/src/account/private_key_ciphertext.rs:115-121 holds a private key, the secret
    "mypassword" and a ciphertext; decrypting the ciphertext the way /root/reference/rust/src/account/encryptor.rs:60-67 does (randomizers =
    hash_many_psd8, blinding = hash_psd2) gives that key, and the 680 plaintext bits in between parse as the struct {key: field, nonce: field};
  * rates 2 and 4: the (private key, view key, address) triples at /root/reference/wasm/src/account/private_key.rs:182-184 and
    /root/reference/sdk/tests/data/account-data.ts:8-19 — view key = sk_sig + r_sig + sk_prf with hash_to_scalar_psd2 / psd4, address = view key * G.
A wrong round count, LFSR tap, MDS convention, state order or preimage layout fails those tests.  UNPINNED: everything over Fq (same
generator, field size 377; 8 + 31 rounds recalled; no reference-held value exists for the prover's transcript).
"""
from __future__ import annotations
from functools import lru_cache

from . import pyref as P

ALPHA = 17
FULL_ROUNDS, PARTIAL_ROUNDS = 8, 31          # curves/src/bls12_377/{fr,fq}.rs PARAMS_OPT_FOR_CONSTRAINTS, every rate 2..8
CAPACITY = 1


class GrainLFSR:
    """fields/src/traits/poseidon_grain_lfsr.rs: 80 state bits = [field type 0b01 | s-box 4 bits | field bits (12) | state width (12) |
    full rounds (10) | partial rounds (10) | thirty ones], 160 warm-up updates, output by pairs (first bit 1 -> emit the second)."""

    def __init__(self, is_sbox_inverse: bool, field_bits: int, width: int, full_rounds: int, partial_rounds: int):
        st = [False] * 80
        st[1] = True                                     # prime field
        st[5] = bool(is_sbox_inverse)                    # s-box x^alpha (0) / x^-1 (1) in the last of its four bits
        def put(lo, hi, v):
            for i in range(hi, lo - 1, -1):
                st[i] = bool(v & 1); v >>= 1
        put(6, 17, field_bits); put(18, 29, width); put(30, 39, full_rounds); put(40, 49, partial_rounds)
        for i in range(50, 80): st[i] = True
        self.st, self.head, self.field_bits = st, 0, field_bits
        for _ in range(160): self._update()

    def _update(self) -> bool:
        s, h = self.st, self.head
        b = s[(h + 62) % 80] ^ s[(h + 51) % 80] ^ s[(h + 38) % 80] ^ s[(h + 23) % 80] ^ s[(h + 13) % 80] ^ s[h]
        s[h] = b; self.head = (h + 1) % 80
        return b

    def bits(self, n: int):
        out = []
        for _ in range(n):
            b = self._update()
            while not b:
                self._update(); b = self._update()
            out.append(self._update())
        return out

    def _int(self) -> int:
        v = 0
        for b in self.bits(self.field_bits):             # most significant bit first
            v = (v << 1) | int(b)
        return v

    def elements_rejection(self, n: int, mod: int):
        out = []
        while len(out) < n:
            v = self._int()
            if v < mod: out.append(v)
        return out

    def elements_mod_p(self, n: int, mod: int):
        return [self._int() % mod for _ in range(n)]


@lru_cache(maxsize=None)
def parameters(mod: int, rate: int):
    """(ark rounds x width, mds width x width) as canonical integers — default_poseidon_parameters::<RATE>()."""
    width = rate + CAPACITY
    lfsr = GrainLFSR(False, mod.bit_length(), width, FULL_ROUNDS, PARTIAL_ROUNDS)
    ark = [lfsr.elements_rejection(width, mod) for _ in range(FULL_ROUNDS + PARTIAL_ROUNDS)]
    xs = lfsr.elements_mod_p(width, mod); ys = lfsr.elements_mod_p(width, mod)
    mds = [[pow((x + y) % mod, -1, mod) for y in ys] for x in xs]
    return ark, mds


def permute(state, mod: int, rate: int):
    ark, mds = parameters(mod, rate)
    half = FULL_ROUNDS // 2
    w = rate + CAPACITY
    for i in range(FULL_ROUNDS + PARTIAL_ROUNDS):
        state = [(s + a) % mod for s, a in zip(state, ark[i])]
        if half <= i < half + PARTIAL_ROUNDS:
            state[0] = pow(state[0], ALPHA, mod)
        else:
            state = [pow(s, ALPHA, mod) for s in state]
        state = [sum(state[j] * mds[r][j] for j in range(w)) % mod for r in range(w)]
    return state


class Sponge:
    """Duplex sponge, state[0] = capacity, state[1..] = rate (console/algorithms/src/poseidon/helpers/sponge.rs; the same construction
    as algorithms/src/crypto_hash/poseidon.rs PoseidonSponge over Fq)."""

    def __init__(self, mod: int, rate: int):
        self.mod, self.rate = mod, rate
        self.state = [0] * (rate + CAPACITY)
        self.absorbing, self.pos = True, 0
        self.permutations = 0

    def _permute(self):
        self.state = permute(self.state, self.mod, self.rate); self.permutations += 1

    def absorb(self, elems):
        elems = [e % self.mod for e in elems]
        if not elems: return
        if not self.absorbing:
            self._permute(); self.pos = 0
        elif self.pos == self.rate:
            self._permute(); self.pos = 0
        self.absorbing = True
        i = 0
        while True:
            take = min(self.rate - self.pos, len(elems) - i)
            for k in range(take):
                self.state[CAPACITY + self.pos + k] = (self.state[CAPACITY + self.pos + k] + elems[i + k]) % self.mod
            i += take; self.pos += take
            if i == len(elems): return
            self._permute(); self.pos = 0

    def squeeze(self, n: int):
        out = []
        if n == 0: return out
        if self.absorbing:
            self._permute(); self.pos = 0; self.absorbing = False
        elif self.pos == self.rate:
            self._permute(); self.pos = 0
        while True:
            take = min(self.rate - self.pos, n - len(out))
            out += self.state[CAPACITY + self.pos: CAPACITY + self.pos + take]
            self.pos += take
            if len(out) == n: return out
            self._permute(); self.pos = 0


# ----------------------------------------------------------------------------------------------
# Console side (Fr): Network::hash_psd{2,4,8}, hash_many_psd8, domain separators
# ----------------------------------------------------------------------------------------------
R = P.FR_MODULUS


def domain_separator(s: str, mod: int = R) -> int:
    """Field::new_domain_separator = from_bytes_le_mod_order(domain.as_bytes())."""
    return int.from_bytes(s.encode(), 'little') % mod


def hash_many(rate: int, inputs, n_out: int):
    """Poseidon<E, RATE>::hash_many: preimage [domain "AleoPoseidon{RATE}", len(input), 0 … up to RATE, input…]."""
    pre = [domain_separator('AleoPoseidon%d' % rate), len(inputs)] + [0] * (rate - 2) + list(inputs)
    sp = Sponge(R, rate)
    sp.absorb(pre)
    return sp.squeeze(n_out)


def hash_psd2(inputs): return hash_many(2, inputs, 1)[0]
def hash_psd4(inputs): return hash_many(4, inputs, 1)[0]
def hash_psd8(inputs): return hash_many(8, inputs, 1)[0]
def hash_many_psd8(inputs, n): return hash_many(8, inputs, n)


# ----------------------------------------------------------------------------------------------
# Ciphertext / Plaintext / PrivateKey wire formats — just enough to replay the reference's known answer
# ----------------------------------------------------------------------------------------------
FR_DATA_BITS = 252                                                  # Field::size_in_data_bits
PRIVATE_KEY_PREFIX = bytes([127, 134, 189, 116, 210, 221, 210, 137, 145, 18, 253])   # "APrivateKey1"
_B58 = '123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz'


def base58_decode(s: str) -> bytes:
    v = 0
    for c in s: v = v * 58 + _B58.index(c)
    body = v.to_bytes((v.bit_length() + 7) // 8, 'big')
    return bytes(len(s) - len(s.lstrip('1'))) + body


def base58_encode(b: bytes) -> str:
    v = int.from_bytes(b, 'big'); out = ''
    while v: v, d = divmod(v, 58); out = _B58[d] + out
    return '1' * (len(b) - len(b.lstrip(b'\0'))) + out


def private_key_seed(s: str) -> int:
    raw = base58_decode(s)
    assert raw[:11] == PRIVATE_KEY_PREFIX and len(raw) == 43, 'not an Aleo private key'
    seed = int.from_bytes(raw[11:], 'little')
    assert seed < R
    return seed


def private_key_string(seed: int) -> str:
    return base58_encode(PRIVATE_KEY_PREFIX + seed.to_bytes(32, 'little'))


def ciphertext_fields(s: str):
    """Ciphertext = bech32m('ciphertext', u16 LE count | count x 32-byte LE field elements)."""
    hrp, data = P.bech32m_decode(s)
    assert hrp == 'ciphertext'
    n = int.from_bytes(data[:2], 'little')
    assert len(data) >= 2 + 32 * n
    f = [int.from_bytes(data[2 + 32 * i: 34 + 32 * i], 'little') for i in range(n)]
    assert all(v < R for v in f)
    return f


def _bits_to_int(bits): return sum(int(b) << i for i, b in enumerate(bits))


def plaintext_from_fields(fields):
    """Plaintext::from_fields + from_bits_le: 252 data bits per field, strip the zero padding and the terminus 1, then parse
    variant (2 bits) | literal: type u8, size u16, value | struct: count u8, then per member name size u8, name, size u16, value."""
    bits = []
    for f in fields: bits += [(f >> i) & 1 for i in range(FR_DATA_BITS)]
    while bits and not bits[-1]: bits.pop()
    assert bits, 'no terminus bit'
    bits.pop()

    def parse(b):
        variant = (b[0], b[1]); pos = 2
        if variant == (0, 0):
            ty = _bits_to_int(b[pos:pos + 8]); size = _bits_to_int(b[pos + 8:pos + 24]); pos += 24
            assert pos + size == len(b), 'literal size'
            return ('literal', ty, _bits_to_int(b[pos:pos + size]), size)
        assert variant == (0, 1), 'plaintext variant'
        n = _bits_to_int(b[pos:pos + 8]); pos += 8; members = {}
        for _ in range(n):
            ln = _bits_to_int(b[pos:pos + 8]); pos += 8
            assert ln % 8 == 0
            name = _bits_to_int(b[pos:pos + ln]).to_bytes(ln // 8, 'little').decode(); pos += ln
            size = _bits_to_int(b[pos:pos + 16]); pos += 16
            members[name] = parse(b[pos:pos + size]); pos += size
        assert pos == len(b), 'struct size'
        return ('struct', members)
    return parse(bits)


def decrypt_symmetric(ciphertext: str, view_key: int, hasher=None):
    """Ciphertext::decrypt_symmetric: randomizers = hash_many_psd8([encryption domain, key], #fields); plaintext_i = c_i − randomizer_i.
    hasher(rate, inputs, n_out): another implementation of Poseidon::hash_many to put through the same known answer (default: this module's)."""
    c = ciphertext_fields(ciphertext)
    rnd = (hasher or hash_many)(8, [domain_separator('AleoSymmetricEncryption0'), view_key], len(c))
    return plaintext_from_fields([(a - b) % R for a, b in zip(c, rnd)])


LITERAL_FIELD = 2                                                   # Literal variants: address 0, boolean 1, field 2, group 3, …


def decrypt_private_key(ciphertext: str, secret: str, hasher=None) -> str:
    """Encryptor::decrypt_private_key_with_secret (/root/reference/rust/src/account/encryptor.rs:31-67)."""
    dom, sec = domain_separator('private_key'), domain_separator(secret)
    kind, members = decrypt_symmetric(ciphertext, sec, hasher)
    assert kind == 'struct' and list(members) == ['key', 'nonce']
    for m in members.values():
        assert m[0] == 'literal' and m[1] == LITERAL_FIELD and m[3] == 253 and m[2] < R
    blinding = (hasher or hash_many)(2, [dom, members['nonce'][2], sec], 1)[0]
    return private_key_string(members['key'][2] * pow(blinding, -1, R) % R)


# ----------------------------------------------------------------------------------------------
# Account derivation (console/account/src/{private_key,compute_key,view_key,address}): a second reference-held known answer, which
# pins rate 4 (hash_to_scalar_psd4) and hash_to_scalar's truncation.  Edwards-BLS12 over Fr: -x^2 + y^2 = 1 + 3021 x^2 y^2.
# ----------------------------------------------------------------------------------------------
ED_D = 3021
ED_SUBGROUP_ORDER = 2111115437357092606062206234695386632838870926408408195193685246394721360383      # cofactor 4
SCALAR_DATA_BITS = 250                                                                                 # Scalar::size_in_data_bits
VIEW_KEY_PREFIX = bytes([14, 138, 223, 204, 247, 224, 122])                                            # "AViewKey1"


def ed_add(p, q):
    x1, y1 = p; x2, y2 = q
    k = ED_D * x1 * x2 * y1 * y2 % R
    return ((x1 * y2 + y1 * x2) * pow(1 + k, -1, R) % R, (y1 * y2 + x1 * x2) * pow(1 - k, -1, R) % R)


def ed_mul(p, k: int):
    acc = (0, 1)
    for b in bin(k)[2:]:
        acc = ed_add(acc, acc)
        if b == '1': acc = ed_add(acc, p)
    return acc


def ed_from_x(x: int):
    """Group::from_x_coordinate: the point with this x in the prime-order subgroup."""
    y = P.fr_sqrt((1 + x * x) * pow((1 - ED_D * x * x) % R, -1, R) % R)
    if y is None: raise ValueError('x is not on the curve')
    for yy in (y, R - y):
        if ed_mul((x, yy), ED_SUBGROUP_ORDER) == (0, 1): return (x, yy)
    raise ValueError('no point of prime order with this x')


def view_key_scalar(s: str) -> int:
    raw = base58_decode(s)
    assert raw[:7] == VIEW_KEY_PREFIX and len(raw) == 39, 'not an Aleo view key'
    return int.from_bytes(raw[7:], 'little')


def view_key_string(v: int) -> str: return base58_encode(VIEW_KEY_PREFIX + v.to_bytes(32, 'little'))


def address_point(s: str):
    hrp, raw = P.bech32m_decode(s)
    assert hrp == 'aleo' and len(raw) == 32
    return ed_from_x(int.from_bytes(raw, 'little'))


def address_string(pt) -> str: return P.bech32m_encode('aleo', pt[0].to_bytes(32, 'little'))


def hash_to_scalar(rate: int, inputs, hasher=None) -> int:
    """Poseidon::hash_to_scalar: the low Scalar::size_in_data_bits bits of the hash."""
    return (hasher or hash_many)(rate, inputs, 1)[0] & ((1 << SCALAR_DATA_BITS) - 1)


def derive_account(private_key: str, generator, hasher=None):
    """PrivateKey::try_from(seed) -> ComputeKey -> ViewKey -> Address.  `generator` is the account generator G (upstream derives it
    with Blake2Xs hash-to-curve, not restated: tests recover it from one reference-held (view key, address) pair as view_key^-1 * address
    and check the others against it)."""
    seed = private_key_seed(private_key)
    sk_sig = hash_to_scalar(2, [domain_separator('AleoAccountSignatureSecretKey0'), seed], hasher)
    r_sig = hash_to_scalar(2, [domain_separator('AleoAccountSignatureRandomizer0.0'), seed], hasher)
    pk_sig, pr_sig = ed_mul(generator, sk_sig), ed_mul(generator, r_sig)
    sk_prf = hash_to_scalar(4, [pk_sig[0], pr_sig[0]], hasher)
    view = (sk_sig + r_sig + sk_prf) % ED_SUBGROUP_ORDER
    return view_key_string(view), address_string(ed_mul(generator, view))


# ----------------------------------------------------------------------------------------------
# The prover's Fiat-Shamir sponge: PoseidonSponge<Fq, 2, 1> behind the AlgebraicSponge interface
# (algorithms/src/crypto_hash/poseidon.rs, algorithms/src/traits/algebraic_sponge.rs, utilities nonnative params [UPSTREAM-RECALL];
#  UNPINNED — the reference holds no transcript value).  Fr elements enter as 5 limbs of 51 bits ("weight-optimised" parameters for a
#  253-bit field inside a 377-bit one), two limbs packed per Fq element; challenges leave as 252-bit (full) or 168-bit (short) integers
#  cut from the low 376 bits of squeezed Fq elements, most significant bit first.
# ----------------------------------------------------------------------------------------------
Q = P.FQ_MODULUS
FS_RATE = 2
FQ_CAPACITY_BITS = P.FQ_BITS - 1          # 376
SHORT_CHALLENGE_BITS = 168
FULL_CHALLENGE_BITS = P.FR_BITS - 1       # 252


def nonnative_parameters(base_bits: int = P.FQ_BITS, target_bits: int = P.FR_BITS):
    """find_parameters(base, target, OptimizationType::Weight) -> (num_limbs, bits_per_limb): the limb size of least 'weight'."""
    surfeit, best = 10, None
    max_limb = min((base_bits - 1 - surfeit - 1) // 2 - 1, target_bits)
    for limb in range(1, max_limb + 1):
        nl = (target_bits + limb - 1) // limb
        group = (base_bits - 1 - surfeit - 1 - 1 - limb + limb - 1) // limb
        ng = (2 * nl - 1 + group - 1) // group
        cost = 6 * nl * nl + 2 * (4 * target_bits) + nl + nl * nl + 2 * (2 * nl - 1) + nl + ng + 6 * ng + (ng - 1) * (2 * limb + surfeit) * 4 + 2
        if best is None or cost < best[0]: best = (cost, nl, limb)
    return best[1], best[2]


NN_LIMBS, NN_LIMB_BITS = nonnative_parameters()          # (5, 51)


class FiatShamir:
    def __init__(self): self.sp = Sponge(Q, FS_RATE)

    def absorb_native(self, elems): self.sp.absorb(list(elems))

    def absorb_points(self, pts):
        """Affine G1 points as (x, y) pairs (ToConstraintField of a short-Weierstrass affine point); infinity = (0, 1)."""
        flat = []
        for p in pts: flat += [0, 1] if p is None else [p[0], p[1]]
        self.absorb_native(flat)

    def absorb_bytes(self, data: bytes):
        """absorb_bytes: the bits of every byte most significant first, cut into chunks of 376 bits, each read as a big-endian integer."""
        bits = ''.join(format(b, '08b') for b in data)
        self.absorb_native([int(bits[i:i + FQ_CAPACITY_BITS], 2) for i in range(0, len(bits), FQ_CAPACITY_BITS)])

    def absorb_nonnative(self, frs):
        """push_elements_to_sponge(…, Weight): limbs most significant first, each tagged with one addition; compress_elements packs two
        neighbouring limbs into one Fq element as first * 2^(bits of the second) + second when both fit 376 bits."""
        limbs = []
        for v in frs:
            v %= R
            limbs += [(v >> (NN_LIMB_BITS * i)) & ((1 << NN_LIMB_BITS) - 1) for i in reversed(range(NN_LIMBS))]
        per = NN_LIMB_BITS + 2                                 # bits_per_limb + overhead(1 + 1) = 53
        out, i = [], 0
        while i < len(limbs):
            if i + 1 < len(limbs) and 2 * per <= FQ_CAPACITY_BITS:
                out.append((limbs[i] << per) + limbs[i + 1]); i += 2
            else:
                out.append(limbs[i]); i += 1
        self.absorb_native(out)

    def _bits(self, n: int) -> str:
        """get_bits: squeeze ⌈n / 376⌉ elements, keep the low 376 bits of each (most significant first), truncate."""
        cnt = (n + FQ_CAPACITY_BITS - 1) // FQ_CAPACITY_BITS
        s = ''.join(format(e & ((1 << FQ_CAPACITY_BITS) - 1), '0%db' % FQ_CAPACITY_BITS) for e in self.sp.squeeze(cnt))
        return s[:n]

    def _fe(self, n: int, width: int):
        if n == 0: return []
        s = self._bits(n * width)
        return [int(s[i * width:(i + 1) * width], 2) % R for i in range(n)]

    def squeeze_nonnative(self, n: int): return self._fe(n, FULL_CHALLENGE_BITS)
    def squeeze_short(self, n: int = 1): return self._fe(n, SHORT_CHALLENGE_BITS)
    def squeeze_short_one(self) -> int: return self._fe(1, SHORT_CHALLENGE_BITS)[0]

"""Static bound check of the 29-bit-limb butterflies (aleo_amd/csrc/fr29.h, ntt.hip dif_group29): replays the operation sequence of a register group on
BOUNDS instead of values — the largest limb, the largest top limb and the largest value (in multiples of r) a register can hold — and asserts every
precondition the kernels rely on: no 32-bit limb wraps, the padded constant covers every subtrahend limb, every product column stays below 2^64, every
value stays below 2^261, and what a group stores is again what a group may load (normalised, below 4.5 r).  Random inputs never reach these bounds;
this test is what says the kernels are safe for ALL inputs.  CPU only."""
import os, re
from fractions import Fraction as Fr

R = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001
B, N = 29, 9
M = (1 << B) - 1
HERE = os.path.dirname(os.path.abspath(__file__))
GEN = open(os.path.join(HERE, '..', 'aleo_amd', 'csrc', 'fr29_mont_gen.h')).read()


def _arr(name):
    m = re.search(r'%s\[9\] = \{([^}]*)\}' % name, GEN)
    return [int(x.strip().rstrip('u'), 16) for x in m.group(1).split(',')]


PAD, P, ONE = _arr('FR29_PAD'), _arr('FR29_P'), _arr('FR29_ONE')
QMAGIC = int(re.search(r'FR29_QMAGIC = (0x[0-9a-f]+)u', GEN).group(1), 16)
val = lambda limbs: sum(x << (B * i) for i, x in enumerate(limbs))
top_of = lambda v_in_r: int(v_in_r * R) >> (B * (N - 1))          # top limb of a normalised number below v * r


class Reg:
    """bounds of one register: limbs 0..7 <= L, top limb <= T, value < V * r"""
    def __init__(self, L, T, V): self.L, self.T, self.V = L, T, Fr(V)


def normalised(V): return Reg(M, top_of(V) + 1, V)


def add(a, b):
    r = Reg(a.L + b.L, a.T + b.T, a.V + b.V)
    assert r.L < 1 << 32 and r.T < 1 << 32, 'a lazy sum wraps a 32-bit limb'
    return r


def sub_pad(u, x):
    assert x.L <= min(PAD[:8]), 'a subtrahend limb can exceed the padded constant: %x > %x' % (x.L, min(PAD[:8]))
    assert x.T <= PAD[8], 'the subtrahend top limb can exceed the padded constant'
    r = Reg(u.L + max(PAD[:8]), u.T + PAD[8], u.V + Fr(val(PAD), R))
    assert r.L < 1 << 32 and r.T < 1 << 32
    return r


def mul(a, tw_V=1):
    """a * (normalised table entry below tw_V * r): column k holds at most 9 products a_i b_j, 9 products m_i p_j and a carry"""
    amax = max(a.L, a.T)
    col = N * amax * M + N * M * M
    assert col + (col >> B) < 1 << 64, 'a product column can overflow 64 bits'
    assert a.V * R < 1 << (B * N), 'a multiplicand can exceed 2^261'
    return normalised(a.V * tw_V * Fr(R, 1 << (B * N)) + 1)


def normalise(a):
    assert a.V * R < 1 << (B * N)
    return normalised(a.V)


def reduce_partial(a):
    assert a.L == M, 'reduce_partial wants a normalised value'
    assert a.T * QMAGIC < 1 << 64
    return normalised(min(a.V, Fr(3)))


def group(G, V0, last_stage_plain):
    """dif_group29<G>: registers enter normalised below V0 * r; returns the bounds of what it stores"""
    K = 1 << G
    v = [normalised(V0) for _ in range(K)]
    for t in range(G):
        d = 1 << (G - 1 - t)
        if G == 3 and t == 2: v[0], v[1] = normalise(v[0]), normalise(v[1])
        for r_ in range(d):
            for blk in range(K // (2 * d)):
                lo = blk * 2 * d + r_; hi = lo + d
                u, x = v[lo], v[hi]
                v[lo] = add(u, x)
                dif = sub_pad(u, x)
                v[hi] = reduce_partial(normalise(dif)) if (last_stage_plain and t == G - 1) else mul(dif)
    for j in range(0, K, 2): v[j] = normalise(v[j])
    v[0] = reduce_partial(v[0])
    return v


def test_constants():
    assert val(P) == R and val(ONE) == (1 << (B * N)) % R
    assert val(PAD) % R == 0 and val(PAD) // R == 19
    d = (R >> (B * (N - 1))) + 1
    assert QMAGIC == (1 << 32) // d
    # q = mulhi(top, QMAGIC) never exceeds floor(value / r) and leaves less than 3 r, for every normalised value below 2^261 (checked on the extremes of each quotient)
    for q_true in list(range(0, 446, 7)) + [1, 2, 444, 445]:
        for v in (q_true * R, q_true * R + R - 1):
            if v >> (B * N): continue
            q = ((v >> (B * (N - 1))) * QMAGIC) >> 32
            assert q <= q_true and v - q * R < 3 * R


def test_group_bounds_close():
    V0 = Fr(9, 2)
    for G in (3, 2, 1):
        for plain in (False, True):
            out = group(G, V0, plain)
            assert all(r.L == M for r in out), 'a group stores an unnormalised register'
            assert max(r.V for r in out) <= V0, (G, plain, float(max(r.V for r in out)))


def test_kernel_edges():
    V0 = Fr(9, 2)
    mul(normalised(2), 1)                      # load: a value from HBM (< 2 r) times a coset power
    x = mul(normalised(V0), Fr(101, 100))      # store: a tile value times the inter-pass factor (a product of two table entries, < 1.01 r)
    assert x.V < Fr(11, 10) and x.V * R < 1 << 256          # fits the 32-byte word form, one conditional subtraction makes it canonical
    y = reduce_partial(normalised(V0))         # plain forward transform: no last product
    assert y.V <= 3 and y.V * R < 1 << 256     # cond_sub<2>, cond_sub<1> finish it

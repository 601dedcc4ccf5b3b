"""TEST INFRASTRUCTURE — the BLS12-377 ate pairing in plain Python integers, so that oracle/varuna_ref.py can check KZG openings the way a
verifier does (pairing equations over public G2 elements) instead of with the trapdoor of the synthetic setup.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Construction (public definitions; constants re-derived and checked in tests/test_oracle.py):
  * tower Fq2 = Fq[u]/(u^2 + 5), Fq6 = Fq2[v]/(v^3 - u), Fq12 = Fq6[w]/(w^2 - v)  [snarkvm-curves bls12_377/{fq2,fq6,fq12}.rs, UPSTREAM-RECALL]
    collapses to ONE extension Fq12 = Fq[w]/(w^12 + 5) with u = w^6: elements are 12 coefficients, multiplication is schoolbook with
    w^12 = -5 — slow and short;
  * G2 is the D-type sextic twist E': y^2 = x^3 + 1/u over Fq2 (pyref.G2_COEFF_B = (0, -1/5) = 1/u); psi(x', y') = (x' w^2, y' w^3) maps it onto
    E: y^2 = x^3 + 1 over Fq12;
  * e(P, Q) = f_{x,psi(Q)}(P)^((q^12 - 1)/r) with the loop over the bits of x = 0x8508c00000000001 (r = x^4 - x^2 + 1): the ate pairing; the line
    through T with twist-side slope lam evaluated at P = (xP, yP) is  yP - lam xP * w + (lam xT - yT) * w^3  (vertical lines vanish in the
    final exponentiation).
Pinned by: bilinearity e(aP, bQ) = e(P, Q)^(ab), non-degeneracy, e(P, Q)^r = 1 (tests/test_oracle.py)."""
from __future__ import annotations
from . import pyref as P

Q = P.FQ_MODULUS
R = P.FR_MODULUS
X_PARAM = 0x8508C00000000001
ONE12 = [1] + [0] * 11


def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                if y: t[i + j] += x * y
    return [(t[i] - 5 * t[i + 12]) % Q if i < 11 else t[i] % Q for i in range(12)]       # w^12 = -5


def f12_pow(a, e):
    out = ONE12; base = a
    while e:
        if e & 1: out = f12_mul(out, base)
        base = f12_mul(base, base); e >>= 1
    return out


def _embed(c, shift):
    """(c0 + c1 u) * w^shift as an Fq12 element (shift + 6 < 12)."""
    o = [0] * 12; o[shift] = c[0] % Q; o[shift + 6] = c[1] % Q
    return o


def _line(lam, T, Pt):
    xP, yP = Pt
    l1 = _embed(P.fq2_mul(lam, ((-xP) % Q, 0)), 1)
    l3 = _embed(P.fq2_sub(P.fq2_mul(lam, T[0]), T[1]), 3)
    o = [(a + b) % Q for a, b in zip(l1, l3)]; o[0] = yP % Q
    return o


def miller_loop(Pt, Qt):
    """f_{x, psi(Qt)}(Pt): Pt affine on E(Fq) (ints), Qt affine on the twist E'(Fq2); None for the point at infinity gives 1."""
    if Pt is None or Qt is None: return ONE12
    f = ONE12; T = Qt
    for bit in bin(X_PARAM)[3:]:
        lam = P.fq2_mul(P.fq2_mul((3, 0), P.fq2_mul(T[0], T[0])), P.fq2_inv(P.fq2_add(T[1], T[1])))          # tangent slope (a = 0)
        f = f12_mul(f12_mul(f, f), _line(lam, T, Pt))
        x3 = P.fq2_sub(P.fq2_mul(lam, lam), P.fq2_add(T[0], T[0])); T = (x3, P.fq2_sub(P.fq2_mul(lam, P.fq2_sub(T[0], x3)), T[1]))
        if bit == '1':
            lam = P.fq2_mul(P.fq2_sub(Qt[1], T[1]), P.fq2_inv(P.fq2_sub(Qt[0], T[0])))                      # chord through T and Q
            f = f12_mul(f, _line(lam, T, Pt))
            x3 = P.fq2_sub(P.fq2_sub(P.fq2_mul(lam, lam), T[0]), Qt[0]); T = (x3, P.fq2_sub(P.fq2_mul(lam, P.fq2_sub(T[0], x3)), T[1]))
    return f


FINAL_EXPONENT = (Q ** 12 - 1) // R


def final_exponentiation(f): return f12_pow(f, FINAL_EXPONENT)


def pairing(Pt, Qt): return final_exponentiation(miller_loop(Pt, Qt))


def pairing_product_is_one(pairs) -> bool:
    """prod_i e(P_i, Q_i) == 1 with one final exponentiation."""
    f = ONE12
    for Pt, Qt in pairs: f = f12_mul(f, miller_loop(Pt, Qt))
    return final_exponentiation(f) == ONE12

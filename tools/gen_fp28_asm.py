#!/usr/bin/env python3
"""Generates aleo_amd/csrc/fp28_mont_gen.h: Fq Montgomery product on 14 x 28-bit limbs (R = 2^392) for gfx950.

Why a second representation.  With 32-bit limbs every v_mad_u64_u32 needs a v_addc to catch its carry-out (the 64-bit
column accumulator can overflow): 288 + 288 instructions per product.  With 28-bit limbs a 64-bit accumulator holds a
whole column (28 products of < 2^29 * 2^29 — see the bounds below), so there are no carry catches: 392 mads + 3 simple
instructions per column = ~465 VALU instead of 649.  q == 1 mod 2^28 as well, so m_k = -acc mod 2^28.

Limb bounds ("classes").  N: every limb < 2^28.  Products accept limbs a_i * b_j < 2^58.2 for all i, j (28 of them plus
the carry-in stay below 2^64); the result is class N with the value < (A*B*q/R + 1) * q, R/q ~ 2^15.2.

Measured (profiles/r01_fq28_mul_bench.txt, tools/ubench/fq28_mul_bench.hip): 463 VALU / 392 mads per product, 81 G products/s
at sustained clocks against 60.7 G/s for the 12 x 32-bit block (short runs of either read low: the clock is still ramping).
Two accumulator chains per column were tried and are slower: a wave issues a mad every ~10 cycles whether or not it depends
on the previous one (profiles/r01_mad_dep_ubench.txt), so only the instruction count matters.
"""
import sys

Q = 0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001
N = 14
B = 28
MASK = (1 << B) - 1
P_LIMBS = [(Q >> (B * i)) & MASK for i in range(N)]
assert P_LIMBS[0] == 1

ACC = 'v[4:5]'; ACC_LO = 'v4'; ACC_HI = 'v5'
ACC1 = 'v[6:7]'; ACC1_LO = 'v6'; ACC1_HI = 'v7'      # second chain: the mads of a column alternate between the two
M0 = 8             # m_k in v[8 + k]
D0 = 22            # squaring: 2 * a_j in v[22 + j]
TMP = 'v36'
TWO_CHAINS = False   # alternate the mads of a column between two accumulators: measured SLOWER (58 vs 62 G/s at 2 waves/SIMD):
                     # a wave issues one mad per ~10 cycles whether or not it depends on the previous one (profiles/r01_mad_dep_ubench.txt)
S0 = 76            # p_j (j >= 1) in s[76 + j - 1]


def gen(name, square=False):
    L = []; w = L.append
    for j in range(1, N):
        w(f's_mov_b32 s{S0 + j - 1}, 0x{P_LIMBS[j]:07x}')
    w(f'v_mov_b32 {ACC_LO}, 0'); w(f'v_mov_b32 {ACC_HI}, 0')
    a = lambda i: f'%{i}'
    b = lambda j: f'%{N + j}'
    m = lambda i: f'v{M0 + i}'
    d = lambda j: f'v{D0 + j}'
    if square:
        for j in range(1, N):
            w(f'v_lshlrev_b32_e32 {d(j)}, 1, {a(j)}')
    for k in range(2 * N - 1):
        lo, hi = max(0, k - N + 1), min(k, N - 1)
        prods = []
        for i in range(lo, hi + 1):
            j = k - i
            if square:
                if i > j: continue
                prods.append((a(i), a(i) if i == j else d(j)))
            else:
                prods.append((a(i), b(j)))
        for i in range(lo, hi + 1):
            j = k - i
            if j >= 1:                       # m_i * p_j; p_0 = 1 is the "m_k * 1" mad below
                prods.append((m(i), f's{S0 + j - 1}'))
        # two dependent chains: even products extend ACC (which already holds the carry of the previous column), odd ones
        # ACC1 (started with a literal-0 addend); one 64-bit add folds them
        used1 = False
        for t, (x, y) in enumerate(prods):
            if not TWO_CHAINS or t % 2 == 0 or len(prods) < 4:
                w(f'v_mad_u64_u32 {ACC}, vcc, {x}, {y}, {ACC}')
            else:
                w(f'v_mad_u64_u32 {ACC1}, vcc, {x}, {y}, {ACC1 if used1 else 0}'); used1 = True
        if used1:
            w(f'v_add_co_u32_e32 {ACC_LO}, vcc, {ACC_LO}, {ACC1_LO}')
            w(f'v_addc_co_u32_e32 {ACC_HI}, vcc, {ACC_HI}, {ACC1_HI}, vcc')
        if k < N:
            w(f'v_sub_u32_e32 {TMP}, 0, {ACC_LO}')
            w(f'v_and_b32_e32 {m(k)}, 0x{MASK:x}, {TMP}')
            w(f'v_mad_u64_u32 {ACC}, vcc, {m(k)}, 1, {ACC}')
        else:
            w(f'v_and_b32_e32 {a(k - N)}, 0x{MASK:x}, {ACC_LO}')
        w(f'v_lshrrev_b64 {ACC}, {B}, {ACC}')
    w(f'v_mov_b32_e32 {a(N - 1)}, {ACC_LO}')
    body = '\\n\\t'.join(L)
    outs = ', '.join(f'"+v"(a[{i}])' for i in range(N))
    clob = ['"vcc"', f'"{ACC_LO}"', f'"{ACC_HI}"', f'"{ACC1_LO}"', f'"{ACC1_HI}"'] + [f'"v{M0 + i}"' for i in range(N)] + [f'"{TMP}"'] + [f'"s{S0 + j - 1}"' for j in range(1, N)]
    nvalu = sum(1 for x in L if x.startswith('v_'))
    nmad = sum(1 for x in L if x.startswith('v_mad'))
    if square:
        clob += [f'"{d(j)}"' for j in range(1, N)]
        return '\n'.join([f'// {name}: a <- a*a*2^-392 mod q (class N out), {nvalu} VALU ({nmad} mads).',
                          f'__device__ __forceinline__ void {name}(uint32_t (&a)[{N}]) {{',
                          f'  asm("{body}"', f'      : {outs}', '      :', f'      : {", ".join(clob)});', '}', ''])
    ins = ', '.join(f'"v"(b[{j}])' for j in range(N))
    return '\n'.join([f'// {name}: a <- a*b*2^-392 mod q (class N out), {nvalu} VALU ({nmad} mads).',
                      f'__device__ __forceinline__ void {name}(uint32_t (&a)[{N}], const uint32_t (&b)[{N}]) {{',
                      f'  asm("{body}"', f'      : {outs}', f'      : {ins}', f'      : {", ".join(clob)});', '}', ''])


HEADER = '''// GENERATED by tools/gen_fp28_asm.py — do not edit.  See that file for the design notes.
#pragma once
#include <stdint.h>

namespace aleo_mi355x {
'''


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else 'aleo_amd/csrc/fp28_mont_gen.h'
    s = HEADER + gen('mont28_mul_inplace') + '\n' + gen('mont28_sqr_inplace', square=True) + '\n}  // namespace aleo_mi355x\n'
    open(path, 'w').write(s)


if __name__ == '__main__':
    main()

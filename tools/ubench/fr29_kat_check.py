"""Checks the KAT lines of tools/ubench/fr29_mul_bench against big integers: r == a * b * 2^-261 mod r, limbs of r below 2^29 (top limb: what is left)."""
import sys
R = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001
val = lambda limbs: sum(int(x, 16) << (29 * i) for i, x in enumerate(limbs))
rows = [l.split()[2:] for l in open(sys.argv[1]) if l.startswith('KAT')]
ok = True
for k in range(0, len(rows), 3):
    a, b, r = val(rows[k]), val(rows[k + 1]), val(rows[k + 2])
    good = (r * (1 << 261) - a * b) % R == 0 and all(int(x, 16) < (1 << 29) for x in rows[k + 2][:8]) and r < (a * b * R // (1 << 261) // R + 2) * R
    print('pair %d: %s (a = %.2f r, b = %.2f r, result = %.3f r)' % (k // 3, 'ok' if good else 'WRONG', a / R, b / R, r / R)); ok &= good
sys.exit(0 if ok else 1)

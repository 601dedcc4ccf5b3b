"""Idle gaps of one steady-state proof in a rocprofv3 kernel trace of tools/varuna_native_prof.py (proofs delimited by k_fr_random): every interval of more than
min_us during which NO kernel runs, with the kernels before and after it.  The window runs from one proof's k_fr_random to the next one's, so it holds the END of a proof
and the START of the following one (result read-back, proof bytes, the caller's loop, the next proof's set-up and assignment upload): gaps behind the window's last
bucket-reduction launch are reported as lying between two proofs, and both sums are printed.  tools/proof_gaps.py <trace.csv> [min_us=20]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
marks = [i for i, r in enumerate(rows) if 'k_fr_random' in r['Kernel_Name']]
a, b = marks[-2], marks[-1]; seg = rows[a:b + 1]; t0 = int(seg[0]['Start_Timestamp'])
nm = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '').replace('aleo_mi355x::', '')[:36]
end = int(seg[0]['End_Timestamp']); last = seg[0]; tot = 0.0; between = 0.0
last_reduce = max((int(r['End_Timestamp']) for r in seg if 'k_prog_final' in r['Kernel_Name'] or 'k_seg_fold' in r['Kernel_Name']), default=None)
for r in seg[1:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > end:
        g = (s - end) / 1e3
        outside = last_reduce is not None and end >= last_reduce
        if g > min_us: print('%9.1f us  gap %7.1f us   after %-36s before %s%s' % ((end - t0) / 1e3, g, nm(last), nm(r), '   (between two proofs)' if outside else ''))
        tot += g
        if outside: between += g
    if e > end: end, last = e, r
print('span %.1f us, idle %.1f us in all: %.1f us inside the proof (its host turn-arounds), %.1f us between two proofs' % ((int(seg[-1]['Start_Timestamp']) - t0) / 1e3, tot, tot - between, between))

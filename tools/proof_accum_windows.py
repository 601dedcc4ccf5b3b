"""One steady-state proof in a rocprofv3 kernel trace of tools/varuna_native_prof.py (delimited by k_fr_random), cut at its accumulation kernels: for every stretch during
which NO k_accum28 runs — where it lies, how long it is, how much of it the card is busy at all, and the kernels that take the most time in it.  The accumulations are the
proof's arithmetic floor; this prints what a big proof spends around them.  tools/proof_accum_windows.py <kernel_trace.csv> [min_us=100]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
marks = [i for i, r in enumerate(rows) if 'k_fr_random' in r['Kernel_Name']]
a, b = marks[-2], marks[-1]; seg = rows[a:b]; t0 = int(seg[0]['Start_Timestamp']); t1 = int(rows[b]['Start_Timestamp'])
nm = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '').replace('aleo_mi355x::', '')[:30]
acc = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg if 'k_accum28' in r['Kernel_Name'])
merged = []
for s, e in acc:
    if merged and s <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], e)
    else: merged.append([s, e])
holes = []; cur = t0
for s, e in merged:
    if s > cur: holes.append((cur, s))
    cur = max(cur, e)
if t1 > cur: holes.append((cur, t1))
def union(iv):
    iv = sorted(iv); tot = 0; cs = ce = None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + (ce - cs if cs is not None else 0)
tot_hole = 0
for hs, he in holes:
    tot_hole += he - hs
    if (he - hs) / 1e3 < min_us: continue
    inside = [(max(int(r['Start_Timestamp']), hs), min(int(r['End_Timestamp']), he), nm(r)) for r in seg if int(r['End_Timestamp']) > hs and int(r['Start_Timestamp']) < he]
    agg = collections.Counter()
    for s, e, n in inside: agg[n] += e - s
    print('%9.1f us  no accumulation for %8.1f us, busy %8.1f us | %s' % ((hs - t0) / 1e3, (he - hs) / 1e3, union([(s, e) for s, e, _ in inside]) / 1e3,
                                                                          ', '.join('%s %.0f' % (k, v / 1e3) for k, v in agg.most_common(6))))
print('span %.1f us, an accumulation running %.1f us, none running %.1f us' % ((t1 - t0) / 1e3, sum(e - s for s, e in merged) / 1e3, tot_hole / 1e3))

"""Every launch of ONE steady-state proof in a rocprofv3 kernel trace of tools/varuna_native_prof.py (proofs delimited by k_fr_random), in start order:
start (us from the proof's first kernel), duration, gap to the latest end before it, kernel, grid.  Then launches, busy time and idle time.
tools/proof_timeline_full.py <kernel_trace.csv> [which=-2 (the last complete proof)]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
marks = [i for i, r in enumerate(rows) if 'k_fr_random' in r['Kernel_Name']]
a, b = marks[which], marks[which + 1]; seg = rows[a:b]; t0 = int(seg[0]['Start_Timestamp'])
nm = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '').replace('aleo_mi355x::', '')[:44]
end = t0; busy = 0.0; idle = 0.0
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - end) / 1e3
    print('%9.1f %8.1f  gap %7.1f  %-44s grid %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, nm(r), r.get('Grid_Size_X', r.get('Grid_Size', ''))))
    if s > end: idle += gap; busy += (e - s) / 1e3
    elif e > end: busy += (e - end) / 1e3
    end = max(end, e)
print('launches %d, span %.1f us (to the next proof\'s first kernel %.1f), busy (union) %.1f us, idle %.1f us' %
      (len(seg), (end - t0) / 1e3, (int(rows[b]['Start_Timestamp']) - t0) / 1e3, busy, idle))

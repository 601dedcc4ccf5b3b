"""Timeline of the long kernels (> min_us) of the last steady-state proof in a rocprofv3 kernel trace of tools/varuna_native_prof.py (proofs are
delimited by their k_fr_random launch): start (us from the proof's first kernel), duration (us), queue, kernel.  tools/proof_timeline.py <trace.csv> [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
marks = [i for i, r in enumerate(rows) if 'k_fr_random' in r['Kernel_Name']]
a, b = marks[-2], marks[-1]; t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if d > min_us: print('%9.1f %9.1f q%s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, d, r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('aleo_mi355x::', '')[:44]))

"""From a rocprofv3 kernel trace: over the last `frac` of the trace — span, union of kernel intervals, time an accumulation kernel (k_accum*) runs, and a
timeline of kernels longer than min_us inside one window of `win_ms` at the end.  tools/trace_window.py <trace.csv> [frac=0.4] [win_ms=40] [min_us=200]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4; win_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0; min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 200.0
t_first, t_last = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
cut = t_last - (t_last - t_first) * frac
seg = [r for r in rows if int(r['Start_Timestamp']) >= cut]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
alliv = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg]
acc = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg if 'k_accum' in r['Kernel_Name']]
span = alliv[-1][1] - alliv[0][0] if alliv else 0
print('last %.0f %% of the trace: span %.1f ms, union busy %.1f ms, an accumulation kernel running %.1f ms, %d launches' % (100 * frac, span / 1e6, union(alliv) / 1e6, union(acc) / 1e6 if acc else 0.0, len(seg)))
t0 = t_last - int(win_ms * 1e6)
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s >= t0 and (e - s) / 1e3 > min_us: print('%9.1f %9.1f q%s %s' % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('aleo_mi355x::', '')[:44]))

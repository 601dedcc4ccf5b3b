"""One lockstep call (tools/lockstep_probe.py <lg> trace <P>) in a rocprofv3 kernel trace, cut at its accumulation kernels: for every stretch between the end of one
k_accum28 launch and the start of the next — wall time, union of kernel intervals (GPU busy), launches, queues in use, and the kernels that take the most time
there.  The accumulations themselves are the call's arithmetic floor; everything this prints is what a lockstep call spends around them.
tools/lockstep_windows.py <kernel_trace.csv> [calls_back=1] [accums_per_call=5]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1; per = int(sys.argv[3]) if len(sys.argv) > 3 else 5
nm = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '').replace('aleo_mi355x::', '')[:34]
acc = [i for i, r in enumerate(rows) if 'k_accum28' in r['Kernel_Name']]
sel = acc[-(back * per + 1):len(acc) - (back - 1) * per] if back > 1 else acc[-(per + 1):]
def union(iv):
    if not iv: return 0
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
tot_gap = tot_acc = 0
for a, b in zip(sel[:-1], sel[1:]):
    e0 = int(rows[a]['End_Timestamp']); s1 = int(rows[b]['Start_Timestamp']); seg = [r for r in rows[a + 1:b] if int(r['Start_Timestamp']) >= int(rows[a]['Start_Timestamp'])]
    iv = [(max(int(r['Start_Timestamp']), e0), min(int(r['End_Timestamp']), s1)) for r in seg if int(r['End_Timestamp']) > e0 and int(r['Start_Timestamp']) < s1]
    agg = collections.Counter(); cnt = collections.Counter()
    for r in seg: agg[nm(r)] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); cnt[nm(r)] += 1
    qs = len({r.get('Queue_Id', '?') for r in seg})
    acc_ms = (int(rows[b]['End_Timestamp']) - s1) / 1e6
    print('gap %7.2f ms  busy %6.2f ms  %4d launches on %d queues, sum of kernel times %6.2f ms | next accumulation %6.2f ms' % ((s1 - e0) / 1e6, union(iv) / 1e6, len(seg), qs, sum(agg.values()) / 1e6, acc_ms))
    print('      ' + ', '.join('%s %dx %.0f us' % (k, cnt[k], v / 1e3) for k, v in agg.most_common(7)))
    tot_gap += s1 - e0; tot_acc += int(rows[b]['End_Timestamp']) - s1
print('between accumulations %.2f ms, accumulations %.2f ms' % (tot_gap / 1e6, tot_acc / 1e6))

"""GPU busy fraction from a rocprofv3 kernel trace (union of kernel intervals / span of the last `frac` of the trace) and the top kernels by time."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
cut = iv[0][0] + (iv[-1][1] - iv[0][0]) * 0.5            # steady state: the second half
iv = [x for x in iv if x[0] >= cut]
span = iv[-1][1] - iv[0][0]
busy = 0; cur_s, cur_e = iv[0][0], iv[0][1]
for s, e, _ in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in iv)
agg = collections.Counter()
for s, e, n in iv: agg[n.split('(')[0].replace('void ', '').replace('aleo_mi355x::', '')[:40]] += e - s
print('span %.1f ms, union busy %.1f ms (%.0f %%), sum of kernel times %.1f ms (overlap factor %.2f)' % (span / 1e6, busy / 1e6, 100.0 * busy / span, tot / 1e6, tot / busy))
for n, t in agg.most_common(8): print('  %-42s %6.1f ms (%.0f %% of the span)' % (n, t / 1e6, 100.0 * t / span))

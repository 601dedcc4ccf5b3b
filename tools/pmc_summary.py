"""Per-kernel average of one counter from a rocprofv3 counter_collection.csv: kernel, dispatches, mean and total counter value."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1]))); ctr = sys.argv[2]
agg = collections.OrderedDict()
for r in rows:
    if r.get('Counter_Name') != ctr: continue
    k = r['Kernel_Name'].split('(')[0]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
print('kernel,dispatches,mean_%s,total_%s' % (ctr, ctr))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%d,%.3f,%.3f' % (k, n, t / n, t))

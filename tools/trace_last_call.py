"""Prints the kernels of the last `count` launches of a rocprofv3 kernel trace (csv) as a timeline: start offset, duration, stream/queue, name.
usage: trace_last_call.py <dir with *_kernel_trace.csv> [marker kernel substring = k_prog_final] [calls back = 1]"""
import csv, glob, sys
d = sys.argv[1]; marker = sys.argv[2] if len(sys.argv) > 2 else 'k_prog_final'; back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
hi = ends[-back]; lo = ends[-back - 1] + 1
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:hi + 1]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f us  %8.1f us  q%-3s %s' % ((st - t0) / 1e3, (en - st) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))

"""Prints the kernel timeline of the LAST MSM in a rocprofv3 --kernel-trace csv (start offset, duration, gap, grid)."""
import csv, sys, glob
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob('gpurun_out/trace/*/*_kernel_trace.csv'))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
i0 = [i for i, r in enumerate(rows) if 'k_part_count' in r['Kernel_Name']][-1]
t0 = int(rows[i0]['Start_Timestamp']); prev = t0
for r in rows[i0:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:6.1f}  grid {r['Grid_Size_X']:>9}  {r['Kernel_Name'].split('(')[0][:44]}")
    prev = e

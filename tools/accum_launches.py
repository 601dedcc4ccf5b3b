"""The k_accum28 launches of the last proof in a rocprofv3 kernel trace of tools/varuna_native_prof.py: duration and grid of each (not a test)."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'k_fr_random' in r['Kernel_Name']]
seg = rows[marks[-2]:marks[-1]]
for r in seg:
    if 'k_accum28' in r['Kernel_Name']:
        print('k_accum28 %8.1f us  grid %s x wg %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))))

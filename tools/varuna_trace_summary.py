"""Per-proof kernel time from a rocprofv3 kernel trace of tools/varuna_native_prof.py: proofs are delimited by their k_fr_random launch."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'k_fr_random' in r['Kernel_Name']]
a, b = marks[-3], marks[-1]                     # two steady-state proofs
seg = rows[a:b]
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 2e6
agg = collections.Counter(); cnt = collections.Counter()
for r in seg:
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('aleo_mi355x::', '')[:48]
    agg[nm] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); cnt[nm] += 1
busy = sum(agg.values()) / 2e6
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg); union = 0; cs, ce = iv[0]
for s_, e_ in iv[1:]:
    if s_ > ce: union += ce - cs; cs, ce = s_, e_
    else: ce = max(ce, e_)
union = (union + ce - cs) / 2e6                 # kernels of different streams overlap: the union is the time the card had anything to do
accum = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg if 'k_accum' in r['Kernel_Name']); au = 0
if accum:
    cs, ce = accum[0]
    for s_, e_ in accum[1:]:
        if s_ > ce: au += ce - cs; cs, ce = s_, e_
        else: ce = max(ce, e_)
    au = (au + ce - cs) / 2e6
print('per proof: span %.3f ms, sum of kernel times %.3f ms, union busy %.3f ms, an accumulation kernel running %.3f ms, %d launches' % (span, busy, union, au, len(seg) // 2))
groups = {'msm': ('k_accum', 'k_seg', 'k_tree', 'k_bucket', 'k_masked', 'k_part', 'k_bin', 'k_scan', 'k_slice', 'k_gather', 'k_mont', 'k_task', 'k_order', 'k_canon'),
          'ntt': ('k_ntt',), 'field': ('k_fr_', 'k_ahp', 'k_spmv', 'k_div', 'k_eval'), 'copies/fills': ('copyBuffer', 'fillBuffer', 'elementwise')}
tot = collections.Counter()
for nm, v in agg.items():
    g = next((g for g, pats in groups.items() if any(p in nm for p in pats)), 'other'); tot[g] += v
print({g: round(v / 2e6, 3) for g, v in tot.items()})
for nm, v in agg.most_common(30): print('%-50s %5.1f x %8.1f us' % (nm, cnt[nm] / 2, v / 2e3))

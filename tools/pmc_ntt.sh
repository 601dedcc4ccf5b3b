#!/bin/bash
# SQ counters of the NTT kernels (one rocprofv3 --pmc pass, --kernel-trace only): tools/pmc_ntt.sh <out_dir under gpurun_out> <lg_n>
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out="gpurun_out/$1"; lg="$2"; mkdir -p "$out"
d="$out/ntt_${lg}_sq"; rm -rf "$d"; mkdir -p "$d"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$d" -o p -- python3 tools/prof_workload.py "ntt:$lg" 3 > "$d/run.json" 2> "$d/run.err" || { echo FAILED; tail -5 "$d/run.err"; exit 1; }
f=$(find "$d" -name '*counter_collection.csv' | head -1)
python3 - "$f" > "$out/ntt_${lg}_sq_per_kernel.json" <<'PY'
import csv, sys, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0]
    if 'ntt' not in k: continue
    a = agg[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
out = {k: {c: v[1] / v[0] for c, v in cs.items()} for k, cs in agg.items()}
for k, cs in out.items():
    if cs.get('SQ_LDS_IDX_ACTIVE'): cs['lds_bank_conflict_share'] = cs['SQ_LDS_BANK_CONFLICT'] / cs['SQ_LDS_IDX_ACTIVE']
    if cs.get('SQ_WAVE_CYCLES'):
        cs['valu_active_share'] = cs['SQ_ACTIVE_INST_VALU'] / cs['SQ_WAVE_CYCLES']; cs['wait_any_share'] = cs['SQ_WAIT_ANY'] / cs['SQ_WAVE_CYCLES']
        cs['wait_inst_lds_share'] = cs['SQ_WAIT_INST_LDS'] / cs['SQ_WAVE_CYCLES']
print(json.dumps({'what': 'rocprofv3 --pmc averages per launch, forward NTT', 'kernels': out}, indent=1))
PY
find "$d" -name '*.csv' -size +1M -delete
cat "$out/ntt_${lg}_sq_per_kernel.json"

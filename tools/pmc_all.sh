#!/bin/bash
# HBM traffic counters per kernel: separate `rocprofv3 --pmc` passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass on
# gfx950), each with --kernel-trace only, of one tools/prof_workload.py workload:
#   tools/pmc_all.sh <out_dir under gpurun_out> <workload> [<workload> ...]
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out="gpurun_out/$1"; shift
mkdir -p "$out"
for w in "$@"; do
  tag="${w//:/_}"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d="$out/${tag}_$ctr"; rm -rf "$d"; mkdir -p "$d"
    rocprofv3 --pmc "$ctr" --kernel-trace --output-format csv -d "$d" -o p -- python3 tools/prof_workload.py "$w" 3 > "$d/run.json" 2> "$d/run.err" || { echo "FAILED $w $ctr"; tail -5 "$d/run.err"; exit 1; }
    f=$(find "$d" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 tools/pmc_summary.py "$f" "$ctr" > "$out/${tag}_${ctr}_per_kernel.csv"
    find "$d" -name '*.csv' -size +2M -delete
  done
  cat "$out/${tag}_FETCH_SIZE/run.json"
done

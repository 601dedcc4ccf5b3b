#!/bin/bash
# Runs each workload of tools/prof_workload.py under `rocprofv3 --kernel-trace --stats` and keeps the per-kernel summary:
#   tools/prof_all.sh <out_dir under gpurun_out> <workload> [<workload> ...]
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out="gpurun_out/$1"; shift
mkdir -p "$out"
for w in "$@"; do
  tag="${w//:/_}"
  d="$out/$tag"; rm -rf "$d"; mkdir -p "$d"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o p -- python3 tools/prof_workload.py "$w" 10 > "$d/run.json" 2> "$d/run.err" || { echo "FAILED $w"; tail -5 "$d/run.err"; exit 1; }
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv"
  find "$d" -name '*.db' -delete; find "$d" -name '*kernel_trace.csv' -delete
  cat "$d/run.json"
done

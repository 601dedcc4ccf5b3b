#!/bin/bash
# The headline command under rocprofv3 (--kernel-trace --stats) with only the timed workload enabled, so the per-kernel averages are
# those of the 2^20 host-scalar MSM; then the full default bench line.  tools/prof_bench.sh <out_dir under gpurun_out>
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out="gpurun_out/$1"; mkdir -p "$out"; d="$out/bench_prof"; rm -rf "$d"; mkdir -p "$d"
rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kzg-chain --proof-proxy-lg 0 --concurrent-callers 0 --no-variants --varuna-lg 0 > "$out/bench_under_rocprof.json" 2> "$d/run.err" || { echo FAILED; tail -5 "$d/run.err"; exit 1; }
f=$(find "$d" -name '*kernel_stats.csv' | head -1); cp "$f" "$out/bench_kernel_stats.csv"
find "$d" -name '*kernel_trace.csv' -delete
python3 bench.py --steps 20 --warmup 3 > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"

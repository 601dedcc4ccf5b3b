#!/bin/bash
# Round-5 evidence in one GPU call: the headline under rocprofv3 (--kernel-trace --stats), the default bench line, the NTT kernel pair, the per-proof
# kernel breakdown + idle gaps of the 2^15-constraint proof, the PMC traffic passes of the headline call.   tools/r05_final_profiles.sh
set -uo pipefail
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
out=gpurun_out/r05final; rm -rf $out; mkdir -p $out
timeout -k 10 500 bash tools/prof_bench.sh r05final > $out/prof_bench.log 2>&1; tail -2 $out/prof_bench.log
timeout -k 10 200 bash tools/prof_all.sh r05final ntt:22 > $out/prof_ntt.log 2>&1
d=$out/vp15; mkdir -p $d
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python3 tools/varuna_native_prof.py 15 12 > $d/run.log 2>&1
f=$(find $d -name '*kernel_trace.csv' | head -1)
python3 tools/varuna_trace_summary.py $f > $out/varuna_native_2^15_per_proof.txt; python3 tools/proof_gaps.py $f 20 > $out/varuna_2^15_idle_gaps.txt; python3 tools/proof_timeline_full.py $f > $out/varuna_2^15_timeline.txt
find $d -name '*.db' -delete; find $d -name '*kernel_trace.csv' -delete
timeout -k 10 300 bash tools/pmc_all.sh r05final host_msm:20 > $out/pmc.log 2>&1
head -3 $out/varuna_native_2^15_per_proof.txt; tail -1 $out/varuna_2^15_idle_gaps.txt; ls $out

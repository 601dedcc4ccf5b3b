#!/bin/bash
# A/B of the order in which the library creates its streams (ALEO_MI355X_STREAM_ORDER: 1 = main streams first, the default; 0 = every context creates its own at first use): lockstep call
# of 8 and 16 proofs, one 2^15 proof, one 2^20 proof, the headline MSM.  (profiles/r05_stream_order_modes*.txt were made with an experimental build that knew modes 1-5:
# 1 = every stream of every context up front, 2 = every main stream, 3 = the first eight main streams [what shipped], 4 = 3 + their side streams, 5 = 4 + their high-priority streams.)
O=${1:-gpurun_out/stream_order_probe.txt}
for v in ${MODES:-0 1}; do
  for P in 8 16; do echo -n "STREAM_ORDER=$v lockstep P $P: " >> $O; ALEO_MI355X_STREAM_ORDER=$v timeout -k 10 120 python3 tools/lockstep_probe.py 15 trace $P 12 2>/dev/null | tail -1 >> $O || exit 1; done
  echo -n "STREAM_ORDER=$v " >> $O; ALEO_MI355X_STREAM_ORDER=$v REPS=20 timeout -k 10 120 python3 tools/prove_quick.py 15 2>/dev/null | tail -1 | cut -c1-170 >> $O || exit 1
  echo -n "STREAM_ORDER=$v " >> $O; ALEO_MI355X_STREAM_ORDER=$v REPS=4 timeout -k 10 200 python3 tools/prove_quick.py 20 2>/dev/null | tail -1 | cut -c1-170 >> $O || exit 1
  echo -n "STREAM_ORDER=$v " >> $O; ALEO_MI355X_STREAM_ORDER=$v timeout -k 10 120 python3 tools/host_scalars_ab.py 20 2>/dev/null | cut -c1-100 | tr '\n' ' ' >> $O || exit 1; echo >> $O
done

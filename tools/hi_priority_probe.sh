# High- against normal-priority `hi` streams (ALEO_MI355X_HI_PRIORITY) for the 2^20-constraint proof and the chunked host-scalar MSM; profiles/r05_hi_priority_ab.txt was made
# with the experimental stream-order modes 3 and 5 (see tools/stream_order_probe.sh); the shipped library knows ALEO_MI355X_STREAM_ORDER=0/1 and ALEO_MI355X_PIPELINE_HI.
O=gpurun_out/r05_hi_priority_ab.txt
for cfg in "1 1" "1 0" "0 1" "1 1" "1 0"; do set -- $cfg
  echo -n "STREAM_ORDER=$1 HI_PRIORITY=$2 " >> $O; ALEO_MI355X_STREAM_ORDER=$1 ALEO_MI355X_HI_PRIORITY=$2 REPS=4 timeout -k 10 200 python3 tools/prove_quick.py 20 2>/dev/null | tail -1 | cut -c1-170 >> $O || exit 1
  echo -n "STREAM_ORDER=$1 HI_PRIORITY=$2 " >> $O; ALEO_MI355X_STREAM_ORDER=$1 ALEO_MI355X_HI_PRIORITY=$2 timeout -k 10 120 python3 tools/host_scalars_ab.py 20 21 22 2>/dev/null | cut -c1-75 | tr '\n' ' ' >> $O || exit 1; echo >> $O
done
cat $O

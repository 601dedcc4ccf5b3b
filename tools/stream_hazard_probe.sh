#!/bin/bash
# tools/stream_hazard_probe.py under the stream arrangements of api.hip (first_use): "ORDER POOL PIPELINE_HI" triples
O=${1:-gpurun_out/stream_hazard_probe.jsonl}
for cfg in ${CFGS:-"1,4,0" "1,4,1" "0,8,1" "1,8,0" "1,2,0" "1,4,0"}; do
  IFS=, read a b c <<< "$cfg"
  ALEO_MI355X_STREAM_ORDER=$a ALEO_MI355X_HI_POOL=$b ALEO_MI355X_PIPELINE_HI=$c timeout -k 10 400 python3 tools/stream_hazard_probe.py 2>/dev/null | tail -1 >> $O || exit 1
done

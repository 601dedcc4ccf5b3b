#!/bin/bash
# Where the time of a large transform goes: the NTT kernels rebuilt with -DALEO_NTT_PROBE=1 (no butterflies: HBM + LDS staging + the inter-pass
# product), =2 (no butterflies, no inter-pass product: the memory phases alone), =3 (no HBM reads or writes: the arithmetic alone), each linked
# into a scratch copy of the library and timed by tools/ntt_ab.py.  The probe builds compute WRONG results by construction; they never replace the
# in-tree library here (on the GPU box the copy is scratch).  Build here (no GPU needed):  tools/ntt_phase_probe.sh build
# Run on the box:  tools/ntt_phase_probe.sh run <out_dir under gpurun_out> <lg_n> ...
set -euo pipefail
cd "$(dirname "$0")/.."
ab=build/ab; lib=aleo_amd/lib; src=aleo_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $ab
  for v in 1 2 3; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xarch_host -mbmi2 -Xarch_host -madx -Wno-unused-function -Wno-unused-variable -DALEO_NTT_PROBE=$v -c $src/ntt.hip -o $ab/ntt_p$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ab/libprobe$v.so $lib/api.o $lib/msm.o $ab/ntt_p$v.o $lib/frops.o $lib/wire.o $lib/g2.o $lib/varuna.o $lib/sponge.o
  done
  exit 0
fi
out="gpurun_out/$2"; mkdir -p "$out"; shift 2
python3 tools/ntt_ab.py "$@" > "$out/full.jsonl"
for v in 1 2 3; do
  ALEO_MI355X_LIB="$PWD/$ab/libprobe$v.so" python3 tools/ntt_ab.py "$@" > "$out/probe$v.jsonl"      # loaded through the override: the in-tree library is never replaced
done

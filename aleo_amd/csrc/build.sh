#!/bin/bash
# Builds libaleo_mi355x.so for gfx950 in-tree (aleo_amd/lib/).  hipcc cross-compiles without a GPU.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"; mkdir -p "$out"
python3 "$here/../../tools/gen_fp_asm.py" "$here/fp_mont_gen.h"
python3 "$here/../../tools/gen_fp28_asm.py" "$here/fp28_mont_gen.h"
python3 "$here/../../tools/gen_fr29_asm.py" "$here/fr29_mont_gen.h"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xarch_host -mbmi2 -Xarch_host -madx -Wall -Wno-unused-function -Wno-unused-variable ${ALEO_MI355X_CXXFLAGS:-}"
objs=(); pids=()
for f in api msm ntt frops wire g2 varuna sponge; do
  rm -f "$out/$f.o"                      # a failed compile must not link a stale object
  "$HIPCC" $FLAGS -c "$here/$f.hip" -o "$out/$f.o" &
  pids+=("$!"); objs+=("$out/$f.o")
done
for p in "${pids[@]}"; do wait "$p" || { echo "build.sh: a compile failed" >&2; exit 1; }; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out/libaleo_mi355x.so" "${objs[@]}"
echo "built $out/libaleo_mi355x.so"

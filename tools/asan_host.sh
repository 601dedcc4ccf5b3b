#!/bin/bash
# Builds the host-only translation units of the product (aleo_amd/csrc/wire.hip: parsers of untrusted bytes; sponge.hip: Poseidon, the Fiat-Shamir
# sponge and the random stream) as plain C++ with AddressSanitizer + UndefinedBehaviorSanitizer, links them with tests/cpp/wire_fuzz.cpp and runs it.
# No GPU, no HIP runtime (the two files contain no kernel and make no HIP call).  Usage: tools/asan_host.sh <out dir> <proof1... string>
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; out="$1"; mkdir -p "$out"
CXXF="-std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -mbmi2 -madx -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$root/include"
printf '#include <string>\nnamespace aleo_mi355x { thread_local std::string g_last_error; }\nextern "C" const char* aleo_mi355x_last_error(void) { return aleo_mi355x::g_last_error.c_str(); }\n' > "$out/stub.cpp"
g++ $CXXF -x c++ -c "$root/aleo_amd/csrc/wire.hip" -o "$out/wire.o" &
g++ $CXXF -x c++ -c "$root/aleo_amd/csrc/sponge.hip" -o "$out/sponge.o" &
g++ $CXXF -c "$out/stub.cpp" -o "$out/stub.o" &
g++ $CXXF -c "$root/tests/cpp/wire_fuzz.cpp" -o "$out/wire_fuzz.o" &
wait
g++ -fsanitize=address,undefined "$out/wire.o" "$out/sponge.o" "$out/stub.o" "$out/wire_fuzz.o" -o "$out/wire_fuzz"
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 "$out/wire_fuzz" "$2"

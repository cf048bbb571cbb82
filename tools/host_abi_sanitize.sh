#!/bin/bash
# Builds tests/host_abi_fuzz.cpp against a HOST-ONLY, sanitizer-instrumented build of the library's own sources (no device code: the two
# .hip translation units are compiled with --cuda-host-only and an empty stub stands in for their fat binaries -- nothing is launched).
#   tools/host_abi_sanitize.sh <out_dir>      -> <out_dir>/host_abi_fuzz
set -e
R=$(cd "$(dirname "$0")/.." && pwd); O=${1:?out dir}; mkdir -p $O; cd $O
CL=/opt/rocm/lib/llvm/bin/clang++
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined"
for f in adanerf_hip launch_f32; do
  hipcc --offload-arch=gfx950 --cuda-host-only -O1 -g -std=c++17 $SAN -I $R/adanerf_amd/csrc -I $R/include -c $R/adanerf_amd/csrc/$f.hip -o $f.o 2>/dev/null &
done
for f in format pack; do $CL -O1 -g -std=c++17 $SAN -I $R/adanerf_amd/csrc -c $R/adanerf_amd/csrc/$f.cpp -o $f.o & done
$CL -O1 -g -std=c++17 $SAN -I $R/include -c $R/tests/host_abi_fuzz.cpp -o harness.o &
wait
# the fat-binary symbols the host-only objects refer to: named after the hash of each translation unit
: > stub.s
for s in $( (hipcc $SAN adanerf_hip.o launch_f32.o format.o pack.o harness.o -o /dev/null 2>&1 || true) | grep -o "__hip_fatbin_[0-9a-f]\+" | sort -u); do
  printf '.globl %s\n.section .hip_fatbin,"a"\n.p2align 12\n%s:\n.zero 4096\n' $s $s >> stub.s
done
$CL -c stub.s -o stub.o
hipcc $SAN adanerf_hip.o launch_f32.o format.o pack.o harness.o stub.o -o host_abi_fuzz
echo built $O/host_abi_fuzz

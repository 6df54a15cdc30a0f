#!/bin/bash
# Builds a variant of the library for same-box A/B timing (tools/ab_bench.sh, tools/ab_long.sh):
#   tools/build_variant.sh <name> [-DFLAG ...]   ->   decodingustools_amd/lib/libcallable_hip_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
S=decodingustools_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function "$@" \
  $S/callable_loci.hip $S/qual_pack.cpp $S/host_coverage.cpp $S/bam_io.cpp $S/report.cpp $S/haplogroup.cpp -lz -ldl \
  -o decodingustools_amd/lib/libcallable_hip_$name.so 2>&1 | grep -E "error|spill" || true
ls -la decodingustools_amd/lib/libcallable_hip_$name.so

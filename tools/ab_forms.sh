#!/bin/bash
# Same-box comparison of the forms of k_pileup on the long-read shape (tuning build: CL_FORCE_LONG picks the form):
#   tools/ab_forms.sh [length]      forms 2 (run table) and 4 (block-parallel CIGAR scan), twice each
T=decodingustools_amd/lib/libcallable_hip_tuning.so
export KB_LEN=${1:-20000000}
for F in 2 4 2 4; do
  echo "== form $F"; DUT_CALLABLE_LIB=$T CL_FORCE_LONG=$F timeout -k 10 300 python tools/longread_bench.py 2>&1 | tail -1 || exit 1
done

#!/bin/bash
# same-box comparison of library builds on the long-read shape: tools/ab_libs.sh <length> <lib> <lib> ...   (each twice, interleaved)
export KB_LEN=$1; shift
for rep in 1 2; do
  for L in "$@"; do
    echo "== $L"; DUT_CALLABLE_LIB=$L timeout -k 10 300 python tools/longread_bench.py 2>&1 | tail -1 || exit 1
  done
done

#!/bin/bash
# same-box A/B of two library builds on the long-read shape (tools/longread_bench.py) and the HiFi shape
A=$1; B=$2
for L in "$A" "$B" "$A" "$B"; do
  echo "== $L"; DUT_CALLABLE_LIB=$L KB_LEN=${KB_LEN:-20000000} python tools/longread_bench.py 2>&1 | tail -1 || exit 1
done
for L in "$A" "$B"; do
  echo "== hifi $L"; DUT_CALLABLE_LIB=$L python tools/hifi_bench.py 2>&1 | tail -2 || exit 1
done

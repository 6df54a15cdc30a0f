#!/bin/bash
# Timing experiments on the long-read shape: CL_ABLATE bits (results are wrong with any bit set):
#   1 = no run is consumed (scan only), 4 = units located and listed but not applied
export KB_LEN=${KB_LEN:-20000000}
for ab in ${LONG_ABLATES:-0 4 1 0}; do
  echo "== CL_ABLATE=$ab"; CL_ABLATE=$ab python tools/longread_bench.py 2>&1 | tail -1 || exit 1
done

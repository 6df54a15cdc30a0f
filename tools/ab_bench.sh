#!/bin/bash
# Same-box A/B of two builds of the library (boxes differ by a few percent, so compare within one call):
#   tools/ab_bench.sh <libA.so> <libB.so> [rounds]
# prints value / ms_per_step / kernel groups of bench.py (headline workload only) for A, B, A, B ...
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do
  for L in "$A" "$B"; do
    DUT_CALLABLE_LIB=$L python bench.py --steps 40 --no-secondary --no-traffic --cpu-sample 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']/1e9,2), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['all_kernel_ms'].items()})" || exit 1
  done
done

#!/bin/bash
# Timing experiments on the run-table form (tuning build; results are wrong with any bit set):
#   0 all, 4 pieces fetched and their qualities loaded but nothing applied, 16 applied without loading qualities,
#   20 entries streamed only, 2 no pieces at all (candidates + final phase)
export KB_LEN=${KB_LEN:-20000000}
export DUT_QUAL_FORM=bytes          # the run table exists in the byte forms only
for ab in ${RT_ABLATES:-0 4 16 20 2 0}; do
  echo "== CL_ABLATE=$ab"; DUT_CALLABLE_LIB=decodingustools_amd/lib/libcallable_hip_tuning.so CL_ABLATE=$ab python tools/longread_bench.py 2>&1 | tail -1 || exit 1
done

#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats over the long-read shape (BASELINE configs[2], 20 Mb by default) and the
# site pileup (configs[4]) -- the kernels the headline bench does not launch.
cd /tmp && export TMPDIR=/tmp
for what in longread site; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_${what}_trace
  mkdir -p "$out"
  if [ $what = longread ]; then export KB_LEN=${KB_LONG_LEN:-20000000}; else unset KB_LEN; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 $GRAFT_REPO_ROOT/tools/${what}_bench.py > "$out/run.log" 2> "$out/run.err" || { echo "$what failed"; tail -3 "$out/run.err"; continue; }
  echo "== $what"; tail -2 "$out/run.log"
  f=$(find "$out" -name '*_kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -8 "$f"
done

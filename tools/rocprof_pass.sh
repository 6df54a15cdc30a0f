#!/bin/bash
# One rocprofv3 pass over bench.py on the GPU box.  usage: tools/rocprof_pass.sh <tag> <rocprofv3 args...>
# (counters go in their own pass, never combined with the tracing domains other than kernel-trace)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p "$out"
rocprofv3 "$@" --output-format csv -d "$out" -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-secondary --no-traffic --no-files --no-wgs-point --min-time 0 --max-blocks 1 > "$out/bench.json" 2> "$out/bench.err"
find "$out" -name '*.csv' | head -20

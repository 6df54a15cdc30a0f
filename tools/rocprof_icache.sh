#!/bin/bash
# GPU box: instruction-cache counters of the long-read k_pileup (20 Mb contig) and of the headline k_pileup.
cd /tmp && export TMPDIR=/tmp
export KB_LEN=${KB_LEN:-20000000}
for what in longread kbench; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_icache_$what
  mkdir -p "$out"
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d "$out" -- python3 $GRAFT_REPO_ROOT/tools/$([ $what = longread ] && echo longread_bench.py || echo kbench.py) > "$out/run.log" 2> "$out/run.err" || { echo "$what failed"; tail -3 "$out/run.err"; continue; }
  echo "== $what"; tail -1 "$out/run.log"
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "clk::" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
done

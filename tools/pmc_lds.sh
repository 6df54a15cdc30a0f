#!/bin/bash
# LDS conflict share of the short-read k_pileup for a library build: tools/pmc_lds.sh <tag> <lib.so>
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmclds_$1; rm -rf "$out"; mkdir -p "$out"
DUT_CALLABLE_LIB=$2 timeout -k 5 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$out" -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-secondary --no-traffic --no-files --no-wgs-point --min-time 0 --max-blocks 1 > "$out/run.log" 2> "$out/run.err" || { echo "pass failed"; tail -3 "$out/run.err"; exit 1; }
python3 - "$out" "$1" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_pileup" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in sorted(d.items())}
print(sys.argv[2], {k.replace("SQ_", ""): "%.4g" % v for k, v in m.items()}, "conflict share %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]))
PY

#!/bin/bash
# Instruction counters of the short-read k_pileup per CL_ABLATE setting (0 all, 1 segments listed but not consumed,
# 2 no pass over the reads: clear + final phase only).  GPU box; one rocprofv3 pass per setting.
cd /tmp && export TMPDIR=/tmp
for ab in 0 1 2; do
  out=$GRAFT_REPO_ROOT/gpurun_out/profshort_ab$ab
  mkdir -p "$out"
  CL_ABLATE=$ab rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$out" -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-secondary --no-traffic --min-time 0 > "$out/run.log" 2> "$out/run.err" || { echo "pass $ab failed"; tail -3 "$out/run.err"; continue; }
  echo "== CL_ABLATE=$ab"
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_pileup" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
done

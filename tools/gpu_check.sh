#!/bin/bash
# GPU box: parity tests, kernel timing (optionally ablations) and the instruction counters of k_pileup.
# usage: tools/gpu_check.sh [ablates, default 0,1,2]
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -2 gpurun_out/pytest_gpu.log
KB_ABLATES=${1:-0,1,2} timeout -k 10 200 python tools/kbench.py 2>&1 | tail -4
CL_ABLATE=0 tools/rocprof_pass.sh ab0 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY > /dev/null
python - <<PY
import csv, glob, collections, os
f=sorted(glob.glob("gpurun_out/prof_ab0/runc/*_counter_collection.csv"), key=os.path.getmtime)[-1]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "pileup" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k.replace("SQ_",""): sum(v)/len(v) for k,v in sorted(d.items())}
print({k: "%.4g"%v for k,v in m.items()})
print("model ms: %.4f" % ((4*m["INSTS_VALU"]+m["INSTS_SALU"])/1024/2.2e9*1e3))
PY

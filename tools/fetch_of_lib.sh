#!/bin/bash
# GPU box: k_pileup time and HBM fetch of an alternative build of the library.  usage: tools/fetch_of_lib.sh <lib.so> <tag>
export DUT_CALLABLE_LIB=$(readlink -f $1)
KB_ABLATES=0 timeout -k 10 200 python tools/kbench.py 2>&1 | tail -1
tools/rocprof_pass.sh $2 --pmc FETCH_SIZE > /dev/null
python - <<PY
import csv,glob,os
f=sorted(glob.glob("gpurun_out/prof_$2/*/*_counter_collection.csv"),key=os.path.getmtime)[-1]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "pileup" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("$2: 2 x FETCH_SIZE =", round(2*sum(v)/len(v)*1024/1e6,1), "MB")
PY

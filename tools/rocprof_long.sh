#!/bin/bash
# rocprofv3 counter passes over tools/longread_bench.py (config-3 shape, 20 Mb) on the GPU box.
set -e
cd /tmp && export TMPDIR=/tmp
export KB_LEN=${KB_LEN:-20000000}
for grp in "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INSTS_FLAT" \
           "mem FETCH_SIZE" ; do
  set -- $grp; tag=$1; shift
  out=$GRAFT_REPO_ROOT/gpurun_out/proflong_$tag
  mkdir -p "$out"
  rocprofv3 --pmc "$@" --output-format csv -d "$out" -- python3 $GRAFT_REPO_ROOT/tools/longread_bench.py > "$out/run.log" 2> "$out/run.err" || { echo "pass $tag failed"; tail -3 "$out/run.err"; continue; }
  echo "pass $tag ok"
done
python3 - <<'PY'
import csv, glob, os, collections
G = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(G + "/proflong_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_pileup" in k or "prep_long" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s %.4g" % (c, sum(v) / len(v)))
PY

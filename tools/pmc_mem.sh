#!/bin/bash
# Memory-pipe counters of k_pileup (vector memory address / data units, L1, address translation, L2, fabric), one
# rocprofv3 --pmc pass per group, over a command:   tools/pmc_mem.sh <tag> <program> [args...]
#
# <program> must be the ELF binary that does the work, e.g. `python3 tools/longread_bench.py` -- never a script started
# through its #! line, `env VAR=.. prog`, `bash -c`, `taskset`, `numactl` or anything else that execs again: under --pmc
# the profiler's preloaded library has initialised the GPU before the program starts, and an exec from such a process
# takes the whole box down on this pool.  Settings go into the environment BEFORE this script is called (export them).
set -e
tag=$1; shift
prog=$(command -v "$1" || true)
case "$(basename "${prog:-$1}")" in env|bash|sh|dash|zsh|taskset|numactl|nice|timeout|stdbuf) echo "pmc_mem.sh: '$1' execs its argument: refused (see the header)"; exit 2;; esac
if [ -z "$prog" ] || [ "$(head -c4 "$prog" | od -An -c | tr -d ' ')" != "177ELF" ]; then
  echo "pmc_mem.sh: '$1' is not an ELF binary (a script would exec its interpreter under the profiler): refused"; exit 2
fi
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmcmem_${tag}_$i
  rm -rf "$out"; mkdir -p "$out"
  timeout -k 5 ${PMC_PASS_TIMEOUT:-150} rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out" -- "$@" > "$out/run.log" 2> "$out/run.err" || { echo "pass $i ($grp) failed"; grep -m2 -i "error\|exceeds" "$out/run.err"; continue; }
  echo "pass $i ok"
done
python3 - "$tag" <<'PY'
import csv, glob, os, collections, sys
G = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(G + f"/pmcmem_{sys.argv[1]}_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_pileup" in k or "k_site" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(G + f"/pmcmem_{sys.argv[1]}_*/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_pileup" in k or "k_site" in k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in acc.items():
    d = dur.get(k, [0])
    print(k, " launches", len(d), " mean us under the counters %.1f" % (sum(d) / max(len(d), 1)))
    for c, v in sorted(cs.items()):
        print("   %-42s %.5g" % (c, sum(v) / len(v)))
PY

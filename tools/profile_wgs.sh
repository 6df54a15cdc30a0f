#!/bin/bash
# GPU box: the whole-genome step on ONE GPU (bench.py --workload wgs: BASELINE configs[3], all 25 hg38 contigs resident,
# what configs.wgs_1gpu of the N = 1 line measures) under rocprofv3 -- a kernel trace with statistics, then FETCH_SIZE
# and WRITE_SIZE, each counter in a pass of its own (never beside a tracing domain other than the kernel trace).
# WGS_SCALE < 1 shortens the contigs (rehearsal).  The program after `--` is python3 itself (no exec hop).
cd /tmp && export TMPDIR=/tmp
# one stream under the profiler: the contigs' kernels then run one after the other and a kernel's traced duration is its own
# (bench.py measures them one at a time for the same reason; the timed step itself enqueues on four streams)
export DUT_WGS_SIDE_STREAMS=0
for pass in "trace --kernel-trace --stats" "fetch --pmc FETCH_SIZE" "write --pmc WRITE_SIZE"; do
  set -- $pass; tag=$1; shift
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_wgs_$tag
  rm -rf "$out"; mkdir -p "$out"
  timeout -k 10 ${WGS_PASS_TIMEOUT:-400} rocprofv3 "$@" --output-format csv -d "$out" -- python3 $GRAFT_REPO_ROOT/bench.py --workload wgs --steps 5 --warmup 2 --min-time 0 --max-blocks 1 --wgs-scale ${WGS_SCALE:-1.0} > "$out/bench.json" 2> "$out/bench.err" || { echo "wgs $tag failed"; tail -3 "$out/bench.err"; continue; }
  echo "wgs $tag ok"
done
python3 - <<'PY'
import csv, glob, os, collections, json
G = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
out = {}
for f in glob.glob(G + "/prof_wgs_trace/*/*_kernel_stats.csv"):
    out["kernel_stats"] = [r for r in csv.DictReader(open(f))][:8]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for tag in ("fetch", "write"):
    for f in glob.glob(G + f"/prof_wgs_{tag}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out["counters_summed_over_launches"] = {k: dict(v) for k, v in acc.items() if "clk::" in k}
out["launches"] = {k: dict(v) for k, v in cnt.items() if "clk::" in k}
try:
    out["bench_line_under_the_tracer"] = json.loads(open(G + "/prof_wgs_trace/bench.json").read().strip().splitlines()[-1])
except Exception as e:
    out["bench_line_under_the_tracer"] = str(e)
json.dump(out, open(G + "/r04_wgs_1gpu_profile.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel_stats", "counters_summed_over_launches", "launches")}, indent=1)[:3000])
PY

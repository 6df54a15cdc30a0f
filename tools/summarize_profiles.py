#!/usr/bin/env python3
"""Tooling: condense the rocprofv3 passes under gpurun_out/prof_* (tools/profile_all.sh) into the files
committed under profiles/: <tag>_kernel_stats.csv, <tag>_pmc_summary.json, traffic.json."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")

def newest(pattern):
    f = sorted(glob.glob(pattern), key=os.path.getmtime)
    return f[-1] if f else None

ks = newest(os.path.join(G, "prof_trace", "**", "*_kernel_stats.csv").replace("**", "*"))
if ks:
    shutil.copy(ks, os.path.join(P, f"{tag}_kernel_stats.csv"))
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("prof_fetch", "prof_write", "prof_sq1", "prof_sq2", "prof_sq3"):
    f = newest(os.path.join(G, d, "*", "*_counter_collection.csv"))
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        pmc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(pmc.items()) if "clk::" in k}
json.dump(summ, open(os.path.join(P, f"{tag}_pmc_summary.json"), "w"), indent=1)
pile = {("k_pileup_rows" if "k_pileup_rows" in k else "k_pileup_byte_form"): v for k, v in summ.items() if "k_pileup" in k}
if pile and all("FETCH_SIZE" in v and "WRITE_SIZE" in v for v in pile.values()):
    import subprocess
    try:
        build = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except Exception:
        build = None
    out = {"build": build, "round": tag,
           "workload": "chr21 30x (bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-secondary --no-traffic --min-time 0; the line's byte_form leg launches the byte kernel on the same contig)",
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py; "
                  "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE counts half of a wide streaming read); raw = FETCH + WRITE"}
    for name, v in pile.items():
        out[name] = {"hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024), "raw_bytes_per_launch": int((v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024),
                     "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"]}
    if "k_pileup_rows" in out:
        out["k_pileup_hbm_bytes_per_launch"] = out["k_pileup_rows"]["hbm_bytes_per_launch"]     # (the key bench.py's traffic_from_profiles reads)
    json.dump(out, open(os.path.join(P, "traffic.json"), "w"), indent=1)
b = os.path.join(G, "bench_final.json")
if os.path.exists(b):
    shutil.copy(b, os.path.join(P, f"{tag}_bench.json"))
print(json.dumps({k: {c: round(v, 1) for c, v in cs.items() if c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")} for k, cs in summ.items()}, indent=1))

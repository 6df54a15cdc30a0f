#!/usr/bin/env python3
"""Tooling: end-to-end `coverage` on files (BGZF inflate + record parse + admission + H2D + kernels +
D2H + BED text), with the host stages timed (DUT_TIMING=1).  Writes a chr21-shaped BAM first.
Not bench.py: the judged metric is the device-resident rate; this is the labelled PCIe/host-inclusive one."""
import os, sys, time, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from decodingustools_amd import synth, build as _b
import e2e_bench_lib as EL

out = os.environ.get("E2E_DIR", "/tmp/e2e")
os.makedirs(out, exist_ok=True)
L = int(os.environ.get("E2E_LEN", 46_709_983))
depth = float(os.environ.get("E2E_DEPTH", 30))
bam, fa = os.path.join(out, "s.bam"), os.path.join(out, "s.fa")
t0 = time.time()
seed = synth.seed_for(2, 20)
rec = synth.short_read_contig(L, depth, seed)
ref = synth.make_reference(L, seed)
print(f"generated {rec.n} reads in {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
EL.write_bam_native(out, bam, "chr21", L, rec, threads=int(os.environ.get("E2E_WTHREADS", 16)))
EL.write_fasta(fa, "chr21", ref)
print(f"wrote {os.path.getsize(bam) / 1e6:.0f} MB BAM in {time.time() - t0:.1f} s", flush=True)
del rec
quiet = os.environ.get("E2E_QUIET") == "1"
for threads in os.environ.get("E2E_THREADS", "16,1").split(","):
    for rep in range(int(os.environ.get("E2E_REPS", 2))):
        for leave in os.environ.get("E2E_TEARDOWN", "0").split(","):     # DUT_CLI_TEARDOWN=1: the tool gives everything back before it leaves
            env = dict(os.environ, DUT_TIMING="1", DUT_THREADS=threads, DUT_CLI_TEARDOWN=leave.rstrip("f"))
            if leave.endswith("f"):
                env["DUT_CLI_FOREGROUND"] = "1"                            # "0f": one process, the caller waits for the exit
            def cpu_stat():
                try:
                    return {k: int(v) for k, v in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))}
                except Exception:
                    return {}
            c0 = cpu_stat()
            t0 = time.time()
            r = subprocess.run([os.environ.get("E2E_CLI", _b.CLI), "coverage", bam, "-r", fa, "-o", os.path.join(out, "o.bed")] + os.environ.get("E2E_ARGS", "").split(),
                               cwd=out, env=env, capture_output=True, text=True)
            t1 = time.time()
            dt = t1 - t0
            c1 = cpu_stat()
            import re
            st = dict(re.findall(r"wall clock at (\w+): ([0-9.]+)", r.stderr))
            if "main" in st and "exit" in st:
                print(f"    before main {float(st['main']) - t0:.3f} s, main {float(st['exit']) - float(st['main']):.3f} s, from the end of main to the caller {t1 - float(st['exit']):.3f} s"
                      + (" (the child releases in the background)" if "return" in st else " (the caller waits for the process's teardown)"), flush=True)
            if c0 and c1:
                print("    cgroup cpu.stat over the run: " + ", ".join(f"{k} +{c1[k] - c0[k]}" for k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec") if k in c0), flush=True)
            print(f"--- DUT_THREADS={threads} DUT_CLI_TEARDOWN={leave} run {rep}: {dt:.3f} s wall, rc={r.returncode}, {L / dt / 1e6:.1f} Mbase/s end to end", flush=True)
            if not quiet or rep == 0:
                print(r.stderr.strip(), flush=True)
if os.path.exists(os.path.join(out, "o.bed")):
    print(json.dumps(dict(bed_lines=sum(1 for _ in open(os.path.join(out, "o.bed"))))))

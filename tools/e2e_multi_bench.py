#!/usr/bin/env python3
"""Tooling: end-to-end `coverage` on a multi-contig BAM + .bai + FASTA, with and without the read-ahead
of the next contig (DUT_PIPELINE).  Labelled host-inclusive timing, never bench.py's `value`."""
import os, sys, time, json, subprocess, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from decodingustools_amd import synth, build as _b
import e2e_bench_lib as EL

out = os.environ.get("E2E_DIR", "/tmp/e2em")
os.makedirs(out, exist_ok=True)
n_contigs = int(os.environ.get("E2E_CONTIGS", 4))
L = int(os.environ.get("E2E_LEN", 30_000_000))
depth = float(os.environ.get("E2E_DEPTH", 30))
bam, fa = os.path.join(out, "m.bam"), os.path.join(out, "m.fa")
t0 = time.time()
contigs, refs = [], []
for i in range(n_contigs):
    seed = synth.seed_for(4, i)
    contigs.append((f"chr{i + 1}", L, synth.short_read_contig(L, depth, seed)))
    refs.append((f"chr{i + 1}", synth.make_reference(L, seed)))
print(f"generated {sum(c[2].n for c in contigs)} reads in {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
EL.write_multi_bam(out, bam, contigs, threads=int(os.environ.get("E2E_WTHREADS", 16)))
EL.write_multi_fasta(fa, refs)
print(f"wrote {os.path.getsize(bam) / 1e6:.0f} MB BAM in {time.time() - t0:.1f} s", flush=True)
del contigs, refs
digests = {}
for pipe in ("1", "0", "1", "0"):
    env = dict(os.environ, DUT_PIPELINE=pipe, DUT_TIMING=os.environ.get("DUT_TIMING", "0"))
    bed = os.path.join(out, f"o{pipe}.bed")
    t0 = time.time()
    r = subprocess.run([_b.CLI, "coverage", bam, "-r", fa, "-o", bed], cwd=out, env=env, capture_output=True, text=True)
    dt = time.time() - t0
    print(f"--- DUT_PIPELINE={pipe}: {dt:.2f} s wall, rc={r.returncode}, {n_contigs * L / dt / 1e6:.1f} Mbase/s end to end", flush=True)
    if r.stderr.strip():
        print(r.stderr.strip(), flush=True)
    digests[pipe] = hashlib.sha256(open(bed, "rb").read()).hexdigest()
print(json.dumps(dict(same_bed=digests["0"] == digests["1"], bed_lines=sum(1 for _ in open(os.path.join(out, "o1.bed"))))))

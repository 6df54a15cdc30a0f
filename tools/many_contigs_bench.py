#!/usr/bin/env python3
"""Tooling: `coverage` on a BAM with many small contigs (per-contig overheads of the file driver)."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bamio import write_bam, write_fasta
from decodingustools_amd import synth, build as _b
out = os.environ.get("E2E_DIR", "/tmp/many"); os.makedirs(out, exist_ok=True)
n = int(os.environ.get("MANY_N", 600)); L = 20000
refs = [(f"scaf{i}", L) for i in range(n)]
recs = {i: synth.short_read_contig(L, 10, 100 + i) for i in range(0, n, 2)}
bam, fa = os.path.join(out, "m.bam"), os.path.join(out, "m.fa")
write_bam(bam, refs, recs)
write_fasta(fa, [(nm, synth.make_reference(L, 5 + i)) for i, (nm, _) in enumerate(refs)])
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([_b.CLI, "coverage", bam, "-r", fa, "-o", os.path.join(out, "m.bed")], cwd=out, capture_output=True, text=True)
    dt = time.time() - t0
    print(f"run {rep}: {n} contigs in {dt:.2f} s = {dt / n * 1e3:.2f} ms per contig, rc={r.returncode} {r.stderr[-200:]}", flush=True)

#!/usr/bin/env python3
"""Tooling: BASELINE.json configs[4] end to end -- `find-y-branch` on a chrY-sized 40x BAM (+ .bai, FASTA)
with a synthetic FTDNA-shaped tree (80 000 nodes, ~2x10^5 sites), through the dut-coverage tool."""
import os, sys, time, json, random, struct, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from decodingustools_amd import synth, build as _b
import e2e_bench_lib as EL

out = os.environ.get("E2E_DIR", "/tmp/branch"); os.makedirs(out, exist_ok=True)
L = int(os.environ.get("BR_LEN", 57_227_415)); depth = float(os.environ.get("BR_DEPTH", 40))
n_nodes = int(os.environ.get("BR_NODES", 80_000))
t0 = time.time()
seed = synth.seed_for(5, 23)
rec = synth.short_read_contig(L, depth, seed)
ref = synth.make_reference(L, seed)
print(f"generated {rec.n} reads in {time.time() - t0:.1f} s", flush=True)
bam, fa, tree = os.path.join(out, "y.bam"), os.path.join(out, "y.fa"), os.path.join(out, "ytree.json")
t0 = time.time()
hdr = "@HD\tVN:1.6\tSO:coordinate\n@PG\tID:bwa\tPN:bwa\n@CO\tAS:GRCh38\n"
EL.write_bam_native(out, bam, "chrY", L, rec, header=hdr)
EL.write_single_ref_bai(bam + ".bai", rec.n)
EL.write_fasta(fa, "chrY", ref)
print(f"wrote {os.path.getsize(bam) / 1e6:.0f} MB BAM in {time.time() - t0:.1f} s", flush=True)
del rec
t0 = time.time()
import test_haplogroup as TH
rng = random.Random(7)
pool = [rng.randrange(min(2_700_000, L // 10), L - 1000) for _ in range(200_000)]
open(tree, "w").write(TH.ftdna_tree(rng, n_nodes, pool))
print(f"tree: {n_nodes} nodes, {os.path.getsize(tree) / 1e6:.0f} MB in {time.time() - t0:.1f} s", flush=True)
env = dict(os.environ, DUT_TIMING="1")
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([_b.CLI, "find-y-branch", bam, "-r", fa, os.path.join(out, "hap.tsv"), "--tree", tree, "--show-snps"], env=env, capture_output=True, text=True)
    print(f"--- run {rep}: {time.time() - t0:.2f} s wall, rc={r.returncode}", flush=True)
    print(r.stderr.strip()[-1500:], flush=True)
print(json.dumps(dict(tsv_lines=sum(1 for _ in open(os.path.join(out, "hap.tsv"))))))

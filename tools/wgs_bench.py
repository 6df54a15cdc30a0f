#!/usr/bin/env python3
"""Tooling: BASELINE.json configs[3] -- the 25 hg38 primary contigs at 30x (config-2 read model), one
after the other on one GPU: per contig generate, first pass through the module API, then the resident
contig timed for a few steps.  Prints one JSON line per contig and the whole-genome totals."""
import os, sys, time, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from decodingustools_amd import CallableOptions, CallableProfiler, ContigProfiler, Engine, process_single_contig, synth

depth = float(os.environ.get("WGS_DEPTH", 30))
steps = int(os.environ.get("WGS_STEPS", 3))
scale = float(os.environ.get("WGS_SCALE", 1.0))          # < 1: shorter contigs (rehearsal)
opt = CallableOptions()
eng = Engine(opt, 0)
eng.set_profiling(True)
d = tempfile.mkdtemp()
counter = CallableProfiler(os.path.join(d, "wgs.bed"))
tot_L = 0; tot_ms = 0.0; tot_first = 0.0; tot_reads = 0; tot_callable = 0; tot_iv = 0
t_all = time.time()
for tid, (name, L0) in enumerate(synth.HG38_PRIMARY):
    L = max(1000, int(L0 * scale))
    t0 = time.time()
    seed = synth.seed_for(4, tid)
    rec = synth.short_read_contig(L, depth, seed)
    ref = synth.make_reference(L, seed)
    t_gen = time.time() - t0
    st = ContigProfiler(name, L)
    t0 = time.time()
    process_single_contig(eng, counter, st, opt, tid, rec, ref)
    t_first = time.time() - t0
    res = eng.contig_collect()
    eng.contig_run(); eng.sync(); eng.reset_kernel_ms()
    for _ in range(steps):
        eng.contig_run()
    eng.sync()
    ms, n = eng.kernel_ms()
    step_ms = sum(ms.values()) / n
    counts = counter.get_contig_counts(name)
    tot_L += L; tot_ms += step_ms; tot_first += t_first; tot_reads += rec.n; tot_callable += counts[1]; tot_iv += res.summary.n_intervals
    print(json.dumps(dict(contig=name, L=L, reads=rec.n, gen_s=round(t_gen, 1), first_pass_s=round(t_first, 2), step_ms=round(step_ms, 3),
                          pileup_ms=round(ms["pileup"] / n, 3), gbase_s=round(L / step_ms / 1e6, 1), intervals=int(res.summary.n_intervals))), flush=True)
    del rec, ref
counter.close()
print(json.dumps(dict(total_bases=tot_L, reads=tot_reads, sum_step_ms=round(tot_ms, 2), gbase_s_resident=round(tot_L / tot_ms / 1e6, 1),
                      sum_first_pass_s=round(tot_first, 1), mbase_s_first_pass=round(tot_L / tot_first / 1e6, 1), callable_fraction=round(tot_callable / tot_L, 4),
                      intervals=tot_iv, bed_bytes=os.path.getsize(os.path.join(d, "wgs.bed")), wall_s=round(time.time() - t_all, 1))), flush=True)

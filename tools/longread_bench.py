#!/usr/bin/env python3
"""Timing of the long-read configuration (BASELINE.json configs[2] shape, scaled) -- tools only."""
import os, sys, time, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from decodingustools_amd import CallableOptions, CallableProfiler, ContigProfiler, Engine, process_single_contig, synth
L = int(os.environ.get("KB_LEN", 3_000_000))
t0 = time.time()
rec = synth.long_read_contig(L, 50, synth.seed_for(3, 23))
ref = synth.make_reference(L, synth.seed_for(3, 23))
print("gen", round(time.time() - t0, 1), "s reads", rec.n, "ops", rec.cigar.shape[0], "bases", rec.qual.shape[0], flush=True)
opt = CallableOptions()
eng = Engine(opt, 0)
counter = CallableProfiler(os.path.join(tempfile.mkdtemp(), "x.bed"))
st = ContigProfiler("chrY", L)
process_single_contig(eng, counter, st, opt, 23, rec, ref)
counter.close()
eng.set_profiling(True)
for _ in range(2): eng.contig_run()
eng.sync(); eng.reset_kernel_ms()
for _ in range(5): eng.contig_run()
eng.sync()
ms, n = eng.kernel_ms()
tot = sum(ms.values()) / n
print(json.dumps(dict(L=L, ms={k: round(v / n, 4) for k, v in ms.items()}, gbase_s=round(L / tot / 1e6, 3))))

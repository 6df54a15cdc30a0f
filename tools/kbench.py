#!/usr/bin/env python3
"""Kernel-level timing of the resident chr21-shaped contig under different CL_ABLATE settings /
library variants (tools only; not part of the product or of bench.py)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from decodingustools_amd import CallableOptions, CallableProfiler, ContigProfiler, Engine, process_single_contig, synth

L = int(os.environ.get("KB_LEN", 46_709_983))
depth = float(os.environ.get("KB_DEPTH", 30))
ablates = [int(x, 0) for x in os.environ.get("KB_ABLATES", "0").split(",")]
steps = int(os.environ.get("KB_STEPS", 10))
seed = synth.seed_for(2, 20)
rec = synth.short_read_contig(L, depth, seed, read_len=int(os.environ.get('KB_READLEN', 150)))
ref = synth.make_reference(L, seed)
opt = CallableOptions()
eng = Engine(opt, 0)
import tempfile
counter = CallableProfiler(os.path.join(tempfile.mkdtemp(), "x.bed"))
st = ContigProfiler("chr21", L)
process_single_contig(eng, counter, st, opt, 20, rec, ref)
counter.close()
eng.set_profiling(True)
for ab in ablates:
    os.environ["CL_ABLATE"] = str(ab)
    for _ in range(3):
        eng.contig_run()
    eng.sync(); eng.reset_kernel_ms()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.contig_run()
    eng.sync()
    dt = (time.perf_counter() - t0) / steps * 1e3
    ms, n = eng.kernel_ms()
    print(json.dumps(dict(ablate=ab, ms_step=round(dt, 4), **{k: round(v / n, 4) for k, v in ms.items()})), flush=True)

#!/usr/bin/env python3
"""Kernel-level timing of the resident chr21-shaped contig (depth / read-length sweeps: KB_LEN, KB_DEPTH, KB_READLEN;
DUT_QUAL_FORM=bytes for the byte forms; a library variant through DUT_CALLABLE_LIB).  Tools only; not part of the
product or of bench.py.  (The CL_ABLATE hooks of the byte forms are read once at cl_create and exist only in the tuning
build: tools/rt_ablate.sh.)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from decodingustools_amd import CallableOptions, CallableProfiler, ContigProfiler, Engine, process_single_contig, synth

L = int(os.environ.get("KB_LEN", 46_709_983))
depth = float(os.environ.get("KB_DEPTH", 30))
ablates = [0]
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

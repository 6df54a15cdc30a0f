#!/usr/bin/env python3
"""Timing of cl_site_pileup at BASELINE.json configs[4] shape (scaled by KB_LEN) -- tools only."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from decodingustools_amd import CallableOptions, Engine, synth
L = int(os.environ.get("KB_LEN", 57_227_415))
t0 = time.time()
ref = synth.make_reference(L, synth.seed_for(5, 23))
rec = synth.short_read_contig(L, 40, synth.seed_for(5, 23), with_seq=True, ref=ref, max_live_assert=0)
rng = np.random.default_rng(5)
sites = rng.choice(np.arange(1, L + 1), size=200_000, replace=False).astype(np.uint32)
print("gen", round(time.time() - t0, 1), "s reads", rec.n, flush=True)
eng = Engine(CallableOptions(), 0)
for i in range(3):
    t0 = time.perf_counter()
    hist = eng.site_pileup(20, L, L, rec, sites)
    dt = time.perf_counter() - t0
    kms, nb = eng.site_pileup_stats()
    print(json.dumps(dict(call_s=round(dt, 4), kernel_ms=round(kms, 4), gb_s=round(nb / kms / 1e6, 1), sites_hit=int((hist.sum(1) > 0).sum()), total=int(hist.sum()))), flush=True)

#!/usr/bin/env python3
"""Tooling: host-to-device rate of cl_push_reads' quality copy (pinned staging ring) for a chr21-sized tile, without
the prefetch overlap: `DUT_COPY_THREADS=n python tools/copy_bench.py`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from decodingustools_amd import CallableOptions, Engine, synth
L = int(os.environ.get("KB_LEN", 46_709_983))
rec = synth.short_read_contig(L, 30, 1)
ref = synth.make_reference(L, 1)
with Engine(CallableOptions(), 0) as eng:
    for i in range(3):
        eng.contig_begin(0, L, ref)
        t0 = time.perf_counter()
        eng.push_reads(rec.pos, rec.mapq, rec.cigar_off, rec.cigar, rec.qual_off, rec.qual)
        dt = time.perf_counter() - t0
        print(f"threads {os.environ.get('DUT_COPY_THREADS', '8')}: push {dt * 1e3:.1f} ms, {rec.qual.shape[0] / dt / 1e9:.1f} GB/s of qualities", flush=True)

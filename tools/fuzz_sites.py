#!/usr/bin/env python3
"""Tooling: time-boxed differential fuzz of cl_site_pileup (config 5) against the CPU oracle: random reads with every
CIGAR operation kind, sequences shorter than / equal to the CIGAR's query length, many operations per read, random
site lists (dense, sparse, beyond the contig, position 0), random quality gates.
    python tools/fuzz_sites.py [seconds] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from decodingustools_amd import CallableOptions, Engine, synth
from decodingustools_amd.records import ContigRecords, SEQ_CODES

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
t_end = time.time() + budget
rnd = 0
eng = Engine(CallableOptions(), 0)
while time.time() < t_end:
    seed = seed0 + rnd
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 3))
    if kind == 0:                                   # generator reads with bases following the reference
        L = int(rng.choice([3000, 20_000, 150_000]))
        ref = synth.make_reference(L, seed + 1)
        rec = synth.short_read_contig(L, float(rng.choice([3, 20, 60])), seed, read_len=int(rng.choice([50, 150, 250])),
                                      with_seq=True, ref=ref, max_live_assert=0)
    else:                                           # hand-made reads: every operation kind, odd lengths, short sequences
        L = int(rng.choice([300, 2048, 9000]))
        ref = synth.make_reference(L, seed + 1)
        n = int(rng.integers(1, 600))
        starts = np.sort(rng.integers(0, L, size=n))
        reads = []
        for i in range(n):
            nops = int(rng.choice([1, 2, 3, 5, 9, 70 if kind == 2 else 4]))
            ops = []
            for k in range(nops):
                o = str(rng.choice(list("MMM=XIDNSHP")))
                ops.append((o, int(rng.integers(1, 40 if nops < 20 else 6))))
            if not any(o in "M=X" for o, _ in ops):
                ops.append(("M", int(rng.integers(1, 30))))
            cig = "".join(f"{l}{o}" for o, l in ops)
            qlen = sum(l for o, l in ops if o in "MIS=X")
            sl = qlen if rng.random() < 0.8 else int(rng.integers(0, qlen + 1))       # a sequence shorter than the CIGAR says
            seq = "".join(rng.choice(list("ACGTN"), size=sl)) if sl else ""
            reads.append((int(starts[i]), cig, int(rng.choice([0, 5, 19, 20, 21, 60])), [30] * sl if sl else None, int(rng.choice([0, 4, 0x400])), f"r{i}", seq))
        rec = ContigRecords.from_reads(reads)
        if rec.seq4 is None:
            rec.seq_off = np.zeros(rec.n + 1, np.uint64); rec.seq4 = np.zeros(1, np.uint8)
    ns = int(rng.choice([1, 7, 200, min(5000, L)]))
    sites = rng.choice(np.arange(0, L + 40), size=min(ns, L + 40), replace=False).astype(np.uint32)     # 1-based positions; 0 and > L never match
    minq = int(rng.choice([0, 1, 20, 61]))
    ref_arg = ref if rng.random() < 0.85 else ref[: L // 2]
    exp = oracle.site_pileup(4, minq, L, ref_arg, rec, sites)
    got = eng.site_pileup(minq, L, ref_arg.shape[0], rec, sites)
    assert np.array_equal(got, exp["hist"]), f"site pileup differs in round {rnd} (seed {seed})"
    print(f"round {rnd} seed {seed} kind {kind} L {L} reads {rec.n} sites {sites.shape[0]} hits {int(got.sum())} ok", flush=True)
    rnd += 1
print(f"{rnd} rounds, no mismatch")

#!/usr/bin/env python3
"""Tooling: time-boxed fuzz of the engine's staging paths (not of the arithmetic: tools/fuzz_parity.py holds that against
the oracle).  ONE long-lived engine takes a random sequence of contigs of every shape -- short reads, indel-rich and
HiFi-like long reads, the record-shape contig, empty ones --, each pushed in random tiles (tiny staged tiles, large direct
ones, a quality prefetch in front of some, a refused tile now and then, a resident re-run), and must give what a fresh
engine gives for the same contig pushed at once: summaries and intervals equal.  Exercises what a context keeps from one
contig to the next (device buffers of the previous form, the aligned quality layout's second buffer, the staging pool).
    python tools/fuzz_tiles.py [seconds] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("DUT_VALIDATE", "1")
import numpy as np
import test_gpu_parity as T
from decodingustools_amd import CallableOptions, Engine, EngineError, synth
from decodingustools_amd.records import ContigRecords

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
t_end = time.time() + budget
opt = CallableOptions()


def push(eng, r):
    eng.push_reads(r.pos, r.mapq, r.cigar_off, r.cigar, r.qual_off, r.qual)


def contig_for(rng, seed):
    kind = int(rng.integers(0, 6))
    L = int(rng.choice([3000, 20_000, 70_000, 300_000]))
    if kind == 0:
        rec = synth.short_read_contig(L, float(rng.choice([3, 30, 60])), seed, max_live_assert=100_000)
    elif kind == 1:
        rec = synth.long_read_contig(max(L, 10_000), float(rng.choice([5, 40])), seed); L = max(L, 10_000)
    elif kind == 2:
        L, rec, _ = T.record_shapes_contig(L=int(rng.choice([20_000, 150_000])), seed=seed, n_plain=int(rng.integers(100, 4000)), short_form=False)
    elif kind == 3:
        rec = synth.adversarial_contig(L, int(rng.integers(1, 1500)), seed, max_len=int(rng.choice([50, 300, 3000])), deep=bool(rng.integers(0, 2)), overhang=False)
    elif kind == 4:
        rec = ContigRecords.empty()
    else:
        L = max(L, 20_000)
        rec = T._stacked_multi_op_reads(int(rng.integers(50, 1500)), min(3000, L // 3), min(9000, L - 200), seed, long_every=int(rng.integers(3, 15)))
    return kind, L, rec, synth.make_reference(L, seed + 1)


rnd = 0
with Engine(opt, 0) as keep:
    while time.time() < t_end:
        seed = seed0 + rnd
        rng = np.random.default_rng(seed)
        kind, L, rec, ref = contig_for(rng, seed)
        if rec.n and not ((rec.pos >= 0) & (rec.pos < L)).all():           # (a tile may only hold reads inside the contig)
            rnd += 1
            continue
        want = err = None
        try:
            with Engine(opt, 0) as fresh:
                fresh.contig_begin(int(rnd % 25), L, ref)
                if rec.n:
                    push(fresh, rec)
                want = fresh.contig_finish()
        except EngineError as e:
            err = str(e)
        # the same contig through the long-lived engine, in tiles
        cuts = sorted(set([0, rec.n] + [int(x) for x in rng.integers(0, rec.n + 1, int(rng.integers(0, 6)))])) if rec.n else [0, 0]
        got = err2 = None
        try:
            keep.contig_begin(int(rnd % 25), L, ref)
            for a, b in zip(cuts[:-1], cuts[1:]):
                if b <= a:
                    continue
                t = rec.slice(a, b)
                if rng.random() < 0.15 and t.n > 1:                        # a tile the engine refuses (unsorted), then the good one
                    bad = rec.slice(a, b)
                    bad.pos = bad.pos.copy(); bad.pos[-1] = max(0, int(bad.pos[0]) - 1) if int(bad.pos[0]) > 0 else bad.pos[-1]
                    if int(bad.pos[-1]) < int(bad.pos[-2]):
                        try:
                            push(keep, bad)
                            raise AssertionError("an unsorted tile was accepted")
                        except EngineError:
                            pass
                push(keep, t)
            got = keep.contig_finish()
            if rng.random() < 0.3:
                keep.contig_run()
                again = keep.contig_collect()
                assert again.as_dict() == got.as_dict() and np.array_equal(again.intervals, got.intervals), f"round {rnd}: re-run differs"
        except EngineError as e:
            err2 = str(e)
        if err is not None or err2 is not None:
            assert (err is None) == (err2 is None), f"round {rnd} seed {seed}: {err!r} vs {err2!r}"
        else:
            assert got.as_dict() == want.as_dict() and np.array_equal(got.intervals, want.intervals), f"round {rnd} seed {seed} kind {kind}: tiled push differs"
        print(f"round {rnd} seed {seed} kind {kind} L {L} reads {rec.n} tiles {len(cuts) - 1} {'error both' if err else 'ok'}", flush=True)
        rnd += 1
print(f"{rnd} rounds, no mismatch")

"""The record builder of the short-read form (callable_loci.hip: gen_read_recs, through the host-only hook
cl_debug_read_records) against a plain restatement of what the reference's column walk sees of one read
(mod.rs:22-37): the read is in every column of [pos, pos + bam_cigar2rlen), and the bases of its M/=/X operations that
have a quality byte are the ones tested against min_base_quality.  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

from decodingustools_amd import _lib

OPS = "MIDNSHP=X"
REF = set("MDN=X")
QRY = set("MIS=X")
MATCH = set("M=X")


def pack(cig):
    return np.asarray([(l << 4) | OPS.index(o) for l, o in cig], np.uint32)


def records(pos, cig, mapq, min_mapq, qoff, qlen, cap=4096):
    lib = _lib.load()
    words = pack(cig)
    out = np.zeros(4 * cap, np.uint32)
    n = C.c_uint32(); ph = C.c_uint32()
    st = lib.cl_debug_read_records(pos, words.ctypes.data if words.shape[0] else None, words.shape[0], mapq, min_mapq, qoff, qlen,
                                   out.ctypes.data, cap, C.byref(n), C.byref(ph))
    assert st == 0
    return [tuple(int(x) for x in out[4 * k:4 * k + 4]) for k in range(min(n.value, cap))], n.value, ph.value


def expected(pos, cig, mapq, min_mapq, qoff, qlen):
    """[(position of the run, quality offset of its first base, length)] for every maximal run of bases the column walk
    tests, the read's span, and the records that follow from them."""
    span = sum(l for l, o in cig if o in REF)
    runs = []
    x, y = pos, 0
    for l, o in cig:
        if o in MATCH:
            lq = max(0, min(l, qlen - y))
            if lq:
                runs.append((x, qoff + y, lq))
        if o in REF:
            x += l
        if o in QRY:
            y += l
    if span == 0:
        return [], span, runs
    pieces = []
    if mapq >= min_mapq:
        for (rx, rq, rl) in runs:
            off = 0
            while off < rl:
                ln = min(rl - off, 0xFFFF)
                pieces.append((rx + off, rq + off, ln))
                off += ln
    head = [pos, span, 0, mapq | 0x100]
    recs = []
    if pieces and pieces[0][0] == pos:
        head[2] = pieces[0][1] & 0xFFFFFFFF
        head[3] |= pieces[0][2] << 16
        pieces = pieces[1:]
    recs.append(tuple(head))
    for (px, pq, pl) in pieces:
        recs.append((px, 0, pq & 0xFFFFFFFF, mapq | (pl << 16)))
    return recs, span, runs


CASES = [
    (100, [(150, "M")], 60, 10, 1000, 150),
    (100, [(5, "S"), (145, "M")], 60, 10, 1000, 150),
    (100, [(3, "I"), (147, "M")], 60, 10, 7, 150),
    (100, [(10, "H"), (75, "M"), (2, "D"), (75, "M")], 60, 10, 0, 150),
    (100, [(60, "M"), (1, "I"), (30, "M"), (4, "D"), (59, "M")], 9, 10, 0, 150),          # below min_mapq: head alone
    (100, [(20, "="), (5, "X"), (30, "="), (1, "I"), (10, "X"), (2, "D"), (84, "=")], 60, 0, 123456789012, 150),
    (100, [(150, "S")], 60, 10, 0, 150),                                                     # no reference span: no record
    (100, [(4, "I")], 60, 10, 0, 4),
    (100, [(2, "D"), (148, "M")], 60, 10, 0, 148),                                           # the first run does not start at pos
    (0, [(70000, "M")], 60, 10, 5, 70000),                                                   # a run longer than 65 535 bases
    (2047, [(66000, "M"), (5, "D"), (3000, "M"), (2, "I"), (100, "M")], 60, 10, 0, 69102),
    (100, [(150, "M")], 60, 10, 0, 50),                                                      # quality string shorter than the read
    (100, [(150, "M")], 60, 10, 0, 0),                                                       # no quality string at all
    (100, [(50, "M"), (300, "N"), (50, "M"), (50, "S")], 60, 10, 0, 150),
    (100, [(7, "M"), (1, "I")] * 40, 60, 10, 0, 320),
]


@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_known_shapes(case):
    pos, cig, mapq, min_mapq, qoff, qlen = case
    want, span, runs = expected(*case)
    got, n, ph = records(*case)
    assert n == len(want) and got == want
    if runs and mapq >= min_mapq and span:
        assert ph == (runs[0][0] - (runs[0][1] - qoff)) % 16


def test_random_cigars():
    rng = np.random.default_rng(20260101)
    for it in range(3000):
        n_ops = int(rng.integers(0, 12))
        cig = []
        for _ in range(n_ops):
            o = OPS[int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 7, 8]))]
            l = int(rng.choice([1, 2, 5, 16, 17, 100, 150, 70000])) if o in "M=X" else int(rng.choice([1, 2, 5, 30, 400]))
            cig.append((l, o))
        qtotal = sum(l for l, o in cig if o in QRY)
        qlen = int(rng.choice([qtotal, qtotal, max(0, qtotal - int(rng.integers(0, 40))), 0]))
        pos = int(rng.integers(0, 1_000_000))
        mapq, min_mapq = int(rng.integers(0, 61)), int(rng.choice([0, 10, 30]))
        qoff = int(rng.choice([0, 77, 2**32 - 5, 2**33 + 11]))
        want, span, runs = expected(pos, cig, mapq, min_mapq, qoff, qlen)
        got, n, ph = records(pos, cig, mapq, min_mapq, qoff, qlen)
        assert n == len(want) and got == want, (it, pos, cig, mapq, min_mapq, qoff, qlen)
        # what the kernel relies on: a head exactly when the read spans something, runs in position order and disjoint,
        # every tested base in exactly one record
        if span:
            assert got[0][3] & 0x100 and got[0][1] == span
            if mapq >= min_mapq:
                assert sum(r[3] >> 16 for r in got) == sum(r[2] for r in runs)
            ends = [(r[0], r[0] + (r[3] >> 16)) for r in got if r[3] >> 16]
            assert all(a[1] <= b[0] for a, b in zip(ends, ends[1:]))

"""TEST INFRASTRUCTURE ONLY -- a second, independent CPU restatement of the coverage path in numpy, used by
tests/test_oracle_bruteforce.py to cross-check oracle/callable_oracle.c (which walks pileup columns over a
linked list the way htslib does) with a formulation that shares nothing with it: every read is expanded
on its own into per-position contributions, positions are classified with vectorised comparisons, runs are
found with numpy.  Only valid while the depth cap never bites (the tests keep the depth below it); the
cap rule itself is the C oracle's and the host library's business.  Parity unpinned, like the C oracle:
the reference has no fixtures for this path.

Follows: src/callable_loci/mod.rs:17-147, profilers/callable_profiler.rs:89-155,
profilers/contig_profiler.rs:47-83 and the htslib facts of SURVEY.md 8a-11 (2)-(6).
"""
import numpy as np

STATE_NAMES = ["REF_N", "CALLABLE", "NO_COVERAGE", "LOW_COVERAGE", "EXCESSIVE_COVERAGE", "POOR_MAPPING_QUALITY"]
REF_CONSUMING = (0, 2, 3, 7, 8)      # M D N = X
QUERY_CONSUMING = (0, 1, 4, 7, 8)    # M I S = X
MATCH = (0, 7, 8)


def contig(options, name, length, ref, rec):
    """options: object with the CallableOptions fields.  -> dict(bed lines, state counts, sums, arrays)."""
    # extent: the pileup emits columns up to the last covered position, even past contig_len (mod.rs:100-117)
    spans = []
    for i in range(rec.n):
        ops = rec.cigar[rec.cigar_off[i]:rec.cigar_off[i + 1]]
        spans.append(int(sum(int(c) >> 4 for c in ops if (int(c) & 15) in REF_CONSUMING)))
    spans = np.asarray(spans, dtype=np.int64) if rec.n else np.zeros(0, np.int64)
    pos = rec.pos.astype(np.int64)
    used = np.ones(rec.n, bool)
    used &= (rec.flag & 4) == 0                          # BAM_FUNMAP reads never enter the pileup
    used &= pos < length                                 # fetch((tid, 0, len)) does not yield them
    used &= spans > 0                                    # no reference span: in no column
    ends = pos + spans
    extent = int(max(length, ends[used].max())) if used.any() else length
    raw = np.zeros(extent, np.int64); qc = np.zeros(extent, np.int64); low = np.zeros(extent, np.int64)
    summed_baseq = quality_bases = summed_mapq = 0
    names = set()
    for i in np.flatnonzero(used):
        p = int(pos[i]); mq = int(rec.mapq[i])
        names.add(bytes(rec.qname[rec.qname_off[i]:rec.qname_off[i + 1]]))
        raw[p:ends[i]] += 1                              # every covering read counts, D/N columns included
        if mq <= options.max_low_mapq:
            low[p:ends[i]] += 1
        if mq < options.min_mapping_quality:
            continue
        summed_mapq += mq * int(spans[i])                # once per column of the read, deletions included
        q = rec.qual[int(rec.qual_off[i]):int(rec.qual_off[i + 1])]
        x, y = p, 0
        for c in rec.cigar[rec.cigar_off[i]:rec.cigar_off[i + 1]]:
            op, l = int(c) & 15, int(c) >> 4
            if op in MATCH:
                have = max(0, min(l, q.shape[0] - y))    # qual().get(qpos) is None past the array
                if have:
                    ok = q[y:y + have] >= options.min_base_quality
                    qc[x:x + have] += ok
                    quality_bases += int(ok.sum())
                    summed_baseq += int(q[y:y + have][ok].astype(np.int64).sum())
            if op in REF_CONSUMING: x += l
            if op in QUERY_CONSUMING: y += l
    refb = np.full(extent, ord("N"), np.uint8)           # a missing base reads as 'N' (mod.rs:79-80)
    if ref is not None:
        n = min(extent, ref.shape[0], length)
        refb[:n] = ref[:n]
    is_n = (refb | 0x20) == ord("n")
    with np.errstate(divide="ignore", invalid="ignore"):
        frac = low.astype(np.float64) / raw.astype(np.float64)
    is_low = (raw >= options.min_depth_for_low_mapq) & (raw > 0) & (frac > options.max_low_mapq_fraction)
    state = np.full(extent, 1, np.uint8)
    if options.max_depth > 0:
        state[qc > options.max_depth] = 4
    state[qc < options.min_depth] = 3
    state[is_low] = 5
    state[raw == 0] = 2
    state[is_n] = 0
    counts = [int((state == k).sum()) for k in range(6)]
    cut = np.flatnonzero(np.diff(state.astype(np.int16)) != 0) + 1 if extent else np.zeros(0, np.int64)
    starts = np.concatenate([[0], cut]) if extent else np.zeros(0, np.int64)
    stops = np.concatenate([cut, [extent]]) if extent else np.zeros(0, np.int64)
    lines = ["%s\t%d\t%d\t%s\n" % (name, s, e, STATE_NAMES[int(state[s])]) for s, e in zip(starts, stops)]
    return dict(lines=lines, state_counts=counts, raw=raw, qc=qc, low=low, state=state, extent=extent,
                n_covered_bases=int((raw > 0).sum()), summed_coverage=int(raw.sum()), summed_baseq=summed_baseq,
                summed_mapq=summed_mapq, quality_bases=quality_bases, n_reads=len(names))


def bed(results):
    """The BED text of consecutive contigs through the writer's state machine (callable_profiler.rs:39-66,
    122-155): a run is written when the next one starts and once more by finish_contig, which does not
    clear it -- so the last line of a contig appears again when the next contig's first run starts (and
    once per contig without any position in between)."""
    out = []
    cur = None
    for r in results:
        for line in r["lines"]:
            if cur is not None:
                out.append(cur)
            cur = line
        if cur is not None:                 # finish_contig
            out.append(cur)
    return "".join(out)

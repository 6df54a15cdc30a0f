"""CPU tests (no GPU): host-side logic of the product library against the oracle, and that the
C-ABI library loads and exports every symbol the headers declare."""
import ctypes as C
import functools
import os
import re

import numpy as np
import pytest

import oracle
from helpers import load_kats, make_options, contig_inputs, oracle_run
from decodingustools_amd import (CallableOptions, CallableProfiler, ContigProfiler, ContigResult, _lib,
                                 admit_reads, compare_contig_names, genome_summary, synth)
from decodingustools_amd.records import ContigRecords

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = set()
    for hdr in sorted(os.listdir(os.path.join(ROOT, "include"))):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"typedef[^;]*\(\*[^;]*;", "", text)              # function-pointer typedefs are not symbols
        declared |= set(re.findall(r"\b((?:cl|dut)_[a-z_0-9]+)\s*\(", text))
    assert len(declared) >= 64
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cl_abi_version() == 1


def test_struct_layouts_match_the_header():
    assert C.sizeof(_lib.cl_options) == 32
    assert C.sizeof(_lib.cl_contig_summary) == 14 * 8
    assert C.sizeof(_lib.cl_interval) == 12
    assert C.sizeof(_lib.dut_contig_stats) == 56
    assert C.sizeof(_lib.dut_export_meta) == 64


def test_no_gpu_means_loud_failure(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    from decodingustools_amd import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(CallableOptions())


def _intervals_from_states(state):
    if state.shape[0] == 0:
        return np.zeros((0, 3), np.uint32)
    cut = np.flatnonzero(np.diff(state.astype(np.int32)) != 0) + 1
    starts = np.concatenate([[0], cut])
    ends = np.concatenate([cut, [state.shape[0]]])
    return np.stack([starts, ends, state[starts]], axis=1).astype(np.uint32)


class _FakeSummary:
    def __init__(self, counts):
        self.state_counts = counts


def _feed_from_oracle(contigs, opt, tmp_path):
    """BED of the product's host writer fed with the oracle's per-position states."""
    out, bed_o = oracle_run(contigs, opt, str(tmp_path / "o.bed"), dump=True)
    prof = CallableProfiler(str(tmp_path / "h.bed"))
    for name, tid, length, ref, rec in contigs:
        st = out[name]["dumps"][3]
        iv = _intervals_from_states(st)
        cnt = [int((st == k).sum()) for k in range(6)]
        prof.feed_contig(name, ContigResult(summary=_FakeSummary(cnt), intervals=iv))
        assert prof.get_contig_counts(name) == out[name]["state_counts"]
    prof.close()
    return open(tmp_path / "h.bed").read(), bed_o


KATS = load_kats()


@pytest.mark.parametrize("case", KATS["cases"], ids=[c["name"] for c in KATS["cases"]])
def test_bed_writer_matches_kats(case, tmp_path):
    opt = make_options({**KATS["default_options"], **case.get("options", {})})
    contigs = []
    for i, c in enumerate(case["contigs"]):
        rec, ref = contig_inputs(c)
        contigs.append((c["name"], c.get("tid", i), c["len"], ref, rec))
    bed_h, bed_o = _feed_from_oracle(contigs, opt, tmp_path)
    assert bed_o == case["bed"]
    assert bed_h == case["bed"]


def test_bed_writer_duplicate_line_multi_contig_random(tmp_path):
    opt = make_options(dict(min_depth=2, min_depth_for_low_mapq=3))
    contigs = []
    for t in range(4):
        L = 500 + 137 * t
        rec = synth.adversarial_contig(L, 60 + 10 * t, 100 + t)
        ref = synth.make_reference(L, 9 + t, lowercase=(t == 1))
        if t == 2:
            ref[:] = ord("N")          # a contig that is one single REF_N run
        contigs.append((f"chr{t + 1}", t, L, ref, rec))
    bed_h, bed_o = _feed_from_oracle(contigs, opt, tmp_path)
    assert bed_h == bed_o
    # every contig but the last has its final line duplicated
    lines = bed_o.splitlines()
    for t in range(3):
        own = [i for i, l in enumerate(lines) if l.startswith(f"chr{t + 1}\t")]
        assert lines[own[-1]] == lines[own[-2]]


def _writer_rules(feeds):
    """callable_profiler.rs:39-66, 122-155 run position by position over the intervals' positions, restated on
    intervals for this test only: the text `dut_profiler_feed_contig` must produce."""
    names = ["REF_N", "CALLABLE", "NO_COVERAGE", "LOW_COVERAGE", "EXCESSIVE_COVERAGE", "POOR_MAPPING_QUALITY"]
    out, cur = [], None

    def write():
        if cur is not None:
            out.append(f"{cur[0]}\t{cur[1]}\t{cur[2]}\t{names[cur[3]]}\n")
    for contig, iv in feeds:
        for s, e, st in iv.tolist():
            if cur is None:
                if st == 0:
                    cur = [contig, 0, e, st]
                else:
                    if s > 0:
                        cur = [contig, 0, s, 0]; write()
                    cur = [contig, s, e, st]
            elif cur[0] == contig and cur[3] == st:
                cur[2] = e
            else:
                write(); cur = [contig, s, e, st]
        write()
    return "".join(out)


@pytest.mark.parametrize("seed", range(6))
def test_bed_text_of_long_interval_lists_is_the_same_in_chunks(seed, tmp_path):
    """Contigs with tens of thousands of intervals have their lines formatted in chunks on all host threads; the
    bytes are those of the one-line-at-a-time rules (also taken when neighbours share a state, and when the plot
    ranges are wanted)."""
    rng = np.random.default_rng(seed)
    feeds = []
    for c in range(3):
        n = int(rng.integers(5000, 60000)) if c != 1 or seed % 2 else 100
        lens = rng.integers(1, 400, size=n)
        ends = np.cumsum(lens) + (int(rng.integers(0, 50)) if seed % 3 else 0)
        starts = ends - lens
        st = rng.integers(0, 6, size=n)
        if seed != 4 or c != 2:                            # maximal runs, as the engine gives them ...
            for i in range(1, n):
                if st[i] == st[i - 1]:
                    st[i] = (st[i] + 1 + int(rng.integers(0, 5))) % 6
                    if st[i] == st[i - 1]: st[i] = (st[i] + 1) % 6
        if seed == 1 and c == 0: st[0] = 0                 # ... the first one REF_N or not, at 0 or not
        feeds.append((f"chr{c + 1}", np.stack([starts, ends, st], axis=1).astype(np.uint32)))
    want = _writer_rules(feeds)
    for plots in (False, True):
        path = str(tmp_path / f"w{int(plots)}.bed")
        prof = CallableProfiler(path)
        if plots:
            prof.enable_plots(int(max(iv[-1, 1] for _, iv in feeds)))
        for name, iv in feeds:
            prof.feed_contig(name, ContigResult(summary=_FakeSummary([0] * 6), intervals=iv))
        prof.close()
        assert open(path).read() == want, plots


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("max_depth", [0, 1, 3, 8, 500])
def test_admission_rule_matches_the_oracle_engine(seed, max_depth):
    L = 800
    rec = synth.adversarial_contig(L, 300, seed, max_len=120, deep=(seed % 3 == 0), overhang=(seed % 4 == 1))
    opt = make_options(dict(max_depth=max_depth))
    tid = seed % 3
    acc_o = oracle.accepted_reads(opt, tid, L, rec)
    o = CallableOptions(max_depth=max_depth)
    acc_h, n_names = admit_reads(o, tid, L, rec)
    # the oracle marks every appended record; the host additionally requires a reference span
    ops = rec.cigar & 15
    lens = (rec.cigar >> 4).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(np.where(np.isin(ops, [0, 2, 3, 7, 8]), lens, 0))])
    rl = cs[rec.cigar_off[1:].astype(np.int64)] - cs[rec.cigar_off[:-1].astype(np.int64)]
    assert np.array_equal(acc_h, acc_o & (rl > 0))
    # distinct names == the oracle's n_reads
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        prof = oracle.Profiler(os.path.join(d, "x.bed"))
        st, _ = oracle.process_single_contig(prof, opt, "c", tid, L, None, rec)
        prof.close()
    assert n_names == st["n_reads"]


def test_admission_unsorted_is_an_error():
    rec = ContigRecords.from_reads([(10, "5M", 60), (5, "5M", 60)])
    from decodingustools_amd import EngineError
    with pytest.raises(EngineError):
        admit_reads(CallableOptions(), 0, 100, rec)
    with pytest.raises(oracle.OracleError):
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            prof = oracle.Profiler(os.path.join(d, "x.bed"))
            try:
                oracle.process_single_contig(prof, make_options({}), "c", 0, 100, None, rec)
            finally:
                prof.close()


def test_admission_region_filter_and_funmap():
    rec = ContigRecords.from_reads([(0, "5M", 60, 30, 0x4, "u"), (3, "5M", 60, 30, 0, "a"),
                                    (50, "5M", 60, 30, 0, "beyond")])
    acc, n = admit_reads(CallableOptions(), 0, 20, rec)
    assert acc.tolist() == [False, True, False] and n == 1


def test_compare_contig_names_matches_oracle():
    names = ["chr10", "chrM", "chr2", "chrX", "chr1", "chrY", "chrUn_KI270302v1", "1", "MT", "X", "chr22",
             "chr1_KI270706v1_random", "HLA-A", "chrEBV", "2", "10", "Y", "chr+5", "chr05"]
    for a in names:
        for b in names:
            so = oracle.compare_contig_names(a, b)
            sh = compare_contig_names(a, b)
            assert (so > 0) - (so < 0) == (sh > 0) - (sh < 0), (a, b)


def test_derived_stats_and_genome_summary_match_oracle():
    rng = np.random.default_rng(5)
    stats, call, ostats = [], [], []
    names = ["chr2", "chr10", "chrX", "chr1", "chrM"]
    for nm in names:
        L = int(rng.integers(1000, 10_000_000))
        cov = int(rng.integers(0, L))
        qb = int(rng.integers(0, 30 * L))
        cp = ContigProfiler(nm, L, n_covered_bases=cov, summed_coverage=int(rng.integers(0, 40 * L)),
                            summed_baseq=int(qb * rng.uniform(15, 35)), summed_mapq=int(rng.integers(0, 60 * 30 * L)),
                            quality_bases=qb, n_reads=int(rng.integers(0, 1 << 20)))
        stats.append(cp); call.append(int(rng.integers(0, L)))
        od = dict(length=L, n_covered_bases=cp.n_covered_bases, summed_coverage=cp.summed_coverage,
                  summed_baseq=cp.summed_baseq, summed_mapq=cp.summed_mapq, quality_bases=cp.quality_bases,
                  n_reads=cp.n_reads)
        ostats.append(od)
        # per contig derived
        st = oracle.orc_contig_stats(L, cp.n_covered_bases, cp.summed_coverage, cp.summed_baseq, cp.summed_mapq,
                                     cp.quality_bases, cp.n_reads, 0)
        d = oracle.orc_contig_derived()
        oracle.lib().orc_contig_derive(C.byref(st), C.byref(d))
        got = cp.derived()
        for f in ("coverage_percent", "average_depth", "average_mapq", "average_baseq", "q30_percentage"):
            assert got[f] == getattr(d, f)
    g = genome_summary(stats, call)
    order = sorted(range(len(names)), key=functools.cmp_to_key(
        lambda i, j: oracle.compare_contig_names(names[i], names[j])))
    assert g["order"] == [names[i] for i in order] == ["chr1", "chr2", "chr10", "chrX", "chrM"]
    og = oracle.genome_summary([ostats[i] for i in order], [call[i] for i in order])
    for k, v in og.items():
        assert g[k] == v, k


@pytest.mark.parametrize("max_depth", [30, 40, 50, 70, 200])
def test_admission_fast_path_threshold_matches_the_oracle(max_depth):
    """30x short reads: about 30-45 records start within one read length of any record, so these caps sit on
    both sides of the point where the host may skip the sequential cap rule (it may only when the cap cannot bite)."""
    L = 20_000
    rec = synth.short_read_contig(L, 30, 77)
    opt = make_options(dict(max_depth=max_depth))
    acc_o = oracle.accepted_reads(opt, 1, L, rec)
    acc_h, _ = admit_reads(CallableOptions(max_depth=max_depth), 1, L, rec)
    ops = rec.cigar & 15
    lens = (rec.cigar >> 4).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(np.where(np.isin(ops, [0, 2, 3, 7, 8]), lens, 0))])
    rl = cs[rec.cigar_off[1:].astype(np.int64)] - cs[rec.cigar_off[:-1].astype(np.int64)]
    assert np.array_equal(acc_h, acc_o & (rl > 0))
    if max_depth <= 30:
        assert (acc_o & (rl > 0)).sum() < ((rec.flag & 4) == 0).sum()      # the cap did drop reads here


@pytest.mark.parametrize("case", [c for c in __import__("helpers").load_kats()["cases"] if c["name"].startswith(("KAT-7", "KAT-9", "KAT-10", "KAT-11"))],
                         ids=lambda c: c["name"].split()[0])
def test_admission_rule_on_the_cap_and_zero_span_kats(case):
    """dut_admit_reads on the hand-derived cap / zero-span cases: the accepted set is what the per-position counters of
    the fixture imply (the names of the reads that appear in columns), and equals the oracle engine's."""
    from helpers import contig_inputs, load_kats
    K = load_kats()
    opt_d = {**K["default_options"], **case.get("options", {})}
    c = case["contigs"][0]
    rec, _ = contig_inputs(c)
    acc_h, n_names = admit_reads(CallableOptions(max_depth=opt_d["max_depth"]), 0, c["len"], rec)
    acc_o = oracle.accepted_reads(make_options(opt_d), 0, c["len"], rec)
    ops = rec.cigar & 15
    lens = (rec.cigar >> 4).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(np.where(np.isin(ops, [0, 2, 3, 7, 8]), lens, 0))])
    rl = cs[rec.cigar_off[1:].astype(np.int64)] - cs[rec.cigar_off[:-1].astype(np.int64)]
    assert np.array_equal(acc_h, acc_o & (rl > 0))
    assert n_names == case["stats"][c["name"]]["n_reads"]
    # the accepted reads' spans add up to the fixture's raw depth
    raw = np.zeros(len(case["per_position"][c["name"]]["raw"]), np.int64)
    for i in np.flatnonzero(acc_h):
        raw[rec.pos[i]:rec.pos[i] + rl[i]] += 1
    assert raw.tolist() == case["per_position"][c["name"]]["raw"]

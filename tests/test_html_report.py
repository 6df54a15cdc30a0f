"""The presentation row of SURVEY 8f-4: the data of the per-contig coverage figure (bit-exact against a restatement
of callable_profiler.rs / histogram_plotter.rs, the leak of the previous contig's last line included), the SVG
this project draws from it, and summary.html with the reference's rows and number formats.  Host code only."""
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from oracle import bruteforce as BF
from oracle import report_oracle as RO
from helpers import make_options
from decodingustools_amd import CallableProfiler, ContigProfiler, synth
from decodingustools_amd.records import ContigRecords
from decodingustools_amd.report import write_html_report

STATE_ID = {n: i for i, n in enumerate(BF.STATE_NAMES)}


def _contigs():
    opt = make_options({})
    spec = [("chr2", 30_000, synth.short_read_contig(30_000, 12, 5)), ("chr1", 9_000, synth.adversarial_contig(9_000, 400, 6)),
            ("empty", 0, ContigRecords.empty()), ("chrM", 16_569, synth.short_read_contig(16_569, 40, 7)),
            ("chrN", 5_000, ContigRecords.empty())]
    out = []
    for name, L, rec in spec:
        ref = synth.make_reference(L, 11) if L else np.zeros(0, np.uint8)
        if name == "chrN":
            ref[:] = ord("N")
        r = BF.contig(opt, name, L, ref, rec)
        runs = []
        for ln in r["lines"]:
            _, s, e, st = ln.rstrip("\n").split("\t")
            runs.append((int(s), int(e), st))
        out.append((name, L, runs, r))
    return out


def _feed(prof, name, r, runs):
    iv = np.array([(s, e, STATE_ID[st]) for s, e, st in runs], np.uint32).reshape(-1, 3)
    class _R:                                     # feed_contig only needs .state_counts and .intervals
        pass
    res = _R(); res.state_counts = r["state_counts"]; res.intervals = iv
    prof.feed_contig(name, res)


def test_coverage_figure_bins_svg_and_html(tmp_path, monkeypatch):
    contigs = _contigs()
    largest = max(L for n, L, _, _ in contigs if n != "chrM")
    want = RO.coverage_plot_bins([(n, L, runs) for n, L, runs, _ in contigs], largest)
    os.makedirs(tmp_path / "out", exist_ok=True)
    prof = CallableProfiler(str(tmp_path / "out" / "x.bed"))
    prof.enable_plots(largest)
    stats, names, counts = [], [], []
    drawn = []
    for (name, L, runs, r), (stride, c, l, n, n_ranges) in zip(contigs, want):
        _feed(prof, name, r, runs)
        gs, gc, gl, gn = prof.plot_bins(name, L)
        assert gs == stride and gc.tolist() == c and gl.tolist() == l and gn.tolist() == n, name
        assert prof.finish_plot(name, L) == (n_ranges > 0), name      # callable_profiler.rs:67
        if n_ranges:
            drawn.append(name)
        st = ContigProfiler(name, L)
        for k in ("n_covered_bases", "summed_coverage", "summed_baseq", "summed_mapq", "quality_bases", "n_reads"):
            setattr(st, k, int(r[k]))
        stats.append(st); names.append(name); counts.append(r["state_counts"])
    prof.close()
    # chr2's figure: exactly its own plotted positions; chr1's also holds chr2's last line (the reference's leak)
    s0, c0, l0, n0, _ = want[0]
    r0 = contigs[0][3]
    assert sum(c0) == r0["state_counts"][1] and sum(l0) == r0["state_counts"][5] and sum(n0) == r0["state_counts"][0]
    last = contigs[0][2][-1]
    if last[2] in RO.PLOTTED:
        s1, c1, l1, n1, _ = want[1]
        r1 = contigs[1][3]
        own = r1["state_counts"][1] + r1["state_counts"][5] + r1["state_counts"][0]
        inside = sum(1 for p in range(last[0], last[1]) if p // s1 < len(c1))
        assert sum(c1) + sum(l1) + sum(n1) == own + inside
    # the drawings are well-formed SVG with rectangles; the BED's directory receives them
    assert len(drawn) >= 4
    for name in drawn:
        p = tmp_path / "out" / f"{name}_coverage.svg"
        assert p.exists(), name
        root = ET.parse(p).getroot()
        assert root.tag.endswith("svg") and len([e for e in root.iter() if e.tag.endswith("rect")]) >= 1
    # summary.html, written with the figures in the working directory
    monkeypatch.chdir(tmp_path / "out")
    html = tmp_path / "out" / "summary.html"
    write_html_report(str(html), stats, names, counts, "bwa", "GRCh38", "Illumina NovaSeq", 150)
    text = html.read_text(encoding="utf-8")
    assert "(based on first 10000 reads)" in text and "<dt>Aligner</dt><dd>bwa</dd>" in text
    order = [o for o in ["chr1", "chr2", "chrM", "chrN", "empty"]]
    pos = [text.index(f">{n}</option>") for n in order]
    assert pos == sorted(pos)                                   # the report's contig order (report.rs:339-393)
    for st, name, c in zip(stats, names, counts):
        d = st.derived()
        assert f"<tr><td>Length</td><td>{st.length} bp</td></tr>" in text
        assert f"<tr><td>Coverage Percent</td><td>{d['coverage_percent']:.2f}%</td></tr>" in text
        assert f"<tr><td>Average Depth</td><td>{d['average_depth']:.2f}×</td></tr>" in text
        assert f"<tr><td>Average MapQ</td><td>{d['average_mapq']:.1f}</td></tr>" in text
        assert f"<tr><td>Q30 Percentage</td><td>{d['q30_percentage']:.2f}%</td></tr>" in text
        assert f"<tr><td>Callable</td><td>{c[1]}</td></tr>" in text
        assert (f'<img src="{name}_coverage.svg"' in text) == (name in drawn)
    assert text.count('class="tab-panel') == 5 and text.count('class="tab-panel active"') == 1


def test_coverage_figure_known_answer(tmp_path):
    """Worked by hand from callable_profiler.rs:39-84 and histogram_plotter.rs:74-101, 412-440.
    Contig A (10 bp): REF_N [0,2), CALLABLE [2,10).  Contig B (6 bp): POOR_MAPPING_QUALITY [0,6).  chrM (3 bp):
    NO_COVERAGE [0,3).  largest (non-chrM) = 10 -> stride 1; chrM: stride ceil(16569/200) = 83.
      figure A: lines written while A is processed: REF_N [0,2) (at position 2), CALLABLE [2,10) (finish) -> 11 bins
      figure B: A's pending CALLABLE [2,10) is written again when B starts -> bins 2..6 of B's 7 (7..9 fall outside),
                then POOR_MAPPING_QUALITY [0,6) at finish
      figure chrM: B's pending line once more: positions 0..5 all fall into bin 0 of chrM's single bin (3/83 + 1 = 1);
                   chrM's own NO_COVERAGE line is not a plotted state."""
    runs = [("A", 10, [(0, 2, "REF_N"), (2, 10, "CALLABLE")]), ("B", 6, [(0, 6, "POOR_MAPPING_QUALITY")]), ("chrM", 3, [(0, 3, "NO_COVERAGE")])]
    want = [(1, [0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0], [0] * 11, [1, 1] + [0] * 9),
            (1, [0, 0, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1, 0], [0] * 7),
            (83, [0], [6], [0])]
    got_oracle = RO.coverage_plot_bins(runs, 10)
    assert [(s, c, l, n) for s, c, l, n, _ in got_oracle] == want
    prof = CallableProfiler(str(tmp_path / "k.bed"))
    prof.enable_plots(10)
    for (name, L, rr), (stride, c, l, n) in zip(runs, want):
        class _R:
            pass
        res = _R(); res.state_counts = [0] * 6; res.intervals = np.array([(s, e, STATE_ID[st]) for s, e, st in rr], np.uint32).reshape(-1, 3)
        prof.feed_contig(name, res)
        gs, gc, gl, gn = prof.plot_bins(name, L)
        assert (gs, gc.tolist(), gl.tolist(), gn.tolist()) == (stride, c, l, n), name
        assert prof.finish_plot(name, L)
    prof.close()
    assert open(tmp_path / "k.bed").read() == ("A\t0\t2\tREF_N\nA\t2\t10\tCALLABLE\nA\t2\t10\tCALLABLE\nB\t0\t6\tPOOR_MAPPING_QUALITY\n"
                                               "B\t0\t6\tPOOR_MAPPING_QUALITY\nchrM\t0\t3\tNO_COVERAGE\n")

"""include/dut_report.h against oracle/report_oracle.py: BamStats, platform inference, aligner /
reference-build detection, serde_json-style f64 text and the CoverageOutput JSON (SURVEY.md 8f-2,
8f-4).  Host-only: runs without a GPU."""
import json
import os
import random
import struct

import numpy as np
import pytest

from decodingustools_amd import report as R
from decodingustools_amd.records import ContigRecords
from oracle import report_oracle as O

# ---- known answers, derived by hand from platform_inference.rs:17-200 ---------------------------
QNAME_KATS = [
    # name, platform, instrument, flow cell
    ("A00123:123:HXXXYDRXX:1:1101:1000:1000", "Illumina", "A00123", "HXXXYDRXX"),
    ("m64023e_230414_133043/1/ccs", "PacBio", "m64023e", None),
    ("m84011_220902_175841_s1/262536/ccs", "PacBio", "m84011", None),
    ("0a1b2c3d-4e5f-6a7b-8c9d-0e1f2a3b4c5d", "Nanopore", "0a1b2c3d", None),
    # > 30 bytes, has '_', contains "ch" and "read": the second Nanopore rule; instrument = prefix before '_'
    ("runid_ch112_read4521_strand_template_x", "Nanopore", "runid", None),
    ("V300012345L1C001R00100000001", "MGI", "V300012345", "L1C001"),
    # the first 'L' of "CL100..." is the split point (platform_inference.rs:177-186)
    ("CL100012345L1C001R001_1", "MGI", "C", "L100012345L1C001"),
    # >= 6 colons, instrument starts with 'V', third field starts with 'L': MGI through the colon rule
    ("VX1:7:L01:C001:R001:12:34", "MGI", "VX1", "7"),
    # 15 bytes or fewer never reach the MGI rules; < 6 colons is not Illumina
    ("V3000123L1C1R1", "Unknown", None, None),
    ("read/1", "Unknown", None, None),
    ("SRR1234567.1", "Unknown", None, None),
    # 'm' + '/' but no '_' in the movie name: not PacBio
    ("m123/45/ccs", "Unknown", None, None),
    # UUID with an upper-case hex digit is still hex; a non-hex letter is not
    ("0A1B2C3D-4E5F-6A7B-8C9D-0E1F2A3B4C5D", "Nanopore", "0A1B2C3D", None),
    ("0g1b2c3d-4e5f-6a7b-8c9d-0e1f2a3b4c5d", "Unknown", None, None),
]


@pytest.mark.parametrize("name,platform,instrument,flow_cell", QNAME_KATS)
def test_qname_known_answers(name, platform, instrument, flow_cell):
    for impl_detect in (lambda q: R.detect_platform_from_qname(q.encode()), O.detect_platform_from_qname):
        assert impl_detect(name) == platform
    if platform == "Unknown":
        return
    got = R.parse_read_name(platform, name.encode())
    assert got is not None
    assert got[0].decode() == instrument
    assert (got[1].decode() if got[1] is not None else None) == flow_cell


def _oracle_parse(platform, q):
    if platform == "Illumina":
        return O.parse_illumina_read_name(q)
    if platform == "PacBio":
        r = O.parse_pacbio_read_name(q); return (r, None) if r is not None else None
    if platform == "Nanopore":
        r = O.parse_nanopore_read_name(q); return (r, None) if r is not None else None
    if platform == "MGI":
        return O.parse_mgi_read_name(q)
    return None


def _random_names(rng, n):
    hexd = "0123456789abcdef"
    out = []
    for _ in range(n):
        k = rng.randrange(9)
        if k == 0:
            s = ":".join([rng.choice(["A00", "D1", "K00", "VH0", "E12", "G3", "CL9", "N7", "x"]) + str(rng.randrange(1000)), str(rng.randrange(300)),
                          rng.choice(["HXX", "L01", "LQ", "H7"]) + str(rng.randrange(99))] + [str(rng.randrange(3000)) for _ in range(rng.randrange(1, 6))])
        elif k == 1:
            s = "m" + rng.choice(["64", "84", "54", "99"]) + str(rng.randrange(1000)) + rng.choice(["_", "", "e_"]) + "230414_133043" + rng.choice(["/", "", "/7/"]) + rng.choice(["ccs", "1", ""])
        elif k == 2:
            parts = ["".join(rng.choice(hexd + ("G" if rng.random() < 0.05 else "")) for _ in range(l)) for l in (8, 4, 4, 4, rng.choice([11, 12, 14]))]
            s = "-".join(parts[: rng.choice([4, 5, 5, 5])]) + rng.choice(["", "", "_x", "-ff"])
        elif k == 3:
            s = rng.choice(["V300", "v300", "E100", "CL100", "G400", "G99", "G98", "V30"]) + "".join(rng.choice("0123456789LCR") for _ in range(rng.randrange(4, 28)))
        elif k == 4:
            s = "".join(rng.choice("abchred_-0123456789") for _ in range(rng.randrange(1, 48)))
        elif k == 5:
            s = "".join(rng.choice("mLCRVEG:/_-019") for _ in range(rng.randrange(1, 40)))
        elif k == 6:
            s = "channel_" + str(rng.randrange(512)) + "_read_" + "".join(rng.choice(hexd) for _ in range(rng.randrange(2, 30)))
        elif k == 7:
            s = "é" * rng.randrange(1, 4) + "".join(rng.choice("V300:L-_/mch read") for _ in range(rng.randrange(0, 40)))
        else:
            s = ""
        out.append(s)
    return out


def test_qname_rules_random_differential():
    rng = random.Random(11)
    seen = set()
    for q in _random_names(rng, 30000):
        if not q.isascii():
            # the reference's `&qname[..5]` panics when byte 5 is not a character boundary and its
            # to_uppercase folds non-ASCII letters; the product is byte-wise: compare ASCII names only
            continue
        pf = O.detect_platform_from_qname(q)
        assert R.detect_platform_from_qname(q.encode()) == pf, q
        seen.add(pf)
        for platform in ("Illumina", "PacBio", "Nanopore", "MGI"):
            exp = _oracle_parse(platform, q)
            got = R.parse_read_name(platform, q.encode())
            if exp is None:
                assert got is None, (platform, q)
            else:
                assert got is not None, (platform, q)
                assert got[0].decode() == exp[0] and (got[1].decode() if got[1] is not None else None) == exp[1], (platform, q)
    assert seen == {"Illumina", "PacBio", "Nanopore", "MGI", "Unknown"}


def test_infer_specific_platform_table():
    cases = [("PacBio", "m84011", "PacBio Revio"), ("PacBio", "m64023e", "PacBio Sequel II/IIe"), ("PacBio", "m54001", "PacBio Sequel"),
             ("PacBio", "m1", "PacBio"), ("PacBio", None, "PacBio"), ("Nanopore", "abc", "Oxford Nanopore"), ("Nanopore", None, "Oxford Nanopore"),
             ("MGI", "V300012345", "MGI DNBSEQ/MGISEQ-2000"), ("MGI", "E100", "MGI MGISEQ-200"), ("MGI", "CL100", "MGI MGISEQ-T7"),
             ("MGI", "G400x", "MGI DNBSEQ-G400"), ("MGI", "G99", "MGI MGISEQ-T1"), ("MGI", "C", "MGI DNBseq"), ("MGI", None, "MGI DNBseq"),
             ("Illumina", "A00123", "NovaSeq"), ("Illumina", "d1", "HiSeq 2500"), ("Illumina", "J9", "HiSeq 3000"), ("Illumina", "K1", "HiSeq 4000"),
             ("Illumina", "E1", "HiSeq X"), ("Illumina", "NB5", "NextSeq"), ("Illumina", "M0", "MiSeq"), ("Illumina", "VH0", "NovaSeq X"),
             ("Illumina", "FS1", "iSeq"), ("Illumina", "ST-E", "Unknown Illumina"), ("Illumina", "", "Unknown Illumina"),
             ("Illumina", None, "Unknown Illumina"), ("Unknown", "A1", "Unknown"), ("Unknown", None, "Unknown")]
    for platform, top, want in cases:
        assert O.infer_specific_platform(platform, top) == want
        assert R.infer_specific_platform(platform, top) == want


HEADERS = [
    ("@HD\tVN:1.6\n@PG\tID:bwa-mem2\tPN:bwa-mem2\n", "BWA-MEM2", "Unknown"),
    ("@HD\tVN:1.6\n@PG\tID:bwa\tPN:bwa\tCL:bwa mem ref.fa\n@SQ\tSN:chr1\tLN:248956422\n", "BWA", "GRCh38"),
    ("@PG\tID:minimap2\tPN:minimap2\n@SQ\tSN:chr1\tLN:248387328\tM5:e469247288ceb332aee524caec92bb22\n", "minimap2", "T2T-CHM13v2.0"),
    ("@PG\tID:pbmm2\tCL:pbmm2 align\n@SQ\tSN:1\tLN:249250621\n", "pbmm2", "GRCh37"),
    ("@PG\tID:Bowtie2\n@SQ\tSN:chr1\tLN:1000\tAS:GRCh37\n", "Bowtie2", "GRCh37"),
    ("@PG\tID:STAR\n@SQ\tSN:chr1\tLN:5\tUR:/refs/chm13v2.fa\n", "STAR", "T2T-CHM13v2.0"),
    ("@PG\tID:samtools\tCL:samtools sort; upstream BWA\n", "BWA", "Unknown"),
    ("@PG\tID:x\tCL:run_minimap2.sh\n@CO\tGCA_000001405.15\n", "minimap2", "GRCh38"),
    # GCA_000001405.15 contains GCA_000001405.1, but GRCh38 is tested first; .14 hits the GRCh37 pattern
    ("@CO\tGCA_000001405.14\n", "Unknown", "GRCh37"),
    ("@PG\tID:novoalign\n@CO\tstarting\n", "STAR", "Unknown"),
    ("@PG\tID:novoalign\n@SQ\tSN:chr1\tLN:248387328\n", "Unknown", "Unknown"),
    ("@SQ\tSN:chr10\tLN:249250621\n", "Unknown", "GRCh37"),        # "SN:1" matches inside "SN:chr10"? no: "SN:1" needs the literal
    ("", "Unknown", "Unknown"),
]


def test_aligner_and_reference_build():
    for text, aligner, build in HEADERS:
        assert O.detect_aligner(text) == aligner, text
        assert R.detect_aligner(text.encode()) == aligner, text
        assert R.reference_build(text.encode()) == O.reference_build(text), text
    # hand-checked subset of the builds
    assert [O.reference_build(t) for t, _, _ in HEADERS[:6]] == ["Unknown", "GRCh38", "T2T-CHM13v2.0", "GRCh37", "GRCh37", "T2T-CHM13v2.0"]
    assert O.reference_build(HEADERS[11][0]) == "Unknown"


F64_KATS = [(0.0, "0.0"), (-0.0, "-0.0"), (100.0, "100.0"), (60.0, "60.0"), (2.5, "2.5"), (1e15, "1000000000000000.0"),
            (1e16, "1e16"), (1e-5, "0.00001"), (1e-6, "1e-6"), (0.1 + 0.2, "0.30000000000000004"), (1 / 3, "0.3333333333333333"),
            (123456789012345680.0, "1.2345678901234568e17"), (5e-324, "5e-324"), (1.7976931348623157e308, "1.7976931348623157e308"),
            (1234.5678, "1234.5678"), (0.001234, "0.001234"), (1.5e-7, "1.5e-7"), (-12.5, "-12.5"), (float("nan"), "null"), (float("inf"), "null")]


def test_f64_text_known_answers_and_random():
    for v, want in F64_KATS:
        assert O.format_f64(v) == want
        assert R.format_f64(v) == want
    rng = random.Random(5)
    for _ in range(40000):
        k = rng.random()
        if k < 0.3: v = struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0]
        elif k < 0.6: v = rng.uniform(0, 100)
        elif k < 0.8: v = rng.randint(0, 10 ** rng.randint(1, 18)) / rng.randint(1, 10 ** rng.randint(0, 9))
        else: v = rng.randint(0, 10 ** 6) * 10.0 ** rng.randint(-12, 20)
        a, b = O.format_f64(v), R.format_f64(v)
        assert a == b, repr(v)
        if a != "null":
            assert float(a) == v and ("." in a or "e" in a)


def _random_records(rng, n, family):
    recs = []
    names = _random_names(rng, 400)
    fam = {"illumina": lambda: f"{rng.choice(['A00123', 'A00123', 'K0077', 'VH01'])}:{rng.randrange(3)}:{rng.choice(['HXXX', 'HYYY'])}:1:1101:{rng.randrange(9999)}:{rng.randrange(9999)}",
           "pacbio": lambda: f"{rng.choice(['m64023e', 'm64023e', 'm84011'])}_230414_133043/{rng.randrange(99999)}/ccs",
           "mixed": lambda: rng.choice(names)}[family]
    for i in range(n):
        flag = 0
        if rng.random() < 0.7:
            flag |= 0x1
            if rng.random() < 0.8: flag |= 0x2
            flag |= 0x40 if rng.random() < 0.5 else 0x80
        if rng.random() < 0.05: flag |= 0x100
        if rng.random() < 0.05: flag |= 0x800
        if rng.random() < 0.03: flag |= 0x4
        l_seq = rng.choice([150, 150, 150, 151, 100, 0, rng.randrange(1, 30000)])
        tlen = rng.choice([0, 350, 350, -350, 412, -2_000_000_000, rng.randrange(-900, 900)])
        q = fam().encode()
        if family == "mixed" and rng.random() < 0.02:
            q = b"\xff\xfe" + q[:10]              # not UTF-8: counted as a read, no platform (bam_stats.rs:81)
        if not q:
            q = b"*"
        recs.append((flag, l_seq, q[:250], tlen))
    return recs


def _check_stats(bs, ob):
    assert bs.read_count() == ob.read_count
    assert bs.average_read_length() == ob.average_read_length()
    cands = ob.modal_read_length_candidates()
    assert bs.modal_read_length() == min(cands)                      # documented tie rule
    assert bs.get_primary_platform() in ob.primary_platform_candidates()
    assert bs.infer_platform() in ob.infer_platform_candidates()
    gs, og = bs.get_stats(), ob.get_stats()
    assert set(gs) == set(og)
    for k, v in og.items():
        if isinstance(v, list):
            assert gs[k] == min(v), k
        else:
            assert gs[k] == v, k


@pytest.mark.parametrize("family,n,max_samples", [("illumina", 3000, 10000), ("pacbio", 500, 200), ("mixed", 6000, 5000), ("mixed", 0, 10)])
def test_bam_stats_stream_differential(family, n, max_samples):
    rng = random.Random(hash((family, n)) & 0xFFFF)
    recs = [r for r in _random_records(rng, n, family) if family != "mixed" or r[2].isascii() or r[2].startswith(b"\xff")]
    bs = R.BamStats(max_samples)
    ob = O.BamStats(max_samples)
    hdr = "@HD\tVN:1.6\n@PG\tID:bwa\n@SQ\tSN:chr1\tLN:248956422\n"
    bs.set_header(hdr.encode()); ob.set_header(hdr)
    more = True
    for fl, ls, q, tl in recs:
        more = bs.add(fl, ls, q, tl)
        if not more:
            break
    ob.collect(recs)
    assert bs.aligner() == ob.aligner == "BWA" and bs.reference_build() == ob.reference_build == "GRCh38"
    _check_stats(bs, ob)
    if family == "illumina":
        assert bs.infer_platform() == "NovaSeq" and bs.get_primary_platform() == "Illumina"
    if family == "pacbio":
        assert bs.infer_platform() == "PacBio Sequel II/IIe"
    if n == 0:
        assert bs.infer_platform() == "Unknown" and bs.average_read_length() == 0 and bs.get_stats() == {}


def test_bam_stats_from_a_file(tmp_path):
    """collect_stats on a BAM: the first max_samples records of the file in file order, mapped or
    not, across contigs and into the unmapped tail."""
    from bamio import write_bam
    rng = random.Random(3)
    per_tid, tl, stream = {}, {}, []
    for tid, n in ((0, 40), (2, 25)):
        recs = _random_records(rng, n, "illumina" if tid == 0 else "mixed")
        recs = [r for r in recs if r[2].isascii()]
        reads = []
        for i, (fl, ls, q, t) in enumerate(recs):
            ls = min(ls, 300)
            reads.append((10 * i, f"{ls}M" if ls else "5M", 60, ([30] * ls) if ls else None, fl, q))
            stream.append((fl, ls, q, t))
        per_tid[tid] = ContigRecords.from_reads(reads)
        tl[tid] = [r[3] for r in recs]
    tail = [(b"A00123:1:HXXX:1:1101:5:5", 0x4D, 150, 0), (b"unm2", 0x4, 0, 0)]
    stream += [(fl, ls, q, t) for q, fl, ls, t in tail]
    bam = str(tmp_path / "s.bam")
    hdr = "@HD\tVN:1.6\tSO:coordinate\n@PG\tID:minimap2\tPN:minimap2\n"
    write_bam(bam, [("chr1", 100000), ("chr2", 5000), ("chrM", 16569)], per_tid, header_text=hdr, tlen=tl, unmapped_tail=tail, block_every=7)
    for max_samples in (10000, 50, 1, 0):
        bs = R.BamStats(max_samples).collect_stats(bam)
        ob = O.BamStats(max_samples); ob.set_header(hdr + "@SQ\tSN:chr1\tLN:100000\n@SQ\tSN:chr2\tLN:5000\n@SQ\tSN:chrM\tLN:16569\n"); ob.collect(stream)
        assert bs.aligner() == "minimap2" and bs.reference_build() == "Unknown"
        _check_stats(bs, ob)
    assert R.BamStats(10000).collect_stats(bam).read_count() == sum(1 for fl, _, _, _ in stream if not fl & 0x900)
    with pytest.raises(Exception, match="Failed to collect BAM stats"):
        R.BamStats(10).collect_stats(str(tmp_path / "missing.bam"))


class _S:
    pass


def _rand_stats(rng, L):
    s = dict(length=L, n_covered_bases=0, summed_coverage=0, summed_baseq=0, summed_mapq=0, quality_bases=0, n_reads=0)
    if L and rng.random() < 0.85:
        s["n_covered_bases"] = rng.randint(1, L)
        s["summed_coverage"] = s["n_covered_bases"] * rng.randint(1, 60) + rng.randint(0, 1000)
        if rng.random() < 0.9:
            s["quality_bases"] = rng.randint(1, s["summed_coverage"])
            s["summed_baseq"] = int(s["quality_bases"] * rng.choice([rng.uniform(2, 45), 30.0, 20.0, 25.0]))
            s["summed_mapq"] = s["quality_bases"] * rng.randint(0, 70) + rng.randint(0, 50)
        s["n_reads"] = rng.randint(0, 2 ** 31)
    return s


def _counts(rng, L):
    cuts = sorted(rng.randint(0, L) for _ in range(5))
    e = [0] + cuts + [L]
    return [e[i + 1] - e[i] for i in range(6)]


def _as_obj(d):
    o = _S()
    o.__dict__.update(d)
    return o


def test_coverage_output_json_text():
    rng = random.Random(9)
    hg = ["chr" + str(i) for i in range(1, 23)] + ["chrX", "chrY", "chrM"]
    name_sets = [hg, ["1", "2", "10", "X", "MT", "GL000207.1"], ["chrUn_KI270742v1", "chr1_KI270706v1_random", "chrEBV", "chr2", "HLA-A*01:01"],
                 ['we"ird\\name', "tab\there", "ctl\x01x", "naïve", "chr9"], [], ["chrM"]]
    for names in name_sets:
        for _ in range(6):
            order = list(names); rng.shuffle(order)
            stats = [_rand_stats(rng, rng.choice([0, 1, 16569, 46709983, 248956422, rng.randint(1, 10 ** 6)])) for _ in order]
            counts = [_counts(rng, s["length"]) for s in stats]
            meta = dict(aligner=rng.choice(["BWA", "Unknown", "minimap2"]), reference_build="GRCh38", platform='Nova"Seq',
                        read_length=rng.randint(0, 20000), bed="out/callable regions.bed", html="summary.html",
                        plots=rng.choice([[], ["chr1_coverage.svg", "chrM_coverage.svg"]]))
            want = O.coverage_output_json(stats, order, counts, meta["aligner"], meta["reference_build"], meta["platform"], meta["read_length"],
                                          meta["bed"], meta["html"], meta["plots"])
            got = R.coverage_output_json([_as_obj(s) for s in stats], order, counts, meta["aligner"], meta["reference_build"], meta["platform"],
                                         meta["read_length"], meta["bed"], meta["html"], meta["plots"])
            assert got == want
            parsed = json.loads(got)                                   # and it is valid JSON with the reference's field order
            assert list(parsed) == ["export", "files"]
            assert list(parsed["export"]) == ["summary", "contigs", "quality_metrics", "total_unique_reads"]
            assert list(parsed["export"]["summary"]) == ["aligner", "reference_build", "sequencing_platform", "read_length", "total_bases",
                                                         "callable_bases", "callable_percentage", "average_depth", "contigs_analyzed"]
            assert list(parsed["files"]) == ["bed_file", "summary_html", "coverage_plots"]
            for c in parsed["export"]["contigs"]:
                assert list(c) == ["name", "length", "unique_reads", "coverage_percent", "average_depth", "covered_bases", "total_bases",
                                   "quality_stats", "state_distribution"]
    # a fixed small document, written out by hand from the structs (coverage.rs:26-45,113-248; api/coverage.rs:134-145)
    s = dict(length=20, n_covered_bases=12, summed_coverage=22, summed_baseq=420, summed_mapq=960, quality_bases=14, n_reads=3)   # KAT-1
    got = R.coverage_output_json([_as_obj(s)], ["chrT"], [[2, 2, 8, 6, 0, 2]], "Unknown", "Unknown", "Unknown", 8, "callable_regions.bed", "summary.html")
    assert got == open(os.path.join(os.path.dirname(__file__), "golden", "summary_kat1.json")).read()

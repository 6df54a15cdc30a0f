"""GPU parity tests (-m gpu): the HIP engine, called through the C ABI, against the CPU oracle on
the same inputs.  Integer / byte work: everything must be bit-exact."""
import json
import os
import sys

import numpy as np
import pytest

import oracle
from helpers import contig_inputs, load_kats, make_options, oracle_run
from decodingustools_amd import (CallableOptions, CallableProfiler, ContigProfiler, Engine, admit_reads,
                                 process_single_contig, synth)
from decodingustools_amd.records import ContigRecords

pytestmark = pytest.mark.gpu
KATS = load_kats()

# The engine has two kinds of pileup kernel: the pass-bit form (default: the base-quality test is taken on the host, the
# device counts rows of bits) and the byte forms of rounds 1-3 (DUT_QUAL_FORM=bytes at cl_create: the quality bytes go to
# the device; records for short reads, the run table for long ones).  Every test runs in the default form; the tests
# named here run in the byte forms too (or only there: what they exercise exists only in those).
BYTES_TOO = {
    "test_kats_on_gpu", "test_adversarial_contigs", "test_short_reads_2mb_30x", "test_long_reads_indel_rich",
    "test_deep_pileup_uses_32bit_counters", "test_record_form_covers_every_read_shape",
    "test_run_table_with_one_base_runs_needs_the_second_sizing_pass", "test_run_table_with_truncated_qualities_and_long_runs",
    "test_hifi_like_long_match_runs_and_truncated_qualities", "test_quality_prefetch_is_claimed_by_the_matching_tile_and_harmless_otherwise",
    "test_tiles_of_mixed_sizes_small_staged_large_direct", "test_a_long_read_shaped_tile_in_a_short_read_contig",
    "test_long_reads_with_eqx_cigars", "test_records_through_small_pinned_buffers", "test_tiled_pushes_on_a_long_lived_engine",
    "test_operation_parallel_variant_with_16bit_and_32bit_counters", "test_deeper_short_read_data_keeps_the_8bit_counters",
    "test_a_refused_tile_leaves_the_contig_as_it_was", "test_long_reads_in_tiles_with_a_refused_tile_between",
    "test_an_error_behind_a_quality_prefetch_leaves_nothing_in_flight", "test_outlier_spans_do_not_widen_every_window",
}
BYTES_TOO.add("test_one_long_indel_rich_read_inside_a_short_read_contig")
BYTES_ONLY = {"test_run_table_through_small_pinned_buffers", "test_run_table_window_whose_last_piece_belongs_to_the_first_read"}


def pytest_generate_tests(metafunc):
    if "qual_form" in metafunc.fixturenames:
        name = metafunc.function.__name__
        forms = ["bytes"] if name in BYTES_ONLY else (["bits", "bytes"] if name in BYTES_TOO else ["bits"])
        metafunc.parametrize("qual_form", forms)


@pytest.fixture(autouse=True)
def qual_form(request, monkeypatch):
    """DUT_QUAL_FORM for the contexts (and child processes) of this test."""
    form = getattr(request, "param", "bits")
    if form == "bytes":
        monkeypatch.setenv("DUT_QUAL_FORM", "bytes")
    else:
        monkeypatch.delenv("DUT_QUAL_FORM", raising=False)
    return form
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _opts(d):
    o = make_options(d)
    return CallableOptions(o.min_depth, o.max_depth, o.min_mapping_quality, o.min_base_quality,
                           o.min_depth_for_low_mapq, o.max_low_mapq, o.max_low_mapq_fraction)


def engine_run(contigs, opt, bed_path, dump=False):
    """The product path: process_single_contig per contig in tid order on one Engine."""
    res = {}
    with Engine(opt, 0) as eng:
        counter = CallableProfiler(bed_path)
        for name, tid, length, ref, rec in contigs:
            st = ContigProfiler(name, length)
            process_single_contig(eng, counter, st, opt, tid, rec, ref)
            entry = dict(stats=st)
            if dump:
                # the resident contig is the accepted subset; dump its per-position counters
                extent = eng.contig_collect().summary.extent
                entry["dumps"] = eng.debug_depths(int(extent)) + (int(extent),)
            res[name] = entry
        for name in res:
            res[name]["state_counts"] = counter.get_contig_counts(name)
        counter.close()
    return res, open(bed_path).read()


def _diag(tag, name, exp, got):
    os.makedirs(OUT, exist_ok=True)
    bad = np.flatnonzero(exp != got)
    with open(os.path.join(OUT, f"mismatch_{tag}.json"), "a") as f:
        json.dump(dict(array=name, n_bad=int(bad.shape[0]), first=bad[:40].tolist(),
                       exp=exp[bad[:40]].tolist(), got=got[bad[:40]].tolist()), f)
        f.write("\n")


def compare(contigs, opt_dict, tmp_path, tag, dump=True):
    oo = make_options(opt_dict)
    opt = _opts(opt_dict)
    o_res, o_bed = oracle_run(contigs, oo, str(tmp_path / "o.bed"), dump=dump)
    g_res, g_bed = engine_run(contigs, opt, str(tmp_path / "g.bed"), dump=dump)
    ok = True
    for name, _, length, _, _ in contigs:
        so, sg = o_res[name]["stats"], g_res[name]["stats"]
        if dump:
            ro, qo, lo, sto, eo = o_res[name]["dumps"]
            rg, qg, lg, stg, eg = g_res[name]["dumps"]
            assert eg == max(eo, length), (eg, eo)
            for nm, a, b in (("raw", ro, rg[:eo]), ("low", lo, lg[:eo]), ("qc", qo, qg[:eo]), ("state", sto, stg[:eo])):
                if not np.array_equal(a, b):
                    _diag(tag, f"{name}.{nm}", a, b)
                    ok = False
        for k in ("n_covered_bases", "summed_coverage", "summed_baseq", "summed_mapq", "quality_bases", "n_reads"):
            assert so[k] == getattr(sg, k), (name, k, so[k], getattr(sg, k))
        assert o_res[name]["state_counts"] == g_res[name]["state_counts"], name
        for k, v in so["derived"].items():
            assert sg.derived()[k] == v, (name, k)
    assert ok, f"per-position mismatch, see gpurun_out/mismatch_{tag}.json"
    if g_bed != o_bed:
        os.makedirs(OUT, exist_ok=True)
        open(os.path.join(OUT, f"bed_{tag}_oracle.bed"), "w").write(o_bed)
        open(os.path.join(OUT, f"bed_{tag}_gpu.bed"), "w").write(g_bed)
    assert g_bed == o_bed
    return o_res, g_res


@pytest.mark.parametrize("case", KATS["cases"], ids=[c["name"] for c in KATS["cases"]])
def test_kats_on_gpu(case, tmp_path):
    opt = {**KATS["default_options"], **case.get("options", {})}
    contigs = []
    for i, c in enumerate(case["contigs"]):
        rec, ref = contig_inputs(c)
        contigs.append((c["name"], c.get("tid", i), c["len"], ref, rec))
    _, g = compare(contigs, opt, tmp_path, "kat")
    assert open(tmp_path / "g.bed").read() == case["bed"]


ADV_OPTS = [
    dict(min_depth=2, min_depth_for_low_mapq=3),
    dict(),                                                            # CLI defaults
    dict(min_depth=1, max_depth=8, min_depth_for_low_mapq=2, max_low_mapq=2, max_low_mapq_fraction=0.25),
    dict(min_base_quality=0, min_mapping_quality=0, max_depth=0),
    dict(min_base_quality=128, min_depth=1),
    dict(min_base_quality=200, min_depth=1, max_low_mapq_fraction=0.0),
    dict(min_base_quality=255, min_mapping_quality=255, max_low_mapq=255, max_low_mapq_fraction=0.999),
    dict(max_depth=3, min_depth=2, max_low_mapq_fraction=1.0),
]


@pytest.mark.parametrize("seed", range(8))
def test_adversarial_contigs(seed, tmp_path):
    L = [777, 2048, 2049, 4096, 5000, 6143, 1, 300][seed]
    n = [200, 500, 500, 900, 1200, 700, 5, 2000][seed]
    rec = synth.adversarial_contig(L, n, 1000 + seed, max_len=min(300, max(2, L)), deep=(seed in (3, 7)),
                                   overhang=(seed in (1, 4, 6)))
    ref = synth.make_reference(L, 50 + seed, lowercase=(seed % 2 == 0))
    compare([("chrA", seed % 3, L, ref, rec)], ADV_OPTS[seed], tmp_path, f"adv{seed}")


def test_multi_contig_bed_with_duplicate_lines(tmp_path):
    contigs = []
    for t in range(5):
        L = [3000, 1, 2500, 4097, 100][t]
        rec = synth.adversarial_contig(L, [300, 0, 10, 800, 40][t], 200 + t, max_len=min(200, max(2, L)))
        ref = synth.make_reference(L, 70 + t)
        if t == 2:
            ref = ref[: L - 100]                 # FASTA shorter than the header length: tail reads 'N'
        contigs.append((["chr1", "chr2", "chr3", "chrX", "chrM"][t], t, L, ref, rec))
    compare(contigs, dict(min_depth=2, min_depth_for_low_mapq=3), tmp_path, "multi")
    # -L style subset: only tids 1 and 3
    compare([contigs[1], contigs[3]], dict(), tmp_path, "subset")


def test_empty_inputs(tmp_path):
    compare([("e1", 0, 0, None, ContigRecords.empty()), ("e2", 1, 5000, None, ContigRecords.empty()),
             ("e3", 2, 4096, synth.make_reference(4096, 3), ContigRecords.empty())], dict(), tmp_path, "empty")


def test_short_reads_2mb_30x(tmp_path):
    L = 2_000_000
    rec = synth.short_read_contig(L, 30, synth.seed_for(2, 20))
    ref = synth.make_reference(L, synth.seed_for(2, 20))
    compare([("chr21", 20, L, ref, rec)], dict(), tmp_path, "short2mb")


def test_long_reads_indel_rich(tmp_path):
    L = 300_000
    rec = synth.long_read_contig(L, 50, synth.seed_for(3, 23))
    ref = synth.make_reference(L, synth.seed_for(3, 23))
    compare([("chrY", 23, L, ref, rec)], dict(), tmp_path, "long")


def test_deep_pileup_uses_32bit_counters(tmp_path):
    # 3000 reads stacked over a 400 bp region: columns far above 255 and above --max-depth
    L = 6000
    rng = np.random.default_rng(4)
    reads = [(int(p), "120M", int(rng.choice([0, 60, 60, 60])), int(rng.choice([10, 30, 40])), 0, f"d{i}")
             for i, p in enumerate(np.sort(rng.integers(2000, 2400, size=3000)))]
    rec = ContigRecords.from_reads(reads)
    ref = synth.make_reference(L, 8)
    o, g = compare([("amp", 0, L, ref, rec)], dict(max_depth=100000, min_depth_for_low_mapq=10), tmp_path, "deep")
    assert o["amp"]["dumps"][0].max() > 600
    # with the CLI's cap (500) the admission rule thins the pile
    o2, _ = compare([("amp", 0, L, ref, rec)], dict(), tmp_path, "deepcap")
    assert o2["amp"]["state_counts"][4] >= 0


def test_window_with_more_than_65535_reads_uses_the_32bit_variant(tmp_path):
    # 70 000 reads on 1 500 start positions: the 16-bit qc counters cannot hold a window like this,
    # k_window_bounds flags it and the engine re-runs the contig with 32-bit counters
    L = 5000
    rng = np.random.default_rng(11)
    n = 70_000
    pos = np.sort(rng.integers(1000, 2500, size=n)).astype(np.int32)
    rl = 40
    rec = ContigRecords(
        pos=pos, flag=np.zeros(n, np.uint16), mapq=rng.choice([0, 20, 60, 60], size=n).astype(np.uint8),
        cigar_off=np.arange(n + 1, dtype=np.uint32), cigar=np.full(n, (rl << 4) | 0, np.uint32),
        qual_off=(np.arange(n + 1, dtype=np.uint64) * np.uint64(rl)),
        qual=rng.choice([5, 25, 40], size=n * rl).astype(np.uint8),
        qname_off=(np.arange(n + 1, dtype=np.uint32) * np.uint32(10)), qname=synth._names_fixed(np.arange(n)).reshape(-1))
    ref = synth.make_reference(L, 12)
    o, g = compare([("deep", 0, L, ref, rec.validate())], dict(max_depth=1_000_000, min_depth_for_low_mapq=10),
                   tmp_path, "deep65k")
    assert o["deep"]["dumps"][0].max() > 1500


def test_resident_rerun_is_idempotent_and_split_api_agrees(tmp_path):
    L = 300_000
    rec = synth.short_read_contig(L, 30, 77)
    ref = synth.make_reference(L, 77)
    opt = CallableOptions()
    acc, _ = admit_reads(opt, 0, L, rec)
    idx = np.flatnonzero(acc)
    with Engine(opt, 0) as eng:
        eng.contig_begin(0, L, ref)
        # push in three tiles with tile-relative offsets
        cuts = [0, len(idx) // 3, 2 * len(idx) // 3, len(idx)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            sel = idx[a:b]
            lens_c = (rec.cigar_off[sel + 1] - rec.cigar_off[sel]).astype(np.int64)
            lens_q = (rec.qual_off[sel + 1] - rec.qual_off[sel]).astype(np.int64)
            cig = np.concatenate([rec.cigar[rec.cigar_off[i]:rec.cigar_off[i + 1]] for i in sel])
            q = np.concatenate([rec.qual[rec.qual_off[i]:rec.qual_off[i + 1]] for i in sel])
            coff = (np.concatenate([[0], np.cumsum(lens_c)]) + 7).astype(np.uint32)   # non-zero base
            cig = np.concatenate([np.zeros(7, np.uint32), cig])
            qoff = (np.concatenate([[0], np.cumsum(lens_q)]) + 5).astype(np.uint64)
            q = np.concatenate([np.zeros(5, np.uint8), q])
            eng.push_reads(rec.pos[sel], rec.mapq[sel], coff, cig, qoff, q)
        eng.contig_upload()
        eng.contig_run()
        r1 = eng.contig_collect()
        eng.contig_run(); eng.contig_run()
        r2 = eng.contig_collect()
    assert r1.as_dict() == r2.as_dict()
    assert np.array_equal(r1.intervals, r2.intervals)
    # same answer as the one-call path + oracle
    o_res, o_bed = oracle_run([("c", 0, L, ref, rec)], make_options({}), str(tmp_path / "o.bed"))
    assert r1.state_counts == o_res["c"]["state_counts"]
    iv = r1.intervals
    assert iv[0, 0] == 0 and iv[-1, 1] == L and np.array_equal(iv[1:, 0], iv[:-1, 1])
    assert np.all(iv[1:, 2] != iv[:-1, 2])
    names = np.array(oracle.STATE_NAMES)
    bed = "".join(f"c\t{s}\t{e}\t{names[k]}\n" for s, e, k in iv.tolist())
    assert bed == o_bed


def test_engine_errors_are_reported():
    from decodingustools_amd import EngineError
    opt = CallableOptions()
    with Engine(opt, 0) as eng:
        eng.contig_begin(0, 100, None)
        with pytest.raises(EngineError):                     # unsorted
            eng.push_reads([10, 5], [60, 60], [0, 1, 2], [5 << 4, 5 << 4], [0, 5, 10], np.full(10, 30, np.uint8))
        eng.contig_begin(0, 100, None)
        with pytest.raises(EngineError):                     # position outside the contig
            eng.push_reads([100], [60], [0, 1], [5 << 4], [0, 5], np.full(5, 30, np.uint8))
        eng.contig_begin(0, 100, None)
        eng.push_reads([10], [60], [0, 1], [(5 << 4) | 2], [0, 0], np.zeros(0, np.uint8))   # a lone "5D"
        with pytest.raises(EngineError):
            eng.contig_finish()
        with pytest.raises(EngineError):                     # call out of sequence
            Engine(opt, 0).contig_run()


def _ops(*pairs):
    code = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6, "=": 7, "X": 8}
    return [(l << 4) | code[o] for o, l in pairs]


@pytest.mark.parametrize("shape", ["short", "long"])
def test_cigar_errors_found_by_the_upload_walk(shape):
    """The read ends and the malformed-CIGAR flags of reads with more than 64 operations (and of every read of a
    long-read shaped contig) come from the host's walk at cl_push_reads: a zero-length reference-consuming operation
    deep inside such a read, and an end beyond the 32-bit coordinate range, are reported like the device-found ones."""
    from decodingustools_amd import EngineError
    opt = CallableOptions()
    # a 101-operation read: 50 x (20M 1I) + 20M; the variant with a 0D at operation 70
    good = _ops(*([("M", 20), ("I", 1)] * 50 + [("M", 20)]))
    bad = list(good); bad[70] = (0 << 4) | 2
    qn = 50 * 21 + 20
    filler = ([5] * 40, [60] * 40, list(range(41)), _ops(*[("M", 10)] * 40), [10 * i for i in range(41)], np.full(400, 30, np.uint8))

    def run(cig):
        with Engine(opt, 0) as eng:
            eng.contig_begin(0, 5000, None)
            if shape == "short":                              # short-read shape: < 8 operations per read on average
                eng.push_reads(*filler)
            eng.push_reads([10], [60], [0, len(cig)], cig, [0, qn], np.full(qn, 30, np.uint8))
            return eng.contig_finish()
    ok = run(good)
    assert ok.summary.summed_coverage == 1020 + (400 if shape == "short" else 0)
    with pytest.raises(EngineError, match="CIGAR"):
        run(bad)
    far = _ops(*([("M", 20), ("I", 1)] * 40 + [("N", (1 << 28) - 1)] * 17 + [("M", 20)]))     # 17 x 2^28 > 2^32
    with pytest.raises(EngineError, match="32-bit"):
        run(far)


def test_long_reads_in_tiles_with_a_refused_tile_between(tmp_path):
    """The CIGAR checkpoints are indexed by the operation's number in the contig's CIGAR array: tiles that cut that
    array at arbitrary places, and a refused tile between two good ones, give the single-push result."""
    from decodingustools_amd import EngineError
    opt = CallableOptions()
    L = 200_000
    rec = synth.long_read_contig(L, 40, 4242)
    ref = synth.make_reference(L, 4242)
    cuts = [0, rec.n // 5, rec.n // 5 + 1, rec.n // 2 + 3, rec.n]

    def push(eng, r):
        eng.push_reads(r.pos, r.mapq, r.cigar_off, r.cigar, r.qual_off, r.qual)
    with Engine(opt, 0) as eng:
        eng.contig_begin(0, L, ref); push(eng, rec)
        want = eng.contig_finish()
        eng.contig_begin(0, L, ref)
        for k in range(len(cuts) - 1):
            t = rec.slice(cuts[k], cuts[k + 1])
            if k == 2:
                bad = rec.slice(cuts[k], cuts[k + 1])
                bad.pos = bad.pos.copy(); bad.pos[-1] = 0                     # unsorted: refused after its walk
                with pytest.raises(EngineError):
                    push(eng, bad)
            push(eng, t)
        got = eng.contig_finish()
    assert got.as_dict() == want.as_dict() and np.array_equal(got.intervals, want.intervals)


def test_quality_prefetch_is_claimed_by_the_matching_tile_and_harmless_otherwise():
    """cl_contig_prefetch_qual: the bytes travel before their tile is pushed; the push that presents exactly them does
    not send them again, any other sequence of calls (another tile first, a reserve in between, no push at all, a
    second prefetch) just drops the prefetch -- the result is the same every time."""
    import ctypes as C
    from decodingustools_amd import _lib
    lib = _lib.load()
    opt = CallableOptions()
    L = 300_000
    rec = synth.short_read_contig(L, 30, 123)
    ref = synth.make_reference(L, 123)
    h = rec.n // 2
    a, b = rec.slice(0, h), rec.slice(h, rec.n)
    assert a.qual.shape[0] > (4 << 20) and b.qual.shape[0] > (4 << 20)

    def push(eng, r):
        eng.push_reads(r.pos, r.mapq, r.cigar_off, r.cigar, r.qual_off, r.qual)

    def prefetch(eng, r):
        q = np.ascontiguousarray(r.qual, np.uint8)
        assert lib.cl_contig_prefetch_qual(eng._h, q.ctypes.data_as(C.c_void_p), q.shape[0]) == 0
        return q                                      # must stay alive until the push that claims it

    with Engine(opt, 0) as eng:
        eng.contig_begin(0, L, ref); push(eng, a); push(eng, b)
        want = eng.contig_finish()
        results = []
        # claimed: prefetch a, push a (same buffer), prefetch b, push b
        eng.contig_begin(0, L, ref)
        qa = prefetch(eng, a); eng.push_reads(a.pos, a.mapq, a.cigar_off, a.cigar, a.qual_off, qa)
        qb = prefetch(eng, b); eng.push_reads(b.pos, b.mapq, b.cigar_off, b.cigar, b.qual_off, qb)
        results.append(eng.contig_finish())
        # not claimed: the prefetch names b's bytes, a is pushed first; then a reserve; then a prefetch nobody claims
        eng.contig_begin(0, L, ref)
        qb = prefetch(eng, b); push(eng, a)
        qb2 = prefetch(eng, b); eng.contig_reserve(rec.n, rec.cigar.shape[0], rec.qual.shape[0]); push(eng, b)
        results.append(eng.contig_finish())
        eng.contig_begin(0, L, ref)
        push(eng, a); push(eng, b); qa2 = prefetch(eng, a)
        results.append(eng.contig_finish())
        # a prefetch, then the contig is abandoned for another one
        eng.contig_begin(0, L, ref); qa3 = prefetch(eng, a)
        eng.contig_begin(0, L, ref); push(eng, a); push(eng, b)
        results.append(eng.contig_finish())
    for r in results:
        assert r.as_dict() == want.as_dict() and np.array_equal(r.intervals, want.intervals)


def test_a_refused_tile_leaves_the_contig_as_it_was():
    """cl_push_reads either takes a tile whole or leaves the context untouched: a contig pushed as good tile,
    refused tile (unsorted / out of range / broken offsets, small and large), good tile gives what the two good
    tiles alone give."""
    from decodingustools_amd import EngineError
    opt = CallableOptions()
    L = 400_000
    rec = synth.short_read_contig(L, 30, 99)
    ref = synth.make_reference(L, 99)
    h = rec.n // 2                                   # (the engine is compared with itself: no admission needed)
    a, b = rec.slice(0, h), rec.slice(h, rec.n)

    def push(eng, r):
        eng.push_reads(r.pos, r.mapq, r.cigar_off, r.cigar, r.qual_off, r.qual)
    with Engine(opt, 0) as eng:
        eng.contig_begin(0, L, ref)
        push(eng, a); push(eng, b)
        want = eng.contig_finish()
        eng.contig_begin(0, L, ref)
        push(eng, a)
        bad = b.slice(0, b.n)
        bad.pos = bad.pos.copy(); bad.pos[bad.n // 2] = 0                      # unsorted in the middle of a large tile (> 4 MiB of qualities)
        assert bad.qual.shape[0] > (4 << 20)
        with pytest.raises(EngineError):
            push(eng, bad)
        with pytest.raises(EngineError):                                      # starts before the previous tile ends
            push(eng, a.slice(0, 10))
        with pytest.raises(EngineError):                                      # outside the contig
            eng.push_reads([L], [60], [0, 1], [5 << 4], [0, 5], np.full(5, 30, np.uint8))
        with pytest.raises(EngineError):                                      # decreasing offsets
            eng.push_reads([L - 10, L - 9], [60, 60], [0, 2, 1], [5 << 4, 5 << 4], [0, 5, 10], np.full(10, 30, np.uint8))
        push(eng, b)
        got = eng.contig_finish()
    assert got.as_dict() == want.as_dict() and np.array_equal(got.intervals, want.intervals)


def test_contexts_that_interleave_their_contigs_do_not_share_staging():
    """The host staging arrays are pooled across contexts (taken at cl_contig_begin, given back at upload): two
    engines whose begin / push / finish calls interleave, and one that abandons a contig half way, give what each
    contig gives alone."""
    opt = CallableOptions()
    L1, L2 = 350_000, 200_000
    r1, r2 = synth.short_read_contig(L1, 30, 11), synth.long_read_contig(L2, 30, 12)
    f1, f2 = synth.make_reference(L1, 11), synth.make_reference(L2, 12)

    def push(eng, r):
        eng.push_reads(r.pos, r.mapq, r.cigar_off, r.cigar, r.qual_off, r.qual)
    with Engine(opt, 0) as a:
        a.contig_begin(0, L1, f1); push(a, r1); want1 = a.contig_finish()
        a.contig_begin(1, L2, f2); push(a, r2); want2 = a.contig_finish()
    with Engine(opt, 0) as a, Engine(opt, 0) as b, Engine(opt, 0) as c:
        for _ in range(2):
            a.contig_begin(0, L1, f1)
            b.contig_begin(1, L2, f2)
            c.contig_begin(0, L1, f1); push(c, r1.slice(0, r1.n // 3))        # never finished
            push(a, r1.slice(0, r1.n // 2))
            push(b, r2)
            got2 = b.contig_finish()
            push(a, r1.slice(r1.n // 2, r1.n))
            got1 = a.contig_finish()
            assert got1.as_dict() == want1.as_dict() and np.array_equal(got1.intervals, want1.intervals)
            assert got2.as_dict() == want2.as_dict() and np.array_equal(got2.intervals, want2.intervals)
            a, b = b, a                                                       # the contigs swap engines


@pytest.mark.parametrize("case", KATS["site_cases"], ids=[c["name"] for c in KATS["site_cases"]])
def test_site_kats_on_gpu(case):
    rec = ContigRecords.from_reads([tuple(r) for r in case["reads"]])
    ref = np.frombuffer(case["ref"].encode(), dtype=np.uint8).copy()
    exp = oracle.site_pileup(case["min_depth"], case["min_quality"], case["contig_len"], ref, rec, case["sites"])
    with Engine(CallableOptions(), 0) as eng:
        hist = eng.site_pileup(case["min_quality"], case["contig_len"], ref.shape[0], rec, case["sites"])
    assert np.array_equal(hist, exp["hist"])


def test_site_pileup_random_vs_oracle():
    L = 400_000
    ref = synth.make_reference(L, 5)
    rec = synth.short_read_contig(L, 40, synth.seed_for(5, 23), with_seq=True, ref=ref)
    rng = np.random.default_rng(9)
    sites = rng.choice(np.arange(1, L + 50), size=5000, replace=False).astype(np.uint32)
    exp = oracle.site_pileup(10, 20, L, ref, rec, sites)
    with Engine(CallableOptions(), 0) as eng:
        hist = eng.site_pileup(20, L, ref.shape[0], rec, sites)
    assert np.array_equal(hist, exp["hist"])
    assert exp["called"].sum() > 1000


def test_coverage_analyzer_and_sharded_driver_on_gpu(tmp_path):
    from decodingustools_amd.coverage import (ContigInput, CoverageAnalyzer, CoverageInput, analyze_sharded,
                                              engine_process_contig)
    names = ["chr1", "chr2", "chr10", "chrX", "chrM"]
    lens = [60_000, 31_000, 42_000, 20_480, 7_000]
    contigs = []
    for t, (nm, L) in enumerate(zip(names, lens)):
        rec = synth.short_read_contig(L, 30, 4000 + t) if t != 4 else synth.adversarial_contig(L, 900, 4100, deep=True)
        contigs.append(ContigInput(nm, L, rec, synth.make_reference(L, 60 + t, lowercase=(t == 4))))
    opt = CallableOptions()
    events = []
    inp = CoverageInput(contigs=contigs, options=opt, selected=["chr1", "chr10", "chrM", "chrZZ"],
                        output_bed=str(tmp_path / "a.bed"))
    out = CoverageAnalyzer(0).with_progress(events.append).analyze(inp)
    assert [e["event"] for e in events] == ["Started", "Completed"]
    keep = [(c.name, t, c.length, c.ref, c.records) for t, c in enumerate(contigs) if c.name in inp.selected]
    o_res, o_bed = oracle_run(keep, make_options({}), str(tmp_path / "o.bed"))
    assert open(out.bed_file).read() == o_bed
    assert [c["name"] for c in out.export["contigs"]] == ["chr1", "chr10", "chrM"]
    for c in out.export["contigs"]:
        st = o_res[c["name"]]["stats"]
        assert c["unique_reads"] == st["n_reads"] and c["covered_bases"] == st["n_covered_bases"]
        assert c["average_depth"] == st["derived"]["average_depth"]
        assert c["quality_stats"]["average_mapq"] == st["derived"]["average_mapq"]
        assert c["state_distribution"]["callable"] == o_res[c["name"]]["state_counts"][1]
    # the sharded driver with one rank is the same computation
    inp2 = CoverageInput(contigs=contigs, options=opt, selected=inp.selected, output_bed=str(tmp_path / "b.bed"))
    with Engine(opt, 0) as eng:
        out2 = analyze_sharded(inp2, 0, 1, lambda tid, c: engine_process_contig(eng, opt, tid, c))
    assert open(out2.bed_file).read() == o_bed and out2.export == out.export


def test_full_size_chr21_properties_and_sampled_regions(tmp_path):
    """BASELINE.json configs[1] at full size (46.7 Mb, 30x): size-independent properties of the
    result, plus bit-exact comparison with the oracle on sampled 60 kb regions (the oracle is run on
    the reads overlapping the region; positions near the region ends are excluded because reads
    were cut there)."""
    L = 46_709_983
    seed = synth.seed_for(2, 20)
    rec = synth.short_read_contig(L, 30, seed)
    ref = synth.make_reference(L, seed)
    opt = CallableOptions()
    acc, n_names = admit_reads(opt, 20, L, rec)
    with Engine(opt, 0) as eng:
        counter = CallableProfiler(str(tmp_path / "g.bed"))
        st = ContigProfiler("chr21", L)
        process_single_contig(eng, counter, st, opt, 20, rec, ref)
        counts = counter.get_contig_counts("chr21")
        counter.close()
        r1 = eng.contig_collect()
        eng.contig_run()
        r2 = eng.contig_collect()
        raw, qc, low, state = eng.debug_depths(L)
    iv = r1.intervals
    # idempotence of the resident re-run
    assert r1.as_dict() == r2.as_dict() and np.array_equal(iv, r2.intervals)
    # runs tile [0, L) exactly, neighbours differ, states valid
    assert iv[0, 0] == 0 and iv[-1, 1] == L and np.array_equal(iv[1:, 0], iv[:-1, 1])
    assert np.all(iv[1:, 2] != iv[:-1, 2]) and iv[:, 2].max() <= 5
    # conservation: run lengths per state == state counts == histogram of the state bytes; sum == L
    lens = (iv[:, 1] - iv[:, 0]).astype(np.int64)
    per_state = [int(lens[iv[:, 2] == k].sum()) for k in range(6)]
    assert per_state == counts == [int((state == k).sum()) for k in range(6)] and sum(counts) == L
    # REF_N positions are exactly the N / n bases of the reference
    assert counts[0] == int(((ref | 0x20) == ord("n")).sum())
    # per-read separable sums against numpy on the accepted reads
    ops = rec.cigar & 15
    lens_c = (rec.cigar >> 4).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(np.where(np.isin(ops, [0, 2, 3, 7, 8]), lens_c, 0))])
    rl = cs[rec.cigar_off[1:].astype(np.int64)] - cs[rec.cigar_off[:-1].astype(np.int64)]
    assert st.summed_coverage == int(rl[acc].sum()) == int(raw.astype(np.int64).sum())
    sel = acc & (rec.mapq >= opt.min_mapping_quality)
    assert st.summed_mapq == int((rec.mapq[sel].astype(np.int64) * rl[sel]).sum())
    assert st.n_covered_bases == int((raw > 0).sum()) and st.quality_bases == int(qc.astype(np.int64).sum())
    assert st.n_reads == n_names and np.all(qc <= raw) and np.all(low <= raw)
    # sampled regions against the oracle
    rng = np.random.default_rng(5)
    for a in rng.integers(20_000, L - 100_000, size=6).tolist() + [0, L - 60_000]:
        b = a + 60_000
        i0 = int(np.searchsorted(rec.pos, a - 400)); i1 = int(np.searchsorted(rec.pos, b))
        sub = ContigRecords(pos=rec.pos[i0:i1], flag=rec.flag[i0:i1], mapq=rec.mapq[i0:i1],
                            cigar_off=(rec.cigar_off[i0:i1 + 1] - rec.cigar_off[i0]).astype(np.uint32),
                            cigar=rec.cigar[rec.cigar_off[i0]:rec.cigar_off[i1]],
                            qual_off=(rec.qual_off[i0:i1 + 1] - rec.qual_off[i0]).astype(np.uint64),
                            qual=rec.qual[int(rec.qual_off[i0]):int(rec.qual_off[i1])],
                            qname_off=(rec.qname_off[i0:i1 + 1] - rec.qname_off[i0]).astype(np.uint32),
                            qname=rec.qname[rec.qname_off[i0]:rec.qname_off[i1]])
        prof = oracle.Profiler(str(tmp_path / "s.bed"))
        _, d = oracle.process_single_contig(prof, make_options({}), "chr21", 20, min(b, L), ref[:min(b, L)], sub, dump=True)
        prof.close()
        lo_p, hi_p = a, min(b, L) - (0 if b >= L else 0)
        for name, arr_o, arr_g in (("raw", d[0], raw), ("qc", d[1], qc), ("low", d[2], low), ("state", d[3], state)):
            assert np.array_equal(arr_o[lo_p:hi_p], arr_g[lo_p:hi_p]), (name, a)


def test_coverage_on_bam_and_fasta_files_and_cli(tmp_path):
    import json as _json
    import subprocess
    from bamio import write_bam, write_fasta
    from decodingustools_amd import build as _b
    from decodingustools_amd.bam import coverage_files
    names = ["chr1", "chr2", "chrX", "chrM"]
    lens = [120_000, 40_000, 30_000, 16_569]
    recs = {0: synth.short_read_contig(lens[0], 30, 700), 1: synth.adversarial_contig(lens[1], 3000, 701, deep=True),
            3: synth.short_read_contig(lens[3], 20, 703)}
    refs = [synth.make_reference(l, 800 + i, lowercase=(i == 3)) for i, l in enumerate(lens)]
    bam = str(tmp_path / "t.bam"); fa = str(tmp_path / "t.fa")
    write_bam(bam, list(zip(names, lens)), recs, block_every=1000)
    write_fasta(fa, list(zip(names, refs)))
    contigs = [(n, t, lens[t], refs[t], recs.get(t, ContigRecords.empty())) for t, n in enumerate(names)]
    o_res, o_bed = oracle_run(contigs, make_options({}), str(tmp_path / "o.bed"))
    bed = str(tmp_path / "g.bed"); js = str(tmp_path / "summary.json")
    rep = str(tmp_path / "rep.html")
    coverage_files(bam, fa, bed, js, CallableOptions(), output_summary=rep)     # indexed, 4 contigs: contig i+1 is read ahead
    assert open(bed).read() == o_bed
    # the HTML report and the coverage figures beside the BED (chrX has no reads: one NO_COVERAGE/REF_N picture or none)
    html = open(rep, encoding="utf-8").read()
    assert html.count('class="tab-panel') == 4 and "<dt>Aligner</dt>" in html
    assert os.path.exists(tmp_path / "chr1_coverage.svg") and os.path.exists(tmp_path / "chrM_coverage.svg")
    # the same without the read-ahead thread, and without an index (one sequential pass over the file)
    import shutil
    os.environ["DUT_PIPELINE"] = "0"
    try:
        coverage_files(bam, fa, str(tmp_path / "g0.bed"), str(tmp_path / "s0.json"), CallableOptions())
    finally:
        del os.environ["DUT_PIPELINE"]
    assert open(tmp_path / "g0.bed").read() == o_bed
    bam_ni = str(tmp_path / "noindex.bam")
    shutil.copy(bam, bam_ni)
    coverage_files(bam_ni, fa, str(tmp_path / "g1.bed"), str(tmp_path / "s1.json"), CallableOptions())
    assert open(tmp_path / "g1.bed").read() == o_bed
    # a file cut off inside a later contig: the error comes back from the read-ahead thread, nothing hangs
    from decodingustools_amd import EngineError
    bam_cut = str(tmp_path / "cut.bam")
    data = open(bam, "rb").read()
    open(bam_cut, "wb").write(data[:int(len(data) * 0.8)])
    shutil.copy(bam + ".bai", bam_cut + ".bai")
    with pytest.raises(EngineError) as ei:
        coverage_files(bam_cut, fa, str(tmp_path / "g2.bed"), None, CallableOptions())
    assert "Error processing contig" in str(ei.value) or "BAM" in str(ei.value)
    # summary.json: the CoverageOutput text, byte for byte (oracle: report.rs:15-134 + serde_json pretty)
    from oracle import report_oracle as RO
    hdr = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in zip(names, lens))
    ob = RO.BamStats(10000); ob.set_header(hdr)
    stream = []
    for t in sorted(recs):
        r = recs[t]
        for i in range(r.n):
            stream.append((int(r.flag[i]), int(r.qual_off[i + 1] - r.qual_off[i]), bytes(r.qname[r.qname_off[i]:r.qname_off[i + 1]]), 0))
    ob.collect(stream)
    assert len(ob.infer_platform_candidates()) == 1
    want = RO.coverage_output_json([o_res[n]["stats"] for n in names], names, [o_res[n]["state_counts"] for n in names],
                                   ob.aligner, ob.reference_build, ob.infer_platform_candidates()[0], ob.average_read_length(),
                                   bed, rep)
    assert open(js).read() == want
    summ = _json.load(open(js))["export"]
    assert [c["name"] for c in summ["contigs"]] == ["chr1", "chr2", "chrX", "chrM"]
    for c in summ["contigs"]:
        st = o_res[c["name"]]["stats"]
        assert c["unique_reads"] == st["n_reads"] and c["average_depth"] == st["derived"]["average_depth"]
        assert c["quality_stats"]["average_baseq"] == st["derived"]["average_baseq"]
        assert c["state_distribution"]["callable"] == o_res[c["name"]]["state_counts"][1]
    # -L subset + option flags through the command line tool
    sub = [contigs[1], contigs[3]]
    oo = dict(min_depth=2, max_depth=50, min_base_quality=13, max_low_mapq_fraction=0.25)
    _, o_bed2 = oracle_run(sub, make_options(oo), str(tmp_path / "o2.bed"))
    out = str(tmp_path / "cli.bed")
    r = subprocess.run([_b.CLI, "coverage", bam, "-r", fa, "-o", out, "-L", "chr2", "-L", "chrM", "-L", "nope",
                        "--min-depth", "2", "--max-depth=50", "--min-base-quality", "13", "--max-low-mapq-fraction", "0.25"],
                       cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out).read() == o_bed2
    cli_js = _json.load(open(tmp_path / "summary.json"))
    plots = [f"{n}_coverage.svg" for n in ("chr2", "chrM") if os.path.exists(tmp_path / f"{n}_coverage.svg")]
    assert cli_js["files"] == {"bed_file": out, "summary_html": "summary.html", "coverage_plots": plots} and len(plots) == 2
    assert "Callable Percentage" in open(tmp_path / "summary.html", encoding="utf-8").read()
    assert [c["name"] for c in cli_js["export"]["contigs"]] == ["chr2", "chrM"] and cli_js["export"]["summary"]["contigs_analyzed"] == 2
    r = subprocess.run([_b.CLI, bam, "-r", fa, "-o", out, "-L", "nope"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 1 and "None of the specified contigs (nope) were found in the BAM file" in r.stderr


def test_hifi_like_long_match_runs_and_truncated_qualities(tmp_path):
    """Reads of 8-20 kb with few, kilobase-long M runs (segments that span whole windows and need
    several trips), plus reads whose quality array is shorter than the CIGAR's query length (the
    reference's `qual().get(qpos)` is None beyond it) and reads with exactly 64 / 65 CIGAR ops
    (either side of the checkpointing threshold).  The shape that used to select the operation-parallel form; every
    long-read shape gets the run-table form now."""
    L = 120_000
    rng = np.random.default_rng(77)
    reads = []
    for i, p in enumerate(np.sort(rng.integers(0, L - 21_000, size=260))):
        tl = int(rng.integers(8_000, 20_000))
        ops, left = [], tl
        if rng.random() < 0.3: ops.append(("S", int(rng.integers(10, 400))))
        while left > 0:
            m = int(min(left, rng.integers(300, 5_000)))
            ops.append((str(rng.choice(["M", "=", "M"])), m)); left -= m
            if left > 0:
                g = rng.random()
                if g < 0.4: ops.append(("I", int(rng.integers(1, 4))))
                elif g < 0.8: ops.append(("D", int(rng.integers(1, 30))))
                else: ops.append(("X", 1))
        qlen = sum(l for o, l in ops if o in "MIS=X")
        q = rng.choice([7, 19, 20, 35, 50], size=qlen).tolist()
        if i % 9 == 0: q = q[: qlen // 2]                      # truncated quality array
        reads.append((int(p), "".join(f"{l}{o}" for o, l in ops), int(rng.choice([0, 5, 40, 60])), q, 0, f"h{i}"))
    for n_ops, p in ((64, 500), (65, 700), (63, 900), (129, 1100)):
        cig = "".join("20M1I" if k % 2 == 0 else "20M2D" for k in range((n_ops - 1) // 2)) + ("30M" if n_ops % 2 else "10M5S")
        from decodingustools_amd.records import cigar_from_string, cigar_query_length
        assert len(cigar_from_string(cig)) == n_ops
        reads.append((p, cig, 60, 30, 0, f"c{n_ops}"))
    reads.sort(key=lambda r: r[0])
    rec = ContigRecords.from_reads(reads)
    ref = synth.make_reference(L, 78)
    compare([("hifi", 3, L, ref, rec)], dict(min_depth=3, min_depth_for_low_mapq=4), tmp_path, "hifi")


def test_find_y_branch_on_files_and_cli(tmp_path):
    """find-y-branch end to end (config 5 plumbing): BAM + .bai, FASTA, a tree JSON of the FTDNA shape ->
    TSV, against oracle site pileup -> call -> score -> report (caller.rs:62-152, scoring.rs, mod.rs:92-258)."""
    import json as _json
    import random
    import subprocess
    from bamio import write_bam, write_fasta
    from decodingustools_amd import build as _b
    from decodingustools_amd import haplogroup as H
    from oracle import haplogroup_oracle as HO
    import test_haplogroup as TH
    L = 300_000
    ref = synth.make_reference(L, 31)
    rng = random.Random(21)
    ok_pos = [p for p in rng.sample(range(20_000, L - 20_000), 900) if chr(ref[p - 1]).upper() in "ACGT"]
    # tree alleles: ancestral = the reference base, derived = another base (1-based positions)
    def fix(nodes):
        for n in nodes.values():
            for v in n["variants"]:
                if v.get("position"):
                    p = abs(v["position"])
                    anc = chr(ref[p - 1]).upper()
                    v["ancestral"] = anc; v["derived"] = rng.choice([b for b in "ACGT" if b != anc])
    text = TH.ftdna_tree(rng, 250, ok_pos, extra=fix)
    tree_path = str(tmp_path / "ytree.json"); open(tree_path, "w").write(text)
    _, ot = HO.load_tree(text, "ftdna")
    # the sample: derived alleles along one root-to-leaf path, written into the reference the reads follow
    nodes = []
    def walk(h, path):
        nodes.append((h, path + [h["name"]]))
        for c in h["children"]: walk(c, path + [h["name"]])
    walk(ot, [])
    deepest = max(nodes, key=lambda x: len(x[1]))[1]
    sample = ref.copy()
    for h, _ in nodes:
        if h["name"] in deepest:
            for l in h["loci"]:
                c = l["coordinates"].get("GRCh38")
                if c: sample[c["position"] - 1] = ord(c["derived"][0])
    rec = synth.short_read_contig(L, 30, 77, with_seq=True, ref=sample)
    names = ["chr1", "chrY", "chrM"]; lens = [248956422, L, 16569]        # the chr1 length marks the header as GRCh38 (types.rs:140-142)
    bam = str(tmp_path / "y.bam"); fa = str(tmp_path / "y.fa")
    write_bam(bam, list(zip(names, lens)), {1: rec}, block_every=5000)
    write_fasta(fa, [("chrY", ref), ("chrM", synth.make_reference(16569, 32))])
    # oracle
    sites, rel = HO.sites_and_relevance(ot, "GRCh38", "chrY")
    exp = oracle.site_pileup(10, 20, L, ref, rec, np.asarray(sites, np.uint32))
    calls = HO.call_sites(sites, rel, exp["hist"], 10)
    assert len(calls) > 300
    for show in (False, True):
        want, rows = HO.report_text(ot, calls, "GRCh38", show)
        assert len(rows) >= len(deepest) - 3 and rows[0]["name"] == deepest[-1]
        out = str(tmp_path / f"hap{int(show)}.tsv")
        H.analyze_haplogroup(bam, fa, tree_path, out, show_snps=show)
        assert open(out).read() == want
    out = str(tmp_path / "cli.tsv")
    r = subprocess.run([_b.CLI, "find-y-branch", bam, "-r", fa, out, "--tree", tree_path, "--min-depth", "12", "--min-quality=30", "--show-snps"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exp2 = oracle.site_pileup(12, 30, L, ref, rec, np.asarray(sites, np.uint32))
    want2, _ = HO.report_text(ot, HO.call_sites(sites, rel, exp2["hist"], 12), "GRCh38", True)
    assert open(out).read() == want2
    # mt tree from FTDNA has no rCRS coordinates: header only (ftdna.rs:30-38, mod.rs:51-54)
    out = str(tmp_path / "mt.tsv")
    H.analyze_haplogroup(bam, fa, tree_path, out, tree_type=H.MTDNA)
    assert open(out).read().count("\n") == 1
    with pytest.raises(Exception, match="Could not determine reference genome"):
        bam2 = str(tmp_path / "n.bam")
        write_bam(bam2, [("chrY", L)], {0: rec})
        H.analyze_haplogroup(bam2, fa, tree_path, out)


def test_outlier_spans_do_not_widen_every_window(tmp_path):
    """A few reads with very long reference spans (spliced 'N' gaps, a megabase deletion) among ordinary
    short reads: they are kept in the engine's wide-read list and looked up per window; results must not
    change, including raw_depth over the gaps (mod.rs:22-23 counts D/N columns) and the windows far
    inside a gap that only the wide reads touch."""
    L = 1_500_000
    base = synth.short_read_contig(L, 12, 555)
    ref = synth.make_reference(L, 556)
    rng = np.random.default_rng(557)
    extra = []
    for i in range(40):
        p = int(rng.integers(0, L - 1_100_000))
        gap = int(rng.choice([20_000, 70_000, 400_000, 1_000_000]))
        op = "N" if i % 2 else "D"
        extra.append((p, f"60M{gap}{op}40M5S", int(rng.choice([0, 30, 60])), 35, 0, f"w{i}"))
    extra.append((10, f"10M{L - 100}N10M", 60, 30, 0, "span_all"))            # touches every window
    extra.append((L - 30_000, "50M20000D50M", 60, 30, 0, "tail"))
    # merge in coordinate order
    wide = ContigRecords.from_reads(sorted(extra, key=lambda r: r[0]))
    reads = []
    def rows(rec):
        for i in range(rec.n):
            cig = "".join(f"{int(c) >> 4}{'MIDNSHP=XB'[int(c) & 15]}" for c in rec.cigar[rec.cigar_off[i]:rec.cigar_off[i + 1]])
            reads.append((int(rec.pos[i]), cig, int(rec.mapq[i]), rec.qual[int(rec.qual_off[i]):int(rec.qual_off[i + 1])].tolist(), int(rec.flag[i]),
                          bytes(rec.qname[rec.qname_off[i]:rec.qname_off[i + 1]]).decode()))
    sub = base.slice(0, min(base.n, 60_000))
    rows(sub); rows(wide)
    reads.sort(key=lambda r: r[0])
    rec = ContigRecords.from_reads(reads)
    compare([("chrW", 5, L, ref, rec)], dict(min_depth=2, min_depth_for_low_mapq=3), tmp_path, "wide")


@pytest.mark.parametrize("reserve", [False, True])
def test_tiles_of_mixed_sizes_small_staged_large_direct(reserve):
    """cl_push_reads copies small tiles to host staging and sends the quality bytes of tiles of >= 4 MiB
    straight to the device; any mix of the two, in any order, must give the one-tile result."""
    L = 900_000
    rec = synth.short_read_contig(L, 30, 91)
    ref = synth.make_reference(L, 92)
    opt = CallableOptions()
    acc, _ = admit_reads(opt, 0, L, rec)
    idx = np.flatnonzero(acc)
    keep = rec.slice(0, rec.n)
    # one tile of everything accepted = the reference result
    def tile(sel):
        lens_c = (rec.cigar_off[sel + 1] - rec.cigar_off[sel]).astype(np.int64)
        lens_q = (rec.qual_off[sel + 1] - rec.qual_off[sel]).astype(np.int64)
        take_c = np.repeat(rec.cigar_off[sel].astype(np.int64), lens_c) + (np.arange(lens_c.sum()) - np.repeat(np.cumsum(lens_c) - lens_c, lens_c))
        take_q = np.repeat(rec.qual_off[sel].astype(np.int64), lens_q) + (np.arange(lens_q.sum()) - np.repeat(np.cumsum(lens_q) - lens_q, lens_q))
        return (rec.pos[sel], rec.mapq[sel], np.concatenate([[0], np.cumsum(lens_c)]).astype(np.uint32), rec.cigar[take_c],
                np.concatenate([[0], np.cumsum(lens_q)]).astype(np.uint64), rec.qual[take_q])
    with Engine(opt, 0) as eng:
        eng.contig_begin(0, L, ref)
        eng.push_reads(*tile(idx))
        want = eng.contig_finish()
    sizes = [100, 40_000, 50, 30_000, 10_000, 1, 45_000]          # reads per tile: 6 MB and 4.5 MB tiles go direct
    with Engine(opt, 0) as eng:
        eng.contig_begin(0, L, ref)
        if reserve:
            eng.contig_reserve(len(idx), int((rec.cigar_off[idx + 1] - rec.cigar_off[idx]).sum()), int((rec.qual_off[idx + 1] - rec.qual_off[idx]).sum()))
        a = 0
        k = 0
        while a < len(idx):
            b = min(len(idx), a + sizes[k % len(sizes)]); k += 1
            eng.push_reads(*tile(idx[a:b]))
            a = b
        got = eng.contig_finish()
    assert k > 7 and got.as_dict() == want.as_dict() and np.array_equal(got.intervals, want.intervals)
    assert want.summary.n_intervals > 1000


def _stacked_multi_op_reads(n, start_lo, start_hi, seed, long_every=0):
    """n reads with >= 9 CIGAR operations each (the operation-parallel kernel variant is chosen at >= 8
    operations per read on average), stacked on a short stretch so that a window sees many of them."""
    rng = np.random.default_rng(seed)
    reads = []
    for i, p in enumerate(np.sort(rng.integers(start_lo, start_hi, size=n))):
        k = int(rng.integers(4, 7))
        ops = []
        for j in range(k):
            ops.append(f"{int(rng.integers(8, 40))}{rng.choice(['M', 'M', '=', 'X'])}")
            ops.append(f"{int(rng.integers(1, 4))}{rng.choice(['I', 'D', 'D', 'N'])}")
        ops.append(f"{int(rng.integers(10, 30))}M")
        if long_every and i % long_every == 0:
            ops.append("3D"); ops.append("150M")                       # a run longer than 64 bases: list + quad loop
        if i % 5 == 0:
            ops = [f"{int(rng.integers(1, 9))}S"] + ops
        reads.append((int(p), "".join(ops), int(rng.choice([0, 1, 20, 60, 60])), int(rng.choice([5, 20, 21, 38])), 0, f"s{i}"))
    return ContigRecords.from_reads(reads)


def test_operation_parallel_variant_with_16bit_and_32bit_counters(tmp_path):
    """A long-read shaped contig (the run-table form; until round 3 the operation-parallel one, hence the name) on
    windows deeper than 255 (16-bit counter fields) and with more than 32 767 candidates (32-bit re-run)."""
    L = 12_000
    ref = synth.make_reference(L, 41)
    rec = _stacked_multi_op_reads(2500, 3000, 3900, 42, long_every=7)     # deeper than 255: the 16-bit fields are needed
    assert rec.cigar.shape[0] >= 8 * rec.n
    o, _ = compare([("m16", 0, L, ref, rec)], dict(max_depth=100_000, min_depth_for_low_mapq=10), tmp_path, "long16")
    assert o["m16"]["dumps"][0].max() > 300
    rec8 = _stacked_multi_op_reads(2500, 3000, 9000, 44, long_every=7)    # > 510 candidates per window, depth < 255: 8-bit sets
    o, _ = compare([("m8w", 0, L, ref, rec8)], dict(max_depth=100_000), tmp_path, "long8w")
    assert 60 < o["m8w"]["dumps"][0].max() < 255
    # the AND form of the byte-parallel quality test (thresholds above 128) in this variant
    compare([("m16q", 0, L, ref, rec)], dict(max_depth=100_000, min_base_quality=200, min_depth=1), tmp_path, "long16q")
    compare([("m8", 0, L, ref, rec.slice(0, 300))], dict(max_depth=100_000, min_base_quality=21, min_depth=1), tmp_path, "long8")
    rec = _stacked_multi_op_reads(34_000, 3000, 4500, 43, long_every=11)
    assert rec.cigar.shape[0] >= 8 * rec.n
    o, _ = compare([("m32", 0, L, ref, rec)], dict(max_depth=1_000_000, min_depth_for_low_mapq=10), tmp_path, "long32")
    assert o["m32"]["dumps"][0].max() > 3_000


def _mgpu_worker(rank, world, port, args):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from decodingustools_amd import coverage_mgpu
    sys.exit(coverage_mgpu.main(args))


def test_coverage_on_files_over_two_processes(tmp_path):
    """The multi-GPU file driver with two ranks (gloo; both on the one GPU of the test box): BED and
    summary.json must equal the single-process `coverage` on the same files."""
    import socket
    import torch.multiprocessing as mp
    from bamio import write_bam, write_fasta
    from decodingustools_amd.bam import coverage_files
    names = ["chr1", "chr2", "chr3", "chrX", "chrM"]
    lens = [90_000, 60_000, 25_000, 30_000, 16_569]
    recs = {0: synth.short_read_contig(lens[0], 30, 710), 1: synth.short_read_contig(lens[1], 12, 711),
            3: synth.adversarial_contig(lens[3], 1500, 712), 4: synth.short_read_contig(lens[4], 20, 713)}
    refs = [synth.make_reference(l, 820 + i, lowercase=(i == 4)) for i, l in enumerate(lens)]
    bam = str(tmp_path / "t.bam"); fa = str(tmp_path / "t.fa")
    write_bam(bam, list(zip(names, lens)), recs, block_every=2000)
    write_fasta(fa, list(zip(names, refs)))
    one = tmp_path / "one"; two = tmp_path / "two"; one.mkdir(); two.mkdir()
    for sel, tag in ((None, "all"), (["chr2", "chrM", "chr3"], "sel")):
        bed1 = str(one / f"{tag}.bed"); js1 = str(one / f"{tag}.json")
        coverage_files(bam, fa, bed1, js1, CallableOptions(), contigs=sel)
        bed2 = str(two / f"{tag}.bed"); js2 = str(two / f"{tag}.json")
        html2 = str(two / f"{tag}.html")
        args = [bam, "-r", fa, "-o", bed2, "--summary-json", js2, "-s", html2, "--backend", "gloo"] + sum((["-L", c] for c in (sel or [])), [])
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=_mgpu_worker, args=(r, 2, port, args)) for r in range(2)]
        for p in procs: p.start()
        for p in procs: p.join(300)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert open(bed2).read() == open(bed1).read()
        j1 = open(js1).read().replace(bed1, "BED"); j2 = open(js2).read().replace(bed2, "BED").replace(html2, "summary.html")
        assert j1 == j2
        # the coverage figures beside the two BED files are the same drawings, and rank 0 wrote the report
        svgs = sorted(f for f in os.listdir(one) if f.endswith("_coverage.svg"))
        assert svgs and svgs == sorted(f for f in os.listdir(two) if f.endswith("_coverage.svg"))
        for f in svgs:
            assert open(one / f).read() == open(two / f).read(), f
        assert "Callable Percentage" in open(html2, encoding="utf-8").read()


def test_many_small_contigs_with_random_options(tmp_path):
    """Differential sweep: small adversarial contigs, several per BED (the duplicated last line between
    them), each round with its own random option set -- depth caps low enough to bite, thresholds on
    both sides of 128, fractions incl. 0 and ~1, windows lengths around multiples of 2048."""
    rng = np.random.default_rng(2024)
    lens_pool = [1, 2047, 2048, 2049, 4096, 5000, 6145, 9000]
    for rnd in range(30):
        opt = dict(min_depth=int(rng.integers(0, 12)), max_depth=int(rng.choice([0, 3, 20, 150, 500, 100000])),
                   min_mapping_quality=int(rng.choice([0, 1, 10, 30, 61])), min_base_quality=int(rng.choice([0, 1, 20, 40, 127, 128, 129, 200, 255])),
                   min_depth_for_low_mapq=int(rng.integers(0, 15)), max_low_mapq=int(rng.choice([0, 1, 5, 60])),
                   max_low_mapq_fraction=float(rng.choice([0.0, 0.05, 0.1, 0.5, 0.999])))
        contigs = []
        for t in range(3):
            L = int(rng.choice(lens_pool))
            n_reads = int(rng.integers(0, 700)) if L > 1 else int(rng.integers(0, 3))
            rec = synth.adversarial_contig(L, n_reads, 5000 + 10 * rnd + t, deep=bool(rng.integers(0, 2)), overhang=bool(rng.integers(0, 2))) if n_reads else ContigRecords.empty()
            contigs.append((f"c{rnd}_{t}", t, L, synth.make_reference(L, 7000 + 10 * rnd + t, lowercase=bool(t == 1)) if rng.random() < 0.9 else None, rec))
        compare(contigs, opt, tmp_path, f"sweep{rnd}")


def test_deeper_short_read_data_keeps_the_8bit_counters(tmp_path):
    """60x and 120x of 150-base reads: more than 510 candidates per window but no position deeper than 255,
    so the 8-bit counter sets stay in use (checked by depth, not by candidate count); a pile deeper than
    255 inside such a contig makes the engine redo it with 16-bit fields.  Results must not change."""
    L = 150_000
    ref = synth.make_reference(L, 61)
    for depth, seed in ((60, 62), (120, 63)):
        rec = synth.short_read_contig(L, depth, seed, max_live_assert=100_000)
        o, _ = compare([(f"d{depth}", 0, L, ref, rec)], dict(max_depth=100_000), tmp_path, f"depth{depth}")
        assert 40 < o[f"d{depth}"]["dumps"][0].max() <= 255 or depth == 120
    # 60x plus a 700-deep pile: the optimistic mode is refused for this contig
    base = synth.short_read_contig(L, 60, 64, max_live_assert=100_000)
    rng = np.random.default_rng(65)
    reads = [(int(p), "100M", 60, 30, 0, f"p{i}") for i, p in enumerate(np.sort(rng.integers(70_000, 70_050, size=700)))]
    def rows(rec):
        out = []
        for i in range(rec.n):
            cig = "".join(f"{int(c) >> 4}{'MIDNSHP=XB'[int(c) & 15]}" for c in rec.cigar[rec.cigar_off[i]:rec.cigar_off[i + 1]])
            out.append((int(rec.pos[i]), cig, int(rec.mapq[i]), rec.qual[int(rec.qual_off[i]):int(rec.qual_off[i + 1])].tolist(), int(rec.flag[i]),
                        bytes(rec.qname[rec.qname_off[i]:rec.qname_off[i + 1]]).decode()))
        return out
    sub = base.slice(int(np.searchsorted(base.pos, 60_000)), int(np.searchsorted(base.pos, 80_000)))
    merged = sorted(rows(sub) + reads, key=lambda r: r[0])
    o, _ = compare([("pile", 0, L, ref, ContigRecords.from_reads(merged))], dict(max_depth=100_000), tmp_path, "pile")
    assert o["pile"]["dumps"][0].max() > 600


def _eqx_split(rec, seed):
    """Every M run of >= 3 bases becomes '=' 'X'(1) '=' at a random cut (the --eqx style of CIGAR): the same
    alignment with three match-type operations in a row."""
    import dataclasses
    rng = np.random.default_rng(seed)
    cig = rec.cigar
    op, ln = cig & 15, cig >> 4
    split = (op == 0) & (ln >= 3)
    cnt = np.where(split, 3, 1)
    start = np.concatenate([[0], np.cumsum(cnt)])
    out = np.empty(int(start[-1]), dtype=np.uint32)
    keep = ~split
    out[start[:-1][keep]] = np.where(op[keep] == 0, (ln[keep] << 4) | 7, cig[keep])      # short M -> '='
    a = (rng.random(int(split.sum())) * (ln[split] - 2)).astype(np.uint32) + 1           # 1 .. len-2
    s0 = start[:-1][split]
    out[s0] = (a << 4) | 7
    out[s0 + 1] = (1 << 4) | 8
    out[s0 + 2] = ((ln[split] - a - 1) << 4) | 7
    off = start[rec.cigar_off.astype(np.int64)].astype(np.uint32)
    return dataclasses.replace(rec, cigar=out, cigar_off=off)


def test_long_reads_with_eqx_cigars(tmp_path):
    """Three or four match-type runs among a lane's four operations (the paths behind the first two runs
    of the four-operations-per-lane variant), '=' and 'X' operations throughout."""
    L = 200_000
    rec = _eqx_split(synth.long_read_contig(L, 40, synth.seed_for(3, 29)), 5)
    rec.validate()
    assert rec.cigar.shape[0] >= 8 * rec.n and rec.qual.shape[0] < 32 * rec.cigar.shape[0]
    assert ((rec.cigar & 15) == 8).sum() > 10_000
    ref = synth.make_reference(L, synth.seed_for(3, 29))
    compare([("chrE", 3, L, ref, rec)], dict(), tmp_path, "eqx")
    compare([("chrE", 3, L, ref, rec)], dict(min_base_quality=25, min_mapping_quality=30, max_depth=30), tmp_path, "eqx2")


# ---- the run-table form of k_pileup (contigs with short match runs): the host's walk at upload ----
_RUN_TABLE_CASE = r"""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import test_gpu_parity as T
from decodingustools_amd import synth
L = 150_000
rec = synth.long_read_contig(L, 50, synth.seed_for(3, 31))
ref = synth.make_reference(L, synth.seed_for(3, 31))
assert rec.cigar.shape[0] >= 8 * rec.n and rec.qual.shape[0] < 56 * rec.cigar.shape[0]
T.compare([("chrR", 5, L, ref, rec)], dict(), pathlib.Path(tempfile.mkdtemp()), "runtab_" + os.environ["DUT_RUN_CHUNK"])
print("RUN_TABLE_OK")
"""


@pytest.mark.parametrize("chunk", ["70", "3000", "40000"])
def test_run_table_through_small_pinned_buffers(chunk):
    """The pieces of a window go to HBM through pinned buffers of DUT_RUN_CHUNK pieces: 70 -- every window is larger than
    a buffer and travels as a block of its own; 3 000 -- a buffer holds less than one ordinary window; 40 000 -- a few
    windows per buffer, the buffer-full path with the window started over.  (The knob is read once per process.)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DUT_RUN_CHUNK=chunk)
    r = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}\n" + _RUN_TABLE_CASE], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RUN_TABLE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_run_table_with_one_base_runs_needs_the_second_sizing_pass(tmp_path):
    """Match runs of one base (a piece per base): more pieces than the first estimate of the device array holds, so the
    walk runs twice; '=' / 'X' runs, reads that start in the middle of a unit, a deletion right at a window seam."""
    L = 9_000
    rng = np.random.default_rng(77)
    reads = []
    for i, p in enumerate(np.sort(rng.integers(0, L - 700, 2600))):
        unit = ["1M1I1M1D", "1=1I1X1D", "2M1D1M1I"][i % 3]
        cig = unit * 60
        if i % 4 == 0:
            cig = f"{int(rng.integers(1, 6))}S" + cig
        reads.append((int(p), cig, int(rng.choice([0, 9, 10, 60, 60])), int(rng.choice([5, 19, 20, 40])), 0, f"o{i}"))
    # a read whose deletion spans the seam between the windows at 2048 and another that ends exactly on it
    reads.append((2040, "8M3D9M", 60, 33, 0, "seam1")); reads.append((2032, "16M", 60, 33, 0, "seam2"))
    reads.sort(key=lambda r: r[0])
    rec = ContigRecords.from_reads(reads)
    assert rec.cigar.shape[0] >= 8 * rec.n and rec.qual.shape[0] < 56 * rec.cigar.shape[0]
    ref = synth.make_reference(L, 78)
    compare([("chr1b", 2, L, ref, rec)], dict(max_depth=100_000, min_depth=2), tmp_path, "runtab1b")
    compare([("chr1b", 2, L, ref, rec)], dict(max_depth=100_000, min_base_quality=0, min_mapping_quality=0), tmp_path, "runtab1b0")


def test_run_table_with_truncated_qualities_and_long_runs(tmp_path):
    """Reads whose quality string is shorter than their CIGAR says (pieces stop where the bytes do), runs longer than two
    units (several pieces per run), long N gaps and reads that span many windows -- in a contig the run-table form takes."""
    L = 60_000
    rng = np.random.default_rng(91)
    reads = []
    for i, p in enumerate(np.sort(rng.integers(0, L - 9000, 500))):
        ops = []
        for j in range(int(rng.integers(20, 60))):
            ops.append(f"{int(rng.integers(1, 120))}M")
            ops.append(rng.choice(["1I", "2D", "1D", "3I", "300N" if j % 17 == 0 else "1D"]))
        cig = "".join(ops)
        reads.append([int(p), cig, int(rng.choice([0, 10, 60])), int(rng.choice([10, 20, 30])), 0, f"t{i}"])
    rec = ContigRecords.from_reads(reads)
    # cut the quality strings of every third read short (records.validate() only checks the totals)
    keep = np.ones(rec.qual.shape[0], bool)
    qoff = rec.qual_off.astype(np.int64)
    newoff = [0]
    for i in range(rec.n):
        a, b = int(qoff[i]), int(qoff[i + 1])
        cut = (b - a) // 3 if i % 3 == 0 else 0
        keep[b - cut:b] = False
        newoff.append(newoff[-1] + (b - a - cut))
    rec.qual = np.ascontiguousarray(rec.qual[keep]); rec.qual_off = np.asarray(newoff, np.uint64)
    rec.validate()
    if not (rec.cigar.shape[0] >= 8 * rec.n and rec.qual.shape[0] < 56 * rec.cigar.shape[0]):
        pytest.skip("shape does not select the run-table form")
    ref = synth.make_reference(L, 92)
    compare([("chrT", 1, L, ref, rec)], dict(min_depth=1), tmp_path, "runtabT")


def test_site_tile_stays_resident_for_several_site_lists():
    """cl_site_upload once, cl_site_run with different lists, qualities and an empty list; a second upload replaces the
    tile; cl_site_run before any upload is an error."""
    from decodingustools_amd import EngineError
    L = 300_000
    ref = synth.make_reference(L, 15)
    rec = synth.short_read_contig(L, 35, synth.seed_for(5, 7), with_seq=True, ref=ref)
    rng = np.random.default_rng(19)
    lists = [rng.choice(np.arange(1, L + 20), size=n, replace=False).astype(np.uint32) for n in (4000, 1, 900)]
    lists.append(np.array([0, 7, 7, L, L + 5, 2**31], np.uint32))            # vcf_pos 0, a duplicate, the end, beyond it
    with Engine(CallableOptions(), 0) as eng:
        with pytest.raises(EngineError):
            eng.site_run(20, lists[0])
        eng.site_upload(L, ref.shape[0], rec)
        for mq, sites in ((20, lists[0]), (0, lists[1]), (61, lists[2]), (10, lists[3]), (20, lists[0])):
            exp = oracle.site_pileup(10, mq, L, ref, rec, sites)
            assert np.array_equal(eng.site_run(mq, sites), exp["hist"]), (mq, sites.shape)
        assert eng.site_run(20, np.zeros(0, np.uint32)).shape == (0, 16)
        sub = rec.slice(0, rec.n // 3)
        eng.site_upload(L, ref.shape[0], sub)
        exp = oracle.site_pileup(10, 20, L, ref, sub, lists[0])
        assert np.array_equal(eng.site_run(20, lists[0]), exp["hist"])
        # the one-call form sends only the reads that overlap a site of ITS list: its tile serves that call alone, a run
        # with another list over it is refused; with DUT_SITE_FILTER=0 the whole tile travels and stays (checked below)
        assert np.array_equal(eng.site_pileup(20, L, ref.shape[0], rec, lists[2]), oracle.site_pileup(10, 20, L, ref, rec, lists[2])["hist"])
        with pytest.raises(EngineError):
            eng.site_run(20, lists[0])
        eng.site_upload(L, ref.shape[0], rec)
        assert np.array_equal(eng.site_run(20, lists[0]), oracle.site_pileup(10, 20, L, ref, rec, lists[0])["hist"])


_SITE_FILTER_CASE = r"""
import os, sys
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import oracle
from decodingustools_amd import CallableOptions, Engine, synth
L = 300_000
ref = synth.make_reference(L, 15)
rec = synth.short_read_contig(L, 35, synth.seed_for(5, 7), with_seq=True, ref=ref)
sites = np.random.default_rng(3).choice(np.arange(1, L + 20), size=700, replace=False).astype(np.uint32)
with Engine(CallableOptions(), 0) as eng:
    got = eng.site_pileup(20, L, ref.shape[0], rec, sites)
    assert np.array_equal(got, oracle.site_pileup(10, 20, L, ref, rec, sites)["hist"])
    other = np.arange(5, 5000, 7, dtype=np.uint32)
    assert np.array_equal(eng.site_run(20, other), oracle.site_pileup(10, 20, L, ref, rec, other)["hist"])   # the whole tile is there
print("SITE_FILTER_OFF_OK")
"""


def test_the_site_tile_filtered_for_its_list_and_the_whole_tile_agree(tmp_path):
    """cl_site_pileup sends only the reads that can add to the histogram (position inside the contig, mapq, a site inside
    the span).  Lists so sparse that most reads stay behind, so dense that all travel, a list beyond the reads, reads that
    start the kept set at odd nibbles, zero-length sequences; tiles with a read of 255 operations or 65 535 bases fall
    back to the whole tile (long reads); DUT_SITE_FILTER=0 (a child process: read once) sends the whole tile always."""
    import subprocess
    L = 200_000
    ref = synth.make_reference(L, 25)
    rec = synth.short_read_contig(L, 30, synth.seed_for(5, 9), with_seq=True, ref=ref)
    rng = np.random.default_rng(29)
    lists = [rng.choice(np.arange(1, L + 1), size=n, replace=False).astype(np.uint32) for n in (3, 40, 600, 20_000)]
    lists.append(np.arange(1, 3000, dtype=np.uint32))                          # every position of a stretch
    lists.append(np.array([L + 500, 2**31], np.uint32))                        # no read at all
    with Engine(CallableOptions(), 0) as eng:
        for mq in (0, 20, 61):
            for sites in lists:
                exp = oracle.site_pileup(10, mq, L, ref, rec, sites)
                assert np.array_equal(eng.site_pileup(mq, L, ref.shape[0], rec, sites), exp["hist"]), (mq, sites.shape)
        # reads of 255 operations and more (the records' escape value): the whole tile travels, same histogram
        reads = []
        for i in range(300):
            many = i % 3 == 0
            cig = "1M1I" * 150 + "40M" if many else "%dM" % int(rng.integers(30, 200))
            ql = 340 if many else int(cig[:-1])
            seq = "".join(rng.choice(list("ACGTN"), size=ql))
            reads.append((int(i * 97 % 50_000), cig, int(rng.choice([0, 20, 60])), [30] * ql, 0, f"m{i}", seq))
        reads.sort(key=lambda r: r[0])
        mrec = ContigRecords.from_reads(reads)
        ref3 = synth.make_reference(60_000, 27)
        s3 = rng.choice(np.arange(1, 60_000), size=800, replace=False).astype(np.uint32)
        assert np.array_equal(eng.site_pileup(10, 60_000, ref3.shape[0], mrec, s3), oracle.site_pileup(10, 10, 60_000, ref3, mrec, s3)["hist"])
    env = dict(os.environ, DUT_SITE_FILTER="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % root + _SITE_FILTER_CASE], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SITE_FILTER_OFF_OK" in r.stdout, r.stderr[-2000:]


def test_an_error_behind_a_quality_prefetch_leaves_nothing_in_flight(tmp_path):
    """a tile with more than 4 MiB of qualities that turns out to be unsorted -- the host driver has
    started the quality prefetch by then.  The driver abandons the contig (cl_contig_abort: the copiers are joined, the
    staging ring is free), the context can be destroyed or reused at once, and another context of the device is not
    blocked by the abandoned transfer."""
    from decodingustools_amd import EngineError
    L = 400_000
    rec = synth.short_read_contig(L, 120, synth.seed_for(2, 5))
    assert rec.qual.shape[0] > (4 << 20)
    ref = synth.make_reference(L, 3)
    bad = ContigRecords(pos=rec.pos.copy(), flag=rec.flag, mapq=rec.mapq, cigar_off=rec.cigar_off, cigar=rec.cigar,
                        qual_off=rec.qual_off, qual=rec.qual, qname_off=rec.qname_off, qname=rec.qname)
    bad.pos[rec.n // 2] = bad.pos[rec.n // 2 - 1] - 50                 # one read out of order, far into the tile
    opt = CallableOptions()
    for reuse in (False, True):
        eng = Engine(opt, 0)
        other = Engine(opt, 0)
        with pytest.raises(EngineError):
            process_single_contig(eng, CallableProfiler(str(tmp_path / "x.bed")), ContigProfiler("c", L), opt, 0, bad, ref)
        # the device's ring is free: another context pushes a large tile at once
        counter = CallableProfiler(str(tmp_path / "o.bed"))
        st = ContigProfiler("c", L)
        process_single_contig(other, counter, st, opt, 0, rec, ref)
        counter.close()
        if reuse:                                                        # ... and the failed context works again
            counter2 = CallableProfiler(str(tmp_path / "r.bed"))
            st2 = ContigProfiler("c", L)
            process_single_contig(eng, counter2, st2, opt, 0, rec, ref)
            counter2.close()
            assert open(tmp_path / "r.bed").read() == open(tmp_path / "o.bed").read()
            assert st2.summed_coverage == st.summed_coverage
        eng.close()                                                      # straight after the error, or after the reuse
        other.close()
    # a prefetch that no tile ever claims, then destroy
    eng = Engine(opt, 0)
    eng.contig_begin(0, L, ref)
    eng._check(eng._lib.cl_contig_prefetch_qual(eng._h, rec.qual.ctypes.data, rec.qual.shape[0]))
    eng.close()
    eng = Engine(opt, 0)
    eng.contig_begin(0, L, ref)
    eng._check(eng._lib.cl_contig_prefetch_qual(eng._h, rec.qual.ctypes.data, rec.qual.shape[0]))
    eng.contig_abort()
    eng.close()


# ---- the record form of the short-read k_pileup: the host's walk at upload turns every read into a head record and
#      one piece record per further M/=/X run (callable_loci.hip: gen_read_recs) ----
def record_shapes_contig(L=200_000, seed=4242, n_plain=4000, short_form=True):
    """A contig the short-read form takes (fewer than 8 operations per read on average) that holds every shape the record
    builder distinguishes: plain reads; reads that start with S / I / H (the head carries the first run or not); runs
    behind deletions, insertions and N gaps (piece records); a read of more than 64 operations; match runs longer than
    65 535 bases (split into several pieces, and wide: their records go through the wide list); quality strings shorter
    than the CIGAR says and absent altogether; reads below every mapping-quality threshold."""
    rng = np.random.default_rng(seed)
    reads = []
    for i, p in enumerate(rng.integers(0, L - 400, n_plain)):
        reads.append([int(p), "150M", int(rng.choice([0, 1, 9, 10, 30, 60, 60, 60])), int(rng.choice([5, 19, 20, 35])), 0, f"p{i}"])
    shapes = ["5S145M", "3I147M", "2S3I95M50M", "10H140M", "75M2D75M", "60M1I30M4D59M", "40M300N60M50S", "1M1D1M1D148M",
              "20=5X30=1I10X2D84=", "150S", "4I", "30M5000N30M", "7M1I" * 40, "2M1D" * 70 + "10M", "148M2S", "1S1M1S"]
    for i in range(1200):
        p = int(rng.integers(0, L - 6000))
        reads.append([p, shapes[i % len(shapes)], int(rng.choice([0, 5, 10, 30, 60])), int(rng.choice([10, 20, 30])), 0, f"s{i}"])
    for i, (p, cig) in enumerate([(100, "70000M"), (2047, "66000M5D3000M2I100M"), (4000, "131071M"), (90_000, "10S65535M1D65536M"),
                                  (90_001, "65536M"), (120_000, "100M70000N100M")]):
        reads.append([p, cig, [60, 60, 3, 60, 10, 60][i], [30, 22, 30, 19, 40, 30][i], 0, f"w{i}"])
    reads.sort(key=lambda r: r[0])
    rec = ContigRecords.from_reads(reads)
    # every seventh read loses the tail of its quality string, every 31st all of it
    keep = np.ones(rec.qual.shape[0], bool)
    qoff = rec.qual_off.astype(np.int64)
    newoff = [0]
    for i in range(rec.n):
        a, b = int(qoff[i]), int(qoff[i + 1])
        cut = (b - a) if i % 31 == 0 else ((b - a) // 3 if i % 7 == 0 else 0)
        keep[b - cut:b] = False
        newoff.append(newoff[-1] + (b - a - cut))
    rec.qual = np.ascontiguousarray(rec.qual[keep]); rec.qual_off = np.asarray(newoff, np.uint64)
    rec.validate()
    assert not short_form or rec.cigar.shape[0] < 8 * rec.n     # the short-read form
    return L, rec, synth.make_reference(L, seed + 1, lowercase=True)


@pytest.mark.parametrize("opts", [dict(), dict(min_mapping_quality=0, min_base_quality=0, min_depth=1),
                                  dict(min_mapping_quality=31, min_base_quality=25, max_depth=20, max_low_mapq=5)])
def test_record_form_covers_every_read_shape(opts, tmp_path):
    L, rec, ref = record_shapes_contig()
    compare([("chrS", 4, L, ref, rec)], opts, tmp_path, "recshapes")


_REC_CHUNK_CASE = r"""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import test_gpu_parity as T
from decodingustools_amd import synth
L, rec, ref = T.record_shapes_contig(L=120_000, n_plain=1500)
T.compare([("chrS", 4, L, ref, rec)], dict(min_mapping_quality=0), pathlib.Path(tempfile.mkdtemp()), "recchunk_" + os.environ["DUT_REC_CHUNK"])
rec = synth.adversarial_contig(5000, 1200, 1004, max_len=300, overhang=True)
T.compare([("chrA", 1, 5000, synth.make_reference(5000, 54), rec)], dict(min_depth=2, min_depth_for_low_mapq=3), pathlib.Path(tempfile.mkdtemp()), "recchunk_adv")
print("REC_CHUNK_OK")
"""


@pytest.mark.parametrize("chunk", ["16", "48", "1040"])
def test_records_through_small_pinned_buffers(chunk):
    """The records go to HBM through pinned buffers of DUT_REC_CHUNK bytes, each filled from the record number it
    starts at: with 1, 3 or 65 records per buffer the seams fall between the head of a read and its pieces and inside the
    pieces of one long run.  (The knob is read once per process.)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DUT_REC_CHUNK=chunk)
    r = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}\n" + _REC_CHUNK_CASE], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "REC_CHUNK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_tiled_pushes_on_a_long_lived_engine(tmp_path):
    """The pass bits of a tile start and end inside words of the contig's bit array (byte forms: small tiles are staged,
    large ones go directly); a context that keeps its buffers from one contig to the next (larger, then smaller, then
    larger) must not read stale data: tiled pushes on a long-lived engine give what a single push on a fresh engine gives
    (which the other tests hold against the oracle)."""
    opt = CallableOptions()
    contigs = []
    for k, (L, depth, seed) in enumerate([(400_000, 30, 11), (90_000, 45, 12), (250_000, 12, 13)]):
        contigs.append((k, L, synth.make_reference(L, seed + 100), synth.short_read_contig(L, depth, seed)))

    def push(eng, r):
        eng.push_reads(r.pos, r.mapq, r.cigar_off, r.cigar, r.qual_off, r.qual)
    want = []
    for tid, L, ref, rec in contigs:
        with Engine(opt, 0) as eng:
            eng.contig_begin(tid, L, ref); push(eng, rec)
            want.append(eng.contig_finish())
    with Engine(opt, 0) as eng:
        for (tid, L, ref, rec), w in zip(contigs, want):
            eng.contig_begin(tid, L, ref)
            cuts = [0, 1, 2, 1000, 1001, rec.n // 3, rec.n // 3 + 7, rec.n]
            for a, b in zip(cuts[:-1], cuts[1:]):
                if b > a:
                    push(eng, rec.slice(a, b))
            got = eng.contig_finish()
            assert got.as_dict() == w.as_dict() and np.array_equal(got.intervals, w.intervals), tid
            lay = eng.contig_layout()
            assert lay["form"] == (3 if os.environ.get("DUT_QUAL_FORM") != "bytes" else 0) and lay["n_reads"] == rec.n
            eng.contig_run()
            again = eng.contig_collect()
            assert again.as_dict() == w.as_dict() and np.array_equal(again.intervals, w.intervals), tid


def test_a_long_read_shaped_tile_in_a_short_read_contig(tmp_path):
    """cl_push_reads counts a read's records while it walks its CIGAR -- except in tiles of 8 or more operations per read,
    which would get a long-read form on their own.  When such a tile is part of a contig that still averages fewer than 8
    (the short-read form), cl_contig_upload makes up for the count: same results as one push, and as the oracle."""
    L = 60_000
    rng = np.random.default_rng(5)
    reads = [[int(p), "150M", 60, 30, 0, f"a{i}"] for i, p in enumerate(np.sort(rng.integers(0, 40_000, 3000)))]
    tail = [[int(p), "5M1I5M1D" * 12 + "20M", int(rng.choice([5, 60])), 25, 0, f"b{i}"] for i, p in enumerate(np.sort(rng.integers(40_000, L - 400, 25)))]
    rec = ContigRecords.from_reads(reads + tail)
    assert rec.cigar.shape[0] < 8 * rec.n
    ref = synth.make_reference(L, 6)
    compare([("chrT", 2, L, ref, rec)], dict(), tmp_path, "mixedtile")
    opt = CallableOptions()

    def push(eng, r):
        eng.push_reads(r.pos, r.mapq, r.cigar_off, r.cigar, r.qual_off, r.qual)
    with Engine(opt, 0) as eng:
        eng.contig_begin(2, L, ref); push(eng, rec)
        want = eng.contig_finish()
        eng.contig_begin(2, L, ref)
        push(eng, rec.slice(0, len(reads))); push(eng, rec.slice(len(reads), rec.n))     # the second tile: 49 operations per read
        got = eng.contig_finish()
    assert got.as_dict() == want.as_dict() and np.array_equal(got.intervals, want.intervals)


def test_run_table_window_whose_last_piece_belongs_to_the_first_read(tmp_path):
    """Found by tools/fuzz_parity.py (seed 300026): a sparse long-read contig whose first window holds pieces of the
    contig's FIRST read only, starting 865 positions into the window.  The lanes behind a window's last entry repeat
    that entry with the valid bit cleared; when they cleared the whole word instead, their (unused) quality loads went to
    unit 0 of the window -- 865 bytes in front of the quality array: a memory access fault when that page is not mapped."""
    L = 40_000
    rec = _eqx_split(synth.long_read_contig(L, 5, 300026), 300026)
    assert rec.cigar.shape[0] >= 8 * rec.n and rec.qual.shape[0] < 56 * rec.cigar.shape[0] and int(rec.pos[0]) > 64
    ref = synth.make_reference(L, 300027)
    compare([("chrF", 1, L, ref, rec)], dict(min_depth=4, max_depth=0, min_mapping_quality=1, min_base_quality=255, min_depth_for_low_mapq=5,
                                             max_low_mapq=1, max_low_mapq_fraction=0.5), tmp_path, "fuzz300026")
    compare([("chrF", 1, L, ref, rec)], dict(), tmp_path, "fuzz300026b")


# ---- the pass-bit form: rows of bits built on the host at upload (pass_rows.h), counted bit-sliced by k_pileup_rows ----
_ROW_CHUNK_CASE = r"""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import test_gpu_parity as T
from decodingustools_amd import synth
L = 150_000
rec = synth.long_read_contig(L, 50, synth.seed_for(3, 31))
T.compare([("chrR", 5, L, synth.make_reference(L, synth.seed_for(3, 31)), rec)], dict(), pathlib.Path(tempfile.mkdtemp()), "rows_long_" + os.environ["DUT_ROW_CHUNK"])
L = 200_000
T.compare([("chr21", 20, L, synth.make_reference(L, 7), synth.short_read_contig(L, 30, 8))], dict(), pathlib.Path(tempfile.mkdtemp()), "rows_short_" + os.environ["DUT_ROW_CHUNK"])
print("ROW_CHUNK_OK")
"""


@pytest.mark.parametrize("chunk", ["1", "9", "64"])
def test_rows_through_small_pinned_buffers(chunk):
    """The groups of a window go to HBM through pinned buffers of DUT_ROW_CHUNK groups (1 KB each): 1 -- every window is
    larger than a buffer and travels as a block of its own; 9 -- a buffer holds about one ordinary window, the buffer-full
    path with the window started over; 64 -- a few windows per buffer.  (The knob is read once per process.)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DUT_ROW_CHUNK=chunk)
    r = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}\n" + _ROW_CHUNK_CASE], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ROW_CHUNK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("depth,planes", [(200, 8), (300, 16), (66_000, 32)])
def test_counter_planes_follow_the_deepest_window(depth, planes, tmp_path):
    """k_pileup_rows counts in 8 bit planes while no window has more than 255 rows (63 groups of 4), in 16 beyond; the
    depth thresholds are compared bit-sliced, so thresholds beyond what the planes can count to must also work."""
    L = 7000
    rng = np.random.default_rng(depth)
    reads = [[int(p), "120M" if depth < 1000 else "20M", int(rng.choice([10, 20, 60, 60])), int(rng.choice([10, 20, 40])), 0, f"d{i}"]
             for i, p in enumerate(np.sort(rng.integers(2000, 2060 if depth < 1000 else 2004, depth)))]
    reads += [[int(p), "100M", 60, 30, 0, f"e{i}"] for i, p in enumerate(np.sort(rng.integers(0, L - 100, 300)))]
    reads.sort(key=lambda r: r[0])
    rec = ContigRecords.from_reads(reads)
    ref = synth.make_reference(L, 5)
    if depth < 1000:
        trials = (dict(), dict(min_depth=150, max_depth=450), dict(min_depth=256, max_depth=499), dict(min_depth=70000, max_depth=100000),
                  dict(min_depth=0, max_depth=255, min_base_quality=0, min_mapping_quality=0))
    else:          # (the depth cap of the admission rule is max_depth: lifted so that all 66 000 reads are in the pileup)
        trials = (dict(max_depth=1_000_000), dict(min_depth=65_536, max_depth=1_000_000))
    for o in trials:
        compare([("chrD", 3, L, ref, rec)], o, tmp_path, f"planes{depth}")
    opt = _opts(trials[0])
    with Engine(opt, 0) as eng:
        eng.contig_begin(3, L, ref)
        eng.push_reads(rec.pos, rec.mapq, rec.cigar_off, rec.cigar, rec.qual_off, rec.qual)
        eng.contig_finish()
        lay = eng.contig_layout()
    assert lay["form"] == 3 and lay["counter_planes"] == planes, lay
    assert lay["n_qual"] == rec.qual.shape[0] and lay["row_groups"] >= lay["max_groups"] > 0


def test_coverage_files_over_several_contexts_of_one_device(tmp_path):
    """dut_coverage_files_multi (the multi-device path BELOW Python: one host thread, one reader pair and one engine
    context per entry of `devices`, contigs dealt by LPT on the index's mapped-read counts, BED written by the caller in
    tid order): with devices = [0, 0] and [0, 0, 0] -- several contexts on this box's one GPU -- the BED, the summary JSON
    and the figures are byte for byte those of the one-device call and of the oracle; an error in a later contig comes
    back as the serial loop's would; the command line tool takes --devices."""
    import subprocess
    from bamio import write_bam, write_fasta
    from decodingustools_amd import EngineError, build as _b
    from decodingustools_amd.bam import coverage_files
    names = ["chr1", "chr2", "chr3", "chrX", "chrY", "chrM"]
    lens = [150_000, 60_000, 90_000, 30_000, 45_000, 16_569]
    recs = {0: synth.short_read_contig(lens[0], 30, 900), 1: synth.adversarial_contig(lens[1], 3000, 901, deep=True),
            2: synth.long_read_contig(lens[2], 20, 902), 4: synth.short_read_contig(lens[4], 12, 904), 5: synth.short_read_contig(lens[5], 20, 905)}
    refs = [synth.make_reference(l, 950 + i, lowercase=(i == 5)) for i, l in enumerate(lens)]
    bam = str(tmp_path / "m.bam"); fa = str(tmp_path / "m.fa")
    write_bam(bam, list(zip(names, lens)), recs, block_every=800)
    write_fasta(fa, list(zip(names, refs)))
    contigs = [(n, t, lens[t], refs[t], recs.get(t, ContigRecords.empty())) for t, n in enumerate(names)]
    _, o_bed = oracle_run(contigs, make_options({}), str(tmp_path / "o.bed"))
    one = tmp_path / "one"; one.mkdir()
    coverage_files(bam, fa, str(one / "g.bed"), str(one / "s.json"), CallableOptions(), output_summary=str(one / "r.html"))
    assert open(one / "g.bed").read() == o_bed
    import json as _json
    for devs in ([0, 0], [0, 0, 0], [0]):
        d = tmp_path / ("multi%d" % len(devs)); d.mkdir()
        coverage_files(bam, fa, str(d / "g.bed"), str(d / "s.json"), CallableOptions(), output_summary=str(d / "r.html"), devices=devs)
        assert open(d / "g.bed").read() == o_bed, devs
        a, b = _json.load(open(one / "s.json")), _json.load(open(d / "s.json"))
        a["files"] = b["files"] = None                              # (the paths differ)
        assert a == b, devs
        for n in names:
            f1, f2 = one / f"{n}_coverage.svg", d / f"{n}_coverage.svg"
            assert f1.exists() == f2.exists() and (not f1.exists() or f1.read_bytes() == f2.read_bytes()), (devs, n)
    # -L subset, through the tool
    out = str(tmp_path / "cli.bed")
    r = subprocess.run([_b.CLI, "coverage", bam, "-r", fa, "-o", out, "--devices", "0,0", "-L", "chr2", "-L", "chrY", "-L", "chrM"],
                       cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    _, o_bed2 = oracle_run([contigs[1], contigs[4], contigs[5]], make_options({}), str(tmp_path / "o2.bed"))
    assert open(out).read() == o_bed2
    # the same tool giving everything back before it leaves (what a library caller's process does)
    r = subprocess.run([_b.CLI, "coverage", bam, "-r", fa, "-o", out, "--devices", "0,0"], cwd=str(tmp_path), capture_output=True, text=True,
                       env=dict(os.environ, DUT_CLI_TEARDOWN="1"))
    assert r.returncode == 0 and open(out).read() == o_bed, r.stderr
    # a file cut off inside a later contig: the first error in tid order is reported, nothing hangs
    import shutil
    bam_cut = str(tmp_path / "cut.bam")
    data = open(bam, "rb").read()
    open(bam_cut, "wb").write(data[:int(len(data) * 0.7)])
    shutil.copy(bam + ".bai", bam_cut + ".bai")
    with pytest.raises(EngineError) as ei:
        coverage_files(bam_cut, fa, str(tmp_path / "g2.bed"), None, CallableOptions(), devices=[0, 0])
    assert "Error processing contig" in str(ei.value) or "BAM" in str(ei.value)
    with pytest.raises(EngineError):
        coverage_files(bam, fa, str(tmp_path / "g3.bed"), None, CallableOptions(), devices=[0, 99])      # no such device


@pytest.mark.parametrize("what,form", [("rows", "bits"), ("rec", "bytes"), ("runtab", "bytes")])
def test_an_index_outside_its_array_is_an_error_return_not_a_fault(what, form, monkeypatch):
    """What the pileup kernels index is checked on the host at every upload, in the product build: a window's range of row
    groups against the resident rows, every record's run and every run-table piece -- the units the kernel loads for it,
    clamped lanes included -- against the padded quality array.  DUT_FAULT_INJECT makes the named builder produce one
    index outside its array (the situation behind the memory access fault of fuzz seed 300026): the upload must refuse
    with CL_ERR_RANGE, launch nothing, and the context must take the next contig as if nothing had happened."""
    from decodingustools_amd import EngineError
    if form == "bytes":
        monkeypatch.setenv("DUT_QUAL_FORM", "bytes")
    L = 120_000
    rec = synth.long_read_contig(L, 20, 41) if what == "runtab" else synth.short_read_contig(L, 30, 41)
    ref = synth.make_reference(L, 42)
    opt = CallableOptions()
    with Engine(opt, 0) as eng:
        eng.contig_begin(1, L, ref)
        eng.push_reads(rec.pos, rec.mapq, rec.cigar_off, rec.cigar, rec.qual_off, rec.qual)
        want = eng.contig_finish()
        assert eng.contig_layout()["form"] == {"rows": 3, "rec": 0, "runtab": 2}[what]
        monkeypatch.setenv("DUT_FAULT_INJECT", what)
        eng.contig_begin(1, L, ref)
        eng.push_reads(rec.pos, rec.mapq, rec.cigar_off, rec.cigar, rec.qual_off, rec.qual)
        with pytest.raises(EngineError) as ei:
            eng.contig_finish()
        assert ei.value.status == -6 and "outside the resident" in str(ei.value)
        monkeypatch.delenv("DUT_FAULT_INJECT")
        eng.contig_begin(1, L, ref)
        eng.push_reads(rec.pos, rec.mapq, rec.cigar_off, rec.cigar, rec.qual_off, rec.qual)
        got = eng.contig_finish()
    assert got.as_dict() == want.as_dict() and np.array_equal(got.intervals, want.intervals)


@pytest.mark.parametrize("thr", [0, 20, 128, 255])
def test_quality_packing_on_this_hosts_cpu(thr):
    """The base-quality test is taken on the host (qual_pack.cpp): scalar, SSE2 and -- where the CPU has it, as the GPU
    boxes' CPUs do and the build container's does not -- AVX2 must agree bit for bit, sums included."""
    import test_qual_rows as TQ
    TQ.test_pass_bits_and_sums_agree_at_every_level(thr)


def test_one_long_indel_rich_read_inside_a_short_read_contig(tmp_path):
    """A hybrid contig: 30x of 150-base reads (fewer than 8 operations per read on average: the record forms) and ONE
    100 kb read with an indel every ~12 bases (some 16 000 operations, wide: it spans 49 windows).  Pass-bit form: the
    read is one head record and a string of bits like any other; byte form: all its ~8 000 piece records are candidates
    of every window it spans -- slower there, the same result everywhere."""
    L = 400_000
    short = synth.short_read_contig(L, 30, 5150)
    rng = np.random.default_rng(5151)
    ops = []
    total = 0
    while total < 100_000:
        m = int(rng.integers(4, 22)); ops.append(f"{m}M"); total += m
        k = int(rng.integers(1, 4)); ops.append(f"{k}{'ID'[int(rng.integers(0, 2))]}")
    cig = "".join(ops) + "30M"
    reads = [(int(short.pos[i]), None) for i in range(short.n)]
    long_rec = ContigRecords.from_reads([(150_000, cig, 60, 30, 0, "long1")])
    # merge: the long read goes where its position sorts
    k = int(np.searchsorted(short.pos, 150_000))
    def cat(a, b, c):
        return np.concatenate([a, b, c])
    co_s, qo_s, no_s = short.cigar_off.astype(np.int64), short.qual_off.astype(np.int64), short.qname_off.astype(np.int64)
    nc, nq, nn = long_rec.cigar.shape[0], long_rec.qual.shape[0], long_rec.qname.shape[0]
    rec = ContigRecords(
        pos=cat(short.pos[:k], long_rec.pos, short.pos[k:]).astype(np.int32), flag=cat(short.flag[:k], long_rec.flag, short.flag[k:]).astype(np.uint16),
        mapq=cat(short.mapq[:k], long_rec.mapq, short.mapq[k:]).astype(np.uint8),
        cigar_off=cat(co_s[:k + 1], co_s[k:k + 1] + nc, co_s[k + 1:] + nc).astype(np.uint32),
        cigar=cat(short.cigar[:co_s[k]], long_rec.cigar, short.cigar[co_s[k]:]).astype(np.uint32),
        qual_off=cat(qo_s[:k + 1], qo_s[k:k + 1] + nq, qo_s[k + 1:] + nq).astype(np.uint64),
        qual=cat(short.qual[:qo_s[k]], long_rec.qual, short.qual[qo_s[k]:]).astype(np.uint8),
        qname_off=cat(no_s[:k + 1], no_s[k:k + 1] + nn, no_s[k + 1:] + nn).astype(np.uint32),
        qname=cat(short.qname[:no_s[k]], long_rec.qname, short.qname[no_s[k]:]).astype(np.uint8)).validate()
    assert rec.cigar.shape[0] < 8 * rec.n and rec.n == short.n + 1
    ref = synth.make_reference(L, 5152)
    compare([("chrH", 7, L, ref, rec)], dict(), tmp_path, "hybrid", dump=False)


def _packed(rec, thr):
    """What cl_push_reads_bits takes instead of the quality bytes (include/callable_loci.h), restated with numpy."""
    nq = int(rec.qual_off[-1])
    bits = np.zeros((nq + 63) // 64 + 1, np.uint64)
    by = np.packbits(rec.qual[:nq] >= thr, bitorder="little")
    bits.view(np.uint8)[:by.shape[0]] = by
    sums = np.zeros(rec.n, np.uint32)
    for i in range(rec.n):
        q = rec.qual[int(rec.qual_off[i]):int(rec.qual_off[i + 1])].astype(np.int64)
        y = 0; tot = 0
        for cw in rec.cigar[int(rec.cigar_off[i]):int(rec.cigar_off[i + 1])]:
            op, ln = int(cw) & 15, int(cw) >> 4
            if op in (0, 7, 8):
                seg = q[y:y + ln]; tot += int(seg[seg >= thr].sum())
            if op in (0, 1, 4, 7, 8):
                y += ln
        sums[i] = tot
    return bits, sums


@pytest.mark.parametrize("shape", ["short", "long", "shapes"])
def test_the_packed_pass_bitmask_variant_gives_what_the_bytes_give(shape, tmp_path, monkeypatch):
    """cl_push_reads_bits (SURVEY 8b's packed variant: the caller has taken the base-quality test) against cl_push_reads on
    the same reads -- summary, runs and per-position counters --, in one tile, in ragged tiles and mixed with byte tiles;
    a context in the byte forms refuses it."""
    if shape == "short":
        L = 200_000; rec = synth.short_read_contig(L, 25, 71)
    elif shape == "long":
        L = 150_000; rec = synth.long_read_contig(L, 10, 72)
    else:
        L, rec, _ = record_shapes_contig(200_000, seed=73, n_plain=1500)
    assert not (rec.pos >= L).any() and not (rec.pos < 0).any()
    ref = synth.make_reference(L, 5)
    opt = CallableOptions(min_base_quality=25)
    bits, sums = _packed(rec, 25)
    with Engine(opt, 0) as eng:
        eng.contig_begin(0, L, ref)
        eng.push_reads(rec.pos, rec.mapq, rec.cigar_off, rec.cigar, rec.qual_off, rec.qual)
        want = eng.contig_finish()
        want_d = eng.debug_depths(int(want.summary.extent))
        for cuts, mixed in (([0, rec.n], False), ([0, 1, 7, rec.n // 2, rec.n // 2 + 3, rec.n], False), ([0, rec.n // 3, 2 * rec.n // 3, rec.n], True)):
            eng.contig_begin(0, L, ref)
            for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
                if mixed and k == 1:
                    eng.push_reads(rec.pos[a:b], rec.mapq[a:b], rec.cigar_off[a:b + 1], rec.cigar, rec.qual_off[a:b + 1], rec.qual)
                else:
                    eng.push_reads_bits(rec.pos[a:b], rec.mapq[a:b], rec.cigar_off[a:b + 1], rec.cigar, rec.qual_off[a:b + 1], bits, sums[a:b])
            got = eng.contig_finish()
            assert got.as_dict() == want.as_dict() and np.array_equal(got.intervals, want.intervals), cuts
            for a, b in zip(want_d, eng.debug_depths(int(got.summary.extent))):
                assert np.array_equal(a, b)
    monkeypatch.setenv("DUT_QUAL_FORM", "bytes")
    with Engine(opt, 0) as eng:
        eng.contig_begin(0, L, ref)
        with pytest.raises(Exception) as e:
            eng.push_reads_bits(rec.pos, rec.mapq, rec.cigar_off, rec.cigar, rec.qual_off, bits, sums)
        assert getattr(e.value, "status", None) == -1             # CL_ERR_INVALID
        eng.push_reads(rec.pos, rec.mapq, rec.cigar_off, rec.cigar, rec.qual_off, rec.qual)
        got = eng.contig_finish()
        assert got.as_dict() == want.as_dict() and np.array_equal(got.intervals, want.intervals)


@pytest.mark.parametrize("head_span", ["1", "37", "1000", "70000"])
def test_spans_cut_into_several_heads(head_span, tmp_path, monkeypatch):
    """k_pileup_rows takes a read's reference span from 8-byte heads of at most 2^31 - 1 positions each; a longer span is cut
    into several heads (only a contig beyond 2 Gb can hold one: DUT_HEAD_SPAN puts the seams into ordinary reads).  The
    +-1 scatter of the pieces is that of the whole span, the reads' separable sums come from the host's walk: everything
    as the oracle has it, including wide reads (their heads come through the wide list) and heads past their window."""
    monkeypatch.setenv("DUT_HEAD_SPAN", head_span)
    L, rec, ref = record_shapes_contig(L=150_000, seed=99, n_plain=1200, short_form=False)
    compare([("chrS", 4, L, ref, rec)], dict(min_mapping_quality=5), tmp_path, "heads_" + head_span)
    if head_span in ("37", "1000"):
        rec = synth.long_read_contig(60_000, 8, 17)
        compare([("chrL", 2, 60_000, synth.make_reference(60_000, 3), rec)], None, tmp_path, "heads_long_" + head_span)


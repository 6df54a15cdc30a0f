"""The C oracle (pileup columns over a linked list, the way htslib walks them) against an independent
numpy restatement that expands every read on its own (oracle/bruteforce.py): two formulations of the
reference's semantics that share no code must agree on every counter, state, sum and BED byte."""
import numpy as np
import pytest

from helpers import contig_inputs, load_kats, make_options, oracle_run
from decodingustools_amd import synth
from decodingustools_amd.records import ContigRecords
from oracle import bruteforce as BF

KATS = load_kats()


def check(contigs, opt, tmp_path):
    o_res, o_bed = oracle_run(contigs, opt, str(tmp_path / "o.bed"), dump=True)
    bf = [BF.contig(opt, name, length, ref, rec) for name, _, length, ref, rec in contigs]
    for (name, _, length, _, _), b in zip(contigs, bf):
        raw, qc, low, state, extent = o_res[name]["dumps"]
        assert extent == b["extent"], name
        assert np.array_equal(raw, b["raw"]) and np.array_equal(qc, b["qc"]) and np.array_equal(low, b["low"]), name
        assert np.array_equal(state, b["state"]), name
        st = o_res[name]["stats"]
        for k in ("n_covered_bases", "summed_coverage", "summed_baseq", "summed_mapq", "quality_bases", "n_reads"):
            assert st[k] == b[k], (name, k)
        assert o_res[name]["state_counts"] == b["state_counts"], name
    assert o_bed == BF.bed(bf)


@pytest.mark.parametrize("case", KATS["cases"], ids=[c["name"] for c in KATS["cases"]])
def test_kats(case, tmp_path):
    opt = make_options({**KATS["default_options"], **case.get("options", {})})
    contigs = []
    for i, c in enumerate(case["contigs"]):
        rec, ref = contig_inputs(c)
        contigs.append((c["name"], c.get("tid", i), c["len"], ref, rec))
    if case.get("cap_bites"):
        pytest.skip("the numpy restatement leaves the depth cap to the C oracle")
    bf = [BF.contig(opt, name, length, ref, rec) for name, _, length, ref, rec in contigs]
    assert BF.bed(bf) == case["bed"]                       # the numpy restatement passes the hand-derived KATs on its own
    check(contigs, opt, tmp_path)


@pytest.mark.parametrize("seed", range(6))
def test_random_contigs(seed, tmp_path):
    rng = np.random.default_rng(300 + seed)
    opt = make_options(dict(max_depth=1_000_000, min_depth=int(rng.integers(0, 8)), min_base_quality=int(rng.choice([0, 13, 20, 40, 200])),
                            min_mapping_quality=int(rng.choice([0, 10, 30])), max_low_mapq=int(rng.choice([0, 1, 10])),
                            min_depth_for_low_mapq=int(rng.integers(0, 12)), max_low_mapq_fraction=float(rng.choice([0.0, 0.1, 0.4]))))
    contigs = []
    for t in range(3):
        L = int(rng.choice([0, 700, 2048, 5000]))
        n = int(rng.integers(0, 500)) if L else 0
        rec = synth.adversarial_contig(L, n, 900 + 10 * seed + t, deep=bool(t == 1), overhang=bool(t == 2)) if n else ContigRecords.empty()
        ref = synth.make_reference(L, 40 + t, lowercase=bool(t == 0)) if rng.random() < 0.85 else None
        contigs.append((f"c{t}", t, L, ref, rec))
    check(contigs, opt, tmp_path)


def test_short_and_long_read_generators(tmp_path):
    L = 30_000
    opt = make_options(dict(max_depth=1_000_000))
    check([("s", 0, L, synth.make_reference(L, 1), synth.short_read_contig(L, 25, 2))], opt, tmp_path)
    check([("l", 0, L, synth.make_reference(L, 3), synth.long_read_contig(L, 12, 4))], opt, tmp_path)


def test_record_shapes_contig(tmp_path):
    """The contig the GPU tests of the record form use (tests/test_gpu_parity.py: record_shapes_contig): leading clips
    and insertions, runs longer than 65 535 bases, truncated and absent quality strings -- the two restatements agree on
    it before the device is held against the C oracle."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("_tgp", os.path.join(os.path.dirname(__file__), "test_gpu_parity.py"))
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    L, rec, ref = T.record_shapes_contig(L=120_000, n_plain=1500)
    for o in (dict(max_depth=1_000_000), dict(max_depth=1_000_000, min_mapping_quality=0, min_base_quality=0, min_depth=1)):
        check([("chrS", 4, L, ref, rec)], make_options(o), tmp_path)

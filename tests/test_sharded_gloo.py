"""The N > 1 path on CPU: two processes, gloo backend.  The per-contig work is done by the oracle
here (there is no GPU and the product has no CPU fallback); what is under test is the product's
sharding, summary all_gather, run-list gather and BED assembly (decodingustools_amd/coverage.py)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from helpers import make_options, oracle_run
from decodingustools_amd import ContigProfiler, synth
from decodingustools_amd.callable_loci import CallableOptions
from decodingustools_amd.coverage import (ApiError, ContigInput, ContigOutcome, CoverageInput, analyze_sharded,
                                          initialize_contig_stats, lpt_assignment, validate_contig_selection)


def _contigs():
    out = []
    names = ["chr1", "chr2", "chr10", "chrX", "chrM"]
    lens = [5000, 3100, 4200, 2048, 700]
    for t, (nm, L) in enumerate(zip(names, lens)):
        rec = synth.adversarial_contig(L, 150 + 40 * t, 900 + t, max_len=160)
        out.append(ContigInput(nm, L, rec, synth.make_reference(L, 30 + t, lowercase=(t == 4))))
    return out


def _intervals(state):
    cut = np.flatnonzero(np.diff(state.astype(np.int32)) != 0) + 1
    starts = np.concatenate([[0], cut]); ends = np.concatenate([cut, [state.shape[0]]])
    return np.stack([starts, ends, state[starts]], axis=1).astype(np.uint32)


def _oracle_process(opt):
    def run(tid, c):
        with tempfile.TemporaryDirectory() as d:
            prof = oracle.Profiler(os.path.join(d, "x.bed"))
            st, dumps = oracle.process_single_contig(prof, opt, c.name, tid, c.length, c.ref, c.records, dump=True)
            counts = prof.contig_counts(c.name)
            prof.close()
        cp = ContigProfiler(c.name, c.length, st["n_covered_bases"], st["summed_coverage"], st["summed_baseq"],
                            st["summed_mapq"], st["quality_bases"], st["n_reads"])
        return ContigOutcome(tid=tid, stats=cp, state_counts=counts, intervals=_intervals(dumps[3]))
    return run


def _worker(rank, world, port, bed_path, selected, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        opt = CallableOptions(min_depth=2, min_depth_for_low_mapq=3)
        inp = CoverageInput(contigs=_contigs(), options=opt, selected=selected, output_bed=bed_path)
        out = analyze_sharded(inp, rank, world, _oracle_process(make_options(dict(min_depth=2, min_depth_for_low_mapq=3))))
        if rank == 0:
            q.put(out.export)
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def _failing_worker(rank, world, port, bed_path, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        opt = CallableOptions()
        inp = CoverageInput(contigs=_contigs(), options=opt, selected=None, output_bed=bed_path)
        good = _oracle_process(make_options({}))

        def run(tid, c):
            if c.name == "chr2":                     # whichever rank is dealt chr2 fails there
                raise RuntimeError("malformed CIGAR in record 17")
            return good(tid, c)
        try:
            analyze_sharded(inp, rank, world, run)
            q.put((rank, "no error"))
        except ApiError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_stops_every_rank_with_the_reference_error_text(tmp_path):
    """One rank's contig is corrupt: no rank may be left waiting in a collective, every rank raises
    'Error processing contig: ...' (api/coverage.rs:251) and no BED is written."""
    bed = str(tmp_path / "never.bed")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, bed, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert set(got) == {0, 1}
    for msg in got.values():
        assert msg.startswith("Error processing contig: malformed CIGAR in record 17"), msg
    assert not os.path.exists(bed)
    # a single process reports the same text
    inp = CoverageInput(contigs=_contigs(), options=CallableOptions(), output_bed=bed)
    with pytest.raises(ApiError, match="Error processing contig: boom"):
        analyze_sharded(inp, 0, 1, lambda tid, c: (_ for _ in ()).throw(RuntimeError("boom")))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("selected", [None, ["chr2", "chrM", "chrX"]])
def test_two_ranks_equal_the_single_process_bed(tmp_path, selected):
    bed = str(tmp_path / "sharded.bed")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bed, selected, q)) for r in range(2)]
    for p in procs:
        p.start()
    export = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    contigs = _contigs()
    keep = [(c.name, t, c.length, c.ref, c.records) for t, c in enumerate(contigs)
            if selected is None or c.name in selected]
    o_res, o_bed = oracle_run(keep, make_options(dict(min_depth=2, min_depth_for_low_mapq=3)), str(tmp_path / "o.bed"))
    assert open(bed).read() == o_bed
    # report order and numbers (report.rs:37-38, 88-126)
    names = [c["name"] for c in export["contigs"]]
    assert names == [n for n in ["chr1", "chr2", "chr10", "chrX", "chrM"] if selected is None or n in selected]
    order = [c for c in keep]
    order.sort(key=lambda c: names.index(c[0]))
    og = oracle.genome_summary([o_res[c[0]]["stats"] for c in order], [o_res[c[0]]["state_counts"][1] for c in order])
    assert export["summary"]["callable_percentage"] == og["callable_percentage"]
    assert export["summary"]["average_depth"] == og["average_depth"]
    assert export["quality_metrics"]["q30_percentage"] == og["q30_percentage"]
    assert export["total_unique_reads"] == og["total_unique_reads"]


def test_lpt_assignment_is_balanced_and_deterministic():
    lens = [n for _, n in synth.HG38_PRIMARY]
    for world in (1, 2, 4, 8):
        r = lpt_assignment(lens, world)
        assert r == lpt_assignment(lens, world)
        load = [sum(l for l, rr in zip(lens, r) if rr == k) for k in range(world)]
        assert max(load) <= sum(lens) / world * 1.08 + 1      # hg38 over 8 GPUs: within 8 % of ideal
        assert set(r) == set(range(world))


def test_contig_selection_errors_like_the_reference():
    inp = CoverageInput(contigs=_contigs(), selected=["chrNope"])
    stats = initialize_contig_stats(inp)
    with pytest.raises(ApiError, match="None of the specified contigs"):
        validate_contig_selection(stats, inp)
    inp2 = CoverageInput(contigs=_contigs(), selected=["chrNope", "chr2"])   # unknown names are ignored
    assert list(initialize_contig_stats(inp2)) == [1]


def _shard_worker(rank, world, port, q):
    """ResidentShard's exchange (decodingustools_amd/wgs.py) with stand-in contigs: what is under test is the
    deal, the fixed-size summary rows, the all_gather and the table every rank rebuilds from it."""
    import types
    import torch
    from decodingustools_amd import wgs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        genome = wgs.genome(1.0 / 1000)
        rank_of = wgs.deal(genome, world)
        shard = wgs.ResidentShard(rank, world, genome, rank_of)
        for (tid, name, L), r in zip(genome, rank_of):
            if r == rank:
                words = torch.tensor([tid * 100 + k for k in range(14)], dtype=torch.int64)
                shard.mine.append(types.SimpleNamespace(tid=tid, name=name, length=L, dev_summary=words))
        table = shard.gather_summaries("cpu")
        q.put((rank, sorted(table), [table[t] for t in sorted(table)], shard.bases))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_resident_shard_gathers_every_contigs_summary_on_every_rank(world):
    from decodingustools_amd import wgs
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    genome = wgs.genome(1.0 / 1000)
    for rank, tids, rows, bases in got:
        assert tids == list(range(25))
        assert rows == [[t * 100 + k for k in range(14)] for t in range(25)]
    assert sum(b for _, _, _, b in got) == sum(L for _, _, L in genome)       # every contig has exactly one owner


def test_wgs_genome_is_the_fixed_hg38_input_and_the_deal_is_the_same_everywhere():
    from decodingustools_amd import wgs
    g = wgs.genome()
    assert len(g) == 25 and sum(L for _, _, L in g) == 3_088_286_401 and g[0][1:] == ("chr1", 248_956_422) and g[24][1:] == ("chrM", 16_569)
    for world in (1, 2, 4, 8):
        r = wgs.deal(g, world)
        load = [sum(L for (_, _, L), rr in zip(g, r) if rr == k) for k in range(world)]
        assert max(load) / (sum(load) / world) < 1.08
    assert [L for _, _, L in wgs.genome(1 / 16)][24] == 1035

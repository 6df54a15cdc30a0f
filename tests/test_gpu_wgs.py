"""GPU tests (-m gpu) of BASELINE.json configs[3]: the whole-genome workload -- the 25 hg38 primary contigs
at 30x -- at full length on one engine (size-independent properties + oracle windows), dealt to two ranks
at 1/16 scale against the oracle's BED, and the summary gather through RCCL (single-rank nccl group: the
collective code path of bench.py --gpus N on the one GPU a test box has)."""
import os
import socket
import sys
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle
from helpers import make_options, oracle_run
from decodingustools_amd import (CallableOptions, CallableProfiler, ContigProfiler, Engine, admit_reads,
                                 process_single_contig, synth, wgs)
from decodingustools_amd.records import ContigRecords

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _window_records(rec, a, b, margin=400):
    i0 = int(np.searchsorted(rec.pos, a - margin)); i1 = int(np.searchsorted(rec.pos, b))
    return ContigRecords(pos=rec.pos[i0:i1], flag=rec.flag[i0:i1], mapq=rec.mapq[i0:i1],
                         cigar_off=(rec.cigar_off[i0:i1 + 1] - rec.cigar_off[i0]).astype(np.uint32),
                         cigar=rec.cigar[rec.cigar_off[i0]:rec.cigar_off[i1]],
                         qual_off=(rec.qual_off[i0:i1 + 1] - rec.qual_off[i0]).astype(np.uint64),
                         qual=rec.qual[int(rec.qual_off[i0]):int(rec.qual_off[i1])],
                         qname_off=(rec.qname_off[i0:i1 + 1] - rec.qname_off[i0]).astype(np.uint32),
                         qname=rec.qname[rec.qname_off[i0]:rec.qname_off[i1]])


def _property_checks(name, L, rec, ref, opt, acc, n_names, st, counts, iv):
    """What must hold for any input size: the runs tile [0, L), neighbours differ, run lengths per state equal
    the state counts, REF_N is exactly the N / n bases, the per-read separable sums equal numpy's."""
    assert iv[0, 0] == 0 and iv[-1, 1] == L and np.array_equal(iv[1:, 0], iv[:-1, 1]), name
    assert np.all(iv[1:, 2] != iv[:-1, 2]) and iv[:, 2].max() <= 5, name
    lens = (iv[:, 1] - iv[:, 0]).astype(np.int64)
    per_state = np.bincount(iv[:, 2], weights=lens, minlength=6).astype(np.int64).tolist()
    assert per_state == counts and sum(counts) == L, name
    assert counts[0] == int(np.count_nonzero((ref | 0x20) == ord("n"))), name
    ops = rec.cigar & 15
    lens_c = (rec.cigar >> 4).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(np.where(np.isin(ops, [0, 2, 3, 7, 8]), lens_c, 0))])
    rl = cs[rec.cigar_off[1:].astype(np.int64)] - cs[rec.cigar_off[:-1].astype(np.int64)]
    assert st.summed_coverage == int(rl[acc].sum()), name
    sel = acc & (rec.mapq >= opt.min_mapping_quality)
    assert st.summed_mapq == int((rec.mapq[sel].astype(np.int64) * rl[sel]).sum()), name
    assert st.n_reads == n_names, name
    # covered positions: every run that is not NO_COVERAGE / REF_N is covered; REF_N may be either
    assert per_state[1] + per_state[3] + per_state[4] + per_state[5] <= st.n_covered_bases <= L - per_state[2], name


def test_whole_genome_25_contigs_full_length_one_engine(tmp_path):
    """Every hg38 primary contig at full length and 30x through one engine, one BED writer: the properties above
    on all 25, per-position counters / states against the oracle on windows of chr1 and on the whole of chrM, the
    duplicated last line of every contig but the last in the BED."""
    opt = CallableOptions()
    contigs = wgs.genome(1.0)
    bed_path = str(tmp_path / "wgs.bed")
    n_iv = 0
    total = 0
    with ThreadPoolExecutor(4) as pool, Engine(opt, 0) as eng:
        counter = CallableProfiler(bed_path)
        futs = {}
        nxt = 0

        def top_up():
            nonlocal nxt
            while nxt < len(contigs) and len(futs) < 4:
                futs[nxt] = pool.submit(wgs.make_contig, contigs[nxt][0], contigs[nxt][2], 30.0)
                nxt += 1
        top_up()
        for i, (tid, name, L) in enumerate(contigs):
            rec, ref = futs.pop(i).result()
            top_up()
            acc, n_names = admit_reads(opt, tid, L, rec)
            st = ContigProfiler(name, L)
            process_single_contig(eng, counter, st, opt, tid, rec, ref)
            counts = counter.get_contig_counts(name)
            res = eng.contig_collect()
            assert res.state_counts == counts and res.summary.extent == L
            _property_checks(name, L, rec, ref, opt, acc, n_names, st, counts, res.intervals)
            n_iv += res.intervals.shape[0]
            total += L
            print(f"[wgs test] {name}: {L} bp, {rec.n} reads, {res.intervals.shape[0]} runs ok", file=sys.stderr, flush=True)
            if name in ("chr1", "chrM"):
                raw, qc, low, state = eng.debug_depths(L)
                assert st.summed_coverage == int(raw.astype(np.int64).sum()) and st.n_covered_bases == int(np.count_nonzero(raw))
                assert st.quality_bases == int(qc.astype(np.int64).sum()) and np.all(qc <= raw) and np.all(low <= raw)
                assert np.bincount(state, minlength=6).tolist() == counts
                if name == "chrM":
                    windows = [(0, L)]
                else:
                    rng = np.random.default_rng(11)
                    windows = [(int(a), int(a) + 60_000) for a in rng.integers(20_000, L - 100_000, size=4)] + [(0, 60_000), (L - 60_000, L)]
                for a, b in windows:
                    sub = _window_records(rec, a, b) if (a, b) != (0, L) else rec
                    prof = oracle.Profiler(str(tmp_path / "s.bed"))
                    _, d = oracle.process_single_contig(prof, make_options({}), name, tid, b, ref[:b], sub, dump=True)
                    prof.close()
                    for nm, arr_o, arr_g in (("raw", d[0], raw), ("qc", d[1], qc), ("low", d[2], low), ("state", d[3], state)):
                        assert np.array_equal(arr_o[a:b], arr_g[a:b]), (name, nm, a)
                del raw, qc, low, state
            del rec, ref
        counter.close()
    assert total == 3_088_286_401
    with open(bed_path, "rb") as f:
        n_lines = sum(chunk.count(b"\n") for chunk in iter(lambda: f.read(1 << 24), b""))
    assert n_lines == n_iv + len(contigs) - 1          # callable_profiler.rs:64-66: the last line of every contig but the last, twice


def _sharded_worker(rank, world, port, bed_path, scale, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from decodingustools_amd.coverage import ContigInput, CoverageInput, analyze_sharded, engine_process_contig
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        opt = CallableOptions()
        genome = wgs.genome(scale)
        inp = CoverageInput(contigs=[ContigInput(nm, L, None, None, weight=L) for _, nm, L in genome],
                            options=opt, output_bed=bed_path)
        seen = []
        with Engine(opt, 0) as eng:
            def run(tid, c):
                rec, ref = wgs.make_contig(tid, c.length, 30.0)
                seen.append(tid)
                return engine_process_contig(eng, opt, tid, ContigInput(c.name, c.length, rec, ref))
            out = analyze_sharded(inp, rank, world, run)
        q.put((rank, seen, out.export if rank == 0 else None))
    finally:
        dist.destroy_process_group()


def test_whole_genome_sharded_over_two_ranks_equals_the_oracle_bed(tmp_path):
    """configs[3] at 1/16 scale (193 Mb), LPT-dealt to two processes (gloo; both on the test box's one GPU), each
    generating and running only its own contigs; rank 0's BED and export against the oracle's serial run over
    all 25 contigs (api/coverage.rs:221-236)."""
    import torch.multiprocessing as mp
    scale = 1.0 / 16
    genome = wgs.genome(scale)
    bed = str(tmp_path / "sharded.bed")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, bed, scale, q)) for r in range(2)]
    for p in procs:
        p.start()
    # meanwhile: the oracle, one thread and one BED writer over the contigs in header order like the reference
    with ThreadPoolExecutor(4) as pool:
        made = list(pool.map(lambda c: wgs.make_contig(c[0], c[2], 30.0), genome))
    keep = [(nm, tid, L, ref, rec) for (tid, nm, L), (rec, ref) in zip(genome, made)]
    o_res, o_bed = oracle_run(keep, make_options({}), str(tmp_path / "o.bed"))
    got = [q.get(timeout=900) for _ in range(2)]
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    got.sort()
    rank_of = wgs.deal(genome, 2)
    for r, seen, _ in got:
        assert seen == [t for (t, _, _), rr in zip(genome, rank_of) if rr == r]      # a rank only touches its own contigs
    assert open(bed).read() == o_bed
    export = got[0][2]
    assert export["summary"]["contigs_analyzed"] == 25 and export["summary"]["total_bases"] == sum(L for _, _, L in genome)
    for c in export["contigs"]:
        so = o_res[c["name"]]
        assert c["unique_reads"] == so["stats"]["n_reads"] and c["covered_bases"] == so["stats"]["n_covered_bases"]
        assert c["average_depth"] == so["stats"]["derived"]["average_depth"]
        assert list(c["state_distribution"].values()) == so["state_counts"]
    names = [c["name"] for c in export["contigs"]]
    order = sorted(range(25), key=lambda i: names.index(genome[i][1]))
    og = oracle.genome_summary([o_res[genome[i][1]]["stats"] for i in order], [o_res[genome[i][1]]["state_counts"][1] for i in order])
    assert export["summary"]["callable_percentage"] == og["callable_percentage"]
    assert export["summary"]["average_depth"] == og["average_depth"]
    assert export["total_unique_reads"] == og["total_unique_reads"]


def test_resident_shard_steps_and_gathers_summaries_through_rccl(tmp_path):
    """The loop bench.py --gpus N runs on every rank, here with N = 1 and the nccl (= RCCL) backend: contigs
    resident on engines that share one stream, a step = run them all + the all_gather of the summary records
    straight from HBM (cl_device_summary); the gathered table equals the first pass, step after step."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        opt = CallableOptions()
        stream = torch.cuda.Stream(device=0)
        keep = {}
        with torch.cuda.stream(stream):
            shard = wgs.build_shard(0, 1, 0, opt, depth=30.0, scale=1.0 / 256, stream=stream.cuda_stream, keep_records=keep)
            assert [c.tid for c in shard.mine] == list(range(25))
            for _ in range(3):
                shard.step()
                rows = shard.summary_rows("cuda")
                out = [torch.empty_like(rows)]
                dist.all_gather(out, rows)                           # RCCL, device tensors
            stream.synchronize()
            table = shard.parse(out)
        for c in shard.mine:
            row = table[c.tid]
            assert [int(x) for x in row[:6]] == c.outcome.state_counts
            assert int(row[11]) == c.length and int(row[13]) == c.outcome.intervals.shape[0]
            assert int(row[6]) == c.outcome.stats.n_covered_bases and int(row[10]) == c.outcome.stats.quality_bases
        # and the runs of a few contigs against the oracle
        for tid in (0, 20, 24):
            rec, ref = keep[tid]
            c = shard.mine[tid]
            o_res, o_bed = oracle_run([(c.name, tid, c.length, ref, rec)], make_options({}), str(tmp_path / "o.bed"))
            assert o_res[c.name]["state_counts"] == c.outcome.state_counts
            lines = o_bed.splitlines()
            assert len(lines) == c.outcome.intervals.shape[0]
            iv = c.outcome.intervals
            assert lines[0] == f"{c.name}\t{iv[0, 0]}\t{iv[0, 1]}\t{oracle.STATE_NAMES[iv[0, 2]]}"
            assert lines[-1] == f"{c.name}\t{iv[-1, 0]}\t{iv[-1, 1]}\t{oracle.STATE_NAMES[iv[-1, 2]]}"
        shard.close()
    finally:
        dist.destroy_process_group()


def test_long_read_chrY_full_size_properties_and_oracle_windows(tmp_path):
    """BASELINE.json configs[2] at full size (chrY-shaped 57.2 Mb, 50x, 1-60 kb reads with an indel every ~15 bases:
    354 M CIGAR operations): the size-independent properties of the result and per-position counters / states against
    the oracle on windows (the oracle gets the reads that can reach the window: those starting within the longest
    reference span before it)."""
    L = 57_227_415
    seed = synth.seed_for(3, 23)
    rec = synth.long_read_contig(L, 50, seed)
    ref = synth.make_reference(L, seed)
    opt = CallableOptions()
    acc, n_names = admit_reads(opt, 23, L, rec)
    with Engine(opt, 0) as eng:
        counter = CallableProfiler(str(tmp_path / "g.bed"))
        st = ContigProfiler("chrY", L)
        process_single_contig(eng, counter, st, opt, 23, rec, ref)
        counts = counter.get_contig_counts("chrY")
        counter.close()
        r1 = eng.contig_collect()
        eng.contig_run()
        r2 = eng.contig_collect()
        raw, qc, low, state = eng.debug_depths(L)
    assert r1.as_dict() == r2.as_dict() and np.array_equal(r1.intervals, r2.intervals)        # idempotent re-run
    _property_checks("chrY", L, rec, ref, opt, acc, n_names, st, counts, r1.intervals)
    assert np.bincount(state, minlength=6).tolist() == counts
    assert st.summed_coverage == int(raw.astype(np.int64).sum()) and st.n_covered_bases == int(np.count_nonzero(raw))
    assert st.quality_bases == int(qc.astype(np.int64).sum()) and np.all(qc <= raw) and np.all(low <= raw)
    # the longest reference span of a read bounds how far before a window a read that reaches it can start
    ops = rec.cigar & 15
    cs = np.concatenate([[0], np.cumsum(np.where(np.isin(ops, [0, 2, 3, 7, 8]), (rec.cigar >> 4).astype(np.int64), 0))])
    span = cs[rec.cigar_off[1:].astype(np.int64)] - cs[rec.cigar_off[:-1].astype(np.int64)]
    margin = int(span.max()) + 1
    rng = np.random.default_rng(7)
    for a in rng.integers(200_000, L - 300_000, size=3).tolist() + [0, L - 40_000]:
        b = min(a + 40_000, L)
        sub = _window_records(rec, a, b, margin=margin)
        prof = oracle.Profiler(str(tmp_path / "s.bed"))
        _, d = oracle.process_single_contig(prof, make_options({}), "chrY", 23, b, ref[:b], sub, dump=True)
        prof.close()
        for nm, arr_o, arr_g in (("raw", d[0], raw), ("qc", d[1], qc), ("low", d[2], low), ("state", d[3], state)):
            assert np.array_equal(arr_o[a:b], arr_g[a:b]), (nm, a)


def test_site_pileup_chrY_full_size_rows_and_oracle_windows():
    """BASELINE.json configs[4] at full size (chrY-shaped 57.2 Mb, 40x, 14.8 M reads with bases, 200 000 sites, base
    quality >= 20): a site's row does not depend on which other sites are asked for, every row is bounded by the
    coverage, and the rows of the sites inside windows equal the oracle's over the reads that can reach them."""
    L = 57_227_415
    seed = synth.seed_for(5, 23)
    ref = synth.make_reference(L, seed)
    rec = synth.short_read_contig(L, 40, seed, with_seq=True, ref=ref, max_live_assert=0)
    rng = np.random.default_rng(5)
    sites = rng.choice(np.arange(1, L + 1), size=200_000, replace=False).astype(np.uint32)
    with Engine(CallableOptions(), 0) as eng:
        hist = eng.site_pileup(20, L, L, rec, sites)
        ms, nbytes = eng.site_pileup_stats()
        half = eng.site_pileup(20, L, L, rec, sites[::2])
        srt = np.sort(sites)
        hist_sorted = eng.site_pileup(20, L, L, rec, srt)
    assert ms > 0 and nbytes > rec.n * 16
    assert np.array_equal(half, hist[::2])                                   # rows are independent of the list
    assert np.array_equal(hist_sorted, hist[np.argsort(sites, kind="stable")])   # ... and of its order
    assert int((hist.sum(1) > 0).sum()) > 150_000 and int(hist.sum(1).max()) < 400
    for a in [0, 1_234_567, 28_000_000, L - 300_000]:
        b = min(a + 300_000, L)
        i0 = int(np.searchsorted(rec.pos, a - 2_000)); i1 = int(np.searchsorted(rec.pos, b))
        sub = rec.slice(i0, i1)
        sel = np.flatnonzero((sites > a) & (sites <= b))                     # 1-based sites over positions [a, b)
        assert sel.shape[0] > 500
        exp = oracle.site_pileup(10, 20, L, ref, sub, sites[sel])
        assert np.array_equal(exp["hist"], hist[sel]), a


def test_bench_whole_genome_line_over_two_ranks(tmp_path):
    """`bench.py --gpus 2` started plainly, with no launcher around it: the script starts its own two ranks (one process
    per GPU under torch.distributed.run; here both on the box's one GPU, gloo for the collectives since nccl needs one
    device per rank) at 1/64 scale: one JSON line from rank 0, strong scaling over the fixed 25-contig input, both
    ranks' bases adding up to it, the gathered summaries checked against every rank's own first pass inside the run."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--wgs-scale", str(1.0 / 64), "--min-time", "0.05"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    if r.returncode != 0:                                     # the whole log, for the one who has to find out why
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "wgs_two_ranks_failed.log"), "w") as f:
            f.write(r.stdout + "\n==== stderr\n" + r.stderr)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    genome = wgs.genome(1.0 / 64)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["contigs"] == 25 and d["config"]["total_bases"] == sum(L for _, _, L in genome)
    assert sum(d["sharding"]["per_rank_bases"]) == d["config"]["total_bases"]
    assert sorted(sum(d["sharding"]["contigs_of_rank"], [])) == sorted(n for _, n, _ in genome)
    assert d["value"] > 0 and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["sharding"]["lpt_imbalance"] < 1.05
    assert len(d["sharding"]["build_s"]) == 2 and d["sharding"]["host_threads_per_rank"] >= 2

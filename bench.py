#!/usr/bin/env python3
"""bench.py -- callable-loci hot path on MI355X: reference bases classified per second.

A step = one pass of the whole device path (read prep -> window bounds -> pileup/classify ->
run-length intervals + summary) over ONE resident chr21-shaped contig (46 709 983 bp, 30x,
2x150 bp paired reads; BASELINE.json configs[1]) per GPU.  Inputs are resident in HBM when the
timed region starts.  N > 1: one process per GPU (torchrun), every rank owns its own contig of
the same shape (contigs shard with no data-path collective -> weak scaling); the per-contig
summaries are gathered with one RCCL all_gather.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(rec, ref, L_sample, opt, tid):
    """The oracle (CPU restatement of the reference algorithm, single thread like the reference)
    on a bounded sample: the first L_sample positions of the same contig."""
    import oracle
    from decodingustools_amd.records import ContigRecords
    n = int(np.searchsorted(rec.pos, L_sample))
    sub = ContigRecords(pos=rec.pos[:n], flag=rec.flag[:n], mapq=rec.mapq[:n],
                        cigar_off=rec.cigar_off[:n + 1], cigar=rec.cigar[:int(rec.cigar_off[n])],
                        qual_off=rec.qual_off[:n + 1], qual=rec.qual[:int(rec.qual_off[n])],
                        qname_off=rec.qname_off[:n + 1], qname=rec.qname[:int(rec.qname_off[n])])
    with tempfile.TemporaryDirectory() as d:
        bed = os.path.join(d, "o.bed")
        prof = oracle.Profiler(bed)
        t0 = time.perf_counter()
        st, _ = oracle.process_single_contig(prof, opt, "chr21", tid, L_sample, ref[:L_sample], sub)
        prof.close()
        dt = time.perf_counter() - t0
        counts = None
        obed = open(bed).read()
    return sub, st, obed, dt


def cpu_baseline_all_cores(rec, ref, opt, tid, start, piece, threads):
    """The same oracle on `threads` host threads at once, each on its own `piece`-long stretch of the
    contig (taken as a contig of its own).  The reference is single-threaded; this is the tile-parallel
    figure SURVEY 8d asks to be shown beside it."""
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    from decodingustools_amd.records import ContigRecords
    jobs = []
    for t in range(threads):
        a = start + t * piece
        if a + piece > ref.shape[0]:
            break
        i0, i1 = int(np.searchsorted(rec.pos, a)), int(np.searchsorted(rec.pos, a + piece))
        sub = rec.slice(i0, i1)
        sub.pos = (sub.pos - np.int32(a)).astype(np.int32)
        jobs.append((sub, ref[a:a + piece]))
    if not jobs:
        return None
    d = tempfile.mkdtemp()

    def run(k):
        prof = oracle.Profiler(os.path.join(d, f"p{k}.bed"))
        oracle.process_single_contig(prof, opt, "chr21", tid, piece, jobs[k][1], jobs[k][0])
        prof.close()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(len(jobs)) as ex:
        list(ex.map(run, range(len(jobs))))
    dt = time.perf_counter() - t0
    return {"value": len(jobs) * piece / dt, "unit": "bases/s", "cores": len(jobs),
            "sample": f"{len(jobs)} stretches of {piece} positions, one oracle thread each, {dt:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--length", type=int, default=46_709_983, help="contig length (default chr21)")
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--cpu-sample", type=int, default=16_000_000, help="positions of the CPU baseline sample (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the multi-rank logic on one GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: using WORLD_SIZE")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # torch first: it brings its own HIP runtime, and the engine's library must resolve against that one
    # (loading the engine before torch leaves the process with two runtimes and no usable device)
    import torch
    import torch.distributed as dist
    # the native pieces are (re)built before anything initialises the GPU: a process that has must not
    # start compilers (normally nothing is stale and this returns at once; every rank checks).  Compile
    # only -- the library itself is loaded further down, after the device is set.
    from decodingustools_amd import build as _native_build
    _native_build.build()
    import oracle
    oracle.build()
    n_dev = torch.cuda.device_count()
    dev_id = local_rank % max(n_dev, 1)          # one GPU per rank on a real node
    torch.cuda.set_device(dev_id)
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_id))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from decodingustools_amd import (CallableOptions, CallableProfiler, ContigProfiler, Engine,
                                     process_single_contig, synth)

    L = args.length
    tid = 20
    seed = synth.seed_for(2, tid) + 1000 * rank
    t0 = time.perf_counter()
    rec = synth.short_read_contig(L, args.depth, seed)
    ref = synth.make_reference(L, seed)
    log(f"[bench r{rank}] synthetic contig: {L} bp, {rec.n} reads, {rec.qual.shape[0]} aligned bases "
        f"({time.perf_counter() - t0:.1f}s)")

    opt = CallableOptions()          # the CLI defaults (cli.rs:34-60)
    eng = Engine(opt, dev_id)
    tmpd = tempfile.mkdtemp()
    # ---- first pass through the module API: admission + H2D + kernels + D2H + BED text ----
    counter = CallableProfiler(os.path.join(tmpd, f"g{rank}.bed"))
    st = ContigProfiler("chr21", L)
    t0 = time.perf_counter()
    process_single_contig(eng, counter, st, opt, tid, rec, ref)
    counter.close()
    e2e = time.perf_counter() - t0
    first = eng.contig_collect()
    log(f"[bench r{rank}] end-to-end first pass (host admission + PCIe + kernels + BED): {e2e:.2f}s "
        f"= {L / e2e / 1e9:.3f} Gbase/s; intervals {first.summary.n_intervals}")

    # ---- timed region: resident contig, K steps ----
    for _ in range(args.warmup):
        eng.contig_run()
    eng.sync()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.contig_run()             # four kernel launches on the engine's stream
    eng.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    # per-kernel durations: the same steps once more with HIP events between the kernel groups on the
    # engine's stream (direct launches; outside the timed region)
    eng.set_profiling(True)
    eng.reset_kernel_ms()
    for _ in range(max(5, min(args.steps, 20))):
        eng.contig_run()
    eng.sync()
    kms, nruns = eng.kernel_ms()
    eng.set_profiling(False)
    again = eng.contig_collect()
    assert again.as_dict() == first.as_dict() and np.array_equal(again.intervals, first.intervals), \
        "resident re-run changed the result"

    # max over ranks + RCCL gather of the per-contig summaries (the path's only exchange)
    tsr = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
    summ = torch.tensor([int(x) for x in first.state_counts] +
                        [int(first.summary.n_covered_bases), int(first.summary.summed_coverage),
                         int(first.summary.summed_baseq), int(first.summary.summed_mapq),
                         int(first.summary.quality_bases), int(first.summary.extent)],
                        dtype=torch.int64, device=coll_dev)
    if world > 1:
        dist.all_reduce(tsr, op=dist.ReduceOp.MAX)
        allsum = [torch.zeros_like(summ) for _ in range(world)]
        dist.all_gather(allsum, summ)
    else:
        allsum = [summ]
    dt_max = float(tsr.item())
    total_bases = sum(int(s[11].item()) for s in allsum)

    if rank == 0:
        ms_step = dt_max * 1e3 / args.steps
        value = total_bases / (dt_max / args.steps)
        inb, outb = eng.contig_bytes()
        pile_ms = kms["pileup"] / max(nruns, 1)
        alg = inb + outb
        achieved = alg / (pile_ms * 1e-3) / 1e9 if pile_ms > 0 else 0.0
        peak = 8000.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("k_pileup_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "reference bases classified/sec (whole node) + BED bit-exact vs CPU ref",
            "value": value, "unit": "bases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "coverage -L chr21 synthetic 30x 150bp paired, device-resident, one contig per GPU",
                       "contig_len": L, "depth": args.depth, "reads_per_contig": rec.n,
                       "aligned_bases_per_contig": int(rec.qual.shape[0]), "parallelism": f"contig-per-gpu x{world}",
                       "options": "cli defaults (4,500,10,20,10,1,0.1)"},
            "roofline": {"bound": "hbm", "kernel": "k_pileup", "achieved": achieved, "peak": peak, "unit": "GB/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg, "kernel_ms": pile_ms,
                         "all_kernel_ms": {k: v / max(nruns, 1) for k, v in kms.items()}},
            "end_to_end_first_pass_s": e2e,
        }
        if world == 1 and args.cpu_sample > 0:
            Ls = min(args.cpu_sample, L)
            sub, ost, obed, cdt = cpu_baseline(rec, ref, Ls, opt, tid)
            # the same sample through the GPU engine: BED must be bit-identical
            counter = CallableProfiler(os.path.join(tmpd, "gs.bed"))
            st2 = ContigProfiler("chr21", Ls)
            process_single_contig(eng, counter, st2, opt, tid, sub, ref[:Ls])
            counter.close()
            gbed = open(os.path.join(tmpd, "gs.bed")).read()
            exact = (gbed == obed) and all(getattr(st2, k) == ost[k] for k in
                                           ("n_covered_bases", "summed_coverage", "summed_baseq", "summed_mapq",
                                            "quality_bases", "n_reads"))
            out["cpu_baseline"] = {"value": Ls / cdt, "unit": "bases/s", "cores": 1, "kind": "port",
                                   "sample": f"first {Ls} positions of the same contig ({sub.n} reads), "
                                             f"oracle/callable_oracle.c single thread, {cdt:.1f}s"}
            try:
                allc = cpu_baseline_all_cores(rec, ref, opt, tid, Ls, 2_000_000, min(16, os.cpu_count() or 1))
                if allc:
                    out["cpu_baseline"]["all_cores"] = allc
            except Exception as e:                      # the single-thread figure above is the contract's
                log(f"[bench] all-cores CPU baseline skipped: {e}")
            out["bed_bit_exact"] = bool(exact)
            if not exact:
                log("[bench] WARNING: GPU BED/summary differs from the oracle on the sample")
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

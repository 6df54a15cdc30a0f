#!/usr/bin/env python3
"""bench.py -- callable-loci hot path on MI355X: reference bases classified per second.

A step = one pass of the whole device path (pileup/classify per window -> run-length intervals + summary:
three launches) over the rank's resident input.  Inputs are resident in HBM when the timed region starts.

  --gpus 1 (default)   BASELINE.json configs[1], the configuration the metric is quoted on: ONE
                       chr21-shaped contig (46 709 983 bp, 30x, 2x150 bp paired reads).  The whole contig
                       is also run through the CPU oracle (single thread, like the reference) and the BED /
                       summary compared bit for bit; configs[2] (long reads) and configs[4] (site pileup) are
                       measured as secondary lines under "configs".
  --gpus N > 1         BASELINE.json configs[3]: the FIXED whole-genome input -- the 25 hg38 primary contigs
                       at 30x -- dealt to the ranks by longest-processing-time-first (strong scaling: what is
                       sharded is the reference's serial contig loop, src/api/coverage.rs:229-234).  Every rank
                       generates, uploads and keeps resident only its own contigs; a step runs them all and
                       ends with the path's only exchange, one all_gather (RCCL over xGMI) of the per-contig
                       summary records taken straight from HBM (cl_device_summary).
  --workload chr21|wgs overrides the choice (e.g. the whole genome on one GPU; chr21 per rank = weak scaling).

K steps are timed exactly as the contract says (barrier + synchronize on both sides, max over ranks).  One
such block lasts ~10 ms at the default K, so the block is repeated until --min-time seconds of timed
blocks have accumulated; ms_per_step is the mean over the blocks (every block's figures are kept).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_HBM_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "reference bases classified/sec (whole node) + BED bit-exact vs CPU ref"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------
# CPU baseline legs (the oracle is the checker and the reported baseline, never the product path)
# ------------------------------------------------------------------------------------------------
def cpu_baseline(rec, ref, L_sample, opt, tid):
    """The oracle (CPU restatement of the reference algorithm, single thread like the reference) on the
    first L_sample positions of the contig (the whole contig by default)."""
    import oracle
    from decodingustools_amd.records import ContigRecords
    if L_sample >= ref.shape[0]:
        sub = rec
    else:
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
        obed = open(bed).read()
    return sub, st, obed, dt


def cpu_baseline_all_cores(rec, ref, opt, tid, piece, threads):
    """The same oracle on `threads` host threads at once, each on its own `piece`-long stretch of the
    contig (taken as a contig of its own).  The reference is single-threaded; this is the tile-parallel
    figure SURVEY 8d asks to be shown beside it."""
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    jobs = []
    for t in range(threads):
        a = 1_000_000 + t * piece
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


# ------------------------------------------------------------------------------------------------
# timing
# ------------------------------------------------------------------------------------------------
def timed_blocks(step, sync, args, world, dist, torch, coll_dev):
    """W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both
    sides; blocks repeat until --min-time seconds of them have accumulated (the count is agreed across
    ranks).  Returns per-block durations (max over ranks), seconds."""
    for _ in range(args.warmup):
        step()
    sync()

    def block():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return time.perf_counter() - t0
    first = block()
    t = torch.tensor([first], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n_more = int(min(args.max_blocks - 1, max(0, math.ceil(args.min_time / max(float(t.item()), 1e-6)) - 1)))
    dts = [first] + [block() for _ in range(n_more)]
    t = torch.tensor(dts, dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t.cpu().tolist()]


def kernel_profile(engines, steps, one_at_a_time=False):
    """Per-kernel durations: the same steps once more with HIP events between the kernel groups on the
    engines' stream (direct launches; outside the timed region).  Returns ({group: ms per step summed over the
    engines}, runs per engine).  one_at_a_time: every engine's run is waited for before the next one's is launched --
    for engines on different streams, whose kernels would otherwise overlap and each look longer than it is."""
    # (a few dozen un-profiled steps first: a secondary measurement starts on a device whose clocks have dropped while the
    # host generated and admitted the contig; the headline's timed region is half a second long, these are not)
    for _ in range(max(0, int(40 / max(1, len(engines))))):
        for e in engines:
            e.contig_run()
    for e in engines:
        e.sync()
        e.set_profiling(True)
        e.reset_kernel_ms()
    n = max(5, min(steps, 20))
    for _ in range(n):
        for e in engines:
            e.contig_run()
            if one_at_a_time:
                e.sync()
    tot = {}
    for e in engines:
        e.sync()
        kms, nruns = e.kernel_ms()
        for k, v in kms.items():
            tot[k] = tot.get(k, 0.0) + v / max(nruns, 1)
        e.set_profiling(False)
    return tot, n


def traffic_from_profiles():
    """The committed profile's figure for the HBM traffic of one k_pileup launch, under its own key with the build it
    was measured on -- beside `traffic`, which measure_traffic() fills from this run's own PMC passes."""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        d = json.load(open(tf))
        return {"k_pileup_hbm_bytes_per_launch": d.get("k_pileup_hbm_bytes_per_launch"),
                "file": "profiles/traffic.json", "build": d.get("build"), "workload": d.get("workload", "chr21 30x")}
    except Exception:
        return None


def save_records(dirpath, tag, rec, ref, extra=None):
    """The inputs of a workload as .npy files for the PMC child passes (they would otherwise generate them again:
    a minute per pass).  Returns False when the scratch directory cannot take them."""
    try:
        d = os.path.join(dirpath, tag)
        os.makedirs(d, exist_ok=True)
        for k in ("pos", "flag", "mapq", "cigar_off", "cigar", "qual_off", "qual", "qname_off", "qname", "seq_off", "seq4"):
            v = getattr(rec, k, None)
            if v is not None:
                np.save(os.path.join(d, k + ".npy"), v)
        np.save(os.path.join(d, "ref.npy"), ref)
        for k, v in (extra or {}).items():
            np.save(os.path.join(d, k + ".npy"), v)
        return True
    except Exception as e:
        log(f"[bench] could not cache the {tag} inputs for the PMC passes: {e}")
        return False


def load_records(dirpath, tag):
    from decodingustools_amd.records import ContigRecords
    d = os.path.join(dirpath, tag)
    f = {k[:-4]: np.load(os.path.join(d, k)) for k in os.listdir(d) if k.endswith(".npy")}
    rec = ContigRecords(pos=f["pos"], flag=f["flag"], mapq=f["mapq"], cigar_off=f["cigar_off"], cigar=f["cigar"],
                        qual_off=f["qual_off"], qual=f["qual"], qname_off=f["qname_off"], qname=f["qname"],
                        seq_off=f.get("seq_off"), seq4=f.get("seq4"))
    return rec, f["ref"], f


def pmc_child(dirpath, dev_id):
    """What runs under `rocprofv3 --pmc <counter>`: every workload cached in `dirpath` once through the engine and a
    few resident steps more.  Prints nothing: the parent reads the profiler's CSV by kernel name."""
    from decodingustools_amd import (CallableOptions, CallableProfiler, ContigProfiler, Engine, process_single_contig)
    opt = CallableOptions()
    tmpd = tempfile.mkdtemp()
    for tag, name, tid in (("chr21", "chr21", 20), ("long", "chrY", 23)):
        if not os.path.isdir(os.path.join(dirpath, tag)):
            continue
        rec, ref, _ = load_records(dirpath, tag)
        with Engine(opt, dev_id) as eng:
            counter = CallableProfiler(os.path.join(tmpd, tag + ".bed"))
            process_single_contig(eng, counter, ContigProfiler(name, ref.shape[0]), opt, tid, rec, ref)
            counter.close()
            for _ in range(4):
                eng.contig_run()
            eng.sync()
        del rec, ref
    if os.path.isdir(os.path.join(dirpath, "site")):
        rec, ref, f = load_records(dirpath, "site")
        with Engine(opt, dev_id) as eng:
            for _ in range(2):
                eng.site_pileup(20, ref.shape[0], ref.shape[0], rec, f["sites"])


def measure_traffic(cache_dir):
    """roofline.traffic of THIS run: HBM bytes per launch from the PMC counters, collected the way MI355X_MICROARCH.md's
    HBM section prescribes -- FETCH_SIZE and WRITE_SIZE in separate `rocprofv3 --pmc` passes, no tracing domain beside
    them -- over two child runs of this script (--pmc-child: the cached inputs of the headline workload and of the
    secondary ones, a first pass and four resident steps each), after the timed region.  bytes = (2 x FETCH_SIZE +
    WRITE_SIZE) x 1024: the counters are in KB and gfx950's FETCH_SIZE counts 64 B per 128-byte request of a wide
    streaming read (calibrated on k_pileup's access pattern with tools/ubench/fetch_calib.hip; the site kernel's
    scattered byte loads are not such a stream, so its figure is reported raw as well).  Returns ({kernel key: {...}},
    None) or (None, reason): a profiler that is missing or refuses never costs the bench line."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "rocprofv3 not found"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        cmd = [rp, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--pmc-child", cache_dir]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
            vals = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    kn = row.get("Kernel_Name", "")
                    if row.get("Counter_Name") != ctr:
                        continue
                    if "k_site_pileup" in kn:
                        key = "k_site_pileup"
                    elif "k_pileup_rows" in kn:
                        # one kernel for every read shape: the workloads are told apart by their grid (windows x 256)
                        key = "k_pileup_rows@" + str(row.get("Grid_Size", "?"))
                    elif "k_pileup" in kn:
                        m = re.search(r"k_pileup<[^>]*?(\d+)\s*>", kn)
                        key = "k_pileup_form" + (m.group(1) if m else "?")
                    else:
                        continue
                    vals.setdefault(key, []).append(float(row["Counter_Value"]))
            if not vals:
                return None, f"no {ctr} rows (rocprofv3 rc {r.returncode}: {(r.stderr or '')[-300:]})"
            got[ctr] = {k: (sum(v) / len(v), len(v)) for k, v in vals.items()}
        except Exception as e:
            return None, f"{ctr} pass failed: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for key in got["FETCH_SIZE"]:
        if key not in got["WRITE_SIZE"]:
            continue
        fk, n = got["FETCH_SIZE"][key]
        wk = got["WRITE_SIZE"][key][0]
        out[key] = {"bytes": int((2.0 * fk + wk) * 1024), "raw_bytes": int((fk + wk) * 1024), "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "launches_averaged": n,
                    "how": "two child runs of bench.py under rocprofv3 --pmc (FETCH_SIZE, then WRITE_SIZE), "
                           "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024"}
    return out, None


def survey_bytes(aligned_bases, reads, positions, q, cigar_ops=0):
    """SURVEY 8d's algorithmic bytes of the fused scatter + classify step, B = D (q + 9 / l) + 1 per reference position
    without the 6c of the counter arrays (the fused design never writes them): quality representation q bytes per
    aligned base (1/8: a host-thresholded bit; 1: a raw Phred byte), 9 bytes per read, 4 bytes per CIGAR operation beyond
    the first of a read (8d: "long reads: add 4 ops / l"), 1 reference byte per position."""
    return int(aligned_bases * q + reads * 9 + 4 * max(0, cigar_ops - reads) + positions)


def roofline(alg_bytes, pile_ms, step_ms, kms, n_gpus=1, layout_bytes=None, kernel="k_pileup_rows", rle_bytes=None):
    """alg_bytes: SURVEY 8d's algorithmic bytes (survey_bytes) of the launches of one step, pile_ms: the pileup kernel's
    time over the same launches (over all GPUs when there are several: the quotient is then the launch-weighted mean per
    GPU); layout_bytes: what the resident form must read at least once, counted strictly (cl_contig_bytes: the rows'
    padding and the 16-byte records included); step_*: the algorithmic bytes over the wall time of a whole step."""
    ach = alg_bytes / (pile_ms * 1e-3) / 1e9 if pile_ms > 0 else 0.0
    step_ach = alg_bytes / n_gpus / (step_ms * 1e-3) / 1e9 if step_ms > 0 else 0.0
    r = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
         "frac": ach / PEAK_HBM_GBS, "traffic": None, "traffic_from_profiles": traffic_from_profiles(),
         "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_survey": alg_bytes,
         "kernel_ms": pile_ms, "step_achieved": step_ach, "step_frac": step_ach / PEAK_HBM_GBS, "all_kernel_ms": kms}
    if layout_bytes is not None:
        lach = layout_bytes / (pile_ms * 1e-3) / 1e9 if pile_ms > 0 else 0.0
        r["layout_bytes_per_launch"] = layout_bytes
        r["layout_achieved"] = lach
        r["layout_frac"] = lach / PEAK_HBM_GBS
    # every kernel a single pass over a contig launches, upload included: the pass-bit form launches nothing at upload
    # (the rows are built by the host's walk and arrive through the pinned ring), so the pass is the step's kernels
    ks = [{"name": kernel, "ms": pile_ms, "bytes": layout_bytes if layout_bytes is not None else alg_bytes,
           "frac": (layout_bytes if layout_bytes is not None else alg_bytes) / (pile_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if pile_ms > 0 else 0.0}]
    rle_ms = kms.get("rle", 0.0)
    if rle_ms:
        e = {"name": "k_fin_windows + k_rle_write", "ms": rle_ms}
        if rle_bytes:
            e["bytes"] = rle_bytes
            e["frac"] = rle_bytes / (rle_ms * 1e-3) / 1e9 / PEAK_HBM_GBS
        ks.append(e)
    r["kernels"] = ks
    r["upload_kernels"] = []
    r["upload_device_ms"] = 0.0
    r["single_pass_device_ms"] = r["upload_device_ms"] + sum(v for v in kms.values())
    return r


def engine_figures(eng, rec, positions):
    """(survey bytes, layout bytes, rle bytes, layout dict) of the contig resident in `eng` (built from `rec`)."""
    lay = eng.contig_layout()
    inb, outb = eng.contig_bytes()
    q = 0.125 if lay["form"] == 3 else 1.0
    ops = int(rec.cigar.shape[0]) if lay["form"] == 2 else 0
    alg = survey_bytes(int(rec.qual.shape[0]), rec.n, positions, q, ops)
    # the two small kernels behind the pileup: a 104-byte window partial, two state bytes and the used prefix of the run
    # list (2 bytes per interval) read per window, 12 bytes per interval written
    n_iv = outb // 12
    rle = int(lay["n_windows"] * (104 + 2 + 4) + n_iv * (2 + 12))
    return alg, inb + outb, rle, lay


# ------------------------------------------------------------------------------------------------
# secondary measurements of the N = 1 line: configs[2] (long reads), configs[4] (site pileup), the same
# chr21 input from BAM + FASTA files, and configs[3] on this one GPU (the 1-GPU point of the scaling curve)
# ------------------------------------------------------------------------------------------------
def byte_form_figures(dev_id, opt, rec, ref, name, tid, L, bed_of_default_form, tmpd):
    """The same contig through the byte forms of the pileup kernel (DUT_QUAL_FORM=bytes, read at cl_create: the quality
    bytes go to HBM and are tested there): first pass, kernel time by HIP events, bytes, and the BED compared with the
    default form's."""
    from decodingustools_amd import CallableProfiler, ContigProfiler, Engine, process_single_contig
    prev = os.environ.get("DUT_QUAL_FORM")
    os.environ["DUT_QUAL_FORM"] = "bytes"
    try:
        with Engine(opt, dev_id) as eng:
            bed = os.path.join(tmpd, name + "_bytes.bed")
            counter = CallableProfiler(bed)
            t0 = time.perf_counter()
            process_single_contig(eng, counter, ContigProfiler(name, L), opt, tid, rec, ref)
            counter.close()
            first = time.perf_counter() - t0
            for _ in range(3):
                eng.contig_run()
            eng.sync()
            kms, _ = kernel_profile([eng], 10)
            alg, layb, rleb, lay = engine_figures(eng, rec, L)
    finally:
        if prev is None:
            os.environ.pop("DUT_QUAL_FORM", None)
        else:
            os.environ["DUT_QUAL_FORM"] = prev
    step = sum(kms.values())
    return {"how": "DUT_QUAL_FORM=bytes at cl_create", "form": lay["form"], "kernel": "k_pileup", "kernel_ms": kms["pileup"], "ms_per_step": step,
            "value": L / (step * 1e-3), "unit": "bases/s", "end_to_end_first_pass_s": first,
            "algorithmic_bytes_survey": alg, "frac": alg / (kms["pileup"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "layout_bytes_per_launch": layb, "layout_frac": layb / (kms["pileup"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "device_bytes": lay["device_bytes"], "upload_h2d_bytes": lay["upload_h2d_bytes"],
            "bed_equals_default_form": open(bed).read() == open(bed_of_default_form).read()}


def long_read_config(dev_id, opt, args, cache_dir, tmpd):
    """configs[2]: chrY-shaped, 50x, long reads (median 10 kb, an indel every ~15 bases); BED and summary compared with
    the oracle over the WHOLE contig."""
    import oracle
    from decodingustools_amd import CallableProfiler, ContigProfiler, Engine, process_single_contig, synth
    L = args.long_length
    t0 = time.perf_counter()
    seed = synth.seed_for(3, 23)
    rec = synth.long_read_contig(L, 50, seed)
    ref = synth.make_reference(L, seed)
    gen = time.perf_counter() - t0
    gbed = os.path.join(tmpd, "long.bed")
    with Engine(opt, dev_id) as eng:
        counter = CallableProfiler(gbed)
        st = ContigProfiler("chrY", L)
        t0 = time.perf_counter()
        process_single_contig(eng, counter, st, opt, 23, rec, ref)
        counter.close()
        first = time.perf_counter() - t0
        for _ in range(2):
            eng.contig_run()
        eng.sync()
        kms, _ = kernel_profile([eng], 5)
        alg, layb, rleb, lay = engine_figures(eng, rec, L)
        # the same contig once more on the warm engine
        counter2 = CallableProfiler(os.path.join(tmpd, "long_again.bed"))
        t0 = time.perf_counter()
        process_single_contig(eng, counter2, ContigProfiler("chrY", L), opt, 23, rec, ref)
        counter2.close()
        nxt = time.perf_counter() - t0
    step = sum(kms.values())
    out = {
        "workload": "coverage -L chrY synthetic 50x long-read (10 kb ONT-style CIGAR with indels), device-resident",
        "contig_len": L, "reads": rec.n, "cigar_ops": int(rec.cigar.shape[0]), "aligned_bases": int(rec.qual.shape[0]),
        "value": L / (step * 1e-3), "unit": "bases/s", "ms_per_step": step, "generated_in_s": round(gen, 1),
        "end_to_end_first_pass_s": first, "end_to_end_next_pass_s": nxt, "layout": lay,
        "roofline": roofline(alg, kms["pileup"], step, kms, layout_bytes=layb, rle_bytes=rleb)}
    out["byte_form"] = byte_form_figures(dev_id, opt, rec, ref, "chrY", 23, L, gbed, tmpd)
    out["roofline"].pop("traffic_from_profiles", None)
    if args.cpu_sample != 0:
        obed = os.path.join(tmpd, "long_o.bed")
        prof = oracle.Profiler(obed)
        t0 = time.perf_counter()
        ost, _ = oracle.process_single_contig(prof, opt, "chrY", 23, L, ref, rec)
        prof.close()
        cdt = time.perf_counter() - t0
        exact = open(gbed).read() == open(obed).read() and all(
            getattr(st, k) == ost[k] for k in ("n_covered_bases", "summed_coverage", "summed_baseq", "summed_mapq", "quality_bases", "n_reads"))
        out["bed_bit_exact"] = bool(exact)
        out["bed_bit_exact_positions"] = L
        out["cpu_baseline"] = {"value": L / cdt, "unit": "bases/s", "cores": 1, "kind": "port",
                               "sample": f"the whole contig, {L} positions, oracle/callable_oracle.c single thread, {cdt:.1f}s"}
        if not exact:
            log("[bench] WARNING: long-read GPU BED/summary differs from the oracle")
    out["_cached"] = save_records(cache_dir, "long", rec, ref) if cache_dir else False
    return out


def site_config(dev_id, opt, args, cache_dir):
    """configs[4]: chrY-shaped, 40x, config-2 read model with bases, 200 000 sites, min_quality 20; the full histogram
    compared with the oracle's."""
    import oracle
    from decodingustools_amd import Engine, synth
    L = args.site_length
    t0 = time.perf_counter()
    seed = synth.seed_for(5, 23)
    ref = synth.make_reference(L, seed)
    rec = synth.short_read_contig(L, 40, seed, with_seq=True, ref=ref, max_live_assert=0)
    rng = np.random.default_rng(5)
    sites = rng.choice(np.arange(1, L + 1), size=min(200_000, L), replace=False).astype(np.uint32)
    gen = time.perf_counter() - t0
    with Engine(opt, dev_id) as eng:
        ms_all, calls = [], []
        for _ in range(3):
            t0 = time.perf_counter()
            hist = eng.site_pileup(20, L, L, rec, sites)
            calls.append(time.perf_counter() - t0)
            kms_, nb = eng.site_pileup_stats()
            ms_all.append(kms_)
        # the two-step form: the whole tile resident (cl_site_upload), another site list (here the same one) costs a run only
        # (the one-call form above sends only the reads that overlap a site of its list: its tile serves that call alone)
        eng.site_upload(L, L, rec)
        t0 = time.perf_counter()
        hist2 = eng.site_run(20, sites)
        rerun = time.perf_counter() - t0
    kms_ = min(ms_all)
    ach = nb / (kms_ * 1e-3) / 1e9 if kms_ > 0 else 0.0
    out = {
        "workload": "find-y-branch pileup path: chrY 40x synthetic, 200 000-site list, device kernel only",
        "contig_len": L, "reads": rec.n, "sites": int(sites.shape[0]), "sites_hit": int((hist.sum(1) > 0).sum()),
        "kernel_ms": kms_, "call_s_incl_h2d": min(calls), "call_s_first": calls[0], "run_s_on_resident_tile": rerun,
        "resident_rerun_equal": bool(np.array_equal(hist, hist2)), "generated_in_s": round(gen, 1),
        "roofline": {"bound": "hbm", "kernel": "k_site_pileup", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": ach / PEAK_HBM_GBS, "traffic": None, "algorithmic_bytes_per_launch": nb,
                     "note": "nominal: SURVEY 8d's byte count includes the 4-bit bases of every read; the kernel touches the bases at sites only"}}
    if args.cpu_sample != 0:
        t0 = time.perf_counter()
        exp = oracle.site_pileup(10, 20, L, ref, rec, sites)
        cdt = time.perf_counter() - t0
        out["hist_exact"] = bool(np.array_equal(exp["hist"], hist))
        out["hist_exact_rows"] = int(sites.shape[0])
        out["cpu_baseline"] = {"value": rec.n / cdt, "unit": "reads/s", "cores": 1, "kind": "port",
                               "sample": f"all {rec.n} reads x {sites.shape[0]} sites, oracle/callable_oracle.c single thread, {cdt:.1f}s"}
        if not out["hist_exact"]:
            log("[bench] WARNING: site histogram differs from the oracle")
    out["_cached"] = save_records(cache_dir, "site", rec, ref, {"sites": sites}) if cache_dir else False
    return out


def files_config(dev_id, opt, rec, ref, L, first_bed_path, tmpd):
    """SURVEY 8d timing (ii): the headline input as a BGZF BAM + .bai + FASTA + .fai on disk (written outside the timed
    call by the harness's own writer) -> dut_coverage_files -> BED + summary.json; the BED must equal the first pass's."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_bench_lib as EL
    from decodingustools_amd import build as _b
    bam, fa = os.path.join(tmpd, "chr21.bam"), os.path.join(tmpd, "chr21.fa")
    t0 = time.perf_counter()
    EL.write_bam_native(tmpd, bam, "chr21", L, rec, threads=min(16, int(os.environ.get("DUT_THREADS", "16"))))
    EL.write_single_ref_bai(bam + ".bai", rec.n)
    EL.write_fasta(fa, "chr21", ref)
    wrote = time.perf_counter() - t0
    out = {"workload": "coverage chr21.bam -r chr21.fa -o out.bed (BGZF inflate + record parse + admission + H2D + kernels + BED + summary.json)",
           "bam_bytes": os.path.getsize(bam), "records": rec.n, "files_written_in_s": round(wrote, 1),
           "host_threads": int(os.environ.get("DUT_THREADS", "0")) or None, "runs": []}
    # the product's own command line tool, a fresh process per run as a user would start it (DUT_TIMING: its host stages)
    for rep in range(3):
        bed = os.path.join(tmpd, f"files{rep}.bed")
        env = dict(os.environ, DUT_TIMING="1")
        if rep == 2:
            env["DUT_CLI_FOREGROUND"] = "1"             # one process: the caller also waits for the exit's teardown
        run = run_tool([_b.CLI, "coverage", bam, "-r", fa, "-o", bed], tmpd, env)
        if "error" in run:
            out["error"] = run["error"]
            return out
        (out["runs"] if rep < 2 else out.setdefault("runs_foreground", [])).append(run)
    best = min(x["wall_s"] for x in out["runs"])
    out["seconds"] = best
    out["value"] = L / best
    out["unit"] = "bases/s"
    out["seconds_foreground"] = out["runs_foreground"][0]["wall_s"]
    out["note"] = ("the tool returns when every output file is written and closed; the process that did the work gives its "
                   "memory and the device context back after that, in the background (seconds_foreground: DUT_CLI_FOREGROUND=1, the caller waits for that too)")
    out["bed_equals_first_pass"] = open(os.path.join(tmpd, "files1.bed")).read() == open(first_bed_path).read()
    return out


def run_tool(cmd, cwd, env):
    """One run of the command line tool: wall time as its caller sees it, the host stages it reports (DUT_TIMING)."""
    import re
    import subprocess
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        return {"error": (r.stderr or "")[-400:]}
    stages = {}
    for ln in (r.stderr or "").splitlines():
        m = ln.split("]", 1)
        if ln.startswith("[dut-timing]") and len(m) == 2 and not m[1].startswith("   ") and "wall clock" not in ln:
            parts = m[1].rsplit(None, 2)
            if len(parts) == 3 and parts[2] == "ms":
                stages[parts[0].strip()] = stages.get(parts[0].strip(), 0.0) + float(parts[1])
    st = dict(re.findall(r"wall clock at (\w+)[^:]*: ([0-9.]+)", r.stderr or ""))
    run = {"wall_s": dt, "stages_ms": stages}
    if "main" in st and "exit" in st:
        run["main_s"] = float(st["exit"]) - float(st["main"])
    return run


def files_multi_config(dev_id, opt, args, tmpd):
    """A whole genome from files: the 25 hg38 primary contigs at 1/16 of their lengths (193 Mb, 30x, one BAM + .bai +
    FASTA + .fai) through the `dut-coverage` tool -- the read-ahead of the next contig, the per-contig fixed costs and
    the tool's start and end are all in the number; then the same over two engine contexts on this box's one GPU
    (--devices 0,0: the several-device path below Python, dut_coverage_files_multi), whose BED must be the same bytes."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_bench_lib as EL
    from decodingustools_amd import build as _b, synth, wgs
    genome = wgs.genome(1.0 / 16.0)
    t0 = time.perf_counter()

    def gen(item):
        tid, name, L = item
        seed = synth.seed_for(4, tid)
        return name, L, synth.short_read_contig(L, args.depth, seed), synth.make_reference(L, seed)
    with ThreadPoolExecutor(max(1, args.gen_threads)) as pool:
        made = list(pool.map(gen, genome))
    gen_s = time.perf_counter() - t0
    bam, fa = os.path.join(tmpd, "wgs16.bam"), os.path.join(tmpd, "wgs16.fa")
    t0 = time.perf_counter()
    EL.write_multi_bam(tmpd, bam, [(n, L, r) for n, L, r, _ in made], threads=min(16, int(os.environ.get("DUT_THREADS", "16"))))
    EL.write_multi_fasta(fa, [(n, ref) for n, _, _, ref in made])
    wrote = time.perf_counter() - t0
    total = sum(L for _, L, _, _ in made)
    out = {"workload": "coverage wgs16.bam -r wgs16.fa (25 hg38 primary contigs at 1/16 length, 30x): BGZF inflate + parse + admission + upload + kernels + BED + summary.json, per contig, the next contig read ahead",
           "contigs": len(made), "total_bases": total, "records": int(sum(r.n for _, _, r, _ in made)), "bam_bytes": os.path.getsize(bam),
           "generated_in_s": round(gen_s, 1), "files_written_in_s": round(wrote, 1), "runs": {}}
    del made
    beds = {}
    for tag, extra in (("one_context", []), ("two_contexts_one_gpu", ["--devices", "0,0"])):
        runs = []
        for rep in range(2):
            bed = os.path.join(tmpd, f"wgs16_{tag}.bed")
            run = run_tool([_b.CLI, "coverage", bam, "-r", fa, "-o", bed] + extra, tmpd, dict(os.environ, DUT_TIMING="1"))
            if "error" in run:
                out["error"] = run["error"]
                return out
            runs.append(run)
        beds[tag] = open(bed, "rb").read()
        best = min(x["wall_s"] for x in runs)
        out["runs"][tag] = {"seconds": best, "value": total / best, "unit": "bases/s", "all": runs}
    out["seconds"] = out["runs"]["one_context"]["seconds"]
    out["value"] = total / out["seconds"]
    out["unit"] = "bases/s"
    out["bed_equal_over_two_contexts"] = beds["one_context"] == beds["two_contexts_one_gpu"]
    out["bed_lines"] = beds["one_context"].count(b"\n")
    return out


WGS_CACHE = [os.path.join(tempfile.gettempdir(), "dut_bench_wgs_1gpu.json"), os.path.join(ROOT, "gpurun_out", "wgs_1gpu_cache.json")]


def cached_wgs_1gpu(total_bases, args, write=None):
    """The 1-GPU figure of the same fixed whole-genome input, left behind by an N = 1 run of this script on this box (or
    in this tree): what an N > 1 line's `value` is to be compared with.  write: the figure to leave."""
    key = {"total_bases": int(total_bases), "depth": args.depth, "wgs_scale": args.wgs_scale}
    if write is not None:
        for p in WGS_CACHE:
            try:
                os.makedirs(os.path.dirname(p), exist_ok=True)
                json.dump(dict(key, value=write["value"], ms_per_step=write["ms_per_step"]), open(p, "w"))
            except Exception:
                pass
        return None
    for p in WGS_CACHE:
        try:
            d = json.load(open(p))
            if all(d.get(k) == v for k, v in key.items()):
                return {"value": d["value"], "ms_per_step": d["ms_per_step"], "from": p}
        except Exception:
            continue
    return None


def wgs_point(args, dev_id, torch, dist, coll_dev):
    """configs[3] on this ONE GPU: the fixed whole-genome input of the N > 1 runs, all 25 contigs resident -- the 1-GPU
    point of the strong-scaling curve."""
    full = run_wgs(args, 0, 1, dev_id, torch, dist, coll_dev)
    keep = {k: full[k] for k in ("value", "unit", "ms_per_step", "scaling", "timing", "roofline", "callable_fraction",
                                 "hbm_resident_bytes_per_rank", "upload_h2d_bytes_per_rank")}
    cached_wgs_1gpu(full["config"]["total_bases"], args, write=full)
    keep["workload"] = full["config"]["workload"]
    keep["total_bases"] = full["config"]["total_bases"]
    keep["per_rank_first_pass_s"] = full["sharding"]["per_rank_first_pass_s"]
    keep["build_s"] = full["sharding"]["build_s"]
    keep["roofline"].pop("traffic_from_profiles", None)
    return keep


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
def run_chr21(args, rank, world, dev_id, torch, dist, coll_dev):
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
    gbed_path = os.path.join(tmpd, f"g{rank}.bed")
    counter = CallableProfiler(gbed_path)
    st = ContigProfiler("chr21", L)
    t0 = time.perf_counter()
    process_single_contig(eng, counter, st, opt, tid, rec, ref)
    counter.close()
    e2e = time.perf_counter() - t0
    first = eng.contig_collect()
    log(f"[bench r{rank}] end-to-end first pass (host admission + PCIe + kernels + BED): {e2e:.2f}s "
        f"= {L / e2e / 1e9:.3f} Gbase/s; intervals {first.summary.n_intervals}")
    # the same call once more: what every contig but a process's first costs (device and staging buffers exist,
    # code objects are loaded); its BED goes to a scratch file
    counter2 = CallableProfiler(os.path.join(tmpd, f"g{rank}_again.bed"))
    t0 = time.perf_counter()
    process_single_contig(eng, counter2, ContigProfiler("chr21", L), opt, tid, rec, ref)
    counter2.close()
    e2e_next = time.perf_counter() - t0
    log(f"[bench r{rank}] end-to-end, the same contig again on the warm engine: {e2e_next:.3f}s")

    # ---- timed region: resident contig, blocks of K steps ----
    dts = timed_blocks(eng.contig_run, eng.sync, args, world, dist, torch, coll_dev)
    kms, _ = kernel_profile([eng], args.steps)
    again = eng.contig_collect()
    assert again.as_dict() == first.as_dict() and np.array_equal(again.intervals, first.intervals), \
        "resident re-run changed the result"

    # RCCL gather of the per-contig summaries (the path's only exchange), straight from HBM
    from decodingustools_amd.coverage import device_summary_tensor
    summ = device_summary_tensor(eng).clone()
    if coll_dev == "cpu":
        summ = summ.cpu()
    if world > 1:
        allsum = [torch.zeros_like(summ) for _ in range(world)]
        dist.all_gather(allsum, summ)
    else:
        allsum = [summ]
    total_bases = sum(int(s[11].item()) for s in allsum)           # word 11: extent
    if rank != 0:
        eng.close()
        return None
    dt_mean = sum(dts) / len(dts)
    ms_step = dt_mean * 1e3 / args.steps
    alg, layb, rleb, lay = engine_figures(eng, rec, L)
    out = {
        "metric": METRIC,
        "value": total_bases / (dt_mean / args.steps), "unit": "bases/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8 (quality bytes thresholded to pass bits on the host, counted bit-sliced on the device)", "data": "synthetic",
        "config": {"workload": "coverage -L chr21 synthetic 30x 150bp paired, device-resident, one contig per GPU",
                   "contig_len": L, "depth": args.depth, "reads_per_contig": rec.n,
                   "aligned_bases_per_contig": int(rec.qual.shape[0]), "parallelism": f"contig-per-gpu x{world}",
                   "options": "cli defaults (4,500,10,20,10,1,0.1)"},
        "timing": {"blocks": len(dts), "steps_per_block": args.steps, "timed_region_s": sum(dts),
                   "ms_per_step_min": min(dts) * 1e3 / args.steps, "ms_per_step_max": max(dts) * 1e3 / args.steps},
        "roofline": roofline(alg, kms["pileup"], ms_step, kms, layout_bytes=layb, rle_bytes=rleb),
        "layout": lay,
        "end_to_end_first_pass_s": e2e,
        "end_to_end_next_pass_s": e2e_next,
    }
    if world == 1 and args.cpu_sample != 0:
        Ls = L if args.cpu_sample < 0 else min(args.cpu_sample, L)
        sub, ost, obed, cdt = cpu_baseline(rec, ref, Ls, opt, tid)
        if Ls == L:
            gbed = open(gbed_path).read()            # the first pass above: the whole contig
            gst = st
        else:                                        # the same sample through the GPU engine
            counter = CallableProfiler(os.path.join(tmpd, "gs.bed"))
            gst = ContigProfiler("chr21", Ls)
            process_single_contig(eng, counter, gst, opt, tid, sub, ref[:Ls])
            counter.close()
            gbed = open(os.path.join(tmpd, "gs.bed")).read()
        exact = (gbed == obed) and all(getattr(gst, k) == ost[k] for k in
                                       ("n_covered_bases", "summed_coverage", "summed_baseq", "summed_mapq",
                                        "quality_bases", "n_reads"))
        out["cpu_baseline"] = {"value": Ls / cdt, "unit": "bases/s", "cores": 1, "kind": "port",
                               "sample": (f"the whole contig, {Ls} positions" if Ls == L else f"first {Ls} positions of the same contig")
                                         + f" ({sub.n} reads), oracle/callable_oracle.c single thread, {cdt:.1f}s"}
        try:
            allc = cpu_baseline_all_cores(rec, ref, opt, tid, 2_000_000, min(16, os.cpu_count() or 1))
            if allc:
                out["cpu_baseline"]["all_cores"] = allc
        except Exception as e:                      # the single-thread figure above is the contract's
            log(f"[bench] all-cores CPU baseline skipped: {e}")
        out["bed_bit_exact"] = bool(exact)
        out["bed_bit_exact_positions"] = Ls
        if not exact:
            log("[bench] WARNING: GPU BED/summary differs from the oracle")
    eng.close()
    if world != 1:
        return out
    # the byte form of rounds 1-3 on the same input (DUT_QUAL_FORM=bytes): its >= 50 %-of-HBM figure stays reproducible
    try:
        out["byte_form"] = byte_form_figures(dev_id, opt, rec, ref, "chr21", tid, L, gbed_path, tmpd)
    except Exception as e:
        out["byte_form"] = {"error": str(e)}
    # ---- the secondary measurements, each under the time budget; none may cost the headline ----
    t_start = args._t_start
    cache_dir = None
    profiled = any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES"))
    want_traffic = not args.no_traffic and not profiled
    if want_traffic:
        try:
            cache_dir = tempfile.mkdtemp(prefix="bench_inputs_", dir="/tmp")
            if not save_records(cache_dir, "chr21", rec, ref):
                cache_dir = None
        except Exception:
            cache_dir = None

    def left():
        return args.time_budget - (time.perf_counter() - t_start)
    skipped = []
    if not args.no_files:
        if left() > 60:
            try:
                out["end_to_end_from_bam"] = files_config(dev_id, opt, rec, ref, L, gbed_path, tmpd)
                log(f"[bench] BAM + FASTA files -> BED: {out['end_to_end_from_bam'].get('seconds')} s")
            except Exception as e:
                out["end_to_end_from_bam"] = {"error": str(e)}
        else:
            skipped.append("end_to_end_from_bam")
    del rec, ref
    if not args.no_files:
        if left() > 150:
            try:
                out["end_to_end_from_bam_multi"] = files_multi_config(dev_id, opt, args, tmpd)
                log(f"[bench] 25-contig BAM + FASTA -> BED: {out['end_to_end_from_bam_multi'].get('seconds')} s")
            except Exception as e:
                out["end_to_end_from_bam_multi"] = {"error": str(e)}
        else:
            skipped.append("end_to_end_from_bam_multi")
    if cache_dir and args.long_length == L:
        log("[bench] note: the long-read contig has the headline's length: their kernels cannot be told apart by grid")
    out["configs"] = {}
    if not args.no_secondary:
        for key, need, fn in (("long_read_chrY_50x", 150, lambda: long_read_config(dev_id, opt, args, cache_dir, tmpd)),
                              ("site_pileup_chrY_40x", 90, lambda: site_config(dev_id, opt, args, cache_dir))):
            if left() < need:
                skipped.append(key)
                continue
            try:
                out["configs"][key] = fn()
            except Exception as e:                          # a secondary line must not cost the headline
                out["configs"][key] = {"error": str(e)}
    if want_traffic and cache_dir:
        if left() > 100:
            t0 = time.perf_counter()
            tr, why = measure_traffic(cache_dir)
            if tr is not None:
                def rows_key(length):
                    # Grid_Size counts work-items: windows rounded up to a multiple of 8, times the threads per window
                    wins = ((((length + 2047) // 2048) + 7) // 8) * 8
                    for bs in (128, 256):
                        if "k_pileup_rows@" + str(wins * bs) in tr:
                            return "k_pileup_rows@" + str(wins * bs)
                    return None
                k0 = tr.get(rows_key(L)) or tr.get("k_pileup_form0")
                if k0:
                    out["roofline"]["traffic"] = k0["bytes"]
                    out["roofline"]["traffic_raw_bytes"] = k0["raw_bytes"]
                    out["roofline"]["traffic_detail"] = k0
                lr = out["configs"].get("long_read_chrY_50x", {})
                k2 = tr.get(rows_key(args.long_length)) or tr.get("k_pileup_form2")
                if "roofline" in lr and k2:
                    lr["roofline"]["traffic"] = k2["bytes"]
                    lr["roofline"]["traffic_raw_bytes"] = k2["raw_bytes"]
                    lr["roofline"]["traffic_detail"] = k2
                sp = out["configs"].get("site_pileup_chrY_40x", {})
                if "roofline" in sp and tr.get("k_site_pileup"):
                    d = tr["k_site_pileup"]
                    rf = sp["roofline"]
                    rf["traffic"] = d["bytes"]
                    rf["traffic_detail"] = d
                    rf["traffic_raw_bytes"] = int((d["FETCH_SIZE_KB"] + d["WRITE_SIZE_KB"]) * 1024)
                    if sp.get("kernel_ms"):
                        # the measured bytes are the figure: the nominal count prices bases the kernel never touches
                        rf["nominal_achieved"], rf["nominal_frac"] = rf["achieved"], rf["frac"]
                        rf["achieved"] = rf["measured_achieved"] = d["bytes"] / (sp["kernel_ms"] * 1e-3) / 1e9
                        rf["frac"] = rf["measured_frac"] = rf["achieved"] / PEAK_HBM_GBS
                        rf["note"] = ("frac is on the MEASURED bytes (PMC, x2 rule; raw figure beside it); nominal_frac prices SURVEY 8d's "
                                      "count, which includes the 4-bit bases of every read -- the kernel touches the bases at sites only")
                log(f"[bench] HBM traffic by PMC ({time.perf_counter() - t0:.0f}s): " +
                    ", ".join(f"{k} {v['bytes'] / 1e6:.1f} MB" for k, v in tr.items()))
            else:
                out["roofline"]["traffic_error"] = why
                log(f"[bench] traffic not measured: {why}")
        else:
            skipped.append("traffic")
    if cache_dir:
        import shutil
        shutil.rmtree(cache_dir, ignore_errors=True)
    for v in out["configs"].values():
        if isinstance(v, dict):
            v.pop("_cached", None)
    if not args.no_wgs_point:
        if left() > 170:
            try:
                out["configs"]["wgs_1gpu"] = wgs_point(args, dev_id, torch, dist, coll_dev)
                # the headline (chr21, one contig per GPU: weak) and the N > 1 lines (the fixed whole genome: strong) are
                # different workloads: THIS is the N = 1 point an N > 1 `value` compares with
                out["scaling_reference"] = {"workload": "wgs", "value": out["configs"]["wgs_1gpu"]["value"], "unit": "bases/s",
                                            "note": "bench.py --gpus N (N > 1) measures the fixed whole-genome input; compare its value with this, not with the chr21 headline"}
            except Exception as e:
                out["configs"]["wgs_1gpu"] = {"error": str(e)}
        else:
            skipped.append("wgs_1gpu")
    if skipped:
        out["skipped_for_time_budget"] = skipped
    out["bench_wall_s"] = round(time.perf_counter() - t_start, 1)
    return out


def run_wgs(args, rank, world, dev_id, torch, dist, coll_dev):
    """BASELINE.json configs[3]: fixed whole-genome input, LPT-dealt, strong scaling."""
    from decodingustools_amd import CallableOptions, wgs
    opt = CallableOptions()
    stream = torch.cuda.Stream(device=dev_id)            # the collective's stream; the rank's engines alternate between it and one more (wgs.py)
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        shard = wgs.build_shard(rank, world, dev_id, opt, depth=args.depth, scale=args.wgs_scale,
                                stream=stream.cuda_stream, gen_threads=args.gen_threads, log=log)
        build_s = time.perf_counter() - t0
        last = {}

        def step():
            shard.step()
            last["g"] = shard.gather_device(coll_dev)     # all_gather of the summary records, device to device

        def sync():
            stream.synchronize()
        dts = timed_blocks(step, sync, args, world, dist, torch, coll_dev)
        table = shard.parse(last["g"])
        kms, _ = kernel_profile([c.engine for c in shard.mine], args.steps, one_at_a_time=shard.side_stream is not None)
        stream.synchronize()
        for st in shard.side_streams:
            st.synchronize()
        # what every rank did in one step, and how even the deal was
        mine_ms = sum(kms.values())
        lays = [c.engine.contig_layout() for c in shard.mine]
        alg_mine = sum(survey_bytes(c.aligned_bases, c.n_reads, c.length, 0.125 if l["form"] == 3 else 1.0) for c, l in zip(shard.mine, lays))
        info = torch.tensor([float(shard.bases), mine_ms, float(sum(c.first_pass_s for c in shard.mine)),
                             float(sum(c.aligned_bases for c in shard.mine)), kms.get("pileup", 0.0),
                             float(alg_mine), build_s, float(sum(sum(c.engine.contig_bytes()) for c in shard.mine)),
                             float(sum(l["device_bytes"] for l in lays)), float(sum(l["upload_h2d_bytes"] for l in lays))],
                            dtype=torch.float64, device=coll_dev)
        infos = [torch.zeros_like(info) for _ in range(world)]
        if world > 1:
            dist.all_gather(infos, info)
        else:
            infos = [info]
    infos = [x.cpu().tolist() for x in infos]
    # every rank holds the whole table: check it against the rank's own first pass
    for c in shard.mine:
        row = table[c.tid]
        assert [int(x) for x in row[:6]] == c.outcome.state_counts and int(row[13]) == c.outcome.intervals.shape[0], \
            f"gathered summary of {c.name} differs from the first pass"
    total_bases = sum(int(table[t][11]) for t, _, _ in shard.contigs)       # word 11: extent (= length, no overhang)
    out = None
    if rank == 0:
        dt_mean = sum(dts) / len(dts)
        ms_step = dt_mean * 1e3 / args.steps
        loads = [x[0] for x in infos]
        alg = sum(x[5] for x in infos)
        pile_ms = sum(x[4] for x in infos)
        callable_b = sum(int(table[t][1]) for t, _, _ in shard.contigs)
        out = {
            "metric": METRIC,
            "value": total_bases / (dt_mean / args.steps), "unit": "bases/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "whole-genome synthetic 30x hg38 (25 primary contigs), contigs LPT-dealt to the GPUs, "
                                   "device-resident, one all_gather of the per-contig summaries per step",
                       "contigs": len(shard.contigs), "total_bases": total_bases, "depth": args.depth, "wgs_scale": args.wgs_scale,
                       "aligned_bases": int(sum(x[3] for x in infos)), "parallelism": f"contig-sharded x{world} (LPT)",
                       "collective_backend": args.backend, "options": "cli defaults (4,500,10,20,10,1,0.1)"},
            "timing": {"blocks": len(dts), "steps_per_block": args.steps, "timed_region_s": sum(dts),
                       "ms_per_step_min": min(dts) * 1e3 / args.steps, "ms_per_step_max": max(dts) * 1e3 / args.steps},
            "sharding": {"per_rank_bases": [int(x) for x in loads], "per_rank_kernel_ms": [x[1] for x in infos],
                         "per_rank_first_pass_s": [x[2] for x in infos],
                         "lpt_imbalance": max(loads) / (sum(loads) / len(loads)),
                         "contigs_of_rank": [[n for (_, n, _), r in zip(shard.contigs, shard.rank_of) if r == k] for k in range(world)],
                         "build_s": [x[6] for x in infos],
                         "host_threads_per_rank": int(os.environ.get("DUT_THREADS", "0")) or None, "gen_threads": args.gen_threads},
            # the dominant kernel over the whole job: every rank's algorithmic bytes / every rank's pileup kernel time
            "roofline": roofline(alg, pile_ms, ms_step, kms, world, layout_bytes=sum(x[7] for x in infos)),
            "callable_fraction": callable_b / max(total_bases, 1),
            # self-describing for a reader of `value` across N: this is the FIXED whole-genome input (strong scaling)
            "total_bases": total_bases,
            "lpt_imbalance": max(loads) / (sum(loads) / len(loads)),
            "hbm_resident_bytes_per_rank": [int(x[8]) for x in infos],
            "upload_h2d_bytes_per_rank": [int(x[9]) for x in infos],
        }
        ref1 = cached_wgs_1gpu(total_bases, args)
        if world > 1 and ref1:
            out["value_vs_wgs_1gpu"] = out["value"] / ref1["value"]
            out["wgs_1gpu_reference"] = ref1
        out["roofline"]["per_rank_frac"] = [x[5] / (x[4] * 1e-3) / 1e9 / PEAK_HBM_GBS if x[4] > 0 else 0.0 for x in infos]
        out["roofline"]["all_kernel_ms_of"] = "rank 0"
    shard.close()
    return out


def host_cpu_budget():
    """CPUs this process may actually use: the affinity mask, cut by the cgroup's CFS quota (a GPU box shows 256
    CPUs and grants 16 CPUs' worth of time)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(math.ceil(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(math.ceil(q / per))))
            break
        except Exception:
            continue
    return max(1, n)


def launch_ranks(args, argv):
    """`bench.py --gpus N` (N > 1) started plainly, with no launcher around it: this process touches neither the GPU
    nor the engine; it compiles what is stale, starts the N ranks as fresh processes under torch.distributed.run (one per
    GPU, rendezvous on 127.0.0.1), hands each its share of the host (DUT_THREADS = cpu budget / N), relays rank 0's single
    JSON line and exits with the children's status.  It never prints a line of its own."""
    import socket
    import subprocess
    from decodingustools_amd import build as _native_build
    _native_build.build()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    budget = host_cpu_budget()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("DUT_THREADS", str(max(2, min(64, budget // args.gpus))))
    env.setdefault("DUT_COPY_THREADS", str(max(2, min(8, budget // args.gpus))))
    env["BENCH_HOST_BUDGET"] = str(budget)
    launcher = os.environ.get("BENCH_LAUNCHER", "torch.distributed.run")
    cmd = [sys.executable, "-m", launcher, "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    log(f"[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks (host budget {budget} CPUs, "
        f"DUT_THREADS={env['DUT_THREADS']} per rank)")
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    except Exception as e:
        log(f"[bench] cannot start the ranks: {e}")
        return 3
    lines = [ln for ln in (r.stdout or "").splitlines() if ln.startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        log(f"[bench] the ranks ended with status {r.returncode} and {len(lines)} JSON line(s): no result")
        return r.returncode or 4
    try:
        ok = json.loads(lines[0]).get("n_gpus") == args.gpus
    except Exception:
        ok = False
    if not ok:
        log(f"[bench] the line the ranks printed is not an n_gpus = {args.gpus} line: not relayed")
        return 5
    print(lines[0], flush=True)
    return 0


def launch_check(args, rank, world, dist, torch):
    """--launch-check: the ranks rendezvous, exchange their numbers and rank 0 prints a line -- nothing touches the GPU or
    the engine (the CPU test of the self-launch path)."""
    t = torch.tensor([rank], dtype=torch.int64)
    got = [torch.zeros_like(t) for _ in range(world)]
    if world > 1:
        dist.all_gather(got, t)
    else:
        got = [t]
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": [int(x.item()) for x in got],
                          "dut_threads": os.environ.get("DUT_THREADS"), "host_budget": os.environ.get("BENCH_HOST_BUDGET")}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["auto", "chr21", "wgs"], default="auto",
                    help="auto: chr21 (configs[1]) on one GPU, the whole genome (configs[3]) on several")
    ap.add_argument("--length", type=int, default=46_709_983, help="chr21 workload: contig length")
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--min-time", type=float, default=0.6, help="seconds of timed K-step blocks to accumulate")
    ap.add_argument("--max-blocks", type=int, default=400)
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="positions of the CPU baseline / bit-exactness sample (-1 = the whole contig, 0 = skip)")
    ap.add_argument("--wgs-scale", type=float, default=1.0, help="wgs workload: contig lengths x this (rehearsals)")
    ap.add_argument("--gen-threads", type=int, default=0, help="wgs workload: contigs generated ahead on host threads (0: by the host budget)")
    ap.add_argument("--long-length", type=int, default=57_227_415, help="secondary config 3: contig length (chrY)")
    ap.add_argument("--site-length", type=int, default=57_227_415, help="secondary config 5: contig length (chrY)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configs[2] / configs[4] lines")
    ap.add_argument("--no-wgs-point", action="store_true", help="N = 1: skip configs.wgs_1gpu (the whole genome on this one GPU)")
    ap.add_argument("--no-files", action="store_true", help="N = 1: skip end_to_end_from_bam (BAM + FASTA files -> BED)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic (N = 1)")
    ap.add_argument("--time-budget", type=float, default=480.0,
                    help="N = 1: seconds after which the remaining secondary measurements are skipped (the headline never is)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the multi-rank logic on one GPU)")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous of the ranks only: no GPU, no engine (tests)")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    args._t_start = time.perf_counter()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started plainly for several GPUs: this process becomes the launcher, before anything touches a GPU
        sys.exit(launch_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: refusing to measure something else than what was asked for")
        sys.exit(2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    budget = int(os.environ.get("BENCH_HOST_BUDGET", "0")) or host_cpu_budget()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # every rank its share of the host (a launcher other than this script's own may not have set it)
    os.environ.setdefault("DUT_THREADS", str(max(2, min(64, budget // max(1, local_world)))))
    if args.gen_threads <= 0:
        args.gen_threads = max(1, min(8, budget // max(1, local_world) // 2))

    # torch first: it brings its own HIP runtime, and the engine's library must resolve against that one
    # (loading the engine before torch leaves the process with two runtimes and no usable device)
    import torch
    import torch.distributed as dist
    if args.launch_check:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            launch_check(args, rank, world, dist, torch)
        finally:
            if world > 1:
                dist.destroy_process_group()
        return
    # the native pieces are (re)built before anything initialises the GPU: a process that has must not
    # start compilers (normally nothing is stale and this returns at once; every rank checks).  Compile
    # only -- the library itself is loaded further down, after the device is set.
    from decodingustools_amd import build as _native_build
    _native_build.build()
    if world == 1:                               # the checker of the N = 1 line; no other rank configuration uses it
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
    if args.pmc_child:
        pmc_child(args.pmc_child, dev_id)
        return
    workload = args.workload if args.workload != "auto" else ("chr21" if world == 1 else "wgs")
    try:
        out = (run_chr21 if workload == "chr21" else run_wgs)(args, rank, world, dev_id, torch, dist, coll_dev)
        if rank == 0:
            print(json.dumps(out), flush=True)
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()

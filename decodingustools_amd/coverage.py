"""Host-side mirror of the reference's coverage driver (src/api/coverage.rs) over decoded,
in-memory contigs, plus the multi-GPU form of it.

    CoverageInput / CoverageOutput / CoverageAnalyzer.analyze     api/coverage.rs:22-145
    initialize_contig_stats / validate_contig_selection           api/coverage.rs:149-204
    process_contigs_api (ascending tid, serial)                   api/coverage.rs:221-236
    build_coverage_export numbers                                 callable_loci/report.rs:15-134

Multi-GPU (`analyze_sharded`): contigs are independent (the only cross-contig state of the
reference is the BED writer's pending line, an output-formatting matter reproduced on rank 0), so
they are dealt to ranks by longest-processing-time-first on their aligned bases; every rank runs
its contigs on its own GPU with no data-path collective; per-contig summaries are exchanged with
one all_gather (RCCL over xGMI when the backend is "nccl") and the run lists are sent to rank 0,
which writes the BED in tid order.
"""
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import numpy as np

from .callable_loci import (CallableOptions, CallableProfiler, ContigProfiler, ContigResult, Engine,
                            admit_reads, genome_summary)
from .records import ContigRecords


class ApiError(RuntimeError):
    """ApiError::Analysis(String), api/mod.rs:24-41."""


@dataclass
class ContigInput:
    name: str
    length: int
    records: Optional[ContigRecords]       # None for contigs that are read from a file when they are processed
    ref: Optional[np.ndarray] = None       # FASTA bytes of the contig, case preserved
    weight: Optional[int] = None           # balancing weight; default: aligned bases + length


@dataclass
class CoverageInput:
    contigs: List[ContigInput]             # header (@SQ) order; index in this list = tid
    options: CallableOptions = field(default_factory=CallableOptions)
    selected: Optional[List[str]] = None   # -L / --contig
    output_bed: str = "callable_regions.bed"


@dataclass
class CoverageOutput:
    export: Dict
    bed_file: str


@dataclass
class ContigOutcome:
    """What one process_single_contig leaves behind (ContigProfiler + counts + runs)."""
    tid: int
    stats: ContigProfiler
    state_counts: List[int]
    intervals: np.ndarray                  # (n,3) uint32
    dev_summary: object = None             # optional: the engine's resident cl_contig_summary (device tensor view)



def initialize_contig_stats(inp: CoverageInput):
    """api/coverage.rs:149-185: one ContigProfiler per header contig, filtered by -L."""
    sel = set(inp.selected) if inp.selected is not None else None
    out = {}
    for tid, c in enumerate(inp.contigs):
        if sel is not None and c.name not in sel:
            continue
        out[tid] = ContigProfiler(c.name, c.length)
    return out


def validate_contig_selection(stats, inp: CoverageInput):
    """api/coverage.rs:187-204."""
    if inp.selected is not None and not stats:
        raise ApiError("None of the specified contigs ({}) were found in the BAM file".format(
            ", ".join(inp.selected)))


def engine_process_contig(engine: Engine, options: CallableOptions, tid: int, c: ContigInput,
                          with_dev_summary: bool = False) -> ContigOutcome:
    """process_single_contig (mod.rs:44-147) on the device engine, returning the runs instead of
    writing them (the caller owns the BED writer)."""
    acc, n_names = admit_reads(options, tid, c.length, c.records)
    rec = c.records
    idx = np.flatnonzero(acc)
    engine.contig_begin(tid, c.length, c.ref)
    if idx.shape[0]:
        if idx.shape[0] == rec.n:
            engine.push_reads(rec.pos, rec.mapq, rec.cigar_off, rec.cigar, rec.qual_off, rec.qual)
        else:
            clen = (rec.cigar_off[1:] - rec.cigar_off[:-1]).astype(np.int64)
            qlen = (rec.qual_off[1:] - rec.qual_off[:-1]).astype(np.int64)
            cig = rec.cigar[np.repeat(acc, clen)]
            qual = rec.qual[np.repeat(acc, qlen)]
            coff = np.concatenate([[0], np.cumsum(clen[idx])]).astype(np.uint32)
            qoff = np.concatenate([[0], np.cumsum(qlen[idx])]).astype(np.uint64)
            engine.push_reads(rec.pos[idx], rec.mapq[idx], coff, cig, qoff, qual)
    res: ContigResult = engine.contig_finish()
    st = ContigProfiler(c.name, c.length)
    s = res.summary
    st.n_covered_bases = int(s.n_covered_bases); st.summed_coverage = int(s.summed_coverage)
    st.summed_baseq = int(s.summed_baseq); st.summed_mapq = int(s.summed_mapq)
    st.quality_bases = int(s.quality_bases); st.n_reads = int(n_names)
    return ContigOutcome(tid=tid, stats=st, state_counts=res.state_counts, intervals=res.intervals,
                         dev_summary=device_summary_tensor(engine) if with_dev_summary else None)


def build_coverage_export(outcomes: List[ContigOutcome]) -> Dict:
    """The numeric part of report.rs:15-134 (BamStats-derived strings are out of scope)."""
    stats = [o.stats for o in outcomes]
    callable_counts = [o.state_counts[1] for o in outcomes]
    g = genome_summary(stats, callable_counts)
    by_name = {o.stats.name: o for o in outcomes}
    contigs = []
    for name in g["order"]:
        o = by_name[name]
        d = o.stats.derived()
        c = o.state_counts
        contigs.append(dict(
            name=name, length=o.stats.length, unique_reads=o.stats.n_reads,
            coverage_percent=d["coverage_percent"], average_depth=d["average_depth"],
            covered_bases=o.stats.n_covered_bases, total_bases=o.stats.length,
            quality_stats=dict(average_mapq=d["average_mapq"], average_baseq=d["average_baseq"],
                               q30_percentage=d["q30_percentage"]),
            state_distribution=dict(ref_n=c[0], callable=c[1], no_coverage=c[2], low_coverage=c[3],
                                    excessive_coverage=c[4], poor_mapping_quality=c[5])))
    return dict(
        summary=dict(total_bases=g["total_bases"], callable_bases=g["callable_bases"],
                     callable_percentage=g["callable_percentage"], average_depth=g["average_depth"],
                     contigs_analyzed=g["contigs_analyzed"]),
        contigs=contigs,
        quality_metrics=dict(average_mapq=g["average_mapq"], average_baseq=g["average_baseq"],
                             q30_percentage=g["q30_percentage"]),
        total_unique_reads=g["total_unique_reads"])


def write_bed(outcomes: List[ContigOutcome], bed_path: str):
    """One CallableProfiler over all contigs in ascending tid (api/coverage.rs:229-234), which
    reproduces the duplicated last line of every contig but the last (callable_profiler.rs:64-66)."""
    counter = CallableProfiler(bed_path)
    # the coverage figures beside the BED (callable_profiler.rs:64-84; stride from the longest contig but chrM,
    # api/coverage.rs:210-215)
    counter.enable_plots(max([o.stats.length for o in outcomes if o.stats.name != "chrM"], default=0))
    try:
        for o in sorted(outcomes, key=lambda o: o.tid):
            class _S:                      # feed_contig only needs .state_counts and .intervals
                pass
            r = _S(); r.state_counts = o.state_counts; r.intervals = o.intervals
            counter.feed_contig(o.stats.name, r)
            counter.finish_plot(o.stats.name, o.stats.length)
    finally:
        counter.close()


class CoverageAnalyzer:
    """CoverageAnalyzer (api/coverage.rs:22-122) for decoded inputs on one GPU."""

    def __init__(self, device_id: int = 0):
        self.device_id = device_id
        self._progress = None

    def with_progress(self, callback: Callable):
        self._progress = callback
        return self

    def _emit(self, kind, task):
        if self._progress:
            self._progress(dict(event=kind, task=task))

    def analyze(self, inp: CoverageInput) -> CoverageOutput:
        self._emit("Started", "Coverage Analysis")
        stats = initialize_contig_stats(inp)
        validate_contig_selection(stats, inp)
        outcomes = []
        with Engine(inp.options, self.device_id) as eng:
            for tid in sorted(stats):
                try:
                    outcomes.append(engine_process_contig(eng, inp.options, tid, inp.contigs[tid]))
                except Exception as e:       # api/coverage.rs:251
                    raise ApiError(f"Error processing contig: {e}") from e
        write_bed(outcomes, inp.output_bed)
        out = CoverageOutput(export=build_coverage_export(outcomes), bed_file=inp.output_bed)
        self._emit("Completed", "Coverage Analysis")
        return out


# ------------------------------------------------------------------------------------------------
# multi-GPU: one process per GPU, contigs dealt by LPT, one all_gather of summaries
# ------------------------------------------------------------------------------------------------
def lpt_assignment(weights: List[int], world: int) -> List[int]:
    """Longest-processing-time-first: returns the rank of every item (deterministic)."""
    load = [0] * world
    rank_of = [0] * len(weights)
    for i in sorted(range(len(weights)), key=lambda i: (-weights[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        rank_of[i] = r
        load[r] += weights[i]
    return rank_of


def _outcome_row(o: ContigOutcome) -> List[int]:
    s = o.stats
    return list(o.state_counts) + [s.n_covered_bases, s.summed_coverage, s.summed_baseq, s.summed_mapq,
                                   s.quality_bases, s.n_reads, s.length, int(o.intervals.shape[0])]


# columns of a gathered summary row: the contig's tid, the 14 words of cl_contig_summary exactly as the
# engine leaves them in HBM (include/callable_loci.h: state_counts[6], n_covered_bases, summed_coverage,
# summed_baseq, summed_mapq, quality_bases, extent, max_raw_depth, n_intervals), then the host-side
# distinct-name count
ROW_WORDS = 16


class DeviceWords:
    """A zero-copy view of `n` 64-bit words at a device address (`__cuda_array_interface__`, which
    torch.as_tensor understands on ROCm too): how the resident cl_contig_summary of an engine
    (cl_device_summary) enters a collective without a host round trip."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr="<i8", data=(int(ptr), False), version=2)


def device_summary_tensor(engine: Engine):
    """The engine's resident summary record as a torch int64 tensor on its device (no copy)."""
    import torch
    ptr, nbytes = engine.device_summary()
    return torch.as_tensor(DeviceWords(ptr, nbytes // 8), device=torch.device("cuda", engine.device_id))


def _exchange_status(rank: int, world: int, error: Optional[str], group=None) -> None:
    """Every rank learns whether any rank failed before the data collectives start: a rank that raised in
    its contig loop would otherwise leave the others blocked in the all_gather (with RCCL until the
    watchdog fires).  Raises the reference's error text (api/coverage.rs:251) on every rank."""
    import torch.distributed as dist
    if world > 1:
        status = [None] * world
        dist.all_gather_object(status, error, group=group)
    else:
        status = [error]
    bad = [(r, m) for r, m in enumerate(status) if m is not None]
    if bad:
        r, m = bad[0]
        raise ApiError(f"Error processing contig: {m}" + (f" (rank {r})" if world > 1 else ""))


def analyze_sharded(inp: CoverageInput, rank: int, world: int, process_contig: Callable[[int, ContigInput], ContigOutcome],
                    device="cpu", group=None, on_assignment: Optional[Callable[[List[int]], None]] = None) -> Optional[CoverageOutput]:
    """Every rank calls this with the same `inp` description (it only touches the contigs it is
    assigned).  `process_contig(tid, contig)` runs one contig on this rank's GPU
    (engine_process_contig bound to the rank's Engine).  Returns the output on rank 0, None elsewhere.
    The only exchanges are a status word per rank, the summary all_gather and the run lists sent to rank 0.

    `device`: where the collectives' tensors live -- "cuda" with the nccl (= RCCL) backend, "cpu" with gloo.
    An outcome may carry `dev_summary`, the engine's resident cl_contig_summary as a device tensor
    (device_summary_tensor): its words then go from HBM into the gathered row device to device."""
    import torch
    import torch.distributed as dist
    stats = initialize_contig_stats(inp)
    validate_contig_selection(stats, inp)
    tids = sorted(stats)
    weights = [inp.contigs[t].weight if inp.contigs[t].weight is not None
               else int(inp.contigs[t].records.qual.shape[0]) + inp.contigs[t].length for t in tids]
    rank_of = lpt_assignment(weights, world)
    mine = [t for t, r in zip(tids, rank_of) if r == rank]
    if on_assignment is not None:
        on_assignment(list(mine))                      # the order process_contig will be called in (read-ahead hook)

    # --- this rank's contigs; summaries land in fixed-size rows (one all_gather below) ---
    per_rank = max(1, max(rank_of.count(r) for r in range(world)) if tids else 1)
    on_dev = str(device).startswith("cuda")
    rows = torch.zeros((per_rank, ROW_WORDS), dtype=torch.int64, device=device if on_dev else "cpu")
    rows[:, 0] = -1
    local = {}
    error = None
    try:
        for i, t in enumerate(mine):
            o = process_contig(t, inp.contigs[t])
            local[t] = o
            dv = getattr(o, "dev_summary", None)
            if dv is not None and on_dev:
                rows[i, 1:15].copy_(dv)                # HBM -> HBM; the engine has synchronised its stream
                torch.cuda.current_stream().synchronize()   # ... and the record is free for the engine's next contig
                o.dev_summary = None                   # the view dies with the engine's next contig
                head_tail = torch.tensor([t, o.stats.n_reads], dtype=torch.int64).to(device)
                rows[i, 0] = head_tail[0]; rows[i, 15] = head_tail[1]
            else:
                r14 = _outcome_row(o)                  # counts[6], 5 sums, n_reads, length, n_intervals
                words = r14[:11] + [o.stats.length, 0, r14[13]]     # extent / max_raw_depth are not used downstream
                rows[i] = torch.tensor([t] + words + [o.stats.n_reads], dtype=torch.int64).to(rows.device)
    except Exception as e:                             # api/coverage.rs:251: the error is reported, not lost in a hang
        error = f"{e}"
    _exchange_status(rank, world, error, group)
    gathered = [torch.zeros_like(rows) for _ in range(world)]
    if world > 1:
        dist.all_gather(gathered, rows, group=group)
    else:
        gathered = [rows]
    table = {}
    for g in gathered:
        for row in g.cpu().tolist():
            if row[0] >= 0:
                table[int(row[0])] = row[1:]

    # --- run lists to rank 0: (start, end, state) as 32-bit triplets, padded to the largest list of any rank ---
    n_iv_rank = [sum(table[t][13] for t, r in zip(tids, rank_of) if r == rr) for rr in range(world)]
    cap = max(1, max(n_iv_rank) if n_iv_rank else 1)
    buf = torch.zeros((cap, 3), dtype=torch.int32)
    off = 0
    for t in mine:
        iv = np.ascontiguousarray(local[t].intervals, np.uint32)
        buf[off:off + iv.shape[0]] = torch.from_numpy(iv.view(np.int32))      # same bits
        off += iv.shape[0]
    buf = buf.to(device)
    if world > 1:
        bufs = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, bufs, dst=0, group=group)
    else:
        bufs = [buf]
    if rank != 0:
        return None
    outcomes = []
    offs = [0] * world
    host_bufs = [b.cpu().numpy().view(np.uint32) for b in bufs]
    for t, r in zip(tids, rank_of):
        row = table[t]
        n_iv = int(row[13])
        iv = host_bufs[r][offs[r]:offs[r] + n_iv].copy()
        offs[r] += n_iv
        c = inp.contigs[t]
        st = ContigProfiler(c.name, c.length, n_covered_bases=int(row[6]), summed_coverage=int(row[7]),
                            summed_baseq=int(row[8]), summed_mapq=int(row[9]), quality_bases=int(row[10]),
                            n_reads=int(row[14]))
        outcomes.append(ContigOutcome(tid=t, stats=st, state_counts=[int(x) for x in row[:6]], intervals=iv))
    write_bed(outcomes, inp.output_bed)
    out = CoverageOutput(export=build_coverage_export(outcomes), bed_file=inp.output_bed)
    out.outcomes = outcomes
    return out


def engine_process_contig_runs(engine: Engine, options: CallableOptions, tid: int, name: str, length: int,
                               rec: ContigRecords, ref: Optional[np.ndarray], with_dev_summary: bool = False) -> ContigOutcome:
    """process_single_contig through the C driver (admission, one zero-copy tile, kernels), returning
    the runs instead of writing them."""
    import ctypes as C
    from . import _lib
    from .callable_loci import EngineError, _ptr, _records_c
    lib = _lib.load()
    ref = np.ascontiguousarray(ref, np.uint8) if ref is not None else np.zeros(0, np.uint8)
    st = ContigProfiler(name, length)
    oc = options.to_c(); rc = _records_c(rec); cs = st._c()
    counts = (C.c_uint64 * 6)(); iv = C.POINTER(_lib.cl_interval)(); n = C.c_size_t()
    status = lib.dut_process_single_contig_runs(engine._h, C.byref(cs), C.byref(oc), tid, length, _ptr(ref), ref.shape[0],
                                                C.byref(rc), counts, C.byref(iv), C.byref(n))
    if status != 0:
        raise EngineError(status, lib.cl_last_error(engine._h).decode() or "admission failed")
    st._load(cs)
    runs = (np.ctypeslib.as_array(C.cast(iv, C.POINTER(C.c_uint32)), shape=(n.value, 3)).copy() if n.value
            else np.zeros((0, 3), np.uint32))
    return ContigOutcome(tid=tid, stats=st, state_counts=[int(x) for x in counts], intervals=runs,
                         dev_summary=device_summary_tensor(engine) if with_dev_summary else None)


def coverage_files_sharded(bam_file: str, reference_file: str, output_bed: str, summary_json: Optional[str],
                           options: CallableOptions, contigs: Optional[List[str]], rank: int, world: int,
                           device_id: int, coll_device="cpu", group=None, output_summary: str = "summary.html"):
    """`coverage` on files over `world` processes, one GPU each (torch.distributed already initialised
    when world > 1): every rank opens the BAM / FASTA itself and decodes only the contigs it is dealt
    (LPT on the index's mapped-read counts, else on contig length); rank 0 writes the BED in tid order
    and the summary.json of main.rs:68-69.  Returns the CoverageOutput on rank 0, None elsewhere."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from .bam import BamReader, FastaReader
    from .report import BamStats, coverage_output_json
    bam = BamReader(bam_file)
    fasta = FastaReader(reference_file)
    # with an index, the rank's next contig is read on a second reader (its own thread; the library calls
    # release the GIL) while the current one runs -- as dut_coverage_files does (DUT_PIPELINE=0: off)
    readers = [(bam, fasta)]
    pool = None
    if bam.has_index and os.environ.get("DUT_PIPELINE", "1") != "0":
        readers.append((BamReader(bam_file), FastaReader(reference_file)))
        pool = ThreadPoolExecutor(1)
    order: List[int] = []
    pending = {}

    def fetch(slot, tid, name):
        b, f = readers[slot]
        # a zero-length contig fetches nothing (the reference's loops over it run zero times, mod.rs:65-147)
        return b.fetch_contig(tid), (f.fetch(name) if b.target_lens[tid] > 0 else np.zeros(0, np.uint8))
    try:
        descr = []
        for t, (nm, ln) in enumerate(zip(bam.target_names, bam.target_lens)):
            w = bam.target_mapped[t]
            descr.append(ContigInput(nm, ln, None, None, weight=(w * 160 + ln) if w >= 0 else ln))
        inp = CoverageInput(contigs=descr, options=options, selected=contigs, output_bed=output_bed)
        with Engine(options, device_id) as eng:
            def run(tid, c):
                i = order.index(tid)
                # a reader's arrays stay valid until its next fetch: contigs alternate between the two readers
                rec, bases = pending.pop(tid).result() if tid in pending else fetch(i & 1 if pool else 0, tid, c.name)
                if pool and i + 1 < len(order):
                    nt = order[i + 1]
                    pending[nt] = pool.submit(fetch, (i + 1) & 1, nt, descr[nt].name)
                return engine_process_contig_runs(eng, options, tid, c.name, c.length, rec, bases,
                                                  with_dev_summary=str(coll_device).startswith("cuda"))
            out = analyze_sharded(inp, rank, world, run, device=coll_device, group=group, on_assignment=order.extend)
        if rank != 0:
            return None
        if summary_json:
            bs = BamStats(10000).collect_stats(bam_file)                       # api/coverage.rs:56-59
            oc = out.outcomes
            text = coverage_output_json([o.stats for o in oc], [o.stats.name for o in oc], [o.state_counts for o in oc],
                                        bs.aligner(), bs.reference_build(), bs.infer_platform(), bs.average_read_length(),
                                        output_bed, output_summary or "summary.html",
                                        [f"{o.stats.name}_coverage.svg" for o in oc if os.path.exists(f"{o.stats.name}_coverage.svg")])
            with open(summary_json, "w") as f:
                f.write(text)
            if output_summary:
                from .report import write_html_report
                write_html_report(output_summary, [o.stats for o in oc], [o.stats.name for o in oc], [o.state_counts for o in oc],
                                  bs.aligner(), bs.reference_build(), bs.infer_platform(), bs.average_read_length())
        return out
    finally:
        if pool:
            pool.shutdown(wait=True)
        for b, f in readers:
            b.close()
            f.close()

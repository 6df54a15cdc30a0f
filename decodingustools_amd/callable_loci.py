"""Host-side mirror of the reference's `callable_loci` module (src/callable_loci/) on the
MI355X engine.  Names follow the reference:

    CallableOptions          options.rs:2-38 (CLI defaults cli.rs:34-60)
    CalledState              types.rs:36-43
    CallableProfiler         profilers/callable_profiler.rs:11-160
    ContigProfiler           profilers/contig_profiler.rs:7-158
    process_single_contig    mod.rs:44-147

Everything numeric happens in libcallable_hip.so (HIP kernels + C++ host mirror); this file is
a thin ctypes layer over include/callable_loci.h and include/dut_coverage.h.
"""
import ctypes as C
import enum
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib
from .records import ContigRecords


class CalledState(enum.IntEnum):
    REF_N = 0
    CALLABLE = 1
    NO_COVERAGE = 2
    LOW_COVERAGE = 3
    EXCESSIVE_COVERAGE = 4
    POOR_MAPPING_QUALITY = 5


class EngineError(RuntimeError):
    """Box<dyn Error> of the reference, with the engine's status code."""

    def __init__(self, status, message):
        super().__init__(f"{message} (cl_status {status})")
        self.status = status


@dataclass
class CallableOptions:
    """options.rs:2-11; defaults are the CLI's (cli.rs:34-60)."""
    min_depth: int = 4
    max_depth: int = 500
    min_mapping_quality: int = 10
    min_base_quality: int = 20
    min_depth_for_low_mapq: int = 10
    max_low_mapq: int = 1
    max_low_mapq_fraction: float = 0.1
    selected_contigs: Optional[List[str]] = None

    def with_contigs(self, contigs):
        self.selected_contigs = list(contigs) if contigs is not None else None
        return self

    def to_c(self):
        return _lib.cl_options(self.min_depth, self.max_depth, self.min_mapping_quality,
                               self.min_base_quality, self.min_depth_for_low_mapq, self.max_low_mapq,
                               float(self.max_low_mapq_fraction))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _records_c(rec: ContigRecords):
    r = _lib.dut_records()
    r.n = rec.n
    r.pos = _ptr(rec.pos); r.flag = _ptr(rec.flag); r.mapq = _ptr(rec.mapq)
    r.cigar_off = _ptr(rec.cigar_off); r.cigar = _ptr(rec.cigar)
    r.qual_off = _ptr(rec.qual_off); r.qual = _ptr(rec.qual)
    r.qname_off = _ptr(rec.qname_off); r.qname = _ptr(rec.qname)
    return r


class Engine:
    """One device context (cl_ctx): one per GPU, driven by one host thread."""

    def __init__(self, options: CallableOptions, device_id: int = 0, stream: int = 0):
        self._lib = _lib.load()
        self.options = options
        self._opt_c = options.to_c()
        h = C.c_void_p()
        st = self._lib.cl_create(C.byref(self._opt_c), device_id, C.c_void_p(stream) if stream else None,
                                 C.byref(h))
        if st != 0:
            raise EngineError(st, "cl_create failed: no usable HIP device (the engine has no CPU fallback)")
        self._h = h
        self.device_id = device_id
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st):
        if st != 0:
            raise EngineError(st, self._lib.cl_last_error(self._h).decode())

    # ---- the C ABI, one to one ----
    def contig_begin(self, tid, contig_len, ref: Optional[np.ndarray]):
        ref = np.ascontiguousarray(ref, dtype=np.uint8) if ref is not None else np.zeros(0, np.uint8)
        self._check(self._lib.cl_contig_begin(self._h, tid, contig_len, _ptr(ref), ref.shape[0]))

    def contig_reserve(self, n_reads, n_cigar_ops, n_qual_bytes):
        self._check(self._lib.cl_contig_reserve(self._h, n_reads, n_cigar_ops, n_qual_bytes))

    def push_reads(self, pos, mapq, cigar_off, cigar, qual_off, qual):
        t = _lib.cl_read_tile()
        arrs = [np.ascontiguousarray(pos, np.int32), np.ascontiguousarray(mapq, np.uint8),
                np.ascontiguousarray(cigar_off, np.uint32), np.ascontiguousarray(cigar, np.uint32),
                np.ascontiguousarray(qual_off, np.uint64), np.ascontiguousarray(qual, np.uint8)]
        t.n_reads = arrs[0].shape[0]
        t.pos, t.mapq, t.cigar_off, t.cigar, t.qual_off, t.qual = [_ptr(a) for a in arrs]
        self._check(self._lib.cl_push_reads(self._h, C.byref(t)))

    def push_reads_bits(self, pos, mapq, cigar_off, cigar, qual_off, pass_bits, pass_sum):
        """cl_push_reads_bits: the packed pass-bitmask variant (the caller has taken the base-quality test)."""
        t = _lib.cl_read_tile_bits()
        arrs = [np.ascontiguousarray(pos, np.int32), np.ascontiguousarray(mapq, np.uint8),
                np.ascontiguousarray(cigar_off, np.uint32), np.ascontiguousarray(cigar, np.uint32),
                np.ascontiguousarray(qual_off, np.uint64), np.ascontiguousarray(pass_bits, np.uint64),
                np.ascontiguousarray(pass_sum, np.uint32)]
        t.n_reads = arrs[0].shape[0]
        t.pos, t.mapq, t.cigar_off, t.cigar, t.qual_off, t.pass_bits, t.pass_sum = [_ptr(a) for a in arrs]
        self._check(self._lib.cl_push_reads_bits(self._h, C.byref(t)))

    def contig_abort(self):
        self._check(self._lib.cl_contig_abort(self._h))

    def contig_upload(self):
        self._check(self._lib.cl_contig_upload(self._h))

    def contig_run(self):
        self._check(self._lib.cl_contig_run(self._h))

    def sync(self):
        self._check(self._lib.cl_sync(self._h))

    def _result(self, s, iv, n):
        n = n.value
        if n:
            arr = np.ctypeslib.as_array(C.cast(iv, C.POINTER(C.c_uint32)), shape=(n, 3)).copy()
        else:
            arr = np.zeros((0, 3), dtype=np.uint32)
        return ContigResult(summary=s, intervals=arr)

    def contig_collect(self):
        s = _lib.cl_contig_summary(); iv = C.POINTER(_lib.cl_interval)(); n = C.c_size_t()
        self._check(self._lib.cl_contig_collect(self._h, C.byref(s), C.byref(iv), C.byref(n)))
        return self._result(s, iv, n)

    def contig_finish(self):
        s = _lib.cl_contig_summary(); iv = C.POINTER(_lib.cl_interval)(); n = C.c_size_t()
        self._check(self._lib.cl_contig_finish(self._h, C.byref(s), C.byref(iv), C.byref(n)))
        return self._result(s, iv, n)

    def device_summary(self):
        p = C.c_void_p(); n = C.c_size_t()
        self._check(self._lib.cl_device_summary(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_profiling(self, on=True):
        self._check(self._lib.cl_set_profiling(self._h, 1 if on else 0))

    def kernel_ms(self):
        ms = (C.c_double * _lib.CL_K_COUNT)(); n = C.c_uint64()
        self._check(self._lib.cl_get_kernel_ms(self._h, ms, C.byref(n)))
        return dict(zip(_lib.CL_K_NAMES, list(ms))), n.value

    def reset_kernel_ms(self):
        self._check(self._lib.cl_reset_kernel_ms(self._h))

    def contig_bytes(self):
        a = C.c_uint64(); b = C.c_uint64()
        self._check(self._lib.cl_contig_bytes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def contig_layout(self):
        """What is resident for the uploaded contig (cl_layout_info as a dict)."""
        li = _lib.cl_layout_info()
        self._check(self._lib.cl_contig_layout(self._h, C.byref(li)))
        return {k: int(getattr(li, k)) for k, _ in li._fields_}

    def debug_depths(self, extent):
        raw = np.zeros(extent, np.uint32); qc = np.zeros(extent, np.uint32)
        low = np.zeros(extent, np.uint32); st = np.zeros(extent, np.uint8)
        self._check(self._lib.cl_debug_depths(self._h, _ptr(raw), _ptr(qc), _ptr(low), _ptr(st), extent))
        return raw, qc, low, st

    def site_pileup(self, min_quality, contig_len, ref_len, rec: ContigRecords, sites):
        sites = np.ascontiguousarray(sites, np.uint32)
        hist = np.zeros((sites.shape[0], 16), np.uint32)
        t = _lib.cl_site_tile()
        t.n_reads = rec.n
        t.pos = _ptr(rec.pos); t.mapq = _ptr(rec.mapq); t.cigar_off = _ptr(rec.cigar_off)
        t.cigar = _ptr(rec.cigar); t.seq_off = _ptr(rec.seq_off); t.seq4 = _ptr(rec.seq4)
        self._check(self._lib.cl_site_pileup(self._h, min_quality, contig_len, ref_len, C.byref(t),
                                             _ptr(sites), sites.shape[0], _ptr(hist)))
        return hist

    def _site_tile(self, rec: ContigRecords):
        t = _lib.cl_site_tile()
        t.n_reads = rec.n
        t.pos = _ptr(rec.pos); t.mapq = _ptr(rec.mapq); t.cigar_off = _ptr(rec.cigar_off)
        t.cigar = _ptr(rec.cigar); t.seq_off = _ptr(rec.seq_off); t.seq4 = _ptr(rec.seq4)
        return t

    def site_upload(self, contig_len, ref_len, rec: ContigRecords):
        """The tile goes to HBM once and stays resident for any number of site_run calls."""
        t = self._site_tile(rec)
        self._check(self._lib.cl_site_upload(self._h, contig_len, ref_len, C.byref(t)))

    def site_run(self, min_quality, sites):
        sites = np.ascontiguousarray(sites, np.uint32)
        hist = np.zeros((sites.shape[0], 16), np.uint32)
        self._check(self._lib.cl_site_run(self._h, min_quality, _ptr(sites), sites.shape[0], _ptr(hist)))
        return hist

    def site_pileup_stats(self):
        """(kernel milliseconds, algorithmic bytes) of the last site_pileup."""
        ms = C.c_double(); b = C.c_uint64()
        self._check(self._lib.cl_site_pileup_stats(self._h, C.byref(ms), C.byref(b)))
        return ms.value, b.value


class HostStage(Engine):
    """A context WITHOUT a device (cl_debug_host_create), for the CPU test suite: contig_begin / push_reads stage a
    contig exactly as a device context does, `pass_rows` runs the upload's row builder over it.  Everything that needs a
    device raises: there is no CPU pileup."""

    def __init__(self, options: CallableOptions):
        self._lib = _lib.load()
        self.options = options
        self._opt_c = options.to_c()
        h = C.c_void_p()
        st = self._lib.cl_debug_host_create(C.byref(self._opt_c), C.byref(h))
        if st != 0:
            raise EngineError(st, "cl_debug_host_create failed")
        self._h = h
        self.device_id = -1
        self._keep = []

    def pass_rows(self):
        """(n_groups per window, rows as uint32 array of 256-word groups window after window, summed_baseq)."""
        nwords = C.c_uint64(); nwin = C.c_uint32(); sq = C.c_uint64()
        self._check(self._lib.cl_debug_pass_rows(self._h, None, 0, None, 0, C.byref(nwords), C.byref(nwin), C.byref(sq)))
        ng = np.zeros(max(nwin.value, 1), np.uint32)
        rows = np.zeros(max(nwords.value, 1), np.uint32)
        self._check(self._lib.cl_debug_pass_rows(self._h, _ptr(ng), nwin.value, _ptr(rows), nwords.value, C.byref(nwords),
                                                 C.byref(nwin), C.byref(sq)))
        return ng[:nwin.value], rows[:nwords.value], int(sq.value)


@dataclass
class ContigResult:
    summary: "_lib.cl_contig_summary"
    intervals: np.ndarray       # (n,3) uint32: start, end (exclusive), state

    @property
    def state_counts(self):
        return [int(x) for x in self.summary.state_counts]

    def as_dict(self):
        s = self.summary
        return dict(state_counts=self.state_counts, n_covered_bases=int(s.n_covered_bases),
                    summed_coverage=int(s.summed_coverage), summed_baseq=int(s.summed_baseq),
                    summed_mapq=int(s.summed_mapq), quality_bases=int(s.quality_bases),
                    extent=int(s.extent), max_raw_depth=int(s.max_raw_depth),
                    n_intervals=int(s.n_intervals))


class CallableProfiler:
    """CallableProfiler (callable_profiler.rs): owns the BED file."""

    def __init__(self, bed_file: str, largest_contig_length: int = 0):
        self._lib = _lib.load()
        self._h = self._lib.dut_profiler_new(bed_file.encode())
        if not self._h:
            raise OSError(f"cannot create {bed_file}")
        self.largest_contig_length = largest_contig_length

    def enable_plots(self, largest_contig_length: Optional[int] = None):
        """From here on every BED line of a plotted state is a range of the current contig's coverage figure
        (callable_profiler.rs:48-59); `finish_plot` writes `<dir of the BED>/<contig>_coverage.svg`."""
        if largest_contig_length is not None:
            self.largest_contig_length = largest_contig_length
        self._lib.dut_profiler_enable_plots(self._h, self.largest_contig_length)

    def plot_bins(self, contig: str, contig_length: int):
        """(stride, callable, low_qual, ref_n): positions of the three plotted states per stride of the
        pending ranges (histogram_plotter.rs:74-101)."""
        n = C.c_size_t(); stride = C.c_uint32()
        st = self._lib.dut_profiler_plot_bins(self._h, contig.encode(), contig_length, C.byref(stride), None, None, None, 0, C.byref(n))
        if st != 0:
            raise EngineError(st, "dut_profiler_plot_bins failed")
        a = [np.zeros(n.value, np.uint32) for _ in range(3)]
        self._lib.dut_profiler_plot_bins(self._h, contig.encode(), contig_length, C.byref(stride), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]),
                                         n.value, C.byref(n))
        return int(stride.value), a[0], a[1], a[2]

    def finish_plot(self, contig: str, contig_length: int) -> bool:
        st = self._lib.dut_profiler_finish_plot(self._h, contig.encode(), contig_length)
        if st < 0:
            raise EngineError(st, "dut_profiler_finish_plot failed")
        return st == 1

    def get_contig_counts(self, contig: str):
        out = (C.c_uint64 * 6)()
        self._lib.dut_profiler_contig_counts(self._h, contig.encode(), out)
        return [int(x) for x in out]

    def feed_contig(self, contig: str, result: ContigResult):
        iv = np.ascontiguousarray(result.intervals, np.uint32)
        cnt = (C.c_uint64 * 6)(*result.state_counts)
        st = self._lib.dut_profiler_feed_contig(self._h, contig.encode(), _ptr(iv), iv.shape[0], cnt)
        if st != 0:
            raise EngineError(st, "dut_profiler_feed_contig failed")

    def close(self):
        """Drop: flush the writer."""
        if self._h:
            self._lib.dut_profiler_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class ContigProfiler:
    """ContigProfiler (contig_profiler.rs): per-contig accumulators + derived statistics."""
    name: str
    length: int
    n_covered_bases: int = 0
    summed_coverage: int = 0
    summed_baseq: int = 0
    summed_mapq: int = 0
    quality_bases: int = 0
    n_reads: int = 0

    def _c(self):
        return _lib.dut_contig_stats(self.length, self.n_covered_bases, self.summed_coverage,
                                     self.summed_baseq, self.summed_mapq, self.quality_bases,
                                     self.n_reads, 0)

    def _load(self, c):
        self.n_covered_bases = int(c.n_covered_bases); self.summed_coverage = int(c.summed_coverage)
        self.summed_baseq = int(c.summed_baseq); self.summed_mapq = int(c.summed_mapq)
        self.quality_bases = int(c.quality_bases); self.n_reads = int(c.n_reads)

    def derived(self):
        d = _lib.dut_contig_derived()
        c = self._c()
        _lib.load().dut_contig_derive(C.byref(c), C.byref(d))
        return dict(coverage_percent=d.coverage_percent, average_depth=d.average_depth,
                    average_mapq=d.average_mapq, average_baseq=d.average_baseq,
                    q30_percentage=d.q30_percentage)

    def get_coverage_stats(self):
        d = self.derived()
        return dict(unique_reads=self.n_reads, coverage_percent=d["coverage_percent"],
                    average_depth=d["average_depth"], covered_bases=self.n_covered_bases,
                    total_bases=self.length)

    def get_quality_stats(self):
        d = self.derived()
        return dict(average_mapq=d["average_mapq"], average_baseq=d["average_baseq"],
                    q30_percentage=d["q30_percentage"])


def admit_reads(options: CallableOptions, tid: int, contig_len: int, rec: ContigRecords):
    """FUNMAP drop + maxcnt rule + region filter; returns (accepted mask, n distinct names)."""
    lib = _lib.load()
    acc = np.zeros(max(rec.n, 1), np.uint8)
    nn = C.c_uint32(); na = C.c_uint64()
    oc = options.to_c(); rc = _records_c(rec)
    st = lib.dut_admit_reads(C.byref(oc), tid, contig_len, C.byref(rc), _ptr(acc), C.byref(nn), C.byref(na))
    if st != 0:
        raise EngineError(st, "reads are not coordinate sorted" if st == -3 else "dut_admit_reads failed")
    return acc[:rec.n].astype(bool), nn.value


def process_single_contig(engine: Engine, counter: CallableProfiler, stats: ContigProfiler,
                          options: CallableOptions, tid: int, rec: ContigRecords,
                          ref: Optional[np.ndarray]):
    """process_single_contig (mod.rs:44-147): `bam`/`fasta`/`header` of the reference become the
    decoded records of the contig, its FASTA bytes and (stats.name, stats.length)."""
    lib = _lib.load()
    ref = np.ascontiguousarray(ref, np.uint8) if ref is not None else np.zeros(0, np.uint8)
    oc = options.to_c(); rc = _records_c(rec); cs = stats._c()
    st = lib.dut_process_single_contig(engine._h, counter._h, C.byref(cs), C.byref(oc), stats.name.encode(),
                                       tid, stats.length, _ptr(ref), ref.shape[0], C.byref(rc))
    if st != 0:
        msg = lib.cl_last_error(engine._h).decode() or "admission failed"
        if st == -3:
            msg = msg or "reads are not coordinate sorted"
        raise EngineError(st, msg)
    stats._load(cs)


def compare_contig_names(a: str, b: str) -> int:
    return _lib.load().dut_compare_contig_names(a.encode(), b.encode())


def genome_summary(stats: List[ContigProfiler], callable_counts: List[int]):
    """report.rs:26-126 over contigs sorted with compare_contig_names (report.rs:37-38)."""
    import functools
    order = sorted(range(len(stats)), key=functools.cmp_to_key(
        lambda i, j: compare_contig_names(stats[i].name, stats[j].name)))
    n = len(order)
    arr = (_lib.dut_contig_stats * max(n, 1))()
    call = (C.c_uint64 * max(n, 1))()
    for k, i in enumerate(order):
        arr[k] = stats[i]._c()
        call[k] = callable_counts[i]
    out = _lib.dut_genome_summary()
    _lib.load().dut_genome_summary_build(arr, call, n, C.byref(out))
    return dict(total_bases=int(out.total_bases), callable_bases=int(out.callable_bases),
                callable_percentage=out.callable_percentage, average_depth=out.average_depth,
                average_mapq=out.average_mapq, average_baseq=out.average_baseq,
                q30_percentage=out.q30_percentage, total_unique_reads=int(out.total_unique_reads),
                contigs_analyzed=int(out.contigs_analyzed),
                order=[stats[i].name for i in order])

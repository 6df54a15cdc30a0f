"""TEST INFRASTRUCTURE ONLY: ctypes wrapper of oracle/liboracle.so (the CPU restatement of the
reference's coverage path).  Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package never imports it."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_LIB: a differently built oracle (the sanitizer build of tests/test_sanitizers.py)
LIB = os.environ.get("ORACLE_LIB") or os.path.join(HERE, "liboracle.so")
STATE_NAMES = ["REF_N", "CALLABLE", "NO_COVERAGE", "LOW_COVERAGE", "EXCESSIVE_COVERAGE",
               "POOR_MAPPING_QUALITY"]


def build(force=False):
    src = [os.path.join(HERE, "callable_oracle.c"), os.path.join(HERE, "callable_oracle.h")]
    if os.environ.get("ORACLE_LIB"):
        return LIB
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
        tmp = f"{LIB}.tmp{os.getpid()}"                 # several processes may find the library stale at once
        subprocess.run(["gcc", "-O2", "-g", "-fPIC", "-std=c11", "-shared", "-o", tmp, src[0]], check=True)
        os.replace(tmp, LIB)
    return LIB


class orc_options(C.Structure):
    _fields_ = [("min_depth", C.c_uint32), ("max_depth", C.c_uint32),
                ("min_mapping_quality", C.c_uint8), ("min_base_quality", C.c_uint8),
                ("min_depth_for_low_mapq", C.c_uint32), ("max_low_mapq", C.c_uint8),
                ("max_low_mapq_fraction", C.c_double)]


class orc_reads(C.Structure):
    _fields_ = [("n", C.c_int64), ("pos", C.c_void_p), ("flag", C.c_void_p), ("mapq", C.c_void_p),
                ("cigar_off", C.c_void_p), ("cigar", C.c_void_p), ("qual_off", C.c_void_p),
                ("qual", C.c_void_p), ("qname_off", C.c_void_p), ("qname", C.c_void_p)]


class orc_contig_stats(C.Structure):
    _fields_ = [("length", C.c_uint64), ("n_covered_bases", C.c_uint64), ("summed_coverage", C.c_uint64),
                ("summed_baseq", C.c_uint64), ("summed_mapq", C.c_uint64), ("quality_bases", C.c_uint64),
                ("n_reads", C.c_uint32), ("n_selected_reads", C.c_uint32)]


class orc_contig_derived(C.Structure):
    _fields_ = [("coverage_percent", C.c_double), ("average_depth", C.c_double), ("average_mapq", C.c_double),
                ("average_baseq", C.c_double), ("q30_percentage", C.c_double)]


class orc_genome_summary(C.Structure):
    _fields_ = [("total_bases", C.c_uint64), ("callable_bases", C.c_uint64),
                ("callable_percentage", C.c_double), ("average_depth", C.c_double),
                ("average_mapq", C.c_double), ("average_baseq", C.c_double), ("q30_percentage", C.c_double),
                ("total_unique_reads", C.c_uint64), ("contigs_analyzed", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.orc_profiler_new.restype = C.c_void_p
        L.orc_profiler_new.argtypes = [C.c_char_p]
        L.orc_profiler_free.argtypes = [C.c_void_p]
        L.orc_profiler_contig_counts.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64)]
        L.orc_process_single_contig.restype = C.c_int
        L.orc_process_single_contig.argtypes = [
            C.c_void_p, C.POINTER(orc_contig_stats), C.POINTER(orc_options), C.c_char_p, C.c_int32,
            C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(orc_reads), C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        L.orc_contig_derive.argtypes = [C.POINTER(orc_contig_stats), C.POINTER(orc_contig_derived)]
        L.orc_compare_contig_names.restype = C.c_int
        L.orc_compare_contig_names.argtypes = [C.c_char_p, C.c_char_p]
        L.orc_genome_summary_build.argtypes = [C.POINTER(orc_contig_stats), C.POINTER(C.c_uint64), C.c_size_t,
                                               C.POINTER(orc_genome_summary)]
        L.orc_accepted_reads.restype = C.c_int
        L.orc_accepted_reads.argtypes = [C.POINTER(orc_options), C.c_int32, C.c_uint32, C.POINTER(orc_reads),
                                         C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_site_pileup.restype = C.c_int
        L.orc_site_pileup.argtypes = [C.c_uint32, C.c_uint8, C.c_uint32, C.c_void_p, C.c_uint64,
                                      C.POINTER(orc_reads), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def options_c(o):
    """o: any object with the CallableOptions field names (options.rs:2-9)."""
    return orc_options(o.min_depth, o.max_depth, o.min_mapping_quality, o.min_base_quality,
                       o.min_depth_for_low_mapq, o.max_low_mapq, float(o.max_low_mapq_fraction))


def reads_c(rec):
    r = orc_reads()
    r.n = rec.n
    r.pos, r.flag, r.mapq = _p(rec.pos), _p(rec.flag), _p(rec.mapq)
    r.cigar_off, r.cigar = _p(rec.cigar_off), _p(rec.cigar)
    r.qual_off, r.qual = _p(rec.qual_off), _p(rec.qual)
    r.qname_off, r.qname = _p(rec.qname_off), _p(rec.qname)
    return r


class OracleError(RuntimeError):
    pass


class Profiler:
    """CallableProfiler of the oracle (owns the BED file)."""

    def __init__(self, bed_path):
        self.h = lib().orc_profiler_new(str(bed_path).encode())
        if not self.h:
            raise OSError(f"cannot create {bed_path}")

    def contig_counts(self, name):
        out = (C.c_uint64 * 6)()
        lib().orc_profiler_contig_counts(self.h, name.encode(), out)
        return [int(x) for x in out]

    def close(self):
        if self.h:
            lib().orc_profiler_free(self.h)
            self.h = None


def process_single_contig(prof, options, name, tid, contig_len, ref, rec, dump=False):
    """Returns (stats dict, dumps or None). dumps = (raw, qc, low, state, extent)."""
    L = lib()
    st = orc_contig_stats()
    st.length = contig_len
    oc = options_c(options)
    rc = reads_c(rec)
    ref = np.ascontiguousarray(ref, np.uint8) if ref is not None else np.zeros(0, np.uint8)
    err = C.create_string_buffer(256)
    ext = C.c_uint64(0)
    if dump:
        span = 0
        if rec.n:
            # upper bound of positions visited: contig_len or the furthest read end
            ops = rec.cigar & 15
            lens = (rec.cigar >> 4).astype(np.int64)
            refadv = np.isin(ops, [0, 2, 3, 7, 8])
            cs = np.concatenate([[0], np.cumsum(np.where(refadv, lens, 0))])
            rl = cs[rec.cigar_off[1:].astype(np.int64)] - cs[rec.cigar_off[:-1].astype(np.int64)]
            span = int((rec.pos.astype(np.int64) + rl).max())
        cap = max(contig_len, span) + 1
        raw = np.zeros(cap, np.uint32); qc = np.zeros(cap, np.uint32); low = np.zeros(cap, np.uint32)
        sta = np.full(cap, 255, np.uint8)
    else:
        cap = 0
        raw = qc = low = sta = None
    code = L.orc_process_single_contig(prof.h, C.byref(st), C.byref(oc), name.encode(), tid, contig_len,
                                       _p(ref), ref.shape[0], C.byref(rc), _p(raw), _p(qc), _p(low), _p(sta),
                                       cap, C.byref(ext), err, 256)
    if code != 0:
        raise OracleError(f"{err.value.decode()} ({code})")
    stats = dict(length=int(st.length), n_covered_bases=int(st.n_covered_bases),
                 summed_coverage=int(st.summed_coverage), summed_baseq=int(st.summed_baseq),
                 summed_mapq=int(st.summed_mapq), quality_bases=int(st.quality_bases),
                 n_reads=int(st.n_reads), n_selected_reads=int(st.n_selected_reads))
    d = orc_contig_derived()
    L.orc_contig_derive(C.byref(st), C.byref(d))
    stats["derived"] = dict(coverage_percent=d.coverage_percent, average_depth=d.average_depth,
                            average_mapq=d.average_mapq, average_baseq=d.average_baseq,
                            q30_percentage=d.q30_percentage)
    dumps = None
    if dump:
        e = int(ext.value)
        dumps = (raw[:e], qc[:e], low[:e], sta[:e], e)
    return stats, dumps


def accepted_reads(options, tid, contig_len, rec):
    acc = np.zeros(max(rec.n, 1), np.uint8)
    oc = options_c(options); rc = reads_c(rec)
    err = C.create_string_buffer(256)
    code = lib().orc_accepted_reads(C.byref(oc), tid, contig_len, C.byref(rc), _p(acc), err, 256)
    if code != 0:
        raise OracleError(f"{err.value.decode()} ({code})")
    return acc[:rec.n].astype(bool)


def compare_contig_names(a, b):
    return lib().orc_compare_contig_names(a.encode(), b.encode())


def genome_summary(stats_list, callable_list):
    """stats_list: dicts as returned by process_single_contig, ALREADY in report order."""
    n = len(stats_list)
    arr = (orc_contig_stats * max(n, 1))()
    call = (C.c_uint64 * max(n, 1))()
    for i, s in enumerate(stats_list):
        arr[i] = orc_contig_stats(s["length"], s["n_covered_bases"], s["summed_coverage"], s["summed_baseq"],
                                  s["summed_mapq"], s["quality_bases"], s["n_reads"], 0)
        call[i] = callable_list[i]
    out = orc_genome_summary()
    lib().orc_genome_summary_build(arr, call, n, C.byref(out))
    return {f: getattr(out, f) for f, _ in orc_genome_summary._fields_}


def site_pileup(min_depth, min_quality, contig_len, ref, rec, sites):
    sites = np.ascontiguousarray(sites, np.uint32)
    n = sites.shape[0]
    ref = np.ascontiguousarray(ref, np.uint8) if ref is not None else np.zeros(0, np.uint8)
    total = np.zeros(n, np.uint32); base = np.zeros(n, np.uint8); count = np.zeros(n, np.uint32)
    called = np.zeros(n, np.uint8); freq = np.zeros(n, np.float64); hist = np.zeros((n, 16), np.uint32)
    rc = reads_c(rec)
    lib().orc_site_pileup(min_depth, min_quality, contig_len, _p(ref), ref.shape[0], C.byref(rc),
                          _p(rec.seq_off), _p(rec.seq4), _p(sites), n, _p(total), _p(base), _p(count),
                          _p(called), _p(freq), _p(hist))
    return dict(total=total, base=base, count=count, called=called.astype(bool), freq=freq, hist=hist)

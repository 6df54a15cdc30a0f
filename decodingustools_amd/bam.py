"""BAM / FASTA input (include/dut_bam.h) and the file-level `coverage` entry point: what the reference
does with rust-htslib before and around the hot path (utils/bam_reader.rs:7-14, mod.rs:53-55,
api/coverage.rs:53-115)."""
import ctypes as C

import numpy as np

from . import _lib
from .callable_loci import CallableOptions, EngineError
from .records import ContigRecords


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class BamReader:
    """IndexedReader stand-in: header + per-contig fetch of decoded records."""

    def __init__(self, path: str):
        self._lib = _lib.load()
        err = C.create_string_buffer(512)
        self._h = self._lib.dut_bam_open(path.encode(), err, 512)
        if not self._h:
            raise OSError(err.value.decode())
        n = self._lib.dut_bam_n_ref(self._h)
        self.target_names = [self._lib.dut_bam_ref_name(self._h, t).decode() for t in range(n)]
        self.target_lens = [int(self._lib.dut_bam_ref_len(self._h, t)) for t in range(n)]
        ln = C.c_size_t()
        p = self._lib.dut_bam_header_text(self._h, C.byref(ln))
        self.header_text = C.string_at(p, ln.value).decode(errors="replace") if ln.value else ""
        self.has_index = bool(self._lib.dut_bam_has_index(self._h))
        # mapped reads per reference from the .bai metadata (-1 = not recorded): the balancing weight
        self.target_mapped = [int(self._lib.dut_bam_ref_mapped(self._h, t)) for t in range(n)]

    def fetch_contig(self, tid: int, with_seq: bool = False) -> ContigRecords:
        r = _lib.dut_records()
        so, sq = C.c_void_p(), C.c_void_p()
        st = self._lib.dut_bam_read_contig(self._h, tid, C.byref(r), C.byref(so) if with_seq else None,
                                           C.byref(sq) if with_seq else None)
        if st != 0:
            raise EngineError(st, self._lib.dut_bam_error(self._h).decode())
        n = int(r.n)
        coff = _arr(r.cigar_off, n + 1, np.uint32); qoff = _arr(r.qual_off, n + 1, np.uint64)
        noff = _arr(r.qname_off, n + 1, np.uint32)
        rec = ContigRecords(
            pos=_arr(r.pos, n, np.int32), flag=_arr(r.flag, n, np.uint16), mapq=_arr(r.mapq, n, np.uint8),
            cigar_off=coff, cigar=_arr(r.cigar, int(coff[-1]), np.uint32), qual_off=qoff,
            qual=_arr(r.qual, int(qoff[-1]), np.uint8), qname_off=noff, qname=_arr(r.qname, int(noff[-1]), np.uint8))
        if with_seq:
            rec.seq_off = _arr(so.value, n + 1, np.uint64)
            rec.seq4 = _arr(sq.value, (int(rec.seq_off[-1]) + 1) // 2, np.uint8)
        return rec.validate()

    def fetch_contig_bits(self, tid: int, min_base_quality: int):
        """dut_bam_read_contig_bits: the records with the base-quality test taken at parse -> (ContigRecords without
        quality bytes, pass_bits uint64, pass_sum uint32)."""
        r = _lib.dut_records()
        st = self._lib.dut_bam_read_contig_bits(self._h, tid, min_base_quality, C.byref(r))
        if st != 0:
            raise EngineError(st, self._lib.dut_bam_error(self._h).decode())
        n = int(r.n)
        coff = _arr(r.cigar_off, n + 1, np.uint32); qoff = _arr(r.qual_off, n + 1, np.uint64)
        noff = _arr(r.qname_off, n + 1, np.uint32)
        rec = ContigRecords(
            pos=_arr(r.pos, n, np.int32), flag=_arr(r.flag, n, np.uint16), mapq=_arr(r.mapq, n, np.uint8),
            cigar_off=coff, cigar=_arr(r.cigar, int(coff[-1]), np.uint32), qual_off=qoff,
            qual=np.zeros(0, np.uint8), qname_off=noff, qname=_arr(r.qname, int(noff[-1]), np.uint8))
        bits = _arr(r.pass_bits, (int(qoff[-1]) + 63) // 64 + 1, np.uint64)
        return rec, bits, _arr(r.pass_sum, n, np.uint32)

    def close(self):
        if self._h:
            self._lib.dut_bam_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class FastaReader:
    def __init__(self, path: str):
        self._lib = _lib.load()
        err = C.create_string_buffer(512)
        self._h = self._lib.dut_fasta_open(path.encode(), err, 512)
        if not self._h:
            raise OSError(err.value.decode())

    def fetch(self, name: str) -> np.ndarray:
        p, n = C.c_void_p(), C.c_uint64()
        st = self._lib.dut_fasta_fetch(self._h, name.encode(), C.byref(p), C.byref(n))
        if st != 0:                                    # fetch_seq(..)? of the reference (mod.rs:79)
            raise OSError(self._lib.dut_fasta_error(self._h).decode())
        return _arr(p.value, int(n.value), np.uint8)

    def close(self):
        if self._h:
            self._lib.dut_fasta_close(self._h)
            self._h = None


def coverage_files(bam_file: str, reference_file: str, output_bed: str = "callable_regions.bed",
                   summary_json: str = None, options: CallableOptions = None, contigs=None, device_id: int = 0,
                   output_summary: str = None, devices=None):
    """CoverageAnalyzer::analyze on files (CoverageInput of api/coverage.rs:124-132); summary_json
    receives the CoverageOutput JSON of main.rs:68-69; output_summary, when given, the HTML report
    (api/coverage.rs:104; None: not written, the JSON then names "summary.html").  The per-contig coverage
    figures `<contig>_coverage.svg` go beside the BED file (callable_profiler.rs:80-84).
    devices: a list of HIP ordinals (one may repeat) -- the contigs are dealt to them inside the library
    (dut_coverage_files_multi: one host thread and one engine context per entry, no torch, no collective)."""
    lib = _lib.load()
    options = options or CallableOptions()
    oc = options.to_c()
    arr = None
    n = 0
    if contigs is not None:
        n = len(contigs)
        arr = (C.c_char_p * max(n, 1))(*[c.encode() for c in contigs])
    err = C.create_string_buffer(1024)
    if devices is not None:
        dv = (C.c_int * len(devices))(*[int(d) for d in devices])
        st = lib.dut_coverage_files_multi(bam_file.encode(), reference_file.encode(), output_bed.encode(),
                                          summary_json.encode() if summary_json else None, output_summary.encode() if output_summary else None,
                                          C.byref(oc), arr, n, dv, len(devices), 0, err, 1024)
    else:
        st = lib.dut_coverage_files(bam_file.encode(), reference_file.encode(), output_bed.encode(),
                                    summary_json.encode() if summary_json else None, output_summary.encode() if output_summary else None,
                                    C.byref(oc), arr, n, device_id, err, 1024)
    if st != 0:
        raise EngineError(st, err.value.decode())

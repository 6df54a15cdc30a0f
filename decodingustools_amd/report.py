"""Host-side mirror of the reference's BamStats / PlatformInference / summary.json writer
(include/dut_report.h; bam_stats.rs, platform_inference.rs, report.rs:15-134, main.rs:68-69)."""
import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

from . import _lib
from .callable_loci import EngineError

PLATFORMS = ("Illumina", "PacBio", "Nanopore", "MGI", "Unknown")      # SequencingPlatform, declaration order


def detect_aligner(header_text: bytes) -> str:
    return _lib.load().dut_detect_aligner(header_text, len(header_text)).decode()


def reference_build(header_text: bytes) -> str:
    return _lib.load().dut_reference_build(header_text, len(header_text)).decode()


def detect_platform_from_qname(qname: bytes) -> str:
    return PLATFORMS[_lib.load().dut_detect_platform_from_qname(qname, len(qname))]


def parse_read_name(platform: str, qname: bytes) -> Optional[Tuple[bytes, Optional[bytes]]]:
    """parse_{illumina,pacbio,nanopore,mgi}_read_name: (instrument, flow cell or None) or None."""
    lib = _lib.load()
    buf = C.create_string_buffer(qname, len(qname))
    ip, il, fp, fl = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
    if not lib.dut_parse_read_name(PLATFORMS.index(platform), buf, len(qname), C.byref(ip), C.byref(il),
                                   C.byref(fp), C.byref(fl)):
        return None
    ins = C.string_at(ip.value, il.value) if il.value else b""
    fc = (C.string_at(fp.value, fl.value) if fl.value else b"") if fp.value else None
    return ins, fc


def infer_specific_platform(platform: str, top_instrument: Optional[str]) -> str:
    t = top_instrument.encode() if top_instrument is not None else None
    return _lib.load().dut_infer_specific_platform(PLATFORMS.index(platform), t).decode()


def format_f64(v: float) -> str:
    b = C.create_string_buffer(48)
    n = _lib.load().dut_format_f64(v, b)
    return b.raw[:n].decode()


class BamStats:
    """BamStats (bam_stats.rs:9-280): header-derived strings and a sample of the first records."""

    def __init__(self, max_samples: int = 10000):
        self._lib = _lib.load()
        self._h = self._lib.dut_bam_stats_new(max_samples)
        self._i = 0

    def close(self):
        if self._h:
            self._lib.dut_bam_stats_free(self._h)
            self._h = None

    __del__ = close

    def set_header(self, text: bytes):
        self._lib.dut_bam_stats_set_header(self._h, text, len(text))

    def add(self, flag: int, l_seq: int, qname: bytes, tlen: int) -> bool:
        more = self._lib.dut_bam_stats_add(self._h, self._i, flag, l_seq, qname, len(qname), tlen)
        self._i += 1
        return bool(more)

    def collect_stats(self, bam_path: str):
        err = C.create_string_buffer(512)
        st = self._lib.dut_bam_stats_collect(self._h, bam_path.encode(), err, 512)
        if st != 0:
            raise EngineError(st, "Failed to collect BAM stats: " + err.value.decode())
        return self

    def aligner(self) -> str:
        return self._lib.dut_bam_stats_aligner(self._h).decode()

    def reference_build(self) -> str:
        return self._lib.dut_bam_stats_reference_build(self._h).decode()

    def infer_platform(self) -> str:
        return self._lib.dut_bam_stats_infer_platform(self._h).decode()

    def get_primary_platform(self) -> str:
        return PLATFORMS[self._lib.dut_bam_stats_primary_platform(self._h)]

    def read_count(self) -> int:
        return self._lib.dut_bam_stats_read_count(self._h)

    def average_read_length(self) -> int:
        return self._lib.dut_bam_stats_average_read_length(self._h)

    def modal_read_length(self) -> int:
        return self._lib.dut_bam_stats_modal_read_length(self._h)

    def get_stats(self) -> Dict[str, float]:
        out = {}
        v = C.c_double()
        for k in ("average_read_length", "paired_percentage", "average_insert_size", "proper_pair_percentage"):
            if self._lib.dut_bam_stats_get(self._h, k.encode(), C.byref(v)):
                out[k] = v.value
        return out


def coverage_output_json(stats: Sequence, names: Sequence[str], state_counts: Sequence[Sequence[int]],
                         aligner: str, reference_build: str, sequencing_platform: str, read_length: int,
                         bed_file: str, summary_html: str, coverage_plots: Sequence[str] = ()) -> str:
    """The CoverageOutput JSON text (serde_json::to_writer_pretty, main.rs:68-69).  stats: objects
    with the ContigProfiler fields (length, n_covered_bases, summed_*, quality_bases, n_reads)."""
    lib = _lib.load()
    n = len(stats)
    cs = (_lib.dut_contig_stats * max(n, 1))()
    for i, s in enumerate(stats):
        cs[i].length = s.length; cs[i].n_covered_bases = s.n_covered_bases
        cs[i].summed_coverage = s.summed_coverage; cs[i].summed_baseq = s.summed_baseq
        cs[i].summed_mapq = s.summed_mapq; cs[i].quality_bases = s.quality_bases
        cs[i].n_reads = s.n_reads
    nm = (C.c_char_p * max(n, 1))(*[x.encode() for x in names])
    cnt = (C.c_uint64 * max(6 * n, 1))(*[int(v) for row in state_counts for v in row])
    plots = (C.c_char_p * max(len(coverage_plots), 1))(*[p.encode() for p in coverage_plots])
    meta = _lib.dut_export_meta(aligner.encode(), reference_build.encode(), sequencing_platform.encode(),
                                read_length, bed_file.encode(), summary_html.encode(), plots, len(coverage_plots))
    out, ln = C.c_void_p(), C.c_size_t()
    st = lib.dut_coverage_output_json(cs, nm, cnt, n, C.byref(meta), C.byref(out), C.byref(ln))
    if st != 0:
        raise EngineError(st, "cannot build the summary")
    try:
        return C.string_at(out.value, ln.value).decode()
    finally:
        lib.dut_free(out)


def write_html_report(path: str, stats: Sequence, names: Sequence[str], state_counts: Sequence[Sequence[int]],
                      aligner: str, reference_build: str, sequencing_platform: str, read_length: int,
                      max_samples: int = 10000) -> None:
    """summary.html (report.rs:136-340): the reference's sections, rows and number formats in this project's own
    markup; a contig's figure is embedded when `<name>_coverage.svg` exists in the working directory."""
    lib = _lib.load()
    n = len(stats)
    cs = (_lib.dut_contig_stats * max(n, 1))()
    for i, s in enumerate(stats):
        cs[i].length = s.length; cs[i].n_covered_bases = s.n_covered_bases
        cs[i].summed_coverage = s.summed_coverage; cs[i].summed_baseq = s.summed_baseq
        cs[i].summed_mapq = s.summed_mapq; cs[i].quality_bases = s.quality_bases
        cs[i].n_reads = s.n_reads
    nm = (C.c_char_p * max(n, 1))(*[x.encode() for x in names])
    cnt = (C.c_uint64 * max(6 * n, 1))(*[int(v) for row in state_counts for v in row])
    plots = (C.c_char_p * 1)()
    meta = _lib.dut_export_meta(aligner.encode(), reference_build.encode(), sequencing_platform.encode(),
                                read_length, b"", b"", plots, 0)
    st = lib.dut_write_html_report(cs, nm, cnt, n, C.byref(meta), max_samples, path.encode())
    if st != 0:
        raise EngineError(st, f"cannot write {path}")

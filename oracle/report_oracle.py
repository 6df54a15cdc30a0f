"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's BamStats / platform inference /
summary.json writer, used by tests/ as the checker for include/dut_report.h.  Nothing in the
product imports this module.

Parity unpinned: the reference has no tests or fixtures for these functions and cannot be built
here (no Rust toolchain); each function below follows the cited lines of the reference source.

  detect_platform_from_qname, parse_*_read_name, infer_specific_platform
                          src/callable_loci/profilers/platform_inference.rs:17-293
  BamStats                src/callable_loci/profilers/bam_stats.rs:9-280
  detect_aligner          src/callable_loci/mod.rs:149-177
  reference_build         src/types.rs:105-156 (ReferenceGenome::from_header + name)
  contig_derived          src/callable_loci/profilers/contig_profiler.rs:93-157
  build_coverage_export   src/callable_loci/report.rs:15-134
  coverage_output_json    src/api/coverage.rs:134-145 + src/export/formats/coverage.rs:26-248 as
                          serde_json::to_writer_pretty prints them (src/main.rs:68-69)

Where the reference takes `max_by_key` over a HashMap (modal length, primary platform, top
instrument), ties are resolved by hash iteration order, i.e. arbitrarily; the *_candidates
functions return every answer the reference can give.
"""
import decimal
import functools
import math
from collections import OrderedDict

ILLUMINA, PACBIO, NANOPORE, MGI, UNKNOWN = "Illumina", "PacBio", "Nanopore", "MGI", "Unknown"


def _is_hex(s):
    return all(c in "0123456789abcdefABCDEF" for c in s)


def detect_platform_from_qname(q: str) -> str:
    if len(q.encode()) > 30 and ("-" in q or "_" in q):
        parts = q.split("-")
        if len(parts) == 5:
            is_uuid = (len(parts[0]) == 8 and len(parts[1]) == 4 and len(parts[2]) == 4
                       and len(parts[3]) == 4 and len(parts[4]) >= 12)
            if is_uuid and all(_is_hex(p) for p in parts):
                return NANOPORE
        if "ch" in q and "read" in q:
            return NANOPORE
    if q.startswith("m") and "/" in q:
        parts = q.split("/")
        if len(parts) >= 2 and "_" in parts[0]:
            return PACBIO
    if len(q.encode()) > 15:
        prefix = q[:5].upper()
        if (prefix.startswith("V300") or prefix.startswith("E100") or prefix.startswith("CL100")
                or prefix.startswith("G400") or prefix.startswith("G99")):
            return MGI
        if q.count(":") >= 6:
            parts = q.split(":")
            if parts[0].startswith("V") or parts[0].startswith("E") or parts[0].startswith("CL") or parts[0].startswith("G"):
                if len(parts) >= 3 and parts[2].startswith("L"):
                    return MGI
    if q.count(":") >= 6:
        return ILLUMINA
    return UNKNOWN


def parse_illumina_read_name(q):
    parts = q.split(":")
    return (parts[0], parts[2]) if len(parts) >= 3 else None


def parse_pacbio_read_name(q):
    i = q.find("/")
    if i >= 0:
        movie = q[:i]
        if movie.startswith("m"):
            u = movie.find("_")
            if u >= 0:
                return movie[:u]
    return None


def parse_nanopore_read_name(q):
    if len(q.encode()) > 30 and "-" in q:
        if len(q.split("-")) >= 5:
            return q.split("_")[0].split("-")[0]
    u = q.find("_")
    if u >= 0:
        return q[:u]
    return "nanopore"


def parse_mgi_read_name(q):
    if q.count(":") >= 3:
        parts = q.split(":")
        if len(parts) >= 3:
            return parts[0], parts[1]
    if len(q.encode()) > 10:
        l_pos = q.find("L")
        if l_pos >= 0:
            rest = q[l_pos:]
            if rest.find("C") >= 0:
                r = rest.find("R")
                end_pos = r if r >= 0 else len(rest)
                return q[:l_pos], rest[:end_pos]
    return None


def infer_specific_platform(platform, top_instrument):
    """top_instrument: the most frequent instrument id, or None when the map is empty."""
    t = top_instrument
    if platform == PACBIO:
        if t is not None:
            if t.startswith("m84"): return "PacBio Revio"
            if t.startswith("m64"): return "PacBio Sequel II/IIe"
            if t.startswith("m54"): return "PacBio Sequel"
            return "PacBio"
        return "PacBio"
    if platform == NANOPORE:
        return "Oxford Nanopore"
    if platform == MGI:
        if t is not None:
            for pre, name in (("V300", "MGI DNBSEQ/MGISEQ-2000"), ("E100", "MGI MGISEQ-200"), ("CL100", "MGI MGISEQ-T7"),
                              ("G400", "MGI DNBSEQ-G400"), ("G99", "MGI MGISEQ-T1")):
                if t.startswith(pre):
                    return name
            return "MGI DNBseq"
        return "MGI DNBseq"
    if platform == ILLUMINA:
        if t is not None:
            c = t[0] if t else " "
            return {"a": "NovaSeq", "d": "HiSeq 2500", "j": "HiSeq 3000", "k": "HiSeq 4000", "e": "HiSeq X",
                    "n": "NextSeq", "m": "MiSeq", "v": "NovaSeq X", "f": "iSeq"}.get(c.lower() if c.isascii() else c, "Unknown Illumina")
        return "Unknown Illumina"
    return "Unknown"


def detect_aligner(header: str) -> str:
    h = "".join(chr(ord(c) + 32) if "A" <= c <= "Z" else c for c in header)
    for pat, name in (("@pg\tid:bwa-mem2", "BWA-MEM2"), ("@pg\tid:bwa", "BWA"), ("@pg\tid:minimap2", "minimap2"),
                      ("@pg\tid:pbmm2", "pbmm2"), ("@pg\tid:bowtie2", "Bowtie2"), ("@pg\tid:star", "STAR"),
                      ("bwa", "BWA"), ("minimap2", "minimap2"), ("bowtie2", "Bowtie2"), ("star", "STAR")):
        if pat in h:
            return name
    return "Unknown"


def reference_build(h: str) -> str:
    if "AS:GRCh38" in h or "GCA_000001405.15" in h: return "GRCh38"
    if "AS:GRCh37" in h or "GCA_000001405.1" in h: return "GRCh37"
    if "AS:CHM13" in h or "GCA_009914755.4" in h: return "T2T-CHM13v2.0"
    if "chm13" in h or "CHM13" in h or "t2t" in h or "T2T" in h: return "T2T-CHM13v2.0"
    if "SN:chr1" in h and "LN:248387328" in h:
        if "M5:e469247288ceb332aee524caec92bb22" in h: return "T2T-CHM13v2.0"
    if "SN:chr1" in h and "LN:248956422" in h: return "GRCh38"
    if "SN:1" in h and "LN:249250621" in h: return "GRCh37"
    return "Unknown"


def _argmax_candidates(d):
    if not d:
        return []
    m = max(d.values())
    return sorted(k for k, v in d.items() if v == m)


class BamStats:
    def __init__(self, max_samples):
        self.max_samples = max_samples
        self.read_count = self.total_read_length = self.paired_reads = self.paired_count = 0
        self.total_insert_size = 0
        self.length_distribution, self.insert_size_distribution = {}, {}
        self.flow_cells, self.instruments, self.platform_counts = {}, {}, {}
        self.aligner = self.reference_build = ""

    def set_header(self, text: str):
        self.aligner = detect_aligner(text)
        self.reference_build = reference_build(text)

    def collect(self, records):
        """records: iterable of (flag, l_seq, qname bytes, tlen) in file order."""
        for i, (flag, l_seq, qname, tlen) in enumerate(records):
            if i >= self.max_samples:
                break
            if not (flag & 0x100) and not (flag & 0x800):
                self.length_distribution[l_seq] = self.length_distribution.get(l_seq, 0) + 1
                self.read_count += 1
                self.total_read_length += l_seq
                try:
                    q = qname.decode("utf-8")
                except UnicodeDecodeError:
                    q = None
                if q is not None:
                    pf = detect_platform_from_qname(q)
                    self.platform_counts[pf] = self.platform_counts.get(pf, 0) + 1
                    ins = fc = None
                    if pf == ILLUMINA:
                        r = parse_illumina_read_name(q)
                        if r: ins, fc = r
                    elif pf == PACBIO:
                        ins = parse_pacbio_read_name(q)
                    elif pf == NANOPORE:
                        ins = parse_nanopore_read_name(q)
                    elif pf == MGI:
                        r = parse_mgi_read_name(q)
                        if r: ins, fc = r
                    if ins is not None:
                        self.instruments[ins] = self.instruments.get(ins, 0) + 1
                    if fc is not None:
                        self.flow_cells[fc] = self.flow_cells.get(fc, 0) + 1
                if flag & 0x1:
                    self.paired_reads += 1
                    if (flag & 0x2) and (flag & 0x40):
                        ins_size = abs(tlen)
                        if ins_size > 0:
                            self.insert_size_distribution[ins_size] = self.insert_size_distribution.get(ins_size, 0) + 1
                            self.total_insert_size += ins_size
                            self.paired_count += 1
        return self

    def average_read_length(self):
        return self.total_read_length // self.read_count if self.read_count > 0 else 0

    def modal_read_length_candidates(self):
        return _argmax_candidates(self.length_distribution) if self.read_count > 0 else [0]

    def primary_platform_candidates(self):
        return _argmax_candidates(self.platform_counts) or [UNKNOWN]

    def infer_platform_candidates(self):
        tops = _argmax_candidates(self.instruments) or [None]
        return sorted({infer_specific_platform(p, t) for p in self.primary_platform_candidates() for t in tops})

    def get_stats(self):
        """Values that do not depend on a tie; the two modal ones as candidate lists."""
        out = {}
        if self.read_count > 0:
            out["average_read_length"] = [float(x) for x in self.modal_read_length_candidates()]
            out["paired_percentage"] = (self.paired_reads / self.read_count) * 100.0
        if self.paired_count > 0:
            out["average_insert_size"] = [float(x) for x in _argmax_candidates(self.insert_size_distribution)]
            out["proper_pair_percentage"] = (self.paired_count * 2.0 / self.paired_reads) * 100.0
        return out


# ---- serde_json output -------------------------------------------------------------------------

def format_f64(v: float) -> str:
    """ryu's `pretty` formatting as serde_json uses it for f64: the shortest digits that round-trip
    (Python's repr gives the same digit string), then plain decimal notation when the decimal
    point falls within (-5, 16], else d.ddde[-]x."""
    if math.isnan(v) or math.isinf(v):
        return "null"
    sign = "-" if math.copysign(1.0, v) < 0 else ""
    v = abs(v)
    if v == 0.0:
        return sign + "0.0"
    _, dig, k = decimal.Decimal(repr(v)).as_tuple()
    digits = "".join(map(str, dig))
    stripped = digits.rstrip("0")
    k += len(digits) - len(stripped)
    digits = stripped
    n = len(digits)
    kk = n + k                                   # value = digits * 10^k; kk = position of the decimal point
    if 0 <= k and kk <= 16:
        return sign + digits + "0" * k + ".0"
    if 0 < kk <= 16:
        return sign + digits[:kk] + "." + digits[kk:]
    if -5 < kk <= 0:
        return sign + "0." + "0" * (-kk) + digits
    if n == 1:
        return sign + digits + "e" + str(kk - 1)
    return sign + digits[0] + "." + digits[1:] + "e" + str(kk - 1)


def _json_str(s: str) -> str:
    out = ['"']
    for ch in s:
        o = ord(ch)
        if ch == '"': out.append('\\"')
        elif ch == "\\": out.append("\\\\")
        elif ch == "\b": out.append("\\b")
        elif ch == "\f": out.append("\\f")
        elif ch == "\n": out.append("\\n")
        elif ch == "\r": out.append("\\r")
        elif ch == "\t": out.append("\\t")
        elif o < 0x20: out.append("\\u%04x" % o)
        else: out.append(ch)
    out.append('"')
    return "".join(out)


class F64(float):
    """marks a value that serde serialises as f64 (ints print bare, f64 always with a fraction)"""


def to_pretty(v, ind=0) -> str:
    pad = "  " * (ind + 1)
    if isinstance(v, F64):
        return format_f64(float(v))
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, int):
        return str(v)
    if isinstance(v, str):
        return _json_str(v)
    if isinstance(v, (list, tuple)):
        if not v:
            return "[]"
        return "[\n" + ",\n".join(pad + to_pretty(x, ind + 1) for x in v) + "\n" + "  " * ind + "]"
    if isinstance(v, dict):
        if not v:
            return "{}"
        return "{\n" + ",\n".join(pad + _json_str(k) + ": " + to_pretty(x, ind + 1) for k, x in v.items()) + "\n" + "  " * ind + "}"
    raise TypeError(type(v))


def contig_derived(s):
    """s: dict with length, n_covered_bases, summed_coverage, summed_baseq, summed_mapq, quality_bases."""
    average_depth = s["summed_coverage"] / s["n_covered_bases"] if s["n_covered_bases"] > 0 else 0.0
    coverage_percent = (s["n_covered_bases"] / s["length"]) * 100.0 if s["length"] > 0 else 0.0
    qb = s["quality_bases"]
    average_mapq = s["summed_mapq"] / qb if qb > 0 else 0.0
    average_baseq = s["summed_baseq"] / qb if qb > 0 else 0.0
    if qb > 0:
        if average_baseq >= 30.0: q30 = 100.0
        elif average_baseq < 20.0: q30 = 0.0
        else: q30 = ((average_baseq - 20.0) / 10.0) * 100.0
    else:
        q30 = 0.0
    return dict(coverage_percent=coverage_percent, average_depth=average_depth, average_mapq=average_mapq,
                average_baseq=average_baseq, q30_percentage=q30)


def coverage_output(stats, names, counts, aligner, reference_build_, platform, read_length, bed_file,
                    summary_html, coverage_plots=(), compare=None):
    """stats: list of dicts (ContigProfiler fields + n_reads); counts: list of 6 ints per contig.
    compare: contig-name comparator (default: the C oracle's restatement of report.rs:339-393)."""
    if compare is None:
        from . import compare_contig_names as compare
    cmpf = compare
    order = sorted(range(len(stats)), key=functools.cmp_to_key(lambda i, j: cmpf(names[i], names[j])))
    total_bases = callable_bases = q30_bases = total_quality_positions = total_unique_reads = 0
    total_depth = total_mapq = total_baseq = 0.0
    contigs = []
    for i in order:
        s, c = stats[i], counts[i]
        d = contig_derived(s)
        L = s["length"]
        total_bases += L
        callable_bases += c[1]
        total_depth += d["average_depth"] * float(L)
        total_mapq += d["average_mapq"] * float(L)
        total_baseq += d["average_baseq"] * float(L)
        q30_bases += int(d["q30_percentage"] / 100.0 * float(L))
        total_quality_positions += L
        total_unique_reads += s["n_reads"]
        contigs.append(OrderedDict([
            ("name", names[i]), ("length", L), ("unique_reads", s["n_reads"]),
            ("coverage_percent", F64(d["coverage_percent"])), ("average_depth", F64(d["average_depth"])),
            ("covered_bases", s["n_covered_bases"]), ("total_bases", L),
            ("quality_stats", OrderedDict([("average_mapq", F64(d["average_mapq"])), ("average_baseq", F64(d["average_baseq"])),
                                           ("q30_percentage", F64(d["q30_percentage"]))])),
            ("state_distribution", OrderedDict([("ref_n", c[0]), ("callable", c[1]), ("no_coverage", c[2]), ("low_coverage", c[3]),
                                                ("excessive_coverage", c[4]), ("poor_mapping_quality", c[5])]))]))
    average_depth = total_depth / float(total_bases) if total_bases > 0 else 0.0
    summary = OrderedDict([
        ("aligner", aligner), ("reference_build", reference_build_), ("sequencing_platform", platform),
        ("read_length", read_length), ("total_bases", total_bases), ("callable_bases", callable_bases),
        ("callable_percentage", F64((callable_bases / total_bases) * 100.0 if total_bases > 0 else 0.0)),
        ("average_depth", F64(average_depth)), ("contigs_analyzed", len(stats))])
    tq = total_quality_positions
    qm = OrderedDict([("average_mapq", F64(total_mapq / float(tq) if tq > 0 else 0.0)),
                      ("average_baseq", F64(total_baseq / float(tq) if tq > 0 else 0.0)),
                      ("q30_percentage", F64((q30_bases / tq) * 100.0 if tq > 0 else 0.0))])
    export = OrderedDict([("summary", summary), ("contigs", contigs), ("quality_metrics", qm),
                          ("total_unique_reads", total_unique_reads)])
    files = OrderedDict([("bed_file", bed_file), ("summary_html", summary_html), ("coverage_plots", list(coverage_plots))])
    return OrderedDict([("export", export), ("files", files)])


def coverage_output_json(*a, **k) -> str:
    return to_pretty(coverage_output(*a, **k))


# ---- the per-contig coverage figure's data (TEST INFRASTRUCTURE, like everything in oracle/) -------------
# callable_profiler.rs:39-84 (write_state pushes every written line of a plotted state; finish_contig writes the
# pending line WITHOUT clearing it, draws and empties the list -- so the previous contig's last line is written,
# and pushed, once more when the next contig starts) and utils/histogram_plotter.rs:74-101, 412-440 (positions
# per stride; stride = ceil(largest / 2000), "chrM": ceil(16569 / 200); n = length / stride + 1 bins).
PLOTTED = ("CALLABLE", "POOR_MAPPING_QUALITY", "REF_N")


def coverage_plot_bins(contigs, largest_contig_length):
    """contigs: [(name, length, [(start, end, state_name), ...])] in processing order, runs maximal and without
    the duplicated lines.  -> [(stride, [callable bins], [low-quality bins], [ref-N bins], n_ranges)] per contig;
    the reference draws a figure only when n_ranges > 0 (callable_profiler.rs:67)."""
    out = []
    cur = None
    ranges = []

    def write_state():
        if cur is not None and cur[2] in PLOTTED:
            ranges.append(cur)

    for name, length, runs in contigs:
        for run in runs:
            if cur is not None:
                write_state()                     # the state or the contig changed
            cur = run
        write_state()                             # finish_contig: the line stays pending
        stride = (16569 + 200 - 1) // 200 if name == "chrM" else (largest_contig_length + 2000 - 1) // 2000
        n = length // stride + 1
        bins = [[0] * n for _ in range(3)]
        for s, e, st in ranges:
            row = bins[PLOTTED.index(st)]
            for pos in range(s & 0xFFFFFFFF, e & 0xFFFFFFFF):
                idx = pos // stride
                if idx < n:
                    row[idx] += 1
        out.append((stride, bins[0], bins[1], bins[2], len(ranges)))
        ranges = []
    return out

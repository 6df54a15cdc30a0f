"""In-memory decoded alignment records of one contig, structure-of-arrays.

This is the hand-over format between a BAM reader and the coverage path: the fields of
SURVEY.md Appendix B (refID is implied by the contig).  Arrays are numpy, C-contiguous.
"""
import re
from dataclasses import dataclass, field

import numpy as np

CIGAR_OPS = "MIDNSHP=XB"
_CIG_RE = re.compile(r"(\d+)([MIDNSHP=XB])")
SEQ_CODES = "=ACMGRSVTWYHKDBN"


def cigar_from_string(s):
    out = []
    for n, op in _CIG_RE.findall(s):
        out.append((int(n) << 4) | CIGAR_OPS.index(op))
    if "".join(f"{n}{op}" for n, op in _CIG_RE.findall(s)) != s:
        raise ValueError(f"bad CIGAR string {s!r}")
    return out


def cigar_query_length(words):
    return sum(w >> 4 for w in words if (w & 15) in (0, 1, 4, 7, 8))


@dataclass
class ContigRecords:
    """All fetched records of one contig in file (coordinate) order."""
    pos: np.ndarray          # int32 [n]
    flag: np.ndarray         # uint16 [n]
    mapq: np.ndarray         # uint8 [n]
    cigar_off: np.ndarray    # uint32 [n+1]
    cigar: np.ndarray        # uint32
    qual_off: np.ndarray     # uint64 [n+1]
    qual: np.ndarray         # uint8
    qname_off: np.ndarray    # uint32 [n+1]
    qname: np.ndarray        # uint8
    seq_off: np.ndarray = field(default=None)   # uint64 [n+1], in bases (config 5 only)
    seq4: np.ndarray = field(default=None)      # uint8, 4-bit packed, high nibble first

    @property
    def n(self):
        return int(self.pos.shape[0])

    def validate(self):
        n = self.n
        assert self.pos.dtype == np.int32 and self.flag.dtype == np.uint16 and self.mapq.dtype == np.uint8
        assert self.cigar_off.dtype == np.uint32 and self.cigar.dtype == np.uint32
        assert self.qual_off.dtype == np.uint64 and self.qual.dtype == np.uint8
        assert self.qname_off.dtype == np.uint32 and self.qname.dtype == np.uint8
        assert self.flag.shape[0] == n and self.mapq.shape[0] == n
        assert self.cigar_off.shape[0] == n + 1 and self.qual_off.shape[0] == n + 1
        assert self.qname_off.shape[0] == n + 1
        assert int(self.cigar_off[-1]) == self.cigar.shape[0]
        assert int(self.qual_off[-1]) == self.qual.shape[0]
        for a in (self.pos, self.flag, self.mapq, self.cigar_off, self.cigar, self.qual_off, self.qual,
                  self.qname_off, self.qname):
            assert a.flags["C_CONTIGUOUS"]
        return self

    def slice(self, a, b):
        """Records [a, b) as their own ContigRecords (offsets rebased; the 4-bit sequence store is
        shared, its offsets stay absolute)."""
        c0, c1 = int(self.cigar_off[a]), int(self.cigar_off[b])
        q0, q1 = int(self.qual_off[a]), int(self.qual_off[b])
        n0, n1 = int(self.qname_off[a]), int(self.qname_off[b])
        return ContigRecords(
            pos=np.ascontiguousarray(self.pos[a:b]), flag=np.ascontiguousarray(self.flag[a:b]),
            mapq=np.ascontiguousarray(self.mapq[a:b]),
            cigar_off=(self.cigar_off[a:b + 1] - np.uint32(c0)).astype(np.uint32), cigar=np.ascontiguousarray(self.cigar[c0:c1]),
            qual_off=(self.qual_off[a:b + 1] - np.uint64(q0)).astype(np.uint64), qual=np.ascontiguousarray(self.qual[q0:q1]),
            qname_off=(self.qname_off[a:b + 1] - np.uint32(n0)).astype(np.uint32), qname=np.ascontiguousarray(self.qname[n0:n1]),
            seq_off=None if self.seq_off is None else np.ascontiguousarray(self.seq_off[a:b + 1]), seq4=self.seq4)

    @staticmethod
    def empty():
        return ContigRecords.from_reads([])

    @staticmethod
    def from_reads(reads):
        """reads: iterable of dicts/tuples (pos, cigar_str, mapq, quals, flag=0, name=None, seq=None).

        quals: list of ints, an int (constant quality over the query length), or None (l_seq = 0).
        seq: string over SEQ_CODES (optional)."""
        pos, flag, mapq, coff, cig, qoff, qual, noff, names = [], [], [], [0], [], [0], [], [0], []
        soff, seqcodes = [0], []
        have_seq = False
        for i, r in enumerate(reads):
            if isinstance(r, dict):
                p, cs, mq = r["pos"], r["cigar"], r["mapq"]
                q, fl, nm, sq = r.get("qual", 30), r.get("flag", 0), r.get("name"), r.get("seq")
            else:
                r = tuple(r)
                p, cs, mq = r[0], r[1], r[2]
                q = r[3] if len(r) > 3 else 30
                fl = r[4] if len(r) > 4 and r[4] is not None else 0
                nm = r[5] if len(r) > 5 else None
                sq = r[6] if len(r) > 6 else None
            words = cigar_from_string(cs)
            qlen = cigar_query_length(words)
            if q is None:
                ql = []
            elif isinstance(q, int):
                ql = [q] * qlen
            else:
                ql = list(q)
            pos.append(p); flag.append(fl); mapq.append(mq)
            cig.extend(words); coff.append(len(cig))
            qual.extend(ql); qoff.append(len(qual))
            nm = nm if nm is not None else f"r{i}"
            names.extend(nm if isinstance(nm, bytes) else nm.encode()); noff.append(len(names))
            if sq is not None:
                have_seq = True
                seqcodes.extend(SEQ_CODES.index(c) for c in sq)
            soff.append(len(seqcodes))
        rec = ContigRecords(
            pos=np.asarray(pos, dtype=np.int32), flag=np.asarray(flag, dtype=np.uint16),
            mapq=np.asarray(mapq, dtype=np.uint8), cigar_off=np.asarray(coff, dtype=np.uint32),
            cigar=np.asarray(cig, dtype=np.uint32), qual_off=np.asarray(qoff, dtype=np.uint64),
            qual=np.asarray(qual, dtype=np.uint8), qname_off=np.asarray(noff, dtype=np.uint32),
            qname=np.asarray(names, dtype=np.uint8))
        if have_seq:
            rec.seq_off = np.asarray(soff, dtype=np.uint64)
            rec.seq4 = pack_seq4(np.asarray(seqcodes, dtype=np.uint8))
        return rec.validate()


def pack_seq4(codes):
    """4-bit codes -> BAM packing (two per byte, first base in the high nibble)."""
    codes = np.asarray(codes, dtype=np.uint8)
    if codes.shape[0] % 2:
        codes = np.concatenate([codes, np.zeros(1, dtype=np.uint8)])
    return ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8)

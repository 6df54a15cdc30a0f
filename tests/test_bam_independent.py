"""An independent reading of the BAM the native reader is tested on.

The native BGZF/BAM/BAI reader (decodingustools_amd/csrc/bam_io.cpp; utils/bam_reader.rs:6-23 in the reference,
which hands this to htslib) is otherwise only ever fed files from the project's own writer (tests/bamio.py), so a
misreading of the format shared by the two could not show.  The decoder below shares nothing with either: it is
written from the SAM/BAM specification (sections 4.1 BGZF, 4.2 BAM, 5.2 BAI) with `zlib` and `struct` alone --
BGZF blocks walked by their BSIZE fields and inflated one by one (so that every record's virtual offset is known),
records unpacked field by field, long CIGARs taken from the CG:B,I tag, the .bai parsed bin by bin.  What the native
reader returns per contig must equal what this decoder finds, field for field, and the linear index in the .bai
must equal the one recomputed here from the records' virtual offsets.
"""
import gzip
import struct
import zlib

import numpy as np
import pytest

from bamio import write_bam
from decodingustools_amd import synth
from decodingustools_amd.bam import BamReader
from decodingustools_amd.records import ContigRecords


# ------------------------------------------------------------------------------------------------
# the independent decoder
# ------------------------------------------------------------------------------------------------
def bgzf_blocks(data):
    """[(file offset of the block, inflated bytes)] -- SAM spec 4.1: gzip members with an extra subfield
    'B','C' holding BSIZE = total block size - 1; CRC32 and ISIZE close every member."""
    out = []
    off = 0
    while off < len(data):
        id1, id2, cm, flg, _mtime, _xfl, _os, xlen = struct.unpack_from("<BBBBIBBH", data, off)
        assert (id1, id2, cm) == (31, 139, 8) and flg & 4, "not a BGZF member"
        extra = data[off + 12:off + 12 + xlen]
        bsize = None
        p = 0
        while p < len(extra):
            si1, si2, slen = struct.unpack_from("<BBH", extra, p)
            if (si1, si2) == (66, 67) and slen == 2:
                bsize = struct.unpack_from("<H", extra, p + 4)[0]
            p += 4 + slen
        assert bsize is not None
        total = bsize + 1
        cdata = data[off + 12 + xlen:off + total - 8]
        crc, isize = struct.unpack_from("<II", data, off + total - 8)
        raw = zlib.decompress(cdata, -15)
        assert len(raw) == isize and zlib.crc32(raw) == crc
        out.append((off, raw))
        off += total
    return out


class Stream:
    """The inflated byte stream with the virtual offset (coffset << 16 | uoffset) of every position."""

    def __init__(self, blocks):
        self.blocks = [(o, r) for o, r in blocks if len(r)]
        self.end = blocks[-1][0] << 16                      # the EOF block
        self.bi = 0
        self.bo = 0
        # a writer may name the end of a block (coffset << 16 | its length) where a reader names the start of the next
        # one: both are the same place in the stream
        self.next_of = {o: (len(r), blocks[i + 1][0]) for i, (o, r) in enumerate(blocks[:-1])}

    def canon(self, v):
        ln_next = self.next_of.get(v >> 16)
        while ln_next is not None and (v & 0xFFFF) == ln_next[0]:
            v = ln_next[1] << 16
            ln_next = self.next_of.get(v >> 16)
        return v

    def voffset(self):
        if self.bi >= len(self.blocks):
            return self.end
        return (self.blocks[self.bi][0] << 16) | self.bo

    def read(self, n):
        parts = []
        while n:
            if self.bi >= len(self.blocks):
                raise EOFError
            raw = self.blocks[self.bi][1]
            take = min(n, len(raw) - self.bo)
            parts.append(raw[self.bo:self.bo + take])
            self.bo += take
            n -= take
            if self.bo == len(raw):
                self.bi += 1
                self.bo = 0
        return b"".join(parts)

    def at_end(self):
        return self.bi >= len(self.blocks)


REF_CONSUMING = {0, 2, 3, 7, 8}     # M D N = X
QUERY_CONSUMING = {0, 1, 4, 7, 8}   # M I S = X


def decode_bam(path):
    data = open(path, "rb").read()
    blocks = bgzf_blocks(data)
    # the last member is the 28-byte empty EOF block
    assert len(blocks[-1][1]) == 0 and len(data) - blocks[-1][0] == 28
    # the whole file is also a valid multi-member gzip stream
    assert gzip.decompress(data) == b"".join(r for _, r in blocks)
    s = Stream(blocks)
    assert s.read(4) == b"BAM\1"
    l_text, = struct.unpack("<i", s.read(4))
    text = s.read(l_text).decode()
    n_ref, = struct.unpack("<i", s.read(4))
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack("<i", s.read(4))
        name = s.read(l_name)
        assert name.endswith(b"\0")
        l_ref, = struct.unpack("<i", s.read(4))
        refs.append((name[:-1].decode(), l_ref))
    records = []
    while not s.at_end():
        vo = s.voffset()
        block_size, = struct.unpack("<i", s.read(4))
        body = s.read(block_size)
        ref_id, pos, l_read_name, mapq, bin_, n_cigar_op, flag, l_seq, next_ref, next_pos, tlen = struct.unpack_from("<iiBBHHHIiii", body, 0)
        p = 32
        qname = body[p:p + l_read_name]; p += l_read_name
        assert qname.endswith(b"\0")
        cigar = list(struct.unpack_from(f"<{n_cigar_op}I", body, p)); p += 4 * n_cigar_op
        seq = body[p:p + (l_seq + 1) // 2]; p += (l_seq + 1) // 2
        qual = body[p:p + l_seq]; p += l_seq
        # optional fields
        tags = {}
        while p < len(body):
            tag = body[p:p + 2].decode(); typ = chr(body[p + 2]); p += 3
            if typ in "AcC":
                val = body[p]; p += 1
            elif typ in "sS":
                val = struct.unpack_from("<H", body, p)[0]; p += 2
            elif typ in "iIf":
                val = struct.unpack_from("<I", body, p)[0]; p += 4
            elif typ in "ZH":
                e = body.index(b"\0", p); val = body[p:e]; p = e + 1
            elif typ == "B":
                sub = chr(body[p]); cnt, = struct.unpack_from("<I", body, p + 1); p += 5
                size = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
                val = (sub, body[p:p + size * cnt]); p += size * cnt
            else:
                raise AssertionError(f"unknown tag type {typ}")
            tags[tag] = val
        assert p == len(body)
        # SAM spec 4.2.2: a CIGAR of more than 65535 operations is stored in CG:B,I and the cigar field reads
        # <l_seq>S<reference length>N
        if "CG" in tags and len(cigar) == 2 and (cigar[0] & 15) == 4 and (cigar[0] >> 4) == l_seq and (cigar[1] & 15) == 3:
            sub, raw = tags["CG"]
            assert sub == "I"
            real = list(struct.unpack(f"<{len(raw) // 4}I", raw))
            assert sum(c >> 4 for c in real if (c & 15) in REF_CONSUMING) == cigar[1] >> 4
            cigar = real
        records.append(dict(voffset=vo, end_voffset=s.voffset(), ref_id=ref_id, pos=pos, mapq=mapq, bin=bin_, flag=flag, l_seq=l_seq,
                            next_ref=next_ref, next_pos=next_pos, tlen=tlen, qname=qname[:-1], cigar=cigar, seq=seq, qual=qual))
    for r in records:
        r["end_voffset"] = s.canon(r["end_voffset"])
    decode_bam.canon = s.canon
    return text, refs, records


def reg2bin(beg, end):
    """SAM spec 5.3 (end exclusive)."""
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def decode_bai(path):
    d = open(path, "rb").read()
    assert d[:4] == b"BAI\1"
    n_ref, = struct.unpack_from("<i", d, 4)
    p = 8
    out = []
    for _ in range(n_ref):
        n_bin, = struct.unpack_from("<i", d, p); p += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", d, p); p += 8
            chunks = [struct.unpack_from("<QQ", d, p + 16 * k) for k in range(n_chunk)]
            p += 16 * n_chunk
            bins[b] = chunks
        n_intv, = struct.unpack_from("<i", d, p); p += 4
        ioff = list(struct.unpack_from(f"<{n_intv}Q", d, p)); p += 8 * n_intv
        out.append((bins, ioff))
    assert p == len(d) or p + 8 == len(d)          # optional n_no_coor
    return out


def ref_span(cigar):
    return sum(c >> 4 for c in cigar if (c & 15) in REF_CONSUMING)


# ------------------------------------------------------------------------------------------------
# the tests
# ------------------------------------------------------------------------------------------------
def _inputs():
    L0, L2, L3 = 60_000, 9_000, 40_000
    ref0 = synth.make_reference(L0, 5)
    recs = {0: synth.short_read_contig(L0, 25, 77, with_seq=True, ref=ref0),
            2: synth.adversarial_contig(L2, 400, 78),
            3: synth.long_read_contig(L3, 10, 79)}
    refs = [("chr1", L0), ("chrEmpty", 700), ("chrX", L2), ("chrY", L3)]
    return refs, recs


def _compare_contig(native: ContigRecords, mine, with_seq):
    assert native.n == len(mine)
    assert native.pos.tolist() == [r["pos"] for r in mine]
    assert native.flag.tolist() == [r["flag"] for r in mine]
    assert native.mapq.tolist() == [r["mapq"] for r in mine]
    cig = [c for r in mine for c in r["cigar"]]
    assert native.cigar.tolist() == cig
    assert native.cigar_off.tolist() == np.concatenate([[0], np.cumsum([len(r["cigar"]) for r in mine])]).tolist()
    assert bytes(native.qual) == b"".join(r["qual"] for r in mine)
    assert native.qual_off.tolist() == np.concatenate([[0], np.cumsum([r["l_seq"] for r in mine])]).tolist()
    assert bytes(native.qname) == b"".join(r["qname"] for r in mine)
    assert native.qname_off.tolist() == np.concatenate([[0], np.cumsum([len(r["qname"]) for r in mine])]).tolist()
    if with_seq:
        # the native store is nibble-continuous across records; the file pads every record to a byte
        codes = []
        for r in mine:
            for j in range(r["l_seq"]):
                b = r["seq"][j >> 1]
                codes.append(b >> 4 if (j & 1) == 0 else b & 15)
        got = []
        for j in range(int(native.seq_off[-1])):
            b = int(native.seq4[j >> 1])
            got.append(b >> 4 if (j & 1) == 0 else b & 15)
        assert got == codes
        assert native.seq_off.tolist() == native.qual_off.tolist()


@pytest.mark.parametrize("layout", ["aligned", "straddling", "cg_tags"])
def test_native_reader_agrees_with_an_independent_decoder(tmp_path, layout):
    refs, recs = _inputs()
    path = str(tmp_path / f"{layout}.bam")
    if layout == "aligned":
        write_bam(path, refs, recs, block_every=37)                 # blocks cut at record boundaries
    elif layout == "straddling":
        write_bam(path, refs, recs)                                 # 64 KiB blocks: records straddle them
    else:
        write_bam(path, refs, recs, long_cigar_tag=True)            # CIGARs of more than 3 operations go to CG:B,I
    text, drefs, drecs = decode_bam(path)
    assert drefs == refs and text.startswith("@HD")
    assert len(drecs) == sum(r.n for r in recs.values())
    if layout == "straddling":
        # some record does start in one block and end in another
        assert any((r["voffset"] >> 16) != (r["end_voffset"] >> 16) and (r["end_voffset"] & 0xFFFF) != 0 for r in drecs)
    if layout == "cg_tags":
        assert max(len(r["cigar"]) for r in drecs) > 100
    # the generator's records survive the writer as this decoder reads them ...
    for tid, rec in recs.items():
        mine = [r for r in drecs if r["ref_id"] == tid]
        assert [r["pos"] for r in mine] == rec.pos.tolist() and [c for r in mine for c in r["cigar"]] == rec.cigar.tolist()
        # ... with the bin the specification asks for (pos .. pos + reference span, one base for a spanless read)
        for r in mine:
            span = ref_span(r["cigar"])
            assert r["bin"] == reg2bin(r["pos"], r["pos"] + max(span, 1))
    # ... and the native reader returns exactly what this decoder finds, contig by contig
    with BamReader(path) as b:
        assert b.target_names == [n for n, _ in refs] and b.target_lens == [l for _, l in refs]
        for tid in range(len(refs)):
            mine = [r for r in drecs if r["ref_id"] == tid]
            _compare_contig(b.fetch_contig(tid, with_seq=(tid == 0)), mine, with_seq=(tid == 0))
        # out of order too (the index is used to seek)
        _compare_contig(b.fetch_contig(3), [r for r in drecs if r["ref_id"] == 3], False)
        _compare_contig(b.fetch_contig(0), [r for r in drecs if r["ref_id"] == 0], False)


def test_bai_equals_the_index_recomputed_from_virtual_offsets(tmp_path):
    """SAM spec 5.2: per reference the bins with their chunks, and the linear index -- for every 16 kb window the
    smallest virtual offset of a record overlapping it."""
    refs, recs = _inputs()
    path = str(tmp_path / "i.bam")
    write_bam(path, refs, recs)
    _, _, drecs = decode_bam(path)
    bai = decode_bai(path + ".bai")
    canon = decode_bam.canon
    assert len(bai) == len(refs)
    for tid, (bins, ioff) in enumerate(bai):
        mine = [r for r in drecs if r["ref_id"] == tid]
        meta = bins.pop(37450, None)                                # the pseudo-bin: file range + mapped / unmapped counts
        if not mine:
            assert not bins and not ioff
            continue
        if meta is not None:
            (beg, end), (n_mapped, n_unmapped) = meta
            assert canon(beg) == mine[0]["voffset"] and canon(end) == mine[-1]["end_voffset"]
            assert n_mapped == sum(1 for r in mine if not r["flag"] & 4) and n_unmapped == sum(1 for r in mine if r["flag"] & 4)
        # every record lies inside a chunk of its bin
        for r in mine:
            assert any(canon(cb) <= r["voffset"] and r["end_voffset"] <= canon(ce) for cb, ce in bins[r["bin"]]), (tid, r["pos"])
        # chunks of a bin cover nothing but records of that bin
        by_bin = {}
        for r in mine:
            by_bin.setdefault(r["bin"], []).append(r)
        assert set(bins) == set(by_bin)
        for b, chunks in bins.items():
            assert chunks == sorted(chunks)
            assert canon(chunks[0][0]) == by_bin[b][0]["voffset"] and canon(chunks[-1][1]) == by_bin[b][-1]["end_voffset"]
        # the linear index
        want = {}
        for r in mine:
            span = max(ref_span(r["cigar"]), 1)
            for w in range(r["pos"] >> 14, ((r["pos"] + span - 1) >> 14) + 1):
                want[w] = min(want.get(w, r["voffset"]), r["voffset"])
        assert len(ioff) == max(want) + 1
        for w, v in enumerate(ioff):
            if w in want:
                assert canon(v) == want[w], (tid, w)
            else:
                # a window no record overlaps: 0, or filled from a neighbour (htslib back-fills with the next window's)
                later = [want[k] for k in want if k > w]
                earlier = [want[k] for k in want if k < w]
                assert v == 0 or canon(v) in (min(later) if later else None, max(earlier) if earlier else None), (tid, w)

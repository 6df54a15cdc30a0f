"""TEST INFRASTRUCTURE: pure-Python writers of BGZF/BAM (+ .bai) and FASTA (+ .fai) files from
ContigRecords, written from the SAM/BAM specification, independent of the C++ reader under test."""
import struct
import zlib

import numpy as np


class _BgzfWriter:
    def __init__(self, path):
        self.f = open(path, "wb")
        self.buf = bytearray()
        self.coff = 0

    def tell(self):
        return (self.coff << 16) | len(self.buf)

    def write(self, data):
        self.buf += data
        while len(self.buf) >= 0xFF00:
            self._flush(0xFF00)

    def _flush(self, n):
        chunk = bytes(self.buf[:n]); del self.buf[:n]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        bsize = len(comp) + 25
        self.f.write(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize))
        self.f.write(comp)
        self.f.write(struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
        self.coff += bsize + 1

    def flush_block(self):
        if self.buf:
            self._flush(len(self.buf))

    def close(self):
        self.flush_block()
        self._flush(0)                      # the EOF marker block
        self.f.close()


def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def write_bam(path, refs, per_tid, header_text="@HD\tVN:1.6\tSO:coordinate\n", write_index=True, block_every=None,
              long_cigar_tag=False, tlen=None, unmapped_tail=(), csi=False):
    """refs: [(name, length)]; per_tid: {tid: ContigRecords} (coordinate sorted).  Writes path and,
    if asked, path + '.bai' (csi=True: path + '.csi' instead).  block_every: start a new BGZF block every that many records (exercises
    records that straddle / start blocks).  long_cigar_tag: store CIGARs with more than 3 ops in a
    CG:B,I tag behind the <l_seq>S<reflen>N placeholder, the way BAM stores > 65535 ops.
    tlen: {tid: sequence of template lengths}; unmapped_tail: [(name bytes, flag, l_seq, tlen)] records
    with refID -1 written after the last contig."""
    text = header_text + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs)
    w = _BgzfWriter(path)
    w.write(b"BAM\1" + struct.pack("<I", len(text)) + text.encode() + struct.pack("<I", len(refs)))
    for n, l in refs:
        w.write(struct.pack("<I", len(n) + 1) + n.encode() + b"\0" + struct.pack("<I", l))
    w.flush_block()
    index = {}
    count = 0
    for tid in sorted(per_tid):
        rec = per_tid[tid]
        bins, lin = {}, {}
        for i in range(rec.n):
            if block_every and count % block_every == 0:
                w.flush_block()
            count += 1
            cig = rec.cigar[rec.cigar_off[i]:rec.cigar_off[i + 1]]
            qual = rec.qual[int(rec.qual_off[i]):int(rec.qual_off[i + 1])]
            name = bytes(rec.qname[rec.qname_off[i]:rec.qname_off[i + 1]]) + b"\0"
            l_seq = len(qual)
            if rec.seq4 is not None:
                s0 = int(rec.seq_off[i]); ls = int(rec.seq_off[i + 1]) - s0
                codes = np.array([(rec.seq4[(s0 + j) >> 1] >> 4) if ((s0 + j) & 1) == 0 else (rec.seq4[(s0 + j) >> 1] & 15)
                                  for j in range(ls)], dtype=np.uint8)
                l_seq = ls
            else:
                codes = np.full(l_seq, 1, dtype=np.uint8)
            if l_seq % 2: codes = np.concatenate([codes, np.zeros(1, np.uint8)])
            seq = bytes(((codes[0::2] << 4) | codes[1::2]).astype(np.uint8))
            q = bytes(qual) if len(qual) == l_seq else b"\xff" * l_seq
            rlen = int(sum(int(c) >> 4 for c in cig if (int(c) & 15) in (0, 2, 3, 7, 8)))
            pos = int(rec.pos[i])
            end = pos + (rlen if rlen else 1)
            aux = b"NM" + b"C" + bytes([0])
            cig_store = cig
            if long_cigar_tag and len(cig) > 3:
                aux += b"CG" + b"B" + b"I" + struct.pack("<I", len(cig)) + np.asarray(cig, dtype="<u4").tobytes()
                cig_store = np.asarray([(l_seq << 4) | 4, (rlen << 4) | 3], dtype=np.uint32)
            body = struct.pack("<iiBBHHHIiii", tid, pos, len(name), int(rec.mapq[i]), _reg2bin(pos, end), len(cig_store),
                               int(rec.flag[i]), l_seq, -1, -1, int(tlen[tid][i]) if tlen and tid in tlen else 0)
            body += name + np.asarray(cig_store, dtype="<u4").tobytes() + seq + q + aux
            v0 = w.tell()
            w.write(struct.pack("<I", len(body)) + body)
            v1 = w.tell()
            b = _reg2bin(pos, end)
            ch = bins.setdefault(b, [])
            if ch and ch[-1][1] == v0: ch[-1][1] = v1
            else: ch.append([v0, v1])
            for win in range(pos >> 14, ((end - 1) >> 14) + 1):
                lin.setdefault(win, v0)
        if rec.n:
            # the metadata pseudo-bin (SAM spec 5.2): [begin, end) virtual offsets, then n_mapped, n_unmapped
            n_unm = int(np.count_nonzero(rec.flag & 4))
            first = min(c[0] for ch in bins.values() for c in ch); last = max(c[1] for ch in bins.values() for c in ch)
            bins[37450] = [[first, last], [rec.n - n_unm, n_unm]]
        index[tid] = (bins, lin)
    for name, flag, l_seq, tl in unmapped_tail:
        nm = bytes(name) + b"\0"
        body = struct.pack("<iiBBHHHIiii", -1, -1, len(nm), 0, 4680, 0, int(flag), l_seq, -1, -1, int(tl))
        body += nm + bytes((l_seq + 1) // 2) + b"\xff" * l_seq
        w.write(struct.pack("<I", len(body)) + body)
    w.close()
    if write_index and csi:
        # .csi with the BAM defaults min_shift 14, depth 5 (same bin numbering as .bai; per bin a loffset, no linear
        # index), BGZF-compressed like every .csi
        cw = _BgzfWriter(path + ".csi")
        cw.write(b"CSI\1" + struct.pack("<iii", 14, 5, 0) + struct.pack("<i", len(refs)))
        for tid in range(len(refs)):
            bins, lin = index.get(tid, ({}, {}))
            cw.write(struct.pack("<i", len(bins)))
            for b in sorted(bins):
                loff = min(v0 for v0, _ in bins[b]) if b != 37450 else 0
                cw.write(struct.pack("<IQi", b, loff, len(bins[b])))
                for v0, v1 in bins[b]:
                    cw.write(struct.pack("<QQ", v0, v1))
        cw.write(struct.pack("<Q", 0))
        cw.close()
    elif write_index:
        with open(path + ".bai", "wb") as f:
            f.write(b"BAI\1" + struct.pack("<I", len(refs)))
            for tid in range(len(refs)):
                bins, lin = index.get(tid, ({}, {}))
                f.write(struct.pack("<I", len(bins)))
                for b in sorted(bins):
                    f.write(struct.pack("<II", b, len(bins[b])))
                    for v0, v1 in bins[b]:
                        f.write(struct.pack("<QQ", v0, v1))
                n_intv = (max(lin) + 1) if lin else 0
                f.write(struct.pack("<I", n_intv))
                last = 0
                for win in range(n_intv):
                    last = lin.get(win, last)
                    f.write(struct.pack("<Q", last))
            f.write(struct.pack("<Q", 0))


def write_fasta(path, seqs, width=60):
    """seqs: [(name, uint8 ndarray)]; writes path and path + '.fai'."""
    fai = []
    with open(path, "wb") as f:
        for name, arr in seqs:
            f.write(b">" + name.encode() + b" synthetic\n")
            off = f.tell()
            b = bytes(arr)
            for i in range(0, len(b), width):
                f.write(b[i:i + width] + b"\n")
            fai.append((name, len(b), off, width, width + 1))
    with open(path + ".fai", "w") as f:
        for e in fai:
            f.write("\t".join(str(x) for x in e) + "\n")

"""Tooling helpers shared by e2e_bench.py and branch_bench.py: native BAM writer, a one-reference .bai, FASTA."""
import ctypes as C, os, struct, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_bam_native(out, path, name, L, rec, level=1, threads=16, header="@HD\tVN:1.6\tSO:coordinate\n@PG\tID:bwa\tPN:bwa\n"):
    so = os.path.join(out, "bamwriter.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "bamwriter.cpp"), "-lz", "-lpthread", "-o", so])
    lib = C.CDLL(so)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.tool_write_bam.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint64] + [C.c_void_p] * 9 + [C.c_int, C.c_int]
    rc = lib.tool_write_bam(path.encode(), header.encode(), name.encode(), L, rec.n, p(rec.pos), p(rec.flag), p(rec.mapq),
                            p(rec.cigar_off), p(rec.cigar), p(rec.qual_off), p(rec.qual), p(rec.qname_off), p(rec.qname), level, threads)
    assert rc == 0


def write_single_ref_bai(path, n_reads, first_voff=None):
    """A minimal index for a one-reference BAM: one bin-0 chunk starting at the first record (the reader only
    needs the start offset) and the metadata pseudo-bin."""
    # the first record follows the header block: virtual offset = (size of block 0) << 16; the caller's BAM has the
    # header alone in block 0, so read its BSIZE
    bam = path[:-4]
    with open(bam, "rb") as f:
        h = f.read(18)
    bsize = struct.unpack_from("<H", h, 16)[0] + 1
    v0 = bsize << 16
    end = os.path.getsize(bam) << 16
    with open(path, "wb") as f:
        f.write(b"BAI\1" + struct.pack("<I", 1))
        f.write(struct.pack("<I", 2))
        f.write(struct.pack("<II", 0, 1) + struct.pack("<QQ", v0, end))
        f.write(struct.pack("<II", 37450, 2) + struct.pack("<QQ", v0, end) + struct.pack("<QQ", n_reads, 0))
        f.write(struct.pack("<I", 1) + struct.pack("<Q", v0))
        f.write(struct.pack("<Q", 0))


def write_fasta(path, name, ref, width=60):
    L = ref.shape[0]
    with open(path, "wb") as f:
        head = (">" + name + "\n").encode()
        f.write(head)
        full = (L // width) * width
        body = np.empty((L // width, width + 1), np.uint8); body[:, :width] = ref[:full].reshape(-1, width); body[:, width] = 10
        f.write(body.tobytes()); f.write(ref[full:].tobytes() + b"\n")
    open(path + ".fai", "w").write(f"{name}\t{L}\t{len(head)}\t{width}\t{width + 1}\n")


def write_multi_bam(out, path, contigs, level=1, threads=16, header="@HD\tVN:1.6\tSO:coordinate\n@PG\tID:bwa\tPN:bwa\n"):
    """contigs: [(name, length, ContigRecords)] in tid order -> BAM + .bai (one chunk per reference: the reader
    only needs each reference's first virtual offset; plus the metadata pseudo-bin with the mapped count)."""
    so = os.path.join(out, "bamwriter.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "bamwriter.cpp"), "-lz", "-lpthread", "-o", so])
    lib = C.CDLL(so)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    names = (C.c_char_p * len(contigs))(*[c[0].encode() for c in contigs])
    lens = (C.c_uint32 * len(contigs))(*[c[1] for c in contigs])
    lib.tool_write_bam_head.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_int]
    assert lib.tool_write_bam_head(path.encode(), header.encode(), len(contigs), names, lens, level) == 0
    lib.tool_write_bam_body.argtypes = [C.c_char_p, C.c_int32, C.c_uint64] + [C.c_void_p] * 9 + [C.c_int, C.c_int]
    spans = []
    for tid, (_, _, rec) in enumerate(contigs):
        v0 = os.path.getsize(path) << 16
        if rec.n:
            assert lib.tool_write_bam_body(path.encode(), tid, rec.n, p(rec.pos), p(rec.flag), p(rec.mapq), p(rec.cigar_off), p(rec.cigar),
                                           p(rec.qual_off), p(rec.qual), p(rec.qname_off), p(rec.qname), level, threads) == 0
        spans.append((v0, os.path.getsize(path) << 16, rec.n, int(((rec.flag & 4) != 0).sum()) if rec.n else 0))
    lib.tool_write_bam_eof.argtypes = [C.c_char_p]
    assert lib.tool_write_bam_eof(path.encode()) == 0
    with open(path + ".bai", "wb") as f:
        f.write(b"BAI\1" + struct.pack("<I", len(contigs)))
        for v0, v1, n, n_un in spans:
            if n == 0:
                f.write(struct.pack("<II", 0, 0))
                continue
            f.write(struct.pack("<I", 2))
            f.write(struct.pack("<II", 0, 1) + struct.pack("<QQ", v0, v1))
            f.write(struct.pack("<II", 37450, 2) + struct.pack("<QQ", v0, v1) + struct.pack("<QQ", n - n_un, n_un))
            f.write(struct.pack("<I", 1) + struct.pack("<Q", v0))
        f.write(struct.pack("<Q", 0))


def write_multi_fasta(path, contigs, width=60):
    """contigs: [(name, uint8 bases)] -> FASTA + .fai"""
    fai = []
    with open(path, "wb") as f:
        for name, ref in contigs:
            head = (">" + name + "\n").encode()
            f.write(head)
            off = f.tell()
            L = ref.shape[0]
            full = (L // width) * width
            body = np.empty((L // width, width + 1), np.uint8); body[:, :width] = ref[:full].reshape(-1, width); body[:, width] = 10
            f.write(body.tobytes())
            if L > full:
                f.write(ref[full:].tobytes() + b"\n")
            fai.append(f"{name}\t{L}\t{off}\t{width}\t{width + 1}\n")
    open(path + ".fai", "w").write("".join(fai))

"""CPU tests of the native BGZF/BAM/BAI and FASTA/FAI reader against files written by the
independent pure-Python writer in tests/bamio.py."""
import os
import numpy as np
import pytest

from bamio import write_bam, write_fasta
from decodingustools_amd import synth
from decodingustools_amd.bam import BamReader, FastaReader
from decodingustools_amd.records import ContigRecords


def _same(a: ContigRecords, b: ContigRecords, seq=False):
    for f in ("pos", "flag", "mapq", "cigar_off", "cigar", "qual_off", "qual", "qname_off", "qname"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    if seq:
        assert np.array_equal(a.seq_off, b.seq_off)
        n = int(a.seq_off[-1])
        ca = np.array([(a.seq4[j >> 1] >> 4) if j % 2 == 0 else (a.seq4[j >> 1] & 15) for j in range(n)])
        cb = np.array([(b.seq4[j >> 1] >> 4) if j % 2 == 0 else (b.seq4[j >> 1] & 15) for j in range(n)])
        assert np.array_equal(ca, cb)


def _dataset():
    refs = [("chr1", 30_000), ("chr2", 5_000), ("chrEmpty", 1_000), ("chrM", 16_569)]
    per = {0: synth.short_read_contig(30_000, 20, 11), 1: synth.adversarial_contig(5_000, 400, 12, overhang=True),
           3: synth.adversarial_contig(16_569, 900, 13, deep=True)}
    return refs, per


@pytest.mark.parametrize("index", [True, False])
@pytest.mark.parametrize("block_every", [None, 7])
def test_bam_round_trip(tmp_path, index, block_every):
    refs, per = _dataset()
    path = str(tmp_path / "t.bam")
    write_bam(path, refs, per, write_index=index, block_every=block_every,
              header_text="@HD\tVN:1.6\tSO:coordinate\n@PG\tID:bwa\tPN:bwa\n")
    with BamReader(path) as r:
        assert r.target_names == [n for n, _ in refs] and r.target_lens == [l for _, l in refs]
        assert r.has_index == index and "@PG\tID:bwa" in r.header_text
        # any order of contigs, including going back and an empty one
        for tid in (3, 0, 2, 1, 0):
            got = r.fetch_contig(tid)
            _same(got, per.get(tid, ContigRecords.empty()))


def test_bam_sequences_and_long_cigar_tag(tmp_path):
    L = 4000
    ref = synth.make_reference(L, 3)
    rec = synth.short_read_contig(L, 15, 21, with_seq=True, ref=ref)
    lr = synth.long_read_contig(20_000, 10, 5)
    refs = [("chrA", L), ("chrB", 20_000)]
    path = str(tmp_path / "s.bam")
    write_bam(path, refs, {0: rec, 1: lr}, long_cigar_tag=True)
    with BamReader(path) as r:
        _same(r.fetch_contig(0, with_seq=True), rec, seq=True)
        _same(r.fetch_contig(1), lr)            # CIGARs come back from the CG tag


def test_fasta_fetch(tmp_path):
    a = synth.make_reference(1234, 1, lowercase=True)
    b = synth.make_reference(61, 2)
    c = np.zeros(0, np.uint8)
    path = str(tmp_path / "r.fa")
    write_fasta(path, [("chr1", a), ("chr2", b), ("empty", c)], width=50)
    f = FastaReader(path)
    assert np.array_equal(f.fetch("chr2"), b) and np.array_equal(f.fetch("chr1"), a)
    assert f.fetch("empty").shape[0] == 0
    with pytest.raises(OSError, match="'nope' not found in the reference FASTA index"):
        f.fetch("nope")                       # fetch_seq(..)? fails the contig in the reference (mod.rs:79)
    assert np.array_equal(f.fetch("chr2"), b)   # the reader is usable after a failed fetch
    f.close()
    with pytest.raises(OSError):
        FastaReader(str(tmp_path / "missing.fa"))


def test_fasta_index_is_built_when_missing_and_validated_when_present(tmp_path):
    """faidx::Reader::from_path (api/coverage.rs:73) builds the .fai when there is none; a malformed one is an
    error at open, not a division by zero or an allocation of a wrapped-around size later."""
    a = synth.make_reference(1234, 1, lowercase=True)
    b = synth.make_reference(61, 2)
    path = str(tmp_path / "r.fa")
    write_fasta(path, [("chr1", a), ("chr2", b), ("empty", np.zeros(0, np.uint8)), ("tail", b[:50])], width=50)
    want = open(path + ".fai").read()
    os.remove(path + ".fai")
    f = FastaReader(path)
    assert np.array_equal(f.fetch("chr1"), a) and np.array_equal(f.fetch("chr2"), b) and np.array_equal(f.fetch("tail"), b[:50])
    assert f.fetch("empty").shape[0] == 0
    f.close()
    got = open(path + ".fai").read()
    rows = lambda t: [r.split("\t") for r in t.splitlines()]
    assert [r[:3] for r in rows(got)] == [r[:3] for r in rows(want)]
    assert [r for r in rows(got) if r[1] != "0"] == [r for r in rows(want) if r[1] != "0"]
    # CRLF line ends and no newline at the end of the file
    crlf = str(tmp_path / "c.fa")
    with open(crlf, "wb") as fo:
        fo.write(b">s1 x\r\nACGTAC\r\nGTNN\r\n>s2\r\nTTTT")
    f = FastaReader(crlf)
    assert bytes(f.fetch("s1")) == b"ACGTACGTNN" and bytes(f.fetch("s2")) == b"TTTT"
    f.close()
    # lines of different lengths inside a sequence cannot be indexed
    bad = str(tmp_path / "b.fa")
    open(bad, "wb").write(b">s\nACGT\nAC\nACGT\n")
    with pytest.raises(OSError, match="different line length"):
        FastaReader(bad)
    # malformed rows of an existing index
    for row, what in (("chr1\t1234\t16\t0\t51\n", "invalid row"), ("chr1\t1234\t16\t50\t49\n", "invalid row"),
                      ("chr1\tabc\n", "malformed row")):
        open(path + ".fai", "w").write(row)
        with pytest.raises(OSError, match=what):
            FastaReader(path)
    # a file shorter than its index says: the bases that are there
    open(path + ".fai", "w").write(want)
    size = os.path.getsize(path)
    with open(path, "r+b") as fo:
        fo.truncate(size - 80)
    f = FastaReader(path)
    assert np.array_equal(f.fetch("chr1"), a)
    assert f.fetch("tail").shape[0] < 50
    f.close()


def test_not_a_bam(tmp_path):
    p = tmp_path / "x.bam"
    p.write_bytes(b"hello world, not bgzf at all........")
    with pytest.raises(OSError):
        BamReader(str(p))


def test_config1_chrM_plumbing_through_files(tmp_path):
    """BASELINE.json configs[0]: coverage -L chrM on a tiny ~16 kb BAM @ 20x -- file plumbing only, the
    per-position work done by the CPU oracle: records read back from BAM/FASTA files give the same
    BED as the in-memory records."""
    import oracle
    from helpers import make_options, oracle_run
    L = 16_569
    seed = synth.seed_for(1, 24)
    rec = synth.short_read_contig(L, 20, seed)
    ref = synth.make_reference(L, seed, lowercase=True)
    refs = [("chr1", 50_000), ("chrM", L)]
    other = synth.short_read_contig(50_000, 5, seed + 1)
    bam = str(tmp_path / "m.bam"); fa = str(tmp_path / "m.fa")
    write_bam(bam, refs, {0: other, 1: rec})
    write_fasta(fa, [("chr1", synth.make_reference(50_000, 9)), ("chrM", ref)])
    with BamReader(bam) as r:
        tid = r.target_names.index("chrM")
        got = r.fetch_contig(tid)
        assert r.target_lens[tid] == L
    f = FastaReader(fa); ref2 = f.fetch("chrM"); f.close()
    assert np.array_equal(ref2, ref)
    opt = make_options({})
    _, bed_files = oracle_run([("chrM", 1, L, ref2, got)], opt, str(tmp_path / "a.bed"))
    _, bed_mem = oracle_run([("chrM", 1, L, ref, rec)], opt, str(tmp_path / "b.bed"))
    assert bed_files == bed_mem and bed_files.count("\n") > 10


def test_corrupted_files_fail_cleanly(tmp_path):
    """Truncated, bit-flipped and structurally corrupted BAM files: the reader either decodes them or
    returns an error; it never reads out of bounds (run under the test process: a fault would kill it)."""
    import random
    import struct
    import zlib
    from bamio import _BgzfWriter
    from decodingustools_amd.callable_loci import EngineError
    from decodingustools_amd.report import BamStats
    L = 30_000
    recs = {0: synth.short_read_contig(L, 15, 1, with_seq=True, ref=synth.make_reference(L, 2)), 1: synth.adversarial_contig(8000, 200, 3)}
    good = str(tmp_path / "g.bam")
    write_bam(good, [("a", L), ("b", 8000)], recs, block_every=40)
    data = open(good, "rb").read()
    rng = random.Random(5)
    bad = str(tmp_path / "x.bam")

    def try_read(with_seq):
        try:
            with BamReader(bad) as r:
                for tid in range(min(len(r.target_names), 4)):
                    x = r.fetch_contig(tid, with_seq=with_seq)
                    x.qual.sum(); x.cigar.sum(); x.qname.sum()
            try:
                BamStats(500).collect_stats(bad)
            except EngineError:
                pass
            return True
        except (EngineError, OSError):
            return False
    # container level: truncation, bit flips, deleted / inserted bytes
    n_ok = 0
    for t in range(60):
        d = bytearray(data)
        mode = t % 4
        if mode == 0: d = d[:rng.randrange(len(d))]
        elif mode == 1:
            for _ in range(rng.randrange(1, 6)): d[rng.randrange(len(d))] ^= 1 << rng.randrange(8)
        elif mode == 2:
            a = rng.randrange(len(d)); d[a:a + rng.randrange(1, 200)] = b""
        else:
            a = rng.randrange(len(d)); d[a:a] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 100)))
        open(bad, "wb").write(bytes(d))
        n_ok += try_read(t % 2 == 0)
    assert n_ok < 30                                   # most container damage must be noticed (CRC, sizes)
    # record level: valid BGZF around damaged BAM content
    o = 0; raw = bytearray()
    while o < len(data):
        bsize = struct.unpack_from("<H", data, o + 16)[0] + 1
        raw += zlib.decompress(data[o + 18:o + bsize - 8], -15)
        o += bsize
    p = 12 + struct.unpack_from("<I", raw, 4)[0]
    for _ in range(struct.unpack_from("<I", raw, p - 4)[0]):
        p += 8 + struct.unpack_from("<I", raw, p)[0]
    n_err = 0
    for t in range(60):
        d = bytearray(raw)
        for _ in range(rng.randrange(1, 4)):
            a = rng.randrange(p, len(d) - 4)
            if t % 3 == 0: d[a] = rng.randrange(256)
            elif t % 3 == 1: d[a] ^= 1 << rng.randrange(8)
            else: d[a:a + 4] = struct.pack("<I", rng.choice([0, 1, 0xFFFFFFFF, 0x7FFFFFFF, 1 << 29, rng.randrange(1 << 32)]))
        w = _BgzfWriter(bad); w.write(bytes(d)); w.close()
        n_err += not try_read(t % 2 == 1)
    assert 0 < n_err < 60


def test_zlib_and_libdeflate_inflate_agree(tmp_path):
    """The reader binds libdeflate at run time when the machine has it and uses zlib otherwise
    (DUT_INFLATE=zlib forces that): both must decode the same records."""
    import subprocess
    import sys
    L = 50_000
    rec = synth.short_read_contig(L, 25, 9, with_seq=True, ref=synth.make_reference(L, 10))
    bam = str(tmp_path / "z.bam")
    write_bam(bam, [("a", L)], {0: rec}, block_every=300)
    code = ("import sys, hashlib; sys.path.insert(0, %r); from decodingustools_amd.bam import BamReader\n"
            "x = BamReader(%r).fetch_contig(0, with_seq=True)\n"
            "h = hashlib.sha256()\n"
            "[h.update(getattr(x, f).tobytes()) for f in ('pos','flag','mapq','cigar_off','cigar','qual_off','qual','qname_off','qname','seq_off','seq4')]\n"
            "print(x.n, h.hexdigest())\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), bam)
    outs = []
    for mode in ("", "zlib"):
        env = dict(os.environ, DUT_INFLATE=mode, DUT_THREADS="3")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout.strip())
    assert outs[0] == outs[1] and outs[0].split()[0] == str(rec.n)


def test_index_metadata_gives_mapped_read_counts(tmp_path):
    refs, per = _dataset()
    path = str(tmp_path / "m.bam")
    write_bam(path, refs, per)
    with BamReader(path) as r:
        for tid in range(len(refs)):
            want = int(np.count_nonzero((per[tid].flag & 4) == 0)) if tid in per and per[tid].n else -1
            assert r.target_mapped[tid] == want
    write_bam(path, refs, per, write_index=False)
    os.remove(path + ".bai") if os.path.exists(path + ".bai") else None
    with BamReader(path) as r:
        assert r.target_mapped == [-1] * len(refs)


def test_csi_index_is_used_like_a_bai(tmp_path):
    """A BAM indexed with .csi (BGZF-compressed, bins with loffset): random access per contig and the mapped counts of
    the metadata pseudo-bin work as with a .bai."""
    refs, per = _dataset()
    path = str(tmp_path / "c.bam")
    write_bam(path, refs, per, csi=True, block_every=50)
    assert os.path.exists(path + ".csi") and not os.path.exists(path + ".bai")
    with BamReader(path) as r:
        assert r.has_index
        for tid in range(len(refs)):
            want = int(np.count_nonzero((per[tid].flag & 4) == 0)) if tid in per and per[tid].n else -1
            assert r.target_mapped[tid] == want
        for tid in reversed(range(len(refs))):                 # out of order: needs the index
            got = r.fetch_contig(tid)
            exp = per.get(tid)
            assert got.n == (exp.n if exp is not None else 0)
            if exp is not None and exp.n:
                assert np.array_equal(got.pos, exp.pos) and np.array_equal(got.cigar, exp.cigar) and np.array_equal(got.qual, exp.qual)


@pytest.mark.parametrize("thr", [0, 20, 41])
@pytest.mark.parametrize("block_every", [None, 5])
def test_the_packed_reader_takes_the_base_quality_test_at_parse(tmp_path, thr, block_every):
    """dut_bam_read_contig_bits: bit qual_off[i] + k <-> (quality value k of read i >= thr), pass_sum[i] = the sum of the
    passing values over the read's M/=/X bases; everything else as dut_bam_read_contig returns it."""
    refs, per = _dataset()
    per[1] = synth.long_read_contig(5_000, 8, 9)            # I / D / S operations, strings longer than a word
    path = str(tmp_path / "p.bam")
    write_bam(path, refs, per, block_every=block_every)
    with BamReader(path) as r:
        for tid in (0, 2, 1, 3, 0):
            want = r.fetch_contig(tid)
            got, bits, sums = r.fetch_contig_bits(tid, thr)
            for f in ("pos", "flag", "mapq", "cigar_off", "cigar", "qual_off", "qname_off", "qname"):
                assert np.array_equal(getattr(got, f), getattr(want, f)), f
            nq = int(want.qual_off[-1])
            ok = want.qual[:nq] >= thr
            have = np.unpackbits(bits.view(np.uint8), bitorder="little")[:nq].astype(bool)
            assert np.array_equal(have, ok)
            assert not np.unpackbits(bits.view(np.uint8), bitorder="little")[nq:].any()       # nothing behind the last value
            for i in range(want.n):
                q = want.qual[int(want.qual_off[i]):int(want.qual_off[i + 1])].astype(np.int64)
                y = 0; tot = 0
                for cw in want.cigar[int(want.cigar_off[i]):int(want.cigar_off[i + 1])]:
                    op, ln = int(cw) & 15, int(cw) >> 4
                    if op in (0, 7, 8):
                        seg = q[y:y + ln]; tot += int(seg[seg >= thr].sum())
                    if op in (0, 1, 4, 7, 8):
                        y += ln
                assert int(sums[i]) == tot, (tid, i)

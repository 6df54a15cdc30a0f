"""The host half of the pass-bit form, without a GPU: the base-quality test taken in cl_push_reads (qual_pack.cpp: one
bit per base, scalar / SSE2 / AVX2) and the rows the upload builds from the bits (pass_rows.h), through a context that
has no device (cl_debug_host_create).  What the rows must say is the reference's own rule (mod.rs:30-37): the number of
rows with bit p set in a window = the number of reads with mapq >= min_mapping_quality that have an M/=/X base at W + p
whose quality byte passes min_base_quality -- taken here from the numpy restatement oracle/bruteforce.py (the checker),
together with summed_baseq (contig_profiler.rs:65-70)."""
import ctypes as C

import numpy as np
import pytest

from decodingustools_amd import CallableOptions, _lib, synth
from decodingustools_amd.callable_loci import EngineError, HostStage
from decodingustools_amd.records import ContigRecords
from oracle import bruteforce

from helpers import make_options

T = 2048


def pack(qual, thr, level):
    lib = _lib.load()
    q = np.ascontiguousarray(qual, np.uint8)
    words = np.zeros((q.shape[0] + 63) // 64 + 1, np.uint64)
    s = C.c_uint64()
    assert lib.cl_debug_qual_pack(q.ctypes.data if q.shape[0] else None, q.shape[0], thr, level, words.ctypes.data, C.byref(s)) == 0
    return words[:(q.shape[0] + 63) // 64], int(s.value)


@pytest.mark.parametrize("thr", [0, 1, 2, 20, 37, 127, 128, 129, 200, 255])
def test_pass_bits_and_sums_agree_at_every_level(thr):
    rng = np.random.default_rng(1000 + thr)
    for n in (0, 1, 15, 16, 31, 32, 33, 63, 64, 65, 150, 1000, 4097):
        q = rng.integers(0, 256, size=n, dtype=np.uint8)
        if n > 20:
            q[3:9] = thr                                        # the boundary value itself
            q[10] = 0xFF                                        # absent qualities pass every threshold
        want_bits = q >= thr
        want_words = np.zeros((n + 63) // 64, np.uint64)
        for i in np.flatnonzero(want_bits):
            want_words[i // 64] |= np.uint64(1) << np.uint64(i % 64)
        want_sum = int(q[want_bits].astype(np.int64).sum())
        for level in (0, 1, 2, 10, 11, 12):
            words, s = pack(q, thr, level)
            assert np.array_equal(words, want_words), (n, level)
            assert s == want_sum, (n, level)


def test_reference_n_bits_agree_at_every_level():
    """cl_contig_upload sends one bit of a reference base in the pass-bit form: is it 'N' / 'n' (mod.rs:100-101), and 1
    for every position beyond the reference (mod.rs:79-80)."""
    lib = _lib.load()
    rng = np.random.default_rng(5)
    for n in (0, 1, 63, 64, 65, 200, 4096, 5000):
        ref = rng.choice(np.frombuffer(b"ACGTacgtNnRYMK.*-", np.uint8), size=n)
        for n_words in ((n + 63) // 64, (n + 63) // 64 + 3, max(0, n // 64 - 1)):
            want = np.ones(n_words * 64, bool)
            k = min(n, n_words * 64)
            want[:k] = (ref[:k] | 0x20) == ord("n")
            for level in (0, 1, 2):
                out = np.full(n_words + 1, 0xABCDABCDABCDABCD, np.uint64)
                assert lib.cl_debug_ref_n_bits(ref.ctypes.data if n else None, n, n_words, level, out.ctypes.data) == 0
                assert out[n_words] == 0xABCDABCDABCDABCD                     # nothing behind the words asked for
                have = np.unpackbits(out[:n_words].view(np.uint8), bitorder="little").astype(bool)
                assert np.array_equal(have, want), (n, n_words, level)


def stage(opt: CallableOptions, contig_len, rec: ContigRecords, tiles):
    """The accepted reads of `rec` pushed in the given tiles ([(first, last)) read ranges) -> (n_groups, rows, sum_q)."""
    with HostStage(opt) as st:
        st.contig_begin(0, contig_len, None)
        for a, b in tiles:
            co = rec.cigar_off[a:b + 1]; qo = rec.qual_off[a:b + 1]
            st.push_reads(rec.pos[a:b], rec.mapq[a:b], co, rec.cigar, qo, rec.qual)
        return st.pass_rows()


def packed(rec: ContigRecords, thr):
    """What a caller of cl_push_reads_bits hands over: one bit per quality value (bit qual_off[i] + k) and per read the sum
    of the passing values over its M/=/X bases (numpy restatement of the header's wording)."""
    nq = int(rec.qual_off[-1])
    ok = rec.qual[:nq] >= thr
    bits = np.zeros((nq + 63) // 64 + 1, np.uint64)
    by = np.packbits(ok, bitorder="little")
    bits.view(np.uint8)[:by.shape[0]] = by
    sums = np.zeros(rec.n, np.uint32)
    for i in range(rec.n):
        q = rec.qual[int(rec.qual_off[i]):int(rec.qual_off[i + 1])].astype(np.int64)
        y = 0; tot = 0
        for cw in rec.cigar[int(rec.cigar_off[i]):int(rec.cigar_off[i + 1])]:
            op, ln = int(cw) & 15, int(cw) >> 4
            if op in (0, 7, 8):
                seg = q[y:y + ln]
                tot += int(seg[seg >= thr].sum())
            if op in (0, 1, 4, 7, 8):
                y += ln
        sums[i] = tot
    return bits, sums


def stage_packed(opt: CallableOptions, contig_len, rec: ContigRecords, tiles, mixed=False):
    """The same tiles through cl_push_reads_bits (every second one through cl_push_reads when `mixed`)."""
    bits, sums = packed(rec, opt.min_base_quality)
    with HostStage(opt) as st:
        st.contig_begin(0, contig_len, None)
        for k, (a, b) in enumerate(tiles):
            co = rec.cigar_off[a:b + 1]; qo = rec.qual_off[a:b + 1]
            if mixed and k % 2:
                st.push_reads(rec.pos[a:b], rec.mapq[a:b], co, rec.cigar, qo, rec.qual)
            else:
                st.push_reads_bits(rec.pos[a:b], rec.mapq[a:b], co, rec.cigar, qo, bits, sums[a:b])
        return st.pass_rows()


def column_sums(ng, rows, extent):
    """Per position: the number of rows of its window with the position's bit set; also checks the layout's bounds."""
    n_win = (extent + T - 1) // T
    assert ng.shape[0] == n_win
    assert rows.shape[0] == int(ng.sum()) * 256
    qc = np.zeros(n_win * T, np.int64)
    off = 0
    for w in range(n_win):
        g = rows[off * 256:(off + int(ng[w])) * 256].reshape(int(ng[w]), 64, 4)     # [group][block][row & 3]
        off += int(ng[w])
        if g.shape[0] == 0:
            continue
        bits = np.unpackbits(g.view(np.uint8).reshape(g.shape[0], 64, 4, 4), axis=-1, bitorder="little")   # [g][block][r][32]
        qc[w * T:(w + 1) * T] = bits.reshape(g.shape[0], 64, 4, 32).sum(axis=(0, 2)).reshape(T)
    return qc[:extent], n_win


def accepted(rec: ContigRecords, contig_len):
    """What the host driver would push: no FUNMAP reads, pos < len (the cap never bites in these inputs)."""
    keep = ((rec.flag & 4) == 0) & (rec.pos < contig_len)
    if keep.all():
        return rec
    idx = np.flatnonzero(keep)
    cig = [rec.cigar[int(rec.cigar_off[i]):int(rec.cigar_off[i + 1])] for i in idx]
    qual = [rec.qual[int(rec.qual_off[i]):int(rec.qual_off[i + 1])] for i in idx]
    names = [rec.qname[int(rec.qname_off[i]):int(rec.qname_off[i + 1])] for i in idx]
    cat = lambda parts, dt: np.concatenate(parts).astype(dt) if parts else np.zeros(0, dt)
    off = lambda parts, dt: np.concatenate([[0], np.cumsum([p.shape[0] for p in parts])]).astype(dt)
    return ContigRecords(pos=np.ascontiguousarray(rec.pos[idx]), flag=np.ascontiguousarray(rec.flag[idx]),
                         mapq=np.ascontiguousarray(rec.mapq[idx]), cigar_off=off(cig, np.uint32), cigar=cat(cig, np.uint32),
                         qual_off=off(qual, np.uint64), qual=cat(qual, np.uint8), qname_off=off(names, np.uint32),
                         qname=cat(names, np.uint8)).validate()


def check(rec, L, opt_d, tiles=None, seed=0):
    opt_ns = make_options(opt_d)
    opt = CallableOptions(**{k: getattr(opt_ns, k) for k in ("min_depth", "max_depth", "min_mapping_quality", "min_base_quality",
                                                             "min_depth_for_low_mapq", "max_low_mapq", "max_low_mapq_fraction")})
    rec = accepted(rec, L)
    want = bruteforce.contig(opt_ns, "c", L, None, rec)
    if tiles is None:
        rng = np.random.default_rng(seed)
        cuts = sorted(set(int(x) for x in rng.integers(0, rec.n + 1, size=int(rng.integers(0, 5)))) | {0, rec.n})
        tiles = list(zip(cuts[:-1], cuts[1:]))
    ng, rows, sum_q = stage(opt, L, rec, tiles)
    qc, n_win = column_sums(ng, rows, want["extent"])
    bad = np.flatnonzero(qc != want["qc"])
    assert bad.shape[0] == 0, (bad[:10], qc[bad[:10]], want["qc"][bad[:10]])
    assert sum_q == want["summed_baseq"]
    assert int(qc.sum()) == want["quality_bases"]
    # the packed variant (cl_push_reads_bits) stages the very same rows, also mixed with byte tiles
    for mixed in (False, True):
        ng2, rows2, sum_q2 = stage_packed(opt, L, rec, tiles, mixed)
        assert np.array_equal(ng2, ng) and np.array_equal(rows2, rows) and sum_q2 == sum_q
    # as many rows as the deepest column of reads that enter the rows needs, rounded up to a group of 4
    return ng


OPTS = [None, dict(min_mapping_quality=0, min_base_quality=0), dict(min_mapping_quality=30, min_base_quality=37),
        dict(min_base_quality=255), dict(min_mapping_quality=255)]


@pytest.mark.parametrize("seed", range(8))
def test_rows_of_adversarial_contigs(seed):
    L = 9000 + 700 * seed
    rec = synth.adversarial_contig(L, 900, 7000 + seed)
    check(rec, L, OPTS[seed % len(OPTS)], seed=seed)


def test_rows_of_the_short_read_generator_in_several_tiles():
    L = 300_000
    rec = synth.short_read_contig(L, 30, synth.seed_for(2, 21))
    ng = check(rec, L, None, seed=5)
    assert 4 <= ng.max() <= 40                                   # 30x: some 30-45 rows in the deepest window


def test_rows_of_long_reads_enter_through_checkpoints():
    L = 120_000
    rec = synth.long_read_contig(L, 12, synth.seed_for(3, 24))
    check(rec, L, None, seed=9)
    check(rec, L, dict(min_base_quality=23), tiles=[(0, accepted(rec, L).n)])


def test_rows_with_truncated_and_absent_qualities_and_zero_span_reads():
    reads = [
        (10, "100M", 60, [30] * 40),                             # 60 bases without a quality byte
        (20, "50M", 60, None),                                   # no quality string at all: nothing passes
        (30, "20S", 60, [40] * 20),                              # no reference span: in no column
        (40, "10M5I10M4D10M", 60, [0xFF] * 35),                  # absent qualities pass
        (2040, "30M", 60, [25] * 30),                            # across the window seam
        (2047, "30M", 9, [25] * 30),                             # below min_mapping_quality: not in the rows
        (4095, "1M", 60, [25]),
    ]
    rec = ContigRecords.from_reads(reads)
    check(rec, 4096, None, tiles=[(0, rec.n)])
    check(rec, 4096, None, tiles=[(0, 3), (3, 4), (4, rec.n)])
    check(rec, 4090, None, tiles=[(0, rec.n)])                   # the last read overhangs the contig: the extent grows


def test_tiles_that_start_and_end_inside_a_word_of_the_bit_array():
    rng = np.random.default_rng(77)
    reads = []
    p = 0
    for i in range(400):
        ln = int(rng.integers(1, 70))                            # quality strings of 1..69 bytes: tiles end mid-word
        p += int(rng.integers(0, 9))
        reads.append((p, "%dM" % ln, 60, [int(x) for x in rng.integers(0, 60, size=ln)]))
    rec = ContigRecords.from_reads(reads)
    L = p + 200
    for cuts in ([0, 400], [0, 1, 2, 3, 400], list(range(0, 401, 7)) + [400], [0, 399, 400]):
        check(rec, L, None, tiles=list(zip(cuts[:-1], cuts[1:])))


def test_a_refused_tile_leaves_the_bits_as_they_were():
    opt = CallableOptions()
    reads = [(i * 3, "50M", 60, [10 + (i * 7) % 40] * 50) for i in range(60)]
    rec = ContigRecords.from_reads(reads)
    want = bruteforce.contig(make_options(None), "c", 4000, None, rec)
    with HostStage(opt) as st:
        st.contig_begin(0, 4000, None)
        st.push_reads(rec.pos[:25], rec.mapq[:25], rec.cigar_off[:26], rec.cigar, rec.qual_off[:26], rec.qual)
        bad_pos = rec.pos[25:40].copy(); bad_pos[5] = 0                          # unsorted: refused
        with pytest.raises(EngineError):
            st.push_reads(bad_pos, rec.mapq[25:40], rec.cigar_off[25:41], rec.cigar, rec.qual_off[25:41], rec.qual)
        st.push_reads(rec.pos[25:], rec.mapq[25:], rec.cigar_off[25:], rec.cigar, rec.qual_off[25:], rec.qual)
        with pytest.raises(EngineError):                                          # a refused tile as the last one, too
            st.push_reads(bad_pos, rec.mapq[25:40], rec.cigar_off[25:41], rec.cigar, rec.qual_off[25:41], rec.qual)
        ng, rows, sum_q = st.pass_rows()
        qc, _ = column_sums(ng, rows, want["extent"])
        assert np.array_equal(qc, want["qc"]) and sum_q == want["summed_baseq"]
        with pytest.raises(EngineError) as e:                                    # no device, no pileup
            st.contig_upload()
        assert e.value.status == -2


def test_a_window_deeper_than_its_first_buffer_of_groups():
    # 300 reads over one stretch: 75 groups, more than the 16 the hook offers first (the builder says so and is asked again)
    reads = [(100 + (i % 3), "120M", 60, [30] * 120) for i in range(300)]
    reads.sort(key=lambda r: r[0])
    rec = ContigRecords.from_reads(reads)
    ng = check(rec, 3000, dict(max_depth=0), tiles=[(0, rec.n)])
    assert ng[0] == 75


def test_rows_of_gapped_reads_and_every_cigar_shape():
    """Reads whose span dwarfs their query (long N gaps) keep query-order bits and are walked per window; every other
    shape is mapped to reference order at push: S / I / H at the start, runs behind D, I and N, '=' / 'X', more than 64
    operations, reads that end on a window seam, a deletion across a seam, absent and truncated quality strings."""
    rng = np.random.default_rng(31)
    shapes = ["5S145M", "3I147M", "2S3I95M50M", "10H140M", "75M2D75M", "60M1I30M4D59M", "40M300N60M50S", "1M1D1M1D148M",
              "20=5X30=1I10X2D84=", "150S", "4I", "30M5000N30M", "7M1I" * 40, "2M1D" * 70 + "10M", "148M2S", "1S1M1S",
              "100M70000N100M", "10M2040N10M", "8M3D9M", "64M", "50M9000N1M"]
    reads = []
    for i in range(900):
        cig = shapes[i % len(shapes)]
        p = int(rng.integers(0, 100_000))
        if cig == "8M3D9M":
            p = 2040 + 2048 * int(rng.integers(0, 40))               # the deletion spans a window seam
        if cig == "64M":
            p = 2048 * int(rng.integers(1, 40)) - 64                  # ends exactly on a seam
        from decodingustools_amd.records import cigar_from_string, cigar_query_length
        ql = cigar_query_length(cigar_from_string(cig))
        quals = [int(x) for x in rng.choice([5, 19, 20, 35, 255], size=ql)]
        if i % 7 == 0:
            quals = quals[:ql // 3]                                   # truncated
        if i % 31 == 0:
            quals = None
        reads.append((p, cig, int(rng.choice([0, 9, 10, 60, 60])), quals))
    reads.sort(key=lambda r: r[0])
    rec = ContigRecords.from_reads(reads)
    L = 190_000
    for o in (None, dict(min_mapping_quality=0, min_base_quality=0), dict(min_base_quality=36)):
        check(rec, L, o, seed=3)
        check(rec, L, o, tiles=[(0, rec.n)])

"""Seeded synthetic inputs for the coverage path: references with N runs and
decoded alignment records in the ContigRecords layout.  Used by tests and bench.py; there is no
network, so there are no real BAMs to read.

PRNG: numpy Generator(PCG64(seed)), seed = 0x5EED0000 + config*256 + tid.
"""
import numpy as np

from .records import ContigRecords, pack_seq4

HG38_PRIMARY = [
    ("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555),
    ("chr5", 181538259), ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636),
    ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
    ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345),
    ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
    ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415),
    ("chrM", 16569),
]
HG38_LEN = dict(HG38_PRIMARY)


def seed_for(config, tid):
    return 0x5EED0000 + config * 256 + tid


def _u8_table(values, weights):
    """256-entry lookup table that maps a uniform byte to `values` with ~`weights`."""
    w = np.asarray(weights, dtype=np.float64)
    edges = np.floor(np.cumsum(w) / w.sum() * 256 + 0.5).astype(int)
    edges[-1] = 256
    cnt = np.diff(np.concatenate([[0], edges]))
    return np.repeat(np.asarray(values, dtype=np.uint8), cnt)


def make_reference(L, seed, lowercase=False):
    """i.i.d. ACGT with N runs: first/last 10 kb of contigs > 1 Mb, one centromere-like run of 3 %
    of L at 40 % of L, 20 short runs (50..5000 bp, scaled down for small contigs).  With
    lowercase=True one run of 'n' and one soft-masked acgt stretch are added (chrM config)."""
    rng = np.random.default_rng(seed ^ 0xA5A5)
    ref = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=L, dtype=np.uint8)].copy()
    if L > 1_000_000:
        ref[:10_000] = ord("N")
        ref[-10_000:] = ord("N")
    if L >= 1000:
        c0 = int(L * 0.40)
        ref[c0:c0 + int(L * 0.03)] = ord("N")
        for _ in range(20):
            ln = int(rng.integers(50, 5001))
            ln = max(1, min(ln, L // 200))
            s = int(rng.integers(0, L - ln))
            ref[s:s + ln] = ord("N")
    if lowercase and L >= 200:
        s = L // 5
        ref[s:s + 37] = ord("n")
        s2 = L // 2 + L // 7
        seg = ref[s2:s2 + 120]
        keep = seg != ord("N")
        seg[keep] = seg[keep] | 0x20          # acgt: soft-masked, must NOT become REF_N
    return ref


def _names_fixed(ids, width=10):
    """fixed-width base-32 names 'Pxxxxxxxxx' for integer ids (vectorised)."""
    alphabet = np.frombuffer(b"0123456789ABCDEFGHJKMNPQRSTVWXYZ", dtype=np.uint8)
    ids = np.asarray(ids, dtype=np.uint64)
    out = np.empty((ids.shape[0], width), dtype=np.uint8)
    out[:, 0] = ord("P")
    v = ids.copy()
    for j in range(width - 1, 0, -1):
        out[:, j] = alphabet[(v & np.uint64(31)).astype(np.int64)]
        v >>= np.uint64(5)
    return out


def _zones(rng, L, frac, lo, hi):
    """non-overlapping-ish runs covering about `frac` of L; returns (starts, ends)."""
    starts, ends, tot = [], [], 0
    target = int(L * frac)
    guard = 0
    while tot < target and guard < 10000:
        guard += 1
        ln = int(rng.integers(lo, hi + 1))
        ln = min(ln, max(1, L // 50))
        s = int(rng.integers(0, max(1, L - ln)))
        starts.append(s); ends.append(s + ln); tot += ln
    return np.asarray(starts, dtype=np.int64), np.asarray(ends, dtype=np.int64)


def _in_zones(pos, zs, ze, L):
    if zs.shape[0] == 0:
        return np.zeros(pos.shape[0], dtype=bool)
    d = np.zeros(L + 1, dtype=np.int32)
    np.add.at(d, zs, 1)
    np.add.at(d, np.minimum(ze, L), -1)
    inz = np.cumsum(d[:-1]) > 0
    return inz[pos]


def short_read_contig(L, depth, seed, read_len=150, with_seq=False, max_live_assert=400, ref=None):
    """Config 2/4/5 read model: paired 2x150, insert ~N(400,50) clipped [200,800];
    CIGAR 96 % 150M, 2 % one small D or I (1-5 bp), 1.5 % soft-clipped end (5-40S), 0.5 % =/X;
    MAPQ 85 % 60, 10 % uniform 2..59, 5 % 0-1, repeat zones (2 % of L) with 90 % MAPQ 0;
    base quality {2: 2 %, 12: 5 %, 23: 13 %, 37: 80 %}; 1 % of L zero coverage, 2 % at 2-3x;
    flags 1 % 0x400, 0.5 % 0x100, 0.5 % 0x800, 0.2 % 0x200, 0.2 % 0x4."""
    rng = np.random.default_rng(seed)
    rl = read_len
    n_pairs = int(L * depth / (2 * rl) * 1.04) + 1      # ~4 % over-sample, thinned by the zones
    ins = np.clip(np.rint(rng.normal(400, 50, n_pairs)), 200, 800).astype(np.int64)
    ins = np.maximum(ins, rl)
    fs = rng.integers(0, max(1, L - 800), size=n_pairs, dtype=np.int64)
    # depth modulation by fragment start
    z0s, z0e = _zones(rng, L, 0.01, 500, 20000)
    z1s, z1e = _zones(rng, L, 0.02, 500, 20000)
    in0 = _in_zones(fs, z0s, z0e, L)
    in1 = _in_zones(fs, z1s, z1e, L)
    keep = ~in0 & (~in1 | (rng.random(n_pairs) < 2.5 / depth)) & (rng.random(n_pairs) < 1 / 1.04 + in1 * 1.0)
    fs, ins = fs[keep], ins[keep]
    n_pairs = fs.shape[0]
    pid = np.arange(n_pairs, dtype=np.uint64)
    pos = np.concatenate([fs, fs + ins - rl])
    pos = np.minimum(pos, L - 1)
    pair = np.concatenate([pid, pid])
    order = np.argsort(pos, kind="stable")
    pos, pair = pos[order], pair[order]
    n = pos.shape[0]
    # CIGAR
    t = rng.random(n)
    kind = np.zeros(n, dtype=np.int8)                       # 0: 150M
    kind[t >= 0.96] = 1                                      # D
    kind[t >= 0.97] = 2                                      # I
    kind[t >= 0.98] = 3                                      # leading S
    kind[t >= 0.9875] = 4                                    # trailing S
    kind[t >= 0.995] = 5                                     # = X =
    a = rng.integers(10, rl - 20, size=n)                    # first block length
    b = rng.integers(1, 6, size=n)                           # indel length
    s = rng.integers(5, 41, size=n)                          # soft clip length
    M, I, D, S, EQ, X = 0, 1, 2, 4, 7, 8
    w = np.zeros((n, 3), dtype=np.uint32)
    valid = np.zeros((n, 3), dtype=bool)

    def enc(length, op):
        return (length.astype(np.uint32) << np.uint32(4)) | np.uint32(op)

    full = np.full(n, rl)
    k0 = kind == 0
    w[k0, 0] = enc(full[k0], M); valid[k0, 0] = True
    k = kind == 1
    w[k, 0] = enc(a[k], M); w[k, 1] = enc(b[k], D); w[k, 2] = enc(rl - a[k], M); valid[k] = True
    k = kind == 2
    w[k, 0] = enc(a[k], M); w[k, 1] = enc(b[k], I); w[k, 2] = enc(rl - a[k] - b[k], M); valid[k] = True
    k = kind == 3
    w[k, 0] = enc(s[k], S); w[k, 1] = enc(rl - s[k], M); valid[k, :2] = True
    k = kind == 4
    w[k, 0] = enc(rl - s[k], M); w[k, 1] = enc(s[k], S); valid[k, :2] = True
    k = kind == 5
    w[k, 0] = enc(a[k], EQ); w[k, 1] = enc(np.ones(n, dtype=np.int64)[k], X); w[k, 2] = enc(rl - a[k] - 1, EQ); valid[k] = True
    cigar = w[valid]
    cigar_off = np.concatenate([[0], np.cumsum(valid.sum(axis=1))]).astype(np.uint32)
    reflen = np.full(n, rl, dtype=np.int64)
    reflen[kind == 1] += b[kind == 1]
    reflen[kind == 2] -= b[kind == 2]
    reflen[kind == 3] -= s[kind == 3]
    reflen[kind == 4] -= s[kind == 4]
    # keep reads inside the contig: clip the start so that pos + reflen <= L
    pos = np.minimum(pos, L - reflen - 1)
    pos = np.maximum(pos, 0)
    order = np.argsort(pos, kind="stable")
    if not np.array_equal(order, np.arange(n)):
        # re-sort every per-read array consistently
        pos, pair, kind, reflen = pos[order], pair[order], kind[order], reflen[order]
        w, valid = w[order], valid[order]
        cigar = w[valid]
        cigar_off = np.concatenate([[0], np.cumsum(valid.sum(axis=1))]).astype(np.uint32)
    # MAPQ
    u = rng.random(n)
    mapq = np.full(n, 60, dtype=np.uint8)
    mid = (u >= 0.85) & (u < 0.95)
    mapq[mid] = rng.integers(2, 60, size=int(mid.sum()), dtype=np.uint8)
    lowm = u >= 0.95
    mapq[lowm] = rng.integers(0, 2, size=int(lowm.sum()), dtype=np.uint8)
    rzs, rze = _zones(rng, L, 0.02, 1000, 50000)
    inrep = _in_zones(pos, rzs, rze, L) & (rng.random(n) < 0.9)
    mapq[inrep] = 0
    # flags
    f = rng.random(n)
    flag = np.zeros(n, dtype=np.uint16)
    flag[f < 0.010] = 0x400
    flag[(f >= 0.010) & (f < 0.015)] = 0x100
    flag[(f >= 0.015) & (f < 0.020)] = 0x800
    flag[(f >= 0.020) & (f < 0.022)] = 0x200
    flag[(f >= 0.022) & (f < 0.024)] = 0x4
    # base qualities
    tab = _u8_table([2, 12, 23, 37], [0.02, 0.05, 0.13, 0.80])
    qual = tab[rng.integers(0, 256, size=n * rl, dtype=np.uint8)]
    qual_off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(rl))
    names = _names_fixed(pair)
    rec = ContigRecords(
        pos=pos.astype(np.int32), flag=flag, mapq=mapq, cigar_off=cigar_off,
        cigar=np.ascontiguousarray(cigar, dtype=np.uint32), qual_off=qual_off, qual=qual,
        qname_off=(np.arange(n + 1, dtype=np.uint32) * np.uint32(names.shape[1])),
        qname=np.ascontiguousarray(names.reshape(-1)))
    if with_seq:
        if ref is not None:
            # read bases follow the reference (indels ignored), 3 % random substitutions
            code_of = np.full(256, 15, dtype=np.uint8)
            for ch, cd in zip(b"ACGTacgt", [1, 2, 4, 8, 1, 2, 4, 8]):
                code_of[ch] = cd
            idx = np.minimum(pos[:, None] + np.arange(rl)[None, :], L - 1)
            codes = code_of[ref[idx]].reshape(-1)
            mut = rng.random(codes.shape[0]) < 0.03
            codes[mut] = np.asarray([1, 2, 4, 8], dtype=np.uint8)[rng.integers(0, 4, size=int(mut.sum()))]
        else:
            codes = np.asarray([1, 2, 4, 8], dtype=np.uint8)[rng.integers(0, 4, size=n * rl, dtype=np.uint8)]
        rec.seq_off = qual_off.copy()
        rec.seq4 = pack_seq4(codes)
    if max_live_assert:
        d = np.zeros(L + 2, dtype=np.int32)
        np.add.at(d, pos, 1)
        np.add.at(d, np.minimum(pos + reflen, L + 1), -1)
        mx = int(np.cumsum(d).max()) if n else 0
        assert mx < max_live_assert, f"synthetic depth {mx} would trigger the pileup depth cap"
    return rec.validate()


def long_read_contig(L, depth, seed, median_len=10_000, max_reads=None):
    """Config 3 read model: read length lognormal (median 10 kb, sigma 0.5, clipped 1-60 kb, scaled
    down with L); CIGAR alternates M runs (geometric, mean 15) with I/D (50/50, geometric mean 2,
    occasional 50-500 bp D), optional leading/trailing S; MAPQ 90 % 60; base quality ~Q22."""
    rng = np.random.default_rng(seed)
    med = min(median_len, max(200, L // 8))
    n = int(L * depth / (med * 1.13)) + 1
    if max_reads:
        n = min(n, max_reads)
    pos = np.sort(rng.integers(0, max(1, L - med // 2), size=n)).astype(np.int64)
    tab = _u8_table(list(range(4, 44, 2)), np.exp(-0.5 * ((np.arange(4, 44, 2) - 22) / 8.0) ** 2))
    cig_all, coff, quals, qoff, mapq = [], [0], [], [0], []
    for i in range(n):
        tl = int(np.clip(rng.lognormal(np.log(med), 0.5), med // 10, med * 6))
        tl = min(tl, L - int(pos[i]) - 1)
        if tl < 20:
            tl = 20
            pos[i] = max(0, L - 40)
        n_runs = tl // 12 + 2
        m = rng.geometric(1 / 15.0, size=n_runs)
        g = rng.geometric(1 / 2.0, size=n_runs)
        isdel = rng.random(n_runs) < 0.5
        big = rng.random(n_runs) < 0.002
        g = np.where(big & isdel, rng.integers(50, 501, size=n_runs), g)
        refadv = m + np.where(isdel, g, 0)
        cut = min(int(np.searchsorted(np.cumsum(refadv), tl)) + 1, n_runs)
        m, g, isdel = m[:cut], g[:cut], isdel[:cut]
        ops = np.empty(2 * cut, dtype=np.uint32)
        ops[0::2] = (m.astype(np.uint32) << 4) | 0
        ops[1::2] = (g.astype(np.uint32) << 4) | np.where(isdel, 2, 1).astype(np.uint32)
        ops = ops[:-1]                                    # end on an M run
        lead = int(rng.integers(0, 3))
        pre = [(int(rng.integers(5, 200)) << 4) | 4] if lead == 1 else []
        post = [(int(rng.integers(5, 200)) << 4) | 4] if lead == 2 else []
        ops = np.concatenate([np.asarray(pre, dtype=np.uint32), ops, np.asarray(post, dtype=np.uint32)])
        # never overhang the contig: trim the last M run if needed
        code = ops & 15
        ln = (ops >> 4).astype(np.int64)
        rlen = int(ln[np.isin(code, [0, 2, 3, 7, 8])].sum())
        over = int(pos[i]) + rlen - L
        if over > 0:
            pos[i] = max(0, int(pos[i]) - over)
        qlen = int(ln[np.isin(code, [0, 1, 4, 7, 8])].sum())
        cig_all.append(ops); coff.append(coff[-1] + ops.shape[0])
        quals.append(tab[rng.integers(0, 256, size=qlen, dtype=np.uint8)]); qoff.append(qoff[-1] + qlen)
        mapq.append(60 if rng.random() < 0.9 else int(rng.integers(0, 60)))
    order = np.argsort(pos, kind="stable")
    pos = pos[order]
    cig_all = [cig_all[j] for j in order]; quals = [quals[j] for j in order]
    mapq = np.asarray(mapq, dtype=np.uint8)[order]
    coff = np.concatenate([[0], np.cumsum([c.shape[0] for c in cig_all])]).astype(np.uint32)
    qoff = np.concatenate([[0], np.cumsum([q.shape[0] for q in quals])]).astype(np.uint64)
    names = _names_fixed(np.arange(n, dtype=np.uint64))
    return ContigRecords(
        pos=pos.astype(np.int32), flag=np.zeros(n, dtype=np.uint16), mapq=mapq, cigar_off=coff,
        cigar=np.ascontiguousarray(np.concatenate(cig_all) if n else np.zeros(0, np.uint32), dtype=np.uint32),
        qual_off=qoff, qual=np.ascontiguousarray(np.concatenate(quals) if n else np.zeros(0, np.uint8)),
        qname_off=(np.arange(n + 1, dtype=np.uint32) * np.uint32(names.shape[1])),
        qname=np.ascontiguousarray(names.reshape(-1))).validate()


def adversarial_contig(L, n_reads, seed, max_len=300, same_start_bursts=True, deep=False,
                       overhang=False):
    """Small contigs that exercise every edge the reference's semantics has: all CIGAR op kinds
    (M I D N S H P = X), leading/trailing clips, reads whose CIGAR consumes no reference, reads with
    l_seq = 0 or 0xFF qualities, qualities around the threshold, every flag bit incl. 0x4, bursts of
    reads at one start position (depth-cap rule), mate pairs sharing a name, optional very deep
    pile-ups (> 255, exercises the 32-bit counter path) and reads overhanging the contig end."""
    rng = np.random.default_rng(seed)
    reads = []
    starts = np.sort(rng.integers(0, L, size=n_reads))
    if same_start_bursts and n_reads > 20:
        for _ in range(max(1, n_reads // 40)):
            j = int(rng.integers(0, n_reads - 12))
            starts[j:j + 12] = starts[j]
        starts = np.sort(starts)
    if deep and n_reads > 1:
        j = n_reads // 3
        cnt = min(n_reads - j - 1, 400)
        starts[j:j + cnt] = starts[j] + rng.integers(0, 30, size=cnt)
        starts = np.sort(starts)
    for i in range(n_reads):
        p = int(starts[i])
        style = rng.random()
        tl = int(rng.integers(1, max_len))
        ops = []
        if style < 0.08:
            ops = [("S", int(rng.integers(1, 30))), ("I", int(rng.integers(1, 5)))]      # no reference span
        elif style < 0.40:
            ops = [("M", tl)]
        else:
            if rng.random() < 0.2: ops.append(("H", int(rng.integers(1, 20))))
            if rng.random() < 0.4: ops.append(("S", int(rng.integers(1, 25))))
            nseg = int(rng.integers(1, 8))
            for sgi in range(nseg):
                ops.append((rng.choice(["M", "=", "X", "M"]), int(rng.integers(1, max(2, tl // nseg + 1)))))
                if sgi + 1 < nseg:
                    g = rng.random()
                    if g < 0.3: ops.append(("I", int(rng.integers(1, 6))))
                    elif g < 0.6: ops.append(("D", int(rng.integers(1, 12))))
                    elif g < 0.75: ops.append(("N", int(rng.integers(1, 60))))
                    elif g < 0.8: ops.append(("P", int(rng.integers(1, 3))))
                    elif g < 0.9:
                        ops.append(("I", int(rng.integers(1, 4)))); ops.append(("D", int(rng.integers(1, 4))))
            if rng.random() < 0.4: ops.append(("S", int(rng.integers(1, 25))))
            if rng.random() < 0.1: ops.append(("H", int(rng.integers(1, 20))))
        rlen = sum(l for o, l in ops if o in "MDN=X")
        if not overhang and p + rlen > L:
            p = max(0, L - rlen)
            if p + rlen > L:
                ops = [("M", max(1, L - p))]
        cig = "".join(f"{l}{o}" for o, l in ops)
        qlen = sum(l for o, l in ops if o in "MIS=X")
        qs = rng.random()
        if qs < 0.05:
            q = None                                       # l_seq == 0
        elif qs < 0.10:
            q = [255] * qlen                               # qualities absent
        else:
            q = rng.choice([0, 2, 12, 19, 20, 21, 37, 41, 93, 127, 128, 200], size=qlen).tolist()
        mq = int(rng.choice([0, 1, 2, 9, 10, 11, 30, 60, 255]))
        fl = int(rng.choice([0, 0, 0, 0x1 | 0x40, 0x1 | 0x80, 0x10, 0x100, 0x200, 0x400, 0x800, 0x4, 0x4 | 0x1]))
        nm = f"q{i // 2}" if rng.random() < 0.5 else f"s{i}"
        reads.append((p, cig, mq, q, fl, nm))
    reads.sort(key=lambda r: r[0])
    return ContigRecords.from_reads(reads)

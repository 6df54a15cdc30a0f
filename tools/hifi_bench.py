#!/usr/bin/env python3
"""Tooling: a HiFi-like shape (reads of ~15 kb, kilobase-long match runs separated by 1-2 base indels)."""
import os, sys, time, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from decodingustools_amd import CallableOptions, CallableProfiler, ContigProfiler, Engine, process_single_contig, synth
from decodingustools_amd.records import ContigRecords
L = int(os.environ.get("KB_LEN", 30_000_000)); depth = float(os.environ.get("KB_DEPTH", 30)); mean_run = int(os.environ.get("KB_RUN", 800))
rng = np.random.default_rng(5)
rl = 15000
n = int(L * depth / rl)
pos = np.sort(rng.integers(0, L - rl - 100, size=n)).astype(np.int32)
n_runs = rng.integers(max(2, rl // mean_run // 2), rl // mean_run * 2 + 2, size=n)
tot_ops = int((2 * n_runs - 1).sum())
cig = np.empty(tot_ops, np.uint32); coff = np.zeros(n + 1, np.uint32); qlen = np.zeros(n, np.int64)
k = 0
for i in range(n):
    m = int(n_runs[i])
    runs = rng.multinomial(rl - m, np.ones(m) / m) + 1
    ops = np.empty(2 * m - 1, np.uint32)
    ops[0::2] = (runs.astype(np.uint32) << 4)
    ind = rng.integers(1, 3, size=m - 1).astype(np.uint32)
    kinds = rng.integers(1, 3, size=m - 1).astype(np.uint32)          # 1 = I, 2 = D
    ops[1::2] = (ind << 4) | kinds
    cig[k:k + 2 * m - 1] = ops; k += 2 * m - 1; coff[i + 1] = k
    qlen[i] = int(runs.sum() + ind[kinds == 1].sum())
qoff = np.concatenate([[0], np.cumsum(qlen)]).astype(np.uint64)
qual = rng.choice(np.array([12, 25, 35, 40], np.uint8), size=int(qoff[-1]))
names = synth._names_fixed(np.arange(n, dtype=np.uint64))
rec = ContigRecords(pos=pos, flag=np.zeros(n, np.uint16), mapq=np.full(n, 60, np.uint8), cigar_off=coff, cigar=cig, qual_off=qoff, qual=qual,
                    qname_off=(np.arange(n + 1, dtype=np.uint32) * np.uint32(names.shape[1])), qname=np.ascontiguousarray(names.reshape(-1))).validate()
ref = synth.make_reference(L, 6)
print("reads", n, "ops", tot_ops, "ops/read", round(tot_ops / n, 1), "bases", int(qoff[-1]), flush=True)
opt = CallableOptions(); eng = Engine(opt, 0)
counter = CallableProfiler(os.path.join(tempfile.mkdtemp(), "x.bed")); st = ContigProfiler("c", L)
process_single_contig(eng, counter, st, opt, 0, rec, ref); counter.close()
eng.set_profiling(True)
for _ in range(2): eng.contig_run()
eng.sync(); eng.reset_kernel_ms()
for _ in range(5): eng.contig_run()
eng.sync()
ms, nr = eng.kernel_ms(); tot = sum(ms.values()) / nr
print(json.dumps(dict(L=L, ms={k2: round(v / nr, 4) for k2, v in ms.items()}, gbase_s=round(L / tot / 1e6, 2))))

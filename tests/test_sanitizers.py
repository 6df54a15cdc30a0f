"""The host-only C++ (BAM / FASTA reader incl. its threaded decode, admission, BamStats, JSON writer, tree
parsing / scoring / report) under AddressSanitizer + UBSan and under ThreadSanitizer, on the CPU
(sanitizers run on the CPU build; the device engine is not part of these binaries)."""
import json
import os
import random
import subprocess

import numpy as np
import pytest

from bamio import write_bam, write_fasta
from decodingustools_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "decodingustools_amd", "csrc", f) for f in ("bam_io.cpp", "host_coverage.cpp", "report.cpp", "haplogroup.cpp", "qual_pack.cpp")]
DRIVER = os.path.join(ROOT, "tests", "native", "sanitize_host.cpp")


def _build(tmp_path, flags, tag):
    exe = str(tmp_path / f"sanitize_host_{tag}")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer"] + flags + [DRIVER] + SRC + ["-lz", "-ldl", "-lpthread", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        # only a toolchain without the sanitizer runtimes is a reason to skip; anything else (a symbol the driver's
        # stubs lack, a compile error) is a failure of this tree
        if any(m in r.stderr for m in ("cannot find -lasan", "cannot find -ltsan", "cannot find -lubsan", "unrecognized command-line option", "unrecognized argument")):
            pytest.skip("sanitizer runtimes not installed here: " + r.stderr[-300:])
        pytest.fail("the sanitizer build of the host sources failed:\n" + r.stderr[-3000:])
    return exe


def _inputs(tmp_path):
    import test_haplogroup as TH
    L = 40_000
    recs = {0: synth.short_read_contig(L, 25, 1, with_seq=True, ref=synth.make_reference(L, 2)), 2: synth.adversarial_contig(9000, 300, 3),
            3: synth.long_read_contig(30_000, 8, 4)}
    refs = [("chr1", L), ("chr2", 500), ("chrX", 9000), ("chrY", 30_000)]
    aligned = str(tmp_path / "aligned.bam"); straddle = str(tmp_path / "straddle.bam"); noidx = str(tmp_path / "noidx.bam")
    write_bam(aligned, refs, recs, block_every=60)                       # blocks cut at record boundaries
    write_bam(straddle, refs, recs, long_cigar_tag=True)                 # records straddle blocks, CG tags
    write_bam(noidx, refs, recs, write_index=False, block_every=13)
    fasta = str(tmp_path / "ref.fa")
    write_fasta(fasta, [(n, synth.make_reference(l, 40 + i)) for i, (n, l) in enumerate(refs)])
    rng = random.Random(4)
    tree = str(tmp_path / "tree.json")
    open(tree, "w").write(TH.ftdna_tree(rng, 300, [rng.randrange(100, 29_000) for _ in range(500)]))
    bad = str(tmp_path / "bad.bam")
    d = bytearray(open(aligned, "rb").read()); d[len(d) // 2] ^= 0x10
    open(bad, "wb").write(bytes(d))
    return [aligned, straddle, noidx, bad], tree, fasta


@pytest.mark.parametrize("tag,flags,envvar", [("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "ASAN_OPTIONS"),
                                              ("tsan", ["-fsanitize=thread"], "TSAN_OPTIONS")])
def test_host_code_under_sanitizers(tmp_path, tag, flags, envvar):
    exe = _build(tmp_path, flags, tag)
    bams, tree, fasta = _inputs(tmp_path)
    outs = {}
    for bam in bams:
        for threads in ("1", "5"):
            env = dict(os.environ, DUT_THREADS=threads)
            env[envvar] = "halt_on_error=1:detect_leaks=0" if tag == "asan" else "halt_on_error=1"
            r = subprocess.run([exe, bam, tree, fasta], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, (bam, threads, r.stderr[-3000:])
            assert "runtime error" not in r.stderr and "Sanitizer" not in r.stderr, r.stderr[-3000:]
            outs.setdefault(bam, set()).add(r.stdout)
    for bam, o in outs.items():
        assert len(o) == 1, bam                                         # the thread count never changes the result
    # the several-"device" call over the stand-in engine (one thread, reader pair and context per device): same BED text
    good = next(iter(outs[bams[0]]))
    assert "pass-bit selftest: 0 mismatch(es)" in good                 # the host half of the pass-bit form under the sanitizers
    assert good.count("bytes same") == 3 and "DIFFERENT" not in good and "a device that does not exist: rc -2" in good, good[-800:]
    # the same records whatever the block layout / index
    body = lambda s: [ln.split(" admit")[0] for ln in s.splitlines() if ln.startswith("tid")]
    assert body(next(iter(outs[bams[0]]))) == body(next(iter(outs[bams[2]]))) == body(next(iter(outs[bams[1]])))


def test_oracle_under_address_and_ub_sanitizers(tmp_path):
    """The C oracle built with -fsanitize=address,undefined runs the known-answer tests and randomized
    contigs (it is the checker of everything else: it must not rely on undefined behaviour)."""
    import sys
    lib = str(tmp_path / "liboracle_asan.so")
    r = subprocess.run(["gcc", "-O1", "-g", "-fPIC", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared",
                        "-o", lib, os.path.join(ROOT, "oracle", "callable_oracle.c")], capture_output=True, text=True)
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if r.returncode != 0 or not os.path.exists(asan_rt):
        pytest.skip("sanitizer build not available here")
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, oracle\n"
        "from helpers import load_kats, make_options, contig_inputs, oracle_run\n"
        "from decodingustools_amd import synth\n"
        "n = 0\n"
        "K = load_kats()\n"
        "for case in K['cases']:\n"
        "    opt = make_options({**K['default_options'], **case.get('options', {})})\n"
        "    cs = []\n"
        "    for i, c in enumerate(case['contigs']):\n"
        "        rec, ref = contig_inputs(c); cs.append((c['name'], c.get('tid', i), c['len'], ref, rec))\n"
        "    res, bed = oracle_run(cs, opt, %r)\n"
        "    assert bed == case['bed'], case['name']; n += 1\n"
        "assert n >= 5\n"
        "for seed in range(4):\n"
        "    L = 6000\n"
        "    rec = synth.adversarial_contig(L, 400, 50 + seed, deep=(seed == 3))\n"
        "    oracle_run([('c', 0, L, synth.make_reference(L, seed), rec)], make_options(dict(max_depth=[500, 0, 30, 100000][seed])), %r)\n"
        "print('ok', n)\n") % (ROOT, os.path.join(ROOT, "tests"), str(tmp_path / "k.bed"), str(tmp_path / "r.bed"))
    env = dict(os.environ, ORACLE_LIB=lib, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stderr[-3000:]
    assert "runtime error" not in r.stderr, r.stderr[-3000:]

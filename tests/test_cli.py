"""The command line tool's argument handling and its error paths that do not need a GPU
(flags and defaults: src/cli.rs:14-105; error texts: src/api/coverage.rs:56-75, api/mod.rs:24-47)."""
import os
import subprocess

import pytest

from bamio import write_bam, write_fasta
from decodingustools_amd import build as _b, synth


def run(*args, cwd=None):
    return subprocess.run([_b.CLI] + list(args), capture_output=True, text=True, cwd=cwd)


def test_usage_and_unknown_arguments():
    _b.build()
    r = run()
    assert r.returncode == 2 and "Usage: dut-coverage" in r.stderr
    r = run("coverage", "x.bam")                                   # -r is required
    assert r.returncode == 2
    r = run("coverage", "x.bam", "-r", "x.fa", "--no-such-flag")
    assert r.returncode == 2 and "unexpected argument '--no-such-flag'" in r.stderr
    r = run("--help")
    assert r.returncode == 0 and "--max-low-mapq-fraction 0.1" in r.stderr
    r = run("find-y-branch", "x.bam", "-r", "x.fa", "out.tsv")     # the tree file is required (no download here)
    assert r.returncode == 2 and "--tree" in r.stderr
    r = run("find-mt-branch", "x.bam", "-r", "x.fa", "out.tsv", "--tree", "t.json", "--provider", "nope")
    assert r.returncode == 2 and "invalid value 'nope'" in r.stderr


def test_errors_before_the_device_is_needed(tmp_path):
    _b.build()
    r = run("coverage", str(tmp_path / "missing.bam"), "-r", str(tmp_path / "missing.fa"), cwd=str(tmp_path))
    assert r.returncode == 1 and "Error: Analysis error: Failed to collect BAM stats" in r.stderr
    L = 3000
    bam = str(tmp_path / "t.bam")
    write_bam(bam, [("chr1", L)], {0: synth.short_read_contig(L, 5, 3)})
    r = run("coverage", bam, "-r", str(tmp_path / "missing.fa"), cwd=str(tmp_path))
    assert r.returncode == 1 and "Failed to open reference" in r.stderr
    fa = str(tmp_path / "t.fa")
    write_fasta(fa, [("chr1", synth.make_reference(L, 4))])
    r = run("coverage", bam, "-r", fa, "-L", "chrZ", cwd=str(tmp_path))
    assert r.returncode == 1 and "None of the specified contigs (chrZ) were found in the BAM file" in r.stderr
    # find-y-branch: the header does not say which genome this is (validation.rs:13-14)
    r = run("find-y-branch", bam, "-r", fa, str(tmp_path / "o.tsv"), "--tree", str(tmp_path / "t.json"))
    assert r.returncode == 1 and "Could not determine reference genome from BAM header" in r.stderr
    write_bam(bam, [("chr1", 248956422), ("chrX", L)], {1: synth.short_read_contig(L, 5, 3)})
    r = run("find-y-branch", bam, "-r", fa, str(tmp_path / "o.tsv"), "--tree", str(tmp_path / "t.json"))
    assert r.returncode == 1 and "No valid sequence found in BAM. Tried: chrY, Y, NC_000024.10, CM000686.2" in r.stderr

"""Shared helpers for the tests: fixture loading and the oracle driver."""
import json
import os
import types

import numpy as np

import oracle
from decodingustools_amd.records import ContigRecords

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_kats():
    with open(os.path.join(GOLDEN, "kats.json")) as f:
        return json.load(f)


def make_options(d):
    base = dict(min_depth=4, max_depth=500, min_mapping_quality=10, min_base_quality=20,
                min_depth_for_low_mapq=10, max_low_mapq=1, max_low_mapq_fraction=0.1)
    base.update(d or {})
    return types.SimpleNamespace(**base)


def contig_inputs(c):
    rec = ContigRecords.from_reads([tuple(r) for r in c["reads"]])
    ref = np.frombuffer(c["ref"].encode(), dtype=np.uint8).copy() if c.get("ref") is not None else None
    return rec, ref


def oracle_run(contigs, options, bed_path, dump=False):
    """contigs: list of (name, tid, length, ref ndarray|None, ContigRecords). Returns per-contig dict."""
    prof = oracle.Profiler(bed_path)
    out = {}
    try:
        for name, tid, length, ref, rec in contigs:
            stats, dumps = oracle.process_single_contig(prof, options, name, tid, length, ref, rec, dump=dump)
            out[name] = dict(stats=stats, dumps=dumps)
        for name in out:
            out[name]["state_counts"] = prof.contig_counts(name)
    finally:
        prof.close()
    with open(bed_path, "rb") as f:
        bed = f.read().decode()
    return out, bed

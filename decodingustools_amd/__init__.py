"""decodingustools_amd -- MI355X-native callable-loci (`coverage`) hot path of DecodingUsTools.

Only what the path needs: csrc/ (HIP kernels + C ABI + C++ host mirror), a ctypes binding and
the host-side mirror of the reference's callable_loci module API.
"""
from .records import ContigRecords  # noqa: F401
from .callable_loci import (CallableOptions, CalledState, CallableProfiler, ContigProfiler,  # noqa: F401
                            ContigResult, Engine, EngineError, admit_reads, compare_contig_names,
                            genome_summary, process_single_contig)

__all__ = ["ContigRecords", "CallableOptions", "CalledState", "CallableProfiler", "ContigProfiler",
           "ContigResult", "Engine", "EngineError", "admit_reads", "compare_contig_names",
           "genome_summary", "process_single_contig"]

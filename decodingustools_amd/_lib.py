"""ctypes binding of libcallable_hip.so (include/callable_loci.h, dut_coverage.h, dut_bam.h, dut_report.h,
dut_haplogroup.h).

There is no fallback: if the shared library is missing or does not load, importing the engine
raises.  Build it with `python -m decodingustools_amd.build` (or __graft_entry__.build()).
"""
import ctypes as C
import os

from . import build as _build

_LIB = None


class cl_options(C.Structure):
    _fields_ = [("min_depth", C.c_uint32), ("max_depth", C.c_uint32),
                ("min_mapping_quality", C.c_uint8), ("min_base_quality", C.c_uint8),
                ("min_depth_for_low_mapq", C.c_uint32), ("max_low_mapq", C.c_uint8),
                ("max_low_mapq_fraction", C.c_double)]


class cl_read_tile(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("pos", C.c_void_p), ("mapq", C.c_void_p),
                ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("qual_off", C.c_void_p), ("qual", C.c_void_p)]


class cl_read_tile_bits(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("pos", C.c_void_p), ("mapq", C.c_void_p),
                ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("qual_off", C.c_void_p), ("pass_bits", C.c_void_p), ("pass_sum", C.c_void_p)]


class cl_contig_summary(C.Structure):
    _fields_ = [("state_counts", C.c_uint64 * 6), ("n_covered_bases", C.c_uint64),
                ("summed_coverage", C.c_uint64), ("summed_baseq", C.c_uint64),
                ("summed_mapq", C.c_uint64), ("quality_bases", C.c_uint64),
                ("extent", C.c_uint64), ("max_raw_depth", C.c_uint64),
                ("n_intervals", C.c_uint64)]


class cl_interval(C.Structure):
    _fields_ = [("start", C.c_uint32), ("end", C.c_uint32), ("state", C.c_uint32)]


class cl_layout_info(C.Structure):
    _fields_ = [("form", C.c_int32), ("counter_planes", C.c_uint32), ("n_reads", C.c_uint64), ("n_records", C.c_uint64),
                ("n_windows", C.c_uint64), ("n_qual", C.c_uint64), ("n_cigar", C.c_uint64), ("row_groups", C.c_uint64),
                ("max_groups", C.c_uint64), ("run_table_entries", C.c_uint64), ("device_bytes", C.c_uint64),
                ("upload_h2d_bytes", C.c_uint64)]


class cl_site_tile(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("pos", C.c_void_p), ("mapq", C.c_void_p),
                ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("seq_off", C.c_void_p), ("seq4", C.c_void_p)]


class dut_records(C.Structure):
    _fields_ = [("n", C.c_uint64), ("pos", C.c_void_p), ("flag", C.c_void_p), ("mapq", C.c_void_p),
                ("cigar_off", C.c_void_p), ("cigar", C.c_void_p), ("qual_off", C.c_void_p),
                ("qual", C.c_void_p), ("qname_off", C.c_void_p), ("qname", C.c_void_p),
                ("pass_bits", C.c_void_p), ("pass_sum", C.c_void_p)]


class dut_contig_stats(C.Structure):
    _fields_ = [("length", C.c_uint64), ("n_covered_bases", C.c_uint64),
                ("summed_coverage", C.c_uint64), ("summed_baseq", C.c_uint64),
                ("summed_mapq", C.c_uint64), ("quality_bases", C.c_uint64),
                ("n_reads", C.c_uint32), ("reserved", C.c_uint32)]


class dut_contig_derived(C.Structure):
    _fields_ = [("coverage_percent", C.c_double), ("average_depth", C.c_double),
                ("average_mapq", C.c_double), ("average_baseq", C.c_double),
                ("q30_percentage", C.c_double)]


class dut_genome_summary(C.Structure):
    _fields_ = [("total_bases", C.c_uint64), ("callable_bases", C.c_uint64),
                ("callable_percentage", C.c_double), ("average_depth", C.c_double),
                ("average_mapq", C.c_double), ("average_baseq", C.c_double),
                ("q30_percentage", C.c_double), ("total_unique_reads", C.c_uint64),
                ("contigs_analyzed", C.c_uint64)]


class dut_export_meta(C.Structure):
    _fields_ = [("aligner", C.c_char_p), ("reference_build", C.c_char_p),
                ("sequencing_platform", C.c_char_p), ("read_length", C.c_uint64),
                ("bed_file", C.c_char_p), ("summary_html", C.c_char_p),
                ("coverage_plots", C.POINTER(C.c_char_p)), ("n_coverage_plots", C.c_size_t)]


class dut_snp_call(C.Structure):
    _fields_ = [("position", C.c_uint32), ("depth", C.c_uint32), ("freq", C.c_double), ("base", C.c_char)]


class dut_haplogroup_result(C.Structure):
    _fields_ = [("name", C.c_char_p), ("score", C.c_double), ("matching_snps", C.c_uint32),
                ("mismatching_snps", C.c_uint32), ("ancestral_matches", C.c_uint32), ("no_calls", C.c_uint32),
                ("total_snps", C.c_uint32), ("cumulative_snps", C.c_uint32), ("depth", C.c_uint32)]


CL_K_NAMES = ("prep", "bounds", "pileup", "rle")
CL_K_COUNT = 4

# every symbol the headers declare: (name, restype, argtypes)
SYMBOLS = [
    ("cl_abi_version", C.c_int, []),
    ("cl_device_count", C.c_int, []),
    ("cl_create", C.c_int, [C.POINTER(cl_options), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("cl_destroy", None, [C.c_void_p]),
    ("cl_last_error", C.c_char_p, [C.c_void_p]),
    ("cl_contig_begin", C.c_int, [C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_uint64]),
    ("cl_contig_reserve", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]),
    ("cl_contig_prefetch_qual", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    ("cl_push_reads", C.c_int, [C.c_void_p, C.POINTER(cl_read_tile)]),
    ("cl_push_reads_bits", C.c_int, [C.c_void_p, C.POINTER(cl_read_tile_bits)]),
    ("cl_contig_finish", C.c_int, [C.c_void_p, C.POINTER(cl_contig_summary),
                                   C.POINTER(C.POINTER(cl_interval)), C.POINTER(C.c_size_t)]),
    ("cl_contig_abort", C.c_int, [C.c_void_p]),
    ("cl_contig_upload", C.c_int, [C.c_void_p]),
    ("cl_contig_run", C.c_int, [C.c_void_p]),
    ("cl_contig_collect", C.c_int, [C.c_void_p, C.POINTER(cl_contig_summary),
                                    C.POINTER(C.POINTER(cl_interval)), C.POINTER(C.c_size_t)]),
    ("cl_sync", C.c_int, [C.c_void_p]),
    ("cl_device_summary", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("cl_set_profiling", C.c_int, [C.c_void_p, C.c_int]),
    ("cl_get_kernel_ms", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    ("cl_reset_kernel_ms", C.c_int, [C.c_void_p]),
    ("cl_contig_bytes", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("cl_contig_layout", C.c_int, [C.c_void_p, C.POINTER(cl_layout_info)]),
    ("cl_debug_ref_n_bits", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]),
    ("cl_debug_qual_pack", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint8, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]),
    ("cl_debug_host_create", C.c_int, [C.POINTER(cl_options), C.POINTER(C.c_void_p)]),
    ("cl_debug_pass_rows", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    ("cl_debug_depths", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    ("cl_debug_read_records", C.c_int, [C.c_int32, C.c_void_p, C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint64, C.c_uint64,
                                        C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("cl_site_pileup", C.c_int, [C.c_void_p, C.c_uint8, C.c_uint32, C.c_uint64, C.POINTER(cl_site_tile),
                                 C.c_void_p, C.c_size_t, C.c_void_p]),
    ("cl_site_upload", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(cl_site_tile)]),
    ("cl_site_run", C.c_int, [C.c_void_p, C.c_uint8, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("cl_site_pileup_stats", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    ("dut_profiler_new", C.c_void_p, [C.c_char_p]),
    ("dut_profiler_free", None, [C.c_void_p]),
    ("dut_profiler_enable_plots", None, [C.c_void_p, C.c_uint32]),
    ("dut_profiler_plot_bins", C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("dut_profiler_finish_plot", C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32]),
    ("dut_profiler_contig_counts", None, [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64)]),
    ("dut_profiler_feed_contig", C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t,
                                           C.POINTER(C.c_uint64)]),
    ("dut_admit_reads", C.c_int, [C.POINTER(cl_options), C.c_int32, C.c_uint32, C.POINTER(dut_records),
                                  C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    ("dut_process_single_contig", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(dut_contig_stats),
                                            C.POINTER(cl_options), C.c_char_p, C.c_int32, C.c_uint32,
                                            C.c_void_p, C.c_uint64, C.POINTER(dut_records)]),
    ("dut_process_single_contig_runs", C.c_int, [C.c_void_p, C.POINTER(dut_contig_stats), C.POINTER(cl_options), C.c_int32,
                                                 C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(dut_records),
                                                 C.POINTER(C.c_uint64), C.POINTER(C.POINTER(cl_interval)),
                                                 C.POINTER(C.c_size_t)]),
    ("dut_contig_derive", None, [C.POINTER(dut_contig_stats), C.POINTER(dut_contig_derived)]),
    ("dut_compare_contig_names", C.c_int, [C.c_char_p, C.c_char_p]),
    ("dut_genome_summary_build", None, [C.POINTER(dut_contig_stats), C.POINTER(C.c_uint64), C.c_size_t,
                                        C.POINTER(dut_genome_summary)]),
    ("dut_state_name", C.c_char_p, [C.c_uint32]),
    ("dut_bam_open", C.c_void_p, [C.c_char_p, C.c_char_p, C.c_size_t]),
    ("dut_bam_close", None, [C.c_void_p]),
    ("dut_bam_error", C.c_char_p, [C.c_void_p]),
    ("dut_bam_n_ref", C.c_int, [C.c_void_p]),
    ("dut_bam_ref_name", C.c_char_p, [C.c_void_p, C.c_int]),
    ("dut_bam_ref_len", C.c_uint32, [C.c_void_p, C.c_int]),
    ("dut_bam_header_text", C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    ("dut_bam_has_index", C.c_int, [C.c_void_p]),
    ("dut_bam_ref_mapped", C.c_int64, [C.c_void_p, C.c_int]),
    ("dut_bam_read_contig", C.c_int, [C.c_void_p, C.c_int, C.POINTER(dut_records), C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_void_p)]),
    ("dut_bam_read_contig_bits", C.c_int, [C.c_void_p, C.c_int, C.c_uint8, C.POINTER(dut_records)]),
    ("dut_fasta_open", C.c_void_p, [C.c_char_p, C.c_char_p, C.c_size_t]),
    ("dut_fasta_close", None, [C.c_void_p]),
    ("dut_fasta_fetch", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    ("dut_fasta_error", C.c_char_p, [C.c_void_p]),
    ("dut_coverage_files", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                     C.POINTER(cl_options), C.POINTER(C.c_char_p), C.c_size_t, C.c_int,
                                     C.c_char_p, C.c_size_t]),
    ("dut_coverage_files_multi", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                           C.POINTER(cl_options), C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(C.c_int), C.c_size_t,
                                           C.c_uint, C.c_char_p, C.c_size_t]),
    ("dut_bam_sample", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    # include/dut_report.h
    ("dut_detect_aligner", C.c_char_p, [C.c_char_p, C.c_size_t]),
    ("dut_reference_build", C.c_char_p, [C.c_char_p, C.c_size_t]),
    ("dut_detect_platform_from_qname", C.c_int, [C.c_char_p, C.c_size_t]),
    ("dut_parse_read_name", C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("dut_infer_specific_platform", C.c_char_p, [C.c_int, C.c_char_p]),
    ("dut_bam_stats_new", C.c_void_p, [C.c_size_t]),
    ("dut_bam_stats_free", None, [C.c_void_p]),
    ("dut_bam_stats_set_header", None, [C.c_void_p, C.c_char_p, C.c_size_t]),
    ("dut_bam_stats_add", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint16, C.c_uint32, C.c_char_p, C.c_size_t,
                                    C.c_int32]),
    ("dut_bam_stats_collect", C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    ("dut_bam_stats_aligner", C.c_char_p, [C.c_void_p]),
    ("dut_bam_stats_reference_build", C.c_char_p, [C.c_void_p]),
    ("dut_bam_stats_infer_platform", C.c_char_p, [C.c_void_p]),
    ("dut_bam_stats_primary_platform", C.c_int, [C.c_void_p]),
    ("dut_bam_stats_read_count", C.c_uint64, [C.c_void_p]),
    ("dut_bam_stats_average_read_length", C.c_uint64, [C.c_void_p]),
    ("dut_bam_stats_modal_read_length", C.c_uint64, [C.c_void_p]),
    ("dut_bam_stats_get", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]),
    ("dut_format_f64", C.c_size_t, [C.c_double, C.c_char_p]),
    ("dut_coverage_output_json", C.c_int, [C.POINTER(dut_contig_stats), C.POINTER(C.c_char_p),
                                           C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(dut_export_meta),
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("dut_free", None, [C.c_void_p]),
    ("dut_write_html_report", C.c_int, [C.POINTER(dut_contig_stats), C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_size_t,
                                        C.POINTER(dut_export_meta), C.c_uint64, C.c_char_p]),
    # include/dut_haplogroup.h
    ("dut_tree_parse", C.c_void_p, [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    ("dut_tree_load", C.c_void_p, [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    ("dut_tree_free", None, [C.c_void_p]),
    ("dut_tree_total_nodes", C.c_size_t, [C.c_void_p]),
    ("dut_tree_built_nodes", C.c_size_t, [C.c_void_p]),
    ("dut_tree_root_name", C.c_char_p, [C.c_void_p]),
    ("dut_tree_collect_sites", C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("dut_call_sites", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("dut_tree_score", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]),
    ("dut_write_haplogroup_report", C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                              C.c_size_t, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]),
    ("dut_validate_reference", C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_size_t, C.c_int,
                                         C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    ("dut_find_branch_files", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint8,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
]


def lib_path():
    # DUT_CALLABLE_LIB: tooling override to time a differently built variant of the same library
    return os.environ.get("DUT_CALLABLE_LIB") or _build.LIB


def load():
    """Load the shared library; raises if it is absent (no CPU fallback exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch ships its own HIP runtime; whichever of two runtimes initialises second in a process finds
    # no device.  With torch imported first this library resolves against torch's runtime and both work,
    # in either order of use -- so import it here when it is installed (DUT_NO_TORCH_PRELOAD=1 skips this).
    import sys
    if "torch" not in sys.modules and os.environ.get("DUT_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m decodingustools_amd.build` "
            "(needs hipcc). The engine has no CPU fallback.")
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.cl_abi_version() != 1:
        raise RuntimeError("libcallable_hip.so ABI version mismatch")
    _LIB = lib
    return lib

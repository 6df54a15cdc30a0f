/*
 * dut_coverage.h -- host-side mirror of the reference's callable_loci module API, in C.
 *
 * These are the pieces of the `coverage` path that stay on the host, above the device engine of
 * callable_loci.h, with the reference's names, argument meaning and error behaviour:
 *
 *   dut_profiler_*            CallableProfiler            profilers/callable_profiler.rs:11-160
 *   dut_contig_stats          ContigProfiler (data part)  profilers/contig_profiler.rs:7-20
 *   dut_admit_reads           what htslib's pileup keeps: BAM_FUNMAP drop + bam_plp_set_maxcnt
 *                             rule (mod.rs:55-60), region filter of
 *                             bam.fetch((tid,0,len)) (mod.rs:53), distinct read names
 *                             (contig_profiler.rs:59-62)
 *   dut_process_single_contig callable_loci::process_single_contig   mod.rs:44-147
 *   dut_contig_derive         get_coverage_stats / get_quality_stats  contig_profiler.rs:93-157
 *   dut_compare_contig_names  report.rs:339-393
 *   dut_genome_summary_build  report.rs:26-126
 *
 * Records are handed over decoded, structure-of-arrays, one contig at a time -- what a BAM
 * reader (rust-htslib in the reference, utils/bam_reader.rs:7-14) yields for the contig.
 */
#ifndef DUT_COVERAGE_H
#define DUT_COVERAGE_H

#include "callable_loci.h"

#ifdef __cplusplus
extern "C" {
#endif

/* All fetched records of one contig, file order.  Same fields as SURVEY.md Appendix B. */
typedef struct dut_records {
    uint64_t        n;
    const int32_t  *pos;
    const uint16_t *flag;
    const uint8_t  *mapq;
    const uint32_t *cigar_off;   /* n+1 */
    const uint32_t *cigar;
    const uint64_t *qual_off;    /* n+1 */
    const uint8_t  *qual;
    const uint32_t *qname_off;   /* n+1 */
    const uint8_t  *qname;
    /* the packed variant (dut_bam_read_contig_bits; both NULL otherwise): the base-quality test already taken -- bit
     * qual_off[i] + k of pass_bits <-> quality value k of read i --, and per read the sum of the passing values over
     * its matched bases (cl_read_tile_bits, callable_loci.h).  `qual` may then be NULL. */
    const uint64_t *pass_bits;
    const uint32_t *pass_sum;
} dut_records;

/* ContigProfiler fields the report reads (contig_profiler.rs:7-20, report.rs:40-86) */
typedef struct dut_contig_stats {
    uint64_t length;
    uint64_t n_covered_bases;
    uint64_t summed_coverage;
    uint64_t summed_baseq;
    uint64_t summed_mapq;
    uint64_t quality_bases;
    uint32_t n_reads;
    uint32_t reserved;
} dut_contig_stats;

typedef struct dut_contig_derived {
    double coverage_percent;
    double average_depth;
    double average_mapq;
    double average_baseq;
    double q30_percentage;
} dut_contig_derived;

typedef struct dut_genome_summary {
    uint64_t total_bases;
    uint64_t callable_bases;
    double   callable_percentage;
    double   average_depth;
    double   average_mapq;
    double   average_baseq;
    double   q30_percentage;
    uint64_t total_unique_reads;
    uint64_t contigs_analyzed;
} dut_genome_summary;

typedef struct dut_profiler dut_profiler;

/* CallableProfiler::new -- creates/truncates the BED file.  NULL on I/O error. */
dut_profiler *dut_profiler_new(const char *bed_path);
/* Drop: flushes; like the reference it does not write a pending state. */
void dut_profiler_free(dut_profiler *p);
/* The per-contig coverage figure (callable_profiler.rs:48-59,64-84; utils/histogram_plotter.rs:74-101,412-440):
 * after enable_plots every BED line of state CALLABLE / POOR_MAPPING_QUALITY / REF_N is also a range of the
 * current figure -- the duplicated last line of the previous contig included, as in the reference.
 * largest_contig_length: the longest selected contig other than "chrM" (api/coverage.rs:210-215).
 * plot_bins: positions of the three states per stride (ceil(largest / 2000); "chrM": ceil(16569 / 200)),
 * n = contig_length / stride + 1 entries each (arrays may be NULL; filled only if cap >= n).
 * finish_plot: what finish_contig does after the last line -- writes `<dir of the BED>/<contig>_coverage.svg`
 * (this project's own drawing of those arrays) and clears the ranges.  Returns 1 if a file was written,
 * 0 if there was nothing to draw, negative on error. */
void dut_profiler_enable_plots(dut_profiler *p, uint32_t largest_contig_length);
int dut_profiler_plot_bins(const dut_profiler *p, const char *contig, uint32_t contig_length, uint32_t *stride,
                           uint32_t *callable, uint32_t *low_qual, uint32_t *ref_n, size_t cap, size_t *n_bins);
int dut_profiler_finish_plot(dut_profiler *p, const char *contig, uint32_t contig_length);
/* get_contig_counts (callable_profiler.rs:158-160): zeros for an unknown contig */
void dut_profiler_contig_counts(const dut_profiler *p, const char *contig, uint64_t out[6]);
/* Feeds one contig's runs (what process_position would have produced position by position,
 * callable_profiler.rs:122-155) and then finish_contig's write_state (:64-66, state is NOT
 * cleared, so the next contig re-emits this contig's last line). */
int dut_profiler_feed_contig(dut_profiler *p, const char *contig, const cl_interval *iv, size_t n_iv,
                             const uint64_t state_counts[6]);

/* Read admission.  accepted[i] (n bytes) = 1 for the reads htslib's pileup would hold AND that
 * span at least one reference position; *n_unique_names = distinct read names among them.
 * maxcnt follows mod.rs:56-60: max_depth if > 0 else 500.
 * Returns CL_OK, CL_ERR_UNSORTED for out-of-order input (htslib aborts the pileup). */
int dut_admit_reads(const cl_options *opt, int32_t tid, uint32_t contig_len, const dut_records *rec,
                    uint8_t *accepted, uint32_t *n_unique_names, uint64_t *n_accepted);

/* process_single_contig (mod.rs:44-147) on the device engine: admission, SoA tiles to
 * cl_push_reads, cl_contig_finish, BED runs to the profiler, ContigProfiler numbers to *stats
 * (stats->length is set to contig_len).  ref/ref_len as cl_contig_begin.
 * Errors: negative cl_status; message via cl_last_error(ctx) ("Error processing contig: ..." is
 * prefixed by the caller as in api/coverage.rs:251). */
int dut_process_single_contig(cl_ctx *ctx, dut_profiler *prof, dut_contig_stats *stats,
                              const cl_options *opt, const char *contig_name, int32_t tid,
                              uint32_t contig_len, const uint8_t *ref, uint64_t ref_len,
                              const dut_records *rec);

/* The same without a BED writer: the contig's runs (context-owned, valid until the next contig of ctx)
 * and state counts are handed back -- what a rank of a multi-GPU run sends to the rank that writes
 * the BED (decodingustools_amd/coverage.py). */
int dut_process_single_contig_runs(cl_ctx *ctx, dut_contig_stats *stats, const cl_options *opt, int32_t tid,
                                   uint32_t contig_len, const uint8_t *ref, uint64_t ref_len,
                                   const dut_records *rec, uint64_t state_counts[6],
                                   const cl_interval **intervals, size_t *n_intervals);

void dut_contig_derive(const dut_contig_stats *s, dut_contig_derived *out);
int  dut_compare_contig_names(const char *a, const char *b);
/* stats/callable for contigs already in dut_compare_contig_names order (report.rs:37-38) */
void dut_genome_summary_build(const dut_contig_stats *stats, const uint64_t *callable, size_t n_contigs,
                              dut_genome_summary *out);

/* Debug names of CalledState (types.rs:36-43) */
const char *dut_state_name(uint32_t state);

#ifdef __cplusplus
}
#endif
#endif /* DUT_COVERAGE_H */

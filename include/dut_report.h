/*
 * dut_report.h -- the step after the hot path: what `coverage` writes to summary.json, in C.
 *
 * Mirrors (host-only; SURVEY.md 8f-2 and the part of 8f-4 it depends on):
 *   BamStats::{new,collect_stats,get_stats,...}   src/callable_loci/profilers/bam_stats.rs:9-280
 *   PlatformInference::*                          src/callable_loci/profilers/platform_inference.rs:13-293
 *   detect_aligner                                src/callable_loci/mod.rs:149-177
 *   ReferenceGenome::from_header / name           src/types.rs:105-156
 *   build_coverage_export                         src/callable_loci/report.rs:15-134
 *   CoverageOutput / CoverageExport as JSON       src/api/coverage.rs:134-145,
 *                                                 src/export/formats/coverage.rs:26-248,
 *                                                 written by serde_json::to_writer_pretty (src/main.rs:68-69)
 *
 * Where the reference picks "the most frequent" entry of a HashMap (modal read length, primary
 * platform, top instrument) its answer depends on hash iteration order when counts tie; here a
 * tie goes to the smallest key (platforms: declaration order), which is one of the answers the
 * reference can give.
 * Header text and read names are compared bytewise; `to_lowercase` / `to_uppercase` fold ASCII only.
 */
#ifndef DUT_REPORT_H
#define DUT_REPORT_H

#include "dut_coverage.h"

#ifdef __cplusplus
extern "C" {
#endif

/* SequencingPlatform, platform_inference.rs:3-10 (declaration order) */
enum dut_platform {
    DUT_PLATFORM_ILLUMINA = 0,
    DUT_PLATFORM_PACBIO   = 1,
    DUT_PLATFORM_NANOPORE = 2,
    DUT_PLATFORM_MGI      = 3,
    DUT_PLATFORM_UNKNOWN  = 4
};

/* detect_aligner (mod.rs:149-177): "BWA-MEM2", "BWA", "minimap2", "pbmm2", "Bowtie2", "STAR", "Unknown" */
const char *dut_detect_aligner(const char *header_text, size_t len);
/* ReferenceGenome::from_header(..).name() or "Unknown" (bam_stats.rs:54-56, types.rs:105-156) */
const char *dut_reference_build(const char *header_text, size_t len);
/* PlatformInference::detect_platform_from_qname (platform_inference.rs:17-90) */
int dut_detect_platform_from_qname(const uint8_t *qname, size_t len);
/* The parse_*_read_name functions (platform_inference.rs:95-200).  On success return 1 and give the
 * instrument (and flow cell, Illumina / MGI only: *fc_len = 0 otherwise) as sub-ranges of qname. */
int dut_parse_read_name(int platform, const uint8_t *qname, size_t len, const uint8_t **instrument,
                        size_t *instrument_len, const uint8_t **flow_cell, size_t *fc_len);
/* PlatformInference::infer_specific_platform (platform_inference.rs:213-293); top_instrument is the
 * most frequent instrument id or NULL when none was seen. */
const char *dut_infer_specific_platform(int platform, const char *top_instrument);

typedef struct dut_bam_stats dut_bam_stats;
dut_bam_stats *dut_bam_stats_new(size_t max_samples);                 /* BamStats::new; the caller uses 10000 (api/coverage.rs:56) */
void dut_bam_stats_free(dut_bam_stats *s);
void dut_bam_stats_set_header(dut_bam_stats *s, const char *header_text, size_t len);
/* One record of `bam.records()`, in file order; index = its 0-based ordinal (bam_stats.rs:60-139).
 * Returns 1 while more records are wanted, 0 once index >= max_samples (the record is ignored). */
int dut_bam_stats_add(dut_bam_stats *s, uint64_t index, uint16_t flag, uint32_t l_seq,
                      const uint8_t *qname, size_t qname_len, int32_t tlen);
/* collect_stats: header + the first max_samples records of the file (dut_bam.h reader). */
int dut_bam_stats_collect(dut_bam_stats *s, const char *bam_path, char *err, size_t err_len);
const char *dut_bam_stats_aligner(const dut_bam_stats *s);
const char *dut_bam_stats_reference_build(const dut_bam_stats *s);
const char *dut_bam_stats_infer_platform(const dut_bam_stats *s);     /* infer_platform, bam_stats.rs:223-226 */
int      dut_bam_stats_primary_platform(const dut_bam_stats *s);      /* get_primary_platform */
uint64_t dut_bam_stats_read_count(const dut_bam_stats *s);
uint64_t dut_bam_stats_average_read_length(const dut_bam_stats *s);   /* integer division, bam_stats.rs:185-191 */
uint64_t dut_bam_stats_modal_read_length(const dut_bam_stats *s);
/* get_stats (bam_stats.rs:145-175): keys "average_read_length", "paired_percentage",
 * "average_insert_size", "proper_pair_percentage".  Returns 1 and *out when the key is present. */
int dut_bam_stats_get(const dut_bam_stats *s, const char *key, double *out);

/* An f64 as serde_json prints it (ryu: shortest digits that round-trip; plain decimals for
 * 1e-5 <= |v| < 1e16 with at least one fractional digit, otherwise d.ddde[-]x).  buf >= 32 bytes.
 * Non-finite values print as null (serde_json). */
size_t dut_format_f64(double v, char *buf);

typedef struct dut_export_meta {
    const char *aligner;              /* CoverageSummary.aligner */
    const char *reference_build;
    const char *sequencing_platform;
    uint64_t    read_length;
    const char *bed_file;             /* OutputFiles, api/coverage.rs:140-145 */
    const char *summary_html;
    const char *const *coverage_plots;
    size_t      n_coverage_plots;
} dut_export_meta;

/* The CoverageOutput of one analysis as serde_json::to_writer_pretty prints it (no trailing
 * newline).  Contigs in any order: they are sorted with dut_compare_contig_names (report.rs:37-38).
 * *json is malloc'd; release with dut_free. */
int dut_coverage_output_json(const dut_contig_stats *stats, const char *const *names,
                             const uint64_t *state_counts /* n x 6 */, size_t n_contigs,
                             const dut_export_meta *meta, char **json, size_t *json_len);
void dut_free(void *p);
/* summary.html (report.rs:136-340): the sections, rows, labels and number formats of the reference's report from
 * the same export (contigs sorted the same way; a contig's figure is embedded when `<name>_coverage.svg` exists in
 * the working directory, as there) -- in this project's own markup; the reference's template files are
 * presentation and are not reproduced.  bam_stats_max_samples: BamStats.max_samples (10000 in the reference). */
int dut_write_html_report(const dut_contig_stats *stats, const char *const *names,
                          const uint64_t *state_counts /* n x 6 */, size_t n_contigs,
                          const dut_export_meta *meta, uint64_t bam_stats_max_samples, const char *html_path);

#ifdef __cplusplus
}
#endif
#endif /* DUT_REPORT_H */

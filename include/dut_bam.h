/*
 * dut_bam.h -- alignment / reference file input for the coverage path, in C.
 *
 * Replaces what the reference gets from rust-htslib before the hot path starts:
 *   BamReaderFactory::open_indexed            src/utils/bam_reader.rs:7-14
 *   bam.fetch((tid, 0, contig_len)) + records src/callable_loci/mod.rs:53-55
 *   faidx::Reader::from_path / fetch_seq      src/api/coverage.rs:73, mod.rs:79-80
 * and the file-level driver CoverageAnalyzer::run_analysis (src/api/coverage.rs:53-115), BamStats, the
 * coverage figures and the HTML report included (dut_report.h, dut_coverage.h).
 *
 * BAM only (BGZF + BAM records; a .bai or .csi beside the file is used to seek to a contig when present,
 * otherwise the file is read forward).  CRAM is not supported.
 */
#ifndef DUT_BAM_H
#define DUT_BAM_H

#include "dut_coverage.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dut_bam dut_bam;
typedef struct dut_fasta dut_fasta;

/* Opens a BAM file and parses its header.  NULL on failure (message in err). */
dut_bam *dut_bam_open(const char *path, char *err, size_t err_len);
void dut_bam_close(dut_bam *b);
const char *dut_bam_error(const dut_bam *b);
int dut_bam_n_ref(const dut_bam *b);
const char *dut_bam_ref_name(const dut_bam *b, int tid);      /* header.tid2name */
uint32_t dut_bam_ref_len(const dut_bam *b, int tid);          /* header.target_len */
const char *dut_bam_header_text(const dut_bam *b, size_t *len);
int dut_bam_has_index(const dut_bam *b);
/* Mapped reads of a reference as the .bai metadata pseudo-bin records them (samtools idxstats), -1 when
 * there is no index or it carries no metadata.  Used to balance contigs over GPUs before decoding. */
int64_t dut_bam_ref_mapped(const dut_bam *b, int tid);

/* All records with refID == tid, file order, decoded into reader-owned SoA buffers that stay
 * valid until the next call on this reader.  seq_off (in bases) / seq4 (4-bit codes, two per
 * byte, continuous) are filled when non-NULL.  Returns 0 or a negative cl_status. */
int dut_bam_read_contig(dut_bam *b, int tid, dut_records *out, const uint64_t **seq_off, const uint8_t **seq4);
/* The same with the base-quality test of mod.rs:33 taken while the records are parsed (their bytes are in the cache
 * then): out->qual is NULL, out->pass_bits / out->pass_sum are filled (dut_records, dut_coverage.h) -- one bit per
 * base instead of one byte leaves the reader: what the file-level coverage path uses in the default pass-bit form. */
int dut_bam_read_contig_bits(dut_bam *b, int tid, uint8_t min_base_quality, dut_records *out);

/* `bam.records()` on a plain reader (utils/bam_reader.rs:16-24): every record of the file, mapped or
 * not, from the first one, until fn returns 0 or the file ends.  index = 0-based ordinal; qname
 * without its NUL; tlen = the template length field.  The reader is left at the first record. */
typedef int (*dut_bam_sample_fn)(void *ud, uint64_t index, uint16_t flag, uint32_t l_seq,
                                 const uint8_t *qname, size_t qname_len, int32_t tlen);
int dut_bam_sample(dut_bam *b, dut_bam_sample_fn fn, void *ud);

/* FASTA with a faidx index (faidx::Reader::from_path, api/coverage.rs:73): the .fai beside the file is read and
 * validated, or built (and written there when possible) when it is missing.  NULL on failure. */
dut_fasta *dut_fasta_open(const char *path, char *err, size_t err_len);
void dut_fasta_close(dut_fasta *f);
/* The bases of one sequence, case preserved, reader-owned until the next call (fetch_seq, mod.rs:79).  A name the
 * index does not hold, a failed seek or read: negative cl_status, text in dut_fasta_error -- the reference fails
 * the contig there ("Error processing contig: ...").  A file shorter than its index says yields the bases that
 * are there (the rest reads as 'N', mod.rs:80). */
int dut_fasta_fetch(dut_fasta *f, const char *name, const uint8_t **bases, uint64_t *len);
const char *dut_fasta_error(const dut_fasta *f);

/* `coverage <bam> -r <fasta> -o <bed> -s <html> [-L contig]...` on one GPU: BamStats over the first
 * 10000 records, then per selected contig (ascending tid) read, admit, run the engine, append BED
 * lines; then write the CoverageOutput JSON (dut_report.h, what main.rs:68-69 puts in
 * ./summary.json) to summary_json (may be NULL).  The per-contig coverage figures `<contig>_coverage.svg` are
 * written beside the BED file (callable_profiler.rs:64-84) and, when summary_html is given, the HTML report
 * (api/coverage.rs:104; dut_report.h) to that path; NULL: no report, the JSON names "summary.html".
 * files.coverage_plots of the JSON lists the figures that exist relative to the working directory
 * (api/coverage.rs:263-274), in tid order.
 * contigs == NULL selects every header contig.  With a .bai and more than one selected contig the next
 * contig's records and bases are read ahead on a second reader while the current one is processed
 * (environment DUT_PIPELINE=0 turns that off; the output is the same either way).
 * Errors: negative cl_status, message in err ("None of the specified contigs (...) were found in
 * the BAM file" for an -L list that matches nothing, api/coverage.rs:187-204). */
int dut_coverage_files(const char *bam_path, const char *fasta_path, const char *bed_path,
                       const char *summary_json, const char *summary_html, const cl_options *opt,
                       const char *const *contigs, size_t n_contigs, int device_id, char *err, size_t err_len);


/* The same over SEVERAL devices of one node, below any Python or torch.distributed layer -- what a Rust caller of
 * process_contigs_api (api/coverage.rs:221-252) can bind: one host thread, one BAM / FASTA reader pair and one engine
 * context per entry of `devices` (HIP ordinals; the same ordinal may appear more than once: several contexts on one
 * device); the selected contigs are dealt to them by longest-processing-time-first on the index's mapped-read counts
 * (contig lengths when the index does not record them); every contig's runs, state counts and ContigProfiler numbers
 * come back through host memory and the BED is written by the calling thread in tid order with the duplicated last line
 * per contig (callable_profiler.rs:64-66), so the files are byte for byte those of the one-device call.  One process,
 * no collective.  The first error in tid order aborts the analysis, as the serial loop's `?` does.
 * flags: DUT_FILES_LEAVE_TO_EXIT -- the caller leaves the process right after the call (the command line tool): the
 * contexts, readers and decode buffers are not given back one by one.  n_devices == 1, flags == 0: dut_coverage_files. */
#define DUT_FILES_LEAVE_TO_EXIT 1u
int dut_coverage_files_multi(const char *bam_path, const char *fasta_path, const char *bed_path,
                             const char *summary_json, const char *summary_html, const cl_options *opt,
                             const char *const *contigs, size_t n_contigs, const int *devices, size_t n_devices,
                             unsigned flags, char *err, size_t err_len);

#ifdef __cplusplus
}
#endif
#endif /* DUT_BAM_H */

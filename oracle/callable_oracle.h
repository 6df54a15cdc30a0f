/*
 * callable_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the DecodingUsTools `coverage` / callable-loci hot path, used
 * as the parity oracle for the HIP implementation and as the `cpu_baseline` leg of bench.py.
 * Nothing in the product path (decodingustools_amd/, include/) may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY PINNING: the reference (Rust + rust-htslib 0.49 -> htslib C) cannot be built in this
 * image (no cargo/rustc, no htslib) and ships no tests, fixtures or golden vectors for this path
 * (SURVEY.md section 4 / 8c).  The oracle is therefore pinned by the hand-derived known-answer
 * tests KAT-1..KAT-6 of SURVEY.md 8(c) (tests/golden/) and by the reference source text itself.
 * The htslib pileup semantics restated here (bam_plp_push / bam_plp_next / bam_plp_auto /
 * resolve_cigar2 in htslib `sam.c`, reached via rust-htslib `IndexedReader::pileup`) are from the
 * published algorithm, not from a source file present in this container: "parity unpinned" at
 * that third-party boundary, as DESIGN.md states.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef CALLABLE_ORACLE_H
#define CALLABLE_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/callable_loci/options.rs:2-9 (selected_contigs is handled by the caller) */
typedef struct orc_options {
    uint32_t min_depth;
    uint32_t max_depth;
    uint8_t  min_mapping_quality;
    uint8_t  min_base_quality;
    uint32_t min_depth_for_low_mapq;
    uint8_t  max_low_mapq;
    double   max_low_mapq_fraction;
} orc_options;

/* src/callable_loci/types.rs:36-43 -- discriminants 0..5 and Debug names */
enum {
    ORC_REF_N = 0, ORC_CALLABLE = 1, ORC_NO_COVERAGE = 2, ORC_LOW_COVERAGE = 3,
    ORC_EXCESSIVE_COVERAGE = 4, ORC_POOR_MAPPING_QUALITY = 5
};

/* The records of ONE contig in file (coordinate) order, exactly the fields the path consumes
 * (SURVEY.md Appendix B).  This is what `bam.fetch((tid,0,len))` + `bam.pileup()` would read
 * (src/callable_loci/mod.rs:53-55).  cigar uses the BAM encoding len<<4|op, ops MIDNSHP=XB. */
typedef struct orc_reads {
    int64_t         n;
    const int32_t  *pos;        /* 0-based leftmost coordinate */
    const uint16_t *flag;       /* SAM flag; only 0x4 is consulted */
    const uint8_t  *mapq;
    const uint32_t *cigar_off;  /* n+1 offsets into cigar[] */
    const uint32_t *cigar;
    const uint64_t *qual_off;   /* n+1 offsets into qual[] (l_seq per read) */
    const uint8_t  *qual;       /* raw Phred, 0xFF when absent */
    const uint32_t *qname_off;  /* n+1 offsets into qname[] */
    const uint8_t  *qname;      /* read names, no NUL */
} orc_reads;

/* src/callable_loci/profilers/contig_profiler.rs:7-20 (fields the caller reads back) */
typedef struct orc_contig_stats {
    uint64_t length;
    uint64_t n_covered_bases;
    uint64_t summed_coverage;
    uint64_t summed_baseq;
    uint64_t summed_mapq;
    uint64_t quality_bases;
    uint32_t n_reads;
    uint32_t n_selected_reads;  /* wraps like the reference's u32 in a release build */
} orc_contig_stats;

/* derived f64 statistics: contig_profiler.rs:93-103,126-157 and report.rs:44-54 */
typedef struct orc_contig_derived {
    double coverage_percent;
    double average_depth;
    double average_mapq;
    double average_baseq;
    double q30_percentage;
} orc_contig_derived;

/* genome-wide numbers of report.rs:26-33,56-63,88-126 */
typedef struct orc_genome_summary {
    uint64_t total_bases;
    uint64_t callable_bases;
    double   callable_percentage;
    double   average_depth;
    double   average_mapq;
    double   average_baseq;
    double   q30_percentage;
    uint64_t total_unique_reads;
    uint64_t contigs_analyzed;
} orc_genome_summary;

typedef struct orc_profiler orc_profiler;   /* CallableProfiler, callable_profiler.rs:11-19 */

/* CallableProfiler::new (callable_profiler.rs:22-37): creates/truncates the BED file. */
orc_profiler *orc_profiler_new(const char *bed_path);
/* Drop: flushes the BufWriter.  Note: does NOT write the pending state (neither does the
 * reference; the last contig's last interval is written by finish_contig). */
void orc_profiler_free(orc_profiler *p);
/* get_contig_counts (callable_profiler.rs:158-160) */
void orc_profiler_contig_counts(const orc_profiler *p, const char *contig, uint64_t out[6]);

/* process_single_contig (src/callable_loci/mod.rs:44-147).
 *   tid         : the contig's index in the BAM header (htslib's iterator starts at tid 0, which
 *                 matters only for the very first push; kept for literalness)
 *   ref/ref_len : the FASTA bytes of this contig (case preserved); positions >= ref_len fetch
 *                 empty and become b'N' (mod.rs:79-80).
 *   dbg_*       : optional per-position dumps (size dbg_cap each) of (raw,qc,low,state) for the
 *                 positions visited; pass NULL to skip.  *dbg_extent receives 1 + last position
 *                 visited (== contig_len unless a read overhangs the contig end).
 * Returns 0, or a negative code with a message in errbuf (unsorted input, malformed CIGAR). */
int orc_process_single_contig(orc_profiler *prof, orc_contig_stats *stats,
                              const orc_options *opt, const char *contig_name, int32_t tid,
                              uint32_t contig_len, const uint8_t *ref, uint64_t ref_len,
                              const orc_reads *reads,
                              uint32_t *dbg_raw, uint32_t *dbg_qc, uint32_t *dbg_low,
                              uint8_t *dbg_state, uint64_t dbg_cap, uint64_t *dbg_extent,
                              char *errbuf, size_t errbuf_len);

/* contig_profiler.rs:93-103,126-157 + report.rs:44-54 */
void orc_contig_derive(const orc_contig_stats *s, orc_contig_derived *out);

/* report.rs:339-393 compare_contig_names: returns <0, 0, >0 */
int orc_compare_contig_names(const char *a, const char *b);

/* report.rs:26-126 over contigs ALREADY sorted by the caller with orc_compare_contig_names.
 * callable[i] = counts[CALLABLE] of contig i. */
void orc_genome_summary_build(const orc_contig_stats *stats, const uint64_t *callable,
                              size_t n_contigs, orc_genome_summary *out);

/* Which reads the pileup accepts (FUNMAP drop + maxcnt rule):
 * accepted[i] = 1 if read i entered the pileup list (and so counts in columns).  Uses the same
 * engine as orc_process_single_contig.  Returns 0 or negative error. */
int orc_accepted_reads(const orc_options *opt, int32_t tid, uint32_t contig_len,
                       const orc_reads *reads,
                       uint8_t *accepted, char *errbuf, size_t errbuf_len);

/* Config 5: haplogroup::caller::process_region (src/haplogroup/caller.rs:62-152) on one contig.
 *   seq4/seq_off : BAM 4-bit packed sequence per read (two bases per byte, high nibble first),
 *                  seq_off in BASES (n+1 entries; l_seq per read)
 *   sites        : 1-based positions (vcf_pos) that have a locus on this contig, any order,
 *                  distinct
 * Output per site i: out_total[i] = bases.len(), out_base[i]/out_count[i] = majority base and
 * its count (0/0 when total==0), out_called[i] = 1 iff total>=min_depth && freq>=0.7,
 * out_freq[i] = count/total (0 if total==0).  base_hist (n_sites*16, optional) = count per
 * 4-bit code.  Ties in the majority are resolved toward the smallest 4-bit code (the reference's
 * HashMap order is unspecified on ties, but a tie can never reach 0.7). */
int orc_site_pileup(uint32_t min_depth, uint8_t min_quality, uint32_t contig_len,
                    const uint8_t *ref, uint64_t ref_len,
                    const orc_reads *reads, const uint64_t *seq_off, const uint8_t *seq4,
                    const uint32_t *sites, size_t n_sites,
                    uint32_t *out_total, uint8_t *out_base, uint32_t *out_count,
                    uint8_t *out_called, double *out_freq, uint32_t *base_hist);

#ifdef __cplusplus
}
#endif
#endif

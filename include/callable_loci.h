/*
 * callable_loci.h -- C ABI of the MI355X (gfx950) callable-loci engine.
 *
 * This is the drop-in boundary for the per-contig hot path of DecodingUsTools' `coverage`
 * subcommand.  The reference has no FFI here: the seam is the Rust function
 *     callable_loci::process_single_contig            (src/callable_loci/mod.rs:44-52)
 * called by process_single_contig_api                 (src/api/coverage.rs:238-252).
 * Everything that function computes per reference position -- the htslib pileup columns
 * (mod.rs:65-71), process_position (mod.rs:17-42), CallableProfiler::process_position /
 * process_state (profilers/callable_profiler.rs:89-155) and ContigProfiler::process_position
 * (profilers/contig_profiler.rs:47-83) -- is replaced by the calls below.  What stays on the
 * caller's side (host code, above this ABI): BAM/FASTA decoding, the FUNMAP / maxcnt read
 * admission rule, the unique-read-name count, the BED text writer with its duplicate-line
 * behaviour (callable_profiler.rs:39-66) and the f64 summary derivation (report.rs:15-134).
 *
 * Plain C types only; no exceptions cross the boundary.  Every function returns CL_OK (0) or a
 * negative cl_status; cl_last_error() returns the message of the last failure on that context.
 * A context is bound to one HIP device and is NOT thread-safe: one host thread drives one
 * context (the reference is single-threaded, src/main.rs:63-67).  There is no CPU fallback: if
 * no HIP device is usable cl_create() fails.
 */
#ifndef CALLABLE_LOCI_H
#define CALLABLE_LOCI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CL_ABI_VERSION 1

typedef enum cl_status {
    CL_OK = 0,
    CL_ERR_INVALID = -1,      /* bad argument / call out of sequence                        */
    CL_ERR_DEVICE = -2,       /* HIP runtime error (message has the hipError string)        */
    CL_ERR_UNSORTED = -3,     /* reads not coordinate sorted (htslib: "unsorted input")     */
    CL_ERR_CIGAR = -4,        /* malformed CIGAR (the cases htslib asserts on)              */
    CL_ERR_NOMEM = -5,
    CL_ERR_RANGE = -6         /* coordinate beyond what the engine addresses (2^32-1)       */
} cl_status;

/* CallableOptions, src/callable_loci/options.rs:2-9 (CLI defaults src/cli.rs:34-60:
 * 4, 500, 10, 20, 10, 1, 0.1).  selected_contigs stays with the caller. */
typedef struct cl_options {
    uint32_t min_depth;
    uint32_t max_depth;
    uint8_t  min_mapping_quality;
    uint8_t  min_base_quality;
    uint32_t min_depth_for_low_mapq;
    uint8_t  max_low_mapq;
    double   max_low_mapq_fraction;
} cl_options;

/* CalledState discriminants, src/callable_loci/types.rs:36-43 */
enum {
    CL_REF_N = 0, CL_CALLABLE = 1, CL_NO_COVERAGE = 2, CL_LOW_COVERAGE = 3,
    CL_EXCESSIVE_COVERAGE = 4, CL_POOR_MAPPING_QUALITY = 5
};

/* One tile of ACCEPTED reads of the current contig, structure-of-arrays, coordinate sorted
 * (tiles are pushed in order; the first read of a tile must not start before the last read of
 * the previous one).  "Accepted" = what htslib's bam_plp_push keeps: the caller has dropped
 * BAM_FUNMAP reads and applied the maxcnt rule (the host library's dut_admit_reads does both).
 * Fields are the ones the path consumes (mod.rs:22-37, contig_profiler.rs:54-76):
 *   pos        0-based leftmost reference coordinate (record.pos())
 *   mapq       record.mapq()
 *   cigar      BAM encoding len<<4|op, op codes M0 I1 D2 N3 S4 H5 P6 =7 X8
 *   qual       record.qual(): raw Phred bytes, l_seq per read, 0xFF when absent
 * The caller owns the buffers; they are copied before the call returns. */
typedef struct cl_read_tile {
    uint64_t        n_reads;
    const int32_t  *pos;        /* n_reads                       */
    const uint8_t  *mapq;       /* n_reads                       */
    const uint32_t *cigar_off;  /* n_reads + 1, cigar_off[0] may be non-zero (tile-relative base) */
    const uint32_t *cigar;      /* indexed by cigar_off          */
    const uint64_t *qual_off;   /* n_reads + 1                   */
    const uint8_t  *qual;       /* indexed by qual_off           */
} cl_read_tile;

/* What the caller reads back per contig (report.rs:40-86):
 *   state_counts      CallableProfiler::get_contig_counts, callable_profiler.rs:158-160
 *   the next five     ContigProfiler fields, contig_profiler.rs:11-15
 *   extent            number of positions classified: max(contig_len, largest read end)
 *                     (a read overhanging the contig end makes the reference walk past it)
 *   max_raw_depth     largest pileup column
 * n_reads (distinct read names) is a host-side count and not part of this struct. */
typedef struct cl_contig_summary {
    uint64_t state_counts[6];
    uint64_t n_covered_bases;
    uint64_t summed_coverage;
    uint64_t summed_baseq;
    uint64_t summed_mapq;
    uint64_t quality_bases;
    uint64_t extent;
    uint64_t max_raw_depth;
    uint64_t n_intervals;
} cl_contig_summary;

/* One run of equal state: [start, end) 0-based half open, exactly one BED line of
 * callable_profiler.rs:42-46 */
typedef struct cl_interval {
    uint32_t start;
    uint32_t end;
    uint32_t state;
} cl_interval;

typedef struct cl_ctx cl_ctx;

/* ---- lifecycle ------------------------------------------------------------------------- */
int  cl_abi_version(void);
int  cl_device_count(void);
/* device_id: HIP ordinal.  stream: an existing hipStream_t to enqueue on (e.g. the caller's
 * current stream), or NULL to let the context create its own. */
cl_status cl_create(const cl_options *opt, int device_id, void *stream, cl_ctx **out);
void cl_destroy(cl_ctx *ctx);
const char *cl_last_error(const cl_ctx *ctx);

/* ---- per contig: the replacement of process_single_contig ------------------------------- */
/* Starts a contig.  ref_bases: the contig_len FASTA bytes, case preserved (mod.rs:79-80: a
 * missing base reads as 'N'); ref_len < contig_len is allowed, the rest is 'N'. */
cl_status cl_contig_begin(cl_ctx *ctx, int32_t tid, uint32_t contig_len,
                          const uint8_t *ref_bases, uint64_t ref_len);
/* Optional size hint for the contig that was just begun: totals over all tiles that will be pushed.
 * Saves regrowing the staging / device buffers; never required.  Given before the first tile, it also lets the engine
 * allocate the contig's device buffers on a thread of its own while the caller admits and pushes the reads
 * (cl_contig_upload, cl_contig_abort, cl_contig_begin and cl_destroy wait for that thread). */
cl_status cl_contig_reserve(cl_ctx *ctx, uint64_t n_reads, uint64_t n_cigar_ops, uint64_t n_qual_bytes);
/* Optional, byte forms only (DUT_QUAL_FORM=bytes; a no-op in the default pass-bit form, where no quality byte goes to
 * the device): starts sending quality bytes to the device before their tile is pushed, so that the transfer runs
 * beside the caller's own work on the records (the host driver's read admission, say).  `qual` must be exactly
 * the bytes the NEXT cl_push_reads will present (tile.qual + tile.qual_off[0], tile.qual_off[n] - tile.qual_off[0]
 * of them) and must stay valid until that call returns; a push that presents anything else simply sends its own
 * bytes (the prefetch is dropped).  Blocks of less than 4 MiB are ignored.  Never required. */
cl_status cl_contig_prefetch_qual(cl_ctx *ctx, const uint8_t *qual, uint64_t n_bytes);
/* Appends a tile (coordinate order across tiles).  The caller's buffers are free again on return.  The base-quality
 * test of mod.rs:33 is taken here, where the quality bytes are read once ("or a packed pass-bitmask
 * variant"; 8d: q = 1/8): one bit per base stays in host staging, together with the tile's share of summed_baseq
 * (contig_profiler.rs:65-70, a per-read separable sum); cl_contig_upload lays the bits out as the rows the pileup
 * kernel counts.  With DUT_QUAL_FORM=bytes in the environment of cl_create the quality bytes themselves go to HBM
 * (small tiles via host staging, tiles of >= 4 MiB through a pinned staging ring) and are tested on the device. */
cl_status cl_push_reads(cl_ctx *ctx, const cl_read_tile *tile);
/* The packed pass-bitmask variant of cl_push_reads: for a caller that has taken the base-quality test of
 * mod.rs:33 itself, where it decodes the records (the library's own BAM reader does: dut_bam_read_contig_bits).  Same
 * tile, with instead of the quality bytes
 *   qual_off   n_reads + 1 offsets of the reads' quality VALUES, as in cl_read_tile -- now bit offsets into pass_bits
 *   pass_bits  bit g (word g / 64, bit g % 64) = 1 iff quality value g >= min_base_quality of the context's options
 *              (an absent quality string, 0xFF bytes, passes); bits beyond qual_off[n] are ignored
 *   pass_sum   per read: the sum of the quality values that pass, over the bases of its M/=/X operations that have a
 *              quality value (its share of summed_baseq, contig_profiler.rs:65-70; a read's sum is below 2^32)
 * Pass-bit form only (the default): with DUT_QUAL_FORM=bytes the engine needs the bytes and refuses with
 * CL_ERR_INVALID.  Tiles of both kinds may be mixed within a contig. */
typedef struct cl_read_tile_bits {
    uint64_t        n_reads;
    const int32_t  *pos;
    const uint8_t  *mapq;
    const uint32_t *cigar_off;
    const uint32_t *cigar;
    const uint64_t *qual_off;
    const uint64_t *pass_bits;
    const uint32_t *pass_sum;
} cl_read_tile_bits;
cl_status cl_push_reads_bits(cl_ctx *ctx, const cl_read_tile_bits *tile);
/* upload + run + collect in one call.  *intervals points at context-owned memory, valid until
 * the next cl_contig_begin / cl_destroy. */
cl_status cl_contig_finish(cl_ctx *ctx, cl_contig_summary *out,
                           const cl_interval **intervals, size_t *n_intervals);
/* Abandons the contig that was begun (the error path of a caller: mod.rs:79 `?` leaves process_single_contig the same
 * way).  A quality prefetch that no tile has claimed is waited for and discarded -- no copy reads the caller's buffer
 * once this returns, and the device's staging ring is free for other contexts --, staged reads are dropped.  The
 * message of cl_last_error() survives.  cl_contig_begin and cl_destroy imply it. */
cl_status cl_contig_abort(cl_ctx *ctx);

/* ---- the same, split so that a caller can keep a contig resident in HBM and re-run it ---- */
cl_status cl_contig_upload(cl_ctx *ctx);      /* H2D of everything pushed; synchronous     */
cl_status cl_contig_run(cl_ctx *ctx);         /* enqueue all kernels on the stream; async  */
cl_status cl_contig_collect(cl_ctx *ctx, cl_contig_summary *out,
                            const cl_interval **intervals, size_t *n_intervals);
cl_status cl_sync(cl_ctx *ctx);
/* device pointer + byte size of the resident per-contig summary record (cl_contig_summary
 * layout, valid after cl_contig_run completes) -- for gathering summaries across GPUs with a
 * collective without a host round trip */
cl_status cl_device_summary(cl_ctx *ctx, void **dev_ptr, size_t *bytes);

/* ---- measurement ------------------------------------------------------------------------ */
enum { CL_K_PREP = 0, CL_K_BOUNDS = 1, CL_K_PILEUP = 2, CL_K_RLE = 3, CL_K_COUNT = 4 };
/* When on, every cl_contig_run brackets each kernel group with hipEvents on the stream. */
cl_status cl_set_profiling(cl_ctx *ctx, int on);
/* Accumulated milliseconds per kernel group and number of runs since the last reset.  The per-read index (read
 * ends, CIGAR checkpoints) and the window bounds are built on the host at cl_contig_upload, not in a run: CL_K_PREP
 * and CL_K_BOUNDS read 0 (the slots are kept so that the table's layout does not change). */
cl_status cl_get_kernel_ms(cl_ctx *ctx, double ms[CL_K_COUNT], uint64_t *n_runs);
cl_status cl_reset_kernel_ms(cl_ctx *ctx);
/* Bytes of the resident inputs the pileup kernel of the contig's form must read at least once, counted strictly --
 * the array elements that kernel addresses: pass-bit rows or quality bytes, read records or run-table entries and
 * per-read fields, reference bases, window records; no CIGAR word (none is resident) -- and of the intervals it leaves
 * behind (12 bytes each; the per-position counters and states never reach HBM): the traffic one cl_contig_run cannot
 * do without. */
cl_status cl_contig_bytes(cl_ctx *ctx, uint64_t *input_bytes, uint64_t *output_bytes);
/* What is resident for the uploaded contig: the form of the pileup kernel (3 pass-bit rows; 0 records + quality bytes,
 * 2 run table + quality bytes: DUT_QUAL_FORM=bytes), element counts, and the HBM the context holds. */
typedef struct cl_layout_info {
    int32_t  form;
    uint32_t counter_planes;      /* pass-bit form: 8, 16 or 32 bit-sliced counter planes                        */
    uint64_t n_reads, n_records, n_windows;
    uint64_t n_qual;              /* quality bytes of the contig (on the device only in the byte forms)          */
    uint64_t n_cigar;             /* CIGAR operations of the contig (never on the device)                        */
    uint64_t row_groups;          /* pass-bit form: 1 KB groups of 4 rows                                        */
    uint64_t max_groups;          /* ... most groups of any window                                               */
    uint64_t run_table_entries;   /* byte form 2                                                                 */
    uint64_t device_bytes;        /* capacity of every device buffer of the context                              */
    uint64_t upload_h2d_bytes;    /* what cl_contig_upload (byte forms: and cl_push_reads) sent over the link    */
} cl_layout_info;
cl_status cl_contig_layout(cl_ctx *ctx, cl_layout_info *out);

/* ---- test hooks ------------------------------------------------------------------------- */
/* Re-runs the resident contig with per-position dumps: raw_depth, qc_depth, low_mapq_count
 * (mod.rs:17-42) and state, each `cap` entries (cap >= extent), host buffers, any may be NULL. */
cl_status cl_debug_depths(cl_ctx *ctx, uint32_t *raw, uint32_t *qc, uint32_t *low,
                          uint8_t *state, uint64_t cap);

/* The records the short-read form of the pileup kernel reads for ONE read (host code only, no device needed): what the
 * upload walk makes of a read at `pos` with the given CIGAR, mapping quality and quality-string length, the quality
 * bytes starting at offset `qual_off` -- a head {pos, span, qual offset of its run, mapq | 0x100 | run length << 16}
 * and a piece {pos of the run, 0, qual offset, mapq | run length << 16} per further M/=/X run (mod.rs:22-37: the read is
 * in every column of its span, the bases of its match operations that have a quality byte are tested).  Writes up to
 * `cap` records of four 32-bit words each to `out`, the count to *n_records (also when it exceeds cap) and the
 * phase (reference position - query offset of the first run, mod 16) to *phase.  Reads below min_mapping_quality get
 * the head alone; a read without a reference span gets no record. */
cl_status cl_debug_read_records(int32_t pos, const uint32_t *cigar, uint32_t n_ops, uint8_t mapq, uint8_t min_mapping_quality,
                                uint64_t qual_off, uint64_t qual_len, uint32_t *out, uint32_t cap,
                                uint32_t *n_records, uint32_t *phase);

/* Host only: the pass bits of n quality bytes (words_out: ceil(n / 64) 64-bit words, bit i of word i / 64 <-> byte i,
 * zeros above byte n) and the sum of the passing bytes, as cl_push_reads takes them -- at `level` 0 scalar, 1 SSE2,
 * 2 AVX2 where the CPU has it (else SSE2); 10, 11, 12: the same levels through the one-pass form (bits and sum together). */
cl_status cl_debug_qual_pack(const uint8_t *qual, uint64_t n, uint8_t min_base_quality, int level, uint64_t *words_out,
                             uint64_t *sum_out);
/* Host only: the reference's "is N" bits as cl_contig_upload sends them in the pass-bit form (mod.rs:100-101: a base
 * that is 'N' or 'n'; mod.rs:79-80: positions beyond the reference read as 'N'): bit i of word w <-> position 64 w + i,
 * n_words words for n_bases bases (levels as above). */
cl_status cl_debug_ref_n_bits(const uint8_t *ref, uint64_t n_bases, uint64_t n_words, int level, uint64_t *words_out);
/* A context WITHOUT a device, for the CPU test suite only: cl_contig_begin / cl_push_reads (pass-bit form) stage a
 * contig on the host exactly as a device context does, and cl_debug_pass_rows runs the upload's row builder over it.
 * Every call that needs a device fails with CL_ERR_DEVICE: there is no CPU pileup. */
cl_status cl_debug_host_create(const cl_options *opt, cl_ctx **out);
/* The pass-bit rows of the staged contig as cl_contig_upload would build them (pass_rows.h): n_groups[w] groups of 4 rows
 * per window of 2048 positions (n_win_cap entries at most; *n_windows = how many there are), the groups themselves window
 * after window in rows[0, cap_words) (256 words each: word (block << 2) | (row & 3) of group row >> 2; *n_words = how
 * many words there are, also when that exceeds cap_words), and the contig's share of summed_baseq from the push walk. */
cl_status cl_debug_pass_rows(cl_ctx *ctx, uint32_t *n_groups, uint32_t n_win_cap, uint32_t *rows, uint64_t cap_words,
                             uint64_t *n_words, uint32_t *n_windows, uint64_t *summed_baseq);

/* ---- config 5: site-list pileup (haplogroup::caller::process_region,
 *      src/haplogroup/caller.rs:62-152) ---------------------------------------------------- */
typedef struct cl_site_tile {
    uint64_t        n_reads;    /* ALL fetched records of the contig, no flag filter (:75-80) */
    const int32_t  *pos;
    const uint8_t  *mapq;
    const uint32_t *cigar_off;
    const uint32_t *cigar;
    const uint64_t *seq_off;    /* n_reads + 1, in BASES                                    */
    const uint8_t  *seq4;       /* BAM 4-bit packed bases, high nibble first                */
} cl_site_tile;
/* hist[n_sites*16]: per site (1-based vcf_pos, caller.rs:94) the number of reads with
 * mapq >= min_quality showing each 4-bit base code at an M/=/X position.
 * Only the reads that can add to the histogram are sent to the device (position inside the contig, mapq >= min_quality,
 * a site inside the reference span: two short reads in five at one site per ~300 bases); tiles with a read of 255
 * CIGAR operations or 65 535 bases and more travel whole.  The tile this call leaves on the device therefore serves this
 * call only: cl_site_run after it is refused (cl_site_upload gives a tile that serves any list). */
cl_status cl_site_pileup(cl_ctx *ctx, uint8_t min_quality, uint32_t contig_len,
                         uint64_t ref_len, const cl_site_tile *tile,
                         const uint32_t *sites, size_t n_sites, uint32_t *hist);

/* The same in two steps: the tile goes to HBM once (through the pinned staging ring) and stays resident -- until the
 * next cl_site_upload or cl_destroy -- and any number of site lists are run over it (find-y-branch --show-snps asks the
 * same reads about several lists; caller.rs:8-59 fetches the region again each time).  cl_site_pileup = both, for one
 * list (and sends less: above). */
cl_status cl_site_upload(cl_ctx *ctx, uint32_t contig_len, uint64_t ref_len, const cl_site_tile *tile);
cl_status cl_site_run(cl_ctx *ctx, uint8_t min_quality, const uint32_t *sites, size_t n_sites, uint32_t *hist);

/* Measurement: duration of the last cl_site_pileup's kernel (HIP events on the context's stream, milliseconds)
 * and its algorithmic bytes (4-bit bases, per-read fields and CIGAR words read, the sites'
 * counters written). */
cl_status cl_site_pileup_stats(cl_ctx *ctx, double *kernel_ms, uint64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif /* CALLABLE_LOCI_H */

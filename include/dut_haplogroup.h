/*
 * dut_haplogroup.h -- the plumbing around the site-list pileup (cl_site_pileup) of
 * `find-y-branch` / `find-mt-branch` (BASELINE config 5), in C.
 *
 * Mirrors (host side; the per-read work is the device engine's cl_site_pileup):
 *   FtdnaTreeProvider::{parse_tree,build_tree}        src/vendor/ftdna.rs:10-168
 *   DecodingUsTreeProvider::{parse_tree,build_tree}   src/vendor/decoding_us.rs:10-220
 *   tree::load_tree (root selection), collect_snps, find_path_to_root   src/haplogroup/tree.rs:7-97
 *   validation::validate_reference                    src/haplogroup/validation.rs:5-35
 *   caller::process_region, the per-site call         src/haplogroup/caller.rs:132-149
 *   scoring::calculate_haplogroup_score               src/haplogroup/scoring.rs:8-148
 *   collect_scored_paths, get_snp_details, the TSV    src/haplogroup/mod.rs:17-258
 *
 * The reference downloads the tree (utils/cache.rs:75-133); here it is read from a local JSON file
 * of the same shape.  Where the reference's result depends on HashMap iteration order the choice
 * made here is stated at the function.
 */
#ifndef DUT_HAPLOGROUP_H
#define DUT_HAPLOGROUP_H

#include "dut_bam.h"

#ifdef __cplusplus
extern "C" {
#endif

enum dut_tree_provider { DUT_PROVIDER_FTDNA = 0, DUT_PROVIDER_DECODINGUS = 1 };   /* cli::TreeProvider */
enum dut_tree_type { DUT_TREE_YDNA = 0, DUT_TREE_MTDNA = 1 };                      /* utils::cache::TreeType */

typedef struct dut_tree dut_tree;

/* provider.parse_tree + load_tree's root selection + provider.build_tree.
 * Errors (NULL, message in err): JSON that does not deserialize into the provider's structs;
 * "No root node found in FTDNA tree" / "Multiple root nodes found in FTDNA tree" (tree.rs:36-43),
 * "Multiple root nodes found in tree" / "No root node found" (decoding_us.rs:92-99),
 * "Failed to build tree" (tree.rs:53).
 * DecodingUs children are attached in ascending node index (the reference: hash order). */
dut_tree *dut_tree_parse(const char *json, size_t len, int provider, int tree_type, char *err, size_t err_len);
dut_tree *dut_tree_load(const char *json_path, int provider, int tree_type, char *err, size_t err_len);
void dut_tree_free(dut_tree *t);
size_t dut_tree_total_nodes(const dut_tree *t);      /* tree.all_nodes.len(), printed by load_tree */
size_t dut_tree_built_nodes(const dut_tree *t);      /* nodes reachable from the root */
const char *dut_tree_root_name(const dut_tree *t);

/* collect_snps (tree.rs:58-79) reduced to what process_region tests per base (caller.rs:96-103):
 * the distinct 1-based positions that carry a SNP locus with coordinates for build_id, ascending;
 * relevant[i] = 1 iff one of the loci there has coordinates[build_id].chromosome == ref_name.
 * Arrays are malloc'd (dut_free). */
int dut_tree_collect_sites(const dut_tree *t, const char *build_id, const char *ref_name,
                           uint32_t **sites, uint8_t **relevant, size_t *n_sites);

/* snp_calls entry: HashMap<u32, (char, u32, f64)> (caller.rs:143-147) */
typedef struct dut_snp_call {
    uint32_t position;      /* 1-based */
    uint32_t depth;         /* bases.len() */
    double   freq;          /* count of the majority base / depth */
    char     base;          /* upper-case decoded base ("=ACMGRSVTWYHKDBN") */
} dut_snp_call;

/* The per-site call of process_region (caller.rs:132-149) from cl_site_pileup's histograms
 * (hist[n*16], 4-bit base codes): depth >= min_depth and majority frequency >= 0.7.  Sites with
 * relevant[i] == 0 are skipped (relevant may be NULL = all).  calls: malloc'd, ascending position. */
int dut_call_sites(const uint32_t *sites, const uint8_t *relevant, const uint32_t *hist, size_t n_sites,
                   uint32_t min_depth, dut_snp_call **calls, size_t *n_calls);

/* HaplogroupResult (haplogroup/types.rs:99-110) */
typedef struct dut_haplogroup_result {
    const char *name;       /* owned by the tree */
    double   score;
    uint32_t matching_snps, mismatching_snps, ancestral_matches, no_calls, total_snps, cumulative_snps, depth;
} dut_haplogroup_result;

/* calculate_haplogroup_score from the root (mod.rs:78-87) followed by collect_scored_paths
 * (mod.rs:196-258).  Rows that tie on (cumulative_snps, score) are ordered by name (the reference:
 * hash order).  calls must be sorted by position.  results: malloc'd (dut_free).
 * Errors: a called position whose locus has an empty derived/ancestral allele (the reference
 * panics on `.chars().next().unwrap()`). */
int dut_tree_score(const dut_tree *t, const dut_snp_call *calls, size_t n_calls, const char *build_id,
                   dut_haplogroup_result **results, size_t *n_results, char *err, size_t err_len);

/* The TSV of analyze_haplogroup (mod.rs:92-138), scores printed with {:.4}. */
int dut_write_haplogroup_report(const char *path, const dut_tree *t, const dut_haplogroup_result *results,
                                size_t n_results, const dut_snp_call *calls, size_t n_calls,
                                const char *build_id, int show_snps, char *err, size_t err_len);

/* validate_reference (validation.rs:5-35): genome from the header text, then the first candidate
 * name present among ref_names.  build_id gets genome.name() for Y, "rCRS" for MT (mod.rs:51-54).
 * Errors: "Could not determine reference genome from BAM header",
 *         "No valid sequence found in BAM. Tried: ...". */
int dut_validate_reference(const char *header_text, size_t len, const char *const *ref_names, size_t n_refs,
                           int tree_type, char *build_id, size_t build_len, char *chromosome, size_t chrom_len,
                           char *err, size_t err_len);

/* `find-y-branch` / `find-mt-branch` on files, one GPU: validate, load the tree, collect the sites,
 * read the chromosome's records, cl_site_pileup, call, score, write the TSV.  The BAM must have
 * its .bai (the reference opens an IndexedReader, mod.rs:46). */
int dut_find_branch_files(const char *bam_path, const char *fasta_path, const char *tree_json_path,
                          const char *output_path, uint32_t min_depth, uint8_t min_quality, int tree_type,
                          int provider, int show_snps, int device_id, char *err, size_t err_len);

#ifdef __cplusplus
}
#endif
#endif /* DUT_HAPLOGROUP_H */

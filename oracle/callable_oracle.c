/*
 * callable_oracle.c -- TEST INFRASTRUCTURE ONLY (see callable_oracle.h for the contract).
 *
 * Column-by-column, single-threaded CPU restatement of the reference's coverage path:
 *   src/callable_loci/mod.rs:17-147
 *   src/callable_loci/profilers/callable_profiler.rs:39-160
 *   src/callable_loci/profilers/contig_profiler.rs:47-83,93-103,126-157
 *   src/callable_loci/report.rs:26-126,339-393
 *   src/haplogroup/caller.rs:62-152
 * plus the htslib pileup engine those call through rust-htslib 0.49 (Cargo.toml:20), restated
 * from the published algorithm (htslib sam.c: bam_plp_push / bam_plp_next / bam_plp_auto /
 * resolve_cigar2) because htslib is not vendored under /root/reference: parity unpinned at that
 * boundary, pinned elsewhere by the KATs in tests/golden/.
 */
#define _POSIX_C_SOURCE 200809L
#include "callable_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* error helper                                                                               */
/* ------------------------------------------------------------------------------------------ */
static int fail(char *errbuf, size_t n, int code, const char *msg)
{
    if (errbuf && n) {
        strncpy(errbuf, msg, n - 1);
        errbuf[n - 1] = 0;
    }
    return code;
}

/* ------------------------------------------------------------------------------------------ */
/* htslib pileup engine (third party)                         */
/* ------------------------------------------------------------------------------------------ */

#define OP_M 0
#define OP_I 1
#define OP_D 2
#define OP_N 3
#define OP_S 4
#define OP_H 5
#define OP_P 6
#define OP_EQ 7
#define OP_X 8

static inline int op_is_match(int op) { return op == OP_M || op == OP_EQ || op == OP_X; }
static inline int op_is_refadv(int op) { return op_is_match(op) || op == OP_D || op == OP_N; }
static inline int op_is_qadv_gap(int op) { return op == OP_I || op == OP_S; }

/* bam_cigar2rlen: sum of M, D, N, =, X lengths */
static int64_t cigar_rlen(const uint32_t *cig, uint32_t n)
{
    int64_t l = 0;
    for (uint32_t k = 0; k < n; ++k)
        if (op_is_refadv((int)(cig[k] & 15))) l += cig[k] >> 4;
    return l;
}

typedef struct lbnode {
    int64_t ridx;            /* the record this node holds (bam_copy1 target) */
    int     tid;
    int64_t beg, end;        /* beg = pos, end = pos + bam_cigar2rlen (raw rlen) */
    int     k;               /* cstate_t: current cigar op index, -1 = never processed */
    int64_t x, y;            /* cstate_t: ref start / query start of op k */
    struct lbnode *next;
} lbnode;

typedef struct plp_aln {
    int64_t ridx;
    int64_t qpos;
    int     is_del;
    int     is_refskip;
} plp_aln;

typedef struct plp_iter {
    lbnode *head, *tail;
    lbnode *freelist;
    int64_t mp_cnt;          /* mempool count: nodes handed out and not yet returned */
    int     tid;
    int64_t pos;
    int     max_tid;
    int64_t max_pos;
    int     is_eof;
    int64_t maxcnt;
    int     error;           /* <0 once broken */
    plp_aln *col;
    int64_t  col_cap;
    const orc_reads *reads;
    int      contig_tid;     /* all records of one fetch share one tid */
} plp_iter;

static lbnode *mp_alloc(plp_iter *it)
{
    lbnode *p;
    ++it->mp_cnt;
    if (it->freelist) {
        p = it->freelist;
        it->freelist = p->next;
    } else {
        p = (lbnode *)calloc(1, sizeof(lbnode));
    }
    p->next = NULL;
    return p;
}

static void mp_free(plp_iter *it, lbnode *p)
{
    --it->mp_cnt;
    p->next = it->freelist;
    it->freelist = p;
}

/* bam_plp_init + bam_plp_set_maxcnt (mod.rs:55-60) */
static void plp_init(plp_iter *it, const orc_reads *reads, int contig_tid, int64_t maxcnt)
{
    memset(it, 0, sizeof(*it));
    it->reads = reads;
    it->contig_tid = contig_tid;
    it->head = it->tail = mp_alloc(it);   /* the list sentinel: mp_cnt == live nodes + 1 */
    it->max_tid = -1;
    it->max_pos = -1;
    it->maxcnt = maxcnt;
    it->tid = 0;
    it->pos = 0;
}

static void plp_destroy(plp_iter *it)
{
    lbnode *p = it->head;
    while (p) { lbnode *q = p->next; free(p); p = q; }
    p = it->freelist;
    while (p) { lbnode *q = p->next; free(p); p = q; }
    free(it->col);
}

/* bam_plp_push.  ridx < 0 means "no more records" (b == NULL).
 * *accepted (optional) = 1 if the record was appended to the list. */
static int plp_push(plp_iter *it, int64_t ridx, int *accepted)
{
    if (accepted) *accepted = 0;
    if (it->error) return -1;
    if (ridx < 0) { it->is_eof = 1; return 0; }
    const orc_reads *R = it->reads;
    const int tid = it->contig_tid;
    /* only unmapped reads are skipped here; rust-htslib installs no further filter */
    if (R->flag[ridx] & 0x4) return 0;
    const int64_t bpos = R->pos[ridx];
    if (it->tid == tid && it->pos == bpos && it->mp_cnt > it->maxcnt) return 0; /* depth cap */
    lbnode *t = it->tail;
    t->ridx = ridx;
    t->tid = tid;
    t->beg = bpos;
    t->end = bpos + cigar_rlen(R->cigar + R->cigar_off[ridx],
                               R->cigar_off[ridx + 1] - R->cigar_off[ridx]);
    t->k = -1; t->x = 0; t->y = 0;
    if (tid < it->max_tid) { it->error = -2; return -1; }
    if (tid == it->max_tid && t->beg < it->max_pos) { it->error = -2; return -1; } /* unsorted */
    it->max_tid = tid;
    it->max_pos = t->beg;
    if (t->end > it->pos || tid > it->tid) {
        t->next = mp_alloc(it);
        it->tail = t->next;
        if (accepted) *accepted = 1;
    }
    return 0;
}

/* resolve_cigar2: incremental CIGAR walk of one node at column `pos` */
static int resolve_cigar(const orc_reads *R, lbnode *p, int64_t pos, plp_aln *out)
{
    const uint32_t *cig = R->cigar + R->cigar_off[p->ridx];
    const int n_cigar = (int)(R->cigar_off[p->ridx + 1] - R->cigar_off[p->ridx]);
    int k;
    if (p->k == -1) {                       /* never processed */
        if (n_cigar == 1) {
            if (op_is_match((int)(cig[0] & 15))) { p->k = 0; p->x = p->beg; p->y = 0; }
            else return -3;                 /* htslib indexes cigar[-1] here: undefined */
        } else {
            p->x = p->beg; p->y = 0;
            for (k = 0; k < n_cigar; ++k) {
                int op = (int)(cig[k] & 15);
                int64_t l = cig[k] >> 4;
                if (op_is_refadv(op)) break;
                else if (op_is_qadv_gap(op)) p->y += l;
            }
            if (k >= n_cigar) return -3;
            p->k = k;
        }
    } else {
        int64_t l = cig[p->k] >> 4;
        if (pos - p->x >= l) {              /* jump to the next reference-consuming op */
            if (p->k + 1 >= n_cigar) return -3;
            if (op_is_match((int)(cig[p->k] & 15))) p->y += l;
            p->x += l;
            for (k = p->k + 1; k < n_cigar; ++k) {
                int op = (int)(cig[k] & 15);
                int64_t ll = cig[k] >> 4;
                if (op_is_refadv(op)) break;
                else if (op_is_qadv_gap(op)) p->y += ll;
            }
            if (k >= n_cigar) return -3;
            p->k = k;
        }
    }
    {
        int op = (int)(cig[p->k] & 15);
        out->ridx = p->ridx;
        out->is_del = 0; out->is_refskip = 0;
        if (op_is_match(op)) {
            out->qpos = p->y + (pos - p->x);
        } else {                            /* D or N */
            out->is_del = 1;
            out->qpos = p->y;
            out->is_refskip = (op == OP_N);
        }
    }
    return 0;
}

/* bam_plp_next: returns number of alignments in the next non-empty column (>0), 0 if no column
 * can be produced yet / any more, <0 on error.  Column position in *cpos. */
static int64_t plp_next(plp_iter *it, int *ctid, int64_t *cpos)
{
    if (it->error) return -1;
    if (it->is_eof && it->head == it->tail) return 0;
    while (it->is_eof || it->max_tid > it->tid ||
           (it->max_tid == it->tid && it->max_pos > it->pos)) {
        int64_t n = 0;
        lbnode **pptr = &it->head;
        while (*pptr != it->tail) {
            lbnode *p = *pptr;
            if (p->tid < it->tid || (p->tid == it->tid && p->end <= it->pos)) {
                *pptr = p->next;
                mp_free(it, p);
            } else {
                if (p->tid == it->tid && p->beg <= it->pos) {
                    if (n == it->col_cap) {
                        it->col_cap = it->col_cap ? it->col_cap * 2 : 256;
                        it->col = (plp_aln *)realloc(it->col, (size_t)it->col_cap * sizeof(plp_aln));
                    }
                    int rc = resolve_cigar(it->reads, p, it->pos, &it->col[n]);
                    if (rc < 0) { it->error = rc; return -1; }
                    ++n;
                }
                pptr = &(*pptr)->next;
            }
        }
        *ctid = it->tid; *cpos = it->pos;
        if (it->head != it->tail) {
            if (it->tid > it->head->tid) { it->error = -2; return -1; }
            if (it->tid < it->head->tid) { it->tid = it->head->tid; it->pos = it->head->beg; }
            else if (it->pos < it->head->beg) it->pos = it->head->beg;
            else ++it->pos;
        } else {
            /* empty list: htslib inspects the sentinel's stale record here, whose tid/beg are
             * never ahead of the cursor, so the cursor just steps */
            ++it->pos;
        }
        if (n) return n;
        if (it->is_eof && it->head == it->tail) break;
    }
    return 0;
}

/* bam_plp_auto driven by the region iterator of `bam.fetch((tid, 0, contig_len))`
 * (mod.rs:53-55): records with pos < contig_len, in file order. */
typedef struct plp_driver {
    plp_iter it;
    int64_t next_rec;
    int64_t n_rec;
    uint32_t contig_len;
    int done_eof;
    uint8_t *accepted;       /* optional */
} plp_driver;

static int64_t driver_read(plp_driver *d)
{
    const orc_reads *R = d->it.reads;
    while (d->next_rec < d->n_rec) {
        int64_t i = d->next_rec++;
        if ((int64_t)R->pos[i] < (int64_t)d->contig_len) return i;   /* overlaps [0,len) */
    }
    return -1;
}

static int64_t plp_auto(plp_driver *d, int *ctid, int64_t *cpos)
{
    plp_iter *it = &d->it;
    if (it->error) return -1;
    int64_t n = plp_next(it, ctid, cpos);
    if (n != 0) return n;
    if (it->is_eof) return 0;
    for (;;) {
        int64_t i = driver_read(d);
        if (i < 0) break;
        int acc = 0;
        if (plp_push(it, i, &acc) < 0) return -1;
        if (d->accepted && acc) d->accepted[i] = 1;
        n = plp_next(it, ctid, cpos);
        if (n != 0) return n;
    }
    if (plp_push(it, -1, NULL) < 0) return -1;
    return plp_next(it, ctid, cpos);
}

/* ------------------------------------------------------------------------------------------ */
/* exact byte-string set for read names (contig_profiler.rs:59-62 HashSet<Vec<u8>>)           */
/* ------------------------------------------------------------------------------------------ */
typedef struct nameset {
    uint64_t *hash;      /* 0 = empty */
    int64_t  *ridx;
    uint64_t  cap, cnt;
} nameset;

static uint64_t hash_bytes(const uint8_t *s, uint32_t n)
{
    uint64_t h = 0xcbf29ce484222325ULL;
    for (uint32_t i = 0; i < n; ++i) { h ^= s[i]; h *= 0x100000001b3ULL; }
    h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ULL; h ^= h >> 32;
    return h ? h : 1;
}

static void nameset_grow(nameset *s)
{
    uint64_t ncap = s->cap ? s->cap * 2 : 1024;
    uint64_t *nh = (uint64_t *)calloc(ncap, sizeof(uint64_t));
    int64_t *nr = (int64_t *)malloc(ncap * sizeof(int64_t));
    for (uint64_t i = 0; i < s->cap; ++i) if (s->hash[i]) {
        uint64_t j = s->hash[i] & (ncap - 1);
        while (nh[j]) j = (j + 1) & (ncap - 1);
        nh[j] = s->hash[i]; nr[j] = s->ridx[i];
    }
    free(s->hash); free(s->ridx);
    s->hash = nh; s->ridx = nr; s->cap = ncap;
}

/* returns 1 if newly inserted */
static int nameset_insert(nameset *s, const orc_reads *R, int64_t ridx)
{
    if ((s->cnt + 1) * 2 > s->cap) nameset_grow(s);
    const uint8_t *nm = R->qname + R->qname_off[ridx];
    uint32_t nl = R->qname_off[ridx + 1] - R->qname_off[ridx];
    uint64_t h = hash_bytes(nm, nl);
    uint64_t j = h & (s->cap - 1);
    while (s->hash[j]) {
        if (s->hash[j] == h) {
            int64_t o = s->ridx[j];
            uint32_t ol = R->qname_off[o + 1] - R->qname_off[o];
            if (ol == nl && memcmp(R->qname + R->qname_off[o], nm, nl) == 0) return 0;
        }
        j = (j + 1) & (s->cap - 1);
    }
    s->hash[j] = h; s->ridx[j] = ridx; ++s->cnt;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* CallableProfiler (callable_profiler.rs)                                                    */
/* ------------------------------------------------------------------------------------------ */
static const char *const STATE_NAMES[6] = {       /* Debug names, types.rs:36-43 */
    "REF_N", "CALLABLE", "NO_COVERAGE", "LOW_COVERAGE", "EXCESSIVE_COVERAGE",
    "POOR_MAPPING_QUALITY"
};

typedef struct contig_counts {
    char *name;
    uint64_t c[6];
} contig_counts;

struct orc_profiler {
    FILE *bed;
    int has_state;                   /* current_state: Option<(String,u64,u64,CalledState)> */
    char *cur_contig;
    uint64_t cur_start, cur_end;
    int cur_state;
    contig_counts *cc;               /* contig_counts: HashMap<String,[u64;6]> */
    size_t n_cc, cap_cc;
};

orc_profiler *orc_profiler_new(const char *bed_path)
{
    orc_profiler *p = (orc_profiler *)calloc(1, sizeof(*p));
    p->bed = fopen(bed_path, "wb");                 /* File::create, callable_profiler.rs:31 */
    if (!p->bed) { free(p); return NULL; }
    setvbuf(p->bed, NULL, _IOFBF, 1 << 20);
    return p;
}

void orc_profiler_free(orc_profiler *p)
{
    if (!p) return;
    if (p->bed) fclose(p->bed);
    free(p->cur_contig);
    for (size_t i = 0; i < p->n_cc; ++i) free(p->cc[i].name);
    free(p->cc);
    free(p);
}

void orc_profiler_contig_counts(const orc_profiler *p, const char *contig, uint64_t out[6])
{
    memset(out, 0, 6 * sizeof(uint64_t));           /* unwrap_or([0;6]), :158-160 */
    for (size_t i = 0; i < p->n_cc; ++i)
        if (strcmp(p->cc[i].name, contig) == 0) { memcpy(out, p->cc[i].c, sizeof(p->cc[i].c)); return; }
}

/* write_state, callable_profiler.rs:39-62 (the coverage_ranges push feeds only the SVG) */
static void prof_write_state(orc_profiler *p)
{
    if (p->has_state)
        fprintf(p->bed, "%s\t%llu\t%llu\t%s\n", p->cur_contig,
                (unsigned long long)p->cur_start, (unsigned long long)p->cur_end,
                STATE_NAMES[p->cur_state]);
}

static void prof_set_state(orc_profiler *p, const char *contig, uint64_t start, uint64_t end,
                           int state)
{
    if (!p->cur_contig || strcmp(p->cur_contig, contig) != 0) {
        free(p->cur_contig);
        p->cur_contig = strdup(contig);
    }
    p->cur_start = start; p->cur_end = end; p->cur_state = state; p->has_state = 1;
}

/* process_state, callable_profiler.rs:122-155 */
static void prof_process_state(orc_profiler *p, contig_counts *cc, const char *contig,
                               uint64_t pos, int state)
{
    cc->c[state] += 1;                                          /* :124-126 */
    if (!p->has_state) {                                        /* :128-141 */
        if (state == ORC_REF_N) {
            prof_set_state(p, contig, 0, pos + 1, state);
        } else {
            if (pos > 0) {
                prof_set_state(p, contig, 0, pos, ORC_REF_N);
                prof_write_state(p);
            }
            prof_set_state(p, contig, pos, pos + 1, state);
        }
        return;
    }
    if (strcmp(p->cur_contig, contig) == 0 && p->cur_state == state) {
        p->cur_end = pos + 1;                                   /* :144-146 */
    } else {
        prof_write_state(p);                                    /* :147-151 */
        prof_set_state(p, contig, pos, pos + 1, state);
    }
}

static contig_counts *prof_counts_entry(orc_profiler *p, const char *contig)
{
    for (size_t i = 0; i < p->n_cc; ++i)
        if (strcmp(p->cc[i].name, contig) == 0) return &p->cc[i];
    if (p->n_cc == p->cap_cc) {
        p->cap_cc = p->cap_cc ? p->cap_cc * 2 : 32;
        p->cc = (contig_counts *)realloc(p->cc, p->cap_cc * sizeof(contig_counts));
    }
    contig_counts *e = &p->cc[p->n_cc++];
    e->name = strdup(contig);
    memset(e->c, 0, sizeof(e->c));
    return e;
}

/* CallableProfiler::process_position, callable_profiler.rs:89-120; returns the state */
static int prof_process_position(orc_profiler *p, const char *contig, uint32_t pos,
                                 uint8_t ref_base, uint32_t raw_depth, uint32_t qc_depth,
                                 uint32_t low_mapq_count, const orc_options *o)
{
    int is_low_mapq = raw_depth >= o->min_depth_for_low_mapq &&
        ((double)low_mapq_count / (double)raw_depth) > o->max_low_mapq_fraction;   /* :100-101 */
    int state;
    if (ref_base == 'N' || ref_base == 'n') state = ORC_REF_N;                     /* :104 */
    else if (raw_depth == 0) state = ORC_NO_COVERAGE;
    else if (is_low_mapq) state = ORC_POOR_MAPPING_QUALITY;
    else if (qc_depth < o->min_depth) state = ORC_LOW_COVERAGE;
    else if (o->max_depth > 0 && qc_depth > o->max_depth) state = ORC_EXCESSIVE_COVERAGE;
    else state = ORC_CALLABLE;
    /* entry(contig).or_insert([0;6]) happens on every call (:124); look it up once per call */
    prof_process_state(p, prof_counts_entry(p, contig), contig, (uint64_t)pos, state);
    return state;
}

/* ------------------------------------------------------------------------------------------ */
/* process_single_contig (mod.rs:44-147)                                                      */
/* ------------------------------------------------------------------------------------------ */
static inline uint8_t fetch_base(const uint8_t *ref, uint64_t ref_len, uint64_t p)
{
    /* fasta.fetch_seq(contig, p, p) -> first byte or b'N' when empty (mod.rs:79-80) */
    return (ref && p < ref_len) ? ref[p] : (uint8_t)'N';
}

int orc_process_single_contig(orc_profiler *prof, orc_contig_stats *stats,
                              const orc_options *opt, const char *contig_name, int32_t tid,
                              uint32_t contig_len, const uint8_t *ref, uint64_t ref_len,
                              const orc_reads *reads,
                              uint32_t *dbg_raw, uint32_t *dbg_qc, uint32_t *dbg_low,
                              uint8_t *dbg_state, uint64_t dbg_cap, uint64_t *dbg_extent,
                              char *errbuf, size_t errbuf_len)
{
    plp_driver d;
    nameset names;
    memset(&names, 0, sizeof(names));
    memset(&d, 0, sizeof(d));
    /* pileup.set_max_depth(if max_depth > 0 { max_depth } else { 500 })  mod.rs:56-60 */
    plp_init(&d.it, reads, /*contig_tid=*/tid, opt->max_depth > 0 ? (int64_t)opt->max_depth : 500);
    d.n_rec = reads ? reads->n : 0;
    d.contig_len = contig_len;

    uint32_t current_pos = 0;
    int rc = 0;
    int ctid = 0; int64_t cpos = 0;
    const orc_reads *R = reads;

#define DBG(p_, raw_, qc_, low_, st_) do { \
        if ((uint64_t)(p_) < dbg_cap) { \
            if (dbg_raw) { dbg_raw[p_] = (raw_); } if (dbg_qc) { dbg_qc[p_] = (qc_); } \
            if (dbg_low) { dbg_low[p_] = (low_); } if (dbg_state) { dbg_state[p_] = (uint8_t)(st_); } } \
    } while (0)

    for (;;) {
        int64_t n = plp_auto(&d, &ctid, &cpos);
        if (n < 0) {
            rc = fail(errbuf, errbuf_len, d.it.error == -2 ? -2 : -3,
                      d.it.error == -2 ? "pileup: the input is not sorted"
                                       : "pileup: malformed CIGAR (no reference-consuming match)");
            break;
        }
        if (n == 0) break;
        if (ctid != tid) break;                                                /* :67-69 */
        uint32_t pos = (uint32_t)cpos;                                         /* :71 */

        while (current_pos < pos) {                                            /* :74-93 */
            uint8_t rb = fetch_base(ref, ref_len, current_pos);
            int st = prof_process_position(prof, contig_name, current_pos, rb, 0, 0, 0, opt);
            DBG(current_pos, 0, 0, 0, st);
            current_pos += 1;
        }

        uint8_t ref_base = fetch_base(ref, ref_len, pos);                      /* :100-101 */

        /* process_position, mod.rs:17-42 */
        uint32_t raw_depth = 0, qc_depth = 0, low_mapq_count = 0;
        for (int64_t a = 0; a < n; ++a) {
            const plp_aln *al = &d.it.col[a];
            raw_depth += 1;
            uint8_t mq = R->mapq[al->ridx];
            if (mq <= opt->max_low_mapq) low_mapq_count += 1;
            if (mq >= opt->min_mapping_quality) {
                if (!(al->is_del || al->is_refskip)) {                         /* qpos() is Some */
                    uint64_t qlen = R->qual_off[al->ridx + 1] - R->qual_off[al->ridx];
                    if ((uint64_t)al->qpos < qlen) {                           /* qual().get(qpos) */
                        uint8_t q = R->qual[R->qual_off[al->ridx] + (uint64_t)al->qpos];
                        if (q >= opt->min_base_quality || al->is_del) qc_depth += 1;
                    }
                }
            }
        }

        int st = prof_process_position(prof, contig_name, pos, ref_base, raw_depth, qc_depth,
                                       low_mapq_count, opt);                   /* :105-113 */
        DBG(pos, raw_depth, qc_depth, low_mapq_count, st);

        /* ContigProfiler::process_position, contig_profiler.rs:47-83 */
        if (stats) {
            for (int64_t a = 0; a < n; ++a) {
                const plp_aln *al = &d.it.col[a];
                uint8_t mq = R->mapq[al->ridx];
                if (nameset_insert(&names, R, al->ridx)) stats->n_reads += 1;  /* :59-62 */
                if (mq >= opt->min_mapping_quality) {
                    if (!(al->is_del || al->is_refskip)) {
                        uint64_t qlen = R->qual_off[al->ridx + 1] - R->qual_off[al->ridx];
                        if ((uint64_t)al->qpos < qlen) {
                            uint8_t q = R->qual[R->qual_off[al->ridx] + (uint64_t)al->qpos];
                            if (q >= opt->min_base_quality) {
                                stats->summed_baseq += q;                      /* :68-70 */
                                stats->quality_bases += 1;
                            }
                        }
                    }
                    stats->summed_mapq += mq;                                  /* :74 */
                    stats->n_selected_reads += 1;                              /* :75 (u32 wrap) */
                }
            }
            if (raw_depth > 0) {                                               /* :79-82 */
                stats->n_covered_bases += 1;
                stats->summed_coverage += raw_depth;
            }
        }
        current_pos = pos + 1;                                                 /* :119 */
    }

    if (rc == 0) {
        while (current_pos < contig_len) {                                     /* :123-142 */
            uint8_t rb = fetch_base(ref, ref_len, current_pos);
            int st = prof_process_position(prof, contig_name, current_pos, rb, 0, 0, 0, opt);
            DBG(current_pos, 0, 0, 0, st);
            current_pos += 1;
        }
        /* counter.finish_contig -> write_state() WITHOUT clearing current_state (:64-66);
         * the SVG branch is presentation and out of scope */
        prof_write_state(prof);
        if (dbg_extent) *dbg_extent = current_pos;
    }
#undef DBG
    free(names.hash); free(names.ridx);
    plp_destroy(&d.it);
    return rc;
}

int orc_accepted_reads(const orc_options *opt, int32_t tid, uint32_t contig_len,
                       const orc_reads *reads, uint8_t *accepted, char *errbuf, size_t errbuf_len)
{
    plp_driver d;
    memset(&d, 0, sizeof(d));
    plp_init(&d.it, reads, tid, opt->max_depth > 0 ? (int64_t)opt->max_depth : 500);
    d.n_rec = reads->n;
    d.contig_len = contig_len;
    d.accepted = accepted;
    memset(accepted, 0, (size_t)reads->n);
    int ctid; int64_t cpos; int rc = 0;
    for (;;) {
        int64_t n = plp_auto(&d, &ctid, &cpos);
        if (n < 0) { rc = fail(errbuf, errbuf_len, d.it.error, "pileup error"); break; }
        if (n == 0) break;
    }
    plp_destroy(&d.it);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* derived statistics                                                                         */
/* ------------------------------------------------------------------------------------------ */
void orc_contig_derive(const orc_contig_stats *s, orc_contig_derived *out)
{
    /* report.rs:44-54 (same expressions as contig_profiler.rs:93-103) */
    out->coverage_percent = s->length > 0
        ? ((double)s->n_covered_bases / (double)s->length) * 100.0 : 0.0;
    out->average_depth = s->n_covered_bases > 0
        ? (double)s->summed_coverage / (double)s->n_covered_bases : 0.0;
    /* contig_profiler.rs:126-157 */
    out->average_mapq = s->quality_bases > 0
        ? (double)s->summed_mapq / (double)s->quality_bases : 0.0;
    out->average_baseq = s->quality_bases > 0
        ? (double)s->summed_baseq / (double)s->quality_bases : 0.0;
    if (s->quality_bases > 0) {
        if (out->average_baseq >= 30.0) out->q30_percentage = 100.0;
        else if (out->average_baseq < 20.0) out->q30_percentage = 0.0;
        else out->q30_percentage = ((out->average_baseq - 20.0) / 10.0) * 100.0;
    } else out->q30_percentage = 0.0;
}

/* split_contig_name, report.rs:385-393: first ASCII digit or 'X' 'Y' 'M' */
static size_t split_pos(const char *s)
{
    size_t i = 0;
    for (; s[i]; ++i) {
        char c = s[i];
        if ((c >= '0' && c <= '9') || c == 'X' || c == 'Y' || c == 'M') break;
    }
    return i;
}

/* order(), report.rs:355-369: u32 parse (accepts a leading '+', rejects overflow/empty) */
static void suffix_order(const char *s, int *cat, uint32_t *num)
{
    const char *t = s;
    if (*t == '+') ++t;
    if (*t) {
        uint64_t v = 0; int ok = 1;
        for (const char *q = t; *q; ++q) {
            if (*q < '0' || *q > '9') { ok = 0; break; }
            v = v * 10 + (uint64_t)(*q - '0');
            if (v > 0xFFFFFFFFull) { ok = 0; break; }
        }
        if (ok) { *cat = 0; *num = (uint32_t)v; return; }
    }
    *num = 0;
    if (strcmp(s, "X") == 0) *cat = 1;
    else if (strcmp(s, "Y") == 0) *cat = 2;
    else if (strcmp(s, "M") == 0 || strcmp(s, "MT") == 0) *cat = 3;
    else *cat = 4;
}

int orc_compare_contig_names(const char *a, const char *b)
{
    size_t sa = split_pos(a), sb = split_pos(b);
    /* a_prefix.cmp(b_prefix): bytewise lexicographic, shorter first on common prefix */
    size_t m = sa < sb ? sa : sb;
    int c = memcmp(a, b, m);
    if (c != 0) return c;
    if (sa != sb) return sa < sb ? -1 : 1;
    int ca, cb; uint32_t na, nb;
    suffix_order(a + sa, &ca, &na);
    suffix_order(b + sb, &cb, &nb);
    if (ca != cb) return ca < cb ? -1 : 1;
    if (ca == 0) return na < nb ? -1 : (na > nb ? 1 : 0);
    return strcmp(a + sa, b + sb);
}

void orc_genome_summary_build(const orc_contig_stats *stats, const uint64_t *callable,
                              size_t n_contigs, orc_genome_summary *out)
{
    /* report.rs:26-33 */
    uint64_t total_bases = 0, callable_bases = 0, q30_bases = 0, total_quality_positions = 0,
             total_unique_reads = 0;
    double total_depth = 0.0, total_mapq = 0.0, total_baseq = 0.0;
    for (size_t i = 0; i < n_contigs; ++i) {                    /* :40-63 */
        orc_contig_derived d;
        orc_contig_derive(&stats[i], &d);
        total_bases += stats[i].length;
        callable_bases += callable[i];
        total_depth += d.average_depth * (double)stats[i].length;
        total_mapq += d.average_mapq * (double)stats[i].length;
        total_baseq += d.average_baseq * (double)stats[i].length;
        {
            /* `as u64`: saturating truncation toward zero (:61) */
            double v = d.q30_percentage / 100.0 * (double)stats[i].length;
            uint64_t t = v <= 0.0 ? 0 : (v >= 18446744073709551615.0 ? UINT64_MAX : (uint64_t)v);
            q30_bases += t;
        }
        total_quality_positions += stats[i].length;
        total_unique_reads += stats[i].n_reads;
    }
    out->total_bases = total_bases;
    out->callable_bases = callable_bases;
    out->callable_percentage = total_bases > 0
        ? ((double)callable_bases / (double)total_bases) * 100.0 : 0.0;     /* :101-105 */
    out->average_depth = total_bases > 0 ? total_depth / (double)total_bases : 0.0; /* :88-92 */
    out->average_mapq = total_quality_positions > 0
        ? total_mapq / (double)total_quality_positions : 0.0;               /* :111-115 */
    out->average_baseq = total_quality_positions > 0
        ? total_baseq / (double)total_quality_positions : 0.0;              /* :116-120 */
    out->q30_percentage = total_quality_positions > 0
        ? ((double)q30_bases / (double)total_quality_positions) * 100.0 : 0.0; /* :121-125 */
    out->total_unique_reads = total_unique_reads;
    out->contigs_analyzed = n_contigs;                                      /* :107 */
}

/* ------------------------------------------------------------------------------------------ */
/* config 5: haplogroup::caller::process_region (src/haplogroup/caller.rs:62-152)             */
/* ------------------------------------------------------------------------------------------ */
int orc_site_pileup(uint32_t min_depth, uint8_t min_quality, uint32_t contig_len,
                    const uint8_t *ref, uint64_t ref_len,
                    const orc_reads *R, const uint64_t *seq_off, const uint8_t *seq4,
                    const uint32_t *sites, size_t n_sites,
                    uint32_t *out_total, uint8_t *out_base, uint32_t *out_count,
                    uint8_t *out_called, double *out_freq, uint32_t *base_hist)
{
    static const char CODE[] = "=ACMGRSVTWYHKDBN";      /* rust-htslib seq().as_bytes() */
    /* positions: HashMap<u32, ...> keyed by 1-based vcf_pos -> dense lookup position->site */
    uint64_t maxp = 0;
    for (size_t i = 0; i < n_sites; ++i) if (sites[i] > maxp) maxp = sites[i];
    int64_t *lut = (int64_t *)malloc((size_t)(maxp + 2) * sizeof(int64_t));
    for (uint64_t i = 0; i <= maxp + 1; ++i) lut[i] = -1;
    for (size_t i = 0; i < n_sites; ++i) lut[sites[i]] = (int64_t)i;
    uint32_t *hist = (uint32_t *)calloc(n_sites * 16, sizeof(uint32_t));

    /* fetch("chr:1-len") (:33-36): records overlapping [0,len) on this contig, NO flag filter */
    for (int64_t r = 0; r < R->n; ++r) {
        if ((int64_t)R->pos[r] >= (int64_t)contig_len) continue;
        if (R->mapq[r] < min_quality) continue;                               /* :80 */
        uint64_t slen = seq_off[r + 1] - seq_off[r];
        uint64_t ref_pos = (uint64_t)R->pos[r];
        uint64_t read_pos = 0;
        const uint32_t *cig = R->cigar + R->cigar_off[r];
        uint32_t nc = R->cigar_off[r + 1] - R->cigar_off[r];
        for (uint32_t k = 0; k < nc; ++k) {
            int op = (int)(cig[k] & 15);
            uint64_t len = cig[k] >> 4;
            if (op_is_match(op)) {                                            /* :91-119 */
                for (uint64_t i = 0; i < len; ++i) {
                    uint64_t vcf_pos = ref_pos + 1;
                    if (vcf_pos <= maxp && lut[vcf_pos] >= 0) {
                        if (read_pos + i < slen) {                            /* :105 */
                            uint64_t bi = seq_off[r] + read_pos + i;
                            uint8_t byte = seq4[bi >> 1];
                            int code = (bi & 1) ? (byte & 15) : (byte >> 4);
                            /* fetch_seq(ref_pos, ref_pos) must be non-empty (:110-113) */
                            if (ref && ref_pos < ref_len) hist[lut[vcf_pos] * 16 + code] += 1;
                        }
                    }
                    ref_pos += 1;
                }
                read_pos += len;
            } else if (op == OP_D || op == OP_N) {
                ref_pos += len;                                               /* :120-122 */
            } else if (op == OP_I || op == OP_S) {
                read_pos += len;                                              /* :123-125 */
            }
        }
    }
    for (size_t i = 0; i < n_sites; ++i) {                                    /* :132-149 */
        uint32_t total = 0, best = 0; int bc = 0;
        /* to_ascii_uppercase of "=ACMGRSVTWYHKDBN" is the identity, so codes <-> chars 1:1 */
        for (int c = 0; c < 16; ++c) {
            uint32_t v = hist[i * 16 + c];
            total += v;
            if (v > best) { best = v; bc = c; }
        }
        out_total[i] = total;
        out_count[i] = best;
        out_base[i] = total ? (uint8_t)CODE[bc] : 0;
        double freq = total ? (double)best / (double)total : 0.0;
        out_freq[i] = freq;
        out_called[i] = (total >= min_depth && total > 0 && freq >= 0.7) ? 1 : 0;
        if (base_hist) memcpy(base_hist + i * 16, hist + i * 16, 16 * sizeof(uint32_t));
    }
    free(hist); free(lut);
    return 0;
}

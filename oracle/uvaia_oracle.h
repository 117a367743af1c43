/*
 * uvaia_oracle.h -- CPU restatement of uvaia's nearest-neighbour hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / reported CPU baseline.  The product (uvaia_amd/, include/) never links or calls it.
 *
 * What it restates (reference = quadram-institute-bioscience/uvaia @ 2024_10_08, paths under
 * /root/reference):
 *   - character classes                      src/utils.c:255-295
 *   - ACGT 2-count scoring kernel            src/fastaseq.c:585-596
 *   - 4-count scoring kernel                 biomcmc-lib (ABSENT: empty submodule, pinned version
 *                                            unknown); semantics from its call sites
 *                                            src/nearest.c:432,491-496 and README.md:249-259,307-316
 *   - valid-site count                       src/fastaseq.c:642-648
 *   - query structure, indices, ordering,    src/fastaseq.c:698-841, src/utils.c:10-48
 *     redundancy pruning, quality filter
 *   - bounded "keep k best" heap             src/min_heap.c:41-158
 *   - batch queue + gate + heap update       src/nearest.c:237-319,367-510
 *   - radius search                          src/fastaseq.c:660-696, src/ball.c:201-259
 *
 * Parity pinning: the scoring kernels are pinned by the reference's own known-answer rows
 * (README.md:227-233 on sequences of data/03.unique_acgt.aln.xz; README.md:307-316 toy example),
 * see tests/test_oracle_kat.py.  The reference itself cannot be compiled here (every src/ file
 * includes <biomcmc.h> from the empty submodule) so no reference-run outputs exist; the gate/heap
 * state machine is restated from the present sources line by line and is pinned only by those
 * sources (file:line cited at each function).
 */
#ifndef UVAIA_ORACLE_H
#define UVAIA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NSCORE 6

/* ---- character classes (src/utils.c:255-295) ---- */
int orc_is_acgt (unsigned char c);
int orc_is_valid (unsigned char c);       /* not in {N,X,-,?,O,.} (either case) */
int orc_iupac_mask (unsigned char c);     /* A=1 C=2 G=4 T=8, ambiguity codes = unions, else 0 */

/* ---- scoring kernels ---- */
/* src/fastaseq.c:585-596 : score[0]=#(both ACGT, differ)  score[1]=#(both ACGT); stops once score[0]>=maxdist */
void orc_score_acgt_and_valid (const char *s1, const char *s2, size_t n, int maxdist, int *score, const size_t *idx);
/* biomcmc_pairwise_score_matches_truncated_idx (absent): r[0]=#(equal & ACGT) r[1]=#(equal & both valid)
 * r[2]=#(IUPAC sets intersect & both valid) r[3]=#(both valid); stops once r[3]-r[0]>=maxdist (src/nearest.c:492) */
void orc_score_matches_truncated_idx (const char *s1, const char *s2, size_t n, int maxdist, int *r, const size_t *idx);
/* src/fastaseq.c:576-583 and :562-574 (radius search distances) */
void orc_dist_acgt (const char *s1, const char *s2, size_t n, int maxdist, int *score, const size_t *idx);
void orc_dist_text_indelcheck (const char *s1, const char *s2, size_t n, int maxdist, int *score, const size_t *idx);
int  orc_count_non_N (const char *s, size_t n);   /* src/fastaseq.c:642-648 */
int  orc_count_acgt (const char *s, size_t n);    /* src/fastaseq.c:650-656 */

/* ---- heap (src/min_heap.h:14-31, src/min_heap.c:52-158) ---- */
typedef struct {
  int score[ORC_NSCORE];
  char *name;
  int64_t ordinal;      /* oracle-only bookkeeping: position of the reference in the input stream */
} orc_item;

typedef struct {
  orc_item *seq;        /* slots 1..n form the heap; root (slot 1) = worst kept item */
  int heap_size, n;
  int max_incompatible;
} orc_heap;

int  orc_compare_score (const int *a, const int *b);
orc_heap *orc_heap_new (int heap_size);
void orc_heap_del (orc_heap *h);
int  orc_heap_insert (orc_heap *h, const orc_item *item);
void orc_heap_finalise (orc_heap *h);

/* ---- query structure (src/fastaseq.h:41-48) ---- */
typedef struct {
  int ntax, nchar;
  char **seq, **name;
  char *consensus;
  size_t *idx_c, *idx_m, *idx, trim;
  int n_idx_c, n_idx_m, n_idx, dist, acgt;
} orc_query;

/* seqs: ntax strings of nchar bytes each (copied; upper-cased as the reference's reader does) */
orc_query *orc_query_new (int ntax, int nchar, const char *const *seqs, const char *const *names,
                          int trim, int dist, int acgt);
void orc_query_del (orc_query *q);
int  orc_query_keep_valid (orc_query *q, double ambiguity);       /* src/utils.c:10-48; returns ntax kept */
void orc_query_create_indices (orc_query *q);                      /* src/fastaseq.c:732-777 */
void orc_query_reorder (orc_query *q);                             /* src/fastaseq.c:779-795 */
void orc_query_exclude_redundant (orc_query *q, int keep_more_resolved); /* src/fastaseq.c:797-841 */
/* the whole preparation sequence of src/nearest.c:203-224 (is_ball=0) or src/ball.c:174-194 (is_ball=1) */
orc_query *orc_query_prepare (int ntax, int nchar, const char *const *seqs, const char *const *names,
                              int trim, int dist, int acgt, double ambig_q, int keep_resolved, int is_ball);

/* ---- nearest-neighbour search (src/nearest.c main loop) ---- */
typedef struct orc_search orc_search;

/* pool = --pool, nbest = --nbest, ambig_r = -A ; exclude_self = -x */
orc_search *orc_search_new (orc_query *q, int pool, int nbest, double ambig_r, int exclude_self);
void orc_search_del (orc_search *s);
/* feed reference sequences in stream order; may be called repeatedly (batches are formed internally,
 * exactly as the reference forms them: only sequences that pass the filters occupy pool slots).
 * Sequences are nchar bytes, already upper-case.  Returns 0, or -1 on a length mismatch. */
int  orc_search_feed (orc_search *s, int n, const char *const *seqs, const char *const *names, const int *lengths);
void orc_search_end_of_file (orc_search *s);   /* boundary between two -r files */
/* multi-rank ring protocol tests only: one slice of a stripe as a batch with the stripe's snapshot (<0: from own state) */
int  orc_search_process_slice (orc_search *s, int n, const char *const *seqs, const char *const *names, const int64_t *ordinals, int snapshot);
int  orc_search_process_slice_range (orc_search *s, int n, const char *const *seqs, const char *const *names, const int64_t *ordinals, int snapshot, int q0, int q1);
int  orc_search_max_T (const orc_search *s);
void orc_search_get_state_range (const orc_search *s, int *blob, int q0, int q1);
void orc_search_set_state_range (orc_search *s, const int *blob, const char *name_prefix, int q0, int q1);
int  orc_search_last_snapshot (const orc_search *s);
size_t orc_search_state_ints (const orc_search *s);
void orc_search_get_state (const orc_search *s, int *blob);
void orc_search_set_state (orc_search *s, const int *blob, const char *name_prefix);
/* flush the last partial batch, sort heaps (src/nearest.c:343-344 -> :513-547) */
void orc_search_finish (orc_search *s);
/* results (valid after finish) */
int  orc_search_nrows (const orc_search *s, int iq);     /* rows the reference would print for query iq */
int  orc_search_heap_n (const orc_search *s, int iq);    /* items actually stored */
const orc_item *orc_search_row (const orc_search *s, int iq, int rank0);
int64_t orc_search_n_saved (const orc_search *s);        /* references written to the .aln dump */
const int64_t *orc_search_saved_ordinals (const orc_search *s);
int64_t orc_search_n_seen (const orc_search *s);
int64_t orc_search_n_lowqual (const orc_search *s);
int64_t orc_search_n_samename (const orc_search *s);
int  orc_search_final_T (const orc_search *s, int iq);   /* heap[iq]->max_incompatible at the end */
void orc_set_threads (int n);
int  orc_max_threads (void);

/* untruncated all-pairs counts for kernel parity checks (not a reference function: it calls the
 * restated kernels with maxdist = INT_MAX over the three index classes and sums them, exactly the
 * quantities src/nearest.c:499-501 or :464-469 would assemble when nothing is truncated).
 * out: [n_ref][ntax][6] */
void orc_allpairs_scores (const orc_query *q, int n_ref, const char *const *refs, int *out);

/* ---- radius search (src/ball.c:201-259 + src/fastaseq.c:660-696) ---- */
/* mindist[i] as left in cq->mindist[c] by the reference; keep[i] = (mindist<=dist) and the non_n filter passed */
void orc_ball (const orc_query *q, double ambig_r, int n_ref, const char *const *refs,
               int *mindist, unsigned char *keep);

#ifdef __cplusplus
}
#endif
#endif

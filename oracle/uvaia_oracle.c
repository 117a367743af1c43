/*
 * uvaia_oracle.c -- CPU restatement of uvaia's nearest-neighbour hot path (see uvaia_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into, nor called by, the product library.
 * All citations are path:line under /root/reference.
 */
#include "uvaia_oracle.h"

#include <ctype.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * character classes -- src/utils.c:255-295
 * ---------------------------------------------------------------------------------------------- */
static unsigned char tab_acgt[256], tab_invalid[256], tab_iupac[256];
static int tab_ready = 0;

static void
tables_init (void)
{
  if (tab_ready) return;
  memset (tab_acgt, 0, sizeof tab_acgt);
  memset (tab_invalid, 0, sizeof tab_invalid);
  memset (tab_iupac, 0, sizeof tab_iupac);
  const char *acgt = "ACGTacgt";                 /* src/utils.c:262 */
  const char *inv  = "NnXx-?Oo.";                /* src/utils.c:263 */
  for (const char *p = acgt; *p; p++) tab_acgt[(unsigned char) *p] = 1;
  for (const char *p = inv;  *p; p++) tab_invalid[(unsigned char) *p] = 1;
  /* IUPAC nucleotide sets, A=1 C=2 G=4 T=8 (the values that reproduce README.md:227-233,307-316) */
  static const struct { char c; int m; } iu[] = {
    {'A',1},{'C',2},{'G',4},{'T',8},{'M',3},{'R',5},{'W',9},{'S',6},{'Y',10},{'K',12},
    {'V',7},{'H',11},{'D',13},{'B',14} };
  for (size_t i = 0; i < sizeof iu / sizeof iu[0]; i++) {
    tab_iupac[(unsigned char) iu[i].c] = (unsigned char) iu[i].m;
    tab_iupac[(unsigned char) tolower (iu[i].c)] = (unsigned char) iu[i].m;
  }
  tab_ready = 1;
}

int orc_is_acgt (unsigned char c)   { tables_init (); return tab_acgt[c]; }
int orc_is_valid (unsigned char c)  { tables_init (); return !tab_invalid[c]; }
int orc_iupac_mask (unsigned char c){ tables_init (); return tab_iupac[c]; }

/* src/utils.c:266-277 */
static inline int pair_acgt_distinct (unsigned char a, unsigned char b) { return (tab_acgt[a] && tab_acgt[b]) ? (a != b) : 0; }
static inline int pair_acgt_valid (unsigned char a, unsigned char b)    { return tab_acgt[a] && tab_acgt[b]; }
static inline int pair_valid (unsigned char a, unsigned char b)         { return !tab_invalid[a] && !tab_invalid[b]; }

/* ------------------------------------------------------------------------------------------------
 * scoring kernels
 * ---------------------------------------------------------------------------------------------- */
void
orc_score_acgt_and_valid (const char *s1, const char *s2, size_t n, int maxdist, int *score, const size_t *idx)
{ /* src/fastaseq.c:585-596 */
  tables_init ();
  int mism = 0, both = 0;
  for (size_t j = 0; j < n && mism < maxdist; j++) {
    unsigned char a = (unsigned char) s1[idx[j]], b = (unsigned char) s2[idx[j]];
    mism += pair_acgt_distinct (a, b);
    both += pair_acgt_valid (a, b);
  }
  score[0] = mism; score[1] = both;
}

void
orc_score_matches_truncated_idx (const char *s1, const char *s2, size_t n, int maxdist, int *r, const size_t *idx)
{ /* biomcmc-lib kernel (absent).  Call sites src/nearest.c:432,491,495; stop criterion src/nearest.c:492;
     column meanings README.md:249-259; values README.md:227-233,307-316. */
  tables_init ();
  int acgt_eq = 0, text_eq = 0, partial = 0, valid = 0;
  for (size_t j = 0; j < n && (valid - acgt_eq) < maxdist; j++) {
    unsigned char a = (unsigned char) s1[idx[j]], b = (unsigned char) s2[idx[j]];
    if (!pair_valid (a, b)) continue;
    valid++;
    if (a == b) { text_eq++; if (tab_acgt[a]) acgt_eq++; }
    if (tab_iupac[a] & tab_iupac[b]) partial++;
  }
  r[0] = acgt_eq; r[1] = text_eq; r[2] = partial; r[3] = valid;
}

void
orc_dist_acgt (const char *s1, const char *s2, size_t n, int maxdist, int *score, const size_t *idx)
{ /* src/fastaseq.c:576-583 */
  tables_init ();
  int d = 0;
  for (size_t j = 0; j < n && d < maxdist; j++) d += pair_acgt_distinct ((unsigned char) s1[idx[j]], (unsigned char) s2[idx[j]]);
  *score = d;
}

void
orc_dist_text_indelcheck (const char *s1, const char *s2, size_t n, int maxdist, int *score, const size_t *idx)
{ /* src/fastaseq.c:562-574 */
  tables_init ();
  int d = 0;
  for (size_t j = 0; j < n && d < maxdist; j++) {
    unsigned char a = (unsigned char) s1[idx[j]], b = (unsigned char) s2[idx[j]];
    if (pair_valid (a, b) && a != b) d++;
  }
  *score = d;
}

int
orc_count_non_N (const char *s, size_t n)
{ /* src/fastaseq.c:642-648 */
  tables_init ();
  int c = 0;
  for (size_t i = 0; i < n; i++) c += !tab_invalid[(unsigned char) s[i]];
  return c;
}

int
orc_count_acgt (const char *s, size_t n)
{ /* src/fastaseq.c:650-656 */
  tables_init ();
  int c = 0;
  for (size_t i = 0; i < n; i++) c += tab_acgt[(unsigned char) s[i]];
  return c;
}

/* ------------------------------------------------------------------------------------------------
 * heap -- src/min_heap.c
 * ---------------------------------------------------------------------------------------------- */
int
orc_compare_score (const int *a, const int *b)
{ /* src/min_heap.c:41-47 : first non-zero of b[i]-a[i]; negative <=> a ranks ahead of b */
  for (int i = 0; i < ORC_NSCORE; i++) { int d = b[i] - a[i]; if (d) return d; }
  return 0;
}

static int
compare_items (const void *a, const void *b)
{ /* src/min_heap.c:35-39 */
  return orc_compare_score (((const orc_item *) a)->score, ((const orc_item *) b)->score);
}

orc_heap *
orc_heap_new (int heap_size)
{ /* src/min_heap.c:52-66 */
  orc_heap *h = (orc_heap *) malloc (sizeof *h);
  h->n = 0;
  h->max_incompatible = 0xffffff;
  h->heap_size = heap_size < 2 ? 2 : heap_size;
  h->seq = (orc_item *) calloc ((size_t) h->heap_size + 1, sizeof (orc_item));
  for (int i = 0; i <= h->heap_size; i++) h->seq[i].ordinal = -1;
  return h;
}

void
orc_heap_del (orc_heap *h)
{
  if (!h) return;
  if (h->seq) { for (int i = 0; i <= h->heap_size; i++) free (h->seq[i].name); free (h->seq); }
  free (h);
}

static void
sift_down (orc_heap *h, int p)
{ /* src/min_heap.c:119-133 (recursion unrolled; same comparisons in the same order) */
  for (;;) {
    int c = 2 * p, pick = p;
    for (int i = 0; i < 2; i++) if (c + i <= h->n)
      if (orc_compare_score (h->seq[pick].score, h->seq[c + i].score) < 0) pick = c + i;
    if (pick == p) return;
    orc_item t = h->seq[p]; h->seq[p] = h->seq[pick]; h->seq[pick] = t;
    p = pick;
  }
}

static void
sift_up (orc_heap *h, int i)
{ /* src/min_heap.c:135-147 */
  while (i > 1) {
    int parent = i / 2;
    if (orc_compare_score (h->seq[parent].score, h->seq[i].score) >= 0) return;
    orc_item t = h->seq[parent]; h->seq[parent] = h->seq[i]; h->seq[i] = t;
    i = parent;
  }
}

int
orc_heap_insert (orc_heap *h, const orc_item *item)
{ /* src/min_heap.c:93-117 */
  if (h->n == h->heap_size) {
    if (orc_compare_score (item->score, h->seq[1].score) >= 0) return 0;
    free (h->seq[1].name);
    h->seq[1].name = item->name ? strdup (item->name) : NULL;
    h->seq[1].ordinal = item->ordinal;
    memcpy (h->seq[1].score, item->score, sizeof item->score);
    sift_down (h, 1);
    return 1;
  }
  h->n++;
  h->seq[h->n].name = item->name ? strdup (item->name) : NULL;
  h->seq[h->n].ordinal = item->ordinal;
  memcpy (h->seq[h->n].score, item->score, sizeof item->score);
  sift_up (h, h->n);
  return 1;
}

void
orc_heap_finalise (orc_heap *h)
{ /* src/min_heap.c:149-158 : move heap [1..n] to [0..n-1], libc qsort best-first, shrink if n < size-1 */
  orc_item t = h->seq[0]; h->seq[0] = h->seq[h->n]; h->seq[h->n] = t;
  qsort (h->seq, (size_t) h->n, sizeof (orc_item), compare_items);
  if (h->n < h->heap_size - 1) h->heap_size = h->n;   /* (the reference also reallocs; irrelevant here) */
}

/* ------------------------------------------------------------------------------------------------
 * query structure -- src/fastaseq.c:698-841, src/utils.c:10-48
 * ---------------------------------------------------------------------------------------------- */
orc_query *
orc_query_new (int ntax, int nchar, const char *const *seqs, const char *const *names, int trim, int dist, int acgt)
{ /* src/fastaseq.c:698-718 */
  tables_init ();
  orc_query *q = (orc_query *) calloc (1, sizeof *q);
  q->ntax = ntax; q->nchar = nchar; q->acgt = acgt ? 1 : 0;
  q->seq  = (char **) calloc ((size_t) ntax, sizeof (char *));
  q->name = (char **) calloc ((size_t) ntax, sizeof (char *));
  for (int i = 0; i < ntax; i++) {
    q->seq[i] = (char *) malloc ((size_t) nchar + 1);
    memcpy (q->seq[i], seqs[i], (size_t) nchar);
    q->seq[i][nchar] = '\0';
    q->name[i] = strdup (names ? names[i] : "");
  }
  if (trim < 0) trim = 0;
  if (trim > nchar / 2.1) trim = (int) (nchar / 2.1);
  q->trim = (size_t) trim;
  if (dist < 0) dist = 0;
  if (dist > (nchar - 2 * trim) / 10) dist = (nchar - 2 * trim) / 10;
  q->dist = dist;
  return q;
}

void
orc_query_del (orc_query *q)
{
  if (!q) return;
  for (int i = 0; i < q->ntax; i++) { free (q->seq[i]); free (q->name[i]); }
  free (q->seq); free (q->name); free (q->consensus);
  free (q->idx_c); free (q->idx_m); free (q->idx);
  free (q);
}

static void
query_reduce (orc_query *q, const int *keep, int n_keep)
{ /* biomcmc char_vector_reduce_to_valid_strings: keep[] is increasing */
  int next = 0;
  for (int i = 0; i < q->ntax; i++) {
    if (next < n_keep && keep[next] == i) {
      q->seq[next] = q->seq[i]; q->name[next] = q->name[i]; next++;
    } else { free (q->seq[i]); free (q->name[i]); }
  }
  q->ntax = n_keep;
}

int
orc_query_keep_valid (orc_query *q, double ambiguity)
{ /* src/utils.c:10-48.  biomcmc_count_sequence_acgt is absent; from src/utils.c:23-31 result[0] is the
     ACGT fraction and result[2] the "N etc." fraction, taken here as the invalid set of src/utils.c:263.
     (UNPINNED detail: keep test queries far from both thresholds.) */
  int *keep = (int *) malloc ((size_t) q->ntax * sizeof (int)), n_keep = 0;
  for (int i = 0; i < q->ntax; i++) {
    if (q->nchar < 5) continue;
    for (int j = 0; j < q->nchar; j++) q->seq[i][j] = (char) toupper ((unsigned char) q->seq[i][j]);
    double f_acgt = (double) orc_count_acgt (q->seq[i], (size_t) q->nchar) / (double) q->nchar;
    double f_n    = 1. - (double) orc_count_non_N (q->seq[i], (size_t) q->nchar) / (double) q->nchar;
    if (f_n > ambiguity) continue;
    if (f_acgt < 1. - 1.1 * ambiguity) continue;
    keep[n_keep++] = i;
  }
  query_reduce (q, keep, n_keep);
  free (keep);
  return n_keep;
}

void
orc_query_create_indices (orc_query *q)
{ /* src/fastaseq.c:732-777 */
  tables_init ();
  int lo = (int) q->trim, hi = q->nchar - (int) q->trim;
  if (!q->consensus) q->consensus = (char *) malloc ((size_t) q->nchar);
  unsigned char *miss = (unsigned char *) calloc ((size_t) q->nchar, 1);
  memset (q->consensus, 'N', (size_t) q->nchar);

  for (int i = lo; i < hi; i++) for (int j = 0; j < q->ntax && q->consensus[i] != '#'; j++) {
    unsigned char s2 = (unsigned char) q->seq[j][i];
    int usable = q->acgt ? tab_acgt[s2] : !tab_invalid[s2];   /* :746 vs :753 */
    if (!usable) { miss[i] = 1; continue; }
    if (q->consensus[i] == 'N') q->consensus[i] = (char) s2;
    else if ((unsigned char) q->consensus[i] != s2) q->consensus[i] = '#';
  }

  size_t cap = (size_t) (hi > lo ? hi - lo : 0) + 1;
  q->idx_c = (size_t *) realloc (q->idx_c, cap * sizeof (size_t));
  q->idx_m = (size_t *) realloc (q->idx_m, cap * sizeof (size_t));
  q->idx   = (size_t *) realloc (q->idx,   cap * sizeof (size_t));
  q->n_idx_c = q->n_idx_m = q->n_idx = 0;
  for (int i = lo; i < hi; i++) if (q->consensus[i] != 'N') {
    if (q->consensus[i] == '#') q->idx[q->n_idx++] = (size_t) i;
    else if (miss[i])            q->idx_m[q->n_idx_m++] = (size_t) i;
    else                         q->idx_c[q->n_idx_c++] = (size_t) i;
  }
  free (miss);
}

typedef struct { int value, pos; } sort_pair;
static int
cmp_sort_pair (const void *a, const void *b)
{ /* increasing value; ties keep input order (biomcmc's new_empfreq_sort_increasing is absent: tie order UNPINNED) */
  const sort_pair *x = (const sort_pair *) a, *y = (const sort_pair *) b;
  if (x->value != y->value) return x->value < y->value ? -1 : 1;
  return x->pos - y->pos;
}

void
orc_query_reorder (orc_query *q)
{ /* src/fastaseq.c:779-795 */
  sort_pair *sp = (sort_pair *) malloc ((size_t) q->ntax * sizeof *sp);
  size_t span = (size_t) q->nchar - 2 * q->trim;
  for (int i = 0; i < q->ntax; i++) {
    sp[i].pos = i;
    sp[i].value = q->acgt ? orc_count_acgt (q->seq[i] + q->trim, span) : orc_count_non_N (q->seq[i] + q->trim, span);
  }
  qsort (sp, (size_t) q->ntax, sizeof *sp, cmp_sort_pair);
  char **s2 = (char **) malloc ((size_t) q->ntax * sizeof (char *)), **n2 = (char **) malloc ((size_t) q->ntax * sizeof (char *));
  for (int i = 0; i < q->ntax; i++) { s2[i] = q->seq[sp[i].pos]; n2[i] = q->name[sp[i].pos]; }
  memcpy (q->seq, s2, (size_t) q->ntax * sizeof (char *));
  memcpy (q->name, n2, (size_t) q->ntax * sizeof (char *));
  free (s2); free (n2); free (sp);
}

static int
resolved_cmp (const char *s1, const char *s2, size_t n, const size_t *idx, int acgt)
{ /* src/fastaseq.c:598-640 : -1 left more resolved, +1 right, 0 same pattern, 0xff incomparable */
  int score = 0;
  for (size_t j = 0; j < n && score < 0xff; j++) {
    unsigned char a = (unsigned char) s1[idx[j]], b = (unsigned char) s2[idx[j]];
    int u = acgt ? tab_acgt[a] : !tab_invalid[a], v = acgt ? tab_acgt[b] : !tab_invalid[b];
    if (u == v) continue;
    if (u > v) { if (score > 0) return 0xff; score = -1; }
    else       { if (score < 0) return 0xff; score = 1; }
  }
  return score;
}

void
orc_query_exclude_redundant (orc_query *q, int keep_more_resolved)
{ /* src/fastaseq.c:797-841 */
  int *valid = (int *) malloc ((size_t) q->ntax * sizeof (int)), n_valid = 0, dist = 0;
  for (int i = 0; i < q->ntax; i++) valid[i] = 1;
  for (int i = 0; i < q->ntax - 1; i++) for (int j = i + 1; j < q->ntax; j++) if (valid[i] && valid[j]) {
    if (q->acgt) orc_dist_acgt (q->seq[i], q->seq[j], (size_t) q->n_idx, 1, &dist, q->idx);
    else         orc_dist_text_indelcheck (q->seq[i], q->seq[j], (size_t) q->n_idx, 1, &dist, q->idx);
    if (dist) continue;
    int red1 = resolved_cmp (q->seq[i], q->seq[j], (size_t) q->n_idx, q->idx, q->acgt);
    if (red1 > 1) continue;
    int red2 = resolved_cmp (q->seq[i], q->seq[j], (size_t) q->n_idx_m, q->idx_m, q->acgt);
    if (red2 > 1) continue;
    if (!red1 && !red2) valid[j] = 0;
    red1 += red2;
    if (!red1) continue;
    if (keep_more_resolved) { if (red1 > 0) valid[i] = 0; else valid[j] = 0; }
    else                    { if (red1 > 0) valid[j] = 0; else valid[i] = 0; }
  }
  for (int i = 0; i < q->ntax; i++) if (valid[i]) valid[n_valid++] = i;
  query_reduce (q, valid, n_valid);
  free (valid);
}

orc_query *
orc_query_prepare (int ntax, int nchar, const char *const *seqs, const char *const *names,
                   int trim, int dist, int acgt, double ambig_q, int keep_resolved, int is_ball)
{ /* src/nearest.c:175-176,203-224 ; src/ball.c:153-154,174-194 */
  if (ambig_q < 0.001) ambig_q = 0.001;
  if (ambig_q > 1.) ambig_q = 1.;
  orc_query *q = orc_query_new (ntax, nchar, seqs, names, trim, dist, acgt);
  orc_query_keep_valid (q, ambig_q);
  if (q->ntax < 1) return q;
  orc_query_create_indices (q);
  orc_query_reorder (q);
  if (is_ball) {                      /* ball always prunes (src/ball.c:190) */
    orc_query_exclude_redundant (q, keep_resolved);
    orc_query_create_indices (q);
  } else if (keep_resolved) {         /* src/nearest.c:219-224 */
    orc_query_exclude_redundant (q, keep_resolved);
    orc_query_create_indices (q);
  }
  return q;
}

/* ------------------------------------------------------------------------------------------------
 * nearest-neighbour search -- src/nearest.c
 * ---------------------------------------------------------------------------------------------- */
struct orc_search {
  orc_query *q;
  int pool, n_query, acgt, exclude_self, non_n_ref;
  int max_incompatible;            /* cq->max_incompatible */
  int *res, *non_n;                /* [4*pool], [pool] */
  unsigned char *is_best;          /* [pool][n_query] */
  char **seq, **name;              /* [pool], owned copies */
  int64_t *ordinal;                /* [pool] */
  int fill;                        /* occupied slots in the current batch */
  orc_heap **heap;
  int64_t n_seen, n_lowqual, n_samename, n_saved, cap_saved, *saved;
  int finished;
  int snap_override;               /* >= 0: use this as cq->max_incompatible for the next batch (ring-protocol tests) */
  int last_snapshot;
};

void orc_set_threads (int n)
{
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads (n);
#else
  (void) n;
#endif
}
int orc_max_threads (void)
{
#ifdef _OPENMP
  return omp_get_max_threads ();
#else
  return 1;
#endif
}

orc_search *
orc_search_new (orc_query *q, int pool, int nbest, double ambig_r, int exclude_self)
{ /* src/nearest.c:177-179,233,237,367-390 */
  tables_init ();
  if (ambig_r < 0.001) ambig_r = 0.001;
  if (ambig_r > 1.) ambig_r = 1.;
  if (nbest < 1) nbest = 1;
  if (pool < 1) pool = 1;
  orc_search *s = (orc_search *) calloc (1, sizeof *s);
  s->q = q; s->pool = pool; s->n_query = q->ntax; s->acgt = q->acgt; s->exclude_self = exclude_self;
  s->non_n_ref = (int) (q->nchar * (1. - ambig_r));
  s->max_incompatible = q->nchar;
  s->res     = (int *) calloc ((size_t) 4 * pool, sizeof (int));
  s->non_n   = (int *) calloc ((size_t) pool, sizeof (int));
  s->is_best = (unsigned char *) calloc ((size_t) pool * (size_t) s->n_query, 1);
  s->seq     = (char **) calloc ((size_t) pool, sizeof (char *));
  s->name    = (char **) calloc ((size_t) pool, sizeof (char *));
  s->ordinal = (int64_t *) calloc ((size_t) pool, sizeof (int64_t));
  s->heap    = (orc_heap **) calloc ((size_t) s->n_query, sizeof (orc_heap *));
  for (int i = 0; i < s->n_query; i++) { s->heap[i] = orc_heap_new (nbest); s->heap[i]->max_incompatible = s->max_incompatible; }
  s->cap_saved = 1024; s->saved = (int64_t *) malloc ((size_t) s->cap_saved * sizeof (int64_t));
  s->snap_override = -1; s->last_snapshot = s->max_incompatible;
  return s;
}

void
orc_search_del (orc_search *s)
{
  if (!s) return;
  for (int c = 0; c < s->pool; c++) { free (s->seq[c]); free (s->name[c]); }
  for (int i = 0; i < s->n_query; i++) orc_heap_del (s->heap[i]);
  free (s->res); free (s->non_n); free (s->is_best); free (s->seq); free (s->name); free (s->ordinal);
  free (s->heap); free (s->saved); free (s);
}

static void
consensus_score (orc_search *s, int c)
{ /* src/nearest.c:428-433 */
  const orc_query *q = s->q;
  if (s->acgt) orc_score_acgt_and_valid (s->seq[c], q->consensus, (size_t) q->n_idx_c, s->max_incompatible, s->res + 4 * c, q->idx_c);
  else  orc_score_matches_truncated_idx (s->seq[c], q->consensus, (size_t) q->n_idx_c, s->max_incompatible, s->res + 4 * c, q->idx_c);
}

static void
update_heap_acgt (orc_search *s, int iq, int c)
{ /* src/nearest.c:442-477 */
  const orc_query *q = s->q;
  orc_heap *h = s->heap[iq];
  const int *res = s->res + 4 * c;
  int r[4], cons_matches = res[1] - res[0], tol = res[0];
  if (tol >= h->max_incompatible) return;
  tol = h->max_incompatible - tol;
  orc_score_acgt_and_valid (s->seq[c], q->seq[iq], (size_t) q->n_idx_m, tol, r, q->idx_m);
  if (r[0] >= tol) return;
  tol -= r[0];
  r[0] += res[0]; r[1] += res[1];
  orc_score_acgt_and_valid (s->seq[c], q->seq[iq], (size_t) q->n_idx, tol, r + 2, q->idx);
  if (r[2] >= tol) return;
  orc_item it; memset (&it, 0, sizeof it);
  it.score[0] = r[1] + r[3] - r[0] - r[2];
  it.score[1] = r[1] + r[3];
  it.score[2] = it.score[0] - cons_matches;
  it.score[3] = s->non_n[c];
  it.score[4] = r[0];
  it.score[5] = r[2];
  it.name = s->name[c]; it.ordinal = s->ordinal[c];
  if (orc_heap_insert (h, &it)) {
    s->is_best[(size_t) s->n_query * c + iq] = 1;
    if (h->n == h->heap_size) h->max_incompatible = h->seq[1].score[1] - h->seq[1].score[0] + 1;
  }
}

static void
update_heap_full (orc_search *s, int iq, int c)
{ /* src/nearest.c:479-510 */
  const orc_query *q = s->q;
  orc_heap *h = s->heap[iq];
  const int *res = s->res + 4 * c;
  int r[8], tol = res[3] - res[0];
  if (tol >= h->max_incompatible) return;
  tol = h->max_incompatible - tol;
  orc_score_matches_truncated_idx (s->seq[c], q->seq[iq], (size_t) q->n_idx_m, tol, r, q->idx_m);
  if (r[3] - r[0] >= tol) return;
  tol = tol - r[3] + r[0];
  orc_score_matches_truncated_idx (s->seq[c], q->seq[iq], (size_t) q->n_idx, tol, r + 4, q->idx);
  if (r[7] - r[4] >= tol) return;
  orc_item it; memset (&it, 0, sizeof it);
  for (int i = 0; i < 4; i++) it.score[i] = r[i] + r[i + 4] + res[i];
  it.score[4] = r[0] + r[4];
  it.score[5] = s->non_n[c];
  it.name = s->name[c]; it.ordinal = s->ordinal[c];
  if (orc_heap_insert (h, &it)) {
    s->is_best[(size_t) s->n_query * c + iq] = 1;
    if (h->n == h->heap_size) h->max_incompatible = h->seq[1].score[3] - h->seq[1].score[0] + 1;
  }
}

static void
process_batch (orc_search *s)
{ /* src/nearest.c:288-319 */
  int c, j;
  memset (s->is_best, 0, (size_t) s->pool * (size_t) s->n_query);
  s->max_incompatible = s->heap[0]->max_incompatible;
  for (j = 1; j < s->n_query; j++) if (s->max_incompatible < s->heap[j]->max_incompatible) s->max_incompatible = s->heap[j]->max_incompatible;
  if (s->snap_override >= 0) { s->max_incompatible = s->snap_override; s->snap_override = -1; }
  s->last_snapshot = s->max_incompatible;

#pragma omp parallel for
  for (c = 0; c < s->fill; c++) consensus_score (s, c);

#pragma omp parallel for private(c)
  for (j = 0; j < s->n_query; j++)
    for (c = 0; c < s->fill; c++) { if (s->acgt) update_heap_acgt (s, j, c); else update_heap_full (s, j, c); }

  for (c = 0; c < s->fill; c++) {
    int any = 0;
    for (j = 0; j < s->n_query; j++) any |= s->is_best[(size_t) s->n_query * c + j];
    if (any) {
      if (s->n_saved == s->cap_saved) { s->cap_saved *= 2; s->saved = (int64_t *) realloc (s->saved, (size_t) s->cap_saved * sizeof (int64_t)); }
      s->saved[s->n_saved++] = s->ordinal[c];
    }
  }
  for (c = 0; c < s->fill; c++) { free (s->seq[c]); free (s->name[c]); s->seq[c] = s->name[c] = NULL; }
  s->fill = 0;
}

int
orc_search_feed (orc_search *s, int n, const char *const *seqs, const char *const *names, const int *lengths)
{ /* src/nearest.c:251-286 (slot filling; a full pool triggers the batch) */
  const orc_query *q = s->q;
  for (int i = 0; i < n; i++) {
    int64_t ord = s->n_seen++;
    int len = lengths ? lengths[i] : q->nchar;
    if (s->exclude_self) {
      int hit = 0;
      for (int j = 0; j < q->ntax && !hit; j++) hit = !strcmp (q->name[j], names[i]);
      if (hit) { s->n_samename++; continue; }
    }
    int nn = orc_count_non_N (seqs[i], (size_t) len);
    if (nn < s->non_n_ref) { s->n_lowqual++; continue; }
    if (len != q->nchar) return -1;
    int c = s->fill++;
    s->non_n[c] = nn;
    s->seq[c] = (char *) malloc ((size_t) len + 1); memcpy (s->seq[c], seqs[i], (size_t) len); s->seq[c][len] = '\0';
    s->name[c] = strdup (names ? names[i] : "");
    s->ordinal[c] = ord;
    if (s->fill == s->pool) process_batch (s);
  }
  return 0;
}

void
orc_search_finish (orc_search *s)
{ /* end of file: the partial batch is processed (src/nearest.c:282-285), then src/nearest.c:513-547 */
  if (s->finished) return;
  process_batch (s);   /* also reproduces the empty trailing batch (only refreshes cq->max_incompatible) */
  for (int i = 0; i < s->n_query; i++) orc_heap_finalise (s->heap[i]);
  s->finished = 1;
}

/* ---- helpers for the multi-rank ring protocol tests (not reference functions) ----
 * A stripe (= one pool of the reference) is cut into slices held by different ranks; each rank processes its slice as a
 * batch of its own but with the stripe's snapshot of cq->max_incompatible, and hands the heap state to the next rank. */
int
orc_search_process_slice (orc_search *s, int n, const char *const *seqs, const char *const *names, const int64_t *ordinals, int snapshot)
{
  if (n > s->pool || s->fill) return -1;
  for (int i = 0; i < n; i++) {
    int c = s->fill++;
    s->non_n[c] = orc_count_non_N (seqs[i], (size_t) s->q->nchar);
    s->seq[c] = (char *) malloc ((size_t) s->q->nchar + 1); memcpy (s->seq[c], seqs[i], (size_t) s->q->nchar); s->seq[c][s->q->nchar] = '\0';
    s->name[c] = strdup (names ? names[i] : "");
    s->ordinal[c] = ordinals[i];
  }
  s->snap_override = snapshot;
  process_batch (s);
  return 0;
}

/* the same for queries [q0,q1) only (the per-query machines are independent): used by the query-group pipelined ring */
int
orc_search_process_slice_range (orc_search *s, int n, const char *const *seqs, const char *const *names, const int64_t *ordinals, int snapshot, int q0, int q1)
{
  if (n > s->pool || s->fill || snapshot < 0) return -1;
  for (int i = 0; i < n; i++) {
    s->non_n[i] = orc_count_non_N (seqs[i], (size_t) s->q->nchar);
    s->seq[i] = (char *) seqs[i];          /* borrowed for the duration of the call */
    s->name[i] = (char *) (names ? names[i] : "");
    s->ordinal[i] = ordinals[i];
  }
  s->max_incompatible = s->last_snapshot = snapshot;
  for (int c = 0; c < n; c++) consensus_score (s, c);
  for (int j = q0; j < q1; j++) for (int c = 0; c < n; c++) { if (s->acgt) update_heap_acgt (s, j, c); else update_heap_full (s, j, c); }
  for (int i = 0; i < n; i++) { s->seq[i] = NULL; s->name[i] = NULL; }
  return 0;
}

int
orc_search_max_T (const orc_search *s)
{ /* what the next batch's snapshot would be (src/nearest.c:290-291) */
  int m = s->heap[0]->max_incompatible;
  for (int j = 1; j < s->n_query; j++) if (m < s->heap[j]->max_incompatible) m = s->heap[j]->max_incompatible;
  return m;
}

int orc_search_last_snapshot (const orc_search *s) { return s->last_snapshot; }

size_t
orc_search_state_ints (const orc_search *s)
{
  size_t per_q = 2 + (size_t) (s->heap[0]->heap_size + 1) * 8;
  return 1 + per_q * (size_t) s->n_query;
}

void
orc_search_get_state_range (const orc_search *s, int *blob, int q0, int q1)
{
  size_t per_q = 2 + (size_t) (s->heap[0]->heap_size + 1) * 8;
  blob[0] = s->last_snapshot;
  for (int q = q0; q < q1; q++) {
    int *b = blob + 1 + per_q * (size_t) (q - q0);
    const orc_heap *h = s->heap[q];
    b[0] = h->n; b[1] = h->max_incompatible;
    for (int e = 0; e <= h->heap_size; e++) {
      for (int k = 0; k < ORC_NSCORE; k++) b[2 + e * 8 + k] = h->seq[e].score[k];
      b[2 + e * 8 + 6] = (int) (uint32_t) (h->seq[e].ordinal & 0xffffffffLL);
      b[2 + e * 8 + 7] = (int) (h->seq[e].ordinal >> 32);
    }
  }
}

void orc_search_get_state (const orc_search *s, int *blob) { orc_search_get_state_range (s, blob, 0, s->n_query); }

void
orc_search_set_state_range (orc_search *s, const int *blob, const char *name_prefix, int q0, int q1)
{
  size_t per_q = 2 + (size_t) (s->heap[0]->heap_size + 1) * 8;
  char buf[64];
  s->last_snapshot = blob[0];
  for (int q = q0; q < q1; q++) {
    const int *b = blob + 1 + per_q * (size_t) (q - q0);
    orc_heap *h = s->heap[q];
    h->n = b[0]; h->max_incompatible = b[1];
    for (int e = 0; e <= h->heap_size; e++) {
      for (int k = 0; k < ORC_NSCORE; k++) h->seq[e].score[k] = b[2 + e * 8 + k];
      h->seq[e].ordinal = (int64_t) (((uint64_t) (uint32_t) b[2 + e * 8 + 7] << 32) | (uint32_t) b[2 + e * 8 + 6]);
      free (h->seq[e].name); h->seq[e].name = NULL;
      if (e >= 1 && e <= h->n) { snprintf (buf, sizeof buf, "%s%lld", name_prefix ? name_prefix : "", (long long) h->seq[e].ordinal); h->seq[e].name = strdup (buf); }
    }
  }
}

void orc_search_set_state (orc_search *s, const int *blob, const char *name_prefix) { orc_search_set_state_range (s, blob, name_prefix, 0, s->n_query); }

void
orc_search_end_of_file (orc_search *s)
{ /* one -r file exhausted: its partial batch is processed before the next file starts (src/nearest.c:245-249,282-285) */
  if (!s->finished) process_batch (s);
}

int orc_search_nrows (const orc_search *s, int iq)  { return s->heap[iq]->heap_size; }
int orc_search_heap_n (const orc_search *s, int iq) { return s->heap[iq]->n; }
const orc_item *orc_search_row (const orc_search *s, int iq, int rank0) { return &s->heap[iq]->seq[rank0]; }
int64_t orc_search_n_saved (const orc_search *s) { return s->n_saved; }
const int64_t *orc_search_saved_ordinals (const orc_search *s) { return s->saved; }
int64_t orc_search_n_seen (const orc_search *s) { return s->n_seen; }
int64_t orc_search_n_lowqual (const orc_search *s) { return s->n_lowqual; }
int64_t orc_search_n_samename (const orc_search *s) { return s->n_samename; }
int orc_search_final_T (const orc_search *s, int iq) { return s->heap[iq]->max_incompatible; }

void
orc_allpairs_scores (const orc_query *q, int n_ref, const char *const *refs, int *out)
{
  tables_init ();
#pragma omp parallel for
  for (int r = 0; r < n_ref; r++) {
    int res[4] = {0, 0, 0, 0}, a[4], b[4];
    int nn = orc_count_non_N (refs[r], (size_t) q->nchar);
    if (q->acgt) orc_score_acgt_and_valid (refs[r], q->consensus, (size_t) q->n_idx_c, INT_MAX, res, q->idx_c);
    else  orc_score_matches_truncated_idx (refs[r], q->consensus, (size_t) q->n_idx_c, INT_MAX, res, q->idx_c);
    for (int iq = 0; iq < q->ntax; iq++) {
      int *S = out + ((size_t) r * q->ntax + iq) * ORC_NSCORE;
      if (q->acgt) { /* src/nearest.c:453-469 with nothing truncated */
        orc_score_acgt_and_valid (refs[r], q->seq[iq], (size_t) q->n_idx_m, INT_MAX, a, q->idx_m);
        orc_score_acgt_and_valid (refs[r], q->seq[iq], (size_t) q->n_idx,   INT_MAX, b, q->idx);
        int r0 = a[0] + res[0], r1 = a[1] + res[1];
        S[0] = r1 + b[1] - r0 - b[0]; S[1] = r1 + b[1]; S[2] = S[0] - (res[1] - res[0]);
        S[3] = nn; S[4] = r0; S[5] = b[0];
      } else {       /* src/nearest.c:491-501 with nothing truncated */
        orc_score_matches_truncated_idx (refs[r], q->seq[iq], (size_t) q->n_idx_m, INT_MAX, a, q->idx_m);
        orc_score_matches_truncated_idx (refs[r], q->seq[iq], (size_t) q->n_idx,   INT_MAX, b, q->idx);
        for (int i = 0; i < 4; i++) S[i] = a[i] + b[i] + res[i];
        S[4] = a[0] + b[0]; S[5] = nn;
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * radius search -- src/fastaseq.c:660-696 driven as in src/ball.c:201-259
 * ---------------------------------------------------------------------------------------------- */
static void
ball_one (const orc_query *q, const char *seq, int *min_dist, int radius)
{ /* src/fastaseq.c:660-696 */
  void (*dist) (const char *, const char *, size_t, int, int *, const size_t *) = q->acgt ? orc_dist_acgt : orc_dist_text_indelcheck;
  int c_dist;
  dist (seq, q->consensus, (size_t) q->n_idx_c, radius, min_dist, q->idx_c);
  if (*min_dist >= radius) return;
  c_dist = *min_dist;
  dist (seq, q->consensus, (size_t) q->n_idx_m, radius, min_dist, q->idx_m);
  *min_dist += c_dist;
  if (*min_dist >= radius) return;
  c_dist = *min_dist;
  for (int i = 0; i < q->ntax && (*min_dist + c_dist) >= radius; i++)
    dist (seq, q->seq[i], (size_t) q->n_idx, radius - c_dist, min_dist, q->idx);
  *min_dist += c_dist;
}

void
orc_ball (const orc_query *q, double ambig_r, int n_ref, const char *const *refs, int *mindist, unsigned char *keep)
{ /* src/ball.c:155-156,201,216-259 */
  tables_init ();
  if (ambig_r < 0.001) ambig_r = 0.001;
  if (ambig_r > 1.) ambig_r = 1.;
  int non_n_ref = (int) (q->nchar * ambig_r);
#pragma omp parallel for
  for (int r = 0; r < n_ref; r++) {
    mindist[r] = 0xffffff; keep[r] = 0;
    if (orc_count_non_N (refs[r], (size_t) q->nchar) < non_n_ref) continue;
    ball_one (q, refs[r], &mindist[r], q->dist + 1);
    keep[r] = (unsigned char) (mindist[r] <= q->dist);
  }
}

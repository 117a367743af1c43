/*
 * wfa_oracle.c -- see wfa_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED for the aligner (third-party WFA
 * v1, absent from /root/reference/submodules/WFA).  Plain scalar C written to be read next to the paper; wavefront memory
 * comes from slabs the aligner keeps across alignments, one aligner per thread, as in the reference (src/align.c:304-310).
 */
#include "wfa_oracle.h"
#include "uvaia_oracle.h"
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OFFSET_NULL (-10)           /* offset of a diagonal a wavefront does not hold */
#define MAXI(a,b) ((a) > (b) ? (a) : (b))
#define MINI(a,b) ((a) < (b) ? (a) : (b))

typedef struct {
  int lo, hi;                       /* effective limits (after reduction), inclusive */
  int lo_base, hi_base;             /* allocated limits */
  int null;
  int *mem;                         /* offsets of lo_base..hi_base; offset = h (text position); v = h - k */
} wavefront;

typedef struct slab { struct slab *next; size_t cap, used; } slab;

struct orc_wfa {
  slab *slabs, *cur;                /* wavefront memory: slabs kept across alignments (the reference's aligners own an mm_allocator: src/align.c:306-308) */
  orc_wfa_penalties pen;
  int min_wavefront_length, max_distance_threshold;   /* min_wavefront_length <= 0: no reduction */
  wavefront **m, **i, **d;          /* per score; NULL = no wavefront at that score */
  int n_alloc, n_used;
  wavefront null_wf;
  char *ops; int ops_cap, ops_begin, ops_end;          /* CIGAR filled from the back by the backtrace */
  int64_t cells; int max_width;
};

static void *
slab_alloc (orc_wfa *w, size_t bytes)
{
  bytes = (bytes + 15) & ~(size_t) 15;
  while (w->cur && w->cur->used + bytes > w->cur->cap) { w->cur = w->cur->next; if (w->cur) w->cur->used = 0; }
  if (!w->cur) {
    size_t cap = bytes > ((size_t) 8 << 20) ? bytes : ((size_t) 8 << 20);         /* BUFFER_SIZE_8M, src/align.c:307 */
    slab *sl = (slab *) malloc (sizeof (slab) + cap);
    sl->next = NULL; sl->cap = cap; sl->used = 0;
    if (!w->slabs) w->slabs = sl; else { slab *t = w->slabs; while (t->next) t = t->next; t->next = sl; }
    w->cur = sl;
  }
  void *p = (char *) (w->cur + 1) + w->cur->used;
  w->cur->used += bytes;
  return p;
}

static int wf_get (const wavefront *w, int k) { return (w->lo <= k && k <= w->hi) ? w->mem[k - w->lo_base] : OFFSET_NULL; }

orc_wfa *
orc_wfa_new (orc_wfa_penalties pen, int min_wavefront_length, int max_distance_threshold)
{
  orc_wfa *w = (orc_wfa *) calloc (1, sizeof (orc_wfa));
  w->pen = pen; w->min_wavefront_length = min_wavefront_length; w->max_distance_threshold = max_distance_threshold;
  w->null_wf.lo = 1; w->null_wf.hi = -1; w->null_wf.lo_base = 1; w->null_wf.hi_base = -1; w->null_wf.null = 1; w->null_wf.mem = NULL;
  return w;
}

static void
clear_wavefronts (orc_wfa *w)
{ /* affine_wavefronts_clear (src/align.c:360): the wavefronts go, their memory stays with the aligner */
  for (int s = 0; s < w->n_used; s++) w->m[s] = w->i[s] = w->d[s] = NULL;
  w->cur = w->slabs; if (w->cur) w->cur->used = 0;
  w->n_used = 0; w->cells = 0; w->max_width = 0;
}

void
orc_wfa_del (orc_wfa *w)
{
  if (!w) return;
  for (slab *sl = w->slabs; sl; ) { slab *nx = sl->next; free (sl); sl = nx; }
  free (w->m); free (w->i); free (w->d); free (w->ops); free (w);
}

static void
reserve_scores (orc_wfa *w, int score)
{
  if (score < w->n_alloc) { if (score >= w->n_used) w->n_used = score + 1; return; }
  int n = w->n_alloc ? w->n_alloc : 1024;
  while (n <= score) n *= 2;
  w->m = (wavefront **) realloc (w->m, (size_t) n * sizeof (wavefront *));
  w->i = (wavefront **) realloc (w->i, (size_t) n * sizeof (wavefront *));
  w->d = (wavefront **) realloc (w->d, (size_t) n * sizeof (wavefront *));
  for (int s = w->n_alloc; s < n; s++) w->m[s] = w->i[s] = w->d[s] = NULL;
  w->n_alloc = n; w->n_used = score + 1;
}

static wavefront *
wf_new (orc_wfa *w, int lo, int hi)
{
  wavefront *f = (wavefront *) slab_alloc (w, sizeof (wavefront) + (size_t) (hi - lo + 1) * sizeof (int));
  f->lo = f->lo_base = lo; f->hi = f->hi_base = hi; f->null = 0;
  f->mem = (int *) (f + 1);
  return f;
}

static const wavefront *
source (const orc_wfa *w, wavefront **v, int score) { return (score < 0 || !v[score]) ? &w->null_wf : v[score]; }

/* ---- exact extension of the M-wavefront along its diagonals (paper algorithm 2) followed by the reduction ---- */
static int
distance_to_end (int plen, int tlen, int offset, int k)
{
  int left_v = plen - (offset - k), left_h = tlen - offset;
  return MAXI (left_v, left_h);
}

static void
reduce_wavefronts (orc_wfa *w, int plen, int tlen, int score)
{ /* paper section 2.4 ("adaptive"): diagonals further than the threshold behind the most advanced one are dropped from
     both ends, never across the diagonal of the end cell; I and D take the limits of M */
  wavefront *m = w->m[score];
  if (!m) return;
  if (m->hi - m->lo + 1 < w->min_wavefront_length) return;
  const int alignment_k = tlen - plen;
  int min_distance = MAXI (plen, tlen);
  for (int k = m->lo; k <= m->hi; k++) { int dist = distance_to_end (plen, tlen, m->mem[k - m->lo_base], k); min_distance = MINI (min_distance, dist); }
  const int top_limit = MINI (alignment_k - 1, m->hi);
  for (int k = m->lo; k < top_limit; k++) {
    if (distance_to_end (plen, tlen, m->mem[k - m->lo_base], k) - min_distance <= w->max_distance_threshold) break;
    m->lo++;
  }
  const int bottom_limit = MAXI (alignment_k + 1, m->lo);
  for (int k = m->hi; k > bottom_limit; k--) {
    if (distance_to_end (plen, tlen, m->mem[k - m->lo_base], k) - min_distance <= w->max_distance_threshold) break;
    m->hi--;
  }
  if (m->lo > m->hi) m->null = 1;
  wavefront *o[2] = { w->i[score], w->d[score] };
  for (int j = 0; j < 2; j++) if (o[j]) {
    if (m->lo > o[j]->lo) o[j]->lo = m->lo;
    if (m->hi < o[j]->hi) o[j]->hi = m->hi;
    if (o[j]->lo > o[j]->hi) o[j]->null = 1;
  }
}

static void
extend_wavefront (orc_wfa *w, const char *pattern, int plen, const char *text, int tlen, int score)
{
  wavefront *m = w->m[score];
  if (!m) return;
  for (int k = m->lo; k <= m->hi; k++) {
    int offset = m->mem[k - m->lo_base];
    unsigned h = (unsigned) offset, v = (unsigned) (offset - k);       /* unsigned: a negative position is out of range too */
    if (h >= (unsigned) tlen || v >= (unsigned) plen) continue;
    while (v < (unsigned) plen && h < (unsigned) tlen && pattern[v] == text[h]) { v++; h++; offset++; }
    m->mem[k - m->lo_base] = offset;
  }
  if (w->min_wavefront_length > 0) reduce_wavefronts (w, plen, tlen, score);
}

static int
end_reached (const orc_wfa *w, int plen, int tlen, int score)
{
  const wavefront *m = w->m[score];
  const int alignment_k = tlen - plen;
  return m && m->lo <= alignment_k && alignment_k <= m->hi && m->mem[alignment_k - m->lo_base] >= tlen;
}

/* ---- next wavefronts from the earlier ones (paper equation 3 / algorithm 3) ---- */
static void
compute_wavefront (orc_wfa *w, int score)
{
  const orc_wfa_penalties *p = &w->pen;
  const wavefront *m_sub = source (w, w->m, score - p->mismatch);
  const wavefront *m_gap = source (w, w->m, score - p->gap_opening - p->gap_extension);
  const wavefront *i_ext = source (w, w->i, score - p->gap_extension);
  const wavefront *d_ext = source (w, w->d, score - p->gap_extension);
  if (m_sub->null && m_gap->null && i_ext->null && d_ext->null) return;
  int lo = m_sub->lo, hi = m_sub->hi;
  lo = MINI (lo, m_gap->lo); lo = MINI (lo, i_ext->lo); lo = MINI (lo, d_ext->lo); lo--;
  hi = MAXI (hi, m_gap->hi); hi = MAXI (hi, i_ext->hi); hi = MAXI (hi, d_ext->hi); hi++;
  wavefront *out_m = w->m[score] = wf_new (w, lo, hi);
  wavefront *out_i = (!m_gap->null || !i_ext->null) ? (w->i[score] = wf_new (w, lo, hi)) : NULL;
  wavefront *out_d = (!m_gap->null || !d_ext->null) ? (w->d[score] = wf_new (w, lo, hi)) : NULL;
  w->cells += hi - lo + 1;
  if (hi - lo + 1 > w->max_width) w->max_width = hi - lo + 1;
  for (int k = lo; k <= hi; k++) {
    int sub = wf_get (m_sub, k); if (m_sub->lo <= k && k <= m_sub->hi) sub++;      /* the +1 belongs to the fetched value only */
    int best = sub;
    if (out_i) {
      int ins = MAXI (wf_get (m_gap, k - 1), wf_get (i_ext, k - 1)) + 1;
      out_i->mem[k - lo] = ins; best = MAXI (best, ins);
    }
    if (out_d) {
      int del = MAXI (wf_get (m_gap, k + 1), wf_get (d_ext, k + 1));
      out_d->mem[k - lo] = del; best = MAXI (best, del);
    }
    out_m->mem[k - lo] = best;
  }
}

/* ---- backtrace (paper section 2.3, last paragraph) ---- */
static void
push_op (orc_wfa *w, char op) { w->ops[--w->ops_begin] = op; }

static int
trace (const orc_wfa *w, wavefront **v, int score, int k, int add)
{ /* a diagonal inside the wavefront's limits gives its offset (+ add), anything else the null offset */
  if (score < 0 || score >= w->n_used || !v[score]) return OFFSET_NULL;
  const wavefront *f = v[score];
  return (f->lo <= k && k <= f->hi) ? f->mem[k - f->lo_base] + add : OFFSET_NULL;
}

static int
backtrace (orc_wfa *w, const char *pattern, int plen, const char *text, int tlen, int alignment_score)
{
  const orc_wfa_penalties *p = &w->pen;
  if (w->ops_cap < plen + tlen + 2) { w->ops_cap = plen + tlen + 2; w->ops = (char *) realloc (w->ops, (size_t) w->ops_cap); }
  w->ops_begin = w->ops_end = w->ops_cap;
  int score = alignment_score, k = tlen - plen;
  int offset = wf_get (w->m[score], k);
  enum { BT_M, BT_I, BT_D } type = BT_M;
  int v = offset - k, h = offset;
  while (v > 0 && h > 0 && score > 0) {
    const int gap_open_score = score - p->gap_opening - p->gap_extension, gap_extend_score = score - p->gap_extension, mismatch_score = score - p->mismatch;
    const int del_ext  = (type == BT_I) ? OFFSET_NULL : trace (w, w->d, gap_extend_score, k + 1, 0);
    const int del_open = (type == BT_I) ? OFFSET_NULL : trace (w, w->m, gap_open_score, k + 1, 0);
    const int ins_ext  = (type == BT_D) ? OFFSET_NULL : trace (w, w->i, gap_extend_score, k - 1, 1);
    const int ins_open = (type == BT_D) ? OFFSET_NULL : trace (w, w->m, gap_open_score, k - 1, 1);
    const int misms    = (type != BT_M) ? OFFSET_NULL : trace (w, w->m, mismatch_score, k, 1);
    const int max_del = MAXI (del_ext, del_open), max_ins = MAXI (ins_ext, ins_open);
    const int max_all = MAXI (misms, MAXI (max_ins, max_del));
    if (type == BT_M) {
      const int num_matches = offset - max_all;
      if (num_matches < 0) return -1;
      for (int j = 0; j < num_matches; j++) { if (pattern[offset - k - 1 - j] != text[offset - 1 - j]) return -1; push_op (w, 'M'); }
      offset = max_all;
    }
    if (max_all == del_ext)       { push_op (w, 'D'); score = gap_extend_score; k++; type = BT_D; }
    else if (max_all == del_open) { push_op (w, 'D'); score = gap_open_score;   k++; type = BT_M; }
    else if (max_all == ins_ext)  { push_op (w, 'I'); score = gap_extend_score; k--; offset--; type = BT_I; }
    else if (max_all == ins_open) { push_op (w, 'I'); score = gap_open_score;   k--; offset--; type = BT_M; }
    else if (max_all == misms)    { push_op (w, 'X'); score = mismatch_score; offset--; }
    else return -1;
    v = offset - k; h = offset;
  }
  if (score == 0) { for (int j = 0; j < v; j++) push_op (w, 'M'); }
  else { while (v > 0) { push_op (w, 'D'); v--; } while (h > 0) { push_op (w, 'I'); h--; } }
  return 0;
}

int
orc_wfa_align (orc_wfa *w, const char *pattern, int plen, const char *text, int tlen, int max_score)
{ /* paper algorithm 1 */
  clear_wavefronts (w);
  reserve_scores (w, 0);
  w->m[0] = wf_new (w, 0, 0); w->m[0]->mem[0] = 0;
  w->cells = 1; w->max_width = 1;
  int score = 0;
  for (;;) {
    extend_wavefront (w, pattern, plen, text, tlen, score);
    if (end_reached (w, plen, tlen, score)) { if (backtrace (w, pattern, plen, text, tlen, score)) return -2; return score; }
    score++;
    if (score > max_score) return -1;
    reserve_scores (w, score);
    compute_wavefront (w, score);
  }
}

const char *orc_wfa_cigar (const orc_wfa *w, int *n_ops) { *n_ops = w->ops_end - w->ops_begin; return w->ops + w->ops_begin; }
int64_t orc_wfa_cells (const orc_wfa *w) { return w->cells; }
int orc_wfa_max_width (const orc_wfa *w) { return w->max_width; }

/* ---- uvaialign ---- */
void
orc_align_project (const char *ops, int n_ops, const char *seq, char *aln)
{ /* src/align.c:366-390 */
  int alg_pos = 0, text_pos = 0;
  for (int i = 0; i < n_ops; i++) switch (ops[i]) {
    case 'M': case 'X': aln[alg_pos++] = seq[text_pos++]; break;
    case 'I': text_pos++; break;
    case 'D': aln[alg_pos++] = '-'; break;
    default: break;
  }
  aln[alg_pos] = '\0';
}

static const orc_wfa_penalties uvaialign_penalties = { 0, 4, 6, 2 };   /* src/align.c:305 */

static int
uvaialign_max_score (int ref_len)
{ /* table size of affine_wavefronts_new_reduced (n_sites, 3 n_sites, ..): src/align.c:308 */
  return ref_len * 4 + 6 + 2 * ref_len * 2;
}

static int
align_query (orc_wfa *w, const char *ref, int ref_len, const char *seq, int seq_len, char *aln)
{ /* src/align.c:357-364 */
  int score = orc_wfa_align (w, ref, ref_len, seq, seq_len, uvaialign_max_score (ref_len));
  if (score >= 0) { int n; const char *ops = orc_wfa_cigar (w, &n); orc_align_project (ops, n, seq, aln); }
  return score;
}

int
orc_uvaialign_query (const char *ref, int ref_len, const char *seq, int seq_len, char *aln, int64_t *cells)
{
  orc_wfa *w = orc_wfa_new (uvaialign_penalties, 128, 512);       /* src/align.c:305-308 */
  int score = align_query (w, ref, ref_len, seq, seq_len, aln);
  if (cells) *cells = orc_wfa_cells (w);
  orc_wfa_del (w);
  return score;
}

int
orc_uvaialign_accepts (const char *seq, size_t seq_len, size_t ref_len, double ambiguity)
{ /* src/align.c:199-213; the fractions as orc_query_keep_valid reads them (biomcmc_count_sequence_acgt is absent) */
  if (3 * seq_len < 2 * ref_len || 2 * seq_len > 3 * ref_len) return 0;
  double l = seq_len ? (double) seq_len : 1.;
  double f_acgt = (double) orc_count_acgt (seq, seq_len) / l, f_n = 1. - (double) orc_count_non_N (seq, seq_len) / l;
  if (f_n > ambiguity) return 0;
  if (f_acgt < 1. - 1.1 * ambiguity) return 0;
  return 1;
}

void
orc_uvaialign_batch (const char *ref, int ref_len, int n, const char *const *seqs, const int *seq_len, char *aln, int *score)
{ /* src/align.c:224-233 with one aligner per thread (new_queue, src/align.c:304-310) */
#pragma omp parallel
  {
    orc_wfa *w = orc_wfa_new (uvaialign_penalties, 128, 512);
#pragma omp for schedule(dynamic)
    for (int c = 0; c < n; c++) score[c] = align_query (w, ref, ref_len, seqs[c], seq_len[c], aln + (size_t) c * ((size_t) ref_len + 1));
    orc_wfa_del (w);
  }
}

/* ---- independent checks ---- */
int
orc_gotoh_score (orc_wfa_penalties pen, const char *pattern, int plen, const char *text, int tlen)
{ /* M/I/D over the full (plen+1) x (tlen+1) table, two rows at a time; a gap of length n costs gap_opening + n * gap_extension */
  const int INF = INT_MAX / 4, o = pen.gap_opening, e = pen.gap_extension, x = pen.mismatch;
  int *M0 = (int *) malloc ((size_t) (tlen + 1) * sizeof (int)), *M1 = (int *) malloc ((size_t) (tlen + 1) * sizeof (int));
  int *D0 = (int *) malloc ((size_t) (tlen + 1) * sizeof (int)), *D1 = (int *) malloc ((size_t) (tlen + 1) * sizeof (int));
  M0[0] = 0; D0[0] = INF;
  for (int h = 1; h <= tlen; h++) { M0[h] = o + e * h; D0[h] = INF; }
  for (int v = 1; v <= plen; v++) {
    int ins = INF;                               /* gap in the pattern: consumes text, same row */
    D1[0] = o + e * v; M1[0] = D1[0];
    for (int h = 1; h <= tlen; h++) {
      D1[h] = MINI (M0[h] + o + e, D0[h] + e);   /* consumes pattern */
      ins = MINI (M1[h - 1] + o + e, ins + e);
      int diag = M0[h - 1] + (pattern[v - 1] == text[h - 1] ? pen.match : x);
      M1[h] = MINI (diag, MINI (D1[h], ins));
    }
    int *t = M0; M0 = M1; M1 = t; t = D0; D0 = D1; D1 = t;
  }
  int r = M0[tlen];
  free (M0); free (M1); free (D0); free (D1);
  return r;
}

int
orc_cigar_score (orc_wfa_penalties pen, const char *ops, int n_ops, const char *pattern, int plen, const char *text, int tlen)
{
  int v = 0, h = 0, score = 0; char last = 0;
  for (int i = 0; i < n_ops; i++) {
    switch (ops[i]) {
      case 'M': if (v >= plen || h >= tlen || pattern[v] != text[h]) return -1; v++; h++; score += pen.match; break;
      case 'X': if (v >= plen || h >= tlen || pattern[v] == text[h]) return -1; v++; h++; score += pen.mismatch; break;
      case 'I': if (h >= tlen) return -1; h++; score += pen.gap_extension + (last == 'I' ? 0 : pen.gap_opening); break;
      case 'D': if (v >= plen) return -1; v++; score += pen.gap_extension + (last == 'D' ? 0 : pen.gap_opening); break;
      default: return -1;
    }
    last = ops[i];
  }
  return (v == plen && h == tlen) ? score : -1;
}

/* seq_query.c -- implementation of fastaseq.h.
 * Behaviour follows the reference (reader: src/fastaseq.c:410-486; query set: src/fastaseq.c:698-841); own code. */
#include "fastaseq.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ reader */
readfasta_t
new_readfasta (const char *seqfilename)
{
  readfasta_t r = (readfasta_t) biomcmc_malloc (sizeof (struct readfasta_struct));
  memset (r, 0, sizeof *r);
  r->seqfile = biomcmc_open_compress (seqfilename, "r");
  return r;
}

static void
promote_pending_name (readfasta_t r)
{
  if (!r->next_name) return;
  free (r->name);
  r->name = r->next_name;
  r->next_name = NULL;
}

/* one pass over a sequence line: white space dropped, letters upper-cased (what remove_space_from_string + uppercase_string do
   in two), appended to the record; returns the number of characters appended */
static size_t
append_sequence_line (readfasta_t r, const char *line, size_t len)
{
  static unsigned char map[256];     /* 0 = white space (dropped), else the character to store */
  static int ready = 0;
  if (!ready) {
    for (int c = 0; c < 256; c++) map[c] = (unsigned char) ((c == ' ' || (c >= '\t' && c <= '\r')) ? 0 : ((c >= 'a' && c <= 'z') ? c - 32 : c));
    ready = 1;
  }
  r->seq = (char *) biomcmc_realloc (r->seq, r->seqlength + len + 1);
  char *w = r->seq + r->seqlength;
  /* the usual line is upper case with nothing to drop except the line end: one vectorisable pass finds out, then a plain copy */
  size_t body = len;
  while (body > 0 && (line[body - 1] == '\n' || line[body - 1] == '\r')) body--;
  size_t odd = 0;
  for (size_t i = 0; i < body; i++) { const unsigned char c = (unsigned char) line[i]; odd += (size_t) ((c <= ' ') | ((unsigned char) (c - 'a') < 26)); }
  if (!odd) { memcpy (w, line, body); w += body; }
  else for (size_t i = 0; i < len; i++) { const unsigned char m = map[(unsigned char) line[i]]; *w = (char) m; w += (m != 0); }
  *w = '\0';
  const size_t added = (size_t) (w - (r->seq + r->seqlength));
  r->seqlength += added;
  return added;
}

int
readfasta_next (readfasta_t r)
{
  if (!r->seqfile) return -1;
  int got;
  while ((got = biomcmc_getline_compress (&r->line_read, &r->linelength, r->seqfile)) != -1) {
    char *line = r->line_read;
    if (!nonempty_fasta_line (line)) continue;             /* stops at the first character that is not white space */
    char *header = (char *) memchr (line, '>', (size_t) got);
    if (header) {              /* a header closes the record being assembled (if any) */
      header++;
      r->newseq = true;
      promote_pending_name (r);
      size_t l = strlen (header);
      r->next_name = (char *) biomcmc_malloc (l + 1);
      memcpy (r->next_name, header, l + 1);
      if (r->seqlength) return (int) r->seqlength;
      continue;
    }
    if (r->newseq) { free (r->seq); r->seq = NULL; r->seqlength = 0; r->newseq = false; }
    append_sequence_line (r, line, (size_t) got);
  }
  biomcmc_close_compress (r->seqfile);
  r->seqfile = NULL;
  free (r->line_read); r->line_read = NULL;
  promote_pending_name (r);
  return (int) r->seqlength;
}

void
del_readfasta (readfasta_t r)
{
  if (!r) return;
  if (r->seqfile) biomcmc_close_compress (r->seqfile);
  free (r->line_read); free (r->seq); free (r->next_name); free (r->name);
  free (r);
}

void
quick_pairwise_score_acgt_and_valid (char *s1, char *s2, size_t nsites, int maxdist, int *score, size_t *idx)
{ /* src/fastaseq.c:585-596: both counters advance site by site until the mismatches reach maxdist */
  int differ = 0, both = 0;
  for (size_t k = 0; k < nsites && differ < maxdist; k++) {
    const char a = s1[idx[k]], b = s2[idx[k]];
    if (!is_site_acgt_pair_valid (a, b)) continue;
    both++;
    differ += is_site_acgt_distinct_pair (a, b);
  }
  score[0] = differ; score[1] = both;
}

int
quick_count_sequence_non_N (char *s, size_t nsites)
{ /* sits in the serial read loop once per reference (src/nearest.c:263): written as byte comparisons the compiler vectorises
     (the invalid set "NnXx-?Oo." of src/utils.c:263; 0xDF folds the letter case) instead of a call per site: 0.3 -> several GB/s */
  const unsigned char *u = (const unsigned char *) s;
  size_t invalid = 0;
  for (size_t i = 0; i < nsites; i++) {
    const unsigned char c = u[i], up = c & 0xDF;
    invalid += (size_t) ((up == 'N') | (up == 'X') | (up == 'O') | (c == '-') | (c == '?') | (c == '.'));
  }
  return (int) (nsites - invalid);
}

static int
count_acgt_sites (const char *s, size_t nsites)
{
  const unsigned char *u = (const unsigned char *) s;
  size_t n = 0;
  for (size_t i = 0; i < nsites; i++) {
    const unsigned char up = u[i] & 0xDF;
    n += (size_t) ((up == 'A') | (up == 'C') | (up == 'G') | (up == 'T'));
  }
  return (int) n;
}

int
quick_count_sequence_acgt (char *s, size_t nsites)
{ return count_acgt_sites (s, nsites); }

/* ------------------------------------------------------------------------------------------------ query set */
query_t
new_query_structure_from_alignment (alignment aln, int trim, int dist, int acgt)
{
  query_t qu = (query_t) biomcmc_malloc (sizeof (struct query_struct));
  memset (qu, 0, sizeof *qu);
  qu->aln = aln;
  qu->acgt = acgt != 0;
  if (trim < 0) trim = 0;
  if (trim > aln->nchar / 2.1) trim = (int) (aln->nchar / 2.1);       /* never trim away more than ~half */
  qu->trim = (size_t) trim;
  if (dist < 0) dist = 0;
  if (dist > (aln->nchar - 2 * trim) / 10) dist = (aln->nchar - 2 * trim) / 10;
  qu->dist = dist;
  return qu;
}

query_t
new_query_structure_from_fasta (char *filename, int trim, int dist, int acgt)
{
  return new_query_structure_from_alignment (read_fasta_alignment_from_file (filename, 0xf), trim, dist, acgt);
}

void uvaia_gpu_forget_query (query_t qu);     /* gpu_glue.c: the engine seq_ball_against_query_structure keeps for a query set */

void
del_query_structure (query_t qu)
{
  if (!qu) return;
  uvaia_gpu_forget_query (qu);
  free (qu->consensus); free (qu->idx_c); free (qu->idx_m); free (qu->idx);
  del_alignment (qu->aln);
  free (qu);
}

/* character classes of src/utils.c:255-295 as comparisons (0xDF folds the letter case), for the loops below that visit every
   character of every query; same sets as is_site_valid / is_site_acgt (tests/test_host_logic.py pins both for all 256 bytes) */
static inline int
char_is_valid (unsigned char c)
{
  const unsigned char up = c & 0xDF;
  return !((up == 'N') | (up == 'X') | (up == 'O') | (c == '-') | (c == '?') | (c == '.'));
}
static inline int
char_is_acgt (unsigned char c)
{
  const unsigned char up = c & 0xDF;
  return (up == 'A') | (up == 'C') | (up == 'G') | (up == 'T');
}
static inline int
usable (const query_t qu, char c) { return qu->acgt ? char_is_acgt ((unsigned char) c) : char_is_valid ((unsigned char) c); }

/* idx / idx_m / idx_c from the per-column result of the walk over the queries (src/fastaseq.c:758-776): consensus[] as the
   reference defines it, some_missing[] its miss[] flags */
static void
indices_from_columns (query_t qu, const unsigned char *some_missing)
{
  const int L = qu->aln->nchar, lo = (int) qu->trim, hi = L - (int) qu->trim;
  size_t span = hi > lo ? (size_t) (hi - lo) : 0;
  qu->idx_c = (size_t *) biomcmc_realloc (qu->idx_c, (span + 1) * sizeof (size_t));
  qu->idx_m = (size_t *) biomcmc_realloc (qu->idx_m, (span + 1) * sizeof (size_t));
  qu->idx   = (size_t *) biomcmc_realloc (qu->idx,   (span + 1) * sizeof (size_t));
  qu->n_idx_c = qu->n_idx_m = qu->n_idx = 0;
  for (int col = lo; col < hi; col++) {
    if (qu->consensus[col] == '#') qu->idx[qu->n_idx++] = (size_t) col;
    else if (qu->consensus[col] != 'N') { if (some_missing[col]) qu->idx_m[qu->n_idx_m++] = (size_t) col; else qu->idx_c[qu->n_idx_c++] = (size_t) col; }
  }
  fprintf (stderr, "Query sequence alignment: %d segregating, %d non-segregating sites with indels, and %d constant sites (all are used in comparisons)\n",
           qu->n_idx, qu->n_idx_m, qu->n_idx_c);
}

void
create_query_indices_given (query_t qu, const char *consensus, const unsigned char *some_missing)
{ /* the column walk was done elsewhere (uvaia_gpu_query_columns): take its result */
  const int L = qu->aln->nchar;
  if (!qu->consensus) qu->consensus = (char *) biomcmc_malloc ((size_t) (L > 0 ? L : 1));
  memcpy (qu->consensus, consensus, (size_t) L);
  indices_from_columns (qu, some_missing);
}

void
create_query_indices (query_t qu)
{ /* classify every column inside the trimmed window from what the usable query characters show there */
  const int L = qu->aln->nchar, lo = (int) qu->trim, hi = L - (int) qu->trim, n = qu->aln->ntax;
  char **s = qu->aln->character->string;
  initialise_acgt ();
  if (!qu->consensus) qu->consensus = (char *) biomcmc_malloc ((size_t) (L > 0 ? L : 1));
  memset (qu->consensus, 'N', (size_t) L);
  /* Per column: the first usable character, whether a later usable one differs (polymorphic) and whether some query is not
     usable there.  Walked query by query (rows are contiguous) in blocks of columns, one block per thread: same classes as a
     column-by-column walk over the queries, without its strided reads. */
  char *shared = (char *) biomcmc_malloc ((size_t) (L > 0 ? L : 1));
  unsigned char *flags = (unsigned char *) biomcmc_malloc ((size_t) (L > 0 ? L : 1));      /* bit 0: polymorphic, bit 1: some query missing */
  memset (shared, 'N', (size_t) (L > 0 ? L : 1));
  memset (flags, 0, (size_t) (L > 0 ? L : 1));
  const int block = 256;      /* 117 blocks at 29 903 columns: work for every thread of the host (2 048 left all but 15 idle) */
#pragma omp parallel for schedule(dynamic)
  for (int b0 = lo; b0 < hi; b0 += block) {
    const int b1 = b0 + block < hi ? b0 + block : hi;
    for (int j = 0; j < n; j++) {
      const char *row = s[j];
      for (int col = b0; col < b1; col++) {
        const char c = row[col];
        if (!usable (qu, c)) flags[col] |= 2;
        else if (shared[col] == 'N') shared[col] = c;
        else if (shared[col] != c) flags[col] |= 1;
      }
    }
  }
  for (int col = lo; col < hi; col++) {
    qu->consensus[col] = (flags[col] & 1) ? '#' : shared[col];
    flags[col] = (unsigned char) ((flags[col] & 2) ? 1 : 0);
  }
  free (shared);
  indices_from_columns (qu, flags);
  free (flags);
}

typedef struct { int key, pos; } keyed_pos;
static int
by_key_then_pos (const void *a, const void *b)
{
  const keyed_pos *x = (const keyed_pos *) a, *y = (const keyed_pos *) b;
  return x->key != y->key ? (x->key < y->key ? -1 : 1) : x->pos - y->pos;
}

void
reorder_query_structure (query_t qu)
{ /* least resolved queries first (fewest usable sites inside the trimmed window); ties keep file order */
  const int n = qu->aln->ntax;
  const size_t span = (size_t) qu->aln->nchar - 2 * qu->trim;
  keyed_pos *kp = (keyed_pos *) biomcmc_malloc ((size_t) (n > 0 ? n : 1) * sizeof *kp);
  int *order = (int *) biomcmc_malloc ((size_t) (n > 0 ? n : 1) * sizeof (int));
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    char *s = qu->aln->character->string[i] + qu->trim;
    kp[i].pos = i;
    kp[i].key = qu->acgt ? count_acgt_sites (s, span) : quick_count_sequence_non_N (s, span);
  }
  qsort (kp, (size_t) n, sizeof *kp, by_key_then_pos);
  for (int i = 0; i < n; i++) order[i] = kp[i].pos;
  char_vector_reorder_strings_from_external_order (qu->aln->character, order);
  char_vector_reorder_strings_from_external_order (qu->aln->taxlabel, order);
  free (kp); free (order);
}

/* do a and b differ at any polymorphic column where both are usable? */
static int
queries_conflict (const query_t qu, const char *a, const char *b)
{
  for (int j = 0; j < qu->n_idx; j++) {
    char x = a[qu->idx[j]], y = b[qu->idx[j]];
    if (x != y && usable (qu, x) && usable (qu, y)) return 1;
  }
  return 0;
}

/* over the given columns: -1 a is usable somewhere b is not (only), +1 the converse, 0 same pattern, 0xff both */
static int
resolution_order (const query_t qu, const char *a, const char *b, const size_t *cols, int n_cols)
{
  int verdict = 0;
  for (int j = 0; j < n_cols; j++) {
    int ua = usable (qu, a[cols[j]]), ub = usable (qu, b[cols[j]]);
    if (ua == ub) continue;
    int side = ua > ub ? -1 : 1;
    if (verdict == -side) return 0xff;
    verdict = side;
  }
  return verdict;
}

void
exclude_redundant_query_sequences (query_t qu, int keep_more_resolved)
{
  exclude_redundant_query_sequences_given (qu, keep_more_resolved, NULL);
}

/* agree (nullable): ntax x ntax bytes, agree[j * ntax + i] != 0 when sequences i and j differ at no polymorphic column where
   both are usable -- the pair test of the loop below, computed elsewhere (uvaia_gpu_agree_on_polymorphic) */
void
exclude_redundant_query_sequences_given (query_t qu, int keep_more_resolved, const unsigned char *agree)
{
  if (!qu->consensus) biomcmc_error ("I can only exclude sequences after indices are created");
  const int n = qu->aln->ntax;
  char **s = qu->aln->character->string;
  int *alive = (int *) biomcmc_malloc ((size_t) (n > 0 ? n : 1) * sizeof (int)), n_alive = 0;
  for (int i = 0; i < n; i++) alive[i] = 1;
  for (int i = 0; i < n - 1; i++) for (int j = i + 1; j < n; j++) {
    if (!alive[i] || !alive[j]) continue;
    if (agree ? !agree[(size_t) j * n + i] : queries_conflict (qu, s[i], s[j])) continue;
    int on_poly = resolution_order (qu, s[i], s[j], qu->idx, qu->n_idx);
    if (on_poly > 1) continue;
    int on_const = resolution_order (qu, s[i], s[j], qu->idx_m, qu->n_idx_m);
    if (on_const > 1) continue;
    if (!on_poly && !on_const) alive[j] = 0;          /* identical patterns: the later copy goes */
    int sum = on_poly + on_const;                       /* <0: i more resolved; >0: j more resolved; 0: complementary */
    if (!sum) continue;
    int i_is_less_resolved = sum > 0;
    if (keep_more_resolved) alive[i_is_less_resolved ? i : j] = 0;
    else                    alive[i_is_less_resolved ? j : i] = 0;
  }
  for (int i = 0; i < n; i++) if (alive[i]) alive[n_alive++] = i;
  char_vector_reduce_to_valid_strings (qu->aln->character, alive, n_alive);
  char_vector_reduce_to_valid_strings (qu->aln->taxlabel, alive, n_alive);
  qu->aln->ntax = n_alive;
  free (alive);
}

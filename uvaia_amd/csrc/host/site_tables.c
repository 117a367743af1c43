/* site_tables.c -- implementation of utils.h (character classes: src/utils.c:255-295; query QC: src/utils.c:10-48). */
#include "utils.h"

#include <ctype.h>
#include <stdlib.h>
#include <string.h>

enum { SITE_ACGT = 1, SITE_INVALID = 2 };
static unsigned char site_class[256];
static int site_class_ready = 0;

void
initialise_acgt (void)
{
  if (site_class_ready) return;
  memset (site_class, 0, sizeof site_class);
  for (const char *p = "ACGTacgt"; *p; p++)  site_class[(unsigned char) *p] |= SITE_ACGT;
  for (const char *p = "NnXx-?Oo."; *p; p++) site_class[(unsigned char) *p] |= SITE_INVALID;
  site_class_ready = 1;
}

#define CLS(c) (site_class[(unsigned char) (c)])

int is_site_acgt (char s1)  { initialise_acgt (); return (CLS (s1) & SITE_ACGT) != 0; }
int is_site_valid (char s1) { initialise_acgt (); return (CLS (s1) & SITE_INVALID) == 0; }
int is_site_pair_valid (char s1, char s2)      { return is_site_valid (s1) && is_site_valid (s2); }
int is_site_acgt_pair_valid (char s1, char s2) { return is_site_acgt (s1) && is_site_acgt (s2); }
int is_site_acgt_distinct_pair (char s1, char s2) { return is_site_acgt_pair_valid (s1, s2) && s1 != s2; }

void
upper_kseq (char *s, unsigned l)
{
  for (unsigned i = 0; i < l; i++) s[i] = (char) toupper ((unsigned char) s[i]);
}

void
uvaia_keep_only_valid_sequences (alignment aln, double ambiguity, bool check_aligned)
{
  int *kept = (int *) biomcmc_malloc ((size_t) (aln->ntax > 0 ? aln->ntax : 1) * sizeof (int)), n_kept = 0;
  long common_length = 0;   /* 0 = none seen yet, -1 = lengths differ */
  /* the per-sequence work (upper-casing, character census) is independent: all host threads; the decisions stay in file order */
  const int ns = aln->character->nstrings;
  double *fr = (double *) biomcmc_malloc ((size_t) (ns > 0 ? ns : 1) * 3 * sizeof (double));
#pragma omp parallel for schedule(static)
  for (int i = 0; i < ns; i++) {
    size_t len = aln->character->nchars[i];
    if (len < 5) continue;
    upper_kseq (aln->character->string[i], (unsigned) len);
    biomcmc_count_sequence_acgt (aln->character->string[i], len, fr + (size_t) i * 3);
  }
  for (int i = 0; i < ns; i++) {
    size_t len = aln->character->nchars[i];
    const double *f = fr + (size_t) i * 3;
    if (len < 5) { fprintf (stderr, "Sequence %s is too short ( = %zu sites), limit is hardcoded at 5bps.\n", aln->taxlabel->string[i], len); continue; }
    if (f[2] > ambiguity) { fprintf (stderr, "Sequence %s has proportion of N etc. (=%lf) above threshold of %lf\n", aln->taxlabel->string[i], f[2], ambiguity); continue; }
    if (f[0] < 1. - 1.1 * ambiguity) { fprintf (stderr, "Sequence %s has proportion of ACGT (=%lf) below threshold of %lf\n", aln->taxlabel->string[i], f[0], 1. - 1.1 * ambiguity); continue; }
    kept[n_kept++] = i;
    if (!common_length) common_length = (long) len;
    else if (common_length != -1 && (size_t) common_length != len) common_length = -1;
  }
  if (check_aligned && common_length == -1) {
    biomcmc_warning ("Reference sequences in file %s are not aligned.\n", aln->filename);
    biomcmc_error ("You can use uvaialign (or mafft, or minimap2) to align them against the same reference.");
  }
  /* aln->nchar is the length of the FIRST record of the file; if that record was dropped above and the kept ones are longer or
   * shorter, trim, idx and the packed query rows would be laid out for the wrong length (reads past the end of every row) */
  if (n_kept && common_length > 0) aln->nchar = (int) common_length;
  char_vector_reduce_to_valid_strings (aln->character, kept, n_kept);
  char_vector_reduce_to_valid_strings (aln->taxlabel, kept, n_kept);
  aln->ntax = n_kept;
  free (kept); free (fr);
}

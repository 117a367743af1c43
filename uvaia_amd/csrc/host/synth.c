/* synth.c -- see synth.h.  Model: a root genome with Wuhan-Hu-1-like base composition; a fixed set of polymorphic
 * columns (~27 % of the length) with Zipf-like mutation weights; a few dozen lineage founders a handful of mutations
 * away from the root; each sequence = its lineage founder + private mutations, N runs at both ends plus
 * amplicon-dropout-like internal runs, sparse gaps and sparse IUPAC partial codes. */
#include "synth.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define N_LINEAGES 48

struct uvaia_synth {
  int nchar, preset, n_poly;
  uint64_t seed;
  char *root;
  int *poly_col;            /* polymorphic columns */
  double *poly_cdf;         /* cumulative mutation weight over poly_col */
  char *founder[N_LINEAGES];
  double lineage_cdf[N_LINEAGES];
};

static inline uint64_t
splitmix64 (uint64_t *state)
{
  uint64_t z = (*state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline double u01 (uint64_t *s) { return (double) (splitmix64 (s) >> 11) * (1.0 / 9007199254740992.0); }
static inline int below (uint64_t *s, int n) { return (int) (u01 (s) * n); }

static int
poisson (uint64_t *s, double lambda)
{ /* multiplication method; lambda stays well below 700 here */
  double limit = exp (-lambda), p = 1.0;
  int k = 0;
  do { k++; p *= u01 (s); } while (p > limit);
  return k - 1;
}

static int
pick_cdf (const double *cdf, int n, double u)
{
  int lo = 0, hi = n - 1;
  while (lo < hi) { int mid = (lo + hi) / 2; if (cdf[mid] < u) lo = mid + 1; else hi = mid; }
  return lo;
}

static void
mutate (const uvaia_synth *g, char *seq, uint64_t *s, int count)
{
  static const char acgt[4] = {'A', 'C', 'G', 'T'};
  for (int i = 0; i < count; i++) {
    int col = g->poly_col[pick_cdf (g->poly_cdf, g->n_poly, u01 (s))];
    char c;
    do c = acgt[below (s, 4)]; while (c == seq[col]);
    seq[col] = c;
  }
}

uvaia_synth *
uvaia_synth_new (int nchar, uint64_t seed, int preset)
{
  if (nchar < 64) return NULL;
  uvaia_synth *g = (uvaia_synth *) calloc (1, sizeof *g);
  g->nchar = nchar; g->seed = seed; g->preset = preset;
  uint64_t s = seed ^ 0xA5A5A5A55A5A5A5AULL;
  g->root = (char *) malloc ((size_t) nchar);
  for (int i = 0; i < nchar; i++) {
    double u = u01 (&s);
    g->root[i] = u < 0.299 ? 'A' : u < 0.483 ? 'C' : u < 0.679 ? 'G' : 'T';
  }
  /* polymorphic columns: a random 26.8 % of the columns (8 011 of 29 903 in the bundled alignment) */
  g->n_poly = (int) (nchar * 0.268);
  int *perm = (int *) malloc ((size_t) nchar * sizeof (int));
  for (int i = 0; i < nchar; i++) perm[i] = i;
  for (int i = 0; i < g->n_poly; i++) { int j = i + below (&s, nchar - i), t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
  g->poly_col = (int *) malloc ((size_t) g->n_poly * sizeof (int));
  g->poly_cdf = (double *) malloc ((size_t) g->n_poly * sizeof (double));
  double total = 0;
  for (int i = 0; i < g->n_poly; i++) { g->poly_col[i] = perm[i]; total += 1.0 / pow (i + 4.0, 0.9); g->poly_cdf[i] = total; }
  for (int i = 0; i < g->n_poly; i++) g->poly_cdf[i] /= total;
  free (perm);
  total = 0;
  for (int l = 0; l < N_LINEAGES; l++) { total += 1.0 / (l + 1.5); g->lineage_cdf[l] = total; }
  for (int l = 0; l < N_LINEAGES; l++) g->lineage_cdf[l] /= total;
  for (int l = 0; l < N_LINEAGES; l++) {
    g->founder[l] = (char *) malloc ((size_t) nchar);
    memcpy (g->founder[l], g->root, (size_t) nchar);
    mutate (g, g->founder[l], &s, poisson (&s, 7.0));
  }
  return g;
}

void
uvaia_synth_free (uvaia_synth *g)
{
  if (!g) return;
  for (int l = 0; l < N_LINEAGES; l++) free (g->founder[l]);
  free (g->root); free (g->poly_col); free (g->poly_cdf); free (g);
}

int uvaia_synth_nchar (const uvaia_synth *g) { return g ? g->nchar : 0; }

static void
fill_n (char *seq, int L, int start, int len)
{
  if (start < 0) { len += start; start = 0; }
  if (start + len > L) len = L - start;
  if (len > 0) memset (seq + start, 'N', (size_t) len);
}

static void
one_sequence (const uvaia_synth *g, uint64_t index, char *seq, int *non_n)
{
  const int L = g->nchar;
  const double scale = (double) L / 29903.0;
  uint64_t s = g->seed * 0x9E3779B97F4A7C15ULL + (index + 1) * 0xD1B54A32D192ED03ULL;
  splitmix64 (&s);
  memcpy (seq, g->founder[pick_cdf (g->lineage_cdf, N_LINEAGES, u01 (&s))], (size_t) L);
  mutate (g, seq, &s, poisson (&s, 8.0));
  /* sparse IUPAC partial codes (~1.7 per genome; Y R K M S W far more often than D H V B) and single-site gaps (0.08 %) */
  for (int i = poisson (&s, 1.7 * scale); i > 0; i--) seq[below (&s, L)] = u01 (&s) < 0.93 ? "YRKMSW"[below (&s, 6)] : "DHVB"[below (&s, 4)];
  for (int i = poisson (&s, 0.0008 * L); i > 0; i--) seq[below (&s, L)] = '-';
  /* invalid runs */
  int lead = (int) ((30 + poisson (&s, 60.0)) * scale), trail = (int) ((20 + poisson (&s, 55.0)) * scale);
  fill_n (seq, L, 0, lead);
  fill_n (seq, L, L - trail, trail);
  if (g->preset == UVAIA_SYNTH_BUNDLED_LIKE) {
    /* target invalid fraction ~ log-normal(median .13, sigma .75), capped at .45 so that every sequence passes -A 0.5 */
    double z = sqrt (-2.0 * log (1.0 - u01 (&s))) * cos (6.283185307179586 * u01 (&s));
    double f = 0.13 * exp (0.75 * z);
    if (f > 0.45) f = 0.45;
    int runs = (int) floor ((f * L - lead - trail) / (350.0 * scale) + 0.5);
    for (int r = 0; r < runs; r++) fill_n (seq, L, below (&s, L), (int) ((300 + below (&s, 101)) * scale));
  } else if (u01 (&s) < 0.3) {
    fill_n (seq, L, below (&s, L), (int) ((300 + below (&s, 101)) * scale));
  }
  if (non_n) {
    int v = 0;
    for (int i = 0; i < L; i++) v += (seq[i] != 'N' && seq[i] != '-');
    *non_n = v;
  }
}

void
uvaia_synth_generate (const uvaia_synth *g, uint64_t first_index, int n, char *rows, size_t pitch, int *non_n)
{
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) one_sequence (g, first_index + (uint64_t) i, rows + (size_t) i * pitch, non_n ? non_n + i : NULL);
}

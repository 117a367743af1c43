/* gpu_glue.c -- see gpu_glue.h. */
#include "gpu_glue.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static uvaia_gpu_query
as_gpu_query (query_t query)
{
  uvaia_gpu_query q;
  memset (&q, 0, sizeof q);
  q.n_query = query->aln->ntax;
  q.nchar = query->aln->nchar;
  q.seq = (const char *const *) query->aln->character->string;
  q.consensus = query->consensus;
  q.idx_c = query->idx_c; q.idx_m = query->idx_m; q.idx = query->idx;
  q.n_idx_c = query->n_idx_c; q.n_idx_m = query->n_idx_m; q.n_idx = query->n_idx;
  q.trim = query->trim;
  q.acgt = query->acgt ? 1 : 0;
  return q;
}

int
uvaia_gpu_open_query (uvaia_gpu_ctx **ctx, query_t query, int heap_size, int device, size_t max_pool)
{
  uvaia_gpu_query q = as_gpu_query (query);
  return uvaia_gpu_open (ctx, &q, heap_size, device, max_pool);
}

int
uvaia_gpu_open_query_tuned (uvaia_gpu_ctx **ctx, query_t query, int heap_size, int device, size_t max_pool, const uvaia_gpu_tuning *tuning)
{
  uvaia_gpu_query q = as_gpu_query (query);
  return uvaia_gpu_open_tuned (ctx, &q, heap_size, device, max_pool, tuning);
}

int
uvaia_gpu_group_open_query (uvaia_gpu_group **group, query_t query, int heap_size, const int *devices, int n_devices, size_t max_pool, size_t piece_refs)
{
  uvaia_gpu_query q = as_gpu_query (query);
  return uvaia_gpu_group_open (group, &q, heap_size, devices, n_devices, max_pool, piece_refs);
}

int
uvaia_parse_device_list (const char *text, int *devices, int max)
{
  int n = 0;
  const char *p = text;
  while (*p) {
    char *end;
    long a = strtol (p, &end, 10), b;
    if (end == p || a < 0) return 0;
    b = a;
    if (*end == '-') { p = end + 1; b = strtol (p, &end, 10); if (end == p || b < a) return 0; }
    for (long d = a; d <= b; d++) { if (n == max) return 0; devices[n++] = (int) d; }
    if (*end == ',') end++; else if (*end) return 0;
    p = end;
  }
  return n;
}

static int
collect_heaps (uvaia_gpu_ctx *ctx, uvaia_gpu_group *group, heap_t *heap, const char *(*name_of) (int64_t, void *), void *user)
{
  uvaia_gpu_ctx *first = group ? uvaia_gpu_group_member (group, 0) : ctx;
  const int nq = uvaia_gpu_n_query (first), slots = uvaia_gpu_heap_slots (first);
  int *n = (int *) malloc ((size_t) nq * sizeof (int)), *T = (int *) malloc ((size_t) nq * sizeof (int));
  int *scores = (int *) malloc ((size_t) nq * (slots + 1) * UVAIA_GPU_NSCORE * sizeof (int));
  int64_t *ord = (int64_t *) malloc ((size_t) nq * (slots + 1) * sizeof (int64_t));
  int rc = (n && T && scores && ord) ? (group ? uvaia_gpu_group_drain (group, n, T, scores, ord) : uvaia_gpu_drain (ctx, n, T, scores, ord)) : UVAIA_GPU_ENOMEM;
  for (int q = 0; q < nq && !rc; q++) {
    heap_t h = heap[q];
    if (h->heap_size != slots) { rc = UVAIA_GPU_EINVAL; break; }
    for (int s = 1; s <= n[q]; s++) {
      size_t e = (size_t) q * (slots + 1) + s;
      free (h->seq[s].name);
      const char *nm = name_of ? name_of (ord[e], user) : NULL;
      h->seq[s].name = nm ? strdup (nm) : NULL;
      memcpy (h->seq[s].score, scores + e * UVAIA_GPU_NSCORE, UVAIA_GPU_NSCORE * sizeof (int));
    }
    h->n = n[q];
    h->max_incompatible = T[q];
  }
  free (n); free (T); free (scores); free (ord);
  return rc;
}

int
uvaia_gpu_collect_heaps (uvaia_gpu_ctx *ctx, heap_t *heap, const char *(*name_of) (int64_t, void *), void *user)
{ return collect_heaps (ctx, NULL, heap, name_of, user); }

int
uvaia_gpu_group_collect_heaps (uvaia_gpu_group *group, heap_t *heap, const char *(*name_of) (int64_t, void *), void *user)
{ return collect_heaps (NULL, group, heap, name_of, user); }

/* ---- seq_ball_against_query_structure (src/fastaseq.h:78): the reference's one-sequence entry point of the radius search, kept for
 * callers of the fastaseq API.  The engine of the query set is opened at the first call and reused until the query set goes away. */
static struct { query_t qu; uvaia_gpu_ctx *ctx; } ball_engine = {NULL, NULL};
/* The reference calls seq_ball_against_query_structure from inside "#pragma omp parallel for" (src/ball.c:248-250).  The engine
 * and its staging / result buffers are one per process: opening, forgetting and the search itself are serialised here, so the kept
 * entry point is safe to call from any number of threads (it is then as fast as one thread: batch through uvaia_gpu_ball instead). */
static pthread_mutex_t ball_lock = PTHREAD_MUTEX_INITIALIZER;

static void
forget_query_locked (query_t qu)
{
  if (!ball_engine.ctx || (qu && ball_engine.qu != qu)) return;
  uvaia_gpu_close (ball_engine.ctx);
  ball_engine.ctx = NULL; ball_engine.qu = NULL;
}

void
uvaia_gpu_forget_query (query_t qu)
{
  pthread_mutex_lock (&ball_lock);
  forget_query_locked (qu);
  pthread_mutex_unlock (&ball_lock);
}

void
seq_ball_against_query_structure (char **seq, int *min_dist, int ball_radius, query_t qu)
{
  if (!seq || !*seq || !min_dist || !qu) biomcmc_error ("seq_ball_against_query_structure: NULL argument");
  pthread_mutex_lock (&ball_lock);      /* (biomcmc_error exits the process: no unlock on those paths) */
  if (ball_engine.qu != qu) {
    forget_query_locked (NULL);
    if (uvaia_gpu_open_query (&ball_engine.ctx, qu, 2, -1, 64)) biomcmc_error ("radius search on the GPU: %s", uvaia_gpu_last_error (NULL));
    ball_engine.qu = qu;
  }
  const char *one[1] = {*seq};
  if (uvaia_gpu_ball (ball_engine.ctx, one, 1, ball_radius, min_dist)) biomcmc_error ("radius search on the GPU: %s", uvaia_gpu_last_error (ball_engine.ctx));
  pthread_mutex_unlock (&ball_lock);
}

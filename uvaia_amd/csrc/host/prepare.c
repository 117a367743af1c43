/* prepare.c -- see prepare.h. */
#include "prepare.h"
#include "gpu_glue.h"
#include <stdlib.h>
#include <string.h>

static int prepare_device = -1; /* GPU the preparation's device steps run on (-1: the current one); the command lines pass --device */

/* exclude_redundant_query_sequences (src/fastaseq.c:797-841) with its O(Q^2) pair test on the device: the engine is opened on
 * the still unpruned query set, the queries go through it as if they were references (uvaia_gpu_agree_on_polymorphic), and the
 * order-dependent walk over the pairs runs here on the finished matrix.  Same survivors as the host-only function. */
static void
exclude_redundant_query_sequences_device (query_t query, int keep_resolved)
{
  const int n = query->aln->ntax;
  const size_t batch = n < 2048 ? (size_t) n : 2048;
  uvaia_gpu_ctx *gpu = NULL;
  if (uvaia_gpu_open_query (&gpu, query, 1, prepare_device, batch))
    biomcmc_error ("pruning redundant queries on the GPU: %s (uvaia_set_prune_mode(1) runs the serial host loop instead)", uvaia_gpu_last_error (NULL));
  unsigned char *agree = (unsigned char *) biomcmc_malloc ((size_t) n * n);
  for (int a = 0; a < n; a += (int) batch) {
    const int m = n - a < (int) batch ? n - a : (int) batch;
    if (uvaia_gpu_agree_on_polymorphic (gpu, (const char *const *) query->aln->character->string + a, m, agree + (size_t) a * n))
      biomcmc_error ("pruning redundant queries on the GPU: %s", uvaia_gpu_last_error (gpu));
  }
  uvaia_gpu_close (gpu);
  exclude_redundant_query_sequences_given (query, keep_resolved, agree);
  free (agree);
}

/* the pair test costs O(Q^2 x polymorphic columns) on one host thread (14 s at 3 000 queries): from this many queries on it
   runs on the device; the O(Q x L) column walk of create_query_indices from 2 048 queries on (below that the upload of the rows
   costs what the host threads need for the walk).  uvaia_set_prune_mode() overrides both (tests). */
#define UVAIA_PRUNE_DEVICE_FROM 512
#define UVAIA_COLUMNS_DEVICE_FROM 2048
static int prune_mode = 0;     /* 0 = by query count, 1 = host, 2 = device */
void uvaia_set_prepare_device (int device) { prepare_device = device < 0 ? -1 : device; }
void uvaia_set_prune_mode (int mode) { prune_mode = (mode == 1 || mode == 2) ? mode : 0; }
static int
prune_on_device (int ntax)
{
  if (prune_mode) return prune_mode == 2;
  return ntax >= UVAIA_PRUNE_DEVICE_FROM;
}

/* create_query_indices (src/fastaseq.c:732-777) with its walk over all query characters on the device */
static void
create_query_indices_where_it_pays (query_t query)
{
  const int n = query->aln->ntax, L = query->aln->nchar;
  const int device = prune_mode ? prune_mode == 2 : n >= UVAIA_COLUMNS_DEVICE_FROM;
  if (!device || L < 1) { create_query_indices (query); return; }
  char *consensus = (char *) biomcmc_malloc ((size_t) L);
  unsigned char *missing = (unsigned char *) biomcmc_malloc ((size_t) L);
  if (uvaia_gpu_query_columns ((const char *const *) query->aln->character->string, n, L, query->trim, query->acgt, prepare_device, consensus, missing))
    biomcmc_error ("column classes of the queries on the GPU: %s (uvaia_set_prune_mode(1) runs the host loop instead)", uvaia_gpu_last_error (NULL));
  create_query_indices_given (query, consensus, missing);
  free (consensus); free (missing);
}

query_t
uvaia_prepare_query (alignment aln, int trim, int dist, int acgt, double ambig_q, int keep_resolved, int is_ball)
{
  if (ambig_q < 0.001) ambig_q = 0.001;
  if (ambig_q > 1.) ambig_q = 1.;
  query_t query = new_query_structure_from_alignment (aln, trim, dist, acgt);
  uvaia_keep_only_valid_sequences (query->aln, ambig_q, true);
  if (query->aln->ntax < 1) return query;
  create_query_indices_where_it_pays (query);
  reorder_query_structure (query);
  if (is_ball || keep_resolved) {
    if (prune_on_device (query->aln->ntax)) exclude_redundant_query_sequences_device (query, keep_resolved);
    else exclude_redundant_query_sequences (query, keep_resolved);
    create_query_indices_where_it_pays (query);
  }
  return query;
}

query_t
uvaia_prepare_query_from_arrays (int ntax, int nchar, const char *const *seqs, const char *const *names,
                                 int trim, int dist, int acgt, double ambig_q, int keep_resolved, int is_ball)
{
  return uvaia_prepare_query (new_alignment_from_arrays (ntax, nchar, seqs, names), trim, dist, acgt, ambig_q, keep_resolved, is_ball);
}

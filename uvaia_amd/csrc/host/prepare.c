/* prepare.c -- see prepare.h. */
#include "prepare.h"

query_t
uvaia_prepare_query (alignment aln, int trim, int dist, int acgt, double ambig_q, int keep_resolved, int is_ball)
{
  if (ambig_q < 0.001) ambig_q = 0.001;
  if (ambig_q > 1.) ambig_q = 1.;
  query_t query = new_query_structure_from_alignment (aln, trim, dist, acgt);
  uvaia_keep_only_valid_sequences (query->aln, ambig_q, true);
  if (query->aln->ntax < 1) return query;
  create_query_indices (query);
  reorder_query_structure (query);
  if (is_ball || keep_resolved) {
    exclude_redundant_query_sequences (query, keep_resolved);
    create_query_indices (query);
  }
  return query;
}

query_t
uvaia_prepare_query_from_arrays (int ntax, int nchar, const char *const *seqs, const char *const *names,
                                 int trim, int dist, int acgt, double ambig_q, int keep_resolved, int is_ball)
{
  return uvaia_prepare_query (new_alignment_from_arrays (ntax, nchar, seqs, names), trim, dist, acgt, ambig_q, keep_resolved, is_ball);
}

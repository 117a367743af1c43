/* prepare.h -- the query preparation sequence of the command lines, callable on in-memory sequences. */
#ifndef UVAIA_HOST_PREPARE_H
#define UVAIA_HOST_PREPARE_H
#include "fastaseq.h"
#ifdef __cplusplus
extern "C" {
#endif
/* What `uvaia` does between reading the query file and building the queue (src/nearest.c:175-176,203-224), or, with
 * is_ball, what `uvaiaball` does (src/ball.c:153-154,174-194: it always prunes redundant queries).  Takes ownership of
 * aln.  The result may hold zero sequences. */
query_t uvaia_prepare_query (alignment aln, int trim, int dist, int acgt, double ambig_q, int keep_resolved, int is_ball);
query_t uvaia_prepare_query_from_arrays (int ntax, int nchar, const char *const *seqs, const char *const *names,
                                         int trim, int dist, int acgt, double ambig_q, int keep_resolved, int is_ball);
/* where the query preprocessing runs -- the O(Q^2) pair test of exclude_redundant_query_sequences and the O(Q x L) column walk of
 * create_query_indices: 0 = on the device from 512 / 2 048 queries on (default), 1 = host, 2 = device */
void uvaia_set_prune_mode (int mode);
/* the GPU those device steps run on (-1 = the current device, the default): what --device / the first of --devices selects */
void uvaia_set_prepare_device (int device);

#ifdef __cplusplus
}
#endif
#endif

/*
 * uvaia_align.h -- C ABI of the MI355X (gfx950) gap-affine wavefront aligner behind `uvaialign`.
 *
 * Drop-in boundary for the alignment loop of the reference (quadram-institute-bioscience/uvaia, paths under
 * /root/reference): src/align.c keeps one WFA aligner per thread (new_queue, src/align.c:286-313), and for every pool
 * of query sequences runs align_query (src/align.c:357-364: affine_wavefronts_clear, affine_wavefronts_align,
 * update_query_aligned) under `#pragma omp parallel for` (src/align.c:224-233).  A maintainer replaces that loop by
 * uvaia_align_batch() and keeps the rest of main() (readers, filters of src/align.c:199-213, writers).
 *
 * What one call computes per query, bit for bit the CPU restatement's result (oracle/wfa_oracle.h; the WFA library
 * itself is an absent submodule: parity UNPINNED beyond the optimal gap-affine score, see that header):
 *   - the gap-affine wavefront alignment of the reference sequence (pattern) against the query (text) with penalties
 *     {match 0, mismatch, gap opening, gap extension} and the adaptive wavefront reduction (minimum wavefront length,
 *     maximum distance threshold) of affine_wavefronts_new_reduced (src/align.c:305-309: {0,4,6,2}, 128, 512);
 *   - its projection on the reference's columns (update_query_aligned, src/align.c:366-390): M/X copy the query
 *     character, I drops it, D writes '-': a row of exactly ref_len characters.
 *
 * Conventions: plain C; 0 on success or a negative UVAIA_ALIGN_E* code, never exit(); uvaia_align_last_error() gives
 * the message.  Sequences are bytes compared for equality as they are (the reference's reader upper-cases them,
 * src/fastaseq.c:462-466).  One aligner = one GPU = one host thread at a time.  There is no CPU fallback:
 * uvaia_align_open fails with UVAIA_ALIGN_ENODEV without a gfx950 device.
 */
#ifndef UVAIA_ALIGN_H
#define UVAIA_ALIGN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  UVAIA_ALIGN_OK      =  0,
  UVAIA_ALIGN_EINVAL  = -1,
  UVAIA_ALIGN_ENODEV  = -2,
  UVAIA_ALIGN_ENOMEM  = -3,   /* device/host allocation failed, or a query needs more wavefront memory than the workspace holds */
  UVAIA_ALIGN_EHIP    = -4,
  UVAIA_ALIGN_ESTATE  = -5
};

typedef struct uvaia_aligner uvaia_aligner;

/* affine_penalties_t + the reduction arguments of affine_wavefronts_new_reduced (src/align.c:305-309) */
typedef struct {
  int mismatch, gap_opening, gap_extension;      /* match is 0 (src/align.c:305); mismatch and extension > 0, opening >= 0, mismatch and opening + extension below 64 */
  int min_wavefront_length;                      /* <= 0: complete wavefronts (no reduction) */
  int max_distance_threshold;
  size_t workspace_bytes;                        /* device memory for the wavefronts of the queries in flight; 0 = starts at 4 GB and grows,
                                                    up to three quarters of the free memory, when queries find it exhausted */
  int max_blocks;                                /* queries in flight (one block of four wavefronts each); 0 = as many as the chip holds */
} uvaia_align_options;

/* the reference's values: {4, 6, 2, 128, 512, 0, 0} */
void uvaia_align_default_options (uvaia_align_options *opt);

/* new_queue (src/align.c:286-313): the reference sequence goes to the device once.  opt may be NULL (defaults). */
int  uvaia_align_open (uvaia_aligner **out, const char *ref, int ref_len, int device, const uvaia_align_options *opt);
void uvaia_align_close (uvaia_aligner *a);
const char *uvaia_align_last_error (const uvaia_aligner *a);

/* One pool (src/align.c:224-233 + :357-390): n queries, seq[i] of seq_len[i] bytes.
 *   aln   : n rows of ref_len + 1 bytes (row i = cq->aln[i], NUL-terminated)
 *   score : n alignment scores (the reference does not report them; nullable)
 * A query whose score would pass the reference's table size (ref_len * 4 + 6 + 4 * ref_len, the allocation of
 * affine_wavefronts_new_reduced (L, 3L, ..)) fails the call with UVAIA_ALIGN_EINVAL: the reference has no check there. */
int  uvaia_align_batch (uvaia_aligner *a, const char *const *seq, const int *seq_len, int n, char *aln, int *score);

/* The same in three steps, for callers that keep a pool resident (bench.py times uvaia_align_run with the queries in HBM):
 * load copies the queries to the device, run aligns the loaded pool and returns when its kernels are done (rows and scores stay
 * on the device), fetch copies them back. */
int  uvaia_align_load (uvaia_aligner *a, const char *const *seq, const int *seq_len, int n);
int  uvaia_align_load_block (uvaia_aligner *a, const char *bytes, const int64_t *offsets /* n + 1 */, int n);
int  uvaia_align_run (uvaia_aligner *a);
int  uvaia_align_sync (uvaia_aligner *a);
int  uvaia_align_fetch (uvaia_aligner *a, char *aln, int *score);

/* work of the last run: M-wavefront cells computed, wavefront bytes written + read by the recurrences (13 + 20 per cell),
 * kernel passes (queries that find the workspace's pool empty are run again in a less crowded pass), kernel time in ms */
int  uvaia_align_stats (uvaia_aligner *a, unsigned long long *cells, double *wavefront_bytes, int *passes, double *kernel_ms);

#ifdef __cplusplus
}
#endif
#endif

/*
 * fastaseq.h -- streaming FASTA reader and the prepared query set (host side).
 * Names, struct fields and call order follow the reference's src/fastaseq.h:32-48,70-83 for the nearest-neighbour path
 * (the cluster/medoid half of that header serves uvaiaclust only and is not provided).
 */
#ifndef UVAIA_HOST_FASTASEQ_H
#define UVAIA_HOST_FASTASEQ_H

#include "utils.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct readfasta_struct *readfasta_t;
typedef struct query_struct *query_t;

struct readfasta_struct {
  file_compress_t seqfile;
  char *line_read, *next_name;
  char *name, *seq;              /* current record; the caller may steal either pointer by setting it to NULL */
  size_t linelength, seqlength;
  bool newseq;
};

struct query_struct {
  alignment aln;                 /* all query sequences, in memory */
  char *consensus;               /* per column: 'N' no usable query, '#' queries disagree, else the shared character */
  size_t *idx_c, *idx_m, *idx, trim;   /* constant & complete / constant with missing / polymorphic columns */
  int n_idx_c, n_idx_m, n_idx, dist;
  bool acgt;
};

/* one record per call; returns its length, or -1 once the file is exhausted */
readfasta_t new_readfasta (const char *seqfilename);
int readfasta_next (readfasta_t rfas);
void del_readfasta (readfasta_t rfas);

/* the --acgt scoring kernel over the sites idx[0..nsites): score[0] = sites where both are ACGT and differ, score[1] = sites where
 * both are ACGT; stops once score[0] reaches maxdist (src/fastaseq.c:585-596, src/fastaseq.h:74).  Scalar, on the host: the one-pair
 * entry point the reference's header exports; the batch loops of src/nearest.c:293-306 run on the GPU (include/uvaia_gpu.h). */
void quick_pairwise_score_acgt_and_valid (char *s1, char *s2, size_t nsites, int maxdist, int *score, size_t *idx);
int quick_count_sequence_non_N (char *s, size_t nsites);   /* valid sites over the given span */
int quick_count_sequence_acgt (char *s, size_t nsites);    /* ACGT sites (src/fastaseq.c:650-656; not in the reference's header) */

/* for uvaiaball (src/fastaseq.h:78, src/fastaseq.c:660-696): *min_dist = what the reference leaves there for ONE sequence and radius
 * ball_radius.  Runs on the GPU through uvaia_gpu_ball() with an engine kept for `qu` (made at the first call, dropped by
 * del_query_structure or by a call with another query set); errors are fatal, as everywhere in this API.  The batch loop of
 * src/ball.c:248-251 should call uvaia_gpu_ball() with the whole batch instead.
 * Threading: safe to call from several threads at once, as the reference does inside "#pragma omp parallel for" (src/ball.c:248-250):
 * the one engine per process, its opening and uvaia_gpu_forget_query() are serialised by a mutex (calls do not run concurrently). */
void seq_ball_against_query_structure (char **seq, int *min_dist, int ball_radius, query_t qu);
query_t new_query_structure_from_fasta (char *filename, int trim, int dist, int acgt);
query_t new_query_structure_from_alignment (alignment aln, int trim, int dist, int acgt);   /* takes ownership of aln */
void del_query_structure (query_t qu);
void create_query_indices (query_t qu);
/* same result from a column walk done elsewhere (the device: uvaia_gpu_query_columns); not in the reference's header */
void create_query_indices_given (query_t qu, const char *consensus, const unsigned char *some_missing);
void reorder_query_structure (query_t qu);
void exclude_redundant_query_sequences (query_t qu, int keep_more_resolved);
/* same walk with the O(Q^2) pair test handed in (NULL: computed here); not in the reference's header */
void exclude_redundant_query_sequences_given (query_t qu, int keep_more_resolved, const unsigned char *agree);

#ifdef __cplusplus
}
#endif
#endif

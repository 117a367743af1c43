/*
 * wfa_oracle.h -- CPU restatement of the `uvaialign` path: gap-affine wavefront alignment of one query
 * against one reference and the projection of the result onto the reference's columns.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as uvaia_oracle.h): only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product (uvaia_amd/, include/) never links or calls it.
 *
 * What it restates (paths under /root/reference):
 *   - the per-query call sequence                 src/align.c:357-364 (clear, align, project)
 *   - CIGAR -> aligned row (M/X copy, I drop,     src/align.c:366-390
 *     D emits '-')
 *   - the aligner set-up                          src/align.c:304-309: penalties {match 0, mismatch 4, gap opening 6,
 *                                                 gap extension 2}, affine_wavefronts_new_reduced (L, 3L, .., 128, 512, ..)
 *   - the query filters of the read loop          src/align.c:199-213
 *   - the aligner itself                          THIRD PARTY, ABSENT: submodules/WFA (.gitmodules:4-6,
 *                                                 https://github.com/leomrtns/WFA.git, a fork of smarco/WFA v1; the
 *                                                 submodule directory is empty, pinned commit unknown).
 *
 * PARITY UNPINNED for the aligner: the reference holds no test, fixture or documented output of uvaialign.
 * The wavefront recurrences, the exact extension, the adaptive reduction ("reduced" wavefronts: minimum
 * wavefront length, maximum distance threshold) and the backtrace order are restated from the published
 * algorithm (Marco-Sola, Moure, Moreto, Espinosa, "Fast gap-affine pairwise alignment using the wavefront
 * algorithm", Bioinformatics 37(4), 2021, sections 2.3-2.4 and algorithms 1-3) in the shape of its v1 C
 * implementation (gap_affine/affine_wavefront_{align,extend,reduction,backtrace}.c).  What the published
 * algorithm fixes and this file therefore pins: the optimal gap-affine score whenever the reduction never
 * trims (checked against an independent O(nm) Gotoh recurrence, orc_gotoh_score).  What it does not fix and
 * is this restatement's choice: the offset of a missing diagonal (-10), the tie order of the backtrace
 * (deletion extend, deletion open, insertion extend, insertion open, mismatch), and that the backtrace looks
 * a diagonal up within the same (reduced) limits the recurrences used.
 */
#ifndef WFA_ORACLE_H
#define WFA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int match, mismatch, gap_opening, gap_extension; } orc_wfa_penalties;
typedef struct orc_wfa orc_wfa;

/* min_wavefront_length <= 0: complete wavefronts (no reduction) */
orc_wfa *orc_wfa_new (orc_wfa_penalties pen, int min_wavefront_length, int max_distance_threshold);
void orc_wfa_del (orc_wfa *w);
/* aligns pattern (vertical, v) against text (horizontal, h); returns the score, or -1 when max_score is passed
 * (the reference sizes its tables for min(plen,3plen)*4 + 6 + 2*|plen-3plen| scores and has no such check) */
int  orc_wfa_align (orc_wfa *w, const char *pattern, int plen, const char *text, int tlen, int max_score);
const char *orc_wfa_cigar (const orc_wfa *w, int *n_ops);   /* 'M','X','I','D', first operation first */
int64_t orc_wfa_cells (const orc_wfa *w);                     /* M-wavefront cells computed by the last call */
int  orc_wfa_max_width (const orc_wfa *w);

/* src/align.c:366-390 */
void orc_align_project (const char *ops, int n_ops, const char *seq, char *aln /* >= plen + 1 bytes */);
/* src/align.c:357-364 with the aligner of src/align.c:304-309; returns the score (-1: see orc_wfa_align) */
int  orc_uvaialign_query (const char *ref, int ref_len, const char *seq, int seq_len, char *aln, int64_t *cells);
/* src/align.c:199-213: 1 = aligned, 0 = rejected (size, N fraction, ACGT fraction; fractions as src/utils.c:23-31 reads them) */
int  orc_uvaialign_accepts (const char *seq, size_t seq_len, size_t ref_len, double ambiguity);
/* many queries over OpenMP threads (src/align.c:224-233); aln: n rows of ref_len + 1 bytes */
void orc_uvaialign_batch (const char *ref, int ref_len, int n, const char *const *seqs, const int *seq_len, char *aln, int *score);

/* independent check: optimal gap-affine score by the full O(plen * tlen) recurrence (Gotoh 1982) */
int  orc_gotoh_score (orc_wfa_penalties pen, const char *pattern, int plen, const char *text, int tlen);
/* score of a given CIGAR under the penalties, or -1 if it does not spell pattern -> text */
int  orc_cigar_score (orc_wfa_penalties pen, const char *ops, int n_ops, const char *pattern, int plen, const char *text, int tlen);

#ifdef __cplusplus
}
#endif
#endif

/*
 * utils.h -- site classification and query quality control (host side).
 * Same entry points as the reference's src/utils.h:12,20-25 for the functions the nearest-neighbour path uses;
 * the legacy/WFA helpers of that header (src/utils.h:13-18) are outside the hot path and not provided.
 */
#ifndef UVAIA_HOST_UTILS_H
#define UVAIA_HOST_UTILS_H

#include "biomcmc_lite.h"

#ifdef __cplusplus
extern "C" {
#endif

/* drops queries that are too short (<5), too ambiguous (fraction of N-like > ambiguity) or too poor in ACGT
 * (< 1 - 1.1*ambiguity); upper-cases the survivors; with check_aligned, unequal lengths are fatal.  src/utils.c:10-48 */
void uvaia_keep_only_valid_sequences (alignment aln, double ambiguity, bool check_aligned);
void upper_kseq (char *s, unsigned l);

void initialise_acgt (void);                      /* idempotent; the predicates below call it themselves */
int is_site_acgt_distinct_pair (char s1, char s2);
int is_site_acgt_pair_valid (char s1, char s2);
int is_site_pair_valid (char s1, char s2);
int is_site_acgt (char s1);
int is_site_valid (char s1);

#ifdef __cplusplus
}
#endif
#endif

/*
 * synth.h -- deterministic generator of SARS-CoV-2-shaped aligned sequences for the benchmark and the parity tests at
 * BASELINE sizes.  Sequence i depends only on (seed, i), so any shard can be produced independently on
 * any rank and the CPU baseline can be given exactly the sequences the GPU holds.  Not part of the reference.
 */
#ifndef UVAIA_HOST_SYNTH_H
#define UVAIA_HOST_SYNTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uvaia_synth uvaia_synth;

enum { UVAIA_SYNTH_BUNDLED_LIKE = 0,   /* invalid-site fraction: median ~.17, mean ~.20, long tail to ~.45 */
       UVAIA_SYNTH_CLEAN = 1 };        /* ~1-3 % invalid sites */

uvaia_synth *uvaia_synth_new (int nchar, uint64_t seed, int preset);
void uvaia_synth_free (uvaia_synth *g);
/* writes n rows of nchar upper-case bytes (row stride `pitch`), sequence numbers first_index .. first_index+n-1;
 * non_n (nullable) receives the valid-site count of each row */
void uvaia_synth_generate (const uvaia_synth *g, uint64_t first_index, int n, char *rows, size_t pitch, int *non_n);
int uvaia_synth_nchar (const uvaia_synth *g);

#ifdef __cplusplus
}
#endif
#endif

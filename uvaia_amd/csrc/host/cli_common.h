/* cli_common.h -- shared pieces of the two command lines (names -> ordinals, output helpers). */
#ifndef UVAIA_HOST_CLI_COMMON_H
#define UVAIA_HOST_CLI_COMMON_H

#include <stdlib.h>
#include <string.h>

#include "biomcmc_lite.h"

#define UVAIA_PACKAGE_STRING "uvaia 2.0.2 (MI355X engine)"
#define UVAIA_PACKAGE_VERSION "2.0.2"

/* names of the references that entered some heap, indexed by the ordinal the engine knows them by */
typedef struct { char **name; int64_t cap; } name_table;

static inline void
name_table_set (name_table *t, int64_t ordinal, const char *name)
{
  if (ordinal >= t->cap) {
    int64_t ncap = t->cap ? t->cap : 1024;
    while (ncap <= ordinal) ncap *= 2;
    t->name = (char **) biomcmc_realloc (t->name, (size_t) ncap * sizeof (char *));
    memset (t->name + t->cap, 0, (size_t) (ncap - t->cap) * sizeof (char *));
    t->cap = ncap;
  }
  if (!t->name[ordinal]) t->name[ordinal] = strdup (name);
}

static inline const char *
name_table_get (int64_t ordinal, void *user)
{
  name_table *t = (name_table *) user;
  return (ordinal >= 0 && ordinal < t->cap) ? t->name[ordinal] : NULL;
}

static inline void
name_table_free (name_table *t)
{
  for (int64_t i = 0; i < t->cap; i++) free (t->name[i]);
  free (t->name);
  t->name = NULL; t->cap = 0;
}

/* "<prefix>.aln.xz" with room to swap the suffix later (the reference builds its file names the same way) */
static inline char *
outfile_from_prefix (const char *prefix, size_t *length)
{
  *length = strlen (prefix);
  char *f = (char *) biomcmc_malloc (*length + 16);
  memcpy (f, prefix, *length);
  strcpy (f + *length, ".aln.xz");
  return f;
}

static inline void
write_fasta_record (file_compress_t out, const char *name, const char *seq)
{
  int bad = 0;
  bad += biomcmc_write_compress (out, ">") != 1;
  bad += biomcmc_write_compress (out, name) != (int) strlen (name);
  bad += biomcmc_write_compress (out, "\n") != 1;
  bad += biomcmc_write_compress (out, seq) != (int) strlen (seq);
  bad += biomcmc_write_compress (out, "\n") != 1;
  if (bad) fprintf (stderr, "File %s may not be correctly compressed, %d error%s occurred when saving sequence %s.\n", out->filename, bad, bad > 1 ? "s" : "", name);
}

#endif

/*
 * biomcmc_lite.h -- the small slice of biomcmc-lib's data model and runtime that uvaia's hot-path sources use,
 * provided natively so the host code needs no third-party library (the reference links the whole of biomcmc-lib;
 * its submodule is empty in /root/reference, see SURVEY.md 8c for the symbol list this covers).
 * Field names follow the reference's usage (query->aln->character->string[i], ->taxlabel->nchars[i], ...),
 * so code written against src/fastaseq.h:41-48 compiles against this header.
 */
#ifndef UVAIA_BIOMCMC_LITE_H
#define UVAIA_BIOMCMC_LITE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct char_vector_struct *char_vector;
struct char_vector_struct {
  char **string;       /* owned strings */
  size_t *nchars;      /* length of each string */
  int nstrings;
};

typedef struct hashtable_struct *hashtable;

typedef struct alignment_struct *alignment;
struct alignment_struct {
  int ntax, nchar;
  char_vector character, taxlabel;
  hashtable taxlabel_hash;
  char *filename;
};

typedef struct file_compress_struct *file_compress_t;
struct file_compress_struct {
  char *filename;
  FILE *fp;
  int piped;           /* fp comes from popen() */
  int writing, at_eof; /* opened for writing; a read hit end of data (then a failing decompressor means truncated input) */
};

/* memory / diagnostics (biomcmc_error = message + exit, biomcmc_warning = message and continue) */
void *biomcmc_malloc (size_t size);
void *biomcmc_realloc (void *ptr, size_t size);
void biomcmc_error (const char *fmt, ...);
void biomcmc_warning (const char *fmt, ...);
void biomcmc_get_time (int64_t time[2]);
double biomcmc_update_elapsed_time (int64_t time[2]);   /* seconds since time[], which is refreshed */

/* strings */
char_vector new_char_vector (int nstrings);
void del_char_vector (char_vector vec);
void char_vector_link_string_at_position (char_vector vec, char *string, int position);   /* takes ownership */
void char_vector_reduce_to_valid_strings (char_vector vec, int *valid, int n_valid);       /* valid[] increasing */
void char_vector_reorder_strings_from_external_order (char_vector vec, int *order);
char *remove_space_from_string (char *string);
char *uppercase_string (char *string);
bool nonempty_fasta_line (char *line);

/* name lookup used by --exclude_self */
hashtable new_hashtable (int size);
void del_hashtable (hashtable ht);
void insert_hashtable (hashtable ht, const char *key, int value);
int lookup_hashtable (hashtable ht, const char *key);      /* -1 if absent */

/* alignments */
alignment read_fasta_alignment_from_file (const char *filename, int flag);
alignment new_alignment_from_arrays (int ntax, int nchar, const char *const *seqs, const char *const *names);
void del_alignment (alignment aln);
void biomcmc_count_sequence_acgt (const char *seq, size_t length, double result[3]);

/* possibly compressed text streams (xz, gz, bz2 by magic number / suffix, through the system tools) */
file_compress_t biomcmc_open_compress (const char *path, const char *mode);
void biomcmc_close_compress (file_compress_t fc);
int biomcmc_getline_compress (char **lineptr, size_t *n, file_compress_t fc);   /* -1 at end of file */
int biomcmc_write_compress (file_compress_t fc, const char *str);              /* bytes written */

#ifdef __cplusplus
}
#endif
#endif

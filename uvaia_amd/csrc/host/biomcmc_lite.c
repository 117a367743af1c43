/* biomcmc_lite.c -- see biomcmc_lite.h. */
#define _GNU_SOURCE
#include "biomcmc_lite.h"

#include <ctype.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

void *
biomcmc_malloc (size_t size)
{
  void *p = malloc (size ? size : 1);
  if (!p) biomcmc_error ("out of memory asking for %zu bytes", size);
  return p;
}

void *
biomcmc_realloc (void *ptr, size_t size)
{
  void *p = realloc (ptr, size ? size : 1);
  if (!p) biomcmc_error ("out of memory asking for %zu bytes", size);
  return p;
}

void
biomcmc_error (const char *fmt, ...)
{
  va_list ap;
  fprintf (stderr, "uvaia error: ");
  va_start (ap, fmt); vfprintf (stderr, fmt, ap); va_end (ap);
  fprintf (stderr, "\n");
  exit (EXIT_FAILURE);
}

void
biomcmc_warning (const char *fmt, ...)
{
  va_list ap;
  fprintf (stderr, "uvaia warning: ");
  va_start (ap, fmt); vfprintf (stderr, fmt, ap); va_end (ap);
  fprintf (stderr, "\n");
}

void
biomcmc_get_time (int64_t t[2])
{
  struct timespec ts;
  clock_gettime (CLOCK_MONOTONIC, &ts);
  t[0] = (int64_t) ts.tv_sec; t[1] = (int64_t) ts.tv_nsec;
}

double
biomcmc_update_elapsed_time (int64_t t[2])
{
  int64_t now[2];
  biomcmc_get_time (now);
  double secs = (double) (now[0] - t[0]) + 1.e-9 * (double) (now[1] - t[1]);
  t[0] = now[0]; t[1] = now[1];
  return secs;
}

/* ---- strings ---- */
char_vector
new_char_vector (int nstrings)
{
  char_vector v = (char_vector) biomcmc_malloc (sizeof (struct char_vector_struct));
  v->nstrings = nstrings;
  v->string = (char **) biomcmc_malloc ((size_t) (nstrings > 0 ? nstrings : 1) * sizeof (char *));
  v->nchars = (size_t *) biomcmc_malloc ((size_t) (nstrings > 0 ? nstrings : 1) * sizeof (size_t));
  for (int i = 0; i < nstrings; i++) { v->string[i] = NULL; v->nchars[i] = 0; }
  return v;
}

void
del_char_vector (char_vector v)
{
  if (!v) return;
  for (int i = 0; i < v->nstrings; i++) free (v->string[i]);
  free (v->string); free (v->nchars); free (v);
}

void
char_vector_link_string_at_position (char_vector v, char *string, int position)
{
  free (v->string[position]);
  v->string[position] = string;
  v->nchars[position] = string ? strlen (string) : 0;
}

void
char_vector_reduce_to_valid_strings (char_vector v, int *valid, int n_valid)
{
  int out = 0;
  for (int i = 0; i < v->nstrings; i++) {
    if (out < n_valid && valid[out] == i) { v->string[out] = v->string[i]; v->nchars[out] = v->nchars[i]; out++; }
    else free (v->string[i]);
  }
  v->nstrings = n_valid;
}

void
char_vector_reorder_strings_from_external_order (char_vector v, int *order)
{
  char **s = (char **) biomcmc_malloc ((size_t) v->nstrings * sizeof (char *));
  size_t *n = (size_t *) biomcmc_malloc ((size_t) v->nstrings * sizeof (size_t));
  for (int i = 0; i < v->nstrings; i++) { s[i] = v->string[order[i]]; n[i] = v->nchars[order[i]]; }
  memcpy (v->string, s, (size_t) v->nstrings * sizeof (char *));
  memcpy (v->nchars, n, (size_t) v->nstrings * sizeof (size_t));
  free (s); free (n);
}

char *
remove_space_from_string (char *string)
{
  char *w = string;
  for (char *r = string; *r; r++) if (!isspace ((unsigned char) *r)) *w++ = *r;
  *w = '\0';
  return string;
}

char *
uppercase_string (char *string)
{
  for (char *p = string; *p; p++) *p = (char) toupper ((unsigned char) *p);
  return string;
}

bool
nonempty_fasta_line (char *line)
{
  for (char *p = line; *p; p++) if (!isspace ((unsigned char) *p)) return true;
  return false;
}

/* ---- name -> index table (open addressing, FNV-1a) ---- */
struct hashtable_struct { int size; char **key; int *value; };

static uint64_t
fnv1a (const char *s)
{
  uint64_t h = 1469598103934665603ULL;
  for (; *s; s++) { h ^= (unsigned char) *s; h *= 1099511628211ULL; }
  return h;
}

hashtable
new_hashtable (int size)
{
  hashtable ht = (hashtable) biomcmc_malloc (sizeof (struct hashtable_struct));
  int cap = 16;
  while (cap < 2 * size + 1) cap <<= 1;
  ht->size = cap;
  ht->key = (char **) calloc ((size_t) cap, sizeof (char *));
  ht->value = (int *) calloc ((size_t) cap, sizeof (int));
  if (!ht->key || !ht->value) biomcmc_error ("out of memory for hashtable");
  return ht;
}

void
del_hashtable (hashtable ht)
{
  if (!ht) return;
  for (int i = 0; i < ht->size; i++) free (ht->key[i]);
  free (ht->key); free (ht->value); free (ht);
}

void
insert_hashtable (hashtable ht, const char *key, int value)
{
  uint64_t i = fnv1a (key) & (uint64_t) (ht->size - 1);
  while (ht->key[i]) { if (!strcmp (ht->key[i], key)) return; i = (i + 1) & (uint64_t) (ht->size - 1); }
  ht->key[i] = strdup (key); ht->value[i] = value;
}

int
lookup_hashtable (hashtable ht, const char *key)
{
  uint64_t i = fnv1a (key) & (uint64_t) (ht->size - 1);
  while (ht->key[i]) { if (!strcmp (ht->key[i], key)) return ht->value[i]; i = (i + 1) & (uint64_t) (ht->size - 1); }
  return -1;
}

/* ---- streams ---- */
static const char *
decompressor_for (const char *path)
{
  unsigned char magic[6] = {0};
  FILE *f = fopen (path, "rb");
  if (!f) return NULL;
  size_t n = fread (magic, 1, sizeof magic, f);
  fclose (f);
  if (n >= 6 && !memcmp (magic, "\xFD" "7zXZ\0", 6)) return "xz -dc";
  if (n >= 2 && magic[0] == 0x1f && magic[1] == 0x8b) return "gzip -dc";
  if (n >= 3 && !memcmp (magic, "BZh", 3)) return "bzip2 -dc";
  return "";
}

static int
ends_with (const char *s, const char *suffix)
{
  size_t a = strlen (s), b = strlen (suffix);
  return a >= b && !strcmp (s + a - b, suffix);
}

static char *
shell_quote (const char *s)
{ /* 'abc' with embedded quotes rewritten as '\'' */
  size_t n = strlen (s), extra = 0;
  for (size_t i = 0; i < n; i++) if (s[i] == '\'') extra += 3;
  char *q = (char *) biomcmc_malloc (n + extra + 3), *w = q;
  *w++ = '\'';
  for (size_t i = 0; i < n; i++) { if (s[i] == '\'') { memcpy (w, "'\\''", 4); w += 4; } else *w++ = s[i]; }
  *w++ = '\''; *w = '\0';
  return q;
}

static int
tool_available (const char *tool)
{
  char cmd[128];
  snprintf (cmd, sizeof cmd, "command -v %s >/dev/null 2>&1", tool);
  return system (cmd) == 0;
}

file_compress_t
biomcmc_open_compress (const char *path, const char *mode)
{
  file_compress_t fc = (file_compress_t) biomcmc_malloc (sizeof (struct file_compress_struct));
  fc->filename = strdup (path); fc->fp = NULL; fc->piped = 0; fc->writing = (mode[0] != 'r'); fc->at_eof = 0;
  char *quoted = shell_quote (path), *cmd = (char *) biomcmc_malloc (strlen (quoted) + 64);
  if (mode[0] == 'r') {
    const char *tool = decompressor_for (path);
    if (!tool) { free (quoted); free (cmd); biomcmc_error ("cannot open file %s for reading", path); }
    if (*tool) { sprintf (cmd, "%s %s", tool, quoted); fc->fp = popen (cmd, "r"); fc->piped = 1; }
    else fc->fp = fopen (path, "r");
    if (fc->fp) setvbuf (fc->fp, NULL, _IOFBF, 1u << 22);     /* 30 kb lines: the default 4 KiB buffer costs a refill per eighth of a line */
  } else {
    const char *tool = NULL;   /* the reference tries xz, then bz2, then gz, then plain text (src/nearest.c:234) */
    if (ends_with (path, ".xz") && tool_available ("xz")) tool = "xz -T0 -c";   /* all cores: the dump of a large search is hundreds of MB of text */
    else if (ends_with (path, ".bz2") && tool_available ("bzip2")) tool = "bzip2 -c";
    else if (ends_with (path, ".gz") && tool_available ("gzip")) tool = "gzip -c";
    if (tool) { sprintf (cmd, "%s > %s", tool, quoted); fc->fp = popen (cmd, "w"); fc->piped = 1; }
    else fc->fp = fopen (path, "w");
  }
  free (quoted); free (cmd);
  if (!fc->fp) biomcmc_error ("cannot open file %s", path);
  return fc;
}

void
biomcmc_close_compress (file_compress_t fc)
{
  if (!fc) return;
  int status = 0;
  if (fc->fp) status = fc->piped ? pclose (fc->fp) : fclose (fc->fp);
  /* A decompressor that fails in the middle of a stream looks like end of data to getline(): a search over a truncated or
   * corrupt reference file must not pass for a complete one.  (A reader closed before its end of data kills the tool with
   * SIGPIPE: that is not an error.)  A failing compressor (disk full, killed xz) leaves an unusable output file: say so. */
  if (status != 0 && !fc->writing && fc->at_eof) {
    char *name = strdup (fc->filename);
    free (fc->filename); free (fc);
    biomcmc_error ("reading %s failed: the decompressor reported an error (truncated or corrupt file?)", name);
  }
  if (status != 0 && fc->writing) biomcmc_warning ("writing %s failed: the compressor or the file system reported an error\n", fc->filename);
  free (fc->filename); free (fc);
}

int
biomcmc_getline_compress (char **lineptr, size_t *n, file_compress_t fc)
{
  ssize_t got = getline (lineptr, n, fc->fp);
  if (got < 0) { fc->at_eof = 1; return -1; }
  while (got > 0 && ((*lineptr)[got - 1] == '\n' || (*lineptr)[got - 1] == '\r')) (*lineptr)[--got] = '\0';   /* lines come back without their terminator */
  return (int) got;
}

int
biomcmc_write_compress (file_compress_t fc, const char *str)
{
  size_t len = strlen (str);
  return (int) fwrite (str, 1, len, fc->fp);
}

/* ---- alignments ---- */
alignment
new_alignment_from_arrays (int ntax, int nchar, const char *const *seqs, const char *const *names)
{
  alignment aln = (alignment) biomcmc_malloc (sizeof (struct alignment_struct));
  aln->ntax = ntax; aln->nchar = nchar; aln->taxlabel_hash = NULL; aln->filename = strdup ("(memory)");
  aln->character = new_char_vector (ntax);
  aln->taxlabel = new_char_vector (ntax);
  for (int i = 0; i < ntax; i++) {
    char *s = (char *) biomcmc_malloc ((size_t) nchar + 1);
    memcpy (s, seqs[i], (size_t) nchar); s[nchar] = '\0';
    aln->character->string[i] = s; aln->character->nchars[i] = (size_t) nchar;
    aln->taxlabel->string[i] = strdup (names ? names[i] : ""); aln->taxlabel->nchars[i] = strlen (aln->taxlabel->string[i]);
  }
  return aln;
}

alignment
read_fasta_alignment_from_file (const char *filename, int flag)
{ /* whole file in memory: names are the header lines without '>', sequences lose all white space */
  (void) flag;
  file_compress_t fc = biomcmc_open_compress (filename, "r");
  int cap = 64, n = 0;
  char **seq = (char **) biomcmc_malloc ((size_t) cap * sizeof (char *)), **name = (char **) biomcmc_malloc ((size_t) cap * sizeof (char *));
  size_t *len = (size_t *) biomcmc_malloc ((size_t) cap * sizeof (size_t));
  char *line = NULL; size_t linecap = 0;
  while (biomcmc_getline_compress (&line, &linecap, fc) != -1) {
    if (!nonempty_fasta_line (line)) continue;
    char *gt = strchr (line, '>');
    if (gt) {
      if (n == cap) { cap *= 2; seq = biomcmc_realloc (seq, (size_t) cap * sizeof (char *)); name = biomcmc_realloc (name, (size_t) cap * sizeof (char *)); len = biomcmc_realloc (len, (size_t) cap * sizeof (size_t)); }
      gt++;
      size_t l = strlen (gt);
      while (l && (gt[l - 1] == '\n' || gt[l - 1] == '\r')) gt[--l] = '\0';
      name[n] = strdup (gt); seq[n] = NULL; len[n] = 0; n++;
    } else if (n) {
      remove_space_from_string (line);
      size_t l = strlen (line);
      seq[n - 1] = (char *) biomcmc_realloc (seq[n - 1], len[n - 1] + l + 1);
      memcpy (seq[n - 1] + len[n - 1], line, l + 1);
      len[n - 1] += l;
    }
  }
  free (line);
  biomcmc_close_compress (fc);
  alignment aln = (alignment) biomcmc_malloc (sizeof (struct alignment_struct));
  aln->ntax = n; aln->nchar = n ? (int) len[0] : 0; aln->taxlabel_hash = NULL; aln->filename = strdup (filename);
  aln->character = new_char_vector (n); aln->taxlabel = new_char_vector (n);
  for (int i = 0; i < n; i++) {
    if (!seq[i]) { seq[i] = strdup (""); }
    aln->character->string[i] = seq[i]; aln->character->nchars[i] = len[i];
    aln->taxlabel->string[i] = name[i]; aln->taxlabel->nchars[i] = strlen (name[i]);
  }
  free (seq); free (name); free (len);
  return aln;
}

void
del_alignment (alignment aln)
{
  if (!aln) return;
  del_char_vector (aln->character); del_char_vector (aln->taxlabel);
  del_hashtable (aln->taxlabel_hash);
  free (aln->filename); free (aln);
}

void
biomcmc_count_sequence_acgt (const char *seq, size_t length, double result[3])
{ /* result[0] = fraction of ACGT, result[1] = fraction of partially ambiguous (other valid) characters,
     result[2] = fraction of "N etc." = the invalid set of src/utils.c:263.  Only [0] and [2] are read by
     uvaia_keep_only_valid_sequences (src/utils.c:23-31); the absent biomcmc original is not pinned. */
  size_t acgt = 0, bad = 0;
  for (size_t i = 0; i < length; i++) {
    switch (seq[i]) {
      case 'A': case 'C': case 'G': case 'T': case 'a': case 'c': case 'g': case 't': acgt++; break;
      case 'N': case 'n': case 'X': case 'x': case '-': case '?': case 'O': case 'o': case '.': bad++; break;
      default: break;
    }
  }
  double l = length ? (double) length : 1.;
  result[0] = (double) acgt / l; result[2] = (double) bad / l; result[1] = 1. - result[0] - result[2];
}

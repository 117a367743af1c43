/* uvdb.c -- see uvdb.h.  Own code. */
#define _GNU_SOURCE
#include "uvdb.h"

#include <fcntl.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

struct uvdb_writer_struct {
  FILE *f;
  struct uvdb_header h;
  uint64_t tiles_written;
  /* sections kept in memory until close (small next to the planes): valid-site counts, side rows, names, exception runs */
  int32_t *non_n; size_t nonn_cap;
  FILE *side_tmp;                      /* side rows go through a temporary file: 256 B per reference */
  uint64_t *name_idx; size_t idx_cap; char *names; size_t names_len, names_cap;
  uint64_t *exc_idx; uvdb_exc *exc; size_t exc_len, exc_cap;
};

static uint64_t align64 (uint64_t x) { return (x + 63u) & ~(uint64_t) 63u; }

static int
pad_to (FILE *f, uint64_t off)
{
  static const char zero[64] = {0};
  long at = ftell (f);
  if (at < 0 || (uint64_t) at > off) return -1;
  while ((uint64_t) at < off) {
    size_t n = (size_t) ((off - (uint64_t) at) < sizeof zero ? (off - (uint64_t) at) : sizeof zero);
    if (fwrite (zero, 1, n, f) != n) return -1;
    at += (long) n;
  }
  return 0;
}

uvdb_writer
uvdb_create (const char *filename, int nchar, size_t tile_bytes, int side_row_ints, double ref_ambiguity)
{
  uvdb_writer w = (uvdb_writer) calloc (1, sizeof *w);
  if (!w) return NULL;
  w->f = fopen (filename, "wb");
  w->side_tmp = tmpfile ();
  if (!w->f || !w->side_tmp) { if (w->f) fclose (w->f); if (w->side_tmp) fclose (w->side_tmp); free (w); return NULL; }
  memcpy (w->h.magic, UVDB_MAGIC, 8);
  w->h.version = 1; w->h.nchar = (uint32_t) nchar; w->h.W4 = (uint32_t) (((nchar + 31) / 32 + 3) / 4);
  w->h.side_row_ints = (uint32_t) side_row_ints; w->h.tile_bytes = tile_bytes; w->h.ref_ambiguity = ref_ambiguity;
  w->h.off_planes = align64 (sizeof (struct uvdb_header));
  if (fwrite (&w->h, sizeof w->h, 1, w->f) != 1 || pad_to (w->f, w->h.off_planes)) { fclose (w->f); fclose (w->side_tmp); free (w); return NULL; }
  return w;
}

int
uvdb_add_reference (uvdb_writer w, const char *name, const char *seq)
{
  const uint64_t i = w->h.n_ref;
  if (i + 2 > w->idx_cap) {
    size_t ncap = w->idx_cap ? w->idx_cap * 2 : 4096;
    w->name_idx = (uint64_t *) realloc (w->name_idx, ncap * sizeof (uint64_t));
    w->exc_idx = (uint64_t *) realloc (w->exc_idx, ncap * sizeof (uint64_t));
    if (!w->name_idx || !w->exc_idx) return -1;
    w->idx_cap = ncap;
  }
  const size_t nl = strlen (name) + 1;
  if (w->names_len + nl > w->names_cap) {
    size_t ncap = w->names_cap ? w->names_cap * 2 : (1u << 20);
    while (ncap < w->names_len + nl) ncap *= 2;
    w->names = (char *) realloc (w->names, ncap);
    if (!w->names) return -1;
    w->names_cap = ncap;
  }
  w->name_idx[i] = w->names_len;
  memcpy (w->names + w->names_len, name, nl);
  w->names_len += nl;
  w->exc_idx[i] = w->exc_len;
  for (uint32_t s = 0; s < w->h.nchar; ) {           /* runs of invalid characters other than N */
    const char ch = seq[s];
    if (ch == '-' || ch == '?' || ch == 'X' || ch == 'O' || ch == '.') {
      uint32_t e = s + 1;
      while (e < w->h.nchar && seq[e] == ch && e - s < 0xFFFFFFu) e++;
      if (w->exc_len + 1 > w->exc_cap) {
        size_t ncap = w->exc_cap ? w->exc_cap * 2 : (1u << 16);
        w->exc = (uvdb_exc *) realloc (w->exc, ncap * sizeof (uvdb_exc));
        if (!w->exc) return -1;
        w->exc_cap = ncap;
      }
      w->exc[w->exc_len].pos = s; w->exc[w->exc_len].len_char = ((e - s) << 8) | (uint32_t) (unsigned char) ch;
      w->exc_len++;
      s = e;
    } else s++;
  }
  w->h.n_ref++;
  w->name_idx[w->h.n_ref] = w->names_len;
  w->exc_idx[w->h.n_ref] = w->exc_len;
  return 0;
}

int
uvdb_add_tiles (uvdb_writer w, size_t n_tiles, const void *planes, const int *non_n, const int *side_rows)
{
  if (fwrite (planes, w->h.tile_bytes, n_tiles, w->f) != n_tiles) return -1;
  const size_t n = n_tiles * 64;
  if ((w->tiles_written + n_tiles) * 64 > w->nonn_cap) {
    size_t ncap = w->nonn_cap ? w->nonn_cap * 2 : (1u << 16);
    while (ncap < (w->tiles_written + n_tiles) * 64) ncap *= 2;
    w->non_n = (int32_t *) realloc (w->non_n, ncap * sizeof (int32_t));
    if (!w->non_n) return -1;
    w->nonn_cap = ncap;
  }
  memcpy (w->non_n + w->tiles_written * 64, non_n, n * sizeof (int32_t));
  if (fwrite (side_rows, (size_t) w->h.side_row_ints * sizeof (int32_t), n, w->side_tmp) != n) return -1;
  w->tiles_written += n_tiles;
  return 0;
}

int
uvdb_close (uvdb_writer w)
{
  int bad = 0;
  struct uvdb_header *h = &w->h;
  h->n_tiles = w->tiles_written;
  if (h->n_tiles != (h->n_ref + 63) / 64) bad = 1;           /* every reference named must have been packed */
  uint64_t at = h->off_planes + h->n_tiles * h->tile_bytes;
  h->off_nonn = align64 (at);
  bad |= pad_to (w->f, h->off_nonn);
  bad |= h->n_tiles && fwrite (w->non_n, sizeof (int32_t), (size_t) h->n_tiles * 64, w->f) != (size_t) h->n_tiles * 64;
  at = h->off_nonn + h->n_tiles * 64 * sizeof (int32_t);
  h->off_side = align64 (at);
  bad |= pad_to (w->f, h->off_side);
  rewind (w->side_tmp);
  {
    char buf[1 << 16];
    size_t n;
    while ((n = fread (buf, 1, sizeof buf, w->side_tmp)) > 0) bad |= fwrite (buf, 1, n, w->f) != n;
  }
  at = h->off_side + h->n_tiles * 64 * (uint64_t) h->side_row_ints * sizeof (int32_t);
  uint64_t zero_idx[1] = {0};
  const uint64_t *nidx = h->n_ref ? w->name_idx : zero_idx, *eidx = h->n_ref ? w->exc_idx : zero_idx;
  h->off_name_idx = align64 (at);
  bad |= pad_to (w->f, h->off_name_idx);
  bad |= fwrite (nidx, sizeof (uint64_t), (size_t) h->n_ref + 1, w->f) != (size_t) h->n_ref + 1;
  at = h->off_name_idx + (h->n_ref + 1) * sizeof (uint64_t);
  h->off_names = align64 (at);
  bad |= pad_to (w->f, h->off_names);
  bad |= w->names_len && fwrite (w->names, 1, w->names_len, w->f) != w->names_len;
  at = h->off_names + w->names_len;
  h->off_exc_idx = align64 (at);
  bad |= pad_to (w->f, h->off_exc_idx);
  bad |= fwrite (eidx, sizeof (uint64_t), (size_t) h->n_ref + 1, w->f) != (size_t) h->n_ref + 1;
  at = h->off_exc_idx + (h->n_ref + 1) * sizeof (uint64_t);
  h->off_exc = align64 (at);
  bad |= pad_to (w->f, h->off_exc);
  bad |= w->exc_len && fwrite (w->exc, sizeof (uvdb_exc), w->exc_len, w->f) != w->exc_len;
  h->file_bytes = h->off_exc + w->exc_len * sizeof (uvdb_exc);
  bad |= fseek (w->f, 0, SEEK_SET) != 0 || fwrite (h, sizeof *h, 1, w->f) != 1;
  bad |= fclose (w->f) != 0;
  fclose (w->side_tmp);
  free (w->non_n); free (w->name_idx); free (w->names); free (w->exc_idx); free (w->exc);
  free (w);
  return bad ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------------ reader */
static void
set_err (char *errbuf, size_t errlen, const char *fmt, ...)
{
  if (!errbuf || !errlen) return;
  va_list ap;
  va_start (ap, fmt);
  vsnprintf (errbuf, errlen, fmt, ap);
  va_end (ap);
}

uvdb_reader
uvdb_open (const char *filename, char *errbuf, size_t errlen)
{
  uvdb_reader r = (uvdb_reader) calloc (1, sizeof *r);
  if (!r) return NULL;
  int fd = open (filename, O_RDONLY);
  struct stat st;
  if (fd < 0 || fstat (fd, &st) != 0) { set_err (errbuf, errlen, "cannot open %s", filename); if (fd >= 0) close (fd); free (r); return NULL; }
  if ((size_t) st.st_size < sizeof (struct uvdb_header)) { set_err (errbuf, errlen, "%s is not a packed uvaia database", filename); close (fd); free (r); return NULL; }
  r->map_len = (size_t) st.st_size;
  r->map = (const unsigned char *) mmap (NULL, r->map_len, PROT_READ, MAP_PRIVATE, fd, 0);
  close (fd);
  if (r->map == MAP_FAILED) { set_err (errbuf, errlen, "cannot map %s", filename); free (r); return NULL; }
  memcpy (&r->h, r->map, sizeof r->h);
  const struct uvdb_header *h = &r->h;
  if (memcmp (h->magic, UVDB_MAGIC, 8) != 0 || h->version != 1) {
    set_err (errbuf, errlen, "%s is not a packed uvaia database (version 1)", filename);
    uvdb_close_reader (r); return NULL;
  }
  const uint64_t n = h->n_ref, flen = (uint64_t) r->map_len;
  /* every section size is computed with overflow checks: a hostile header must not wrap a product into a small number */
  uint64_t sz_planes = 0, sz_nonn = 0, sz_side = 0, sz_idx = 0;
  int bad = h->file_bytes != flen || n > (UINT64_MAX >> 8) || h->n_tiles != (n + 63) / 64 || h->nchar == 0 ||
            h->W4 != ((h->nchar + 31) / 32 + 3) / 4 || h->tile_bytes != (uint64_t) h->W4 * 4 * 64 * 16 || h->side_row_ints != UVDB_SIDE_ROW_INTS;
  bad = bad || __builtin_mul_overflow (h->n_tiles, h->tile_bytes, &sz_planes) || __builtin_mul_overflow (h->n_tiles, (uint64_t) 64 * 4, &sz_nonn) ||
        __builtin_mul_overflow (h->n_tiles, (uint64_t) 64 * 4 * h->side_row_ints, &sz_side) || __builtin_mul_overflow (n + 1, (uint64_t) 8, &sz_idx);
  /* sections in file order, each inside the file and in front of the next one */
  bad = bad || h->off_planes < sizeof (struct uvdb_header) || h->off_planes > flen || sz_planes > flen - h->off_planes || h->off_planes + sz_planes > h->off_nonn ||
        h->off_nonn > flen || sz_nonn > flen - h->off_nonn || h->off_nonn + sz_nonn > h->off_side ||
        h->off_side > flen || sz_side > flen - h->off_side || h->off_side + sz_side > h->off_name_idx ||
        h->off_name_idx > flen || sz_idx > flen - h->off_name_idx || h->off_name_idx + sz_idx > h->off_names ||
        h->off_names > h->off_exc_idx || h->off_exc_idx > flen || sz_idx > flen - h->off_exc_idx || h->off_exc_idx + sz_idx > h->off_exc || h->off_exc > flen ||
        (h->off_planes | h->off_nonn | h->off_side | h->off_name_idx | h->off_exc_idx | h->off_exc) % 8 != 0;
  if (bad) {
    set_err (errbuf, errlen, "%s is truncated or inconsistent", filename);
    uvdb_close_reader (r); return NULL;
  }
  r->non_n = (const int32_t *) (r->map + h->off_nonn);
  r->name_idx = (const uint64_t *) (r->map + h->off_name_idx);
  r->names = (const char *) (r->map + h->off_names);
  r->exc_idx = (const uint64_t *) (r->map + h->off_exc_idx);
  r->exc = (const uvdb_exc *) (r->map + h->off_exc);
  {  /* the two index arrays: start at 0, never decrease, end inside their sections; every name ends in NUL inside the names section */
    const uint64_t names_len = h->off_exc_idx - h->off_names, exc_cap = (flen - h->off_exc) / sizeof (uvdb_exc);
    int ok = r->name_idx[0] == 0 && r->exc_idx[0] == 0 && r->name_idx[n] <= names_len && r->exc_idx[n] <= exc_cap &&
             h->off_exc + r->exc_idx[n] * sizeof (uvdb_exc) == flen;
    for (uint64_t i = 0; ok && i < n; i++)
      ok = r->name_idx[i] < r->name_idx[i + 1] && r->name_idx[i + 1] <= names_len && r->exc_idx[i] <= r->exc_idx[i + 1] && r->exc_idx[i + 1] <= exc_cap &&
           r->names[r->name_idx[i + 1] - 1] == '\0';
    if (!ok) {
      set_err (errbuf, errlen, "%s has inconsistent index sections", filename);
      uvdb_close_reader (r); return NULL;
    }
  }
  {  /* what goes to the device unchecked otherwise: valid-site counts within the alignment, side rows that list words of the alignment
      * (a count above the capacity only says "incomplete", as the engine writes it) */
    const uint32_t n_words = h->W4 * 4;
    const int32_t *side = (const int32_t *) (r->map + h->off_side);
    int ok = 1;
    for (uint64_t i = 0; ok && i < n; i++) {
      const int32_t *row = side + i * h->side_row_ints;
      ok = r->non_n[i] >= 0 && (uint32_t) r->non_n[i] <= h->nchar && row[0] >= 0;
      for (int k = 0; ok && k < UVDB_SIDE_LISTED && k < row[0]; k++) ok = row[1 + k] >= 0 && (uint32_t) row[1 + k] < n_words;
    }
    if (!ok) {
      set_err (errbuf, errlen, "%s holds valid-site counts or ambiguity rows outside the alignment", filename);
      uvdb_close_reader (r); return NULL;
    }
  }
  return r;
}

const char *
uvdb_name (uvdb_reader r, uint64_t i)
{
  return i < r->h.n_ref ? r->names + r->name_idx[i] : NULL;
}

const void *
uvdb_tile_planes (uvdb_reader r, uint64_t tile)
{
  return tile < r->h.n_tiles ? r->map + r->h.off_planes + tile * r->h.tile_bytes : NULL;
}

const int32_t *
uvdb_tile_side_rows (uvdb_reader r, uint64_t tile)
{
  return tile < r->h.n_tiles ? (const int32_t *) (r->map + r->h.off_side) + tile * 64 * r->h.side_row_ints : NULL;
}

void
uvdb_unpack_reference (uvdb_reader r, uint64_t i, char *out)
{
  /* IUPAC character of a set of bases (bit 0 = A, 1 = C, 2 = G, 3 = T); the empty set is 'N' unless an exception run says otherwise */
  static const char code[16] = {'N', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N'};
  const uint32_t *w = (const uint32_t *) uvdb_tile_planes (r, i / 64);
  const unsigned lane = (unsigned) (i & 63);
  const uint32_t nchar = r->h.nchar;
  for (uint32_t s = 0; s < nchar; s++) {
    const uint32_t word = s >> 5, w4 = word >> 2, j = word & 3, bit = s & 31;
    const size_t base = ((size_t) w4 * 4 * 64 + lane) * 4 + j;          /* plane 0 of this lane: 16-byte words, 64 lanes per plane */
    const unsigned a = (w[base] >> bit) & 1u, c = (w[base + 256] >> bit) & 1u, g = (w[base + 512] >> bit) & 1u, t = (w[base + 768] >> bit) & 1u;
    out[s] = code[a | (c << 1) | (g << 2) | (t << 3)];
  }
  out[nchar] = '\0';
  for (uint64_t e = r->exc_idx[i]; e < r->exc_idx[i + 1]; e++) {
    const uint32_t pos = r->exc[e].pos, len = r->exc[e].len_char >> 8;
    const char ch = (char) (r->exc[e].len_char & 0xFFu);
    for (uint32_t s = pos; s < pos + len && s < nchar; s++) out[s] = ch;
  }
}

void
uvdb_close_reader (uvdb_reader r)
{
  if (!r) return;
  if (r->map && r->map != MAP_FAILED) munmap ((void *) r->map, r->map_len);
  free (r);
}

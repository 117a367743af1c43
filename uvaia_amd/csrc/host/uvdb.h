/*
 * uvdb.h -- packed on-disk reference database.  Own format, no counterpart in the reference: it replaces,
 * for a database that is searched more than once, the serial text path readfasta_next() (src/fastaseq.c:422-474) + the slot
 * filling loop of src/nearest.c:251-286 (length check, -A filter, quick_count_sequence_non_N) by arrays the GPU engine takes
 * as they are (uvaia_gpu_db_append_packed, include/uvaia_gpu.h).
 *
 * File (little endian, sections 64-byte aligned, offsets from the start of the file):
 *   header   struct uvdb_header
 *   planes   n_tiles x tile_bytes    tiles of 64 references, [word group][plane A,C,G,T][lane] 16-byte words
 *   non_n    n_tiles*64 x int32      valid sites of every reference (what the -A filter and score 6 use)
 *   side     n_tiles*64 x side_row_ints x int32   partially ambiguous words of every reference
 *   name_idx (n_ref+1) x uint64      byte offsets into names
 *   names    NUL-terminated names, in stream order
 *   exc_idx  (n_ref+1) x uint64      record offsets into exc
 *   exc      (uint32 pos, uint32 len<<8 | char)   runs of invalid sites whose character is not 'N' ('-', '?', 'X', 'O', '.'):
 *            with them the exact (upper-case) text of a reference is recovered from its planes, as the .aln.xz dump needs
 * References that fail the -A filter or the length check are not stored: the filter value is recorded in the header.
 */
#ifndef UVAIA_HOST_UVDB_H
#define UVAIA_HOST_UVDB_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UVDB_MAGIC "UVAIADB1"
#define UVDB_SIDE_ROW_INTS 64   /* ints per side row: [0] count, [1..11] listed alignment words, [12 + 4k + p] plane p of the k-th listed word (uvaia_gpu_db_side_row_ints()) */
#define UVDB_SIDE_LISTED 11

struct uvdb_header {
  char magic[8];
  uint32_t version, nchar, W4, side_row_ints;
  uint64_t n_ref, n_tiles, tile_bytes;
  double ref_ambiguity;                 /* -A used while packing */
  uint64_t off_planes, off_nonn, off_side, off_name_idx, off_names, off_exc_idx, off_exc, file_bytes;
  uint64_t reserved[2];
};

typedef struct { uint32_t pos, len_char; } uvdb_exc;

/* ---- writer: tiles are appended in stream order, the index sections go to the end on close */
typedef struct uvdb_writer_struct *uvdb_writer;
uvdb_writer uvdb_create (const char *filename, int nchar, size_t tile_bytes, int side_row_ints, double ref_ambiguity);
/* text of the next reference: recorded for the name table and the exception runs (the planes come from the engine) */
int uvdb_add_reference (uvdb_writer w, const char *name, const char *seq);
/* the next n_tiles tiles in the engine's export form */
int uvdb_add_tiles (uvdb_writer w, size_t n_tiles, const void *planes, const int *non_n, const int *side_rows);
int uvdb_close (uvdb_writer w);       /* 0 on success */

/* ---- reader: the file is mapped read-only; every pointer below points into the mapping */
typedef struct uvdb_reader_struct {
  struct uvdb_header h;
  const unsigned char *map; size_t map_len;
  const uint64_t *name_idx; const char *names;
  const uint64_t *exc_idx; const uvdb_exc *exc;
  const int32_t *non_n;
} *uvdb_reader;
uvdb_reader uvdb_open (const char *filename, char *errbuf, size_t errlen);
const char *uvdb_name (uvdb_reader r, uint64_t i);
const void *uvdb_tile_planes (uvdb_reader r, uint64_t tile);        /* h.tile_bytes per tile, consecutive tiles are contiguous */
const int32_t *uvdb_tile_side_rows (uvdb_reader r, uint64_t tile);   /* 64 * h.side_row_ints ints per tile, contiguous */
/* exact upper-case text of reference i (nchar + 1 bytes) */
void uvdb_unpack_reference (uvdb_reader r, uint64_t i, char *out);
void uvdb_close_reader (uvdb_reader r);

#ifdef __cplusplus
}
#endif
#endif

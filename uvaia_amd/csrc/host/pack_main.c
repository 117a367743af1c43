/*
 * pack_main.c -- `uvaiapack`: aligned reference FASTA (raw/gz/xz/bz2, several files) -> packed database (uvdb.h).
 * No counterpart in the reference.  The filters of the reference's slot-filling loop that do not depend on
 * the queries are applied here once: the length check and the -A ambiguity filter (src/nearest.c:263-268).  The packing itself
 * runs on the GPU (uvaia_gpu_db_append + uvaia_gpu_db_export): the file holds exactly what the engine keeps resident.  Own code.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <libgen.h>

#include "cli_common.h"
#include "fastaseq.h"
#include "uvdb.h"
#include "../../../include/uvaia_gpu.h"

#define PACK_BATCH 4096      /* references per engine round trip (a multiple of 64) */

static void
usage (const char *prog)
{
  printf ("%s \n", UVAIA_PACKAGE_STRING);
  printf ("Packs an aligned reference FASTA into the bit-plane database the MI355X engine searches without parsing text.\n\n");
  printf (" %s [-hv] [-A <double>] [--device=<int>] -o <out.uvdb> <ref.fa(.gz,.xz)> [<ref.fa(.gz,.xz)>]...\n\n", prog);
  printf ("  -A, --ref_ambiguity=<double>     maximum allowed ambiguity for a REFERENCE sequence to be kept (default=0.5); `uvaia --packed` must use the same value\n");
  printf ("  -o, --output=<file>              packed database to write\n");
  printf ("  --device=<int>                   GPU to use (default: current device)\n");
}

int
main (int argc, char **argv)
{
  double ambig_r = 0.5;
  const char *out = NULL;
  int device = -1, ch, errors = 0;
  static const struct option longopts[] = {{"help", no_argument, 0, 'h'}, {"version", no_argument, 0, 'v'}, {"ref_ambiguity", required_argument, 0, 'A'},
    {"output", required_argument, 0, 'o'}, {"device", required_argument, 0, 1002}, {0, 0, 0, 0}};
  while ((ch = getopt_long (argc, argv, "hvA:o:", longopts, NULL)) != -1) switch (ch) {
    case 'h': usage (basename (argv[0])); return EXIT_SUCCESS;
    case 'v': printf ("%s\n", UVAIA_PACKAGE_VERSION); return EXIT_SUCCESS;
    case 'A': ambig_r = atof (optarg); break;
    case 'o': out = optarg; break;
    case 1002: device = atoi (optarg); break;
    default: errors++;
  }
  if (errors || !out || optind >= argc) { printf ("Error when reading arguments from command line:\n"); usage (basename (argv[0])); return EXIT_FAILURE; }
  if (ambig_r < 0.001) ambig_r = 0.001;
  if (ambig_r > 1.) ambig_r = 1.;
  int64_t time0[2];
  biomcmc_get_time (time0);

  uvaia_gpu_ctx *gpu = NULL;
  uvdb_writer w = NULL;
  char **seq = (char **) biomcmc_malloc (PACK_BATCH * sizeof (char *));
  int *non_n = (int *) biomcmc_malloc (PACK_BATCH * sizeof (int));
  void *planes = NULL; int *tile_nonn = NULL, *side = NULL;
  int nchar = 0, non_n_ref = 0, fill = 0;
  long count = 0, kept = 0, n_invalid = 0;

  for (int j = optind; j <= argc; j++) {
    readfasta_t rfas = j < argc ? new_readfasta (argv[j]) : NULL;
    for (;;) {
      const int have = rfas ? (readfasta_next (rfas) >= 0) : 0;
      if (have) {
        count++;
        if (!gpu) {       /* the first record fixes the alignment length; the engine needs some query to exist: a plain ACGT string */
          nchar = (int) rfas->seqlength;
          non_n_ref = (int) (nchar * (1. - ambig_r));
          char *dummy = (char *) biomcmc_malloc ((size_t) nchar + 1);
          for (int s = 0; s < nchar; s++) dummy[s] = "ACGT"[s & 3];
          dummy[nchar] = '\0';
          const char *one[1] = {dummy};
          uvaia_gpu_query q;
          memset (&q, 0, sizeof q);
          q.n_query = 1; q.nchar = nchar; q.seq = one; q.consensus = dummy;
          if (uvaia_gpu_open (&gpu, &q, 1, device, PACK_BATCH)) biomcmc_error ("%s", uvaia_gpu_last_error (NULL));
          free (dummy);
          if (uvaia_gpu_db_reserve (gpu, PACK_BATCH)) biomcmc_error ("%s", uvaia_gpu_last_error (gpu));
          const size_t tb = uvaia_gpu_db_tile_bytes (gpu);
          planes = biomcmc_malloc ((PACK_BATCH / 64) * tb);
          tile_nonn = (int *) biomcmc_malloc (PACK_BATCH * sizeof (int));
          side = (int *) biomcmc_malloc ((size_t) PACK_BATCH * (size_t) uvaia_gpu_db_side_row_ints () * sizeof (int));
          w = uvdb_create (out, nchar, tb, uvaia_gpu_db_side_row_ints (), ambig_r);
          if (!w) biomcmc_error ("cannot create %s", out);
        }
        /* as the reference's fill loop (src/nearest.c:263-278): low-quality records are dropped first, only then must the length fit */
        const int nn = quick_count_sequence_non_N (rfas->seq, rfas->seqlength);
        if (nn < non_n_ref) { n_invalid++; continue; }
        if (rfas->seqlength != (size_t) nchar) {
          biomcmc_warning ("Reference sequence '%s' has %zu sites but the first sequence has %d sites\n", rfas->name, rfas->seqlength, nchar);
          biomcmc_error ("all sequences must be aligned");
        }
        if (uvdb_add_reference (w, rfas->name, rfas->seq)) biomcmc_error ("out of memory while indexing %s", rfas->name);
        non_n[fill] = nn;
        seq[fill++] = rfas->seq; rfas->seq = NULL;
        kept++;
      }
      if (fill == PACK_BATCH || (!have && j == argc && fill)) {     /* a full batch, or the tail after the last file */
        const size_t nt = ((size_t) fill + 63) / 64;
        if (uvaia_gpu_db_append (gpu, (const char *const *) seq, non_n, fill) || uvaia_gpu_db_export (gpu, 0, nt, planes, tile_nonn, side) ||
            uvaia_gpu_db_clear (gpu)) biomcmc_error ("%s", uvaia_gpu_last_error (gpu));
        if (uvdb_add_tiles (w, nt, planes, tile_nonn, side)) biomcmc_error ("cannot write to %s", out);
        for (int c = 0; c < fill; c++) free (seq[c]);
        fill = 0;
      }
      if (!have) break;
    }
    if (rfas) {
      del_readfasta (rfas);
      fprintf (stderr, "Finished reading file %s in %.3lf secs; %ld sequences so far, %ld kept, %ld too ambiguous.\n", argv[j], biomcmc_update_elapsed_time (time0), count, kept, n_invalid);
    }
  }
  if (!w) biomcmc_error ("no sequence found");
  if (uvdb_close (w)) biomcmc_error ("problem writing %s", out);
  fprintf (stderr, "Packed %ld of %ld sequences (%d sites) into %s\n", kept, count, nchar, out);
  uvaia_gpu_close (gpu);
  free (seq); free (non_n); free (planes); free (tile_nonn); free (side);
  return EXIT_SUCCESS;
}

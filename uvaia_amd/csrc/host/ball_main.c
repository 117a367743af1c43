/*
 * ball_main.c -- `uvaiaball`: keeps the reference sequences within a distance radius of any query sequence.
 * Same options and output as the reference's src/ball.c; the per-batch loop (src/ball.c:248-251) runs on the GPU.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <libgen.h>
#include <omp.h>

#include "cli_common.h"
#include "gpu_glue.h"
#include "prepare.h"

int
main (int argc, char **argv)
{
  int help = 0, version = 0, acgt = 0, keep_resolved = 0, dist = 1, trim = 0, pool = 0, device = -1, n_ref = 0, errors = 0, ch;
  double ambig_q = 0.5, ambig_r = 0.5;
  const char *out = NULL, *qfile = NULL;
  const char **ref = (const char **) biomcmc_malloc ((size_t) argc * sizeof (char *));
  static const struct option longopts[] = {
    {"help", no_argument, 0, 'h'}, {"version", no_argument, 0, 'v'}, {"acgt", no_argument, 0, 1000}, {"keep_resolved", no_argument, 0, 'k'},
    {"distance", required_argument, 0, 'd'}, {"trim", required_argument, 0, 1001}, {"query_ambiguity", required_argument, 0, 'a'},
    {"ref_ambiguity", required_argument, 0, 'A'}, {"pool", required_argument, 0, 'p'}, {"reference", required_argument, 0, 'r'},
    {"nthreads", required_argument, 0, 't'}, {"output", required_argument, 0, 'o'}, {"device", required_argument, 0, 1002}, {0, 0, 0, 0}};
  while ((ch = getopt_long (argc, argv, "hvkd:a:A:p:r:t:o:", longopts, NULL)) != -1) switch (ch) {
    case 'h': help = 1; break;
    case 'v': version = 1; break;
    case 1000: acgt = 1; break;
    case 'k': keep_resolved = 1; break;
    case 'd': dist = atoi (optarg); break;
    case 1001: trim = atoi (optarg); break;
    case 'a': ambig_q = atof (optarg); break;
    case 'A': ambig_r = atof (optarg); break;
    case 'p': pool = atoi (optarg); break;
    case 'r': ref[n_ref++] = optarg; break;
    case 't': break;                                  /* host threads do not matter here */
    case 'o': out = optarg; break;
    case 1002: device = atoi (optarg); break;
    default: errors++;
  }
  if (optind < argc) qfile = argv[optind++];
  if (version) { printf ("%s\n", UVAIA_PACKAGE_VERSION); return EXIT_SUCCESS; }
  if (help || errors || !qfile || !n_ref) {
    printf ("%s \nSearch reference alignment for sequences within a distance radius of the query sequences (experimental).\n\n", UVAIA_PACKAGE_STRING);
    printf (" %s [-hvk] [--acgt] [-d <int>] [--trim=<int>] [-A <double>] [-a <double>] [-p <int>] -r <ref.fa(.gz,.xz)>... <seqs.fa(.gz,.xz)> [-o <without suffix>]\n",
            basename (argv[0]));
    return help ? EXIT_SUCCESS : EXIT_FAILURE;
  }
  if (ambig_q < 0.001) ambig_q = 0.001;
  if (ambig_q > 1.) ambig_q = 1.;
  if (ambig_r < 0.001) ambig_r = 0.001;
  if (ambig_r > 1.) ambig_r = 1.;
  int n_clust = omp_get_max_threads ();
  if (pool >= n_clust) n_clust = pool;                /* src/ball.c:161-166 */
  fprintf (stderr, "Experimental program: %s package: %s\n", basename (argv[0]), UVAIA_PACKAGE_STRING);
  fprintf (stderr, "Creating a queue of %d sequences; radius distance is %d (refs more distant than this are excluded)\n", n_clust, dist);

  size_t outlength = 0;
  char *outfilename = outfile_from_prefix (out ? out : "ball_uvaia", &outlength);
  int64_t time0[2];
  biomcmc_get_time (time0);
  alignment aln = read_fasta_alignment_from_file (qfile, 0xf);
  uvaia_set_prepare_device (device);
  query_t query = uvaia_prepare_query (aln, trim, dist, acgt, ambig_q, keep_resolved, 1);
  fprintf (stderr, "Query database now composed of %d valid references, after removing redundant (%s resolved) sequences.\n", query->aln->ntax, keep_resolved ? "less" : "more");
  if (query->aln->ntax < 1) biomcmc_error ("No valid reference sequences found. Please check file %s.", qfile);

  uvaia_gpu_ctx *gpu = NULL;
  if (uvaia_gpu_open_query (&gpu, query, 2, device, (size_t) n_clust)) biomcmc_error ("%s", uvaia_gpu_last_error (NULL));
  char **seq = (char **) biomcmc_malloc ((size_t) n_clust * sizeof (char *)), **name = (char **) biomcmc_malloc ((size_t) n_clust * sizeof (char *));
  int *mindist = (int *) biomcmc_malloc ((size_t) n_clust * sizeof (int));
  file_compress_t outstream = biomcmc_open_compress (outfilename, "w");
  const int non_n_ref = (int) (query->aln->nchar * ambig_r);   /* src/ball.c:201 (note: not 1-A as in uvaia) */
  int count = 0, n_invalid = 0, n_output = 0;

  for (int j = 0; j < n_ref; j++) {
    readfasta_t rfas = new_readfasta (ref[j]);
    bool end_of_file = false;
    while (!end_of_file) {
      int fill = 0;
      while (fill < n_clust && !end_of_file) {
        if (readfasta_next (rfas) < 0) { end_of_file = true; break; }
        count++;
        if (quick_count_sequence_non_N (rfas->seq, rfas->seqlength) < non_n_ref) { n_invalid++; continue; }
        if (rfas->seqlength != (size_t) query->aln->nchar) {
          biomcmc_warning ("Reference sequence '%s' has %zu sites but query sequences have %d sites\n", rfas->name, rfas->seqlength, query->aln->nchar);
          biomcmc_error ("all sequences must be aligned");
        }
        seq[fill] = rfas->seq; rfas->seq = NULL;
        name[fill] = rfas->name; rfas->name = NULL;
        fill++;
      }
      if (fill) {
        if (uvaia_gpu_ball (gpu, (const char *const *) seq, fill, query->dist + 1, mindist)) biomcmc_error ("%s", uvaia_gpu_last_error (gpu));
        for (int c = 0; c < fill; c++) {
          if (mindist[c] <= query->dist) { n_output++; write_fasta_record (outstream, name[c], seq[c]); }
          free (seq[c]); free (name[c]);
        }
      }
    }
    del_readfasta (rfas);
    fprintf (stderr, "Finished reading file %s in %.3lf secs; Total of %d sequences read, %d sequences within radius (kept), %d too ambiguous (excluded)\n",
             ref[j], biomcmc_update_elapsed_time (time0), count, n_output, n_invalid);
  }
  fprintf (stderr, "Saved %d sequences to file %s\n", n_output, outstream->filename);
  biomcmc_close_compress (outstream);
  uvaia_gpu_close (gpu);
  del_query_structure (query);
  free (seq); free (name); free (mindist); free (ref); free (outfilename);
  return EXIT_SUCCESS;
}

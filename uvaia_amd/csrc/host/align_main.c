/*
 * align_main.c -- `uvaialign`: aligns query sequences against one reference sequence and writes them on the reference's
 * columns.  Same options, filters, messages and output as the reference's src/align.c; the per-pool loop over align_query
 * (src/align.c:224-233,357-390) runs on the GPU through include/uvaia_align.h.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <inttypes.h>
#include <libgen.h>
#include <omp.h>

#include "cli_common.h"
#include "fastaseq.h"
#include "gpu_glue.h"
#include "../../../include/uvaia_align.h"

int
main (int argc, char **argv)
{
  int help = 0, version = 0, to_screen = 0, pool = 256 * omp_get_max_threads (), device = 0, errors = 0, n_fasta = 0, ch;   /* src/align.c:59-63 */
  int devices[64], n_devices = 0;
  double ambig = 0.5;
  const char *out = NULL, *ref_file = NULL;
  static const struct option longopts[] = {
    {"help", no_argument, 0, 'h'}, {"version", no_argument, 0, 'v'}, {"stdout", no_argument, 0, 1000}, {"ambiguity", required_argument, 0, 'a'},
    {"pool", required_argument, 0, 'p'}, {"reference", required_argument, 0, 'r'}, {"nthreads", required_argument, 0, 't'},
    {"output", required_argument, 0, 'o'}, {"device", required_argument, 0, 1001}, {"devices", required_argument, 0, 1002}, {0, 0, 0, 0}};
  while ((ch = getopt_long (argc, argv, "hva:p:r:t:o:", longopts, NULL)) != -1) switch (ch) {
    case 'h': help = 1; break;
    case 'v': version = 1; break;
    case 1000: to_screen = 1; break;
    case 'a': ambig = atof (optarg); break;
    case 'p': pool = atoi (optarg); break;
    case 'r': if (ref_file) errors++; ref_file = optarg; break;
    case 't': break;                                  /* the alignments run on the GPU: host threads do not matter */
    case 'o': out = optarg; break;
    case 1001: device = atoi (optarg); break;
    case 1002: n_devices = uvaia_parse_device_list (optarg, devices, 64); if (!n_devices) { fprintf (stderr, "--devices: expected a list such as 0-7 or 0,2,3\n"); exit (EXIT_FAILURE); } break;
    default: errors++;
  }
  const char **fasta = (const char **) argv + optind;
  n_fasta = argc - optind;
  if (version) { printf ("%s\n", UVAIA_PACKAGE_VERSION); return EXIT_SUCCESS; }
  if (help || errors || !ref_file || n_fasta < 1 || pool < 1) {
    printf ("%s \nAlign query sequences against a reference\nThe complete syntax is:\n\n", UVAIA_PACKAGE_STRING);
    printf (" %s [-hv] [--stdout] [-p <int>] [-t <int>] [-o <without suffix>] [-a <double>] -r <ref.fa|ref.fa.xz> <seqs.fa|seqs.fa.xz> [<seqs.fa|seqs.fa.xz>]...\n\n", basename (argv[0]));
    printf ("  -h, --help                       print a longer help and exit\n  -v, --version                    print version and exit\n");
    printf ("  --stdout                         print alignment to stdout (to redirect/pipe) instead of compress to file; much faster but may generate a big output\n");
    printf ("  -p, --pool=<int>                 How many query sequences are read in batch, to be aligned in parallel (defaults to 256 per thread)\n");
    printf ("  -t, --nthreads=<int>             accepted for compatibility (the alignments run on the GPU)\n");
    printf ("  -o, --output=<without suffix>    prefix of xzipped output alignment\n");
    printf ("  -a, --ambiguity=<double>         maximum allowed ambiguity for sequence to be excluded (default=0.5)\n");
    printf ("  -r, --reference=<ref.fa|ref.fa.xz> reference sequence in fasta format, possibly compressed with gz, xz, bz2\n");
    printf ("  <seqs.fa|seqs.fa.xz>             sequences to align in fasta format, possibly compressed with gz, xz, bz2 (can be multiple files)\n");
    printf ("  --device=<int>                   GPU to use (default 0)\n");
    printf ("  --devices=<list>                 several GPUs, e.g. 0-7 or 0,2,3: every pool of queries is cut among them (same rows as one GPU)\n");
    if (help) {
      printf ("Based on the wavefront algorithm (WFA, https://github.com/smarco/WFA), computed on the GPU.\n");
      printf ("Since the sequences are assumed to be similar, sequences too short or too big w.r.t. the reference are rejected.\n\n");
      printf ("The reference sequence and the unaligned fasta files can be compressed with gz, xz, bz2. The alignment output will be compressed with xz, ");
      printf ("unless you miss the tool. In this case the next available compression is tried (then the file extension might not correspond to it).\n\n");
      printf ("The command `pool` is the number of unaligned sequences read into memory at once (the higher the better, given your memory constraints).\n");
    }
    return (help && !errors) ? EXIT_SUCCESS : EXIT_FAILURE;
  }
  if (ambig < 0.001) ambig = 0.001;                   /* src/align.c:132-133 */
  if (ambig > 1.) ambig = 1.;
  int64_t time0[2], time1[2];
  biomcmc_get_time (time0);
  fprintf (stderr, "program: %s package: %s\n", basename (argv[0]), UVAIA_PACKAGE_STRING);

  size_t outlength = 0;
  char *outfilename = NULL;
  if (to_screen) fprintf (stderr, "Sequences will be shown uncompressed in screen (to redirect to file or pipe into another software).\n");
  else {
    char randname[32];
    if (!out) { sprintf (randname, "uvaia.%" PRIx64, (uint64_t) time0[1] & 0xffffff); out = randname; }     /* src/align.c:155-159 */
    outfilename = outfile_from_prefix (out, &outlength);
    fprintf (stderr, "Sequences will be compressed (if possible) and saved into file %s.\n", outfilename);
  }

  /* 1. the reference sequence (src/align.c:163-175): first record of the file */
  readfasta_t rfas = new_readfasta (ref_file);
  if (readfasta_next (rfas) < 1) biomcmc_error ("Error reading reference sequence %s", ref_file);
  char *refseq = rfas->seq; rfas->seq = NULL;
  const size_t aln_length = rfas->seqlength;
  del_readfasta (rfas);
  if (aln_length > 0x3fffffff) biomcmc_error ("reference sequence of %zu sites is too long", aln_length);
  /* alignments of different queries do not depend on each other: with several GPUs every pool is cut into contiguous shares, one
     aligner and one host thread per GPU (replicas of src/align.c's per-thread aligners, no exchange) */
  if (!n_devices) { n_devices = 1; devices[0] = device; }
  uvaia_aligner *gpu[64];
  for (int d = 0; d < n_devices; d++) if (uvaia_align_open (&gpu[d], refseq, (int) aln_length, devices[d], NULL)) biomcmc_error ("%s", uvaia_align_last_error (NULL));
  if (n_devices > 1) fprintf (stderr, "Batches of %d sequences will be read and aligned on %d GPUs.\n", pool, n_devices);
  else fprintf (stderr, "Batches of %d sequences will be read and aligned on GPU %d.\n", pool, devices[0]);

  file_compress_t outstream = to_screen ? NULL : biomcmc_open_compress (outfilename, "w");
  char **seq = (char **) biomcmc_malloc ((size_t) pool * sizeof (char *)), **name = (char **) biomcmc_malloc ((size_t) pool * sizeof (char *));
  int *len = (int *) biomcmc_malloc ((size_t) pool * sizeof (int));
  char *aln = (char *) biomcmc_malloc ((size_t) pool * (aln_length + 1));
  int count = 0, n_output = 0;
  const int print_interval = 5000;
  double result[3];

  biomcmc_get_time (time1);
  for (int j = 0; j < n_fasta; j++) {
    fprintf (stderr, "Started  reading file %s\n", fasta[j]);
    rfas = new_readfasta (fasta[j]);
    bool end_of_file = false;
    while (!end_of_file) {
      int fill = 0;
      while (fill < pool) {                           /* the serial slot-filling loop with its filters (src/align.c:189-221) */
        if (readfasta_next (rfas) < 0) { end_of_file = true; break; }
        if (!rfas->seq) continue;                     /* a header without sequence lines */
        count++;
        bool seq_valid = true;
        if (((3 * rfas->seqlength) < (2 * aln_length)) || ((2 * rfas->seqlength) > (3 * aln_length))) {
          fprintf (stderr, "Sequence %s has size too different from reference (%lu vs %lu)\n", rfas->name, (unsigned long) rfas->seqlength, (unsigned long) aln_length);
          seq_valid = false;
        }
        if (seq_valid) biomcmc_count_sequence_acgt (rfas->seq, rfas->seqlength, result);
        if (seq_valid && (result[2] > ambig)) {
          fprintf (stderr, "Sequence %s has proportion of N etc. (=%lf) above threshold of %lf\n", rfas->name, result[2], ambig);
          seq_valid = false;
        }
        if (seq_valid && (result[0] < 1. - 1.1 * ambig)) {
          fprintf (stderr, "Sequence %s has proportion of ACGT (=%lf) below threshold of %lf\n", rfas->name, result[0], 1. - 1.1 * ambig);
          seq_valid = false;
        }
        if (!seq_valid) continue;                     /* readfasta_next frees what it still owns */
        seq[fill] = rfas->seq; rfas->seq = NULL;
        name[fill] = rfas->name; rfas->name = NULL;
        len[fill] = (int) rfas->seqlength;
        fill++;
      }
      if (fill) {
        int failed = -1;
#pragma omp parallel for num_threads(n_devices) schedule(static, 1)
        for (int d = 0; d < n_devices; d++) {
          const int a = (int) ((long long) fill * d / n_devices), b = (int) ((long long) fill * (d + 1) / n_devices);
          if (b > a && uvaia_align_batch (gpu[d], (const char *const *) seq + a, len + a, b - a, aln + (size_t) a * (aln_length + 1), NULL)) {
#pragma omp critical
            failed = d;
          }
        }
        if (failed >= 0) {   /* what was aligned so far stays a complete file: close the stream before giving up */
          const int a = (int) ((long long) fill * failed / n_devices), b = (int) ((long long) fill * (failed + 1) / n_devices);
          if (!to_screen) biomcmc_close_compress (outstream);
          biomcmc_error ("%s (counted from sequence %s, the first of the %d handed to device %d; %d sequences were written before)",
                         uvaia_align_last_error (gpu[failed]), name[a], b - a, failed, n_output);
        }
        for (int c = 0; c < fill; c++) {
          const char *row = aln + (size_t) c * (aln_length + 1);
          n_output++;
          if (to_screen) printf (">%s\n%s\n", name[c], row);
          else write_fasta_record (outstream, name[c], row);
          free (seq[c]); free (name[c]);
        }
      }
      if ((count >= print_interval) && ((count % print_interval) < pool)) {
        fprintf (stderr, "%d\t sequences read, %d \t aligned. %.3lf secs elapsed.\n", count, n_output, biomcmc_update_elapsed_time (time1));
        fflush (stderr);
      }
    }
    del_readfasta (rfas);
    fprintf (stderr, "Finished reading file %s. In total %d sequences have been read.\n", fasta[j], count);
    fflush (stderr);
  }
  if (to_screen) fprintf (stderr, "Output %d aligned sequences. Total elapsed time: %.3lf secs\n", n_output, biomcmc_update_elapsed_time (time0));
  else {
    biomcmc_close_compress (outstream);
    fprintf (stderr, "Saved %d sequences to file %s\nTotal elapsed time: %.3lf secs\n", n_output, outfilename, biomcmc_update_elapsed_time (time0));
  }
  for (int d = 0; d < n_devices; d++) uvaia_align_close (gpu[d]);
  free (seq); free (name); free (len); free (aln); free (refseq); free (outfilename);
  return EXIT_SUCCESS;
}

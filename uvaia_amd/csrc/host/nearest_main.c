/*
 * nearest_main.c -- `uvaia`: for every query sequence, the closest neighbours in a (streamed) reference alignment.
 * Same options, progress messages, output files and table columns as the reference's src/nearest.c; the batch loop
 * (src/nearest.c:288-306) runs on the GPU through include/uvaia_gpu.h.  Own code.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <libgen.h>
#include <omp.h>

#include "cli_common.h"
#include "gpu_glue.h"
#include "prepare.h"
#include "uvdb.h"

typedef struct {
  int help, version, acgt, keep_resolved, exclude_self, nbest, trim, pool, threads, threads_given, device, devices[64], n_devices;
  double ambig_q, ambig_r;
  const char *out, *query, *packed;
  const char **ref; int n_ref;
} options;

static void
usage (const char *prog, int long_help)
{
  printf ("%s \n", UVAIA_PACKAGE_STRING);
  printf ("For every query sequence, finds closest neighbours in reference alignment. \n");
  printf ("The score-distance scan and the neighbour heaps run on an AMD MI355X GPU.\n\n");
  printf ("The complete syntax is:\n\n %s [-hvkx] [--acgt] [-n <int>] [--trim=<int>] [-A <double>] [-a <double>] [-p <int>] -r <ref.fa(.gz,.xz)> [-r <ref.fa(.gz,.xz)>]... <seqs.fa(.gz,.xz)> [-t <int>] [-o <without suffix>]\n\n", prog);
  printf ("  -h, --help                       print a longer help and exit\n");
  printf ("  -v, --version                    print version and exit\n");
  printf ("  --acgt                           considers only ACGT sites (i.e. unambiguous SNP differences) in query sequences (mismatch-based)\n");
  printf ("  -k, --keep_resolved              keep more resolved and exclude redundant query seqs (default is to keep all)\n");
  printf ("  -x, --exclude_self               Exclude reference sequences with same name as a query sequence\n");
  printf ("  -n, --nbest=<int>                number of best reference sequences per query to store (default=100)\n");
  printf ("  --trim=<int>                     number of sites to trim from both ends (default=0, suggested for sarscov2=230)\n");
  printf ("  -A, --ref_ambiguity=<double>     maximum allowed ambiguity for REFERENCE sequence to be excluded (default=0.5)\n");
  printf ("  -a, --query_ambiguity=<double>   maximum allowed ambiguity for QUERY sequence to be excluded (default=0.5)\n");
  printf ("  -p, --pool=<int>                 Pool size, i.e. how many reference seqs are sent to the GPU per batch (defaults to 64 per host thread; larger is faster)\n");
  printf ("  -r, --reference=<ref.fa(.gz,.xz)> aligned reference sequences (can be several files)\n");
  printf ("  --packed=<db.uvdb>               reference database packed by `uvaiapack` (instead of -r): loaded as it is, no text parsing\n");
  printf ("  <seqs.fa(.gz,.xz)>               aligned query sequences\n");
  printf ("  -t, --nthreads=<int>             suggested number of host threads (only sets the default pool size here)\n");
  printf ("  -o, --output=<without suffix>    prefix of xzipped output alignment and table with nearest neighbour sequences\n");
  printf ("  --device=<int>                   GPU to use (default: current device)\n");
  printf ("  --devices=<list>                 several GPUs, e.g. 0-7 or 0,2,3: every GPU scans its share of the references against all\n                                   queries and keeps the neighbours of its share of the queries (same results as one GPU)\n");
  if (long_help) {
    printf ("\nNeighbours are sorted in the same order as the table columns, using the next column to break ties:\n");
    printf (" 1. ACGT_matches -- considering only ACGT \n 2. text_matches -- exact matches, thus M-M is a match but M-A is not\n");
    printf (" 3. partial_matches -- M-A is considered a match since the partially ambiguous `M` equals {A,C}. The fully ambiguous `N` is neglected\n");
    printf (" 4. valid_pair_comparisons -- the `effective` sequence length for the comparison (sites without gaps or N in any of the two sequences)\n");
    printf (" 5. ACGT_matches_unique -- matches outside the sites where all queries agree\n");
    printf (" 6. valid_ref_sites -- if everything else is the same, then sequences with less gaps and Ns are preferred\n");
    printf ("With '--acgt' only ACGT is considered and the columns are ACGT_matches, valid_ACGT_comparisons, ACGT_matches_unique, valid_ref_sites,\n dist_consensus and dist_unique (their sum is the usual SNP distance).\n");
  }
}

static options
parse_options (int argc, char **argv)
{
  options o;
  memset (&o, 0, sizeof o);
  o.nbest = 100; o.ambig_q = o.ambig_r = 0.5; o.pool = 64 * omp_get_max_threads (); o.device = -1;
  o.ref = (const char **) biomcmc_malloc ((size_t) argc * sizeof (char *));
  static const struct option longopts[] = {
    {"help", no_argument, 0, 'h'}, {"version", no_argument, 0, 'v'}, {"acgt", no_argument, 0, 1000},
    {"keep_resolved", no_argument, 0, 'k'}, {"exclude_self", no_argument, 0, 'x'}, {"nbest", required_argument, 0, 'n'},
    {"trim", required_argument, 0, 1001}, {"query_ambiguity", required_argument, 0, 'a'}, {"ref_ambiguity", required_argument, 0, 'A'},
    {"pool", required_argument, 0, 'p'}, {"reference", required_argument, 0, 'r'}, {"nthreads", required_argument, 0, 't'},
    {"output", required_argument, 0, 'o'}, {"device", required_argument, 0, 1002}, {"packed", required_argument, 0, 1003}, {"devices", required_argument, 0, 1004}, {0, 0, 0, 0}};
  int ch, errors = 0;
  while ((ch = getopt_long (argc, argv, "hvkxn:a:A:p:r:t:o:", longopts, NULL)) != -1) switch (ch) {
    case 'h': o.help = 1; break;
    case 'v': o.version = 1; break;
    case 1000: o.acgt = 1; break;
    case 'k': o.keep_resolved = 1; break;
    case 'x': o.exclude_self = 1; break;
    case 'n': o.nbest = atoi (optarg); break;
    case 1001: o.trim = atoi (optarg); break;
    case 'a': o.ambig_q = atof (optarg); break;
    case 'A': o.ambig_r = atof (optarg); break;
    case 'p': o.pool = atoi (optarg); break;
    case 'r': o.ref[o.n_ref++] = optarg; break;
    case 't': o.threads = atoi (optarg); o.threads_given = 1; break;
    case 'o': o.out = optarg; break;
    case 1002: o.device = atoi (optarg); break;
    case 1004: o.n_devices = uvaia_parse_device_list (optarg, o.devices, 64); if (!o.n_devices) { fprintf (stderr, "--devices: expected a list such as 0-7 or 0,2,3\n"); exit (EXIT_FAILURE); } break;
    case 1003: o.packed = optarg; break;
    default: errors++;
  }
  if (optind < argc) o.query = argv[optind++];
  if (optind < argc) errors++;
  if (o.version) { printf ("%s\n", UVAIA_PACKAGE_VERSION); exit (EXIT_SUCCESS); }
  if (o.help) { usage (basename (argv[0]), 1); exit (EXIT_SUCCESS); }
  if (errors || !o.query || (!o.n_ref && !o.packed) || (o.n_ref && o.packed)) {
    printf ("Error when reading arguments from command line:\n");
    usage (basename (argv[0]), 0);
    exit (EXIT_FAILURE);
  }
  return o;
}

static void
save_distance_table (heap_t *heap, query_t query, const char *filename)
{
  file_compress_t xz = biomcmc_open_compress (filename, "w");
  const char *header = query->acgt
    ? "query,reference,rank,ACGT_matches,valid_ACGT_comparisons,ACGT_matches_unique,valid_ref_sites,dist_consensus,dist_unique\n"
    : "query,reference,rank,ACGT_matches,text_matches,partial_matches,valid_pair_comparisons,ACGT_matches_unique,valid_ref_sites\n";
  if (biomcmc_write_compress (xz, header) != (int) strlen (header)) biomcmc_warning ("problem saving header of compressed file %s;", xz->filename);
  int errors = 0;
  for (int i = 0; i < query->aln->ntax; i++) {
    heap_finalise_heap_qsort (heap[i]);
    /* the reference prints heap_size rows; a heap holding exactly heap_size-1 items makes it read an unused slot
       (src/min_heap.c:152-157, src/nearest.c:531-532): only the stored items are printed here */
    for (int j = 0; j < heap[i]->n; j++) {
      const q_item *it = &heap[i]->seq[j];
      size_t len = strlen (it->name ? it->name : "") + query->aln->taxlabel->nchars[i] + 96;
      char *line = (char *) biomcmc_malloc (len);
      int w = snprintf (line, len, "%s,%s,%d", query->aln->taxlabel->string[i], it->name ? it->name : "", j + 1);
      for (int k = 0; k < 6; k++) w += snprintf (line + w, len - (size_t) w, ",%d", it->score[k]);
      snprintf (line + w, len - (size_t) w, "\n");
      if (biomcmc_write_compress (xz, line) != (int) strlen (line)) errors++;
      free (line);
    }
  }
  if (errors) fprintf (stderr, "File %s may not have been correctly compressed, %d error%s occurred.\n", xz->filename, errors, errors > 1 ? "s" : "");
  biomcmc_close_compress (xz);
}

int
main (int argc, char **argv)
{
  int64_t time0[2], time1[2];
  biomcmc_get_time (time0);
  options o = parse_options (argc, argv);
  if (o.ambig_q < 0.001) o.ambig_q = 0.001;
  if (o.ambig_q > 1.) o.ambig_q = 1.;
  if (o.ambig_r < 0.001) o.ambig_r = 0.001;
  if (o.ambig_r > 1.) o.ambig_r = 1.;
  if (o.nbest < 1) o.nbest = 1;
  if (o.pool < 1) o.pool = 1;
  fprintf (stderr, "program: %s package: %s\n", basename (argv[0]), UVAIA_PACKAGE_STRING);
  if (o.threads_given) {
    int max_threads = omp_get_max_threads ();
    if (o.threads < 1 || o.threads > max_threads) o.threads = max_threads;
    omp_set_num_threads (o.threads);
  }
  fprintf (stderr, "Creating a queue of %d sequences; for each query, the %d closest sequences will be stored\n", o.pool, o.nbest);

  size_t outlength = 0;
  char *outfilename = outfile_from_prefix (o.out ? o.out : (o.acgt ? "nn_uvaia_acgt" : "nn_uvaia"), &outlength);

  /* 1. queries: read, quality filter, column classes, ordering, optional pruning */
  alignment aln = read_fasta_alignment_from_file (o.query, 0xf);
  fprintf (stderr, "Finished reading %d query references in %lf secs;\n", aln->ntax, biomcmc_update_elapsed_time (time0));
  uvaia_set_prepare_device (o.n_devices ? o.devices[0] : o.device);
  query_t query = uvaia_prepare_query (aln, o.trim, 1, o.acgt, o.ambig_q, o.keep_resolved, 0);
  fprintf (stderr, "Query database composed of %d valid references, after excluding low quality%s.\n", query->aln->ntax,
           o.keep_resolved ? " and redundant (less resolved) sequences" : "");
  if (query->aln->ntax < 1) biomcmc_error ("No valid reference sequences found. Please check file %s.", o.query);
  biomcmc_get_time (time1);
  if (query->acgt) fprintf (stderr, "Considering ACGT differences only (excluding all other characters). \n");
  else             fprintf (stderr, "Considering text match and partially ambiguous (excluding only gaps and Ns).\n");
  if (o.exclude_self) {
    fprintf (stderr, "Reference sequences with same name as query sequences will be excluded from the comparison.\n");
    query->aln->taxlabel_hash = new_hashtable (query->aln->ntax);
    for (int j = 0; j < query->aln->ntax; j++) insert_hashtable (query->aln->taxlabel_hash, query->aln->taxlabel->string[j], j);
  }

  /* 2. the engine and the host-side batch */
  /* one GPU (--device) or several (--devices): a group of one is a plain context */
  uvaia_gpu_group *grp = NULL;
  if (!o.n_devices) { o.n_devices = 1; o.devices[0] = o.device; }
  if (uvaia_gpu_group_open_query (&grp, query, o.nbest, o.devices, o.n_devices, (size_t) (o.n_devices > 1 && o.pool < 64 ? 64 : o.pool), 0)) biomcmc_error ("%s", o.n_devices > 1 ? uvaia_gpu_group_last_error (NULL) : uvaia_gpu_last_error (NULL));
  uvaia_gpu_ctx *gpu = uvaia_gpu_group_member (grp, 0);
  if (o.n_devices > 1) fprintf (stderr, "Using %d GPUs: references and queries are shared out among them.\n", o.n_devices);
  char **seq = (char **) biomcmc_malloc ((size_t) o.pool * sizeof (char *)), **name = (char **) biomcmc_malloc ((size_t) o.pool * sizeof (char *));
  int *non_n = (int *) biomcmc_malloc ((size_t) o.pool * sizeof (int));
  uint8_t *entered = (uint8_t *) biomcmc_malloc ((size_t) o.pool);
  name_table names = {NULL, 0};
  file_compress_t outstream = biomcmc_open_compress (outfilename, "w");
  const int non_n_ref = (int) (query->aln->nchar * (1. - o.ambig_r));
  const int print_interval = 10000;
  int count = 0, n_invalid = 0, same_name = 0, n_output = 0;
  int64_t ordinal = 0;

  fprintf (stderr, "\n Notice that the number of sites used in the comparisons (i.e. non-indel and non-N in at least one query) is %d, and the total alignment length is %d",
           query->n_idx + query->n_idx_c + query->n_idx_m, query->aln->nchar);
  fprintf (stderr, "\n The next step is main comparison, which may take a while\n\n");

  if (o.packed) {     /* resident search over a packed database: replaces the read/filter/fill loop below (src/nearest.c:251-286) */
    char msg[512];
    uvdb_reader db = uvdb_open (o.packed, msg, sizeof msg);
    if (!db) biomcmc_error ("%s", msg);
    if ((int) db->h.nchar != query->aln->nchar) biomcmc_error ("packed database %s has %u sites but query sequences have %d sites; all sequences must be aligned", o.packed, db->h.nchar, query->aln->nchar);
    if (db->h.ref_ambiguity != o.ambig_r) biomcmc_error ("packed database %s was filtered with -A %g: use the same value (its filter cannot be undone or tightened here)", o.packed, db->h.ref_ambiguity);
    if (db->h.side_row_ints != (uint32_t) uvaia_gpu_db_side_row_ints () || db->h.tile_bytes != uvaia_gpu_db_tile_bytes (gpu)) biomcmc_error ("packed database %s does not match this engine's tile layout", o.packed);
    const uint64_t n_all = db->h.n_ref, chunk_tiles = 256;
    /* -x: references named like a query leave the stream (src/nearest.c:257-262); the others move up, lane by lane */
    uint64_t *keep = NULL, n = n_all;
    if (o.exclude_self) {
      keep = (uint64_t *) biomcmc_malloc ((size_t) (n_all ? n_all : 1) * sizeof (uint64_t));
      n = 0;
      for (uint64_t i = 0; i < n_all; i++) { if (lookup_hashtable (query->aln->taxlabel_hash, (char *) uvdb_name (db, i)) > -1) same_name++; else keep[n++] = i; }
      if (n == n_all) { free (keep); keep = NULL; }
    }
    if (uvaia_gpu_group_db_reserve (grp, (size_t) (n ? n : 1))) biomcmc_error ("%s", uvaia_gpu_group_last_error (grp));
    if (!keep) {
      for (uint64_t t = 0; t < db->h.n_tiles; t += chunk_tiles) {
        const uint64_t nt = (db->h.n_tiles - t < chunk_tiles) ? db->h.n_tiles - t : chunk_tiles;
        const uint64_t first = t * 64, cnt = (first + nt * 64 > n) ? n - first : nt * 64;
        if (uvaia_gpu_group_db_append_packed (grp, uvdb_tile_planes (db, t), db->non_n + first, uvdb_tile_side_rows (db, t), (int) cnt)) biomcmc_error ("%s", uvaia_gpu_group_last_error (grp));
      }
    } else {
      const size_t tb = (size_t) db->h.tile_bytes, row = (size_t) db->h.side_row_ints, pieces = tb / (64 * 16);   /* 16-byte pieces per lane */
      unsigned char *planes = (unsigned char *) biomcmc_malloc (chunk_tiles * tb);
      int32_t *nn = (int32_t *) biomcmc_malloc (chunk_tiles * 64 * sizeof (int32_t)), *side = (int32_t *) biomcmc_malloc (chunk_tiles * 64 * row * sizeof (int32_t));
      for (uint64_t s0 = 0; s0 < n; s0 += chunk_tiles * 64) {
        const uint64_t cnt = (n - s0 < chunk_tiles * 64) ? n - s0 : chunk_tiles * 64;
        memset (planes, 0, chunk_tiles * tb); memset (nn, 0, chunk_tiles * 64 * sizeof (int32_t)); memset (side, 0, chunk_tiles * 64 * row * sizeof (int32_t));
#pragma omp parallel for schedule(static)
        for (uint64_t k = 0; k < cnt; k++) {
          const uint64_t r = keep[s0 + k];
          const unsigned char *src = (const unsigned char *) uvdb_tile_planes (db, r / 64) + (r % 64) * 16;
          unsigned char *dst = planes + (k / 64) * tb + (k % 64) * 16;
          for (size_t p = 0; p < pieces; p++) memcpy (dst + p * 1024, src + p * 1024, 16);
          nn[k] = db->non_n[r];
          memcpy (side + k * row, uvdb_tile_side_rows (db, r / 64) + (r % 64) * row, row * sizeof (int32_t));
        }
        if (uvaia_gpu_group_db_append_packed (grp, planes, nn, side, (int) cnt)) biomcmc_error ("%s", uvaia_gpu_group_last_error (grp));
      }
      free (planes); free (nn); free (side);
    }
    count = (int) n_all;
    fprintf (stderr, "Loaded %d packed sequences from %s in %.3lf secs;\n", (int) n, o.packed, biomcmc_update_elapsed_time (time0));
    uint8_t *ent = (uint8_t *) biomcmc_malloc ((size_t) (n ? n : 1));
    if (n && uvaia_gpu_group_search_resident (grp, (size_t) o.pool, 0, ent)) biomcmc_error ("%s", uvaia_gpu_group_last_error (grp));
    char *text = (char *) biomcmc_malloc ((size_t) query->aln->nchar + 1);
    for (uint64_t i = 0; i < n; i++) if (ent[i]) {     /* dump every sequence that entered some heap, in stream order */
      const uint64_t r = keep ? keep[i] : i;
      n_output++;
      uvdb_unpack_reference (db, r, text);
      write_fasta_record (outstream, uvdb_name (db, r), text);
      name_table_set (&names, (int64_t) i, uvdb_name (db, r));
    }
    free (text); free (ent); free (keep);
    fprintf (stderr, "Total of %d sequences searched; %d saved sequences include closest neighbours and intermediate. %.3lf secs elapsed. \n", count, n_output, biomcmc_update_elapsed_time (time1));
    if (o.exclude_self) fprintf (stderr, " %d reference sequences already present in query alignment (based on name only).\n", same_name);
    uvdb_close_reader (db);
  }
  for (int j = 0; j < o.n_ref; j++) {
    readfasta_t rfas = new_readfasta (o.ref[j]);
    bool end_of_file = false;
    while (!end_of_file) {
      int fill = 0;
      while (fill < o.pool && !end_of_file) {          /* the serial slot-filling loop of the reference (src/nearest.c:251-286) */
        if (readfasta_next (rfas) < 0) { end_of_file = true; break; }
        count++;
        if (o.exclude_self && lookup_hashtable (query->aln->taxlabel_hash, rfas->name) > -1) { same_name++; continue; }
        int nn = quick_count_sequence_non_N (rfas->seq, rfas->seqlength);
        if (nn < non_n_ref) { n_invalid++; continue; }
        if (rfas->seqlength != (size_t) query->aln->nchar) {
          biomcmc_warning ("Reference sequence '%s' has %zu sites but query sequences have %d sites\n", rfas->name, rfas->seqlength, query->aln->nchar);
          biomcmc_error ("all sequences must be aligned");
        }
        non_n[fill] = nn;
        seq[fill] = rfas->seq; rfas->seq = NULL;       /* steal the buffers, as the reference does */
        name[fill] = rfas->name; rfas->name = NULL;
        fill++;
      }
      if (fill) {
        if (uvaia_gpu_group_push (grp, (const char *const *) seq, non_n, fill, ordinal, entered)) biomcmc_error ("%s", uvaia_gpu_group_last_error (grp));
        for (int c = 0; c < fill; c++) if (entered[c]) {   /* dump every sequence that entered some heap, in stream order */
          n_output++;
          write_fasta_record (outstream, name[c], seq[c]);
          name_table_set (&names, ordinal + c, name[c]);
        }
        ordinal += fill;
        for (int c = 0; c < fill; c++) { free (seq[c]); free (name[c]); }
      }
      if (count >= print_interval && (count % print_interval) < o.pool) {
        int highest = 0;       /* cq->max_incompatible: the largest tolerance over all heaps (src/nearest.c:290-291,323-324) */
        for (int d = 0; d < o.n_devices; d++) {
          int v = 0, q0 = 0, q1 = query->aln->ntax;
          uvaia_gpu_ctx *cx = uvaia_gpu_group_member (grp, d);
          if (o.n_devices > 1) { uvaia_gpu_group_sync (grp); uvaia_gpu_group_query_shard (grp, d, &q0, &q1); if (q1 <= q0) continue; uvaia_gpu_set_active_queries (cx, q0, q1); }
          if (!uvaia_gpu_max_tolerance (cx, &v) && v > highest) highest = v;
          if (o.n_devices > 1) uvaia_gpu_set_active_queries (cx, 0, query->aln->ntax);
        }
        fprintf (stderr, "Total: %d sequences analysed, %d saved, %d poorly resolved. Highest number of ACGT mismatches = %d in current neighbourhood. %.3lf secs elapsed. ",
                 count, n_output, n_invalid, highest, biomcmc_update_elapsed_time (time1));
        if (o.exclude_self) fprintf (stderr, " %d already present in query alignment.\n", same_name); else fprintf (stderr, "\n");
      }
    }
    del_readfasta (rfas);
    fprintf (stderr, "Finished reading file %s in %.3lf secs;\n", o.ref[j], biomcmc_update_elapsed_time (time0));
    fprintf (stderr, "Total of %d sequences read; %d saved sequences include closest neighbours and intermediate, %d too ambiguous (excluded). %.3lf secs elapsed. \n",
             count, n_output, n_invalid, biomcmc_update_elapsed_time (time1));
    if (o.exclude_self) fprintf (stderr, " %d reference sequences already present in query alignment (based on name only).\n", same_name);
  }
  biomcmc_close_compress (outstream);
  fprintf (stderr, "Saved %d sequences to file %s , %.3lf secs elapsed.\n", n_output, outfilename, biomcmc_update_elapsed_time (time0));

  /* 3. heaps back to the host, table */
  heap_t *heap = (heap_t *) biomcmc_malloc ((size_t) query->aln->ntax * sizeof (heap_t));
  for (int i = 0; i < query->aln->ntax; i++) heap[i] = new_heap_t (o.nbest);
  if (uvaia_gpu_group_collect_heaps (grp, heap, name_table_get, &names)) biomcmc_error ("%s", uvaia_gpu_group_last_error (grp));
  strcpy (outfilename + outlength, ".csv.xz");
  save_distance_table (heap, query, outfilename);
  fprintf (stderr, "Saved distance table to file %s , %.3lf secs elapsed.\n", outfilename, biomcmc_update_elapsed_time (time0));

  for (int i = 0; i < query->aln->ntax; i++) del_heap_t (heap[i]);
  free (heap); free (seq); free (name); free (non_n); free (entered); free (o.ref);
  name_table_free (&names);
  uvaia_gpu_group_close (grp);
  del_query_structure (query);
  free (outfilename);
  return EXIT_SUCCESS;
}

/*
 * uvaia_gpu.h -- C ABI of the MI355X (gfx950) nearest-neighbour engine.
 *
 * This is the drop-in boundary for uvaia's hot path.  The reference (quadram-institute-bioscience/uvaia,
 * paths below are under /root/reference) has no plugin/FFI layer: the seam is the three OpenMP loops of
 * src/nearest.c:293-306 (and src/ball.c:248-251) plus the serial per-batch bookkeeping around them.  A
 * maintainer replaces those loops by calls to this library and keeps everything else (CLI, FASTA streaming,
 * heap_t/query_t host structures, writers).  See INTEGRATION.md for the patch.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a negative UVAIA_GPU_E* code and never
 *     calls exit(); uvaia_gpu_last_error() gives the message (the reference's biomcmc_error() = message + exit,
 *     src/nearest.c:208,277: the caller turns a code into that behaviour).
 *   - sequences are nchar upper-case bytes (what readfasta_next() src/fastaseq.c:422-474 and upper_kseq()
 *     src/utils.c:22 produce).  Alphabet: ACGT, IUPAC partial codes MRWSYKVHDB, and the invalid set
 *     N X - ? O . (src/utils.c:263).  Any other byte is refused with UVAIA_GPU_EALPHABET (its treatment by the
 *     absent biomcmc kernel is not pinned by anything in the reference).
 *   - a "batch" is one pool of the reference (src/nearest.c:249-251): the consensus pre-score of every sequence
 *     in it is truncated at the largest per-query tolerance at batch start (src/nearest.c:290-291,431-432), so
 *     batch boundaries are part of the semantics and are chosen by the caller, exactly as --pool is.
 *   - references are identified by a 64-bit ordinal supplied by the caller; names stay on the host
 *     (the reference strdup()s names into the heaps, src/min_heap.c:101,112).
 *   - one context = one GPU = one host thread at a time.
 */
#ifndef UVAIA_GPU_H
#define UVAIA_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UVAIA_GPU_NSCORE 6          /* q_item.score[6], src/min_heap.h:14-18 */

enum {
  UVAIA_GPU_OK         =  0,
  UVAIA_GPU_EINVAL     = -1,   /* bad argument */
  UVAIA_GPU_ENODEV     = -2,   /* no usable gfx950 device / HIP runtime error at start-up */
  UVAIA_GPU_ENOMEM     = -3,   /* device or host allocation failed */
  UVAIA_GPU_EHIP       = -4,   /* HIP runtime error (message in last_error) */
  UVAIA_GPU_EALPHABET  = -5,   /* a sequence holds a byte outside the supported alphabet */
  UVAIA_GPU_ESTATE     = -6    /* call sequence error (e.g. push larger than max_pool) */
};

typedef struct uvaia_gpu_ctx uvaia_gpu_ctx;

/* The prepared query set: the fields of struct query_struct (src/fastaseq.h:41-48) the hot path reads, after
 * create_query_indices()/reorder_query_structure() (src/fastaseq.c:732-795).  consensus[i] is 'N' where no query
 * is usable, '#' where queries disagree, else the shared character (src/fastaseq.c:742-756). */
typedef struct {
  int n_query;                 /* query->aln->ntax */
  int nchar;                   /* query->aln->nchar */
  const char *const *seq;      /* query->aln->character->string[i], nchar bytes each, upper-case */
  const char *consensus;       /* query->consensus, nchar bytes */
  const size_t *idx_c, *idx_m, *idx;   /* increasing site indices, as create_query_indices() builds them */
  int n_idx_c, n_idx_m, n_idx;
  size_t trim;                 /* query->trim */
  int acgt;                    /* query->acgt (--acgt) */
} uvaia_gpu_query;

/* Replaces new_queue() (src/nearest.c:367-390): per-query heaps of max(2,heap_size) slots (src/min_heap.c:58) with
 * max_incompatible = nchar (src/nearest.c:375,387), plus device buffers for batches of up to max_pool references.
 * device: HIP device index (-1 = current). */
int uvaia_gpu_open (uvaia_gpu_ctx **ctx, const uvaia_gpu_query *query, int heap_size, int device, size_t max_pool);

/* The same with tuning.  Every field 0 = the library's own choice; the values change speed, never results (the library reads no
 * environment variables). */
enum { UVAIA_GPU_SCAN_AUTO = 0,        /* by query count: packed planes up to 32 queries, column-compressed above */
       UVAIA_GPU_SCAN_PACKED = 1,      /* two-counter scan straight over the packed planes (nothing derived per query set) */
       UVAIA_GPU_SCAN_COMPRESSED = 2,  /* column-compressed scan over planes derived for the query set */
       UVAIA_GPU_SCAN_WIDE = 3 };      /* four 32-bit counters per pair (what alignments above 49 000 columns get) */
typedef struct {
  size_t subslice_refs;        /* resident search: references per scan launch (pools are cut into slices of about this length; >= 64) */
  int rare_max;                /* a polymorphic column counts as "rare" when all but at most this many queries carry the same base; -1 = no rare columns */
  int scan;                    /* UVAIA_GPU_SCAN_* */
  int serial;                  /* 1 = no overlap between the rebuild of the derived planes, the scans and the replays (isolated kernel timings) */
  int scan_tiles_per_wave;     /* column-compressed scan: 1, 2 or 4 tiles of 64 references per wave (default 2; 4 only with 8 waves per block) */
  int scan_waves_per_block;    /* column-compressed scan: 4 or 8 waves share a super-tile of 64 queries (default 8) */
  int rederive_streams;        /* uvaia_gpu_db_rederive: its chunks alternate over 1..3 streams (default 3: all chunks in flight at once, the first still done first) */
  int ball_gather;             /* radius search: the references' planes on the columns of query->idx are gathered 1 = by a pass of its own over the
                                  references that go on to the queries, 2 = by the consensus pass itself, for every reference (default 2) */
  int query_tables;            /* the scans' query-side tables (plane words, column classes, rare columns, compressed planes, item streams) are built
                                  1 = by host threads, 2 = on the device from the raw rows (default 2); same bytes either way */
  int replay_extras;           /* default mode: 0 = the library's choice (as 2 over the packed-plane scan, i.e. up to 32 queries; as 1 above), 2 = the scan also
                                  leaves text - ACGT and partial - text matches of every pair and the replay admits without a memory round trip (up to 128
                                  queries; over the column-compressed scan two kernels after the scan make them), 1 = the replay fetches them per admitted pair */
  int replay_cus;              /* compute units set aside for the replay's stream, the scans and the rebuild getting the others (CU masks): 0 = the
                                  library's choice (where replay_extras applies as many as hold one replay block per query at once, 8 up to 32 queries;
                                  none otherwise), -1 = none, n > 0 = n; 1000 * t + n also sets the replay's staging depth to t = 32, 16 or 8 tiles */
  int scan_streams;            /* resident search: consecutive slices' scans alternate over 1..3 streams (0 = the library's choice by launch size);
                                  100 + p (p = 1..98): a pool's first slice is p % of an equal share (default 70), 199: equal slices */
  int pipeline;                /* resident search over the column-compressed scan: 2 = a slice's replay runs next to its scan and follows its progress counters
                                  (slices then merge into long launches); 0 / 1 = the replay of a slice starts when its scan has ended (default: the
                                  thousand waiting replay waves cost the scan a block per CU, measured slower at config[1]) */
  int head_scan;               /* resident search over the column-compressed scan: 2 = the stream's first 128 references take the four-counter scan, so that the
                                  heaps fill without a memory round trip per admission; 0 / 1 = they go through the slices like the rest (default: the
                                  63 blocks of that scan run for 0.4 ms on their own, more than the hundred round trips they save) */
} uvaia_gpu_tuning;
/* Diagnostics: a copy of one of the query-side tables the scans read, as the open call left it on the device (tests compare the two ways
 * of building them).  which: 0 query plane words, 1 recoded planes (default mode), 2 ambiguity-word lists, 3 column classes, 4 rare-column
 * mask, 5 compressed polymorphic planes, 6 planes on the rare columns, 7 item streams, 8 their directory, 9 the rebuild's column split,
 * 10 = eleven ints { polymorphic columns, rare columns, their word groups (2), rare_max, scan choice, groups needing E / V / counts /
 * rare planes, replay_lq }.  *n_bytes = size of the table; copied only when it fits cap. */
int uvaia_gpu_export_query_table (uvaia_gpu_ctx *ctx, int which, void *out, size_t cap, size_t *n_bytes);
int uvaia_gpu_open_tuned (uvaia_gpu_ctx **ctx, const uvaia_gpu_query *query, int heap_size, int device, size_t max_pool, const uvaia_gpu_tuning *tuning /* may be NULL */);
void uvaia_gpu_close (uvaia_gpu_ctx *ctx);
const char *uvaia_gpu_last_error (const uvaia_gpu_ctx *ctx);   /* ctx may be NULL: error of the last failed open */

/* Replaces one turn of the batch loop, src/nearest.c:288-306: snapshot of max_incompatible, the consensus pre-score
 * (queue_distance_to_consensus, :428-433), the per-query gate + heap update over the batch in order
 * (queue_update_min_heaps{,_full,_acgt}, :435-510) and the OR over is_best (:303-306).
 *   seq[i]     n_ref sequences of nchar bytes, in stream order (cq->seq[c]); entries must be non-NULL
 *   non_n[i]   quick_count_sequence_non_N() of each (cq->non_n[c], src/nearest.c:263); NULL = count on the device
 *   ordinal0   ordinal of seq[0]; seq[i] gets ordinal0+i
 *   entered[i] out, 1 if the sequence entered the heap of any query during this batch (cq->is_best[n_query*c],
 *              src/nearest.c:310: such sequences are written to the .aln dump by the caller) */
int uvaia_gpu_push (uvaia_gpu_ctx *ctx, const char *const *seq, const int *non_n, int n_ref, int64_t ordinal0,
                    uint8_t *entered);

/* Reads the heaps back.  All arrays are caller-allocated:
 *   n_items[q]                      heap[q]->n
 *   max_incompatible[q]             heap[q]->max_incompatible
 *   scores[(q*(slots+1)+s)*6 + i]   heap[q]->seq[s].score[i], s = 1..n in the reference's binary-heap layout
 *   ordinals[q*(slots+1)+s]         ordinal of heap[q]->seq[s]
 * with slots = uvaia_gpu_heap_slots().  Slot 0 is unused, as in src/min_heap.c.  The caller then runs
 * heap_finalise_heap_qsort() (src/min_heap.c:149-158) on its own heap_t copies. */
int uvaia_gpu_drain (uvaia_gpu_ctx *ctx, int *n_items, int *max_incompatible, int *scores, int64_t *ordinals);
int uvaia_gpu_heap_slots (const uvaia_gpu_ctx *ctx);      /* max(2,heap_size) */
int uvaia_gpu_n_query (const uvaia_gpu_ctx *ctx);
/* Back to the state right after uvaia_gpu_open() (heaps empty); a resident database is kept. */
int uvaia_gpu_reset (uvaia_gpu_ctx *ctx);

/* ---- HBM-resident database (the measured configuration: references packed once, scanned many times) ----
 * Sequences are packed into bit-planes and appended to a device-resident database; they must already have passed
 * the caller's filters (src/nearest.c:255-278).  non_n as in uvaia_gpu_push (NULL = count on the device). */
int uvaia_gpu_db_reserve (uvaia_gpu_ctx *ctx, size_t n_ref_capacity);
int uvaia_gpu_db_append (uvaia_gpu_ctx *ctx, const char *const *seq, const int *non_n, int n_ref);
/* same, from one host block of n_ref rows of `pitch` bytes (pitch >= nchar) */
int uvaia_gpu_db_append_block (uvaia_gpu_ctx *ctx, const char *rows, size_t pitch, const int *non_n, int n_ref);
size_t uvaia_gpu_db_size (const uvaia_gpu_ctx *ctx);
/* Scans the resident database in batches of `pool` references (ordinals = position in the database + ordinal0),
 * i.e. the whole while-loop of src/nearest.c:249-330 for one reference file.  entered (may be NULL): db_size bytes.
 * Asynchronous with respect to the host unless entered != NULL; uvaia_gpu_sync() waits. */
int uvaia_gpu_search_resident (uvaia_gpu_ctx *ctx, size_t pool, int64_t ordinal0, uint8_t *entered);
int uvaia_gpu_sync (uvaia_gpu_ctx *ctx);

/* ---- ring mode for several GPUs (one context per GPU/process; the caller moves the state blob between ranks, e.g. with
 * RCCL send/recv).  A stripe = one batch of the reference (one pool) = the concatenation of one slice per rank, in rank
 * order; every rank holds its slices in its resident database.  Per stripe, on every rank:
 *     uvaia_gpu_slice_scan()                         (asynchronous; may be issued stripes ahead: two counter buffers)
 *     [rank > 0 or not the first stripe]  receive blob from the previous rank in the ring; uvaia_gpu_state_import()
 *     uvaia_gpu_slice_replay(stripe_start = (rank == 0))
 *     uvaia_gpu_state_export(); send blob to the next rank
 * The blob holds the batch snapshot of src/nearest.c:290-291 taken by rank 0, so truncation semantics are those of a
 * single process running the same stripes as pools.  After the last stripe the last rank holds the final heaps. */
size_t uvaia_gpu_state_bytes (const uvaia_gpu_ctx *ctx);
int uvaia_gpu_state_export (uvaia_gpu_ctx *ctx, void *dst);          /* device or host pointer; returns when dst is complete */
int uvaia_gpu_state_import (uvaia_gpu_ctx *ctx, const void *src);    /* src may be reused once this returns */
int uvaia_gpu_slice_scan (uvaia_gpu_ctx *ctx, size_t first, size_t n, int buf);      /* buf in [0, uvaia_gpu_slice_buffers()) */
int uvaia_gpu_slice_replay (uvaia_gpu_ctx *ctx, int buf, int64_t ordinal0, int stripe_start);
int uvaia_gpu_slice_buffers (void);     /* number of counter buffers: a scan may be issued that many slices ahead of its replay */
/* The per-query machines are independent, so the state can travel in several blobs, one per contiguous group of queries
 * [q0,q1): while one rank replays group j of a slice the next rank already replays group j-1 of its own slice.
 * take_snapshot: only on the rank that opens a stripe, once it holds the state of ALL queries (the snapshot is a maximum
 * over every query); it is a no-op for the results when the query set has no constant-and-complete column (n_idx_c == 0). */
size_t uvaia_gpu_state_range_bytes (const uvaia_gpu_ctx *ctx, int q0, int q1);
int uvaia_gpu_state_export_range (uvaia_gpu_ctx *ctx, void *dst, int q0, int q1);
int uvaia_gpu_state_import_range (uvaia_gpu_ctx *ctx, const void *src, int q0, int q1);
int uvaia_gpu_slice_replay_range (uvaia_gpu_ctx *ctx, int buf, int64_t ordinal0, int q0, int q1, int take_snapshot);
int uvaia_gpu_entered_flags (uvaia_gpu_ctx *ctx, uint8_t *out, int clear);   /* db_size bytes; see uvaia_gpu_search_resident */

/* ---- radius search: replaces the loop of src/ball.c:248-251 (seq_ball_against_query_structure,
 * src/fastaseq.c:660-696) for one batch.  radius = cq->dist + 1.  mindist[i] receives what the reference leaves in
 * cq->mindist[c]; the caller keeps sequence i iff mindist[i] <= radius-1 (src/ball.c:255). */
int uvaia_gpu_ball (uvaia_gpu_ctx *ctx, const char *const *seq, int n_ref, int radius, int *mindist);
/* the same for references [first, first+n) of the resident database.  Both stop where the reference stops: a reference whose
 * distance to the queries' consensus reaches the radius costs one pass over its packed planes, the queries are looked at only
 * for the references the reference's own loop would look at them for (twice the consensus distance >= radius). */
int uvaia_gpu_ball_resident (uvaia_gpu_ctx *ctx, size_t first, size_t n, int radius, int *mindist);
unsigned long long uvaia_gpu_ball_asked (uvaia_gpu_ctx *ctx, int reset);    /* references sent on to the queries since the last reset */
/* device time in ms since the last reset of the three kernels of the radius search: [0] the consensus pass, [1] the gather of the asked
   references' columns (with the read-back of their number), [2] the pair scan on the gathered tiles */
void uvaia_gpu_ball_kernel_ms (uvaia_gpu_ctx *ctx, double out[3], int reset);

/* Query preprocessing: the O(Q^2) test of exclude_redundant_query_sequences (src/fastaseq.c:797-841, the
 * call at :806-808).  For each of n_seq sequences (<= max_pool) and each query q of the open set,
 *   out[i * n_query + q] = 1  when quick_pairwise_score_truncated_idx_indelcheck (default; src/fastaseq.c:562-574) or
 *                             quick_pairwise_score_acgt (--acgt; :576-583) with maxdist 1 over query->idx would return 0,
 * i.e. no polymorphic column where both are valid (--acgt: both ACGT) and differ; 0 otherwise.  Handing in the query
 * sequences themselves gives the whole pair matrix; the host then walks it in the reference's order. */
int uvaia_gpu_agree_on_polymorphic (uvaia_gpu_ctx *ctx, const char *const *seq, int n_seq, uint8_t *out);

/* create_query_indices (src/fastaseq.c:732-777) on the device; no context needed (device < 0: the current one).  seq: n_query rows of
 * nchar bytes as the reference holds them at that point (upper-case).  Out, per column: consensus[i] as the reference defines it
 * ('N' outside the trimmed window or where no query is usable -- valid, with acgt != 0 ACGT --, '#' where two usable characters
 * differ, else the shared character) and some_missing[i] = 1 where some query is not usable (the reference's miss[], :744-756).
 * The caller builds idx / idx_m / idx_c from them exactly as src/fastaseq.c:763-770 does. */
int uvaia_gpu_query_columns (const char *const *seq, int n_query, int nchar, size_t trim, int acgt, int device, char *consensus, unsigned char *some_missing);

/* ---- introspection used by tests and bench.py ---- */
/* untruncated pair scores of the last batch: out[(i*n_query+q)*6 + s] = the score[] vector src/nearest.c:499-501
 * (or :464-469 with --acgt) assembles for (reference i of the batch, query q) when nothing is truncated. */
int uvaia_gpu_last_batch_scores (uvaia_gpu_ctx *ctx, int *out, int n_ref);
/* device time (ms) and launch count of the dominant kernel (the pair scan), measured with HIP events on the
 * context's stream since the last call with reset != 0; bytes = algorithmic bytes those launches covered. */
int uvaia_gpu_scan_stats (uvaia_gpu_ctx *ctx, double *ms, long long *launches, double *algorithmic_bytes, int reset);
/* counters of the ordered replay since the last reset: out[0] = admissions into heaps, out[1] = pairs whose remaining
 * counters were evaluated on demand, out[2] = of those, evaluated by a dense rescan (ambiguity lists overflowed) */
int uvaia_gpu_replay_stats (uvaia_gpu_ctx *ctx, unsigned long long out[3], int reset);
/* (query, tile of 64 references) pairs whose counters the replay of the packed-plane scan looked at since the last reset (up to 32 queries,
 * default mode: how sharp the per-tile bounds are) */
int uvaia_gpu_replay_tiles_opened (uvaia_gpu_ctx *ctx, unsigned long long *out, int reset);
/* Diagnostics of an engine built with -DREPLAY_TIMING (all zeros otherwise): wall-clock ticks (100 MHz) summed over the replay waves of the
 * packed-plane scan -- [0] waiting for staged counters, [1] requesting them, [2] inside opened tiles, [3] of that in admissions,
 * [4] late fetches, [5] their number, [6] whole waves, [7] their prologues (heap and tables into LDS), [8] waves; [9..11] unused.  (replay2_kernel:
 * [0] waits for on-demand words, [1] exact comparisons, [2] heap updates, [3] waits for a group's counters, [4] tolerance rises, [5] groups,
 * [6] whole waves, [7] the slowest wave.) */
int uvaia_gpu_replay_timing (uvaia_gpu_ctx *ctx, unsigned long long out[12], int reset);
/* tuning knob: queries held per pass of the packed-plane and four-counter scans (8, 16 or 32); 0 = default */
int uvaia_gpu_set_query_tile (uvaia_gpu_ctx *ctx, int qt);
/* ---- query shards: several GPUs, each holding the whole database and the heaps of a contiguous range of the queries.  The
 * per-query machines of src/nearest.c:435-510 are independent given the column classes of the WHOLE query set (which the context
 * was opened with), so a rank scans and replays only its range; no data-path exchange.  The one coupling between queries is the
 * batch snapshot cq->max_incompatible = max over ALL heaps (src/nearest.c:290-291), which matters only when the query set has
 * constant-and-complete columns (n_idx_c > 0): then the driver runs pool by pool, all-reduces (max) uvaia_gpu_max_tolerance()
 * over the ranks and passes the result as `snapshot`.
 *   set_active_queries: resident and slice calls act on queries [q0, q1) only (q0 a multiple of 64: the scan's super-tile of queries; any q0 under reference shards, where the range only selects tolerances); push/ball need the full range
 *   search_resident_pool: one batch [first, first+n) of the resident database, n <= max_pool; snapshot < 0 = take it from this
 *                         context's active queries */
int uvaia_gpu_set_active_queries (uvaia_gpu_ctx *ctx, int q0, int q1);
int uvaia_gpu_max_tolerance (uvaia_gpu_ctx *ctx, int *out);
int uvaia_gpu_search_resident_pool (uvaia_gpu_ctx *ctx, size_t first, size_t n, int64_t ordinal0, int snapshot);

/* ---- reference shards: several GPUs, each keeping, deriving and scanning 1/N of the references against ALL queries and replaying 1/N of
 * the queries over ALL references (the default layout for several GPUs; DESIGN.md "Multi-GPU").  The scores of a pair do not
 * depend on anything but the pair, so the scan -- more than nine tenths of a search -- shards by reference; the gate + heap machine
 * of a query is sequential in stream order (src/nearest.c:488,504-508) but independent of the other queries, so the replay shards
 * by query; in between, the pair counters of a slice move once: rank r sends to rank d the rows of d's queries (one all-to-all per
 * slice over RCCL between processes -- uvaia_amd/refshard.py -- or peer copies inside one process -- uvaia_gpu_group_*).
 *   - the stream is dealt in pieces of piece_refs references (a whole number of tiles of 64); piece p belongs to rank p % world
 *   - a rank KEEPS only its own pieces: packed planes, side rows, counts and the planes derived for the query set.  Appends are handed
 *     the whole stream in order (uvaia_gpu_db_append* keep the context's share and count the rest; uvaia_gpu_db_skip moves the
 *     stream position over references of other ranks without handing them over): every rank stages and packs 1/N of the database.
 *   - a replaying rank needs, of a piece scanned elsewhere: the rows of its queries (cnt, tmin), a small block per reference (`aux`:
 *     valid sites and, with constant-and-complete query columns, the untruncated consensus pre-score), and -- for the few pairs that
 *     reach the exact comparison -- words of the reference's packed planes and its side row, which it READS IN PLACE from the rank
 *     that keeps them: uvaia_gpu_shard_set_peer (one process: the other member's device pointers, peer access enabled) or
 *     uvaia_gpu_shard_ipc_handles / _open (one process per GPU: hipIpc mappings, over xGMI on a node)
 *   - the one coupling between queries, the batch snapshot cq->max_incompatible = max over ALL heaps (src/nearest.c:290-291), is
 *     exchanged per batch when it can matter (query sets with constant-and-complete columns): uvaia_gpu_max_tolerance on the
 *     active queries of every rank, maximum over the ranks, uvaia_gpu_set_snapshot.
 * set_shard: before the database is reserved.  shard_scan: references [first, first+n) inside ONE owned piece, n <= max_pool;
 *   cnt  uint32 [uvaia_gpu_shard_rows()][tiles * 64] pair counters as the scan kernels write them (first | second << 16: ACGT matches and
 *                                                    valid pairs, with --acgt mismatches and comparable sites), row = query, a reference
 *                                                    in column (position - 64 * (first / 64)); tiles = tiles the range touches
 *   tmin int2 [uvaia_gpu_shard_rows()][tiles]        the two bounds per (query, tile) the replay skips tiles by
 *   aux  uvaia_gpu_shard_aux_bytes(tiles) bytes      int valid_sites[tiles * 64], then (constant-and-complete columns) int4 prescore[tiles * 64]
 *   all caller-owned device buffers; asynchronous on the scan stream, uvaia_gpu_scan_wait() returns when they are complete.
 * shard_replay: gate + heaps of queries [q0, q1) over references [first, first+n) -- a piece of rank `owner` -- from buffers of the same
 *   layout that hold ONLY the rows q0 .. q1-1 and the piece's aux block; pieces must be replayed in stream order; asynchronous (uvaia_gpu_sync). */
int uvaia_gpu_db_set_shard (uvaia_gpu_ctx *ctx, int rank, int world, size_t piece_refs);
int uvaia_gpu_shard_rows (const uvaia_gpu_ctx *ctx);
size_t uvaia_gpu_shard_aux_bytes (const uvaia_gpu_ctx *ctx, size_t n_tiles);
int uvaia_gpu_shard_scan (uvaia_gpu_ctx *ctx, size_t first, size_t n, void *cnt, void *tmin, void *aux);
int uvaia_gpu_db_skip (uvaia_gpu_ctx *ctx, size_t n_ref);
int uvaia_gpu_shard_set_peer (uvaia_gpu_ctx *ctx, int rank, const void *planes, const void *side_rows);
const void *uvaia_gpu_shard_planes (const uvaia_gpu_ctx *ctx);
const void *uvaia_gpu_shard_side_rows (const uvaia_gpu_ctx *ctx);
int uvaia_gpu_shard_ipc_handle_bytes (void);
int uvaia_gpu_shard_ipc_handles (uvaia_gpu_ctx *ctx, void *out /* uvaia_gpu_shard_ipc_handle_bytes() bytes */);
int uvaia_gpu_shard_ipc_open (uvaia_gpu_ctx *ctx, int rank, const void *handles);
int uvaia_gpu_shard_ipc_close (uvaia_gpu_ctx *ctx);     /* unmaps them again: every rank closes, then a barrier, then the owners may free */
int uvaia_gpu_scan_wait (uvaia_gpu_ctx *ctx);
int uvaia_gpu_replay_wait (uvaia_gpu_ctx *ctx);      /* replays issued so far are complete: their counter buffers may be overwritten */
int uvaia_gpu_set_snapshot (uvaia_gpu_ctx *ctx, int snapshot);
/* Ordering against a stream the caller owns (a hipStream_t: the stream its collectives run on), by events, without the host: the
 * reference-shard driver queues scan -> exchange -> replay of stripe after stripe and never blocks.
 *   mark              remembers, under slot 0..7, the scans (or replays) issued so far
 *   stream_wait_mark  `stream` waits for that mark (the scan whose output the exchange sends; the replays that still read a buffer the
 *                     exchange is about to refill)
 *   wait_stream       the scans (or replays) issued from now on wait for what `stream` holds now (the exchange still reading the buffer a
 *                     scan overwrites; the exchange that delivers a replay's counters) */
enum { UVAIA_GPU_SCANS = 0, UVAIA_GPU_REPLAYS = 1 };
int uvaia_gpu_mark (uvaia_gpu_ctx *ctx, int what, int slot);
int uvaia_gpu_stream_wait_mark (uvaia_gpu_ctx *ctx, void *stream, int slot);
int uvaia_gpu_wait_stream (uvaia_gpu_ctx *ctx, int what, void *stream);
int uvaia_gpu_shard_replay (uvaia_gpu_ctx *ctx, const void *cnt, const void *tmin, const void *aux, int owner, size_t first, size_t n, int64_t ordinal0, int q0, int q1);

/* ---- a group of contexts driven by ONE host thread (the command line's --devices): the reference-shard protocol with peer copies
 * as the exchange; replaces the batch loop of src/nearest.c:245-330 for several GPUs.  devices[i] = HIP device of member i (a
 * device may be listed more than once); piece_refs = 0 picks a default.  Calls mirror the single-context ones: the database is
 * appended to every member, search_resident / push run the sharded search, drain collects every query's heap from the member
 * that replays it.  A group of one device is a plain context. */
typedef struct uvaia_gpu_group uvaia_gpu_group;
int  uvaia_gpu_group_open (uvaia_gpu_group **group, const uvaia_gpu_query *query, int heap_size, const int *devices, int n_devices, size_t max_pool, size_t piece_refs);
void uvaia_gpu_group_close (uvaia_gpu_group *group);
const char *uvaia_gpu_group_last_error (const uvaia_gpu_group *group);     /* group may be NULL: error of the last failed open */
int  uvaia_gpu_group_size (const uvaia_gpu_group *group);
uvaia_gpu_ctx *uvaia_gpu_group_member (uvaia_gpu_group *group, int i);
int  uvaia_gpu_group_query_shard (const uvaia_gpu_group *group, int i, int *q0, int *q1);
int  uvaia_gpu_group_db_reserve (uvaia_gpu_group *group, size_t n_ref_capacity);
int  uvaia_gpu_group_db_append (uvaia_gpu_group *group, const char *const *seq, const int *non_n, int n_ref);
int  uvaia_gpu_group_db_append_packed (uvaia_gpu_group *group, const void *planes, const int *non_n, const int *side_rows, int n_ref);
int  uvaia_gpu_group_db_clear (uvaia_gpu_group *group);
int  uvaia_gpu_group_db_rederive (uvaia_gpu_group *group);
size_t uvaia_gpu_group_db_size (const uvaia_gpu_group *group);
int  uvaia_gpu_group_reset (uvaia_gpu_group *group);
int  uvaia_gpu_group_search_resident (uvaia_gpu_group *group, size_t pool, int64_t ordinal0, uint8_t *entered);
int  uvaia_gpu_group_push (uvaia_gpu_group *group, const char *const *seq, const int *non_n, int n_ref, int64_t ordinal0, uint8_t *entered);
int  uvaia_gpu_group_drain (uvaia_gpu_group *group, int *n_items, int *max_incompatible, int *scores, int64_t *ordinals);
int  uvaia_gpu_group_sync (uvaia_gpu_group *group);

/* ---- packed interchange form (packed on-disk database).  Replaces, for a database that was packed once,
 * the serial text path of the reference (readfasta_next src/fastaseq.c:422-474 + the slot filling of src/nearest.c:251-286 +
 * quick_count_sequence_non_N src/fastaseq.c:642-648): tiles of 64 references, each uvaia_gpu_db_tile_bytes() long, laid out
 * [word group][plane A,C,G,T][lane] as 16-byte words (plane bit s of word w = site 32 w + s carries that IUPAC set bit), the
 * valid-site counts, and per reference a side row of uvaia_gpu_db_side_row_ints() ints (its partially ambiguous words).  The form
 * does not depend on the query set or on --acgt (an --acgt context re-codes the planes while importing).
 *   export:        from a default-mode (4-plane) context whose database was filled by uvaia_gpu_db_append*; arrays hold n_tiles*64
 *                  entries (lanes past the last reference are zero)
 *   append_packed: n_ref references = ceil(n_ref/64) tiles; the database must hold a whole number of tiles before the call */
size_t uvaia_gpu_db_tile_bytes (const uvaia_gpu_ctx *ctx);
int    uvaia_gpu_db_clear (uvaia_gpu_ctx *ctx);                      /* empties the resident database, keeps its capacity */
/* Rebuilds, for every resident reference, the planes the scan reads for the open query set (the appends build them for the rows
 * they add): the per-reference share of the work that depends on the query set (the reference does it implicitly, its loops
 * read the raw sequences through idx_c / idx_m / idx, src/nearest.c:428-510).  Asynchronous; later searches wait for it.  Call
 * between searches (after uvaia_gpu_sync), e.g. to time a search of a resident database together with this work. */
int    uvaia_gpu_db_rederive (uvaia_gpu_ctx *ctx);
int    uvaia_gpu_db_side_row_ints (void);
int    uvaia_gpu_db_export (uvaia_gpu_ctx *ctx, size_t first_tile, size_t n_tiles, void *planes, int *non_n, int *side_rows);
int    uvaia_gpu_db_append_packed (uvaia_gpu_ctx *ctx, const void *planes, const int *non_n, const int *side_rows, int n_ref);

/* bytes the pair scan reads per reference (the default scan reads planes derived from the packed record for this query set) */
size_t uvaia_gpu_scan_bytes_per_ref (const uvaia_gpu_ctx *ctx);
/* the pair scan of this context: 2 = column-compressed scan over planes derived for the query set (default above 32 queries),
 * 0 = two-counter scan straight over the packed planes (default up to 32 queries: nothing is derived), -1 = four-counter scan
 * (alignments above 49 000 columns) */
int uvaia_gpu_scan_variant (const uvaia_gpu_ctx *ctx);
/* bytes per reference uvaia_gpu_db_rederive writes (the planes that depend on the query set; the appends also write the
 * valid-site plane, which does not) */
size_t uvaia_gpu_derived_bytes_per_ref (const uvaia_gpu_ctx *ctx);
/* bytes per packed reference in HBM */
size_t uvaia_gpu_packed_bytes_per_ref (const uvaia_gpu_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif

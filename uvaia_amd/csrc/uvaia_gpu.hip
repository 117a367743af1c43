// uvaia_gpu.hip -- MI355X (gfx950 / CDNA4) engine behind include/uvaia_gpu.h.
//
// What it replaces in the reference (paths under /root/reference): the three OpenMP loops of
// src/nearest.c:293-306 -- consensus pre-score, per-query gate + heap update, is_best OR -- and the scoring
// kernels they call (src/fastaseq.c:585-596 and the absent biomcmc 4-count kernel, see oracle/uvaia_oracle.h).
//
// Design:
//   * sequences live in HBM as bit-planes, interleaved per tile of 64 references so that one lane owns one
//     reference: tile[t][w4][plane][lane] is a uint4 holding alignment words 4*w4..4*w4+3 (32 sites per word) of
//     that plane for reference 64*t+lane.  One wave-wide dwordx4 load = 1 KiB contiguous.
//   * the scan kernel keeps QT queries' accumulators in VGPRs; query words are wave-uniform and arrive through
//     scalar loads (SGPR operands of v_bitop3/v_and/v_xor), so the inner loop is pure VALU + v_bcnt with no
//     LDS traffic and no cross-lane reduction.  No MFMA: this is a popcount scan.
//   * the order-dependent gate/heap state machine of src/nearest.c:479-510 runs on the device, one wave per
//     query, over the dense pair counts of a batch, reproducing the reference's heap layout slot for slot.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/uvaia_gpu.h"

// ------------------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------------------
#define TT_A 0xF0u   // v_bitop3 truth-table columns: src0, src1, src2
#define TT_B 0xCCu
#define TT_C 0xAAu
#define B3(a, b, c, tt) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt) & 0xFFu)

static __device__ __forceinline__ uint32_t u4c(const uint4 &v, int j)
{ // component j of a uint4; folds to a register pick once j is a constant (no address is taken)
  return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
}

typedef uint32_t u32x4_ld __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ uint4 ld_stream(const uint4 *p)
{ // packed planes a kernel reads exactly once: a non-temporal load (the packed-plane scans went from 0.69 to 0.77 of the HBM peak with it)
  const u32x4_ld v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_ld *>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

static __device__ __forceinline__ int bcnt_acc(uint32_t x, int acc)
{ // v_bcnt_u32_b32 d, x, acc : popcount with free accumulate (keeps one VALU op per count and word)
  int r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}

namespace {

constexpr int HEAP_ENTRY = 8;          // 6 scores + 64-bit ordinal (lo, hi)
constexpr int AMB_CAP = 11;            // alignment words with a partially ambiguous site remembered per sequence
constexpr int AMB_STRIDE = AMB_CAP + 1;  // ints per QUERY: count (uncapped) + word indices
constexpr int AMB_ROW = 64;            // ints per REFERENCE side row (256 B, one coalesced wave load):
                                       //   [0] count (uncapped)  [1..11] word indices  [12 + 4k + p] plane p of the k-th listed word
constexpr int PACK_CHUNK = 4096;       // references per host->device staging round (multiple of 64)
constexpr uint32_t SCAN3_BIAS = 16384u; // scan3_kernel: bias of a pair's low counter half (what the rare items and, --acgt, the polymorphic columns may take away)
constexpr int SCAN_STRIPE_TILES = 64;  // pipelined search: tiles of 64 references per stripe of the scan's progress counters = one round of the replay's walk (a block of scan3_kernel
                                       // takes R <= 4 consecutive tiles starting at a multiple of R: it never straddles two stripes)
constexpr int NBUF = 4;                // counter buffers: the scan may run this many slices ahead of the gate/replay

thread_local std::string g_open_error;

struct ScanEvt { hipEvent_t a, b; double bytes; };

}  // namespace

struct uvaia_gpu_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t scan_streams[3] = {};       // extra scan streams: small launches (few active query tiles) overlap on up to three streams
  unsigned scan_rr = 0;
  int scan_nstreams = 1;                  // streams consecutive scans alternate over (set per search from the launch size)
  int scan_nstreams_forced = 0;           // tuning.scan_streams
  int first_slice_pct = 70;               // a pool's first slice is this share of an equal one (tuning.scan_streams = 100 + p sets p; 199 = equal slices)
  hipStream_t scan_stream = nullptr;      // ring mode: scans of later slices run here while the replay chain waits
  hipEvent_t scan_done[NBUF] = {}, replay_done[NBUF] = {};
  // uvaia_gpu_db_rederive: chunks of tiles rebuilt on their own stream; a scan waits for the chunks its slice touches
  struct DeriveChunk { long long t0, t1; hipEvent_t done; };
  hipStream_t derive_stream = nullptr;
  hipStream_t derive_streams[3] = {};     // [0] = derive_stream; the chunks of a rebuild alternate over the first derive_nstreams
  int derive_nstreams = 3; bool derive_forced = false;   // (forced: tuning.rederive_streams was given)
  std::vector<DeriveChunk> derive_chunks;
  hipEvent_t derive_fence[4] = {};
  size_t derive_pending = 0;          // chunks of the last rederive a scan may still have to wait for
  bool replay_recorded[NBUF] = {}, slice_scanned[NBUF] = {}, slice_cons_done[NBUF] = {};
  // pipelined search (column-compressed scan): the replay of a slice runs next to its scan and follows its progress counters
  // (off unless tuning.pipeline = 2: measured slower at config[1])
  bool pipeline = false, pipeline_now = false, slice_piped[NBUF] = {}, pipe_used = false;
  unsigned *d_progress[NBUF] = {}; size_t progress_cap[NBUF] = {};
  hipEvent_t scan_started[NBUF] = {};
  int *d_pipe_err = nullptr;
  size_t slice_cap[NBUF] = {};              // pairs each counter buffer holds (grown when a slice needs more: slices may exceed a pool, see plan_subslices)
  uint32_t *d_cntb[NBUF] = {};            // counter buffers 1..NBUF-1 (buffer 0 is d_cnt2), allocated on first use
  int2 *d_tmin[NBUF] = {};                // per (query, tile of 64 references): {smallest mismatch count, largest ACGT-match count}, one per counter buffer
  uint32_t *d_extb[NBUF] = {};            // packed-plane scan, default mode: per pair the other two counters (scan2_extras), one per counter buffer; sized like it
  uint32_t *d_rtpb[NBUF] = {};            // ... and per reference the consensus pre-score packed into one dword (query sets with constant-and-complete columns)
  uint4 *d_tb8[NBUF] = {};                // ... and per (query, tile of 64) the eight-entry bounds replay3_kernel walks (tile_bounds8)
  bool use_ext = false;                   // the scan leaves the extras and replay3_kernel runs (default mode: packed-plane scan, or the column-compressed one up to 128 queries)
  int4 *d_rtb[NBUF] = {};                 // per reference of a slice: untruncated consensus pre-score (query sets with constant-and-complete columns), one per counter buffer
  int slice_tiles[NBUF] = {}, slice_rb[NBUF] = {}, slice_re[NBUF] = {};
  long long slice_tf[NBUF] = {};
  size_t subslice = 25088;                // resident search: pools are cut into slices of about this size (exact: see search_resident).  (32 768 until round 4: at config[1]
                                          // four slices of 25 024 references instead of three of 33 334 cost 9 % more scan time -- a launch carries about 70 us of ramp and
                                          // tail -- and still end 4 % sooner: the first replay starts earlier, the last one is shorter)
  bool subslice_forced = false;           // the length was given (tests): taken as it is
  int nq = 0, nq_pad = 0, nchar = 0, W = 0, W4 = 0, P = 4, NQ = 6, acgt = 0, k = 2, qt = 16, n_idx_c = 0, n_idx_m = 0;
  size_t trim = 0;
  size_t max_pool = 0, pool_pad = 0;
  // query side
  uint32_t *d_qp = nullptr;      // [nq_pad][W4][4][NQ]   full-information query planes
  uint32_t *d_qp2 = nullptr;     // [nq_pad][W4][4][4]    (lo, hi, isACGT, valid) for the two-counter scan (default mode)
  int scan_variant = 2;          // 2 = column-compressed scan3_kernel (default above 16 queries); 0 = scan2_*_kernel over the packed planes
  // column-compressed scan: classes of the alignment columns for this query set, compressed/dirty query planes, derived reference planes
  uint32_t *d_cls = nullptr;     // [W4*4][4]  cL, cH, constMask, polyMask
  uint32_t *d_qpl = nullptr;     // [nq_pad][NP4][L,H,I,-][4]   compressed polymorphic columns of the queries
  uint32_t *d_stream = nullptr;  // per query tile: the dirty-word item stream of scan3_kernel (layout: see the kernel)
  uint32_t *d_sdir = nullptr;    // [nq_pad/64][16] per super-tile and wave: {first dword, number} of its group records and of its rare records
  int NP = 0, NP4 = 0;           // polymorphic columns counted densely
  int NR = 0, NR4 = 0, rare_max = -1;   // "rare" columns: all but <= rare_max queries carry the same base; sparse (items), groups follow the dense ones
  uint32_t *d_rmask = nullptr;   // [W4*4] mask of the rare columns
  int *d_split = nullptr;        // derive_all_kernel: w4 range and first gathered bit of each of its four waves
  // reference shards (uvaia_gpu_db_set_shard): the stream is dealt in pieces of shard_pt tiles, piece p belongs to rank p % world; the packed
  // planes of ALL references are resident (the replay reads them), the planes derived for the query set only for the owned pieces
  int shard_rank = 0, shard_world = 1; long long shard_pt = 0;
  const uint4 *peer_db[64] = {}; const int *peer_amb[64] = {};      // packed planes and side rows of every rank's pieces, as this process can address them
  void *ipc_opened[64][2] = {};                                    // mappings opened by uvaia_gpu_shard_ipc_open (closed with the context)
  uint32_t *d_qrare = nullptr;   // [nq][NR4*4][lo, hi, isACGT] the queries on the rare columns (--acgt: dist_unique of admitted pairs)
  int need_e_groups = 0, need_v_groups = 0, need_g_groups = 0, need_r_groups = 0;   // word groups whose E / V plane some query tile has to read (for the byte accounting)
  int act_q0 = 0, act_q1 = 0;    // active query range of the resident/slice paths (query shards across GPUs); whole set by default
  bool serial = false;           // tuning.serial: no scan/replay overlap (to time the kernels in isolation)
  int replay_lq = -1;            // replay caches the query's planes in LDS (22 KB per block): -1 = only with few queries (see open)
  int replay_prio = 1;           // replay waves raise their issue priority
  bool head_full = false;        // tuning.head_scan = 2: the stream's first two tiles go through the four-counter scan (heaps fill without on-demand fetches; measured slower)
  int replay_half = 32;          // tiles per staging buffer of replay3_kernel (32, 16 or 8: its LDS decides how many of its blocks share a compute unit)
  int replay_cus = 0;            // compute units set aside for the replay kernels of the resident search (0: none, the streams share the chip by priority)
  hipStream_t rep_stream = nullptr; hipEvent_t rep_ev[2] = {};   // ... the stream masked to them, and the events that splice its kernels into `stream`'s order
  int scan_R = 2;                // reference tiles per wave of scan3_kernel (the item stream is built for it)
  int scan_NW = 8;               // waves per block of scan3_kernel = shares a super-tile's records are cut into
  uint4 *d_batch_ev = nullptr, *d_batch_poly = nullptr, *d_db_ev = nullptr, *d_db_poly = nullptr;
  uint32_t *d_batch_grp = nullptr, *d_db_grp = nullptr;   // [tile][W4][64]  popc(E) | popc(V) << 16 of each word group (for queries that are all-N there)
  int *d_batch_tote = nullptr, *d_db_tote = nullptr;
  int *d_amb_q = nullptr;        // [nq][AMB_STRIDE] ambiguity-word lists of the queries
  struct { const void *p; size_t n; } qtab[10] = {};   // the query-side tables as uvaia_gpu_export_query_table numbers them (device pointer, bytes)
  int *d_batch_amb = nullptr, *d_db_amb = nullptr;   // same for the references of the batch buffer / database
  int *d_batch_tot = nullptr, *d_db_tot = nullptr;   // per reference: valid sites (default) / ACGT sites (--acgt), counted by pack_refs_kernel
  uint32_t *d_cnt2 = nullptr;    // [nq_pad][pool_pad] two-counter scan output, one dword per pair: first | second << 16 
  unsigned long long *d_stats = nullptr;             // admissions, on-demand evaluations, dense fallbacks, tiles opened (replay3_kernel)
  hipEvent_t order_ev[16] = {}; unsigned order_rr = 0;   // uvaia_gpu_wait_stream: ordering against a caller-owned stream
  hipEvent_t mark_ev[8][3] = {}; bool mark_set[8][3] = {}; // uvaia_gpu_mark
  bool fullscan = false;         // four-counter scan + the replay over it (alignments above 49 000 columns; tuning.scan = UVAIA_GPU_SCAN_WIDE)
  size_t cnt_cap = 0;            // int4 elements allocated in d_cnt (lazily)
  uint32_t *d_cp = nullptr;      // consensus restricted to idx_c, one row [W4][4][NQ]
  uint32_t *d_cpm = nullptr;     // consensus restricted to idx_m (radius search)
  uint32_t *d_qpoly = nullptr;   // queries restricted to idx (radius search, redundancy test), [nq_pad][W4][4][NQ]: d_qp masked by d_pmask, built by the first call that needs it
  uint32_t *d_pmask = nullptr;   // [W4*4] mask of the polymorphic query columns (query->idx)
  int *d_mindist = nullptr, *d_ball_list = nullptr, *d_ball_cdist = nullptr, *d_ball_n = nullptr; size_t ball_cap = 0;   // radius search: results, the references that go on to the queries
  uint4 *d_ball_tiles = nullptr; size_t ball_tiles_cap = 0; unsigned long long ball_asked = 0;
  bool ball_fused = true; uint4 *d_ball_ga = nullptr; size_t ball_ga_tiles = 0;   // stage 1 gathers every reference's columns of query->idx itself (tuning.ball_gather)
  hipEvent_t ball_ev[4] = {}; double ball_ms[3] = {0., 0., 0.};   // per-kernel time of the radius search (host_ball.inc)
  int *d_idx_cols = nullptr; int n_idx = 0, NG4 = 0;       // query->idx (the polymorphic query columns) and the word groups they fill once gathered
  std::vector<int> idx_cols; uint32_t *d_ball_masks = nullptr; int NH4 = 0;   // their order in the gathered words: masks [W4][hot 4 | others 4], hot word groups (ensure_qgather)
  uint32_t *d_qg = nullptr;                                 // the queries on those columns (kernels_ball.inc), built by the first radius search
  unsigned long long *d_ball_key = nullptr;                 // per listed reference: first query that ends the reference's loop (query << 32 | distance)
  // heaps / state
  int *d_heap = nullptr, *d_n = nullptr, *d_T = nullptr, *d_snap = nullptr, *d_err = nullptr;
  // batch buffers
  uint4 *d_batch = nullptr;      // packed tiles of the current batch
  int *d_batch_nonn = nullptr;
  int4 *d_cnt = nullptr;         // [nq_pad][pool_pad]
  int4 *d_rt = nullptr, *d_tr = nullptr;   // [pool_pad]
  uint8_t *d_entered = nullptr;  // [pool_pad] (push) or [db_cap] (resident)
  size_t entered_cap = 0;
  uint8_t *d_stage = nullptr;    // device staging for raw characters (2 x PACK_CHUNK rows)
  uint8_t *h_stage = nullptr;    // pinned host staging (2 x PACK_CHUNK rows)
  hipEvent_t stage_free[2] = {}; bool stage_busy[2] = {};
  size_t pitch = 0;
  // resident database
  uint4 *d_db = nullptr;
  int *d_db_nonn = nullptr;
  size_t db_cap = 0, db_n = 0, db_local_tiles = 0;   // (db_n counts the stream; a context of a reference shard keeps db_local_tiles tiles of it)
  // last batch (introspection)
  const uint4 *last_tiles = nullptr; const int *last_nonn = nullptr; int last_n = 0, last_rbegin = 0, last_ppad = 0, last_ntiles = 0;
  long long last_tile_first = 0;
  const int4 *last_rt = nullptr;
  // stats
  std::vector<ScanEvt> evts;
  std::vector<hipEvent_t> ev_pool;      // timing events of earlier launches, reused (creating and destroying a pair per launch was 50-100 us of host time per step)
  double scan_ms = 0, scan_bytes = 0; long long scan_launches = 0;
  bool profile = true;
  std::string err;
};

namespace {

int fail(uvaia_gpu_ctx *c, int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (c) c->err = buf; else g_open_error = buf;
  return code;
}

// One check for every HIP call; what happens on failure is the caller's hook (the code maps out-of-memory to UVAIA_GPU_ENOMEM).
#define HIP_TRY(call, on_fail) do { hipError_t e_ = (call); if (e_ != hipSuccess) { const int code_ = e_ == hipErrorOutOfMemory ? UVAIA_GPU_ENOMEM : UVAIA_GPU_EHIP; (void)code_; on_fail; } } while (0)
// inside an entry point that has a context: message into the context, return the code
#define HIPCHK(c, call) HIP_TRY(call, return fail((c), code_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__))

// Reference shards: the per-reference arrays of the resident database -- packed planes, side rows, counts, derived planes -- hold the
// context's OWN pieces only, numbered densely ("local" tiles: dtile_of); positions in the stream, ordinals and the dump flags stay
// global.  owns_tile: does this context keep (global) tile t;  dtile_of: its local number on the context that keeps it (the same formula
// on every rank: a replaying rank uses it to find a reference in the memory of the rank that scanned it).
inline bool owns_tile(const uvaia_gpu_ctx *c, long long t) { return c->shard_world == 1 || (t / c->shard_pt) % c->shard_world == c->shard_rank; }
inline long long dtile_of(const uvaia_gpu_ctx *c, long long t)
{ return c->shard_world == 1 ? t : (t / (c->shard_pt * c->shard_world)) * c->shard_pt + t % c->shard_pt; }
// the owned parts of the global tiles [gt0, gt1): f(global first tile, local first tile, number of tiles) per part inside one piece
template <class F> inline int for_owned_tiles(const uvaia_gpu_ctx *c, long long gt0, long long gt1, F f)
{
  if (c->shard_world == 1) return gt1 > gt0 ? f(gt0, gt0, gt1 - gt0) : 0;
  for (long long a = gt0; a < gt1;) {
    const long long b = std::min(gt1, (a / c->shard_pt + 1) * c->shard_pt);
    if (owns_tile(c, a)) { const int rc = f(a, dtile_of(c, a), b - a); if (rc) return rc; }
    a = b;
  }
  return 0;
}
inline size_t derived_tiles(const uvaia_gpu_ctx *c, size_t tiles)
{ return c->shard_world == 1 ? tiles : (size_t)((tiles + (size_t)(c->shard_pt * c->shard_world) - 1) / (size_t)(c->shard_pt * c->shard_world)) * (size_t)c->shard_pt; }

// IUPAC code table: 1..15 = nucleotide set (A=1 C=2 G=4 T=8), 0 = invalid site (N X - ? O .), 0xFF = refused
void fill_code_table(uint8_t *t)
{
  memset(t, 0xFF, 256);
  const char *inv = "NnXx-?Oo.";                      // src/utils.c:263
  for (const char *p = inv; *p; p++) t[(unsigned char)*p] = 0;
  static const struct { char c; uint8_t m; } iu[] = {
    {'A',1},{'C',2},{'G',4},{'T',8},{'M',3},{'R',5},{'W',9},{'S',6},{'Y',10},{'K',12},{'V',7},{'H',11},{'D',13},{'B',14}};
  for (auto &e : iu) { t[(unsigned char)e.c] = e.m; t[(unsigned char)(e.c + 32)] = e.m; }
}

}  // namespace

#include "kernels_pack.inc"
#include "kernels_consensus.inc"
#include "kernels_scan_history.inc"
#include "kernels_scan3.inc"
#include "kernels_replay.inc"
#include "kernels_ball.inc"
#include "kernels_qprep.inc"

// ------------------------------------------------------------------------------------------------------------
// host side, in sections (one translation unit: the kernels above are templates the sections instantiate)
// ------------------------------------------------------------------------------------------------------------
#include "host_launch.inc"
#include "host_qtables.inc"
#include "host_qprep.inc"
#include "host_open.inc"
#include "host_batch.inc"
#include "host_resident.inc"
#include "host_shards.inc"
#include "host_ball.inc"

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
#include <cstring>
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
  hipStream_t scan_stream = nullptr;      // ring mode: scans of later slices run here while the replay chain waits
  hipEvent_t scan_done[NBUF] = {}, replay_done[NBUF] = {};
  // uvaia_gpu_db_rederive: chunks of tiles rebuilt on their own stream; a scan waits for the chunks its slice touches
  struct DeriveChunk { long long t0, t1; hipEvent_t done; };
  hipStream_t derive_stream = nullptr;
  hipStream_t derive_streams[3] = {};     // [0] = derive_stream; the chunks of a rebuild alternate over the first derive_nstreams
  int derive_nstreams = 3;
  std::vector<DeriveChunk> derive_chunks;
  hipEvent_t derive_fence[4] = {};
  size_t derive_pending = 0;          // chunks of the last rederive a scan may still have to wait for
  bool replay_recorded[NBUF] = {}, slice_scanned[NBUF] = {}, slice_cons_done[NBUF] = {};
  size_t slice_cap[NBUF] = {};              // pairs each counter buffer holds (grown when a slice needs more: slices may exceed a pool, see plan_subslices)
  uint32_t *d_cntb[NBUF] = {};            // counter buffers 1..NBUF-1 (buffer 0 is d_cnt2), allocated on first use
  int2 *d_tmin[NBUF] = {};                // per (query, tile of 64 references): {smallest mismatch count, largest ACGT-match count}, one per counter buffer
  int4 *d_rtb[NBUF] = {};                 // per reference of a slice: untruncated consensus pre-score (query sets with constant-and-complete columns), one per counter buffer
  int slice_tiles[NBUF] = {}, slice_rb[NBUF] = {}, slice_re[NBUF] = {};
  long long slice_tf[NBUF] = {};
  size_t subslice = 32768;                // resident search: pools are cut into slices of this size (exact: see search_resident)
  bool subslice_forced = false;           // the length was given (tests): taken as it is
  int nq = 0, nq_pad = 0, nchar = 0, W = 0, W4 = 0, P = 4, NQ = 6, acgt = 0, k = 2, qt = 16, n_idx_c = 0;
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
  uint32_t *d_qrare = nullptr;   // [nq][NR4*4][lo, hi, isACGT] the queries on the rare columns (--acgt: dist_unique of admitted pairs)
  int need_e_groups = 0, need_v_groups = 0, need_g_groups = 0, need_r_groups = 0;   // word groups whose E / V plane some query tile has to read (for the byte accounting)
  int act_q0 = 0, act_q1 = 0;    // active query range of the resident/slice paths (query shards across GPUs); whole set by default
  bool serial = false;           // tuning.serial: no scan/replay overlap (to time the kernels in isolation)
  int replay_lq = -1;            // replay caches the query's planes in LDS (22 KB per block): -1 = only with few queries (see open)
  int replay_prio = 1;           // replay waves raise their issue priority
  int scan_R = 2;                // reference tiles per wave of scan3_kernel (the item stream is built for it)
  int scan_NW = 8;               // waves per block of scan3_kernel = shares a super-tile's records are cut into
  uint4 *d_batch_ev = nullptr, *d_batch_poly = nullptr, *d_db_ev = nullptr, *d_db_poly = nullptr;
  uint32_t *d_batch_grp = nullptr, *d_db_grp = nullptr;   // [tile][W4][64]  popc(E) | popc(V) << 16 of each word group (for queries that are all-N there)
  int *d_batch_tote = nullptr, *d_db_tote = nullptr;
  int *d_amb_q = nullptr;        // [nq][AMB_STRIDE] ambiguity-word lists of the queries
  int *d_batch_amb = nullptr, *d_db_amb = nullptr;   // same for the references of the batch buffer / database
  int *d_batch_tot = nullptr, *d_db_tot = nullptr;   // per reference: valid sites (default) / ACGT sites (--acgt), counted by pack_refs_kernel
  uint32_t *d_cnt2 = nullptr;    // [nq_pad][pool_pad] two-counter scan output, one dword per pair: first | second << 16 
  unsigned long long *d_stats = nullptr;             // admissions, on-demand evaluations, dense fallbacks
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
  int ball_split[10] = {};                                  // ball_gather_cols_kernel: word groups and gathered columns before each of its four waves
  int *d_idx_cols = nullptr; int n_idx = 0, NG4 = 0;       // query->idx (the polymorphic query columns) and the word groups they fill once gathered
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
  size_t db_cap = 0, db_n = 0;
  // last batch (introspection)
  const uint4 *last_tiles = nullptr; const int *last_nonn = nullptr; int last_n = 0, last_rbegin = 0, last_ppad = 0, last_ntiles = 0;
  long long last_tile_first = 0;
  const int4 *last_rt = nullptr;
  // stats
  std::vector<ScanEvt> evts;
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

#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return fail((c), e_ == hipErrorOutOfMemory ? UVAIA_GPU_ENOMEM : UVAIA_GPU_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// reference shards: does this context keep derived planes for packed tile t, and under which number
inline bool owns_tile(const uvaia_gpu_ctx *c, long long t) { return c->shard_world == 1 || (t / c->shard_pt) % c->shard_world == c->shard_rank; }
inline long long dtile_of(const uvaia_gpu_ctx *c, long long t)
{ return c->shard_world == 1 ? t : (t / (c->shard_pt * c->shard_world)) * c->shard_pt + t % c->shard_pt; }
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

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
namespace {

// host-side preparation of a query set is O(queries x columns) several times over: spread the independent pieces over threads
template <class F>
static void parallel_for(int n, F f)
{
  const unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u);   // (16 until round 2: 0.33 s of a 10 000-query open were these loops)
  if (n < 32 || nt < 2) { for (int i = 0; i < n; i++) f(i); return; }
  std::atomic<int> next(0);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++) th.emplace_back([&]() { for (;;) { const int a = next.fetch_add(4); if (a >= n) break; for (int i = a; i < std::min(n, a + 4); i++) f(i); } });
  for (auto &x : th) x.join();
}

// Query planes restricted to the polymorphic columns (query->idx): what the radius search and the redundancy test compare references
// with.  Packing a query row with every other site left out gives its full planes under the mask of those columns, so they are made
// on the device from d_qp by the first call that needs them.
__global__ void mask_query_planes_kernel(const uint32_t *__restrict__ qp, const uint32_t *__restrict__ pmask, uint32_t *__restrict__ out, size_t n_words_total, int words_per_row, int nq_planes)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_words_total) return;
  const int w = (int)((i / (size_t)nq_planes) % (size_t)words_per_row);       // layout [query][word][plane]
  out[i] = qp[i] & pmask[w];
}

int ensure_qpoly(uvaia_gpu_ctx *c)
{
  if (c->d_qpoly) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)c->nq_pad * c->W4 * 4 * c->NQ;
  uint32_t *d = nullptr;
  HIPCHK(c, hipMalloc(&d, n * 4));
  hipLaunchKernelGGL(mask_query_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_qp, c->d_pmask, d, n, c->W4 * 4, c->NQ);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) { hipFree(d); return fail(c, UVAIA_GPU_EHIP, "query planes on the polymorphic columns: %s", hipGetErrorString(e)); }
  c->d_qpoly = d;
  return 0;
}

// The queries on the columns of query->idx, bit-gathered (kernels_ball.inc): what stage 2 of the radius search compares references with.
int ensure_qgather(uvaia_gpu_ctx *c)
{
  if (c->d_qg) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  const int NG = c->acgt ? 3 : 5;
  const size_t n = (size_t)c->nq_pad * c->NG4 * 4 * NG;
  uint32_t *d = nullptr;
  HIPCHK(c, hipMalloc(&d, n * 4));
  hipError_t e = hipMemsetAsync(d, 0, n * 4, c->stream);
  dim3 grid((unsigned)((c->NG4 * 4 + 63) / 64), (unsigned)c->nq);
  if (e == hipSuccess) {
    if (c->acgt) hipLaunchKernelGGL((ball_gather_queries_kernel<true>), grid, dim3(64), 0, c->stream, c->d_qp, c->nq, c->W4, c->d_idx_cols, c->n_idx, c->NG4, d);
    else         hipLaunchKernelGGL((ball_gather_queries_kernel<false>), grid, dim3(64), 0, c->stream, c->d_qp, c->nq, c->W4, c->d_idx_cols, c->n_idx, c->NG4, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) { hipFree(d); return fail(c, UVAIA_GPU_EHIP, "query planes on the polymorphic columns: %s", hipGetErrorString(e)); }
  c->d_qg = d;
  return 0;
}

// Buffers of a streamed batch (uvaia_gpu_push, uvaia_gpu_ball, uvaia_gpu_agree_on_polymorphic): packed tiles of max_pool references,
// the planes derived from them, side rows.  A context that only searches a resident database never needs them.
int ensure_batch_buffers(uvaia_gpu_ctx *c)
{
  if (c->d_batch) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t tile_u4 = (size_t)c->W4 * c->P * 64, tiles = c->pool_pad / 64;
  HIPCHK(c, hipMalloc(&c->d_batch_nonn, c->pool_pad * sizeof(int)));
  HIPCHK(c, hipMemset(c->d_batch_nonn, 0, c->pool_pad * sizeof(int)));
  HIPCHK(c, hipMalloc(&c->d_batch_ev, tiles * (size_t)c->W4 * 2 * 64 * sizeof(uint4)));
  HIPCHK(c, hipMalloc(&c->d_batch_grp, tiles * (size_t)c->W4 * 64 * sizeof(uint32_t)));
  HIPCHK(c, hipMalloc(&c->d_batch_poly, tiles * (size_t)std::max(c->NP4 + c->NR4, 1) * 3 * 64 * sizeof(uint4)));
  HIPCHK(c, hipMalloc(&c->d_batch_tote, c->pool_pad * sizeof(int)));
  HIPCHK(c, hipMalloc(&c->d_batch_tot, c->pool_pad * sizeof(int)));
  HIPCHK(c, hipMemset(c->d_batch_tot, 0, c->pool_pad * sizeof(int)));
  HIPCHK(c, hipMalloc(&c->d_batch_amb, c->pool_pad * AMB_ROW * sizeof(int)));
  HIPCHK(c, hipMemset(c->d_batch_amb, 0, c->pool_pad * AMB_ROW * sizeof(int)));
  HIPCHK(c, hipMalloc(&c->d_batch, tiles * tile_u4 * sizeof(uint4)));       // last: its presence says all of them are there
  HIPCHK(c, hipMemset(c->d_batch, 0, tiles * tile_u4 * sizeof(uint4)));
  return 0;
}

// Packs one character row restricted to `keep` (nullable: keep everything inside [lo,hi)) into query-plane words:
// dst[(w4*4 + j)*NQ + plane].  is_poly marks query->idx columns (--acgt: fourth plane).
int pack_query_row(const uint8_t *code_tab, const char *row, int nchar, int lo, int hi, const uint8_t *keep, const uint8_t *is_poly,
                   bool acgt, int NQ, uint32_t *dst, int *bad_byte)
{
  for (int s = lo; s < hi; s++) {
    if (keep && !keep[s]) continue;
    const uint8_t code = code_tab[(unsigned char)row[s]];
    if (code == 0xFF) { *bad_byte = (unsigned char)row[s]; return -1; }
    if (!code) continue;
    const int w = s >> 5, b = s & 31;
    uint32_t *d = dst + (size_t)w * NQ;      // (w4*4+j) == w
    const bool one = (code & (code - 1)) == 0;
    if (acgt) {
      if (!one) continue;
      const uint32_t two = code == 2 ? 1u : code == 4 ? 2u : code == 8 ? 3u : 0u;
      d[0] |= (two & 1u) << b; d[1] |= (two >> 1) << b; d[2] |= 1u << b;
      if (is_poly && is_poly[s]) d[3] |= 1u << b;
    } else {
      for (int p = 0; p < 4; p++) d[p] |= (uint32_t)((code >> p) & 1u) << b;
      d[4] |= 1u << b;
      if (one) d[5] |= 1u << b;
    }
  }
  return 0;
}

int launch_scan(uvaia_gpu_ctx *c, const uint4 *tiles, long long tile_first, int n_tiles, const uint32_t *qp, int n_rows, int4 *out, int ppad, double bytes)
{
  if (n_tiles <= 0) return 0;
  dim3 grid((unsigned)((n_rows + c->qt - 1) / c->qt), (unsigned)((n_tiles + 3) / 4)), block(256);   // only tiles holding real queries
  ScanEvt ev{};
  if (c->profile) {
    HIPCHK(c, hipEventCreate(&ev.a)); HIPCHK(c, hipEventCreate(&ev.b));
    HIPCHK(c, hipEventRecord(ev.a, c->stream));
  }
#define LAUNCH(K, QT) hipLaunchKernelGGL((K<QT>), grid, block, 0, c->stream, tiles, tile_first, n_tiles, c->W4, qp, out, ppad)
  if (c->acgt) { switch (c->qt) { case 8: LAUNCH(scan_acgt_kernel, 8); break; case 32: LAUNCH(scan_acgt_kernel, 32); break; default: LAUNCH(scan_acgt_kernel, 16); } }
  else         { switch (c->qt) { case 8: LAUNCH(scan_iupac_kernel, 8); break; case 32: LAUNCH(scan_iupac_kernel, 32); break; default: LAUNCH(scan_iupac_kernel, 16); } }
#undef LAUNCH
  HIPCHK(c, hipGetLastError());
  if (c->profile) { HIPCHK(c, hipEventRecord(ev.b, c->stream)); ev.bytes = bytes; c->evts.push_back(ev); }
  return 0;
}

// rt (nullable unless the query set has constant-and-complete columns): the untruncated consensus pre-score of the slice's references,
// by the packed-plane scans themselves or, next to the column-compressed scan, by consensus_rt_kernel on the same stream
int launch_scan2(uvaia_gpu_ctx *c, const uint4 *tiles, const int *tot_tile0, long long tile_first, int n_tiles, uint32_t *out, int ppad, double bytes, hipStream_t stream,
                 int2 *tmin, int r_lo, int r_hi, int4 *rt)
{
  if (n_tiles <= 0) return 0;
  if (!stream) stream = c->stream;
  const bool cons = c->n_idx_c > 0;
  if (cons && !rt) return fail(c, UVAIA_GPU_ESTATE, "no buffer for the consensus pre-score");
  const long long ptile_first = tile_first;       // packed tiles (tile_first may be renumbered for the derived planes below)
  auto consensus_rt = [&]() {
    if (!cons) return;
    if (c->acgt) hipLaunchKernelGGL((consensus_rt_kernel<true>), dim3((n_tiles + 3) / 4), dim3(256), 0, stream, tiles, ptile_first, n_tiles, c->W4, c->d_cp, rt);
    else         hipLaunchKernelGGL((consensus_rt_kernel<false>), dim3((n_tiles + 3) / 4), dim3(256), 0, stream, tiles, ptile_first, n_tiles, c->W4, c->d_cp, rt);
  };
  // Query tile of the packed-plane scans (their partial sums share LDS: no 32).  With eight queries counted per plane word the
  // kernel's VALU time equals its HBM time; a set of one, two or four queries is not padded to eight: the counting shrinks with it
  // and the scan stays bound by HBM.
  const int qt2 = c->qt != 8 ? 16 : c->nq <= 1 ? 1 : c->nq <= 2 ? 2 : c->nq <= 4 ? 4 : 8;
  const int n_qtiles = (c->nq + qt2 - 1) / qt2;
  dim3 grid(scan_grid_size(n_qtiles, n_tiles)), block(256);      // the packed-plane scans: one block per (query tile, tile of references)
  ScanEvt ev_{};
  if (c->profile) {
    HIPCHK(c, hipEventCreate(&ev_.a)); HIPCHK(c, hipEventCreate(&ev_.b));
    HIPCHK(c, hipEventRecord(ev_.a, stream));
  }
  const uint32_t *qp = c->acgt ? c->d_qp : c->d_qp2;
  if (c->scan_variant == 2) {
    const bool is_db = (tiles == c->d_db);
    const uint4 *ev = is_db ? c->d_db_ev : c->d_batch_ev, *poly = is_db ? c->d_db_poly : c->d_batch_poly;
    const uint32_t *grp = is_db ? c->d_db_grp : c->d_batch_grp;
    if (is_db && c->shard_world > 1) {    // the derived planes of an owned piece are numbered densely (uvaia_gpu_db_set_shard)
      if (!owns_tile(c, tile_first) || tile_first / c->shard_pt != (tile_first + n_tiles - 1) / c->shard_pt)
        return fail(c, UVAIA_GPU_ESTATE, "tiles %lld..%lld are not inside one piece of this context's reference shard", tile_first, tile_first + n_tiles - 1);
      tile_first = dtile_of(c, tile_first);
    }
    const int *tote = (is_db ? c->d_db_tote : c->d_batch_tote) + tile_first * 64;
    constexpr int QS = 64;                       // queries of a super-tile of scan3_kernel
    if (c->act_q0 % QS) return fail(c, UVAIA_GPU_ESTATE, "the scan works on super-tiles of %d queries: active queries start at a multiple of that", QS);
    const int st_first = c->act_q0 / QS, n_st = (c->act_q1 + QS - 1) / QS - st_first;
    const int R = c->scan_R;
    dim3 grid3(scan_grid_size(n_st, (n_tiles + R - 1) / R));
#define SCAN3_LAUNCH(NWW, A, RR) hipLaunchKernelGGL((scan3_kernel<NWW, A, RR>), grid3, dim3(64 * NWW), 0, stream, ev, poly, tile_first, n_tiles, c->W4, c->NP4, c->NP4 + c->NR4, c->d_qpl, c->d_stream, c->d_sdir, grp, tote, tot_tile0, out, ppad, n_st, tmin, r_lo, r_hi, st_first)
#define SCAN3_NW(A, RR) { if (c->scan_NW == 8) SCAN3_LAUNCH(8, A, RR); else SCAN3_LAUNCH(4, A, RR); }
    if (R == 4)      { if (c->acgt) SCAN3_LAUNCH(8, true, 4); else SCAN3_LAUNCH(8, false, 4); }     // four tiles per wave: eight waves only (open_tuned)
    else if (R == 2) { if (c->acgt) SCAN3_NW(true, 2) else SCAN3_NW(false, 2) }
    else             { if (c->acgt) SCAN3_NW(true, 1) else SCAN3_NW(false, 1) }
#undef SCAN3_NW
#undef SCAN3_LAUNCH
    HIPCHK(c, hipGetLastError());
    if (c->profile) { HIPCHK(c, hipEventRecord(ev_.b, stream)); ev_.bytes = bytes; c->evts.push_back(ev_); }
    consensus_rt();
    HIPCHK(c, hipGetLastError());
    return 0;
  }
#define LAUNCH(K, QT, CN) hipLaunchKernelGGL((K<QT, CN>), grid, block, 0, stream, tiles, tile_first, n_tiles, c->W4, qp, out, ppad, n_qtiles, tot_tile0, tmin, r_lo, r_hi, c->d_cp, rt)
#define LAUNCH_QT(K, CN) switch (qt2) { case 1: LAUNCH(K, 1, CN); break; case 2: LAUNCH(K, 2, CN); break; case 4: LAUNCH(K, 4, CN); break; case 8: LAUNCH(K, 8, CN); break; default: LAUNCH(K, 16, CN); }
  if (c->acgt) { if (cons) { LAUNCH_QT(scan2_acgt_kernel, true) } else { LAUNCH_QT(scan2_acgt_kernel, false) } }
  else         { if (cons) { LAUNCH_QT(scan2_iupac_kernel, true) } else { LAUNCH_QT(scan2_iupac_kernel, false) } }
#undef LAUNCH_QT
#undef LAUNCH
  HIPCHK(c, hipGetLastError());
  if (c->profile) { HIPCHK(c, hipEventRecord(ev_.b, stream)); ev_.bytes = bytes; c->evts.push_back(ev_); }
  return 0;
}

int ensure_cnt4(uvaia_gpu_ctx *c, size_t elems)
{
  if (c->cnt_cap >= elems) return 0;
  if (c->d_cnt) { HIPCHK(c, hipFree(c->d_cnt)); c->d_cnt = nullptr; c->cnt_cap = 0; }
  HIPCHK(c, hipMalloc(&c->d_cnt, elems * sizeof(int4)));
  c->cnt_cap = elems;
  return 0;
}

int collect_events(uvaia_gpu_ctx *c)
{
  for (auto &e : c->evts) {
    HIPCHK(c, hipEventSynchronize(e.b));
    float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, e.a, e.b));
    c->scan_ms += ms; c->scan_bytes += e.bytes; c->scan_launches++;
    hipEventDestroy(e.a); hipEventDestroy(e.b);
  }
  c->evts.clear();
  return 0;
}

// One batch = one pool of the reference (src/nearest.c:288-306), on tiles [tile_first, tile_first+n_tiles) of `tiles`;
// references r_begin..r_end-1 (relative to the first tile) are the batch, in order.
int run_batch(uvaia_gpu_ctx *c, const uint4 *tiles, const int *nonn_tile0, const int *amb_tile0, long long tile_first, int n_tiles, int r_begin, int r_end,
              long long ord_base, uint8_t *entered_tile0)
{
  if (r_end <= r_begin) {   // an empty trailing batch only refreshes cq->max_incompatible (src/nearest.c:290-291)
    return 0;
  }
  const int ppad = n_tiles * 64;
  hipLaunchKernelGGL(snapshot_kernel, dim3(1), dim3(256), 0, c->stream, c->d_T, c->nq, c->d_snap);
  if (c->n_idx_c > 0 && c->fullscan) {   // with no constant-and-complete column every pre-score counter is zero (common: gappy query sets)
    if (c->acgt) hipLaunchKernelGGL((consensus_kernel<true>), dim3((n_tiles + 3) / 4), dim3(256), 0, c->stream, tiles, tile_first, n_tiles, c->W4, c->d_cp, c->d_snap, c->d_rt, c->d_tr);
    else         hipLaunchKernelGGL((consensus_kernel<false>), dim3((n_tiles + 3) / 4), dim3(256), 0, c->stream, tiles, tile_first, n_tiles, c->W4, c->d_cp, c->d_snap, c->d_rt, c->d_tr);
  }
  HIPCHK(c, hipGetLastError());
  const double bytes = (double)(r_end - r_begin) * (double)c->W4 * 16.0 * c->P + (double)c->nq * (double)c->W4 * 16.0 * c->P;
  size_t lds = (size_t)(c->k + 1) * HEAP_ENTRY * sizeof(int);
  const int lq_words = (c->replay_lq && !c->acgt && !c->fullscan && lds + (size_t)c->W4 * 4 * 6 * 4 + 128 <= 64 * 1024) ? c->W4 * 4 * 6 : 0;   // query planes cached in LDS
  if (c->fullscan) {
    int rc = ensure_cnt4(c, (size_t)c->nq_pad * c->pool_pad); if (rc) return rc;
    rc = launch_scan(c, tiles, tile_first, n_tiles, c->d_qp, c->nq, c->d_cnt, ppad, bytes);
    if (rc) return rc;
    if (c->acgt) hipLaunchKernelGGL((replay_kernel<true>), dim3(c->nq), dim3(64), lds, c->stream, c->d_cnt, ppad, c->d_rt, c->d_tr, nonn_tile0, r_begin, r_end, ord_base, c->d_heap, c->d_n, c->d_T, c->d_snap, entered_tile0, c->k);
    else         hipLaunchKernelGGL((replay_kernel<false>), dim3(c->nq), dim3(64), lds, c->stream, c->d_cnt, ppad, c->d_rt, c->d_tr, nonn_tile0, r_begin, r_end, ord_base, c->d_heap, c->d_n, c->d_T, c->d_snap, entered_tile0, c->k);
  } else {
    int rc = launch_scan2(c, tiles, (tiles == c->d_db ? c->d_db_tot : c->d_batch_tot) + tile_first * 64, tile_first, n_tiles, c->d_cnt2, ppad, bytes, nullptr, c->d_tmin[0], r_begin, r_end, c->d_rtb[0]);
    if (rc) return rc;
#define REPLAY2(A, B) hipLaunchKernelGGL((replay2_kernel<A, B>), dim3(c->nq), dim3(64), lds + (size_t)lq_words * 4 + 128, c->stream, c->d_cnt2, ppad, c->d_rtb[0], c->d_cp, nonn_tile0, amb_tile0, r_begin, r_end, ord_base, \
                                    c->d_heap, c->d_n, c->d_T, c->d_snap, entered_tile0, c->k, tiles, tile_first, c->W4, c->d_qp, c->d_amb_q, c->d_stats, 0, (c->scan_variant == 2 || c->scan_variant == 0) ? c->d_tmin[0] : (const int2 *)nullptr, \
                                    c->scan_variant == 2 ? c->d_qpl : (const uint32_t *)nullptr, lq_words, c->replay_prio, (tiles == c->d_db ? c->d_db_poly : c->d_batch_poly), c->NP4 + c->NR4, c->NP4, c->NR4, c->d_qrare)
    if (c->acgt) { if (c->n_idx_c > 0) REPLAY2(true, true); else REPLAY2(true, false); }
    else         { if (c->n_idx_c > 0) REPLAY2(false, true); else REPLAY2(false, false); }
#undef REPLAY2
  }
  HIPCHK(c, hipGetLastError());
  c->last_tiles = tiles; c->last_nonn = nonn_tile0; c->last_n = r_end - r_begin; c->last_rbegin = r_begin; c->last_ppad = ppad; c->last_rt = c->fullscan ? c->d_rt : c->d_rtb[0];
  c->last_ntiles = n_tiles; c->last_tile_first = tile_first;
  return 0;
}

// planes derived for the open query set (column-compressed scan) for the whole tiles that hold slots slot0 .. slot0 + n_ref - 1
// (resident database with reference shards: only the tiles of the pieces this context owns)
int derive_rows(uvaia_gpu_ctx *c, uint4 *tiles, long long slot0, int n_ref, hipStream_t st = nullptr, bool v_in_place = false)
{
  if (!st) st = c->stream;
  if (c->fullscan || c->scan_variant != 2 || n_ref <= 0 || !c->d_split) return 0;     // only the column-compressed scan reads derived planes
  const bool is_db = (tiles == c->d_db);
  uint4 *ev = is_db ? c->d_db_ev : c->d_batch_ev, *poly = is_db ? c->d_db_poly : c->d_batch_poly;
  int *tote = is_db ? c->d_db_tote : c->d_batch_tote;
  uint32_t *grp = is_db ? c->d_db_grp : c->d_batch_grp;
  const long long t0 = slot0 / 64, t1 = (slot0 + n_ref - 1) / 64;
  const bool sharded = is_db && c->shard_world > 1;
  for (long long a = t0; a <= t1;) {
    // [a, b]: the whole range, or its part inside one piece of the shard map
    const long long b = sharded ? std::min(t1, (a / c->shard_pt + 1) * c->shard_pt - 1) : t1;
    if (!sharded || owns_tile(c, a)) {
      const int nblk = (int)(b - a + 1);
      const long long dt = sharded ? dtile_of(c, a) : a;
#define DERIVE_ALL(A, V) hipLaunchKernelGGL((derive_all_kernel<A, V>), dim3(nblk), dim3(256), 0, st, tiles, a, dt, c->W4, c->d_cls, c->d_rmask, c->d_split, c->NP4, c->NR4, ev, tote, grp, poly)
      if (c->acgt) { if (v_in_place) DERIVE_ALL(true, false); else DERIVE_ALL(true, true); }
      else         { if (v_in_place) DERIVE_ALL(false, false); else DERIVE_ALL(false, true); }
#undef DERIVE_ALL
    }
    a = b + 1;
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}

// a rebuild of the derived planes still in flight (uvaia_gpu_db_rederive) must end before the database changes
static int settle_derive(uvaia_gpu_ctx *c)
{
  if (c->derive_pending) { for (int i_ = 0; i_ < 3; i_++) if (c->derive_streams[i_]) HIPCHK(c, hipStreamSynchronize(c->derive_streams[i_])); c->derive_pending = 0; }
  return 0;
}

// stage + pack n_ref rows (either scattered pointers or one pitched block) into `tiles` starting at slot0
int pack_rows(uvaia_gpu_ctx *c, const char *const *seq, const char *rows, size_t rows_pitch, const int *non_n, int n_ref,
              uint4 *tiles, int *nonn_dev, int *amb_dev, int *tot_dev, long long slot0)
{
  // two staging buffers: while chunk k crosses PCIe and is packed, the host threads copy chunk k + 1 into the other pinned buffer
  // (the hand-over of raw characters, 30 KB per reference, is what bounds the streaming entry points, not the kernels)
  for (int done = 0, k = 0; done < n_ref; done += PACK_CHUNK, k ^= 1) {
    const int m = std::min(PACK_CHUNK, n_ref - done);
    if (c->stage_busy[k]) { HIPCHK(c, hipEventSynchronize(c->stage_free[k])); c->stage_busy[k] = false; }     // its previous chunk has left the buffer
    uint8_t *hs = c->h_stage + (size_t)k * PACK_CHUNK * c->pitch, *ds = c->d_stage + (size_t)k * PACK_CHUNK * c->pitch;
    for (int i = 0; i < m; i++) if (!(seq ? seq[done + i] : rows)) return fail(c, UVAIA_GPU_EINVAL, "NULL sequence at position %d", done + i);
    parallel_for(m, [&](int i) {
      const char *src = seq ? seq[done + i] : rows + (size_t)(done + i) * rows_pitch;
      memcpy(hs + (size_t)i * c->pitch, src, (size_t)c->nchar);
    });
    HIPCHK(c, hipMemcpyAsync(ds, hs, (size_t)m * c->pitch, hipMemcpyHostToDevice, c->stream));
    const long long s0 = slot0 + done, t0 = s0 / 64, t1 = (s0 + m - 1) / 64;
    const int nblk = (int)(t1 - t0 + 1);
    int *nn_out = non_n ? nullptr : nonn_dev;
    if (c->acgt) hipLaunchKernelGGL((pack_refs_kernel<3>), dim3(nblk), dim3(256), 0, c->stream, ds, c->pitch, c->nchar, s0, m, c->W4, tiles, t0, nn_out, (int *)nullptr, tot_dev, c->d_err);
    else         hipLaunchKernelGGL((pack_refs_kernel<4>), dim3(nblk), dim3(256), 0, c->stream, ds, c->pitch, c->nchar, s0, m, c->W4, tiles, t0, nn_out, amb_dev, tot_dev, c->d_err);
    HIPCHK(c, hipGetLastError());
    if (non_n) HIPCHK(c, hipMemcpyAsync(nonn_dev + s0, non_n + done, (size_t)m * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->stage_free[k], c->stream)); c->stage_busy[k] = true;
  }
  { int rc = derive_rows(c, tiles, slot0, n_ref); if (rc) return rc; }
  HIPCHK(c, hipStreamSynchronize(c->stream));     // scans may start on another stream: the packed and derived planes must be complete
  int bad = 0;
  HIPCHK(c, hipMemcpy(&bad, c->d_err, sizeof(int), hipMemcpyDeviceToHost));
  if (bad) {
    HIPCHK(c, hipMemset(c->d_err, 0, sizeof(int)));
    return fail(c, UVAIA_GPU_EALPHABET, "a reference sequence holds a byte outside ACGT / MRWSYKVHDB / NX-?O.");
  }
  return 0;
}

}  // namespace

extern "C" {

int uvaia_gpu_slice_scan(uvaia_gpu_ctx *c, size_t first, size_t n, int buf);
int uvaia_gpu_slice_replay(uvaia_gpu_ctx *c, int buf, int64_t ordinal0, int stripe_start);
size_t uvaia_gpu_state_range_bytes(const uvaia_gpu_ctx *c, int q0, int q1);

const char *uvaia_gpu_last_error(const uvaia_gpu_ctx *ctx) { return ctx ? ctx->err.c_str() : g_open_error.c_str(); }

void uvaia_gpu_close(uvaia_gpu_ctx *c)
{
  if (!c) return;
  if (c->stream) hipStreamSynchronize(c->stream);
  for (auto &e : c->evts) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  void *dev[] = {c->d_idx_cols, c->d_qg, c->d_ball_key, c->d_split, c->d_qrare, c->d_rmask, c->d_batch_grp, c->d_db_grp, c->d_cls, c->d_qpl, c->d_stream, c->d_sdir, c->d_batch_ev, c->d_batch_poly, c->d_db_ev, c->d_db_poly, c->d_batch_tote, c->d_db_tote,
                 c->d_batch_tot, c->d_db_tot, c->d_mindist, c->d_ball_list, c->d_ball_cdist, c->d_ball_n, c->d_ball_tiles, c->d_qp2, c->d_amb_q, c->d_batch_amb, c->d_db_amb, c->d_cnt2, c->d_stats, c->d_qp, c->d_cp, c->d_cpm, c->d_qpoly, c->d_pmask, c->d_heap, c->d_n, c->d_T, c->d_snap, c->d_err, c->d_batch, c->d_batch_nonn,
                 c->d_cnt, c->d_rt, c->d_tr, c->d_entered, c->d_stage, c->d_db, c->d_db_nonn};
  for (void *p : dev) if (p) hipFree(p);
  if (c->h_stage) hipHostFree(c->h_stage);
  for (int i = 0; i < 2; i++) if (c->stage_free[i]) hipEventDestroy(c->stage_free[i]);
  for (int i = 0; i < 16; i++) if (c->order_ev[i]) hipEventDestroy(c->order_ev[i]);
  for (int i = 0; i < 8; i++) for (int j = 0; j < 3; j++) if (c->mark_ev[i][j]) hipEventDestroy(c->mark_ev[i][j]);
  for (int i = 0; i < NBUF; i++) { if (c->d_cntb[i]) hipFree(c->d_cntb[i]); if (c->d_tmin[i]) hipFree(c->d_tmin[i]); if (c->d_rtb[i]) hipFree(c->d_rtb[i]); }
  for (int i = 0; i < NBUF; i++) { if (c->scan_done[i]) hipEventDestroy(c->scan_done[i]); if (c->replay_done[i]) hipEventDestroy(c->replay_done[i]); }
  for (int i_ = 0; i_ < 3; i_++) if (c->derive_streams[i_]) { hipStreamSynchronize(c->derive_streams[i_]); hipStreamDestroy(c->derive_streams[i_]); }
  for (auto &d : c->derive_chunks) hipEventDestroy(d.done);
  for (int i = 0; i < 4; i++) if (c->derive_fence[i]) hipEventDestroy(c->derive_fence[i]);
  if (c->scan_stream) hipStreamDestroy(c->scan_stream);
  for (int i = 1; i < 3; i++) if (c->scan_streams[i]) hipStreamDestroy(c->scan_streams[i]);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int uvaia_gpu_open(uvaia_gpu_ctx **out, const uvaia_gpu_query *q, int heap_size, int device, size_t max_pool)
{ return uvaia_gpu_open_tuned(out, q, heap_size, device, max_pool, nullptr); }

int uvaia_gpu_open_tuned(uvaia_gpu_ctx **out, const uvaia_gpu_query *q, int heap_size, int device, size_t max_pool, const uvaia_gpu_tuning *tune)
{
  uvaia_gpu_tuning tn;
  memset(&tn, 0, sizeof tn);
  if (tune) tn = *tune;
  if (tn.scan < 0 || tn.scan > UVAIA_GPU_SCAN_WIDE || (tn.scan_tiles_per_wave != 0 && tn.scan_tiles_per_wave != 1 && tn.scan_tiles_per_wave != 2 && tn.scan_tiles_per_wave != 4) ||
      (tn.scan_waves_per_block != 0 && tn.scan_waves_per_block != 4 && tn.scan_waves_per_block != 8) || (tn.subslice_refs != 0 && tn.subslice_refs < 64))
    return fail(nullptr, UVAIA_GPU_EINVAL, "bad tuning values");
  if (!out) return fail(nullptr, UVAIA_GPU_EINVAL, "ctx is NULL");
  *out = nullptr;
  if (!q || q->n_query < 1 || q->nchar < 1 || !q->seq || !q->consensus) return fail(nullptr, UVAIA_GPU_EINVAL, "empty or incomplete query set");
  if ((q->n_idx_c && !q->idx_c) || (q->n_idx_m && !q->idx_m) || (q->n_idx && !q->idx)) return fail(nullptr, UVAIA_GPU_EINVAL, "index arrays missing");
  if (max_pool < 1) return fail(nullptr, UVAIA_GPU_EINVAL, "max_pool must be >= 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, UVAIA_GPU_ENODEV, "no HIP device available: the MI355X engine has no CPU fallback");
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
  if (device >= ndev) return fail(nullptr, UVAIA_GPU_ENODEV, "device %d out of range (%d devices)", device, ndev);
  if (hipSetDevice(device) != hipSuccess) return fail(nullptr, UVAIA_GPU_ENODEV, "hipSetDevice(%d) failed", device);

  uvaia_gpu_ctx *c = new uvaia_gpu_ctx();
  c->device = device;
  c->nq = q->n_query; c->act_q0 = 0; c->act_q1 = q->n_query; c->nchar = q->nchar; c->acgt = q->acgt ? 1 : 0; c->trim = q->trim; c->n_idx_c = q->n_idx_c;
  c->P = c->acgt ? 3 : 4; c->NQ = c->acgt ? 4 : 6;
  c->W = (c->nchar + 31) / 32; c->W4 = (c->W + 3) / 4;
  c->k = heap_size < 2 ? 2 : heap_size;                      // src/min_heap.c:58
  c->qt = c->nq <= 8 ? 8 : 16;
  // Up to two query tiles: the column-compressed scan would move 25 KB per reference (building its planes) to read 1-8 KB; the
  // two-counter scan over the packed planes reads each reference once per query tile (15 KB) and needs nothing derived (DESIGN.md
  // 4.1; measured per 1 M references: 16 queries 5.3 ms against 8.8, 32 queries 8.2 against 9.8, 64 queries 12.6 against 8.9).
  c->scan_variant = tn.scan == UVAIA_GPU_SCAN_PACKED ? 0 : tn.scan == UVAIA_GPU_SCAN_COMPRESSED ? 2 : (c->nq <= 32 ? 0 : 2);
  c->fullscan = tn.scan == UVAIA_GPU_SCAN_WIDE;
  c->serial = tn.serial != 0;
  if (tn.scan_tiles_per_wave) c->scan_R = tn.scan_tiles_per_wave;
  if (tn.scan_waves_per_block) c->scan_NW = tn.scan_waves_per_block;
  if (c->scan_R == 4 && c->scan_NW != 8) { delete c; return fail(nullptr, UVAIA_GPU_EINVAL, "four reference tiles per wave go with eight waves per block"); }
  if (tn.subslice_refs) { c->subslice = tn.subslice_refs; c->subslice_forced = true; }
  if (tn.rederive_streams >= 1 && tn.rederive_streams <= 3) c->derive_nstreams = tn.rederive_streams;
  // the default scan keeps per-pair deficits in 16-bit halves (LDS counters): alignments of more than ~49 000 columns take the
  // four-counter scan instead (32-bit counts, same results, slower)
  if (c->nchar > 49000) c->fullscan = true;
  c->nq_pad = ((c->nq + 127) / 128) * 128;                   // multiple of every supported query tile and of the scan's super-tiles (64 or 128 queries)
  c->max_pool = max_pool; c->pool_pad = ((max_pool + 63) / 64) * 64 + 64;
  c->pitch = ((size_t)c->nchar + 63) / 64 * 64;
  if ((size_t)(c->k + 1) * HEAP_ENTRY * sizeof(int) + 128 > 160 * 1024) { delete c; return fail(nullptr, UVAIA_GPU_EINVAL, "nbest=%d does not fit the per-query LDS heap (max 5115)", heap_size); }

#define OPENCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int code_ = fail(nullptr, e_ == hipErrorOutOfMemory ? UVAIA_GPU_ENOMEM : UVAIA_GPU_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); uvaia_gpu_close(c); return code_; } } while (0)
  {  // the gate/replay stream outranks the scan stream: its few waves sit on the critical path of the state chain
    int prio_least = 0, prio_greatest = 0;
    OPENCHK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    OPENCHK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest));
    OPENCHK(hipStreamCreateWithPriority(&c->scan_stream, hipStreamNonBlocking, prio_least));
    c->scan_streams[0] = c->scan_stream;
    for (int i = 1; i < 3; i++) OPENCHK(hipStreamCreateWithPriority(&c->scan_streams[i], hipStreamNonBlocking, prio_least));
    for (int i = 0; i < NBUF; i++) { OPENCHK(hipEventCreateWithFlags(&c->scan_done[i], hipEventDisableTiming)); OPENCHK(hipEventCreateWithFlags(&c->replay_done[i], hipEventDisableTiming)); }
    // between the scan (lowest) and the replay (highest): its blocks take the slots scan blocks free, ahead of the next scan blocks
    for (int i = 0; i < 3; i++) OPENCHK(hipStreamCreateWithPriority(&c->derive_streams[i], hipStreamNonBlocking, (prio_least + prio_greatest) / 2));
    c->derive_stream = c->derive_streams[0];
  }
  uint8_t code_tab[256]; fill_code_table(code_tab);
  OPENCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_code), code_tab, 256));

  // ---- query planes (trimmed: sites outside [trim, nchar-trim) never count, src/fastaseq.c:744,750,763)
  const int lo = (int)std::min<size_t>(c->trim, (size_t)c->nchar), hi = std::max(lo, c->nchar - (int)c->trim);
  const size_t row_words = (size_t)c->W4 * 4 * c->NQ;
  std::vector<uint8_t> in_c(c->nchar, 0), in_m(c->nchar, 0), in_p(c->nchar, 0);
  for (int i = 0; i < q->n_idx_c; i++) if (q->idx_c[i] < (size_t)c->nchar) in_c[q->idx_c[i]] = 1;
  for (int i = 0; i < q->n_idx_m; i++) if (q->idx_m[i] < (size_t)c->nchar) in_m[q->idx_m[i]] = 1;
  for (int i = 0; i < q->n_idx; i++)   if (q->idx[i]   < (size_t)c->nchar) in_p[q->idx[i]] = 1;
  std::vector<uint32_t> qp((size_t)c->nq_pad * row_words, 0u), cp(row_words, 0u), cpm(row_words, 0u);
  int bad = 0;
  for (int i = 0; i < c->nq; i++) if (!q->seq[i]) { uvaia_gpu_close(c); return fail(nullptr, UVAIA_GPU_EINVAL, "query %d is NULL", i); }
  {
    std::atomic<int> first_bad(c->nq);
    std::vector<int> bad_byte((size_t)c->nq, 0);
    parallel_for(c->nq, [&](int i) {
      if (pack_query_row(code_tab, q->seq[i], c->nchar, lo, hi, nullptr, in_p.data(), c->acgt, c->NQ, qp.data() + (size_t)i * row_words, &bad_byte[(size_t)i])) {
        int cur = first_bad.load();
        while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
      }
    });
    if (first_bad.load() < c->nq) {
      const int i = first_bad.load();
      uvaia_gpu_close(c); return fail(nullptr, UVAIA_GPU_EALPHABET, "query %d holds byte 0x%02x outside the supported alphabet", i, bad_byte[(size_t)i]);
    }
  }
  if (pack_query_row(code_tab, q->consensus, c->nchar, 0, c->nchar, in_c.data(), nullptr, c->acgt, c->NQ, cp.data(), &bad) ||
      pack_query_row(code_tab, q->consensus, c->nchar, 0, c->nchar, in_m.data(), nullptr, c->acgt, c->NQ, cpm.data(), &bad)) {
    uvaia_gpu_close(c); return fail(nullptr, UVAIA_GPU_EALPHABET, "consensus holds byte 0x%02x outside the supported alphabet", bad);
  }
  OPENCHK(hipMalloc(&c->d_qp, qp.size() * 4)); OPENCHK(hipMemcpy(c->d_qp, qp.data(), qp.size() * 4, hipMemcpyHostToDevice));
  {  // recoded planes and ambiguity-word lists for the two-counter path
    std::vector<int> ambq((size_t)c->nq * AMB_STRIDE, 0);
    if (!c->acgt) {
      const size_t row2 = (size_t)c->W4 * 16;
      std::vector<uint32_t> qp2((size_t)c->nq_pad * row2, 0u);
      parallel_for(c->nq, [&](int i) { for (int w = 0; w < c->W4 * 4; w++) {
        const uint32_t *s6 = qp.data() + (size_t)i * row_words + (size_t)w * 6;
        uint32_t *d4 = qp2.data() + (size_t)i * row2 + (size_t)w * 4;
        const uint32_t one = s6[5];
        d4[0] = (s6[1] | s6[3]) & one; d4[1] = (s6[2] | s6[3]) & one; d4[2] = one; d4[3] = s6[4];
        if (s6[4] & ~one) { int &cnt = ambq[(size_t)i * AMB_STRIDE]; if (cnt < AMB_CAP) ambq[(size_t)i * AMB_STRIDE + 1 + cnt] = w; cnt++; }
      } });
      OPENCHK(hipMalloc(&c->d_qp2, qp2.size() * 4)); OPENCHK(hipMemcpy(c->d_qp2, qp2.data(), qp2.size() * 4, hipMemcpyHostToDevice));
    }
    {  // column classes and compressed/dirty query planes for scan3_kernel
      const int Wp = c->W4 * 4;
      auto QL = [&](int i, int w, int pl) -> uint32_t {          // (lo, hi, isACGT, valid-or-isACGT) of query i, word w
        if (c->acgt) { const uint32_t *s4 = qp.data() + (size_t)i * row_words + (size_t)w * 4; return pl == 3 ? s4[2] : s4[pl]; }
        const uint32_t *s6 = qp.data() + (size_t)i * row_words + (size_t)w * 6; const uint32_t one = s6[5];
        return pl == 0 ? ((s6[1] | s6[3]) & one) : pl == 1 ? ((s6[2] | s6[3]) & one) : pl == 2 ? one : s6[4];
      };
      std::vector<uint32_t> cls((size_t)Wp * 4, 0u);
      parallel_for(Wp, [&](int w) {
        uint32_t cL = 0, cH = 0, seen = 0, poly = 0;
        for (int i = 0; i < c->nq; i++) {
          const uint32_t qL = QL(i, w, 0), qH = QL(i, w, 1), qI = QL(i, w, 2);
          poly |= seen & qI & ((qL ^ cL) | (qH ^ cH));
          const uint32_t fresh = qI & ~seen;
          cL |= qL & fresh; cH |= qH & fresh; seen |= qI;
        }
        cls[(size_t)w * 4 + 0] = cL & ~poly; cls[(size_t)w * 4 + 1] = cH & ~poly; cls[(size_t)w * 4 + 2] = seen & ~poly; cls[(size_t)w * 4 + 3] = poly;
      });
      // Rare columns: polymorphic, but all except a few queries carry the same base (private mutations, sequencing noise: 95 % of
      // the polymorphic columns of the benchmark queries).  They are handled like constant columns with that base; the few
      // queries that differ are "dirty" there (their E bit is taken away by the usual items) and get the true comparison from a
      // sparse item on the gathered planes of the rare columns.  Dense work remains for the truly polymorphic columns only.
      std::vector<uint32_t> rmask((size_t)Wp, 0u);
      {
        c->rare_max = tn.rare_max > 0 ? tn.rare_max : tn.rare_max < 0 ? 0 : (c->nq < 64 ? 0 : std::min(64, std::max(4, c->nq / 64)));
        if (c->fullscan || c->scan_variant != 2) c->rare_max = 0;
      }
      if (c->rare_max > 0) {
        parallel_for(Wp, [&](int w) {
          const uint32_t pm = cls[(size_t)w * 4 + 3];
          if (!pm) return;
          int cnt[32 * 4] = {0};
          for (int i = 0; i < c->nq; i++) {
            const uint32_t qL = QL(i, w, 0), qH = QL(i, w, 1);
            for (uint32_t m = QL(i, w, 2) & pm; m; m &= m - 1) { const int b = __builtin_ctz(m); cnt[(size_t)b * 4 + (((qL >> b) & 1u) | (((qH >> b) & 1u) << 1))]++; }
          }
          for (uint32_t m = pm; m; m &= m - 1) {
            const int b = __builtin_ctz(m);
            int major = 0, total = 0;
            for (int k = 0; k < 4; k++) { total += cnt[(size_t)b * 4 + k]; if (cnt[(size_t)b * 4 + k] > cnt[(size_t)b * 4 + major]) major = k; }
            if (total - cnt[(size_t)b * 4 + major] > c->rare_max) continue;
            const uint32_t bit = 1u << b;
            cls[(size_t)w * 4 + 0] = (cls[(size_t)w * 4 + 0] & ~bit) | ((major & 1) ? bit : 0u);
            cls[(size_t)w * 4 + 1] = (cls[(size_t)w * 4 + 1] & ~bit) | ((major & 2) ? bit : 0u);
            cls[(size_t)w * 4 + 2] |= bit; cls[(size_t)w * 4 + 3] &= ~bit; rmask[(size_t)w] |= bit;
          }
        });
      }
      for (int w = 0; w < Wp; w++) { c->NP += __builtin_popcount(cls[(size_t)w * 4 + 3]); c->NR += __builtin_popcount(rmask[(size_t)w]); }
      c->NP4 = ((c->NP + 31) / 32 + 3) / 4;
      c->NR4 = ((c->NR + 31) / 32 + 3) / 4;
      // the compressed scan keeps a pair's deficit in 16 bits around SCAN3_BIAS: what the polymorphic and the rare columns can take away
      // has to stay below it (16 000 such columns of at most 49 000: no SARS-CoV-2 query set comes near); else the packed-plane scan
      if (c->scan_variant == 2 && (size_t)c->NP4 * 128 + (size_t)c->NR4 * 128 > SCAN3_BIAS - 256) c->scan_variant = 0;
      struct RareWord { int word; uint32_t m, l, h; };                  // one query's minority sites in one compressed word of the rare columns
      std::vector<std::vector<RareWord>> rare_q((size_t)c->nq_pad);
      std::vector<uint32_t> qrare((size_t)c->nq * std::max(c->NR4, 1) * 12, 0u);
      const size_t prow = (size_t)std::max(c->NP4, 1) * 16, crow = (size_t)c->W4 * 8;
      std::vector<uint32_t> qpl((size_t)c->nq_pad * prow, 0u), qcv((size_t)c->nq_pad * crow, 0u), flg((size_t)(c->nq_pad / 16) * c->W4 * 2, 0u);
      parallel_for(c->nq_pad / 16, [&](int tile_) { for (int i = tile_ * 16; i < tile_ * 16 + 16; i++) {   // a tile's 16 queries share flag words
        int k = 0, kr = 0;                                           // compressed bit position among the dense / the rare columns
        bool full = false;                                           // all 128 columns of the current word group are N/gap
        for (int w = 0; w < Wp; w++) {
          const bool real = i < c->nq;
          const uint32_t qL = real ? QL(i, w, 0) : 0u, qH = real ? QL(i, w, 1) : 0u, qI = real ? QL(i, w, 2) : 0u, qV = real ? QL(i, w, 3) : 0u;
          for (uint32_t m = cls[(size_t)w * 4 + 3]; m; m &= m - 1, k++) {
            const int b = __builtin_ctz(m);
            uint32_t *d = qpl.data() + (size_t)i * prow + (size_t)(k >> 7) * 16 + ((k >> 5) & 3);     // [p4][L,H,I,-][word of the group]
            d[0] |= ((qL >> b) & 1u) << (k & 31); d[4] |= ((qH >> b) & 1u) << (k & 31); d[8] |= ((qI >> b) & 1u) << (k & 31);
          }
          // dirty on a constant (or rare) column = not carrying the column's base there
          const uint32_t eqb = qI & ~((qL ^ cls[(size_t)w * 4 + 0]) | (qH ^ cls[(size_t)w * 4 + 1]));
          const uint32_t nI = ~eqb & cls[(size_t)w * 4 + 2], nV = ~qV;
          for (uint32_t m = rmask[(size_t)w]; m; m &= m - 1, kr++) {
            const int b = __builtin_ctz(m);
            if (real) { uint32_t *d = qrare.data() + ((size_t)i * c->NR4 * 4 + (size_t)(kr >> 5)) * 3;
                        d[0] |= ((qL >> b) & 1u) << (kr & 31); d[1] |= ((qH >> b) & 1u) << (kr & 31); d[2] |= ((qI >> b) & 1u) << (kr & 31); }
            if (!real || !((qI >> b) & 1u) || ((eqb >> b) & 1u)) continue;          // only ACGT queries that differ from the majority
            if (rare_q[(size_t)i].empty() || rare_q[(size_t)i].back().word != (kr >> 5)) rare_q[(size_t)i].push_back({kr >> 5, 0u, 0u, 0u});
            RareWord &rw = rare_q[(size_t)i].back();
            rw.m |= 1u << (kr & 31); rw.l |= ((qL >> b) & 1u) << (kr & 31); rw.h |= ((qH >> b) & 1u) << (kr & 31);
          }
          qcv[(size_t)i * crow + (size_t)(w >> 2) * 8 + (w & 3)] = nI; qcv[(size_t)i * crow + (size_t)(w >> 2) * 8 + 4 + (w & 3)] = nV;
          uint32_t *fw = &flg[((size_t)(i / 16) * c->W4 + (w >> 2)) * 2];
          if (real && nI) fw[0] |= 1u << (i % 16);       // padding queries of the last tile are never read back: keep them "clean"
          if (real && nV) fw[0] |= 0x10000u << (i % 16);
          if ((w & 3) == 0) full = real;
          full = full && nI == cls[(size_t)w * 4 + 2] && nV == 0xFFFFFFFFu;
          if ((w & 3) == 3 && full) { fw[0] &= ~(0x10001u << (i % 16)); fw[1] |= 1u << (i % 16); }
        }
      } });
      for (int g = 0; g < c->W4; g++) {
        uint32_t u = 0;
        uint32_t uy = 0;
        for (int t = 0; t < c->nq_pad / 16; t++) { u |= flg[((size_t)t * c->W4 + g) * 2]; uy |= flg[((size_t)t * c->W4 + g) * 2 + 1]; }
        c->need_e_groups += (u & 0xFFFFu) != 0; c->need_v_groups += (u >> 16) != 0; c->need_g_groups += uy != 0;
      }
      // Next to a running scan (8 blocks x 16.9 KB of LDS per CU) a replay block with the 22 KB query row fits once per CU, without
      // it seven times: with many queries the replay then waits for LDS, not for work (5.48 -> 5.04 ms per config[1] search).
      if (c->replay_lq < 0) c->replay_lq = (c->nq < 256) ? 1 : 0;
      OPENCHK(hipMalloc(&c->d_cls, cls.size() * 4)); OPENCHK(hipMemcpy(c->d_cls, cls.data(), cls.size() * 4, hipMemcpyHostToDevice));
      OPENCHK(hipMalloc(&c->d_qrare, qrare.size() * 4)); OPENCHK(hipMemcpy(c->d_qrare, qrare.data(), qrare.size() * 4, hipMemcpyHostToDevice));
      OPENCHK(hipMalloc(&c->d_rmask, rmask.size() * 4)); OPENCHK(hipMemcpy(c->d_rmask, rmask.data(), rmask.size() * 4, hipMemcpyHostToDevice));
      {   // derive_all_kernel: word groups per wave and the bit positions its gathered columns start at
        int split[15];
        for (int v = 0; v <= 4; v++) split[v] = (int)((long long)c->W4 * v / 4);
        for (int v = 0; v <= 4; v++) {
          int nd = 0, nr = 0;
          for (int w = 0; w < split[v] * 4; w++) { nd += __builtin_popcount(cls[(size_t)w * 4 + 3]); nr += __builtin_popcount(rmask[(size_t)w]); }
          split[5 + v] = nd; split[10 + v] = nr;
        }
        OPENCHK(hipMalloc(&c->d_split, sizeof split)); OPENCHK(hipMemcpy(c->d_split, split, sizeof split, hipMemcpyHostToDevice));
      }
      OPENCHK(hipMalloc(&c->d_qpl, qpl.size() * 4)); OPENCHK(hipMemcpy(c->d_qpl, qpl.data(), qpl.size() * 4, hipMemcpyHostToDevice));
      // the item stream of every query tile (layout: see scan3_kernel)
      const int NWs = c->scan_NW, QS = 64;
      const uint32_t row_b = 256u * (uint32_t)c->scan_R;      // bytes of a query's counter row in a wave's LDS block: 64 lanes x R tiles x 4
      // one stream per super-tile of 64 queries (scan3_kernel: four waves share the counters and the stream)
      struct Rec { size_t at; uint32_t cost; };
      struct TileStream { std::vector<uint32_t> u; std::vector<Rec> rec, rare; };
      const int n_st = c->nq_pad / QS;
      std::vector<TileStream> ts((size_t)n_st);
      std::vector<uint8_t> rare_groups_needed((size_t)std::max(c->NR4, 1), 0);
      parallel_for(n_st, [&](int st) {
        TileStream &S = ts[(size_t)st];
        std::vector<uint32_t> &strm = S.u;
        std::vector<int> full, gen, wrd[4];
        for (int g = 0; g < c->W4; g++) {
          full.clear(); gen.clear(); for (auto &w : wrd) w.clear();
          for (int ql = 0; ql < QS; ql++) {
            const int q = st * QS + ql, t = q / 16, b = q % 16;
            const uint32_t fx = flg[((size_t)t * c->W4 + g) * 2], fy = flg[((size_t)t * c->W4 + g) * 2 + 1];
            if ((fy >> b) & 1u) { full.push_back(ql); continue; }
            if (!(((fx | (fx >> 16)) >> b) & 1u)) continue;
            // a dirty query whose non-ACGT / invalid sites of this group all lie in ONE 32-column word (an isolated N or ambiguity
            // code: more than half of the partially dirty cases) gets a 4-dword "word item" instead of the 12-dword general one
            const uint32_t *src = qcv.data() + (size_t)q * crow + (size_t)g * 8;
            int words = 0, last = 0;
            for (int j = 0; j < 4; j++) if (src[j] | src[4 + j]) { words++; last = j; }
            if (words == 1) wrd[last].push_back(ql); else gen.push_back(ql);
          }
          if (full.empty() && gen.empty() && wrd[0].empty() && wrd[1].empty() && wrd[2].empty() && wrd[3].empty()) continue;
          const size_t hdr = strm.size();
          const uint32_t n_full4 = (uint32_t)(full.size() + 3) / 4u;
          strm.push_back((uint32_t)g * 2048u);
          strm.push_back(n_full4 | (uint32_t)gen.size() << 16);
          strm.push_back((uint32_t)wrd[0].size() | (uint32_t)wrd[1].size() << 8 | (uint32_t)wrd[2].size() << 16 | (uint32_t)wrd[3].size() << 24);
          strm.push_back(0u);
          for (int ql : full) strm.push_back((uint32_t)ql * row_b);
          while (strm.size() & 3) strm.push_back((uint32_t)QS * row_b);                         // scratch row
          for (int ql : gen) {
            const uint32_t *src = qcv.data() + (size_t)(st * QS + ql) * crow + (size_t)g * 8;
            strm.insert(strm.end(), src, src + 8);
            strm.push_back((uint32_t)ql * row_b); strm.push_back(0u); strm.push_back(0u); strm.push_back(0u);
          }
          for (int j = 0; j < 4; j++)
            for (int ql : wrd[j]) {
              const uint32_t *src = qcv.data() + (size_t)(st * QS + ql) * crow + (size_t)g * 8;
              strm.push_back(src[j]); strm.push_back(src[4 + j]); strm.push_back((uint32_t)ql * row_b); strm.push_back(0u);
            }
          strm[hdr + 3] = (uint32_t)(strm.size() - hdr);
          S.rec.push_back({hdr, 60u + 4u * n_full4 + 30u * (uint32_t)gen.size() + 14u * (uint32_t)(wrd[0].size() + wrd[1].size() + wrd[2].size() + wrd[3].size())});
        }
        // the walk runs two headers ahead and takes a record's length from its header: a zero header of length 4 ends the group records
        strm.push_back(0u); strm.push_back(0u); strm.push_back(0u); strm.push_back(4u);
        strm.push_back(0u); strm.push_back(0u); strm.push_back(0u); strm.push_back(4u);
        // rare records: { byte offset of the rare group's planes in the tile's gathered planes, word-item counts (8 bits each), 0, 0 }
        // + items { sites, their lo bits, their hi bits, LDS offset } listed word by word
        for (int r4 = 0; r4 < c->NR4; r4++) {
          uint32_t nw[4] = {0, 0, 0, 0};
          for (int ql = 0; ql < QS; ql++) for (const RareWord &rw : rare_q[(size_t)st * QS + ql]) if ((rw.word >> 2) == r4) nw[rw.word & 3]++;
          if (!(nw[0] | nw[1] | nw[2] | nw[3])) continue;
          S.rare.push_back({strm.size(), nw[0] + nw[1] + nw[2] + nw[3]});
          strm.push_back((uint32_t)(c->NP4 + r4) * 3072u); strm.push_back(nw[0] | nw[1] << 8 | nw[2] << 16 | nw[3] << 24); strm.push_back(0u); strm.push_back(0u);
          for (int j = 0; j < 4; j++)
            for (int ql = 0; ql < QS; ql++) for (const RareWord &rw : rare_q[(size_t)st * QS + ql]) if (rw.word == r4 * 4 + j) {
              strm.push_back(rw.m); strm.push_back(rw.l); strm.push_back(rw.h); strm.push_back((uint32_t)ql * row_b);
            }
          rare_groups_needed[(size_t)r4] = 1;     // (a byte set to 1 by several threads)
        }
      });
      std::vector<uint32_t> strm, sdir((size_t)n_st * 4 * NWs, 0u);
      auto split4 = [NWs](const std::vector<Rec> &r, size_t base, size_t end_at, uint32_t *dir) {   // NW contiguous shares of about the same cost
        uint64_t total = 0;
        for (const Rec &x : r) total += x.cost;
        size_t i = 0; uint64_t done = 0;
        for (int w = 0; w < NWs; w++) {
          const size_t i0 = i;
          const uint64_t goal = total * (uint64_t)(w + 1) / (uint64_t)NWs;
          while (i < r.size() && (w == NWs - 1 || done + r[i].cost / 2 < goal)) { done += r[i].cost; i++; }
          dir[2 * w] = (uint32_t)(base + (i0 < r.size() ? r[i0].at : end_at));
          dir[2 * w + 1] = (uint32_t)(i - i0);
        }
      };
      for (int st = 0; st < n_st; st++) {
        const TileStream &S = ts[(size_t)st];
        const size_t base = strm.size(), zero_hdr = S.rec.empty() ? 0 : 0;
        (void)zero_hdr;
        // a wave without records still looks at two headers: point it at the zero headers that end the group records
        size_t end_at = S.u.size() - 8;
        for (const Rec &x : S.rare) { end_at = std::min(end_at, x.at - 8); break; }
        split4(S.rec, base, end_at, &sdir[(size_t)st * 4 * NWs]);
        split4(S.rare, base, S.u.size() - 8, &sdir[(size_t)st * 4 * NWs + 2 * NWs]);
        strm.insert(strm.end(), S.u.begin(), S.u.end());
      }
      for (uint8_t u : rare_groups_needed) c->need_r_groups += u;
      strm.resize(strm.size() + 64, 0u);                                // the kernel prefetches items and headers past the end
      OPENCHK(hipMalloc(&c->d_stream, strm.size() * 4)); OPENCHK(hipMemcpy(c->d_stream, strm.data(), strm.size() * 4, hipMemcpyHostToDevice));
      OPENCHK(hipMalloc(&c->d_sdir, sdir.size() * 4)); OPENCHK(hipMemcpy(c->d_sdir, sdir.data(), sdir.size() * 4, hipMemcpyHostToDevice));
    }
    OPENCHK(hipMalloc(&c->d_amb_q, ambq.size() * sizeof(int))); OPENCHK(hipMemcpy(c->d_amb_q, ambq.data(), ambq.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  {  // the queries restricted to query->idx are the query planes under the mask of those columns: kept as the mask (ensure_qpoly)
    std::vector<uint32_t> pmask((size_t)c->W4 * 4, 0u);
    for (int sidx = lo; sidx < hi; sidx++) if (in_p[(size_t)sidx]) pmask[(size_t)sidx >> 5] |= 1u << (sidx & 31);
    OPENCHK(hipMalloc(&c->d_pmask, pmask.size() * 4)); OPENCHK(hipMemcpy(c->d_pmask, pmask.data(), pmask.size() * 4, hipMemcpyHostToDevice));
    std::vector<int> cols;
    for (int sidx = lo; sidx < hi; sidx++) if (in_p[(size_t)sidx]) cols.push_back(sidx);
    c->n_idx = (int)cols.size(); c->NG4 = std::max(1, ((c->n_idx + 31) / 32 + 3) / 4);
    for (int v = 0; v <= 4; v++) {      // four shares of about the same number of columns, cut at word groups
      int g = 0, bits = 0;
      const int goal = (int)((long long)c->n_idx * v / 4);
      while (g < c->W4 && (v == 4 || bits < goal)) { for (int j = 0; j < 4; j++) bits += __builtin_popcount(pmask[(size_t)g * 4 + j]); g++; }
      c->ball_split[v] = v == 0 ? 0 : g; c->ball_split[5 + v] = v == 0 ? 0 : bits;
    }
    cols.resize(cols.size() + 1, 0);
    OPENCHK(hipMalloc(&c->d_idx_cols, cols.size() * sizeof(int))); OPENCHK(hipMemcpy(c->d_idx_cols, cols.data(), cols.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  OPENCHK(hipMalloc(&c->d_cp, cp.size() * 4)); OPENCHK(hipMemcpy(c->d_cp, cp.data(), cp.size() * 4, hipMemcpyHostToDevice));
  OPENCHK(hipMalloc(&c->d_cpm, cpm.size() * 4)); OPENCHK(hipMemcpy(c->d_cpm, cpm.data(), cpm.size() * 4, hipMemcpyHostToDevice));

  // ---- state
  OPENCHK(hipMalloc(&c->d_heap, (size_t)c->nq * (c->k + 1) * HEAP_ENTRY * sizeof(int)));
  OPENCHK(hipMalloc(&c->d_n, (size_t)c->nq * sizeof(int)));
  OPENCHK(hipMalloc(&c->d_T, (size_t)c->nq * sizeof(int)));
  OPENCHK(hipMalloc(&c->d_snap, sizeof(int)));
  OPENCHK(hipMalloc(&c->d_err, sizeof(int)));
  OPENCHK(hipMemset(c->d_err, 0, sizeof(int)));
  // ---- batch buffers
  // (the buffers of a streamed batch -- packed tiles, their derived planes, side rows: 25 KB per reference of max_pool -- are allocated
  // by the first call that streams sequences in: ensure_batch_buffers)
  if (!c->fullscan) { OPENCHK(hipMalloc(&c->d_cnt2, (size_t)c->nq_pad * c->pool_pad * sizeof(uint32_t))); c->slice_cap[0] = (size_t)c->nq_pad * c->pool_pad; }
  OPENCHK(hipMalloc(&c->d_tmin[0], (size_t)c->nq_pad * (c->pool_pad / 64) * sizeof(int2)));
  OPENCHK(hipMalloc(&c->d_rtb[0], c->pool_pad * sizeof(int4)));
  OPENCHK(hipMemset(c->d_rtb[0], 0, c->pool_pad * sizeof(int4)));
  OPENCHK(hipMalloc(&c->d_stats, 4 * sizeof(unsigned long long)));
  OPENCHK(hipMemset(c->d_stats, 0, 4 * sizeof(unsigned long long)));
  OPENCHK(hipMalloc(&c->d_rt, c->pool_pad * sizeof(int4)));
  OPENCHK(hipMalloc(&c->d_tr, c->pool_pad * sizeof(int4)));
  OPENCHK(hipMemset(c->d_rt, 0, c->pool_pad * sizeof(int4)));      // stay zero when idx_c is empty (the pre-score is skipped)
  OPENCHK(hipMemset(c->d_tr, 0, c->pool_pad * sizeof(int4)));
  OPENCHK(hipMalloc(&c->d_entered, c->pool_pad)); c->entered_cap = c->pool_pad;
  OPENCHK(hipMemset(c->d_entered, 0, c->pool_pad));
  OPENCHK(hipMalloc(&c->d_stage, (size_t)2 * PACK_CHUNK * c->pitch));
  OPENCHK(hipHostMalloc(&c->h_stage, (size_t)2 * PACK_CHUNK * c->pitch, hipHostMallocDefault));
  memset(c->h_stage, 'N', (size_t)2 * PACK_CHUNK * c->pitch);
  for (int i = 0; i < 2; i++) OPENCHK(hipEventCreateWithFlags(&c->stage_free[i], hipEventDisableTiming));
  const size_t lds = (size_t)(c->k + 1) * HEAP_ENTRY * sizeof(int) + 128;      // heap + the listed-words bitmap of replay2_kernel
  if (lds > 64 * 1024) {
    OPENCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&replay_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    OPENCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&replay_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    OPENCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&replay2_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    OPENCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&replay2_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    OPENCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&replay2_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    OPENCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&replay2_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#define BIGHEAP(A, B, PF_) OPENCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&replay2_kernel<A, B, PF_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
    BIGHEAP(true, true, 1); BIGHEAP(true, false, 1); BIGHEAP(false, true, 1); BIGHEAP(false, false, 1);
    BIGHEAP(true, true, 2); BIGHEAP(true, false, 2); BIGHEAP(false, true, 2); BIGHEAP(false, false, 2);
#undef BIGHEAP
  }
#undef OPENCHK
  int rc = uvaia_gpu_reset(c);
  if (rc) { g_open_error = c->err; uvaia_gpu_close(c); return rc; }
  *out = c;
  return UVAIA_GPU_OK;
}

int uvaia_gpu_reset(uvaia_gpu_ctx *c)
{
  if (!c) return UVAIA_GPU_EINVAL;
  HIPCHK(c, hipMemsetAsync(c->d_heap, 0, (size_t)c->nq * (c->k + 1) * HEAP_ENTRY * sizeof(int), c->stream));
  hipLaunchKernelGGL(init_state_kernel, dim3((c->nq + 255) / 256), dim3(256), 0, c->stream, c->d_T, c->d_n, c->nq, c->nchar);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->d_snap, &c->nchar, sizeof(int), hipMemcpyHostToDevice, c->stream));   // cq->max_incompatible = n_sites (src/nearest.c:375)
  if (c->d_entered && c->db_n) HIPCHK(c, hipMemsetAsync(c->d_entered, 0, ((c->db_n + 63) / 64) * 64, c->stream));
  for (int i_ = 0; i_ < 3; i_++) if (c->scan_streams[i_]) HIPCHK(c, hipStreamSynchronize(c->scan_streams[i_]));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int uvaia_gpu_heap_slots(const uvaia_gpu_ctx *c) { return c ? c->k : 0; }
int uvaia_gpu_n_query(const uvaia_gpu_ctx *c) { return c ? c->nq : 0; }
size_t uvaia_gpu_packed_bytes_per_ref(const uvaia_gpu_ctx *c) { return c ? (size_t)c->W4 * 16 * c->P : 0; }
size_t uvaia_gpu_scan_bytes_per_ref(const uvaia_gpu_ctx *c)
{ // distinct bytes of a reference the default scan has to read at least once: the word groups of the two derived planes that
  // some query tile needs (groups where every query is clean are never loaded) + three planes of the gathered polymorphic columns
  if (!c) return 0;
  return (c->fullscan || c->scan_variant != 2) ? (size_t)c->W4 * 16 * c->P
                                                : (size_t)(c->need_e_groups + c->need_v_groups) * 16 + (size_t)c->need_g_groups * 4 + (size_t)(c->NP4 + c->need_r_groups) * 16 * 3;
}

int uvaia_gpu_scan_variant(const uvaia_gpu_ctx *c)
{ // which pair scan this context runs: 2 column-compressed (scan3_kernel), 0 two counters over the packed planes (scan2_*_kernel;
  // default for at most 16 queries), 1 its LDS-broadcast form, -1 four counters (alignments above 49 000 columns)
  return !c ? -2 : c->fullscan ? -1 : c->scan_variant;
}

size_t uvaia_gpu_derived_bytes_per_ref(const uvaia_gpu_ctx *c)
{ // bytes per reference uvaia_gpu_db_rederive writes for the open query set (E, group counts, gathered columns, total)
  if (!c || c->fullscan || c->scan_variant != 2) return 0;
  return (size_t)c->W4 * 16 + (size_t)c->W4 * 4 + (size_t)(c->NP4 + c->NR4) * 48 + 4;   // E, grp, gathered planes, total; V is written once by the appends
}

int uvaia_gpu_set_query_tile(uvaia_gpu_ctx *c, int qt)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (qt == 0) qt = 16;
  if (qt != 8 && qt != 16 && qt != 32) return fail(c, UVAIA_GPU_EINVAL, "query tile must be 8, 16 or 32");
  c->qt = qt;
  return 0;
}

int uvaia_gpu_agree_on_polymorphic(uvaia_gpu_ctx *c, const char *const *seq, int n_seq, uint8_t *out)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (n_seq < 0 || (n_seq > 0 && (!seq || !out))) return fail(c, UVAIA_GPU_EINVAL, "bad batch");
  if ((size_t)n_seq > c->max_pool) return fail(c, UVAIA_GPU_ESTATE, "batch of %d exceeds max_pool %zu", n_seq, c->max_pool);
  if (n_seq == 0) return 0;
  int rc = ensure_batch_buffers(c); if (rc) return rc;
  rc = pack_rows(c, seq, nullptr, 0, nullptr, n_seq, c->d_batch, c->d_batch_nonn, c->d_batch_amb, c->d_batch_tot, 0);
  if (rc) return rc;
  const int n_tiles = (n_seq + 63) / 64, ppad = n_tiles * 64;
  rc = ensure_cnt4(c, (size_t)c->nq_pad * c->pool_pad); if (rc) return rc;
  const bool prof = c->profile; c->profile = false;     // not the nearest-neighbour scan the statistics describe
  rc = ensure_qpoly(c); if (rc) return rc;
  rc = launch_scan(c, c->d_batch, 0, n_tiles, c->d_qpoly, c->nq, c->d_cnt, ppad, 0.0);
  c->profile = prof;
  if (rc) return rc;
  uint8_t *d_out = nullptr;
  const size_t bytes = (size_t)n_seq * c->nq;
  HIPCHK(c, hipMalloc(&d_out, bytes));
  dim3 grid((n_seq + 255) / 256, c->nq);
  if (c->acgt) hipLaunchKernelGGL((agree_kernel<true>), grid, dim3(256), 0, c->stream, c->d_cnt, ppad, c->nq, n_seq, d_out);
  else         hipLaunchKernelGGL((agree_kernel<false>), grid, dim3(256), 0, c->stream, c->d_cnt, ppad, c->nq, n_seq, d_out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  hipFree(d_out);
  if (e != hipSuccess) return fail(c, UVAIA_GPU_EHIP, "agree_on_polymorphic: %s", hipGetErrorString(e));
  return 0;
}

// create_query_indices (src/fastaseq.c:732-777) needs no context: the query rows go to the device in batches through a pinned
// staging buffer (host copies threaded), two small kernels per batch, the per-column result comes back as two byte arrays.
int uvaia_gpu_query_columns(const char *const *seq, int n_query, int nchar, size_t trim, int acgt, int device, char *consensus, unsigned char *some_missing)
{
  if (!seq || !consensus || !some_missing || n_query < 1 || nchar < 1) return fail(nullptr, UVAIA_GPU_EINVAL, "empty query set");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, UVAIA_GPU_ENODEV, "no HIP device available: the MI355X engine has no CPU fallback");
  int caller_device = -1;
  if (hipGetDevice(&caller_device) != hipSuccess) caller_device = -1;
  if (device < 0) device = caller_device < 0 ? 0 : caller_device;
  if (device >= ndev || hipSetDevice(device) != hipSuccess) return fail(nullptr, UVAIA_GPU_ENODEV, "device %d is not usable", device);
  struct RestoreDevice { int d; ~RestoreDevice() { if (d >= 0) hipSetDevice(d); } } restore_{caller_device};   // no context: the caller's current device is left as it was
  const int lo = (int)std::min<size_t>(trim, (size_t)nchar), hi = std::max(lo, nchar - (int)std::min<size_t>(trim, (size_t)nchar));
  memset(consensus, 'N', (size_t)nchar);
  memset(some_missing, 0, (size_t)nchar);
  if (hi <= lo) return 0;
  const size_t pitch = ((size_t)nchar + 63) / 64 * 64;
  const int batch = (int)std::max<size_t>(64, std::min<size_t>(4096, ((size_t)96 << 20) / pitch) / 64 * 64);      // rows per batch: at most 96 MB of staging
  const int groups = batch / 64;
  uint8_t *h_rows = nullptr, *d_rows = nullptr, *d_pf = nullptr, *d_pl = nullptr, *d_first = nullptr, *d_flags = nullptr;
  hipStream_t st = nullptr;
  hipError_t e = hipSuccess;
  hipEvent_t freed[2] = {nullptr, nullptr};
  auto done = [&](int rc) { if (st) { hipStreamSynchronize(st); hipStreamDestroy(st); } for (int i = 0; i < 2; i++) if (freed[i]) hipEventDestroy(freed[i]);
                            hipFree(d_rows); hipFree(d_pf); hipFree(d_pl); hipFree(d_first); hipFree(d_flags); if (h_rows) hipHostFree(h_rows); return rc; };
#define QCHK(call) do { e = (call); if (e != hipSuccess) return done(fail(nullptr, e == hipErrorOutOfMemory ? UVAIA_GPU_ENOMEM : UVAIA_GPU_EHIP, "%s failed: %s", #call, hipGetErrorString(e))); } while (0)
  QCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  QCHK(hipHostMalloc(&h_rows, (size_t)2 * batch * pitch, hipHostMallocDefault));
  QCHK(hipMalloc(&d_rows, (size_t)2 * batch * pitch));
  QCHK(hipMalloc(&d_pf, (size_t)groups * nchar)); QCHK(hipMalloc(&d_pl, (size_t)groups * nchar));
  QCHK(hipMalloc(&d_first, (size_t)nchar)); QCHK(hipMalloc(&d_flags, (size_t)nchar));
  QCHK(hipMemsetAsync(d_first, 'N', (size_t)nchar, st)); QCHK(hipMemsetAsync(d_flags, 0, (size_t)nchar, st));
  for (int i = 0; i < 2; i++) QCHK(hipEventCreateWithFlags(&freed[i], hipEventDisableTiming));
  const unsigned gx = (unsigned)((hi - lo + 255) / 256);
  int k = 0;
  for (int a = 0; a < n_query; a += batch, k++) {
    const int m = std::min(batch, n_query - a), b = k & 1;
    if (k >= 2) QCHK(hipEventSynchronize(freed[b]));                 // the staging half's previous batch has been consumed
    uint8_t *hb = h_rows + (size_t)b * batch * pitch, *db = d_rows + (size_t)b * batch * pitch;
    for (int i = 0; i < m; i++) if (!seq[a + i]) return done(fail(nullptr, UVAIA_GPU_EINVAL, "query %d is NULL", a + i));
    parallel_for(m, [&](int i) { memcpy(hb + (size_t)i * pitch, seq[a + i], (size_t)nchar); });
    QCHK(hipMemcpyAsync(db, hb, (size_t)m * pitch, hipMemcpyHostToDevice, st));
    const int ng = (m + 63) / 64;
    hipLaunchKernelGGL(query_columns_partial_kernel, dim3(gx, (unsigned)ng), dim3(256), 0, st, db, pitch, m, lo, hi, acgt ? 1 : 0, d_pf, d_pl, nchar);
    hipLaunchKernelGGL(query_columns_merge_kernel, dim3(gx), dim3(256), 0, st, d_pf, d_pl, ng, lo, hi, nchar, d_first, d_flags);
    QCHK(hipGetLastError());
    QCHK(hipEventRecord(freed[b], st));
  }
  std::vector<uint8_t> first((size_t)nchar), flags((size_t)nchar);
  QCHK(hipMemcpyAsync(first.data(), d_first, (size_t)nchar, hipMemcpyDeviceToHost, st));
  QCHK(hipMemcpyAsync(flags.data(), d_flags, (size_t)nchar, hipMemcpyDeviceToHost, st));
  QCHK(hipStreamSynchronize(st));
#undef QCHK
  for (int c = lo; c < hi; c++) {
    consensus[c] = (flags[(size_t)c] & 1) ? '#' : (char)first[(size_t)c];
    some_missing[c] = (flags[(size_t)c] & 2) ? 1 : 0;
  }
  return done(0);
}

int uvaia_gpu_push(uvaia_gpu_ctx *c, const char *const *seq, const int *non_n, int n_ref, int64_t ordinal0, uint8_t *entered)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (n_ref < 0 || (n_ref > 0 && !seq)) return fail(c, UVAIA_GPU_EINVAL, "bad batch");
  if ((size_t)n_ref > c->max_pool) return fail(c, UVAIA_GPU_ESTATE, "batch of %d exceeds max_pool %zu", n_ref, c->max_pool);
  if (c->act_q0 != 0 || c->act_q1 != c->nq) return fail(c, UVAIA_GPU_ESTATE, "streamed batches act on the whole query set: query shards use the resident calls");
  if (n_ref == 0) return 0;
  int rc = ensure_batch_buffers(c); if (rc) return rc;
  rc = pack_rows(c, seq, nullptr, 0, non_n, n_ref, c->d_batch, c->d_batch_nonn, c->d_batch_amb, c->d_batch_tot, 0);
  if (rc) return rc;
  const int n_tiles = (n_ref + 63) / 64;
  HIPCHK(c, hipMemsetAsync(c->d_entered, 0, (size_t)n_tiles * 64, c->stream));
  rc = run_batch(c, c->d_batch, c->d_batch_nonn, c->d_batch_amb, 0, n_tiles, 0, n_ref, ordinal0, c->d_entered);
  if (rc) return rc;
  if (entered) HIPCHK(c, hipMemcpyAsync(entered, c->d_entered, (size_t)n_ref, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return collect_events(c);
}

int uvaia_gpu_drain(uvaia_gpu_ctx *c, int *n_items, int *max_incompatible, int *scores, int64_t *ordinals)
{
  if (!c || !n_items || !scores || !ordinals) return c ? fail(c, UVAIA_GPU_EINVAL, "NULL output") : UVAIA_GPU_EINVAL;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t ne = (size_t)c->nq * (c->k + 1);
  std::vector<int> h(ne * HEAP_ENTRY), T(c->nq);
  HIPCHK(c, hipMemcpy(h.data(), c->d_heap, h.size() * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(n_items, c->d_n, (size_t)c->nq * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(T.data(), c->d_T, (size_t)c->nq * sizeof(int), hipMemcpyDeviceToHost));
  for (size_t e = 0; e < ne; e++) {
    for (int s = 0; s < 6; s++) scores[e * 6 + s] = h[e * HEAP_ENTRY + s];
    ordinals[e] = (int64_t)(((uint64_t)(uint32_t)h[e * HEAP_ENTRY + 7] << 32) | (uint32_t)h[e * HEAP_ENTRY + 6]);
  }
  if (max_incompatible) memcpy(max_incompatible, T.data(), (size_t)c->nq * sizeof(int));
  return collect_events(c);
}

int uvaia_gpu_db_reserve(uvaia_gpu_ctx *c, size_t cap)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (cap <= c->db_cap) return 0;
  if (c->db_n) return fail(c, UVAIA_GPU_ESTATE, "reserve the database before appending to it");
  if (c->d_db) { hipFree(c->d_db); hipFree(c->d_db_nonn); hipFree(c->d_db_amb); hipFree(c->d_db_tot); hipFree(c->d_db_ev); hipFree(c->d_db_poly); hipFree(c->d_db_tote); hipFree(c->d_db_grp); c->d_db_grp = nullptr;
                 c->d_db = nullptr; c->d_db_nonn = nullptr; c->d_db_amb = nullptr; c->d_db_tot = nullptr; c->d_db_ev = c->d_db_poly = nullptr; c->d_db_tote = nullptr; }
  const size_t tiles = (cap + 63) / 64 + 1, tile_u4 = (size_t)c->W4 * c->P * 64;
  HIPCHK(c, hipMalloc(&c->d_db, tiles * tile_u4 * sizeof(uint4)));
  HIPCHK(c, hipMemset(c->d_db, 0, tiles * tile_u4 * sizeof(uint4)));
  HIPCHK(c, hipMalloc(&c->d_db_nonn, tiles * 64 * sizeof(int)));
  HIPCHK(c, hipMemset(c->d_db_nonn, 0, tiles * 64 * sizeof(int)));
  const size_t dtiles = derived_tiles(c, tiles);      // reference shards: derived planes for the owned pieces only
  HIPCHK(c, hipMalloc(&c->d_db_ev, dtiles * (size_t)c->W4 * 2 * 64 * sizeof(uint4)));
  HIPCHK(c, hipMalloc(&c->d_db_grp, dtiles * (size_t)c->W4 * 64 * sizeof(uint32_t)));
  HIPCHK(c, hipMalloc(&c->d_db_poly, dtiles * (size_t)std::max(c->NP4 + c->NR4, 1) * 3 * 64 * sizeof(uint4)));
  HIPCHK(c, hipMalloc(&c->d_db_tote, dtiles * 64 * sizeof(int)));
  HIPCHK(c, hipMalloc(&c->d_db_tot, tiles * 64 * sizeof(int)));
  HIPCHK(c, hipMemset(c->d_db_tot, 0, tiles * 64 * sizeof(int)));
  HIPCHK(c, hipMalloc(&c->d_db_amb, tiles * 64 * AMB_ROW * sizeof(int)));
  HIPCHK(c, hipMemset(c->d_db_amb, 0, tiles * 64 * AMB_ROW * sizeof(int)));
  c->db_cap = tiles * 64 - 64;
  if (c->entered_cap < tiles * 64) {
    hipFree(c->d_entered); c->d_entered = nullptr;
    HIPCHK(c, hipMalloc(&c->d_entered, tiles * 64)); c->entered_cap = tiles * 64;
  }
  HIPCHK(c, hipMemset(c->d_entered, 0, c->entered_cap));
  return 0;
}

static int db_append_common(uvaia_gpu_ctx *c, const char *const *seq, const char *rows, size_t pitch, const int *non_n, int n_ref)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (n_ref < 0) return fail(c, UVAIA_GPU_EINVAL, "negative count");
  if (n_ref == 0) return 0;
  { int rc = settle_derive(c); if (rc) return rc; }
  if (c->db_n + (size_t)n_ref > c->db_cap) {
    if (c->db_n) return fail(c, UVAIA_GPU_ESTATE, "database capacity %zu exceeded: call uvaia_gpu_db_reserve first", c->db_cap);
    int rc = uvaia_gpu_db_reserve(c, (size_t)n_ref); if (rc) return rc;
  }
  int rc = pack_rows(c, seq, rows, pitch, non_n, n_ref, c->d_db, c->d_db_nonn, c->d_db_amb, c->d_db_tot, (long long)c->db_n);
  if (rc) return rc;
  c->db_n += (size_t)n_ref;
  return 0;
}

int uvaia_gpu_db_append(uvaia_gpu_ctx *c, const char *const *seq, const int *non_n, int n_ref)
{ if (c && n_ref > 0 && !seq) return fail(c, UVAIA_GPU_EINVAL, "NULL seq"); return db_append_common(c, seq, nullptr, 0, non_n, n_ref); }

int uvaia_gpu_db_append_block(uvaia_gpu_ctx *c, const char *rows, size_t pitch, const int *non_n, int n_ref)
{
  if (c && n_ref > 0 && (!rows || pitch < (size_t)c->nchar)) return fail(c, UVAIA_GPU_EINVAL, "bad block");
  return db_append_common(c, nullptr, rows, pitch, non_n, n_ref);
}

size_t uvaia_gpu_db_size(const uvaia_gpu_ctx *c) { return c ? c->db_n : 0; }

int uvaia_gpu_db_clear(uvaia_gpu_ctx *c)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (!c->d_db || !c->db_n) { c->db_n = 0; return 0; }
  { int rc = settle_derive(c); if (rc) return rc; }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i_ = 0; i_ < 3; i_++) if (c->scan_streams[i_]) HIPCHK(c, hipStreamSynchronize(c->scan_streams[i_]));
  const size_t tiles = (c->db_n + 63) / 64;       // lanes past the last reference of a tile must read as zero planes
  HIPCHK(c, hipMemsetAsync(c->d_db, 0, tiles * (size_t)c->W4 * c->P * 64 * sizeof(uint4), c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_db_nonn, 0, tiles * 64 * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_db_tot, 0, tiles * 64 * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_db_amb, 0, tiles * 64 * AMB_ROW * sizeof(int), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->db_n = 0;
  return 0;
}

size_t uvaia_gpu_db_tile_bytes(const uvaia_gpu_ctx *c) { return c ? (size_t)c->W4 * 4 * 64 * sizeof(uint4) : 0; }
int uvaia_gpu_db_side_row_ints(void) { return AMB_ROW; }

int uvaia_gpu_db_export(uvaia_gpu_ctx *c, size_t first_tile, size_t n_tiles, void *planes, int *non_n, int *side_rows)
{
  if (!c || !planes || !non_n || !side_rows) return UVAIA_GPU_EINVAL;
  if (c->acgt) return fail(c, UVAIA_GPU_ESTATE, "the interchange form is the four IUPAC planes: export from a default-mode context");
  if ((first_tile + n_tiles) * 64 > ((c->db_n + 63) / 64) * 64) return fail(c, UVAIA_GPU_EINVAL, "tiles %zu..%zu lie outside the database", first_tile, first_tile + n_tiles);
  if (!n_tiles) return 0;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t tb = uvaia_gpu_db_tile_bytes(c);
  HIPCHK(c, hipMemcpy(planes, reinterpret_cast<const char *>(c->d_db) + first_tile * tb, n_tiles * tb, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(non_n, c->d_db_nonn + first_tile * 64, n_tiles * 64 * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(side_rows, c->d_db_amb + first_tile * 64 * AMB_ROW, n_tiles * 64 * AMB_ROW * sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}

int uvaia_gpu_db_append_packed(uvaia_gpu_ctx *c, const void *planes, const int *non_n, const int *side_rows, int n_ref)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (n_ref < 0) return fail(c, UVAIA_GPU_EINVAL, "negative count");
  if (n_ref == 0) return 0;
  if (!planes || !non_n || (!c->acgt && !side_rows)) return fail(c, UVAIA_GPU_EINVAL, "NULL packed arrays");
  if (c->db_n % 64) return fail(c, UVAIA_GPU_ESTATE, "packed tiles can only follow a whole number of tiles (database holds %zu references)", c->db_n);
  { int rc = settle_derive(c); if (rc) return rc; }
  if (c->db_n + (size_t)n_ref > c->db_cap) {
    if (c->db_n) return fail(c, UVAIA_GPU_ESTATE, "database capacity %zu exceeded: call uvaia_gpu_db_reserve first", c->db_cap);
    int rc = uvaia_gpu_db_reserve(c, (size_t)n_ref); if (rc) return rc;
  }
  const size_t tb = uvaia_gpu_db_tile_bytes(c), n_tiles = ((size_t)n_ref + 63) / 64;
  const long long t0 = (long long)(c->db_n / 64);
  if (!c->acgt) {     // same form as the resident planes: straight into place, then the totals
    HIPCHK(c, hipMemcpyAsync(reinterpret_cast<char *>(c->d_db) + (size_t)t0 * tb, planes, n_tiles * tb, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL((import_tiles_kernel<4>), dim3((unsigned)n_tiles), dim3(256), 0, c->stream, c->d_db + (size_t)t0 * c->W4 * 4 * 64, c->W4, (uint4 *)nullptr, t0, c->d_db_tot);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->d_db_amb + (size_t)t0 * 64 * AMB_ROW, side_rows, n_tiles * 64 * AMB_ROW * sizeof(int), hipMemcpyHostToDevice, c->stream));
  } else {            // re-code through a staging buffer, a few tiles at a time
    const size_t chunk = 64;
    uint4 *d_tmp = nullptr;
    HIPCHK(c, hipMalloc(&d_tmp, chunk * tb));
    for (size_t a = 0; a < n_tiles; a += chunk) {
      const size_t m = std::min(chunk, n_tiles - a);
      hipError_t e = hipMemcpyAsync(d_tmp, reinterpret_cast<const char *>(planes) + a * tb, m * tb, hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess) { hipLaunchKernelGGL((import_tiles_kernel<3>), dim3((unsigned)m), dim3(256), 0, c->stream, d_tmp, c->W4, c->d_db, t0 + (long long)a, c->d_db_tot); e = hipGetLastError(); }
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) { hipFree(d_tmp); return fail(c, UVAIA_GPU_EHIP, "import of packed tiles: %s", hipGetErrorString(e)); }
    }
    hipFree(d_tmp);
  }
  HIPCHK(c, hipMemcpyAsync(c->d_db_nonn + (size_t)t0 * 64, non_n, n_tiles * 64 * sizeof(int), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(sanitise_import_kernel, dim3((unsigned)((n_tiles * 64 + 255) / 256)), dim3(256), 0, c->stream, c->acgt ? (int *)nullptr : c->d_db_amb + (size_t)t0 * 64 * AMB_ROW,
                     c->d_db_nonn + (size_t)t0 * 64, (long long)(n_tiles * 64), c->W4 * 4, c->nchar);
  HIPCHK(c, hipGetLastError());
  int rc = derive_rows(c, c->d_db, (long long)c->db_n, n_ref); if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->db_n += (size_t)n_ref;
  return 0;
}

// Sub-slices of the pools that tile [first, first + n): {first reference, length, opens a pool}.  A pool boundary only retakes
// the snapshot of the tolerances (src/nearest.c:290-291), which happens at a pool's first sub-slice, so cutting pools is exact.
struct SubSlice { size_t first, n; bool pool_start; };
static std::vector<SubSlice> plan_subslices(const uvaia_gpu_ctx *c, size_t first, size_t n, size_t pool)
{
  std::vector<SubSlice> subs;
  // with few queries the replay is negligible and small launches only cost: one slice per pool then.  The sub-slice length is
  // tuned for 63 query tiles (1 000 queries); with fewer active tiles (query shards) it grows so that a launch still fills the chip
  const int nq_act = c->act_q1 - c->act_q0, nqt = (c->act_q1 + 15) / 16 - c->act_q0 / 16;
  // Pool boundaries act through the snapshot only, and the snapshot only through the consensus counters: without constant-and-
  // complete query columns (n_idx_c == 0) they have no effect at all and the slices are laid over the whole range.
  if (c->n_idx_c == 0) pool = std::max<size_t>(n, 1);
  size_t sub = c->subslice;
  // With at most half the benchmark's query tiles (query shards, smaller query sets) the rebuild of the derived planes weighs
  // more against the scan: slices of half the waves let the first scan start earlier and hide more of it (measured with the
  // rebuild inside the step: 6.18 -> 6.03, 7.24 -> 7.10, 9.89 -> 9.26 ms for rank 0 of 2, 4, 8 query shards; 63 tiles: worse).
  if (nqt <= 32) sub /= 2;
  if (nqt < 63) sub = std::min(pool, (sub * 63 / (size_t)std::max(nqt, 1) + 63) / 64 * 64);
  // At most three query tiles: the scan is bound by HBM and the replay has a handful of waves; what pays is running the replay
  // of one slice next to the scan of the following ones (the pre-score no longer waits for the batch snapshot: DESIGN.md 2.3):
  // four slices per pool, none below 65 536 references.
  if (nqt < 4) sub = std::min(pool, std::max<size_t>(65536, ((pool + 3) / 4 + 63) / 64 * 64));
  if (c->subslice_forced) sub = std::min(pool, c->subslice);
  (void)nq_act;
  for (size_t a = first; a < first + n; a += pool) {
    const size_t pe = std::min(first + n, a + pool);
    if (nqt < 4 && !c->subslice_forced && pe - a >= 8 * 65536) {
      // Few queries, a long pool: the replay of a pool's FIRST references is the expensive one (the heaps fill and turn over fast,
      // every admission a dependent round trip to memory for a handful of waves) and the replay of its LAST slice is exposed.  A
      // short head lets the first start early, next to the scans of the rest; a short tail keeps the exposed part small.
      const size_t head = 65536, tail = 65536, mid = pe - a - head - tail, nm = std::max<size_t>(1, (mid + sub / 2) / sub);
      const size_t each = ((mid + nm - 1) / nm + 63) / 64 * 64;
      subs.push_back({a, head, true});
      for (size_t x = a + head; x < pe - tail; x += each) subs.push_back({x, std::min(each, pe - tail - x), false});
      subs.push_back({pe - tail, tail, false});
      continue;
    }
    // near-equal slices (multiples of 64), as many as the pool holds sub-slice lengths, rounded: a pool of 1.05 sub-slices is
    // one launch, not a full one plus a sliver whose launch latency and replay would sit on the critical path
    const size_t len = pe - a, ns = std::max<size_t>(1, (len + sub / 2) / sub);
    const size_t each = ((len + ns - 1) / ns + 63) / 64 * 64;
    for (size_t x = a; x < pe; x += each) subs.push_back({x, std::min(each, pe - x), x == a});
  }
  return subs;
}

int uvaia_gpu_db_rederive(uvaia_gpu_ctx *c)
{ // the reference-side work a query set costs on a database that is already resident: E/V/grp planes and gathered columns of
  // every tile for the open query set.  Appends do this for the rows they add; a caller that times "one search of a resident
  // database" without its appends calls this first so that the figure holds everything that depends on the query set.
  // Issued on its own stream in the chunks the search will scan, one event each: the first slice's scan starts as soon as its
  // chunk is done and the rest is rebuilt next to it (the rebuild is bound by HBM, the scan by instruction issue).
  if (!c) return UVAIA_GPU_EINVAL;
  if (!c->d_db || !c->db_n || c->fullscan || c->scan_variant != 2) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  {   // searches still in flight read the planes: the rebuild queues behind them
    hipStream_t busy[4] = {c->stream, c->scan_streams[0], c->scan_streams[1], c->scan_streams[2]};
    for (int i = 0; i < 4; i++) {
      if (!busy[i]) continue;
      if (!c->derive_fence[i]) HIPCHK(c, hipEventCreateWithFlags(&c->derive_fence[i], hipEventDisableTiming));
      HIPCHK(c, hipEventRecord(c->derive_fence[i], busy[i]));
      for (int j = 0; j < c->derive_nstreams; j++) HIPCHK(c, hipStreamWaitEvent(c->derive_streams[j], c->derive_fence[i], 0));
    }
  }
  std::vector<SubSlice> plan = plan_subslices(c, 0, c->db_n, c->max_pool);
  size_t k = 0;
  long long t_done = 0;                 // slices that are not tile aligned share a tile: it belongs to the earlier chunk
  for (const SubSlice &sl : plan) {
    const long long t0 = std::max(t_done, (long long)(sl.first / 64)), t1 = (long long)((sl.first + sl.n + 63) / 64);
    if (t0 >= t1) continue;
    t_done = t1;
    if (k == c->derive_chunks.size()) {
      uvaia_gpu_ctx::DeriveChunk d{0, 0, nullptr};
      HIPCHK(c, hipEventCreateWithFlags(&d.done, hipEventDisableTiming));
      c->derive_chunks.push_back(d);
    }
    hipStream_t ds = c->derive_streams[k % (size_t)c->derive_nstreams];
    int rc = derive_rows(c, c->d_db, t0 * 64, (int)((t1 - t0) * 64), ds, true); if (rc) return rc;
    c->derive_chunks[k].t0 = t0; c->derive_chunks[k].t1 = t1;
    HIPCHK(c, hipEventRecord(c->derive_chunks[k].done, ds));
    k++;
  }
  c->derive_pending = k;
  return 0;
}

// Two streams and a ring of NBUF counter buffers: the scan needs no state, so it runs up to NBUF-1 slices ahead of the replay.
// snapshot >= 0: the first pool's snapshot is given (query shards: the maximum over all ranks); only valid for a single pool.
static int run_subslices(uvaia_gpu_ctx *c, const std::vector<SubSlice> &subs, int64_t ordinal_of_db0, int snapshot)
{
  const size_t ns = subs.size();
  size_t issued = 0;
  {   // a launch of fewer waves than ~2 rounds of the chip's 8 192 wave slots leaves it half empty at start and end: let such
      // launches of consecutive slices overlap (they write different buffers)
    const int nqt = (c->act_q1 + 15) / 16 - c->act_q0 / 16;
    const size_t waves = ns ? (size_t)nqt * ((subs[0].n + 63) / 64) : 0;
    // (a single query tile: the scan is bound by HBM, launches next to each other only slow each other down)
    c->scan_nstreams = (waves && waves < 30000 && nqt >= 4) ? 3 : 1;
  }
  for (size_t i = 0; i < ns; i++) {
    const bool serial_ = c->serial;
    while (issued < ns && issued < i + (serial_ ? 1 : NBUF)) {          // keep the scan stream fed
      int rc = uvaia_gpu_slice_scan(c, subs[issued].first, subs[issued].n, (int)(issued % NBUF));
      if (rc) return rc;
      issued++;
      if (serial_) for (int i_ = 0; i_ < 3; i_++) hipStreamSynchronize(c->scan_streams[i_]);
    }
    int take = subs[i].pool_start ? 1 : 0;
    if (take && snapshot >= 0) { HIPCHK(c, hipMemcpyAsync(c->d_snap, &snapshot, sizeof(int), hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); take = 0; }
    int rc = uvaia_gpu_slice_replay(c, (int)(i % NBUF), ordinal_of_db0 + (long long)subs[i].first, take);
    if (rc) return rc;
    if (serial_) hipStreamSynchronize(c->stream);
  }
  return 0;
}

int uvaia_gpu_search_resident(uvaia_gpu_ctx *c, size_t pool, int64_t ordinal0, uint8_t *entered)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (pool < 1 || pool > c->max_pool) return fail(c, UVAIA_GPU_EINVAL, "pool must be in [1, max_pool=%zu]", c->max_pool);
  if (!c->db_n) return 0;
  HIPCHK(c, hipMemsetAsync(c->d_entered, 0, ((c->db_n + 63) / 64) * 64, c->stream));
  if (!c->fullscan) {
    int rc = run_subslices(c, plan_subslices(c, 0, c->db_n, pool), ordinal0, -1);
    if (rc) return rc;
  } else
  for (size_t a = 0; a < c->db_n; a += pool) {
    const size_t b = std::min(c->db_n, a + pool);
    const long long tf = (long long)(a / 64);
    const int n_tiles = (int)((b + 63) / 64 - a / 64);
    const int rb = (int)(a - (size_t)tf * 64), re = (int)(b - (size_t)tf * 64);
    int rc = run_batch(c, c->d_db, c->d_db_nonn + tf * 64, c->d_db_amb + tf * 64 * AMB_ROW, tf, n_tiles, rb, re, ordinal0 + (long long)a, c->d_entered + tf * 64);
    if (rc) return rc;
  }
  if (entered) {
    HIPCHK(c, hipMemcpyAsync(entered, c->d_entered, c->db_n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return collect_events(c);
  }
  return 0;
}

int uvaia_gpu_search_resident_pool(uvaia_gpu_ctx *c, size_t first, size_t n, int64_t ordinal0, int snapshot)
{ // one batch ("pool") [first, first + n) of the resident database; entered flags accumulate (uvaia_gpu_entered_flags)
  if (!c) return UVAIA_GPU_EINVAL;
  if (c->fullscan) return fail(c, UVAIA_GPU_ESTATE, "per-pool search needs the default scan");
  if (n < 1 || n > c->max_pool || first + n > c->db_n) return fail(c, UVAIA_GPU_EINVAL, "pool [%zu,+%zu) outside the database or above max_pool=%zu", first, n, c->max_pool);
  return run_subslices(c, plan_subslices(c, first, n, n), ordinal0 - (int64_t)first, snapshot);
}

int uvaia_gpu_sync(uvaia_gpu_ctx *c)
{
  if (!c) return UVAIA_GPU_EINVAL;
  for (int i_ = 0; i_ < 3; i_++) if (c->derive_streams[i_]) HIPCHK(c, hipStreamSynchronize(c->derive_streams[i_]));
  c->derive_pending = 0;
  for (int i_ = 0; i_ < 3; i_++) if (c->scan_streams[i_]) HIPCHK(c, hipStreamSynchronize(c->scan_streams[i_]));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return collect_events(c);
}

int uvaia_gpu_last_batch_scores(uvaia_gpu_ctx *c, int *out, int n_ref)
{
  if (!c || !out) return UVAIA_GPU_EINVAL;
  if (n_ref != c->last_n || !c->last_nonn) return fail(c, UVAIA_GPU_ESTATE, "last batch held %d references, not %d", c->last_n, n_ref);
  int *d_out = nullptr;
  const size_t bytes = (size_t)n_ref * c->nq * 6 * sizeof(int);
  if (!c->fullscan) {   // the production path keeps two counters per pair: recount the batch with the four-counter kernel
    int rc = ensure_cnt4(c, (size_t)c->nq_pad * c->last_ppad); if (rc) return rc;
    const bool prof = c->profile; c->profile = false;
    rc = launch_scan(c, c->last_tiles, c->last_tile_first, c->last_ntiles, c->d_qp, c->nq, c->d_cnt, c->last_ppad, 0.0);
    c->profile = prof;
    if (rc) return rc;
  }
  HIPCHK(c, hipMalloc(&d_out, bytes));
  dim3 grid((n_ref + 255) / 256, c->nq);
  if (c->acgt) hipLaunchKernelGGL((batch_scores_kernel<true>), grid, dim3(256), 0, c->stream, c->d_cnt, c->last_ppad, c->last_rt, c->last_nonn, c->last_rbegin, n_ref, c->nq, d_out);
  else         hipLaunchKernelGGL((batch_scores_kernel<false>), grid, dim3(256), 0, c->stream, c->d_cnt, c->last_ppad, c->last_rt, c->last_nonn, c->last_rbegin, n_ref, c->nq, d_out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  hipFree(d_out);
  if (e != hipSuccess) return fail(c, UVAIA_GPU_EHIP, "batch_scores: %s", hipGetErrorString(e));
  return 0;
}

int uvaia_gpu_scan_stats(uvaia_gpu_ctx *c, double *ms, long long *launches, double *bytes, int reset)
{
  if (!c) return UVAIA_GPU_EINVAL;
  int rc = collect_events(c); if (rc) return rc;
  if (ms) *ms = c->scan_ms;
  if (launches) *launches = c->scan_launches;
  if (bytes) *bytes = c->scan_bytes;
  if (reset) { c->scan_ms = 0; c->scan_bytes = 0; c->scan_launches = 0; }
  return 0;
}

int uvaia_gpu_replay_stats(uvaia_gpu_ctx *c, unsigned long long out[3], int reset)
{
  if (!c || !out) return UVAIA_GPU_EINVAL;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  unsigned long long h[3] = {0, 0, 0};
  HIPCHK(c, hipMemcpy(h, c->d_stats, sizeof h, hipMemcpyDeviceToHost));
  out[0] = h[0]; out[1] = h[1]; out[2] = h[2];
  if (reset) HIPCHK(c, hipMemset(c->d_stats, 0, sizeof h));
  return 0;
}

// ---- ring mode (multi-GPU, DESIGN.md "Multi-GPU"): the database is dealt block-cyclically, every rank scans its slice
// of a stripe concurrently, and the small per-query state travels rank to rank so that each query still sees the
// references in stream order.  state blob = snapshot, n[q], T[q], heap[q][k+1][8]  (all int32).
size_t uvaia_gpu_state_range_bytes(const uvaia_gpu_ctx *c, int q0, int q1)
{
  if (!c || q0 < 0 || q1 > c->nq || q1 < q0) return 0;
  const size_t nqr = (size_t)(q1 - q0);
  return sizeof(int) * (4 + 2 * nqr + nqr * (c->k + 1) * HEAP_ENTRY);
}
size_t uvaia_gpu_state_bytes(const uvaia_gpu_ctx *c) { return c ? uvaia_gpu_state_range_bytes(c, 0, c->nq) : 0; }

int uvaia_gpu_state_export_range(uvaia_gpu_ctx *c, void *dst, int q0, int q1)
{ // dst: device (or host) memory of uvaia_gpu_state_range_bytes(); ordered on the replay stream, then waited for
  if (!c || !dst || q0 < 0 || q1 > c->nq || q1 < q0) return UVAIA_GPU_EINVAL;
  int *d = (int *)dst;
  const size_t nqr = (size_t)(q1 - q0), he = (size_t)(c->k + 1) * HEAP_ENTRY;
  HIPCHK(c, hipMemcpyAsync(d, c->d_snap, sizeof(int), hipMemcpyDefault, c->stream));
  if (nqr) {
    HIPCHK(c, hipMemcpyAsync(d + 4, c->d_n + q0, nqr * sizeof(int), hipMemcpyDefault, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + 4 + nqr, c->d_T + q0, nqr * sizeof(int), hipMemcpyDefault, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + 4 + 2 * nqr, c->d_heap + (size_t)q0 * he, nqr * he * sizeof(int), hipMemcpyDefault, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int uvaia_gpu_state_import_range(uvaia_gpu_ctx *c, const void *src, int q0, int q1)
{
  if (!c || !src || q0 < 0 || q1 > c->nq || q1 < q0) return UVAIA_GPU_EINVAL;
  const int *d = (const int *)src;
  const size_t nqr = (size_t)(q1 - q0), he = (size_t)(c->k + 1) * HEAP_ENTRY;
  HIPCHK(c, hipMemcpyAsync(c->d_snap, d, sizeof(int), hipMemcpyDefault, c->stream));
  if (nqr) {
    HIPCHK(c, hipMemcpyAsync(c->d_n + q0, d + 4, nqr * sizeof(int), hipMemcpyDefault, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_T + q0, d + 4 + nqr, nqr * sizeof(int), hipMemcpyDefault, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_heap + (size_t)q0 * he, d + 4 + 2 * nqr, nqr * he * sizeof(int), hipMemcpyDefault, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));     // src may be reused or freed once this returns
  return 0;
}

int uvaia_gpu_state_export(uvaia_gpu_ctx *c, void *dst) { return c ? uvaia_gpu_state_export_range(c, dst, 0, c->nq) : UVAIA_GPU_EINVAL; }
int uvaia_gpu_state_import(uvaia_gpu_ctx *c, const void *src) { return c ? uvaia_gpu_state_import_range(c, src, 0, c->nq) : UVAIA_GPU_EINVAL; }

// counts of database references [first, first+n) into counter buffer `buf` (0/1), asynchronously on the scan stream
int uvaia_gpu_slice_scan(uvaia_gpu_ctx *c, size_t first, size_t n, int buf)
{
  if (!c || buf < 0 || buf >= NBUF) return UVAIA_GPU_EINVAL;
  if (c->fullscan) return fail(c, UVAIA_GPU_ESTATE, "ring mode needs the two-counter scan");
  if (first + n > c->db_n) return fail(c, UVAIA_GPU_EINVAL, "slice [%zu,+%zu) outside the database", first, n);
  // a slice is at most a pool when the batch snapshot can matter (n_idx_c > 0); otherwise pools have no effect and slices are free
  if (n > c->max_pool && c->n_idx_c > 0) return fail(c, UVAIA_GPU_EINVAL, "slice of %zu references above max_pool %zu", n, c->max_pool);
  {
    const size_t ppad_ = ((first + n + 63) / 64 - first / 64) * 64;
    // the scans write whole query tiles up to the last active one: super-tiles of 64 queries (scan3_kernel), tiles of 16 otherwise
    const size_t qtile = c->scan_variant == 2 ? 64 : 16;
    const size_t rows = std::min<size_t>((size_t)c->nq_pad, ((size_t)c->act_q1 + qtile - 1) / qtile * qtile);
    const size_t need = std::max(rows * ppad_, buf == 0 ? c->slice_cap[0] : (size_t)0);
    if (need > c->slice_cap[buf] || !c->d_tmin[buf]) {
      for (int i_ = 0; i_ < 3; i_++) if (c->scan_streams[i_]) HIPCHK(c, hipStreamSynchronize(c->scan_streams[i_]));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      const size_t cap = std::max(need, (size_t)c->nq_pad * c->pool_pad);
      uint32_t *&cb = buf ? c->d_cntb[buf] : c->d_cnt2;
      if (cap > c->slice_cap[buf] || !cb) { if (cb) hipFree(cb); cb = nullptr; HIPCHK(c, hipMalloc(&cb, cap * sizeof(uint32_t))); }
      if (c->d_tmin[buf]) hipFree(c->d_tmin[buf]);
      c->d_tmin[buf] = nullptr;
      HIPCHK(c, hipMalloc(&c->d_tmin[buf], (cap / 64) * sizeof(int2)));
      if (c->d_rtb[buf]) hipFree(c->d_rtb[buf]);
      c->d_rtb[buf] = nullptr;
      HIPCHK(c, hipMalloc(&c->d_rtb[buf], (std::max(cap / (size_t)c->nq_pad, ppad_) + 64) * sizeof(int4)));
      c->slice_cap[buf] = cap;
    }
  }
  hipStream_t ss = c->scan_streams[c->scan_nstreams > 1 ? (c->scan_rr++ % c->scan_nstreams) : 0];
  if (c->replay_recorded[buf]) HIPCHK(c, hipStreamWaitEvent(ss, c->replay_done[buf], 0));   // the buffer's previous reader
  for (size_t k = 0; k < c->derive_pending; k++) {                                          // planes being rebuilt (uvaia_gpu_db_rederive)
    const auto &d = c->derive_chunks[k];
    if (d.t0 < (long long)((first + n + 63) / 64) && d.t1 > (long long)(first / 64)) HIPCHK(c, hipStreamWaitEvent(ss, d.done, 0));
  }
  const long long tf = (long long)(first / 64);
  const int n_tiles = n ? (int)((first + n + 63) / 64 - first / 64) : 0;
  c->slice_tf[buf] = tf; c->slice_tiles[buf] = n_tiles;
  c->slice_rb[buf] = (int)(first - (size_t)tf * 64); c->slice_re[buf] = c->slice_rb[buf] + (int)n;
  c->slice_scanned[buf] = true; c->slice_cons_done[buf] = false;
  const double bytes = (double)n * (double)c->W4 * 16.0 * c->P + (double)c->nq * (double)c->W4 * 16.0 * c->P;
  int rc = launch_scan2(c, c->d_db, c->d_db_tot + tf * 64, tf, n_tiles, buf ? c->d_cntb[buf] : c->d_cnt2, n_tiles * 64, bytes, ss, c->d_tmin[buf], c->slice_rb[buf], c->slice_re[buf], c->d_rtb[buf]);
  if (rc) return rc;
  HIPCHK(c, hipEventRecord(c->scan_done[buf], ss));
  return 0;
}

// gate + heaps of queries [q0,q1) over the slice scanned into `buf`, from the state currently held (imported or local).
// take_snapshot != 0: this call opens a batch for the whole query set, so the batch snapshot (cq->max_incompatible,
// src/nearest.c:290-291) is taken from the state of ALL queries now held; otherwise the imported snapshot is used.
int uvaia_gpu_slice_replay_range(uvaia_gpu_ctx *c, int buf, int64_t ordinal0, int q0, int q1, int take_snapshot)
{
  if (!c || buf < 0 || buf >= NBUF) return UVAIA_GPU_EINVAL;
  if (!c->slice_scanned[buf]) return fail(c, UVAIA_GPU_ESTATE, "slice_replay without slice_scan");
  if (q0 < 0 || q1 > c->nq || q1 < q0) return fail(c, UVAIA_GPU_EINVAL, "bad query range [%d,%d)", q0, q1);
  const int n_tiles = c->slice_tiles[buf], rb = c->slice_rb[buf], re = c->slice_re[buf];
  const long long tf = c->slice_tf[buf];
  if (take_snapshot) { hipLaunchKernelGGL(snapshot_kernel, dim3(1), dim3(256), 0, c->stream, c->d_T + c->act_q0, c->act_q1 - c->act_q0, c->d_snap); c->slice_cons_done[buf] = false; }
  if (re <= rb || q1 == q0) return 0;
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->scan_done[buf], 0));
  const int ppad = n_tiles * 64;
  const size_t lds = (size_t)(c->k + 1) * HEAP_ENTRY * sizeof(int);
  const int lq_words = (c->replay_lq && !c->acgt && lds + (size_t)c->W4 * 4 * 6 * 4 + 128 <= 64 * 1024) ? c->W4 * 4 * 6 : 0;
  const uint32_t *cnt = buf ? c->d_cntb[buf] : c->d_cnt2;
  const int *nonn = c->d_db_nonn + tf * 64, *amb = c->d_db_amb + tf * 64 * AMB_ROW;
  uint8_t *ent = c->d_entered + tf * 64;
#define REPLAY2P(A, B, PF_) hipLaunchKernelGGL((replay2_kernel<A, B, PF_>), dim3(q1 - q0), dim3(64), lds + (size_t)lq_words * 4 + 128, c->stream, cnt, ppad, c->d_rtb[buf], c->d_cp, nonn, amb, rb, re, (long long)ordinal0, \
                                  c->d_heap, c->d_n, c->d_T, c->d_snap, ent, c->k, c->d_db, tf, c->W4, c->d_qp, c->d_amb_q, c->d_stats, q0, (c->scan_variant == 2 || c->scan_variant == 0) ? c->d_tmin[buf] : (const int2 *)nullptr, \
                                  (c->scan_variant == 2 && c->shard_world == 1) ? c->d_qpl : (const uint32_t *)nullptr, lq_words, c->replay_prio, c->d_db_poly, c->NP4 + c->NR4, c->NP4, c->NR4, c->d_qrare)
  // Candidates of a tile whose on-demand counters are requested ahead.  The bookkeeping of the request slots costs more than the
  // latency it hides (measured on one box: config[1] 3.69 / 3.54 / 3.60 ms per step with 3 / 2 / 1, 4 queries x 1 M references
  // 4.37 / 4.03 / 3.89; with 6 or 8 over 7 ms): two for large query sets, one -- request, then use -- for a handful of queries.
  const int pf = (q1 - q0) <= 64 ? 1 : 2;
#define REPLAY2(A, B) { if (pf == 1) REPLAY2P(A, B, 1); else REPLAY2P(A, B, 2); }
  if (c->acgt) { if (c->n_idx_c > 0) REPLAY2(true, true) else REPLAY2(true, false) }
  else         { if (c->n_idx_c > 0) REPLAY2(false, true) else REPLAY2(false, false) }
#undef REPLAY2P
#undef REPLAY2
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->replay_done[buf], c->stream));
  c->replay_recorded[buf] = true;
  c->last_tiles = c->d_db; c->last_nonn = nonn; c->last_n = re - rb; c->last_rbegin = rb; c->last_ppad = ppad; c->last_ntiles = n_tiles; c->last_tile_first = tf; c->last_rt = c->d_rtb[buf];
  return 0;
}

int uvaia_gpu_slice_buffers(void) { return NBUF; }

int uvaia_gpu_slice_replay(uvaia_gpu_ctx *c, int buf, int64_t ordinal0, int stripe_start)
{ return c ? uvaia_gpu_slice_replay_range(c, buf, ordinal0, c->act_q0, c->act_q1, stripe_start) : UVAIA_GPU_EINVAL; }

int uvaia_gpu_set_active_queries(uvaia_gpu_ctx *c, int q0, int q1)
{
  if (!c) return UVAIA_GPU_EINVAL;
  // (with reference shards the range only selects whose tolerances uvaia_gpu_max_tolerance looks at -- every scan covers all queries --
  // and may start anywhere; a range that is scanned starts at a super-tile of 64 queries)
  if (q0 < 0 || q1 > c->nq || q1 <= q0 || ((q0 % 64) && c->shard_world == 1))
    return fail(c, UVAIA_GPU_EINVAL, "active queries [%d,%d): need 0 <= q0 < q1 <= %d and q0 a multiple of 64", q0, q1, c->nq);
  if ((c->fullscan || c->scan_variant != 2) && c->shard_world == 1) { if (q0 != 0 || q1 != c->nq) return fail(c, UVAIA_GPU_ESTATE, "query shards need the default scan"); }
  c->act_q0 = q0; c->act_q1 = q1;
  return 0;
}

int uvaia_gpu_max_tolerance(uvaia_gpu_ctx *c, int *out)
{ // max over the active queries of max_incompatible: a rank's contribution to the batch snapshot (src/nearest.c:290-291)
  if (!c || !out) return UVAIA_GPU_EINVAL;
  int *d_tmp = nullptr;
  HIPCHK(c, hipMalloc(&d_tmp, sizeof(int)));
  hipLaunchKernelGGL(snapshot_kernel, dim3(1), dim3(256), 0, c->stream, c->d_T + c->act_q0, c->act_q1 - c->act_q0, d_tmp);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_tmp, sizeof(int), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  hipFree(d_tmp);
  if (e != hipSuccess) return fail(c, UVAIA_GPU_EHIP, "max_tolerance: %s", hipGetErrorString(e));
  return 0;
}

int uvaia_gpu_entered_flags(uvaia_gpu_ctx *c, uint8_t *out, int clear)
{ // "entered any heap" flags of the resident database accumulated by slice replays (and by search_resident)
  if (!c) return UVAIA_GPU_EINVAL;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (out && c->db_n) HIPCHK(c, hipMemcpy(out, c->d_entered, c->db_n, hipMemcpyDeviceToHost));
  if (clear && c->db_n) HIPCHK(c, hipMemset(c->d_entered, 0, ((c->db_n + 63) / 64) * 64));
  return 0;
}

// ---- reference shards (several GPUs; DESIGN.md "Multi-GPU").  Every context holds the packed planes of ALL references (what the
// replay's on-demand counters read; query-independent, loaded once) but derives and scans only its share: the stream is dealt in
// pieces of piece_refs references (whole tiles), piece p belongs to rank p % world.  A rank scans a piece against ALL queries into
// caller-owned buffers, the caller moves the rows of each query shard to the rank that replays those queries (RCCL all-to-all
// between processes, peer copies inside one: uvaia_gpu_group_*), and every rank replays its queries over the pieces in stream order.
int uvaia_gpu_db_set_shard(uvaia_gpu_ctx *c, int rank, int world, size_t piece_refs)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (world < 1 || rank < 0 || rank >= world) return fail(c, UVAIA_GPU_EINVAL, "rank %d of %d", rank, world);
  if (world > 1 && (piece_refs < 64 || piece_refs % 64 || piece_refs > c->max_pool)) return fail(c, UVAIA_GPU_EINVAL, "a piece holds a whole number of tiles of 64 references, at most max_pool = %zu (got %zu)", c->max_pool, piece_refs);
  if (c->d_db || c->db_n) return fail(c, UVAIA_GPU_ESTATE, "the reference shard is set before the database is reserved");
  if (world > 1 && c->fullscan) return fail(c, UVAIA_GPU_ESTATE, "reference shards need the two-counter scans (alignments up to 49 000 columns)");
  c->shard_rank = rank; c->shard_world = world; c->shard_pt = world > 1 ? (long long)(piece_refs / 64) : 0;
  return 0;
}

int uvaia_gpu_shard_rows(const uvaia_gpu_ctx *c) { return c ? c->nq_pad : 0; }

// Pair counters of the references [first, first + n) (inside one piece of this context's shard) against ALL queries:
//   cnt  [uvaia_gpu_shard_rows()][tiles * 64] int2,  tmin [uvaia_gpu_shard_rows()][tiles] int2,  tiles = the tiles the range touches;
// a reference sits in column (its position - 64 * (first / 64)).  Asynchronous; uvaia_gpu_scan_wait() waits for the scans issued so far.
int uvaia_gpu_shard_scan(uvaia_gpu_ctx *c, size_t first, size_t n, void *cnt, void *tmin)
{
  if (!c || !cnt || !tmin) return c ? fail(c, UVAIA_GPU_EINVAL, "NULL buffer") : UVAIA_GPU_EINVAL;
  if (c->fullscan) return fail(c, UVAIA_GPU_ESTATE, "reference shards need the two-counter scans");
  if (n < 1 || first + n > c->db_n) return fail(c, UVAIA_GPU_EINVAL, "range [%zu,+%zu) outside the database", first, n);
  const long long tf = (long long)(first / 64);
  const int n_tiles = (int)((first + n + 63) / 64 - first / 64);
  if ((size_t)n_tiles * 64 > c->pool_pad) return fail(c, UVAIA_GPU_EINVAL, "range of %zu references above max_pool %zu", n, c->max_pool);
  hipStream_t ss = c->scan_streams[0];
  for (size_t k = 0; k < c->derive_pending; k++) {
    const auto &d = c->derive_chunks[k];
    if (d.t0 < tf + n_tiles && d.t1 > tf) HIPCHK(c, hipStreamWaitEvent(ss, d.done, 0));
  }
  const int rb = (int)(first - (size_t)tf * 64);
  const double bytes = (double)n * (double)c->W4 * 16.0 * c->P + (double)c->nq * (double)c->W4 * 16.0 * c->P;
  return launch_scan2(c, c->d_db, c->d_db_tot + tf * 64, tf, n_tiles, (uint32_t *)cnt, n_tiles * 64, bytes, ss, (int2 *)tmin, rb, rb + (int)n, c->d_rtb[0]);
}

int uvaia_gpu_scan_wait(uvaia_gpu_ctx *c)
{
  if (!c) return UVAIA_GPU_EINVAL;
  for (int i_ = 0; i_ < 3; i_++) if (c->scan_streams[i_]) HIPCHK(c, hipStreamSynchronize(c->scan_streams[i_]));
  return 0;
}

// waits for the replays issued so far (their counter buffers may then be overwritten); scans keep running
int uvaia_gpu_replay_wait(uvaia_gpu_ctx *c)
{
  if (!c) return UVAIA_GPU_EINVAL;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ---- ordering against a stream the caller owns (the stream its collectives run on) by events, without involving the host: what lets
// the reference-shard driver queue scan -> exchange -> replay of stripe after stripe asynchronously (uvaia_amd/refshard.py).
// A mark = the scans (or replays) issued so far, remembered under a slot number; a caller stream can be made to wait for a mark, and
// the engine's later scans (or replays) for what a caller stream holds now.
int uvaia_gpu_mark(uvaia_gpu_ctx *c, int what, int slot)
{
  if (!c || slot < 0 || slot >= 8 || (what != UVAIA_GPU_SCANS && what != UVAIA_GPU_REPLAYS)) return c ? fail(c, UVAIA_GPU_EINVAL, "mark: what = scans or replays, slot 0..7") : UVAIA_GPU_EINVAL;
  for (int i = 0; i < 3; i++) {
    hipEvent_t &e = c->mark_ev[slot][i];
    hipStream_t st = what == UVAIA_GPU_SCANS ? c->scan_streams[i] : (i == 0 ? c->stream : nullptr);
    c->mark_set[slot][i] = false;
    if (!st) continue;
    if (!e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(e, st));
    c->mark_set[slot][i] = true;
  }
  return 0;
}
int uvaia_gpu_stream_wait_mark(uvaia_gpu_ctx *c, void *stream, int slot)
{
  if (!c || slot < 0 || slot >= 8) return c ? fail(c, UVAIA_GPU_EINVAL, "slot 0..7") : UVAIA_GPU_EINVAL;
  for (int i = 0; i < 3; i++) if (c->mark_set[slot][i]) HIPCHK(c, hipStreamWaitEvent((hipStream_t)stream, c->mark_ev[slot][i], 0));
  return 0;
}
int uvaia_gpu_wait_stream(uvaia_gpu_ctx *c, int what, void *stream)
{
  if (!c || (what != UVAIA_GPU_SCANS && what != UVAIA_GPU_REPLAYS)) return c ? fail(c, UVAIA_GPU_EINVAL, "wait_stream: what = scans or replays") : UVAIA_GPU_EINVAL;
  hipEvent_t &e = c->order_ev[c->order_rr++ % 16];
  if (!e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(e, (hipStream_t)stream));
  if (what == UVAIA_GPU_REPLAYS) HIPCHK(c, hipStreamWaitEvent(c->stream, e, 0));
  else for (int i = 0; i < 3; i++) if (c->scan_streams[i]) HIPCHK(c, hipStreamWaitEvent(c->scan_streams[i], e, 0));
  return 0;
}

// cq->max_incompatible of the batch that starts now (src/nearest.c:290-291), given by the caller: the maximum over the ranks of
// uvaia_gpu_max_tolerance().  Only matters when the query set has constant-and-complete columns.
int uvaia_gpu_set_snapshot(uvaia_gpu_ctx *c, int snapshot)
{
  if (!c) return UVAIA_GPU_EINVAL;
  HIPCHK(c, hipMemcpyAsync(c->d_snap, &snapshot, sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// Gate + heaps of queries [q0, q1) over references [first, first + n) of the resident database, from counters laid out as
// uvaia_gpu_shard_scan writes them but holding the rows of queries q0 .. q1-1 only (row 0 = query q0).  Asynchronous on the
// replay stream; the buffers must stay valid and complete until uvaia_gpu_sync().
int uvaia_gpu_shard_replay(uvaia_gpu_ctx *c, const void *cnt, const void *tmin, size_t first, size_t n, int64_t ordinal0, int q0, int q1)
{
  if (!c || !cnt || !tmin) return c ? fail(c, UVAIA_GPU_EINVAL, "NULL buffer") : UVAIA_GPU_EINVAL;
  if (c->fullscan) return fail(c, UVAIA_GPU_ESTATE, "reference shards need the two-counter scans");
  if (q0 < 0 || q1 > c->nq || q1 < q0) return fail(c, UVAIA_GPU_EINVAL, "bad query range [%d,%d)", q0, q1);
  if (first + n > c->db_n) return fail(c, UVAIA_GPU_EINVAL, "range [%zu,+%zu) outside the database", first, n);
  if (n < 1 || q1 == q0) return 0;
  const long long tf = (long long)(first / 64);
  const int n_tiles = (int)((first + n + 63) / 64 - first / 64), ppad = n_tiles * 64;
  if ((size_t)ppad > c->pool_pad) return fail(c, UVAIA_GPU_EINVAL, "range of %zu references above max_pool %zu", n, c->max_pool);
  const int rb = (int)(first - (size_t)tf * 64), re = rb + (int)n;
  if (c->n_idx_c > 0) {   // the pre-score of the piece's references, from the packed planes this rank holds of every reference
    if (c->acgt) hipLaunchKernelGGL((consensus_rt_kernel<true>), dim3((n_tiles + 3) / 4), dim3(256), 0, c->stream, c->d_db, tf, n_tiles, c->W4, c->d_cp, c->d_rt);
    else         hipLaunchKernelGGL((consensus_rt_kernel<false>), dim3((n_tiles + 3) / 4), dim3(256), 0, c->stream, c->d_db, tf, n_tiles, c->W4, c->d_cp, c->d_rt);
  }
  const size_t lds = (size_t)(c->k + 1) * HEAP_ENTRY * sizeof(int);
  const int lq_words = (c->replay_lq && !c->acgt && lds + (size_t)c->W4 * 4 * 6 * 4 + 128 <= 64 * 1024) ? c->W4 * 4 * 6 : 0;
  // the kernel indexes rows by query number: shift the bases so that row q0 is the buffer's first row
  const uint32_t *cntp = (const uint32_t *)cnt - (ptrdiff_t)q0 * ppad;
  const int2 *tminp = (const int2 *)tmin - (ptrdiff_t)q0 * (ppad / 64);
  const int *nonn = c->d_db_nonn + tf * 64, *amb = c->d_db_amb + tf * 64 * AMB_ROW;
  uint8_t *ent = c->d_entered + tf * 64;
  // --acgt: dist_unique of the pairs that reach a heap is counted from the packed planes (the scan's per-pair count stays on the scanning rank)
#define REPLAY2(A, B) hipLaunchKernelGGL((replay2_kernel<A, B>), dim3(q1 - q0), dim3(64), lds + (size_t)lq_words * 4 + 128, c->stream, cntp, ppad, c->d_rt, c->d_cp, nonn, amb, rb, re, (long long)ordinal0, \
                                  c->d_heap, c->d_n, c->d_T, c->d_snap, ent, c->k, c->d_db, tf, c->W4, c->d_qp, c->d_amb_q, c->d_stats, q0, tminp, (const uint32_t *)nullptr, lq_words, c->replay_prio, \
                                  (const uint4 *)nullptr, c->NP4 + c->NR4, c->NP4, 0, (const uint32_t *)nullptr)
  if (c->acgt) { if (c->n_idx_c > 0) REPLAY2(true, true); else REPLAY2(true, false); }
  else         { if (c->n_idx_c > 0) REPLAY2(false, true); else REPLAY2(false, false); }
#undef REPLAY2
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---- a group of contexts in ONE process (the C command line's --devices): the reference-shard protocol above with peer copies as
// the exchange.  Replaces the batch loop of src/nearest.c:245-330 for several GPUs.  One host thread drives all devices: every
// step is an asynchronous launch or copy, ordered by events across the devices' streams.
struct uvaia_gpu_group {
  int n = 0, nq = 0, rows = 0, cons = 0;
  size_t piece = 0;
  std::vector<uvaia_gpu_ctx *> ctx;
  std::vector<int> q0, q1;                      // query shard of each member (multiples of 16)
  struct Member {
    uint32_t *send_cnt[2] = {}, *recv_cnt[2] = {};
    int2 *send_tmin[2] = {}, *recv_tmin[2] = {};
    hipStream_t copy = nullptr;
    hipEvent_t scanned[2] = {}, fetched[2] = {}, replayed[2] = {};    // scan into send[b] done; this member's copies out of everyone's send[b] done; replays from recv[b] done
    bool fetched_rec[2] = {}, replayed_rec[2] = {};
  };
  std::vector<Member> m;
  std::string err;
};

static int gfail(uvaia_gpu_group *g, int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (g) g->err = buf; else g_open_error = buf;
  return code;
}
#define GCHK(g, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return gfail((g), UVAIA_GPU_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define GCTX(g, i, call) do { int rc_ = (call); if (rc_) return gfail((g), rc_, "device %d: %s", (g)->ctx[(size_t)(i)]->device, uvaia_gpu_last_error((g)->ctx[(size_t)(i)])); } while (0)

const char *uvaia_gpu_group_last_error(const uvaia_gpu_group *g) { return g ? g->err.c_str() : g_open_error.c_str(); }
int uvaia_gpu_group_size(const uvaia_gpu_group *g) { return g ? g->n : 0; }
uvaia_gpu_ctx *uvaia_gpu_group_member(uvaia_gpu_group *g, int i) { return (g && i >= 0 && i < g->n) ? g->ctx[(size_t)i] : nullptr; }

void uvaia_gpu_group_close(uvaia_gpu_group *g)
{
  if (!g) return;
  for (int i = 0; i < (int)g->ctx.size(); i++) {
    if (!g->ctx[(size_t)i]) continue;
    hipSetDevice(g->ctx[(size_t)i]->device);
    uvaia_gpu_sync(g->ctx[(size_t)i]);
    if (i < (int)g->m.size()) {
      auto &mm = g->m[(size_t)i];
      if (mm.copy) { hipStreamSynchronize(mm.copy); hipStreamDestroy(mm.copy); }
      for (int b = 0; b < 2; b++) {
        void *p[] = {mm.send_cnt[b], mm.send_tmin[b], mm.recv_cnt[b], mm.recv_tmin[b]};
        for (void *x : p) if (x) hipFree(x);
        hipEvent_t ev[] = {mm.scanned[b], mm.fetched[b], mm.replayed[b]};
        for (hipEvent_t e : ev) if (e) hipEventDestroy(e);
      }
    }
    uvaia_gpu_close(g->ctx[(size_t)i]);
  }
  delete g;
}

// devices[i]: HIP device of member i (a device may be listed more than once: several contexts on one GPU).  piece_refs: references per
// piece of the shard map (multiple of 64, at most max_pool); 0 = max_pool rounded down to whole tiles, at most 8 192.
int uvaia_gpu_group_open(uvaia_gpu_group **out, const uvaia_gpu_query *q, int heap_size, const int *devices, int n_devices, size_t max_pool, size_t piece_refs)
{
  if (!out) return gfail(nullptr, UVAIA_GPU_EINVAL, "group is NULL");
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > 64) return gfail(nullptr, UVAIA_GPU_EINVAL, "1 to 64 devices");
  if (max_pool < 64 && n_devices > 1) return gfail(nullptr, UVAIA_GPU_EINVAL, "several devices need max_pool >= 64");
  uvaia_gpu_group *g = new uvaia_gpu_group();
  g->n = n_devices;
  g->piece = n_devices == 1 ? 0 : (piece_refs ? piece_refs : std::min<size_t>(8192, max_pool / 64 * 64));
  if (g->n > 1 && (g->piece < 64 || g->piece % 64 || g->piece > max_pool)) { delete g; return gfail(nullptr, UVAIA_GPU_EINVAL, "piece of %zu references: need a multiple of 64 up to max_pool", piece_refs); }
  g->ctx.assign((size_t)g->n, nullptr);
  for (int i = 0; i < g->n; i++) {
    int rc = uvaia_gpu_open(&g->ctx[(size_t)i], q, heap_size, devices[i], max_pool);
    if (rc) { uvaia_gpu_group_close(g); return rc; }                      // message already in the open error
    rc = uvaia_gpu_db_set_shard(g->ctx[(size_t)i], i, g->n, g->piece);
    if (rc) { gfail(nullptr, rc, "%s", uvaia_gpu_last_error(g->ctx[(size_t)i])); uvaia_gpu_group_close(g); return rc; }
  }
  g->nq = q->n_query; g->rows = g->ctx[0]->nq_pad; g->cons = q->n_idx_c > 0;
  {   // contiguous query shards, whole query tiles
    int per = (g->nq + g->n - 1) / g->n; per = (per + 15) / 16 * 16;
    for (int i = 0; i < g->n; i++) { const int a = std::min(g->nq, i * per); g->q0.push_back(a); g->q1.push_back(std::min(g->nq, a + per)); }
  }
  g->m.resize((size_t)g->n);
  if (g->n > 1) {
    const size_t pcols = g->piece + 64;                      // a range that starts inside a tile touches one tile more
    for (int i = 0; i < g->n; i++) {
      auto &mm = g->m[(size_t)i];
      const size_t myrows = (size_t)std::max(1, g->q1[(size_t)i] - g->q0[(size_t)i]);
      hipError_t e = hipSetDevice(g->ctx[(size_t)i]->device);
      if (e == hipSuccess) e = hipStreamCreateWithFlags(&mm.copy, hipStreamNonBlocking);
      for (int b = 0; b < 2 && e == hipSuccess; b++) {
        e = hipMalloc(&mm.send_cnt[b], (size_t)g->rows * pcols * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&mm.send_tmin[b], (size_t)g->rows * (pcols / 64) * sizeof(int2));
        if (e == hipSuccess) e = hipMalloc(&mm.recv_cnt[b], (size_t)g->n * myrows * pcols * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&mm.recv_tmin[b], (size_t)g->n * myrows * (pcols / 64) * sizeof(int2));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&mm.scanned[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&mm.fetched[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&mm.replayed[b], hipEventDisableTiming);
      }
      if (e != hipSuccess) { const int code = gfail(nullptr, e == hipErrorOutOfMemory ? UVAIA_GPU_ENOMEM : UVAIA_GPU_EHIP, "group buffers on device %d: %s", devices[i], hipGetErrorString(e)); uvaia_gpu_group_close(g); return code; }
      for (int j = 0; j < g->n; j++) if (devices[j] != devices[i]) { int can = 0; if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) { hipError_t pe = hipDeviceEnablePeerAccess(devices[j], 0); if (pe == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError(); } }
    }
  }
  *out = g;
  return 0;
}

int uvaia_gpu_group_query_shard(const uvaia_gpu_group *g, int i, int *q0, int *q1)
{ if (!g || i < 0 || i >= g->n || !q0 || !q1) return UVAIA_GPU_EINVAL; *q0 = g->q0[(size_t)i]; *q1 = g->q1[(size_t)i]; return 0; }

#define EACH_MEMBER(g, expr) do { for (int i_ = 0; i_ < (g)->n; i_++) { GCHK(g, hipSetDevice((g)->ctx[(size_t)i_]->device)); uvaia_gpu_ctx *cx = (g)->ctx[(size_t)i_]; GCTX(g, i_, (expr)); } } while (0)
int uvaia_gpu_group_db_reserve(uvaia_gpu_group *g, size_t cap) { if (!g) return UVAIA_GPU_EINVAL; EACH_MEMBER(g, uvaia_gpu_db_reserve(cx, cap)); return 0; }
int uvaia_gpu_group_db_clear(uvaia_gpu_group *g) { if (!g) return UVAIA_GPU_EINVAL; EACH_MEMBER(g, uvaia_gpu_db_clear(cx)); return 0; }
int uvaia_gpu_group_reset(uvaia_gpu_group *g) { if (!g) return UVAIA_GPU_EINVAL; EACH_MEMBER(g, uvaia_gpu_reset(cx)); return 0; }
int uvaia_gpu_group_sync(uvaia_gpu_group *g)
{
  if (!g) return UVAIA_GPU_EINVAL;
  for (int i = 0; i < g->n; i++) { GCHK(g, hipSetDevice(g->ctx[(size_t)i]->device)); if (g->m[(size_t)i].copy) GCHK(g, hipStreamSynchronize(g->m[(size_t)i].copy)); GCTX(g, i, uvaia_gpu_sync(g->ctx[(size_t)i])); }
  return 0;
}
// every member receives the packed planes of every reference (the replay reads them); each derives the planes of its own pieces
int uvaia_gpu_group_db_append(uvaia_gpu_group *g, const char *const *seq, const int *non_n, int n_ref) { if (!g) return UVAIA_GPU_EINVAL; EACH_MEMBER(g, uvaia_gpu_db_append(cx, seq, non_n, n_ref)); return 0; }
int uvaia_gpu_group_db_append_packed(uvaia_gpu_group *g, const void *planes, const int *non_n, const int *side_rows, int n_ref)
{ if (!g) return UVAIA_GPU_EINVAL; EACH_MEMBER(g, uvaia_gpu_db_append_packed(cx, planes, non_n, side_rows, n_ref)); return 0; }
int uvaia_gpu_group_db_rederive(uvaia_gpu_group *g) { if (!g) return UVAIA_GPU_EINVAL; EACH_MEMBER(g, uvaia_gpu_db_rederive(cx)); return 0; }
size_t uvaia_gpu_group_db_size(const uvaia_gpu_group *g) { return g ? uvaia_gpu_db_size(g->ctx[0]) : 0; }
#undef EACH_MEMBER

// The whole while-loop of src/nearest.c:249-330 over the resident database in batches of `pool` references, sharded: per batch
// the snapshot of the tolerances is the maximum over ALL members' heaps (src/nearest.c:290-291; only taken when the query set has
// constant-and-complete columns, otherwise batches have no effect); the pieces of the shard map inside the batch are scanned by
// their owners, `n` pieces (one per member) at a time, the rows of every query shard are copied to the member that replays them,
// and each member replays its queries over the pieces in stream order.  Asynchronous unless entered != NULL.
int uvaia_gpu_group_search_resident(uvaia_gpu_group *g, size_t pool, int64_t ordinal0, uint8_t *entered)
{
  if (!g) return UVAIA_GPU_EINVAL;
  const size_t total = uvaia_gpu_db_size(g->ctx[0]);
  if (g->n == 1) {
    GCHK(g, hipSetDevice(g->ctx[0]->device));
    GCTX(g, 0, uvaia_gpu_search_resident(g->ctx[0], pool, ordinal0, entered));
    return 0;
  }
  if (pool < 1) return gfail(g, UVAIA_GPU_EINVAL, "pool must be positive");
  for (int i = 0; i < g->n; i++) {
    if (uvaia_gpu_db_size(g->ctx[(size_t)i]) != total) return gfail(g, UVAIA_GPU_ESTATE, "members hold different databases");
    GCHK(g, hipSetDevice(g->ctx[(size_t)i]->device));
    GCHK(g, hipMemsetAsync(g->ctx[(size_t)i]->d_entered, 0, ((total + 63) / 64) * 64, g->ctx[(size_t)i]->stream));
  }
  if (!total) return 0;
  if (!g->cons) pool = total;                          // batches act through the snapshot only (see plan_subslices)
  struct Piece { size_t first, n; int owner; };
  unsigned stripe_no = 0;
  for (size_t a = 0; a < total; a += pool) {
    const size_t b = std::min(total, a + pool);
    if (g->cons) {   // the batch snapshot: needs every member's tolerances as the previous batch left them
      int snap = -0x7fffffff;
      for (int i = 0; i < g->n; i++) { GCHK(g, hipSetDevice(g->ctx[(size_t)i]->device)); uvaia_gpu_ctx *cx = g->ctx[(size_t)i]; const int a0 = cx->act_q0, a1 = cx->act_q1;
        if (g->q1[(size_t)i] > g->q0[(size_t)i]) { cx->act_q0 = g->q0[(size_t)i]; cx->act_q1 = g->q1[(size_t)i]; int v = 0; const int rc = uvaia_gpu_max_tolerance(cx, &v); cx->act_q0 = a0; cx->act_q1 = a1; GCTX(g, i, rc); snap = std::max(snap, v); } }
      for (int i = 0; i < g->n; i++) { GCHK(g, hipSetDevice(g->ctx[(size_t)i]->device)); GCTX(g, i, uvaia_gpu_set_snapshot(g->ctx[(size_t)i], snap)); }
    }
    std::vector<Piece> pieces;                          // the parts of the shard map's pieces inside [a, b), in stream order
    for (size_t x = a; x < b;) { const size_t pe = std::min(b, (x / g->piece + 1) * g->piece); pieces.push_back({x, pe - x, (int)((x / g->piece) % (size_t)g->n)}); x = pe; }
    for (size_t s0 = 0; s0 < pieces.size(); s0 += (size_t)g->n, stripe_no++) {
      const size_t s1 = std::min(pieces.size(), s0 + (size_t)g->n);
      const int bsel = (int)(stripe_no & 1u);
      // 1. scans, each on its owner (consecutive pieces have distinct owners); a send buffer is reused once everyone has fetched from it
      for (size_t k = s0; k < s1; k++) {
        const Piece &pc = pieces[k]; uvaia_gpu_ctx *cx = g->ctx[(size_t)pc.owner]; auto &mo = g->m[(size_t)pc.owner];
        GCHK(g, hipSetDevice(cx->device));
        for (int d = 0; d < g->n; d++) if (g->m[(size_t)d].fetched_rec[bsel]) GCHK(g, hipStreamWaitEvent(cx->scan_streams[0], g->m[(size_t)d].fetched[bsel], 0));
        GCTX(g, pc.owner, uvaia_gpu_shard_scan(cx, pc.first, pc.n, mo.send_cnt[bsel], mo.send_tmin[bsel]));
        GCHK(g, hipEventRecord(mo.scanned[bsel], cx->scan_streams[0]));
      }
      // 2. every member fetches the rows of its queries from every owner, 3. and replays them in stream order
      for (int d = 0; d < g->n; d++) {
        auto &md = g->m[(size_t)d]; uvaia_gpu_ctx *cd = g->ctx[(size_t)d];
        const size_t myrows = (size_t)(g->q1[(size_t)d] - g->q0[(size_t)d]);
        GCHK(g, hipSetDevice(cd->device));
        if (md.replayed_rec[bsel]) GCHK(g, hipStreamWaitEvent(md.copy, md.replayed[bsel], 0));      // the receive buffer's previous readers
        for (size_t k = s0; k < s1 && myrows; k++) {
          const Piece &pc = pieces[k]; auto &mo = g->m[(size_t)pc.owner];
          const size_t tiles = (pc.first + pc.n + 63) / 64 - pc.first / 64, ppad = tiles * 64, slot = k - s0;
          GCHK(g, hipStreamWaitEvent(md.copy, mo.scanned[bsel], 0));
          GCHK(g, hipMemcpyPeerAsync(md.recv_cnt[bsel] + slot * myrows * (g->piece + 64), cd->device, mo.send_cnt[bsel] + (size_t)g->q0[(size_t)d] * ppad, g->ctx[(size_t)pc.owner]->device,
                                     myrows * ppad * sizeof(uint32_t), md.copy));
          GCHK(g, hipMemcpyPeerAsync(md.recv_tmin[bsel] + slot * myrows * ((g->piece + 64) / 64), cd->device, mo.send_tmin[bsel] + (size_t)g->q0[(size_t)d] * tiles, g->ctx[(size_t)pc.owner]->device,
                                     myrows * tiles * sizeof(int2), md.copy));
        }
        GCHK(g, hipEventRecord(md.fetched[bsel], md.copy)); md.fetched_rec[bsel] = true;
        GCHK(g, hipStreamWaitEvent(cd->stream, md.fetched[bsel], 0));
        for (size_t k = s0; k < s1 && myrows; k++) {
          const Piece &pc = pieces[k]; const size_t slot = k - s0;
          GCTX(g, d, uvaia_gpu_shard_replay(cd, md.recv_cnt[bsel] + slot * myrows * (g->piece + 64), md.recv_tmin[bsel] + slot * myrows * ((g->piece + 64) / 64), pc.first, pc.n,
                                            ordinal0 + (int64_t)pc.first, g->q0[(size_t)d], g->q1[(size_t)d]));
        }
        GCHK(g, hipEventRecord(md.replayed[bsel], cd->stream)); md.replayed_rec[bsel] = true;
      }
    }
  }
  if (entered) {   // a reference is dumped if it entered the heap of ANY query (src/nearest.c:303-306): OR over the members
    std::vector<uint8_t> part(total);
    memset(entered, 0, total);
    for (int i = 0; i < g->n; i++) {
      GCHK(g, hipSetDevice(g->ctx[(size_t)i]->device));
      if (g->m[(size_t)i].copy) GCHK(g, hipStreamSynchronize(g->m[(size_t)i].copy));
      GCTX(g, i, uvaia_gpu_sync(g->ctx[(size_t)i]));
      GCTX(g, i, uvaia_gpu_entered_flags(g->ctx[(size_t)i], part.data(), 0));
      for (size_t x = 0; x < total; x++) entered[x] |= part[x];
    }
  }
  return 0;
}

// one batch of the reference loop, sequences from host memory (uvaia_gpu_push for the group): every member packs the batch, the
// scan of its pieces is shared out as above
int uvaia_gpu_group_push(uvaia_gpu_group *g, const char *const *seq, const int *non_n, int n_ref, int64_t ordinal0, uint8_t *entered)
{
  if (!g) return UVAIA_GPU_EINVAL;
  if (n_ref < 0 || (n_ref > 0 && !seq)) return gfail(g, UVAIA_GPU_EINVAL, "bad batch");
  if (n_ref == 0) return 0;
  if (g->n == 1) { GCHK(g, hipSetDevice(g->ctx[0]->device)); GCTX(g, 0, uvaia_gpu_push(g->ctx[0], seq, non_n, n_ref, ordinal0, entered)); return 0; }
  int rc = uvaia_gpu_group_sync(g); if (rc) return rc;
  rc = uvaia_gpu_group_db_clear(g); if (rc) return rc;
  rc = uvaia_gpu_group_db_append(g, seq, non_n, n_ref); if (rc) return rc;
  std::vector<uint8_t> ent((size_t)n_ref);
  rc = uvaia_gpu_group_search_resident(g, (size_t)n_ref, ordinal0, entered ? entered : ent.data());
  return rc;
}

// heaps of all queries, each from the member that replays it (arrays as uvaia_gpu_drain)
int uvaia_gpu_group_drain(uvaia_gpu_group *g, int *n_items, int *max_incompatible, int *scores, int64_t *ordinals)
{
  if (!g || !n_items || !scores || !ordinals) return UVAIA_GPU_EINVAL;
  if (g->n == 1) { GCHK(g, hipSetDevice(g->ctx[0]->device)); GCTX(g, 0, uvaia_gpu_drain(g->ctx[0], n_items, max_incompatible, scores, ordinals)); return 0; }
  const size_t slots = (size_t)uvaia_gpu_heap_slots(g->ctx[0]) + 1;
  std::vector<int> n((size_t)g->nq), T((size_t)g->nq), sc((size_t)g->nq * slots * 6);
  std::vector<int64_t> od((size_t)g->nq * slots);
  for (int i = 0; i < g->n; i++) {
    GCHK(g, hipSetDevice(g->ctx[(size_t)i]->device));
    if (g->m[(size_t)i].copy) GCHK(g, hipStreamSynchronize(g->m[(size_t)i].copy));
    GCTX(g, i, uvaia_gpu_drain(g->ctx[(size_t)i], n.data(), T.data(), sc.data(), od.data()));
    for (int q = g->q0[(size_t)i]; q < g->q1[(size_t)i]; q++) {
      n_items[q] = n[(size_t)q];
      if (max_incompatible) max_incompatible[q] = T[(size_t)q];
      memcpy(scores + (size_t)q * slots * 6, sc.data() + (size_t)q * slots * 6, slots * 6 * sizeof(int));
      memcpy(ordinals + (size_t)q * slots, od.data() + (size_t)q * slots, slots * sizeof(int64_t));
    }
  }
  return 0;
}

// radius search over references [r_lo, r_hi) (relative to tile tile_first of `tiles`): stage 1 for all, the queries for the few
static int ball_range(uvaia_gpu_ctx *c, const uint4 *tiles, long long tile_first, int n_tiles, int r_lo, int r_hi, int radius, int *mindist_host)
{
  const int n = r_hi - r_lo;
  if (n <= 0) return 0;
  if (!c->d_mindist || c->ball_cap < (size_t)n_tiles * 64) {
    if (c->d_mindist) { hipFree(c->d_mindist); hipFree(c->d_ball_list); hipFree(c->d_ball_cdist); c->d_mindist = nullptr; }
    c->ball_cap = std::max<size_t>((size_t)n_tiles * 64, c->pool_pad);
    HIPCHK(c, hipMalloc(&c->d_mindist, c->ball_cap * sizeof(int)));
    HIPCHK(c, hipMalloc(&c->d_ball_list, c->ball_cap * sizeof(int)));
    HIPCHK(c, hipMalloc(&c->d_ball_cdist, c->ball_cap * sizeof(int)));
    if (!c->d_ball_n) HIPCHK(c, hipMalloc(&c->d_ball_n, sizeof(int)));
  }
  HIPCHK(c, hipMemsetAsync(c->d_ball_n, 0, sizeof(int), c->stream));
  if (c->acgt) hipLaunchKernelGGL((ball_stage1_kernel<true>), dim3((n_tiles + 3) / 4), dim3(256), 0, c->stream, tiles, tile_first, n_tiles, c->W4, c->d_cp, c->d_cpm, radius, r_lo, r_hi, c->d_mindist, c->d_ball_cdist, c->d_ball_list, c->d_ball_n);
  else         hipLaunchKernelGGL((ball_stage1_kernel<false>), dim3((n_tiles + 3) / 4), dim3(256), 0, c->stream, tiles, tile_first, n_tiles, c->W4, c->d_cp, c->d_cpm, radius, r_lo, r_hi, c->d_mindist, c->d_ball_cdist, c->d_ball_list, c->d_ball_n);
  HIPCHK(c, hipGetLastError());
  int n_ask = 0;
  HIPCHK(c, hipMemcpyAsync(&n_ask, c->d_ball_n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->ball_asked += (unsigned long long)n_ask;
  // the references whose answer depends on the queries: their planes on the columns of query->idx gathered into dense tiles, the pair
  // scan on those, the reference's walk over the queries folded into it (kernels_ball.inc)
  if (n_ask > 0) {
    int rc = ensure_qgather(c); if (rc) return rc;
    const int mt = (n_ask + 63) / 64;
    const size_t tile_u4 = (size_t)c->NG4 * c->P * 64;
    if (c->ball_tiles_cap < (size_t)mt) {
      if (c->d_ball_tiles) hipFree(c->d_ball_tiles);
      if (c->d_ball_key) hipFree(c->d_ball_key);
      c->d_ball_tiles = nullptr; c->d_ball_key = nullptr; c->ball_tiles_cap = 0;
      const size_t cap = (size_t)mt + (size_t)mt / 4 + 16;
      HIPCHK(c, hipMalloc(&c->d_ball_tiles, cap * tile_u4 * sizeof(uint4)));
      HIPCHK(c, hipMalloc(&c->d_ball_key, cap * 64 * sizeof(unsigned long long)));
      c->ball_tiles_cap = cap;
    }
    HIPCHK(c, hipMemsetAsync(c->d_ball_key, 0xFF, (size_t)mt * 64 * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_ball_tiles, 0, (size_t)mt * tile_u4 * sizeof(uint4), c->stream));
    BallSplit sp;
    for (int v = 0; v < 5; v++) { sp.w4[v] = c->ball_split[v]; sp.bit[v] = c->ball_split[5 + v]; }
    if (c->acgt) hipLaunchKernelGGL((ball_gather_cols_kernel<3>), dim3(mt), dim3(256), 0, c->stream, tiles, tile_first, c->W4, c->d_pmask, c->d_ball_list, n_ask, c->NG4, c->d_ball_tiles, sp);
    else         hipLaunchKernelGGL((ball_gather_cols_kernel<4>), dim3(mt), dim3(256), 0, c->stream, tiles, tile_first, c->W4, c->d_pmask, c->d_ball_list, n_ask, c->NG4, c->d_ball_tiles, sp);
    HIPCHK(c, hipGetLastError());
    constexpr int QTB = 16;
    const int n_qtiles = (c->nq + QTB - 1) / QTB;
    dim3 grid(scan_grid_size(n_qtiles, (mt + 3) / 4));
    if (c->acgt) hipLaunchKernelGGL((ball_scan_kernel<true, QTB>), grid, dim3(256), 0, c->stream, c->d_ball_tiles, mt, c->NG4, c->d_qg, c->nq, n_qtiles, c->d_ball_cdist, n_ask, radius, c->d_ball_key);
    else         hipLaunchKernelGGL((ball_scan_kernel<false, QTB>), grid, dim3(256), 0, c->stream, c->d_ball_tiles, mt, c->NG4, c->d_qg, c->nq, n_qtiles, c->d_ball_cdist, n_ask, radius, c->d_ball_key);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(ball_finish2_kernel, dim3((n_ask + 255) / 256), dim3(256), 0, c->stream, c->d_ball_key, c->d_ball_list, c->d_ball_cdist, n_ask, radius, r_lo, c->d_mindist);
    HIPCHK(c, hipGetLastError());
  }
  if (mindist_host) HIPCHK(c, hipMemcpyAsync(mindist_host, c->d_mindist, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int uvaia_gpu_ball(uvaia_gpu_ctx *c, const char *const *seq, int n_ref, int radius, int *mindist)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (n_ref < 0 || (n_ref > 0 && (!seq || !mindist))) return fail(c, UVAIA_GPU_EINVAL, "bad batch");
  if ((size_t)n_ref > c->max_pool) return fail(c, UVAIA_GPU_ESTATE, "batch of %d exceeds max_pool %zu", n_ref, c->max_pool);
  if (c->act_q0 != 0 || c->act_q1 != c->nq) return fail(c, UVAIA_GPU_ESTATE, "the radius search acts on the whole query set");
  if (n_ref == 0) return 0;
  int rc = ensure_batch_buffers(c); if (rc) return rc;
  rc = pack_rows(c, seq, nullptr, 0, nullptr, n_ref, c->d_batch, c->d_batch_nonn, c->d_batch_amb, c->d_batch_tot, 0);
  if (rc) return rc;
  return ball_range(c, c->d_batch, 0, (n_ref + 63) / 64, 0, n_ref, radius, mindist);
}

// the same over references [first, first + n) of the resident database (uvaia_gpu_db_append*): mindist[i] for reference first + i
int uvaia_gpu_ball_resident(uvaia_gpu_ctx *c, size_t first, size_t n, int radius, int *mindist)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (first + n > c->db_n) return fail(c, UVAIA_GPU_EINVAL, "range [%zu,+%zu) outside the database", first, n);
  if (c->act_q0 != 0 || c->act_q1 != c->nq) return fail(c, UVAIA_GPU_ESTATE, "the radius search acts on the whole query set");
  for (int i_ = 0; i_ < 3; i_++) if (c->scan_streams[i_]) HIPCHK(c, hipStreamSynchronize(c->scan_streams[i_]));
  const size_t step = (size_t)1 << 22;                       // stage 1 needs no more than 12 bytes per reference of work space
  for (size_t a = first; a < first + n; a += step) {
    const size_t b = std::min(first + n, a + step);
    const long long tf = (long long)(a / 64);
    int rc = ball_range(c, c->d_db, tf, (int)((b + 63) / 64 - a / 64), (int)(a - (size_t)tf * 64), (int)(b - (size_t)tf * 64), radius, mindist ? mindist + (a - first) : nullptr);
    if (rc) return rc;
  }
  return 0;
}

// references the last radius searches sent on to the queries (since the last call with reset != 0)
unsigned long long uvaia_gpu_ball_asked(uvaia_gpu_ctx *c, int reset) { if (!c) return 0; const unsigned long long v = c->ball_asked; if (reset) c->ball_asked = 0; return v; }

}  // extern "C"

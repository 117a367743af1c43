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
constexpr int SCAN_LOCKSTEP = 2;       // scan3, qblock mode: word groups between two barriers of a block
constexpr int AMB_ROW = 64;            // ints per REFERENCE side row (256 B, one coalesced wave load):
                                       //   [0] count (uncapped)  [1..11] word indices  [12 + 4k + p] plane p of the k-th listed word
constexpr int PACK_CHUNK = 4096;       // references per host->device staging round (multiple of 64)
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
  bool replay_recorded[NBUF] = {}, slice_scanned[NBUF] = {}, slice_cons_done[NBUF] = {};
  size_t slice_cap[NBUF] = {};              // pairs each counter buffer holds (grown when a slice needs more: slices may exceed a pool, see plan_subslices)
  int2 *d_cntb[NBUF] = {};                // counter buffers 1..NBUF-1 (buffer 0 is d_cnt2), allocated on first use
  int2 *d_tmin[NBUF] = {};                // per (query, tile of 64 references): {smallest mismatch count, largest ACGT-match count}, one per counter buffer
  int *d_mp[NBUF] = {};                   // --acgt: mismatches on the polymorphic columns per pair (dist_unique), one per counter buffer
  int slice_tiles[NBUF] = {}, slice_rb[NBUF] = {}, slice_re[NBUF] = {};
  long long slice_tf[NBUF] = {};
  size_t subslice = 32768;                // resident search: pools are cut into slices of this size (exact: see search_resident)
  int nq = 0, nq_pad = 0, nchar = 0, W = 0, W4 = 0, P = 4, NQ = 6, acgt = 0, k = 2, qt = 16, n_idx_c = 0;
  size_t trim = 0;
  size_t max_pool = 0, pool_pad = 0;
  // query side
  uint32_t *d_qp = nullptr;      // [nq_pad][W4][4][NQ]   full-information query planes
  uint32_t *d_qp2 = nullptr;     // [nq_pad][W4][4][4]    (lo, hi, isACGT, valid) for the two-counter scan (default mode)
  uint4 *d_qv = nullptr;         // [nq_pad/16][W4pad][16][4]  the same planes laid out for LDS staging (scan2v_kernel)
  int W4pad = 0;
  int scan_variant = 2;          // 2 = column-compressed scan3_kernel (default); 0 = scalar-operand scan2_*_kernel, 1 = LDS-broadcast
                                 // scan2v_kernel (UVAIA_GPU_SCAN=sgpr|lds), kept for A/B measurements
  // column-compressed scan: classes of the alignment columns for this query set, compressed/dirty query planes, derived reference planes
  uint32_t *d_cls = nullptr;     // [W4*4][4]  cL, cH, constMask, polyMask
  uint32_t *d_qpl = nullptr;     // [nq_pad][NP4][L,H,I,-][4]   compressed polymorphic columns of the queries
  uint32_t *d_stream = nullptr;  // per query tile: the dirty-word item stream of scan3_kernel (layout: see the kernel)
  uint2 *d_sdir = nullptr;       // [nq_pad/16] {first dword of the tile's stream, number of group records}
  int NP = 0, NP4 = 0;           // polymorphic columns counted densely
  int NR = 0, NR4 = 0, rare_max = -1;   // "rare" columns: all but <= rare_max queries carry the same base; sparse (items), groups follow the dense ones
  uint32_t *d_rmask = nullptr;   // [W4*4] mask of the rare columns
  uint32_t *d_qrare = nullptr;   // [nq][NR4*4][lo, hi, isACGT] the queries on the rare columns (--acgt: dist_unique of admitted pairs)
  int need_e_groups = 0, need_v_groups = 0, need_g_groups = 0, need_r_groups = 0;   // word groups whose E / V plane some query tile has to read (for the byte accounting)
  int act_q0 = 0, act_q1 = 0;    // active query range of the resident/slice paths (query shards across GPUs); whole set by default
  int scan_qblock = -1;          // UVAIA_GPU_SCAN_QBLOCK: block = 4 query tiles x 1 reference tile (1) or 1 x 4 (0); default by active query tiles
  bool serial = false;           // UVAIA_GPU_SERIAL: no scan/replay overlap (to time the kernels in isolation)
  int subslice_minq = 256;       // sub-slicing of pools only from this many active queries (UVAIA_GPU_SUBSLICE_MINQ)
  int scan_lds_pad = 0;          // extra (unused) LDS per scan block: caps the scan's blocks per CU so that replay waves find free slots
  int replay_lq = -1;            // replay caches the query's planes in LDS (22 KB per block): -1 = only with few queries (see open)
  int replay_prio = 1;           // replay waves raise their issue priority (UVAIA_GPU_REPLAY_PRIO=0 to compare)
  int scan_parts = 3;            // timing experiments only (UVAIA_GPU_SCAN_PARTS): bit 0 = polymorphic loop, bit 1 = constant/validity loop
  uint4 *d_batch_ev = nullptr, *d_batch_poly = nullptr, *d_db_ev = nullptr, *d_db_poly = nullptr;
  uint32_t *d_batch_grp = nullptr, *d_db_grp = nullptr;   // [tile][W4][64]  popc(E) | popc(V) << 16 of each word group (for queries that are all-N there)
  int *d_batch_tote = nullptr, *d_db_tote = nullptr;
  int *d_amb_q = nullptr;        // [nq][AMB_STRIDE] ambiguity-word lists of the queries
  int *d_batch_amb = nullptr, *d_db_amb = nullptr;   // same for the references of the batch buffer / database
  int *d_batch_tot = nullptr, *d_db_tot = nullptr;   // per reference: valid sites (default) / ACGT sites (--acgt), counted by pack_refs_kernel
  int2 *d_cnt2 = nullptr;        // [nq_pad][pool_pad] two-counter scan output
  unsigned long long *d_stats = nullptr;             // admissions, on-demand evaluations, dense fallbacks
  bool fullscan = false;         // UVAIA_GPU_FULLSCAN=1: four-counter scan + replay over it (kept for A/B and tests)
  size_t cnt_cap = 0;            // int4 elements allocated in d_cnt (lazily)
  uint32_t *d_cp = nullptr;      // consensus restricted to idx_c, one row [W4][4][NQ]
  uint32_t *d_cpm = nullptr;     // consensus restricted to idx_m (radius search)
  uint32_t *d_qpoly = nullptr;   // queries restricted to idx (radius search), [nq_pad][W4][4][NQ]
  uint32_t *d_cmrows = nullptr;  // 32 pseudo-query rows: [0] consensus on idx_c, [1] consensus on idx_m (radius search)
  int4 *d_cnt_cm = nullptr; size_t cnt_cm_cap = 0; int *d_mindist = nullptr;
  std::vector<uint32_t> h_qp_poly_src;  // kept to build d_qpoly lazily
  // heaps / state
  int *d_heap = nullptr, *d_n = nullptr, *d_T = nullptr, *d_snap = nullptr, *d_err = nullptr;
  // batch buffers
  uint4 *d_batch = nullptr;      // packed tiles of the current batch
  int *d_batch_nonn = nullptr;
  int4 *d_cnt = nullptr;         // [nq_pad][pool_pad]
  int4 *d_rt = nullptr, *d_tr = nullptr;   // [pool_pad]
  uint8_t *d_entered = nullptr;  // [pool_pad] (push) or [db_cap] (resident)
  size_t entered_cap = 0;
  uint8_t *d_stage = nullptr;    // device staging for raw characters (PACK_CHUNK rows)
  uint8_t *h_stage = nullptr;    // pinned host staging
  size_t pitch = 0;
  // resident database
  uint4 *d_db = nullptr;
  int *d_db_nonn = nullptr;
  size_t db_cap = 0, db_n = 0;
  // last batch (introspection)
  const uint4 *last_tiles = nullptr; const int *last_nonn = nullptr; int last_n = 0, last_rbegin = 0, last_ppad = 0, last_ntiles = 0;
  long long last_tile_first = 0;
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

// ------------------------------------------------------------------------------------------------------------
// device: packing raw characters into tile-interleaved bit-planes
// ------------------------------------------------------------------------------------------------------------
__constant__ uint8_t c_code[256];

// One block per tile of 64 database slots; wave v handles word groups w4 = v, v+4, ...  Slots outside
// [slot0, slot0+n_ref) are left untouched (the database is zero-initialised), so appends need not be tile-aligned.
// non_n_out (nullable): valid-site count over the FULL length (src/fastaseq.c:642-648).
// amb_out (nullable): per slot a side row of AMB_ROW ints = number of alignment words holding a partially ambiguous (valid,
// non-ACGT) site, up to AMB_CAP of their indices and the four plane words of each; a count above AMB_CAP means "list
// incomplete, rescan densely".
template <int P>
__global__ __launch_bounds__(256) void pack_refs_kernel(const uint8_t *__restrict__ chars, size_t pitch, int nchar,
                                                         long long slot0, int n_ref, int W4, uint4 *__restrict__ tiles,
                                                         long long tile_base, int *__restrict__ non_n_out, int *__restrict__ amb_out,
                                                         int *__restrict__ tot_out, int *__restrict__ errflag)
{
  __shared__ uint8_t lut[256];
  __shared__ int partial[4][64];
  __shared__ int partial_acgt[4][64];
  __shared__ int amb_n[64];
  __shared__ int amb_w[64][AMB_CAP];
  __shared__ uint32_t amb_p[64][AMB_CAP][4];
  lut[threadIdx.x] = c_code[threadIdx.x];
  if (threadIdx.x < 64) amb_n[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long tile = tile_base + blockIdx.x;
  const long long slot = tile * 64 + lane;
  const long long i = slot - slot0;                  // row in chars
  const bool active = (i >= 0 && i < n_ref);
  const uint8_t *row = chars + (active ? (size_t)i * pitch : 0);
  const bool vec = ((pitch & 15) == 0) && ((((uintptr_t)chars) & 15) == 0);
  int valid = 0, n_acgt = 0, bad = 0;
  for (int w4 = wv; w4 < W4; w4 += 4) {
    uint32_t pl[4][4];                               // [plane][j]
#pragma unroll
    for (int p = 0; p < 4; p++) { pl[p][0] = pl[p][1] = pl[p][2] = pl[p][3] = 0; }
    if (active) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int site0 = (w4 * 4 + j) * 32;
        if (site0 >= nchar) break;
        uint8_t b[32];
        if (vec && site0 + 32 <= (int)pitch) {
          const uint4 *v = reinterpret_cast<const uint4 *>(row + site0);
          uint4 x0 = v[0], x1 = v[1];
          memcpy(b, &x0, 16); memcpy(b + 16, &x1, 16);
        } else {
          for (int s = 0; s < 32; s++) b[s] = (site0 + s < nchar) ? row[site0 + s] : (uint8_t)'N';
        }
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, partial_code = 0;
#pragma unroll
        for (int s = 0; s < 32; s++) {
          uint32_t code = (site0 + s < nchar) ? lut[b[s]] : 0u;
          if (code == 0xFFu) { bad = 1; code = 0; }
          valid += (code != 0);
          n_acgt += (code != 0) & ((code & (code - 1)) == 0);
          partial_code |= code & (code - 1);
          if (P == 4) {
            a0 |= (code & 1u) << s; a1 |= ((code >> 1) & 1u) << s; a2 |= ((code >> 2) & 1u) << s; a3 |= ((code >> 3) & 1u) << s;
          } else {  // 2-bit code + "is ACGT" plane: A=0 C=1 G=2 T=3
            const uint32_t one = (code != 0) & ((code & (code - 1)) == 0);
            const uint32_t two = (code == 2) ? 1u : (code == 4) ? 2u : (code == 8) ? 3u : 0u;
            a0 |= (two & 1u & one) << s; a1 |= ((two >> 1) & one) << s; a2 |= one << s;
          }
        }
        pl[0][j] = a0; pl[1][j] = a1; pl[2][j] = a2; pl[3][j] = a3;
        if (partial_code) {
          const int pos = atomicAdd(&amb_n[lane], 1);
          if (pos < AMB_CAP) { amb_w[lane][pos] = w4 * 4 + j; amb_p[lane][pos][0] = a0; amb_p[lane][pos][1] = a1; amb_p[lane][pos][2] = a2; amb_p[lane][pos][3] = a3; }
        }
      }
      uint4 *dst = tiles + ((size_t)(tile * W4 + w4) * P) * 64 + lane;
#pragma unroll
      for (int p = 0; p < P; p++) dst[(size_t)p * 64] = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
    }
  }
  partial[wv][lane] = valid; partial_acgt[wv][lane] = n_acgt;
  __syncthreads();
  if (wv == 0 && active && non_n_out) non_n_out[slot] = partial[0][lane] + partial[1][lane] + partial[2][lane] + partial[3][lane];
  // total the two-counter scan subtracts from: valid sites (default) or ACGT sites (--acgt) of the whole sequence
  if (wv == 0 && active && tot_out) tot_out[slot] = (P == 4) ? (partial[0][lane] + partial[1][lane] + partial[2][lane] + partial[3][lane])
                                                           : (partial_acgt[0][lane] + partial_acgt[1][lane] + partial_acgt[2][lane] + partial_acgt[3][lane]);
  if (wv == 0 && active && amb_out) {
    int *a = amb_out + (size_t)slot * AMB_ROW;
    a[0] = amb_n[lane];
    for (int k = 0; k < AMB_CAP; k++) {
      const bool have = k < amb_n[lane];
      a[1 + k] = have ? amb_w[lane][k] : 0;
      for (int pp = 0; pp < 4; pp++) a[12 + 4 * k + pp] = have ? (int)amb_p[lane][k][pp] : 0;
    }
  }
  if (bad) atomicOr(errflag, 1);
}

// Packed interchange form -> this context's planes.  The on-disk database (uvaia_amd/csrc/host/uvdb.h) always holds the four
// IUPAC planes; a default-mode context copies them as they are, an --acgt context re-codes them to (lo, hi, isACGT).  Also
// writes the per-reference total the two-counter scan subtracts from (valid sites / ACGT sites).  One block per tile.
template <int P>
__global__ __launch_bounds__(256) void import_tiles_kernel(const uint4 *src, int W4, uint4 *dst /* P == 4: may be NULL = planes are in place already */, long long tile_base,
                                                            int *__restrict__ tot_out)
{
  __shared__ int partial[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long tile = tile_base + blockIdx.x;
  const uint4 *t = src + (size_t)blockIdx.x * W4 * 4 * 64 + lane;
  uint4 *o = dst + (size_t)tile * W4 * P * 64 + lane;
  int tot = 0;
  for (int w4 = wv; w4 < W4; w4 += 4) {
    const uint4 pA = t[(size_t)(w4 * 4 + 0) * 64], pC = t[(size_t)(w4 * 4 + 1) * 64], pG = t[(size_t)(w4 * 4 + 2) * 64], pT = t[(size_t)(w4 * 4 + 3) * 64];
    if (P == 4) {
      if (dst) { o[(size_t)(w4 * 4 + 0) * 64] = pA; o[(size_t)(w4 * 4 + 1) * 64] = pC; o[(size_t)(w4 * 4 + 2) * 64] = pG; o[(size_t)(w4 * 4 + 3) * 64] = pT; }
#pragma unroll
      for (int j = 0; j < 4; j++) tot += __popc(u4c(pA, j) | u4c(pC, j) | u4c(pG, j) | u4c(pT, j));
    } else {
      uint32_t L[4], H[4], I[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t a = u4c(pA, j), cc = u4c(pC, j), g = u4c(pG, j), tt = u4c(pT, j);
        const uint32_t par = a ^ cc ^ g ^ tt, three = (a & cc & (g | tt)) | (g & tt & (a | cc));
        I[j] = par & ~three; L[j] = (cc | tt) & I[j]; H[j] = (g | tt) & I[j];
        tot += __popc(I[j]);
      }
      o[(size_t)(w4 * 3 + 0) * 64] = make_uint4(L[0], L[1], L[2], L[3]);
      o[(size_t)(w4 * 3 + 1) * 64] = make_uint4(H[0], H[1], H[2], H[3]);
      o[(size_t)(w4 * 3 + 2) * 64] = make_uint4(I[0], I[1], I[2], I[3]);
    }
  }
  partial[wv][lane] = tot;
  __syncthreads();
  if (wv == 0) tot_out[tile * 64 + lane] = partial[0][lane] + partial[1][lane] + partial[2][lane] + partial[3][lane];
}

// ------------------------------------------------------------------------------------------------------------
// device: the pair scan (dominant kernel)
// ------------------------------------------------------------------------------------------------------------
template <int N> struct QWords { uint32_t v[N]; };
template <int N, typename PTR> static __device__ __forceinline__ void load_qwords(QWords<N> &d, PTR p)
{
#pragma unroll
  for (int i = 0; i < N; i++) d.v[i] = p[i];   // wave-uniform address -> s_load_dwordx8/x16
}

// Default mode: 4 planes (A,C,G,T bits of the IUPAC set).  Per (reference, query) pair and 32-site word:
//   r0 = #(equal & ACGT)  r1 = #(equal & both valid)  r2 = #(sets intersect)  r3 = #(both valid)
// (the four counters of the biomcmc kernel, call sites src/nearest.c:491,495).  15 VALU ops per pair-word.
// query planes per word: A,C,G,T, valid, is-ACGT.
template <int QT>
__global__ __launch_bounds__(256) void scan_iupac_kernel(const uint4 *__restrict__ db, long long tile_first, int n_tiles, int W4,
                                                          const uint32_t *__restrict__ qp, int4 *__restrict__ out, int ppad)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int trel = blockIdx.y * 4 + wave;
  if (trel >= n_tiles) return;
  const int q0 = blockIdx.x * QT;
  int acc[QT][4];
#pragma unroll
  for (int q = 0; q < QT; q++) { acc[q][0] = acc[q][1] = acc[q][2] = acc[q][3] = 0; }
  const uint4 *t = db + (size_t)(tile_first + trel) * W4 * 4 * 64 + lane;
  const size_t qstride = (size_t)W4 * 24;
  const uint32_t *qb = qp + (size_t)q0 * qstride;
  for (int w4 = 0; w4 < W4; w4++) {
    const uint4 pA = t[(size_t)(w4 * 4 + 0) * 64], pC = t[(size_t)(w4 * 4 + 1) * 64], pG = t[(size_t)(w4 * 4 + 2) * 64], pT = t[(size_t)(w4 * 4 + 3) * 64];
    const uint32_t rA[4] = {pA.x, pA.y, pA.z, pA.w}, rC[4] = {pC.x, pC.y, pC.z, pC.w}, rG[4] = {pG.x, pG.y, pG.z, pG.w}, rT[4] = {pT.x, pT.y, pT.z, pT.w};
    uint32_t rv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) rv[j] = rA[j] | rC[j] | rG[j] | rT[j];
    const uint32_t *s0 = qb + (size_t)w4 * 24;
    QWords<24> cur, nxt;
    load_qwords(cur, s0);
#pragma unroll
    for (int q = 0; q < QT; q++) {
      if (q + 1 < QT) load_qwords(nxt, s0 + (size_t)(q + 1) * qstride);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t qA = cur.v[j * 6 + 0], qC = cur.v[j * 6 + 1], qG = cur.v[j * 6 + 2], qT_ = cur.v[j * 6 + 3], qv = cur.v[j * 6 + 4], qa = cur.v[j * 6 + 5];
        uint32_t d = rA[j] ^ qA;
        d = B3(rC[j], qC, d, (TT_A ^ TT_B) | TT_C);
        d = B3(rG[j], qG, d, (TT_A ^ TT_B) | TT_C);
        const uint32_t nd = B3(rT[j], qT_, d, ~((TT_A ^ TT_B) | TT_C));   // all four planes equal
        uint32_t x = rA[j] & qA;
        x = B3(rC[j], qC, x, (TT_A & TT_B) | TT_C);
        x = B3(rG[j], qG, x, (TT_A & TT_B) | TT_C);
        x = B3(rT[j], qT_, x, (TT_A & TT_B) | TT_C);                       // sets intersect (implies both valid)
        acc[q][0] = bcnt_acc(nd & qa, acc[q][0]);
        acc[q][1] = bcnt_acc(nd & rv[j], acc[q][1]);
        acc[q][2] = bcnt_acc(x, acc[q][2]);
        acc[q][3] = bcnt_acc(rv[j] & qv, acc[q][3]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (q + 1 < QT) cur = nxt;
    }
  }
  const size_t r = (size_t)trel * 64 + lane;
#pragma unroll
  for (int q = 0; q < QT; q++) out[(size_t)(q0 + q) * ppad + r] = make_int4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
}

// --acgt mode: 3 planes (lo, hi of the 2-bit code, is-ACGT).  Per pair-word:
//   c0 = #(both ACGT & differ)  c1 = #(both ACGT)  c2 = #(both ACGT & differ) on polymorphic query columns
// (src/fastaseq.c:585-596; c2 separates score[5] from score[4], src/nearest.c:468-469).  8 VALU ops per pair-word.
// query planes per word: lo, hi, is-ACGT, is-ACGT restricted to the polymorphic columns (query->idx).
template <int QT>
__global__ __launch_bounds__(256) void scan_acgt_kernel(const uint4 *__restrict__ db, long long tile_first, int n_tiles, int W4,
                                                         const uint32_t *__restrict__ qp, int4 *__restrict__ out, int ppad)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int trel = blockIdx.y * 4 + wave;
  if (trel >= n_tiles) return;
  const int q0 = blockIdx.x * QT;
  int acc[QT][3];
#pragma unroll
  for (int q = 0; q < QT; q++) { acc[q][0] = acc[q][1] = acc[q][2] = 0; }
  const uint4 *t = db + (size_t)(tile_first + trel) * W4 * 3 * 64 + lane;
  const size_t qstride = (size_t)W4 * 16;
  const uint32_t *qb = qp + (size_t)q0 * qstride;
  for (int w4 = 0; w4 < W4; w4++) {
    const uint4 pL = t[(size_t)(w4 * 3 + 0) * 64], pH = t[(size_t)(w4 * 3 + 1) * 64], pI = t[(size_t)(w4 * 3 + 2) * 64];
    const uint32_t rL[4] = {pL.x, pL.y, pL.z, pL.w}, rH[4] = {pH.x, pH.y, pH.z, pH.w}, rI[4] = {pI.x, pI.y, pI.z, pI.w};
    const uint32_t *s0 = qb + (size_t)w4 * 16;
    QWords<16> cur, nxt;
    load_qwords(cur, s0);
#pragma unroll
    for (int q = 0; q < QT; q++) {
      if (q + 1 < QT) load_qwords(nxt, s0 + (size_t)(q + 1) * qstride);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t qL = cur.v[j * 4 + 0], qH = cur.v[j * 4 + 1], qI = cur.v[j * 4 + 2], qIp = cur.v[j * 4 + 3];
        const uint32_t d = rL[j] ^ qL;
        const uint32_t y = B3(rH[j], qH, d, (TT_A ^ TT_B) | TT_C);     // codes differ
        acc[q][0] = bcnt_acc(B3(y, rI[j], qI, TT_A & TT_B & TT_C), acc[q][0]);
        acc[q][1] = bcnt_acc(rI[j] & qI, acc[q][1]);
        acc[q][2] = bcnt_acc(B3(y, rI[j], qIp, TT_A & TT_B & TT_C), acc[q][2]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (q + 1 < QT) cur = nxt;
    }
  }
  const size_t r = (size_t)trel * 64 + lane;
#pragma unroll
  for (int q = 0; q < QT; q++) out[(size_t)(q0 + q) * ppad + r] = make_int4(acc[q][0], acc[q][1], acc[q][2], 0);
}

// XCD-aware work mapping for the scans (1-D grid).  Blocks are dealt round-robin over the 8 XCDs (each with its own L2), so
// block b runs on XCD b % 8.  All query tiles of one reference group are given ids with the same b % 8: the group's tiles
// are then fetched by ONE L2 instead of eight, and consecutive slots of an XCD share the group (temporal locality).
// Placement only affects speed/traffic, never results.
static __device__ __forceinline__ bool scan_work_item(int n_qtiles, int n_groups, int &qtile, int &group)
{
  const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
  qtile = slot % n_qtiles;
  group = (slot / n_qtiles) * 8 + xcd;
  return group < n_groups;
}
static inline unsigned scan_grid_size(int n_qtiles, int n_groups) { return (unsigned)n_qtiles * (unsigned)((n_groups + 7) / 8) * 8u; }

// ------------------------------------------------------------------------------------------------------------
// device: the two-counter pair scan (production path)
// ------------------------------------------------------------------------------------------------------------
// The gate of src/nearest.c:488-496 only needs the mismatch count m = valid - ACGT_matches of a pair, and the heap
// order is decided by ACGT_matches first (src/min_heap.c:41-47).  The dense pass therefore counts just
//   default: r0 = #(equal & ACGT), r3 = #(both valid)          --acgt: c0 = #(both ACGT & differ), c1 = #(both ACGT)
// and the replay kernel fetches the remaining counters of the few pairs that reach the heap (below).
// Reference words are re-coded per word into (lo, hi, isACGT, valid); query planes arrive in that coding.
// 4 logic ops + 2 v_bcnt per pair-word.
template <int QT>
__global__ __launch_bounds__(256) void scan2_iupac_kernel(const uint4 *__restrict__ db, long long tile_first, int n_tiles, int W4,
                                                           const uint32_t *__restrict__ qp2, int2 *__restrict__ out, int ppad, int n_qtiles, const int *__restrict__ tot)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int qtile, group;
  if (!scan_work_item(n_qtiles, (n_tiles + 3) / 4, qtile, group)) return;
  const int trel = group * 4 + wave;
  if (trel >= n_tiles) return;
  const int q0 = qtile * QT;
  int acc[QT][2];
#pragma unroll
  for (int q = 0; q < QT; q++) { acc[q][0] = acc[q][1] = 0; }
  const uint4 *t = db + (size_t)(tile_first + trel) * W4 * 4 * 64 + lane;
  const size_t qstride = (size_t)W4 * 16;
  const uint32_t *qb = qp2 + (size_t)q0 * qstride;
  for (int w4 = 0; w4 < W4; w4++) {
    const uint4 pA = t[(size_t)(w4 * 4 + 0) * 64], pC = t[(size_t)(w4 * 4 + 1) * 64], pG = t[(size_t)(w4 * 4 + 2) * 64], pT = t[(size_t)(w4 * 4 + 3) * 64];
    const uint32_t rA[4] = {pA.x, pA.y, pA.z, pA.w}, rC[4] = {pC.x, pC.y, pC.z, pC.w}, rG[4] = {pG.x, pG.y, pG.z, pG.w}, rT[4] = {pT.x, pT.y, pT.z, pT.w};
    uint32_t rL[4], rH[4], rI[4], rV[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t par = B3(rA[j], rC[j], rG[j], TT_A ^ TT_B ^ TT_C) ^ rT[j];                 // odd number of set planes
      const uint32_t ac = rA[j] & rC[j], gt = rG[j] & rT[j];
      const uint32_t three = B3(ac, rG[j], rT[j], TT_A & (TT_B | TT_C)) | B3(gt, rA[j], rC[j], TT_A & (TT_B | TT_C));
      rI[j] = par & ~three;                                                                      // exactly one plane set
      rL[j] = B3(rC[j], rT[j], rI[j], (TT_A | TT_B) & TT_C);                                     // A=0 C=1 G=2 T=3
      rH[j] = B3(rG[j], rT[j], rI[j], (TT_A | TT_B) & TT_C);
      rV[j] = B3(rA[j], rC[j], rG[j], TT_A | TT_B | TT_C) | rT[j];
    }
    const uint32_t *s0 = qb + (size_t)w4 * 16;
    QWords<16> cur, nxt;
    load_qwords(cur, s0);
#pragma unroll
    for (int q = 0; q < QT; q++) {
      if (q + 1 < QT) load_qwords(nxt, s0 + (size_t)(q + 1) * qstride);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t qL = cur.v[j * 4 + 0], qH = cur.v[j * 4 + 1], qI = cur.v[j * 4 + 2];
        const uint32_t d = rL[j] ^ qL;
        const uint32_t y = B3(rH[j], qH, d, (TT_A ^ TT_B) | TT_C);
        acc[q][0] = bcnt_acc(B3(y, rI[j], qI, ~TT_A & TT_B & TT_C), acc[q][0]);
      }
      // valid pairs = valid(reference) - #(reference valid & query invalid): only word groups where this query has an invalid
      // site can contribute (N runs, gaps, the trimmed ends) -- a wave-uniform test on scalar registers
      if (~(cur.v[3] & cur.v[7] & cur.v[11] & cur.v[15]) != 0u) {
#pragma unroll
        for (int j = 0; j < 4; j++) acc[q][1] = bcnt_acc(rV[j] & ~cur.v[j * 4 + 3], acc[q][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (q + 1 < QT) cur = nxt;
    }
  }
  const size_t r = (size_t)trel * 64 + lane;
  const int total = tot[r];
#pragma unroll
  for (int q = 0; q < QT; q++) out[(size_t)(q0 + q) * ppad + r] = make_int2(acc[q][0], total - acc[q][1]);
}

template <int QT>
__global__ __launch_bounds__(256) void scan2_acgt_kernel(const uint4 *__restrict__ db, long long tile_first, int n_tiles, int W4,
                                                          const uint32_t *__restrict__ qp, int2 *__restrict__ out, int ppad, int n_qtiles, const int *__restrict__ tot)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int qtile, group;
  if (!scan_work_item(n_qtiles, (n_tiles + 3) / 4, qtile, group)) return;
  const int trel = group * 4 + wave;
  if (trel >= n_tiles) return;
  const int q0 = qtile * QT;
  int acc[QT][2];
#pragma unroll
  for (int q = 0; q < QT; q++) { acc[q][0] = acc[q][1] = 0; }
  const uint4 *t = db + (size_t)(tile_first + trel) * W4 * 3 * 64 + lane;
  const size_t qstride = (size_t)W4 * 16;
  const uint32_t *qb = qp + (size_t)q0 * qstride;
  for (int w4 = 0; w4 < W4; w4++) {
    const uint4 pL = t[(size_t)(w4 * 3 + 0) * 64], pH = t[(size_t)(w4 * 3 + 1) * 64], pI = t[(size_t)(w4 * 3 + 2) * 64];
    const uint32_t rL[4] = {pL.x, pL.y, pL.z, pL.w}, rH[4] = {pH.x, pH.y, pH.z, pH.w}, rI[4] = {pI.x, pI.y, pI.z, pI.w};
    const uint32_t *s0 = qb + (size_t)w4 * 16;
    QWords<16> cur, nxt;
    load_qwords(cur, s0);
#pragma unroll
    for (int q = 0; q < QT; q++) {
      if (q + 1 < QT) load_qwords(nxt, s0 + (size_t)(q + 1) * qstride);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t qL = cur.v[j * 4 + 0], qH = cur.v[j * 4 + 1], qI = cur.v[j * 4 + 2];
        const uint32_t d = rL[j] ^ qL;
        const uint32_t y = B3(rH[j], qH, d, (TT_A ^ TT_B) | TT_C);
        acc[q][0] = bcnt_acc(B3(y, rI[j], qI, TT_A & TT_B & TT_C), acc[q][0]);
      }
      // comparable sites = ACGT(reference) - #(reference ACGT & query not ACGT): only groups where the query is not all ACGT
      if (~(cur.v[2] & cur.v[6] & cur.v[10] & cur.v[14]) != 0u) {
#pragma unroll
        for (int j = 0; j < 4; j++) acc[q][1] = bcnt_acc(rI[j] & ~cur.v[j * 4 + 2], acc[q][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (q + 1 < QT) cur = nxt;
    }
  }
  const size_t r = (size_t)trel * 64 + lane;
  const int total = tot[r];
#pragma unroll
  for (int q = 0; q < QT; q++) out[(size_t)(q0 + q) * ppad + r] = make_int2(acc[q][0], total - acc[q][1]);
}

// ------------------------------------------------------------------------------------------------------------
// device: column-compressed two-counter scan (production path)
// ------------------------------------------------------------------------------------------------------------
// uvaia compares each reference once against a consensus of the queries on the columns where the queries agree
// (src/nearest.c:428-433, src/fastaseq.c:744-768).  The same idea, restated for bit-planes and kept exact for every pair:
//   * a column is CONSTANT if all queries that are ACGT there carry the same base b (queries that are N/gap/ambiguous there
//     simply do not count), POLYMORPHIC if two queries carry different bases.
//   * constant columns:   ACGT matches(q,r) = popc(E_r & qI) = popc(E_r) - popc(E_r & ~qI),   E_r = [r is ACGT and equals b]
//                         (--acgt: mismatches = popc(D_r & qI), D_r = [r is ACGT and differs from b])
//     so, exactly like the valid-pair count, only the word groups where the query is NOT ACGT cost anything;
//   * polymorphic columns (typically 14-25 % of the alignment) are bit-gathered into contiguous words and compared densely.
// derive_ev_kernel / gather_poly_kernel build the per-reference planes for a given query set; scan3_kernel consumes them.

// one block per tile, wave v handles word groups v, v+4, ...:  ev[tile][w4][0] = E (or D with ACGT), [1] = valid (or is-ACGT)
template <bool ACGT>
__global__ __launch_bounds__(256) void derive_ev_kernel(const uint4 *__restrict__ tiles, long long tile_base, int W4,
                                                         const uint32_t *__restrict__ cls /*[W4*4][4]: cL, cH, constMask, polyMask*/,
                                                         uint4 *__restrict__ ev, int *__restrict__ tot_e, uint32_t *__restrict__ grp)
{
  constexpr int P = ACGT ? 3 : 4;
  __shared__ int partial[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long tile = tile_base + blockIdx.x;
  const uint4 *t = tiles + (size_t)tile * W4 * P * 64 + lane;
  uint4 *o = ev + (size_t)tile * W4 * 2 * 64 + lane;
  int te = 0;
  for (int w4 = wv; w4 < W4; w4 += 4) {
    const uint4 p0 = t[(size_t)(w4 * P + 0) * 64], p1 = t[(size_t)(w4 * P + 1) * 64], p2 = t[(size_t)(w4 * P + 2) * 64], p3 = t[(size_t)(w4 * P + (P - 1)) * 64];
    uint32_t e[4], v[4];
    int ge = 0, gv = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t rL, rH, rI, rV;
      if (ACGT) { rL = u4c(p0, j); rH = u4c(p1, j); rI = u4c(p2, j); rV = rI; }
      else {
        const uint32_t a = u4c(p0, j), cc = u4c(p1, j), g = u4c(p2, j), tt = u4c(p3, j);
        const uint32_t par = a ^ cc ^ g ^ tt, three = (a & cc & (g | tt)) | (g & tt & (a | cc));
        rI = par & ~three; rL = (cc | tt) & rI; rH = (g | tt) & rI; rV = a | cc | g | tt;
      }
      const uint32_t *c4 = cls + (size_t)(w4 * 4 + j) * 4;
      const uint32_t diff = (rL ^ c4[0]) | (rH ^ c4[1]);
      e[j] = rI & c4[2] & (ACGT ? diff : ~diff);
      v[j] = rV;
      ge += __popc(e[j]); gv += __popc(v[j]);
    }
    te += ge;
    grp[((size_t)tile * W4 + w4) * 64 + lane] = (uint32_t)ge | ((uint32_t)gv << 16);
    o[(size_t)(w4 * 2 + 0) * 64] = make_uint4(e[0], e[1], e[2], e[3]);
    o[(size_t)(w4 * 2 + 1) * 64] = make_uint4(v[0], v[1], v[2], v[3]);
  }
  partial[wv][lane] = te;
  __syncthreads();
  if (wv == 0) tot_e[tile * 64 + lane] = partial[0][lane] + partial[1][lane] + partial[2][lane] + partial[3][lane];
}

// one wave per tile: compresses the polymorphic columns (uniform masks) of each lane's reference into NPw contiguous words
// poly[tile][p4][plane L,H,I][lane] (uint4 = 4 consecutive compressed words)
// mask[w * mstride]: the columns of word w to gather; they land in groups g0, g0+1, ... of the tile's NG groups
template <bool ACGT>
__global__ __launch_bounds__(64) void gather_poly_kernel(const uint4 *__restrict__ tiles, long long tile_base, int W4, int NG, int g0,
                                                          const uint32_t *__restrict__ mask, int mstride, uint4 *__restrict__ poly)
{
  constexpr int P = ACGT ? 3 : 4;
  const int lane = threadIdx.x;
  const long long tile = tile_base + blockIdx.x;
  const uint4 *t = tiles + (size_t)tile * W4 * P * 64 + lane;
  uint4 *o = poly + ((size_t)tile * NG + g0) * 3 * 64 + lane;
  unsigned long long bL = 0, bH = 0, bI = 0;     // bit staging (uniform fill level)
  int fill = 0, ow = 0;                          // bits staged, compressed words emitted
  uint32_t wL[4] = {0, 0, 0, 0}, wH[4] = {0, 0, 0, 0}, wI[4] = {0, 0, 0, 0};
  auto flush_word = [&]() {
    wL[ow & 3] = (uint32_t)bL; wH[ow & 3] = (uint32_t)bH; wI[ow & 3] = (uint32_t)bI;
    bL >>= 32; bH >>= 32; bI >>= 32; fill -= 32;
    if ((ow & 3) == 3) {
      const int p4 = ow >> 2;
      o[(size_t)(p4 * 3 + 0) * 64] = make_uint4(wL[0], wL[1], wL[2], wL[3]);
      o[(size_t)(p4 * 3 + 1) * 64] = make_uint4(wH[0], wH[1], wH[2], wH[3]);
      o[(size_t)(p4 * 3 + 2) * 64] = make_uint4(wI[0], wI[1], wI[2], wI[3]);
      wL[0] = wL[1] = wL[2] = wL[3] = wH[0] = wH[1] = wH[2] = wH[3] = wI[0] = wI[1] = wI[2] = wI[3] = 0;
    }
    ow++;
  };
  for (int w4 = 0; w4 < W4; w4++) {
    const uint32_t *c4 = mask + (size_t)w4 * 4 * mstride;
    if ((c4[0] | c4[mstride] | c4[2 * mstride] | c4[3 * mstride]) == 0u) continue;   // no such column in this group (uniform)
    const uint4 p0 = t[(size_t)(w4 * P + 0) * 64], p1 = t[(size_t)(w4 * P + 1) * 64], p2 = t[(size_t)(w4 * P + 2) * 64], p3 = t[(size_t)(w4 * P + (P - 1)) * 64];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t m = c4[j * mstride];
      if (!m) continue;
      uint32_t rL, rH, rI;
      if (ACGT) { rL = u4c(p0, j); rH = u4c(p1, j); rI = u4c(p2, j); }
      else {
        const uint32_t a = u4c(p0, j), cc = u4c(p1, j), g = u4c(p2, j), tt = u4c(p3, j);
        const uint32_t par = a ^ cc ^ g ^ tt, three = (a & cc & (g | tt)) | (g & tt & (a | cc));
        rI = par & ~three; rL = (cc | tt) & rI; rH = (g | tt) & rI;
      }
      while (m) {                                                      // uniform loop over the polymorphic columns of the word
        const int b = __ffs(m) - 1; m &= m - 1;
        bL |= (unsigned long long)((rL >> b) & 1u) << fill;
        bH |= (unsigned long long)((rH >> b) & 1u) << fill;
        bI |= (unsigned long long)((rI >> b) & 1u) << fill;
        if (++fill == 32) flush_word();
      }
    }
  }
  if (fill > 0) { fill = 32; flush_word(); }
  while (ow & 3) { fill = 32; flush_word(); }                          // pad the last group with zero words
}

// scan over the derived planes.  Two loops per (16 queries x 64 references) pass of a wave:
//  1. polymorphic columns, dense: static 16-query unroll, counts in VGPRs, query words by scalar loads (VALU-bound);
//  2. constant columns + validity: only where a query is "dirty".  Measured (profiles/r01_issue_rate_microbench.txt): a CU
//     issues ONE scalar-ALU instruction per cycle for all four SIMDs, and VALU + SALU together ~1.9 per cycle, so a statically
//     unrolled chain of per-query bit tests is bound by its scalar bookkeeping, not by the popcounts.  The dirty work is
//     therefore a precomputed, query-tile-specific ITEM STREAM walked by a dynamic loop: every item carries its eight mask
//     words and the LDS offset of its query's counter, the counters (two u16 halves in one dword: constant-column deficit |
//     validity deficit << 16) live in LDS and take one ds_add_u32 per item, and a query that is N/gap over a whole 128-column
//     group costs a single ds_add of the reference's own per-group counts (grp[]).
//   out.x = ACGT matches (default) or ACGT mismatches (--acgt),  out.y = valid pairs (default) or comparable sites (--acgt)
// stream (dwords), per query tile, sdir[qtile] = {first dword, number of group records | number of rare records << 16}:
//   record = { w4 * 2048 (byte offset of the group's E plane in the tile) | needE | needV << 1,  n_full4 = ceil(n_full / 4),  n_generic,
//              length of the record in dwords }  + 4 n_full4 LDS offsets (padded with the scratch row 16 * 256)
//            (bit 2 of the first word: some query is all-N here, the group's grp[] row at byte offset w4 * 256 is needed)
//            + n_generic x { ~qI & constMask [4],  ~qV (default) / ~qI (--acgt) [4],  LDS offset, 0, 0, 0 }
//            + word items { ~qI & constMask, ~qV, LDS offset, 0 } of the queries that are dirty in ONE word of the group only, listed word
//              by word; their four counts sit in bits 4.. of the second header word (5 bits each)
//   rare record (after the group records) = { byte offset of a rare group's planes in the tile's gathered planes, word-item counts << 4, 0, 0 }
//            + items { sites, their lo bits, their hi bits, LDS offset }, word by word
// qpl[q][p4][L,H,I,-][4]: compressed planes of the polymorphic columns
template <int QT, bool ACGT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void scan3_kernel(const uint4 *__restrict__ ev, const uint4 *__restrict__ poly, long long tile_first, int n_tiles,
                                                     int W4, int NP4, int NPT, const uint32_t *__restrict__ qpl, const uint32_t *__restrict__ stream,
                                                     const uint2 *__restrict__ sdir, const uint32_t *__restrict__ grp,
                                                     const int *__restrict__ tot_e, const int *__restrict__ tot_v,
                                                     int2 *__restrict__ out, int ppad, int n_qtiles, int2 *__restrict__ tmin, int r_lo, int r_hi,
                                                     int *__restrict__ mp_out, int parts, int qtile_first, int qblock)
{
  static_assert(QT == 16, "stream offsets are laid out for tiles of 16 queries");
  constexpr uint32_t RARE_BIAS = 8192u;                      // the low counter half also takes what rare items give back: keep it positive
  __shared__ uint32_t lacc[4][QT + 1][64];                     // per wave: one packed counter per (query, lane) + a scratch row
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform by construction: say so
  int qtile, group, trel;
  if (qblock) {   // the four waves of a block = four query tiles over ONE reference tile: with few query tiles the passes over a tile
    int qg;       // then share its planes in L1/L2 instead of drifting apart and re-reading them from HBM
    if (!scan_work_item((n_qtiles + 3) / 4, n_tiles, qg, group)) return;
    qtile = qg * 4 + wave; trel = group;
    if (qtile >= n_qtiles) {          // no query tile for this wave: it only keeps the block's barriers company
      for (int cp = 0; cp < W4; cp += SCAN_LOCKSTEP) __builtin_amdgcn_s_barrier();
      return;
    }
  } else {        // four reference tiles, one query tile
    if (!scan_work_item(n_qtiles, (n_tiles + 3) / 4, qtile, group)) return;
    trel = group * 4 + wave;
    if (trel >= n_tiles) return;
  }
  qtile += qtile_first;                                        // only the active query tiles are scanned (query shards)
  const int q0 = qtile * QT;
  const size_t r = (size_t)trel * 64 + lane;
  typedef __attribute__((address_space(3))) uint32_t lds_u32;   // explicit LDS pointer: the stream loads stay scalar (no may-alias with the atomics)
  lds_u32 *my = (lds_u32 *)&lacc[wave][0][lane];
  {
    int acc[QT];
#pragma unroll
    for (int q = 0; q < QT; q++) acc[q] = 0;
    // ---- polymorphic columns, dense
    if (parts & 1) {
      const uint4 *t = poly + (size_t)(tile_first + trel) * NPT * 3 * 64 + lane;
      const size_t qstride = (size_t)NP4 * 16;
      const uint32_t *qb = qpl + (size_t)q0 * qstride;
      for (int p4 = 0; p4 < NP4; p4++) {
        const uint4 pL = t[(size_t)(p4 * 3 + 0) * 64], pH = t[(size_t)(p4 * 3 + 1) * 64], pI = t[(size_t)(p4 * 3 + 2) * 64];
        const uint32_t rL[4] = {pL.x, pL.y, pL.z, pL.w}, rH[4] = {pH.x, pH.y, pH.z, pH.w}, rI[4] = {pI.x, pI.y, pI.z, pI.w};
        const uint32_t *s0 = qb + (size_t)p4 * 16;
        QWords<12> cur, nxt;                                  // L[4], H[4], I[4] of the group: one s_load_dwordx8 + one x4
        load_qwords(cur, s0);
#pragma unroll
        for (int q = 0; q < QT; q++) {
          if (q + 1 < QT) load_qwords(nxt, s0 + (size_t)(q + 1) * qstride);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint32_t d = rL[j] ^ cur.v[j];
            const uint32_t y = B3(rH[j], cur.v[4 + j], d, (TT_A ^ TT_B) | TT_C);
            // --acgt: mismatches (they are also dist_unique).  default: NON-matches (padding bits included), which start the
            // constant-column deficit:  matches = te + 128 NP4 - (non-matches here + what the dirty words take away)
            acc[q] = bcnt_acc(ACGT ? B3(y, rI[j], cur.v[8 + j], TT_A & TT_B & TT_C) : B3(y, rI[j], cur.v[8 + j], ~(~TT_A & TT_B & TT_C)), acc[q]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (q + 1 < QT) cur = nxt;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < QT; q++) {
      if (ACGT) { mp_out[(size_t)(q0 + q) * ppad + r] = acc[q]; my[q * 64] = RARE_BIAS; }
      else my[q * 64] = (uint32_t)acc[q] + RARE_BIAS;
    }
  }
  // ---- constant columns and validity: the item stream of this query tile
  if (parts & 2) {
    const uint4 *t = ev + (size_t)(tile_first + trel) * W4 * 2 * 64 + lane;
    const uint32_t *gt = grp + (size_t)(tile_first + trel) * W4 * 64 + lane;
    const uint2 dir = sdir[qtile];
    typedef __attribute__((address_space(4))) const uint32_t cst_u32;   // constant address space: uniform loads from it are scalar loads
    const cst_u32 *sp = (const cst_u32 *)(stream + dir.x);
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char *myc = (lds_char *)my;
#define LDS_ADD(byte_off, val) __hip_atomic_fetch_add((lds_u32 *)(myc + (byte_off)), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define COUNT_ITEM(it)                                                                                              \
    {                                                                                                               \
      int e_ = 0, v_ = 0;                                                                                           \
      _Pragma("unroll") for (int j = 0; j < 4; j++) e_ = bcnt_acc(rE[j] & it.v[j], e_);                             \
      _Pragma("unroll") for (int j = 0; j < 4; j++) v_ = bcnt_acc(rV[j] & it.v[4 + j], v_);                         \
      LDS_ADD(it.v[8], (uint32_t)e_ | ((uint32_t)v_ << 16));                                                        \
    }
#define TOUCH_ITEM(it) asm volatile("" ::"s"(it.v[0]), "s"(it.v[1]), "s"(it.v[2]), "s"(it.v[3]), "s"(it.v[4]), "s"(it.v[5]), "s"(it.v[6]), "s"(it.v[7]), "s"(it.v[8]))
    // A plane that no item of a record needs is not loaded; its registers then hold whatever they held, which is harmless: every
    // use is an AND with a mask word that is zero for such a plane.  (The empty asm only tells the compiler the registers are
    // defined, so that it does not spend moves on zeroing them.)  The header of the NEXT record is requested as soon as the
    // current one is known (its length travels in the header): the walk never waits for a header.
    uint4 pE, pV;
    uint32_t g = 0u;
    asm volatile("" : "=v"(pE.x), "=v"(pE.y), "=v"(pE.z), "=v"(pE.w), "=v"(pV.x), "=v"(pV.y), "=v"(pV.z), "=v"(pV.w));
    // qblock: the four waves of the block walk the SAME reference tile for four query tiles.  They are kept within SCAN_LOCKSTEP
    // word groups of each other by barriers, so that a plane fetched by one of them is still in L1/L2 when the others want it
    // (left alone they drift apart and every pass re-reads the tile from HBM: 4x the traffic at 8 query tiles, FETCH_SIZE).
    int next_cp = 0;
    QWords<4> h;
    load_qwords(h, sp);
    for (uint32_t rec = 0; rec < (dir.y & 0xFFFFu); rec++) {
      const uint32_t h0 = h.v[0], n_full4 = h.v[1] & 15u, n_words = h.v[1] >> 4, n_gen = h.v[2];
      const cst_u32 *sp_next = sp + h.v[3];
      QWords<4> hn;
      load_qwords(hn, sp_next);                 // past the last record this reads the next tile's first header or the padding
      sp += 4;
      if (qblock) for (const int g_ = (int)(h0 >> 11); next_cp <= g_; next_cp += SCAN_LOCKSTEP) __builtin_amdgcn_s_barrier();
      {
        const char *tg = reinterpret_cast<const char *>(t) + (h0 & ~1023u);            // E plane of the group, V plane 1 KiB further
        if (h0 & 1u) pE = *reinterpret_cast<const uint4 *>(tg);
        if (h0 & 2u) pV = *reinterpret_cast<const uint4 *>(tg + 1024);
        if (n_full4) g = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(gt) + ((h0 & ~1023u) >> 3));
      }
      if (n_full4) {    // all-N queries: what they take away is the reference's own count for the group (same packing as the counters)
#pragma unroll 1
        for (uint32_t k = 0; k < n_full4; k++, sp += 4) {      // four LDS offsets per step; the list is padded with a scratch row
          QWords<4> o;
          load_qwords(o, sp);
          LDS_ADD(o.v[0], g); LDS_ADD(o.v[1], g); LDS_ADD(o.v[2], g); LDS_ADD(o.v[3], g);
        }
      }
      if (n_gen) {
        const uint32_t rE[4] = {pE.x, pE.y, pE.z, pE.w}, rV[4] = {pV.x, pV.y, pV.z, pV.w};
        // two items in flight: the words of the next one arrive while the current one is counted.  TOUCH_ITEM pins the wait for
        // a prefetched item BEFORE the following prefetch is issued (scalar loads return out of order: the only wait is "all").
        const cst_u32 *ip = sp;
        sp += n_gen * 12u;
        QWords<9> a, b;
        load_qwords(a, ip);
        TOUCH_ITEM(a);
        for (uint32_t k = 0; k < n_gen; k += 2, ip += 24) {
          load_qwords(b, ip + 12);       // the stream is padded: reading past the last item is harmless
          __builtin_amdgcn_sched_barrier(0);
          COUNT_ITEM(a);
          __builtin_amdgcn_sched_barrier(0);
          TOUCH_ITEM(b);
          if (k + 1 < n_gen) {
            load_qwords(a, ip + 24);
            __builtin_amdgcn_sched_barrier(0);
            COUNT_ITEM(b);
            __builtin_amdgcn_sched_barrier(0);
            TOUCH_ITEM(a);
          }
        }
      }
      if (n_words) {    // queries dirty in a single word of the group, listed word by word: { ~qI & constMask, ~qV, LDS offset, 0 } -- 6 VALU each
        const uint32_t rE[4] = {pE.x, pE.y, pE.z, pE.w}, rV[4] = {pV.x, pV.y, pV.z, pV.w};
#define WORD_ITEMS(J)                                                                                                             \
        _Pragma("unroll 1") for (uint32_t k = (n_words >> (5 * J)) & 31u; k > 0; k--, sp += 4) {                                  \
          QWords<3> it;                                                                                                           \
          load_qwords(it, sp);                                                                                                    \
          LDS_ADD(it.v[2], (uint32_t)bcnt_acc(rE[J] & it.v[0], 0) | ((uint32_t)bcnt_acc(rV[J] & it.v[1], 0) << 16));             \
        }
        WORD_ITEMS(0) WORD_ITEMS(1) WORD_ITEMS(2) WORD_ITEMS(3)
#undef WORD_ITEMS
      }
      sp = sp_next; h = hn;
    }
    if (qblock) for (; next_cp < W4; next_cp += SCAN_LOCKSTEP) __builtin_amdgcn_s_barrier();   // every wave passes the same number of barriers
    // ---- rare columns: the queries that do not carry a rare column's majority base were made dirty there above (their E bit is
    // taken away); here they get the true comparison on the gathered planes of the rare columns, as a negative deficit
    {
      const char *tr = reinterpret_cast<const char *>(poly + (size_t)(tile_first + trel) * NPT * 3 * 64 + lane);
      for (uint32_t rec = 0; rec < (dir.y >> 16); rec++) {
        QWords<2> h;
        load_qwords(h, sp);
        sp += 4;
        const uint4 pL = *reinterpret_cast<const uint4 *>(tr + h.v[0]), pH = *reinterpret_cast<const uint4 *>(tr + h.v[0] + 1024),
                    pI = *reinterpret_cast<const uint4 *>(tr + h.v[0] + 2048);
        const uint32_t rL[4] = {pL.x, pL.y, pL.z, pL.w}, rH[4] = {pH.x, pH.y, pH.z, pH.w}, rI[4] = {pI.x, pI.y, pI.z, pI.w};
        const uint32_t n_words = h.v[1] >> 4;
#define RARE_ITEMS(J)                                                                                                             \
        _Pragma("unroll 1") for (uint32_t k = (n_words >> (5 * J)) & 31u; k > 0; k--, sp += 4) {                                  \
          QWords<4> it;                                         /* sites, their lo bits, their hi bits, LDS offset */             \
          load_qwords(it, sp);                                                                                                    \
          const uint32_t d_ = rL[J] ^ it.v[1];                                                                                    \
          const uint32_t y_ = B3(rH[J], it.v[2], d_, (TT_A ^ TT_B) | TT_C);                                                       \
          const uint32_t g_ = ACGT ? B3(y_, rI[J], it.v[0], TT_A & TT_B & TT_C) : B3(y_, rI[J], it.v[0], ~TT_A & TT_B & TT_C);   \
          LDS_ADD(it.v[3], 0u - (uint32_t)bcnt_acc(g_, 0));     /* matches (default) / mismatches (--acgt) found: deficit goes down */ \
        }
        RARE_ITEMS(0) RARE_ITEMS(1) RARE_ITEMS(2) RARE_ITEMS(3)
#undef RARE_ITEMS
      }
    }
#undef LDS_ADD
#undef COUNT_ITEM
#undef TOUCH_ITEM
  }
  const int te = tot_e[r], tv = tot_v[r];
  const bool in_batch = ((int)r >= r_lo && (int)r < r_hi);
  const int *mp_in = mp_out;
  asm volatile("" : "+s"(mp_in));          // recompute the addresses here instead of keeping 16 of them alive across the loop above
#pragma unroll
  for (int q = 0; q < QT; q++) {
    const uint32_t pk = my[q * 64];
    const int c0 = (ACGT ? mp_in[(size_t)(q0 + q) * ppad + r] + te : te + NP4 * 128) - ((int)(pk & 0xFFFFu) - (int)RARE_BIAS), c1 = tv - (int)(pk >> 16);
    out[(size_t)(q0 + q) * ppad + r] = make_int2(c0, c1);
    // Two bounds per tile let the replay skip tiles that cannot admit anything: the smallest mismatch count (the gate of
    // src/nearest.c:488 needs mismatches < tolerance) and the largest ACGT-match count (a full heap only takes a key that is not
    // below its worst one, and ACGT matches are the first key: src/min_heap.c:95).  The second is by far the sharper one.
    const int mm = ACGT ? c0 : c1 - c0, kk = ACGT ? c1 - c0 : c0;
    int m = in_batch ? mm : 0x7fffffff, k = in_batch ? kk : (int)0x80000000;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = min(m, __shfl_xor(m, o)); k = max(k, __shfl_xor(k, o)); }
    if (lane == 0) tmin[(size_t)(q0 + q) * (ppad >> 6) + trel] = make_int2(m, k);
    __builtin_amdgcn_sched_barrier(0);                                // one query at a time: keeps the epilogue from inflating the register budget
  }
}

// LDS-broadcast variant of the two-counter scan.  Measured on MI355X (profiles/r01_valu_rate_microbench.txt): a VALU
// op with an SGPR source issues at half rate, so here the query words are staged through LDS (one broadcast
// ds_read_b128 per query word = its four planes in VGPRs) and every logic op has VGPR sources only.  Each wave owns
// R tiles (R references per lane) so that one LDS read feeds R pair-words.
template <int QT, int R, bool ACGT>
__global__ __launch_bounds__(256) void scan2v_kernel(const uint4 *__restrict__ db, long long tile_first, int n_tiles, int W4, int W4pad,
                                                      const uint4 *__restrict__ qv, int2 *__restrict__ out, int ppad, int n_qtiles)
{
  constexpr int CHW = 8, P = ACGT ? 3 : 4, CHUNK = CHW * QT * 4, PER_THREAD = CHUNK / 256;
  static_assert(CHUNK % 256 == 0, "staging assumes a multiple of the block size");
  __shared__ uint4 lq[2][CHUNK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int qtile, group;
  if (!scan_work_item(n_qtiles, (n_tiles + 4 * R - 1) / (4 * R), qtile, group)) return;     // whole block leaves together
  const int t0 = (group * 4 + wave) * R;
  int acc[QT][R][2];
#pragma unroll
  for (int q = 0; q < QT; q++)
#pragma unroll
    for (int r = 0; r < R; r++) { acc[q][r][0] = acc[q][r][1] = 0; }
  const uint4 *qsrc = qv + (size_t)qtile * W4pad * QT * 4;
  const int nchunks = W4pad / CHW;
  uint4 st[PER_THREAD];
#pragma unroll
  for (int k = 0; k < PER_THREAD; k++) lq[0][threadIdx.x + k * 256] = qsrc[threadIdx.x + k * 256];
  __syncthreads();
  for (int c = 0; c < nchunks; c++) {
    if (c + 1 < nchunks) {
#pragma unroll
      for (int k = 0; k < PER_THREAD; k++) st[k] = qsrc[(size_t)(c + 1) * CHUNK + threadIdx.x + k * 256];
    }
    const uint4 *lc = lq[c & 1];
    for (int w4l = 0; w4l < CHW; w4l++) {
      const int w4 = c * CHW + w4l;
      if (w4 >= W4) break;
      uint32_t rL[R][4], rH[R][4], rI[R][4], rV[R][4];
#pragma unroll
      for (int r = 0; r < R; r++) {
        // waves past the last tile recount the last one (their results are not stored): keeps the loads unconditional
        const int tr_ = (t0 + r) < n_tiles ? (t0 + r) : (n_tiles - 1);
        const uint4 *t = db + (size_t)(tile_first + tr_) * W4 * P * 64 + lane;
        const uint4 p0 = t[(size_t)(w4 * P + 0) * 64], p1 = t[(size_t)(w4 * P + 1) * 64], p2 = t[(size_t)(w4 * P + 2) * 64],
                    p3 = t[(size_t)(w4 * P + (P - 1)) * 64];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (ACGT) { rL[r][j] = u4c(p0, j); rH[r][j] = u4c(p1, j); rI[r][j] = u4c(p2, j); rV[r][j] = rI[r][j]; }
          else {
            const uint32_t a = u4c(p0, j), cc = u4c(p1, j), g = u4c(p2, j), tt = u4c(p3, j);
            const uint32_t par = B3(a, cc, g, TT_A ^ TT_B ^ TT_C) ^ tt;
            const uint32_t three = B3(a & cc, g, tt, TT_A & (TT_B | TT_C)) | B3(g & tt, a, cc, TT_A & (TT_B | TT_C));
            rI[r][j] = par & ~three;
            rL[r][j] = B3(cc, tt, rI[r][j], (TT_A | TT_B) & TT_C);
            rH[r][j] = B3(g, tt, rI[r][j], (TT_A | TT_B) & TT_C);
            rV[r][j] = B3(a, cc, g, TT_A | TT_B | TT_C) | tt;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < QT; q++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint4 qq = lc[(w4l * QT + q) * 4 + j];       // wave-uniform address: LDS broadcast
#pragma unroll
          for (int r = 0; r < R; r++) {
            const uint32_t d = rL[r][j] ^ qq.x;
            const uint32_t y = B3(rH[r][j], qq.y, d, (TT_A ^ TT_B) | TT_C);
            if (ACGT) {
              acc[q][r][0] = bcnt_acc(B3(y, rI[r][j], qq.z, TT_A & TT_B & TT_C), acc[q][r][0]);
              acc[q][r][1] = bcnt_acc(rI[r][j] & qq.z, acc[q][r][1]);
            } else {
              acc[q][r][0] = bcnt_acc(B3(y, rI[r][j], qq.z, ~TT_A & TT_B & TT_C), acc[q][r][0]);
              acc[q][r][1] = bcnt_acc(rV[r][j] & qq.w, acc[q][r][1]);
            }
          }
        }
      }
    }
    if (c + 1 < nchunks) {
#pragma unroll
      for (int k = 0; k < PER_THREAD; k++) lq[(c + 1) & 1][threadIdx.x + k * 256] = st[k];
    }
    __syncthreads();
  }
  const int q0 = qtile * QT;
#pragma unroll
  for (int r = 0; r < R; r++) {
    if (t0 + r >= n_tiles) continue;
    const size_t ri = (size_t)(t0 + r) * 64 + lane;
#pragma unroll
    for (int q = 0; q < QT; q++) out[(size_t)(q0 + q) * ppad + ri] = make_int2(acc[q][r][0], acc[q][r][1]);
  }
}

// ------------------------------------------------------------------------------------------------------------
// device: consensus pre-score with the reference's truncation (queue_distance_to_consensus, src/nearest.c:428-433)
// ------------------------------------------------------------------------------------------------------------
// One lane per reference walks the alignment words in increasing site order against the consensus restricted to
// query->idx_c.  rt = untruncated counters; tr = the counters the reference's early-exit loop would return with
// maxdist = *snap (it stops right after the site at which the mismatch counter reaches maxdist).
static __device__ __forceinline__ uint32_t prefix_through_nth_bit(uint32_t m, int nth)
{ // mask of all bit positions up to and including the nth (1-based) set bit of m
  for (int k = 1; k < nth; k++) m &= m - 1;
  const uint32_t bit = m & (0u - m);
  return bit | (bit - 1u);
}

template <bool ACGT>
__global__ __launch_bounds__(256) void consensus_kernel(const uint4 *__restrict__ db, long long tile_first, int n_tiles, int W4,
                                                         const uint32_t *__restrict__ cp, const int *__restrict__ snap_ptr,
                                                         int4 *__restrict__ rt, int4 *__restrict__ tr)
{
  constexpr int P = ACGT ? 3 : 4, NQ = ACGT ? 4 : 6;
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int trel = blockIdx.x * 4 + wave;
  if (trel >= n_tiles) return;
  const int snap = *snap_ptr;
  const uint4 *t = db + (size_t)(tile_first + trel) * W4 * P * 64 + lane;
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
  bool done = (snap <= 0);
  for (int w4 = 0; w4 < W4; w4++) {
    uint4 pl[P];
#pragma unroll
    for (int p = 0; p < P; p++) pl[p] = t[(size_t)(w4 * P + p) * 64];
    const uint32_t *s = cp + (size_t)w4 * 4 * NQ;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t k0, k1, k2, k3, M;     // planes of the four counters, mismatch plane
      if (ACGT) {
        const uint32_t rL = (&pl[0].x)[j], rH = (&pl[1].x)[j], rI = (&pl[2].x)[j];
        const uint32_t qL = s[j * 4 + 0], qH = s[j * 4 + 1], qI = s[j * 4 + 2];
        const uint32_t y = (rL ^ qL) | (rH ^ qH);
        k1 = rI & qI; k0 = y & k1; k2 = 0; k3 = 0; M = k0;
      } else {
        const uint32_t rA = (&pl[0].x)[j], rC = (&pl[1].x)[j], rG = (&pl[2].x)[j], rT = (&pl[3].x)[j];
        const uint32_t qA = s[j * 6 + 0], qC = s[j * 6 + 1], qG = s[j * 6 + 2], qT_ = s[j * 6 + 3], qv = s[j * 6 + 4], qa = s[j * 6 + 5];
        const uint32_t rv = rA | rC | rG | rT;
        const uint32_t nd = ~((rA ^ qA) | (rC ^ qC) | (rG ^ qG) | (rT ^ qT_));
        k0 = nd & qa; k1 = nd & rv; k2 = (rA & qA) | (rC & qC) | (rG & qG) | (rT & qT_); k3 = rv & qv;
        M = k3 & ~k0;
      }
      const int pm = __popc(M);
      const int mcur = ACGT ? c0 : (c3 - c0);
      if (!done && mcur + pm >= snap) {
        const uint32_t pmask = prefix_through_nth_bit(M, snap - mcur);
        t0 = c0 + __popc(k0 & pmask); t1 = c1 + __popc(k1 & pmask); t2 = c2 + __popc(k2 & pmask); t3 = c3 + __popc(k3 & pmask);
        done = true;
      }
      c0 += __popc(k0); c1 += __popc(k1); c2 += __popc(k2); c3 += __popc(k3);
    }
  }
  if (!done) { t0 = c0; t1 = c1; t2 = c2; t3 = c3; }
  const size_t r = (size_t)trel * 64 + lane;
  rt[r] = make_int4(c0, c1, c2, c3);
  tr[r] = make_int4(t0, t1, t2, t3);
}

// ------------------------------------------------------------------------------------------------------------
// device: score assembly and the ordered gate / heap replay
// ------------------------------------------------------------------------------------------------------------
// Assembles score[6] exactly as src/nearest.c:499-501 (default) or :464-469 (--acgt) from
//   cnt = pair counters over all compared columns, rt = untruncated consensus counters over idx_c,
//   rc  = the consensus counters the reference would hold in cq->res (possibly truncated), nn = cq->non_n.
// Pair counters over idx_m + idx are cnt - rt because every query equals the consensus on idx_c.
template <bool ACGT>
static __device__ __forceinline__ void assemble_scores(const int4 cnt, const int4 rt, const int4 rc, int nn, int S[6], int &mism)
{
  if (ACGT) {
    const int pmm = cnt.x - rt.x, pba = cnt.y - rt.y;   // mismatches / comparable sites outside idx_c
    S[0] = (pba + rc.y) - (pmm + rc.x);
    S[1] = pba + rc.y;
    S[2] = S[0] - (rc.y - rc.x);
    S[3] = nn;
    S[4] = rc.x + (pmm - cnt.z);
    S[5] = cnt.z;
    mism = S[1] - S[0];
  } else {
    const int p0 = cnt.x - rt.x;
    S[0] = p0 + rc.x;
    S[1] = cnt.y - rt.y + rc.y;
    S[2] = cnt.z - rt.z + rc.z;
    S[3] = cnt.w - rt.w + rc.w;
    S[4] = p0;
    S[5] = nn;
    mism = S[3] - S[0];
  }
}

static __device__ __forceinline__ bool lex_better(const int a[6], const int b[6])
{ // compare_q_item_score(a,b) < 0 (src/min_heap.c:41-47): a ranks strictly ahead of b
#pragma unroll
  for (int i = 0; i < 6; i++) { if (a[i] != b[i]) return a[i] > b[i]; }
  return false;
}

static __device__ __forceinline__ bool entry_better(const int *a, const int *b)
{
  for (int i = 0; i < 6; i++) { if (a[i] != b[i]) return a[i] > b[i]; }
  return false;
}

static __device__ void heap_sift_down(int *h, int n, int p)
{ // heap_bubble_down, src/min_heap.c:119-133: root = worst; swap towards the worse child while p is better than it
  for (;;) {
    int c = 2 * p, pick = p;
    for (int i = 0; i < 2; i++) if (c + i <= n && entry_better(h + pick * HEAP_ENTRY, h + (c + i) * HEAP_ENTRY)) pick = c + i;
    if (pick == p) return;
    for (int i = 0; i < HEAP_ENTRY; i++) { int tmp = h[p * HEAP_ENTRY + i]; h[p * HEAP_ENTRY + i] = h[pick * HEAP_ENTRY + i]; h[pick * HEAP_ENTRY + i] = tmp; }
    p = pick;
  }
}

static __device__ void heap_sift_up(int *h, int i)
{ // heap_bubble_up, src/min_heap.c:135-147
  while (i > 1) {
    const int parent = i / 2;
    if (!entry_better(h + parent * HEAP_ENTRY, h + i * HEAP_ENTRY)) return;
    for (int k = 0; k < HEAP_ENTRY; k++) { int tmp = h[parent * HEAP_ENTRY + k]; h[parent * HEAP_ENTRY + k] = h[i * HEAP_ENTRY + k]; h[i * HEAP_ENTRY + k] = tmp; }
    i = parent;
  }
}

template <bool ACGT> static __device__ __forceinline__ int entry_mismatches(const int *e)
{ return ACGT ? (e[1] - e[0]) : (e[3] - e[0]); }   // src/nearest.c:475 / :508

static __device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// One wave per query.  The wave reads 64 consecutive pair counters of its query at a time, filters them with
// bounds that can only tighten (T_ub = max mismatches held + 1, W = worst kept key: see DESIGN.md "gate"), and
// serialises the survivors in reference order through the exact test of src/nearest.c:488-508.
template <bool ACGT>
__global__ __launch_bounds__(64) void replay_kernel(const int4 *__restrict__ cnt, int ppad, const int4 *__restrict__ rt, const int4 *__restrict__ tr,
                                                     const int *__restrict__ nonn, int r_begin, int r_end, long long ord_base,
                                                     int *__restrict__ heap_g, int *__restrict__ n_g, int *__restrict__ T_g,
                                                     const int *__restrict__ snap_ptr, uint8_t *__restrict__ entered, int k)
{
  extern __shared__ int h[];                 // (k+1) entries of HEAP_ENTRY ints, slot 0 unused
  const int q = blockIdx.x, lane = threadIdx.x;
  int *hg = heap_g + (size_t)q * (k + 1) * HEAP_ENTRY;
  int n = n_g[q], T = T_g[q];
  const int snap = *snap_ptr;
  for (int i = lane; i < (n + 1) * HEAP_ENTRY; i += 64) h[i] = hg[i];
  __syncthreads();
  bool full = (n == k);
  int W[6] = {0, 0, 0, 0, 0, 0};
  int Tub = T;
  if (full) {
    int mx = 0;
    for (int s = 1 + lane; s <= n; s += 64) mx = max(mx, entry_mismatches<ACGT>(h + s * HEAP_ENTRY));
    Tub = wave_max(mx) + 1;
#pragma unroll
    for (int i = 0; i < 6; i++) W[i] = h[HEAP_ENTRY + i];
  }
  const int4 *crow = cnt + (size_t)q * ppad;
  bool dirty = false;
  for (int base = r_begin; base < r_end; base += 64) {
    const int r = base + lane;
    const bool valid = r < r_end;
    int S[6] = {0, 0, 0, 0, 0, 0}, m = 0x7fffffff;
    if (valid) {
      const int4 c = crow[r], a = rt[r];
      const int mc_true = ACGT ? a.x : (a.w - a.x);
      const int4 rc = (mc_true >= snap) ? tr[r] : a;     // what cq->res holds after src/nearest.c:431-432
      assemble_scores<ACGT>(c, a, rc, nonn[r], S, m);
    }
    bool cand = valid && m < Tub && (!full || lex_better(S, W));
    unsigned long long mask = __ballot(cand);
    while (mask) {
      const int i = __ffsll((long long)mask) - 1;
      int Si[6];
#pragma unroll
      for (int s = 0; s < 6; s++) Si[s] = __shfl(S[s], i);
      const int mi = __shfl(m, i);
      const bool accept = (mi < T) && (!full || lex_better(Si, W));   // src/nearest.c:488-496 + heap_insert :93-117
      if (!accept) { mask &= mask - 1; continue; }
      if (lane == 0) {
        const long long ord = ord_base + (base + i - r_begin);
        const int slot = full ? 1 : n + 1;
        int *e = h + slot * HEAP_ENTRY;
#pragma unroll
        for (int s = 0; s < 6; s++) e[s] = Si[s];
        e[6] = (int)(unsigned)(ord & 0xffffffffll); e[7] = (int)(ord >> 32);
        if (full) heap_sift_down(h, n, 1); else heap_sift_up(h, n + 1);
        entered[base + i] = 1;
      }
      if (!full) n++;
      dirty = true;
      __syncthreads();
      const bool was_full = full;
      full = (n == k);
      if (full) {
#pragma unroll
        for (int s = 0; s < 6; s++) W[s] = h[HEAP_ENTRY + s];
        T = entry_mismatches<ACGT>(W) + 1;               // src/nearest.c:506-508 / :474-475
        if (!was_full) {
          int mx = 0;
          for (int s = 1 + lane; s <= n; s += 64) mx = max(mx, entry_mismatches<ACGT>(h + s * HEAP_ENTRY));
          Tub = wave_max(mx) + 1;
        }
      }
      cand = valid && lane > i && m < Tub && (!full || lex_better(S, W));
      mask = __ballot(cand);
    }
  }
  if (dirty) {
    __syncthreads();
    for (int i = lane; i < (n + 1) * HEAP_ENTRY; i += 64) hg[i] = h[i];
    if (lane == 0) { n_g[q] = n; T_g[q] = T; }
  }
}

// ---- wave-cooperative heap operations (same comparisons and swaps as heap_sift_down/up above, hence the same layout):
// lanes 0..7 each hold one of the 8 ints of an entry, the lexicographic compare is a ballot + first-set-bit, a swap is one
// parallel read and write.  Every lane of the wave must call these (uniform control flow); h lives in LDS.
static __device__ __forceinline__ bool wave_entry_better(const int *h, int a, int b, int lane)
{ // entry a ranks strictly ahead of entry b
  const int i = lane & 7;
  const int va = h[a * HEAP_ENTRY + i], vb = h[b * HEAP_ENTRY + i];
  const unsigned long long m = __ballot((i < 6) && (va != vb)) & 0xFFull;
  if (!m) return false;
  const int first = __ffsll((long long)m) - 1;
  return __shfl(va, first) > __shfl(vb, first);
}

static __device__ __forceinline__ void wave_entry_swap(int *h, int a, int b, int lane)
{
  if (lane < HEAP_ENTRY) {
    const int va = h[a * HEAP_ENTRY + lane], vb = h[b * HEAP_ENTRY + lane];
    h[a * HEAP_ENTRY + lane] = vb; h[b * HEAP_ENTRY + lane] = va;
  }
  __syncthreads();     // one wave per block: an LDS fence + wave barrier
}

static __device__ void wave_sift_down(int *h, int n, int p, int lane)
{ // src/min_heap.c:119-133
  for (;;) {
    const int c = 2 * p;
    int pick = p;
    if (c <= n && wave_entry_better(h, pick, c, lane)) pick = c;
    if (c + 1 <= n && wave_entry_better(h, pick, c + 1, lane)) pick = c + 1;
    if (pick == p) return;
    wave_entry_swap(h, p, pick, lane);
    p = pick;
  }
}

// Root replacement for a full heap: the same comparisons and the same final layout as "overwrite slot 1, sift down"
// (src/min_heap.c:93-133), but the incoming entry stays in registers (`ment`: lanes l and l+8 hold its int l, l < 8) while
// the children on its way move up: per level one 64-byte LDS read (both children side by side in lanes 0-15), two
// lexicographic compares done with ballots and a DPP row shift, one 32-byte write.  No barrier inside: one wave, LDS
// operations of a wave complete in order.
static __device__ void wave_replace_root(int *h, int n, int ment, int lane)
{
  int p = 1;
  for (;;) {
    const int c = 2 * p;
    if (c > n) break;
    int v = 0;
    if (lane < 16 && c + (lane >> 3) <= n) v = h[c * HEAP_ENTRY + lane];               // lanes 0-7: child c, lanes 8-15: child c + 1
    const unsigned long long ne1 = __ballot(lane < 6 && ment != v), gt1 = __ballot(lane < 6 && ment > v);
    const bool b1 = ne1 && ((gt1 >> (__ffsll((long long)ne1) - 1)) & 1ull);             // the entry ranks strictly ahead of child c
    const int c1up = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);         // row_shr:8 -> lanes 8-15 see child c
    const int pk = b1 ? c1up : ment;                                                    // lanes 8-13: keys of the pick so far
    const unsigned long long ne2 = __ballot(lane >= 8 && lane < 14 && pk != v), gt2 = __ballot(lane >= 8 && lane < 14 && pk > v);
    const bool b2 = (c + 1 <= n) && ne2 && ((gt2 >> (__ffsll((long long)ne2) - 1)) & 1ull);   // ... strictly ahead of child c + 1
    if (!b1 && !b2) break;
    const int c2dn = __builtin_amdgcn_update_dpp(0, v, 0x108, 0xF, 0xF, false);         // row_shl:8 -> lanes 0-7 see child c + 1
    if (lane < HEAP_ENTRY) h[p * HEAP_ENTRY + lane] = b2 ? c2dn : v;
    p = b2 ? c + 1 : c;
  }
  if (lane < HEAP_ENTRY) h[p * HEAP_ENTRY + lane] = ment;
  __builtin_amdgcn_wave_barrier();
}

// sum over the wave by DPP (row shifts, then row broadcasts): six VALU steps instead of six LDS-crossbar shuffles
static __device__ __forceinline__ int wave_sum_dpp(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);    // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);    // row_shr:8  -> lane 15 of every row holds the row's sum
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);    // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);    // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}

static __device__ void wave_sift_up(int *h, int i, int lane)
{ // src/min_heap.c:135-147
  while (i > 1) {
    const int parent = i / 2;
    if (!wave_entry_better(h, parent, i, lane)) return;
    wave_entry_swap(h, parent, i, lane);
    i = parent;
  }
}

// ---- on-demand counters for the pairs that reach the heap -------------------------------------------------------
static __device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// contribution of one alignment word to  d1 = #(equal & valid & not ACGT)  and  d2 = #(sets intersect & not equal)
static __device__ __forceinline__ void iupac_word_extra(const uint32_t *__restrict__ dbw, size_t tile_abs, int lane_r, int W4, const uint32_t *__restrict__ qrow6,
                                                        int w, int &d1, int &d2)
{
  const size_t base = (((size_t)tile_abs * W4 + (w >> 2)) * 4) * 256 + (size_t)lane_r * 4 + (w & 3);   // u32 index of plane 0
  const uint32_t rA = dbw[base], rC = dbw[base + 256], rG = dbw[base + 512], rT = dbw[base + 768];
  const uint32_t *s = qrow6 + (size_t)w * 6;
  const uint32_t qA = s[0], qC = s[1], qG = s[2], qT_ = s[3], qa = s[5];
  const uint32_t rv = rA | rC | rG | rT;
  const uint32_t nd = ~((rA ^ qA) | (rC ^ qC) | (rG ^ qG) | (rT ^ qT_));
  const uint32_t e = nd & rv, a0 = nd & qa, x = (rA & qA) | (rC & qC) | (rG & qG) | (rT & qT_);
  d1 += __popc(e & ~a0);
  d2 += __popc(x & ~e);
}

// Default mode: text_matches - ACGT_matches and partial_matches - text_matches of one pair.  Both differences live on
// sites where the query or the reference carries a partially ambiguous code, so only the alignment words listed for
// either sequence are visited (all words if a list overflowed).  Whole wave cooperates; result in every lane.
// one word's contribution given both sets of four planes
static __device__ __forceinline__ void iupac_planes_extra(uint32_t rA, uint32_t rC, uint32_t rG, uint32_t rT, uint32_t qA, uint32_t qC, uint32_t qG, uint32_t qT_,
                                                          uint32_t qa, int &d1, int &d2)
{
  const uint32_t rv = rA | rC | rG | rT;
  const uint32_t nd = ~((rA ^ qA) | (rC ^ qC) | (rG ^ qG) | (rT ^ qT_));
  const uint32_t e = nd & rv, a0 = nd & qa, x = (rA & qA) | (rC & qC) | (rG & qG) | (rT & qT_);
  d1 += __popc(e & ~a0);
  d2 += __popc(x & ~e);
}

// Single round trip: the reference's side row (count, listed words and their planes) is one coalesced 256-B load, the
// reference planes at the query's listed words are requested at the same time (they do not depend on the row), and the
// query's planes come from `qw` (LDS copy of its row when it fits, else global memory).  The request is split from its use
// so that the replay can have the rows of the next candidates in flight while it works on the heap.
// Roles by lane: lanes 0..nq-1 take the query's listed words (reference planes gathered from the database tile), lanes
// nq..nq+AMB_CAP-1 the reference's listed words (index and planes straight from its side row), every lane the row's count.
struct ExtraReq { int row /* --acgt: dist_unique; default: the reference's count */; int w; uint32_t rA, rC, rG, rT; int nn; };

static __device__ __forceinline__ ExtraReq wave_iupac_request(const uint4 *__restrict__ db, size_t tile_abs, int lane_r, int W4, const int *__restrict__ ref_row,
                                                              int nq, int wq, int lane)
{
  const uint32_t *dbw = reinterpret_cast<const uint32_t *>(db);
  ExtraReq e;
  e.row = ref_row[0];
  e.w = -1; e.rA = e.rC = e.rG = e.rT = 0u; e.nn = 0;
  if (nq <= AMB_CAP) {
    if (lane < nq) {
      const size_t base = (((size_t)tile_abs * W4 + (wq >> 2)) * 4) * 256 + (size_t)lane_r * 4 + (wq & 3);
      e.w = wq; e.rA = dbw[base]; e.rC = dbw[base + 256]; e.rG = dbw[base + 512]; e.rT = dbw[base + 768];
    } else if (lane < nq + AMB_CAP) {
      const int kr = lane - nq;
      const int4 pl = *reinterpret_cast<const int4 *>(ref_row + 12 + 4 * kr);
      e.w = ref_row[1 + kr]; e.rA = (uint32_t)pl.x; e.rC = (uint32_t)pl.y; e.rG = (uint32_t)pl.z; e.rT = (uint32_t)pl.w;
    }
  }
  return e;
}

static __device__ int2 wave_iupac_finish(const ExtraReq &e, const uint4 *__restrict__ db, size_t tile_abs, int lane_r, int W4, const uint32_t *qw,
                                         const uint32_t *qlisted /*LDS bitmap of the words the query lists*/, int nq, int lane, bool &dense)
{
  const uint32_t *dbw = reinterpret_cast<const uint32_t *>(db);
  const int nr = e.row;
  dense = (nr > AMB_CAP || nq > AMB_CAP);
  if (!dense) {
    int pk = 0;
    // the reference's words the query lists itself are already covered by the query's lanes
    const bool mine = lane < nq || (lane < nq + nr && !((qlisted[e.w >> 5] >> (e.w & 31)) & 1u));
    if (mine) {
      const uint32_t *sq = qw + (size_t)e.w * 6;
      int d1 = 0, d2 = 0;
      iupac_planes_extra(e.rA, e.rC, e.rG, e.rT, sq[0], sq[1], sq[2], sq[3], sq[5], d1, d2);
      pk = d1 | (d2 << 16);                                   // at most 64 x 32 per half
    }
    const int tot = wave_sum_dpp(pk);
    return make_int2(tot & 0xFFFF, (int)((unsigned)tot >> 16));
  }
  int d1 = 0, d2 = 0;
  for (int w = lane; w < W4 * 4; w += 64) {
    const size_t base = (((size_t)tile_abs * W4 + (w >> 2)) * 4) * 256 + (size_t)lane_r * 4 + (w & 3);
    const uint32_t *sq = qw + (size_t)w * 6;
    iupac_planes_extra(dbw[base], dbw[base + 256], dbw[base + 512], dbw[base + 768], sq[0], sq[1], sq[2], sq[3], sq[5], d1, d2);
  }
  return make_int2(wave_sum(d1), wave_sum(d2));
}

// --acgt mode: mismatches on the polymorphic query columns (score[5], src/nearest.c:469) of one pair, dense.
static __device__ int wave_acgt_poly_mismatches(const uint4 *__restrict__ db, size_t tile_abs, int lane_r, int W4, const uint32_t *__restrict__ qrow4, int lane)
{
  int mp = 0;
  for (int w4 = lane; w4 < W4; w4 += 64) {
    const uint4 *t = db + ((size_t)tile_abs * W4 + w4) * 3 * 64 + lane_r;
    const uint4 pL = t[0], pH = t[64], pI = t[128];
    const uint32_t *s = qrow4 + (size_t)w4 * 16;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t y = ((&pL.x)[j] ^ s[j * 4 + 0]) | ((&pH.x)[j] ^ s[j * 4 + 1]);
      mp += __popc(y & (&pI.x)[j] & s[j * 4 + 3]);
    }
  }
  return wave_sum(mp);
}

// --acgt with rare columns: the scan's dense count covers the truly polymorphic columns only; the mismatches of one pair on the
// rare columns (which belong to dist_unique as well) are counted here on the gathered planes: NR4 * 4 words, one per lane.
static __device__ int wave_rare_mismatches(const uint4 *__restrict__ polyp, size_t tile_abs, int lane_r, int NPT, int NP4, int NR4,
                                           const uint32_t *__restrict__ qr /* [NR4 * 4][lo, hi, isACGT] */, int lane)
{
  const uint32_t *pw = reinterpret_cast<const uint32_t *>(polyp);
  int mp = 0;
  for (int w = lane; w < NR4 * 4; w += 64) {
    const size_t base = (((size_t)tile_abs * NPT + NP4 + (w >> 2)) * 3) * 256 + (size_t)lane_r * 4 + (w & 3);
    const uint32_t rL = pw[base], rH = pw[base + 256], rI = pw[base + 512];
    const uint32_t *s_ = qr + (size_t)w * 3;
    mp += __popc(((rL ^ s_[0]) | (rH ^ s_[1])) & rI & s_[2]);
  }
  return wave_sum_dpp(mp);
}

// Replay over the two-counter scan.  One wave per query walks the batch in reference order, 256 references per
// round.  Between two admissions the heap state is constant, so the exact tests of src/nearest.c:488-496 and
// src/min_heap.c:95 are evaluated for 64 references at once (ballot); the first survivor in order gets its missing
// counters on demand, is compared exactly, and if admitted the remaining lanes are re-tested against the new state.
// CONS = false: no column is constant and complete (query->n_idx_c == 0), every consensus counter is zero and is not loaded;
// in default mode valid_ref_sites (key 5) is then fetched only for the pairs that reach the exact compare.
template <bool ACGT, bool CONS>
__global__ __launch_bounds__(64) void replay2_kernel(const int2 *__restrict__ cnt, int ppad, const int4 *__restrict__ rt, const int4 *__restrict__ tr,
                                                      const int *__restrict__ nonn, const int *__restrict__ amb, int r_begin, int r_end, long long ord_base,
                                                      int *__restrict__ heap_g, int *__restrict__ n_g, int *__restrict__ T_g,
                                                      const int *__restrict__ snap_ptr, uint8_t *__restrict__ entered, int k,
                                                      const uint4 *__restrict__ db, long long tile_first, int W4,
                                                      const uint32_t *__restrict__ qfull, const int *__restrict__ amb_q,
                                                      unsigned long long *__restrict__ stats, int q_first, const int2 *__restrict__ tmin,
                                                      const int *__restrict__ mpbuf, int lq_words, int prio_,
                                                      const uint4 *__restrict__ polyp, int NPT, int NP4, int NR4, const uint32_t *__restrict__ qrare)
{
  extern __shared__ int h[];
  if (prio_) __builtin_amdgcn_s_setprio(3);     // few latency-bound waves on the critical path: win issue arbitration against co-resident scan waves
  const int q = blockIdx.x + q_first, lane = threadIdx.x;
  int *hg = heap_g + (size_t)q * (k + 1) * HEAP_ENTRY;
  int n = min(max(n_g[q], 0), k), T = T_g[q];       // clamp: an imported state blob is external input
  const int snap = *snap_ptr;
  for (int i = lane; i < (n + 1) * HEAP_ENTRY; i += 64) h[i] = hg[i];
  __syncthreads();
  bool full = (n == k);
  int W[6] = {0, 0, 0, 0, 0, 0};
  if (full) {
#pragma unroll
    for (int i = 0; i < 6; i++) W[i] = h[HEAP_ENTRY + i];
  }
  const int2 *crow = cnt + (size_t)q * ppad;
  const uint32_t *qrow = qfull + (size_t)q * W4 * 4 * (ACGT ? 4 : 6);
  // default mode: keep the query's full planes in LDS (behind the heap) when they fit; the on-demand counters then need a
  // single global round trip per pair
  const uint32_t *qw = qrow;
  if (!ACGT && lq_words > 0) {
    uint32_t *lq = reinterpret_cast<uint32_t *>(h + (k + 1) * HEAP_ENTRY);
    for (int i = lane; i < lq_words; i += 64) lq[i] = qrow[i];
    qw = lq;
  }
  const int aqv = (lane < AMB_STRIDE) ? amb_q[(size_t)q * AMB_STRIDE + lane] : 0;      // the query's ambiguity-word list, one int per lane
  const int aq_n = __shfl(aqv, 0);                                                      // how many words it lists
  const int aq_w = __shfl(aqv, 1 + ((lane < aq_n && lane < AMB_CAP) ? lane : 0));       // lane l: the l-th listed word
  uint32_t *qlisted = reinterpret_cast<uint32_t *>(h + (k + 1) * HEAP_ENTRY) + lq_words;   // 32-word bitmap of the listed words
  if (lane < 32) qlisted[lane] = 0u;
  __syncthreads();
  if (lane < aq_n && aq_n <= AMB_CAP) atomicOr(&qlisted[aq_w >> 5], 1u << (aq_w & 31));
  __syncthreads();
  bool dirty = false;
  unsigned n_admit = 0, n_demand = 0, n_dense = 0;
  // software pipeline: the counters of round i+1 are requested before round i is processed (the kernel is latency bound:
  // one wave per query, a few hundred dependent rounds)
  // Traversal.  Tiles (64 references) whose smallest mismatch count is not below the current tolerance cannot produce an
  // admission, and the tolerance only changes through admissions; so the wave walks the tile minima (64 tiles per load) and
  // fetches pair counters only for tiles that can pass the gate now, four tiles in flight.  After an admission that changed
  // the tolerance the set of needed tiles is derived again (a tile skipped earlier may qualify once the tolerance rises).
  // Exact for any sequence of tolerances.
  const bool use_tmin = tmin != nullptr;
  const int2 *tmrow = use_tmin ? tmin + (size_t)q * (ppad >> 6) : nullptr;
  const int n_slice_tiles = (r_end + 63) >> 6;
  constexpr int D = 8;
  int2 nb = make_int2(-1, 0x7fffffff);             // bounds of the next 64 tiles, requested one round ahead
  if (use_tmin && lane < n_slice_tiles) nb = tmrow[lane];
  for (int tb = 0; tb < n_slice_tiles; tb += 64) {
    int tm = 0x7fffffff, tk = 0x7fffffff;            // tile bounds: smallest mismatch count, largest first key
    if (tb + lane < n_slice_tiles) { tm = nb.x; tk = nb.y; }
    if (use_tmin && tb + 64 + lane < n_slice_tiles) nb = tmrow[tb + 64 + lane];
    // a tile can admit only if some reference passes the gate and (heap full) some reference's first key reaches the worst kept one
    // With consensus counters (CONS) a pre-score cut short at the snapshot lowers a pair's mismatch count, but never below the
    // snapshot, and never raises its first key: the bounds stay valid with "tm < T" widened to "tm < T or snapshot < T".
    auto needed = [&]() -> unsigned long long { return __ballot((tm < T || (CONS && snap < T)) && (!full || tk >= W[0])); };
    unsigned long long P = needed();
    while (P) {
      int tsel[D]; int2 c[D]; int4 a[D], rc[D]; int nn[D], m[D], K0[D], K1[D], K2[D], K3[D]; bool valid[D];
      {
        unsigned long long rest = P;
#pragma unroll
        for (int i = 0; i < D; i++) { tsel[i] = rest ? (__ffsll((long long)rest) - 1) : -1; rest &= rest - 1; }
      }
#pragma unroll
      for (int i = 0; i < D; i++) {                                    // request the counters of up to D needed tiles
        c[i] = make_int2(0, 0); a[i] = rc[i] = make_int4(0, 0, 0, 0); nn[i] = 0; valid[i] = false;
        if (tsel[i] >= 0) {
          const int r = (tb + tsel[i]) * 64 + lane;
          valid[i] = (r >= r_begin && r < r_end);
          if (valid[i]) {
            c[i] = crow[r];
            if (CONS) { a[i] = rt[r]; rc[i] = tr[r]; }
            if (CONS || ACGT) nn[i] = nonn[r];
          }
        }
      }
      bool regroup = false;
#pragma unroll
      for (int u = 0; u < D; u++) {
        if (tsel[u] < 0 || regroup) continue;
        {
          const int mc_true = ACGT ? a[u].x : (a[u].w - a[u].x);
          if (mc_true < snap) rc[u] = a[u];                 // cq->res was not cut short (src/nearest.c:431-432)
          if (ACGT) {     // keys 0..3 are known from the two counters: matches, valid, unique matches, valid ref sites
            K1[u] = c[u].y - a[u].y + rc[u].y;
            K0[u] = K1[u] - (c[u].x - a[u].x + rc[u].x);
            K2[u] = K0[u] - (rc[u].y - rc[u].x);
            K3[u] = nn[u];
            m[u] = K1[u] - K0[u];
          } else {        // only key 0 (ACGT matches) and the pair's valid count are known
            K0[u] = c[u].x - a[u].x + rc[u].x;
            K3[u] = c[u].y - a[u].w + rc[u].w;
            K1[u] = K2[u] = 0;
            m[u] = K3[u] - K0[u];
          }
        }
        const int T_used = T;
        const int base_u = (tb + tsel[u]) * 64;
        auto may_enter = [&]() -> bool {
          if (!valid[u] || m[u] >= T) return false;
          if (!full) return true;
          if (ACGT) {
            if (K0[u] != W[0]) return K0[u] > W[0];
            if (K1[u] != W[1]) return K1[u] > W[1];
            if (K2[u] != W[2]) return K2[u] > W[2];
            return K3[u] >= W[3];
          }
          return K0[u] >= W[0];
        };
        unsigned long long mask = __ballot(may_enter());
        // What a candidate needs beyond the two scan counters (side row / dist_unique, valid sites of the reference) is requested
        // for the next PF candidates of the tile while the current one goes through the heap: the requests depend on the
        // reference only, never on the heap state, so a candidate that drops out after an admission merely wastes its request.
        constexpr int PF = 3;
        int pf_idx[PF]; ExtraReq pf[PF];
#pragma unroll
        for (int s_ = 0; s_ < PF; s_++) pf_idx[s_] = -1;
        auto request = [&](int i_) -> ExtraReq {
          const int rl_ = base_u + i_;
          ExtraReq e;
          if (ACGT) { e.row = mpbuf ? mpbuf[(size_t)q * ppad + rl_] : 0; e.w = -1; e.rA = e.rC = e.rG = e.rT = 0u; }
          else e = wave_iupac_request(db, (size_t)tile_first + (size_t)(rl_ >> 6), rl_ & 63, W4, amb + (size_t)rl_ * AMB_ROW, aq_n, aq_w, lane);
          e.nn = (CONS || ACGT) ? 0 : nonn[rl_];
          return e;
        };
        while (mask) {
          const int i = __ffsll((long long)mask) - 1;
          {   // keep the first PF candidates of the mask requested
            unsigned long long rest = mask;
#pragma unroll
            for (int d_ = 0; d_ < PF; d_++) {
              if (!rest) break;
              const int ib = __ffsll((long long)rest) - 1; rest &= rest - 1;
              bool have = false;
#pragma unroll
              for (int s_ = 0; s_ < PF; s_++) have |= (pf_idx[s_] == ib);
              if (!have) {
                // free slot: one whose candidate is no longer among the first PF of the mask (or empty)
                // (the tolerance can rise after an admission, so the mask can gain candidates ahead of the ones already requested:
                // with no stale slot, the slot of the farthest candidate is given up)
                int slot = -1, far = 0;
#pragma unroll
                for (int s_ = PF - 1; s_ >= 0; s_--) {
                  if (pf_idx[s_] < 0 || pf_idx[s_] < i || !((mask >> pf_idx[s_]) & 1ull)) slot = s_;
                  if (pf_idx[s_] > pf_idx[far]) far = s_;
                }
                if (slot < 0) slot = far;
                const ExtraReq e = request(ib);
#pragma unroll
                for (int s_ = 0; s_ < PF; s_++) if (s_ == slot) { pf[s_] = e; pf_idx[s_] = ib; }
              }
            }
          }
          ExtraReq cur = pf[0];
#pragma unroll
          for (int s_ = 1; s_ < PF; s_++) if (pf_idx[s_] == i) cur = pf[s_];
#pragma unroll
          for (int s_ = 0; s_ < PF; s_++) if (pf_idx[s_] == i) pf_idx[s_] = -1;
          const int rl = base_u + i;                                      // index relative to the first tile of the batch
          const int cx = __shfl(c[u].x, i), cy = __shfl(c[u].y, i);
          const int4 ai = make_int4(__shfl(a[u].x, i), __shfl(a[u].y, i), __shfl(a[u].z, i), __shfl(a[u].w, i));
          const int4 ri = make_int4(__shfl(rc[u].x, i), __shfl(rc[u].y, i), __shfl(rc[u].z, i), __shfl(rc[u].w, i));
          const int nni = (CONS || ACGT) ? __shfl(nn[u], i) : cur.nn;
          const size_t tile_abs = (size_t)tile_first + (size_t)(rl >> 6);
          int Si[6], mi;
          n_demand++;
          if (ACGT) {
            // with the column-compressed scan the dense count on the polymorphic columns IS score[5] (src/nearest.c:469)
            int mp = mpbuf ? cur.row : wave_acgt_poly_mismatches(db, tile_abs, rl & 63, W4, qrow, lane);
            if (mpbuf && NR4 > 0) mp += wave_rare_mismatches(polyp, tile_abs, rl & 63, NPT, NP4, NR4, qrare + (size_t)q * NR4 * 12, lane);
            assemble_scores<true>(make_int4(cx, cy, mp, 0), ai, ri, nni, Si, mi);
          } else {
            bool dense;
            const int2 d = wave_iupac_finish(cur, db, tile_abs, rl & 63, W4, qw, qlisted, aq_n, lane, dense);
            n_dense += dense;
            assemble_scores<false>(make_int4(cx, cx + d.x, cx + d.x + d.y, cy), ai, ri, nni, Si, mi);
          }
          const bool accept = (mi < T) && (!full || lex_better(Si, W));   // src/nearest.c:488-496 + heap_insert :93-117
          if (!accept) { mask &= mask - 1; continue; }
          {
            const long long ord = ord_base + (rl - r_begin);
            int v = (int)(unsigned)(ord & 0xffffffffll);                   // lanes l and l + 8 hold int l of the new entry
            if ((lane & 7) == 7) v = (int)(ord >> 32);
#pragma unroll
            for (int sidx = 0; sidx < 6; sidx++) if ((lane & 7) == sidx) v = Si[sidx];
            if (lane == 0) entered[rl] = 1;
            if (full) wave_replace_root(h, n, v, lane);
            else {
              if (lane < HEAP_ENTRY) h[(n + 1) * HEAP_ENTRY + lane] = v;
              __syncthreads();
              wave_sift_up(h, n + 1, lane);
            }
          }
          if (!full) n++;
          dirty = true; n_admit++;
          __syncthreads();
          full = (n == k);
          if (full) {
#pragma unroll
            for (int sidx = 0; sidx < 6; sidx++) W[sidx] = h[HEAP_ENTRY + sidx];
            T = entry_mismatches<ACGT>(W) + 1;                             // src/nearest.c:506-508 / :474-475
          }
          mask = __ballot(lane > i && may_enter());
        }
        P &= ~(1ull << tsel[u]);                                          // this tile is done
        if (T != T_used) {
          // Needed tiles under the new tolerance, after this one.  A lower tolerance only removes tiles: the ones already in
          // flight are processed anyway (their ballots come out empty).  A higher tolerance can add a tile that lies BEFORE
          // the next tile in flight; only then must the group be formed again to keep the stream order.
          P = needed() & ~((2ull << tsel[u]) - 1ull);
          unsigned long long inflight = 0ull; int last = -1;
#pragma unroll
          for (int v = 0; v < D; v++) if (v > u && tsel[v] >= 0) { inflight |= (1ull << tsel[v]); last = tsel[v]; }
          // a needed tile that is not in flight but precedes the last tile in flight would be visited out of order
          if (T > T_used && last >= 0 && (P & ~inflight & ((1ull << last) - 1ull)) != 0ull) regroup = true;
          else P |= inflight;                                             // keep the tiles in flight in the pending set
        }
      }
    }
  }
  if (dirty) {
    __syncthreads();
    for (int i = lane; i < (n + 1) * HEAP_ENTRY; i += 64) hg[i] = h[i];
    if (lane == 0) { n_g[q] = n; T_g[q] = T; }
  }
  if (stats && lane == 0) { atomicAdd(&stats[0], (unsigned long long)n_admit); atomicAdd(&stats[1], (unsigned long long)n_demand); atomicAdd(&stats[2], (unsigned long long)n_dense); }
}

// Radius search, per reference: replays seq_ball_against_query_structure() (src/fastaseq.c:660-696) on exact distances.
// dist_cm[0][r], dist_cm[1][r]: distance to the consensus on idx_c / idx_m; dist_q[q][r]: distance to query q on idx.
// A truncated scan of the reference returns min(true distance, maxdist), which is all that is needed here.
template <bool ACGT>
__global__ void ball_reduce_kernel(const int4 *__restrict__ cnt_cm, int ppad_cm, const int4 *__restrict__ cnt_q, int ppad, int nq, int n_ref, int radius,
                                   int *__restrict__ mindist)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_ref) return;
  auto dist = [](const int4 c) { return ACGT ? c.x : (c.w - c.y); };   // both ACGT & differ | both valid & characters differ
  int md = min(dist(cnt_cm[r]), radius);                                  // idx_c pass, maxdist = radius
  if (md < radius) {
    md += min(dist(cnt_cm[(size_t)ppad_cm + r]), radius);                 // idx_m pass against the consensus
    if (md < radius) {
      const int c = md;                                                   // c_dist; *min_dist == c on loop entry
      int cur = c;
      for (int q = 0; q < nq && cur + c >= radius; q++) cur = min(dist(cnt_q[(size_t)q * ppad + r]), radius - c);
      md = cur + c;
    }
  }
  mindist[r] = md;
}

__global__ void snapshot_kernel(const int *__restrict__ T, int nq, int *__restrict__ snap)
{ // cq->max_incompatible = max over heaps (src/nearest.c:290-291)
  __shared__ int red[256];
  int v = -0x7fffffff;
  for (int i = threadIdx.x; i < nq; i += 256) v = max(v, T[i]);
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
  if (threadIdx.x == 0) *snap = red[0];
}

__global__ void init_state_kernel(int *__restrict__ T, int *__restrict__ n, int nq, int nchar)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) { T[i] = nchar; n[i] = 0; }    // src/nearest.c:375,387 ; src/min_heap.c:56
}

// untruncated score vectors of a batch, for parity tests: out[(i*nq+q)*6+s]
template <bool ACGT>
__global__ void batch_scores_kernel(const int4 *__restrict__ cnt, int ppad, const int4 *__restrict__ rt, const int *__restrict__ nonn,
                                    int r_begin, int n_ref, int nq, int *__restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
  if (i >= n_ref) return;
  const int r = r_begin + i;
  int S[6], m;
  assemble_scores<ACGT>(cnt[(size_t)q * ppad + r], rt[r], rt[r], nonn[r], S, m);
  for (int s = 0; s < 6; s++) out[((size_t)i * nq + q) * 6 + s] = S[s];
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
namespace {

// Packs one character row restricted to `keep` (nullable: keep everything inside [lo,hi)) into query-plane words:
// dst[(w4*4 + j)*NQ + plane].  is_poly marks query->idx columns (--acgt: fourth plane).
// host-side preparation of a query set is O(queries x columns) several times over: spread the independent pieces over threads
template <class F>
static void parallel_for(int n, F f)
{
  const unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  if (n < 32 || nt < 2) { for (int i = 0; i < n; i++) f(i); return; }
  std::atomic<int> next(0);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++) th.emplace_back([&]() { for (;;) { const int a = next.fetch_add(4); if (a >= n) break; for (int i = a; i < std::min(n, a + 4); i++) f(i); } });
  for (auto &x : th) x.join();
}

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

int launch_scan2(uvaia_gpu_ctx *c, const uint4 *tiles, const int *tot_tile0, long long tile_first, int n_tiles, int2 *out, int ppad, double bytes, hipStream_t stream,
                 int2 *tmin, int r_lo, int r_hi, int *mp)
{
  if (n_tiles <= 0) return 0;
  if (!stream) stream = c->stream;
  const int n_qtiles = (c->nq + c->qt - 1) / c->qt;
  dim3 grid(scan_grid_size(n_qtiles, (n_tiles + 3) / 4)), block(256);
  ScanEvt ev_{};
  if (c->profile) {
    HIPCHK(c, hipEventCreate(&ev_.a)); HIPCHK(c, hipEventCreate(&ev_.b));
    HIPCHK(c, hipEventRecord(ev_.a, stream));
  }
  const uint32_t *qp = c->acgt ? c->d_qp : c->d_qp2;
  if (c->scan_variant == 2) {
    const bool is_db = (tiles == c->d_db);
    const uint4 *ev = is_db ? c->d_db_ev : c->d_batch_ev, *poly = is_db ? c->d_db_poly : c->d_batch_poly;
    const int *tote = (is_db ? c->d_db_tote : c->d_batch_tote) + tile_first * 64;
    const uint32_t *grp = is_db ? c->d_db_grp : c->d_batch_grp;
    const int qt_first = c->act_q0 / 16, nqt3 = (c->act_q1 + 15) / 16 - qt_first;
    const int qblock = c->scan_qblock >= 0 ? c->scan_qblock : (nqt3 == 3 || nqt3 == 4 ? 1 : 0);      // measured: helps with one block of query tiles per reference tile (33..64 queries), not beyond
    dim3 grid3(qblock ? scan_grid_size((nqt3 + 3) / 4, n_tiles) : scan_grid_size(nqt3, (n_tiles + 3) / 4));
#define SCAN3_LAUNCH(A) hipLaunchKernelGGL((scan3_kernel<16, A>), grid3, block, c->scan_lds_pad, stream, ev, poly, tile_first, n_tiles, c->W4, c->NP4, c->NP4 + c->NR4, c->d_qpl, c->d_stream, c->d_sdir, grp, tote, tot_tile0, out, ppad, nqt3, tmin, r_lo, r_hi, mp, c->scan_parts, qt_first, qblock)
    if (c->acgt) SCAN3_LAUNCH(true); else SCAN3_LAUNCH(false);
#undef SCAN3_LAUNCH
    HIPCHK(c, hipGetLastError());
    if (c->profile) { HIPCHK(c, hipEventRecord(ev_.b, stream)); ev_.bytes = bytes; c->evts.push_back(ev_); }
    return 0;
  }
  if (c->scan_variant == 1) {
    constexpr int QTV = 16, RV = 2;
    const int nqtv = (c->nq + QTV - 1) / QTV;
    dim3 gridv(scan_grid_size(nqtv, (n_tiles + 4 * RV - 1) / (4 * RV)));
    if (c->acgt) hipLaunchKernelGGL((scan2v_kernel<QTV, RV, true>), gridv, block, 0, stream, tiles, tile_first, n_tiles, c->W4, c->W4pad, c->d_qv, out, ppad, nqtv);
    else         hipLaunchKernelGGL((scan2v_kernel<QTV, RV, false>), gridv, block, 0, stream, tiles, tile_first, n_tiles, c->W4, c->W4pad, c->d_qv, out, ppad, nqtv);
    HIPCHK(c, hipGetLastError());
    if (c->profile) { HIPCHK(c, hipEventRecord(ev_.b, stream)); ev_.bytes = bytes; c->evts.push_back(ev_); }
    return 0;
  }
#define LAUNCH(K, QT) hipLaunchKernelGGL((K<QT>), grid, block, 0, stream, tiles, tile_first, n_tiles, c->W4, qp, out, ppad, n_qtiles, tot_tile0)
  if (c->acgt) { switch (c->qt) { case 8: LAUNCH(scan2_acgt_kernel, 8); break; case 32: LAUNCH(scan2_acgt_kernel, 32); break; default: LAUNCH(scan2_acgt_kernel, 16); } }
  else         { switch (c->qt) { case 8: LAUNCH(scan2_iupac_kernel, 8); break; case 32: LAUNCH(scan2_iupac_kernel, 32); break; default: LAUNCH(scan2_iupac_kernel, 16); } }
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
  if (c->n_idx_c > 0) {   // with no constant-and-complete column every pre-score counter is zero (common: gappy query sets)
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
    int rc = launch_scan2(c, tiles, (tiles == c->d_db ? c->d_db_tot : c->d_batch_tot) + tile_first * 64, tile_first, n_tiles, c->d_cnt2, ppad, bytes, nullptr, c->d_tmin[0], r_begin, r_end, c->d_mp[0]);
    if (rc) return rc;
#define REPLAY2(A, B) hipLaunchKernelGGL((replay2_kernel<A, B>), dim3(c->nq), dim3(64), lds + (size_t)lq_words * 4 + 128, c->stream, c->d_cnt2, ppad, c->d_rt, c->d_tr, nonn_tile0, amb_tile0, r_begin, r_end, ord_base, \
                                    c->d_heap, c->d_n, c->d_T, c->d_snap, entered_tile0, c->k, tiles, tile_first, c->W4, c->d_qp, c->d_amb_q, c->d_stats, 0, c->scan_variant == 2 ? c->d_tmin[0] : (const int2 *)nullptr, \
                                    c->scan_variant == 2 ? c->d_mp[0] : (const int *)nullptr, lq_words, c->replay_prio, (tiles == c->d_db ? c->d_db_poly : c->d_batch_poly), c->NP4 + c->NR4, c->NP4, c->NR4, c->d_qrare)
    if (c->acgt) { if (c->n_idx_c > 0) REPLAY2(true, true); else REPLAY2(true, false); }
    else         { if (c->n_idx_c > 0) REPLAY2(false, true); else REPLAY2(false, false); }
#undef REPLAY2
  }
  HIPCHK(c, hipGetLastError());
  c->last_tiles = tiles; c->last_nonn = nonn_tile0; c->last_n = r_end - r_begin; c->last_rbegin = r_begin; c->last_ppad = ppad;
  c->last_ntiles = n_tiles; c->last_tile_first = tile_first;
  return 0;
}

// planes derived for the open query set (column-compressed scan) for the whole tiles that hold slots slot0 .. slot0 + n_ref - 1
int derive_rows(uvaia_gpu_ctx *c, uint4 *tiles, long long slot0, int n_ref)
{
  if (c->fullscan || n_ref <= 0) return 0;
  const bool is_db = (tiles == c->d_db);
  uint4 *ev = is_db ? c->d_db_ev : c->d_batch_ev, *poly = is_db ? c->d_db_poly : c->d_batch_poly;
  int *tote = is_db ? c->d_db_tote : c->d_batch_tote;
  uint32_t *grp = is_db ? c->d_db_grp : c->d_batch_grp;
  const long long t0 = slot0 / 64, t1 = (slot0 + n_ref - 1) / 64;
  const int nblk = (int)(t1 - t0 + 1);
  if (c->acgt) {
    hipLaunchKernelGGL((derive_ev_kernel<true>), dim3(nblk), dim3(256), 0, c->stream, tiles, t0, c->W4, c->d_cls, ev, tote, grp);
    if (c->NP4) hipLaunchKernelGGL((gather_poly_kernel<true>), dim3(nblk), dim3(64), 0, c->stream, tiles, t0, c->W4, c->NP4 + c->NR4, 0, c->d_cls + 3, 4, poly);
    if (c->NR4) hipLaunchKernelGGL((gather_poly_kernel<true>), dim3(nblk), dim3(64), 0, c->stream, tiles, t0, c->W4, c->NP4 + c->NR4, c->NP4, c->d_rmask, 1, poly);
  } else {
    hipLaunchKernelGGL((derive_ev_kernel<false>), dim3(nblk), dim3(256), 0, c->stream, tiles, t0, c->W4, c->d_cls, ev, tote, grp);
    if (c->NP4) hipLaunchKernelGGL((gather_poly_kernel<false>), dim3(nblk), dim3(64), 0, c->stream, tiles, t0, c->W4, c->NP4 + c->NR4, 0, c->d_cls + 3, 4, poly);
    if (c->NR4) hipLaunchKernelGGL((gather_poly_kernel<false>), dim3(nblk), dim3(64), 0, c->stream, tiles, t0, c->W4, c->NP4 + c->NR4, c->NP4, c->d_rmask, 1, poly);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}

// stage + pack n_ref rows (either scattered pointers or one pitched block) into `tiles` starting at slot0
int pack_rows(uvaia_gpu_ctx *c, const char *const *seq, const char *rows, size_t rows_pitch, const int *non_n, int n_ref,
              uint4 *tiles, int *nonn_dev, int *amb_dev, int *tot_dev, long long slot0)
{
  for (int done = 0; done < n_ref; done += PACK_CHUNK) {
    const int m = std::min(PACK_CHUNK, n_ref - done);
    for (int i = 0; i < m; i++) {
      const char *src = seq ? seq[done + i] : rows + (size_t)(done + i) * rows_pitch;
      if (!src) return fail(c, UVAIA_GPU_EINVAL, "NULL sequence at position %d", done + i);
      memcpy(c->h_stage + (size_t)i * c->pitch, src, (size_t)c->nchar);
    }
    HIPCHK(c, hipMemcpyAsync(c->d_stage, c->h_stage, (size_t)m * c->pitch, hipMemcpyHostToDevice, c->stream));
    const long long s0 = slot0 + done, t0 = s0 / 64, t1 = (s0 + m - 1) / 64;
    const int nblk = (int)(t1 - t0 + 1);
    int *nn_out = non_n ? nullptr : nonn_dev;
    if (c->acgt) hipLaunchKernelGGL((pack_refs_kernel<3>), dim3(nblk), dim3(256), 0, c->stream, c->d_stage, c->pitch, c->nchar, s0, m, c->W4, tiles, t0, nn_out, (int *)nullptr, tot_dev, c->d_err);
    else         hipLaunchKernelGGL((pack_refs_kernel<4>), dim3(nblk), dim3(256), 0, c->stream, c->d_stage, c->pitch, c->nchar, s0, m, c->W4, tiles, t0, nn_out, amb_dev, tot_dev, c->d_err);
    HIPCHK(c, hipGetLastError());
    if (non_n) HIPCHK(c, hipMemcpyAsync(nonn_dev + s0, non_n + done, (size_t)m * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // h_stage is reused by the next round
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
  void *dev[] = {c->d_qrare, c->d_rmask, c->d_batch_grp, c->d_db_grp, c->d_cls, c->d_qpl, c->d_stream, c->d_sdir, c->d_batch_ev, c->d_batch_poly, c->d_db_ev, c->d_db_poly, c->d_batch_tote, c->d_db_tote,
                 c->d_batch_tot, c->d_db_tot, c->d_cmrows, c->d_cnt_cm, c->d_mindist, c->d_qv, c->d_qp2, c->d_amb_q, c->d_batch_amb, c->d_db_amb, c->d_cnt2, c->d_stats, c->d_qp, c->d_cp, c->d_cpm, c->d_qpoly, c->d_heap, c->d_n, c->d_T, c->d_snap, c->d_err, c->d_batch, c->d_batch_nonn,
                 c->d_cnt, c->d_rt, c->d_tr, c->d_entered, c->d_stage, c->d_db, c->d_db_nonn};
  for (void *p : dev) if (p) hipFree(p);
  if (c->h_stage) hipHostFree(c->h_stage);
  for (int i = 0; i < NBUF; i++) { if (c->d_cntb[i]) hipFree(c->d_cntb[i]); if (c->d_tmin[i]) hipFree(c->d_tmin[i]); if (c->d_mp[i]) hipFree(c->d_mp[i]); }
  for (int i = 0; i < NBUF; i++) { if (c->scan_done[i]) hipEventDestroy(c->scan_done[i]); if (c->replay_done[i]) hipEventDestroy(c->replay_done[i]); }
  if (c->scan_stream) hipStreamDestroy(c->scan_stream);
  for (int i = 1; i < 3; i++) if (c->scan_streams[i]) hipStreamDestroy(c->scan_streams[i]);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int uvaia_gpu_open(uvaia_gpu_ctx **out, const uvaia_gpu_query *q, int heap_size, int device, size_t max_pool)
{
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
  const char *env_qt = getenv("UVAIA_GPU_QT");
  if (env_qt) { int v = atoi(env_qt); if (v == 8 || v == 16 || v == 32) c->qt = v; }
  const char *env_scan = getenv("UVAIA_GPU_SCAN");
  if (env_scan) c->scan_variant = (strcmp(env_scan, "lds") == 0) ? 1 : (strcmp(env_scan, "sgpr") == 0) ? 0 : 2;
  const char *env_full = getenv("UVAIA_GPU_FULLSCAN");
  c->fullscan = env_full && atoi(env_full) != 0;
  // the default scan keeps per-pair deficits in 16-bit halves (LDS counters): alignments of more than ~49 000 columns take the
  // four-counter scan instead (32-bit counts, same results, slower)
  if (c->nchar > 49000) c->fullscan = true;
  c->nq_pad = ((c->nq + 31) / 32) * 32;                      // multiple of every supported query tile
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
  std::vector<uint32_t> qp((size_t)c->nq_pad * row_words, 0u), cp(row_words, 0u), cpm(row_words, 0u), qpoly((size_t)c->nq_pad * row_words, 0u);
  int bad = 0;
  for (int i = 0; i < c->nq; i++) if (!q->seq[i]) { uvaia_gpu_close(c); return fail(nullptr, UVAIA_GPU_EINVAL, "query %d is NULL", i); }
  {
    std::atomic<int> first_bad(c->nq);
    std::vector<int> bad_byte((size_t)c->nq, 0);
    parallel_for(c->nq, [&](int i) {
      if (pack_query_row(code_tab, q->seq[i], c->nchar, lo, hi, nullptr, in_p.data(), c->acgt, c->NQ, qp.data() + (size_t)i * row_words, &bad_byte[(size_t)i]) ||
          pack_query_row(code_tab, q->seq[i], c->nchar, lo, hi, in_p.data(), in_p.data(), c->acgt, c->NQ, qpoly.data() + (size_t)i * row_words, &bad_byte[(size_t)i])) {
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
    {  // LDS-staging layout: [query tile of 16][w4 (padded to 8)][query in tile][word j] -> uint4 of the four planes
      c->W4pad = (c->W4 + 7) / 8 * 8;
      const int ntile = c->nq_pad / 16;
      std::vector<uint32_t> qvh((size_t)ntile * c->W4pad * 16 * 4 * 4, 0u);
      parallel_for(c->nq, [&](int i) { for (int w = 0; w < c->W4 * 4; w++) {
        uint32_t pl[4];
        if (c->acgt) { const uint32_t *s4 = qp.data() + (size_t)i * row_words + (size_t)w * 4; pl[0] = s4[0]; pl[1] = s4[1]; pl[2] = s4[2]; pl[3] = s4[3]; }
        else { const uint32_t *s6 = qp.data() + (size_t)i * row_words + (size_t)w * 6; const uint32_t one = s6[5];
               pl[0] = (s6[1] | s6[3]) & one; pl[1] = (s6[2] | s6[3]) & one; pl[2] = one; pl[3] = s6[4]; }
        uint32_t *d = qvh.data() + ((((size_t)(i / 16) * c->W4pad + (w >> 2)) * 16 + (i % 16)) * 4 + (w & 3)) * 4;
        d[0] = pl[0]; d[1] = pl[1]; d[2] = pl[2]; d[3] = pl[3];
      } });
      OPENCHK(hipMalloc(&c->d_qv, qvh.size() * 4)); OPENCHK(hipMemcpy(c->d_qv, qvh.data(), qvh.size() * 4, hipMemcpyHostToDevice));
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
        const char *er = getenv("UVAIA_GPU_RARE_MAX");
        c->rare_max = er ? atoi(er) : (c->nq < 64 ? 0 : std::min(64, std::max(4, c->nq / 64)));
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
      { const char *ep = getenv("UVAIA_GPU_SCAN_PARTS"); if (ep) c->scan_parts = atoi(ep); }
      { const char *ep = getenv("UVAIA_GPU_REPLAY_PRIO"); if (ep) c->replay_prio = atoi(ep); }
      { const char *ep = getenv("UVAIA_GPU_SCAN_LDS_PAD"); if (ep) c->scan_lds_pad = atoi(ep); }
      { const char *ep = getenv("UVAIA_GPU_REPLAY_LQ"); if (ep) c->replay_lq = atoi(ep); }
      { const char *ep = getenv("UVAIA_GPU_SUBSLICE_MINQ"); if (ep) c->subslice_minq = atoi(ep); }
      c->serial = getenv("UVAIA_GPU_SERIAL") != nullptr;
      { const char *ep = getenv("UVAIA_GPU_SCAN_QBLOCK"); if (ep) c->scan_qblock = atoi(ep); }
      // Next to a running scan (8 blocks x 16.9 KB of LDS per CU) a replay block with the 22 KB query row fits once per CU, without
      // it seven times: with many queries the replay then waits for LDS, not for work (5.48 -> 5.04 ms per config[1] search).
      if (c->replay_lq < 0) c->replay_lq = (c->nq < 256) ? 1 : 0;
      OPENCHK(hipMalloc(&c->d_cls, cls.size() * 4)); OPENCHK(hipMemcpy(c->d_cls, cls.data(), cls.size() * 4, hipMemcpyHostToDevice));
      OPENCHK(hipMalloc(&c->d_qrare, qrare.size() * 4)); OPENCHK(hipMemcpy(c->d_qrare, qrare.data(), qrare.size() * 4, hipMemcpyHostToDevice));
      OPENCHK(hipMalloc(&c->d_rmask, rmask.size() * 4)); OPENCHK(hipMemcpy(c->d_rmask, rmask.data(), rmask.size() * 4, hipMemcpyHostToDevice));
      OPENCHK(hipMalloc(&c->d_qpl, qpl.size() * 4)); OPENCHK(hipMemcpy(c->d_qpl, qpl.data(), qpl.size() * 4, hipMemcpyHostToDevice));
      // the item stream of every query tile (layout: see scan3_kernel)
      std::vector<uint32_t> strm, sdir((size_t)(c->nq_pad / 16) * 2, 0u);
      std::vector<uint8_t> rare_groups_needed((size_t)std::max(c->NR4, 1), 0);
      for (int t = 0; t < c->nq_pad / 16; t++) {
        sdir[(size_t)t * 2] = (uint32_t)strm.size();
        uint32_t nrec = 0;
        size_t prev_hdr = (size_t)-1;
        for (int g = 0; g < c->W4; g++) {
          const uint32_t fx = flg[((size_t)t * c->W4 + g) * 2], fy = flg[((size_t)t * c->W4 + g) * 2 + 1];
          if ((fx | fy) == 0u) continue;
          const uint32_t fa = (fx | (fx >> 16)) & 0xFFFFu;
          // a dirty query whose non-ACGT / invalid sites of this group all lie in ONE 32-column word (an isolated N or ambiguity
          // code: more than half of the partially dirty cases) gets a 4-dword "word item" instead of the 12-dword general one
          uint32_t f4 = 0, f1 = 0;
          for (uint32_t m = fa; m; m &= m - 1) {
            const int q = __builtin_ctz(m);
            const uint32_t *src = qcv.data() + (size_t)(t * 16 + q) * crow + (size_t)g * 8;
            int words = 0;
            for (int j = 0; j < 4; j++) words += (src[j] | src[4 + j]) != 0u;
            if (words == 1) f1 |= 1u << q; else f4 |= 1u << q;
          }
          uint32_t nw[4] = {0, 0, 0, 0};                // word items per word of the group, listed word by word
          auto word_of = [&](int q) { const uint32_t *src = qcv.data() + (size_t)(t * 16 + q) * crow + (size_t)g * 8; int j = 0; while (!(src[j] | src[4 + j])) j++; return j; };
          for (uint32_t m = f1; m; m &= m - 1) nw[word_of(__builtin_ctz(m))]++;
          const uint32_t hw0 = (uint32_t)g * 2048u | ((fx & 0xFFFFu) ? 1u : 0u) | ((fx >> 16) ? 2u : 0u) | (fy ? 4u : 0u);
          prev_hdr = strm.size();
          strm.push_back(hw0);
          strm.push_back((uint32_t)(__builtin_popcount(fy) + 3) / 4u | nw[0] << 4 | nw[1] << 9 | nw[2] << 14 | nw[3] << 19);
          strm.push_back((uint32_t)__builtin_popcount(f4)); strm.push_back(0u);
          for (uint32_t m = fy; m; m &= m - 1) strm.push_back((uint32_t)__builtin_ctz(m) * 256u);
          while (strm.size() & 3) strm.push_back(16u * 256u);                                  // scratch row
          for (uint32_t m = f4; m; m &= m - 1) {
            const int q = __builtin_ctz(m);
            const uint32_t *src = qcv.data() + (size_t)(t * 16 + q) * crow + (size_t)g * 8;
            strm.insert(strm.end(), src, src + 8);
            strm.push_back((uint32_t)q * 256u); strm.push_back(0u); strm.push_back(0u); strm.push_back(0u);
          }
          for (int j = 0; j < 4; j++)
            for (uint32_t m = f1; m; m &= m - 1) {
              const int q = __builtin_ctz(m);
              if (word_of(q) != j) continue;
              const uint32_t *src = qcv.data() + (size_t)(t * 16 + q) * crow + (size_t)g * 8;
              strm.push_back(src[j]); strm.push_back(src[4 + j]); strm.push_back((uint32_t)q * 256u); strm.push_back(0u);
            }
          strm[prev_hdr + 3] = (uint32_t)(strm.size() - prev_hdr);
          nrec++;
        }
        // rare records: { byte offset of the rare group's planes in the tile's gathered planes, word-item counts << 4, 0, 0 }
        // + items { sites, their lo bits, their hi bits, LDS offset } listed word by word
        uint32_t nrare = 0;
        for (int r4 = 0; r4 < c->NR4; r4++) {
          uint32_t nw[4] = {0, 0, 0, 0};
          for (int q = 0; q < 16; q++) for (const RareWord &rw : rare_q[(size_t)t * 16 + q]) if ((rw.word >> 2) == r4) nw[rw.word & 3]++;
          if (!(nw[0] | nw[1] | nw[2] | nw[3])) continue;
          strm.push_back((uint32_t)(c->NP4 + r4) * 3072u); strm.push_back(nw[0] << 4 | nw[1] << 9 | nw[2] << 14 | nw[3] << 19); strm.push_back(0u); strm.push_back(0u);
          for (int j = 0; j < 4; j++)
            for (int q = 0; q < 16; q++) for (const RareWord &rw : rare_q[(size_t)t * 16 + q]) if (rw.word == r4 * 4 + j) {
              strm.push_back(rw.m); strm.push_back(rw.l); strm.push_back(rw.h); strm.push_back((uint32_t)q * 256u);
            }
          nrare++; rare_groups_needed[(size_t)r4] = 1;
        }
        sdir[(size_t)t * 2 + 1] = nrec | (nrare << 16);
      }
      for (uint8_t u : rare_groups_needed) c->need_r_groups += u;
      strm.resize(strm.size() + 64, 0u);                                // the kernel prefetches one item past the end
      OPENCHK(hipMalloc(&c->d_stream, strm.size() * 4)); OPENCHK(hipMemcpy(c->d_stream, strm.data(), strm.size() * 4, hipMemcpyHostToDevice));
      OPENCHK(hipMalloc(&c->d_sdir, sdir.size() * 4)); OPENCHK(hipMemcpy(c->d_sdir, sdir.data(), sdir.size() * 4, hipMemcpyHostToDevice));
    }
    OPENCHK(hipMalloc(&c->d_amb_q, ambq.size() * sizeof(int))); OPENCHK(hipMemcpy(c->d_amb_q, ambq.data(), ambq.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  OPENCHK(hipMalloc(&c->d_qpoly, qpoly.size() * 4)); OPENCHK(hipMemcpy(c->d_qpoly, qpoly.data(), qpoly.size() * 4, hipMemcpyHostToDevice));
  OPENCHK(hipMalloc(&c->d_cp, cp.size() * 4)); OPENCHK(hipMemcpy(c->d_cp, cp.data(), cp.size() * 4, hipMemcpyHostToDevice));
  OPENCHK(hipMalloc(&c->d_cpm, cpm.size() * 4)); OPENCHK(hipMemcpy(c->d_cpm, cpm.data(), cpm.size() * 4, hipMemcpyHostToDevice));
  OPENCHK(hipMalloc(&c->d_cmrows, 32 * row_words * 4)); OPENCHK(hipMemset(c->d_cmrows, 0, 32 * row_words * 4));
  OPENCHK(hipMemcpy(c->d_cmrows, cp.data(), row_words * 4, hipMemcpyHostToDevice));
  OPENCHK(hipMemcpy(c->d_cmrows + row_words, cpm.data(), row_words * 4, hipMemcpyHostToDevice));

  // ---- state
  OPENCHK(hipMalloc(&c->d_heap, (size_t)c->nq * (c->k + 1) * HEAP_ENTRY * sizeof(int)));
  OPENCHK(hipMalloc(&c->d_n, (size_t)c->nq * sizeof(int)));
  OPENCHK(hipMalloc(&c->d_T, (size_t)c->nq * sizeof(int)));
  OPENCHK(hipMalloc(&c->d_snap, sizeof(int)));
  OPENCHK(hipMalloc(&c->d_err, sizeof(int)));
  OPENCHK(hipMemset(c->d_err, 0, sizeof(int)));
  // ---- batch buffers
  const size_t tile_u4 = (size_t)c->W4 * c->P * 64;
  OPENCHK(hipMalloc(&c->d_batch, (c->pool_pad / 64) * tile_u4 * sizeof(uint4)));
  OPENCHK(hipMemset(c->d_batch, 0, (c->pool_pad / 64) * tile_u4 * sizeof(uint4)));
  OPENCHK(hipMalloc(&c->d_batch_nonn, c->pool_pad * sizeof(int)));
  OPENCHK(hipMemset(c->d_batch_nonn, 0, c->pool_pad * sizeof(int)));
  OPENCHK(hipMalloc(&c->d_batch_ev, (c->pool_pad / 64) * (size_t)c->W4 * 2 * 64 * sizeof(uint4)));
  OPENCHK(hipMalloc(&c->d_batch_grp, (c->pool_pad / 64) * (size_t)c->W4 * 64 * sizeof(uint32_t)));
  OPENCHK(hipMalloc(&c->d_batch_poly, (c->pool_pad / 64) * (size_t)std::max(c->NP4 + c->NR4, 1) * 3 * 64 * sizeof(uint4)));
  OPENCHK(hipMalloc(&c->d_batch_tote, c->pool_pad * sizeof(int)));
  OPENCHK(hipMalloc(&c->d_batch_tot, c->pool_pad * sizeof(int)));
  OPENCHK(hipMemset(c->d_batch_tot, 0, c->pool_pad * sizeof(int)));
  OPENCHK(hipMalloc(&c->d_batch_amb, c->pool_pad * AMB_ROW * sizeof(int)));
  OPENCHK(hipMemset(c->d_batch_amb, 0, c->pool_pad * AMB_ROW * sizeof(int)));
  if (!c->fullscan) { OPENCHK(hipMalloc(&c->d_cnt2, (size_t)c->nq_pad * c->pool_pad * sizeof(int2))); c->slice_cap[0] = (size_t)c->nq_pad * c->pool_pad; }
  OPENCHK(hipMalloc(&c->d_tmin[0], (size_t)c->nq_pad * (c->pool_pad / 64) * sizeof(int2)));
  { const char *env_sub = getenv("UVAIA_GPU_SUBSLICE"); if (env_sub && atol(env_sub) >= 64) c->subslice = (size_t)atol(env_sub); }
  if (c->acgt && !c->fullscan && c->scan_variant == 2) OPENCHK(hipMalloc(&c->d_mp[0], (size_t)c->nq_pad * c->pool_pad * sizeof(int)));
  OPENCHK(hipMalloc(&c->d_stats, 4 * sizeof(unsigned long long)));
  OPENCHK(hipMemset(c->d_stats, 0, 4 * sizeof(unsigned long long)));
  OPENCHK(hipMalloc(&c->d_rt, c->pool_pad * sizeof(int4)));
  OPENCHK(hipMalloc(&c->d_tr, c->pool_pad * sizeof(int4)));
  OPENCHK(hipMemset(c->d_rt, 0, c->pool_pad * sizeof(int4)));      // stay zero when idx_c is empty (the pre-score is skipped)
  OPENCHK(hipMemset(c->d_tr, 0, c->pool_pad * sizeof(int4)));
  OPENCHK(hipMalloc(&c->d_entered, c->pool_pad)); c->entered_cap = c->pool_pad;
  OPENCHK(hipMemset(c->d_entered, 0, c->pool_pad));
  OPENCHK(hipMalloc(&c->d_stage, (size_t)PACK_CHUNK * c->pitch));
  OPENCHK(hipHostMalloc(&c->h_stage, (size_t)PACK_CHUNK * c->pitch, hipHostMallocDefault));
  memset(c->h_stage, 'N', (size_t)PACK_CHUNK * c->pitch);
  const size_t lds = (size_t)(c->k + 1) * HEAP_ENTRY * sizeof(int) + 128;      // heap + the listed-words bitmap of replay2_kernel
  if (lds > 64 * 1024) {
    OPENCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&replay_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    OPENCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&replay_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    OPENCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&replay2_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    OPENCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&replay2_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    OPENCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&replay2_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    OPENCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&replay2_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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

int uvaia_gpu_set_query_tile(uvaia_gpu_ctx *c, int qt)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (qt == 0) qt = 16;
  if (qt != 8 && qt != 16 && qt != 32) return fail(c, UVAIA_GPU_EINVAL, "query tile must be 8, 16 or 32");
  c->qt = qt;
  return 0;
}

int uvaia_gpu_push(uvaia_gpu_ctx *c, const char *const *seq, const int *non_n, int n_ref, int64_t ordinal0, uint8_t *entered)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (n_ref < 0 || (n_ref > 0 && !seq)) return fail(c, UVAIA_GPU_EINVAL, "bad batch");
  if ((size_t)n_ref > c->max_pool) return fail(c, UVAIA_GPU_ESTATE, "batch of %d exceeds max_pool %zu", n_ref, c->max_pool);
  if (c->act_q0 != 0 || c->act_q1 != c->nq) return fail(c, UVAIA_GPU_ESTATE, "streamed batches act on the whole query set: query shards use the resident calls");
  if (n_ref == 0) return 0;
  int rc = pack_rows(c, seq, nullptr, 0, non_n, n_ref, c->d_batch, c->d_batch_nonn, c->d_batch_amb, c->d_batch_tot, 0);
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
  HIPCHK(c, hipMalloc(&c->d_db_ev, tiles * (size_t)c->W4 * 2 * 64 * sizeof(uint4)));
  HIPCHK(c, hipMalloc(&c->d_db_grp, tiles * (size_t)c->W4 * 64 * sizeof(uint32_t)));
  HIPCHK(c, hipMalloc(&c->d_db_poly, tiles * (size_t)std::max(c->NP4 + c->NR4, 1) * 3 * 64 * sizeof(uint4)));
  HIPCHK(c, hipMalloc(&c->d_db_tote, tiles * 64 * sizeof(int)));
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
  if (nqt < 63) sub = std::min(pool, (sub * 63 / (size_t)std::max(nqt, 1) + 63) / 64 * 64);
  if (nq_act < c->subslice_minq && (c->n_idx_c > 0 || nqt < 4)) sub = pool;
  for (size_t a = first; a < first + n; a += pool) {
    const size_t pe = std::min(first + n, a + pool);
    // near-equal slices (multiples of 64), as many as the pool holds sub-slice lengths, rounded: a pool of 1.05 sub-slices is
    // one launch, not a full one plus a sliver whose launch latency and replay would sit on the critical path
    const size_t len = pe - a, ns = std::max<size_t>(1, (len + sub / 2) / sub);
    const size_t each = ((len + ns - 1) / ns + 63) / 64 * 64;
    for (size_t x = a; x < pe; x += each) subs.push_back({x, std::min(each, pe - x), x == a});
  }
  return subs;
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
    const char *es = getenv("UVAIA_GPU_SCAN_STREAMS");
    c->scan_nstreams = es ? std::min(3, std::max(1, atoi(es))) : (waves && waves < 30000 ? 3 : 1);
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
  if (c->acgt) hipLaunchKernelGGL((batch_scores_kernel<true>), grid, dim3(256), 0, c->stream, c->d_cnt, c->last_ppad, c->d_rt, c->last_nonn, c->last_rbegin, n_ref, c->nq, d_out);
  else         hipLaunchKernelGGL((batch_scores_kernel<false>), grid, dim3(256), 0, c->stream, c->d_cnt, c->last_ppad, c->d_rt, c->last_nonn, c->last_rbegin, n_ref, c->nq, d_out);
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
  unsigned long long h[4] = {0, 0, 0, 0};
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
    const size_t rows = std::min<size_t>((size_t)c->nq_pad, ((size_t)c->act_q1 + 15) / 16 * 16);     // the scan writes whole query tiles up to the last active one
    const size_t need = std::max(rows * ppad_, buf == 0 ? c->slice_cap[0] : (size_t)0);
    if (need > c->slice_cap[buf] || !c->d_tmin[buf]) {
      for (int i_ = 0; i_ < 3; i_++) if (c->scan_streams[i_]) HIPCHK(c, hipStreamSynchronize(c->scan_streams[i_]));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      const size_t cap = std::max(need, (size_t)c->nq_pad * c->pool_pad);
      int2 *&cb = buf ? c->d_cntb[buf] : c->d_cnt2;
      if (cap > c->slice_cap[buf] || !cb) { if (cb) hipFree(cb); cb = nullptr; HIPCHK(c, hipMalloc(&cb, cap * sizeof(int2))); }
      if (c->d_tmin[buf]) hipFree(c->d_tmin[buf]);
      c->d_tmin[buf] = nullptr;
      HIPCHK(c, hipMalloc(&c->d_tmin[buf], (cap / 64) * sizeof(int2)));
      if (c->d_mp[0] || (c->acgt && c->scan_variant == 2)) { if (c->d_mp[buf]) hipFree(c->d_mp[buf]); c->d_mp[buf] = nullptr; HIPCHK(c, hipMalloc(&c->d_mp[buf], cap * sizeof(int))); }
      c->slice_cap[buf] = cap;
    }
  }
  hipStream_t ss = c->scan_streams[c->scan_nstreams > 1 ? (c->scan_rr++ % c->scan_nstreams) : 0];
  if (c->replay_recorded[buf]) HIPCHK(c, hipStreamWaitEvent(ss, c->replay_done[buf], 0));   // the buffer's previous reader
  const long long tf = (long long)(first / 64);
  const int n_tiles = n ? (int)((first + n + 63) / 64 - first / 64) : 0;
  c->slice_tf[buf] = tf; c->slice_tiles[buf] = n_tiles;
  c->slice_rb[buf] = (int)(first - (size_t)tf * 64); c->slice_re[buf] = c->slice_rb[buf] + (int)n;
  c->slice_scanned[buf] = true; c->slice_cons_done[buf] = false;
  const double bytes = (double)n * (double)c->W4 * 16.0 * c->P + (double)c->nq * (double)c->W4 * 16.0 * c->P;
  int rc = launch_scan2(c, c->d_db, c->d_db_tot + tf * 64, tf, n_tiles, buf ? c->d_cntb[buf] : c->d_cnt2, n_tiles * 64, bytes, ss, c->d_tmin[buf], c->slice_rb[buf], c->slice_re[buf], c->d_mp[buf]);
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
  if (c->n_idx_c > 0 && !c->slice_cons_done[buf]) {     // once per slice: the pre-score does not depend on the query
    if (c->acgt) hipLaunchKernelGGL((consensus_kernel<true>), dim3((n_tiles + 3) / 4), dim3(256), 0, c->stream, c->d_db, tf, n_tiles, c->W4, c->d_cp, c->d_snap, c->d_rt, c->d_tr);
    else         hipLaunchKernelGGL((consensus_kernel<false>), dim3((n_tiles + 3) / 4), dim3(256), 0, c->stream, c->d_db, tf, n_tiles, c->W4, c->d_cp, c->d_snap, c->d_rt, c->d_tr);
    c->slice_cons_done[buf] = true;
  }
  const size_t lds = (size_t)(c->k + 1) * HEAP_ENTRY * sizeof(int);
  const int lq_words = (c->replay_lq && !c->acgt && lds + (size_t)c->W4 * 4 * 6 * 4 + 128 <= 64 * 1024) ? c->W4 * 4 * 6 : 0;
  const int2 *cnt = buf ? c->d_cntb[buf] : c->d_cnt2;
  const int *nonn = c->d_db_nonn + tf * 64, *amb = c->d_db_amb + tf * 64 * AMB_ROW;
  uint8_t *ent = c->d_entered + tf * 64;
#define REPLAY2(A, B) hipLaunchKernelGGL((replay2_kernel<A, B>), dim3(q1 - q0), dim3(64), lds + (size_t)lq_words * 4 + 128, c->stream, cnt, ppad, c->d_rt, c->d_tr, nonn, amb, rb, re, (long long)ordinal0, \
                                  c->d_heap, c->d_n, c->d_T, c->d_snap, ent, c->k, c->d_db, tf, c->W4, c->d_qp, c->d_amb_q, c->d_stats, q0, c->scan_variant == 2 ? c->d_tmin[buf] : (const int2 *)nullptr, \
                                  c->scan_variant == 2 ? c->d_mp[buf] : (const int *)nullptr, lq_words, c->replay_prio, c->d_db_poly, c->NP4 + c->NR4, c->NP4, c->NR4, c->d_qrare)
  if (c->acgt) { if (c->n_idx_c > 0) REPLAY2(true, true); else REPLAY2(true, false); }
  else         { if (c->n_idx_c > 0) REPLAY2(false, true); else REPLAY2(false, false); }
#undef REPLAY2
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->replay_done[buf], c->stream));
  c->replay_recorded[buf] = true;
  c->last_tiles = c->d_db; c->last_nonn = nonn; c->last_n = re - rb; c->last_rbegin = rb; c->last_ppad = ppad; c->last_ntiles = n_tiles; c->last_tile_first = tf;
  return 0;
}

int uvaia_gpu_slice_buffers(void) { return NBUF; }

int uvaia_gpu_slice_replay(uvaia_gpu_ctx *c, int buf, int64_t ordinal0, int stripe_start)
{ return c ? uvaia_gpu_slice_replay_range(c, buf, ordinal0, c->act_q0, c->act_q1, stripe_start) : UVAIA_GPU_EINVAL; }

int uvaia_gpu_set_active_queries(uvaia_gpu_ctx *c, int q0, int q1)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (q0 < 0 || q1 > c->nq || q1 <= q0 || (q0 % 16)) return fail(c, UVAIA_GPU_EINVAL, "active queries [%d,%d): need 0 <= q0 < q1 <= %d and q0 a multiple of 16", q0, q1, c->nq);
  if (c->fullscan || c->scan_variant != 2) { if (q0 != 0 || q1 != c->nq) return fail(c, UVAIA_GPU_ESTATE, "query shards need the default scan"); }
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

int uvaia_gpu_ball(uvaia_gpu_ctx *c, const char *const *seq, int n_ref, int radius, int *mindist)
{
  if (!c) return UVAIA_GPU_EINVAL;
  if (n_ref < 0 || (n_ref > 0 && (!seq || !mindist))) return fail(c, UVAIA_GPU_EINVAL, "bad batch");
  if ((size_t)n_ref > c->max_pool) return fail(c, UVAIA_GPU_ESTATE, "batch of %d exceeds max_pool %zu", n_ref, c->max_pool);
  if (c->act_q0 != 0 || c->act_q1 != c->nq) return fail(c, UVAIA_GPU_ESTATE, "the radius search acts on the whole query set");
  if (n_ref == 0) return 0;
  int rc = pack_rows(c, seq, nullptr, 0, nullptr, n_ref, c->d_batch, c->d_batch_nonn, c->d_batch_amb, c->d_batch_tot, 0);
  if (rc) return rc;
  const int n_tiles = (n_ref + 63) / 64, ppad = n_tiles * 64;
  rc = ensure_cnt4(c, (size_t)c->nq_pad * c->pool_pad); if (rc) return rc;
  if (c->cnt_cm_cap < (size_t)32 * c->pool_pad) {
    if (c->d_cnt_cm) HIPCHK(c, hipFree(c->d_cnt_cm));
    HIPCHK(c, hipMalloc(&c->d_cnt_cm, (size_t)32 * c->pool_pad * sizeof(int4))); c->cnt_cm_cap = (size_t)32 * c->pool_pad;
    if (!c->d_mindist) HIPCHK(c, hipMalloc(&c->d_mindist, c->pool_pad * sizeof(int)));
  }
  const bool prof = c->profile; c->profile = false;     // these launches are not the nearest-neighbour scan the statistics describe
  rc = launch_scan(c, c->d_batch, 0, n_tiles, c->d_cmrows, 2, c->d_cnt_cm, ppad, 0.0);
  if (!rc) rc = launch_scan(c, c->d_batch, 0, n_tiles, c->d_qpoly, c->nq, c->d_cnt, ppad, 0.0);
  c->profile = prof;
  if (rc) return rc;
  if (c->acgt) hipLaunchKernelGGL((ball_reduce_kernel<true>), dim3((n_ref + 255) / 256), dim3(256), 0, c->stream, c->d_cnt_cm, ppad, c->d_cnt, ppad, c->nq, n_ref, radius, c->d_mindist);
  else         hipLaunchKernelGGL((ball_reduce_kernel<false>), dim3((n_ref + 255) / 256), dim3(256), 0, c->stream, c->d_cnt_cm, ppad, c->d_cnt, ppad, c->nq, n_ref, radius, c->d_mindist);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(mindist, c->d_mindist, (size_t)n_ref * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"

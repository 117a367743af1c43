// uvaia_align.hip -- MI355X (gfx950 / CDNA4) gap-affine wavefront aligner behind include/uvaia_align.h.
//
// What it replaces in the reference (paths under /root/reference): the OpenMP loop of src/align.c:224-233 over
// align_query (src/align.c:357-364) and update_query_aligned (src/align.c:366-390), and the per-thread WFA aligners of
// new_queue (src/align.c:286-313).  The WFA library is an absent submodule; the algorithm is the published one
// (Marco-Sola et al. 2021) as oracle/wfa_oracle.h states it, and this file reproduces that statement bit for bit.
//
// Design ("uvaialign"):
//   * one wavefront (64 lanes) per query, persistent: a block takes the next query from an atomic counter until none is left.
//     The work of one query is a chain of a few thousand dependent steps (one per score), each a handful of cells wide most of
//     the time: throughput comes from thousands of queries in flight, not from width.  No barrier between waves, no LDS tiles.
//   * lane = diagonal.  A step computes I, D, M of 64 diagonals at a time from the wavefronts of score - e, score - o - e,
//     score - x, extends M along the diagonal in registers (byte compares; long runs are extended by the whole wave, 256
//     characters per round trip) and stores the three offsets once.  Reference and queries are read through L2.
//   * every wavefront stays in the block's share of the workspace (the backtrace needs all of them); a ring of the last 64
//     headers lives in LDS so that a step finds its three source wavefronts without a trip to memory.
//   * the backtrace runs on the same wave right after the last step and writes the projected row (ref_len characters)
//     directly: match runs are copied 64 characters at a time, the five candidate predecessors of a step are fetched by
//     five lanes at once.
//   * a query that needs more wavefront memory than the block's share is flagged and run again with a larger share.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/uvaia_align.h"

namespace {

constexpr int WFA_NULL = -10;            // offset of a diagonal a wavefront does not hold (oracle/wfa_oracle.c)
constexpr int RING = 64;                 // scores whose headers stay in LDS; penalties are below this
constexpr int HDR_INTS = 8;              // per score, at the end of the block's share, growing downwards
enum { ST_OK = 0, ST_OVERFLOW = 1, ST_MAXSCORE = 2, ST_BACKTRACE = 3 };

struct WfaParams { int x, oe, e, min_wf_len, max_dist_thr, max_score; };

// header of the wavefronts of one score: the three arrays share their limits (the reduction trims M and hands its limits to
// I and D); flags bit 0 = M exists, bit 1 = I, bit 2 = D; arrays of w = hi_base - lo_base + 1 offsets at word `off`: M, then I, then D
struct Hdr { int lo, hi, lo_base, flags; uint32_t off; int w; };

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int fetch(const uint32_t *A, const Hdr &h, int which, int k)
{ // which: 0 = M, 1 = I, 2 = D.  A missing wavefront or a diagonal outside its limits gives the null offset
  if (!((h.flags >> which) & 1) || k < h.lo || k > h.hi) return WFA_NULL;
  const int slot = which == 0 ? 0 : (which == 1 ? 1 : ((h.flags >> 1) & 1) + 1);
  return (int)A[h.off + (uint32_t)slot * (uint32_t)h.w + (uint32_t)(k - h.lo_base)];
}

__device__ __forceinline__ int dist_to_end(int plen, int tlen, int offset, int k)
{
  return max(plen - (offset - k), tlen - offset);
}

__global__ __launch_bounds__(64) void wfa_align_kernel(const uint8_t *__restrict__ ref, int plen, const uint8_t *__restrict__ seqs, const long long *__restrict__ seq_off,
                                                        const int *__restrict__ todo, int n_todo, uint8_t *__restrict__ aln, size_t aln_pitch, int *__restrict__ score_out,
                                                        int *__restrict__ status_out, unsigned long long *__restrict__ cells_out, uint32_t *__restrict__ arena,
                                                        unsigned long long share_words, int *next_query, WfaParams P)
{
  __shared__ int ring[RING][HDR_INTS];
  const int lane = threadIdx.x;
  uint32_t *A = arena + (size_t)blockIdx.x * share_words;
  int *H = reinterpret_cast<int *>(A + share_words);          // header of score s: H - (s + 1) * HDR_INTS
  unsigned long long cells_total = 0;

  for (;;) {
    int qi = 0;
    if (lane == 0) qi = atomicAdd(next_query, 1);
    qi = rfl(qi);
    if (qi >= n_todo) break;                                  // every wave reaches this: the counter only grows
    const int q = todo ? todo[qi] : qi;
    const uint8_t *text = seqs + seq_off[q];
    const int tlen = (int)(seq_off[q + 1] - seq_off[q]);
    const int alignment_k = tlen - plen;
    uint8_t *row = aln + (size_t)q * aln_pitch;

    for (int i = lane; i < RING * HDR_INTS; i += 64) (&ring[0][0])[i] = 0;
    __syncthreads();

    unsigned long long used = 0, cells = 0;                    // words of the share taken by offsets
    int score = 0, status = ST_OK;
    bool reached = false;
    // ---------------- forward: one step per score ----------------
    for (;;) {
      // source wavefronts (paper eq. 3): M of score - x, M of score - o - e, I and D of score - e
      Hdr hs{}, hg{}, he{};
      bool have = false;
      int lo = 0, hi = 0;
      if (score == 0) { have = true; }
      else {
        auto load = [&](int s, Hdr &h) {
          if (s < 0) { h.flags = 0; return; }
          const int *r = ring[s & (RING - 1)];
          h.lo = r[0]; h.hi = r[1]; h.lo_base = r[2]; h.flags = r[3]; h.off = (uint32_t)r[4]; h.w = r[5];
        };
        load(score - P.x, hs); load(score - P.oe, hg); load(score - P.e, he);
        const bool n_sub = !(hs.flags & 1), n_gap = !(hg.flags & 1), n_i = !(he.flags & 2), n_d = !(he.flags & 4);
        if (!(n_sub && n_gap && n_i && n_d)) {
          have = true;
          lo = min(min(n_sub ? 1 : hs.lo, n_gap ? 1 : hg.lo), min(n_i ? 1 : he.lo, n_d ? 1 : he.lo)) - 1;
          hi = max(max(n_sub ? -1 : hs.hi, n_gap ? -1 : hg.hi), max(n_i ? -1 : he.hi, n_d ? -1 : he.hi)) + 1;
        }
      }
      const size_t hdr_words = (size_t)(score + 1) * HDR_INTS;
      if (used + hdr_words > share_words) { status = ST_OVERFLOW; break; }
      int *hdr_out = H - (size_t)(score + 1) * HDR_INTS;
      int *ring_out = ring[score & (RING - 1)];
      if (!have) {
        if (lane < HDR_INTS) { ring_out[lane] = 0; hdr_out[lane] = 0; }
      } else {
        const bool has_i = score > 0 && ((hg.flags & 1) || (he.flags & 2)), has_d = score > 0 && ((hg.flags & 1) || (he.flags & 4));
        const int w = hi - lo + 1;
        const size_t need = (size_t)w * (1 + (has_i ? 1 : 0) + (has_d ? 1 : 0));
        if (used + need + hdr_words > share_words) { status = ST_OVERFLOW; break; }
        const uint32_t off = (uint32_t)used;
        used += need; cells += (unsigned)w;
        uint32_t *out_m = A + off, *out_i = out_m + w, *out_d = out_m + (size_t)w * (has_i ? 2 : 1);
        int min_distance = max(plen, tlen);
        bool hit_end = false;
        for (int k0 = lo; k0 <= hi; k0 += 64) {
          const int k = k0 + lane;
          const bool act = k <= hi;
          int m = 0;
          if (score > 0) {
            int sub = fetch(A, hs, 0, k);
            if ((hs.flags & 1) && k >= hs.lo && k <= hs.hi) sub++;        // the + 1 belongs to a fetched value only
            m = sub;
            if (has_i) { const int ins = max(fetch(A, hg, 0, k - 1), fetch(A, he, 1, k - 1)) + 1; if (act) out_i[k - lo] = (uint32_t)ins; m = max(m, ins); }
            if (has_d) { const int del = max(fetch(A, hg, 0, k + 1), fetch(A, he, 2, k + 1));     if (act) out_d[k - lo] = (uint32_t)del; m = max(m, del); }
          }
          // exact extension along the diagonal (paper algorithm 2): a few characters per lane, long runs by the whole wave
          int v = m - k, h = m;
          bool go = act && (unsigned)h < (unsigned)tlen && (unsigned)v < (unsigned)plen;
          for (int i = 0; i < 4; i++) {
            if (!__any(go)) break;
            if (go) { if (ref[v] == text[h]) { v++; h++; m++; go = v < plen && h < tlen; } else go = false; }
          }
          unsigned long long more = __ballot(go);
          while (more) {
            const int j = __builtin_ctzll(more);
            more &= more - 1;
            const int vj = __shfl(v, j), hj = __shfl(h, j);
            int ext = 0;
            for (;;) {
              unsigned long long bad[4];
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const int pv = vj + ext + u * 64 + lane, ph = hj + ext + u * 64 + lane;
                const bool ok = pv < plen && ph < tlen && ref[pv] == text[ph];
                bad[u] = __ballot(!ok);
              }
              if (bad[0]) { ext += __builtin_ctzll(bad[0]); break; }
              if (bad[1]) { ext += 64 + __builtin_ctzll(bad[1]); break; }
              if (bad[2]) { ext += 128 + __builtin_ctzll(bad[2]); break; }
              if (bad[3]) { ext += 192 + __builtin_ctzll(bad[3]); break; }
              ext += 256;
            }
            if (lane == j) m += ext;
          }
          if (act) {
            out_m[k - lo] = (uint32_t)m;
            min_distance = min(min_distance, dist_to_end(plen, tlen, m, k));
            if (k == alignment_k && m >= tlen) hit_end = true;
          }
        }
        reached = __any(hit_end);
        __syncthreads();                                        // the offsets just stored are read by other lanes from here on
        // adaptive reduction (paper section 2.4): trim both ends of a long wavefront, never across the end cell's diagonal
        int rlo = lo, rhi = hi;
        if (P.min_wf_len > 0 && w >= P.min_wf_len) {
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) min_distance = min(min_distance, __shfl_xor(min_distance, o));
          const int top_limit = min(alignment_k - 1, hi);
          if (lo < top_limit) {
            rlo = top_limit;
            for (int k0 = lo; k0 < top_limit; k0 += 64) {
              const int k = k0 + lane;
              const bool keep = k < top_limit && dist_to_end(plen, tlen, (int)out_m[k - lo], k) - min_distance <= P.max_dist_thr;
              const unsigned long long b = __ballot(keep);
              if (b) { rlo = k0 + __builtin_ctzll(b); break; }
            }
          }
          const int bottom_limit = max(alignment_k + 1, rlo);
          if (hi > bottom_limit) {
            rhi = bottom_limit;
            for (int k0 = hi; k0 > bottom_limit; k0 -= 64) {
              const int k = k0 - lane;
              const bool keep = k > bottom_limit && dist_to_end(plen, tlen, (int)out_m[k - lo], k) - min_distance <= P.max_dist_thr;
              const unsigned long long b = __ballot(keep);
              if (b) { rhi = k0 - __builtin_ctzll(b); break; }
            }
          }
        }
        if (lane < HDR_INTS) {
          const int flags = 1 | (has_i ? 2 : 0) | (has_d ? 4 : 0);
          const int val = lane == 0 ? rlo : lane == 1 ? rhi : lane == 2 ? lo : lane == 3 ? flags : lane == 4 ? (int)off : lane == 5 ? w : 0;
          ring_out[lane] = val; hdr_out[lane] = val;
        }
      }
      __syncthreads();                                          // ring entry visible to every lane
      if (reached) break;
      score++;
      if (score > P.max_score) { status = ST_MAXSCORE; break; }
    }
    // ---------------- backtrace + projection on the reference's columns (src/align.c:366-390) ----------------
    if (status == ST_OK) {
      auto hdr_of = [&](int s, Hdr &h) {
        const int *r = H - (size_t)(s + 1) * HDR_INTS;
        h.lo = r[0]; h.hi = r[1]; h.lo_base = r[2]; h.flags = r[3]; h.off = (uint32_t)r[4]; h.w = r[5];
      };
      int s = score, k = alignment_k, type = 0;               // type: 0 = M, 1 = I, 2 = D
      Hdr hf; hdr_of(s, hf);
      int offset = fetch(A, hf, 0, k);
      int v = offset - k, h = offset;
      bool broken = false;
      while (v > 0 && h > 0 && s > 0) {
        const int s_oe = s - P.oe, s_e = s - P.e, s_x = s - P.x;
        // lane 0: deletion extend, 1: deletion open, 2: insertion extend, 3: insertion open, 4: mismatch
        int val = WFA_NULL;
        if (lane < 5) {
          const int src = (lane == 0 || lane == 2) ? s_e : (lane == 4 ? s_x : s_oe);
          const int which = lane == 0 ? 2 : (lane == 2 ? 1 : 0);
          const int kk = lane < 2 ? k + 1 : (lane < 4 ? k - 1 : k);
          const bool allowed = lane < 2 ? type != 1 : (lane < 4 ? type != 2 : type == 0);
          if (allowed && src >= 0) {
            Hdr hh; hdr_of(src, hh);
            if (((hh.flags >> which) & 1) && kk >= hh.lo && kk <= hh.hi) val = fetch(A, hh, which, kk) + (lane >= 2 ? 1 : 0);
          }
        }
        const int del_ext = __shfl(val, 0), del_open = __shfl(val, 1), ins_ext = __shfl(val, 2), ins_open = __shfl(val, 3), misms = __shfl(val, 4);
        const int max_all = max(misms, max(max(ins_ext, ins_open), max(del_ext, del_open)));
        if (type == 0) {
          const int nm = offset - max_all;
          if (nm < 0 || (nm > 0 && (max_all < 0 || max_all - k < 0 || v > plen + 1 || h > tlen + 1))) { broken = true; break; }   // (a consistent backtrace never gets here)
          bool bad = false;
          for (int j0 = 0; j0 < nm; j0 += 64) {
            const int j = j0 + lane;
            if (j < nm) { const uint8_t c = text[h - 1 - j]; if (c != ref[v - 1 - j]) bad = true; row[v - 1 - j] = c; }
          }
          if (__any(bad)) { broken = true; break; }
          offset = max_all;
          v = offset - k; h = offset;
        }
        if (max_all == del_ext)       { if (lane == 0 && v > 0 && v <= plen) row[v - 1] = '-'; s = s_e;  k++; type = 2; }
        else if (max_all == del_open) { if (lane == 0 && v > 0 && v <= plen) row[v - 1] = '-'; s = s_oe; k++; type = 0; }
        else if (max_all == ins_ext)  { s = s_e;  k--; offset--; type = 1; }
        else if (max_all == ins_open) { s = s_oe; k--; offset--; type = 0; }
        else if (max_all == misms)    { if (lane == 0 && v > 0 && h > 0 && v <= plen + 1) row[v - 1] = text[h - 1]; s = s_x; offset--; }
        else { broken = true; break; }
        v = offset - k; h = offset;
      }
      if (broken) status = ST_BACKTRACE;
      else if (s == 0) { for (int j = lane; j < min(v, plen); j += 64) row[j] = text[j]; }      // the last stroke of matches (k = 0 at score 0)
      else { for (int j = lane; j < min(v, plen); j += 64) row[j] = '-'; }                      // leading deletions; leading insertions leave no trace
      if (lane == 0) row[plen] = 0;
    }
    if (lane == 0) { score_out[q] = status == ST_OK ? score : -1; status_out[q] = status; }
    cells_total += cells;
    __syncthreads();
  }
  if (lane == 0 && cells_total) atomicAdd(cells_out, cells_total);
}

thread_local std::string g_align_open_error;

}  // namespace

struct uvaia_aligner {
  int device = 0;
  hipStream_t stream = nullptr;
  int plen = 0;
  WfaParams P{};
  uint8_t *d_ref = nullptr, *d_seqs = nullptr, *d_aln = nullptr;
  long long *d_off = nullptr;
  int *d_score = nullptr, *d_status = nullptr, *d_next = nullptr, *d_todo = nullptr;
  unsigned long long *d_cells = nullptr;
  uint32_t *d_arena = nullptr;
  size_t seqs_cap = 0, n_cap = 0, arena_words = 0, workspace_request = 0;
  int n = 0, max_blocks = 0, passes = 0;
  bool ran = false;
  unsigned long long cells = 0;
  double kernel_ms = 0;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  std::vector<char> h_bytes; std::vector<long long> h_off;
  std::string err;
};

namespace {

int afail(uvaia_aligner *a, int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (a) a->err = buf; else g_align_open_error = buf;
  return code;
}

#define ACHK(a, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return afail((a), e_ == hipErrorOutOfMemory ? UVAIA_ALIGN_ENOMEM : UVAIA_ALIGN_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

int ensure_pool(uvaia_aligner *a, size_t bytes, int n)
{
  if (bytes + 512 > a->seqs_cap) {
    if (a->d_seqs) hipFree(a->d_seqs);
    a->d_seqs = nullptr; a->seqs_cap = 0;
    const size_t cap = std::max<size_t>(bytes + 512, 1u << 20) * 5 / 4;
    ACHK(a, hipMalloc(&a->d_seqs, cap)); a->seqs_cap = cap;
  }
  if ((size_t)n > a->n_cap) {
    hipFree(a->d_off); hipFree(a->d_aln); hipFree(a->d_score); hipFree(a->d_status); hipFree(a->d_todo);
    a->d_off = nullptr; a->d_aln = nullptr; a->d_score = a->d_status = a->d_todo = nullptr; a->n_cap = 0;
    const size_t cap = std::max<size_t>((size_t)n * 5 / 4, 256);
    ACHK(a, hipMalloc(&a->d_off, (cap + 1) * sizeof(long long)));
    ACHK(a, hipMalloc(&a->d_aln, cap * ((size_t)a->plen + 1)));
    ACHK(a, hipMalloc(&a->d_score, cap * sizeof(int)));
    ACHK(a, hipMalloc(&a->d_status, cap * sizeof(int)));
    ACHK(a, hipMalloc(&a->d_todo, cap * sizeof(int)));
    a->n_cap = cap;
  }
  return 0;
}

int ensure_arena(uvaia_aligner *a)
{
  if (a->d_arena) return 0;
  size_t free_b = 0, total_b = 0;
  ACHK(a, hipMemGetInfo(&free_b, &total_b));
  size_t want = a->workspace_request ? a->workspace_request : std::min<size_t>(free_b / 2, (size_t)a->max_blocks * (24u << 20));
  want = std::max<size_t>(want, 64u << 20) / 4096 * 4096;
  ACHK(a, hipMalloc(&a->d_arena, want));
  a->arena_words = want / sizeof(uint32_t);
  return 0;
}

}  // namespace

extern "C" {

void uvaia_align_default_options(uvaia_align_options *opt)
{ // src/align.c:305-309
  if (!opt) return;
  opt->mismatch = 4; opt->gap_opening = 6; opt->gap_extension = 2;
  opt->min_wavefront_length = 128; opt->max_distance_threshold = 512;
  opt->workspace_bytes = 0; opt->max_blocks = 0;
}

const char *uvaia_align_last_error(const uvaia_aligner *a) { return a ? a->err.c_str() : g_align_open_error.c_str(); }

void uvaia_align_close(uvaia_aligner *a)
{
  if (!a) return;
  hipSetDevice(a->device);
  if (a->stream) hipStreamSynchronize(a->stream);
  hipFree(a->d_ref); hipFree(a->d_seqs); hipFree(a->d_aln); hipFree(a->d_off); hipFree(a->d_score); hipFree(a->d_status); hipFree(a->d_next);
  hipFree(a->d_todo); hipFree(a->d_cells); hipFree(a->d_arena);
  if (a->ev_a) hipEventDestroy(a->ev_a);
  if (a->ev_b) hipEventDestroy(a->ev_b);
  if (a->stream) hipStreamDestroy(a->stream);
  delete a;
}

int uvaia_align_open(uvaia_aligner **out, const char *ref, int ref_len, int device, const uvaia_align_options *opt_in)
{
  if (!out) return UVAIA_ALIGN_EINVAL;
  *out = nullptr;
  if (!ref || ref_len < 1) return afail(nullptr, UVAIA_ALIGN_EINVAL, "empty reference sequence");
  uvaia_align_options opt; uvaia_align_default_options(&opt);
  if (opt_in) opt = *opt_in;
  if (opt.mismatch < 1 || opt.gap_opening < 0 || opt.gap_extension < 1 || opt.mismatch >= RING || opt.gap_opening + opt.gap_extension >= RING)
    return afail(nullptr, UVAIA_ALIGN_EINVAL, "penalties must be positive and below %d (mismatch %d, gap opening %d, gap extension %d)", RING, opt.mismatch, opt.gap_opening, opt.gap_extension);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return afail(nullptr, UVAIA_ALIGN_ENODEV, "no HIP device: the aligner runs on an MI355X (gfx950) and has no CPU path");
  if (device < 0 || device >= ndev) return afail(nullptr, UVAIA_ALIGN_EINVAL, "device %d out of range (%d devices)", device, ndev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return afail(nullptr, UVAIA_ALIGN_ENODEV, "cannot query device %d", device);
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return afail(nullptr, UVAIA_ALIGN_ENODEV, "device %d is %s: this library is built for gfx950 only", device, prop.gcnArchName);
  if (hipSetDevice(device) != hipSuccess) return afail(nullptr, UVAIA_ALIGN_ENODEV, "cannot select device %d", device);
  uvaia_aligner *a = new uvaia_aligner();
  a->device = device; a->plen = ref_len;
  a->P.x = opt.mismatch; a->P.oe = opt.gap_opening + opt.gap_extension; a->P.e = opt.gap_extension;
  a->P.min_wf_len = opt.min_wavefront_length; a->P.max_dist_thr = opt.max_distance_threshold;
  // table size of affine_wavefronts_new_reduced (L, 3 L, ..): min(L, 3L) * mismatch + gap_opening + |L - 3L| * gap_extension
  const long long ms = (long long)ref_len * opt.mismatch + opt.gap_opening + 2LL * ref_len * opt.gap_extension;
  a->P.max_score = (int)std::min<long long>(ms, 0x3fffffff);
  a->workspace_request = opt.workspace_bytes;
  a->max_blocks = opt.max_blocks > 0 ? opt.max_blocks : prop.multiProcessorCount * 32;      // eight waves per SIMD
#define OCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int rc_ = afail(nullptr, e_ == hipErrorOutOfMemory ? UVAIA_ALIGN_ENOMEM : UVAIA_ALIGN_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); uvaia_align_close(a); return rc_; } } while (0)
  OCHK(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
  OCHK(hipEventCreate(&a->ev_a)); OCHK(hipEventCreate(&a->ev_b));
  OCHK(hipMalloc(&a->d_ref, (size_t)ref_len + 512));
  OCHK(hipMemcpy(a->d_ref, ref, (size_t)ref_len, hipMemcpyHostToDevice));
  OCHK(hipMalloc(&a->d_next, sizeof(int)));
  OCHK(hipMalloc(&a->d_cells, sizeof(unsigned long long)));
#undef OCHK
  *out = a;
  return 0;
}

int uvaia_align_load_block(uvaia_aligner *a, const char *bytes, const int64_t *offsets, int n)
{
  if (!a) return UVAIA_ALIGN_EINVAL;
  if (n < 0 || (n > 0 && (!bytes || !offsets))) return afail(a, UVAIA_ALIGN_EINVAL, "bad pool");
  ACHK(a, hipSetDevice(a->device));
  a->n = 0; a->ran = false;
  if (n == 0) return 0;
  for (int i = 0; i < n; i++) if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0x3fffffff) return afail(a, UVAIA_ALIGN_EINVAL, "bad length of sequence %d", i);
  const size_t total = (size_t)(offsets[n] - offsets[0]);
  int rc = ensure_pool(a, total, n); if (rc) return rc;
  a->h_off.resize((size_t)n + 1);
  for (int i = 0; i <= n; i++) a->h_off[(size_t)i] = (long long)(offsets[i] - offsets[0]);
  ACHK(a, hipStreamSynchronize(a->stream));
  if (total) ACHK(a, hipMemcpyAsync(a->d_seqs, bytes + offsets[0], total, hipMemcpyHostToDevice, a->stream));
  ACHK(a, hipMemcpyAsync(a->d_off, a->h_off.data(), ((size_t)n + 1) * sizeof(long long), hipMemcpyHostToDevice, a->stream));
  ACHK(a, hipStreamSynchronize(a->stream));
  a->n = n;
  return 0;
}

int uvaia_align_load(uvaia_aligner *a, const char *const *seq, const int *seq_len, int n)
{
  if (!a) return UVAIA_ALIGN_EINVAL;
  if (n < 0 || (n > 0 && (!seq || !seq_len))) return afail(a, UVAIA_ALIGN_EINVAL, "bad pool");
  std::vector<int64_t> off((size_t)n + 1, 0);
  for (int i = 0; i < n; i++) { if (seq_len[i] < 0 || !seq[i]) return afail(a, UVAIA_ALIGN_EINVAL, "bad sequence %d", i); off[(size_t)i + 1] = off[(size_t)i] + seq_len[i]; }
  a->h_bytes.resize((size_t)off[(size_t)n] + 1);
  for (int i = 0; i < n; i++) memcpy(a->h_bytes.data() + off[(size_t)i], seq[i], (size_t)seq_len[i]);
  return uvaia_align_load_block(a, a->h_bytes.data(), off.data(), n);
}

int uvaia_align_run(uvaia_aligner *a)
{
  if (!a) return UVAIA_ALIGN_EINVAL;
  ACHK(a, hipSetDevice(a->device));
  a->passes = 0; a->cells = 0; a->kernel_ms = 0; a->ran = false;
  if (a->n == 0) { a->ran = true; return 0; }
  int rc = ensure_arena(a); if (rc) return rc;
  ACHK(a, hipMemsetAsync(a->d_cells, 0, sizeof(unsigned long long), a->stream));
  ACHK(a, hipEventRecord(a->ev_a, a->stream));
  int n_todo = a->n, blocks = std::min(a->n, a->max_blocks);
  const int *todo = nullptr;
  std::vector<int> status((size_t)a->n), list;
  for (;;) {
    const unsigned long long share = (unsigned long long)(a->arena_words / (size_t)blocks) / 16 * 16;
    if (share < 1024) return afail(a, UVAIA_ALIGN_ENOMEM, "workspace of %zu bytes is too small for %d queries in flight", a->arena_words * 4, blocks);
    ACHK(a, hipMemsetAsync(a->d_next, 0, sizeof(int), a->stream));
    hipLaunchKernelGGL(wfa_align_kernel, dim3((unsigned)blocks), dim3(64), 0, a->stream, a->d_ref, a->plen, a->d_seqs, a->d_off, todo, n_todo, a->d_aln, (size_t)a->plen + 1,
                       a->d_score, a->d_status, a->d_cells, a->d_arena, share, a->d_next, a->P);
    ACHK(a, hipGetLastError());
    a->passes++;
    ACHK(a, hipMemcpyAsync(status.data(), a->d_status, (size_t)a->n * sizeof(int), hipMemcpyDeviceToHost, a->stream));
    ACHK(a, hipStreamSynchronize(a->stream));
    list.clear();
    for (int i = 0; i < a->n; i++) {
      if (status[(size_t)i] == ST_OVERFLOW) list.push_back(i);
      else if (status[(size_t)i] == ST_MAXSCORE) return afail(a, UVAIA_ALIGN_EINVAL, "sequence %d: alignment score above %d, the size of the reference's score table", i, a->P.max_score);
      else if (status[(size_t)i] == ST_BACKTRACE) return afail(a, UVAIA_ALIGN_ESTATE, "sequence %d: inconsistent backtrace", i);
    }
    if (list.empty()) break;
    if (blocks == 1) return afail(a, UVAIA_ALIGN_ENOMEM, "sequence %d needs more than the whole workspace (%zu bytes) for its wavefronts", list[0], a->arena_words * 4);
    blocks = std::max(1, std::min((int)list.size(), blocks / 8));
    ACHK(a, hipMemcpyAsync(a->d_todo, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice, a->stream));
    todo = a->d_todo; n_todo = (int)list.size();
  }
  ACHK(a, hipEventRecord(a->ev_b, a->stream));
  ACHK(a, hipEventSynchronize(a->ev_b));
  float ms = 0; ACHK(a, hipEventElapsedTime(&ms, a->ev_a, a->ev_b)); a->kernel_ms = ms;
  ACHK(a, hipMemcpy(&a->cells, a->d_cells, sizeof(unsigned long long), hipMemcpyDeviceToHost));
  a->ran = true;
  return 0;
}

int uvaia_align_sync(uvaia_aligner *a)
{
  if (!a) return UVAIA_ALIGN_EINVAL;
  ACHK(a, hipSetDevice(a->device));
  ACHK(a, hipStreamSynchronize(a->stream));
  return 0;
}

int uvaia_align_fetch(uvaia_aligner *a, char *aln, int *score)
{
  if (!a) return UVAIA_ALIGN_EINVAL;
  if (!a->ran) return afail(a, UVAIA_ALIGN_ESTATE, "fetch without a completed run");
  if (a->n == 0) return 0;
  if (!aln) return afail(a, UVAIA_ALIGN_EINVAL, "NULL row buffer");
  ACHK(a, hipSetDevice(a->device));
  ACHK(a, hipMemcpyAsync(aln, a->d_aln, (size_t)a->n * ((size_t)a->plen + 1), hipMemcpyDeviceToHost, a->stream));
  if (score) ACHK(a, hipMemcpyAsync(score, a->d_score, (size_t)a->n * sizeof(int), hipMemcpyDeviceToHost, a->stream));
  ACHK(a, hipStreamSynchronize(a->stream));
  return 0;
}

int uvaia_align_batch(uvaia_aligner *a, const char *const *seq, const int *seq_len, int n, char *aln, int *score)
{
  int rc = uvaia_align_load(a, seq, seq_len, n); if (rc) return rc;
  rc = uvaia_align_run(a); if (rc) return rc;
  return uvaia_align_fetch(a, aln, score);
}

int uvaia_align_stats(uvaia_aligner *a, unsigned long long *cells, double *wavefront_bytes, int *passes, double *kernel_ms)
{
  if (!a) return UVAIA_ALIGN_EINVAL;
  if (cells) *cells = a->cells;
  if (wavefront_bytes) *wavefront_bytes = (double)a->cells * 32.0;     // three offsets written, five read per cell
  if (passes) *passes = a->passes;
  if (kernel_ms) *kernel_ms = a->kernel_ms;
  return 0;
}

}  // extern "C"

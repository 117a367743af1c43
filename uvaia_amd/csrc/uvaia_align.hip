// uvaia_align.hip -- MI355X (gfx950 / CDNA4) gap-affine wavefront aligner behind include/uvaia_align.h.
//
// What it replaces in the reference (paths under /root/reference): the OpenMP loop of src/align.c:224-233 over
// align_query (src/align.c:357-364) and update_query_aligned (src/align.c:366-390), and the per-thread WFA aligners of
// new_queue (src/align.c:286-313).  The WFA library is an absent submodule; the algorithm is the published one
// (Marco-Sola et al. 2021) as oracle/wfa_oracle.h states it, and this file reproduces that statement bit for bit.
//
// Design ("uvaialign"):
//   * one block of four waves per query, persistent: a block takes the next query from an atomic counter until none is left.
//     A query is a chain of thousands of dependent steps (one per score); N-rich sequences (the normal SARS-CoV-2 case) make
//     the wavefronts 1 000-2 000 diagonals wide for most of them, so a step has work for 256 lanes and one barrier.
//   * lane = diagonal.  A step computes I, D, M of 256 diagonals at a time from the wavefronts of score - e, score - o - e,
//     score - x, extends M along the diagonal in registers (eight characters per lane in one unaligned 8-byte compare; longer
//     runs by the whole wave, 1 024 characters per round trip) and stores M once.  Reference and queries are read through L2.
//     A wave takes its diagonals two groups of 64 at a time and asks for the characters of both before it compares either.
//   * the wavefronts a step reads are those of the last few steps: they stay in LDS (16-bit offsets, up to 2 560 diagonals) and
//     the stores to the history in memory are not waited for; wider wavefronts and unusual penalties fall back to memory.
//     The step is compiled twice: the usual one, whose wavefront and sources are all in LDS, carries none of the bookkeeping of
//     the fallbacks (the kernel is bound by instruction issue: what a step executes besides its cells is what it costs).
//   * what is kept for the backtrace is 5 bytes per cell, not 12: the M offset and one provenance byte (which of the five
//     predecessors gave the maximum, in the backtrace's tie order; whether the I and the D cell extend or open).  I and D
//     offsets only feed the next e scores and live in a ring chunk.  A ring of the last 64 headers lives in LDS.
//   * wavefront memory comes in chunks from a pool shared by the blocks in flight (a query takes what its score needs: 2 MB
//     to more than 1 GB); a query that finds the pool empty is flagged and run again in a later, less crowded pass.
//   * the backtrace runs on wave 0 right after the last step and writes the projected row (ref_len characters) directly:
//     match runs are copied 64 characters at a time, runs of mismatches (N runs) are taken 63 steps per round trip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/uvaia_align.h"

// build-time shape (tools/ab_align.sh measures variants): groups of 64 diagonals a wave has in flight, waves per block
#ifndef WFA_GROUP
#define WFA_GROUP 2
#endif
#ifndef WFA_NW
#define WFA_NW 4
#endif

namespace {

constexpr int WFA_NULL = -10;            // offset of a diagonal a wavefront does not hold (oracle/wfa_oracle.c)
constexpr int NW = WFA_NW;               // waves per block = per query
constexpr int TPB = 64 * NW;
constexpr int RING = 64;                 // scores whose headers stay in LDS; penalties are below this
constexpr int HDR_INTS = 8;
constexpr int HDR_PAGE_SCORES = 2048;    // headers of all scores live in pages of the query's own memory (the backtrace reads them)
constexpr int MAX_HDR_PAGES = 256;
constexpr int MAX_OWN = 512;             // chunks a query may hold on top of the block's two permanent ones
constexpr int WL = 2560;                 // widest wavefront kept in LDS (16-bit offsets + 16)
constexpr int RM = 5, RID = 2;           // LDS slots: M wavefronts of the last RM steps, I and D of the last RID
enum { ST_OK = 0, ST_OVERFLOW = 1, ST_MAXSCORE = 2, ST_BACKTRACE = 3, ST_TOOWIDE = 4, ST_HDRPAGES = 5 };
enum { C_DEL_EXT = 0, C_DEL_OPEN = 1, C_INS_EXT = 2, C_INS_OPEN = 3, C_MISMATCH = 4, C_I_EXT = 8, C_D_EXT = 16 };

struct WfaParams { int x, oe, e, min_wf_len, max_dist_thr, max_score, g; };
template <bool B> struct BoolTag { static constexpr bool value = B; };

// Workspace: chunks of 2^chunk_log2 words handed out from a stack under a spin lock (one thread of a block at a time, the others
// wait at a barrier; a query takes a chunk every few hundred thousand cells).  Block b owns chunks 2b (first history chunk) and
// 2b + 1 (ring of the I and D wavefronts) for the whole launch.
struct PoolCtl { int lock, top, n_chunks, chunk_log2; };

// Header of the wavefronts of one score.  M, I and D share their limits (the reduction trims M and hands its limits to I and D).
// flags bit 0 = M exists, bit 1 = I, bit 2 = D.  off16 / id16: where the arrays start, in units of 16 words from the pool's base:
// history = M offsets (w words, padded to 16) followed by one provenance byte per cell; I and D (w16 words each) in the ring chunk.
struct Hdr { int lo, hi, lo_base, flags; uint32_t off16; int w; uint32_t id16; int res; };   // res: 1 + number of the step if its wavefronts went to LDS, else 0

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int w16_of(int w) { return (w + 15) & ~15; }
__device__ __forceinline__ bool in_range(const Hdr &h, int bit, int k) { return ((h.flags >> bit) & 1) && k >= h.lo && k <= h.hi; }
__device__ __forceinline__ int dist_to_end(int plen, int tlen, int offset, int k) { return max(plen - (offset - k), tlen - offset); }

__device__ __forceinline__ int matching_prefix8(const uint8_t *a, const uint8_t *b)
{ // leading bytes (0..8) on which the two strings agree; the compiler picks the load width it may use for unaligned addresses
  unsigned long long x, y;
  __builtin_memcpy(&x, a, 8); __builtin_memcpy(&y, b, 8);
  const unsigned long long d = x ^ y;
  return d ? (int)(__builtin_ctzll(d) >> 3) : 8;
}

// minimum over the wave by DPP (row shifts, then row broadcasts): six VALU steps, no trip through the LDS crossbar
__device__ __forceinline__ int wave_min_dpp(int v)
{
  const int big = 0x7fffffff;
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x111, 0xF, 0xF, false));    // row_shr:1
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x112, 0xF, 0xF, false));    // row_shr:2
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x114, 0xF, 0xF, false));    // row_shr:4
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x118, 0xF, 0xF, false));    // row_shr:8  -> lane 15 of every row holds the row's minimum
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x142, 0xA, 0xF, false));    // row_bcast:15 into rows 1 and 3
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x143, 0xC, 0xF, false));    // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}

// barrier for steps whose wavefronts are exchanged through LDS only: the stores to the history in memory stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void pool_lock(PoolCtl *c) { while (atomicCAS(&c->lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(4); __threadfence(); }
__device__ __forceinline__ void pool_unlock(PoolCtl *c) { __threadfence(); atomicExch(&c->lock, 0); }

__global__ __launch_bounds__(TPB) void wfa_align_kernel(const uint8_t *__restrict__ ref, int plen, const uint8_t *__restrict__ seqs, const long long *__restrict__ seq_off,
                                                         const int *__restrict__ todo, int n_todo, uint8_t *__restrict__ aln, size_t aln_pitch, int *__restrict__ score_out,
                                                         int *__restrict__ status_out, unsigned long long *__restrict__ cells_out, uint32_t *__restrict__ pool,
                                                         PoolCtl *ctl, int *stack, int *next_query, WfaParams P)
{
  __shared__ int ring[RING][HDR_INTS];
  __shared__ int own[MAX_OWN];
  __shared__ uint32_t hdr_page[MAX_HDR_PAGES];
  __shared__ int wsync[2][NW][2];
  __shared__ int dend[2][2][64];           // distance to the end cell of a wavefront's first and last 64 diagonals (for the reduction)
  __shared__ int bc[2];
  // The wavefronts a step reads -- M of score - x and score - o - e, I and D of score - e -- are those of the last few steps: they stay
  // in LDS (offset + 16 in 16 bits: the null offset is 6) as long as they are at most WL wide, and the history in memory is written
  // without being waited for.  A step then costs LDS latency plus one round trip for the characters of the extension.
  __shared__ uint16_t lm[RM][WL], li[RID][WL], ld[RID][WL];
  const int tid = threadIdx.x, lane = tid & 63, wave = rfl(tid >> 6);
  const int chunk_log2 = ctl->chunk_log2;
  const unsigned chunk_words = 1u << chunk_log2;
  const size_t home_base = (size_t)(2 * blockIdx.x) << chunk_log2, ring_base = (size_t)(2 * blockIdx.x + 1) << chunk_log2;
  // the I and D wavefronts a step reads are those of score - e, at most e / gcd steps back: more of them than LDS slots, and every step
  // keeps a copy of its I and D in memory
  const bool id_deep = P.e / P.g >= RID;
  unsigned long long cells_total = 0;

  for (;;) {
    if (tid == 0) bc[0] = atomicAdd(next_query, 1);
    __syncthreads();
    const int qi = rfl(bc[0]);                                // (read back from LDS: the same in every lane, and the compiler should know)
    if (qi >= n_todo) break;                                  // every wave of every block reaches this: the counter only grows
    const int q = todo ? todo[qi] : qi;
    const uint8_t *text = seqs + seq_off[q];
    const int tlen = (int)(seq_off[q + 1] - seq_off[q]);
    const int alignment_k = tlen - plen;
    uint8_t *row = aln + (size_t)q * aln_pitch;

    for (int i = tid; i < RING * HDR_INTS; i += TPB) (&ring[0][0])[i] = 0;
    __syncthreads();

    // bump allocation inside the current chunk; every thread keeps the same state
    size_t cur_base = home_base;
    unsigned cur_used = 0, ring_pos = 0, widest16 = 16;
    int n_own = 0;
    unsigned long long cells = 0;
    int score = 0, status = ST_OK, step = 0, cslot = 0, cislot = 0;      // cslot = step mod RM, cislot = step mod RID
    bool reached = false;

    auto take = [&](unsigned words, size_t &at) -> bool {    // `words` (a multiple of 16, at most a chunk) of the query's memory; false = the pool is empty
      if (cur_used + words > chunk_words) {
        __syncthreads();
        if (tid == 0) {
          int id = -1;
          if (n_own < MAX_OWN) {
            pool_lock(ctl);
            const int top = atomicAdd(&ctl->top, 0);
            if (top > 0) { id = atomicAdd(&stack[top - 1], 0); atomicExch(&ctl->top, top - 1); }
            pool_unlock(ctl);
            if (id >= 0) own[n_own] = id;
          }
          bc[1] = id;
        }
        __syncthreads();
        const int id = rfl(bc[1]);
        if (id < 0) return false;
        n_own++;
        cur_base = (size_t)id << chunk_log2; cur_used = 0;
      }
      at = cur_base + cur_used;
      cur_used += words;
      return true;
    };

    // ---------------- forward: one step per score ----------------
    for (;;) {
      if (score / HDR_PAGE_SCORES != (score - P.g) / HDR_PAGE_SCORES || score == 0) {     // a new page of headers
        const int page = score / HDR_PAGE_SCORES;
        size_t at = 0;
        if (page >= MAX_HDR_PAGES) { status = ST_HDRPAGES; break; }
        if (!take(HDR_PAGE_SCORES * HDR_INTS, at)) { status = ST_OVERFLOW; break; }
        if (tid == 0) hdr_page[page] = (uint32_t)(at >> 4);
        __syncthreads();
      }
      // the headers of the three source scores: the same for every lane, kept in scalar registers (a score below zero has none; where
      // the arrays lie in memory -- off16, w, id16 -- is looked up only by a step that has to read a source from there)
      auto load = [&](int s) -> Hdr {
        Hdr h{};
        if (s >= 0) { const int *r = ring[s & (RING - 1)]; h.lo = rfl(r[0]); h.hi = rfl(r[1]); h.lo_base = rfl(r[2]); h.flags = rfl(r[3]); h.res = rfl(r[7]); }
        return h;
      };
      const Hdr hs = load(score - P.x), hg = load(score - P.oe), he = load(score - P.e);
      const bool n_sub = !(hs.flags & 1), n_gap = !(hg.flags & 1), n_i = !(he.flags & 2), n_d = !(he.flags & 4);
      const bool have = score == 0 || !(n_sub && n_gap && n_i && n_d);
      int lo = 0, hi = 0;
      if (score > 0 && have) {
        lo = min(min(n_sub ? 1 : hs.lo, n_gap ? 1 : hg.lo), min(n_i ? 1 : he.lo, n_d ? 1 : he.lo)) - 1;
        hi = max(max(n_sub ? -1 : hs.hi, n_gap ? -1 : hg.hi), max(n_i ? -1 : he.hi, n_d ? -1 : he.hi)) + 1;
      }
      int *hdr_out = reinterpret_cast<int *>(pool + ((size_t)hdr_page[score / HDR_PAGE_SCORES] << 4)) + (size_t)(score & (HDR_PAGE_SCORES - 1)) * HDR_INTS;
      int *ring_out = ring[score & (RING - 1)];
      if (!have) {
        // every wave writes the (identical) ring entry it is going to read: no barrier for it; the header in memory is for the backtrace
        if (lane < HDR_INTS) { ring_out[lane] = 0; if (wave == 0) hdr_out[lane] = 0; }
      } else {
        const bool has_i = score > 0 && ((hg.flags & 1) || (he.flags & 2)), has_d = score > 0 && ((hg.flags & 1) || (he.flags & 4));
        const int w = hi - lo + 1, w16 = w16_of(w);
        const unsigned hist_words = (unsigned)w16 + (unsigned)w16_of((w + 3) / 4);
        // where the wavefronts of this step and of its sources live
        const bool fits = w <= WL && tlen + step + 32 < 65535;              // offsets grow by at most one per step beyond tlen
        // LDS slots: this step's (cslot, cislot: step mod RM, step mod RID, kept as running counters) and, d steps back, its sources'
        const int d_s = step - (hs.res - 1), d_g = step - (hg.res - 1), d_e = step - (he.res - 1);
        const bool lds_s = hs.res && d_s < RM, lds_g = hg.res && d_g < RM, lds_e = he.res && d_e < RID;
        const int slot_s = lds_s ? (cslot >= d_s ? cslot - d_s : cslot - d_s + RM) : 0, slot_g = lds_g ? (cslot >= d_g ? cslot - d_g : cslot - d_g + RM) : 0;
        const int slot_e = lds_e ? (cislot >= d_e ? cislot - d_e : cislot - d_e + RID) : 0;
        const bool sources_in_lds = (!(hs.flags & 1) || lds_s) && (!(hg.flags & 1) || lds_g) && (!(he.flags & 6) || lds_e);
        // The step proper, compiled twice.  LDS_ONLY: the usual step -- this wavefront fits in LDS, so do all it reads, all five source
        // wavefronts exist, and I and D need no copy in memory: nothing but the history of the backtrace leaves the CU, and none of the
        // bookkeeping of the other kind of step (positions in the ring chunk, pointers to sources in memory, which of them to wait for,
        // which wavefronts there are at all) is executed.  Returns true when the forward pass has to stop (status set).
        const bool all_sources = score > 0 && (hs.flags & 1) && (hg.flags & 1) && (he.flags & 6) == 6;
        auto step_body = [&](auto lds_only_tag) -> bool {
          constexpr bool LDS_ONLY = decltype(lds_only_tag)::value;
          const bool resident = LDS_ONLY || fits, some = LDS_ONLY || score > 0, with_i = LDS_ONLY || has_i, with_d = LDS_ONLY || has_d;
          auto inside = [&](const Hdr &h, int bit, int k) -> bool { return LDS_ONLY ? (k >= h.lo && k <= h.hi) : in_range(h, bit, k); };
          size_t id_at = 0;
          if (!LDS_ONLY) {
            // the ring chunk keeps the I and D wavefronts of the last e + 1 steps: e + 2 pairs of the widest one must fit (one is lost to the wrap)
            widest16 = max(widest16, (unsigned)w16);
            if ((unsigned long long)(P.e + 2) * 2ull * widest16 > chunk_words || hist_words > chunk_words) { status = ST_TOOWIDE; return true; }
            if (ring_pos + 2u * (unsigned)w16 > chunk_words) ring_pos = 0;
            id_at = ring_base + ring_pos;
            ring_pos += 2u * (unsigned)w16;
          }
          size_t m_at = 0;
          if (!take(hist_words, m_at)) { status = ST_OVERFLOW; return true; }
          cells += (unsigned)w;
          const bool id_to_memory = !LDS_ONLY && (!resident || id_deep);      // (more existing scores between s - e and s than LDS slots: keep a copy)
          const bool all_lds = LDS_ONLY || sources_in_lds;
          // a source that left LDS is read from memory, where its step wrote it without waiting: make those stores complete first
          if (!LDS_ONLY && (((hs.flags & 1) && hs.res && !lds_s) || ((hg.flags & 1) && hg.res && !lds_g) || ((he.flags & 6) && he.res && !lds_e))) __syncthreads();
          uint32_t *out_m = pool + m_at, *out_i = pool + id_at, *out_d = out_i + w16;
          uint8_t *out_c = reinterpret_cast<uint8_t *>(out_m + w16);
          // sources in memory (a step that is not all_lds): M of score - x, M of score - o - e, I and D of score - e
          auto mem_m = [&](int s_, const Hdr &h_) -> const uint32_t * { return pool + ((size_t)(uint32_t)ring[s_ & (RING - 1)][4] << 4) - h_.lo_base; };
          auto mem_i = [&](int s_, const Hdr &h_) -> const uint32_t * { return pool + ((size_t)(uint32_t)ring[s_ & (RING - 1)][6] << 4) - h_.lo_base; };
          int min_distance = max(plen, tlen);
          bool hit_end = false;
          // A cell in two halves.  front: its offset before the extension (five offsets from LDS, I and D stored) and the request for the
          // first eight characters of either sequence; back: the extension and the stores.  A wave takes its diagonals WFA_GROUP x 64 at
          // a time, all fronts before the first back: one LDS and one memory round trip per group instead of one per 64 diagonals.
          struct Cell { int k, m; unsigned code; bool act, go; unsigned long long x, y; };
          auto front = [&](int k0) -> Cell {
            Cell c;
            const int k = k0 + lane;
            const bool act = k <= hi;
            int m = 0;
            unsigned code = C_MISMATCH;
            if (some) {
              const bool in_s = inside(hs, 0, k), in_gm = inside(hg, 0, k - 1), in_gp = inside(hg, 0, k + 1), in_i = inside(he, 1, k - 1), in_d = inside(he, 2, k + 1);
              int r_s, r_gm, r_gp, r_i, r_d;
              if (all_lds) {     // the usual case: five unconditional LDS reads at clamped positions, the range tests as selects (no branches)
                const int xs = min(max(k - hs.lo_base, 0), WL - 1), xm = min(max(k - 1 - hg.lo_base, 0), WL - 1), xp = min(max(k + 1 - hg.lo_base, 0), WL - 1);
                const int xi = min(max(k - 1 - he.lo_base, 0), WL - 1), xd = min(max(k + 1 - he.lo_base, 0), WL - 1);
                const int a_s = (int)lm[slot_s][xs] - 16, a_m = (int)lm[slot_g][xm] - 16, a_p = (int)lm[slot_g][xp] - 16, a_i = (int)li[slot_e][xi] - 16, a_d = (int)ld[slot_e][xd] - 16;
                r_s = in_s ? a_s : WFA_NULL; r_gm = in_gm ? a_m : WFA_NULL; r_gp = in_gp ? a_p : WFA_NULL; r_i = in_i ? a_i : WFA_NULL; r_d = in_d ? a_d : WFA_NULL;
              } else {
                const uint32_t *ms = (hs.flags & 1) && !lds_s ? mem_m(score - P.x, hs) : pool, *mg = (hg.flags & 1) && !lds_g ? mem_m(score - P.oe, hg) : pool;
                const uint32_t *ie = (he.flags & 6) && !lds_e ? mem_i(score - P.e, he) : pool, *de = ie + ((he.flags & 6) && !lds_e ? w16_of(ring[(score - P.e) & (RING - 1)][5]) : 0);
                r_s  = in_s  ? (lds_s ? (int)lm[slot_s][k - hs.lo_base] - 16     : (int)ms[k])     : WFA_NULL;
                r_gm = in_gm ? (lds_g ? (int)lm[slot_g][k - 1 - hg.lo_base] - 16 : (int)mg[k - 1]) : WFA_NULL;
                r_gp = in_gp ? (lds_g ? (int)lm[slot_g][k + 1 - hg.lo_base] - 16 : (int)mg[k + 1]) : WFA_NULL;
                r_i  = in_i  ? (lds_e ? (int)li[slot_e][k - 1 - he.lo_base] - 16 : (int)ie[k - 1]) : WFA_NULL;
                r_d  = in_d  ? (lds_e ? (int)ld[slot_e][k + 1 - he.lo_base] - 16 : (int)de[k + 1]) : WFA_NULL;
              }
              // the five predecessors as the backtrace sees them (a "+ 1" belongs to a fetched value only)
              const int v_sub = in_s ? r_s + 1 : WFA_NULL, v_io = in_gm ? r_gm + 1 : WFA_NULL, v_ie = in_i ? r_i + 1 : WFA_NULL, v_do = r_gp, v_de = r_d;
              m = v_sub;
              if (with_i) { const int ins = max(r_gm, r_i) + 1; if (act) { if (resident) li[cislot][k - lo] = (uint16_t)(ins + 16); if (id_to_memory) out_i[k - lo] = (uint32_t)ins; } m = max(m, ins); }
              if (with_d) { const int del = max(r_gp, r_d);     if (act) { if (resident) ld[cislot][k - lo] = (uint16_t)(del + 16); if (id_to_memory) out_d[k - lo] = (uint32_t)del; } m = max(m, del); }
              const int bt = max(v_sub, max(max(v_io, v_ie), max(v_do, v_de)));
              code = bt == v_de ? C_DEL_EXT : bt == v_do ? C_DEL_OPEN : bt == v_ie ? C_INS_EXT : bt == v_io ? C_INS_OPEN : C_MISMATCH;   // the backtrace's tie order
              code |= (v_ie >= v_io ? C_I_EXT : 0) | (v_de >= v_do ? C_D_EXT : 0);
            }
            // exact extension along the diagonal (paper algorithm 2): eight characters per lane in one round trip (the loads may run up to
            // seven bytes past a sequence: both buffers are padded) ...
            const int v = m - k, h = m;
            c.k = k; c.m = m; c.code = code; c.act = act;
            c.go = act && (unsigned)h < (unsigned)tlen && (unsigned)v < (unsigned)plen;
            c.x = 0ull; c.y = 0ull;
            if (c.go) { __builtin_memcpy(&c.x, ref + v, 8); __builtin_memcpy(&c.y, text + h, 8); }
            return c;
          };
          auto back = [&](const Cell &c, int k0) {
            const int k = c.k;
            int m = c.m, v = m - k, h = m;
            bool go = c.go;
            {   // (no branch around this: a wait for the characters that a path can skip makes the next cells wait for this cell's stores)
              const unsigned long long d = c.x ^ c.y;
              const int nmat = go ? min(d ? (int)(__builtin_ctzll(d) >> 3) : 8, min(plen - v, tlen - h)) : 0;
              v += nmat; h += nmat; m += nmat;
              go = go && nmat == 8 && v < plen && h < tlen;
            }
            // ... longer runs by the whole wave, 1 024 characters per round trip
            unsigned long long more = __ballot(go);
            while (more) {
              const int j = __builtin_ctzll(more);
              more &= more - 1;
              const int vj = __shfl(v, j), hj = __shfl(h, j);
              int ext = 0;
              for (;;) {
                unsigned long long part[2];
                int nm_[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                  const int pv = vj + ext + (u * 64 + lane) * 8, ph = hj + ext + (u * 64 + lane) * 8;
                  nm_[u] = (pv < plen && ph < tlen) ? min(matching_prefix8(ref + pv, text + ph), min(plen - pv, tlen - ph)) : 0;
                  part[u] = __ballot(nm_[u] < 8);
                }
                if (part[0]) { const int l = __builtin_ctzll(part[0]); ext += 8 * l + __shfl(nm_[0], l); break; }
                if (part[1]) { const int l = __builtin_ctzll(part[1]); ext += 512 + 8 * l + __shfl(nm_[1], l); break; }
                ext += 1024;
              }
              if (lane == j) m += ext;
            }
            const int dk_ = c.act ? dist_to_end(plen, tlen, m, k) : 0x7fffffff;
            if (k0 == lo) dend[step & 1][0][lane] = dk_;
            if (k0 + 64 > hi) dend[step & 1][1][lane] = dk_;
            if (c.act) {
              if (resident) lm[cslot][k - lo] = (uint16_t)(m + 16);
              out_m[k - lo] = (uint32_t)m;
              out_c[k - lo] = (uint8_t)c.code;
              min_distance = min(min_distance, dk_);
              if (k == alignment_k && m >= tlen) hit_end = true;
            }
          };
          for (int k0 = lo + 64 * wave; k0 <= hi; k0 += WFA_GROUP * TPB) {
            Cell c[WFA_GROUP];
#pragma unroll
            for (int g = 0; g < WFA_GROUP; g++) if (k0 + g * TPB <= hi) c[g] = front(k0 + g * TPB);
            __builtin_amdgcn_sched_barrier(0);                   // (the scheduler would pair every front with its back again)
#pragma unroll
            for (int g = 0; g < WFA_GROUP; g++) if (k0 + g * TPB <= hi) back(c[g], k0 + g * TPB);
          }
          min_distance = wave_min_dpp(min_distance);
          const bool wave_hit = __any(hit_end);
          if (lane == 0) { wsync[step & 1][wave][0] = min_distance; wsync[step & 1][wave][1] = wave_hit ? 1 : 0; }
          if (resident) lds_barrier(); else __syncthreads();      // the one barrier of a step: offsets stored by other waves are read from here on
          min_distance = wsync[step & 1][0][0]; reached = wsync[step & 1][0][1] != 0;
#pragma unroll
          for (int u = 1; u < NW; u++) { min_distance = min(min_distance, wsync[step & 1][u][0]); reached = reached || wsync[step & 1][u][1] != 0; }
          min_distance = rfl(min_distance); reached = rfl(reached ? 1 : 0) != 0;     // uniform: the loop over the scores is a scalar loop
          const int par = step & 1;
          step++;
          // adaptive reduction (paper section 2.4): trim both ends of a long wavefront, never across the end cell's diagonal.
          // Every wave computes the same limits for itself, as a rule from the two ends' distances in LDS.
          int rlo = lo, rhi = hi;
          if (P.min_wf_len > 0 && w >= P.min_wf_len) {
            const int top_limit = min(alignment_k - 1, hi);
            if (lo < top_limit) {
              rlo = top_limit;
              const unsigned long long b0 = __ballot(lo + lane < top_limit && dend[par][0][lane] - min_distance <= P.max_dist_thr);
              if (b0) rlo = lo + __builtin_ctzll(b0);
              else for (int k0 = lo + 64; k0 < top_limit; k0 += 64) {
                const int k = k0 + lane;
                const bool keep = k < top_limit && dist_to_end(plen, tlen, resident ? (int)lm[cslot][k - lo] - 16 : (int)out_m[k - lo], k) - min_distance <= P.max_dist_thr;
                const unsigned long long b = __ballot(keep);
                if (b) { rlo = k0 + __builtin_ctzll(b); break; }
              }
            }
            const int bottom_limit = max(alignment_k + 1, rlo);
            if (hi > bottom_limit) {
              rhi = bottom_limit;
              const int kh = lo + ((hi - lo) / 64) * 64;                  // first diagonal of the chunk that holds hi
              const unsigned long long b0 = __ballot(kh + lane <= hi && kh + lane > bottom_limit && dend[par][1][lane] - min_distance <= P.max_dist_thr);
              if (b0) rhi = kh + 63 - __builtin_clzll(b0);
              else for (int k0 = kh - 1; k0 > bottom_limit; k0 -= 64) {
                const int k = k0 - lane;
                const bool keep = k > bottom_limit && dist_to_end(plen, tlen, resident ? (int)lm[cslot][k - lo] - 16 : (int)out_m[k - lo], k) - min_distance <= P.max_dist_thr;
                const unsigned long long b = __ballot(keep);
                if (b) { rhi = k0 - __builtin_ctzll(b); break; }
              }
            }
          }
          if (lane < HDR_INTS) {
            const int flags = 1 | (has_i ? 2 : 0) | (has_d ? 4 : 0);
            const int val = lane == 0 ? rlo : lane == 1 ? rhi : lane == 2 ? lo : lane == 3 ? flags : lane == 4 ? (int)(uint32_t)(m_at >> 4) : lane == 5 ? w : lane == 6 ? (int)(uint32_t)(id_at >> 4) : (resident ? step : 0);
            ring_out[lane] = val;
            if (wave == 0) hdr_out[lane] = val;
          }
          cslot = cslot + 1 == RM ? 0 : cslot + 1; cislot = cislot + 1 == RID ? 0 : cislot + 1;
          return false;
        };
        if ((fits && all_sources && sources_in_lds && !id_deep) ? step_body(BoolTag<true>()) : step_body(BoolTag<false>())) break;
      }
      if (reached) break;
      score += P.g;                                             // scores that are no multiple of gcd(x, o + e, e) have no wavefront (and no header: never looked up)
      if (score > P.max_score) { status = ST_MAXSCORE; break; }
    }
    __syncthreads();                                            // headers in memory, last offsets: visible to the backtrace
    // ---------------- backtrace + projection on the reference's columns (src/align.c:366-390): wave 0 ----------------
    if (status == ST_OK && wave == 0) {
      auto hdr_of = [&](int s, Hdr &h) {
        const int *r = reinterpret_cast<const int *>(pool + ((size_t)hdr_page[s / HDR_PAGE_SCORES] << 4)) + (size_t)(s & (HDR_PAGE_SCORES - 1)) * HDR_INTS;
        h.lo = r[0]; h.hi = r[1]; h.lo_base = r[2]; h.flags = r[3]; h.off16 = (uint32_t)r[4]; h.w = r[5];
      };
      auto m_of = [&](const Hdr &h, int k) { return (int)pool[((size_t)h.off16 << 4) + (size_t)(k - h.lo_base)]; };
      auto code_of = [&](const Hdr &h, int k) { return (int)reinterpret_cast<const uint8_t *>(pool + ((size_t)h.off16 << 4) + w16_of(h.w))[k - h.lo_base]; };
      auto cell_ok = [&](const Hdr &h, int k) { return (h.flags & 1) && k >= h.lo_base && k < h.lo_base + h.w; };
      int s = score, k = alignment_k, state = 0;              // state: 0 = M, 1 = I, 2 = D
      Hdr hc; hdr_of(s, hc);
      int offset = in_range(hc, 0, k) ? m_of(hc, k) : WFA_NULL;
      int v = offset - k, h = offset;
      bool broken = false;
      while (v > 0 && h > 0 && s > 0) {
        if (state == 0) {
          // A run of mismatches on one diagonal with no matches between them (an N run of the query): lane j looks at the cell
          // j mismatches back; the leading lanes whose cell came from a mismatch and was not extended are taken in one go.
          const int sj = s - lane * P.x;
          int cj = -1, mj = WFA_NULL;
          if (sj >= 0) { Hdr hj; hdr_of(sj, hj); if (in_range(hj, 0, k)) { mj = m_of(hj, k); cj = code_of(hj, k) & 7; } }
          const int mnext = __shfl_down(mj, 1);
          const bool pure = lane < 63 && sj > 0 && cj == C_MISMATCH && mnext != WFA_NULL && mj == mnext + 1 && v - lane > 0 && h - lane > 0;
          const unsigned long long np = ~__ballot(pure);
          const int n = np ? __builtin_ctzll(np) : 63;
          if (n > 0) {
            if (lane < n && v - 1 - lane <= plen) row[v - 1 - lane] = text[h - 1 - lane];
            s -= n * P.x; offset -= n; v -= n; h -= n;
            continue;
          }
          const int c = __shfl(cj, 0);
          if (c < 0) { broken = true; break; }
          int max_all = WFA_NULL;
          if (c == C_MISMATCH) { const int mn = __shfl(mj, 1); if (mn == WFA_NULL) { broken = true; break; } max_all = mn + 1; }
          else if (c == C_INS_OPEN || c == C_DEL_OPEN) {
            const int ss = s - P.oe, kk = c == C_INS_OPEN ? k - 1 : k + 1;
            Hdr ho; if (ss < 0) { broken = true; break; } hdr_of(ss, ho);
            if (!in_range(ho, 0, kk)) { broken = true; break; }
            max_all = m_of(ho, kk) + (c == C_INS_OPEN ? 1 : 0);
          } else {                                               // a gap being extended: follow it to the M cell it was opened from
            const int dk = c == C_INS_EXT ? -1 : 1, bit = c == C_INS_EXT ? C_I_EXT : C_D_EXT;
            int ss = s - P.e, kk = k + dk, cnt = 0;
            for (;;) {
              Hdr hw; if (ss < 0) { broken = true; break; } hdr_of(ss, hw);
              if (!cell_ok(hw, kk)) { broken = true; break; }
              cnt++;
              if (!(code_of(hw, kk) & bit)) {
                Hdr ho; if (ss - P.oe < 0) { broken = true; break; } hdr_of(ss - P.oe, ho);
                if (!in_range(ho, 0, kk + dk)) { broken = true; break; }
                max_all = m_of(ho, kk + dk) + (c == C_INS_EXT ? cnt + 1 : 0);
                break;
              }
              ss -= P.e; kk += dk;
            }
            if (broken) break;
          }
          const int nm = offset - max_all;
          if (nm < 0 || (nm > 0 && (max_all < 0 || max_all - k < 0 || v > plen + 1 || h > tlen + 1))) { broken = true; break; }   // (a consistent backtrace never gets here)
          bool bad = false;
          for (int j0 = 0; j0 < nm; j0 += 64) {
            const int j = j0 + lane;
            if (j < nm) { const uint8_t ch = text[h - 1 - j]; if (ch != ref[v - 1 - j]) bad = true; row[v - 1 - j] = ch; }
          }
          if (__any(bad)) { broken = true; break; }
          offset = max_all;
          v = offset - k; h = offset;
          if (c == C_DEL_EXT)        { if (lane == 0 && v > 0 && v <= plen) row[v - 1] = '-'; s -= P.e;  k++; state = 2; }
          else if (c == C_DEL_OPEN)  { if (lane == 0 && v > 0 && v <= plen) row[v - 1] = '-'; s -= P.oe; k++; state = 0; }
          else if (c == C_INS_EXT)   { s -= P.e;  k--; offset--; state = 1; }
          else if (c == C_INS_OPEN)  { s -= P.oe; k--; offset--; state = 0; }
          else                       { if (lane == 0 && v > 0 && h > 0 && v <= plen + 1) row[v - 1] = text[h - 1]; s -= P.x; offset--; }
        } else {
          Hdr hw; hdr_of(s, hw);
          if (!cell_ok(hw, k)) { broken = true; break; }
          const int cb = code_of(hw, k);
          if (state == 1) { const bool ext = cb & C_I_EXT; s -= ext ? P.e : P.oe; k--; offset--; state = ext ? 1 : 0; }
          else { const bool ext = cb & C_D_EXT; if (lane == 0 && v > 0 && v <= plen) row[v - 1] = '-'; s -= ext ? P.e : P.oe; k++; state = ext ? 2 : 0; }
        }
        v = offset - k; h = offset;
      }
      if (broken || s < 0) status = ST_BACKTRACE;
      else if (s == 0) { for (int j = lane; j < min(v, plen); j += 64) row[j] = text[j]; }      // the last stroke of matches (k = 0 at score 0)
      else { for (int j = lane; j < min(v, plen); j += 64) row[j] = '-'; }                      // leading deletions; leading insertions leave no trace
      if (lane == 0) row[plen] = 0;
    }
    if (tid == 0) {
      score_out[q] = status == ST_OK ? score : -1; status_out[q] = status;     // (thread 0 belongs to the wave that ran the backtrace)
      if (n_own > 0) {                                           // the chunks taken for this query go back
        pool_lock(ctl);
        int top = atomicAdd(&ctl->top, 0);
        for (int i = 0; i < n_own; i++) atomicExch(&stack[top++], own[i]);
        atomicExch(&ctl->top, top);
        pool_unlock(ctl);
      }
    }
    if (status == ST_OK) cells_total += cells;                   // (a query that is run again is counted by the pass that completes it)
    __syncthreads();
  }
  if (tid == 0 && cells_total) atomicAdd(cells_out, cells_total);
}

// What a query is going to cost, roughly: every site that is no A, C, G or T is a mismatch against the reference (4 each, and N runs
// are what makes scores of tens of thousands), every character of length difference a gap extension.  The blocks are persistent and
// take the next query when they finish one, so the last queries to start decide how long the launch's tail is: the dearer half of
// the pool is started first, in the order it came (all of the dearest at once would also want all of the workspace at once), then
// the cheaper half in descending order of the estimate; a small pool is in descending order throughout.  One block per query.
__global__ __launch_bounds__(256) void expected_cost_kernel(const uint8_t *__restrict__ seqs, const long long *__restrict__ seq_off, int plen, int n, int *__restrict__ cost)
{
  __shared__ int part[4];
  const int q = blockIdx.x;
  if (q >= n) return;
  const uint8_t *t = seqs + seq_off[q];
  const int tlen = (int)(seq_off[q + 1] - seq_off[q]);
  int c = 0;
  for (int i = threadIdx.x; i < tlen; i += 256) { const uint8_t ch = t[i] & 0xDF; c += !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) cost[q] = 4 * (part[0] + part[1] + part[2] + part[3]) + 2 * abs(tlen - plen);
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
  int *d_score = nullptr, *d_status = nullptr, *d_next = nullptr, *d_todo = nullptr, *d_order = nullptr;   // d_order: the pool's queries, the dearest first
  unsigned long long *d_cells = nullptr;
  uint32_t *d_pool = nullptr;               // wavefront memory: n_chunks chunks of 2^chunk_log2 words
  PoolCtl *d_ctl = nullptr; int *d_stack = nullptr;
  int n_chunks = 0, chunk_log2 = 0;
  std::vector<int> h_stack;
  size_t seqs_cap = 0, n_cap = 0, workspace_request = 0;
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
    hipFree(a->d_off); hipFree(a->d_aln); hipFree(a->d_score); hipFree(a->d_status); hipFree(a->d_todo); hipFree(a->d_order);
    a->d_off = nullptr; a->d_aln = nullptr; a->d_score = a->d_status = a->d_todo = a->d_order = nullptr; a->n_cap = 0;
    const size_t cap = std::max<size_t>((size_t)n * 5 / 4, 256);
    ACHK(a, hipMalloc(&a->d_off, (cap + 1) * sizeof(long long)));
    ACHK(a, hipMalloc(&a->d_aln, cap * ((size_t)a->plen + 1)));
    ACHK(a, hipMalloc(&a->d_score, cap * sizeof(int)));
    ACHK(a, hipMalloc(&a->d_status, cap * sizeof(int)));
    ACHK(a, hipMalloc(&a->d_todo, cap * sizeof(int)));
    ACHK(a, hipMalloc(&a->d_order, cap * sizeof(int)));
    a->n_cap = cap;
  }
  return 0;
}

// The workspace: with workspace_bytes == 0 it starts at 4 GB (a small job should not pay for mapping 200 GB) and grows by a factor
// of four, up to three quarters of the free device memory, whenever a pass leaves queries that found the pool empty.
int ensure_pool_memory(uvaia_aligner *a, bool grow, bool *grew)
{
  if (grew) *grew = false;
  if (a->d_pool && !grow) return 0;
  // a chunk holds the ring of I and D wavefronts (e + 2 pairs of the widest wavefront, at most all plen + tlen + 1 diagonals of a
  // query of up to 1.5 reference lengths, src/align.c:199) and serves as the unit the queries' histories grow by
  const unsigned long long widest = (unsigned long long)a->plen * 5 / 2 + 64;
  int lg = 19;
  while ((1ull << lg) < (unsigned long long)(a->P.e + 2) * 2ull * widest) lg++;
  if (lg > 28) return afail(a, UVAIA_ALIGN_EINVAL, "reference of %d sites is too long for the wavefront workspace", a->plen);
  const size_t chunk_bytes = (size_t)4 << lg, have = (size_t)a->n_chunks * chunk_bytes;
  if (a->d_pool && a->workspace_request) return 0;                  // a workspace of a given size does not grow
  size_t free_b = 0, total_b = 0;
  ACHK(a, hipMemGetInfo(&free_b, &total_b));
  const size_t cap = std::min((free_b + have) / 4 * 3, (size_t)0xffffffffull * 64);   // array positions are kept in units of 64 bytes in 32 bits
  size_t want = a->workspace_request ? std::min(a->workspace_request, (size_t)0xffffffffull * 64) : std::min(cap, a->d_pool ? have * 4 : std::max((size_t)4 << 30, 8 * chunk_bytes));
  if (a->d_pool && want <= have) return 0;                          // already as large as it gets
  size_t n_chunks = want / chunk_bytes;
  if (n_chunks < 3) return afail(a, UVAIA_ALIGN_ENOMEM, "workspace of %zu bytes is below three chunks of %zu bytes", want, chunk_bytes);
  if (a->d_pool) { ACHK(a, hipStreamSynchronize(a->stream)); hipFree(a->d_pool); hipFree(a->d_ctl); hipFree(a->d_stack); a->d_pool = nullptr; a->d_ctl = nullptr; a->d_stack = nullptr; a->n_chunks = 0; }
  ACHK(a, hipMalloc(&a->d_pool, n_chunks * chunk_bytes));
  ACHK(a, hipMalloc(&a->d_ctl, sizeof(PoolCtl)));
  ACHK(a, hipMalloc(&a->d_stack, n_chunks * sizeof(int)));
  a->n_chunks = (int)n_chunks; a->chunk_log2 = lg;
  a->h_stack.resize(n_chunks);
  if (grew) *grew = true;
  return 0;
}

// The kernel over all queries of the pool, then again over those that found the workspace empty, until none is left.
int run_passes(uvaia_aligner *a)
{
  int n_todo = a->n;
  int blocks = std::max(1, std::min(std::min(n_todo, a->max_blocks), a->n_chunks / 3));
  const int *todo = a->d_order;                                       // the first launch: every query of the pool, the dearest first
  std::vector<int> status((size_t)a->n), list, todo_list;             // todo_list empty: all queries of the pool
  for (;;) {
    // every block keeps two chunks for the whole launch; the rest of the pool is what the queries in flight share
    const PoolCtl ctl{0, a->n_chunks - 2 * blocks, a->n_chunks, a->chunk_log2};
    for (int i = 0; i < ctl.top; i++) a->h_stack[(size_t)i] = 2 * blocks + i;
    ACHK(a, hipMemcpyAsync(a->d_ctl, &ctl, sizeof ctl, hipMemcpyHostToDevice, a->stream));
    if (ctl.top > 0) ACHK(a, hipMemcpyAsync(a->d_stack, a->h_stack.data(), (size_t)ctl.top * sizeof(int), hipMemcpyHostToDevice, a->stream));
    ACHK(a, hipMemsetAsync(a->d_next, 0, sizeof(int), a->stream));
    ACHK(a, hipStreamSynchronize(a->stream));                     // (ctl is on the stack of this function)
    hipLaunchKernelGGL(wfa_align_kernel, dim3((unsigned)blocks), dim3(TPB), 0, a->stream, a->d_ref, a->plen, a->d_seqs, a->d_off, todo, n_todo, a->d_aln, (size_t)a->plen + 1,
                       a->d_score, a->d_status, a->d_cells, a->d_pool, a->d_ctl, a->d_stack, a->d_next, a->P);
    ACHK(a, hipGetLastError());
    a->passes++;
    ACHK(a, hipMemcpyAsync(status.data(), a->d_status, (size_t)a->n * sizeof(int), hipMemcpyDeviceToHost, a->stream));
    ACHK(a, hipStreamSynchronize(a->stream));
    list.clear();
    auto look = [&](int i) -> int {
      if (status[(size_t)i] == ST_OVERFLOW) list.push_back(i);
      else if (status[(size_t)i] == ST_MAXSCORE) return afail(a, UVAIA_ALIGN_EINVAL, "sequence %d: alignment score above %d, the size of the reference's score table", i, a->P.max_score);
      else if (status[(size_t)i] == ST_HDRPAGES) return afail(a, UVAIA_ALIGN_EINVAL, "sequence %d: alignment score above %d, the most this aligner keeps score headers for (%d pages of %d scores)", i, MAX_HDR_PAGES * HDR_PAGE_SCORES - 1, MAX_HDR_PAGES, HDR_PAGE_SCORES);
      else if (status[(size_t)i] == ST_TOOWIDE) return afail(a, UVAIA_ALIGN_EINVAL, "sequence %d: a wavefront wider than the workspace's chunks hold (%zu bytes each)", i, (size_t)4 << a->chunk_log2);
      else if (status[(size_t)i] != ST_OK) return afail(a, UVAIA_ALIGN_ESTATE, "sequence %d: inconsistent backtrace", i);
      return 0;
    };
    // (the status array holds what the last kernel to run a query left: only the queries of this launch are looked at)
    if (todo_list.empty()) { for (int i = 0; i < a->n; i++) { const int rc = look(i); if (rc) return rc; } }
    else for (int i : todo_list) { const int rc = look(i); if (rc) return rc; }
    if (list.empty()) break;
    bool grew = false;                                             // queries found the pool empty: a workspace of the library's choosing grows first, ...
    { const int rc = ensure_pool_memory(a, true, &grew); if (rc) return rc; }
    if (!grew) {                                                   // ... then fewer queries go in flight
      if (blocks == 1) return afail(a, UVAIA_ALIGN_ENOMEM, "sequence %d needs more than the whole workspace (%zu bytes) for its wavefronts", list[0], ((size_t)a->n_chunks * 4) << a->chunk_log2);
      blocks = std::max(1, std::min((int)list.size(), blocks / 4));
    }
    todo_list = list;
    ACHK(a, hipMemcpyAsync(a->d_todo, todo_list.data(), todo_list.size() * sizeof(int), hipMemcpyHostToDevice, a->stream));
    ACHK(a, hipStreamSynchronize(a->stream));
    todo = a->d_todo; n_todo = (int)todo_list.size();
    if (grew) blocks = std::max(1, std::min(std::min(n_todo, a->max_blocks), a->n_chunks / 3));
  }
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
  hipFree(a->d_todo); hipFree(a->d_order); hipFree(a->d_cells); hipFree(a->d_pool); hipFree(a->d_ctl); hipFree(a->d_stack);
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
    return afail(nullptr, UVAIA_ALIGN_EINVAL, "mismatch and gap extension must be positive, gap opening not negative, mismatch and opening + extension below %d (mismatch %d, gap opening %d, gap extension %d)", RING, opt.mismatch, opt.gap_opening, opt.gap_extension);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return afail(nullptr, UVAIA_ALIGN_ENODEV, "no HIP device: the aligner runs on an MI355X (gfx950) and has no CPU path");
  if (device < 0 || device >= ndev) return afail(nullptr, UVAIA_ALIGN_EINVAL, "device %d out of range (%d devices)", device, ndev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return afail(nullptr, UVAIA_ALIGN_ENODEV, "cannot query device %d", device);
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return afail(nullptr, UVAIA_ALIGN_ENODEV, "device %d is %s: this library is built for gfx950 only", device, prop.gcnArchName);
  int caller_device = -1;
  if (hipGetDevice(&caller_device) != hipSuccess) caller_device = -1;
  if (hipSetDevice(device) != hipSuccess) return afail(nullptr, UVAIA_ALIGN_ENODEV, "cannot select device %d", device);
  struct RestoreDevice { int d; ~RestoreDevice() { if (d >= 0) hipSetDevice(d); } } restore_{caller_device};   // every later call selects a->device itself
  uvaia_aligner *a = new uvaia_aligner();
  a->device = device; a->plen = ref_len;
  a->P.x = opt.mismatch; a->P.oe = opt.gap_opening + opt.gap_extension; a->P.e = opt.gap_extension;
  a->P.min_wf_len = opt.min_wavefront_length; a->P.max_dist_thr = opt.max_distance_threshold;
  { int g = a->P.x, b = a->P.oe; while (b) { const int t = g % b; g = b; b = t; } b = a->P.e; while (b) { const int t = g % b; g = b; b = t; } a->P.g = g; }
  // table size of affine_wavefronts_new_reduced (L, 3 L, ..): min(L, 3L) * mismatch + gap_opening + |L - 3L| * gap_extension
  const long long ms = (long long)ref_len * opt.mismatch + opt.gap_opening + 2LL * ref_len * opt.gap_extension;
  a->P.max_score = (int)std::min<long long>(ms, 0x3fffffff);
  a->workspace_request = opt.workspace_bytes;
  {   // as many blocks as the chip holds at once (the blocks are persistent: more would only queue behind them)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wfa_align_kernel, TPB, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    a->max_blocks = opt.max_blocks > 0 ? opt.max_blocks : prop.multiProcessorCount * per_cu;
  }
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
  {   // the order the queries are started in (d_status is free until the run: it takes the estimates)
    hipLaunchKernelGGL(expected_cost_kernel, dim3((unsigned)n), dim3(256), 0, a->stream, a->d_seqs, a->d_off, a->plen, n, a->d_status);
    ACHK(a, hipGetLastError());
    std::vector<int> cost((size_t)n), order((size_t)n);
    ACHK(a, hipMemcpyAsync(cost.data(), a->d_status, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, a->stream));
    ACHK(a, hipStreamSynchronize(a->stream));
    for (int i = 0; i < n; i++) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost[(size_t)x] > cost[(size_t)y]; });
    // a pool of fewer than three rounds of blocks stays in descending order throughout (longest first: with so few queries per block
    // the order is most of the launch's length, and the dearest third of such a pool in flight at once is what the workspace of the
    // library's choosing holds: measured at 2 000 queries); a larger one would put only its very dearest in flight at once
    if (a->workspace_request != 0 || (long long)n * 4 > (long long)a->max_blocks * 11) std::sort(order.begin(), order.begin() + n / 2);
    ACHK(a, hipMemcpyAsync(a->d_order, order.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, a->stream));
  }
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
  int rc = ensure_pool_memory(a, false, nullptr); if (rc) return rc;
  ACHK(a, hipMemsetAsync(a->d_cells, 0, sizeof(unsigned long long), a->stream));
  ACHK(a, hipEventRecord(a->ev_a, a->stream));
  rc = run_passes(a); if (rc) return rc;
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
  if (wavefront_bytes) *wavefront_bytes = (double)a->cells * 33.0;     // per cell: M offset + provenance byte kept, I and D offsets to the ring, five offsets read
  if (passes) *passes = a->passes;
  if (kernel_ms) *kernel_ms = a->kernel_ms;
  return 0;
}

}  // extern "C"

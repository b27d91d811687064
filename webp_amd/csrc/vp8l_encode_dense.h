// vp8l_encode_dense.h -- exact match search (LZ77<uint32_t>::pack, lz77.h:51-72) for tiles that the candidate lists of
// k_lz77_sparse give up on, before the exhaustive kernel has to take them.  Included by vp8l_encode.hip.
//
// k_lz77_dominant: tiles in which ONE value holds most of the pixels (predictor residuals of flat or smooth pictures:
// zeros in runs, single other pixels between them).  The candidate lists fail there because every run of the frequent
// value v0 is a source for every position inside such a run (150 runs in a window, two candidates each).  For a position
// p inside a run of v0 with F pixels of v0 from p on and the pixel y != v0 behind them (at p' = p + F):
//   * a source q of value v0 with Z(q) pixels of v0 from q on matches for min(Z(q), F) pixels when Z(q) != F, so its
//     length is min(Z(q), F, d): over the runs of v0 in the window that is a question about run lengths alone --
//     the first run (farthest) whose part inside the window is at least F long gives F, else the longest gives its
//     length; within p's own run the farthest source gives min(F, its distance).  Answered from the list of v0's runs
//     in position order with maxima over groups of eight.
//   * Z(q) == F (aligned): the match goes on behind the run iff the source's next pixel is y too, i.e. the source is
//     q = c - F for an occurrence c of y with at least F pixels of v0 in front of it, and its length is
//     F + lcp(c, p') -- the same for every position of the run but for F.  So every run's successor pixel p' ("anchor")
//     lists the occurrences c of y among the 1024 before it that follow a pixel of v0, each with the number of v0
//     pixels in front of it and lcp(c, p'), once; the run's positions only look through that list.
// Every other position (15 % of such tiles) is searched as k_lz77_sparse does it for tiles with runs (first pixel of
// every source run of its value and the aligned one), with v0's runs left out of the buckets; an anchor is such a
// position, and the lcp its own search needs is the lcp its list needs.  These positions ("queries") are few, but their
// buckets differ in length by a factor of twenty, and a thread per query left the workgroup waiting for its longest
// walk: so all (query, bucket entry) visits of a tile are numbered through, every thread takes an equal stretch of them,
// and what survives the walk (same value, among the 1024 before the query) is scored where it is found; the scored
// pairs, in the order of the queries, are the anchors' lists.
// Results are identical to the exhaustive kernel's (tests/test_gpu_match_search.py, per position against the oracle).
#pragma once

namespace vp8l {

constexpr uint32_t kDmThreads = 512;
constexpr uint32_t kDmWaves = kDmThreads / 64;
constexpr uint32_t kDmTile = kLzTile;
constexpr uint32_t kDmMaxW = kWindow + kDmTile + kMaxLen;  // the 1024 before the tile, the tile, 127 of look-ahead (+1)
constexpr uint32_t kDmSeg = 7;                             // window pixels per thread in the segment pass
constexpr uint32_t kDmMaxStarts = 1024;                    // run starts of other values a window may hold
constexpr uint32_t kDmMaxRuns = kDmMaxStarts + 2;          // runs of v0 in a window: something else stands between two of them
constexpr uint32_t kDmSub = 1024;                          // positions of a tile that are searched together (a stretch)
constexpr uint32_t kDmMinSub = 32;                         // ... down to this many where the queries find too much
constexpr uint32_t kDmPool = 3072;                         // (query, source) pairs of a stretch
constexpr uint32_t kDmMaxVisits = kDmThreads * 64;         // bucket entries all queries of a tile may have to look at
constexpr uint32_t kDmGrid = 2048;                         // workgroups (two fit a CU)
constexpr uint32_t kDmFewRuns = 32;                        // when this kernel looks first: fewer runs of the frequent value than this are left to the candidate lists
static_assert(kDmThreads * kDmSeg >= kDmMaxW, "segment pass covers the window");
static_assert(kDmMaxW <= (1u << kSpPosBits), "window indices fit an entry");

// Why k_lz77_dominant left tiles to the exhaustive kernel, counted since the last vp8l_debug_dominant_reasons() call:
// [0] no frequent value, [1] too many run starts of other values, [2] too many bucket entries to look at, [3] too many pairs.
// Counted by measurement builds only: additions to one address from every workgroup go one after another, about 12 ns each
// (65 thousand rejected tiles of photo-like planes: 0.75 ms, all that the kernel took for them).
__device__ unsigned int g_dm_reason[4];
#ifdef VP8L_PHASE_TIMING
#define DM_REASON(k) atomicAdd(&g_dm_reason[k], 1u)
#else
#define DM_REASON(k) do {} while (0)
#endif

// Per-phase cycle counts (a measurement build only, -DVP8L_PHASE_TIMING: thread 0 of every workgroup adds the cycles
// between phase marks; read with vp8l_debug_dominant_phase_cycles)
#ifdef VP8L_PHASE_TIMING
__device__ unsigned long long g_dm_phase[12];
#define DM_PHASE_BEGIN() unsigned long long dm_t = __builtin_readcyclecounter()
#define DM_PHASE_MARK(k)                                                       \
  do {                                                                          \
    const unsigned long long dm_n = __builtin_readcyclecounter();               \
    if (threadIdx.x == 0) atomicAdd(&g_dm_phase[k], dm_n - dm_t);               \
    dm_t = dm_n;                                                                \
  } while (0)
#else
#define DM_PHASE_BEGIN() do {} while (0)
#define DM_PHASE_MARK(k) do {} while (0)
#endif

struct DmShared {
  uint32_t px[kDmMaxW + 4];
  uint32_t pool[kDmPool];           // the pairs (query, source) of a stretch, in the order of the queries: first query | source << 11, then
                                    // source | v0 pixels in front of it << 12 | lcp << 19 (a query's pairs are its list)
  uint32_t keys[kDmSub + 2];        // per query of the stretch: its best match so far, length << 11 | distance
  uint32_t cursor2[kSpBuckets / 2]; // bucket ends, 16 bits each
  uint32_t items[kDmMaxStarts];     // run starts of the other values: tag | index
  uint32_t qcnt2[(kDmSub + 4) / 2]; // per query of the stretch: entries of its list, 16 bits each (added to by LDS atomics)
  uint16_t runlen[kDmMaxW];         // pixels equal to pixel i from there on (within the window)
  uint16_t zr[kDmMaxW];             // index of the first run of v0 that ends behind pixel i
  uint16_t rs[kDmMaxRuns], re[kDmMaxRuns];  // runs of v0: first pixel, one past the last
  uint16_t qof[kDmMaxRuns];         // per run of v0 whose anchor is a position of the tile: the anchor's number in `others`
  uint16_t grp[kDmMaxRuns / 8 + 1]; // per 8 runs: longest (capped at 255) << 3 | which one first
  uint16_t others[kDmTile + 2];     // window indices of the tile's positions of other values, in order
  uint16_t qbeg[kDmSub + 2];        // per query of the stretch: where its bucket begins in items
  uint16_t qps[kDmSub + 4];         // per query of the stretch: bucket entries of the queries before it (and the total behind the last)
  uint16_t qoff[kDmSub + 4];        // per query of the stretch: where its list begins in pool
  uint16_t subq[kDmTile / kDmSub + 1];  // per stretch of the tile: tile positions of other values in front of it
  uint8_t len8[kDmMaxRuns];         // run lengths capped at 255
  uint32_t wsum[kDmWaves], wsum2[kDmWaves], wsum3[kDmWaves], wmin[kDmWaves];
};

// Where a tile lies: its image's coded size and width, its positions [a, b), the window [lo, hi) around them
struct DmGeo {
  uint32_t M, xsize, xinv, a, b, lo, W;
  uint64_t pos0;
};

__device__ __forceinline__ DmGeo dm_geometry(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                             const uint32_t *__restrict__ sb_first, uint32_t n, uint32_t tile) {
  constexpr uint32_t kPerSuper = kSuper / kDmTile;
  DmGeo g;
  const uint32_t sb = tile / kPerSuper, img = find_image(sb_first, n, sb);
  g.M = st[img].M; g.xsize = st[img].xsize; g.xinv = st[img].xinv;
  g.a = (tile - sb_first[img] * kPerSuper) * kDmTile;
  g.b = g.a + kDmTile < g.M ? g.a + kDmTile : g.M;
  g.lo = g.a > kWindow ? g.a - kWindow : 0;
  const uint32_t hi = g.b + kMaxLen - 1 < g.M ? g.b + kMaxLen - 1 : g.M;
  g.W = g.a < g.M ? hi - g.lo : 0;  // 0: the tile holds no pixels
  g.pos0 = imgs[img].pos0;
  return g;
}

constexpr uint32_t kDmLoads = (kDmMaxW + 4 * kDmThreads - 1) / (4 * kDmThreads);  // 16-byte loads per thread for a window

// The tile's window must be in S.px (dm_commit_window) before the call.
// first_look: the candidate lists have not seen the tile yet and take it if it is left (then only tiles with many runs of the
// frequent value are searched here).
__device__ __forceinline__ bool lz77_dominant_tile(DmShared &S, const DmGeo &G, uint32_t *__restrict__ match,
                                                   uint8_t *__restrict__ adv, uint8_t *__restrict__ dense_tile, bool first_look) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t M = G.M, xsize = G.xsize, xinv = G.xinv, a = G.a, b = G.b, lo = G.lo, W = G.W;
  const uint64_t pos0 = G.pos0;
  uint32_t *px = S.px;
  DM_PHASE_BEGIN();
  for (uint32_t i = tid; i < kSpBuckets / 2; i += kDmThreads) S.cursor2[i] = 0;
  __syncthreads();
  DM_PHASE_MARK(0);  // (the window was loaded ahead)

  // ---- the frequent value
  const uint32_t s0 = tid * kDmSeg;
  uint32_t v[kDmSeg + 1];
#pragma unroll
  for (uint32_t j = 0; j <= kDmSeg; j++) v[j] = s0 + j < W ? px[s0 + j] : 0u;
  const uint32_t prev = s0 > 0 && s0 - 1 < W ? px[s0 - 1] : 0u;
  // the commonest among 64 samples of the window, of which 16 stand for election (every wave finds it for itself, with the same result)
  uint32_t v0;
  {
    const uint32_t smp = px[(2 * lane + 1) * W / 128];
    uint32_t best = 0;  // votes << 6 | 63 - lane of the probe with the most votes among 16 of the 64 samples (the first among equals)
#pragma unroll
    for (int k = 0; k < 64; k += 4) {
      const uint32_t votes = (uint32_t)__popcll(__ballot((uint32_t)__builtin_amdgcn_readlane((int)smp, k) == smp));
      const uint32_t cand = (votes << 6) | (63u - (uint32_t)k);
      best = cand > best ? cand : best;
    }
    if ((best >> 6) < 6) {  // no value holds a tenth of the samples: the tile stays with the exhaustive kernel
      if (tid == 0) DM_REASON(0);
      return false;
    }
    v0 = (uint32_t)__shfl((int)smp, (int)(63u - (best & 63u)), 64);
  }

  // ---- segment pass: run lengths of all values, the runs of v0 in order, the other values' run starts into buckets
  uint32_t ends = 0, ends0 = 0, starts0 = 0, other_start = 0, mine_other = 0;
#pragma unroll
  for (uint32_t j = 0; j < kDmSeg; j++) {
    const uint32_t i = s0 + j;
    if (i < W) {
      const bool last = i == W - 1 || v[j] != v[j + 1];
      const uint32_t before = j ? v[j - 1] : prev;
      const bool first = i == 0 || before != v[j];
      if (last) ends |= 1u << j;
      if (v[j] == v0) {
        if (last) ends0 |= 1u << j;
        if (first) starts0 |= 1u << j;
      } else {
        if (first) other_start |= 1u << j;
        if (i >= a - lo && i < b - lo) mine_other |= 1u << j;
      }
    }
  }
  {  // run lengths: nearest run end at or behind every pixel (suffix minimum over the threads' first ends)
    const uint32_t firstb = ends ? s0 + (uint32_t)__builtin_ctz(ends) : 0xffffu;
    uint32_t m = firstb;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t other = (uint32_t)__shfl_down((int)m, o, 64);
      if (lane + (uint32_t)o < 64) m = other < m ? other : m;
    }
    if (lane == 0) S.wmin[wave] = m;
    uint32_t after = (uint32_t)__shfl_down((int)m, 1, 64);
    if (lane == 63) after = 0xffffu;
    // counts for the three scans: ends of v0's runs, run starts of other values, the tile's positions of other values
    const uint32_t c_e = (uint32_t)__popc(ends0), c_o = (uint32_t)__popc(other_start), c_m = (uint32_t)__popc(mine_other);
    const uint32_t i_e = wave_incl_scan_u32(c_e, (int)lane), i_o = wave_incl_scan_u32(c_o, (int)lane), i_m = wave_incl_scan_u32(c_m, (int)lane);
    if (lane == 63) { S.wsum[wave] = i_e; S.wsum2[wave] = i_o; S.wsum3[wave] = i_m; }
    __syncthreads();
    {  // too many runs of other values for the buckets (and then, perhaps, of v0 for its run list): not this kernel's tile
      uint32_t starts = 0;
      for (uint32_t w = 0; w < kDmWaves; w++) starts += S.wsum2[w];
      if (starts > kDmMaxStarts) {  // (the same in every thread)
        if (tid == 0) DM_REASON(1);
        return false;
      }
      if (first_look) {
        uint32_t runs = 0;
        for (uint32_t w = 0; w < kDmWaves; w++) runs += S.wsum[w];
        if (runs < kDmFewRuns) return false;  // the candidate lists do such a tile faster
      }
    }
    for (uint32_t w = wave + 1; w < kDmWaves; w++) after = S.wmin[w] < after ? S.wmin[w] : after;
    uint32_t end = after;
#pragma unroll
    for (int j = (int)kDmSeg - 1; j >= 0; j--) {
      if ((ends >> j) & 1u) end = s0 + (uint32_t)j;
      if (s0 + (uint32_t)j < W) S.runlen[s0 + j] = (uint16_t)(end - (s0 + j) + 1);
    }
    uint32_t base_e = i_e - c_e, base_m = i_m - c_m;
    for (uint32_t w = 0; w < wave; w++) { base_e += S.wsum[w]; base_m += S.wsum3[w]; }
    // run k of v0 = the k-th to end; its start has seen k ends before it as well (runs do not nest)
#pragma unroll
    for (uint32_t j = 0; j < kDmSeg; j++) {
      const uint32_t i = s0 + j;
      if (i < W) {
        const uint32_t k = base_e + (uint32_t)__popc(ends0 & ((1u << j) - 1u));
        S.zr[i] = (uint16_t)k;
        if ((starts0 >> j) & 1u) S.rs[k] = (uint16_t)i;
        if ((ends0 >> j) & 1u) S.re[k] = (uint16_t)(i + 1);
        if (i >= a - lo && i < b - lo && ((i - (a - lo)) & (kDmSub - 1)) == 0)  // queries in front of every stretch of the tile
          S.subq[(i - (a - lo)) / kDmSub] = (uint16_t)(base_m + (uint32_t)__popc(mine_other & ((1u << j) - 1u)));
        if ((mine_other >> j) & 1u) {
          const uint32_t qi = base_m + (uint32_t)__popc(mine_other & ((1u << j) - 1u));
          S.others[qi] = (uint16_t)i;
          if (i > 0 && (j ? v[j - 1] : prev) == v0) S.qof[k - 1] = (uint16_t)qi;  // the run that ends in front of it is run k - 1
        }
      }
    }
#pragma unroll
    for (uint32_t j = 0; j < kDmSeg; j++)
      if ((other_start >> j) & 1u) {
        const uint32_t hb = (v[j] * kSpMul) >> (32 - kSpPosBits);
        atomicAdd(&S.cursor2[hb >> 1], 1u << (16 * (hb & 1u)));
      }
  }
  __syncthreads();
  DM_PHASE_MARK(1);  // frequent value, segment pass
  uint32_t R = 0, n_other_starts = 0, n_mine = 0;
  for (uint32_t w = 0; w < kDmWaves; w++) { R += S.wsum[w]; n_other_starts += S.wsum2[w]; n_mine += S.wsum3[w]; }
  // exclusive scan of the bucket counts (8 per thread), then the scatter; afterwards cursor[h] = end of bucket h
  {
    uint32_t c[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t two = S.cursor2[tid * 4 + k];
      c[2 * k] = two & 0xffffu;
      c[2 * k + 1] = two >> 16;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) sum += c[k];
    const uint32_t incl = wave_incl_scan_u32(sum, (int)lane);
    __syncthreads();  // everyone has read the totals of the scans above
    if (lane == 63) S.wsum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t w = 0; w < wave; w++) run += S.wsum[w];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t lo16 = run, hi16 = run + c[2 * k];
      S.cursor2[tid * 4 + k] = lo16 | (hi16 << 16);
      run = hi16 + c[2 * k + 1];
    }
  }
  for (uint32_t r = tid; r < R; r += kDmThreads) {
    const uint32_t l = (uint32_t)S.re[r] - S.rs[r];
    S.len8[r] = (uint8_t)(l < 255 ? l : 255);
  }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < kDmSeg; j++)
    if ((other_start >> j) & 1u) {
      const uint32_t mx = v[j] * kSpMul, hb = mx >> (32 - kSpPosBits), sh16 = 16 * (hb & 1u);
      S.items[(atomicAdd(&S.cursor2[hb >> 1], 1u << sh16) >> sh16) & 0xffffu] = (mx << kSpTagShift) | (s0 + j);
    }
  for (uint32_t g = tid; g * 8 < R; g += kDmThreads) {
    uint32_t best = 0, at = 0;
    for (uint32_t k = 0; k < 8 && g * 8 + k < R; k++) {
      const uint32_t l = S.len8[g * 8 + k];
      if (l > best) { best = l; at = k; }
    }
    S.grp[g] = (uint16_t)((best << 3) | at);
  }
  __syncthreads();
  DM_PHASE_MARK(2);  // buckets, run groups
  const uint16_t *cursor = reinterpret_cast<const uint16_t *>(S.cursor2);
  const uint16_t *qcnt = reinterpret_cast<const uint16_t *>(S.qcnt2);
  const uint32_t ia = a - lo, ib = b - lo;

  // lcp of the pixels from q and from ip on (q < ip), by whole runs, at most `cap` (ip + cap <= W)
  auto lcp_runs = [&](uint32_t q, uint32_t ip, uint32_t cap) {
    uint32_t l = 0;
    do {
      const uint32_t rq = S.runlen[q + l], rp = S.runlen[ip + l];
      l += rq < rp ? rq : rp;
    } while (l < cap && px[q + l] == px[ip + l]);
    return l < cap ? l : cap;
  };

  // ---- the tile in stretches of kDmSub positions: first the queries of the stretch (its positions of other values, in
  // order, and behind them the anchor of the run that holds the stretch's last pixel, if that anchor lies behind the
  // stretch: its list is needed, its own match is not), then the stretch's positions inside runs of v0.
  // A stretch whose queries have more bucket entries to look at, or find more pairs, than there is room for is tried
  // again at half its length (down to kDmMinSub positions; then the tile is the exhaustive kernel's).
  auto queries_before = [&](uint32_t index) -> uint32_t {  // tile positions of other values in front of window index `index`
    if (index >= ib) return n_mine;
    if (((index - ia) & (kDmSub - 1)) == 0) return S.subq[(index - ia) / kDmSub];
    uint32_t l = 0, h = n_mine;
    while (l < h) {
      const uint32_t mid = (l + h) >> 1;
      if (S.others[mid] < index) l = mid + 1; else h = mid;
    }
    return l;
  };
  uint32_t sub = kDmSub;
  for (uint32_t sa = ia; sa < ib;) {
    const uint32_t sb_ = sa + sub < ib ? sa + sub : ib;
    const uint32_t qlo = queries_before(sa), nS = queries_before(sb_) - qlo;
    uint32_t nR = nS, extra_ip = 0;
    if (px[sb_ - 1] == v0) {
      const uint32_t c0 = S.re[S.zr[sb_ - 1]];
      if (c0 < W) { extra_ip = c0; nR++; }
    }
    auto query_index = [&](uint32_t qi) -> uint32_t { return qi < nS ? S.others[qlo + qi] : extra_ip; };
    {  // per query: its bucket, and the numbering of all bucket entries the queries will look at (three queries per thread)
      uint32_t sz[3], sum = 0;
#pragma unroll
      for (uint32_t k = 0; k < 3; k++) {
        const uint32_t qi = tid * 3 + k;
        sz[k] = 0;
        if (qi < nR) {
          const uint32_t h = (px[query_index(qi)] * kSpMul) >> (32 - kSpPosBits);
          const uint32_t beg = h ? cursor[h - 1] : 0u;
          S.qbeg[qi] = (uint16_t)beg;
          sz[k] = cursor[h] - beg;
          S.keys[qi] = 0;
        }
        sum += sz[k];
      }
      for (uint32_t i = tid; i < (kDmSub + 4) / 2; i += kDmThreads) S.qcnt2[i] = 0;
      const uint32_t incl = wave_incl_scan_u32(sum, (int)lane);
      if (lane == 63) S.wsum[wave] = incl;
      __syncthreads();
      uint32_t at = incl - sum, total = 0;
      for (uint32_t w = 0; w < kDmWaves; w++) { if (w < wave) at += S.wsum[w]; total += S.wsum[w]; }
      if (total > kDmMaxVisits) {  // (every thread sees the same total) too much to look at
        if (sub > kDmMinSub) { sub >>= 1; __syncthreads(); continue; }
        if (tid == 0) DM_REASON(2);
        return false;
      }
#pragma unroll
      for (uint32_t k = 0; k < 3; k++) {
        const uint32_t qi = tid * 3 + k;
        if (qi <= nR) S.qps[qi] = (uint16_t)at;  // qps[nR] = the total
        at += sz[k];
      }
    }
    static_assert(kDmThreads * 3 >= kDmSub + 2 && kDmMaxVisits < 65536, "query scan");
    __syncthreads();
    DM_PHASE_MARK(3);  // the stretch's queries numbered
    const uint32_t T = S.qps[nR];
    const uint32_t K = (T + kDmThreads - 1) / kDmThreads;  // visits per thread, at most 64
    const uint32_t w0 = tid * K < T ? tid * K : T, w1 = w0 + K < T ? w0 + K : T;
    // the walk: which of this thread's visits find a source of the query's value among the 1024 before it
    uint64_t hit = 0;
    uint32_t q_first = 0;
    if (w0 < w1) {
      uint32_t lo_q = 0, hi_q = nR;  // largest qi with qps[qi] <= w0
      while (hi_q - lo_q > 1) {
        const uint32_t mid = (lo_q + hi_q) >> 1;
        if (S.qps[mid] <= w0) lo_q = mid; else hi_q = mid;
      }
      q_first = lo_q;
    }
    {
      uint32_t qi = q_first;
      for (uint32_t w = w0; w < w1;) {
        while (S.qps[qi + 1] <= w) qi++;
        const uint32_t next = S.qps[qi + 1], stop = next < w1 ? next : w1;
        const uint32_t ip = query_index(qi);
        const uint32_t tag = (px[ip] * kSpMul) << kSpTagShift, base = (uint32_t)S.qbeg[qi] - S.qps[qi];
        uint32_t local = 0;
        for (; w < stop; w++) {
          const uint32_t it = S.items[base + w], s = it & ((1u << kSpPosBits) - 1);
          if (!((it ^ tag) >> kSpPosBits) && s < ip && s + kWindow >= ip) { hit |= 1ull << (w - w0); local++; }
        }
        if (local) atomicAdd(&S.qcnt2[qi >> 1], local << (16 * (qi & 1u)));
      }
    }
    DM_PHASE_MARK(4);  // the walk
    // The pairs found go into `pool` in the order of the visits, which is the order of the queries: every query's pairs
    // are a stretch of it (its list).  First as (query, source); then every thread takes every kDmThreads-th pair, whoever
    // found it, and replaces it by what the list holds (so a thread that walked through a long bucket of its own
    // query's value does not score all of it alone).
    uint32_t n_pairs = 0;
    {
      const uint32_t c = (uint32_t)__popcll(hit);
      const uint32_t incl = wave_incl_scan_u32(c, (int)lane);
      if (lane == 63) S.wsum2[wave] = incl;
      __syncthreads();  // the counts per query are complete, too
      uint32_t my_off = incl - c;
      for (uint32_t w = 0; w < kDmWaves; w++) { if (w < wave) my_off += S.wsum2[w]; n_pairs += S.wsum2[w]; }
      if (n_pairs > kDmPool) {  // (the same in every thread)
        if (sub > kDmMinSub) { sub >>= 1; __syncthreads(); continue; }
        if (tid == 0) DM_REASON(3);
        return false;
      }
      uint32_t cq[3], sum = 0;
#pragma unroll
      for (uint32_t k = 0; k < 3; k++) {
        const uint32_t qi = tid * 3 + k;
        cq[k] = qi < nR ? qcnt[qi] : 0u;
        sum += cq[k];
      }
      const uint32_t inq = wave_incl_scan_u32(sum, (int)lane);
      if (lane == 63) S.wsum3[wave] = inq;
      uint32_t qi = q_first, next = 0, base = 0;
      uint64_t todo = hit;
      while (todo) {
        const uint32_t k = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t w = w0 + k;
        if (w >= next) {
          while (S.qps[qi + 1] <= w) qi++;
          next = S.qps[qi + 1];
          base = (uint32_t)S.qbeg[qi] - S.qps[qi];
        }
        S.pool[my_off++] = qi | ((S.items[base + w] & ((1u << kSpPosBits) - 1)) << 11);
      }
      __syncthreads();
      uint32_t at = inq - sum;
      for (uint32_t w = 0; w < wave; w++) at += S.wsum3[w];
#pragma unroll
      for (uint32_t k = 0; k < 3; k++) {
        const uint32_t qk = tid * 3 + k;
        if (qk < nR) S.qoff[qk] = (uint16_t)at;
        at += cq[k];
      }
    }
    static_assert(kDmSub + 2 <= (1u << 11), "a query's number fits eleven bits of a pair");
    DM_PHASE_MARK(5);  // the pairs listed (after waiting for the slowest walk)
    // the pairs: lcp of the source run's first pixel and the query (for the query's list and for its own match), and the
    // source aligned with the end of the query's own run (as k_lz77_sparse does it for tiles with runs)
    for (uint32_t j = tid; j < n_pairs; j += kDmThreads) {
      const uint32_t pr = S.pool[j], qi = pr & 2047u, s = pr >> 11;
      const uint32_t ip = query_index(qi);
      const uint32_t cap = W - ip < kMaxLen ? W - ip : kMaxLen;
      const uint32_t e = lcp_runs(s, ip, cap);
      uint32_t zb = 0;  // pixels of v0 in front of the source: 0 keeps it out of every list
      if (s > 0 && px[s - 1] == v0) {
        zb = s - S.rs[S.zr[s - 1]];
        zb = zb < 127 ? zb : 127u;
      }
      S.pool[j] = s | (zb << 12) | ((e < 127 ? e : 127u) << 19);
      if (qi < nS) {  // the query's own match
        const uint32_t p = lo + ip, rem = M - p < kMaxLen ? M - p : kMaxLen, d = ip - s;
        uint32_t l = e < d ? e : d;
        l = l < rem ? l : rem;
        uint32_t key = (l << 11) | d;
        const uint32_t F = S.runlen[ip], r = S.runlen[s];
        if (F < kMaxLen && r > F && s + r <= ip) {
          const uint32_t q2 = s + r - F, d2 = ip - q2, maxl = d2 < rem ? d2 : rem;
          const uint32_t k2 = (lcp_runs(q2, ip, maxl) << 11) | d2;
          key = k2 > key ? k2 : key;
        }
        atomicMax(&S.keys[qi], key);
      }
    }
    DM_PHASE_MARK(6);  // the pairs
    __syncthreads();
    DM_PHASE_MARK(7);  // waiting for the slowest thread's pairs
    // the queries' own matches (a run of the query's value that the window start cuts is one more source run)
    for (uint32_t qi = tid; qi < nS; qi += kDmThreads) {
      const uint32_t ip = S.others[qlo + qi], p = lo + ip, x = px[ip];
      uint32_t key = S.keys[qi];
      const uint32_t wlo = ip > kWindow ? ip - kWindow : 0u;
      if (wlo > 0 && px[wlo] == x && px[wlo - 1] == x) {
        const uint32_t rem = M - p < kMaxLen ? M - p : kMaxLen, F = S.runlen[ip], r = S.runlen[wlo];
        const uint32_t d = ip - wlo, maxl = d < rem ? d : rem;
        const uint32_t k1 = (lcp_runs(wlo, ip, maxl) << 11) | d;
        key = k1 > key ? k1 : key;
        if (F < kMaxLen && r > F && wlo + r <= ip) {
          const uint32_t q2 = wlo + r - F, d2 = ip - q2, maxl2 = d2 < rem ? d2 : rem;
          const uint32_t k2 = (lcp_runs(q2, ip, maxl2) << 11) | d2;
          key = k2 > key ? k2 : key;
        }
      }
      if (p == M - 1) key = 0;  // a lone look-ahead pixel is always a literal (lz77.h:42-49)
      const uint32_t o = key ? ((key >> 11) | ((key & 2047u) << 8) | (dist_code_of(xsize, xinv, key & 2047u) << 19)) : 0u;
      match[pos0 + p] = o;
      adv[pos0 + p] = (uint8_t)((o & 0xff) ? (o & 0xff) : 1u);
    }

    DM_PHASE_MARK(8);  // own matches stored
    // ---- the stretch's positions inside runs of v0
    for (uint32_t ip = sa + tid; ip < sb_; ip += kDmThreads) {
      if (px[ip] != v0) continue;
      const uint32_t p = lo + ip;
      const uint32_t r = S.zr[ip], rs_own = S.rs[r], E = S.re[r];
      const uint32_t rem = M - p < kMaxLen ? M - p : kMaxLen;
      const uint32_t F = E - ip < rem ? E - ip : rem;
      const uint32_t wlo = ip > kWindow ? ip - kWindow : 0u;
      uint32_t key = 0;
      {  // sources in the position's own run: the farthest one
        const uint32_t s_own = rs_own > wlo ? rs_own : wlo, d_own = ip - s_own;
        if (d_own) key = ((F < d_own ? F : d_own) << 11) | d_own;
      }
      // earlier runs: the first whose part inside the window is F long, else the longest (the first of those)
      uint32_t q = S.zr[wlo];
      if (q < r) {
        uint32_t best = 0, best_s = 0;
        bool found = false;
        {
          const uint32_t s = S.rs[q] > wlo ? S.rs[q] : wlo, l = (uint32_t)S.re[q] - s;
          best = l < 255 ? l : 255;
          best_s = s;
          found = best >= F;
          q++;
        }
        while (!found && q < r && (q & 7u)) {
          const uint32_t l = S.len8[q];
          if (l > best) { best = l; best_s = S.rs[q]; found = l >= F; }
          q++;
        }
        while (!found && q + 8 <= r) {
          const uint32_t g = S.grp[q >> 3], l = g >> 3;
          if (l >= F) {  // the first run of the group that is F long (not the group's longest)
            uint32_t k = 0;
            while (S.len8[q + k] < F) k++;
            best = S.len8[q + k];
            best_s = S.rs[q + k];
            found = true;
          } else if (l > best) {
            best = l;
            best_s = S.rs[q + (g & 7u)];
          }
          q += 8;
        }
        while (!found && q < r) {
          const uint32_t l = S.len8[q];
          if (l > best) { best = l; best_s = S.rs[q]; found = l >= F; }
          q++;
        }
        const uint32_t kk = ((best < F ? best : F) << 11) | (ip - best_s);
        key = kk > key ? kk : key;
      }
      if (F < rem) {  // aligned sources: the list of the run's anchor (F < rem: the run ends inside the window, so it has one)
        const uint32_t qa = E < sb_ ? (uint32_t)S.qof[r] - qlo : nS, off = S.qoff[qa], cnt = qcnt[qa];
        auto entry = [&](uint32_t ent) {
          if (((ent >> 12) & 127u) < F) return;
          const uint32_t d = ip + F - (ent & ((1u << kSpPosBits) - 1));
          uint32_t l = F + (ent >> 19);
          l = l < d ? l : d;
          l = l < rem ? l : rem;
          const uint32_t kk = (l << 11) | d;
          key = kk > key ? kk : key;
        };
        uint32_t i = 0;
        for (; i + 4 <= cnt; i += 4) {  // independent reads, four at a time
          uint32_t ent[4];
#pragma unroll
          for (uint32_t u = 0; u < 4; u++) ent[u] = S.pool[off + i + u];
#pragma unroll
          for (uint32_t u = 0; u < 4; u++) entry(ent[u]);
        }
        for (; i < cnt; i++) entry(S.pool[off + i]);
      }
      if (p == M - 1) key = 0;
      const uint32_t o = key ? ((key >> 11) | ((key & 2047u) << 8) | (dist_code_of(xsize, xinv, key & 2047u) << 19)) : 0u;
      match[pos0 + p] = o;
      adv[pos0 + p] = (uint8_t)((o & 0xff) ? (o & 0xff) : 1u);
    }
    DM_PHASE_MARK(9);  // positions of v0
    __syncthreads();  // the stretch's keys, numbering and lists are the next stretch's
    DM_PHASE_MARK(10);  // waiting for the slowest of them
    sa = sb_;
    if (((sa - ia) & (kDmSub - 1)) == 0) sub = kDmSub;  // back to whole stretches at their borders
  }
  if (tid == 0) dense_tile[(pos0 + a) / kLzTile] = kTileDoneDominant;
  return true;
}

// =============================================================================================================
// k_lz77_grams' part: tiles without a frequent value but with few values that repeat everywhere (palette indices
// bundled into words).  Half of the positions of such tiles have matches of 128 words, the others of 2 to 15, and
// 30 to 110 of the 1024 sources start with the position's word: lists of single words are long, and comparing word
// by word is slow.  Here the sources are numbered by what follows them: id1[i] numbers the word at i (equal numbers,
// equal words), id2[i] the pair of words at i, then id4, id8, id16 -- each from the pair (id[i], id[i + half]) of the
// level below, through an exact table (the pair is the key, compared in full; the number is its slot).  Then
//   * lcp(q, p) takes steps of 16, then 4, then 1 (at most 14 comparisons for 128 words);
//   * the sources of a position are looked up by the longest gram first: the list of sources with the position's
//     16-gram holds every source that matches for 16 words or more, so if the best of them (scored in full) reaches 16,
//     nothing else need be looked at; else the list of its 4-gram, with the same argument for 4; then pairs; and what is
//     still open then matches for one word at best, where the farthest source of the word wins (lz77.h:68).
// Positions that a level settles leave; the others are collected for the next level.
// =============================================================================================================
constexpr uint32_t kGmSlots = 8192;          // slots of the table that numbers the grams of a level (at most kDmMaxW are used)
constexpr uint32_t kGmEmpty = 0xffffffffu;
constexpr uint32_t kGmNone = 0xffffu;        // end of a list
constexpr uint32_t kGmUndefined = kGmSlots;  // numbers from here on: a gram that reaches past the window's end (each one its own)
constexpr uint32_t kGmBudget = 6000;         // list entries + comparisons one thread may spend on its positions
constexpr uint8_t kTileDoneGrams = 5;        // tile flag: searched here
static_assert(kGmUndefined + kDmMaxW < 65536, "numbers fit 16 bits");

#ifdef VP8L_PHASE_TIMING
__device__ unsigned long long g_gm_phase[12];
#define GM_PHASE_MARK(k)                                                       \
  do {                                                                          \
    const unsigned long long gm_n = __builtin_readcyclecounter();               \
    if (threadIdx.x == 0) atomicAdd(&g_gm_phase[k], gm_n - gm_t);               \
    gm_t = gm_n;                                                                \
  } while (0)
#define GM_PHASE_BEGIN() unsigned long long gm_t = __builtin_readcyclecounter()
#else
#define GM_PHASE_MARK(k) do {} while (0)
#define GM_PHASE_BEGIN() do {} while (0)
#endif

struct GmShared {
  union {
    uint32_t px[kDmMaxW];              // the window (until the words are numbered)
    struct {
      uint32_t keys[kDmTile];          // per position of the tile: best match so far, length << 11 | distance
      uint16_t todo0[kDmTile];         // positions (window indices) still open: the list of the levels 16 and 2
    } s;
  };
  uint32_t table[kGmSlots];            // numbering a level: the keys; searching a level: per number the first source of its list
  uint16_t id1[kDmMaxW], id4[kDmMaxW];
  uint16_t idx[kDmMaxW];               // id2 while id4 is made from it, id16 for the search of that level, id2 again for the search of pairs
  uint16_t next[kDmMaxW];              // numbering: id8; searching: the next source of the same list
  uint16_t todo1[kDmTile];             // ... the list of the levels 4 and 1
  uint32_t n_open[2];
  uint32_t abort_flag;
};
static_assert(sizeof(GmShared) + 64 <= 80 * 1024 && sizeof(DmShared) + 64 <= 80 * 1024, "two workgroups share a CU's 160 KB");

// number of `key` in the table (inserted if new); keys are below kGmEmpty
__device__ __forceinline__ uint32_t gm_number(uint32_t *table, uint32_t key) {
  uint32_t slot = (key * 0x9E3779B1u) >> (32 - 13);
  for (;;) {
    const uint32_t old = atomicCAS(&table[slot], kGmEmpty, key);
    if (old == kGmEmpty || old == key) return slot;
    slot = (slot + 1) & (kGmSlots - 1);
  }
}

__device__ __forceinline__ bool lz77_grams_tile(GmShared &S, const DmGeo &G, uint32_t *__restrict__ match,
                                                uint8_t *__restrict__ adv, uint8_t *__restrict__ dense_tile) {
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t M = G.M, xsize = G.xsize, xinv = G.xinv, a = G.a, b = G.b, lo = G.lo, W = G.W;
  const uint64_t pos0 = G.pos0;
  const uint32_t ia = a - lo, ib = b - lo;
  GM_PHASE_BEGIN();
  // ---- numbering: words, then pairs of numbers, level by level
  for (uint32_t i = tid; i < kGmSlots; i += kDmThreads) S.table[i] = kGmEmpty;
  if (tid == 0) { S.abort_flag = 0; S.n_open[0] = 0; S.n_open[1] = 0; }
  __syncthreads();
  for (uint32_t i = tid; i < W; i += kDmThreads) {
    const uint32_t v = S.px[i];
    // (a word equal to the empty mark is numbered under another key that no word has twice: its own complement's slot is as good as any)
    S.id1[i] = (uint16_t)gm_number(S.table, v == kGmEmpty ? 0xfffffffeu : v);
  }
  __syncthreads();
  // a word 0xfffffffe and a word 0xffffffff would share a number: tell them apart again (both in one window: never seen, but exactness)
  // -- by giving every 0xffffffff the undefined number of its place when a 0xfffffffe exists is not exact either; so such a tile is left
  {
    bool clash = false;
    for (uint32_t i = tid; i < W; i += kDmThreads) clash |= S.px[i] == kGmEmpty;
    if (__syncthreads_or(clash)) {
      bool other = false;
      for (uint32_t i = tid; i < W; i += kDmThreads) other |= S.px[i] == 0xfffffffeu;
      if (__syncthreads_or(other)) return false;  // (the same in every thread)
    }
  }
  auto level_up = [&](const uint16_t *from, uint16_t *to, uint32_t half) {  // to[i] numbers (from[i], from[i + half])
    for (uint32_t i = tid; i < kGmSlots; i += kDmThreads) S.table[i] = kGmEmpty;
    __syncthreads();
    for (uint32_t i = tid; i < W; i += kDmThreads) {
      uint32_t id = kGmUndefined + i;
      if (i + half < W) {
        const uint32_t x = from[i], y = from[i + half];
        if (x < kGmUndefined && y < kGmUndefined) id = gm_number(S.table, (x << 13) | y);
      }
      to[i] = (uint16_t)id;
    }
    __syncthreads();
  };
  GM_PHASE_MARK(0);  // words numbered
  level_up(S.id1, S.idx, 1);    // id2, on the way to id4 (made again when the search gets to pairs)
  level_up(S.idx, S.id4, 2);
  level_up(S.id4, S.next, 4);   // id8, only on the way to id16
  level_up(S.next, S.idx, 8);   // id16
  GM_PHASE_MARK(1);  // grams numbered
  // (the window's words are not needed any more: every comparison is one of numbers) -- their memory holds the keys from here on
  for (uint32_t t = tid; t < kDmTile; t += kDmThreads) {
    S.s.keys[t] = 0;
    if (ia + t < ib) S.s.todo0[t] = (uint16_t)(ia + t);
  }
  uint32_t n_open = ib - ia;
  uint32_t spent = 0;
  bool have16 = true;
  auto lcp = [&](uint32_t q, uint32_t ip, uint32_t cap) {
    uint32_t l = 0;
    // (steps of 16 only while the 16-grams are there: on the later levels no match reaches 16)
    while (have16 && l + 16 <= cap && S.idx[q + l] == S.idx[ip + l]) { l += 16; spent++; }
    while (l + 4 <= cap && S.id4[q + l] == S.id4[ip + l]) { l += 4; spent++; }
    while (l < cap && S.id1[q + l] == S.id1[ip + l]) { l++; spent++; }
    return l;
  };
  const uint32_t nsrc = ib;  // sources: window indices below the tile's end
  uint32_t cur = 0;
#pragma unroll 1
  for (uint32_t level = 0; level < 4; level++) {
    const uint16_t *ids = level == 0 ? S.idx : level == 1 ? S.id4 : level == 2 ? S.idx : S.id1;
    const uint32_t L = level == 0 ? 16u : level == 1 ? 4u : level == 2 ? 2u : 1u;
    if (level == 1) have16 = false;
    if (level == 2) level_up(S.id1, S.idx, 1);  // the pairs again (their memory held the 16-grams meanwhile)
    // the lists of this level: every source whose gram lies inside the window, under its number
    for (uint32_t i = tid; i < kGmSlots; i += kDmThreads) S.table[i] = kGmNone;
    if (tid == 0) S.n_open[cur ^ 1] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < nsrc; i += kDmThreads) {
      const uint32_t x = ids[i];
      if (x < kGmUndefined) S.next[i] = (uint16_t)atomicExch(&S.table[x], i);
    }
    __syncthreads();
    GM_PHASE_MARK(2);  // lists (and the pairs numbered again)
    for (uint32_t j = tid; j < ((n_open + 63) & ~63u); j += kDmThreads) {
      bool still_open = false;
      uint32_t ip = 0;
      if (j < n_open) {
        ip = (cur ? S.todo1 : S.s.todo0)[j];
        const uint32_t p = lo + ip, t = ip - ia;
        const uint32_t rem = M - p < kMaxLen ? M - p : kMaxLen;
        uint32_t key = S.s.keys[t];
        const uint32_t x = ids[ip];
        if (x < kGmUndefined) {
          uint32_t q = S.table[x];
          if (L == 1) {
            uint32_t far = 0;
            while (q != kGmNone) {
              spent++;
              if (q < ip && q + kWindow >= ip && ip - q > far) far = ip - q;
              q = S.next[q];
            }
            const uint32_t k1 = far ? ((1u << 11) | far) : 0u;
            key = k1 > key ? k1 : key;
          } else {
            while (q != kGmNone) {
              spent++;
              if (q < ip && q + kWindow >= ip) {
                const uint32_t d = ip - q, cap = d < rem ? d : rem;
                // what it must reach to replace the best so far: longer, or as long and farther (lz77.h:68)
                const uint32_t need = (key >> 11) + (d > (key & 2047u) ? 0u : 1u);
                if (cap >= need && (need <= 1 || S.id1[q + need - 1] == S.id1[ip + need - 1])) {
                  const uint32_t l = lcp(q, ip, cap);
                  if (l >= need) key = (l << 11) | d;
                }
              }
              q = S.next[q];
            }
          }
        }
        if (L > 1 && (key >> 11) < L && p != M - 1) {
          S.s.keys[t] = key;
          still_open = true;
        } else {
          if (p == M - 1) key = 0;  // a lone look-ahead pixel is always a literal (lz77.h:42-49)
          const uint32_t o = key ? ((key >> 11) | ((key & 2047u) << 8) | (dist_code_of(xsize, xinv, key & 2047u) << 19)) : 0u;
          match[pos0 + p] = o;
          adv[pos0 + p] = (uint8_t)((o & 0xff) ? (o & 0xff) : 1u);
        }
      }
      // the positions that go on to the next level (whole waves get here)
      const uint64_t m = __ballot(still_open);
      if (m) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&S.n_open[cur ^ 1], (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (still_open) (cur ? S.s.todo0 : S.todo1)[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)ip;
      }
      if (spent > kGmBudget) S.abort_flag = 1;
    }
    GM_PHASE_MARK(3 + level);  // search of the level (thread 0's share)
    __syncthreads();
    GM_PHASE_MARK(7 + level);  // waiting for the others
    if (S.abort_flag) return false;  // too much to look at: the exhaustive kernel redoes the tile
    cur ^= 1;
    n_open = S.n_open[cur];
    if (n_open == 0) break;
  }
  if (tid == 0) dense_tile[(pos0 + a) / kLzTile] = kTileDoneGrams;
  return true;
}

// The tiles that k_lz77_sparse left to the exhaustive kernel.  Workgroups stay and take tiles from a counter, 16
// consecutive tile numbers at a time (a look at their flags tells which of them are waiting); the window of a
// workgroup's next tile is on its way from memory, into registers, while the current one is searched.  A tile that is
// searched here gets the flag kTileDoneDominant; one that is not (no frequent value, too many run starts or list
// entries) keeps its flag and is the exhaustive kernel's.  `counter` starts at 0 (cleared with the flags).
constexpr uint32_t kDmChunk = 16, kDmChunkFirstLook = 4;
// SEARCH(S, geo) searches one tile whose window is in S.px; the rest is the same for both kernels of this file.
template <class Shared, class Search>
__device__ __forceinline__ void dense_tiles(Shared &S, const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                            const uint32_t *__restrict__ sb_first, uint32_t n, uint32_t ntiles, const uint32_t *__restrict__ pix,
                                            const uint8_t *__restrict__ dense_tile, uint32_t *__restrict__ counter, uint32_t take_flag,
                                            const uint32_t *__restrict__ waiting_tiles, Search search) {
  __shared__ uint32_t s_chunk;
  __shared__ uint8_t s_worth[64];  // first look: the chunk's tiles that have a frequent value
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  // waiting_tiles (may be null): an upper bound of the tiles with the flag, counted by the kernel that set it.  None: nothing
  // to do (a look at a million flags of a batch that needs none of this took 0.76 ms); few: the flags are looked at 64 at a time
  const uint32_t n_waiting = waiting_tiles ? *waiting_tiles : ntiles;
  if (n_waiting == 0) return;
  // many waiting: few tile numbers at a time, so that the last workgroups do not work on alone (one tile at a time when there are
  // hardly more tiles than workgroups); few waiting among many: 64 flags at a look
  // (a first look finds its tiles next to one another, 50 us each: in sixteens, the last workgroups worked on alone for 0.8 ms)
  const uint32_t per_group = ntiles / (4 * gridDim.x), most_at_once = take_flag == 0 ? kDmChunkFirstLook : kDmChunk;
  const uint32_t chunk_tiles = (uint64_t)n_waiting * 8 > ntiles ? (per_group < 1 ? 1u : per_group > most_at_once ? most_at_once : per_group) : 64u;
  uint32_t chunk = 0;
  uint64_t waiting = 0;  // tiles of the chunk that are still to do (the same in every thread)
  // the next waiting tile, or ~0u when the counter has run past the last tile; called by all threads
  auto next_tile = [&]() -> uint32_t {
    for (;;) {
      if (waiting) {
        const uint32_t k = (uint32_t)__builtin_ctzll(waiting);
        waiting &= waiting - 1;
        return chunk + k;
      }
      if (tid == 0) s_chunk = atomicAdd(counter, chunk_tiles);
      __syncthreads();
      chunk = s_chunk;
      // When this kernel looks first (every tile is a candidate), the tiles without a frequent value are told here, each from 64
      // samples read straight from memory (the vote that the search begins with, on the same samples), the waves taking the
      // chunk's tiles in turn: the tiles of photo-like predictor planes, three quarters of a mixed batch, cost a window, two
      // barriers and - the most - the chain of loads to their geometry each, one tile after another.
      if (take_flag == 0 && chunk < ntiles) {
        for (uint32_t k = tid >> 6; k < chunk_tiles && chunk + k < ntiles; k += kDmThreads / 64) {
          const DmGeo g = dm_geometry(imgs, st, sb_first, n, chunk + k);
          uint32_t most = 64;
          if (g.W) {
            const uint32_t smp = pix[g.pos0 + g.lo + (2 * lane + 1) * g.W / 128];
            most = 0;
#pragma unroll
            for (int j = 0; j < 64; j += 4) {
              const uint32_t votes = (uint32_t)__popcll(__ballot((uint32_t)__builtin_amdgcn_readlane((int)smp, j) == smp));
              most = votes > most ? votes : most;
            }
          }
          if (lane == 0) {
            s_worth[k] = most >= 6;
            if (most < 6) DM_REASON(0);
          }
        }
      }
      __syncthreads();  // everyone has read s_chunk before it is written again
      if (chunk >= ntiles) return ~0u;
      waiting = __ballot(lane < chunk_tiles && chunk + lane < ntiles && dense_tile[chunk + lane] == take_flag && (take_flag != 0 || s_worth[lane]));
    }
  };
  uint4 ahead[kDmLoads];
  auto fetch_window = [&](const DmGeo &g) {
    const uint32_t *data = pix + g.pos0 + g.lo;  // lo is a multiple of 4 and the pixel array has readable slack past M: 16-byte loads
#pragma unroll
    for (uint32_t k = 0; k < kDmLoads; k++) {
      const uint32_t i = (k * kDmThreads + tid) * 4;
      ahead[k] = i < g.W ? *reinterpret_cast<const uint4 *>(data + i) : make_uint4(0, 0, 0, 0);
    }
  };
  uint32_t tile = next_tile();
  DmGeo geo;
  if (tile != ~0u) { geo = dm_geometry(imgs, st, sb_first, n, tile); fetch_window(geo); }
  while (tile != ~0u) {
#pragma unroll
    for (uint32_t k = 0; k < kDmLoads; k++) {
      const uint32_t i = (k * kDmThreads + tid) * 4;
      if (i < geo.W) *reinterpret_cast<uint4 *>(&S.px[i]) = ahead[k];
    }
    const DmGeo cur = geo;
    const uint32_t following = next_tile();
    if (following != ~0u) { geo = dm_geometry(imgs, st, sb_first, n, following); fetch_window(geo); }
    __syncthreads();  // the window is in place
    if (cur.W) search(S, cur);
    __syncthreads();  // the tile's shared memory is the next tile's
    tile = following;
  }
}

__global__ __launch_bounds__(kDmThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_lz77_dominant(const EncImage *__restrict__ imgs, const EncState *__restrict__ st, const uint32_t *__restrict__ sb_first, uint32_t n,
                     uint32_t ntiles, const uint32_t *__restrict__ pix, uint32_t *__restrict__ match, uint8_t *__restrict__ adv,
                     uint8_t *__restrict__ dense_tile, uint32_t *__restrict__ counter, uint32_t take_flag,
                     const uint32_t *__restrict__ waiting_tiles) {
  __shared__ __attribute__((aligned(16))) DmShared S;
  dense_tiles(S, imgs, st, sb_first, n, ntiles, pix, dense_tile, counter, take_flag, waiting_tiles,
              [&](DmShared &sh, const DmGeo &g) { (void)lz77_dominant_tile(sh, g, match, adv, dense_tile, take_flag == 0); });
}

// What k_lz77_dominant left of the tiles that the candidate lists gave up on.
__global__ __launch_bounds__(kDmThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_lz77_grams(const EncImage *__restrict__ imgs, const EncState *__restrict__ st, const uint32_t *__restrict__ sb_first, uint32_t n,
                  uint32_t ntiles, const uint32_t *__restrict__ pix, uint32_t *__restrict__ match, uint8_t *__restrict__ adv,
                  uint8_t *__restrict__ dense_tile, uint32_t *__restrict__ counter, const uint32_t *__restrict__ waiting_tiles) {
  __shared__ __attribute__((aligned(16))) GmShared S;
  dense_tiles(S, imgs, st, sb_first, n, ntiles, pix, dense_tile, counter, (uint32_t)kTileExhaustive, waiting_tiles,
              [&](GmShared &sh, const DmGeo &g) { (void)lz77_grams_tile(sh, g, match, adv, dense_tile); });
}

}  // namespace vp8l

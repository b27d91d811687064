// Match search along the parse, for palette images (lz77.h:28-93 once more; included by vp8l_encode.hip).
//
// The kernels before this one give every position its best match, and the parse then visits a fraction of them: a token
// of n pixels skips n - 1 positions.  Where the search is expensive per position - bundled palette indices, where every
// word has tens of equal words among the 1024 before it and the exhaustive kernel is the only one that copes - it pays
// to search only where a parse can get to.  Which positions those are is not known inside a tile (the parse enters it
// somewhere among its first 128 positions: a token is at most 128 long), but chains of tokens that start at
// neighbouring positions fall in step after a few tokens, so the union of the chains from ALL 128 possible entries is
// hardly larger than one chain: a fifth to a quarter of the positions on the benchmark's palettes.  Every position of
// that union gets its exact best match; the others keep "no match, advance by one", which no parse ever reads (the
// parse kernels follow the chain from the image's first position and only look at positions on it).
//
// A workgroup of four waves per tile.  A palette image's coded words are 0xff000000 | byte << 8 (vp8l.h:438-459), so the
// window is kept as bytes.  Sources are listed per byte value, in the order of the 256-position parts of the window they
// lie in, so that a position's candidates - the sources of its value in the 1024 before it - are a contiguous piece of a
// list with at most two parts of strangers around it.  A wave searches one position at a time: 64 candidates per round,
// each lane extending its own by four bytes per step; the key length << 11 | distance makes the maximum the reference's
// choice (longest, then farthest: lz77.h:68).  Once a candidate reaches the longest length possible at the position, only
// the rest of its part can hold a farther one of that length, and the rounds stop there.
// Wave 0 follows the chains from the 128 entries, in the order of the positions (a bitmap of positions to look at: a
// searched position adds the one behind its token); waves 1-3 start a chain each at a quarter of the tile, which the
// chains from the entries soon run into - a position searched by anyone is marked, and whoever finds the mark stops
// there, because the one who set it goes on from there.  So every chain from an entry is searched to the tile's end by
// someone, which is all that is needed; what the other waves search beside it is correct and unused.
#pragma once

namespace vp8l {

constexpr uint32_t kWkThreads = 256;
constexpr uint32_t kWkPart = 256;                                  // sources are listed by value and by part of this many window positions
constexpr uint32_t kWkSources = kWindow + kLzTile;                 // window indices that can be a source of a position of the tile
constexpr uint32_t kWkParts = kWkSources / kWkPart;
constexpr uint32_t kWkWords = (kDmMaxW + 16 + 3) / 4;              // the window as bytes, four to a word, with room to read past its end
constexpr uint32_t kWkGrid = 2048;
constexpr uint32_t kWkBudget = 20000;                              // rounds + extension steps a wave may spend on a tile
constexpr uint8_t kTileDoneWalk = 6;                               // tile flag: searched here, along the parse
static_assert(kWkParts == 12 && kWkPart * kWkParts == kWkSources && kWkPart == 256 && kWkThreads == 256, "parts of 256 positions; a thread per byte value");

#ifdef VP8L_PHASE_TIMING
// measurement builds: tiles, positions searched, rounds of candidates, rounds with a live candidate, extension steps, tiles given up
__device__ unsigned long long g_wk_stat[8];
#define WK_STAT(k, n) do { if (lane == 0) atomicAdd(&g_wk_stat[k], (unsigned long long)(n)); } while (0)
#else
#define WK_STAT(k, n) do {} while (0)
#endif

struct WkShared {
  uint32_t by4[kWkWords];
  uint16_t list[kWkSources];              // window indices, grouped by (value, part)
  uint16_t ends[kWkParts][256];           // after the lists are filled: where the entries of (part, value) end (= where those of the next part begin)
  uint32_t vis[kLzTile / 32];             // positions of the tile that somebody searches or has searched
  uint32_t wsum[kWkThreads / 64];
  uint32_t abort_flag;
};

// four bytes of the window from byte `off` on
__device__ __forceinline__ uint32_t wk_bytes(const uint32_t *by4, uint32_t off) {
  const uint32_t w0 = by4[off >> 2], w1 = by4[(off >> 2) + 1];
  return __builtin_amdgcn_alignbyte(w1, w0, off & 3u);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const uint32_t u = (uint32_t)__shfl_xor((int)v, o); v = u < v ? u : v; }
  return v;
}

// The tile's window is loaded here.  Returns false (in every thread) when the tile is left to the exhaustive kernel.
__device__ __forceinline__ bool lz77_walk_tile(WkShared &S, const DmGeo &G, const uint32_t *__restrict__ pix, uint32_t *__restrict__ match,
                                               uint8_t *__restrict__ adv, uint8_t *__restrict__ dense_tile) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t M = G.M, xsize = G.xsize, xinv = G.xinv, a = G.a, b = G.b, lo = G.lo, W = G.W;
  const uint64_t pos0 = G.pos0;
  const uint32_t ia = a - lo, ib = b - lo, nt = b - a;
  uint32_t *ends32 = reinterpret_cast<uint32_t *>(&S.ends[0][0]);
  for (uint32_t i = tid; i < kWkParts * 256 / 2; i += kWkThreads) ends32[i] = 0;
  for (uint32_t i = tid; i < kLzTile / 32; i += kWkThreads) S.vis[i] = 0;
  if (tid == 0) S.abort_flag = 0;
  // positions nobody gets to: no match, advance by one
  for (uint32_t t = tid * 4; t < nt; t += kWkThreads * 4) {
    if (t + 4 <= nt) {
      *reinterpret_cast<uint4 *>(match + pos0 + a + t) = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint32_t *>(adv + pos0 + a + t) = 0x01010101u;
    } else {
      for (uint32_t k = t; k < nt; k++) { match[pos0 + a + k] = 0; adv[pos0 + a + k] = 1; }
    }
  }
  // (these stores have arrived before anyone stores a result over them: a wave's stores to one address keep their order, but
  // the results come from other waves)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- the window as bytes, and how many sources every (part, value) has
  const uint32_t *data = pix + pos0 + lo;  // lo is a multiple of 4 and the pixel array has readable slack past M: 16-byte loads
  for (uint32_t i4 = tid; i4 < kWkWords; i4 += kWkThreads) {
    const uint32_t i = i4 * 4;
    uint32_t packed = 0;
    if (i < W) {
      const uint4 v = *reinterpret_cast<const uint4 *>(data + i);
      packed = ((v.x >> 8) & 0xffu) | (v.y & 0xff00u) | ((v.z << 8) & 0xff0000u) | ((v.w << 16) & 0xff000000u);
      if (i + 4 > W) packed &= 0xffffffffu >> (8 * (i + 4 - W));  // (what lies past the window reads as zero)
#pragma unroll
      for (uint32_t j = 0; j < 4; j++)
        if (i + j < ib) {
          const uint32_t c = ((i >> 8) << 8) | ((packed >> (8 * j)) & 0xffu);
          atomicAdd(&ends32[c >> 1], (c & 1u) ? 65536u : 1u);
        }
    }
    S.by4[i4] = packed;
  }
  __syncthreads();
  // ---- counts -> where each (value, part) begins: values in order, the parts of a value in order
  {
    uint32_t cnt[kWkParts], total = 0;
#pragma unroll
    for (uint32_t k = 0; k < kWkParts; k++) { cnt[k] = S.ends[k][tid]; total += cnt[k]; }
    const uint32_t incl = wave_incl_scan_u32(total, (int)lane);
    if (lane == 63) S.wsum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - total;
    for (uint32_t w = 0; w < wave; w++) run += S.wsum[w];
#pragma unroll
    for (uint32_t k = 0; k < kWkParts; k++) { S.ends[k][tid] = (uint16_t)run; run += cnt[k]; }
  }
  __syncthreads();
  // ---- the lists (a cursor ends where the next part's entries begin)
  for (uint32_t i = tid; i < ib; i += kWkThreads) {
    const uint32_t v = (S.by4[i >> 2] >> (8 * (i & 3u))) & 0xffu;
    const uint32_t c = ((i >> 8) << 8) | v;
    const uint32_t old = atomicAdd(&ends32[c >> 1], (c & 1u) ? 65536u : 1u);
    S.list[(c & 1u) ? (old >> 16) : (old & 0xffffu)] = (uint16_t)i;
  }
  __syncthreads();

  // ---- the walk
  // positions of the tile this wave still has to look at: lane l holds those of 32l .. 32l+31
  uint32_t pend = 0;
  if (wave == 0) {
    const uint32_t entries = a == 0 ? 1u : (nt < kMaxLen ? nt : kMaxLen);  // (an image's parse starts at its first position)
    if (lane * 32 < entries) pend = entries - lane * 32 >= 32 ? 0xffffffffu : (1u << (entries - lane * 32)) - 1u;
  } else if (wave * (kLzTile / 4) < nt) {
    if (lane == wave * (kLzTile / 4) / 32) pend = 1u;
  }
  uint32_t spent = 0;
  for (;;) {
    const uint64_t have = __ballot(pend != 0);
    if (!have) break;
    const uint32_t l0 = (uint32_t)__builtin_ctzll(have);
    const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)pend, (int)l0);
    const uint32_t bit = (uint32_t)__builtin_ctz(word);
    if (lane == l0) pend &= pend - 1;
    const uint32_t t = l0 * 32 + bit;
    uint32_t seen = 0;
    if (lane == 0) seen = atomicOr(&S.vis[t >> 5], 1u << (t & 31u));
    seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)seen);
    if ((seen >> (t & 31u)) & 1u) continue;
    if (S.abort_flag) break;
    const uint32_t ip = ia + t, p = a + t;
    uint32_t key = 0;
    if (p != M - 1) {  // a lone look-ahead pixel is always a literal (lz77.h:42-49)
      const uint32_t rem = M - p < kMaxLen ? M - p : kMaxLen;
      const uint32_t v = (S.by4[ip >> 2] >> (8 * (ip & 3u))) & 0xffu;
      const uint32_t pa = ip >= kWindow ? (ip - kWindow) >> 8 : 0u, pb = ip >> 8;
      const uint32_t cs = pa ? S.ends[pa - 1][v] : (v ? S.ends[kWkParts - 1][v - 1] : 0u);
      const uint32_t ce = S.ends[pb][v];
      uint32_t stop_at = ce;
      uint32_t best = 0;
      const uint32_t t0 = wk_bytes(S.by4, ip);
      WK_STAT(1, 1);
      for (uint32_t c = cs; c < stop_at; c += 64) {
        spent++;
        WK_STAT(2, 1);
        const uint32_t idx = c + lane;
        const uint32_t q = idx < ce ? S.list[idx] : 0xffffu;
        const bool live = q < ip && q + kWindow >= ip;
        if (!__ballot(live)) continue;
        WK_STAT(3, 1);
        const uint32_t d = ip - q, cap = d < rem ? d : rem;  // (meaningless in lanes that are not live)
        uint32_t l = 0;
        bool open = live;
        {
          const uint32_t x = wk_bytes(S.by4, live ? q : 0u) ^ t0;
          if (x) { l = (uint32_t)__builtin_ctz(x) >> 3; open = false; } else { l = 4; open = live && cap > 4; }
        }
        for (uint32_t k = 4; __ballot(open); k += 4) {
          spent++;
          WK_STAT(4, 1);
          const uint32_t x = wk_bytes(S.by4, (open ? q : 0u) + k) ^ wk_bytes(S.by4, ip + k);
          if (open) {
            if (x) { l = k + ((uint32_t)__builtin_ctz(x) >> 3); open = false; }
            else { l = k + 4; open = k + 4 < cap; }
          }
        }
        l = l < cap ? l : cap;
        if (live) { const uint32_t kk = (l << 11) | d; best = kk > best ? kk : best; }
        // a candidate as long as a match can be here: the farthest such wins, and a farther one can only lie in the same part or before
        const uint64_t full = __ballot(live && l == rem);
        if (full) {
          const uint32_t qmin = wave_min_u32(live && l == rem ? q : 0xffffu);
          const uint32_t part_end = S.ends[qmin >> 8][v];
          stop_at = part_end < stop_at ? part_end : stop_at;
        }
      }
      key = wave_max_dpp(best);
    }
    const uint32_t len = key >> 11;
    if (lane == 0) {
      const uint32_t o = key ? (len | ((key & 2047u) << 8) | (dist_code_of(xsize, xinv, key & 2047u) << 19)) : 0u;
      match[pos0 + p] = o;
      adv[pos0 + p] = (uint8_t)(len ? len : 1u);
    }
    const uint32_t t2 = t + (len ? len : 1u);
    if (t2 < nt && lane == (t2 >> 5)) pend |= 1u << (t2 & 31u);
    if (spent > kWkBudget) S.abort_flag = 1;
  }
  __syncthreads();
  const bool ok = S.abort_flag == 0;
  if (tid == 0) { WK_STAT(0, 1); WK_STAT(5, ok ? 0 : 1); }
  if (ok && tid == 0) dense_tile[(pos0 + a) / kLzTile] = kTileDoneWalk;
  __syncthreads();  // (the flag is read before the next tile clears it)
  return ok;
}

// The tiles of palette images that the kernels before have left to the exhaustive kernel.  Workgroups stay and take tile
// numbers from a counter (cleared with the flags), a few at a time.
__global__ __launch_bounds__(kWkThreads) void k_lz77_walk(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                         const uint32_t *__restrict__ sb_first, uint32_t n, uint32_t ntiles,
                                                         const uint32_t *__restrict__ pix, uint32_t *__restrict__ match, uint8_t *__restrict__ adv,
                                                         uint8_t *__restrict__ dense_tile, uint32_t *__restrict__ counter,
                                                         const uint32_t *__restrict__ waiting_tiles) {
  __shared__ __attribute__((aligned(16))) WkShared S;
  __shared__ uint32_t s_chunk;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t n_waiting = *waiting_tiles;  // tiles given up on by the candidate lists (an upper bound of what is left)
  if (n_waiting == 0) return;
  const uint32_t per_group = ntiles / (4 * gridDim.x);
  const uint32_t chunk_tiles = (uint64_t)n_waiting * 8 > ntiles ? (per_group < 1 ? 1u : per_group > 16u ? 16u : per_group) : 64u;
  for (;;) {
    if (tid == 0) s_chunk = atomicAdd(counter, chunk_tiles);
    __syncthreads();
    const uint32_t chunk = s_chunk;
    __syncthreads();  // everyone has read it before it is written again
    if (chunk >= ntiles) return;
    uint64_t waiting = __ballot(lane < chunk_tiles && chunk + lane < ntiles && dense_tile[chunk + lane] == kTileExhaustive);
    while (waiting) {
      const uint32_t tile = chunk + (uint32_t)__builtin_ctzll(waiting);
      waiting &= waiting - 1;
      constexpr uint32_t kPerSuper = kSuper / kLzTile;
      const uint32_t img = find_image(sb_first, n, tile / kPerSuper);
      if (st[img].ncolors == 0) continue;  // not bytes
      const DmGeo g = dm_geometry(imgs, st, sb_first, n, tile);
      if (g.W) (void)lz77_walk_tile(S, g, pix, match, adv, dense_tile);
    }
  }
}

}  // namespace vp8l

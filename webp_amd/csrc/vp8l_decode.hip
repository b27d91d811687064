// vp8l_decode.hip -- MI355X (gfx950) kernels of the VP8L decoder and the batch decoder context.
//
//   k_decode_entropy     stream grammar, Huffman tables, LZ77 / colour-cache pixel decode
//                        (vp8l.h:83-328, huffman_io.h:74-219): one wave per image; lane 0 drives the
//                        sequential state machine of vp8l_dec_core.h out of LDS, the whole wave
//                        performs every back-reference copy.
//   k_inverse_*          the inverse transforms in reverse order of appearance (vp8l.h:343-344,
//                        transform.h:59-275): per-pixel parallel except the predictor, which runs
//                        as a skewed wavefront (row y trails row y-1 by two pixels).
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <vector>

#include "vp8l_amd.h"
#include "vp8l_dec_core.h"
#include "vp8l_host.h"
#include "vp8l_kernels.h"

namespace vp8l {

struct DecImage {  // host-written
  const uint8_t *data;
  uint64_t len;
  uint32_t w, h;
  uint32_t *out;
  uint32_t *aux;          // tile images / palette / entropy image
  uint64_t aux_cap;
  uint32_t *scratch;      // copy of the bundled indices for the colour-indexing inverse
  GroupTables *groups;    // tables for streams with more than one Huffman group
  uint32_t *cache;        // colour cache, 2048 entries
  uint16_t *sorted;       // (group_cap + 1) * kSortedPerGroup: symbols in code order, read for codes longer than the lookup table
  uint32_t group_cap;
  uint32_t pad;
  struct ParImage *par;   // workspace of the segment kernels (one large stream decoded by the whole device), or null
};

struct DecResult {  // device-written
  int32_t status;
  uint32_t need_groups;
  uint32_t nxf;
  uint32_t pad;
  uint32_t standard;  // the stream's dialect (the predictor's rule for the last column depends on it)
  uint32_t pad2;
  XformRec xf[4];
};

constexpr uint32_t kRingPixels = 2048;   // recent output kept in LDS for back references
constexpr uint32_t kRingNear = 1920;     // references up to this distance read the ring, farther ones HBM
constexpr uint32_t kStreamWindow = 2048;  // bytes of the stream kept in LDS

// ---------------------------------------------------------------------------------------------
// Pixels of a coded image with one Huffman group and no colour cache, decoded by the whole wave.
// A single reader is bound by the chain symbol -> code length -> position of the next symbol: about
// a thousand cycles per literal pixel for one wave.  Here lane o decodes the literal that WOULD start
// at bit o of a 128-bit buffer (four chained table lookups, all 64 lanes at once); the true starts
// are then found by hopping from literal to literal with lane reads of the lengths, and every lane
// that holds a true start stores its pixel.  One round resolves every literal inside 64 bits.
// Anything else (a back reference, a code longer than the lookup table) is decoded uniformly by
// the wave.  Everything that steers the loop is wave-uniform (scalar).
// Output goes to an LDS ring and from there to HBM in aligned blocks of 64 pixels; back references
// are copied by the wave, from the ring when they are near.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }
__device__ __forceinline__ uint32_t lane_read(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

struct WaveBits {  // all members wave-uniform
  uint64_t lo, hi;  // the next `cnt` bits of the stream, bit 0 of lo first; zero above cnt
  uint32_t cnt;
  uint32_t widx;    // next word of the window
  uint64_t win_lo;  // stream offset of the window
};

template <class WinPtr>
__device__ __forceinline__ void window_load(WinPtr win, const DecImage &im, uint64_t lo, uint32_t lane) {
  // the stream is 4-byte aligned (checked by the host); bytes past its end read as zero
  const uint32_t *src = reinterpret_cast<const uint32_t *>(im.data);
  for (uint32_t i = lane; i < kStreamWindow / 4; i += 64) {
    const uint64_t pos = lo + 4ull * i;
    uint32_t v = 0;
    if (pos + 4 <= im.len) v = src[pos >> 2];
    else for (uint32_t k = 0; k < 4; k++) if (pos + k < im.len) v |= (uint32_t)im.data[pos + k] << (8 * k);
    win[i] = v;
  }
}

template <class WinPtr>
__device__ __forceinline__ void wb_refill(WaveBits &b, WinPtr win, const DecImage &im, uint32_t lane) {  // more than 96 bits
  while (b.cnt <= 96) {
    if (b.widx == kStreamWindow / 4) {
      b.win_lo += kStreamWindow;
      __syncthreads();
      window_load(win, im, b.win_lo, lane);
      __syncthreads();
      b.widx = 0;
    }
    const uint64_t w = uni(win[b.widx]);
    if (b.cnt < 64) {
      b.lo |= w << b.cnt;
      if (b.cnt > 32) b.hi |= w >> (64 - b.cnt);
    } else {
      b.hi |= w << (b.cnt - 64);
    }
    b.cnt += 32;
    b.widx++;
  }
}
__device__ __forceinline__ void wb_skip(WaveBits &b, uint32_t n) {  // n <= cnt
  if (n >= 64) { b.lo = b.hi >> (n - 64); b.hi = 0; }
  else if (n) { b.lo = (b.lo >> n) | (b.hi << (64 - n)); b.hi >>= n; }
  b.cnt -= n;
}
__device__ __forceinline__ uint32_t wb_get(WaveBits &b, uint32_t n) {  // n <= 32 and available
  const uint32_t v = (uint32_t)(b.lo & ((1ull << n) - 1ull));
  wb_skip(b, n);
  return v;
}

// One symbol, uniformly (codes are at most 32 bits long; at least that many are buffered).  A code
// longer than the lookup table is found by all lengths at once: lane l tests whether the first l
// bits are a canonical code of length l; the shortest hit is the code (prefix-free).
template <class Table, class Sorted>
__device__ __forceinline__ uint32_t wb_sym(WaveBits &b, const Table &t, Sorted sorted, uint32_t lane) {
  const uint32_t bits = (uint32_t)b.lo;
  const uint32_t e = uni(t.lut[bits & ((1u << kLutBits) - 1u)]);
  if (e != kLongCode) { wb_skip(b, e & 15u); return e >> 4; }
  const uint32_t maxlen = uni(t.maxlen);
  uint32_t d = 0;
  bool hit = false;
  if (lane >= 1 && lane <= maxlen) {
    d = (__brev(bits) >> (32 - lane)) - t.first[lane];
    hit = d < t.count[lane];
  }
  const uint64_t hits = __ballot(hit);
  if (hits == 0) { wb_skip(b, maxlen); return 0; }  // unreachable for a complete code
  const uint32_t len = (uint32_t)__builtin_ctzll(hits);
  uint32_t sym = 0;
  if (hit) sym = sorted[t.sorted_off + t.index[lane] + d];
  wb_skip(b, len);
  return lane_read(sym, len);
}

// A function of its own: inlined into the kernel's state-machine loop it is compiled far worse.
// Its arguments point into LDS and are typed so (ds_read / ds_write instead of flat accesses).
#define VP8L_LDS __attribute__((address_space(3)))
__device__ __noinline__ void wave_decode_pixels(VP8L_LDS DecState *st_, const VP8L_LDS GroupTables *g_, VP8L_LDS uint32_t *ring,
                                                VP8L_LDS uint32_t *win, const DecImage &im, uint32_t lane) {
  VP8L_LDS DecState &st = *st_;
  const VP8L_LDS GroupTables &g = *g_;
  WaveBits b;
  b.lo = uni64(st.buf);
  b.hi = 0;
  b.cnt = uni(st.cnt);
  {
    const uint64_t bytepos = uni64(st.bytepos), wlo = uni64(st.win_lo), whi = uni64(st.win_hi);
    if (bytepos >= wlo && bytepos + 4 <= whi && whi == wlo + kStreamWindow) {
      b.win_lo = wlo;
      b.widx = (uint32_t)(bytepos - wlo) >> 2;
    } else {
      b.win_lo = bytepos;
      b.widx = 0;
      __syncthreads();
      window_load(win, im, bytepos, lane);
      __syncthreads();
    }
  }
  const uint32_t total = uni((uint32_t)st.total), xsize = uni(st.xsize | (st.standard << 31));  // bit 31: the dialect, for the distance codes
  uint32_t *dst = st.dst;
  const VP8L_LDS uint16_t *lutG = g.t[kGreen].lut, *lutR = g.t[kRed].lut, *lutB = g.t[kBlue].lut, *lutA = g.t[kAlpha].lut;
  uint32_t fill = 0, flushed = 0;  // flushed: multiple of 64, everything before it is in HBM
  int32_t status = kOk;
  while (fill < total) {
    wb_refill(b, win, im, lane);
    // ---- every lane: the literal that would start at bit `lane`
    uint64_t w = b.lo >> lane;
    if (lane) w |= b.hi << (64 - lane);
    const uint32_t eG = lutG[(uint32_t)w & ((1u << kLutBits) - 1u)];
    uint32_t n = eG & 15u;
    const uint32_t eR = lutR[(uint32_t)(w >> n) & ((1u << kLutBits) - 1u)];
    n += eR & 15u;
    const uint32_t eB = lutB[(uint32_t)(w >> n) & ((1u << kLutBits) - 1u)];
    n += eB & 15u;
    const uint32_t eA = lutA[(uint32_t)(w >> n) & ((1u << kLutBits) - 1u)];
    n += eA & 15u;
    const bool lit = eG < (256u << 4) && eR != kLongCode && eB != kLongCode && eA != kLongCode && n != 0;
    const uint32_t hop = lit ? n : 0u;  // 0: not a literal this path handles
    const uint32_t px = ((eA >> 4) << 24) | ((eR >> 4) << 16) | ((eG >> 4) << 8) | (eB >> 4);
    // ---- the true starts: hop from literal to literal
    uint64_t starts = 0;
    uint32_t o = 0, found = 0;
    const uint32_t room = total - fill;
    while (o < 64 && found < room) {
      const uint32_t h = lane_read(hop, o);
      if (h == 0 || o + h > b.cnt) break;  // not a plain literal, or one that runs past the buffered bits
      starts |= 1ull << o;
      o += h;
      found++;
    }
    if (found) {
      const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts, 0u));
      if ((starts >> lane) & 1ull) ring[(fill + below) & (kRingPixels - 1)] = px;
      wb_skip(b, o);
      fill += found;
      while (fill - flushed >= 64) {
        dst[flushed + lane] = ring[(flushed + lane) & (kRingPixels - 1)];
        flushed += 64;
      }
      continue;
    }
    // ---- the token at bit 0 is something else: decode it uniformly
    const uint32_t S = wb_sym(b, g.t[kGreen], g.sorted, lane);
    if (S < 256) {
      const uint32_t r2 = wb_sym(b, g.t[kRed], g.sorted, lane);
      wb_refill(b, win, im, lane);
      const uint32_t b2 = wb_sym(b, g.t[kBlue], g.sorted, lane);
      const uint32_t a2 = wb_sym(b, g.t[kAlpha], g.sorted, lane);
      if (lane == 0) ring[fill & (kRingPixels - 1)] = (a2 << 24) | (r2 << 16) | (S << 8) | b2;
      fill++;
    } else if (S < 280) {
      uint32_t base, nb;
      prefix_decode_base(S - 256, base, nb);
      const uint32_t length = base + wb_get(b, nb);
      prefix_decode_base(wb_sym(b, g.t[kDist], g.sorted, lane), base, nb);  // 32 + 10 + 32 bits at most so far
      wb_refill(b, win, im, lane);
      const uint32_t dcode = base + wb_get(b, nb);
      const uint32_t dist = code_to_dist(xsize & 0x7fffffffu, dcode, (xsize >> 31) != 0);
      // The reference copies without checking (vp8l.h:298-300); a reference before the start
      // or past the end of the image is reported instead of reading/writing out of bounds.
      if (dist > fill || length > total - fill) { status = kErrInvalidVp8l; break; }
      const bool near = dist <= kRingNear;
      if (!near) __threadfence_block();
      for (uint32_t c = 0; c < length; c += 64) {
        const uint32_t e = c + lane;
        if (e < length) {
          // element e of the copy equals element (e mod dist) of the source: right for overlapping references too
          const uint32_t s = fill - dist + (dist >= length ? e : e % dist);
          const uint32_t v = near ? ring[s & (kRingPixels - 1)] : dst[s];
          ring[(fill + e) & (kRingPixels - 1)] = v;
        }
        // completed blocks of 64 go out before the ring can wrap onto them
        const uint32_t upto = fill + (c + 64 < length ? c + 64 : length);
        while (upto - flushed >= 64) {
          dst[flushed + lane] = ring[(flushed + lane) & (kRingPixels - 1)];
          flushed += 64;
        }
      }
      fill += length;
      continue;
    } else {
      status = kErrInvalidVp8l;  // no colour cache: no such symbol in a 280-symbol alphabet
      break;
    }
    if (fill - flushed >= 64) {
      dst[flushed + lane] = ring[(flushed + lane) & (kRingPixels - 1)];
      flushed += 64;
    }
  }
  if (status == kOk && flushed + lane < fill) dst[flushed + lane] = ring[(flushed + lane) & (kRingPixels - 1)];
  __threadfence_block();
  // hand the reader back as the machine keeps it: at most 64 buffered bits, whole words un-read
  uint32_t unread = 0;
  while (b.cnt > 64) { b.cnt -= 32; unread++; }
  const bool in_window = b.widx >= unread;
  if (lane == 0) {
    st.fill = fill;
    st.buf = b.cnt == 64 ? b.lo : b.lo & ((1ull << b.cnt) - 1ull);
    st.cnt = b.cnt;
    st.bytepos = b.win_lo + 4ull * b.widx - 4ull * unread;
    st.win_lo = in_window ? b.win_lo : 0;
    st.win_hi = in_window ? b.win_lo + kStreamWindow : 0;
    st.win_final = in_window && b.win_lo + kStreamWindow >= im.len;
    if (status == kOk) st.phase = st.after;
    else { st.status = status; st.phase = kPhDone; }
  }
}

__global__ __launch_bounds__(64) void k_decode_entropy(const DecImage *__restrict__ imgs, DecResult *__restrict__ res) {
  __shared__ DecState st;
  __shared__ GroupTables g0;
  // the ring is used while pixels are decoded, the table scratch while a code is read: never at the same time
  // (a coded image's codes are read before its first pixel, and every coded image starts a fresh ring)
  __shared__ union RingOrScratch {
    uint32_t ring[kRingPixels];
    DecTableScratch scratch;
  } ros;
  static_assert(sizeof(DecTableScratch) <= sizeof(uint32_t) * kRingPixels, "the scratch fits under the ring");
  uint32_t *ring = ros.ring;
  __shared__ uint32_t win[kStreamWindow / 4];
  const uint32_t img = blockIdx.x, lane = threadIdx.x;
  const DecImage im = imgs[img];
  DecAccel acc;
  acc.win = win;
  acc.win_bytes = kStreamWindow;
  acc.ring = ring;
  acc.ring_mask = kRingPixels - 1;
  acc.g0 = &g0;
  acc.driver_pixels = true;
  acc.on = true;
  if (lane == 0) {
    dec_init(st, im.data, im.len, im.out, im.aux, im.aux_cap, &g0, im.groups, im.group_cap, im.cache, &ros.scratch, im.sorted);
    st.w = im.w;
    st.h = im.h;
  }
  __syncthreads();
  for (;;) {
    uint32_t ev = kEvDone;
    if (lane == 0) ev = dec_step(st, acc);
    ev = (uint32_t)__shfl((int)ev, 0, 64);
    if (ev == kEvDone) break;
    __syncthreads();  // lane 0's request (LDS) and pixel stores (HBM) before the wave acts on them
    if (ev == kEvPixels) {
      wave_decode_pixels((VP8L_LDS DecState *)&st, (const VP8L_LDS GroupTables *)&g0, (VP8L_LDS uint32_t *)ring,
                         (VP8L_LDS uint32_t *)win, im, lane);
    } else if (ev == kEvRefill) {
      const uint64_t lo = st.bytepos;
      window_load(win, im, lo, lane);
      if (lane == 0) { st.win_lo = lo; st.win_hi = lo + kStreamWindow; st.win_final = lo + kStreamWindow >= im.len; }
    } else {
      __threadfence_block();
      uint32_t *dst = st.dst;
      const uint64_t pos = st.copy_pos;
      const uint32_t dist = st.copy_dist, len = st.copy_len;
      // pixel i of the copy equals pixel (i mod dist) of the source window: correct for overlapping
      // references too, and every source is already final (in HBM: lane 0 stores every pixel there too)
      for (uint32_t i = lane; i < len; i += 64) {
        const uint32_t v = dst[pos - dist + (dist >= len ? i : i % dist)];
        dst[pos + i] = v;
        if (i + kRingPixels >= len) ring[(uint32_t)(pos + i) & (kRingPixels - 1)] = v;
      }
      __threadfence_block();
    }
    __syncthreads();
  }
  if (lane == 0) {
    DecResult &r = res[img];
    r.status = st.status;
    r.need_groups = st.need_groups;
    r.nxf = st.nxf;
    r.standard = st.standard;
    for (uint32_t k = 0; k < 4; k++) r.xf[k] = st.xf[k];
  }
}

// ---------------------------------------------------------------------------------------------
// The same pixels decoded by a workgroup of 1024 threads: the default form.
// One wave resolves the literals inside 64 stream bits per round; here a round covers 1024 bits:
//   1. thread t decodes the token that WOULD start at bit offset t of the span -- a literal (four
//      symbols) or a back reference (length symbol + extra bits, distance symbol + extra bits), codes
//      longer than the lookup table included -- from a 64-bit piece of the stream window.  That gives
//      next[t] = t + bits of the token at t (a sink when the token leaves the span or is one of the
//      rare things decoded below).
//   2. the tokens that really start in the span are the chain 0, next[0], next[next[0]], ...  It is
//      found without a serial walk: jump tables next^(4^k) by pointer jumping (k < 4), then marks
//      spread from offset 0 with the widest jump first, so 128 tokens are marked in 4 steps.
//   3. a scan of the marked tokens' pixel counts gives every token its output position.  Literals go
//      straight into the LDS ring.  Every pixel of the round gets the position its value comes from:
//      itself (literal), an older pixel, or a pixel of this round that a reference makes; the last kind
//      is replaced by that pixel's own source until none is left (pointer jumping again: the
//      reference encoder codes a run as copies of 1, 2, 4, 8 ... pixels, each reading the one before).
//      Then every pixel is fetched from its source.  A round produces at most 2048 pixels.
//   4. complete blocks of 64 pixels go from the ring to HBM.
// What a round cannot start with (a token longer than 64 bits, a code of zero bits, a reference
// longer than 255) is decoded by wave 0 alone, uniformly, as in the one-wave form.
// With all 1024 threads busy the workgroup is bound by its CU's instruction issue (16 waves on 4
// SIMDs), not by latency: what counts is instructions per round.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kWgThreads = 1024;
constexpr uint32_t kWgWaves = kWgThreads / 64;
constexpr uint32_t kWgSpan = kWgThreads;      // bit offsets looked at per round, one per thread
constexpr uint32_t kWgLevels = 4;             // jump tables next^(4^k), k < 4: at most 2 * 64 = 128 tokens per round
constexpr uint32_t kWgRing = 4096;            // pixels kept in LDS
constexpr uint32_t kWgNear = 2048;            // pixels a round may produce = how far back the ring is read
constexpr uint32_t kWgWindow = 4096;          // stream bytes kept in LDS
constexpr uint32_t kWgWindowWords = kWgWindow / 4 + 4;  // + what the last offsets of a span read past it
constexpr uint32_t kWgMaxLen = 255;           // longer references take the uniform path
constexpr uint32_t kWgOwnCopy = 8;            // references up to this length are copied by their own thread
constexpr uint32_t kWgSorted = 280 + 3 * 256 + 40;  // code-order symbol lists of the five codes without a colour cache
constexpr uint32_t kWgLongLens = kSaneCodeLen - kLutBits;  // code lengths above the lookup table: 11..32
constexpr uint32_t kWgLongBatch = 6;          // lengths compared at once
constexpr uint32_t kWgLongSlots = 24;         // kWgLongLens rounded up to whole batches
constexpr uint32_t kWgJumpSteps = 12;         // source chains are at most 2048 long: 11 halvings
static_assert(kWgSpan + 128 <= kWgWindow * 8 / 2, "a span fits the window with room to move");
static_assert(kWgLongSlots % kWgLongBatch == 0 && kWgLongSlots >= kWgLongLens, "long-code slots");

// Colour cache (vp8l.h:286-324) in the workgroup form.
struct WgCacheShared {
  uint32_t last[1u << kMaxCacheBits];  // per cache slot: 1 + the last pixel of the round with that key (0: none); all zero between uses
  uint16_t key[kWgNear];               // cache key of every pixel of the round
};

struct WgShared {
  uint32_t srcp[kWgNear];                     // per pixel of the round: the position its value comes from (| kFinal)
  uint32_t changed[kWgJumpSteps];
  uint16_t ja[kWgSpan + 2], jb[kWgSpan + 2];  // jump tables (ping-pong); entry kWgSpan is the sink
  uint8_t mark[kWgSpan + 4];                  // 0: not on the chain, else the step that marked it + 1
  uint32_t wsum[kWgWaves];
  // codes longer than the lookup table, per code and length l = kLutBits + 1 + slot:
  //   lim = the largest 32-bit left-justified stream value whose code is at most l bits long,
  //   base + (code as a number) = index of the symbol in `sorted`
  uint32_t lim[kNumCodes][kWgLongSlots];
  uint32_t base[kNumCodes][kWgLongSlots];
  uint32_t defer_off, defer_px, end_off, end_px, cut_off;
  int32_t status;
  uint32_t s_bits, s_fill, s_flushed;         // what the uniform path did
  uint16_t sorted[kWgSorted];                 // symbols in code order
};

__device__ __forceinline__ void wg_window_load(VP8L_LDS uint32_t *win, const DecImage &im, uint64_t lo, uint32_t tid,
                                               uint32_t words = kWgWindowWords, uint32_t threads = kWgThreads) {
  const uint32_t *src = reinterpret_cast<const uint32_t *>(im.data);  // 4-byte aligned; bytes past the end read as zero
  for (uint32_t i = tid; i < words; i += threads) {
    const uint64_t pos = lo + 4ull * i;
    uint32_t v = 0;
    if (pos + 4 <= im.len) v = src[pos >> 2];
    else for (uint32_t k = 0; k < 4; k++) if (pos + k < im.len) v |= (uint32_t)im.data[pos + k] << (8 * k);
    win[i] = v;
  }
}

// One symbol at bit n of w (64 stream bits) by one thread; false when w does not hold all of it.
// far: the group's whole symbol list in HBM, for alphabets that the LDS copy does not hold (a green code with colour-cache
// symbols); null: the LDS copy.
template <class Sh>
__device__ __forceinline__ bool wg_sym(uint64_t w, uint32_t &n, const VP8L_LDS HuffTable &t, const VP8L_LDS Sh &sh, uint32_t code,
                                       uint32_t &sym, const uint16_t *far = nullptr) {
  if (n > 64 - kLutBits) return false;
  const uint32_t bits = (uint32_t)(w >> n);
  const uint32_t e = t.lut[bits & ((1u << kLutBits) - 1u)];
  if (e != kLongCode) { n += e & 15u; sym = e >> 4; return true; }
  // A canonical code read as a left-justified number grows with its length, so the length is 11 + the number
  // of lengths whose largest value is still below the stream's: a few independent reads instead of a walk.
  asm volatile("" ::: "memory");  // a caller's token loop must not load the limits of every code ahead of the loop, into a hundred registers
  const uint32_t v = __brev(bits);
  uint32_t slot = 0;
  for (uint32_t s0 = 0; s0 < kWgLongSlots; s0 += kWgLongBatch) {
    uint32_t above = 0;
#pragma unroll
    for (uint32_t k = 0; k < kWgLongBatch; k++) above += v > sh.lim[code][s0 + k] ? 1u : 0u;
    slot = s0 + above;
    if (above < kWgLongBatch) break;
  }
  const uint32_t l = kLutBits + 1 + slot;
  if (slot >= kWgLongLens || n + l > 64) return false;
  const uint32_t at = sh.base[code][slot] + (v >> (32 - l));
  sym = far ? far[at] : sh.sorted[at];
  n += l;
  return true;
}

// What wg_sym needs for codes longer than the lookup table (a barrier must follow before the first use).
template <class Sh>
__device__ __forceinline__ void wg_prepare_long(VP8L_LDS Sh &sh, const VP8L_LDS GroupTables &g, uint32_t tid, uint32_t threads = kWgThreads) {
  for (uint32_t i = tid; i < kWgSorted; i += threads) sh.sorted[i] = g.sorted[i];
  if (tid < kNumCodes * kWgLongSlots) {
    const uint32_t c = tid / kWgLongSlots, slot = tid % kWgLongSlots, l = kLutBits + 1 + slot;
    const VP8L_LDS HuffTable &t = g.t[c];
    uint32_t lim = 0xffffffffu, base = 0;
    if (l <= t.maxlen) {
      lim = (uint32_t)(((uint64_t)(t.first[l] + t.count[l]) << (32 - l)) - 1ull);  // not zero: some code is at most 9 bits long
      base = t.sorted_off + t.index[l] - t.first[l];
    }
    sh.lim[c][slot] = lim;
    sh.base[c][slot] = base;
  }
}

// The token that starts at bit 0 of w: hop = its bits (0: not decodable from w alone), cnt = pixels it
// makes, val = the pixel (literal) or the distance (ref).
// far (see wg_sym): set for images with a colour cache; a cache symbol then is a token of one pixel: cnt = 1, ref = true,
// val = kWgCacheTok | index (the caller resolves it).
constexpr uint32_t kWgCacheTok = 0x80000000u;  // no distance reaches this bit
// alpha_single: the alpha code's only symbol when it has just one (an opaque image: the code takes no bits, and the lookup
// that would say so is a fourth dependent LDS read per literal), else 0xffff.
template <bool kValue = true, class Sh>
__device__ __forceinline__ void wg_token(uint64_t w, const VP8L_LDS GroupTables &g, const VP8L_LDS Sh &sh, uint32_t xsize,
                                         uint32_t &hop, uint32_t &cnt, uint32_t &val, bool &ref, const uint16_t *far = nullptr,
                                         uint32_t alpha_single = 0xffffu) {
  hop = 0; cnt = 0; val = 0; ref = false;
  uint32_t n = 0, S = 0;
  if (!wg_sym(w, n, g.t[kGreen], sh, kGreen, S, far)) return;
  if (far && S >= 280) {
    val = kWgCacheTok | (S - 280);
    cnt = 1;
    ref = true;
    hop = n;
    return;
  }
  if (S < 256) {
    uint32_t r = 0, b = 0, a = 0;
    if (!wg_sym(w, n, g.t[kRed], sh, kRed, r, far) || !wg_sym(w, n, g.t[kBlue], sh, kBlue, b, far)) return;
    if (alpha_single != 0xffffu) a = alpha_single;
    else if (!wg_sym(w, n, g.t[kAlpha], sh, kAlpha, a, far)) return;
    if (kValue) val = (a << 24) | (r << 16) | (S << 8) | b;
    cnt = 1;
    hop = n;
  } else if (S < 280) {
    uint32_t base, nb;
    prefix_decode_base(S - 256, base, nb);
    if (n + nb > 64) return;
    const uint32_t length = base + (nb ? (uint32_t)(w >> n) & ((1u << nb) - 1u) : 0u);
    n += nb;
    uint32_t ds = 0;
    if (!wg_sym(w, n, g.t[kDist], sh, kDist, ds, far)) return;
    prefix_decode_base(ds, base, nb);
    if (n + nb > 64) return;
    const uint32_t dcode = base + (nb ? (uint32_t)(w >> n) & ((1u << nb) - 1u) : 0u);
    n += nb;
    if (length > kWgMaxLen) return;
    if (kValue) val = code_to_dist(xsize & 0x7fffffffu, dcode, (xsize >> 31) != 0);  // else only the token's bits and pixels are wanted
    cnt = length;
    ref = true;
    hop = n;
  }
}

// Several Huffman groups (vp8l.h:250-256): the group is that of the tile in which a token STARTS, so a round runs with the tables of
// the group at the current pixel and takes only the tokens that start before the next tile with another group; when the group
// changes, that group's tables are copied from HBM into the same LDS place (a few words per thread).
//
// A colour cache (vp8l.h:286-324): a cache symbol stands for the pixel last stored under its index, and every pixel is stored
// under the key of its value.  The key of a cache token's pixel is its index (whatever was stored there has that key), the key
// of a copied pixel is that of its source, so after the sources are resolved every pixel of the round has a key although the
// cache tokens' values are still unknown.  A cache token then takes the round's last earlier pixel with its key as its source
// (a wave looks back over the keys 64 at a time), or the slot's value from before the round, and a second round of source
// jumping gives every pixel its value.  The slots are updated with the round's last pixel of every key.  One case breaks the
// "key = index" rule: a slot that was never written holds 0, whose key is 0; a token that asks for such a slot (no valid
// encoder does) ends the round before it and is decoded by wave 0 on its own.
__device__ __noinline__ void wg_decode_pixels(VP8L_LDS DecState *st_, VP8L_LDS GroupTables *g_, VP8L_LDS uint32_t *ring,
                                              VP8L_LDS uint32_t *win, VP8L_LDS WgShared *sh_, VP8L_LDS WgCacheShared *cs_,
                                              const DecImage &im, uint32_t tid) {
  VP8L_LDS DecState &st = *st_;
  VP8L_LDS GroupTables &g = *g_;
  VP8L_LDS WgShared &sh = *sh_;
  VP8L_LDS WgCacheShared &cs = *cs_;
  const uint32_t cb = st.cache_bits;
  uint32_t *cache = st.cache;  // HBM (L2): 1 << cb slots, zeroed when the image began
  constexpr uint32_t kPend = 0x40000000u, kRoot = 0x20000000u;  // in srcp: a cache token's pixel (| index) / a copy of one (| its place in the round)
  if (cb) for (uint32_t i = tid; i < (1u << kMaxCacheBits); i += kWgThreads) cs.last[i] = 0;  // a barrier follows before the first use
  const bool multi = st.ngroups > 1;
  const uint32_t *meta = st.meta;
  const uint32_t mbits = st.meta_bits, mxs = st.meta_xsize, width = st.xsize, rows = st.ysize;
  const GroupTables *all_groups = st.groups;   // HBM, when there are several
  uint32_t gcur = 0xffffffffu;                 // the group whose tables are in LDS
  const uint32_t lane = tid & 63, wave = tid >> 6;
  constexpr uint32_t kMask = kWgRing - 1;
  uint64_t P = st.bytepos * 8 - st.cnt;  // position in the stream, in bits
  const uint32_t total = (uint32_t)st.total, xsize = st.xsize | (st.standard << 31);  // bit 31: the dialect, for the distance codes
  uint32_t *dst = st.dst;
  if (!multi) wg_prepare_long(sh, g, tid);
  uint32_t fill = 0, flushed = 0;  // flushed: multiple of 64, everything before it is in HBM
  uint64_t win_bit0 = 0;
  bool have_window = false;
  int32_t status = kOk;
  if (tid == 0) {
    sh.status = kOk;
    sh.ja[kWgSpan] = (uint16_t)kWgSpan;
    sh.jb[kWgSpan] = (uint16_t)kWgSpan;
  }
  while (fill < total) {
    uint32_t span_px = 0xffffffffu;  // tokens that start this many pixels or more after `fill` wait for another round
    if (multi) {
      // the run of tiles from the current pixel on that share its group (up to eight tiles ahead, all looked up at once)
      const uint32_t y = fill / width, x = fill - y * width;
      const uint32_t tiles_x = (width + (1u << mbits) - 1) >> mbits;  // tiles that hold coded pixels (the entropy image may be wider)
      uint32_t gi[9], px_end[9];
      uint32_t tx = x >> mbits, ty = y, acc_px = 0, cx = x;
#pragma unroll
      for (uint32_t k = 0; k < 9; k++) {
        gi[k] = meta[((ty < rows ? ty : rows - 1) >> mbits) * mxs + tx];  // past the last row: only looked at, never used
        const uint32_t tile_end = ((tx + 1) << mbits) < width ? ((tx + 1) << mbits) : width;
        acc_px += tile_end - cx;
        px_end[k] = acc_px;
        cx = tile_end;
        tx++;
        if (tx >= tiles_x) { tx = 0; cx = 0; ty++; }
      }
      span_px = px_end[0];
#pragma unroll
      for (uint32_t k = 1; k < 9; k++) {
        if (gi[k] != gi[0]) break;
        span_px = px_end[k];
      }
      if (gi[0] != gcur) {
        __syncthreads();  // nobody still reads the tables of the group before
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&all_groups[gi[0]]);
        for (uint32_t i = tid; i < sizeof(GroupTables) / 4; i += kWgThreads) reinterpret_cast<VP8L_LDS uint32_t *>(&g)[i] = src[i];
        __syncthreads();
        wg_prepare_long(sh, g, tid);
        gcur = gi[0];
      }
    }
    if (!have_window || P < win_bit0 || P + kWgSpan + 128 > win_bit0 + kWgWindow * 8) {
      __syncthreads();
      const uint64_t lo = (P >> 3) & ~3ull;
      wg_window_load(win, im, lo, tid);
      win_bit0 = lo * 8;
      have_window = true;
    }
    __syncthreads();
    // ---- 1. the token that would start at offset tid
    uint32_t hop, cnt, val;
    bool ref;
    {
      const uint32_t a = (uint32_t)(P - win_bit0) + tid, idx = a >> 5, sft = a & 31u;
      const uint64_t x = (uint64_t)win[idx] | ((uint64_t)win[idx + 1] << 32);
      const uint64_t w = sft ? (x >> sft) | ((uint64_t)win[idx + 2] << (64 - sft)) : x;
      wg_token(w, g, sh, xsize, hop, cnt, val, ref, cb ? g.sorted : (const uint16_t *)nullptr);
    }
    const bool cch = ref && (val & kWgCacheTok);  // a colour-cache token (only with cb != 0)
    // ---- 2. jump tables next^(4^k) for k < 4, then the marks
    // Every step costs an LDS round trip and a barrier whatever it does, so a step quadruples the jump:
    // s1/s2/s3[k] are 1, 2 and 3 jumps of 4^k tokens from this offset, the fourth is the next table's entry.
    uint32_t s1[kWgLevels], s2[kWgLevels], s3[kWgLevels];
    s1[0] = hop && tid + hop < kWgSpan ? tid + hop : kWgSpan;
    sh.ja[tid] = (uint16_t)s1[0];
    sh.mark[tid] = tid == 0 ? (uint8_t)(kWgLevels + 1) : (uint8_t)0;
    VP8L_LDS uint16_t *cur = sh.ja, *oth = sh.jb;
#pragma unroll
    for (uint32_t k = 0; k < kWgLevels; k++) {
      __syncthreads();
      s2[k] = cur[s1[k]];
      s3[k] = cur[s2[k]];
      if (k + 1 < kWgLevels) {
        s1[k + 1] = cur[s3[k]];
        oth[tid] = (uint16_t)s1[k + 1];
        VP8L_LDS uint16_t *t16 = cur; cur = oth; oth = t16;
      }
    }
    // chain positions 0..127 = i3 * 64 + i2 * 16 + i1 * 4 + i0 with i3 < 2 and the others < 4: offset 0 marks its
    // 64-token jump, then every marked offset marks its three 16-token jumps, and so on down
    const uint32_t beyond = s2[kWgLevels - 1];  // thread 0: where the 129th token starts, if the chain is that long
#pragma unroll
    for (int k = (int)kWgLevels - 1; k >= 0; k--) {
      if (k != (int)kWgLevels - 1) __syncthreads();  // the last table step ended with no write: its barrier is this step's
      // only marks older than this step spread: a mark set during the step (value k + 1) does not
      if (sh.mark[tid] > (uint32_t)k + 1) {
        sh.mark[s1[k]] = (uint8_t)(k + 1);
        if (k != (int)kWgLevels - 1) {
          sh.mark[s2[k]] = (uint8_t)(k + 1);
          sh.mark[s3[k]] = (uint8_t)(k + 1);
        }
      }
    }
    __syncthreads();
    const bool on = sh.mark[tid] != 0;
    // ---- 3. output positions
    const uint32_t c = on && hop ? cnt : 0u;
    const uint32_t incl = wave_incl_scan_u32(c, (int)lane);
    if (lane == 63) sh.wsum[wave] = incl;
    if (tid == 0) { sh.defer_off = 0xffffffffu; sh.cut_off = 0xffffffffu; sh.end_off = 0; sh.end_px = 0; }
    __syncthreads();
    // the waves' sums, scanned by every wave for itself
    const uint32_t wsums = wave_incl_scan_u32(lane < kWgWaves ? sh.wsum[lane] : 0u, (int)lane);
    const uint32_t before = wave ? lane_read(wsums, wave - 1) : 0u, all = lane_read(wsums, kWgWaves - 1);
    const uint32_t ex = before + incl - c;
    const uint32_t left = total - fill, room = left < kWgNear ? left : kWgNear;
    if (tid == 0 && beyond != kWgSpan) { sh.end_off = beyond; sh.end_px = all; }
    // Literals go to the ring.  Every pixel of the round gets a source: itself (literal: final), an older pixel
    // (final), or a pixel of this round that a reference makes (to be followed further).
    constexpr uint32_t kFinal = 0x80000000u;
    bool expand = false;  // a reference of this round whose pixels get their sources
    const uint32_t pos = fill + ex;
    if (on) {
      if (hop == 0) {
        sh.end_off = tid;  // the chain ends at something this round cannot decode
        sh.end_px = ex;
      } else {
        if (s1[0] == kWgSpan) { sh.end_off = tid + hop; sh.end_px = ex + cnt; }
        if (ex + cnt > room || ex >= span_px) {
          // the first token that does not fit, or that starts in a tile with other codes: the next round starts with it
          // (tokens lie in stream order, so the smallest offset is the first; offset and pixels in one word)
          __hip_atomic_fetch_min(&sh.defer_off, (tid << 16) | (ex < 0xffffu ? ex : 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (ex + cnt > room && ex <= room && ex < span_px && room == left && ex < room) sh.status = kErrInvalidVp8l;  // a reference that runs past the end of the image
        } else if (cch) {
          sh.srcp[ex] = kPend | (val & 0xffffu);
        } else if (!ref) {
          ring[pos & kMask] = val;
          sh.srcp[ex] = pos | kFinal;
        } else if (val > pos) {
          // The reference copies without checking (vp8l.h:298-300); a reference before the start of the image is reported.
          sh.status = kErrInvalidVp8l;
        } else {
          expand = true;
        }
      }
    }
    if (expand && cnt <= kWgOwnCopy) {
      if (val >= cnt) {
        for (uint32_t e = 0; e < cnt; e++) {
          const uint32_t s = pos - val + e;
          sh.srcp[ex + e] = s < fill ? s | kFinal : s;
        }
      } else {  // overlapping: the first `val` pixels repeat
        for (uint32_t e = 0; e < cnt; e++) {
          const uint32_t s = pos - val + e % val;
          sh.srcp[ex + e] = s < fill ? s | kFinal : s;
        }
      }
    }
    {
      uint64_t m = __ballot(expand && cnt > kWgOwnCopy);  // the longer ones: the wave writes 64 sources at a time
      while (m) {
        const uint32_t l = (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        const uint32_t p2 = lane_read(pos, l), n = lane_read(cnt, l), dist = lane_read(val, l), x2 = lane_read(ex, l);
        for (uint32_t c2 = 0; c2 < n; c2 += 64) {
          const uint32_t e = c2 + lane;
          if (e < n) {
            uint32_t s = p2 - dist + e;
            if (dist < n) s = p2 - dist + e % dist;  // uniform
            sh.srcp[x2 + e] = s < fill ? s | kFinal : s;
          }
        }
      }
    }
    if (tid < kWgJumpSteps) sh.changed[tid] = 0;
    __syncthreads();
    status = sh.status;
    if (status != kOk) break;
    const bool deferred = sh.defer_off != 0xffffffffu;
    uint32_t adv = deferred ? sh.defer_off >> 16 : sh.end_off, produced = deferred ? sh.defer_off & 0xffffu : sh.end_px;
    // Sources that are made by a reference of this round are replaced by that pixel's source until every pixel
    // points at a final one: chains of references (a run coded as copies of 1, 2, 4, 8 ... pixels) take
    // log2(chain length) steps instead of one copy after the other.  A source read while its owner replaces
    // it is the old or the new value; both are sources of the same pixel value.
    // (With a colour cache a chain may also end at a cache token's pixel, kPend, whose value comes later: the pixels that
    // copy it remember where it is, kRoot.)
    auto jump_sources = [&]() {
      for (uint32_t it = 0; it < kWgJumpSteps; it++) {
        bool open = false;
#pragma unroll
        for (uint32_t k = 0; k < kWgNear / kWgThreads; k++) {
          const uint32_t q = k * kWgThreads + tid;
          if (q < produced) {
            const uint32_t v = sh.srcp[q];
            if (!(v & (kFinal | kPend | kRoot))) {
              uint32_t t = sh.srcp[v - fill];
              if (t & kPend) t = kRoot | (v - fill);
              sh.srcp[q] = t;
              open = open || !(t & (kFinal | kRoot));
            }
          }
        }
        if (open) sh.changed[it] = 1;
        __syncthreads();
        if (!sh.changed[it]) break;
      }
    };
    jump_sources();
#pragma unroll
    for (uint32_t k = 0; k < kWgNear / kWgThreads; k++) {
      const uint32_t q = k * kWgThreads + tid;
      if (q < produced) {
        const uint32_t sv = sh.srcp[q], s = sv & ~kFinal;
        if ((sv & kFinal) && s != fill + q) ring[(fill + q) & kMask] = s + kWgNear >= fill ? ring[s & kMask] : dst[s];
      }
    }
    __syncthreads();
    if (cb) {
      // ---- keys of all pixels of the round; pixels whose value is in the ring now point at themselves
      uint32_t mykey[kWgNear / kWgThreads];
#pragma unroll
      for (uint32_t k = 0; k < kWgNear / kWgThreads; k++) {
        const uint32_t q = k * kWgThreads + tid;
        mykey[k] = 0;
        if (q < produced) {
          const uint32_t sv = sh.srcp[q];
          if (sv & kFinal) mykey[k] = cache_key(ring[(fill + q) & kMask], cb);
          else if (sv & kPend) mykey[k] = sv & 0xffffu;
          else mykey[k] = sh.srcp[sv & 0xffffu] & 0xffffu;  // kRoot: the key of the cache token it copies
          cs.key[q] = (uint16_t)mykey[k];
        }
      }
      if (tid < kWgJumpSteps) sh.changed[tid] = 0;
      __syncthreads();
#pragma unroll
      for (uint32_t k = 0; k < kWgNear / kWgThreads; k++) {
        const uint32_t q = k * kWgThreads + tid;
        if (q < produced) {
          const uint32_t sv = sh.srcp[q];
          if (sv & kFinal) sh.srcp[q] = (fill + q) | kFinal;
          else if (sv & kRoot) sh.srcp[q] = fill + (sv & 0xffffu);  // an open source again: the cache token's pixel
        }
      }
      // ---- every cache token: the round's last earlier pixel with its key, else the slot as it was before the round
      const bool mine = on && hop && cch && ex < produced;
      uint32_t found = 0xffffffffu;
      for (uint64_t m = __ballot(mine); m; m &= m - 1) {
        const uint32_t l = (uint32_t)__builtin_ctzll(m);
        const uint32_t q0 = lane_read(ex, l), k0 = lane_read(val & 0xffffu, l);
        uint32_t f = 0xffffffffu;
        for (int base = (int)q0 - 1; base >= 0; base -= 64) {
          const int i = base - (int)lane;
          const uint64_t hit = __ballot(i >= 0 && cs.key[i] == k0);
          if (hit) { f = (uint32_t)base - (uint32_t)__builtin_ctzll(hit); break; }
        }
        if (lane == l) found = f;
      }
      if (mine) {
        const uint32_t idx = val & 0xffffu;
        if (found != 0xffffffffu) {
          sh.srcp[ex] = fill + found;
        } else {
          const uint32_t older = cache[idx];
          if (cache_key(older, cb) != idx) {
            // a slot nobody wrote: the pixel is 0 and does not have the key this round gave it; everything from here on waits
            __hip_atomic_fetch_min(&sh.cut_off, (tid << 16) | ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            ring[pos & kMask] = older;
            sh.srcp[ex] = pos | kFinal;
          }
        }
      }
      __syncthreads();
      if (sh.cut_off != 0xffffffffu) { adv = sh.cut_off >> 16; produced = sh.cut_off & 0xffffu; }
      jump_sources();
#pragma unroll
      for (uint32_t k = 0; k < kWgNear / kWgThreads; k++) {
        const uint32_t q = k * kWgThreads + tid;
        if (q < produced) {
          const uint32_t s = sh.srcp[q] & ~kFinal;
          if (s != fill + q) ring[(fill + q) & kMask] = ring[s & kMask];  // a pixel of this round, in the ring
        }
      }
      __syncthreads();
      // ---- the slots take the round's last pixel of every key
      uint32_t myval[kWgNear / kWgThreads];
#pragma unroll
      for (uint32_t k = 0; k < kWgNear / kWgThreads; k++) {
        const uint32_t q = k * kWgThreads + tid;
        if (q < produced) {
          myval[k] = ring[(fill + q) & kMask];
          mykey[k] = cache_key(myval[k], cb);
          __hip_atomic_fetch_max(&cs.last[mykey[k]], q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      __syncthreads();
#pragma unroll
      for (uint32_t k = 0; k < kWgNear / kWgThreads; k++) {
        const uint32_t q = k * kWgThreads + tid;
        if (q < produced && cs.last[mykey[k]] == q + 1) {
          cache[mykey[k]] = myval[k];
          cs.last[mykey[k]] = 0;
        }
      }
      __syncthreads();
    }
    fill += produced;
    P += adv;
    {
      const uint32_t upto = fill & ~63u;
      for (uint32_t p = flushed + tid; p < upto; p += kWgThreads) dst[p] = ring[p & kMask];
      flushed = upto;
    }
    if (adv == 0 && produced == 0) {
      // ---- the token at P is one a round cannot decode: wave 0 decodes it uniformly
      if (wave == 0) {
        __threadfence_block();
        const uint32_t a = (uint32_t)(P - win_bit0), idx = a >> 5, sft = a & 31u;
        const uint64_t x = (uint64_t)uni(win[idx]) | ((uint64_t)uni(win[idx + 1]) << 32);
        const uint64_t y = (uint64_t)uni(win[idx + 2]) | ((uint64_t)uni(win[idx + 3]) << 32);
        const uint64_t z = (uint64_t)uni(win[idx + 4]);
        WaveBits b;
        b.lo = sft ? (x >> sft) | (y << (64 - sft)) : x;
        b.hi = sft ? (y >> sft) | (z << (64 - sft)) : y;
        b.cnt = 128;
        b.widx = 0;
        b.win_lo = 0;
        const VP8L_LDS uint16_t *sorted = sh.sorted;
        uint32_t nfill = fill, nflushed = flushed;
        int32_t rc = kOk;
        // with a colour cache the symbol lists are longer than the LDS copy: the group's own list in HBM then
        auto usym = [&](const VP8L_LDS HuffTable &t) { return cb ? wb_sym(b, t, g.sorted, lane) : wb_sym(b, t, sorted, lane); };
        const uint32_t S = usym(g.t[kGreen]);
        if (S < 256 || (cb && S >= 280)) {
          uint32_t px;
          if (S < 256) {
            const uint32_t r2 = usym(g.t[kRed]);
            const uint32_t b2 = usym(g.t[kBlue]);
            const uint32_t a2 = usym(g.t[kAlpha]);
            px = (a2 << 24) | (r2 << 16) | (S << 8) | b2;
          } else {
            px = uni(cache[S - 280]);  // vp8l.h:315-324; an index is below 1 << cb: the alphabet ends there
          }
          if (lane == 0) {
            ring[fill & kMask] = px;
            if (cb) cache[cache_key(px, cb)] = px;
          }
          nfill = fill + 1;
        } else if (S < 280) {
          uint32_t base, nb;
          prefix_decode_base(S - 256, base, nb);
          const uint32_t length = base + wb_get(b, nb);
          prefix_decode_base(usym(g.t[kDist]), base, nb);
          const uint32_t dcode = base + wb_get(b, nb);
          const uint32_t dist = code_to_dist(xsize & 0x7fffffffu, dcode, (xsize >> 31) != 0);
          if (dist > fill || length > total - fill) {
            rc = kErrInvalidVp8l;
          } else {
            // in pieces of 1024 pixels, each flushed before the next, so the ring never wraps onto what is still read
            for (uint32_t c0 = 0; c0 < length; c0 += 1024) {
              const uint32_t piece = length - c0 < 1024 ? length - c0 : 1024, start = fill + c0;
              for (uint32_t c2 = 0; c2 < piece; c2 += 64) {
                const uint32_t e = c0 + c2 + lane;
                const bool in = c2 + lane < piece;
                uint32_t v = 0;
                if (in) {
                  const uint32_t s = fill - dist + (dist >= length ? e : e % dist);
                  v = s + kWgNear >= start ? ring[s & kMask] : dst[s];
                  ring[(fill + e) & kMask] = v;
                }
                if (cb) {
                  // the copied pixels enter the cache in order: of the 64 of this step the last one of every key stays
                  // (this wave's LDS operations complete in order; the chunks follow each other)
                  const uint32_t key = cache_key(v, cb);
                  if (in) __hip_atomic_fetch_max(&cs.last[key], lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  if (in && cs.last[key] == lane + 1) {
                    cache[key] = v;
                    cs.last[key] = 0;
                  }
                }
              }
              const uint32_t upto = (start + piece) & ~63u;
              while (nflushed < upto) {
                dst[nflushed + lane] = ring[(nflushed + lane) & kMask];
                nflushed += 64;
              }
              __threadfence_block();
            }
            nfill = fill + length;
          }
        } else {
          rc = kErrInvalidVp8l;  // without a colour cache there is no such symbol in a 280-symbol alphabet
        }
        if (lane == 0) {
          sh.s_bits = 128 - b.cnt;
          sh.s_fill = nfill;
          sh.s_flushed = nflushed;
          if (rc != kOk) sh.status = rc;
        }
      }
      __syncthreads();
      P += sh.s_bits;
      fill = sh.s_fill;
      flushed = sh.s_flushed;
      status = sh.status;
      if (status != kOk) break;
      const uint32_t upto = fill & ~63u;
      for (uint32_t p = flushed + tid; p < upto; p += kWgThreads) dst[p] = ring[p & kMask];
      flushed = upto;
    }
  }
  if (status == kOk)
    for (uint32_t p = flushed + tid; p < fill; p += kWgThreads) dst[p] = ring[p & kMask];
  __threadfence_block();
  if (tid == 0) {
    // hand the reader back as the machine keeps it: the next unread word, and the bits of the current one
    BitSrc src;
    src.init(im.data, im.len);
    const uint64_t word = (P >> 5) * 4;
    const uint32_t used = (uint32_t)(P & 31u);
    st.fill = fill;
    st.bytepos = used ? word + 4 : word;
    st.cnt = used ? 32 - used : 0;
    st.buf = used ? (uint64_t)(src.load32(word) >> used) : 0ull;
    st.win_lo = 0;
    st.win_hi = 0;
    st.win_final = 0;
    if (status == kOk) st.phase = st.after;
    else { st.status = status; st.phase = kPhDone; }
  }
}


// =============================================================================================
// Streams decoded by the whole device.
// A workgroup decodes a stream at the rate of one CU.  Here the main image's part of a stream is cut into
// segments of 65536 stream bits, and three sets of kernels find the tokens of all segments at once.
// Tried first (mode 1), both ending in k_par_verify (a serial check per stream that every segment was entered
// where the one before it was left) and k_par_place (token records -> literals and reference sources):
//   k_par_scout_lanes  a lane per piece of 8192 bits decodes token after token; entries are handed from lane to
//                      lane until they settle (the default)
//   k_par_scout        a workgroup per segment decodes a token at every bit offset and follows the chain after a
//                      warm-up of 4096 bits (the first version, VP8L_DEC_OPT_SCOUT_FORM = 1)
// For a stream whose scouts were not in step (mode 2), exact for any stream at twice the token decoding:
//   k_par_maps    per segment and for each of the 64 bit offsets at which the token chain can enter it (a
//                 token is at most 64 bits long): where the chain leaves the segment and how many pixels it
//                 makes on the way.  Per span this is pointer jumping over (next offset, exit, pixels) for all
//                 1024 offsets, then the spans' maps are composed; no pixel is produced.
//   k_par_chain   per stream, serially over its segments: the entry offset and the first pixel of every
//                 segment (one table lookup per segment).
//   k_par_pixels  per segment again, now knowing where to start: tokens -> literals into the image, and for
//                 every pixel a reference makes the position its value comes from.
//   k_par_jump    pointer jumping over all pixels until every pixel points at a literal (log4 of the longest
//                 chain of references passes); a pixel takes its value in the pass that finds its literal.
// A token the span decoder cannot take (longer than 64 bits, zero bits, a reference longer than 255), or a
// stream that ends early, sends the stream back to the one-workgroup form (status kParFallback).
// =============================================================================================
constexpr uint32_t kParSpans = 64;
constexpr uint32_t kParSegBits = kParSpans * kWgSpan;
constexpr uint32_t kParWindowWords = kParSegBits / 32 + 16;
constexpr uint32_t kParStop = 0x7f, kParNone = 0xff;   // exit codes next to the bit offsets 0..63
constexpr uint32_t kParMinPixels = 1u << 16;            // smaller images are not worth the extra passes
constexpr uint32_t kParJumpPasses = 16;                  // a pass makes every open chain at least four times shorter
constexpr int32_t kParPending = 1, kParFallback = 2, kParRetryMaps = 3;  // DecResult.status values that are not verdicts
constexpr uint32_t kParWarm = 4096;                   // stream bits a scout decodes before its segment to fall in step with the token chain
constexpr uint32_t kParScoutWords = (kParSegBits + kParWarm) / 32 + 16;
constexpr uint32_t kParRecCap = 16384;                 // token records per segment (8 bytes each: 16 times the segment's stream bytes);
                                                       // a segment of 4-bit tokens fills them, shorter tokens send the stream to the entry maps
constexpr uint32_t kFinalBit = 0x80000000u;             // in the per-pixel source array: this position holds its final value

struct ParImage {
  // set by the host
  uint32_t *srcg;      // per pixel: position its value comes from (| kFinalBit)
  uint64_t *segmap;    // nseg * 64: exit | pixels << 32 for each entry offset
  uint32_t *segentry;  // nseg + 1: bit offset at which the chain enters the segment
  uint32_t *segbase;   // nseg + 1: first pixel of the segment (>= total: nothing left for it)
  uint32_t nseg;
  uint32_t mode;       // 1: scouts and token records, 2: entry maps (see below)
  uint64_t *records;   // mode 1: nseg * kParRecCap token records: value | pixels << 32 | is-reference << 40
  // set on the device
  uint32_t active;
  int32_t status;
  uint32_t total, xsize;
  uint32_t tag;        // mode 1, images of at most 2^24 pixels whose alpha code has one symbol: 0x100 | a byte that is no pixel's alpha.  A
                       // pixel that a reference makes is then kept in the image itself until its chain is resolved, as tag byte << 24 |
                       // position of its source, and the per-pixel source array is not used (4 bytes less written per literal, one store
                       // and one read less per copied pixel); 0: not so
  uint32_t fold_green; // the transform to undo first is subtract green: literals are stored with green added (copies then carry it), the pass is dropped
  uint64_t P0;         // stream position of the first token, in bits
  uint32_t *dst;
  uint32_t changed[kParJumpPasses];
  GroupTables tables;
};

__device__ __forceinline__ uint64_t par_window_bits(const VP8L_LDS uint32_t *win, uint32_t a) {  // 64 stream bits from bit a of the window
  const uint32_t idx = a >> 5, sft = a & 31u;
  const uint64_t x = (uint64_t)win[idx] | ((uint64_t)win[idx + 1] << 32);
  return sft ? (x >> sft) | ((uint64_t)win[idx + 2] << (64 - sft)) : x;
}

// ParImage::tag for an image whose alpha code's only symbol is alpha_single (0xffff: it has several) and that has `total` pixels
__device__ __forceinline__ uint32_t par_pixel_tag(uint32_t alpha_single, uint64_t total) {
#ifdef VP8L_NO_PIXEL_TAGS
  return 0u;
#else
  return alpha_single != 0xffffu && total <= (1ull << 24) ? 0x100u | ((alpha_single ^ 0x80u) & 0xffu) : 0u;
#endif
}

__device__ __forceinline__ void par_fail(ParImage *par, int32_t code) {
  // handing the stream to another form wins over a verdict (that form then gives the verdict); among verdicts the first stays
  if (code == kParFallback) atomicExch(&par->status, code);
  else if (code == kParRetryMaps) { atomicCAS(&par->status, (int32_t)kOk, code); atomicCAS(&par->status, (int32_t)kErrInvalidVp8l, code); }
  else atomicCAS(&par->status, (int32_t)kOk, code);
}

// One round of the workgroup decoder without its output stage: the tokens at window bits a + tid (tid < lim; the
// others end the chain), the chain from offset 0, and for every token on it the pixels and tokens before it.
// Leaves sh.end_off (bits the round consumed) set after the caller's next barrier; pixels made = r.all_px.
struct WgRound {
  uint32_t hop, cnt, val, ex, rank, all_px, all_tok;
  bool ref, on;
};
__device__ __forceinline__ void wg_round(VP8L_LDS uint32_t *win, uint32_t a, uint32_t lim, const VP8L_LDS GroupTables &g, VP8L_LDS WgShared &sh,
                                         uint32_t xsize, uint32_t tid, WgRound &r) {
  const uint32_t lane = tid & 63, wave = tid >> 6;
  r.hop = 0; r.cnt = 0; r.val = 0; r.ref = false;
  if (tid < lim) wg_token(par_window_bits(win, a + tid), g, sh, xsize, r.hop, r.cnt, r.val, r.ref);
  uint32_t s1[kWgLevels], s2[kWgLevels], s3[kWgLevels];
  s1[0] = r.hop && tid + r.hop < kWgSpan ? tid + r.hop : kWgSpan;
  sh.ja[tid] = (uint16_t)s1[0];
  sh.mark[tid] = tid == 0 ? (uint8_t)(kWgLevels + 1) : (uint8_t)0;
  if (tid == 0) { sh.ja[kWgSpan] = (uint16_t)kWgSpan; sh.jb[kWgSpan] = (uint16_t)kWgSpan; }
  VP8L_LDS uint16_t *cur = sh.ja, *oth = sh.jb;
#pragma unroll
  for (uint32_t k = 0; k < kWgLevels; k++) {
    __syncthreads();
    s2[k] = cur[s1[k]];
    s3[k] = cur[s2[k]];
    if (k + 1 < kWgLevels) {
      s1[k + 1] = cur[s3[k]];
      oth[tid] = (uint16_t)s1[k + 1];
      VP8L_LDS uint16_t *t16 = cur; cur = oth; oth = t16;
    }
  }
  const uint32_t beyond = s2[kWgLevels - 1];
#pragma unroll
  for (int k = (int)kWgLevels - 1; k >= 0; k--) {
    if (k != (int)kWgLevels - 1) __syncthreads();
    if (sh.mark[tid] > (uint32_t)k + 1) {
      sh.mark[s1[k]] = (uint8_t)(k + 1);
      if (k != (int)kWgLevels - 1) {
        sh.mark[s2[k]] = (uint8_t)(k + 1);
        sh.mark[s3[k]] = (uint8_t)(k + 1);
      }
    }
  }
  __syncthreads();
  r.on = sh.mark[tid] != 0;
  const uint32_t c = r.on && r.hop ? r.cnt | (1u << 16) : 0u;  // pixels (at most 128 * 255) and tokens (at most 128) in one scan
  const uint32_t incl = wave_incl_scan_u32(c, (int)lane);
  if (lane == 63) sh.wsum[wave] = incl;
  if (tid == 0) { sh.defer_off = 0xffffffffu; sh.end_off = 0; sh.end_px = 0; }
  __syncthreads();
  const uint32_t wsums = wave_incl_scan_u32(lane < kWgWaves ? sh.wsum[lane] : 0u, (int)lane);
  const uint32_t before = wave ? lane_read(wsums, wave - 1) : 0u, all = lane_read(wsums, kWgWaves - 1);
  const uint32_t exc = before + incl - c;
  r.ex = exc & 0xffffu;
  r.rank = exc >> 16;
  r.all_px = all & 0xffffu;
  r.all_tok = all >> 16;
  // where the round ends: after the last token of the chain, at a token the round cannot take, or after 128 tokens
  if (tid == 0 && beyond != kWgSpan) sh.end_off = beyond;
  if (r.on) {
    if (r.hop == 0) sh.end_off = tid;
    else if (s1[0] == kWgSpan) sh.end_off = tid + r.hop;
  }
}

// Tables and long-code lists into LDS, the segment's stream bytes into the window.  Returns the window's first bit.
__device__ __forceinline__ uint64_t par_setup(VP8L_LDS GroupTables &g, VP8L_LDS WgShared &sh, VP8L_LDS uint32_t *win, const DecImage &im,
                                              const ParImage *par, uint32_t seg, uint32_t tid) {
  for (uint32_t i = tid; i < sizeof(GroupTables) / 4; i += kWgThreads)
    reinterpret_cast<VP8L_LDS uint32_t *>(&g)[i] = reinterpret_cast<const uint32_t *>(&par->tables)[i];
  const uint64_t sb = par->P0 + (uint64_t)seg * kParSegBits, lo = (sb >> 3) & ~3ull;
  wg_window_load(win, im, lo, tid, kParWindowWords);
  __syncthreads();
  wg_prepare_long(sh, g, tid);
  __syncthreads();
  return lo * 8;
}


__global__ __launch_bounds__(kWgThreads, 8) void k_par_maps(const DecImage *__restrict__ imgs, const uint32_t *__restrict__ segfirst, uint32_t n) {
  __shared__ GroupTables g;
  __shared__ uint32_t win[kParWindowWords];
  __shared__ __attribute__((aligned(16))) WgShared sh;
  __shared__ uint64_t ea[kWgSpan + 1], eb[kWgSpan + 1];  // per offset: next offset | exit << 16 | pixels << 32; entry kWgSpan is the sink
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t img = find_image(segfirst, n, blockIdx.x), seg = blockIdx.x - segfirst[img];
  const DecImage im = imgs[img];
  ParImage *par = im.par;
  if (!par || !par->active || par->mode != 2 || seg >= par->nseg) return;
  const uint64_t bit0 = par_setup((VP8L_LDS GroupTables &)g, (VP8L_LDS WgShared &)sh, (VP8L_LDS uint32_t *)win, im, par, seg, tid);
  const uint32_t first = (uint32_t)(par->P0 + (uint64_t)seg * kParSegBits - bit0), xsize = par->xsize;
  const uint64_t sink = (uint64_t)kWgSpan | ((uint64_t)kParNone << 16);
  if (tid == 0) { ea[kWgSpan] = sink; eb[kWgSpan] = sink; }
  uint32_t at = lane, made = 0;  // wave 0, lane e: where the chain that entered the segment at offset e is now, pixels so far
  for (uint32_t s = 0; s < kParSpans; s++) {
    __syncthreads();
    uint32_t hop, cnt, val;
    bool ref;
    wg_token(par_window_bits((VP8L_LDS uint32_t *)win, first + s * kWgSpan + tid), (const VP8L_LDS GroupTables &)g, (const VP8L_LDS WgShared &)sh, xsize,
             hop, cnt, val, ref);
    const uint32_t nxt = hop && tid + hop < kWgSpan ? tid + hop : kWgSpan;
    const uint32_t ex0 = hop == 0 ? kParStop : (tid + hop >= kWgSpan ? tid + hop - kWgSpan : kParNone);
    uint64_t pk = (uint64_t)nxt | ((uint64_t)ex0 << 16) | ((uint64_t)(hop ? cnt : 0u) << 32);
    ea[tid] = pk;
    VP8L_LDS uint64_t *cur = (VP8L_LDS uint64_t *)ea, *oth = (VP8L_LDS uint64_t *)eb;
#pragma unroll
    for (uint32_t lev = 0; lev < 5; lev++) {  // 4^5 tokens: more than a span can hold
      __syncthreads();
      const uint64_t e1 = cur[(uint32_t)pk & 0xffffu], e2 = cur[(uint32_t)e1 & 0xffffu], e3 = cur[(uint32_t)e2 & 0xffffu];
      uint32_t ex = (uint32_t)(pk >> 16) & 0xffu;  // the first exit on the way
      if (ex == kParNone) ex = (uint32_t)(e1 >> 16) & 0xffu;
      if (ex == kParNone) ex = (uint32_t)(e2 >> 16) & 0xffu;
      if (ex == kParNone) ex = (uint32_t)(e3 >> 16) & 0xffu;
      const uint32_t px = (uint32_t)(pk >> 32) + (uint32_t)(e1 >> 32) + (uint32_t)(e2 >> 32) + (uint32_t)(e3 >> 32);
      pk = (e3 & 0xffffull) | ((uint64_t)ex << 16) | ((uint64_t)px << 32);
      if (lev < 4) {
        oth[tid] = pk;
        VP8L_LDS uint64_t *t = cur; cur = oth; oth = t;
      }
    }
    if (wave == 0) {  // lane e now holds the span's map for entry e; move every tracked chain through it
      const uint32_t ex = (uint32_t)(pk >> 16) & 0xffu, px = (uint32_t)(pk >> 32);
      const uint32_t nex = (uint32_t)__shfl((int)ex, (int)(at & 63u), 64), npx = (uint32_t)__shfl((int)px, (int)(at & 63u), 64);
      if (at != kParStop) { made += npx; at = nex; }
    }
  }
  if (wave == 0) par->segmap[(size_t)seg * 64 + lane] = (uint64_t)at | ((uint64_t)made << 32);
}

__global__ __launch_bounds__(256) void k_par_chain(const DecImage *__restrict__ imgs) {
  constexpr uint32_t kChunk = 32;
  __shared__ uint64_t chunk[kChunk * 64];
  const uint32_t tid = threadIdx.x;
  ParImage *par = imgs[blockIdx.x].par;
  if (!par || !par->active || par->mode != 2) return;
  const uint32_t nseg = par->nseg, total = par->total;
  uint32_t e = 0;
  uint64_t base = 0;
  bool stopped = false;  // thread 0
  for (uint32_t g0 = 0; g0 < nseg; g0 += kChunk) {
    const uint32_t cnt = nseg - g0 < kChunk ? nseg - g0 : kChunk;
    for (uint32_t i = tid; i < cnt * 64; i += 256) chunk[i] = par->segmap[(size_t)g0 * 64 + i];
    __syncthreads();
    if (tid == 0) {
      for (uint32_t k = 0; k < cnt; k++) {
        par->segentry[g0 + k] = e;
        par->segbase[g0 + k] = base < total && !stopped ? (uint32_t)base : total;
        if (base >= total || stopped) continue;
        const uint64_t m = chunk[k * 64 + e];
        base += m >> 32;
        const uint32_t x = (uint32_t)m & 0xffu;
        if (x == kParStop) stopped = true;
        else e = x;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    par->segentry[nseg] = e;
    par->segbase[nseg] = total;
    // the chain must reach the end of the image inside the segments, without meeting a token the span decoder cannot take
    if (base < total) par_fail(par, kParFallback);
  }
}

__global__ __launch_bounds__(kWgThreads, 8) void k_par_pixels(const DecImage *__restrict__ imgs, const uint32_t *__restrict__ segfirst, uint32_t n) {
  __shared__ GroupTables g_;
  __shared__ uint32_t win_[kParWindowWords];
  __shared__ __attribute__((aligned(16))) WgShared sh_;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t img = find_image(segfirst, n, blockIdx.x), seg = blockIdx.x - segfirst[img];
  const DecImage im = imgs[img];
  ParImage *par = im.par;
  if (!par || !par->active || par->mode != 2 || seg >= par->nseg || par->status != kOk) return;
  const uint32_t total = par->total, xsize = par->xsize;
  uint32_t fill = par->segbase[seg];
  if (fill >= total) return;
  VP8L_LDS GroupTables &g = (VP8L_LDS GroupTables &)g_;
  VP8L_LDS WgShared &sh = (VP8L_LDS WgShared &)sh_;
  VP8L_LDS uint32_t *win = (VP8L_LDS uint32_t *)win_;
  const uint64_t bit0 = par_setup(g, sh, win, im, par, seg, tid);
  const uint32_t first = (uint32_t)(par->P0 + (uint64_t)seg * kParSegBits - bit0);
  // window bit positions: where this segment's chain starts, and where the next segment's takes over
  uint32_t P = first + par->segentry[seg];
  const uint32_t hard_end = kParWindowWords * 32 - 192;  // nothing is read beyond the window
  const uint32_t P_end = par->segbase[seg + 1] < total ? first + kParSegBits + par->segentry[seg + 1] : hard_end;
  uint32_t *dst = par->dst, *srcg = par->srcg;
  const bool fold_green = par->fold_green != 0;
  if (tid == 0) sh.status = kOk;
  int32_t status = kOk;
  WgRound r;
  while (P < P_end && fill < total) {
    __syncthreads();
    wg_round(win, P, P_end - P, g, sh, xsize, tid, r);  // offsets from P_end on belong to the next segment
    const uint32_t pos = fill + r.ex, cnt = r.cnt, val = r.val;
    bool expand = false;
    if (r.on && r.hop && pos < total) {  // tokens past the last pixel are not the image's
      if (pos + cnt > total) sh.status = kErrInvalidVp8l;  // a reference that runs past the end of the image
      else if (!r.ref) { dst[pos] = fold_green ? add_green(val) : val; srcg[pos] = pos | kFinalBit; }
      else if (val > pos) sh.status = kErrInvalidVp8l;     // a reference before the start of the image
      else expand = true;
    }
    // pixels that references make: only where their values come from, to be resolved by k_par_jump
    if (expand && cnt <= kWgOwnCopy) {
      if (val >= cnt) {
        for (uint32_t e = 0; e < cnt; e++) srcg[pos + e] = pos - val + e;
      } else {
        for (uint32_t e = 0; e < cnt; e++) srcg[pos + e] = pos - val + e % val;
      }
    }
    uint64_t m = __ballot(expand && cnt > kWgOwnCopy);
    while (m) {
      const uint32_t l = (uint32_t)__builtin_ctzll(m);
      m &= m - 1;
      const uint32_t p2 = lane_read(pos, l), n2 = lane_read(cnt, l), dist = lane_read(val, l);
      for (uint32_t c2 = 0; c2 < n2; c2 += 64) {
        const uint32_t e = c2 + lane;
        if (e < n2) srcg[p2 + e] = p2 - dist + (dist < n2 ? e % dist : e);
      }
    }
    __syncthreads();
    status = sh.status;
    if (status != kOk) break;
    const uint32_t adv = sh.end_off;
    if (adv == 0) { status = kParFallback; break; }  // not expected after k_par_maps found its way through
    fill += r.all_px;
    P += adv;
  }
  if (tid == 0) {
    if (status != kOk) par_fail(par, status);
    // the chain of this segment must end exactly where the next one starts
    else if (fill < total && P != P_end) par_fail(par, kParFallback);
  }
}


// ---- mode 1: scouts.  A Huffman-coded stream falls in step by itself: a decoder started at a wrong bit almost
// always lands on a true token start within a few tokens and stays on the chain from then on.  A scout starts
// kParWarm bits before its segment, records every token of its segment (value, pixels, kind) once, and says where it
// entered and left the segment; k_par_verify accepts the records only if every scout entered where its predecessor
// left (segment 0 starts at the true position, so by induction all of them were on the true chain).  If one was not
// (codes of one length never fall in step), the stream is decoded again with the entry maps of mode 2.
__global__ __launch_bounds__(kWgThreads, 8) void k_par_scout(const DecImage *__restrict__ imgs, const uint32_t *__restrict__ segfirst, uint32_t n) {
  __shared__ GroupTables g_;
  __shared__ uint32_t win_[kParScoutWords];
  __shared__ __attribute__((aligned(16))) WgShared sh_;
  const uint32_t tid = threadIdx.x;
  const uint32_t img = find_image(segfirst, n, blockIdx.x), seg = blockIdx.x - segfirst[img];
  const DecImage im = imgs[img];
  ParImage *par = im.par;
  if (!par || !par->active || par->mode != 1 || seg >= par->nseg) return;
  VP8L_LDS GroupTables &g = (VP8L_LDS GroupTables &)g_;
  VP8L_LDS WgShared &sh = (VP8L_LDS WgShared &)sh_;
  VP8L_LDS uint32_t *win = (VP8L_LDS uint32_t *)win_;
  for (uint32_t i = tid; i < sizeof(GroupTables) / 4; i += kWgThreads)
    reinterpret_cast<VP8L_LDS uint32_t *>(&g)[i] = reinterpret_cast<const uint32_t *>(&par->tables)[i];
  const uint64_t sb = par->P0 + (uint64_t)seg * kParSegBits;  // seg >= 1: sb - kParWarm >= P0
  const uint32_t warm = seg ? kParWarm : 0;
  const uint64_t lo = ((sb - warm) >> 3) & ~3ull;
  wg_window_load(win, im, lo, tid, kParScoutWords);
  __syncthreads();
  wg_prepare_long(sh, g, tid);
  __syncthreads();
  const uint32_t xsize = par->xsize;
  const uint32_t seg_start = (uint32_t)(sb - lo * 8), seg_end = seg_start + kParSegBits;  // window bit positions
  uint32_t P = seg_start - warm;
  WgRound r;
  while (P < seg_start) {  // falling in step: nothing is kept
    __syncthreads();
    wg_round(win, P, seg_start - P, g, sh, xsize, tid, r);
    __syncthreads();
    const uint32_t adv = sh.end_off;
    P += adv ? adv : 1;  // a token this decoder cannot take at a position that need not be a token start: try the next bit
  }
  const uint32_t entry = P - seg_start;
  uint64_t *rec = par->records + (size_t)seg * kParRecCap;
  uint32_t ntok = 0, npx = 0, flags = 0;
  while (P < seg_end) {
    __syncthreads();
    wg_round(win, P, seg_end - P, g, sh, xsize, tid, r);
    if (r.on && r.hop) {
      const uint32_t i = ntok + r.rank;
      if (i < kParRecCap) rec[i] = (uint64_t)r.val | ((uint64_t)r.cnt << 32) | ((uint64_t)(r.ref ? 1u : 0u) << 40);
    }
    __syncthreads();
    const uint32_t adv = sh.end_off;
    ntok += r.all_tok;
    npx += r.all_px;
    if (adv == 0) { flags |= 1; break; }  // a token the round decoder cannot take: the stream goes to the one-workgroup form
    P += adv;
  }
  if (ntok > kParRecCap) flags |= 2;
  if (tid == 0) {
    uint64_t *slot = par->segmap + (size_t)seg * 64;
    slot[0] = (uint64_t)(P - seg_end) | ((uint64_t)npx << 32);
    slot[1] = (uint64_t)entry | ((uint64_t)ntok << 32);
    slot[2] = flags;
    slot[3] = 0;  // the records lie in one piece
  }
}

// ---- mode 1, lane form of the scouts.  k_par_scout finds the token chain of a span by decoding a token at every one
// of its 1024 bit offsets; here a lane decodes tokens one after another, as a serial decoder would.  The stream is cut
// into pieces of kLaneBits bits, one per lane; a segment is kLanesPerSeg pieces, a workgroup takes kLaneSegsPerWg
// segments of one stream (and the piece before them: warm-up) and keeps the stream's code tables in LDS.  Every lane
// first decodes its piece from the piece's first bit (a guess), then again from where the lane before it really left
// its piece, until no lane's entry changes.  The warm-up lane's entry stays a guess (it falls in step over its piece,
// as k_par_scout does over kParWarm bits; the stream's first piece starts at the first token), so after pass k the
// first k lanes are final, and because a Huffman-coded stream falls in step by itself - within a piece, nearly
// always - two or three passes settle all lanes.  A last pass writes the token records.  The stream bits are decoded
// about four times instead of 64 times.
// Lanes that each walk through their own kilobyte of the stream would fetch a cache line per 8 bytes they use, and
// with 2048 lanes on a CU no cache keeps those lines until their next use (measured: 4.4 times slower than this).
// So the lanes move in rounds of kLaneChunkBits bits: the workgroup loads the next 64 bytes of every lane's piece
// with whole-line reads into registers while the lanes decode, and puts them into a ring in LDS at the round's end.
// What the kernel leaves behind (records, entry, exit, counts, flags per segment) is what k_par_scout leaves, so
// k_par_verify and k_par_place follow.
constexpr uint32_t kLaneBits = 8192;
constexpr uint32_t kLanesPerSeg = kParSegBits / kLaneBits;
#ifndef VP8L_LANE_THREADS
#define VP8L_LANE_THREADS 256
#endif
constexpr uint32_t kLaneThreads = VP8L_LANE_THREADS;  // 256 or 512 (a compile-time choice, kept for experiments)
constexpr uint32_t kLaneSegsPerWg = kLaneThreads / kLanesPerSeg - 1;  // the last lane of the first group has the warm-up piece
constexpr uint32_t kLaneChunkWords = 4;                               // 64-bit words of every piece loaded at a time
constexpr uint32_t kLaneChunkBits = 64 * kLaneChunkWords;
constexpr uint32_t kLaneRounds = kLaneBits / kLaneChunkBits;
constexpr uint32_t kLaneRing = 2 * kLaneChunkWords;                   // the chunk being decoded and the one after it
constexpr uint32_t kLaneRingStride = kLaneThreads + 1;                // words of one lane in different banks
constexpr uint32_t kLanePasses = 8;   // entries still changing after this many passes: the stream does not fall in step
constexpr uint32_t kLaneGuessRound = kLaneRounds * 3 / 4;             // the guess of a lane's exit starts here (from 1/2: 10.9 ms, 3/4: 9.5, 7/8: 9.8 per 1080p batch)
constexpr uint32_t kLaneRecCap = kParRecCap / kLanesPerSeg;           // token records per lane: tokens of 4 bits fill them
static_assert(4 + kLanesPerSeg <= 64, "a segment's slots hold the lanes' record counts");
static_assert(kLaneRecCap % 2 == 0, "records are stored in pairs");
// What a store of several records at once relies on (a lane's part of the record array is its own; `tok` counts the lane's
// records since the pass began):
//   * the array starts at a multiple of 16 bytes (hipMalloc gives 256) and a segment's records, kParRecCap * 8 bytes, and a
//     lane's part, kLaneRecCap * 8 bytes, are multiples of the store's width: the pair that ends at an odd `tok` starts at
//     rec + tok - 1, a multiple of 16 bytes.  Groups of G records would need kLaneRecCap % G == 0 (true up to 2048) AND a flush
//     of up to G - 1 held records on every way out of the pass (the end of the piece, a token that cannot be taken, the cap) -
//     with pairs that is the one record `held`;
//   * a record is held or stored only while tok < cap, and cap is even, so a held record always has room for its neighbour;
//   * a pass that decodes a piece again rewrites the part from its first record; records behind the new count are stale and
//     never read (the counts per part are what k_par_place goes by).
// Wider groups were tried (four records, 32 bytes): 14 more registers, four waves per SIMD instead of five, and no gain
// over pairs beyond the one the pairs already give; pairs are kept.
static_assert((kParRecCap * 8) % 16 == 0 && (kLaneRecCap * 8) % 16 == 0, "every lane's part of the records starts at a multiple of 16 bytes");
static_assert(kLaneBits < kParSegBits && (kLanesPerSeg & (kLanesPerSeg - 1)) == 0 && kLanesPerSeg <= 64, "the lanes of a segment sit in one wave");
static_assert(kLaneThreads % 64 == 0 && kLaneThreads >= kNumCodes * kWgLongSlots, "whole waves; one thread per long-code slot");
static_assert(64 % kLaneChunkWords == 0 && kLaneBits % kLaneChunkBits == 0 && kLaneBits % 64 == 0, "chunks");
// What the kernel assumes about its workgroup size, so that it can be built with 256 or 512 threads (both are tested):
static_assert(kParStop < 256 && kParStop > 63 && kParNone < 256, "exit codes and the bit offsets 0..63 fit the uint8_t exit_off entries");
static_assert(kLaneThreads / kLanesPerSeg >= 2, "a workgroup holds the warm-up piece's group of lanes and at least one segment");
static_assert(kLaneThreads <= 1024, "one workgroup");

struct LaneShared {
  uint32_t lim[kNumCodes][kWgLongSlots];   // as in WgShared
  uint32_t base[kNumCodes][kWgLongSlots];
  uint16_t sorted[kWgSorted];
  uint8_t exit_off[kLaneThreads];          // where the chain of lane j enters piece j + 1 (0..63), or kParStop
  uint64_t ring[kLaneRing][kLaneRingStride];  // [word of the piece mod kLaneRing][lane]
};

// The stream as 64-bit words from its first byte rounded down to 8 bytes; bytes beyond its end read as zero.
struct LaneStream {
  const uint64_t *base;
  uint64_t nbytes;   // from `base` to the stream's end
  int64_t word0;     // the workgroup's lane 0 would start in this word (lanes below the warm-up lane do not exist)
  __device__ __forceinline__ uint64_t word(int64_t i) const {
    const uint64_t at = 8ull * (uint64_t)i;
    if (i < 0 || at >= nbytes) return 0;
    const uint64_t v = base[i];  // an aligned word that holds a byte of the stream: readable to its end
    return at + 8 <= nbytes ? v : v & (~0ull >> (8 * (8 - (uint32_t)(nbytes - at))));
  }
  // chunk c of the pieces of the 64 lanes of a wave, loaded by that wave: kLaneChunkWords neighbouring threads per lane
  __device__ __forceinline__ void load(uint32_t c, uint32_t tid, uint64_t (&regs)[kLaneChunkWords]) const {
#pragma unroll
    for (uint32_t i = 0; i < kLaneChunkWords; i++) {
      const uint32_t l = (tid & ~63u) + (tid & 63u) / kLaneChunkWords + i * (64 / kLaneChunkWords), w = tid % kLaneChunkWords;
      regs[i] = word(word0 + (int64_t)l * (kLaneBits / 64) + c * kLaneChunkWords + w);
    }
  }
  __device__ __forceinline__ void store(uint32_t c, uint32_t tid, const uint64_t (&regs)[kLaneChunkWords], VP8L_LDS LaneShared &sh) const {
#pragma unroll
    for (uint32_t i = 0; i < kLaneChunkWords; i++) {
      const uint32_t l = (tid & ~63u) + (tid & 63u) / kLaneChunkWords + i * (64 / kLaneChunkWords), w = tid % kLaneChunkWords;
      sh.ring[(c * kLaneChunkWords + w) % kLaneRing][l] = regs[i];
    }
  }
};

// A wave's part of the ring is written and read by that wave alone: its LDS operations complete in order, the
// compiler must keep them in order too.
__device__ __forceinline__ void lane_ring_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One pass of the whole workgroup over its pieces; every thread calls it (the waves do not wait for each other).  A
// lane with `run` decodes the tokens from bit `in` of round r0 of its piece until one starts in the next piece (`out`:
// how far in), or until it meets a token it cannot take (out = kParStop; a warm-up lane moves on by a bit instead).
// kRecord: the records go to rec[0..cap).
template <bool kRecord>
__device__ __forceinline__ void lanes_pass(const LaneStream &ls, uint32_t s0, bool run, bool warm, uint32_t in, const VP8L_LDS GroupTables &g,
                                           VP8L_LDS LaneShared &sh, uint32_t xsize, uint32_t tid, uint64_t *rec, uint32_t cap,
                                           uint32_t r0, uint32_t &out, uint32_t &ntok, uint32_t &npx) {
  const uint32_t alpha_single = g.t[kAlpha].single;  // the same for the whole workgroup
  uint64_t regs[kLaneChunkWords];
  ls.load(r0, tid, regs);
  ls.store(r0, tid, regs, sh);
  ls.load(r0 + 1, tid, regs);
  ls.store(r0 + 1, tid, regs, sh);
  lane_ring_sync();
  uint32_t rel = r0 * kLaneChunkBits + in, k = (s0 + rel) >> 6, s = (s0 + rel) & 63u, tok = 0, px = 0;
  uint64_t A = sh.ring[k % kLaneRing][tid], B = sh.ring[(k + 1) % kLaneRing][tid], C = sh.ring[(k + 2) % kLaneRing][tid];
  uint64_t held = 0;  // an even-numbered record waiting for its neighbour
  bool stopped = false;
#pragma unroll 1
  for (uint32_t r = r0; r < kLaneRounds; r++) {
    const bool more = r + 2 <= kLaneRounds;   // the last round reads into the first chunk of the next piece
    if (more) ls.load(r + 2, tid, regs);
    if (run && !stopped) {
      const uint32_t lim = (r + 1) * kLaneChunkBits;
      while (rel < lim) {
        uint32_t hop, cnt, val;
        bool ref;
        wg_token<kRecord>(s ? (A >> s) | (B << (64 - s)) : A, g, sh, xsize, hop, cnt, val, ref, nullptr, alpha_single);
        if (hop == 0) {
          if (!warm) { stopped = true; break; }
          hop = 1;  // this need not be a token start, try the next bit
        } else {
          if (kRecord && tok < cap) {
            // two records to a 16-byte store (the parts of the record array start at multiples of 16 bytes, cap is even)
            const uint64_t rcd = (uint64_t)val | ((uint64_t)cnt << 32) | ((uint64_t)(ref ? 1u : 0u) << 40);
            if (tok & 1u) *reinterpret_cast<ulonglong2 *>(rec + tok - 1) = make_ulonglong2(held, rcd);
            else held = rcd;
          }
          tok++;
          px += cnt;
        }
        rel += hop;
        s += hop;
        if (s >= 64) { s -= 64; k++; A = B; B = C; C = sh.ring[(k + 2) % kLaneRing][tid]; }
      }
    }
    lane_ring_sync();
    if (more) ls.store(r + 2, tid, regs, sh);
    lane_ring_sync();
  }
  if (kRecord && run && (tok & 1u) && tok - 1 < cap) rec[tok - 1] = held;
  if (run) { out = stopped ? kParStop : rel - kLaneBits; ntok = tok; npx = px; }
}

static_assert(sizeof(LaneShared) + sizeof(GroupTables) <= (64u << 10), "static LDS of k_par_scout_lanes");
__global__ __launch_bounds__(kLaneThreads, 5) void k_par_scout_lanes(const DecImage *__restrict__ imgs, const uint32_t *__restrict__ wgfirst, uint32_t n) {
  __shared__ GroupTables g_;
  __shared__ __attribute__((aligned(16))) LaneShared sh_;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t img = find_image(wgfirst, n, blockIdx.x), wg = blockIdx.x - wgfirst[img];
  const DecImage im = imgs[img];
  ParImage *par = im.par;
  if (!par || !par->active || par->mode != 1 || wg * kLaneSegsPerWg >= par->nseg) return;
  VP8L_LDS GroupTables &g = (VP8L_LDS GroupTables &)g_;
  VP8L_LDS LaneShared &sh = (VP8L_LDS LaneShared &)sh_;
  for (uint32_t i = tid; i < sizeof(GroupTables) / 4; i += kLaneThreads)
    reinterpret_cast<VP8L_LDS uint32_t *>(&g)[i] = reinterpret_cast<const uint32_t *>(&par->tables)[i];
  __syncthreads();
  wg_prepare_long(sh, g, tid, kLaneThreads);
  __syncthreads();
  const uint32_t xsize = par->xsize;
  // lanes 0 .. kLanesPerSeg - 2: none; lane kLanesPerSeg - 1: the piece before the workgroup's segments; then the segments' pieces
  const uint32_t first = wg ? kLanesPerSeg - 1 : kLanesPerSeg;      // the lane whose entry is fixed
  const uint32_t seg = wg * kLaneSegsPerWg + tid / kLanesPerSeg - 1;  // this lane's segment (tid >= kLanesPerSeg)
  const int64_t q = (int64_t)wg * kLaneSegsPerWg * kLanesPerSeg + tid - kLanesPerSeg;  // this lane's piece of the stream
  const uint32_t down = (uint32_t)(reinterpret_cast<uintptr_t>(im.data) & 7u);          // the stream is 4-byte aligned
  const uint64_t bit0 = par->P0 + 8 * down;  // the first token, counted from the stream's first byte rounded down to 8 bytes
  LaneStream ls;
  ls.base = reinterpret_cast<const uint64_t *>(im.data - down);
  ls.nbytes = im.len + down;
  ls.word0 = (int64_t)(bit0 >> 6) + ((int64_t)wg * kLaneSegsPerWg * kLanesPerSeg - kLanesPerSeg) * (kLaneBits / 64);
  const uint32_t s0 = (uint32_t)bit0 & 63u;  // every piece starts at this bit of its first word
  // Pieces that start beyond the stream's last byte hold no token of a complete stream (and zeros never fall in step:
  // they repeat one token); a stream that ends before its image does gets its verdict from the one-workgroup form.
  const bool live = tid >= kLanesPerSeg && seg < par->nseg;
  const bool active = tid >= first && (tid < kLanesPerSeg || live) && par->P0 + (uint64_t)q * kLaneBits < 8ull * im.len;
  const bool warm_lane = tid < kLanesPerSeg;
  const uint32_t sub = lane & (kLanesPerSeg - 1);
  // this lane's part of its segment's records: written again whenever the lane decodes again
  uint64_t *rec = par->records + (size_t)(live ? seg : 0u) * kParRecCap + sub * kLaneRecCap;
  uint32_t in = 0, done_in = 0xffffffffu, out = 0, ntok = 0, npx = 0;
  // The guess: where a decoder started in the last quarter of the piece leaves it.  It has fallen in step by then
  // nearly always, so this is where the true chain leaves the piece, found at a quarter of the price of a pass.
  lanes_pass<false>(ls, s0, active, true, 0, g, sh, xsize, tid, nullptr, 0, kLaneGuessRound, out, ntok, npx);
  if (active) sh.exit_off[tid] = (uint8_t)out;
  __syncthreads();
  if (tid > first && active) in = sh.exit_off[tid - 1];
  bool settled = false;
  for (uint32_t pass = 0; pass < kLanePasses; pass++) {
    const bool redo = active && in != done_in && in != kParStop;
    if (!__syncthreads_or(redo)) { settled = true; break; }
    lanes_pass<true>(ls, s0, redo, warm_lane, in, g, sh, xsize, tid, rec, live ? kLaneRecCap : 0u, 0, out, ntok, npx);
    if (redo) {
      done_in = in;
      sh.exit_off[tid] = (uint8_t)out;
    }
    __syncthreads();
    if (tid > first && active) in = sh.exit_off[tid - 1];
  }
  // per segment (kLanesPerSeg neighbouring lanes): the first lane that met a token it cannot take ends the records
  uint32_t stop = live && active && (in == kParStop || out == kParStop) ? sub : kLanesPerSeg;
#pragma unroll
  for (uint32_t o = 1; o < kLanesPerSeg; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_xor((int)stop, (int)o, 64);
    stop = t < stop ? t : stop;
  }
  if (!live || !active || sub > stop || in == kParStop) { ntok = 0; npx = 0; }
  uint32_t itok = ntok, all_px = npx;
#pragma unroll
  for (uint32_t o = 1; o < kLanesPerSeg; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)itok, o, 64);
    if (sub >= o) itok += t;
    all_px += (uint32_t)__shfl_xor((int)all_px, (int)o, 64);
  }
  const uint32_t all_tok = (uint32_t)__shfl((int)itok, (int)(lane | (kLanesPerSeg - 1)), 64);
  uint32_t over = ntok > kLaneRecCap ? 1u : 0u;  // a lane with more tokens than records
#pragma unroll
  for (uint32_t o = 1; o < kLanesPerSeg; o <<= 1) over |= (uint32_t)__shfl_xor((int)over, (int)o, 64);
  uint32_t flags = 0;
  if (!settled) flags |= 2;                 // not in step: the entry maps decode this stream
  if (over) flags |= 2;
  if (stop < kLanesPerSeg) flags |= 1;      // the stream goes to the one-workgroup form if the image is not complete by then
  if (live) {
    uint64_t *slot = par->segmap + (size_t)seg * 64;
    slot[3] = 1;                            // the records lie in kLanesPerSeg parts ...
    slot[4 + sub] = itok - ntok;            // ... and this many tokens come before this lane's part
    if (sub == kLanesPerSeg - 1) {
      slot[0] = (uint64_t)(stop < kLanesPerSeg || !active ? 0u : out) | ((uint64_t)all_px << 32);
      slot[2] = flags;
    }
    if (sub == 0) slot[1] = (uint64_t)(in == kParStop ? 0u : in) | ((uint64_t)all_tok << 32);
  }
}

// The segments' records are accepted if every scout entered where the one before it left (segment 0 starts at the true
// position, so by induction all were on the true chain) and none ran out of records; the pixel counts add up to each
// segment's first pixel.  256 segments a round: the checks are independent once the sums are known (a scan), the first
// segment that fails decides (an LDS minimum) - one thread walking the segments took 2.2 ms for a 16383 x 16383 image.
__global__ __launch_bounds__(256) void k_par_verify(const DecImage *__restrict__ imgs) {
  constexpr uint32_t kChunk = 256;
  __shared__ uint32_t s_exit[kChunk], wsum[kChunk / 64], s_first;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ParImage *par = imgs[blockIdx.x].par;
  if (!par || !par->active || par->mode != 1) return;
  const uint32_t nseg = par->nseg, total = par->total;
  uint32_t expect = 0;  // where the chain enters the round's first segment
  uint64_t base0 = 0;   // pixels before the round's first segment
  int32_t verdict = kOk;
  for (uint32_t g0 = 0; g0 < nseg; g0 += kChunk) {
    const uint32_t cnt = nseg - g0 < kChunk ? nseg - g0 : kChunk;
    uint32_t px = 0, entry = 0, leave = 0, fl = 0;
    if (tid < cnt) {
      const uint64_t *slot = par->segmap + (size_t)(g0 + tid) * 64;
      const uint64_t v0 = slot[0], v1 = slot[1], v2 = slot[2];
      px = (uint32_t)(v0 >> 32);
      leave = (uint32_t)v0 & 0xffu;
      entry = (uint32_t)v1 & 0xffu;
      fl = (uint32_t)v2;
    }
    const uint32_t incl = wave_incl_scan_u32(px, (int)lane);
    if (lane == 63) wsum[wave] = incl;
    s_exit[tid] = leave;
    if (tid == 0) s_first = 0xffffffffu;
    __syncthreads();
    uint64_t before = base0, all = 0;
    for (uint32_t k = 0; k < kChunk / 64; k++) {
      if (k < wave) before += wsum[k];
      all += wsum[k];
    }
    const uint64_t base = before + incl - px;  // pixels before this segment
    if (tid < cnt) {
      const bool complete = base >= total;     // the image is complete before this segment: nothing of it is used
      par->segbase[g0 + tid] = complete ? total : (uint32_t)base;
      if (!complete) {
        uint32_t code = 0;
        if (entry != (tid ? s_exit[tid - 1] : expect) || (fl & 2u)) code = 1;  // not on the chain, or more tokens than records
        else if ((fl & 1u) && base + px < total) code = 2;                     // a token the segment kernels do not take
        if (code) atomicMin(&s_first, (tid << 2) | code);
      }
    }
    __syncthreads();
    if (s_first != 0xffffffffu) {  // the first segment that fails decides
      verdict = (s_first & 3u) == 1 ? kParRetryMaps : kParFallback;
      break;
    }
    base0 += all;
    expect = s_exit[cnt - 1];
    __syncthreads();
  }
  if (tid == 0) {
    if (verdict == kOk && base0 < total) verdict = kParFallback;  // the stream ends before the image does
    if (verdict != kOk) par_fail(par, verdict);
  }
}

// 256 threads: eight workgroups per CU, so that the chains of dependent loads each starts with (image, workspace, segment)
// overlap; a workgroup of 1024 threads per segment left two per CU (5.9 ms per 1080p batch).
constexpr uint32_t kPlaceThreads = 256;
constexpr uint32_t kPlaceTokens = 2;  // tokens per thread and round (1: 4.39 ms per 1080p batch, 2: 4.19, 4: 4.51)
__global__ __launch_bounds__(kPlaceThreads) void k_par_place(const DecImage *__restrict__ imgs, const uint32_t *__restrict__ segfirst, uint32_t n) {
  __shared__ uint32_t wsum[kPlaceTokens * (kPlaceThreads / 64)];
  static_assert(kPlaceTokens * (kPlaceThreads / 64) <= 64, "one wave scans the waves' sums");
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t img = find_image(segfirst, n, blockIdx.x), seg = blockIdx.x - segfirst[img];
  ParImage *par = imgs[img].par;
  if (!par || !par->active || par->mode != 1 || seg >= par->nseg || par->status != kOk) return;
  const uint32_t total = par->total;
  uint32_t fill = par->segbase[seg];
  if (fill >= total) return;
  const uint64_t *slot = par->segmap + (size_t)seg * 64;
  const uint32_t ntok = (uint32_t)(slot[1] >> 32);
  const uint64_t *rec = par->records + (size_t)seg * kParRecCap;
  // the lane scouts leave the records in kLanesPerSeg parts: tokens before each part
  const bool parts = slot[3] != 0;
  uint32_t before_part[kLanesPerSeg];
#pragma unroll
  for (uint32_t j = 0; j < kLanesPerSeg; j++) before_part[j] = parts ? (uint32_t)slot[4 + j] : 0u;
  // with pixel tags (ParImage::tag) the sources go into the image itself, marked by the tag byte; else into the source array, where
  // a position that holds its value is marked by kFinalBit
  const uint32_t tagword = (par->tag & 0x100u) ? (par->tag & 0xffu) << 24 : 0u;
  const bool tagged = (par->tag & 0x100u) != 0;
  uint32_t *dst = par->dst, *srcg = tagged ? par->dst : par->srcg;
  const bool fold_green = par->fold_green != 0;
  bool bad = false;
  auto record = [&](uint32_t i) -> uint64_t {  // token i of the segment
    if (i >= ntok) return 0ull;
    uint32_t at = i;
    if (parts) {
      uint32_t part = 0;
#pragma unroll
      for (uint32_t j = 1; j < kLanesPerSeg; j++) part += before_part[j] <= i ? 1u : 0u;
      uint32_t b4 = 0;
#pragma unroll
      for (uint32_t j = 1; j < kLanesPerSeg; j++) b4 = part == j ? before_part[j] : b4;
      at = part * kLaneRecCap + (i - b4);
    }
    return rec[at];
  };
  // A round places kPlaceTokens tokens per thread (token t0 + j * 256 + tid for j = 0 ...): a round costs two barriers and
  // the latency of a scan whatever it places.
  uint64_t rc_next[kPlaceTokens];
#pragma unroll
  for (uint32_t j = 0; j < kPlaceTokens; j++) rc_next[j] = record(j * kPlaceThreads + tid);
  for (uint32_t t0 = 0; t0 < ntok && fill < total; t0 += kPlaceThreads * kPlaceTokens) {
    uint32_t val[kPlaceTokens], cnt[kPlaceTokens], incl[kPlaceTokens];
    bool ref[kPlaceTokens];
#pragma unroll
    for (uint32_t j = 0; j < kPlaceTokens; j++) {
      const uint64_t rc = rc_next[j];
      rc_next[j] = record(t0 + (kPlaceTokens + j) * kPlaceThreads + tid);  // on its way while this round's tokens are placed
      val[j] = (uint32_t)rc;
      cnt[j] = (uint32_t)(rc >> 32) & 0xffu;
      ref[j] = (rc >> 40) & 1ull;
      incl[j] = wave_incl_scan_u32(cnt[j], (int)lane);
    }
    __syncthreads();
    if (lane == 63) {
#pragma unroll
      for (uint32_t j = 0; j < kPlaceTokens; j++) wsum[j * (kPlaceThreads / 64) + wave] = incl[j];
    }
    __syncthreads();
    constexpr uint32_t kSums = kPlaceTokens * (kPlaceThreads / 64);  // in token order: j-major, then the wave
    const uint32_t wsums = wave_incl_scan_u32(lane < kSums ? wsum[lane] : 0u, (int)lane);
    const uint32_t all = lane_read(wsums, kSums - 1);
#pragma unroll
    for (uint32_t j = 0; j < kPlaceTokens; j++) {
      const uint32_t idx = j * (kPlaceThreads / 64) + wave;
      const uint32_t before = idx ? lane_read(wsums, idx - 1) : 0u;
      const uint32_t pos = fill + before + incl[j] - cnt[j];
      // The first pixel of a token gets its entry here whatever the token is (a literal points at itself, a reference
      // at its source: one store for both, and the many references of one pixel are done with it); the other pixels of
      // a reference follow in the loops below.
      bool expand = false;
      if (cnt[j] && pos < total) {
        if (pos + cnt[j] > total) bad = true;        // a reference that runs past the end of the image
        else if (ref[j] && val[j] > pos) bad = true; // a reference before the start of the image
        else {
          if (!ref[j]) dst[pos] = fold_green ? add_green(val[j]) : val[j];
          if (ref[j]) srcg[pos] = tagword | (pos - val[j]);
          else if (!tagged) srcg[pos] = pos | kFinalBit;
          expand = ref[j] && cnt[j] > 1;
        }
      }
      if (expand && cnt[j] <= kWgOwnCopy) {
        if (val[j] >= cnt[j]) {
          for (uint32_t e = 1; e < cnt[j]; e++) srcg[pos + e] = tagword | (pos - val[j] + e);
        } else {
          for (uint32_t e = 1; e < cnt[j]; e++) srcg[pos + e] = tagword | (pos - val[j] + e % val[j]);
        }
      }
      uint64_t m = __ballot(expand && cnt[j] > kWgOwnCopy);
      while (m) {
        const uint32_t l = (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        const uint32_t p2 = lane_read(pos, l), n2 = lane_read(cnt[j], l), dist = lane_read(val[j], l);
        for (uint32_t c2 = 0; c2 < n2; c2 += 64) {
          const uint32_t e = c2 + lane;
          if (e < n2) srcg[p2 + e] = tagword | (p2 - dist + (dist < n2 ? e % dist : e));
        }
      }
    }
    fill += all;
  }
  if (bad) par_fail(par, kErrInvalidVp8l);
}

// sb_first / units as for the inverse transforms: a unit of kUnit pixels at a time.  The grid is a few workgroups per
// CU that stride over the units: most of the passes find nothing left to do, and a launch of one workgroup per unit
// that only looks and leaves took 0.18 ms for a gigapixel (29 of 32 passes, 5 ms of a 48 ms batch).
constexpr uint32_t kParJumpGrid = 4096;
__global__ __launch_bounds__(256) void k_par_jump(const DecImage *__restrict__ imgs, const uint32_t *__restrict__ sb_first, uint32_t n, uint32_t pass,
                                                  uint32_t units, uint32_t *__restrict__ any_open) {
  if (pass > 0 && !any_open[pass - 1]) return;  // no image of the call has an open chain left
  for (uint32_t unit = blockIdx.x; unit < units; unit += gridDim.x) {
    const uint32_t sb = unit / kUnitsPerSuper, img = find_image(sb_first, n, sb);
    ParImage *par = imgs[img].par;
    if (!par || !par->active || par->status != kOk) continue;
    if (pass > 0 && !par->changed[pass - 1]) continue;  // the pass before left nothing open
    const uint64_t p0 = (uint64_t)(unit - sb_first[img] * kUnitsPerSuper) * kUnit;
    uint32_t *srcg = par->srcg, *dst = par->dst;
    const uint32_t total = par->total;
    bool open = false;
    // A thread's pixels of the unit in groups of kJumpBatch: their chains are followed side by side (a hop is a
    // dependent read of a microsecond; one pixel after the other left a wave with one read in flight per lane).
    constexpr uint32_t kJumpBatch = 4;
    static_assert((kUnit / 256) % kJumpBatch == 0, "groups of pixels per thread");
#ifndef VP8L_JUMP_HOPS
#define VP8L_JUMP_HOPS 5
#endif
    if (par->mode == 1 && (par->tag & 0x100u)) {
      // Pixel tags (ParImage::tag): a pixel that a reference made holds tag byte << 24 | position of its source until a hop
      // finds a pixel without the tag, which is its value.  Read while its owner replaces it, a pixel is its old source, a
      // nearer one or the value: all of them lead to the same literal.
      const uint32_t tagbyte = par->tag & 0xffu;
      for (uint32_t k0 = 0; k0 < kUnit / 256; k0 += kJumpBatch) {
        uint32_t t[kJumpBatch];
        bool mine[kJumpBatch];
#pragma unroll
        for (uint32_t j = 0; j < kJumpBatch; j++) {
          const uint64_t p = p0 + (k0 + j) * 256 + threadIdx.x;
          t[j] = p < total ? dst[p] : (tagbyte ^ 0xffu) << 24;
          mine[j] = (t[j] >> 24) == tagbyte;
        }
#pragma unroll
        for (int hop = 0; hop < VP8L_JUMP_HOPS; hop++) {
#pragma unroll
          for (uint32_t j = 0; j < kJumpBatch; j++)
            if ((t[j] >> 24) == tagbyte) t[j] = __hip_atomic_load(&dst[t[j] & 0xffffffu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (uint32_t j = 0; j < kJumpBatch; j++) {
          if (!mine[j]) continue;
          const uint64_t p = p0 + (k0 + j) * 256 + threadIdx.x;
          __hip_atomic_store(&dst[p], t[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((t[j] >> 24) == tagbyte) open = true;
        }
      }
    } else
    for (uint32_t k0 = 0; k0 < kUnit / 256; k0 += kJumpBatch) {
      uint32_t t[kJumpBatch];
      bool mine[kJumpBatch];  // a pixel whose chain is still open
#pragma unroll
      for (uint32_t j = 0; j < kJumpBatch; j++) {
        const uint64_t p = p0 + (k0 + j) * 256 + threadIdx.x;
        t[j] = p < total ? srcg[p] : kFinalBit;
        mine[j] = !(t[j] & kFinalBit);
      }
      // the source's own source, several steps on; read while its owner replaces it, it is the old or the new one: both
      // lead to the same literal
#pragma unroll
      for (int hop = 0; hop < VP8L_JUMP_HOPS; hop++) {
#pragma unroll
        for (uint32_t j = 0; j < kJumpBatch; j++)
          if (!(t[j] & kFinalBit)) t[j] = __hip_atomic_load(&srcg[t[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (uint32_t j = 0; j < kJumpBatch; j++) {
        if (!mine[j]) continue;
        const uint64_t p = p0 + (k0 + j) * 256 + threadIdx.x;
        __hip_atomic_store(&srcg[p], t[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // a chain that has reached its literal (whose pixel the kernels before this one wrote) gives the pixel its value
        // here, once: the position is final from now on and no pass looks at it again
        if (t[j] & kFinalBit) dst[p] = dst[t[j] & ~kFinalBit];
        else open = true;
      }
    }
    if (__syncthreads_or(open) && threadIdx.x == 0) { par->changed[pass] = 1; any_open[pass] = 1; }
  }
}

// After everything else of the segment path: the verdict the inverse transforms and the host see.
__global__ void k_par_finish(const DecImage *__restrict__ imgs, DecResult *__restrict__ res) {
  ParImage *par = imgs[blockIdx.x].par;
  if (!par || !par->active) return;
  int32_t status = par->status;
  if (status == kOk && par->changed[kParJumpPasses - 1]) status = kParFallback;
  res[blockIdx.x].status = status;
  res[blockIdx.x].pad = 1;  // decoded by the segment kernels
}

__global__ __launch_bounds__(kWgThreads) void k_decode_entropy_wg(const DecImage *__restrict__ imgs, DecResult *__restrict__ res) {
  __shared__ DecState st;
  __shared__ GroupTables g0;
  // the ring is used while pixels are decoded, the table scratch while a code is read: never at the same time
  __shared__ union RingOrScratch {
    uint32_t ring[kWgRing];
    DecTableScratch scratch;
  } ros;
  static_assert(sizeof(DecTableScratch) <= sizeof(uint32_t) * kWgRing, "the scratch fits under the ring");
  __shared__ uint32_t win[kWgWindowWords];
  __shared__ __attribute__((aligned(16))) WgShared sh;
  __shared__ WgCacheShared cs;
  __shared__ uint32_t s_ev;
  const uint32_t img = blockIdx.x, tid = threadIdx.x;
  const DecImage im = imgs[img];
  if (res[img].pad2) return;  // k_decode_headers gave its verdict or handed the main image to the segment kernels
  DecAccel acc;
  acc.win = win;
  acc.win_bytes = kWgWindow;
  acc.ring = ros.ring;
  acc.ring_mask = kWgRing - 1;
  acc.g0 = &g0;
  acc.driver_pixels = true;
  acc.driver_multi = true;
  acc.driver_cache = true;
  acc.on = true;
  if (tid == 0) {
    dec_init(st, im.data, im.len, im.out, im.aux, im.aux_cap, &g0, im.groups, im.group_cap, im.cache, &ros.scratch, im.sorted);
    st.w = im.w;
    st.h = im.h;
  }
  __syncthreads();
  for (;;) {
    if (tid == 0) s_ev = dec_step(st, acc);
    __syncthreads();  // the request, lane 0's LDS state and its pixel stores before the others act on them
    const uint32_t ev = s_ev;
    if (ev == kEvDone) break;
    if (ev == kEvPixels) {
      if (im.par && st.dst == st.out && st.total >= kParMinPixels && st.ngroups == 1 && st.cache_bits == 0) {
        // the main image of a large stream: its pixels are decoded by the segment kernels that follow, from these tables
        ParImage *par = im.par;
        for (uint32_t i = tid; i < sizeof(GroupTables) / 4; i += kWgThreads)
          reinterpret_cast<uint32_t *>(&par->tables)[i] = reinterpret_cast<const uint32_t *>(&g0)[i];
        if (tid == 0) {
          par->P0 = st.bytepos * 8 - st.cnt;
          par->total = (uint32_t)st.total;
          par->xsize = st.xsize | (st.standard << 31);  // bit 31: the dialect, for the distance codes
          par->dst = st.dst;
          par->active = 1;
          // Subtract green is undone pixel by pixel (transform.h:94-103), and a back reference copies whole pixels: if it is the
          // first transform to undo, the segment kernels store every literal with green already added and the copies inherit it,
          // which saves a pass over the image.  (A stream the segment kernels hand back is decoded again from the start.)
          par->fold_green = st.nxf && st.xf[st.nxf - 1].type == 2 ? 1u : 0u;
          if (par->fold_green) st.nxf--;
          par->tag = par_pixel_tag(g0.t[kAlpha].single, st.total);
          st.status = kParPending;
        }
        __syncthreads();
        break;
      }
      wg_decode_pixels((VP8L_LDS DecState *)&st, (VP8L_LDS GroupTables *)&g0, (VP8L_LDS uint32_t *)ros.ring,
                       (VP8L_LDS uint32_t *)win, (VP8L_LDS WgShared *)&sh, (VP8L_LDS WgCacheShared *)&cs, im, tid);
    } else if (ev == kEvRefill) {
      const uint64_t lo = st.bytepos;
      wg_window_load((VP8L_LDS uint32_t *)win, im, lo, tid);
      if (tid == 0) { st.win_lo = lo; st.win_hi = lo + kWgWindow; st.win_final = lo + kWgWindow >= im.len; }
    } else {
      __threadfence_block();
      uint32_t *dst = st.dst;
      const uint64_t pos = st.copy_pos;
      const uint32_t dist = st.copy_dist, len = st.copy_len;
      // pixel i of the copy equals pixel (i mod dist) of the source window: correct for overlapping
      // references too, and every source is already final (in HBM: lane 0 stores every pixel there too)
      for (uint32_t i = tid; i < len; i += kWgThreads) {
        const uint32_t v = dst[pos - dist + (dist >= len ? i : i % dist)];
        dst[pos + i] = v;
        if (i + kWgRing >= len) ros.ring[(uint32_t)(pos + i) & (kWgRing - 1)] = v;
      }
      __threadfence_block();
    }
    __syncthreads();
  }
  if (tid == 0) {
    DecResult &r = res[img];
    r.status = st.status;
    r.need_groups = st.need_groups;
    r.nxf = st.nxf;
    r.standard = st.standard;
    for (uint32_t k = 0; k < 4; k++) r.xf[k] = st.xf[k];
  }
}

// Calls whose streams go to the segment kernels start with this one: a wave per stream reads the header and the tables (and
// decodes the transforms' sub-images), and hands the main image over exactly as k_decode_entropy_wg would.  A workgroup of
// 1024 threads per stream is one stream per CU, and parsing a header takes a lane about 0.75 ms: 512 streams took two
// rounds of that, 3584 small ones fourteen; a wave per stream lets seven share a CU.  Streams it does not hand over
// (small images, several Huffman groups, a colour cache) are left untouched for k_decode_entropy_wg, which skips the others.
// small_too: images too small for the segment kernels are decoded here to the end, as k_decode_entropy would (a call with
// thousands of small streams: seven of them share a CU here, one at a time has it in the workgroup form; 4096 streams of
// 64x64 pixels 16.8 -> 4.5 ms).
__global__ __launch_bounds__(64) void k_decode_headers(const DecImage *__restrict__ imgs, DecResult *__restrict__ res, uint32_t small_too) {
  __shared__ DecState st;
  __shared__ GroupTables g0;
  __shared__ union RingOrScratch {
    uint32_t ring[kRingPixels];
    DecTableScratch scratch;
  } ros;
  __shared__ uint32_t win[kStreamWindow / 4];
  const uint32_t img = blockIdx.x, lane = threadIdx.x;
  const DecImage im = imgs[img];
  const bool small = (uint64_t)im.w * im.h < kParMinPixels;  // too small for the segment kernels whatever it holds
  if (small ? !small_too : !im.par) return;
  DecAccel acc;
  acc.win = win;
  acc.win_bytes = kStreamWindow;
  acc.ring = ros.ring;
  acc.ring_mask = kRingPixels - 1;
  acc.g0 = &g0;
  acc.driver_pixels = true;
  acc.stop_at_main = !small;
  acc.on = true;
  if (lane == 0) {
    dec_init(st, im.data, im.len, im.out, im.aux, im.aux_cap, &g0, im.groups, im.group_cap, im.cache, &ros.scratch, im.sorted);
    st.w = im.w;
    st.h = im.h;
  }
  __syncthreads();
  bool finished = false;  // a verdict (an error), or the main image handed over: k_decode_entropy_wg passes the stream by
  for (;;) {
    uint32_t ev = kEvDone;
    if (lane == 0) ev = dec_step(st, acc);
    ev = (uint32_t)__shfl((int)ev, 0, 64);
    if (ev == kEvDone) { finished = small || st.status != kOk; break; }  // a small image is done; a large one only gets here with an error (its main image stops the loop)
    __syncthreads();
    if (ev == kEvPixels) {
      if (!small && st.dst == st.out) {
        if (st.total >= kParMinPixels && st.ngroups == 1 && st.cache_bits == 0) {
          ParImage *par = im.par;
          for (uint32_t i = lane; i < sizeof(GroupTables) / 4; i += 64)
            reinterpret_cast<uint32_t *>(&par->tables)[i] = reinterpret_cast<const uint32_t *>(&g0)[i];
          if (lane == 0) {
            par->P0 = st.bytepos * 8 - st.cnt;
            par->total = (uint32_t)st.total;
            par->xsize = st.xsize | (st.standard << 31);
            par->dst = st.dst;
            par->active = 1;
            par->fold_green = st.nxf && st.xf[st.nxf - 1].type == 2 ? 1u : 0u;  // see k_decode_entropy_wg
            if (par->fold_green) st.nxf--;
            par->tag = par_pixel_tag(g0.t[kAlpha].single, st.total);
            st.status = kParPending;
          }
          __syncthreads();
          finished = true;
        }
        break;
      }
      wave_decode_pixels((VP8L_LDS DecState *)&st, (const VP8L_LDS GroupTables *)&g0, (VP8L_LDS uint32_t *)ros.ring,
                         (VP8L_LDS uint32_t *)win, im, lane);
    } else if (ev == kEvRefill) {
      const uint64_t lo = st.bytepos;
      window_load(win, im, lo, lane);
      if (lane == 0) { st.win_lo = lo; st.win_hi = lo + kStreamWindow; st.win_final = lo + kStreamWindow >= im.len; }
    } else {
      __threadfence_block();
      uint32_t *dst = st.dst;
      const uint64_t pos = st.copy_pos;
      const uint32_t dist = st.copy_dist, len = st.copy_len;
      for (uint32_t i = lane; i < len; i += 64) {
        const uint32_t v = dst[pos - dist + (dist >= len ? i : i % dist)];
        dst[pos + i] = v;
        if (i + kRingPixels >= len) ros.ring[(uint32_t)(pos + i) & (kRingPixels - 1)] = v;
      }
      __threadfence_block();
    }
    __syncthreads();
  }
  if (lane == 0 && finished) {
    DecResult &r = res[img];
    r.status = st.status;
    r.need_groups = st.need_groups;
    r.nxf = st.nxf;
    r.standard = st.standard;
    r.pad2 = 1;  // done with: the workgroup form passes it by
    for (uint32_t k = 0; k < 4; k++) r.xf[k] = st.xf[k];
  }
}

// Which transform this pass undoes for an image: pass k undoes the (k+1)-th from the end.
__device__ __forceinline__ bool pass_transform(const DecResult &r, uint32_t pass, XformRec &x) {
  if (r.status != 0 || pass >= r.nxf) return false;
  x = r.xf[r.nxf - 1 - pass];
  return true;
}

// What the four passes have to do over the whole call: kinds[pass] has kXfStash / kXfPointwise / kXfPredictor set if
// some image needs that kernel in that pass.  Most calls need one pass (subtract green); the launches of the other
// passes, a workgroup per 4096 pixels that looks up its image and leaves, took 1.3 ms per gigapixel.
constexpr uint32_t kXfStash = 1, kXfPointwise = 2, kXfPredictor = 4;
__global__ __launch_bounds__(256) void k_pass_kinds(const DecResult *__restrict__ res, uint32_t n, uint32_t *__restrict__ kinds) {
  const uint32_t img = blockIdx.x * 256 + threadIdx.x;
  if (img >= n) return;
  for (uint32_t pass = 0; pass < 4; pass++) {
    XformRec x;
    if (!pass_transform(res[img], pass, x)) continue;
    const uint32_t k = x.type == 0 ? kXfPredictor : (x.type == 3 && x.bits != 0 ? (kXfStash | kXfPointwise) : kXfPointwise);
    if ((kinds[pass] & k) != k) atomicOr(&kinds[pass], k);
  }
}

// Colour indexing expands bundled indices in place, so the indices are set aside first
// (the reference copies the whole image, transform.h:62).
__global__ __launch_bounds__(256) void k_inverse_stash(const DecImage *__restrict__ imgs, const DecResult *__restrict__ res,
                                                       const uint32_t *__restrict__ sb_first, uint32_t n, uint32_t pass,
                                                       const uint32_t *__restrict__ kinds, uint32_t units) {
  if (!(kinds[pass] & kXfStash)) return;
  for (uint32_t unit = blockIdx.x; unit < units; unit += gridDim.x) {  // the grid is smaller than `units` for the later passes
    const uint32_t sb = unit / kUnitsPerSuper, img = find_image(sb_first, n, sb);
    XformRec x;
    if (!pass_transform(res[img], pass, x) || x.type != 3 || x.bits == 0) continue;
    const DecImage im = imgs[img];
    const uint64_t cnt = (uint64_t)((im.w + (1u << x.bits) - 1) >> x.bits) * im.h;
    const uint64_t p0 = (uint64_t)(unit - sb_first[img] * kUnitsPerSuper) * kUnit;
    for (uint32_t k = 0; k < kUnit / 256; k++) {
      const uint64_t p = p0 + k * 256 + threadIdx.x;
      if (p < cnt) im.scratch[p] = im.out[p];
    }
  }
}

__device__ __forceinline__ void inverse_pointwise_unit(const DecImage *__restrict__ imgs, const DecResult *__restrict__ res,
                                                       const uint32_t *__restrict__ sb_first, uint32_t n, uint32_t pass, uint32_t unit,
                                                       uint32_t *pal) {
  const uint32_t sb = unit / kUnitsPerSuper, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  XformRec x;
  if (!pass_transform(res[img], pass, x) || x.type == 0) return;
  const DecImage im = imgs[img];
  const uint64_t npx = (uint64_t)im.w * im.h;
  const uint64_t p0 = (uint64_t)(unit - sb_first[img] * kUnitsPerSuper) * kUnit;
  if (p0 >= npx) return;
  const uint32_t *td = im.aux + x.data_off;
  if (x.type == 2) {  // transform.h:94-103
    for (uint32_t k = 0; k < kUnit / 256; k++) {
      const uint64_t p = p0 + k * 256 + tid;
      if (p < npx) im.out[p] = add_green(im.out[p]);
    }
  } else if (x.type == 1) {  // transform.h:196-218
    for (uint32_t k = 0; k < kUnit / 256; k++) {
      const uint64_t p = p0 + k * 256 + tid;
      if (p >= npx) break;
      const uint32_t y = (uint32_t)(p / x.img_w), xx = (uint32_t)(p - (uint64_t)y * x.img_w);
      if (y >= im.h) break;
      im.out[p] = inverse_cross_color(im.out[p], td[(y >> x.bits) * x.xsize + (xx >> x.bits)]);
    }
  } else {  // transform.h:59-87; only `width` pixels per row are produced (the reference overruns the row)
    const uint32_t ncolors = x.xsize;
    __syncthreads();  // the unit before this one may still be reading its palette
    if (tid < ncolors) pal[tid] = td[tid];
    __syncthreads();
    const uint32_t bits = x.bits, per = 1u << bits, bpp = 8u >> bits, mask = (1u << bpp) - 1u;
    const uint32_t cix = (im.w + per - 1) >> bits;
    const uint32_t *idx = bits ? im.scratch : im.out;
    for (uint32_t k = 0; k < kUnit / 256; k++) {
      const uint64_t p = p0 + k * 256 + tid;
      if (p >= npx) break;
      const uint32_t y = (uint32_t)(p / im.w), xx = (uint32_t)(p - (uint64_t)y * im.w);
      const uint32_t g = (idx[(uint64_t)y * cix + (xx >> bits)] >> 8) & 0xff;
      const uint32_t c = (g >> ((xx & (per - 1)) * bpp)) & mask;
      im.out[p] = c < ncolors ? pal[c] : 0u;  // the reference does not range-check the index (transform.h:82)
    }
  }
}

__global__ __launch_bounds__(256) void k_inverse_pointwise(const DecImage *__restrict__ imgs, const DecResult *__restrict__ res,
                                                           const uint32_t *__restrict__ sb_first, uint32_t n, uint32_t pass,
                                                           const uint32_t *__restrict__ kinds, uint32_t units) {
  __shared__ uint32_t pal[kMaxPalette];
  if (!(kinds[pass] & kXfPointwise)) return;
  // the grid is smaller than `units` for the later passes, which most calls do not need
  for (uint32_t unit = blockIdx.x; unit < units; unit += gridDim.x) inverse_pointwise_unit(imgs, res, sb_first, n, pass, unit, pal);
}

// Predictor inverse (transform.h:110-191).  Pixel (x,y) needs L, T, TL and TR = (x+1, y-1), so row y
// may run two pixels behind row y-1: thread r of the block handles row y0+r and at step t the pixel
// x = t - 2r.  One barrier per step; a band of `blockDim` rows at a time.
// Nothing on the step-to-step path touches HBM: the pixel a thread makes is handed to the thread below through
// LDS (the row above's pixel x+1 is exactly what that thread made one step earlier; T and TL are the two hand-overs
// before), residuals are fetched eight steps ahead into registers and staged in an LDS ring, the tile's mode is
// fetched when the previous tile starts.  Only the finished pixels are stored to HBM.
constexpr uint32_t kPredRows = 1024;
__global__ __launch_bounds__(kPredRows) void k_inverse_predictor(const DecImage *__restrict__ imgs, DecResult *__restrict__ res,
                                                                  uint32_t pass) {
  __shared__ uint32_t hand[2][kPredRows + 1];  // hand[step parity][r + 1] = the pixel thread r made at that step; [0] = row above the band
  __shared__ uint32_t ring[8][kPredRows];      // residuals of the current eight steps
  __shared__ uint32_t upring[8];               // thread 0: pixels of the row above the band
  const uint32_t img = blockIdx.x, r = threadIdx.x;
  XformRec xf;
  if (!pass_transform(res[img], pass, xf) || xf.type != 0) return;
  const DecImage im = imgs[img];
  const uint32_t w = xf.img_w, h = im.h, bits = xf.bits, tmask = (1u << bits) - 1u;
  const bool standard = res[img].standard != 0;
  const uint32_t *modes = im.aux + xf.data_off;
  uint32_t *out = im.out;
  bool bad = false;
  for (uint32_t y0 = 0; y0 < h; y0 += kPredRows) {
    const uint32_t y = y0 + r;
    const bool live = y < h;
    uint32_t *row = out + (uint64_t)(live ? y : 0) * w;
    const uint32_t *up = out + (uint64_t)(y0 ? y0 - 1 : 0) * w;  // thread 0 of a later band: the last row of the band before
    const uint32_t *mrow = modes + (live ? (y >> bits) : 0u) * xf.xsize;
    uint32_t L = 0, T = 0, TL = 0, row0 = 0;  // row0: the row's first pixel (the standard dialect's top-right of the last column)
    if (r == 0 && y0) T = up[0];
    const uint32_t rows_here = h - y0 < kPredRows ? h - y0 : kPredRows;
    const uint32_t steps = w + 2 * (rows_here - 1);
    // residuals (and, for thread 0, the row above) of the eight steps after the current eight
    uint32_t nxt[8], upn[8];
    auto fetch = [&](uint32_t t8) {  // for steps t8 .. t8 + 7
#pragma unroll
      for (uint32_t k = 0; k < 8; k++) {
        const int x = (int)(t8 + k) - 2 * (int)r;
        nxt[k] = live && x >= 0 && x < (int)w ? row[x] : 0u;
        upn[k] = r == 0 && y0 && t8 + k + 1 < w ? up[t8 + k + 1] : 0u;
      }
    };
    fetch(0);
    uint32_t mode = live ? (mrow[0] >> 8) & 0xff : 0u, mode_next = mode;
    __syncthreads();  // the band before is done with the LDS arrays
    for (uint32_t t = 0; t < steps; t++) {
      if ((t & 7u) == 0) {
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) ring[k][r] = nxt[k];
        if (r == 0) {
#pragma unroll
          for (uint32_t k = 0; k < 8; k++) upring[k] = upn[k];
        }
        fetch(t + 8);  // a thread reads back only what it staged itself: no barrier
      }
      const int x = (int)t - 2 * (int)r;
      // what the thread above made one step ago: its pixel x + 1 (thread 0: the row above the band)
      const uint32_t above = r ? hand[(t + 1) & 1u][r] : upring[t & 7u];
      if (live && x == -1) T = above;  // pixel 0 of the row above
      if (live && x >= 0 && x < (int)w) {
        uint32_t P;
        if (y == 0) {
          P = x == 0 ? 0xff000000u : L;
        } else if (x == 0) {
          P = T;
          TL = T;
          T = above;
        } else {
          if (((uint32_t)x & tmask) == 0) mode = mode_next;
          const uint32_t TR = (x == (int)w - 1) ? (standard ? row0 : L) : above;  // transform.h:137; standard: the next pixel in memory
          if (mode > 13) bad = true;
          P = predict(mode, L, T, TR, TL);
          TL = T;
          T = TR;
        }
        // the mode of the next tile, fetched when this one starts
        if (((uint32_t)x & tmask) == 0 && (((uint32_t)x >> bits) + 1) < xf.xsize) mode_next = (mrow[((uint32_t)x >> bits) + 1] >> 8) & 0xff;
        L = px_add(ring[t & 7u][r], P);
        if (x == 0) row0 = L;
        row[x] = L;
        hand[t & 1u][r + 1] = L;
      }
      __syncthreads();
    }
    __threadfence_block();  // the next band's thread 0 reads this band's last row from HBM
  }
  if (bad) atomicExch(&res[img].status, (int32_t)kErrInvalidVp8l);
}

// The same inverse spread over the device: one CU runs out of instruction issue (the 14 modes diverge within a wave),
// so an image is cut into bands of 64 rows, one wave each, that run as a pipeline: a band's first row needs the row
// above it, which the band before publishes as it goes: that row's pixels and a progress counter are stored and read
// past the caches (device-scope release/acquire would write back and invalidate whole L2s every few steps).  A band therefore trails the one before by about 140 steps.  Workgroups are
// dispatched in index order and band b waits only for band b-1 of the same image, which has the lower index, so
// the wait cannot deadlock; it is bounded all the same, and an image whose wait ran out is decoded again by the
// one-workgroup kernel (status kPredRetry).
constexpr uint32_t kBandRows = 64;
constexpr int32_t kPredRetry = 4;  // DecResult.status value that is not a verdict
__global__ __launch_bounds__(64) void k_inverse_predictor_bands(const DecImage *__restrict__ imgs, DecResult *__restrict__ res, uint32_t pass,
                                                                uint32_t nimg, uint32_t max_bands, uint32_t *__restrict__ progress) {
  __shared__ uint32_t ring[8][kBandRows];
  // workgroups go round-robin to the 8 XCDs: the bands of an image stay on one XCD (one L2 between neighbours), eight
  // images side by side; band b-1 of the image is workgroup blockIdx - 8
  const uint32_t col = blockIdx.x % 8, j = blockIdx.x / 8, img = (j / max_bands) * 8 + col, band = j % max_bands, r = threadIdx.x;
  if (img >= nimg) return;
  XformRec xf;
  if (!pass_transform(res[img], pass, xf) || xf.type != 0) return;
  const DecImage im = imgs[img];
  const uint32_t w = xf.img_w, h = im.h, bits = xf.bits, tmask = (1u << bits) - 1u;
  const bool standard = res[img].standard != 0;
  const uint32_t y0 = band * kBandRows;
  if (y0 >= h) return;
  const uint32_t *modes = im.aux + xf.data_off;
  uint32_t *out = im.out;
  uint32_t *mine = progress + (size_t)img * max_bands + band;            // pixels of this band's last row that are final
  const uint32_t *theirs = band ? mine - 1 : nullptr;                    // the same of the band above
  const uint32_t y = y0 + r;
  const bool live = y < h;
  uint32_t *row = out + (uint64_t)(live ? y : 0) * w;
  const uint32_t *up = out + (uint64_t)(y0 ? y0 - 1 : 0) * w;
  const uint32_t *mrow = modes + (live ? (y >> bits) : 0u) * xf.xsize;
  const uint32_t rows_here = h - y0 < kBandRows ? h - y0 : kBandRows;
  const uint32_t steps = w + 2 * (rows_here - 1);
  bool bad = false, late = false;
  // waits until the band above has finished `need` pixels of its last row
  uint32_t seen = 0;  // what the band above had finished when it was last asked (it only grows: no need to ask again below that)
  auto wait_for = [&](uint32_t need) {
    if (!theirs || late) return;
    need = need < w ? need : w;
    if (seen >= need) return;  // a read past the caches takes a microsecond or two, and this is asked every eight steps
    for (uint32_t spin = 0;; spin++) {
      // relaxed, past the caches: the row above is read the same way, so no cache has to be invalidated for it
      const uint32_t have = uni(__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (have == 0xffffffffu || spin > (1u << 22)) { late = true; return; }  // the band above gave up, or is not coming
      seen = have;
      if (have >= need) return;
      __builtin_amdgcn_s_sleep(8);
    }
  };
  uint32_t nxt[8], upn = 0;  // residuals of the eight steps after the current eight; lane k < 8 of upn: the row above for them
  auto fetch = [&](uint32_t t8) {
    wait_for(t8 + 9);
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      const int x = (int)(t8 + k) - 2 * (int)r;
      nxt[k] = live && x >= 0 && x < (int)w ? row[x] : 0u;
    }
    upn = 0;
    if (y0 && !late && r < 8 && t8 + r + 1 < w) upn = __hip_atomic_load(&up[t8 + r + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  uint32_t L = 0, T = 0, TL = 0, made = 0, upcur = 0, row0 = 0;
  if (y0) {
    wait_for(1);
    if (!late) T = __hip_atomic_load(&up[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // used by lane 0 only
  }
  fetch(0);
  uint32_t mode = live ? (mrow[0] >> 8) & 0xff : 0u, mode_next = mode;
  for (uint32_t t = 0; t < steps && !late; t++) {
    if ((t & 7u) == 0) {
#pragma unroll
      for (uint32_t k = 0; k < 8; k++) ring[k][r] = nxt[k];
      upcur = upn;
      // what the last row has finished so far goes out before the next fetch can wait on the band above
      if (r == rows_here - 1) {
        const int xdone = (int)t - 2 * (int)r;  // pixels 0 .. xdone-1 of this row are stored
        if (xdone > 0) {
          // the row's pixels were stored past the caches; once they have arrived (no cache write-back needed) the count
          // follows.  The wait is spelled out: a workgroup-scope release need not wait for stores to global memory.
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(mine, (uint32_t)xdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      fetch(t + 8);
    }
    const int x = (int)t - 2 * (int)r;
    // what the lane above made one step ago: its pixel x + 1 (lane 0: the row above the band)
    uint32_t above = (uint32_t)__shfl_up((int)made, 1, 64);
    if (r == 0) above = lane_read(upcur, t & 7u);
    if (live && x == -1) T = above;
    if (live && x >= 0 && x < (int)w) {
      uint32_t P;
      if (y == 0) {
        P = x == 0 ? 0xff000000u : L;
      } else if (x == 0) {
        P = T;
        TL = T;
        T = above;
      } else {
        if (((uint32_t)x & tmask) == 0) mode = mode_next;
        const uint32_t TR = (x == (int)w - 1) ? (standard ? row0 : L) : above;  // transform.h:137; standard: the next pixel in memory
        if (mode > 13) bad = true;
        P = predict(mode, L, T, TR, TL);
        TL = T;
        T = TR;
      }
      if (((uint32_t)x & tmask) == 0 && (((uint32_t)x >> bits) + 1) < xf.xsize) mode_next = (mrow[((uint32_t)x >> bits) + 1] >> 8) & 0xff;
      L = px_add(ring[t & 7u][r], P);
      if (x == 0) row0 = L;
      if (r == rows_here - 1) __hip_atomic_store(&row[x], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the band below reads this row
      else row[x] = L;
      made = L;
    }
  }
  if (r == rows_here - 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row's last pixels have arrived before the count says so
    __hip_atomic_store(mine, late ? 0xffffffffu : w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (late) atomicExch(&res[img].status, kPredRetry);
  else if (bad) atomicCAS(&res[img].status, (int32_t)kOk, (int32_t)kErrInvalidVp8l);
}

}  // namespace vp8l

// =============================================================================================
// Host side: batch decoder context
// =============================================================================================
using namespace vp8l;

struct vp8l_decoder {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::mutex mu;
  DeviceBuffer d_imgs, d_res, d_sbfirst, d_aux, d_scratch, d_groups, d_cache, d_sorted;
  DeviceBuffer d_par, d_srcg, d_segmap, d_segidx, d_segfirst, d_lanefirst, d_jumpflags, d_records;  // the segment kernels' workspace
  DeviceBuffer d_progress;  // the banded predictor inverse: pixels finished per band
  DeviceBuffer d_kinds;     // which inverse-transform kernels each of the four passes needs
  double last_seconds = 0;
  int threads_per_stream = 0;  // 0: chosen per call
  int whole_device = 0;        // 0: chosen per call, 1: never, 2: whenever the stream qualifies
  uint32_t last_whole_device = 0;  // streams of the last call that the segment kernels decoded
  uint32_t last_scouted = 0;       // ... of them, with the scouts' token records (the others needed the entry maps)
  int scout_form = 0;          // 0: a lane per piece of a segment (k_par_scout_lanes), 1: a token at every bit offset (k_par_scout)
};

extern "C" int vp8l_decoder_set_option(vp8l_decoder *d, int option, int value) {
  if (!d) return VP8L_ERR_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lock(d->mu);
  if (option == VP8L_DEC_OPT_THREADS_PER_STREAM && (value == 0 || value == 64 || value == (int)kWgThreads)) {
    d->threads_per_stream = value;
    return VP8L_OK;
  }
  if (option == VP8L_DEC_OPT_WHOLE_DEVICE && value >= 0 && value <= 2) {
    d->whole_device = value;
    return VP8L_OK;
  }
  if (option == VP8L_DEC_OPT_SCOUT_FORM && (value == 0 || value == 1)) {
    d->scout_form = value;
    return VP8L_OK;
  }
  return VP8L_ERR_BAD_ARGUMENT;
}

extern "C" int vp8l_decoder_create(int device, vp8l_decoder **out) {
  if (!out) return VP8L_ERR_BAD_ARGUMENT;
  int rc = select_device(device, &device);
  if (rc != VP8L_OK) return rc;
  vp8l_decoder *d = new (std::nothrow) vp8l_decoder;
  if (!d) return VP8L_ERR_NO_MEMORY;
  d->device = device;
  if (!hip_ok(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking), "hipStreamCreate") ||
      !hip_ok(hipEventCreate(&d->ev0), "hipEventCreate") || !hip_ok(hipEventCreate(&d->ev1), "hipEventCreate")) {
    vp8l_decoder_destroy(d);
    return VP8L_ERR_HIP;
  }
  *out = d;
  return VP8L_OK;
}

extern "C" void vp8l_decoder_destroy(vp8l_decoder *d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  if (d->ev0) (void)hipEventDestroy(d->ev0);
  if (d->ev1) (void)hipEventDestroy(d->ev1);
  DeviceBuffer *bufs[] = {&d->d_imgs, &d->d_res, &d->d_sbfirst, &d->d_aux, &d->d_scratch, &d->d_groups, &d->d_cache, &d->d_sorted,
                          &d->d_par, &d->d_srcg, &d->d_segmap, &d->d_segidx, &d->d_segfirst, &d->d_lanefirst, &d->d_jumpflags, &d->d_records, &d->d_progress, &d->d_kinds};
  for (auto *b : bufs) b->release();
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
}

namespace {

// One pass over the images listed in `idx` with `group_cap` multi-group table sets per image.
int decode_pass(vp8l_decoder *d, const std::vector<uint32_t> &idx, const uint8_t *const *d_data, const size_t *len,
                const uint32_t *width, const uint32_t *height, uint32_t *const *d_argb, uint32_t group_cap, int whole_mode, bool pred_bands,
                std::vector<DecResult> &results) {
  const bool whole_device = whole_mode != 0;  // 1: scouts and records, 2: entry maps
  const uint32_t m = (uint32_t)idx.size();
  std::vector<DecImage> imgs(m);
  std::vector<uint32_t> sb_first(m + 1, 0), seg_first(m + 1, 0), lane_first(m + 1, 0);
  std::vector<ParImage> pars(whole_device ? m : 0);
  uint64_t aux_total = 0, scratch_total = 0, sb_total = 0, px_total = 0, seg_total = 0;
  for (uint32_t k = 0; k < m; k++) {
    const uint32_t i = idx[k], w = width[i], h = height[i];
    DecImage &im = imgs[k];
    im.data = d_data[i];
    im.len = len[i];
    im.w = w;
    im.h = h;
    im.out = d_argb[i];
    im.aux_cap = 3ull * ((w + 3) / 4) * ((h + 3) / 4) + 256;
    im.aux = reinterpret_cast<uint32_t *>(aux_total);       // offsets for now, bases added below
    im.scratch = reinterpret_cast<uint32_t *>(scratch_total);
    im.group_cap = group_cap;
    aux_total += (im.aux_cap + 63) & ~63ull;
    scratch_total += ((uint64_t)((w + 1) / 2) * h + 63) & ~63ull;
    sb_first[k] = (uint32_t)sb_total;
    sb_total += ((uint64_t)w * h + kSuper - 1) / kSuper;
    if (sb_total > 0x3fffffffu) return VP8L_ERR_NO_MEMORY;
    im.par = nullptr;
    if (whole_device) {
      // segments that can hold tokens: the stream's bits, and one more for the bits read as zero past its end
      const uint64_t nseg = ((uint64_t)len[i] * 8 + kParSegBits - 1) / kParSegBits + 1;
      memset(&pars[k], 0, sizeof(ParImage));
      pars[k].nseg = (uint32_t)nseg;
      pars[k].mode = (uint32_t)whole_mode;
      pars[k].records = reinterpret_cast<uint64_t *>(seg_total * kParRecCap);
      pars[k].srcg = reinterpret_cast<uint32_t *>(px_total);  // offsets for now
      pars[k].segmap = reinterpret_cast<uint64_t *>(seg_total * 64);
      pars[k].segentry = reinterpret_cast<uint32_t *>((seg_total + k) * 2);
      seg_first[k] = (uint32_t)seg_total;
      lane_first[k + 1] = lane_first[k] + (uint32_t)((nseg + kLaneSegsPerWg - 1) / kLaneSegsPerWg);
      px_total += ((uint64_t)w * h + 63) & ~63ull;
      seg_total += nseg;
      if (seg_total > 0x3fffffffu) return VP8L_ERR_NO_MEMORY;
    }
  }
  sb_first[m] = (uint32_t)sb_total;
  seg_first[m] = (uint32_t)seg_total;
  // Thousands of small streams: a wave per stream gets through them faster than a workgroup per stream (see k_decode_headers).
  uint32_t small_streams = 0;
  for (uint32_t k = 0; k < m; k++) small_streams += (uint64_t)imgs[k].w * imgs[k].h < kParMinPixels ? 1u : 0u;
  const bool many_small = d->threads_per_stream == 0 && small_streams >= 1024;
  uint32_t max_bands = 1;
  for (uint32_t k = 0; k < m; k++) max_bands = std::max(max_bands, (height[idx[k]] + kBandRows - 1) / kBandRows);
  const bool ok = d->d_imgs.reserve(m * sizeof(DecImage)) && d->d_res.reserve(m * sizeof(DecResult)) &&
                  d->d_progress.reserve((size_t)m * max_bands * 4) && d->d_kinds.reserve(4 * 4) &&
                  d->d_sbfirst.reserve((m + 1) * 4) && d->d_aux.reserve(aux_total * 4) &&
                  d->d_scratch.reserve(scratch_total * 4) &&
                  d->d_groups.reserve((size_t)m * group_cap * sizeof(GroupTables) + 16) &&
                  d->d_cache.reserve((size_t)m * (4u << kMaxCacheBits)) &&
                  d->d_sorted.reserve((size_t)m * (group_cap + 1) * kSortedPerGroup * 2);
  if (!ok) return VP8L_ERR_NO_MEMORY;
  if (whole_device) {
    if (!d->d_par.reserve(m * sizeof(ParImage)) || !d->d_srcg.reserve(px_total * 4) || !d->d_segmap.reserve(seg_total * 64 * 8) ||
        !d->d_segidx.reserve((seg_total + m) * 2 * 4) || !d->d_segfirst.reserve((m + 1) * 4) || !d->d_lanefirst.reserve((m + 1) * 4) || !d->d_jumpflags.reserve(kParJumpPasses * 4) ||
        (whole_mode == 1 && !d->d_records.reserve(seg_total * kParRecCap * 8)))
      return VP8L_ERR_NO_MEMORY;
    // the scouts store records 16 bytes at a time: the array must start at a multiple of 16 (see kLaneRecCap)
    if (whole_mode == 1 && (reinterpret_cast<uintptr_t>(d->d_records.as<uint64_t>()) & 15u) != 0) return VP8L_ERR_HIP;
    for (uint32_t k = 0; k < m; k++) {
      pars[k].records = whole_mode == 1 ? d->d_records.as<uint64_t>() + reinterpret_cast<uintptr_t>(pars[k].records) : nullptr;
      pars[k].srcg = d->d_srcg.as<uint32_t>() + reinterpret_cast<uintptr_t>(pars[k].srcg);
      pars[k].segmap = d->d_segmap.as<uint64_t>() + reinterpret_cast<uintptr_t>(pars[k].segmap);
      pars[k].segentry = d->d_segidx.as<uint32_t>() + reinterpret_cast<uintptr_t>(pars[k].segentry);
      pars[k].segbase = pars[k].segentry + pars[k].nseg + 1;
      imgs[k].par = d->d_par.as<ParImage>() + k;
    }
  }
  for (uint32_t k = 0; k < m; k++) {
    imgs[k].aux = d->d_aux.as<uint32_t>() + reinterpret_cast<uintptr_t>(imgs[k].aux);
    imgs[k].scratch = d->d_scratch.as<uint32_t>() + reinterpret_cast<uintptr_t>(imgs[k].scratch);
    imgs[k].groups = d->d_groups.as<GroupTables>() + (size_t)k * group_cap;
    imgs[k].cache = d->d_cache.as<uint32_t>() + ((size_t)k << kMaxCacheBits);
    imgs[k].sorted = d->d_sorted.as<uint16_t>() + (size_t)k * (group_cap + 1) * kSortedPerGroup;
  }
  hipStream_t s = d->stream;
  auto *dimgs = d->d_imgs.as<DecImage>();
  auto *dres = d->d_res.as<DecResult>();
  auto *sbf = d->d_sbfirst.as<uint32_t>();
  if (!hip_ok(hipMemcpyAsync(dimgs, imgs.data(), m * sizeof(DecImage), hipMemcpyHostToDevice, s), "upload images") ||
      !hip_ok(hipMemcpyAsync(sbf, sb_first.data(), (m + 1) * 4, hipMemcpyHostToDevice, s), "upload table") ||
      !hip_ok(hipMemsetAsync(dres, 0, m * sizeof(DecResult), s), "clear results"))
    return VP8L_ERR_HIP;
  if (whole_device &&
      (!hip_ok(hipMemcpyAsync(d->d_par.ptr, pars.data(), m * sizeof(ParImage), hipMemcpyHostToDevice, s), "upload segment workspace") ||
       !hip_ok(hipMemcpyAsync(d->d_segfirst.ptr, seg_first.data(), (m + 1) * 4, hipMemcpyHostToDevice, s), "upload segment table") ||
       !hip_ok(hipMemcpyAsync(d->d_lanefirst.ptr, lane_first.data(), (m + 1) * 4, hipMemcpyHostToDevice, s), "upload piece table") ||
       !hip_ok(hipMemsetAsync(d->d_jumpflags.ptr, 0, kParJumpPasses * 4, s), "clear pass flags")))
    return VP8L_ERR_HIP;
  const uint32_t gunits = (uint32_t)(sb_total * kUnitsPerSuper);
  (void)hipEventRecord(d->ev0, s);
  // A workgroup per stream decodes one stream several times faster than a wave per stream; the waves only win
  // with thousands of small streams of noisy content, so they are an option, not the default.
  if (d->threads_per_stream == 64) hipLaunchKernelGGL(k_decode_entropy, dim3(m), dim3(64), 0, s, dimgs, dres);
  else {
    if (whole_device || many_small) hipLaunchKernelGGL(k_decode_headers, dim3(m), dim3(64), 0, s, dimgs, dres, many_small ? 1u : 0u);
    hipLaunchKernelGGL(k_decode_entropy_wg, dim3(m), dim3(kWgThreads), 0, s, dimgs, dres);
  }
  if (whole_device) {
    // streams whose main image k_decode_entropy_wg left to the segment kernels (the others are finished: these kernels pass them by)
    const uint32_t *segf = d->d_segfirst.as<uint32_t>();
    if (whole_mode == 1) {
      if (d->scout_form == 1) hipLaunchKernelGGL(k_par_scout, dim3((uint32_t)seg_total), dim3(kWgThreads), 0, s, dimgs, segf, m);
      else hipLaunchKernelGGL(k_par_scout_lanes, dim3(lane_first[m]), dim3(kLaneThreads), 0, s, dimgs, d->d_lanefirst.as<uint32_t>(), m);
      hipLaunchKernelGGL(k_par_verify, dim3(m), dim3(256), 0, s, dimgs);
      hipLaunchKernelGGL(k_par_place, dim3((uint32_t)seg_total), dim3(kPlaceThreads), 0, s, dimgs, segf, m);
    } else {
      hipLaunchKernelGGL(k_par_maps, dim3((uint32_t)seg_total), dim3(kWgThreads), 0, s, dimgs, segf, m);
      hipLaunchKernelGGL(k_par_chain, dim3(m), dim3(256), 0, s, dimgs);
      hipLaunchKernelGGL(k_par_pixels, dim3((uint32_t)seg_total), dim3(kWgThreads), 0, s, dimgs, segf, m);
    }
    for (uint32_t pass = 0; pass < kParJumpPasses; pass++)
      hipLaunchKernelGGL(k_par_jump, dim3(std::min(gunits, kParJumpGrid)), dim3(256), 0, s, dimgs, sbf, m, pass, gunits, d->d_jumpflags.as<uint32_t>());
    hipLaunchKernelGGL(k_par_finish, dim3(m), dim3(1), 0, s, dimgs, dres);
  }
  uint32_t *kinds = d->d_kinds.as<uint32_t>();
  (void)hipMemsetAsync(kinds, 0, 4 * 4, s);
  hipLaunchKernelGGL(k_pass_kinds, dim3((m + 255) / 256), dim3(256), 0, s, dres, m, kinds);
  for (uint32_t pass = 0; pass < 4; pass++) {
    // the first pass has work in nearly every call; a later pass rarely, so it gets a grid that costs little to launch in vain
    const uint32_t grid = pass == 0 ? gunits : std::min(gunits, 8192u);
    hipLaunchKernelGGL(k_inverse_stash, dim3(grid), dim3(256), 0, s, dimgs, dres, sbf, m, pass, kinds, gunits);
    hipLaunchKernelGGL(k_inverse_pointwise, dim3(grid), dim3(256), 0, s, dimgs, dres, sbf, m, pass, kinds, gunits);
    if (pred_bands) {
      (void)hipMemsetAsync(d->d_progress.ptr, 0, (size_t)m * max_bands * 4, s);
      // 36 KiB of LDS nobody uses: at most four of these one-wave workgroups then share a CU (one per SIMD); left to
      // itself the dispatcher packs dozens onto a few CUs and they take turns on the same SIMDs
      hipLaunchKernelGGL(k_inverse_predictor_bands, dim3((m + 7) / 8 * 8 * max_bands), dim3(64), 36 << 10, s, dimgs, dres, pass, m, max_bands,
                         d->d_progress.as<uint32_t>());
    } else {
      hipLaunchKernelGGL(k_inverse_predictor, dim3(m), dim3(kPredRows), 0, s, dimgs, dres, pass);
    }
  }
  (void)hipEventRecord(d->ev1, s);
  if (!hip_ok(hipGetLastError(), "kernel launch")) return VP8L_ERR_HIP;
  results.resize(m);
  if (!hip_ok(hipMemcpyAsync(results.data(), dres, m * sizeof(DecResult), hipMemcpyDeviceToHost, s), "download results") ||
      !hip_ok(hipStreamSynchronize(s), "decode batch"))
    return VP8L_ERR_HIP;
  float ms = 0;
  (void)hipEventElapsedTime(&ms, d->ev0, d->ev1);
  d->last_seconds += ms * 1e-3;
  return VP8L_OK;
}

}  // namespace

static int vp8l_decoder_decode_device_impl(vp8l_decoder *d, size_t n, const uint8_t *const *d_data, const size_t *len,
                                          const uint32_t *width, const uint32_t *height, uint32_t *const *d_argb, int *status) {
  if (!d || !d_data || !len || !width || !height || !d_argb || !status) return VP8L_ERR_BAD_ARGUMENT;
  if (n == 0) return VP8L_OK;
  if (n > 0x7fffffffu) return VP8L_ERR_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lock(d->mu);
  if (!hip_ok(hipSetDevice(d->device), "hipSetDevice")) return VP8L_ERR_HIP;
  d->last_seconds = 0;
  std::vector<uint32_t> pending;
  for (size_t i = 0; i < n; i++) {
    status[i] = VP8L_OK;
    if (!d_data[i] || !d_argb[i] || width[i] == 0 || height[i] == 0 || width[i] > kMaxDim || height[i] > kMaxDim ||
        ((uintptr_t)d_data[i] & 3)) {
      status[i] = VP8L_ERR_BAD_ARGUMENT;
      continue;
    }
    pending.push_back((uint32_t)i);
  }
  // The segment kernels (streams spread over the device) decode faster than a workgroup per stream for any number of
  // streams (6.7 against 4.2 GP/s photo-like at 256 streams, 3 ms against 500 for one 4K stream); they need enough
  // stream bytes to make segments worth launching, and 16 bytes of token records per stream byte, so a large batch
  // goes through them in passes of at most 2 GiB of streams.
  if (pending.empty()) return VP8L_OK;
  uint64_t stream_bytes = 0;
  for (uint32_t i : pending) stream_bytes += len[i];
  // ... or streams large enough on average that cutting them up beats a workgroup each: one 512 x 512 photo-like stream (88 KB)
  // takes 1.5 ms that way against 3.5 ms, a 1024 x 1024 one 2.3 against 8.2; the fixed cost of the segment kernels' launches
  // is what a 35 KB stream takes in one workgroup
  const bool large_on_average = stream_bytes >= (48u << 10) && stream_bytes / pending.size() >= (32u << 10);
  const bool whole_device = d->threads_per_stream != 64 &&
                            (d->whole_device == 2 || (d->whole_device == 0 && (stream_bytes >= (2u << 20) || large_on_average)));
  d->last_whole_device = 0;
  d->last_scouted = 0;
  // Passes: first everything (whole-device streams with scouts), then again what a pass handed on: streams whose
  // scouts were not in step (entry maps), streams the segment kernels cannot take (one workgroup), streams with more
  // Huffman groups than the pass provided.
  struct Todo { std::vector<uint32_t> idx; int mode; uint32_t group_cap; bool pred_bands; };
  std::vector<Todo> todo;
  if (whole_device) {
    constexpr uint64_t kPassBytes = 2ull << 30;
    std::vector<std::vector<uint32_t>> parts(1);
    uint64_t acc = 0;
    for (uint32_t i : pending) {
      if (!parts.back().empty() && acc + len[i] > kPassBytes) { parts.emplace_back(); acc = 0; }
      parts.back().push_back(i);
      acc += len[i];
    }
    for (auto it = parts.rbegin(); it != parts.rend(); ++it) todo.push_back({std::move(*it), 1, 8, true});  // popped from the back
  } else {
    todo.push_back({pending, 0, 8, true});
  }
  while (!todo.empty()) {
    Todo t = std::move(todo.back());
    todo.pop_back();
    std::vector<DecResult> results;
    const int rc = decode_pass(d, t.idx, d_data, len, width, height, d_argb, t.group_cap, t.mode, t.pred_bands, results);
    if (t.group_cap > 64) {  // a stream with thousands of groups sized these tables: not kept for the calls that follow
      d->d_groups.release();
      d->d_sorted.release();
    }
    if (rc == VP8L_ERR_NO_MEMORY) {
      // no room for this pass's workspace: half the streams at a time; a single stream first without the segment kernels'
      // workspace (one workgroup), and if even that does not fit, the stream alone gets the verdict, not the call
      (void)hipGetLastError();  // the failed allocation is dealt with here; the next pass checks for launch errors
      if (t.idx.size() > 1) {
        const size_t half = t.idx.size() / 2;
        todo.push_back({std::vector<uint32_t>(t.idx.begin() + half, t.idx.end()), t.mode, t.group_cap, t.pred_bands});
        todo.push_back({std::vector<uint32_t>(t.idx.begin(), t.idx.begin() + half), t.mode, t.group_cap, t.pred_bands});
      } else if (t.mode != 0) {
        todo.push_back({t.idx, 0, t.group_cap, t.pred_bands});
      } else {
        status[t.idx[0]] = VP8L_ERR_NO_MEMORY;
      }
      continue;
    }
    if (rc != VP8L_OK) return rc;
    std::map<uint32_t, std::vector<uint32_t>> more_groups;
    Todo with_maps{{}, 2, t.group_cap, t.pred_bands}, one_workgroup{{}, 0, t.group_cap, t.pred_bands},
        one_block_predictor{{}, t.mode, t.group_cap, false};
    for (size_t k = 0; k < t.idx.size(); k++) {
      const DecResult &r = results[k];
      if (r.status == kErrWorkspace && r.need_groups > t.group_cap) {
        // a pass sizes every image's tables by one count: images that need more go on together with those whose need rounds up to the
        // same power of two (so one stream with thousands of groups sizes only its own pass)
        uint32_t cap = 8;
        while (cap < r.need_groups) cap <<= 1;
        more_groups[cap].push_back(t.idx[k]);
      } else if (r.status == kPredRetry) {
        one_block_predictor.idx.push_back(t.idx[k]);  // a band of the predictor inverse waited too long for the band above
      } else if (r.status == kParRetryMaps && t.mode == 1) {
        with_maps.idx.push_back(t.idx[k]);
      } else if (r.status == kParFallback || r.status == kParPending || r.status == kParRetryMaps) {
        one_workgroup.idx.push_back(t.idx[k]);
      } else {
        status[t.idx[k]] = r.status;
        if (t.mode && r.pad == 1) d->last_whole_device++;
        if (t.mode == 1 && r.pad == 1) d->last_scouted++;
      }
    }
    for (auto &mg : more_groups) todo.push_back({std::move(mg.second), t.mode, mg.first, t.pred_bands});
    for (Todo *n2 : {&with_maps, &one_workgroup, &one_block_predictor})
      if (!n2->idx.empty()) todo.push_back(std::move(*n2));
  }
  return VP8L_OK;
}

extern "C" int vp8l_decoder_decode_device(vp8l_decoder *d, size_t n, const uint8_t *const *d_data, const size_t *len,
                                          const uint32_t *width, const uint32_t *height, uint32_t *const *d_argb, int *status) {
  return (int)guarded([&] { return (int)vp8l_decoder_decode_device_impl(d, n, d_data, len, width, height, d_argb, status); });
}

extern "C" long vp8l_decoder_last_whole_device_streams(const vp8l_decoder *d) { return d ? (long)d->last_whole_device : 0; }

extern "C" long vp8l_decoder_last_scout_streams(const vp8l_decoder *d) { return d ? (long)d->last_scouted : 0; }

extern "C" double vp8l_decoder_last_device_seconds(const vp8l_decoder *d) { return d ? d->last_seconds : 0.0; }

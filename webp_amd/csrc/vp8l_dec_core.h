// vp8l_dec_core.h -- the sequential part of the decoder: stream grammar, Huffman table
// construction and the entropy-decode loop, written as a RESUMABLE state machine.
//
// One bitstream is one serial dependency chain, so on the GPU one lane per image drives
// `dec_step()` out of LDS while the other 63 lanes of its wave wait; whenever the chain
// produces a back reference the machine returns kEvCopy and the whole wave performs the copy
// (k_decode_entropy in vp8l_decode.hip).  The same source compiles for the host so the
// grammar and table logic can be tested without a GPU.  The per-pixel inverse transforms are
// separate data-parallel kernels; their scalar pixel functions live at the end of this file.
#pragma once
#include <string.h>

#include "vp8l_common.h"

namespace vp8l {

enum : int {
  kOk = 0,
  kErrUnsupportedVp8 = -4,  // UnsupportedVP8  vp8l.h:87-88
  kErrInvalidVp8l = -5,     // InvalidVP8L     vp8l.h:107-108,236-237; transform.h:185-186
  kErrInvalidHuffman = -6,  // InvalidHuffman  huffman_io.h:105,140,184; huffman_coding.h:162
  kErrWorkspace = -11       // more Huffman groups / tile data than the workspace holds
};

// ---- bit reader (webp/utils/bit_readed.h:27-63; bytes past the end read as zero) -----------
struct BitSrc {
  const uint8_t *data;
  uint64_t len, bytepos, buf;
  uint32_t cnt;
  // Optional copy of the stream bytes [win_lo, win_hi) as little-endian words (the device decoder keeps
  // one in LDS; bytes past `len` are zero in it).  bytepos, win_lo and win_hi are multiples of 4.
  const uint32_t *win;
  uint64_t win_lo, win_hi;
  VP8L_HD void init(const uint8_t *d, uint64_t l) { data = d; len = l; bytepos = 0; buf = 0; cnt = 0; win = nullptr; win_lo = win_hi = 0; }
  VP8L_HD uint32_t load32(uint64_t pos) const {
    if (pos >= win_lo && pos + 4 <= win_hi) return win[(pos - win_lo) >> 2];
    uint32_t x = 0;
    if (pos + 4 <= len) {
      x = (uint32_t)data[pos] | ((uint32_t)data[pos + 1] << 8) | ((uint32_t)data[pos + 2] << 16) | ((uint32_t)data[pos + 3] << 24);
    } else {
      for (int i = 0; i < 4; i++) if (pos + i < len) x |= (uint32_t)data[pos + i] << (8 * i);
    }
    return x;
  }
  VP8L_HD void fill() {
    if (cnt < 32) { buf |= (uint64_t)load32(bytepos) << cnt; cnt += 32; bytepos += 4; }
  }
  VP8L_HD uint32_t peek32() { fill(); return (uint32_t)buf; }
  VP8L_HD void skip(uint32_t n) { buf >>= n; cnt -= n; }
  VP8L_HD uint32_t get(uint32_t n) {  // n <= 32
    if (n == 0) return 0;
    fill();
    const uint32_t v = (uint32_t)(buf & (n >= 32 ? 0xffffffffull : ((1ull << n) - 1ull)));
    skip(n);
    return v;
  }
};

// ---- Huffman decode tables -------------------------------------------------------------------
// The reference walks a node tree one bit at a time (huffman_io.h:85-91); the symbol that walk
// reaches is the canonical-code symbol, so a table gives the same result: codes of up to
// kLutBits bits resolve in one lookup, longer ones by the per-length counts.
constexpr int kLutBits = 10;
constexpr uint16_t kLongCode = 0xffff;
constexpr int kClLutBits = 7;  // lookup bits of the code-length code's table
struct HuffTable {
  uint16_t lut[1 << kLutBits];  // (symbol << 4) | length; kLongCode = longer than kLutBits; a one-leaf code has length 0
  uint16_t count[kSaneCodeLen + 1];
  uint16_t index[kSaneCodeLen + 1];  // symbols with shorter codes (position of the first code of this length in `sorted`)
  uint32_t first[kSaneCodeLen + 1];  // first canonical code of each length
  uint16_t single;              // the symbol of a one-leaf code (it consumes no bits), else 0xffff
  uint16_t maxlen;
  uint32_t sorted_off;          // where this code's (length, symbol)-ordered symbols start in `sorted`
};

constexpr int kSortedPerGroup = kMaxAlphabet + 3 * 256 + kAlphabetDist;
struct GroupTables {
  HuffTable t[kNumCodes];
  uint16_t *sorted;  // kSortedPerGroup entries; only codes longer than the lookup table read it, so it may live in slow memory
};

VP8L_HD uint32_t dec_bit_reverse(uint32_t v, int len) {  // 1 <= len <= 32
#if defined(__clang__)
  return __builtin_bitreverse32(v) >> (32 - len);
#else
  uint32_t r = 0;
  for (int i = 0; i < len; i++) r |= ((v >> i) & 1u) << (len - 1 - i);
  return r;
#endif
}

struct TableScratch {
  uint16_t off[kSaneCodeLen + 2];
  uint32_t next[kSaneCodeLen + 2];
};

// From code lengths (dec::HuffmanTree::implicit_init, huffman_coding.h:188-232).  That
// constructor accepts exactly the complete codes: every symbol gets its canonical code, a
// collision or a missing leaf fails add_symbol / is_full.  A single used symbol becomes the root
// whatever its length.  Lengths above 32 overflow the reference's 32-bit codes; rejected here.
// lut_bits: index bits of the lookup table that gets filled (the code-length code, read a few hundred times, gets a small one:
// filling 1024 entries cost more than its symbols take to read).
VP8L_HD int build_table(HuffTable &t, uint16_t *sorted, const uint8_t *lengths, int n, TableScratch &s, int lut_bits = kLutBits) {
  for (int l = 0; l <= kSaneCodeLen; l++) t.count[l] = 0;
  int used = 0, last = 0, maxl = 0;
  for (int i = 0; i < n; i++) {
    const int l = lengths[i];
    if (!l) continue;
    if (l > kSaneCodeLen) return kErrInvalidHuffman;
    t.count[l]++;
    used++;
    last = i;
    if (l > maxl) maxl = l;
  }
  t.single = 0xffff;
  t.maxlen = (uint16_t)maxl;
  if (used == 0) return kErrInvalidHuffman;
  if (used == 1) {
    t.single = (uint16_t)last;
    for (int i = 0; i < (1 << lut_bits); i++) t.lut[i] = (uint16_t)(last << 4);
    return kOk;
  }
  uint64_t kraft = 0;
  for (int l = 1; l <= maxl; l++) kraft += (uint64_t)t.count[l] << (32 - l);
  if (kraft != (1ull << 32)) return kErrInvalidHuffman;
  uint32_t code = 0;
  s.off[1] = 0;
  s.next[0] = 0;
  for (int l = 1; l <= maxl; l++) {
    code = (code + (l > 1 ? t.count[l - 1] : 0)) << 1;
    s.next[l] = code;
    s.off[l + 1] = (uint16_t)(s.off[l] + t.count[l]);
    t.first[l] = code;
    t.index[l] = s.off[l];
  }
  // a complete code without long codes fills every entry below
  if (maxl > lut_bits)
    for (int i = 0; i < (1 << lut_bits); i++) t.lut[i] = kLongCode;
  for (int i = 0; i < n; i++) {
    const int l = lengths[i];
    if (!l) continue;
    sorted[s.off[l]++] = (uint16_t)i;
    const uint32_t c = s.next[l]++;
    if (l <= lut_bits) {
      const uint32_t r = dec_bit_reverse(c, l);
      for (uint32_t k = r; k < (1u << lut_bits); k += 1u << l) t.lut[k] = (uint16_t)((i << 4) | l);
    }
  }
  return kOk;
}

// Simple code (huffman_io.h:152-178, dec::HuffmanTree::explicit_init huffman_coding.h:167-187).
VP8L_HD int build_simple(HuffTable &t, uint16_t *sorted, int nsym, uint32_t s0, uint32_t s1, int alphabet) {
  if ((int)s0 >= alphabet || (nsym == 2 && (int)s1 >= alphabet)) return kErrInvalidHuffman;
  for (int l = 0; l <= kSaneCodeLen; l++) t.count[l] = 0;
  t.single = 0xffff;
  if (nsym == 1) {
    t.single = (uint16_t)s0;
    t.maxlen = 0;
    for (int i = 0; i < (1 << kLutBits); i++) t.lut[i] = (uint16_t)(s0 << 4);
    return kOk;
  }
  t.maxlen = 1;
  t.count[1] = 2;
  t.first[1] = 0;
  t.index[1] = 0;
  sorted[0] = (uint16_t)s0;
  sorted[1] = (uint16_t)s1;
  for (int i = 0; i < (1 << kLutBits); i++) t.lut[i] = (uint16_t)((((i & 1) ? s1 : s0) << 4) | 1);
  return kOk;
}

VP8L_HD uint32_t read_sym(BitSrc &br, const HuffTable &t, const uint16_t *sorted, int lut_bits = kLutBits) {
  const uint32_t bits = br.peek32();
  const uint32_t e = t.lut[bits & ((1u << lut_bits) - 1u)];
  if (e != kLongCode) { br.skip(e & 15u); return e >> 4; }  // length 0: the symbol of a one-leaf code, no bits read
  uint32_t code = 0, first = 0, index = 0;
  for (int l = 1; l <= (int)t.maxlen; l++) {
    code |= (bits >> (l - 1)) & 1u;
    const uint32_t c = t.count[l];
    if (code < first + c) { br.skip((uint32_t)l); return sorted[t.sorted_off + index + (code - first)]; }
    index += c;
    first = (first + c) << 1;
    code <<= 1;
  }
  br.skip(t.maxlen);
  return 0;  // unreachable for a complete code
}

// ---- state machine ---------------------------------------------------------------------------
enum : uint32_t { kEvDone = 0, kEvCopy = 1, kEvRefill = 2, kEvPixels = 3 };
enum : uint32_t {
  kPhStart = 0, kPhXform, kPhImgBegin, kPhPixels, kPhAfterCopy, kPhPaletteFix, kPhMain, kPhMetaDone,
  kPhGroups, kPhDone
};

struct XformRec {
  uint32_t type, bits, xsize, ysize;
  uint64_t data_off;  // offset of the tile data / palette in the aux buffer (uint32 units)
  uint32_t img_w;     // width of the image the transform works on: the image's, or (standard dialect, after a colour-indexing
  uint32_t pad;       // transform that bundles pixels) the bundled width; the reference always uses the image's (vp8l.h:117-126)
};

// What reading one code needs (huffman_io.h:92-195): the code-length code and the lengths being assembled.
struct DecTableScratch {
  TableScratch ts;
  HuffTable cl;
  uint16_t cl_sorted[kNumClSyms];
  uint8_t lengths[kMaxAlphabet];
};

struct DecState {
  // set by the caller
  const uint8_t *data;
  uint64_t len;
  uint32_t *out;          // width*height pixels: the entropy decoder's target and the final image
  uint32_t *aux;          // tile images, palette, entropy image
  uint64_t aux_cap;
  GroupTables *groups;    // tables in use: `groups1` for sub-images and one-group images, else `groups_multi`
  GroupTables *groups1;   // one table set (LDS on the device)
  GroupTables *groups_multi;
  uint32_t multi_cap;
  // bit reader
  uint64_t bytepos, buf;
  uint32_t cnt;
  // image level
  uint32_t w, h, nxf, seen, cix, main_cache_bits;
  uint32_t standard;      // the stream is in the standard dialect (its first 32 bits, the chunk size, are not zero)
  XformRec xf[4];
  uint64_t aux_used;
  // the coded image being read
  uint32_t phase, after;
  uint32_t *dst;
  uint32_t xsize, ysize, x, y, cache_bits, ngroups, meta_bits, meta_xsize;
  uint32_t *meta;
  uint64_t total, fill;
  // pending back reference (kEvCopy): copy `copy_len` pixels to dst[copy_pos..] from copy_dist back
  uint64_t copy_pos;
  uint32_t copy_dist, copy_len;
  int32_t status;
  uint32_t need_groups;
  // stream window kept by the driver (see DecAccel); win_final: it reaches the end of the stream
  uint64_t win_lo, win_hi;
  uint32_t win_final;
  uint32_t *cache;        // 1 << kMaxCacheBits entries
  uint16_t *sorted_multi; // multi_cap * kSortedPerGroup entries for groups_multi
  DecTableScratch *tsc;   // only used while a code is being read: may share memory with anything the pixel phases own
};

// What the device driver adds around the machine; all null on the host.
//   win   copy of win_bytes stream bytes from st.win_lo on, refilled by the driver on kEvRefill
//   ring  the last ring_mask+1 pixels of the coded image being read (index = position & ring_mask):
//         short back references are resolved from it without leaving the machine
//   g0    st.groups1, for images with one Huffman group
struct DecAccel {
  const uint32_t *win = nullptr;
  uint32_t win_bytes = 0;
  uint32_t *ring = nullptr;
  uint32_t ring_mask = 0;
  const GroupTables *g0 = nullptr;
  // the driver decodes the pixels of one-group, cache-less coded images itself (kEvPixels): it must
  // leave st.fill = st.total, the bit reader state and st.phase = st.after (or a failure) behind
  bool driver_pixels = false;
  bool driver_multi = false;  // ... also those with several Huffman groups
  bool driver_cache = false;  // ... also those with a colour cache
  bool stop_at_main = false;  // kEvPixels for the main image whatever it holds (a driver that only wants to know what comes)
  bool on = false;  // win, ring and g0 are set (tested instead of the pointers)
};
constexpr uint32_t kInlineCopyMax = 16;   // longer copies go to the driver (kEvCopy)
constexpr uint32_t kWinPixelSlack = 64;   // bytes one token can need at most, with room to spare
constexpr uint32_t kWinPhaseSlack = 1024;

VP8L_HD void dec_init(DecState &st, const uint8_t *data, uint64_t len, uint32_t *out, uint32_t *aux, uint64_t aux_cap,
                      GroupTables *groups1, GroupTables *groups_multi, uint32_t multi_cap, uint32_t *cache,
                      DecTableScratch *tsc, uint16_t *sorted /* (multi_cap + 1) * kSortedPerGroup */) {
  st.cache = cache; st.win_lo = st.win_hi = 0; st.win_final = 0;
  st.tsc = tsc;
  groups1->sorted = sorted;
  st.sorted_multi = sorted + kSortedPerGroup;
  st.data = data; st.len = len; st.out = out; st.aux = aux; st.aux_cap = aux_cap;
  st.groups = groups1; st.groups1 = groups1; st.groups_multi = groups_multi; st.multi_cap = multi_cap;
  st.bytepos = 0; st.buf = 0; st.cnt = 0;
  st.w = st.h = st.nxf = st.seen = st.cix = st.main_cache_bits = 0;
  st.standard = 0;
  st.aux_used = 0;
  st.phase = kPhStart; st.after = kPhDone;
  st.status = kOk; st.need_groups = 0;
}

// One code description -> table (huffman_io.h:147-195, read_code_length :92-146).
VP8L_HD int read_code(BitSrc &br, int alphabet, HuffTable &t, uint16_t *sorted, DecState &st) {
  if (br.get(1)) {
    const int nsym = (int)br.get(1) + 1;
    const uint32_t first8 = br.get(1);
    const uint32_t s0 = br.get(1 + 7 * first8);
    const uint32_t s1 = nsym == 2 ? br.get(8) : 0;
    return build_simple(t, sorted + t.sorted_off, nsym, s0, s1, alphabet);
  }
  DecTableScratch &sc = *st.tsc;
  const Dialect dl = dialect_of(st.standard != 0);
  uint8_t *cl_len = sc.lengths;  // first used for the code-length-code lengths
  for (int i = 0; i < kNumClSyms; i++) cl_len[i] = 0;
  const uint32_t ncodes = br.get((uint32_t)dl.count_bits) + 4;
  if (ncodes > (uint32_t)dl.ncl) return kErrInvalidHuffman;
  for (uint32_t i = 0; i < ncodes; i++) cl_len[dl.order((int)i)] = (uint8_t)br.get((uint32_t)dl.len_bits);
  sc.cl.sorted_off = 0;
  int rc = build_table(sc.cl, sc.cl_sorted, cl_len, dl.ncl, sc.ts, kClLutBits);
  if (rc < 0) return rc;
  uint32_t max_symbol;
  if (br.get(1)) {
    const uint32_t nb = 2 + 2 * br.get(3);
    max_symbol = (2 + br.get(nb)) & 0xffffu;  // the reference keeps it in a uint16_t (huffman_io.h:95,104)
    if (max_symbol > (uint32_t)alphabet) return kErrInvalidHuffman;
  } else {
    max_symbol = (uint32_t)alphabet;
  }
  uint8_t *lengths = sc.lengths;
  for (int i = 0; i < alphabet; i++) lengths[i] = 0;
  int sym = 0, prev = 8;
  while (sym < alphabet) {
    if (max_symbol-- == 0) break;
    const uint32_t c = read_sym(br, sc.cl, sc.cl_sorted, kClLutBits);
    if (c < (uint32_t)dl.repeat) {
      lengths[sym++] = (uint8_t)c;
      if (c) prev = (int)c;
    } else {
      int reps;
      int v = 0;
      if (c == (uint32_t)dl.repeat) { reps = (int)br.get(2) + 3; v = prev; }
      else if (c == (uint32_t)dl.zero11) reps = (int)br.get(3) + 3;
      else reps = (int)br.get(7) + 11;
      if (sym + reps > alphabet) return kErrInvalidHuffman;
      for (int i = 0; i < reps; i++) lengths[sym++] = (uint8_t)v;
    }
  }
  return build_table(t, sorted + t.sorted_off, lengths, alphabet, sc.ts);
}

// Five codes of one group (huffman_io.h:200-210).
VP8L_HD int read_group(BitSrc &br, GroupTables &g, uint32_t cache_size, DecState &st) {
  uint32_t off = 0;
  for (int c = 0; c < kNumCodes; c++) {
    const int alphabet = alphabet_size(c) + (c == kGreen ? (int)cache_size : 0);
    g.t[c].sorted_off = off;
    const int rc = read_code(br, alphabet, g.t[c], g.sorted, st);
    if (rc < 0) return rc;
    off += (uint32_t)alphabet;
  }
  return kOk;
}

VP8L_HD int read_cache_bits(BitSrc &br, uint32_t &bits) {  // vp8l.h:230-241
  bits = 0;
  if (br.get(1)) {
    bits = br.get(4);
    if (bits > kMaxCacheBits) return kErrInvalidVp8l;
  }
  return kOk;
}

VP8L_HD uint64_t div_round_up(uint64_t a, uint32_t bits) { return (a + (1ull << bits) - 1) >> bits; }

VP8L_HD bool aux_alloc(DecState &st, uint64_t n, uint64_t &off) {
  if (st.aux_used + n > st.aux_cap) return false;
  off = st.aux_used;
  st.aux_used += n;
  return true;
}

VP8L_HD void begin_sub_image(DecState &st, uint32_t *dst, uint32_t xsize, uint32_t ysize, uint32_t after) {
  st.dst = dst; st.xsize = xsize; st.ysize = ysize; st.after = after; st.phase = kPhImgBegin;
}

// The pixels of one coded image (ReadLZ77CodedImage vp8l.h:263-328) from st.fill on.
enum : uint32_t { kPxFinished = 0, kPxCopy, kPxRefill, kPxFail };
template <bool kOneGroup>
VP8L_HD uint32_t decode_pixels(DecState &st, BitSrc &br, const DecAccel &acc) {
  uint64_t fill = st.fill;
  const uint64_t total = st.total;
  uint32_t x = st.x, y = st.y;
  const uint32_t xsize = st.xsize, cb = st.cache_bits;
  uint32_t *dst = st.dst;
  uint32_t *cache = st.cache;
  uint32_t *ring = acc.ring;
  const uint32_t rmask = acc.ring_mask;
  const bool refillable = acc.on && !st.win_final;
  const GroupTables *g = acc.on ? acc.g0 : st.groups;
  const GroupTables *groups = st.groups;
  const uint32_t *meta = st.meta;
  const uint32_t meta_bits = st.meta_bits, meta_xsize = st.meta_xsize;
  uint32_t rc = kPxFinished;
  while (fill < total) {
    if (refillable && br.bytepos + kWinPixelSlack > br.win_hi) { rc = kPxRefill; break; }
    if (!kOneGroup) g = &groups[meta[(y >> meta_bits) * meta_xsize + (x >> meta_bits)]];
    const uint32_t S = read_sym(br, g->t[kGreen], g->sorted);
    uint32_t px;
    if (S < 256) {
      const uint32_t r = read_sym(br, g->t[kRed], g->sorted);
      const uint32_t b = read_sym(br, g->t[kBlue], g->sorted);
      const uint32_t a = read_sym(br, g->t[kAlpha], g->sorted);
      px = (a << 24) + (r << 16) + (S << 8) + b;
    } else if (S < 280) {
      uint32_t base, nb;
      prefix_decode_base(S - 256, base, nb);
      const uint32_t length = base + br.get(nb);
      prefix_decode_base(read_sym(br, g->t[kDist], g->sorted), base, nb);
      const uint32_t dcode = base + br.get(nb);
      const uint32_t dist = code_to_dist(xsize, dcode, st.standard != 0);
      // The reference copies without checking (vp8l.h:298-300); a reference before the start
      // or past the end of the image is reported instead of reading/writing out of bounds.
      if (dist > fill || fill + length > total) { st.status = kErrInvalidVp8l; rc = kPxFail; break; }
      if (acc.on && length <= kInlineCopyMax && dist <= rmask + 1) {
        for (uint32_t i = 0; i < length; i++) {  // in order: an overlapping reference repeats its own output
          const uint32_t v = ring[(uint32_t)(fill - dist) & rmask];
          ring[(uint32_t)fill & rmask] = v;
          dst[fill++] = v;
          if (cb) cache[cache_key(v, cb)] = v;
        }
        x += length;
        while (x >= xsize) { x -= xsize; y++; }
        continue;
      }
      st.copy_pos = fill; st.copy_dist = dist; st.copy_len = length;
      rc = kPxCopy;
      break;
    } else {
      px = cache[S - 280];
    }
    if (acc.on) ring[(uint32_t)fill & rmask] = px;
    dst[fill++] = px;
    // The reference inserts lazily (vp8l.h:286-288,308-310,315-324) but always in pixel order and
    // always before the next lookup, which is what inserting every pixel at once gives.
    if (cb) cache[cache_key(px, cb)] = px;
    if (++x >= xsize) { x = 0; y++; }
  }
  st.fill = fill; st.x = x; st.y = y;
  return rc;
}

// Runs the machine until it needs its driver (kEvCopy: a back reference to perform; kEvRefill: move
// the stream window to st.bytepos) or finishes (kEvDone; st.status says how).
VP8L_HD uint32_t dec_step(DecState &st, const DecAccel &acc = DecAccel()) {
  BitSrc br;
  br.data = st.data; br.len = st.len; br.bytepos = st.bytepos; br.buf = st.buf; br.cnt = st.cnt;
  br.win = acc.win; br.win_lo = st.win_lo; br.win_hi = acc.on ? st.win_hi : 0;
#define VP8L_SAVE_BR() do { st.bytepos = br.bytepos; st.buf = br.buf; st.cnt = br.cnt; } while (0)
#define VP8L_FAIL(code) do { st.status = (code); st.phase = kPhDone; VP8L_SAVE_BR(); return kEvDone; } while (0)
  for (;;) {
    if (acc.on && !st.win_final && st.phase != kPhDone && br.bytepos + kWinPhaseSlack > br.win_hi) { VP8L_SAVE_BR(); return kEvRefill; }
    switch (st.phase) {
      case kPhStart: {  // ReadInfo vp8l.h:83-94
        st.standard = br.get(32) != 0 ? 1u : 0u;  // the reference ignores these bits (vp8l.h:85) and never writes anything but zero there
        if (br.get(8) != 0x2f) VP8L_FAIL(kErrUnsupportedVp8);
        const uint32_t w = br.get(14) + 1, h = br.get(14) + 1;
        (void)br.get(1);
        (void)br.get(3);
        if (w != st.w || h != st.h) {
          // the caller sized `out` from the same 9 header bytes; a mismatch means it was not given this stream
          if (st.w || st.h) VP8L_FAIL(kErrInvalidVp8l);
          st.w = w; st.h = h;
        }
        st.phase = kPhXform;
        break;
      }
      case kPhXform: {  // vp8l.h:339-340, ReadTransform :101-154
        if (!br.get(1)) { st.phase = kPhMain; break; }
        const uint32_t type = br.get(2);
        if (st.seen & (1u << type)) VP8L_FAIL(kErrInvalidVp8l);
        st.seen |= 1u << type;
        XformRec &x = st.xf[st.nxf++];
        x.type = type; x.bits = 0; x.xsize = 0; x.ysize = 0; x.data_off = 0; x.pad = 0;
        x.img_w = st.standard && st.cix ? st.cix : st.w;
        if (type == 0 || type == 1) {
          x.bits = br.get(3) + 2;
          x.xsize = (uint32_t)div_round_up(x.img_w, x.bits);
          x.ysize = (uint32_t)div_round_up(st.h, x.bits);
          if (!aux_alloc(st, (uint64_t)x.xsize * x.ysize, x.data_off)) VP8L_FAIL(kErrWorkspace);
          begin_sub_image(st, st.aux + x.data_off, x.xsize, x.ysize, kPhXform);
        } else if (type == 3) {
          const uint32_t ncolors = br.get(8) + 1;
          x.bits = palette_bits(ncolors);
          x.xsize = ncolors;
          x.ysize = 1;
          st.cix = (uint32_t)div_round_up(st.w, x.bits);
          if (!aux_alloc(st, 256, x.data_off)) VP8L_FAIL(kErrWorkspace);
          begin_sub_image(st, st.aux + x.data_off, ncolors, 1, kPhPaletteFix);
        }
        break;
      }
      case kPhImgBegin: {  // ReadEntropyCodedImage vp8l.h:161-173
        uint32_t cb;
        int rc = read_cache_bits(br, cb);
        if (rc < 0) VP8L_FAIL(rc);
        rc = read_group(br, st.groups[0], cb ? 1u << cb : 0u, st);
        if (rc < 0) VP8L_FAIL(rc);
        st.cache_bits = cb;
        st.ngroups = 1;
        st.meta = nullptr;
        st.total = (uint64_t)st.xsize * st.ysize;
        st.fill = 0; st.x = 0; st.y = 0;
        if (cb) for (uint32_t i = 0; i < (1u << cb); i++) st.cache[i] = 0;
        st.phase = kPhPixels;
        break;
      }
      case kPhPixels: {
        if (acc.driver_pixels && st.fill == 0 && st.total < 0x7fffffffu &&
            (((st.ngroups == 1 || acc.driver_multi) && (st.cache_bits == 0 || acc.driver_cache)) || (acc.stop_at_main && st.dst == st.out))) {
          VP8L_SAVE_BR();
          return kEvPixels;
        }
        const uint32_t rc = st.ngroups == 1 ? decode_pixels<true>(st, br, acc) : decode_pixels<false>(st, br, acc);
        if (rc == kPxFail) { st.phase = kPhDone; VP8L_SAVE_BR(); return kEvDone; }
        if (rc == kPxCopy) { st.phase = kPhAfterCopy; VP8L_SAVE_BR(); return kEvCopy; }
        if (rc == kPxRefill) { VP8L_SAVE_BR(); return kEvRefill; }
        st.phase = st.after;
        break;
      }
      case kPhAfterCopy: {
        const uint32_t n = st.copy_len;
        if (st.cache_bits)
          for (uint32_t i = 0; i < n; i++) { const uint32_t px = st.dst[st.fill + i]; st.cache[cache_key(px, st.cache_bits)] = px; }
        st.fill += n;
        const uint64_t xx = (uint64_t)st.x + n;
        st.y += (uint32_t)(xx / st.xsize);
        st.x = (uint32_t)(xx % st.xsize);
        st.phase = kPhPixels;
        break;
      }
      case kPhPaletteFix: {  // vp8l.h:143-145: byte-wise running sum over the palette
        XformRec &x = st.xf[st.nxf - 1];
        uint32_t *pal = st.aux + x.data_off;
        for (uint32_t i = 1; i < x.xsize; i++) pal[i] = px_add(pal[i], pal[i - 1]);
        st.phase = kPhXform;
        break;
      }
      case kPhMain: {  // ReadSpatiallyCodedImage vp8l.h:180-215
        uint32_t cb;
        const int rc = read_cache_bits(br, cb);
        if (rc < 0) VP8L_FAIL(rc);
        st.main_cache_bits = cb;
        st.ngroups = 1;
        st.meta = nullptr;
        st.meta_bits = 0; st.meta_xsize = 0;
        if (br.get(1)) {
          const uint32_t mb = br.get(3) + 2;
          // the reference sizes the entropy image by the image's width (vp8l.h:195-196), the standard by the coded (bundled) width
          const uint32_t mx = (uint32_t)div_round_up(st.standard && st.cix ? st.cix : st.w, mb), my = (uint32_t)div_round_up(st.h, mb);
          uint64_t off;
          if (!aux_alloc(st, (uint64_t)mx * my, off)) VP8L_FAIL(kErrWorkspace);
          st.meta_bits = mb;  // the sub-image reader below runs with one group and ignores it
          begin_sub_image(st, st.aux + off, mx, my, kPhMetaDone);
        } else {
          st.phase = kPhGroups;
        }
        break;
      }
      case kPhMetaDone: {  // vp8l.h:205-214
        uint32_t *m = st.dst;
        uint32_t ng = 1;
        for (uint64_t i = 0; i < st.total; i++) {
          const uint32_t idx = (m[i] >> 8) & 0xffffu;
          m[i] = idx;
          if (idx >= ng) ng = idx + 1;
        }
        st.meta = m;
        st.meta_xsize = st.xsize;
        st.ngroups = ng;
        st.phase = kPhGroups;
        break;
      }
      case kPhGroups: {  // vp8l.h:217-222
        if (st.ngroups > 1) {
          // A group takes at least 20 stream bits (five one-symbol codes of 4 bits, huffman_io.h:152-178) and one that starts
          // past the stream's end reads zeros only, which is no code (InvalidHuffman).  So a count the rest of the stream
          // cannot hold is refused before any workspace is sized by it (a 0xffff index in a tiny stream asks for 65536 groups).
          const uint64_t pos_bits = br.bytepos * 8 - br.cnt, len_bits = br.len * 8;
          const uint64_t bits_left = len_bits > pos_bits ? len_bits - pos_bits : 0;
          if ((uint64_t)(st.ngroups - 1) * 20 > bits_left) VP8L_FAIL(kErrInvalidHuffman);
          if (st.ngroups > st.multi_cap) { st.need_groups = st.ngroups; VP8L_FAIL(kErrWorkspace); }
          st.groups = st.groups_multi;
        }
        const uint32_t cb = st.main_cache_bits;
        for (uint32_t gi = 0; gi < st.ngroups; gi++) {
          if (st.ngroups > 1) st.groups[gi].sorted = st.sorted_multi + (size_t)gi * kSortedPerGroup;
          const int rc = read_group(br, st.groups[gi], cb ? 1u << cb : 0u, st);
          if (rc < 0) VP8L_FAIL(rc);
        }
        st.cache_bits = cb;
        if (cb) for (uint32_t i = 0; i < (1u << cb); i++) st.cache[i] = 0;
        st.dst = st.out;
        st.xsize = st.cix ? st.cix : st.w;
        st.ysize = st.h;
        st.total = (uint64_t)st.xsize * st.ysize;
        st.fill = 0; st.x = 0; st.y = 0;
        st.after = kPhDone;
        st.phase = kPhPixels;
        break;
      }
      default:
        VP8L_SAVE_BR();
        return kEvDone;
    }
  }
#undef VP8L_SAVE_BR
#undef VP8L_FAIL
}

// ---- pixel functions of the inverse transforms (transform.cpp:21-86, transform.h:110-218) ---
VP8L_HD uint32_t px_avg2(uint32_t a, uint32_t b) {  // per byte (a+b)/2 without carries across bytes
  return (((a ^ b) & 0xfefefefeu) >> 1) + (a & b);
}
VP8L_HD int iabs_(int v) { return v < 0 ? -v : v; }
// Select as called at transform.h:177: the callee (transform.cpp:33-54) receives the TOP pixel as
// its first argument and keeps it when its distance to the estimate is not larger.
VP8L_HD uint32_t px_select(uint32_t T, uint32_t L, uint32_t TL) {
  int dT = 0, dL = 0;
  for (int s = 0; s < 32; s += 8) {
    const int t = (int)((T >> s) & 0xff), l = (int)((L >> s) & 0xff), tl = (int)((TL >> s) & 0xff);
    const int p = t + l - tl;
    dT += iabs_(p - t);
    dL += iabs_(p - l);
  }
  return dT <= dL ? T : L;
}
VP8L_HD uint32_t clamp255(int v) { return v < 0 ? 0u : v > 255 ? 255u : (uint32_t)v; }
VP8L_HD uint32_t px_clamp_add_sub_full(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r = 0;
  for (int s = 0; s < 32; s += 8) r |= clamp255((int)((a >> s) & 0xff) + (int)((b >> s) & 0xff) - (int)((c >> s) & 0xff)) << s;
  return r;
}
VP8L_HD uint32_t px_clamp_add_sub_half(uint32_t a, uint32_t b) {
  uint32_t r = 0;
  for (int s = 0; s < 32; s += 8) {
    const int av = (int)((a >> s) & 0xff), bv = (int)((b >> s) & 0xff);
    r |= clamp255(av + (av - bv) / 2) << s;  // truncating division, as in C
  }
  return r;
}
// Prediction for an interior pixel; mode > 13 is invalid (transform.h:185-186).
VP8L_HD uint32_t predict(uint32_t mode, uint32_t L, uint32_t T, uint32_t TR, uint32_t TL) {
  switch (mode) {
    case 0: return 0xff000000u;
    case 1: return L;
    case 2: return T;
    case 3: return TR;
    case 4: return TL;
    case 5: return px_avg2(px_avg2(L, TR), T);
    case 6: return px_avg2(L, TL);
    case 7: return px_avg2(L, T);
    case 8: return px_avg2(TL, T);
    case 9: return px_avg2(T, TR);
    case 10: return px_avg2(px_avg2(L, TL), px_avg2(T, TR));
    case 11: return px_select(T, L, TL);
    case 12: return px_clamp_add_sub_full(L, T, TL);
    default: return px_clamp_add_sub_half(px_avg2(L, T), TL);
  }
}
// transform.h:192-218; tile element bytes: red = red_to_blue, green = green_to_blue, blue = green_to_red
VP8L_HD uint32_t inverse_cross_color(uint32_t px, uint32_t e) {
  const int8_t r2b = (int8_t)(e >> 16), g2b = (int8_t)(e >> 8), g2r = (int8_t)e;
  const int8_t green = (int8_t)(px >> 8);
  uint32_t red = (px >> 16) & 0xff, blue = px & 0xff;
  red = (red + (uint32_t)(int32_t)(int8_t)(((int)g2r * (int)green) >> 5)) & 0xff;
  blue = (blue + (uint32_t)(int32_t)(int8_t)(((int)g2b * (int)green) >> 5)) & 0xff;
  blue = (blue + (uint32_t)(int32_t)(int8_t)(((int)r2b * (int)(int8_t)red) >> 5)) & 0xff;
  return (px & 0xff00ff00u) | (red << 16) | blue;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host driver of the machine plus host inverse transforms: TEST USE ONLY (core_hosttest.cpp).
inline int host_decode(const uint8_t *data, size_t len, uint32_t *argb, size_t cap_pixels, uint32_t *w_out, uint32_t *h_out) {
  BitSrc hb;
  hb.init(data, len);
  (void)hb.get(32);
  if (hb.get(8) != 0x2f) return kErrUnsupportedVp8;
  const uint32_t w = hb.get(14) + 1, h = hb.get(14) + 1;
  if ((size_t)w * h > cap_pixels) return kErrWorkspace;
  const uint64_t tile_max = (uint64_t)((w + 3) / 4) * ((h + 3) / 4);
  const uint64_t aux_cap = 3 * tile_max + 256;
  uint32_t *aux = new uint32_t[aux_cap];
  uint32_t gcap = 64;
  DecState *st = new DecState;
  uint32_t *cache = new uint32_t[1u << kMaxCacheBits];
  DecTableScratch *tsc = new DecTableScratch;
  int status;
  for (;;) {
    GroupTables *groups = new GroupTables[gcap + 1];
    uint16_t *sorted = new uint16_t[(size_t)(gcap + 1) * kSortedPerGroup];
    dec_init(*st, data, len, argb, aux, aux_cap, groups, groups + 1, gcap, cache, tsc, sorted);
    st->w = w; st->h = h;
    while (dec_step(*st) == kEvCopy)
      for (uint32_t i = 0; i < st->copy_len; i++) st->dst[st->copy_pos + i] = st->dst[st->copy_pos + i - st->copy_dist];
    status = st->status;
    delete[] groups;
    delete[] sorted;
    if (status == kErrWorkspace && st->need_groups > gcap) { gcap = st->need_groups; continue; }
    break;
  }
  if (status == kOk) {
    for (int k = (int)st->nxf - 1; k >= 0 && status == kOk; k--) {  // reverse order: vp8l.h:110,343-344
      const XformRec &x = st->xf[k];
      const uint32_t *td = aux + x.data_off;
      if (x.type == 2) {
        for (size_t i = 0; i < (size_t)w * h; i++) argb[i] = add_green(argb[i]);
      } else if (x.type == 1) {
        const uint32_t tw = x.img_w;
        for (uint32_t y = 0; y < h; y++)
          for (uint32_t xx = 0; xx < tw; xx++)
            argb[(size_t)y * tw + xx] = inverse_cross_color(argb[(size_t)y * tw + xx], td[(y >> x.bits) * x.xsize + (xx >> x.bits)]);
      } else if (x.type == 0) {
        const uint32_t tw = x.img_w;
        for (uint32_t y = 0; y < h && status == kOk; y++)
          for (uint32_t xx = 0; xx < tw; xx++) {
            const size_t i = (size_t)y * tw + xx;
            uint32_t P;
            if (xx == 0 && y == 0) P = 0xff000000u;
            else if (xx == 0) P = argb[i - tw];
            else if (y == 0) P = argb[i - 1];
            else {
              const uint32_t mode = (td[(y >> x.bits) * x.xsize + (xx >> x.bits)] >> 8) & 0xff;
              if (mode > 13) { status = kErrInvalidVp8l; break; }
              const uint32_t L = argb[i - 1];
              // last column: the reference takes the left pixel (transform.h:137), the standard the next pixel in memory, which is
              // the first pixel of the current row
              P = predict(mode, L, argb[i - tw], xx == tw - 1 && !st->standard ? L : argb[i - tw + 1], argb[i - tw - 1]);
            }
            argb[i] = px_add(argb[i], P);
          }
      } else {
        const uint32_t bits = x.bits, per = 1u << bits, bpp = 8u >> bits, mask = (1u << bpp) - 1;
        const uint32_t cix = (w + per - 1) >> bits;
        uint32_t *idx = new uint32_t[(size_t)cix * h];
        memcpy(idx, argb, sizeof(uint32_t) * (size_t)cix * h);
        for (uint32_t y = 0; y < h; y++)
          for (uint32_t xx = 0; xx < w; xx++) {
            const uint32_t g = (idx[(size_t)y * cix + (xx >> bits)] >> 8) & 0xff;
            const uint32_t k2 = (g >> ((xx & (per - 1)) * bpp)) & mask;
            argb[(size_t)y * w + xx] = k2 < x.xsize ? td[k2] : 0;
          }
        delete[] idx;
      }
    }
  }
  *w_out = w;
  *h_out = h;
  delete st;
  delete[] cache;
  delete tsc;
  delete[] aux;
  return status;
}
#endif

}  // namespace vp8l

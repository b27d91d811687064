// vp8l_enc_core.h -- the small sequential parts of the encoder: Huffman code construction,
// code-description serialisation, the stream header and the palette sub-image.
//
// These run once per image on a few hundred symbols, so they are written as plain sequential
// host/device code: on the GPU one lane per image executes them out of LDS (k_entropy_header),
// and the same source compiles for the host so it can be unit-tested without a GPU.
// The per-pixel work (transforms, match search, parse, histograms, bit emission) lives in the
// kernels of vp8l_encode.hip.
#pragma once
#include "vp8l_common.h"

namespace vp8l {

// LSB-first bit writer into 32-bit words (webp/utils/bit_writer.h:45-64 semantics: bit i of the
// value goes out first, the tail is zero padded).
struct BitSink {
  uint32_t *words;
  uint64_t acc;
  uint32_t nacc;
  uint64_t nwords;
  VP8L_HD void init(uint32_t *dst) { words = dst; acc = 0; nacc = 0; nwords = 0; }
  VP8L_HD void put(uint32_t v, uint32_t n) {  // n <= 32
    if (n == 0) return;
    const uint64_t m = n >= 32 ? 0xffffffffull : ((1ull << n) - 1ull);
    acc |= ((uint64_t)v & m) << nacc;
    nacc += n;
    if (nacc >= 32) {
      words[nwords++] = (uint32_t)acc;
      acc >>= 32;
      nacc -= 32;
    }
  }
  VP8L_HD uint64_t bits() const { return nwords * 32 + nacc; }
  // Writes the partial last word zero padded (later writers OR their bits into it).
  VP8L_HD void flush() { words[nwords] = (uint32_t)acc; }
};

template <int N>
struct HuffScratchT {  // for alphabets of up to N symbols
  uint32_t wt[2 * N];   // (also the work area of canonical_reversed_codes once the lengths are known)
  uint16_t parent[2 * N];
  uint16_t ord[N];
  uint16_t stack[N];
  uint16_t depth[2 * N];
};
using HuffScratch = HuffScratchT<kCodeStride>;
using HuffScratchBig = HuffScratchT<kExtGreenMax>;

VP8L_HD uint32_t bit_reverse(uint32_t v, int len) {
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
  v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
  v = (v >> 16) | (v << 16);
  return v >> (32 - len);
}

// Code lengths of the reference's Huffman construction (huffman_coding.h:366-419).
// The reference keeps an array of roots sorted ascending by weight and, after joining the two
// lightest, re-sorts with a stable sort with the new root in slot 0: the new root therefore
// lands in front of every root of equal weight, and leaves of equal weight stay in symbol order.
// That tie rule decides the lengths, so it is reproduced exactly, in linear time after the sort:
// joined weights never decrease, so the internal nodes are created in ascending weight order and
// the array order among roots of one weight is: internal nodes newest first, then the leaves in
// symbol order.  Leaves are taken from their sorted list, internal nodes from a stack that holds
// the lightest run of equal-weight internal nodes (newest on top); later, heavier ones wait in
// creation order (`pend`) and are moved to the stack run by run.
// Returns the number of used symbols, or -3 (TooBigCodeLength, huffman_coding.h:410-413).
// n <= kCodeStride.  presorted_used >= 0: the caller has already filled s.wt[0..used) with the used
// symbols' weights in symbol order and s.ord with the leaf ids sorted by (weight, id).
// The joins alone: s.wt[0..used) and s.ord as above; fills s.parent and returns the number of nodes (the root is the last).
template <class Scratch>
VP8L_HD int huffman_join(Scratch &s, int used) {
  int nn = used;
  int lp = 0;          // next leaf in ord
  int top = 0;         // stack height (s.stack)
  int pend = used;     // internal nodes pend..nn-1 are not on the stack yet
  uint32_t run_w = 0;  // weight of the run on the stack
  for (int joins = used - 1; joins > 0; joins--) {
    int pick[2];
    for (int k = 0; k < 2; k++) {
      if (top == 0 && pend < nn) {  // next run of equal-weight internal nodes, oldest first (newest ends on top)
        run_w = s.wt[pend];
        while (pend < nn && s.wt[pend] == run_w) s.stack[top++] = (uint16_t)pend++;
      }
      const bool have_leaf = lp < used;
      if (top > 0 && (!have_leaf || run_w <= s.wt[s.ord[lp]])) pick[k] = s.stack[--top];
      else pick[k] = s.ord[lp++];
    }
    const uint32_t w = s.wt[pick[0]] + s.wt[pick[1]];
    s.wt[nn] = w;
    s.parent[pick[0]] = s.parent[pick[1]] = (uint16_t)nn;
    if (top > 0 && pend == nn && w == run_w) { s.stack[top++] = (uint16_t)nn; pend = nn + 1; }  // joins the run on the stack, newest
    nn++;
  }
  return nn;
}

template <class Scratch>
VP8L_HD int huffman_lengths(const uint32_t *histo, int n, int max_len, uint8_t *lengths, Scratch &s, int presorted_used = -1) {
  int nn = 0;
  if (presorted_used < 0) {
    for (int i = 0; i < n; i++)
      if (histo[i]) s.wt[nn++] = histo[i];
    // leaf k is the k-th used symbol; stable insertion sort of leaf ids by weight
    for (int i = 0; i < nn; i++) {
      int j = i;
      const uint32_t w = s.wt[i];
      while (j > 0 && s.wt[s.ord[j - 1]] > w) { s.ord[j] = s.ord[j - 1]; j--; }
      s.ord[j] = (uint16_t)i;
    }
  } else {
    nn = presorted_used;
  }
  const int used = nn;
  for (int i = 0; i < n; i++) lengths[i] = 0;
  if (used == 0) return 0;
  nn = huffman_join(s, used);
  // depths from the root down (a parent is always created after its children)
  s.depth[nn - 1] = 0;
  for (int v = nn - 2; v >= 0; v--) s.depth[v] = (uint16_t)(s.depth[s.parent[v]] + 1);
  int maxl = 0, leaf = 0;
  for (int i = 0; i < n; i++) {
    if (!histo[i]) continue;
    int d = s.depth[leaf];
    if (used == 1) d = 1;  // a lone symbol gets length 1 (huffman_coding.h:398-402)
    lengths[i] = (uint8_t)(d > 255 ? 255 : d);
    if (d > maxl) maxl = d;
    leaf++;
  }
  if (maxl > max_len || maxl > kSaneCodeLen) return -3;
  return used;
}

// Standard dialect: codes of at most `limit` bits (15, or 7 for the code-length code).  The reference has no such step (its
// dialect allows 64); any complete prefix code within the limit is a valid stream there, so the usual remedy is used: counts
// below a floor are raised to the floor and the code is built again, the floor doubling until the longest code fits.  The
// floor starts where a code of `limit` bits becomes possible at all (total / 2^limit).  `histo` is changed in place.
template <class Scratch>
VP8L_HD int huffman_lengths_limited(uint32_t *histo, int n, int limit, uint8_t *lengths, Scratch &s, int presorted_used = -1) {
  int used = huffman_lengths(histo, n, limit, lengths, s, presorted_used);
  if (used != -3) return used;
  uint64_t total = 0;
  for (int i = 0; i < n; i++) total += histo[i];
  uint32_t floor_count = 2;
  while (((uint64_t)floor_count << limit) < total) floor_count <<= 1;
  floor_count >>= 1;
  if (floor_count < 2) floor_count = 2;
  for (;;) {
    for (int i = 0; i < n; i++)
      if (histo[i] && histo[i] < floor_count) histo[i] = floor_count;
    used = huffman_lengths(histo, n, limit, lengths, s);
    if (used != -3 || floor_count >= 0x40000000u) return used;
    floor_count <<= 1;
  }
}

// Canonical codes in symbol order (huffman_coding.cpp:8-49), bit-reversed for the LSB-first
// writer (huffman_coding.h:356-363,415-418).  Lengths <= 32.
// work: kCanonWork words for the two small tables (symbols per length, next code per length).  On the GPU a caller passes
// shared memory: arrays of a function that are indexed by a variable live in a lane's private memory there, and the two
// loops over the symbols are chains of read-modify-writes of it, each several hundred cycles.  Null: local arrays.
constexpr int kCanonWork = 2 * (kSaneCodeLen + 2);
VP8L_HD void canonical_reversed_codes(const uint8_t *lengths, int n, uint32_t *codes, uint32_t *work = nullptr) {
  uint32_t local[kCanonWork];
  uint32_t *count = work ? work : local, *next = count + kSaneCodeLen + 2;
  for (int l = 0; l <= kSaneCodeLen + 1; l++) count[l] = 0;
  int maxl = 0;
  for (int i = 0; i < n; i++) { count[lengths[i]]++; if (lengths[i] > maxl) maxl = lengths[i]; }
  count[0] = 0;
  uint32_t cur = 0;
  next[0] = 0;
  for (int l = 1; l <= maxl; l++) { cur = (cur + count[l - 1]) << 1; next[l] = cur; }
  for (int i = 0; i < n; i++) codes[i] = lengths[i] ? bit_reverse(next[lengths[i]]++, lengths[i]) : 0;
}

template <int N>
struct DescScratchT {  // for code descriptions of alphabets of up to N symbols
  HuffScratch huff;  // the code-length code has at most 68 symbols
  uint8_t rle_sym[N + 8];
  uint8_t rle_extra[N + 8];
  uint32_t cl_histo[kNumClSyms];
  uint32_t cl_code[kNumClSyms];
  uint8_t cl_len[kNumClSyms];
};
using DescScratch = DescScratchT<kCodeStride>;
using DescScratchBig = DescScratchT<kExtGreenMax>;

// Run-length form of a code-length array (huffman_io.h:226-326).
VP8L_HD int rle_code_lengths(const uint8_t *lengths, int n, uint8_t *sym, uint8_t *extra, const Dialect &dl = dialect_of(false)) {
  int k = 0, prev = 8;
  int i = 0;
  while (i < n) {
    const int v = lengths[i];
    int j = i + 1;
    while (j < n && lengths[j] == v) j++;
    int reps = j - i;
    i = j;
    if (v == 0) {
      while (reps >= 139) { sym[k] = (uint8_t)dl.zero138; extra[k++] = 127; reps -= 138; }
      if (reps >= 11) { sym[k] = (uint8_t)dl.zero138; extra[k++] = (uint8_t)(reps - 11); }
      else if (reps >= 3) { sym[k] = (uint8_t)dl.zero11; extra[k++] = (uint8_t)(reps - 3); }
      else for (; reps > 0; reps--) { sym[k] = 0; extra[k++] = 0; }
    } else {
      if (prev != v) { sym[k] = (uint8_t)v; extra[k++] = 0; reps--; }
      while (reps >= 7) { sym[k] = (uint8_t)dl.repeat; extra[k++] = 3; reps -= 6; }
      if (reps >= 3) { sym[k] = (uint8_t)dl.repeat; extra[k++] = (uint8_t)(reps - 3); }
      else for (; reps > 0; reps--) { sym[k] = (uint8_t)v; extra[k++] = 0; }
      prev = v;
    }
  }
  return k;
}

// One code description (huffman_io.h:331-410).  Returns 0 or -3.  standard: the same description in the standard
// dialect's code-length alphabet and field widths.
template <class Desc>
VP8L_HD int write_code_desc(BitSink &bw, const uint8_t *lengths, int n, Desc &s, bool standard = false) {
  const Dialect dl = dialect_of(standard);
  int count = 0;
  uint32_t first[2] = {0, 0};
  for (int i = 0; i < n && count <= 2; i++)
    if (lengths[i]) { if (count < 2) first[count] = (uint32_t)i; count++; }
  if (count == 0) { bw.put(1, 4); return 0; }  // bits 1,0,0,0 (huffman_io.h:387-392)
  if (count <= 2 && first[0] < 256 && first[1] < 256) {
    bw.put(1, 1);
    bw.put((uint32_t)(count - 1), 1);
    if (first[0] <= 1) { bw.put(0, 1); bw.put(first[0], 1); }
    else { bw.put(1, 1); bw.put(first[0], 8); }
    if (count == 2) bw.put(first[1], 8);
    return 0;
  }
  bw.put(0, 1);
  const int nitems = rle_code_lengths(lengths, n, s.rle_sym, s.rle_extra, dl);
  for (int i = 0; i < kNumClSyms; i++) s.cl_histo[i] = 0;
  for (int i = 0; i < nitems; i++) s.cl_histo[s.rle_sym[i]]++;
  const int used = standard ? huffman_lengths_limited(s.cl_histo, dl.ncl, dl.max_cl_len, s.cl_len, s.huff)
                            : huffman_lengths(s.cl_histo, dl.ncl, dl.max_cl_len, s.cl_len, s.huff);
  if (used < 0) return used;
  canonical_reversed_codes(s.cl_len, dl.ncl, s.cl_code, s.huff.wt);  // (the code is built: its scratch is free)
  int nwrite = dl.ncl;
  while (nwrite > 4 && s.cl_len[dl.order(nwrite - 1)] == 0) nwrite--;
  bw.put((uint32_t)(nwrite - 4), (uint32_t)dl.count_bits);
  for (int i = 0; i < nwrite; i++) bw.put(s.cl_len[dl.order(i)], (uint32_t)dl.len_bits);
  bw.put(0, 1);  // no max_symbol (huffman_io.h:358)
  for (int i = 0; i < nitems; i++) {
    const int sy = s.rle_sym[i];
    // With a single used code-length symbol its length is 1 and one 0 bit per item is written,
    // although the reference reader consumes none (huffman_io.h:363 vs huffman_coding.h:207-214):
    // kept byte-identical; the reference cannot decode such a stream either.  A standard reader takes no bits for a
    // one-symbol code either, and there none are written.
    if (!(standard && used == 1)) bw.put(s.cl_code[sy], s.cl_len[sy]);
    if (sy == dl.repeat) bw.put(s.rle_extra[i], 2);
    else if (sy == dl.zero11) bw.put(s.rle_extra[i], 3);
    else if (sy == dl.zero138) bw.put(s.rle_extra[i], 7);
  }
  return 0;
}

// Encoder-side tables of one image: 5 codes, stride kCodeStride.
struct CodeTables {
  uint32_t code[kNumCodes * kCodeStride];
  uint8_t len[kNumCodes * kCodeStride];
  uint32_t used[kNumCodes];  // number of symbols with non-zero count
  uint8_t wbits[kNumCodes * kCodeStride];  // GPU encoder: bits one occurrence of the symbol takes in the stream (0 for a code that is not
                                           // written, plus the extra bits behind a length or distance prefix); filled by k_entropy_header
};

// Builds the five codes from the five histograms and writes their descriptions
// (vp8l.h:536-540).  Returns 0 or -3.
VP8L_HD int build_and_write_codes(BitSink &bw, uint32_t *histo /*[5*kCodeStride]; changed in the standard dialect*/, CodeTables &t,
                                  DescScratch &s, bool standard = false) {
  for (int c = 0; c < kNumCodes; c++) {
    const int n = alphabet_size(c);
    uint8_t *len = t.len + c * kCodeStride;
    const int used = standard ? huffman_lengths_limited(histo + c * kCodeStride, n, dialect_of(true).max_len, len, s.huff)
                              : huffman_lengths(histo + c * kCodeStride, n, kMaxCodeLen, len, s.huff);
    if (used < 0) return used;
    t.used[c] = (uint32_t)used;
    canonical_reversed_codes(len, n, t.code + c * kCodeStride, s.huff.wt);
    const int rc = write_code_desc(bw, len, n, s, standard);
    if (rc < 0) return rc;
  }
  return 0;
}

// Greedy parse of a SMALL array (n <= kSmallMax: the palette, or the predictor modes) with the reference's rules
// (lz77.h:28-93): window 1024, match length <= min(distance, 128, remaining), shortest match 1,
// farthest candidate wins ties, the final element is always a literal.  adv[p] = match length at
// p (0 = literal), dst[p] = its distance.  O(n^2): next_mis[d] tracks, while p runs backwards,
// the first position q >= p with data[q-d] != data[q].
struct alignas(4) SmallLzScratch {
  uint16_t next_mis[kSmallMax + 1];
  uint16_t len[kSmallMax];
  uint16_t dist[kSmallMax];
};

VP8L_HD void small_best_matches(const uint32_t *data, int n, SmallLzScratch &s) {
  for (int d = 0; d <= n; d++) s.next_mis[d] = (uint16_t)n;
  for (int p = n - 1; p >= 0; p--) {
    int best = 0, bestd = 0;
    const int rem = n - p < (int)kMaxLen ? n - p : (int)kMaxLen;
    for (int d = 1; d <= p && d <= (int)kWindow; d++) {
      if (data[p - d] != data[p]) s.next_mis[d] = (uint16_t)p;
      int l = s.next_mis[d] - p;
      if (l > d) l = d;
      if (l > rem) l = rem;
      if (l > 0 && l >= best) { best = l; bestd = d; }
    }
    if (rem <= 1) best = 0;
    s.len[p] = (uint16_t)best;
    s.dist[p] = (uint16_t)bestd;
  }
}

struct SubImageScratch {
  DescScratch desc;
  SmallLzScratch lz;
  CodeTables tables;
  uint32_t histo[kNumCodes * kCodeStride];
  uint32_t delta[kSmallMax];  // the sub-image: palette deltas, or one predictor mode per tile
};

// Token bits of one position (vp8l.h:545-583).  Emits through `put(value, nbits)`.
template <class Put>
VP8L_HD void emit_token(Put &&put, const CodeTables &t, uint32_t pixel, uint32_t len, uint32_t dist, uint32_t xsize) {
  if (len == 0) {
    const uint32_t g = (pixel >> 8) & 0xff, r = (pixel >> 16) & 0xff, b = pixel & 0xff, a = pixel >> 24;
    if (t.used[kGreen] > 1) put(t.code[kGreen * kCodeStride + g], t.len[kGreen * kCodeStride + g]);
    if (t.used[kRed] > 1) put(t.code[kRed * kCodeStride + r], t.len[kRed * kCodeStride + r]);
    if (t.used[kBlue] > 1) put(t.code[kBlue * kCodeStride + b], t.len[kBlue * kCodeStride + b]);
    if (t.used[kAlpha] > 1) put(t.code[kAlpha * kCodeStride + a], t.len[kAlpha * kCodeStride + a]);
  } else {
    uint32_t sym, nb, eb;
    prefix_encode(len, sym, nb, eb);
    put(t.code[kGreen * kCodeStride + 256 + sym], t.len[kGreen * kCodeStride + 256 + sym]);  // always written (vp8l.h:570)
    if (nb) put(eb, nb);
    prefix_encode(dist_to_code(xsize, dist), sym, nb, eb);
    if (t.used[kDist] > 1) put(t.code[kDist * kCodeStride + sym], t.len[kDist * kCodeStride + sym]);
    if (nb) put(eb, nb);
  }
}

VP8L_HD void histo_token(uint32_t *histo, uint32_t pixel, uint32_t len, uint32_t dist, uint32_t xsize) {
  if (len == 0) {
    histo[kGreen * kCodeStride + ((pixel >> 8) & 0xff)]++;
    histo[kRed * kCodeStride + ((pixel >> 16) & 0xff)]++;
    histo[kBlue * kCodeStride + (pixel & 0xff)]++;
    histo[kAlpha * kCodeStride + (pixel >> 24)]++;
  } else {
    uint32_t sym, nb, eb;
    prefix_encode(len, sym, nb, eb);
    histo[kGreen * kCodeStride + 256 + sym]++;
    prefix_encode(dist_to_code(xsize, dist), sym, nb, eb);
    histo[kDist * kCodeStride + sym]++;
  }
}

// The pieces of WriteEntropyCodedImage for the palette (vp8l.h:477-510, called from :457 with xsize = n,
// ysize = 1).  The device computes the matches and the code
// lengths with many lanes and calls the pieces; write_small_coded_image is the same thing in one go.
// xsize: the sub-image's width (distance codes depend on it): n for the palette (n x 1), tile columns for predictor modes
VP8L_HD void small_histogram(const uint32_t *data, int n, uint32_t xsize, SubImageScratch &s) {
  for (int i = 0; i < kNumCodes * kCodeStride; i++) s.histo[i] = 0;
  for (int p = 0; p < n; p += s.lz.len[p] ? s.lz.len[p] : 1) histo_token(s.histo, data[p], s.lz.len[p], s.lz.dist[p], xsize);
}
VP8L_HD void small_write_tokens(BitSink &bw, const uint32_t *data, int n, uint32_t xsize, SubImageScratch &s) {
  for (int p = 0; p < n; p += s.lz.len[p] ? s.lz.len[p] : 1)
    emit_token([&](uint32_t v, uint32_t nb) { bw.put(v, nb); }, s.tables, data[p], s.lz.len[p], s.lz.dist[p], xsize);
}
VP8L_HD int write_small_coded_image(BitSink &bw, const uint32_t *data, int n, uint32_t xsize, SubImageScratch &s, bool standard = false) {
  bw.put(0, 1);  // no colour cache
  small_best_matches(data, n, s.lz);
  small_histogram(data, n, xsize, s);
  const int rc = build_and_write_codes(bw, s.histo, s.tables, s.desc, standard);
  if (rc < 0) return rc;
  small_write_tokens(bw, data, n, xsize, s);
  return 0;
}

// Everything in front of the main image's codes (vp8l.h:585-612, WriteTransforms :411-459).
// standard: the alpha-is-used bit is set (the reference always writes 0, vp8l.h:465; standard readers may drop the alpha
// channel when it is 0), and the first 32 bits are patched with the chunk size once the stream's length is known.
VP8L_HD void write_stream_head(BitSink &bw, uint32_t w, uint32_t h, uint32_t ncolors, bool standard = false) {
  bw.put(0, 32);
  bw.put(0x2f, 8);
  bw.put(w - 1, 14);
  bw.put(h - 1, 14);
  bw.put(standard ? 1 : 0, 1);
  bw.put(0, 3);
  bw.put(1, 1);
  if (ncolors == 0) {
    bw.put(2, 2);
  } else {
    bw.put(3, 2);
    bw.put(ncolors - 1, 8);
  }
}
VP8L_HD void palette_deltas(const uint32_t *palette, uint32_t ncolors, uint32_t *delta) {  // vp8l.h:455-456
  delta[0] = palette[0];
  for (uint32_t i = 1; i < ncolors; i++) delta[i] = px_sub(palette[i], palette[i - 1]);
}
// Predictor transform header (ReadTransform vp8l.h:111-121 reads it): present, type 0, tile bits - 2.  Not something the
// reference encoder writes (vp8l.h:594-610); its decoder reads it.
VP8L_HD void write_predictor_head(BitSink &bw, uint32_t bits) {
  bw.put(1, 1);
  bw.put(0, 2);
  bw.put(bits - 2, 3);
}
// Cross-colour transform header (ReadTransform vp8l.h:111-121): present, type 1, tile bits - 2.
VP8L_HD void write_cross_color_head(BitSink &bw, uint32_t bits) {
  bw.put(1, 1);
  bw.put(1, 2);
  bw.put(bits - 2, 3);
}
VP8L_HD void write_stream_tail(BitSink &bw) {
  bw.put(0, 1);
  bw.put(0, 1);
  bw.put(0, 1);
}
VP8L_HD int write_stream_prefix(BitSink &bw, uint32_t w, uint32_t h, const uint32_t *palette, uint32_t ncolors,
                                SubImageScratch &s, bool standard = false) {
  write_stream_head(bw, w, h, ncolors, standard);
  if (ncolors) {
    palette_deltas(palette, ncolors, s.delta);
    const int rc = write_small_coded_image(bw, s.delta, (int)ncolors, ncolors, s, standard);
    if (rc < 0) return rc;
  }
  write_stream_tail(bw);
  return 0;
}

}  // namespace vp8l

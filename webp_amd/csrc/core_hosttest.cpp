// core_hosttest.cpp -- TEST-ONLY host build of the sequential host/device cores
// (vp8l_common.h, vp8l_enc_core.h, vp8l_dec_core.h) so they can be checked against the oracle
// on a machine without a GPU.  Not part of the shipped library: libvp8l_amd.so does not contain
// this file and has no CPU fallback.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "vp8l_dec_core.h"
#include "vp8l_enc_core.h"
#include "vp8l_split.h"

using namespace vp8l;

extern "C" {

int core_huffman(const uint32_t *histo, int n, int max_len, uint8_t *lengths, uint32_t *codes) {
  HuffScratch s;
  int used = huffman_lengths(histo, n, max_len, lengths, s);
  if (used >= 0) canonical_reversed_codes(lengths, n, codes);
  return used;
}

long core_write_code_desc(const uint8_t *lengths, int n, uint8_t *buf, size_t cap) {
  std::vector<uint32_t> words(cap / 4 + 2, 0);
  BitSink bw;
  bw.init(words.data());
  DescScratch s;
  int rc = write_code_desc(bw, lengths, n, s);
  if (rc < 0) return rc;
  bw.flush();
  memcpy(buf, words.data(), cap);
  return (long)bw.bits();
}

void core_prefix_encode(uint32_t v, uint32_t *sym, uint32_t *nb, uint32_t *eb) { prefix_encode(v, *sym, *nb, *eb); }
uint32_t core_dist_to_code(uint32_t xsize, uint32_t dist) { return dist_to_code(xsize, dist); }
uint32_t core_code_to_dist(uint32_t xsize, uint32_t code) { return code_to_dist(xsize, code); }
uint32_t core_subtract_green(uint32_t p) { return subtract_green(p); }
// split_by_weight (vp8l_split.h): first[0..parts] into `first`
void core_split_by_weight(const uint64_t *weight, size_t n, size_t parts, size_t *first) {
  const std::vector<size_t> f = split_by_weight(std::vector<uint64_t>(weight, weight + n), parts);
  for (size_t i = 0; i <= parts; i++) first[i] = f[i];
}
uint32_t core_add_green(uint32_t p) { return add_green(p); }

// The encoder assembled from the core pieces in the order the kernels use them, with a
// brute-force match search standing in for the match kernel.  Returns bytes or a negative status.
long core_encode_ex(const uint32_t *argb, uint32_t w, uint32_t h, uint8_t *out, size_t cap, int standard) {
  const size_t n = (size_t)w * h;
  std::vector<uint32_t> pal;
  bool has_pal = true;
  for (size_t i = 0; i < n && has_pal; i++) {
    size_t lo = 0, hi = pal.size();
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (pal[mid] < argb[i]) lo = mid + 1; else hi = mid; }
    if (lo < pal.size() && pal[lo] == argb[i]) continue;
    if (pal.size() == kMaxPalette) { has_pal = false; break; }
    pal.insert(pal.begin() + lo, argb[i]);
  }
  const uint32_t ncolors = has_pal ? (uint32_t)pal.size() : 0;
  const uint32_t bits = ncolors ? palette_bits(ncolors) : 0;
  const uint32_t xsize = (w + (1u << bits) - 1) >> bits;
  const size_t m = (size_t)xsize * h;
  std::vector<uint32_t> pix(m);
  if (!ncolors) {
    for (size_t i = 0; i < n; i++) pix[i] = subtract_green(argb[i]);
  } else {
    const uint32_t depth = 8u >> bits;
    for (uint32_t y = 0; y < h; y++)
      for (uint32_t xp = 0; xp < xsize; xp++) {
        uint32_t code = 0xff000000u;
        for (uint32_t k = 0; k < (1u << bits); k++) {
          uint32_t x = (xp << bits) + k;
          if (x >= w) break;
          uint32_t v = argb[(size_t)y * w + x];
          uint32_t idx = (uint32_t)(std::lower_bound(pal.begin(), pal.end(), v) - pal.begin());
          code |= idx << (8 + depth * k);
        }
        pix[(size_t)y * xsize + xp] = code;
      }
  }
  std::vector<uint32_t> words(cap / 4 + 16, 0);
  BitSink bw;
  bw.init(words.data());
  static SubImageScratch sub;
  int rc = write_stream_prefix(bw, w, h, pal.data(), ncolors, sub, standard != 0);
  if (rc < 0) return rc;
  std::vector<uint32_t> len(m), dist(m);
  for (size_t p = 0; p < m; p++) {
    uint32_t rem = (uint32_t)std::min<size_t>(kMaxLen, m - p), best = 0, bestd = 0;
    for (uint32_t d = 1; d <= kWindow && d <= p; d++) {
      uint32_t l = 0;
      while (l < d && l < rem && pix[p - d + l] == pix[p + l]) l++;
      if (l && l >= best) { best = l; bestd = d; }
    }
    if (rem <= 1) best = 0;
    len[p] = best;
    dist[p] = bestd;
  }
  std::vector<uint32_t> histo(kNumCodes * kCodeStride, 0);
  for (size_t p = 0; p < m; p += len[p] ? len[p] : 1) histo_token(histo.data(), pix[p], len[p], dist[p], xsize);
  static CodeTables tables;
  static DescScratch desc;
  rc = build_and_write_codes(bw, histo.data(), tables, desc, standard != 0);
  if (rc < 0) return rc;
  for (size_t p = 0; p < m; p += len[p] ? len[p] : 1)
    emit_token([&](uint32_t v, uint32_t nb) { bw.put(v, nb); }, tables, pix[p], len[p], dist[p], xsize);
  bw.flush();
  size_t bytes = (size_t)((bw.bits() + 7) / 8);
  if (bytes > cap) return -8;
  if (standard) words[0] = (uint32_t)(bytes - 4);  // the chunk size: what follows these four bytes
  memcpy(out, words.data(), bytes);
  return (long)bytes;
}
long core_encode(const uint32_t *argb, uint32_t w, uint32_t h, uint8_t *out, size_t cap) { return core_encode_ex(argb, w, h, out, cap, 0); }

// Sequential decoder core (header parse + entropy decode + inverse transforms on the host).
int core_decode(const uint8_t *data, size_t len, uint32_t *argb, size_t cap_pixels, uint32_t *w, uint32_t *h) {
  return host_decode(data, len, argb, cap_pixels, w, h);
}

}  // extern "C"

// ref_parts_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" shims over the REFERENCE's own classes, compiled together with the
// reference's own .cpp files where they lie (see oracle/Makefile, target `ref`).  Nothing
// here restates an algorithm: each function only moves data in and out of a reference
// object.  The result (oracle/_ref/libref_parts.so) pins the CPU oracle component by
// component.  webp/vp8l/vp8l.h (the encoder/decoder drivers) is not included because it
// needs <png.h>, which this image lacks.
#include "webp/platform.h"
#include "webp/exception/exception.h"
#include "webp/utils/utils.h"
#include "webp/utils/bit_readed.h"
#include "webp/utils/bit_writer.h"
#include "webp/lz77/lz77.h"
#include "webp/huffman_coding/huffman_coding.h"
#include "webp/vp8l/huffman_io.h"
#include "webp/vp8l/transform.h"
#include "webp/vp8l/color_cache.h"

#include <unistd.h>

using namespace webp;

namespace {
struct Init {
  Init() { vp8l::huffman_io::init_array(); }  // webp.cpp:128
} g_init;

// BitWriter only exposes its bytes through save2file (append mode): go through a temp file.
size_t dump_writer(const utils::BitWriter &bw, uint8_t *buf, size_t cap) {
  char path[] = "/tmp/ref_parts_XXXXXX";
  int fd = mkstemp(path);
  if (fd < 0) return 0;
  close(fd);
  unlink(path);
  bw.save2file(path);
  FILE *fp = fopen(path, "rb");
  size_t n = 0;
  if (fp) {
    n = fread(buf, 1, cap, fp);
    fclose(fp);
  }
  unlink(path);
  return n;
}
}  // namespace

extern "C" {

// lz77.h:28-93.  out = ntok * {distance, length, symbol}; match symbols are zeroed because the
// reference fills them with an unused (and at end of data out-of-bounds) look-ahead value.
size_t ref_lz77(const uint32_t *data, uint32_t n, uint32_t *out, size_t cap_tokens) {
  // one spare element: lz77.h:69 reads data[length + index] one past the end on a final match
  uint32_t *copy = new uint32_t[n + 1];
  memcpy(copy, data, sizeof(uint32_t) * n);
  copy[n] = 0;
  lz77::LZ77<uint32_t> lz(1024, 128, copy, n);
  size_t k = lz.output().size();
  for (size_t i = 0; i < k && i < cap_tokens; i++) {
    out[3 * i] = lz.output()[i].distance;
    out[3 * i + 1] = lz.output()[i].length;
    bool lit = lz.output()[i].distance == 0 && lz.output()[i].length == 0;
    out[3 * i + 2] = lit ? lz.output()[i].symbol : 0;
  }
  delete[] copy;
  return k;
}

void ref_prefix_encode(uint32_t v, uint32_t *sym, uint32_t *nbits, uint32_t *bits) {
  uint16_t s;
  size_t nb, eb;
  lz77::prefix_coding_encode(v, s, nb, eb);
  *sym = s;
  *nbits = (uint32_t)nb;
  *bits = (uint32_t)eb;
}

uint32_t ref_prefix_decode(uint32_t sym, const uint8_t *data, size_t len, uint32_t *nbits_used) {
  utils::BitReader br(data, len);
  uint32_t v = lz77::prefix_coding_decode(sym, br);
  *nbits_used = sym < 4 ? 0 : (sym - 2) >> 1;
  return v;
}

uint32_t ref_dist_to_code(uint64_t xsize, uint64_t dist) {
  size_t xs = xsize, d = dist;
  return (uint32_t)lz77::distance2dist_code(xs, d);
}

uint32_t ref_code_to_dist(uint32_t xsize, uint32_t code) { return lz77::distance_code2distance(xsize, code); }

// huffman_coding.h:424.  Returns get_num_nodes(), or -3 when TooBigCodeLength is thrown.
int ref_huffman_build(const uint32_t *histo, int n, int max_len, uint8_t *lengths, uint32_t *codes) {
  histoarray h(n);
  for (int i = 0; i < n; i++) h[i] = histo[i];
  try {
    huffman_coding::enc::HuffmanTree t(h, (size_t)max_len);
    for (int i = 0; i < n; i++) {
      lengths[i] = t.get_lengths()[i];
      codes[i] = t.get_lengths()[i] ? t.get_codes()[i] : 0;
    }
    return (int)t.get_num_nodes();
  } catch (exception::TooBigCodeLength &) {
    return -3;
  }
}

// huffman_io.h:417-421 (write_codes).  A terminating 1 bit is appended so the caller can
// recover the exact bit count from the returned bytes.  Returns bytes, or -3.
long ref_write_code(const uint32_t *histo, int n, int max_len, uint8_t *buf, size_t cap) {
  histoarray h(n);
  for (int i = 0; i < n; i++) h[i] = histo[i];
  try {
    huffman_coding::enc::HuffmanTree t(h, (size_t)max_len);
    utils::BitWriter bw;
    vp8l::huffman_io::enc::VP8_LOSSLESS_HUFFMAN io(&bw, t);
    bw.WriteBit(1);
    return (long)dump_writer(bw, buf, cap);
  } catch (exception::TooBigCodeLength &) {
    return -3;
  }
}

// huffman_io.h:200-214: read one 5-code group, then decode nsyms symbols, symbol i with the
// code which[i] (0..4).  Returns 0, or -6 when InvalidHuffman is thrown.
int ref_read_group(const uint8_t *data, size_t len, uint32_t cache_size, const uint8_t *which,
                   size_t nsyms, int32_t *syms) {
  // slack: the reference reader touches one byte past the end before reporting eos
  uint8_t *copy = new uint8_t[len + 16];
  memcpy(copy, data, len);
  memset(copy + len, 0, 16);
  int rc = 0;
  try {
    utils::BitReader br(copy, len);
    vp8l::huffman_io::dec::VP8_LOSSLESS_HUFFMAN g(&br, cache_size);
    for (size_t i = 0; i < nsyms; i++) syms[i] = g.read_symbol((vp8l::huffman_io::MetaHuffmanCode)which[i]);
  } catch (exception::InvalidHuffman &) {
    rc = -6;
  }
  delete[] copy;
  return rc;
}

// transform.h:265-275.  tdata has txsize*tysize entries.  Returns 0, or -5 on InvalidVP8L.
int ref_inverse_transform(int type, uint32_t *argb, uint32_t w, uint32_t h, const uint32_t *tdata,
                          uint32_t txsize, uint32_t tysize, uint32_t tbits) {
  vp8l::VP8_LOSSLESS_TRANSFORM t((vp8l::VP8_LOSSLESS_TRANSFORM::Type)type);
  if (tdata) {
    t.init_data(txsize, tysize, tbits);
    for (uint32_t i = 0; i < txsize * tysize; i++) t.data()[i] = tdata[i];
  }
  // slack: InverseColorIndexingTransform writes whole pixel groups past the row end
  utils::pixel_array img(w * h + 16);
  for (uint32_t i = 0; i < w * h; i++) img[i] = argb[i];
  int rc = 0;
  try {
    t.inverse(img, w, h);
  } catch (exception::InvalidVP8L &) {
    rc = -5;
  }
  for (uint32_t i = 0; i < w * h; i++) argb[i] = img[i];
  return rc;
}

void ref_bundle(const uint8_t *row, int width, int xbits, uint32_t *dst) { vp8l::BundleColorMap(row, width, xbits, dst); }

uint32_t ref_pixels_sub(uint32_t a, uint32_t b) {
  vp8l::PixelsSub(&a, b);
  return a;
}

uint32_t ref_pixels_sum(uint32_t a, uint32_t b) {
  vp8l::PixelsSum(&a, b);
  return a;
}

// color_cache.h:29-54: insert colors[0..n) in order, then return the table (1<<bits entries).
void ref_color_cache(uint32_t bits, const uint32_t *colors, size_t n, uint32_t *table) {
  vp8l::VP8_LOSSLESS_COLOR_CACHE cc(bits);
  for (size_t i = 0; i < n; i++) cc.insert(colors[i]);
  for (uint32_t k = 0; k < (1u << bits); k++) table[k] = cc.get(k);
}

// bit_writer.h:60-64 then bit_readed.h:54-63: write (value,count) pairs, return the bytes.
size_t ref_bit_writer(const uint32_t *values, const uint32_t *counts, size_t n, uint8_t *buf, size_t cap) {
  utils::BitWriter bw;
  for (size_t i = 0; i < n; i++) bw.WriteBits(values[i], counts[i]);
  return dump_writer(bw, buf, cap);
}

}  // extern "C"

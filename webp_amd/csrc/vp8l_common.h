// vp8l_common.h -- constants and small scalar helpers shared by host and device code.
//
// Everything here is integer arithmetic on 0xAARRGGBB pixels (byte 0 = blue), the layout
// the reference's byte accessors assume (webp/utils/utils.cpp:55-73).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VP8L_HD __host__ __device__ __forceinline__
#else
#define VP8L_HD inline
#endif

namespace vp8l {

// Dialect constants (webp/vp8l/huffman_io.h:26-50, webp/vp8l/vp8l.h:23-26).
constexpr int kNumCodes = 5;           // GREEN, RED, BLUE, ALPHA, DIST
constexpr int kGreen = 0, kRed = 1, kBlue = 2, kAlpha = 3, kDist = 4;
constexpr int kAlphabetGreen = 280;    // 256 literals + 24 length prefixes (+ cache on decode)
constexpr int kAlphabetDist = 40;
constexpr int kMaxAlphabet = 280 + 2048;
constexpr int kCodeStride = 280;       // per-code stride of encoder-side tables
constexpr int kMaxCodeLen = 64;        // MAX_ALLOWED_CODE_LENGTH
constexpr int kMaxClCodeLen = 32;      // MAX_ALLOWED_CODE_LENGTH_OF_RLE_TREE
constexpr int kSaneCodeLen = 32;       // codes are 32-bit words in the reference (huffman_coding.h:12)
constexpr int kClRepeat = 65, kClZero11 = 66, kClZero138 = 67, kNumClSyms = 68;
constexpr int kClCountBits = 6, kClLenBits = 6;
constexpr uint32_t kWindow = 1024;     // LZ77_MAX_DISTANCE
constexpr uint32_t kMaxLen = 128;      // LZ77_MAX_LENGTH
constexpr uint32_t kMaxDim = 16384;    // MAX_ARGB_IMAGE_SIZE
constexpr uint32_t kMaxCacheBits = 11; // MAX_COLOR_CACHE_BITS
constexpr uint32_t kMaxPalette = 256;  // PALLETE_MAX_COLORS
constexpr int kExtGreenMax = kAlphabetGreen + (1 << 11);  // the green alphabet with the largest colour cache (MAX_COLOR_CACHE_BITS)
constexpr uint32_t kSmallMax = 1024;   // largest sub-image the header coder takes (a palette, or predictor modes of <= 1024 tiles)

VP8L_HD int alphabet_size(int code) { return code == kGreen ? kAlphabetGreen : code == kDist ? kAlphabetDist : 256; }

// kCodeLengthCodeOrder after init_array() (webp/vp8l/huffman_io.h:49-65)
VP8L_HD int cl_order(int i) {
  return i >= 9 ? i - 3 : i == 0 ? kClZero11 : i == 1 ? kClZero138 : i == 8 ? kClRepeat : i - 2;
}

// ---- the two bitstream dialects ---------------------------------------------------------------
// The reference writes and reads a private variant of VP8L (huffman_io.h:37-50,63-64; vp8l.h:461).  The format as
// libwebp and the WebP lossless specification define it differs in exactly these points, and nothing else on the path:
//                                   reference                         standard
//   code-length alphabet            68 symbols: lengths 0..64,        19 symbols: lengths 0..15,
//                                   65 repeat, 66 / 67 zero runs      16 repeat, 17 / 18 zero runs
//   number of code-length codes     6 bits (count - 4)                4 bits (count - 4)
//   each code-length-code length    6 bits (up to 32)                 3 bits (up to 7)
//   longest code                    64 (32 in practice)               15
//   first 32 bits of the stream     zero (never patched)              the number of bytes that follow them (RIFF chunk size)
//   predictor, last column          top-right := left (transform.h:137)   top-right := first pixel of the current row
//   plane code giving a distance < 1   wraps (unsigned arithmetic)    distance 1
//   tile grids after colour indexing   over the image's width            over the bundled width (entropy image; predictor and
//   with bundled pixels                (vp8l.h:117-126,195-196)          cross-colour tiles)
// A stream says which one it is: the reference never writes anything but zero into the first 32 bits, a standard
// stream's chunk size is never zero.
struct Dialect {
  int ncl;            // symbols of the code-length alphabet
  int count_bits;     // bits of (number of code-length codes - 4)
  int len_bits;       // bits of one code-length-code length
  int repeat, zero11, zero138;
  int max_len;        // longest code of the five pixel codes
  int max_cl_len;     // longest code of the code-length code
  VP8L_HD int order(int i) const { return i >= 9 ? i - 3 : i == 0 ? zero11 : i == 1 ? zero138 : i == 8 ? repeat : i - 2; }
};
VP8L_HD Dialect dialect_of(bool standard) {
  Dialect d;
  if (standard) { d.ncl = 19; d.count_bits = 4; d.len_bits = 3; d.repeat = 16; d.zero11 = 17; d.zero138 = 18; d.max_len = 15; d.max_cl_len = 7; }
  else { d.ncl = kNumClSyms; d.count_bits = kClCountBits; d.len_bits = kClLenBits; d.repeat = kClRepeat; d.zero11 = kClZero11; d.zero138 = kClZero138;
         d.max_len = kMaxCodeLen; d.max_cl_len = kMaxClCodeLen; }
  return d;
}

// ---- pixels (webp/vp8l/transform.cpp:8-19) -------------------------------------------------
VP8L_HD uint32_t px_add(uint32_t a, uint32_t b) {
  const uint32_t ag = (a & 0xff00ff00u) + (b & 0xff00ff00u);
  const uint32_t rb = (a & 0x00ff00ffu) + (b & 0x00ff00ffu);
  return (ag & 0xff00ff00u) | (rb & 0x00ff00ffu);
}
VP8L_HD uint32_t px_sub(uint32_t a, uint32_t b) {
  const uint32_t ag = 0x00ff00ffu + (a & 0xff00ff00u) - (b & 0xff00ff00u);
  const uint32_t rb = 0xff00ff00u + (a & 0x00ff00ffu) - (b & 0x00ff00ffu);
  return (ag & 0xff00ff00u) | (rb & 0x00ff00ffu);
}
// vp8l.h:413-416 (forward) and transform.h:94-103 (inverse): green lives in byte 1.
VP8L_HD uint32_t subtract_green(uint32_t p) {
  const uint32_t g = (p >> 8) & 0xffu;
  return (p & 0xff00ff00u) | ((p - (g << 16)) & 0x00ff0000u) | ((p - g) & 0xffu);
}
VP8L_HD uint32_t add_green(uint32_t p) {
  const uint32_t g = (p >> 8) & 0xffu;
  return (p & 0xff00ff00u) | ((p + (g << 16)) & 0x00ff0000u) | ((p + g) & 0xffu);
}
// vp8l.h:423-426 / :132-135
VP8L_HD uint32_t palette_bits(uint32_t ncolors) { return ncolors > 16 ? 0u : ncolors > 4 ? 1u : ncolors > 2 ? 2u : 3u; }
// color_cache.h:46
VP8L_HD uint32_t cache_key(uint32_t argb, uint32_t bits) { return (0x1e35a7bdu * argb) >> (32 - bits); }

VP8L_HD int floor_log2(uint32_t v) {  // v != 0
#if defined(__HIP_DEVICE_COMPILE__)
  return 31 - __clz((int)v);
#else
  return 31 - __builtin_clz(v);
#endif
}

// lz77.cpp:89-102: value >= 1 -> (prefix symbol, number of extra bits, extra bits)
VP8L_HD void prefix_encode(uint32_t value, uint32_t &sym, uint32_t &nbits, uint32_t &bits) {
  const uint32_t v = value - 1;
  if (v < 2) { sym = v; nbits = 0; bits = 0; return; }
  const int hb = floor_log2(v);
  nbits = (uint32_t)(hb - 1);
  bits = v & ((1u << nbits) - 1u);
  sym = (uint32_t)(2 * hb) + ((v >> (hb - 1)) & 1u);
}
// lz77.cpp:50-57 without the bit read: symbol -> (base value, number of extra bits)
VP8L_HD void prefix_decode_base(uint32_t sym, uint32_t &base, uint32_t &nbits) {
  if (sym < 4) { base = sym + 1; nbits = 0; return; }
  nbits = (sym - 2) >> 1;
  base = ((2 + (sym & 1)) << nbits) + 1;
}

// lz77.cpp:38-47 / :104-112.  The reference computes in size_t, so "xsize - 8" wraps for
// xsize < 8 and its second branch is then never taken.
VP8L_HD uint32_t dist_xy_to_code(uint32_t xsize, uint32_t x, uint32_t y, uint32_t dist) {  // dist = y*xsize + x, x < xsize
  constexpr uint8_t lut[128] = {
      96,  73,  55,  39,  23,  13,  5,   1,   255, 255, 255, 255, 255, 255, 255, 255,
      101, 78,  58,  42,  26,  16,  8,   2,   0,   3,   9,   17,  27,  43,  59,  79,
      102, 86,  62,  46,  32,  20,  10,  6,   4,   7,   11,  21,  33,  47,  63,  87,
      105, 90,  70,  52,  37,  28,  18,  14,  12,  15,  19,  29,  38,  53,  71,  91,
      110, 99,  82,  66,  48,  35,  30,  24,  22,  25,  31,  36,  49,  67,  83,  100,
      115, 108, 94,  76,  64,  50,  44,  40,  34,  41,  45,  51,  65,  77,  95,  109,
      118, 113, 103, 92,  80,  68,  60,  56,  54,  57,  61,  69,  81,  93,  104, 114,
      119, 116, 111, 106, 97,  88,  84,  74,  72,  75,  85,  89,  98,  107, 112, 117};
  if (x <= 8 && y < 8) return (uint32_t)lut[y * 16 + 8 - x] + 1;
  if (xsize >= 8 && x > xsize - 8 && y < 7) return (uint32_t)lut[(y + 1) * 16 + 8 + (xsize - x)] + 1;  // never for xsize < 8 (size_t wrap)
  return dist + 120;
}
VP8L_HD uint32_t dist_to_code(uint32_t xsize, uint32_t dist) {
  const uint32_t y = dist / xsize;
  return dist_xy_to_code(xsize, dist - y * xsize, y, dist);
}
// The same with the division done by a multiplication: xinv = ceil(2^32 / xsize) for xsize >= 2.
// Exact for dist * xsize < 2^32 (here dist <= 1024, xsize <= 16384).
VP8L_HD uint32_t dist_to_code_inv(uint32_t xsize, uint32_t xinv, uint32_t dist) {
  const uint32_t y = xsize == 1 ? dist : (uint32_t)(((uint64_t)dist * xinv) >> 32);
  return dist_xy_to_code(xsize, dist - y * xsize, y, dist);
}
VP8L_HD uint32_t dist_inverse(uint32_t xsize) { return xsize >= 2 ? (uint32_t)((0x100000000ull + xsize - 1) / xsize) : 0u; }

// lz77.cpp:20-35 / :59-70 in uint32 arithmetic like the reference (a negative dx with a small
// dy*xsize wraps to a huge distance; it is not clamped to 1).  The table is the inverse of the
// one in dist_to_code: entry (code-1) = dy*16 + (8 - dx).
VP8L_HD uint32_t code_to_dist(uint32_t xsize, uint32_t code, bool standard = false) {
  constexpr uint8_t inv[120] = {
      0x18, 0x07, 0x17, 0x19, 0x28, 0x06, 0x27, 0x29, 0x16, 0x1a, 0x26, 0x2a, 0x38, 0x05, 0x37, 0x39,
      0x15, 0x1b, 0x36, 0x3a, 0x25, 0x2b, 0x48, 0x04, 0x47, 0x49, 0x14, 0x1c, 0x35, 0x3b, 0x46, 0x4a,
      0x24, 0x2c, 0x58, 0x45, 0x4b, 0x34, 0x3c, 0x03, 0x57, 0x59, 0x13, 0x1d, 0x56, 0x5a, 0x23, 0x2d,
      0x44, 0x4c, 0x55, 0x5b, 0x33, 0x3d, 0x68, 0x02, 0x67, 0x69, 0x12, 0x1e, 0x66, 0x6a, 0x22, 0x2e,
      0x54, 0x5c, 0x43, 0x4d, 0x65, 0x6b, 0x32, 0x3e, 0x78, 0x01, 0x77, 0x79, 0x53, 0x5d, 0x11, 0x1f,
      0x64, 0x6c, 0x42, 0x4e, 0x76, 0x7a, 0x21, 0x2f, 0x75, 0x7b, 0x31, 0x3f, 0x63, 0x6d, 0x52, 0x5e,
      0x00, 0x74, 0x7c, 0x41, 0x4f, 0x10, 0x20, 0x62, 0x6e, 0x30, 0x73, 0x7d, 0x51, 0x5f, 0x40, 0x72,
      0x7e, 0x61, 0x6f, 0x50, 0x71, 0x7f, 0x60, 0x70};
  if (code <= 120) {
    const uint32_t e = inv[code - 1];
    const uint32_t d = (uint32_t)(8 - (int32_t)(e & 15)) + (e >> 4) * xsize;
    if (standard && (int32_t)d < 1) return 1;  // signed arithmetic there: a plane code that points at or before the pixel itself means distance 1
    return d >= 1 ? d : 1;
  }
  return code - 120;
}

}  // namespace vp8l

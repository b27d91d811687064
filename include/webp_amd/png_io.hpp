// png_io.hpp -- PNG input and output for the command line tool, on zlib alone.
//
// The reference reads PNG through libpng (read_png, webp.cpp:16-117) and writes it through libpng
// (WebP_DECODER::save2png, webp/webp.h:83-144); libpng's headers are not part of this image, so the
// two functions are restated here on top of zlib's inflate/deflate.  read_png() applies the same
// transformation chain the reference asks libpng for (webp.cpp:60-77):
//   16-bit samples -> their high byte, packed 1/2/4-bit samples -> one byte each,
//   palette -> RGB, grey of less than 8 bits -> scaled to 8 bits, grey -> RGB,
//   a tRNS chunk -> an alpha channel (alpha 0 for the transparent colour / the palette's alpha),
//   Adam7 interlacing resolved,
// and then fills 0xAARRGGBB pixels exactly as the reference does (webp.cpp:93-110) -- including its
// byte order for images WITH alpha, where the four bytes of a pixel (R,G,B,A as libpng delivers them)
// are taken as A,R,G,B.  write_png() writes 8-bit RGB, not interlaced, like save2png (alpha dropped).
// No ancillary chunk is interpreted (no gamma, as in the reference's default libpng set-up).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <string>
#include <vector>

namespace webp {
namespace png_io {

struct Error {
  std::string message;
};

namespace detail {
inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline void put_be32(std::vector<uint8_t> &v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
// Undoes the scanline filters of one (sub-)image in place; `raw` holds height rows of 1 + rowbytes bytes.
inline void unfilter(uint8_t *raw, size_t height, size_t rowbytes, size_t bpp) {
  std::vector<uint8_t> zero(rowbytes, 0);
  const uint8_t *prev = zero.data();
  for (size_t y = 0; y < height; y++) {
    uint8_t *row = raw + y * (rowbytes + 1);
    const int ft = row[0];
    uint8_t *cur = row + 1;
    for (size_t i = 0; i < rowbytes; i++) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int v = cur[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: throw Error{"PNG: bad filter type"};
      }
      cur[i] = (uint8_t)v;
    }
    prev = cur;
  }
}
}  // namespace detail

// Decoded image after the reference's transformation chain: 3 or 4 bytes per pixel.
struct Image {
  uint32_t width = 0, height = 0;
  bool has_alpha = false;
  std::vector<uint8_t> bytes;  // row-major, (has_alpha ? 4 : 3) bytes per pixel
};

// Inflates `in` into exactly out.size() bytes.  zlib counts in 32 bits, so both sides are offered in pieces of at most
// `piece` bytes (a parameter so that a test can force many pieces).  Never Z_FINISH: with less output room than the data
// needs, zlib turns a call that made progress into Z_BUF_ERROR.  Ends when the output is full, the stream ends, zlib
// reports a real error, or a call moves neither side.
inline bool inflate_exactly(const std::vector<uint8_t> &in, std::vector<uint8_t> &out, size_t piece) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit(&zs) != Z_OK) return false;
  if (piece == 0 || piece > ((size_t)1 << 30)) piece = (size_t)1 << 30;
  size_t in_pos = 0, got = 0;
  bool ok = true;
  while (got < out.size()) {
    if (zs.avail_in == 0 && in_pos < in.size()) {
      const size_t n = std::min(piece, in.size() - in_pos);
      zs.next_in = const_cast<uint8_t *>(in.data()) + in_pos;
      zs.avail_in = (uInt)n;
      in_pos += n;
    }
    const size_t room = std::min(piece, out.size() - got);
    const uInt in_before = zs.avail_in;
    zs.next_out = out.data() + got;
    zs.avail_out = (uInt)room;
    const int rc = inflate(&zs, Z_NO_FLUSH);
    got += room - zs.avail_out;
    if (rc == Z_STREAM_END) break;
    if (rc != Z_OK && rc != Z_BUF_ERROR) { ok = false; break; }
    if (zs.avail_out == room && zs.avail_in == in_before && in_pos == in.size()) break;  // nothing moved, nothing left to offer
  }
  inflateEnd(&zs);
  return ok && got == out.size();
}

inline Image read_png_bytes(const std::vector<uint8_t> &file, size_t inflate_piece = (size_t)1 << 30) {
  using namespace detail;
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) throw Error{"PNG: bad signature"};
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  bool have_trns = false, have_ihdr = false;
  size_t pos = 8;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const uint8_t *type = &file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) throw Error{"PNG: truncated chunk"};
    const uint8_t *data = &file[pos + 8];
    if (!memcmp(type, "IHDR", 4)) {
      if (len < 13) throw Error{"PNG: bad IHDR"};
      w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
      have_ihdr = true;
      // Nothing is sized by the header's untrusted numbers beyond what the encoder accepts (vp8l.h:23-26,589-590).  The
      // text is what the reference's tool prints for such a file (TooBigARGBImage prints from its constructor and main adds
      // an empty line, exception.h:144-149, webp.cpp:186-188), so the command line tool's output stays the same.
      if (w > 16384 || h > 16384) throw Error{"Too big ARGB Image. Max allowed width(height)=16384\n"};
    } else if (!memcmp(type, "PLTE", 4)) {
      plte.assign(data, data + len);
    } else if (!memcmp(type, "tRNS", 4)) {
      trns.assign(data, data + len);
      have_trns = true;
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || w == 0 || h == 0) throw Error{"PNG: no image header"};
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw Error{"PNG: bad colour type"};
  }
  if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) throw Error{"PNG: bad bit depth"};
  if (ctype == 3 && (depth == 16 || plte.empty())) throw Error{"PNG: bad palette image"};
  if ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8) throw Error{"PNG: bad bit depth for the colour type"};
  const size_t bits_pp = (size_t)channels * depth;
  const size_t filt_bpp = bits_pp >= 8 ? bits_pp / 8 : 1;

  // sizes of the (up to seven) passes
  struct Pass { uint32_t x0, y0, dx, dy; };
  static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  static const Pass whole = {0, 0, 1, 1};
  const int npass = interlace ? 7 : 1;
  size_t raw_size = 0;
  for (int p = 0; p < npass; p++) {
    const Pass &ps = interlace ? adam7[p] : whole;
    const size_t pw = w > ps.x0 ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0, ph = h > ps.y0 ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
    if (pw && ph) raw_size += ph * (1 + (pw * bits_pp + 7) / 8);
  }
  std::vector<uint8_t> raw(raw_size);
  if (!inflate_exactly(idat, raw, inflate_piece)) throw Error{"PNG: bad image data"};

  // samples at file depth: channels x uint16 per pixel
  std::vector<uint16_t> smp((size_t)w * h * channels);
  size_t off = 0;
  for (int p = 0; p < npass; p++) {
    const Pass &ps = interlace ? adam7[p] : whole;
    const size_t pw = w > ps.x0 ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0, ph = h > ps.y0 ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
    if (!pw || !ph) continue;
    const size_t rowbytes = (pw * bits_pp + 7) / 8;
    unfilter(raw.data() + off, ph, rowbytes, filt_bpp);
    for (size_t py = 0; py < ph; py++) {
      const uint8_t *row = raw.data() + off + py * (rowbytes + 1) + 1;
      const size_t y = ps.y0 + py * ps.dy;
      for (size_t pxi = 0; pxi < pw; pxi++) {
        const size_t x = ps.x0 + pxi * ps.dx;
        uint16_t *o = &smp[(y * w + x) * channels];
        for (int c = 0; c < channels; c++) {
          const size_t s = pxi * channels + c;
          if (depth == 8) o[c] = row[s];
          else if (depth == 16) o[c] = (uint16_t)((row[2 * s] << 8) | row[2 * s + 1]);
          else {
            const size_t bit = s * depth;
            o[c] = (uint16_t)((row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1));
          }
        }
      }
    }
    off += ph * (rowbytes + 1);
  }

  Image im;
  im.width = w;
  im.height = h;
  im.has_alpha = have_trns || (ctype & 4);  // webp.cpp:71-76
  const int oc = im.has_alpha ? 4 : 3;
  im.bytes.resize((size_t)w * h * oc);
  const auto to8 = [&](uint16_t v) -> uint8_t {  // strip_16 keeps the high byte; grey below 8 bits is scaled up
    if (depth == 16) return (uint8_t)(v >> 8);
    if (depth == 8) return (uint8_t)v;
    return (uint8_t)(v * 255 / ((1 << depth) - 1));
  };
  uint16_t tkey[3] = {0, 0, 0};
  if (have_trns && ctype == 0 && trns.size() >= 2) tkey[0] = (uint16_t)((trns[0] << 8) | trns[1]);
  if (have_trns && ctype == 2 && trns.size() >= 6)
    for (int c = 0; c < 3; c++) tkey[c] = (uint16_t)((trns[2 * c] << 8) | trns[2 * c + 1]);
  for (size_t i = 0; i < (size_t)w * h; i++) {
    const uint16_t *s = &smp[i * channels];
    uint8_t r, g, b, a = 255;
    switch (ctype) {
      case 0:
        r = g = b = to8(s[0]);
        if (have_trns && s[0] == tkey[0]) a = 0;
        break;
      case 2:
        r = to8(s[0]); g = to8(s[1]); b = to8(s[2]);
        if (have_trns && s[0] == tkey[0] && s[1] == tkey[1] && s[2] == tkey[2]) a = 0;
        break;
      case 3: {
        const size_t k = s[0];
        if (3 * k + 2 >= plte.size()) throw Error{"PNG: palette index out of range"};
        r = plte[3 * k]; g = plte[3 * k + 1]; b = plte[3 * k + 2];
        if (have_trns && k < trns.size()) a = trns[k];
        break;
      }
      case 4:
        r = g = b = to8(s[0]);
        a = to8(s[1]);
        break;
      default:
        r = to8(s[0]); g = to8(s[1]); b = to8(s[2]); a = to8(s[3]);
        break;
    }
    uint8_t *o = &im.bytes[i * oc];
    o[0] = r; o[1] = g; o[2] = b;
    if (oc == 4) o[3] = a;
  }
  return im;
}

// The reference's pixel fill (webp.cpp:93-110).  With alpha, the four bytes of a pixel are taken as A,R,G,B
// although they are R,G,B,A: kept, so that both tools produce the same stream from the same file.
inline void to_argb(const Image &im, std::vector<uint32_t> &argb) {
  argb.resize((size_t)im.width * im.height);
  for (size_t i = 0; i < argb.size(); i++) {
    if (im.has_alpha) {
      const uint8_t *p = &im.bytes[4 * i];
      argb[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    } else {
      const uint8_t *p = &im.bytes[3 * i];
      argb[i] = 0xff000000u | ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | p[2];
    }
  }
}

inline std::vector<uint8_t> read_file(const std::string &name) {
  FILE *fp = fopen(name.c_str(), "rb");
  if (!fp) throw Error{"File operation exception"};
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t n;
  while ((n = fread(tmp, 1, sizeof(tmp), fp)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  fclose(fp);
  return buf;
}

// save2png (webp.h:83-144): 8-bit RGB from 0xAARRGGBB pixels, not interlaced.
inline void write_png(const std::string &name, const uint32_t *argb, uint32_t w, uint32_t h) {
  using namespace detail;
  std::vector<uint8_t> raw((size_t)h * (1 + 3 * (size_t)w));
  for (uint32_t y = 0; y < h; y++) {
    uint8_t *row = &raw[(size_t)y * (1 + 3 * (size_t)w)];
    row[0] = 0;  // filter: none
    for (uint32_t x = 0; x < w; x++) {
      const uint32_t p = argb[(size_t)y * w + x];
      row[1 + 3 * x] = (uint8_t)(p >> 16); row[2 + 3 * x] = (uint8_t)(p >> 8); row[3 + 3 * x] = (uint8_t)p;
    }
  }
  uLongf clen = compressBound((uLong)raw.size());
  std::vector<uint8_t> comp(clen);
  if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw Error{"PNG Error"};
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  const auto chunk = [&](const char *type, const uint8_t *data, size_t len) {
    put_be32(out, (uint32_t)len);
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), data, data + len);
    put_be32(out, (uint32_t)crc32(0, &out[start], (uInt)(out.size() - start)));
  };
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, w); put_be32(ihdr, h);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk("IHDR", ihdr.data(), ihdr.size());
  chunk("IDAT", comp.data(), clen);
  chunk("IEND", nullptr, 0);
  FILE *fp = fopen(name.c_str(), "wb");
  if (!fp) throw Error{"File operation exception"};
  fwrite(out.data(), 1, out.size(), fp);
  fclose(fp);
}

}  // namespace png_io
}  // namespace webp

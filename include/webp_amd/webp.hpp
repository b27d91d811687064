// webp.hpp -- C++ host-side mirror of the reference's codec interface on top of the C ABI
// (include/vp8l_amd.h).  Same namespaces, class names, constructor signatures, accessors and
// exception types as the reference, so code written against
//     webp/vp8l/vp8l.h   (VP8_LOSSLESS_ENCODER :585-615, VP8_LOSSLESS_DECODER :330-351)
//     webp/webp.h        (WebP_ENCODER :152-175, WebP_DECODER :20-149, RIFF container)
//     webp/utils/utils.h (array<T> :12-107), webp/utils/bit_writer.h (BitWriter :20-92)
//     webp/exception/exception.h
// compiles against this header instead; all codec work runs on the GPU behind the C ABI.
// Header-only; link with libvp8l_amd.so.  WebP_DECODER::save2png (webp.h:83-144) writes its PNG
// through png_io.hpp (zlib instead of libpng: a program that calls it links with -lz); read_png
// is part of the reference's command line tool (webp.cpp:16-117), not of this interface, and
// has its counterpart in png_io.hpp as well.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <iostream>
#include <string>
#include <vector>

#include "../vp8l_amd.h"
#include "png_io.hpp"

namespace webp {

namespace exception {  // webp/exception/exception.h:14-153 (same messages)
class Exception {
 public:
  std::string message;
  Exception() : message("Exception") {}
  virtual ~Exception() {}
};
#define WEBP_AMD_EXC(name, text)                    \
  class name : public Exception {                   \
   public:                                          \
    name() { message = text; }                      \
  }
WEBP_AMD_EXC(FileOperationException, "File operation exception");
WEBP_AMD_EXC(MemoryAllocationException, "Memory allocation exception");
WEBP_AMD_EXC(InvalidWebPFileFormat, "Invalid WebP File format");
WEBP_AMD_EXC(UnsupportedVP8, "VP8 is unsupported");
WEBP_AMD_EXC(InvalidVP8L, "Invalid VP8L");
WEBP_AMD_EXC(InvalidHuffman, "Invalid format");
WEBP_AMD_EXC(InvalidARGBImage, "Invalid ARGB image");
WEBP_AMD_EXC(PNGError, "PNG error");
#undef WEBP_AMD_EXC
class TooBigCodeLength : public Exception {  // exception.h:121-131 prints instead of setting a message
 public:
  TooBigCodeLength(uint32_t max_allowed_code_length, uint32_t max_code_length) {
    std::cout << "Too big Huffman code length(" << max_code_length << "). Max allowed code length = " << max_allowed_code_length << std::endl;
  }
};
class TooBigARGBImage : public Exception {  // exception.h:144-153
 public:
  explicit TooBigARGBImage(size_t max_size) { std::cout << "Too big ARGB Image. Max allowed width(height)=" << max_size << std::endl; }
};
// No reference counterpart: the GPU or the HIP runtime is unavailable / failed.
class DeviceError : public Exception {
 public:
  explicit DeviceError(int status) { message = std::string(vp8l_status_string(status)) + ": " + vp8l_last_error(); }
};

inline void throw_status(int status) {  // C ABI status -> the reference's exception type
  switch (status) {
    case VP8L_OK: return;
    case VP8L_ERR_INVALID_ARGB: throw InvalidARGBImage();
    case VP8L_ERR_TOO_BIG_ARGB: throw TooBigARGBImage(16384);
    case VP8L_ERR_TOO_BIG_CODE_LENGTH: throw TooBigCodeLength(64, 65);
    case VP8L_ERR_UNSUPPORTED_VP8: throw UnsupportedVP8();
    case VP8L_ERR_INVALID_VP8L: throw InvalidVP8L();
    case VP8L_ERR_INVALID_HUFFMAN: throw InvalidHuffman();
    case VP8L_ERR_NO_MEMORY: throw MemoryAllocationException();
    default: throw DeviceError(status);
  }
}
}  // namespace exception

namespace utils {
// webp/utils/utils.h:12-107
template <class T>
class array {
  std::vector<T> m_;

 public:
  array() {}
  explicit array(const uint32_t &size) : m_(size, T()) {}
  void realloc(const uint32_t &size) { m_.assign(size, T()); }
  T &operator[](const size_t &i) { return m_[i]; }
  const T &operator[](const size_t &i) const { return m_[i]; }
  const size_t size() const { return m_.size(); }
  void fill(int c) { if (!m_.empty()) memset(m_.data(), c, m_.size() * sizeof(T)); }
  T *data() { return m_.data(); }
  const T *data() const { return m_.data(); }
};
typedef array<uint32_t> pixel_array;
typedef array<uint8_t> byte_array;

// webp/utils/utils.cpp:55-73: pointers to the bytes of a 0xAARRGGBB pixel in memory (little endian), which the
// reference's read_png writes through (webp.cpp:99-108).
inline uint8_t *ALPHA(const uint32_t &argb) { return (uint8_t *)(&argb) + 3; }
inline uint8_t *RED(const uint32_t &argb) { return (uint8_t *)(&argb) + 2; }
inline uint8_t *GREEN(const uint32_t &argb) { return (uint8_t *)(&argb) + 1; }
inline uint8_t *BLUE(const uint32_t &argb) { return (uint8_t *)(&argb); }

// The encoder's output as the reference exposes it (bit_writer.h:65-90): size() bytes,
// save2file() APPENDS to the file.
class BitWriter {
  std::vector<uint8_t> bytes_;

 public:
  void assign(const uint8_t *p, size_t n) { bytes_.assign(p, p + n); }
  const size_t size() const { return bytes_.size(); }
  const uint8_t *data() const { return bytes_.data(); }
  void save2file(const std::string &file_name) const {
    FILE *fp = fopen(file_name.c_str(), "ab+");
    if (!fp) throw exception::FileOperationException();
    if (!bytes_.empty()) fwrite(bytes_.data(), bytes_.size(), 1, fp);
    fclose(fp);
  }
};
}  // namespace utils

namespace vp8l {
namespace huffman_io {
// The reference's main() must call this before any codec use (webp.cpp:128, huffman_io.h:53-65).
// Kept for source compatibility; the table it fills there is a compile-time constant here.
inline void init_array() { (void)vp8l_init(); }
}  // namespace huffman_io

class VP8_LOSSLESS_ENCODER {  // vp8l.h:356-616
  utils::BitWriter m_bit_writer;

 public:
  VP8_LOSSLESS_ENCODER(const utils::pixel_array &argb_image, const size_t &width, const size_t &height) {
    if (argb_image.size() == 0 || width == 0 || height == 0) throw exception::InvalidARGBImage();  // :587-588
    if (width > 16384 || height > 16384) throw exception::TooBigARGBImage(16384);                  // :589-590
    if (argb_image.size() < width * height) throw exception::InvalidARGBImage();
    uint8_t *out = nullptr;
    size_t len = 0;
    exception::throw_status(vp8l_encode(argb_image.data(), (uint32_t)width, (uint32_t)height, &out, &len, 0));
    m_bit_writer.assign(out, len);
    vp8l_free(out);
  }
  const utils::BitWriter &get_bit_writer() { return m_bit_writer; }
};

class VP8_LOSSLESS_DECODER {  // vp8l.h:30-354
  uint32_t m_image_width, m_image_height;

 public:
  VP8_LOSSLESS_DECODER(const uint8_t *const data, uint32_t data_length, utils::pixel_array &argb_image)
      : m_image_width(0), m_image_height(0) {
    uint32_t *px = nullptr;
    exception::throw_status(vp8l_decode(data, data_length, &px, &m_image_width, &m_image_height));
    argb_image.realloc(m_image_width * m_image_height);
    memcpy(argb_image.data(), px, sizeof(uint32_t) * (size_t)m_image_width * m_image_height);
    vp8l_free(px);
  }
  const uint32_t image_width() { return m_image_width; }
  const uint32_t image_height() { return m_image_height; }
};
}  // namespace vp8l

// RIFF container, webp/webp.h:152-175: "RIFF", u32 (stream size + 8), "WEBP", "VP8L", stream.
class WebP_ENCODER {
 public:
  WebP_ENCODER(const utils::pixel_array &argb_image, const size_t &width, const size_t &height, const std::string &output) {
    vp8l::VP8_LOSSLESS_ENCODER encoder(argb_image, width, height);
    FILE *fp = fopen(output.c_str(), "wb");
    if (!fp) throw exception::FileOperationException();
    fwrite("RIFF", 1, 4, fp);
    uint32_t filesize = (uint32_t)encoder.get_bit_writer().size() + 8;
    fwrite(&filesize, 4, 1, fp);
    fwrite("WEBP", 1, 4, fp);
    fwrite("VP8L", 1, 4, fp);
    fclose(fp);
    encoder.get_bit_writer().save2file(output);
  }
};

// webp/webp.h:20-149; the decoded pixels are exposed as well (argb()).
class WebP_DECODER {
  utils::pixel_array m_argb_image;
  uint32_t m_image_width, m_image_height;

 public:
  explicit WebP_DECODER(const std::string &file_name) : m_image_width(0), m_image_height(0) {
    FILE *fp = fopen(file_name.c_str(), "rb");
    if (!fp) throw exception::FileOperationException();
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), fp)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(fp);
    if (buf.size() < 12) throw exception::InvalidWebPFileFormat();                   // :79-80
    if (memcmp(buf.data(), "RIFF", 4) != 0) throw exception::InvalidWebPFileFormat();  // :34-35
    uint32_t file_size;
    memcpy(&file_size, buf.data() + 4, 4);
    if (memcmp(buf.data() + 8, "WEBP", 4) != 0) throw exception::InvalidWebPFileFormat();  // :41-42
    if (buf.size() < 16 || memcmp(buf.data() + 12, "VP8L", 4) != 0) throw exception::UnsupportedVP8();  // :51-56,70-71
    uint32_t vp8_len = file_size - 8;  // :61
    if ((size_t)vp8_len > buf.size() - 16) vp8_len = (uint32_t)(buf.size() - 16);  // the reference reads past the buffer here
    vp8l::VP8_LOSSLESS_DECODER decoder(buf.data() + 16, vp8_len, m_argb_image);
    m_image_width = decoder.image_width();
    m_image_height = decoder.image_height();
  }
  // webp.h:83-144: 8-bit RGB, not interlaced, alpha dropped.  A file that cannot be opened is a
  // FileOperationException, any other failure a PNGError, as there.
  void save2png(const std::string &file_name) {
    FILE *fp = fopen(file_name.c_str(), "wb");
    if (!fp) throw exception::FileOperationException();
    fclose(fp);
    try {
      png_io::write_png(file_name, m_argb_image.data(), m_image_width, m_image_height);
    } catch (const png_io::Error &) {
      throw exception::PNGError();
    }
  }
  const utils::pixel_array &argb() const { return m_argb_image; }
  uint32_t image_width() const { return m_image_width; }
  uint32_t image_height() const { return m_image_height; }
};

}  // namespace webp

// Exercises the C++ mirror of the reference interface (include/webp_amd/webp.hpp) the way the
// reference's own main() uses its classes (webp.cpp:126-190): init_array, encode to a .webp file,
// decode it again and save the picture as PNG (webp.cpp:173-184).  Writes the raw VP8L stream next
// to it so the Python test can compare it with the oracle.
// Usage: shim_test <width> <height> <mode 0=gradient 1=two-colour> <out.webp> <out.vp8l> <out.png>
#include <stdio.h>
#include <stdlib.h>

#include "webp_amd/webp.hpp"

int main(int argc, char **argv) {
  if (argc != 7) return 2;
  const size_t w = (size_t)atoi(argv[1]), h = (size_t)atoi(argv[2]);
  const int mode = atoi(argv[3]);
  webp::vp8l::huffman_io::init_array();
  try {
    webp::utils::pixel_array argb((uint32_t)(w * h));
    for (size_t y = 0; y < h; y++)
      for (size_t x = 0; x < w; x++)
        argb[y * w + x] = mode == 0 ? 0xff000000u | (uint32_t)((x * 7 + y * 3) & 0xff) << 16 | (uint32_t)((x + y) & 0xff) << 8 | (uint32_t)((x ^ y) & 0xff)
                                    : (((x / 3 + y / 2) & 1) ? 0xff112233u : 0x80ffeeddu);
    remove(argv[4]);
    remove(argv[5]);
    webp::WebP_ENCODER enc(argb, w, h, argv[4]);
    webp::vp8l::VP8_LOSSLESS_ENCODER raw(argb, w, h);
    raw.get_bit_writer().save2file(argv[5]);
    {  // the two statements of the reference's decode branch, webp.cpp:175-176
      webp::WebP_DECODER webp(argv[4]);
      webp.save2png(argv[6]);
    }
    webp::WebP_DECODER dec(argv[4]);
    if (dec.image_width() != w || dec.image_height() != h) { printf("size mismatch\n"); return 1; }
    for (size_t i = 0; i < w * h; i++)
      if (dec.argb()[i] != argb[i]) { printf("pixel %zu differs\n", i); return 1; }
    // error mapping: same exception types as the reference
    int caught = 0;
    try { webp::utils::pixel_array e; webp::vp8l::VP8_LOSSLESS_ENCODER bad(e, 0, 0); } catch (webp::exception::InvalidARGBImage &x) { caught += x.message == "Invalid ARGB image"; }
    try { webp::utils::pixel_array big(16385); webp::vp8l::VP8_LOSSLESS_ENCODER bad(big, 16385, 1); } catch (webp::exception::TooBigARGBImage &) { caught++; }
    try { uint8_t junk[16] = {0, 0, 0, 0, 0x2e}; webp::utils::pixel_array o; webp::vp8l::VP8_LOSSLESS_DECODER bad(junk, 16, o); } catch (webp::exception::UnsupportedVP8 &x) { caught += x.message == "VP8 is unsupported"; }
    try { webp::WebP_DECODER d(argv[4]); d.save2png(std::string(argv[6]) + ".no-such-dir/x.png"); } catch (webp::exception::FileOperationException &x) { caught += x.message == "File operation exception"; }
    if (caught != 4) { printf("exception mapping failed (%d)\n", caught); return 1; }
    printf("shim ok %zu bytes\n", raw.get_bit_writer().size());
    return 0;
  } catch (webp::exception::Exception &e) {
    printf("exception: %s\n", e.message.c_str());
    return 3;
  }
}

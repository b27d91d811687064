// png_dump in.png out.argb -- test helper: the pixels read_png gives (0xAARRGGBB, after the reference's fill)
// as width, height (uint32 each) followed by the pixels.  Host only; used by tests/test_png_io.py.
// png_dump in.png out.argb PIECE: the same with zlib fed PIECE bytes at a time (the many-pieces path of the reader).
#include <stdio.h>

#include "webp_amd/png_io.hpp"

int main(int argc, char **argv) {
  if (argc == 4 && std::string(argv[1]) == "-w") {  // png_dump -w in.argb out.png : the writer
    std::vector<uint8_t> f = webp::png_io::read_file(argv[2]);
    uint32_t wh[2];
    memcpy(wh, f.data(), 8);
    webp::png_io::write_png(argv[3], reinterpret_cast<const uint32_t *>(f.data() + 8), wh[0], wh[1]);
    return 0;
  }
  if (argc != 3 && argc != 4) return 2;
  try {
    const size_t piece = argc == 4 ? (size_t)strtoull(argv[3], nullptr, 10) : (size_t)1 << 30;
    const webp::png_io::Image im = webp::png_io::read_png_bytes(webp::png_io::read_file(argv[1]), piece);
    std::vector<uint32_t> argb;
    webp::png_io::to_argb(im, argb);
    FILE *fp = fopen(argv[2], "wb");
    if (!fp) return 3;
    const uint32_t wh[2] = {im.width, im.height};
    fwrite(wh, 4, 2, fp);
    fwrite(argb.data(), 4, argb.size(), fp);
    fclose(fp);
  } catch (webp::png_io::Error &e) {
    fprintf(stderr, "%s\n", e.message.c_str());
    return 1;
  }
  return 0;
}

// webp_ -- the reference's command line tool (webp.cpp:119-188) on the MI355X codec core:
//   webp_ -e input.png output.webp     encode
//   webp_ -d input.webp output.png     decode
//   webp_ -h
// Same options, messages and return codes; PNG goes through include/webp_amd/png_io.hpp (zlib) instead of libpng.
#include <iostream>
#include <string>

#include "webp_amd/png_io.hpp"
#include "webp_amd/webp.hpp"

static void print_help() {
  std::cout << "WebP Decoded/Encoder\n";
  std::cout << "\t-h - this help\n";
  std::cout << "\t-d|-e input_file_name output_file_name - decode|encode input file to output file\n";
}

int main(int argc, char *argv[]) {
  (void)argc;
  webp::vp8l::huffman_io::init_array();
  std::string input, output;
  bool encode = false, decode = false;
  for (++argv; argv[0]; ++argv) {
    if (argv[0] == std::string("-d")) decode = true;
    else if (argv[0] == std::string("-e")) encode = true;
    else if (argv[0] == std::string("-h")) { print_help(); return 0; }
    else if (input.empty()) input = argv[0];
    else if (output.empty()) output = argv[0];
    else { printf("What you mean? %s\n", argv[0]); print_help(); return 1; }
  }
  if ((encode && decode) || (!encode && !decode)) { printf("Specify key -d or -e\n"); print_help(); return 1; }
  if (input.empty()) { printf("Specify input file name\n"); print_help(); return 1; }
  if (output.empty()) { printf("Specify output file name\n"); print_help(); return 1; }
  try {
    if (decode) {
      webp::WebP_DECODER webp(input);
      webp::png_io::write_png(output, webp.argb().data(), webp.image_width(), webp.image_height());
      return 0;
    }
    const webp::png_io::Image png = webp::png_io::read_png_bytes(webp::png_io::read_file(input));
    std::vector<uint32_t> argb;
    webp::png_io::to_argb(png, argb);
    webp::utils::pixel_array image((uint32_t)argb.size());
    for (size_t i = 0; i < argb.size(); i++) image[i] = argb[i];
    webp::WebP_ENCODER encoder(image, png.width, png.height, output);
    return 0;
  } catch (webp::exception::Exception &e) {
    std::cout << e.message << std::endl;
  } catch (webp::png_io::Error &e) {
    std::cout << e.message << std::endl;
  }
  return 0;
}

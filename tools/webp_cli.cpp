// webp_ -- command line front end of the MI355X codec core, with the observable behaviour of the reference's tool
// (webp.cpp:119-188): the same options, messages and exit codes.
//   webp_ -e input.png output.webp     encode
//   webp_ -d input.webp output.png     decode
//   webp_ -h
// PNG goes through include/webp_amd/png_io.hpp (zlib) instead of libpng.
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "webp_amd/png_io.hpp"
#include "webp_amd/webp.hpp"

namespace {

enum Direction { kNone = 0, kEncode = 1, kDecode = 2 };

struct Request {
  unsigned directions = 0;          // bit set of Direction values seen on the command line
  std::vector<std::string> files;   // positional arguments in order: input, output
};

// What the scan of the command line decided: run the request, or leave with `code` after printing `complaint`
// (if any) and the usage text.
struct Verdict {
  bool run;
  int code;
  std::string complaint;
};

const struct { const char *text; Direction dir; } kSwitches[] = {{"-e", kEncode}, {"-d", kDecode}};

void usage() {
  static const char *const lines[] = {"WebP Decoded/Encoder", "\t-h - this help",
                                      "\t-d|-e input_file_name output_file_name - decode|encode input file to output file"};
  for (const char *l : lines) std::cout << l << "\n";
}

// The reference acts on arguments strictly left to right: "-h" ends the run on the spot with success, a third file
// name ends it with failure, whatever follows.  The remaining checks happen once everything has been read.
Verdict scan(int argc, char **argv, Request &req) {
  for (int i = 1; i < argc; i++) {
    const char *a = argv[i];
    if (!std::strcmp(a, "-h")) return {false, 0, ""};
    bool is_switch = false;
    for (const auto &s : kSwitches)
      if (!std::strcmp(a, s.text)) { req.directions |= s.dir; is_switch = true; }
    if (is_switch) continue;
    if (req.files.size() == 2) return {false, 1, std::string("What you mean? ") + a};
    // the reference tests its two slots with size() == 0, so an empty argument fills nothing
    if (*a) req.files.push_back(a);
  }
  if (req.directions != kEncode && req.directions != kDecode) return {false, 1, "Specify key -d or -e"};
  if (req.files.empty()) return {false, 1, "Specify input file name"};
  if (req.files.size() < 2) return {false, 1, "Specify output file name"};
  return {true, 0, ""};
}

void decode_to_png(const std::string &webp_path, const std::string &png_path) {
  webp::WebP_DECODER decoded(webp_path);
  decoded.save2png(png_path);  // webp.cpp:175-176
}

void encode_from_png(const std::string &png_path, const std::string &webp_path) {
  const webp::png_io::Image png = webp::png_io::read_png_bytes(webp::png_io::read_file(png_path));
  std::vector<uint32_t> argb;
  webp::png_io::to_argb(png, argb);
  webp::utils::pixel_array pixels((uint32_t)argb.size());
  for (size_t i = 0; i < argb.size(); i++) pixels[i] = argb[i];
  webp::WebP_ENCODER(pixels, png.width, png.height, webp_path);  // writes the file from its constructor (webp.h:154-174)
}

}  // namespace

int main(int argc, char **argv) {
  webp::vp8l::huffman_io::init_array();
  Request req;
  const Verdict v = scan(argc, argv, req);
  if (!v.run) {
    if (v.code != 0) std::printf("%s\n", v.complaint.c_str());
    std::fflush(stdout);
    usage();
    return v.code;
  }
  // A codec or file failure prints its message and still exits with 0, as the reference does (webp.cpp:186-189).
  try {
    if (req.directions == kDecode) decode_to_png(req.files[0], req.files[1]);
    else encode_from_png(req.files[0], req.files[1]);
  } catch (const webp::exception::Exception &e) {
    std::cout << e.message << std::endl;
  } catch (const webp::png_io::Error &e) {
    std::cout << e.message << std::endl;
  }
  return 0;
}

/*
 * vp8l_amd.h -- C ABI of the MI355X VP8L (lossless WebP, reference dialect) codec core.
 *
 * This is the drop-in boundary for the reference's webp/vp8l, webp/lz77 and
 * webp/huffman_coding subsystems (RuslanKutdusov/WebP).  Plain pointers and sizes only.
 * Streams are exactly the bytes the reference's utils::BitWriter holds after
 * VP8_LOSSLESS_ENCODER's constructor (vp8l.h:585-612): four zero "length" bytes, 0x2F, the
 * 14+14+1+3 bit header, transforms, codes, pixels -- no RIFF container.
 *
 * All work runs on the GPU (gfx950).  There is no CPU fallback: without a usable device
 * every entry point returns VP8L_ERR_NO_DEVICE.
 */
#ifndef VP8L_AMD_H_
#define VP8L_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One code per reference exception type (webp/exception/exception.h:14-153). */
typedef enum {
  VP8L_OK = 0,
  VP8L_ERR_INVALID_ARGB = -1,        /* exception::InvalidARGBImage   vp8l.h:587-588            */
  VP8L_ERR_TOO_BIG_ARGB = -2,        /* exception::TooBigARGBImage    vp8l.h:589-590            */
  VP8L_ERR_TOO_BIG_CODE_LENGTH = -3, /* exception::TooBigCodeLength   huffman_coding.h:410-413  */
  VP8L_ERR_UNSUPPORTED_VP8 = -4,     /* exception::UnsupportedVP8     vp8l.h:87-88              */
  VP8L_ERR_INVALID_VP8L = -5,        /* exception::InvalidVP8L        vp8l.h:107-108,236-237; transform.h:185-186 */
  VP8L_ERR_INVALID_HUFFMAN = -6,     /* exception::InvalidHuffman     huffman_io.h:105,140,184  */
  VP8L_ERR_NO_DEVICE = -7,           /* no gfx950 device / HIP runtime failure at start-up      */
  VP8L_ERR_NO_MEMORY = -8,           /* exception::MemoryAllocationException                    */
  VP8L_ERR_HIP = -9,                 /* a HIP call failed; vp8l_last_error() has the text       */
  VP8L_ERR_BAD_ARGUMENT = -10,
  VP8L_ERR_WORKSPACE = -11           /* internal: decode workspace too small (retried inside)   */
} vp8l_status;

/* Replaces huffman_io::init_array() (huffman_io.h:53-65), which the reference's main() must call
 * before any codec use (webp.cpp:128).  The code-length order table is a compile-time constant
 * here, so this only checks that a device is usable; calling it is optional.
 * Returns the number of usable devices (>= 1) or VP8L_ERR_NO_DEVICE. */
int vp8l_init(void);

const char *vp8l_status_string(int status);
const char *vp8l_last_error(void); /* thread-local text of the last HIP failure */

/* ---- one image, host memory ------------------------------------------------------------
 * vp8l_encode replaces `VP8_LOSSLESS_ENCODER(argb, width, height)` + `get_bit_writer()`
 * (vp8l.h:585-615).  argb: width*height pixels 0xAARRGGBB, borrowed.  *out is allocated by the
 * library (release with vp8l_free) and holds *out_len bytes.  flags = 0 gives the reference encoder's bytes;
 * flags & VP8L_FLAG_PREDICTOR_BITS_MASK = 2..9 adds the predictor transform with that tile size, (flags &
 * VP8L_FLAG_CROSS_COLOR_BITS_MASK) >> 4 = 2..9 the cross-colour transform, see VP8L_OPT_PREDICTOR_BITS and
 * VP8L_OPT_CROSS_COLOR_BITS below. */
#define VP8L_FLAG_PREDICTOR_BITS_MASK 0xFu
#define VP8L_FLAG_CROSS_COLOR_BITS_MASK 0xF0u
#define VP8L_FLAG_CACHE_BITS_MASK 0xF000u   /* (flags & mask) >> 12 = VP8L_OPT_CACHE_BITS */
#define VP8L_FLAG_META_BITS_MASK 0xF0000u   /* (flags & mask) >> 16 = VP8L_OPT_META_BITS (with the default number of groups) */
#define VP8L_FLAG_STANDARD 0x100u /* write the standard VP8L dialect (libwebp reads it) instead of the reference's, see VP8L_OPT_STANDARD */
int vp8l_encode(const uint32_t *argb, uint32_t width, uint32_t height, uint8_t **out, size_t *out_len,
                uint32_t flags);

/* vp8l_decode replaces `VP8_LOSSLESS_DECODER(data, data_length, argb_image)` + image_width()/
 * image_height() (vp8l.h:330-351).  data = the bytes that follow the "VP8L" fourcc in a .webp
 * file (webp.h:61-66).  *argb is allocated by the library (vp8l_free), width*height pixels. */
int vp8l_decode(const uint8_t *data, size_t len, uint32_t **argb, uint32_t *width, uint32_t *height);

/* ReadInfo (vp8l.h:83-94) on the host: dimensions from the first 9 bytes. */
int vp8l_get_info(const uint8_t *data, size_t len, uint32_t *width, uint32_t *height);

/* Releases a result of the host-memory entry points (never free() them: they are page-locked blocks, which the
 * library keeps for later calls, up to VP8L_HOST_POOL_MB megabytes, default 2048; 0 = ordinary malloc'ed memory). */
void vp8l_free(void *p);

/* ---- batches, host memory ----------------------------------------------------------------
 * n independent images on one device (device < 0: current device).  Per-image results in
 * status[i]; the return value is VP8L_OK when the batch ran (even if single images failed). */
int vp8l_encode_batch(int device, size_t n, const uint32_t *const *argb, const uint32_t *width,
                      const uint32_t *height, uint8_t **out, size_t *out_len, int *status);
int vp8l_decode_batch(int device, size_t n, const uint8_t *const *data, const size_t *len, uint32_t **argb,
                      uint32_t *width, uint32_t *height, int *status);

/* The same over several devices of one node: the images are dealt out in contiguous blocks balanced by pixel count, one
 * block per entry of `devices` (an entry may repeat), each block runs as vp8l_encode_batch / vp8l_decode_batch on its device
 * from a host thread of its own.  Images are independent: nothing is exchanged between devices. */
int vp8l_encode_batch_devices(const int *devices, size_t ndevices, size_t n, const uint32_t *const *argb, const uint32_t *width,
                              const uint32_t *height, uint8_t **out, size_t *out_len, int *status);
int vp8l_decode_batch_devices(const int *devices, size_t ndevices, size_t n, const uint8_t *const *data, const size_t *len,
                              uint32_t **argb, uint32_t *width, uint32_t *height, int *status);

/* ---- batches, device memory (pipelines, benchmarks) ---------------------------------------
 * The reference has no counterpart: it is single-image, host-only.  These are the batch forms
 * with inputs and outputs resident in HBM.  A context owns its workspace and one HIP stream;
 * calls on one context are serialised, different contexts are independent. */
typedef struct vp8l_encoder vp8l_encoder;
typedef struct vp8l_decoder vp8l_decoder;

int vp8l_encoder_create(int device, vp8l_encoder **enc);
void vp8l_encoder_destroy(vp8l_encoder *enc);
/* Upper bound of one stream's size in bytes (what to reserve per image in d_out). */
size_t vp8l_encode_bound(uint32_t width, uint32_t height);
/* d_argb[i]: device pointer to width[i]*height[i] pixels (host array of device pointers).
 * d_out: device buffer of out_cap bytes, 256-byte aligned.  Stream i is written at
 * d_out + out_offset[i] (out_offset is an OUTPUT, host array) with out_len[i] bytes. */
int vp8l_encoder_encode_device(vp8l_encoder *enc, size_t n, const uint32_t *const *d_argb, const uint32_t *width,
                               const uint32_t *height, uint8_t *d_out, size_t out_cap, size_t *out_offset,
                               size_t *out_len, int *status);

/* ---- one image, device memory, no allocation ------------------------------------------------
 * No counterpart in the reference.  For callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  The same kernels and the same bytes as
 * vp8l_encoder_encode_device with one image and default options, but the workspace is a static array in device memory
 * (about 20 MB, held by every GPU the library runs on until the process ends) instead of a context's hipMalloc'ed
 * buffers, and the call makes no hipMalloc / hipFree / host allocation.  At most VP8L_NOALLOC_MAX_PIXELS pixels
 * (1920x1080 fits); larger images return VP8L_ERR_NO_MEMORY.
 *   device    < 0 = the current device; >= 0: that device, made current for the length of the call only
 *   stream    a hipStream_t (NULL = the null stream); every kernel of the call is launched on it
 *   d_out     device buffer, 256-byte aligned, out_cap >= vp8l_encode_bound(width, height)
 *   d_result  optional device pointer to two uint64: [0] = status (a vp8l_status, sign-extended), [1] = stream bytes
 *   out_len   optional host pointer.  Non-NULL: the call waits for the stream and returns the image's status with
 *             *out_len set.  NULL: nothing is waited for (the call only launches) and VP8L_OK means "launched".
 * The address of the static array is looked up once per device: by vp8l_init() for the device that is current when it
 * is called, else by the first call on the device.
 * All calls on one GPU share the array: calls that overlap in time on one GPU (two streams, two threads) race. */
#define VP8L_NOALLOC_MAX_PIXELS (1u << 21)
int vp8l_encode_device_noalloc(int device, void *stream, const uint32_t *d_argb, uint32_t width, uint32_t height,
                               uint8_t *d_out, size_t out_cap, uint64_t *d_result, size_t *out_len);

/* Options.  VP8L_OPT_MATCH_SEARCH: 0 (default) = candidate-list match search; tiles with heavy duplication go on to the
 * forms made for them (one frequent value; pairs of pixels) and last to the exhaustive search; 1 = exhaustive search
 * everywhere; 2 = the candidate lists refuse every tile, so the forms behind them see all of them; 3 = as 2, with one more
 * form (sources numbered by the 16, 4, 2 words that follow them) in front of the exhaustive search; 4 = as 0, and the
 * tiles of palette images that are left for the exhaustive search are searched only at the positions a parse can reach
 * (the other positions of such a tile then read "no match" in the match array; no parse looks at them).  All give
 * identical streams; the option exists for verification and measurement. */
enum { VP8L_OPT_MATCH_SEARCH = 1, VP8L_OPT_PREDICTOR_BITS = 2, VP8L_OPT_CROSS_COLOR_BITS = 3, VP8L_OPT_STANDARD = 4,
       VP8L_OPT_CACHE_BITS = 5, VP8L_OPT_META_BITS = 6, VP8L_OPT_META_GROUPS = 7 };
/* VP8L_OPT_PREDICTOR_BITS: 0 (default) = the reference encoder's output (subtract green or colour indexing only,
 * vp8l.h:594-610).  2..9 = additionally apply the predictor transform (14 modes, one per tile of 2^bits pixels
 * a side, chosen per tile; the tile size grows until an image has at most 1024 tiles) in front of the same
 * coder.  The reference DECODER reads such streams (ReadTransform vp8l.h:111-121, transform.h:110-191); they
 * are much smaller for natural images but of course differ from the reference encoder's bytes. */
/* VP8L_OPT_CROSS_COLOR_BITS: 0 (default) = none; 2..9 = apply the cross-colour transform (per tile of 2^bits pixels a
 * side three multipliers that take green out of red and green and red out of blue; the tile size grows like the
 * predictor's) after subtract green and the predictor transform, if any.  Read by the reference decoder
 * (transform.h:192-218); pays on images whose channels are correlated by factors other than one. */
/* VP8L_OPT_CACHE_BITS: 0 (default) = no colour cache, as the reference encoder writes (vp8l.h:513); 1..11 = a colour cache of 2^bits entries
 * (read by the reference decoder: ReadColorCacheBits vp8l.h:230-241, color_cache.h:29-54): a pixel the parse codes by itself is written
 * as its cache index when the decoder's cache holds it at that moment.
 * VP8L_OPT_META_BITS: 0 (default) = one set of five Huffman codes for the image (vp8l.h:514); 2..9 = an entropy image with tiles of 2^bits
 * pixels a side (the tile size grows until the image has at most 1024 tiles) that gives every tile one of VP8L_OPT_META_GROUPS (2..8,
 * default 4) sets of codes (read by ReadSpatiallyCodedImage vp8l.h:191-223, SelectMetaHuffman :250-256); tiles are grouped by how busy
 * their residuals are.  Both go with every other option and both dialects; neither is something the reference encoder produces. */
/* VP8L_OPT_STANDARD: 0 (default) = the reference's private dialect (68-symbol code-length alphabet, 6-bit fields, codes of up to 64
 * bits, a zero chunk size: huffman_io.h:37-50,63-64, vp8l.h:461; libwebp cannot read it).  1 = the VP8L format of the WebP lossless
 * specification: 19-symbol code-length alphabet with 4-bit count and 3-bit lengths, codes limited to 15 bits, the real chunk size in
 * the first four bytes, the alpha-is-used bit set, and (with the predictor option) the specification's top-right rule for the last
 * column (transform.h:137 is the reference's).  Same transforms, same match search and parse.  Wrap the stream as "RIFF", u32 size,
 * "WEBP", "VP8L", stream (+ one zero byte if its length is odd; size = 4 + 4 + stream bytes + that byte) and libwebp decodes it.
 * The decoder needs no option: the reference never writes anything but zero into the first 32 bits and a standard stream's chunk size
 * is never zero, so a stream says itself which dialect it is. */
int vp8l_encoder_set_option(vp8l_encoder *enc, int option, int value);
/* Number of 2048-position tiles of the last call that needed the exhaustive search (and the total). */
long vp8l_encoder_last_dense_tiles(vp8l_encoder *enc, long *total_tiles);
/* How the tiles of the last call were searched: counts8[k] = tiles with flag k (0 candidate lists, 1 exhaustive search,
 * 2 lists on pairs of pixels, 4 lists around one frequent value; every form gives the same matches).  Returns the total.
 * No counterpart in the reference (lz77.h:28-93 has one form): a measurement aid. */
long vp8l_encoder_last_tile_kinds(vp8l_encoder *enc, long *counts8);

/* Stage outputs of the LAST encode call on this context, for parity checks of the intermediate
 * planes (copied to host memory).  what: */
enum {
  VP8L_STAGE_RESIDUAL = 0, /* transformed (subtract-green / indexed+bundled) pixels, uint32 x M        */
  VP8L_STAGE_MATCH = 1,    /* per position: length | distance<<8 | parse_flag<<31, uint32 x M          */
  VP8L_STAGE_HISTO = 2,    /* 5 x 280 uint32 (GREEN, RED, BLUE, ALPHA, DIST; unused tail zero)           */
  VP8L_STAGE_CODELEN = 3,  /* 5 x 280 uint8 code lengths                                                 */
  VP8L_STAGE_INFO = 4,     /* uint32[4]: ncolors (0 = subtract-green path), bundling bits, xsize, M     */
  VP8L_STAGE_PRED_MODES = 5, /* uint32[3] tile bits, columns, rows, then one uint32 per tile (0 bytes: none) */
  VP8L_STAGE_CROSS_ELEMS = 6, /* the same layout for the cross-colour elements                                */
  VP8L_STAGE_META_GROUPS = 7  /* the same layout: the Huffman group of every tile of the entropy image (0 bytes: none) */
};
long vp8l_encoder_fetch_stage(vp8l_encoder *enc, size_t image, int what, void *host_buf, size_t cap_bytes);

int vp8l_decoder_create(int device, vp8l_decoder **dec);
void vp8l_decoder_destroy(vp8l_decoder *dec);
/* d_data[i]: device pointer to stream i (len[i] bytes; 4-byte aligned).  width/height: from
 * vp8l_get_info.  d_argb[i]: device pointer receiving width[i]*height[i] pixels. */
int vp8l_decoder_decode_device(vp8l_decoder *dec, size_t n, const uint8_t *const *d_data, const size_t *len,
                               const uint32_t *width, const uint32_t *height, uint32_t *const *d_argb, int *status);
/* VP8L_DEC_OPT_THREADS_PER_STREAM: how many threads decode one stream's entropy-coded pixels.  0 or 1024
 * (default) = one workgroup per stream; 64 = one wave per stream (seven times as many streams in flight, each
 * several times slower: only faster for thousands of small streams of noisy content).  Identical pixels. */
/* VP8L_DEC_OPT_WHOLE_DEVICE: streams are cut into segments of 65536 bits that many workgroups decode (a scout per
 * segment records its tokens, a serial check accepts them, pointer jumping over the pixels resolves the references; a
 * 3840x2160 stream in 3 ms instead of half a second, 29 GP/s instead of 4.2 for a large batch).  0 (default) = when a
 * call has at least 2 MiB of streams, or at least 48 KiB of them in streams of 32 KiB and more on average (one 512x512
 * photo-like stream: 1.5 instead of 3.5 ms); 1 = never; 2 = always.  Images below 65536 pixels, streams with a colour cache,
 * several Huffman groups or tokens the segment decoder does not take are decoded by one workgroup each as before.
 * Workspace: 16 bytes per stream byte and 4 bytes per pixel, in passes of at most 2 GiB of streams.  Identical pixels
 * and verdicts. */
/* VP8L_DEC_OPT_SCOUT_FORM: how a segment's scout finds its tokens.  0 (default) = the segment is cut into pieces of 8192
 * bits, a lane decodes a piece token after token, first from a guessed entry and again from where the lane before it
 * left, until no entry changes; 1 = a token is decoded at every bit offset and the chain is found by pointer jumping
 * (the first version: three to five times slower for batches).  Identical
 * pixels and verdicts. */
enum { VP8L_DEC_OPT_THREADS_PER_STREAM = 1, VP8L_DEC_OPT_WHOLE_DEVICE = 2, VP8L_DEC_OPT_SCOUT_FORM = 3 };
int vp8l_decoder_set_option(vp8l_decoder *dec, int option, int value);
/* Streams of the last call on the context that the segment kernels decoded. */
long vp8l_decoder_last_whole_device_streams(const vp8l_decoder *dec);
/* ... and of those, the streams whose scouts were accepted (the others were decoded again with the entry maps). */
long vp8l_decoder_last_scout_streams(const vp8l_decoder *dec);

/* Seconds spent in the device part of the last call on the context (HIP events on the
 * context's stream), and per-kernel-stage seconds for the encoder
 * (stage order: palette, transform, match, parse, histogram, header, bitcount+scan, emit; a ninth
 * value is the part of "match" spent in the exhaustive fallback kernel). */
double vp8l_encoder_last_device_seconds(const vp8l_encoder *enc);
int vp8l_encoder_last_stage_seconds(const vp8l_encoder *enc, double *seconds, int cap);
double vp8l_decoder_last_device_seconds(const vp8l_decoder *dec);

/* Synthetic benchmark/test images generated on the device (seeded LCG, same definitions as the
 * survey's known-answer images: mode 3 photo-like, 4 graphics, 5 palette with ncolors). */
int vp8l_generate_device(int device, int mode, uint32_t width, uint32_t height, uint32_t seed, uint32_t ncolors,
                         uint32_t *d_argb);

#ifdef __cplusplus
}
#endif
#endif /* VP8L_AMD_H_ */

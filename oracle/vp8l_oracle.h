/*
 * vp8l_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded restatement of the reference codec's VP8L hot path
 * (RuslanKutdusov/WebP: webp/vp8l, webp/lz77, webp/huffman_coding, webp/utils bit I/O).
 * It exists to CHECK the HIP implementation; nothing in the shipped library may
 * include, link or call it.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it.
 *
 * Parity status: pinned.  Every component below that the reference can build
 * without <png.h> (LZ77 parse, prefix/distance coding, Huffman lengths/codes,
 * code-length RLE + header bits, Huffman header reader, inverse transforms,
 * colour cache, bundling) is compared against the compiled reference sources
 * (oracle/_ref, built from /root/reference in place), and the whole-stream
 * encoder is pinned by the size + sha256 known-answer vectors recorded from the
 * compiled reference for seven synthetic images (tests/golden/kat_streams.json).
 * The reference driver header webp/vp8l/vp8l.h itself cannot be compiled in this
 * image because it includes <png.h>, which is absent.
 */
#ifndef VP8L_ORACLE_H_
#define VP8L_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes: one per reference exception type (webp/exception/exception.h:14-153). */
enum {
  ORA_OK = 0,
  ORA_INVALID_ARGB = -1,        /* InvalidARGBImage   vp8l.h:587-588 */
  ORA_TOO_BIG_ARGB = -2,        /* TooBigARGBImage    vp8l.h:589-590 */
  ORA_TOO_BIG_CODE_LENGTH = -3, /* TooBigCodeLength   huffman_coding.h:410-413 */
  ORA_UNSUPPORTED_VP8 = -4,     /* UnsupportedVP8     vp8l.h:87-88 */
  ORA_INVALID_VP8L = -5,        /* InvalidVP8L        vp8l.h:107-108,236-237; transform.h:185-186 */
  ORA_INVALID_HUFFMAN = -6,     /* InvalidHuffman     huffman_io.h:105,140,184; huffman_coding.h:162 */
  ORA_NOMEM = -8
};

typedef struct {
  uint32_t distance; /* 0 for a literal */
  uint32_t length;   /* 0 for a literal */
  uint32_t symbol;   /* the pixel for a literal, 0 for a match */
} ora_token;

/* Intermediate results of one encode, for stage-by-stage parity checks. */
typedef struct {
  int use_palette;        /* 1 = colour-indexing path, 0 = subtract-green path */
  uint32_t ncolors;       /* palette size (0 when !use_palette) */
  uint32_t palette[256];  /* sorted ascending, before delta coding */
  uint32_t bits;          /* pixel-bundling bits 0..3 */
  uint32_t xsize;         /* width of the coded (possibly bundled) image */
  uint32_t ysize;
  uint32_t *residual;     /* xsize*ysize transformed pixels (malloc'd) */
  ora_token *tokens;      /* greedy parse of residual (malloc'd) */
  size_t ntokens;
  uint32_t histo[5][280]; /* GREEN(280) RED BLUE ALPHA(256) DIST(40) */
  uint8_t lengths[5][280];
  uint32_t codes[5][280]; /* bit-reversed canonical codes as written */
} ora_encode_trace;

void ora_trace_free(ora_encode_trace *t);

/* Synthetic image generators (mode 3 photo-like, 4 graphics, 5 palette-`ncolors`). */
void ora_gen_image(int mode, uint32_t w, uint32_t h, uint32_t seed, uint32_t ncolors,
                   uint32_t *argb);

/* vp8l.h:585-612.  *out is malloc'd; free with ora_free.  trace may be NULL. */
int ora_encode(const uint32_t *argb, uint32_t w, uint32_t h, uint8_t **out, size_t *out_len,
               ora_encode_trace *trace);
/* vp8l.h:330-345.  *argb is malloc'd (w*h). */
int ora_decode(const uint8_t *data, size_t len, uint32_t **argb, uint32_t *w, uint32_t *h);
int ora_get_info(const uint8_t *data, size_t len, uint32_t *w, uint32_t *h);
void ora_free(void *p);

/* lz77.h:28-93 */
int ora_lz77(const uint32_t *data, uint32_t n, ora_token **tokens, size_t *ntokens);
/* Per-position best (length, distance) as the parse at that position would see it. */
void ora_lz77_best(const uint32_t *data, uint32_t n, uint32_t p, uint32_t *len, uint32_t *dist);
/* lz77.cpp:89-102 */
void ora_prefix_encode(uint32_t value, uint32_t *symbol, uint32_t *nbits, uint32_t *bits);
/* lz77.cpp:104-112 */
uint32_t ora_dist_to_code(uint64_t xsize, uint64_t dist);
/* lz77.cpp:59-70 */
uint32_t ora_code_to_dist(uint32_t xsize, uint32_t code);
/* huffman_coding.h:366-419.  Returns number of used symbols or a negative status. */
int ora_huffman_build(const uint32_t *histo, int n, int max_len, uint8_t *lengths,
                      uint32_t *codes);
/* huffman_io.h:372-410: serialise one code description into bits.  Returns the number
 * of bits written into buf (cap bytes, zero-filled by the callee) or a negative status. */
long ora_write_code(const uint8_t *lengths, int n, uint8_t *buf, size_t cap);
/* huffman_io.h:147-195: parse one code description; lengths[n] out; *bitpos advanced. */
int ora_read_code(const uint8_t *data, size_t len, uint64_t *bitpos, int alphabet,
                  uint8_t *lengths, int *is_simple, uint16_t simple_syms[2], int *simple_n);

/* transform.h:59-218: one inverse transform applied in place to a w*h image.
 * type 0 predictor, 1 cross-colour, 2 subtract-green, 3 colour-indexing.
 * tdata = tile data / palette, txsize = tile columns, tbits = tile bits / bundling bits. */
int ora_inverse_transform(int type, uint32_t *argb, uint32_t w, uint32_t h,
                          const uint32_t *tdata, uint32_t txsize, uint32_t tbits);
/* vp8l.h:411-416 (pixel part only) */
void ora_subtract_green(uint32_t *argb, size_t n);
/* transform.cpp:88-106 */
void ora_bundle(const uint8_t *row, int width, int xbits, uint32_t *dst);

/*
 * Test-vector encoder: emits the decoder-only features of the dialect (predictor and
 * cross-colour transforms with caller-supplied tile data, colour cache, meta-Huffman
 * entropy image).  The reference ENCODER never produces these (vp8l.h:594-610); the
 * reference DECODER accepts them (vp8l.h:101-328).  Used to make decode inputs.
 */
typedef struct {
  int use_subtract_green;
  int predictor_bits;      /* 0 = none, else 2..9 */
  const uint32_t *predictor_modes; /* tile image (mode in green byte), or NULL = mode chosen per tile by index */
  int color_bits;          /* 0 = none, else 2..9 */
  const uint32_t *color_elems;     /* tile image, or NULL = synthetic */
  int use_palette;         /* colour indexing if image has <=256 colours */
  int cache_bits;          /* 0..11 */
  int meta_bits;           /* 0 = single group, else 2..9 */
  int meta_groups;         /* number of Huffman groups when meta_bits != 0 */
  int transform_order;     /* permutation selector for the order transforms are written */
  const uint32_t *meta_image;      /* Huffman group of every tile of the entropy image, or NULL = synthetic pattern */
  int stray_cache;         /* n > 0: every n-th literal of the main image becomes a cache token with an unrelated index
                            * (decoder test vectors: such a stream does not decode to the input) */
} ora_ext_params;

int ora_encode_ext(const uint32_t *argb, uint32_t w, uint32_t h, const ora_ext_params *prm,
                   uint8_t **out, size_t *out_len);

#ifdef __cplusplus
}
#endif
#endif

/*
 * vp8l_oracle.c -- TEST INFRASTRUCTURE ONLY (see vp8l_oracle.h).
 *
 * CPU restatement of the reference VP8L path.  Every function cites the reference
 * file:line it follows (paths relative to the reference checkout).  The code is
 * deliberately scalar and close to the reference's control flow (brute-force LZ77,
 * bit-at-a-time reader semantics) because it doubles as the host-CPU baseline.
 */
#include "vp8l_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* dialect constants: webp/vp8l/huffman_io.h:26-50, webp/vp8l/vp8l.h:23-26    */
/* ------------------------------------------------------------------------- */
#define MAX_CODE_LEN 64          /* MAX_ALLOWED_CODE_LENGTH */
#define MAX_RLE_TREE_LEN 32      /* MAX_ALLOWED_CODE_LENGTH_OF_RLE_TREE */
#define SYM_REPEAT 65            /* NON_ZERO_REPS_CODE */
#define SYM_ZERO11 66            /* ZERO_11_REPS_CODE */
#define SYM_ZERO138 67           /* ZERO_138_REPS_CODE */
#define NUM_CL_SYMS 68           /* RLE_CODES_COUNT */
#define CL_COUNT_BITS 6          /* (int)log2f(68) */
#define CL_LEN_BITS 6            /* (int)log2f(32)+1 */
#define LZ_WINDOW 1024
#define LZ_MAXLEN 128
#define MAX_DIM 16384
#define MAX_CACHE_BITS 11
/* Codes longer than 32 bits overflow the reference's 32-bit code_t
 * (huffman_coding.h:12) and produce undefined output there; the oracle reports them. */
#define MAX_SANE_CODE_LEN 32

static const int kAlphabet[5] = {280, 256, 256, 256, 40};

/* huffman_io.h:49-65 after init_array() */
static int cl_order(int i) {
  static const int head[9] = {SYM_ZERO11, SYM_ZERO138, 0, 1, 2, 3, 4, 5, SYM_REPEAT};
  return i < 9 ? head[i] : i - 3;
}

void ora_free(void *p) { free(p); }

/* ------------------------------------------------------------------------- */
/* bit writer: webp/utils/bit_writer.h:45-64 (LSB first, zero padded)          */
/* ------------------------------------------------------------------------- */
typedef struct {
  uint8_t *buf;
  size_t cap;
  uint64_t nbits;
  int oom;
} bitw;

static void bw_init(bitw *w) { memset(w, 0, sizeof(*w)); }

static void bw_put1(bitw *w, uint32_t bit) {
  size_t byte = (size_t)(w->nbits >> 3);
  if (byte >= w->cap) {
    size_t ncap = w->cap ? w->cap * 2 : 4096;
    uint8_t *nb = (uint8_t *)realloc(w->buf, ncap);
    if (!nb) { w->oom = 1; return; }
    memset(nb + w->cap, 0, ncap - w->cap);
    w->buf = nb;
    w->cap = ncap;
  }
  w->buf[byte] |= (uint8_t)((bit & 1u) << (w->nbits & 7));
  w->nbits++;
}

static void bw_put(bitw *w, uint32_t v, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) bw_put1(w, (v >> i) & 1u);
}

static size_t bw_bytes(const bitw *w) { return (size_t)((w->nbits + 7) >> 3); }

/* ------------------------------------------------------------------------- */
/* bit reader: webp/utils/bit_readed.h:27-63.  Bytes past the end read as 0    */
/* (the reference reads one byte out of bounds, then returns zeros).           */
/* ------------------------------------------------------------------------- */
typedef struct {
  const uint8_t *data;
  size_t len;
  uint64_t pos;
} bitr;

static uint32_t br_get1(bitr *r) {
  size_t byte = (size_t)(r->pos >> 3);
  uint32_t b = byte < r->len ? (r->data[byte] >> (r->pos & 7)) & 1u : 0u;
  r->pos++;
  return b;
}

static uint32_t br_get(bitr *r, uint32_t n) {
  uint32_t v = 0;
  for (uint32_t i = 0; i < n; i++) v |= br_get1(r) << i;
  return v;
}

/* ------------------------------------------------------------------------- */
/* pixel helpers: webp/vp8l/transform.cpp:8-106, webp/utils/utils.cpp:55-73    */
/* ------------------------------------------------------------------------- */
#define CH_A(p) (((p) >> 24) & 0xffu)
#define CH_R(p) (((p) >> 16) & 0xffu)
#define CH_G(p) (((p) >> 8) & 0xffu)
#define CH_B(p) ((p) & 0xffu)

static uint32_t px_add(uint32_t a, uint32_t b) { /* PixelsSum transform.cpp:8-13 */
  uint32_t ag = (a & 0xff00ff00u) + (b & 0xff00ff00u);
  uint32_t rb = (a & 0x00ff00ffu) + (b & 0x00ff00ffu);
  return (ag & 0xff00ff00u) | (rb & 0x00ff00ffu);
}

static uint32_t px_sub(uint32_t a, uint32_t b) { /* PixelsSub transform.cpp:15-19 */
  uint32_t ag = 0x00ff00ffu + (a & 0xff00ff00u) - (b & 0xff00ff00u);
  uint32_t rb = 0xff00ff00u + (a & 0x00ff00ffu) - (b & 0x00ff00ffu);
  return (ag & 0xff00ff00u) | (rb & 0x00ff00ffu);
}

static uint32_t px_avg2(uint32_t a, uint32_t b) { /* Average2 transform.cpp:21-30 */
  uint32_t r = 0;
  for (int s = 0; s < 32; s += 8) r |= ((((a >> s) & 0xff) + ((b >> s) & 0xff)) / 2) << s;
  return r;
}

static int iabs(int v) { return v < 0 ? -v : v; }

/* Select as CALLED at transform.h:177: Select(T, L, TL) binds the callee's first
 * parameter ("L", transform.cpp:33) to the top pixel and its second to the left one. */
static uint32_t px_select(uint32_t T, uint32_t L, uint32_t TL) {
  int pa = (int)CH_A(T) + (int)CH_A(L) - (int)CH_A(TL);
  int pr = (int)CH_R(T) + (int)CH_R(L) - (int)CH_R(TL);
  int pg = (int)CH_G(T) + (int)CH_G(L) - (int)CH_G(TL);
  int pb = (int)CH_B(T) + (int)CH_B(L) - (int)CH_B(TL);
  int d_first = iabs(pa - (int)CH_A(T)) + iabs(pr - (int)CH_R(T)) + iabs(pg - (int)CH_G(T)) +
                iabs(pb - (int)CH_B(T));
  int d_second = iabs(pa - (int)CH_A(L)) + iabs(pr - (int)CH_R(L)) + iabs(pg - (int)CH_G(L)) +
                 iabs(pb - (int)CH_B(L));
  return d_first <= d_second ? T : L;
}

static uint32_t clamp255(int v) { return v < 0 ? 0u : v > 255 ? 255u : (uint32_t)v; }

static uint32_t px_clamp_full(uint32_t a, uint32_t b, uint32_t c) { /* transform.cpp:61-73 */
  uint32_t r = 0;
  for (int s = 0; s < 32; s += 8)
    r |= clamp255((int)((a >> s) & 0xff) + (int)((b >> s) & 0xff) - (int)((c >> s) & 0xff)) << s;
  return r;
}

static uint32_t px_clamp_half(uint32_t a, uint32_t b) { /* transform.cpp:75-86 */
  uint32_t r = 0;
  for (int s = 0; s < 32; s += 8) {
    int av = (int)((a >> s) & 0xff), bv = (int)((b >> s) & 0xff);
    r |= clamp255(av + (av - bv) / 2) << s; /* C division truncates toward zero */
  }
  return r;
}

void ora_bundle(const uint8_t *row, int width, int xbits, uint32_t *dst) { /* transform.cpp:88-106 */
  if (xbits > 0) {
    int depth = 1 << (3 - xbits), mask = (1 << xbits) - 1;
    uint32_t code = 0xff000000u;
    for (int x = 0; x < width; x++) {
      int sub = x & mask;
      if (sub == 0) code = 0xff000000u;
      code |= (uint32_t)row[x] << (8 + depth * sub);
      dst[x >> xbits] = code;
    }
  } else {
    for (int x = 0; x < width; x++) dst[x] = 0xff000000u | ((uint32_t)row[x] << 8);
  }
}

void ora_subtract_green(uint32_t *argb, size_t n) { /* vp8l.h:413-416 */
  for (size_t i = 0; i < n; i++) {
    uint32_t p = argb[i], g = CH_G(p);
    uint32_t r = (CH_R(p) - g) & 0xff, b = (CH_B(p) - g) & 0xff;
    argb[i] = (p & 0xff00ff00u) | (r << 16) | b;
  }
}

/* ------------------------------------------------------------------------- */
/* LZ77: webp/lz77/lz77.h:28-93                                                */
/* ------------------------------------------------------------------------- */
void ora_lz77_best(const uint32_t *data, uint32_t n, uint32_t p, uint32_t *len, uint32_t *dist) {
  uint32_t rem = n - p < LZ_MAXLEN ? n - p : LZ_MAXLEN;
  uint32_t best = 0, bestd = 0;
  *len = 0;
  *dist = 0;
  if (rem <= 1) return; /* lz77.h:42-49: a single look-ahead pixel is always a literal */
  uint32_t start = p > LZ_WINDOW ? p - LZ_WINDOW : 0;
  for (int64_t j = (int64_t)p - 1; j >= (int64_t)start; j--) { /* lz77.h:56 nearest first */
    uint32_t d = p - (uint32_t)j, l = 0;
    /* source stops at p (sb_stop lz77.h:54), look-ahead stops at p+rem (la_stop :55) */
    while (l < d && l < rem && data[j + l] == data[p + l]) l++;
    if (l && l >= best) { /* lz77.h:67-69: ">=" keeps the farthest among equals */
      best = l;
      bestd = d;
    }
  }
  *len = best;
  *dist = bestd;
}

int ora_lz77(const uint32_t *data, uint32_t n, ora_token **tokens, size_t *ntokens) {
  ora_token *t = (ora_token *)malloc(sizeof(ora_token) * (n ? n : 1));
  size_t k = 0;
  if (!t) return ORA_NOMEM;
  for (uint32_t p = 0; p < n;) {
    uint32_t l, d;
    ora_lz77_best(data, n, p, &l, &d);
    if (l) {
      t[k].distance = d; t[k].length = l; t[k].symbol = 0;
      p += l;
    } else {
      t[k].distance = 0; t[k].length = 0; t[k].symbol = data[p];
      p += 1;
    }
    k++;
  }
  *tokens = t;
  *ntokens = k;
  return ORA_OK;
}

/* lz77.cpp:72-102 */
void ora_prefix_encode(uint32_t value, uint32_t *symbol, uint32_t *nbits, uint32_t *bits) {
  uint32_t v = value - 1;
  int hb = -1;
  for (int i = 31; i >= 0; i--) if ((v >> i) & 1u) { hb = i; break; }
  if (hb > 0) {
    uint32_t second = (v >> (hb - 1)) & 1u;
    *nbits = (uint32_t)(hb - 1);
    *bits = v & ((1u << (hb - 1)) - 1u);
    *symbol = (uint32_t)(2 * hb) + second;
  } else {
    *nbits = 0;
    *bits = 0;
    *symbol = hb == 0 ? 1u : 0u;
  }
}

/* lz77.cpp:38-47 */
static const uint8_t kPlaneLut[128] = {
    96,  73,  55,  39,  23,  13,  5,   1,   255, 255, 255, 255, 255, 255, 255, 255,
    101, 78,  58,  42,  26,  16,  8,   2,   0,   3,   9,   17,  27,  43,  59,  79,
    102, 86,  62,  46,  32,  20,  10,  6,   4,   7,   11,  21,  33,  47,  63,  87,
    105, 90,  70,  52,  37,  28,  18,  14,  12,  15,  19,  29,  38,  53,  71,  91,
    110, 99,  82,  66,  48,  35,  30,  24,  22,  25,  31,  36,  49,  67,  83,  100,
    115, 108, 94,  76,  64,  50,  44,  40,  34,  41,  45,  51,  65,  77,  95,  109,
    118, 113, 103, 92,  80,  68,  60,  56,  54,  57,  61,  69,  81,  93,  104, 114,
    119, 116, 111, 106, 97,  88,  84,  74,  72,  75,  85,  89,  98,  107, 112, 117};

/* lz77.cpp:104-112 -- all arithmetic in 64-bit unsigned like the reference's size_t,
 * so "xsize - 8" wraps for xsize < 8 and the second branch is then dead. */
uint32_t ora_dist_to_code(uint64_t xsize, uint64_t dist) {
  uint64_t y = dist / xsize, x = dist - y * xsize;
  if (x <= 8 && y < 8) return (uint32_t)kPlaneLut[y * 16 + 8 - x] + 1;
  if (x > xsize - 8 && y < 7) return (uint32_t)kPlaneLut[(y + 1) * 16 + 8 + (xsize - x)] + 1;
  return (uint32_t)(dist + 120);
}

/* lz77.cpp:20-35: (dx, dy) per plane code 1..120 */
static const int8_t kPlaneXY[120][2] = {
    {0, 1},  {1, 0},  {1, 1},  {-1, 1}, {0, 2},  {2, 0},  {1, 2},  {-1, 2}, {2, 1},  {-2, 1},
    {2, 2},  {-2, 2}, {0, 3},  {3, 0},  {1, 3},  {-1, 3}, {3, 1},  {-3, 1}, {2, 3},  {-2, 3},
    {3, 2},  {-3, 2}, {0, 4},  {4, 0},  {1, 4},  {-1, 4}, {4, 1},  {-4, 1}, {3, 3},  {-3, 3},
    {2, 4},  {-2, 4}, {4, 2},  {-4, 2}, {0, 5},  {3, 4},  {-3, 4}, {4, 3},  {-4, 3}, {5, 0},
    {1, 5},  {-1, 5}, {5, 1},  {-5, 1}, {2, 5},  {-2, 5}, {5, 2},  {-5, 2}, {4, 4},  {-4, 4},
    {3, 5},  {-3, 5}, {5, 3},  {-5, 3}, {0, 6},  {6, 0},  {1, 6},  {-1, 6}, {6, 1},  {-6, 1},
    {2, 6},  {-2, 6}, {6, 2},  {-6, 2}, {4, 5},  {-4, 5}, {5, 4},  {-5, 4}, {3, 6},  {-3, 6},
    {6, 3},  {-6, 3}, {0, 7},  {7, 0},  {1, 7},  {-1, 7}, {5, 5},  {-5, 5}, {7, 1},  {-7, 1},
    {4, 6},  {-4, 6}, {6, 4},  {-6, 4}, {2, 7},  {-2, 7}, {7, 2},  {-7, 2}, {3, 7},  {-3, 7},
    {7, 3},  {-7, 3}, {5, 6},  {-5, 6}, {6, 5},  {-6, 5}, {8, 0},  {4, 7},  {-4, 7}, {7, 4},
    {-7, 4}, {8, 1},  {8, 2},  {6, 6},  {-6, 6}, {8, 3},  {5, 7},  {-5, 7}, {7, 5},  {-7, 5},
    {8, 4},  {6, 7},  {-6, 7}, {7, 6},  {-7, 6}, {8, 5},  {7, 7},  {-7, 7}, {8, 6},  {8, 7}};

/* lz77.cpp:59-70 -- uint32 arithmetic as in the reference (a negative dx with a small
 * dy*xsize wraps to a huge distance instead of clamping to 1). */
uint32_t ora_code_to_dist(uint32_t xsize, uint32_t code) {
  if (code <= 120) {
    uint32_t d = (uint32_t)(int32_t)kPlaneXY[code - 1][0] + (uint32_t)kPlaneXY[code - 1][1] * xsize;
    return d >= 1 ? d : 1;
  }
  return code - 120;
}

static uint32_t prefix_decode(uint32_t sym, bitr *r) { /* lz77.cpp:50-57 */
  if (sym < 4) return sym + 1;
  uint32_t eb = (sym - 2) >> 1;
  uint32_t off = (2 + (sym & 1)) << eb;
  return off + br_get(r, eb) + 1;
}

/* ------------------------------------------------------------------------- */
/* Huffman encoder side: webp/huffman_coding/huffman_coding.h:312-444,         */
/* huffman_coding.cpp:8-49                                                     */
/* ------------------------------------------------------------------------- */
typedef struct {
  uint64_t w;
  int left, right; /* node indices, -1 for a leaf */
  int sym;
} hnode;

static void assign_depth(const hnode *nodes, int idx, int depth, uint8_t *lengths) {
  /* make_codes huffman_coding.h:340-350 */
  if (nodes[idx].left < 0) { lengths[nodes[idx].sym] = (uint8_t)depth; return; }
  assign_depth(nodes, nodes[idx].left, depth + 1, lengths);
  assign_depth(nodes, nodes[idx].right, depth + 1, lengths);
}

static void canonical_codes(const uint8_t *lengths, int n, uint32_t *codes) {
  /* make_canonical_codes huffman_coding.cpp:8-49 */
  uint32_t count[MAX_CODE_LEN + 1] = {0}, next[MAX_CODE_LEN + 1] = {0};
  int maxl = 0;
  for (int s = 0; s < n; s++) { if (lengths[s] > maxl) maxl = lengths[s]; count[lengths[s]]++; }
  count[0] = 0;
  uint32_t cur = 0;
  for (int l = 1; l <= maxl; l++) { cur = (cur + count[l - 1]) << 1; next[l] = cur; }
  for (int s = 0; s < n; s++) codes[s] = lengths[s] ? next[lengths[s]]++ : 0xffffffffu;
}

static uint32_t reverse_bits(uint32_t code, int len) { /* huffman_coding.h:356-363 */
  uint32_t r = 0;
  for (int i = 0; i < len; i++) r |= ((code >> i) & 1u) << (len - 1 - i);
  return r;
}

int ora_huffman_build(const uint32_t *histo, int n, int max_len, uint8_t *lengths,
                      uint32_t *codes) {
  hnode *nodes = (hnode *)malloc(sizeof(hnode) * 2 * (size_t)(n > 0 ? n : 1));
  int *order = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)); /* roots sorted ascending by weight */
  int nn = 0, nroots = 0;
  if (!nodes || !order) { free(nodes); free(order); return ORA_NOMEM; }
  if (n > 0) {
    memset(lengths, 0, (size_t)n);
    memset(codes, 0, sizeof(uint32_t) * (size_t)n);
  }
  for (int s = 0; s < n; s++) {
    if (!histo[s]) continue;
    nodes[nn].w = histo[s]; nodes[nn].left = nodes[nn].right = -1; nodes[nn].sym = s;
    order[nroots++] = nn++;
  }
  int used = nroots;
  if (!used) { free(nodes); free(order); return 0; } /* huffman_coding.h:380-381 */
  /* huffman_coding.h:382: qsort (glibc: stable merge sort) ascending by weight; leaves were
   * appended in symbol order, so a stable insertion sort gives the same arrangement. */
  for (int i = 1; i < nroots; i++) {
    int v = order[i], j = i;
    while (j > 0 && nodes[order[j - 1]].w > nodes[v].w) { order[j] = order[j - 1]; j--; }
    order[j] = v;
  }
  /* huffman_coding.h:384-396: merge the two smallest; the merged node is put in slot 0 and the
   * array re-sorted stably, i.e. it lands BEFORE every existing root of equal weight. */
  while (nroots > 1) {
    int a = order[0], b = order[1];
    nodes[nn].w = nodes[a].w + nodes[b].w; nodes[nn].left = a; nodes[nn].right = b; nodes[nn].sym = -1;
    int pos = 2;
    while (pos < nroots && nodes[order[pos]].w < nodes[nn].w) pos++;
    /* order[2..pos) slide down by one, new node goes to pos-1... after removing two heads */
    for (int i = 2; i < pos; i++) order[i - 2] = order[i];
    order[pos - 2] = nn;
    for (int i = pos; i < nroots; i++) order[i - 1] = order[i];
    nroots--;
    nn++;
  }
  if (nodes[order[0]].left < 0) lengths[nodes[order[0]].sym] = 1; /* single symbol :398-402 */
  else assign_depth(nodes, order[0], 0, lengths);
  free(nodes);
  free(order);
  int maxl = 0;
  for (int s = 0; s < n; s++) if (lengths[s] > maxl) maxl = lengths[s];
  if (maxl > max_len || maxl > MAX_SANE_CODE_LEN) return ORA_TOO_BIG_CODE_LENGTH; /* :410-413 */
  canonical_codes(lengths, n, codes);
  for (int s = 0; s < n; s++) codes[s] = lengths[s] ? reverse_bits(codes[s], lengths[s]) : 0;
  return used;
}

/* code-length RLE: webp/vp8l/huffman_io.h:226-326 */
typedef struct { uint16_t sym; uint8_t extra; } rle_item;

static int rle_lengths(const uint8_t *lengths, int n, rle_item *out) {
  int k = 0;
  int prev = 8;
  for (int i = 0; i < n;) {
    int v = lengths[i], j = i + 1;
    while (j < n && lengths[j] == v) j++;
    int reps = j - i;
    if (v == 0) { /* zero_reps :238-262 */
      while (reps >= 1) {
        if (reps < 3) { for (int t = 0; t < reps; t++) { out[k].sym = 0; out[k++].extra = 0; } break; }
        else if (reps < 11) { out[k].sym = SYM_ZERO11; out[k++].extra = (uint8_t)(reps - 3); break; }
        else if (reps < 139) { out[k].sym = SYM_ZERO138; out[k++].extra = (uint8_t)(reps - 11); break; }
        else { out[k].sym = SYM_ZERO138; out[k++].extra = 138 - 11; reps -= 138; }
      }
    } else { /* code_length_reps :263-284 */
      if (prev != v) { out[k].sym = (uint16_t)v; out[k++].extra = 0; reps--; }
      while (reps >= 1) {
        if (reps < 3) { for (int t = 0; t < reps; t++) { out[k].sym = (uint16_t)v; out[k++].extra = 0; } break; }
        else if (reps < 7) { out[k].sym = SYM_REPEAT; out[k++].extra = (uint8_t)(reps - 3); break; }
        else { out[k].sym = SYM_REPEAT; out[k++].extra = 3; reps -= 6; }
      }
      prev = v;
    }
    i = j;
  }
  return k;
}

/* huffman_io.h:331-410 */
static int write_code(bitw *w, const uint8_t *lengths, int n) {
  int count = 0;
  uint32_t syms[2] = {0, 0};
  for (int i = 0; i < n; i++) {
    if (lengths[i]) { if (count < 2) syms[count] = (uint32_t)i; count++; }
    if (count > 2) break;
  }
  if (count == 0) { bw_put(w, 1, 1); bw_put(w, 0, 1); bw_put(w, 0, 1); bw_put(w, 0, 1); return ORA_OK; }
  if (count <= 2 && syms[0] < 256 && syms[1] < 256) {
    bw_put(w, 1, 1);
    bw_put(w, (uint32_t)(count - 1), 1);
    if (syms[0] <= 1) { bw_put(w, 0, 1); bw_put(w, syms[0], 1); }
    else { bw_put(w, 1, 1); bw_put(w, syms[0], 8); }
    if (count == 2) bw_put(w, syms[1], 8);
    return ORA_OK;
  }
  bw_put(w, 0, 1);
  rle_item items[280 + 2048 + 8];
  int nitems = rle_lengths(lengths, n, items);
  uint32_t histo[NUM_CL_SYMS] = {0};
  for (int i = 0; i < nitems; i++) histo[items[i].sym]++;
  uint8_t cl_len[NUM_CL_SYMS];
  uint32_t cl_code[NUM_CL_SYMS];
  int rc = ora_huffman_build(histo, NUM_CL_SYMS, MAX_RLE_TREE_LEN, cl_len, cl_code);
  if (rc < 0) return rc;
  int nwrite = NUM_CL_SYMS;
  for (; nwrite > 4; nwrite--) if (cl_len[cl_order(nwrite - 1)] != 0) break;
  bw_put(w, (uint32_t)(nwrite - 4), CL_COUNT_BITS);
  for (int i = 0; i < nwrite; i++) bw_put(w, cl_len[cl_order(i)], CL_LEN_BITS);
  bw_put(w, 0, 1); /* "use max_symbol" = 0 (huffman_io.h:358) */
  for (int i = 0; i < nitems; i++) {
    int s = items[i].sym;
    /* A single-symbol code-length tree still has length 1 here, so one bit per item is
     * written although the reference reader consumes none (huffman_io.h:363 vs
     * huffman_coding.h:207-214): reproduced, such streams do not decode. */
    bw_put(w, cl_code[s], cl_len[s]);
    if (s == SYM_REPEAT) bw_put(w, items[i].extra, 2);
    if (s == SYM_ZERO11) bw_put(w, items[i].extra, 3);
    if (s == SYM_ZERO138) bw_put(w, items[i].extra, 7);
  }
  return ORA_OK;
}

long ora_write_code(const uint8_t *lengths, int n, uint8_t *buf, size_t cap) {
  bitw w;
  bw_init(&w);
  int rc = write_code(&w, lengths, n);
  if (rc < 0) { free(w.buf); return rc; }
  memset(buf, 0, cap);
  size_t nb = bw_bytes(&w);
  if (nb > cap) nb = cap;
  if (nb) memcpy(buf, w.buf, nb);
  long bits = (long)w.nbits;
  free(w.buf);
  return bits;
}

/* ------------------------------------------------------------------------- */
/* entropy-coded image writer: vp8l.h:477-583                                  */
/* ------------------------------------------------------------------------- */
static int write_coded_image(bitw *w, const uint32_t *data, uint32_t n, uint32_t xsize,
                             int spatial, ora_encode_trace *trace) {
  bw_put(w, 0, 1);              /* no colour cache  vp8l.h:478 / :512 */
  if (spatial) bw_put(w, 0, 1); /* no meta-Huffman  vp8l.h:513 */
  ora_token *tok;
  size_t ntok;
  int rc = ora_lz77(data, n, &tok, &ntok);
  if (rc < 0) return rc;
  uint32_t histo[5][280];
  memset(histo, 0, sizeof(histo));
  for (size_t i = 0; i < ntok; i++) { /* vp8l.h:517-535 */
    if (tok[i].length == 0 && tok[i].distance == 0) {
      uint32_t p = tok[i].symbol;
      histo[0][CH_G(p)]++; histo[1][CH_R(p)]++; histo[2][CH_B(p)]++; histo[3][CH_A(p)]++;
    } else {
      uint32_t s, nb, eb;
      ora_prefix_encode(tok[i].length, &s, &nb, &eb);
      histo[0][256 + s]++;
      ora_prefix_encode(ora_dist_to_code(xsize, tok[i].distance), &s, &nb, &eb);
      histo[4][s]++;
    }
  }
  uint8_t len[5][280];
  uint32_t code[5][280];
  int used[5];
  for (int c = 0; c < 5; c++) { /* vp8l.h:537-540 */
    used[c] = ora_huffman_build(histo[c], kAlphabet[c], MAX_CODE_LEN, len[c], code[c]);
    if (used[c] < 0) { free(tok); return used[c]; }
    rc = write_code(w, len[c], kAlphabet[c]);
    if (rc < 0) { free(tok); return rc; }
  }
  for (size_t i = 0; i < ntok; i++) { /* vp8l.h:545-583 */
    if (tok[i].length == 0 && tok[i].distance == 0) {
      uint32_t p = tok[i].symbol;
      if (used[0] > 1) bw_put(w, code[0][CH_G(p)], len[0][CH_G(p)]);
      if (used[1] > 1) bw_put(w, code[1][CH_R(p)], len[1][CH_R(p)]);
      if (used[2] > 1) bw_put(w, code[2][CH_B(p)], len[2][CH_B(p)]);
      if (used[3] > 1) bw_put(w, code[3][CH_A(p)], len[3][CH_A(p)]);
    } else {
      uint32_t s, nb, eb;
      ora_prefix_encode(tok[i].length, &s, &nb, &eb);
      bw_put(w, code[0][256 + s], len[0][256 + s]);
      if (nb) bw_put(w, eb, nb);
      ora_prefix_encode(ora_dist_to_code(xsize, tok[i].distance), &s, &nb, &eb);
      if (used[4] > 1) bw_put(w, code[4][s], len[4][s]);
      if (nb) bw_put(w, eb, nb);
    }
  }
  if (trace && spatial) {
    trace->tokens = tok;
    trace->ntokens = ntok;
    memcpy(trace->histo, histo, sizeof(histo));
    memcpy(trace->lengths, len, sizeof(len));
    memcpy(trace->codes, code, sizeof(code));
  } else {
    free(tok);
  }
  return w->oom ? ORA_NOMEM : ORA_OK;
}

/* CreatePallete vp8l.h:372-380: distinct colours, ascending; 0 if more than 256. */
static uint32_t build_palette(const uint32_t *argb, size_t n, uint32_t *pal) {
  uint32_t np = 0;
  for (size_t i = 0; i < n; i++) {
    uint32_t v = argb[i];
    uint32_t lo = 0, hi = np;
    while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (pal[mid] < v) lo = mid + 1; else hi = mid; }
    if (lo < np && pal[lo] == v) continue;
    if (np == 256) return 0;
    memmove(pal + lo + 1, pal + lo, sizeof(uint32_t) * (np - lo));
    pal[lo] = v;
    np++;
  }
  return np;
}

static uint32_t palette_bits(uint32_t ncolors) { /* vp8l.h:423-426 / :132-135 */
  return ncolors > 16 ? 0 : ncolors > 4 ? 1 : ncolors > 2 ? 2 : 3;
}

void ora_trace_free(ora_encode_trace *t) {
  if (!t) return;
  free(t->residual);
  free(t->tokens);
  t->residual = NULL;
  t->tokens = NULL;
}

int ora_encode(const uint32_t *argb, uint32_t w, uint32_t h, uint8_t **out, size_t *out_len,
               ora_encode_trace *trace) {
  if (trace) memset(trace, 0, sizeof(*trace));
  if (!argb || w == 0 || h == 0) return ORA_INVALID_ARGB;  /* vp8l.h:587-588 */
  if (w > MAX_DIM || h > MAX_DIM) return ORA_TOO_BIG_ARGB; /* vp8l.h:589-590 */
  size_t n = (size_t)w * h;
  bitw bw;
  bw_init(&bw);
  /* write_info vp8l.h:460-467 */
  bw_put(&bw, 0, 32);
  bw_put(&bw, 0x2f, 8);
  bw_put(&bw, w - 1, 14);
  bw_put(&bw, h - 1, 14);
  bw_put(&bw, 0, 1);
  bw_put(&bw, 0, 3);

  uint32_t pal[257];
  uint32_t ncolors = build_palette(argb, n, pal);
  uint32_t *image = NULL;
  uint32_t xsize = w;
  int rc = ORA_OK;
  if (ncolors == 0) { /* vp8l.h:598-600 */
    image = (uint32_t *)malloc(sizeof(uint32_t) * n);
    if (!image) return ORA_NOMEM;
    memcpy(image, argb, sizeof(uint32_t) * n);
    ora_subtract_green(image, n);
    bw_put(&bw, 1, 1);
    bw_put(&bw, 2, 2);
  } else { /* ApplyColorIndexingTransform vp8l.h:420-459 */
    uint32_t bits = palette_bits(ncolors);
    xsize = (w + (1u << bits) - 1) >> bits;
    image = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)xsize * h);
    uint8_t *row = (uint8_t *)malloc(w);
    if (!image || !row) { free(image); free(row); return ORA_NOMEM; }
    for (uint32_t y = 0; y < h; y++) {
      for (uint32_t x = 0; x < w; x++) {
        uint32_t pix = argb[(size_t)y * w + x];
        for (uint32_t i = 0; i < ncolors; i++) if (pal[i] == pix) { row[x] = (uint8_t)i; break; }
      }
      ora_bundle(row, (int)w, (int)bits, image + (size_t)y * xsize);
    }
    free(row);
    bw_put(&bw, 1, 1);
    bw_put(&bw, 3, 2);
    bw_put(&bw, ncolors - 1, 8);
    if (trace) { trace->use_palette = 1; trace->ncolors = ncolors; trace->bits = bits;
                 memcpy(trace->palette, pal, sizeof(uint32_t) * ncolors); }
    uint32_t delta[256];
    memcpy(delta, pal, sizeof(uint32_t) * ncolors);
    for (uint32_t i = ncolors - 1; i >= 1; i--) delta[i] = px_sub(delta[i], delta[i - 1]); /* :455-456 */
    rc = write_coded_image(&bw, delta, ncolors, ncolors, 0, NULL); /* :457 */
  }
  if (rc == ORA_OK) {
    bw_put(&bw, 0, 1); /* no more transforms vp8l.h:609 */
    if (trace) { trace->xsize = xsize; trace->ysize = h; }
    rc = write_coded_image(&bw, image, (uint32_t)((size_t)xsize * h), xsize, 1, trace);
  }
  if (trace && rc == ORA_OK) trace->residual = image; else free(image);
  if (rc != ORA_OK) { free(bw.buf); if (trace) ora_trace_free(trace); return rc; }
  *out = bw.buf;
  *out_len = bw_bytes(&bw);
  return ORA_OK;
}

/* ------------------------------------------------------------------------- */
/* Huffman decoder side: huffman_coding.h:73-282, huffman_io.h:74-219          */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t children; /* <0 empty, 0 leaf, >0 offset to the left child (huffman_coding.h:29-71) */
  uint16_t sym;
} dnode;

typedef struct {
  dnode *nodes;
  uint32_t max_nodes, num_nodes;
} dtree;

static void dtree_free(dtree *t) { free(t->nodes); t->nodes = NULL; }

static int dtree_init(dtree *t, size_t leaves) { /* huffman_coding.h:109-119 */
  t->nodes = NULL;
  if (leaves == 0) return 0;
  t->max_nodes = (uint32_t)(2 * leaves - 1);
  t->nodes = (dnode *)malloc(sizeof(dnode) * t->max_nodes);
  if (!t->nodes) return 0;
  for (uint32_t i = 0; i < t->max_nodes; i++) { t->nodes[i].children = -1; t->nodes[i].sym = 0; }
  t->num_nodes = 1;
  return 1;
}

static int dtree_add(dtree *t, uint16_t sym, uint32_t code, int len) { /* :130-161 */
  uint32_t node = 0;
  while (len > 0) {
    len--;
    if (node >= t->max_nodes) return 0;
    if (t->nodes[node].children < 0) {
      if (t->num_nodes == t->max_nodes) return 0;
      t->nodes[node].children = (int32_t)(t->num_nodes - node);
      t->num_nodes += 2;
    } else if (t->nodes[node].children == 0) {
      return 0;
    }
    node += (uint32_t)t->nodes[node].children + ((code >> len) & 1u);
  }
  if (t->nodes[node].children < 0) t->nodes[node].children = 0;
  else if (t->nodes[node].children != 0) return 0;
  t->nodes[node].sym = sym;
  return 1;
}

static int dtree_from_lengths(dtree *t, const uint8_t *lengths, int n) { /* implicit_init :188-232 */
  size_t used = 0;
  int root = 0;
  for (int s = 0; s < n; s++) if (lengths[s]) { used++; root = s; if (lengths[s] > MAX_SANE_CODE_LEN) return 0; }
  if (!dtree_init(t, used)) return 0;
  if (used == 1) {
    if (!dtree_add(t, (uint16_t)root, 0, 0)) return 0;
  } else {
    uint32_t *codes = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    if (!codes) return 0;
    canonical_codes(lengths, n, codes);
    for (int s = 0; s < n; s++)
      if (lengths[s] && !dtree_add(t, (uint16_t)s, codes[s], lengths[s])) { free(codes); return 0; }
    free(codes);
  }
  return t->num_nodes == t->max_nodes;
}

static int dtree_read(const dtree *t, bitr *r) { /* huffman_io.h:85-91 */
  uint32_t node = 0;
  while (t->nodes[node].children != 0) node += (uint32_t)t->nodes[node].children + br_get1(r);
  return t->nodes[node].sym;
}

/* read_code huffman_io.h:147-195 + read_code_length :92-146.  Fills lengths[alphabet]
 * for a normal code; for a simple code reports the symbols. */
static int read_code_desc(bitr *r, int alphabet, uint8_t *lengths, int *is_simple,
                          uint16_t simple_syms[2], int *simple_n) {
  memset(lengths, 0, (size_t)alphabet);
  *is_simple = (int)br_get(r, 1);
  if (*is_simple) {
    int nsym = (int)br_get(r, 1) + 1;
    uint32_t first8 = br_get(r, 1);
    simple_syms[0] = (uint16_t)br_get(r, 1 + 7 * first8);
    simple_syms[1] = 0;
    if (nsym == 2) simple_syms[1] = (uint16_t)br_get(r, 8);
    *simple_n = nsym;
    return ORA_OK;
  }
  uint8_t cl[NUM_CL_SYMS];
  memset(cl, 0, sizeof(cl));
  uint32_t ncodes = br_get(r, CL_COUNT_BITS) + 4;
  if (ncodes > NUM_CL_SYMS) return ORA_INVALID_HUFFMAN; /* :184-185 */
  for (uint32_t i = 0; i < ncodes; i++) cl[cl_order((int)i)] = (uint8_t)br_get(r, CL_LEN_BITS);
  dtree t = {NULL, 0, 0};
  if (!dtree_from_lengths(&t, cl, NUM_CL_SYMS)) { dtree_free(&t); return ORA_INVALID_HUFFMAN; }
  uint16_t max_symbol;
  if (br_get(r, 1)) { /* :101-107 */
    uint32_t nb = 2 + 2 * br_get(r, 3);
    max_symbol = (uint16_t)(2 + br_get(r, nb)); /* symbol_t is 16-bit: truncation reproduced */
    if (max_symbol > alphabet) { dtree_free(&t); return ORA_INVALID_HUFFMAN; }
  } else {
    max_symbol = (uint16_t)alphabet;
  }
  int sym = 0, prev = 8;
  while (sym < alphabet) {
    if (max_symbol-- == 0) break;
    int cl_sym = dtree_read(&t, r);
    if (cl_sym < SYM_REPEAT) {
      lengths[sym++] = (uint8_t)cl_sym;
      if (cl_sym) prev = cl_sym;
    } else {
      int reps = 0, use_prev = 0;
      if (cl_sym == SYM_REPEAT) { reps = (int)br_get(r, 2) + 3; use_prev = 1; }
      if (cl_sym == SYM_ZERO11) reps = (int)br_get(r, 3) + 3;
      if (cl_sym == SYM_ZERO138) reps = (int)br_get(r, 7) + 11;
      if (sym + reps > alphabet) { dtree_free(&t); return ORA_INVALID_HUFFMAN; } /* :140-141 */
      for (int i = 0; i < reps; i++) lengths[sym++] = (uint8_t)(use_prev ? prev : 0);
    }
  }
  dtree_free(&t);
  return ORA_OK;
}

int ora_read_code(const uint8_t *data, size_t len, uint64_t *bitpos, int alphabet,
                  uint8_t *lengths, int *is_simple, uint16_t simple_syms[2], int *simple_n) {
  bitr r = {data, len, *bitpos};
  int rc = read_code_desc(&r, alphabet, lengths, is_simple, simple_syms, simple_n);
  *bitpos = r.pos;
  return rc;
}

typedef struct { dtree t[5]; } hgroup;

static void hgroup_free(hgroup *g) { for (int i = 0; i < 5; i++) dtree_free(&g->t[i]); }

static int hgroup_read(hgroup *g, bitr *r, uint32_t cache_size) { /* huffman_io.h:200-210 */
  memset(g, 0, sizeof(*g));
  uint8_t *lengths = (uint8_t *)malloc(280 + 2048);
  if (!lengths) return ORA_NOMEM;
  for (int c = 0; c < 5; c++) {
    int alphabet = kAlphabet[c] + (c == 0 ? (int)cache_size : 0);
    int simple, nsimple;
    uint16_t ss[2];
    int rc = read_code_desc(r, alphabet, lengths, &simple, ss, &nsimple);
    if (rc < 0) { free(lengths); hgroup_free(g); return rc; }
    int ok;
    if (simple) { /* explicit_init huffman_coding.h:167-187 */
      ok = dtree_init(&g->t[c], (size_t)nsimple);
      for (int i = 0; ok && i < nsimple; i++) {
        if (ss[i] >= alphabet) ok = 0;
        else ok = dtree_add(&g->t[c], ss[i], (uint32_t)i, nsimple - 1);
      }
      if (ok) ok = g->t[c].num_nodes == g->t[c].max_nodes;
    } else {
      ok = dtree_from_lengths(&g->t[c], lengths, alphabet);
    }
    if (!ok) { free(lengths); hgroup_free(g); return ORA_INVALID_HUFFMAN; }
  }
  free(lengths);
  return ORA_OK;
}

/* ------------------------------------------------------------------------- */
/* decoder: vp8l.h:83-345, color_cache.h:29-54                                  */
/* ------------------------------------------------------------------------- */
typedef struct {
  uint32_t bits;
  uint32_t *slots;
} ccache;

static uint32_t cache_key(uint32_t argb, uint32_t bits) { return (0x1e35a7bdu * argb) >> (32 - bits); }

static int read_cache_bits(bitr *r, uint32_t *bits) { /* vp8l.h:230-241 */
  *bits = 0;
  if (br_get(r, 1)) {
    uint32_t b = br_get(r, 4);
    if (b > MAX_CACHE_BITS) return ORA_INVALID_VP8L;
    *bits = b;
  }
  return ORA_OK;
}

/* ReadLZ77CodedImage vp8l.h:263-328.  groups[ngroups]; meta image (already reduced to
 * group indices) selects per pixel when ngroups > 1.  The reference does not bound-check
 * back references or run-off; out-of-range ones are reported as InvalidVP8L here. */
static int read_pixels(bitr *r, const hgroup *groups, uint32_t ngroups, const uint32_t *meta,
                       uint32_t meta_bits, uint32_t meta_xsize, uint32_t xsize, uint32_t ysize,
                       uint32_t *data, uint32_t cache_bits) {
  uint64_t total = (uint64_t)xsize * ysize;
  ccache cc = {cache_bits, NULL};
  if (cache_bits) {
    cc.slots = (uint32_t *)calloc((size_t)1 << cache_bits, sizeof(uint32_t));
    if (!cc.slots) return ORA_NOMEM;
  }
  uint64_t fill = 0, cached = 0;
  uint32_t x = 0, y = 0;
  int rc = ORA_OK;
  while (fill != total) {
    const hgroup *g = &groups[ngroups == 1 ? 0 : meta[(y >> meta_bits) * meta_xsize + (x >> meta_bits)]];
    int S = dtree_read(&g->t[0], r);
    if (S < 256) {
      uint32_t red = (uint32_t)dtree_read(&g->t[1], r);
      uint32_t blue = (uint32_t)dtree_read(&g->t[2], r);
      uint32_t alpha = (uint32_t)dtree_read(&g->t[3], r);
      data[fill++] = (alpha << 24) + (red << 16) + ((uint32_t)S << 8) + blue;
      if (++x >= xsize) { x = 0; y++; }
    } else if (S < 256 + 24) {
      uint32_t length = prefix_decode((uint32_t)S - 256, r);
      uint32_t dsym = (uint32_t)dtree_read(&g->t[4], r);
      uint32_t dcode = prefix_decode(dsym, r);
      uint32_t dist = ora_code_to_dist(xsize, dcode);
      if (dist > fill || fill + length > total) { rc = ORA_INVALID_VP8L; break; }
      for (uint32_t i = 0; i < length; i++, fill++) {
        data[fill] = data[fill - dist];
        if (++x >= xsize) { x = 0; y++; }
      }
    } else {
      uint32_t key = (uint32_t)S - 280;
      if (cc.slots) while (cached < fill) { cc.slots[cache_key(data[cached], cc.bits)] = data[cached]; cached++; }
      data[fill++] = cc.slots ? cc.slots[key] : 0;
      if (++x >= xsize) { x = 0; y++; }
    }
    /* The reference inserts lazily (vp8l.h:286-288,308-310,315-324); inserting every
     * finished pixel in order before the next cache lookup is equivalent. */
    if (cc.slots) while (cached < fill) { cc.slots[cache_key(data[cached], cc.bits)] = data[cached]; cached++; }
  }
  free(cc.slots);
  return rc;
}

/* ReadEntropyCodedImage vp8l.h:161-173 */
static int read_entropy_image(bitr *r, uint32_t xsize, uint32_t ysize, uint32_t *data) {
  uint32_t cache_bits;
  int rc = read_cache_bits(r, &cache_bits);
  if (rc < 0) return rc;
  hgroup g;
  rc = hgroup_read(&g, r, cache_bits ? 1u << cache_bits : 0);
  if (rc < 0) return rc;
  rc = read_pixels(r, &g, 1, NULL, 0, 0, xsize, ysize, data, cache_bits);
  hgroup_free(&g);
  return rc;
}

typedef struct {
  int type;
  uint32_t xsize, ysize, bits;
  uint32_t *data;
} xform;

static int inverse_predictor(uint32_t *argb, uint32_t w, uint32_t h, const uint32_t *modes,
                             uint32_t txsize, uint32_t bits) { /* transform.h:110-191 */
  for (uint32_t y = 0; y < h; y++)
    for (uint32_t x = 0; x < w; x++) {
      size_t i = (size_t)y * w + x;
      uint32_t P;
      if (x == 0 && y == 0) P = 0xff000000u;
      else if (x == 0) P = argb[i - w];
      else if (y == 0) P = argb[i - 1];
      else {
        uint32_t L = argb[i - 1], T = argb[i - w];
        uint32_t TR = (x == w - 1) ? L : argb[i - w + 1]; /* transform.h:137 */
        uint32_t TL = argb[i - w - 1];
        uint32_t mode = CH_G(modes[(y >> bits) * txsize + (x >> bits)]);
        switch (mode) {
          case 0: P = 0xff000000u; break;
          case 1: P = L; break;
          case 2: P = T; break;
          case 3: P = TR; break;
          case 4: P = TL; break;
          case 5: P = px_avg2(px_avg2(L, TR), T); break;
          case 6: P = px_avg2(L, TL); break;
          case 7: P = px_avg2(L, T); break;
          case 8: P = px_avg2(TL, T); break;
          case 9: P = px_avg2(T, TR); break;
          case 10: P = px_avg2(px_avg2(L, TL), px_avg2(T, TR)); break;
          case 11: P = px_select(T, L, TL); break;
          case 12: P = px_clamp_full(L, T, TL); break;
          case 13: P = px_clamp_half(px_avg2(L, T), TL); break;
          default: return ORA_INVALID_VP8L;
        }
      }
      argb[i] = px_add(argb[i], P);
    }
  return ORA_OK;
}

static int8_t color_delta(int8_t t, int8_t c) { return (int8_t)(((int)t * (int)c) >> 5); } /* :192-195 */

static void inverse_cross_color(uint32_t *argb, uint32_t w, uint32_t h, const uint32_t *elems,
                                uint32_t txsize, uint32_t bits) { /* transform.h:196-218 */
  for (uint32_t y = 0; y < h; y++)
    for (uint32_t x = 0; x < w; x++) {
      size_t i = (size_t)y * w + x;
      uint32_t e = elems[(y >> bits) * txsize + (x >> bits)];
      uint8_t r2b = (uint8_t)CH_R(e), g2b = (uint8_t)CH_G(e), g2r = (uint8_t)CH_B(e); /* :29-34 */
      uint8_t red = (uint8_t)CH_R(argb[i]), green = (uint8_t)CH_G(argb[i]), blue = (uint8_t)CH_B(argb[i]);
      red = (uint8_t)(red + color_delta((int8_t)g2r, (int8_t)green));
      blue = (uint8_t)(blue + color_delta((int8_t)g2b, (int8_t)green));
      blue = (uint8_t)(blue + color_delta((int8_t)r2b, (int8_t)red));
      argb[i] = (argb[i] & 0xff00ff00u) | ((uint32_t)red << 16) | blue;
    }
}

static int inverse_color_indexing(uint32_t *argb, uint32_t w, uint32_t h, const uint32_t *pal,
                                  uint32_t ncolors, uint32_t bits) { /* transform.h:59-87 */
  uint32_t per = 1u << bits, bpp = 8u >> bits, mask = (1u << bpp) - 1;
  uint32_t cix = (w + per - 1) >> bits;
  uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)cix * h);
  if (!idx) return ORA_NOMEM;
  memcpy(idx, argb, sizeof(uint32_t) * (size_t)cix * h);
  for (uint32_t y = 0; y < h; y++)
    for (uint32_t x = 0; x < w; x++) { /* the reference writes cix<<bits per row (overrun); only w kept */
      uint32_t g = CH_G(idx[(size_t)y * cix + (x >> bits)]);
      uint32_t k = (g >> ((x & (per - 1)) * bpp)) & mask;
      argb[(size_t)y * w + x] = k < ncolors ? pal[k] : 0; /* reference reads out of bounds */
    }
  free(idx);
  return ORA_OK;
}

int ora_inverse_transform(int type, uint32_t *argb, uint32_t w, uint32_t h, const uint32_t *tdata,
                          uint32_t txsize, uint32_t tbits) { /* transform.h:265-275 */
  switch (type) {
    case 0: return inverse_predictor(argb, w, h, tdata, txsize, tbits);
    case 1: inverse_cross_color(argb, w, h, tdata, txsize, tbits); return ORA_OK;
    case 2: /* transform.h:94-103 */
      for (size_t i = 0; i < (size_t)w * h; i++) {
        uint32_t p = argb[i], g = CH_G(p);
        argb[i] = (p & 0xff00ff00u) | (((CH_R(p) + g) & 0xff) << 16) | ((CH_B(p) + g) & 0xff);
      }
      return ORA_OK;
    case 3: return inverse_color_indexing(argb, w, h, tdata, txsize, tbits);
  }
  return ORA_INVALID_VP8L;
}

int ora_get_info(const uint8_t *data, size_t len, uint32_t *w, uint32_t *h) { /* vp8l.h:83-94 */
  bitr r = {data, len, 0};
  (void)br_get(&r, 32);
  if (br_get(&r, 8) != 0x2f) return ORA_UNSUPPORTED_VP8;
  *w = br_get(&r, 14) + 1;
  *h = br_get(&r, 14) + 1;
  return ORA_OK;
}

int ora_decode(const uint8_t *data, size_t len, uint32_t **argb_out, uint32_t *w_out, uint32_t *h_out) {
  bitr r = {data, len, 0};
  (void)br_get(&r, 32);
  if (br_get(&r, 8) != 0x2f) return ORA_UNSUPPORTED_VP8;
  uint32_t w = br_get(&r, 14) + 1, h = br_get(&r, 14) + 1;
  (void)br_get(&r, 1);
  (void)br_get(&r, 3);
  uint32_t *argb = (uint32_t *)calloc((size_t)w * h, sizeof(uint32_t)); /* vp8l.h:335-336 */
  if (!argb) return ORA_NOMEM;
  xform xf[4];
  int nxf = 0, seen[4] = {0, 0, 0, 0};
  uint32_t cix = 0;
  int rc = ORA_OK;
  while (rc == ORA_OK && br_get(&r, 1)) { /* vp8l.h:339-340, ReadTransform :101-154 */
    int type = (int)br_get(&r, 2);
    if (seen[type]) { rc = ORA_INVALID_VP8L; break; }
    seen[type] = 1;
    xform *t = &xf[nxf];
    memset(t, 0, sizeof(*t));
    t->type = type;
    if (type == 0 || type == 1) {
      t->bits = br_get(&r, 3) + 2;
      t->xsize = (w + (1u << t->bits) - 1) >> t->bits;
      t->ysize = (h + (1u << t->bits) - 1) >> t->bits;
      t->data = (uint32_t *)calloc((size_t)t->xsize * t->ysize, sizeof(uint32_t));
      if (!t->data) { rc = ORA_NOMEM; break; }
      nxf++;
      rc = read_entropy_image(&r, t->xsize, t->ysize, t->data);
    } else if (type == 3) {
      uint32_t ncolors = br_get(&r, 8) + 1;
      t->bits = palette_bits(ncolors);
      cix = (w + (1u << t->bits) - 1) >> t->bits;
      t->xsize = ncolors;
      t->ysize = 1;
      t->data = (uint32_t *)calloc(ncolors, sizeof(uint32_t));
      if (!t->data) { rc = ORA_NOMEM; break; }
      nxf++;
      rc = read_entropy_image(&r, ncolors, 1, t->data);
      uint8_t *b = (uint8_t *)t->data; /* vp8l.h:143-145 (little-endian byte view) */
      for (size_t i = 4; i < 4 * (size_t)ncolors; i++) b[i] = (uint8_t)(b[i] + b[i - 4]);
    } else {
      nxf++;
    }
  }
  if (rc == ORA_OK) { /* ReadSpatiallyCodedImage vp8l.h:180-223 */
    uint32_t cache_bits;
    rc = read_cache_bits(&r, &cache_bits);
    uint32_t cache_size = cache_bits ? 1u << cache_bits : 0;
    uint32_t *meta = NULL, meta_bits = 0, meta_xsize = 0, ngroups = 1;
    if (rc == ORA_OK && br_get(&r, 1)) {
      meta_bits = br_get(&r, 3) + 2;
      meta_xsize = (w + (1u << meta_bits) - 1) >> meta_bits;
      uint32_t meta_ysize = (h + (1u << meta_bits) - 1) >> meta_bits;
      meta = (uint32_t *)calloc((size_t)meta_xsize * meta_ysize, sizeof(uint32_t));
      if (!meta) rc = ORA_NOMEM;
      else rc = read_entropy_image(&r, meta_xsize, meta_ysize, meta);
      if (rc == ORA_OK)
        for (size_t i = 0; i < (size_t)meta_xsize * meta_ysize; i++) {
          meta[i] = (meta[i] >> 8) & 0xffff;
          if (meta[i] >= ngroups) ngroups = meta[i] + 1;
        }
    }
    hgroup *groups = NULL;
    uint32_t nread = 0;
    if (rc == ORA_OK) {
      groups = (hgroup *)calloc(ngroups, sizeof(hgroup));
      if (!groups) rc = ORA_NOMEM;
    }
    for (; rc == ORA_OK && nread < ngroups; nread++) {
      rc = hgroup_read(&groups[nread], &r, cache_size);
      if (rc < 0) break;
    }
    if (rc == ORA_OK)
      rc = read_pixels(&r, groups, ngroups, meta, meta_bits, meta_xsize, cix ? cix : w, h, argb, cache_bits);
    for (uint32_t i = 0; i < nread; i++) hgroup_free(&groups[i]);
    free(groups);
    free(meta);
  }
  for (int i = nxf - 1; rc == ORA_OK && i >= 0; i--) /* reverse order vp8l.h:110,343-344 */
    rc = ora_inverse_transform(xf[i].type, argb, w, h, xf[i].data, xf[i].xsize, xf[i].bits);
  for (int i = 0; i < nxf; i++) free(xf[i].data);
  if (rc != ORA_OK) { free(argb); return rc; }
  *argb_out = argb;
  *w_out = w;
  *h_out = h;
  return ORA_OK;
}

/* ------------------------------------------------------------------------- */
/* synthetic generators (seeded LCG; definitions fixed by the known-answer set) */
/* ------------------------------------------------------------------------- */
static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

void ora_gen_image(int mode, uint32_t w, uint32_t h, uint32_t seed, uint32_t ncolors, uint32_t *argb) {
  uint32_t s = seed;
  uint32_t pal[256];
  if (mode == 5) {
    uint32_t t = 777;
    if (ncolors == 0 || ncolors > 256) ncolors = 200;
    for (uint32_t i = 0; i < ncolors; i++) pal[i] = 0xff000000u | (lcg(&t) & 0xffffffu);
  }
  uint32_t dw = w > 1 ? w - 1 : 1, dh = h > 1 ? h - 1 : 1;
  for (uint32_t y = 0; y < h; y++)
    for (uint32_t x = 0; x < w; x++) {
      uint32_t p;
      if (mode == 3) {
        int n[3];
        for (int c = 0; c < 3; c++) { uint32_t a = lcg(&s), b = lcg(&s); n[c] = (int)(a & 3) + (int)(b & 3) - 3; }
        int L = (int)(x * 160 / dw + y * 80 / dh);
        p = 0xff000000u | (clamp255(L + 10 + n[0]) << 16) | (clamp255(L + n[1]) << 8) |
            clamp255(L * 3 / 4 + 5 + n[2]);
      } else if (mode == 4) {
        uint32_t bs = ((y / 16) * 7919u + (x / 16) * 104729u) * 2654435761u;
        p = 0xff000000u | ((bs >> 8) & 0xf0f0f0u);
        if ((lcg(&s) & 31) == 0) p ^= 0x00010101u;
      } else {
        uint32_t l = lcg(&s);
        p = pal[((x / 7) * 13 + (y / 5) * 31 + ((l & 15) == 0)) % ncolors];
      }
      argb[(size_t)y * w + x] = p;
    }
}

/* ------------------------------------------------------------------------- */
/* Test-vector encoder for the decoder-only features (see vp8l_oracle.h).       */
/* Forward transforms here are the exact inverses of transform.h:110-218; the   */
/* container grammar follows what the reference decoder parses (vp8l.h:101-328).*/
/* ------------------------------------------------------------------------- */
static uint32_t predict_pixel(const uint32_t *img, uint32_t w, uint32_t x, uint32_t y, uint32_t mode) {
  size_t i = (size_t)y * w + x;
  if (x == 0 && y == 0) return 0xff000000u;
  if (x == 0) return img[i - w];
  if (y == 0) return img[i - 1];
  uint32_t L = img[i - 1], T = img[i - w], TR = (x == w - 1) ? L : img[i - w + 1], TL = img[i - w - 1];
  switch (mode) {
    case 0: return 0xff000000u;
    case 1: return L;
    case 2: return T;
    case 3: return TR;
    case 4: return TL;
    case 5: return px_avg2(px_avg2(L, TR), T);
    case 6: return px_avg2(L, TL);
    case 7: return px_avg2(L, T);
    case 8: return px_avg2(TL, T);
    case 9: return px_avg2(T, TR);
    case 10: return px_avg2(px_avg2(L, TL), px_avg2(T, TR));
    case 11: return px_select(T, L, TL);
    case 12: return px_clamp_full(L, T, TL);
    default: return px_clamp_half(px_avg2(L, T), TL);
  }
}

typedef struct {
  uint32_t kind; /* 0 literal, 1 match, 2 cache */
  uint32_t a, b; /* literal: pixel; match: length, distance; cache: key */
  uint32_t group;
} xtoken;

/* stray_cache: test vectors for decoders only.  Every stray_cache-th literal becomes a colour-cache token with an index
 * that has nothing to do with the pixel (slots nobody wrote included), so the stream no longer decodes to the input; what
 * it decodes to is defined by vp8l.h:315-324 all the same, and two decoders must agree on it. */
static int write_general_image(bitw *w, const uint32_t *data, uint32_t n, uint32_t xsize,
                               uint32_t cache_bits, const uint32_t *meta, uint32_t meta_bits,
                               uint32_t meta_xsize, uint32_t ngroups, uint32_t stray_cache) {
  uint32_t cache_size = cache_bits ? 1u << cache_bits : 0;
  int green_n = 280 + (int)cache_size;
  uint32_t *cache = cache_size ? (uint32_t *)calloc(cache_size, sizeof(uint32_t)) : NULL;
  xtoken *tok = (xtoken *)malloc(sizeof(xtoken) * (n ? n : 1));
  size_t ntok = 0;
  if (!tok || (cache_size && !cache)) { free(tok); free(cache); return ORA_NOMEM; }
  for (uint32_t p = 0; p < n;) {
    uint32_t x = p % xsize, y = p / xsize, l, d, adv;
    uint32_t g = ngroups > 1 ? meta[(y >> meta_bits) * meta_xsize + (x >> meta_bits)] : 0;
    ora_lz77_best(data, n, p, &l, &d);
    if (l) { tok[ntok].kind = 1; tok[ntok].a = l; tok[ntok].b = d; adv = l; }
    else if (cache && cache[cache_key(data[p], cache_bits)] == data[p]) {
      tok[ntok].kind = 2; tok[ntok].a = cache_key(data[p], cache_bits); tok[ntok].b = 0; adv = 1;
    } else { tok[ntok].kind = 0; tok[ntok].a = data[p]; tok[ntok].b = 0; adv = 1; }
    if (cache && stray_cache && tok[ntok].kind == 0 && ntok % stray_cache == 0) {
      tok[ntok].kind = 2; tok[ntok].a = (uint32_t)((ntok * 7 + 3) % cache_size);
    }
    tok[ntok++].group = g;
    if (cache) for (uint32_t i = 0; i < adv; i++) cache[cache_key(data[p + i], cache_bits)] = data[p + i];
    p += adv;
  }
  free(cache);
  size_t gstride = (size_t)green_n;
  uint32_t *histo = (uint32_t *)calloc((size_t)ngroups * 5 * gstride, sizeof(uint32_t));
  uint8_t *len = (uint8_t *)calloc((size_t)ngroups * 5 * gstride, 1);
  uint32_t *code = (uint32_t *)calloc((size_t)ngroups * 5 * gstride, sizeof(uint32_t));
  int *used = (int *)calloc((size_t)ngroups * 5, sizeof(int));
  int rc = ORA_OK;
  if (!histo || !len || !code || !used) rc = ORA_NOMEM;
#define HG(g, c) (histo + ((size_t)(g) * 5 + (c)) * gstride)
#define LG(g, c) (len + ((size_t)(g) * 5 + (c)) * gstride)
#define CG(g, c) (code + ((size_t)(g) * 5 + (c)) * gstride)
  for (size_t i = 0; rc == ORA_OK && i < ntok; i++) {
    uint32_t g = tok[i].group, s, nb, eb;
    if (tok[i].kind == 0) {
      uint32_t p = tok[i].a;
      HG(g, 0)[CH_G(p)]++; HG(g, 1)[CH_R(p)]++; HG(g, 2)[CH_B(p)]++; HG(g, 3)[CH_A(p)]++;
    } else if (tok[i].kind == 1) {
      ora_prefix_encode(tok[i].a, &s, &nb, &eb);
      HG(g, 0)[256 + s]++;
      ora_prefix_encode(ora_dist_to_code(xsize, tok[i].b), &s, &nb, &eb);
      HG(g, 4)[s]++;
    } else {
      HG(g, 0)[280 + tok[i].a]++;
    }
  }
  for (uint32_t g = 0; rc == ORA_OK && g < ngroups; g++)
    for (int c = 0; rc == ORA_OK && c < 5; c++) {
      int alpha = c == 0 ? green_n : kAlphabet[c];
      int u = ora_huffman_build(HG(g, c), alpha, MAX_CODE_LEN, LG(g, c), CG(g, c));
      if (u < 0) { rc = u; break; }
      used[g * 5 + c] = u;
      rc = write_code(w, LG(g, c), alpha);
    }
  for (size_t i = 0; rc == ORA_OK && i < ntok; i++) {
    uint32_t g = tok[i].group, s, nb, eb;
    const int *u = used + g * 5;
    if (tok[i].kind == 0) {
      uint32_t p = tok[i].a;
      if (u[0] > 1) bw_put(w, CG(g, 0)[CH_G(p)], LG(g, 0)[CH_G(p)]);
      if (u[1] > 1) bw_put(w, CG(g, 1)[CH_R(p)], LG(g, 1)[CH_R(p)]);
      if (u[2] > 1) bw_put(w, CG(g, 2)[CH_B(p)], LG(g, 2)[CH_B(p)]);
      if (u[3] > 1) bw_put(w, CG(g, 3)[CH_A(p)], LG(g, 3)[CH_A(p)]);
    } else if (tok[i].kind == 1) {
      ora_prefix_encode(tok[i].a, &s, &nb, &eb);
      if (u[0] > 1) bw_put(w, CG(g, 0)[256 + s], LG(g, 0)[256 + s]);
      if (nb) bw_put(w, eb, nb);
      ora_prefix_encode(ora_dist_to_code(xsize, tok[i].b), &s, &nb, &eb);
      if (u[4] > 1) bw_put(w, CG(g, 4)[s], LG(g, 4)[s]);
      if (nb) bw_put(w, eb, nb);
    } else {
      if (u[0] > 1) bw_put(w, CG(g, 0)[280 + tok[i].a], LG(g, 0)[280 + tok[i].a]);
    }
  }
#undef HG
#undef LG
#undef CG
  free(histo); free(len); free(code); free(used); free(tok);
  if (rc == ORA_OK && w->oom) rc = ORA_NOMEM;
  return rc;
}

/* entropy_image := cache_info huffman_group pixels (vp8l.h:161-173); written without a cache */
static int write_sub_image(bitw *w, const uint32_t *data, uint32_t n, uint32_t xsize) {
  bw_put(w, 0, 1);
  return write_general_image(w, data, n, xsize, 0, NULL, 0, 0, 1, 0);
}

int ora_encode_ext(const uint32_t *argb, uint32_t w, uint32_t h, const ora_ext_params *prm,
                   uint8_t **out, size_t *out_len) {
  if (!argb || w == 0 || h == 0) return ORA_INVALID_ARGB;
  if (w > MAX_DIM || h > MAX_DIM) return ORA_TOO_BIG_ARGB;
  size_t n = (size_t)w * h;
  bitw bw;
  bw_init(&bw);
  bw_put(&bw, 0, 32); bw_put(&bw, 0x2f, 8); bw_put(&bw, w - 1, 14); bw_put(&bw, h - 1, 14);
  bw_put(&bw, 0, 1); bw_put(&bw, 0, 3);
  uint32_t *img = (uint32_t *)malloc(sizeof(uint32_t) * n);
  if (!img) return ORA_NOMEM;
  memcpy(img, argb, sizeof(uint32_t) * n);
  int rc = ORA_OK;
  uint32_t xsize = w;
  uint32_t pal[257];
  uint32_t ncolors = prm->use_palette ? build_palette(argb, n, pal) : 0;
  if (ncolors) { /* colour indexing alone (other pixel transforms would act on bundled data) */
    uint32_t bits = palette_bits(ncolors);
    xsize = (w + (1u << bits) - 1) >> bits;
    uint8_t *row = (uint8_t *)malloc(w);
    uint32_t *packed = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)xsize * h);
    if (!row || !packed) { free(row); free(packed); free(img); return ORA_NOMEM; }
    for (uint32_t y = 0; y < h; y++) {
      for (uint32_t x = 0; x < w; x++) {
        uint32_t pix = argb[(size_t)y * w + x], lo = 0, hi = ncolors;
        while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (pal[mid] < pix) lo = mid + 1; else hi = mid; }
        row[x] = (uint8_t)lo;
      }
      ora_bundle(row, (int)w, (int)bits, packed + (size_t)y * xsize);
    }
    free(row); free(img);
    img = packed;
    bw_put(&bw, 1, 1); bw_put(&bw, 3, 2); bw_put(&bw, ncolors - 1, 8);
    uint32_t delta[256];
    memcpy(delta, pal, sizeof(uint32_t) * ncolors);
    for (uint32_t i = ncolors - 1; i >= 1; i--) delta[i] = px_sub(delta[i], delta[i - 1]);
    rc = write_sub_image(&bw, delta, ncolors, ncolors);
  } else {
    /* the order transforms are WRITTEN (and applied forward); the decoder undoes them last-first */
    static const int kOrders[6][3] = {{2, 0, 1}, {0, 1, 2}, {1, 0, 2}, {2, 1, 0}, {0, 2, 1}, {1, 2, 0}};
    const int *order = kOrders[((prm->transform_order % 6) + 6) % 6];
    for (int k = 0; rc == ORA_OK && k < 3; k++) {
      int type = order[k];
      if (type == 2 && prm->use_subtract_green) {
        ora_subtract_green(img, n);
        bw_put(&bw, 1, 1); bw_put(&bw, 2, 2);
      } else if (type == 0 && prm->predictor_bits) {
        uint32_t bits = (uint32_t)prm->predictor_bits;
        uint32_t tx = (w + (1u << bits) - 1) >> bits, ty = (h + (1u << bits) - 1) >> bits;
        uint32_t *modes = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)tx * ty);
        uint32_t *res = (uint32_t *)malloc(sizeof(uint32_t) * n);
        if (!modes || !res) { free(modes); free(res); rc = ORA_NOMEM; break; }
        for (uint32_t i = 0; i < tx * ty; i++)
          modes[i] = prm->predictor_modes ? prm->predictor_modes[i] : 0xff000000u | (((i * 5 + i / tx) % 14) << 8);
        for (uint32_t y = 0; y < h; y++)
          for (uint32_t x = 0; x < w; x++) {
            uint32_t mode = CH_G(modes[(y >> bits) * tx + (x >> bits)]);
            res[(size_t)y * w + x] = px_sub(img[(size_t)y * w + x], predict_pixel(img, w, x, y, mode));
          }
        memcpy(img, res, sizeof(uint32_t) * n);
        free(res);
        bw_put(&bw, 1, 1); bw_put(&bw, 0, 2); bw_put(&bw, bits - 2, 3);
        rc = write_sub_image(&bw, modes, tx * ty, tx);
        free(modes);
      } else if (type == 1 && prm->color_bits) {
        uint32_t bits = (uint32_t)prm->color_bits;
        uint32_t tx = (w + (1u << bits) - 1) >> bits, ty = (h + (1u << bits) - 1) >> bits;
        uint32_t *el = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)tx * ty);
        if (!el) { rc = ORA_NOMEM; break; }
        for (uint32_t i = 0; i < tx * ty; i++)
          el[i] = prm->color_elems ? prm->color_elems[i]
                                   : 0xff000000u | (((i * 7 + i / 3) % 23) << 16) | (((i * 5) % 13 + 240) << 8) | ((i * 3 + i / 5) % 29);
          /* (not a permutation of 0..255: 256 equally likely values would give all-8 code lengths, which the
           * reference serialises into a stream its own reader rejects) */
        for (uint32_t y = 0; y < h; y++)
          for (uint32_t x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            uint32_t e = el[(y >> bits) * tx + (x >> bits)];
            uint8_t r2b = (uint8_t)CH_R(e), g2b = (uint8_t)CH_G(e), g2r = (uint8_t)CH_B(e);
            uint8_t red = (uint8_t)CH_R(img[i]), green = (uint8_t)CH_G(img[i]), blue = (uint8_t)CH_B(img[i]);
            uint8_t nr = (uint8_t)(red - color_delta((int8_t)g2r, (int8_t)green));
            uint8_t nb = (uint8_t)(blue - color_delta((int8_t)g2b, (int8_t)green));
            nb = (uint8_t)(nb - color_delta((int8_t)r2b, (int8_t)red));
            img[i] = (img[i] & 0xff00ff00u) | ((uint32_t)nr << 16) | nb;
          }
        bw_put(&bw, 1, 1); bw_put(&bw, 1, 2); bw_put(&bw, bits - 2, 3);
        rc = write_sub_image(&bw, el, tx * ty, tx);
        free(el);
      }
    }
  }
  if (rc == ORA_OK) {
    bw_put(&bw, 0, 1); /* no more transforms */
    uint32_t cache_bits = (uint32_t)prm->cache_bits;
    if (cache_bits) { bw_put(&bw, 1, 1); bw_put(&bw, cache_bits, 4); } else bw_put(&bw, 0, 1);
    uint32_t *meta = NULL, mbits = 0, mxs = 0, ngroups = 1;
    if (prm->meta_bits) {
      /* the entropy image is addressed with IMAGE width/height even when pixels are bundled (vp8l.h:195-196) */
      mbits = (uint32_t)prm->meta_bits;
      mxs = (w + (1u << mbits) - 1) >> mbits;
      uint32_t mys = (h + (1u << mbits) - 1) >> mbits;
      ngroups = prm->meta_groups > 0 ? (uint32_t)prm->meta_groups : 2;
      meta = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)mxs * mys);
      uint32_t *mpix = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)mxs * mys);
      if (!meta || !mpix) { free(meta); free(mpix); free(img); free(bw.buf); return ORA_NOMEM; }
      uint32_t maxg = 0;
      for (uint32_t i = 0; i < mxs * mys; i++) {
        meta[i] = prm->meta_image ? prm->meta_image[i] : (i % mxs + 2 * (i / mxs)) % ngroups;
        if (meta[i] > maxg) maxg = meta[i];
        mpix[i] = 0xff000000u | (meta[i] << 8);
      }
      ngroups = maxg + 1; /* the decoder derives the group count from the largest index (vp8l.h:205-214) */
      bw_put(&bw, 1, 1); bw_put(&bw, mbits - 2, 3);
      rc = write_sub_image(&bw, mpix, mxs * mys, mxs);
      free(mpix);
    } else {
      bw_put(&bw, 0, 1);
    }
    if (rc == ORA_OK)
      rc = write_general_image(&bw, img, (uint32_t)((size_t)xsize * h), xsize, cache_bits, meta, mbits, mxs, ngroups,
                               prm->stray_cache > 0 ? (uint32_t)prm->stray_cache : 0u);
    free(meta);
  }
  free(img);
  if (rc != ORA_OK) { free(bw.buf); return rc; }
  *out = bw.buf;
  *out_len = bw_bytes(&bw);
  return ORA_OK;
}

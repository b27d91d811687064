// vp8l_encode_ext.h -- the coder with a colour cache and several Huffman groups (a meta-Huffman entropy image).
//
// The reference ENCODER writes neither (vp8l.h:594-610: one group, "no cache"); its DECODER reads both
// (ReadSpatiallyCodedImage vp8l.h:180-223, ReadColorCacheBits :230-241, SelectMetaHuffman :250-256, the pixel loop
// :263-328, color_cache.h:29-54).  This file is the encoder side, used when VP8L_OPT_CACHE_BITS or VP8L_OPT_META_BITS is
// set; the transforms, the match search and the parse in front of it are the default path's.  Included by
// vp8l_encode.hip (it uses that file's bit writers and scans).
//
// Tokens.  At a token start p the parse gives a match (length > 0) or one pixel.  One pixel is a cache reference when
// the decoder's cache holds exactly that pixel at that moment, i.e. when the LAST pixel before p with the same cache key
// is equal to it (or, with no such pixel, when the pixel is 0: the cache starts zeroed); else a literal.  Every pixel is
// inserted, also those a match copies.  "The last pixel before p with the same key" over a whole image:
//   k_cache_rows   per superblock, the last position of every key inside it (LDS atomicMax)
//   k_cache_scan   per image and key, a serial walk over the superblocks turns that into "last position before the
//                  superblock"
//   k_cache_hits   per superblock: the same for its four quarters in LDS, then one wave per quarter walks its 4096
//                  positions 64 at a time: a position takes the table's entry, or the nearest lower lane of its own
//                  chunk with the same key (found by ballots, only for keys that occur twice in the chunk)
// Groups.  One Huffman group per tile of 2^bits coded pixels a side: tiles are ranked by the rough bit cost of their
// residuals (k_meta_feature) and cut into equally large classes (k_meta_assign), so that flat and busy regions get
// codes of their own.  Histograms per group (k_ext_histogram), codes and their descriptions per group and code in
// parallel (k_ext_codes), concatenated by the header kernel behind the entropy image; k_ext_bitcount / k_ext_emit are
// k_bitcount / k_emit with the tables of the position's group read from HBM.
#pragma once

namespace vp8l {

__device__ __forceinline__ uint32_t ext_code_offset(int c) {
  return c == kGreen ? 0u : c == kRed ? kExtRed : c == kBlue ? kExtBlue : c == kAlpha ? kExtAlpha : kExtDist;
}

// ---------------------------------------------------------------------------------------------
// Colour cache
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cache_rows(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                    const uint32_t *__restrict__ sb_first, uint32_t n, const uint32_t *__restrict__ pix,
                                                    ExtArgs x) {
  __shared__ uint32_t last[1u << kMaxCacheBits];
  const uint32_t sb = blockIdx.x, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  const uint32_t M = st[img].M, size = 1u << x.cache_bits;
  const uint32_t p0 = (sb - sb_first[img]) * kSuper;
  for (uint32_t i = tid; i < size; i += 256) last[i] = 0;
  __syncthreads();
  const uint32_t *data = pix + imgs[img].pos0;
  if (p0 < M)
    for (uint32_t k = 0; k < kSuper / 256; k++) {
      const uint32_t p = p0 + k * 256 + tid;
      if (p < M) atomicMax(&last[cache_key(data[p], x.cache_bits)], p + 1);
    }
  __syncthreads();
  for (uint32_t i = tid; i < size; i += 256) x.rows[(size_t)sb * size + i] = last[i];
}

// rows[sb][key] := last position + 1 with that key BEFORE superblock sb of its image (0: none)
__global__ __launch_bounds__(256) void k_cache_scan(const EncState *__restrict__ st, const uint32_t *__restrict__ sb_first, ExtArgs x) {
  const uint32_t img = blockIdx.x, size = 1u << x.cache_bits;
  const uint32_t nsb = (st[img].M + kTailPad + kSuper - 1) / kSuper, sb0 = sb_first[img];
  for (uint32_t key = threadIdx.x; key < size; key += 256) {
    uint32_t carry = 0;
    for (uint32_t s = 0; s < nsb; s++) {
      uint32_t *slot = x.rows + (size_t)(sb0 + s) * size + key;
      const uint32_t t = *slot;
      *slot = carry;
      if (t) carry = t;
    }
  }
}

__global__ __launch_bounds__(256) void k_cache_hits(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                    const uint32_t *__restrict__ sb_first, uint32_t n, const uint32_t *__restrict__ pix,
                                                    const uint32_t *__restrict__ match, const uint32_t *__restrict__ flags,
                                                    uint32_t *__restrict__ hitbits, ExtArgs x) {
  constexpr uint32_t kQuarter = kSuper / 4;
  __shared__ uint32_t cur[4][1u << kMaxCacheBits];  // per quarter: last position + 1 of every key before the position being looked at
  const uint32_t sb = blockIdx.x, img = find_image(sb_first, n, sb), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t M = st[img].M, size = 1u << x.cache_bits, bits = x.cache_bits;
  const uint32_t p0 = (sb - sb_first[img]) * kSuper;
  if (p0 >= M) return;
  const uint64_t pos0 = imgs[img].pos0;
  const uint32_t *data = pix + pos0;
  for (uint32_t i = tid; i < 4 * size; i += 256) cur[i / size][i % size] = 0;
  __syncthreads();
  for (uint32_t k = 0; k < kSuper / 256; k++) {
    const uint32_t o = k * 256 + tid, p = p0 + o;
    if (p < M) atomicMax(&cur[o / kQuarter][cache_key(data[p], bits)], p + 1);
  }
  __syncthreads();
  for (uint32_t key = tid; key < size; key += 256) {
    uint32_t carry = x.rows[(size_t)sb * size + key];
    for (uint32_t q = 0; q < 4; q++) {
      const uint32_t t = cur[q][key];
      cur[q][key] = carry;
      if (t) carry = t;
    }
  }
  __syncthreads();
  uint32_t *mine = cur[wave];
  for (uint32_t c = 0; c < kQuarter / 64; c++) {
    const uint32_t base = p0 + wave * kQuarter + c * 64, p = base + lane;
    if (base >= M) break;  // wave-uniform
    const bool valid = p < M;
    const uint32_t px = valid ? data[p] : 0u, key = cache_key(px, bits);
    uint32_t prev = 0;
    bool lost = false;
    if (valid) {
      prev = mine[key];
      atomicMax(&mine[key], p + 1);
    }
    __builtin_amdgcn_wave_barrier();
    if (valid) lost = mine[key] != p + 1;   // a higher lane of this chunk has the same key
    uint64_t pend = __ballot(lost);
    while (pend) {
      const uint32_t l0 = (uint32_t)__builtin_ctzll(pend);
      const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)l0);
      const uint64_t m = __ballot(valid && key == k0);
      const uint64_t below = m & ((1ull << lane) - 1ull);
      if (((m >> lane) & 1ull) && below) prev = base + (63u - (uint32_t)__builtin_clzll(below)) + 1u;
      pend &= ~m;
    }
    // only a pixel that the parse codes by itself can become a cache reference
    const uint64_t gp = pos0 + p;
    bool hit = false;
    if (valid && ((flags[gp >> 5] >> (gp & 31)) & 1u) && (match[gp] & 0xffu) == 0) hit = prev ? data[prev - 1] == px : px == 0u;
    const uint64_t hb = __ballot(hit);
    if (lane == 0) { hitbits[gp >> 5] = (uint32_t)hb; hitbits[(gp >> 5) + 1] = (uint32_t)(hb >> 32); }
  }
}

// ---------------------------------------------------------------------------------------------
// Huffman groups
// ---------------------------------------------------------------------------------------------
// The tile grid of the entropy image: tiles of 2^bits pixels a side over the IMAGE width and height (vp8l.h:195-196), the tile
// size grown until the header coder takes the image (kSmallMax tiles).  Positions are looked up with CODED coordinates (:250-256).
__device__ __forceinline__ void meta_grid(uint32_t w, uint32_t h, uint32_t want_bits, uint32_t &bits, uint32_t &tx, uint32_t &ty) {
  bits = want_bits;
  while (bits < 9 && ((w + (1u << bits) - 1) >> bits) * ((h + (1u << bits) - 1) >> bits) > kSmallMax) bits++;
  tx = (w + (1u << bits) - 1) >> bits;
  ty = (h + (1u << bits) - 1) >> bits;
}

__global__ __launch_bounds__(64) void k_meta_feature(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                     const uint32_t *__restrict__ pix, ExtArgs x) {
  const uint32_t img = blockIdx.x / kSmallMax, tile = blockIdx.x % kSmallMax, lane = threadIdx.x;
  const EncState &s = st[img];
  if (s.meta_bits == 0 || tile >= s.meta_tx * s.meta_ty) return;
  const uint32_t xsize = s.xsize, h = imgs[img].h, bits = s.meta_bits, side = 1u << bits;
  const uint32_t x0 = (tile % s.meta_tx) << bits, y0 = (tile / s.meta_tx) << bits;
  const uint32_t *src = pix + imgs[img].pos0;
  const uint32_t per = side < 32 ? side : 32, step = side / per;  // up to 32 x 32 samples
  uint32_t cost = 0, cnt = 0;
  for (uint32_t i = lane; i < per * per; i += 64) {
    const uint32_t xx = x0 + (i % per) * step, yy = y0 + (i / per) * step;
    if (xx < xsize && yy < h) { cost += abs_cost(src[(uint64_t)yy * xsize + xx] & 0x00ffffffu); cnt++; }
  }
  cost = wave_sum_u32(cost);
  cnt = wave_sum_u32(cnt);
  if (lane == 0) x.feature[(size_t)img * kSmallMax + tile] = cnt ? cost * 16u / cnt : 0u;
}

__global__ __launch_bounds__(256) void k_meta_assign(EncState *__restrict__ st, ExtArgs x) {
  __shared__ uint32_t f[kSmallMax];
  const uint32_t img = blockIdx.x, tid = threadIdx.x;
  EncState &s = st[img];
  const uint32_t nt = s.meta_tx * s.meta_ty;
  if (s.meta_bits == 0 || nt == 0) { if (tid == 0) s.ngroups = 1; return; }
  for (uint32_t i = tid; i < nt; i += 256) f[i] = x.feature[(size_t)img * kSmallMax + i];
  __syncthreads();
  const uint32_t G = x.groups < nt ? x.groups : nt;
  for (uint32_t i = tid; i < nt; i += 256) {
    const uint32_t v = f[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < nt; j++) rank += (f[j] < v || (f[j] == v && j < i)) ? 1u : 0u;
    x.metaimg[(size_t)img * kSmallMax + i] = rank * G / nt;  // every class is used: the largest index is G - 1
  }
  if (tid == 0) s.ngroups = G;
}

// The group of coded position p.
struct MetaLookup {
  uint32_t xsize, bits, tx;
  const uint8_t *gmap;  // LDS: group of every tile
  bool multi;
  __device__ __forceinline__ uint32_t group(uint32_t p) const {
    if (!multi) return 0;
    const uint32_t y = p / xsize, xx = p - y * xsize;
    return gmap[(y >> bits) * tx + (xx >> bits)];
  }
};
__device__ __forceinline__ MetaLookup meta_lookup_load(const EncState &s, const ExtArgs &x, uint32_t img, uint8_t *gmap, uint32_t tid, uint32_t threads) {
  MetaLookup m;
  m.xsize = s.xsize; m.bits = s.meta_bits; m.tx = s.meta_tx; m.gmap = gmap; m.multi = s.ngroups > 1;
  if (m.multi)
    for (uint32_t i = tid; i < s.meta_tx * s.meta_ty; i += threads) gmap[i] = (uint8_t)x.metaimg[(size_t)img * kSmallMax + i];
  return m;  // a barrier must follow before the first lookup
}

__global__ __launch_bounds__(256) void k_ext_histogram(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                       const uint32_t *__restrict__ sb_first, uint32_t n,
                                                       const uint32_t *__restrict__ pix, const uint32_t *__restrict__ match,
                                                       const uint32_t *__restrict__ flags, ExtArgs x) {
  __shared__ uint32_t h[kExtMaxGroups][kNumCodes * kCodeStride];  // everything but the cache symbols, which go straight to HBM
  __shared__ uint8_t gmap[kSmallMax];
  const uint32_t sb = blockIdx.x, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  const EncState &s = st[img];
  const uint32_t M = s.M;
  const uint32_t p0 = (sb - sb_first[img]) * kSuper;
  if (p0 >= M) return;
  const uint32_t G = s.ngroups;
  for (uint32_t i = tid; i < G * kNumCodes * kCodeStride; i += 256) (&h[0][0])[i] = 0;
  const MetaLookup ml = meta_lookup_load(s, x, img, gmap, tid, 256);
  __syncthreads();
  const uint64_t pos0 = imgs[img].pos0;
  uint32_t *gh = x.histo + (size_t)img * kExtMaxGroups * kExtSyms;
  for (uint32_t k = 0; k < kSuper / 256; k++) {
    const uint32_t p = p0 + k * 256 + tid;
    if (p >= M) break;
    const uint64_t gp = pos0 + p;
    if (!((flags[gp >> 5] >> (gp & 31)) & 1u)) continue;
    const uint32_t m = match[gp], len = m & 0xff, g = ml.group(p);
    uint32_t *hg = h[g];
    if (len == 0) {
      const uint32_t px = pix[gp];
      if (x.cache_bits && ((x.hitbits[gp >> 5] >> (gp & 31)) & 1u)) {
        atomicAdd(&gh[(size_t)g * kExtSyms + 280 + cache_key(px, x.cache_bits)], 1u);
      } else {
        atomicAdd(&hg[kGreen * kCodeStride + ((px >> 8) & 0xff)], 1u);
        atomicAdd(&hg[kRed * kCodeStride + ((px >> 16) & 0xff)], 1u);
        atomicAdd(&hg[kBlue * kCodeStride + (px & 0xff)], 1u);
        atomicAdd(&hg[kAlpha * kCodeStride + (px >> 24)], 1u);
      }
    } else {
      uint32_t sym, nb, eb;
      prefix_encode(len, sym, nb, eb);
      atomicAdd(&hg[kGreen * kCodeStride + 256 + sym], 1u);
      prefix_encode((m >> 19) & 0x7ff, sym, nb, eb);
      atomicAdd(&hg[kDist * kCodeStride + sym], 1u);
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < G * kNumCodes * kCodeStride; i += 256) {
    const uint32_t v = (&h[0][0])[i];
    if (!v) continue;
    const uint32_t g = i / (kNumCodes * kCodeStride), r = i % (kNumCodes * kCodeStride), c = r / kCodeStride, sym = r % kCodeStride;
    atomicAdd(&gh[(size_t)g * kExtSyms + ext_code_offset((int)c) + sym], v);
  }
}

// One workgroup (a wave) per image, group and code: the code from its histogram, its table entries, its description.
struct ExtCodeShared {
  HuffScratchBig huff;
  DescScratchBig desc;
  uint8_t len[kExtGreenMax];
};
__global__ __launch_bounds__(64) void k_ext_codes(const EncState *__restrict__ st, ExtArgs x) {
  __shared__ ExtCodeShared sh;
  __shared__ BitSink bw;
  __shared__ int s_used;
  const uint32_t img = blockIdx.x / (kExtMaxGroups * kNumCodes), g = (blockIdx.x / kNumCodes) % kExtMaxGroups, c = blockIdx.x % kNumCodes;
  const uint32_t lane = threadIdx.x;
  if (g >= st[img].ngroups) return;
  const int nsym = c == (uint32_t)kGreen ? kAlphabetGreen + (x.cache_bits ? (1 << x.cache_bits) : 0) : alphabet_size((int)c);
  const size_t grp = (size_t)img * kExtMaxGroups + g;
  uint32_t *h = x.histo + grp * kExtSyms + ext_code_offset((int)c);
  uint32_t *codes = x.codes + grp * kExtSyms + ext_code_offset((int)c);
  uint2 *tab = x.tab + grp * kExtTabStride + ext_code_offset((int)c);
  const int used0 = wave_prepare_leaves(h, nsym, sh.huff, lane);
  if (lane == 0) {
    const Dialect dl = dialect_of(x.standard != 0);
    int u = x.standard ? huffman_lengths_limited(h, nsym, dl.max_len, sh.len, sh.huff, used0) : huffman_lengths(h, nsym, dl.max_len, sh.len, sh.huff, used0);
    int rc = u < 0 ? u : 0;
    if (u >= 0) {
      canonical_reversed_codes(sh.len, nsym, codes, sh.huff.wt);
      bw.init(x.desc + (grp * kNumCodes + c) * kExtDescWords);
      rc = write_code_desc(bw, sh.len, nsym, sh.desc, x.standard != 0);
      bw.flush();
      x.descbits[grp * kNumCodes + c] = (uint32_t)bw.bits();
    }
    x.rc[grp * kNumCodes + c] = rc;
    s_used = u;
  }
  __syncthreads();
  const int used = s_used;
  // a code with at most one used symbol is not written in the pixel data (its reader takes no bits, huffman_coding.h:207-214)
  for (int i = (int)lane; i < nsym; i += 64) tab[i] = used > 1 ? make_uint2(codes[i], sh.len[i]) : make_uint2(0u, 0u);
  if (c == 0 && lane == 0) x.tab[grp * kExtTabStride + kExtZero] = make_uint2(0u, 0u);
}

// ---------------------------------------------------------------------------------------------
// Token bits with the position's group
// ---------------------------------------------------------------------------------------------
// The (at most four) parts of one token: table indices and the extra bits of a back reference.
struct ExtParts {
  uint32_t i0, i1, i2, i3;
  uint2 x1, x3;
};
__device__ __forceinline__ ExtParts ext_parts(uint32_t pixel, uint32_t m, bool hit, uint32_t cache_bits) {
  ExtParts t;
  t.x1 = t.x3 = make_uint2(0, 0);
  const uint32_t len = m & 0xff;
  if (len) {
    uint32_t sym, nb, eb;
    prefix_encode(len, sym, nb, eb);
    t.i0 = 256 + sym;
    t.x1 = make_uint2(eb, nb);
    prefix_encode((m >> 19) & 0x7ff, sym, nb, eb);
    t.i2 = kExtDist + sym;
    t.x3 = make_uint2(eb, nb);
    t.i1 = t.i3 = kExtZero;
  } else if (hit) {
    t.i0 = 280 + cache_key(pixel, cache_bits);
    t.i1 = t.i2 = t.i3 = kExtZero;
  } else {
    t.i0 = (pixel >> 8) & 0xff;
    t.i1 = kExtRed + ((pixel >> 16) & 0xff);
    t.i2 = kExtBlue + (pixel & 0xff);
    t.i3 = kExtAlpha + (pixel >> 24);
  }
  return t;
}

__global__ __launch_bounds__(256) void k_ext_bitcount(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                      const uint32_t *__restrict__ sb_first, uint32_t n,
                                                      const uint32_t *__restrict__ pix, const uint32_t *__restrict__ match,
                                                      const uint32_t *__restrict__ flags, uint16_t *__restrict__ group_bits,
                                                      uint32_t *__restrict__ unit_bits, ExtArgs x) {
  __shared__ uint8_t gmap[kSmallMax];
  __shared__ uint32_t wsum[4];
  const uint32_t unit = blockIdx.x, sb = unit / kUnitsPerSuper, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  const EncState &s = st[img];
  const uint32_t M = s.M;
  const uint32_t p0 = (unit - sb_first[img] * kUnitsPerSuper) * kUnit;
  if (p0 >= M) return;
  const MetaLookup ml = meta_lookup_load(s, x, img, gmap, tid, 256);
  __syncthreads();
  const uint64_t pos0 = imgs[img].pos0;
  const uint2 *tabs = x.tab + (size_t)img * kExtMaxGroups * kExtTabStride;
  uint32_t total = 0;
  for (uint32_t k = 0; k < kUnit / 1024; k++) {
    const uint32_t p = p0 + (k * 256 + tid) * 4;
    uint32_t bits = 0;
    if (p < M) {
      uint32_t fl = (flags[(pos0 + p) >> 5] >> ((pos0 + p) & 31)) & 15u;
      const uint32_t hits = x.cache_bits ? (x.hitbits[(pos0 + p) >> 5] >> ((pos0 + p) & 31)) & 15u : 0u;
      if (p + 4 > M) fl &= (1u << (M - p)) - 1u;
      for (uint32_t j = 0; j < 4; j++) {
        if (!((fl >> j) & 1u)) continue;
        const ExtParts t = ext_parts(pix[pos0 + p + j], match[pos0 + p + j], (hits >> j) & 1u, x.cache_bits);
        const uint2 *tab = tabs + (size_t)ml.group(p + j) * kExtTabStride;
        bits += tab[t.i0].y + tab[t.i1].y + tab[t.i2].y + tab[t.i3].y + t.x1.y + t.x3.y;
      }
    }
    total += bits;
    const uint32_t g = quad_sum_u32(bits);  // positions p-12 .. p+3: one group of 16
    if ((tid & 3) == 3 && (p - 12) < M) group_bits[(pos0 + p) >> 4] = (uint16_t)g;
  }
  const uint32_t sum = wave_sum_u32(total);
  if ((tid & 63) == 0) wsum[tid >> 6] = sum;
  __syncthreads();
  if (tid == 0) unit_bits[unit] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

template <class Put>
__device__ __forceinline__ void ext_emit_token(Put &put, const uint2 *tab, const ExtParts &t) {
  const uint2 e0 = tab[t.i0], t1 = tab[t.i1], e2 = tab[t.i2], t3 = tab[t.i3];
  const uint2 e1 = make_uint2(t1.x | t.x1.x, t1.y | t.x1.y), e3 = make_uint2(t3.x | t.x3.x, t3.y | t.x3.y);
  put(e0.x, e0.y); put(e1.x, e1.y); put(e2.x, e2.y); put(e3.x, e3.y);
}

template <class Put>
__device__ __forceinline__ void ext_emit_group(Put &put, const uint2 *tabs, const MetaLookup &ml, const uint32_t *__restrict__ px,
                                               const uint32_t *__restrict__ mt, uint32_t fl, uint32_t hits, uint32_t pb, uint32_t cache_bits) {
  for (uint32_t j = 0; j < 16; j++) {
    if (!((fl >> j) & 1u)) continue;
    const ExtParts t = ext_parts(px[j], mt[j], (hits >> j) & 1u, cache_bits);
    ext_emit_token(put, tabs + (size_t)ml.group(pb + j) * kExtTabStride, t);
  }
  put.finish();
}

__global__ __launch_bounds__(256) void k_ext_emit(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                  const uint32_t *__restrict__ sb_first, uint32_t n,
                                                  const uint32_t *__restrict__ pix, const uint32_t *__restrict__ match,
                                                  const uint32_t *__restrict__ flags, const uint16_t *__restrict__ group_bits,
                                                  const uint64_t *__restrict__ unit_off, uint8_t *__restrict__ out, ExtArgs x) {
  __shared__ uint8_t gmap[kSmallMax];
  __shared__ uint32_t stage[kWinWords + 2];
  __shared__ uint32_t wsum[4];
  const uint32_t sb = blockIdx.x, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = tid >> 6;
  const EncState &s = st[img];
  if (s.status != 0) return;
  const uint32_t M = s.M;
  if ((sb - sb_first[img]) * kSuper >= M) return;
  const MetaLookup ml = meta_lookup_load(s, x, img, gmap, tid, 256);
  const EncImage im = imgs[img];
  const uint64_t pos0 = im.pos0;
  const uint2 *tabs = x.tab + (size_t)img * kExtMaxGroups * kExtTabStride;
  uint32_t *out32 = reinterpret_cast<uint32_t *>(out + im.out_off);
  for (uint32_t u = 0; u < kUnitsPerSuper; u++) {
    const uint32_t unit = sb * kUnitsPerSuper + u;
    const uint32_t p0 = (sb - sb_first[img]) * kSuper + u * kUnit;
    if (p0 >= M) break;
    const uint32_t pb = p0 + tid * 16;
    const uint32_t *mt = match + pos0 + pb, *px = pix + pos0 + pb;
    uint32_t fl = 0, hits = 0, mybits = 0;
    if (pb < M) {
      fl = (flags[(pos0 + pb) >> 5] >> ((pos0 + pb) & 31)) & 0xffffu;
      if (x.cache_bits) hits = (x.hitbits[(pos0 + pb) >> 5] >> ((pos0 + pb) & 31)) & 0xffffu;
      if (pb + 16 > M) fl &= (1u << (M - pb)) - 1u;
      mybits = group_bits[(pos0 + pb) >> 4];
    }
    const uint32_t incl = wave_incl_scan_u32(mybits, (int)lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();  // also: the group map is in LDS, the previous unit's words are out
    uint32_t before = 0;
    for (uint32_t k = 0; k < wave; k++) before += wsum[k];
    const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    if (total == 0) continue;
    const uint64_t start_bit = unit_off[unit], end_bit = start_bit + total;
    const uint64_t my_bit = start_bit + before + incl - mybits;
    const uint64_t first_word = start_bit >> 5, last_word = (end_bit - 1) >> 5;
    const uint64_t nwords = last_word - first_word + 1;
    constexpr long long kWinBits = (long long)kWinWords * 32;
    for (uint64_t wbase = 0; wbase < nwords; wbase += kWinWords) {
      const uint64_t cntw = nwords - wbase < kWinWords ? nwords - wbase : kWinWords;
      for (uint32_t i = tid; i < cntw + 2; i += 256) stage[i] = 0;
      __syncthreads();
      const long long rel = (long long)my_bit - (long long)((first_word + wbase) << 5);
      if (mybits && rel + (long long)mybits > 0 && rel < kWinBits) {
        if (rel >= 0 && rel + (long long)mybits <= kWinBits) {
          RunPut put(stage, (uint32_t)rel);
          ext_emit_group(put, tabs, ml, px, mt, fl, hits, pb, x.cache_bits);
        } else {
          StagePut put{stage, rel};
          ext_emit_group(put, tabs, ml, px, mt, fl, hits, pb, x.cache_bits);
        }
      }
      __syncthreads();
      for (uint32_t i = tid; i < cntw; i += 256) {
        const uint64_t gw = first_word + wbase + i;
        const bool shared = (gw == first_word && (start_bit & 31)) || (gw == last_word && (end_bit & 31));
        if (shared) atomicOr(&out32[gw], stage[i]); else out32[gw] = stage[i];
      }
      __syncthreads();
    }
  }
}

}  // namespace vp8l

// vp8l_encode.hip -- MI355X (gfx950) kernels of the VP8L encoder and the batch encoder context.
//
// Pipeline for a batch of independent images, everything resident in HBM, no host round trip
// between stages:
//   k_palette_scan / k_palette_finalize   CreatePallete            vp8l.h:372-380
//   k_transform                           subtract green / colour indexing + bundling
//                                                                  vp8l.h:411-459, transform.cpp:88-106
//   k_pred_modes / k_pred_residual        optional forward predictor (inverse: transform.h:110-191)
//   k_lz77_sparse                         LZ77<uint32_t>::pack best match at every position, from candidate
//                                         lists (also makes the subtract-green residual when it can)
//   k_lz77 / k_lz77_head                  the same by testing all 1024 distances: fallback for tiles the
//                                         candidate lists give up on (_head: an image's first 1024 positions)
//                                                                                     lz77.h:51-72
//   k_parse_exits / k_parse_chain / k_parse_mark   the greedy parse (which positions start a token)
//                                                                  lz77.h:33-92
//                                         (k_parse_mark also counts the symbols: the five histograms, vp8l.h:516-535)
//   k_entropy_header                      Huffman codes + everything in front of the tokens
//                                                                  vp8l.h:460-467,536-540; huffman_io.h:331-410
//   k_bitcount / k_scan / k_emit          token bits              vp8l.h:545-583, bit_writer.h:45-64
#include <hip/hip_runtime.h>

#include <type_traits>

#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <string>
#include <vector>

#include "vp8l_amd.h"
#include "vp8l_dec_core.h"  // the predictors (predict, px_*): the forward transform must mirror the inverse
#include "vp8l_enc_core.h"
#include "vp8l_host.h"
#include "vp8l_kernels.h"

namespace vp8l {

// =============================================================================================
// Palette detection.  The reference inserts every pixel into a std::set and gives up as soon as
// the set holds more than 256 colours.  Here every block keeps an exact hash set of the colours
// of its 16384 pixels in LDS, gives up (for the whole image) once it has seen more than 256, and
// otherwise merges into the image's set in HBM.
// =============================================================================================
constexpr uint32_t kSetSlots = 1024;
constexpr uint64_t kEmptySlot = ~0ull;

__device__ __forceinline__ uint32_t hash_px(uint32_t v) { return (v * 0x9E3779B1u) >> 22; }

// Returns 1 if newly inserted, 0 if already present, 2 if the table is full.
template <class Slot>
__device__ __forceinline__ int set_insert(Slot *set, uint32_t v) {
  uint32_t h = hash_px(v);
  for (uint32_t probe = 0; probe < kSetSlots; probe++) {
    // A look first: a slot that holds a colour keeps it, so what is seen there holds, and the lanes of a wave that ask for one
    // colour of a small palette read one address at once instead of queueing for it (a slot seen empty is then claimed, or
    // found taken, by the compare-and-swap).
    unsigned long long old = *(volatile Slot *)&set[h];
    if (old == kEmptySlot) {
      old = atomicCAS((unsigned long long *)&set[h], (unsigned long long)kEmptySlot, (unsigned long long)v);
      if (old == kEmptySlot) return 1;
    }
    if ((uint32_t)old == v) return 0;
    h = (h + 1) & (kSetSlots - 1);
  }
  return 2;
}

// A look at the first pixels of every image, before the full scan: an image with more than 256 colours nearly always shows
// them at once, and then none of its blocks needs to start (all blocks of an image are dispatched together, so without
// this each of them would fill its own set until it overflows: 0.84 -> 0.2 ms per 256 photo-like 4K images).
constexpr uint32_t kProbePixels = 4096;
__global__ __launch_bounds__(256) void k_palette_probe(const EncImage *__restrict__ imgs, EncState *__restrict__ st) {
  __shared__ unsigned long long set[kSetSlots];
  __shared__ uint32_t s_count;
  const uint32_t img = blockIdx.x, tid = threadIdx.x;
  const EncImage im = imgs[img];
  const uint64_t npx = (uint64_t)im.w * im.h;
  for (uint32_t i = tid; i < kSetSlots; i += 256) set[i] = kEmptySlot;
  if (tid == 0) s_count = 0;
  __syncthreads();
  for (uint32_t it = 0; it < kProbePixels / 256; it++) {
    const uint64_t i = (uint64_t)it * 256 + tid;
    if (i < npx && set_insert(set, im.argb[i]) == 1) atomicAdd(&s_count, 1u);  // 1024 slots never fill up with 256 pixels a step
    __syncthreads();
    const bool too_many = s_count > kMaxPalette;
    __syncthreads();
    if (too_many) {
      if (tid == 0) st[img].pal_overflow = 1;
      return;
    }
    if ((uint64_t)(it + 1) * 256 >= npx) return;
  }
}

__global__ __launch_bounds__(256) void k_palette_scan(const EncImage *__restrict__ imgs, EncState *__restrict__ st,
                                                      const uint32_t *__restrict__ sb_first, uint32_t n,
                                                      unsigned long long *__restrict__ gsets) {
  __shared__ unsigned long long set[kSetSlots];
  __shared__ uint32_t s_count, s_over;
  const uint32_t sb = blockIdx.x, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  const EncImage im = imgs[img];
  const uint64_t npx = (uint64_t)im.w * im.h;
  const uint64_t base = (uint64_t)(sb - sb_first[img]) * kSuper;
  if (base >= npx) return;
  if (__hip_atomic_load(&st[img].pal_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  for (uint32_t i = tid; i < kSetSlots; i += 256) set[i] = kEmptySlot;
  if (tid == 0) { s_count = 0; s_over = 0; }
  __syncthreads();
  // A round is kScanRows rows of 1024 pixels, four neighbours per thread and row (one 16-byte load where the caller's pixels
  // are aligned so): enough bytes on their way per thread to keep the memory busy, and the next round's are loaded while this
  // round's are looked up.  A round ends with the look at the count (a round may bring more colours than the set holds: then the
  // set says so, and the count is past 256 as well).
  constexpr uint32_t kScanRows = 2, kRound = 1024 * kScanRows;
  const bool aligned = (reinterpret_cast<uintptr_t>(im.argb) & 15) == 0;
  auto load4 = [&](uint64_t i) -> uint4 {
    if (aligned && i + 4 <= npx) return *reinterpret_cast<const uint4 *>(im.argb + i);
    uint4 q = make_uint4(0, 0, 0, 0);
    if (i < npx) q.x = im.argb[i];
    if (i + 1 < npx) q.y = im.argb[i + 1];
    if (i + 2 < npx) q.z = im.argb[i + 2];
    if (i + 3 < npx) q.w = im.argb[i + 3];
    return q;
  };
  uint4 ahead[kScanRows];
#pragma unroll
  for (uint32_t j = 0; j < kScanRows; j++) ahead[j] = load4(base + j * 1024 + tid * 4);
  for (uint32_t it = 0; it < kSuper / kRound; it++) {
    const uint64_t i0 = base + (uint64_t)it * kRound + tid * 4;
    uint4 v[kScanRows];
#pragma unroll
    for (uint32_t j = 0; j < kScanRows; j++) v[j] = ahead[j];
    if (it + 1 < kSuper / kRound) {
#pragma unroll
      for (uint32_t j = 0; j < kScanRows; j++) ahead[j] = load4(i0 + kRound + j * 1024);
    }
#pragma unroll
    for (uint32_t j = 0; j < kScanRows; j++) {
      // a pixel equal to the one before it adds nothing: flat areas cost one insertion per run and wave
      const uint64_t i = i0 + j * 1024;
      const uint32_t four[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
      const uint32_t left = (uint32_t)__shfl_up((int)four[3], 1, 64);
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        const bool fresh = q ? four[q] != four[q - 1] : ((tid & 63) == 0 || four[0] != left);
        if (i + q < npx && fresh) {
          const int r = set_insert(set, four[q]);
          if (r == 1) atomicAdd(&s_count, 1u);
          if (r == 2) s_over = 1;
        }
      }
    }
    if (tid == 0 && (it & 1) == 1 &&
        __hip_atomic_load(&st[img].pal_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      s_over = 1;
    __syncthreads();
    const bool too_many = s_count > kMaxPalette;
    const bool stop = too_many || s_over;
    __syncthreads();
    if (stop) {
      if (tid == 0 && too_many) __hip_atomic_store(&st[img].pal_overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (base + (uint64_t)(it + 1) * kRound >= npx) break;
  }
  unsigned long long *gset = gsets + (size_t)img * kSetSlots;
  for (uint32_t s = tid; s < kSetSlots; s += 256) {
    const unsigned long long e = set[s];
    if (e == kEmptySlot) continue;
    if (__hip_atomic_load(&st[img].pal_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    const int r = set_insert(gset, (uint32_t)e);
    if (r == 2 || (r == 1 && atomicAdd(&st[img].pal_count, 1u) + 1 > kMaxPalette))
      __hip_atomic_store(&st[img].pal_overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(256) void k_palette_finalize(const EncImage *__restrict__ imgs, EncState *__restrict__ st,
                                                          const unsigned long long *__restrict__ gsets, uint32_t fuse, uint32_t pred_bits,
                                                          uint32_t cross_bits, uint32_t meta_bits) {
  __shared__ uint32_t list[kSetSlots];
  __shared__ uint32_t s_k;
  const uint32_t img = blockIdx.x, tid = threadIdx.x;
  const EncImage im = imgs[img];
  EncState &s = st[img];
  if (tid == 0) {
    s_k = 0;
    s.ngroups = 1;
  }
  // the entropy image's tiles: over the image's width and height in the reference's dialect (vp8l.h:195-196), over the coded
  // (bundled) width in the standard one; the tile size grows until the header coder takes the image
  auto meta_grid = [&](uint32_t grid_w) {
    if (!meta_bits) return;
    uint32_t b = meta_bits & 15u;
    while (b < 9 && ((grid_w + (1u << b) - 1) >> b) * ((im.h + (1u << b) - 1) >> b) > kSmallMax) b++;
    s.meta_bits = b;
    s.meta_tx = (grid_w + (1u << b) - 1) >> b;
    s.meta_ty = (im.h + (1u << b) - 1) >> b;
  };
  __syncthreads();
  const bool none = s.pal_overflow || s.pal_count > kMaxPalette || s.pal_count == 0;
  if (none) {
    if (tid == 0) {
      s.ncolors = 0; s.bits = 0; s.xsize = im.w; s.M = im.w * im.h; s.xinv = dist_inverse(im.w);
      meta_grid(im.w);
      // the candidate-list kernel reads 16 bytes at a time straight from the caller's pixels
      s.fused = fuse && (reinterpret_cast<uintptr_t>(im.argb) & 15) == 0;
      if (pred_bits) {  // the smallest tile size >= the requested one whose mode image fits the header coder
        uint32_t b = pred_bits;
        while (b < 9 && ((im.w + (1u << b) - 1) >> b) * ((im.h + (1u << b) - 1) >> b) > kSmallMax) b++;
        s.pred_bits = b;
        s.pred_tx = (im.w + (1u << b) - 1) >> b;
        s.pred_ty = (im.h + (1u << b) - 1) >> b;
      }
      if (cross_bits) {
        uint32_t b = cross_bits;
        while (b < 9 && ((im.w + (1u << b) - 1) >> b) * ((im.h + (1u << b) - 1) >> b) > kSmallMax) b++;
        s.cross_bits = b;
        s.cross_tx = (im.w + (1u << b) - 1) >> b;
        s.cross_ty = (im.h + (1u << b) - 1) >> b;
      }
    }
    return;
  }
  const unsigned long long *gset = gsets + (size_t)img * kSetSlots;
  for (uint32_t i = tid; i < kSetSlots; i += 256) {
    const unsigned long long e = gset[i];
    if (e != kEmptySlot) list[atomicAdd(&s_k, 1u)] = (uint32_t)e;
  }
  __syncthreads();
  const uint32_t k = s_k;
  if (tid < k) {  // ascending order, as iterating the reference's std::set gives (vp8l.h:604-605)
    const uint32_t v = list[tid];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < k; j++) rank += list[j] < v;
    s.palette[rank] = v;
  }
  if (tid == 0) {
    const uint32_t bits = palette_bits(k);
    s.ncolors = k;
    s.bits = bits;
    s.xsize = (im.w + (1u << bits) - 1) >> bits;
    s.M = s.xsize * im.h;
    s.xinv = dist_inverse(s.xsize);
    meta_grid((meta_bits & 16u) ? s.xsize : im.w);  // bit 4: the standard dialect
  }
}

// =============================================================================================
// Forward transform: one coded pixel per thread-iteration, 8 B of HBM traffic per pixel on the
// subtract-green path, 4 + 4/2^bits on the colour-indexing path.
// =============================================================================================
// Colour indexing of one superblock of coded pixels, 2^BITS pixels to a coded one (vp8l.h:438-443, 598-611).  A few coded pixels
// per thread at a time, all their pixels loaded before the first is looked up (clamped addresses, so that no load waits for a
// test): one pixel after the other, the kernel waited for memory 64 times per thread.
template <uint32_t BITS>
__device__ __forceinline__ void index_superblock(const uint32_t *__restrict__ argb, uint32_t w, uint32_t xsize, uint32_t M, uint32_t p0,
                                                 uint32_t tid, const unsigned long long *pal_index, uint32_t *__restrict__ dst) {
  constexpr uint32_t kPer = 1u << BITS, kDepth = 8u >> BITS, kAtOnce = kPer >= 4 ? 2 : 4;
#pragma unroll 1
  for (uint32_t k = 0; k < kSuper / 256 && p0 + k * 256 < M; k += kAtOnce) {
    uint32_t v[kAtOnce][kPer];
#pragma unroll
    for (uint32_t u = 0; u < kAtOnce; u++) {
      const uint32_t p = p0 + (k + u) * 256 + tid, pc = p < M ? p : M - 1;
      const uint32_t y = pc / xsize, x0 = (pc - y * xsize) << BITS;
#pragma unroll
      for (uint32_t j = 0; j < kPer; j++) v[u][j] = argb[(uint64_t)y * w + (x0 + j < w ? x0 + j : w - 1)];  // (past the row's end: not used)
    }
#pragma unroll
    for (uint32_t u = 0; u < kAtOnce; u++) {
      const uint32_t p = p0 + (k + u) * 256 + tid, pc = p < M ? p : M - 1;
      const uint32_t x0 = (pc % xsize) << BITS;
      uint32_t code = 0xff000000u;
#pragma unroll
      for (uint32_t j = 0; j < kPer; j++) {
        uint32_t h = hash_px(v[u][j]);
        unsigned long long e = pal_index[h];
        for (uint32_t probe = 0; (uint32_t)e != v[u][j] && probe < kSetSlots; probe++) {  // (every pixel's colour is in the palette)
          h = (h + 1) & (kSetSlots - 1);
          e = pal_index[h];
        }
        if (x0 + j < w) code |= (uint32_t)(e >> 32) << (8 + kDepth * j);
      }
      if (p < M) dst[p] = code;
    }
  }
}

__global__ __launch_bounds__(256) void k_transform(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                   const uint32_t *__restrict__ sb_first, uint32_t n,
                                                   uint32_t *__restrict__ pix) {
  // a superblock per workgroup: most calls have nothing for this kernel to do (the match kernel makes the residual of
  // aligned non-palette images itself), and workgroups that only look and leave cost what their launch costs
  __shared__ unsigned long long pal_index[kSetSlots];  // index << 32 | colour
  const uint32_t sb = blockIdx.x, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  const EncImage im = imgs[img];
  const uint32_t M = st[img].M, ncolors = st[img].ncolors;
  const uint32_t p0 = (sb - sb_first[img]) * kSuper;
  if (p0 >= M || st[img].fused) return;
  if (ncolors == 0 && st[img].pred_bits) return;  // the predictor's kernels read the caller's pixels and subtract green themselves
  uint32_t *dst = pix + im.pos0;
  if (ncolors == 0) {
#pragma unroll 4
    for (uint32_t k = 0; k < kSuper / 256; k++) {
      const uint32_t p = p0 + k * 256 + tid;
      if (p < M) dst[p] = subtract_green(im.argb[p]);
    }
    return;
  }
  // index of a colour in the sorted palette (vp8l.h:438-443; the colours are distinct): from a table on the colour's hash, one or
  // two reads instead of the eight dependent ones of a search through 256 colours
  for (uint32_t i = tid; i < kSetSlots; i += 256) pal_index[i] = kEmptySlot;
  __syncthreads();
  if (tid < ncolors) {
    const uint32_t v = st[img].palette[tid];
    uint32_t h = hash_px(v);
    while (atomicCAS(&pal_index[h], (unsigned long long)kEmptySlot, ((unsigned long long)tid << 32) | v) != kEmptySlot) h = (h + 1) & (kSetSlots - 1);
  }
  __syncthreads();
  const uint32_t xsize = st[img].xsize;
  switch (st[img].bits) {
    case 0: index_superblock<0>(im.argb, im.w, xsize, M, p0, tid, pal_index, dst); break;
    case 1: index_superblock<1>(im.argb, im.w, xsize, M, p0, tid, pal_index, dst); break;
    case 2: index_superblock<2>(im.argb, im.w, xsize, M, p0, tid, pal_index, dst); break;
    default: index_superblock<3>(im.argb, im.w, xsize, M, p0, tid, pal_index, dst); break;
  }
}

// =============================================================================================
// LZ77 best match at EVERY position (lz77.h:51-72): over distances d = 1..min(p,1024),
// len(d) = min(run of data[q-d]==data[q] from q=p, d, 128, M-p); the longest wins, the largest d
// among equals.  The parse later picks the positions it actually visits.
//
// Formulation: for a fixed distance the equality run length at p is known from the run at p+1,
// so positions are scanned BACKWARDS and every distance keeps one register: the first position
// >= p where data[q-d] != data[q].  One wave owns a tile of kLzTile positions; lane l owns the
// 16 distances 16l+1..16l+16, which at position p need the 16 consecutive pixels
// data[p-16l-16 .. p-16l-1]: a window that slides by one pixel per step, held in registers and
// refilled with one 64-byte load per 16 steps.  Cost is 1024 cell updates per position
// whatever the image content.  Positions are handled in aligned chunks of 16 and coordinates are
// kept relative to the current group of four positions (see lz77_chunk) so every per-step
// constant is an inline constant.
// =============================================================================================
// max over the 64 lanes with DPP lane moves fused into the max (no LDS round trips); the result
// is read from lane 63 into an SGPR.  Each step reads the register the previous VALU instruction
// wrote, which needs two wait states in front of a DPP operand.
#define VP8L_DPP_MAX(v, ctrl) asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 " ctrl : "+v"(v))
__device__ __forceinline__ uint32_t wave_max_dpp(uint32_t v) {
  VP8L_DPP_MAX(v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
  VP8L_DPP_MAX(v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
  VP8L_DPP_MAX(v, "row_half_mirror row_mask:0xf bank_mask:0xf");
  VP8L_DPP_MAX(v, "row_mirror row_mask:0xf bank_mask:0xf");   // every lane of a 16-lane row holds the row's max
  VP8L_DPP_MAX(v, "row_bcast:15 row_mask:0xa bank_mask:0xf");  // rows 1 and 3 also take the row before
  VP8L_DPP_MAX(v, "row_bcast:31 row_mask:0xc bank_mask:0xf");  // rows 2 and 3 take lane 31: lane 63 = wave max
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ void load16(const uint32_t *__restrict__ src, uint32_t (&L)[16]) {
  const uint4 *q = reinterpret_cast<const uint4 *>(src);
  const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
  L[0] = v0.x; L[1] = v0.y; L[2] = v0.z; L[3] = v0.w; L[4] = v1.x; L[5] = v1.y; L[6] = v1.z; L[7] = v1.w;
  L[8] = v2.x; L[9] = v2.y; L[10] = v2.z; L[11] = v2.w; L[12] = v3.x; L[13] = v3.y; L[14] = v3.z; L[15] = v3.w;
}

__device__ __forceinline__ uint32_t min3_u32(uint32_t a, uint32_t b, uint32_t c) { a = a < b ? a : b; return a < c ? a : c; }
__device__ __forceinline__ uint32_t max3_u32(uint32_t a, uint32_t b, uint32_t c) { a = a > b ? a : b; return a > c ? a : c; }

// One chunk of 16 positions P, P-1, .., P-15 (P = 15 mod 16).  EMIT: the positions belong to the
// tile (best matches are produced); otherwise they only warm the run state up.
// One "slot" = the 64 cells (lanes) of one j.  A cell's state is (first mismatch << 4 | j), so the largest state of a
// lane names its cell.  Positions are counted within groups of four (3, 2, 1, 0 from the top; every state moves up by 4
// after a group), which keeps the value a cell takes on a mismatch, position << 4 | j, below 64: an inline constant of
// the select instruction, which can read neither a literal nor a scalar register beside its mask.
// A position's best match is found in two steps.  First the largest state of the wave, without the limits of a match
// (it cannot overlap its source: at most its distance; at most 128 pixels): three instructions per cell less than
// limiting every cell.  If the limits do not touch the winning cell, it is the best match: no other cell's limited
// value exceeds its own state, so none exceeds the winner's, and an equal one is nearer by the packing.  Otherwise the
// position is scored again with the limits applied to every cell, and counted in `rescored`.
// CAP: every state is first cut to 128 pixels above the position (one more instruction per cell).  Cut states stay
// valid, the limit only falls from one position to the next below it.  With it, matches of 128 pixels and more (cells
// above the limit by different amounts, where the farthest must win) need no second scoring; the caller switches it
// on per chunk where such matches are.
template <bool HEAD, bool EMIT, bool CAP>
__device__ __forceinline__ uint32_t lz77_chunk(uint32_t (&mis)[16], const uint32_t (&win)[31], const uint32_t xcur, const int P,
                                               const uint32_t M, const int lane, uint32_t &rescored) {
  uint32_t res = 0;  // lane s: packed (end of best match << 10 | distance-1) of position P-s
#pragma unroll
  for (int s = 0; s < 16; s++) {
    const uint32_t p = (uint32_t)(P - s);
    const uint32_t here = 3u - ((uint32_t)s & 3u);
    if (p < M) {  // else past the end of the image (last chunk only)
      const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)xcur, 15 - s);  // data[p], the same for all lanes
#pragma unroll
      for (int j = 0; j < 16; j++) {
        bool eq = win[s + j] == x;
        // distance 16 * lane + j + 1 <= p: those reaching before the first pixel never match.  (The right side is kept a
        // scalar on purpose: moved to the left, the compiler holds 31 sums in vector registers for the whole kernel.)
        if (HEAD) eq = eq && 16 * lane + 1 <= __builtin_amdgcn_readfirstlane((int)p - j);
        uint32_t m = mis[j];
        if (EMIT && CAP) m = m < (((here + kMaxLen) << 4) | (uint32_t)j) ? m : (((here + kMaxLen) << 4) | (uint32_t)j);
        mis[j] = eq ? m : ((here << 4) | (uint32_t)j);
      }
      if (EMIT) {
        // (end of match, distance-1): the largest distance wins among equal lengths (lz77.h:68).
        // No equal pixel anywhere in the window: every state is `here`, length 0.
        uint32_t m = max3_u32(mis[0], mis[1], mis[2]);
        m = max3_u32(m, mis[3], mis[4]);
        m = max3_u32(m, mis[5], mis[6]);
        m = max3_u32(m, mis[7], mis[8]);
        m = max3_u32(m, mis[9], mis[10]);
        m = max3_u32(m, mis[11], mis[12]);
        m = max3_u32(m, mis[13], mis[14]);
        m = m > mis[15] ? m : mis[15];
        uint32_t K = wave_max_dpp(((m & ~15u) << 6) | ((uint32_t)lane << 4) | (m & 15u));
        const uint32_t dist = (K & 1023u) + 1u;
        if ((K >> 10) > here + (dist < kMaxLen ? dist : kMaxLen)) {  // the limits cut the winner: apply them to every cell
          rescored++;
          const uint32_t lim = ((here + kMaxLen) << 4) | 15u;
          const uint32_t cap0 = ((16u * (uint32_t)lane + 1u + here) << 4) | 15u;  // position + distance of cell 0
          uint32_t key = 0;
#pragma unroll
          for (int j = 0; j < 16; j++) {
            const uint32_t t = min3_u32(mis[j], cap0 + ((uint32_t)j << 4), lim);
            const uint32_t kj = (t & ~15u) | (uint32_t)j;
            key = key > kj ? key : kj;
          }
          K = wave_max_dpp(((key & ~15u) << 6) | ((uint32_t)lane << 4) | (key & 15u));
        }
        res = lane == s ? K : res;
      }
    }
    if ((s & 3) == 3) {
#pragma unroll
      for (int j = 0; j < 16; j++) mis[j] += 4u << 4;
    }
  }
  return res;
}

// a, b, M are wave-uniform (scalar registers).
template <bool HEAD>
__device__ __forceinline__ void lz77_tile(const uint32_t *__restrict__ data, uint32_t *__restrict__ match,
                                          uint8_t *__restrict__ adv, const uint32_t M, const uint32_t a, const uint32_t b,
                                          const int lane, const uint32_t xsize, const uint32_t xinv) {
  const uint32_t qs = (b + 127 < M - 1) ? b + 127 : M - 1;
  const int ctop = (int)(qs >> 4);
  const int cbot = (int)(a >> 4);
  int P = ctop * 16 + 15;
  uint32_t mis[16];  // per distance: first mismatch position at or after the current one (see lz77_chunk) << 4 | j
  {
    const uint32_t lim = ((uint32_t)P + 1 < M ? (uint32_t)P + 1 : M) - (uint32_t)P + 3;
#pragma unroll
    for (int j = 0; j < 16; j++) mis[j] = (lim << 4) | (uint32_t)j;
  }
  // Whether the chunks cut their states to 128 pixels (see lz77_chunk): not while few of a chunk's positions have to be
  // scored again, and for the next eight chunks after a chunk where three or more had to.
  uint32_t cap_chunks = 0;
  uint32_t win[31];  // win[k] = data[P - 16*lane - 1 - k]
  uint32_t nxt[16];  // the 16 pixels that enter the window with the next chunk (loaded one chunk ahead)
  uint32_t xnext;    // lane k (< 16): pixel P-15+k of the next chunk
#pragma unroll
  for (int i = 0; i < 16; i++) nxt[i] = 0;
  {
    const int a1 = P - 16 * lane - 15;  // 16 aligned pixels, the last one (distance 0) unused
    uint32_t L[16];
#pragma unroll
    for (int i = 0; i < 16; i++) L[i] = 0;
    if (!HEAD || a1 >= 0) load16(data + a1, L);
#pragma unroll
    for (int i = 0; i < 15; i++) win[14 - i] = L[i];
    const int a0 = P - 16 * lane - 31;
    if (!HEAD || a0 >= 0) load16(data + a0, nxt);
    xnext = data[P - 15 + (lane & 15)];
  }
  for (int ck = ctop; ck >= cbot; ck--, P -= 16) {
#pragma unroll
    for (int i = 0; i < 16; i++) win[30 - i] = nxt[i];
    const uint32_t xcur = xnext;
    if (ck > cbot) {  // prefetch for the chunk below
      const int a0 = P - 16 - 16 * lane - 31;
      if (!HEAD || a0 >= 0) load16(data + a0, nxt);
      xnext = data[P - 31 + (lane & 15)];
    }
    if ((uint32_t)(P - 15) < b) {
      uint32_t rescored = 0, res;
      if (cap_chunks) {
        cap_chunks--;
        res = lz77_chunk<HEAD, true, true>(mis, win, xcur, P, M, lane, rescored);
      } else {
        res = lz77_chunk<HEAD, true, false>(mis, win, xcur, P, M, lane, rescored);
        if (rescored >= 3) cap_chunks = 8;
      }
      if (lane < 16) {
        const uint32_t p = (uint32_t)(P - lane);
        if (p < b && p < M) {
          uint32_t len = (res >> 10) - (3u - ((uint32_t)lane & 3u));
          const uint32_t dist = (res & 1023u) + 1u;
          if (p == M - 1) len = 0;  // a lone look-ahead pixel is always a literal (lz77.h:42-49)
          match[p] = len ? (len | (dist << 8) | (dist_to_code_inv(xsize, xinv, dist) << 19)) : 0u;
          adv[p] = (uint8_t)(len ? len : 1u);
        }
      }
    } else {
      uint32_t unused = 0;
      (void)lz77_chunk<HEAD, false, false>(mis, win, xcur, P, M, lane, unused);
    }
#pragma unroll
    for (int k = 0; k < 15; k++) win[k] = win[k + 16];
  }
}

// =============================================================================================
// Forward predictor transform (optional; the reference encoder has none, its decoder reads it:
// ReadTransform vp8l.h:111-121, inverse transform.h:110-191).  Works on the subtract-green image.
// The residual of a pixel needs only ORIGINAL neighbours, so it is one thread per pixel; the mode of
// a tile is the one with the smallest rough bit cost of the residual bytes (taken as signed), estimated
// on every second row and column of the tile.  Border rules as in the inverse: (0,0) predicts
// 0xff000000, the first row predicts L, the first column T, and TR of the last column is L.
// =============================================================================================
__device__ __forceinline__ uint32_t abs_cost(uint32_t r) {  // rough bit cost of the four residual bytes
  // Per byte, taken as a signed value v: nothing for zero, else 1 + 2 * (bits of |v|).  All four bytes at once (this is most of
  // what the choice of a tile's mode costs, 14 times a sample): |v| per byte, its bits smeared to the right so that a byte
  // holds as many ones as |v| has bits, and then the sum over the bytes is a population count.
  const uint32_t neg = (r >> 7) & 0x01010101u;
  uint32_t a = (r ^ (neg * 0xffu)) + neg;  // 256 - b where b >= 128 (no byte carries: at most 0x7f + 1)
  a |= (a >> 1) & 0x7f7f7f7fu;
  a |= (a >> 2) & 0x3f3f3f3fu;
  a |= (a >> 4) & 0x0f0f0f0fu;
  return (uint32_t)__popc(a & 0x01010101u) + 2u * (uint32_t)__popc(a);
}

__global__ __launch_bounds__(256) void k_pred_modes(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                    uint32_t *__restrict__ modes, uint32_t standard) {
  __shared__ uint32_t wcost[4][14];
  const uint32_t img = blockIdx.x / kSmallMax, tile = blockIdx.x % kSmallMax, tid = threadIdx.x;
  const EncState &s = st[img];
  const uint32_t bits = s.pred_bits;
  if (bits == 0 || s.ncolors || tile >= s.pred_tx * s.pred_ty) return;
  const uint32_t w = imgs[img].w, h = imgs[img].h;
  const uint32_t *src = imgs[img].argb;  // green is subtracted on the way in: k_transform leaves predictor images alone
  const uint32_t tx = tile % s.pred_tx, ty = tile / s.pred_tx, side = 1u << bits;
  const uint32_t x0 = tx << bits, y0 = ty << bits;
  uint32_t cost[14];
#pragma unroll
  for (int m = 0; m < 14; m++) cost[m] = 0;
  const uint32_t half = side >> 1;  // samples per tile side
  for (uint32_t k = tid; k < half * half; k += 256) {
    const uint32_t x = x0 + 2 * (k % half) + 1, y = y0 + 2 * (k / half) + 1;  // odd offsets: never the first row or column
    if (x >= w || y >= h) continue;
    const uint64_t i = (uint64_t)y * w + x;
    const uint32_t cur = subtract_green(src[i]), L = subtract_green(src[i - 1]), T = subtract_green(src[i - w]), TL = subtract_green(src[i - w - 1]);
    const uint32_t TR = x == w - 1 && !standard ? L : subtract_green(src[i - w + 1]);  // standard dialect: the next pixel in memory (first of the current row)
#pragma unroll
    for (int m = 0; m < 14; m++) cost[m] += abs_cost(px_sub(cur, predict((uint32_t)m, L, T, TR, TL)));
  }
#pragma unroll
  for (int m = 0; m < 14; m++) {
    const uint32_t v = wave_sum_u32(cost[m]);
    if ((tid & 63) == 0) wcost[tid >> 6][m] = v;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t best = 0, bestc = 0xffffffffu;
    for (uint32_t m = 0; m < 14; m++) {
      const uint32_t c = wcost[0][m] + wcost[1][m] + wcost[2][m] + wcost[3][m];
      if (c < bestc) { bestc = c; best = m; }
    }
    modes[(size_t)img * kSmallMax + tile] = 0xff000000u | (best << 8);  // the mode sits in the green byte (transform.h:181)
  }
}

// dst = residual (predictor images, made from the caller's pixels: green is subtracted on the way in) or a copy of the
// transformed pixels (all others), so that later stages read one array
__global__ __launch_bounds__(256) void k_pred_residual(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                       const uint32_t *__restrict__ sb_first, uint32_t n,
                                                       const uint32_t *__restrict__ pix, const uint32_t *__restrict__ modes,
                                                       uint32_t *__restrict__ dst, uint32_t standard) {
  const uint32_t unit = blockIdx.x, sb = unit / kUnitsPerSuper, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  const EncState &s = st[img];
  const uint32_t M = s.M;
  const uint32_t p0 = (unit - sb_first[img] * kUnitsPerSuper) * kUnit;
  if (p0 >= M) return;
  const uint64_t pos0 = imgs[img].pos0;
  uint32_t *out = dst + pos0;
  const uint32_t bits = s.ncolors ? 0u : s.pred_bits, w = imgs[img].w;
  if (bits == 0) {
    const uint32_t *src = pix + pos0;
    for (uint32_t k = 0; k < kUnit / 256; k++) {
      const uint32_t p = p0 + k * 256 + tid;
      if (p < M) out[p] = src[p];
    }
    return;
  }
  const uint32_t *src = imgs[img].argb;
  const uint32_t *mimg = modes + (size_t)img * kSmallMax;
  const uint32_t tiles_x = s.pred_tx;
  auto mode_at = [&](uint32_t x, uint32_t y) { return (mimg[(y >> bits) * tiles_x + (x >> bits)] >> 8) & 0xff; };
  // one pixel, whatever its place (transform.h:110-191: (0,0) predicts 0xff000000, the first row L, the first column T)
  auto one = [&](uint32_t p, uint32_t x, uint32_t y) -> uint32_t {
    const uint32_t cur = subtract_green(src[p]);
    uint32_t P;
    if (y == 0) P = x == 0 ? 0xff000000u : subtract_green(src[p - 1]);
    else if (x == 0) P = subtract_green(src[p - w]);
    else {
      const uint32_t L = subtract_green(src[p - 1]);
      P = predict(mode_at(x, y), L, subtract_green(src[p - w]), x == w - 1 && !standard ? L : subtract_green(src[p - w + 1]), subtract_green(src[p - w - 1]));
    }
    return px_sub(cur, P);
  };
  // Four neighbouring pixels a thread: inside a row (not its first or last pixels, not the first row) they share their
  // neighbours, 11 pixels read and turned instead of 20, and are stored as one 16-byte word (the region is aligned so).
  // Row and column are divided out once per thread, then moved on by 1024 pixels a step.
  uint32_t y = (p0 + tid * 4) / w, x = (p0 + tid * 4) - y * w;
  for (uint32_t k = 0; k < kUnit / 1024; k++) {
    const uint32_t p = p0 + k * 1024 + tid * 4;
    if (p >= M) break;
    uint32_t r[4];
    if (y > 0 && x > 0 && x + 4 < w && p + 4 <= M) {
      uint32_t c[5], t[6];  // c[0] = L, c[1..4] the pixels; t[0] = TL of the first, t[1..4] above the pixels, t[5] = TR of the last
#pragma unroll
      for (int j = 0; j < 5; j++) c[j] = subtract_green(src[p - 1 + j]);
#pragma unroll
      for (int j = 0; j < 6; j++) t[j] = subtract_green(src[p - w - 1 + j]);
      const uint32_t m_first = mode_at(x, y), m_last = mode_at(x + 3, y);
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        const uint32_t mode = ((x + j) >> bits) == (x >> bits) ? m_first : m_last;
        r[j] = px_sub(c[j + 1], predict(mode, c[j], t[j + 1], t[j + 2], t[j]));
      }
    } else {
      uint32_t xx = x, yy = y;
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        r[j] = p + j < M ? one(p + j, xx, yy) : 0u;
        if (++xx == w) { xx = 0; yy++; }
      }
    }
    *reinterpret_cast<uint4 *>(out + p) = make_uint4(r[0], r[1], r[2], r[3]);  // (the region has slack past M)
    x += 1024;
    if (w >= 1024) {
      if (x >= w) { x -= w; y++; }
    } else {
      y += x / w;
      x %= w;
    }
  }
}

// =============================================================================================
// Forward cross-colour transform (optional, like the predictor: the reference's decoder reads it, ReadTransform
// vp8l.h:111-121, inverse transform.h:192-218).  Per tile three signed multipliers: red loses green * g2r / 32, blue
// loses green * g2b / 32 and (original) red * r2b / 32.  A multiplier is found by a halving search (steps 32 ... 1 around
// the best so far, from 0) on a rough bit cost of the residual bytes over up to 32 x 32 samples of the tile, and used only
// if it takes an eighth off that sum, so tiles without correlation keep zeros, which cost next to nothing in the tile image.
// An element is 0xff000000 | r2b << 16 | g2b << 8 | g2r (transform.h:29-34).
// =============================================================================================
__device__ __forceinline__ int32_t color_delta(int32_t t, int32_t c) { return (int32_t)(int8_t)((t * c) >> 5); }  // transform.h:192-195
// What a residual byte costs, roughly: nothing for zero, else 1 + 2 * (bits of |v|), v taken modulo 256 as a signed byte.
// (The plain sum of |v| prefers many small non-zero residuals to a few larger ones; the entropy coder does not.)
__device__ __forceinline__ uint32_t abs_byte(int32_t v) {
  const int32_t b = (int32_t)(int8_t)v;
  const uint32_t a = (uint32_t)(b < 0 ? -b : b);
  return a ? 1u + 2u * (32u - (uint32_t)__clz((int)a)) : 0u;
}

constexpr uint32_t kCrossPer = 32;  // up to 32 x 32 samples of a tile
__global__ __launch_bounds__(64) void k_cross_elems(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                    const uint32_t *__restrict__ pix, uint32_t *__restrict__ elems) {
  const uint32_t img = blockIdx.x / kSmallMax, tile = blockIdx.x % kSmallMax, lane = threadIdx.x;
  const EncState &s = st[img];
  const uint32_t bits = s.cross_bits;
  if (bits == 0 || s.ncolors || tile >= s.cross_tx * s.cross_ty) return;
  const uint32_t w = imgs[img].w, h = imgs[img].h;
  const uint32_t *src = pix + imgs[img].pos0;
  const uint32_t x0 = (tile % s.cross_tx) << bits, y0 = (tile / s.cross_tx) << bits, side = 1u << bits;
  const uint32_t per = side < kCrossPer ? side : kCrossPer, step = side / per;  // samples per tile side and their spacing
  constexpr uint32_t kMine = kCrossPer * kCrossPer / 64;
  int32_t r[kMine], g[kMine], b[kMine];
#pragma unroll
  for (uint32_t k = 0; k < kMine; k++) {
    const uint32_t i = k * 64 + lane, x = x0 + (i % per) * step, y = y0 + (i / per) * step;
    r[k] = g[k] = b[k] = 0;  // a sample outside the tile or the image costs nothing whatever the multiplier
    if (i < per * per && x < w && y < h) {
      const uint32_t p = src[(uint64_t)y * w + x];
      r[k] = (int32_t)(int8_t)(p >> 16);
      g[k] = (int32_t)(int8_t)(p >> 8);
      b[k] = (int32_t)(int8_t)p;
    }
  }
  // cost of multiplier t for residual = base - delta(t, factor), summed over the tile's samples
  auto search = [&](const int32_t *base, const int32_t *factor) -> int32_t {
    auto cost = [&](int32_t t) {
      uint32_t c = 0;
#pragma unroll
      for (uint32_t k = 0; k < kMine; k++) c += abs_byte(base[k] - color_delta(t, factor[k]));
      return wave_sum_u32(c);
    };
    int32_t best = 0;
    const uint32_t c0 = cost(0);
    uint32_t bestc = c0;
    for (int32_t stepv = 32; stepv >= 1; stepv >>= 1) {
      const int32_t lo = best - stepv < -128 ? -128 : best - stepv, hi = best + stepv > 127 ? 127 : best + stepv;
      const uint32_t cl = cost(lo), ch = cost(hi);
      if (cl < bestc && cl <= ch) { bestc = cl; best = lo; }
      else if (ch < bestc) { bestc = ch; best = hi; }
    }
    // The samples are a part of a large tile and the cost is not the entropy: a multiplier has to take an eighth off
    // the sampled cost (and more than rounding noise) to be used, or it only stirs residuals that were already small.
    return bestc + (c0 >> 3) + 32 <= c0 ? best : 0;
  };
  const int32_t g2r = search(r, g);
  const int32_t g2b = search(b, g);
  int32_t b2[kMine];
#pragma unroll
  for (uint32_t k = 0; k < kMine; k++) b2[k] = b[k] - color_delta(g2b, g[k]);
  const int32_t r2b = search(b2, r);
  if (lane == 0)
    elems[(size_t)img * kSmallMax + tile] = 0xff000000u | ((uint32_t)(r2b & 0xff) << 16) | ((uint32_t)(g2b & 0xff) << 8) | (uint32_t)(g2r & 0xff);
}

// dst = src with the tiles' multipliers applied (cross-colour images) or a copy (all others); dst may be src
__global__ __launch_bounds__(256) void k_cross_apply(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                     const uint32_t *__restrict__ sb_first, uint32_t n, const uint32_t *src_all,
                                                     const uint32_t *__restrict__ elems, uint32_t *dst_all) {
  const uint32_t unit = blockIdx.x, sb = unit / kUnitsPerSuper, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  const EncState &s = st[img];
  const uint32_t M = s.M;
  const uint32_t p0 = (unit - sb_first[img] * kUnitsPerSuper) * kUnit;
  if (p0 >= M) return;
  const uint64_t pos0 = imgs[img].pos0;
  const uint32_t *src = src_all + pos0;
  uint32_t *out = dst_all + pos0;
  const uint32_t bits = s.ncolors ? 0u : s.cross_bits, w = imgs[img].w;
  if (bits == 0 && src == out) return;
  const uint32_t *eimg = elems + (size_t)img * kSmallMax;
  for (uint32_t k = 0; k < kUnit / 256; k++) {
    const uint32_t p = p0 + k * 256 + tid;
    if (p >= M) break;
    const uint32_t cur = src[p];
    if (bits == 0) { out[p] = cur; continue; }
    const uint32_t y = p / w, x = p - y * w;
    const uint32_t e = eimg[(y >> bits) * s.cross_tx + (x >> bits)];
    const int32_t r2b = (int32_t)(int8_t)(e >> 16), g2b = (int32_t)(int8_t)(e >> 8), g2r = (int32_t)(int8_t)e;
    const int32_t red = (int32_t)(int8_t)(cur >> 16), green = (int32_t)(int8_t)(cur >> 8);
    const uint32_t nr = ((cur >> 16) - (uint32_t)color_delta(g2r, green)) & 0xffu;
    const uint32_t nb = (cur - (uint32_t)color_delta(g2b, green) - (uint32_t)color_delta(r2b, red)) & 0xffu;
    out[p] = (cur & 0xff00ff00u) | (nr << 16) | nb;
  }
}

// =============================================================================================
// Candidate-list form of the same search, for tiles whose window does not hold too many scattered
// equal pixels (natural images, graphics, flat areas).  Exhaustive search scores every distance
// whose first pixel matches; most of those cannot win:
//   * Let p start a run of F equal pixels, and let q-1 and q be two sources of the same value next to
//     each other (both within the 1024 before p).  With R(q) = length of the run of equal pixels
//     starting at q, the match from q runs min(R(q), F) pixels when R(q) != F, and on into whatever
//     follows when R(q) == F.  Unless R(q) == F the match from q-1 is at least as long, and it is
//     farther, so it wins (lz77.h:68).  Per run of equal source pixels that leaves two candidates:
//     its first pixel inside the window, and the one whose run remainder equals F ("aligned").
//   * Lengths are capped at 128, so for F >= 128 only the first pixel is needed.
// The first pixels of the source runs are bucketed by a hash of the pixel value (counting sort in
// LDS; the hash is a bijection split into bucket and tag, so bucket + tag equality is value equality).
// A position looks at its bucket, adds the run that is cut by the window start, and extends each
// candidate run by run (both sides advance by min of their run remainders).  Work is proportional
// to the number of equal-valued runs; a tile that exceeds a work budget (e.g. two-colour noise)
// is handed to the exhaustive kernel (k_lz77) through `dense_tile`.  Both give identical results.
// =============================================================================================
// Distance code of a distance <= 1024.  In an image wider than 1024 + 8 pixels such a distance stays in the
// row (dy = 0, dx = distance), so only dx <= 8 has a short code (lz77.cpp:104-112).
__device__ __forceinline__ uint32_t dist_code_of(uint32_t xsize, uint32_t xinv, uint32_t dist) {
  if (xsize > kWindow + 8) return dist <= 8 ? dist_xy_to_code(xsize, dist, 0, dist) : dist + 120;
  return dist_to_code_inv(xsize, xinv, dist);
}

// Per-phase cycle counts of k_lz77_sparse (a measurement build only, -DVP8L_PHASE_TIMING: the first wave of every
// workgroup adds the cycles between phase marks to g_sp_phase; read with vp8l_debug_phase_cycles).
#ifdef VP8L_PHASE_TIMING
__device__ unsigned long long g_sp_phase[8];
#define VP8L_PHASE_BEGIN() unsigned long long ph_t = __builtin_readcyclecounter()
#define VP8L_PHASE_MARK(k)                                                     \
  do {                                                                          \
    const unsigned long long ph_n = __builtin_readcyclecounter();               \
    if (threadIdx.x == 0) atomicAdd(&g_sp_phase[k], ph_n - ph_t);               \
    ph_t = ph_n;                                                                \
  } while (0)
#else
#define VP8L_PHASE_BEGIN() do {} while (0)
#define VP8L_PHASE_MARK(k) do {} while (0)
#endif

// Orders a wave's LDS writes before the reads other lanes of the same wave make afterwards
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr uint32_t kSpThreads = 512;
constexpr uint32_t kSpWaves = kSpThreads / 64;
constexpr uint32_t kSpTile = 2048;                   // positions per workgroup
constexpr uint32_t kSpSources = kWindow + kSpTile;   // sources of a tile: the 1024 before it and the tile itself
constexpr uint32_t kSpPosBits = 12;                  // bits of a source index
constexpr uint32_t kSpBuckets = 1u << kSpPosBits;    // top bits of the mixed pixel select the bucket ...
constexpr uint32_t kSpTagShift = kSpPosBits;         // ... the remaining low bits are kept with the entry
constexpr uint32_t kSpMul = 0x9E3779B1u;             // odd: x -> x * kSpMul is a bijection on 32 bits, so
                                                     // same bucket + same tag <=> same pixel value
constexpr uint32_t kSpMaxBucket = 192;   // a bucket this full means heavy duplication: not a sparse tile
constexpr uint32_t kSpHeavyBucket = 24;  // no bucket this full: the tile is sparse, no need to count its values
constexpr uint32_t kSpMaxMeanBucket = 30;  // an entry shares its bucket with more than this many on average: not a sparse tile
constexpr uint32_t kSpPairsMeanBucket = 8;  // from this mean bucket length on, a tile without runs is searched through pairs of pixels
constexpr uint32_t kSpPairsGrid = 4096;     // workgroups of k_lz77_pairs (four fit a CU)
constexpr uint32_t kSpBudget = 700;      // candidate tests + extension steps one thread may spend (4 positions)
constexpr uint32_t kSpSeg = 7;           // window pixels per thread in the run-length pass
static_assert(kSpSources <= (1u << kSpPosBits) && kSpSources % kSpThreads == 0 && kSpTile % kSpThreads == 0, "entry layout");
static_assert(kSpThreads * kSpSeg >= kSpSources + kMaxLen, "run-length pass covers the window");
static_assert(kSpBuckets == 8 * kSpThreads, "8 buckets per thread in the scan");
static_assert(kSpTile == kLzTile, "one flag per tile");
constexpr uint32_t kSpQueue = 64;                           // candidates a wave collects before it scores them
constexpr uint32_t kSpWaveAux = 2 * 64 + kSpQueue;          // words per wave: 128 keys + the queue
constexpr uint32_t kSpAuxWords = (kSpSources + kMaxLen) / 2;
static_assert(kSpWaves * kSpWaveAux <= kSpAuxWords, "keys and queues fit where the run lengths would be");
static_assert(kSpTile / kSpThreads == 4, "positions of a thread are handled as two pairs");

// Flags of a tile (`dense_tile`): 0 done here, kTileExhaustive: left to the exhaustive kernel, kTilePairs: left to
// k_lz77_pairs.  That kernel is this one again for the tiles without runs whose buckets are long (PAIRS_KERNEL, see
// `grouped` below): with that path compiled in, the common path of this kernel got 12 % slower (60 instead of 40
// registers, other schedules), and those tiles are few or many by the kind of image, so they are listed and redone.
constexpr uint8_t kTileExhaustive = 1, kTilePairs = 2;
// Leaves a tile to the exhaustive search: its flag, the count of such tiles, and - for the first kFewTiles of a call - its
// number in the list behind the counters (given_up points at the count; the list begins kFewListAt words behind it).
constexpr uint32_t kFewListAt = 14;
__device__ __forceinline__ void give_up_tile(uint8_t *__restrict__ dense_tile, uint32_t *__restrict__ given_up, uint32_t tile) {
  dense_tile[tile] = kTileExhaustive;
  // (additions to one address go one after another, 11 ns each; counting one tile in 64 for 64 once the count is large gave
  // nothing for the 57 thousand given-up tiles of a palette batch: the additions wait while others search)
  const uint32_t k = atomicAdd(given_up, 1u);
  if (k < kFewTiles) given_up[kFewListAt + k] = tile;
}
constexpr uint8_t kTileDoneDominant = 4;  // searched by k_lz77_dominant (vp8l_encode_dense.h), before or after this kernel
template <bool PAIRS_KERNEL>
__device__ __forceinline__ void lz77_sparse_tile(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                 const uint32_t *__restrict__ sb_first, uint32_t n, const uint32_t tile,
                                                 uint32_t *__restrict__ pix, uint32_t *__restrict__ match,
                                                 uint8_t *__restrict__ adv, uint8_t *__restrict__ dense_tile, uint32_t tune_mean_bucket,
                                                 uint32_t tune_budget, uint32_t tune_max_bucket, uint32_t *__restrict__ given_up) {
  // given_up: counts the tiles left to the exhaustive kernel (a wave that gives up alone counts for itself: an upper bound)
  // (the top bit of tune_mean_bucket: k_lz77_dominant has looked at the tiles first, and the ones it searched are to be skipped)
  __shared__ __attribute__((aligned(16))) uint32_t px[kSpSources + kMaxLen + 4];
  // Tiles with runs: runlen[i] = pixels equal to pixel i from there on (within the window), 16 bits each.  Tiles without
  // runs use the same memory per wave: the best match so far of the 128 positions the wave is working on and a queue of
  // 64 candidates (see the search below).
  __shared__ uint32_t aux[kSpAuxWords];
  uint16_t *runlen = reinterpret_cast<uint16_t *>(aux);
  // counts, then insert positions; after the scatter cursor[h] = end of bucket h.  16 bits each (at most kSpSources), two to
  // a word that the LDS atomics add to: 8 KB less LDS lets a fourth workgroup onto the CU
  __shared__ uint32_t cursor2[kSpBuckets / 2];
  const uint16_t *cursor = reinterpret_cast<const uint16_t *>(cursor2);
  __shared__ uint32_t items[kSpSources];   // (low bits of the mixed pixel) << 11 | index in px
  __shared__ uint32_t wsum[kSpWaves];
  __shared__ uint32_t wmin[kSpWaves];
  __shared__ uint32_t s_abort, s_inside, s_sumsq, s_heavy, s_given;
  constexpr uint32_t kPerSuper = kSuper / kSpTile;
  const uint32_t tid = threadIdx.x;
  const uint32_t sb = tile / kPerSuper, img = find_image(sb_first, n, sb);
  const uint32_t M = st[img].M, xsize = st[img].xsize, xinv = st[img].xinv;
  const uint32_t a = (tile - sb_first[img] * kPerSuper) * kSpTile;
  if (a >= M) return;
  const uint32_t b = a + kSpTile < M ? a + kSpTile : M;
  const uint32_t lo = a > kWindow ? a - kWindow : 0;              // first possible source
  const uint32_t hi = b + kMaxLen - 1 < M ? b + kMaxLen - 1 : M;  // one past the last pixel an extension may read
  const uint32_t W = hi - lo;
  const uint64_t pos0 = imgs[img].pos0;
  if (PAIRS_KERNEL && dense_tile[(pos0 + a) / kLzTile] != kTilePairs) return;
  if (!PAIRS_KERNEL && (tune_mean_bucket >> 31) && dense_tile[(pos0 + a) / kLzTile] == kTileDoneDominant) return;  // (that kernel ran first)
  tune_mean_bucket &= 0x7fffffffu;
  const uint32_t *data = pix + pos0;
  const uint32_t lane = tid & 63, wave = tid >> 6;
  VP8L_PHASE_BEGIN();
  for (uint32_t i = tid; i < kSpBuckets / 2; i += kSpThreads) cursor2[i] = 0;
  if (tid == 0) { s_abort = 0; s_inside = 0; s_sumsq = 0; s_heavy = 0; s_given = 0; }
  // lo is a multiple of 4 (a and kWindow are) and the pixel array has readable slack past M: 16-byte loads
  if (!PAIRS_KERNEL && st[img].fused) {
    // Subtract-green path: the residual is made here from the caller's pixels (vp8l.h:594-597,
    // transform.cpp:88-106) and this tile's part of it is stored for the later stages; k_transform skips the image.
    const uint32_t *argb = imgs[img].argb;  // 16-byte aligned, exactly M pixels
    for (uint32_t i = tid * 4; i < W; i += kSpThreads * 4) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (lo + i + 4 <= M) {
        v = *reinterpret_cast<const uint4 *>(argb + lo + i);
      } else {
        v.x = argb[lo + i];
        if (lo + i + 1 < M) v.y = argb[lo + i + 1];
        if (lo + i + 2 < M) v.z = argb[lo + i + 2];
      }
      v.x = subtract_green(v.x); v.y = subtract_green(v.y); v.z = subtract_green(v.z); v.w = subtract_green(v.w);
      *reinterpret_cast<uint4 *>(&px[i]) = v;
      const uint32_t p = lo + i;
      if (p >= a && p < b) *reinterpret_cast<uint4 *>(pix + pos0 + p) = v;  // the region has slack past M
    }
  } else {
    for (uint32_t i = tid * 4; i < W; i += kSpThreads * 4) *reinterpret_cast<uint4 *>(&px[i]) = *reinterpret_cast<const uint4 *>(data + lo + i);
  }
  __syncthreads();
  VP8L_PHASE_MARK(0);  // window loaded (and the residual made and stored)
  const uint32_t nsrc = b - lo;  // pixels lo..b-1 can be sources
  uint32_t mixed[kSpSources / kSpThreads];  // this thread's sources tid, tid+512, ...: pixel * kSpMul
  uint32_t first_of_run = 0;         // bit k: source k*512+tid starts a run (or the window)
  uint32_t valid = 0;                // bit k: source k*512+tid exists
#pragma unroll
  for (uint32_t k = 0; k < kSpSources / kSpThreads; k++) {
    const uint32_t i = k * kSpThreads + tid;
    mixed[k] = px[i] * kSpMul;  // in bounds: px has more than kSpSources entries
    if (i < nsrc) {
      valid |= 1u << k;
      if (i == 0 || px[i - 1] != px[i]) first_of_run |= 1u << k;
    }
  }
  // Hardly any equal neighbours (natural images): every pixel is treated as a run of one, which is the
  // same search without the run bookkeeping.
  {
    const uint32_t inside = wave_sum_u32((uint32_t)__popc(valid & ~first_of_run));
    if (lane == 0 && inside) atomicAdd(&s_inside, inside);
  }
  __syncthreads();
  const bool plain = PAIRS_KERNEL || s_inside * 16 < nsrc;  // (the tiles of the pairs kernel have been found to be so)
  if (plain) first_of_run = valid;
  if (!plain) {
    // ---- run lengths: a run ends at j when px[j+1] differs or j is the last pixel of the window
    const uint32_t s0 = tid * kSpSeg;
    uint32_t v[kSpSeg + 1];
#pragma unroll
    for (uint32_t j = 0; j <= kSpSeg; j++) v[j] = s0 + j < W ? px[s0 + j] : 0u;
    uint32_t ends = 0;  // bit j: a run ends at s0 + j
#pragma unroll
    for (uint32_t j = 0; j < kSpSeg; j++)
      if (s0 + j < W && (s0 + j == W - 1 || v[j] != v[j + 1])) ends |= 1u << j;
    const uint32_t firstb = ends ? s0 + (uint32_t)__builtin_ctz(ends) : 0xffffu;
    // nearest run end in a later segment: suffix minimum over the threads
    uint32_t m = firstb;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t other = (uint32_t)__shfl_down((int)m, o, 64);
      if (lane + (uint32_t)o < 64) m = other < m ? other : m;
    }
    if (lane == 0) wmin[wave] = m;
    uint32_t after = (uint32_t)__shfl_down((int)m, 1, 64);
    if (lane == 63) after = 0xffffu;
    __syncthreads();
    for (uint32_t w = wave + 1; w < kSpWaves; w++) after = wmin[w] < after ? wmin[w] : after;
    uint32_t end = after;
#pragma unroll
    for (int j = (int)kSpSeg - 1; j >= 0; j--) {
      if ((ends >> j) & 1u) end = s0 + (uint32_t)j;
      if (s0 + (uint32_t)j < W) runlen[s0 + j] = (uint16_t)(end - (s0 + j) + 1);
    }
  }
  VP8L_PHASE_MARK(1);  // hashes, run starts, run lengths
  if (!PAIRS_KERNEL) {
#pragma unroll
    for (uint32_t k = 0; k < kSpSources / kSpThreads; k++)
      if ((first_of_run >> k) & 1u) {
        const uint32_t hb = mixed[k] >> (32 - kSpPosBits);
        atomicAdd(&cursor2[hb >> 1], 1u << (16 * (hb & 1u)));
      }
  }
  __syncthreads();
  VP8L_PHASE_MARK(2);  // bucket counts
  // exclusive scan of the bucket counts (8 per thread)
  uint32_t c[8], sum = 0, mx = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t two = PAIRS_KERNEL ? 0u : cursor2[tid * 4 + k];
    c[2 * k] = two & 0xffffu;
    c[2 * k + 1] = two >> 16;
  }
#pragma unroll
  for (int k = 0; k < 8; k++) { sum += c[k]; mx = c[k] > mx ? c[k] : mx; }
  const uint32_t incl = wave_incl_scan_u32(sum, (int)lane);
  if (lane == 63) wsum[wave] = incl;
  if (mx > tune_max_bucket) s_abort = 1;
  if (mx >= kSpHeavyBucket) s_heavy = 1;
  __syncthreads();
  if (s_abort) { if (tid == 0) { give_up_tile(dense_tile, given_up, (uint32_t)((pos0 + a) / kLzTile)); } return; }
  uint32_t run = incl - sum;
  for (uint32_t k = 0; k < wave; k++) run += wsum[k];
  uint32_t entries = 0;  // of all buckets (counted for tiles with a full bucket only)
  if (s_heavy) {
    // Some value is frequent.  Many entries of few values (two-colour noise bundled eight pixels to a word, 16 colours
    // two to a word): every position has dozens of candidates that extend a little, and the exhaustive kernel is
    // cheaper than finding that out by running out of budget.  The measure is the bucket length an entry sees on
    // average, sum of squares over sum: a position tests about a third of that many candidates (the window is a
    // third of the sources).  The count per distinct value, used before, sent predictor residuals of photographs away
    // too (500 values, 6 entries each, but 6 candidates per position: 1.6 times faster here than there).
    uint32_t sq = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) sq += c[k] * c[k];
    sq = wave_sum_u32(sq);
    if (lane == 0 && sq) atomicAdd(&s_sumsq, sq);
    __syncthreads();
    for (uint32_t k = 0; k < kSpWaves; k++) entries += wsum[k];
    if (s_sumsq > tune_mean_bucket * entries) { if (tid == 0) { give_up_tile(dense_tile, given_up, (uint32_t)((pos0 + a) / kLzTile)); } return; }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t lo16 = run, hi16 = run + c[2 * k];
    cursor2[tid * 4 + k] = lo16 | (hi16 << 16);
    run = hi16 + c[2 * k + 1];
  }
  __syncthreads();
  // A tile without runs that has a full bucket scatters its sources group by group (a group = the kSpThreads sources
  // of one round, consecutive indices), so that a bucket holds its entries in the order of the groups: a walk can then
  // stop where the groups begin that hold no source of its position (see collect_and_score).
  // ... and whose entries share their bucket with kSpPairsMeanBucket others or more on average: below that, walks are
  // short enough and two more counting sorts do not pay.
  const bool grouped = PAIRS_KERNEL || (plain && s_heavy && s_sumsq >= kSpPairsMeanBucket * entries);
  if (!PAIRS_KERNEL && grouped) {
    if (tid == 0) dense_tile[(pos0 + a) / kLzTile] = kTilePairs;
    return;
  }
  if (!grouped) {
#pragma unroll
    for (uint32_t k = 0; k < kSpSources / kSpThreads; k++) {
      const uint32_t i = k * kSpThreads + tid;
      if ((first_of_run >> k) & 1u) {
        const uint32_t hb = mixed[k] >> (32 - kSpPosBits), sh16 = 16 * (hb & 1u);
        items[(atomicAdd(&cursor2[hb >> 1], 1u << sh16) >> sh16) & 0xffffu] = (mixed[k] << kSpTagShift) | i;
      }
    }
  }
  __syncthreads();
  // The table of a grouped tile, built by the whole workgroup: on the sources' first two pixels (PAIRS) or on their
  // first one.  The mix of two pixels is no bijection: an entry whose tag fits is compared with both pixels.
  auto pair_mix = [](uint32_t mixed_first, uint32_t second) {
    uint32_t h = mixed_first + second * 0x85EBCA77u;
    h ^= h >> 15;
    return h * 0x2C1B3C6Du;
  };
  auto build_grouped = [&](auto pairs_tag) {
    constexpr bool PAIRS = decltype(pairs_tag)::value;
    for (uint32_t i = tid; i < kSpBuckets / 2; i += kSpThreads) cursor2[i] = 0;
    uint32_t hs[kSpSources / kSpThreads];
#pragma unroll
    for (uint32_t k = 0; k < kSpSources / kSpThreads; k++) hs[k] = PAIRS ? pair_mix(mixed[k], px[k * kSpThreads + tid + 1]) : mixed[k];
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kSpSources / kSpThreads; k++)
      if ((valid >> k) & 1u) {
        const uint32_t hb = hs[k] >> (32 - kSpPosBits);
        atomicAdd(&cursor2[hb >> 1], 1u << (16 * (hb & 1u)));
      }
    __syncthreads();
    uint32_t cc[8], total = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t two = cursor2[tid * 4 + k];
      cc[2 * k] = two & 0xffffu;
      cc[2 * k + 1] = two >> 16;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) total += cc[k];
    const uint32_t inc = wave_incl_scan_u32(total, (int)lane);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t at = inc - total;
    for (uint32_t k = 0; k < wave; k++) at += wsum[k];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t lo16 = at, hi16 = at + cc[2 * k];
      cursor2[tid * 4 + k] = lo16 | (hi16 << 16);
      at = hi16 + cc[2 * k + 1];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kSpSources / kSpThreads; k++) {  // group by group: a barrier after each round
      const uint32_t i = k * kSpThreads + tid;
      if ((valid >> k) & 1u) {
        const uint32_t hb = hs[k] >> (32 - kSpPosBits), sh16 = 16 * (hb & 1u);
        items[(atomicAdd(&cursor2[hb >> 1], 1u << sh16) >> sh16) & 0xffffu] = (hs[k] << kSpTagShift) | i;
      }
      __syncthreads();
    }
  };
  VP8L_PHASE_MARK(3);  // scan of the counts and scatter into the buckets
  // ---- tiles without runs: collect, then score.  A wave walks the buckets of 64 positions at a time and only *collects* the sources that
  // hold the position's value within its window (most walks find none, and a wave that scored them where it found
  // them would run the whole scoring code for one or two lanes).  Candidates go into the wave's queue; whenever 64 are
  // waiting, and at the end of two rounds of positions, every lane scores one: it extends the match and raises the
  // position's key length << 11 | distance with an LDS maximum - the longest match wins, among equals the farthest
  // (lz77.h:68), which is the order of the keys, so candidates can be scored in any order.  Queue and keys (`mem`: 128
  // keys, 64 queue entries) belong to the wave alone: no workgroup barrier in here, and results are stored as soon as a
  // pair of rounds is done (a tile that gives up later is redone as a whole by the exhaustive kernel).
  auto collect_and_score = [&](uint32_t *keys, auto grouped_tag, auto pairs_tag, uint64_t (&open_positions)[4]) -> bool {
    // GROUPED: the buckets hold their entries group by group (above).  The sources of a position lie in the groups from
    // the one of its window's first index to its own, so positions of the tile's first half walk a bucket forwards and
    // stop at the first entry of a later group than their own, the others walk it backwards and stop at the first entry
    // of a group before their window's: 58 % of a bucket on average instead of all of it.  Worth its five barriers and
    // two instructions per step only where buckets are long.
    // PAIRS: the table is on the sources' first two pixels: every candidate matches for two pixels at least, and the
    // positions left with a shorter match or none are not stored but returned in `open_positions` (a mask of lanes per
    // round): their best match is the farthest source of their value, which the table on single pixels will tell.
    // Returns true when the tile was given up (then the exhaustive kernel redoes it).
    constexpr bool GROUPED = decltype(grouped_tag)::value;
    constexpr bool PAIRS = decltype(pairs_tag)::value;
    uint32_t *queue = keys + 128;
    uint32_t spent = 0;
#pragma unroll 1
    for (uint32_t pair = 0; pair < 2; pair++) {
      keys[lane] = 0;
      keys[64 + lane] = 0;
      const uint32_t ip0 = a - lo + pair * (2 * kSpThreads) + wave * 64;  // window index of lane 0's position in round 0
      uint32_t qn = 0;                                                       // candidates waiting (the same in every lane)
      auto score = [&](uint32_t n) {
        wave_sync_lds();
        if (lane < n) {
          const uint32_t e = queue[lane];
          const uint32_t s = e & ((1u << kSpPosBits) - 1), slot = e >> kSpPosBits;  // slot = round * 64 + owner's lane
          const uint32_t ip = ip0 + (slot >> 6) * kSpThreads + (slot & 63u);
          const uint32_t left = M - (lo + ip), rem = left < kMaxLen ? left : kMaxLen;
          const uint32_t cur = keys[slot];  // what it must beat (others may be raising it meanwhile)
          uint32_t key = cur;
          // scores the source at index q (same value as the position, among the 1024 before it)
          auto candidate = [&](uint32_t q) {
            const uint32_t d = ip - q;
            const uint32_t maxl = d < rem ? d : rem;  // no overlap with the source, at most 128, inside the image
            // what it must reach to replace the best so far: longer, or as long and farther (lz77.h:68)
            const uint32_t need = (key >> 11) + (d > (key & 2047u) ? 0u : 1u);
            if (maxl < need) return;
            if (need > 1 && px[q + need - 1] != px[ip + need - 1]) return;
            uint32_t l = 1;
            while (l < maxl && px[q + l] == px[ip + l]) l++;
            spent += l;
            if (l >= need) key = (l << 11) | d;
          };
          candidate(s);
          if (key > cur) atomicMax(&keys[slot], key);
        }
        wave_sync_lds();
      };
      // the whole wave offers: lanes with `cand` put source t of their position of round kk into the queue; a lane
      // whose candidate found no room gets false and offers it again (the queue is scored and emptied when it is full)
      auto offer = [&](bool cand, uint32_t t, uint32_t kk) {
        const uint64_t cm = __ballot(cand);
        if (!cm) return true;
        const uint32_t at = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
        const bool fits = at < kSpQueue;
        if (cand && fits) queue[at] = ((kk * 64 + lane) << kSpPosBits) | t;
        qn += (uint32_t)__popcll(cm);
        if (qn >= kSpQueue) {
          score(kSpQueue);
          qn = 0;
        }
        return !cand || fits;
      };
      // A source whose second pixel differs matches for one pixel whatever else holds, and among those the farthest
      // wins: they need no scoring, only the largest distance (kept here per round as key 1 << 11 | distance).  Where
      // few values make up much of a tile (predictor residuals), these are nearly all the candidates.
      uint32_t one_pixel[2] = {0, 0};
#pragma unroll
      for (uint32_t kk = 0; kk < 2; kk++) {
        const uint32_t ip = ip0 + kk * kSpThreads + lane;
        uint32_t i = 0, stop = 0, step = 1, tag = 0, span = 0, x0 = 0, x1 = 0;
        uint32_t first_index = 0, indices = 1u << kSpPosBits;  // GROUPED: entries with other indices end the walk
        if (lo + ip < b) {
          const uint32_t x = px[ip];
          x1 = px[ip + 1];
          x0 = x;
          const uint32_t mxd = PAIRS ? pair_mix(x * kSpMul, x1) : x * kSpMul, h = mxd >> (32 - kSpPosBits);
          tag = mxd << kSpTagShift;
          stop = cursor[h];
          i = h ? cursor[h - 1] : 0u;
          span = ip < kWindow ? ip : kWindow;  // sources ip - span .. ip - 1
          if (GROUPED) {
            if (ip < kSpTile) {
              indices = (ip + kSpThreads - 1) & ~(kSpThreads - 1);  // up to the end of the position's own group
            } else {
              first_index = (ip - span) & ~(kSpThreads - 1);  // from the group of the window's first index on
              const uint32_t beg = i;
              i = stop - 1;
              stop = beg - 1;
              step = ~0u;
            }
          }
        }
        for (;;) {
          bool live = i != stop;
          if (!__ballot(live)) break;
          // an entry of the same value has the tag's bits, so t is its index; any other t is at least 4096 and fails
          // the one comparison that also keeps the index among the `span` before the position
          uint32_t t = 0;
          bool cand = false;
          if (live) {
            const uint32_t e = items[i];
            if (GROUPED && (e & ((1u << kSpPosBits) - 1)) - first_index >= indices) {
              i = stop;  // the groups from here on hold no source of this position
              live = false;
            } else {
              t = e ^ tag;
              cand = ip - 1 - t < span;
              if (PAIRS) {
                if (cand && (px[t] != x0 || px[t + 1] != x1)) cand = false;  // another pair with the same mix
              } else if (cand && px[t + 1] != x1) {
                const uint32_t k1 = (1u << 11) | (ip - t);
                one_pixel[kk] = one_pixel[kk] > k1 ? one_pixel[kk] : k1;
                cand = false;
              }
            }
          }
          const bool taken = offer(cand, t, kk);  // by every lane: the queue is the wave's
          if (live && taken) { i += step; spent++; }
        }
      }
      if (qn) score(qn);
      uint64_t still_open[2] = {0, 0};
#pragma unroll
      for (uint32_t kk = 0; kk < 2; kk++) {
        const uint32_t p = lo + ip0 + kk * kSpThreads + lane;
        bool open = false;
        if (p < b) {
          uint32_t key = keys[kk * 64 + lane];
          key = key > one_pixel[kk] ? key : one_pixel[kk];
          if (p == M - 1) key = 0;  // a lone look-ahead pixel is always a literal (lz77.h:42-49)
          open = PAIRS && key < (2u << 11) && p != M - 1;
          if (!open) {
            const uint32_t o = key ? ((key >> 11) | ((key & 2047u) << 8) | (dist_code_of(xsize, xinv, key & 2047u) << 19)) : 0u;
            match[pos0 + p] = o;
            adv[pos0 + p] = (uint8_t)((o & 0xff) ? (o & 0xff) : 1u);
          }
        }
        if (PAIRS) still_open[kk] = __ballot(open);
      }
      if (PAIRS) {
        if (pair == 0) { open_positions[0] = still_open[0]; open_positions[1] = still_open[1]; }
        else { open_positions[2] = still_open[0]; open_positions[3] = still_open[1]; }
      }
      if (__ballot(spent > tune_budget)) {  // too much work for this form: the exhaustive kernel redoes the tile
        if (lane == 0 && atomicExch(&s_given, 1u) == 0) give_up_tile(dense_tile, given_up, (uint32_t)((pos0 + a) / kLzTile));  // (once per tile: every wave gets here on its own)
        return true;
      }
    }
    return false;
  };
  // The positions that the table on pairs left open (a mask of lanes per round of the wave): the farthest source of the
  // position's value among the 1024 before it matches for one pixel and is the best match (lz77.h:68); none: a literal.
  // 64 open positions at a time walk the table on single pixels; `list` is the wave's own memory.
  auto farthest_of_value = [&](uint16_t *list, const uint64_t (&open_positions)[4]) {
    uint32_t count = 0;
#pragma unroll
    for (uint32_t r = 0; r < 4; r++) {
      const uint64_t m = open_positions[r];
      if ((m >> lane) & 1u) {
        const uint32_t at = count + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        list[at] = (uint16_t)(a - lo + (r >> 1) * (2 * kSpThreads) + (r & 1u) * kSpThreads + wave * 64 + lane);
      }
      count += (uint32_t)__popcll(m);
    }
    wave_sync_lds();
    for (uint32_t base = 0; base < count; base += 64) {
      const bool mine = base + lane < count;
      const uint32_t ip = mine ? list[base + lane] : 0u;
      uint32_t i = 0, stop = 0, step = 1, tag = 0, span = 0, far = 0;
      uint32_t first_index = 0, indices = 1u << kSpPosBits;
      if (mine) {
        const uint32_t mxd = px[ip] * kSpMul, h = mxd >> (32 - kSpPosBits);
        tag = mxd << kSpTagShift;
        stop = cursor[h];
        i = h ? cursor[h - 1] : 0u;
        span = ip < kWindow ? ip : kWindow;
        if (ip < kSpTile) {
          indices = (ip + kSpThreads - 1) & ~(kSpThreads - 1);
        } else {
          first_index = (ip - span) & ~(kSpThreads - 1);
          const uint32_t beg = i;
          i = stop - 1;
          stop = beg - 1;
          step = ~0u;
        }
      }
      while (__ballot(i != stop)) {
        if (i != stop) {
          const uint32_t e = items[i];
          if ((e & ((1u << kSpPosBits) - 1)) - first_index >= indices) {
            i = stop;
          } else {
            const uint32_t t = e ^ tag;
            if (ip - 1 - t < span && ip - t > far) far = ip - t;
            i += step;
          }
        }
      }
      if (mine) {
        const uint32_t p = lo + ip;
        const uint32_t o = far ? (1u | (far << 8) | (dist_code_of(xsize, xinv, far) << 19)) : 0u;
        match[pos0 + p] = o;
        adv[pos0 + p] = 1;
      }
    }
  };
  if (plain) {  // keys and queues where the run lengths of other tiles are
    uint64_t open_positions[4] = {0, 0, 0, 0};
    if constexpr (PAIRS_KERNEL) {
     if (grouped) {
      // Full buckets: a few values hold much of the tile, and a walk over all sources of a position's value is long.
      // Pairs of pixels hardly repeat: the table on pairs finds every match of two pixels and more with short walks,
      // and only the positions that have none walk the table on single pixels afterwards, 64 to a round.
      build_grouped(std::true_type{});
      VP8L_PHASE_MARK(4);  // (tiles of k_lz77_pairs: 4 the table on pairs, 5 its walks and scores, 6 the table on pixels, 7 its walks)
      const bool gave_up = collect_and_score(aux + wave * kSpWaveAux, std::true_type{}, std::true_type{}, open_positions);
      __syncthreads();  // every wave is done with the table on pairs
      VP8L_PHASE_MARK(5);
      build_grouped(std::false_type{});
      VP8L_PHASE_MARK(6);
      if (!gave_up) farthest_of_value(reinterpret_cast<uint16_t *>(aux + wave * kSpWaveAux), open_positions);
      VP8L_PHASE_MARK(7);
      return;
     }
    }
    (void)collect_and_score(aux + wave * kSpWaveAux, std::false_type{}, std::false_type{}, open_positions);
    VP8L_PHASE_MARK(4);  // the search (and the results stored as they are known)
#ifdef VP8L_PHASE_TIMING
    __syncthreads();     // (measurement builds only: the waves of these tiles do not wait for one another otherwise)
    VP8L_PHASE_MARK(5);
#endif
    return;
  }
  // ---- tiles with runs: candidates are scored where they are found.  (Collecting them first, as above, made these
  // tiles slower, 4.4 -> 5.0 ms per 64 graphics images of 3840x2160: flat areas give a position many candidates, and
  // scored one after the other nearly all of them fail the first comparison against the best so far.)
  // Every thread: positions a+tid, a+tid+512, ...
  uint32_t out[kSpTile / kSpThreads];
  uint32_t spent = 0;
  bool over = false;
#pragma unroll 1
  for (uint32_t k = 0; k < kSpTile / kSpThreads; k++) {
    const uint32_t p = a + k * kSpThreads + tid;
    uint32_t key = 0;  // best so far: length << 11 | distance
    if (p < b && !over) {
      const uint32_t ip = p - lo, x = px[ip];
      const uint32_t mxd = x * kSpMul, h = mxd >> (32 - kSpPosBits), tag = mxd << kSpTagShift;
      const uint32_t end = cursor[h], beg = h ? cursor[h - 1] : 0u;
      const uint32_t rem = M - p < kMaxLen ? M - p : kMaxLen;
      const uint32_t F = runlen[ip];                          // >= 128 means "at least 128"
      const uint32_t wlo = ip > kWindow ? ip - kWindow : 0u;  // farthest source of p
      // scores the source at index q (same value as p, wlo <= q < ip)
      auto candidate = [&](uint32_t q) {
        const uint32_t d = ip - q;
        const uint32_t maxl = d < rem ? d : rem;  // no overlap with the source, at most 128, inside the image
        // what it must reach to replace the best so far: longer, or as long and farther (lz77.h:68)
        const uint32_t need = (key >> 11) + (d > (key & 2047u) ? 0u : 1u);
        if (maxl < need) return;
        if (need > 1 && px[q + need - 1] != px[ip + need - 1]) return;
        uint32_t l = 0;
        do {  // both sides advance by whole runs
          const uint32_t rq = runlen[q + l], rp = runlen[ip + l];
          l += rq < rp ? rq : rp;
          spent++;
        } while (l < maxl && px[q + l] == px[ip + l]);
        l = l < maxl ? l : maxl;
        if (l >= need) key = (l << 11) | d;
      };
      // a source run and, when it ends before p and is longer than p's run, the source aligned with the end of p's run
      auto source_run = [&](uint32_t s) {
        candidate(s);
        const uint32_t r = runlen[s];
        if (F < kMaxLen && r > F && s + r <= ip) candidate(s + r - F);
      };
      if (wlo > 0 && px[wlo] == x && px[wlo - 1] == x) source_run(wlo);  // the run that the window start cuts
      for (uint32_t i = beg; i < end; i++) {
        const uint32_t it = items[i];
        const uint32_t s = it & ((1u << kSpPosBits) - 1);
        spent++;
        if ((it ^ tag) >> kSpPosBits || s >= ip || s < wlo) continue;  // another value, or not one of the 1024 before p
        source_run(s);
      }
      if (spent > tune_budget) over = true;
      if (p == M - 1) key = 0;  // a lone look-ahead pixel is always a literal (lz77.h:42-49)
    }
    out[k] = key ? ((key >> 11) | ((key & 2047u) << 8) | (dist_code_of(xsize, xinv, key & 2047u) << 19)) : 0u;
  }
  VP8L_PHASE_MARK(4);  // the search (first wave's share; the workgroup ends with its slowest wave)
  if (over) s_abort = 1;
  __syncthreads();
  VP8L_PHASE_MARK(5);  // waiting for the other waves
  if (s_abort) { if (tid == 0) { give_up_tile(dense_tile, given_up, (uint32_t)((pos0 + a) / kLzTile)); } return; }
#pragma unroll
  for (uint32_t k = 0; k < kSpTile / kSpThreads; k++) {
    const uint32_t p = a + k * kSpThreads + tid;
    if (p < b) {
      match[pos0 + p] = out[k];
      adv[pos0 + p] = (uint8_t)((out[k] & 0xff) ? (out[k] & 0xff) : 1u);
    }
  }
  VP8L_PHASE_MARK(6);  // results stored
}

__global__ __launch_bounds__(kSpThreads) void k_lz77_sparse(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                     const uint32_t *__restrict__ sb_first, uint32_t n, uint32_t ntiles,
                                                     uint32_t *__restrict__ pix, uint32_t *__restrict__ match,
                                                     uint8_t *__restrict__ adv, uint8_t *__restrict__ dense_tile, uint32_t tune_mean_bucket,
                                                     uint32_t tune_budget, uint32_t tune_max_bucket, uint32_t *__restrict__ given_up) {
  // Workgroups go round-robin to the 8 XCDs, each with its own L2.  Neighbouring tiles share half of their
  // window, so every XCD takes one contiguous eighth of the tiles: the overlap is then read from its L2, not twice from HBM.
  const uint32_t per_xcd = gridDim.x / 8;  // the grid is a multiple of 8
  const uint32_t tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (tile >= ntiles) return;
  lz77_sparse_tile<false>(imgs, st, sb_first, n, tile, pix, match, adv, dense_tile, tune_mean_bucket, tune_budget, tune_max_bucket, given_up);
}

// The tiles that k_lz77_sparse marked kTilePairs (it has stored their residual already).  Often there are none, and a
// workgroup per tile that only looks and leaves costs 1.7 ms per 259200 tiles: a fixed grid, each workgroup taking every
// gridDim.x-th tile.  (Eight waves per SIMD are asked for by name: left alone, the compiler spends 113 registers on the
// loop and only two workgroups fit a CU.)
__global__ __launch_bounds__(kSpThreads, 8) void k_lz77_pairs(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                    const uint32_t *__restrict__ sb_first, uint32_t n, uint32_t ntiles,
                                                    uint32_t *__restrict__ pix, uint32_t *__restrict__ match,
                                                    uint8_t *__restrict__ adv, uint8_t *__restrict__ dense_tile, uint32_t tune_mean_bucket,
                                                    uint32_t tune_budget, uint32_t tune_max_bucket, uint32_t *__restrict__ given_up) {
  // 64 of the workgroup's tiles at a time: every lane looks at one flag, then the marked ones are done one after the other
  for (uint32_t base = blockIdx.x; base < ntiles; base += 64 * gridDim.x) {
    const uint32_t mine = base + (threadIdx.x & 63u) * gridDim.x;
    uint64_t marked = __ballot(mine < ntiles && dense_tile[mine] == kTilePairs);
    while (marked) {
      const uint32_t k = (uint32_t)__builtin_ctzll(marked);
      marked &= marked - 1;
      lz77_sparse_tile<true>(imgs, st, sb_first, n, base + k * gridDim.x, pix, match, adv, dense_tile, tune_mean_bucket, tune_budget, tune_max_bucket, given_up);
      __syncthreads();  // the tile's shared memory is the next tile's
    }
  }
}

}  // namespace vp8l
#include "vp8l_encode_dense.h"
#include "vp8l_encode_walk.h"
namespace vp8l {

__global__ __launch_bounds__(256, 4) void k_lz77(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                              const uint32_t *__restrict__ sb_first, uint32_t n,
                                              const uint32_t *__restrict__ pix, uint32_t *__restrict__ match,
                                              uint8_t *__restrict__ adv, const uint8_t *__restrict__ dense_tile,
                                              const uint32_t *__restrict__ given_up) {
  constexpr uint32_t kPerSuper = kSuper / kLzBlock;
  // none given up on: nothing to do; few: k_lz77_pieces searches them
  if (given_up && (uint32_t)__builtin_amdgcn_readfirstlane((int)*given_up) <= kFewTiles) return;
  // tiles the candidate-list kernel completed are done; it flags the ones it gave up on (tile number = wave number)
  if (dense_tile && __builtin_amdgcn_readfirstlane((int)dense_tile[blockIdx.x * kLzWaves + (threadIdx.x >> 6)]) != kTileExhaustive) return;
  const uint32_t blk = blockIdx.x, sb = blk / kPerSuper, img = find_image(sb_first, n, sb);
  const uint32_t M = (uint32_t)__builtin_amdgcn_readfirstlane((int)st[img].M);
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t a = (blk - sb_first[img] * kPerSuper) * kLzBlock + wave * kLzTile;
  const uint32_t b = a + kLzTile < M ? a + kLzTile : M;
  if (a == 0) a = kLzHead;  // the first positions of an image are k_lz77_head's
  if (a >= b) return;
  const uint64_t pos0 = imgs[img].pos0;
  const uint32_t xsize = st[img].xsize, xinv = st[img].xinv;
  lz77_tile<false>(pix + pos0, match + pos0, adv + pos0, M, a, b, lane, xsize, xinv);
}

// The same when a call leaves few tiles to this search (one small image, a few odd tiles of a batch): a wave per tile would
// leave most of the device idle for the 0.3 ms a tile takes, so each tile is cut into pieces of kLzPiece positions, a wave
// each (a piece first warms up on the 143 positions behind it: half as much work again, done by eight times the waves).
// Workgroup k takes the k-th tile of the list that give_up_tile keeps.
constexpr uint32_t kLzPiece = 256;
__global__ __launch_bounds__(64 * (kLzTile / kLzPiece)) void k_lz77_pieces(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                                          const uint32_t *__restrict__ sb_first, uint32_t n,
                                                                          const uint32_t *__restrict__ pix, uint32_t *__restrict__ match,
                                                                          uint8_t *__restrict__ adv, const uint8_t *__restrict__ dense_tile,
                                                                          const uint32_t *__restrict__ given_up) {
  const uint32_t count = (uint32_t)__builtin_amdgcn_readfirstlane((int)*given_up);
  if (count > kFewTiles || blockIdx.x >= count) return;
  const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)given_up[kFewListAt + blockIdx.x]);
  if (dense_tile[tile] != kTileExhaustive) return;  // (another form has searched it since)
  constexpr uint32_t kPerSuper = kSuper / kLzTile;
  const uint32_t img = find_image(sb_first, n, tile / kPerSuper);
  const uint32_t M = (uint32_t)__builtin_amdgcn_readfirstlane((int)st[img].M);
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t t0 = (tile - sb_first[img] * kPerSuper) * kLzTile;
  uint32_t a = t0 + wave * kLzPiece;
  uint32_t b = a + kLzPiece;
  b = b < M ? b : M;
  if (a < kLzHead) a = kLzHead;  // the first positions of an image are k_lz77_head's
  if (a >= b) return;
  const uint64_t pos0 = imgs[img].pos0;
  lz77_tile<false>(pix + pos0, match + pos0, adv + pos0, M, a, b, lane, st[img].xsize, st[img].xinv);
}

// The first kLzHead positions of every image: there, distances reach before the first pixel.  A kernel of its own,
// because testing for that costs 40 vector registers more, and the kernel above runs four waves per SIMD without them;
// in pieces of 128 positions, a wave each, so that it does not take as long as a whole tile does.
__global__ __launch_bounds__(64) void k_lz77_head(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                  const uint32_t *__restrict__ sb_first, uint32_t n,
                                                  const uint32_t *__restrict__ pix, uint32_t *__restrict__ match,
                                                  uint8_t *__restrict__ adv, const uint8_t *__restrict__ dense_tile) {
  const uint32_t img = blockIdx.x / (kLzHead / kLzHeadPiece);
  if (img >= n) return;
  if (dense_tile && __builtin_amdgcn_readfirstlane((int)dense_tile[(size_t)sb_first[img] * (kSuper / kLzTile)]) != kTileExhaustive) return;
  const uint32_t M = (uint32_t)__builtin_amdgcn_readfirstlane((int)st[img].M);
  const uint32_t a = (blockIdx.x % (kLzHead / kLzHeadPiece)) * kLzHeadPiece;
  uint32_t b = a + kLzHeadPiece;
  b = b < M ? b : M;
  if (a >= b) return;
  const uint64_t pos0 = imgs[img].pos0;
  lz77_tile<true>(pix + pos0, match + pos0, adv + pos0, M, a, b, (int)threadIdx.x, st[img].xsize, st[img].xinv);
}

// =============================================================================================
// Greedy parse.  adv[p] = max(1, len[p]); the token starts are p0 = 0, p_{k+1} = p_k + adv[p_k].
// A jump is at most 128 long, so the chain enters every 128-position block through one of its
// 128 positions.  (1) per block, exit[p] = where the chain that passes through p first lands in
// the next block; (2) per superblock of 128 blocks, the exit for each of the 128 possible
// entries; (3) per image, a short serial walk over superblocks; (4) per superblock, entries of
// its blocks, then every block marks its token starts.  All out of LDS.
// =============================================================================================
constexpr uint32_t kRowPad = kBlockLen + 4;  // LDS row stride: spreads the 128 block rows over the banks

// adv8[p] = max(1, len[p]) as written by the match kernels; positions at and past M read as 1
__device__ __forceinline__ void load_advances(const uint8_t *__restrict__ adv8, uint32_t M, uint32_t p0, uint8_t *adv, uint32_t tid, uint32_t threads = 128) {
  for (uint32_t i = tid * 16; i < kSuper; i += threads * 16) {
    uint4 v = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
    if (p0 + i + 15 < M) {
      v = *reinterpret_cast<const uint4 *>(adv8 + p0 + i);
    } else if (p0 + i < M) {
      uint32_t w[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
      for (uint32_t k = 0; k < 16 && p0 + i + k < M; k++)
        w[k >> 2] = (w[k >> 2] & ~(0xffu << (8 * (k & 3)))) | ((uint32_t)adv8[p0 + i + k] << (8 * (k & 3)));
      v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    const uint32_t row = i / kBlockLen, col = i % kBlockLen;
    uint32_t *d = reinterpret_cast<uint32_t *>(adv + row * kRowPad + col);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

// exit[p] for all p of block `row` (thread per block, backwards), in place: exit[i] needs adv[i] and exit[q] for q > i only
__device__ __forceinline__ void block_exits_in_place(uint8_t *adv, uint32_t row) {
  uint8_t *e = adv + row * kRowPad;
  for (int i = (int)kBlockLen - 1; i >= 0; i--) {
    const uint32_t q = (uint32_t)i + e[i];
    e[i] = (uint8_t)(q >= kBlockLen ? q - kBlockLen : e[q]);
  }
}

__global__ __launch_bounds__(128) void k_parse_exits(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                     const uint32_t *__restrict__ sb_first, uint32_t n,
                                                     const uint8_t *__restrict__ adv8, uint8_t *__restrict__ sx) {
  __shared__ __attribute__((aligned(16))) uint8_t ex[kBlocksPerSuper * kRowPad];
  const uint32_t sb = blockIdx.x, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  const uint32_t M = st[img].M;
  const uint32_t p0 = (sb - sb_first[img]) * kSuper;
  if (p0 >= M) return;
  load_advances(adv8 + imgs[img].pos0, M, p0, ex, tid);
  __syncthreads();
  block_exits_in_place(ex, tid);
  __syncthreads();
  uint32_t e = tid;
  for (uint32_t r = 0; r < kBlocksPerSuper; r++) e = ex[r * kRowPad + e];
  sx[(size_t)sb * kBlockLen + tid] = (uint8_t)e;
}

// Per image: where the chain enters every superblock, from the superblocks' exit tables.  One entry value goes through
// all of them, a chain of dependent lookups; 128 tables a round out of LDS, in three steps: every group of 8 tables is
// composed into one (all threads: 16 groups x 128 entries, 8 lookups each), one thread takes the entry through the 16
// group tables, 16 threads take their group's entry through its 8 tables.  (One thread walking all the tables took 0.57 ms
// for a 16383 x 16383 image, a tenth of its encode time.)
__global__ __launch_bounds__(256) void k_parse_chain(const EncState *__restrict__ st, const uint32_t *__restrict__ sb_first,
                                                     const uint8_t *__restrict__ sx, uint8_t *__restrict__ entry) {
  constexpr uint32_t kRound = 128, kGroup = 8, kGroups = kRound / kGroup;
  __shared__ __attribute__((aligned(16))) uint8_t tab[kRound * kBlockLen];
  __shared__ uint8_t gmap[kGroups * kBlockLen];  // exit of a group of tables for every entry
  __shared__ uint8_t gent[kGroups];              // where the chain enters each group
  __shared__ uint32_t s_e;
  const uint32_t img = blockIdx.x, tid = threadIdx.x;
  const uint32_t nsb = (st[img].M + kSuper - 1) / kSuper, sb0 = sb_first[img];
  uint32_t e = 0;
  // a round's tables (16 KiB) are on their way from memory while the round before is walked: the load's latency, a
  // microsecond or two, was most of a round's time
  constexpr uint32_t kLoads = kRound * kBlockLen / 16 / 256;
  uint4 ahead[kLoads];
  auto fetch = [&](uint32_t g) {
    const uint32_t cnt = nsb - g < kRound ? nsb - g : kRound;
    const uint4 *src = reinterpret_cast<const uint4 *>(sx + (size_t)(sb0 + g) * kBlockLen);
#pragma unroll
    for (uint32_t k = 0; k < kLoads; k++) {
      const uint32_t i = k * 256 + tid;
      ahead[k] = i < cnt * kBlockLen / 16 ? src[i] : make_uint4(0, 0, 0, 0);
    }
  };
  if (nsb) fetch(0);
  for (uint32_t g = 0; g < nsb; g += kRound) {
    const uint32_t cnt = nsb - g < kRound ? nsb - g : kRound;
#pragma unroll
    for (uint32_t k = 0; k < kLoads; k++) reinterpret_cast<uint4 *>(tab)[k * 256 + tid] = ahead[k];
    if (g + kRound < nsb) fetch(g + kRound);
    __syncthreads();
    for (uint32_t k = tid; k < kGroups * kBlockLen; k += 256) {
      const uint32_t grp = k / kBlockLen;
      uint32_t x = k % kBlockLen;
      for (uint32_t j = 0; j < kGroup && grp * kGroup + j < cnt; j++) x = tab[(grp * kGroup + j) * kBlockLen + x];
      gmap[k] = (uint8_t)x;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t x = e;
      for (uint32_t grp = 0; grp < kGroups; grp++) { gent[grp] = (uint8_t)x; x = gmap[grp * kBlockLen + x]; }
      s_e = x;
    }
    __syncthreads();
    if (tid < kGroups) {
      uint32_t x = gent[tid];
      for (uint32_t j = 0; j < kGroup && tid * kGroup + j < cnt; j++) {
        entry[sb0 + g + tid * kGroup + j] = (uint8_t)x;
        x = tab[(tid * kGroup + j) * kBlockLen + x];
      }
    }
    e = s_e;
    __syncthreads();
  }
}

// The same for images of many superblocks (one 16383 x 16383 image has 16382: the kernel above walks them in 128 rounds, one
// after another, 0.4 ms of the image's 4.4), in two launches with a workgroup per round of 128 tables:
//   k_parse_chain_maps     composes the round's tables into one, the round's exit for each of the 128 entries (rmap);
//   k_parse_chain_entries  takes entry 0 through the maps of the rounds before its own (at most 127 lookups out of LDS), then
//                          through its own round as the kernel above does, and writes the entries of its superblocks.
// rmap: scratch of 128 bytes per round; the rounds of image i start at round sb_first[i] / 128 + i (no two images share one).
constexpr uint32_t kChainRound = 128, kChainGroup = 8, kChainGroups = kChainRound / kChainGroup;
constexpr uint32_t kChainMaxRounds = 128;  // (16384 superblocks: the largest image, 16384 x 16384, has as many)
constexpr uint32_t kChainSerialMax = 1024; // images up to so many superblocks (8 rounds) are walked by k_parse_chain, one workgroup each
// loads the tables of round r of an image (cnt of them) into tab and composes every group of 8 into gmap; ends with a barrier
__device__ __forceinline__ void chain_round_tables(const uint8_t *__restrict__ sx, uint32_t sb, uint32_t cnt, uint8_t *tab, uint8_t *gmap, uint32_t tid) {
  const uint4 *src = reinterpret_cast<const uint4 *>(sx + (size_t)sb * kBlockLen);
  for (uint32_t i = tid; i < kChainRound * kBlockLen / 16; i += 256)
    reinterpret_cast<uint4 *>(tab)[i] = i < cnt * kBlockLen / 16 ? src[i] : make_uint4(0, 0, 0, 0);
  __syncthreads();
  for (uint32_t k = tid; k < kChainGroups * kBlockLen; k += 256) {
    const uint32_t grp = k / kBlockLen;
    uint32_t x = k % kBlockLen;
    for (uint32_t j = 0; j < kChainGroup && grp * kChainGroup + j < cnt; j++) x = tab[(grp * kChainGroup + j) * kBlockLen + x];
    gmap[k] = (uint8_t)x;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_parse_chain_maps(const EncState *__restrict__ st, const uint32_t *__restrict__ sb_first,
                                                          const uint8_t *__restrict__ sx, uint8_t *__restrict__ rmap) {
  __shared__ __attribute__((aligned(16))) uint8_t tab[kChainRound * kBlockLen];
  __shared__ uint8_t gmap[kChainGroups * kBlockLen];
  const uint32_t img = blockIdx.y, r = blockIdx.x, tid = threadIdx.x;
  const uint32_t nsb = (st[img].M + kSuper - 1) / kSuper, sb0 = sb_first[img];
  if (r * kChainRound >= nsb) return;
  const uint32_t cnt = nsb - r * kChainRound < kChainRound ? nsb - r * kChainRound : kChainRound;
  chain_round_tables(sx, sb0 + r * kChainRound, cnt, tab, gmap, tid);
  if (tid < kBlockLen) {
    uint32_t x = tid;
    for (uint32_t grp = 0; grp < kChainGroups; grp++) x = gmap[grp * kBlockLen + x];  // (groups past cnt map every entry to itself)
    rmap[((size_t)(sb0 / kChainRound + img) + r) * kBlockLen + tid] = (uint8_t)x;
  }
}

__global__ __launch_bounds__(256) void k_parse_chain_entries(const EncState *__restrict__ st, const uint32_t *__restrict__ sb_first,
                                                             const uint8_t *__restrict__ sx, const uint8_t *__restrict__ rmap,
                                                             uint8_t *__restrict__ entry) {
  __shared__ __attribute__((aligned(16))) uint8_t tab[kChainRound * kBlockLen];  // first the maps of the rounds before, then the round's tables
  __shared__ uint8_t gmap[kChainGroups * kBlockLen];
  __shared__ uint8_t gent[kChainGroups];
  __shared__ uint32_t s_e;
  const uint32_t img = blockIdx.y, r = blockIdx.x, tid = threadIdx.x;
  const uint32_t nsb = (st[img].M + kSuper - 1) / kSuper, sb0 = sb_first[img];
  if (r * kChainRound >= nsb) return;
  if (r >= kChainMaxRounds) return;  // (the launch takes this form only where the maps of an image fit tab)
  const uint32_t cnt = nsb - r * kChainRound < kChainRound ? nsb - r * kChainRound : kChainRound;
  const uint4 *maps = reinterpret_cast<const uint4 *>(rmap + (size_t)(sb0 / kChainRound + img) * kBlockLen);
  for (uint32_t i = tid; i < r * kBlockLen / 16; i += 256) reinterpret_cast<uint4 *>(tab)[i] = maps[i];
  __syncthreads();
  if (tid == 0) {
    uint32_t x = 0;  // the chain enters an image at its first position
    for (uint32_t k = 0; k < r; k++) x = tab[k * kBlockLen + x];
    s_e = x;
  }
  __syncthreads();
  const uint32_t e = s_e;
  __syncthreads();
  chain_round_tables(sx, sb0 + r * kChainRound, cnt, tab, gmap, tid);
  if (tid == 0) {
    uint32_t x = e;
    for (uint32_t grp = 0; grp < kChainGroups; grp++) { gent[grp] = (uint8_t)x; x = gmap[grp * kBlockLen + x]; }
  }
  __syncthreads();
  if (tid < kChainGroups) {
    uint32_t x = gent[tid];
    for (uint32_t j = 0; j < kChainGroup && tid * kChainGroup + j < cnt; j++) {
      entry[sb0 + r * kChainRound + tid * kChainGroup + j] = (uint8_t)x;
      x = tab[(tid * kChainGroup + j) * kBlockLen + x];
    }
  }
}

// Marks the token starts of a superblock and, in the same kernel, counts their symbols (vp8l.h:516-535): the walks above are
// chains of dependent LDS reads that leave the memory system idle, the histogram is loads and LDS atomics, so the workgroups of
// one CU overlap the two.  The histogram takes over the LDS of the advances once the marks are made.
constexpr uint32_t kHistCopies = 3;  // interleaved copies of every counter: equal symbols in one wave meet less often
constexpr uint32_t kMarkThreads = 256;
__global__ __launch_bounds__(kMarkThreads) void k_parse_mark(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                             const uint32_t *__restrict__ sb_first, uint32_t n,
                                                             const uint8_t *__restrict__ adv8, const uint8_t *__restrict__ entry,
                                                             uint32_t *__restrict__ flags, const uint32_t *__restrict__ pix,
                                                             const uint32_t *__restrict__ match, uint32_t *__restrict__ histo,
                                                             uint8_t *__restrict__ unit_hist) {
  constexpr uint32_t kAdvBytes = kBlocksPerSuper * kRowPad, kHistWords = kNumCodes * kCodeStride * kHistCopies;
  static_assert(kHistWords * 4 <= kAdvBytes, "the histogram fits where the advances were");
  __shared__ __attribute__((aligned(16))) uint8_t adv[kAdvBytes];  // exits first, then the advances again, then the histogram
  __shared__ uint8_t ent[kBlocksPerSuper];
  __shared__ __attribute__((aligned(16))) uint32_t fl[kSuper / 32];  // token starts of the superblock
  const uint32_t sb = blockIdx.x, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  const uint32_t M = st[img].M;
  const uint32_t p0 = (sb - sb_first[img]) * kSuper;
  if (p0 >= M) return;
  const uint64_t pos0 = imgs[img].pos0;
  load_advances(adv8 + pos0, M, p0, adv, tid, kMarkThreads);
  __syncthreads();
  if (tid < kBlocksPerSuper) block_exits_in_place(adv, tid);
  __syncthreads();
  if (tid == 0) {
    uint32_t e = entry[sb];
    for (uint32_t r = 0; r < kBlocksPerSuper; r++) { ent[r] = (uint8_t)e; e = adv[r * kRowPad + e]; }
  }
  __syncthreads();
  load_advances(adv8 + pos0, M, p0, adv, tid, kMarkThreads);  // the same 16 KiB again, from cache
  __syncthreads();
  if (tid < kBlocksPerSuper) {
    unsigned long long lo = 0, hi = 0;  // token starts of this thread's block
    const uint32_t bstart = p0 + tid * kBlockLen;
    for (uint32_t q = ent[tid]; q < kBlockLen && bstart + q < M; q += adv[tid * kRowPad + q]) {
      if (q < 64) lo |= 1ull << q; else hi |= 1ull << (q - 64);
    }
    uint4 out;
    out.x = (uint32_t)lo; out.y = (uint32_t)(lo >> 32); out.z = (uint32_t)hi; out.w = (uint32_t)(hi >> 32);
    *reinterpret_cast<uint4 *>(flags + (pos0 + bstart) / 32) = out;
    *reinterpret_cast<uint4 *>(&fl[tid * 4]) = out;
  }
  if (!histo) return;  // the coder with several groups counts by itself
  __syncthreads();
  // The symbols are counted unit by unit (kUnit positions; a token belongs to the unit it starts in).  unit_hist (may be null):
  // the counts of every unit are left, 16 bits each, where the unit's advances were (this workgroup has read its superblock's
  // advances for the last time, and nobody else reads them): with them, the bits of a unit are a sum over the symbols of
  // count x (code length + extra bits) once the codes are known (k_unit_bits), instead of another pass over pixels and matches.
  uint32_t *h = reinterpret_cast<uint32_t *>(adv);
  constexpr uint32_t kCounters = kNumCodes * kCodeStride, kMine = (kCounters + kMarkThreads - 1) / kMarkThreads;
  static_assert(kCounters * 2 <= kUnit, "a unit's counts fit where its advances were");
  uint32_t tot[kMine];
#pragma unroll
  for (uint32_t j = 0; j < kMine; j++) tot[j] = 0;
  for (uint32_t i = tid; i < kHistWords; i += kMarkThreads) h[i] = 0;
  __syncthreads();
#pragma unroll 1
  for (uint32_t u = 0; u < kUnitsPerSuper && p0 + u * kUnit < M; u++) {
    // four consecutive positions per thread, everything loaded before anything is used (the arrays have readable slack past M)
#pragma unroll 2
    for (uint32_t kk = 0; kk < kUnit / (4 * kMarkThreads); kk++) {
      const uint32_t o = u * kUnit + (kk * kMarkThreads + tid) * 4, p = p0 + o;
      uint32_t f4 = p < M ? (fl[o >> 5] >> (o & 31)) & 15u : 0u;
      if (p < M && p + 4 > M) f4 &= (1u << (M - p)) - 1u;
      if (!f4) continue;
      const uint4 m4 = *reinterpret_cast<const uint4 *>(match + pos0 + p);
      const uint4 x4 = *reinterpret_cast<const uint4 *>(pix + pos0 + p);
      const uint32_t mm[4] = {m4.x, m4.y, m4.z, m4.w}, xx[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (!((f4 >> j) & 1u)) continue;
        const uint32_t m = mm[j], len = m & 0xff;
        if (len == 0) {
          // One count per channel and active lane.  Lanes that all hold the same pixel (a flat area) or the same alpha (an
          // opaque image) would serialise on one LDS address: then one lane adds the number of lanes.  One test per pixel
          // and one for alpha: the other channels of a pixel that differs somewhere spread over the counters' copies.
          const uint32_t px = xx[j];
          const uint64_t active = __ballot(1);
          const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)px), diff = px ^ first;
          const bool leader = (uint32_t)__builtin_ctzll(active) == (tid & 63u);
          const uint32_t lanes = (uint32_t)__popcll(active), copy = tid % kHistCopies;
          if (!__ballot(diff != 0)) {
            if (leader) {
              atomicAdd(&h[(kGreen * kCodeStride + ((first >> 8) & 0xff)) * kHistCopies], lanes);
              atomicAdd(&h[(kRed * kCodeStride + ((first >> 16) & 0xff)) * kHistCopies], lanes);
              atomicAdd(&h[(kBlue * kCodeStride + (first & 0xff)) * kHistCopies], lanes);
              atomicAdd(&h[(kAlpha * kCodeStride + (first >> 24)) * kHistCopies], lanes);
            }
          } else {
            atomicAdd(&h[(kGreen * kCodeStride + ((px >> 8) & 0xff)) * kHistCopies + copy], 1u);
            atomicAdd(&h[(kRed * kCodeStride + ((px >> 16) & 0xff)) * kHistCopies + copy], 1u);
            atomicAdd(&h[(kBlue * kCodeStride + (px & 0xff)) * kHistCopies + copy], 1u);
            if (!__ballot((diff >> 24) != 0)) {
              if (leader) atomicAdd(&h[(kAlpha * kCodeStride + (first >> 24)) * kHistCopies], lanes);
            } else {
              atomicAdd(&h[(kAlpha * kCodeStride + (px >> 24)) * kHistCopies + copy], 1u);
            }
          }
        } else {
          uint32_t sym, nb, eb;
          prefix_encode(len, sym, nb, eb);
          atomicAdd(&h[(kGreen * kCodeStride + 256 + sym) * kHistCopies + (tid % kHistCopies)], 1u);
          prefix_encode((m >> 19) & 0x7ff, sym, nb, eb);  // distance code, computed with the match
          atomicAdd(&h[(kDist * kCodeStride + sym) * kHistCopies + (tid % kHistCopies)], 1u);
        }
      }
    }
    __syncthreads();
    uint16_t *uh = unit_hist ? reinterpret_cast<uint16_t *>(unit_hist + pos0 + p0 + u * kUnit) : nullptr;
#pragma unroll
    for (uint32_t j = 0; j < kMine; j++) {
      const uint32_t i = j * kMarkThreads + tid;
      if (i < kCounters) {
        const uint32_t v = h[i * kHistCopies] + h[i * kHistCopies + 1] + h[i * kHistCopies + 2];
        h[i * kHistCopies] = 0; h[i * kHistCopies + 1] = 0; h[i * kHistCopies + 2] = 0;  // for the next unit
        tot[j] += v;
        if (uh) uh[i] = (uint16_t)v;
      }
    }
    __syncthreads();  // the counters are clear before the next unit counts
  }
  uint32_t *g = histo + (size_t)img * kNumCodes * kCodeStride;
#pragma unroll
  for (uint32_t j = 0; j < kMine; j++) {
    const uint32_t i = j * kMarkThreads + tid;
    if (i < kCounters && tot[j]) atomicAdd(&g[i], tot[j]);
  }
}

__device__ __forceinline__ uint32_t flag_at(const uint32_t *__restrict__ flags, uint64_t gpos) {
  return (flags[gpos >> 5] >> (gpos & 31)) & 1u;
}

// =============================================================================================
// Per image: the five Huffman codes and every bit in front of the first token.  A few hundred
// symbols of strictly sequential work (sorted-merge Huffman with the reference's tie rule, RLE,
// nested code): lane 0 of one wave runs the host/device core out of LDS.
// =============================================================================================
// A code's description takes at most 1 + 6 + 68 * 6 + 1 bits of preamble and, per run-length item (at most 280), a
// code-length symbol and up to 7 extra bits; the code-length code is built from at most 280 counts, so none of its codes is
// longer than 11 bits (a code of depth d needs Fibonacci(d + 2) counts): 5456 bits at most.
constexpr uint32_t kDescBufWords = 256;
static_assert(kDescBufWords * 32 >= 1 + 6 + 68 * 6 + 1 + 280 * (11 + 7), "a description fits its buffer");
struct HeaderShared {
  SubImageScratch sub;     // its histogram and tables serve the sub-images first, then the main image
  uint32_t palette[kMaxPalette];
  DescScratch desc[3];     // waves 1..3; wave 0 uses sub.desc
  int rc[kNumCodes];
  // the five descriptions of the main image's codes, written side by side and then appended to the stream: dbits[c] bits each,
  // in five buffers of kDescBufWords words that lie where the sub-images' match search had its arrays (sub.lz; the
  // sub-images are written by then) - with buffers of their own only two workgroups fit a CU instead of three
  uint32_t dbits[kNumCodes];
  int drc[kNumCodes];
};
static_assert(sizeof(SmallLzScratch) >= sizeof(uint32_t) * kNumCodes * kDescBufWords && alignof(SmallLzScratch) >= 4, "the descriptions fit where the sub-image search was");
static_assert(sizeof(HeaderShared) + 64 <= 160 * 1024 / 3, "three workgroups of k_entropy_header per CU");

// The used symbols' weights in symbol order and their ids sorted by (weight, id), by the whole wave
// (what huffman_lengths otherwise does with a serial insertion sort).  Returns the number of used symbols.
template <class Scratch>
__device__ __forceinline__ int wave_prepare_leaves(const uint32_t *h, int n, Scratch &hs, uint32_t lane) {
  uint32_t count = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + (int)lane;
    const uint32_t w = i < n ? h[i] : 0u;
    const uint64_t mask = __ballot(w != 0);
    if (w) hs.wt[count + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = w;
    count += (uint32_t)__popcll(mask);
  }
  __builtin_amdgcn_wave_barrier();
  for (uint32_t i = lane; i < count; i += 64) {
    const uint32_t wi = hs.wt[i];
    uint32_t r = 0;
    for (uint32_t j = 0; j < count; j++) {
      const uint32_t wj = hs.wt[j];
      r += (wj < wi || (wj == wi && j < i)) ? 1u : 0u;
    }
    hs.ord[r] = (uint16_t)i;
  }
  __builtin_amdgcn_wave_barrier();
  return (int)count;
}

// huffman_lengths (vp8l_enc_core.h) by a whole wave, after wave_prepare_leaves: the joins are one lane's (each needs the one
// before), the rest is not - the depth of every node by pointer jumping (a node's entry is parent << 16 | steps to it; eight
// rounds of "add my parent's steps, take its parent" reach 256 levels, and anything deeper than the longest code allowed
// fails either way), the lengths of the symbols from the depths of their leaves, the longest by a wave maximum.  One lane
// going through the same loops took 0.1 ms of dependent shared-memory reads per code.  Returns the same in every lane.
template <class Scratch>
__device__ __forceinline__ int wave_huffman_lengths(const uint32_t *h, int n, int max_len, uint8_t *lengths, Scratch &hs, int used, uint32_t lane) {
  constexpr int kPerLane = (2 * kCodeStride + 63) / 64;
  int nn = used;
  if (lane == 0 && used > 1) nn = huffman_join(hs, used);
  nn = __builtin_amdgcn_readfirstlane(nn);
  __builtin_amdgcn_wave_barrier();
  uint32_t *x = hs.wt;  // (the weights are not needed any more)
  for (int v = (int)lane; v < nn; v += 64) x[v] = v == nn - 1 ? ((uint32_t)v << 16) : (((uint32_t)hs.parent[v] << 16) | 1u);
  __builtin_amdgcn_wave_barrier();
#pragma unroll 1
  for (int round = 0; round < 8; round++) {
    uint32_t next[kPerLane];
#pragma unroll
    for (int k = 0; k < kPerLane; k++) {
      const int v = k * 64 + (int)lane;
      next[k] = 0;
      if (v < nn) {
        const uint32_t mine = x[v], up = x[mine >> 16];
        next[k] = (up & 0xffff0000u) | ((mine & 0xffffu) + (up & 0xffffu));
      }
    }
    __builtin_amdgcn_wave_barrier();  // (a wave's shared-memory accesses keep their order: every read above comes before the writes below)
#pragma unroll
    for (int k = 0; k < kPerLane; k++) {
      const int v = k * 64 + (int)lane;
      if (v < nn) x[v] = next[k];
    }
    __builtin_amdgcn_wave_barrier();
  }
  uint32_t leaf0 = 0, maxl = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + (int)lane;
    const uint32_t w = i < n ? h[i] : 0u;
    const uint64_t mask = __ballot(w != 0);
    if (i < n) {
      uint32_t d = 0;
      if (w) {
        d = x[leaf0 + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] & 0xffffu;
        if (used == 1) d = 1;  // a lone symbol gets length 1 (huffman_coding.h:398-402)
      }
      lengths[i] = (uint8_t)(d > 255 ? 255 : d);
      maxl = d > maxl ? d : maxl;
    }
    leaf0 += (uint32_t)__popcll(mask);
  }
  __builtin_amdgcn_wave_barrier();
  maxl = wave_max_dpp(maxl);
  if ((int)maxl > max_len || (int)maxl > kSaneCodeLen) return -3;
  return used;
}

// canonical_reversed_codes (vp8l_enc_core.h) by a whole wave: symbols per length by shared-memory atomics, the first code of
// every length by one lane (32 steps), then 64 symbols at a time in symbol order - a symbol's code is the next free one of
// its length plus the number of lanes before it with the same length.  work: kCanonWork words of shared memory.
__device__ __forceinline__ void wave_canonical_codes(const uint8_t *lengths, int n, uint32_t *codes, uint32_t *work, uint32_t lane) {
  uint32_t *count = work, *next = work + kSaneCodeLen + 2;
  if (lane < (uint32_t)kSaneCodeLen + 2) count[lane] = 0;
  __builtin_amdgcn_wave_barrier();
  for (int i = (int)lane; i < n; i += 64) { const uint32_t l = lengths[i]; if (l) atomicAdd(&count[l], 1u); }
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    uint32_t cur = 0;
    next[0] = 0;
    for (int l = 1; l <= kSaneCodeLen; l++) { cur = (cur + count[l - 1]) << 1; next[l] = cur; }
  }
  __builtin_amdgcn_wave_barrier();
  for (int base = 0; base < n; base += 64) {
    const int i = base + (int)lane;
    const uint32_t l = i < n ? lengths[i] : 0u;
    uint32_t code = 0;
    uint64_t todo = __ballot(l != 0);
    while (todo) {
      const uint32_t first = (uint32_t)__builtin_ctzll(todo);
      const uint32_t l0 = (uint32_t)__builtin_amdgcn_readlane((int)l, (int)first);
      const uint64_t same = __ballot(l == l0);
      const uint32_t free0 = next[l0];
      if (l == l0) code = free0 + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
      __builtin_amdgcn_wave_barrier();
      if (lane == first) next[l0] = free0 + (uint32_t)__popcll(same);
      __builtin_amdgcn_wave_barrier();
      todo &= ~same;
    }
    if (i < n) codes[i] = l ? (__builtin_bitreverse32(code) >> (32u - l)) : 0u;
  }
  __builtin_amdgcn_wave_barrier();
}

// The five codes of one coded image from its five histograms, by four waves: green, red, blue and alpha side
// by side (wave 0 then adds the distance code); sort, depths, lengths and codes of each are done by its wave, the joining of
// the tree and the description by the wave's first lane.  rc[c] < 0: that code failed (TooBigCodeLength).  Ends with a barrier.
// standard: codes of at most 15 bits (the histogram is then changed in place, huffman_lengths_limited).
// describe: a wave's first lane also writes the description of each code it has built (vp8l.h:536-540) into the code's
// own bit buffer dbuf[c] (dbits[c] bits, drc[c] = status), so the five descriptions are written side by side.
__device__ __forceinline__ void build_codes_4waves(uint32_t *histo5, CodeTables &t, int *rc, DescScratch &ds, uint32_t wave,
                                                   uint32_t lane, bool standard, uint32_t (*dbuf)[kDescBufWords] = nullptr,
                                                   uint32_t *dbits = nullptr, int *drc = nullptr) {
  HuffScratch &hs = ds.huff;
  for (uint32_t c = wave; c < (uint32_t)kNumCodes; c += 4) {
    const int nsym = alphabet_size((int)c);
    uint32_t *h = histo5 + c * kCodeStride;
    uint8_t *len = t.len + c * kCodeStride;
    const int used = wave_prepare_leaves(h, nsym, hs, lane);
    int u = wave_huffman_lengths(h, nsym, standard ? dialect_of(true).max_len : kMaxCodeLen, len, hs, used, lane);
    if (standard && u == -3) {  // too long for the standard dialect: one lane raises the small counts until it fits (rare)
      if (lane == 0) u = huffman_lengths_limited(h, nsym, dialect_of(true).max_len, len, hs);
      u = __builtin_amdgcn_readfirstlane(u);
      __builtin_amdgcn_wave_barrier();
    }
    if (u >= 0) wave_canonical_codes(len, nsym, t.code + c * kCodeStride, hs.wt, lane);
    if (lane == 0) {
      rc[c] = u;
      if (u >= 0) t.used[c] = (uint32_t)u;
      if (dbuf) {
        BitSink b;
        b.init(dbuf[c]);
        drc[c] = u < 0 ? u : write_code_desc(b, len, nsym, ds, standard);  // the code is built: its scratch is free again
        b.flush();
        dbits[c] = (uint32_t)b.bits();
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
}

// index i of the 5 x kCodeStride tables belongs to a code whose symbols are not written
__device__ __forceinline__ bool code_unwritten(const CodeTables *T, uint32_t i) {
  const uint32_t c = i / kCodeStride, sym = i % kCodeStride;
  return T->used[c] <= 1 && !(c == (uint32_t)kGreen && sym >= 256);
}

// Per image: the codes and every bit in front of the first token.  The parts with many independent items
// (palette matches, sorting) use the workgroup; the bits are written by one lane, in stream order.
__global__ __launch_bounds__(256) void k_entropy_header(const EncImage *__restrict__ imgs, EncState *__restrict__ st,
                                                        const uint32_t *__restrict__ histo, CodeTables *__restrict__ tables,
                                                        const uint32_t *__restrict__ modes, const uint32_t *__restrict__ elems,
                                                        uint8_t *__restrict__ out, uint32_t standard, ExtArgs ext) {
  __shared__ HeaderShared sh;
  __shared__ BitSink bw;
  __shared__ int s_rc;
  const uint32_t img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const EncImage im = imgs[img];
  const uint32_t ncolors = st[img].ncolors;
  const uint32_t *g = histo + (size_t)img * kNumCodes * kCodeStride;
  for (uint32_t i = tid; i < kMaxPalette; i += 256) sh.palette[i] = st[img].palette[i];
  if (tid == 0) {
    bw.init(reinterpret_cast<uint32_t *>(out + im.out_off));
    write_stream_head(bw, im.w, im.h, ncolors, standard != 0);
    s_rc = 0;
  }
  __syncthreads();
  DescScratch &ds = wave == 0 ? sh.sub.desc : sh.desc[wave - 1];
  // Sub-images go in front of the main image's codes: the palette as ncolors x 1 (vp8l.h:455-457), or the predictor
  // modes as pred_tx x pred_ty and then the cross-colour elements as cross_tx x cross_ty (read by ReadTransform
  // vp8l.h:111-121), in the order the transforms were applied.
  const uint32_t pred_bits = ncolors ? 0u : st[img].pred_bits, cross_bits = ncolors ? 0u : st[img].cross_bits;
  // which = 2: the entropy image (Huffman group of every tile), behind the transforms and the colour-cache bits
  // (ReadSpatiallyCodedImage vp8l.h:180-215); only the coder with several groups writes one.
  const uint32_t ngroups = ext.on ? st[img].ngroups : 1u;
  for (int which = 0; which < 3; which++) {
    const bool palette = which == 0 && ncolors, pred = which == 0 && pred_bits, cross = which == 1 && cross_bits, meta = which == 2 && ngroups > 1;
    if (which == 2 && ext.on && tid == 0 && s_rc == 0) {
      bw.put(0, 1);  // no more transforms
      if (ext.cache_bits) { bw.put(1, 1); bw.put(ext.cache_bits, 4); } else bw.put(0, 1);   // vp8l.h:230-241
      if (meta) { bw.put(1, 1); bw.put(st[img].meta_bits - 2, 3); } else bw.put(0, 1);      // vp8l.h:191-193
    }
    if (!palette && !pred && !cross && !meta) continue;
    const int np = palette ? (int)ncolors : pred ? (int)(st[img].pred_tx * st[img].pred_ty) : cross ? (int)(st[img].cross_tx * st[img].cross_ty)
                                                                                                   : (int)(st[img].meta_tx * st[img].meta_ty);
    const uint32_t sub_xsize = palette ? ncolors : pred ? st[img].pred_tx : cross ? st[img].cross_tx : st[img].meta_tx;
    uint32_t *d = sh.sub.delta;
    if (palette) {
      if (tid < ncolors) d[tid] = tid ? px_sub(sh.palette[tid], sh.palette[tid - 1]) : sh.palette[0];
    } else if (meta) {
      const uint32_t *gi = ext.metaimg + (size_t)img * kSmallMax;
      for (int i = (int)tid; i < np; i += 256) d[i] = 0xff000000u | (gi[i] << 8);  // the group index sits in the red and green bytes (vp8l.h:205-214)
    } else {
      const uint32_t *tiles = (pred ? modes : elems) + (size_t)img * kSmallMax;
      for (int i = (int)tid; i < np; i += 256) d[i] = tiles[i];
      if (tid == 0) { if (pred) write_predictor_head(bw, pred_bits); else write_cross_color_head(bw, cross_bits); }
    }
    __syncthreads();
    // Equal elements from each position on, at most 128 (matches are no longer): a mode image of a smooth picture is
    // nearly one value, and comparing such candidates element by element cost a thread 128 steps for each of its up to
    // 1023 candidates (5 ms per call on 64 photo-like 4K images); run by run it is one step.
    uint16_t *runlen = sh.sub.lz.next_mis;  // (the host form's array; this search does not use it otherwise)
    for (int p = (int)tid; p < np; p += 256) {
      const uint32_t x = d[p];
      int r = 1;
      while (r < (int)kMaxLen && p + r < np && d[p + r] == x) r++;
      runlen[p] = (uint16_t)r;
    }
    __syncthreads();
    for (int p = (int)tid; p < np; p += 256) {  // best match of every position, same rules as small_best_matches
      const int rem = np - p < (int)kMaxLen ? np - p : (int)kMaxLen;
      int best = 0, bestd = 0;
      const uint32_t x = d[p];
      for (int dd = 1; dd <= p; dd++) {
        if (d[p - dd] != x) continue;
        const int maxl = dd < rem ? dd : rem;
        if (maxl < best) continue;  // cannot reach the best so far; an equally long, farther one replaces it
        int l = 0;
        do {  // both sides advance by whole runs
          const int rq = runlen[p - dd + l], rp = runlen[p + l];
          l += rq < rp ? rq : rp;
        } while (l < maxl && d[p - dd + l] == d[p + l]);
        l = l < maxl ? l : maxl;
        if (l >= best) { best = l; bestd = dd; }  // farthest among equals
      }
      if (rem <= 1) best = 0;
      sh.sub.lz.len[p] = (uint16_t)best;
      sh.sub.lz.dist[p] = (uint16_t)bestd;
    }
    __syncthreads();
    if (tid == 0) small_histogram(d, np, sub_xsize, sh.sub);
    __syncthreads();
    build_codes_4waves(sh.sub.histo, sh.sub.tables, sh.rc, ds, wave, lane, standard != 0);
    if (tid == 0 && s_rc == 0) {
      bw.put(0, 1);  // no colour cache
      int rc = 0;
      for (int c = 0; c < kNumCodes && rc == 0; c++)
        rc = sh.rc[c] < 0 ? sh.rc[c] : write_code_desc(bw, sh.sub.tables.len + c * kCodeStride, alphabet_size(c), sh.sub.desc, standard != 0);
      if (rc == 0) small_write_tokens(bw, d, np, sub_xsize, sh.sub);
      s_rc = rc;
    }
    __syncthreads();
  }
  if (ext.on) {
    // the groups' codes were built and described by k_ext_codes: their descriptions follow one another (vp8l.h:217-222)
    if (tid == 0) {
      int rc = s_rc;
      const size_t grp0 = (size_t)img * kExtMaxGroups;
      for (uint32_t gk = 0; gk < ngroups * kNumCodes && rc == 0; gk++) {
        rc = ext.rc[grp0 * kNumCodes + gk];
        if (rc != 0) break;
        const uint32_t *w = ext.desc + (grp0 * kNumCodes + gk) * kExtDescWords;
        uint32_t left = ext.descbits[grp0 * kNumCodes + gk];
        for (; left >= 32; left -= 32) bw.put(*w++, 32);
        if (left) bw.put(*w, left);
      }
      bw.flush();
      st[img].hdr_bits = (uint32_t)bw.bits();
      st[img].status = rc;
    }
    return;
  }
  // the main image: its histogram into the place the sub-images' had (the loop above ends with a barrier)
  for (uint32_t i = tid; i < kNumCodes * kCodeStride; i += 256) sh.sub.histo[i] = g[i];
  __syncthreads();
  uint32_t (*dbuf)[kDescBufWords] = reinterpret_cast<uint32_t (*)[kDescBufWords]>(&sh.sub.lz);
  build_codes_4waves(sh.sub.histo, sh.sub.tables, sh.rc, ds, wave, lane, standard != 0, dbuf, sh.dbits, sh.drc);
  if (tid == 0) {
    int rc = s_rc;
    if (rc == 0) write_stream_tail(bw);
    for (int c = 0; c < kNumCodes && rc == 0; c++) {  // vp8l.h:536-540
      rc = sh.drc[c];
      const uint32_t *w = dbuf[c];
      uint32_t left = rc == 0 ? sh.dbits[c] : 0u;
      for (; left >= 32; left -= 32) bw.put(*w++, 32);
      if (left) bw.put(*w & ((1u << left) - 1u), left);
    }
    bw.flush();
    st[img].hdr_bits = (uint32_t)bw.bits();
    st[img].status = rc;
  }
  // what one occurrence of every symbol costs in bits (k_unit_bits multiplies the units' symbol counts with it)
  for (uint32_t i = tid; i < kNumCodes * kCodeStride; i += 256) {
    const uint32_t c = i / kCodeStride, sym = i % kCodeStride;
    uint32_t w = code_unwritten(&sh.sub.tables, i) ? 0u : sh.sub.tables.len[i];
    const uint32_t prefix = c == (uint32_t)kGreen && sym >= 256 ? sym - 256 : c == (uint32_t)kDist ? sym : 0u;
    if (prefix >= 4) w += (prefix - 2) >> 1;  // extra bits of a prefix symbol (lz77.cpp:89-102)
    if (sym >= (uint32_t)alphabet_size((int)c)) w = 0;  // (places of the row that no symbol has)
    sh.sub.tables.wbits[i] = (uint8_t)w;
  }
  __syncthreads();
  const uint32_t *src = reinterpret_cast<const uint32_t *>(&sh.sub.tables);
  uint32_t *dst = reinterpret_cast<uint32_t *>(&tables[img]);
  for (uint32_t i = tid; i < sizeof(CodeTables) / 4; i += 256) dst[i] = src[i];
}

// =============================================================================================
// Token bits: count per unit, scan per image, emit through an LDS staging window.
// =============================================================================================
// A match word holds length (bits 0-7), distance (8-18) and the distance code (19-29).
// Bits of one token with the code lengths alone (vp8l.h:545-583).
// `lens` has zeros for the codes that are not written (vp8l.h:552-566: a code with at most one used symbol), except
// the length prefixes, which are always written (vp8l.h:570).
__device__ __forceinline__ uint32_t token_bit_count(const uint8_t *lens, uint32_t pixel, uint32_t m) {
  const uint32_t len = m & 0xff;
  if (len == 0)
    return (uint32_t)lens[kGreen * kCodeStride + ((pixel >> 8) & 0xff)] + lens[kRed * kCodeStride + ((pixel >> 16) & 0xff)] +
           lens[kBlue * kCodeStride + (pixel & 0xff)] + lens[kAlpha * kCodeStride + (pixel >> 24)];
  uint32_t sym, nb, eb, bits;
  prefix_encode(len, sym, nb, eb);
  bits = lens[kGreen * kCodeStride + 256 + sym] + nb;
  prefix_encode((m >> 19) & 0x7ff, sym, nb, eb);
  return bits + nb + lens[kDist * kCodeStride + sym];
}

// sum over 4 adjacent lanes, valid in the last of them
__device__ __forceinline__ uint32_t quad_sum_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
  return v;
}

// Token bits per group of 16 positions (what one k_emit thread writes) and per unit.
__global__ __launch_bounds__(256) void k_bitcount(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                  const uint32_t *__restrict__ sb_first, uint32_t n,
                                                  const uint32_t *__restrict__ pix, const uint32_t *__restrict__ match,
                                                  const uint32_t *__restrict__ flags, const CodeTables *__restrict__ tables,
                                                  uint16_t *__restrict__ group_bits, uint32_t *__restrict__ unit_bits) {
  __shared__ __attribute__((aligned(16))) uint8_t lens[kNumCodes * kCodeStride];
  __shared__ uint32_t wsum[4];
  const uint32_t unit = blockIdx.x, sb = unit / kUnitsPerSuper, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  const uint32_t M = st[img].M;
  const uint32_t p0 = (unit - sb_first[img] * kUnitsPerSuper) * kUnit;
  if (p0 >= M) return;
  const CodeTables *T = &tables[img];
  for (uint32_t i = tid; i < kNumCodes * kCodeStride / 4; i += 256) {  // kCodeStride is a multiple of 4, 256 too
    const uint32_t l4 = reinterpret_cast<const uint32_t *>(T->len)[i];
    reinterpret_cast<uint32_t *>(lens)[i] = code_unwritten(T, 4 * i) ? 0u : l4;
  }
  __syncthreads();
  const uint64_t pos0 = imgs[img].pos0;
  uint32_t total = 0;
  // four consecutive positions per thread, all loads of an iteration issued together
#pragma unroll 2
  for (uint32_t k = 0; k < kUnit / 1024; k++) {
    const uint32_t p = p0 + (k * 256 + tid) * 4;
    uint32_t bits = 0;
    if (p < M) {
      const uint4 m4 = *reinterpret_cast<const uint4 *>(match + pos0 + p);
      const uint4 x4 = *reinterpret_cast<const uint4 *>(pix + pos0 + p);
      uint32_t fl = (flags[(pos0 + p) >> 5] >> ((pos0 + p) & 31)) & 15u;
      if (p + 4 > M) fl &= (1u << (M - p)) - 1u;
      if (fl & 1u) bits += token_bit_count(lens, x4.x, m4.x);
      if (fl & 2u) bits += token_bit_count(lens, x4.y, m4.y);
      if (fl & 4u) bits += token_bit_count(lens, x4.z, m4.z);
      if (fl & 8u) bits += token_bit_count(lens, x4.w, m4.w);
    }
    total += bits;
    const uint32_t g = quad_sum_u32(bits);  // positions p-12 .. p+3: one group
    if ((tid & 3) == 3 && (p - 12) < M) group_bits[(pos0 + p) >> 4] = (uint16_t)g;
  }
  const uint32_t s = wave_sum_u32(total);
  if ((tid & 63) == 0) wsum[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) unit_bits[unit] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// The same unit sums from the symbol counts that k_parse_mark left per unit (16 bits each, where the unit's advances were):
// bits of a unit = sum over the symbols of count x (code length, or nothing for a code that is not written, + the extra bits
// that follow a length or distance prefix).  A superblock per workgroup: 2.8 KB read per unit instead of 8 bytes per position.
constexpr uint32_t kUnitBitsThreads = 192;
__global__ __launch_bounds__(kUnitBitsThreads) void k_unit_bits(const EncImage *__restrict__ imgs, const EncState *__restrict__ st,
                                                                const uint32_t *__restrict__ sb_first, uint32_t n,
                                                                const uint8_t *__restrict__ unit_hist, const CodeTables *__restrict__ tables,
                                                                uint32_t *__restrict__ unit_bits) {
  constexpr uint32_t kCounters = kNumCodes * kCodeStride;
  static_assert(kCounters % 8 == 0 && kCounters / 8 <= kUnitBitsThreads, "eight counters (one 16-byte load) per thread");
  __shared__ uint32_t wsum[kUnitBitsThreads / 64];
  const uint32_t sb = blockIdx.x, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  const uint32_t M = st[img].M;
  const uint32_t p0 = (sb - sb_first[img]) * kSuper;
  if (p0 >= M) return;
  const bool mine = tid < kCounters / 8;
  uint2 w8 = make_uint2(0, 0);  // the bits of this thread's eight symbols, a byte each (CodeTables::wbits)
  if (mine) w8 = reinterpret_cast<const uint2 *>(tables[img].wbits)[tid];
  const uint64_t pos0 = imgs[img].pos0;
  for (uint32_t u = 0; u < kUnitsPerSuper && p0 + u * kUnit < M; u++) {
    uint32_t bits = 0;
    if (mine) {
      const uint4 c = reinterpret_cast<const uint4 *>(unit_hist + pos0 + p0 + u * kUnit)[tid];
      bits = (c.x & 0xffffu) * (w8.x & 0xffu) + (c.x >> 16) * ((w8.x >> 8) & 0xffu) + (c.y & 0xffffu) * ((w8.x >> 16) & 0xffu) + (c.y >> 16) * (w8.x >> 24) +
             (c.z & 0xffffu) * (w8.y & 0xffu) + (c.z >> 16) * ((w8.y >> 8) & 0xffu) + (c.w & 0xffffu) * ((w8.y >> 16) & 0xffu) + (c.w >> 16) * (w8.y >> 24);
    }
    const uint32_t sum = wave_sum_u32(bits);
    if ((tid & 63) == 0) wsum[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) unit_bits[sb * kUnitsPerSuper + u] = wsum[0] + wsum[1] + wsum[2];
    __syncthreads();
  }
}

// Exclusive scan of the unit bit counts of one image (starting at the header length).  Also clears
// the stream word in which each unit starts: k_emit ORs into those shared words and plainly stores
// all others, so no bulk clear of the output buffer is needed.
__global__ __launch_bounds__(256) void k_scan(const EncImage *__restrict__ imgs, EncState *__restrict__ st,
                                              const uint32_t *__restrict__ sb_first, const uint32_t *__restrict__ unit_bits,
                                              uint64_t *__restrict__ unit_off, uint8_t *__restrict__ out, uint32_t standard) {
  __shared__ uint32_t wsum[4];
  const uint32_t img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const EncImage im = imgs[img];
  EncState &s = st[img];
  if (s.status != 0) return;
  const uint32_t nunits = (s.M + kUnit - 1) / kUnit, unit0 = sb_first[img] * kUnitsPerSuper;
  uint32_t *out32 = reinterpret_cast<uint32_t *>(out + im.out_off);
  const uint64_t hdr_word = s.hdr_bits >> 5;
  uint64_t running = s.hdr_bits;
  // kScanPer consecutive units a thread and round (one image of 16383 x 16383 has 65528 units: one unit a thread, the 256 rounds
  // with their loads and barriers took 0.16 ms)
  constexpr uint32_t kScanPer = 8;
  for (uint32_t base = 0; base < nunits; base += 256 * kScanPer) {
    const uint32_t u0 = base + tid * kScanPer;
    uint32_t v[kScanPer];
    if (u0 + kScanPer <= nunits) {  // (unit0 is a multiple of 4 and so is u0: 16-byte loads)
#pragma unroll
      for (uint32_t j = 0; j < kScanPer; j += 4) {
        const uint4 q = *reinterpret_cast<const uint4 *>(unit_bits + unit0 + u0 + j);
        v[j] = q.x; v[j + 1] = q.y; v[j + 2] = q.z; v[j + 3] = q.w;
      }
    } else {
#pragma unroll
      for (uint32_t j = 0; j < kScanPer; j++) v[j] = u0 + j < nunits ? unit_bits[unit0 + u0 + j] : 0u;
    }
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < kScanPer; j++) mine += v[j];  // (a thread's units hold less than 2^32 bits: 8 x 4096 tokens of at most 75 bits)
    const uint32_t incl = wave_incl_scan_u32(mine, (int)lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t before = running;
    for (uint32_t k = 0; k < wave; k++) before += wsum[k];
    const uint64_t total = (uint64_t)wsum[0] + wsum[1] + wsum[2] + wsum[3];
    uint64_t off = before + incl - mine;
#pragma unroll
    for (uint32_t j = 0; j < kScanPer; j++) {
      const uint32_t u = u0 + j;
      if (u < nunits) {
        unit_off[unit0 + u] = off;
        // (inside the image's output only: a stream that does not fit fails below, and unit sizes that are not what they should
        // be must not turn into stores elsewhere)
        if (u > 0 && (off >> 5) > hdr_word && (off >> 3) + 4 <= im.out_cap) out32[off >> 5] = 0;
      }
      off += v[j];
    }
    running += total;
    __syncthreads();
  }
  if (tid == 0) {
    s.total_bits = running;
    if ((running >> 5) > hdr_word && (running >> 3) + 4 <= im.out_cap) out32[running >> 5] = 0;
    if ((running + 7) / 8 + 8 > im.out_cap) s.status = VP8L_ERR_NO_MEMORY;
    // standard dialect: the first 32 bits are the RIFF chunk size, the bytes that follow them (the reference leaves zero, vp8l.h:461)
    if (standard) out32[0] = (uint32_t)((running + 7) / 8 - 4);
  }
}

constexpr uint32_t kWinWords = 4096;  // LDS staging window: 16 KiB = 131072 stream bits

// Writer for bits that may fall partly outside the window: clips and ORs (any thread, any order).
struct StagePut {
  uint32_t *stage;
  long long rel;  // bit position relative to the window start (may be negative / beyond)
  __device__ __forceinline__ void operator()(uint32_t v, uint32_t n) {
    constexpr long long kWinBits = (long long)kWinWords * 32;
    long long r = rel;
    rel += n;
    if (n == 0 || r + (long long)n <= 0 || r >= kWinBits) return;
    unsigned long long val = v;
    int nn = (int)n;
    if (r < 0) { val >>= (int)(-r); nn += (int)r; r = 0; }
    if (r + nn > kWinBits) { nn = (int)(kWinBits - r); val &= (1ull << nn) - 1ull; }
    const uint32_t w = (uint32_t)(r >> 5), sh = (uint32_t)(r & 31);
    val <<= sh;
    atomicOr(&stage[w], (uint32_t)val);
    const uint32_t hi = (uint32_t)(val >> 32);
    if (hi) atomicOr(&stage[w + 1], hi);
  }
  __device__ __forceinline__ void finish() {}
};

// Writer for a thread whose whole bit run lies inside the window.  A thread's tokens are one
// contiguous run of bits: words the run covers completely are plain stores, only its first and
// last (partial) words are shared with the neighbouring threads and are ORed.
struct RunPut {
  uint32_t *stage;
  uint32_t w, fill;
  unsigned long long acc;
  bool first;
  __device__ __forceinline__ RunPut(uint32_t *s, uint32_t rel) : stage(s), w(rel >> 5), fill(rel & 31), acc(0), first(true) {}
  __device__ __forceinline__ void operator()(uint32_t v, uint32_t n) {
    acc |= (unsigned long long)v << fill;
    fill += n;
    if (fill >= 32) {
      if (first) { atomicOr(&stage[w], (uint32_t)acc); first = false; } else stage[w] = (uint32_t)acc;
      w++;
      acc >>= 32;
      fill -= 32;
    }
  }
  __device__ __forceinline__ void finish() { if (fill) atomicOr(&stage[w], (uint32_t)acc); }
};

// Writer for a thread's run into its private buffer, from bit 0: words beyond the buffer are dropped but counted.
constexpr uint32_t kPrivWords = 12;                  // words of a thread's private run (16 tokens of 24 bits)
constexpr uint32_t kPrivStride = kPrivWords + 1;     // odd: the threads' buffers start in different LDS banks
struct PrivPut {
  uint32_t *buf;
  uint32_t w, fill;
  unsigned long long acc;
  __device__ __forceinline__ explicit PrivPut(uint32_t *b) : buf(b), w(0), fill(0), acc(0) {}
  __device__ __forceinline__ void operator()(uint32_t v, uint32_t n) {
    acc |= (unsigned long long)v << fill;
    fill += n;
    if (fill >= 32) {
      if (w < kPrivWords) buf[w] = (uint32_t)acc;
      w++;
      acc >>= 32;
      fill -= 32;
    }
  }
  __device__ __forceinline__ void finish() { if (fill && w < kPrivWords) buf[w] = (uint32_t)acc; }
  __device__ __forceinline__ uint32_t bits() const { return w * 32 + fill; }
};

// `tab` has (0, 0) for the codes that are not written and one more (0, 0) entry at kTabZero.  Literals and back
// references differ only in where their four parts come from, so a wave with both kinds runs the lookups, the
// assembly and the write once.  A token of at most 32 bits (nearly all) goes out in one piece.
constexpr uint32_t kTabZero = kNumCodes * kCodeStride;
template <class Put>
__device__ __forceinline__ void emit_token_packed(Put &put, const uint2 *tab, uint32_t pixel, uint32_t m) {
  const uint32_t len = m & 0xff;
  uint32_t i0 = kGreen * kCodeStride + ((pixel >> 8) & 0xff), i1 = kRed * kCodeStride + ((pixel >> 16) & 0xff);
  uint32_t i2 = kBlue * kCodeStride + (pixel & 0xff), i3 = kAlpha * kCodeStride + (pixel >> 24);
  uint2 x1 = make_uint2(0, 0), x3 = make_uint2(0, 0);  // extra bits of a back reference
  if (len) {
    uint32_t sym, nb, eb;
    prefix_encode(len, sym, nb, eb);
    i0 = kGreen * kCodeStride + 256 + sym;
    x1 = make_uint2(eb, nb);
    prefix_encode((m >> 19) & 0x7ff, sym, nb, eb);
    i2 = kDist * kCodeStride + sym;
    x3 = make_uint2(eb, nb);
    i1 = i3 = kTabZero;
  }
  const uint2 e0 = tab[i0], t1 = tab[i1], e2 = tab[i2], t3 = tab[i3];
  const uint2 e1 = make_uint2(t1.x | x1.x, t1.y | x1.y), e3 = make_uint2(t3.x | x3.x, t3.y | x3.y);
  const uint32_t n1 = e0.y + e1.y, n2 = n1 + e2.y, n = n2 + e3.y;
  if (n <= 32) {
    // a part with no bits has value 0, so a shift count of 32 (taken mod 32) cannot leak anything
    put(e0.x | (e1.x << (e0.y & 31)) | (e2.x << (n1 & 31)) | (e3.x << (n2 & 31)), n);
  } else {
    put(e0.x, e0.y); put(e1.x, e1.y); put(e2.x, e2.y); put(e3.x, e3.y);
  }
}

// Bits of one token, from the same table (the lengths of emit_token_packed's four parts and the extra bits).
__device__ __forceinline__ uint32_t token_bits_packed(const uint2 *tab, uint32_t pixel, uint32_t m) {
  const uint32_t len = m & 0xff;
  uint32_t i0 = kGreen * kCodeStride + ((pixel >> 8) & 0xff), i1 = kRed * kCodeStride + ((pixel >> 16) & 0xff);
  uint32_t i2 = kBlue * kCodeStride + (pixel & 0xff), i3 = kAlpha * kCodeStride + (pixel >> 24);
  uint32_t extra = 0;
  if (len) {
    uint32_t sym, nb, eb;
    prefix_encode(len, sym, nb, eb);
    i0 = kGreen * kCodeStride + 256 + sym;
    extra = nb;
    prefix_encode((m >> 19) & 0x7ff, sym, nb, eb);
    i2 = kDist * kCodeStride + sym;
    extra += nb;
    i1 = i3 = kTabZero;
  }
  return tab[i0].y + tab[i1].y + tab[i2].y + tab[i3].y + extra;
}

// Bits of the tokens of 16 consecutive positions (what k_bitcount leaves per group when it runs).
__device__ __forceinline__ uint32_t count_group(const uint2 *tab, const uint32_t *__restrict__ px, const uint32_t *__restrict__ mt, uint32_t fl) {
  uint4 m4[4], p4[4];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    m4[g] = reinterpret_cast<const uint4 *>(mt)[g];
    p4[g] = reinterpret_cast<const uint4 *>(px)[g];
  }
  uint32_t bits = 0;
#pragma unroll
  for (int g = 0; g < 4; g++, fl >>= 4) {
    if ((fl & 15u) == 0) continue;
    if (fl & 1u) bits += token_bits_packed(tab, p4[g].x, m4[g].x);
    if (fl & 2u) bits += token_bits_packed(tab, p4[g].y, m4[g].y);
    if (fl & 4u) bits += token_bits_packed(tab, p4[g].z, m4[g].z);
    if (fl & 8u) bits += token_bits_packed(tab, p4[g].w, m4[g].w);
  }
  return bits;
}

// The tokens of 16 consecutive positions; all eight 16-byte loads are issued before the first token is written.
template <class Put>
__device__ __forceinline__ void emit_group(Put &put, const uint2 *tab, const uint32_t *__restrict__ px,
                                           const uint32_t *__restrict__ mt, uint32_t fl) {
  uint4 m4[4], p4[4];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    m4[g] = reinterpret_cast<const uint4 *>(mt)[g];
    p4[g] = reinterpret_cast<const uint4 *>(px)[g];
  }
#pragma unroll
  for (int g = 0; g < 4; g++, fl >>= 4) {
    if ((fl & 15u) == 0) continue;
    if (fl & 1u) emit_token_packed(put, tab, p4[g].x, m4[g].x);
    if (fl & 2u) emit_token_packed(put, tab, p4[g].y, m4[g].y);
    if (fl & 4u) emit_token_packed(put, tab, p4[g].z, m4[g].z);
    if (fl & 8u) emit_token_packed(put, tab, p4[g].w, m4[g].w);
  }
  put.finish();
}

// kPrivate: no bit counts per 16 positions exist (group_bits is null) and a thread does not count its tokens first either: it
// writes its run into a private LDS buffer from bit 0, which tells its length; after the scan of the lengths the run is
// shifted into the window (five instructions per word instead of twenty-five per token to count).  A run longer than the
// buffer, or one that crosses a window edge, is written from the tokens again as in the other form.
template <bool kPrivate>
__global__ __launch_bounds__(256) void k_emit(const EncImage *__restrict__ imgs, EncState *__restrict__ st,
                                              const uint32_t *__restrict__ sb_first, uint32_t n,
                                              const uint32_t *__restrict__ pix, const uint32_t *__restrict__ match,
                                              const uint32_t *__restrict__ flags, const CodeTables *__restrict__ tables,
                                              const uint16_t *__restrict__ group_bits, const uint32_t *__restrict__ unit_bits,
                                              const uint64_t *__restrict__ unit_off, uint8_t *__restrict__ out) {
  __shared__ uint2 tab[kNumCodes * kCodeStride + 1];  // (code, length), and a zero entry
  __shared__ uint32_t stage[kWinWords + 2];
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t priv[kPrivate ? 256 * kPrivStride : 1];
  // one superblock (four units) per workgroup: the code tables are fetched into LDS once for 16384 positions
  const uint32_t sb = blockIdx.x, img = find_image(sb_first, n, sb), tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = tid >> 6;
  if (st[img].status != 0) return;
  const uint32_t M = st[img].M;
  if ((sb - sb_first[img]) * kSuper >= M) return;
  const CodeTables *T = &tables[img];
  for (uint32_t i = tid; i < kNumCodes * kCodeStride / 4; i += 256) {  // four (code, length) entries at a time
    const uint4 c4 = reinterpret_cast<const uint4 *>(T->code)[i];
    const uint32_t l4 = reinterpret_cast<const uint32_t *>(T->len)[i];
    if (i == 0) tab[kTabZero] = make_uint2(0, 0);
    const bool off = code_unwritten(T, 4 * i);  // kCodeStride is a multiple of 4: the four entries belong to one code part
    tab[4 * i + 0] = off ? make_uint2(0, 0) : make_uint2(c4.x, l4 & 0xff);
    tab[4 * i + 1] = off ? make_uint2(0, 0) : make_uint2(c4.y, (l4 >> 8) & 0xff);
    tab[4 * i + 2] = off ? make_uint2(0, 0) : make_uint2(c4.z, (l4 >> 16) & 0xff);
    tab[4 * i + 3] = off ? make_uint2(0, 0) : make_uint2(c4.w, l4 >> 24);
  }
  const EncImage im = imgs[img];
  const uint64_t pos0 = im.pos0;
  uint32_t *out32 = reinterpret_cast<uint32_t *>(out + im.out_off);
  if (kPrivate || !group_bits) __syncthreads();  // the tables are in LDS: the threads count (or write) their tokens with them
  for (uint32_t u = 0; u < kUnitsPerSuper; u++) {
  const uint32_t unit = sb * kUnitsPerSuper + u;
  const uint32_t p0 = (sb - sb_first[img]) * kSuper + u * kUnit;
  if (p0 >= M) break;
  // this thread's 16 consecutive positions (the arrays have readable slack past M)
  const uint32_t pb = p0 + tid * 16;
  const uint32_t *mt = match + pos0 + pb, *px = pix + pos0 + pb;
  uint32_t fl = 0, mybits = 0;
  if (pb < M) {
    fl = (flags[(pos0 + pb) >> 5] >> ((pos0 + pb) & 31)) & 0xffffu;
    if (pb + 16 > M) fl &= (1u << (M - pb)) - 1u;
    // group_bits: what k_bitcount counted for these 16 positions; null: no such pass was made (the units' sums come from the
    // symbol counts, k_unit_bits), and the thread counts its own tokens here (it loads them again to write them: holding them
    // in registers over the barriers in between made the kernel half as fast again)
    if (kPrivate) {
      PrivPut put(priv + tid * kPrivStride);
      emit_group(put, tab, px, mt, fl);
      mybits = put.bits();
    } else {
      mybits = group_bits ? group_bits[(pos0 + pb) >> 4] : count_group(tab, px, mt, fl);
    }
  }
  const uint32_t incl = wave_incl_scan_u32(mybits, (int)lane);
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();  // also: the tables are in LDS, the previous unit's words are out
  uint32_t before = 0;
  for (uint32_t k = 0; k < wave; k++) before += wsum[k];
  const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();  // wsum is read before the next unit overwrites it
  // The unit's place in the stream was made for unit_bits[unit] bits, counted another way (k_unit_bits from the symbol counts, or
  // k_bitcount): they are the same number by construction, and writing any other number of bits would run into the neighbours
  // or past the buffer - so nothing is written then and the image fails.
  if (total != unit_bits[unit]) {
    if (tid == 0) st[img].status = VP8L_ERR_HIP;
    continue;
  }
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
      if (kPrivate && rel >= 0 && rel + (long long)mybits <= kWinBits && mybits <= kPrivWords * 32) {
        // the private run, shifted to its place: the first and the last word are shared with the neighbours' runs
        const uint32_t *buf = priv + tid * kPrivStride;
        const uint32_t sh = (uint32_t)rel & 31u, w0 = (uint32_t)rel >> 5, nw = (mybits + 31) >> 5, nout = (sh + mybits + 31) >> 5;
        uint32_t prev = 0;
        for (uint32_t j = 0; j < nout; j++) {
          const uint32_t cur = j < nw ? buf[j] : 0u;
          const uint32_t o = sh ? (cur << sh) | (prev >> (32 - sh)) : cur;
          if (j == 0 || j == nout - 1) atomicOr(&stage[w0 + j], o); else stage[w0 + j] = o;
          prev = cur;
        }
      } else if (rel >= 0 && rel + (long long)mybits <= kWinBits) {
        RunPut put(stage, (uint32_t)rel);
        emit_group(put, tab, px, mt, fl);
      } else {  // the run crosses a window edge
        StagePut put{stage, rel};
        emit_group(put, tab, px, mt, fl);
      }
    }
    __syncthreads();
    for (uint32_t i = tid; i < cntw; i += 256) {
      const uint64_t gw = first_word + wbase + i;
      // a word is shared with the neighbouring unit (or the header) when a unit boundary falls inside it
      const bool shared = (gw == first_word && (start_bit & 31)) || (gw == last_word && (end_bit & 31));
      if (shared) atomicOr(&out32[gw], stage[i]); else out32[gw] = stage[i];
    }
    __syncthreads();
  }
  }
}

}  // namespace vp8l
#include "vp8l_encode_ext.h"
namespace vp8l {

// =============================================================================================
// Synthetic images on the device: x(k) = k-th state of the 32-bit LCG reached by jumping ahead.
// =============================================================================================
__device__ __forceinline__ uint32_t clamp_u8(int v) { return v < 0 ? 0u : v > 255 ? 255u : (uint32_t)v; }

__device__ __forceinline__ uint32_t lcg_state(uint32_t seed, uint64_t k) {  // state after k steps
  uint32_t ra = 1, rc = 0, ba = 1664525u, bc = 1013904223u;
  while (k) {
    if (k & 1) { rc = ba * rc + bc; ra = ba * ra; }
    bc = ba * bc + bc;
    ba = ba * ba;
    k >>= 1;
  }
  return ra * seed + rc;
}

__global__ __launch_bounds__(256) void k_generate(int mode, uint32_t w, uint32_t h, uint32_t seed, uint32_t ncolors,
                                                  uint32_t *__restrict__ out) {
  __shared__ uint32_t pal[256];
  if (mode == 5) {
    if (threadIdx.x < ncolors) pal[threadIdx.x] = 0xff000000u | ((lcg_state(777u, threadIdx.x + 1) >> 8) & 0xffffffu);
    __syncthreads();
  }
  const uint64_t npx = (uint64_t)w * h;
  const uint32_t dw = w > 1 ? w - 1 : 1, dh = h > 1 ? h - 1 : 1;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (uint64_t)gridDim.x * 256) {
    const uint32_t x = (uint32_t)(i % w), y = (uint32_t)(i / w);
    uint32_t p;
    if (mode == 3) {
      uint32_t s = lcg_state(seed, 6 * i);
      int nz[3];
      for (int c = 0; c < 3; c++) {
        s = s * 1664525u + 1013904223u; const uint32_t a = s >> 8;
        s = s * 1664525u + 1013904223u; const uint32_t b = s >> 8;
        nz[c] = (int)(a & 3) + (int)(b & 3) - 3;
      }
      const int L = (int)(x * 160 / dw + y * 80 / dh);
      p = 0xff000000u | (clamp_u8(L + 10 + nz[0]) << 16) | (clamp_u8(L + nz[1]) << 8) | clamp_u8(L * 3 / 4 + 5 + nz[2]);
    } else if (mode == 4) {
      const uint32_t bs = ((y / 16) * 7919u + (x / 16) * 104729u) * 2654435761u;
      p = 0xff000000u | ((bs >> 8) & 0xf0f0f0u);
      if (((lcg_state(seed, i + 1) >> 8) & 31) == 0) p ^= 0x00010101u;
    } else {
      const uint32_t l = lcg_state(seed, i + 1) >> 8;
      p = pal[((x / 7) * 13 + (y / 5) * 31 + ((l & 15) == 0)) % ncolors];
    }
    out[i] = p;
  }
}

}  // namespace vp8l

// =============================================================================================
// Host side: batch encoder context
// =============================================================================================
using namespace vp8l;

struct vp8l_encoder {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[10] = {};
  std::mutex mu;
  // workspace (grown on demand)
  DeviceBuffer d_imgs, d_state, d_sbfirst, d_gsets, d_pix, d_match, d_flags, d_histo, d_tables, d_sx, d_entry,
      d_unit_bits, d_unit_off, d_dense, d_group_bits, d_adv, d_pix2, d_modes, d_elems;
  int match_mode = 0;  // 0 = candidate lists with the other forms behind them, 1 = exhaustive only, 2 = the candidate lists give up on every tile, 3 = 2 + grams, 4 = 0 + search along the parse
                       // (verification of what comes behind them), 3 = the same with k_lz77_grams between k_lz77_dominant and the exhaustive kernel
  const uint32_t *last_pix = nullptr;  // coded pixels of the last call (for fetch_stage)
  bool last_pred = false;
  bool last_cross = false;
  int cross_bits = 0;  // 0 = no cross-colour transform, else requested tile bits 2..9
  int cache_bits = 0;   // 0 = no colour cache, else 1..11 (vp8l.h:230-241)
  int meta_bits = 0;    // 0 = one Huffman group, else the requested tile bits 2..9 of the entropy image
  int meta_groups = 4;  // Huffman groups when meta_bits is set (2..kExtMaxGroups)
  DeviceBuffer x_hit, x_meta, x_feat, x_histo, x_tab, x_codes, x_desc, x_descbits, x_rc, x_rows;  // the extended coder's workspace
  bool standard = false;  // write the standard dialect (what libwebp reads) instead of the reference's
  int pred_bits = 0;   // 0 = no predictor transform (the reference encoder's output), else requested tile bits 2..9
  // last batch
  std::vector<EncImage> imgs;
  std::vector<EncState> states;
  std::vector<uint32_t> sb_first;
  uint8_t *last_out = nullptr;
  double last_seconds = 0;
  double stage_seconds[9] = {};
};

// Heuristics of the candidate-list kernel (when a tile goes to the exhaustive kernel instead); VP8L_TUNE="mean_bucket,budget,max_bucket,dense"
// overrides them for measurements (dense: bit 0 = k_lz77_dominant runs, bit 1 = k_lz77_grams too, bit 2 = k_lz77_walk too, bit 3 = the
// bits of the units are counted by a pass over pixels and matches, k_bitcount, as they used to be, bit 4 = a thread of k_emit counts its
// tokens before it writes them instead of writing them to a private buffer first; every setting gives the same streams).
static uint32_t g_tune[4] = {kSpMaxMeanBucket, kSpBudget, kSpMaxBucket, 1};
static const bool g_tune_read = [] {
  const char *v = getenv("VP8L_TUNE");
  if (v) sscanf(v, "%u,%u,%u,%u", &g_tune[0], &g_tune[1], &g_tune[2], &g_tune[3]);
  return true;
}();

// The device arrays one encode call works in (a context's grow-only workspace, or parts of the static scratch array).
struct EncBuffers {
  EncImage *imgs;
  EncState *st;
  uint32_t *sbf;
  unsigned long long *gsets;
  uint32_t *pix_t;  // output of the pixel transform
  uint32_t *pix;    // what the match search and the coder read (pix_t unless a predictor / cross-colour pass follows)
  uint32_t *modes, *elems;
  uint32_t *match, *flags, *histo;
  CodeTables *tables;
  uint8_t *sx, *entry;
  uint32_t *ubits;
  uint64_t *uoff;
  uint8_t *dense;
  uint16_t *gbits;
  uint8_t *adv8;
};

// Clears what must start at zero and launches every kernel of the encode path for nl images that occupy sb_total
// superblocks; imgs and sbf are already on their way on the same stream.  ev: ten events for stage times, or null.
static int launch_encode_kernels(hipStream_t s, const EncBuffers &B, uint32_t nl, uint32_t sb_total, uint32_t pred, uint32_t cross,
                                 int match_mode, uint8_t *d_out, hipEvent_t *ev, uint32_t standard = 0, const ExtArgs *extp = nullptr,
                                 uint32_t meta_bits = 0, uint32_t max_sb_per_image = 0) {
  // max_sb_per_image: the most superblocks any image of the call has (0: not known; picks the form of the parse's chain walk)
  // match_mode 2: the candidate lists refuse every tile (no bucket may hold anything), so that the forms behind them see all tiles
  const uint32_t tune_max_bucket = (match_mode == 2 || match_mode == 3) ? 0u : g_tune[2];
  const bool with_grams = match_mode == 3 || (g_tune[3] & 2u);
  const bool with_walk = match_mode == 4 || (g_tune[3] & 4u);  // match_mode 4: as 0, with the search along the parse for palette images
  if (match_mode >= 2) match_mode = 0;
  bool h = true;
  h = h && hip_ok(hipMemsetAsync(B.st, 0, nl * sizeof(EncState), s), "clear state");
  h = h && hip_ok(hipMemsetAsync(B.gsets, 0xff, (size_t)nl * kSetSlots * sizeof(unsigned long long), s), "clear sets");
  h = h && hip_ok(hipMemsetAsync(B.histo, 0, (size_t)nl * kNumCodes * kCodeStride * 4, s), "clear histograms");
  h = h && hip_ok(hipMemsetAsync(B.dense, 0, (size_t)sb_total * (kSuper / kLzTile) + 16 + kDenseTail, s), "clear tile flags");  // and the work counter behind them
  if (!h) return VP8L_ERR_HIP;
  auto mark = [&](int k) { if (ev) (void)hipEventRecord(ev[k], s); };
  const uint32_t gsb = sb_total, gunits = sb_total * kUnitsPerSuper;
  mark(0);
  hipLaunchKernelGGL(k_palette_probe, dim3(nl), dim3(256), 0, s, B.imgs, B.st);
  hipLaunchKernelGGL(k_palette_scan, dim3(gsb), dim3(256), 0, s, B.imgs, B.st, B.sbf, nl, B.gsets);
  hipLaunchKernelGGL(k_palette_finalize, dim3(nl), dim3(256), 0, s, B.imgs, B.st, B.gsets, (match_mode == 0 && !pred && !cross) ? 1u : 0u, pred, cross, meta_bits | (meta_bits && standard ? 16u : 0u));
  mark(1);
  hipLaunchKernelGGL(k_transform, dim3(gsb), dim3(256), 0, s, B.imgs, B.st, B.sbf, nl, B.pix_t);
  if (pred) {
    hipLaunchKernelGGL(k_pred_modes, dim3(nl * kSmallMax), dim3(256), 0, s, B.imgs, B.st, B.modes, standard);
    hipLaunchKernelGGL(k_pred_residual, dim3(gunits), dim3(256), 0, s, B.imgs, B.st, B.sbf, nl, B.pix_t, B.modes, B.pix, standard);
  }
  if (cross) {  // on what the predictor left (in place), or on the subtract-green image
    const uint32_t *from = pred ? B.pix : B.pix_t;
    hipLaunchKernelGGL(k_cross_elems, dim3(nl * kSmallMax), dim3(64), 0, s, B.imgs, B.st, from, B.elems);
    hipLaunchKernelGGL(k_cross_apply, dim3(gunits), dim3(256), 0, s, B.imgs, B.st, B.sbf, nl, from, B.elems, B.pix);
  }
  mark(2);
  if (match_mode == 0)
  {
    uint32_t *dm_counter = reinterpret_cast<uint32_t *>(B.dense + (((size_t)sb_total * (kSuper / kLzTile) + 15) & ~(size_t)15));
    const uint32_t dm_grid = std::min<uint32_t>(gsb * (kSuper / kSpTile), kDmGrid);
    // After the predictor the planes are mostly one value in runs, which the candidate lists give up on after spending their budget:
    // there the kernel made for such tiles looks at every tile first and leaves the candidate lists what suits them.
    const bool dominant_first = pred != 0 && (g_tune[3] & 1u);
    if (dominant_first)
      hipLaunchKernelGGL(k_lz77_dominant, dim3(dm_grid), dim3(kDmThreads), 0, s, B.imgs, B.st, B.sbf, nl, gsb * (kSuper / kSpTile), B.pix, B.match,
                         B.adv8, B.dense, dm_counter, 0u, (const uint32_t *)nullptr);
    hipLaunchKernelGGL(k_lz77_sparse, dim3((gsb * (kSuper / kSpTile) + 7) / 8 * 8), dim3(kSpThreads), 0, s, B.imgs, B.st, B.sbf, nl,
                       gsb * (kSuper / kSpTile), B.pix, B.match, B.adv8, B.dense, g_tune[0] | (dominant_first ? 0x80000000u : 0u), g_tune[1], tune_max_bucket, dm_counter + 2);
    hipLaunchKernelGGL(k_lz77_pairs, dim3(std::min<uint32_t>(gsb * (kSuper / kSpTile), kSpPairsGrid)), dim3(kSpThreads), 0, s, B.imgs, B.st, B.sbf, nl,
                       gsb * (kSuper / kSpTile), B.pix, B.match, B.adv8, B.dense, g_tune[0], g_tune[1], tune_max_bucket, dm_counter + 2);
    if (g_tune[3] & 1u)  // what the candidate lists gave up on (after a first look: what that left them and they could not do either)
      hipLaunchKernelGGL(k_lz77_dominant, dim3(dm_grid), dim3(kDmThreads), 0, s, B.imgs, B.st, B.sbf, nl, gsb * (kSuper / kSpTile), B.pix, B.match,
                         B.adv8, B.dense, dm_counter + 1, (uint32_t)kTileExhaustive, (const uint32_t *)(dm_counter + 2));
    // ... and what has no frequent value either: exact, but two to four times slower than the exhaustive kernel on the bundled palettes
    // it was made for (its lists are chains of dependent reads), so only on request
    if (with_grams)
      hipLaunchKernelGGL(k_lz77_grams, dim3(dm_grid), dim3(kDmThreads), 0, s, B.imgs, B.st, B.sbf, nl, gsb * (kSuper / kSpTile), B.pix, B.match,
                         B.adv8, B.dense, dm_counter + 3, (const uint32_t *)(dm_counter + 2));
    // ... and of palette images, what is still waiting, only where a parse can get to: exact; a fifth faster than the exhaustive kernel
    // on 16-colour images (two pixels to a byte), several times slower where a tile begins inside a flat or periodic stretch (the
    // chains from its 128 possible entries then never join), so only on request as well
    if (with_walk)
      hipLaunchKernelGGL(k_lz77_walk, dim3(std::min<uint32_t>(gsb * (kSuper / kSpTile), kWkGrid)), dim3(kWkThreads), 0, s, B.imgs, B.st, B.sbf, nl,
                         gsb * (kSuper / kSpTile), B.pix, B.match, B.adv8, B.dense, dm_counter + 4, (const uint32_t *)(dm_counter + 2));
  }
  mark(9);  // also when the candidate-list kernel is switched off
  {
    const uint32_t *given_up = match_mode == 0 ? reinterpret_cast<const uint32_t *>(B.dense + (((size_t)sb_total * (kSuper / kLzTile) + 15) & ~(size_t)15)) + 2 : nullptr;
    hipLaunchKernelGGL(k_lz77, dim3(gsb * (kSuper / kLzBlock)), dim3(256), 0, s, B.imgs, B.st, B.sbf, nl, B.pix, B.match, B.adv8,
                       match_mode == 0 ? B.dense : (const uint8_t *)nullptr, given_up);
    if (given_up)
      hipLaunchKernelGGL(k_lz77_pieces, dim3(kFewTiles), dim3(64 * (kLzTile / kLzPiece)), 0, s, B.imgs, B.st, B.sbf, nl, B.pix, B.match, B.adv8,
                         (const uint8_t *)B.dense, given_up);
  }
  hipLaunchKernelGGL(k_lz77_head, dim3(nl * (kLzHead / kLzHeadPiece)), dim3(64), 0, s, B.imgs, B.st, B.sbf, nl, B.pix, B.match, B.adv8,
                     match_mode == 0 ? B.dense : (const uint8_t *)nullptr);
  mark(3);
  hipLaunchKernelGGL(k_parse_exits, dim3(gsb), dim3(128), 0, s, B.imgs, B.st, B.sbf, nl, B.adv8, B.sx);
  if (max_sb_per_image > kChainSerialMax && max_sb_per_image <= kChainMaxRounds * kChainRound && nl <= 65535u) {  // (a grid's second dimension)
    // the maps of the rounds lie where the token-start bits will be (k_parse_mark writes those after this, every word of them)
    uint8_t *rmap = reinterpret_cast<uint8_t *>(B.flags);
    const dim3 grid((max_sb_per_image + kChainRound - 1) / kChainRound, nl);
    hipLaunchKernelGGL(k_parse_chain_maps, grid, dim3(256), 0, s, B.st, B.sbf, B.sx, rmap);
    hipLaunchKernelGGL(k_parse_chain_entries, grid, dim3(256), 0, s, B.st, B.sbf, B.sx, (const uint8_t *)rmap, B.entry);
  } else {
    hipLaunchKernelGGL(k_parse_chain, dim3(nl), dim3(256), 0, s, B.st, B.sbf, B.sx, B.entry);
  }
  const bool ext_on = extp && extp->on;
  // the units' bit sums from the symbol counts of the units (k_unit_bits) instead of a pass over pixels and matches (k_bitcount);
  // VP8L_TUNE's fourth value, bit 3: the pass
  const bool unit_counts = !ext_on && !(g_tune[3] & 8u);
  hipLaunchKernelGGL(k_parse_mark, dim3(gsb), dim3(kMarkThreads), 0, s, B.imgs, B.st, B.sbf, nl, B.adv8, B.entry, B.flags, B.pix, B.match,
                     ext_on ? (uint32_t *)nullptr : B.histo, unit_counts ? B.adv8 : (uint8_t *)nullptr);
  mark(4);
  ExtArgs ext;
  memset(&ext, 0, sizeof(ext));
  if (extp) ext = *extp;
  ext.standard = standard;
  if (ext.on) {
    // the coder with a colour cache and / or several Huffman groups (vp8l_encode_ext.h); stage times: "histogram" = everything up
    // to the codes, the others as in the default coder
    uint32_t *hitbits = const_cast<uint32_t *>(ext.hitbits);
    if (ext.cache_bits) {
      hipLaunchKernelGGL(k_cache_rows, dim3(gsb), dim3(256), 0, s, B.imgs, B.st, B.sbf, nl, B.pix, ext);
      hipLaunchKernelGGL(k_cache_scan, dim3(nl), dim3(256), 0, s, B.st, B.sbf, ext);
      hipLaunchKernelGGL(k_cache_hits, dim3(gsb), dim3(256), 0, s, B.imgs, B.st, B.sbf, nl, B.pix, B.match, B.flags, hitbits, ext);
    }
    if (ext.groups > 1) {
      hipLaunchKernelGGL(k_meta_feature, dim3(nl * kSmallMax), dim3(64), 0, s, B.imgs, B.st, B.pix, ext);
      hipLaunchKernelGGL(k_meta_assign, dim3(nl), dim3(256), 0, s, B.st, ext);
    }
    if (!hip_ok(hipMemsetAsync(ext.histo, 0, (size_t)nl * kExtMaxGroups * kExtSyms * 4, s), "clear group histograms")) return VP8L_ERR_HIP;
    hipLaunchKernelGGL(k_ext_histogram, dim3(gsb), dim3(256), 0, s, B.imgs, B.st, B.sbf, nl, B.pix, B.match, B.flags, ext);
    hipLaunchKernelGGL(k_ext_codes, dim3(nl * kExtMaxGroups * kNumCodes), dim3(64), 0, s, B.st, ext);
    mark(5);
    hipLaunchKernelGGL(k_entropy_header, dim3(nl), dim3(256), 0, s, B.imgs, B.st, B.histo, B.tables, B.modes, B.elems, d_out, standard, ext);
    mark(6);
    hipLaunchKernelGGL(k_ext_bitcount, dim3(gunits), dim3(256), 0, s, B.imgs, B.st, B.sbf, nl, B.pix, B.match, B.flags, B.gbits, B.ubits, ext);
    hipLaunchKernelGGL(k_scan, dim3(nl), dim3(256), 0, s, B.imgs, B.st, B.sbf, B.ubits, B.uoff, d_out, standard);
    mark(7);
    hipLaunchKernelGGL(k_ext_emit, dim3(gsb), dim3(256), 0, s, B.imgs, B.st, B.sbf, nl, B.pix, B.match, B.flags, B.gbits, B.uoff, d_out, ext);
  } else {
  mark(5);  // the symbols were counted by k_parse_mark
  hipLaunchKernelGGL(k_entropy_header, dim3(nl), dim3(256), 0, s, B.imgs, B.st, B.histo, B.tables, B.modes, B.elems, d_out, standard, ext);
  mark(6);
  if (unit_counts) hipLaunchKernelGGL(k_unit_bits, dim3(gsb), dim3(kUnitBitsThreads), 0, s, B.imgs, B.st, B.sbf, nl, (const uint8_t *)B.adv8, B.tables, B.ubits);
  else hipLaunchKernelGGL(k_bitcount, dim3(gunits), dim3(256), 0, s, B.imgs, B.st, B.sbf, nl, B.pix, B.match, B.flags, B.tables, B.gbits, B.ubits);
  hipLaunchKernelGGL(k_scan, dim3(nl), dim3(256), 0, s, B.imgs, B.st, B.sbf, B.ubits, B.uoff, d_out, standard);
  mark(7);
  if (unit_counts && !(g_tune[3] & 16u))
    hipLaunchKernelGGL(k_emit<true>, dim3(gsb), dim3(256), 0, s, B.imgs, B.st, B.sbf, nl, B.pix, B.match, B.flags, B.tables,
                       (const uint16_t *)nullptr, (const uint32_t *)B.ubits, B.uoff, d_out);
  else
    hipLaunchKernelGGL(k_emit<false>, dim3(gsb), dim3(256), 0, s, B.imgs, B.st, B.sbf, nl, B.pix, B.match, B.flags, B.tables,
                       unit_counts ? (const uint16_t *)nullptr : (const uint16_t *)B.gbits, (const uint32_t *)B.ubits, B.uoff, d_out);
  }
  mark(8);
  if (!hip_ok(hipGetLastError(), "kernel launch")) return VP8L_ERR_HIP;
  return VP8L_OK;
}

extern "C" size_t vp8l_encode_bound(uint32_t w, uint32_t h) {
  // a Huffman code never spends more than a fixed-length code would: <= 9+8+8+8 bits per literal,
  // <= 9+5+6+9 per match, at most one token per pixel; plus headers and the palette sub-image
  const uint64_t px = (uint64_t)w * h;
  return (size_t)(((px * 33 + 7) / 8 + 16384 + 255) & ~(uint64_t)255);
}

extern "C" int vp8l_encoder_create(int device, vp8l_encoder **out) {
  if (!out) return VP8L_ERR_BAD_ARGUMENT;
  int rc = select_device(device, &device);
  if (rc != VP8L_OK) return rc;
  vp8l_encoder *e = new (std::nothrow) vp8l_encoder;
  if (!e) return VP8L_ERR_NO_MEMORY;
  e->device = device;
  if (!hip_ok(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking), "hipStreamCreate")) { delete e; return VP8L_ERR_HIP; }
  for (auto &ev : e->ev)
    if (!hip_ok(hipEventCreate(&ev), "hipEventCreate")) { vp8l_encoder_destroy(e); return VP8L_ERR_HIP; }
  *out = e;
  return VP8L_OK;
}

extern "C" void vp8l_encoder_destroy(vp8l_encoder *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  for (auto &ev : e->ev) if (ev) (void)hipEventDestroy(ev);
  DeviceBuffer *bufs[] = {&e->d_imgs, &e->d_state, &e->d_sbfirst, &e->d_gsets, &e->d_pix, &e->d_match, &e->d_flags,
                          &e->d_histo, &e->d_tables, &e->d_sx, &e->d_entry, &e->d_unit_bits, &e->d_unit_off, &e->d_dense, &e->d_group_bits, &e->d_adv, &e->d_pix2, &e->d_modes, &e->d_elems,
                          &e->x_hit, &e->x_meta, &e->x_feat, &e->x_histo, &e->x_tab, &e->x_codes, &e->x_desc, &e->x_descbits, &e->x_rc, &e->x_rows};
  for (auto *b : bufs) b->release();
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

static int vp8l_encoder_encode_device_impl(vp8l_encoder *e, size_t n, const uint32_t *const *d_argb, const uint32_t *width,
                                          const uint32_t *height, uint8_t *d_out, size_t out_cap, size_t *out_offset,
                                          size_t *out_len, int *status) {
  if (!e || !d_argb || !width || !height || !d_out || !out_offset || !out_len || !status) return VP8L_ERR_BAD_ARGUMENT;
  if (n == 0) return VP8L_OK;
  if (n > 0x7fffffffu) return VP8L_ERR_BAD_ARGUMENT;
  if (((uintptr_t)d_out & 255) != 0) return VP8L_ERR_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lock(e->mu);
  if (!hip_ok(hipSetDevice(e->device), "hipSetDevice")) return VP8L_ERR_HIP;

  // ---- layout (host) -----------------------------------------------------------------------
  e->imgs.assign(n, EncImage{});
  e->sb_first.assign(n + 1, 0);
  e->states.assign(n, EncState{});
  uint64_t sb_total = 0, out_used = 0;
  std::vector<uint32_t> live;  // images that passed the argument checks (vp8l.h:587-590)
  for (size_t i = 0; i < n; i++) {
    status[i] = VP8L_OK;
    out_offset[i] = 0;
    out_len[i] = 0;
    if (!d_argb[i] || width[i] == 0 || height[i] == 0) { status[i] = VP8L_ERR_INVALID_ARGB; continue; }
    if (width[i] > kMaxDim || height[i] > kMaxDim) { status[i] = VP8L_ERR_TOO_BIG_ARGB; continue; }
    live.push_back((uint32_t)i);
  }
  const uint32_t nl = (uint32_t)live.size();
  if (nl == 0) return VP8L_OK;
  uint64_t max_sb = 0;
  for (uint32_t k = 0; k < nl; k++) {
    const size_t i = live[k];
    const uint64_t px = (uint64_t)width[i] * height[i];
    EncImage &im = e->imgs[k];
    im.argb = d_argb[i];
    im.w = width[i];
    im.h = height[i];
    im.pos0 = sb_total * kSuper;
    im.out_off = out_used;
    im.out_cap = vp8l_encode_bound(width[i], height[i]);
    e->sb_first[k] = (uint32_t)sb_total;
    sb_total += (px + kTailPad + kSuper - 1) / kSuper;
    max_sb = std::max<uint64_t>(max_sb, (px + kSuper - 1) / kSuper);  // superblocks that hold pixels (coded pixels are no more)
    out_used += im.out_cap;
    if (out_used > out_cap) return VP8L_ERR_NO_MEMORY;
    if (sb_total > 0x3fffffffu) return VP8L_ERR_NO_MEMORY;
    out_offset[i] = (size_t)im.out_off;
  }
  e->sb_first[nl] = (uint32_t)sb_total;
  int rc_launch = VP8L_OK;
  const uint64_t npos = sb_total * kSuper;
  const uint64_t nunits = sb_total * kUnitsPerSuper;

  // ---- workspace ---------------------------------------------------------------------------
  bool ok = e->d_imgs.reserve(nl * sizeof(EncImage)) && e->d_state.reserve(nl * sizeof(EncState)) &&
            e->d_sbfirst.reserve((nl + 1) * sizeof(uint32_t)) &&
            e->d_gsets.reserve((size_t)nl * kSetSlots * sizeof(unsigned long long)) &&
            e->d_pix.reserve(npos * 4) && e->d_match.reserve(npos * 4) && e->d_flags.reserve(npos / 8) &&
            e->d_histo.reserve((size_t)nl * kNumCodes * kCodeStride * 4) && e->d_tables.reserve((size_t)nl * sizeof(CodeTables)) &&
            e->d_sx.reserve(sb_total * kBlockLen) && e->d_entry.reserve(sb_total) &&
            e->d_unit_bits.reserve(nunits * 4) && e->d_unit_off.reserve(nunits * 8) &&
            e->d_dense.reserve(sb_total * (kSuper / kLzTile) + 16 + kDenseTail) && e->d_group_bits.reserve(npos / 16 * 2) && e->d_adv.reserve(npos) &&
            ((e->pred_bits == 0 && e->cross_bits == 0) ||
             (e->d_pix2.reserve(npos * 4) && e->d_modes.reserve((size_t)nl * kSmallMax * 4) && e->d_elems.reserve((size_t)nl * kSmallMax * 4)));
  if (!ok) return VP8L_ERR_NO_MEMORY;

  hipStream_t s = e->stream;
  const uint32_t pred = (uint32_t)e->pred_bits, cross = (uint32_t)e->cross_bits;
  EncBuffers B;
  B.imgs = e->d_imgs.as<EncImage>();
  B.st = e->d_state.as<EncState>();
  B.sbf = e->d_sbfirst.as<uint32_t>();
  B.gsets = e->d_gsets.as<unsigned long long>();
  B.pix_t = e->d_pix.as<uint32_t>();
  B.pix = (pred || cross) ? e->d_pix2.as<uint32_t>() : B.pix_t;
  B.modes = (pred || cross) ? e->d_modes.as<uint32_t>() : nullptr;
  B.elems = (pred || cross) ? e->d_elems.as<uint32_t>() : nullptr;
  B.match = e->d_match.as<uint32_t>();
  B.flags = e->d_flags.as<uint32_t>();
  B.histo = e->d_histo.as<uint32_t>();
  B.tables = e->d_tables.as<CodeTables>();
  B.sx = e->d_sx.as<uint8_t>();
  B.entry = e->d_entry.as<uint8_t>();
  B.ubits = e->d_unit_bits.as<uint32_t>();
  B.uoff = e->d_unit_off.as<uint64_t>();
  B.dense = e->d_dense.as<uint8_t>();
  B.gbits = e->d_group_bits.as<uint16_t>();
  B.adv8 = e->d_adv.as<uint8_t>();
  auto *st = B.st;
  auto *pix = B.pix;

  bool h = true;
  h = h && hip_ok(hipMemcpyAsync(B.imgs, e->imgs.data(), nl * sizeof(EncImage), hipMemcpyHostToDevice, s), "upload images");
  h = h && hip_ok(hipMemcpyAsync(B.sbf, e->sb_first.data(), (nl + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s), "upload table");
  if (!h) return VP8L_ERR_HIP;
  ExtArgs ext;
  memset(&ext, 0, sizeof(ext));
  if (e->cache_bits || e->meta_bits) {
    const size_t groups = (size_t)nl * kExtMaxGroups;
    if (!e->x_hit.reserve(npos / 8 + 256) || !e->x_meta.reserve((size_t)nl * kSmallMax * 4) || !e->x_feat.reserve((size_t)nl * kSmallMax * 4) ||
        !e->x_histo.reserve(groups * kExtSyms * 4) || !e->x_tab.reserve(groups * kExtTabStride * sizeof(uint2)) || !e->x_codes.reserve(groups * kExtSyms * 4) ||
        !e->x_desc.reserve(groups * kNumCodes * kExtDescWords * 4) || !e->x_descbits.reserve(groups * kNumCodes * 4) || !e->x_rc.reserve(groups * kNumCodes * 4) ||
        (e->cache_bits && !e->x_rows.reserve((sb_total << e->cache_bits) * 4)))
      return VP8L_ERR_NO_MEMORY;
    ext.on = 1;
    ext.cache_bits = (uint32_t)e->cache_bits;
    ext.groups = e->meta_bits ? (uint32_t)e->meta_groups : 1u;
    ext.hitbits = e->x_hit.as<uint32_t>();
    ext.metaimg = e->x_meta.as<uint32_t>();
    ext.feature = e->x_feat.as<uint32_t>();
    ext.histo = e->x_histo.as<uint32_t>();
    ext.tab = e->x_tab.as<uint2>();
    ext.codes = e->x_codes.as<uint32_t>();
    ext.desc = e->x_desc.as<uint32_t>();
    ext.descbits = e->x_descbits.as<uint32_t>();
    ext.rc = e->x_rc.as<int32_t>();
    ext.rows = e->x_rows.as<uint32_t>();
  }
  rc_launch = launch_encode_kernels(s, B, nl, (uint32_t)sb_total, pred, cross, e->match_mode, d_out, e->ev, e->standard ? 1u : 0u, &ext,
                                    (uint32_t)e->meta_bits, (uint32_t)max_sb);
  if (rc_launch != VP8L_OK) return rc_launch;
  if (!hip_ok(hipMemcpyAsync(e->states.data(), st, nl * sizeof(EncState), hipMemcpyDeviceToHost, s), "download state")) return VP8L_ERR_HIP;
  if (!hip_ok(hipStreamSynchronize(s), "encode batch")) return VP8L_ERR_HIP;

  float ms = 0;
  (void)hipEventElapsedTime(&ms, e->ev[0], e->ev[8]);
  e->last_seconds = ms * 1e-3;
  for (int k = 0; k < 8; k++) {
    ms = 0;
    (void)hipEventElapsedTime(&ms, e->ev[k], e->ev[k + 1]);
    e->stage_seconds[k] = ms * 1e-3;
  }
  ms = 0;
  (void)hipEventElapsedTime(&ms, e->ev[9], e->ev[3]);  // the exhaustive kernel alone (part of stage 2)
  e->stage_seconds[8] = ms * 1e-3;
  e->last_out = d_out;
  e->last_pix = pix;
  e->last_pred = pred != 0;
  e->last_cross = cross != 0;
  for (uint32_t k = 0; k < nl; k++) {
    const size_t i = live[k];
    status[i] = e->states[k].status;
    if (status[i] == VP8L_OK) out_len[i] = (size_t)((e->states[k].total_bits + 7) / 8);
  }
  // keep `imgs` indexable by the caller's image number for fetch_stage
  if (nl != n) {
    std::vector<EncImage> byidx(n, EncImage{});
    std::vector<EncState> stidx(n, EncState{});
    for (uint32_t k = 0; k < nl; k++) { byidx[live[k]] = e->imgs[k]; stidx[live[k]] = e->states[k]; }
    e->imgs.swap(byidx);
    e->states.swap(stidx);
  }
  return VP8L_OK;
}

extern "C" int vp8l_encoder_encode_device(vp8l_encoder *e, size_t n, const uint32_t *const *d_argb, const uint32_t *width,
                                          const uint32_t *height, uint8_t *d_out, size_t out_cap, size_t *out_offset,
                                          size_t *out_len, int *status) {
  return (int)guarded([&] { return (int)vp8l_encoder_encode_device_impl(e, n, d_argb, width, height, d_out, out_cap, out_offset, out_len, status); });
}

extern "C" long vp8l_encoder_fetch_stage(vp8l_encoder *e, size_t image, int what, void *host_buf, size_t cap) {
  if (!e || !host_buf) return VP8L_ERR_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lock(e->mu);
  if (image >= e->imgs.size() || e->imgs[image].argb == nullptr) return VP8L_ERR_BAD_ARGUMENT;
  if (!hip_ok(hipSetDevice(e->device), "hipSetDevice")) return VP8L_ERR_HIP;
  const EncImage &im = e->imgs[image];
  const EncState &st = e->states[image];
  // index of this image among the live ones = its slot in the device tables
  size_t slot = 0;
  for (size_t i = 0; i < image; i++) slot += e->imgs[i].argb != nullptr;
  const void *src = nullptr;
  size_t bytes = 0;
  uint32_t info[4];
  switch (what) {
    case VP8L_STAGE_RESIDUAL: src = e->last_pix + im.pos0; bytes = (size_t)st.M * 4; break;
    case VP8L_STAGE_PRED_MODES:  // uint32[3] = bits, tile columns, tile rows; then one uint32 per tile (mode << 8 | 0xff000000)
      if (!e->last_pred || st.ncolors || st.pred_bits == 0) return 0;
      info[0] = st.pred_bits; info[1] = st.pred_tx; info[2] = st.pred_ty;
      bytes = 12 + (size_t)st.pred_tx * st.pred_ty * 4;
      if (bytes > cap) return VP8L_ERR_BAD_ARGUMENT;
      memcpy(host_buf, info, 12);
      if (!hip_ok(hipMemcpy(static_cast<uint8_t *>(host_buf) + 12, e->d_modes.as<uint32_t>() + slot * kSmallMax, bytes - 12, hipMemcpyDeviceToHost), "fetch modes")) return VP8L_ERR_HIP;
      return (long)bytes;
    case VP8L_STAGE_CROSS_ELEMS:  // uint32[3] = bits, tile columns, tile rows; then one element per tile
      if (!e->last_cross || st.ncolors || st.cross_bits == 0) return 0;
      info[0] = st.cross_bits; info[1] = st.cross_tx; info[2] = st.cross_ty;
      bytes = 12 + (size_t)st.cross_tx * st.cross_ty * 4;
      if (bytes > cap) return VP8L_ERR_BAD_ARGUMENT;
      memcpy(host_buf, info, 12);
      if (!hip_ok(hipMemcpy(static_cast<uint8_t *>(host_buf) + 12, e->d_elems.as<uint32_t>() + slot * kSmallMax, bytes - 12, hipMemcpyDeviceToHost), "fetch elements")) return VP8L_ERR_HIP;
      return (long)bytes;
    case VP8L_STAGE_META_GROUPS:  // the same layout: the Huffman group of every tile of the entropy image
      if (!e->meta_bits || st.meta_bits == 0) return 0;
      info[0] = st.meta_bits; info[1] = st.meta_tx; info[2] = st.meta_ty;
      bytes = 12 + (size_t)st.meta_tx * st.meta_ty * 4;
      if (bytes > cap) return VP8L_ERR_BAD_ARGUMENT;
      memcpy(host_buf, info, 12);
      if (!hip_ok(hipMemcpy(static_cast<uint8_t *>(host_buf) + 12, e->x_meta.as<uint32_t>() + slot * kSmallMax, bytes - 12, hipMemcpyDeviceToHost), "fetch groups")) return VP8L_ERR_HIP;
      return (long)bytes;
    case VP8L_STAGE_MATCH: src = e->d_match.as<uint32_t>() + im.pos0; bytes = (size_t)st.M * 4; break;
    case VP8L_STAGE_HISTO: src = e->d_histo.as<uint32_t>() + slot * kNumCodes * kCodeStride; bytes = kNumCodes * kCodeStride * 4; break;
    case VP8L_STAGE_CODELEN: src = e->d_tables.as<CodeTables>()[slot].len; bytes = kNumCodes * kCodeStride; break;
    case VP8L_STAGE_INFO:
      info[0] = st.ncolors; info[1] = st.bits; info[2] = st.xsize; info[3] = st.M;
      if (cap < sizeof(info)) return VP8L_ERR_BAD_ARGUMENT;
      memcpy(host_buf, info, sizeof(info));
      return (long)sizeof(info);
    default: return VP8L_ERR_BAD_ARGUMENT;
  }
  if (bytes > cap) return VP8L_ERR_BAD_ARGUMENT;
  if (!hip_ok(hipMemcpy(host_buf, src, bytes, hipMemcpyDeviceToHost), "fetch stage")) return VP8L_ERR_HIP;
  if (what == VP8L_STAGE_MATCH) {
    // fold the parse flags (kept as a bit array on the device) into bit 31
    std::vector<uint32_t> fl((st.M + 31) / 32);
    if (!hip_ok(hipMemcpy(fl.data(), e->d_flags.as<uint32_t>() + im.pos0 / 32, fl.size() * 4, hipMemcpyDeviceToHost), "fetch flags"))
      return VP8L_ERR_HIP;
    uint32_t *m = static_cast<uint32_t *>(host_buf);
    for (uint32_t p = 0; p < st.M; p++) m[p] |= ((fl[p >> 5] >> (p & 31)) & 1u) << 31;
  }
  return (long)bytes;
}

extern "C" double vp8l_encoder_last_device_seconds(const vp8l_encoder *e) { return e ? e->last_seconds : 0.0; }

extern "C" int vp8l_encoder_set_option(vp8l_encoder *e, int option, int value) {
  if (!e) return VP8L_ERR_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lock(e->mu);
  if (option == VP8L_OPT_MATCH_SEARCH && value >= 0 && value <= 4) { e->match_mode = value; return VP8L_OK; }
  if (option == VP8L_OPT_PREDICTOR_BITS && (value == 0 || (value >= 2 && value <= 9))) { e->pred_bits = value; return VP8L_OK; }
  if (option == VP8L_OPT_CROSS_COLOR_BITS && (value == 0 || (value >= 2 && value <= 9))) { e->cross_bits = value; return VP8L_OK; }
  if (option == VP8L_OPT_STANDARD && (value == 0 || value == 1)) { e->standard = value != 0; return VP8L_OK; }
  if (option == VP8L_OPT_CACHE_BITS && value >= 0 && value <= (int)kMaxCacheBits) { e->cache_bits = value; return VP8L_OK; }
  if (option == VP8L_OPT_META_BITS && (value == 0 || (value >= 2 && value <= 9))) { e->meta_bits = value; return VP8L_OK; }
  if (option == VP8L_OPT_META_GROUPS && value >= 2 && value <= (int)kExtMaxGroups) { e->meta_groups = value; return VP8L_OK; }
  return VP8L_ERR_BAD_ARGUMENT;
}

extern "C" long vp8l_encoder_last_dense_tiles(vp8l_encoder *e, long *total_tiles) {
  if (!e) return VP8L_ERR_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lock(e->mu);
  if (e->sb_first.empty()) return 0;
  if (!hip_ok(hipSetDevice(e->device), "hipSetDevice")) return VP8L_ERR_HIP;
  const size_t ntiles = (size_t)e->sb_first.back() * (kSuper / kLzTile);
  std::vector<uint8_t> f(ntiles);
  if (!hip_ok(hipMemcpy(f.data(), e->d_dense.ptr, ntiles, hipMemcpyDeviceToHost), "fetch tile flags")) return VP8L_ERR_HIP;
  long dense = 0, total = 0;
  // count only tiles that hold pixels
  for (size_t i = 0; i < e->imgs.size(); i++) {
    if (!e->imgs[i].argb) continue;
    const uint64_t t0 = e->imgs[i].pos0 / kLzTile, nt = ((uint64_t)e->states[i].M + kLzTile - 1) / kLzTile;
    for (uint64_t t = 0; t < nt; t++) dense += f[t0 + t] == kTileExhaustive;
    total += (long)nt;
  }
  if (total_tiles) *total_tiles = total;
  return e->match_mode != 1 ? dense : total;
}

// How the tiles of the last call were searched: counts[flag] for the tile flags 0..7 (0 candidate lists, 1 exhaustive,
// 2 pairs of pixels, 4 frequent value); tiles that hold pixels only.  Returns the number of such tiles.
extern "C" long vp8l_encoder_last_tile_kinds(vp8l_encoder *e, long *counts8) {
  if (!e || !counts8) return VP8L_ERR_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lock(e->mu);
  for (int k = 0; k < 8; k++) counts8[k] = 0;
  if (e->sb_first.empty()) return 0;
  if (!hip_ok(hipSetDevice(e->device), "hipSetDevice")) return VP8L_ERR_HIP;
  const size_t ntiles = (size_t)e->sb_first.back() * (kSuper / kLzTile);
  std::vector<uint8_t> f(ntiles);
  if (!hip_ok(hipMemcpy(f.data(), e->d_dense.ptr, ntiles, hipMemcpyDeviceToHost), "fetch tile flags")) return VP8L_ERR_HIP;
  long total = 0;
  for (size_t i = 0; i < e->imgs.size(); i++) {
    if (!e->imgs[i].argb) continue;
    const uint64_t t0 = e->imgs[i].pos0 / kLzTile, nt = ((uint64_t)e->states[i].M + kLzTile - 1) / kLzTile;
    for (uint64_t t = 0; t < nt; t++) counts8[f[t0 + t] & 7]++;
    total += (long)nt;
  }
  return total;
}

extern "C" int vp8l_encoder_last_stage_seconds(const vp8l_encoder *e, double *seconds, int cap) {
  if (!e || !seconds) return VP8L_ERR_BAD_ARGUMENT;
  const int k = cap < 9 ? cap : 9;
  for (int i = 0; i < k; i++) seconds[i] = e->stage_seconds[i];
  return k;
}

extern "C" int vp8l_generate_device(int device, int mode, uint32_t w, uint32_t h, uint32_t seed, uint32_t ncolors, uint32_t *d_argb) {
  if (!d_argb || w == 0 || h == 0 || mode < 3 || mode > 5) return VP8L_ERR_BAD_ARGUMENT;
  int rc = select_device(device, &device);
  if (rc != VP8L_OK) return rc;
  if (mode == 5 && (ncolors == 0 || ncolors > 256)) ncolors = 200;
  const uint64_t npx = (uint64_t)w * h;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((npx + 255) / 256, 8192);
  hipLaunchKernelGGL(k_generate, dim3(grid), dim3(256), 0, 0, mode, w, h, seed, ncolors, d_argb);
  if (!hip_ok(hipGetLastError(), "k_generate")) return VP8L_ERR_HIP;
  if (!hip_ok(hipDeviceSynchronize(), "k_generate")) return VP8L_ERR_HIP;
  return VP8L_OK;
}

// =============================================================================================
// One image with no allocation at all: the encode path's workspace lives in a static device array.
// For callers that cannot call hipMalloc (code under HIP graph capture, serving loops, harnesses that forbid it).
// Every GPU the library runs on holds its own copy of the array (at its own address) until the process ends; calls
// that overlap on one GPU would share it and race.
// =============================================================================================
namespace vp8l {
namespace {
constexpr uint64_t kNaMaxPixels = VP8L_NOALLOC_MAX_PIXELS;
constexpr uint64_t kNaSuper = (kNaMaxPixels + kTailPad + kSuper - 1) / kSuper;  // superblocks of the largest image
constexpr uint64_t kNaPos = kNaSuper * kSuper;
constexpr uint64_t na_align(uint64_t v) { return (v + 255) & ~(uint64_t)255; }
// offsets of the parts, each 256-byte aligned
constexpr uint64_t kNaImgs = 0;
constexpr uint64_t kNaState = kNaImgs + na_align(sizeof(EncImage));
constexpr uint64_t kNaSbf = kNaState + na_align(sizeof(EncState));
constexpr uint64_t kNaResult = kNaSbf + na_align(2 * sizeof(uint32_t));
constexpr uint64_t kNaGsets = kNaResult + na_align(2 * sizeof(uint64_t));
constexpr uint64_t kNaHisto = kNaGsets + na_align(kSetSlots * sizeof(unsigned long long));
constexpr uint64_t kNaTables = kNaHisto + na_align(kNumCodes * kCodeStride * 4);
constexpr uint64_t kNaSx = kNaTables + na_align(sizeof(CodeTables));
constexpr uint64_t kNaEntry = kNaSx + na_align(kNaSuper * kBlockLen);
constexpr uint64_t kNaUbits = kNaEntry + na_align(kNaSuper);
constexpr uint64_t kNaUoff = kNaUbits + na_align(kNaSuper * kUnitsPerSuper * 4);
constexpr uint64_t kNaDense = kNaUoff + na_align(kNaSuper * kUnitsPerSuper * 8);
constexpr uint64_t kNaGbits = kNaDense + na_align(kNaSuper * (kSuper / kLzTile) + 16 + kDenseTail);  // (the counters behind the flags start at a multiple of 16)
constexpr uint64_t kNaFlags = kNaGbits + na_align(kNaPos / 16 * 2);
constexpr uint64_t kNaAdv = kNaFlags + na_align(kNaPos / 8);
constexpr uint64_t kNaPix = kNaAdv + na_align(kNaPos);
constexpr uint64_t kNaMatch = kNaPix + na_align(kNaPos * 4);
constexpr uint64_t kNaBytes = kNaMatch + na_align(kNaPos * 4);
static_assert(kNaBytes < (24u << 20), "the static workspace stays modest");
}  // namespace

// In host code this name is only a handle for hipGetSymbolAddress: the memory is reached through the per-device
// address that call returns, never through the name.
static __device__ __align__(256) uint8_t scratch[kNaBytes];

namespace {
constexpr int kNaMaxDevices = 64;
uint8_t *g_na_ptr[kNaMaxDevices] = {};  // address of `scratch` on each device, 0 = not asked yet
std::mutex g_na_mu;

// Address of the static workspace on `device` (which must be the current device); asked once per device.
int noalloc_workspace(int device, uint8_t **out) {
  if (device < 0 || device >= kNaMaxDevices) return VP8L_ERR_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lock(g_na_mu);
  if (!g_na_ptr[device]) {
    hipDeviceProp_t prop;
    if (!hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) return VP8L_ERR_NO_DEVICE;
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
      last_error_slot() = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
      return VP8L_ERR_NO_DEVICE;
    }
    void *p = nullptr;
    if (!hip_ok(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)") || !p) return VP8L_ERR_HIP;
    g_na_ptr[device] = static_cast<uint8_t *>(p);
  }
  *out = g_na_ptr[device];
  return VP8L_OK;
}
}  // namespace

// Called by vp8l_init(): looks the workspace address up on the calling thread's current device, so that the entry
// point's first call there does nothing but launch.  Other devices are looked up by their own first call: asking on a
// device creates a context and loads the code object there (with the workspace), which a process that uses one GPU of
// eight should not pay eight times.
void noalloc_setup_current_device() {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); return; }
  uint8_t *p = nullptr;
  (void)noalloc_workspace(device, &p);
}

// Writes the one-image tables a batch call uploads from the host.
__global__ void k_noalloc_setup(EncImage *imgs, uint32_t *sbf, const uint32_t *argb, uint32_t w, uint32_t h, uint64_t out_cap,
                                uint32_t sb_total) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    EncImage im;
    im.argb = argb; im.w = w; im.h = h; im.pos0 = 0; im.out_off = 0; im.out_cap = out_cap;
    imgs[0] = im;
    sbf[0] = 0;
    sbf[1] = sb_total;
  }
}

// result[0] = status (a vp8l_status, sign-extended), result[1] = stream bytes
__global__ void k_noalloc_result(const EncState *st, uint64_t *result, uint64_t *result2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const uint64_t s = (uint64_t)(int64_t)st[0].status, len = st[0].status == 0 ? (st[0].total_bits + 7) / 8 : 0;
    result[0] = s; result[1] = len;
    if (result2) { result2[0] = s; result2[1] = len; }
  }
}
}  // namespace vp8l

extern "C" int vp8l_encode_device_noalloc(int device, void *stream, const uint32_t *d_argb, uint32_t width, uint32_t height,
                                          uint8_t *d_out, size_t out_cap, uint64_t *d_result, size_t *out_len) {
  if (out_len) *out_len = 0;
  if (!d_out || ((uintptr_t)d_out & 255) != 0) return VP8L_ERR_BAD_ARGUMENT;
  if (!d_argb || width == 0 || height == 0) return VP8L_ERR_INVALID_ARGB;         // vp8l.h:587-588
  if (width > kMaxDim || height > kMaxDim) return VP8L_ERR_TOO_BIG_ARGB;          // vp8l.h:589-590
  const uint64_t px = (uint64_t)width * height;
  if (px > kNaMaxPixels) {
    last_error_slot() = "vp8l_encode_device_noalloc takes at most VP8L_NOALLOC_MAX_PIXELS pixels";
    return VP8L_ERR_NO_MEMORY;
  }
  if (out_cap < vp8l_encode_bound(width, height)) return VP8L_ERR_NO_MEMORY;
  // a device >= 0 is made current for the length of this call only (the caller's own choice comes back on every path)
  struct DeviceScope {
    int before = -1;
    ~DeviceScope() { if (before >= 0) (void)hipSetDevice(before); }
  } scope;
  if (device < 0) {
    if (!hip_ok(hipGetDevice(&device), "hipGetDevice")) return VP8L_ERR_NO_DEVICE;
  } else {
    int current = -1;
    if (!hip_ok(hipGetDevice(&current), "hipGetDevice")) return VP8L_ERR_NO_DEVICE;
    if (current != device) {
      if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return VP8L_ERR_NO_DEVICE;
      scope.before = current;
    }
  }
  uint8_t *ws = nullptr;
  int rc = noalloc_workspace(device, &ws);
  if (rc != VP8L_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  EncBuffers B;
  B.imgs = reinterpret_cast<EncImage *>(ws + kNaImgs);
  B.st = reinterpret_cast<EncState *>(ws + kNaState);
  B.sbf = reinterpret_cast<uint32_t *>(ws + kNaSbf);
  B.gsets = reinterpret_cast<unsigned long long *>(ws + kNaGsets);
  B.pix_t = B.pix = reinterpret_cast<uint32_t *>(ws + kNaPix);
  B.modes = B.elems = nullptr;
  B.match = reinterpret_cast<uint32_t *>(ws + kNaMatch);
  B.flags = reinterpret_cast<uint32_t *>(ws + kNaFlags);
  B.histo = reinterpret_cast<uint32_t *>(ws + kNaHisto);
  B.tables = reinterpret_cast<CodeTables *>(ws + kNaTables);
  B.sx = ws + kNaSx;
  B.entry = ws + kNaEntry;
  B.ubits = reinterpret_cast<uint32_t *>(ws + kNaUbits);
  B.uoff = reinterpret_cast<uint64_t *>(ws + kNaUoff);
  B.dense = ws + kNaDense;
  B.gbits = reinterpret_cast<uint16_t *>(ws + kNaGbits);
  B.adv8 = ws + kNaAdv;
  uint64_t *result = reinterpret_cast<uint64_t *>(ws + kNaResult);
  const uint32_t sb_total = (uint32_t)((px + kTailPad + kSuper - 1) / kSuper);
  hipLaunchKernelGGL(k_noalloc_setup, dim3(1), dim3(64), 0, s, B.imgs, B.sbf, d_argb, width, height, (uint64_t)vp8l_encode_bound(width, height), sb_total);
  rc = launch_encode_kernels(s, B, 1, sb_total, 0, 0, 0, d_out, nullptr, 0, nullptr, 0, (uint32_t)((px + kSuper - 1) / kSuper));
  if (rc != VP8L_OK) return rc;
  hipLaunchKernelGGL(k_noalloc_result, dim3(1), dim3(64), 0, s, B.st, result, d_result);
  if (!hip_ok(hipGetLastError(), "kernel launch")) return VP8L_ERR_HIP;
  if (!out_len) return VP8L_OK;  // the caller reads d_result behind its own synchronisation
  uint64_t host_result[2] = {0, 0};
  if (!hip_ok(hipMemcpyAsync(host_result, result, sizeof(host_result), hipMemcpyDeviceToHost, s), "download result")) return VP8L_ERR_HIP;
  if (!hip_ok(hipStreamSynchronize(s), "encode (no allocation)")) return VP8L_ERR_HIP;
  *out_len = (size_t)host_result[1];
  return (int)(int64_t)host_result[0];
}

// Measurement aid: why k_lz77_dominant left tiles to the exhaustive kernel since the last call (then cleared); see g_dm_reason
// (zeros from a library that was not built with -DVP8L_PHASE_TIMING: the counting costs time).
extern "C" int vp8l_debug_dominant_reasons(unsigned int *out4) {
  if (!out4) return VP8L_ERR_BAD_ARGUMENT;
  if (!hip_ok(hipDeviceSynchronize(), "sync")) return VP8L_ERR_HIP;
  if (!hip_ok(hipMemcpyFromSymbol(out4, HIP_SYMBOL(vp8l::g_dm_reason), 4 * sizeof(unsigned int)), "read reasons")) return VP8L_ERR_HIP;
  unsigned int zero[4] = {};
  if (!hip_ok(hipMemcpyToSymbol(HIP_SYMBOL(vp8l::g_dm_reason), zero, sizeof(zero)), "clear reasons")) return VP8L_ERR_HIP;
  return VP8L_OK;
}

#ifdef VP8L_PHASE_TIMING
// Measurement builds only: what k_lz77_walk did since the last call (then cleared); see g_wk_stat.
extern "C" int vp8l_debug_walk_stats(unsigned long long *out8) {
  if (!out8) return VP8L_ERR_BAD_ARGUMENT;
  if (!hip_ok(hipDeviceSynchronize(), "sync")) return VP8L_ERR_HIP;
  if (!hip_ok(hipMemcpyFromSymbol(out8, HIP_SYMBOL(vp8l::g_wk_stat), 8 * sizeof(unsigned long long)), "read stats")) return VP8L_ERR_HIP;
  unsigned long long zero[8] = {};
  if (!hip_ok(hipMemcpyToSymbol(HIP_SYMBOL(vp8l::g_wk_stat), zero, sizeof(zero)), "clear stats")) return VP8L_ERR_HIP;
  return VP8L_OK;
}
// Measurement builds only: cycles per phase of k_lz77_sparse summed over the workgroups since the last call (then cleared).
extern "C" int vp8l_debug_dominant_phase_cycles(unsigned long long *out12) {
  if (!hip_ok(hipDeviceSynchronize(), "sync")) return VP8L_ERR_HIP;
  if (!hip_ok(hipMemcpyFromSymbol(out12, HIP_SYMBOL(vp8l::g_dm_phase), 12 * sizeof(unsigned long long)), "read phase counters")) return VP8L_ERR_HIP;
  unsigned long long zero[12] = {};
  if (!hip_ok(hipMemcpyToSymbol(HIP_SYMBOL(vp8l::g_dm_phase), zero, sizeof(zero)), "clear phase counters")) return VP8L_ERR_HIP;
  return VP8L_OK;
}
extern "C" int vp8l_debug_grams_phase_cycles(unsigned long long *out12) {
  if (!hip_ok(hipDeviceSynchronize(), "sync")) return VP8L_ERR_HIP;
  if (!hip_ok(hipMemcpyFromSymbol(out12, HIP_SYMBOL(vp8l::g_gm_phase), 12 * sizeof(unsigned long long)), "read phase counters")) return VP8L_ERR_HIP;
  unsigned long long zero[12] = {};
  if (!hip_ok(hipMemcpyToSymbol(HIP_SYMBOL(vp8l::g_gm_phase), zero, sizeof(zero)), "clear phase counters")) return VP8L_ERR_HIP;
  return VP8L_OK;
}
extern "C" int vp8l_debug_phase_cycles(unsigned long long *out8) {
  if (!hip_ok(hipDeviceSynchronize(), "sync")) return VP8L_ERR_HIP;
  if (!hip_ok(hipMemcpyFromSymbol(out8, HIP_SYMBOL(vp8l::g_sp_phase), 8 * sizeof(unsigned long long)), "read phase counters")) return VP8L_ERR_HIP;
  unsigned long long zero[8] = {};
  if (!hip_ok(hipMemcpyToSymbol(HIP_SYMBOL(vp8l::g_sp_phase), zero, sizeof(zero)), "clear phase counters")) return VP8L_ERR_HIP;
  return VP8L_OK;
}
#endif

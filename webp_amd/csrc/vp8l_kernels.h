// vp8l_kernels.h -- device-side layout and helpers shared by the encode and decode kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "vp8l_common.h"

namespace vp8l {

// ---- batch layout in HBM ---------------------------------------------------------------------
// Every image of a batch owns a region of the per-position work arrays (transformed pixels,
// matches, parse flags) that starts at a multiple of kSuper positions.  All kernels work on
// units that divide kSuper, so a unit never straddles two images and a block finds its image
// with a binary search over the images' first-superblock table.
constexpr uint32_t kSuper = 16384;      // positions per superblock (parse granule)
constexpr uint32_t kBlockLen = 128;     // positions per parse block = longest match
constexpr uint32_t kBlocksPerSuper = kSuper / kBlockLen;
constexpr uint32_t kUnit = 4096;        // positions per histogram / bit-count / emit unit
constexpr uint32_t kUnitsPerSuper = kSuper / kUnit;
constexpr uint32_t kLzTile = 2048;      // positions per match-search wave
constexpr uint32_t kLzWaves = 4;        // waves per match-search block
constexpr uint32_t kLzBlock = kLzTile * kLzWaves;
constexpr uint32_t kTailPad = 64;       // readable slack behind every image's region

struct EncImage {          // host-written
  const uint32_t *argb;
  uint32_t w, h;
  uint64_t pos0;           // first element of the image's region in the per-position arrays
  uint64_t out_off;        // byte offset of its stream in the output buffer (256-aligned)
  uint64_t out_cap;        // bytes reserved there
};

struct EncState {          // device-written
  uint32_t ncolors;        // 0 = subtract-green path
  uint32_t bits;           // pixel bundling: 1<<bits pixels per coded pixel
  uint32_t xsize;          // coded width
  uint32_t M;              // coded pixels = xsize*h
  uint32_t xinv;           // ceil(2^32 / xsize): division by xsize as a multiplication (dist_to_code_inv)
  uint32_t fused;          // the match kernel makes the subtract-green residual itself (k_transform skips the image)
  uint32_t pred_bits;      // predictor transform: tile size 2^bits (0 = none), pred_tx x pred_ty tiles
  uint32_t pred_tx, pred_ty;
  uint32_t cross_bits;     // cross-colour transform: tile size 2^bits (0 = none), cross_tx x cross_ty tiles
  uint32_t cross_tx, cross_ty;
  uint32_t pad_cross;
  uint32_t hdr_bits;       // bits in front of the first token
  int32_t status;
  uint64_t total_bits;
  uint32_t pal_count;      // distinct colours seen so far (palette detection)
  uint32_t pal_overflow;
  uint32_t palette[kMaxPalette];
};

__device__ __forceinline__ uint32_t find_image(const uint32_t *__restrict__ sb_first, uint32_t n, uint32_t sb) {
  uint32_t lo = 0, hi = n;  // largest i with sb_first[i] <= sb
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (sb_first[mid] <= sb) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
  return v;
}

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

}  // namespace vp8l

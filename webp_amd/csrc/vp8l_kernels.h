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
constexpr uint32_t kLzHead = 1024;      // positions at the start of an image that have fewer than 1024 distances (= the window)
constexpr uint32_t kLzHeadPiece = 128;  // ... searched in pieces of this many
constexpr uint32_t kTailPad = 64;       // readable slack behind every image's region
constexpr uint32_t kFewTiles = 512;    // a call that leaves at most this many tiles to the exhaustive search has them searched in pieces
constexpr uint32_t kDenseTail = 64 + kFewTiles * 4;  // bytes behind the per-tile flags of the match search, cleared with them: work counters, then the numbers of the first kFewTiles tiles given up on

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
  uint32_t ngroups;        // Huffman groups of the main image (1 unless an entropy image is written)
  uint32_t meta_bits;      // entropy image: tile size 2^bits (0 = none), meta_tx x meta_ty tiles over the image's width and height
  uint32_t meta_tx, meta_ty;
  uint32_t pad_meta;
  uint32_t hdr_bits;       // bits in front of the first token
  int32_t status;
  uint64_t total_bits;
  uint32_t pal_count;      // distinct colours seen so far (palette detection)
  uint32_t pal_overflow;
  uint32_t palette[kMaxPalette];
};

// ---- the coder with a colour cache and several Huffman groups (vp8l_encode_ext.h) -------------------------
constexpr uint32_t kExtGreen = (uint32_t)kExtGreenMax;  // 280 + 2048
constexpr uint32_t kExtRed = kExtGreen, kExtBlue = kExtRed + 256, kExtAlpha = kExtBlue + 256, kExtDist = kExtAlpha + 256;
constexpr uint32_t kExtSyms = kExtDist + kAlphabetDist;  // symbols of one group, the five codes one after the other
constexpr uint32_t kExtZero = kExtSyms;                  // a (0, 0) entry behind them
constexpr uint32_t kExtTabStride = kExtSyms + 2;
constexpr uint32_t kExtMaxGroups = 8;
constexpr uint32_t kExtDescWords = 4096;                 // room for one code description: 2328 items of at most 39 bits and the head

struct ExtArgs {   // by value to the kernels
  uint32_t on;            // 0: the default coder
  uint32_t cache_bits;    // 0 = no colour cache
  uint32_t groups;        // 1 = no entropy image, else the number of groups asked for (an image may end up with fewer)
  uint32_t standard;
  const uint32_t *hitbits;   // per position (bit array like the parse flags): the pixel is in the cache
  uint32_t *metaimg;         // per image kSmallMax tiles: group of the tile
  uint32_t *feature;         // per image kSmallMax tiles: rough bit cost per pixel, in 1/16
  uint32_t *histo;           // per image kExtMaxGroups x kExtSyms counts
  uint2 *tab;                // per image kExtMaxGroups x kExtTabStride (code, length); length 0 for codes that are not written
  uint32_t *codes;           // per image kExtMaxGroups x kExtSyms: scratch for the codes
  uint32_t *desc;            // per image kExtMaxGroups x 5 x kExtDescWords: the code descriptions, bit-packed
  uint32_t *descbits;        // per image kExtMaxGroups x 5: their lengths in bits
  int32_t *rc;               // per image kExtMaxGroups x 5: 0 or the failure of that code
  uint32_t *rows;            // per superblock 2^cache_bits: last position + 1 of every key (k_cache_rows / k_cache_scan)
};

__device__ __forceinline__ uint32_t find_image(const uint32_t *__restrict__ sb_first, uint32_t n, uint32_t sb) {
  uint32_t lo = 0, hi = n;  // largest i with sb_first[i] <= sb
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (sb_first[mid] <= sb) lo = mid; else hi = mid;
  }
  return lo;
}

// Inclusive scan across the 64 lanes of a wave with DPP lane moves (no trips through the LDS crossbar, which a shuffle
// is): doubling steps inside the rows of 16 lanes (lanes without a source add 0), then the total of row 0 into row 1
// and of row 2 into row 3, then the total of rows 0-1 into rows 2 and 3.  Every lane must be active.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int /*lane*/) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}

// sum over the 64 lanes of a wave, in every lane (all lanes active)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v, 0), 63);
}

}  // namespace vp8l

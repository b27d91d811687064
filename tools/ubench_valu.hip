// ubench_valu.hip -- measurement aid: issue cost of single VALU instructions on gfx950 with
// 4 waves per SIMD (the match kernel's occupancy).  Prints cycles per wave64 instruction per SIMD
// derived from s_memtime.  Build: hipcc --offload-arch=gfx950 -O2 ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(name, ASM)                                                                                   \
  __global__ __launch_bounds__(256) void k_##name(unsigned *out, unsigned long long *cyc, int iters) {   \
    unsigned v0 = threadIdx.x, v1 = threadIdx.x * 3 + 1, v2 = 7, v3 = 9, v4 = 11, v5 = 13, v6 = 17, v7 = 19; \
    unsigned s0 = iters & 0xff;                                                                           \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                 \
    for (int i = 0; i < iters; i++) {                                                                     \
      asm volatile(REP16(ASM) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "vcc", "s40", "s41", "s42", "s43"); \
    }                                                                                                     \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                 \
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                          \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                      \
  }

// each ASM string = 4 independent instructions
BODY(fma_f32, "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %4, %5, %3\n v_fma_f32 %6, %7, %1, %6\n v_fma_f32 %2, %4, %5, %2\n")
BODY(add_u32, "v_add_u32 %0, %1, %0\n v_add_u32 %3, %4, %3\n v_add_u32 %6, %7, %6\n v_add_u32 %2, %5, %2\n")
BODY(min_u32, "v_min_u32 %0, %1, %0\n v_min_u32 %3, %4, %3\n v_min_u32 %6, %7, %6\n v_min_u32 %2, %5, %2\n")
BODY(min3_u32, "v_min3_u32 %0, %1, %2, %0\n v_min3_u32 %3, %4, %5, %3\n v_min3_u32 %6, %7, %1, %6\n v_min3_u32 %2, %4, %5, %2\n")
BODY(max3_u32, "v_max3_u32 %0, %1, %2, %0\n v_max3_u32 %3, %4, %5, %3\n v_max3_u32 %6, %7, %1, %6\n v_max3_u32 %2, %4, %5, %2\n")
BODY(min3_f32, "v_min3_f32 %0, %1, %2, %0\n v_min3_f32 %3, %4, %5, %3\n v_min3_f32 %6, %7, %1, %6\n v_min3_f32 %2, %4, %5, %2\n")
BODY(max3_f32, "v_max3_f32 %0, %1, %2, %0\n v_max3_f32 %3, %4, %5, %3\n v_max3_f32 %6, %7, %1, %6\n v_max3_f32 %2, %4, %5, %2\n")
BODY(min_f32, "v_min_f32 %0, %1, %0\n v_min_f32 %3, %4, %3\n v_min_f32 %6, %7, %6\n v_min_f32 %2, %5, %2\n")
BODY(lshl_or, "v_lshl_or_b32 %0, %1, 4, %0\n v_lshl_or_b32 %3, %4, 4, %3\n v_lshl_or_b32 %6, %7, 4, %6\n v_lshl_or_b32 %2, %5, 4, %2\n")
BODY(and_b32, "v_and_b32 %0, %1, %0\n v_and_b32 %3, %4, %3\n v_and_b32 %6, %7, %6\n v_and_b32 %2, %5, %2\n")
BODY(mov_b32, "v_mov_b32 %0, %1\n v_mov_b32 %3, %4\n v_mov_b32 %6, %7\n v_mov_b32 %2, %5\n")
BODY(cmp_vcc, "v_cmp_eq_u32 vcc, %0, %1\n v_cmp_eq_u32 vcc, %3, %4\n v_cmp_eq_u32 vcc, %6, %7\n v_cmp_eq_u32 vcc, %2, %5\n")
BODY(cmp_sgpr, "v_cmp_eq_u32 s[40:41], %0, %1\n v_cmp_eq_u32 s[42:43], %3, %4\n v_cmp_eq_u32 s[40:41], %6, %7\n v_cmp_eq_u32 s[42:43], %2, %5\n")
BODY(cmp_sgprsrc, "v_cmp_eq_u32 s[40:41], %8, %1\n v_cmp_eq_u32 s[42:43], %8, %4\n v_cmp_eq_u32 s[40:41], %8, %7\n v_cmp_eq_u32 s[42:43], %8, %5\n")
BODY(cndmask, "v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %3, %4, %3, vcc\n v_cndmask_b32 %6, %7, %6, vcc\n v_cndmask_b32 %2, %5, %2, vcc\n")
BODY(cndmask_const, "v_cndmask_b32 %0, 15, %0, vcc\n v_cndmask_b32 %3, 15, %3, vcc\n v_cndmask_b32 %6, 15, %6, vcc\n v_cndmask_b32 %2, 15, %2, vcc\n")
BODY(cmp_cnd_pair, "v_cmp_eq_u32 s[40:41], %8, %1\n v_cmp_eq_u32 s[42:43], %8, %4\n v_cndmask_b32 %0, 15, %0, s[40:41]\n v_cndmask_b32 %3, 15, %3, s[42:43]\n")
BODY(pk_min_u16, "v_pk_min_u16 %0, %1, %0\n v_pk_min_u16 %3, %4, %3\n v_pk_min_u16 %6, %7, %6\n v_pk_min_u16 %2, %5, %2\n")
BODY(pk_max_u16, "v_pk_max_u16 %0, %1, %0\n v_pk_max_u16 %3, %4, %3\n v_pk_max_u16 %6, %7, %6\n v_pk_max_u16 %2, %5, %2\n")
BODY(pk_add_u16, "v_pk_add_u16 %0, %1, %0\n v_pk_add_u16 %3, %4, %3\n v_pk_add_u16 %6, %7, %6\n v_pk_add_u16 %2, %5, %2\n")
BODY(pk_fma_f32_like, "v_pk_mul_lo_u16 %0, %1, %0\n v_pk_mul_lo_u16 %3, %4, %3\n v_pk_mul_lo_u16 %6, %7, %6\n v_pk_mul_lo_u16 %2, %5, %2\n")
BODY(sad_u8, "v_sad_u8 %0, %1, %2, %0\n v_sad_u8 %3, %4, %5, %3\n v_sad_u8 %6, %7, %1, %6\n v_sad_u8 %2, %4, %5, %2\n")
BODY(or3, "v_or3_b32 %0, %1, %2, %0\n v_or3_b32 %3, %4, %5, %3\n v_or3_b32 %6, %7, %1, %6\n v_or3_b32 %2, %4, %5, %2\n")
BODY(bfi, "v_bfi_b32 %0, %1, %2, %0\n v_bfi_b32 %3, %4, %5, %3\n v_bfi_b32 %6, %7, %1, %6\n v_bfi_b32 %2, %4, %5, %2\n")
BODY(mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0\n v_mad_u32_u24 %3, %4, %5, %3\n v_mad_u32_u24 %6, %7, %1, %6\n v_mad_u32_u24 %2, %4, %5, %2\n")
BODY(cmp_f32, "v_cmp_eq_f32 s[40:41], %0, %1\n v_cmp_eq_f32 s[42:43], %3, %4\n v_cmp_eq_f32 s[40:41], %6, %7\n v_cmp_eq_f32 s[42:43], %2, %5\n")
BODY(subb, "v_subb_co_u32 %0, vcc, %1, %0, vcc\n v_subb_co_u32 %3, vcc, %4, %3, vcc\n v_subb_co_u32 %6, vcc, %7, %6, vcc\n v_subb_co_u32 %2, vcc, %5, %2, vcc\n")
BODY(dpp_max, "v_max_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %3, %4, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %6, %7, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %2, %5, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
BODY(sdwa_cnd, "v_cndmask_b32_sdwa %0, %1, %0, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1\n v_cndmask_b32_sdwa %3, %4, %3, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1\n v_cndmask_b32_sdwa %6, %7, %6, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1\n v_cndmask_b32_sdwa %2, %5, %2, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1\n")

typedef void (*kern_t)(unsigned *, unsigned long long *, int);
struct Entry { const char *name; kern_t k; };

int main(int argc, char **argv) {
  const int blocks = 256 * 4, iters = 2000;  // 4 blocks of 4 waves per CU -> 4 waves per SIMD
  unsigned *out; unsigned long long *cyc;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
  std::vector<Entry> es = {
#define E(n) {#n, k_##n}
    E(fma_f32), E(add_u32), E(min_u32), E(min3_u32), E(max3_u32), E(min3_f32), E(max3_f32), E(min_f32), E(lshl_or), E(and_b32), E(mov_b32),
    E(cmp_vcc), E(cmp_sgpr), E(cmp_sgprsrc), E(cndmask), E(cndmask_const), E(cmp_cnd_pair), E(pk_min_u16), E(pk_max_u16), E(pk_add_u16),
    E(pk_fma_f32_like), E(sad_u8), E(or3), E(bfi), E(mad_u32_u24), E(cmp_f32), E(subb), E(dpp_max), E(sdwa_cnd)};
  std::vector<unsigned long long> h(blocks);
  for (auto &e : es) {
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= blocks;
    // per wave: iters*64 instructions in `avg` cycles while sharing its SIMD with 3 other waves
    double per_instr_wave = avg / (iters * 64.0);
    printf("%-16s wall %.3f ms  cycles/instr seen by one wave %.2f  => per SIMD issue %.2f cycles (4 waves)\n", e.name, ms, per_instr_wave, per_instr_wave / 4.0);
  }
  return 0;
}

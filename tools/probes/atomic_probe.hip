// Measurement aid: what a flush of per-workgroup histograms into per-image histograms costs, as k_parse_mark does it
// (every workgroup adds its ~1100 counters to its image's, 506 workgroups per image), against plain stores of as many words.
// Build: hipcc --offload-arch=gfx950 -O2 -o atomic_probe atomic_probe.hip ; run: ./atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr unsigned kCounters = 1120, kPerImage = 506, kImages = 256;
__global__ void k_flush(unsigned *g, unsigned nonzero_mod) {
  const unsigned img = blockIdx.x / kPerImage;
  for (unsigned i = threadIdx.x; i < kCounters; i += 256)
    if (i % nonzero_mod == 0) atomicAdd(&g[img * kCounters + i], blockIdx.x + 1);
}
__global__ void k_store(unsigned *g) {
  for (unsigned i = threadIdx.x; i < kCounters; i += 256) g[(size_t)blockIdx.x * kCounters + i] = blockIdx.x + 1;
}
__global__ void k_one(unsigned *g) { if (threadIdx.x == 0) atomicAdd(g, 1u); }
int main() {
  unsigned *g; CK(hipMalloc(&g, (size_t)kPerImage * kImages * kCounters * 4));
  CK(hipMemset(g, 0, (size_t)kPerImage * kImages * kCounters * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms;
  for (unsigned mod : {1u, 2u, 4u}) {
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(a)); k_flush<<<kPerImage * kImages, 256>>>(g, mod); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("flush every %u-th counter: %.3f ms (%.1f M atomics)\n", mod, ms, kPerImage * kImages * (kCounters / mod) / 1e6);
  }
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(a)); k_store<<<kPerImage * kImages, 256>>>(g); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
  }
  printf("plain stores of as many words: %.3f ms\n", ms);
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(a)); k_one<<<65536, 64>>>(g); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
  }
  printf("65536 workgroups adding to one address: %.3f ms\n", ms);
  CK(hipFree(g));
  return 0;
}

// like placement.hip for the linear weights: several 400 MB tables, random 4-byte read-modify-writes (32 per wavefront-lane group)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x;
}
__global__ void __launch_bounds__(256) k_w(float* __restrict__ tab, uint64_t n, uint64_t total, uint64_t salt) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= total) return;
  const uint64_t j = (uint64_t)(((unsigned __int128)mix64(tid + salt) * n) >> 64);
  tab[j] = tab[j] * 0.999f + 1.0f;
}
int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 16;
  const uint64_t n = 100000000ull;
  const uint64_t total = 1ull << 25;
  std::vector<float*> tabs(T);
  for (int i = 0; i < T; i++) { CK(hipMalloc(&tabs[i], n * 4)); CK(hipMemset(tabs[i], 0, n * 4)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> sum(T, 0);
  for (int r = 0; r < 5; r++)
    for (int i = 0; i < T; i++) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_w, dim3((unsigned)(total / 256)), dim3(256), 0, 0, tabs[i], n, total, (uint64_t)r * 977 + 1);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) sum[i] += ms;
    }
  for (int i = 0; i < T; i++) printf("w table %2d: mean %.3f ms -> %.1f G read-modify-writes/s\n", i, sum[i] / 4, total / (sum[i] / 4 * 1e-3) / 1e9);
  return 0;
}

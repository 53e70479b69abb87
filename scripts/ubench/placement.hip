// micro-benchmark: does the rate of a random 256-byte row gather (+ write-back) depend on WHICH allocation the table is?
// Several tables of the same size are allocated in one process and the same kernel runs over each of them, round-robin.
//   hipcc --offload-arch=gfx950 -O3 placement.hip -o placement ; ./placement [tables=8] [GB each=24] [rounds=4]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x;
}
// one wavefront per "example": 32 random rows of 64 floats read (nt), scaled, written back (nt) -- the traffic shape of the FM step
__global__ void __launch_bounds__(256) k_rows(float* __restrict__ tab, uint64_t n_rows, uint32_t n_ex, uint64_t salt) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= n_ex) return;
  float v[32];
#pragma unroll
  for (int t = 0; t < 32; t++) {
    const uint64_t r = (uint64_t)(((unsigned __int128)mix64((uint64_t)wave * 32 + t + salt) * n_rows) >> 64);
    v[t] = __builtin_nontemporal_load(tab + r * 64 + lane);
  }
#pragma unroll
  for (int t = 0; t < 32; t++) {
    const uint64_t r = (uint64_t)(((unsigned __int128)mix64((uint64_t)wave * 32 + t + salt) * n_rows) >> 64);
    __builtin_nontemporal_store(v[t] * 0.999f, tab + r * 64 + lane);
  }
}
int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 8;
  const double gb = argc > 2 ? atof(argv[2]) : 24.0;
  const int rounds = argc > 3 ? atoi(argv[3]) : 4;
  const uint64_t n_rows = (uint64_t)(gb * 1e9 / 256.0);
  const uint32_t n_ex = 1u << 20;
  std::vector<float*> tabs(T);
  for (int i = 0; i < T; i++) { CK(hipMalloc(&tabs[i], n_rows * 256)); CK(hipMemset(tabs[i], 0, n_rows * 256)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> best(T, 1e9), sum(T, 0);
  for (int r = 0; r < rounds + 1; r++)
    for (int i = 0; i < T; i++) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_rows, dim3(n_ex / 4), dim3(256), 0, 0, tabs[i], n_rows, n_ex, (uint64_t)r * 977 + 1);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) { sum[i] += ms; if (ms < best[i]) best[i] = ms; }
    }
  for (int i = 0; i < T; i++)
    printf("table %d at %p: mean %.3f ms  -> %.2f TB/s of rows read + written\n", i, (void*)tabs[i], sum[i] / rounds,
           (double)n_ex * 32 * 512 / (sum[i] / rounds * 1e-3) / 1e12);
  return 0;
}

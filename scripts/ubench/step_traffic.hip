// the headline step's traffic shape -- per example 32 random 256-byte rows read-modify-written (non-temporal) + 32 random 4-byte weights
// read-modify-written -- with the ROW table allocated plain or uncached (hipDeviceMallocUncached) and the weights plain, to see whether the
// 4-byte weight gathers (64-byte sectors: 15 % of the step's traffic) can be kept in the 256 MB Infinity Cache once the rows stop passing
// through it.   hipcc --offload-arch=gfx950 -O3 -o step_traffic step_traffic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__host__ __device__ inline unsigned long long mix(unsigned long long x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x; }
template <bool NT>
__global__ void __launch_bounds__(256) k_step(float* V, float* w, unsigned long long n, unsigned n_ex, unsigned long long salt, int with_w) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= n_ex) return;
  unsigned long long id = 0; float wv = 0.f;
  if (lane < 32) { id = (unsigned long long)(((unsigned __int128)mix((unsigned long long)e * 32 + lane + salt) * n) >> 64); if (with_w) wv = w[id]; }
  float v[32];
#pragma unroll
  for (int t = 0; t < 32; t++) {
    const unsigned long long j = __shfl(id, t);
    v[t] = NT ? __builtin_nontemporal_load(V + j * 64 + lane) : V[j * 64 + lane];
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 32; t++) s += v[t];
  if (lane < 32 && with_w) w[id] = wv * 0.999f + 1e-9f * s;
#pragma unroll
  for (int t = 0; t < 32; t++) {
    const unsigned long long j = __shfl(id, t);
    if (NT) __builtin_nontemporal_store(v[t] * 0.999f, V + j * 64 + lane); else V[j * 64 + lane] = v[t] * 0.999f;
  }
}
int main() {
  const unsigned long long n = 100000000ull;      // 25.6 GB of rows, 400 MB of weights
  const unsigned n_ex = 262144;
  float* w; CHK(hipMalloc(&w, n * 4)); CHK(hipMemset(w, 0, n * 4));
  for (int kind = 0; kind < 2; kind++) {
    float* V = nullptr;
    if (kind == 0) CHK(hipMalloc(&V, n * 256));
    else { hipError_t e = hipExtMallocWithFlags((void**)&V, n * 256, hipDeviceMallocUncached); if (e != hipSuccess) { printf("uncached allocation: %s\n", hipGetErrorString(e)); return 0; } }
    CHK(hipMemset(V, 0, n * 256));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (int nt = 1; nt >= 0; nt--)
      for (int with_w = 1; with_w >= 0; with_w--) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
          CHK(hipEventRecord(a));
          for (int it = 0; it < 4; it++) {
            if (nt) hipLaunchKernelGGL(k_step<true>, dim3(n_ex / 4), dim3(256), 0, 0, V, w, n, n_ex, (unsigned long long)(rep * 4 + it) * 7919, with_w);
            else hipLaunchKernelGGL(k_step<false>, dim3(n_ex / 4), dim3(256), 0, 0, V, w, n, n_ex, (unsigned long long)(rep * 4 + it) * 7919, with_w);
          }
          CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
          float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (rep && ms < best) best = ms;
        }
        printf("rows %-9s %-12s weights %-3s: %7.1f us per 262144 examples = %6.1f M examples/s\n", kind ? "uncached" : "plain", nt ? "non-temporal" : "temporal", with_w ? "yes" : "no",
               best / 4 * 1e3, n_ex / (best / 4 * 1e-3) / 1e6);
      }
    CHK(hipFree(V));
  }
  return 0;
}

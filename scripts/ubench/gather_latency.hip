// dependent random 256-byte row gathers from a big table (what a small batch's chain is made of): ns per trip by how the table was allocated
// (hipMalloc vs hipMemCreate chunks mapped into one range, as fmx_create's arena does) and by table size.
// hipcc --offload-arch=gfx950 -O3 -o gather_latency gather_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fill(float* t, size_t rows) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * 64; i += (size_t)gridDim.x * blockDim.x) t[i] = (float)(i % 97) * 1e-3f;
}
// every wavefront walks its own chain: next row = hash(sum of the row just read)
__global__ void k_chase(const float* t, size_t rows, unsigned steps, unsigned long long* out, unsigned long long salt) {
  const unsigned lane = threadIdx.x & 63u, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  unsigned long long r = (wave * 0x9E3779B97F4A7C15ull + salt) % rows;
  float acc = 0.f;
  const unsigned long long t0 = wall_clock64();
  for (unsigned s = 0; s < steps; s++) {
    const float v = __builtin_nontemporal_load(t + r * 64 + lane);
    acc += v;
    const unsigned u = __float_as_uint(__shfl(v, 0)) ;
    unsigned long long x = (r + 1) * 0xD6E8FEB86659FD93ull + u + s;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    r = x % rows;
  }
  const unsigned long long t1 = wall_clock64();
  if (lane == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = (unsigned long long)acc; }
}
static float* vmm_alloc(size_t bytes) {
  hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  const size_t CH = (size_t)1 << 30; const size_t n = (bytes + CH - 1) / CH;
  void* va = nullptr; CHK(hipMemAddressReserve(&va, n * CH, CH, nullptr, 0));
  for (size_t i = 0; i < n; i++) { hipMemGenericAllocationHandle_t h; CHK(hipMemCreate(&h, CH, &prop, 0)); CHK(hipMemMap((char*)va + i * CH, CH, 0, h, 0)); }
  hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  CHK(hipMemSetAccess(va, n * CH, &acc, 1));
  return (float*)va;
}
int main() {
  unsigned long long* out; CHK(hipMalloc(&out, 2 * 8192 * 8));
  std::vector<unsigned long long> h(2 * 8192);
  for (int kind = 0; kind < 2; kind++) {
    for (size_t gb : {1, 8, 24}) {
      const size_t bytes = gb << 30, rows = bytes / 256;
      float* t = nullptr;
      if (kind == 0) CHK(hipMalloc(&t, bytes)); else t = vmm_alloc(bytes);
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, t, rows); CHK(hipDeviceSynchronize());
      for (unsigned waves : {1u, 64u, 512u, 4096u}) {
        const unsigned steps = 256;
        hipLaunchKernelGGL(k_chase, dim3((waves + 3) / 4), dim3(waves < 4 ? 64 * waves : 256), 0, 0, t, rows, steps, out, 12345ull); CHK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_chase, dim3((waves + 3) / 4), dim3(waves < 4 ? 64 * waves : 256), 0, 0, t, rows, steps, out, 777ull); CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h.data(), out, 2 * waves * 8, hipMemcpyDeviceToHost));
        double mean = 0; for (unsigned w = 0; w < waves; w++) mean += (double)h[2 * w]; mean = mean / waves * 10.0 / steps;
        printf("%-10s %3zu GiB  %5u wavefronts: %7.0f ns per dependent 256-B row gather\n", kind ? "vmm-1GiB" : "hipMalloc", gb, waves, mean);
      }
      if (kind == 0) CHK(hipFree(t));
    }
  }
  return 0;
}

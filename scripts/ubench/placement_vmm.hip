// micro-benchmark (round 3): WHAT distinguishes a fast factor table from a slow one?  placement.hip showed two rate classes for
// same-sized hipMalloc tables of one process, with identical virtual alignment.  Hypothesis: the physical backing -- hipMalloc carves
// a 24 GB table out of whatever physical extents are free, and extents smaller than a 2 MB huge page cost the random row gather
// TLB reach.  Test: build tables from PHYSICAL chunks of a chosen size through the virtual-memory API (hipMemCreate + hipMemMap:
// every chunk is one physically contiguous allocation) and run placement.hip's kernel over them, next to plain hipMalloc tables.
//   hipcc --offload-arch=gfx950 -O3 placement_vmm.hip -o placement_vmm ; ./placement_vmm [GB each=24] [rounds=3]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x;
}
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
struct Tab { float* p; const char* kind; size_t chunk; };
static float* vmm_table(size_t bytes, size_t chunk, size_t va_align) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  if (chunk % gran) chunk = (chunk / gran + 1) * gran;
  const size_t total = (bytes + chunk - 1) / chunk * chunk;
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, total, va_align, nullptr, 0));
  for (size_t off = 0; off < total; off += chunk) {
    hipMemGenericAllocationHandle_t hnd;
    CK(hipMemCreate(&hnd, chunk, &prop, 0));
    CK(hipMemMap((char*)va + off, chunk, 0, hnd, 0));
    CK(hipMemRelease(hnd));
  }
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, total, &acc, 1));
  return (float*)va;
}
int main(int argc, char** argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 24.0;
  const int rounds = argc > 2 ? atoi(argv[2]) : 3;
  const uint64_t n_rows = (uint64_t)(gb * 1e9 / 256.0);
  const size_t bytes = n_rows * 256;
  const uint32_t n_ex = 1u << 20;
  { hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("allocation granularity: minimum %zu, recommended %zu bytes\n", gmin, grec); }
  std::vector<Tab> tabs;
  // fragment the free list a little first, the way earlier work of a process does: many mid-sized blocks, every other one freed
  { std::vector<void*> junk(64);
    for (auto& j : junk) CK(hipMalloc(&j, (size_t)192 << 20));
    for (size_t i = 0; i < junk.size(); i += 2) CK(hipFree(junk[i]));
    for (int i = 0; i < 3; i++) { float* p; CK(hipMalloc(&p, bytes)); tabs.push_back({p, "hipMalloc", 0}); }
    for (size_t i = 1; i < junk.size(); i += 2) CK(hipFree(junk[i])); }
  for (size_t chunk : {(size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30})
    for (int rep = 0; rep < 2; rep++) tabs.push_back({vmm_table(bytes, chunk, chunk), "hipMemCreate chunks of", chunk});
  for (auto& t : tabs) CK(hipMemset(t.p, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> sum(tabs.size(), 0);
  for (int r = 0; r < rounds + 1; r++)
    for (size_t i = 0; i < tabs.size(); i++) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_rows, dim3(n_ex / 4), dim3(256), 0, 0, tabs[i].p, n_rows, n_ex, (uint64_t)r * 977 + 1);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) sum[i] += ms;
    }
  for (size_t i = 0; i < tabs.size(); i++)
    printf("table %zu %-22s %6zu MB at %p: mean %.3f ms -> %.2f TB/s of rows read + written\n", i, tabs[i].kind, tabs[i].chunk >> 20,
           (void*)tabs[i].p, sum[i] / rounds, (double)n_ex * 32 * 512 / (sum[i] / rounds * 1e-3) / 1e12);
  return 0;
}

// what a device allocation costs on this box: hipMalloc / hipFree by size, and the same bytes through the virtual-memory API
// (hipMemCreate + hipMemMap + hipMemSetAccess), before and after a pool of 1 GiB chunks has been taken and returned (what fmx_create's
// placement does).  Build: hipcc --offload-arch=gfx950 -O2 -o alloc_cost alloc_cost.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void t_malloc(const char* tag, size_t bytes, int reps) {
  std::vector<void*> p(reps, nullptr);
  double t0 = now();
  for (int i = 0; i < reps; i++) if (hipMalloc(&p[i], bytes) != hipSuccess) { printf("%s: hipMalloc failed\n", tag); return; }
  double t1 = now();
  for (int i = 0; i < reps; i++) hipMemsetAsync(p[i], 0, 4096, 0);
  hipDeviceSynchronize();
  double t2 = now();
  for (int i = 0; i < reps; i++) hipFree(p[i]);
  double t3 = now();
  printf("%-28s %6.2f GiB x %d: hipMalloc %8.3f ms each, first touch %7.3f ms, hipFree %8.3f ms each\n", tag, bytes / 1073741824.0, reps,
         1e3 * (t1 - t0) / reps, 1e3 * (t2 - t1), 1e3 * (t3 - t2) / reps);
}
static void t_vmm(const char* tag, size_t bytes, int reps) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0;
  hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
  bytes = (bytes + gran - 1) / gran * gran;
  std::vector<void*> va(reps, nullptr); std::vector<hipMemGenericAllocationHandle_t> hd(reps);
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  double t0 = now();
  for (int i = 0; i < reps; i++) {
    if (hipMemAddressReserve(&va[i], bytes, gran, nullptr, 0) != hipSuccess || hipMemCreate(&hd[i], bytes, &prop, 0) != hipSuccess ||
        hipMemMap(va[i], bytes, 0, hd[i], 0) != hipSuccess || hipMemSetAccess(va[i], bytes, &acc, 1) != hipSuccess) { printf("%s: vmm failed\n", tag); return; }
  }
  double t1 = now();
  for (int i = 0; i < reps; i++) hipMemsetAsync(va[i], 0, 4096, 0);
  hipDeviceSynchronize();
  double t2 = now();
  for (int i = 0; i < reps; i++) { hipMemUnmap(va[i], bytes); hipMemRelease(hd[i]); hipMemAddressFree(va[i], bytes); }
  double t3 = now();
  printf("%-28s %6.2f GiB x %d: vmm alloc %8.3f ms each, first touch %7.3f ms, free %8.3f ms each (granularity %zu)\n", tag, bytes / 1073741824.0, reps,
         1e3 * (t1 - t0) / reps, 1e3 * (t2 - t1), 1e3 * (t3 - t2) / reps, gran);
}
int main() {
  hipSetDevice(0); hipFree(0);
  const size_t G = 1ull << 30;
  t_malloc("fresh process", G, 4);
  t_malloc("again", G, 4);
  t_malloc("again 4 GiB", 4 * G, 2);
  t_malloc("64 MiB", 64ull << 20, 8);
  t_vmm("vmm fresh", G, 4);
  t_vmm("vmm 4 GiB", 4 * G, 2);
  {  // a pool of 96 chunks of 1 GiB taken through the virtual-memory API and returned (fmx_create's placement)
    t_vmm("vmm pool of 96 x 1 GiB", G, 96);
  }
  t_malloc("after the pool", G, 4);
  t_malloc("after the pool, 4 GiB", 4 * G, 2);
  t_vmm("vmm after the pool", G, 4);
  // a big resident table next to it (the 26 GB parameter table)
  void* big = nullptr; double t0 = now(); hipMalloc(&big, 26 * G); printf("hipMalloc 26 GiB: %.3f ms\n", 1e3 * (now() - t0));
  t_malloc("next to 26 GiB", G, 4);
  t0 = now(); hipFree(big); printf("hipFree 26 GiB: %.3f ms\n", 1e3 * (now() - t0));
  return 0;
}

// micro-benchmark (round 3, after placement_chunks.hip): the rate class is a property of a whole ALLOCATION (36 separate 1 GB chunks
// of one pool: all 4.9 TB/s, two hipMalloc tables next to them: 4.9 and 6.0).  What kind of property -- the physical memory, the
// virtual address, or the order of allocation?
//   (a) tables allocated in a sequence (hipMalloc x2, a pool of hipMemCreate chunks, hipMalloc x2), each probed as a whole and in
//       1 GB pieces;
//   (b) the SAME physical chunks of the pool unmapped and mapped again at other virtual addresses;
//   (c) a hipMalloc table freed and allocated again (same size: usually the same physical memory and address).
//   hipcc --offload-arch=gfx950 -O3 placement_order.hip -o placement_order ; ./placement_order [table GB=24]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x;
}
__global__ void __launch_bounds__(256) k_rows(float* __restrict__ tab, uint64_t row0, uint64_t n_rows, uint32_t n_ex, uint64_t salt) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= n_ex) return;
  float v[32]; uint64_t at[32];
#pragma unroll
  for (int t = 0; t < 32; t++) {
    const uint64_t r = row0 + (uint64_t)(((unsigned __int128)mix64((uint64_t)wave * 32 + t + salt) * n_rows) >> 64);
    at[t] = r * 64 + lane;
    v[t] = __builtin_nontemporal_load(tab + at[t]);
  }
#pragma unroll
  for (int t = 0; t < 32; t++) __builtin_nontemporal_store(v[t] * 0.999f, tab + at[t]);
}
static hipEvent_t e0, e1;
static double probe(float* tab, uint64_t row0, uint64_t n_rows, uint32_t n_ex, int rounds = 3) {
  double sum = 0;
  for (int r = 0; r < rounds + 1; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_rows, dim3(n_ex / 4), dim3(256), 0, 0, tab, row0, n_rows, n_ex, (uint64_t)r * 977 + 1);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r) sum += ms;
  }
  return (double)n_ex * 32 * 512 / (sum / rounds * 1e-3) / 1e12;
}
static void report(const char* name, float* p, size_t bytes) {
  const uint64_t rows = bytes / 256, per_gb = ((size_t)1 << 30) / 256;
  printf("%-34s at %p: whole %.2f TB/s; per GB:", name, (void*)p, probe(p, 0, rows, 1u << 20));
  for (uint64_t r0 = 0; r0 + per_gb <= rows; r0 += per_gb * 4) printf(" %.2f", probe(p, r0, per_gb, 1u << 18, 2));     // every fourth GB
  printf("\n");
}
int main(int argc, char** argv) {
  const size_t gb = argc > 1 ? atoi(argv[1]) : 24;
  const size_t bytes = gb << 30, chunk = (size_t)1 << 30;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  printf("device memory: %.1f GB free of %.1f\n", fr / 1e9, tot / 1e9);
  // (a) a sequence of allocations
  float* m[4];
  for (int i = 0; i < 2; i++) { CK(hipMalloc(&m[i], bytes)); CK(hipMemset(m[i], 0, bytes)); }
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, bytes, chunk, nullptr, 0));
  std::vector<hipMemGenericAllocationHandle_t> hnd(gb);
  for (size_t c = 0; c < gb; c++) { CK(hipMemCreate(&hnd[c], chunk, &prop, 0)); CK(hipMemMap((char*)va + chunk * c, chunk, 0, hnd[c], 0)); }
  CK(hipMemSetAccess(va, bytes, &acc, 1));
  CK(hipMemset(va, 0, bytes));
  for (int i = 2; i < 4; i++) { CK(hipMalloc(&m[i], bytes)); CK(hipMemset(m[i], 0, bytes)); }
  report("hipMalloc #0 (first)", m[0], bytes);
  report("hipMalloc #1", m[1], bytes);
  report("pool of 1 GB chunks (third)", (float*)va, bytes);
  report("hipMalloc #2 (after the pool)", m[2], bytes);
  report("hipMalloc #3", m[3], bytes);
  // (b) the pool's physical chunks at other virtual addresses
  CK(hipMemUnmap(va, bytes));
  const uintptr_t hints[4] = {0x100000000000ull, 0x200000000000ull, (uintptr_t)m[1] + ((size_t)64 << 30), 0};
  for (int t = 0; t < 4; t++) {
    void* vb = nullptr;
    if (hipMemAddressReserve(&vb, bytes, chunk, (void*)hints[t], 0) != hipSuccess) { printf("reserve with hint %p refused\n", (void*)hints[t]); continue; }
    for (size_t c = 0; c < gb; c++) CK(hipMemMap((char*)vb + chunk * c, chunk, 0, hnd[(t & 1) ? gb - 1 - c : c], 0));   // (odd t: chunks in reverse order)
    CK(hipMemSetAccess(vb, bytes, &acc, 1));
    char nm[80]; snprintf(nm, sizeof nm, "pool remapped (hint %p%s)", (void*)hints[t], (t & 1) ? ", reversed" : "");
    report(nm, (float*)vb, bytes);
    CK(hipMemUnmap(vb, bytes));
    CK(hipMemAddressFree(vb, bytes));
  }
  // (c) free and allocate again
  for (int rep = 0; rep < 3; rep++) {
    for (int i = 0; i < 4; i++) CK(hipFree(m[i]));
    for (int i = 0; i < 4; i++) { CK(hipMalloc(&m[i], bytes)); CK(hipMemset(m[i], 0, bytes)); }
    printf("freed and allocated again (%d):", rep);
    for (int i = 0; i < 4; i++) printf("  #%d %p %.2f", i, (void*)m[i], probe(m[i], 0, bytes / 256, 1u << 20));
    printf("\n");
  }
  // (d) the same tables probed in the opposite order (is it the table, or what ran before it?)
  printf("probed in reverse order:");
  for (int i = 3; i >= 0; i--) printf("  #%d %.2f", i, probe(m[i], 0, bytes / 256, 1u << 20));
  printf("\n");
  return 0;
}

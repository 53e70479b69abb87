// micro-benchmark (round 3, after placement_order.hip): the rate class of a table is PHYSICAL (the same chunks mapped at other
// virtual addresses keep their rate; a freed and re-allocated table at the same address changes it) and it is a property of the SPREAD
// of a table: every 1 GB piece of every table runs at 4.9 TB/s, whole tables at 4.9 ... 6.0.  Hypothesis: physical memory falls into a
// few classes (high physical address bits: ranks / stack ids of the HBM dies) and random rows over memory of ONE class conflict more
// than rows spread over several.  Test: take (nearly) all of the device's memory as 1 GB chunks, classify every chunk by probing it
// TOGETHER with a reference chunk (same class: the one-chunk rate; another class: more), print the class map in allocation order,
// then probe 24-chunk tables composed of one class / two / three, evenly and unevenly.
//   hipcc --offload-arch=gfx950 -O3 placement_classes.hip -o placement_classes ; ./placement_classes [leave GB=12] [chunk MB=1024]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x;
}
// random 256-byte rows over the chunks list[0 .. n_list) of the pool (rows_per_chunk rows each), read and written back
__global__ void __launch_bounds__(256) k_rows(float* __restrict__ tab, const uint32_t* __restrict__ list, uint32_t n_list, uint64_t rows_per_chunk,
                                              uint32_t shift, uint32_t n_ex, uint64_t salt) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= n_ex) return;
  float v[32]; uint64_t at[32];
#pragma unroll
  for (int t = 0; t < 32; t++) {
    const uint64_t r = (uint64_t)(((unsigned __int128)mix64((uint64_t)wave * 32 + t + salt) * (rows_per_chunk * n_list)) >> 64);
    const uint64_t c = list[r >> shift];                                  // (rows_per_chunk = 1 << shift: no 64-bit division in the probe)
    at[t] = ((c << shift) + (r & (rows_per_chunk - 1))) * 64 + lane;
    v[t] = __builtin_nontemporal_load(tab + at[t]);
  }
#pragma unroll
  for (int t = 0; t < 32; t++) __builtin_nontemporal_store(v[t] * 0.999f, tab + at[t]);
}
static hipEvent_t e0, e1;
static uint32_t* d_list;
static float* tab;
static uint64_t rows_per_chunk;
static double probe(const std::vector<uint32_t>& list, uint32_t n_ex = 1u << 18, int rounds = 2) {
  CK(hipMemcpy(d_list, list.data(), list.size() * 4, hipMemcpyHostToDevice));
  double sum = 0;
  for (int r = 0; r < rounds + 1; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_rows, dim3(n_ex / 4), dim3(256), 0, 0, tab, d_list, (uint32_t)list.size(), rows_per_chunk, (uint32_t)__builtin_ctzll(rows_per_chunk), n_ex, (uint64_t)r * 977 + 1);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r) sum += ms;
  }
  return (double)n_ex * 32 * 512 / (sum / rounds * 1e-3) / 1e12;
}
// a STREAM instead of random rows: copy `n4` float4 from src to dst (grid-stride, non-temporal) -- does a sequential read + write mix care
// which classes the two arrays lie in?
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ src, f4* __restrict__ dst, uint64_t n4) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) {
    const f4 v = __builtin_nontemporal_load(src + i);
    __builtin_nontemporal_store(v, dst + i);
  }
}
static double copy_rate(const float* src, float* dst, size_t bytes) {
  double sum = 0;
  for (int r = 0; r < 4; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, (const f4*)src, (f4*)dst, (uint64_t)(bytes / 16));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r) sum += ms;
  }
  return 2.0 * bytes / (sum / 3 * 1e-3) / 1e12;                          // TB/s read + written
}
int main(int argc, char** argv) {
  const size_t leave = (size_t)(argc > 1 ? atoi(argv[1]) : 12) << 30;
  const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 20;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  const int pool = (int)((fr - leave) / chunk);
  printf("device memory: %.1f GB free of %.1f; pool of %d chunks of %zu MB\n", fr / 1e9, tot / 1e9, pool, chunk >> 20);
  CK(hipMalloc(&d_list, 4096 * 4));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, chunk * pool, (size_t)1 << 30, nullptr, 0));
  int got = 0;
  for (int c = 0; c < pool; c++) {
    hipMemGenericAllocationHandle_t hnd;
    if (hipMemCreate(&hnd, chunk, &prop, 0) != hipSuccess) break;
    CK(hipMemMap((char*)va + chunk * c, chunk, 0, hnd, 0));
    CK(hipMemRelease(hnd));
    got++;
  }
  printf("got %d chunks\n", got);
  CK(hipMemSetAccess(va, chunk * got, &acc, 1));
  CK(hipMemset(va, 0, chunk * got));
  tab = (float*)va;
  rows_per_chunk = chunk / 256;
  if (rows_per_chunk & (rows_per_chunk - 1)) { printf("chunk size must be a power of two\n"); return 1; }
  // the one-chunk rate
  double base = 0;
  for (int c = 0; c < 8 && c < got; c++) base += probe({(uint32_t)(c * (got / 8))});
  base /= std::min(8, got);
  printf("one chunk alone: %.2f TB/s\n", base);
  // classes: a chunk belongs to the class of the first reference chunk it does NOT speed up with
  std::vector<int> cls(got, -1);
  std::vector<uint32_t> refs;
  for (int c = 0; c < got; c++) {
    for (size_t r = 0; r < refs.size() && cls[c] < 0; r++) {
      if ((uint32_t)c == refs[r]) { cls[c] = (int)r; break; }
      if (probe({refs[r], (uint32_t)c}) < base * 1.03) cls[c] = (int)r;
    }
    if (cls[c] < 0) { if (refs.size() < 16) { refs.push_back((uint32_t)c); cls[c] = (int)refs.size() - 1; } else cls[c] = 15; }
  }
  printf("classes in allocation order (%zu classes):\n", refs.size());
  for (int c = 0; c < got; c++) { printf("%c", 'A' + cls[c]); if (c % 64 == 63) printf("\n"); }
  printf("\n");
  std::vector<std::vector<uint32_t>> of(refs.size());
  for (int c = 0; c < got; c++) of[cls[c]].push_back((uint32_t)c);
  for (size_t k = 0; k < of.size(); k++) printf("class %c: %zu chunks\n", 'A' + (int)k, of[k].size());
  // pair rates between the references
  printf("two chunks of classes (X, Y), TB/s:\n");
  for (size_t a = 0; a < refs.size(); a++) {
    for (size_t b = 0; b < refs.size(); b++) printf(" %.2f", a == b ? (of[a].size() > 1 ? probe({of[a][0], of[a][1]}) : 0.0) : probe({refs[a], refs[b]}));
    printf("\n");
  }
  // tables of T chunks composed of given shares of the classes
  const int T = (int)(((size_t)24 << 30) / chunk);
  auto compose = [&](std::vector<int> share, const char* name) {
    std::vector<uint32_t> list;
    std::vector<size_t> used(of.size(), 0);
    int total = 0; for (int s : share) total += s;
    for (int i = 0; (int)list.size() < T; i++) {
      int pos = i % total, k = 0;
      while (pos >= share[k]) { pos -= share[k]; k++; }
      if ((size_t)k >= of.size() || used[k] >= of[k].size()) { printf("%-40s not enough chunks\n", name); return; }
      list.push_back(of[k][used[k]++]);
    }
    printf("%-40s %.2f TB/s\n", name, probe(list, 1u << 20, 3));
  };
  compose({1}, "24 GB of class A");
  if (of.size() >= 2) {
    compose({0, 1}, "24 GB of class B");
    compose({1, 1}, "A : B = 1 : 1");
    compose({3, 1}, "A : B = 3 : 1");
    compose({7, 1}, "A : B = 7 : 1");
  }
  if (of.size() >= 3) {
    compose({0, 0, 1}, "24 GB of class C");
    compose({1, 1, 1}, "A : B : C = 1 : 1 : 1");
    compose({2, 1, 1}, "A : B : C = 2 : 1 : 1");
    compose({1, 0, 1}, "A : C = 1 : 1");
    compose({0, 1, 1}, "B : C = 1 : 1");
  }
  if (of.size() >= 4) {
    compose({1, 1, 1, 1}, "A : B : C : D = 1 : 1 : 1 : 1");
    compose({0, 0, 0, 1}, "24 GB of class D");
  }
  if (of.size() >= 6) compose({1, 1, 1, 1, 1, 1}, "A .. F evenly");
  if (of.size() >= 8) compose({1, 1, 1, 1, 1, 1, 1, 1}, "A .. H evenly");
  // streams: a 1 GB chunk copied into another chunk of the same class / of another class
  if (of.size() >= 2 && of[0].size() >= 3 && of[1].size() >= 2) {
    auto at = [&](uint32_t c) { return tab + (size_t)c * rows_per_chunk * 64; };
    printf("copy of one chunk (1 GB read + 1 GB written), TB/s:  A -> A %.2f   A -> B %.2f   B -> B %.2f   B -> A %.2f\n",
           copy_rate(at(of[0][0]), at(of[0][1]), chunk), copy_rate(at(of[0][0]), at(of[1][0]), chunk),
           copy_rate(at(of[1][0]), at(of[1][1]), chunk), copy_rate(at(of[1][1]), at(of[0][2]), chunk));
    // the same amount of data as two half-chunk copies running over chunks of both classes at once: [A lo -> A' lo] then interleaved is not
    // expressible with one kernel; instead: source = first halves of an A and a B chunk alternately is what a balanced buffer looks like
  }
  { std::vector<uint32_t> all(got); for (int c = 0; c < got; c++) all[c] = (uint32_t)c; printf("%-40s %.2f TB/s\n", "the whole pool", probe(all, 1u << 20, 3)); }
  return 0;
}

// micro-benchmark (round 3): is memory mapped through the virtual-memory API (hipMemCreate + hipMemMap, the arena of fmx_create) as
// good as hipMalloc memory when the access pattern is LATENCY-bound?  (Two feature shards on one device ran 8 % slower out of arenas,
// one shard and one unsharded handle did not: the shard kernels keep 16 rows per wavefront in flight, not 32.)
// Tables of `gb` GiB: hipMalloc; hipMemCreate chunks of 1 GiB / 4 GiB / one chunk, each mapped on its own and with one hipMemSetAccess
// over the whole range.  Patterns: (1) dependent chain of random 4-byte loads (one wavefront, 64 chains): ns per step;
// (2) read-only random row gather, R rows per wavefront in flight (R = 4, 8, 16, 32), 2^18 wavefronts: ms and TB/s.
//   hipcc --offload-arch=gfx950 -O3 placement_latency.hip -o placement_latency ; ./placement_latency [GiB=12]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x;
}
__global__ void __launch_bounds__(64) k_chase(const uint32_t* __restrict__ tab, uint64_t n_words, uint32_t steps, uint32_t* out) {
  uint64_t at = mix64(threadIdx.x + 1) % n_words;
  uint32_t acc = 0;
  for (uint32_t s = 0; s < steps; s++) {
    const uint32_t v = __builtin_nontemporal_load(tab + at);              // (the table holds zeros: the next address depends on the value read)
    acc += v;
    at = mix64(at + v + s) % n_words;
  }
  out[threadIdx.x] = acc;
}
template <int R>
__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ tab, uint64_t n_rows, uint32_t n_ex, uint64_t salt, float* out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= n_ex) return;
  float v[R];
#pragma unroll
  for (int t = 0; t < R; t++) {
    const uint64_t r = (uint64_t)(((unsigned __int128)mix64((uint64_t)wave * R + t + salt) * n_rows) >> 64);
    v[t] = __builtin_nontemporal_load(tab + r * 64 + lane);
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < R; t++) s += v[t];
  if (s == 12345.f) out[wave] = s;
}
static hipEvent_t e0, e1;
template <int R> static double gather_ms(const float* tab, uint64_t n_rows, float* out) {
  const uint32_t n_ex = 1u << 18;
  double sum = 0;
  for (int r = 0; r < 4; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_gather<R>, dim3(n_ex / 4), dim3(256), 0, 0, tab, n_rows, n_ex, (uint64_t)r * 977 + 1, out);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r) sum += ms;
  }
  return sum / 3;
}
static float* vmm_table(size_t bytes, size_t chunk, bool access_per_chunk) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, bytes, (size_t)1 << 30, nullptr, 0));
  for (size_t off = 0; off < bytes; off += chunk) {
    hipMemGenericAllocationHandle_t hnd;
    CK(hipMemCreate(&hnd, chunk, &prop, 0));
    CK(hipMemMap((char*)va + off, chunk, 0, hnd, 0));
    CK(hipMemRelease(hnd));
    if (access_per_chunk) CK(hipMemSetAccess((char*)va + off, chunk, &acc, 1));
  }
  if (!access_per_chunk) CK(hipMemSetAccess(va, bytes, &acc, 1));
  return (float*)va;
}
int main(int argc, char** argv) {
  const size_t gb = argc > 1 ? atoi(argv[1]) : 12;
  const size_t bytes = gb << 30;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* out; CK(hipMalloc(&out, (size_t)4 << 20));
  struct T { const char* name; float* p; };
  std::vector<T> tabs;
  { float* p; CK(hipMalloc(&p, bytes)); tabs.push_back({"hipMalloc", p}); }
  tabs.push_back({"chunks of 1 GiB, access per chunk", vmm_table(bytes, (size_t)1 << 30, true)});
  tabs.push_back({"chunks of 1 GiB, access once", vmm_table(bytes, (size_t)1 << 30, false)});
  tabs.push_back({"chunks of 4 GiB", vmm_table(bytes, (size_t)4 << 30, false)});
  tabs.push_back({"one chunk", vmm_table(bytes, bytes, false)});
  { float* p; CK(hipMalloc(&p, bytes)); tabs.push_back({"hipMalloc (second)", p}); }
  for (auto& t : tabs) CK(hipMemset(t.p, 0, bytes));
  const uint64_t n_rows = bytes / 256;
  for (int rep = 0; rep < 2; rep++)
    for (auto& t : tabs) {
      const uint32_t steps = 20000;
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, (const uint32_t*)t.p, (uint64_t)(bytes / 4), steps, (uint32_t*)out);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      const double g4 = gather_ms<4>(t.p, n_rows, out), g8 = gather_ms<8>(t.p, n_rows, out), g16 = gather_ms<16>(t.p, n_rows, out), g32 = gather_ms<32>(t.p, n_rows, out);
      auto tbs = [&](double ms_, int R) { return (double)(1u << 18) * R * 256 / (ms_ * 1e-3) / 1e12; };
      printf("%-36s chase %.0f ns/step | gather R=4 %.3f ms (%.2f TB/s)  R=8 %.3f (%.2f)  R=16 %.3f (%.2f)  R=32 %.3f (%.2f)\n", t.name, ms * 1e6 / steps,
             g4, tbs(g4, 4), g8, tbs(g8, 8), g16, tbs(g16, 16), g32, tbs(g32, 32));
    }
  return 0;
}

// micro-benchmark (round 3, follow-up of placement_vmm.hip): is the rate class of a factor table a property of its PARTS?
// placement_vmm.hip found tables of every construction in every class, with intermediate levels (4.9 / 5.3 / 5.6 / 6.0 TB/s) -- what a
// mixture of fast and slow regions would look like.  Test: one pool of `pool` chunks of `chunk_mb` MB (hipMemCreate, mapped back to back
// into one reserved range), (1) the random row read + write-back restricted to ONE chunk at a time, (2) the same over the whole pool,
// (3) two tables of `keep` chunks each mapped from the pool's fastest / slowest chunks (a physical chunk may be mapped twice), probed
// as whole tables.  If (1) is bimodal and (3) separates, fmx_create can SELECT its table instead of drawing it.
//   hipcc --offload-arch=gfx950 -O3 placement_chunks.hip -o placement_chunks ; ./placement_chunks [chunk MB=1024] [pool=36] [keep=24]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x;
}
// every wavefront reads 32 random 256-byte rows of tab[row0 .. row0 + n_rows) and writes them back scaled
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
static double probe(float* tab, uint64_t row0, uint64_t n_rows, uint32_t n_ex, int rounds) {
  double sum = 0;
  for (int r = 0; r < rounds + 1; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_rows, dim3(n_ex / 4), dim3(256), 0, 0, tab, row0, n_rows, n_ex, (uint64_t)r * 977 + 1);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r) sum += ms;
  }
  return (double)n_ex * 32 * 512 / (sum / rounds * 1e-3) / 1e12;      // TB/s of rows read + written
}
int main(int argc, char** argv) {
  const size_t chunk = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20;
  const int pool = argc > 2 ? atoi(argv[2]) : 36;
  const int keep = argc > 3 ? atoi(argv[3]) : 24;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, chunk * pool, (size_t)1 << 30, nullptr, 0));
  std::vector<hipMemGenericAllocationHandle_t> hnd(pool);
  for (int c = 0; c < pool; c++) {
    CK(hipMemCreate(&hnd[c], chunk, &prop, 0));
    CK(hipMemMap((char*)va + chunk * c, chunk, 0, hnd[c], 0));
  }
  CK(hipMemSetAccess(va, chunk * pool, &acc, 1));
  CK(hipMemset(va, 0, chunk * pool));
  float* tab = (float*)va;
  const uint64_t rows_per_chunk = chunk / 256;
  // (1) one chunk at a time
  std::vector<double> rate(pool);
  for (int c = 0; c < pool; c++) rate[c] = probe(tab, rows_per_chunk * c, rows_per_chunk, 1u << 18, 3);
  printf("pool of %d chunks of %zu MB; one chunk at a time (TB/s of rows read + written):\n", pool, chunk >> 20);
  for (int c = 0; c < pool; c++) printf("%s%.2f", c ? " " : "  ", rate[c]);
  printf("\n");
  // ... and once more, to see whether a chunk's figure is its own
  std::vector<double> rate2(pool);
  for (int c = 0; c < pool; c++) rate2[c] = probe(tab, rows_per_chunk * c, rows_per_chunk, 1u << 18, 3);
  printf("again:\n");
  for (int c = 0; c < pool; c++) printf("%s%.2f", c ? " " : "  ", rate2[c]);
  printf("\n");
  // (2) the whole pool, and its halves
  printf("whole pool: %.2f   first half: %.2f   second half: %.2f\n", probe(tab, 0, rows_per_chunk * pool, 1u << 20, 3),
         probe(tab, 0, rows_per_chunk * (pool / 2), 1u << 20, 3), probe(tab, rows_per_chunk * (pool / 2), rows_per_chunk * (pool - pool / 2), 1u << 20, 3));
  // (3) tables mapped from the fastest / slowest `keep` chunks
  std::vector<int> order(pool);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return rate[a] + rate2[a] > rate[b] + rate2[b]; });
  for (int which = 0; which < 2; which++) {
    void* vb = nullptr;
    CK(hipMemAddressReserve(&vb, chunk * keep, (size_t)1 << 30, nullptr, 0));
    bool ok = true;
    for (int i = 0; i < keep && ok; i++) {
      const int c = which == 0 ? order[i] : order[pool - 1 - i];
      hipError_t er = hipMemMap((char*)vb + chunk * i, chunk, 0, hnd[c], 0);
      if (er != hipSuccess) { printf("second mapping of a chunk refused (%s): unmapping the pool first is needed\n", hipGetErrorString(er)); ok = false; }
    }
    if (!ok) break;
    CK(hipMemSetAccess(vb, chunk * keep, &acc, 1));
    const double r = probe((float*)vb, 0, rows_per_chunk * keep, 1u << 20, 3);
    printf("table of the %s %d chunks: %.2f TB/s\n", which == 0 ? "FASTEST" : "SLOWEST", keep, r);
    CK(hipMemUnmap(vb, chunk * keep));
    CK(hipMemAddressFree(vb, chunk * keep));
  }
  // (4) for reference: plain hipMalloc tables of the same size as the kept table
  for (int i = 0; i < 2; i++) {
    float* p = nullptr;
    if (hipMalloc(&p, chunk * keep) != hipSuccess) break;
    CK(hipMemset(p, 0, chunk * keep));
    printf("hipMalloc table %d of %zu MB: %.2f TB/s\n", i, (chunk * keep) >> 20, probe(p, 0, rows_per_chunk * keep, 1u << 20, 3));
  }
  return 0;
}

// micro-benchmark: what does ONE random 4-byte read (and read-modify-write) of a large fp32 table cost on gfx950,
// under every load flavour / allocation kind the hardware offers?  This is the `w_j` access of the FM step
// (libfm_amd/csrc/fmx_kernels.h load_w): 32 of them per example, 1.5 % of the algorithmic bytes and, at one
// 128-byte line each, a quarter of the measured traffic.
//
//   hipcc --offload-arch=gfx950 -O3 w_gather.hip -o w_gather
//   ./w_gather [n_floats=800000000] [accesses=268435456] [kind=-1]  # timing table (kind: 0 hipMalloc 1 uncached 2 fine-grained)
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./w_gather     # bytes per access (one counter per pass)
//
// Kernels (each flavour is its own template instance so that rocprofv3 reports it separately):
//   k_gather<F>  : every lane loads table[hash(i)] with load flavour F, 8 loads in flight per lane
//   k_sgather    : the index is wave-uniform -> scalar s_load_dword through the scalar cache (EPI = 1 kernels hold
//                  the id in an SGPR, fmx_kernels.h bcast_u32)
//   k_rmw<F>     : table[hash(i)] += 1 with store flavour F (plain / nt / atomic add without return)
// Allocation kinds: hipMalloc (coarse-grained, cached in L2), hipExtMallocWithFlags uncached / fine-grained.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27; x *= 0x94D049BB133111EBULL;
  x ^= x >> 31;
  return x;
}
__device__ __forceinline__ uint64_t rnd_index(uint64_t i, uint64_t n) {
  return (uint64_t)(((unsigned __int128)mix64(i * 0x9E3779B97F4A7C15ULL + 12345) * n) >> 64);
}

enum { F_PLAIN = 0, F_NT = 1, F_SC0 = 2, F_SC1 = 3, F_SC0SC1 = 4, F_SC0SC1NT = 5 };

// Eight loads and their wait are ONE asm statement: a global load writes its destination register when the data comes
// back, so the compiler must not get the chance to reuse an output (or to read it) before the s_waitcnt -- with one asm
// per load it did both (observed: destination registers recycled as address registers -> memory fault).
#define LOAD8(SUFFIX)                                                                                        \
  asm volatile("global_load_dword %0, %8, off" SUFFIX "\n\tglobal_load_dword %1, %9, off" SUFFIX "\n\t"         \
               "global_load_dword %2, %10, off" SUFFIX "\n\tglobal_load_dword %3, %11, off" SUFFIX "\n\t"       \
               "global_load_dword %4, %12, off" SUFFIX "\n\tglobal_load_dword %5, %13, off" SUFFIX "\n\t"       \
               "global_load_dword %6, %14, off" SUFFIX "\n\tglobal_load_dword %7, %15, off" SUFFIX "\n\t"       \
               "s_waitcnt vmcnt(0)"                                                                          \
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])   \
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory")
template <int F> __device__ __forceinline__ void load8(const float* (&p)[8], float (&v)[8]) {
  if constexpr (F == F_PLAIN)     LOAD8("");
  if constexpr (F == F_NT)        LOAD8(" nt");
  if constexpr (F == F_SC0)       LOAD8(" sc0");
  if constexpr (F == F_SC1)       LOAD8(" sc1");
  if constexpr (F == F_SC0SC1)    LOAD8(" sc0 sc1");
  if constexpr (F == F_SC0SC1NT)  LOAD8(" sc0 sc1 nt");
}

template <int F>
__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ tab, uint64_t n, uint64_t total, float* __restrict__ out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (uint64_t i = tid; i < total; i += stride * 8) {
    const float* p[8]; float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint64_t ii = i + u * stride;
      p[u] = tab + rnd_index(ii < total ? ii : tid, n);
    }
    load8<F>(p, v);
#pragma unroll
    for (int u = 0; u < 8; u++) acc += v[u];
  }
  if (acc == 123.456f) out[tid] = acc;            // keep the loads alive
}

// wave-uniform index: one scalar load per access and wavefront
__global__ void __launch_bounds__(256) k_sgather(const float* __restrict__ tab, uint64_t n, uint64_t total, float* __restrict__ out) {
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  float acc = 0.f;
  for (uint64_t i = wave; i < total; i += nwaves * 4) {
    const float* p[4]; float v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint64_t ii = i + u * nwaves;
      const uint64_t idx = rnd_index(ii < total ? ii : wave, n);
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)idx), hi = __builtin_amdgcn_readfirstlane((uint32_t)(idx >> 32));
      p[u] = tab + (((uint64_t)hi << 32) | lo);
    }
    asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %5, 0x0\n\ts_load_dword %2, %6, 0x0\n\ts_load_dword %3, %7, 0x0\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(v[0]), "=&s"(v[1]), "=&s"(v[2]), "=&s"(v[3]) : "s"(p[0]), "s"(p[1]), "s"(p[2]), "s"(p[3]) : "memory");
#pragma unroll
    for (int u = 0; u < 4; u++) acc += v[u];
  }
  if (acc == 123.456f) out[wave] = acc;
}

enum { S_PLAIN = 0, S_NT = 1, S_ATOMIC = 2, S_SC0SC1 = 3 };
#define LOAD4(SUFFIX)                                                                                        \
  asm volatile("global_load_dword %0, %4, off" SUFFIX "\n\tglobal_load_dword %1, %5, off" SUFFIX "\n\t"         \
               "global_load_dword %2, %6, off" SUFFIX "\n\tglobal_load_dword %3, %7, off" SUFFIX "\n\ts_waitcnt vmcnt(0)" \
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory")
#define STORE4(OP, SUFFIX)                                                                                   \
  asm volatile(OP " %0, %4, off" SUFFIX "\n\t" OP " %1, %5, off" SUFFIX "\n\t" OP " %2, %6, off" SUFFIX "\n\t"     \
               OP " %3, %7, off" SUFFIX "\n\ts_waitcnt vmcnt(0)"                                              \
               :: "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "memory")
template <int F>
__global__ void __launch_bounds__(256) k_rmw(float* __restrict__ tab, uint64_t n, uint64_t total) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = tid; i < total; i += stride * 4) {
    float* p[4]; float v[4] = {0.f, 0.f, 0.f, 0.f}, w[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint64_t ii = i + u * stride;
      p[u] = tab + rnd_index(ii < total ? ii : tid, n);
    }
    if constexpr (F == S_SC0SC1) LOAD4(" sc0 sc1");
    else if constexpr (F != S_ATOMIC) LOAD4("");
#pragma unroll
    for (int u = 0; u < 4; u++) w[u] = v[u] + 1.0f;
    if constexpr (F == S_PLAIN)  STORE4("global_store_dword", "");
    if constexpr (F == S_NT)     STORE4("global_store_dword", " nt");
    if constexpr (F == S_SC0SC1) STORE4("global_store_dword", " sc0 sc1");
    if constexpr (F == S_ATOMIC) STORE4("global_atomic_add_f32", "");
  }
}

// reference point: a 256-byte row gather (the V rows): 64 lanes x 4 bytes, one row per wavefront and access
__global__ void __launch_bounds__(256) k_rowgather(const float* __restrict__ tab, uint64_t n_rows, uint64_t total, float* __restrict__ out) {
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t lane = threadIdx.x & 63u;
  float acc = 0.f;
  for (uint64_t i = wave; i < total; i += nwaves * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint64_t ii = i + u * nwaves;
      v[u] = __builtin_nontemporal_load(tab + rnd_index(ii < total ? ii : wave, n_rows) * 64 + lane);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) acc += v[u];
  }
  if (acc == 123.456f) out[wave * 64 + lane] = acc;
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  void start() { CK(hipEventRecord(a, 0)); }
  float stop() { CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

template <class K, class... A> void run(const char* name, const char* alloc, uint64_t accesses, int bytes_alg, K k, dim3 g, A... args) {
  Timer t;
  printf("%-28s %-12s ...", name, alloc); fflush(stdout);    // (named before it runs: a faulting flavour is identified)
  hipLaunchKernelGGL(k, g, dim3(256), 0, 0, args...);     // warm-up
  CK(hipDeviceSynchronize());
  printf("\r");
  t.start();
  hipLaunchKernelGGL(k, g, dim3(256), 0, 0, args...);
  const float ms = t.stop();
  CK(hipGetLastError());
  printf("%-28s %-12s %8.3f ms  %8.2f G access/s  %7.1f GB/s algorithmic (%d B each)\n", name, alloc, ms, accesses / ms * 1e-6,
         accesses * (double)bytes_alg / ms * 1e-6, bytes_alg);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 800000000ull;       // 3.2 GB: far beyond the 256 MB Infinity Cache
  const uint64_t total = argc > 2 ? strtoull(argv[2], 0, 10) : (1ull << 28);
  float* out; CK(hipMalloc(&out, (size_t)(1 << 24) * 4));
  const dim3 g(256 * 8);
  struct { const char* name; unsigned flags; bool ext; } kinds[] = {
    {"hipMalloc", 0, false}, {"uncached", hipDeviceMallocUncached, true}, {"fine-grained", hipDeviceMallocFinegrained, true}};
  const int only = argc > 3 ? atoi(argv[3]) : -1;               // 0 hipMalloc, 1 uncached, 2 fine-grained (PMC passes: one kind per run)
  int kind_no = -1;
  for (auto& kd : kinds) {
    kind_no++;
    if (only >= 0 && only != kind_no) continue;
    float* tab = nullptr;
    hipError_t e = kd.ext ? hipExtMallocWithFlags((void**)&tab, n * 4, kd.flags) : hipMalloc((void**)&tab, n * 4);
    if (e != hipSuccess) { printf("%s: allocation failed (%s)\n", kd.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    CK(hipMemset(tab, 0, n * 4));
    CK(hipDeviceSynchronize());
    run("load plain", kd.name, total, 4, k_gather<F_PLAIN>, g, (const float*)tab, n, total, out);
    run("load nt", kd.name, total, 4, k_gather<F_NT>, g, (const float*)tab, n, total, out);
    run("load sc0", kd.name, total, 4, k_gather<F_SC0>, g, (const float*)tab, n, total, out);
    run("load sc1", kd.name, total, 4, k_gather<F_SC1>, g, (const float*)tab, n, total, out);
    run("load sc0 sc1", kd.name, total, 4, k_gather<F_SC0SC1>, g, (const float*)tab, n, total, out);
    run("load sc0 sc1 nt", kd.name, total, 4, k_gather<F_SC0SC1NT>, g, (const float*)tab, n, total, out);
    run("scalar s_load_dword", kd.name, total / 16, 4, k_sgather, g, (const float*)tab, n, total / 16, out);
    run("rmw plain store", kd.name, total, 8, k_rmw<S_PLAIN>, g, tab, n, total);
    run("rmw nt store", kd.name, total, 8, k_rmw<S_NT>, g, tab, n, total);
    run("rmw sc0 sc1 load+store", kd.name, total, 8, k_rmw<S_SC0SC1>, g, tab, n, total);
    run("atomic add (no return)", kd.name, total, 8, k_rmw<S_ATOMIC>, g, tab, n, total);
    if (!kd.ext) run("256-B row gather (nt)", kd.name, total / 16, 256, k_rowgather, g, (const float*)tab, n / 64, total / 16, out);
    CK(hipFree(tab));
  }
  return 0;
}

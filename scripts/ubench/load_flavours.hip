// dependent-load latency by load flavour and working-set size on gfx950 (one wavefront, lane 0 chases a random cycle):
// which global_load forms are served by the per-CU L1, which by the XCD's L2, which go to the fabric.  Used to choose the loads of the
// XCD-resident epoch (libfm_amd/csrc/fmx_xcd_kernels.h).   hipcc --offload-arch=gfx950 -O3 -o load_flavours load_flavours.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int F> __device__ __forceinline__ unsigned ld(const unsigned* p) {
  unsigned v;
  if constexpr (F == 0) asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if constexpr (F == 1) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if constexpr (F == 2) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if constexpr (F == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if constexpr (F == 4) asm volatile("global_load_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if constexpr (F == 5) asm volatile("global_load_dword %0, %1, off sc0 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if constexpr (F == 6) asm volatile("global_load_dword %0, %1, off sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int F> __global__ void chase(const unsigned* buf, unsigned steps, unsigned long long* out, unsigned warm) {
  if (threadIdx.x != 0) return;
  unsigned i = 0;
  for (unsigned s = 0; s < warm; s++) i = ld<F>(buf + (size_t)i * 16);        // one element per 64-byte line
  const unsigned long long t0 = wall_clock64();
  for (unsigned s = 0; s < steps; s++) i = ld<F>(buf + (size_t)i * 16);
  const unsigned long long t1 = wall_clock64();
  out[0] = t1 - t0; out[1] = i;
}
int main() {
  const char* names[7] = {"plain", "sc0", "sc1", "sc0 sc1", "nt", "sc0 nt", "sc1 nt"};
  const size_t lines_list[] = {64, 256, 4096, 32768, 1u << 22};   // 4 KB, 16 KB, 256 KB, 2 MB, 256 MB (one dword used per 64-byte line)
  unsigned long long* out; CHK(hipMalloc(&out, 16));
  printf("%-10s", "lines");
  for (auto n : names) printf(" %9s", n);
  printf("   (ns per dependent load, second pass over the cycle)\n");
  for (size_t lines : lines_list) {
    std::vector<unsigned> perm(lines); std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(7); std::shuffle(perm.begin() + 1, perm.end(), rng);
    std::vector<unsigned> host(lines * 16, 0u);
    for (size_t k = 0; k < lines; k++) host[(size_t)perm[k] * 16] = perm[(k + 1) % lines];
    unsigned* buf; CHK(hipMalloc(&buf, host.size() * 4)); CHK(hipMemcpy(buf, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    printf("%-10zu", lines);
    const unsigned steps = (unsigned)std::min<size_t>(lines, 4096), warm = (unsigned)std::min<size_t>(lines, 1u << 16);
    for (int f = 0; f < 7; f++) {
      unsigned long long h[2];
#define RUN(F) case F: hipLaunchKernelGGL(chase<F>, dim3(1), dim3(64), 0, 0, buf, steps, out, warm); break;
      switch (f) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) }
      CHK(hipDeviceSynchronize()); CHK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
      printf(" %9.0f", (double)h[0] * 10.0 / steps);
    }
    printf("\n");
    CHK(hipFree(buf));
  }
  return 0;
}

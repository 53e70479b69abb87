// micro-benchmark (round 3): the bias recurrence of a batch on its own -- the serial chain that bounds short-row shapes (BASELINE
// configs[1]) and a P = 8 rank.  The library's kernels (k_scan: one wavefront, lane-strided, any micro-chunk; k_scan1: one wavefront on the
// chain, four contiguous examples per lane, operands prepared off the chain, a four-wavefront tile pipeline) over the same
// rest / target arrays: microseconds per 262 144 examples, nanoseconds per micro-chunk, and the bias each of them ends with.
//   hipcc --offload-arch=gfx950 -O3 -I../../libfm_amd/csrc scan_chain.hip -o scan_chain ; ./scan_chain [rows=262144] [chunk=256]
#include "fmx_kernels.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace fmx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv) {
  const uint32_t rows = argc > 1 ? (uint32_t)atoi(argv[1]) : 262144u;
  const uint32_t chunk = argc > 2 ? (uint32_t)atoi(argv[2]) : 256u;
  std::vector<float> r(rows), y(rows);
  uint64_t s = 12345;
  for (uint32_t i = 0; i < rows; i++) {
    s = mix64(s + i);
    r[i] = (float)((double)(s >> 11) / 9007199254740992.0 * 4.0 - 2.0);
    y[i] = ((s >> 3) & 1) ? 1.f : -1.f;
  }
  float *d_r, *d_y, *d_m; double *d_w0;
  CK(hipMalloc(&d_r, rows * 4)); CK(hipMalloc(&d_y, rows * 4)); CK(hipMalloc(&d_m, rows * 4)); CK(hipMalloc(&d_w0, 16));
  CK(hipMemcpy(d_r, r.data(), rows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_y, y.data(), rows * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int task = 1; task >= 0; task--) {
    Hyper h = {};
    h.lr = 0.01f; h.reg0 = 0.001f; h.task = task; h.k0 = 1; h.k1 = 1; h.min_target = -1.5f; h.max_target = 1.5f;
    // fp64 host restatement of the chain (fm_learn_sgd_element.h:57-65 per example, summed per micro-chunk)
    double w0 = 0.1;
    for (uint32_t c0 = 0; c0 < rows; c0 += chunk) {
      const uint32_t nc = std::min(chunk, rows - c0);
      const float w0s = (float)w0;
      double g = 0;
      for (uint32_t i = 0; i < nc; i++) {
        const double p = (double)w0s + r[c0 + i], yy = y[c0 + i];
        g += task ? -yy * (1.0 - 1.0 / (1.0 + std::exp(-yy * p))) : -(yy - std::fmin(1.5, std::fmax(-1.5, p)));
      }
      w0 -= (double)h.lr * (g + (double)nc * (double)h.reg0 * (double)w0s);
    }
    printf("task %s, %u rows, micro-chunk %u: host fp64 bias %.9f\n", task ? "classification" : "regression", rows, chunk, w0);
    for (int variant = 0; variant < 6; variant++) {
      const bool wm = variant >= 3;
      const int kind = variant % 3;
      if (kind == 1) continue;                               // (round 2's k_scan4 -- four wavefronts sharing a piece, LDS exchange per piece -- is gone: 329 vs 174 ns, profiles/r03_scan_chain.txt)
      if (kind == 2 && chunk % 256) continue;
      auto launch = [&]() {
        const double init = 0.1;
        CK(hipMemcpyAsync(d_w0, &init, 8, hipMemcpyHostToDevice, 0));
#define RUN(K, GRID, LDS) hipLaunchKernelGGL(K, dim3(1), dim3(GRID), LDS, 0, d_r, d_y, rows, chunk, h, d_w0, d_w0 + 1, wm ? d_m : nullptr)
        if (kind == 0) { if (task) { if (wm) RUN((k_scan<true, 1>), 64, 0); else RUN((k_scan<false, 1>), 64, 0); } else { if (wm) RUN((k_scan<true, 0>), 64, 0); else RUN((k_scan<false, 0>), 64, 0); } }
        if (kind == 2 && chunk == 256) { if (task) { if (wm) RUN((k_scan1<true, 1, true>), 256, SCAN4_LDS_BYTES); else RUN((k_scan1<false, 1, true>), 256, SCAN4_LDS_BYTES); }
                         else      { if (wm) RUN((k_scan1<true, 0, true>), 256, SCAN4_LDS_BYTES); else RUN((k_scan1<false, 0, true>), 256, SCAN4_LDS_BYTES); } }
        if (kind == 2 && chunk != 256) { if (task) { if (wm) RUN((k_scan1<true, 1, false>), 256, SCAN4_LDS_BYTES); else RUN((k_scan1<false, 1, false>), 256, SCAN4_LDS_BYTES); }
                         else      { if (wm) RUN((k_scan1<true, 0, false>), 256, SCAN4_LDS_BYTES); else RUN((k_scan1<false, 0, false>), 256, SCAN4_LDS_BYTES); } }
#undef RUN
      };
#define RAISE(K) CK(hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCAN4_LDS_BYTES))
      RAISE((k_scan1<true, 1, true>)); RAISE((k_scan1<false, 1, true>)); RAISE((k_scan1<true, 0, true>)); RAISE((k_scan1<false, 0, true>));
      RAISE((k_scan1<true, 1, false>)); RAISE((k_scan1<false, 1, false>)); RAISE((k_scan1<true, 0, false>)); RAISE((k_scan1<false, 0, false>));
#undef RAISE
      launch(); CK(hipDeviceSynchronize());
      double sum = 0;
      for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms;
      }
      double out = 0; CK(hipMemcpy(&out, d_w0 + 1, 8, hipMemcpyDeviceToHost));
      double msum = 0;
      if (wm) { std::vector<float> m(rows); CK(hipMemcpy(m.data(), d_m, rows * 4, hipMemcpyDeviceToHost)); for (float v : m) msum += v; }
      const char* names[3] = {"k_scan  (1 wavefront, strided)", "k_scan4 (4 wavefronts, LDS exchange)", "k_scan1 (1 wavefront, contiguous)"};
      printf("  %-38s %s: %8.1f us per batch, %6.1f ns per micro-chunk, bias %.9f (dev from fp64 %.2e)%s\n", names[kind], wm ? "+ multipliers" : "bias only   ",
             sum / 5 * 1e3, sum / 5 * 1e6 / ((rows + chunk - 1) / chunk), out, std::fabs(out - w0), wm ? "" : "");
      (void)msum;
    }
  }
  return 0;
}

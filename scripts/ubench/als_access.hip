// micro-benchmark: access patterns for one (factor, level) step of the ALS sweep on one-hot data.
//   A: column-driven (current k_als_draw): random 16-B gather + random 16-B read-modify-write of EQ
//   B: row-driven sums: sequential EQ + 4-B gather of theta + 2 fp64 atomics per entry into per-feature accumulators
//   C: row-driven update: sequential EQ read-modify-write + 8-B gather of delta
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics als_access.hip -o als_access
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct EQ { double e, q; };
struct TE { uint32_t e; float x; };
struct RE { uint32_t j; float x; };

__global__ void kA_sum(const TE* __restrict__ te, const uint32_t* __restrict__ rel, uint32_t ncol, const float* __restrict__ vf, EQ* eq, double* __restrict__ out) {
  const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, lane = threadIdx.x & 7;
  if (g >= ncol) return;
  const uint32_t a = rel[g], b = rel[g + 1];
  const double th = vf[g];
  double he = 0, hh = 0;
  for (uint32_t i = a + lane; i < b; i += 8) { TE t = te[i]; EQ c = eq[t.e]; double x = t.x, h = x * (c.q - x * th); he += h * c.e; hh += h * h; }
  for (int o = 4; o; o >>= 1) { he += __shfl_xor(he, o); hh += __shfl_xor(hh, o); }
  const double d = 1e-9 * he / (1.0 + hh);
  for (uint32_t i = a + lane; i < b; i += 8) { TE t = te[i]; EQ c = eq[t.e]; double x = t.x; c.e -= x * (c.q - x * th) * d; c.q -= x * d; eq[t.e] = c; }
  if (lane == 0) out[g] = d;
}
__global__ void kB_sum(const RE* __restrict__ re, uint32_t nrow, const float* __restrict__ vf, const EQ* __restrict__ eq, double* acc) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nrow) return;
  RE r = re[c]; EQ q = eq[c];
  const double th = vf[r.j], x = r.x, h = x * (q.q - x * th);
  atomicAdd(&acc[2 * (size_t)r.j], h * q.e);
  atomicAdd(&acc[2 * (size_t)r.j + 1], h * h);
}
__global__ void kB_mid(uint32_t ncol, double* acc, double* delta) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncol) return;
  delta[j] = 1e-9 * acc[2 * j] / (1.0 + acc[2 * j + 1]); acc[2 * j] = 0; acc[2 * j + 1] = 0;
}
__global__ void kC_upd(const RE* __restrict__ re, uint32_t nrow, const float* __restrict__ vf, const double* __restrict__ delta, EQ* eq) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nrow) return;
  RE r = re[c]; EQ q = eq[c];
  const double d = delta[r.j], th = vf[r.j], x = r.x;
  q.e -= x * (q.q - x * th) * d; q.q -= x * d; eq[c] = q;
}
// fused: update of the previous level + sums of the next one in one pass over the rows
__global__ void kCB(const RE* __restrict__ re0, const RE* __restrict__ re1, uint32_t nrow, const float* __restrict__ vf0, const float* __restrict__ vf1,
                    const double* __restrict__ delta, EQ* eq, double* acc) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nrow) return;
  RE r = re0[c]; EQ q = eq[c];
  { const double d = delta[r.j], th = vf0[r.j], x = r.x; q.e -= x * (q.q - x * th) * d; q.q -= x * d; }
  eq[c] = q;
  RE s = re1[c];
  const double th = vf1[s.j], x = s.x, h = x * (q.q - x * th);
  atomicAdd(&acc[2 * (size_t)s.j], h * q.e);
  atomicAdd(&acc[2 * (size_t)s.j + 1], h * h);
}

int main() {
  const uint32_t nrow = 1u << 22, ncol = 625000;
  std::mt19937 rng(1);
  std::vector<uint32_t> colof(nrow), rel(ncol + 1, 0);
  for (auto& c : colof) c = rng() % ncol;
  for (auto c : colof) rel[c + 1]++;
  for (uint32_t j = 0; j < ncol; j++) rel[j + 1] += rel[j];
  std::vector<TE> te(nrow); std::vector<RE> re(nrow);
  { std::vector<uint32_t> pos(rel.begin(), rel.end() - 1);
    for (uint32_t c = 0; c < nrow; c++) { te[pos[colof[c]]++] = TE{c, 1.0f}; re[c] = RE{colof[c], 1.0f}; } }
  TE* d_te; RE* d_re; uint32_t* d_rel; float* d_vf; EQ* d_eq; double *d_acc, *d_delta, *d_out;
  CK(hipMalloc(&d_te, nrow * sizeof(TE))); CK(hipMalloc(&d_re, nrow * sizeof(RE))); CK(hipMalloc(&d_rel, (ncol + 1) * 4));
  CK(hipMalloc(&d_vf, ncol * 4)); CK(hipMalloc(&d_eq, nrow * sizeof(EQ))); CK(hipMalloc(&d_acc, ncol * 16)); CK(hipMalloc(&d_delta, ncol * 8)); CK(hipMalloc(&d_out, ncol * 8));
  CK(hipMemcpy(d_te, te.data(), nrow * sizeof(TE), hipMemcpyHostToDevice)); CK(hipMemcpy(d_re, re.data(), nrow * sizeof(RE), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rel, rel.data(), (ncol + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_vf, 0, ncol * 4)); CK(hipMemset(d_eq, 0, nrow * sizeof(EQ))); CK(hipMemset(d_acc, 0, ncol * 16)); CK(hipMemset(d_delta, 0, ncol * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 50; float ms;
  auto report = [&](const char* name) { hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("%-34s %8.1f us per step\n", name, ms * 1e3 / reps); };
  for (int w = 0; w < 2; w++) {
    hipEventRecord(e0); for (int r = 0; r < reps; r++) kA_sum<<<(ncol * 8 + 255) / 256, 256>>>(d_te, d_rel, ncol, d_vf, d_eq, d_out); report("A column-driven (current)");
    hipEventRecord(e0); for (int r = 0; r < reps; r++) kB_sum<<<nrow / 256, 256>>>(d_re, nrow, d_vf, d_eq, d_acc); report("B row sums + fp64 atomics");
    hipEventRecord(e0); for (int r = 0; r < reps; r++) kB_mid<<<(ncol + 255) / 256, 256>>>(ncol, d_acc, d_delta); report("B' per-feature solve");
    hipEventRecord(e0); for (int r = 0; r < reps; r++) kC_upd<<<nrow / 256, 256>>>(d_re, nrow, d_vf, d_delta, d_eq); report("C row update");
    hipEventRecord(e0); for (int r = 0; r < reps; r++) { kCB<<<nrow / 256, 256>>>(d_re, d_re, nrow, d_vf, d_vf, d_delta, d_eq, d_acc); kB_mid<<<(ncol + 255) / 256, 256>>>(ncol, d_acc, d_delta); } report("C+B fused row pass + solve");
  }
  return 0;
}

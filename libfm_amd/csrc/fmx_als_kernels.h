// fmx_als_kernels.h -- gfx950 kernels of the ALS / MCMC learner (coordinate-wise Gauss-Seidel over X^T).
//
// Reference (restated, never copied):
//   fm_learn_mcmc::predict_data_and_write_to_eterms  /root/reference/src/libfm/src/fm_learn_mcmc.h:148-378
//   add_main_q :406-428, draw_all :430-641, draw_w0 :643-683, draw_w :685-732, draw_v :792-847
//   fm_learn_mcmc_simultaneous::_learn               /root/reference/src/libfm/src/fm_learn_mcmc_simultaneous.h:56-270
//
// State on the device: e[c] = y-hat(c) - target(c) and q[f][c] = sum_j v_fj x_cj in fp64 (the reference keeps
// both in fp64: e_q_term, fm_learn_mcmc.h:46-49); parameters stay in the fp32 table shared with the SGD path,
// every new value is rounded to fp32 BEFORE its delta is pushed into e/q, so the caches always describe the
// stored parameters.
//
// Parallelism: two coordinates conflict iff their columns share a row.  The host assigns every feature a
// LEVEL = 1 + max level of the earlier (smaller id) features it shares a row with; features of one level are
// mutually independent and all their predecessors are in lower levels, so "level by level, all features of a
// level in parallel" performs EXACTLY the reference's sequential sweep (one-hot field data: level = field).
#pragma once

#include "fmx_kernels.h"

namespace fmx {

// e_q_term (fm_learn_mcmc.h:46-49): the residual and the CURRENT factor's q of one training row side by side,
// so that a coordinate update touches ONE 16-byte slot per entry (one cache line instead of two)
struct __attribute__((aligned(16))) EQ { double e, q; };

__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

// reference's erf polynomial / cdf_gaussian (random.h:45-67), needed bit-compatibly for the probit target
__device__ __forceinline__ double ref_erf(double x) {
  const double t = (x >= 0) ? 1.0 / (1.0 + 0.3275911 * x) : 1.0 / (1.0 - 0.3275911 * x);
  const double r = 1.0 - (t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429))))) * exp(-x * x);
  return (x >= 0) ? r : -r;
}
__device__ __forceinline__ double ref_cdf_gaussian(double x) { return 0.5 + 0.5 * ref_erf(0.707106781 * x); }

// ----------------------------------------------------------------------------------------------
// k_als_eterms: one wavefront per row.  e[c] = y-hat(c) (fp64 accumulation, fm_model.h:105-127 arithmetic; the
// reference's 2k+1 passes over X^T, fm_learn_mcmc.h:176-368, compute the same number) and q[f][c] for the
// coming sweep (add_main_q :406-428 evaluated for every factor at once: v_f does not change before its own
// sweep, so the value is the one the reference would compute later).
// ----------------------------------------------------------------------------------------------
// Rows per wavefront pass: the q_f of R consecutive rows leave through an LDS tile, so a store covers 8 R contiguous bytes
// of every factor's row of q[KP][n_rows] instead of 8 (one lane per factor, one row at a time).
template <int KP> struct EtermsRows { static constexpr int R = (KP > 128) ? 4 : 8; };
template <int KP>
__global__ void __launch_bounds__(256)
k_als_eterms(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, uint32_t n_rows, const Tab tb,
             int k0, int k1, const double* __restrict__ w0_ptr, EQ* __restrict__ eq, double* __restrict__ q /* [KP][n_rows] or null */,
             double* __restrict__ e_part /* feature shard: lin - 0.5 * sum of squares goes here (k_als_set_e finishes y-hat) */) {
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR, EPI = Map<KP>::EPI, R = EtermsRows<KP>::R, U = 8;
  __shared__ double tile[4][R][KP + 1];                            // [wavefront][row][factor], padded against bank conflicts
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t g = lane / LPR, f = lane % LPR;                  // EPI entries per load instruction, sub-group g takes one
  const double w0 = k0 ? *w0_ptr : 0.0;
  // every wavefront of a workgroup makes the same number of trips (the tile hand-over uses the workgroup barrier)
  for (uint64_t cb = (uint64_t)blockIdx.x * 4 * R; cb < n_rows; cb += (uint64_t)gridDim.x * 4 * R) {
    const uint64_t c0 = cb + (uint64_t)wv * R;
    double e_keep = 0.0;                                           // lane r keeps the scalar of row c0 + r
    for (int r = 0; r < R; r++) {
      const uint64_t c = c0 + r;
      if (c >= n_rows) break;                                      // (wave-uniform)
      const uint64_t a = row_ptr[c];
      const uint32_t size = (uint32_t)(row_ptr[c + 1] - a);
      double sum[VEC]; double part = 0.0;                          // part: this lane's share of lin - 0.5 * sum of squares
#pragma unroll
      for (int v = 0; v < VEC; v++) sum[v] = 0.0;
      for (uint32_t base = 0; base < size; base += 64) {
        const uint32_t cnt = min(64u, size - base);
        Entry en; en.id = 0; en.value = 0.f;
        if (lane < cnt) {
          en = load_stream8(ent + a + base + lane);                // one entry per lane, broadcast below
          if (k1) part += (double)load_w(tb.w + (size_t)en.id * tb.ws) * (double)en.value;
        }
        for (uint32_t i = 0; i < cnt; i += EPI * U) {
          float vr[U][VEC]; float xs[U];
#pragma unroll
          for (int u = 0; u < U; u++) {                            // U rows of V in flight per lane
            const uint32_t idx = i + u * EPI + g;
            const uint32_t id = bcast_u32<EPI>(en.id, idx & 63u);
            xs[u] = bcast_f32<EPI>(en.value, idx & 63u);
            if (idx < cnt) row_ld<VEC, 4>(tb, (size_t)id, f * VEC, vr[u]);
            else {
              xs[u] = 0.f;
#pragma unroll
              for (int v = 0; v < VEC; v++) vr[u][v] = 0.f;
            }
          }
#pragma unroll
          for (int u = 0; u < U; u++)
#pragma unroll
            for (int v = 0; v < VEC; v++) {
              const double d = (double)vr[u][v] * (double)xs[u];
              sum[v] += d;
              part -= 0.5 * d * d;
            }
        }
      }
      if constexpr (EPI > 1) {                                     // the sub-groups' partial factor sums -> complete, in every lane
#pragma unroll
        for (int v = 0; v < VEC; v++)
#pragma unroll
          for (int o = LPR; o < 64; o <<= 1) sum[v] += __shfl_xor(sum[v], o);
      }
      if (g == 0) {
        if (!e_part) {                                             // (a shard's factor sums are partial: their squares wait for the all-reduce)
#pragma unroll
          for (int v = 0; v < VEC; v++) part += 0.5 * sum[v] * sum[v];
        }
#pragma unroll
        for (int v = 0; v < VEC; v++) tile[wv][r][f * VEC + v] = sum[v];
      }
      part = wave_sum_f64(part);
      if (lane == (uint32_t)r) e_keep = part;
    }
    if (lane < R && c0 + lane < n_rows) { if (e_part) e_part[c0 + lane] = e_keep; else eq[c0 + lane].e = w0 + e_keep; }
    __syncthreads();
    if (q && c0 < n_rows) {
      const uint32_t nr = (uint32_t)min((uint64_t)R, (uint64_t)n_rows - c0);
      for (uint32_t i = lane; i < (uint32_t)R * KP; i += 64) {     // R consecutive rows of one factor per 8 R bytes
        const uint32_t r = i % R, ff = i / R;
        if (r < nr) q[(size_t)ff * n_rows + c0 + r] = tile[wv][r][ff];
      }
    }
    __syncthreads();
  }
}

// e[c] -= target[c] (initialisation, _learn :70-86)
static __global__ void k_als_sub_target(EQ* __restrict__ eq, const float* __restrict__ target, uint32_t n) {
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) eq[c].e -= (double)target[c];
}
// add_main_q (:406-428) for factor f was evaluated by k_als_eterms; move it next to e for the coming sweep
static __global__ void k_als_load_q(EQ* __restrict__ eq, const double* __restrict__ qf, uint32_t n) {
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) eq[c].q = qf[c];
}

// feature shards: a shard's re-prediction leaves c_part = lin - 0.5 * sum_f sum_i (v x)^2 and its partial q_f; after the
// all-reduce of both   e = w0 + c + 0.5 * sum_f q_f^2   (fm_model.h:116-125: the square is of the COMPLETE factor sum)
static __global__ void k_als_set_e(EQ* __restrict__ eq, const double* __restrict__ c_sum, const double* __restrict__ q, int k, uint32_t n,
                                   int k0, const double* __restrict__ w0_ptr) {
  const double w0 = k0 ? *w0_ptr : 0.0;
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
    double p = w0 + c_sum[c];
    for (int f = 0; f < k; f++) { const double s = q[(size_t)f * n + c]; p += 0.5 * s * s; }
    eq[c].e = p;
  }
}
// feature shards: the draws of a level RECORD what they would do to {e, q} (delta[row], rows disjoint inside a level);
// after the all-reduce every shard applies the same sum, so the replicas of the cache stay identical bit for bit
static __global__ void k_als_apply_delta(EQ* __restrict__ eq, EQ* __restrict__ delta, uint32_t n) {
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
    const EQ d = delta[c];
    if (d.e != 0.0 || d.q != 0.0) {
      EQ v = eq[c]; v.e += d.e; v.q += d.q; eq[c] = v;
      EQ z; z.e = 0.0; z.q = 0.0; delta[c] = z;
    }
  }
}

// counter-based uniforms / normals for the Gibbs variant (statistical, not bitwise, parity with libc rand())
__device__ __forceinline__ double unif_hash(uint64_t seed, uint64_t stream, uint64_t idx, uint32_t attempt) {
  const uint64_t hh = mix64(seed ^ (stream * 0x9E3779B97F4A7C15ULL) ^ (idx * 0xD6E8FEB86659FD93ULL + attempt * 0xA24BAED4963EE407ULL + 0x9FB21C651E98DF25ULL));
  return ((double)(hh >> 11) + 0.5) * (1.0 / 9007199254740992.0);     // (0,1)
}
__device__ __forceinline__ double gauss_hash2(uint64_t seed, uint64_t stream, uint64_t idx, uint32_t attempt) {
  const double u1 = unif_hash(seed, stream, idx, 2 * attempt), u2 = unif_hash(seed, stream, idx, 2 * attempt + 1);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
// ran_left_tgaussian(left): standard normal conditioned on z >= left (random.h:70-101): naive rejection for
// left <= 0, Robert's translated-exponential rejection otherwise.  Bounded loops (64 attempts) for safety.
__device__ __forceinline__ double left_tgauss(double left, uint64_t seed, uint64_t stream, uint64_t idx) {
  if (left <= 0.0) {
    double r = 0.0;
    for (uint32_t a = 0; a < 64; a++) { r = gauss_hash2(seed, stream, idx, a); if (r >= left) return r; }
    return fmax(r, left);
  }
  const double alpha_star = 0.5 * (left + sqrt(left * left + 4.0));
  double zz = left;
  for (uint32_t a = 0; a < 64; a++) {
    zz = -log(1.0 - unif_hash(seed, stream, idx, 2 * a)) / alpha_star + left;
    double d = zz - alpha_star;
    d = exp(-(d * d) / 2);
    if (unif_hash(seed, stream, idx, 2 * a + 1) < d) return zz;
  }
  return zz;
}

// after a sweep, e holds y-hat: accumulate the train metric and turn e back into the residual
//   regression    (_learn :139-150): rmse over clamped predictions, e -= y
//   classification(_learn :163-196): accuracy of cdf_gaussian(e) vs sign, e -= E[truncated normal] (do_sample = 0)
// acc[0] = sum err^2 / #correct
static __global__ void __launch_bounds__(256)
k_als_targets(EQ* __restrict__ eq, const float* __restrict__ target, uint32_t n, int task,
              double min_target, double max_target, double* __restrict__ acc,
              int do_sample, uint64_t seed, uint64_t stream) {
  double s = 0.0;
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
    const double yh = eq[c].e;
    const double y = (double)target[c];
    if (task == 0) {
      double p = fmin(max_target, yh);
      p = fmax(min_target, p);
      const double err = p - y;
      s += err * err;
      eq[c].e = yh - y;
    } else {
      const double p = ref_cdf_gaussian(yh);
      if (((p >= 0.5) && (y > 0.0)) || ((p < 0.5) && (y < 0.0))) s += 1.0;
      double st;
      if (do_sample) {                                                        // :172-175, :184-186
        // ran_left_tgaussian(0, mu, 1) = mu + ltg(-mu);  ran_right_tgaussian(0, mu, 1) = mu - ltg(mu)   (random.h:99-112)
        st = (y >= 0.0) ? yh + left_tgauss(-yh, seed, stream, c) : yh - left_tgauss(yh, seed, stream, c);
      } else {
        const double phi_minus_mu = exp(-yh * yh / 2.0) / sqrt(3.141 * 2);     // sic: 3.141 (:179,:189)
        const double Phi_minus_mu = ref_cdf_gaussian(-yh);
        st = (y >= 0.0) ? yh + phi_minus_mu / (1 - Phi_minus_mu) : yh - phi_minus_mu / Phi_minus_mu;
      }
      eq[c].e = yh - st;
    }
  }
  s = wave_sum_f64(s);
  if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(acc, s);
}

// sum_c e[c] (draw_w0's numerator, :650-652: sum (e - w0) = sum e - N w0) and sum e^2 (draw_alpha :918-920)
static __global__ void __launch_bounds__(256)
k_als_sum_e(const EQ* __restrict__ eq, uint32_t n, double* __restrict__ acc) {
  double s = 0.0, s2 = 0.0;
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) { const double v = eq[c].e; s += v; s2 += v * v; }
  s = wave_sum_f64(s); s2 = wave_sum_f64(s2);
  if ((threadIdx.x & 63) == 0) { unsafeAtomicAdd(acc, s); unsafeAtomicAdd(acc + 1, s2); }
}
static __global__ void k_als_add_const(EQ* __restrict__ eq, uint32_t n, double d) {
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) eq[c].e += d;
}

// per attribute group g: sum_{j in g} theta_j and sum_{j in g} theta_j^2 of EVERY coordinate family in one pass over the
// parameter rows (the statistics of draw_w_mu/_lambda and draw_v_mu/_lambda, fm_learn_mcmc.h:941-1097).
// out: [1 + KP][G][2] doubles (row 0 = w, row 1+f = v_f), accumulated with atomics (zero it first).
// One row of V per LPR lanes (coalesced 4*KP bytes); a lane owns VEC consecutive factors.
//   GROUPED = false (G == 1): register accumulators, one atomic per (wave, factor) at the end.
//   GROUPED = true : LDS table [1+KP][G][2] per workgroup when it fits (lds_ok), else atomics straight to `out`.
template <int KP, bool GROUPED>
__global__ void __launch_bounds__(256)
k_group_moments(const Tab tb, uint64_t n_local, int k1, const uint32_t* __restrict__ grp, uint32_t G, int lds_ok,
                double* __restrict__ out) {
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR, EPI = Map<KP>::EPI;
  extern __shared__ double mom_lds[];
  const uint32_t lane = threadIdx.x & 63u, sub = lane / LPR, fl = lane % LPR;
  const uint64_t wave0 = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  const size_t cells = (size_t)(1 + KP) * G * 2;
  if (GROUPED && lds_ok) {
    for (size_t i = threadIdx.x; i < cells; i += blockDim.x) mom_lds[i] = 0.0;
    __syncthreads();
  }
  double sv[VEC], sv2[VEC], sw = 0.0, sw2 = 0.0;
#pragma unroll
  for (int v = 0; v < VEC; v++) { sv[v] = 0.0; sv2[v] = 0.0; }
  for (uint64_t j0 = wave0 * EPI; j0 < n_local; j0 += nwaves * EPI) {
    const uint64_t j = j0 + sub;
    if (j >= n_local) continue;
    double t[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) t[v] = (fl * VEC + v < tb.rs) ? (double)tb.V[(size_t)j * tb.rs + fl * VEC + v] : 0.0;   // (rows are tb.rs floats)
    const double tw = (k1 && fl == 0) ? (double)tb.w[(size_t)j * tb.ws] : 0.0;
    if (!GROUPED) {
#pragma unroll
      for (int v = 0; v < VEC; v++) { sv[v] += t[v]; sv2[v] += t[v] * t[v]; }
      sw += tw; sw2 += tw * tw;
    } else {
      const uint32_t g = grp[j];
      double* dst = lds_ok ? mom_lds : out;
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        double* c = dst + ((size_t)(1 + fl * VEC + v) * G + g) * 2;
        unsafeAtomicAdd(c, t[v]); unsafeAtomicAdd(c + 1, t[v] * t[v]);
      }
      if (k1 && fl == 0) { unsafeAtomicAdd(dst + (size_t)g * 2, tw); unsafeAtomicAdd(dst + (size_t)g * 2 + 1, tw * tw); }
    }
  }
  if (!GROUPED) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {                      // the EPI rows a wavefront handles side by side
#pragma unroll
      for (int v = 0; v < VEC; v++) { sv[v] += __shfl_xor(sv[v], o); sv2[v] += __shfl_xor(sv2[v], o); }
      sw += __shfl_xor(sw, o); sw2 += __shfl_xor(sw2, o);
    }
    if (sub == 0) {
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        unsafeAtomicAdd(out + (size_t)(1 + fl * VEC + v) * 2, sv[v]); unsafeAtomicAdd(out + (size_t)(1 + fl * VEC + v) * 2 + 1, sv2[v]);
      }
      if (k1 && fl == 0) { unsafeAtomicAdd(out, sw); unsafeAtomicAdd(out + 1, sw2); }
    }
  } else if (lds_ok) {
    __syncthreads();
    for (size_t i = threadIdx.x; i < cells; i += blockDim.x) { const double x = mom_lds[i]; if (x != 0.0) unsafeAtomicAdd(out + i, x); }
  }
}

// counter-based N(0,1) for the sampling variant (MCMC): Box-Muller on two splitmix64 hashes of (seed, stream, index)
// N(0,1) of a coordinate draw: counter hash -> Box-Muller in fp32 with the hardware log / cos (24-bit uniforms: tails to 5.8 sigma).
// The draw is stored as an fp32 parameter anyway, and the parity of a sampled chain is statistical; in fp64
// (software log / cos / sqrt, ~200 instructions) this was 40 % of k_als_draw and most of k_als_unseen_v at configs[4]'s shape
// (3 M short columns per level, 1.2e10 prior draws per sweep).
__device__ __forceinline__ double gauss_hash(uint64_t seed, uint64_t stream, uint64_t idx) {
  const uint64_t h1 = mix64(seed ^ (stream * 0x9E3779B97F4A7C15ULL) ^ (idx * 0xD6E8FEB86659FD93ULL + 0x632BE59BD9B4E019ULL));
  const uint64_t h2 = mix64(h1 + 0x9E3779B97F4A7C15ULL);
  const float u1 = ((float)(uint32_t)(h1 >> 40) + 1.0f) * (1.0f / 16777216.0f);    // (0,1]
  const float u2 = (float)(uint32_t)(h2 >> 40) * (1.0f / 16777216.0f);             // [0,1)
  return (double)(sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2));
}

// ----------------------------------------------------------------------------------------------
// k_als_draw: one wavefront per feature of the current level.
//   IS_V = false: draw_w (:685-732)    theta = w_j,    h = x
//   IS_V = true : draw_v (:792-847)    theta = v_fj,   h = x (q_c - x theta)
// posterior: sigma^2 = 1/(lambda + alpha sum h^2), mean = -sigma^2 (alpha (sum h e - theta sum h^2) - mu lambda);
// new theta = mean (ALS) or mean + sigma N(0,1) (MCMC); NaN/Inf sigma^2 -> 0, NaN/Inf theta -> keep (:704-724).
// then e_c -= h (theta_old - theta), q_c -= x (theta_old - theta).
// seg_list: the level's segments; a segment = one feature's column inside t_ent (sorted by feature).
// ----------------------------------------------------------------------------------------------
// G lanes cooperate on one feature (G = 64: one wavefront per column; smaller G: 64/G columns per wavefront, chosen on
// the host from the mean column length -- sparse one-hot data has columns of a handful of rows)
template <int G> __device__ __forceinline__ double group_sum_f64(double x) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

// cache policy of the sweep (FMX_ALS_NT bits; measured at n=1e7, k=64: bit 2 alone 0.257 -> 0.244 s per sweep, bit 1 hurts): 1 = the X^T entries are a stream (read once per
// (factor, level) launch), 2 = the parameter scalars are one 4-byte touch per 128-B line of a table far larger than
// any cache; both evict the e/q cache (16 B per row, the only data with reuse across launches) unless hinted away.
#ifndef FMX_ALS_NT
#define FMX_ALS_NT 2
#endif
__device__ __forceinline__ TEntry als_stream8(const TEntry* p) {
#if (FMX_ALS_NT & 1)
  const uint64_t u = __builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(p));
  TEntry t; __builtin_memcpy(&t, &u, 8); return t;
#else
  return *p;
#endif
}
__device__ __forceinline__ float als_param_load(const float* p) {
#if (FMX_ALS_NT & 2)
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void als_param_store(float* p, float v) {
#if (FMX_ALS_NT & 2)
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// ---- factor-major shadow of the factors of the features that have a training column ------------------------------
// A draw touches ONE float of its feature's 4*KP-byte row: 4 useful bytes per 128-byte line, read and written, once per
// (factor, feature) -- k * 2 line touches per feature and sweep.  Vt[f][pos] (pos = position of the feature's segment in
// the level-ordered list) holds the same numbers factor-major: the draws of one (factor, level) launch then read and
// write CONSECUTIVE floats.  k_als_pack / k_als_unpack move a sweep's factors between the two layouts in one coalesced
// pass each (whole rows in, 64 consecutive positions per factor out, through an LDS tile).
template <int KP, bool PACK>
__global__ void __launch_bounds__(256)
k_als_shadow(const uint32_t* __restrict__ level_list, const uint32_t* __restrict__ seg_feat, uint32_t nseg, const Tab tb,
             float* __restrict__ vt, size_t vt_stride, uint32_t k /* rows of vt: num_factor (<= KP) */) {
  constexpr uint32_t TP = (KP > 512) ? 32u : 64u;               // positions per block (1024 factors: 32 -- the tile must fit the LDS)
  __shared__ float tile[TP][KP + 1];                            // [position in block][factor], padded against bank conflicts
  const uint32_t tid = threadIdx.x;
  for (uint32_t p0 = blockIdx.x * TP; p0 < nseg; p0 += gridDim.x * TP) {
    const uint32_t np = min(TP, nseg - p0);
    // the table side, a wavefront per 16 positions: the features' ids are looked up ONCE (two dependent 4-byte loads per position, not per
    // float), then whole rows move -- EPI rows per wave-wide access, all of a wavefront's rows in flight together (round 5: at configs[4]'s
    // shape the element-wise form ran at 1.5 TB/s packing and 0.8 TB/s unpacking, 48 ms of a 494 ms sweep)
    constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR, EPI = Map<KP>::EPI, STEPS = 16 / (EPI < 16 ? EPI : 16);
    const uint32_t lane = tid & 63u, wv = tid >> 6, sub = lane / LPR, fl = lane % LPR;
    uint32_t feat_l = 0;
    if (lane < 16u && wv * 16u + lane < np) feat_l = seg_feat[level_list[p0 + wv * 16u + lane]];
    if (PACK) {
      float vals[STEPS][VEC];
#pragma unroll
      for (int st = 0; st < STEPS; st++) {
        const uint32_t q = (uint32_t)st * EPI + sub;               // position inside the wavefront's 16
        const uint32_t feat = (uint32_t)__shfl((int)feat_l, (int)(q & 15u));
        if (q < 16u && wv * 16u + q < np && fl * VEC < tb.rs) load_vec<VEC>(tb.V + (size_t)feat * tb.rs + fl * VEC, vals[st]);
        else {
#pragma unroll
          for (int v = 0; v < VEC; v++) vals[st][v] = 0.f;         // (beyond the row: rows are tb.rs floats, fmx_kernels.h row_ld)
        }
      }
#pragma unroll
      for (int st = 0; st < STEPS; st++) {
        const uint32_t q = (uint32_t)st * EPI + sub;
        if (q < 16u && wv * 16u + q < np) {
#pragma unroll
          for (int v = 0; v < VEC; v++) tile[wv * 16u + q][fl * VEC + v] = vals[st][v];
        }
      }
      __syncthreads();
      for (uint32_t i = tid; i < k * TP; i += 256) {             // per factor TP consecutive positions
        const uint32_t f = i / TP, r = i % TP;
        if (r < np) vt[(size_t)f * vt_stride + p0 + r] = tile[r][f];
      }
    } else {
      for (uint32_t i = tid; i < k * TP; i += 256) {
        const uint32_t f = i / TP, r = i % TP;
        if (r < np) tile[r][f] = vt[(size_t)f * vt_stride + p0 + r];
      }
      __syncthreads();
#pragma unroll
      for (int st = 0; st < STEPS; st++) {
        const uint32_t q = (uint32_t)st * EPI + sub;
        const uint32_t feat = (uint32_t)__shfl((int)feat_l, (int)(q & 15u));
        if (q < 16u && wv * 16u + q < np && fl * VEC < tb.rs) {
          float* row = tb.V + (size_t)feat * tb.rs + fl * VEC;
          if (k == (uint32_t)KP) {
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; v++) o[v] = tile[wv * 16u + q][fl * VEC + v];
            store_vec<VEC>(row, o);
          } else {
#pragma unroll
            for (int v = 0; v < VEC; v++)
              if (fl * VEC + v < k) row[v] = tile[wv * 16u + q][fl * VEC + v];   // (the padding columns stay zero)
          }
        }
      }
    }
    __syncthreads();
  }
}

// param_by_pos: the coordinate of list entry li lives at param[(pos0 + li) * pstride] (the factor-major shadow) instead of
// param[feature * pstride] (the table itself)
// the columns of the sweep in LEVEL order: {feature, first entry, one past the last entry} of list position li.  k_als_draw reads them
// as one coalesced 16-byte stream; through seg_list -> seg_feat / seg_rel they were three dependent 4-byte gathers per column, and
// with the short columns of a wide id space (configs[4]: 1.4 entries per column) the draw was bound by exactly those requests.
static __global__ void __launch_bounds__(256)
k_als_ldesc(const uint32_t* __restrict__ seg_list, uint32_t n_list, const uint32_t* __restrict__ seg_feat, const uint32_t* __restrict__ seg_rel,
            uint32_t nseg_total, uint32_t nnz, uint4* __restrict__ out) {
  for (uint32_t li = blockIdx.x * blockDim.x + threadIdx.x; li < n_list; li += gridDim.x * blockDim.x) {
    const uint32_t s = seg_list[li];
    out[li] = make_uint4(seg_feat[s], seg_rel[s], (s + 1 < nseg_total) ? seg_rel[s + 1] : nnz, s);
  }
}

template <bool IS_V, int G>
__global__ void __launch_bounds__(256)
k_als_draw(const TEntry* __restrict__ t_ent, const uint32_t* __restrict__ t_row, const uint4* __restrict__ ldesc, uint32_t n_list,
           float* __restrict__ param, uint32_t pstride, int param_by_pos, uint32_t pos0, EQ* __restrict__ eq,
           double alpha, const double* __restrict__ lambda_g, const double* __restrict__ mu_g, const uint32_t* __restrict__ attr_group,
           int do_sample, uint64_t seed, uint64_t stream, const Shard sh, EQ* __restrict__ delta, float2* __restrict__ dth) {
  // sh: the Gibbs noise of a coordinate is keyed by the feature's GLOBAL id, so a sharded chain draws what the unsharded
  // one draws.  delta != nullptr (feature shards): {e, q} are left alone and the change is recorded instead.
  // dth != nullptr (split step): only the column sums and the draw happen here; {old, new} value of list entry li goes to
  // dth[li] and k_als_rows applies the change to {e, q} in ROW order (streams instead of a second random pass).
  // t_row != nullptr: every value of this level's columns is 1 (one-hot fields): the column is read as a 4-byte row stream (the kernel is
  // bound by fabric REQUESTS -- 52.7 G/s, DESIGN.md section 4b --: half the stream's bytes are 6 % fewer requests)
  constexpr uint32_t GPW = 64 / G;                      // feature groups per wavefront
  const uint32_t lane = (threadIdx.x & 63u) % G;         // lane inside its group
  const uint32_t grp = (threadIdx.x & 63u) / G;
  const uint32_t wave0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t lw = wave0 * GPW; lw < n_list; lw += nwaves * GPW) {
    const uint32_t li = lw + grp;
    const bool have = li < n_list;                       // idle groups still take part in the shuffles
    const uint4 dsc = ldesc[have ? li : n_list - 1];
    const uint32_t j = dsc.x;
    const uint32_t g = attr_group ? attr_group[j] : 0u;  // meta->attr_group(j): the prior of this coordinate (:464-466, :583-585)
    const double lambda = lambda_g[g], mu = mu_g[g];
    const uint32_t a = dsc.y;
    const uint32_t b = have ? dsc.z : a;
    float* pt = param + (size_t)(param_by_pos ? pos0 + (have ? li : n_list - 1) : j) * pstride;
    const double th = (double)(param_by_pos ? *pt : als_param_load(pt));
    double t_he = 0.0, t_hh = 0.0;
    for (uint32_t i = a + lane; i < b; i += G) {
      TEntry te;
      if (t_row) { te.e = __builtin_nontemporal_load(t_row + i); te.x = 1.f; } else te = als_stream8(t_ent + i);
      const double x = (double)te.x;
      const EQ c = eq[te.e];                                       // one 16-byte gather: e and q of the row
      double h;
      if (IS_V) h = x * (c.q - x * th); else h = x;
      if (IS_V) { t_he += h * c.e; } else { t_he += x * (c.e - th * x); }
      t_hh += h * h;
    }
    t_he = group_sum_f64<G>(t_he);
    t_hh = group_sum_f64<G>(t_hh);
    if (!have) continue;
    if (IS_V) t_he -= th * t_hh;                                   // :803
    const double sigma_sqr = 1.0 / (lambda + alpha * t_hh);
    double mean = -sigma_sqr * (alpha * t_he - mu * lambda);
    double nt;
    if (isnan(sigma_sqr) || isinf(sigma_sqr)) nt = 0.0;
    else nt = do_sample ? mean + sqrt(sigma_sqr) * gauss_hash(seed, stream, sh.global(j)) : mean;
    if (isnan(nt) || isinf(nt)) {                                  // keep the old value, caches untouched
      if (dth && lane == 0) dth[li] = make_float2((float)th, (float)th);
      continue;
    }
    const float ntf = (float)nt;
    const double d = th - (double)ntf;                             // theta_old - theta (of the STORED value)
    if (lane == 0) { if (param_by_pos) *pt = ntf; else als_param_store(pt, ntf); }
    if (dth) { if (lane == 0) dth[li] = make_float2((float)th, ntf); continue; }
    if (d != 0.0) {
      // update e (and q): one lane per (row, run of occurrences).  A row holding this feature more than once has its
      // occurrences adjacent (the sort is stable in row order); the first one walks the run sequentially exactly
      // like the reference loop (:839-846: q is updated between the occurrences), the others skip.  Rows are
      // disjoint between lanes and between the wavefronts of a level, so plain read-modify-writes suffice.
      for (uint32_t i = a + lane; i < b; i += G) {
        const TEntry te = als_stream8(t_ent + i);
        if (i > a && t_ent[i - 1].e == te.e) continue;
        EQ c = eq[te.e];
        double ec = c.e;
        if (IS_V) {
          double qc = c.q;
          for (uint32_t i2 = i; i2 < b; i2++) {
            const TEntry t2 = als_stream8(t_ent + i2);
            if (t2.e != te.e) break;
            const double x = (double)t2.x;
            const double h = x * (qc - x * th);
            qc -= x * d;
            ec -= h * d;
          }
          c.q = delta ? qc - c.q : qc;
        } else {
          for (uint32_t i2 = i; i2 < b; i2++) {
            const TEntry t2 = als_stream8(t_ent + i2);
            if (t2.e != te.e) break;
            ec -= (double)t2.x * d;
          }
        }
        if (delta) { EQ dd; dd.e = ec - c.e; dd.q = IS_V ? c.q : 0.0; delta[te.e] = dd; }   // (c.q already holds qc - q_old below)
        else { c.e = ec; eq[te.e] = c; }                           // one 16-byte store
      }
    }
  }
}

// ---- split step, second half: the level's entries in ROW order ------------------------------------------------------
// Features of one level never share a row, so a row holds at most one run of entries of the level (the occurrences of ONE
// feature).  r_row / r_pos / r_x: row, position of the entry's feature in the level's list (index into dth), value -- sorted
// by row, so {e, q} are read and written as a stream and the only gather is the 8-byte {old, new} pair of a table the size
// of the level (cache-resident), instead of a second random pass over the 16-byte {e, q} cache of ALL rows.
#ifndef FMX_ALS_ROWS_NT
#define FMX_ALS_ROWS_NT 1
#endif
template <bool IS_V>
__global__ void __launch_bounds__(256)
k_als_rows(const uint32_t* __restrict__ r_row, const uint32_t* __restrict__ r_pos, const float* __restrict__ r_x, uint32_t n_ent,
           const float2* __restrict__ dth, EQ* __restrict__ eq) {
  constexpr bool NT_EQ = FMX_ALS_ROWS_NT;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_ent; t += gridDim.x * blockDim.x) {
    const uint32_t row = __builtin_nontemporal_load(r_row + t);
    if (t > 0 && r_row[t - 1] == row) continue;                     // a later occurrence: walked by the first of its run
    const float2 tt = dth[__builtin_nontemporal_load(r_pos + t)];
    const double th = (double)tt.x, d = (double)tt.x - (double)tt.y;   // theta_old - theta
    if (d == 0.0) continue;
    EQ c;                                                           // {e, q} is a stream here: keep it out of the way of dth in L2
    if (NT_EQ) { c.e = __builtin_nontemporal_load(&eq[row].e); c.q = __builtin_nontemporal_load(&eq[row].q); } else c = eq[row];
    for (uint32_t i2 = t; i2 < n_ent; i2++) {                       // :839-846: q is updated between the occurrences
      if (i2 > t && r_row[i2] != row) break;
      const double x = (double)__builtin_nontemporal_load(r_x + i2);
      if (IS_V) { const double h = x * (c.q - x * th); c.q -= x * d; c.e -= h * d; }
      else c.e -= x * d;
    }
    if (NT_EQ) { __builtin_nontemporal_store(c.e, &eq[row].e); __builtin_nontemporal_store(c.q, &eq[row].q); } else eq[row] = c;
  }
}
// the same for a level whose entries are the rows 0 .. N-1 in order, one each -- every level of one-hot field data: the row index is the
// entry index (no r_row stream, no run detection), {e, q} is one 16-byte non-temporal load / store, and the load does not wait for the pair
template <bool IS_V>
__global__ void __launch_bounds__(256)
k_als_rows_dense(const uint32_t* __restrict__ r_pos, const float* __restrict__ r_x, uint32_t n_ent, const float2* __restrict__ dth, EQ* __restrict__ eq) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_ent; t += gridDim.x * blockDim.x) {
    const uint32_t pos = __builtin_nontemporal_load(r_pos + t);
    const double x = r_x ? (double)__builtin_nontemporal_load(r_x + t) : 1.0;       // (r_x == nullptr: a level of unit values)
    v2d c = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(eq + t));      // {e, q}
    const float2 tt = dth[pos];
    const double th = (double)tt.x, d = (double)tt.x - (double)tt.y;              // theta_old - theta
    if (d == 0.0) continue;
    if (IS_V) { const double h = x * (c.y - x * th); c.y -= x * d; c.x -= h * d; }
    else c.x -= x * d;
    __builtin_nontemporal_store(c, reinterpret_cast<v2d*>(eq + t));
  }
}
// r_row[t] == t for the n entries of a level?  (flag raised otherwise)
static __global__ void __launch_bounds__(256)
k_als_rows_check_dense(const uint32_t* __restrict__ r_row, const float* __restrict__ r_x, uint32_t n, uint32_t* __restrict__ flag) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    if (r_row[t] != t) atomicOr(flag, 1u);                          // bit 0: not "row t at position t"
    if (r_x[t] != 1.0f) atomicOr(flag, 2u);                         // bit 1: a value other than 1
  }
}
// the rows of X^T's entries alone (for the levels of unit values)
static __global__ void __launch_bounds__(256)
k_als_trow(const TEntry* __restrict__ t_ent, uint32_t nnz, uint32_t* __restrict__ t_row) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += gridDim.x * blockDim.x) t_row[i] = t_ent[i].e;
}
// set-up of the row-ordered lists: key = (level of the entry's feature, row), value = entry index in X^T
static __global__ void __launch_bounds__(256)
k_als_rowkeys(const TEntry* __restrict__ t_ent, const uint32_t* __restrict__ seg_rel, uint32_t nseg, uint32_t nnz,
              const uint32_t* __restrict__ seg_level, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += gridDim.x * blockDim.x) {
    uint32_t lo = 0, hi = nseg;                                     // the segment of entry i: last sg with seg_rel[sg] <= i
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seg_rel[mid] <= i) lo = mid; else hi = mid; }
    keys[i] = ((uint64_t)seg_level[lo] << 32) | (uint64_t)t_ent[i].e;
    vals[i] = i;
  }
}
static __global__ void __launch_bounds__(256)
k_als_rowfill(const TEntry* __restrict__ t_ent, const uint32_t* __restrict__ seg_rel, uint32_t nseg, uint32_t nnz,
              const uint32_t* __restrict__ seg_pos, const uint32_t* __restrict__ vals, uint32_t* __restrict__ r_row,
              uint32_t* __restrict__ r_pos, float* __restrict__ r_x) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += gridDim.x * blockDim.x) {
    const uint32_t i = vals[t];
    uint32_t lo = 0, hi = nseg;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seg_rel[mid] <= i) lo = mid; else hi = mid; }
    const TEntry te = t_ent[i];
    r_row[t] = te.e; r_x[t] = te.x; r_pos[t] = seg_pos[lo];
  }
}

// ----------------------------------------------------------------------------------------------
// `-relation` blocks kept apart from the main rows (FMX_BLOCKS_KEEP): the reference's per-block-row caches
// (relation_cache, fm_learn_mcmc.h:50-58; set-up :478-527 / :603-615, draw_w_rel :734-790, draw_v_rel :849-909), restated.
// A block attribute j has value x_bj in block row b, i.e. in EVERY main row c that maps to b.  With
//   wnum[b] = #{c -> b},  we[b] = sum_{c->b} e_c,  and for a factor:  qb[b] = the block row's own factor sum,
//   qrest(c) = q_c - qb[b(c)],  weq[b] = sum e_c qrest(c),  wc[b] = sum qrest(c),  wc2[b] = sum qrest(c)^2
// the sums over the main rows a draw needs collapse to sums over the block's column:
//   w:  sum h e = sum_b x we[b],                         sum h^2 = sum_b x^2 wnum[b]
//   v:  h_b = x (qb[b] - x v);  sum h e = sum_b (h_b we[b] + x weq[b]);  sum h^2 = sum_b (h_b^2 wnum + 2 wc x h_b + x^2 wc2)
// and a new value changes the caches of its column only; the main rows are touched twice per (block, family): once to build
// the caches (k_rel_aggregate), once to push the accumulated changes back (k_rel_sync):
//   e_c += dy[b] + qrest(c) (qb[b] - qb0[b]),   q_c = qrest(c) + qb[b]
// cache: struct of arrays [7][B]: 0 we, 1 weq, 2 wc, 3 wc2, 4 qb, 5 dy, 6 qb0.
// ----------------------------------------------------------------------------------------------
template <bool IS_V, int G>
__global__ void __launch_bounds__(256)
k_rel_aggregate(const uint32_t* __restrict__ brow_ptr, const uint32_t* __restrict__ brow_list, uint32_t B, const EQ* __restrict__ eq,
                double* __restrict__ cache) {
  const uint32_t lane = (threadIdx.x & 63u) % G, grp = (threadIdx.x & 63u) / G;
  constexpr uint32_t GPW = 64 / G;
  const uint32_t wave0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t b0 = wave0 * GPW; b0 < B; b0 += nwaves * GPW) {
    const uint32_t b = b0 + grp;
    const bool have = b < B;
    const uint32_t a0 = have ? brow_ptr[b] : 0u, a1 = have ? brow_ptr[b + 1] : 0u;
    const double qb = (IS_V && have) ? cache[(size_t)4 * B + b] : 0.0;
    double we = 0.0, weq = 0.0, wc = 0.0, wc2 = 0.0;
    for (uint32_t i = a0 + lane; i < a1; i += G) {
      const EQ c = eq[brow_list[i]];
      we += c.e;
      if (IS_V) { const double qr = c.q - qb; weq += c.e * qr; wc += qr; wc2 += qr * qr; }
    }
    we = group_sum_f64<G>(we);
    if (IS_V) { weq = group_sum_f64<G>(weq); wc = group_sum_f64<G>(wc); wc2 = group_sum_f64<G>(wc2); }
    if (have && lane == 0) {
      cache[b] = we;
      if (IS_V) { cache[(size_t)B + b] = weq; cache[(size_t)2 * B + b] = wc; cache[(size_t)3 * B + b] = wc2; cache[(size_t)6 * B + b] = qb; }
      cache[(size_t)5 * B + b] = 0.0;
    }
  }
}

// one group of G lanes per block attribute of the current level (columns of the block's X^T: {block row, value})
template <bool IS_V, int G>
__global__ void __launch_bounds__(256)
k_rel_draw(const TEntry* __restrict__ t_ent, const uint32_t* __restrict__ seg_feat, const uint32_t* __restrict__ seg_rel,
           uint32_t nseg_total, uint32_t nnz, const uint32_t* __restrict__ seg_list, uint32_t n_list,
           float* __restrict__ param, uint32_t pstride, uint32_t attr_offset, const uint32_t* __restrict__ brow_ptr, uint32_t B,
           double* __restrict__ cache, double alpha, const double* __restrict__ lambda_g, const double* __restrict__ mu_g,
           const uint32_t* __restrict__ attr_group, int do_sample, uint64_t seed, uint64_t stream) {
  constexpr uint32_t GPW = 64 / G;
  const uint32_t lane = (threadIdx.x & 63u) % G, grp = (threadIdx.x & 63u) / G;
  const uint32_t wave0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  double* we = cache; double* weq = cache + B; double* wc = cache + (size_t)2 * B; double* wc2 = cache + (size_t)3 * B;
  double* qb = cache + (size_t)4 * B; double* dy = cache + (size_t)5 * B;
  for (uint32_t lw = wave0 * GPW; lw < n_list; lw += nwaves * GPW) {
    const uint32_t li = lw + grp;
    const bool have = li < n_list;
    const uint32_t s = seg_list[have ? li : n_list - 1];
    const uint32_t j = seg_feat[s] + attr_offset;                  // global attribute id (libfm.cpp:213-216)
    const uint32_t g = attr_group ? attr_group[j] : 0u;
    const double lambda = lambda_g[g], mu = mu_g[g];
    const uint32_t a = seg_rel[s];
    const uint32_t b = have ? ((s + 1 < nseg_total) ? seg_rel[s + 1] : nnz) : a;
    float* pt = param + (size_t)j * pstride;
    const double th = (double)*pt;
    double t_he = 0.0, t_hh = 0.0;
    for (uint32_t i = a + lane; i < b; i += G) {
      const TEntry te = t_ent[i];
      const double x = (double)te.x;
      const double wnum = (double)(brow_ptr[te.e + 1] - brow_ptr[te.e]);
      if (IS_V) {
        const double h = x * (qb[te.e] - x * th);
        t_he += h * we[te.e] + x * weq[te.e];                        // :858
        t_hh += h * h * wnum + 2 * wc[te.e] * x * h + x * x * wc2[te.e];   // :859
      } else {
        t_he += x * we[te.e];                                        // :744
        t_hh += x * x * wnum;                                        // :745
      }
    }
    t_he = group_sum_f64<G>(t_he);
    t_hh = group_sum_f64<G>(t_hh);
    if (!have) continue;
    t_he -= th * t_hh;                                               // :749, :863
    const double sigma_sqr = 1.0 / (lambda + alpha * t_hh);
    const double mean = -sigma_sqr * (alpha * t_he - mu * lambda);
    double nt;
    if (isnan(sigma_sqr) || isinf(sigma_sqr)) nt = 0.0;
    else nt = do_sample ? mean + sqrt(sigma_sqr) * gauss_hash(seed, stream, j) : mean;
    if (isnan(nt) || isinf(nt)) continue;
    const float ntf = (float)nt;
    const double d = th - (double)ntf;                               // theta_old - theta (of the STORED value)
    if (lane == 0) *pt = ntf;
    if (d != 0.0) {
      // the caches of the column's block rows: one lane per (block row, run of occurrences); a block row holding the
      // attribute more than once has its occurrences adjacent and is walked sequentially by the first (as in k_als_draw)
      for (uint32_t i = a + lane; i < b; i += G) {
        const TEntry te = t_ent[i];
        if (i > a && t_ent[i - 1].e == te.e) continue;
        const uint32_t r = te.e;
        const double wnum = (double)(brow_ptr[r + 1] - brow_ptr[r]);
        double c_we = we[r], c_dy = dy[r];
        if (IS_V) {
          double c_weq = weq[r], c_q = qb[r];
          const double c_wc = wc[r], c_wc2 = wc2[r];
          for (uint32_t i2 = i; i2 < b; i2++) {
            const TEntry t2 = t_ent[i2];
            if (t2.e != r) break;
            const double x = (double)t2.x;
            const double h = x * (c_q - x * th);                     // :898
            c_we -= d * (h * wnum + x * c_wc);                       // :899
            c_q -= d * x;                                            // :900
            c_weq -= d * (h * c_wc + x * c_wc2);                     // :901
            c_dy -= d * h;                                           // :902  y += (v - v_old) h
          }
          weq[r] = c_weq; qb[r] = c_q;
        } else {
          for (uint32_t i2 = i; i2 < b; i2++) {
            const TEntry t2 = t_ent[i2];
            if (t2.e != r) break;
            const double x = (double)t2.x;
            c_we -= x * d * wnum;                                    // :786
            c_dy -= d * x;                                           // :787  y += (w - w_old) h
          }
        }
        we[r] = c_we; dy[r] = c_dy;
      }
    }
  }
}

// push the block's accumulated changes back into the main rows' {e, q}
template <bool IS_V>
__global__ void __launch_bounds__(256)
k_rel_sync(const uint32_t* __restrict__ map, uint32_t n, uint32_t B, const double* __restrict__ cache, EQ* __restrict__ eq) {
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
    const uint32_t b = map[c];
    EQ v = eq[c];
    if (IS_V) {
      const double qb = cache[(size_t)4 * B + b], qb0 = cache[(size_t)6 * B + b];
      const double qr = v.q - qb0;
      v.e += cache[(size_t)5 * B + b] + qr * (qb - qb0);            // :631
      v.q = qr + qb;                                                 // :632
    } else {
      v.e += cache[(size_t)5 * B + b];                               // :506
    }
    eq[c] = v;
  }
}
// the block rows' factor sums of factor f become the block's qb for the coming sweep of v_f
static __global__ void k_rel_load_qb(double* __restrict__ cache, const double* __restrict__ qb_f, uint32_t B) {
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) cache[(size_t)4 * B + b] = qb_f[b];
}
// re-prediction of a slot with kept blocks: add the block rows' partial sums (c = lin - 0.5 sum of squares, q_f) to the main rows'
static __global__ void __launch_bounds__(256)
k_rel_combine(const uint32_t* __restrict__ map, uint32_t n, uint32_t B, const double* __restrict__ cpart_b, const double* __restrict__ qb_all,
              int k, double* __restrict__ cpart, double* __restrict__ q) {
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
    const uint32_t b = map[c];
    cpart[c] += cpart_b[b];
    for (int f = 0; f < k; f++) q[(size_t)f * n + c] += qb_all[(size_t)f * B + b];
  }
}
// the fp32 form for fmx_predict / fmx_evaluate: partial buffers [rows][KP] + [rows] (k_rowsums<KP, true, false>)
template <int KP>
__global__ void __launch_bounds__(256)
k_rel_add_partial(const uint32_t* __restrict__ map, uint32_t n, uint32_t B, const float* __restrict__ pb, float* __restrict__ pm) {
  const uint64_t total = (uint64_t)n * (KP + 1);
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = (uint32_t)(t / (KP + 1)); const uint32_t f = (uint32_t)(t % (KP + 1));
    const uint32_t b = map[c];
    if (f < KP) pm[(size_t)c * KP + f] += pb[(size_t)b * KP + f];
    else pm[(size_t)n * KP + c] += pb[(size_t)B * KP + b];
  }
}

// features without a training column: the empty-row draw (:467-476, :586-595): theta = prior mean mu = 0
// (sigma^2 = 1/lambda; lambda = 0 -> sigma^2 = inf -> theta = 0); MCMC: mu + N(0,1)/sqrt(lambda)
static __global__ void __launch_bounds__(256)
k_als_unseen(const uint8_t* __restrict__ seen, uint64_t n_local, float* __restrict__ param, uint32_t pstride,
             const double* __restrict__ lambda_g, const double* __restrict__ mu_g, const uint32_t* __restrict__ grp,
             int do_sample, uint64_t seed, uint64_t stream, const Shard sh) {
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_local; j += (uint64_t)gridDim.x * blockDim.x)
    if (!seen[j]) {
      const uint32_t g = grp ? grp[j] : 0u;
      const double lambda = lambda_g[g], mu = mu_g[g];
      const double sigma_sqr = 1.0 / lambda;
      double nt;
      if (isnan(sigma_sqr) || isinf(sigma_sqr)) nt = 0.0;
      else nt = do_sample ? mu + sqrt(sigma_sqr) * gauss_hash(seed, stream, sh.global(j)) : mu;
      if (isnan(nt) || isinf(nt)) continue;
      param[(size_t)j * pstride] = (float)nt;
    }
}

// the same for ALL factors of the unseen features in one pass over the parameter rows (one launch per sweep instead of
// one per factor: at n = 1e8 the per-factor form re-reads seen[] and scatters 4-byte writes k times).  Legal because an
// unseen feature's draw depends on nothing but its prior (no data rows), and the priors of a sweep are fixed before
// the sweep starts (the hyper-prior statistics are taken at sweep start).  prior: [1 + k][2][G] (row 1+f: lambda, mu).
// Random stream of (f, j) = the per-factor kernel's: stream0 + f.
template <int KP>
__global__ void __launch_bounds__(256)
k_als_unseen_v(const uint8_t* __restrict__ seen, uint64_t n_local, const Tab tb, int k, const double* __restrict__ prior, uint32_t G,
               const uint32_t* __restrict__ grp, int do_sample, uint64_t seed, uint64_t stream0, const Shard sh) {
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR, EPI = Map<KP>::EPI;
  const uint32_t lane = threadIdx.x & 63u, sub = lane / LPR, fl = lane % LPR;
  const uint64_t wave0 = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  // one attribute group (no `-meta`): a lane's factors have ONE prior each for the whole table -- sigma and mu are taken once, not per feature
  // (the fp64 divide + square root per coordinate made this stream compute-bound at configs[4]'s shape: 26 ms for 25 GB, round 5)
  const bool one_group = (grp == nullptr);
  double sd1[VEC], mu1[VEC]; bool zero1[VEC];
#pragma unroll
  for (int v = 0; v < VEC; v++) {
    const int f = (int)fl * VEC + v;
    sd1[v] = 0.0; mu1[v] = 0.0; zero1[v] = false;
    if (one_group && f < k) {
      const double* row = prior + (size_t)(1 + f) * 2 * G;
      const double sigma_sqr = 1.0 / row[0];
      zero1[v] = isnan(sigma_sqr) || isinf(sigma_sqr);
      sd1[v] = sqrt(sigma_sqr); mu1[v] = row[G];
    }
  }
  for (uint64_t j0 = wave0 * EPI; j0 < n_local; j0 += nwaves * EPI) {
    const uint64_t j = j0 + sub;
    if (j >= n_local || seen[j]) continue;
    const uint32_t g = grp ? grp[j] : 0u;
#pragma unroll
    for (int v = 0; v < VEC; v++) {
      const int f = (int)fl * VEC + v;
      if (f >= k) continue;
      double nt;
      if (one_group) {
        if (zero1[v]) nt = 0.0;
        else nt = do_sample ? mu1[v] + sd1[v] * gauss_hash(seed, stream0 + (uint64_t)f, sh.global(j)) : mu1[v];
      } else {
        const double* row = prior + (size_t)(1 + f) * 2 * G;
        const double lambda = row[g], mu = row[G + g];
        const double sigma_sqr = 1.0 / lambda;
        if (isnan(sigma_sqr) || isinf(sigma_sqr)) nt = 0.0;
        else nt = do_sample ? mu + sqrt(sigma_sqr) * gauss_hash(seed, stream0 + (uint64_t)f, sh.global(j)) : mu;
      }
      if (isnan(nt) || isinf(nt)) continue;
      tb.V[(size_t)j * tb.rs + f] = (float)nt;
    }
  }
}

}  // namespace fmx

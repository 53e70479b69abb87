// fmx_kernels.h -- hand-written CDNA4 (gfx950) kernels of the FM hot path.
//
// Reference algorithms (restated, never copied):
//   fm_model::predict   /root/reference/src/fm_core/fm_model.h:105-127   (sum / sum-of-squares trick)
//   fm_SGD              /root/reference/src/fm_core/fm_sgd.h:33-51
//   loss multiplier     /root/reference/src/libfm/src/fm_learn_sgd_element.h:58-65
//   evaluate            /root/reference/src/libfm/src/fm_learn.h:113-153
//
// Device layout: V is FEATURE-major fp32, V[j*KP + f], KP = power of two >= k (padding factors are 0 and
// stay 0 under the update), so the k factors of one feature are ONE coalesced segment (256 B at k=64).
// w is fp32 [n].  w0 is one fp64 scalar.  Rows are the reference's AoS {u32 id; f32 value} CSR.
//
// Wavefront mapping (64 lanes, no 32-wide assumptions): one wavefront per example.
//   VEC = max(1, KP/64) floats per lane, LPR = KP/VEC lanes cover one V row, EPI = 64/LPR rows are
//   fetched by ONE wave-wide load instruction (k=64: 1 row = 256 B; k=32: 2 rows; k=8: 8 rows).
//   Lane (g, f) = (lane / LPR, lane % LPR) owns factors [f*VEC, f*VEC+VEC) of the entries i == g (mod EPI);
//   sum_f needs no cross-lane traffic until the final EPI-way butterfly + one 64-lane reduction.
// All kernels are HBM-bandwidth bound random gathers/scatters: no MFMA.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace fmx {

struct Entry { uint32_t id; float value; };   // sparse_entry<float>, fmatrix.h:34-37

template <int KP> struct Map {
  static constexpr int VEC = (KP >= 64) ? KP / 64 : 1;
  static constexpr int LPR = KP / VEC;
  static constexpr int EPI = 64 / LPR;
};

// parameter table: V rows of `rs` floats (rs >= KP); w(j) lives at w[j*ws].  In the co-located layout
// w = V + KP and ws = rs: the linear weight sits in the same DRAM page as its factor row, so one feature
// costs ONE row activation instead of two (measured: the separate w[] gather cost 35-50 % of the step).
struct Tab { float* V; float* w; uint32_t rs; uint32_t ws; };

struct Hyper {            // per-launch scalars (fm_model.h:56-57, fm_learn_sgd.h:42, fm_learn.h:41-45)
  float lr, reg0, regw, regv;
  float min_target, max_target;
  int   task, k0, k1;
  double lr_d, reg0_d, regw_d, regv_d, min_d, max_d;   // unrounded copies for the fp64 sequential kernel
  int   sgda;              // 1: the multiplier of fm_learn_sgd_element_adapt_reg (2 (p - y) for regression, :142)
};

// counter-hash helper: identical definition in oracle/fm_oracle.c (fmo_mix64)
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27; x *= 0x94D049BB133111EBULL;
  x ^= x >> 31;
  return x;
}

// ----------------------------------------------------------------------------------------------
// Which shard owns feature j, and where it lives there (fmx_config::shard_rank / shard_world / shard_hash).
//   plain : owner = j mod W, local row = j div W
//   hashed: the same on p(j), p a pseudo-random PERMUTATION of [0, n): a 4-round Feistel network on 2*half_bits bits
//           (round function = mix64 of the half and the round number), cycle-walked into [0, n) -- still a bijection, so
//           the local table stays dense (n_local = ceil((n - rank) / W)) and a local row knows its global id (inverse).
//           Balanced for any id structure (ids that are all multiples of W, one field per residue class, ...).
// ----------------------------------------------------------------------------------------------
struct Shard {
  uint64_t n;
  uint32_t rank, world, hashed, half_bits;
  __host__ __device__ __forceinline__ uint32_t round_fn(uint32_t half, uint32_t round) const {
    return (uint32_t)mix64(((uint64_t)half << 3 | round) + 0x5851F42D4C957F2DULL) & ((1u << half_bits) - 1u);
  }
  __host__ __device__ __forceinline__ uint32_t perm(uint32_t j) const {
    const uint32_t mask = (1u << half_bits) - 1u;
    uint32_t x = j;
    do {
      uint32_t l = x >> half_bits, r = x & mask;
      for (uint32_t i = 0; i < 4; i++) { const uint32_t t = l ^ round_fn(r, i); l = r; r = t; }
      x = (l << half_bits) | r;
    } while ((uint64_t)x >= n);
    return x;
  }
  __host__ __device__ __forceinline__ uint32_t perm_inv(uint32_t p) const {
    const uint32_t mask = (1u << half_bits) - 1u;
    uint32_t x = p;
    do {
      uint32_t l = x >> half_bits, r = x & mask;
      for (uint32_t i = 4; i-- > 0;) { const uint32_t t = r ^ round_fn(l, i); r = l; l = t; }
      x = (l << half_bits) | r;
    } while ((uint64_t)x >= n);
    return x;
  }
  __host__ __device__ __forceinline__ uint32_t placed(uint32_t j) const { return hashed ? perm(j) : j; }
  __host__ __device__ __forceinline__ bool owns(uint32_t j) const { return world == 1 || placed(j) % world == rank; }
  __host__ __device__ __forceinline__ uint32_t local(uint32_t j) const { return world == 1 ? j : placed(j) / world; }
  // (owner test and local row in one evaluation of the permutation)
  __host__ __device__ __forceinline__ bool place(uint32_t j, uint32_t* local_row) const {
    if (world == 1) { *local_row = j; return true; }
    const uint32_t p = placed(j);
    *local_row = p / world;
    return p % world == rank;
  }
  __host__ __device__ __forceinline__ uint32_t global(uint64_t local_row) const {
    if (world == 1) return (uint32_t)local_row;
    const uint32_t p = (uint32_t)(local_row * world + rank);
    return hashed ? perm_inv(p) : p;
  }
};

// ----------------------------------------------------------------------------------------------
// small device helpers
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
__device__ __forceinline__ double wave_sum_d(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
// 64-lane sum through DPP (no LDS crossbar traffic): two quad_perm butterflies, two row rotates, then the
// gfx9 row_bcast15 / row_bcast31 steps leave the total in lane 63; returned wave-uniform.
__device__ __forceinline__ float wave_sum_dpp(float x) {
#define FMX_DPP_ADD(ctrl, rmask)                                                                        \
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xf, false))
  FMX_DPP_ADD(0xB1, 0xf);    // quad_perm(1,0,3,2)
  FMX_DPP_ADD(0x4E, 0xf);    // quad_perm(2,3,0,1)
  FMX_DPP_ADD(0x124, 0xf);   // row_ror:4
  FMX_DPP_ADD(0x128, 0xf);   // row_ror:8
  FMX_DPP_ADD(0x142, 0xa);   // row_bcast:15 -> rows 1,3
  FMX_DPP_ADD(0x143, 0xc);   // row_bcast:31 -> rows 2,3
#undef FMX_DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
// sum over ONE group of C consecutive lanes (C = 16 / 32 / 64; group `sub` = lanes [sub*C, sub*C + C)), returned wave-uniform: the four
// intra-row steps of wave_sum_dpp leave every lane with its DPP row's sum, row_bcast15 / 31 then fold rows into halves / the wavefront.
// What the sub-piece form of the bias recurrence reduces a micro-chunk of 16 / 32 / 64 examples with (k_scan1, scan_small).
template <int C> __device__ __forceinline__ float group_sum_dpp(float x, uint32_t sub) {
  static_assert(C == 16 || C == 32 || C == 64, "one DPP row, half a wavefront or the wavefront");
#define FMX_DPP_ADD(ctrl, rmask)                                                                        \
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xf, false))
  FMX_DPP_ADD(0xB1, 0xf);    // quad_perm(1,0,3,2)
  FMX_DPP_ADD(0x4E, 0xf);    // quad_perm(2,3,0,1)
  FMX_DPP_ADD(0x124, 0xf);   // row_ror:4
  FMX_DPP_ADD(0x128, 0xf);   // row_ror:8
  if constexpr (C >= 32) FMX_DPP_ADD(0x142, 0xa);   // row_bcast:15 -> rows 1,3 hold rows 0+1, 2+3
  if constexpr (C >= 64) FMX_DPP_ADD(0x143, 0xc);   // row_bcast:31 -> row 3 holds everything
#undef FMX_DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), (int)(sub * (uint32_t)C + (uint32_t)C - 1u)));
}
// all-reduce over the EPI sub-groups that hold the same factor (lane bits >= log2(LPR))
template <int LPR> __device__ __forceinline__ float subgroup_allsum(float x) {
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) x += __shfl_xor(x, o);
  return x;
}
template <int EPI> __device__ __forceinline__ uint32_t bcast_u32(uint32_t v, uint32_t idx) {
  if constexpr (EPI == 1) return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)idx);   // idx is wave-uniform
  else return (uint32_t)__shfl((int)v, (int)idx);
}
template <int EPI> __device__ __forceinline__ float bcast_f32(float v, uint32_t idx) {
  return __uint_as_float(bcast_u32<EPI>(__float_as_uint(v), idx));
}
template <int VEC> __device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&out)[VEC]) {
  if constexpr (VEC == 1) { out[0] = p[0]; }
  else if constexpr (VEC == 2) { float2 t = *reinterpret_cast<const float2*>(p); out[0] = t.x; out[1] = t.y; }
  else {
#pragma unroll
    for (int c = 0; c < VEC / 4; c++) {
      float4 t = reinterpret_cast<const float4*>(p)[c];
      out[4 * c] = t.x; out[4 * c + 1] = t.y; out[4 * c + 2] = t.z; out[4 * c + 3] = t.w;
    }
  }
}
template <int VEC> __device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&in)[VEC]) {
  if constexpr (VEC == 1) { p[0] = in[0]; }
  else if constexpr (VEC == 2) { *reinterpret_cast<float2*>(p) = make_float2(in[0], in[1]); }
  else {
#pragma unroll
    for (int c = 0; c < VEC / 4; c++)
      reinterpret_cast<float4*>(p)[c] = make_float4(in[4 * c], in[4 * c + 1], in[4 * c + 2], in[4 * c + 3]);
  }
}
// V-row accesses with a non-temporal hint.  The 4*KP-byte rows stream through (every row is touched once per example
// and the table is far larger than any cache): without the hint they are allocated in -- and later evicted from -- the
// L2 / Infinity Cache path for nothing, and push out the lines that do have reuse (the 128-B lines holding w_j, the
// S buffer of the minibatch step).  Measured on MI355X, north-star shape: fused step 211 -> 270 M examples/s, and
// still 251 M examples/s with a 205 GB table.
// FMX_V_NT bits (default all on): 1 fused loads, 2 fused stores, 4 gather (row_sums) loads, 8 update (row_apply /
// k_apply_seg) loads + stores.
#ifndef FMX_V_NT
#define FMX_V_NT 15
#endif
template <int VEC, int BIT> __device__ __forceinline__ void load_row(const float* __restrict__ p, float (&out)[VEC]) {
  if constexpr ((FMX_V_NT & BIT) != 0) {
    if constexpr (VEC == 1) { out[0] = __builtin_nontemporal_load(p); }
    else if constexpr (VEC == 2) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      const v2f t = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(p)); out[0] = t.x; out[1] = t.y;
    } else {
      typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int c = 0; c < VEC / 4; c++) {
        const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p) + c);
        out[4 * c] = t.x; out[4 * c + 1] = t.y; out[4 * c + 2] = t.z; out[4 * c + 3] = t.w;
      }
    }
  } else {
    load_vec<VEC>(p, out);
  }
}
template <int VEC, int BIT> __device__ __forceinline__ void store_row(float* __restrict__ p, const float (&in)[VEC]) {
  if constexpr ((FMX_V_NT & BIT) != 0) {
    if constexpr (VEC == 1) { __builtin_nontemporal_store(in[0], p); }
    else if constexpr (VEC == 2) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      v2f t; t.x = in[0]; t.y = in[1]; __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(p));
    } else {
      typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int c = 0; c < VEC / 4; c++) {
        v4f t; t.x = in[4 * c]; t.y = in[4 * c + 1]; t.z = in[4 * c + 2]; t.w = in[4 * c + 3];
        __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p) + c);
      }
    }
  } else {
    store_vec<VEC>(p, in);
  }
}

// a parameter row is tb.rs floats: the factor count rounded up to 16 floats (one 64-byte sector) -- NOT the power of two KP the lane mapping is
// built on (k = 100: rows of 112 floats, 448 B, where the padded 128 moved 512).  The lanes whose elements lie beyond the row take no part in
// row accesses: they read zeros (a sum over the lanes is unchanged) and store nothing.  rs == KP for every power of two.
template <int VEC, int BIT> __device__ __forceinline__ void row_ld(const Tab& tb, size_t id, uint32_t off, float (&out)[VEC]) {
  if (off < tb.rs) load_row<VEC, BIT>(tb.V + id * tb.rs + off, out);
  else {
#pragma unroll
    for (int v = 0; v < VEC; v++) out[v] = 0.f;
  }
}
template <int VEC, int BIT> __device__ __forceinline__ void row_st(const Tab& tb, size_t id, uint32_t off, const float (&in)[VEC]) {
  if (off < tb.rs) store_row<VEC, BIT>(tb.V + id * tb.rs + off, in);
}

// the row entries ({id, value}, 8 bytes) and the 4-byte gathers of a linear weight: plain loads.  (Non-temporal / agent-scope variants of
// both, write-through row stores, a sixth wavefront per SIMD and the in-launch publish of S_e were measured in rounds 1-4 and lost or
// bought nothing: scripts/experiments/r04_variants.patch re-creates them, DESIGN.md section 4 has the numbers.)
template <class T> __device__ __forceinline__ T load_stream8(const T* p) {
  static_assert(sizeof(T) == 8, "8-byte records");
  return *p;
}
__device__ __forceinline__ float load_w(const float* p) { return *p; }
// loads that are served by the L2 -- they bypass the per-CU L1, which no other CU's store ever refreshes: what a persistent kernel reads the
// words of other workgroups OF ITS OWN DIE with (fmx_xcd_kernels.h).  Two forms skip the L1 (scripts/ubench/load_flavours.hip: 105-110 ns
// per dependent load for either, 72 ns for an L1 hit): device scope (sc1) keeps the line in the L2 like a plain load -- for what is read
// again soon (frequent rows, sums, multipliers, flags); non-temporal marks it to be evicted first -- for the rows that pass through once
// (ld_l2_stream), so that they do not push the frequent rows out of the die's 4 MiB (profiles/r06_criteo_hops.txt).
__device__ __forceinline__ float ld_l2(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_l2(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_l2(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float ld_l2_stream(const float* p) { return __builtin_nontemporal_load(p); }
// (words that cross dies -- the membership counters of a launch -- are read at device scope)
__device__ __forceinline__ unsigned ld_dev(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// loss multiplier, fm_learn_sgd_element.h:58-65
__device__ __forceinline__ float multiplier(const Hyper& h, float p, float y) {
  if (h.task == 0) {
    p = fminf(h.max_target, p);
    p = fmaxf(h.min_target, p);
    return h.sgda ? 2.0f * (p - y) : -(y - p);
  }
  return -y * (1.0f - 1.0f / (1.0f + __expf(-y * p)));
}

// same multiplier with the hardware reciprocal (1 ulp) -- used on the serial w0 recurrence where the IEEE
// division sequence sits on the critical path
// the same with the task resolved at compile time: no branch between the elements of an unrolled block, so their LDS
// reads and transcendentals overlap (k_scan)
template <int TASK> __device__ __forceinline__ float multiplier_task(const Hyper& h, float p, float y) {
  if constexpr (TASK == 0) {
    p = fminf(h.max_target, p);
    p = fmaxf(h.min_target, p);
    return h.sgda ? 2.0f * (p - y) : -(y - p);
  } else {
    return -y * (1.0f - __builtin_amdgcn_rcpf(1.0f + __expf(-y * p)));
  }
}
__device__ __forceinline__ float multiplier_fast(const Hyper& h, float p, float y) {
  if (h.task == 0) {
    p = fminf(h.max_target, p);
    p = fmaxf(h.min_target, p);
    return -(y - p);
  }
  return -y * (1.0f - __builtin_amdgcn_rcpf(1.0f + __expf(-y * p)));
}

// ----------------------------------------------------------------------------------------------
// Device-side hand-off between the launch stream and the recurrence's side stream (the one-pass batch rule at large batches).
// The bias a batch's multipliers use is produced by the one-workgroup recurrence of an EARLIER batch on another stream.  Ordering the
// two streams with events costs four queue packets per batch (record + wait on either side, ~10 us of idle chip between the launches of
// a 1 ms batch); here the data itself is the signal:
//   * the bias slots W[b] (one per batch of the epoch) start as W0_PENDING; the recurrence publishes its result with ONE agent-scope
//     8-byte store and a reader that still finds the sentinel waits (bounded; a time-out raises the handle's error flag);
//   * "k_fused of batch b has completed" is a counter the first workgroup of the NEXT launch on the same stream (the deferred-feature
//     pass of batch b) advances: that launch starts only after k_fused retired (in-order stream), and the recurrence kernel -- resident
//     on its own stream since the previous one finished -- polls it, then invalidates its caches (agent-scope acquire) and reads rest[].
// ----------------------------------------------------------------------------------------------
constexpr unsigned long long W0_PENDING = 0x7FF8DEADBEEF0001ull;       // a quiet NaN no recurrence produces
constexpr uint32_t HANDOFF_SPINS = 1u << 21;                           // x ~1 us per poll: seconds, then the error flag instead of a hang
__device__ __forceinline__ unsigned long long handoff_peek(const double* p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// *timed_out: the slot never arrived (bounded wait): the error word gets bit 1 and the CALLER takes no step with it (k_fused: multiplier 0
// for the example -- the parameters stay valid numbers, the epoch reports FMX_STAT_HANDOFF_TIMEOUT and the handle falls back to events)
__device__ __forceinline__ double handoff_wait(const double* p, unsigned long long u, uint32_t* err, bool* timed_out) {
  for (uint32_t t = 0; u == W0_PENDING && t < HANDOFF_SPINS; t++) { __builtin_amdgcn_s_sleep(16); u = handoff_peek(p); }
  if (u == W0_PENDING) { atomicOr(err, 1u); u = 0ull; *timed_out = true; }
  return __longlong_as_double((long long)u);
}
__device__ __forceinline__ void handoff_publish(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// counter side: the waiter is ONE thread of the recurrence workgroup (the others sit at the barrier behind it)
struct Handoff { const unsigned long long* ctr; unsigned long long need; uint32_t* err; };
// returns false when the counter never arrived (bounded wait; error bit 2): rest[] of the batch may be incomplete then, and the caller
// must NOT evaluate the recurrence on it -- it hands the incoming bias on unchanged (handoff_pass_on) so that nothing downstream waits
__device__ __forceinline__ bool handoff_wait_counter(const Handoff hw) {
  if (!hw.ctr) return true;
  __shared__ uint32_t s_handoff_ok;
  if (threadIdx.x == 0) {
    unsigned long long c = __hip_atomic_load(hw.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t t = 0; c < hw.need && t < HANDOFF_SPINS * 4u; t++) { __builtin_amdgcn_s_sleep(32); c = __hip_atomic_load(hw.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (c < hw.need) atomicOr(hw.err, 2u);
    s_handoff_ok = (c >= hw.need) ? 1u : 0u;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                 // what the launch stream wrote before the counter moved
  return s_handoff_ok != 0u;
}
// the recurrence of a batch whose hand-off timed out: the bias goes on as it came (one thread of the first workgroup)
__device__ __forceinline__ void handoff_pass_on(const double* w0_in, double* w0_out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) handoff_publish(w0_out, *w0_in);
}
// Do two kernels on two streams of this process run CONCURRENTLY right now?  Each sets its own flag and waits (bounded, ~20 ms) for the
// other's: both succeed only when both are resident at the same time.  Under a profiler that serialises dispatches (rocprofv3 --pmc),
// AMD_SERIALIZE_KERNEL or a partitioned device one of them runs into its bound -- then the device-side hand-off (whose waits are satisfied
// by LATER launches) must not be used and the streams are ordered by events.  flags[0 / 1]: "I am here", flags[2 / 3]: result (1 = met).
static __global__ void k_concurrency_probe(unsigned* flags, unsigned me) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  __hip_atomic_store(flags + me, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned other = 0;
  for (uint32_t t = 0; t < 20000u && !other; t++) { __builtin_amdgcn_s_sleep(16); other = __hip_atomic_load(flags + (me ^ 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __hip_atomic_store(flags + 2u + me, other ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __global__ void k_handoff_signal(unsigned long long* ctr, unsigned long long val) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(ctr, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __global__ void k_handoff_arm(double* W, uint32_t n) {         // W[1 .. n] = pending
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    reinterpret_cast<unsigned long long*>(W)[1 + i] = W0_PENDING;
}

// ----------------------------------------------------------------------------------------------
// pass 1 of a row: the sums of fm_model.h:110-125.
//   sum[v]  : sum_f for the lane's factors (complete over the row only AFTER subgroup_allsum when EPI>1)
//   sq      : this lane's share of  sum_f sum_i (v x)^2
//   lin     : this lane's share of  sum_i w[id_i] x_i
// ----------------------------------------------------------------------------------------------
// wside (optional): the row's slice of the slot's WEIGHT SIDE STREAM -- wside[i] is either w[id_i] as it stands or NaN ("gather it").
// A 4-byte w_j out of its own array costs a 64-byte fabric request per entry (a third of a predict pass's requests for 1.5 % of its
// bytes); the stream costs 4 coalesced bytes.  Maintained by k_fused<EXACT> for the entries that are the LAST occurrence of their
// feature in the slot and are updated by their own example (FMX_FLAG_KEEP_WSIDE; fmx_sgd.hip), NaN for everything else.
template <int KP, int U>
__device__ __forceinline__ void row_sums(const Entry* __restrict__ ent, uint32_t size,
                                         const Tab tb, int k1,
                                         float (&sum)[Map<KP>::VEC], float& sq, float& lin, const float* __restrict__ wside = nullptr) {
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR, EPI = Map<KP>::EPI;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t g = lane / LPR, f = lane % LPR;
#pragma unroll
  for (int v = 0; v < VEC; v++) sum[v] = 0.f;
  sq = 0.f; lin = 0.f;
  for (uint32_t base = 0; base < size; base += 64) {
    const uint32_t cnt = min(64u, size - base);
    Entry e; e.id = 0; e.value = 0.f;
    if (lane < cnt) {
      e = load_stream8(ent + base + lane);
      if (k1) {
        float wv = wside ? __builtin_nontemporal_load(wside + base + lane) : __builtin_nanf("");
        if (wv != wv) wv = load_w(tb.w + (size_t)e.id * tb.ws);
        lin += wv * e.value;
      }
    }
    for (uint32_t i = 0; i < cnt; i += EPI * U) {
      float vr[U][VEC]; float xs[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t idx = i + u * EPI + g;
        const uint32_t id = bcast_u32<EPI>(e.id, idx & 63u);
        xs[u] = bcast_f32<EPI>(e.value, idx & 63u);
        if (idx < cnt) {
          row_ld<VEC, 4>(tb, (size_t)id, f * VEC, vr[u]);
        } else {
          xs[u] = 0.f;
#pragma unroll
          for (int v = 0; v < VEC; v++) vr[u][v] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int v = 0; v < VEC; v++) {
          const float d = vr[u][v] * xs[u];
          sum[v] += d;
          sq = fmaf(d, d, sq);
        }
    }
  }
}

// pass 2 of a row: fm_sgd.h:38-50 with sum_f (complete) and the multiplier given.
//   ATOMIC: scatter-add of the delta (fp32 atomics execute at L2); else read-modify-write store.
template <int KP, int U, bool ATOMIC>
__device__ __forceinline__ void row_apply(const Entry* __restrict__ ent, uint32_t size,
                                          const Tab tb, const Hyper& h,
                                          const float (&sum)[Map<KP>::VEC], float mult) {
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR, EPI = Map<KP>::EPI;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t g = lane / LPR, f = lane % LPR;
  for (uint32_t base = 0; base < size; base += 64) {
    const uint32_t cnt = min(64u, size - base);
    Entry e; e.id = 0; e.value = 0.f;
    if (lane < cnt) {
      e = load_stream8(ent + base + lane);
      if (h.k1) {                                           // fm_sgd.h:38-43
        const float wv = tb.w[(size_t)e.id * tb.ws];
        const float dw = -h.lr * (mult * e.value + h.regw * wv);
        if (ATOMIC) unsafeAtomicAdd(tb.w + (size_t)e.id * tb.ws, dw); else tb.w[(size_t)e.id * tb.ws] = wv + dw;
      }
    }
    for (uint32_t i = 0; i < cnt; i += EPI * U) {
      float vr[U][VEC]; float xs[U]; uint32_t ids[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t idx = i + u * EPI + g;
        ids[u] = bcast_u32<EPI>(e.id, idx & 63u);
        xs[u] = bcast_f32<EPI>(e.value, idx & 63u);
        if (idx < cnt) row_ld<VEC, 8>(tb, (size_t)ids[u], f * VEC, vr[u]);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t idx = i + u * EPI + g;
        if (idx < cnt && f * VEC < tb.rs) {
          float* p = tb.V + (size_t)ids[u] * tb.rs + f * VEC;
          const float x = xs[u];
          float nv[VEC];
#pragma unroll
          for (int v = 0; v < VEC; v++) {                   // fm_sgd.h:44-50
            const float vv = vr[u][v];
            const float grad = sum[v] * x - vv * x * x;
            const float dv = -h.lr * (mult * grad + h.regv * vv);
            if (ATOMIC) unsafeAtomicAdd(p + v, dv); else nv[v] = vv + dv;
          }
          if (!ATOMIC) store_row<VEC, 8>(p, nv);
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// k_rowsums: one wavefront per example.  Writes
//   S[e][0..KP)           (WRITE_S)  the (partial) factor sums
//   scal[e]               FINISH ? rest_e = lin + 0.5*(sum_f S_ef^2 - Q_e)   (single device)
//                                : c_e    = lin - 0.5*Q_e                    (feature shard; all-reduced later)
// Used by predict / evaluate (WRITE_S = false, FINISH = true) and by the minibatch step.
// ----------------------------------------------------------------------------------------------
template <int KP, bool WRITE_S, bool FINISH>
__global__ void __launch_bounds__(256)
k_rowsums(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, uint64_t row0, uint32_t n_rows,
          const Tab tb, int k1,
          float* __restrict__ S, float* __restrict__ scal, const float* __restrict__ wside) {
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t e = wave0; e < n_rows; e += nwaves) {
    const uint64_t a = row_ptr[row0 + e];
    const uint32_t size = (uint32_t)(row_ptr[row0 + e + 1] - a);
    float sum[VEC], sq, lin;
    row_sums<KP, 8>(ent + a, size, tb, k1, sum, sq, lin, wside ? wside + a : nullptr);
#pragma unroll
    for (int v = 0; v < VEC; v++) sum[v] = subgroup_allsum<LPR>(sum[v]);
    if (WRITE_S && lane < LPR) store_vec<VEC>(S + (size_t)e * KP + lane * VEC, sum);
    float part = lin - 0.5f * sq;
    if (FINISH && lane < LPR) {
#pragma unroll
      for (int v = 0; v < VEC; v++) part = fmaf(0.5f * sum[v], sum[v], part);
    }
    part = wave_sum_dpp(part);
    if (lane == 0) scal[e] = part;
  }
}

// ----------------------------------------------------------------------------------------------
// SHORT rows (round 5): several examples per wavefront.  A feature shard of P GPUs sees nnz / P entries per example -- 4 at P = 8 -- and a
// wavefront that gathers 4 rows, reduces, stores and retires keeps 4 x 256 B in flight where the one-pass kernel of a full row keeps 32:
// k_rowsums ran at 3.2 TB/s on a P = 8 shard (profiles/r04_shard_probe_p8_kernel_stats.csv).  Here a wavefront takes G CONSECUTIVE examples
// (G ~ 24 / mean row length): their entries are one contiguous run of the CSR, gathered 32 rows at a time into registers exactly like
// k_fused does for one long row; lane l holds entry l and the example it belongs to, the factor sums are accumulated entry by entry (ids,
// values, example indices broadcast by v_readlane) and an example is flushed -- S row, scalar -- when the next one begins.  Empty examples
// flush zeros.  Needs one row per wave-wide load (KP >= 64).
// ----------------------------------------------------------------------------------------------
constexpr int MULTI_ZR = 32;                                           // row slots per round
constexpr int MULTI_GMAX = 16;                                         // examples per wavefront, at most
template <int KP, bool WRITE_S, bool FINISH>
__global__ void __launch_bounds__(256)
k_rowsums_multi(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, uint64_t row0, uint32_t n_rows,
                const Tab tb, int k1, float* __restrict__ S, float* __restrict__ scal, uint32_t G) {
  constexpr int VEC = Map<KP>::VEC;
  static_assert(Map<KP>::EPI == 1, "one row per wave-wide load");
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t e0 = wave0 * G; e0 < n_rows; e0 += nwaves * G) {
    const uint32_t ge = min(G, n_rows - e0);                          // examples of this group
    const uint64_t rp = row_ptr[row0 + e0 + min(lane, ge)];            // lane j <= ge: where example e0 + j begins (lane ge: where the group ends)
    const uint64_t a0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rp >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rp);
    const uint32_t rel = (uint32_t)(rp - a0);                          // ... relative to the group's first entry
    const uint32_t total = bcast_u32<1>(rel, ge);
    float sum[VEC]; float sq = 0.f, lin = 0.f;
#pragma unroll
    for (int v = 0; v < VEC; v++) sum[v] = 0.f;
    uint32_t cur = 0;                                                  // the example being accumulated
    auto flush = [&]() {
      const size_t e = (size_t)e0 + cur;
      if (WRITE_S) store_vec<VEC>(S + e * KP + lane * VEC, sum);
      float part = -0.5f * sq;
      if (FINISH) {
#pragma unroll
        for (int v = 0; v < VEC; v++) part = fmaf(0.5f * sum[v], sum[v], part);
      }
      part = wave_sum_dpp(part);
      if (lane == 0) scal[e] = part + lin;
#pragma unroll
      for (int v = 0; v < VEC; v++) sum[v] = 0.f;
      sq = 0.f; lin = 0.f; cur++;
    };
    for (uint32_t base = 0; base < total; base += MULTI_ZR) {
      const uint32_t cnt = min((uint32_t)MULTI_ZR, total - base);
      Entry en; en.id = 0; en.value = 0.f;
      float wl = 0.f;
      if (lane < cnt) {
        en = load_stream8(ent + a0 + base + lane);
        if (k1) wl = load_w(tb.w + (size_t)en.id * tb.ws) * en.value;
      }
      uint32_t ex = 0;                                                 // which example entry base + lane belongs to: the begins at or before it
      for (uint32_t j = 1; j < ge; j++) ex += (base + lane >= bcast_u32<1>(rel, j)) ? 1u : 0u;
      float vr[MULTI_ZR][VEC];
#pragma unroll
      for (int t = 0; t < MULTI_ZR; t++) {
        const uint32_t id = bcast_u32<1>(en.id, t);
        if ((uint32_t)t < cnt) row_ld<VEC, 4>(tb, (size_t)id, lane * VEC, vr[t]);
      }
#pragma unroll
      for (int t = 0; t < MULTI_ZR; t++) {
        const uint32_t ex_t = bcast_u32<1>(ex, t);
        const float x = bcast_f32<1>(en.value, t), wx = bcast_f32<1>(wl, t);
        if ((uint32_t)t < cnt) {
          while (cur < ex_t) flush();                                  // (wave-uniform) the examples that ended before this entry, empty ones included
          lin += wx;
#pragma unroll
          for (int v = 0; v < VEC; v++) { const float d = vr[t][v] * x; sum[v] += d; sq = fmaf(d, d, sq); }
        }
      }
    }
    while (cur < ge) flush();
  }
}

// The UPDATE of a shard's short rows, example-major: the second half of the step for the features that occur ONCE in their batch (the others
// are k_apply_seg's, like in k_fused<FUSED_APPLY>).  The dense owner pass (k_apply_seg over ALL segments) re-reads the 256-byte S_e per ENTRY
// -- feature order scatters an example's entries -- which is a third of its traffic at 4 entries per example; here S_e is read once per
// example: the wavefront stages the reduced sums of its G examples in LDS (a per-lane column store: lane f only ever reads what lane f
// wrote, no barrier) and every entry picks its example's row by a wave-uniform index.
// FUSED (bias-lag schedule): rest_e = c_e + 1/2 sum_f S_ef^2 and the multiplier are computed HERE from the reduced buffer -- the launches of
// k_rest_from_partial and k_mult and their round trips through rest[] / mult[] drop out; rest[] and mult[] are still written (the bias
// recurrence reads rest[], the deferred-feature pass reads mult[]).
template <int KP, bool FUSED>
__global__ void __launch_bounds__(256)
k_apply_multi(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, const float* __restrict__ target, uint64_t row0, uint32_t n_rows,
              const Tab tb, Hyper h, const double* __restrict__ w0_ptr, const float* __restrict__ S, const float* __restrict__ cpart,
              float* __restrict__ rest_out, float* __restrict__ mult, const uint64_t* __restrict__ cmask, uint32_t G) {
  constexpr int VEC = Map<KP>::VEC;
  static_assert(Map<KP>::EPI == 1, "one row per wave-wide load");
  __shared__ float s_S[4][MULTI_GMAX][KP];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  float (*sS)[KP] = s_S[wib];
  const float w0s = (FUSED && h.k0) ? (float)(*w0_ptr) : 0.f;
  for (uint32_t e0 = wave0 * G; e0 < n_rows; e0 += nwaves * G) {
    const uint32_t ge = min(G, n_rows - e0);
    const uint64_t rp = row_ptr[row0 + e0 + min(lane, ge)];
    const uint64_t a0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rp >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rp);
    const uint32_t rel = (uint32_t)(rp - a0);
    const uint32_t total = bcast_u32<1>(rel, ge);
    // everything that depends only on the row offsets is asked for TOGETHER -- the examples' reduced sums, their scalars, the first round's
    // entries and masks: one round trip, not one per kind (the stores of rest[] / mult[] below would otherwise hold the entry loads back)
    auto entry_place = [&](uint32_t base, uint32_t& ex, uint32_t& st) {   // which example entry base + lane belongs to, and where that one begins
      ex = 0; st = 0;
      for (uint32_t j = 1; j < ge; j++) {
        const uint32_t rj = bcast_u32<1>(rel, j);
        if (base + lane >= rj) { ex++; st = rj; }
      }
    };
    uint32_t ex0, st0;
    entry_place(0, ex0, st0);
    Entry en0; en0.id = 0; en0.value = 0.f;
    uint64_t cm0 = 0;
    if (lane < min((uint32_t)MULTI_ZR, total)) { en0 = load_stream8(ent + a0 + lane); cm0 = cmask[row0 + e0 + ex0]; }
    // the examples' reduced sums: lane f keeps column f of every example in LDS; multipliers: lane j holds example j's
    float sv[MULTI_GMAX][VEC];
#pragma unroll
    for (int j = 0; j < MULTI_GMAX; j++)
      if ((uint32_t)j < ge) load_vec<VEC>(S + ((size_t)e0 + j) * KP + lane * VEC, sv[j]);
    float mreg = 0.f, creg = 0.f, yreg = 0.f, rreg = 0.f;
    if (lane < ge) {
      if (FUSED) { creg = cpart[e0 + lane]; yreg = target[row0 + e0 + lane]; }
      else mreg = mult[e0 + lane];
    }
#pragma unroll
    for (int j = 0; j < MULTI_GMAX; j++) {
      if ((uint32_t)j < ge) {
#pragma unroll
        for (int v = 0; v < VEC; v++) sS[j][lane * VEC + v] = sv[j][v];
        if (FUSED) {
          float part = 0.f;
#pragma unroll
          for (int v = 0; v < VEC; v++) part = fmaf(0.5f * sv[j][v], sv[j][v], part);
          const float rest = wave_sum_dpp(part) + bcast_f32<1>(creg, j);
          const float m = multiplier(h, w0s + rest, bcast_f32<1>(yreg, j));
          if (lane == (uint32_t)j) { mreg = m; rreg = rest; }
        }
      }
    }
    if (FUSED && lane < ge) { rest_out[e0 + lane] = rreg; mult[e0 + lane] = mreg; }
    for (uint32_t base = 0; base < total; base += MULTI_ZR) {
      const uint32_t cnt = min((uint32_t)MULTI_ZR, total - base);
      Entry en = en0;
      uint32_t ex = ex0, st = st0;
      uint64_t cm = cm0;
      if (base) {                                                      // (a group beyond one round: rare by construction of G)
        entry_place(base, ex, st);
        en.id = 0; en.value = 0.f; cm = 0;
        if (lane < cnt) { en = load_stream8(ent + a0 + base + lane); cm = cmask[row0 + e0 + ex]; }
      }
      bool def = false;                                                // the entry's feature occurs more than once in the batch: k_apply_seg's
      if (lane < cnt) {
        const uint32_t pos = base + lane - st;
        def = pos >= 64u || ((cm >> pos) & 1ull);
      }
      const uint64_t defm = __ballot(def);
      const float ml = __shfl(mreg, (int)ex);                          // (all lanes: the source lane is the example's)
      // the linear weight is ASKED FOR before the rows and WRITTEN after they have been asked for: a store to w[] between the two would
      // hold the row loads back (w and V may alias for the compiler) -- one dependent round trip more per round
      const bool upd_w = h.k1 && lane < cnt && !def;
      float* pw = tb.w + (size_t)en.id * tb.ws;
      float wv = 0.f;
      if (upd_w) wv = load_w(pw);
      float vr[MULTI_ZR][VEC];
#pragma unroll
      for (int t = 0; t < MULTI_ZR; t++) {
        const uint32_t id = bcast_u32<1>(en.id, t);
        if ((uint32_t)t < cnt && !((defm >> t) & 1ull)) row_ld<VEC, 8>(tb, (size_t)id, lane * VEC, vr[t]);
      }
      if (upd_w) *pw = wv - h.lr * (ml * en.value + h.regw * wv);      // fm_sgd.h:38-43
#pragma unroll
      for (int t = 0; t < MULTI_ZR; t++) {
        const uint32_t id = bcast_u32<1>(en.id, t);
        const uint32_t ex_t = bcast_u32<1>(ex, t);
        const float x = bcast_f32<1>(en.value, t);
        const float m = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(mreg), (int)ex_t));
        if ((uint32_t)t < cnt && !((defm >> t) & 1ull) && lane * VEC < tb.rs) {
          float* pv = tb.V + (size_t)id * tb.rs + lane * VEC;
          float nv[VEC];
#pragma unroll
          for (int v = 0; v < VEC; v++) {                               // fm_sgd.h:44-50
            const float vv = vr[t][v];
            const float grad = sS[ex_t][lane * VEC + v] * x - vv * x * x;
            nv[v] = vv - h.lr * (m * grad + h.regv * vv);
          }
          store_row<VEC, 8>(pv, nv);
        }
      }
    }
  }
}

// after the all-reduce of a feature-sharded partial buffer: rest_e = c_e + 0.5 * sum_f S_ef^2
template <int KP>
__global__ void __launch_bounds__(256)
k_rest_from_partial(const float* __restrict__ S, const float* __restrict__ c, uint32_t n_rows, float* __restrict__ rest) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  if constexpr (KP >= 4) {
    // S is a dense [n_rows][KP] stream: 16-byte loads, KP/4 lanes per example, 4 loads in flight per lane
    constexpr uint32_t L4 = (KP / 4 < 64) ? KP / 4 : 64;          // lanes that share one example
    constexpr uint32_t R = KP / 4 / L4;                           // float4 per lane and example (KP > 256: not reached)
    constexpr uint32_t EPW = 64 / L4;                             // examples per wave-wide load
    constexpr uint32_t U = 4;
    const uint32_t g = lane / L4, f = lane % L4;
    for (uint32_t e0 = wave0 * EPW * U; e0 < n_rows; e0 += nwaves * EPW * U) {
      float4 t[U][R];
#pragma unroll
      for (uint32_t u = 0; u < U; u++) {
        const uint32_t e = e0 + u * EPW + g;
#pragma unroll
        for (uint32_t r = 0; r < R; r++)
          t[u][r] = (e < n_rows) ? reinterpret_cast<const float4*>(S + (size_t)e * KP)[f * R + r] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (uint32_t u = 0; u < U; u++) {
        const uint32_t e = e0 + u * EPW + g;
        float part = 0.f;
#pragma unroll
        for (uint32_t r = 0; r < R; r++) {
          part = fmaf(0.5f * t[u][r].x, t[u][r].x, part); part = fmaf(0.5f * t[u][r].y, t[u][r].y, part);
          part = fmaf(0.5f * t[u][r].z, t[u][r].z, part); part = fmaf(0.5f * t[u][r].w, t[u][r].w, part);
        }
#pragma unroll
        for (uint32_t o = 1; o < L4; o <<= 1) part += __shfl_xor(part, o);
        if (e < n_rows && f == 0) rest[e] = part + c[e];
      }
    }
  } else {
    for (uint32_t e = wave0 * 64 + lane; e < n_rows; e += nwaves * 64) {
      float part = 0.f;
#pragma unroll
      for (int v = 0; v < KP; v++) { const float x = S[(size_t)e * KP + v]; part = fmaf(0.5f * x, x, part); }
      rest[e] = part + c[e];
    }
  }
}

// ----------------------------------------------------------------------------------------------
// k_scan: step 2 of the minibatch rule (oracle/fm_oracle.h): w0 advances in micro-chunks of `chunk`
// consecutive examples; mult_e is computed with the w0 of its chunk.
//   fm_learn_sgd_element.h:57-65 (p, multiplier) + fm_sgd.h:34-37 (w0) per example.
// ONE workgroup: 1024 threads stage rest/target tiles through LDS (the recurrence itself is serial, so
// its loads must not be HBM round trips); wavefront 0 runs the recurrence out of LDS with a DPP reduction
// per chunk; the multipliers go back to HBM in coalesced tiles.
// ----------------------------------------------------------------------------------------------
constexpr int SCAN_TILE = 4096;
// ONE wavefront, <= 32 VGPRs, 64 KiB LDS: it must be placeable on a CU whose SIMDs are already full of gather
// wavefronts (the hogwild launch it overlaps leaves 32 free VGPRs per lane), otherwise it only starts when that
// launch drains and the next one stalls behind it (measured: +35 % per launch with a 4-wavefront, 72-VGPR version).
// Tiles arrive by LDS-DMA (global_load_lds: no VGPR round trip), tile t+1 in flight while tile t is scanned.
__device__ __forceinline__ void scan_fetch_array(const float* __restrict__ g, uint32_t cnt, float* s, uint32_t lane) {
  if ((((uintptr_t)g) & 15u) == 0) {                           // 16-byte pieces: 1 KiB per wave instruction
    const uint32_t full = cnt & ~3u;                           // whole 16-byte groups only: never read past g[cnt - 1]
    for (uint32_t base = 0; base < full; base += 256)
      if (base + 4 * lane < full) __builtin_amdgcn_global_load_lds(g + base + 4 * lane, s + base, 16, 0, 0);
    if (full + lane < cnt) __builtin_amdgcn_global_load_lds(g + full + lane, s + full, 4, 0, 0);   // ragged tail (< 4 floats)
  } else {                                                     // unaligned start (row0 not a multiple of 4): dwords
    for (uint32_t base = 0; base < cnt; base += 64)
      if (base + lane < cnt) __builtin_amdgcn_global_load_lds(g + base + lane, s + base, 4, 0, 0);
  }
}
__device__ __forceinline__ void scan_fetch_tile(const float* __restrict__ g_rest, const float* __restrict__ g_y,
                                                uint32_t cnt, float* s_r, float* s_t, uint32_t lane) {
  scan_fetch_array(g_rest, cnt, s_r, lane);
  scan_fetch_array(g_y, cnt, s_t, lane);
}
template <bool WRITE_MULT, int TASK>
__global__ void __launch_bounds__(64)
k_scan(const float* __restrict__ rest, const float* __restrict__ target, uint32_t n_rows, uint32_t chunk,
       Hyper h, const double* __restrict__ w0_in, double* __restrict__ w0_out, float* __restrict__ mult, const Handoff hw) {
  __shared__ float s_rest[2][SCAN_TILE];
  __shared__ float s_y[2][SCAN_TILE];
  if (!handoff_wait_counter(hw)) { handoff_pass_on(w0_in, w0_out); return; }
  // as the youngest wavefront on its SIMD it would only get leftover issue slots: raise the priority
  __builtin_amdgcn_s_setprio(3);
  const uint32_t lane = threadIdx.x;
  const uint32_t n_tiles = (n_rows + SCAN_TILE - 1) / SCAN_TILE;
  double w0 = *w0_in;
  uint32_t chunk_pos = 0;
  float acc = 0.f;
  scan_fetch_tile(rest, target, min((uint32_t)SCAN_TILE, n_rows), s_rest[0], s_y[0], lane);
  for (uint32_t t = 0; t < n_tiles; t++) {
    const uint32_t t0 = t * SCAN_TILE;
    const uint32_t tn = min((uint32_t)SCAN_TILE, n_rows - t0);
    const uint32_t cur = t & 1u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // tile t has landed in LDS (issued one tile ago)
    __builtin_amdgcn_s_barrier();                              // single wavefront: orders the DMA writes before the reads
    if (t + 1 < n_tiles) {
      const uint32_t u0 = t0 + SCAN_TILE;
      scan_fetch_tile(rest + u0, target + u0, min((uint32_t)SCAN_TILE, n_rows - u0), s_rest[cur ^ 1u], s_y[cur ^ 1u], lane);
    }
    const float* sr = s_rest[cur];
    const float* sy = s_y[cur];
    uint32_t i = 0;
    while (i < tn) {
      const uint32_t take = min(chunk - chunk_pos, tn - i);
      const float w0s = h.k0 ? (float)w0 : 0.f;
      uint32_t tt = 0;
      for (; tt + 256 <= take; tt += 256) {                   // 4 independent elements per lane: reads first, then math
        const uint32_t q = i + tt + lane;                     // (kept to 32 VGPRs: the wavefront must fit next to a
        float m0 = sr[q], m1 = sr[q + 64], m2 = sr[q + 128], m3 = sr[q + 192];   // chip-filling gather)
        const float y0 = sy[q], y1 = sy[q + 64], y2 = sy[q + 128], y3 = sy[q + 192];
        m0 = multiplier_task<TASK>(h, w0s + m0, y0);
        m1 = multiplier_task<TASK>(h, w0s + m1, y1);
        m2 = multiplier_task<TASK>(h, w0s + m2, y2);
        m3 = multiplier_task<TASK>(h, w0s + m3, y3);
        if (WRITE_MULT) { mult[t0 + q] = m0; mult[t0 + q + 64] = m1; mult[t0 + q + 128] = m2; mult[t0 + q + 192] = m3; }
        acc += (m0 + m1) + (m2 + m3);
      }
      for (tt += lane; tt < take; tt += 64) {
        const float m = multiplier_task<TASK>(h, w0s + sr[i + tt], sy[i + tt]);
        if (WRITE_MULT) mult[t0 + i + tt] = m;
        acc += m;
      }
      i += take; chunk_pos += take;
      if (chunk_pos == chunk || t0 + i == n_rows) {
        const float tot = wave_sum_dpp(acc);
        if (h.k0) w0 -= (double)h.lr * ((double)tot + (double)chunk_pos * (double)h.reg0 * (double)w0s);
        acc = 0.f; chunk_pos = 0;
      }
    }
  }
  if (lane == 0) { if (hw.ctr) handoff_publish(w0_out, w0); else *w0_out = w0; }
}

// tile pipeline of k_scan1: four buffers of the 160 KiB LDS, tiles t+1 .. t+3 in flight while t is scanned.  Under a chip-filling
// gather an LDS-DMA tile takes ~10-15 us to land while 4096 examples are scanned in a few, so one tile of prefetch left the recurrence
// DMA-latency-bound (its kernel time tracked the gather's, ~1 ms per 262 144 examples against 0.3 ms alone).
constexpr int SCAN4_BUFS = 4;
constexpr size_t SCAN4_LDS_BYTES = (size_t)(SCAN4_BUFS * 2 * SCAN_TILE + 8) * sizeof(float);   // 128 KiB (+ 32 spare bytes)
// k_scan1: the recurrence for micro-chunks that are multiples of 256 examples.  Four wavefronts fetch a quarter of every tile by LDS-DMA
// (three tiles in flight); ONE wavefront evaluates the chain -- a piece of 256 examples is four CONTIGUOUS examples per lane (one
// ds_read_b128 per array), so nothing crosses wavefronts.  (Round 2's k_scan4 spread a piece over the four wavefronts: its LDS exchange +
// barrier per piece cost more than the three quarters of the transcendentals they took off the chain -- 329 vs 174 ns per micro-chunk of 256,
// profiles/r03_scan_chain.txt; removed.)  What does not depend on the
// bias is taken off the chain: the operands of piece p+1 are read while piece p is reduced, and for classification the multiplier
//   -y (1 - 1/(1 + e^(-y p))) = -y / (1 + e^(y p)) = -y / (1 + 2^(a w0 + b)),  a = y log2(e), b = a rest   (fm_learn_sgd_element.h:61-65)
// leaves fma -> v_exp -> add -> v_rcp -> fma on the chain, a and b prepared ahead.  Summation order: the four examples of a lane
// pairwise, the lanes by DPP, pieces of a chunk in order (deterministic; not bit-identical to k_scan / k_scan4, which no result is
// compared with bit by bit across kernels).  60 VGPRs: fits next to five 85-VGPR gather wavefronts per SIMD.
// CH: 256 = the micro-chunk IS one piece (every piece ends one); 0 = any multiple of 256; 16 / 32 / 64 / 128 = the SUB-PIECE form (round 5):
// the reference moves w0 after every example (fm_sgd.h:34-37) and how finely the recurrence follows that path is what sets the batch rule's
// distance from the online loop (micro-chunk 256 ends 0.025 from the online bias, 32 ends 0.0014), so the default
// micro-chunk is now below one piece.  A piece is then read as four 64-example vectors (element j of lane l = example 64 j + l: conflict-free
// dword reads, coalesced multiplier stores) and a micro-chunk is one DPP row (16), half a wavefront (32), a vector (64) or two (128): per
// micro-chunk the chain is fma -> v_exp -> add -> v_rcp -> mul -> 4-6 DPP adds -> v_readlane -> two fp64 fma -> cvt, ~60 ns.
template <bool WRITE_MULT, int TASK, int CH>
__global__ void __launch_bounds__(256)
k_scan1(const float* __restrict__ rest, const float* __restrict__ target, uint32_t n_rows, uint32_t chunk,
        Hyper h, const double* __restrict__ w0_in, double* __restrict__ w0_out, float* __restrict__ mult, const Handoff hw) {
  extern __shared__ float scan_lds[];
  if (!handoff_wait_counter(hw)) { handoff_pass_on(w0_in, w0_out); return; }   // (device hand-off: resident and polling until its batch's rest[] is complete)
  __builtin_amdgcn_s_setprio(3);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr uint32_t QT = SCAN_TILE / 4;
  constexpr uint32_t D = SCAN4_BUFS - 1;
  const uint32_t n_tiles = (n_rows + SCAN_TILE - 1) / SCAN_TILE;
  const bool aligned = ((((uintptr_t)rest) | ((uintptr_t)target)) & 15u) == 0;
  const bool mult16 = WRITE_MULT && ((((uintptr_t)mult) & 15u) == 0);
  double w0 = *w0_in;                                            // (launched only for models with a bias: h.k0)
  float w0s = (float)w0;
  const double reg0_d = (double)h.reg0, neg_lr_d = -(double)h.lr;
  float chunk_acc = 0.f;
  uint32_t chunk_pos = 0;
  auto fetch = [&](uint32_t t) {
    const uint32_t buf = t % SCAN4_BUFS;
    const uint32_t t0 = t * SCAN_TILE;
    const uint32_t tn = min((uint32_t)SCAN_TILE, n_rows - t0);
    const uint32_t q0 = wv * QT;
    float* sr = scan_lds + buf * 2 * SCAN_TILE;
    if (q0 < tn) scan_fetch_tile(rest + t0 + q0, target + t0 + q0, min(QT, tn - q0), sr + q0, sr + SCAN_TILE + q0, lane);
  };
  for (uint32_t t = 0; t < D && t < n_tiles; t++) fetch(t);
  constexpr float LOG2E = 1.4426950408889634f;
  for (uint32_t t = 0; t < n_tiles; t++) {
    const uint32_t t0 = t * SCAN_TILE;
    const uint32_t tn = min((uint32_t)SCAN_TILE, n_rows - t0);
    const uint32_t last_issued = min(t + D - 1, n_tiles - 1);
    const uint32_t later = last_issued - t;
    const bool counted = !WRITE_MULT && aligned && ((uint64_t)(last_issued + 1) * SCAN_TILE <= n_rows);
    if (counted && later == 2)      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (counted && later == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                             // tile t has landed for everybody; tile t-1 is no longer read
    if (t + D < n_tiles) fetch(t + D);
    if (wv != 0) continue;                                       // (the other wavefronts only fetch)
    const float* sr = scan_lds + (t % SCAN4_BUFS) * 2 * SCAN_TILE;
    const float* sy = sr + SCAN_TILE;
    // One piece = 256 examples, four contiguous ones per lane.  The chain is bound by the NUMBER of instructions one wavefront issues per
    // piece (scripts/ubench/scan_chain.hip: 300 ns per piece for classification AND for regression before this was counted), so a full
    // piece carries no bounds masks, the loop is unrolled (no copies between the operand sets of consecutive pieces) and everything
    // that does not depend on the bias (LDS reads, a = y log2 e, b = a rest) is free to move ahead of the previous piece's reduction.
    float4 rN = *reinterpret_cast<const float4*>(sr + 4 * lane), yN = *reinterpret_cast<const float4*>(sy + 4 * lane);
    // tnc: the tile's length as the piece sees it -- the literal SCAN_TILE on the unrolled path of a full tile, so that "is there a next
    // piece" and "is this piece full" fold away and the 16 pieces are ONE basic block the scheduler can move the LDS reads through
    auto piece = [&](uint32_t c0, uint32_t tnc, auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      const uint32_t q = c0 + 4 * lane;
      const float4 r4 = rN, y4 = yN;
      if (c0 + 256 < tnc) {                                      // the next piece's operands leave LDS while this one is evaluated
        rN = *reinterpret_cast<const float4*>(sr + q + 256);
        yN = *reinterpret_cast<const float4*>(sy + q + 256);
      }
      float rr[4] = {r4.x, r4.y, r4.z, r4.w}, yy[4] = {y4.x, y4.y, y4.z, y4.w};
      if constexpr (!FULL) {
#pragma unroll
        for (int j = 0; j < 4; j++) { const bool ok = q + j < tnc; rr[j] = ok ? rr[j] : 0.f; yy[j] = ok ? yy[j] : 0.f; }
      }
      float m[4];
      if constexpr (TASK == 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float a = LOG2E * yy[j];                          // (y = 0 past the end: a = b = 0, multiplier -0 * 1/2 = 0)
          const float u = __builtin_amdgcn_exp2f(fmaf(a, w0s, a * rr[j]));
          m[j] = -yy[j] * __builtin_amdgcn_rcpf(1.0f + u);
        }
      } else {
        const float gs = h.sgda ? 2.0f : 1.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float pc = fmaxf(h.min_target, fminf(h.max_target, w0s + rr[j]));
          m[j] = gs * (pc - yy[j]);
          if constexpr (!FULL) m[j] = (q + j < tnc) ? m[j] : 0.f;
        }
      }
      if constexpr (WRITE_MULT) {
        if (mult16 && (FULL || q + 4 <= tnc)) *reinterpret_cast<float4*>(mult + t0 + q) = make_float4(m[0], m[1], m[2], m[3]);
        else {
#pragma unroll
          for (int j = 0; j < 4; j++) if (FULL || q + j < tnc) mult[t0 + q + j] = m[j];
        }
      }
      const float tot = wave_sum_dpp((m[0] + m[1]) + (m[2] + m[3]));
      const uint32_t n_here = FULL ? 256u : min(256u, tnc - c0);
      chunk_acc += tot;
      chunk_pos += n_here;
      if ((CH == 256 && FULL) || chunk_pos == chunk || t0 + c0 + n_here == n_rows) {
        double term = (double)chunk_acc;
        term = fma((double)chunk_pos * reg0_d, (double)w0s, term);
        w0 = fma(neg_lr_d, term, w0);
        w0s = (float)w0;
        chunk_acc = 0.f; chunk_pos = 0;
      }
    };
    if constexpr (CH == 0 || CH == 256) {
    if (tn == SCAN_TILE) {
#pragma unroll
      for (uint32_t c0 = 0; c0 < SCAN_TILE; c0 += 256) piece(c0, (uint32_t)SCAN_TILE, std::true_type());
    } else {
      for (uint32_t c0 = 0; c0 < tn; c0 += 256) {
        if (c0 + 256 <= tn) piece(c0, tn, std::true_type()); else piece(c0, tn, std::false_type());
      }
    }
    } else {
      // ---- sub-piece form: micro-chunks of CH = 16 / 32 / 64 / 128 examples ------------------------------------------------
      constexpr int CL = (CH >= 64) ? 64 : CH;                   // lanes of one micro-chunk
      constexpr int NS = 64 / CL;                                // micro-chunks per 64-example vector
      constexpr int EPC = (CH >= 64) ? CH / 64 : 1;              // vectors per micro-chunk (CH = 128: two)
      float rS[4], yS[4];                                        // the operands of the NEXT piece (read off the chain)
#pragma unroll
      for (int j = 0; j < 4; j++) { rS[j] = sr[64 * j + lane]; yS[j] = sy[64 * j + lane]; }
      auto piece_sub = [&](uint32_t c0, uint32_t tnc, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        float rr[4], yy[4]; bool ok[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          ok[j] = FULL || (c0 + 64u * (uint32_t)j + lane < tnc);
          rr[j] = ok[j] ? rS[j] : 0.f; yy[j] = ok[j] ? yS[j] : 0.f;    // (past the batch: stale tile contents, never NaN into the sums)
        }
        if (c0 + 256 < tnc) {                                    // (c0 + 511 < SCAN_TILE: inside the tile buffer)
#pragma unroll
          for (int j = 0; j < 4; j++) { rS[j] = sr[c0 + 256 + 64 * j + lane]; yS[j] = sy[c0 + 256 + 64 * j + lane]; }
        }
        float aa[4], bb[4];
        if constexpr (TASK == 1) {
#pragma unroll
          for (int j = 0; j < 4; j++) { aa[j] = LOG2E * yy[j]; bb[j] = aa[j] * rr[j]; }
        }
        const float gs = h.sgda ? 2.0f : 1.0f;
#pragma unroll
        for (int j0 = 0; j0 < 4; j0 += EPC) {
#pragma unroll
          for (int sb = 0; sb < NS; sb++) {
            float m[EPC]; float msum = 0.f;
#pragma unroll
            for (int q = 0; q < EPC; q++) {
              const int j = j0 + q;
              if constexpr (TASK == 1) {
                m[q] = -yy[j] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(aa[j], w0s, bb[j])));   // (y = 0 past the end: -0)
              } else {
                const float pc = fmaxf(h.min_target, fminf(h.max_target, w0s + rr[j]));
                m[q] = ok[j] ? gs * (pc - yy[j]) : 0.f;
              }
              msum += m[q];
            }
            if constexpr (WRITE_MULT) {
#pragma unroll
              for (int q = 0; q < EPC; q++)
                if (ok[j0 + q] && (NS == 1 || lane / (uint32_t)CL == (uint32_t)sb)) mult[t0 + c0 + 64u * (uint32_t)(j0 + q) + lane] = m[q];
            }
            const float tot = group_sum_dpp<CL>(msum, (uint32_t)sb);
            const uint32_t lo = c0 + 64u * (uint32_t)j0 + (uint32_t)(sb * CL);       // first example of this micro-chunk inside the tile
            const uint32_t n_here = FULL ? (uint32_t)CH : (lo < tnc ? min((uint32_t)CH, tnc - lo) : 0u);
            double term = (double)tot;                             // (n_here = 0: every multiplier was masked, the bias stays)
            term = fma((double)n_here * reg0_d, (double)w0s, term);
            w0 = fma(neg_lr_d, term, w0);
            w0s = (float)w0;
          }
        }
      };
#pragma unroll 1
      for (uint32_t c0 = 0; c0 < tn; c0 += 256) {
        if (c0 + 256 <= tn) piece_sub(c0, tn, std::true_type()); else piece_sub(c0, tn, std::false_type());
      }
    }
  }
  if (threadIdx.x == 0) { if (hw.ctr) handoff_publish(w0_out, w0); else *w0_out = w0; }
}

// ----------------------------------------------------------------------------------------------
// k_scan_pit: the bias recurrence of a batch solved PARALLEL IN TIME (round 5).
// The recurrence  w_{c+1} = w_c - lr (sum_{i in chunk c} m_i(w_c) + n_c reg0 w_c)   (fm_sgd.h:34-37 summed per micro-chunk; m_i the loss
// multiplier of fm_learn_sgd_element.h:58-65 at p_i = w + rest_i) is a chain of B / chunk dependent steps: ~60 ns each on one wavefront,
// whatever the chip does -- 0.5 ms per 262 144 examples at micro-chunk 32, a ceiling of 0.5 G examples/s that no number of GPUs lifts.
// Newton's method on the WHOLE path removes the chain: with d_c the path (offset from the batch-start bias) of the previous iterate, every
// chunk's step is linearised around d_c,
//     d'_{c+1} = A_c d'_c + B_c,   A_c = 1 - lr D_c,   B_c = -lr (F_c - D_c d_c),   F_c = sum m_i + n_c reg0 (w + d_c),  D_c = sum m_i' + n_c reg0,
// an AFFINE recurrence -- composition of affine maps is associative, so the new path is a parallel prefix scan -- and the iteration converges
// quadratically (the linearised chain is a contraction: 0 < A_c <= 1 in the stable regime lr chunk curvature <= 1): path changes of
// 3e-1, 7e-4, 4e-7 on a 262 144-example batch from a constant first guess.  The converged path IS the serial recurrence to fp32 rounding (~1e-8
// on a bias of 0.05; scripts/cpu_pit_check.py holds the arithmetic against the serial loop) whatever the micro-chunk -- so the micro-chunk can
// be as small as the reference's own (1 example) at no cost.
// Layout: up to 32 workgroups of 256 threads, each keeps 8192 examples' {rest, y} in LDS (one DMA round trip) + its chunks' path; a wavefront
// owns 2048 contiguous examples (32 vectors of 64).  One Newton iteration = sweep 1 (every wavefront composes the maps of its examples at the
// current path into one), ONE grid-wide exchange of the workgroups' composites (a slot array + arrival counter in HBM, agent scope; bounded
// spin), sweep 2 (every wavefront replays its maps from its incoming value and stores the new path).  The exchange also carries the largest
// path change of the previous sweep 2, so all workgroups take the same decision to stop; then the composite applied to 0 is the bias change
// of the batch.  If Newton does not settle (PIT_MAX_IT; the serial recurrence itself is then oscillating) workgroup 0 evaluates the chain
// serially -- same result as k_scan.
// ----------------------------------------------------------------------------------------------
constexpr uint32_t PIT_SEG = 8192;                                    // examples per workgroup (4096 at micro-chunk 1: the chunk maps need the room)
constexpr uint32_t PIT_MAX_WG = 64;
constexpr uint32_t PIT_MAX_ROWS = 262144;                             // the default batch: 32 workgroups x 8192 (64 x 4096 at micro-chunk 1)
constexpr uint32_t PIT_MAX_CHUNK = 1024;                              // a micro-chunk never straddles two wavefronts
constexpr uint32_t PIT_MAX_IT = 12;
constexpr float    PIT_TOL = 5e-4f;                                   // the path moved less than this: the NEXT iterate is exact to ~tol^2
// LDS: {rest, y} of the segment + per chunk {path offset, map a, map b}: 8192 x 8 + 4096 x 12 (chunk >= 2) or 4096 x 8 + 4096 x 12 (chunk 1)
constexpr size_t   PIT_LDS_BYTES = (size_t)(2 * PIT_SEG + 3 * (PIT_SEG / 2)) * sizeof(float);
__host__ __device__ __forceinline__ uint32_t pit_seg(uint32_t chunk) { return chunk == 1u ? PIT_SEG / 2 : PIT_SEG; }
struct PitSync { unsigned long long* ctr; double* slots; uint32_t* err; uint32_t spins; };   // ctr[PIT_MAX_IT + 1] zeroed before the launch ([PIT_MAX_IT]: who took the serial
                                                                     // fall-back); slots[2][PIT_MAX_WG][4]; spins: bound of an exchange's wait (HANDOFF_SPINS; tests shorten it)
struct AMap { float a, b; };                                          // x -> a x + b
__device__ __forceinline__ AMap amap_after(const AMap first, const AMap then) { return AMap{then.a * first.a, fmaf(then.a, first.b, then.b)}; }
// max that does NOT drop a NaN (fmaxf does): a path that left the numbers must read as "not converged", never as "no change"
__device__ __forceinline__ float nanmax(float a, float b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ double nanmax(double a, double b) { return (a > b || a != a) ? a : b; }
#define FMX_DPP_F(old, x, ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), ctrl, rmask, 0xf, false))
// the sum over the lane's group of C consecutive lanes, in EVERY lane of the group (C = 1 .. 64)
template <int C> __device__ __forceinline__ float seg_allsum(float x, uint32_t lane) {
  if constexpr (C >= 2)  x += FMX_DPP_F(0.f, x, 0xB1, 0xf);            // quad_perm(1,0,3,2)
  if constexpr (C >= 4)  x += FMX_DPP_F(0.f, x, 0x4E, 0xf);            // quad_perm(2,3,0,1)
  if constexpr (C >= 8)  x += FMX_DPP_F(0.f, x, 0x141, 0xf);           // row_half_mirror: the other quad of the half row
  if constexpr (C >= 16) x += FMX_DPP_F(0.f, x, 0x140, 0xf);           // row_mirror: the other half of the row
  if constexpr (C >= 32) {
    const float s0 = bcast_f32<1>(x, 0), s1 = bcast_f32<1>(x, 16), s2 = bcast_f32<1>(x, 32), s3 = bcast_f32<1>(x, 48);
    if constexpr (C == 32) x = (lane < 32u) ? s0 + s1 : s2 + s3; else x = (s0 + s1) + (s2 + s3);
  }
  return x;
}
// inclusive scan of 64 affine maps across the lanes (lane order = order of application): the DPP prefix pattern (row_shr 1/2/4/8, then the
// row broadcasts); a lane without a source keeps its map (old = the identity)
__device__ __forceinline__ void amap_scan64(float& a, float& b) {
#define FMX_AMAP_STEP(ctrl, rmask) { const float pa = FMX_DPP_F(1.0f, a, ctrl, rmask), pb = FMX_DPP_F(0.0f, b, ctrl, rmask); b = fmaf(a, pb, b); a = a * pa; }
  FMX_AMAP_STEP(0x111, 0xf) FMX_AMAP_STEP(0x112, 0xf) FMX_AMAP_STEP(0x114, 0xf) FMX_AMAP_STEP(0x118, 0xf)
  FMX_AMAP_STEP(0x142, 0xa) FMX_AMAP_STEP(0x143, 0xc)
#undef FMX_AMAP_STEP
}

template <bool WRITE_MULT, int TASK>
__global__ void __launch_bounds__(256)
k_scan_pit(const float* __restrict__ rest, const float* __restrict__ target, uint32_t n_rows, uint32_t chunk,
           Hyper h, const double* __restrict__ w0_in, double* __restrict__ w0_out, float* __restrict__ mult, const Handoff hw, const PitSync ps) {
  extern __shared__ float pit_lds[];
  const uint32_t SEG = pit_seg(chunk), WSEG = SEG / 4u;               // examples per workgroup / per wavefront
  float* s_r = pit_lds;
  float* s_y = pit_lds + SEG;
  float* s_d = pit_lds + 2 * SEG;                                      // per chunk of the segment: path offset at its start,
  float* s_ma = s_d + PIT_SEG / 2;                                     // its affine map (after the scan: the composite of the wavefront's
  float* s_mb = s_ma + PIT_SEG / 2;                                    // chunks BEFORE it)
  __shared__ float s_map[4][2];                                       // the wavefronts' composite maps
  __shared__ float s_chg[4];
  __shared__ double s_bcast[4];                                       // [0] incoming value of the workgroup, [1] value after the last one, [2] largest change
  __shared__ uint32_t s_abort;                                        // an exchange ran into its bound (or another workgroup's did)
  if (!handoff_wait_counter(hw)) { handoff_pass_on(w0_in, w0_out); return; }   // (device hand-off: resident and polling until its batch's rest[] is complete)
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t g = blockIdx.x, nwg = gridDim.x;
  const uint32_t s0 = g * SEG;
  const uint32_t sn = (s0 < n_rows) ? min(SEG, n_rows - s0) : 0u;      // examples of this workgroup
  const uint32_t q0 = wv * WSEG;
  const double w0 = *w0_in;                                           // (launched only for models with a bias: h.k0)
  const float w0s = (float)w0;
  const uint32_t lc = 31u - (uint32_t)__builtin_clz(chunk);            // chunk = 1 << lc
  const uint32_t wch = WSEG >> lc, c_w0 = q0 >> lc;                    // chunks per wavefront, the wavefront's first chunk
  constexpr float LOG2E = 1.4426950408889634f;
  // the segment into LDS: one DMA round trip for everything this workgroup will ever read
  if (q0 < sn) scan_fetch_tile(rest + s0 + q0, target + s0 + q0, min(WSEG, sn - q0), s_r + q0, s_y + q0, lane);
  for (uint32_t c = threadIdx.x; c < (SEG >> lc); c += 256u) s_d[c] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // one example at the current path: multiplier and its derivative
  auto eval = [&](uint32_t i, float d, float& m, float& dm) {
    const bool ok = i < sn;
    const float r = ok ? s_r[i] : 0.f, y = ok ? s_y[i] : 0.f;
    const float p = (w0s + d) + r;
    if constexpr (TASK == 1) {
      const float inv = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(LOG2E * y * p));   // 1 / (1 + e^{y p})
      m = -y * inv;                                                   // fm_learn_sgd_element.h:64 (y = 0 past the end: 0)
      dm = (y * y) * inv * (1.0f - inv);                              // d m / d p = e^{yp} / (1 + e^{yp})^2
    } else {
      const float gs = h.sgda ? 2.0f : 1.0f;
      const float pc = fmaxf(h.min_target, fminf(h.max_target, p));
      m = ok ? gs * (pc - y) : 0.f;                                   // fm_learn_sgd_element.h:60-62
      dm = (ok && p > h.min_target && p < h.max_target) ? gs : 0.f;
    }
  };
  // the affine map of one chunk from its sums
  auto chunk_map = [&](float F, float D, float d, uint32_t cstart) -> AMap {
    const float nc = (cstart < sn) ? (float)min(chunk, sn - cstart) : 0.f;
    F = fmaf(nc * h.reg0, w0s + d, F);
    D = fmaf(nc, h.reg0, D);
    return AMap{1.0f - h.lr * D, -h.lr * (F - D * d)};
  };
  // EVAL: every chunk of the wavefront gets its map at the current path (independent vectors: four in flight)
  auto eval_small = [&](auto ctag) {                                  // chunk <= 64: the lanes of a chunk sum by DPP
    constexpr int C = decltype(ctag)::value;
    constexpr int U = 4;
    for (uint32_t v = 0; v < WSEG / 64u; v += U) {
      if (q0 + 64u * v >= sn) break;                                   // (wave-uniform: nothing left in this sub-segment)
      float m[U], dm[U], d[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t i = q0 + 64u * (v + u) + lane;
        d[u] = s_d[i >> lc];
        eval(i, d[u], m[u], dm[u]);
      }
#pragma unroll
      for (int u = 0; u < U; u++) { m[u] = seg_allsum<C>(m[u], lane); dm[u] = seg_allsum<C>(dm[u], lane); }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t i = q0 + 64u * (v + u) + lane;
        const AMap mp = chunk_map(m[u], dm[u], d[u], (i >> lc) << lc);
        if ((i & (uint32_t)(C - 1)) == 0u) { s_ma[i >> lc] = mp.a; s_mb[i >> lc] = mp.b; }
      }
    }
  };
  auto eval_large = [&]() {                                           // chunk >= 128: vectors of a chunk accumulate (wave-uniform sums)
    const uint32_t vpc = chunk >> 6;
    float F = 0.f, D = 0.f;
    for (uint32_t v = 0; v < WSEG / 64u; v++) {
      const uint32_t c = (q0 + 64u * v) >> lc;
      const float d = s_d[c];
      float m, dm;
      eval(q0 + 64u * v + lane, d, m, dm);
      F += wave_sum_dpp(m); D += wave_sum_dpp(dm);
      if (((v + 1u) % vpc) == 0u) {
        const AMap mp = chunk_map(F, D, d, c << lc);                   // (a chunk past the batch: the identity)
        if (lane == 0) { s_ma[c] = mp.a; s_mb[c] = mp.b; }
        F = 0.f; D = 0.f;
      }
    }
  };
  auto eval_all = [&]() {
    switch (lc) {
      case 0: eval_small(std::integral_constant<int, 1>()); break;
      case 1: eval_small(std::integral_constant<int, 2>()); break;
      case 2: eval_small(std::integral_constant<int, 4>()); break;
      case 3: eval_small(std::integral_constant<int, 8>()); break;
      case 4: eval_small(std::integral_constant<int, 16>()); break;
      case 5: eval_small(std::integral_constant<int, 32>()); break;
      case 6: eval_small(std::integral_constant<int, 64>()); break;
      default: eval_large(); break;
    }
  };
  float my_chg = 3.0e38f;                                             // largest change of this workgroup's last path update (nothing yet)
  bool converged = false, aborted = false;
  unsigned long long* const claim = ps.ctr + PIT_MAX_IT;               // zeroed with the arrival counters before every launch
  if (threadIdx.x == 0) s_abort = 0u;
  double x_end = 0.0;
  for (uint32_t it = 0; it < PIT_MAX_IT; it++) {
    eval_all();
    // SCAN: the wavefront's chunk maps, 64 at a time (lane = chunk): each is replaced by the composite of the wavefront's chunks BEFORE it;
    // `run` ends as the composite of all of them.  (A chunk whose examples lie past the batch was never evaluated: the identity.)
    AMap run = AMap{1.f, 0.f};
    for (uint32_t cb = 0; cb < wch; cb += 64u) {
      const uint32_t c = c_w0 + cb + lane;
      const bool have = cb + lane < wch && (c << lc) < sn;
      float a = have ? s_ma[c] : 1.f, b = have ? s_mb[c] : 0.f;
      amap_scan64(a, b);                                               // inclusive over the 64 chunks of this round
      const float ea = __shfl_up(a, 1u), eb = __shfl_up(b, 1u);
      AMap ex = (lane == 0u) ? AMap{1.f, 0.f} : AMap{ea, eb};          // exclusive
      ex = amap_after(run, ex);                                        // ... behind what the wavefront did before this round
      if (cb + lane < wch) { s_ma[c] = ex.a; s_mb[c] = ex.b; }
      run = amap_after(run, AMap{bcast_f32<1>(a, 63), bcast_f32<1>(b, 63)});
    }
    if (lane == 0) { s_map[wv][0] = run.a; s_map[wv][1] = run.b; }
    __syncthreads();
    // ---- the exchange: publish this workgroup's composite + its last change, wait for everybody's ----
    double* slot = ps.slots + (size_t)(it & 1u) * PIT_MAX_WG * 4;
    if (threadIdx.x == 0) {
      double A = 1.0, B = 0.0;
      for (int w = 0; w < 4; w++) { B = (double)s_map[w][0] * B + (double)s_map[w][1]; A = (double)s_map[w][0] * A; }
      __hip_atomic_store(slot + 4 * g + 0, A, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(slot + 4 * g + 1, B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float wc = nanmax(nanmax(s_chg[0], s_chg[1]), nanmax(s_chg[2], s_chg[3]));
      if (it == 0) wc = my_chg;
      __hip_atomic_store(slot + 4 * g + 2, (double)wc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (nwg > 1) {
        // every workgroup of the grid must be RESIDENT for this to complete (the host gates the launch on the occupancy of the device,
        // launch_scan); if one never arrives all the same -- the chip is shared with launches the host cannot see -- the wait is bounded, the
        // error word gets bit 4, and ONE of the workgroups that are here evaluates the chain serially (below): the batch's bias is still
        // the rule's, the epoch reports FMX_STAT_SCAN_FALLBACK and the handle stops using this kernel
        __hip_atomic_fetch_add(ps.ctr + it, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long cnt = __hip_atomic_load(ps.ctr + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool gone = false;
        for (uint32_t t = 0; cnt < (unsigned long long)nwg && t < ps.spins && !gone; t++) {
          __builtin_amdgcn_s_sleep(1);
          cnt = __hip_atomic_load(ps.ctr + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((t & 255u) == 255u) gone = __hip_atomic_load(claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;   // somebody gave up already
        }
        if (cnt < (unsigned long long)nwg) { atomicOr(ps.err, 4u); s_abort = 1u; }
      }
    }
    __syncthreads();
    if (s_abort) { aborted = true; break; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (wv == 0) {                                                    // lane j holds workgroup j's composite: an inclusive scan across the lanes
      double A = 1.0, B = 0.0, C = 0.0;
      if (lane < nwg) {
        A = __hip_atomic_load(slot + 4 * lane + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        B = __hip_atomic_load(slot + 4 * lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        C = __hip_atomic_load(slot + 4 * lane + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (uint32_t off = 1; off < 64u; off <<= 1) {
        const double pa = __shfl_up(A, off), pb = __shfl_up(B, off), pc = __shfl_up(C, off);
        if (lane >= off) { B = fma(A, pb, B); A = A * pa; C = nanmax(pc, C); }
      }
      const double b_prev = __shfl_up(B, 1u);                          // value after the workgroups before lane's
      const double b_mine = __shfl(b_prev, (int)g), b_all = __shfl(B, (int)(nwg - 1u)), c_all = __shfl(C, (int)(nwg - 1u));
      if (lane == 0) { s_bcast[0] = g ? b_mine : 0.0; s_bcast[1] = b_all; s_bcast[2] = c_all; }
    }
    __syncthreads();
    if (s_bcast[2] < (double)PIT_TOL && s_bcast[1] - s_bcast[1] == 0.0) { converged = true; x_end = s_bcast[1]; break; }   // the path these maps were taken at had settled (and the end value is a number)
    // ---- APPLY: the new path, every chunk at once ----
    float x = (float)s_bcast[0];
    for (uint32_t w = 0; w < wv; w++) x = fmaf(s_map[w][0], x, s_map[w][1]);
    float chg = 0.f;
    for (uint32_t cb = lane; cb < wch; cb += 64u) {
      const uint32_t c = c_w0 + cb;
      if ((c << lc) < sn) {
        const float dn = fmaf(s_ma[c], x, s_mb[c]);
        chg = nanmax(fabsf(dn - s_d[c]), chg);
        s_d[c] = dn;
      }
    }
    for (int o = 32; o > 0; o >>= 1) chg = nanmax(__shfl_xor(chg, o), chg);
    __syncthreads();                                                  // (s_map / s_bcast were read by everybody)
    if (lane == 0) s_chg[wv] = chg;
  }
  if (!converged) {
    // Newton did not settle: the serial chain, by one wavefront of workgroup 0 straight from global memory (k_scan's arithmetic) -- or an
    // exchange timed out: by one wavefront of the first workgroup to claim it
    bool owner = g == 0;
    if (aborted) {
      if (threadIdx.x == 0) s_abort = (atomicCAS(claim, 0ull, 1ull + g) == 0ull) ? 2u : 1u;
      __syncthreads();
      owner = s_abort == 2u;
    }
    if (!owner || wv != 0) return;
    double w = w0;
    for (uint32_t c0 = 0; c0 < n_rows; c0 += chunk) {
      const uint32_t nc = min(chunk, n_rows - c0);
      const float ws = (float)w;
      float a = 0.f;
      for (uint32_t i = lane; i < nc; i += 64u) {
        const float m = multiplier_task<TASK>(h, ws + rest[c0 + i], target[c0 + i]);
        if constexpr (WRITE_MULT) mult[c0 + i] = m;
        a += m;
      }
      const float tot = wave_sum_dpp(a);
      w -= (double)h.lr * ((double)tot + (double)nc * (double)h.reg0 * (double)ws);
    }
    if (lane == 0) { if (hw.ctr) handoff_publish(w0_out, w); else *w0_out = w; }
    return;
  }
  if constexpr (WRITE_MULT) {                                         // the multipliers at the settled path (two-pass form of the rule)
    for (uint32_t v = 0; v < WSEG / 64u; v++) {
      const uint32_t i = q0 + 64u * v + lane;
      if (i < sn) {
        float m, dm;
        eval(i, s_d[i >> lc], m, dm);
        mult[s0 + i] = m;
      }
    }
  }
  if (g == 0 && threadIdx.x == 0) { const double w = w0 + x_end; if (hw.ctr) handoff_publish(w0_out, w); else *w0_out = w; }
}

// no bias: the multipliers are independent of each other
static __global__ void __launch_bounds__(256)
k_mult(const float* __restrict__ rest, const float* __restrict__ target, uint32_t n_rows, Hyper h,
       const double* __restrict__ w0_ptr, float* __restrict__ mult) {
  const float w0 = (h.k0 && w0_ptr) ? (float)(*w0_ptr) : 0.f;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n_rows; e += gridDim.x * blockDim.x)
    mult[e] = multiplier(h, w0 + rest[e], target[e]);
}

// ----------------------------------------------------------------------------------------------
// k_apply: step 3 of the minibatch rule -- one wavefront per example scatters its deltas.
// ----------------------------------------------------------------------------------------------
template <int KP, bool ATOMIC>
__global__ void __launch_bounds__(256)
k_apply(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, uint64_t row0, uint32_t n_rows,
        const Tab tb, Hyper h,
        const float* __restrict__ S, const float* __restrict__ mult) {
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR;
  const uint32_t lane = threadIdx.x & 63u, f = lane % LPR;
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t e = wave0; e < n_rows; e += nwaves) {
    const uint64_t a = row_ptr[row0 + e];
    const uint32_t size = (uint32_t)(row_ptr[row0 + e + 1] - a);
    float sum[VEC];
    load_vec<VEC>(S + (size_t)e * KP + f * VEC, sum);
    const float m = mult[e];
    row_apply<KP, 8, ATOMIC>(ent + a, size, tb, h, sum, m);
  }
}

// ----------------------------------------------------------------------------------------------
// Per-batch transposed rows ("segments"): libFM never shuffles (fm_learn_sgd_element.h:56), so the batches
// are static and their entries can be bucketed ONCE by (batch, feature).  A segment = all occurrences of
// one feature inside one batch, in example order.  k_apply_seg then owns each touched V row exclusively:
// no atomics, no lost updates, and exactly the batch-start-parameter rule of the oracle:
//   v_new = v0 - lr*( sum_e mult_e*x_e*S_ef  -  v0*sum_e mult_e*x_e^2  +  n_occ*regv*v0 )   (fm_sgd.h:44-50 per occurrence)
//   w_new = w0 - lr*( sum_e mult_e*x_e + n_occ*regw*w0 )                                     (fm_sgd.h:38-43)
// ----------------------------------------------------------------------------------------------
struct TEntry { uint32_t e; float x; };     // (example index inside its batch, value)
// one listed (deferred) segment as a 32-byte record: where it is AND its first two occurrences, so that the deferred pass reads the list as one
// coalesced stream and goes straight to the multipliers / sums of those occurrences -- through the index it was descriptor -> {entry, entry} ->
// {multiplier, sums}: one dependent round trip and two 64-byte requests per segment more (97 % of the bench's deferred features are pairs)
struct CDesc { uint32_t feat, a, b, loc; uint32_t e0; float x0; uint32_t e1; float x1; };

// sort keys: (batch << fbits) | feature id, fbits = bits of the largest local feature id (the radix sort then runs over fbits + bits of the
// batch count: 31 bits = 4 passes at the bench shape, where a 32-bit feature field took 5) ; payload: (value bits << 32) | example-in-batch
// (== TEntry in memory)
static __global__ void __launch_bounds__(256)
k_seg_keys(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, uint32_t n_rows, uint32_t B, uint32_t fbits,
           uint64_t* __restrict__ keys, uint64_t* __restrict__ vals, uint32_t* __restrict__ not_ones) {
  // *not_ones is raised when an entry's value is not 1.0f: one-hot data (libFM's usual input) takes the multiplication-free row arithmetic
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  bool other = false;
  for (uint32_t r = wave0; r < n_rows; r += nwaves) {
    const uint64_t a = row_ptr[r], b = row_ptr[r + 1];
    const uint64_t hi = (uint64_t)(r / B) << fbits;
    const uint32_t eb = r % B;
    for (uint64_t i = a + lane; i < b; i += 64) {
      const Entry e = ent[i];
      keys[i] = hi | e.id;
      vals[i] = ((uint64_t)__float_as_uint(e.value) << 32) | eb;
      other |= (e.value != 1.0f);
    }
  }
  if (other) atomicOr(not_ones, 1u);
}
static __global__ void __launch_bounds__(256)
k_seg_heads(const uint64_t* __restrict__ keys, uint64_t nnz, uint32_t* __restrict__ flags) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * blockDim.x)
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// pos = inclusive scan of flags (1-based segment number of every entry)
static __global__ void __launch_bounds__(256)
k_seg_fill(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ pos,
           uint64_t nnz, const uint64_t* __restrict__ row_ptr, uint32_t B, uint32_t fbits,
           uint32_t* __restrict__ seg_feat, uint32_t* __restrict__ seg_rel) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * blockDim.x)
    if (flags[i]) {
      const uint32_t s = pos[i] - 1;
      const uint64_t key = keys[i];
      seg_feat[s] = (uint32_t)(key & ((1ull << fbits) - 1ull));
      seg_rel[s] = (uint32_t)(i - row_ptr[(uint64_t)(key >> fbits) * B]);   // offset inside the batch's entries
    }
}
static __global__ void __launch_bounds__(256)
k_seg_batches(const uint32_t* __restrict__ pos, uint64_t nnz, const uint64_t* __restrict__ row_ptr,
              uint32_t n_rows, uint32_t B, uint32_t n_batches, uint32_t* __restrict__ batch_seg, uint64_t* __restrict__ batch_base) {
  const uint32_t nseg = nnz ? pos[nnz - 1] : 0u;
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b <= n_batches; b += gridDim.x * blockDim.x) {
    const uint64_t r = min((uint64_t)b * B, (uint64_t)n_rows);
    const uint64_t p = row_ptr[r];
    batch_seg[b] = (p < nnz) ? pos[p] - 1 : nseg;
    batch_base[b] = p;                                            // first entry of every batch (row_ptr sampled at multiples of B)
  }
}

// the longest segment = how often the most frequent feature occurs inside ONE batch: the batch rule applies that many
// contributions to the feature at once (a step of lr * count on it), which the caller should keep well below the
// curvature bound (fmx_epoch_stats::max_feature_count).  head[s] = sorted position of the segment's first entry.
static __global__ void __launch_bounds__(256)
k_seg_head_pos(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ pos, uint64_t nnz, uint32_t* __restrict__ head) {
  const uint32_t nseg = nnz ? pos[nnz - 1] : 0u;                   // (read on the device: the host learns the counts in ONE read-back, later)
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nnz; i += (uint64_t)gridDim.x * blockDim.x) {
    if (i == nnz) head[nseg] = (uint32_t)nnz;
    else if (flags[i]) head[pos[i] - 1] = (uint32_t)i;
  }
}
static __global__ void __launch_bounds__(256)
k_seg_max_count(const uint32_t* __restrict__ head, const uint32_t* __restrict__ pos, uint64_t nnz, uint32_t* __restrict__ out) {
  const uint32_t nseg = nnz ? pos[nnz - 1] : 0u;
  uint32_t m = 0;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += gridDim.x * blockDim.x) m = max(m, head[s + 1] - head[s]);
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63u) == 0 && m) atomicMax(out, m);
}

// ---- weight side stream (row_sums): which entries are the LAST occurrence of their feature in the slot's row order ----------------
// last[id mod M] = max over the entries of (entry index + 1); an entry is flagged when the table holds ITS index (a folded table --
// M < n -- only loses flags: two features sharing a bucket flag the later one's last entry, which is a last occurrence all the same)
static __global__ void __launch_bounds__(256)
k_wside_last(const Entry* __restrict__ ent, uint64_t nnz, uint32_t M, uint32_t* __restrict__ last) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * blockDim.x)
    atomicMax(last + (ent[i].id % M), (uint32_t)i + 1u);
}
static __global__ void __launch_bounds__(256)
k_wside_mask(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, uint32_t n_rows, uint32_t M, const uint32_t* __restrict__ last,
             uint64_t* __restrict__ lmask, float* __restrict__ wside) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t r = wave0; r < n_rows; r += nwaves) {
    const uint64_t a = row_ptr[r], b = row_ptr[r + 1];
    bool flag = false;
    if (a + lane < b) flag = last[ent[a + lane].id % M] == (uint32_t)(a + lane) + 1u;
    const uint64_t bits = __ballot(flag);
    if (lane == 0) lmask[r] = bits;                             // (entries beyond the 64th of a row are never streamed)
    for (uint64_t i = a + lane; i < b; i += 64) wside[i] = __builtin_nanf("");   // nothing is known yet: every entry gathers
  }
}

// ---- what FUSED_EXACT needs to know about a batch (built once with the segments) ----------------------------------
// rows that do not fit the register path of k_fused are deferred as a whole
static __global__ void __launch_bounds__(256)
k_seg_slow_rows(const uint64_t* __restrict__ row_ptr, uint32_t n_rows, uint32_t cap, uint64_t* __restrict__ cmask) {
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += gridDim.x * blockDim.x)
    cmask[r] = ((uint32_t)(row_ptr[r + 1] - row_ptr[r]) > cap) ? ~0ull : 0ull;
}
// one thread per segment: cflag[s] = the feature occurs more than once in its batch, or its (only) example is a deferred
// row; for multi-occurrence features the bit of every occurrence is set in its row's mask (the row is searched for the
// id: <= 64 entries, one-time cost).  keys / vals are the sorted (batch, feature) keys and {example, value} payloads.
static __global__ void __launch_bounds__(256)
k_seg_mark(const uint64_t* __restrict__ keys, const TEntry* __restrict__ vals, const uint32_t* __restrict__ head, const uint32_t* __restrict__ pos, uint64_t nnz,
           const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, uint32_t B, uint32_t fbits, uint32_t cap, uint32_t long_rows,
           uint64_t* __restrict__ cmask, uint32_t* __restrict__ cflag) {
  // long_rows == 0: no row of the slot exceeds the register path (max_row <= cap) -- a feature that occurs once is never deferred and the two
  // random row_ptr reads per segment drop out (they were 17 GB of 64-byte sectors for the bench's 134 M segments: most of this kernel)
  const uint32_t nseg = nnz ? pos[nnz - 1] : 0u;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += gridDim.x * blockDim.x) {
    const uint32_t a = head[s], b = head[s + 1];
    bool coll = (b - a) > 1;
    if (!coll && !long_rows) { cflag[s] = 0u; continue; }
    const uint64_t key = keys[a];
    const uint64_t rbase = (uint64_t)(key >> fbits) * B;
    const uint32_t feat = (uint32_t)(key & ((1ull << fbits) - 1ull));
    if (!coll) {
      const uint64_t r = rbase + vals[a].e;
      coll = (uint32_t)(row_ptr[r + 1] - row_ptr[r]) > cap;
    } else {
      for (uint32_t i = a; i < b; i++) {
        const uint64_t r = rbase + vals[i].e;
        const uint64_t ra = row_ptr[r];
        const uint32_t size = (uint32_t)(row_ptr[r + 1] - ra);
        if (size > cap) continue;                              // deferred row: mask is all ones already
        uint64_t bits = 0;
        for (uint32_t q = 0; q < size; q++) if (ent[ra + q].id == feat) bits |= 1ull << q;
        atomicOr((unsigned long long*)(cmask + r), (unsigned long long)bits);
      }
    }
    cflag[s] = coll ? 1u : 0u;
  }
}
// compaction: cseg[cpos[s]] = batch-local index of the flagged segment s (cpos = exclusive scan of cflag)
static __global__ void __launch_bounds__(256)
k_seg_compact(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ head, const uint32_t* __restrict__ cflag,
              const uint32_t* __restrict__ cpos, uint32_t nseg, const uint32_t* __restrict__ batch_seg, uint32_t* __restrict__ cseg,
              const uint32_t* __restrict__ seg_feat, const uint32_t* __restrict__ seg_rel, const uint64_t* __restrict__ row_ptr,
              uint32_t n_rows, uint32_t B, uint32_t fbits, CDesc* __restrict__ cdesc, const TEntry* __restrict__ vals) {
  // cdesc: the listed segment as ONE record {feature, first entry, end entry (both relative to the batch's entries), batch-local index}:
  // the deferred pass reads the list as a coalesced 16-byte stream instead of an index followed by three dependent gathers
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += gridDim.x * blockDim.x)
    if (cflag[s]) {
      const uint32_t bt = (uint32_t)(keys[head[s]] >> fbits);
      const uint32_t loc = s - batch_seg[bt];
      cseg[cpos[s]] = loc;
      const uint64_t r0 = (uint64_t)bt * B, r1 = min((uint64_t)(bt + 1) * B, (uint64_t)n_rows);
      const uint32_t end = (s + 1 < batch_seg[bt + 1]) ? seg_rel[s + 1] : (uint32_t)(row_ptr[r1] - row_ptr[r0]);
      CDesc d;
      d.feat = seg_feat[s]; d.a = seg_rel[s]; d.b = end; d.loc = loc;
      const uint32_t h0 = head[s];                               // sorted position of the segment's first entry (== its entry in t_ent)
      const TEntry t0 = vals[h0];
      d.e0 = t0.e; d.x0 = t0.x; d.e1 = 0u; d.x1 = 0.f;
      if (end - d.a >= 2u) { const TEntry t1 = vals[h0 + 1]; d.e1 = t1.e; d.x1 = t1.x; }
      cdesc[cpos[s]] = d;
    }
}
// the three counts the host needs to size the slot's arrays, in one place: {segments, deferred segments, longest segment}
static __global__ void k_seg_counts(const uint32_t* __restrict__ pos, uint64_t nnz, const uint32_t* __restrict__ cpos, const uint32_t* __restrict__ cflag,
                                    const uint32_t* __restrict__ max_count, uint32_t* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const uint32_t nseg = nnz ? pos[nnz - 1] : 0u;
    out[0] = nseg;
    out[1] = nseg ? cpos[nseg - 1] + cflag[nseg - 1] : 0u;
    out[2] = *max_count;
  }
}
static __global__ void __launch_bounds__(256)
k_seg_cbatch(const uint32_t* __restrict__ cpos, const uint32_t* __restrict__ cflag, uint32_t nseg,
             const uint32_t* __restrict__ batch_seg, uint32_t n_batches, uint32_t* __restrict__ cbatch) {
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b <= n_batches; b += gridDim.x * blockDim.x) {
    const uint32_t s = batch_seg[b];
    cbatch[b] = (s < nseg) ? cpos[s] : (nseg ? cpos[nseg - 1] + cflag[nseg - 1] : 0u);
  }
}

// One wavefront owns blocks of 64 consecutive segments: the descriptors, first occurrences and their
// multipliers are fetched lane-parallel (coalesced), then U segment groups at a time are broadcast and their
// V rows + S rows gathered together (2*U row loads in flight per wavefront).
// seg_idx == nullptr: all nseg (== nseg_batch) segments of the batch; else the nseg listed ones (the features that
// occur more than once in the batch: what FUSED_EXACT leaves behind).
struct SegWork {
  const TEntry* t_ent; const uint32_t* seg_feat; const uint32_t* seg_rel; const uint32_t* seg_idx;
  uint32_t nseg, nseg_batch, batch_nnz;
  const float* S; const float* mult;
  const CDesc* cdesc;      // with seg_idx: the listed segments as 32-byte records incl. their first two occurrences (nullptr: look them up)
  unsigned long long* done_ctr; unsigned long long done_val;   // device hand-off: "everything before this launch on its stream has completed"
};
// SPW = segments per wavefront-block (<= 64).  64 for the dense form (millions of segments: plenty of wavefronts); 16 for
// the short list of deferred features (a few 100 000): with 64 the pass ran on ~5 000 wavefronts, each a serial chain of
// eight dependent gather rounds -- 166 us for 350 MB (2 TB/s, latency-bound) -- four times as many shorter chains fill the chip.
// PRE2 (the deferred list, where every feature has >= 2 occurrences): the SECOND occurrence's descriptor, multiplier and S row
// are fetched in the same pipelined rounds as the first instead of in the serial tail loop (entry -> multiplier -> S row, three
// dependent gathers per segment) -- same operations in the same order, the loop only starts at the third occurrence.
// SGDA (fm_learn_sgd_element_adapt_reg, batch form): the regularisation is the learned per-group table (2 reg(g) theta, :155-163)
// and the gradient sums of the step are kept for the lambda step (gw, gv: :153, :161).
struct SgdaExtra { const double* reg; const uint32_t* grp; float* gw; float* gv; };
template <int KP, int U, int SPW = 64, bool PRE2 = (SPW < 64), bool SGDA = false>
__device__ __forceinline__ void apply_seg_block(const SegWork& sw, uint32_t blk, const Tab tb, const Hyper& h, const SgdaExtra sx = SgdaExtra{}) {
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR, EPI = Map<KP>::EPI;
  const uint32_t lane = threadIdx.x & 63u, g = lane / LPR, f = lane % LPR;
  const TEntry* __restrict__ t_ent = sw.t_ent;
  const float* __restrict__ S = sw.S;
  const float* __restrict__ mult = sw.mult;
  const uint32_t cnt = min((uint32_t)SPW, sw.nseg - blk);
  uint32_t jl = 0, al = 0, bl = 0, el = 0, e2l = 0; float xl = 0.f, ml = 0.f, x2l = 0.f, m2l = 0.f;
  if (lane < cnt) {
    if (sw.cdesc) {
      const uint4* rec = reinterpret_cast<const uint4*>(sw.cdesc + blk + lane);
      const uint4 d = rec[0], o = rec[1];
      jl = d.x; al = d.y; bl = d.z;
      el = o.x; xl = __uint_as_float(o.y);
      if (PRE2 && bl - al >= 2) { e2l = o.z; x2l = __uint_as_float(o.w); m2l = mult[e2l]; }
    } else {
      const uint32_t s = sw.seg_idx ? sw.seg_idx[blk + lane] : blk + lane;
      jl = sw.seg_feat[s];
      al = sw.seg_rel[s];
      bl = (s + 1 < sw.nseg_batch) ? sw.seg_rel[s + 1] : sw.batch_nnz;
      const TEntry te = load_stream8(t_ent + al);
      el = te.e; xl = te.x;
      if (PRE2 && bl - al >= 2) { const TEntry t2 = load_stream8(t_ent + al + 1); e2l = t2.e; x2l = t2.x; m2l = mult[e2l]; }
    }
    ml = mult[el];
  }
  // one segment per wavefront (the small-batch path, where a batch is a latency chain): the third and later occurrences -- the dense fields'
  // ids are met ~8-47 times per 512 rows -- are asked for NOW, with the first two's rows and sums, not a round trip later
  TEntry te_pre; te_pre.e = 0; te_pre.x = 0.f; float tm_pre = 0.f;
  if constexpr (SPW == 1 && EPI == 1 && PRE2) {
    const uint32_t a0 = bcast_u32<1>(al, 0), b0 = bcast_u32<1>(bl, 0);
    if (cnt && a0 + 2u + lane < b0) { te_pre = load_stream8(t_ent + a0 + 2u + lane); tm_pre = mult[te_pre.e]; }
  }
  for (uint32_t i = 0; i < cnt; i += EPI * U) {
    float v0[U][VEC], sf[U][VEC], sf2[PRE2 ? U : 1][VEC];
    float wv0[U];                                                // the segments' linear weights: asked for WITH their rows (read after the row stores
#pragma unroll                                                   // they cost one more dependent round trip per round)
    for (int u = 0; u < U; u++) {
      const uint32_t idx = i + u * EPI + g;
      const uint32_t j = bcast_u32<EPI>(jl, idx & 63u);
      const uint32_t e = bcast_u32<EPI>(el, idx & 63u);
      uint32_t e2 = 0, n2 = 0;
      if (PRE2) { e2 = bcast_u32<EPI>(e2l, idx & 63u); n2 = bcast_u32<EPI>(bl - al, idx & 63u); }
      wv0[u] = 0.f;
      if (idx < cnt) {
        if (h.k1 && f == 0) wv0[u] = tb.w[(size_t)j * tb.ws];
        row_ld<VEC, 8>(tb, (size_t)j, f * VEC, v0[u]);
        load_vec<VEC>(S + (size_t)e * KP + f * VEC, sf[u]);
        if (PRE2) {
          if (n2 >= 2) load_vec<VEC>(S + (size_t)e2 * KP + f * VEC, sf2[u]);
          else {
#pragma unroll
            for (int v = 0; v < VEC; v++) sf2[u][v] = 0.f;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t idx = i + u * EPI + g;
      const uint32_t j = bcast_u32<EPI>(jl, idx & 63u);
      const uint32_t a = bcast_u32<EPI>(al, idx & 63u);
      const uint32_t b = bcast_u32<EPI>(bl, idx & 63u);
      const float x = bcast_f32<EPI>(xl, idx & 63u);
      const float m = bcast_f32<EPI>(ml, idx & 63u);
      float x2 = 0.f, m2 = 0.f;                               // (cross-lane reads run with every lane enabled: outside the guard)
      if (PRE2) { x2 = bcast_f32<EPI>(x2l, idx & 63u); m2 = bcast_f32<EPI>(m2l, idx & 63u); }
      if (idx < cnt) {
        float G[VEC]; float A, Gw;
        const float mx = m * x;
#pragma unroll
        for (int v = 0; v < VEC; v++) G[v] = mx * sf[u][v];
        A = mx * x; Gw = mx;
        uint32_t i2 = PRE2 ? a + 2 : a + 1;
        if (PRE2 && b - a >= 2) {
          const float mx2 = m2 * x2;
#pragma unroll
          for (int v = 0; v < VEC; v++) G[v] = fmaf(mx2, sf2[u][v], G[v]);
          A = fmaf(mx2, x2, A); Gw += mx2;
        }
        // further occurrences of the feature in this batch (a frequent feature of Criteo-shaped rows has dozens to thousands per
        // batch).  Added in occurrence order, so the result does not depend on how they are fetched:
        if constexpr (EPI == 1) {
          // one row per wave-wide load: the whole wavefront is here together.  64 descriptors and their multipliers arrive
          // lane-parallel (one coalesced load + one gather), then the S rows TL at a time with the descriptors broadcast by
          // v_readlane -- per TL occurrences ONE dependent round trip instead of three per occurrence.
          // (32 rows per round for the one-segment-per-wavefront path was measured in round 4: k_apply_seg_scan 7.6 -> 8.1 us per
          //  512-row batch of Criteo-shaped rows -- most segments there hold ~8 occurrences and pay for the longer unrolled round)
          constexpr int TL = (VEC == 1) ? 16 : 8;
          for (uint32_t base = i2; base < b; base += 64) {
            const uint32_t cc = min(64u, b - base);
            TEntry te; te.e = 0; te.x = 0.f; float tm = 0.f;
            if (SPW == 1 && PRE2 && base == i2) { te = te_pre; tm = tm_pre; }          // (asked for with the descriptor, above)
            else if (lane < cc) { te = load_stream8(t_ent + base + lane); tm = mult[te.e]; }
            for (uint32_t q0 = 0; q0 < cc; q0 += TL) {
              float s2[TL][VEC];
#pragma unroll
              for (int q = 0; q < TL; q++) {
                const uint32_t e2 = bcast_u32<1>(te.e, (q0 + q) & 63u);
                if (q0 + q < cc) load_vec<VEC>(S + (size_t)e2 * KP + f * VEC, s2[q]);
                else {
#pragma unroll
                  for (int v = 0; v < VEC; v++) s2[q][v] = 0.f;
                }
              }
#pragma unroll
              for (int q = 0; q < TL; q++) {
                const float x2 = bcast_f32<1>(te.x, (q0 + q) & 63u), m2 = bcast_f32<1>(tm, (q0 + q) & 63u);
                if (q0 + q < cc) {
                  const float mx2 = m2 * x2;
#pragma unroll
                  for (int v = 0; v < VEC; v++) G[v] = fmaf(mx2, s2[q][v], G[v]);
                  A = fmaf(mx2, x2, A); Gw += mx2;
                }
              }
            }
          }
        } else {
          // several rows per wave-wide load (KP < 64): every lane group walks its own segment, TL occurrences at a time
          constexpr int TL = 8;
          for (; i2 < b; i2 += TL) {
            TEntry tt[TL]; float mm[TL]; float s2[TL][VEC];
#pragma unroll
            for (int q = 0; q < TL; q++) { tt[q].e = 0; tt[q].x = 0.f; if (i2 + q < b) tt[q] = t_ent[i2 + q]; }
#pragma unroll
            for (int q = 0; q < TL; q++) {
              mm[q] = 0.f;
              if (i2 + q < b) { mm[q] = mult[tt[q].e]; load_vec<VEC>(S + (size_t)tt[q].e * KP + f * VEC, s2[q]); }
              else {
#pragma unroll
                for (int v = 0; v < VEC; v++) s2[q][v] = 0.f;
              }
            }
#pragma unroll
            for (int q = 0; q < TL; q++) {
              if (i2 + q < b) {
                const float mx2 = mm[q] * tt[q].x;
#pragma unroll
                for (int v = 0; v < VEC; v++) G[v] = fmaf(mx2, s2[q][v], G[v]);
                A = fmaf(mx2, tt[q].x, A); Gw += mx2;
              }
            }
          }
        }
        const float nocc = (float)(b - a);
        float nv[VEC];
        if constexpr (SGDA) {
          const double* rg = sx.reg + (size_t)(sx.grp ? sx.grp[j] : 0u) * (1 + KP);
          float sh[VEC];
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            const float vv = v0[u][v];
            sh[v] = G[v] - vv * A;                               // sum over the occurrences of mult x (S_f - v x)  (:161)
            nv[v] = vv - h.lr * (sh[v] + nocc * 2.0f * (float)rg[1 + f * VEC + v] * vv);
          }
          if (f * VEC < tb.rs) {
            store_vec<VEC>(tb.V + (size_t)j * tb.rs + f * VEC, nv);
            store_vec<VEC>(sx.gv + (size_t)j * tb.rs + f * VEC, sh);
          }
          if (h.k1 && f == 0) {
            float* pw = tb.w + (size_t)j * tb.ws;
            const float wv = wv0[u];
            sx.gw[j] = Gw;                                       // :153
            *pw = wv - h.lr * (Gw + nocc * 2.0f * (float)rg[0] * wv);
          }
        } else {
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            const float vv = v0[u][v];
            nv[v] = vv - h.lr * (G[v] - vv * A + nocc * h.regv * vv);
          }
          row_st<VEC, 8>(tb, (size_t)j, f * VEC, nv);
          if (h.k1 && f == 0) {
            float* pw = tb.w + (size_t)j * tb.ws;
            const float wv = wv0[u];
            *pw = wv - h.lr * (Gw + nocc * h.regw * wv);
          }
        }
      }
    }
  }
}
// the bias recurrence of a SMALL batch (the batches the stability cut leaves of rows with frequent features: a few hundred to a few
// thousand examples) on one wavefront, straight from global memory -- k_scan's arithmetic (micro-chunks of `chunk` examples, every
// example of a chunk sees the bias of the chunk start, fm_sgd.h:34-37 summed per chunk) without its LDS tiles
struct ScanSmall { const float* rest; const float* target; const double* w0_in; double* w0_out; uint32_t n_rows, chunk; };
// COH: rest[] and the incoming bias were written by other workgroups of THIS launch (the XCD-resident epoch): L2-served loads
template <bool COH = false>
__device__ __forceinline__ void scan_small(const ScanSmall sc, const Hyper& h) {
  const uint32_t lane = threadIdx.x & 63u;
  double w0 = COH ? ld_l2(sc.w0_in) : *sc.w0_in;
  if (sc.n_rows <= 1024u && ((sc.chunk & 63u) == 0 || sc.chunk == 32u || sc.chunk == 16u)) {
    // the whole batch in registers first (one round trip to memory, not one per micro-chunk): element c0 + i + 64 j of a chunk
    // sits in lane i, register (c0 / 64 + j)
    float r[16], y[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t i = (uint32_t)j * 64u + lane;
      r[j] = (i < sc.n_rows) ? (COH ? ld_l2(sc.rest + i) : sc.rest[i]) : 0.f;
      y[j] = (i < sc.n_rows) ? sc.target[i] : 0.f;
    }
    if (sc.chunk < 64u) {
      // micro-chunks of 16 / 32 examples (the default since round 5): a DPP row / half of the 64-example register vector at a time
      const uint32_t C = sc.chunk, ns = 64u / C;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        if ((uint32_t)j * 64u < sc.n_rows) {                     // (wave-uniform)
          const uint32_t i = (uint32_t)j * 64u + lane;
          for (uint32_t sb = 0; sb < ns; sb++) {
            const float w0s = h.k0 ? (float)w0 : 0.f;
            const float m = (i < sc.n_rows) ? multiplier_fast(h, w0s + r[j], y[j]) : 0.f;
            const float tot = (C == 32u) ? group_sum_dpp<32>(m, sb) : group_sum_dpp<16>(m, sb);
            const uint32_t lo = (uint32_t)j * 64u + sb * C;
            const uint32_t nc = (lo < sc.n_rows) ? min(C, sc.n_rows - lo) : 0u;
            if (h.k0 && nc) w0 -= (double)h.lr * ((double)tot + (double)nc * (double)h.reg0 * (double)w0s);
          }
        }
      }
      if (lane == 0) *sc.w0_out = w0;
      return;
    }
    const uint32_t per = sc.chunk >> 6;                        // registers per micro-chunk
    for (uint32_t c0 = 0; c0 < sc.n_rows; c0 += sc.chunk) {
      const uint32_t nc = min(sc.chunk, sc.n_rows - c0);
      const float w0s = h.k0 ? (float)w0 : 0.f;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const uint32_t i = (uint32_t)j * 64u + lane;
        if ((uint32_t)j >= (c0 >> 6) && (uint32_t)j < (c0 >> 6) + per && i < sc.n_rows) acc += multiplier_fast(h, w0s + r[j], y[j]);
      }
      const float tot = wave_sum_dpp(acc);
      if (h.k0) w0 -= (double)h.lr * ((double)tot + (double)nc * (double)h.reg0 * (double)w0s);
    }
  } else {
    for (uint32_t c0 = 0; c0 < sc.n_rows; c0 += sc.chunk) {
      const uint32_t nc = min(sc.chunk, sc.n_rows - c0);
      const float w0s = h.k0 ? (float)w0 : 0.f;
      float acc = 0.f;
      for (uint32_t i = lane; i < nc; i += 64) acc += multiplier_fast(h, w0s + (COH ? ld_l2(sc.rest + c0 + i) : sc.rest[c0 + i]), sc.target[c0 + i]);
      const float tot = wave_sum_dpp(acc);
      if (h.k0) w0 -= (double)h.lr * ((double)tot + (double)nc * (double)h.reg0 * (double)w0s);
    }
  }
  if (lane == 0) *sc.w0_out = w0;
}
// k_apply_seg_scan: the deferred features of a small batch AND its bias recurrence in one launch (the last workgroup's first
// wavefront runs the recurrence): a small batch is a few microseconds of work, every launch it needs costs as much again
template <int KP, int U, int SPW>
__global__ void __launch_bounds__(256)
k_apply_seg_scan(const SegWork sw, const Tab tb, Hyper h, const ScanSmall sc) {
  if (blockIdx.x == gridDim.x - 1) {
    if (threadIdx.x < 64 && h.k0) scan_small(sc, h);
    return;
  }
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const uint32_t nwaves = (gridDim.x - 1) * (blockDim.x >> 6);
  for (uint32_t blk = wave0 * (uint32_t)SPW; blk < sw.nseg; blk += nwaves * (uint32_t)SPW) apply_seg_block<KP, U, SPW>(sw, blk, tb, h);
}
template <int KP, int U, int SPW>
__global__ void __launch_bounds__(256)
k_apply_seg(const SegWork sw, const Tab tb, Hyper h) {
  if (sw.done_ctr && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(sw.done_ctr, sw.done_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t blk = wave0 * (uint32_t)SPW; blk < sw.nseg; blk += nwaves * (uint32_t)SPW) apply_seg_block<KP, U, SPW>(sw, blk, tb, h);
}

// ----------------------------------------------------------------------------------------------
// k_fused: one wavefront per example, ONE pass over HBM: the gathered V rows stay in registers (ZR row-slots x VEC
// floats per lane), the prediction, multiplier and fm_SGD update are computed in-register and the rows are written
// straight back.  V is read once and written once -- the algorithmic minimum of a training step.
// Rows longer than ZR*EPI (or 64) take the two-pass path (row_sums + row_apply; the second read is an L2 hit).
// w0 is FROZEN for the launch: a per-example read-modify-write of one scalar from ~6000 resident wavefronts is a
// single-address hot spot and, worse, an unstable recurrence at that staleness.  The kernel writes
// rest_e = y-hat_e - w0 and k_scan advances w0 afterwards with the exact micro-chunk recurrence (fm_sgd.h:34-37).
//
// Variants:
//   FUSED_STORE / FUSED_ATOMIC : HOGWILD -- every entry is written back (plain stores / fp32 atomic adds); rows in
//       flight race on shared features.
//   FUSED_EXACT : the MINIBATCH rule (oracle fmo_sgd_epoch_minibatch_ex, bias_lag >= 1) in one pass.  The rule takes
//       every sum and every gradient from BATCH-START parameters, so an entry whose feature occurs ONCE in the batch
//       can be updated by its own example's wavefront right away -- nobody else reads or writes that row in this
//       launch.  cmask[row] has bit i set when entry i's feature occurs more than once in the batch (or the row does
//       not fit the register path): those entries are NOT written here (their rows keep the batch-start value for
//       every reader); the example leaves its factor sums S_e and multiplier behind and k_apply_seg finishes exactly
//       those features afterwards (one owner per feature, all occurrences summed).  Bit-for-bit the batch rule.
//   FUSED_APPLY : the second half of that for the SPLIT step (a feature shard after the exchange, the two-pass form): the
//       complete sums S_e and the multiplier are GIVEN (S_out / mult_out are read), nothing is predicted; the rows of the
//       batch-unique features are gathered, updated and written back, the others are left to k_apply_seg as above.
// ----------------------------------------------------------------------------------------------
enum { FUSED_STORE = 0, FUSED_ATOMIC = 1, FUSED_EXACT = 2, FUSED_APPLY = 3 };
template <int KP, int ZR, int VAR>
__global__ void __launch_bounds__(256)
k_fused(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, const float* __restrict__ target,
        uint64_t row0, uint32_t n_rows, const Tab tb, Hyper h,
        const double* __restrict__ w0_ptr, float* __restrict__ rest_out,
        const uint64_t* __restrict__ cmask, float* __restrict__ S_out, float* __restrict__ mult_out, uint32_t fixed_nnz,
        uint32_t* __restrict__ handoff_err, const uint64_t* __restrict__ lmask, float* __restrict__ wside) {
  // wside != nullptr (FUSED_EXACT, FMX_FLAG_KEEP_WSIDE): the slot's weight side stream is kept current -- entry i of a row gets the NEW
  // w_j when lmask says it is the last occurrence of its feature in the slot and this wavefront is the one that updates it (not deferred),
  // NaN otherwise: 128 coalesced bytes per example (see row_sums)
  // handoff_err != nullptr: *w0_ptr is a hand-off slot (published by the recurrence kernel of an earlier batch on the side stream, possibly
  // still W0_PENDING): read it past the caches, and wait -- only where the multiplier needs it, behind the row gathers -- if it is not there yet
  // fixed_nnz != 0: every row of the slot holds exactly that many entries (one-hot field data): the entry list of example e starts at
  // (row0 + e) * fixed_nnz and the row_ptr round trip drops out of the chain row_ptr -> entries -> rows (what a small batch consists of)
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR, EPI = Map<KP>::EPI;
  constexpr bool ATOMIC = (VAR == FUSED_ATOMIC), EXACT = (VAR == FUSED_EXACT), APPLY = (VAR == FUSED_APPLY), MASKED = EXACT || APPLY;
  const uint32_t lane = threadIdx.x & 63u, g = lane / LPR, f = lane % LPR;
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  float w0s = 0.f;
  unsigned long long w0u = 0ull;
  bool w0_late = false;                                          // the bias is still to be taken out of w0u (hand-off)
  bool w0_bad = false;                                           // ... and never arrived: the examples of this wavefront take no step
  if (h.k0) {
    if (handoff_err) { w0u = handoff_peek(w0_ptr); w0_late = true; }
    else w0s = (float)(*w0_ptr);
  }
  for (uint32_t e = wave0; e < n_rows; e += nwaves) {
    const uint64_t a = fixed_nnz ? (row0 + e) * (uint64_t)fixed_nnz : row_ptr[row0 + e];
    const uint32_t size = fixed_nnz ? fixed_nnz : (uint32_t)(row_ptr[row0 + e + 1] - a);
    const Entry* __restrict__ row = ent + a;
    const float y = APPLY ? 0.f : target[row0 + e];
    uint64_t cm = 0, lm = 0;
    if constexpr (MASKED) cm = cmask[row0 + e];
    if constexpr (EXACT) { if (wside) lm = lmask[row0 + e]; }   // (with the mask: one round trip, long before it is needed)
    if (size <= (uint32_t)(ZR * EPI) && size <= 64u) {
      Entry en; en.id = 0; en.value = 0.f;
      float wv = 0.f;
      if (lane < size) {
        en = load_stream8(row + lane);
        if (h.k1) wv = load_w(tb.w + (size_t)en.id * tb.ws);
      }
      // hand-off: the bias slot was asked for before the entry list, so it is here when the entries are (loads return in order) and leaves
      // its registers before the row slots fill; a slot that is still pending (never, at bias_lag >= 2) is waited for here
      if (w0_late) { w0s = (float)handoff_wait(w0_ptr, w0u, handoff_err, &w0_bad); w0_late = false; }
      // phase A: issue every gather of the row back-to-back (ids / values are re-broadcast later instead of
      // being kept: with EPI == 1 they are wave-uniform and live in SGPRs for the duration of one use).
      // Every cross-lane broadcast below runs with ALL lanes active (outside the idx < size guards): a ds_bpermute
      // from a lane that is masked off returns 0, and with EPI > 1 the source lane t*EPI+g of a short row's last
      // entries can belong to a lane whose own idx is >= size.
      float vr[ZR][VEC];
#pragma unroll
      for (int t = 0; t < ZR; t++) {
        const uint32_t idx = t * EPI + g;
        const uint32_t id = bcast_u32<EPI>(en.id, idx & 63u);
        if (idx < size) {                                      // (APPLY: deferred rows are gathered too -- 4 % of them; a test per row slot costs more)
          row_ld<VEC, 1>(tb, (size_t)id, f * VEC, vr[t]);
        } else {
#pragma unroll
          for (int v = 0; v < VEC; v++) vr[t][v] = 0.f;
        }
      }
      // phase B: sums (fm_model.h:116-125); lanes >= size hold value 0, so no guard is needed on x
      float sum[VEC]; float sq = 0.f;
      float mult;
      if constexpr (APPLY) {                                   // complete sums and multiplier of the example: given
        load_vec<VEC>(S_out + (size_t)e * KP + f * VEC, sum);
        mult = mult_out[e];
      } else {
#pragma unroll
      for (int v = 0; v < VEC; v++) sum[v] = 0.f;
#pragma unroll
      for (int t = 0; t < ZR; t++) {
        const uint32_t idx = t * EPI + g;
        float x = bcast_f32<EPI>(en.value, idx & 63u);
        if (idx >= size) x = 0.f;
#pragma unroll
        for (int v = 0; v < VEC; v++) {
          const float d = vr[t][v] * x;
          sum[v] += d;
          sq = fmaf(d, d, sq);
        }
      }
#pragma unroll
      for (int v = 0; v < VEC; v++) sum[v] = subgroup_allsum<LPR>(sum[v]);
      float part = wv * en.value - 0.5f * sq;
      if (lane < LPR) {
#pragma unroll
        for (int v = 0; v < VEC; v++) part = fmaf(0.5f * sum[v], sum[v], part);
      }
      const float rest = wave_sum_dpp(part);
      if (lane == 0) rest_out[e] = rest;
      mult = w0_bad ? 0.f : multiplier(h, w0s + rest, y);
      }
      if constexpr (EXACT) {
        if (cm != 0) {                                         // some feature of this example is finished by k_apply_seg
          if (lane < LPR) store_vec<VEC>(S_out + (size_t)e * KP + lane * VEC, sum);
          if (lane == 0) mult_out[e] = mult;
        }
      }
      float w_keep = __builtin_nanf("");
      if (h.k1 && lane < size && !(MASKED && ((cm >> lane) & 1ull))) {            // fm_sgd.h:38-43
        const float dw = -h.lr * (mult * en.value + h.regw * wv);
        if (ATOMIC) unsafeAtomicAdd(tb.w + (size_t)en.id * tb.ws, dw);
        else tb.w[(size_t)en.id * tb.ws] = wv + dw;
        w_keep = wv + dw;
      }
      if constexpr (EXACT) {
        if (wside && lane < size) {
          __builtin_nontemporal_store(((lm >> lane) & 1ull) ? w_keep : __builtin_nanf(""), wside + a + lane);
        }
      }
      // phase C: fm_sgd.h:44-50 on the register-resident rows, written straight back
#pragma unroll
      for (int t = 0; t < ZR; t++) {
        const uint32_t idx = t * EPI + g;
        const uint32_t id = bcast_u32<EPI>(en.id, idx & 63u);
        const float x = bcast_f32<EPI>(en.value, idx & 63u);
        if (idx < size && !(MASKED && ((cm >> (idx & 63u)) & 1ull)) && f * VEC < tb.rs) {
          float* pv = tb.V + (size_t)id * tb.rs + f * VEC;
          float nv[VEC];
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            const float vv = vr[t][v];
            const float grad = sum[v] * x - vv * x * x;
            const float dv = -h.lr * (mult * grad + h.regv * vv);
            if (ATOMIC) unsafeAtomicAdd(pv + v, dv); else nv[v] = vv + dv;
          }
          if (!ATOMIC) store_row<VEC, 2>(pv, nv);
        }
      }
    } else if constexpr (!APPLY) {                             // (APPLY: such a row is deferred as a whole, cm == ~0)
      if (w0_late) { w0s = (float)handoff_wait(w0_ptr, w0u, handoff_err, &w0_bad); w0_late = false; }
      float sum[VEC], sq, lin;
      row_sums<KP, 8>(row, size, tb, h.k1, sum, sq, lin);
#pragma unroll
      for (int v = 0; v < VEC; v++) sum[v] = subgroup_allsum<LPR>(sum[v]);
      float part = lin - 0.5f * sq;
      if (lane < LPR) {
#pragma unroll
        for (int v = 0; v < VEC; v++) part = fmaf(0.5f * sum[v], sum[v], part);
      }
      const float rest = wave_sum_dpp(part);
      if (lane == 0) rest_out[e] = rest;
      const float mult = w0_bad ? 0.f : multiplier(h, w0s + rest, y);
      if constexpr (EXACT) {                                   // the whole row is deferred (cmask = all ones)
        if (lane < LPR) store_vec<VEC>(S_out + (size_t)e * KP + lane * VEC, sum);
        if (lane == 0) mult_out[e] = mult;
      } else {
        row_apply<KP, 8, ATOMIC>(row, size, tb, h, sum, mult);
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// k_place_probe: what a training step does to the factor table, on a table that holds nothing yet -- every wavefront reads 32
// random rows of `row_floats` floats and writes them back scaled (non-temporal both ways; the table is all zeros at that point).  fmx_create times it on candidate
// allocations of the table: on this part the rate of exactly this access pattern depends on WHERE in HBM an allocation landed
// (scripts/ubench/placement.hip: 5.9 vs 5.3 TB/s for same-sized tables of one process), so the table goes where it runs fastest.
// ----------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256)
k_place_probe(float* __restrict__ tab, uint64_t n_rows, uint32_t row_floats, uint32_t n_waves, uint64_t salt) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= n_waves) return;
  const uint32_t lanes_per_row = min(64u, row_floats), rows_per_load = 64u / lanes_per_row;
  const uint32_t sub = lane / lanes_per_row, col = lane % lanes_per_row;
  float v[32]; uint64_t at[32];
#pragma unroll
  for (int t = 0; t < 32; t++) {
    const uint64_t r = (uint64_t)(((unsigned __int128)mix64(((uint64_t)wave * 32 + t) * rows_per_load + sub + salt) * n_rows) >> 64);
    at[t] = r * row_floats + col;
    v[t] = __builtin_nontemporal_load(tab + at[t]);
  }
#pragma unroll
  for (int t = 0; t < 32; t++) __builtin_nontemporal_store(v[t] * 0.999f, tab + at[t]);   // (a plain write-back of the loaded value is a no-op the
}                                                                                           //  compiler removes together with the load; the table is zero)

// k_place_pair: the probe that CLASSIFIES physical memory (fmx_create, scripts/ubench/placement_classes.hip): every wavefront reads 32
// random 256-byte rows of the union of two equally sized pieces a and b (1 << rows_shift rows each; a == b: one piece alone) and
// writes them back.  Two pieces of the same memory class run at the one-piece rate, pieces of different classes ~25 % faster.
static __global__ void __launch_bounds__(256)
k_place_pair(float* __restrict__ a, float* __restrict__ b, uint32_t rows_shift, uint32_t n_waves, uint64_t salt) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= n_waves) return;
  float v[32]; float* at[32];
#pragma unroll
  for (int t = 0; t < 32; t++) {
    const uint64_t r = mix64((uint64_t)wave * 32 + t + salt) >> (63 - rows_shift);          // rows_shift + 1 random bits
    at[t] = ((r >> rows_shift) ? b : a) + (r & ((1ull << rows_shift) - 1)) * 64 + lane;
    v[t] = __builtin_nontemporal_load(at[t]);
  }
#pragma unroll
  for (int t = 0; t < 32; t++) __builtin_nontemporal_store(v[t] * 0.999f, at[t]);
}

// the same for the linear weights: random 4-byte read-modify-writes of a zeroed table (32 w_j per example in the step)
static __global__ void __launch_bounds__(256)
k_place_probe_w(float* __restrict__ tab, uint64_t n, uint64_t total, uint64_t salt) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= total) return;
  const uint64_t j = (uint64_t)(((unsigned __int128)mix64(tid + salt) * n) >> 64);
  tab[j] = tab[j] * 0.999f;
}

// ----------------------------------------------------------------------------------------------
// k_sequential: the reference trajectory (batch = 1, storage order) on ONE wavefront, for parity.
// Loads bypass the per-CU L1 (agent-scope relaxed atomics -> sc1) and every row ends with a drain of
// the store queue, so row r+1 sees row r's update exactly like fm_learn_sgd_element.h:56-67.
// Entries are updated one at a time in row order, so a repeated id inside a row sees its own earlier
// update (fm_sgd.h:44-50 semantics).  Sums are accumulated in fp64; parameters are stored fp32.
// ----------------------------------------------------------------------------------------------
template <int KP>
__global__ void __launch_bounds__(64)
k_sequential(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, const float* __restrict__ target,
             uint32_t n_rows, const Tab tb, Hyper h, double* w0_ptr) {
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR;
  const uint32_t lane = threadIdx.x;
  const bool act = lane < LPR && lane * VEC < tb.rs;           // (the lanes of the row's elements: rows are tb.rs floats, row_ld above)
  double w0 = *w0_ptr;
  for (uint32_t r = 0; r < n_rows; r++) {
    const uint64_t a = row_ptr[r];
    const uint32_t size = (uint32_t)(row_ptr[r + 1] - a);
    double sum[VEC]; double sq = 0.0, lin = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; v++) sum[v] = 0.0;
    for (uint32_t i = 0; i < size; i++) {
      const Entry e = ent[a + i];
      if (h.k1 && lane == 0)
        lin += (double)__hip_atomic_load(tb.w + (size_t)e.id * tb.ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * (double)e.value;
      if (act) {
#pragma unroll
        for (int v = 0; v < VEC; v++) {
          const float vv = __hip_atomic_load(tb.V + (size_t)e.id * tb.rs + lane * VEC + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const double d = (double)vv * (double)e.value;
          sum[v] += d;
          sq += d * d;
        }
      }
    }
    double part = lin - 0.5 * sq;
    if (act) {
#pragma unroll
      for (int v = 0; v < VEC; v++) part += 0.5 * sum[v] * sum[v];
    }
    double p = (h.k0 ? w0 : 0.0) + wave_sum_d(part);
    const double y = (double)target[r];
    double mult;
    if (h.task == 0) {
      p = fmin(h.max_d, p);
      p = fmax(h.min_d, p);
      mult = -(y - p);
    } else {
      mult = -y * (1.0 - 1.0 / (1.0 + exp(-y * p)));
    }
    if (h.k0) w0 -= h.lr_d * (mult + h.reg0_d * w0);
    for (uint32_t i = 0; i < size; i++) {
      const Entry e = ent[a + i];
      const double x = (double)e.value;
      if (h.k1 && lane == 0) {
        const double wv = (double)__hip_atomic_load(tb.w + (size_t)e.id * tb.ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(tb.w + (size_t)e.id * tb.ws, (float)(wv - h.lr_d * (mult * x + h.regw_d * wv)),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (act) {
#pragma unroll
        for (int v = 0; v < VEC; v++) {
          float* pv = tb.V + (size_t)e.id * tb.rs + lane * VEC + v;
          const double vv = (double)__hip_atomic_load(pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const double grad = sum[v] * x - vv * x * x;
          __hip_atomic_store(pv, (float)(vv - h.lr_d * (mult * grad + h.regv_d * vv)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // a later entry of this row (repeated id) and the next row must observe these stores
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if (lane == 0) *w0_ptr = w0;
}

// ----------------------------------------------------------------------------------------------
// k_sgda: fm_learn_sgd_element_adapt_reg (`-method sgda`, src/libfm/src/fm_learn_sgd_element_adapt_reg.h), ONE
// wavefront, the reference's strictly online interleaving: for every train row a theta step (:136-169: like fm_SGD
// but mult = 2(p-y), regularisation 2*reg*theta with the LEARNED reg_w / reg_v[f], and the gradient of every
// touched parameter remembered in grad_w / grad_v), then (from the 2nd epoch on) a lambda step on the next
// validation row (:201-248 through predict_scaled :171-199).  One attribute group.  Sums in fp64, parameters and
// shadow gradients stored fp32.  reg: [0] = reg_w, [1+f] = reg_v[f].
// ----------------------------------------------------------------------------------------------
template <int KP>
__global__ void __launch_bounds__(64)
k_sgda(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, const float* __restrict__ target, uint32_t n_rows,
       const Entry* __restrict__ vent, const uint64_t* __restrict__ vrow_ptr, const float* __restrict__ vtarget, uint32_t v_rows,
       const Tab tb, float* gw, float* gv, Hyper h, double* w0_ptr, double* reg, int do_lambda) {
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR;
  const uint32_t lane = threadIdx.x;
  const bool act = lane < LPR && lane * VEC < tb.rs;           // (the lanes of the row's elements: rows are tb.rs floats, row_ld above)
  double w0 = *w0_ptr;
  double reg_w = reg[0];
  double reg_v[VEC];
#pragma unroll
  for (int v = 0; v < VEC; v++) reg_v[v] = act ? reg[1 + lane * VEC + v] : 0.0;
  uint32_t vpos = 0;                                                        // validation->data->begin() (:266)
#define LD(p) ((double)__hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
#define ST(p, val) __hip_atomic_store((p), (float)(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
  for (uint32_t r = 0; r < n_rows; r++) {
    // ---------------- theta step (:136-169)
    const uint64_t a = row_ptr[r];
    const uint32_t size = (uint32_t)(row_ptr[r + 1] - a);
    double sum[VEC]; double sq = 0.0, lin = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; v++) sum[v] = 0.0;
    for (uint32_t i = 0; i < size; i++) {
      const Entry e = ent[a + i];
      if (h.k1 && lane == 0) lin += LD(tb.w + (size_t)e.id * tb.ws) * (double)e.value;
      if (act) {
#pragma unroll
        for (int v = 0; v < VEC; v++) {
          const double d = LD(tb.V + (size_t)e.id * tb.rs + lane * VEC + v) * (double)e.value;
          sum[v] += d; sq += d * d;
        }
      }
    }
    double part = lin - 0.5 * sq;
    if (act) {
#pragma unroll
      for (int v = 0; v < VEC; v++) part += 0.5 * sum[v] * sum[v];
    }
    double p = (h.k0 ? w0 : 0.0) + wave_sum_d(part);
    const double y = (double)target[r];
    double mult;
    if (h.task == 0) { p = fmin(h.max_d, p); p = fmax(h.min_d, p); mult = 2 * (p - y); }
    else mult = y * ((1.0 / (1.0 + exp(-y * p))) - 1.0);
    if (h.k0) w0 -= h.lr_d * (mult + 2 * 0.0 * w0);                          // reg_0 = 0 (:100)
    for (uint32_t i = 0; i < size; i++) {
      const Entry e = ent[a + i];
      const double x = (double)e.value;
      if (h.k1 && lane == 0) {
        float* pw = tb.w + (size_t)e.id * tb.ws;
        const double wv = LD(pw);
        const double g = mult * x;
        ST(gw + e.id, g);
        ST(pw, wv - h.lr_d * ((double)(float)g + 2 * reg_w * wv));
      }
      if (act) {
#pragma unroll
        for (int v = 0; v < VEC; v++) {
          float* pv = tb.V + (size_t)e.id * tb.rs + lane * VEC + v;
          const double vv = LD(pv);
          const double g = mult * (x * (sum[v] - vv * x));
          ST(gv + (size_t)e.id * tb.rs + lane * VEC + v, g);
          ST(pv, vv - h.lr_d * ((double)(float)g + 2 * reg_v[v] * vv));
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!do_lambda || v_rows == 0) continue;
    // ---------------- lambda step on the next validation row (:271-276, :201-248)
    if (vpos >= v_rows) vpos = 0;
    const uint64_t va = vrow_ptr[vpos];
    const uint32_t vsize = (uint32_t)(vrow_ptr[vpos + 1] - va);
    const double vy = (double)vtarget[vpos];
    vpos++;
    double lw = 0.0, plin = 0.0;
    double s_dash[VEC], s_f[VEC], s_df[VEC]; double q_dash = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; v++) { s_dash[v] = 0.0; s_f[v] = 0.0; s_df[v] = 0.0; }
    for (uint32_t i = 0; i < vsize; i++) {
      const Entry e = vent[va + i];
      const double x = (double)e.value;
      if (h.k1 && lane == 0) {
        const double wv = LD(tb.w + (size_t)e.id * tb.ws);
        const double w_dash = wv - h.lr_d * (LD(gw + e.id) + 2 * reg_w * wv);   // predict_scaled :178-184
        plin += w_dash * x;
        lw += x * wv;                                                            // :215-218
      }
      if (act) {
#pragma unroll
        for (int v = 0; v < VEC; v++) {
          const double vv = LD(tb.V + (size_t)e.id * tb.rs + lane * VEC + v);
          const double v_dash = vv - h.lr_d * (LD(gv + (size_t)e.id * tb.rs + lane * VEC + v) + 2 * reg_v[v] * vv);
          const double d = v_dash * x;
          s_dash[v] += d; q_dash += d * d;                                       // :186-196
          s_f[v] += vv * x;                                                      // :233-238
          s_df[v] += d * vv * x;
        }
      }
    }
    double vpart = plin - 0.5 * q_dash;
    if (act) {
#pragma unroll
      for (int v = 0; v < VEC; v++) vpart += 0.5 * s_dash[v] * s_dash[v];
    }
    double vp = (h.k0 ? w0 : 0.0) + wave_sum_d(vpart);
    double grad_loss;
    if (h.task == 0) { vp = fmin(h.max_d, vp); vp = fmax(h.min_d, vp); grad_loss = 2 * (vp - vy); }
    else grad_loss = vy * ((1.0 / (1.0 + exp(-vy * vp))) - 1.0);
    if (h.k1) {                                                                  // :213-224
      const double lwt = -2 * h.lr_d * wave_sum_d(lw);
      reg_w -= h.lr_d * grad_loss * lwt;
      reg_w = fmax(0.0, reg_w);
    }
    if (act) {                                                                   // :240-246
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        const double lambda_v_grad = -2 * h.lr_d * (s_dash[v] * s_f[v] - s_df[v]);
        reg_v[v] -= h.lr_d * grad_loss * lambda_v_grad;
        reg_v[v] = fmax(0.0, reg_v[v]);
      }
    }
  }
#undef LD
#undef ST
  if (lane == 0) { *w0_ptr = w0; reg[0] = reg_w; }
  if (act) {
#pragma unroll
    for (int v = 0; v < VEC; v++) reg[1 + lane * VEC + v] = reg_v[v];
  }
}

// ----------------------------------------------------------------------------------------------
// SGDA in batch form (oracle fmo_sgda_epoch_minibatch; C-ABI fmx_sgda_epoch_minibatch).  The theta step of a batch is the
// minibatch rule (k_rowsums -> k_scan -> k_sgda_apply_seg) with the learner's multiplier (Hyper::sgda), reg_0 = 0 and the
// LEARNED regularisation 2 reg(g[,f]) theta per occurrence; the shadow gradient of a touched parameter becomes the sum of
// its occurrences' gradients.  The lambda step of the batch: one wavefront per validation row (k_sgda_lambda) evaluates
// sgd_lambda_step (:201-248 through predict_scaled :171-199) with the regularisation frozen at its batch-start values and
// leaves its workgroup's summed changes in dpart; k_sgda_reg_update adds the partials in a fixed order, clamped at 0.
// reg / dreg: [G][1 + KP] doubles, reg[g*(1+KP)] = reg_w(g), reg[g*(1+KP)+1+f] = reg_v(g,f).
// ----------------------------------------------------------------------------------------------
template <int KP, int U>
__global__ void __launch_bounds__(256)
k_sgda_apply_seg(const SegWork sw, const Tab tb, Hyper h, const double* __restrict__ reg, const uint32_t* __restrict__ grp,
                 float* __restrict__ gw, float* __restrict__ gv) {
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  const SgdaExtra sx{reg, grp, gw, gv};
  for (uint32_t blk = wave0 * 64u; blk < sw.nseg; blk += nwaves * 64u) apply_seg_block<KP, U, 64, false, true>(sw, blk, tb, h, sx);
}

// one wavefront (= one workgroup) per validation row at a time; rows vpos0 .. vpos0 + n_rows - 1, cyclic.  The workgroup SUMS the
// changes of all its rows (registers when there is one attribute group, LDS tables per group otherwise) and leaves ONE partial
// [G][1 + KP] in dpart[blockIdx.x]; k_sgda_reg_update adds the partials in a fixed order (deterministic; the first version added
// every row's 1 + k changes to dreg with fp64 atomics -- 65 536 rows on 65 addresses: 0.85 ms of a 1.2 ms batch).
// Entries are loaded one per lane and broadcast; U parameter rows and their gradient rows are in flight per lane.
template <int KP, bool GROUPED>
__global__ void __launch_bounds__(64)
k_sgda_lambda(const Entry* __restrict__ vent, const uint64_t* __restrict__ vrow_ptr, const float* __restrict__ vtarget, uint32_t v_rows,
              uint32_t vpos0, uint32_t n_rows, const Tab tb, const float* __restrict__ gw, const float* __restrict__ gv, Hyper h,
              const double* __restrict__ w0_ptr, const double* __restrict__ reg, double* __restrict__ dpart,
              const uint32_t* __restrict__ grp, uint32_t G) {
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR, U = 4;
  extern __shared__ double lam_lds[];
  double* lwg = lam_lds;                       // GROUPED: [G]         sum x w of the row, per group
  double* sfg = lwg + G;                       //          [G][KP]     sum v x
  double* sdfg = sfg + (size_t)G * KP;         //          [G][KP]     sum v' x v x
  double* acc = sdfg + (size_t)G * KP;         //          [G][1 + KP] this workgroup's changes
  const uint32_t lane = threadIdx.x;
  const bool act = lane < LPR && lane * VEC < tb.rs;           // (the lanes of the row's elements: rows are tb.rs floats, row_ld above)
  const double w0 = h.k0 ? *w0_ptr : 0.0;
  const uint32_t cells = G * (1 + KP);
  double acc_w = 0.0, acc_v[VEC];              // !GROUPED: the same in registers (lane 0 / factor lanes)
  double rg0 = 0.0, rgv[VEC];
#pragma unroll
  for (int v = 0; v < VEC; v++) { acc_v[v] = 0.0; rgv[v] = 0.0; }
  if (!GROUPED) {
    rg0 = reg[0];
    if (act) {
#pragma unroll
      for (int v = 0; v < VEC; v++) rgv[v] = reg[1 + lane * VEC + v];
    }
  } else {
    for (uint32_t c = lane; c < cells; c += 64) acc[c] = 0.0;
  }
  for (uint32_t t = blockIdx.x; t < n_rows; t += gridDim.x) {
    const uint32_t r = (uint32_t)(((uint64_t)vpos0 + t) % v_rows);
    const uint64_t va = vrow_ptr[r];
    const uint32_t vsize = (uint32_t)(vrow_ptr[r + 1] - va);
    const double vy = (double)vtarget[r];
    if (GROUPED) {
      for (uint32_t c = lane; c < G; c += 64) lwg[c] = 0.0;
      for (uint32_t c = lane; c < G * KP; c += 64) { sfg[c] = 0.0; sdfg[c] = 0.0; }
      __syncthreads();
    }
    double vpart = 0.0, lw = 0.0;              // per lane: share of (linear part - 0.5 sum (v' x)^2); sum x w (one group)
    double s_dash[VEC], sf[VEC], sdf[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) { s_dash[v] = 0.0; sf[v] = 0.0; sdf[v] = 0.0; }
    for (uint32_t base = 0; base < vsize; base += 64) {
      const uint32_t cnt = min(64u, vsize - base);
      Entry en; en.id = 0; en.value = 0.f;
      uint32_t gl = 0;
      if (lane < cnt) {
        en = load_stream8(vent + va + base + lane);
        if (GROUPED) gl = grp ? grp[en.id] : 0u;
        if (h.k1) {
          const double x = (double)en.value;
          const double wv = (double)tb.w[(size_t)en.id * tb.ws];
          const double r0 = GROUPED ? reg[(size_t)gl * (1 + KP)] : rg0;
          vpart += (wv - h.lr_d * ((double)gw[en.id] + 2 * r0 * wv)) * x;           // predict_scaled :178-184
          if (GROUPED) unsafeAtomicAdd(lwg + gl, x * wv); else lw += x * wv;           // :215-218
        }
      }
      for (uint32_t i = 0; i < cnt; i += U) {
        float vr[U][VEC], gr[U][VEC]; float xs[U]; uint32_t gs[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const uint32_t idx = i + u;
          const uint32_t id = bcast_u32<1>(en.id, idx & 63u);
          xs[u] = bcast_f32<1>(en.value, idx & 63u);
          gs[u] = GROUPED ? bcast_u32<1>(gl, idx & 63u) : 0u;
          if (idx < cnt && act) {
            load_vec<VEC>(tb.V + (size_t)id * tb.rs + lane * VEC, vr[u]);
            load_vec<VEC>(gv + (size_t)id * tb.rs + lane * VEC, gr[u]);
          } else {
            xs[u] = 0.f;
#pragma unroll
            for (int v = 0; v < VEC; v++) { vr[u][v] = 0.f; gr[u][v] = 0.f; }
          }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (i + u >= cnt || !act) continue;
          const double x = (double)xs[u];
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            const double vv = (double)vr[u][v];
            const double rv = GROUPED ? reg[(size_t)gs[u] * (1 + KP) + 1 + lane * VEC + v] : rgv[v];
            const double v_dash = vv - h.lr_d * ((double)gr[u][v] + 2 * rv * vv);
            const double d = v_dash * x;
            s_dash[v] += d; vpart -= 0.5 * d * d;                                      // :186-196
            if (GROUPED) {
              const size_t c = (size_t)gs[u] * KP + lane * VEC + v;
              sfg[c] += vv * x;                                                        // :233-238 (one lane per cell: no race)
              sdfg[c] += d * vv * x;
            } else { sf[v] += vv * x; sdf[v] += d * vv * x; }
          }
        }
      }
    }
    if (act) {
#pragma unroll
      for (int v = 0; v < VEC; v++) vpart += 0.5 * s_dash[v] * s_dash[v];
    }
    double vp = w0 + wave_sum_d(vpart);
    double grad_loss;
    if (h.task == 0) { vp = fmin(h.max_d, vp); vp = fmax(h.min_d, vp); grad_loss = 2 * (vp - vy); }
    else grad_loss = vy * ((1.0 / (1.0 + exp(-vy * vp))) - 1.0);
    const double scale = -h.lr_d * grad_loss * (-2 * h.lr_d);
    if (!GROUPED) {
      lw = wave_sum_d(lw);
      if (h.k1 && lane == 0) acc_w += scale * lw;                                    // :219-221
      if (act) {
#pragma unroll
        for (int v = 0; v < VEC; v++) acc_v[v] += scale * (s_dash[v] * sf[v] - sdf[v]);   // :240-243
      }
    } else {
      __syncthreads();
      for (uint32_t g = 0; g < G; g++) {
        if (h.k1 && lane == 0) acc[(size_t)g * (1 + KP)] += scale * lwg[g];
        if (act) {
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            const size_t c = (size_t)g * KP + lane * VEC + v;
            acc[(size_t)g * (1 + KP) + 1 + lane * VEC + v] += scale * (s_dash[v] * sfg[c] - sdfg[c]);
          }
        }
      }
      __syncthreads();
    }
  }
  double* out = dpart + (size_t)blockIdx.x * cells;
  if (!GROUPED) {
    if (lane == 0) out[0] = acc_w;
    if (act) {
#pragma unroll
      for (int v = 0; v < VEC; v++) out[1 + lane * VEC + v] = acc_v[v];
    }
  } else {
    __syncthreads();
    for (uint32_t c = lane; c < cells; c += 64) out[c] = acc[c];
  }
}
// reg += sum over the workgroups' partials (fixed order), clamped at 0 (:222, :245)
static __global__ void __launch_bounds__(256)
k_sgda_reg_update(double* __restrict__ reg, const double* __restrict__ dpart, uint32_t n_part, uint32_t cells, int KP, int k1) {
  __shared__ double red[256];
  for (uint32_t c = blockIdx.x; c < cells; c += gridDim.x) {
    double a = 0.0;
    for (uint32_t p = threadIdx.x; p < n_part; p += 256) a += dpart[(size_t)p * cells + c];
    red[threadIdx.x] = a;
    __syncthreads();
    for (uint32_t o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) {
      const bool is_w = (c % (uint32_t)(1 + KP)) == 0;
      if (!is_w || k1) reg[c] = fmax(0.0, reg[c] + red[0]);
    }
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------------------------
// block-structured rows (`-relation`, src/libfm/src/relation.h:32-60): a main row c is followed by the row
// data_row_to_relation_row[c] of every relation block, block attribute ids shifted by the block's attr_offset
// (libfm.cpp:213-216).  The learner of the reference never materialises these rows (fm_learn_mcmc.h:478-527 works
// on per-block caches); on the device the expanded CSR is built once at upload and every kernel runs unchanged.
// ----------------------------------------------------------------------------------------------
#define FMX_MAX_RELATIONS 8
struct BlockRel { const Entry* ent; const uint64_t* row_ptr; const uint32_t* map; uint32_t attr_offset; uint32_t pad; };
struct BlockRels { BlockRel r[FMX_MAX_RELATIONS]; uint32_t n; };

static __global__ void __launch_bounds__(256)
k_block_sizes(const uint64_t* __restrict__ main_ptr, uint32_t n_rows, BlockRels rels, uint64_t* __restrict__ sizes) {
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c <= n_rows; c += gridDim.x * blockDim.x) {
    uint64_t s = 0;
    if (c < n_rows) {
      s = main_ptr[c + 1] - main_ptr[c];
      for (uint32_t r = 0; r < rels.n; r++) { const uint32_t b = rels.r[r].map[c]; s += rels.r[r].row_ptr[b + 1] - rels.r[r].row_ptr[b]; }
    }
    sizes[c] = s;                                            // sizes[n_rows] = 0: the scan then yields row_ptr[n_rows] = nnz
  }
}

static __global__ void __launch_bounds__(256)
k_block_fill(const Entry* __restrict__ main_ent, const uint64_t* __restrict__ main_ptr, uint32_t n_rows, BlockRels rels,
             const uint64_t* __restrict__ out_ptr, Entry* __restrict__ out) {
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n_rows; c += gridDim.x * blockDim.x) {
    uint64_t o = out_ptr[c];
    for (uint64_t i = main_ptr[c]; i < main_ptr[c + 1]; i++) out[o++] = main_ent[i];
    for (uint32_t r = 0; r < rels.n; r++) {
      const BlockRel& b = rels.r[r];
      const uint32_t br = b.map[c];
      for (uint64_t i = b.row_ptr[br]; i < b.row_ptr[br + 1]; i++) { Entry e = b.ent[i]; e.id += b.attr_offset; out[o++] = e; }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// k_sgda_groups: the same learner with attribute groups (`-meta`): reg_w(g), reg_v(g,f) and the per-group sums of the
// lambda step (lambda_w_grad(g), sum_f(g), sum_f_dash_f(g); :96-98, :213-247) live in LDS:
//   regw[G] | regv[G][KP] | lwg[G] | sfg[G][KP] | sdfg[G][KP] | stamp[G]
// Only the groups present in a validation row are zeroed / updated: for an absent group the reference's update is
// reg -= lr * grad_loss * (-0.0), i.e. the identity for every finite grad_loss.  Each (g, factor) cell is owned by
// one lane; lane 0 owns the linear cells and the stamps, hence the barriers around the stamp reads.
// reg (global): [G][1 + KP], reg[g*(1+KP)] = reg_w(g), reg[g*(1+KP)+1+f] = reg_v(g,f).
// ----------------------------------------------------------------------------------------------
template <int KP>
__global__ void __launch_bounds__(64)
k_sgda_groups(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, const float* __restrict__ target, uint32_t n_rows,
              const Entry* __restrict__ vent, const uint64_t* __restrict__ vrow_ptr, const float* __restrict__ vtarget, uint32_t v_rows,
              const Tab tb, float* gw, float* gv, Hyper h, double* w0_ptr, double* reg, int do_lambda,
              const uint32_t* __restrict__ grp, uint32_t G) {
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR;
  extern __shared__ double sgda_lds[];
  double* regw = sgda_lds;
  double* regv = regw + G;
  double* lwg = regv + (size_t)G * KP;
  double* sfg = lwg + G;
  double* sdfg = sfg + (size_t)G * KP;
  uint32_t* stamp = (uint32_t*)(sdfg + (size_t)G * KP);
  const uint32_t lane = threadIdx.x;
  const bool act = lane < LPR && lane * VEC < tb.rs;           // (the lanes of the row's elements: rows are tb.rs floats, row_ld above)
  for (uint32_t g = lane; g < G; g += 64) { regw[g] = reg[(size_t)g * (1 + KP)]; stamp[g] = 0; }
  for (uint32_t i = lane; i < G * KP; i += 64) regv[i] = reg[(size_t)(i / KP) * (1 + KP) + 1 + (i % KP)];
  __syncthreads();
  double w0 = *w0_ptr;
  uint32_t vpos = 0, cur = 0;
#define LD(p) ((double)__hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
#define ST(p, val) __hip_atomic_store((p), (float)(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
  for (uint32_t r = 0; r < n_rows; r++) {
    // ---------------- theta step (:136-169)
    const uint64_t a = row_ptr[r];
    const uint32_t size = (uint32_t)(row_ptr[r + 1] - a);
    double sum[VEC]; double sq = 0.0, lin = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; v++) sum[v] = 0.0;
    for (uint32_t i = 0; i < size; i++) {
      const Entry e = ent[a + i];
      if (h.k1 && lane == 0) lin += LD(tb.w + (size_t)e.id * tb.ws) * (double)e.value;
      if (act) {
#pragma unroll
        for (int v = 0; v < VEC; v++) {
          const double d = LD(tb.V + (size_t)e.id * tb.rs + lane * VEC + v) * (double)e.value;
          sum[v] += d; sq += d * d;
        }
      }
    }
    double part = lin - 0.5 * sq;
    if (act) {
#pragma unroll
      for (int v = 0; v < VEC; v++) part += 0.5 * sum[v] * sum[v];
    }
    double p = (h.k0 ? w0 : 0.0) + wave_sum_d(part);
    const double y = (double)target[r];
    double mult;
    if (h.task == 0) { p = fmin(h.max_d, p); p = fmax(h.min_d, p); mult = 2 * (p - y); }
    else mult = y * ((1.0 / (1.0 + exp(-y * p))) - 1.0);
    if (h.k0) w0 -= h.lr_d * (mult + 2 * 0.0 * w0);                          // reg_0 = 0 (:100)
    for (uint32_t i = 0; i < size; i++) {
      const Entry e = ent[a + i];
      const uint32_t g = grp[e.id];
      const double x = (double)e.value;
      if (h.k1 && lane == 0) {
        float* pw = tb.w + (size_t)e.id * tb.ws;
        const double wv = LD(pw);
        const double gr = mult * x;
        ST(gw + e.id, gr);
        ST(pw, wv - h.lr_d * ((double)(float)gr + 2 * regw[g] * wv));
      }
      if (act) {
#pragma unroll
        for (int v = 0; v < VEC; v++) {
          float* pv = tb.V + (size_t)e.id * tb.rs + lane * VEC + v;
          const double vv = LD(pv);
          const double gr = mult * (x * (sum[v] - vv * x));
          ST(gv + (size_t)e.id * tb.rs + lane * VEC + v, gr);
          ST(pv, vv - h.lr_d * ((double)(float)gr + 2 * regv[(size_t)g * KP + lane * VEC + v] * vv));
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!do_lambda || v_rows == 0) continue;
    // ---------------- lambda step on the next validation row (:271-276, :201-248)
    if (vpos >= v_rows) vpos = 0;
    const uint64_t va = vrow_ptr[vpos];
    const uint32_t vsize = (uint32_t)(vrow_ptr[vpos + 1] - va);
    const double vy = (double)vtarget[vpos];
    vpos++;
    cur += 2;                                                                    // stamp == cur: sums valid; cur+1: updated
    double plin = 0.0, q_dash = 0.0;
    double s_dash[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) s_dash[v] = 0.0;
    for (uint32_t i = 0; i < vsize; i++) {
      const Entry e = vent[va + i];
      const uint32_t g = grp[e.id];
      const double x = (double)e.value;
      const bool fresh = stamp[g] != cur;
      __syncthreads();
      if (fresh) {
        if (lane == 0) { lwg[g] = 0.0; stamp[g] = cur; }
        if (act) {
#pragma unroll
          for (int v = 0; v < VEC; v++) { sfg[(size_t)g * KP + lane * VEC + v] = 0.0; sdfg[(size_t)g * KP + lane * VEC + v] = 0.0; }
        }
      }
      __syncthreads();
      if (h.k1 && lane == 0) {
        const double wv = LD(tb.w + (size_t)e.id * tb.ws);
        const double w_dash = wv - h.lr_d * (LD(gw + e.id) + 2 * regw[g] * wv);   // predict_scaled :178-184
        plin += w_dash * x;
        lwg[g] += x * wv;                                                          // :215-218
      }
      if (act) {
#pragma unroll
        for (int v = 0; v < VEC; v++) {
          const size_t c = (size_t)g * KP + lane * VEC + v;
          const double vv = LD(tb.V + (size_t)e.id * tb.rs + lane * VEC + v);
          const double v_dash = vv - h.lr_d * (LD(gv + (size_t)e.id * tb.rs + lane * VEC + v) + 2 * regv[c] * vv);
          const double d = v_dash * x;
          s_dash[v] += d; q_dash += d * d;                                       // :186-196
          sfg[c] += vv * x;                                                      // :233-238
          sdfg[c] += d * vv * x;
        }
      }
    }
    double vpart = plin - 0.5 * q_dash;
    if (act) {
#pragma unroll
      for (int v = 0; v < VEC; v++) vpart += 0.5 * s_dash[v] * s_dash[v];
    }
    double vp = (h.k0 ? w0 : 0.0) + wave_sum_d(vpart);
    double grad_loss;
    if (h.task == 0) { vp = fmin(h.max_d, vp); vp = fmax(h.min_d, vp); grad_loss = 2 * (vp - vy); }
    else grad_loss = vy * ((1.0 / (1.0 + exp(-vy * vp))) - 1.0);
    for (uint32_t i = 0; i < vsize; i++) {                                       // every group of the row, once
      const uint32_t g = grp[vent[va + i].id];
      const bool todo = stamp[g] == cur;
      __syncthreads();
      if (todo) {
        if (lane == 0) {
          stamp[g] = cur + 1;
          if (h.k1) {                                                            // :213-224
            const double lwt = -2 * h.lr_d * lwg[g];
            regw[g] = fmax(0.0, regw[g] - h.lr_d * grad_loss * lwt);
          }
        }
        if (act) {                                                               // :240-246
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            const size_t c = (size_t)g * KP + lane * VEC + v;
            const double lambda_v_grad = -2 * h.lr_d * (s_dash[v] * sfg[c] - sdfg[c]);
            regv[c] = fmax(0.0, regv[c] - h.lr_d * grad_loss * lambda_v_grad);
          }
        }
      }
      __syncthreads();
    }
  }
#undef LD
#undef ST
  __syncthreads();
  if (lane == 0) *w0_ptr = w0;
  for (uint32_t g = lane; g < G; g += 64) reg[(size_t)g * (1 + KP)] = regw[g];
  for (uint32_t i = lane; i < G * KP; i += 64) reg[(size_t)(i / KP) * (1 + KP) + 1 + (i % KP)] = regv[i];
}

// ----------------------------------------------------------------------------------------------
// evaluation: y-hat = w0 + rest, then the reductions of fm_learn.h:113-153
// acc[0] = sum err^2 (clamped), acc[1] = sum |err|, acc[2] = #correct sign
// ----------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256)
k_eval(const float* __restrict__ rest, const float* __restrict__ target, uint32_t n_rows, Hyper h,
       const double* __restrict__ w0_ptr, double* __restrict__ acc) {
  const float w0 = h.k0 ? (float)(*w0_ptr) : 0.f;
  double se = 0, ae = 0, nc = 0;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n_rows; e += gridDim.x * blockDim.x) {
    const float p = w0 + rest[e];
    const float y = target[e];
    if (h.task == 0) {
      const float pc = fmaxf(h.min_target, fminf(h.max_target, p));
      const double err = (double)pc - (double)y;
      se += err * err; ae += fabs(err);
    } else {
      if (((p >= 0) && (y >= 0)) || ((p < 0) && (y < 0))) nc += 1;
    }
  }
  se = wave_sum_d(se); ae = wave_sum_d(ae); nc = wave_sum_d(nc);
  if ((threadIdx.x & 63) == 0) {
    unsafeAtomicAdd(acc + 0, se); unsafeAtomicAdd(acc + 1, ae); unsafeAtomicAdd(acc + 2, nc);
  }
}

static __global__ void __launch_bounds__(256)
k_yhat(const float* __restrict__ rest, uint32_t n_rows, int k0, const double* __restrict__ w0_ptr, float* __restrict__ yhat) {
  const float w0 = k0 ? (float)(*w0_ptr) : 0.f;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n_rows; e += gridDim.x * blockDim.x)
    yhat[e] = w0 + rest[e];
}

// ----------------------------------------------------------------------------------------------
// parameter staging: reference layout (fp64, factor-major v[f][j]) <-> device (fp32, V[local*KP+f]).
// stage holds `cnt` consecutive GLOBAL features j0.. of factor rows: stage[f*cnt + (j-j0)].
// Ownership and local row of a feature: Shard (above).
// ----------------------------------------------------------------------------------------------
static __global__ void k_stage_in(const double* __restrict__ stage, uint64_t j0, uint32_t cnt, int k, int KP, Shard sh, Tab tb) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t total = (uint64_t)cnt * tb.rs;                  // (rows are tb.rs floats: the padding behind the k factors is zeroed)
  if (t >= total) return;
  const uint32_t jj = (uint32_t)(t / tb.rs); const int f = (int)(t % tb.rs);
  uint32_t jl;
  if (!sh.place((uint32_t)(j0 + jj), &jl)) return;
  tb.V[(size_t)jl * tb.rs + f] = (f < k) ? (float)stage[(size_t)f * cnt + jj] : 0.f;
}
static __global__ void k_stage_out(double* __restrict__ stage, uint64_t j0, uint32_t cnt, int k, int KP, Shard sh, Tab tb) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t total = (uint64_t)cnt * k;
  if (t >= total) return;
  const int f = (int)(t / cnt); const uint32_t jj = (uint32_t)(t % cnt);
  uint32_t jl;
  if (!sh.place((uint32_t)(j0 + jj), &jl)) return;
  stage[(size_t)f * cnt + jj] = (double)tb.V[(size_t)jl * tb.rs + f];
}
static __global__ void k_w_in(const double* __restrict__ stage, uint64_t j0, uint32_t cnt, Shard sh, Tab tb) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cnt) return;
  uint32_t jl;
  if (sh.place((uint32_t)(j0 + t), &jl)) tb.w[(size_t)jl * tb.ws] = (float)stage[t];
}
static __global__ void k_w_out(double* __restrict__ stage, uint64_t j0, uint32_t cnt, Shard sh, Tab tb) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cnt) return;
  uint32_t jl;
  if (sh.place((uint32_t)(j0 + t), &jl)) stage[t] = (double)tb.w[(size_t)jl * tb.ws];
}

static __global__ void k_fetch_rows(const uint32_t* __restrict__ ids, uint32_t count, int k, Shard sh, Tab tb,
                             double* __restrict__ w_out, double* __restrict__ v_out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)count * (uint64_t)(k + 1)) return;
  const uint32_t i = (uint32_t)(t / (k + 1)); const int f = (int)(t % (k + 1));
  const size_t jl = sh.local(ids[i]);
  if (f == k) w_out[i] = (double)tb.w[jl * tb.ws];
  else v_out[(size_t)i * k + f] = (double)tb.V[jl * tb.rs + f];
}

// counter-hash helpers: identical definitions in oracle/fm_oracle.c (fmo_synth_id, ...)
__host__ __device__ __forceinline__ uint64_t synth_key(uint64_t seed, uint64_t row, uint32_t field) {
  return mix64(seed + 0x9E3779B97F4A7C15ULL * (row + 1) + 0xC2B2AE3D27D4EB4FULL * ((uint64_t)field + 1));
}

static __global__ void k_init_params(Tab tb, uint64_t n_local, int k, int KP,
                              Shard sh, float mean, double stdev, uint64_t seed) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = t; i < n_local * (uint64_t)tb.rs; i += stride) {    // whole row incl. padding (and w if co-located)
    const uint64_t jl = i / tb.rs; const int f = (int)(i % tb.rs);
    const uint64_t j = sh.global(jl);
    float val = 0.f;
    if (f < k) {
      const uint64_t hsh = mix64(seed ^ (j * 0x9E3779B97F4A7C15ULL + (uint64_t)f * 0xD6E8FEB86659FD93ULL + 0x1234567ULL));
      const double u = (double)(hsh >> 11) * (1.0 / 9007199254740992.0);
      val = mean + (float)(stdev * (2.0 * u - 1.0) * 1.7320508075688772);
    }
    tb.V[i] = val;
  }
  if (tb.ws == 1) for (uint64_t i = t; i < n_local; i += stride) tb.w[i] = 0.f;
}

// synthetic rows: field t owns ids [t*fs,(t+1)*fs); this shard keeps the ids it owns (Shard)
// pass 1 (count==true): row_cnt[r] = #kept entries; pass 2: fill at row_ptr[r]
// shape 0: uniform within field.  shape 1 ("Criteo-shaped", BASELINE configs[2]): fields 0..12 hold 100 ids each, drawn with a
// geometric profile (a binned numeric column: bin = floor(12 * Exp(1)), capped at 99: the first bin has 8 % of the rows); the
// other fields share the remaining ids equally (fs2 each) and draw Zipf(1.05) through the inverse CDF of the continuous
// power law on [1, fs2 + 1): id 0 of such a field is met by 9 % of the rows at fs2 = 1.27e6.
constexpr uint32_t SYNTH_DENSE_FIELDS = 13, SYNTH_DENSE_IDS = 100;
__device__ __forceinline__ uint32_t synth_id(uint64_t hsh, uint32_t t, uint32_t fs, uint32_t shape) {
  if (shape == 0) return t * fs + (uint32_t)(((hsh >> 32) * (uint64_t)fs) >> 32);
  const double u = (double)(hsh >> 11) * (1.0 / 9007199254740992.0);          // [0, 1)
  if (t < SYNTH_DENSE_FIELDS) {
    const uint32_t bin = (uint32_t)fmin(99.0, floor(-12.0 * log(1.0 - u)));
    return t * SYNTH_DENSE_IDS + bin;
  }
  const double e = 1.0 - 1.05;                                                  // x = (1 + u ((fs + 1)^e - 1))^(1 / e)
  const double x = pow(1.0 + u * (pow((double)fs + 1.0, e) - 1.0), 1.0 / e);
  const uint32_t off = (uint32_t)fmin((double)(fs - 1), fmax(0.0, floor(x) - 1.0));
  return SYNTH_DENSE_FIELDS * SYNTH_DENSE_IDS + (t - SYNTH_DENSE_FIELDS) * fs + off;
}
static __global__ void k_synth(uint64_t seed, uint64_t row0, uint32_t n_rows, uint32_t nnz, uint32_t fs, uint32_t shape, Shard sh,
                        uint32_t* __restrict__ row_cnt, const uint64_t* __restrict__ row_ptr,
                        Entry* __restrict__ ent, float* __restrict__ target) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  uint64_t pos = row_ptr ? row_ptr[r] : 0;
  uint32_t c = 0;
  for (uint32_t t = 0; t < nnz; t++) {
    const uint32_t id = synth_id(synth_key(seed, row0 + r, t), t, fs, shape);
    uint32_t jl;
    if (sh.place(id, &jl)) {
      if (ent) { ent[pos].id = jl; ent[pos].value = 1.0f; pos++; }
      c++;
    }
  }
  if (row_cnt) row_cnt[r] = c;
  if (target) {
    const uint64_t hb = synth_key(seed, row0 + r, 0xFFFFFFFFu);
    target[r] = (shape == 0 ? (hb & 1) : ((hb & 3) == 0)) ? 1.0f : -1.0f;      // Criteo-shaped: one positive in four
  }
}

// ---- a feature shard's view of uploaded rows, filtered on the device (fmx_group_upload_rows): the entries whose feature the shard
// owns, ids rewritten to local rows.  pass 1 (ent_out == nullptr): row_cnt[r] = kept entries (+ the longest kept row);
// pass 2: fill at out_ptr[r].  `bad` is raised for an id >= num_attribute (the reference asserts it, fm_model.h:112).
static __global__ void __launch_bounds__(256)
k_shard_rows(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, uint32_t n_rows, uint64_t n, Shard sh,
             uint32_t* __restrict__ row_cnt, uint32_t* __restrict__ max_row, uint32_t* __restrict__ bad,
             const uint64_t* __restrict__ out_ptr, Entry* __restrict__ ent_out) {
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += gridDim.x * blockDim.x) {
    uint32_t c = 0;
    uint64_t pos = out_ptr ? out_ptr[r] : 0;
    for (uint64_t i = row_ptr[r]; i < row_ptr[r + 1]; i++) {
      const Entry e = ent[i];
      if ((uint64_t)e.id >= n) { if (bad) atomicOr(bad, 1u); continue; }
      uint32_t jl;
      if (sh.place(e.id, &jl)) {
        if (ent_out) { Entry o; o.id = jl; o.value = e.value; ent_out[pos++] = o; }
        c++;
      }
    }
    if (row_cnt) { row_cnt[r] = c; if (c) atomicMax(max_row, c); }
  }
}

// ---- collision mass of a row set: C = mean over pairs of DIFFERENT rows of sum_j |x_ej| |x_e'j|
//      = (sum_j (sum_rows |x_j|)^2 - sum_entries x^2) / (N (N - 1))      (fmx_sgd_opts::batch, fmx_sgd_batch_info) ------------
// hist[id mod M] += |x| per entry (M = table size, or 2^27 buckets for larger tables: folding can only raise C, i.e. cut the batch more)
// (sumsq: sum over the entries of x^2 -- what a row shares with ITSELF, taken out of the pair statistic afterwards)
static __global__ void __launch_bounds__(256)
k_coll_hist(const Entry* __restrict__ ent, uint64_t nnz, uint32_t M, double* __restrict__ hist, double* __restrict__ sumsq) {
  // fp64 buckets (native atomic add on gfx950): an fp32 bucket of unit values stops growing at 2^24 occurrences -- a bias-like column of a
  // slot of >= 16.7 M rows would then UNDER-estimate C, i.e. the very divergence the cut is there to prevent (round-3 advisor finding);
  // with 53 bits the sums of the usual 0/1 and small-integer values are exact, so C is also the same from run to run
  double own = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * blockDim.x) {
    const Entry e = ent[i];
    unsafeAtomicAdd(hist + (e.id % M), (double)fabsf(e.value));
    own += (double)e.value * (double)e.value;
  }
  own = wave_sum_d(own);
  if ((threadIdx.x & 63u) == 0 && own != 0.0) unsafeAtomicAdd(sumsq, own);
}
static __global__ void __launch_bounds__(256)
k_coll_sumsq(const double* __restrict__ hist, uint32_t M, double* __restrict__ out) {
  double a = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (uint64_t)gridDim.x * blockDim.x) {
    const double f = hist[i];
    a += f * f;
  }
  a = wave_sum_d(a);
  if ((threadIdx.x & 63u) == 0 && a != 0.0) unsafeAtomicAdd(out, a);
}

}  // namespace fmx

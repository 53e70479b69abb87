// k_small_one: a SMALL batch of the minibatch rule (FMX_APPLY_FUSED; what the stability cut leaves of rows with frequent features: 512 rows of
// Criteo-shaped data) as ONE launch across all dies -- examples, deferred features and the bias recurrence, where the in-stream schedule takes
// two (k_fused<FUSED_EXACT>, then k_apply_seg_scan) and pays the launch boundary plus the drain of the first between them.
//
// Three roles by workgroup index:
//   examples  (the first n_ex_wg workgroups, one wavefront per example): k_fused<FUSED_EXACT>'s body -- rows in registers, sums, multiplier,
//             write-back of the features that occur once in the batch -- and what the others need goes out as it is known: {tag, rest_e} as one
//             8-byte agent-scope store (the recurrence polls it), then S_e (a plain 256-byte store), a release fence, and {tag, mult_e} (the
//             owners poll it).  Nothing an example waits for.
//   owners    (one wavefront per deferred feature): the feature's row and its occurrence list are asked for at once (batch-start values,
//             static tables); every lane polls the multiplier slot of ITS occurrence (the poll that succeeds is the load), then the S_e
//             rows are read past this die's caches, TL at a time, their tags checked (read again if an element is not there yet), and
//             summed in occurrence order; one owner per row, written once.
//             An example publishes its multiplier only after it has gathered all its rows, so when an owner has seen every occurrence's
//             tag nobody will read the batch-start row again: the owner's store cannot overtake a reader.
//   recurrence (the last workgroup, one wavefront): at bias lag >= 2 the PREVIOUS batch's (its rest_e are complete and what it writes is read
//             two launches on: off the critical path; the epoch's last launch runs its own batch's too); at lag 1 this batch's, polled into
//             the LDS as the examples publish -- the bias ring slot it then writes is the one this batch's examples read, and they read it
//             before they publish anything.
// Waits only go examples <- {owners, recurrence}: no cycle, and every poll is bounded (RUN_ERR_EXCHANGE: the epoch fails loudly, the handle
// goes back to two launches).  The exchange is the tagged-slot one of k_run_fused (fmx_seq_kernels.h): no read-modify-write, no barrier.
// Same rule, same numbers to fp32 rounding as the two launches (tests/test_gpu_small_one.py).
#pragma once

namespace fmx {

#ifndef FMX_SMALL_NT
#define FMX_SMALL_NT 3                                                  // bit 1: the examples' row loads non-temporal, bit 2: their row stores
#endif
constexpr uint32_t SMALL_ONE_MAX = 1024;                              // examples per batch (the recurrence holds the batch in registers)
struct SmallSync { unsigned long long* mslot; unsigned long long* rslot; const unsigned long long* rslot_prev; uint32_t tag, tag_prev; uint32_t* err; uint32_t spins;
                   unsigned long long* trace; };   // trace (FMX_SMALL_TRACE=<file>, one batch of the epoch): wall_clock64 time stamps (ns on this part), [0] first / [1] last example started,
                                                   // [2] last rows gathered, [3] last multiplier published, [4] last example done, [5] first owner started,
                                                   // [6] last owner saw its tags, [7] last owner holds its S_e rows, [8] last owner done, [9] recurrence done, [10] last owner started,
                                                   // [11] last owner has its entry list
__device__ __forceinline__ void trace_min(unsigned long long* t, int i) { if (t && (threadIdx.x & 63u) == 0) atomicMin(t + i, (unsigned long long)wall_clock64()); }
__device__ __forceinline__ void trace_max(unsigned long long* t, int i) { if (t && (threadIdx.x & 63u) == 0) atomicMax(t + i, (unsigned long long)wall_clock64()); }

__device__ __forceinline__ bool slot_wait(const unsigned long long* p, uint32_t tag, uint32_t spins, uint32_t& lo) {
  unsigned long long u = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (uint32_t t = 0; (uint32_t)(u >> 32) != tag && t < spins; t++) {
    __builtin_amdgcn_s_sleep(2);
    u = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  lo = (uint32_t)u;
  return (uint32_t)(u >> 32) == tag;
}
__device__ __forceinline__ void slot_put(unsigned long long* p, uint32_t tag, float v) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int KP, int ZR>
__global__ void __launch_bounds__(256)
k_small_one(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, const float* __restrict__ target, uint64_t row0, uint32_t n_rows,
            const Tab tb, Hyper h, const double* __restrict__ w0_ptr, const uint64_t* __restrict__ cmask, float* __restrict__ S_out,
            uint32_t fixed_nnz, const SegWork sw, const ScanSmall sc_prev, const ScanSmall sc, const SmallSync sy, uint32_t n_ex_wg,
            const uint64_t* __restrict__ lmask, float* __restrict__ wside) {
  // wside != nullptr (FMX_FLAG_KEEP_WSIDE): the slot's weight side stream is kept current exactly as k_fused<FUSED_EXACT> keeps it
  // KP < 64 (k <= 32): EPI = 64 / KP rows per wave-wide load -- lane group g = lane / KP holds entry t * EPI + g of row slot t (k_fused's layout)
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR, EPI = Map<KP>::EPI;
  __shared__ float s_rest[SMALL_ONE_MAX];
  const uint32_t lane = threadIdx.x & 63u, g = lane / LPR, f = lane % LPR;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (blockIdx.x < n_ex_wg) {
    // ---------------------------------------------------------------- an example
    const uint32_t e = blockIdx.x * 4u + wv;
    if (e >= n_rows) return;
    const float w0s = h.k0 ? (float)(*w0_ptr) : 0.f;                 // (before anything is published: the recurrence rewrites this slot)
    trace_min(sy.trace, 0); trace_max(sy.trace, 1);
    const uint64_t a = fixed_nnz ? (row0 + e) * (uint64_t)fixed_nnz : row_ptr[row0 + e];
    const uint32_t size = fixed_nnz ? fixed_nnz : (uint32_t)(row_ptr[row0 + e + 1] - a);
    const Entry* __restrict__ row = ent + a;
    const float y = target[row0 + e];
    const uint64_t cm = cmask[row0 + e];
    const uint64_t lm = wside ? lmask[row0 + e] : 0ull;
    if (size <= (uint32_t)(ZR * EPI) && size <= 64u) {
      Entry en; en.id = 0; en.value = 0.f;
      float wl = 0.f;
      if (lane < size) {
        en = load_stream8(row + lane);
        if (h.k1) wl = load_w(tb.w + (size_t)en.id * tb.ws);
      }
      float vr[ZR][VEC];
#pragma unroll
      for (int t = 0; t < ZR; t++) {                                 // (cross-lane reads with every lane enabled: outside the guards)
        const uint32_t idx = (uint32_t)t * EPI + g;
        const uint32_t id = bcast_u32<EPI>(en.id, idx & 63u);
        if (idx < size) {
          row_ld<VEC, (FMX_SMALL_NT & 1)>(tb, (size_t)id, f * VEC, vr[t]);
        } else {
#pragma unroll
          for (int v = 0; v < VEC; v++) vr[t][v] = 0.f;
        }
      }
      float sum[VEC]; float sq = 0.f;                               // fm_model.h:116-125
#pragma unroll
      for (int v = 0; v < VEC; v++) sum[v] = 0.f;
#pragma unroll
      for (int t = 0; t < ZR; t++) {
        const uint32_t idx = (uint32_t)t * EPI + g;
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
      float part = wl * en.value - 0.5f * sq;
      if (lane < LPR) {
#pragma unroll
        for (int v = 0; v < VEC; v++) part = fmaf(0.5f * sum[v], sum[v], part);
      }
      const float rest = wave_sum_dpp(part);
      trace_max(sy.trace, 2);
      if (lane == 0) slot_put(sy.rslot + e, sy.tag, rest);
      const float mult = multiplier(h, w0s + rest, y);
      // what the owners need, BEFORE the example's own updates and without waiting for anything: every element of S_e as a self-validating
      // 8-byte {tag, value} store, then the tagged multiplier.  (A release fence per example writes back the die's whole L2: 22.6 us per
      // batch; plain write-through stores + s_waitcnt for their acknowledgement before the multiplier: 13.1-13.9.)
      if (cm != 0) {
        unsigned long long* Sx = reinterpret_cast<unsigned long long*>(S_out);
        if (lane < LPR) {
#pragma unroll
          for (int v = 0; v < VEC; v++)
            __hip_atomic_store(Sx + (size_t)e * KP + lane * VEC + v, ((unsigned long long)sy.tag << 32) | (unsigned long long)__float_as_uint(sum[v]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) slot_put(sy.mslot + e, sy.tag, mult);
        trace_max(sy.trace, 3);
      }
      float w_keep = __builtin_nanf("");
      if (h.k1 && lane < size && !((cm >> lane) & 1ull)) {            // fm_sgd.h:38-43
        w_keep = wl - h.lr * (mult * en.value + h.regw * wl);
        tb.w[(size_t)en.id * tb.ws] = w_keep;
      }
      if (wside && lane < size) __builtin_nontemporal_store(((lm >> lane) & 1ull) ? w_keep : __builtin_nanf(""), wside + a + lane);
#pragma unroll
      for (int t = 0; t < ZR; t++) {                                 // fm_sgd.h:44-50 on the register-resident rows
        const uint32_t idx = (uint32_t)t * EPI + g;
        const uint32_t id = bcast_u32<EPI>(en.id, idx & 63u);
        const float x = bcast_f32<EPI>(en.value, idx & 63u);
        if (idx < size && !((cm >> (idx & 63u)) & 1ull) && f * VEC < tb.rs) {
          float* pv = tb.V + (size_t)id * tb.rs + f * VEC;
          float nv[VEC];
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            const float vv = vr[t][v];
            const float grad = sum[v] * x - vv * x * x;
            nv[v] = vv - h.lr * (mult * grad + h.regv * vv);
          }
          store_row<VEC, (FMX_SMALL_NT & 2)>(pv, nv);
        }
      }
      trace_max(sy.trace, 4);
    } else {                                                         // a row beyond the register path: deferred as a whole (cm = all ones)
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
      if (lane == 0) slot_put(sy.rslot + e, sy.tag, rest);
      const float mult = multiplier(h, w0s + rest, y);
      unsigned long long* Sx = reinterpret_cast<unsigned long long*>(S_out);
      if (lane < LPR) {
#pragma unroll
        for (int v = 0; v < VEC; v++)
          __hip_atomic_store(Sx + (size_t)e * KP + lane * VEC + v, ((unsigned long long)sy.tag << 32) | (unsigned long long)__float_as_uint(sum[v]),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) slot_put(sy.mslot + e, sy.tag, mult);
    }
    return;
  }
  if (blockIdx.x == gridDim.x - 1) {
    // ---------------------------------------------------------------- the bias recurrence
    // bias lag >= 2: the recurrence of the PREVIOUS batch runs here -- its rest_e are complete, nothing to wait for, and what it writes is
    // read two launches on -- so the recurrence is off this launch's critical path; the epoch's last launch also runs its own batch's.
    // Bias lag 1: this batch's, polled as the examples publish.
    if (wv != 0 || !h.k0) return;
    if (sc_prev.n_rows) {
      bool ok = true;
      for (uint32_t i = lane; i < sc_prev.n_rows; i += 64u) {
        uint32_t lo;
        ok &= slot_wait(sy.rslot_prev + i, sy.tag_prev, sy.spins, lo);
        s_rest[i] = __uint_as_float(lo);
      }
      if (__any(!ok)) { if (lane == 0) atomicOr(sy.err, RUN_ERR_EXCHANGE); return; }
      ScanSmall s2 = sc_prev;
      s2.rest = s_rest;
      scan_small<false>(s2, h);
      if (sc.n_rows) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");   // (the bias it left is what the next one starts from)
    }
    if (sc.n_rows) {
      bool ok = true;
      for (uint32_t i = lane; i < sc.n_rows; i += 64u) {
        uint32_t lo;
        ok &= slot_wait(sy.rslot + i, sy.tag, sy.spins, lo);
        s_rest[i] = __uint_as_float(lo);
      }
      if (__any(!ok)) { if (lane == 0) atomicOr(sy.err, RUN_ERR_EXCHANGE); return; }
      ScanSmall s2 = sc;
      s2.rest = s_rest;
      scan_small<false>(s2, h);
    }
    trace_max(sy.trace, 9);
    return;
  }
  // ------------------------------------------------------------------ an owner of deferred features
  const uint32_t n_own_waves = (gridDim.x - 1u - n_ex_wg) * 4u;
  for (uint32_t s = (blockIdx.x - n_ex_wg) * 4u + wv; s < sw.nseg; s += n_own_waves) {
    trace_min(sy.trace, 5); trace_max(sy.trace, 10);
    const uint4 d0 = reinterpret_cast<const uint4*>(sw.cdesc + s)[0];
    const uint32_t j = __builtin_amdgcn_readfirstlane(d0.x), a = __builtin_amdgcn_readfirstlane(d0.y), b = __builtin_amdgcn_readfirstlane(d0.z);
    const bool act = lane < LPR;                                     // (KP < 64: the row's lanes; one row per wave-wide load all the same)
    float v0[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) v0[v] = 0.f;
    if (act) row_ld<VEC, 8>(tb, (size_t)j, lane * VEC, v0);
    float wv0 = 0.f;
    if (h.k1 && lane == 0) wv0 = tb.w[(size_t)j * tb.ws];
    float G[VEC]; float A = 0.f, Gw = 0.f;
#pragma unroll
    for (int v = 0; v < VEC; v++) G[v] = 0.f;
    constexpr int TL = (VEC == 1) ? 16 : 8;                          // S_e rows in flight per round
    const unsigned long long* Sx = reinterpret_cast<const unsigned long long*>(sw.S);
    for (uint32_t base = a; base < b; base += 64u) {
      const uint32_t cc = min(64u, b - base);
      TEntry te; te.e = 0; te.x = 0.f; float tm = 0.f;
      bool ok = true;
      if (lane < cc) te = load_stream8(sw.t_ent + base + lane);
      if (sy.trace && __any(te.e == 0xFFFFFFFFu)) return;            // (forces the entry list in before the stamp)
      trace_max(sy.trace, 11);
      if (lane < cc) {
        uint32_t lo;
        ok = slot_wait(sy.mslot + te.e, sy.tag, sy.spins, lo);
        tm = __uint_as_float(lo);
      }
      if (__any(!ok)) { if (lane == 0) atomicOr(sy.err, RUN_ERR_EXCHANGE); return; }   // (the feature takes no step)
      trace_max(sy.trace, 6);
      for (uint32_t q0 = 0; q0 < cc; q0 += TL) {
        // the S_e elements validate themselves: they left their example before its multiplier did, so they are normally here; else again
        unsigned long long u[TL][VEC];
        bool stale = false;
        for (uint32_t t = 0; t <= sy.spins; t++) {
          stale = false;
#pragma unroll
          for (int q = 0; q < TL; q++) {
            const uint32_t e2 = bcast_u32<1>(te.e, (q0 + q) & 63u);
#pragma unroll
            for (int v = 0; v < VEC; v++) {
              u[q][v] = (q0 + q < cc && act) ? __hip_atomic_load(Sx + (size_t)e2 * KP + lane * VEC + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)sy.tag << 32);
              stale |= (uint32_t)(u[q][v] >> 32) != sy.tag;
            }
          }
          if (!__any(stale)) break;
          __builtin_amdgcn_s_sleep(2);
        }
        if (__any(stale)) { if (lane == 0) atomicOr(sy.err, RUN_ERR_EXCHANGE); return; }
        trace_max(sy.trace, 7);
#pragma unroll
        for (int q = 0; q < TL; q++) {
          const float x2 = bcast_f32<1>(te.x, (q0 + q) & 63u), m2 = bcast_f32<1>(tm, (q0 + q) & 63u);
          if (q0 + q < cc) {
            const float mx2 = m2 * x2;
#pragma unroll
            for (int v = 0; v < VEC; v++) G[v] = fmaf(mx2, __uint_as_float((uint32_t)u[q][v]), G[v]);
            A = fmaf(mx2, x2, A); Gw += mx2;
          }
        }
      }
    }
    const float nocc = (float)(b - a);
    float nv[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) {
      const float vv = v0[v];
      nv[v] = vv - h.lr * (G[v] - vv * A + nocc * h.regv * vv);
    }
    if (act) row_st<VEC, 8>(tb, (size_t)j, lane * VEC, nv);
    if (h.k1 && lane == 0) tb.w[(size_t)j * tb.ws] = wv0 - h.lr * (Gw + nocc * h.regw * wv0);
    trace_max(sy.trace, 8);
  }
}

}  // namespace fmx

// XCD-resident epoch: the one-pass MINIBATCH rule (k_fused<EXACT> + k_apply_seg_scan, fmx_kernels.h) for SMALL batches -- what the stability
// cut leaves of rows with frequent features (BASELINE configs[2]: 512 rows per batch at learn_rate 0.01) -- as ONE launch per epoch whose
// workgroups all sit on ONE accelerator complex die.
//
// Why one XCD.  A 512-row batch is not bandwidth (5 MB of traffic = 1 us of HBM) but a chain of dependent round trips:
//   examples -> (their sums and multipliers) -> frequent features -> (their new rows) -> the examples of the next batch
// (fm_learn_sgd_element.h:56-67 is this chain with a batch of one).  As two chip-wide launches per batch every hop is a kernel boundary plus
// HBM-latency gathers: 9.1 + 7.3 us per batch, 0.08 of the byte roofline (round-5 verdict).  Inside one launch a chip-wide hop is no
// cheaper (a grid barrier costs 4.1 us: the per-XCD L2s are not coherent, so every hop is a write-back + invalidate through the fabric).
// Inside ONE XCD the L2 is the coherence point: a plain store is visible to every CU of the die as soon as the L2 acknowledged it, a load
// that bypasses the per-CU L1 (sc1) reads it back at L2 latency, and a barrier is one 4-byte store + one 512-byte poll in that L2.  The
// frequent rows (the 1300 ids of the 13 dense fields, the heads of the Zipf fields: < 1 MB) never leave the die's 4 MiB L2.  The die's
// share of the fabric (1/8) still carries the cold rows of a batch (1.5 MB in, 1.5 MB out) in ~4 us.
//
// Protocol (no assumption about workgroup placement beyond "every workgroup of the launch is resident", which the host guarantees by
// sizing the grid from the occupancy query):
//   * every workgroup reads HW_REG_XCC_ID; the die of workgroup 0 is the target; workgroups on it register as MEMBERS (device-scope
//     atomics, once per launch), every workgroup counts itself as started, everyone else exits.  Members wait until all workgroups have
//     started -- then the member count is final.  A bounded wait; if it runs out NOTHING has been touched and the host takes the
//     two-launch path (one consensus word decides for all members).
//   * work is dealt statically: example e of a batch -> item wavefront e mod NW, deferred segment s -> item wavefront s mod NW; the last
//     wavefront of the last member only runs the bias recurrence (scan_small).
//   * every mutable word (V rows, w, S_e, multipliers, rest, the bias ring) is read with sc1 loads and written with plain stores; a
//     wavefront drains its stores (s_waitcnt vmcnt(0)) before its workgroup arrives at a barrier; barrier = per-member generation
//     words in one 1 KiB array, polled lane-parallel by the first wavefront of every member.
// The arithmetic is k_fused<EXACT>'s and apply_seg_block's, operation for operation (same order of additions): the parity tests of the
// two-launch path hold for this one unchanged.
#pragma once

namespace fmx {

constexpr uint32_t XCD_MAX_MEMBERS = 256;
constexpr uint32_t XCD_CTL_WORDS = 16;             // [0] target die + 1, [1] members, [2] started, [3] abort, [4] decision (1 go, 2 do not start)
constexpr uint32_t XCD_ERR_BARRIER = 16u;          // handle error word: a barrier of the XCD-resident epoch ran into its bound

struct XcdSync {
  unsigned* ctl;        // [XCD_CTL_WORDS], zeroed on the stream before the launch
  unsigned* flags;      // [XCD_MAX_MEMBERS] barrier generation every member has reached, zeroed with ctl
  uint32_t* err;        // the handle's error word
  uint32_t spins;       // bound of every wait (polls)
};

struct XcdEpoch {
  const Entry* ent; const uint64_t* row_ptr; const float* target; const uint64_t* cmask;
  uint32_t fixed_nnz, n_rows, B, n_batch, d, chunk, Bc;
  const uint32_t* cbatch;          // [n_batch + 1] first deferred segment of every batch (device copy of Slot::cbatch)
  const uint64_t* batch_base;      // [n_batch + 1] first entry of every batch in t_ent
  const TEntry* t_ent; const CDesc* cdesc;
  float* S; float* mult; float* rest;          // [B][KP], [B], [d][Bc]
  double* w0_ring;                              // [d] (fmx_sgd.hip: slot (b + 1) % d holds the bias after the recurrence of batch b)
  uint32_t flags;                               // experiments: 1 touch every row of the next example (not only the ones it writes), 2 no stream touches
  unsigned long long* trace; uint32_t trace_batches;   // not nullptr: 8 time stamps (s_memrealtime, 10 ns) per batch of member 0's first wavefront
};

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xfu;
}
template <int VEC, bool STREAM = false> __device__ __forceinline__ void ld_l2_vec(const float* p, float (&out)[VEC]) {
  static_assert(VEC == 1 || VEC == 2, "rows of 64 or 128 floats");
  if constexpr (STREAM) {
    if constexpr (VEC == 1) out[0] = __builtin_nontemporal_load(p);
    else {
      typedef float v2f __attribute__((ext_vector_type(2)));
      const v2f t = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(p)); out[0] = t.x; out[1] = t.y;
    }
  } else if constexpr (VEC == 1) out[0] = ld_l2(p);
  else {
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[0] = __uint_as_float((uint32_t)u); out[1] = __uint_as_float((uint32_t)(u >> 32));
  }
}
// a parameter row through the die's L2 (rows are tb.rs floats: the lanes beyond take no part, fmx_kernels.h row_ld)
template <int VEC> __device__ __forceinline__ void xcd_row_ld(const Tab& tb, size_t id, float (&out)[VEC]) {
  const uint32_t off = (threadIdx.x & 63u) * VEC;
  if (off < tb.rs) ld_l2_vec<VEC, false>(tb.V + id * tb.rs + off, out);
  else {
#pragma unroll
    for (int v = 0; v < VEC; v++) out[v] = 0.f;
  }
}
template <int VEC> __device__ __forceinline__ void xcd_row_st(const Tab& tb, size_t id, const float (&in)[VEC]) {
  const uint32_t off = (threadIdx.x & 63u) * VEC;
  if (off < tb.rs) store_vec<VEC>(tb.V + id * tb.rs + off, in);
}
// a plain 4-byte store (stays in the die's L2; an agent-scope atomic store would drop the line to the fabric)
__device__ __forceinline__ void st_plain(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory"); }

// the barrier in two halves, so that a wavefront can ask for what it needs NEXT (read-only streams, rows nobody writes in the running phase)
// in the barrier's shadow.  arrive: every store of this workgroup is in the L2, then its generation word says so.
__device__ __forceinline__ void xcd_arrive(const XcdSync& sy, uint32_t idx, uint32_t gen) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) st_plain(sy.flags + idx, gen);
}
// wait: the first wavefront of the workgroup polls every member's word (one 1 KiB array, lane-parallel); false: the epoch is given up
__device__ __forceinline__ bool xcd_wait(const XcdSync& sy, uint32_t P, uint32_t gen, uint32_t* s_ok) {
  if (threadIdx.x < 64u) {
    const uint32_t lane = threadIdx.x;
    bool ok = false;
    for (uint32_t t = 0; t < sy.spins; t++) {
      bool behind = false;
      unsigned fl[XCD_MAX_MEMBERS / 64u];
#pragma unroll
      for (uint32_t i = 0; i < XCD_MAX_MEMBERS / 64u; i++) fl[i] = ld_l2(sy.flags + lane + 64u * i);     // (the array is XCD_MAX_MEMBERS words whatever P)
#pragma unroll
      for (uint32_t i = 0; i < XCD_MAX_MEMBERS / 64u; i++)
        if (lane + 64u * i < P && (int)(fl[i] - gen) < 0) behind = true;
      if (__ballot(behind) == 0ull) { ok = true; break; }
      if ((t & 255u) == 255u && ld_dev(sy.ctl + 3) != 0u) break;        // somebody gave up
      __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) {
      if (!ok) { atomicOr(sy.err, XCD_ERR_BARRIER); __hip_atomic_store(sy.ctl + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      *s_ok = ok ? 1u : 0u;
    }
  }
  __syncthreads();
  return *s_ok != 0u;
}

// what a wavefront knows of an example before it may touch a parameter: where its entries are, its label, which entries are deferred, the
// entries themselves (lane i holds entry i) -- all read-only, so it is asked for a barrier ahead
struct XcdRow { uint32_t e, size; float y; uint64_t cm; Entry en; bool valid; };
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ XcdRow xcd_row(const XcdEpoch& ep, uint64_t row0, uint32_t e, uint32_t nb) {
  XcdRow r; r.e = e; r.size = 0; r.y = 0.f; r.cm = 0; r.en.id = 0; r.en.value = 0.f; r.valid = e < nb;
  if (r.valid) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t a = ep.fixed_nnz ? (row0 + e) * (uint64_t)ep.fixed_nnz : ep.row_ptr[row0 + e];
    r.size = uni(ep.fixed_nnz ? ep.fixed_nnz : (uint32_t)(ep.row_ptr[row0 + e + 1] - a));      // (wave-uniform by construction: say so)
    r.y = __uint_as_float(uni(__float_as_uint(ep.target[row0 + e])));
    const uint64_t cm = ep.cmask[row0 + e];
    r.cm = ((uint64_t)uni((uint32_t)(cm >> 32)) << 32) | uni((uint32_t)cm);
    if (lane < r.size) r.en = ep.ent[a + lane];
  }
  return r;
}

// a lane's value as a scalar (wave-uniform lane index)
template <int T> __device__ __forceinline__ uint32_t lane_val(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, T); }
// the same value behind an empty statement: the compiler would otherwise keep the 2 x 40 scalars it broadcast for the gathers alive until
// the stores (spilled into vector lanes); broadcasts of the copy are separate, short-lived values
__device__ __forceinline__ uint32_t opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
// one example of the batch on one wavefront: k_fused<KP, ZR, FUSED_EXACT>'s register path (fm_model.h:116-125, fm_sgd.h:38-50) with
// L2-served loads; the host guarantees size <= ZR (Slot::max_row).
// ONES (every value of the slot is 1.0f -- one-hot data): no multiplication by x, and fm_sgd.h:47-49's
//   v - lr (mult (sum - v) + regv v)   is evaluated as   v (1 + lr (mult - regv)) - lr mult sum   -- one fused multiply-add per element
// with two per-example constants (a few ulp from the other form; one XCD has an eighth of the chip's vector units, they are the budget here)
template <int KP, int ZR, bool ONES, int T>
__device__ __forceinline__ void xcd_rows_store(const Tab& tb, const Hyper& h, uint32_t ids, float xs, uint32_t size, uint64_t cm, const float (&vr)[ZR][Map<KP>::VEC],
                                               const float (&sum)[Map<KP>::VEC], const float (&cbs)[Map<KP>::VEC], float ca, float mult) {
  constexpr int VEC = Map<KP>::VEC;
  if constexpr (T < ZR) {
    if ((uint32_t)T < size && !((cm >> (uint32_t)T) & 1ull)) {
      const uint32_t lane = threadIdx.x & 63u;
      const uint32_t id = lane_val<T>(ids);
      float nv[VEC];
      if constexpr (ONES) {
#pragma unroll
        for (int v = 0; v < VEC; v++) nv[v] = fmaf(vr[T][v], ca, cbs[v]);
      } else {
        const float x = __uint_as_float(lane_val<T>(__float_as_uint(xs)));
#pragma unroll
        for (int v = 0; v < VEC; v++) {
          const float vv = vr[T][v];
          const float grad = sum[v] * x - vv * x * x;
          nv[v] = vv + (-h.lr * (mult * grad + h.regv * vv));
        }
      }
      xcd_row_st<VEC>(tb, (size_t)id, nv);
    }
    xcd_rows_store<KP, ZR, ONES, T + 1>(tb, h, ids, xs, size, cm, vr, sum, cbs, ca, mult);
  }
}
template <int KP, int ZR, bool ONES, int T>
__device__ __forceinline__ void xcd_rows_load(const Tab& tb, uint32_t ids, uint32_t size, uint64_t cm, float (&vr)[ZR][Map<KP>::VEC]) {
  constexpr int VEC = Map<KP>::VEC;
  if constexpr (T < ZR) {
    if ((uint32_t)T < size) {
      const uint32_t id = lane_val<T>(ids);
      xcd_row_ld<VEC>(tb, (size_t)id, vr[T]);
    } else {
#pragma unroll
      for (int v = 0; v < VEC; v++) vr[T][v] = 0.f;
    }
    xcd_rows_load<KP, ZR, ONES, T + 1>(tb, ids, size, cm, vr);
  }
}
template <int KP, int ZR, bool ONES, int T>
__device__ __forceinline__ void xcd_rows_sum(float xs, uint32_t size, const float (&vr)[ZR][Map<KP>::VEC], float (&sum)[Map<KP>::VEC], float& sq) {
  constexpr int VEC = Map<KP>::VEC;
  if constexpr (T < ZR) {
    if constexpr (ONES) {                                          // (row slots beyond the example hold zeros)
#pragma unroll
      for (int v = 0; v < VEC; v++) { sum[v] += vr[T][v]; sq = fmaf(vr[T][v], vr[T][v], sq); }
    } else {
      float x = __uint_as_float(lane_val<T>(__float_as_uint(xs)));
      if ((uint32_t)T >= size) x = 0.f;
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        const float dd = vr[T][v] * x;
        sum[v] += dd;
        sq = fmaf(dd, dd, sq);
      }
    }
    xcd_rows_sum<KP, ZR, ONES, T + 1>(xs, size, vr, sum, sq);
  }
}
template <int KP, int ZR, bool ONES>
__device__ __forceinline__ void xcd_example(const XcdEpoch& ep, const Tab& tb, const Hyper& h, const XcdRow& r, float w0s, float* __restrict__ rest_out,
                                            unsigned long long* tr = nullptr) {
  constexpr int VEC = Map<KP>::VEC;
  static_assert(Map<KP>::EPI == 1, "one row per wave-wide load");
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t size = r.size, e = r.e;
  // (the entry registers pass an empty statement once: they were loaded a barrier ago, and as "values of a load" the compiler would wait
  //  for EVERY outstanding load -- the row gathers -- at each of the branches below before it broadcasts a lane of them)
  Entry en; en.id = opaque(r.en.id); en.value = __uint_as_float(opaque(__float_as_uint(r.en.value)));
  const uint64_t cm = r.cm;
  float wv = 0.f;
  if (h.k1 && lane < size) wv = ld_l2(tb.w + (size_t)en.id * tb.ws);
  float vr[ZR][VEC];
  xcd_rows_load<KP, ZR, ONES, 0>(tb, en.id, size, cm, vr);
  if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) tr[8] = wall_clock64(); }     // rows and weights are here
  float sum[VEC]; float sq = 0.f;
#pragma unroll
  for (int v = 0; v < VEC; v++) sum[v] = 0.f;
  xcd_rows_sum<KP, ZR, ONES, 0>(en.value, size, vr, sum, sq);
  float part = (ONES ? wv : wv * en.value) - 0.5f * sq;
#pragma unroll
  for (int v = 0; v < VEC; v++) part = fmaf(0.5f * sum[v], sum[v], part);
  const float rest = wave_sum_dpp(part);
  if (lane == 0) rest_out[e] = rest;
  const float mult = multiplier(h, w0s + rest, r.y);
  if (cm != 0) {                                                  // some feature of this example is finished by its owner (xcd_segment)
    store_vec<VEC>(ep.S + (size_t)e * KP + lane * VEC, sum);
    if (lane == 0) ep.mult[e] = mult;
  }
  if (h.k1 && lane < size && !((cm >> lane) & 1ull)) {            // fm_sgd.h:38-43
    const float dw = -h.lr * (mult * en.value + h.regw * wv);
    tb.w[(size_t)en.id * tb.ws] = wv + dw;
  }
  float cbs[VEC]; const float ca = 1.0f + h.lr * (mult - h.regv);
#pragma unroll
  for (int v = 0; v < VEC; v++) cbs[v] = -h.lr * mult * sum[v];
  xcd_rows_store<KP, ZR, ONES, 0>(tb, h, opaque(en.id), __uint_as_float(opaque(__float_as_uint(en.value))), size, cm, vr, sum, cbs, ca, mult);   // fm_sgd.h:44-50 on the register-resident rows
}
template <int KP, int ZR, int T>
__device__ __forceinline__ void xcd_rows_touch(const Tab& tb, uint32_t ids, uint32_t size, uint64_t cm, float (&vt)[ZR][Map<KP>::VEC]) {
  constexpr int VEC = Map<KP>::VEC;
  if constexpr (T < ZR) {
#pragma unroll
    for (int v = 0; v < VEC; v++) vt[T][v] = 0.f;
    if ((uint32_t)T < size && !((cm >> (uint32_t)T) & 1ull)) {     // (a deferred row is in the L2 already: its owner has just written it)
      const uint32_t id = lane_val<T>(ids);
      xcd_row_ld<VEC>(tb, (size_t)id, vt[T]);
    }
    xcd_rows_touch<KP, ZR, T + 1>(tb, ids, size, cm, vt);
  }
}
// the rows (and weights) an example will gather, asked for a barrier ahead: the values may still change -- nobody uses them -- but the lines are in the die's L2 when the example asks again
template <int KP, int ZR>
__device__ __forceinline__ void xcd_touch(const Tab& tb, const Hyper& h, const XcdRow& r0, float (&vt)[ZR][Map<KP>::VEC], float& wt, bool all) {
  XcdRow r = r0; if (all) r.cm = 0;
  constexpr int VEC = Map<KP>::VEC;
  const uint32_t lane = threadIdx.x & 63u;
  wt = 0.f;
  const uint32_t ids = opaque(r.en.id);
  if (r.valid && h.k1 && lane < r.size && !((r.cm >> lane) & 1ull)) wt = ld_l2(tb.w + (size_t)ids * tb.ws);
  xcd_rows_touch<KP, ZR, 0>(tb, ids, r.valid ? r.size : 0u, r.cm, vt);
}
// (an empty statement that needs the registers: the loads are kept and waited for, no instruction is spent on them)
template <int ZR, int VEC>
__device__ __forceinline__ void xcd_touch_done(const float (&vt)[ZR][VEC], float wt) {
  asm volatile("" :: "v"(wt));
#pragma unroll
  for (int t = 0; t < ZR; t++) {
#pragma unroll
    for (int v = 0; v < VEC; v++) asm volatile("" :: "v"(vt[t][v]));
  }
}

// one deferred feature of the batch on one wavefront: apply_seg_block's arithmetic (all occurrences summed in example order, one owner per
// row) with L2-served loads.  rows in flight per round: 16
template <int KP>
__device__ __forceinline__ void xcd_segment(const CDesc* __restrict__ dp, const TEntry* __restrict__ t_ent, const XcdEpoch& ep, const Tab& tb, const Hyper& h) {
  constexpr int VEC = Map<KP>::VEC, TL = 8;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t j = dp->feat, a = dp->a, b = dp->b;             // (wave-uniform: scalar loads of a read-only record)
  float v0[VEC];
  xcd_row_ld<VEC>(tb, (size_t)j, v0);
  const float wv0 = h.k1 ? ld_l2(tb.w + (size_t)j * tb.ws) : 0.f;
  float G[VEC]; float A = 0.f, Gw = 0.f;
#pragma unroll
  for (int v = 0; v < VEC; v++) G[v] = 0.f;
  for (uint32_t base = a; base < b; base += 64u) {
    const uint32_t cc = min(64u, b - base);
    TEntry te; te.e = 0; te.x = 0.f; float tm = 0.f;
    if (lane < cc) { te = t_ent[base + lane]; tm = ld_l2(ep.mult + te.e); }
    for (uint32_t q0 = 0; q0 < cc; q0 += TL) {
      float s2[TL][VEC];
#pragma unroll
      for (int q = 0; q < TL; q++) {
        const uint32_t e2 = bcast_u32<1>(te.e, (q0 + q) & 63u);
        if (q0 + q < cc) ld_l2_vec<VEC>(ep.S + (size_t)e2 * KP + lane * VEC, s2[q]);
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
  const float nocc = (float)(b - a);
  float nv[VEC];
#pragma unroll
  for (int v = 0; v < VEC; v++) {
    const float vv = v0[v];
    nv[v] = vv - h.lr * (G[v] - vv * A + nocc * h.regv * vv);
  }
  xcd_row_st<VEC>(tb, (size_t)j, nv);
  if (h.k1 && lane == 0) tb.w[(size_t)j * tb.ws] = wv0 - h.lr * (Gw + nocc * h.regw * wv0);
}

// up to XCD_ITEMS deferred features of one wavefront in LOCKSTEP: what does not depend on the running batch's examples (the descriptor, the
// occurrence list, the feature's row and weight -- the rule leaves a deferred row untouched while the examples run) is asked for in the shadow
// of the barrier that ends the examples (xcd_items_ask); after it the multipliers and the first 8 sums rows of ALL items are gathered
// together, one round trip for the lot (xcd_items_finish).  Same operations in the same order as xcd_segment.
constexpr int XCD_ITEMS = 4;
template <int VEC> struct XcdItems {
  uint32_t j[XCD_ITEMS], cnt[XCD_ITEMS];                         // cnt 0: no item in this slot
  TEntry te[XCD_ITEMS]; float v0[XCD_ITEMS][VEC]; float wv0[XCD_ITEMS];
};
template <int KP>
__device__ __forceinline__ void xcd_items_ask(XcdItems<Map<KP>::VEC>& it, const XcdEpoch& ep, const Tab& tb, const Hyper& h, const TEntry* __restrict__ t_ent,
                                              uint32_t s0, uint32_t stride, uint32_t c1) {
  constexpr int VEC = Map<KP>::VEC;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t a[XCD_ITEMS];
#pragma unroll
  for (int i = 0; i < XCD_ITEMS; i++) {
    const uint32_t s = s0 + (uint32_t)i * stride;
    it.cnt[i] = 0; it.j[i] = 0; a[i] = 0;
    if (s < c1) {
      const CDesc* dp = ep.cdesc + s;
      it.j[i] = dp->feat; a[i] = dp->a;
      const uint32_t n = dp->b - dp->a;
      it.cnt[i] = (n <= 64u) ? n : 0xFFFFFFFFu;                  // longer lists: xcd_segment
    }
  }
#pragma unroll
  for (int i = 0; i < XCD_ITEMS; i++) {
    it.te[i].e = 0; it.te[i].x = 0.f; it.wv0[i] = 0.f;
#pragma unroll
    for (int v = 0; v < VEC; v++) it.v0[i][v] = 0.f;
    if (it.cnt[i] != 0u && it.cnt[i] != 0xFFFFFFFFu) {
      if (lane < it.cnt[i]) it.te[i] = t_ent[a[i] + lane];
      xcd_row_ld<VEC>(tb, (size_t)it.j[i], it.v0[i]);
      if (h.k1) it.wv0[i] = ld_l2(tb.w + (size_t)it.j[i] * tb.ws);
    }
  }
}
template <int KP>
__device__ __forceinline__ void xcd_items_finish(const XcdItems<Map<KP>::VEC>& it, const XcdEpoch& ep, const Tab& tb, const Hyper& h, unsigned long long* tr = nullptr) {
  constexpr int VEC = Map<KP>::VEC, TL0 = 8, TL = 24;
  const uint32_t lane = threadIdx.x & 63u;
  float tm[XCD_ITEMS];
#pragma unroll
  for (int i = 0; i < XCD_ITEMS; i++) {
    tm[i] = 0.f;
    if (it.cnt[i] != 0u && it.cnt[i] != 0xFFFFFFFFu && lane < it.cnt[i]) tm[i] = ld_l2(ep.mult + it.te[i].e);
  }
  if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) tr[9] = wall_clock64(); }     // occurrence lists, rows (asked before the barrier) and multipliers are here
  float s1[XCD_ITEMS][TL0][VEC];
#pragma unroll
  for (int i = 0; i < XCD_ITEMS; i++) {
    const bool on = it.cnt[i] != 0u && it.cnt[i] != 0xFFFFFFFFu;
#pragma unroll
    for (int q = 0; q < TL0; q++) {
      const uint32_t e2 = bcast_u32<1>(it.te[i].e, (uint32_t)q);
      if (on && (uint32_t)q < it.cnt[i]) ld_l2_vec<VEC>(ep.S + (size_t)e2 * KP + lane * VEC, s1[i][q]);
      else {
#pragma unroll
        for (int v = 0; v < VEC; v++) s1[i][q][v] = 0.f;
      }
    }
  }
  if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) tr[10] = wall_clock64(); }    // the first 8 sums rows of every item are here
#pragma unroll
  for (int i = 0; i < XCD_ITEMS; i++) {
    if (it.cnt[i] == 0u || it.cnt[i] == 0xFFFFFFFFu) continue;
    const uint32_t cc = it.cnt[i];
    float G[VEC]; float A = 0.f, Gw = 0.f;
#pragma unroll
    for (int v = 0; v < VEC; v++) G[v] = 0.f;
#pragma unroll
    for (int q = 0; q < TL0; q++) {
      const float x2 = bcast_f32<1>(it.te[i].x, (uint32_t)q), m2 = bcast_f32<1>(tm[i], (uint32_t)q);
      if ((uint32_t)q < cc) {
        const float mx2 = m2 * x2;
#pragma unroll
        for (int v = 0; v < VEC; v++) G[v] = fmaf(mx2, s1[i][q][v], G[v]);
        A = fmaf(mx2, x2, A); Gw += mx2;
      }
    }
#pragma unroll 1
    for (uint32_t q0 = TL0; q0 < cc; q0 += TL) {
      float s2[TL][VEC];
#pragma unroll
      for (int q = 0; q < TL; q++) {
        const uint32_t e2 = bcast_u32<1>(it.te[i].e, (q0 + q) & 63u);
        if (q0 + q < cc) ld_l2_vec<VEC>(ep.S + (size_t)e2 * KP + lane * VEC, s2[q]);
        else {
#pragma unroll
          for (int v = 0; v < VEC; v++) s2[q][v] = 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < TL; q++) {
        const float x2 = bcast_f32<1>(it.te[i].x, (q0 + q) & 63u), m2 = bcast_f32<1>(tm[i], (q0 + q) & 63u);
        if (q0 + q < cc) {
          const float mx2 = m2 * x2;
#pragma unroll
          for (int v = 0; v < VEC; v++) G[v] = fmaf(mx2, s2[q][v], G[v]);
          A = fmaf(mx2, x2, A); Gw += mx2;
        }
      }
    }
    const float nocc = (float)cc;
    float nv[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) {
      const float vv = it.v0[i][v];
      nv[v] = vv - h.lr * (G[v] - vv * A + nocc * h.regv * vv);
    }
    xcd_row_st<VEC>(tb, (size_t)it.j[i], nv);
    if (h.k1 && lane == 0) tb.w[(size_t)it.j[i] * tb.ws] = it.wv0[i] - h.lr * (Gw + nocc * h.regw * it.wv0[i]);
  }
}

template <int KP, int ZR, bool ONES>
__global__ void __launch_bounds__(256, (Map<KP>::VEC * ZR <= 48) ? 4 : 2)
k_xcd_epoch(const XcdEpoch ep, const Tab tb, const Hyper h, const XcdSync sy) {
  constexpr int VEC = Map<KP>::VEC;
  __shared__ uint32_t s_idx, s_members, s_ok;
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  // ---- membership (device-scope: the workgroups of the launch sit on all dies) ----
  if (threadIdx.x == 0) {
    const uint32_t me = xcc_id() + 1u;
    if (blockIdx.x == 0) __hip_atomic_store(sy.ctl + 0, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t tgt = ld_dev(sy.ctl + 0);
    for (uint32_t t = 0; tgt == 0u && t < sy.spins; t++) { __builtin_amdgcn_s_sleep(8); tgt = ld_dev(sy.ctl + 0); }
    uint32_t idx = 0xFFFFFFFFu;
    if (tgt == me) idx = __hip_atomic_fetch_add(sy.ctl + 1, 1u, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t before = __hip_atomic_fetch_add(sy.ctl + 2, 1u + (idx & 0u), __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t ok = 0u, members = 0u;
    if (idx < XCD_MAX_MEMBERS) {
      uint32_t started = before + 1u;
      for (uint32_t t = 0; started < gridDim.x && t < sy.spins; t++) { __builtin_amdgcn_s_sleep(8); started = ld_dev(sy.ctl + 2); }
      // one word decides for everybody: 1 = every workgroup has started (the member count is final), 2 = do not start
      unsigned expected = 0u;
      const unsigned mine = (started >= gridDim.x) ? 1u : 2u;
      if (__hip_atomic_compare_exchange_strong(sy.ctl + 4, &expected, mine, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_AGENT)) expected = mine;
      ok = (expected == 1u) ? 1u : 0u;                           // (a failed exchange left the decision in `expected`)
      members = min(__hip_atomic_load(sy.ctl + 1, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_AGENT), XCD_MAX_MEMBERS);
    }
    s_idx = idx; s_members = members; s_ok = ok;
  }
  __syncthreads();
  if (s_ok == 0u || s_idx >= XCD_MAX_MEMBERS) return;
  const uint32_t idx = s_idx, P = s_members;
  __syncthreads();                                               // (s_ok is reused by the barriers)
  const uint32_t NW = P * 4u;                                    // wavefronts of the die; examples are dealt over all of them
  const uint32_t w = uni(idx * 4u + wv);
  const bool scan_wave = (w == NW - 1u) && h.k0;                  // the last wavefront runs the batch's bias recurrence instead of deferred features
  const uint32_t NWI = h.k0 ? NW - 1u : NW;                       // wavefronts the deferred features are dealt over (NW >= 4)
  const bool tracer = ep.trace != nullptr && idx == 0u && wv == 0u;
  uint32_t gen = 0;
  XcdRow nx = xcd_row(ep, 0, w, min(ep.B, ep.n_rows));           // this wavefront's first example of batch 0
  for (uint32_t b = 0; b < ep.n_batch; b++) {
    const uint64_t row0 = (uint64_t)b * ep.B;
    const uint32_t nb = (uint32_t)min((uint64_t)ep.B, (uint64_t)ep.n_rows - row0);
    float* __restrict__ rest = ep.rest + (size_t)(b % ep.d) * ep.Bc;
    unsigned long long* tr = (tracer && b < ep.trace_batches) ? ep.trace + (size_t)b * 16 : nullptr;
    if (tr && lane == 0) tr[0] = wall_clock64();
    // the read-only streams that are asked for in the next barrier's shadow -- this batch's deferred-feature records and occurrence lists, the
    // next batch's entries / masks / labels -- are contiguous: every wavefront brings ONE 4 KiB piece (64 lines) of them into the die's L2 now
    const uint32_t c0 = ep.cbatch[b], c1 = ep.cbatch[b + 1u];
    const uint64_t tb0 = ep.batch_base[b], tb1 = ep.batch_base[b + 1u];
    unsigned pre = 0;
    {
      const uint64_t r1 = row0 + ep.B;
      const uint32_t nb1 = (b + 1u < ep.n_batch) ? (uint32_t)min((uint64_t)ep.B, (uint64_t)ep.n_rows - r1) : 0u;
      const char* p0 = (const char*)(ep.cdesc + c0);            const uint64_t n0 = ((uint64_t)(c1 - c0) * sizeof(CDesc) + 63u) >> 6;
      const char* p1 = (const char*)(ep.t_ent + tb0);           const uint64_t n1 = ((tb1 - tb0) * sizeof(TEntry) + 63u) >> 6;
      const uint64_t ea = nb1 ? (ep.fixed_nnz ? r1 * ep.fixed_nnz : tb1) : 0;       // (entries of the next batch start where this batch's end: batch_base counts them)
      const char* p2 = (const char*)(ep.ent + ea);              const uint64_t n2 = nb1 ? (((ep.fixed_nnz ? (uint64_t)nb1 * ep.fixed_nnz : (ep.batch_base[min(b + 2u, ep.n_batch)] - tb1)) * sizeof(Entry) + 63u) >> 6) : 0;
      const char* p3 = (const char*)(ep.cmask + r1);            const uint64_t n3 = ((uint64_t)nb1 * 8u + 63u) >> 6;
      const char* p4 = (const char*)(ep.target + r1);           const uint64_t n4 = ((uint64_t)nb1 * 4u + 63u) >> 6;
      uint64_t l = (uint64_t)w * 64u + lane;
      const char* q = nullptr;
      if (l < n0) q = p0 + (l << 6); else { l -= n0;
      if (l < n1) q = p1 + (l << 6); else { l -= n1;
      if (l < n2) q = p2 + (l << 6); else { l -= n2;
      if (l < n3) q = p3 + (l << 6); else { l -= n3;
      if (l < n4) q = p4 + (l << 6); } } } }
      if (q && !(ep.flags & 2u)) pre = *(const unsigned*)((uintptr_t)q & ~(uintptr_t)3);
    }
    // ---- examples: the multipliers use the bias after the recurrence of batch b - d (slot (b + 1) % d of the ring) ----
    {
      const float w0s = h.k0 ? (float)ld_l2(ep.w0_ring + ((b + 1u) % ep.d)) : 0.f;
      XcdRow r = nx;                                              // (asked for a barrier ago; further examples of a long batch: here)
#pragma unroll 1
      for (uint32_t e = w; e < nb; e += NW) {
        if (e != w) r = xcd_row(ep, row0, e, nb);
        xcd_example<KP, ZR, ONES>(ep, tb, h, r, w0s, rest, e == w ? tr : nullptr);
      }
    }
    asm volatile("" :: "v"(pre));                                // (the piece is in the L2: loads return in order)
    if (tr && lane == 0) tr[1] = wall_clock64();
    xcd_arrive(sy, idx, ++gen);
    if (tr && lane == 0) tr[2] = wall_clock64();
    // in the barrier's shadow: this wavefront's deferred features of the batch (everything the examples do not write), and its first
    // example of the NEXT batch (read-only)
    const TEntry* __restrict__ te = ep.t_ent + tb0;
    XcdItems<VEC> it;
    const uint32_t s0 = scan_wave ? c1 : c0 + w;
    xcd_items_ask<KP>(it, ep, tb, h, te, s0, NWI, c1);
    {
      const uint64_t row1 = row0 + ep.B;
      const uint32_t nb1 = (b + 1u < ep.n_batch) ? (uint32_t)min((uint64_t)ep.B, (uint64_t)ep.n_rows - row1) : 0u;
      nx = xcd_row(ep, row1, w, nb1);
    }
    if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) tr[12] = wall_clock64(); }   // what was asked for in the shadow is here
    if (!xcd_wait(sy, P, gen, &s_ok)) return;
    if (tr && lane == 0) tr[3] = wall_clock64();
    // ---- the features that occur more than once in the batch, one owner each; the bias recurrence of the batch on its own wavefront ----
    if (scan_wave) {
      const ScanSmall sc{rest, ep.target + row0, ep.w0_ring + (b % ep.d), ep.w0_ring + ((b + 1u) % ep.d), nb, ep.chunk};
      scan_small<true>(sc, h);
    } else {
      xcd_items_finish<KP>(it, ep, tb, h, tr);
      if (tr && lane == 0) tr[11] = wall_clock64();
      uint32_t k = 0;                                            // what the lockstep slots did not take: lists beyond 64 occurrences, a fifth item
#pragma unroll 1
      for (uint32_t s = s0; s < c1; s += NWI, k++) {
        if (k < (uint32_t)XCD_ITEMS && ep.cdesc[s].b - ep.cdesc[s].a <= 64u) continue;
        xcd_segment<KP>(ep.cdesc + s, te, ep, tb, h);
      }
    }
    if (tr && lane == 0) tr[4] = wall_clock64();
    xcd_arrive(sy, idx, ++gen);
    if (tr && lane == 0) tr[5] = wall_clock64();
    // in the barrier's shadow: the rows the next example gathers are brought into the die's L2
    float vt[ZR][VEC]; float wt;
    xcd_touch<KP, ZR>(tb, h, nx, vt, wt, (ep.flags & 1u) != 0u);
    if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) tr[13] = wall_clock64(); }   // the next example's lines are in the L2
    if (!xcd_wait(sy, P, gen, &s_ok)) return;
    xcd_touch_done<ZR, VEC>(vt, wt);
    if (tr && lane == 0) tr[6] = wall_clock64();
  }
}

}  // namespace fmx

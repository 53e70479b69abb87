// k_sequential_rows: the reference's own trajectory (fm_learn_sgd_element.h:56-67: one example at a time, in file order; fm_sgd.h:33-51) on ONE
// wavefront, a row at a time instead of an entry at a time.
//
// k_sequential (fmx_kernels.h) walks a row entry by entry and drains the store queue after each: 32 dependent round trips per example,
// 46 us per example at the bench shape -- slower than the CPU it is the parity instrument for (21.8 k vs 26.8 k examples/s).  The loop of
// the reference does not need that: its update of entry i reads the sums of the WHOLE row (fm.m_sum, taken before the first update) and the
// parameter v_{j_i, f} itself, which only an EARLIER entry of the same row with the same id can have changed.  So for a row without a
// repeated id (every one-hot row; checked per row) the example is: gather all its rows at once (one round trip), sums in fp64 in entry
// order, multiplier, bias step, every row updated from its registers, stored.  What the NEXT example must see is only what it shares with
// this one -- so its rows are asked for BEFORE this example's stores are issued (their latency runs under this example's arithmetic) and
// the few it shares with this example (found by comparing the ids) are read again after the stores have drained.
// Rows with a repeated id, and rows beyond the register path, take the entry-by-entry loop (seq_row_entries: k_sequential's body).
// The update arithmetic is fp32 with its per-example constants formed in fp64 (the parameters are fp32; the sums stay fp64).
//
// Second half of the file: CONFLICT-FREE RUNS -- where consecutive rows rarely share a feature the same trajectory runs at batch speed, one
// launch per run of a few hundred rows (k_run_fused, k_run_apply; ~70 x the rate of the kernels above at BASELINE's headline shape).
#pragma once

namespace fmx {

// one row, entry by entry, every store drained before the next entry (k_sequential's loop body: repeated ids see their own earlier update)
template <int KP>
__device__ __forceinline__ void seq_row_entries(const Entry* __restrict__ ent, uint64_t a, uint32_t size, float yf, const Tab& tb, const Hyper& h, double& w0) {
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR;
  const uint32_t lane = threadIdx.x & 63u;
  const bool act = lane < LPR && lane * VEC < tb.rs;
  double sum[VEC]; double sq = 0.0, lin = 0.0;
#pragma unroll
  for (int v = 0; v < VEC; v++) sum[v] = 0.0;
  for (uint32_t i = 0; i < size; i++) {
    const Entry e = ent[a + i];
    if (h.k1 && lane == 0) lin += (double)ld_l2(tb.w + (size_t)e.id * tb.ws) * (double)e.value;
    if (act) {
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        const double d = (double)ld_l2(tb.V + (size_t)e.id * tb.rs + lane * VEC + v) * (double)e.value;
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
  const double y = (double)yf;
  double mult;
  if (h.task == 0) { p = fmin(h.max_d, p); p = fmax(h.min_d, p); mult = -(y - p); }
  else mult = -y * (1.0 - 1.0 / (1.0 + exp(-y * p)));
  if (h.k0) w0 -= h.lr_d * (mult + h.reg0_d * w0);
  for (uint32_t i = 0; i < size; i++) {
    const Entry e = ent[a + i];
    const double x = (double)e.value;
    if (h.k1 && lane == 0) {
      const double wv = (double)ld_l2(tb.w + (size_t)e.id * tb.ws);
      tb.w[(size_t)e.id * tb.ws] = (float)(wv - h.lr_d * (mult * x + h.regw_d * wv));
    }
    if (act) {
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        float* pv = tb.V + (size_t)e.id * tb.rs + lane * VEC + v;
        const double vv = (double)ld_l2(pv);
        const double grad = sum[v] * x - vv * x * x;
        *pv = (float)(vv - h.lr_d * (mult * grad + h.regv_d * vv));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // a later entry of this row (repeated id) and the next row must observe these stores
  }
}

struct SeqRow { uint64_t a; uint32_t size; float y; Entry en; };
__device__ __forceinline__ SeqRow seq_meta(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, const float* __restrict__ target, uint32_t r, uint32_t n_rows) {
  SeqRow m; m.a = 0; m.size = 0; m.y = 0.f; m.en.id = 0; m.en.value = 0.f;
  if (r < n_rows) {
    const uint32_t lane = threadIdx.x & 63u;
    m.a = row_ptr[r];
    m.size = uni((uint32_t)(row_ptr[r + 1] - m.a));
    m.y = __uint_as_float(uni(__float_as_uint(target[r])));
    if (lane < m.size && m.size <= 64u) m.en = ent[m.a + lane];
  }
  return m;
}
// rows (and weights) of an example into registers, L2-served
template <int KP, int ZR, int T>
__device__ __forceinline__ void seq_rows_load(const Tab& tb, uint32_t ids, uint32_t size, float (&vr)[ZR][Map<KP>::VEC]) {
  constexpr int VEC = Map<KP>::VEC;
  if constexpr (T < ZR) {
    if ((uint32_t)T < size) xcd_row_ld<VEC>(tb, (size_t)lane_val<T>(ids), vr[T]);
    else {
#pragma unroll
      for (int v = 0; v < VEC; v++) vr[T][v] = 0.f;
    }
    seq_rows_load<KP, ZR, T + 1>(tb, ids, size, vr);
  }
}
// the rows of the NEXT example that this example has just written (bit t of `again`): read once more
template <int KP, int ZR, int T>
__device__ __forceinline__ void seq_rows_again(const Tab& tb, uint32_t ids, uint64_t again, float (&vr)[ZR][Map<KP>::VEC]) {
  constexpr int VEC = Map<KP>::VEC;
  if constexpr (T < ZR) {
    if ((again >> T) & 1ull) xcd_row_ld<VEC>(tb, (size_t)lane_val<T>(ids), vr[T]);
    seq_rows_again<KP, ZR, T + 1>(tb, ids, again, vr);
  }
}
template <int KP, int ZR, int T>
__device__ __forceinline__ void seq_rows_sum(uint32_t xs, uint32_t size, const float (&vr)[ZR][Map<KP>::VEC], double (&sum)[Map<KP>::VEC], double& sq) {
  constexpr int VEC = Map<KP>::VEC;
  if constexpr (T < ZR) {
    if ((uint32_t)T < size) {                                      // (wave-uniform; in entry order, like the reference's loop)
      const double x = (double)__uint_as_float(lane_val<T>(xs));
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        const double d = (double)vr[T][v] * x;
        sum[v] += d;
        sq += d * d;
      }
    }
    seq_rows_sum<KP, ZR, T + 1>(xs, size, vr, sum, sq);
  }
}
template <int KP, int ZR, int T>
__device__ __forceinline__ void seq_rows_store(const Tab& tb, uint32_t ids, uint32_t xs, uint32_t size, const float (&vr)[ZR][Map<KP>::VEC], const float (&sumf)[Map<KP>::VEC],
                                               float lm, float lrv) {
  constexpr int VEC = Map<KP>::VEC;
  if constexpr (T < ZR) {
    if ((uint32_t)T < size) {
      const float x = __uint_as_float(lane_val<T>(xs));
      float nv[VEC];
#pragma unroll
      for (int v = 0; v < VEC; v++) {                              // fm_sgd.h:47-49: v -= lr (mult (sum x - v x x) + regv v)
        const float vv = vr[T][v];
        const float grad = sumf[v] * x - vv * x * x;
        nv[v] = vv - (lm * grad + lrv * vv);
      }
      xcd_row_st<VEC>(tb, (size_t)lane_val<T>(ids), nv);
    }
    seq_rows_store<KP, ZR, T + 1>(tb, ids, xs, size, vr, sumf, lm, lrv);
  }
}

// one example: `cur` with its rows in R / weights in wv (asked for while the previous example ran); asks for `nxt`'s rows into Rn / wn
template <int KP, int ZR>
__device__ __forceinline__ void seq_step(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, const float* __restrict__ target, uint32_t r, uint32_t n_rows,
                                         const Tab& tb, const Hyper& h, double& w0, SeqRow& cur, float (&R)[ZR][Map<KP>::VEC], float& wv, bool& have,
                                         SeqRow& nxt, float (&Rn)[ZR][Map<KP>::VEC], float& wn, bool& have_n) {
  constexpr int VEC = Map<KP>::VEC;
  const uint32_t lane = threadIdx.x & 63u;
  nxt = seq_meta(ent, row_ptr, target, r + 1, n_rows);            // read-only: asked for now, needed when this example is done
  have_n = false;
  if (r >= n_rows) return;
  // the entry registers pass an empty statement: as "values of a load" the compiler would drain every outstanding load before each broadcast
  const uint32_t ids = opaque(cur.en.id), xs = opaque(__float_as_uint(cur.en.value));
  bool fast = have && cur.size <= (uint32_t)ZR;
  if (fast) {                                                     // a repeated id inside the row: the entry-by-entry loop (fm_sgd.h:44-50 semantics)
    bool dup = false;
    for (uint32_t t = 1; t < cur.size; t++) {
      const uint32_t idt = bcast_u32<1>(ids, t);
      dup |= (lane < t) && (ids == idt);
    }
    if (__ballot(dup) != 0ull) fast = false;
  }
  if (!fast) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    seq_row_entries<KP>(ent, cur.a, cur.size, cur.y, tb, h, w0);  // (drains its own stores)
    if (nxt.size && nxt.size <= (uint32_t)ZR) {                  // nothing is in flight: the next example's rows are read after this one's stores
      const uint32_t idn = opaque(nxt.en.id);
      wn = 0.f;
      if (h.k1 && lane < nxt.size) wn = ld_l2(tb.w + (size_t)idn * tb.ws);
      seq_rows_load<KP, ZR, 0>(tb, idn, nxt.size, Rn);
      have_n = true;
    }
    return;
  }
  // ---- sums (fm_model.h:116-125) in fp64, in entry order ----
  double sum[VEC]; double sq = 0.0;
#pragma unroll
  for (int v = 0; v < VEC; v++) sum[v] = 0.0;
  seq_rows_sum<KP, ZR, 0>(xs, cur.size, R, sum, sq);
  const float xl = __uint_as_float(xs);
  double part = ((h.k1 && lane < cur.size) ? (double)wv * (double)xl : 0.0) - 0.5 * sq;
  if (lane * VEC < tb.rs) {
#pragma unroll
    for (int v = 0; v < VEC; v++) part += 0.5 * sum[v] * sum[v];
  }
  double p = (h.k0 ? w0 : 0.0) + wave_sum_d(part);
  const double y = (double)cur.y;
  double mult;
  if (h.task == 0) { p = fmin(h.max_d, p); p = fmax(h.min_d, p); mult = -(y - p); }
  else mult = -y * (1.0 - 1.0 / (1.0 + exp(-y * p)));
  if (h.k0) w0 -= h.lr_d * (mult + h.reg0_d * w0);                // fm_sgd.h:34-37
  // ---- the next example's rows: asked for BEFORE this example's stores (their latency runs under the update below) ----
  uint32_t idn = 0;
  const bool pre = nxt.size != 0u && nxt.size <= (uint32_t)ZR;
  // (what the PREVIOUS examples stored must be in memory before these loads are issued -- their stores were issued an example ago, the wait
  //  is over before it starts; what THIS example is about to store is handled below)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (pre) {
    idn = opaque(nxt.en.id);
    wn = 0.f;
    if (h.k1 && lane < nxt.size) wn = ld_l2(tb.w + (size_t)idn * tb.ws);
    seq_rows_load<KP, ZR, 0>(tb, idn, nxt.size, Rn);
    have_n = true;
  }
  // ---- update from the registers (fm_sgd.h:38-50), fp32 with the example's constants formed in fp64 ----
  const float lm = (float)(h.lr_d * mult), lrv = (float)(h.lr_d * h.regv_d), lrw = (float)(h.lr_d * h.regw_d);
  if (h.k1 && lane < cur.size) tb.w[(size_t)ids * tb.ws] = wv - (lm * xl + lrw * wv);
  float sumf[VEC];
#pragma unroll
  for (int v = 0; v < VEC; v++) sumf[v] = (float)sum[v];
  seq_rows_store<KP, ZR, 0>(tb, ids, xs, cur.size, R, sumf, lm, lrv);
  // ---- what the next example shares with this one was read too early: once more, after the stores have drained ----
  if (pre) {
    bool hit = false;
    for (uint32_t t = 0; t < cur.size; t++) {
      const uint32_t idt = bcast_u32<1>(ids, t);
      hit |= (idn == idt);
    }
    const uint64_t again = __ballot(hit && lane < nxt.size);
    if (again != 0ull) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (h.k1 && lane < nxt.size && ((again >> lane) & 1ull)) wn = ld_l2(tb.w + (size_t)idn * tb.ws);
      seq_rows_again<KP, ZR, 0>(tb, idn, again, Rn);
    }
  }
}

template <int KP, int ZR>
__global__ void __launch_bounds__(64)
k_sequential_rows(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, const float* __restrict__ target,
                  uint32_t n_rows, const Tab tb, Hyper h, double* w0_ptr) {
  static_assert(Map<KP>::EPI == 1, "one row per wave-wide load");
  constexpr int VEC = Map<KP>::VEC;
  const uint32_t lane = threadIdx.x & 63u;
  double w0 = *w0_ptr;
  float A[ZR][VEC], B[ZR][VEC];
  float wA = 0.f, wB = 0.f;
  bool haveA = false, haveB = false;
  SeqRow ra = seq_meta(ent, row_ptr, target, 0, n_rows), rb;
  if (ra.size && ra.size <= (uint32_t)ZR) {
    const uint32_t id0 = opaque(ra.en.id);
    if (h.k1 && lane < ra.size) wA = ld_l2(tb.w + (size_t)id0 * tb.ws);
    seq_rows_load<KP, ZR, 0>(tb, id0, ra.size, A);
    haveA = true;
  }
#pragma unroll 1
  for (uint32_t r = 0; r < n_rows; r += 2) {
    seq_step<KP, ZR>(ent, row_ptr, target, r, n_rows, tb, h, w0, ra, A, wA, haveA, rb, B, wB, haveB);
    seq_step<KP, ZR>(ent, row_ptr, target, r + 1, n_rows, tb, h, w0, rb, B, wB, haveB, ra, A, wA, haveA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) *w0_ptr = w0;
}

}  // namespace fmx

namespace fmx {

// ----------------------------------------------------------------------------------------------
// k_sequential_wg: the same trajectory with EIGHT wavefronts on each example.  One wavefront issues ~3 000 instructions per 32-entry example
// and nothing hides their latency (k_sequential_rows: 7.6 us per example); here wavefront w takes the entries t = w, w + 8, ... : its part
// of the gathers, of the fp64 sums (combined through the LDS, one workgroup barrier per example), of the update.  Everything scalar -- the
// prediction, the multiplier, the bias step -- is computed by every wavefront from the same numbers in the same order, so all of them hold
// the same w0.  The next example's rows are asked for before this example's stores, as in k_sequential_rows; what the two share is found
// through a table of stamps in the LDS (bucket = hash of the id; a false hit only costs a second read) and read again behind a second
// barrier.  A repeated id inside a row (found the same way, confirmed exactly) or a row beyond 64 entries: the first wavefront runs the
// entry-by-entry loop and hands the bias on.
// ----------------------------------------------------------------------------------------------
constexpr int SEQ_W = 8, SEQ_SLOTS = 8;                          // wavefronts per example, row slots per wavefront (64 entries)
constexpr uint32_t SEQ_BUCKETS = 8192;
template <int KP> struct SeqLds {
  double sum[2][SEQ_W][KP];
  double sq[2][SEQ_W][64];
  uint32_t stamp[2][SEQ_BUCKETS];
  uint32_t dup[SEQ_BUCKETS];
  double w0;
};
__device__ __forceinline__ uint32_t seq_bucket(uint32_t id) { return (id * 0x9E3779B1u) >> 19; }   // 13 bits

template <int KP>
__device__ __forceinline__ void seq_wg_ask(const Tab& tb, const Hyper& h, const SeqRow& m, uint32_t ids, uint32_t wv, float (&R)[SEQ_SLOTS][Map<KP>::VEC], float& wl) {
  constexpr int VEC = Map<KP>::VEC;
  const uint32_t lane = threadIdx.x & 63u;
  wl = 0.f;
  if (h.k1 && lane < m.size) wl = ld_l2(tb.w + (size_t)ids * tb.ws);
#pragma unroll
  for (int i = 0; i < SEQ_SLOTS; i++) {
    const uint32_t t = wv + (uint32_t)SEQ_W * i;
#pragma unroll
    for (int v = 0; v < VEC; v++) R[i][v] = 0.f;
    if (t < m.size) xcd_row_ld<VEC>(tb, (size_t)bcast_u32<1>(ids, t), R[i]);
  }
}

// the read-only part of an example, asked for TWO examples ahead: its entries start where the previous row's end (no dependent load: the
// address is known), its end offset and label are single loads; nothing here is waited for before the example after next begins
struct SeqAhead { uint64_t a, end; float y; Entry en; bool in; };
__device__ __forceinline__ SeqAhead seq_ahead(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, const float* __restrict__ target,
                                              uint32_t r, uint32_t n_rows, uint64_t a, uint64_t nnz) {
  SeqAhead m; m.a = a; m.end = a; m.y = 0.f; m.en.id = 0; m.en.value = 0.f; m.in = r < n_rows;
  if (m.in) {
    const uint32_t lane = threadIdx.x & 63u;
    m.end = row_ptr[r + 1];
    m.y = target[r];
    if (nnz) m.en = ent[min(a + lane, nnz - 1)];                 // (the lanes beyond the row are dropped when its length is known)
  }
  return m;
}
__device__ __forceinline__ SeqRow seq_finish(const SeqAhead& m) {
  SeqRow o; o.a = m.a; o.size = 0; o.y = 0.f; o.en.id = 0; o.en.value = 0.f;
  if (m.in) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t len = m.end - m.a;
    o.size = uni((uint32_t)(len > 0xFFFFFFFFull ? 0xFFFFFFFFull : len));
    o.y = __uint_as_float(uni(__float_as_uint(m.y)));
    if (lane < o.size && o.size <= 64u) o.en = m.en;
  }
  return o;
}

template <int KP>
__device__ __forceinline__ void seq_wg_step(SeqLds<KP>& L, const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, const float* __restrict__ target,
                                            uint32_t r, uint32_t n_rows, uint64_t nnz, const Tab& tb, const Hyper& h, double& w0,
                                            const SeqRow& cur, float (&R)[SEQ_SLOTS][Map<KP>::VEC], float& wl, bool& have,
                                            SeqRow& nxt, const SeqAhead& nxt_raw, float (&Rn)[SEQ_SLOTS][Map<KP>::VEC], float& wn, bool& have_n,
                                            SeqAhead& after) {
  constexpr int VEC = Map<KP>::VEC;
  const uint32_t lane = threadIdx.x & 63u, wv = uni(threadIdx.x >> 6);
  const uint32_t par = r & 1u;
  nxt = seq_finish(nxt_raw);                                      // (asked for a whole example ago)
  have_n = false;
  after.in = false; after.a = nxt.a + nxt.size; after.end = after.a; after.y = 0.f; after.en.id = 0; after.en.value = 0.f;
  if (r >= n_rows) return;
  const uint32_t ids = opaque(cur.en.id), xs = opaque(__float_as_uint(cur.en.value));
  bool fast = have && cur.size <= 64u;
  if (fast) {                                                     // a repeated id?  every wavefront asks the same table the same question
    const uint32_t b = seq_bucket(ids);
    volatile uint32_t* dupt = L.dup;                              // (volatile: the read must come from the LDS -- another LANE may have written the bucket)
    if (lane < cur.size) dupt[b] = lane;
    __builtin_amdgcn_s_waitcnt(0xc07f);                           // lgkmcnt(0)
    bool maybe = lane < cur.size && dupt[b] != lane;
    if (__ballot(maybe) != 0ull) {
      bool dup = false;
      for (uint32_t t = 1; t < cur.size; t++) dup |= (lane < t) && (ids == bcast_u32<1>(ids, t));
      if (__ballot(dup) != 0ull) fast = false;
    }
  }
  if (!fast) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wv == 0) {
      seq_row_entries<KP>(ent, cur.a, cur.size, cur.y, tb, h, w0);
      if (lane == 0) L.w0 = w0;
    }
    __syncthreads();
    w0 = L.w0;
    if (nxt.size && nxt.size <= 64u) { seq_wg_ask<KP>(tb, h, nxt, opaque(nxt.en.id), wv, Rn, wn); have_n = true; }
    after = seq_ahead(ent, row_ptr, target, r + 2, n_rows, nxt.a + nxt.size, nnz);
    return;
  }
  // ---- this wavefront's part of the sums, fp64, in entry order ----
  double sum[VEC]; double sq = 0.0;
#pragma unroll
  for (int v = 0; v < VEC; v++) sum[v] = 0.0;
#pragma unroll
  for (int i = 0; i < SEQ_SLOTS; i++) {
    const uint32_t t = wv + (uint32_t)SEQ_W * i;
    if (t < cur.size) {
      const double x = (double)__uint_as_float(bcast_u32<1>(xs, t));
#pragma unroll
      for (int v = 0; v < VEC; v++) { const double d = (double)R[i][v] * x; sum[v] += d; sq += d * d; }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; v++) L.sum[par][wv][lane * VEC + v] = sum[v];
  L.sq[par][wv][lane] = sq;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wavefront's stores of the previous example are in memory ...
  __syncthreads();                                               // ... and so are everybody's; the partial sums are in the LDS
  // ---- asked for NOW: the next example's rows (before this example's stores: their latency runs under everything below), and the
  //      read-only part of the example after it (nothing waits for that before the next barrier) ----
  const bool pre = nxt.size != 0u && nxt.size <= 64u;
  uint32_t idn = 0;
  if (pre) { idn = opaque(nxt.en.id); seq_wg_ask<KP>(tb, h, nxt, idn, wv, Rn, wn); have_n = true; }
  after = seq_ahead(ent, row_ptr, target, r + 2, n_rows, nxt.a + nxt.size, nnz);
  double tot[VEC]; double tsq = 0.0;
#pragma unroll
  for (int v = 0; v < VEC; v++) tot[v] = 0.0;
#pragma unroll
  for (int q = 0; q < SEQ_W; q++) {                              // (the same order in every wavefront: they all hold the same numbers)
#pragma unroll
    for (int v = 0; v < VEC; v++) tot[v] += L.sum[par][q][lane * VEC + v];
    tsq += L.sq[par][q][lane];
  }
  const float xl = __uint_as_float(xs);
  double part = ((h.k1 && lane < cur.size) ? (double)wl * (double)xl : 0.0) - 0.5 * tsq;
#pragma unroll
  for (int v = 0; v < VEC; v++) part += 0.5 * tot[v] * tot[v];
  double p = (h.k0 ? w0 : 0.0) + wave_sum_d(part);
  const double y = (double)cur.y;
  double mult;
  if (h.task == 0) { p = fmin(h.max_d, p); p = fmax(h.min_d, p); mult = -(y - p); }
  else mult = -y * (1.0 - 1.0 / (1.0 + exp(-y * p)));
  if (h.k0) w0 -= h.lr_d * (mult + h.reg0_d * w0);                // fm_sgd.h:34-37
  // ---- this wavefront's part of the update (fm_sgd.h:38-50) ----
  const float lm = (float)(h.lr_d * mult), lrv = (float)(h.lr_d * h.regv_d), lrw = (float)(h.lr_d * h.regw_d);
  if (wv == 0 && h.k1 && lane < cur.size) tb.w[(size_t)ids * tb.ws] = wl - (lm * xl + lrw * wl);
  float totf[VEC];
#pragma unroll
  for (int v = 0; v < VEC; v++) totf[v] = (float)tot[v];
#pragma unroll
  for (int i = 0; i < SEQ_SLOTS; i++) {
    const uint32_t t = wv + (uint32_t)SEQ_W * i;
    if (t < cur.size) {
      const float x = __uint_as_float(bcast_u32<1>(xs, t));
      float nv[VEC];
#pragma unroll
      for (int v = 0; v < VEC; v++) { const float vv = R[i][v]; nv[v] = vv - (lm * (totf[v] * x - vv * x * x) + lrv * vv); }
      xcd_row_st<VEC>(tb, (size_t)bcast_u32<1>(ids, t), nv);
    }
  }
  // ---- what the next example shares with this one was read too early (every wavefront finds the same answer) ----
  if (pre) {
    volatile uint32_t* stt = L.stamp[par];
    if (lane < cur.size) stt[seq_bucket(ids)] = r + 1u;
    __builtin_amdgcn_s_waitcnt(0xc07f);                           // lgkmcnt(0)
    const bool hit = lane < nxt.size && stt[seq_bucket(idn)] == r + 1u;
    const uint64_t again = __ballot(hit);
    if (again != 0ull) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                           // everybody's stores of this example are in memory
      if (h.k1 && ((again >> lane) & 1ull)) wn = ld_l2(tb.w + (size_t)idn * tb.ws);
#pragma unroll
      for (int i = 0; i < SEQ_SLOTS; i++) {
        const uint32_t t = wv + (uint32_t)SEQ_W * i;
        if (t < nxt.size && ((again >> t) & 1ull)) xcd_row_ld<VEC>(tb, (size_t)bcast_u32<1>(idn, t), Rn[i]);
      }
    }
  }
}

template <int KP>
__global__ void __launch_bounds__(64 * SEQ_W)
k_sequential_wg(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, const float* __restrict__ target,
                uint32_t n_rows, uint64_t nnz, const Tab tb, Hyper h, double* w0_ptr) {
  static_assert(Map<KP>::EPI == 1, "one row per wave-wide load");
  constexpr int VEC = Map<KP>::VEC;
  extern __shared__ __attribute__((aligned(16))) unsigned char seq_lds_raw[];
  SeqLds<KP>& L = *reinterpret_cast<SeqLds<KP>*>(seq_lds_raw);
  const uint32_t wv = uni(threadIdx.x >> 6);
  for (uint32_t i = threadIdx.x; i < SEQ_BUCKETS; i += blockDim.x) { L.stamp[0][i] = 0u; L.stamp[1][i] = 0u; L.dup[i] = 0u; }
  __syncthreads();
  double w0 = *w0_ptr;
  float A[SEQ_SLOTS][VEC], B[SEQ_SLOTS][VEC];
  float wA = 0.f, wB = 0.f;
  bool haveA = false, haveB = false;
  // three examples' read-only parts rotate (this one, the next: finished at its step's start, the one after: being read), two row buffers
  SeqRow c0 = seq_meta(ent, row_ptr, target, 0, n_rows), c1, c2;
  c1.a = 0; c1.size = 0; c1.y = 0.f; c1.en.id = 0; c1.en.value = 0.f; c2 = c1;
  SeqAhead m0, m1 = seq_ahead(ent, row_ptr, target, 1, n_rows, c0.a + c0.size, nnz), m2;
  m0 = m1; m0.in = false; m2 = m0;
  if (c0.size && c0.size <= 64u) { seq_wg_ask<KP>(tb, h, c0, opaque(c0.en.id), wv, A, wA); haveA = true; }
#pragma unroll 1
  for (uint32_t r = 0; r < n_rows; r += 6) {
    seq_wg_step<KP>(L, ent, row_ptr, target, r + 0, n_rows, nnz, tb, h, w0, c0, A, wA, haveA, c1, m1, B, wB, haveB, m2);
    seq_wg_step<KP>(L, ent, row_ptr, target, r + 1, n_rows, nnz, tb, h, w0, c1, B, wB, haveB, c2, m2, A, wA, haveA, m0);
    seq_wg_step<KP>(L, ent, row_ptr, target, r + 2, n_rows, nnz, tb, h, w0, c2, A, wA, haveA, c0, m0, B, wB, haveB, m1);
    seq_wg_step<KP>(L, ent, row_ptr, target, r + 3, n_rows, nnz, tb, h, w0, c0, B, wB, haveB, c1, m1, A, wA, haveA, m2);
    seq_wg_step<KP>(L, ent, row_ptr, target, r + 4, n_rows, nnz, tb, h, w0, c1, A, wA, haveA, c2, m2, B, wB, haveB, m0);
    seq_wg_step<KP>(L, ent, row_ptr, target, r + 5, n_rows, nnz, tb, h, w0, c2, B, wB, haveB, c0, m0, A, wA, haveA, m1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) *w0_ptr = w0;
}

}  // namespace fmx

namespace fmx {

// ----------------------------------------------------------------------------------------------
// Conflict-free runs: the reference's trajectory at batch speed where the data allows it.
// Inside a run of consecutive rows that share NO feature with each other, the online loop (fm_learn_sgd_element.h:56-67) and "one batch
// with the bias recurrence coupled example by example" are the same computation: the sums of a row only read parameters no other row of the
// run writes, the bias is the one sequential thread (fm_sgd.h:34-37) -- and that is exactly what k_rowsums -> k_scan (micro-chunk 1, the
// multipliers come out of the recurrence) -> k_apply compute for a batch.  So the slot is cut, once, into maximal such runs (greedy, in file
// order) and an epoch is ONE launch per run where the rows fit the registers (k_run_fused, below), two or three otherwise (fmx_sgd.hip
// seq_runs_epoch); a row that repeats an id (fm_sgd.h:44-50: the second occurrence sees the first's update)
// is a run of its own and goes through the entry-by-entry kernel.
//   k_run_keys / k_run_prev: prev[r] = 1 + the latest earlier row that shares a feature with row r (0: none), bit 31: the row repeats an id --
//   through one radix sort of (feature << 32 | row) keys; the greedy cut itself is a loop over prev[] on the host (fmx_sgd.hip ensure_runs).
// ----------------------------------------------------------------------------------------------
// ---- a run's bias recurrence + update in ONE launch (k_run_apply) ----
// The recurrence of a run (micro-chunk 1: the multiplier of example e is taken at the bias after example e - 1, fm_sgd.h:34-37) is a chain of
// n dependent steps -- ~0.2 us each on k_scan's wavefront, 80 us for a run of 400 rows whose sums and update take 7 + 12 us.  It reads 8 bytes
// per example, so EVERY workgroup of the update launch solves it for itself, parallel in time (Newton on the whole path with affine prefix
// scans: k_scan_pit's arithmetic on one workgroup, fmx_kernels.h) -- the same instructions on the same numbers in every workgroup, so all
// of them hold the same multipliers -- and then updates its rows from them.  While the recurrence is solved the rows the wavefront is going to
// update are on their way into the L2 (run_touch_row).  Workgroup 0 stores the bias behind the run; it goes to ANOTHER slot than the one the
// run started from (a workgroup that starts late still reads the start value).  Runs of up to RUN_FUSED_MAX rows; longer ones take
// k_scan_pit on one workgroup between the two launches.
constexpr uint32_t RUN_FUSED_MAX = 2048;

template <int TASK>
__device__ __forceinline__ void run_eval(const Hyper& h, float w0s, float d, float r, float y, float& m, float& dm) {
  constexpr float LOG2E = 1.4426950408889634f;
  const float p = (w0s + d) + r;
  if constexpr (TASK == 1) {
    const float inv = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(LOG2E * y * p));   // 1 / (1 + e^{y p})
    m = -y * inv;                                                   // fm_learn_sgd_element.h:64
    dm = (y * y) * inv * (1.0f - inv);
  } else {
    const float gs = h.sgda ? 2.0f : 1.0f;
    const float pc = fmaxf(h.min_target, fminf(h.max_target, p));
    m = gs * (pc - y);                                              // fm_learn_sgd_element.h:60-62
    dm = (p > h.min_target && p < h.max_target) ? gs : 0.f;
  }
}

// every thread of the workgroup (256) calls this; on return s_d[i] holds the multiplier of example i of the run and the return value is the bias
// behind the run.  s_r, s_y, s_d, s_a, s_b: n floats each.
template <int TASK, bool PRELOADED = false>
__device__ __forceinline__ double run_scan_wg(const float* __restrict__ rest, const float* __restrict__ target, uint32_t n, const Hyper& h, double w0,
                                              float* s_r, float* s_y, float* s_d, float* s_a, float* s_b, float (*s_map)[2], float* s_chg, double* s_end) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float w0s = (float)w0;
  for (uint32_t i = tid; i < n; i += 256u) {
    if (!PRELOADED) s_r[i] = rest[i];                                // (PRELOADED: the caller filled s_r -- k_run_fused, out of the run's tagged slots)
    s_y[i] = target[i]; s_d[i] = 0.f;
  }
  __syncthreads();
  const uint32_t wseg = ((n + 255u) / 256u) * 64u;                  // examples per wavefront, whole vectors of 64
  const uint32_t q0 = wv * wseg;
  bool converged = false;
  float x_end = 0.f;
  for (uint32_t it = 0; it < PIT_MAX_IT; it++) {
    // every example's step linearised at the current path (an affine map), the maps composed by a prefix scan: each example gets the
    // composite of the wavefront's examples BEFORE it, `run` ends as the composite of all of them
    AMap run = AMap{1.f, 0.f};
    for (uint32_t cb = 0; cb < wseg; cb += 64u) {
      const uint32_t i = q0 + cb + lane;
      if (q0 + cb >= n) break;                                       // (wave-uniform)
      float a = 1.f, b = 0.f;
      if (i < n) {
        const float d = s_d[i];
        float m, dm;
        run_eval<TASK>(h, w0s, d, s_r[i], s_y[i], m, dm);
        const float F = fmaf(h.reg0, w0s + d, m), D = dm + h.reg0;
        a = 1.0f - h.lr * D; b = -h.lr * (F - D * d);
      }
      amap_scan64(a, b);
      const float ea = __shfl_up(a, 1u), eb = __shfl_up(b, 1u);
      AMap ex = (lane == 0u) ? AMap{1.f, 0.f} : AMap{ea, eb};
      ex = amap_after(run, ex);
      if (i < n) { s_a[i] = ex.a; s_b[i] = ex.b; }
      run = amap_after(run, AMap{bcast_f32<1>(a, 63), bcast_f32<1>(b, 63)});
    }
    if (lane == 0) { s_map[wv][0] = run.a; s_map[wv][1] = run.b; }
    __syncthreads();
    float x = 0.f;                                                   // path offset at the wavefront's first example
    for (uint32_t w = 0; w < wv; w++) x = fmaf(s_map[w][0], x, s_map[w][1]);
    float xe = x;
    for (uint32_t w = wv; w < 4u; w++) xe = fmaf(s_map[w][0], xe, s_map[w][1]);   // ... behind the run's last one (the same chain of fmas in every wavefront)
    float chg = 0.f;
    for (uint32_t cb = lane; cb < wseg; cb += 64u) {
      const uint32_t i = q0 + cb;
      if (i < n) {
        const float dn = fmaf(s_a[i], x, s_b[i]);
        chg = nanmax(fabsf(dn - s_d[i]), chg);
        s_d[i] = dn;
      }
    }
    for (int o = 32; o > 0; o >>= 1) chg = nanmax(__shfl_xor(chg, o), chg);
    if (lane == 0) s_chg[wv] = chg;
    __syncthreads();
    const float tot = nanmax(nanmax(s_chg[0], s_chg[1]), nanmax(s_chg[2], s_chg[3]));
    x_end = xe;
    __syncthreads();                                                 // (s_map / s_chg are written again by the next iteration)
    if (tot < PIT_TOL && xe - xe == 0.f) { converged = true; break; }   // the path moved by less than tol: what it moved TO is exact to ~tol^2
  }
  if (converged) {
    for (uint32_t i = tid; i < n; i += 256u) {
      float m, dm;
      run_eval<TASK>(h, w0s, s_d[i], s_r[i], s_y[i], m, dm);
      s_d[i] = m;
    }
    __syncthreads();
    return w0 + (double)x_end;
  }
  // Newton did not settle (the chain itself oscillates): one thread walks it, k_scan's arithmetic
  if (tid == 0) {
    double w = w0;
    for (uint32_t i = 0; i < n; i++) {
      const float ws = (float)w;
      const float m = multiplier_task<TASK>(h, ws + s_r[i], s_y[i]);
      s_d[i] = m;
      w -= (double)h.lr * ((double)m + (double)h.reg0 * (double)ws);
    }
    *s_end = w;
  }
  __syncthreads();
  return *s_end;
}

// the rows an example is going to update, asked for (one dword per 128 bytes) so that they are in the L2 when row_apply reads them
template <int KP>
__device__ __forceinline__ float run_touch_row(const Entry* __restrict__ ent, uint32_t size, const Tab& tb, const Hyper& h) {
  const uint32_t lane = threadIdx.x & 63u;
  float acc = 0.f;
  for (uint32_t base = 0; base < size; base += 64u) {
    if (base + lane < size) {
      const uint32_t id = ent[base + lane].id;
      if (h.k1) acc += tb.w[(size_t)id * tb.ws];
      const float* row = tb.V + (size_t)id * tb.rs;
      for (uint32_t o = 0; o < tb.rs; o += 32u) acc += row[o];
    }
  }
  return acc;
}

template <int KP, int TASK>
__global__ void __launch_bounds__(256)
k_run_apply(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, uint64_t row0, uint32_t n_rows, const Tab tb, Hyper h,
            const float* __restrict__ S, const float* __restrict__ rest, const float* __restrict__ target,
            const double* __restrict__ w0_in, double* __restrict__ w0_out) {
  extern __shared__ float run_lds[];                                 // 5 x n_rows floats
  __shared__ float s_map[4][2];
  __shared__ float s_chg[4];
  __shared__ double s_end;
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR;
  const uint32_t lane = threadIdx.x & 63u, f = lane % LPR;
  const uint32_t wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
  const uint32_t nwaves = gridDim.x * 4u;
  float touched = 0.f;
  if (wave0 < n_rows) {
    const uint64_t a = row_ptr[row0 + wave0];
    touched = run_touch_row<KP>(ent + a, (uint32_t)(row_ptr[row0 + wave0 + 1] - a), tb, h);
  }
  float* s_d = run_lds + 2 * (size_t)n_rows;
  if (h.k0) {
    const double w_end = run_scan_wg<TASK>(rest, target + row0, n_rows, h, *w0_in, run_lds, run_lds + n_rows, s_d, run_lds + 3 * (size_t)n_rows,
                                           run_lds + 4 * (size_t)n_rows, s_map, s_chg, &s_end);
    if (blockIdx.x == 0 && threadIdx.x == 0) *w0_out = w_end;
  } else {                                                           // no bias: the multipliers do not depend on each other
    for (uint32_t i = threadIdx.x; i < n_rows; i += 256u) s_d[i] = multiplier_task<TASK>(h, rest[i], target[row0 + i]);
    __syncthreads();
  }
  if (touched == 1.2345e-38f) s_chg[0] = touched;                    // (keeps the touching loads)
  for (uint32_t e = wave0; e < n_rows; e += nwaves) {
    const uint64_t a = row_ptr[row0 + e];
    const uint32_t size = (uint32_t)(row_ptr[row0 + e + 1] - a);
    float sum[VEC];
    load_vec<VEC>(S + (size_t)e * KP + f * VEC, sum);
    row_apply<KP, 8, false>(ent + a, size, tb, h, sum, s_d[e]);
  }
}

// ---- a run in ONE launch (k_run_fused) ----
// One wavefront per example, the example's rows stay in its registers (k_fused's layout): gather, sums, the example's rest_e published as ONE
// 8-byte agent-scope store {tag of the run, rest_e} into the run's slot array -- no arrival counter: 400 read-modify-writes of one word from
// eight dies serialise (measured: 12 .. 55 us per run with one) -- and every thread polls the slots it is going to need until they carry the
// run's tag: the poll that succeeds IS the load of rest_e.  Then every workgroup solves the bias recurrence for itself (run_scan_wg) and
// updates its rows from the registers.  V is read once and written once, and a run costs one launch boundary instead of two.  Every
// workgroup of the launch must be resident at the same time (the host gates the launch on the occupancy query; runs of up to RUN_ONE_MAX
// rows); a poll that runs into its bound all the same takes NO step for the workgroup's rows, raises RUN_ERR_EXCHANGE in the handle's error
// word, and the epoch returns FMX_E_HIP (the handle takes two launches per run from then on).  Tags: 1 + the run's index in the epoch; the
// slot array is zeroed at the start of every epoch.
constexpr uint32_t RUN_ONE_MAX = 1024;
constexpr uint32_t RUN_ERR_EXCHANGE = 32u;
struct RunSync { unsigned long long* slots; uint32_t tag; uint32_t* err; uint32_t spins; };

template <int KP, int ZR, int TASK>
__global__ void __launch_bounds__(256)
k_run_fused(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, const float* __restrict__ target, uint64_t row0, uint32_t n_rows,
            const Tab tb, Hyper h, const double* __restrict__ w0_in, double* __restrict__ w0_out, const RunSync rs, uint32_t fixed_nnz) {
  // KP < 64 (k <= 32): EPI = 64 / KP rows per wave-wide load -- lane group g = lane / KP holds entry t * EPI + g of row slot t (k_fused's layout)
  extern __shared__ float run_lds[];                                 // 5 x n_rows floats
  __shared__ float s_map[4][2];
  __shared__ float s_chg[4];
  __shared__ double s_end;
  constexpr int VEC = Map<KP>::VEC, LPR = Map<KP>::LPR, EPI = Map<KP>::EPI;
  const uint32_t lane = threadIdx.x & 63u, g = lane / LPR, f = lane % LPR;
  const uint32_t e = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
  const bool have = e < n_rows;
  Entry en; en.id = 0; en.value = 0.f;
  float wv = 0.f;
  uint32_t size = 0;
  float vr[ZR][VEC];
  float sum[VEC];
#pragma unroll
  for (int v = 0; v < VEC; v++) sum[v] = 0.f;
  if (have) {
    // fixed_nnz != 0: every row of the slot holds that many entries -- the row_ptr round trip drops out of the chain row_ptr -> entries -> rows
    const uint64_t a = fixed_nnz ? (row0 + e) * (uint64_t)fixed_nnz : row_ptr[row0 + e];
    size = fixed_nnz ? fixed_nnz : (uint32_t)(row_ptr[row0 + e + 1] - a);   // (<= min(64, ZR * EPI): the host checked the slot's longest row)
    if (lane < size) {
      en = load_stream8(ent + a + lane);
      if (h.k1) wv = load_w(tb.w + (size_t)en.id * tb.ws);
    }
#pragma unroll
    for (int t = 0; t < ZR; t++) {                                   // (cross-lane reads with every lane enabled: `have` is wave-uniform)
      const uint32_t idx = (uint32_t)t * EPI + g;
      const uint32_t id = bcast_u32<EPI>(en.id, idx & 63u);
      if (idx < size) {
        row_ld<VEC, 1>(tb, (size_t)id, f * VEC, vr[t]);
      } else {
#pragma unroll
        for (int v = 0; v < VEC; v++) vr[t][v] = 0.f;
      }
    }
    float sq = 0.f;                                                  // fm_model.h:116-125
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
    float part = wv * en.value - 0.5f * sq;
    if (lane < LPR) {
#pragma unroll
      for (int v = 0; v < VEC; v++) part = fmaf(0.5f * sum[v], sum[v], part);
    }
    const float rest = wave_sum_dpp(part);
    if (lane == 0)
      __hip_atomic_store(rs.slots + e, ((unsigned long long)rs.tag << 32) | (unsigned long long)__float_as_uint(rest), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  bool late = false;
  for (uint32_t i = threadIdx.x; i < n_rows; i += 256u) {
    unsigned long long u = __hip_atomic_load(rs.slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t t = 0; (uint32_t)(u >> 32) != rs.tag && t < rs.spins; t++) {
      __builtin_amdgcn_s_sleep(4);
      u = __hip_atomic_load(rs.slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    late |= (uint32_t)(u >> 32) != rs.tag;
    run_lds[i] = __uint_as_float((uint32_t)u);
  }
  if (late) atomicOr(rs.err, RUN_ERR_EXCHANGE);
  if (__syncthreads_or(late ? 1 : 0)) return;
  float* s_d = run_lds + 2 * (size_t)n_rows;
  if (h.k0) {
    const double w_end = run_scan_wg<TASK, true>(nullptr, target + row0, n_rows, h, *w0_in, run_lds, run_lds + n_rows, s_d, run_lds + 3 * (size_t)n_rows,
                                                 run_lds + 4 * (size_t)n_rows, s_map, s_chg, &s_end);
    if (blockIdx.x == 0 && threadIdx.x == 0) *w0_out = w_end;
  } else {
    for (uint32_t i = threadIdx.x; i < n_rows; i += 256u) s_d[i] = multiplier_task<TASK>(h, run_lds[i], target[row0 + i]);
    __syncthreads();
  }
  if (!have) return;
  const float mult = s_d[e];
  if (h.k1 && lane < size) {                                         // fm_sgd.h:38-43
    tb.w[(size_t)en.id * tb.ws] = wv - h.lr * (mult * en.value + h.regw * wv);
  }
#pragma unroll
  for (int t = 0; t < ZR; t++) {                                     // fm_sgd.h:44-50 on the register-resident rows
    const uint32_t idx = (uint32_t)t * EPI + g;
    const uint32_t id = bcast_u32<EPI>(en.id, idx & 63u);
    const float x = bcast_f32<EPI>(en.value, idx & 63u);
    if (idx < size && f * VEC < tb.rs) {
      float* pv = tb.V + (size_t)id * tb.rs + f * VEC;
      float nv[VEC];
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        const float vv = vr[t][v];
        const float grad = sum[v] * x - vv * x * x;
        nv[v] = vv - h.lr * (mult * grad + h.regv * vv);
      }
      store_row<VEC, 2>(pv, nv);
    }
  }
}

static __global__ void __launch_bounds__(256)
k_run_keys(const Entry* __restrict__ ent, const uint64_t* __restrict__ row_ptr, uint32_t n_rows, uint64_t* __restrict__ keys) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t r = wave0; r < n_rows; r += nwaves) {
    const uint64_t a = row_ptr[r], b = row_ptr[r + 1];
    for (uint64_t i = a + lane; i < b; i += 64) keys[i] = ((uint64_t)ent[i].id << 32) | r;
  }
}
static __global__ void __launch_bounds__(256)
k_run_prev(const uint64_t* __restrict__ keys, uint64_t nnz, uint32_t* __restrict__ prev) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < nnz; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t k1 = keys[i], k0 = keys[i - 1];
    if ((k1 >> 32) != (k0 >> 32)) continue;
    const uint32_t r = (uint32_t)k1, p = (uint32_t)k0;
    if (p == r) atomicOr(prev + r, 0x80000000u);
    else atomicMax(prev + r, (p + 1u) & 0x7FFFFFFFu);             // (rows below 2^31 - 1: checked by the caller; the flag bit is ORed in separately)
  }
}

}  // namespace fmx

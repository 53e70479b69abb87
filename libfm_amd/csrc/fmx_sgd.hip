// fmx_sgd.hip -- C-ABI (include/fmx.h): the SGD family -- sequential / minibatch / hogwild epochs, the split step of
// the multi-GPU driver, the bias-lag bookkeeping, the (batch, feature) segment build, and SGDA.
#include "fmx_internal.h"
#include "fmx_xcd_kernels.h"
#include "fmx_seq_kernels.h"
#include "fmx_small_kernels.h"

namespace {
// default micro-chunk of the bias recurrence.  The reference moves w0 after EVERY example (fm_sgd.h:34-37); summing the
// multipliers of a chunk and applying them at once is a batch step of size lr * chunk on a coordinate whose curvature is
// up to 1 (regression) or 1/4 (logistic) PER EXAMPLE, so lr * chunk * curvature must stay below 2 or the bias
// oscillates with growing amplitude (observed: classification, lr = 0.02, chunk = 1024 -> w0 = +-3..6, accuracy 0.50).
// Default: the largest power of two with lr * chunk * curvature <= 1, capped at FMX_W0_CHUNK_CAP.  The cap is what sets how
// closely the batch rule follows the reference's bias PATH (and through it the reference's predictions): at the bench shape the
// rule's bias ends 0.025 from the online loop's at chunk 256 and 0.0014 at chunk 32, so the cap went from
// 256 to 32 in round 5 -- the recurrence is solved parallel in time (k_scan_pit) at 40-80 us per 262 144 examples whatever the chunk; as a
// one-wavefront chain (k_scan1's sub-piece form, FMX_SCAN=serial) such chunks cost ~60 ns each.  An explicit w0_chunk is honoured.
uint32_t default_w0_chunk(const fmx_config& c) { return fmx_default_w0_chunk(c.learn_rate, c.task); }

// row slots (ZR) of the k_fused instance used for a slot: the smallest instance whose registers hold the longest row
template <int KP> constexpr bool fused_zr_ok(int zr) { return Map<KP>::VEC * zr <= 128; }
template <int KP> int fused_zr_select(uint32_t max_row) {
  constexpr int EPI = Map<KP>::EPI;
  const uint32_t need = (max_row + EPI - 1) / EPI;         // row slots per lane to keep a whole row in registers
  for (int zr : {8, 16, 32, 40, 64})                       // 40: Criteo-shaped rows (39 fields)
    if (fused_zr_ok<KP>(zr) && need <= (uint32_t)zr) return zr;
  return 8;                                                // rows too long for the register file: the kernel's two-pass branch
}
// longest row the selected instance keeps in registers (longer rows take its two-pass branch; FUSED_EXACT defers them)
template <int KP> uint32_t fused_row_cap_kp(uint32_t max_row) {
  return std::min<uint32_t>(64u, (uint32_t)fused_zr_select<KP>(max_row) * (uint32_t)Map<KP>::EPI);
}

template <int KP, int VAR>
int launch_fused_zr(fmx_handle h, const Slot& s, const Hyper& hy, uint64_t row0, uint32_t n_rows, hipStream_t st,
                    const double* w0_in, float* rest_out, const uint64_t* cmask = nullptr, float* S_out = nullptr,
                    float* mult_out = nullptr, uint32_t* handoff_err = nullptr, bool keep_wside = false) {
#define FMX_LAUNCH_ZR(ZRV)                                                                                  \
  if constexpr (fused_zr_ok<KP>(ZRV)) {                                                                     \
    FMX_LAUNCH_WAVES((k_fused<KP, ZRV, VAR>), n_rows, st, s.ent, s.row_ptr, s.target, row0,                 \
                     n_rows, h->tb, hy, w0_in, rest_out, cmask, S_out, mult_out, s.fixed_nnz, handoff_err,                  \
                     (const uint64_t*)(keep_wside ? s.lmask : nullptr), keep_wside ? s.wside : (float*)nullptr); }
  switch (fused_zr_select<KP>(s.max_row)) {
    case 8:  FMX_LAUNCH_ZR(8);  break;
    case 16: FMX_LAUNCH_ZR(16); break;
    case 32: FMX_LAUNCH_ZR(32); break;
    case 40: FMX_LAUNCH_ZR(40); break;
    case 64: FMX_LAUNCH_ZR(64); break;
  }
#undef FMX_LAUNCH_ZR
  return FMX_OK;
}

}  // namespace

// SHORT rows (a feature shard of P GPUs sees nnz / P entries per example): examples per wavefront of k_rowsums_multi / k_apply_multi --
// about 24 entries per 32-slot round (mean + 1.7 sigma of a shard's binomial row lengths stays inside one round); 0 = rows are long enough
// for the one-example-per-wavefront kernels (or KP < 64: several rows per wave-wide load, not built)
extern "C++" uint32_t multi_group_size(const Slot& s, int KP) {
  if (KP < 64 || KP > 256 || s.n_rows == 0) return 0;          // (beyond 256 factors the sums of a group of examples do not fit the LDS)
  const double avg = (double)s.nnz / (double)s.n_rows;
  if (avg > 12.0) return 0;
  static const double fill = []() { const char* e = getenv("FMX_MULTI_FILL"); const double f = e ? atof(e) : 0.0; return (f >= 8.0 && f <= 32.0) ? f : 24.0; }();
  const double g = avg > 0.0 ? fill / avg : (double)MULTI_GMAX;   // (FMX_MULTI_FILL: A/B knob for the probe, 8 .. 32)
  return (uint32_t)std::max(2.0, std::min((double)MULTI_GMAX, std::floor(g)));
}

extern "C" {

// ---------------------------------------------------------------------------------------------
// SGD
// ---------------------------------------------------------------------------------------------
uint32_t fmx_default_w0_chunk(double learn_rate, int task) {
  const double curv = (task == FMX_TASK_REGRESSION) ? 1.0 : 0.25;
  const double lim = (learn_rate > 0) ? 1.0 / (learn_rate * curv) : (double)FMX_W0_CHUNK_CAP;
  uint32_t chunk = 1;
  while (chunk < (uint32_t)FMX_W0_CHUNK_CAP && (double)(chunk * 2) <= lim) chunk *= 2;
  return chunk;
}
int fmx_partial_floats(fmx_handle h, uint32_t batch, uint64_t* n_floats) {
  if (!h || !n_floats) return FMX_E_ARG;
  *n_floats = (uint64_t)batch * (uint64_t)(h->KP + 1);
  return FMX_OK;
}

// partial buffer layout: [n_rows][KP] factor sums, then [n_rows] scalars
int fmx_sgd_partial(fmx_handle h, int slot, uint64_t row0, uint32_t n_rows, float* d_partial, void* stream) {
  int rc = check_slot(h, slot, false);
  if (rc) return rc;
  const Slot& s = h->slots[slot];
  if (row0 + n_rows > s.n_rows) return fail(h, FMX_E_ARG, "fmx_sgd_partial: rows [%llu,+%u) outside slot (%u rows)",
                                            (unsigned long long)row0, n_rows, s.n_rows);
  if (!s.blocks.empty()) return fail(h, FMX_E_UNSUPPORTED, "relations are not supported with SGD");   // fm_learn_sgd.h:61-63
  if (!d_partial) return fail(h, FMX_E_ARG, "fmx_sgd_partial: d_partial is NULL");
  if (n_rows == 0) return FMX_OK;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  float* S = d_partial;
  float* c = d_partial + (size_t)n_rows * h->KP;
  return sgd_partial_rows(h, s, row0, n_rows, S, c, st);
}

// a run of rows of a batch into the batch's partial buffer (S: [batch rows][KP], c: [batch rows]) -- the chunked exchange of
// fmx_group_sgd_epoch enqueues the all-reduce of one run while the next one is being summed
extern "C++" int sgd_partial_rows(fmx_handle h, const Slot& s, uint64_t row0, uint32_t n_rows, float* S, float* c, hipStream_t st) {
  if (n_rows == 0) return FMX_OK;
  if (const uint32_t G = multi_group_size(s, h->KP)) {          // short rows: several examples per wavefront
    KP_SWITCH(h->KP, { if constexpr (KP >= 64 && KP <= 256) { FMX_LAUNCH_WAVES((k_rowsums_multi<KP, true, false>), ((uint64_t)n_rows + G - 1) / G, st,
                                                                  s.ent, s.row_ptr, row0, n_rows, h->tb, h->cfg.k1, S, c, G); } });
    HIPCHK(h, hipGetLastError());
    return FMX_OK;
  }
  KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_rowsums<KP, true, false>), n_rows, st, s.ent, s.row_ptr, row0, n_rows, h->tb, h->cfg.k1, S, c, (const float*)nullptr));
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

// the batch an epoch of the MINIBATCH rule runs with on this slot (fmx_sgd_opts::batch): the rows' collision mass -- summed over
// the feature shards when the handle is one -- and the stability cut of the default
extern "C++" int sgd_resolve_batch(fmx_handle h, Slot& s, const fmx_sgd_opts* opts, fmx_batch_info* bi) {
  const int slot = (int)(&s - h->slots);
  double C = 0.0;
  if (h->group && !h->owns_group && h->group->hs.size() > 1) {          // one process, several shards: their shares add up
    for (fmx_handle m : h->group->hs) {
      if (!m) return fail(h, FMX_E_STATE, "a member of the group was destroyed");
      int rc = ensure_coll_mass(m, m->slots[slot]);
      if (rc) { h->err = m->err; return rc; }
      C += m->slots[slot].coll_mass;
    }
  } else {
    int rc = ensure_coll_mass(h, s);
    if (rc) return rc;
    C = s.coll_mass;
    if (h->comm && h->cfg.shard_world > 1) {                 // one process per GPU: a collective on EVERY call (include/fmx.h) -- a per-rank
      rc = comm_sum_double(h, &C);                            // cache made the decision to enter it per-rank too: a rank that re-uploaded its
      if (rc) return rc;                                      // slot entered alone and the job hung (round-4 advisor).  8 bytes per epoch.
      s.coll_mass_world = C;
    }
  }
  resolve_batch(h->cfg, C, opts ? opts->batch : 0u, FMX_DEFAULT_BATCH, 1.0, bi);
  return FMX_OK;
}

int fmx_sgd_batch_info(fmx_handle h, int slot, const fmx_sgd_opts* opts, fmx_batch_info* out) {
  int rc = check_slot(h, slot, false);
  if (rc) return rc;
  if (!out) return fail(h, FMX_E_ARG, "fmx_sgd_batch_info: out is NULL");
  HIPCHK(h, hipSetDevice(h->device));
  return sgd_resolve_batch(h, h->slots[slot], opts, out);
}

// builds the (batch, feature) segments of a slot for batch size B (device radix sort; once per data set).
// Round 6: ONE scratch allocation (fmx_dev_alloc: built from <= 1 GiB chunks, see fmx_internal.h -- ten plain allocations of > 1 GiB were 2.4 s
// of a 2.46 s preparation), the temporary-storage sizes of the three hipCUB calls asked for once, the counts read back ONCE, every per-batch
// table (first segment / first deferred segment / first entry of a batch) computed on the device and read back together: two host
// synchronisations instead of seven, no copy of row_ptr to the host.  FMX_TRACE_SETUP=1 prints where the time goes.
extern "C++" int ensure_segments(fmx_handle h, Slot& s, uint32_t B) {
  if (s.seg_B == B && s.t_ent) return FMX_OK;
  if (&s >= h->slots && &s < h->slots + FMX_MAX_SLOTS) {          // (the rows of a kept `-relation` block are bucketed by their own session)
    int _rc = slot_in_session(h, (int)(&s - h->slots), "re-bucketing the rows for another batch size"); if (_rc) return _rc;
  }
  free_segments(s);
  const auto t_setup0 = std::chrono::steady_clock::now();
  static const bool trace = getenv("FMX_TRACE_SETUP") != nullptr;
  auto tp = [&](const char* what) { if (trace) { (void)hipStreamSynchronize(h->stream); fprintf(stderr, "[fmx setup] segments %-22s %8.3f ms\n", what,
                                     1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_setup0).count()); } };
  struct Acc { fmx_handle h; std::chrono::steady_clock::time_point t0;
               ~Acc() { h->setup_acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } acc_{h, t_setup0};
  const uint64_t nnz = s.nnz;
  const uint32_t n_batches = (s.n_rows + B - 1) / B;
  if (nnz >= (1ull << 31)) return fail(h, FMX_E_UNSUPPORTED, "segmented apply: nnz >= 2^31 in one slot (split the data set)");
  hipStream_t st = h->stream;
  uint32_t cap = 64;
  KP_SWITCH(h->KP, cap = fused_row_cap_kp<KP>(s.max_row));
  uint32_t fbits = 1; while (fbits < 32 && (1ull << fbits) < std::max<uint64_t>(h->n_local, 2)) fbits++;   // bits of a local feature id
  int bits_batch = 1; while ((1ull << bits_batch) < n_batches) bits_batch++;
  char* scratch = nullptr;
  int rc = FMX_OK;
  const size_t cnt = (size_t)std::max<uint64_t>(nnz, 1);
  const size_t nb1 = (size_t)n_batches + 1;
#define SEG_CHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    rc = fail(h, FMX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); goto done; } } while (0)
  {
    // temporary storage of the three hipCUB calls (a size query touches no memory)
    size_t tmp_sort = 0, tmp_scan = 0, tmp_xscan = 0;
    SEG_CHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t*)nullptr,
                                               (int)cnt, 0, (int)fbits + bits_batch, st));
    SEG_CHK(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_scan, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)cnt, st));
    SEG_CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_xscan, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)cnt, st));
    const size_t tmp_bytes = std::max<size_t>(std::max(tmp_sort, std::max(tmp_scan, tmp_xscan)), 256);
    // scratch layout (256-byte aligned pieces): keys_a | keys_b | vals_a | flags | pos | cflag | cpos | counts | hipCUB temp
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t off = 0;
    const size_t o_ka = off; off += al(cnt * 8);
    const size_t o_kb = off; off += al(cnt * 8);
    const size_t o_va = off; off += al(cnt * 8);
    const size_t o_fl = off; off += al(cnt * 4);
    const size_t o_po = off; off += al(cnt * 4);
    const size_t o_cf = off; off += al(cnt * 4);
    const size_t o_cp = off; off += al(cnt * 4);
    const size_t o_ct = off; off += 256;
    const size_t o_tmp = off; off += al(tmp_bytes);
    SEG_CHK(fmx_dev_alloc(&scratch, off));
    uint64_t* keys_a = (uint64_t*)(scratch + o_ka); uint64_t* keys_b = (uint64_t*)(scratch + o_kb); uint64_t* vals_a = (uint64_t*)(scratch + o_va);
    uint32_t* flags = (uint32_t*)(scratch + o_fl); uint32_t* pos = (uint32_t*)(scratch + o_po);
    uint32_t* cflag = (uint32_t*)(scratch + o_cf); uint32_t* cpos = (uint32_t*)(scratch + o_cp);
    uint32_t* d_counts = (uint32_t*)(scratch + o_ct);            // {segments, deferred segments, longest segment, -}
    void* tmp = scratch + o_tmp;
    uint64_t* vals_b = nullptr;                                    // the sort's output payload IS t_ent (payload layout == TEntry): allocated on its own, kept
    SEG_CHK(fmx_dev_alloc(&vals_b, cnt * 8));
    s.t_ent = reinterpret_cast<TEntry*>(vals_b);
    SEG_CHK(fmx_dev_alloc(&s.cmask, (size_t)std::max<uint32_t>(s.n_rows, 1) * 8));
    SEG_CHK(fmx_dev_alloc(&s.d_batch_seg, nb1 * 4));
    SEG_CHK(fmx_dev_alloc(&s.d_cbatch, nb1 * 4));
    SEG_CHK(fmx_dev_alloc(&s.d_batch_base, nb1 * 8));
    tp("allocations");
    hipLaunchKernelGGL(k_seg_slow_rows, dim3(std::min<uint32_t>((s.n_rows + 255) / 256, 2048)), dim3(256), 0, st, s.row_ptr, s.n_rows, cap, s.cmask);
    s.fused_cap = cap;
    SEG_CHK(hipMemsetAsync(d_counts, 0, 16, st));
    if (nnz) {
      hipLaunchKernelGGL(k_seg_keys, dim3(wave_grid(s.n_rows)), dim3(256), 0, st, s.ent, s.row_ptr, s.n_rows, B, fbits, keys_a, vals_a, d_counts + 3);
      size_t tb_ = tmp_bytes;
      SEG_CHK(hipcub::DeviceRadixSort::SortPairs(tmp, tb_, keys_a, keys_b, vals_a, vals_b, (int)nnz, 0, (int)fbits + bits_batch, st));
      hipLaunchKernelGGL(k_seg_heads, dim3(2048), dim3(256), 0, st, keys_b, nnz, flags);
      tp("keys + sort + heads");
      tb_ = tmp_bytes;
      SEG_CHK(hipcub::DeviceScan::InclusiveSum(tmp, tb_, flags, pos, (int)nnz, st));
      // per-batch tables on the device (kept: the persistent small-batch kernel and the group driver's captured graphs read them there)
      hipLaunchKernelGGL(k_seg_batches, dim3((n_batches + 256) / 256), dim3(256), 0, st, pos, nnz, s.row_ptr, s.n_rows, B, n_batches, s.d_batch_seg, s.d_batch_base);
      // longest segment (fmx_epoch_stats::max_feature_count); keys_a is free after the sort: head[nseg + 1] <= 8 bytes per entry
      uint32_t* head = reinterpret_cast<uint32_t*>(keys_a);
      hipLaunchKernelGGL(k_seg_head_pos, dim3(2048), dim3(256), 0, st, flags, pos, nnz, head);
      hipLaunchKernelGGL(k_seg_max_count, dim3(2048), dim3(256), 0, st, head, pos, nnz, d_counts + 2);
      // what the one-pass kernel (FMX_APPLY_FUSED) must leave to k_apply_seg: features occurring more than once in their batch + rows too
      // long for its registers
      hipLaunchKernelGGL(k_seg_mark, dim3(2048), dim3(256), 0, st, keys_b, reinterpret_cast<const TEntry*>(vals_b), head, pos, nnz,
                         s.ent, s.row_ptr, B, fbits, cap, (uint32_t)(s.max_row > cap ? 1u : 0u), s.cmask, cflag);
      SEG_CHK(hipGetLastError());
      tb_ = tmp_bytes;
      SEG_CHK(hipcub::DeviceScan::ExclusiveSum(tmp, tb_, cflag, cpos, (int)nnz, st));   // (over nnz >= nseg elements: the tail is never read)
      hipLaunchKernelGGL(k_seg_counts, dim3(1), dim3(64), 0, st, pos, nnz, cpos, cflag, d_counts + 2, d_counts);
      uint32_t counts[4] = {0, 0, 0, 0};
      SEG_CHK(hipMemcpyAsync(counts, d_counts, 16, hipMemcpyDeviceToHost, st));
      SEG_CHK(hipStreamSynchronize(st));                         // host synchronisation 1 of 2: the sizes of the slot's arrays
      tp("scan + mark + counts");
      const uint32_t nseg = counts[0];
      s.nseg = nseg; s.ncseg = counts[1]; s.max_seg_count = counts[2]; s.all_ones = counts[3] == 0u;
      SEG_CHK(fmx_dev_alloc(&s.seg_feat, (size_t)std::max<uint32_t>(nseg, 1) * 4));
      SEG_CHK(fmx_dev_alloc(&s.seg_rel, (size_t)std::max<uint32_t>(nseg, 1) * 4));
      SEG_CHK(fmx_dev_alloc(&s.cseg, (size_t)std::max<uint32_t>(s.ncseg, 1) * 4));
      SEG_CHK(fmx_dev_alloc(&s.cdesc, (size_t)std::max<uint32_t>(s.ncseg, 1) * sizeof(CDesc)));
      hipLaunchKernelGGL(k_seg_fill, dim3(2048), dim3(256), 0, st, keys_b, flags, pos, nnz, s.row_ptr, B, fbits, s.seg_feat, s.seg_rel);
      hipLaunchKernelGGL(k_seg_compact, dim3(2048), dim3(256), 0, st, keys_b, head, cflag, cpos, nseg, s.d_batch_seg, s.cseg,
                         s.seg_feat, s.seg_rel, s.row_ptr, s.n_rows, B, fbits, s.cdesc, reinterpret_cast<const TEntry*>(vals_b));
      hipLaunchKernelGGL(k_seg_cbatch, dim3((n_batches + 256) / 256), dim3(256), 0, st, cpos, cflag, nseg, s.d_batch_seg, n_batches, s.d_cbatch);
      SEG_CHK(hipGetLastError());
    } else {
      s.nseg = 0; s.ncseg = 0; s.max_seg_count = 0; s.all_ones = false;
      SEG_CHK(hipMemsetAsync(s.d_batch_seg, 0, nb1 * 4, st));
      SEG_CHK(hipMemsetAsync(s.d_cbatch, 0, nb1 * 4, st));
      SEG_CHK(hipMemsetAsync(s.d_batch_base, 0, nb1 * 8, st));
      SEG_CHK(fmx_dev_alloc(&s.seg_feat, 4)); SEG_CHK(fmx_dev_alloc(&s.seg_rel, 4));
      SEG_CHK(fmx_dev_alloc(&s.cseg, 4));
      SEG_CHK(fmx_dev_alloc(&s.cdesc, sizeof(CDesc)));
    }
    s.batch_seg.resize(nb1); s.cbatch.resize(nb1); s.batch_base.resize(nb1);
    SEG_CHK(hipMemcpyAsync(s.batch_seg.data(), s.d_batch_seg, nb1 * 4, hipMemcpyDeviceToHost, st));
    SEG_CHK(hipMemcpyAsync(s.cbatch.data(), s.d_cbatch, nb1 * 4, hipMemcpyDeviceToHost, st));
    SEG_CHK(hipMemcpyAsync(s.batch_base.data(), s.d_batch_base, nb1 * 8, hipMemcpyDeviceToHost, st));
    SEG_CHK(hipStreamSynchronize(st));                           // host synchronisation 2 of 2: the per-batch tables
    tp("fill + compact + tables");
    s.seg_B = B;
  }
done:
#undef SEG_CHK
  if (scratch) fmx_dev_free(scratch);
  tp("frees");
  if (rc) free_segments(s);
  return rc;
}

// the device's error word after launches of k_scan_pit whose streams have been drained by the caller.  Bit 4 = a grid-wide exchange ran
// into its bound and one workgroup evaluated the batch's chain serially instead (fmx_kernels.h): the numbers are the rule's, so this is
// not a failure -- the epoch reports FMX_STAT_SCAN_FALLBACK and the handle takes the one-wavefront chain from now on.  Only bit 4 is
// cleared here (the hand-off's bits 1 and 2 belong to the epoch's own check).
extern "C++" int scan_error_check(fmx_handle h) {
  if (!h->pit_used) return FMX_OK;
  h->pit_used = false;
  uint32_t e = 0;
  HIPCHK(h, hipMemcpy(&e, h->handoff_err, sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (e & 4u) {
    const uint32_t rest_bits = e & ~4u;
    (void)hipMemcpy(h->handoff_err, &rest_bits, sizeof(uint32_t), hipMemcpyHostToDevice);
    h->scan_pit = false;
    h->run_status |= FMX_STAT_SCAN_FALLBACK;
  }
  return FMX_OK;
}

// do the handle's two streams run concurrently?  (k_concurrency_probe, once per handle: ~50 us when they do, ~40 ms when they do not)
extern "C++" bool streams_concurrent(fmx_handle h) {
  if (h->concurrent >= 0) return h->concurrent != 0;
  h->concurrent = 0;
  if (!h->probe_flags && fmx_dev_alloc(&h->probe_flags, 4 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return false; }
  unsigned res[4] = {0, 0, 0, 0};
  hipError_t e = hipMemsetAsync(h->probe_flags, 0, 4 * sizeof(unsigned), h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream2);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_concurrency_probe, dim3(1), dim3(64), 0, h->stream2, h->probe_flags, 1u);
    hipLaunchKernelGGL(k_concurrency_probe, dim3(1), dim3(64), 0, h->stream, h->probe_flags, 0u);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream2);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipMemcpy(res, h->probe_flags, sizeof(res), hipMemcpyDeviceToHost);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  h->concurrent = (res[2] == 1u && res[3] == 1u) ? 1 : 0;
  return h->concurrent != 0;
}

// workgroups of k_scan_pit that must be resident TOGETHER on the handle's device: every shard of a loopback group runs the whole
// recurrence itself, on the same device (round-5 advisor: 16 loopback shards x 32 workgroups on 256 CUs waited for each other)
static uint32_t pit_concurrent_kernels(fmx_handle h) {
  uint32_t n = 1;
  if (h->group && h->group->hs.size() > 1) {
    n = 0;
    for (fmx_handle m : h->group->hs) if (m && m->device == h->device) n++;
    if (n == 0) n = 1;
  }
  return n;
}

static int launch_scan(fmx_handle h, const float* rest, const float* target, uint32_t n_rows, uint32_t chunk,
                       const Hyper& hy, float* mult, hipStream_t st, const double* w0_in = nullptr, double* w0_out = nullptr,
                       const Handoff hw = Handoff{nullptr, 0ull, nullptr}, bool short_pit = false) {
  if (hy.k0) {
    const double* wi = w0_in ? w0_in : h->w0;
    double* wo = w0_out ? w0_out : h->w0;
    // round 5: the recurrence solved parallel in time (k_scan_pit: Newton on the whole path, affine prefix scans; up to 64 workgroups) for
    // micro-chunks that are powers of two up to 1024 on batches of 4097 .. 262 144 examples -- every default.  Same result as the chain to
    // fp32 rounding; 40-80 us per 262 144 examples whatever the micro-chunk where the chain takes 0.18 ms (256) .. 0.5 ms (32).
    // (the arrival counters of its exchanges are zeroed on the launch's own stream, in front of it)
    if (h->scan_pit && (n_rows > 4096u || (short_pit && n_rows >= 256u)) && chunk <= PIT_MAX_CHUNK && (chunk & (chunk - 1u)) == 0u) {
      // every workgroup of the launch spins on a grid-wide arrival counter: all of them must fit the device TOGETHER -- next to the same
      // kernel of the other shards on this device.  112 KiB of LDS per workgroup = one per CU (asked once per handle).
      if (h->pit_occ < 0) {
        int per_cu = 0;
        auto kf = k_scan_pit<false, 1>;
        (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PIT_LDS_BYTES);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kf, 256, PIT_LDS_BYTES) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        h->pit_occ = per_cu * h->num_cu;
      }
      const uint32_t piece_rows = PIT_MAX_ROWS;
      const uint32_t nwg_max = (std::min(n_rows, piece_rows) + pit_seg(chunk) - 1) / pit_seg(chunk);
      static const bool trace_pit = getenv("FMX_TRACE_PIT") != nullptr;
      if (trace_pit) fprintf(stderr, "[fmx pit] rows %u chunk %u: %u workgroups x %u kernels, device holds %d\n", n_rows, chunk, nwg_max, pit_concurrent_kernels(h), h->pit_occ);
      if ((uint64_t)nwg_max * pit_concurrent_kernels(h) <= (uint64_t)std::max(h->pit_occ, 0)) {
        // a batch longer than PIT_MAX_ROWS (an explicit batch, a HOGWILD macro-batch): pieces of PIT_MAX_ROWS examples, the bias handed from
        // piece to piece on the launch's stream (round-5 advisor: such batches silently took the one-wavefront chain, 8x the steps at the new
        // default micro-chunk); only the first piece waits for the hand-off counter, only the last one publishes
        const uint32_t n_pieces = (n_rows + piece_rows - 1) / piece_rows;
        if (n_pieces > 1 && !h->pit_tmp) HIPCHK(h, fmx_dev_alloc(&h->pit_tmp, 64 * sizeof(double)));
        if (n_pieces <= 64) {
          const PitSync ps{h->pit_ctr, h->pit_slots, h->handoff_err, h->pit_spins};   // (FMX_DEBUG_PIT_SPINS=0 at fmx_create: every exchange gives up at once -- the test of the fall-back)
          for (uint32_t pc = 0; pc < n_pieces; pc++) {
            const uint32_t r0 = pc * piece_rows, rn = std::min(piece_rows, n_rows - r0);
            const uint32_t nwg = (rn + pit_seg(chunk) - 1) / pit_seg(chunk);
            const double* pin = pc == 0 ? wi : h->pit_tmp + (pc - 1);
            double* pout = pc + 1 == n_pieces ? wo : h->pit_tmp + pc;
            Handoff ph{nullptr, 0ull, nullptr};                    // a piece in the middle: plain load, plain store, stream order
            if (hw.ctr) {
              if (pc == 0) ph = hw;                                // waits for the batch's rest[] (and stores with the hand-off store: harmless for pit_tmp)
              else if (pc + 1 == n_pieces) ph = Handoff{hw.ctr, 0ull, hw.err};   // publishes; nothing left to wait for
            }
            HIPCHK(h, hipMemsetAsync(h->pit_ctr, 0, (PIT_MAX_IT + 1) * sizeof(unsigned long long), st));
            float* mp = mult ? mult + r0 : nullptr;
#define FMX_PIT(WM, TK) do { auto kf = k_scan_pit<WM, TK>;                                                                       \
            if (!h->lds_raised.count((const void*)kf)) { HIPCHK(h, hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PIT_LDS_BYTES)); h->lds_raised.insert((const void*)kf); } \
            hipLaunchKernelGGL(kf, dim3(nwg), dim3(256), PIT_LDS_BYTES, st, rest + r0, target + r0, rn, chunk, hy, pin, pout, mp, ph, ps); } while (0)
            if (hy.task == 0) { if (mult) FMX_PIT(true, 0); else FMX_PIT(false, 0); }
            else              { if (mult) FMX_PIT(true, 1); else FMX_PIT(false, 1); }
#undef FMX_PIT
          }
          h->pit_used = true;
          h->run_status |= FMX_STAT_SCAN_PIT;
          HIPCHK(h, hipGetLastError());
          return FMX_OK;
        }
      }
    }
    h->run_status |= FMX_STAT_SCAN_SERIAL;
    // micro-chunks that are multiples of 256 examples: k_scan1 (one wavefront on the chain, four contiguous examples per lane, a 128 KiB
    // tile pipeline fed by the workgroup's four wavefronts); else, and for
    // batches of a few thousand rows (the 128 KiB-LDS workgroup costs more to place than the recurrence takes): one plain wavefront
    // ... or, round 5, the sub-piece form of the same kernel for micro-chunks of 16 / 32 / 64 / 128 examples (the default is now below 256)
    const bool sub = chunk == 16u || chunk == 32u || chunk == 64u || chunk == 128u;
    const bool tiled = n_rows > 8192u && ((chunk % 256u) == 0 || sub);
    // (function attributes are per-device state: the 128 KiB dynamic-LDS limit is raised once per handle, not per process)
#define FMX_SCAN1(WM, TK, C256) do { auto kf = k_scan1<WM, TK, C256>;                                                              \
      if (!h->lds_raised.count((const void*)kf)) { HIPCHK(h, hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCAN4_LDS_BYTES)); h->lds_raised.insert((const void*)kf); } \
      hipLaunchKernelGGL(kf, dim3(1), dim3(256), SCAN4_LDS_BYTES, st, rest, target, n_rows, chunk, hy, wi, wo, mult, hw); } while (0)
#define FMX_SCAN(WM, TK) do { \
      if (tiled && chunk == 256u) FMX_SCAN1(WM, TK, 256); \
      else if (tiled && chunk == 128u) FMX_SCAN1(WM, TK, 128); \
      else if (tiled && chunk == 64u)  FMX_SCAN1(WM, TK, 64); \
      else if (tiled && chunk == 32u)  FMX_SCAN1(WM, TK, 32); \
      else if (tiled && chunk == 16u)  FMX_SCAN1(WM, TK, 16); \
      else if (tiled)             FMX_SCAN1(WM, TK, 0); \
      else                        hipLaunchKernelGGL((k_scan<WM, TK>), dim3(1), dim3(64), 0, st, rest, target, n_rows, chunk, hy, wi, wo, mult, hw); } while (0)
    if (hy.task == 0) { if (mult) FMX_SCAN(true, 0); else FMX_SCAN(false, 0); }
    else              { if (mult) FMX_SCAN(true, 1); else FMX_SCAN(false, 1); }
#undef FMX_SCAN
#undef FMX_SCAN1
  } else if (mult) {
    hipLaunchKernelGGL(k_mult, dim3(std::min<uint32_t>((n_rows + 255) / 256, 2048)), dim3(256), 0, st, rest, target, n_rows, hy,
                       (const double*)nullptr, mult);
  }
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

// ---- FMX_FLAG_BIAS_LAG (split step): the w0 recurrence of batch b runs on stream2 while the main stream goes on ----
// Depth d = opts->bias_lag (>= 1): the multipliers of batch b use the bias as it was after the recurrence of batch b - d.
// Ring of R = d + 1 bias slots: slot (b + 1) % R holds the bias after batch b (all slots start as the current bias);
// the recurrence of batch b reads slot b % R and writes slot (b + 1) % R; the multipliers of batch b read slot
// (b - d + 1) % R, which is rewritten next by the recurrence of batch b + 1 -- enqueued after them.  d + 1 rest buffers.
extern "C++" int lag_flush(fmx_handle h) {                      // make h->w0 the current bias again
  LagState& L = h->lag;
  if (!L.active) return FMX_OK;
  if (L.in_stream) { HIPCHK(h, hipStreamSynchronize(L.in_stream)); L.in_stream = nullptr; }
  HIPCHK(h, hipStreamSynchronize(h->stream2));
  // on the handle's OWN stream, then drained: the streams are non-blocking, so a device-to-device hipMemcpy (null stream; it may return before
  // the copy has run) is not ordered against what the next epoch enqueues on h->stream -- its ring of bias slots was then, once in a few
  // thousand epochs, initialised from the bias of BEFORE this epoch (round 6: tests/test_gpu_fuzz.py seed 16 ended at w0 = -0.0076 instead of
  // -0.0163: exactly the second epoch started from the first epoch's start value)
  HIPCHK(h, hipMemcpyAsync(h->w0, h->w0_pp + (L.step % (L.depth + 1)), sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  L.active = false; L.step = 0;
  return scan_error_check(h);                                 // (callers that drive fmx_sgd_partial / _finish themselves and then read parameters)
}
// the driver's batch is small (below the size at which the recurrence gets its own stream): in-stream schedule, see sgd_finish_impl
static bool sgd_small_batch(const fmx_sgd_opts* opts) { return opts && opts->batch != 0 && opts->batch < 32768u && (opts->flags & FMX_FLAG_BIAS_LAG); }
// call BEFORE producing the rest buffer of this step on `st`; returns which of the d + 1 rest buffers to use
static int lag_prepare(fmx_handle h, hipStream_t st, uint32_t depth, int* slot, bool in_stream = false) {
  LagState& L = h->lag;
  if (!L.ev_rest) {
    HIPCHK(h, hipEventCreateWithFlags(&L.ev_rest, hipEventDisableTiming));
    for (auto& e : L.ev_scan) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  if (L.active && L.depth != depth) { int rc = lag_flush(h); if (rc) return rc; }
  if (!L.active) {
    L.depth = depth;
    for (uint32_t r = 0; r <= depth; r++) HIPCHK(h, hipMemcpyAsync(h->w0_pp + r, h->w0, sizeof(double), hipMemcpyDeviceToDevice, st));
    L.active = true; L.step = 0;
  } else if (L.step > depth && !in_stream) {
    HIPCHK(h, hipStreamWaitEvent(st, L.ev_scan[(L.step - depth - 1) % LagState::RING], 0));   // recurrence (step - d - 1) is done with this rest buffer
  }
  *slot = (int)(L.step % (depth + 1));
  return FMX_OK;
}
// call AFTER `rest` is complete on `st`: starts the recurrence on the side stream and leaves the multipliers of
// this batch (lagged bias) in h->mult on `st`
// the three pieces of a lagged step: `st` waits for the recurrence whose bias this batch's multipliers use; where that bias lives; the
// recurrence of THIS batch goes to the side stream behind whatever produced rest[] on `st`
static int lag_wait_bias(fmx_handle h, const Hyper& hy, hipStream_t st) {
  LagState& L = h->lag;
  if (hy.k0 && L.step >= L.depth) HIPCHK(h, hipStreamWaitEvent(st, L.ev_scan[(L.step - L.depth) % LagState::RING], 0));   // recurrence of batch b - d
  return FMX_OK;
}
static const double* lag_bias_slot(fmx_handle h) {
  const LagState& L = h->lag;
  const uint32_t R = L.depth + 1;
  return h->w0_pp + ((L.step + R - L.depth + 1) % R);
}
static int lag_start_scan(fmx_handle h, const float* rest, const float* target, uint32_t n_rows, uint32_t chunk, const Hyper& hy, hipStream_t st) {
  LagState& L = h->lag;
  const uint64_t b = L.step;
  const uint32_t R = L.depth + 1;
  if (hy.k0) {
    HIPCHK(h, hipEventRecord(L.ev_rest, st));
    HIPCHK(h, hipStreamWaitEvent(h->stream2, L.ev_rest, 0));
    int rc = launch_scan(h, rest, target, n_rows, chunk, hy, nullptr, h->stream2, h->w0_pp + (b % R), h->w0_pp + ((b + 1) % R));
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(L.ev_scan[b % LagState::RING], h->stream2));
  }
  return FMX_OK;
}
static int lag_step(fmx_handle h, const float* rest, const float* target, uint32_t n_rows, uint32_t chunk,
                    const Hyper& hy, hipStream_t st) {
  int rc = lag_start_scan(h, rest, target, n_rows, chunk, hy, st);
  if (rc) return rc;
  rc = lag_wait_bias(h, hy, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_mult, dim3(std::min<uint32_t>((n_rows + 255) / 256, 2048)), dim3(256), 0, st, rest, target, n_rows, hy,
                     lag_bias_slot(h), h->mult);
  HIPCHK(h, hipGetLastError());
  h->lag.step++;
  return FMX_OK;
}

// steps 2 and 3 of the minibatch rule for rows [row0,row0+n_rows); `seg_batch` = batch index when the rows are
// exactly one batch of the slot's segment structure (else -1: per-example apply only)
// short rows (a feature shard), library's choice of the update: example-major over the batch-unique features (k_apply_multi: several examples
// per wavefront, S_e read once per example) + k_apply_seg over the batch's deferred list
static bool short_row_update(fmx_handle h, const Slot& s, const fmx_sgd_opts* opts, int64_t seg_batch) {
  const int apply = opts ? opts->apply : FMX_APPLY_DEFAULT;
  return (apply == FMX_APPLY_DEFAULT || apply == FMX_APPLY_FUSED) && seg_batch >= 0 && s.cmask && !s.cbatch.empty() &&
         multi_group_size(s, h->KP) != 0;
}
static int launch_deferred(fmx_handle h, const Slot& s, const Hyper& hy, const float* S, size_t b, hipStream_t st) {
  const uint32_t c0 = s.cbatch[b], c1 = s.cbatch[b + 1];
  if (c1 > c0) {
    const uint32_t s0 = s.batch_seg[b], s1 = s.batch_seg[b + 1];
    const uint64_t base = s.batch_base[b];
    SegWork sw{s.t_ent + base, s.seg_feat + s0, s.seg_rel + s0, s.cseg + c0, c1 - c0, s1 - s0, (uint32_t)(s.batch_base[b + 1] - base), S, h->mult, s.cdesc + c0};
    KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_apply_seg<KP, 8, 16>), ((uint64_t)sw.nseg + 15) / 16, st, sw, h->tb, hy));
  }
  return FMX_OK;
}
// cpart != nullptr: `rest` has NOT been computed -- the short-row update derives it (and the multipliers) from S and cpart itself
static int sgd_finish_impl(fmx_handle h, const Slot& s, uint64_t row0, uint32_t n_rows, const float* S,
                           float* rest, const fmx_sgd_opts* opts, hipStream_t st,
                           hipEvent_t ev_a, hipEvent_t ev_b, int64_t seg_batch, const float* cpart = nullptr) {
  const Hyper hy = make_hyper(h->cfg);
  const uint32_t chunk = (opts && opts->w0_chunk) ? opts->w0_chunk : default_w0_chunk(h->cfg);
  const bool lag = opts && (opts->flags & FMX_FLAG_BIAS_LAG);
  int rc = FMX_OK;
  if (cpart) {
    // fused short-row step (bias-lag schedule on a feature shard): [wait for the bias of batch b - d] -> k_apply_multi<FUSED> (rest, multipliers,
    // update of the batch-unique features) -> the recurrence of this batch starts on the side stream -> the deferred features
    const uint32_t G = multi_group_size(s, h->KP);
    if (sgd_small_batch(opts)) {
      // small batches (what the stability cut leaves of rows with frequent features: BASELINE configs[2] runs 512 rows per batch): a batch is a
      // few microseconds of work and every call the host makes for it costs as much again -- the recurrence leaves the side stream and rides
      // in the launch of the deferred features (k_apply_seg_scan, as on an unsharded handle), no events: TWO launches per shard and batch.
      // Same rule, same ring of bias slots: what a batch reads does not depend on which stream wrote it.
      const LagState& L = h->lag;
      const uint32_t R = L.depth + 1;
      const size_t b = (size_t)seg_batch;
      if (ev_a) HIPCHK(h, hipEventRecord(ev_a, st));
      KP_SWITCH(h->KP, { if constexpr (KP >= 64 && KP <= 256) { FMX_LAUNCH_WAVES((k_apply_multi<KP, true>), ((uint64_t)n_rows + G - 1) / G, st, s.ent, s.row_ptr, s.target, row0, n_rows,
                                                                    h->tb, hy, lag_bias_slot(h), S, cpart, rest, h->mult, (const uint64_t*)s.cmask, G); } });
      const uint32_t c0 = s.cbatch[b], c1 = s.cbatch[b + 1];
      const uint32_t s0 = s.batch_seg[b], s1 = s.batch_seg[b + 1];
      const uint64_t base = s.batch_base[b];
      SegWork sw{s.t_ent + base, s.seg_feat + s0, s.seg_rel + s0, s.cseg + c0, c1 - c0, s1 - s0, (uint32_t)(s.batch_base[b + 1] - base), S, h->mult, s.cdesc + c0};
      sw.done_ctr = nullptr; sw.done_val = 0ull;
      const ScanSmall sc{rest, s.target + row0, h->w0_pp + (L.step % R), h->w0_pp + ((L.step + 1) % R), n_rows, chunk};
      KP_SWITCH(h->KP, hipLaunchKernelGGL((k_apply_seg_scan<KP, 8, 1>), dim3((sw.nseg + 3) / 4 + 1), dim3(256), 0, st, sw, h->tb, hy, sc));
      h->lag.step++;
      h->lag.in_stream = st;
      h->run_status |= FMX_STAT_SCAN_SERIAL;
      if (ev_b) HIPCHK(h, hipEventRecord(ev_b, st));
      HIPCHK(h, hipGetLastError());
      return FMX_OK;
    }
    rc = lag_wait_bias(h, hy, st);
    if (rc) return rc;
    if (ev_a) HIPCHK(h, hipEventRecord(ev_a, st));
    KP_SWITCH(h->KP, { if constexpr (KP >= 64 && KP <= 256) { FMX_LAUNCH_WAVES((k_apply_multi<KP, true>), ((uint64_t)n_rows + G - 1) / G, st, s.ent, s.row_ptr, s.target, row0, n_rows,
                                                                  h->tb, hy, lag_bias_slot(h), S, cpart, rest, h->mult, (const uint64_t*)s.cmask, G); } });
    HIPCHK(h, hipGetLastError());
    rc = lag_start_scan(h, rest, s.target + row0, n_rows, chunk, hy, st);
    if (rc) return rc;
    h->lag.step++;
    rc = launch_deferred(h, s, hy, S, (size_t)seg_batch, st);
    if (rc) return rc;
    if (ev_b) HIPCHK(h, hipEventRecord(ev_b, st));
    HIPCHK(h, hipGetLastError());
    return FMX_OK;
  }
  rc = lag ? lag_step(h, rest, s.target + row0, n_rows, chunk, hy, st)
           : launch_scan(h, rest, s.target + row0, n_rows, chunk, hy, h->mult, st);
  if (rc) return rc;
  if (short_row_update(h, s, opts, seg_batch)) {                // (exact chunk coupling: the multipliers came out of the recurrence)
    const uint32_t G = multi_group_size(s, h->KP);
    if (ev_a) HIPCHK(h, hipEventRecord(ev_a, st));
    KP_SWITCH(h->KP, { if constexpr (KP >= 64 && KP <= 256) { FMX_LAUNCH_WAVES((k_apply_multi<KP, false>), ((uint64_t)n_rows + G - 1) / G, st, s.ent, s.row_ptr, s.target, row0, n_rows,
                                                                  h->tb, hy, (const double*)h->w0, S, (const float*)nullptr, (float*)nullptr, h->mult, (const uint64_t*)s.cmask, G); } });
    HIPCHK(h, hipGetLastError());
    rc = launch_deferred(h, s, hy, S, (size_t)seg_batch, st);
    if (rc) return rc;
    if (ev_b) HIPCHK(h, hipEventRecord(ev_b, st));
    HIPCHK(h, hipGetLastError());
    return FMX_OK;
  }
  int apply = opts ? opts->apply : FMX_APPLY_DEFAULT;
  // split step, library's choice (DEFAULT / FUSED): the features that occur once in the batch are written back example-major
  // (k_fused<FUSED_APPLY>: the wavefront holds S_e, no gather per occurrence), the others by their owner (k_apply_seg over the
  // batch's cseg list) -- the second half of what k_fused<EXACT> + k_apply_seg do in one pass on an unsharded handle.
  // FMX_APPLY_SEGMENTED keeps the dense owner-per-feature pass.
  // ... where rows are long enough to fill a wavefront's gather: measured per rank of a P-way sharded step (scripts/gpu_shard_probe.py,
  // dense vs example-major second pass): 32 entries per row +5 %, 16 entries -4 %, 8 entries -18 %, 4 entries -17 %.
  const bool masked = (apply == FMX_APPLY_DEFAULT || apply == FMX_APPLY_FUSED) && s.cmask && !s.cbatch.empty() &&
                      s.nnz >= (uint64_t)24 * s.n_rows;
  if (apply == FMX_APPLY_DEFAULT || apply == FMX_APPLY_FUSED) apply = FMX_APPLY_SEGMENTED;   // split step: same rule, two passes
  if (apply == FMX_APPLY_SEGMENTED && seg_batch < 0) return fail(h, FMX_E_STATE, "segmented apply needs batch-aligned rows");
  if (ev_a) HIPCHK(h, hipEventRecord(ev_a, st));
  if (apply == FMX_APPLY_SEGMENTED && masked) {
    const size_t b = (size_t)seg_batch;
    KP_SWITCH(h->KP, rc = (launch_fused_zr<KP, FUSED_APPLY>(h, s, hy, row0, n_rows, st, (const double*)h->w0, nullptr,
                                                               (const uint64_t*)s.cmask, const_cast<float*>(S), h->mult)));
    if (rc) return rc;
    const uint32_t c0 = s.cbatch[b], c1 = s.cbatch[b + 1];
    if (c1 > c0) {
      const uint32_t s0 = s.batch_seg[b], s1 = s.batch_seg[b + 1];
      const uint64_t base = s.batch_base[b];
      SegWork sw{s.t_ent + base, s.seg_feat + s0, s.seg_rel + s0, s.cseg + c0, c1 - c0, s1 - s0, (uint32_t)(s.batch_base[b + 1] - base), S, h->mult, s.cdesc + c0};
      KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_apply_seg<KP, 8, 16>), ((uint64_t)sw.nseg + 15) / 16, st, sw, h->tb, hy));
    }
  } else if (apply == FMX_APPLY_SEGMENTED) {
    const uint32_t s0 = s.batch_seg[(size_t)seg_batch], s1 = s.batch_seg[(size_t)seg_batch + 1];
    const uint64_t base = s.batch_base[(size_t)seg_batch];
    const uint32_t bnnz = (uint32_t)(s.batch_base[(size_t)seg_batch + 1] - base);
    const uint32_t nseg = s1 - s0;
    if (nseg) {
      SegWork sw{s.t_ent + base, s.seg_feat + s0, s.seg_rel + s0, nullptr, nseg, nseg, bnnz, S, h->mult};
      KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_apply_seg<KP, 8, 64>), ((uint64_t)nseg + 63) / 64, st, sw, h->tb, hy));
    }
  } else if (apply == FMX_APPLY_ATOMIC) {
    KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_apply<KP, true>), n_rows, st,
                                       s.ent, s.row_ptr, row0, n_rows, h->tb, hy, S, h->mult));
  } else if (apply == FMX_APPLY_STORE) {
    KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_apply<KP, false>), n_rows, st,
                                       s.ent, s.row_ptr, row0, n_rows, h->tb, hy, S, h->mult));
  } else {
    return fail(h, FMX_E_ARG, "unknown apply mode %d", apply);
  }
  if (ev_b) HIPCHK(h, hipEventRecord(ev_b, st));
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

int fmx_sgd_finish(fmx_handle h, int slot, uint64_t row0, uint32_t n_rows, const float* d_partial,
                   const fmx_sgd_opts* opts, void* stream) {
  touch_w(h);
  int rc = check_slot(h, slot, true);
  if (rc) return rc;
  Slot& s = h->slots[slot];
  if (row0 + n_rows > s.n_rows) return fail(h, FMX_E_ARG, "fmx_sgd_finish: rows outside slot");
  if (!s.blocks.empty()) return fail(h, FMX_E_UNSUPPORTED, "relations are not supported with SGD");   // fm_learn_sgd.h:61-63
  if (!d_partial) return fail(h, FMX_E_ARG, "fmx_sgd_finish: d_partial is NULL");
  if (((uintptr_t)d_partial & 15u) != 0) return fail(h, FMX_E_ARG, "fmx_sgd_finish: d_partial must be 16-byte aligned");
  if (n_rows == 0) return FMX_OK;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  const bool lag = opts && (opts->flags & FMX_FLAG_BIAS_LAG);
  const uint32_t Bcap = (opts && opts->batch) ? std::max(opts->batch, n_rows) : n_rows;
  const uint32_t lag_depth = (opts && opts->bias_lag) ? opts->bias_lag : 1u;
  if (lag_depth > 4) return fail(h, FMX_E_ARG, "bias_lag %u: at most 4 batches", lag_depth);
  rc = ensure_scratch(h, n_rows, (size_t)Bcap * (lag_depth + 1));
  if (rc) return rc;
  const float* S = d_partial;
  const float* c = d_partial + (size_t)n_rows * h->KP;
  int64_t seg_batch = -1;
  const int apply = opts ? opts->apply : FMX_APPLY_DEFAULT;
  if (apply == FMX_APPLY_DEFAULT || apply == FMX_APPLY_SEGMENTED || apply == FMX_APPLY_FUSED) {
    // the driver walks the slot in batches of opts->batch rows (the last one may be short)
    const uint32_t B = (opts && opts->batch) ? opts->batch : n_rows;
    if (row0 % B != 0 || n_rows > B || n_rows != std::min<uint64_t>(B, s.n_rows - row0))
      return fail(h, FMX_E_ARG, "fmx_sgd_finish: rows [%llu,+%u) are not batch %u of the slot", (unsigned long long)row0, n_rows, B);
    rc = ensure_segments(h, h->slots[slot], B);
    if (rc) return rc;
    seg_batch = (int64_t)(row0 / B);
  }
  int rslot = 0;
  if (lag) { rc = lag_prepare(h, st, lag_depth, &rslot, sgd_small_batch(opts) && short_row_update(h, s, opts, seg_batch)); if (rc) return rc; }
  else { rc = lag_flush(h); if (rc) return rc; }
  float* rest_buf = h->rest + (size_t)rslot * Bcap;
  // short rows under the bias-lag schedule: rest_e and the multipliers come out of the update kernel itself (no k_rest_from_partial, no k_mult)
  if (lag && short_row_update(h, s, opts, seg_batch))
    return sgd_finish_impl(h, s, row0, n_rows, S, rest_buf, opts, st, nullptr, nullptr, seg_batch, c);
  KP_SWITCH(h->KP, hipLaunchKernelGGL((k_rest_from_partial<KP>), dim3(wave_grid((n_rows + Map<KP>::EPI - 1) / Map<KP>::EPI)),
                                        dim3(256), 0, st, S, c, n_rows, rest_buf));
  HIPCHK(h, hipGetLastError());
  return sgd_finish_impl(h, s, row0, n_rows, S, rest_buf, opts, st, nullptr, nullptr, seg_batch);
}

int fmx_predict_finish(fmx_handle h, uint32_t n_rows, const float* d_partial, float* d_yhat, void* stream) {
  if (!h || !d_partial || !d_yhat) return FMX_E_ARG;
  if (((uintptr_t)d_partial & 15u) != 0) return fail(h, FMX_E_ARG, "fmx_predict_finish: d_partial must be 16-byte aligned");
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  if (n_rows == 0) return FMX_OK;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  int rc = ensure_scratch(h, 0, n_rows);
  if (rc) return rc;
  const float* S = d_partial;
  const float* c = d_partial + (size_t)n_rows * h->KP;
  KP_SWITCH(h->KP, hipLaunchKernelGGL((k_rest_from_partial<KP>), dim3(wave_grid((n_rows + Map<KP>::EPI - 1) / Map<KP>::EPI)),
                                        dim3(256), 0, st, S, c, n_rows, h->rest));
  hipLaunchKernelGGL(k_yhat, dim3(std::min<uint32_t>((n_rows + 255) / 256, 2048)), dim3(256), 0, st,
                     h->rest, n_rows, h->cfg.k0, h->w0, d_yhat);
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

// ---- the XCD-resident epoch (fmx_xcd_kernels.h): every batch of the epoch in ONE launch whose workgroups sit on one die ----
// returns FMX_OK with *ran = true when the epoch has been computed; *ran = false: nothing was touched (not eligible, or the launch could not
// assemble its members -- the handle then stops trying) and the caller runs the two launches per batch
static int xcd_epoch(fmx_handle h, Slot& s, const Hyper& hy, uint32_t B, uint32_t d, uint32_t chunk, uint64_t n_batch, bool* ran) {
  *ran = false;
  if (!h->xcd || B > h->xcd_max_batch || (h->KP != 64 && h->KP != 128) || s.max_row > 64u || !s.d_cbatch || !s.d_batch_base) return FMX_OK;
  hipStream_t st = h->stream;
  const uint32_t Bc = std::min<uint32_t>(B, s.n_rows);
  if (!h->xcd_sync) HIPCHK(h, fmx_dev_alloc(&h->xcd_sync, (XCD_CTL_WORDS + XCD_MAX_MEMBERS) * sizeof(unsigned)));
  static const char* trace_file = getenv("FMX_XCD_TRACE");
  const uint32_t trace_batches = trace_file ? (uint32_t)std::min<uint64_t>(n_batch, 256) : 0u;
  if (trace_file && !h->xcd_trace) {
    HIPCHK(h, fmx_dev_alloc(&h->xcd_trace, 256 * 16 * sizeof(unsigned long long)));
  }
  if (trace_file) HIPCHK(h, hipMemsetAsync(h->xcd_trace, 0, 256 * 16 * sizeof(unsigned long long), st));
  XcdEpoch ep;
  ep.ent = s.ent; ep.row_ptr = s.row_ptr; ep.target = s.target; ep.cmask = (const uint64_t*)s.cmask;
  ep.fixed_nnz = s.fixed_nnz; ep.n_rows = s.n_rows; ep.B = B; ep.n_batch = (uint32_t)n_batch; ep.d = d; ep.chunk = chunk; ep.Bc = Bc;
  ep.cbatch = s.d_cbatch; ep.batch_base = s.d_batch_base; ep.t_ent = s.t_ent; ep.cdesc = s.cdesc;
  ep.S = h->partial; ep.mult = h->mult; ep.rest = h->rest; ep.w0_ring = h->w0_pp;
  { static const char* xf = getenv("FMX_XCD_FLAGS"); ep.flags = xf ? (uint32_t)strtoul(xf, nullptr, 10) : 0u; }
  ep.trace = trace_file ? h->xcd_trace : nullptr; ep.trace_batches = trace_batches;
  const XcdSync sy{h->xcd_sync, h->xcd_sync + XCD_CTL_WORDS, h->handoff_err, 1u << 22};
  HIPCHK(h, hipMemsetAsync(h->xcd_sync, 0, (XCD_CTL_WORDS + XCD_MAX_MEMBERS) * sizeof(unsigned), st));
  bool launched = false;
  uint32_t grid_used = 0;
#define FMX_XCD_LAUNCH1(KPV, ZRV, ON) do { auto kf = k_xcd_epoch<KPV, ZRV, ON>;                                                            \
    int per_cu = 0;                                                                                                                        \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kf, 256, 0) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; } \
    per_cu = std::min(per_cu, 4);        /* every workgroup of the launch must be resident: the member count is final only then */          \
    if (per_cu >= 1) { const uint32_t grid = std::min<uint32_t>((uint32_t)per_cu * (uint32_t)h->num_cu, 8u * XCD_MAX_MEMBERS);              \
      hipLaunchKernelGGL(kf, dim3(grid), dim3(256), 0, st, ep, h->tb, hy, sy); launched = true; grid_used = grid; } } while (0)
#define FMX_XCD_LAUNCH(KPV, ZRV) do { if (h->KP == KPV && zr == ZRV) { if (s.all_ones) FMX_XCD_LAUNCH1(KPV, ZRV, true); else FMX_XCD_LAUNCH1(KPV, ZRV, false); } } while (0)
  int zr = (h->KP == 64) ? fused_zr_select<64>(s.max_row) : fused_zr_select<128>(s.max_row);
  if (s.max_row > (uint32_t)zr) return FMX_OK;                                   // (rows beyond the register path)
  if (zr == 8) zr = 16;                                                         // (three instances per row width: 16, 40, 64 row slots)
  if (zr == 32) zr = 40;
  FMX_XCD_LAUNCH(64, 16);  FMX_XCD_LAUNCH(64, 40);  FMX_XCD_LAUNCH(64, 64);
  FMX_XCD_LAUNCH(128, 16); FMX_XCD_LAUNCH(128, 40); FMX_XCD_LAUNCH(128, 64);
#undef FMX_XCD_LAUNCH1
#undef FMX_XCD_LAUNCH
  if (!launched) return FMX_OK;
  HIPCHK(h, hipGetLastError());
  unsigned ctl[XCD_CTL_WORDS];
  HIPCHK(h, hipMemcpyAsync(ctl, h->xcd_sync, sizeof(ctl), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (ctl[4] != 1u) {            // the members never saw every workgroup start (the device is shared, partitioned or profiled): nothing was touched
    h->xcd = false;
    return FMX_OK;
  }
  uint32_t e = 0;
  HIPCHK(h, hipMemcpy(&e, h->handoff_err, sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (e & XCD_ERR_BARRIER) {
    const uint32_t rest_bits = e & ~XCD_ERR_BARRIER;
    (void)hipMemcpy(h->handoff_err, &rest_bits, sizeof(uint32_t), hipMemcpyHostToDevice);
    h->xcd = false;
    return fail(h, FMX_E_HIP, "the XCD-resident epoch gave up at a barrier (%u members on die %u): the epoch is incomplete -- reload the parameters; "
                              "the handle takes the two launches per batch from now on", ctl[1], ctl[0] ? ctl[0] - 1u : 0u);
  }
  if (trace_file) {
    std::vector<unsigned long long> tr((size_t)trace_batches * 16);
    HIPCHK(h, hipMemcpy(tr.data(), h->xcd_trace, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(trace_file, "a")) {
      fprintf(f, "# XCD-resident epoch: grid %u workgroups, die %u, %u member workgroups (%u wavefronts), batch %u, %llu batches; member 0 / wavefront 0, 10 ns ticks -> us\n"
                 "# columns (us): batch | examples: gather, rest | arrive | shadow: asked-for data here, then wait | owners: lists+multipliers, first sums rows, rest of the lockstep items, others | arrive | shadow: touches here, then wait | total\n",
              grid_used, ctl[0] - 1u, ctl[1], ctl[1] * 4u, B, (unsigned long long)n_batch);
      auto us = [](unsigned long long a, unsigned long long b) { return (b >= a && a) ? (double)(b - a) * 0.01 : 0.0; };
      for (uint32_t b = 0; b < trace_batches; b++) {
        const unsigned long long* t = tr.data() + (size_t)b * 16;
        fprintf(f, "%6u | %5.2f %5.2f | %5.2f | %5.2f %5.2f | %5.2f %5.2f %5.2f %5.2f | %5.2f | %5.2f %5.2f | %6.2f\n", b,
                us(t[0], t[8]), us(t[8], t[1]), us(t[1], t[2]), us(t[2], t[12]), us(t[12], t[3]),
                us(t[3], t[9]), us(t[9], t[10]), us(t[10], t[11]), us(t[11], t[4]), us(t[4], t[5]), us(t[5], t[13]), us(t[13], t[6]), us(t[0], t[6]));
      }
      fclose(f);
    }
  }
  *ran = true;
  return FMX_OK;
}

// MINIBATCH rule, FMX_APPLY_FUSED: per batch ONE pass over the examples (k_fused<FUSED_EXACT>: gather, predict, multiplier,
// write-back of every feature that occurs once in the batch) + k_apply_seg over the features that occur more than once.
// The multipliers of batch b use the bias as it was after the recurrence of batch b - d (d = opts->bias_lag >= 1; oracle
// fmo_sgd_epoch_minibatch_ex with bias_lag = d): the recurrence of batch b (k_scan, one workgroup) runs on the side stream
// under the launches of batches b+1 .. b+d-1 and only the launch of batch b+d waits for it.
static int sgd_epoch_fused(fmx_handle h, Slot& s, const fmx_sgd_opts* opts, const Hyper& hy, uint64_t* batches,
                           uint64_t* launches, uint64_t* deferred, bool* kept_wside) {
  const uint32_t B = opts->batch;                           // resolved by the caller (sgd_resolve_batch)
  const uint32_t d = opts->bias_lag ? opts->bias_lag : 1u;
  if (d > 4) return fail(h, FMX_E_ARG, "bias_lag %u: at most 4 batches", d);
  const uint32_t chunk = opts->w0_chunk ? opts->w0_chunk : default_w0_chunk(h->cfg);
  int rc = ensure_segments(h, s, B);
  if (rc) return rc;
  // FMX_FLAG_KEEP_WSIDE: this epoch keeps the slot's weight side stream current (for the evaluation passes that follow it)
  const bool keep_wside = (opts->flags & FMX_FLAG_KEEP_WSIDE) && hy.k1 && h->cfg.shard_world == 1 && s.blocks.empty();
  if (keep_wside) { rc = ensure_wside(h, s); if (rc) return rc; }
  const bool keep = keep_wside && s.wside != nullptr;
  *kept_wside = keep;
  const uint32_t Bc = std::min<uint32_t>(B, s.n_rows);
  rc = ensure_scratch(h, (size_t)Bc * 2, (size_t)Bc * (d + 1));       // S / mult of two consecutive batches, d (+ 1: hand-off) rest buffers
  if (rc) return rc;
  const uint64_t n_batch = ((uint64_t)s.n_rows + B - 1) / B;
  // small batches (what the stability cut leaves of data with frequent features): a batch is a few microseconds of work, so
  // the recurrence leaves the side stream and rides in the launch of the deferred features (k_apply_seg_scan) -- no events, two
  // launches per batch.  Same rule, same ring of bias slots: what a batch reads does not depend on which stream wrote it.
  const bool side = B >= 32768u;
  // large batches: the recurrence runs on the side stream.  handoff (default; FMX_HANDOFF=0 at fmx_create goes back to events): the two
  // streams are ordered by the DATA -- bias slots W[0 .. n_batch] that start as "pending", a counter the deferred-feature launch of a batch
  // advances (fmx_kernels.h: "Device-side hand-off") -- instead of four event packets per batch; W[i] = the bias after the recurrence of batch
  // i - 1, k_fused of batch b reads W[max(0, b - d + 1)], the recurrence of batch b reads W[b] and publishes W[b + 1].
  // (bias_lag 1 keeps the events: there every wavefront of k_fused WAITS for a slot the one-workgroup recurrence of the previous batch
  //  publishes, and nothing guarantees that kernel a CU once k_fused has filled the chip -- round-4 advisor; at lag >= 2 the slot is a batch old)
  // ... and only where the two streams really run side by side (streams_concurrent: probed once per handle): under serialised dispatch the
  // waits of the hand-off are satisfied by LATER launches and every batch would run into its bound
  const bool handoff = side && h->handoff && hy.k0 && d >= 2 && !(opts->flags & FMX_FLAG_EVENT_SYNC) && streams_concurrent(h);
  if (side && hy.k0 && !handoff) h->run_status |= FMX_STAT_EVENT_SYNC;
  if (!side && hy.k0) h->run_status |= FMX_STAT_SCAN_SERIAL;
  double* W = nullptr;
  unsigned long long hbase = 0;
  if (handoff) {
    if (h->w0_slots_cap < n_batch + 1) {
      if (h->w0_slots) HIPCHK(h, fmx_dev_free(h->w0_slots));
      h->w0_slots = nullptr; h->w0_slots_cap = 0;
      HIPCHK(h, fmx_dev_alloc(&h->w0_slots, (size_t)(n_batch + 1) * sizeof(double)));
      h->w0_slots_cap = n_batch + 1;
    }
    W = h->w0_slots;
    hbase = h->handoff_seq; h->handoff_seq += n_batch + 1;           // the counter only grows: no reset to order against the side stream
  }
  if (side && !handoff) while (h->ev_sync.size() < 2 * n_batch + 1) { hipEvent_t e; HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->ev_sync.push_back(e); }
  hipStream_t st = h->stream;
  HIPCHK(h, hipEventRecord(h->ev0, st));                    // do not bill the one-time bucketing to the epoch
  for (uint32_t r = 0; r < d; r++) HIPCHK(h, hipMemcpyAsync(h->w0_pp + r, h->w0, sizeof(double), hipMemcpyDeviceToDevice, st));
  if (handoff) {
    hipLaunchKernelGGL(k_handoff_arm, dim3((unsigned)std::min<uint64_t>((n_batch + 255) / 256, 64)), dim3(256), 0, st, W, (uint32_t)n_batch);
    HIPCHK(h, hipMemcpyAsync(W, h->w0, sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  auto seg_work = [&](uint64_t b, SegWork* sw) {                 // the deferred features of batch b
    const uint32_t c0 = s.cbatch[(size_t)b], c1 = s.cbatch[(size_t)b + 1];
    const uint32_t s0 = s.batch_seg[(size_t)b], s1 = s.batch_seg[(size_t)b + 1];
    const uint64_t base = s.batch_base[(size_t)b];
    sw->t_ent = s.t_ent + base; sw->seg_feat = s.seg_feat + s0; sw->seg_rel = s.seg_rel + s0; sw->seg_idx = s.cseg + c0;
    sw->nseg = c1 - c0; sw->nseg_batch = s1 - s0; sw->batch_nnz = (uint32_t)(s.batch_base[(size_t)b + 1] - base);
    sw->S = h->partial + (size_t)(b & 1) * Bc * (size_t)(h->KP + 1);
    sw->mult = h->mult + (size_t)(b & 1) * Bc;
    sw->cdesc = s.cdesc + c0;
    sw->done_ctr = nullptr; sw->done_val = 0ull;
  };
  if (!side && !keep) {
    bool ran = false;
    rc = xcd_epoch(h, s, hy, B, d, chunk, n_batch, &ran);
    if (rc) return rc;
    if (ran) {
      for (uint64_t b = 0; b < n_batch; b++) *deferred += s.cbatch[(size_t)b + 1] - s.cbatch[(size_t)b];
      *batches += n_batch; *launches += 1;
      h->run_status |= FMX_STAT_XCD_RESIDENT;
      if (hy.k0) HIPCHK(h, hipMemcpyAsync(h->w0, h->w0_pp + (n_batch % d), sizeof(double), hipMemcpyDeviceToDevice, st));
      return FMX_OK;
    }
    HIPCHK(h, hipEventRecord(h->ev0, st));                  // (not eligible / could not start: the epoch's clock starts over)
    for (uint32_t r = 0; r < d; r++) HIPCHK(h, hipMemcpyAsync(h->w0_pp + r, h->w0, sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  // small batches as ONE launch per batch across all dies (k_small_one, fmx_small_kernels.h; FMX_SMALL_ONE=0 at fmx_create: two launches)
  const bool small_one = !side && h->small_one && (h->KP == 8 || h->KP == 16 || h->KP == 32 || h->KP == 64 || h->KP == 128) && s.cdesc && Bc <= SMALL_ONE_MAX && !hy.sgda;
  int szr = 0, small_cap = 0;
  if (small_one) {
    if (h->KP < 64) szr = h->KP;                                  // (k <= 32: 64 / KP entries per row slot, KP slots hold any row of <= 64 entries)
    else {
      szr = (h->KP == 64) ? fused_zr_select<64>(s.max_row) : fused_zr_select<128>(s.max_row);
      if (szr == 8) szr = 16;                                     // (three instances per row width: 16, 40, 64 row slots)
      if (szr == 32) szr = 40;
    }
    if (!h->small_slots) HIPCHK(h, fmx_dev_alloc(&h->small_slots, ((size_t)3 * SMALL_ONE_MAX + 16) * sizeof(unsigned long long)));
    HIPCHK(h, hipMemsetAsync(h->small_slots, 0, (size_t)3 * SMALL_ONE_MAX * sizeof(unsigned long long), st));
    // ... and the tagged S_e elements (the scratch may hold anything, e.g. last epoch's elements under the same tags)
    HIPCHK(h, hipMemsetAsync(h->partial, 0, (size_t)Bc * (size_t)h->KP * sizeof(unsigned long long), st));
  }
  // FMX_SMALL_TRACE=<file>: device time stamps of the epoch's middle batch (fmx_small_kernels.h SmallSync::trace), appended to the file
  static const char* small_trace_file = getenv("FMX_SMALL_TRACE");
  unsigned long long* small_trace = nullptr;
  if (small_one && small_trace_file) {
    small_trace = h->small_slots + (size_t)3 * SMALL_ONE_MAX;
    unsigned long long init[16];
    for (int i = 0; i < 16; i++) init[i] = (i == 0 || i == 5) ? ~0ull : 0ull;
    HIPCHK(h, hipMemcpyAsync(small_trace, init, sizeof(init), hipMemcpyHostToDevice, st));
    HIPCHK(h, hipStreamSynchronize(st));
  }
  for (uint64_t b = 0; b < n_batch; b++) {
    const uint64_t row0 = b * B;
    const uint32_t nb = (uint32_t)std::min<uint64_t>(B, s.n_rows - row0);
    if (small_one) {
      SegWork sw;
      seg_work(b, &sw);
      *deferred += sw.nseg;
      float* S = h->partial;                                      // [nb][KP] x {tag, value}: 8 bytes per element (the two batches' worth of the scratch)
      sw.S = S;
      // the recurrence: at lag >= 2 one launch behind (k_small_one), the last launch catches up
      const bool defer = d >= 2u;
      const bool own = !defer || b + 1 == n_batch, prev = defer && b >= 1;
      const ScanSmall sc{nullptr, s.target + row0, h->w0_pp + (b % d), h->w0_pp + ((b + 1) % d), own ? nb : 0u, chunk};
      const ScanSmall sc_prev{nullptr, s.target + (prev ? row0 - B : 0), h->w0_pp + ((b + d - 1) % d), h->w0_pp + (b % d), prev ? B : 0u, chunk};
      unsigned long long* rs0 = h->small_slots + SMALL_ONE_MAX;
      const SmallSync sy{h->small_slots, rs0 + (size_t)(b & 1) * SMALL_ONE_MAX, rs0 + (size_t)((b + 1) & 1) * SMALL_ONE_MAX, (uint32_t)(b + 1), (uint32_t)b,
                         h->handoff_err, std::min<uint32_t>(h->pit_spins, 1u << 21), (small_trace && b == n_batch / 2) ? small_trace : nullptr};
      const uint32_t n_ex_wg = (nb + 3u) / 4u;
      bool launched = false;
#define FMX_SMALL1(KPV, ZRV) do { if (h->KP == KPV && szr == ZRV) {                                                                      \
        auto kf = k_small_one<KPV, ZRV>;                                                                                                   \
        if (!small_cap) {                                                                                                                  \
          int per_cu = 0;                                                                                                                  \
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kf, 256, 0) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; } \
          small_cap = std::max(1, per_cu * h->num_cu);                                                                                     \
        }                                                                                                                                  \
        const uint32_t room = (uint32_t)small_cap > n_ex_wg + 1u ? (uint32_t)small_cap - n_ex_wg - 1u : 1u;                                \
        const uint32_t own_wg = sw.nseg ? std::max(1u, std::min((sw.nseg + 3u) / 4u, room)) : 0u;                                          \
        hipLaunchKernelGGL(kf, dim3(n_ex_wg + own_wg + 1u), dim3(256), 0, st, s.ent, s.row_ptr, s.target, row0, nb, h->tb, hy,             \
                           (const double*)(h->w0_pp + ((b + 1) % d)), (const uint64_t*)s.cmask, S, s.fixed_nnz, sw, sc_prev, sc, sy, n_ex_wg,  \
                           (const uint64_t*)(keep ? s.lmask : nullptr), keep ? s.wside : (float*)nullptr);                                 \
        launched = true; } } while (0)
      FMX_SMALL1(8, 8);    FMX_SMALL1(16, 16);  FMX_SMALL1(32, 32);
      FMX_SMALL1(64, 16);  FMX_SMALL1(64, 40);  FMX_SMALL1(64, 64);
      FMX_SMALL1(128, 16); FMX_SMALL1(128, 40); FMX_SMALL1(128, 64);
#undef FMX_SMALL1
      if (launched) {
        HIPCHK(h, hipGetLastError());
        h->small_one_used = true;
        h->run_status |= FMX_STAT_SMALL_ONE;
        (*batches)++; (*launches)++;
        continue;
      }
    }
    // rest buffers: with events the launch of batch b waits for the recurrence of batch b - d, which was the last reader of buffer b % d;
    // with the hand-off k_fused writes rest[] BEFORE it asks for that bias, so it takes a buffer whose reader (batch b - d - 1) is known
    // to be done: every wavefront of the previous launch has consumed its result
    float* rest = h->rest + (size_t)(handoff ? b % (d + 1) : b % d) * Bc;
    float* S = h->partial + (size_t)(b & 1) * Bc * (size_t)(h->KP + 1);
    float* mult = h->mult + (size_t)(b & 1) * Bc;
    if (side && !handoff && b >= d) HIPCHK(h, hipStreamWaitEvent(st, h->ev_sync[2 * (b - d) + 1], 0));   // recurrence of batch b - d is done
    const double* w0_in = handoff ? W + (b + 1 >= d ? b + 1 - d : 0)
                                  : h->w0_pp + ((b + 1) % d);  // written by the recurrence of batch b - d (initial bias for b < d)
    KP_SWITCH(h->KP, { rc = launch_fused_zr<KP, FUSED_EXACT>(h, s, hy, row0, nb, st, w0_in, rest, s.cmask, S, mult, handoff ? h->handoff_err : nullptr, keep); });
    if (rc) return rc;
    HIPCHK(h, hipGetLastError());
    if (side && !handoff) HIPCHK(h, hipEventRecord(h->ev_sync[2 * b], st));
    SegWork sw;                                              // the batch's deferred features
    seg_work(b, &sw);
    *deferred += sw.nseg;
    if (!side) {
      // small batch: deferred features and bias recurrence in ONE launch (one workgroup per four segments + one for the recurrence)
      const ScanSmall sc{rest, s.target + row0, h->w0_pp + (b % d), h->w0_pp + ((b + 1) % d), nb, chunk};
      const uint32_t grid = (sw.nseg + 3) / 4 + 1;
      KP_SWITCH(h->KP, hipLaunchKernelGGL((k_apply_seg_scan<KP, 8, 1>), dim3(grid), dim3(256), 0, st, sw, h->tb, hy, sc));
      HIPCHK(h, hipGetLastError());
    } else {
      if (handoff) { sw.done_ctr = h->handoff_ctr; sw.done_val = hbase + b + 1; }
      if (handoff && !sw.nseg) hipLaunchKernelGGL(k_handoff_signal, dim3(1), dim3(64), 0, st, h->handoff_ctr, hbase + b + 1);
      if (sw.nseg) {
        // segments per wavefront of the deferred-feature pass: 16 (measured best at the bench shape: 8 / 16 / 32 / 64 -> 244.5 / 245.0 /
        // 243.0 / 241.0 M examples/s), fewer when the list is short so that the pass still spreads over the chip
        // (rows in flight per round: 8; round 6 measured 4 / 8 / 16 at the bench shape: 274.7 / 274.3 / 267.2 M examples/s)
        if (sw.nseg >= 16u * 2048u)     { KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_apply_seg<KP, 8, 16>), ((uint64_t)sw.nseg + 15) / 16, st, sw, h->tb, hy)); }
        else if (sw.nseg >= 4u * 2048u) { KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_apply_seg<KP, 8, 4>), ((uint64_t)sw.nseg + 3) / 4, st, sw, h->tb, hy)); }
        else                            { KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_apply_seg<KP, 8, 1>), (uint64_t)sw.nseg, st, sw, h->tb, hy)); }
        HIPCHK(h, hipGetLastError());
      }
      if (handoff) {
        rc = launch_scan(h, rest, s.target + row0, nb, chunk, hy, nullptr, h->stream2, W + b, W + b + 1,
                         Handoff{h->handoff_ctr, hbase + b + 1, h->handoff_err});
        if (rc) return rc;
      } else {
        HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_sync[2 * b], 0));
        rc = launch_scan(h, rest, s.target + row0, nb, chunk, hy, nullptr, h->stream2, h->w0_pp + (b % d), h->w0_pp + ((b + 1) % d));
        if (rc) return rc;
        HIPCHK(h, hipEventRecord(h->ev_sync[2 * b + 1], h->stream2));
      }
    }
    (*batches)++; (*launches)++;
  }
  if (small_trace) {
    unsigned long long tr[16];
    HIPCHK(h, hipMemcpyAsync(tr, small_trace, sizeof(tr), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    if (FILE* f = fopen(small_trace_file, "a")) {
      const double t0 = (double)tr[0];
      fprintf(f, "batch %llu of %llu (ns after the first example started): last example started %.0f, rows gathered %.0f, multiplier published %.0f, example done %.0f | "
                 "first owner started %.0f, last owner saw its tags %.0f, holds its S rows %.0f, done %.0f | recurrence done %.0f | last owner started %.0f, has its entry list %.0f\n",
              (unsigned long long)(n_batch / 2), (unsigned long long)n_batch, tr[1] - t0, tr[2] - t0, tr[3] - t0, tr[4] - t0, (double)tr[5] - t0, tr[6] - t0, tr[7] - t0, tr[8] - t0, tr[9] - t0, tr[10] - t0, tr[11] - t0);
      fclose(f);
    }
  }
  if (handoff) {                                              // ONE event per epoch: the last recurrence, then the bias goes home
    if (h->ev_sync.empty()) { hipEvent_t e; HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->ev_sync.push_back(e); }
    HIPCHK(h, hipEventRecord(h->ev_sync[0], h->stream2));
    HIPCHK(h, hipStreamWaitEvent(st, h->ev_sync[0], 0));
    HIPCHK(h, hipMemcpyAsync(h->w0, W + n_batch, sizeof(double), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(&h->handoff_err_host, h->handoff_err, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    return FMX_OK;
  }
  if (side) HIPCHK(h, hipStreamWaitEvent(st, h->ev_sync[2 * (n_batch - 1) + 1], 0));
  if (hy.k0) HIPCHK(h, hipMemcpyAsync(h->w0, h->w0_pp + (n_batch % d), sizeof(double), hipMemcpyDeviceToDevice, st));
  return FMX_OK;
}

// ---- FMX_SGD_SEQUENTIAL as conflict-free runs (fmx_seq_kernels.h) ----
// cuts the slot into maximal runs of consecutive rows that share no feature (once per slot; host wall-clock goes to setup_seconds)
static int ensure_runs(fmx_handle h, Slot& s) {
  if (!s.run_start.empty()) return FMX_OK;
  struct Acc { fmx_handle h; std::chrono::steady_clock::time_point t0;
               ~Acc() { h->setup_acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } acc_{h, std::chrono::steady_clock::now()};
  const uint32_t n = s.n_rows;
  if (n == 0 || n >= 0x7FFFFFFEu || s.nnz >= (1ull << 31)) { s.run_start = {0u, n}; s.run_single = {2u}; return FMX_OK; }   // (2: not cut -- the caller takes the kernel)
  hipStream_t st = h->stream;
  std::vector<uint32_t> prev(n, 0u);
  if (s.nnz) {
    uint32_t fbits = 1; while (fbits < 32 && (1ull << fbits) < std::max<uint64_t>(h->n_local, 2)) fbits++;
    size_t tmp_bytes = 0;
    HIPCHK(h, hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (int)s.nnz, 0, 32 + (int)fbits, st));
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_a = 0, o_b = al(s.nnz * 8), o_p = o_b + al(s.nnz * 8), o_t = o_p + al((size_t)n * 4), total = o_t + al(std::max<size_t>(tmp_bytes, 256));
    char* scratch = nullptr;
    HIPCHK(h, fmx_dev_alloc(&scratch, total));
    uint64_t* ka = (uint64_t*)(scratch + o_a); uint64_t* kb = (uint64_t*)(scratch + o_b); uint32_t* d_prev = (uint32_t*)(scratch + o_p);
    hipError_t e = hipMemsetAsync(d_prev, 0, (size_t)n * 4, st);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_run_keys, dim3(wave_grid(n)), dim3(256), 0, st, s.ent, s.row_ptr, n, ka);
      size_t tb_ = tmp_bytes;
      e = hipcub::DeviceRadixSort::SortKeys(scratch + o_t, tb_, ka, kb, (int)s.nnz, 0, 32 + (int)fbits, st);
    }
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_run_prev, dim3(2048), dim3(256), 0, st, kb, s.nnz, d_prev);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(prev.data(), d_prev, (size_t)n * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    fmx_dev_free(scratch);
    if (e != hipSuccess) return fail(h, FMX_E_HIP, "ensure_runs failed: %s", hipGetErrorString(e));
  }
  // greedy, in file order: a run ends in front of the first row that shares a feature with a row of the run (prev - 1 >= start), in front of
  // and behind a row that repeats an id, and at 4096 rows (one wavefront evaluates the run's bias recurrence)
  std::vector<uint32_t> starts; std::vector<uint8_t> single;
  uint32_t start = 0;
  for (uint32_t r = 0; r < n; r++) {
    const bool rep = (prev[r] & 0x80000000u) != 0;
    const uint32_t p = prev[r] & 0x7FFFFFFFu;                     // 1 + the latest earlier row sharing a feature, 0: none
    if (rep) {
      if (r > start) { starts.push_back(start); single.push_back(0); }
      starts.push_back(r); single.push_back(1);
      start = r + 1;
    } else if ((p > start && r > start) || r - start >= 4096u) {
      starts.push_back(start); single.push_back(0);
      start = r;
    }
  }
  if (start < n) { starts.push_back(start); single.push_back(0); }
  starts.push_back(n);
  s.run_start.swap(starts); s.run_single.swap(single);
  return FMX_OK;
}
// a run as ONE launch (k_run_fused): 1 = launched, 0 = not this run (rows too long for the registers, no instance, the device does not hold
// the run's workgroups at once), < 0 = error
static int launch_run_one(fmx_handle h, const Slot& s, const Hyper& hy, uint32_t row0, uint32_t nb, int zr, const double* bias_in, double* bias_out,
                          unsigned long long* slots, uint32_t tag) {
  hipStream_t st = h->stream;
  const size_t lds = (size_t)nb * 5 * sizeof(float);
  const dim3 grid((nb + 3u) / 4u);
  const RunSync rs{slots, tag, h->handoff_err, std::min<uint32_t>(h->pit_spins, 1u << 21)};   // (FMX_DEBUG_PIT_SPINS=0 at fmx_create: every poll gives up at once)
  int launched = 0;
#define FMX_RUN1(KPV, ZRV, TK) do { if (h->KP == KPV && zr == ZRV && hy.task == TK) {                                                    \
    auto kf = k_run_fused<KPV, ZRV, TK>;                                                                                                 \
    auto it = h->run_one_occ.find((const void*)kf);                                                                                      \
    if (it == h->run_one_occ.end()) {                                                                                                    \
      int per_cu = 0;                                                                                                                    \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kf, 256, (size_t)RUN_ONE_MAX * 5 * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; } \
      it = h->run_one_occ.emplace((const void*)kf, per_cu * h->num_cu).first;                                                            \
    }                                                                                                                                    \
    if ((int)grid.x <= it->second) {                                                                                                     \
      hipLaunchKernelGGL(kf, grid, dim3(256), lds, st, s.ent, s.row_ptr, s.target, (uint64_t)row0, nb, h->tb, hy, bias_in, bias_out, rs, s.fixed_nnz); \
      launched = 1; } } } while (0)
  FMX_RUN1(8, 8, 0);    FMX_RUN1(16, 16, 0);  FMX_RUN1(32, 32, 0);  FMX_RUN1(8, 8, 1);    FMX_RUN1(16, 16, 1);  FMX_RUN1(32, 32, 1);
  FMX_RUN1(64, 16, 0);  FMX_RUN1(64, 40, 0);  FMX_RUN1(64, 64, 0);  FMX_RUN1(64, 16, 1);  FMX_RUN1(64, 40, 1);  FMX_RUN1(64, 64, 1);
  FMX_RUN1(128, 16, 0); FMX_RUN1(128, 40, 0); FMX_RUN1(128, 64, 0); FMX_RUN1(128, 16, 1); FMX_RUN1(128, 40, 1); FMX_RUN1(128, 64, 1);
#undef FMX_RUN1
  return launched;
}

// one epoch over the runs.  A run of up to RUN_ONE_MAX rows that fit the registers is ONE launch (k_run_fused: the rows stay in the wavefronts'
// registers across the run's bias recurrence); up to RUN_FUSED_MAX rows TWO: the sums, then the update whose every workgroup solves the
// recurrence for itself (k_run_apply); a longer one three (sums, k_scan_pit on one workgroup at micro-chunk 1, update).
// FMX_SEQ_RUNS_FUSED=0: always three, with the one-wavefront chain (what the first version did); FMX_SEQ_RUNS_ONE=0: never one.
static int seq_runs_epoch(fmx_handle h, Slot& s, const Hyper& hy) {
  hipStream_t st = h->stream;
  static const bool fused = []() { const char* e = getenv("FMX_SEQ_RUNS_FUSED"); return !(e && e[0] == '0'); }();
  static const bool one_env = []() { const char* e = getenv("FMX_SEQ_RUNS_ONE"); return !(e && e[0] == '0'); }();
  uint32_t longest = 1;
  for (size_t i = 0; i + 1 < s.run_start.size(); i++) longest = std::max(longest, s.run_start[i + 1] - s.run_start[i]);
  int rc = ensure_scratch(h, longest, 0);
  if (rc) return rc;
  if (!h->pit_tmp) HIPCHK(h, fmx_dev_alloc(&h->pit_tmp, 64 * sizeof(double)));
  double* bias[2] = {h->w0, h->pit_tmp + 48};                     // the fused forms read the bias in one slot and leave it in the other
  if (!h->run_slots) HIPCHK(h, fmx_dev_alloc(&h->run_slots, (size_t)RUN_ONE_MAX * sizeof(unsigned long long)));
  unsigned long long* slots = h->run_slots;                       // {tag, rest_e} of a one-launch run's examples
  int zr = 0;
  bool one = fused && one_env && h->run_one && (h->KP == 8 || h->KP == 16 || h->KP == 32 || h->KP == 64 || h->KP == 128) && s.max_row <= 64u;
  if (one && h->KP < 64) zr = h->KP;                              // (k <= 32: 64 / KP entries per row slot, KP slots hold any row of <= 64 entries)
  else if (one) {
    zr = (h->KP == 64) ? fused_zr_select<64>(s.max_row) : fused_zr_select<128>(s.max_row);
    if (s.max_row > (uint32_t)zr) one = false;                    // (rows beyond the register path)
    if (zr == 8) zr = 16;                                         // (three instances per row width: 16, 40, 64 row slots)
    if (zr == 32) zr = 40;
  }
  if (one) HIPCHK(h, hipMemsetAsync(slots, 0, (size_t)RUN_ONE_MAX * sizeof(unsigned long long), st));
  int cur = 0;
  for (size_t i = 0; i + 1 < s.run_start.size(); i++) {
    const uint32_t row0 = s.run_start[i], nb = s.run_start[i + 1] - row0;
    if (s.run_single[i]) {                                        // a row that repeats an id: entry by entry (fm_sgd.h:44-50)
      KP_SWITCH(h->KP, hipLaunchKernelGGL((k_sequential<KP>), dim3(1), dim3(64), 0, st, s.ent, s.row_ptr + row0, s.target + row0, nb, h->tb, hy, bias[cur]));
      continue;
    }
    if (one && nb <= RUN_ONE_MAX) {
      rc = launch_run_one(h, s, hy, row0, nb, zr, bias[cur], bias[cur ^ 1], slots, (uint32_t)i + 1u);
      if (rc < 0) return rc;
      if (rc == 1) { h->run_one_used = true; if (hy.k0) cur ^= 1; continue; }
    }
    float* S = h->partial;
    float* rest = S + (size_t)nb * h->KP;
    KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_rowsums<KP, true, true>), nb, st, s.ent, s.row_ptr, (uint64_t)row0, nb, h->tb, h->cfg.k1, S, rest, (const float*)nullptr));
    if (fused && nb <= RUN_FUSED_MAX) {
      const dim3 grid((nb + 3u) / 4u);
      const size_t lds = (size_t)nb * 5 * sizeof(float);
      if (hy.task == 0) { KP_SWITCH(h->KP, hipLaunchKernelGGL((k_run_apply<KP, 0>), grid, dim3(256), lds, st, s.ent, s.row_ptr, (uint64_t)row0, nb, h->tb, hy, S, rest, s.target, bias[cur], bias[cur ^ 1])); }
      else              { KP_SWITCH(h->KP, hipLaunchKernelGGL((k_run_apply<KP, 1>), grid, dim3(256), lds, st, s.ent, s.row_ptr, (uint64_t)row0, nb, h->tb, hy, S, rest, s.target, bias[cur], bias[cur ^ 1])); }
      if (hy.k0) cur ^= 1;
      continue;
    }
    rc = launch_scan(h, rest, s.target + row0, nb, 1u, hy, h->mult, st, bias[cur], bias[cur], Handoff{nullptr, 0ull, nullptr}, fused);
    if (rc) return rc;
    KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_apply<KP, false>), nb, st, s.ent, s.row_ptr, (uint64_t)row0, nb, h->tb, hy, S, h->mult));
  }
  if (cur) HIPCHK(h, hipMemcpyAsync(h->w0, bias[1], sizeof(double), hipMemcpyDeviceToDevice, st));
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

int fmx_sgd_epoch(fmx_handle h, int slot, const fmx_sgd_opts* opts, fmx_epoch_stats* stats) {
  int rc = check_slot(h, slot, true);
  if (rc) return rc;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  if (!opts) return fail(h, FMX_E_ARG, "fmx_sgd_epoch: opts is NULL");
  HIPCHK(h, hipSetDevice(h->device));
  Slot& s = h->slots[slot];
  if (stats) memset(stats, 0, sizeof(*stats));
  if (s.n_rows == 0) return FMX_OK;
  if (!s.blocks.empty()) return fail(h, FMX_E_UNSUPPORTED, "relations are not supported with SGD");   // fm_learn_sgd.h:61-63
  if (!(h->group && !h->owns_group)) { h->setup_acc = 0.0; h->run_status = 0; }   // (a group clears its members' before the epoch)
  if (h->cfg.shard_world > 1 && opts->mode != FMX_SGD_MINIBATCH)
    return fail(h, FMX_E_UNSUPPORTED, "feature shards train with FMX_SGD_MINIBATCH (the split step)");
  if (h->cfg.shard_world > 1 || h->comm) return comm_sgd_epoch(h, slot, opts, stats);   // (a communicator of one rank drives the same schedule)
  const Hyper hy = make_hyper(h->cfg);
  fmx_sgd_opts ropts = *opts;                               // MINIBATCH: `batch` resolved against the rows' collision mass
  fmx_batch_info bi; memset(&bi, 0, sizeof(bi));
  if (opts->mode == FMX_SGD_MINIBATCH) {
    rc = sgd_resolve_batch(h, s, opts, &bi);
    if (rc) return rc;
    if ((opts->flags & FMX_FLAG_REJECT_UNSTABLE) && (bi.status & FMX_STAT_UNSTABLE)) {
      if (opts->batch == 0)       // the library's own choice is at its floor and still unstable: say THAT, not "let the library choose"
        return fail(h, FMX_E_ARG, "these rows are too dense for the batch rule: at the smallest batch the library takes (%u rows) learn_rate * "
                                  "curvature * batch * collision mass = %.3g > 2 (collision mass %.4g) -- lower learn_rate or train with "
                                  "FMX_SGD_SEQUENTIAL", bi.batch, bi.batch_gain, bi.collision_mass);
      return fail(h, FMX_E_ARG, "batch %u on these rows: learn_rate * curvature * batch * collision mass = %.3g > 2 -- the batch rule "
                                "diverges (collision mass %.4g; batch 0 lets the library choose)", bi.batch, bi.batch_gain, bi.collision_mass);
    }
    ropts.batch = bi.batch;
    opts = &ropts;
  } else if (opts->mode == FMX_SGD_HOGWILD) {
    // the asynchronous step freezes nothing, but an update becomes visible only when its wavefront retires: the rows in flight
    // (5 wavefronts per SIMD x 4 SIMDs x CUs on this part, one example each) are the window the criterion of MINIBATCH applies to
    rc = ensure_coll_mass(h, s);
    if (rc) return rc;
    const uint32_t in_flight = (uint32_t)std::min<uint64_t>(s.n_rows, (uint64_t)20 * (uint64_t)h->num_cu);
    resolve_batch(h->cfg, s.coll_mass, in_flight, in_flight, 1.0, &bi);
    if ((opts->flags & FMX_FLAG_REJECT_UNSTABLE) && (bi.status & FMX_STAT_UNSTABLE))
      return fail(h, FMX_E_ARG, "HOGWILD on these rows: learn_rate * curvature * %u rows in flight * collision mass = %.3g > 2 -- the asynchronous "
                                "step diverges (collision mass %.4g; use FMX_SGD_MINIBATCH with batch 0)", in_flight, bi.batch_gain, bi.collision_mass);
  }
  const bool timed = (opts->flags & FMX_FLAG_TIME_MAIN_KERNEL) != 0;
  uint64_t batches = 0, main_launches = 0, deferred = 0;
  bool kept_wside = false;
  touch_w(h);                                                // (every side stream is stale from here on; this epoch may re-validate ITS slot's)
  size_t ev_used = 0;
  auto get_event = [&](hipEvent_t* ev) -> hipError_t {
    if (ev_used == h->ev_pool.size()) { hipEvent_t e; hipError_t er = hipEventCreate(&e); if (er != hipSuccess) return er; h->ev_pool.push_back(e); }
    *ev = h->ev_pool[ev_used++];
    return hipSuccess;
  };
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  if (opts->mode == FMX_SGD_SEQUENTIAL) {
    // rows of 64 / 128 lanes: a row at a time with the next example's rows in flight (fmx_seq_kernels.h: 20 k -> ~1 M examples/s on the
    // reference's own trajectory); FMX_SEQ_ROWS=0 and the other row widths: entry by entry
    static const bool seq_rows = []() { const char* e = getenv("FMX_SEQ_ROWS"); return !(e && e[0] == '0'); }();
    // where consecutive rows rarely share a feature: the same trajectory as conflict-free runs at batch speed (fmx_seq_kernels.h); taken when
    // the slot's runs average >= 16 rows (FMX_SEQ_RUNS=0: never, =1: always)
    const char* sr = getenv("FMX_SEQ_RUNS");
    bool use_runs = !(sr && sr[0] == '0');
    if (use_runs) {
      rc = ensure_runs(h, s);
      if (rc) return rc;
      const size_t n_runs = s.run_start.size() - 1;
      use_runs = !(s.run_single.size() == 1 && s.run_single[0] == 2) && ((sr && sr[0] == '1') || (uint64_t)s.n_rows >= 16ull * n_runs);
      HIPCHK(h, hipEventRecord(h->ev0, h->stream));             // (the one-time cut is not the epoch's time)
    }
    static const bool seq_wg = []() { const char* e = getenv("FMX_SEQ_WG"); return !(e && e[0] == '0'); }();
    if (use_runs) {
      rc = seq_runs_epoch(h, s, hy);
      if (rc) return rc;
      h->run_status |= FMX_STAT_SEQ_RUNS;
      batches = s.run_start.size() - 1; main_launches = 1;
    } else
    if (seq_rows && seq_wg && h->KP <= 128) {
      // eight wavefronts on each example (fmx_seq_kernels.h k_sequential_wg); FMX_SEQ_WG=0: one wavefront, a row at a time.
      // Fewer than 33 factors run the 64-lane instance too: a row access masks the lanes beyond the row (tb.rs), whatever the lane mapping
      if (h->KP <= 64) {
        auto kf = k_sequential_wg<64>;
        if (!h->lds_raised.count((const void*)kf)) { HIPCHK(h, hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SeqLds<64>))); h->lds_raised.insert((const void*)kf); }
        hipLaunchKernelGGL(kf, dim3(1), dim3(64 * SEQ_W), sizeof(SeqLds<64>), h->stream, s.ent, s.row_ptr, s.target, s.n_rows, s.nnz, h->tb, hy, h->w0);
      } else {
        auto kf = k_sequential_wg<128>;
        if (!h->lds_raised.count((const void*)kf)) { HIPCHK(h, hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SeqLds<128>))); h->lds_raised.insert((const void*)kf); }
        hipLaunchKernelGGL(kf, dim3(1), dim3(64 * SEQ_W), sizeof(SeqLds<128>), h->stream, s.ent, s.row_ptr, s.target, s.n_rows, s.nnz, h->tb, hy, h->w0);
      }
    } else if (seq_rows && h->KP <= 128) {
      const bool wide = s.max_row > 32u;
      if (h->KP <= 64) { if (wide) hipLaunchKernelGGL((k_sequential_rows<64, 64>), dim3(1), dim3(64), 0, h->stream, s.ent, s.row_ptr, s.target, s.n_rows, h->tb, hy, h->w0);
                         else      hipLaunchKernelGGL((k_sequential_rows<64, 32>), dim3(1), dim3(64), 0, h->stream, s.ent, s.row_ptr, s.target, s.n_rows, h->tb, hy, h->w0); }
      else             { if (wide) hipLaunchKernelGGL((k_sequential_rows<128, 64>), dim3(1), dim3(64), 0, h->stream, s.ent, s.row_ptr, s.target, s.n_rows, h->tb, hy, h->w0);
                         else      hipLaunchKernelGGL((k_sequential_rows<128, 32>), dim3(1), dim3(64), 0, h->stream, s.ent, s.row_ptr, s.target, s.n_rows, h->tb, hy, h->w0); }
    } else
    KP_SWITCH(h->KP, hipLaunchKernelGGL((k_sequential<KP>), dim3(1), dim3(64), 0, h->stream, s.ent, s.row_ptr,
                                          s.target, s.n_rows, h->tb, hy, h->w0));
    HIPCHK(h, hipGetLastError());
    if (!use_runs) batches = s.n_rows;
    main_launches = 1;
  } else if (opts->mode == FMX_SGD_HOGWILD) {
    if (opts->apply == FMX_APPLY_SEGMENTED) return fail(h, FMX_E_ARG, "HOGWILD has no segmented apply");
    // rows per launch M: w0 is frozen inside a launch.  The bias recurrence of launch i (k_scan, one wavefront) runs
    // on a side stream WHILE launches i+1, i+2 stream; launch i reads the w0 produced by scan i-3 (a ring of 3
    // slots / rest buffers, so the result does not depend on timing and a slow scan has two launches of slack).
    const uint32_t M = opts->batch ? opts->batch : 262144u;
    const uint32_t chunk = opts->w0_chunk ? opts->w0_chunk : default_w0_chunk(h->cfg);
    const uint32_t cap = std::min<uint32_t>(M, s.n_rows);
    rc = ensure_scratch(h, 0, (size_t)cap * 3);
    if (rc) return rc;
    const uint64_t n_launch = ((uint64_t)s.n_rows + M - 1) / M;
    while (h->ev_sync.size() < 2 * n_launch) { hipEvent_t e; HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->ev_sync.push_back(e); }
    for (int r = 0; r < 3; r++) HIPCHK(h, hipMemcpyAsync(h->w0_pp + r, h->w0, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    for (uint64_t i = 0; i < n_launch; i++) {
      const uint64_t row0 = i * M;
      const uint32_t nb = (uint32_t)std::min<uint64_t>(M, s.n_rows - row0);
      float* rest = h->rest + (size_t)(i % 3) * cap;
      hipStream_t fs = h->stream;
      if (i >= 3) HIPCHK(h, hipStreamWaitEvent(fs, h->ev_sync[2 * (i - 3) + 1], 0));   // scan i-3 done
      hipEvent_t ea = nullptr, eb = nullptr;
      // (no per-launch events here: a timing event between two launches costs ~13 % on this path; the epoch is
      //  bracketed by ev0/ev1 on the launch stream and the average launch time is epoch time / launches)
      main_launches++;
      const double* w0_in = h->w0_pp + ((i + 1) % 3);      // slot written by scan i-3 (initial value for i < 3)
      if (opts->apply == FMX_APPLY_ATOMIC) {
        KP_SWITCH(h->KP, { rc = launch_fused_zr<KP, FUSED_ATOMIC>(h, s, hy, row0, nb, fs, w0_in, rest); });
      } else {
        KP_SWITCH(h->KP, { rc = launch_fused_zr<KP, FUSED_STORE>(h, s, hy, row0, nb, fs, w0_in, rest); });
      }
      if (rc) return rc;
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipEventRecord(h->ev_sync[2 * i], fs));
      HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_sync[2 * i], 0));
      rc = launch_scan(h, rest, s.target + row0, nb, chunk, hy, nullptr, h->stream2, h->w0_pp + (i % 3), h->w0_pp + ((i + 1) % 3));
      if (rc) return rc;
      HIPCHK(h, hipEventRecord(h->ev_sync[2 * i + 1], h->stream2));
      batches++;
    }
    // stream2 is in order: its last event covers every scan, and scan i waited for launch i
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_sync[2 * (n_launch - 1) + 1], 0));
    if (hy.k0) HIPCHK(h, hipMemcpyAsync(h->w0, h->w0_pp + (n_launch % 3), sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  } else if (opts->mode == FMX_SGD_MINIBATCH && opts->apply == FMX_APPLY_FUSED) {
    rc = sgd_epoch_fused(h, s, opts, hy, &batches, &main_launches, &deferred, &kept_wside);
    if (rc) return rc;
  } else if (opts->mode == FMX_SGD_MINIBATCH) {
    const uint32_t B = opts->batch;
    const bool lag = (opts->flags & FMX_FLAG_BIAS_LAG) != 0;
    const uint32_t lag_depth = opts->bias_lag ? opts->bias_lag : 1u;
    if (lag_depth > 4) return fail(h, FMX_E_ARG, "bias_lag %u: at most 4 batches", lag_depth);
    const uint32_t Bc = std::min<uint32_t>(B, s.n_rows);
    rc = ensure_scratch(h, (size_t)Bc * (lag ? lag_depth + 1 : 1), 0);
    if (rc) return rc;
    const bool segmented = (opts->apply == FMX_APPLY_DEFAULT || opts->apply == FMX_APPLY_SEGMENTED);
    if (segmented) {
      rc = ensure_segments(h, h->slots[slot], B);
      if (rc) return rc;
      HIPCHK(h, hipEventRecord(h->ev0, h->stream));           // do not bill the one-time bucketing to the epoch
    }
    for (uint64_t row0 = 0; row0 < s.n_rows; row0 += B) {
      const uint32_t nb = (uint32_t)std::min<uint64_t>(B, s.n_rows - row0);
      int pslot = 0;
      if (lag) { rc = lag_prepare(h, h->stream, lag_depth, &pslot); if (rc) return rc; }
      float* S = h->partial + (size_t)pslot * Bc * (size_t)(h->KP + 1);
      float* rest = S + (size_t)nb * h->KP;
      KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_rowsums<KP, true, true>), nb, h->stream,
                                         s.ent, s.row_ptr, row0, nb, h->tb, h->cfg.k1, S, rest, (const float*)nullptr));
      hipEvent_t ea = nullptr, eb = nullptr;
      if (timed) { HIPCHK(h, get_event(&ea)); HIPCHK(h, get_event(&eb)); main_launches++; }
      rc = sgd_finish_impl(h, s, row0, nb, S, rest, opts, h->stream, ea, eb, segmented ? (int64_t)(row0 / B) : -1);
      if (rc) return rc;
      batches++;
    }
  } else {
    return fail(h, FMX_E_ARG, "unknown SGD mode %d", opts->mode);
  }
  if (h->lag.active) {     // the last recurrence must finish inside the timed region; then w0 returns to h->w0
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->lag.ev_scan[(h->lag.step - 1) % LagState::RING], 0));
  }
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  rc = lag_flush(h);
  if (rc) return rc;
  rc = scan_error_check(h);
  if (rc) return rc;
  if (h->small_one_used) {                                      // one launch per small batch: did every owner / the recurrence see its examples?
    h->small_one_used = false;
    uint32_t e = 0;
    HIPCHK(h, hipMemcpy(&e, h->handoff_err, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (e & RUN_ERR_EXCHANGE) {
      const uint32_t rest_bits = e & ~RUN_ERR_EXCHANGE;
      (void)hipMemcpy(h->handoff_err, &rest_bits, sizeof(uint32_t), hipMemcpyHostToDevice);
      h->small_one = false;
      h->run_status |= FMX_STAT_HANDOFF_TIMEOUT;
      if (stats) stats->status = bi.status | h->run_status;
      return fail(h, FMX_E_HIP, "a one-launch batch never saw all of its examples (the device is shared or partitioned): the features concerned took no "
                                "step (the parameters are valid numbers, the epoch is not the batch rule's) -- reload the parameters; the handle takes "
                                "two launches per batch from now on");
    }
  }
  if (h->run_one_used) {                                        // one-launch runs of FMX_SGD_SEQUENTIAL: did every workgroup see its run arrive?
    h->run_one_used = false;
    uint32_t e = 0;
    HIPCHK(h, hipMemcpy(&e, h->handoff_err, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (e & RUN_ERR_EXCHANGE) {
      const uint32_t rest_bits = e & ~RUN_ERR_EXCHANGE;
      (void)hipMemcpy(h->handoff_err, &rest_bits, sizeof(uint32_t), hipMemcpyHostToDevice);
      h->run_one = false;
      h->run_status |= FMX_STAT_HANDOFF_TIMEOUT;
      if (stats) stats->status = bi.status | h->run_status;
      return fail(h, FMX_E_HIP, "a conflict-free run never saw all of its workgroups (the device is shared or partitioned): the rows concerned took no "
                                "step (the parameters are valid numbers, the epoch is not the reference's) -- reload the parameters; the handle takes "
                                "two launches per run from now on");
    }
  }
  if (h->handoff_err_host & 3u) {
    // a hand-off wait ran into its bound (the probe said the streams run side by side, and then they did not: the device is shared with
    // work that starved one of them).  Nothing was computed from a bias that was not there: the examples concerned took no step, a
    // recurrence that never saw its batch handed the bias on unchanged (fmx_kernels.h) -- every parameter is a valid number, but the
    // epoch is not the rule's.  The handle orders its streams with events from now on.
    const uint32_t e = h->handoff_err_host;
    h->handoff_err_host = 0;
    uint32_t dev = 0;
    if (hipMemcpy(&dev, h->handoff_err, sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess) {
      dev &= ~3u;                                               // (only the hand-off's own bits: bit 4 belongs to scan_error_check)
      (void)hipMemcpy(h->handoff_err, &dev, sizeof(uint32_t), hipMemcpyHostToDevice);
    }
    h->handoff = false;
    h->run_status |= FMX_STAT_HANDOFF_TIMEOUT;
    if (stats) stats->status = bi.status | h->run_status;
    return fail(h, FMX_E_HIP, "the bias hand-off between the launch stream and the recurrence timed out (flags %u): the examples concerned took no "
                              "step (the parameters are valid numbers, the epoch is not the batch rule's); the handle orders its streams with "
                              "events from now on", e);
  }
  if (kept_wside) s.wside_version = h->w_version;            // the epoch is complete: the stream holds this w
  if (stats) {
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    stats->rows = s.n_rows;
    stats->batches = batches;
    stats->device_seconds = ms * 1e-3;
    if (opts->mode == FMX_SGD_MINIBATCH && s.seg_B) stats->max_feature_count = s.max_seg_count;
    stats->deferred_features = deferred;
    if (opts->mode == FMX_SGD_MINIBATCH || opts->mode == FMX_SGD_HOGWILD) {     // (HOGWILD: batch_used = the rows in flight)
      stats->batch_used = bi.batch; stats->collision_mass = bi.collision_mass; stats->batch_gain = bi.batch_gain; stats->status = bi.status;
    }
    stats->status |= h->run_status;
    stats->setup_seconds = h->setup_acc;
    if (opts->mode != FMX_SGD_SEQUENTIAL) stats->w0_chunk_used = opts->w0_chunk ? opts->w0_chunk : default_w0_chunk(h->cfg);
    if (opts->mode == FMX_SGD_MINIBATCH && timed && opts->apply != FMX_APPLY_FUSED) {
      double tot = 0;
      for (size_t i = 0; i + 1 < ev_used; i += 2) {
        float m2 = 0;
        HIPCHK(h, hipEventElapsedTime(&m2, h->ev_pool[i], h->ev_pool[i + 1]));
        tot += m2 * 1e-3;
      }
      stats->main_kernel_seconds = tot;
      stats->main_kernel_launches = main_launches;
    } else {
      stats->main_kernel_seconds = stats->device_seconds;
      stats->main_kernel_launches = main_launches;
    }
  }
  return FMX_OK;
}

// ---------------------------------------------------------------------------------------------
// SGDA
// ---------------------------------------------------------------------------------------------
extern "C++" void sgda_free(fmx_handle h) {
  if (h->sgda.gw) fmx_dev_free(h->sgda.gw);
  if (h->sgda.gv) fmx_dev_free(h->sgda.gv);
  if (h->sgda.reg) fmx_dev_free(h->sgda.reg);
  if (h->sgda.dreg) fmx_dev_free(h->sgda.dreg);
  h->sgda = SgdaState();
}

int fmx_sgda_end(fmx_handle h) {
  if (!h) return FMX_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  sgda_free(h);
  return FMX_OK;
}

int fmx_sgda_begin(fmx_handle h) {
  touch_w(h);
  if (!h) return FMX_E_ARG;
  if (h->cfg.shard_world > 1) return fail(h, FMX_E_UNSUPPORTED, "SGDA on a feature shard is not implemented");
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  sgda_free(h);
  const size_t nv = h->n_local * (size_t)h->tb.rs, nreg = (size_t)h->num_groups * (1 + (size_t)h->KP);   // [G][1 + KP]
  HIPCHK(h, fmx_dev_alloc(&h->sgda.gw, h->n_local * sizeof(float)));
  HIPCHK(h, fmx_dev_alloc(&h->sgda.gv, nv * sizeof(float)));
  HIPCHK(h, fmx_dev_alloc(&h->sgda.reg, nreg * sizeof(double)));
  HIPCHK(h, hipMemsetAsync(h->sgda.gw, 0, h->n_local * sizeof(float), h->stream));
  HIPCHK(h, hipMemsetAsync(h->sgda.gv, 0, nv * sizeof(float), h->stream));
  HIPCHK(h, hipMemsetAsync(h->sgda.reg, 0, nreg * sizeof(double), h->stream));
  // fm->w.init(0) (:256): the linear weights restart from zero
  if (h->tb.ws == 1) HIPCHK(h, hipMemsetAsync(h->tb.w, 0, h->n_local * sizeof(float), h->stream));
  else HIPCHK(h, hipMemset2DAsync(h->tb.w, (size_t)h->tb.ws * sizeof(float), 0, sizeof(float), h->n_local, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return FMX_OK;
}

int fmx_sgda_get_reg(fmx_handle h, double* reg) {
  if (!h || !reg) return FMX_E_ARG;
  if (!h->sgda.reg) return fail(h, FMX_E_STATE, "fmx_sgda_get_reg before fmx_sgda_begin");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t G = h->num_groups, k = (size_t)h->cfg.num_factor, KP = (size_t)h->KP;
  std::vector<double> dev(G * (1 + KP));
  HIPCHK(h, hipMemcpyAsync(dev.data(), h->sgda.reg, dev.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (size_t g = 0; g < G; g++) memcpy(reg + g * (1 + k), dev.data() + g * (1 + KP), (1 + k) * sizeof(double));
  return FMX_OK;
}

int fmx_sgda_epoch(fmx_handle h, int train_slot, int validation_slot, int do_lambda_steps, fmx_epoch_stats* stats) {
  touch_w(h);
  int rc = check_slot(h, train_slot, true);
  if (rc) return rc;
  rc = check_slot(h, validation_slot, true);
  if (rc) return rc;
  if (!h->sgda.reg) return fail(h, FMX_E_STATE, "fmx_sgda_epoch before fmx_sgda_begin");
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  HIPCHK(h, hipSetDevice(h->device));
  const Slot& s = h->slots[train_slot];
  const Slot& v = h->slots[validation_slot];
  if (!s.blocks.empty() || !v.blocks.empty()) return fail(h, FMX_E_UNSUPPORTED, "relations are not supported with SGD");   // fm_learn_sgd.h:61-63
  const Hyper hy = make_hyper(h->cfg);
  if (stats) memset(stats, 0, sizeof(*stats));
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  if (h->num_groups <= 1) {
    KP_SWITCH(h->KP, hipLaunchKernelGGL((k_sgda<KP>), dim3(1), dim3(64), 0, h->stream, s.ent, s.row_ptr, s.target, s.n_rows,
                                          v.ent, v.row_ptr, v.target, v.n_rows, h->tb, h->sgda.gw, h->sgda.gv, hy, h->w0,
                                          h->sgda.reg, do_lambda_steps));
  } else {
    // LDS tables of k_sgda_groups: regw[G] regv[G][KP] lwg[G] sfg[G][KP] sdfg[G][KP] (doubles) + stamp[G] (u32)
    const size_t G = h->num_groups, lds = (2 * G + 3 * G * (size_t)h->KP) * sizeof(double) + G * sizeof(uint32_t);
    const size_t lds_max = h->prop.sharedMemPerBlock ? h->prop.sharedMemPerBlock : 64 * 1024;
    if (lds > lds_max)
      return fail(h, FMX_E_UNSUPPORTED, "SGDA: %zu attribute groups x %d factors need %zu bytes of LDS (limit %zu)", G, h->KP, lds, lds_max);
    KP_SWITCH(h->KP, hipLaunchKernelGGL((k_sgda_groups<KP>), dim3(1), dim3(64), lds, h->stream, s.ent, s.row_ptr, s.target, s.n_rows,
                                          v.ent, v.row_ptr, v.target, v.n_rows, h->tb, h->sgda.gw, h->sgda.gv, hy, h->w0,
                                          h->sgda.reg, do_lambda_steps, h->grp, (uint32_t)G));
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (stats) {
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    stats->rows = s.n_rows; stats->batches = s.n_rows; stats->device_seconds = ms * 1e-3;
    stats->main_kernel_seconds = stats->device_seconds; stats->main_kernel_launches = 1;
  }
  return FMX_OK;
}

// the batch form of the learner (oracle fmo_sgda_epoch_minibatch): per batch of `batch` train rows the theta step as a
// minibatch rule, then -- do_lambda_steps -- the lambda steps of the next `batch` validation rows, summed and applied once
int fmx_sgda_epoch_minibatch(fmx_handle h, int train_slot, int validation_slot, int do_lambda_steps, uint32_t batch,
                             uint32_t w0_chunk, fmx_epoch_stats* stats) {
  touch_w(h);
  int rc = check_slot(h, train_slot, true);
  if (rc) return rc;
  rc = check_slot(h, validation_slot, true);
  if (rc) return rc;
  if (!h->sgda.reg) return fail(h, FMX_E_STATE, "fmx_sgda_epoch_minibatch before fmx_sgda_begin");
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  HIPCHK(h, hipSetDevice(h->device));
  Slot& s = h->slots[train_slot];
  const Slot& v = h->slots[validation_slot];
  if (stats) memset(stats, 0, sizeof(*stats));
  if (s.n_rows == 0) return FMX_OK;
  if (!s.blocks.empty() || !v.blocks.empty()) return fail(h, FMX_E_UNSUPPORTED, "relations are not supported with SGD");   // fm_learn_sgd.h:61-63
  Hyper hy = make_hyper(h->cfg);
  hy.sgda = 1; hy.reg0 = 0.f; hy.reg0_d = 0.0;                  // reg_0 = 0 (:100, :149)
  // batch 0: the default cut to the rows' stability bound; this learner's regression multiplier is 2 (p - y): twice the curvature
  fmx_batch_info bi;
  rc = ensure_coll_mass(h, s);
  if (rc) return rc;
  resolve_batch(h->cfg, s.coll_mass, batch, FMX_DEFAULT_BATCH, h->cfg.task == FMX_TASK_REGRESSION ? 2.0 : 1.0, &bi);
  const uint32_t B = bi.batch;
  // micro-chunk of the bias: this learner's regression multiplier is 2 (p - y), i.e. twice the curvature of plain SGD
  uint32_t chunk = w0_chunk ? w0_chunk : std::max<uint32_t>(1u, default_w0_chunk(h->cfg) / (h->cfg.task == FMX_TASK_REGRESSION ? 2u : 1u));
  rc = ensure_segments(h, s, B);
  if (rc) return rc;
  const uint32_t Bc = std::min<uint32_t>(B, s.n_rows);
  rc = ensure_scratch(h, Bc, 0);
  if (rc) return rc;
  const size_t G = h->num_groups, cells = G * (1 + (size_t)h->KP);
  const bool grouped = G > 1;                                    // one group: the row's tables and the workgroup's sums live in registers
  const size_t lds = grouped ? (G + 2 * G * (size_t)h->KP + cells) * sizeof(double) : 0;
  if (do_lambda_steps && lds > 64 * 1024)
    return fail(h, FMX_E_UNSUPPORTED, "SGDA batch form: %zu attribute groups x %d factors need %zu bytes of LDS per validation row (limit 65536)", G, h->KP, lds);
  // workgroups of the lambda step (one partial [G][1 + KP] each, summed in a fixed order by k_sgda_reg_update)
  const uint32_t n_wg = (uint32_t)std::max<size_t>(64, std::min<size_t>(4096, ((size_t)8 << 20) / cells));
  if (do_lambda_steps && h->sgda.dreg_cap < (size_t)n_wg * cells) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->sgda.dreg) { fmx_dev_free(h->sgda.dreg); h->sgda.dreg = nullptr; h->sgda.dreg_cap = 0; }
    HIPCHK(h, fmx_dev_alloc(&h->sgda.dreg, (size_t)n_wg * cells * sizeof(double)));
    h->sgda.dreg_cap = (size_t)n_wg * cells;
  }
  hipStream_t st = h->stream;
  HIPCHK(h, hipEventRecord(h->ev0, st));
  uint64_t vpos = 0, batches = 0;                                // validation->data->begin() at the start of the epoch (:266)
  for (uint64_t row0 = 0; row0 < s.n_rows; row0 += B) {
    const uint32_t nb = (uint32_t)std::min<uint64_t>(B, s.n_rows - row0);
    float* S = h->partial;
    float* rest = S + (size_t)nb * h->KP;
    KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_rowsums<KP, true, true>), nb, st, s.ent, s.row_ptr, row0, nb, h->tb, h->cfg.k1, S, rest, (const float*)nullptr));
    rc = launch_scan(h, rest, s.target + row0, nb, chunk, hy, h->mult, st);
    if (rc) return rc;
    const size_t bi = (size_t)(row0 / B);
    const uint32_t s0 = s.batch_seg[bi], s1 = s.batch_seg[bi + 1];
    const uint64_t base = s.batch_base[bi];
    if (s1 > s0) {
      SegWork sw{s.t_ent + base, s.seg_feat + s0, s.seg_rel + s0, nullptr, s1 - s0, s1 - s0, (uint32_t)(s.batch_base[bi + 1] - base), S, h->mult};
      KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_sgda_apply_seg<KP, 8>), ((uint64_t)(s1 - s0) + 63) / 64, st, sw, h->tb, hy,
                                         (const double*)h->sgda.reg, (const uint32_t*)h->grp, h->sgda.gw, h->sgda.gv));
    }
    if (do_lambda_steps && v.n_rows) {
      const uint32_t wg = std::min<uint32_t>(nb, n_wg);
      if (grouped) { KP_SWITCH(h->KP, hipLaunchKernelGGL((k_sgda_lambda<KP, true>), dim3(wg), dim3(64), lds, st, v.ent, v.row_ptr, v.target,
                                            v.n_rows, (uint32_t)vpos, nb, h->tb, (const float*)h->sgda.gw, (const float*)h->sgda.gv, hy,
                                            (const double*)h->w0, (const double*)h->sgda.reg, h->sgda.dreg, (const uint32_t*)h->grp, (uint32_t)G)); }
      else         { KP_SWITCH(h->KP, hipLaunchKernelGGL((k_sgda_lambda<KP, false>), dim3(wg), dim3(64), 0, st, v.ent, v.row_ptr, v.target,
                                            v.n_rows, (uint32_t)vpos, nb, h->tb, (const float*)h->sgda.gw, (const float*)h->sgda.gv, hy,
                                            (const double*)h->w0, (const double*)h->sgda.reg, h->sgda.dreg, (const uint32_t*)h->grp, (uint32_t)G)); }
      hipLaunchKernelGGL(k_sgda_reg_update, dim3((unsigned)std::min<size_t>(cells, 1024)), dim3(256), 0, st, h->sgda.reg, (const double*)h->sgda.dreg,
                         wg, (uint32_t)cells, h->KP, h->cfg.k1);
      vpos = (vpos + nb) % v.n_rows;
    }
    HIPCHK(h, hipGetLastError());
    batches++;
  }
  HIPCHK(h, hipEventRecord(h->ev1, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (stats) {
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    stats->rows = s.n_rows; stats->batches = batches; stats->device_seconds = ms * 1e-3;
    stats->main_kernel_seconds = stats->device_seconds; stats->main_kernel_launches = batches;
    stats->max_feature_count = s.max_seg_count;
    stats->batch_used = bi.batch; stats->collision_mass = bi.collision_mass; stats->batch_gain = bi.batch_gain; stats->status = bi.status;
  }
  return FMX_OK;
}

}  // extern "C"

// fmx_als.hip -- C-ABI (include/fmx.h): the ALS / MCMC learner (level-scheduled coordinate sweeps, fmx_als_kernels.h).
#define FMX_GRID_OVER_DEFAULT 2                    // = FMX_GRID_OVER_ALS (fmx_internal.h): the column kernels keep grid-stride workgroups
#include "fmx_internal.h"

extern "C" {

// ---------------------------------------------------------------------------------------------
// ALS / MCMC
// ---------------------------------------------------------------------------------------------
extern "C++" void als_free(fmx_handle h) {
  AlsState& a = h->als;
  if (a.e) fmx_dev_free(a.e);
  if (a.q) fmx_dev_free(a.q);
  if (a.seen) fmx_dev_free(a.seen);
  if (a.level_list) fmx_dev_free(a.level_list);
  if (a.ldesc) fmx_dev_free(a.ldesc);
  if (a.prior) fmx_dev_free(a.prior);
  if (a.vt) fmx_dev_free(a.vt);
  if (a.delta) fmx_dev_free(a.delta);
  if (a.epart) fmx_dev_free(a.epart);
  if (a.r_row) fmx_dev_free(a.r_row);
  if (a.r_pos) fmx_dev_free(a.r_pos);
  if (a.r_x) fmx_dev_free(a.r_x);
  if (a.t_row) fmx_dev_free(a.t_row);
  if (a.dth) fmx_dev_free(a.dth);
  for (AlsBlock& b : a.blk) {
    if (b.level_list) fmx_dev_free(b.level_list);
    if (b.cache) fmx_dev_free(b.cache);
    if (b.qb_all) fmx_dev_free(b.qb_all);
    if (b.cpart) fmx_dev_free(b.cpart);
  }
  a = AlsState();
}

int fmx_als_end(fmx_handle h) {
  touch_w(h);
  if (!h) return FMX_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  als_free(h);
  return FMX_OK;
}

static int als_eterms(fmx_handle h, const Slot& s, EQ* e, double* q, double* e_part = nullptr) {
  KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_als_eterms<KP>), ((uint64_t)s.n_rows + EtermsRows<KP>::R - 1) / EtermsRows<KP>::R, h->stream, s.ent, s.row_ptr, s.n_rows, h->tb,
                                     h->cfg.k0, h->cfg.k1, h->w0, e, q, e_part));
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

// the dependency levels of the columns of a (transposed) data set: level(j) = 1 + max level of the earlier columns sharing a
// row with j (host, O(nnz)); fills level_ptr and uploads the level-ordered column list; marks seen[id_offset + feature]
static int build_levels(fmx_handle h, const Slot& s, std::vector<uint32_t>& level_ptr, uint32_t** d_level_list, std::vector<uint8_t>* seen,
                        uint32_t id_offset, std::vector<uint32_t>* seg_level = nullptr, std::vector<uint32_t>* seg_pos = nullptr,
                        std::vector<uint32_t>* lev_ent = nullptr) {
  const uint32_t N = s.n_rows, nseg = s.nseg;
  std::vector<uint32_t> seg_feat(nseg), seg_rel((size_t)nseg + 1), lvl(nseg), rowlevel(std::max<uint32_t>(N, 1), 0);
  std::vector<TEntry> tent((size_t)s.nnz);
  if (nseg) {
    HIPCHK(h, hipMemcpy(seg_feat.data(), s.seg_feat, (size_t)nseg * 4, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(seg_rel.data(), s.seg_rel, (size_t)nseg * 4, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(tent.data(), s.t_ent, (size_t)s.nnz * sizeof(TEntry), hipMemcpyDeviceToHost));
  }
  seg_rel[nseg] = (uint32_t)s.nnz;
  uint32_t n_levels = 0;
  for (uint32_t sg = 0; sg < nseg; sg++) {
    uint32_t l = 0;
    for (uint32_t i = seg_rel[sg]; i < seg_rel[sg + 1]; i++) l = std::max(l, rowlevel[tent[i].e]);
    l += 1;
    for (uint32_t i = seg_rel[sg]; i < seg_rel[sg + 1]; i++) rowlevel[tent[i].e] = l;
    lvl[sg] = l - 1;
    n_levels = std::max(n_levels, l);
  }
  level_ptr.assign((size_t)n_levels + 1, 0);
  for (uint32_t sg = 0; sg < nseg; sg++) level_ptr[lvl[sg] + 1]++;
  for (uint32_t l = 0; l < n_levels; l++) level_ptr[l + 1] += level_ptr[l];
  std::vector<uint32_t> list(std::max<uint32_t>(nseg, 1)), fill(level_ptr.begin(), level_ptr.end());
  if (seg_pos) seg_pos->assign(std::max<uint32_t>(nseg, 1), 0);
  if (lev_ent) lev_ent->assign((size_t)n_levels + 1, 0);
  for (uint32_t sg = 0; sg < nseg; sg++) {
    if (seg_pos) (*seg_pos)[sg] = fill[lvl[sg]] - level_ptr[lvl[sg]];      // position inside its level's list
    if (lev_ent) (*lev_ent)[lvl[sg] + 1] += seg_rel[sg + 1] - seg_rel[sg];
    list[fill[lvl[sg]]++] = sg;
  }
  if (lev_ent) for (uint32_t l = 0; l < n_levels; l++) (*lev_ent)[l + 1] += (*lev_ent)[l];
  if (seg_level) seg_level->swap(lvl);
  if (seen) for (uint32_t sg = 0; sg < nseg; sg++) (*seen)[(size_t)id_offset + seg_feat[sg]] = 1;
  HIPCHK(h, fmx_dev_alloc(d_level_list, list.size() * 4));
  HIPCHK(h, hipMemcpy(*d_level_list, list.data(), list.size() * 4, hipMemcpyHostToDevice));
  return FMX_OK;
}

// y-hat (and q_f for the coming sweep) of the session's rows.  With kept blocks the main rows and every block's rows are
// evaluated separately -- a block row once, however many main rows map to it -- and combined through the mappings;
// the squares are of the COMPLETE factor sums (k_als_set_e).
static int als_repredict(fmx_handle h, const Slot& s, AlsState& a) {
  if (s.blocks.empty()) return als_eterms(h, s, a.e, a.q);
  const uint32_t N = s.n_rows;
  const dim3 g1(std::min<uint32_t>((N + 255) / 256, 2048)), b1(256);
  int rc = als_eterms(h, s, a.e, a.q, a.epart);
  if (rc) return rc;
  for (size_t r = 0; r < s.blocks.size(); r++) {
    const BlockRows& br = *s.blocks[r];
    AlsBlock& ab = a.blk[r];
    const uint32_t B = br.rows.n_rows;
    Tab tb = h->tb;                                          // the block's attribute 0 is global attribute attr_offset
    tb.V += (size_t)br.attr_offset * tb.rs; tb.w += (size_t)br.attr_offset * tb.ws;
    KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_als_eterms<KP>), ((uint64_t)B + EtermsRows<KP>::R - 1) / EtermsRows<KP>::R, h->stream, br.rows.ent, br.rows.row_ptr, B, tb,
                                       0, h->cfg.k1, (const double*)h->w0, (EQ*)nullptr, ab.qb_all, ab.cpart));
    hipLaunchKernelGGL(k_rel_combine, g1, b1, 0, h->stream, br.map, N, B, ab.cpart, ab.qb_all, h->cfg.num_factor, a.epart, a.q);
  }
  hipLaunchKernelGGL(k_als_set_e, g1, b1, 0, h->stream, a.e, a.epart, a.q, h->cfg.num_factor, N, h->cfg.k0, h->w0);
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

// the split step's row-ordered copy of X^T (AlsState::r_*): device radix sort of the entries by (level of the feature, row)
static int als_build_rows(fmx_handle h, const Slot& s, AlsState& a, const std::vector<uint32_t>& seg_level, const std::vector<uint32_t>& seg_pos) {
  const uint32_t nnz = (uint32_t)s.nnz, nseg = s.nseg;
  hipStream_t st = h->stream;
  uint64_t *ka = nullptr, *kb = nullptr; uint32_t *va = nullptr, *vb = nullptr, *d_lvl = nullptr, *d_pos = nullptr; void* tmp = nullptr;
  int rc = FMX_OK;
#define ROW_CHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    rc = fail(h, FMX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); goto done; } } while (0)
  {
    uint32_t big = 1;
    for (size_t l = 0; l + 1 < a.level_ptr.size(); l++) big = std::max(big, a.level_ptr[l + 1] - a.level_ptr[l]);
    int bits_level = 1; while ((1ull << bits_level) < a.level_ptr.size()) bits_level++;
    size_t tmp_bytes = 0;
    const dim3 gr(std::min<uint32_t>((nnz + 255) / 256, 8192)), bl(256);
    ROW_CHK(fmx_dev_alloc(&ka, (size_t)nnz * 8)); ROW_CHK(fmx_dev_alloc(&kb, (size_t)nnz * 8));
    ROW_CHK(fmx_dev_alloc(&va, (size_t)nnz * 4)); ROW_CHK(fmx_dev_alloc(&vb, (size_t)nnz * 4));
    ROW_CHK(fmx_dev_alloc(&d_lvl, (size_t)nseg * 4)); ROW_CHK(fmx_dev_alloc(&d_pos, (size_t)nseg * 4));
    ROW_CHK(fmx_dev_alloc(&a.r_row, (size_t)nnz * 4)); ROW_CHK(fmx_dev_alloc(&a.r_pos, (size_t)nnz * 4)); ROW_CHK(fmx_dev_alloc(&a.r_x, (size_t)nnz * 4));
    ROW_CHK(fmx_dev_alloc(&a.dth, (size_t)big * sizeof(float2)));
    ROW_CHK(hipMemcpyAsync(d_lvl, seg_level.data(), (size_t)nseg * 4, hipMemcpyHostToDevice, st));
    ROW_CHK(hipMemcpyAsync(d_pos, seg_pos.data(), (size_t)nseg * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_als_rowkeys, gr, bl, 0, st, s.t_ent, s.seg_rel, nseg, nnz, d_lvl, ka, va);
    ROW_CHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, ka, kb, va, vb, (int)nnz, 0, 32 + bits_level, st));
    ROW_CHK(fmx_dev_alloc(&tmp, tmp_bytes));
    ROW_CHK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, ka, kb, va, vb, (int)nnz, 0, 32 + bits_level, st));
    hipLaunchKernelGGL(k_als_rowfill, gr, bl, 0, st, s.t_ent, s.seg_rel, nseg, nnz, d_pos, vb, a.r_row, a.r_pos, a.r_x);
    ROW_CHK(hipGetLastError());
    {  // which levels hold every row exactly once, in order (one-hot fields)
      const size_t nl = a.lev_ent.size() - 1;
      a.lev_dense.assign(nl, 0);
      std::vector<uint32_t> flags(nl, 1u);
      ROW_CHK(hipMemsetAsync(d_lvl, 0, std::min<size_t>(nl, nseg) * 4, st));          // (d_lvl is free after the sort keys were made)
      for (size_t l = 0; l < nl && l < nseg; l++)
        if (a.lev_ent[l + 1] - a.lev_ent[l] == s.n_rows)
          hipLaunchKernelGGL(k_als_rows_check_dense, dim3(std::min<uint32_t>((s.n_rows + 255) / 256, 4096)), bl, 0, st, a.r_row + a.lev_ent[l],
                             a.r_x + a.lev_ent[l], s.n_rows, d_lvl + l);
      ROW_CHK(hipMemcpyAsync(flags.data(), d_lvl, std::min<size_t>(nl, nseg) * 4, hipMemcpyDeviceToHost, st));
      ROW_CHK(hipStreamSynchronize(st));
      bool any_unit = false;
      for (size_t l = 0; l < nl && l < nseg; l++) {
        const bool dense = a.lev_ent[l + 1] - a.lev_ent[l] == s.n_rows && (flags[l] & 1u) == 0;
        a.lev_dense[l] = dense ? ((flags[l] & 2u) ? 1 : 2) : 0;
        any_unit = any_unit || a.lev_dense[l] == 2;
      }
      if (any_unit) {                                                // the row-only stream of X^T for the levels of unit values
        ROW_CHK(fmx_dev_alloc(&a.t_row, (size_t)nnz * 4));
        hipLaunchKernelGGL(k_als_trow, gr, bl, 0, st, s.t_ent, (uint32_t)nnz, a.t_row);
        ROW_CHK(hipGetLastError());
        ROW_CHK(hipStreamSynchronize(st));
      }
    }
  }
done:
#undef ROW_CHK
  for (void* p : {(void*)ka, (void*)kb, (void*)va, (void*)vb, (void*)d_lvl, (void*)d_pos, tmp}) if (p) fmx_dev_free(p);
  return rc;
}

static int als_begin_impl(fmx_handle h, int train_slot);
int fmx_als_begin(fmx_handle h, int train_slot) {
  const int rc = als_begin_impl(h, train_slot);
  if (rc != FMX_OK && h && h->als.slot == train_slot) als_free(h);      // a failed set-up leaves no half-built session behind
  return rc;
}
static int als_begin_impl(fmx_handle h, int train_slot) {
  touch_w(h);
  int rc = check_slot(h, train_slot, true);
  if (rc) return rc;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  if (h->cfg.shard_world > 1) return fail(h, FMX_E_STATE, "ALS / MCMC on a feature shard: use fmx_group_als_begin (the dependency levels are global)");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  als_free(h);
  Slot& s = h->slots[train_slot];
  if (s.n_rows == 0) return fail(h, FMX_E_ARG, "fmx_als_begin: empty training set");
  rc = ensure_segments(h, s, s.n_rows);          // one "batch" = the whole data set: X^T with columns in id order
  if (rc) return rc;
  AlsState& a = h->als;
  a.slot = train_slot;
  const uint32_t N = s.n_rows;
  // ---- dependency levels of the main features, then -- kept `-relation` blocks -- of every block's attributes over
  //      the block's own rows (two block attributes conflict iff they share a block row)
  std::vector<uint8_t> seen((size_t)h->n_local, 0);
  {
    // levels with many entries take the split step (column sums + draw, then the {e, q} update as a row-ordered stream):
    // a fused draw pays two random touches of the {e, q} cache per entry, the split one.  fmx_config::als_split_min: threshold in
    // entries per level; small levels are launch-bound and stay fused.
    a.split_min = h->cfg.als_split_min == 0 ? 65536u : (h->cfg.als_split_min == 0xFFFFFFFFu ? 0u : h->cfg.als_split_min);
    std::vector<uint32_t> seg_level, seg_pos;
    rc = build_levels(h, s, a.level_ptr, &a.level_list, &seen, 0, &seg_level, &seg_pos, &a.lev_ent);
    if (rc) return rc;
    bool any = false;
    for (size_t l = 0; a.split_min && l + 1 < a.lev_ent.size(); l++) any = any || (a.lev_ent[l + 1] - a.lev_ent[l] >= a.split_min);
    if (any && s.nnz > 0 && s.nnz < (1ull << 32)) { rc = als_build_rows(h, s, a, seg_level, seg_pos); if (rc) return rc; }
    else a.split_min = 0;
  }
  a.blk.resize(s.blocks.size());
  for (size_t r = 0; r < s.blocks.size(); r++) {
    BlockRows& br = *s.blocks[r];
    AlsBlock& ab = a.blk[r];
    const uint32_t B = br.rows.n_rows;
    rc = ensure_segments(h, br.rows, std::max<uint32_t>(B, 1));            // X^T of the block (device radix sort)
    if (rc) return rc;
    rc = build_levels(h, br.rows, ab.level_ptr, &ab.level_list, &seen, br.attr_offset);
    if (rc) return rc;
    HIPCHK(h, fmx_dev_alloc(&ab.cache, (size_t)7 * std::max<uint32_t>(B, 1) * sizeof(double)));
    HIPCHK(h, hipMemsetAsync(ab.cache, 0, (size_t)7 * std::max<uint32_t>(B, 1) * sizeof(double), h->stream));
    HIPCHK(h, fmx_dev_alloc(&ab.qb_all, (size_t)h->KP * std::max<uint32_t>(B, 1) * sizeof(double)));
    HIPCHK(h, fmx_dev_alloc(&ab.cpart, (size_t)std::max<uint32_t>(B, 1) * sizeof(double)));
    const double avg_col = br.rows.nseg ? (double)br.rows.nnz / (double)br.rows.nseg : 0.0;
    ab.lanes = avg_col <= 5.0 ? 4 : (avg_col <= 12.0 ? 8 : (avg_col <= 40.0 ? 16 : 64));
  }
  const uint32_t nseg = s.nseg;
  HIPCHK(h, fmx_dev_alloc(&a.seen, seen.size()));
  HIPCHK(h, hipMemcpy(a.seen, seen.data(), seen.size(), hipMemcpyHostToDevice));
  if (!s.blocks.empty()) HIPCHK(h, fmx_dev_alloc(&a.epart, (size_t)N * sizeof(double)));
  HIPCHK(h, fmx_dev_alloc(&a.e, (size_t)N * sizeof(EQ)));
  HIPCHK(h, fmx_dev_alloc(&a.q, (size_t)N * (size_t)h->KP * sizeof(double)));
  if (h->cfg.num_factor > 0 && nseg > 0) {      // (the env switch is the A/B knob of the profile)
    a.vt_stride = ((size_t)nseg + 63) & ~(size_t)63;
    HIPCHK(h, fmx_dev_alloc(&a.vt, (size_t)h->cfg.num_factor * a.vt_stride * sizeof(float)));
  }
  // ---- first prediction and e -= target (fm_learn_mcmc_simultaneous.h:69-86)
  rc = als_repredict(h, s, a);
  if (rc) return rc;
  hipLaunchKernelGGL(k_als_sub_target, dim3(std::min<uint32_t>((N + 255) / 256, 2048)), dim3(256), 0, h->stream, a.e, s.target, N);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return FMX_OK;
}

int fmx_als_moments(fmx_handle h, double* out) {
  if (!h || !out) return FMX_E_ARG;
  AlsState& a = h->als;
  if (a.slot < 0) return fail(h, FMX_E_STATE, "fmx_als_moments before fmx_als_begin");
  HIPCHK(h, hipSetDevice(h->device));
  const Slot& s = h->slots[a.slot];
  const int k = h->cfg.num_factor;
  const uint32_t G = h->num_groups;
  const size_t cells = (size_t)(1 + h->KP) * G * 2;              // device block [1 + KP][G][2]
  double* d = nullptr;
  HIPCHK(h, fmx_dev_alloc(&d, (2 + cells) * sizeof(double)));
  hipStream_t st = h->stream;
  hipError_t er = hipMemsetAsync(d, 0, (2 + cells) * sizeof(double), st);
  const dim3 b1(256), ge(std::min<uint32_t>((s.n_rows + 255) / 256, 2048));
  hipLaunchKernelGGL(k_als_sum_e, ge, b1, 0, st, a.e, s.n_rows, d);            // d[0] = sum e, d[1] = sum e^2
  const bool grouped = G > 1;
  const int lds_ok = cells * sizeof(double) <= 64 * 1024;
  const size_t lds = (grouped && lds_ok) ? cells * sizeof(double) : 0;
  int rc = FMX_OK;
  do {
    KP_SWITCH(h->KP, {
      const uint64_t waves = (h->n_local + Map<KP>::EPI - 1) / Map<KP>::EPI;
      if (grouped) { auto kf = k_group_moments<KP, true>;
        hipLaunchKernelGGL(kf, dim3(resident_grid(h, (const void*)kf, waves)), b1, lds, st, h->tb, h->n_local, h->cfg.k1, h->grp, G, lds_ok, d + 2);
      } else { auto kf = k_group_moments<KP, false>;
        hipLaunchKernelGGL(kf, dim3(resident_grid(h, (const void*)kf, waves)), b1, 0, st, h->tb, h->n_local, h->cfg.k1, h->grp, G, 0, d + 2);
      }
    });
  } while (0);
  std::vector<double> host(2 + cells);
  if (er == hipSuccess) er = hipGetLastError();
  if (er == hipSuccess) er = hipMemcpyAsync(host.data(), d, (2 + cells) * sizeof(double), hipMemcpyDeviceToHost, st);
  if (er == hipSuccess) er = hipStreamSynchronize(st);
  fmx_dev_free(d);
  if (er != hipSuccess) return fail(h, FMX_E_HIP, "fmx_als_moments: %s", hipGetErrorString(er));
  out[0] = host[1]; out[1] = host[0];                                          // documented order: sum e^2 first
  memcpy(out + 2, host.data() + 2, (size_t)(1 + k) * G * 2 * sizeof(double)); // rows 0..k of [1 + KP][G][2]
  return rc;
}

// one iteration of _learn over a list of feature shards (one unsharded handle: the list has one member and nothing is
// exchanged).  Shards: the {e, q} cache is replicated; the draws of a (family, level) step record their changes
// (AlsState::delta), one all-reduce per step sums them and every shard applies the same sum -- the replicas stay identical
// bit for bit; the re-prediction all-reduces the partial y-hat and q_f of the shards ("the simple
// version").  The levels are GLOBAL (fmx_group_als_begin), so the sweep is the reference's Gauss-Seidel order whatever
// the number of shards.
static int als_sweep_shards(const std::vector<fmx_handle>& hs, fmx_group g, const fmx_als_opts* opts, fmx_als_stats* stats) {
  for (fmx_handle m : hs) touch_w(m);
  fmx_handle h = hs[0];
  const size_t P = hs.size();
  const bool sharded = g != nullptr && P > 1;
  for (fmx_handle x : hs) {
    if (x->als.slot < 0) return fail(h, FMX_E_STATE, "fmx_als_sweep before fmx_als_begin");
    if (opts->num_groups != 0 && opts->num_groups != x->num_groups)     // validated BEFORE the first device write of the sweep
      return fail(h, FMX_E_ARG, "fmx_als_sweep: opts->num_groups = %u but the handle has %u attribute groups", opts->num_groups, x->num_groups);
  }
  AlsState& a0 = h->als;
  const uint32_t N = h->slots[a0.slot].n_rows;
  const uint32_t n_levels = (uint32_t)a0.level_ptr.size() - 1;
  const dim3 g1(std::min<uint32_t>((N + 255) / 256, 2048)), b1(256);
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  double acc[4] = {0, 0, 0, 0};
  // sum e, sum e^2 (draw_w0's numerator; draw_alpha's statistic for the caller): the replicas are identical, shard 0 answers
  HIPCHK(h, hipMemsetAsync(h->acc, 0, 4 * sizeof(double), h->stream));
  hipLaunchKernelGGL(k_als_sum_e, g1, b1, 0, h->stream, a0.e, N, h->acc);
  HIPCHK(h, hipMemcpyAsync(acc, h->acc, sizeof(acc), hipMemcpyDeviceToHost, h->stream));
  double w0 = 0;
  HIPCHK(h, hipMemcpyAsync(&w0, h->w0, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (stats) stats->sum_e_sqr = acc[1];
  std::mt19937_64 rng(opts->seed * 0x9E3779B97F4A7C15ull + a0.iter + 1);
  std::normal_distribution<double> nd(0.0, 1.0);
  if (h->cfg.k0) {                                         // draw_w0, fm_learn_mcmc.h:643-683 (w0_mean_0 = 0)
    double mean = acc[0] - (double)N * w0;
    const double sigma_sqr = 1.0 / (h->cfg.reg0 + opts->alpha * (double)N);
    mean = -sigma_sqr * (opts->alpha * mean - 0.0 * h->cfg.reg0);
    double nw0 = opts->do_sample ? mean + std::sqrt(sigma_sqr) * nd(rng) : mean;
    if (!(std::isnan(nw0) || std::isinf(nw0))) {
      for (fmx_handle x : hs) {
        HIPCHK(x, hipSetDevice(x->device));
        HIPCHK(x, hipMemcpyAsync(x->w0, &nw0, sizeof(double), hipMemcpyHostToDevice, x->stream));
        hipLaunchKernelGGL(k_als_add_const, g1, b1, 0, x->stream, x->als.e, N, nw0 - w0);
        HIPCHK(x, hipStreamSynchronize(x->stream));        // nw0 lives on this stack frame
      }
    }
  }
  // ---- priors per coordinate family and attribute group: [1 + k][2][NG] = lambda[NG] then mu[NG]
  const uint32_t NG = h->num_groups;
  const int kf = h->cfg.num_factor;
  const bool tabs = opts->num_groups != 0;
  std::vector<int> lanes(P);
  for (size_t i = 0; i < P; i++) {
    fmx_handle x = hs[i];
    AlsState& a = x->als;
    HIPCHK(x, hipSetDevice(x->device));
    a.prior_host.resize((size_t)(1 + kf) * 2 * NG);
    for (uint32_t gg = 0; gg < NG; gg++) {
      a.prior_host[gg] = (tabs && opts->w_lambda_g) ? opts->w_lambda_g[gg] : opts->w_lambda;
      a.prior_host[NG + gg] = (tabs && opts->w_mu_g) ? opts->w_mu_g[gg] : opts->w_mu;
      for (int f = 0; f < kf; f++) {
        double* row = a.prior_host.data() + (size_t)(1 + f) * 2 * NG;
        row[gg] = (tabs && opts->v_lambda_gf) ? opts->v_lambda_gf[(size_t)gg * kf + f] : (opts->v_lambda_f ? opts->v_lambda_f[f] : opts->v_lambda);
        row[NG + gg] = (tabs && opts->v_mu_gf) ? opts->v_mu_gf[(size_t)gg * kf + f] : (opts->v_mu_f ? opts->v_mu_f[f] : opts->v_mu);
      }
    }
    if (!a.prior) HIPCHK(x, fmx_dev_alloc(&a.prior, a.prior_host.size() * sizeof(double)));
    HIPCHK(x, hipMemcpyAsync(a.prior, a.prior_host.data(), a.prior_host.size() * sizeof(double), hipMemcpyHostToDevice, x->stream));
    // lanes per column from the mean column length (one-hot data: a handful of rows per feature)
    const Slot& s = x->slots[a.slot];
    if (!a.ldesc && s.nseg) {                                // the level-ordered column records, once per session
      HIPCHK(x, fmx_dev_alloc(&a.ldesc, (size_t)s.nseg * sizeof(uint4)));
      hipLaunchKernelGGL(k_als_ldesc, dim3(std::min<uint32_t>((s.nseg + 255) / 256, 8192)), dim3(256), 0, x->stream, a.level_list, s.nseg,
                         s.seg_feat, s.seg_rel, s.nseg, (uint32_t)s.nnz, a.ldesc);
      HIPCHK(x, hipGetLastError());
    }
    const double avg_col = s.nseg ? (double)s.nnz / (double)s.nseg : 0.0;
    // (measured at 6.7 entries per column, inside one process: 4 lanes 161.4 ms per sweep, 8 lanes 164.4, 16 lanes 195)
    lanes[i] = avg_col <= 8.0 ? 4 : (avg_col <= 16.0 ? 8 : (avg_col <= 40.0 ? 16 : 64));
  }
#define FMX_ALS_DRAW(ISV, cnt, ...)                                                                          \
  do {                                                                                                        \
    if (G == 4)       FMX_LAUNCH_WAVES((k_als_draw<ISV, 4>), ((uint64_t)(cnt) + 15) / 16, st, __VA_ARGS__);    \
    else if (G == 8)  FMX_LAUNCH_WAVES((k_als_draw<ISV, 8>), ((uint64_t)(cnt) + 7) / 8, st, __VA_ARGS__);      \
    else if (G == 16) FMX_LAUNCH_WAVES((k_als_draw<ISV, 16>), ((uint64_t)(cnt) + 3) / 4, st, __VA_ARGS__);     \
    else              FMX_LAUNCH_WAVES((k_als_draw<ISV, 64>), (uint64_t)(cnt), st, __VA_ARGS__);               \
  } while (0)
  // one (family, level) step on every shard, then -- shards only -- the exchange of the recorded changes
  auto level_step = [&](uint32_t l, int f /* -1: linear weights */) -> int {
    bool any = false;
    for (size_t i = 0; i < P; i++) {
      fmx_handle h = hs[i];                                  // (the launch macros refer to `h`)
      AlsState& a = h->als;
      const Slot& s = h->slots[a.slot];
      const uint32_t cnt = a.level_ptr[l + 1] - a.level_ptr[l];
      if (!cnt) continue;
      any = true;
      HIPCHK(h, hipSetDevice(h->device));
      hipStream_t st = h->stream;
      const int G = lanes[i];
      const Shard sh = make_shard(h->cfg);
      EQ* delta = sharded ? a.delta : nullptr;
      const uint32_t n_ent = (!sharded && a.split_min && a.r_row) ? a.lev_ent[l + 1] - a.lev_ent[l] : 0u;
      float2* dth = (n_ent && n_ent >= a.split_min) ? a.dth : nullptr;     // split step for this level?
      const bool unit = dth && l < a.lev_dense.size() && a.lev_dense[l] == 2 && a.t_row;   // every value of the level is 1: 4-byte streams
      const uint32_t* unit_rows = unit ? a.t_row : nullptr;
      const float* lev_x = (dth && !unit) ? a.r_x + a.lev_ent[l] : nullptr;   // (no split step: the row-ordered lists do not exist)
      if (f < 0) {
        FMX_ALS_DRAW(false, cnt, s.t_ent, unit_rows, a.ldesc + a.level_ptr[l], cnt,
                     h->tb.w, h->tb.ws, 0, 0u, a.e, opts->alpha, a.prior, a.prior + NG, h->grp, opts->do_sample,
                     opts->seed, (uint64_t)(a.iter * 1024 + 1000), sh, delta, dth);
        if (dth && a.lev_dense[l]) hipLaunchKernelGGL((k_als_rows_dense<false>), dim3(std::min<uint32_t>((n_ent + 255) / 256, 16384)), dim3(256), 0, st,
                                    a.r_pos + a.lev_ent[l], lev_x, n_ent, dth, a.e);
        else if (dth) hipLaunchKernelGGL((k_als_rows<false>), dim3(std::min<uint32_t>((n_ent + 255) / 256, 16384)), dim3(256), 0, st,
                                    a.r_row + a.lev_ent[l], a.r_pos + a.lev_ent[l], a.r_x + a.lev_ent[l], n_ent, dth, a.e);
      } else {
        const double* v_lambda = a.prior + (size_t)(1 + f) * 2 * NG;
        const double* v_mu = v_lambda + NG;
        if (a.vt)
          FMX_ALS_DRAW(true, cnt, s.t_ent, unit_rows, a.ldesc + a.level_ptr[l], cnt,
                       a.vt + (size_t)f * a.vt_stride, 1u, 1, a.level_ptr[l], a.e, opts->alpha, v_lambda, v_mu, h->grp, opts->do_sample,
                       opts->seed, (uint64_t)(a.iter * 1024 + f), sh, delta, dth);
        else
          FMX_ALS_DRAW(true, cnt, s.t_ent, unit_rows, a.ldesc + a.level_ptr[l], cnt,
                       h->tb.V + f, h->tb.rs, 0, 0u, a.e, opts->alpha, v_lambda, v_mu, h->grp, opts->do_sample,
                       opts->seed, (uint64_t)(a.iter * 1024 + f), sh, delta, dth);
        if (dth && a.lev_dense[l]) hipLaunchKernelGGL((k_als_rows_dense<true>), dim3(std::min<uint32_t>((n_ent + 255) / 256, 16384)), dim3(256), 0, st,
                                    a.r_pos + a.lev_ent[l], lev_x, n_ent, dth, a.e);
        else if (dth) hipLaunchKernelGGL((k_als_rows<true>), dim3(std::min<uint32_t>((n_ent + 255) / 256, 16384)), dim3(256), 0, st,
                                    a.r_row + a.lev_ent[l], a.r_pos + a.lev_ent[l], a.r_x + a.lev_ent[l], n_ent, dth, a.e);
      }
      HIPCHK(h, hipGetLastError());
    }
    if (sharded && any) {
      std::vector<double*> bufs(P);
      for (size_t i = 0; i < P; i++) bufs[i] = reinterpret_cast<double*>(hs[i]->als.delta);
      int rc = group_allreduce_f64(g, bufs, (size_t)N * 2);
      if (rc) return rc;
      for (fmx_handle x : hs) {
        HIPCHK(x, hipSetDevice(x->device));
        hipLaunchKernelGGL(k_als_apply_delta, g1, b1, 0, x->stream, x->als.e, x->als.delta, N);
        HIPCHK(x, hipGetLastError());
      }
    }
    return FMX_OK;
  };
  // kept `-relation` blocks (one unsharded handle): after the main features of a family, every block in turn -- caches from
  // the main rows, the block's attributes level by level on the caches, the accumulated changes back (fm_learn_mcmc.h:478-509
  // for w, :603-633 for v_f)
  auto block_steps = [&](int f /* -1: linear weights */) -> int {
    fmx_handle h = hs[0];
    AlsState& a = h->als;
    const Slot& s = h->slots[a.slot];
    hipStream_t st = h->stream;
    for (size_t r = 0; r < s.blocks.size(); r++) {
      const BlockRows& br = *s.blocks[r];
      AlsBlock& ab = a.blk[r];
      const uint32_t B = br.rows.n_rows;
      if (!B) continue;
      const dim3 gb(std::min<uint32_t>((B + 31) / 32, 2048));
      if (f >= 0) {
        hipLaunchKernelGGL(k_rel_load_qb, dim3(std::min<uint32_t>((B + 255) / 256, 2048)), b1, 0, st, ab.cache, (const double*)(ab.qb_all + (size_t)f * B), B);
        hipLaunchKernelGGL((k_rel_aggregate<true, 8>), gb, b1, 0, st, br.brow_ptr, br.brow_list, B, (const EQ*)a.e, ab.cache);
      } else {
        hipLaunchKernelGGL((k_rel_aggregate<false, 8>), gb, b1, 0, st, br.brow_ptr, br.brow_list, B, (const EQ*)a.e, ab.cache);
      }
      const double* lam = (f < 0) ? a.prior : a.prior + (size_t)(1 + f) * 2 * NG;
      const double* mu = lam + NG;
      float* param = (f < 0) ? h->tb.w : h->tb.V + f;
      const uint32_t pstride = (f < 0) ? h->tb.ws : h->tb.rs;
      const uint64_t stream_id = (uint64_t)(a.iter * 1024 + (f < 0 ? 1000 : f));
      const uint32_t nl = (uint32_t)ab.level_ptr.size() - 1;
      for (uint32_t l = 0; l < nl; l++) {
        const uint32_t cnt = ab.level_ptr[l + 1] - ab.level_ptr[l];
        if (!cnt) continue;
#define FMX_REL_DRAW(ISV, GG) FMX_LAUNCH_WAVES((k_rel_draw<ISV, GG>), ((uint64_t)cnt * GG + 63) / 64, st, br.rows.t_ent, br.rows.seg_feat,        \
          br.rows.seg_rel, br.rows.nseg, (uint32_t)br.rows.nnz, ab.level_list + ab.level_ptr[l], cnt, param, pstride, br.attr_offset,             \
          br.brow_ptr, B, ab.cache, opts->alpha, lam, mu, h->grp, opts->do_sample, opts->seed, stream_id)
        if (f < 0) { if (ab.lanes <= 4) FMX_REL_DRAW(false, 4); else if (ab.lanes == 8) FMX_REL_DRAW(false, 8); else if (ab.lanes == 16) FMX_REL_DRAW(false, 16); else FMX_REL_DRAW(false, 64); }
        else       { if (ab.lanes <= 4) FMX_REL_DRAW(true, 4);  else if (ab.lanes == 8) FMX_REL_DRAW(true, 8);  else if (ab.lanes == 16) FMX_REL_DRAW(true, 16);  else FMX_REL_DRAW(true, 64); }
#undef FMX_REL_DRAW
      }
      if (f >= 0) hipLaunchKernelGGL((k_rel_sync<true>), g1, b1, 0, st, br.map, N, B, (const double*)ab.cache, a.e);
      else        hipLaunchKernelGGL((k_rel_sync<false>), g1, b1, 0, st, br.map, N, B, (const double*)ab.cache, a.e);
      HIPCHK(h, hipGetLastError());
    }
    return FMX_OK;
  };
  const bool has_blocks = !sharded && !h->slots[a0.slot].blocks.empty();
  if (h->cfg.k1) {                                         // draw_w per level, :454-476
    for (uint32_t l = 0; l < n_levels; l++) { int rc = level_step(l, -1); if (rc) return rc; }
    if (has_blocks) { int rc = block_steps(-1); if (rc) return rc; }
    for (fmx_handle x : hs) {
      HIPCHK(x, hipSetDevice(x->device));
      const dim3 gu((uint32_t)std::min<uint64_t>((x->n_local + 255) / 256, 2048));
      hipLaunchKernelGGL(k_als_unseen, gu, b1, 0, x->stream, x->als.seen, x->n_local, x->tb.w, x->tb.ws, x->als.prior, x->als.prior + NG, x->grp,
                         opts->do_sample, opts->seed, (uint64_t)(x->als.iter * 1024 + 1001), make_shard(x->cfg));
    }
  }
  // the factors of the features with a training column, factor-major for the duration of the sweep (k_als_shadow)
  for (fmx_handle h : hs) {
    AlsState& a = h->als;
    const Slot& s = h->slots[a.slot];
    if (!a.vt || !s.nseg) continue;
    HIPCHK(h, hipSetDevice(h->device));
    KP_SWITCH(h->KP, hipLaunchKernelGGL((k_als_shadow<KP, true>), dim3(std::min<uint32_t>((s.nseg + 63) / 64, 4096)), dim3(256), 0, h->stream,
                                          a.level_list, s.seg_feat, s.nseg, h->tb, a.vt, a.vt_stride, (uint32_t)h->cfg.num_factor));
  }
  for (int f = 0; f < kf; f++) {                            // per factor: q_f is ready (k_als_eterms), draw_v per level :528-595
    for (fmx_handle x : hs) {
      HIPCHK(x, hipSetDevice(x->device));
      hipLaunchKernelGGL(k_als_load_q, g1, b1, 0, x->stream, x->als.e, x->als.q + (size_t)f * N, N);
    }
    for (uint32_t l = 0; l < n_levels; l++) { int rc = level_step(l, f); if (rc) return rc; }
    if (has_blocks) { int rc = block_steps(f); if (rc) return rc; }
  }
  for (fmx_handle h : hs) {
    AlsState& a = h->als;
    const Slot& s = h->slots[a.slot];
    HIPCHK(h, hipSetDevice(h->device));
    if (a.vt && s.nseg)
      KP_SWITCH(h->KP, hipLaunchKernelGGL((k_als_shadow<KP, false>), dim3(std::min<uint32_t>((s.nseg + 63) / 64, 4096)), dim3(256), 0, h->stream,
                                            a.level_list, s.seg_feat, s.nseg, h->tb, a.vt, a.vt_stride, (uint32_t)h->cfg.num_factor));
    if (kf > 0) {                                          // empty-row draws of every factor (:586-595) in one pass
      hipStream_t st = h->stream;
      KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_als_unseen_v<KP>), (h->n_local + Map<KP>::EPI - 1) / Map<KP>::EPI, st, a.seen, h->n_local,
                                         h->tb, kf, a.prior, NG, h->grp, opts->do_sample, opts->seed,
                                         (uint64_t)(a.iter * 1024 + 512), make_shard(h->cfg)));
    }
    HIPCHK(h, hipGetLastError());
  }
#undef FMX_ALS_DRAW
  // full re-prediction (fm_learn_mcmc_simultaneous.h:122), train metric and new residuals (:139-196)
  for (fmx_handle x : hs) {
    AlsState& a = x->als;
    HIPCHK(x, hipSetDevice(x->device));
    int rc = sharded ? als_eterms(x, x->slots[a.slot], a.e, a.q, a.epart) : als_repredict(x, x->slots[a.slot], a);
    if (rc) return rc;
  }
  if (sharded) {
    std::vector<double*> bq(P), be(P);
    for (size_t i = 0; i < P; i++) { bq[i] = hs[i]->als.q; be[i] = hs[i]->als.epart; }
    int rc = kf > 0 ? group_allreduce_f64(g, bq, (size_t)kf * N) : FMX_OK;      // q is [KP][N]: the first k factor rows
    if (rc == FMX_OK) rc = group_allreduce_f64(g, be, N);
    if (rc) return rc;
    for (fmx_handle x : hs) {
      HIPCHK(x, hipSetDevice(x->device));
      hipLaunchKernelGGL(k_als_set_e, g1, b1, 0, x->stream, x->als.e, x->als.epart, x->als.q, x->cfg.num_factor, N, x->cfg.k0, x->w0);
    }
  }
  for (fmx_handle x : hs) {
    AlsState& a = x->als;
    const Slot& s = x->slots[a.slot];
    HIPCHK(x, hipSetDevice(x->device));
    HIPCHK(x, hipMemsetAsync(x->acc, 0, 4 * sizeof(double), x->stream));
    hipLaunchKernelGGL(k_als_targets, g1, b1, 0, x->stream, a.e, s.target, N, x->cfg.task, x->cfg.min_target, x->cfg.max_target, x->acc,
                       opts->do_sample, opts->seed, (uint64_t)(a.iter * 1024 + 1002));
    HIPCHK(x, hipGetLastError());
  }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipMemcpyAsync(acc, h->acc, sizeof(acc), hipMemcpyDeviceToHost, h->stream));
  for (fmx_handle x : hs) { HIPCHK(x, hipSetDevice(x->device)); HIPCHK(x, hipStreamSynchronize(x->stream)); x->als.iter++; }
  if (stats) {
    float ms = 0;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    stats->device_seconds = ms * 1e-3;
    stats->levels = n_levels;
    stats->train_metric = (h->cfg.task == FMX_TASK_REGRESSION) ? std::sqrt(acc[0] / N) : acc[0] / N;
  }
  return FMX_OK;
}

// ---- fm_learn_mcmc over feature shards (fmx_group) --------------------------------------------------------------
// fmx_group_als_begin: every shard builds X^T of its own features; the dependency LEVELS are computed over the union of
// the shards' columns in GLOBAL feature order (level(j) = 1 + max level of the smaller-id features sharing a row with j),
// so that "level by level, every shard its features of the level" is exactly the reference's sequential sweep.
int fmx_group_als_begin(fmx_group g, int train_slot) {
  if (!g) return FMX_E_ARG;
  for (fmx_handle x : g->hs) if (!x) return FMX_E_STATE;
  fmx_handle h = g->hs[0];
  if (g->kind == GROUP_SINGLE) { int rc = fmx_als_begin(h, train_slot); if (rc) g->err = h->err; return rc; }
  const size_t P = g->hs.size();
  struct Col { uint32_t gid, shard, seg; };
  std::vector<Col> cols;
  std::vector<std::vector<uint32_t>> seg_feat(P), seg_rel(P);
  std::vector<std::vector<TEntry>> tent(P);
  uint32_t N = 0;
  auto bail = [&](int rc, fmx_handle x) { g->err = x->err; for (fmx_handle y : g->hs) als_free(y); return rc; };
  for (size_t i = 0; i < P; i++) {
    fmx_handle x = g->hs[i];
    int rc = check_slot(x, train_slot, true);
    if (rc == FMX_OK) rc = lag_flush(x);
    if (rc) return bail(rc, x);
    if (hipSetDevice(x->device) != hipSuccess || hipStreamSynchronize(x->stream) != hipSuccess) return bail(fail(x, FMX_E_HIP, "fmx_group_als_begin: device %d", x->device), x);
    als_free(x);
    Slot& s = x->slots[train_slot];
    if (s.n_rows == 0) return bail(fail(x, FMX_E_ARG, "fmx_group_als_begin: empty training set"), x);
    if (!s.blocks.empty()) return bail(fail(x, FMX_E_UNSUPPORTED, "fmx_group_als_begin: block-structured rows on feature shards are not implemented"), x);
    if (i == 0) N = s.n_rows; else if (s.n_rows != N) return bail(fail(x, FMX_E_STATE, "the shards hold different numbers of rows"), x);
    rc = ensure_segments(x, s, s.n_rows);
    if (rc) return bail(rc, x);
    x->als.slot = train_slot;
    const uint32_t nseg = s.nseg;
    seg_feat[i].resize(nseg); seg_rel[i].resize((size_t)nseg + 1); tent[i].resize((size_t)s.nnz);
    if (nseg) {
      if (hipMemcpy(seg_feat[i].data(), s.seg_feat, (size_t)nseg * 4, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(seg_rel[i].data(), s.seg_rel, (size_t)nseg * 4, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(tent[i].data(), s.t_ent, (size_t)s.nnz * sizeof(TEntry), hipMemcpyDeviceToHost) != hipSuccess)
        return bail(fail(x, FMX_E_HIP, "fmx_group_als_begin: copying the columns of shard %zu failed", i), x);
    }
    seg_rel[i][nseg] = (uint32_t)s.nnz;
    const Shard sh = make_shard(x->cfg);
    for (uint32_t sg = 0; sg < nseg; sg++) cols.push_back(Col{sh.global(seg_feat[i][sg]), (uint32_t)i, sg});
  }
  std::sort(cols.begin(), cols.end(), [](const Col& a, const Col& b) { return a.gid < b.gid; });
  std::vector<uint32_t> rowlevel(N, 0);
  std::vector<std::vector<uint32_t>> lvl(P);
  for (size_t i = 0; i < P; i++) lvl[i].resize(seg_feat[i].size());
  uint32_t n_levels = 0;
  for (const Col& c : cols) {
    const auto& rel = seg_rel[c.shard];
    const auto& te = tent[c.shard];
    uint32_t l = 0;
    for (uint32_t q = rel[c.seg]; q < rel[c.seg + 1]; q++) l = std::max(l, rowlevel[te[q].e]);
    l += 1;
    for (uint32_t q = rel[c.seg]; q < rel[c.seg + 1]; q++) rowlevel[te[q].e] = l;
    lvl[c.shard][c.seg] = l - 1;
    n_levels = std::max(n_levels, l);
  }
  for (size_t i = 0; i < P; i++) {
    fmx_handle x = g->hs[i];
    AlsState& a = x->als;
    Slot& s = x->slots[train_slot];
    const uint32_t nseg = s.nseg;
    a.level_ptr.assign((size_t)n_levels + 1, 0);                  // the SAME number of levels on every shard (some may be empty)
    for (uint32_t sg = 0; sg < nseg; sg++) a.level_ptr[lvl[i][sg] + 1]++;
    for (uint32_t l = 0; l < n_levels; l++) a.level_ptr[l + 1] += a.level_ptr[l];
    std::vector<uint32_t> list(std::max<uint32_t>(nseg, 1)), fill(a.level_ptr.begin(), a.level_ptr.end());
    for (uint32_t sg = 0; sg < nseg; sg++) list[fill[lvl[i][sg]]++] = sg;
    std::vector<uint8_t> seen((size_t)x->n_local, 0);
    for (uint32_t sg = 0; sg < nseg; sg++) seen[seg_feat[i][sg]] = 1;
    hipError_t er = hipSetDevice(x->device);
    if (er == hipSuccess) er = fmx_dev_alloc(&a.level_list, list.size() * 4);
    if (er == hipSuccess) er = hipMemcpy(a.level_list, list.data(), list.size() * 4, hipMemcpyHostToDevice);
    if (er == hipSuccess) er = fmx_dev_alloc(&a.seen, seen.size());
    if (er == hipSuccess) er = hipMemcpy(a.seen, seen.data(), seen.size(), hipMemcpyHostToDevice);
    if (er == hipSuccess) er = fmx_dev_alloc(&a.e, (size_t)N * sizeof(EQ));
    if (er == hipSuccess) er = fmx_dev_alloc(&a.q, (size_t)N * (size_t)x->KP * sizeof(double));
    if (er == hipSuccess) er = fmx_dev_alloc(&a.delta, (size_t)N * sizeof(EQ));
    if (er == hipSuccess) er = hipMemsetAsync(a.delta, 0, (size_t)N * sizeof(EQ), x->stream);
    if (er == hipSuccess) er = fmx_dev_alloc(&a.epart, (size_t)N * sizeof(double));
    if (er == hipSuccess && x->cfg.num_factor > 0 && nseg > 0) {
      a.vt_stride = ((size_t)nseg + 63) & ~(size_t)63;
      er = fmx_dev_alloc(&a.vt, (size_t)x->cfg.num_factor * a.vt_stride * sizeof(float));
    }
    if (er != hipSuccess) return bail(fail(x, FMX_E_HIP, "fmx_group_als_begin: %s", hipGetErrorString(er)), x);
    // first prediction (fm_learn_mcmc_simultaneous.h:69-86): partial sums of this shard
    int rc = als_eterms(x, s, a.e, a.q, a.epart);
    if (rc) return bail(rc, x);
  }
  {
    std::vector<double*> bq(P), be(P);
    for (size_t i = 0; i < P; i++) { bq[i] = g->hs[i]->als.q; be[i] = g->hs[i]->als.epart; }
    int rc = h->cfg.num_factor > 0 ? group_allreduce_f64(g, bq, (size_t)h->cfg.num_factor * N) : FMX_OK;
    if (rc == FMX_OK) rc = group_allreduce_f64(g, be, N);
    if (rc) return bail(rc, h);
  }
  const dim3 g1(std::min<uint32_t>((N + 255) / 256, 2048)), b1(256);
  for (fmx_handle x : g->hs) {
    hipSetDevice(x->device);
    hipLaunchKernelGGL(k_als_set_e, g1, b1, 0, x->stream, x->als.e, x->als.epart, x->als.q, x->cfg.num_factor, N, x->cfg.k0, x->w0);
    hipLaunchKernelGGL(k_als_sub_target, g1, b1, 0, x->stream, x->als.e, x->slots[train_slot].target, N);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(x->stream) != hipSuccess)
      return bail(fail(x, FMX_E_HIP, "fmx_group_als_begin: first prediction failed on shard device %d", x->device), x);
  }
  return FMX_OK;
}

int fmx_group_als_sweep(fmx_group g, const fmx_als_opts* opts, fmx_als_stats* stats) {
  if (!g || !opts) return FMX_E_ARG;
  for (fmx_handle x : g->hs) if (!x) return FMX_E_STATE;
  int rc = als_sweep_shards(g->hs, g->kind == GROUP_SINGLE ? nullptr : g, opts, stats);
  if (rc) g->err = g->hs[0]->err;
  return rc;
}

// fmx_als_moments over the shards: the residual statistics are those of the (replicated) cache, the per-group parameter
// sums add up over the shards
int fmx_group_als_moments(fmx_group g, double* out) {
  if (!g || !out) return FMX_E_ARG;
  for (auto m : g->hs) if (!m) { g->err = "a member of the group was destroyed"; return FMX_E_STATE; }
  fmx_handle h = g->hs[0];
  const size_t cnt = 2 + 2 * (size_t)h->num_groups * (size_t)(1 + h->cfg.num_factor);
  std::vector<double> part(cnt);
  for (size_t i = 0; i < g->hs.size(); i++) {
    int rc = fmx_als_moments(g->hs[i], part.data());
    if (rc) { g->err = g->hs[i]->err; return rc; }
    if (i == 0) memcpy(out, part.data(), cnt * sizeof(double));
    else for (size_t c = 2; c < cnt; c++) out[c] += part[c];
  }
  return FMX_OK;
}

int fmx_group_als_end(fmx_group g) {
  if (!g) return FMX_E_ARG;
  for (fmx_handle x : g->hs) if (x) { int rc = fmx_als_end(x); if (rc) { g->err = x->err; return rc; } }
  return FMX_OK;
}

int fmx_als_sweep(fmx_handle h, const fmx_als_opts* opts, fmx_als_stats* stats) {
  if (!h || !opts) return FMX_E_ARG;
  if (h->cfg.shard_world > 1) return fail(h, FMX_E_STATE, "fmx_als_sweep on a feature shard: use fmx_group_als_sweep");
  return als_sweep_shards(std::vector<fmx_handle>(1, h), nullptr, opts, stats);
}

}  // extern "C"

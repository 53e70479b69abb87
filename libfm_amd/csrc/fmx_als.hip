// fmx_als.hip -- C-ABI (include/fmx.h): the ALS / MCMC learner (level-scheduled coordinate sweeps, fmx_als_kernels.h).
#include "fmx_internal.h"

extern "C" {

// ---------------------------------------------------------------------------------------------
// ALS / MCMC
// ---------------------------------------------------------------------------------------------
extern "C++" void als_free(fmx_handle h) {
  AlsState& a = h->als;
  if (a.e) hipFree(a.e);
  if (a.q) hipFree(a.q);
  if (a.seen) hipFree(a.seen);
  if (a.level_list) hipFree(a.level_list);
  if (a.prior) hipFree(a.prior);
  if (a.vt) hipFree(a.vt);
  a = AlsState();
}

int fmx_als_end(fmx_handle h) {
  if (!h) return FMX_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  als_free(h);
  return FMX_OK;
}

static int als_eterms(fmx_handle h, const Slot& s, EQ* e, double* q) {
  KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_als_eterms<KP>), s.n_rows, h->stream, s.ent, s.row_ptr, s.n_rows, h->tb,
                                     h->cfg.k0, h->cfg.k1, h->w0, e, q));
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

static int als_begin_impl(fmx_handle h, int train_slot);
int fmx_als_begin(fmx_handle h, int train_slot) {
  const int rc = als_begin_impl(h, train_slot);
  if (rc != FMX_OK && h && h->als.slot == train_slot) als_free(h);      // a failed set-up leaves no half-built session behind
  return rc;
}
static int als_begin_impl(fmx_handle h, int train_slot) {
  int rc = check_slot(h, train_slot, true);
  if (rc) return rc;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  if (h->cfg.shard_world > 1) return fail(h, FMX_E_UNSUPPORTED, "ALS on a feature shard is not implemented");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  als_free(h);
  Slot& s = h->slots[train_slot];
  if (s.n_rows == 0) return fail(h, FMX_E_ARG, "fmx_als_begin: empty training set");
  rc = ensure_segments(h, s, s.n_rows);          // one "batch" = the whole data set: X^T with columns in id order
  if (rc) return rc;
  AlsState& a = h->als;
  a.slot = train_slot;
  const uint32_t N = s.n_rows, nseg = s.nseg;
  // ---- dependency levels (host, O(nnz)): level(j) = 1 + max level of earlier features sharing a row with j
  std::vector<uint32_t> seg_feat(nseg), seg_rel(nseg + 1), lvl(nseg), rowlevel(N, 0);
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
    if (getenv("FMX_ALS_SEQUENTIAL")) l = sg + 1;      // debugging aid: one feature per level (the reference's order, serial)
    for (uint32_t i = seg_rel[sg]; i < seg_rel[sg + 1]; i++) rowlevel[tent[i].e] = l;
    lvl[sg] = l - 1;
    n_levels = std::max(n_levels, l);
  }
  a.level_ptr.assign((size_t)n_levels + 1, 0);
  for (uint32_t sg = 0; sg < nseg; sg++) a.level_ptr[lvl[sg] + 1]++;
  for (uint32_t l = 0; l < n_levels; l++) a.level_ptr[l + 1] += a.level_ptr[l];
  std::vector<uint32_t> list(std::max<uint32_t>(nseg, 1)), fill(a.level_ptr.begin(), a.level_ptr.end());
  for (uint32_t sg = 0; sg < nseg; sg++) list[fill[lvl[sg]]++] = sg;
  std::vector<uint8_t> seen((size_t)h->n_local, 0);
  for (uint32_t sg = 0; sg < nseg; sg++) seen[seg_feat[sg]] = 1;
  HIPCHK(h, hipMalloc(&a.level_list, list.size() * 4));
  HIPCHK(h, hipMemcpy(a.level_list, list.data(), list.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMalloc(&a.seen, seen.size()));
  HIPCHK(h, hipMemcpy(a.seen, seen.data(), seen.size(), hipMemcpyHostToDevice));
  HIPCHK(h, hipMalloc(&a.e, (size_t)N * sizeof(EQ)));
  HIPCHK(h, hipMalloc(&a.q, (size_t)N * (size_t)h->KP * sizeof(double)));
  if (h->cfg.num_factor > 0 && nseg > 0 && !getenv("FMX_ALS_NO_SHADOW")) {      // (the env switch is the A/B knob of the profile)
    a.vt_stride = ((size_t)nseg + 63) & ~(size_t)63;
    HIPCHK(h, hipMalloc(&a.vt, (size_t)h->cfg.num_factor * a.vt_stride * sizeof(float)));
  }
  // ---- first prediction and e -= target (fm_learn_mcmc_simultaneous.h:69-86)
  rc = als_eterms(h, s, a.e, a.q);
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
  HIPCHK(h, hipMalloc(&d, (2 + cells) * sizeof(double)));
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
  hipFree(d);
  if (er != hipSuccess) return fail(h, FMX_E_HIP, "fmx_als_moments: %s", hipGetErrorString(er));
  out[0] = host[1]; out[1] = host[0];                                          // documented order: sum e^2 first
  memcpy(out + 2, host.data() + 2, (size_t)(1 + k) * G * 2 * sizeof(double)); // rows 0..k of [1 + KP][G][2]
  return rc;
}

int fmx_als_sweep(fmx_handle h, const fmx_als_opts* opts, fmx_als_stats* stats) {
  if (!h || !opts) return FMX_E_ARG;
  AlsState& a = h->als;
  if (a.slot < 0) return fail(h, FMX_E_STATE, "fmx_als_sweep before fmx_als_begin");
  if (opts->num_groups != 0 && opts->num_groups != h->num_groups)     // validated BEFORE the first device write of the sweep
    return fail(h, FMX_E_ARG, "fmx_als_sweep: opts->num_groups = %u but the handle has %u attribute groups", opts->num_groups, h->num_groups);
  HIPCHK(h, hipSetDevice(h->device));
  const Slot& s = h->slots[a.slot];
  const uint32_t N = s.n_rows;
  const uint32_t n_levels = (uint32_t)a.level_ptr.size() - 1;
  hipStream_t st = h->stream;
  const dim3 g1(std::min<uint32_t>((N + 255) / 256, 2048)), b1(256);
  HIPCHK(h, hipEventRecord(h->ev0, st));
  double acc[4] = {0, 0, 0, 0};
  // sum e, sum e^2 (draw_w0's numerator; draw_alpha's statistic for the caller)
  HIPCHK(h, hipMemsetAsync(h->acc, 0, 4 * sizeof(double), st));
  hipLaunchKernelGGL(k_als_sum_e, g1, b1, 0, st, a.e, N, h->acc);
  HIPCHK(h, hipMemcpyAsync(acc, h->acc, sizeof(acc), hipMemcpyDeviceToHost, st));
  double w0 = 0;
  HIPCHK(h, hipMemcpyAsync(&w0, h->w0, sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (stats) stats->sum_e_sqr = acc[1];
  std::mt19937_64 rng(opts->seed * 0x9E3779B97F4A7C15ull + a.iter + 1);
  std::normal_distribution<double> nd(0.0, 1.0);
  if (h->cfg.k0) {                                         // draw_w0, fm_learn_mcmc.h:643-683 (w0_mean_0 = 0)
    double mean = acc[0] - (double)N * w0;
    const double sigma_sqr = 1.0 / (h->cfg.reg0 + opts->alpha * (double)N);
    mean = -sigma_sqr * (opts->alpha * mean - 0.0 * h->cfg.reg0);
    double nw0 = opts->do_sample ? mean + std::sqrt(sigma_sqr) * nd(rng) : mean;
    if (!(std::isnan(nw0) || std::isinf(nw0))) {
      HIPCHK(h, hipMemcpyAsync(h->w0, &nw0, sizeof(double), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_als_add_const, g1, b1, 0, st, a.e, N, nw0 - w0);
      HIPCHK(h, hipStreamSynchronize(st));                 // nw0 lives on this stack frame
    }
  }
  // ---- priors per coordinate family and attribute group: [1 + k][2][NG] = lambda[NG] then mu[NG]
  const uint32_t NG = h->num_groups;
  const int kf = h->cfg.num_factor;
  const bool tabs = opts->num_groups != 0;
  a.prior_host.resize((size_t)(1 + kf) * 2 * NG);
  for (uint32_t g = 0; g < NG; g++) {
    a.prior_host[g] = (tabs && opts->w_lambda_g) ? opts->w_lambda_g[g] : opts->w_lambda;
    a.prior_host[NG + g] = (tabs && opts->w_mu_g) ? opts->w_mu_g[g] : opts->w_mu;
    for (int f = 0; f < kf; f++) {
      double* row = a.prior_host.data() + (size_t)(1 + f) * 2 * NG;
      row[g] = (tabs && opts->v_lambda_gf) ? opts->v_lambda_gf[(size_t)g * kf + f] : (opts->v_lambda_f ? opts->v_lambda_f[f] : opts->v_lambda);
      row[NG + g] = (tabs && opts->v_mu_gf) ? opts->v_mu_gf[(size_t)g * kf + f] : (opts->v_mu_f ? opts->v_mu_f[f] : opts->v_mu);
    }
  }
  if (!a.prior) HIPCHK(h, hipMalloc(&a.prior, a.prior_host.size() * sizeof(double)));
  HIPCHK(h, hipMemcpyAsync(a.prior, a.prior_host.data(), a.prior_host.size() * sizeof(double), hipMemcpyHostToDevice, st));
  const uint32_t nseg = s.nseg, nnz = (uint32_t)s.nnz;
  const dim3 gu((uint32_t)std::min<uint64_t>((h->n_local + 255) / 256, 2048));
  // lanes per column from the mean column length (one-hot data: a handful of rows per feature)
  const double avg_col = nseg ? (double)nnz / (double)nseg : 0.0;
  const int G = avg_col <= 5.0 ? 4 : (avg_col <= 12.0 ? 8 : (avg_col <= 40.0 ? 16 : 64));
#define FMX_ALS_DRAW(ISV, cnt, ...)                                                                          \
  do {                                                                                                        \
    if (G == 4)       FMX_LAUNCH_WAVES((k_als_draw<ISV, 4>), ((uint64_t)(cnt) + 15) / 16, st, __VA_ARGS__);    \
    else if (G == 8)  FMX_LAUNCH_WAVES((k_als_draw<ISV, 8>), ((uint64_t)(cnt) + 7) / 8, st, __VA_ARGS__);      \
    else if (G == 16) FMX_LAUNCH_WAVES((k_als_draw<ISV, 16>), ((uint64_t)(cnt) + 3) / 4, st, __VA_ARGS__);     \
    else              FMX_LAUNCH_WAVES((k_als_draw<ISV, 64>), (uint64_t)(cnt), st, __VA_ARGS__);               \
  } while (0)
  if (h->cfg.k1) {                                         // draw_w per level, :454-476
    for (uint32_t l = 0; l < n_levels; l++) {
      const uint32_t cnt = a.level_ptr[l + 1] - a.level_ptr[l];
      if (!cnt) continue;
      FMX_ALS_DRAW(false, cnt, s.t_ent, s.seg_feat, s.seg_rel, nseg, nnz, a.level_list + a.level_ptr[l], cnt,
                   h->tb.w, h->tb.ws, 0, 0u, a.e, opts->alpha, a.prior, a.prior + NG, h->grp, opts->do_sample,
                   opts->seed, (uint64_t)(a.iter * 1024 + 1000));
    }
    hipLaunchKernelGGL(k_als_unseen, gu, b1, 0, st, a.seen, h->n_local, h->tb.w, h->tb.ws, a.prior, a.prior + NG, h->grp,
                       opts->do_sample, opts->seed, (uint64_t)(a.iter * 1024 + 1001));
  }
  // the factors of the features with a training column, factor-major for the duration of the sweep (k_als_shadow)
  const bool shadow = a.vt != nullptr && nseg > 0 && h->cfg.num_factor > 0;
  if (shadow) {
    KP_SWITCH(h->KP, hipLaunchKernelGGL((k_als_shadow<KP, true>), dim3(std::min<uint32_t>((nseg + 63) / 64, 4096)), dim3(256), 0, st,
                                          a.level_list, s.seg_feat, nseg, h->tb, a.vt, a.vt_stride));
  }
  for (int f = 0; f < h->cfg.num_factor; f++) {            // per factor: q_f is ready (k_als_eterms), draw_v per level :528-595
    double* qf = a.q + (size_t)f * N;
    const double* v_lambda = a.prior + (size_t)(1 + f) * 2 * NG;
    const double* v_mu = v_lambda + NG;
    hipLaunchKernelGGL(k_als_load_q, g1, b1, 0, st, a.e, qf, N);
    for (uint32_t l = 0; l < n_levels; l++) {
      const uint32_t cnt = a.level_ptr[l + 1] - a.level_ptr[l];
      if (!cnt) continue;
      if (shadow)
        FMX_ALS_DRAW(true, cnt, s.t_ent, s.seg_feat, s.seg_rel, nseg, nnz, a.level_list + a.level_ptr[l], cnt,
                     a.vt + (size_t)f * a.vt_stride, 1u, 1, a.level_ptr[l], a.e, opts->alpha, v_lambda, v_mu, h->grp, opts->do_sample,
                     opts->seed, (uint64_t)(a.iter * 1024 + f));
      else
        FMX_ALS_DRAW(true, cnt, s.t_ent, s.seg_feat, s.seg_rel, nseg, nnz, a.level_list + a.level_ptr[l], cnt,
                     h->tb.V + f, h->tb.rs, 0, 0u, a.e, opts->alpha, v_lambda, v_mu, h->grp, opts->do_sample,
                     opts->seed, (uint64_t)(a.iter * 1024 + f));
    }
  }
  if (shadow) {
    KP_SWITCH(h->KP, hipLaunchKernelGGL((k_als_shadow<KP, false>), dim3(std::min<uint32_t>((nseg + 63) / 64, 4096)), dim3(256), 0, st,
                                          a.level_list, s.seg_feat, nseg, h->tb, a.vt, a.vt_stride));
  }
  if (h->cfg.num_factor > 0) {                             // empty-row draws of every factor (:586-595) in one pass
    KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_als_unseen_v<KP>), (h->n_local + Map<KP>::EPI - 1) / Map<KP>::EPI, st, a.seen, h->n_local,
                                       h->tb, h->cfg.num_factor, a.prior, NG, h->grp, opts->do_sample, opts->seed,
                                       (uint64_t)(a.iter * 1024 + 512)));
  }
  HIPCHK(h, hipGetLastError());
  // full re-prediction (fm_learn_mcmc_simultaneous.h:122), train metric and new residuals (:139-196)
  int rc = als_eterms(h, s, a.e, a.q);
  if (rc) return rc;
  HIPCHK(h, hipMemsetAsync(h->acc, 0, 4 * sizeof(double), st));
  hipLaunchKernelGGL(k_als_targets, g1, b1, 0, st, a.e, s.target, N, h->cfg.task, h->cfg.min_target, h->cfg.max_target, h->acc,
                     opts->do_sample, opts->seed, (uint64_t)(a.iter * 1024 + 1002));
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev1, st));
  HIPCHK(h, hipMemcpyAsync(acc, h->acc, sizeof(acc), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  a.iter++;
  if (stats) {
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    stats->device_seconds = ms * 1e-3;
    stats->levels = n_levels;
    stats->train_metric = (h->cfg.task == FMX_TASK_REGRESSION) ? std::sqrt(acc[0] / N) : acc[0] / N;
  }
  return FMX_OK;
}

}  // extern "C"

/*
 * fm_oracle_als.c -- CPU restatement of the ALS (coordinate descent) learner.
 * TEST INFRASTRUCTURE ONLY (see fm_oracle.h).
 */
#include "fm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Data::create_data_t                          /root/reference/src/libfm/src/Data.h:292-341
 * counting sort of the entries by feature id; within a feature the row ids ascend. */
void fmo_transpose(const fmo_data *d, uint64_t n, fmo_entry **entries_t, uint64_t **col_ptr) {
  uint64_t nnz = d->row_ptr[d->n_rows];
  uint64_t *cp = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
  fmo_entry *et = (fmo_entry *)malloc(sizeof(fmo_entry) * (size_t)(nnz ? nnz : 1));
  for (uint64_t i = 0; i < nnz; i++) cp[d->entries[i].id + 1]++;
  for (uint64_t j = 0; j < n; j++) cp[j + 1] += cp[j];
  uint64_t *fill = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
  memcpy(fill, cp, sizeof(uint64_t) * (size_t)n);
  for (uint32_t r = 0; r < d->n_rows; r++)
    for (uint64_t i = d->row_ptr[r]; i < d->row_ptr[r + 1]; i++) {
      uint64_t pos = fill[d->entries[i].id]++;
      et[pos].id = r;
      et[pos].value = d->entries[i].value;
    }
  free(fill);
  *entries_t = et;
  *col_ptr = cp;
}

#define VV(m, f, j) ((m)->v[(size_t)(f) * (size_t)(m)->n + (size_t)(j)])

/* erf / cdf_gaussian                                       /root/reference/src/util/random.h:45-67 */
double fmo_erf(double x) {
  double t;
  if (x >= 0) t = 1.0 / (1.0 + 0.3275911 * x);
  else        t = 1.0 / (1.0 - 0.3275911 * x);
  double result = 1.0 - (t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429))))) * exp(-x * x);
  return (x >= 0) ? result : -result;
}
double fmo_cdf_gaussian(double x) { return 0.5 + 0.5 * fmo_erf(0.707106781 * x); }

/* predict_data_and_write_to_eterms                          /root/reference/src/libfm/src/fm_learn_mcmc.h:148-378
 * (non-relational branches :176-201, :229-240, :255-281, :308-329, :353-368), one data set at a time: the passes of
 * different data sets never mix values, so per-data-set evaluation gives the same numbers. */
void fmo_als_predict_eterms(const fmo_model *m, const fmo_data_t *dt, fmo_eq *cache) {
  const uint32_t N = dt->n_rows;
  for (uint32_t c = 0; c < N; c++) { cache[c].e = 0.0; cache[c].q = 0.0; }          /* :157-164 */
  for (int f = 0; f < m->k; f++) {                                                   /* :176-240 */
    for (uint64_t j = 0; j < dt->n; j++) {
      const double v_if = VV(m, f, j);
      for (uint64_t i = dt->col_ptr[j]; i < dt->col_ptr[j + 1]; i++)
        cache[dt->entries[i].id].q += v_if * dt->entries[i].value;
    }
    for (uint32_t c = 0; c < N; c++) {
      double q_all = cache[c].q;
      cache[c].e += 0.5 * q_all * q_all;
      cache[c].q = 0.0;
    }
  }
  for (int f = 0; f < m->k; f++) {                                                   /* :255-281 */
    for (uint64_t j = 0; j < dt->n; j++) {
      const double v_if = VV(m, f, j);
      for (uint64_t i = dt->col_ptr[j]; i < dt->col_ptr[j + 1]; i++) {
        const float x = dt->entries[i].value;
        cache[dt->entries[i].id].q -= 0.5 * v_if * v_if * x * x;
      }
    }
  }
  if (m->k1) {                                                                       /* :308-329 */
    for (uint64_t j = 0; j < dt->n; j++) {
      const double w_i = m->w[j];
      for (uint64_t i = dt->col_ptr[j]; i < dt->col_ptr[j + 1]; i++)
        cache[dt->entries[i].id].q += w_i * dt->entries[i].value;
    }
  }
  for (uint32_t c = 0; c < N; c++) {                                                 /* :353-368 */
    double q_all = cache[c].q;
    cache[c].e = cache[c].e + q_all;
    if (m->k0) cache[c].e += m->w0;
    cache[c].q = 0.0;
  }
}

/* draw_w0 / draw_w / draw_v with do_sample = 0 and alpha = 1 (fm_learn_mcmc.h:643-683, 685-732, 792-847) */
static void als_draw_w(double *w, double w_mu, double w_lambda, double alpha,
                       const fmo_entry *col, uint64_t size, fmo_eq *cache) {
  double w_sigma_sqr = 0, w_mean = 0;
  for (uint64_t i = 0; i < size; i++) {
    float x_li = col[i].value;
    w_mean += x_li * (cache[col[i].id].e - (*w) * x_li);
    w_sigma_sqr += x_li * x_li;
  }
  w_sigma_sqr = (double)1.0 / (w_lambda + alpha * w_sigma_sqr);
  w_mean = -w_sigma_sqr * (alpha * w_mean - w_mu * w_lambda);
  double w_old = *w;
  if (isnan(w_sigma_sqr) || isinf(w_sigma_sqr)) *w = 0.0; else *w = w_mean;
  if (isnan(*w) || isinf(*w)) { *w = w_old; return; }
  for (uint64_t i = 0; i < size; i++) {
    double h = col[i].value;
    cache[col[i].id].e -= h * (w_old - *w);
  }
}

static void als_draw_v(double *v, double v_mu, double v_lambda, double alpha,
                       const fmo_entry *col, uint64_t size, fmo_eq *cache) {
  double v_sigma_sqr = 0, v_mean = 0;
  for (uint64_t i = 0; i < size; i++) {
    float x_li = col[i].value;
    fmo_eq *c = &cache[col[i].id];
    double h = x_li * (c->q - x_li * (*v));
    v_mean += h * c->e;
    v_sigma_sqr += h * h;
  }
  v_mean -= (*v) * v_sigma_sqr;
  v_sigma_sqr = (double)1.0 / (v_lambda + alpha * v_sigma_sqr);
  v_mean = -v_sigma_sqr * (alpha * v_mean - v_mu * v_lambda);
  double v_old = *v;
  if (isnan(v_sigma_sqr) || isinf(v_sigma_sqr)) *v = 0.0; else *v = v_mean;
  if (isnan(*v) || isinf(*v)) { *v = v_old; return; }
  for (uint64_t i = 0; i < size; i++) {
    float x_li = col[i].value;
    fmo_eq *c = &cache[col[i].id];
    double h = x_li * (c->q - x_li * v_old);
    c->q -= x_li * (v_old - *v);
    c->e -= h * (v_old - *v);
  }
}

/* draw_all with do_sample = 0 (fm_learn_mcmc.h:430-641); lambda per attribute group: w_lambda(g), v_lambda(g,f)
 * (:464-466, :583-585), g = meta->attr_group(feature) */
void fmo_als_sweep_groups(fmo_model *m, const fmo_data_t *dt, fmo_eq *cache, const fmo_als_reg *reg) {
  const double alpha = 1.0, mu = 0.0;                      /* fm_learn_mcmc.h:1106-1112, draw_alpha :912-915 */
  const uint32_t N = dt->n_rows;
  const int k = m->k;
#define AG(j) (reg->group ? reg->group[(j)] : 0u)
  if (m->k0) {                                             /* draw_w0 :643-683 (reg = fm->reg0, w0_mean_0 = 0) */
    double w0_mean = 0;
    for (uint32_t i = 0; i < N; i++) w0_mean += cache[i].e - m->w0;
    double w0_sigma_sqr = (double)1.0 / (m->reg0 + alpha * N);
    w0_mean = -w0_sigma_sqr * (alpha * w0_mean - 0.0 * m->reg0);
    double w0_old = m->w0;
    m->w0 = w0_mean;
    if (isnan(m->w0) || isinf(m->w0)) m->w0 = w0_old;
    else for (uint32_t i = 0; i < N; i++) cache[i].e -= (w0_old - m->w0);
  }
  if (m->k1) {                                             /* :441-476 */
    for (uint64_t j = 0; j < dt->n; j++)
      als_draw_w(&m->w[j], mu, reg->w_lambda[AG(j)], alpha, dt->entries + dt->col_ptr[j], dt->col_ptr[j + 1] - dt->col_ptr[j], cache);
    for (uint64_t j = dt->n; j < m->n; j++) als_draw_w(&m->w[j], mu, reg->w_lambda[AG(j)], alpha, NULL, 0, cache);
  }
  for (int f = 0; f < k; f++) {                            /* :528-595 */
    for (uint32_t c = 0; c < N; c++) cache[c].q = 0.0;
    for (uint64_t j = 0; j < dt->n; j++) {                 /* add_main_q :406-428 */
      const double v_if = VV(m, f, j);
      for (uint64_t i = dt->col_ptr[j]; i < dt->col_ptr[j + 1]; i++)
        cache[dt->entries[i].id].q += v_if * dt->entries[i].value;
    }
    for (uint64_t j = 0; j < dt->n; j++)
      als_draw_v(&VV(m, f, j), mu, reg->v_lambda[(size_t)AG(j) * k + f], alpha, dt->entries + dt->col_ptr[j], dt->col_ptr[j + 1] - dt->col_ptr[j], cache);
    for (uint64_t j = dt->n; j < m->n; j++) als_draw_v(&VV(m, f, j), mu, reg->v_lambda[(size_t)AG(j) * k + f], alpha, NULL, 0, cache);
  }
#undef AG
}

void fmo_als_sweep(fmo_model *m, const fmo_data_t *dt, fmo_eq *cache, double w_lambda, double v_lambda) {
  double *vl = (double *)malloc(sizeof(double) * (size_t)(m->k > 0 ? m->k : 1));
  for (int f = 0; f < m->k; f++) vl[f] = v_lambda;
  fmo_als_reg reg = { NULL, 1, &w_lambda, vl };
  fmo_als_sweep_groups(m, dt, cache, &reg);
  free(vl);
}

void fmo_als_learn(fmo_model *m, const fmo_data *train, const fmo_data *test, int task, int num_iter,
                   double w_lambda, double v_lambda, double min_target, double max_target,
                   double *test_pred_this, double *train_metric) {
  double *vl = (double *)malloc(sizeof(double) * (size_t)(m->k > 0 ? m->k : 1));
  for (int f = 0; f < m->k; f++) vl[f] = v_lambda;
  fmo_als_reg reg = { NULL, 1, &w_lambda, vl };
  fmo_als_learn_groups(m, train, test, task, num_iter, &reg, min_target, max_target, test_pred_this, train_metric);
  free(vl);
}

void fmo_als_learn_groups(fmo_model *m, const fmo_data *train, const fmo_data *test, int task, int num_iter,
                          const fmo_als_reg *reg, double min_target, double max_target,
                          double *test_pred_this, double *train_metric) {
  /* X^T of each data set has as many rows as THAT data set has features (Data.h:300-301) */
  uint64_t n_train = 0, n_test = 0;
  for (uint64_t i = 0; i < train->row_ptr[train->n_rows]; i++) if (train->entries[i].id + 1 > n_train) n_train = train->entries[i].id + 1;
  for (uint64_t i = 0; i < test->row_ptr[test->n_rows]; i++) if (test->entries[i].id + 1 > n_test) n_test = test->entries[i].id + 1;
  fmo_entry *et_tr, *et_te; uint64_t *cp_tr, *cp_te;
  fmo_transpose(train, n_train, &et_tr, &cp_tr);
  fmo_transpose(test, n_test, &et_te, &cp_te);
  fmo_data_t dtr = { et_tr, cp_tr, n_train, train->n_rows };
  fmo_data_t dte = { et_te, cp_te, n_test, test->n_rows };
  fmo_eq *cache = (fmo_eq *)malloc(sizeof(fmo_eq) * (train->n_rows ? train->n_rows : 1));
  fmo_eq *cache_test = (fmo_eq *)malloc(sizeof(fmo_eq) * (test->n_rows ? test->n_rows : 1));

  fmo_als_predict_eterms(m, &dtr, cache);                  /* _learn :69 */
  fmo_als_predict_eterms(m, &dte, cache_test);
  for (uint32_t c = 0; c < train->n_rows; c++) cache[c].e = cache[c].e - train->target[c];   /* :70-86 */

  for (int it = 0; it < num_iter; it++) {                  /* :88 */
    fmo_als_sweep_groups(m, &dtr, cache, reg);             /* draw_all :94 */
    fmo_als_predict_eterms(m, &dtr, cache);                /* :122 */
    fmo_als_predict_eterms(m, &dte, cache_test);
    if (task == FMO_TASK_REGRESSION) {                     /* :127-150 */
      if (test_pred_this) for (uint32_t c = 0; c < test->n_rows; c++) test_pred_this[c] = cache_test[c].e;
      double rmse = 0;
      for (uint32_t c = 0; c < train->n_rows; c++) {
        double p = cache[c].e;
        p = (max_target < p) ? max_target : p;
        p = (min_target > p) ? min_target : p;
        double err = p - train->target[c];
        rmse += err * err;
        cache[c].e = cache[c].e - train->target[c];
      }
      if (train_metric) train_metric[it] = sqrt(rmse / train->n_rows);
    } else {                                               /* :151-196 */
      if (test_pred_this) for (uint32_t c = 0; c < test->n_rows; c++) test_pred_this[c] = fmo_cdf_gaussian(cache_test[c].e);
      uint32_t acc = 0;
      for (uint32_t c = 0; c < train->n_rows; c++) {
        double p = fmo_cdf_gaussian(cache[c].e);
        if (((p >= 0.5) && (train->target[c] > 0.0)) || ((p < 0.5) && (train->target[c] < 0.0))) acc++;
        double mu = cache[c].e, sampled_target;
        double phi_minus_mu = exp(-mu * mu / 2.0) / sqrt(3.141 * 2);
        double Phi_minus_mu = fmo_cdf_gaussian(-mu);
        if (train->target[c] >= 0.0) sampled_target = mu + phi_minus_mu / (1 - Phi_minus_mu);
        else                         sampled_target = mu - phi_minus_mu / Phi_minus_mu;
        cache[c].e = cache[c].e - sampled_target;
      }
      if (train_metric) train_metric[it] = (double)acc / train->n_rows;
    }
  }
  free(cache); free(cache_test); free(et_tr); free(cp_tr); free(et_te); free(cp_te);
}

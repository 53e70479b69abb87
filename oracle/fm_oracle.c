/*
 * fm_oracle.c -- CPU restatement of the libFM predict + SGD hot path.
 * TEST INFRASTRUCTURE ONLY (see fm_oracle.h).  Every function cites the reference lines it follows.
 * Build: oracle/Makefile  ->  oracle/libfm_oracle.so
 */
#include "fm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define V(m, f, j) ((m)->v[(size_t)(f) * (size_t)(m)->n + (size_t)(j)])

/* fm_model::predict(x, sum, sum_sqr)                      /root/reference/src/fm_core/fm_model.h:105-127
 * Same loop nest and summation order: bias, linear terms in row order, then f outer / i inner. */
double fmo_predict_row(const fmo_model *m, const fmo_entry *row, uint32_t size,
                       double *sum, double *sum_sqr) {
  double result = 0;
  if (m->k0) result += m->w0;                                   /* :107-109 */
  if (m->k1) {
    for (uint32_t i = 0; i < size; i++)                         /* :110-115 */
      result += m->w[row[i].id] * row[i].value;
  }
  for (int f = 0; f < m->k; f++) {                              /* :116-126 */
    sum[f] = 0;
    sum_sqr[f] = 0;
    for (uint32_t i = 0; i < size; i++) {
      double d = V(m, f, row[i].id) * row[i].value;
      sum[f] += d;
      sum_sqr[f] += d * d;
    }
    result += 0.5 * (sum[f] * sum[f] - sum_sqr[f]);
  }
  return result;
}

/* fm_SGD                                                   /root/reference/src/fm_core/fm_sgd.h:33-51
 * Update order: w0, then every w in row order, then V f-major; sum[f] is NOT refreshed. */
void fmo_sgd_step(fmo_model *m, double learn_rate, const fmo_entry *row, uint32_t size,
                  double multiplier, const double *sum) {
  if (m->k0) {
    m->w0 -= learn_rate * (multiplier + m->reg0 * m->w0);       /* :34-37 */
  }
  if (m->k1) {
    for (uint32_t i = 0; i < size; i++) {                       /* :38-43 */
      double *w = &m->w[row[i].id];
      *w -= learn_rate * (multiplier * row[i].value + m->regw * (*w));
    }
  }
  for (int f = 0; f < m->k; f++) {                              /* :44-50 */
    for (uint32_t i = 0; i < size; i++) {
      double *v = &V(m, f, row[i].id);
      double grad = sum[f] * row[i].value - (*v) * row[i].value * row[i].value;
      *v -= learn_rate * (multiplier * grad + m->regv * (*v));
    }
  }
}

/* loss multiplier                   /root/reference/src/libfm/src/fm_learn_sgd_element.h:58-65 */
double fmo_multiplier(int task, double p, double y, double min_target, double max_target) {
  double mult = 0;
  if (task == FMO_TASK_REGRESSION) {
    p = (max_target < p) ? max_target : p;                      /* std::min(max_target, p) */
    p = (min_target > p) ? min_target : p;                      /* std::max(min_target, p) */
    mult = -(y - p);
  } else if (task == FMO_TASK_CLASSIFICATION) {
    mult = -y * (1.0 - 1.0 / (1.0 + exp(-y * p)));
  }
  return mult;
}

/* one pass of the row loop             /root/reference/src/libfm/src/fm_learn_sgd_element.h:56-67 */
void fmo_sgd_epoch_online(fmo_model *m, const fmo_data *d, int task, double learn_rate,
                          double min_target, double max_target) {
  double *sum = (double *)malloc(sizeof(double) * (size_t)(m->k > 0 ? m->k : 1));
  double *sum_sqr = (double *)malloc(sizeof(double) * (size_t)(m->k > 0 ? m->k : 1));
  for (uint32_t r = 0; r < d->n_rows; r++) {
    const fmo_entry *row = d->entries + d->row_ptr[r];
    uint32_t size = (uint32_t)(d->row_ptr[r + 1] - d->row_ptr[r]);
    double p = fmo_predict_row(m, row, size, sum, sum_sqr);
    double mult = fmo_multiplier(task, p, (double)d->target[r], min_target, max_target);
    fmo_sgd_step(m, learn_rate, row, size, mult, sum);
  }
  free(sum);
  free(sum_sqr);
}

/* predict_case in a loop                      /root/reference/src/libfm/src/fm_learn.h:63-65 */
void fmo_predict_raw(const fmo_model *m, const fmo_data *d, double *out) {
  double *sum = (double *)malloc(sizeof(double) * (size_t)(m->k > 0 ? m->k : 1));
  double *sum_sqr = (double *)malloc(sizeof(double) * (size_t)(m->k > 0 ? m->k : 1));
  for (uint32_t r = 0; r < d->n_rows; r++) {
    const fmo_entry *row = d->entries + d->row_ptr[r];
    uint32_t size = (uint32_t)(d->row_ptr[r + 1] - d->row_ptr[r]);
    out[r] = fmo_predict_row(m, row, size, sum, sum_sqr);
  }
  free(sum);
  free(sum_sqr);
}

/* fm_learn_sgd::predict                    /root/reference/src/libfm/src/fm_learn_sgd.h:76-90 */
void fmo_predict_out(const fmo_model *m, const fmo_data *d, int task,
                     double min_target, double max_target, double *out) {
  fmo_predict_raw(m, d, out);
  for (uint32_t r = 0; r < d->n_rows; r++) {
    double p = out[r];
    if (task == FMO_TASK_REGRESSION) {
      p = (max_target < p) ? max_target : p;
      p = (min_target > p) ? min_target : p;
    } else {
      p = 1.0 / (1.0 + exp(-p));
    }
    out[r] = p;
  }
}

/* fm_learn::evaluate_classification / evaluate_regression   /root/reference/src/libfm/src/fm_learn.h:113-153 */
double fmo_evaluate(const fmo_model *m, const fmo_data *d, int task,
                    double min_target, double max_target, double *mae) {
  double *sum = (double *)malloc(sizeof(double) * (size_t)(m->k > 0 ? m->k : 1));
  double *sum_sqr = (double *)malloc(sizeof(double) * (size_t)(m->k > 0 ? m->k : 1));
  double result;
  if (task == FMO_TASK_CLASSIFICATION) {
    int num_correct = 0;                                       /* :113-130 */
    for (uint32_t r = 0; r < d->n_rows; r++) {
      const fmo_entry *row = d->entries + d->row_ptr[r];
      uint32_t size = (uint32_t)(d->row_ptr[r + 1] - d->row_ptr[r]);
      double p = fmo_predict_row(m, row, size, sum, sum_sqr);
      double y = (double)d->target[r];
      if (((p >= 0) && (y >= 0)) || ((p < 0) && (y < 0))) num_correct++;
    }
    if (mae) *mae = NAN;
    result = (double)num_correct / (double)d->n_rows;
  } else {
    double rmse_sum_sqr = 0, mae_sum_abs = 0;                  /* :132-153 */
    for (uint32_t r = 0; r < d->n_rows; r++) {
      const fmo_entry *row = d->entries + d->row_ptr[r];
      uint32_t size = (uint32_t)(d->row_ptr[r + 1] - d->row_ptr[r]);
      double p = fmo_predict_row(m, row, size, sum, sum_sqr);
      p = (max_target < p) ? max_target : p;
      p = (min_target > p) ? min_target : p;
      double err = p - (double)d->target[r];
      rmse_sum_sqr += err * err;
      mae_sum_abs += fabs(err);
    }
    if (mae) *mae = mae_sum_abs / d->n_rows;
    result = sqrt(rmse_sum_sqr / d->n_rows);
  }
  free(sum);
  free(sum_sqr);
  return result;
}

/* The minibatch restatement (see fm_oracle.h).  Derived from fm_sgd.h:33-51 applied per occurrence
 * with batch-start parameters; reduces to fmo_sgd_epoch_online when batch == w0_chunk == 1. */
void fmo_sgd_epoch_minibatch(fmo_model *m, const fmo_data *d, int task, double learn_rate,
                             double min_target, double max_target,
                             uint32_t batch, uint32_t w0_chunk) {
  fmo_sgd_epoch_minibatch_ex(m, d, task, learn_rate, min_target, max_target, batch, w0_chunk, 0);
}

static void minibatch_impl(fmo_model *m, const fmo_data *d, int task, double learn_rate, double min_target, double max_target,
                           uint32_t batch, uint32_t w0_chunk, int bias_lag, int stale, const uint8_t *hot);

void fmo_sgd_epoch_minibatch_ex(fmo_model *m, const fmo_data *d, int task, double learn_rate,
                                double min_target, double max_target,
                                uint32_t batch, uint32_t w0_chunk, int bias_lag) {
  minibatch_impl(m, d, task, learn_rate, min_target, max_target, batch, w0_chunk, bias_lag, 0, NULL);
}

/* the same rule with a HOT set (hot[j] != 0): the linear weights of hot features advance WITH the bias in the micro-chunk
 * recurrence of step 2 (same chunks, same multipliers me), and everybody else sees them with the bias's lag.  Why: the
 * batch rule freezes a parameter for `batch` examples; that is harmless for a feature met once or twice per batch, but the
 * linear weights of frequent features (a 100-id field, the head of a Zipf field) together span the same stiff direction as
 * the bias -- a step of learn_rate * sum over the batch of [same id in some field] -- and diverge like the bias would
 * (tests: the Criteo-shaped set of tests/datagen.py diverges at batch 1024 without a hot set, follows the online loop with
 * one).  Still fm_sgd.h:38-43 per occurrence; batch = w0_chunk = 1 is the reference loop whatever the hot set. */
void fmo_sgd_epoch_minibatch_hot(fmo_model *m, const fmo_data *d, int task, double learn_rate,
                                 double min_target, double max_target,
                                 uint32_t batch, uint32_t w0_chunk, int bias_lag, const uint8_t *hot) {
  minibatch_impl(m, d, task, learn_rate, min_target, max_target, batch, w0_chunk, bias_lag, 0, hot);
}

/* the pipelined multi-GPU schedule (libfm_amd/distributed.py, pipeline=True): the sums of batch b+1 are gathered
 * BEFORE the update of batch b is applied (their all-reduce overlaps that update), so step 1 of batch b reads the
 * parameters as they were before the update of batch b-1 was applied ("one batch stale"); steps 2 and 3 are
 * unchanged (current w0, current w / v in the gradient and the regulariser).  Stale by one only inside an epoch:
 * the epoch drains, the next one starts from the final parameters. */
void fmo_sgd_epoch_minibatch_pipelined(fmo_model *m, const fmo_data *d, int task, double learn_rate,
                                       double min_target, double max_target,
                                       uint32_t batch, uint32_t w0_chunk, int bias_lag) {
  minibatch_impl(m, d, task, learn_rate, min_target, max_target, batch, w0_chunk, bias_lag, 1, NULL);
}

/* the two-level rule (fm_oracle.h): fm_sgd.h:33-51 per occurrence; hot features frozen per window, cold ones per batch */
void fmo_sgd_epoch_twolevel(fmo_model *m, const fmo_data *d, int task, double learn_rate,
                            double min_target, double max_target,
                            uint32_t batch, uint32_t window, uint32_t w0_chunk, int bias_lag, const uint8_t *hot) {
  const int k = m->k;
  const size_t n = (size_t)m->n;
  const size_t kk = (size_t)(k > 0 ? k : 1);
  if (batch == 0 || batch > d->n_rows) batch = d->n_rows;
  if (window == 0 || window > batch) window = batch;
  if (w0_chunk == 0 || w0_chunk > window) w0_chunk = window;
  if (bias_lag > 8) bias_lag = 8;
  double *S = (double *)malloc(sizeof(double) * (size_t)window * kk);
  double *rest = (double *)malloc(sizeof(double) * window);
  double *mult = (double *)malloc(sizeof(double) * window);
  /* accumulated changes: cold features until the batch ends, hot ones until the window ends (separate arrays so that a window's end
   * touches the hot features only; `touched` lists keep both applications proportional to the entries) */
  double *dw = (double *)calloc(n ? n : 1, sizeof(double));
  double *dv = (double *)calloc((n * kk) > 0 ? n * kk : 1, sizeof(double));
  uint32_t *t_hot = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)window * 64 + 1)), n_thot = 0, cap_hot = window * 64u;
  uint8_t *mark = (uint8_t *)calloc(n ? n : 1, 1);               /* 1: in t_hot, 2: in t_cold */
  size_t cap_cold = 1024, n_tcold = 0;
  uint32_t *t_cold = (uint32_t *)malloc(sizeof(uint32_t) * cap_cold);
  double w0_hist[8];
  uint32_t wno = 0;
  for (uint32_t r0 = 0; r0 < d->n_rows; r0 += batch) {
    const uint32_t nb = (d->n_rows - r0 < batch) ? (d->n_rows - r0) : batch;
    for (uint32_t q0 = 0; q0 < nb; q0 += window) {
      const uint32_t nw = (nb - q0 < window) ? (nb - q0) : window;
      /* sums: cold rows have not moved since the batch started, hot rows since the window started -- the model as it stands */
      for (uint32_t e = 0; e < nw; e++) {
        const fmo_entry *row = d->entries + d->row_ptr[r0 + q0 + e];
        const uint32_t size = (uint32_t)(d->row_ptr[r0 + q0 + e + 1] - d->row_ptr[r0 + q0 + e]);
        double res = 0;
        if (m->k1) for (uint32_t i = 0; i < size; i++) res += m->w[row[i].id] * row[i].value;
        for (int f = 0; f < k; f++) {
          double s = 0, q = 0;
          for (uint32_t i = 0; i < size; i++) { const double dd = V(m, f, row[i].id) * row[i].value; s += dd; q += dd * dd; }
          S[(size_t)e * kk + f] = s;
          res += 0.5 * (s * s - q);
        }
        rest[e] = res;
      }
      /* bias: micro-chunks inside the window; the multipliers of the updates see it lagged by bias_lag windows */
      w0_hist[wno % 8] = m->k0 ? m->w0 : 0.0;
      const uint32_t lag_w = (bias_lag > 0 && wno + 1 >= (uint32_t)bias_lag) ? wno + 1 - (uint32_t)bias_lag : 0;
      const double w0_lag = w0_hist[lag_w % 8];
      wno++;
      for (uint32_t c0 = 0; c0 < nw; c0 += w0_chunk) {
        const uint32_t nc = (nw - c0 < w0_chunk) ? (nw - c0) : w0_chunk;
        const double w0s = m->k0 ? m->w0 : 0.0;
        double acc = 0;
        for (uint32_t e = c0; e < c0 + nc; e++) {
          const double y = (double)d->target[r0 + q0 + e];
          const double me = fmo_multiplier(task, w0s + rest[e], y, min_target, max_target);
          mult[e] = bias_lag ? fmo_multiplier(task, w0_lag + rest[e], y, min_target, max_target) : me;
          acc += me + m->reg0 * w0s;
          if (nc == 1 && m->k0) m->w0 -= learn_rate * (me + m->reg0 * m->w0);
        }
        if (m->k0 && nc != 1) m->w0 -= learn_rate * acc;
      }
      /* per-occurrence changes from the values the sums used (fm_sgd.h:38-50) */
      for (uint32_t e = 0; e < nw; e++) {
        const fmo_entry *row = d->entries + d->row_ptr[r0 + q0 + e];
        const uint32_t size = (uint32_t)(d->row_ptr[r0 + q0 + e + 1] - d->row_ptr[r0 + q0 + e]);
        for (uint32_t i = 0; i < size; i++) {
          const uint32_t j = row[i].id;
          const double x = row[i].value;
          if (!mark[j]) {
            if (hot && hot[j]) {
              if (n_thot == cap_hot) { cap_hot *= 2; t_hot = (uint32_t *)realloc(t_hot, sizeof(uint32_t) * ((size_t)cap_hot + 1)); }
              mark[j] = 1; t_hot[n_thot++] = j;
            } else {
              if (n_tcold == cap_cold) { cap_cold *= 2; t_cold = (uint32_t *)realloc(t_cold, sizeof(uint32_t) * cap_cold); }
              mark[j] = 2; t_cold[n_tcold++] = j;
            }
          }
          if (m->k1) dw[j] += -(learn_rate * (mult[e] * x + m->regw * m->w[j]));
          for (int f = 0; f < k; f++) {
            const double v = V(m, f, j);
            dv[(size_t)f * n + j] += -(learn_rate * (mult[e] * (S[(size_t)e * kk + f] * x - v * x * x) + m->regv * v));
          }
        }
      }
      /* the window ends: its hot features move */
      for (uint32_t t = 0; t < n_thot; t++) {
        const uint32_t j = t_hot[t];
        if (m->k1) { m->w[j] += dw[j]; dw[j] = 0.0; }
        for (int f = 0; f < k; f++) { V(m, f, j) += dv[(size_t)f * n + j]; dv[(size_t)f * n + j] = 0.0; }
        mark[j] = 0;
      }
      n_thot = 0;
    }
    /* the batch ends: its cold features move */
    for (size_t t = 0; t < n_tcold; t++) {
      const uint32_t j = t_cold[t];
      if (m->k1) { m->w[j] += dw[j]; dw[j] = 0.0; }
      for (int f = 0; f < k; f++) { V(m, f, j) += dv[(size_t)f * n + j]; dv[(size_t)f * n + j] = 0.0; }
      mark[j] = 0;
    }
    n_tcold = 0;
  }
  free(S); free(rest); free(mult); free(dw); free(dv); free(t_hot); free(t_cold); free(mark);
}

static void minibatch_impl(fmo_model *m, const fmo_data *d, int task, double learn_rate, double min_target, double max_target,
                           uint32_t batch, uint32_t w0_chunk, int bias_lag, int stale, const uint8_t *hot) {
  const int k = m->k;
  const size_t n = (size_t)m->n;
  if (!m->k1) hot = NULL;                                      /* no linear term: nothing to advance */
  /* hot set: slot of every hot feature, its weight at the start of the last 8 batches (what a lagged reader sees), and
   * the chunk's gradient sums */
  uint32_t nH = 0;
  int64_t *hslot = NULL; uint32_t *hfeat = NULL; double *whist = NULL, *hacc = NULL; uint32_t *hcnt = NULL;
  if (hot) {
    hslot = (int64_t *)malloc(sizeof(int64_t) * (n ? n : 1));
    for (size_t j = 0; j < n; j++) hslot[j] = hot[j] ? (int64_t)nH++ : -1;
    hfeat = (uint32_t *)malloc(sizeof(uint32_t) * (nH ? nH : 1));
    for (size_t j = 0; j < n; j++) if (hslot[j] >= 0) hfeat[hslot[j]] = (uint32_t)j;
    whist = (double *)malloc(sizeof(double) * 8 * (nH ? nH : 1));
    hacc = (double *)calloc(nH ? nH : 1, sizeof(double));
    hcnt = (uint32_t *)calloc(nH ? nH : 1, sizeof(uint32_t));
  }
  if (batch == 0 || batch > d->n_rows) batch = d->n_rows;
  if (w0_chunk == 0 || w0_chunk > batch) w0_chunk = batch;
  const size_t kk = (size_t)(k > 0 ? k : 1);
  double *S = (double *)malloc(sizeof(double) * (size_t)batch * kk);
  double *rest = (double *)malloc(sizeof(double) * batch);
  double *mult = (double *)malloc(sizeof(double) * batch);
  double *sum_sqr = (double *)malloc(sizeof(double) * kk);
  double *dw = (double *)calloc(n, sizeof(double));
  double *dv = (double *)calloc(n * kk, sizeof(double));
  /* stale: the parameters before the latest update (what the early gather of the next batch sees) */
  double *w_prev = stale ? (double *)malloc(sizeof(double) * (n ? n : 1)) : NULL;
  double *v_prev = stale ? (double *)malloc(sizeof(double) * ((n * kk) > 0 ? n * kk : 1)) : NULL;
  /* bias_lag = d >= 1: the multipliers handed to step 3 of batch b use the bias as it was BEFORE the recurrence of batch
   * b - d + 1, i.e. after the recurrence of batch b - d (d = 1: the bias of the batch start).  w0_hist[i % 8] = bias at
   * the start of batch i. */
  double w0_hist[8];
  uint32_t bno = 0;
  if (bias_lag > 8) bias_lag = 8;

  for (uint32_t r0 = 0; r0 < d->n_rows; r0 += batch) {
    uint32_t nb = (d->n_rows - r0 < batch) ? (d->n_rows - r0) : batch;
    const double *w_src = (stale && r0 > 0) ? w_prev : m->w;
    const double *v_src = (stale && r0 > 0) ? v_prev : m->v;
    /* which snapshot a lagged reader of batch bno sees (bias and hot linear weights alike) */
    if (hot) for (uint32_t s = 0; s < nH; s++) whist[(size_t)(bno % 8) * nH + s] = m->w[hfeat[s]];
    const uint32_t lag_snap = (bias_lag > 0 && bno + 1 >= (uint32_t)bias_lag) ? bno + 1 - (uint32_t)bias_lag : (bias_lag > 0 ? 0 : bno);
    const double *w_lag = hot ? whist + (size_t)(lag_snap % 8) * nH : NULL;
    /* step 1: sums from batch-start parameters (fm_model.h:110-126 without the bias); hot linear weights: lagged snapshot */
    for (uint32_t e = 0; e < nb; e++) {
      const fmo_entry *row = d->entries + d->row_ptr[r0 + e];
      uint32_t size = (uint32_t)(d->row_ptr[r0 + e + 1] - d->row_ptr[r0 + e]);
      double res = 0;
      if (m->k1)
        for (uint32_t i = 0; i < size; i++)
          res += ((hot && hslot[row[i].id] >= 0) ? w_lag[hslot[row[i].id]] : w_src[row[i].id]) * row[i].value;
      for (int f = 0; f < k; f++) {
        double s = 0, q = 0;
        for (uint32_t i = 0; i < size; i++) {
          double dd = v_src[(size_t)f * n + row[i].id] * row[i].value;
          s += dd;
          q += dd * dd;
        }
        S[(size_t)e * kk + f] = s;
        sum_sqr[f] = q;
        res += 0.5 * (s * s - q);
      }
      rest[e] = res;
    }
    /* step 2: w0 micro-chunks (fm_learn_sgd_element.h:57-65 + fm_sgd.h:34-37).
     * bias_lag: the multipliers handed to step 3 use the w0 of the BATCH start (frozen), while w0 itself still
     * advances through the micro-chunks with its own multipliers -- this takes the serial recurrence off the
     * critical path of the step (it overlaps the next batch's gather).  Identical to the plain rule at batch 1. */
    w0_hist[bno % 8] = m->k0 ? m->w0 : 0.0;
    const uint32_t lag_b = (bias_lag > 0 && bno + 1 >= (uint32_t)bias_lag) ? bno + 1 - (uint32_t)bias_lag : 0;
    const double w0_batch = w0_hist[lag_b % 8];
    bno++;
    for (uint32_t c0 = 0; c0 < nb; c0 += w0_chunk) {
      uint32_t nc = (nb - c0 < w0_chunk) ? (nb - c0) : w0_chunk;
      double w0s = m->k0 ? m->w0 : 0.0;
      double acc = 0;
      for (uint32_t e = c0; e < c0 + nc; e++) {
        double p = w0s + rest[e];
        if (hot) {                                             /* the hot weights as they are at the chunk start, not the snapshot */
          const fmo_entry *row = d->entries + d->row_ptr[r0 + e];
          uint32_t size = (uint32_t)(d->row_ptr[r0 + e + 1] - d->row_ptr[r0 + e]);
          for (uint32_t i = 0; i < size; i++) {
            const int64_t s = hslot[row[i].id];
            if (s >= 0) p += (m->w[row[i].id] - w_lag[s]) * row[i].value;
          }
        }
        double me = fmo_multiplier(task, p, (double)d->target[r0 + e], min_target, max_target);
        mult[e] = bias_lag ? fmo_multiplier(task, w0_batch + rest[e], (double)d->target[r0 + e], min_target, max_target) : me;
        acc += me + m->reg0 * w0s;
        if (hot) {
          const fmo_entry *row = d->entries + d->row_ptr[r0 + e];
          uint32_t size = (uint32_t)(d->row_ptr[r0 + e + 1] - d->row_ptr[r0 + e]);
          for (uint32_t i = 0; i < size; i++) {
            const int64_t s = hslot[row[i].id];
            if (s >= 0) { hacc[s] += me * row[i].value; hcnt[s]++; }
          }
        }
        if (nc == 1 && m->k0) m->w0 -= learn_rate * (me + m->reg0 * m->w0);  /* literal form at chunk 1 */
      }
      if (m->k0 && nc != 1) m->w0 -= learn_rate * acc;
      if (hot) {                                               /* fm_sgd.h:38-43 per occurrence, from the chunk-start weight */
        for (uint32_t e = c0; e < c0 + nc; e++) {
          const fmo_entry *row = d->entries + d->row_ptr[r0 + e];
          uint32_t size = (uint32_t)(d->row_ptr[r0 + e + 1] - d->row_ptr[r0 + e]);
          for (uint32_t i = 0; i < size; i++) {
            const int64_t s = hslot[row[i].id];
            if (s >= 0 && hcnt[s]) {
              double *w = &m->w[row[i].id];
              *w -= learn_rate * (hacc[s] + (double)hcnt[s] * m->regw * (*w));
              hacc[s] = 0.0; hcnt[s] = 0;
            }
          }
        }
      }
    }
    if (stale) {                                               /* what batch b+1's early gather will see */
      memcpy(w_prev, m->w, sizeof(double) * n);
      memcpy(v_prev, m->v, sizeof(double) * n * (size_t)k);
    }
    /* step 3: per-occurrence deltas from batch-start w, v (fm_sgd.h:38-50) */
    for (uint32_t e = 0; e < nb; e++) {
      const fmo_entry *row = d->entries + d->row_ptr[r0 + e];
      uint32_t size = (uint32_t)(d->row_ptr[r0 + e + 1] - d->row_ptr[r0 + e]);
      if (m->k1)
        for (uint32_t i = 0; i < size; i++) {
          if (hot && hslot[row[i].id] >= 0) continue;          /* advanced by the recurrence of step 2 */
          double w = m->w[row[i].id];
          dw[row[i].id] += -(learn_rate * (mult[e] * row[i].value + m->regw * w));
        }
      for (int f = 0; f < k; f++)
        for (uint32_t i = 0; i < size; i++) {
          double v = V(m, f, row[i].id);
          double x = row[i].value;
          double grad = S[(size_t)e * kk + f] * x - v * x * x;
          dv[(size_t)f * n + row[i].id] += -(learn_rate * (mult[e] * grad + m->regv * v));
        }
    }
    /* apply and clear the touched deltas */
    for (uint32_t e = 0; e < nb; e++) {
      const fmo_entry *row = d->entries + d->row_ptr[r0 + e];
      uint32_t size = (uint32_t)(d->row_ptr[r0 + e + 1] - d->row_ptr[r0 + e]);
      for (uint32_t i = 0; i < size; i++) {
        size_t j = row[i].id;
        if (m->k1 && dw[j] != 0.0) { m->w[j] += dw[j]; dw[j] = 0.0; }
        for (int f = 0; f < k; f++) {
          double *p = &dv[(size_t)f * n + j];
          if (*p != 0.0) { V(m, f, j) += *p; *p = 0.0; }
        }
      }
    }
  }
  free(S); free(rest); free(mult); free(sum_sqr); free(dw); free(dv); free(w_prev); free(v_prev);
  free(hslot); free(hfeat); free(whist); free(hacc); free(hcnt);
}

/* ------------------------------- SGDA ------------------------------- */
#define GRP(st, id) ((st)->group ? (st)->group[(id)] : 0u)      /* meta->attr_group(id), Data.h:41 */

/* sgd_theta_step                    /root/reference/src/libfm/src/fm_learn_sgd_element_adapt_reg.h:136-169 */
static void sgda_theta_step(fmo_model *m, fmo_sgda_state *st, const fmo_entry *row, uint32_t size, double target,
                            int task, double lr, double min_target, double max_target, double *sum, double *sum_sqr) {
  double p = fmo_predict_row(m, row, size, sum, sum_sqr);
  double mult = 0;
  if (task == 0) {
    p = (max_target < p) ? max_target : p;
    p = (min_target > p) ? min_target : p;
    mult = 2 * (p - target);                                   /* :142 (note the factor 2, unlike plain SGD) */
  } else if (task == 1) {
    mult = target * ((1.0 / (1.0 + exp(-target * p))) - 1.0);  /* :144 */
  }
  const size_t n = (size_t)m->n;
  const int k = m->k;
  if (m->k0) m->w0 -= lr * (mult + 2 * 0.0 * m->w0);           /* reg_0 = 0 (:100,:149) */
  if (m->k1)
    for (uint32_t i = 0; i < size; i++) {
      uint32_t g = GRP(st, row[i].id);
      double *w = &m->w[row[i].id];
      st->grad_w[row[i].id] = mult * row[i].value;
      *w -= lr * (st->grad_w[row[i].id] + 2 * st->reg_w[g] * (*w));
    }
  for (int f = 0; f < k; f++)
    for (uint32_t i = 0; i < size; i++) {
      uint32_t g = GRP(st, row[i].id);
      double *v = &V(m, f, row[i].id);
      double *gr = &st->grad_v[(size_t)f * n + row[i].id];
      *gr = mult * (row[i].value * (sum[f] - (*v) * row[i].value));
      *v -= lr * (*gr + 2 * st->reg_v[(size_t)g * k + f] * (*v));
    }
}

/* predict_scaled (:171-199) + sgd_lambda_step (:201-248) */
static void sgda_lambda_step(fmo_model *m, fmo_sgda_state *st, const fmo_entry *row, uint32_t size, double target,
                             int task, double lr, double min_target, double max_target) {
  const size_t n = (size_t)m->n;
  const int k = m->k;
  const uint32_t G = st->num_groups;
  double p = 0.0;
  if (m->k0) p += m->w0;
  if (m->k1)
    for (uint32_t i = 0; i < size; i++) {
      uint32_t g = GRP(st, row[i].id);
      double w = m->w[row[i].id];
      double w_dash = w - lr * (st->grad_w[row[i].id] + 2 * st->reg_w[g] * w);
      p += w_dash * row[i].value;
    }
  for (int f = 0; f < k; f++) {
    double s = 0.0, q = 0.0;
    for (uint32_t i = 0; i < size; i++) {
      uint32_t g = GRP(st, row[i].id);
      double v = V(m, f, row[i].id);
      double v_dash = v - lr * (st->grad_v[(size_t)f * n + row[i].id] + 2 * st->reg_v[(size_t)g * k + f] * v);
      double d = v_dash * row[i].value;
      s += d; q += d * d;
    }
    p += 0.5 * (s * s - q);
  }
  double grad_loss = 0;
  if (task == 0) {
    p = (max_target < p) ? max_target : p;
    p = (min_target > p) ? min_target : p;
    grad_loss = 2 * (p - target);
  } else if (task == 1) {
    grad_loss = target * ((1.0 / (1.0 + exp(-target * p))) - 1.0);
  }
  double *acc = (double *)calloc((size_t)2 * G, sizeof(double));   /* lambda_w_grad / sum_f, sum_f_dash_f (:96-98) */
  if (m->k1) {                                                 /* :213-224 */
    for (uint32_t i = 0; i < size; i++) acc[GRP(st, row[i].id)] += row[i].value * m->w[row[i].id];
    for (uint32_t g = 0; g < G; g++) {
      double lw = -2 * lr * acc[g];
      st->reg_w[g] -= lr * grad_loss * lw;
      st->reg_w[g] = (0.0 > st->reg_w[g]) ? 0.0 : st->reg_w[g];
    }
  }
  for (int f = 0; f < k; f++) {                                /* :225-247 */
    double sum_f_dash = 0.0;
    double *sum_f = acc, *sum_f_dash_f = acc + G;
    for (uint32_t g = 0; g < 2 * G; g++) acc[g] = 0.0;
    for (uint32_t i = 0; i < size; i++) {
      uint32_t g = GRP(st, row[i].id);
      double v = V(m, f, row[i].id);
      double v_dash = v - lr * (st->grad_v[(size_t)f * n + row[i].id] + 2 * st->reg_v[(size_t)g * k + f] * v);
      sum_f_dash += v_dash * row[i].value;
      sum_f[g] += v * row[i].value;
      sum_f_dash_f[g] += v_dash * row[i].value * v * row[i].value;
    }
    for (uint32_t g = 0; g < G; g++) {
      double lambda_v_grad = -2 * lr * (sum_f_dash * sum_f[g] - sum_f_dash_f[g]);
      double *rv = &st->reg_v[(size_t)g * k + f];
      *rv -= lr * grad_loss * lambda_v_grad;
      *rv = (0.0 > *rv) ? 0.0 : *rv;
    }
  }
  free(acc);
}

/* The minibatch restatement of SGDA used by the GPU batch form (fmx_sgda_epoch_minibatch), in the spirit of
 * fmo_sgd_epoch_minibatch: per batch of B train rows
 *   theta: steps 1-3 of the batch rule with this learner's multiplier (2 (p - y) for regression, :142), reg_0 = 0 and the
 *          LEARNED regularisation 2 reg(g[,f]) theta per occurrence (:149-166); the shadow gradient of a touched parameter
 *          becomes the SUM of its occurrences' gradients in the batch (one occurrence: the reference's value);
 *   lambda (do_lambda_steps): the next B validation rows (cyclic, :271-274), every one evaluated like sgd_lambda_step
 *          (:201-248 through predict_scaled :171-199) with the parameters and shadows the theta step just left and the
 *          regularisation values of the batch START; their reg changes are summed and applied -- clamped at 0 -- once.
 * batch == 1 and w0_chunk == 1: the reference loop (rows without a repeated id), see the tests. */
void fmo_sgda_epoch_minibatch(fmo_model *m, fmo_sgda_state *st, const fmo_data *train, const fmo_data *val, int task,
                              double learn_rate, double min_target, double max_target, int do_lambda_steps,
                              uint32_t batch, uint32_t w0_chunk) {
  const int k = m->k;
  const size_t n = (size_t)m->n, kk = (size_t)(k > 0 ? k : 1);
  const uint32_t G = st->num_groups;
  const double lr = learn_rate;
  if (batch == 0 || batch > train->n_rows) batch = train->n_rows;
  if (w0_chunk == 0 || w0_chunk > batch) w0_chunk = batch;
  double *S = (double *)malloc(sizeof(double) * (size_t)batch * kk);
  double *rest = (double *)malloc(sizeof(double) * batch);
  double *mult = (double *)malloc(sizeof(double) * batch);
  double *dw = (double *)calloc(n, sizeof(double)), *dv = (double *)calloc(n * kk, sizeof(double));
  double *gw = (double *)calloc(n, sizeof(double)), *gv = (double *)calloc(n * kk, sizeof(double));   /* this batch's shadow sums */
  unsigned char *touched = (unsigned char *)calloc(n ? n : 1, 1);
  double *dreg_w = (double *)malloc(sizeof(double) * G), *dreg_v = (double *)malloc(sizeof(double) * G * kk);
  double *reg_w0 = (double *)malloc(sizeof(double) * G), *reg_v0 = (double *)malloc(sizeof(double) * G * kk);
  double *acc = (double *)malloc(sizeof(double) * 2 * G);
  st->val_pos = 0;
  for (uint32_t r0 = 0; r0 < train->n_rows; r0 += batch) {
    const uint32_t nb = (train->n_rows - r0 < batch) ? (train->n_rows - r0) : batch;
    /* ---- theta, step 1 */
    for (uint32_t e = 0; e < nb; e++) {
      const fmo_entry *row = train->entries + train->row_ptr[r0 + e];
      const uint32_t size = (uint32_t)(train->row_ptr[r0 + e + 1] - train->row_ptr[r0 + e]);
      double res = 0;
      if (m->k1) for (uint32_t i = 0; i < size; i++) res += m->w[row[i].id] * row[i].value;
      for (int f = 0; f < k; f++) {
        double s = 0, q = 0;
        for (uint32_t i = 0; i < size; i++) { double d = V(m, f, row[i].id) * row[i].value; s += d; q += d * d; }
        S[(size_t)e * kk + f] = s;
        res += 0.5 * (s * s - q);
      }
      rest[e] = res;
    }
    /* ---- theta, step 2: bias micro-chunks, reg_0 = 0 */
    for (uint32_t c0 = 0; c0 < nb; c0 += w0_chunk) {
      const uint32_t nc = (nb - c0 < w0_chunk) ? (nb - c0) : w0_chunk;
      const double w0s = m->k0 ? m->w0 : 0.0;
      double a = 0;
      for (uint32_t e = c0; e < c0 + nc; e++) {
        double p = w0s + rest[e];
        const double y = (double)train->target[r0 + e];
        double me;
        if (task == 0) { p = (max_target < p) ? max_target : p; p = (min_target > p) ? min_target : p; me = 2 * (p - y); }
        else me = y * ((1.0 / (1.0 + exp(-y * p))) - 1.0);
        mult[e] = me;
        a += me;
      }
      if (m->k0) m->w0 -= lr * a;
    }
    /* ---- theta, step 3 */
    for (uint32_t e = 0; e < nb; e++) {
      const fmo_entry *row = train->entries + train->row_ptr[r0 + e];
      const uint32_t size = (uint32_t)(train->row_ptr[r0 + e + 1] - train->row_ptr[r0 + e]);
      for (uint32_t i = 0; i < size; i++) {
        const size_t j = row[i].id;
        const uint32_t g = GRP(st, j);
        const double x = row[i].value;
        touched[j] = 1;
        if (m->k1) { const double gr = mult[e] * x; gw[j] += gr; dw[j] += -(lr * (gr + 2 * st->reg_w[g] * m->w[j])); }
        for (int f = 0; f < k; f++) {
          const double v = V(m, f, j);
          const double gr = mult[e] * (x * (S[(size_t)e * kk + f] - v * x));
          gv[(size_t)f * n + j] += gr;
          dv[(size_t)f * n + j] += -(lr * (gr + 2 * st->reg_v[(size_t)g * k + f] * v));
        }
      }
    }
    for (uint32_t e = 0; e < nb; e++) {                           /* apply, move the shadows, clear */
      const fmo_entry *row = train->entries + train->row_ptr[r0 + e];
      const uint32_t size = (uint32_t)(train->row_ptr[r0 + e + 1] - train->row_ptr[r0 + e]);
      for (uint32_t i = 0; i < size; i++) {
        const size_t j = row[i].id;
        if (!touched[j]) continue;
        touched[j] = 0;
        if (m->k1) { m->w[j] += dw[j]; st->grad_w[j] = gw[j]; dw[j] = 0; gw[j] = 0; }
        for (int f = 0; f < k; f++) {
          const size_t c = (size_t)f * n + j;
          V(m, f, j) += dv[c]; st->grad_v[c] = gv[c]; dv[c] = 0; gv[c] = 0;
        }
      }
    }
    if (!do_lambda_steps || val->n_rows == 0) continue;
    /* ---- lambda over the next nb validation rows, regularisation frozen at its batch-start values */
    memcpy(reg_w0, st->reg_w, sizeof(double) * G);
    memcpy(reg_v0, st->reg_v, sizeof(double) * G * (size_t)k);
    for (uint32_t g = 0; g < G; g++) dreg_w[g] = 0;
    for (size_t c = 0; c < (size_t)G * kk; c++) dreg_v[c] = 0;
    for (uint32_t t = 0; t < nb; t++) {
      if (st->val_pos >= val->n_rows) st->val_pos = 0;
      const fmo_entry *row = val->entries + val->row_ptr[st->val_pos];
      const uint32_t size = (uint32_t)(val->row_ptr[st->val_pos + 1] - val->row_ptr[st->val_pos]);
      const double target = (double)val->target[st->val_pos];
      st->val_pos++;
      double p = 0.0;
      if (m->k0) p += m->w0;
      if (m->k1)
        for (uint32_t i = 0; i < size; i++) {
          const uint32_t g = GRP(st, row[i].id);
          const double w = m->w[row[i].id];
          p += (w - lr * (st->grad_w[row[i].id] + 2 * reg_w0[g] * w)) * row[i].value;
        }
      for (int f = 0; f < k; f++) {
        double s = 0.0, q = 0.0;
        for (uint32_t i = 0; i < size; i++) {
          const uint32_t g = GRP(st, row[i].id);
          const double v = V(m, f, row[i].id);
          const double d = (v - lr * (st->grad_v[(size_t)f * n + row[i].id] + 2 * reg_v0[(size_t)g * k + f] * v)) * row[i].value;
          s += d; q += d * d;
        }
        p += 0.5 * (s * s - q);
      }
      double grad_loss;
      if (task == 0) { p = (max_target < p) ? max_target : p; p = (min_target > p) ? min_target : p; grad_loss = 2 * (p - target); }
      else grad_loss = target * ((1.0 / (1.0 + exp(-target * p))) - 1.0);
      if (m->k1) {
        for (uint32_t g = 0; g < G; g++) acc[g] = 0;
        for (uint32_t i = 0; i < size; i++) acc[GRP(st, row[i].id)] += row[i].value * m->w[row[i].id];
        for (uint32_t g = 0; g < G; g++) dreg_w[g] -= lr * grad_loss * (-2 * lr * acc[g]);
      }
      for (int f = 0; f < k; f++) {
        double sum_f_dash = 0.0;
        double *sum_f = acc, *sum_f_dash_f = acc + G;
        for (uint32_t g = 0; g < 2 * G; g++) acc[g] = 0.0;
        for (uint32_t i = 0; i < size; i++) {
          const uint32_t g = GRP(st, row[i].id);
          const double v = V(m, f, row[i].id);
          const double v_dash = v - lr * (st->grad_v[(size_t)f * n + row[i].id] + 2 * reg_v0[(size_t)g * k + f] * v);
          sum_f_dash += v_dash * row[i].value;
          sum_f[g] += v * row[i].value;
          sum_f_dash_f[g] += v_dash * row[i].value * v * row[i].value;
        }
        for (uint32_t g = 0; g < G; g++)
          dreg_v[(size_t)g * k + f] -= lr * grad_loss * (-2 * lr * (sum_f_dash * sum_f[g] - sum_f_dash_f[g]));
      }
    }
    for (uint32_t g = 0; g < G; g++) { const double x = reg_w0[g] + dreg_w[g]; st->reg_w[g] = m->k1 ? ((0.0 > x) ? 0.0 : x) : reg_w0[g]; }
    for (size_t c = 0; c < (size_t)G * (size_t)k; c++) { const double x = reg_v0[c] + dreg_v[c]; st->reg_v[c] = (0.0 > x) ? 0.0 : x; }
  }
  free(S); free(rest); free(mult); free(dw); free(dv); free(gw); free(gv); free(touched);
  free(dreg_w); free(dreg_v); free(reg_w0); free(reg_v0); free(acc);
}

/* one iteration of fm_learn_sgd_element_adapt_reg::learn (:262-279) */
void fmo_sgda_epoch(fmo_model *m, fmo_sgda_state *st, const fmo_data *train, const fmo_data *val, int task,
                    double learn_rate, double min_target, double max_target, int do_lambda_steps) {
  double *sum = (double *)malloc(sizeof(double) * (size_t)(m->k > 0 ? m->k : 1));
  double *sum_sqr = (double *)malloc(sizeof(double) * (size_t)(m->k > 0 ? m->k : 1));
  st->val_pos = 0;                                             /* validation->data->begin() (:266) */
  for (uint32_t r = 0; r < train->n_rows; r++) {
    const fmo_entry *row = train->entries + train->row_ptr[r];
    uint32_t size = (uint32_t)(train->row_ptr[r + 1] - train->row_ptr[r]);
    sgda_theta_step(m, st, row, size, (double)train->target[r], task, learn_rate, min_target, max_target, sum, sum_sqr);
    if (do_lambda_steps) {
      if (st->val_pos >= val->n_rows) st->val_pos = 0;         /* :271-274 */
      const fmo_entry *vrow = val->entries + val->row_ptr[st->val_pos];
      uint32_t vsize = (uint32_t)(val->row_ptr[st->val_pos + 1] - val->row_ptr[st->val_pos]);
      sgda_lambda_step(m, st, vrow, vsize, (double)val->target[st->val_pos], task, learn_rate, min_target, max_target);
      st->val_pos++;
    }
  }
  free(sum); free(sum_sqr);
}

/* ------------------------------- synthetic workload ------------------------------- */

uint64_t fmo_mix64(uint64_t x) {          /* splitmix64 finaliser */
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27; x *= 0x94D049BB133111EBULL;
  x ^= x >> 31;
  return x;
}

static uint64_t synth_key(uint64_t seed, uint64_t row, uint32_t field) {
  return fmo_mix64(seed + 0x9E3779B97F4A7C15ULL * (row + 1) + 0xC2B2AE3D27D4EB4FULL * ((uint64_t)field + 1));
}

uint32_t fmo_synth_id(uint64_t seed, uint64_t row, uint32_t field, uint32_t field_size) {
  uint64_t h = synth_key(seed, row, field);
  uint32_t off = (uint32_t)(((h >> 32) * (uint64_t)field_size) >> 32);   /* multiply-high range map */
  return field * field_size + off;
}

float fmo_synth_target(uint64_t seed, uint64_t row) {
  return (synth_key(seed, row, 0xFFFFFFFFu) & 1) ? 1.0f : -1.0f;
}

void fmo_synth_rows(uint64_t seed, uint64_t row0, uint32_t n_rows, uint32_t nnz, uint64_t n,
                    fmo_entry *entries, uint64_t *row_ptr, float *target) {
  uint32_t fs = (uint32_t)(n / nnz);
  for (uint32_t r = 0; r < n_rows; r++) {
    row_ptr[r] = (uint64_t)r * nnz;
    for (uint32_t t = 0; t < nnz; t++) {
      entries[(size_t)r * nnz + t].id = fmo_synth_id(seed, row0 + r, t, fs);
      entries[(size_t)r * nnz + t].value = 1.0f;
    }
    target[r] = fmo_synth_target(seed, row0 + r);
  }
  row_ptr[n_rows] = (uint64_t)n_rows * nnz;
}

double fmo_init_value(uint64_t seed, uint64_t j, uint32_t f, double stdev) {
  uint64_t h = fmo_mix64(seed ^ (j * 0x9E3779B97F4A7C15ULL + (uint64_t)f * 0xD6E8FEB86659FD93ULL + 0x1234567ULL));
  double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);   /* [0,1) */
  return stdev * (2.0 * u - 1.0) * 1.7320508075688772;
}

typedef struct { fmo_model *m; uint64_t seed; double stdev; int tid, nt; } fill_arg;

static void *fill_worker(void *p) {
  fill_arg *a = (fill_arg *)p;
  const size_t n = (size_t)a->m->n;
  const size_t lo = n * (size_t)a->tid / (size_t)a->nt, hi = n * (size_t)(a->tid + 1) / (size_t)a->nt;
  for (int f = 0; f < a->m->k; f++)
    for (size_t j = lo; j < hi; j++) V(a->m, f, j) = fmo_init_value(a->seed, j, (uint32_t)f, a->stdev);
  for (size_t j = lo; j < hi; j++) a->m->w[j] = 0.0;
  return NULL;
}

void fmo_fill_params(fmo_model *m, uint64_t seed, double stdev, int threads) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pthread_t th[256];
  fill_arg args[256];
  for (int t = 0; t < threads; t++) {
    args[t].m = m; args[t].seed = seed; args[t].stdev = stdev; args[t].tid = t; args[t].nt = threads;
    pthread_create(&th[t], NULL, fill_worker, &args[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  m->w0 = 0.0;
}

double fmo_time_sgd_synth(fmo_model *m, uint64_t seed, uint64_t row0, uint32_t n_rows, uint32_t nnz,
                          int task, double learn_rate) {
  fmo_entry *ent = (fmo_entry *)malloc(sizeof(fmo_entry) * (size_t)n_rows * nnz);
  uint64_t *rp = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)n_rows + 1));
  float *y = (float *)malloc(sizeof(float) * n_rows);
  fmo_synth_rows(seed, row0, n_rows, nnz, m->n, ent, rp, y);
  fmo_data d; d.entries = ent; d.row_ptr = rp; d.target = y; d.n_rows = n_rows;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  fmo_sgd_epoch_online(m, &d, task, learn_rate, -1.0, 1.0);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(ent); free(rp); free(y);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

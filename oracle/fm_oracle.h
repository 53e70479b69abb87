/*
 * fm_oracle.h -- CPU restatement of the libFM hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X HIP path.  It is NOT part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may link or call it.  It restates, in plain C and in the reference's own
 * precision (fp32 data, fp64 parameters and accumulators), the algorithms of
 *
 *   fm_model::predict            /root/reference/src/fm_core/fm_model.h:105-127
 *   fm_SGD                       /root/reference/src/fm_core/fm_sgd.h:33-51
 *   fm_learn_sgd_element::learn  /root/reference/src/libfm/src/fm_learn_sgd_element.h:48-78
 *   fm_learn_sgd::predict        /root/reference/src/libfm/src/fm_learn_sgd.h:76-90
 *   fm_learn::evaluate_*         /root/reference/src/libfm/src/fm_learn.h:113-153
 *   fm_learn_mcmc (ALS part)     /root/reference/src/libfm/src/fm_learn_mcmc.h:148-378,406-428,643-732,792-847
 *
 * Parity pin: the restatement is checked bit-for-bit (fp64) against the real
 * reference compiled from /root/reference (oracle/_ref/ref_harness, built by
 * oracle/Makefile) and against the committed fixtures in tests/golden/ that the
 * harness generated (tests/golden/make_golden.py).
 *
 * Parameter layout is the reference's (fm_model.h:46-48, matrix.h:165-170):
 *   w0 double; w[n] double; v[k][n] double, FACTOR-major (v[f*n + j]), but with
 *   size_t indexing so that k*n >= 2^32 works (the stock reference overflows there).
 * Input layout is the reference's (fmatrix.h:34-42, Data.h:237-270): one contiguous
 *   array of {uint32 id; float value} in row order + row offsets.
 */
#ifndef FM_ORACLE_H_
#define FM_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint32_t id; float value; } fmo_entry;   /* sparse_entry<float>, fmatrix.h:34-37 */

typedef struct {
  uint64_t n;        /* num_attribute                       fm_model.h:51 */
  int32_t  k;        /* num_factor                          fm_model.h:54 */
  int32_t  k0, k1;   /* use bias / use 1-way interactions   fm_model.h:53 */
  double   w0;
  double  *w;        /* [n]                                  */
  double  *v;        /* [k][n] factor-major                  */
  double   reg0, regw, regv;
} fmo_model;

typedef struct {
  const fmo_entry *entries;   /* contiguous, row order */
  const uint64_t  *row_ptr;   /* [n_rows+1]            */
  const float     *target;    /* [n_rows]              */
  uint32_t         n_rows;
} fmo_data;

enum { FMO_TASK_REGRESSION = 0, FMO_TASK_CLASSIFICATION = 1 };  /* fm_learn.h:45-47 */

/* fm_model::predict with side outputs sum[k], sum_sqr[k]  (fm_model.h:105-127) */
double fmo_predict_row(const fmo_model *m, const fmo_entry *row, uint32_t size,
                       double *sum, double *sum_sqr);

/* fm_SGD  (fm_sgd.h:33-51) */
void fmo_sgd_step(fmo_model *m, double learn_rate, const fmo_entry *row, uint32_t size,
                  double multiplier, const double *sum);

/* loss multiplier of fm_learn_sgd_element::learn (fm_learn_sgd_element.h:58-65) */
double fmo_multiplier(int task, double p, double y, double min_target, double max_target);

/* one epoch of the strictly-online loop (fm_learn_sgd_element.h:56-67) */
void fmo_sgd_epoch_online(fmo_model *m, const fmo_data *d, int task, double learn_rate,
                          double min_target, double max_target);

/* raw y-hat for every row (fm_learn.h:63-65 predict_case in a loop) */
void fmo_predict_raw(const fmo_model *m, const fmo_data *d, double *out);

/* fm_learn_sgd::predict: clamp (regression) or sigmoid (classification)  (fm_learn_sgd.h:76-90) */
void fmo_predict_out(const fmo_model *m, const fmo_data *d, int task,
                     double min_target, double max_target, double *out);

/* fm_learn::evaluate_regression -> rmse (and mae), evaluate_classification -> accuracy
 * (fm_learn.h:113-153).  Returns the value evaluate() returns; *mae may be NULL. */
double fmo_evaluate(const fmo_model *m, const fmo_data *d, int task,
                    double min_target, double max_target, double *mae);

/*
 * The minibatch restatement of fm_SGD used by the GPU "minibatch" mode:
 *   1. for every example e of the batch, from the BATCH-START parameters:
 *        S_ef, Q_e, lin_e  (the sums of fm_model.h:111-125);
 *        rest_e = lin_e + 0.5*(sum_f S_ef^2 - Q_e)
 *   2. w0 is advanced in micro-chunks of `w0_chunk` consecutive examples: inside a chunk every
 *      example sees the w0 at chunk start, p_e = w0 + rest_e, mult_e = fmo_multiplier(p_e, y_e),
 *      then w0 -= lr * sum_{e in chunk}(mult_e + reg0*w0)              (fm_sgd.h:34-37 per example)
 *   3. every occurrence (e, j, x) adds, computed from the batch-start w[j], v[f][j]:
 *        dw[j]    += -lr*(mult_e*x + regw*w[j])                          (fm_sgd.h:38-43)
 *        dv[f][j] += -lr*(mult_e*(S_ef*x - v[f][j]*x*x) + regv*v[f][j])  (fm_sgd.h:44-50)
 *      and the sums are applied at batch end.
 * With batch == 1 and w0_chunk == 1 this IS the reference loop for rows without a repeated id, up to
 * the last ulp of p (the rule adds w0 to rest_e last, fm_model.h:107-115 adds it first); the tests
 * hold it to rtol 1e-11 against the real reference.  batch == 0 means "whole data set".
 */
void fmo_sgd_epoch_minibatch(fmo_model *m, const fmo_data *d, int task, double learn_rate,
                             double min_target, double max_target,
                             uint32_t batch, uint32_t w0_chunk);
/* same with bias_lag = d >= 1: step 3 of batch b uses mult_e = multiplier(w0 + rest_e) with the w0 of the START of batch
 * b - d + 1 (d = 1: the start of the batch itself; the first d - 1 batches use the initial bias) while step 2's recurrence
 * still advances w0 chunk by chunk (GPU: FMX_FLAG_BIAS_LAG / fmx_sgd_opts::bias_lag -- the one-workgroup recurrence
 * leaves the critical path and, for d >= 2, hides completely under the next launches). */
void fmo_sgd_epoch_minibatch_ex(fmo_model *m, const fmo_data *d, int task, double learn_rate,
                                double min_target, double max_target,
                                uint32_t batch, uint32_t w0_chunk, int bias_lag);
/* the same with a HOT set (hot[j] != 0, [n] bytes; NULL = none): the linear weights of the hot features advance with the bias
 * in step 2 -- inside a micro-chunk every example sees w0 and the hot w_j of the chunk start, p_e = w0 + rest_e with the hot
 * part of rest_e re-evaluated from them; after the chunk w_j -= lr * sum_{e in chunk, j in e}(mult_e * x + regw * w_j)
 * (fm_sgd.h:38-43 per occurrence) -- and step 3 leaves them alone.  Readers that lag the bias (bias_lag) see the hot weights
 * with the same lag.  Everything else (V of all features, w of the others) is the rule above.  An instrument of DESIGN.md section 3a (it shows why the linear weights alone are not enough); no product mode runs it. */
void fmo_sgd_epoch_minibatch_hot(fmo_model *m, const fmo_data *d, int task, double learn_rate,
                                 double min_target, double max_target,
                                 uint32_t batch, uint32_t w0_chunk, int bias_lag, const uint8_t *hot);
/* The TWO-LEVEL batch rule (round 4; what FMX_APPLY_FUSED runs when the rows' collision mass cuts the batch -- data with frequent
 * features, BASELINE configs[2]).  The batch rule freezes every parameter for one batch, and the stability bound
 *   learn_rate * curvature * batch * C <= 1,   C = sum_j (sum_e |x_ej|)^2 / N^2  (collision mass)
 * is set by a handful of FREQUENT features (a 100-id field, the head of a Zipf field): C = C_hot + C_cold over any split of the
 * features, and the frozen window a parameter tolerates depends on ITS share.  So the frozen window is chosen per class:
 *   - HOT features (hot[j] != 0) are frozen for one WINDOW of `window` rows: read at window-start values, the sum of the window's
 *     occurrences (each computed from the window-start value, fm_sgd.h:38-50) applied at the window's end;
 *   - COLD features are frozen for one BATCH of `batch` rows (a multiple of the window): read at batch-start values, the sum of the
 *     batch's occurrences applied at the batch's end;
 *   - an example's sums (fm_model.h:110-125) take hot rows as they are at ITS window's start and cold rows as they are at its
 *     batch's start; its multiplier uses the bias lagged by `bias_lag` WINDOWS (>= 1: the bias at the start of window
 *     w - bias_lag + 1; 0: the micro-chunk's own bias, as in the plain rule), and the bias advances in micro-chunks of w0_chunk
 *     examples inside every window (fm_sgd.h:34-37 summed per chunk).
 * Limits: hot = everything (or window == batch) is fmo_sgd_epoch_minibatch_ex at batch = window; hot = NULL (nothing) is that rule
 * at batch = `batch` with the bias lag counted in windows; batch = window = w0_chunk = 1 is the reference's online loop.
 * Stable while  learn_rate * curvature * (window * C_hot + batch * C_cold) <= 1  (tests/test_oracle_stability.py). */
void fmo_sgd_epoch_twolevel(fmo_model *m, const fmo_data *d, int task, double learn_rate,
                            double min_target, double max_target,
                            uint32_t batch, uint32_t window, uint32_t w0_chunk, int bias_lag, const uint8_t *hot);

/* the pipelined multi-GPU schedule: step 1 of batch b reads the parameters as they were BEFORE the update of batch
 * b-1 was applied (the gather of batch b overlaps the exchange / update of batch b-1); everything else as above. */
void fmo_sgd_epoch_minibatch_pipelined(fmo_model *m, const fmo_data *d, int task, double learn_rate,
                                       double min_target, double max_target,
                                       uint32_t batch, uint32_t w0_chunk, int bias_lag);

/* ---------------- SGDA (adaptive regularisation, Rendle WSDM'12) ---------------------------------------------
 * fm_learn_sgd_element_adapt_reg (src/libfm/src/fm_learn_sgd_element_adapt_reg.h).
 * state: reg_w[G], reg_v[G][k] per attribute group (`-meta`, :84-85; G = 1 without a meta file), and the shadow
 * gradients grad_w[n], grad_v[k][n] of the last theta step that touched each parameter (:90-95).  One epoch
 * (:262-279): for every train row a theta step (:136-169); from the second epoch on, each theta step is followed by
 * a lambda step (:201-248, through predict_scaled :171-199) on the next validation row (cyclic).  The learner
 * zeroes w and the model's reg values at the start of learn (:256-262). */
typedef struct {
  double *reg_w;     /* [G] */
  double *reg_v;     /* [G][k] */
  double *grad_w;    /* [n] */
  double *grad_v;    /* [k][n] factor-major */
  uint32_t val_pos;  /* next validation row */
  uint32_t num_groups;       /* G >= 1 */
  const uint32_t *group;     /* [n] attribute -> group (DataMetaInfo::attr_group, Data.h:41); NULL = all 0 */
} fmo_sgda_state;
void fmo_sgda_epoch(fmo_model *m, fmo_sgda_state *st, const fmo_data *train, const fmo_data *val, int task,
                    double learn_rate, double min_target, double max_target, int do_lambda_steps);
/* the batch restatement (GPU: fmx_sgda_epoch_minibatch): theta steps as a minibatch rule with the learned regularisation
 * and summed shadow gradients, then the batch's lambda steps with the regularisation frozen at its batch-start values,
 * summed and clamped once.  batch = w0_chunk = 1 is the reference loop (see fm_oracle.c). */
void fmo_sgda_epoch_minibatch(fmo_model *m, fmo_sgda_state *st, const fmo_data *train, const fmo_data *val, int task,
                              double learn_rate, double min_target, double max_target, int do_lambda_steps,
                              uint32_t batch, uint32_t w0_chunk);

/* ---------------- ALS (coordinate descent; "MCMC without sampling", libfm.cpp:135-139) ---------------- */

typedef struct {
  const fmo_entry *entries;   /* X^T: per FEATURE list of {row id, value}  (Data.h:292-341) */
  const uint64_t  *col_ptr;   /* [n+1] */
  uint64_t         n;         /* features */
  uint32_t         n_rows;
} fmo_data_t;

typedef struct { double e, q; } fmo_eq;   /* e_q_term, fm_learn_mcmc.h:46-49 */

/* build X^T from X (Data::create_data_t, Data.h:292-341); caller frees *entries_t, *col_ptr */
void fmo_transpose(const fmo_data *d, uint64_t n, fmo_entry **entries_t, uint64_t **col_ptr);

/* predict_data_and_write_to_eterms, non-relational part (fm_learn_mcmc.h:148-378): cache[c].e = y-hat(c),
 * cache[c].q = 0, computed through X^T in the reference's pass order (k passes of q, k passes of -0.5 v^2 x^2,
 * one linear pass, then + w0).  dt may describe fewer features than the model (a test set). */
void fmo_als_predict_eterms(const fmo_model *m, const fmo_data_t *dt, fmo_eq *cache);

/* one sweep = draw_all with do_sample = 0, do_multilevel = 0 (fm_learn_mcmc.h:430-641): alpha = alpha_0 = 1,
 * mu = mu_0 = 0; draw_w0 (:643-683) with reg = m->reg0; draw_w (:685-732) for every feature in X^T row order
 * with lambda = w_lambda, features without a training column get the empty-row draw (:467-476); per factor
 * add_main_q (:406-428) then draw_v (:792-847) with lambda = v_lambda.  cache[c].e holds (y-hat - target). */
void fmo_als_sweep(fmo_model *m, const fmo_data_t *dt, fmo_eq *cache, double w_lambda, double v_lambda);

/* the same with regularisation per attribute group (`-meta` + `-regular 'r0,w_1..w_G,v_1..v_G'`, libfm.cpp:353-363):
 * w_lambda[G], v_lambda[G][k] (the reference's DMatrix v_lambda(g,f), fm_learn_mcmc.h:1121-1122) */
typedef struct {
  const uint32_t *group;     /* [m->n] attribute -> group; NULL = all 0 */
  uint32_t        num_groups;
  const double   *w_lambda;  /* [G] */
  const double   *v_lambda;  /* [G][k] */
} fmo_als_reg;
void fmo_als_sweep_groups(fmo_model *m, const fmo_data_t *dt, fmo_eq *cache, const fmo_als_reg *reg);

/* fm_learn_mcmc::learn + fm_learn_mcmc_simultaneous::_learn with do_sample = 0 (fm_learn_mcmc.h:1160-1201,
 * fm_learn_mcmc_simultaneous.h:56-270): num_iter sweeps, e recomputed after each.  task regression: e -= y;
 * classification: e -= E[truncated normal] (:172-194, with the reference's 3.141 and its erf polynomial,
 * random.h:45-67).  Outputs (any may be NULL): test_pred_this[n_test] = pred_this of the last iteration
 * (unclamped y-hat for regression, cdf_gaussian(y-hat) for classification), train_metric[num_iter] =
 * rmse_train / acc_train per iteration. */
void fmo_als_learn(fmo_model *m, const fmo_data *train, const fmo_data *test, int task, int num_iter,
                   double w_lambda, double v_lambda, double min_target, double max_target,
                   double *test_pred_this, double *train_metric);
void fmo_als_learn_groups(fmo_model *m, const fmo_data *train, const fmo_data *test, int task, int num_iter,
                          const fmo_als_reg *reg, double min_target, double max_target,
                          double *test_pred_this, double *train_metric);

/* the reference's 5-term erf polynomial and cdf_gaussian (random.h:45-67) */
double fmo_erf(double x);
double fmo_cdf_gaussian(double x);

/* ---------------- synthetic workload (shared definition with the HIP generator) ---------------- */

/* One-hot field-structured rows: field t owns ids [t*fs, (t+1)*fs), fs = n/nnz;
 * id = t*fs + hash(seed,row,t) mod-ish fs; value 1.0; target = +1/-1 from a hash bit. */
uint64_t fmo_mix64(uint64_t x);
uint32_t fmo_synth_id(uint64_t seed, uint64_t row, uint32_t field, uint32_t field_size);
float    fmo_synth_target(uint64_t seed, uint64_t row);
void     fmo_synth_rows(uint64_t seed, uint64_t row0, uint32_t n_rows, uint32_t nnz, uint64_t n,
                        fmo_entry *entries, uint64_t *row_ptr, float *target);

/* deterministic cheap parameter fill used by bench.py for BOTH legs (not the reference RNG):
 * v[f][j] = stdev * u(seed, j, f) with u uniform in [-sqrt3, sqrt3) (unit variance). */
double fmo_init_value(uint64_t seed, uint64_t j, uint32_t f, double stdev);
/* fills m->v with fmo_init_value (and w, w0 with 0) using `threads` host threads (untimed set-up of the
 * cpu_baseline leg: 6.4e9 values at the north-star size) */
void fmo_fill_params(fmo_model *m, uint64_t seed, double stdev, int threads);
/* cpu_baseline: generates rows [row0,row0+n_rows) of the synthetic workload and runs ONE pass of the
 * online loop (fmo_sgd_epoch_online) over them on ONE thread; returns the seconds of the loop alone. */
double fmo_time_sgd_synth(fmo_model *m, uint64_t seed, uint64_t row0, uint32_t n_rows, uint32_t nnz,
                          int task, double learn_rate);

#ifdef __cplusplus
}
#endif
#endif /* FM_ORACLE_H_ */

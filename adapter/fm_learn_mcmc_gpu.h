// adapter/fm_learn_mcmc_gpu.h -- REFERENCE-SIDE binding of the ALS / MCMC learner (`-method als` = MCMC without
// sampling, src/libfm/libfm.cpp:135-139; `-method mcmc` = Gibbs sampling with hyper-priors) to libfmx.  Same rules as
// fm_learn_sgd_gpu.h: include it in the one translation unit that includes the reference headers, after
// fm_learn_mcmc_simultaneous.h.
//
// It derives from fm_learn_mcmc so that main()'s casts and field writes keep working unchanged
// (libfm.cpp:284-290: num_iter, num_eval_cases, do_sample, do_multilevel; :335-352: w_lambda / v_lambda from
// -regular) and replaces the iteration loop of fm_learn_mcmc_simultaneous::_learn (fm_learn_mcmc_simultaneous.h:
// 56-270) by fmx_als_begin / fmx_als_sweep.  Attribute groups (`-meta`) are passed through: meta->attr_group ->
// fmx_set_groups, w_lambda(g) / v_lambda(g,f) / w_mu(g) / v_mu(g,f) -> the opts tables.  Relations (`-relation`,
// Data::relation) go to fmx_upload_block_rows_ex: on one handle main rows and blocks stay apart (FMX_BLOCKS_KEEP) and the
// sweeps run over per-block-row caches like fm_learn_mcmc.h:478-527 / 734-790 / 849-909; on feature shards (or with
// gpu_blocks_expand) the library joins the blocks on the device instead.
// With do_multilevel the hyper-prior draws (draw_alpha :911-939, draw_w_lambda / draw_w_mu :941-1017, draw_v_lambda /
// draw_v_mu :1019-1097) stay on the host, in the reference's order and with the reference's own ran_gamma /
// ran_gaussian (src/util/random.h), fed by the per-group statistics fmx_als_moments reduces on the device; with
// do_sample the coordinate draws happen on the device from a counter-based generator (statistical, not bitwise,
// agreement with the stock sampler -- a parallel sampler cannot replay one sequential libc rand() stream).
#ifndef FM_LEARN_MCMC_GPU_H_
#define FM_LEARN_MCMC_GPU_H_

#include <cstring>
#include <ctime>
#include <sstream>
#include <vector>
#include <string>
#include "fmx.h"

class fm_learn_als_gpu : public fm_learn_mcmc {
 public:
  int gpu_device;
  // several GPUs from this ONE process (BASELINE configs[4]): the devices of the feature shards; empty = one unsharded
  // handle on gpu_device.  Distinct ordinals: RCCL; the same ordinal repeated: shards on one device (loopback exchange).
  std::vector<int> gpu_devices;
  bool gpu_blocks_expand;           // relations: materialise the joined rows on the device instead of keeping the blocks
  unsigned long long gpu_seed;                             // seed of the device-side coordinate draws (do_sample); 0 = derive
  fm_learn_als_gpu() : gpu_device(-1), gpu_blocks_expand(false), gpu_seed(0), h(NULL), grp(NULL) {}   // it from libc rand(), i.e. from main's -seed (libfm.cpp:115-116)
  virtual ~fm_learn_als_gpu() { if (grp) fmx_group_destroy(grp); for (size_t i = 0; i < hs.size(); i++) fmx_destroy(hs[i]); }

  virtual void learn(Data& train, Data& test) {            // fm_learn_mcmc::learn (:1160-1201) + _learn
    pred_sum_all.setSize(test.num_cases); pred_sum_all_but5.setSize(test.num_cases); pred_this.setSize(test.num_cases);
    pred_sum_all.init(0.0); pred_sum_all_but5.init(0.0); pred_this.init(0.0);
    const int world = gpu_devices.empty() ? 1 : (int)gpu_devices.size();
    const uint G = meta->num_attr_groups;
    for (int r = 0; r < world; r++) {
      fmx_config c;
      memset(&c, 0, sizeof(c));                              // (place_candidates, als_split_min, exchange_runs: library defaults)
      c.num_attribute = fm->num_attribute; c.num_factor = fm->num_factor; c.k0 = fm->k0; c.k1 = fm->k1;
      c.task = task; c.reg0 = fm->reg0; c.regw = fm->regw; c.regv = fm->regv; c.learn_rate = 0;
      c.min_target = min_target; c.max_target = max_target; c.device = gpu_devices.empty() ? gpu_device : gpu_devices[r];
      c.shard_rank = r; c.shard_world = world; c.shard_hash = world > 1 ? 1 : 0;
      fmx_handle x = NULL;
      if (fmx_create(&c, &x) != FMX_OK) throw std::string(fmx_last_error(NULL));
      hs.push_back(x);
      if (r == 0) h = x;
      if (G > 1 && fmx_set_groups(x, (const uint32_t*)meta->attr_group.value, G) != FMX_OK) throw std::string(fmx_last_error(x));   // DVector<uint>, Data.h:41
    }
    if (fmx_group_create(&hs[0], world, &grp) != FMX_OK) throw std::string(fmx_last_error(hs[0]));
    gcheck(fmx_group_set_params(grp, fm->w0, fm->w.value, fm->num_factor > 0 ? fm->v.value[0] : NULL));   // the block crosses PCIe once
    upload(0, train); upload(1, test);
    gcheck(fmx_group_als_begin(grp, 0));
    fmx_als_opts o;
    memset(&o, 0, sizeof(o));
    const int k = fm->num_factor;
    if (do_sample && gpu_seed == 0) gpu_seed = (((unsigned long long)rand()) << 31) ^ (unsigned long long)rand() ^ 0x9E3779B97F4A7C15ULL;
    std::vector<double> mom(2 + (size_t)2 * G * (1 + k));
    std::vector<double> p(test.num_cases);
    for (uint i = 0; i < num_iter; i++) {
      double iteration_time = getusertime();                                 // _learn :85-87
      clock_t iteration_time3 = clock();
      double iteration_time4 = getusertime4();
      if (do_multilevel) {                                                   // the prior draws of draw_all (:433-452, :519-527)
        gcheck(fmx_group_als_moments(grp, &mom[0]));
        draw_priors_from_moments(mom, train.num_cases);
      } else {
        alpha = alpha_0; w_mu.init(mu_0); if (k > 0) v_mu.init(mu_0);        // draw_alpha / draw_*_mu without multilevel
      }
      if (log != NULL) {                                                     // what draw_all logs (:434-436, :446-451, :518-525)
        std::ostringstream ss;
        log->log("alpha", alpha);
        for (uint g = 0; g < G; g++) {
          if (fm->k1) {
            ss.str(""); ss << "wmu[" << g << "]"; log->log(ss.str(), w_mu(g));
            ss.str(""); ss << "wlambda[" << g << "]"; log->log(ss.str(), w_lambda(g));
          }
          for (int f = 0; f < k; f++) {
            ss.str(""); ss << "vmu[" << g << "," << f << "]"; log->log(ss.str(), v_mu(g, f));
            ss.str(""); ss << "vlambda[" << g << "," << f << "]"; log->log(ss.str(), v_lambda(g, f));
          }
        }
      }
      o.alpha = alpha; o.do_sample = do_sample ? 1 : 0; o.seed = (uint64_t)gpu_seed;
      o.w_mu = w_mu(0); o.w_lambda = w_lambda(0);
      o.v_mu = k > 0 ? v_mu(0, 0) : 0.0; o.v_lambda = k > 0 ? v_lambda(0, 0) : 0.0;
      if (G > 1) {                                                           // DVector[G], DMatrix[G][k]: the reference's own storage
        o.num_groups = G; o.w_lambda_g = w_lambda.value; o.w_mu_g = w_mu.value;
        o.v_lambda_gf = k > 0 ? v_lambda.value[0] : NULL; o.v_mu_gf = k > 0 ? v_mu.value[0] : NULL;
      } else {
        o.v_lambda_f = k > 0 ? v_lambda.value[0] : NULL; o.v_mu_f = k > 0 ? v_mu.value[0] : NULL;
      }
      fmx_als_stats st;
      gcheck(fmx_group_als_sweep(grp, &o, &st));
      gcheck(fmx_group_predict(grp, 1, p.empty() ? NULL : &p[0]));
      for (uint c2 = 0; c2 < test.num_cases; c2++) {                       // _learn :127-138 / :151-161
        double v = p[c2];
        if (task == TASK_REGRESSION) {
          pred_this(c2) = v;
          v = std::min(max_target, v); v = std::max(min_target, v);
        } else {
          v = cdf_gaussian(v);
          pred_this(c2) = v;
        }
        pred_sum_all(c2) += v;
        if (i >= 5) pred_sum_all_but5(c2) += v;
      }
      iteration_time = (getusertime() - iteration_time);                     // :199-206
      iteration_time3 = clock() - iteration_time3;
      iteration_time4 = (getusertime4() - iteration_time4);
      if (log != NULL) {
        log->log("time_learn", iteration_time);
        log->log("time_learn2", (double)iteration_time3 / CLOCKS_PER_SEC);
        log->log("time_learn4", (double)iteration_time4);
      }
      // the test metrics over the first num_eval_cases cases (:209-264): this iteration's prediction, the running mean over all iterations
      // and over all but the first five -- the reference's own normalisers, 1.0 / (i - 5 + 1) in unsigned arithmetic for i < 5 included
      if (task == TASK_REGRESSION) {
        double rmse_this, mae_this, rmse_all, mae_all, rmse_but5, mae_but5;
        eval_regression(pred_this, test.target, 1.0, rmse_this, mae_this, num_eval_cases);
        eval_regression(pred_sum_all, test.target, 1.0 / (i + 1), rmse_all, mae_all, num_eval_cases);
        eval_regression(pred_sum_all_but5, test.target, 1.0 / (i - 5 + 1), rmse_but5, mae_but5, num_eval_cases);
        std::cout << "#Iter=" << std::setw(3) << i << "\tTrain=" << st.train_metric << "\tTest=" << rmse_all << std::endl;
        if (log != NULL) {
          log->log("rmse", rmse_all); log->log("mae", mae_all);
          log->log("rmse_mcmc_this", rmse_this); log->log("rmse_mcmc_all", rmse_all); log->log("rmse_mcmc_all_but5", rmse_but5);
          log->newLine();
        }
      } else {
        double acc_this, acc_all, acc_but5, ll_this, ll_all, ll_but5;
        eval_classification(pred_this, test.target, 1.0, acc_this, ll_this, num_eval_cases);
        eval_classification(pred_sum_all, test.target, 1.0 / (i + 1), acc_all, ll_all, num_eval_cases);
        eval_classification(pred_sum_all_but5, test.target, 1.0 / (i - 5 + 1), acc_but5, ll_but5, num_eval_cases);
        std::cout << "#Iter=" << std::setw(3) << i << "\tTrain=" << st.train_metric << "\tTest=" << acc_all << "\tTest(ll)=" << ll_all << std::endl;
        if (log != NULL) {
          log->log("accuracy", acc_all);
          log->log("acc_mcmc_this", acc_this); log->log("acc_mcmc_all", acc_all); log->log("acc_mcmc_all_but5", acc_but5);
          log->log("ll_mcmc_this", ll_this); log->log("ll_mcmc_all", ll_all); log->log("ll_mcmc_all_but5", ll_but5);
          log->newLine();
        }
      }
    }
    gcheck(fmx_group_als_end(grp));
    for (size_t r = 0; r < hs.size(); r++)                   // every shard writes its own features into the host block
      if (fmx_get_params(hs[r], &fm->w0, fm->w.value, fm->num_factor > 0 ? fm->v.value[0] : NULL) != FMX_OK) throw std::string(fmx_last_error(hs[r]));
  }

 protected:
  fmx_handle h;                                            // shard 0 (the only handle without gpu_devices)
  std::vector<fmx_handle> hs;
  fmx_group grp;
  // fm_learn_mcmc_simultaneous::_evaluate / _evaluate_class (fm_learn_mcmc_simultaneous.h:272-313) over the cases [0, num_eval_cases): the
  // scaled prediction clamped to the target range -> RMSE, MAE; the scaled probability -> accuracy at 0.5 and the base-10 log-likelihood of
  // the probability clipped to [0.01, 0.99]
  void eval_regression(DVector<double>& pred, DVector<DATA_FLOAT>& target, double normalizer, double& rmse, double& mae, uint upto) {
    double se = 0, ae = 0; uint cnt = 0;
    for (uint c = 0; c < std::min((uint)pred.dim, upto); c++) {
      double q = pred(c) * normalizer;
      q = std::min(max_target, q); q = std::max(min_target, q);
      const double err = q - target(c);
      se += err * err; ae += std::abs((double)err); cnt++;
    }
    rmse = std::sqrt(se / cnt); mae = ae / cnt;
  }
  void eval_classification(DVector<double>& pred, DVector<DATA_FLOAT>& target, double normalizer, double& accuracy, double& loglikelihood, uint upto) {
    double ll = 0; uint acc = 0, cnt = 0;
    for (uint c = 0; c < std::min((uint)pred.dim, upto); c++) {
      const double q = pred(c) * normalizer;
      if (((q >= 0.5) && (target(c) > 0.0)) || ((q < 0.5) && (target(c) < 0.0))) acc++;
      const double m = (target(c) + 1.0) * 0.5;
      double pll = q;
      if (pll > 0.99) pll = 0.99;
      if (pll < 0.01) pll = 0.01;
      ll -= m * log10(pll) + (1 - m) * log10(1 - pll);
      cnt++;
    }
    loglikelihood = ll / cnt; accuracy = (double)acc / cnt;
  }
  void check(int rc) { if (rc != FMX_OK) throw std::string(fmx_last_error(h)); }
  void gcheck(int rc) { if (rc != FMX_OK) throw std::string(fmx_group_last_error(grp)); }

  // mom = fmx_als_moments: {sum e^2, sum e} then [1 + k][G][2] = per coordinate family and group {sum theta, sum theta^2}
  void draw_priors_from_moments(const std::vector<double>& mom, uint num_train_total) {
    const uint G = meta->num_attr_groups;
    const int k = fm->num_factor;
    {                                                                        // draw_alpha :911-939
      const double alpha_n = alpha_0 + num_train_total, gamma_n = gamma_0 + mom[0];
      const double a = ran_gamma(alpha_n / 2.0, gamma_n / 2.0);
      if (!(std::isnan(a) || std::isinf(a))) alpha = a;
    }
    for (int fam = 0; fam <= k; fam++) {                                     // family 0 = w (:941-1017), 1+f = v_f (:1019-1097)
      if (fam == 0 && !fm->k1) continue;
      const double* m = &mom[2 + (size_t)fam * G * 2];
      // all lambdas of the family first, then all mus -- and for v: draw_v_lambda() over every f, THEN draw_v_mu()
      // (:519-521); both orders agree here because lambda(g,f) only needs mu(g,f) of the previous iteration
      for (uint g = 0; g < G; g++) {
        double& lam = (fam == 0) ? w_lambda(g) : v_lambda(g, fam - 1);
        const double mu = (fam == 0) ? w_mu(g) : v_mu(g, fam - 1);
        const double n_g = meta->num_attr_per_group(g);
        // sum_j (theta_j - mu)^2 = sum theta^2 - 2 mu sum theta + n_g mu^2  (:987-992, :1067-1072)
        const double gam = beta_0 * (mu - mu_0) * (mu - mu_0) + gamma_0 + (m[2 * g + 1] - 2 * mu * m[2 * g] + n_g * mu * mu);
        const double l = ran_gamma((alpha_0 + n_g + 1) / 2.0, gam / 2.0);
        if (!(std::isnan(l) || std::isinf(l))) lam = l;
      }
      for (uint g = 0; g < G; g++) {
        double& mu = (fam == 0) ? w_mu(g) : v_mu(g, fam - 1);
        const double lam = (fam == 0) ? w_lambda(g) : v_lambda(g, fam - 1);
        const double n_g = meta->num_attr_per_group(g);
        const double mean = (m[2 * g] + beta_0 * mu_0) / (n_g + beta_0);   // :946-953, :1026-1033
        const double x = ran_gaussian(mean, std::sqrt(1.0 / ((n_g + beta_0) * lam)));
        if (!(std::isnan(x) || std::isinf(x))) mu = x;
      }
    }
  }
  // ALS data sets are loaded transposed-only by main (has_x = false, libfm.cpp:143-147): rebuild rows from X^T
  struct Rows { std::vector<uint64> row_ptr; std::vector< sparse_entry<DATA_FLOAT> > ent; };
  static void rows_from_xt(LargeSparseMatrix<DATA_FLOAT>* xt, uint num_cases, Rows& out) {
    out.row_ptr.assign((size_t)num_cases + 1, 0);
    for (xt->begin(); !xt->end(); xt->next()) {
      sparse_row<DATA_FLOAT>& col = xt->getRow();
      for (uint i = 0; i < col.size; i++) out.row_ptr[col.data[i].id + 1]++;
    }
    for (uint r = 0; r < num_cases; r++) out.row_ptr[r + 1] += out.row_ptr[r];
    out.ent.resize(out.row_ptr[num_cases]);
    std::vector<uint64> fill(out.row_ptr.begin(), out.row_ptr.end() - 1);
    for (xt->begin(); !xt->end(); xt->next()) {
      sparse_row<DATA_FLOAT>& col = xt->getRow();
      uint j = xt->getRowIndex();
      for (uint i = 0; i < col.size; i++) { sparse_entry<DATA_FLOAT> e; e.id = j; e.value = col.data[i].value; out.ent[fill[col.data[i].id]++] = e; }
    }
  }
  void upload(int slot, Data& d) {
    Rows main;
    rows_from_xt(d.data_t, d.num_cases, main);
    std::vector<Rows> blocks(d.relation.dim);
    std::vector<fmx_relation> rel(d.relation.dim);
    for (uint r = 0; r < d.relation.dim; r++) {                             // RelationJoin, relation.h:53-60
      RelationData* rd = d.relation(r).data;
      rows_from_xt(rd->data_t, rd->num_cases, blocks[r]);
      memset(&rel[r], 0, sizeof(fmx_relation));
      rel[r].entries = blocks[r].ent.empty() ? NULL : &blocks[r].ent[0];
      rel[r].row_ptr = (const uint64_t*)&blocks[r].row_ptr[0];
      rel[r].n_rows = rd->num_cases; rel[r].nnz = blocks[r].ent.size();
      rel[r].data_row_to_relation_row = (const uint32_t*)d.relation(r).data_row_to_relation_row.value;
      rel[r].attr_offset = rd->attr_offset;
    }
    uint32_t how = (hs.size() > 1 || gpu_blocks_expand || rel.empty()) ? FMX_BLOCKS_EXPAND : FMX_BLOCKS_KEEP;   // shards sweep joined rows
    if (rel.empty()) {                                                      // plain rows: across PCIe once, filtered per shard on its device
      gcheck(fmx_group_upload_rows(grp, slot, main.ent.empty() ? NULL : &main.ent[0], (const uint64_t*)&main.row_ptr[0], d.target.value,
                                   d.num_cases, main.ent.size()));
      return;
    }
    for (size_t r = 0; r < hs.size(); r++)                                  // (a shard joins the blocks on the host and keeps its own features)
      if (fmx_upload_block_rows_ex(hs[r], slot, main.ent.empty() ? NULL : &main.ent[0], (const uint64_t*)&main.row_ptr[0], d.target.value,
                                   d.num_cases, main.ent.size(), rel.empty() ? NULL : &rel[0], (uint32_t)rel.size(), how) != FMX_OK)
        throw std::string(fmx_last_error(hs[r]));
  }
};

#endif /* FM_LEARN_MCMC_GPU_H_ */

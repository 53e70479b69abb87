// adapter/fm_learn_mcmc_gpu.h -- REFERENCE-SIDE binding of the ALS learner (`-method als` = MCMC without sampling,
// src/libfm/libfm.cpp:135-139) to libfmx.  Same rules as fm_learn_sgd_gpu.h: include it in the one translation unit
// that includes the reference headers, after fm_learn_mcmc_simultaneous.h.
//
// It derives from fm_learn_mcmc so that main()'s casts and field writes keep working unchanged
// (libfm.cpp:284-290: num_iter, num_eval_cases, do_sample, do_multilevel; :335-352: w_lambda / v_lambda from
// -regular) and replaces the iteration loop of fm_learn_mcmc_simultaneous::_learn (fm_learn_mcmc_simultaneous.h:
// 56-270) by fmx_als_begin / fmx_als_sweep.  Only do_sample = 0, do_multilevel = 0.  Attribute groups (`-meta`) are
// passed through: meta->attr_group -> fmx_set_groups, w_lambda(g) / v_lambda(g,f) -> the opts tables.  Relations
// (`-relation`, Data::relation) go to fmx_upload_block_rows, which joins the blocks on the device.
#ifndef FM_LEARN_MCMC_GPU_H_
#define FM_LEARN_MCMC_GPU_H_

#include <cstring>
#include <vector>
#include <string>
#include "fmx.h"

class fm_learn_als_gpu : public fm_learn_mcmc {
 public:
  int gpu_device;
  fm_learn_als_gpu() : gpu_device(-1), h(NULL) {}
  virtual ~fm_learn_als_gpu() { if (h) fmx_destroy(h); }

  virtual void learn(Data& train, Data& test) {            // fm_learn_mcmc::learn (:1160-1201) + _learn
    if (do_sample || do_multilevel) throw "fm_learn_als_gpu: only -method als (no sampling) is bound";
    pred_sum_all.setSize(test.num_cases); pred_sum_all_but5.setSize(test.num_cases); pred_this.setSize(test.num_cases);
    pred_sum_all.init(0.0); pred_sum_all_but5.init(0.0); pred_this.init(0.0);
    fmx_config c;
    c.num_attribute = fm->num_attribute; c.num_factor = fm->num_factor; c.k0 = fm->k0; c.k1 = fm->k1;
    c.task = task; c.reg0 = fm->reg0; c.regw = fm->regw; c.regv = fm->regv; c.learn_rate = 0;
    c.min_target = min_target; c.max_target = max_target; c.device = gpu_device;
    c.shard_rank = 0; c.shard_world = 1; c.reserved = 0;
    if (fmx_create(&c, &h) != FMX_OK) throw std::string(fmx_last_error(NULL));
    check(fmx_set_params(h, fm->w0, fm->w.value, fm->num_factor > 0 ? fm->v.value[0] : NULL));
    upload(0, train); upload(1, test);
    const uint G = meta->num_attr_groups;
    if (G > 1) check(fmx_set_groups(h, (const uint32_t*)meta->attr_group.value, G));   // DVector<uint>, Data.h:41
    check(fmx_als_begin(h, 0));
    fmx_als_opts o;
    memset(&o, 0, sizeof(o));
    o.alpha = alpha_0; o.w_mu = mu_0; o.v_mu = mu_0;                       // draw_alpha / draw_*_mu without multilevel
    o.w_lambda = w_lambda(0); o.v_lambda = fm->num_factor > 0 ? v_lambda(0, 0) : 0.0;
    if (G > 1) {                                                           // w_lambda(g), v_lambda(g,f): DVector[G], DMatrix[G][k]
      o.num_groups = G; o.w_lambda_g = w_lambda.value;
      o.v_lambda_gf = fm->num_factor > 0 ? v_lambda.value[0] : NULL;
    }
    std::vector<double> p(test.num_cases);
    for (uint i = 0; i < num_iter; i++) {
      fmx_als_stats st;
      check(fmx_als_sweep(h, &o, &st));
      check(fmx_predict(h, 1, p.empty() ? NULL : &p[0]));
      double rmse_or_acc = 0;
      for (uint c2 = 0; c2 < test.num_cases; c2++) {                       // _learn :127-138 / :151-161
        double v = p[c2];
        if (task == TASK_REGRESSION) {
          pred_this(c2) = v;
          v = std::min(max_target, v); v = std::max(min_target, v);
          pred_sum_all(c2) += v;
          double err = pred_sum_all(c2) / (i + 1) - test.target(c2);
          rmse_or_acc += err * err;
        } else {
          v = cdf_gaussian(v);
          pred_this(c2) = v;
          pred_sum_all(c2) += v;
          if (((pred_sum_all(c2) / (i + 1) >= 0.5) && (test.target(c2) >= 0)) || ((pred_sum_all(c2) / (i + 1) < 0.5) && (test.target(c2) < 0))) rmse_or_acc += 1;
        }
      }
      rmse_or_acc = (task == TASK_REGRESSION) ? std::sqrt(rmse_or_acc / test.num_cases) : rmse_or_acc / test.num_cases;
      std::cout << "#Iter=" << std::setw(3) << i << "\tTrain=" << st.train_metric << "\tTest=" << rmse_or_acc << std::endl;
    }
    check(fmx_als_end(h));
    check(fmx_get_params(h, &fm->w0, fm->w.value, fm->num_factor > 0 ? fm->v.value[0] : NULL));
  }

 protected:
  fmx_handle h;
  void check(int rc) { if (rc != FMX_OK) throw std::string(fmx_last_error(h)); }
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
    check(fmx_upload_block_rows(h, slot, main.ent.empty() ? NULL : &main.ent[0], (const uint64_t*)&main.row_ptr[0], d.target.value,
                                d.num_cases, main.ent.size(), rel.empty() ? NULL : &rel[0], (uint32_t)rel.size()));
  }
};

#endif /* FM_LEARN_MCMC_GPU_H_ */

// adapter/fm_learn_sgd_gpu.h -- the REFERENCE-SIDE binding of libfmx (include/fmx.h).
//
// This header is what a libFM maintainer adds to src/libfm/src/ : `fm_learn` subclasses that replace
//   fm_learn_sgd_element            (/root/reference/src/libfm/src/fm_learn_sgd_element.h:34-78)            -> fm_learn_sgd_gpu
//   fm_learn_sgd_element_adapt_reg  (/root/reference/src/libfm/src/fm_learn_sgd_element_adapt_reg.h:44-345) -> fm_learn_sgda_gpu
// and forward the hot path -- fm_model::predict + fm_SGD over a data set -- to the MI355X library through its C-ABI.
// Everything else (CLI, Data loading, fm_model, rlog, -out / -save_model) stays the reference's own code.
//
// It must be included in the ONE translation unit that includes the reference headers (they define non-inline
// functions, src/libfm/libfm.cpp:49-57), after fm_learn_sgd.h / fm_learn_sgd_element_adapt_reg.h.  See INTEGRATION.md
// for the patch of main().  Compiled and exercised against the real reference classes by oracle/ref_harness.cpp
// (modes "sgd_gpu", "sgda_gpu").
#ifndef FM_LEARN_SGD_GPU_H_
#define FM_LEARN_SGD_GPU_H_

#include <cstring>
#include <vector>
#include <string>
#include "fmx.h"

// what both learners share: the device context, the uploaded data sets, evaluate and predict through the C-ABI.
// Base = the reference class main() expects behind the fm_learn pointer (its casts and field writes keep working).
template <class Base>
class fmx_sgd_binding : public Base {
 public:
  int gpu_device;
  // several GPUs from this ONE process: the devices of the feature shards (empty = one unsharded handle on gpu_device).
  // Distinct ordinals -> RCCL over xGMI; the same ordinal repeated -> shards on one device with the library's loopback
  // exchange (tests, single-GPU boxes).  The model is split by the hashed ownership rule (fmx_config::shard_hash = 1).
  std::vector<int> gpu_devices;
  fmx_sgd_binding() : gpu_device(-1), h(NULL), grp(NULL) {}
  virtual ~fmx_sgd_binding() {
    if (grp) fmx_group_destroy(grp);
    for (size_t i = 0; i < hs.size(); i++) fmx_destroy(hs[i]);
  }

  virtual double evaluate(Data& data) {                   // fm_learn::evaluate (fm_learn.h:93-153)
    open();
    return evaluate_slot(slot_of(data));
  }

  virtual void predict(Data& data, DVector<double>& out) {   // fm_learn_sgd::predict (fm_learn_sgd.h:76-90)
    assert(data.data->getNumRows() == out.dim);
    open();
    gcheck(fmx_group_predict(grp, slot_of(data), out.value));
    for (uint i = 0; i < out.dim; i++) {
      double p = out(i);
      if (this->task == Base::TASK_REGRESSION) {
        p = std::min(this->max_target, p);
        p = std::max(this->min_target, p);
      } else if (this->task == Base::TASK_CLASSIFICATION) {
        p = 1.0 / (1.0 + exp(-p));
      } else {
        throw "task not supported";
      }
      out(i) = p;
    }
  }

 protected:
  fmx_handle h;                                            // shard 0 (the only handle without gpu_devices)
  std::vector<fmx_handle> hs;                              // every shard
  fmx_group grp;
  std::vector<Data*> slots;

  void check(int rc) { if (rc != FMX_OK) throw std::string(fmx_last_error(h)); }
  void gcheck(int rc) { if (rc != FMX_OK) throw std::string(fmx_group_last_error(grp)); }

  void open() {                                            // once: device context(s) + parameters (fm_model.h:46-48)
    if (h) return;
    fm_model* fm = this->fm;
    const int world = gpu_devices.empty() ? 1 : (int)gpu_devices.size();
    for (int r = 0; r < world; r++) {
      fmx_config c;
      memset(&c, 0, sizeof(c));                              // (place_candidates, als_split_min, exchange_runs: library defaults)
      c.num_attribute = fm->num_attribute; c.num_factor = fm->num_factor; c.k0 = fm->k0; c.k1 = fm->k1;
      c.task = this->task; c.reg0 = fm->reg0; c.regw = fm->regw; c.regv = fm->regv; c.learn_rate = this->learn_rate;
      c.min_target = this->min_target; c.max_target = this->max_target;
      c.device = gpu_devices.empty() ? gpu_device : gpu_devices[r];
      c.shard_rank = r; c.shard_world = world; c.shard_hash = world > 1 ? 1 : 0;
      fmx_handle x = NULL;
      if (fmx_create(&c, &x) != FMX_OK) throw std::string(fmx_last_error(NULL));
      hs.push_back(x);
      if (r == 0) h = x;
    }
    if (fmx_group_create(&hs[0], world, &grp) != FMX_OK) throw std::string(fmx_last_error(hs[0]));
    // the block crosses PCIe once; every shard keeps its own features (fmx_group_set_params)
    gcheck(fmx_group_set_params(grp, fm->w0, fm->w.value, fm->num_factor > 0 ? fm->v.value[0] : NULL));
  }

  void fetch_params() {                                    // main() reads fm afterwards (libfm.cpp:418-434)
    fm_model* fm = this->fm;
    for (size_t r = 0; r < hs.size(); r++)                 // every shard writes its own features into the host block
      if (fmx_get_params(hs[r], &fm->w0, fm->w.value, fm->num_factor > 0 ? fm->v.value[0] : NULL) != FMX_OK) throw std::string(fmx_last_error(hs[r]));
  }

  int slot_of(Data& d) {                                   // uploads a Data set once (Data.h:49-73)
    for (size_t i = 0; i < slots.size(); i++) if (slots[i] == &d) return (int)i;
    if (slots.size() >= FMX_MAX_SLOTS) throw "too many data sets";
    // generic over LargeSparseMatrixMemory / LargeSparseMatrixHD: walk the iterator protocol (fmatrix.h:52-62)
    std::vector< sparse_entry<DATA_FLOAT> > ent;
    std::vector<uint64> row_ptr;
    ent.reserve(d.data->getNumValues());
    row_ptr.reserve(d.data->getNumRows() + 1);
    row_ptr.push_back(0);
    for (d.data->begin(); !d.data->end(); d.data->next()) {
      sparse_row<DATA_FLOAT>& r = d.data->getRow();
      ent.insert(ent.end(), r.data, r.data + r.size);
      row_ptr.push_back(ent.size());
    }
    // the rows cross PCIe once; a shard keeps the entries of its own features (filtered on its device)
    gcheck(fmx_group_upload_rows(grp, (int)slots.size(), ent.empty() ? NULL : &ent[0], (const uint64_t*)&row_ptr[0],
                                 d.target.value, d.data->getNumRows(), ent.size()));
    slots.push_back(&d);
    return (int)slots.size() - 1;
  }

  double evaluate_slot(int s) {
    fmx_eval ev;
    gcheck(fmx_group_evaluate(grp, s, &ev));
    if (this->log != NULL) {                                // same rlog fields as fm_learn.h:124-127,146-150
      if (this->task == Base::TASK_REGRESSION) { this->log->log("rmse", ev.rmse); this->log->log("mae", ev.mae); }
      else { this->log->log("accuracy", ev.accuracy); }
      this->log->log("time_pred", ev.device_seconds);
    }
    return this->task == Base::TASK_REGRESSION ? ev.rmse : ev.accuracy;
  }
};

// ---- `-method sgd` -------------------------------------------------------------------------------------------
class fm_learn_sgd_gpu : public fmx_sgd_binding<fm_learn_sgd> {
 public:
  // GPU-only knobs (defaults = library defaults); everything else is inherited and set by main() as before
  int gpu_mode;          // FMX_SGD_SEQUENTIAL | FMX_SGD_MINIBATCH | FMX_SGD_HOGWILD
  int gpu_apply;         // FMX_APPLY_*
  uint gpu_batch, gpu_w0_chunk, gpu_flags, gpu_bias_lag;   // fmx_sgd_opts::flags / ::bias_lag

  // defaults: the batch rule in one pass (FMX_APPLY_FUSED, bias lag 2), batch chosen by the library from the rows' collision
  // mass (fmx_sgd_opts::batch = 0), an explicit batch the rule diverges at is refused (FMX_E_ARG -> thrown like any error)
  fm_learn_sgd_gpu() : gpu_mode(FMX_SGD_MINIBATCH), gpu_apply(FMX_APPLY_FUSED), gpu_batch(0), gpu_w0_chunk(0), gpu_flags(FMX_FLAG_REJECT_UNSTABLE | FMX_FLAG_KEEP_WSIDE), gpu_bias_lag(2) {}   // (learn() evaluates the train set after every epoch, fm_learn_sgd_element.h:69-70: the epochs keep the slot's weight side stream for it)

  virtual void init() {                                   // fm_learn_sgd_element::init (:40-46)
    fm_learn_sgd::init();
    if (log != NULL) log->addField("rmse_train", std::numeric_limits<double>::quiet_NaN());
  }

  virtual void learn(Data& train, Data& test) {           // fm_learn_sgd_element::learn (:48-78)
    fm_learn_sgd::learn(train, test);                     // prints learnrate/#iterations, rejects relations
    std::cout << "SGD: DON'T FORGET TO SHUFFLE THE ROWS IN TRAINING DATA TO GET THE BEST RESULTS." << std::endl;
    open();
    const int s_train = slot_of(train), s_test = slot_of(test);
    fmx_sgd_opts opts; opts.mode = gpu_mode; opts.apply = gpu_apply; opts.batch = gpu_batch;
    opts.w0_chunk = gpu_w0_chunk; opts.flags = gpu_flags; opts.bias_lag = gpu_bias_lag;
    for (int i = 0; i < num_iter; i++) {
      fmx_epoch_stats st;
      gcheck(fmx_group_sgd_epoch(grp, s_train, &opts, &st));
      if (i == 0 && (st.status & FMX_STAT_BATCH_CUT))
        std::cerr << "libfmx: batch " << st.batch_used << " (collision mass of the rows " << st.collision_mass << ", gain " << st.batch_gain << ")" << std::endl;
      double rmse_train = evaluate_slot(s_train);
      double rmse_test = evaluate_slot(s_test);
      std::cout << "#Iter=" << std::setw(3) << i << "\tTrain=" << rmse_train << "\tTest=" << rmse_test << std::endl;
      if (log != NULL) {
        log->log("rmse_train", rmse_train);
        log->log("time_learn", st.device_seconds);
        log->newLine();
      }
    }
    fetch_params();
  }
};

// ---- `-method sgda` (self-adaptive regularisation) -----------------------------------------------------------
// init() is the reference's own (fm_learn_sgd_element_adapt_reg::init :80-134 sizes reg_w / reg_v / grad_* and registers
// the rlog fields); main() sets `validation` through its cast (libfm.cpp:276-279).  learn() replaces :250-345:
// theta steps on the train rows interleaved with lambda steps on the validation rows run on the device
// (fmx_sgda_epoch), attribute groups (`-meta`) included; the learned reg_w(g), reg_v(g,f) come back after every
// iteration so that -rlog reports them like the stock learner.  The host-side shadow gradients grad_w / grad_v are
// not mirrored (they are scratch of the device kernel).
#ifdef FM_LEARN_SGD_ELEMENT_ADAPT_REG_H_
class fm_learn_sgda_gpu : public fmx_sgd_binding<fm_learn_sgd_element_adapt_reg> {
 public:
  // 0 (default): the reference's strictly online order on one wavefront (a parity instrument, slower than the CPU);
  // > 0: the batch form (fmx_sgda_epoch_minibatch) with batches of gpu_batch rows and bias micro-chunks of gpu_w0_chunk
  uint gpu_batch, gpu_w0_chunk;
  fm_learn_sgda_gpu() : gpu_batch(0), gpu_w0_chunk(0) {}
  virtual void learn(Data& train, Data& test) {
    fm_learn_sgd::learn(train, test);                     // prints learnrate/#iterations, rejects relations
    if (validation == NULL) throw "sgda needs a validation set";
    std::cout << "Training using self-adaptive-regularization SGD." << std::endl
              << "DON'T FORGET TO SHUFFLE THE ROWS IN TRAINING AND VALIDATION DATA TO GET THE BEST RESULTS." << std::endl;
    fm->w.init(0); fm->reg0 = 0; fm->regw = 0; fm->regv = 0;          // :256-259 (the device does the same in fmx_sgda_begin)
    reg_w.init(0.0); reg_v.init(0.0);
    std::cout << "Using " << train.data->getNumRows() << " rows for training model parameters and "
              << validation->data->getNumRows() << " for training shrinkage." << std::endl;
    open();
    const uint G = meta->num_attr_groups;
    if (G > 1) check(fmx_set_groups(h, (const uint32_t*)meta->attr_group.value, G));   // DVector<uint>, Data.h:41
    const int s_train = slot_of(train), s_test = slot_of(test), s_val = slot_of(*validation);
    check(fmx_sgda_begin(h));
    std::vector<double> reg((size_t)G * (1 + fm->num_factor));
    for (int i = 0; i < num_iter; i++) {
      fmx_epoch_stats st;
      if (gpu_batch > 0) check(fmx_sgda_epoch_minibatch(h, s_train, s_val, i > 0, gpu_batch, gpu_w0_chunk, &st));
      else check(fmx_sgda_epoch(h, s_train, s_val, i > 0, &st));        // no lambda steps in the first iteration (:269)
      double rmse_val = evaluate_slot(s_val);
      double rmse_train = evaluate_slot(s_train);
      double rmse_test = evaluate_slot(s_test);
      std::cout << "#Iter=" << std::setw(3) << i << "\tTrain=" << rmse_train << "\tTest=" << rmse_test << std::endl;
      check(fmx_sgda_get_reg(h, &reg[0]));
      for (uint g = 0; g < G; g++) {
        reg_w(g) = reg[(size_t)g * (1 + fm->num_factor)];
        for (int f = 0; f < fm->num_factor; f++) reg_v(g, f) = reg[(size_t)g * (1 + fm->num_factor) + 1 + f];
      }
      if (log != NULL) {                                                // the reg fields of :318-339
        for (uint g = 0; g < G; g++) {
          { std::ostringstream ss; ss << "regw[" << g << "]"; log->log(ss.str(), reg_w(g)); }
          for (int f = 0; f < fm->num_factor; f++) { std::ostringstream ss; ss << "regv[" << g << "," << f << "]"; log->log(ss.str(), reg_v(g, f)); }
        }
        log->log("rmse_train", rmse_train);
        log->log("rmse_val", rmse_val);
        log->log("time_learn", st.device_seconds);
        log->newLine();
      }
    }
    check(fmx_sgda_end(h));
    fetch_params();
  }
};
#endif

#endif /* FM_LEARN_SGD_GPU_H_ */

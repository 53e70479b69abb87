// adapter/fm_learn_sgd_gpu.h -- the REFERENCE-SIDE binding of libfmx (include/fmx.h).
//
// This header is what a libFM maintainer adds to src/libfm/src/ : an `fm_learn` subclass that replaces
// fm_learn_sgd_element (/root/reference/src/libfm/src/fm_learn_sgd_element.h:34-78) and forwards the hot path
// -- fm_model::predict + fm_SGD over a data set -- to the MI355X library through its C-ABI.  Everything else
// (CLI, Data loading, fm_model, rlog, -out / -save_model) stays the reference's own code.
//
// It must be included in the ONE translation unit that includes the reference headers (they define non-inline
// functions, src/libfm/libfm.cpp:49-57), after fm_learn_sgd.h.  See INTEGRATION.md for the 8-line patch of main().
// It is compiled and exercised against the real reference classes by oracle/ref_harness.cpp (mode "sgd_gpu").
#ifndef FM_LEARN_SGD_GPU_H_
#define FM_LEARN_SGD_GPU_H_

#include <vector>
#include <string>
#include "fmx.h"

class fm_learn_sgd_gpu : public fm_learn_sgd {
 public:
  // GPU-only knobs (defaults = library defaults); everything else is inherited and set by main() as before
  int gpu_mode;          // FMX_SGD_SEQUENTIAL | FMX_SGD_MINIBATCH | FMX_SGD_HOGWILD
  int gpu_apply;         // FMX_APPLY_*
  uint gpu_batch, gpu_w0_chunk;
  int gpu_device;

  fm_learn_sgd_gpu() : gpu_mode(FMX_SGD_MINIBATCH), gpu_apply(FMX_APPLY_DEFAULT), gpu_batch(0), gpu_w0_chunk(0),
                       gpu_device(-1), h(NULL), n_slots(0) {}
  virtual ~fm_learn_sgd_gpu() { if (h) fmx_destroy(h); }

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
    opts.w0_chunk = gpu_w0_chunk; opts.flags = 0; opts.reserved = 0;
    for (int i = 0; i < num_iter; i++) {
      fmx_epoch_stats st;
      check(fmx_sgd_epoch(h, s_train, &opts, &st));
      double rmse_train = evaluate_slot(s_train);
      double rmse_test = evaluate_slot(s_test);
      std::cout << "#Iter=" << std::setw(3) << i << "\tTrain=" << rmse_train << "\tTest=" << rmse_test << std::endl;
      if (log != NULL) {
        log->log("rmse_train", rmse_train);
        log->log("time_learn", st.device_seconds);
        log->newLine();
      }
    }
    // main() reads fm afterwards (evaluate, -out, -save_model: libfm.cpp:418-434): bring the parameters home
    check(fmx_get_params(h, &fm->w0, fm->w.value, fm->num_factor > 0 ? fm->v.value[0] : NULL));
  }

  virtual double evaluate(Data& data) {                   // fm_learn::evaluate (fm_learn.h:93-153)
    open();
    return evaluate_slot(slot_of(data));
  }

  virtual void predict(Data& data, DVector<double>& out) {   // fm_learn_sgd::predict (fm_learn_sgd.h:76-90)
    assert(data.data->getNumRows() == out.dim);
    open();
    check(fmx_predict(h, slot_of(data), out.value));
    for (uint i = 0; i < out.dim; i++) {
      double p = out(i);
      if (task == TASK_REGRESSION) {
        p = std::min(max_target, p);
        p = std::max(min_target, p);
      } else if (task == TASK_CLASSIFICATION) {
        p = 1.0 / (1.0 + exp(-p));
      } else {
        throw "task not supported";
      }
      out(i) = p;
    }
  }

 protected:
  fmx_handle h;
  std::vector<Data*> slots;
  int n_slots;

  void check(int rc) { if (rc != FMX_OK) throw std::string(fmx_last_error(h)); }

  void open() {                                            // once: device context + parameters (fm_model.h:46-48)
    if (h) return;
    fmx_config c;
    c.num_attribute = fm->num_attribute; c.num_factor = fm->num_factor; c.k0 = fm->k0; c.k1 = fm->k1;
    c.task = task; c.reg0 = fm->reg0; c.regw = fm->regw; c.regv = fm->regv; c.learn_rate = learn_rate;
    c.min_target = min_target; c.max_target = max_target; c.device = gpu_device;
    c.shard_rank = 0; c.shard_world = 1; c.reserved = 0;
    if (fmx_create(&c, &h) != FMX_OK) throw std::string(fmx_last_error(NULL));
    check(fmx_set_params(h, fm->w0, fm->w.value, fm->num_factor > 0 ? fm->v.value[0] : NULL));
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
    check(fmx_upload_rows(h, (int)slots.size(), ent.empty() ? NULL : &ent[0], (const uint64_t*)&row_ptr[0],
                          d.target.value, d.data->getNumRows(), ent.size()));
    slots.push_back(&d);
    return (int)slots.size() - 1;
  }

  double evaluate_slot(int s) {
    fmx_eval ev;
    check(fmx_evaluate(h, s, &ev));
    if (log != NULL) {                                      // same rlog fields as fm_learn.h:124-127,146-150
      if (task == TASK_REGRESSION) { log->log("rmse", ev.rmse); log->log("mae", ev.mae); }
      else { log->log("accuracy", ev.accuracy); }
      log->log("time_pred", ev.device_seconds);
    }
    return task == TASK_REGRESSION ? ev.rmse : ev.accuracy;
  }
};

#endif /* FM_LEARN_SGD_GPU_H_ */

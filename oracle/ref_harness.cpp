// ref_harness.cpp -- drives the REAL reference (srendle/libfm) classes and dumps full-precision state.
//
// TEST INFRASTRUCTURE ONLY.  This translation unit is ours; it #includes the reference headers where
// they lie (-I/root/reference/src, never copied) and is compiled by oracle/Makefile into
// oracle/_ref/ref_harness.  It exists because the stock binary only prints 6 significant digits
// (matrix.h:332-342, fm_model.h:132-154); the parity pin needs raw doubles.
//
// The reference defines non-inline functions in its headers, so everything lives in this one TU
// (same constraint as src/libfm/libfm.cpp:49-57).
//
// usage:
//   ref_harness sgd  <train> <test> <task r|c> <k0> <k1> <k> <iters> <lr> <reg0> <regw> <regv> <init_stdev> <seed> <out_prefix>
//   ref_harness sgd_gpu <same arguments as sgd> [mode 0|1|2] [batch] [w0_chunk]
//                    (same driver, but the learner is adapter/fm_learn_sgd_gpu.h -> libfmx.so; needs a GPU;
//                     env FMX_GPU_DEVICES="0,0" | "0,1,..": one feature shard per listed device, FMX_GPU_APPLY / FMX_GPU_FLAGS /
//                     FMX_GPU_BIAS_LAG: fmx_sgd_opts fields)
//   ref_harness sgda <same arguments as sgd> <validation>   (fm_learn_sgd_element_adapt_reg; also dumps .reg.txt: reg_w, reg_v[f])
//   ref_harness sgda_gpu <same arguments as sgda>           (the learner is fm_learn_sgda_gpu of adapter/fm_learn_sgd_gpu.h; needs a GPU)
//   ref_harness mcmc_gpu <same arguments as mcmc>           (Gibbs sampling through adapter/fm_learn_mcmc_gpu.h; needs a GPU)
//   ref_harness als_gpu <same arguments as als>             (the learner is adapter/fm_learn_mcmc_gpu.h -> libfmx.so; needs a GPU)
//   ref_harness als  <train> <test> <task r|c> <k0> <k1> <k> <iters> <reg0> <regw> <regv> <init_stdev> <seed> <out_prefix>
//   ref_harness mcmc <train> <test> <task r|c> <k0> <k1> <k> <iters> <init_stdev> <seed> <out_prefix>
//   ref_harness time_sgd <n> <k> <nnz> <rows> <seed>        (CPU baseline: reference fm_model::predict + fm_SGD on
//                                                             the synthetic workload of oracle/fm_oracle.c, 1 thread)
//   ref_harness time_sgd_rows <rows.bin> <n> <k> <lr> <regv> <init_stdev>   (the same loop over rows from a file)
// environment: FMX_META=<file>  attribute groups (`-meta`, one group id per line, Data.h:85-97);
//              FMX_GROUP_REG=w_1,..,w_G,v_1,..,v_G  per-group lambdas for als (the tail of `-regular`, libfm.cpp:353-363)
//              FMX_RELATIONS=<prefix>[,<prefix>]  block-structured data for als / mcmc (`-relation`, libfm.cpp:172-196)
// outputs (<out_prefix>.*):
//   .init.bin / .final.bin : magic 'FMXP', u64 n, i32 k, f64 w0, f64 w[n], f64 v[k][n]  (reference layout)
//   .pred_raw.bin          : f64[num_test]  fm_learn::predict_case per test row after training
//   .pred_out.bin          : f64[num_test]  fml->predict(test, pred)   (what -out writes)
//   .eval.txt              : one line per epoch: evaluate(train) evaluate(test) with %.17g
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>
#include <iterator>
#include <algorithm>
#include <iomanip>
#include <sys/time.h>
#include "util/util.h"
#include "util/cmdline.h"
#include "fm_core/fm_model.h"
#include "libfm/src/Data.h"
#include "libfm/src/relation.h"
#include "libfm/src/fm_learn.h"
#include "libfm/src/fm_learn_sgd.h"
#include "libfm/src/fm_learn_sgd_element.h"
#include "libfm/src/fm_learn_sgd_element_adapt_reg.h"
#include "libfm/src/fm_learn_mcmc_simultaneous.h"

#include "fm_oracle.h"   // only for the synthetic-row generator used by time_sgd
#ifdef FMX_WITH_GPU_ADAPTER
#include "../adapter/fm_learn_sgd_gpu.h"   // the reference-side binding of libfmx, exercised by mode sgd_gpu
#include "../adapter/fm_learn_mcmc_gpu.h"  // ... and the ALS binding, exercised by mode als_gpu
#endif

static void dump_params(const std::string& path, fm_model& fm) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { perror(path.c_str()); exit(2); }
  const char magic[4] = {'F', 'M', 'X', 'P'};
  unsigned long long n = fm.num_attribute;
  int k = fm.num_factor;
  fwrite(magic, 1, 4, f);
  fwrite(&n, sizeof(n), 1, f);
  fwrite(&k, sizeof(k), 1, f);
  fwrite(&fm.w0, sizeof(double), 1, f);
  fwrite(fm.w.value, sizeof(double), n, f);
  for (int i = 0; i < k; i++) fwrite(fm.v.value[i], sizeof(double), n, f);
  fclose(f);
}

static void dump_vec(const std::string& path, const double* p, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { perror(path.c_str()); exit(2); }
  fwrite(p, sizeof(double), n, f);
  fclose(f);
}

// exposes the protected per-row predictor of the learner base class
template <class L> struct Open : public L {
  double raw(Data& d) { return this->predict_case(d); }
};

static void set_task(fm_learn* fml, const std::string& task, Data& train, Data& test) {
  // src/libfm/libfm.cpp:298-309
  if (task == "r") {
    fml->task = 0;
  } else {
    fml->task = 1;
    for (uint i = 0; i < train.target.dim; i++) { if (train.target(i) <= 0.0) { train.target(i) = -1.0; } else { train.target(i) = 1.0; } }
    for (uint i = 0; i < test.target.dim; i++) { if (test.target(i) <= 0.0) { test.target(i) = -1.0; } else { test.target(i) = 1.0; } }
  }
}

static double wall() { struct timeval tv; gettimeofday(&tv, 0); return tv.tv_sec + 1e-6 * tv.tv_usec; }

static int run_time_sgd(int argc, char** argv) {
  if (argc < 7) { fprintf(stderr, "time_sgd <n> <k> <nnz> <rows> <seed>\n"); return 2; }
  unsigned long long n = strtoull(argv[2], 0, 10);
  int k = atoi(argv[3]); uint nnz = atoi(argv[4]); uint rows = atoi(argv[5]);
  unsigned long long seed = strtoull(argv[6], 0, 10);
  if ((unsigned long long)k * n >= (1ULL << 32)) {
    fprintf(stderr, "k*n >= 2^32: the stock reference cannot allocate this (matrix.h:167-169)\n");
    return 3;
  }
  fm_model fm;
  fm.num_attribute = (uint)n; fm.num_factor = k; fm.k0 = true; fm.k1 = true;
  fm.init_stdev = 0; fm.init_mean = 0;       // ran_gaussian short-circuits (random.h:164-166): fast fill
  fm.init();
  fm.regv = 0.001;
  for (int f = 0; f < k; f++) for (unsigned long long j = 0; j < n; j += 1) fm.v.value[f][j] = 0.01 * (((j * 2654435761ULL + f * 40503ULL) & 1023) / 512.0 - 1.0);
  std::vector<fmo_entry> ent((size_t)rows * nnz); std::vector<uint64_t> rp(rows + 1); std::vector<float> y(rows);
  fmo_synth_rows(seed, 0, rows, nnz, n, ent.data(), rp.data(), y.data());
  DVector<double> sum, sum_sqr; sum.setSize(k); sum_sqr.setSize(k);
  double t0 = wall();
  for (uint r = 0; r < rows; r++) {
    sparse_row<FM_FLOAT> row; row.data = (sparse_entry<FM_FLOAT>*)&ent[(size_t)r * nnz]; row.size = nnz;
    double p = fm.predict(row, sum, sum_sqr);
    double mult = -y[r] * (1.0 - 1.0 / (1.0 + exp(-y[r] * p)));
    fm_SGD(&fm, 0.01, row, mult, sum);
  }
  double t1 = wall();
  printf("{\"rows\": %u, \"seconds\": %.6f, \"examples_per_sec\": %.3f, \"w0\": %.17g}\n", rows, t1 - t0, rows / (t1 - t0), fm.w0);
  return 0;
}

// time_sgd_rows <rows.bin> <n> <k> <lr> <regv> <init_stdev>: the same loop over rows handed over in a file (bench.py: the
// Criteo-shaped rows of BASELINE configs[2], copied back from the device).  File: u32 n_rows, u32 pad, u64 nnz,
// u64 row_ptr[n_rows + 1], {u32 id; f32 value}[nnz], f32 target[n_rows].
static int run_time_sgd_rows(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "time_sgd_rows <rows.bin> <n> <k> <lr> <regv> <init_stdev>\n"); return 2; }
  unsigned long long n = strtoull(argv[3], 0, 10);
  int k = atoi(argv[4]);
  const double lr = atof(argv[5]), regv = atof(argv[6]), stdev = atof(argv[7]);
  if ((unsigned long long)k * n >= (1ULL << 32)) { fprintf(stderr, "k*n >= 2^32: the stock reference cannot allocate this (matrix.h:167-169)\n"); return 3; }
  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 2; }
  uint32_t hdr[2]; uint64_t nnz = 0;
  if (fread(hdr, 4, 2, f) != 2 || fread(&nnz, 8, 1, f) != 1) { fprintf(stderr, "short file\n"); return 2; }
  const uint32_t rows = hdr[0];
  std::vector<uint64_t> rp((size_t)rows + 1); std::vector<fmo_entry> ent(nnz); std::vector<float> y(rows);
  if (fread(rp.data(), 8, rp.size(), f) != rp.size() || fread(ent.data(), 8, nnz, f) != nnz || fread(y.data(), 4, rows, f) != rows) { fprintf(stderr, "short file\n"); return 2; }
  fclose(f);
  fm_model fm;
  fm.num_attribute = (uint)n; fm.num_factor = k; fm.k0 = true; fm.k1 = true;
  fm.init_stdev = 0; fm.init_mean = 0;
  fm.init();
  fm.regv = regv;
  for (int q = 0; q < k; q++) for (unsigned long long j = 0; j < n; j += 1) fm.v.value[q][j] = stdev * (((j * 2654435761ULL + q * 40503ULL) & 1023) / 512.0 - 1.0);
  DVector<double> sum, sum_sqr; sum.setSize(k); sum_sqr.setSize(k);
  double t0 = wall();
  for (uint r = 0; r < rows; r++) {
    sparse_row<FM_FLOAT> row; row.data = (sparse_entry<FM_FLOAT>*)&ent[rp[r]]; row.size = (uint)(rp[r + 1] - rp[r]);
    double p = fm.predict(row, sum, sum_sqr);
    double mult = -y[r] * (1.0 - 1.0 / (1.0 + exp(-y[r] * p)));           // fm_learn_sgd_element.h:61-65
    fm_SGD(&fm, lr, row, mult, sum);
  }
  double t1 = wall();
  printf("{\"rows\": %u, \"seconds\": %.6f, \"examples_per_sec\": %.3f, \"w0\": %.17g}\n", rows, t1 - t0, rows / (t1 - t0), fm.w0);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "see header of ref_harness.cpp\n"); return 2; }
  std::string mode = argv[1];
  if (mode == "time_sgd") return run_time_sgd(argc, argv);
  if (mode == "time_sgd_rows") return run_time_sgd_rows(argc, argv);
  try {
    int a = 2;
    std::string train_file = argv[a++], test_file = argv[a++], task = argv[a++];
    int k0 = atoi(argv[a++]), k1 = atoi(argv[a++]), k = atoi(argv[a++]);
    int iters = atoi(argv[a++]);
    double lr = 0, reg0 = 0, regw = 0, regv = 0;
    if (mode == "sgd" || mode == "sgd_gpu" || mode == "sgda" || mode == "sgda_gpu") lr = atof(argv[a++]);
    if (mode != "mcmc" && mode != "mcmc_gpu") { reg0 = atof(argv[a++]); regw = atof(argv[a++]); regv = atof(argv[a++]); }
    const bool als_gpu = (mode == "als_gpu") || (mode == "mcmc_gpu");     // both through adapter/fm_learn_mcmc_gpu.h
    if (mode == "als_gpu") mode = "als";
    if (mode == "mcmc_gpu") mode = "mcmc";
    const bool sgda_gpu = (mode == "sgda_gpu");
    if (sgda_gpu) mode = "sgda";
    double init_stdev = atof(argv[a++]);
    long seed = atol(argv[a++]);
    std::string prefix = argv[a++];

    srand(seed);                                               // libfm.cpp:115-116
    const bool is_sgd = (mode == "sgd" || mode == "sgd_gpu" || mode == "sgda");
    Data train(0, is_sgd, !is_sgd);                            // libfm.cpp:143-148
    train.load(train_file);
    Data test(0, is_sgd, !is_sgd);
    test.load(test_file);
    // relations (block structure, `-relation a,b`; libfm.cpp:172-196): FMX_RELATIONS=<prefix>[,<prefix>...]; every
    // prefix names <prefix>.x / .xt (binary design matrix of the block), <prefix>.train / .test (main row -> block row)
    // and optionally <prefix>.groups
    std::vector<std::string> rel;
    if (getenv("FMX_RELATIONS")) { std::string t = getenv("FMX_RELATIONS"); size_t p0 = 0; while (p0 < t.size()) { size_t p1 = t.find(',', p0); if (p1 == std::string::npos) p1 = t.size(); rel.push_back(t.substr(p0, p1 - p0)); p0 = p1 + 1; } }
    DVector<RelationData*> relation;
    relation.setSize(rel.size());
    train.relation.setSize(rel.size()); test.relation.setSize(rel.size());
    for (uint i = 0; i < rel.size(); i++) {
      relation(i) = new RelationData(0, is_sgd, !is_sgd);
      relation(i)->load(rel[i]);
      train.relation(i).data = relation(i);
      test.relation(i).data = relation(i);
      train.relation(i).load(rel[i] + ".train", train.num_cases);
      test.relation(i).load(rel[i] + ".test", test.num_cases);
    }
    uint num_all_attribute = std::max(train.num_feature, test.num_feature);   // libfm.cpp:203
    DataMetaInfo meta_main(num_all_attribute);
    if (getenv("FMX_META")) meta_main.loadGroupsFromFile(getenv("FMX_META"));  // -meta, libfm.cpp:207-210
    for (uint r = 0; r < train.relation.dim; r++) {                            // block attributes follow the main ones (:213-216)
      train.relation(r).data->attr_offset = num_all_attribute;
      num_all_attribute += train.relation(r).data->num_feature;
    }
    DataMetaInfo meta(num_all_attribute);                                      // the joined table (:217-242)
    {
      meta.num_attr_groups = meta_main.num_attr_groups;
      for (uint r = 0; r < relation.dim; r++) meta.num_attr_groups += relation(r)->meta->num_attr_groups;
      meta.num_attr_per_group.setSize(meta.num_attr_groups);
      meta.num_attr_per_group.init(0);
      uint at = 0, gr = 0;
      for (uint i = 0; i < meta_main.attr_group.dim; i++, at++) { meta.attr_group(at) = meta_main.attr_group(i); meta.num_attr_per_group(meta.attr_group(at))++; }
      gr = meta_main.num_attr_groups;
      for (uint r = 0; r < relation.dim; r++) {
        for (uint i = 0; i < relation(r)->meta->attr_group.dim; i++, at++) { meta.attr_group(at) = gr + relation(r)->meta->attr_group(i); meta.num_attr_per_group(meta.attr_group(at))++; }
        gr += relation(r)->meta->num_attr_groups;
      }
    }
    meta.num_relations = train.relation.dim;

    fm_model fm;                                               // libfm.cpp:245-258
    fm.num_attribute = num_all_attribute;
    fm.init_stdev = init_stdev;
    fm.k0 = k0 != 0; fm.k1 = k1 != 0; fm.num_factor = k;
    fm.init();

    FILE* ev = fopen((prefix + ".eval.txt").c_str(), "w");
#ifdef FMX_WITH_GPU_ADAPTER
    if (mode == "sgd_gpu") {
      fm_learn_sgd_gpu* fml = new fm_learn_sgd_gpu();
      if (argc > a) fml->gpu_mode = atoi(argv[a++]);
      if (argc > a) fml->gpu_batch = atoi(argv[a++]);
      if (argc > a) fml->gpu_w0_chunk = atoi(argv[a++]);
      if (const char* dv = getenv("FMX_GPU_DEVICES")) {      // e.g. "0,1,2,3" (RCCL) or "0,0" (two shards on one device)
        for (const char* p = dv; *p;) { fml->gpu_devices.push_back(atoi(p)); while (*p && *p != ',') p++; if (*p) p++; }
      }
      if (const char* e = getenv("FMX_GPU_APPLY")) fml->gpu_apply = atoi(e);
      if (const char* e = getenv("FMX_GPU_FLAGS")) fml->gpu_flags = (uint)atoi(e);
      if (const char* e = getenv("FMX_GPU_BIAS_LAG")) fml->gpu_bias_lag = (uint)atoi(e);
      fml->num_iter = iters;
      fml->fm = &fm; fml->max_target = train.max_target; fml->min_target = train.min_target; fml->meta = &meta;
      set_task(fml, task, train, test);
      fml->log = NULL;
      fml->init();
      fm.reg0 = reg0; fm.regw = regw; fm.regv = regv;
      fml->learn_rate = lr; fml->learn_rates.init(lr);
      dump_params(prefix + ".init.bin", fm);
      fml->learn(train, test);
      fprintf(ev, "%.17g %.17g\n", fml->evaluate(train), fml->evaluate(test));
      dump_params(prefix + ".final.bin", fm);
      DVector<double> pred; pred.setSize(test.num_cases);
      fml->predict(test, pred);
      dump_vec(prefix + ".pred_out.bin", pred.value, pred.dim);
    } else
#endif
    if (mode == "sgda") {
      std::string val_file = argv[a++];
      Data validation(0, true, false);
      validation.load(val_file);
      validation.relation.setSize(0);
      fm_learn_sgd_element_adapt_reg* fml;
#ifdef FMX_WITH_GPU_ADAPTER
      if (sgda_gpu) {
        fm_learn_sgda_gpu* gl = new fm_learn_sgda_gpu();
        if (const char* e = getenv("FMX_GPU_BATCH")) gl->gpu_batch = (uint)atoi(e);            // > 0: the batch form of the learner
        if (const char* e = getenv("FMX_GPU_W0_CHUNK")) gl->gpu_w0_chunk = (uint)atoi(e);
        fml = gl;
      } else
#endif
      fml = new fm_learn_sgd_element_adapt_reg();
      fml->num_iter = iters;
      fml->validation = &validation;                           // libfm.cpp:279
      fml->fm = &fm; fml->max_target = train.max_target; fml->min_target = train.min_target; fml->meta = &meta;
      set_task(fml, task, train, test);
      if (task != "r") for (uint i = 0; i < validation.target.dim; i++) { if (validation.target(i) <= 0.0) { validation.target(i) = -1.0; } else { validation.target(i) = 1.0; } }
      fml->log = NULL;
      fml->init();
      fm.reg0 = reg0; fm.regw = regw; fm.regv = regv;
      fml->learn_rate = lr; fml->learn_rates.init(lr);
      dump_params(prefix + ".init.bin", fm);
      fml->learn(train, test);
      fprintf(ev, "%.17g %.17g\n", fml->evaluate(train), fml->evaluate(test));
      dump_params(prefix + ".final.bin", fm);
      DVector<double> pred; pred.setSize(test.num_cases);
      fml->predict(test, pred);
      dump_vec(prefix + ".pred_out.bin", pred.value, pred.dim);
      FILE* rf = fopen((prefix + ".reg.txt").c_str(), "w");
      for (uint g = 0; g < meta.num_attr_groups; g++) {          // [G][1+k]: reg_w(g), reg_v(g,0..k-1)
        fprintf(rf, "%.17g\n", fml->reg_w(g));
        for (int f = 0; f < k; f++) fprintf(rf, "%.17g\n", fml->reg_v(g, f));
      }
      fclose(rf);
    } else
    if (is_sgd) {
      Open<fm_learn_sgd_element>* fml = new Open<fm_learn_sgd_element>();
      fml->num_iter = 1;
      fml->fm = &fm; fml->max_target = train.max_target; fml->min_target = train.min_target; fml->meta = &meta;
      set_task(fml, task, train, test);
      fml->log = NULL;
      fml->init();
      fm.reg0 = reg0; fm.regw = regw; fm.regv = regv;         // libfm.cpp:366-385
      fml->learn_rate = lr; fml->learn_rates.init(lr);        // libfm.cpp:391-395
      dump_params(prefix + ".init.bin", fm);
      for (int it = 0; it < iters; it++) {
        fml->learn(train, test);                               // one epoch of fm_learn_sgd_element.h:53-67
        fprintf(ev, "%.17g %.17g\n", fml->evaluate(train), fml->evaluate(test));
      }
      dump_params(prefix + ".final.bin", fm);
      std::vector<double> raw(test.num_cases);
      for (test.data->begin(); !test.data->end(); test.data->next()) raw[test.data->getRowIndex()] = fml->raw(test);
      dump_vec(prefix + ".pred_raw.bin", raw.data(), raw.size());
      DVector<double> pred; pred.setSize(test.num_cases);
      fml->predict(test, pred);
      dump_vec(prefix + ".pred_out.bin", pred.value, pred.dim);
    } else {
      const bool is_als = (mode == "als");
      fm.w.init_normal(fm.init_mean, fm.init_stdev);          // libfm.cpp:283
      fm_learn_mcmc* fml;
#ifdef FMX_WITH_GPU_ADAPTER
      if (als_gpu) {
        fm_learn_als_gpu* gl = new fm_learn_als_gpu();
        if (const char* dv = getenv("FMX_GPU_DEVICES"))        // one feature shard per listed device ("0,0": two on one GPU)
          for (const char* p = dv; *p;) { gl->gpu_devices.push_back(atoi(p)); while (*p && *p != ',') p++; if (*p) p++; }
        if (const char* e = getenv("FMX_GPU_BLOCKS")) gl->gpu_blocks_expand = !strcmp(e, "expand");   // default: blocks kept apart
        fml = gl;
      } else
#endif
      fml = new fm_learn_mcmc_simultaneous();
      fml->validation = NULL;
      fml->num_iter = iters;
      fml->num_eval_cases = test.num_cases;
      if (const char* e = getenv("FMX_NUM_EVAL_CASES")) fml->num_eval_cases = (uint)atoi(e);   // -num_eval_cases, libfm.cpp:286
      fml->do_sample = !is_als;                                // libfm.cpp:135-139
      fml->do_multilevel = !is_als;
      fml->fm = &fm; fml->max_target = train.max_target; fml->min_target = train.min_target; fml->meta = &meta;
      set_task(fml, task, train, test);
      fml->log = NULL;
      RLog* rlog = NULL;                                       // -rlog, libfm.cpp:311-324: the learner registers its fields in init()
      if (const char* e = getenv("FMX_RLOG")) { rlog = new RLog(new std::ofstream(e)); fml->log = rlog; }
      fml->init();
      if (rlog) rlog->init();                                  // libfm.cpp:405-407: the header line
      fm.reg0 = reg0; fm.regw = regw; fm.regv = regv;         // libfm.cpp:346-352
      fml->w_lambda.init(fm.regw); fml->v_lambda.init(fm.regv);
      if (getenv("FMX_GROUP_REG")) {                           // -regular 'r0,w_1..w_G,v_1..v_G', libfm.cpp:353-363
        std::vector<double> reg; { std::string t = getenv("FMX_GROUP_REG"); size_t p0 = 0; while (p0 < t.size()) { size_t p1 = t.find(',', p0); if (p1 == std::string::npos) p1 = t.size(); reg.push_back(atof(t.substr(p0, p1 - p0).c_str())); p0 = p1 + 1; } }
        if (reg.size() != 2 * meta.num_attr_groups) { fprintf(stderr, "FMX_GROUP_REG needs 2*G values\n"); return 2; }
        fm.regw = 0.0; fm.regv = 0.0;
        int j = 0;
        for (uint g = 0; g < meta.num_attr_groups; g++) fml->w_lambda(g) = reg[j++];
        for (uint g = 0; g < meta.num_attr_groups; g++) { for (int f = 0; f < fm.num_factor; f++) fml->v_lambda(g, f) = reg[j]; j++; }
      }
      dump_params(prefix + ".init.bin", fm);
      const double t_learn0 = wall();
      fml->learn(train, test);
      // (bench.py's cpu_baseline of the als / mcmc extra keys: the reference's own fm_learn_mcmc::learn, one thread)
      printf("{\"learn_seconds\": %.6f, \"iters\": %d, \"rows\": %u}\n", wall() - t_learn0, iters, (unsigned)train.num_cases);
      dump_params(prefix + ".final.bin", fm);
      DVector<double> pred; pred.setSize(test.num_cases);
      fml->predict(test, pred);                                // fm_learn_mcmc.h:380-404
      dump_vec(prefix + ".pred_out.bin", pred.value, pred.dim);
    }
    fclose(ev);
  } catch (std::string& e) {
    std::cerr << "ERROR: " << e << std::endl; return 1;
  } catch (char const*& e) {
    std::cerr << "ERROR: " << e << std::endl; return 1;
  }
  return 0;
}

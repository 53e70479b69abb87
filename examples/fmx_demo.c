/* fmx_demo.c -- the C-ABI of libfmx.so used from plain C (no Python, no torch, no C++):
 *   gcc -O2 -Iinclude examples/fmx_demo.c -Llibfm_amd -lfmx -Wl,-rpath,$PWD/libfm_amd -Wl,-rpath-link,/opt/rocm/lib -lm -o fmx_demo
 * Builds a small synthetic classification problem on the device, runs the three SGD modes and one ALS sweep and
 * prints the metrics; exits non-zero (with fmx_last_error) on any failure -- there is no CPU path to fall back to.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmx.h"

#define CHK(h, call) do { int rc_ = (call); if (rc_ != FMX_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, fmx_last_error(h)); return 1; } } while (0)

int main(void) {
  fmx_config c;
  memset(&c, 0, sizeof(c));
  c.num_attribute = 640000; c.num_factor = 16; c.k0 = 1; c.k1 = 1; c.task = FMX_TASK_CLASSIFICATION;
  c.regv = 0.001; c.learn_rate = 0.02; c.min_target = -1; c.max_target = 1; c.device = -1; c.shard_world = 1;
  fmx_handle h = NULL;
  CHK(NULL, fmx_create(&c, &h));
  fmx_info info;
  CHK(h, fmx_get_info(h, &info));
  printf("device %s (%s), %llu features x %d padded factors, %.1f MB of parameters\n", info.device_name, info.arch,
         (unsigned long long)info.n_local, info.k_padded, info.bytes_params / 1e6);
  const int modes[3] = {FMX_SGD_SEQUENTIAL, FMX_SGD_MINIBATCH, FMX_SGD_HOGWILD};
  const char *names[3] = {"sequential", "minibatch", "hogwild"};
  for (int m = 0; m < 3; m++) {
    CHK(h, fmx_init_params(h, 0.0, 0.05, 1));
    CHK(h, fmx_synth_rows(h, 0, 123, 0, m == 0 ? 20000 : 200000, 16));
    fmx_sgd_opts o; memset(&o, 0, sizeof(o));
    o.mode = modes[m]; o.batch = m == 1 ? 4096 : 0;
    fmx_epoch_stats st; fmx_eval ev;
    for (int it = 0; it < 3; it++) CHK(h, fmx_sgd_epoch(h, 0, &o, &st));
    CHK(h, fmx_evaluate(h, 0, &ev));
    printf("%-10s 3 epochs over %llu rows: train accuracy %.4f, last epoch %.3f ms\n", names[m],
           (unsigned long long)st.rows, ev.accuracy, st.device_seconds * 1e3);
    if (!(ev.accuracy > 0.6)) { fprintf(stderr, "mode %s did not learn\n", names[m]); return 2; }
  }
  CHK(h, fmx_init_params(h, 0.0, 0.05, 1));
  CHK(h, fmx_synth_rows(h, 0, 123, 0, 200000, 16));
  CHK(h, fmx_als_begin(h, 0));
  fmx_als_opts ao; memset(&ao, 0, sizeof(ao));
  ao.alpha = 1.0; ao.w_lambda = 1.0; ao.v_lambda = 10.0;
  fmx_als_stats as;
  for (int it = 0; it < 2; it++) CHK(h, fmx_als_sweep(h, &ao, &as));
  CHK(h, fmx_als_end(h));
  printf("als        2 sweeps: train accuracy %.4f, %u levels, last sweep %.3f ms\n", as.train_metric, as.levels, as.device_seconds * 1e3);
  CHK(h, fmx_destroy(h));
  printf("ok\n");
  return 0;
}

/*
 * fmx.h -- C-ABI of the MI355X-native Factorization Machine hot path (libfmx.so).
 *
 * This is the drop-in boundary for srendle/libfm's learner interface.  libFM has no plugin/FFI API;
 * its de-facto operator interface is `class fm_learn` (/root/reference/src/libfm/src/fm_learn.h:31-60)
 * driven by main() (/root/reference/src/libfm/libfm.cpp:271-434).  Each entry point below names the
 * reference interface it replaces.  The reference-side bindings (fm_learn subclasses calling these
 * functions) are shown in INTEGRATION.md and live in adapter/fm_learn_sgd_gpu.h (-method sgd, sgda) and
 * adapter/fm_learn_mcmc_gpu.h (-method als, mcmc); examples/fmx_demo.c uses the library from plain C.
 *
 * Conventions
 *   - plain C: opaque handle, POD structs, raw pointers and sizes; no C++/torch types cross the boundary.
 *   - every call returns FMX_OK (0) or a negative FMX_E_* code; fmx_last_error(h) gives the text.
 *     (The reference throws std::string / const char* caught in main, libfm.cpp:436-440; the adapter
 *      re-throws the text so that behaviour is preserved.)
 *   - host parameter layout is the reference's: w0 double, w[n] double, v[k][n] double FACTOR-major
 *     (fm_model.h:46-48, matrix.h:165-170: fm->v.value[0] is one contiguous k*n block).
 *   - host row layout is the reference's: one contiguous array of sparse_entry<float> {uint32 id; float value}
 *     in row order (fmatrix.h:34-42; Data.h:237-270 allocates exactly this) plus row offsets.
 *   - device layout (inside the library): V is FEATURE-major fp32, rows of k rounded up to 16 floats (one 64-byte
 *     sector; the lanes of a wavefront are mapped on the next power of two, those beyond the row take no part),
 *     so one gathered row is one coalesced segment (256 B at k=64, 448 B at k=100).  See DESIGN.md section 2.
 *   - single calling thread per handle (the reference is single threaded).
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with FMX_E_HIP.
 */
#ifndef FMX_H_
#define FMX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMX_ABI_VERSION 7

enum {
  FMX_OK = 0,
  FMX_E_ARG = -1,       /* bad argument */
  FMX_E_HIP = -2,       /* HIP runtime error (no device, out of memory, launch failure) */
  FMX_E_STATE = -3,     /* call order violated (e.g. slot not uploaded) */
  FMX_E_UNSUPPORTED = -4
};

enum { FMX_TASK_REGRESSION = 0, FMX_TASK_CLASSIFICATION = 1 };   /* fm_learn.h:45-47 */

/* SGD update modes */
enum {
  FMX_SGD_SEQUENTIAL = 0,  /* batch = 1, rows in storage order: the reference trajectory
                              (fm_learn_sgd_element.h:56-67), exact to 1e-4 against the reference's final parameters.
                              Two forms (libfm_amd/csrc/fmx_seq_kernels.h), chosen per slot:
                              - conflict-free runs (FMX_STAT_SEQ_RUNS), where consecutive rows rarely share a feature (the
                                slot's runs average >= 16 rows): the slot is cut once into maximal runs of consecutive rows
                                that share no feature; inside a run the online loop IS one batch step with the bias coupled
                                example by example, and a run is one launch.  BASELINE's headline shape: ~26 M examples/s,
                                ~1000x the reference on one core (24-27 k examples/s);
                              - one example at a time on eight wavefronts (k <= 128) otherwise: ~360 k examples/s whatever
                                the shape -- 14x the reference where its model misses the caches, 30x SLOWER than it where
                                the model is cache-resident (configs[0]: 11 M examples/s on the CPU).
                              The parity mode; the adapters default to MINIBATCH. */
  FMX_SGD_MINIBATCH = 1,   /* restated batch rule (oracle/fm_oracle.h fmo_sgd_epoch_minibatch):
                              partial sums -> [all-reduce] -> w0 micro-chunks + multipliers -> scatter-add */
  FMX_SGD_HOGWILD = 2      /* fused single pass per example (gather, predict, update in registers);
                              asynchronous between wavefronts; single device only.  An update becomes visible when its wavefront
                              retires, so the ~5 000 rows in flight act like a batch: on rows with frequent features (collision mass C,
                              see fmx_sgd_opts::batch) it diverges where learn_rate * curvature * 5 120 * C > 2 -- fmx_epoch_stats
                              reports the gain (batch_used = rows in flight) and FMX_STAT_UNSTABLE, FMX_FLAG_REJECT_UNSTABLE refuses */
};

enum {
  FMX_APPLY_DEFAULT = 0,   /* MINIBATCH: SEGMENTED, HOGWILD: STORE */
  FMX_APPLY_ATOMIC = 1,    /* one wavefront per example, fp32 atomic scatter-add of the per-occurrence deltas
                              (reads of a feature that collides inside the batch may see a partial update) */
  FMX_APPLY_STORE = 2,     /* one wavefront per example, plain read-modify-write stores (colliding ids in a
                              batch lose updates; exact when a batch has no repeated feature) */
  FMX_APPLY_SEGMENTED = 3, /* MINIBATCH only: entries pre-bucketed per (batch, feature); one owner per touched
                              row, no atomics, exactly the batch rule of oracle/fm_oracle.h (deterministic) */
  FMX_APPLY_FUSED = 4      /* MINIBATCH only, implies FMX_FLAG_BIAS_LAG: the SAME batch rule in one pass over HBM.  The rule
                              takes every sum and gradient from batch-start parameters, so a feature that occurs once in its
                              batch is gathered, used and written back by its own example's wavefront (V read once, written
                              once -- the algorithmic minimum); the features that occur more than once in the batch are
                              finished by the segmented kernel from the factor sums their examples leave behind.
                              Deterministic; the result of FMX_APPLY_SEGMENTED with the same bias_lag to fp32 rounding. */
};

typedef struct fmx_context_s *fmx_handle;

/* replaces the fields main() writes into fm_model / fm_learn (libfm.cpp:245-256, 294-309, 366-404) */
typedef struct fmx_config {
  uint64_t num_attribute;   /* fm_model::num_attribute (global n)            fm_model.h:51 */
  int32_t  num_factor;      /* fm_model::num_factor (k), 0 .. 1024           fm_model.h:54 */
  int32_t  k0;              /* use bias                                       fm_model.h:53 */
  int32_t  k1;              /* use 1-way interactions                         fm_model.h:53 */
  int32_t  task;            /* fm_learn::task                                 fm_learn.h:45 */
  double   reg0, regw, regv;/* fm_model::reg0/regw/regv                       fm_model.h:56-57 */
  double   learn_rate;      /* fm_learn_sgd::learn_rate                       fm_learn_sgd.h:42 */
  double   min_target;      /* fm_learn::min_target (of the TRAIN set)        libfm.cpp:295-296 */
  double   max_target;
  int32_t  device;          /* HIP device ordinal; -1 = current device */
  int32_t  shard_rank;      /* feature sharding: this handle owns the features j with            */
  int32_t  shard_world;     /*   owner(j) == shard_rank  (shard_world = 1: unsharded)            */
  int32_t  shard_hash;      /* 0: owner(j) = j mod shard_world, local row j div shard_world.
                               1: owner(j) = p(j) mod shard_world, local row p(j) div shard_world with p a fixed pseudo-random
                                  PERMUTATION of [0, num_attribute) (4-round Feistel network, cycle-walked; north_star:
                                  "row-sharded by feature-id hash"): balanced whatever the structure of the ids, still a dense
                                  local table, and invertible, so the shard knows its features' global ids. */
  int32_t  place_candidates;/* placement of the parameter tables (see fmx_create).  0 = default: tables of >= 2 GiB are built from 1 GiB
                               chunks of TWO memory classes of the device, found by probing (transient pool of at most 3 tables' worth
                               of chunks + 64 GiB, never more than half of the free memory); tables of 256 MB .. 2 GiB: best of 2 candidate
                               allocations.  1 = first fit: plain allocations, no probe, no transient memory.  2 .. 6 = the bound of the
                               transient pool in tables' worth (big tables) / the number of candidates (small ones) */
  uint32_t als_split_min;   /* ALS / MCMC: dependency levels with at least this many entries update {e, q} as a row-ordered stream
                               (the split step) instead of inside the fused draw.  0 = library default (65536),
                               1 = every level, 0xFFFFFFFF = never */
  uint32_t exchange_runs;   /* several GPUs over RCCL: a batch's partial sums are exchanged in this many runs of rows, the all-reduce of
                               one run travelling while the next is summed.  0 = default (4), 1 = one all-reduce per batch */
  uint32_t exchange_algo;   /* several GPUs: how the per-batch exchange of the partial sums is carried.  FMX_EXCHANGE_ALLREDUCE (0): one
                               all-reduce (ncclAllReduce; RCCL picks ring / tree).  FMX_EXCHANGE_RS_AG (1): reduce-scatter + all-gather --
                               on a fully connected xGMI node every GPU reduces its 1/P slice over its P - 1 direct links and broadcasts it
                               back over the same links, instead of a ring that is bound by ONE link.  Pieces whose length
                               is not a multiple of the shard count fall back to the all-reduce.  Same sums either way (the loopback exchange
                               of shards sharing a device adds in the same order: bit-identical). */
} fmx_config;
#define FMX_EXCHANGE_ALLREDUCE 0u
#define FMX_EXCHANGE_RS_AG     1u

typedef struct fmx_sgd_opts {
  int32_t  mode;            /* FMX_SGD_* */
  int32_t  apply;           /* FMX_APPLY_* (MINIBATCH and HOGWILD) */
  uint32_t batch;           /* MINIBATCH: rows per minibatch.  0 = library default: 262144, CUT to the largest power of two that
                               keeps the batch rule stable on THIS data set: learn_rate * curvature * batch * C <= 1, C = the
                               collision mass of the slot's rows (fmx_sgd_batch_info).  The rule freezes every parameter for one
                               batch; two examples of a batch that share features push those in the same direction from the
                               same stale state, so the step along "what the rows have in common" is learn_rate * batch * C where
                               the reference (batch 1, fm_sgd.h:33-51) takes `batch` small steps.  Uniform ids over 1e8 features:
                               C = 1e-5, no cut; Criteo-shaped rows (100-id fields, Zipf heads): C = 1, batch 256 at
                               learn_rate 0.01; never below 32).  An explicit batch is honoured (fmx_epoch_stats::status reports FMX_STAT_UNSTABLE;
                               with FMX_FLAG_REJECT_UNSTABLE the call fails with FMX_E_ARG instead of training).
                               HOGWILD: rows per launch during which w0 is frozen (macro-batch), 0 = 262144 */
  uint32_t w0_chunk;        /* w0 micro-chunk of the bias recurrence; 0 = library default (fmx_default_w0_chunk): the largest power
                             * of two <= FMX_W0_CHUNK_CAP with learn_rate * chunk * curvature <= 1 (curvature 1 for regression, 1/4
                             * for classification).  The reference moves w0 after every example (fm_sgd.h:34-37); a chunk is one
                             * batch step of size learn_rate * chunk on the bias and oscillates when that product exceeds
                             * 2 / curvature -- and the smaller the chunk, the closer the rule's bias follows the reference's path
                             * (chunk 1 IS the reference's bias path).  fmx_epoch_stats::w0_chunk_used reports the value used. */
  uint32_t flags;           /* FMX_FLAG_* */
  uint32_t bias_lag;        /* with FMX_FLAG_BIAS_LAG / FMX_APPLY_FUSED: the multipliers of batch b use the bias as it was after
                               the recurrence of batch b - bias_lag (0 = 1 = the bias of the batch start).  fmx_sgd_epoch with
                               FMX_APPLY_FUSED honours 1..4: with bias_lag >= 2 the one-workgroup recurrence of a batch hides
                               under the next launches instead of sitting between them.  The split step
                               (fmx_sgd_partial / fmx_sgd_finish) implements 1 only.
                               Oracle: fmo_sgd_epoch_minibatch_ex(..., bias_lag). */
} fmx_sgd_opts;

#define FMX_FLAG_TIME_MAIN_KERNEL 1u  /* bracket every launch of the dominant kernel with HIP events */
#define FMX_FLAG_PIPELINE 4u          /* fmx_group_sgd_epoch: gather the sums of batch b+1 BEFORE the update of batch b lands, so that
                                         their exchange runs under that update (one batch stale; oracle
                                         fmo_sgd_epoch_minibatch_pipelined) */
#define FMX_FLAG_REJECT_UNSTABLE 8u    /* MINIBATCH: fail with FMX_E_ARG when learn_rate * curvature * batch * C > 2 (the batch rule diverges
                                         there; tests/test_oracle_stability.py) instead of training -- for an explicit batch, and for batch 0
                                         when even the library's floor of 32 rows is unstable (rows with C in the hundreds: the message then
                                         names FMX_SGD_SEQUENTIAL / a lower learn_rate, not "batch 0") */
#define FMX_FLAG_KEEP_WSIDE 32u        /* FMX_APPLY_FUSED on an unsharded handle: this epoch keeps the slot's WEIGHT SIDE STREAM current -- per entry
                                         the new w_j where the entry is the last occurrence of its feature in the slot and its own example
                                         updates it, "gather" otherwise (+128 coalesced bytes per example, 4 bytes per entry of memory).
                                         fmx_predict / fmx_evaluate on that slot then take those weights out of a 4-byte stream instead of one
                                         64-byte fabric request each (fm_model.h:110-115's loop over w: a third of the pass's requests) until
                                         anything else changes w.  For learners that evaluate the train set after every epoch, as
                                         fm_learn_sgd_element::learn does (fm_learn_sgd_element.h:69-70); same numbers either way */
#define FMX_FLAG_EVENT_SYNC 16u        /* FMX_APPLY_FUSED at batches >= 32768: order the launch stream and the recurrence's side stream with
                                         events (four queue packets per batch) instead of the device-side hand-off (bias slots that start as
                                         "pending" + a completion counter: DESIGN.md section 4).  Same numbers either way; the environment
                                         variable FMX_HANDOFF=0 at fmx_create selects events for every epoch of the handle */
#define FMX_FLAG_BIAS_LAG 2u          /* MINIBATCH: the multipliers of a batch use the w0 of the batch START; w0 itself still
                                         advances through the micro-chunk recurrence, which then runs on a side stream
                                         overlapped with the next batch (oracle: fmo_sgd_epoch_minibatch_ex, bias_lag = 1) */

typedef struct fmx_epoch_stats {
  uint64_t rows;            /* examples processed */
  uint64_t batches;
  double   device_seconds;  /* HIP-event time of the epoch's kernels on the handle's stream */
  double   main_kernel_seconds; /* HIP-event time summed over launches of the dominant kernel only */
  uint64_t main_kernel_launches;
  uint32_t max_feature_count; /* MINIBATCH with the segmented update: occurrences of the most frequent feature inside one
                               * batch (the longest segment k_apply_seg sums) */
  uint32_t batch_used;      /* MINIBATCH: rows per batch this epoch ran with (the resolved fmx_sgd_opts::batch) */
  uint64_t deferred_features; /* FMX_APPLY_FUSED: (batch, feature) pairs finished by the segmented kernel, summed over the
                               * epoch's batches (features occurring more than once in their batch; the rest was one pass) */
  double   collision_mass;  /* C of the slot's rows: the mean over pairs of DIFFERENT rows of sum_j |x_ej| |x_e'j| = the expected
                               number of features two rows share (value-weighted) */
  double   batch_gain;      /* learn_rate * curvature * batch_used * C (curvature 1 regression, 1/4 classification): the batch
                               rule follows the reference's online loop for <= 1, degrades above and diverges beyond ~2 */
  uint32_t status;          /* FMX_STAT_* */
  uint32_t w0_chunk_used;   /* MINIBATCH / HOGWILD: the micro-chunk of the bias recurrence this epoch ran with */
  double   setup_seconds;   /* host wall-clock this call spent on ONE-TIME work for the slot (not part of device_seconds): the rows' collision mass
                               and the bucketing of the entries by (batch, feature) -- the segments, masks and lists the MINIBATCH forms
                               walk; libFM never shuffles (fm_learn_sgd_element.h:56), so they are built once per (slot, batch size).
                               0 when everything was cached */
  double   phase_seconds[4]; /* fmx_group_sgd_epoch / feature shards with FMX_FLAG_TIME_MAIN_KERNEL, HIP events on the first local
                               shard's compute stream, summed over the epoch's batches: [0] partial sums of the batch (fmx_sgd_partial),
                               [1] exposed exchange (end of the sums -> the reduced sums are available: what the wire costs beyond
                               what the sums hid), [2] update (multipliers, recurrence hand-off, write-back of the local rows),
                               [3] unused.  Zero on a single unsharded handle. */
} fmx_epoch_stats;

#define FMX_STAT_BATCH_CUT 1u   /* batch = 0 resolved below the 262144 default because of the rows' collision mass */
#define FMX_STAT_UNSTABLE  2u   /* batch_gain > 2: an explicit batch the rule is not stable at on this data */
/* (ABI 7) how the epoch's bias recurrences and stream ordering actually ran -- same numbers in every form; for tests and diagnosis: */
#define FMX_STAT_SCAN_PIT      4u  /* at least one batch's recurrence was solved parallel in time (k_scan_pit) */
#define FMX_STAT_SCAN_SERIAL   8u  /* at least one ran as the one-wavefront chain (k_scan1 / k_scan / the small-batch form) */
#define FMX_STAT_SCAN_FALLBACK 16u /* a grid-wide exchange of k_scan_pit ran into its bound (a workgroup was not resident: the device is
                                      shared with work the library cannot see); one workgroup evaluated that batch's chain serially -- the
                                      result is still the rule's -- and the handle uses the one-wavefront chain from now on */
#define FMX_STAT_EVENT_SYNC    32u /* the launch stream and the recurrence's side stream were ordered by events, not by the device-side
                                      hand-off: requested (FMX_FLAG_EVENT_SYNC / FMX_HANDOFF=0), bias_lag 1, or the handle found that its two
                                      streams do not run concurrently (serialised dispatch: a counter-collecting profiler, AMD_SERIALIZE_KERNEL) */
#define FMX_STAT_XCD_RESIDENT 128u /* small batches (what the stability cut leaves of rows with frequent features): the epoch ran as ONE launch whose
                                      workgroups sit on one accelerator complex die (libfm_amd/csrc/fmx_xcd_kernels.h) instead of two launches per batch */
#define FMX_STAT_SEQ_RUNS 256u     /* FMX_SGD_SEQUENTIAL ran as conflict-free runs: maximal runs of consecutive rows that share no feature, each one
                                      batch step with the bias recurrence coupled example by example -- the same trajectory as the online loop.
                                      fmx_epoch_stats::batches = the number of runs.  (A run whose workgroups never all arrive -- a shared or
                                      partitioned device -- takes no step for the rows concerned: FMX_E_HIP + FMX_STAT_HANDOFF_TIMEOUT, and the
                                      handle takes two launches per run from then on.) */
#define FMX_STAT_SMALL_ONE 512u    /* small batches (< 1 025 rows) of the one-pass minibatch rule ran as ONE launch per batch across all dies: examples,
                                      deferred features and the bias recurrence exchange through tagged slots (libfm_amd/csrc/fmx_small_kernels.h) */
#define FMX_STAT_HANDOFF_TIMEOUT 64u /* a device-side hand-off wait ran into its bound all the same: the examples concerned took NO step
                                      (multiplier 0; a recurrence that never saw its batch handed the bias on unchanged), every parameter is a
                                      valid number, the call returns FMX_E_HIP with this status set, and the handle orders by events from now on */

/* what fmx_sgd_epoch would use for `batch` on this slot (no training): the rows' collision mass (computed once per slot on the
 * device: one histogram pass over the entries), the resolved batch and its gain.  opts may be NULL (= batch 0). */
typedef struct fmx_batch_info {
  double   collision_mass;
  double   batch_gain;
  uint32_t batch;
  uint32_t status;          /* FMX_STAT_* */
} fmx_batch_info;

/* what fm_learn::evaluate_regression / evaluate_classification compute (fm_learn.h:113-153) */
typedef struct fmx_eval {
  double   rmse;            /* regression: sqrt(sum err^2 / N) with clamped predictions */
  double   mae;             /* regression */
  double   accuracy;        /* classification: sign agreement (p>=0 vs y>=0) */
  double   device_seconds;
  uint64_t rows;
  uint32_t flags;           /* FMX_EVAL_WSIDE: the linear weights came out of the slot's weight side stream (FMX_FLAG_KEEP_WSIDE) */
  uint32_t reserved;
} fmx_eval;
#define FMX_EVAL_WSIDE 1u

/* ---- lifetime ------------------------------------------------------------------------------- */
/* replaces: fm_model fm; fm.init() allocation (fm_model.h:91-99) + new fm_learn_* (libfm.cpp:271-293)
 * Parameter tables are PLACED (fmx_config::place_candidates): on MI355X physical memory falls into three classes of ~96 GB (what one
 * would expect of the three ranks of the 12-high HBM3E stacks) and random row traffic inside ONE class runs at 4.9 TB/s, spread over
 * two at 6.1 (scripts/ubench/placement_classes.hip) -- a plain allocation lands in one class or straddles two by
 * luck, which moved the training step by 10-20 % from process to process.  fmx_create therefore takes 1 GiB chunks (hipMemCreate),
 * classifies each by probing it together with a reference chunk, maps chunks of two classes alternately into one virtual range, puts
 * V at its start and w across a chunk boundary behind it, and returns the chunks it does not need.  fmx_place_info reports what it
 * found.  If the virtual-memory API is unavailable the tables are plain allocations (best of two candidates). */
/* Environment read by fmx_create (A/B knobs; none changes a result beyond fp32 rounding): FMX_HANDOFF=0 (events instead of the device-side
 * bias hand-off), FMX_SCAN=serial (the bias recurrence as a one-wavefront chain instead of parallel in time), FMX_ARENA_CACHE=0 (no arena
 * kept for the next handle), FMX_MULTI_FILL=8..32 (entries per 32-slot round the short-row shard kernels group examples for; default 24). */
int fmx_create(const fmx_config *cfg, fmx_handle *out);
int fmx_destroy(fmx_handle h);
/* fmx_destroy keeps ONE placed arena per device alive for the next fmx_create on that device (which then takes it over without
 * probing: fmx_place_info::pool = 0) -- the memory of the largest arena destroyed so far stays allocated until it is reused, this call
 * returns it, ANY device allocation of the library runs short of memory (it is then given back and the allocation retried once), or
 * the process ends.  Other processes on the device do not see it as free before that.  FMX_ARENA_CACHE=0 in the environment turns this off. */
int fmx_release_cached_memory(void);
/* how the parameter tables of a handle were placed */
typedef struct fmx_place_info {
  int32_t  method;          /* 0 = plain allocations (small tables / first fit), 1 = best of several candidate allocations,
                               2 = arena of chunks from two memory classes */
  uint32_t chunks;          /* method 2: 1 GiB chunks the arena holds */
  uint32_t per_class[2];    /*           ... of the first / the second class (equal up to one = balanced) */
  uint32_t pool;            /*           chunks taken and probed to find them (the others were returned before fmx_create ended) */
  uint32_t classes_seen;    /*           memory classes seen among the pool */
  double   seconds;         /* host time of the placement */
} fmx_place_info;
int fmx_get_place_info(fmx_handle h, fmx_place_info *out);
/* the layout of that arena for tables of v_bytes and w_bytes (host arithmetic, no device needed): V at offset 0, w at *w_offset --
 * centred on a chunk boundary --, *chunks chunks of *chunk_bytes */
int fmx_place_layout(uint64_t v_bytes, uint64_t w_bytes, uint64_t *chunk_bytes, uint32_t *chunks, uint64_t *w_offset);
/* text of the last error on this handle (h may be NULL: last creation error). Never NULL. */
const char *fmx_last_error(fmx_handle h);
int fmx_abi_version(void);
/* fmx_sgd_opts::w0_chunk = 0 resolves to this (host arithmetic; task = FMX_TASK_*): the largest power of two <= FMX_W0_CHUNK_CAP with
 * learn_rate * chunk * curvature <= 1.  Until ABI 5 the cap was 256; the reference advances the bias per example (fm_sgd.h:34-37) and the
 * finer recurrence ends 17x closer to its bias path at the bench shape. */
#define FMX_W0_CHUNK_CAP 32
uint32_t fmx_default_w0_chunk(double learn_rate, int task);
/* number of visible HIP devices (0 when none / no driver) */
int fmx_device_count(void);

/* ---- parameters: the fm_model block (fm_model.h:46-48) -------------------------------------- */
/* host -> device. w may be NULL when k1 == 0, v may be NULL when k == 0.  In sharded mode the FULL
 * arrays are passed and the handle keeps its own features. */
int fmx_set_params(fmx_handle h, double w0, const double *w, const double *v);
/* device -> host, same layout; after learn() the host fm_model must hold the result (libfm.cpp:431-434).
 * Sharded mode: only this shard's features are written, the rest of w / v is left untouched. */
int fmx_get_params(fmx_handle h, double *w0, double *w, double *v);
/* device-side fill for workloads too large to stage through the host (bench.py):
 * w0 = 0, w = 0, v[f][j] = mean + fmo_init_value(seed, j, f, stdev) -- a counter-hash uniform with unit
 * variance (same definition as oracle/fm_oracle.c), NOT the reference's rand() stream (fm_model.h:96). */
int fmx_init_params(fmx_handle h, double init_mean, double init_stdev, uint64_t seed);
/* selected rows of the parameter block (spot checks at sizes where the full fm_model does not fit the host):
 * for i < count: w_out[i] = w[ids[i]], v_out[i*num_factor + f] = v[f][ids[i]].  Sharded handles accept only their
 * own features. */
int fmx_get_param_rows(fmx_handle h, const uint32_t *ids, uint32_t count, double *w_out, double *v_out);
/* fm_model::saveModel / loadModel (fm_model.h:132-190; `-save_model`, `-load_model`, libfm.cpp:258-269, 431-434): the
 * reference's text file ("#global bias W0", "#unary interactions Wj", "#pairwise interactions Vj,f": one line per feature,
 * its factors separated by blanks; doubles as an ostream prints them).  The device table is feature-major -- the file's own
 * order -- so it is streamed in row chunks without ever building the fp64 factor-major block on the host.
 * fmx_load_model fails with FMX_E_ARG "malformed model file" where loadModel returns 0 (wrong number of factors per line,
 * truncated file); a k = 1 file, which the reference's splitString cannot read back (fm_model.h:195-205), is accepted.
 * Shards: every shard may load the same file (it keeps its own features); saving a sharded model goes through fmx_get_params. */
int fmx_save_model(fmx_handle h, const char *path);
int fmx_load_model(fmx_handle h, const char *path);
/* the scalar bias alone (cheap; used between minibatches by multi-process drivers) */
int fmx_get_w0(fmx_handle h, double *w0);

/* ---- attribute groups (`-meta`): DataMetaInfo::attr_group, src/libfm/src/Data.h:39-46, loaded by
 * loadGroupsFromFile (:85-97).  group_of_feature[num_attribute] (host, ids < num_groups), indexed by GLOBAL feature id
 * (a feature shard keeps the entries of its own features).
 * Groups select the prior of a coordinate in ALS / MCMC (w_lambda(g), w_mu(g), v_lambda(g,f), v_mu(g,f);
 * fm_learn_mcmc.h:464-466, :583-585) and the learned regularisation in SGDA (reg_w(g), reg_v(g,f);
 * fm_learn_sgd_element_adapt_reg.h:155-166); plain SGD ignores them like the reference does.
 * NULL or num_groups <= 1 goes back to one group.  Not allowed while an ALS / SGDA session is open. */
int fmx_set_groups(fmx_handle h, const uint32_t *group_of_feature, uint32_t num_groups);

/* ---- rows: what Data::load produces (Data.h:237-270) ---------------------------------------- */
/* uploads a data set into `slot` (0..FMX_MAX_SLOTS-1).  entries: {uint32 id; float value}[nnz] in row
 * order (the buffer Data::load allocates), row_ptr: uint64[n_rows+1] prefix sums of sparse_row::size,
 * target: float[n_rows] (already rewritten to +-1 for classification, libfm.cpp:302-306); may be NULL
 * for predict-only slots.  ids must be < num_attribute (asserted by the reference, fm_model.h:112). */
#define FMX_MAX_SLOTS 8
int fmx_upload_rows(fmx_handle h, int slot, const void *entries, const uint64_t *row_ptr,
                    const float *target, uint32_t n_rows, uint64_t nnz);
/* block-structured data (`-relation`; RelationData / RelationJoin, src/libfm/src/relation.h:32-60, loaded at
 * libfm.cpp:172-196): every main row c is joined with row data_row_to_relation_row[c] of each relation block, whose
 * attribute ids start at attr_offset (libfm.cpp:213-216).  The rows are expanded ON THE DEVICE into one CSR
 * (main entries, then the blocks in order) so that the slot behaves like any other; the learners then compute what
 * the reference's per-block caches compute (fm_learn_mcmc.h:478-527, 734-790, 849-909) on the same design matrix.
 * At most 8 relations.  A feature shard (shard_world > 1) joins the rows on the host and keeps its own features. */
typedef struct fmx_relation {
  const void     *entries;                    /* the block's own rows: sparse_entry<float>[nnz], ids local to the block */
  const uint64_t *row_ptr;                    /* [n_rows + 1] */
  uint32_t        n_rows;                     /* RelationData::num_cases */
  uint32_t        reserved;
  uint64_t        nnz;
  const uint32_t *data_row_to_relation_row;   /* [n_rows of the MAIN data]; RelationJoin, relation.h:56 */
  uint64_t        attr_offset;                /* RelationData::attr_offset */
} fmx_relation;
int fmx_upload_block_rows(fmx_handle h, int slot, const void *entries, const uint64_t *row_ptr, const float *target,
                          uint32_t n_rows, uint64_t nnz, const fmx_relation *relations, uint32_t n_relations);
/* the same with a choice of representation:
 *   FMX_BLOCKS_EXPAND  the joined rows are materialised on the device (what fmx_upload_block_rows does): any learner runs on
 *                      them, at the data volume of the joined table;
 *   FMX_BLOCKS_KEEP    main rows and blocks stay apart, as in the reference: fmx_predict / fmx_evaluate add the block rows'
 *                      sums through the mapping, and fmx_als_* sweeps a block's attributes through per-block-row caches
 *                      (fm_learn_mcmc.h:478-527 cache set-up, :734-790 draw_w_rel, :849-909 draw_v_rel, restated): a block
 *                      attribute costs its column in the BLOCK plus two passes over the main rows per block and coordinate
 *                      family, not a column of the joined table.  ALS / MCMC and predict only (the reference's SGD learners
 *                      reject relations too, fm_learn_sgd.h:61-63). */
#define FMX_BLOCKS_EXPAND 0u
#define FMX_BLOCKS_KEEP 1u
int fmx_upload_block_rows_ex(fmx_handle h, int slot, const void *entries, const uint64_t *row_ptr, const float *target,
                             uint32_t n_rows, uint64_t nnz, const fmx_relation *relations, uint32_t n_relations, uint32_t flags);
/* synthetic one-hot field rows generated on the device (bench workload; same
 * definition as oracle/fm_oracle.c fmo_synth_rows): rows row0 .. row0+n_rows-1 */
int fmx_synth_rows(fmx_handle h, int slot, uint64_t seed, uint64_t row0, uint32_t n_rows, uint32_t nnz);
/* the same with a choice of id distribution:
 *   FMX_SYNTH_UNIFORM  uniform ids within each of the nnz fields (fmx_synth_rows)
 *   FMX_SYNTH_CRITEO   BASELINE configs[2] / SURVEY section 8d "Criteo-shaped": the first 13 fields ("numeric", binned) hold at
 *                      most 100 ids each with a geometric frequency profile, the other nnz - 13 fields share the remaining ids
 *                      equally and draw them Zipf(s = 1.05) (inverse-CDF of the continuous power law); labels +-1 with a 1:3
 *                      imbalance.  Needs nnz > 13 and num_attribute >= 1300 + (nnz - 13). */
#define FMX_SYNTH_UNIFORM 0u
#define FMX_SYNTH_CRITEO 1u
int fmx_synth_rows_ex(fmx_handle h, int slot, uint64_t seed, uint64_t row0, uint32_t n_rows, uint32_t nnz, uint32_t shape);
int fmx_free_rows(fmx_handle h, int slot);
/* copies a slot back to the host (tests / debugging): sizes via fmx_rows_info first.  Sharded handles return
 * their LOCAL rows (kept entries only, ids = local row of the feature on this shard). Any pointer may be NULL. */
int fmx_rows_info(fmx_handle h, int slot, uint32_t *n_rows, uint64_t *nnz);
int fmx_download_rows(fmx_handle h, int slot, void *entries, uint64_t *row_ptr, float *target);

/* ---- host-side reader of libFM's text format (no device, no handle) ----------------------------------------
 * Data::load, text branch (src/libfm/src/Data.h:180-285): lines "target id:value id:value ...", blank lines and lines
 * starting with '#' skipped, leading / trailing blanks and tabs allowed, anything else rejected with the reference's
 * message ("cannot parse line ..." / "unable to open ...") in `err`.  The buffers are malloc'ed; release them with
 * fmx_free_host_rows.  They are exactly what fmx_upload_rows takes. */
typedef struct fmx_host_rows {
  void     *entries;        /* sparse_entry<float>[nnz] */
  uint64_t *row_ptr;        /* [n_rows + 1] */
  float    *target;         /* [n_rows] */
  uint32_t  n_rows;
  uint32_t  num_feature;    /* largest id + 1 (Data.h:226-228) */
  uint64_t  nnz;
  float     min_target, max_target;   /* Data.h:205-206 */
  uint32_t  flags;          /* FMX_HOST_PINNED: the buffers are page-locked (hipHostMalloc) */
  uint32_t  reserved;
} fmx_host_rows;
#define FMX_HOST_PINNED 1u
int  fmx_read_libsvm(const char *path, fmx_host_rows *out, char *err, size_t err_len);
/* Data::load, binary branch (src/libfm/src/Data.h:119-178): <prefix>.x + <prefix>.y as written by tools/convert.cpp:137-200
 * (LargeSparseMatrix::saveToBinaryFile, src/util/fmatrix.h:44-50,121-140; DVector::saveToBinaryFile, matrix.h:344-358), or --
 * when only the transpose is on disk, which is what als / mcmc runs keep (libfm.cpp:143-147, tools/transpose.cpp) --
 * <prefix>.xt + <prefix>.y, from which the rows are rebuilt; the older .data / .datat / .target names work too.
 * With a HIP device present the buffers are page-locked, so fmx_upload_rows moves them by DMA while it checks the ids. */
int  fmx_read_binary(const char *prefix, fmx_host_rows *out, char *err, size_t err_len);
void fmx_free_host_rows(fmx_host_rows *rows);

/* ---- fm_model::predict over a data set (fm_model.h:105-127 via fm_learn.h:63-65) -------------- */
/* raw y-hat per row (no clamp / sigmoid: fm_learn_sgd::predict applies those on the host,
 * fm_learn_sgd.h:80-87).  out: double[n_rows]. Sharded handles return the PARTIAL sums only. */
int fmx_predict(fmx_handle h, int slot, double *out);
/* fm_learn::evaluate (fm_learn.h:93-153) fused on the device */
int fmx_evaluate(fmx_handle h, int slot, fmx_eval *out);

/* ---- fm_learn_sgd_element::learn, one epoch (fm_learn_sgd_element.h:56-67) -------------------- */
int fmx_sgd_epoch(fmx_handle h, int slot, const fmx_sgd_opts *opts, fmx_epoch_stats *stats);
/* On a handle that is one rank of a communicator (fmx_comm_init_rank, one process per GPU) this call is COLLECTIVE, every time: the
 * shards' shares of the collision mass are summed over RCCL (one 8-byte all-reduce), so every rank must make it -- and fmx_sgd_epoch,
 * which resolves its batch through it, is collective in the same sense.  (Until ABI 5 the sum was cached per rank, which made the
 * decision to enter the collective per rank too: a rank that re-uploaded its slot entered alone.) */
int fmx_sgd_batch_info(fmx_handle h, int slot, const fmx_sgd_opts *opts, fmx_batch_info *out);
/* the same decision as host arithmetic (no device needed): the batch fmx_sgd_epoch runs with for rows of the given collision mass --
 * requested != 0: that batch, with its gain and status; requested == 0: 262144 cut to the largest power of two (never below 32) with
 * learn_rate * curvature * batch * collision_mass <= 1 (curvature 1 for regression, 1/4 for classification) */
int fmx_batch_rule(int32_t task, double learn_rate, double collision_mass, uint32_t requested, fmx_batch_info *out);

/* ---- minibatch step split at the exchange point, for one-process-per-GPU drivers --------------
 * partial: floats per batch = fmx_partial_floats(h, batch): [batch][KP] partial factor sums followed by
 *          [batch] scalars (linear term - 0.5*sum of squares).  d_partial is DEVICE memory, 16-byte aligned.
 * The driver all-reduces (sum) d_partial across the feature shards (RCCL), then calls finish on every
 * rank: multipliers + w0 micro-chunks (identical on all ranks) and the scatter-add into the local shard.
 * `stream` is a hipStream_t (NULL = the handle's own stream). */
int fmx_partial_floats(fmx_handle h, uint32_t batch, uint64_t *n_floats);
int fmx_sgd_partial(fmx_handle h, int slot, uint64_t row0, uint32_t n_rows, float *d_partial, void *stream);
int fmx_sgd_finish(fmx_handle h, int slot, uint64_t row0, uint32_t n_rows, const float *d_partial,
                   const fmx_sgd_opts *opts, void *stream);
/* sharded predict: finish a partial buffer into y-hat (device float[n_rows]) */
int fmx_predict_finish(fmx_handle h, uint32_t n_rows, const float *d_partial, float *d_yhat, void *stream);

/* ---- several GPUs --------------------------------------------------------------------------------------------------
 * libFM is ONE process that constructs the learner and calls learn() (libfm.cpp:271-293, :415).  To use P GPUs from that
 * process, create P handles -- shard i of P (fmx_config::shard_rank / shard_world, normally shard_hash = 1) on device i --
 * give every one of them the full parameter block and the full rows (each keeps its own features), and tie them together:
 *   fmx_group_create     P handles on P DISTINCT devices: one RCCL communicator per handle (ncclCommInitRank inside one
 *                        ncclGroupStart/End); P handles on ONE device: "loopback" -- the exchange is a local reduction kernel
 *                        (tests and single-GPU boxes); a single unsharded handle: pass-through.
 *   fmx_group_sgd_epoch  one epoch of the minibatch rule: per batch the partial sums of every shard (fmx_sgd_partial), ONE
 *                        all-reduce of [batch][KP + 1] floats over xGMI, then multipliers / bias recurrence (redundantly,
 *                        identically on every shard) and the update of the shard's own rows (fmx_sgd_finish).  The rule is
 *                        the one a single handle runs with FMX_APPLY_SEGMENTED / FMX_APPLY_FUSED and the same bias_lag.
 *   fmx_group_predict / fmx_group_evaluate   fm_learn::predict / evaluate over the shards.
 * One process PER GPU instead (torchrun-style launchers): rank 0 calls fmx_comm_unique_id and hands the 128 bytes to the
 * other ranks by whatever means the launcher has; every rank creates its shard handle, calls fmx_comm_init_rank and then
 * plain fmx_sgd_epoch, which runs the same schedule with its one local shard.
 * RCCL (librccl.so.1) is loaded on first use; without it only loopback groups and single handles work. */
/* the ownership rule (fmx_config::shard_hash) as host arithmetic -- no device needed: owner[i] / local_row[i] of feature
 * ids[i] (either output may be NULL), and the inverse: the global id of a shard's local row. */
int fmx_shard_place(uint64_t num_attribute, int shard_world, int shard_hash, const uint32_t *ids, uint64_t count,
                    int32_t *owner, uint32_t *local_row);
int fmx_shard_global(uint64_t num_attribute, int shard_world, int shard_hash, int shard_rank, const uint32_t *local_rows,
                     uint64_t count, uint32_t *ids);
#define FMX_COMM_ID_BYTES 128
typedef struct fmx_group_s *fmx_group;
int fmx_comm_unique_id(void *id128);
int fmx_comm_init_rank(fmx_handle h, const void *id128, int rank, int world);
int fmx_comm_destroy(fmx_handle h);
int fmx_group_create(fmx_handle *handles, int n, fmx_group *out);
int fmx_group_destroy(fmx_group g);
const char *fmx_group_last_error(fmx_group g);
/* fmx_set_params / fmx_upload_rows for ALL shards of a group with the host data crossing PCIe ONCE: the fp64 block (51 GB at the
 * north-star size) and the rows are staged on the first shard's device, reach the other devices over xGMI (hipMemcpyPeer), and
 * every shard converts / filters its own features on its device (k_stage_in, k_shard_rows).  Same result as calling the
 * per-handle functions on every shard with the full arrays.  Not for slots with `-relation` blocks. */
int fmx_group_set_params(fmx_group g, double w0, const double *w, const double *v);
int fmx_group_upload_rows(fmx_group g, int slot, const void *entries, const uint64_t *row_ptr, const float *target,
                          uint32_t n_rows, uint64_t nnz);
int fmx_group_sgd_epoch(fmx_group g, int slot, const fmx_sgd_opts *opts, fmx_epoch_stats *stats);
int fmx_group_predict(fmx_group g, int slot, double *out);
int fmx_group_evaluate(fmx_group g, int slot, fmx_eval *out);

/* ---- fm_learn_mcmc (ALS = MCMC without sampling, libfm.cpp:135-139) ---------------------------------
 * The learner keeps e(c) = y-hat(c) - target(c) and q_f(c) per training row (e_q_term, fm_learn_mcmc.h:46-49)
 * and sweeps the coordinates through X^T (built on the device from the slot's rows, Data.h:292-341).
 *   fmx_als_begin : fm_learn_mcmc::learn set-up (:1160-1172) + _learn's first prediction and e -= target
 *                   (fm_learn_mcmc_simultaneous.h:69-86).
 *   fmx_als_sweep : one iteration of _learn (:88-196): draw_all (fm_learn_mcmc.h:430-641: draw_w0, draw_w per
 *                   feature, per factor add_main_q + draw_v), full re-prediction of the train rows, train metric,
 *                   new residuals.  Test predictions of the iteration = fmx_predict on the test slot.
 *   fmx_als_end   : frees the caches (:1192-1200).
 * do_sample = 0 is ALS (alpha = 1, mu = 0, lambdas from -regular: reg0 -> w0, w_lambda, v_lambda; libfm.cpp:326-365).
 * do_sample = 1 draws every coordinate from its posterior N(mean, sigma^2) with a counter-based generator (NOT the
 * reference's libc rand() stream: statistical, not bitwise, parity); alpha and the prior means/precisions are
 * supplied per sweep by the caller (the hyper-prior draws of :911-1097 are scalar work that stays on the host).
 * No relations (block structure) -- out of scope. */
typedef struct fmx_als_opts {
  double   alpha;           /* fm_learn_mcmc::alpha (1 for ALS) */
  double   w_mu, w_lambda;  /* prior of the linear weights */
  double   v_mu, v_lambda;  /* prior of the factors */
  int32_t  do_sample;       /* 0 = ALS, 1 = Gibbs draws */
  int32_t  reserved;
  uint64_t seed;
  const double *v_mu_f;     /* optional per-factor prior means v_mu(g=0,f) [num_factor]  (fm_learn_mcmc.h:76); NULL = v_mu */
  const double *v_lambda_f; /* optional per-factor prior precisions v_lambda(g=0,f);              NULL = v_lambda */
  /* with attribute groups (fmx_set_groups, G = num_groups > 1) the priors are per group; any table may be NULL
   * (falls back to the fields above).  Layouts are the reference's: w_*(g) DVector[G], v_*(g,f) DMatrix[G][num_factor]
   * (fm_learn_mcmc.h:1116-1122). num_groups must be 0 (no tables) or equal the handle's group count. */
  uint32_t      num_groups;
  uint32_t      reserved2;
  const double *w_mu_g, *w_lambda_g;     /* [G] */
  const double *v_mu_gf, *v_lambda_gf;   /* [G][num_factor] */
} fmx_als_opts;

typedef struct fmx_als_stats {
  double   train_metric;    /* rmse_train (regression, clamped) or acc_train (classification) of this iteration */
  double   device_seconds;
  uint32_t levels;          /* dependency levels of the sweep (1 launch per level and coordinate family) */
  uint32_t reserved;
  double   sum_e_sqr;       /* sum over train rows of e^2 BEFORE the sweep (draw_alpha's statistic, :918-920) */
} fmx_als_stats;

int fmx_als_begin(fmx_handle h, int train_slot);
/* statistics the hyper-prior draws need (draw_alpha :911-939, draw_w_mu/_lambda :941-1017, draw_v_mu/_lambda
 * :1019-1097), reduced on the device in fp64:
 *   out[0] = sum_c e_c^2 over the train rows (current residuals), out[1] = sum_c e_c,
 *   then a [1 + num_factor][G][2] block (G = attribute groups, 1 without fmx_set_groups): row 0 = w, row 1+f = v_f;
 *   per group g the pair {sum_{j in g} theta_j, sum_{j in g} theta_j^2}  (the per-group loops of :946-951, :987-992,
 *   :1026-1031, :1067-1072).  With one group: out[2] = sum w, out[3] = sum w^2, out[4+2f] = sum v_f, out[5+2f] = sum v_f^2.
 * out must hold 2 + 2*G*(1 + num_factor) doubles. */
int fmx_als_moments(fmx_handle h, double *out);
int fmx_als_sweep(fmx_handle h, const fmx_als_opts *opts, fmx_als_stats *stats);
int fmx_als_end(fmx_handle h);
/* the same learner over the feature shards of a group (BASELINE configs[4]: "V sharded across 8 x MI355X"): every shard
 * sweeps its own features, level by level in the GLOBAL dependency order; the {e, q} cache is replicated, one all-reduce per
 * (coordinate family, level) carries the changes of the level's draws, and the re-prediction all-reduces the shards' partial
 * y-hat and q_f (fm_learn_mcmc.h:430-641 with e / q of :46-49 replicated).  Same results as one
 * unsharded handle: ALS to rounding, MCMC draw for draw (the noise of a coordinate is keyed by its GLOBAL feature id). */
int fmx_group_als_begin(fmx_group g, int train_slot);
int fmx_group_als_moments(fmx_group g, double *out);
int fmx_group_als_sweep(fmx_group g, const fmx_als_opts *opts, fmx_als_stats *stats);
int fmx_group_als_end(fmx_group g);

/* ---- fm_learn_sgd_element_adapt_reg (`-method sgda`; src/libfm/src/fm_learn_sgd_element_adapt_reg.h) -------------
 * Self-adaptive regularisation: theta steps on the train rows alternate with lambda steps on the validation rows,
 * strictly online: fmx_sgda_epoch is that order on one wavefront (a parity instrument), fmx_sgda_epoch_minibatch the batch form.  Regularisation is
 * learned per attribute group (fmx_set_groups): reg_w(g), reg_v(g,f).
 *   fmx_sgda_begin : the learner's start of learn() (:256-262): w := 0, reg_w := 0, reg_v := 0, shadow gradients := 0
 *   fmx_sgda_epoch : one iteration of the epoch loop (:262-279); do_lambda_steps = 0 in the first iteration (:269)
 *   fmx_sgda_get_reg : [G][1 + num_factor]: reg[g*(1+k)] = reg_w(g), reg[g*(1+k)+1+f] = reg_v(g,f)
 *                      (what -rlog reports as regw[g], regv[g,f]; :119-132)
 */
int fmx_sgda_begin(fmx_handle h);
int fmx_sgda_epoch(fmx_handle h, int train_slot, int validation_slot, int do_lambda_steps, fmx_epoch_stats *stats);
/* the learner in BATCH form (oracle fmo_sgda_epoch_minibatch; the online order above is one wavefront and slower than the
 * reference's CPU): per batch of `batch` train rows (0 = the library's choice, as fmx_sgd_opts::batch, with this learner's doubled regression curvature) the theta steps as a minibatch rule -- this learner's
 * multiplier, reg_0 = 0, the learned regularisation 2 reg(g[,f]) theta per occurrence, the shadow gradient of a touched
 * parameter = the sum of its occurrences' gradients -- then, do_lambda_steps, the lambda steps of the next `batch` validation
 * rows (cyclic, :271-274), each as sgd_lambda_step (:201-248) with the regularisation of the batch start, their changes
 * summed and applied, clamped at 0, once.  batch = w0_chunk = 1 is the reference's loop.  w0_chunk 0 = library default. */
int fmx_sgda_epoch_minibatch(fmx_handle h, int train_slot, int validation_slot, int do_lambda_steps, uint32_t batch,
                             uint32_t w0_chunk, fmx_epoch_stats *stats);
int fmx_sgda_get_reg(fmx_handle h, double *reg /* [G][1 + num_factor] */);
int fmx_sgda_end(fmx_handle h);

/* ---- introspection --------------------------------------------------------------------------- */
typedef struct fmx_info {
  uint64_t n_local;         /* features held by this handle */
  int32_t  k_padded;        /* floats per example in the partial-sum buffers: the power of two >= k (a device ROW is k rounded up to 16) */
  int32_t  device;
  uint64_t bytes_params;    /* device bytes of w + V */
  char     device_name[64];
  char     arch[32];        /* e.g. "gfx950" */
} fmx_info;
int fmx_get_info(fmx_handle h, fmx_info *out);
int fmx_synchronize(fmx_handle h);

#ifdef __cplusplus
}
#endif
#endif /* FMX_H_ */

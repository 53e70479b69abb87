// fmx_internal.h -- shared by the translation units of libfmx.so (fmx_core.hip, fmx_sgd.hip, fmx_als.hip).
// The C-ABI is include/fmx.h; nothing declared here is exported.
#pragma once
#pragma GCC visibility push(default)      // the C-ABI is the only thing libfmx.so exports (-fvisibility=hidden)
#include "../../include/fmx.h"
#pragma GCC visibility pop
#include "fmx_kernels.h"
#include "fmx_als_kernels.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

using namespace fmx;

struct Slot {
  Entry*    ent = nullptr;
  uint64_t* row_ptr = nullptr;
  float*    target = nullptr;
  uint32_t  n_rows = 0;
  uint64_t  nnz = 0;
  uint32_t  max_row = 0;
  uint32_t  fixed_nnz = 0;           // != 0: every row holds exactly this many entries (row r starts at r * fixed_nnz)
  bool      used = false;
  double    coll_mass = -1.0;        // collision mass of the rows (ensure_coll_mass); < 0: not computed yet
  double    coll_mass_world = -1.0;  // one process per GPU: the shards' shares summed over the communicator (one collective per slot)
  // per-batch transposed rows for FMX_APPLY_SEGMENTED (built lazily for one batch size)
  uint32_t  seg_B = 0;
  TEntry*   t_ent = nullptr;
  uint32_t* seg_feat = nullptr;
  uint32_t* seg_rel = nullptr;
  uint32_t  nseg = 0;
  bool      all_ones = false;        // every value of the slot is 1.0f (found while the segments are built)
  uint32_t  max_seg_count = 0;       // occurrences of the most frequent feature inside one batch
  std::vector<uint32_t> batch_seg;   // [n_batches+1] first segment of every batch
  std::vector<uint64_t> batch_base;  // [n_batches+1] first entry of every batch
  // FMX_APPLY_FUSED (k_fused<FUSED_EXACT>): which entries the one-pass kernel must leave to k_apply_seg
  uint64_t* cmask = nullptr;         // [n_rows] bit i: entry i's feature occurs more than once in the row's batch
  uint32_t* cseg = nullptr;          // [ncseg] batch-local indices of the segments k_apply_seg finishes
  uint32_t  ncseg = 0;
  CDesc*    cdesc = nullptr;         // [ncseg] the same list as 32-byte records {feature, first entry, end entry, index, first two occurrences}
  uint32_t  fused_cap = 0;           // longest row the k_fused instance of this slot keeps in registers
  std::vector<uint32_t> cbatch;      // [n_batches+1] first cseg entry of every batch
  uint32_t* d_batch_seg = nullptr;   // the three per-batch tables on the device (same contents as batch_seg / cbatch / batch_base)
  uint32_t* d_cbatch = nullptr;
  uint64_t* d_batch_base = nullptr;
  // weight side stream (fmx_kernels.h row_sums; FMX_FLAG_KEEP_WSIDE): wside[i] = w[id of entry i] or NaN, lmask[row] = which entries are
  // the last occurrence of their feature in the slot; current while wside_version == the handle's w_version
  float*    wside = nullptr;
  uint64_t* lmask = nullptr;
  uint64_t  wside_version = 0;
  // FMX_SGD_SEQUENTIAL: the slot cut into maximal runs of consecutive rows that share no feature (fmx_seq_kernels.h "Conflict-free runs")
  std::vector<uint32_t> run_start;   // [n_runs + 1]; empty: not built
  std::vector<uint8_t>  run_single;  // [n_runs] 1: one row that repeats an id (entry-by-entry kernel)
  std::vector<struct BlockRows*> blocks;   // `-relation` blocks kept apart from these (main) rows; empty: plain / expanded rows
};

// one `-relation` block kept apart from the main rows (fmx_upload_block_rows_ex, FMX_BLOCKS_KEEP): RelationData +
// RelationJoin, src/libfm/src/relation.h:32-60
struct BlockRows {
  Slot      rows;                  // the block's own rows (ids local to the block); owns the block's X^T segments too
  uint32_t* map = nullptr;         // [main rows] -> block row (RelationJoin::data_row_to_relation_row)
  uint32_t  attr_offset = 0;       // RelationData::attr_offset: global id of the block's attribute 0
  uint32_t* brow_ptr = nullptr;    // [block rows + 1] / brow_list [main rows]: the main rows that map to a block row
  uint32_t* brow_list = nullptr;
  float*    pbuf = nullptr;        // [block rows][KP + 1] partial sums of the block rows (predict)
};
void free_block(BlockRows* b);

struct AlsBlock {                  // ALS / MCMC state of one kept block (fm_learn_mcmc.h:50-58 relation_cache, restated)
  uint32_t* level_list = nullptr;
  std::vector<uint32_t> level_ptr;
  double*   cache = nullptr;       // [7][block rows]: we, weq, wc, wc2, qb, dy, qb0   (struct of arrays)
  double*   qb_all = nullptr;      // [KP][block rows]: the block rows' factor sums of the latest re-prediction
  double*   cpart = nullptr;       // [block rows]: lin - 0.5 * sum of squares of the block rows
  int       lanes = 8;
};

struct AlsState {
  int       slot = -1;
  EQ*       e = nullptr;          // [N] {residual, current factor's q}
  double*   q = nullptr;          // [KP][N]
  uint8_t*  seen = nullptr;       // [n_local] feature has a training column
  uint32_t* level_list = nullptr; // segments ordered by level
  uint4*    ldesc = nullptr;      // the same list as {feature, first entry, end entry, segment} records (k_als_ldesc)
  std::vector<uint32_t> level_ptr;
  uint64_t  iter = 0;
  std::vector<AlsBlock> blk;      // kept blocks of the train slot
  EQ*       delta = nullptr;      // feature shards: [N] what the draws of the current (family, level) step change in {e, q}
  double*   epart = nullptr;      // feature shards: [N] partial y-hat of the re-prediction
  float*    vt = nullptr;         // [num_factor][vt_stride] factor-major shadow of the seen features' factors, level order
  size_t    vt_stride = 0;
  // split step (large levels of an unsharded session): the entries of every level in ROW order + the draws' {old, new} pairs
  uint32_t* r_row = nullptr; uint32_t* r_pos = nullptr; float* r_x = nullptr;   // [nnz], level after level
  float2*   dth = nullptr;        // [largest level]
  std::vector<uint32_t> lev_ent;  // [n_levels + 1] entry range of each level in r_*
  std::vector<uint8_t> lev_dense; // [n_levels] the level's entries are the rows 0 .. N-1 in order (k_als_rows_dense); 2: ... and every value is 1
  uint32_t* t_row = nullptr;      // [nnz] the rows of X^T's entries alone: what a level of unit values streams instead of {row, value}
  uint32_t  split_min = 0;        // levels with at least this many entries take the split step (0: none)
  double*   prior = nullptr;      // [1 + k][2][G]: per coordinate family (row 0 = w, 1+f = v_f) lambda[G] then mu[G]
  std::vector<double> prior_host;
};

struct LagState {                 // FMX_FLAG_BIAS_LAG bookkeeping (split step): w0 lives in w0_pp[step & 1] while active
  static constexpr uint32_t RING = 8;                 // >= depth + 2
  bool       active = false;
  uint64_t   step = 0;
  uint32_t   depth = 1;           // batches the multipliers' bias lags behind (fmx_sgd_opts::bias_lag)
  hipEvent_t ev_rest = nullptr, ev_scan[RING] = {};
  hipStream_t in_stream = nullptr;  // small batches: the recurrences ride in the update's launches on THIS stream (not the side stream): lag_flush drains it
};

struct SgdaState { float* gw = nullptr; float* gv = nullptr; double* reg = nullptr; double* dreg = nullptr;   // dreg: [workgroups][G][1 + KP] partial lambda changes
                   size_t dreg_cap = 0; };

// The parameter tables of a big model live in an ARENA: one virtual range backed by 1 GiB physical chunks (hipMemCreate) taken
// alternately from two memory classes of the device (fmx_create)
struct Arena {
  void*    va = nullptr;         // reserved range; chunk i is mapped at va + i * chunk_bytes
  size_t   bytes = 0, chunk_bytes = 0;
  size_t   reserved_bytes = 0;   // size of the virtual range (>= bytes: a range taken over from the per-device cache may be longer)
  uint32_t n_chunks = 0;
  uint32_t per_class[2] = {0, 0};   // chunks of the two classes the tables are built from
  uint32_t pool = 0;             // chunks that were taken and classified to get them (the others were returned)
  uint32_t classes_seen = 0;
  double   seconds = 0.0;        // host time of the whole placement
  int      method = 0;           // fmx_place_info::method
  std::vector<uint8_t> cls;      // [n_chunks] which class chunk i came from (0 / 1: the two the tables are built from, 2: a filler) -- what a
                                 // later, smaller model that takes over a PREFIX of this arena recounts per_class from (arena_build)
};

struct fmx_context_s {
  fmx_config cfg;
  AlsState   als;
  SgdaState  sgda;
  LagState   lag;
  int        KP = 1;
  double     setup_acc = 0.0;    // host seconds of one-time slot preparation since the epoch entry point last cleared it
  uint32_t*  grp = nullptr;      // [n_local] attribute -> group (fmx_set_groups); nullptr = one group
  uint32_t   num_groups = 1;
  uint64_t   n_local = 0;
  int        device = 0;
  hipStream_t stream = nullptr;
  Tab        tb = {nullptr, nullptr, 0, 0};   // V rows (+ co-located w), see fmx_kernels.h
  float*     w_sep = nullptr;    // the w[] array
  Arena      arena;              // when arena.va != nullptr both tables are parts of it (nothing to hipFree)
  double*    w0 = nullptr;       // device scalar
  double*    w0_pp = nullptr;    // 8 doubles: ring of bias copies for the overlapped recurrence (hogwild, fused, bias lag)
  hipStream_t stream2 = nullptr; // side stream of the hogwild bias scan
  // device-side hand-off of the bias between the two streams (one-pass batch rule at large batches; fmx_kernels.h)
  bool        handoff = true;               // FMX_HANDOFF=0 in the environment of fmx_create: events instead
  double*     w0_slots = nullptr;           // [w0_slots_cap] bias after every batch of the running epoch
  uint64_t    w0_slots_cap = 0;
  unsigned long long* handoff_ctr = nullptr;   // device: batches whose k_fused has completed (monotonic over the handle's life)
  unsigned long long  handoff_seq = 0;         // host: value of the counter before the running epoch
  uint32_t*   handoff_err = nullptr;        // device: a wait ran into its bound
  uint32_t    handoff_err_host = 0;
  // the parallel-in-time bias recurrence (k_scan_pit): arrival counters of its grid-wide exchanges + the workgroups' slots
  bool        scan_pit = true;              // FMX_SCAN=serial in the environment of fmx_create: the one-wavefront chain (k_scan1 / k_scan) instead
  unsigned long long* pit_ctr = nullptr;    // [PIT_MAX_IT + 1], zeroed on the launch's stream before every launch
  double*     pit_slots = nullptr;          // [2][PIT_MAX_WG][4]
  bool        pit_used = false;             // a launch since the error word was last read
  uint32_t    run_status = 0;               // FMX_STAT_SCAN_* / _EVENT_SYNC / _HANDOFF_TIMEOUT of the running epoch (launch_scan, sgd_epoch_fused)
  int         concurrent = -1;              // do the handle's two streams run concurrently? -1: not probed yet (streams_concurrent)
  unsigned*   probe_flags = nullptr;        // device: 4 words of k_concurrency_probe
  // the XCD-resident epoch (fmx_xcd_kernels.h): small batches as one launch on one die
  unsigned*   xcd_sync = nullptr;           // device: XCD_CTL_WORDS control words + XCD_MAX_MEMBERS barrier words
  bool        xcd = false;                  // FMX_XCD=1 at fmx_create (opt-in: it measured SLOWER than the two launches per batch, profiles/r06_criteo_hops.txt); cleared when a launch could not assemble its members
  uint32_t    xcd_max_batch = 4096;         // FMX_XCD_MAX_BATCH: larger batches are bandwidth, not latency -- they take the chip-wide launches
  unsigned long long* xcd_trace = nullptr;  // FMX_XCD_TRACE=<file>: time stamps of the first batches
  uint32_t    pit_spins = 0;                // bound of a grid-wide exchange's wait (fmx_create: HANDOFF_SPINS, or FMX_DEBUG_PIT_SPINS from the environment)
  int         pit_occ = -1;                 // workgroups of k_scan_pit the device holds at once (occupancy x CUs); -1: not asked yet
  double*     pit_tmp = nullptr;            // bias between the pieces of a batch longer than PIT_MAX_ROWS
  uint64_t    w_version = 1;                // bumped by every entry point that may change a linear weight (a slot's side stream is valid for ONE value)
  int        num_cu = 256;
  double*    acc = nullptr;      // 4 doubles of reduction scratch
  Slot       slots[FMX_MAX_SLOTS];
  float*     partial = nullptr;  // [cap][KP] + [cap]
  float*     mult = nullptr;     // [cap]
  float*     rest = nullptr;     // [cap_rest]
  size_t     cap = 0, cap_rest = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> ev_pool;
  std::vector<hipEvent_t> ev_sync;   // untimed events ordering the two hogwild streams
  std::string err;
  hipDeviceProp_t prop;
  // several GPUs (fmx_comm.hip)
  void*       comm = nullptr;         // ncclComm_t of a one-process-per-GPU job (fmx_comm_init_rank)
  struct fmx_group_s* group = nullptr;
  bool        owns_group = false;     // the 1-handle group fmx_sgd_epoch builds around `comm`
  float*      xbuf[2] = {nullptr, nullptr};   // exchange buffers [batch][KP + 1]
  size_t      xcap = 0;
  hipStream_t stream_comm = nullptr;  // the all-reduce runs here, ordered by ev_x: [which] partial ready, [2 + which] sum ready
  hipEvent_t  ev_x[4] = {};
  std::unordered_map<const void*, int> occ_cache;   // resident_grid: occupancy per kernel on THIS handle's device
  std::unordered_set<const void*> lds_raised;       // kernels whose dynamic-LDS limit was raised on THIS handle's device
  // FMX_SGD_SEQUENTIAL, a conflict-free run as ONE launch (k_run_fused): workgroups of the instance the device holds at once; off after a time-out
  std::unordered_map<const void*, int> run_one_occ;
  bool run_one = true, run_one_used = false;
  unsigned long long* run_slots = nullptr;          // [RUN_ONE_MAX] {tag, rest_e} of a one-launch run's examples
  // small batches of the minibatch rule as ONE launch per batch (k_small_one, fmx_small_kernels.h): {tag, mult_e} and {tag, rest_e} slots; off after a time-out
  unsigned long long* small_slots = nullptr;        // [3 * SMALL_ONE_MAX]: multipliers, rest_e of even / odd batches
  bool small_one = true, small_one_used = false;
};

// ---- helpers shared between the translation units -------------------------------------------------------------
int fail(fmx_handle h, int code, const char* fmt, ...);                 // fmx_core.hip
Hyper make_hyper(const fmx_config& c);
Shard make_shard(const fmx_config& c);
constexpr int FMX_GRID_OVER_SGD = 1 << 16;   // effectively uncapped
constexpr int FMX_GRID_OVER_ALS = 2;
#ifndef FMX_GRID_OVER_DEFAULT
#define FMX_GRID_OVER_DEFAULT FMX_GRID_OVER_SGD      // fmx_als.hip defines it as FMX_GRID_OVER_ALS before including this header
#endif
uint32_t resident_grid(fmx_handle h, const void* kernel, uint64_t n_waves_wanted, int over_default = FMX_GRID_OVER_DEFAULT);
int ensure_scratch(fmx_handle h, size_t batch_cap, size_t rest_cap);
int check_slot(fmx_handle h, int slot, bool need_target);
int slot_in_session(fmx_handle h, int slot, const char* what);
void free_segments(Slot& s);
void free_slot(Slot& s);
int launch_rest(fmx_handle h, const Slot& s, uint64_t row0, uint32_t n, float* rest, hipStream_t st);
int ensure_segments(fmx_handle h, Slot& s, uint32_t B);                 // fmx_sgd.hip
int ensure_coll_mass(fmx_handle h, Slot& s);                             // fmx_core.hip
int ensure_wside(fmx_handle h, Slot& s);                                 // fmx_core.hip
inline void touch_w(fmx_handle h) { if (h) h->w_version++; }
void resolve_batch(const fmx_config& cfg, double coll_mass, uint32_t requested, uint32_t dflt, double curv_scale, fmx_batch_info* out);
constexpr uint32_t FMX_DEFAULT_BATCH = 262144u;                          // fmx_sgd_opts::batch = 0, before the stability cut
int sgd_resolve_batch(fmx_handle h, Slot& s, const fmx_sgd_opts* opts, fmx_batch_info* bi);   // fmx_sgd.hip: + the shards' shares
int sgd_partial_rows(fmx_handle h, const Slot& s, uint64_t row0, uint32_t n_rows, float* S, float* c, hipStream_t st);   // fmx_sgd.hip
int lag_flush(fmx_handle h);                                             // fmx_sgd.hip
int scan_error_check(fmx_handle h);                                      // fmx_sgd.hip: the device's error word after k_scan_pit launches (streams drained)
uint32_t multi_group_size(const Slot& s, int KP);                        // fmx_sgd.hip: examples per wavefront of the short-row kernels (0: rows are long)
bool streams_concurrent(fmx_handle h);                                   // fmx_sgd.hip: probed once per handle
void sgda_free(fmx_handle h);                                            // fmx_sgd.hip
void als_free(fmx_handle h);                                             // fmx_als.hip
enum { GROUP_SINGLE = 0, GROUP_LOOPBACK = 1, GROUP_RCCL = 2 };
struct fmx_group_s {                      // fmx_comm.hip
  std::vector<fmx_handle> hs;
  int kind = GROUP_SINGLE;
  std::vector<void*> comms;               // GROUP_RCCL: one ncclComm_t per handle
  bool owns_comms = false;                // created by fmx_group_create (not borrowed from fmx_comm_init_rank)
  std::vector<hipEvent_t> ev_part;        // loopback: the buffer of shard i is ready
  hipEvent_t ev_sum = nullptr;            // loopback: the sum is ready
  std::string err;
};
int group_allreduce_f64(fmx_group_s* g, const std::vector<double*>& bufs, size_t count);   // fmx_comm.hip: in place, on the shards' streams
void comm_free(fmx_handle h);                                            // fmx_comm.hip: communicator, group membership, exchange buffers
int comm_sgd_epoch(fmx_handle h, int slot, const fmx_sgd_opts* opts, fmx_epoch_stats* stats);   // fmx_comm.hip
int comm_sum_double(fmx_handle h, double* v);                            // fmx_comm.hip

// Device allocations of the library go through fmx_dev_alloc / fmx_dev_free (fmx_core.hip), never through hipMalloc directly:
//   * one classified arena per device outlives its handle (fmx_core.hip, "arena cache") and can hold tens of GB that nothing uses; an
//     allocation that runs out of memory gives that cache back and tries once more (round-4 advisor);
//   * allocations of 768 MiB and more are built from <= 1 GiB physical chunks through the virtual-memory API (hipMemCreate + hipMemMap):
//     once a process has taken and returned a pool of chunks (what the table placement of fmx_create does), a plain hipMalloc of more
//     than 1 GiB takes 0.5 - 1.6 SECONDS on this part (scripts/ubench/alloc_cost.hip, profiles/r06_alloc_cost.txt: 4 GiB 1566 ms, through
//     the virtual-memory API 0.05 ms) -- that was 96 % of the 1.3 - 2.5 s of one-time slot preparation the round-5 verdict found.
// fmx_dev_free takes either kind (and nullptr).
void arena_cache_drop(int device);                                       // fmx_core.hip
hipError_t fmx_dev_alloc_bytes(void** p, size_t bytes);                  // fmx_core.hip
hipError_t fmx_dev_free(void* p);                                        // fmx_core.hip
template <class T> inline hipError_t fmx_dev_alloc(T** p, size_t bytes) { return fmx_dev_alloc_bytes(reinterpret_cast<void**>(p), bytes); }

#define HIPCHK(h, expr)                                                                         \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return fail((h), FMX_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// wave-per-example grids: 4 waves per 256-thread block, capped so that the launch is >> 256 workgroups
// but grid-strides the rest (guide: memory-bound ops, 256 CUs x 8 blocks).
inline uint32_t wave_grid(uint64_t n_waves_wanted) {
  uint64_t blocks = (n_waves_wanted + 3) / 4;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 8) blocks = 256 * 8;
  return (uint32_t)blocks;
}

#define FMX_LAUNCH_WAVES(kfn, waves, st, ...)                                                        \
  do { auto _k = kfn; hipLaunchKernelGGL(_k, dim3(resident_grid(h, (const void*)_k, (waves))), dim3(256), 0, st, __VA_ARGS__); } while (0)

#define KP_SWITCH(KPV, ...)                                            \
  switch (KPV) {                                                       \
    case 1:   { constexpr int KP = 1;   __VA_ARGS__; } break;          \
    case 2:   { constexpr int KP = 2;   __VA_ARGS__; } break;          \
    case 4:   { constexpr int KP = 4;   __VA_ARGS__; } break;          \
    case 8:   { constexpr int KP = 8;   __VA_ARGS__; } break;          \
    case 16:  { constexpr int KP = 16;  __VA_ARGS__; } break;          \
    case 32:  { constexpr int KP = 32;  __VA_ARGS__; } break;          \
    case 64:  { constexpr int KP = 64;  __VA_ARGS__; } break;          \
    case 128: { constexpr int KP = 128; __VA_ARGS__; } break;          \
    case 256: { constexpr int KP = 256; __VA_ARGS__; } break;          \
    case 512: { constexpr int KP = 512; __VA_ARGS__; } break;          \
    case 1024: { constexpr int KP = 1024; __VA_ARGS__; } break;        \
    default: return fail(h, FMX_E_UNSUPPORTED, "num_factor > 1024 is not supported");   \
  }


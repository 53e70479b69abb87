// fmx_api.hip -- C-ABI implementation (include/fmx.h) over the gfx950 kernels in fmx_kernels.h.
// Build: python -m libfm_amd.build   (hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -shared -fPIC)
//
// No CPU fallback lives here: every compute entry point needs a HIP device and fails with FMX_E_HIP
// otherwise.  Nothing in this library links or calls oracle/.
#include "../../include/fmx.h"
#include "fmx_kernels.h"
#include "fmx_als_kernels.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

using namespace fmx;

namespace {

struct Slot {
  Entry*    ent = nullptr;
  uint64_t* row_ptr = nullptr;
  float*    target = nullptr;
  uint32_t  n_rows = 0;
  uint64_t  nnz = 0;
  uint32_t  max_row = 0;
  bool      used = false;
  // per-batch transposed rows for FMX_APPLY_SEGMENTED (built lazily for one batch size)
  uint32_t  seg_B = 0;
  TEntry*   t_ent = nullptr;
  uint32_t* seg_feat = nullptr;
  uint32_t* seg_rel = nullptr;
  uint32_t  nseg = 0;
  std::vector<uint32_t> batch_seg;   // [n_batches+1] first segment of every batch
  std::vector<uint64_t> batch_base;  // [n_batches+1] first entry of every batch
};

thread_local std::string g_create_error = "";

}  // namespace

struct AlsState {
  int       slot = -1;
  EQ*       e = nullptr;          // [N] {residual, current factor's q}
  double*   q = nullptr;          // [KP][N]
  uint8_t*  seen = nullptr;       // [n_local] feature has a training column
  uint32_t* level_list = nullptr; // segments ordered by level
  std::vector<uint32_t> level_ptr;
  uint64_t  iter = 0;
};

struct LagState {                 // FMX_FLAG_BIAS_LAG bookkeeping: w0 lives in w0_pp[step & 1] while active
  bool       active = false;
  uint64_t   step = 0;
  hipEvent_t ev_rest = nullptr, ev_scan[2] = {nullptr, nullptr};
};

struct SgdaState { float* gw = nullptr; float* gv = nullptr; double* reg = nullptr; };

struct fmx_context_s {
  fmx_config cfg;
  AlsState   als;
  SgdaState  sgda;
  LagState   lag;
  int        KP = 1;
  uint64_t   n_local = 0;
  int        device = 0;
  hipStream_t stream = nullptr;
  Tab        tb = {nullptr, nullptr, 0, 0};   // V rows (+ co-located w), see fmx_kernels.h
  float*     w_sep = nullptr;    // separate w[] array (only when FMX_WPAD=0)
  double*    w0 = nullptr;       // device scalar
  double*    w0_pp = nullptr;    // 2 doubles: ping-pong copies of w0 for the overlapped hogwild bias scan
  hipStream_t stream2 = nullptr; // side stream of the hogwild bias scan
  hipStream_t stream3 = nullptr; // second launch stream of the hogwild macro-batches (odd launches)
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
};

namespace {

int fail(fmx_handle h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->err = buf; else g_create_error = buf;
  return code;
}

#define HIPCHK(h, expr)                                                                         \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return fail((h), FMX_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

int next_pow2(int k) { int p = 1; while (p < k) p <<= 1; return p; }

Hyper make_hyper(const fmx_config& c) {
  Hyper h;
  h.lr = (float)c.learn_rate; h.reg0 = (float)c.reg0; h.regw = (float)c.regw; h.regv = (float)c.regv;
  h.min_target = (float)c.min_target; h.max_target = (float)c.max_target;
  h.task = c.task; h.k0 = c.k0; h.k1 = c.k1;
  h.lr_d = c.learn_rate; h.reg0_d = c.reg0; h.regw_d = c.regw; h.regv_d = c.regv;
  h.min_d = c.min_target; h.max_d = c.max_target;
  return h;
}

// wave-per-example grids: 4 waves per 256-thread block, capped so that the launch is >> 256 workgroups
// but grid-strides the rest (guide: memory-bound ops, 256 CUs x 8 blocks).
inline uint32_t wave_grid(uint64_t n_waves_wanted) {
  uint64_t blocks = (n_waves_wanted + 3) / 4;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 8) blocks = 256 * 8;
  return (uint32_t)blocks;
}

// persistent sizing: never launch more workgroups than can be resident (a second, partially filled round of
// equally long grid-stride workgroups is pure tail), never more than the work needs.
uint32_t resident_grid(fmx_handle h, const void* kernel, uint64_t n_waves_wanted) {
  static std::unordered_map<const void*, int> cache;
  auto it = cache.find(kernel);
  int occ;
  if (it == cache.end()) {
    occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, 0) != hipSuccess || occ < 1) occ = 4;
    if (occ > 8) occ = 8;
    cache[kernel] = occ;
  } else {
    occ = it->second;
  }
  uint64_t blocks = (n_waves_wanted + 3) / 4;
  // over-subscribe: more (shorter) workgroups than can be resident let the dispatcher balance the tail;
  // measured better than an exactly-resident persistent grid for these gather kernels
  int over = 2;
  if (const char* e = getenv("FMX_GRID_OVER")) over = atoi(e) > 0 ? atoi(e) : 1;
  const uint64_t cap = (uint64_t)occ * (uint64_t)h->num_cu * (uint64_t)over;
  if (blocks < 1) blocks = 1;
  if (blocks > cap) blocks = cap;
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
    default: return fail(h, FMX_E_UNSUPPORTED, "num_factor > 256 is not supported yet");   \
  }

int ensure_scratch(fmx_handle h, size_t batch_cap, size_t rest_cap) {
  if (batch_cap > h->cap) {
    if (h->partial) hipFree(h->partial);
    if (h->mult) hipFree(h->mult);
    h->partial = nullptr; h->mult = nullptr; h->cap = 0;
    HIPCHK(h, hipMalloc(&h->partial, batch_cap * (size_t)(h->KP + 1) * sizeof(float)));
    HIPCHK(h, hipMalloc(&h->mult, batch_cap * sizeof(float)));
    h->cap = batch_cap;
  }
  if (rest_cap > h->cap_rest) {
    if (h->rest) hipFree(h->rest);
    h->rest = nullptr; h->cap_rest = 0;
    HIPCHK(h, hipMalloc(&h->rest, rest_cap * sizeof(float)));
    h->cap_rest = rest_cap;
  }
  return FMX_OK;
}

int check_slot(fmx_handle h, int slot, bool need_target) {
  if (!h) return FMX_E_ARG;
  if (slot < 0 || slot >= FMX_MAX_SLOTS) return fail(h, FMX_E_ARG, "slot %d out of range", slot);
  if (!h->slots[slot].used) return fail(h, FMX_E_STATE, "slot %d holds no rows (call fmx_upload_rows first)", slot);
  if (need_target && !h->slots[slot].target) return fail(h, FMX_E_STATE, "slot %d was uploaded without targets", slot);
  return FMX_OK;
}

void free_segments(Slot& s) {
  if (s.t_ent) hipFree(s.t_ent);
  if (s.seg_feat) hipFree(s.seg_feat);
  if (s.seg_rel) hipFree(s.seg_rel);
  s.t_ent = nullptr; s.seg_feat = nullptr; s.seg_rel = nullptr; s.seg_B = 0; s.nseg = 0;
  s.batch_seg.clear(); s.batch_base.clear();
}

void free_slot(Slot& s) {
  free_segments(s);
  if (s.ent) hipFree(s.ent);
  if (s.row_ptr) hipFree(s.row_ptr);
  if (s.target) hipFree(s.target);
  s = Slot();
}

// rest[e] (= y-hat - w0) for rows [row0,row0+n) of a slot, single device
int launch_rest(fmx_handle h, const Slot& s, uint64_t row0, uint32_t n, float* rest, hipStream_t st) {
  KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_rowsums<KP, false, true>), n, st,
                                     s.ent, s.row_ptr, row0, n, h->tb, h->cfg.k1, (float*)nullptr, rest));
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

template <int KP, bool ATOMIC>
int launch_fused_zr(fmx_handle h, const Slot& s, const Hyper& hy, uint64_t row0, uint32_t n_rows, hipStream_t st,
                    const double* w0_in, float* rest_out) {
  constexpr int VEC = Map<KP>::VEC, EPI = Map<KP>::EPI;
  const uint32_t need = (s.max_row + EPI - 1) / EPI;      // row slots per lane to keep a whole row in registers
#define FMX_LAUNCH_ZR(ZRV)                                                                                  \
  FMX_LAUNCH_WAVES((k_fused<KP, ZRV, ATOMIC>), n_rows, st, s.ent, s.row_ptr, s.target, row0, \
                   n_rows, h->tb, hy, w0_in, rest_out)
  if constexpr (VEC * 8 <= 128) { if (need <= 8) { FMX_LAUNCH_ZR(8); return FMX_OK; } }
  if constexpr (VEC * 16 <= 128) { if (need <= 16) { FMX_LAUNCH_ZR(16); return FMX_OK; } }
  if constexpr (VEC * 32 <= 128) { if (need <= 32) { FMX_LAUNCH_ZR(32); return FMX_OK; } }
  if constexpr (VEC * 64 <= 128) { if (need <= 64) { FMX_LAUNCH_ZR(64); return FMX_OK; } }
  // rows too long for the register file: the kernel's two-pass branch handles them (ZR = 8 instance)
  FMX_LAUNCH_ZR(8);
#undef FMX_LAUNCH_ZR
  return FMX_OK;
}

}  // namespace

static void als_free(fmx_handle h);
static void sgda_free(fmx_handle h);
static int lag_flush(fmx_handle h);

extern "C" {

int fmx_abi_version(void) { return FMX_ABI_VERSION; }

int fmx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* fmx_last_error(fmx_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int fmx_create(const fmx_config* cfg, fmx_handle* out) {
  if (!cfg || !out) return fail(nullptr, FMX_E_ARG, "fmx_create: null argument");
  *out = nullptr;
  if (cfg->num_attribute == 0) return fail(nullptr, FMX_E_ARG, "num_attribute must be > 0");
  if (cfg->num_attribute > 0xFFFFFFFFull) return fail(nullptr, FMX_E_ARG, "num_attribute must fit uint32 (fm_model.h:51)");
  if (cfg->num_factor < 0) return fail(nullptr, FMX_E_ARG, "num_factor must be >= 0");
  if (cfg->num_factor > 256) return fail(nullptr, FMX_E_UNSUPPORTED, "num_factor > 256 is not supported yet");
  if (cfg->task != FMX_TASK_REGRESSION && cfg->task != FMX_TASK_CLASSIFICATION)
    return fail(nullptr, FMX_E_ARG, "unknown task");                       // fm_learn.h:81 "unknown task"
  if (cfg->shard_world < 1 || cfg->shard_rank < 0 || cfg->shard_rank >= cfg->shard_world)
    return fail(nullptr, FMX_E_ARG, "bad shard_rank/shard_world");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, FMX_E_HIP, "no HIP device available (%s); libfmx has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  int dev = cfg->device;
  if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
  if (dev >= ndev) return fail(nullptr, FMX_E_ARG, "device %d out of range (%d devices)", dev, ndev);

  fmx_handle h = new fmx_context_s();
  h->cfg = *cfg;
  h->device = dev;
  h->KP = next_pow2(std::max(cfg->num_factor, 1));
  const uint64_t n = cfg->num_attribute, W = (uint64_t)cfg->shard_world, R = (uint64_t)cfg->shard_rank;
  h->n_local = (n > R) ? (n - R + W - 1) / W : 0;
  if (h->n_local == 0) h->n_local = 1;
#define CREATE_CHK(expr)                                                                     \
  do { hipError_t _e = (expr); if (_e != hipSuccess) {                                       \
      fail(nullptr, FMX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));               \
      fmx_destroy(h); return FMX_E_HIP; } } while (0)
  CREATE_CHK(hipSetDevice(dev));
  CREATE_CHK(hipGetDeviceProperties(&h->prop, dev));
  CREATE_CHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  CREATE_CHK(hipEventCreate(&h->ev0));
  CREATE_CHK(hipEventCreate(&h->ev1));
  {
    // row layout: default = separate w[] array (FMX_WPAD=0).  FMX_WPAD=<floats> co-locates w behind the factors
    // ([KP factors | w | padding]); measured on MI355X it does NOT pay: HBM fetches 64-B sectors, so a 4-B w
    // costs one sector wherever it lives, and the unaligned rows cost more.
    int wpad = 0;
    if (const char* e = getenv("FMX_WPAD")) wpad = atoi(e);
    if (wpad < 0 || (wpad % 4) != 0) wpad = 0;
    h->tb.rs = (uint32_t)(h->KP + wpad);
    CREATE_CHK(hipMalloc(&h->tb.V, h->n_local * (size_t)h->tb.rs * sizeof(float)));
    CREATE_CHK(hipMemsetAsync(h->tb.V, 0, h->n_local * (size_t)h->tb.rs * sizeof(float), h->stream));
    if (wpad == 0) {
      CREATE_CHK(hipMalloc(&h->w_sep, h->n_local * sizeof(float)));
      CREATE_CHK(hipMemsetAsync(h->w_sep, 0, h->n_local * sizeof(float), h->stream));
      h->tb.w = h->w_sep; h->tb.ws = 1;
    } else {
      h->tb.w = h->tb.V + h->KP; h->tb.ws = h->tb.rs;
    }
  }
  CREATE_CHK(hipMalloc(&h->w0, sizeof(double)));
  CREATE_CHK(hipMalloc(&h->w0_pp, 4 * sizeof(double)));
  {  // the side stream runs the one-workgroup bias recurrence next to chip-filling gathers: give it priority so
     // that its workgroup is placed as soon as any CU has room
    int lo = 0, hi = 0;
    CREATE_CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CREATE_CHK(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, hi));
  }
  CREATE_CHK(hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking));
  h->num_cu = h->prop.multiProcessorCount > 0 ? h->prop.multiProcessorCount : 256;
  CREATE_CHK(hipMalloc(&h->acc, 4 * sizeof(double)));
  CREATE_CHK(hipMemsetAsync(h->w0, 0, sizeof(double), h->stream));
  CREATE_CHK(hipStreamSynchronize(h->stream));
#undef CREATE_CHK
  *out = h;
  return FMX_OK;
}

int fmx_destroy(fmx_handle h) {
  if (!h) return FMX_OK;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  als_free(h);
  sgda_free(h);
  for (auto& s : h->slots) free_slot(s);
  if (h->tb.V) hipFree(h->tb.V);
  if (h->w_sep) hipFree(h->w_sep);
  if (h->w0) hipFree(h->w0);
  if (h->w0_pp) hipFree(h->w0_pp);
  if (h->stream2) hipStreamDestroy(h->stream2);
  if (h->stream3) hipStreamDestroy(h->stream3);
  if (h->acc) hipFree(h->acc);
  if (h->partial) hipFree(h->partial);
  if (h->mult) hipFree(h->mult);
  if (h->rest) hipFree(h->rest);
  for (auto ev : h->ev_pool) hipEventDestroy(ev);
  for (auto ev : h->ev_sync) hipEventDestroy(ev);
  if (h->lag.ev_rest) hipEventDestroy(h->lag.ev_rest);
  if (h->lag.ev_scan[0]) hipEventDestroy(h->lag.ev_scan[0]);
  if (h->lag.ev_scan[1]) hipEventDestroy(h->lag.ev_scan[1]);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
  return FMX_OK;
}

int fmx_get_info(fmx_handle h, fmx_info* out) {
  if (!h || !out) return FMX_E_ARG;
  memset(out, 0, sizeof(*out));
  out->n_local = h->n_local;
  out->k_padded = h->KP;
  out->device = h->device;
  out->bytes_params = h->n_local * (size_t)(h->tb.rs + (h->w_sep ? 1 : 0)) * sizeof(float);
  snprintf(out->device_name, sizeof(out->device_name), "%s", h->prop.name);
  snprintf(out->arch, sizeof(out->arch), "%s", h->prop.gcnArchName);
  return FMX_OK;
}

int fmx_synchronize(fmx_handle h) {
  if (!h) return FMX_E_ARG;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return FMX_OK;
}

// ---------------------------------------------------------------------------------------------
// parameters
// ---------------------------------------------------------------------------------------------
static int stage_params(fmx_handle h, bool to_device, double* w0, double* w, double* v) {
  HIPCHK(h, hipSetDevice(h->device));
  const uint64_t n = h->cfg.num_attribute;
  const int k = h->cfg.num_factor, KP = h->KP;
  const int R = h->cfg.shard_rank, W = h->cfg.shard_world;
  if (to_device) {
    HIPCHK(h, hipMemcpyAsync(h->w0, w0, sizeof(double), hipMemcpyHostToDevice, h->stream));
  } else {
    HIPCHK(h, hipMemcpyAsync(w0, h->w0, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  const uint32_t chunk = (uint32_t)std::min<uint64_t>(n, 1u << 18);
  double* stage = nullptr;
  HIPCHK(h, hipMalloc(&stage, (size_t)chunk * (size_t)std::max(k, 1) * sizeof(double)));
  int rc = FMX_OK;
#define STAGE_CHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    rc = fail(h, FMX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); goto done; } } while (0)
  for (uint64_t j0 = 0; j0 < n; j0 += chunk) {
    const uint32_t cnt = (uint32_t)std::min<uint64_t>(chunk, n - j0);
    if (w) {
      if (to_device) {
        STAGE_CHK(hipMemcpyAsync(stage, w + j0, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_w_in, dim3((cnt + 255) / 256), dim3(256), 0, h->stream, stage, j0, cnt, R, W, h->tb);
      } else {
        if (W > 1) STAGE_CHK(hipMemcpyAsync(stage, w + j0, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_w_out, dim3((cnt + 255) / 256), dim3(256), 0, h->stream, stage, j0, cnt, R, W, h->tb);
        STAGE_CHK(hipMemcpyAsync(w + j0, stage, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      }
      STAGE_CHK(hipStreamSynchronize(h->stream));
    }
    if (v && k > 0) {
      if (to_device) {
        for (int f = 0; f < k; f++)
          STAGE_CHK(hipMemcpyAsync(stage + (size_t)f * cnt, v + (size_t)f * n + j0, cnt * sizeof(double),
                                   hipMemcpyHostToDevice, h->stream));
        const uint64_t total = (uint64_t)cnt * KP;
        hipLaunchKernelGGL(k_stage_in, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, h->stream,
                           stage, j0, cnt, k, KP, R, W, h->tb);
      } else {
        if (W > 1)
          for (int f = 0; f < k; f++)
            STAGE_CHK(hipMemcpyAsync(stage + (size_t)f * cnt, v + (size_t)f * n + j0, cnt * sizeof(double),
                                     hipMemcpyHostToDevice, h->stream));
        const uint64_t total = (uint64_t)cnt * k;
        hipLaunchKernelGGL(k_stage_out, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, h->stream,
                           stage, j0, cnt, k, KP, R, W, h->tb);
        for (int f = 0; f < k; f++)
          STAGE_CHK(hipMemcpyAsync(v + (size_t)f * n + j0, stage + (size_t)f * cnt, cnt * sizeof(double),
                                   hipMemcpyDeviceToHost, h->stream));
      }
      STAGE_CHK(hipStreamSynchronize(h->stream));
    }
  }
  STAGE_CHK(hipGetLastError());
  STAGE_CHK(hipStreamSynchronize(h->stream));
done:
#undef STAGE_CHK
  hipFree(stage);
  return rc;
}

int fmx_set_params(fmx_handle h, double w0, const double* w, const double* v) {
  if (!h) return FMX_E_ARG;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  if (h->cfg.num_factor > 0 && !v) return fail(h, FMX_E_ARG, "fmx_set_params: v is NULL but num_factor > 0");
  double w0c = w0;
  return stage_params(h, true, &w0c, const_cast<double*>(w), const_cast<double*>(v));
}

int fmx_get_params(fmx_handle h, double* w0, double* w, double* v) {
  if (!h || !w0) return FMX_E_ARG;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  return stage_params(h, false, w0, w, v);
}

int fmx_get_param_rows(fmx_handle h, const uint32_t* ids, uint32_t count, double* w_out, double* v_out) {
  if (!h || !ids || !w_out) return FMX_E_ARG;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  const int k = h->cfg.num_factor;
  if (k > 0 && !v_out) return fail(h, FMX_E_ARG, "fmx_get_param_rows: v_out is NULL");
  if (count == 0) return FMX_OK;
  for (uint32_t i = 0; i < count; i++) {
    if (ids[i] >= h->cfg.num_attribute) return fail(h, FMX_E_ARG, "feature id %u >= num_attribute", ids[i]);
    if ((int)(ids[i] % (uint32_t)h->cfg.shard_world) != h->cfg.shard_rank) return fail(h, FMX_E_ARG, "feature id %u is not on this shard", ids[i]);
  }
  HIPCHK(h, hipSetDevice(h->device));
  uint32_t* d_ids = nullptr; double* d_out = nullptr;
  const size_t nout = (size_t)count * (size_t)(k + 1);
  HIPCHK(h, hipMalloc(&d_ids, (size_t)count * 4));
  HIPCHK(h, hipMalloc(&d_out, nout * sizeof(double)));
  hipError_t er = hipMemcpyAsync(d_ids, ids, (size_t)count * 4, hipMemcpyHostToDevice, h->stream);
  if (er == hipSuccess) {
    hipLaunchKernelGGL(k_fetch_rows, dim3((uint32_t)((nout + 255) / 256)), dim3(256), 0, h->stream, d_ids, count, k,
                       h->cfg.shard_world, h->tb, d_out, d_out + count);
    er = hipGetLastError();
  }
  if (er == hipSuccess) er = hipMemcpyAsync(w_out, d_out, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (er == hipSuccess && k > 0) er = hipMemcpyAsync(v_out, d_out + count, (size_t)count * k * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (er == hipSuccess) er = hipStreamSynchronize(h->stream);
  hipFree(d_ids); hipFree(d_out);
  if (er != hipSuccess) return fail(h, FMX_E_HIP, "fmx_get_param_rows: %s", hipGetErrorString(er));
  return FMX_OK;
}

int fmx_get_w0(fmx_handle h, double* w0) {
  if (!h || !w0) return FMX_E_ARG;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(w0, h->w0, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return FMX_OK;
}

int fmx_init_params(fmx_handle h, double init_mean, double init_stdev, uint64_t seed) {
  if (!h) return FMX_E_ARG;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  HIPCHK(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(k_init_params, dim3(256 * 8), dim3(256), 0, h->stream, h->tb, h->n_local,
                     h->cfg.num_factor, h->KP, h->cfg.shard_rank, h->cfg.shard_world, (float)init_mean, init_stdev, seed);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemsetAsync(h->w0, 0, sizeof(double), h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return FMX_OK;
}

// ---------------------------------------------------------------------------------------------
// rows
// ---------------------------------------------------------------------------------------------
int fmx_free_rows(fmx_handle h, int slot) {
  if (!h || slot < 0 || slot >= FMX_MAX_SLOTS) return FMX_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_slot(h->slots[slot]);
  return FMX_OK;
}

int fmx_upload_rows(fmx_handle h, int slot, const void* entries, const uint64_t* row_ptr, const float* target,
                    uint32_t n_rows, uint64_t nnz) {
  if (!h) return FMX_E_ARG;
  if (slot < 0 || slot >= FMX_MAX_SLOTS) return fail(h, FMX_E_ARG, "slot %d out of range", slot);
  if (!row_ptr || (nnz > 0 && !entries)) return fail(h, FMX_E_ARG, "fmx_upload_rows: null entries/row_ptr");
  if (row_ptr[0] != 0 || row_ptr[n_rows] != nnz) return fail(h, FMX_E_ARG, "row_ptr[0] must be 0 and row_ptr[n_rows] == nnz");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_slot(h->slots[slot]);
  const Entry* src = static_cast<const Entry*>(entries);
  const uint64_t n = h->cfg.num_attribute;
  const uint32_t W = (uint32_t)h->cfg.shard_world, R = (uint32_t)h->cfg.shard_rank;
  std::vector<Entry> local_ent;
  std::vector<uint64_t> local_ptr;
  const Entry* up_ent = src;
  const uint64_t* up_ptr = row_ptr;
  uint64_t up_nnz = nnz;
  uint32_t max_row = 0;
  // bound check: the reference asserts id < num_attribute (fm_model.h:112)
  for (uint64_t i = 0; i < nnz; i++)
    if (src[i].id >= n) return fail(h, FMX_E_ARG, "feature id %u >= num_attribute %llu (row entry %llu)", src[i].id,
                                    (unsigned long long)n, (unsigned long long)i);
  if (W > 1) {   // keep this shard's features, ids become local rows (j / world)
    local_ptr.resize((size_t)n_rows + 1);
    local_ent.reserve((size_t)(nnz / W + n_rows));
    for (uint32_t r = 0; r < n_rows; r++) {
      local_ptr[r] = local_ent.size();
      for (uint64_t i = row_ptr[r]; i < row_ptr[r + 1]; i++)
        if (src[i].id % W == R) { Entry e; e.id = src[i].id / W; e.value = src[i].value; local_ent.push_back(e); }
    }
    local_ptr[n_rows] = local_ent.size();
    up_ent = local_ent.data(); up_ptr = local_ptr.data(); up_nnz = local_ent.size();
  }
  for (uint32_t r = 0; r < n_rows; r++) max_row = std::max<uint32_t>(max_row, (uint32_t)(up_ptr[r + 1] - up_ptr[r]));
  Slot s;
  HIPCHK(h, hipMalloc(&s.ent, std::max<uint64_t>(up_nnz, 1) * sizeof(Entry)));
  HIPCHK(h, hipMalloc(&s.row_ptr, ((size_t)n_rows + 1) * sizeof(uint64_t)));
  if (up_nnz) HIPCHK(h, hipMemcpy(s.ent, up_ent, up_nnz * sizeof(Entry), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(s.row_ptr, up_ptr, ((size_t)n_rows + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
  if (target) {
    HIPCHK(h, hipMalloc(&s.target, std::max<uint32_t>(n_rows, 1) * sizeof(float)));
    if (n_rows) HIPCHK(h, hipMemcpy(s.target, target, (size_t)n_rows * sizeof(float), hipMemcpyHostToDevice));
  }
  s.n_rows = n_rows; s.nnz = up_nnz; s.max_row = max_row; s.used = true;
  h->slots[slot] = s;
  return FMX_OK;
}

int fmx_rows_info(fmx_handle h, int slot, uint32_t* n_rows, uint64_t* nnz) {
  int rc = check_slot(h, slot, false);
  if (rc) return rc;
  if (n_rows) *n_rows = h->slots[slot].n_rows;
  if (nnz) *nnz = h->slots[slot].nnz;
  return FMX_OK;
}

int fmx_download_rows(fmx_handle h, int slot, void* entries, uint64_t* row_ptr, float* target) {
  int rc = check_slot(h, slot, false);
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const Slot& s = h->slots[slot];
  if (entries && s.nnz) HIPCHK(h, hipMemcpy(entries, s.ent, s.nnz * sizeof(Entry), hipMemcpyDeviceToHost));
  if (row_ptr) HIPCHK(h, hipMemcpy(row_ptr, s.row_ptr, ((size_t)s.n_rows + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
  if (target && s.target && s.n_rows) HIPCHK(h, hipMemcpy(target, s.target, (size_t)s.n_rows * sizeof(float), hipMemcpyDeviceToHost));
  return FMX_OK;
}

int fmx_synth_rows(fmx_handle h, int slot, uint64_t seed, uint64_t row0, uint32_t n_rows, uint32_t nnz) {
  if (!h) return FMX_E_ARG;
  if (slot < 0 || slot >= FMX_MAX_SLOTS) return fail(h, FMX_E_ARG, "slot %d out of range", slot);
  if (nnz == 0 || n_rows == 0) return fail(h, FMX_E_ARG, "fmx_synth_rows: empty workload");
  const uint64_t n = h->cfg.num_attribute;
  const uint32_t fs = (uint32_t)(n / nnz);
  if (fs == 0) return fail(h, FMX_E_ARG, "fmx_synth_rows: num_attribute < nnz");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_slot(h->slots[slot]);
  const int R = h->cfg.shard_rank, W = h->cfg.shard_world;
  Slot s;
  uint32_t* cnt = nullptr;
  HIPCHK(h, hipMalloc(&cnt, ((size_t)n_rows + 1) * sizeof(uint32_t)));
  HIPCHK(h, hipMemsetAsync(cnt, 0, ((size_t)n_rows + 1) * sizeof(uint32_t), h->stream));
  HIPCHK(h, hipMalloc(&s.row_ptr, ((size_t)n_rows + 1) * sizeof(uint64_t)));
  HIPCHK(h, hipMalloc(&s.target, (size_t)n_rows * sizeof(float)));
  const dim3 grid((n_rows + 255) / 256), block(256);
  hipLaunchKernelGGL(k_synth, grid, block, 0, h->stream, seed, row0, n_rows, nnz, fs, R, W, cnt,
                     (const uint64_t*)nullptr, (Entry*)nullptr, s.target);
  HIPCHK(h, hipGetLastError());
  {  // exclusive prefix sum u32 -> u64 over n_rows+1 items (last = total)
    void* tmp = nullptr; size_t tmp_bytes = 0;
    auto conv = hipcub::TransformInputIterator<uint64_t, hipcub::CastOp<uint64_t>, uint32_t*>(cnt, hipcub::CastOp<uint64_t>());
    HIPCHK(h, hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, conv, s.row_ptr, (int)(n_rows + 1), h->stream));
    HIPCHK(h, hipMalloc(&tmp, tmp_bytes));
    HIPCHK(h, hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, conv, s.row_ptr, (int)(n_rows + 1), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    hipFree(tmp);
  }
  uint64_t total = 0;
  HIPCHK(h, hipMemcpy(&total, s.row_ptr + n_rows, sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMalloc(&s.ent, std::max<uint64_t>(total, 1) * sizeof(Entry)));
  hipLaunchKernelGGL(k_synth, grid, block, 0, h->stream, seed, row0, n_rows, nnz, fs, R, W, (uint32_t*)nullptr,
                     (const uint64_t*)s.row_ptr, s.ent, (float*)nullptr);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  hipFree(cnt);
  s.n_rows = n_rows; s.nnz = total; s.max_row = nnz; s.used = true;
  h->slots[slot] = s;
  return FMX_OK;
}

// ---------------------------------------------------------------------------------------------
// predict / evaluate
// ---------------------------------------------------------------------------------------------
int fmx_predict(fmx_handle h, int slot, double* out) {
  int rc = check_slot(h, slot, false);
  if (rc) return rc;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  if (!out) return fail(h, FMX_E_ARG, "fmx_predict: out is NULL");
  HIPCHK(h, hipSetDevice(h->device));
  const Slot& s = h->slots[slot];
  if (s.n_rows == 0) return FMX_OK;
  rc = ensure_scratch(h, 0, (size_t)s.n_rows * 2);
  if (rc) return rc;
  rc = launch_rest(h, s, 0, s.n_rows, h->rest, h->stream);
  if (rc) return rc;
  float* yhat = h->rest + s.n_rows;
  const int k0 = (h->cfg.shard_world > 1) ? (h->cfg.shard_rank == 0 ? h->cfg.k0 : 0) : h->cfg.k0;
  hipLaunchKernelGGL(k_yhat, dim3(std::min<uint32_t>((s.n_rows + 255) / 256, 2048)), dim3(256), 0, h->stream,
                     h->rest, s.n_rows, k0, h->w0, yhat);
  HIPCHK(h, hipGetLastError());
  std::vector<float> tmp(s.n_rows);
  HIPCHK(h, hipMemcpyAsync(tmp.data(), yhat, (size_t)s.n_rows * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (uint32_t r = 0; r < s.n_rows; r++) out[r] = (double)tmp[r];
  return FMX_OK;
}

int fmx_evaluate(fmx_handle h, int slot, fmx_eval* out) {
  int rc = check_slot(h, slot, true);
  if (rc) return rc;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  if (!out) return fail(h, FMX_E_ARG, "fmx_evaluate: out is NULL");
  if (h->cfg.shard_world > 1) return fail(h, FMX_E_UNSUPPORTED, "fmx_evaluate on a feature shard: use fmx_sgd_partial + all-reduce + fmx_predict_finish");
  HIPCHK(h, hipSetDevice(h->device));
  const Slot& s = h->slots[slot];
  memset(out, 0, sizeof(*out));
  out->rows = s.n_rows;
  if (s.n_rows == 0) return FMX_OK;
  rc = ensure_scratch(h, 0, (size_t)s.n_rows * 2);
  if (rc) return rc;
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  rc = launch_rest(h, s, 0, s.n_rows, h->rest, h->stream);
  if (rc) return rc;
  HIPCHK(h, hipMemsetAsync(h->acc, 0, 4 * sizeof(double), h->stream));
  hipLaunchKernelGGL(k_eval, dim3(std::min<uint32_t>((s.n_rows + 255) / 256, 2048)), dim3(256), 0, h->stream,
                     h->rest, s.target, s.n_rows, make_hyper(h->cfg), h->w0, h->acc);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  double acc[4];
  HIPCHK(h, hipMemcpyAsync(acc, h->acc, sizeof(acc), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  out->device_seconds = ms * 1e-3;
  out->rmse = std::sqrt(acc[0] / s.n_rows);          // fm_learn.h:152
  out->mae = acc[1] / s.n_rows;                      // fm_learn.h:148
  out->accuracy = acc[2] / s.n_rows;                 // fm_learn.h:129
  return FMX_OK;
}

// ---------------------------------------------------------------------------------------------
// SGD
// ---------------------------------------------------------------------------------------------
int fmx_partial_floats(fmx_handle h, uint32_t batch, uint64_t* n_floats) {
  if (!h || !n_floats) return FMX_E_ARG;
  *n_floats = (uint64_t)batch * (uint64_t)(h->KP + 1);
  return FMX_OK;
}

// partial buffer layout: [n_rows][KP] factor sums, then [n_rows] scalars
int fmx_sgd_partial(fmx_handle h, int slot, uint64_t row0, uint32_t n_rows, float* d_partial, void* stream) {
  int rc = check_slot(h, slot, false);
  if (rc) return rc;
  const Slot& s = h->slots[slot];
  if (row0 + n_rows > s.n_rows) return fail(h, FMX_E_ARG, "fmx_sgd_partial: rows [%llu,+%u) outside slot (%u rows)",
                                            (unsigned long long)row0, n_rows, s.n_rows);
  if (!d_partial) return fail(h, FMX_E_ARG, "fmx_sgd_partial: d_partial is NULL");
  if (n_rows == 0) return FMX_OK;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  float* S = d_partial;
  float* c = d_partial + (size_t)n_rows * h->KP;
  KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_rowsums<KP, true, false>), n_rows, st,
                                     s.ent, s.row_ptr, row0, n_rows, h->tb, h->cfg.k1, S, c));
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

// builds the (batch, feature) segments of a slot for batch size B (device radix sort; once per data set)
static int ensure_segments(fmx_handle h, Slot& s, uint32_t B) {
  if (s.seg_B == B && s.t_ent) return FMX_OK;
  free_segments(s);
  const uint64_t nnz = s.nnz;
  const uint32_t n_batches = (s.n_rows + B - 1) / B;
  if (nnz >= (1ull << 31)) return fail(h, FMX_E_UNSUPPORTED, "segmented apply: nnz >= 2^31 in one slot (split the data set)");
  hipStream_t st = h->stream;
  uint64_t *keys_a = nullptr, *keys_b = nullptr, *vals_a = nullptr, *vals_b = nullptr;
  uint32_t *flags = nullptr, *pos = nullptr, *d_batch_seg = nullptr;
  void* tmp = nullptr;
  int rc = FMX_OK;
  const size_t cnt = (size_t)std::max<uint64_t>(nnz, 1);
#define SEG_CHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    rc = fail(h, FMX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); goto done; } } while (0)
  SEG_CHK(hipMalloc(&keys_a, cnt * 8)); SEG_CHK(hipMalloc(&keys_b, cnt * 8));
  SEG_CHK(hipMalloc(&vals_a, cnt * 8)); SEG_CHK(hipMalloc(&vals_b, cnt * 8));
  SEG_CHK(hipMalloc(&flags, cnt * 4)); SEG_CHK(hipMalloc(&pos, cnt * 4));
  SEG_CHK(hipMalloc(&d_batch_seg, ((size_t)n_batches + 1) * 4));
  if (nnz) {
    hipLaunchKernelGGL(k_seg_keys, dim3(wave_grid(s.n_rows)), dim3(256), 0, st, s.ent, s.row_ptr, s.n_rows, B, keys_a, vals_a);
    int bits_batch = 1; while ((1ull << bits_batch) < n_batches) bits_batch++;
    size_t tmp_bytes = 0;
    SEG_CHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys_a, keys_b, vals_a, vals_b, (int)nnz, 0, 32 + bits_batch, st));
    SEG_CHK(hipMalloc(&tmp, tmp_bytes));
    SEG_CHK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys_a, keys_b, vals_a, vals_b, (int)nnz, 0, 32 + bits_batch, st));
    hipLaunchKernelGGL(k_seg_heads, dim3(2048), dim3(256), 0, st, keys_b, nnz, flags);
    SEG_CHK(hipStreamSynchronize(st));
    hipFree(tmp); tmp = nullptr; tmp_bytes = 0;
    SEG_CHK(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, flags, pos, (int)nnz, st));
    SEG_CHK(hipMalloc(&tmp, tmp_bytes));
    SEG_CHK(hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, flags, pos, (int)nnz, st));
    uint32_t nseg = 0;
    SEG_CHK(hipMemcpyAsync(&nseg, pos + (nnz - 1), 4, hipMemcpyDeviceToHost, st));
    SEG_CHK(hipStreamSynchronize(st));
    s.nseg = nseg;
    SEG_CHK(hipMalloc(&s.seg_feat, (size_t)nseg * 4));
    SEG_CHK(hipMalloc(&s.seg_rel, (size_t)nseg * 4));
    hipLaunchKernelGGL(k_seg_fill, dim3(2048), dim3(256), 0, st, keys_b, flags, pos, nnz, s.row_ptr, B, s.seg_feat, s.seg_rel);
    hipLaunchKernelGGL(k_seg_batches, dim3((n_batches + 256) / 256), dim3(256), 0, st, pos, nnz, nseg, s.row_ptr,
                       s.n_rows, B, n_batches, d_batch_seg);
    SEG_CHK(hipGetLastError());
    s.batch_seg.resize((size_t)n_batches + 1);
    SEG_CHK(hipMemcpyAsync(s.batch_seg.data(), d_batch_seg, ((size_t)n_batches + 1) * 4, hipMemcpyDeviceToHost, st));
    SEG_CHK(hipStreamSynchronize(st));
    s.t_ent = reinterpret_cast<TEntry*>(vals_b); vals_b = nullptr;      // payload layout == TEntry
  } else {
    s.nseg = 0;
    s.batch_seg.assign((size_t)n_batches + 1, 0);
    SEG_CHK(hipMalloc(&s.t_ent, 8));
  }
  {  // first entry of every batch (row_ptr sampled at multiples of B)
    std::vector<uint64_t> rp((size_t)s.n_rows + 1);
    SEG_CHK(hipMemcpy(rp.data(), s.row_ptr, rp.size() * 8, hipMemcpyDeviceToHost));
    s.batch_base.resize((size_t)n_batches + 1);
    for (uint32_t b = 0; b <= n_batches; b++) s.batch_base[b] = rp[std::min<uint64_t>((uint64_t)b * B, s.n_rows)];
  }
  s.seg_B = B;
done:
#undef SEG_CHK
  if (keys_a) hipFree(keys_a);
  if (keys_b) hipFree(keys_b);
  if (vals_a) hipFree(vals_a);
  if (vals_b) hipFree(vals_b);
  if (flags) hipFree(flags);
  if (pos) hipFree(pos);
  if (d_batch_seg) hipFree(d_batch_seg);
  if (tmp) hipFree(tmp);
  if (rc) free_segments(s);
  return rc;
}

static int launch_scan(fmx_handle h, const float* rest, const float* target, uint32_t n_rows, uint32_t chunk,
                       const Hyper& hy, float* mult, hipStream_t st, const double* w0_in = nullptr, double* w0_out = nullptr) {
  if (hy.k0) {
    if (mult) hipLaunchKernelGGL(k_scan<true>, dim3(1), dim3(64), 0, st, rest, target, n_rows, chunk, hy,
                                 w0_in ? w0_in : h->w0, w0_out ? w0_out : h->w0, mult);
    else      hipLaunchKernelGGL(k_scan<false>, dim3(1), dim3(64), 0, st, rest, target, n_rows, chunk, hy,
                                 w0_in ? w0_in : h->w0, w0_out ? w0_out : h->w0, mult);
  } else if (mult) {
    hipLaunchKernelGGL(k_mult, dim3(std::min<uint32_t>((n_rows + 255) / 256, 2048)), dim3(256), 0, st, rest, target, n_rows, hy,
                       (const double*)nullptr, mult);
  }
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

// ---- FMX_FLAG_BIAS_LAG: the w0 recurrence of batch b runs on stream2 while the main stream goes on ------------
static int lag_flush(fmx_handle h) {                      // make h->w0 the current bias again
  LagState& L = h->lag;
  if (!L.active) return FMX_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream2));
  HIPCHK(h, hipMemcpy(h->w0, h->w0_pp + (L.step & 1), sizeof(double), hipMemcpyDeviceToDevice));
  L.active = false; L.step = 0;
  return FMX_OK;
}
// call BEFORE producing the rest buffer of this step on `st`; returns which of the two rest buffers to use
static int lag_prepare(fmx_handle h, hipStream_t st, int* slot) {
  LagState& L = h->lag;
  if (!L.ev_rest) {
    HIPCHK(h, hipEventCreateWithFlags(&L.ev_rest, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&L.ev_scan[0], hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&L.ev_scan[1], hipEventDisableTiming));
  }
  if (!L.active) {
    HIPCHK(h, hipMemcpyAsync(h->w0_pp, h->w0, sizeof(double), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->w0_pp + 1, h->w0, sizeof(double), hipMemcpyDeviceToDevice, st));
    L.active = true; L.step = 0;
  } else {
    HIPCHK(h, hipStreamWaitEvent(st, L.ev_scan[L.step & 1], 0));      // scan(step-2) is done with this buffer
  }
  *slot = (int)(L.step & 1);
  return FMX_OK;
}
// call AFTER `rest` is complete on `st`: starts the recurrence on the side stream and leaves the multipliers of
// this batch (batch-start bias) in h->mult on `st`
static int lag_step(fmx_handle h, const float* rest, const float* target, uint32_t n_rows, uint32_t chunk,
                    const Hyper& hy, hipStream_t st) {
  LagState& L = h->lag;
  const uint64_t b = L.step;
  if (hy.k0) {
    HIPCHK(h, hipEventRecord(L.ev_rest, st));
    HIPCHK(h, hipStreamWaitEvent(h->stream2, L.ev_rest, 0));
    int rc = launch_scan(h, rest, target, n_rows, chunk, hy, nullptr, h->stream2, h->w0_pp + (b & 1), h->w0_pp + ((b + 1) & 1));
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(L.ev_scan[b & 1], h->stream2));
    if (b >= 1) HIPCHK(h, hipStreamWaitEvent(st, L.ev_scan[(b + 1) & 1], 0));   // scan(b-1) wrote w0_pp[b & 1]
  }
  hipLaunchKernelGGL(k_mult, dim3(std::min<uint32_t>((n_rows + 255) / 256, 2048)), dim3(256), 0, st, rest, target, n_rows, hy,
                     (const double*)(h->w0_pp + (b & 1)), h->mult);
  HIPCHK(h, hipGetLastError());
  L.step++;
  return FMX_OK;
}

// steps 2 and 3 of the minibatch rule for rows [row0,row0+n_rows); `seg_batch` = batch index when the rows are
// exactly one batch of the slot's segment structure (else -1: per-example apply only)
static int sgd_finish_impl(fmx_handle h, const Slot& s, uint64_t row0, uint32_t n_rows, const float* S,
                           const float* rest, const fmx_sgd_opts* opts, hipStream_t st,
                           hipEvent_t ev_a, hipEvent_t ev_b, int64_t seg_batch) {
  const Hyper hy = make_hyper(h->cfg);
  const uint32_t chunk = (opts && opts->w0_chunk) ? opts->w0_chunk : 256u;
  const bool lag = opts && (opts->flags & FMX_FLAG_BIAS_LAG);
  int rc = lag ? lag_step(h, rest, s.target + row0, n_rows, chunk, hy, st)
               : launch_scan(h, rest, s.target + row0, n_rows, chunk, hy, h->mult, st);
  if (rc) return rc;
  int apply = opts ? opts->apply : FMX_APPLY_DEFAULT;
  if (apply == FMX_APPLY_DEFAULT) apply = FMX_APPLY_SEGMENTED;
  if (apply == FMX_APPLY_SEGMENTED && seg_batch < 0) return fail(h, FMX_E_STATE, "segmented apply needs batch-aligned rows");
  if (ev_a) HIPCHK(h, hipEventRecord(ev_a, st));
  if (apply == FMX_APPLY_SEGMENTED) {
    const uint32_t s0 = s.batch_seg[(size_t)seg_batch], s1 = s.batch_seg[(size_t)seg_batch + 1];
    const uint64_t base = s.batch_base[(size_t)seg_batch];
    const uint32_t bnnz = (uint32_t)(s.batch_base[(size_t)seg_batch + 1] - base);
    const uint32_t nseg = s1 - s0;
    if (nseg) {
      KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_apply_seg<KP, 8>), ((uint64_t)nseg + 63) / 64, st,
                                         s.t_ent + base, s.seg_feat + s0, s.seg_rel + s0, nseg, bnnz, h->tb, hy, S, h->mult));
    }
  } else if (apply == FMX_APPLY_ATOMIC) {
    KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_apply<KP, true>), n_rows, st,
                                       s.ent, s.row_ptr, row0, n_rows, h->tb, hy, S, h->mult));
  } else if (apply == FMX_APPLY_STORE) {
    KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_apply<KP, false>), n_rows, st,
                                       s.ent, s.row_ptr, row0, n_rows, h->tb, hy, S, h->mult));
  } else {
    return fail(h, FMX_E_ARG, "unknown apply mode %d", apply);
  }
  if (ev_b) HIPCHK(h, hipEventRecord(ev_b, st));
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

int fmx_sgd_finish(fmx_handle h, int slot, uint64_t row0, uint32_t n_rows, const float* d_partial,
                   const fmx_sgd_opts* opts, void* stream) {
  int rc = check_slot(h, slot, true);
  if (rc) return rc;
  Slot& s = h->slots[slot];
  if (row0 + n_rows > s.n_rows) return fail(h, FMX_E_ARG, "fmx_sgd_finish: rows outside slot");
  if (!d_partial) return fail(h, FMX_E_ARG, "fmx_sgd_finish: d_partial is NULL");
  if (n_rows == 0) return FMX_OK;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  const bool lag = opts && (opts->flags & FMX_FLAG_BIAS_LAG);
  const uint32_t Bcap = (opts && opts->batch) ? std::max(opts->batch, n_rows) : n_rows;
  rc = ensure_scratch(h, n_rows, (size_t)Bcap * 2);
  if (rc) return rc;
  int rslot = 0;
  if (lag) { rc = lag_prepare(h, st, &rslot); if (rc) return rc; } else { rc = lag_flush(h); if (rc) return rc; }
  float* rest_buf = h->rest + (size_t)rslot * Bcap;
  const float* S = d_partial;
  const float* c = d_partial + (size_t)n_rows * h->KP;
  KP_SWITCH(h->KP, hipLaunchKernelGGL((k_rest_from_partial<KP>), dim3(wave_grid((n_rows + Map<KP>::EPI - 1) / Map<KP>::EPI)),
                                        dim3(256), 0, st, S, c, n_rows, rest_buf));
  HIPCHK(h, hipGetLastError());
  int64_t seg_batch = -1;
  const int apply = opts ? opts->apply : FMX_APPLY_DEFAULT;
  if (apply == FMX_APPLY_DEFAULT || apply == FMX_APPLY_SEGMENTED) {
    // the driver walks the slot in batches of opts->batch rows (the last one may be short)
    const uint32_t B = (opts && opts->batch) ? opts->batch : n_rows;
    if (row0 % B != 0 || (n_rows != B && row0 + n_rows != s.n_rows))
      return fail(h, FMX_E_ARG, "fmx_sgd_finish: rows [%llu,+%u) are not batch %u of the slot", (unsigned long long)row0, n_rows, B);
    rc = ensure_segments(h, h->slots[slot], B);
    if (rc) return rc;
    seg_batch = (int64_t)(row0 / B);
  }
  return sgd_finish_impl(h, s, row0, n_rows, S, rest_buf, opts, st, nullptr, nullptr, seg_batch);
}

int fmx_predict_finish(fmx_handle h, uint32_t n_rows, const float* d_partial, float* d_yhat, void* stream) {
  if (!h || !d_partial || !d_yhat) return FMX_E_ARG;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  if (n_rows == 0) return FMX_OK;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  int rc = ensure_scratch(h, 0, n_rows);
  if (rc) return rc;
  const float* S = d_partial;
  const float* c = d_partial + (size_t)n_rows * h->KP;
  KP_SWITCH(h->KP, hipLaunchKernelGGL((k_rest_from_partial<KP>), dim3(wave_grid((n_rows + Map<KP>::EPI - 1) / Map<KP>::EPI)),
                                        dim3(256), 0, st, S, c, n_rows, h->rest));
  hipLaunchKernelGGL(k_yhat, dim3(std::min<uint32_t>((n_rows + 255) / 256, 2048)), dim3(256), 0, st,
                     h->rest, n_rows, h->cfg.k0, h->w0, d_yhat);
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

int fmx_sgd_epoch(fmx_handle h, int slot, const fmx_sgd_opts* opts, fmx_epoch_stats* stats) {
  int rc = check_slot(h, slot, true);
  if (rc) return rc;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  if (!opts) return fail(h, FMX_E_ARG, "fmx_sgd_epoch: opts is NULL");
  HIPCHK(h, hipSetDevice(h->device));
  Slot& s = h->slots[slot];
  if (stats) memset(stats, 0, sizeof(*stats));
  if (s.n_rows == 0) return FMX_OK;
  if (h->cfg.shard_world > 1 && opts->mode != FMX_SGD_MINIBATCH)
    return fail(h, FMX_E_UNSUPPORTED, "feature-sharded handles train through fmx_sgd_partial / fmx_sgd_finish");
  if (h->cfg.shard_world > 1)
    return fail(h, FMX_E_UNSUPPORTED, "fmx_sgd_epoch on a feature shard: drive fmx_sgd_partial + all-reduce + fmx_sgd_finish");
  const Hyper hy = make_hyper(h->cfg);
  const bool timed = (opts->flags & FMX_FLAG_TIME_MAIN_KERNEL) != 0;
  uint64_t batches = 0, main_launches = 0;
  size_t ev_used = 0;
  auto get_event = [&](hipEvent_t* ev) -> hipError_t {
    if (ev_used == h->ev_pool.size()) { hipEvent_t e; hipError_t er = hipEventCreate(&e); if (er != hipSuccess) return er; h->ev_pool.push_back(e); }
    *ev = h->ev_pool[ev_used++];
    return hipSuccess;
  };
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  if (opts->mode == FMX_SGD_SEQUENTIAL) {
    KP_SWITCH(h->KP, hipLaunchKernelGGL((k_sequential<KP>), dim3(1), dim3(64), 0, h->stream, s.ent, s.row_ptr,
                                          s.target, s.n_rows, h->tb, hy, h->w0));
    HIPCHK(h, hipGetLastError());
    batches = s.n_rows; main_launches = 1;
  } else if (opts->mode == FMX_SGD_HOGWILD) {
    if (opts->apply == FMX_APPLY_SEGMENTED) return fail(h, FMX_E_ARG, "HOGWILD has no segmented apply");
    // rows per launch M: w0 is frozen inside a launch.  The bias recurrence of launch i (k_scan, one wavefront) runs
    // on a side stream WHILE launches i+1, i+2 stream; launch i reads the w0 produced by scan i-3 (a ring of 3
    // slots / rest buffers, so the result does not depend on timing and a slow scan has two launches of slack).
    const uint32_t M = opts->batch ? opts->batch : 262144u;
    const uint32_t chunk = opts->w0_chunk ? opts->w0_chunk : 256u;
    const uint32_t cap = std::min<uint32_t>(M, s.n_rows);
    rc = ensure_scratch(h, 0, (size_t)cap * 3);
    if (rc) return rc;
    const uint64_t n_launch = ((uint64_t)s.n_rows + M - 1) / M;
    while (h->ev_sync.size() < 2 * n_launch + 1) { hipEvent_t e; HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->ev_sync.push_back(e); }
    for (int r = 0; r < 3; r++) HIPCHK(h, hipMemcpyAsync(h->w0_pp + r, h->w0, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    // even launches go to `stream`, odd ones to `stream3`: the drain of one macro-batch overlaps the ramp-up of
    // the next (rows of different launches are as independent as rows of one launch)
    const bool two_streams = getenv("FMX_HOGWILD_TWO_STREAMS") != nullptr;   // +6 % but launches overlap (timing per launch blurs)
    hipEvent_t ev_start = h->ev_sync[2 * n_launch];
    HIPCHK(h, hipEventRecord(ev_start, h->stream));
    if (two_streams) HIPCHK(h, hipStreamWaitEvent(h->stream3, ev_start, 0));
    for (uint64_t i = 0; i < n_launch; i++) {
      const uint64_t row0 = i * M;
      const uint32_t nb = (uint32_t)std::min<uint64_t>(M, s.n_rows - row0);
      float* rest = h->rest + (size_t)(i % 3) * cap;
      hipStream_t fs = (two_streams && (i & 1)) ? h->stream3 : h->stream;
      if (i >= 3) HIPCHK(h, hipStreamWaitEvent(fs, h->ev_sync[2 * (i - 3) + 1], 0));   // scan i-3 done
      hipEvent_t ea = nullptr, eb = nullptr;
      // (no per-launch events here: a timing event between two launches costs ~13 % on this path; the epoch is
      //  bracketed by ev0/ev1 on the launch stream and the average launch time is epoch time / launches)
      main_launches++;
      const double* w0_in = h->w0_pp + ((i + 1) % 3);      // slot written by scan i-3 (initial value for i < 3)
      if (opts->apply == FMX_APPLY_ATOMIC) {
        KP_SWITCH(h->KP, { rc = launch_fused_zr<KP, true>(h, s, hy, row0, nb, fs, w0_in, rest); });
      } else {
        KP_SWITCH(h->KP, { rc = launch_fused_zr<KP, false>(h, s, hy, row0, nb, fs, w0_in, rest); });
      }
      if (rc) return rc;
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipEventRecord(h->ev_sync[2 * i], fs));
      HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_sync[2 * i], 0));
      rc = launch_scan(h, rest, s.target + row0, nb, chunk, hy, nullptr, h->stream2, h->w0_pp + (i % 3), h->w0_pp + ((i + 1) % 3));
      if (rc) return rc;
      HIPCHK(h, hipEventRecord(h->ev_sync[2 * i + 1], h->stream2));
      batches++;
    }
    // stream2 is in order: its last event covers every scan, and scan i waited for launch i
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_sync[2 * (n_launch - 1) + 1], 0));
    if (hy.k0) HIPCHK(h, hipMemcpyAsync(h->w0, h->w0_pp + (n_launch % 3), sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  } else if (opts->mode == FMX_SGD_MINIBATCH) {
    const uint32_t B = opts->batch ? opts->batch : 16384u;
    const bool lag = (opts->flags & FMX_FLAG_BIAS_LAG) != 0;
    const uint32_t Bc = std::min<uint32_t>(B, s.n_rows);
    rc = ensure_scratch(h, (size_t)Bc * (lag ? 2 : 1), 0);
    if (rc) return rc;
    const bool segmented = (opts->apply == FMX_APPLY_DEFAULT || opts->apply == FMX_APPLY_SEGMENTED);
    if (segmented) {
      rc = ensure_segments(h, h->slots[slot], B);
      if (rc) return rc;
      HIPCHK(h, hipEventRecord(h->ev0, h->stream));           // do not bill the one-time bucketing to the epoch
    }
    for (uint64_t row0 = 0; row0 < s.n_rows; row0 += B) {
      const uint32_t nb = (uint32_t)std::min<uint64_t>(B, s.n_rows - row0);
      int pslot = 0;
      if (lag) { rc = lag_prepare(h, h->stream, &pslot); if (rc) return rc; }
      float* S = h->partial + (size_t)pslot * Bc * (size_t)(h->KP + 1);
      float* rest = S + (size_t)nb * h->KP;
      KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_rowsums<KP, true, true>), nb, h->stream,
                                         s.ent, s.row_ptr, row0, nb, h->tb, h->cfg.k1, S, rest));
      hipEvent_t ea = nullptr, eb = nullptr;
      if (timed) { HIPCHK(h, get_event(&ea)); HIPCHK(h, get_event(&eb)); main_launches++; }
      rc = sgd_finish_impl(h, s, row0, nb, S, rest, opts, h->stream, ea, eb, segmented ? (int64_t)(row0 / B) : -1);
      if (rc) return rc;
      batches++;
    }
  } else {
    return fail(h, FMX_E_ARG, "unknown SGD mode %d", opts->mode);
  }
  if (h->lag.active) {     // the last recurrence must finish inside the timed region; then w0 returns to h->w0
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->lag.ev_scan[(h->lag.step + 1) & 1], 0));
  }
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  rc = lag_flush(h);
  if (rc) return rc;
  if (stats) {
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    stats->rows = s.n_rows;
    stats->batches = batches;
    stats->device_seconds = ms * 1e-3;
    if (opts->mode == FMX_SGD_MINIBATCH && timed) {
      double tot = 0;
      for (size_t i = 0; i + 1 < ev_used; i += 2) {
        float m2 = 0;
        HIPCHK(h, hipEventElapsedTime(&m2, h->ev_pool[i], h->ev_pool[i + 1]));
        tot += m2 * 1e-3;
      }
      stats->main_kernel_seconds = tot;
      stats->main_kernel_launches = main_launches;
    } else {
      stats->main_kernel_seconds = stats->device_seconds;
      stats->main_kernel_launches = main_launches;
    }
  }
  return FMX_OK;
}

// ---------------------------------------------------------------------------------------------
// SGDA
// ---------------------------------------------------------------------------------------------
static void sgda_free(fmx_handle h) {
  if (h->sgda.gw) hipFree(h->sgda.gw);
  if (h->sgda.gv) hipFree(h->sgda.gv);
  if (h->sgda.reg) hipFree(h->sgda.reg);
  h->sgda = SgdaState();
}

int fmx_sgda_end(fmx_handle h) {
  if (!h) return FMX_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  sgda_free(h);
  return FMX_OK;
}

int fmx_sgda_begin(fmx_handle h) {
  if (!h) return FMX_E_ARG;
  if (h->cfg.shard_world > 1) return fail(h, FMX_E_UNSUPPORTED, "SGDA on a feature shard is not implemented");
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  sgda_free(h);
  const size_t nv = h->n_local * (size_t)h->tb.rs, nreg = 1 + (size_t)h->KP;
  HIPCHK(h, hipMalloc(&h->sgda.gw, h->n_local * sizeof(float)));
  HIPCHK(h, hipMalloc(&h->sgda.gv, nv * sizeof(float)));
  HIPCHK(h, hipMalloc(&h->sgda.reg, nreg * sizeof(double)));
  HIPCHK(h, hipMemsetAsync(h->sgda.gw, 0, h->n_local * sizeof(float), h->stream));
  HIPCHK(h, hipMemsetAsync(h->sgda.gv, 0, nv * sizeof(float), h->stream));
  HIPCHK(h, hipMemsetAsync(h->sgda.reg, 0, nreg * sizeof(double), h->stream));
  // fm->w.init(0) (:256): the linear weights restart from zero
  if (h->tb.ws == 1) HIPCHK(h, hipMemsetAsync(h->tb.w, 0, h->n_local * sizeof(float), h->stream));
  else HIPCHK(h, hipMemset2DAsync(h->tb.w, (size_t)h->tb.ws * sizeof(float), 0, sizeof(float), h->n_local, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return FMX_OK;
}

int fmx_sgda_get_reg(fmx_handle h, double* reg) {
  if (!h || !reg) return FMX_E_ARG;
  if (!h->sgda.reg) return fail(h, FMX_E_STATE, "fmx_sgda_get_reg before fmx_sgda_begin");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(reg, h->sgda.reg, (1 + (size_t)h->cfg.num_factor) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return FMX_OK;
}

int fmx_sgda_epoch(fmx_handle h, int train_slot, int validation_slot, int do_lambda_steps, fmx_epoch_stats* stats) {
  int rc = check_slot(h, train_slot, true);
  if (rc) return rc;
  rc = check_slot(h, validation_slot, true);
  if (rc) return rc;
  if (!h->sgda.reg) return fail(h, FMX_E_STATE, "fmx_sgda_epoch before fmx_sgda_begin");
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  HIPCHK(h, hipSetDevice(h->device));
  const Slot& s = h->slots[train_slot];
  const Slot& v = h->slots[validation_slot];
  const Hyper hy = make_hyper(h->cfg);
  if (stats) memset(stats, 0, sizeof(*stats));
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  KP_SWITCH(h->KP, hipLaunchKernelGGL((k_sgda<KP>), dim3(1), dim3(64), 0, h->stream, s.ent, s.row_ptr, s.target, s.n_rows,
                                        v.ent, v.row_ptr, v.target, v.n_rows, h->tb, h->sgda.gw, h->sgda.gv, hy, h->w0,
                                        h->sgda.reg, do_lambda_steps));
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (stats) {
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    stats->rows = s.n_rows; stats->batches = s.n_rows; stats->device_seconds = ms * 1e-3;
    stats->main_kernel_seconds = stats->device_seconds; stats->main_kernel_launches = 1;
  }
  return FMX_OK;
}

// ---------------------------------------------------------------------------------------------
// ALS / MCMC
// ---------------------------------------------------------------------------------------------
static void als_free(fmx_handle h) {
  AlsState& a = h->als;
  if (a.e) hipFree(a.e);
  if (a.q) hipFree(a.q);
  if (a.seen) hipFree(a.seen);
  if (a.level_list) hipFree(a.level_list);
  a = AlsState();
}

int fmx_als_end(fmx_handle h) {
  if (!h) return FMX_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  als_free(h);
  return FMX_OK;
}

static int als_eterms(fmx_handle h, const Slot& s, EQ* e, double* q) {
  KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_als_eterms<KP>), s.n_rows, h->stream, s.ent, s.row_ptr, s.n_rows, h->tb,
                                     h->cfg.k0, h->cfg.k1, h->w0, e, q));
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

int fmx_als_begin(fmx_handle h, int train_slot) {
  int rc = check_slot(h, train_slot, true);
  if (rc) return rc;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  if (h->cfg.shard_world > 1) return fail(h, FMX_E_UNSUPPORTED, "ALS on a feature shard is not implemented");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  als_free(h);
  Slot& s = h->slots[train_slot];
  if (s.n_rows == 0) return fail(h, FMX_E_ARG, "fmx_als_begin: empty training set");
  rc = ensure_segments(h, s, s.n_rows);          // one "batch" = the whole data set: X^T with columns in id order
  if (rc) return rc;
  AlsState& a = h->als;
  a.slot = train_slot;
  const uint32_t N = s.n_rows, nseg = s.nseg;
  // ---- dependency levels (host, O(nnz)): level(j) = 1 + max level of earlier features sharing a row with j
  std::vector<uint32_t> seg_feat(nseg), seg_rel(nseg + 1), lvl(nseg), rowlevel(N, 0);
  std::vector<TEntry> tent((size_t)s.nnz);
  if (nseg) {
    HIPCHK(h, hipMemcpy(seg_feat.data(), s.seg_feat, (size_t)nseg * 4, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(seg_rel.data(), s.seg_rel, (size_t)nseg * 4, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(tent.data(), s.t_ent, (size_t)s.nnz * sizeof(TEntry), hipMemcpyDeviceToHost));
  }
  seg_rel[nseg] = (uint32_t)s.nnz;
  uint32_t n_levels = 0;
  for (uint32_t sg = 0; sg < nseg; sg++) {
    uint32_t l = 0;
    for (uint32_t i = seg_rel[sg]; i < seg_rel[sg + 1]; i++) l = std::max(l, rowlevel[tent[i].e]);
    l += 1;
    if (getenv("FMX_ALS_SEQUENTIAL")) l = sg + 1;      // debugging aid: one feature per level (the reference's order, serial)
    for (uint32_t i = seg_rel[sg]; i < seg_rel[sg + 1]; i++) rowlevel[tent[i].e] = l;
    lvl[sg] = l - 1;
    n_levels = std::max(n_levels, l);
  }
  a.level_ptr.assign((size_t)n_levels + 1, 0);
  for (uint32_t sg = 0; sg < nseg; sg++) a.level_ptr[lvl[sg] + 1]++;
  for (uint32_t l = 0; l < n_levels; l++) a.level_ptr[l + 1] += a.level_ptr[l];
  std::vector<uint32_t> list(std::max<uint32_t>(nseg, 1)), fill(a.level_ptr.begin(), a.level_ptr.end());
  for (uint32_t sg = 0; sg < nseg; sg++) list[fill[lvl[sg]]++] = sg;
  std::vector<uint8_t> seen((size_t)h->n_local, 0);
  for (uint32_t sg = 0; sg < nseg; sg++) seen[seg_feat[sg]] = 1;
  HIPCHK(h, hipMalloc(&a.level_list, list.size() * 4));
  HIPCHK(h, hipMemcpy(a.level_list, list.data(), list.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMalloc(&a.seen, seen.size()));
  HIPCHK(h, hipMemcpy(a.seen, seen.data(), seen.size(), hipMemcpyHostToDevice));
  HIPCHK(h, hipMalloc(&a.e, (size_t)N * sizeof(EQ)));
  HIPCHK(h, hipMalloc(&a.q, (size_t)N * (size_t)h->KP * sizeof(double)));
  // ---- first prediction and e -= target (fm_learn_mcmc_simultaneous.h:69-86)
  rc = als_eterms(h, s, a.e, a.q);
  if (rc) return rc;
  hipLaunchKernelGGL(k_als_sub_target, dim3(std::min<uint32_t>((N + 255) / 256, 2048)), dim3(256), 0, h->stream, a.e, s.target, N);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return FMX_OK;
}

int fmx_als_moments(fmx_handle h, double* out) {
  if (!h || !out) return FMX_E_ARG;
  AlsState& a = h->als;
  if (a.slot < 0) return fail(h, FMX_E_STATE, "fmx_als_moments before fmx_als_begin");
  HIPCHK(h, hipSetDevice(h->device));
  const Slot& s = h->slots[a.slot];
  const int k = h->cfg.num_factor;
  const size_t cnt = 4 + 2 * (size_t)k;
  double* d = nullptr;
  HIPCHK(h, hipMalloc(&d, cnt * sizeof(double)));
  hipStream_t st = h->stream;
  HIPCHK(h, hipMemsetAsync(d, 0, cnt * sizeof(double), st));
  const dim3 b1(256), ge(std::min<uint32_t>((s.n_rows + 255) / 256, 2048)), gp((uint32_t)std::min<uint64_t>((h->n_local + 255) / 256, 2048));
  hipLaunchKernelGGL(k_als_sum_e, ge, b1, 0, st, a.e, s.n_rows, d);            // d[0] = sum e, d[1] = sum e^2
  hipLaunchKernelGGL(k_param_moments, gp, b1, 0, st, h->tb.w, h->tb.ws, h->n_local, d + 2);
  for (int f = 0; f < k; f++) hipLaunchKernelGGL(k_param_moments, gp, b1, 0, st, h->tb.V + f, h->tb.rs, h->n_local, d + 4 + 2 * f);
  hipError_t er = hipGetLastError();
  if (er == hipSuccess) er = hipMemcpyAsync(out, d, cnt * sizeof(double), hipMemcpyDeviceToHost, st);
  if (er == hipSuccess) er = hipStreamSynchronize(st);
  hipFree(d);
  if (er != hipSuccess) return fail(h, FMX_E_HIP, "fmx_als_moments: %s", hipGetErrorString(er));
  std::swap(out[0], out[1]);                                                   // documented order: sum e^2 first
  return FMX_OK;
}

int fmx_als_sweep(fmx_handle h, const fmx_als_opts* opts, fmx_als_stats* stats) {
  if (!h || !opts) return FMX_E_ARG;
  AlsState& a = h->als;
  if (a.slot < 0) return fail(h, FMX_E_STATE, "fmx_als_sweep before fmx_als_begin");
  HIPCHK(h, hipSetDevice(h->device));
  const Slot& s = h->slots[a.slot];
  const uint32_t N = s.n_rows;
  const uint32_t n_levels = (uint32_t)a.level_ptr.size() - 1;
  hipStream_t st = h->stream;
  const dim3 g1(std::min<uint32_t>((N + 255) / 256, 2048)), b1(256);
  HIPCHK(h, hipEventRecord(h->ev0, st));
  double acc[4] = {0, 0, 0, 0};
  // sum e, sum e^2 (draw_w0's numerator; draw_alpha's statistic for the caller)
  HIPCHK(h, hipMemsetAsync(h->acc, 0, 4 * sizeof(double), st));
  hipLaunchKernelGGL(k_als_sum_e, g1, b1, 0, st, a.e, N, h->acc);
  HIPCHK(h, hipMemcpyAsync(acc, h->acc, sizeof(acc), hipMemcpyDeviceToHost, st));
  double w0 = 0;
  HIPCHK(h, hipMemcpyAsync(&w0, h->w0, sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (stats) stats->sum_e_sqr = acc[1];
  std::mt19937_64 rng(opts->seed * 0x9E3779B97F4A7C15ull + a.iter + 1);
  std::normal_distribution<double> nd(0.0, 1.0);
  if (h->cfg.k0) {                                         // draw_w0, fm_learn_mcmc.h:643-683 (w0_mean_0 = 0)
    double mean = acc[0] - (double)N * w0;
    const double sigma_sqr = 1.0 / (h->cfg.reg0 + opts->alpha * (double)N);
    mean = -sigma_sqr * (opts->alpha * mean - 0.0 * h->cfg.reg0);
    double nw0 = opts->do_sample ? mean + std::sqrt(sigma_sqr) * nd(rng) : mean;
    if (!(std::isnan(nw0) || std::isinf(nw0))) {
      HIPCHK(h, hipMemcpyAsync(h->w0, &nw0, sizeof(double), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_als_add_const, g1, b1, 0, st, a.e, N, nw0 - w0);
      HIPCHK(h, hipStreamSynchronize(st));                 // nw0 lives on this stack frame
    }
  }
  const uint32_t nseg = s.nseg, nnz = (uint32_t)s.nnz;
  const dim3 gu((uint32_t)std::min<uint64_t>((h->n_local + 255) / 256, 2048));
  // lanes per column from the mean column length (one-hot data: a handful of rows per feature)
  const double avg_col = nseg ? (double)nnz / (double)nseg : 0.0;
  const int G = avg_col <= 5.0 ? 4 : (avg_col <= 12.0 ? 8 : (avg_col <= 40.0 ? 16 : 64));
#define FMX_ALS_DRAW(ISV, cnt, ...)                                                                          \
  do {                                                                                                        \
    if (G == 4)       FMX_LAUNCH_WAVES((k_als_draw<ISV, 4>), ((uint64_t)(cnt) + 15) / 16, st, __VA_ARGS__);    \
    else if (G == 8)  FMX_LAUNCH_WAVES((k_als_draw<ISV, 8>), ((uint64_t)(cnt) + 7) / 8, st, __VA_ARGS__);      \
    else if (G == 16) FMX_LAUNCH_WAVES((k_als_draw<ISV, 16>), ((uint64_t)(cnt) + 3) / 4, st, __VA_ARGS__);     \
    else              FMX_LAUNCH_WAVES((k_als_draw<ISV, 64>), (uint64_t)(cnt), st, __VA_ARGS__);               \
  } while (0)
  if (h->cfg.k1) {                                         // draw_w per level, :454-476
    for (uint32_t l = 0; l < n_levels; l++) {
      const uint32_t cnt = a.level_ptr[l + 1] - a.level_ptr[l];
      if (!cnt) continue;
      FMX_ALS_DRAW(false, cnt, s.t_ent, s.seg_feat, s.seg_rel, nseg, nnz, a.level_list + a.level_ptr[l], cnt,
                   h->tb.w, h->tb.ws, a.e, opts->alpha, opts->w_lambda, opts->w_mu, opts->do_sample,
                   opts->seed, (uint64_t)(a.iter * 1024 + 1000));
    }
    hipLaunchKernelGGL(k_als_unseen, gu, b1, 0, st, a.seen, h->n_local, h->tb.w, h->tb.ws, opts->w_lambda, opts->w_mu,
                       opts->do_sample, opts->seed, (uint64_t)(a.iter * 1024 + 1001));
  }
  for (int f = 0; f < h->cfg.num_factor; f++) {            // per factor: q_f is ready (k_als_eterms), draw_v per level :528-595
    double* qf = a.q + (size_t)f * N;
    const double v_lambda = opts->v_lambda_f ? opts->v_lambda_f[f] : opts->v_lambda;
    const double v_mu = opts->v_mu_f ? opts->v_mu_f[f] : opts->v_mu;
    hipLaunchKernelGGL(k_als_load_q, g1, b1, 0, st, a.e, qf, N);
    for (uint32_t l = 0; l < n_levels; l++) {
      const uint32_t cnt = a.level_ptr[l + 1] - a.level_ptr[l];
      if (!cnt) continue;
      FMX_ALS_DRAW(true, cnt, s.t_ent, s.seg_feat, s.seg_rel, nseg, nnz, a.level_list + a.level_ptr[l], cnt,
                   h->tb.V + f, h->tb.rs, a.e, opts->alpha, v_lambda, v_mu, opts->do_sample,
                   opts->seed, (uint64_t)(a.iter * 1024 + f));
    }
    hipLaunchKernelGGL(k_als_unseen, gu, b1, 0, st, a.seen, h->n_local, h->tb.V + f, h->tb.rs, v_lambda, v_mu,
                       opts->do_sample, opts->seed, (uint64_t)(a.iter * 1024 + 512 + f));
  }
  HIPCHK(h, hipGetLastError());
  // full re-prediction (fm_learn_mcmc_simultaneous.h:122), train metric and new residuals (:139-196)
  int rc = als_eterms(h, s, a.e, a.q);
  if (rc) return rc;
  HIPCHK(h, hipMemsetAsync(h->acc, 0, 4 * sizeof(double), st));
  hipLaunchKernelGGL(k_als_targets, g1, b1, 0, st, a.e, s.target, N, h->cfg.task, h->cfg.min_target, h->cfg.max_target, h->acc,
                     opts->do_sample, opts->seed, (uint64_t)(a.iter * 1024 + 1002));
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev1, st));
  HIPCHK(h, hipMemcpyAsync(acc, h->acc, sizeof(acc), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  a.iter++;
  if (stats) {
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    stats->device_seconds = ms * 1e-3;
    stats->levels = n_levels;
    stats->train_metric = (h->cfg.task == FMX_TASK_REGRESSION) ? std::sqrt(acc[0] / N) : acc[0] / N;
  }
  return FMX_OK;
}

}  // extern "C"

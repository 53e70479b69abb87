// fmx_comm.hip -- C-ABI (include/fmx.h): several GPUs.  Feature shards (one handle per shard) trained by ONE host
// process (the shape libFM has: a single process constructs and calls the learner, libfm.cpp:271-293,415) or by one
// process per GPU; the per-minibatch exchange of the [B][KP + 1] partial sums is an RCCL all-reduce over xGMI, or -- for
// shards that share a device (tests, single-GPU boxes) -- a local reduction kernel ("loopback").
// RCCL is bound at run time (dlopen of librccl.so.1): libfmx.so has no link-time dependency on it, a process that never
// creates a group with more than one device never loads it.
#include "fmx_internal.h"

#include <atomic>
#include <thread>
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl g_rccl;
bool g_rccl_tried = false;
Rccl* rccl() {                                   // bound once per process; nullptr (text in g_rccl.err) when unavailable
  Rccl& r = g_rccl;
  if (g_rccl_tried) return r.lib ? &r : nullptr;
  g_rccl_tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {                  // an already loaded copy (e.g. the one PyTorch ships) wins
    r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (r.lib) break;
  }
  for (size_t i = 0; !r.lib && i < sizeof(names) / sizeof(names[0]); i++) r.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!r.lib) { const char* e = dlerror(); r.err = std::string("cannot load librccl.so.1: ") + (e ? e : "?"); return nullptr; }
#define BIND(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym)); \
  if (!r.field) { r.err = std::string("librccl lacks ") + sym; r.lib = nullptr; return nullptr; }
  BIND(GetUniqueId, "ncclGetUniqueId") BIND(CommInitRank, "ncclCommInitRank") BIND(CommDestroy, "ncclCommDestroy")
  BIND(AllReduce, "ncclAllReduce")
  BIND(GroupStart, "ncclGroupStart") BIND(GroupEnd, "ncclGroupEnd")
  BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
  // the second exchange path's two symbols are optional: only FMX_EXCHANGE_RS_AG needs them (rccl_has_rsag; round-5 advisor)
  r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(dlsym(r.lib, "ncclReduceScatter"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
  return &r;
}
const char* rccl_why() { return g_rccl.err.c_str(); }
bool rccl_has_rsag() { Rccl* R = rccl(); return R && R->ReduceScatter && R->AllGather; }

#define NCCLCHK(h, expr)                                                                                        \
  do { ncclResult_t _r = (expr); if (_r != ncclSuccess)                                                         \
      return fail((h), FMX_E_HIP, "%s failed: %s", #expr, rccl()->GetErrorString(_r)); } while (0)

// loopback exchange: EVERY shard's buffer receives the sum over the shards' buffers (fixed order -> deterministic), the
// in-place semantics of an all-reduce: a shard then reads only its own buffer, so no shard's next gather can overwrite
// what another shard's update is still reading
struct BufList { float* p[16]; int n; };
__global__ void __launch_bounds__(256) k_sum_shards(BufList bufs, size_t n4, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<const float4*>(bufs.p[0])[i];
    for (int r = 1; r < bufs.n; r++) {
      const float4 b = reinterpret_cast<const float4*>(bufs.p[r])[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    for (int r = 0; r < bufs.n; r++) reinterpret_cast<float4*>(bufs.p[r])[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {               // tail (< 4 floats)
    const size_t i = (n & ~(size_t)3) + threadIdx.x;
    float a = bufs.p[0][i];
    for (int r = 1; r < bufs.n; r++) a += bufs.p[r][i];
    for (int r = 0; r < bufs.n; r++) bufs.p[r][i] = a;
  }
}

// the same sum as reduce-scatter + all-gather (fmx_config::exchange_algo = FMX_EXCHANGE_RS_AG on shards that share a device): shard r
// reduces slice r of everybody's buffer into its own (same order of addition as k_sum_shards: bit-identical), then every shard copies
// slice r from shard r.  n4 = float4s per slice.
__global__ void __launch_bounds__(256) k_rs_shards(BufList bufs, size_t n4) {
  for (int r = 0; r < bufs.n; r++)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
      const size_t at = (size_t)r * n4 + i;
      float4 a = reinterpret_cast<const float4*>(bufs.p[0])[at];
      for (int q = 1; q < bufs.n; q++) {
        const float4 b = reinterpret_cast<const float4*>(bufs.p[q])[at];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      reinterpret_cast<float4*>(bufs.p[r])[at] = a;                 // (slice r of shard r only: nobody else's slice r is read after this)
    }
}
__global__ void __launch_bounds__(256) k_ag_shards(BufList bufs, size_t n4) {
  for (int r = 0; r < bufs.n; r++)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
      const size_t at = (size_t)r * n4 + i;
      const float4 a = reinterpret_cast<const float4*>(bufs.p[r])[at];
      for (int q = 0; q < bufs.n; q++) if (q != r) reinterpret_cast<float4*>(bufs.p[q])[at] = a;
    }
}

}  // namespace

static int gfail(fmx_group g, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  if (g) { g->err = buf; if (!g->hs.empty()) g->hs[0]->err = buf; }
  return code;
}
#define GCHK(g, call) do { int _rc = (call); if (_rc != FMX_OK) { (g)->err = fmx_last_error(cur); return _rc; } } while (0)

static int ensure_xbuf(fmx_handle h, size_t floats) {
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->stream_comm) {
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream_comm, hipStreamNonBlocking));
    for (auto& e : h->ev_x) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  if (floats <= h->xcap) return FMX_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream_comm));
  for (auto& b : h->xbuf) { if (b) fmx_dev_free(b); b = nullptr; }
  h->xcap = 0;
  for (auto& b : h->xbuf) HIPCHK(h, fmx_dev_alloc(&b, floats * sizeof(float)));
  h->xcap = floats;
  return FMX_OK;
}

// the sum of one piece [b, b + cnt) over the communicator, on the members' comm streams (callers bracket the events): ncclAllReduce, or --
// fmx_config::exchange_algo = FMX_EXCHANGE_RS_AG and cnt a multiple of the shard count -- ncclReduceScatter + ncclAllGather in place (rank r
// reduces slice r, then gathers the others')
static int rccl_sum_piece(fmx_group g, int which, size_t off, size_t cnt, size_t off2 = 0, size_t cnt2 = 0) {
  Rccl* R = rccl();
  const size_t n = g->hs.size();
  const uint32_t W = (uint32_t)g->hs[0]->cfg.shard_world;
  if (g->hs[0]->cfg.exchange_algo != FMX_EXCHANGE_RS_AG) {          // all-reduce: both pieces of a run in ONE group (one fused launch)
    if (n > 1 || cnt2) NCCLCHK(g->hs[0], R->GroupStart());
    for (size_t i = 0; i < n; i++) {
      fmx_handle h = g->hs[i];
      HIPCHK(h, hipSetDevice(h->device));
      float* b = h->xbuf[which];
      NCCLCHK(h, R->AllReduce(b + off, b + off, cnt, ncclFloat32, ncclSum, (ncclComm_t)g->comms[i], h->stream_comm));
      if (cnt2) NCCLCHK(h, R->AllReduce(b + off2, b + off2, cnt2, ncclFloat32, ncclSum, (ncclComm_t)g->comms[i], h->stream_comm));
    }
    if (n > 1 || cnt2) NCCLCHK(g->hs[0], R->GroupEnd());
    return FMX_OK;
  }
  if (cnt2) { int prc = rccl_sum_piece(g, which, off2, cnt2); if (prc) return prc; }
  const bool rsag = cnt % W == 0 && cnt > 0;
  if (!rsag) {
    if (n > 1) NCCLCHK(g->hs[0], R->GroupStart());
    for (size_t i = 0; i < n; i++) {
      fmx_handle h = g->hs[i];
      HIPCHK(h, hipSetDevice(h->device));
      float* b = h->xbuf[which] + off;
      NCCLCHK(h, R->AllReduce(b, b, cnt, ncclFloat32, ncclSum, (ncclComm_t)g->comms[i], h->stream_comm));
    }
    if (n > 1) NCCLCHK(g->hs[0], R->GroupEnd());
    return FMX_OK;
  }
  const size_t sl = cnt / W;
  if (n > 1) NCCLCHK(g->hs[0], R->GroupStart());
  for (size_t i = 0; i < n; i++) {
    fmx_handle h = g->hs[i];
    HIPCHK(h, hipSetDevice(h->device));
    float* b = h->xbuf[which] + off;
    NCCLCHK(h, R->ReduceScatter(b, b + (size_t)h->cfg.shard_rank * sl, sl, ncclFloat32, ncclSum, (ncclComm_t)g->comms[i], h->stream_comm));
  }
  if (n > 1) NCCLCHK(g->hs[0], R->GroupEnd());
  if (n > 1) NCCLCHK(g->hs[0], R->GroupStart());
  for (size_t i = 0; i < n; i++) {
    fmx_handle h = g->hs[i];
    HIPCHK(h, hipSetDevice(h->device));
    float* b = h->xbuf[which] + off;
    NCCLCHK(h, R->AllGather(b + (size_t)h->cfg.shard_rank * sl, b, sl, ncclFloat32, (ncclComm_t)g->comms[i], h->stream_comm));
  }
  if (n > 1) NCCLCHK(g->hs[0], R->GroupEnd());
  return FMX_OK;
}

// the exchange of one partial buffer: sum over the shards.  exchange_begin is called right after the partial sums were
// enqueued on every shard's stream; exchange_end makes every shard's stream wait for the result (sum_of()).
//   RCCL: in-place all-reduce on the shard's own COMM stream (ordered behind the gather by an event), so that it can run
//         under whatever the compute stream does next (FMX_FLAG_PIPELINE: the update of the previous batch);
//   loopback (all shards on one device): a reduction kernel on shard 0's stream that leaves the sum in every shard's buffer.
// in_stream (small batches, fmx_group_sgd_epoch): no comm stream and no events -- RCCL: the all-reduce is enqueued on the shard's compute stream;
// loopback: every shard's launches of the batch are on shard 0's stream already
static int exchange_begin(fmx_group g, int which, size_t count, bool in_stream = false) {
  const size_t n = g->hs.size();
  if (g->kind == GROUP_RCCL && in_stream) {
    Rccl* R = rccl();
    if (n > 1) NCCLCHK(g->hs[0], R->GroupStart());
    for (size_t i = 0; i < n; i++) {
      fmx_handle h = g->hs[i];
      HIPCHK(h, hipSetDevice(h->device));
      float* b = h->xbuf[which];
      NCCLCHK(h, R->AllReduce(b, b, count, ncclFloat32, ncclSum, (ncclComm_t)g->comms[i], h->stream));
    }
    if (n > 1) NCCLCHK(g->hs[0], R->GroupEnd());
    return FMX_OK;
  }
  if (g->kind == GROUP_RCCL) {
    Rccl* R = rccl();
    for (size_t i = 0; i < n; i++) {
      fmx_handle h = g->hs[i];
      HIPCHK(h, hipSetDevice(h->device));
      HIPCHK(h, hipEventRecord(h->ev_x[which], h->stream));
      HIPCHK(h, hipStreamWaitEvent(h->stream_comm, h->ev_x[which], 0));
    }
    { int prc = rccl_sum_piece(g, which, 0, count); if (prc) return prc; }
    for (size_t i = 0; i < n; i++) {
      fmx_handle h = g->hs[i];
      HIPCHK(h, hipSetDevice(h->device));
      HIPCHK(h, hipEventRecord(h->ev_x[2 + which], h->stream_comm));
    }
  } else if (g->kind == GROUP_LOOPBACK) {
    fmx_handle h0 = g->hs[0];
    HIPCHK(h0, hipSetDevice(h0->device));
    BufList bl; bl.n = (int)n;
    for (size_t i = 0; i < n; i++) {
      bl.p[i] = g->hs[i]->xbuf[which];
      if (i && !in_stream) { HIPCHK(h0, hipEventRecord(g->ev_part[i], g->hs[i]->stream)); HIPCHK(h0, hipStreamWaitEvent(h0->stream, g->ev_part[i], 0)); }
    }
    if (h0->cfg.exchange_algo == FMX_EXCHANGE_RS_AG && count % (4 * n) == 0 && count > 0) {       // the two-phase form of the same sum
      const size_t n4 = count / 4 / n;
      const dim3 grid((unsigned)std::min<size_t>((n4 + 255) / 256, 2048));
      hipLaunchKernelGGL(k_rs_shards, grid, dim3(256), 0, h0->stream, bl, n4);
      hipLaunchKernelGGL(k_ag_shards, grid, dim3(256), 0, h0->stream, bl, n4);
    } else
    hipLaunchKernelGGL(k_sum_shards, dim3((unsigned)std::min<size_t>((count / 4 + 255) / 256 + 1, 2048)), dim3(256), 0, h0->stream,
                       bl, count / 4, count);
    HIPCHK(h0, hipGetLastError());
    if (!in_stream) {
      HIPCHK(h0, hipEventRecord(g->ev_sum, h0->stream));
      for (size_t i = 1; i < n; i++) HIPCHK(h0, hipStreamWaitEvent(g->hs[i]->stream, g->ev_sum, 0));
    }
  }
  return FMX_OK;
}
// RCCL, chunked: the all-reduce of ONE run of rows of the batch buffer (its [rows][KP] factor sums and its [rows] scalars),
// enqueued behind the sums of that run; the next run is summed on the compute stream meanwhile.  `last`: the event the
// update waits for (exchange_end) is recorded behind this chunk.
static int exchange_rows(fmx_group g, int which, size_t off_s, size_t cnt_s, size_t off_c, size_t cnt_c, bool last) {
  const size_t n = g->hs.size();
  Rccl* R = rccl();
  for (size_t i = 0; i < n; i++) {
    fmx_handle h = g->hs[i];
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventRecord(h->ev_x[which], h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->stream_comm, h->ev_x[which], 0));
  }
  { int prc = rccl_sum_piece(g, which, off_s, cnt_s, off_c, cnt_c); if (prc) return prc; }
  if (last)
    for (size_t i = 0; i < n; i++) {
      fmx_handle h = g->hs[i];
      HIPCHK(h, hipSetDevice(h->device));
      HIPCHK(h, hipEventRecord(h->ev_x[2 + which], h->stream_comm));
    }
  return FMX_OK;
}
static int exchange_end(fmx_group g, int which, bool in_stream = false) {
  if (g->kind == GROUP_RCCL && !in_stream)
    for (fmx_handle h : g->hs) { HIPCHK(h, hipSetDevice(h->device)); HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_x[2 + which], 0)); }
  return FMX_OK;
}
// in-place sum of one fp64 buffer per shard (ALS / MCMC: the {e, q} changes of a level, the partial re-prediction),
// enqueued on the shards' own streams.  Loopback: every shard's buffer receives the total (same order of addition for all).
template <class T> struct PtrList { T* p[16]; int n; };
__global__ void __launch_bounds__(256) k_allsum_f64(PtrList<double> bufs, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double a = bufs.p[0][i];
    for (int r = 1; r < bufs.n; r++) a += bufs.p[r][i];
    for (int r = 0; r < bufs.n; r++) bufs.p[r][i] = a;
  }
}
int group_allreduce_f64(fmx_group_s* g, const std::vector<double*>& bufs, size_t count) {
  const size_t n = g->hs.size();
  if (count == 0 || n < 2) return FMX_OK;
  if (g->kind == GROUP_RCCL) {
    Rccl* R = rccl();
    NCCLCHK(g->hs[0], R->GroupStart());
    for (size_t i = 0; i < n; i++) {
      fmx_handle h = g->hs[i];
      HIPCHK(h, hipSetDevice(h->device));
      NCCLCHK(h, R->AllReduce(bufs[i], bufs[i], count, ncclFloat64, ncclSum, (ncclComm_t)g->comms[i], h->stream));
    }
    NCCLCHK(g->hs[0], R->GroupEnd());
  } else if (g->kind == GROUP_LOOPBACK) {
    fmx_handle h0 = g->hs[0];
    HIPCHK(h0, hipSetDevice(h0->device));
    PtrList<double> bl; bl.n = (int)n;
    for (size_t i = 0; i < n; i++) {
      bl.p[i] = bufs[i];
      if (i) { HIPCHK(h0, hipEventRecord(g->ev_part[i], g->hs[i]->stream)); HIPCHK(h0, hipStreamWaitEvent(h0->stream, g->ev_part[i], 0)); }
    }
    hipLaunchKernelGGL(k_allsum_f64, dim3((unsigned)std::min<size_t>((count + 255) / 256, 2048)), dim3(256), 0, h0->stream, bl, count);
    HIPCHK(h0, hipGetLastError());
    HIPCHK(h0, hipEventRecord(g->ev_sum, h0->stream));
    for (size_t i = 1; i < n; i++) HIPCHK(h0, hipStreamWaitEvent(g->hs[i]->stream, g->ev_sum, 0));
  } else {
    return fail(g->hs[0], FMX_E_STATE, "group_allreduce_f64: the group has no exchange");
  }
  return FMX_OK;
}

// one double summed over the ranks of a one-process-per-GPU job (the shards' shares of the rows' collision mass)
int comm_sum_double(fmx_handle h, double* v) {
  if (!h->comm) return FMX_OK;
  Rccl* R = rccl();
  if (!R) return fail(h, FMX_E_UNSUPPORTED, "comm_sum_double: %s", rccl_why());
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(h->acc, v, sizeof(double), hipMemcpyHostToDevice, h->stream));
  NCCLCHK(h, R->AllReduce(h->acc, h->acc, 1, ncclFloat64, ncclSum, (ncclComm_t)h->comm, h->stream));
  HIPCHK(h, hipMemcpyAsync(v, h->acc, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return FMX_OK;
}

static const float* sum_of(fmx_group g, size_t i, int which) { return g->hs[i]->xbuf[which]; }

// a handle leaves: its group is told (a multi-handle group becomes unusable), its communicator and buffers are released
void comm_free(fmx_handle h) {
  hipSetDevice(h->device);
  if (h->group) {
    fmx_group g = h->group;
    if (h->owns_group) { h->group = nullptr; h->owns_group = false; fmx_group_destroy(g); }
    else { for (auto& x : g->hs) if (x == h) x = nullptr; h->group = nullptr; }
  }
  fmx_comm_destroy(h);
  for (auto& b : h->xbuf) { if (b) fmx_dev_free(b); b = nullptr; }
  h->xcap = 0;
  if (h->stream_comm) { hipStreamDestroy(h->stream_comm); h->stream_comm = nullptr; }
  for (auto& e : h->ev_x) { if (e) hipEventDestroy(e); e = nullptr; }
}

// fmx_sgd_epoch on a feature shard: the shard of a one-process-per-GPU job (fmx_comm_init_rank) runs the group schedule
// with its one local member; a member of a multi-handle group must be driven through fmx_group_sgd_epoch
int comm_sgd_epoch(fmx_handle h, int slot, const fmx_sgd_opts* opts, fmx_epoch_stats* stats) {
  if (h->group && !h->owns_group)
    return fail(h, FMX_E_STATE, "this shard belongs to a group: call fmx_group_sgd_epoch");
  if (!h->comm)
    return fail(h, FMX_E_STATE, "fmx_sgd_epoch on a feature shard needs a communicator (fmx_comm_init_rank) or a group (fmx_group_create); "
                                "or drive fmx_sgd_partial + all-reduce + fmx_sgd_finish yourself");
  if (!h->group) {
    fmx_group g = new fmx_group_s();
    g->hs.push_back(h); g->kind = GROUP_RCCL; g->comms.push_back(h->comm);
    h->group = g; h->owns_group = true;
  }
  return fmx_group_sgd_epoch(h->group, slot, opts, stats);
}

extern "C" {

// the ownership rule as host arithmetic (no device): what sharding.py, the CPU tests and host-side bucketing use
int fmx_shard_place(uint64_t num_attribute, int shard_world, int shard_hash, const uint32_t* ids, uint64_t count,
                    int32_t* owner, uint32_t* local_row) {
  if (!ids && count) return FMX_E_ARG;
  if (num_attribute == 0 || num_attribute > 0xFFFFFFFFull || shard_world < 1) return FMX_E_ARG;
  fmx_config c; memset(&c, 0, sizeof(c));
  c.num_attribute = num_attribute; c.shard_world = shard_world; c.shard_hash = shard_hash; c.shard_rank = 0;
  const Shard sh = make_shard(c);
  for (uint64_t i = 0; i < count; i++) {
    if ((uint64_t)ids[i] >= num_attribute) return FMX_E_ARG;
    const uint32_t p = sh.placed(ids[i]);
    if (owner) owner[i] = (int32_t)(p % (uint32_t)shard_world);
    if (local_row) local_row[i] = p / (uint32_t)shard_world;
  }
  return FMX_OK;
}
int fmx_shard_global(uint64_t num_attribute, int shard_world, int shard_hash, int shard_rank, const uint32_t* local_rows,
                     uint64_t count, uint32_t* ids) {
  if ((!local_rows || !ids) && count) return FMX_E_ARG;
  if (num_attribute == 0 || num_attribute > 0xFFFFFFFFull || shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world) return FMX_E_ARG;
  fmx_config c; memset(&c, 0, sizeof(c));
  c.num_attribute = num_attribute; c.shard_world = shard_world; c.shard_hash = shard_hash; c.shard_rank = shard_rank;
  const Shard sh = make_shard(c);
  for (uint64_t i = 0; i < count; i++) {
    if ((uint64_t)local_rows[i] * (uint64_t)shard_world + (uint64_t)shard_rank >= num_attribute) return FMX_E_ARG;
    ids[i] = sh.global(local_rows[i]);
  }
  return FMX_OK;
}

int fmx_comm_unique_id(void* id128) {
  if (!id128) return FMX_E_ARG;
  Rccl* R = rccl();
  if (!R) return fail(nullptr, FMX_E_UNSUPPORTED, "fmx_comm_unique_id: %s", rccl_why());
  static_assert(sizeof(ncclUniqueId) == FMX_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  if (R->GetUniqueId(&id) != ncclSuccess) return fail(nullptr, FMX_E_HIP, "ncclGetUniqueId failed");
  memcpy(id128, &id, sizeof(id));
  return FMX_OK;
}

int fmx_comm_init_rank(fmx_handle h, const void* id128, int rank, int world) {
  if (!h || !id128) return FMX_E_ARG;
  if (h->cfg.shard_rank != rank || h->cfg.shard_world != world)
    return fail(h, FMX_E_ARG, "fmx_comm_init_rank: rank %d of %d but the handle is shard %d of %d", rank, world, h->cfg.shard_rank, h->cfg.shard_world);
  if (h->comm) return fail(h, FMX_E_STATE, "fmx_comm_init_rank: the handle already has a communicator");
  Rccl* R = rccl();
  if (!R) return fail(h, FMX_E_UNSUPPORTED, "fmx_comm_init_rank: %s", rccl_why());
  if (h->cfg.exchange_algo == FMX_EXCHANGE_RS_AG && !rccl_has_rsag())
    return fail(h, FMX_E_UNSUPPORTED, "fmx_comm_init_rank: exchange_algo = FMX_EXCHANGE_RS_AG but librccl lacks ncclReduceScatter / ncclAllGather");
  HIPCHK(h, hipSetDevice(h->device));
  ncclUniqueId id; memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  NCCLCHK(h, R->CommInitRank(&c, world, id, rank));
  h->comm = c;
  {  // every rank must issue the SAME collectives per batch: the ranks' exchange_algo compared in one 16-byte all-reduce (max of {a, -a})
    double v[2] = {(double)h->cfg.exchange_algo, -(double)h->cfg.exchange_algo};
    HIPCHK(h, hipMemcpyAsync(h->acc, v, sizeof(v), hipMemcpyHostToDevice, h->stream));
    NCCLCHK(h, R->AllReduce(h->acc, h->acc, 2, ncclFloat64, ncclMax, c, h->stream));
    HIPCHK(h, hipMemcpyAsync(v, h->acc, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (v[0] != -v[1]) { fmx_comm_destroy(h); return fail(h, FMX_E_ARG, "fmx_comm_init_rank: the ranks disagree on fmx_config::exchange_algo (%g .. %g)", -v[1], v[0]); }
  }
  for (auto& sl : h->slots) sl.coll_mass_world = -1.0;               // (summed over THIS communicator on first use)
  return FMX_OK;
}

int fmx_comm_destroy(fmx_handle h) {
  if (!h) return FMX_E_ARG;
  if (h->comm) {
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    if (Rccl* R = rccl()) R->CommDestroy((ncclComm_t)h->comm);
    h->comm = nullptr;
    for (auto& sl : h->slots) sl.coll_mass_world = -1.0;
  }
  return FMX_OK;
}

int fmx_group_create(fmx_handle* handles, int n, fmx_group* out) {
  if (!handles || !out || n < 1 || n > 16) return fail(nullptr, FMX_E_ARG, "fmx_group_create: 1..16 handles");
  *out = nullptr;
  for (int i = 0; i < n; i++) {
    fmx_handle h = handles[i];
    if (!h) return fail(nullptr, FMX_E_ARG, "fmx_group_create: handle %d is NULL", i);
    if (h->cfg.shard_world != n || h->cfg.shard_rank != i)
      return fail(h, FMX_E_ARG, "fmx_group_create: handle %d must be shard %d of %d (is %d of %d)", i, i, n, h->cfg.shard_rank, h->cfg.shard_world);
    if (h->KP != handles[0]->KP || h->cfg.num_attribute != handles[0]->cfg.num_attribute || h->cfg.shard_hash != handles[0]->cfg.shard_hash)
      return fail(h, FMX_E_ARG, "fmx_group_create: the shards describe different models");
    if (h->cfg.exchange_algo != handles[0]->cfg.exchange_algo)       // (members issuing different collectives would wait for each other for ever)
      return fail(h, FMX_E_ARG, "fmx_group_create: handle %d has exchange_algo %u, handle 0 has %u", i, h->cfg.exchange_algo, handles[0]->cfg.exchange_algo);
    if (h->group) return fail(h, FMX_E_STATE, "fmx_group_create: handle %d already belongs to a group", i);
  }
  fmx_group g = new fmx_group_s();
  g->hs.assign(handles, handles + n);
  bool same_dev = true, distinct = true;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) { if (handles[i]->device != handles[j]->device) same_dev = false; else distinct = false; }
  if (n == 1) {
    g->kind = handles[0]->comm ? GROUP_RCCL : GROUP_SINGLE;           // one local shard of a multi-process job, or no sharding at all
    if (handles[0]->comm) g->comms.push_back(handles[0]->comm);
  } else if (same_dev) {
    g->kind = GROUP_LOOPBACK;
    fmx_handle h0 = handles[0];
    hipSetDevice(h0->device);
    g->ev_part.resize(n, nullptr);
    bool ok = hipEventCreateWithFlags(&g->ev_sum, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < n; i++) ok = hipEventCreateWithFlags(&g->ev_part[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { fmx_group_destroy(g); return fail(h0, FMX_E_HIP, "fmx_group_create: event creation failed"); }
  } else if (distinct) {
    Rccl* R = rccl();
    if (!R) { delete g; return fail(handles[0], FMX_E_UNSUPPORTED, "fmx_group_create: %s", rccl_why()); }
    g->kind = GROUP_RCCL;
    g->comms.assign(n, nullptr);
    g->owns_comms = true;
    ncclUniqueId id;
    ncclResult_t r = R->GetUniqueId(&id);
    if (r == ncclSuccess) r = R->GroupStart();                         // one thread, several devices: ranks initialised as a group
    for (int i = 0; r == ncclSuccess && i < n; i++) {
      hipSetDevice(handles[i]->device);
      { ncclComm_t c = nullptr; r = R->CommInitRank(&c, n, id, i); g->comms[i] = c; }
    }
    if (r == ncclSuccess) r = R->GroupEnd();
    if (r != ncclSuccess) { int rc = fail(handles[0], FMX_E_HIP, "fmx_group_create: RCCL initialisation failed: %s", R->GetErrorString(r)); fmx_group_destroy(g); return rc; }
  } else {
    delete g;
    return fail(handles[0], FMX_E_UNSUPPORTED, "fmx_group_create: the shards must sit on pairwise distinct devices (RCCL) or all on one (loopback)");
  }
  for (int i = 0; i < n; i++) handles[i]->group = g;
  *out = g;
  return FMX_OK;
}

int fmx_group_destroy(fmx_group g) {
  if (!g) return FMX_OK;
  for (auto h : g->hs) if (h) { hipSetDevice(h->device); hipStreamSynchronize(h->stream); if (h->group == g) { h->group = nullptr; h->owns_group = false; } }
  if (g->owns_comms) if (Rccl* R = rccl()) for (auto c : g->comms) if (c) R->CommDestroy((ncclComm_t)c);
  for (auto e : g->ev_part) if (e) hipEventDestroy(e);
  if (g->ev_sum) hipEventDestroy(g->ev_sum);
  delete g;
  return FMX_OK;
}

const char* fmx_group_last_error(fmx_group g) { return g ? g->err.c_str() : fmx_last_error(nullptr); }

// ---- parameters and rows for all shards of a group, the host arrays crossing PCIe once ------------------------------------
// A block of `bytes` staged on the first shard's device reaches shard i's device: the same pointer when they share the device,
// else a buffer there filled by hipMemcpyPeer (xGMI).  Blocking: this is set-up work.
namespace {
struct Staged {
  std::vector<void*> buf;                 // per shard (buf[i] == buf[0] when it shares shard 0's device)
  std::vector<bool> own;
  ~Staged() { }
};
int staged_alloc(fmx_group g, size_t bytes, Staged* st) {
  const size_t n = g->hs.size();
  st->buf.assign(n, nullptr); st->own.assign(n, false);
  for (size_t i = 0; i < n; i++) {
    fmx_handle h = g->hs[i];
    size_t same = n;
    for (size_t j = 0; j < i; j++) if (g->hs[j]->device == h->device) { same = j; break; }
    if (same < n) { st->buf[i] = st->buf[same]; continue; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, fmx_dev_alloc(&st->buf[i], std::max<size_t>(bytes, 8)));
    st->own[i] = true;
  }
  return FMX_OK;
}
void staged_free(fmx_group g, Staged* st) {
  for (size_t i = 0; i < st->buf.size(); i++) if (st->own[i] && st->buf[i]) { hipSetDevice(g->hs[i]->device); fmx_dev_free(st->buf[i]); }
  st->buf.clear(); st->own.clear();
}
// host -> first device, then to every other device that holds a copy of the stage
int staged_fill(fmx_group g, Staged* st, const void* host, size_t bytes, size_t offset = 0) {
  if (!bytes) return FMX_OK;
  fmx_handle h0 = g->hs[0];
  HIPCHK(h0, hipSetDevice(h0->device));
  HIPCHK(h0, hipMemcpy((char*)st->buf[0] + offset, host, bytes, hipMemcpyHostToDevice));
  for (size_t i = 1; i < st->buf.size(); i++)
    if (st->own[i]) HIPCHK(g->hs[i], hipMemcpyPeer((char*)st->buf[i] + offset, g->hs[i]->device, (char*)st->buf[0] + offset, h0->device, bytes));
  return FMX_OK;
}
}  // namespace

int fmx_group_set_params(fmx_group g, double w0, const double* w, const double* v) {
  if (!g) return FMX_E_ARG;
  for (auto m : g->hs) if (!m) return gfail(g, FMX_E_STATE, "a member of the group was destroyed");
  fmx_handle cur = g->hs[0];
  const uint64_t n = cur->cfg.num_attribute;
  const int k = cur->cfg.num_factor, KP = cur->KP;
  if (k > 0 && !v) return gfail(g, FMX_E_ARG, "fmx_group_set_params: v is NULL but num_factor > 0");
  for (auto m : g->hs) { cur = m; GCHK(g, lag_flush(m)); touch_w(m); }   // (new linear weights: every slot's weight side stream is stale -- a
                                                                          //  one-handle group forwards FMX_FLAG_KEEP_WSIDE / predict to its member)
  const uint32_t chunk = (uint32_t)std::min<uint64_t>(n, 1u << 18);
  Staged st;
  int rc = staged_alloc(g, (size_t)chunk * (size_t)std::max(k, 1) * sizeof(double), &st);
  for (uint64_t j0 = 0; j0 < n && rc == FMX_OK; j0 += chunk) {
    const uint32_t cnt = (uint32_t)std::min<uint64_t>(chunk, n - j0);
    if (w) {
      rc = staged_fill(g, &st, w + j0, (size_t)cnt * sizeof(double));
      for (size_t i = 0; i < g->hs.size() && rc == FMX_OK; i++) {
        fmx_handle h = g->hs[i];
        if (hipSetDevice(h->device) != hipSuccess) { rc = FMX_E_HIP; break; }
        hipLaunchKernelGGL(k_w_in, dim3((cnt + 255) / 256), dim3(256), 0, h->stream, (const double*)st.buf[i], j0, cnt, make_shard(h->cfg), h->tb);
      }
      for (size_t i = 0; i < g->hs.size() && rc == FMX_OK; i++) { hipSetDevice(g->hs[i]->device); if (hipStreamSynchronize(g->hs[i]->stream) != hipSuccess) rc = FMX_E_HIP; }
    }
    if (v && k > 0 && rc == FMX_OK) {
      for (int f = 0; f < k && rc == FMX_OK; f++) rc = staged_fill(g, &st, v + (size_t)f * n + j0, (size_t)cnt * sizeof(double), (size_t)f * cnt * sizeof(double));
      for (size_t i = 0; i < g->hs.size() && rc == FMX_OK; i++) {
        fmx_handle h = g->hs[i];
        if (hipSetDevice(h->device) != hipSuccess) { rc = FMX_E_HIP; break; }
        const uint64_t total = (uint64_t)cnt * KP;
        hipLaunchKernelGGL(k_stage_in, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, h->stream, (const double*)st.buf[i], j0, cnt, k, KP, make_shard(h->cfg), h->tb);
      }
      for (size_t i = 0; i < g->hs.size() && rc == FMX_OK; i++) { hipSetDevice(g->hs[i]->device); if (hipStreamSynchronize(g->hs[i]->stream) != hipSuccess) rc = FMX_E_HIP; }
    }
  }
  for (size_t i = 0; i < g->hs.size() && rc == FMX_OK; i++) {
    fmx_handle h = g->hs[i];
    hipSetDevice(h->device);
    if (hipMemcpy(h->w0, &w0, sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = FMX_E_HIP;
  }
  staged_free(g, &st);
  if (rc) return gfail(g, rc, "fmx_group_set_params: staging failed (%s)", g->hs[0]->err.c_str());
  return FMX_OK;
}

int fmx_group_upload_rows(fmx_group g, int slot, const void* entries, const uint64_t* row_ptr, const float* target, uint32_t n_rows, uint64_t nnz) {
  if (!g) return FMX_E_ARG;
  for (auto m : g->hs) if (!m) return gfail(g, FMX_E_STATE, "a member of the group was destroyed");
  fmx_handle cur = g->hs[0];
  if (g->hs.size() == 1 || g->kind == GROUP_SINGLE) { for (auto m : g->hs) { cur = m; GCHK(g, fmx_upload_rows(m, slot, entries, row_ptr, target, n_rows, nnz)); } return FMX_OK; }
  if (slot < 0 || slot >= FMX_MAX_SLOTS) return gfail(g, FMX_E_ARG, "slot %d out of range", slot);
  if (!row_ptr || (nnz > 0 && !entries)) return gfail(g, FMX_E_ARG, "fmx_group_upload_rows: null entries/row_ptr");
  if (row_ptr[0] != 0 || row_ptr[n_rows] != nnz) return gfail(g, FMX_E_ARG, "row_ptr[0] must be 0 and row_ptr[n_rows] == nnz");
  for (auto m : g->hs) { cur = m; GCHK(g, slot_in_session(m, slot, "fmx_group_upload_rows")); }
  Staged se, sp;
  int rc = staged_alloc(g, nnz * sizeof(Entry), &se);
  if (rc == FMX_OK) rc = staged_alloc(g, ((size_t)n_rows + 1) * sizeof(uint64_t), &sp);
  if (rc == FMX_OK) rc = staged_fill(g, &se, entries, nnz * sizeof(Entry));
  if (rc == FMX_OK) rc = staged_fill(g, &sp, row_ptr, ((size_t)n_rows + 1) * sizeof(uint64_t));
  const uint64_t n = g->hs[0]->cfg.num_attribute;
  for (size_t i = 0; i < g->hs.size() && rc == FMX_OK; i++) {
    fmx_handle h = g->hs[i];
    cur = h;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    free_slot(h->slots[slot]);
    Slot s;
    uint32_t* cnt = nullptr; void* tmp = nullptr;
    uint32_t host_flags[2] = {0, 0};                           // [0] longest kept row, [1] bad id seen
    hipError_t er = fmx_dev_alloc(&cnt, ((size_t)n_rows + 3) * sizeof(uint32_t));
    uint32_t* d_flags = cnt ? cnt + n_rows + 1 : nullptr;
    if (er == hipSuccess) er = hipMemsetAsync(cnt, 0, ((size_t)n_rows + 3) * sizeof(uint32_t), h->stream);
    if (er == hipSuccess) er = fmx_dev_alloc(&s.row_ptr, ((size_t)n_rows + 1) * sizeof(uint64_t));
    const dim3 grid(std::min<uint32_t>((n_rows + 255) / 256 + 1, 4096)), block(256);
    const Shard sh = make_shard(h->cfg);
    if (er == hipSuccess) {
      hipLaunchKernelGGL(k_shard_rows, grid, block, 0, h->stream, (const Entry*)se.buf[i], (const uint64_t*)sp.buf[i], n_rows, n, sh, cnt, d_flags, d_flags + 1,
                         (const uint64_t*)nullptr, (Entry*)nullptr);
      er = hipGetLastError();
    }
    if (er == hipSuccess) {                                    // exclusive prefix sum u32 -> u64 over n_rows + 1 items (last = total)
      size_t tmp_bytes = 0;
      auto conv = hipcub::TransformInputIterator<uint64_t, hipcub::CastOp<uint64_t>, uint32_t*>(cnt, hipcub::CastOp<uint64_t>());
      er = hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, conv, s.row_ptr, (int)(n_rows + 1), h->stream);
      if (er == hipSuccess) er = fmx_dev_alloc(&tmp, std::max<size_t>(tmp_bytes, 8));
      if (er == hipSuccess) er = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, conv, s.row_ptr, (int)(n_rows + 1), h->stream);
    }
    uint64_t total = 0;
    if (er == hipSuccess) er = hipMemcpyAsync(&total, s.row_ptr + n_rows, sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream);
    if (er == hipSuccess) er = hipMemcpyAsync(host_flags, d_flags, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
    if (er == hipSuccess) er = hipStreamSynchronize(h->stream);
    if (er == hipSuccess && host_flags[1]) { if (cnt) fmx_dev_free(cnt); if (tmp) fmx_dev_free(tmp); free_slot(s); staged_free(g, &se); staged_free(g, &sp);
      return gfail(g, FMX_E_ARG, "a feature id of the rows is >= num_attribute %llu", (unsigned long long)n); }
    if (er == hipSuccess) er = fmx_dev_alloc(&s.ent, std::max<uint64_t>(total, 1) * sizeof(Entry));
    if (er == hipSuccess) {
      hipLaunchKernelGGL(k_shard_rows, grid, block, 0, h->stream, (const Entry*)se.buf[i], (const uint64_t*)sp.buf[i], n_rows, n, sh, (uint32_t*)nullptr, (uint32_t*)nullptr,
                         (uint32_t*)nullptr, (const uint64_t*)s.row_ptr, s.ent);
      er = hipGetLastError();
    }
    if (er == hipSuccess && target) {
      er = fmx_dev_alloc(&s.target, std::max<uint32_t>(n_rows, 1) * sizeof(float));
      if (er == hipSuccess && n_rows) er = hipMemcpyAsync(s.target, target, (size_t)n_rows * sizeof(float), hipMemcpyHostToDevice, h->stream);
    }
    if (er == hipSuccess) er = hipStreamSynchronize(h->stream);
    if (cnt) fmx_dev_free(cnt);
    if (tmp) fmx_dev_free(tmp);
    if (er != hipSuccess) { free_slot(s); rc = fail(h, FMX_E_HIP, "fmx_group_upload_rows: %s", hipGetErrorString(er)); break; }
    s.n_rows = n_rows; s.nnz = total; s.max_row = host_flags[0]; s.used = true;
    h->slots[slot] = s;
  }
  staged_free(g, &se); staged_free(g, &sp);
  if (rc) { g->err = fmx_last_error(cur); return rc; }
  return FMX_OK;
}

// ---- small batches, one host thread per shard -------------------------------------------------------------------------------------------
// A 512-row batch on a feature shard is three launches of a few microseconds each; issued for eight shards by ONE thread the host is the
// bottleneck twice over (its calls are serial, and on one stream so are the shards' launches).  Here every shard has its own host thread
// and its own stream: per batch the thread enqueues the shard's sums, the exchange, the shard's update.  RCCL: the all-reduce goes onto the
// shard's compute stream from its own thread (one communicator per thread, the usual one-thread-per-device form) -- no host synchronisation
// at all.  Loopback (the shards share a device): shard 0's thread launches the reduction between two host barriers (the events it waits
// for must have been RECORDED by the other threads; they wait for its event after the second).  A thread that fails keeps arriving at the
// barriers and skips its work; the first error is reported.  Opt-in (FMX_GROUP_THREADS=1), see fmx_group_sgd_epoch for what it measured.
namespace {
struct SpinBarrier {
  std::atomic<uint32_t> count{0}, gen{0};
  uint32_t n;
  explicit SpinBarrier(uint32_t n_) : n(n_) {}
  void wait() {
    const uint32_t g = gen.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) { count.store(0, std::memory_order_relaxed); gen.fetch_add(1, std::memory_order_release); }
    else { uint32_t spins = 0; while (gen.load(std::memory_order_acquire) == g) { if (++spins > 4096) std::this_thread::yield(); } }
  }
};
}  // namespace
static int group_small_epoch_threads(fmx_group g, int slot, const fmx_sgd_opts& opts, uint32_t B, uint32_t n_rows, uint64_t n_batch,
                                     uint64_t n_timed) {
  const size_t n = g->hs.size();
  const size_t kp1 = (size_t)g->hs[0]->KP + 1;
  SpinBarrier bar((uint32_t)n);
  std::atomic<int> failed{FMX_OK};
  std::vector<int> rcs(n, FMX_OK);
  const bool loop = g->kind == GROUP_LOOPBACK;
  Rccl* R = loop ? nullptr : rccl();
  auto body = [&](size_t i) {
    fmx_handle h = g->hs[i];
    fmx_handle h0 = g->hs[0];
    int rc = FMX_OK;
    auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == FMX_OK) rc = fail(h, FMX_E_HIP, "%s failed: %s", what, hipGetErrorString(e)); };
    chk(hipSetDevice(h->device), "hipSetDevice");
    for (uint64_t b = 0; b < n_batch; b++) {
      const uint32_t nb = (uint32_t)std::min<uint64_t>(B, n_rows - b * B);
      const int which = (int)(b & 1);
      const bool ok = rc == FMX_OK && failed.load(std::memory_order_relaxed) == FMX_OK;
      if (ok && i == 0 && b < n_timed) chk(hipEventRecord(h0->ev_pool[4 * b + 0], h0->stream), "hipEventRecord");
      if (ok) { const int prc = fmx_sgd_partial(h, slot, b * B, nb, h->xbuf[which], h->stream); if (prc) rc = prc; }
      if (ok && i == 0 && b < n_timed) chk(hipEventRecord(h0->ev_pool[4 * b + 1], h0->stream), "hipEventRecord");
      if (loop) {
        if (ok && rc == FMX_OK && i) chk(hipEventRecord(g->ev_part[i], h->stream), "hipEventRecord");
        if (rc && failed.load() == FMX_OK) failed.store(rc);
        bar.wait();                                              // every shard's "sums are enqueued" event has been recorded
        if (i == 0 && failed.load() == FMX_OK) {
          BufList bl; bl.n = (int)n;
          for (size_t q = 0; q < n; q++) { bl.p[q] = g->hs[q]->xbuf[which]; if (q) chk(hipStreamWaitEvent(h0->stream, g->ev_part[q], 0), "hipStreamWaitEvent"); }
          const size_t count = (size_t)nb * kp1;
          hipLaunchKernelGGL(k_sum_shards, dim3((unsigned)std::min<size_t>((count / 4 + 255) / 256 + 1, 2048)), dim3(256), 0, h0->stream, bl, count / 4, count);
          chk(hipGetLastError(), "k_sum_shards");
          chk(hipEventRecord(g->ev_sum, h0->stream), "hipEventRecord");
          if (rc && failed.load() == FMX_OK) failed.store(rc);
        }
        bar.wait();                                              // the "sum is ready" event has been recorded
        if (rc == FMX_OK && failed.load() == FMX_OK && i) chk(hipStreamWaitEvent(h->stream, g->ev_sum, 0), "hipStreamWaitEvent");
      } else if (ok && rc == FMX_OK) {
        float* xb = h->xbuf[which];
        const ncclResult_t nr = R->AllReduce(xb, xb, (size_t)nb * kp1, ncclFloat32, ncclSum, (ncclComm_t)g->comms[i], h->stream);
        if (nr != ncclSuccess) rc = fail(h, FMX_E_HIP, "ncclAllReduce failed: %s", R->GetErrorString(nr));
      }
      if (rc == FMX_OK && failed.load(std::memory_order_relaxed) == FMX_OK) {
        if (i == 0 && b < n_timed) chk(hipEventRecord(h0->ev_pool[4 * b + 2], h0->stream), "hipEventRecord");
        const int frc = fmx_sgd_finish(h, slot, b * B, nb, h->xbuf[which], &opts, h->stream);
        if (frc) rc = frc;
        if (i == 0 && b < n_timed) chk(hipEventRecord(h0->ev_pool[4 * b + 3], h0->stream), "hipEventRecord");
      }
      if (rc && failed.load() == FMX_OK) failed.store(rc);
    }
    rcs[i] = rc;
  };
  std::vector<std::thread> th;
  th.reserve(n - 1);
  for (size_t i = 1; i < n; i++) th.emplace_back(body, i);
  body(0);
  for (auto& t : th) t.join();
  for (size_t i = 0; i < n; i++) if (rcs[i]) { g->err = g->hs[i]->err; return rcs[i]; }
  return FMX_OK;
}

// one epoch of the minibatch rule over feature shards:  per batch  partial sums on every shard -> ONE exchange ->
// multipliers / bias recurrence (redundantly on every shard) + update of the local rows.
//   exact (default): the batch rule of oracle fmo_sgd_epoch_minibatch_ex -- identical, shard count aside, to what a single
//                    unsharded handle computes with FMX_APPLY_SEGMENTED / FMX_APPLY_FUSED and the same bias_lag;
//   FMX_FLAG_PIPELINE: the sums of batch b+1 are gathered BEFORE the update of batch b lands, so that their exchange runs
//                    under that update ("one batch stale", oracle fmo_sgd_epoch_minibatch_pipelined).
int fmx_group_sgd_epoch(fmx_group g, int slot, const fmx_sgd_opts* opts_in, fmx_epoch_stats* stats) {
  if (!g || !opts_in) return FMX_E_ARG;
  for (auto h : g->hs) if (!h) return gfail(g, FMX_E_STATE, "a member of the group was destroyed");
  const size_t n = g->hs.size();
  fmx_handle cur = g->hs[0];
  if (stats) memset(stats, 0, sizeof(*stats));
  if (g->kind == GROUP_SINGLE) { GCHK(g, fmx_sgd_epoch(cur, slot, opts_in, stats)); return FMX_OK; }
  if (opts_in->mode != FMX_SGD_MINIBATCH) return gfail(g, FMX_E_UNSUPPORTED, "feature shards train with FMX_SGD_MINIBATCH (the split step)");
  for (size_t i = 0; i < n; i++) {
    cur = g->hs[i];
    GCHK(g, check_slot(cur, slot, true));
    if (cur->slots[slot].n_rows != g->hs[0]->slots[slot].n_rows) return gfail(g, FMX_E_STATE, "the shards hold different numbers of rows in slot %d", slot);
  }
  fmx_sgd_opts opts = *opts_in;
  if (opts.apply == FMX_APPLY_FUSED) opts.apply = FMX_APPLY_DEFAULT;             // the split step's own choice (sgd_finish_impl)
  if (opts_in->apply == FMX_APPLY_FUSED) opts.flags |= FMX_FLAG_BIAS_LAG;       // FUSED implies the lag on one device: same rule here
  const bool pipeline = (opts.flags & FMX_FLAG_PIPELINE) != 0;
  const uint32_t n_rows = g->hs[0]->slots[slot].n_rows;
  for (auto m : g->hs) { m->setup_acc = 0.0; m->run_status = 0; }
  fmx_batch_info bi;                                              // the same batch as one unsharded handle would choose
  cur = g->hs[0];
  GCHK(g, sgd_resolve_batch(cur, cur->slots[slot], opts_in, &bi));
  if ((opts_in->flags & FMX_FLAG_REJECT_UNSTABLE) && (bi.status & FMX_STAT_UNSTABLE))
    return gfail(g, FMX_E_ARG, "batch %u on these rows: learn_rate * curvature * batch * collision mass = %.3g > 2 -- the batch rule diverges",
                 bi.batch, bi.batch_gain);
  const uint32_t B = bi.batch;
  opts.batch = B;
  const size_t kp1 = (size_t)g->hs[0]->KP + 1;
  const size_t cap = (size_t)std::min<uint32_t>(B, std::max<uint32_t>(n_rows, 1)) * kp1;
  for (size_t i = 0; i < n; i++) { cur = g->hs[i]; GCHK(g, ensure_xbuf(cur, cap)); GCHK(g, ensure_segments(cur, cur->slots[slot], B)); }
  fmx_handle h0 = g->hs[0];
  HIPCHK(h0, hipSetDevice(h0->device));
  HIPCHK(h0, hipEventRecord(h0->ev0, h0->stream));
  const uint64_t n_batch = ((uint64_t)n_rows + B - 1) / B;
  auto rows_of = [&](uint64_t b) { return (uint32_t)std::min<uint64_t>(B, n_rows - b * B); };
  // RCCL: the batch is summed and exchanged in runs of rows, so that the wire works while the next run is being summed (the sums of
  // a row do not depend on the other rows of the batch: nothing changes in the rule).  fmx_config::exchange_runs: runs per batch
  // (1 = one exchange per batch, as the loopback exchange always does).
  const uint32_t xchunks = g->hs[0]->cfg.exchange_runs ? g->hs[0]->cfg.exchange_runs : 4u;
  // small batches (BASELINE configs[2] as it is worded: Criteo-shaped rows over 8 shards run 512 rows per batch): a batch is a few microseconds
  // of work per shard, so every call the host makes for it counts.  In-stream schedule: one launch for the shard's sums, the exchange
  // without a comm stream or events (RCCL: on the compute stream; loopback: every shard's launches go to shard 0's stream, in order), two
  // launches for the update (sgd_finish_impl): 3 launches (+ 1 collective call) per shard and batch where the general schedule makes ~20 calls.
  // FMX_GROUP_IN_STREAM=0 keeps the general schedule.
  const bool in_stream_ok = []() { const char* e = getenv("FMX_GROUP_IN_STREAM"); return !(e && e[0] == '0'); }();   // (read per epoch: tests switch it)
  bool small = in_stream_ok && B < 32768u && (opts.flags & FMX_FLAG_BIAS_LAG) && !pipeline && !(opts.flags & FMX_FLAG_TIME_MAIN_KERNEL && false);
  if (small)
    for (auto m : g->hs) small = small && multi_group_size(m->slots[slot], m->KP) != 0 && (opts.apply == FMX_APPLY_DEFAULT || opts.apply == FMX_APPLY_FUSED);
  auto stream_of = [&](size_t i) -> void* { return (small && g->kind == GROUP_LOOPBACK) ? (void*)h0->stream : (void*)g->hs[i]->stream; };
  auto gather = [&](uint64_t b) -> int {
    const uint32_t nb = rows_of(b);
    const int which = (int)(b & 1);
    if (small) {
      for (size_t i = 0; i < n; i++) { cur = g->hs[i]; GCHK(g, fmx_sgd_partial(cur, slot, b * B, nb, cur->xbuf[which], stream_of(i))); }
      return exchange_begin(g, which, (size_t)nb * kp1, true);
    }
    if (g->kind == GROUP_RCCL && xchunks > 1 && nb >= 64 * xchunks) {
      const uint32_t step = ((nb + xchunks - 1) / xchunks + 63u) & ~63u;
      for (uint32_t r0 = 0; r0 < nb; r0 += step) {
        const uint32_t nr = std::min(step, nb - r0);
        const size_t off_s = (size_t)r0 * (kp1 - 1), off_c = (size_t)nb * (kp1 - 1) + r0;
        for (size_t i = 0; i < n; i++) {
          cur = g->hs[i];
          HIPCHK(cur, hipSetDevice(cur->device));
          GCHK(g, sgd_partial_rows(cur, cur->slots[slot], b * B + r0, nr, cur->xbuf[which] + off_s, cur->xbuf[which] + off_c, cur->stream));
        }
        int erc = exchange_rows(g, which, off_s, (size_t)nr * (kp1 - 1), off_c, nr, r0 + nr >= nb);
        if (erc) return erc;
      }
      return FMX_OK;
    }
    for (size_t i = 0; i < n; i++) { cur = g->hs[i]; GCHK(g, fmx_sgd_partial(cur, slot, b * B, nb, cur->xbuf[which], cur->stream)); }
    return exchange_begin(g, which, (size_t)nb * kp1);
  };
  // FMX_FLAG_TIME_MAIN_KERNEL: where a batch's time goes on the first local shard (fmx_epoch_stats::phase_seconds) -- four timing
  // events per batch on its compute stream, read after the epoch (at most the first 1024 batches)
  const bool timed = (opts.flags & FMX_FLAG_TIME_MAIN_KERNEL) != 0;
  const uint64_t n_timed = timed ? std::min<uint64_t>(n_batch, 1024) : 0;
  while (h0->ev_pool.size() < 4 * n_timed) { hipEvent_t e; HIPCHK(h0, hipEventCreate(&e)); h0->ev_pool.push_back(e); }
  auto mark = [&](uint64_t b, int which) -> int {
    if (b >= n_timed) return FMX_OK;
    HIPCHK(h0, hipSetDevice(h0->device));
    HIPCHK(h0, hipEventRecord(h0->ev_pool[4 * b + which], h0->stream));
    return FMX_OK;
  };
  auto gather_timed = [&](uint64_t b) -> int {
    int erc = mark(b, 0);
    if (erc == FMX_OK) erc = gather(b);
    if (erc == FMX_OK) erc = mark(b, 1);
    return erc;
  };
  auto update = [&](uint64_t b) -> int {
    int erc = exchange_end(g, (int)(b & 1), small);
    if (erc) return erc;
    erc = mark(b, 2);
    if (erc) return erc;
    for (size_t i = 0; i < n; i++) { cur = g->hs[i]; GCHK(g, fmx_sgd_finish(cur, slot, b * B, rows_of(b), sum_of(g, i, (int)(b & 1)), &opts, stream_of(i))); }
    return mark(b, 3);
  };
  int rc = FMX_OK;
  // opt-in (FMX_GROUP_THREADS=1): on one device the runtime serialises the eight threads' calls and the step got SLOWER -- 8 loopback shards of
  // BASELINE configs[2]: 1.16 M examples/s threaded against 2.28 M from one thread in one stream (1.08 M for the general schedule; round 6, call 17)
  const bool threads_ok = []() { const char* e = getenv("FMX_GROUP_THREADS"); return e && e[0] == '1'; }();
  const bool threaded = small && threads_ok && n > 1;
  if (threaded) {
    rc = group_small_epoch_threads(g, slot, opts, B, n_rows, n_batch, n_timed);
    HIPCHK(h0, hipSetDevice(h0->device));
  }
  if (n_batch && pipeline) rc = gather_timed(0);
  for (uint64_t b = 0; b < n_batch && rc == FMX_OK && !threaded; b++) {
    if (pipeline) { if (b + 1 < n_batch) rc = gather_timed(b + 1); }      // reads the parameters before update(b): one batch stale
    else rc = gather_timed(b);
    if (rc == FMX_OK) rc = update(b);
  }
  if (rc) { g->err = g->hs[0]->err; return rc; }
  for (size_t i = 0; i < n; i++) { cur = g->hs[i]; GCHK(g, fmx_synchronize(cur)); }       // drains the side streams, bias back in h->w0
  HIPCHK(h0, hipSetDevice(h0->device));
  HIPCHK(h0, hipEventRecord(h0->ev1, h0->stream));
  HIPCHK(h0, hipEventSynchronize(h0->ev1));
  if (stats) {
    float ms = 0;
    HIPCHK(h0, hipEventElapsedTime(&ms, h0->ev0, h0->ev1));
    stats->rows = n_rows; stats->batches = n_batch; stats->device_seconds = ms * 1e-3;
    stats->main_kernel_seconds = stats->device_seconds; stats->main_kernel_launches = n_batch;
    stats->max_feature_count = h0->slots[slot].max_seg_count;
    stats->batch_used = bi.batch; stats->collision_mass = bi.collision_mass; stats->batch_gain = bi.batch_gain; stats->status = bi.status;
    stats->w0_chunk_used = opts.w0_chunk ? opts.w0_chunk : fmx_default_w0_chunk(h0->cfg.learn_rate, h0->cfg.task);
    for (auto m : g->hs) { stats->setup_seconds = std::max(stats->setup_seconds, m->setup_acc); stats->status |= m->run_status; }
    for (uint64_t b = 0; b < n_timed; b++) {
      float a = 0, x = 0, u = 0;
      HIPCHK(h0, hipEventElapsedTime(&a, h0->ev_pool[4 * b], h0->ev_pool[4 * b + 1]));
      // exposed exchange = the wait right in front of update(b).  Exact schedule: from the end of the batch's own sums.  Pipelined:
      // gather(b + 1) is enqueued between update(b - 1) and update(b), so the wait starts where THAT ends (mark 1 of b + 1) -- measured
      // from mark 1 of b it would contain the whole update of batch b - 1 (round-3 advisor finding)
      hipEvent_t from = h0->ev_pool[4 * b + 1];
      if (pipeline) {
        if (b + 1 < n_timed && b + 1 < n_batch) from = h0->ev_pool[4 * (b + 1) + 1];
        else if (b > 0) from = h0->ev_pool[4 * (b - 1) + 3];
      }
      HIPCHK(h0, hipEventElapsedTime(&x, from, h0->ev_pool[4 * b + 2]));
      if (x < 0.f) x = 0.f;
      HIPCHK(h0, hipEventElapsedTime(&u, h0->ev_pool[4 * b + 2], h0->ev_pool[4 * b + 3]));
      stats->phase_seconds[0] += a * 1e-3; stats->phase_seconds[1] += x * 1e-3; stats->phase_seconds[2] += u * 1e-3;
    }
  }
  return FMX_OK;
}

// y-hat of every row of a slot over the shards (raw, like fmx_predict): partial sums -> exchange -> finish on shard 0
int fmx_group_predict(fmx_group g, int slot, double* out) {
  if (!g || !out) return FMX_E_ARG;
  for (auto m : g->hs) if (!m) return gfail(g, FMX_E_STATE, "a member of the group was destroyed");
  fmx_handle cur = g->hs[0];
  if (g->kind == GROUP_SINGLE) { GCHK(g, fmx_predict(cur, slot, out)); return FMX_OK; }
  const size_t n = g->hs.size();
  for (size_t i = 0; i < n; i++) { cur = g->hs[i]; GCHK(g, check_slot(cur, slot, false)); GCHK(g, lag_flush(cur)); }
  fmx_handle h0 = g->hs[0];
  const uint32_t n_rows = h0->slots[slot].n_rows;
  const size_t kp1 = (size_t)h0->KP + 1;
  const uint32_t chunk = 1u << 18;
  std::vector<float> tmp(std::min<uint32_t>(chunk, std::max<uint32_t>(n_rows, 1)));
  for (size_t i = 0; i < n; i++) { cur = g->hs[i]; GCHK(g, ensure_xbuf(cur, (size_t)tmp.size() * kp1)); }
  float* d_y = nullptr;
  HIPCHK(h0, hipSetDevice(h0->device));
  HIPCHK(h0, fmx_dev_alloc(&d_y, tmp.size() * sizeof(float)));
  int rc = FMX_OK;
  for (uint64_t r0 = 0; r0 < n_rows && rc == FMX_OK; r0 += chunk) {
    const uint32_t nb = (uint32_t)std::min<uint64_t>(chunk, n_rows - r0);
    for (size_t i = 0; i < n && rc == FMX_OK; i++) { cur = g->hs[i]; rc = fmx_sgd_partial(cur, slot, r0, nb, cur->xbuf[0], cur->stream); }
    if (rc == FMX_OK) rc = exchange_begin(g, 0, (size_t)nb * kp1);
    if (rc == FMX_OK) rc = exchange_end(g, 0);
    if (rc == FMX_OK) rc = fmx_predict_finish(h0, nb, sum_of(g, 0, 0), d_y, h0->stream);
    if (rc == FMX_OK && hipMemcpyAsync(tmp.data(), d_y, (size_t)nb * sizeof(float), hipMemcpyDeviceToHost, h0->stream) != hipSuccess) rc = FMX_E_HIP;
    for (size_t i = 0; i < n && rc == FMX_OK; i++) { hipSetDevice(g->hs[i]->device); if (hipStreamSynchronize(g->hs[i]->stream) != hipSuccess) rc = FMX_E_HIP; }
    if (rc == FMX_OK) for (uint32_t r = 0; r < nb; r++) out[r0 + r] = (double)tmp[r];
  }
  hipSetDevice(h0->device);
  fmx_dev_free(d_y);
  if (rc) { g->err = fmx_last_error(cur); return rc; }
  return FMX_OK;
}

// fm_learn::evaluate (fm_learn.h:93-153) over the shards: predictions as above, metric on the host like the reference
int fmx_group_evaluate(fmx_group g, int slot, fmx_eval* out) {
  if (!g || !out) return FMX_E_ARG;
  for (auto m : g->hs) if (!m) return gfail(g, FMX_E_STATE, "a member of the group was destroyed");
  fmx_handle h0 = g->hs[0];
  if (g->kind == GROUP_SINGLE) { fmx_handle cur = h0; GCHK(g, fmx_evaluate(cur, slot, out)); return FMX_OK; }
  { fmx_handle cur = h0; GCHK(g, check_slot(cur, slot, true)); }
  const Slot& s = h0->slots[slot];
  memset(out, 0, sizeof(*out));
  out->rows = s.n_rows;
  if (s.n_rows == 0) return FMX_OK;
  std::vector<double> p(s.n_rows);
  std::vector<float> y(s.n_rows);
  int rc = fmx_group_predict(g, slot, p.data());
  if (rc) return rc;
  HIPCHK(h0, hipSetDevice(h0->device));
  HIPCHK(h0, hipMemcpy(y.data(), s.target, (size_t)s.n_rows * sizeof(float), hipMemcpyDeviceToHost));
  double se = 0, ae = 0, nc = 0;
  for (uint32_t r = 0; r < s.n_rows; r++) {
    if (h0->cfg.task == FMX_TASK_REGRESSION) {
      const double pc = std::max(h0->cfg.min_target, std::min(h0->cfg.max_target, p[r]));       // fm_learn.h:138-139
      const double e = pc - (double)y[r];
      se += e * e; ae += std::fabs(e);
    } else if ((p[r] >= 0 && y[r] >= 0) || (p[r] < 0 && y[r] < 0)) nc += 1;                   // fm_learn.h:118
  }
  out->rmse = std::sqrt(se / s.n_rows); out->mae = ae / s.n_rows; out->accuracy = nc / s.n_rows;
  return FMX_OK;
}

}  // extern "C"

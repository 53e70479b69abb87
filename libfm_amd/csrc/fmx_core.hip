// fmx_core.hip -- C-ABI (include/fmx.h): lifetime, parameters, rows, predict / evaluate.
// Build: python -m libfm_amd.build   (hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -shared -fPIC)
//
// No CPU fallback lives in this library: every compute entry point needs a HIP device and fails with FMX_E_HIP
// otherwise.  Nothing here links or calls oracle/.
#include "fmx_internal.h"
#include <atomic>
#include <mutex>

static thread_local std::string g_create_error = "";

int fail(fmx_handle h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->err = buf; else g_create_error = buf;
  return code;
}

static int next_pow2(int k) { int p = 1; while (p < k) p <<= 1; return p; }

Hyper make_hyper(const fmx_config& c) {
  Hyper h;
  h.lr = (float)c.learn_rate; h.reg0 = (float)c.reg0; h.regw = (float)c.regw; h.regv = (float)c.regv;
  h.min_target = (float)c.min_target; h.max_target = (float)c.max_target;
  h.task = c.task; h.k0 = c.k0; h.k1 = c.k1;
  h.lr_d = c.learn_rate; h.reg0_d = c.reg0; h.regw_d = c.regw; h.regv_d = c.regv;
  h.min_d = c.min_target; h.max_d = c.max_target;
  h.sgda = 0;
  return h;
}

Shard make_shard(const fmx_config& c) {
  Shard sh;
  sh.n = c.num_attribute; sh.rank = (uint32_t)c.shard_rank; sh.world = (uint32_t)c.shard_world;
  sh.hashed = (c.shard_hash != 0 && c.shard_world > 1) ? 1u : 0u;
  uint32_t bits = 1; while (bits < 32 && (1ull << bits) < c.num_attribute) bits++;
  sh.half_bits = std::max<uint32_t>(1u, (bits + 1) / 2);         // 2 * half_bits >= bits: the Feistel domain covers [0, n)
  return sh;
}

// grid sizing: never more workgroups than the work needs, and at most `over` x what can be resident (the kernels stride
// over their work).  Which `over` is a measured per-family choice (scripts/gpu_ab_inprocess.py FMX_GRID_OVER a b <what>,
// alternating inside one process): the row-gather kernels of the SGD family want NO cap -- one workgroup per four examples and
// the dispatcher balancing them beats grid-stride workgroups (one-pass step: x1 -7 %, x2 0, x8 +1.7 %, uncapped +3.0 %; two-pass
// +5.9 %, hogwild +4.7 %, predict +3.2 %) -- while the column kernels of the ALS sweep lose 8 % uncapped and keep x2.
uint32_t resident_grid(fmx_handle h, const void* kernel, uint64_t n_waves_wanted, int over_default) {
  auto& cache = h->occ_cache;                 // per handle (= per device, one calling thread): no process-wide state
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
  const int over = over_default;
  const uint64_t cap = (uint64_t)occ * (uint64_t)h->num_cu * (uint64_t)over;
  if (blocks < 1) blocks = 1;
  if (blocks > cap) blocks = cap;
  return (uint32_t)blocks;
}
int ensure_scratch(fmx_handle h, size_t batch_cap, size_t rest_cap) {
  if (batch_cap > h->cap) {
    if (h->partial) fmx_dev_free(h->partial);
    if (h->mult) fmx_dev_free(h->mult);
    h->partial = nullptr; h->mult = nullptr; h->cap = 0;
    HIPCHK(h, fmx_dev_alloc(&h->partial, batch_cap * (size_t)(h->KP + 1) * sizeof(float)));
    HIPCHK(h, fmx_dev_alloc(&h->mult, batch_cap * sizeof(float)));
    h->cap = batch_cap;
  }
  if (rest_cap > h->cap_rest) {
    if (h->rest) fmx_dev_free(h->rest);
    h->rest = nullptr; h->cap_rest = 0;
    HIPCHK(h, fmx_dev_alloc(&h->rest, rest_cap * sizeof(float)));
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

// an open ALS / MCMC session keeps device structures derived from its train slot (X^T segments, level lists, e/q of
// that size): the slot must not be replaced, freed or re-bucketed until fmx_als_end
int slot_in_session(fmx_handle h, int slot, const char* what) {
  if (h->als.slot >= 0 && h->als.slot == slot)
    return fail(h, FMX_E_STATE, "%s: slot %d is the train slot of an open ALS / MCMC session (call fmx_als_end first)", what, slot);
  return FMX_OK;
}

void free_segments(Slot& s) {
  if (s.t_ent) fmx_dev_free(s.t_ent);
  if (s.seg_feat) fmx_dev_free(s.seg_feat);
  if (s.seg_rel) fmx_dev_free(s.seg_rel);
  if (s.cmask) fmx_dev_free(s.cmask);
  if (s.cseg) fmx_dev_free(s.cseg);
  if (s.cdesc) fmx_dev_free(s.cdesc);
  if (s.d_batch_seg) fmx_dev_free(s.d_batch_seg);
  if (s.d_cbatch) fmx_dev_free(s.d_cbatch);
  if (s.d_batch_base) fmx_dev_free(s.d_batch_base);
  s.cdesc = nullptr; s.d_batch_seg = nullptr; s.d_cbatch = nullptr; s.d_batch_base = nullptr;
  s.t_ent = nullptr; s.seg_feat = nullptr; s.seg_rel = nullptr; s.seg_B = 0; s.nseg = 0;
  s.cmask = nullptr; s.cseg = nullptr; s.ncseg = 0; s.fused_cap = 0;
  s.batch_seg.clear(); s.batch_base.clear(); s.cbatch.clear();
}

void free_block(BlockRows* b) {
  if (!b) return;
  free_slot(b->rows);
  if (b->map) fmx_dev_free(b->map);
  if (b->brow_ptr) fmx_dev_free(b->brow_ptr);
  if (b->brow_list) fmx_dev_free(b->brow_list);
  if (b->pbuf) fmx_dev_free(b->pbuf);
  delete b;
}

void free_slot(Slot& s) {
  for (BlockRows* b : s.blocks) free_block(b);
  s.blocks.clear();
  free_segments(s);
  if (s.ent) fmx_dev_free(s.ent);
  if (s.row_ptr) fmx_dev_free(s.row_ptr);
  if (s.target) fmx_dev_free(s.target);
  if (s.wside) fmx_dev_free(s.wside);
  if (s.lmask) fmx_dev_free(s.lmask);
  s = Slot();
}

// the slot's weight side stream (row_sums, fmx_kernels.h): allocated and armed once -- every entry "gather", the flags of the entries
// that are the last occurrence of their feature in the slot -- and from then on kept current by the one-pass epochs run with
// FMX_FLAG_KEEP_WSIDE.  4 bytes per entry + 8 per row; temporary: 4 bytes per feature (at most 2^27 buckets).
int ensure_wside(fmx_handle h, Slot& s) {
  if (s.wside) return FMX_OK;
  if (s.nnz == 0 || s.nnz >= (1ull << 32) - 1) return FMX_OK;        // (entry indices + 1 live in 32 bits; such a slot simply keeps gathering)
  struct Acc { fmx_handle h; std::chrono::steady_clock::time_point t0;
               ~Acc() { h->setup_acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } acc_{h, std::chrono::steady_clock::now()};
  const uint32_t M = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(h->n_local, 1), 1ull << 27);
  uint32_t* last = nullptr;
  hipError_t er = fmx_dev_alloc(&last, (size_t)M * 4);
  if (er == hipSuccess) er = fmx_dev_alloc(&s.wside, (size_t)s.nnz * 4);
  if (er == hipSuccess) er = fmx_dev_alloc(&s.lmask, (size_t)s.n_rows * 8);
  if (er == hipSuccess) er = hipMemsetAsync(last, 0, (size_t)M * 4, h->stream);
  if (er == hipSuccess) {
    hipLaunchKernelGGL(k_wside_last, dim3((unsigned)std::min<uint64_t>((s.nnz + 255) / 256, 8192)), dim3(256), 0, h->stream, s.ent, s.nnz, M, last);
    hipLaunchKernelGGL(k_wside_mask, dim3((unsigned)std::min<uint32_t>((s.n_rows + 3) / 4, 16384)), dim3(256), 0, h->stream, s.ent, s.row_ptr, s.n_rows, M,
                       (const uint32_t*)last, s.lmask, s.wside);
    er = hipGetLastError();
  }
  if (er == hipSuccess) er = hipStreamSynchronize(h->stream);
  if (last) fmx_dev_free(last);
  if (er != hipSuccess) {
    if (s.wside) fmx_dev_free(s.wside);
    if (s.lmask) fmx_dev_free(s.lmask);
    s.wside = nullptr; s.lmask = nullptr;
    return fail(h, FMX_E_HIP, "weight side stream of the slot: %s", hipGetErrorString(er));
  }
  s.wside_version = 0;
  if (getenv("FMX_TRACE_SETUP")) fprintf(stderr, "[fmx setup] weight side stream %8.3f ms\n", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - acc_.t0).count());
  return FMX_OK;
}

// rest[e] (= y-hat - w0) for rows [row0,row0+n) of a slot, single device
int launch_rest(fmx_handle h, const Slot& s, uint64_t row0, uint32_t n, float* rest, hipStream_t st) {
  if (!s.blocks.empty()) {
    // kept `-relation` blocks: the partial sums of the main rows, plus -- through the mappings -- those of the block rows
    // (every block row is evaluated once, however many main rows use it), then rest = c + 0.5 * sum_f S_f^2
    if (row0 != 0 || n != s.n_rows) return fail(h, FMX_E_ARG, "rows with kept blocks are predicted as a whole");
    int rc = ensure_scratch(h, n, 0);
    if (rc) return rc;
    float* S = h->partial;
    float* c = S + (size_t)n * h->KP;
    KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_rowsums<KP, true, false>), n, st, s.ent, s.row_ptr, (uint64_t)0, n, h->tb, h->cfg.k1, S, c, (const float*)nullptr));
    for (BlockRows* br : s.blocks) {
      const uint32_t B = br->rows.n_rows;
      if (!B) continue;
      Tab tb = h->tb;                                        // the block's attribute 0 is global attribute attr_offset
      tb.V += (size_t)br->attr_offset * tb.rs; tb.w += (size_t)br->attr_offset * tb.ws;
      float* Sb = br->pbuf;
      KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_rowsums<KP, true, false>), B, st, br->rows.ent, br->rows.row_ptr, (uint64_t)0, B, tb, h->cfg.k1,
                                         Sb, Sb + (size_t)B * h->KP, (const float*)nullptr));
      KP_SWITCH(h->KP, hipLaunchKernelGGL((k_rel_add_partial<KP>), dim3(std::min<uint32_t>((uint32_t)(((uint64_t)n * (h->KP + 1) + 255) / 256), 4096)),
                                            dim3(256), 0, st, br->map, n, B, (const float*)Sb, S));
    }
    KP_SWITCH(h->KP, hipLaunchKernelGGL((k_rest_from_partial<KP>), dim3(wave_grid((n + Map<KP>::EPI - 1) / Map<KP>::EPI)), dim3(256), 0, st,
                                          (const float*)S, (const float*)c, n, rest));
    HIPCHK(h, hipGetLastError());
    return FMX_OK;
  }
  // the linear weights come out of the slot's side stream where it is current (it is after a one-pass epoch on THIS slot that kept it,
  // and until anything else touches w): 4 coalesced bytes per entry instead of a 64-byte fabric request
  const float* ws = (s.wside && s.wside_version == h->w_version && h->cfg.shard_world == 1) ? s.wside : nullptr;
  KP_SWITCH(h->KP, FMX_LAUNCH_WAVES((k_rowsums<KP, false, true>), n, st,
                                     s.ent, s.row_ptr, row0, n, h->tb, h->cfg.k1, (float*)nullptr, rest, ws));
  HIPCHK(h, hipGetLastError());
  return FMX_OK;
}

// collision mass of a slot's rows (cached in the slot): one histogram pass over the entries + one reduction.  On a feature
// shard this is the shard's share (its features only); the shares add up to the rows' C (fmx_comm.hip sums them).
int ensure_coll_mass(fmx_handle h, Slot& s) {
  if (s.coll_mass >= 0.0) return FMX_OK;
  if (s.nnz == 0 || s.n_rows == 0) { s.coll_mass = 0.0; return FMX_OK; }
  struct Acc { fmx_handle h; std::chrono::steady_clock::time_point t0;
               ~Acc() { h->setup_acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } acc_{h, std::chrono::steady_clock::now()};
  HIPCHK(h, hipSetDevice(h->device));
  const uint32_t M = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(h->n_local, 1), 1ull << 27);
  double* hist = nullptr;
  HIPCHK(h, fmx_dev_alloc(&hist, (size_t)M * sizeof(double)));
  hipError_t er = hipMemsetAsync(hist, 0, (size_t)M * sizeof(double), h->stream);
  if (er == hipSuccess) er = hipMemsetAsync(h->acc, 0, 2 * sizeof(double), h->stream);
  if (er == hipSuccess) {
    hipLaunchKernelGGL(k_coll_hist, dim3((unsigned)std::min<uint64_t>((s.nnz + 255) / 256, 8192)), dim3(256), 0, h->stream, s.ent, s.nnz, M, hist, h->acc + 1);
    hipLaunchKernelGGL(k_coll_sumsq, dim3((unsigned)std::min<uint64_t>(((uint64_t)M + 255) / 256, 4096)), dim3(256), 0, h->stream,
                       (const double*)hist, M, h->acc);
    er = hipGetLastError();
  }
  double c[2] = {0.0, 0.0};
  if (er == hipSuccess) er = hipMemcpyAsync(c, h->acc, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (er == hipSuccess) er = hipStreamSynchronize(h->stream);
  fmx_dev_free(hist);
  if (er != hipSuccess) return fail(h, FMX_E_HIP, "collision mass of the rows: %s", hipGetErrorString(er));
  // pairs of DIFFERENT rows: what a row shares with itself (sum of x^2 over its entries) is taken out.  On a feature shard this is
  // the shard's share of the numerator over the same N (N - 1): the shares add up.
  const double N = (double)s.n_rows;
  s.coll_mass = (s.n_rows > 1) ? std::max(0.0, c[0] - c[1]) / (N * (N - 1.0)) : 0.0;
  if (getenv("FMX_TRACE_SETUP")) fprintf(stderr, "[fmx setup] collision mass     %8.3f ms\n", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - acc_.t0).count());
  return FMX_OK;
}

// the batch an epoch runs with (fmx_sgd_opts::batch): explicit, or the default cut to learn_rate * curvature * batch * C <= 1
void resolve_batch(const fmx_config& cfg, double coll_mass, uint32_t requested, uint32_t dflt, double curv_scale, fmx_batch_info* out) {
  const double curv = ((cfg.task == FMX_TASK_REGRESSION) ? 1.0 : 0.25) * curv_scale;
  const double per_row = cfg.learn_rate * curv * coll_mass;         // gain of one row of batch
  uint32_t B = requested ? requested : dflt;
  uint32_t status = 0;
  if (!requested && per_row > 0.0 && (double)B * per_row > 1.0) {
    uint32_t cut = 1;
    while ((double)(cut * 2) * per_row <= 1.0 && cut * 2 <= dflt) cut *= 2;
    B = std::max(cut, 32u);                                      // (rows that dense -- C in the hundreds -- are SEQUENTIAL's case: the gain says so)
    status |= FMX_STAT_BATCH_CUT;
  }
  out->collision_mass = coll_mass;
  out->batch = B;
  out->batch_gain = (double)B * per_row;
  if (out->batch_gain > 2.0) status |= FMX_STAT_UNSTABLE;
  out->status = status;
}

// ---- placement of big parameter tables: an arena of chunks from two memory classes ---------------------------------------------
// scripts/ubench/placement_chunks / _order / _classes.hip (profiles/r03_placement_*.txt): the rate of random 256-byte row traffic is a
// property of the PHYSICAL memory (the same chunks mapped elsewhere keep it) and of its SPREAD: any 1 GB piece of any table runs at
// 4.9 TB/s, and so does a table whose pieces all come from one of three classes of ~96 GB the device's memory falls into; pieces of
// two classes evenly mixed run at 6.1 (3 : 1 -> 5.8, 7 : 1 -> 5.4; three classes are no better than two).  A plain allocation lies in
// one class or straddles two by luck.  Here: chunks are taken one at a time, each is classified by probing it together with the
// reference chunk of every class seen so far (k_place_pair), until two classes can supply half of the arena each; those are mapped
// alternately into one range, everything else goes back.  Any failure of the virtual-memory API leaves nothing behind and the caller
// falls back to plain allocations.
namespace {
constexpr size_t ARENA_CHUNK = (size_t)1 << 30;
struct ArenaPool {
  void* va = nullptr; size_t cap = 0;                              // scratch range the pool's chunks are mapped into
  std::vector<hipMemGenericAllocationHandle_t> hnd;                // chunk i is mapped at va + i * ARENA_CHUNK while mapped[i]
  std::vector<char> mapped;
  std::vector<int> cls;
};
void arena_pool_release(ArenaPool& P) {
  for (size_t i = 0; i < P.hnd.size(); i++) {
    if (P.mapped[i]) (void)hipMemUnmap((char*)P.va + i * ARENA_CHUNK, ARENA_CHUNK);
    (void)hipMemRelease(P.hnd[i]);
  }
  if (P.va) (void)hipMemAddressFree(P.va, P.cap * ARENA_CHUNK);
  P.hnd.clear(); P.mapped.clear(); P.cls.clear(); P.va = nullptr;
}
}  // namespace

// One classified arena per device outlives its handle: the next fmx_create on that device that needs no more chunks takes it over (a
// prefix of an alternating sequence alternates) instead of probing again -- the bench's later legs, a learner's second handle.  The
// memory stays allocated until it is reused, fmx_release_cached_memory() is called, an arena_build runs short of memory, or the
// process ends; FMX_ARENA_CACHE=0 turns the cache off.
namespace {
struct ArenaCacheEntry { Arena a; bool full = false; };
std::mutex g_arena_mu;
ArenaCacheEntry g_arena_cache[16];
void arena_release(Arena& A, std::string* err) {
  if (!A.va) return;
  // chunk by chunk, as mapped: a failing unmap must not keep the other chunks
  for (size_t c = 0; c < A.n_chunks; c++) {
    const hipError_t e = hipMemUnmap((char*)A.va + c * A.chunk_bytes, A.chunk_bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); if (err) *err = std::string("arena_free: hipMemUnmap failed: ") + hipGetErrorString(e); }
  }
  (void)hipMemAddressFree(A.va, A.reserved_bytes ? A.reserved_bytes : A.bytes);
  A = Arena();
}
bool arena_cache_on() { const char* e = getenv("FMX_ARENA_CACHE"); return !(e && e[0] == '0'); }
}  // namespace

void arena_cache_drop(int device) {
  std::lock_guard<std::mutex> lk(g_arena_mu);
  for (int d = 0; d < 16; d++)
    if ((device < 0 || d == device) && g_arena_cache[d].full) {
      int cur = 0; (void)hipGetDevice(&cur); (void)hipSetDevice(d);
      arena_release(g_arena_cache[d].a, nullptr);
      g_arena_cache[d].full = false;
      (void)hipSetDevice(cur);
    }
}

void arena_free(fmx_handle h) {
  Arena& A = h->arena;
  if (!A.va) return;
  if (A.method == 2 && h->device >= 0 && h->device < 16 && arena_cache_on()) {
    (void)hipStreamSynchronize(h->stream);
    std::lock_guard<std::mutex> lk(g_arena_mu);
    ArenaCacheEntry& c = g_arena_cache[h->device];
    if (!c.full || c.a.n_chunks < A.n_chunks) {                    // keep the larger one
      if (c.full) arena_release(c.a, nullptr);
      c.a = A; c.full = true;
      A = Arena();
      return;
    }
  }
  arena_release(A, &h->err);
}

// V (v_bytes) at the start of the arena, w (w_bytes) centred on a chunk boundary behind it (both classes under it too)
static hipError_t arena_build(fmx_handle h, size_t v_bytes, size_t w_bytes, int bound_tables, float** V_out, float** w_out) {
  const auto t_start = std::chrono::steady_clock::now();
  const size_t CH = ARENA_CHUNK;
  uint64_t ch_ = 0, w_off_ = 0; uint32_t T = 0;
  if (fmx_place_layout(v_bytes, w_bytes, &ch_, &T, &w_off_) != FMX_OK) return hipErrorInvalidValue;
  const size_t w_off = (size_t)w_off_;
  hipError_t er = hipSuccess;
  if (h->device >= 0 && h->device < 16) {                            // an arena this device's previous handle left behind?
    std::lock_guard<std::mutex> lk(g_arena_mu);
    ArenaCacheEntry& c = g_arena_cache[h->device];
    // a PREFIX of the cached chunk order serves a smaller model only if it still spreads both tables over both classes: the order is a
    // Bresenham spread (X X Y X Y ... for odd T or a short second class), so the prefix is recounted, never assumed to alternate
    auto prefix_ok = [&](const Arena& A) -> bool {
      if (A.cls.size() < T) return false;
      auto both = [&](size_t c0, size_t c1) {                        // chunks [c0, c1]: one chunk cannot span two classes; more must
        if (c1 <= c0) return true;
        bool s0 = false, s1 = false;
        for (size_t k = c0; k <= c1 && k < T; k++) { s0 |= A.cls[k] == 0; s1 |= A.cls[k] == 1; }
        return s0 && s1;
      };
      const size_t v_last = v_bytes ? (v_bytes - 1) / CH : 0;
      const size_t w_first = w_off / CH, w_last = w_bytes ? (w_off + w_bytes - 1) / CH : w_first;
      return both(0, v_last) && both(w_first, w_last);
    };
    if (c.full && c.a.chunk_bytes == CH && c.a.n_chunks >= T && !prefix_ok(c.a)) { arena_release(c.a, nullptr); c.full = false; }   // probe again
    if (c.full && c.a.chunk_bytes == CH && c.a.n_chunks >= T) {
      Arena A = c.a;
      c.a = Arena(); c.full = false;
      for (size_t k = T; k < A.n_chunks; k++) (void)hipMemUnmap((char*)A.va + k * CH, CH);   // the tail goes back to the device
      A.n_chunks = T; A.bytes = (size_t)T * CH;
      A.cls.resize(T);
      A.per_class[0] = A.per_class[1] = 0;
      for (uint8_t cc : A.cls) if (cc < 2) A.per_class[cc]++;       // (recounted from the prefix actually taken)
      A.pool = 0;                                                    // nothing probed this time
      er = hipMemsetAsync(A.va, 0, A.bytes, h->stream);
      if (er == hipSuccess) {
        h->arena = A;
        *V_out = (float*)A.va;
        *w_out = (float*)((char*)A.va + w_off);
        h->arena.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        return hipSuccess;
      }
      (void)hipGetLastError();
      arena_release(A, nullptr);
    }
  }
  size_t free_b = 0, total_b = 0;
  er = hipMemGetInfo(&free_b, &total_b);
  if (er != hipSuccess) return er;
  if ((size_t)T * CH + ((size_t)2 << 30) > free_b) {                  // short of memory: whatever the cache holds goes back first
    arena_cache_drop(h->device);
    er = hipMemGetInfo(&free_b, &total_b);
    if (er != hipSuccess) return er;
  }
  if ((size_t)T * CH + ((size_t)2 << 30) > free_b) return hipErrorOutOfMemory;    // (the caller's plain allocation reports it properly)
  // the pool: at most bound_tables arenas' worth + 64 chunks (a class comes in runs of up to 64 GB in allocation order), at most
  // half of what is free beyond the arena itself
  size_t max_pool = std::min<size_t>((size_t)bound_tables * T + 64, T + (free_b - (size_t)T * CH) / CH / 2);
  max_pool = std::max<size_t>(max_pool, T);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = h->device;
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = h->device; acc.flags = hipMemAccessFlagsProtReadWrite;
  ArenaPool P;
  P.cap = max_pool;
  er = hipMemAddressReserve(&P.va, P.cap * CH, CH, nullptr, 0);
  if (er != hipSuccess) { P.va = nullptr; return er; }
  auto chunk_va = [&](size_t i) { return (float*)((char*)P.va + i * CH); };
  const uint32_t rows_shift = 22;                                   // 1 GiB / 256 B
  const uint32_t waves = 1u << 17;                                  // 2.1 GB of traffic per probe, ~0.4 ms
  auto pair_ms = [&](size_t i, size_t j, float* ms) -> hipError_t {
    float best = 1e30f;
    for (int rep = 0; rep < 2; rep++) {                             // (first launch: page-table warm-up; the second one is the measurement)
      hipError_t e = hipEventRecord(h->ev0, h->stream);
      hipLaunchKernelGGL(k_place_pair, dim3(waves / 4), dim3(256), 0, h->stream, chunk_va(i), chunk_va(j), rows_shift, waves, (uint64_t)rep * 7919 + 1);
      if (e == hipSuccess) e = hipGetLastError();
      if (e == hipSuccess) e = hipEventRecord(h->ev1, h->stream);
      if (e == hipSuccess) e = hipEventSynchronize(h->ev1);
      float t = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&t, h->ev0, h->ev1);
      if (e != hipSuccess) return e;
      if (rep && t < best) best = t;
    }
    *ms = best;
    return hipSuccess;
  };
  std::vector<size_t> refs;                                          // first chunk of every class
  std::vector<std::vector<size_t>> of;                               // chunks per class
  std::vector<size_t> mixed;                                         // chunks that straddle a border between classes: fillers only
  std::vector<float> alone;                                          // a chunk's time under the probe on its own
  float slow_ms = 0.f;                                               // ... of a chunk that lies in ONE class (the median of the first eight)
  auto enough = [&](size_t* x, size_t* y) -> bool {                  // two classes with ceil(T/2) and floor(T/2) chunks?
    size_t a = SIZE_MAX, b = SIZE_MAX;
    for (size_t k = 0; k < of.size(); k++) {
      if (a == SIZE_MAX || of[k].size() > of[a].size()) { b = a; a = k; }
      else if (b == SIZE_MAX || of[k].size() > of[b].size()) b = k;
    }
    *x = a; *y = b;
    return a != SIZE_MAX && b != SIZE_MAX && of[a].size() >= (T + 1) / 2 && of[b].size() >= T / 2;
  };
  auto take = [&]() -> hipError_t {                                  // one more chunk: created, mapped into the pool's range, written, timed alone
    const size_t i = P.hnd.size();
    hipMemGenericAllocationHandle_t hd;
    hipError_t e = hipMemCreate(&hd, CH, &prop, 0);
    if (e != hipSuccess) return e;
    P.hnd.push_back(hd); P.mapped.push_back(0); P.cls.push_back(-1); alone.push_back(0.f);
    e = hipMemMap((char*)P.va + i * CH, CH, 0, hd, 0);
    if (e != hipSuccess) return e;
    P.mapped[i] = 1;
    e = hipMemSetAccess((char*)P.va + i * CH, CH, &acc, 1);
    if (e == hipSuccess) e = hipMemsetAsync(chunk_va(i), 0, CH, h->stream);   // (a never-written allocation answers a probe in microseconds)
    // timed on its own only while the one-class rate is being established (the first eight): later chunks go straight to the pair probe --
    // a chunk that straddles a class border then matches no reference and opens a class of its own, which never gets stocked
    if (e == hipSuccess && i < 8) e = pair_ms(i, i, &alone[i]);
    return e;
  };
  // same class <=> the two chunks together are no faster than one alone; a chunk that is faster ALONE straddles a border (it must not
  // become a reference: everything would look like its class)
  int last_cls = -1;                                                 // classes come in runs of 32 / 64 chunks in allocation order: try the
  auto classify = [&](size_t i) -> hipError_t {                      // previous chunk's class first (one probe per chunk inside a run)
    if (alone[i] > 0.f && alone[i] < 0.95f * slow_ms) { P.cls[i] = -2; mixed.push_back(i); return hipSuccess; }
    for (size_t t = 0; t < refs.size() && P.cls[i] < 0; t++) {
      const size_t k = (last_cls >= 0) ? (t == 0 ? (size_t)last_cls : (t <= (size_t)last_cls ? t - 1 : t)) : t;
      float ms = 0.f;
      const hipError_t e = pair_ms(refs[k], i, &ms);
      if (e != hipSuccess) return e;
      if (ms > slow_ms / 1.08f) { P.cls[i] = (int)k; of[k].push_back(i); last_cls = (int)k; }
    }
    if (P.cls[i] < 0) { refs.push_back(i); of.push_back({i}); P.cls[i] = (int)refs.size() - 1; last_cls = P.cls[i]; }
    return hipSuccess;
  };
  size_t cx = SIZE_MAX, cy = SIZE_MAX;
  bool out_of_memory = false;
  const size_t first = std::min<size_t>(8, max_pool);
  while (er == hipSuccess && P.hnd.size() < first) er = take();
  if (er == hipSuccess) {
    std::vector<float> srt(alone);
    std::sort(srt.begin(), srt.end());
    slow_ms = srt[srt.size() / 2];
    for (size_t i = 0; i < first && er == hipSuccess; i++) er = classify(i);
  }
  while (er == hipSuccess && P.hnd.size() < max_pool && !enough(&cx, &cy)) {
    er = take();
    if (er != hipSuccess) {                                           // memory ran out: go with the pool in hand
      if (P.hnd.size() > T) { (void)hipGetLastError(); er = hipSuccess; out_of_memory = true;
        if (!P.mapped.back()) { (void)hipMemRelease(P.hnd.back()); P.hnd.pop_back(); P.mapped.pop_back(); P.cls.pop_back(); alone.pop_back(); } }
      break;
    }
    er = classify(P.hnd.size() - 1);
  }
  (void)out_of_memory;
  if (er != hipSuccess || P.hnd.size() < T) { arena_pool_release(P); return er != hipSuccess ? er : hipErrorOutOfMemory; }
  (void)enough(&cx, &cy);
  // the arena's chunks in mapping order: the two best-stocked classes alternately; when the second runs short (pool bound reached)
  // its chunks are spread evenly among the first's, then any other class fills up
  std::vector<size_t> pick;
  std::vector<uint8_t> pick_cls;
  {
    std::vector<size_t> X = (cx != SIZE_MAX) ? of[cx] : std::vector<size_t>(), Y = (cy != SIZE_MAX) ? of[cy] : std::vector<size_t>();
    std::vector<size_t> rest;
    for (size_t k = 0; k < of.size(); k++) if (k != cx && k != cy) rest.insert(rest.end(), of[k].begin(), of[k].end());
    rest.insert(rest.end(), mixed.begin(), mixed.end());
    const size_t ny = std::min<size_t>(Y.size(), T / 2);
    size_t nx = std::min<size_t>(X.size(), T - ny);
    size_t ix = 0, iy = 0, ir = 0, err_acc = 0;
    for (uint32_t c = 0; c < T; c++) {
      err_acc += ny;
      if (iy < ny && err_acc >= T) { err_acc -= T; pick.push_back(Y[iy++]); pick_cls.push_back(1); }
      else if (ix < nx) { pick.push_back(X[ix++]); pick_cls.push_back(0); }
      else if (iy < Y.size()) { pick.push_back(Y[iy++]); pick_cls.push_back(1); }
      else if (ir < rest.size()) { pick.push_back(rest[ir++]); pick_cls.push_back(2); }
    }
    if (pick.size() < T) { arena_pool_release(P); return hipErrorOutOfMemory; }
    h->arena.per_class[0] = (uint32_t)ix; h->arena.per_class[1] = (uint32_t)iy;
  }
  void* va = nullptr;
  er = hipMemAddressReserve(&va, (size_t)T * CH, CH, nullptr, 0);
  if (er != hipSuccess) { arena_pool_release(P); return er; }
  er = hipStreamSynchronize(h->stream);
  uint32_t moved = 0;                                                // chunks mapped into the arena's range so far (counted AFTER a successful map)
  while (moved < T && er == hipSuccess) {
    const size_t i = pick[moved];
    er = hipMemUnmap((char*)P.va + i * CH, CH);
    if (er != hipSuccess) break;
    P.mapped[i] = 0;
    er = hipMemMap((char*)va + (size_t)moved * CH, CH, 0, P.hnd[i], 0);
    if (er == hipSuccess) moved++;
  }
  if (er == hipSuccess) er = hipMemSetAccess(va, (size_t)T * CH, &acc, 1);
  if (er != hipSuccess) {
    for (uint32_t c = 0; c < moved; c++) (void)hipMemUnmap((char*)va + (size_t)c * CH, CH);   // chunk by chunk: exactly what was mapped
    (void)hipMemAddressFree(va, (size_t)T * CH);
    arena_pool_release(P);
    return er;
  }
  const uint32_t pool = (uint32_t)P.hnd.size();
  arena_pool_release(P);                                             // the unused chunks go back; the mapped ones live on through their mapping
  Arena& A = h->arena;
  A.va = va; A.bytes = (size_t)T * CH; A.reserved_bytes = A.bytes; A.chunk_bytes = CH; A.n_chunks = T; A.pool = pool; A.classes_seen = (uint32_t)refs.size(); A.method = 2;
  A.cls = pick_cls;
  er = hipMemsetAsync(va, 0, A.bytes, h->stream);
  if (er != hipSuccess) { arena_free(h); return er; }
  *V_out = (float*)va;
  *w_out = (float*)((char*)va + w_off);
  A.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  return hipSuccess;
}

extern "C" {

int fmx_abi_version(void) { return FMX_ABI_VERSION; }


// ---- device allocations (fmx_internal.h: fmx_dev_alloc / fmx_dev_free) -----------------------------------------------------------
namespace {
std::atomic<uint64_t> g_devices_used{0};                              // devices fmx_create has opened a handle on in this process
struct BigAlloc { size_t reserved = 0; int device = 0; std::vector<hipMemGenericAllocationHandle_t> hnd; std::vector<size_t> sz; };
std::mutex g_big_mu;
std::unordered_map<void*, BigAlloc> g_big;
constexpr size_t BIG_MIN = (size_t)768 << 20, BIG_CHUNK = (size_t)1 << 30, BIG_ALIGN = (size_t)2 << 20;
bool big_alloc_on() { static const bool on = []() { const char* e = getenv("FMX_BIG_ALLOC"); return !(e && e[0] == '0'); }(); return on; }

hipError_t plain_alloc(void** p, size_t bytes) {          // hipMalloc; out of memory: the idle arena cache of the device goes back, once
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    int d = 0;
    if (hipGetDevice(&d) == hipSuccess) { arena_cache_drop(d); e = hipMalloc(p, bytes); }
  }
  return e;
}
void big_release(void* va, BigAlloc& b) {
  size_t off = 0;
  for (size_t i = 0; i < b.hnd.size(); i++) { (void)hipMemUnmap((char*)va + off, b.sz[i]); (void)hipMemRelease(b.hnd[i]); off += b.sz[i]; }
  if (va) (void)hipMemAddressFree(va, b.reserved);
}
// one virtual range backed by physical chunks of at most 1 GiB; false: nothing is left behind and the caller takes hipMalloc
bool big_alloc(void** p, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  BigAlloc b; b.device = dev;
  b.reserved = (bytes + BIG_ALIGN - 1) / BIG_ALIGN * BIG_ALIGN;
  void* va = nullptr;
  static const bool trace = getenv("FMX_TRACE_ALLOC") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&]() { return 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  if (hipMemAddressReserve(&va, b.reserved, BIG_ALIGN, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
  const double t_res = ms();
  bool ok = true;
  for (size_t off = 0; off < b.reserved && ok; off += BIG_CHUNK) {
    const size_t sz = std::min(BIG_CHUNK, b.reserved - off);
    hipMemGenericAllocationHandle_t hd;
    if (hipMemCreate(&hd, sz, &prop, 0) != hipSuccess) { ok = false; break; }
    if (hipMemMap((char*)va + off, sz, 0, hd, 0) != hipSuccess) { (void)hipMemRelease(hd); ok = false; break; }
    b.hnd.push_back(hd); b.sz.push_back(sz);
  }
  const double t_map = ms();
  if (ok) {
    // read / write for the owning device and -- where the runtime grants it -- for the OTHER DEVICES THIS PROCESS HOLDS HANDLES ON
    // (fmx_group_upload_rows copies staged rows between the devices of a one-process group); a peer the runtime refuses is not an error
    // here.  A process that drives one GPU (one process per GPU: the multi-GPU bench) never touches another device.
    int ndev = 0; (void)hipGetDeviceCount(&ndev);
    hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
    ok = hipMemSetAccess(va, b.reserved, &acc, 1) == hipSuccess;
    const uint64_t used = g_devices_used.load(std::memory_order_relaxed);
    for (int d = 0; ok && d < ndev && d < 64; d++) {
      if (d == dev || !((used >> d) & 1ull)) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, d, dev) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
      acc.location.id = d;
      if (hipMemSetAccess(va, b.reserved, &acc, 1) != hipSuccess) (void)hipGetLastError();
    }
  }
  if (trace) fprintf(stderr, "[fmx alloc] big %zu bytes: reserve %.3f ms, create+map %.3f ms, access %.3f ms\n", bytes, t_res, t_map - t_res, ms() - t_map);
  if (!ok) { (void)hipGetLastError(); big_release(va, b); return false; }
  { std::lock_guard<std::mutex> lk(g_big_mu); g_big.emplace(va, std::move(b)); }
  *p = va;
  return true;
}
}  // namespace

static hipError_t dev_alloc_untraced(void** p, size_t bytes);
extern "C++" hipError_t fmx_dev_alloc_bytes(void** p, size_t bytes) {
  static const bool trace = getenv("FMX_TRACE_ALLOC") != nullptr;
  if (!trace) return dev_alloc_untraced(p, bytes);
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = dev_alloc_untraced(p, bytes);
  const double ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (ms > 0.5) fprintf(stderr, "[fmx alloc] %12zu bytes %9.3f ms%s\n", bytes, ms, e == hipSuccess ? "" : " FAILED");
  return e;
}
static hipError_t dev_alloc_untraced(void** p, size_t bytes) {
  if (!p) return hipErrorInvalidValue;
  *p = nullptr;
  if (bytes >= BIG_MIN && big_alloc_on()) {
    if (big_alloc(p, bytes)) return hipSuccess;
    int d = 0;                                               // (out of memory next to an idle cached arena? give it back and try once more)
    if (hipGetDevice(&d) == hipSuccess) { arena_cache_drop(d); if (big_alloc(p, bytes)) return hipSuccess; }
  }
  return plain_alloc(p, bytes);
}
extern "C++" hipError_t fmx_dev_free(void* p) {
  if (!p) return hipSuccess;
  BigAlloc b; bool big = false;
  {
    std::lock_guard<std::mutex> lk(g_big_mu);
    auto it = g_big.find(p);
    if (it != g_big.end()) { b = std::move(it->second); g_big.erase(it); big = true; }
  }
  if (!big) return hipFree(p);
  int cur = 0; (void)hipGetDevice(&cur);
  if (cur != b.device) (void)hipSetDevice(b.device);
  (void)hipDeviceSynchronize();                              // (hipFree waits for the device's work too: nothing may still read the range)
  big_release(p, b);
  if (cur != b.device) (void)hipSetDevice(cur);
  return hipSuccess;
}

int fmx_release_cached_memory(void) { arena_cache_drop(-1); return FMX_OK; }

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
  if (cfg->num_factor > 1024) return fail(nullptr, FMX_E_UNSUPPORTED, "num_factor > 1024 is not supported");
  if (cfg->task != FMX_TASK_REGRESSION && cfg->task != FMX_TASK_CLASSIFICATION)
    return fail(nullptr, FMX_E_ARG, "unknown task");                       // fm_learn.h:81 "unknown task"
  if (cfg->shard_world < 1 || cfg->shard_rank < 0 || cfg->shard_rank >= cfg->shard_world)
    return fail(nullptr, FMX_E_ARG, "bad shard_rank/shard_world");
  if (cfg->shard_hash != 0 && cfg->shard_hash != 1) return fail(nullptr, FMX_E_ARG, "shard_hash must be 0 or 1");
  if (cfg->place_candidates < 0 || cfg->place_candidates > 6) return fail(nullptr, FMX_E_ARG, "place_candidates must be 0 (default) .. 6");
  if (cfg->exchange_algo != FMX_EXCHANGE_ALLREDUCE && cfg->exchange_algo != FMX_EXCHANGE_RS_AG)
    return fail(nullptr, FMX_E_ARG, "exchange_algo %u: FMX_EXCHANGE_ALLREDUCE (0) or FMX_EXCHANGE_RS_AG (1)", cfg->exchange_algo);
  if ((uint64_t)cfg->shard_world > cfg->num_attribute) return fail(nullptr, FMX_E_ARG, "more feature shards than features");
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
  if (dev >= 0 && dev < 64) g_devices_used.fetch_or(1ull << dev, std::memory_order_relaxed);
  CREATE_CHK(hipGetDeviceProperties(&h->prop, dev));
  CREATE_CHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  CREATE_CHK(hipEventCreate(&h->ev0));
  CREATE_CHK(hipEventCreate(&h->ev1));
  {
    // row layout: V rows of KP floats, the linear weights in an array of their own (co-locating w_j behind its row was measured
    // in rounds 1 and 2: HBM fetches 64-byte sectors, a 4-byte w costs one wherever it lives, and unaligned rows cost more)
    // ... rows of num_factor floats rounded up to 16 (one 64-byte sector), not to the power of two KP of the lane mapping: k = 100 keeps rows of
    // 112 floats (448 B) where the padded 128 moved 512 (round-5 verdict item 9; fmx_kernels.h row_ld).  FMX_ROW_STRIDE_KP=1: the padded rows.
    h->tb.rs = (uint32_t)h->KP;
    { const char* rk = getenv("FMX_ROW_STRIDE_KP");
      if (h->KP >= 32 && !(rk && rk[0] == '1')) h->tb.rs = std::min<uint32_t>((uint32_t)h->KP, ((uint32_t)cfg->num_factor + 15u) / 16u * 16u); }
    // Placement of the parameter tables: big ones (>= 2 GiB) in an arena of chunks from two memory classes (arena_build above).  Smaller
    // ones, and every table when the virtual-memory API fails: a table of >= 256 MB is allocated up to fmx_config::place_candidates
    // times (default 2; the earlier candidate is held meanwhile, so that the later one is other memory), each candidate is zeroed and
    // timed under a probe with the step's traffic shape, the fastest is kept (a plain allocation straddles two classes or not, by
    // luck).  place_candidates = 1: first fit, no probe.
    auto alloc_placed = [&](float** out, size_t bytes, int max_tries, bool rows, const char* what) -> hipError_t {
      constexpr int MAXC = 6;
      int tries = 1;
      if (bytes >= ((size_t)256 << 20) && max_tries > 1) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
          tries = (int)std::min<size_t>((size_t)std::min(max_tries, MAXC), std::max<size_t>(1, free_b / (bytes + ((size_t)4 << 30))));   // candidates sit side by side
      }
      float* cand[MAXC] = {};
      float ms_of[MAXC] = {};
      int n_cand = 0, best = 0;
      hipError_t er = hipSuccess;
      for (int c = 0; c < tries && er == hipSuccess; c++) {
        if (plain_alloc((void**)&cand[c], bytes) != hipSuccess) { (void)hipGetLastError(); cand[c] = nullptr; break; }
        n_cand = c + 1;
        er = hipMemsetAsync(cand[c], 0, bytes, h->stream);   // (also: a never-written allocation answers the probe in 30 us)
        if (tries == 1 || er != hipSuccess) break;
        for (int rep = 0; rep < 2 && er == hipSuccess; rep++) {           // (first launch: page-table warm-up)
          er = hipEventRecord(h->ev0, h->stream);
          if (rows) {                                         // 2^19 wavefronts x 32 rows: 8 GB of traffic at k = 64, ~1.5 ms
            const uint32_t waves = 1u << 19;
            hipLaunchKernelGGL(k_place_probe, dim3(waves / 4), dim3(256), 0, h->stream, cand[c], (uint64_t)h->n_local, h->tb.rs, waves, (uint64_t)rep * 7919 + 1);
          } else {                                            // 2^24 random 4-byte read-modify-writes, ~0.5 ms
            const uint64_t total = 1ull << 24;
            hipLaunchKernelGGL(k_place_probe_w, dim3((unsigned)(total / 256)), dim3(256), 0, h->stream, cand[c], (uint64_t)(bytes / sizeof(float)), total, (uint64_t)rep * 7919 + 1);
          }
          if (er == hipSuccess) er = hipGetLastError();
          if (er == hipSuccess) er = hipEventRecord(h->ev1, h->stream);
          if (er == hipSuccess) er = hipEventSynchronize(h->ev1);
          if (er == hipSuccess) er = hipEventElapsedTime(&ms_of[c], h->ev0, h->ev1);
        }
        if (ms_of[c] < ms_of[best]) best = c;
        float worst = 0.f;
        for (int i = 0; i <= c; i++) worst = std::max(worst, ms_of[i]);
        if (c >= 1 && ms_of[best] < 0.95f * worst) break;     // two classes seen: the fast one is in hand
      }
      if (er == hipSuccess && !cand[0]) {                                    // out of memory: whatever the arena cache holds goes back first
        arena_cache_drop(h->device);
        er = plain_alloc((void**)&cand[0], bytes);                                   // (reports the allocation failure)
        if (er == hipSuccess) er = hipMemsetAsync(cand[0], 0, bytes, h->stream);
      }
      (void)what;
      for (int c = 0; c < MAXC; c++) if ((c != best || er != hipSuccess) && cand[c]) { fmx_dev_free(cand[c]); cand[c] = nullptr; }
      *out = cand[best];
      return er;
    };
    const auto t_place = std::chrono::steady_clock::now();
    const size_t v_bytes = h->n_local * (size_t)h->tb.rs * sizeof(float), w_bytes = h->n_local * sizeof(float);
    bool placed = false;
    if (cfg->place_candidates != 1 && v_bytes >= ((size_t)2 << 30)) {
      // big tables: an arena of 1 GiB chunks from two memory classes (arena_build above); on any failure plain allocations below
      const hipError_t ae = arena_build(h, v_bytes, w_bytes, cfg->place_candidates > 1 ? cfg->place_candidates : 3, &h->tb.V, &h->w_sep);
      if (ae == hipSuccess) placed = true; else { (void)hipGetLastError(); h->tb.V = nullptr; h->w_sep = nullptr; }
    }
    if (!placed) {
      const int cand = cfg->place_candidates > 0 ? std::min(cfg->place_candidates, 6) : 2;
      CREATE_CHK(alloc_placed(&h->tb.V, v_bytes, cand, true, "factor table"));
      CREATE_CHK(alloc_placed(&h->w_sep, w_bytes, cand, false, "linear weights"));
      h->arena.method = (cand > 1 && v_bytes >= ((size_t)256 << 20)) ? 1 : 0;
      h->arena.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_place).count();
    }
    h->tb.w = h->w_sep; h->tb.ws = 1;
  }
  CREATE_CHK(fmx_dev_alloc(&h->w0, sizeof(double)));
  CREATE_CHK(fmx_dev_alloc(&h->w0_pp, 8 * sizeof(double)));
  {  // device-side hand-off of the bias (fmx_sgd.hip: sgd_epoch_fused): a counter + an error word, both start at 0
    CREATE_CHK(fmx_dev_alloc(&h->handoff_ctr, 2 * sizeof(unsigned long long)));
    h->handoff_err = reinterpret_cast<uint32_t*>(h->handoff_ctr + 1);
    CREATE_CHK(hipMemsetAsync(h->handoff_ctr, 0, 2 * sizeof(unsigned long long), h->stream));
    const char* e = getenv("FMX_HANDOFF");
    h->handoff = !(e && e[0] == '0');
    CREATE_CHK(fmx_dev_alloc(&h->pit_ctr, (PIT_MAX_IT + 1) * sizeof(unsigned long long)));
    CREATE_CHK(fmx_dev_alloc(&h->pit_slots, (size_t)2 * PIT_MAX_WG * 4 * sizeof(double)));
    CREATE_CHK(hipMemsetAsync(h->pit_slots, 0, (size_t)2 * PIT_MAX_WG * 4 * sizeof(double), h->stream));
    const char* sc = getenv("FMX_SCAN");
    h->scan_pit = !(sc && strcmp(sc, "serial") == 0);
    const char* xe = getenv("FMX_XCD");
    h->xcd = (xe && xe[0] == '1');                              // opt-in: measured slower than the two launches per batch (profiles/r06_criteo_hops.txt)
    const char* so = getenv("FMX_SMALL_ONE");
    h->small_one = !(so && so[0] == '0');                       // small batches as one launch per batch (fmx_small_kernels.h); FMX_SMALL_ONE=0: two
    const char* xb = getenv("FMX_XCD_MAX_BATCH");
    if (xb) h->xcd_max_batch = (uint32_t)strtoul(xb, nullptr, 10);
    const char* sp = getenv("FMX_DEBUG_PIT_SPINS");
    h->pit_spins = sp ? (uint32_t)strtoul(sp, nullptr, 10) : HANDOFF_SPINS;
  }
  {  // the side stream runs the one-workgroup bias recurrence next to chip-filling gathers: give it priority so
     // that its workgroup is placed as soon as any CU has room
    int lo = 0, hi = 0;
    CREATE_CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CREATE_CHK(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, hi));
  }
  h->num_cu = h->prop.multiProcessorCount > 0 ? h->prop.multiProcessorCount : 256;
  CREATE_CHK(fmx_dev_alloc(&h->acc, 4 * sizeof(double)));
  CREATE_CHK(hipMemsetAsync(h->w0, 0, sizeof(double), h->stream));
  CREATE_CHK(hipStreamSynchronize(h->stream));
#undef CREATE_CHK
  *out = h;
  return FMX_OK;
}

// the arena of a model: V from offset 0, w centred on the first chunk boundary behind it (half of it in either chunk: chunks alternate
// between two memory classes, so w lies in both like V does), whole chunks.  Host arithmetic, no device needed.
int fmx_place_layout(uint64_t v_bytes, uint64_t w_bytes, uint64_t* chunk_bytes, uint32_t* chunks, uint64_t* w_offset) {
  if (!chunk_bytes || !chunks || !w_offset || v_bytes == 0) return FMX_E_ARG;
  const uint64_t CH = ARENA_CHUNK;
  const uint64_t w_half = ((w_bytes / 2) + 255) & ~(uint64_t)255;                // (256-byte granules: w stays aligned like V's rows)
  const uint64_t boundary = (v_bytes + w_half + CH - 1) / CH;                     // w straddles the start of chunk `boundary`
  const uint64_t w_off = boundary * CH - w_half;
  const uint64_t T = (w_off + w_bytes + CH - 1) / CH;
  if (T > 0xFFFFFFFFull) return FMX_E_ARG;
  *chunk_bytes = CH; *chunks = (uint32_t)T; *w_offset = w_off;
  return FMX_OK;
}

int fmx_batch_rule(int32_t task, double learn_rate, double collision_mass, uint32_t requested, fmx_batch_info* out) {
  if (!out || (task != FMX_TASK_REGRESSION && task != FMX_TASK_CLASSIFICATION) || !(learn_rate >= 0.0) || !(collision_mass >= 0.0)) return FMX_E_ARG;
  fmx_config c = {};
  c.task = task; c.learn_rate = learn_rate;
  resolve_batch(c, collision_mass, requested, FMX_DEFAULT_BATCH, 1.0, out);
  return FMX_OK;
}

int fmx_get_place_info(fmx_handle h, fmx_place_info* out) {
  if (!h || !out) return FMX_E_ARG;
  const Arena& A = h->arena;
  out->method = A.method; out->chunks = A.n_chunks; out->per_class[0] = A.per_class[0]; out->per_class[1] = A.per_class[1];
  out->pool = A.pool; out->classes_seen = A.classes_seen; out->seconds = A.seconds;
  return FMX_OK;
}

int fmx_destroy(fmx_handle h) {
  if (!h) return FMX_OK;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  comm_free(h);
  als_free(h);
  sgda_free(h);
  for (auto& s : h->slots) free_slot(s);
  if (h->grp) fmx_dev_free(h->grp);
  if (h->arena.va) arena_free(h);                              // (both tables are parts of it)
  else { if (h->tb.V) fmx_dev_free(h->tb.V); if (h->w_sep) fmx_dev_free(h->w_sep); }
  if (h->w0) fmx_dev_free(h->w0);
  if (h->w0_pp) fmx_dev_free(h->w0_pp);
  if (h->handoff_ctr) fmx_dev_free(h->handoff_ctr);
  if (h->pit_ctr) fmx_dev_free(h->pit_ctr);
  if (h->xcd_sync) fmx_dev_free(h->xcd_sync);
  if (h->xcd_trace) fmx_dev_free(h->xcd_trace);
  if (h->probe_flags) fmx_dev_free(h->probe_flags);
  if (h->pit_tmp) fmx_dev_free(h->pit_tmp);
  if (h->run_slots) fmx_dev_free(h->run_slots);
  if (h->small_slots) fmx_dev_free(h->small_slots);
  if (h->pit_slots) fmx_dev_free(h->pit_slots);
  if (h->w0_slots) fmx_dev_free(h->w0_slots);
  if (h->stream2) hipStreamDestroy(h->stream2);
  if (h->acc) fmx_dev_free(h->acc);
  if (h->partial) fmx_dev_free(h->partial);
  if (h->mult) fmx_dev_free(h->mult);
  if (h->rest) fmx_dev_free(h->rest);
  for (auto ev : h->ev_pool) hipEventDestroy(ev);
  for (auto ev : h->ev_sync) hipEventDestroy(ev);
  if (h->lag.ev_rest) hipEventDestroy(h->lag.ev_rest);
  for (auto e : h->lag.ev_scan) if (e) hipEventDestroy(e);
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
  return scan_error_check(h);
}

// ---------------------------------------------------------------------------------------------
// parameters
// ---------------------------------------------------------------------------------------------
static int stage_params(fmx_handle h, bool to_device, double* w0, double* w, double* v) {
  if (to_device) touch_w(h);
  HIPCHK(h, hipSetDevice(h->device));
  const uint64_t n = h->cfg.num_attribute;
  const int k = h->cfg.num_factor, KP = h->KP;
  const int W = h->cfg.shard_world;
  const Shard sh = make_shard(h->cfg);
  if (to_device) {
    HIPCHK(h, hipMemcpyAsync(h->w0, w0, sizeof(double), hipMemcpyHostToDevice, h->stream));
  } else {
    HIPCHK(h, hipMemcpyAsync(w0, h->w0, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  const uint32_t chunk = (uint32_t)std::min<uint64_t>(n, 1u << 18);
  double* stage = nullptr;
  HIPCHK(h, fmx_dev_alloc(&stage, (size_t)chunk * (size_t)std::max(k, 1) * sizeof(double)));
  int rc = FMX_OK;
#define STAGE_CHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    rc = fail(h, FMX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); goto done; } } while (0)
  for (uint64_t j0 = 0; j0 < n; j0 += chunk) {
    const uint32_t cnt = (uint32_t)std::min<uint64_t>(chunk, n - j0);
    if (w) {
      if (to_device) {
        STAGE_CHK(hipMemcpyAsync(stage, w + j0, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_w_in, dim3((cnt + 255) / 256), dim3(256), 0, h->stream, stage, j0, cnt, sh, h->tb);
      } else {
        if (W > 1) STAGE_CHK(hipMemcpyAsync(stage, w + j0, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_w_out, dim3((cnt + 255) / 256), dim3(256), 0, h->stream, stage, j0, cnt, sh, h->tb);
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
                           stage, j0, cnt, k, KP, sh, h->tb);
      } else {
        if (W > 1)
          for (int f = 0; f < k; f++)
            STAGE_CHK(hipMemcpyAsync(stage + (size_t)f * cnt, v + (size_t)f * n + j0, cnt * sizeof(double),
                                     hipMemcpyHostToDevice, h->stream));
        const uint64_t total = (uint64_t)cnt * k;
        hipLaunchKernelGGL(k_stage_out, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, h->stream,
                           stage, j0, cnt, k, KP, sh, h->tb);
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
  fmx_dev_free(stage);
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
  const Shard sh = make_shard(h->cfg);
  if (k > 0 && !v_out) return fail(h, FMX_E_ARG, "fmx_get_param_rows: v_out is NULL");
  if (count == 0) return FMX_OK;
  for (uint32_t i = 0; i < count; i++) {
    if (ids[i] >= h->cfg.num_attribute) return fail(h, FMX_E_ARG, "feature id %u >= num_attribute", ids[i]);
    if (!sh.owns(ids[i])) return fail(h, FMX_E_ARG, "feature id %u is not on this shard", ids[i]);
  }
  HIPCHK(h, hipSetDevice(h->device));
  uint32_t* d_ids = nullptr; double* d_out = nullptr;
  const size_t nout = (size_t)count * (size_t)(k + 1);
  HIPCHK(h, fmx_dev_alloc(&d_ids, (size_t)count * 4));
  HIPCHK(h, fmx_dev_alloc(&d_out, nout * sizeof(double)));
  hipError_t er = hipMemcpyAsync(d_ids, ids, (size_t)count * 4, hipMemcpyHostToDevice, h->stream);
  if (er == hipSuccess) {
    hipLaunchKernelGGL(k_fetch_rows, dim3((uint32_t)((nout + 255) / 256)), dim3(256), 0, h->stream, d_ids, count, k,
                       sh, h->tb, d_out, d_out + count);
    er = hipGetLastError();
  }
  if (er == hipSuccess) er = hipMemcpyAsync(w_out, d_out, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (er == hipSuccess && k > 0) er = hipMemcpyAsync(v_out, d_out + count, (size_t)count * k * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (er == hipSuccess) er = hipStreamSynchronize(h->stream);
  fmx_dev_free(d_ids); fmx_dev_free(d_out);
  if (er != hipSuccess) return fail(h, FMX_E_HIP, "fmx_get_param_rows: %s", hipGetErrorString(er));
  return FMX_OK;
}

// ---- fm_model::saveModel / loadModel (fm_model.h:132-190) straight from / into the device table ----------------------
// The device keeps the factors FEATURE-major, which is the order of the file's "#pairwise interactions Vj,f" section (one
// line per feature, its k factors in order): the table is streamed in chunks of rows, never converted to the reference's
// fp64 factor-major block (51 GB at the north-star size).  Numbers are written as the reference's ostream writes doubles
// (6 significant digits, printf "%g").
int fmx_save_model(fmx_handle h, const char* path) {
  if (!h || !path) return FMX_E_ARG;
  if (h->cfg.shard_world > 1) return fail(h, FMX_E_UNSUPPORTED, "fmx_save_model on a feature shard: collect the parameters with fmx_get_params");
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  FILE* f = fopen(path, "w");
  if (!f) return fail(h, FMX_E_ARG, "fmx_save_model: cannot open %s", path);
  const uint64_t n = h->cfg.num_attribute;
  const int k = h->cfg.num_factor;
  const size_t chunk = 1u << 16;
  std::vector<float> buf(chunk * (size_t)std::max<uint32_t>(h->tb.rs, 1));
  int rc = FMX_OK;
  if (h->cfg.k0) {
    double w0 = 0;
    if (hipMemcpy(&w0, h->w0, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = FMX_E_HIP;
    fprintf(f, "#global bias W0\n%g\n", w0);
  }
  if (rc == FMX_OK && h->cfg.k1) {
    fprintf(f, "#unary interactions Wj\n");
    for (uint64_t j0 = 0; j0 < n && rc == FMX_OK; j0 += chunk) {
      const size_t cnt = (size_t)std::min<uint64_t>(chunk, n - j0);
      if (h->tb.ws == 1) { if (hipMemcpy(buf.data(), h->tb.w + j0, cnt * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = FMX_E_HIP; }
      else if (hipMemcpy2D(buf.data(), sizeof(float), h->tb.w + j0 * h->tb.ws, (size_t)h->tb.ws * sizeof(float), sizeof(float), cnt, hipMemcpyDeviceToHost) != hipSuccess) rc = FMX_E_HIP;
      for (size_t i = 0; i < cnt && rc == FMX_OK; i++) fprintf(f, "%g\n", (double)buf[i]);
    }
  }
  if (rc == FMX_OK) {
    fprintf(f, "#pairwise interactions Vj,f\n");
    for (uint64_t j0 = 0; j0 < n && rc == FMX_OK; j0 += chunk) {
      const size_t cnt = (size_t)std::min<uint64_t>(chunk, n - j0);
      if (k > 0 && hipMemcpy(buf.data(), h->tb.V + j0 * h->tb.rs, cnt * (size_t)h->tb.rs * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = FMX_E_HIP;
      for (size_t i = 0; i < cnt && rc == FMX_OK; i++) {
        const float* row = buf.data() + i * h->tb.rs;
        for (int ff = 0; ff < k; ff++) { fprintf(f, "%g", (double)row[ff]); if (ff != k - 1) fputc(' ', f); }
        fputc('\n', f);
      }
    }
  }
  const bool werr = ferror(f) != 0;
  if (fclose(f) != 0 || werr) return fail(h, FMX_E_ARG, "fmx_save_model: write error on %s", path);
  if (rc != FMX_OK) return fail(h, rc, "fmx_save_model: device copy failed");
  return FMX_OK;
}

// returns FMX_E_ARG ("malformed model file") where fm_model::loadModel returns 0 (libfm.cpp:264-267).  Every shard of a
// sharded model may load the same file: it keeps its own features.
int fmx_load_model(fmx_handle h, const char* path) {
  touch_w(h);
  if (!h || !path) return FMX_E_ARG;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  FILE* f = fopen(path, "r");
  if (!f) return fail(h, FMX_E_ARG, "malformed model file (cannot open %s)", path);
  const uint64_t n = h->cfg.num_attribute;
  const int k = h->cfg.num_factor;
  const Shard sh = make_shard(h->cfg);
  char* line = nullptr; size_t cap = 0;
  auto next = [&]() -> bool { return getline(&line, &cap, f) >= 0; };
  int rc = FMX_OK;
  const char* what = "";
  std::vector<float> stage;
  std::vector<uint32_t> rows;
  auto flush_w = [&]() {                                    // scatter the staged linear weights to their local rows
    for (size_t i = 0; i < rows.size() && rc == FMX_OK; i++)
      if (hipMemcpyAsync(h->tb.w + (size_t)rows[i] * h->tb.ws, &stage[i], sizeof(float), hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = FMX_E_HIP;
    if (hipStreamSynchronize(h->stream) != hipSuccess) rc = FMX_E_HIP;
    rows.clear(); stage.clear();
  };
  if (h->cfg.k0) {
    double w0 = 0;
    if (!next() || !next()) { rc = FMX_E_ARG; what = "bias section"; }
    else { w0 = atof(line); if (hipMemcpy(h->w0, &w0, sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = FMX_E_HIP; }
  }
  if (rc == FMX_OK && h->cfg.k1) {
    if (!next()) { rc = FMX_E_ARG; what = "linear section"; }
    const size_t chunk = 1u << 16;
    std::vector<float> wbuf;
    uint64_t j0 = 0;
    for (uint64_t j = 0; j < n && rc == FMX_OK; j++) {
      if (!next()) { rc = FMX_E_ARG; what = "linear weights"; break; }
      const float val = (float)atof(line);
      if (sh.world == 1) {
        wbuf.push_back(val);
        if (wbuf.size() == chunk || j + 1 == n) {
          if (h->tb.ws == 1) { if (hipMemcpy(h->tb.w + j0, wbuf.data(), wbuf.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = FMX_E_HIP; }
          else if (hipMemcpy2D(h->tb.w + j0 * h->tb.ws, (size_t)h->tb.ws * sizeof(float), wbuf.data(), sizeof(float), sizeof(float), wbuf.size(), hipMemcpyHostToDevice) != hipSuccess) rc = FMX_E_HIP;
          j0 = j + 1; wbuf.clear();
        }
      } else {
        uint32_t jl;
        if (sh.place((uint32_t)j, &jl)) { rows.push_back(jl); stage.push_back(val); if (rows.size() == chunk) flush_w(); }
      }
    }
    if (rc == FMX_OK && !rows.empty()) flush_w();
  }
  if (rc == FMX_OK && !next()) { rc = FMX_E_ARG; what = "factor section"; }
  if (rc == FMX_OK) {
    const size_t chunk = 1u << 14;
    std::vector<float> vbuf(chunk * (size_t)h->tb.rs, 0.f);
    size_t filled = 0; uint64_t j0 = 0;
    for (uint64_t j = 0; j < n && rc == FMX_OK; j++) {
      if (!next()) { rc = FMX_E_ARG; what = "factor rows"; break; }
      uint32_t jl = (uint32_t)j;
      const bool mine = sh.place((uint32_t)j, &jl);
      float* row = vbuf.data() + filled * h->tb.rs;
      // the reference splits at single blanks and wants exactly num_factor tokens (fm_model.h:176-179)
      int cnt = 0;
      char* p = line;
      size_t len = strlen(p);
      while (len && (p[len - 1] == '\n' || p[len - 1] == '\r')) p[--len] = 0;
      if (k > 0) {
        for (char* tok = p;; ) {
          char* sp = strchr(tok, ' ');
          if (sp) *sp = 0;
          if (cnt < k) row[cnt] = (float)atof(tok);
          cnt++;
          if (!sp) break;
          tok = sp + 1;
        }
        if (cnt != k) { rc = FMX_E_ARG; what = "factor count of a row"; break; }
      }
      if (sh.world == 1) {
        filled++;
        if (filled == chunk || j + 1 == n) {
          if (k > 0 && hipMemcpy(h->tb.V + j0 * h->tb.rs, vbuf.data(), filled * (size_t)h->tb.rs * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = FMX_E_HIP;
          j0 = j + 1; filled = 0;
        }
      } else if (mine && k > 0) {
        if (hipMemcpy(h->tb.V + (size_t)jl * h->tb.rs, row, (size_t)k * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = FMX_E_HIP;
      }
    }
  }
  free(line);
  fclose(f);
  if (rc == FMX_E_ARG) return fail(h, FMX_E_ARG, "malformed model file (%s)", what);
  if (rc != FMX_OK) return fail(h, rc, "fmx_load_model: device copy failed");
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

int fmx_set_groups(fmx_handle h, const uint32_t* group_of_feature, uint32_t num_groups) {
  if (!h) return FMX_E_ARG;
  if (h->als.slot >= 0 || h->sgda.reg) return fail(h, FMX_E_STATE, "fmx_set_groups while an ALS / SGDA session is open");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->grp) { fmx_dev_free(h->grp); h->grp = nullptr; }
  h->num_groups = 1;
  if (!group_of_feature || num_groups <= 1) return FMX_OK;
  // the array is indexed by GLOBAL feature id ([num_attribute]); a shard keeps the entries of its own features
  const Shard sh = make_shard(h->cfg);
  std::vector<uint32_t> local(h->n_local);
  for (uint64_t jl = 0; jl < h->n_local; jl++) {
    const uint32_t j = sh.global(jl);
    local[jl] = ((uint64_t)j < h->cfg.num_attribute) ? group_of_feature[j] : 0u;
    if (local[jl] >= num_groups)
      return fail(h, FMX_E_ARG, "fmx_set_groups: feature %u has group %u >= num_groups %u", j, local[jl], num_groups);
  }
  HIPCHK(h, fmx_dev_alloc(&h->grp, h->n_local * sizeof(uint32_t)));
  HIPCHK(h, hipMemcpy(h->grp, local.data(), h->n_local * sizeof(uint32_t), hipMemcpyHostToDevice));
  h->num_groups = num_groups;
  return FMX_OK;
}

int fmx_init_params(fmx_handle h, double init_mean, double init_stdev, uint64_t seed) {
  touch_w(h);
  if (!h) return FMX_E_ARG;
  { int _rc = lag_flush(h); if (_rc) return _rc; }
  HIPCHK(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(k_init_params, dim3(256 * 8), dim3(256), 0, h->stream, h->tb, h->n_local,
                     h->cfg.num_factor, h->KP, make_shard(h->cfg), (float)init_mean, init_stdev, seed);
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
  { int _rc = slot_in_session(h, slot, "fmx_free_rows"); if (_rc) return _rc; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_slot(h->slots[slot]);
  return FMX_OK;
}

int fmx_upload_rows(fmx_handle h, int slot, const void* entries, const uint64_t* row_ptr, const float* target,
                    uint32_t n_rows, uint64_t nnz) {
  if (!h) return FMX_E_ARG;
  if (slot < 0 || slot >= FMX_MAX_SLOTS) return fail(h, FMX_E_ARG, "slot %d out of range", slot);
  { int _rc = slot_in_session(h, slot, "fmx_upload_rows"); if (_rc) return _rc; }
  if (!row_ptr || (nnz > 0 && !entries)) return fail(h, FMX_E_ARG, "fmx_upload_rows: null entries/row_ptr");
  if (row_ptr[0] != 0 || row_ptr[n_rows] != nnz) return fail(h, FMX_E_ARG, "row_ptr[0] must be 0 and row_ptr[n_rows] == nnz");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_slot(h->slots[slot]);
  const Entry* src = static_cast<const Entry*>(entries);
  const uint64_t n = h->cfg.num_attribute;
  const uint32_t W = (uint32_t)h->cfg.shard_world;
  const Shard sh = make_shard(h->cfg);
  std::vector<Entry> local_ent;
  std::vector<uint64_t> local_ptr;
  const Entry* up_ent = src;
  const uint64_t* up_ptr = row_ptr;
  uint64_t up_nnz = nnz;
  uint32_t max_row = 0;
  // unsharded: the entries go to the device as they are -- start the transfer now (DMA at PCIe speed when the buffer is
  // page-locked, e.g. from fmx_read_binary) and check the ids on the host while it runs
  Slot s;
  if (W == 1) {
    HIPCHK(h, fmx_dev_alloc(&s.ent, std::max<uint64_t>(nnz, 1) * sizeof(Entry)));
    if (nnz && hipMemcpyAsync(s.ent, src, nnz * sizeof(Entry), hipMemcpyHostToDevice, h->stream) != hipSuccess) {
      fmx_dev_free(s.ent); return fail(h, FMX_E_HIP, "fmx_upload_rows: copy of the entries failed");
    }
  }
  // bound check: the reference asserts id < num_attribute (fm_model.h:112)
  for (uint64_t i = 0; i < nnz; i++)
    if (src[i].id >= n) {
      if (s.ent) { hipStreamSynchronize(h->stream); fmx_dev_free(s.ent); }
      return fail(h, FMX_E_ARG, "feature id %u >= num_attribute %llu (row entry %llu)", src[i].id,
                  (unsigned long long)n, (unsigned long long)i);
    }
  if (W > 1) {   // keep this shard's features, ids become local rows (Shard::place)
    local_ptr.resize((size_t)n_rows + 1);
    local_ent.reserve((size_t)(nnz / W + n_rows));
    for (uint32_t r = 0; r < n_rows; r++) {
      local_ptr[r] = local_ent.size();
      for (uint64_t i = row_ptr[r]; i < row_ptr[r + 1]; i++)
        { Entry e; if (sh.place(src[i].id, &e.id)) { e.value = src[i].value; local_ent.push_back(e); } }
    }
    local_ptr[n_rows] = local_ent.size();
    up_ent = local_ent.data(); up_ptr = local_ptr.data(); up_nnz = local_ent.size();
  }
  uint32_t min_row = n_rows ? 0xFFFFFFFFu : 0u;
  for (uint32_t r = 0; r < n_rows; r++) {
    const uint32_t sz = (uint32_t)(up_ptr[r + 1] - up_ptr[r]);
    max_row = std::max(max_row, sz); min_row = std::min(min_row, sz);
  }
  hipError_t er = hipSuccess;
  if (!s.ent) {
    er = fmx_dev_alloc(&s.ent, std::max<uint64_t>(up_nnz, 1) * sizeof(Entry));
    if (er == hipSuccess && up_nnz) er = hipMemcpyAsync(s.ent, up_ent, up_nnz * sizeof(Entry), hipMemcpyHostToDevice, h->stream);
  }
  if (er == hipSuccess) er = fmx_dev_alloc(&s.row_ptr, ((size_t)n_rows + 1) * sizeof(uint64_t));
  if (er == hipSuccess) er = hipMemcpyAsync(s.row_ptr, up_ptr, ((size_t)n_rows + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream);
  if (er == hipSuccess && target) {
    er = fmx_dev_alloc(&s.target, std::max<uint32_t>(n_rows, 1) * sizeof(float));
    if (er == hipSuccess && n_rows) er = hipMemcpyAsync(s.target, target, (size_t)n_rows * sizeof(float), hipMemcpyHostToDevice, h->stream);
  }
  if (hipStreamSynchronize(h->stream) != hipSuccess && er == hipSuccess) er = hipGetLastError();   // the host buffers may go away after return
  if (er != hipSuccess) { free_slot(s); return fail(h, FMX_E_HIP, "fmx_upload_rows: %s", hipGetErrorString(er)); }
  s.n_rows = n_rows; s.nnz = up_nnz; s.max_row = max_row; s.used = true;
  s.fixed_nnz = (n_rows && min_row == max_row && max_row > 0) ? max_row : 0u;
  h->slots[slot] = s;
  return FMX_OK;
}

// FMX_BLOCKS_KEEP: the main rows and every block stay apart on the device -- nothing of the size of the joined table is
// built.  fm_model::predict and the ALS / MCMC sweeps then work block-wise (fmx_als.hip, k_rel_*).
static int upload_blocks_kept(fmx_handle h, int slot, const void* entries, const uint64_t* row_ptr, const float* target,
                              uint32_t n_rows, uint64_t nnz, const fmx_relation* relations, uint32_t n_relations) {
  int rc = fmx_upload_rows(h, slot, entries, row_ptr, target, n_rows, nnz);          // the main rows as any other data set
  if (rc) return rc;
  Slot& s = h->slots[slot];
  for (uint32_t r = 0; r < n_relations; r++) {
    const fmx_relation& q = relations[r];
    BlockRows* b = new BlockRows();
    s.blocks.push_back(b);
    b->attr_offset = (uint32_t)q.attr_offset;
    std::vector<uint32_t> cnt((size_t)q.n_rows + 1, 0), list(std::max<uint32_t>(n_rows, 1));
    for (uint32_t c = 0; c < n_rows; c++) cnt[q.data_row_to_relation_row[c] + 1]++;
    for (uint32_t i = 0; i < q.n_rows; i++) cnt[i + 1] += cnt[i];
    { std::vector<uint32_t> fill(cnt.begin(), cnt.end() - 1);
      for (uint32_t c = 0; c < n_rows; c++) list[fill[q.data_row_to_relation_row[c]]++] = c; }   // ascending main row inside a block row
    uint32_t max_row = 0;
    for (uint32_t i = 0; i < q.n_rows; i++) max_row = std::max<uint32_t>(max_row, (uint32_t)(q.row_ptr[i + 1] - q.row_ptr[i]));
    hipError_t er = fmx_dev_alloc(&b->rows.ent, std::max<uint64_t>(q.nnz, 1) * sizeof(Entry));
    if (er == hipSuccess && q.nnz) er = hipMemcpy(b->rows.ent, q.entries, q.nnz * sizeof(Entry), hipMemcpyHostToDevice);
    if (er == hipSuccess) er = fmx_dev_alloc(&b->rows.row_ptr, ((size_t)q.n_rows + 1) * sizeof(uint64_t));
    if (er == hipSuccess) er = hipMemcpy(b->rows.row_ptr, q.row_ptr, ((size_t)q.n_rows + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (er == hipSuccess) er = fmx_dev_alloc(&b->map, std::max<uint32_t>(n_rows, 1) * sizeof(uint32_t));
    if (er == hipSuccess && n_rows) er = hipMemcpy(b->map, q.data_row_to_relation_row, (size_t)n_rows * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (er == hipSuccess) er = fmx_dev_alloc(&b->brow_ptr, cnt.size() * sizeof(uint32_t));
    if (er == hipSuccess) er = hipMemcpy(b->brow_ptr, cnt.data(), cnt.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (er == hipSuccess) er = fmx_dev_alloc(&b->brow_list, list.size() * sizeof(uint32_t));
    if (er == hipSuccess) er = hipMemcpy(b->brow_list, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (er == hipSuccess) er = fmx_dev_alloc(&b->pbuf, std::max<size_t>((size_t)q.n_rows, 1) * (size_t)(h->KP + 1) * sizeof(float));
    if (er != hipSuccess) { free_slot(s); return fail(h, FMX_E_HIP, "fmx_upload_block_rows_ex: %s", hipGetErrorString(er)); }
    b->rows.n_rows = q.n_rows; b->rows.nnz = q.nnz; b->rows.max_row = max_row; b->rows.used = true;
  }
  return FMX_OK;
}

int fmx_upload_block_rows(fmx_handle h, int slot, const void* entries, const uint64_t* row_ptr, const float* target,
                          uint32_t n_rows, uint64_t nnz, const fmx_relation* relations, uint32_t n_relations) {
  return fmx_upload_block_rows_ex(h, slot, entries, row_ptr, target, n_rows, nnz, relations, n_relations, FMX_BLOCKS_EXPAND);
}

int fmx_upload_block_rows_ex(fmx_handle h, int slot, const void* entries, const uint64_t* row_ptr, const float* target,
                             uint32_t n_rows, uint64_t nnz, const fmx_relation* relations, uint32_t n_relations, uint32_t flags) {
  if (!h) return FMX_E_ARG;
  if (n_relations == 0) return fmx_upload_rows(h, slot, entries, row_ptr, target, n_rows, nnz);
  if (slot < 0 || slot >= FMX_MAX_SLOTS) return fail(h, FMX_E_ARG, "slot %d out of range", slot);
  { int _rc = slot_in_session(h, slot, "fmx_upload_block_rows"); if (_rc) return _rc; }
  if (!relations || n_relations > FMX_MAX_RELATIONS) return fail(h, FMX_E_ARG, "fmx_upload_block_rows: 1..%d relations", FMX_MAX_RELATIONS);
  if (h->cfg.shard_world > 1 && (flags & FMX_BLOCKS_KEEP))
    return fail(h, FMX_E_UNSUPPORTED, "kept relation blocks on a feature shard are not implemented: upload them with FMX_BLOCKS_EXPAND");
  if (!row_ptr || (nnz > 0 && !entries)) return fail(h, FMX_E_ARG, "fmx_upload_block_rows: null entries/row_ptr");
  if (row_ptr[0] != 0 || row_ptr[n_rows] != nnz) return fail(h, FMX_E_ARG, "row_ptr[0] must be 0 and row_ptr[n_rows] == nnz");
  const uint64_t n = h->cfg.num_attribute;
  const Entry* src = static_cast<const Entry*>(entries);
  for (uint64_t i = 0; i < nnz; i++)
    if (src[i].id >= n) return fail(h, FMX_E_ARG, "feature id %u >= num_attribute %llu", src[i].id, (unsigned long long)n);
  for (uint32_t r = 0; r < n_relations; r++) {
    const fmx_relation& q = relations[r];
    if (!q.row_ptr || !q.data_row_to_relation_row || (q.nnz && !q.entries) || q.row_ptr[0] != 0 || q.row_ptr[q.n_rows] != q.nnz)
      return fail(h, FMX_E_ARG, "relation %u: malformed rows", r);
    const Entry* qe = static_cast<const Entry*>(q.entries);
    for (uint64_t i = 0; i < q.nnz; i++)
      if ((uint64_t)qe[i].id + q.attr_offset >= n)
        return fail(h, FMX_E_ARG, "relation %u: attribute %u + offset %llu >= num_attribute %llu", r, qe[i].id,
                    (unsigned long long)q.attr_offset, (unsigned long long)n);
    for (uint32_t c = 0; c < n_rows; c++)
      if (q.data_row_to_relation_row[c] >= q.n_rows)
        return fail(h, FMX_E_ARG, "relation %u: main row %u maps to block row %u >= %u", r, c, q.data_row_to_relation_row[c], q.n_rows);
  }
  if (flags & FMX_BLOCKS_KEEP) return upload_blocks_kept(h, slot, entries, row_ptr, target, n_rows, nnz, relations, n_relations);
  if (h->cfg.shard_world > 1) {
    // a feature shard: the joined rows (main entries, then every block's mapped row with its ids shifted, libfm.cpp:213-216) are
    // built on the host and go through fmx_upload_rows, which keeps the shard's own features like for any rows
    std::vector<uint64_t> jp((size_t)n_rows + 1, 0);
    for (uint32_t c = 0; c < n_rows; c++) {
      uint64_t sz = row_ptr[c + 1] - row_ptr[c];
      for (uint32_t r = 0; r < n_relations; r++) { const uint32_t b = relations[r].data_row_to_relation_row[c]; sz += relations[r].row_ptr[b + 1] - relations[r].row_ptr[b]; }
      jp[c + 1] = jp[c] + sz;
    }
    std::vector<Entry> je((size_t)jp[n_rows]);
    for (uint32_t c = 0; c < n_rows; c++) {
      uint64_t o = jp[c];
      for (uint64_t i = row_ptr[c]; i < row_ptr[c + 1]; i++) je[o++] = src[i];
      for (uint32_t r = 0; r < n_relations; r++) {
        const fmx_relation& q = relations[r];
        const Entry* qe = static_cast<const Entry*>(q.entries);
        const uint32_t b = q.data_row_to_relation_row[c];
        for (uint64_t i = q.row_ptr[b]; i < q.row_ptr[b + 1]; i++) { Entry e = qe[i]; e.id += (uint32_t)q.attr_offset; je[o++] = e; }
      }
    }
    return fmx_upload_rows(h, slot, je.data(), jp.data(), target, n_rows, jp[n_rows]);
  }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_slot(h->slots[slot]);
  std::vector<void*> tmp;                                     // staging buffers, freed on every exit path
  auto up = [&](const void* p, size_t bytes) -> void* {
    void* d = nullptr;
    if (fmx_dev_alloc(&d, std::max<size_t>(bytes, 8)) != hipSuccess) return nullptr;
    tmp.push_back(d);
    if (bytes && hipMemcpy(d, p, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
  };
  Slot s;
  bool committed = false;
  auto cleanup = [&]() { for (void* d : tmp) fmx_dev_free(d); if (!committed) free_slot(s); };
#define BLK_CHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail(h, FMX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
#define BLK_UP(dst, type, p, bytes) do { dst = (type)up((p), (bytes)); if (!dst) { cleanup(); return fail(h, FMX_E_HIP, "fmx_upload_block_rows: staging %s failed", #p); } } while (0)
  const Entry* d_ent; const uint64_t* d_ptr;
  BLK_UP(d_ent, const Entry*, entries, nnz * sizeof(Entry));
  BLK_UP(d_ptr, const uint64_t*, row_ptr, ((size_t)n_rows + 1) * sizeof(uint64_t));
  BlockRels rels;
  memset(&rels, 0, sizeof(rels));
  rels.n = n_relations;
  for (uint32_t r = 0; r < n_relations; r++) {
    const fmx_relation& q = relations[r];
    BLK_UP(rels.r[r].ent, const Entry*, q.entries, q.nnz * sizeof(Entry));
    BLK_UP(rels.r[r].row_ptr, const uint64_t*, q.row_ptr, ((size_t)q.n_rows + 1) * sizeof(uint64_t));
    BLK_UP(rels.r[r].map, const uint32_t*, q.data_row_to_relation_row, (size_t)n_rows * sizeof(uint32_t));
    rels.r[r].attr_offset = (uint32_t)q.attr_offset;
  }
  uint64_t* sizes = nullptr;
  BLK_CHK(fmx_dev_alloc(&sizes, ((size_t)n_rows + 1) * sizeof(uint64_t)));
  tmp.push_back(sizes);
  BLK_CHK(fmx_dev_alloc(&s.row_ptr, ((size_t)n_rows + 1) * sizeof(uint64_t)));
  const dim3 g1(std::min<uint32_t>((n_rows + 256) / 256, 2048)), b1(256);
  hipLaunchKernelGGL(k_block_sizes, g1, b1, 0, h->stream, d_ptr, n_rows, rels, sizes);
  size_t scan_bytes = 0;
  BLK_CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, sizes, s.row_ptr, (int)(n_rows + 1), h->stream));
  void* scan_tmp = nullptr;
  BLK_CHK(fmx_dev_alloc(&scan_tmp, std::max<size_t>(scan_bytes, 8)));
  tmp.push_back(scan_tmp);
  BLK_CHK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, sizes, s.row_ptr, (int)(n_rows + 1), h->stream));
  std::vector<uint64_t> hs((size_t)n_rows + 1);
  uint64_t total = 0;
  BLK_CHK(hipMemcpyAsync(hs.data(), sizes, hs.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
  BLK_CHK(hipMemcpyAsync(&total, s.row_ptr + n_rows, sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
  BLK_CHK(hipStreamSynchronize(h->stream));
  uint32_t max_row = 0;
  for (uint32_t c = 0; c < n_rows; c++) max_row = std::max<uint32_t>(max_row, (uint32_t)hs[c]);
  BLK_CHK(fmx_dev_alloc(&s.ent, std::max<uint64_t>(total, 1) * sizeof(Entry)));
  hipLaunchKernelGGL(k_block_fill, g1, b1, 0, h->stream, d_ent, d_ptr, n_rows, rels, s.row_ptr, s.ent);
  BLK_CHK(hipGetLastError());
  if (target) {
    BLK_CHK(fmx_dev_alloc(&s.target, std::max<uint32_t>(n_rows, 1) * sizeof(float)));
    if (n_rows) BLK_CHK(hipMemcpy(s.target, target, (size_t)n_rows * sizeof(float), hipMemcpyHostToDevice));
  }
  BLK_CHK(hipStreamSynchronize(h->stream));
#undef BLK_CHK
#undef BLK_UP
  committed = true;
  cleanup();
  s.n_rows = n_rows; s.nnz = total; s.max_row = max_row; s.used = true;
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
  return fmx_synth_rows_ex(h, slot, seed, row0, n_rows, nnz, FMX_SYNTH_UNIFORM);
}

int fmx_synth_rows_ex(fmx_handle h, int slot, uint64_t seed, uint64_t row0, uint32_t n_rows, uint32_t nnz, uint32_t shape) {
  if (!h) return FMX_E_ARG;
  if (slot < 0 || slot >= FMX_MAX_SLOTS) return fail(h, FMX_E_ARG, "slot %d out of range", slot);
  { int _rc = slot_in_session(h, slot, "fmx_synth_rows"); if (_rc) return _rc; }
  if (nnz == 0 || n_rows == 0) return fail(h, FMX_E_ARG, "fmx_synth_rows: empty workload");
  if (shape != FMX_SYNTH_UNIFORM && shape != FMX_SYNTH_CRITEO) return fail(h, FMX_E_ARG, "fmx_synth_rows_ex: unknown shape %u", shape);
  const uint64_t n = h->cfg.num_attribute;
  uint32_t fs = (uint32_t)(n / nnz);
  if (shape == FMX_SYNTH_CRITEO) {
    const uint64_t dense = (uint64_t)SYNTH_DENSE_FIELDS * SYNTH_DENSE_IDS;
    if (nnz <= SYNTH_DENSE_FIELDS || n < dense + (nnz - SYNTH_DENSE_FIELDS))
      return fail(h, FMX_E_ARG, "fmx_synth_rows_ex: the Criteo shape needs nnz > 13 and num_attribute >= 1300 + (nnz - 13)");
    fs = (uint32_t)((n - dense) / (nnz - SYNTH_DENSE_FIELDS));
  }
  if (fs == 0) return fail(h, FMX_E_ARG, "fmx_synth_rows: num_attribute < nnz");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_slot(h->slots[slot]);
  const Shard sh = make_shard(h->cfg);
  Slot s;
  uint32_t* cnt = nullptr;
  void* tmp = nullptr;
  uint64_t total = 0;
  // every error path releases what was allocated so far
#define SYN_CHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { if (cnt) fmx_dev_free(cnt); if (tmp) fmx_dev_free(tmp); free_slot(s); \
    return fail(h, FMX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
  SYN_CHK(fmx_dev_alloc(&cnt, ((size_t)n_rows + 1) * sizeof(uint32_t)));
  SYN_CHK(hipMemsetAsync(cnt, 0, ((size_t)n_rows + 1) * sizeof(uint32_t), h->stream));
  SYN_CHK(fmx_dev_alloc(&s.row_ptr, ((size_t)n_rows + 1) * sizeof(uint64_t)));
  SYN_CHK(fmx_dev_alloc(&s.target, (size_t)n_rows * sizeof(float)));
  const dim3 grid((n_rows + 255) / 256), block(256);
  hipLaunchKernelGGL(k_synth, grid, block, 0, h->stream, seed, row0, n_rows, nnz, fs, shape, sh, cnt,
                     (const uint64_t*)nullptr, (Entry*)nullptr, s.target);
  SYN_CHK(hipGetLastError());
  {  // exclusive prefix sum u32 -> u64 over n_rows+1 items (last = total)
    size_t tmp_bytes = 0;
    auto conv = hipcub::TransformInputIterator<uint64_t, hipcub::CastOp<uint64_t>, uint32_t*>(cnt, hipcub::CastOp<uint64_t>());
    SYN_CHK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, conv, s.row_ptr, (int)(n_rows + 1), h->stream));
    SYN_CHK(fmx_dev_alloc(&tmp, tmp_bytes));
    SYN_CHK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, conv, s.row_ptr, (int)(n_rows + 1), h->stream));
    SYN_CHK(hipStreamSynchronize(h->stream));
    fmx_dev_free(tmp); tmp = nullptr;
  }
  SYN_CHK(hipMemcpy(&total, s.row_ptr + n_rows, sizeof(uint64_t), hipMemcpyDeviceToHost));
  SYN_CHK(fmx_dev_alloc(&s.ent, std::max<uint64_t>(total, 1) * sizeof(Entry)));
  hipLaunchKernelGGL(k_synth, grid, block, 0, h->stream, seed, row0, n_rows, nnz, fs, shape, sh, (uint32_t*)nullptr,
                     (const uint64_t*)s.row_ptr, s.ent, (float*)nullptr);
  SYN_CHK(hipGetLastError());
  SYN_CHK(hipStreamSynchronize(h->stream));
#undef SYN_CHK
  fmx_dev_free(cnt);
  s.n_rows = n_rows; s.nnz = total; s.max_row = nnz; s.used = true;
  s.fixed_nnz = (h->cfg.shard_world == 1) ? nnz : 0u;        // (a shard keeps a varying part of every row)
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
  if (s.wside && s.wside_version == h->w_version && s.blocks.empty()) out->flags |= FMX_EVAL_WSIDE;
  return FMX_OK;
}

}  // extern "C"

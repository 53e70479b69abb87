// fmx_io.hip -- C-ABI (include/fmx.h): host-side readers of libFM's data formats.  No device code, no handle: these
// produce exactly the buffers fmx_upload_rows takes.  Behaviour follows Data::load (src/libfm/src/Data.h:180-285):
// the same tokens are accepted (sscanf "%f" for the target, "%d:%f" pairs), the same lines are skipped (blank, '#'),
// and a line the reference would throw on is rejected with the reference's message.
#pragma GCC visibility push(default)      // the C-ABI is the only thing libfmx.so exports (-fvisibility=hidden)
#include "../../include/fmx.h"
#pragma GCC visibility pop

#include <hip/hip_runtime_api.h>          // hipHostMalloc for the binary reader's pinned staging buffers (no kernels here)

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

namespace {

struct Entry { uint32_t id; float value; };                 // sparse_entry<float>, src/util/fmatrix.h:34-37
static_assert(sizeof(Entry) == 8, "AoS entry layout");

// what sscanf("%d", ...) consumes: optional white space, optional sign, decimal digits
bool scan_int(const char*& p, long& out) {
  const char* q = p;
  while (isspace((unsigned char)*q)) q++;
  char* end = nullptr;
  if (!(isdigit((unsigned char)*q) || ((*q == '-' || *q == '+') && isdigit((unsigned char)q[1])))) return false;
  out = strtol(q, &end, 10);
  if (end == q) return false;
  p = end;
  return true;
}
// what sscanf("%f", ...) consumes: optional white space, then a strtof number
bool scan_float(const char*& p, float& out) {
  const char* q = p;
  while (isspace((unsigned char)*q)) q++;
  char* end = nullptr;
  out = strtof(q, &end);
  if (end == q) return false;
  p = end;
  return true;
}

int io_fail(char* err, size_t err_len, int code, const std::string& msg) {
  if (err && err_len) { snprintf(err, err_len, "%s", msg.c_str()); }
  return code;
}

}  // namespace

extern "C" {

void fmx_free_host_rows(fmx_host_rows* r) {
  if (!r) return;
  if (r->flags & FMX_HOST_PINNED) { if (r->entries) hipHostFree(r->entries); if (r->row_ptr) hipHostFree(r->row_ptr); if (r->target) hipHostFree(r->target); }
  else { free(r->entries); free(r->row_ptr); free(r->target); }
  memset(r, 0, sizeof(*r));
}

}  // extern "C"

// ---- Data::load, binary branch (src/libfm/src/Data.h:119-178) ---------------------------------------------------------
// <prefix>.x   LargeSparseMatrix::saveToBinaryFile (src/util/fmatrix.h:44-50 header, :121-140 body): file_header
//              {uint id = 2; uint float_size = 4; uint64 num_values; uint num_rows; uint num_cols} then per row
//              {uint size; sparse_entry<float>[size]}  -- written by tools/convert.cpp:137-200
// <prefix>.xt  the transposed matrix in the same format (tools/transpose.cpp; what als / mcmc read, libfm.cpp:143-147)
// <prefix>.y   DVector<float>::saveToBinaryFile (src/util/matrix.h:344-358): {uint 1; uint 4; uint dim} + floats
// (and the older names .data / .datat / .target, Data.h:120-121).  The rows land in page-locked host memory when a HIP
// device is present, so fmx_upload_rows moves them by DMA at PCIe speed while it validates the ids.
namespace {
struct XHeader { uint32_t id, float_size; uint64_t num_values; uint32_t num_rows, num_cols; };
static_assert(sizeof(XHeader) == 24, "file_header layout (fmatrix.h:44-50)");

struct HostBuf {                       // page-locked when possible
  bool pinned = false;
  void* alloc(size_t bytes) {
    void* p = nullptr;
    if (pinned && hipHostMalloc(&p, std::max<size_t>(bytes, 8), hipHostMallocDefault) == hipSuccess) return p;
    if (pinned) { (void)hipGetLastError(); return nullptr; }
    return malloc(std::max<size_t>(bytes, 8));
  }
};

bool file_exists(const std::string& p) { FILE* f = fopen(p.c_str(), "rb"); if (f) fclose(f); return f != nullptr; }

// reads one matrix file into (entries, row_ptr); returns "" or the error text
std::string read_matrix(const std::string& path, HostBuf& hb, Entry** ent_out, uint64_t** ptr_out, XHeader* hdr_out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return "could not open " + path;                                            // fmatrix.h:197
  XHeader h;
  if (fread(&h, sizeof(h), 1, f) != 1) { fclose(f); return path + ": truncated header"; }
  if (h.id != 2 || h.float_size != 4) { fclose(f); return path + ": not a libFM binary matrix (file id / float size; fmatrix.h:188-189)"; }
  {  // the header's counts size the buffers: check them against the file before allocating (every row costs 4 bytes, every entry 8)
    const long at = ftell(f);
    fseek(f, 0, SEEK_END);
    const uint64_t rest = (uint64_t)(ftell(f) - at);
    fseek(f, at, SEEK_SET);
    if (h.num_values > rest / sizeof(Entry) || (uint64_t)h.num_rows * 4u + h.num_values * sizeof(Entry) > rest) { fclose(f); return path + ": truncated (the header announces more than the file holds)"; }
  }
  Entry* ent = (Entry*)hb.alloc(h.num_values * sizeof(Entry));
  *ent_out = ent;                                                                    // owned by the caller from here on (also on failure below)
  uint64_t* ptr = (uint64_t*)hb.alloc(((size_t)h.num_rows + 1) * sizeof(uint64_t));
  *ptr_out = ptr;
  if (!ent || !ptr) { fclose(f); return path + ": out of (page-locked) memory"; }
  uint64_t pos = 0;
  ptr[0] = 0;
  for (uint32_t r = 0; r < h.num_rows; r++) {
    uint32_t size;
    if (fread(&size, 4, 1, f) != 1) { fclose(f); return path + ": truncated (row sizes)"; }
    if (pos + size > h.num_values) { fclose(f); return path + ": more entries than its header announces"; }
    if (size && fread(ent + pos, sizeof(Entry), size, f) != size) { fclose(f); return path + ": truncated (row entries)"; }
    pos += size;
    ptr[r + 1] = pos;
  }
  fclose(f);
  if (pos != h.num_values) return path + ": fewer entries than its header announces";
  *hdr_out = h;
  return "";
}
}  // namespace

extern "C" {

int fmx_read_binary(const char* prefix, fmx_host_rows* out, char* err, size_t err_len) {
  if (!prefix || !out) return io_fail(err, err_len, FMX_E_ARG, "fmx_read_binary: null argument");
  memset(out, 0, sizeof(*out));
  const std::string p(prefix);
  std::string fx, fxt, fy;
  if (file_exists(p + ".target") && (file_exists(p + ".data") || file_exists(p + ".datat"))) {       // Data.h:120-121
    fy = p + ".target"; if (file_exists(p + ".data")) fx = p + ".data"; else fxt = p + ".datat";
  } else if (file_exists(p + ".y") && (file_exists(p + ".x") || file_exists(p + ".xt"))) {          // Data.h:122-123
    fy = p + ".y"; if (file_exists(p + ".x")) fx = p + ".x"; else fxt = p + ".xt";
  } else {
    return io_fail(err, err_len, FMX_E_ARG, "fmx_read_binary: neither " + p + ".x/.xt + .y nor " + p + ".data/.datat + .target exist");
  }
  HostBuf hb;
  int ndev = 0;
  hb.pinned = (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0);
  if (!hb.pinned) (void)hipGetLastError();
  out->flags = hb.pinned ? FMX_HOST_PINNED : 0u;
  // ---- targets
  FILE* f = fopen(fy.c_str(), "rb");
  if (!f) return io_fail(err, err_len, FMX_E_ARG, "could not open " + fy);
  uint32_t yh[3];
  if (fread(yh, 4, 3, f) != 3 || yh[0] != 1 || yh[1] != 4) { fclose(f); return io_fail(err, err_len, FMX_E_ARG, fy + ": not a libFM binary float vector (matrix.h:360-380)"); }
  const uint32_t n_rows = yh[2];
  out->target = (float*)hb.alloc((size_t)n_rows * sizeof(float));
  if (!out->target || (n_rows && fread(out->target, 4, n_rows, f) != n_rows)) { fclose(f); fmx_free_host_rows(out); return io_fail(err, err_len, FMX_E_ARG, fy + ": truncated or out of memory"); }
  fclose(f);
  // ---- design matrix
  XHeader h;
  Entry* ent = nullptr; uint64_t* ptr = nullptr;
  const std::string msg = read_matrix(fx.empty() ? fxt : fx, hb, &ent, &ptr, &h);
  out->entries = ent; out->row_ptr = ptr;
  if (!msg.empty()) { fmx_free_host_rows(out); return io_fail(err, err_len, FMX_E_ARG, msg); }
  if (!fx.empty()) {
    if (h.num_rows != n_rows) { fmx_free_host_rows(out); return io_fail(err, err_len, FMX_E_ARG, fx + ": row count differs from " + fy + " (Data.h:144)"); }
    out->num_feature = h.num_cols;                                                    // Data.h:145
  } else {
    // only X^T on disk (the als / mcmc input, libfm.cpp:143-147): rebuild the rows.  Column c of X^T lists {row, value} in
    // ascending row order, so one counting pass + one scatter gives every row with ascending feature ids.
    if (h.num_cols != n_rows) { fmx_free_host_rows(out); return io_fail(err, err_len, FMX_E_ARG, fxt + ": column count differs from " + fy); }
    Entry* x = (Entry*)hb.alloc(h.num_values * sizeof(Entry));
    uint64_t* xp = (uint64_t*)hb.alloc(((size_t)n_rows + 1) * sizeof(uint64_t));
    bool ok = x && xp;
    if (ok) {
      memset(xp, 0, ((size_t)n_rows + 1) * sizeof(uint64_t));
      for (uint64_t i = 0; i < h.num_values && ok; i++) { if (ent[i].id >= n_rows) ok = false; else xp[ent[i].id + 1]++; }
    }
    if (ok) {
      for (uint32_t r = 0; r < n_rows; r++) xp[r + 1] += xp[r];
      std::vector<uint64_t> fill(xp, xp + n_rows);
      for (uint32_t j = 0; j < h.num_rows; j++)
        for (uint64_t i = ptr[j]; i < ptr[j + 1]; i++) { Entry e; e.id = j; e.value = ent[i].value; x[fill[ent[i].id]++] = e; }
    }
    if (hb.pinned) { hipHostFree(ent); hipHostFree(ptr); } else { free(ent); free(ptr); }
    out->entries = x; out->row_ptr = xp;
    if (!ok) { fmx_free_host_rows(out); return io_fail(err, err_len, FMX_E_ARG, fxt + ": a row id exceeds the number of cases, or out of memory"); }
    out->num_feature = h.num_rows;                                                    // Data.h:157
  }
  out->n_rows = n_rows; out->nnz = h.num_values;
  float min_t = std::numeric_limits<float>::max(), max_t = -std::numeric_limits<float>::max();     // Data.h:166-171
  for (uint32_t r = 0; r < n_rows; r++) { min_t = std::min(min_t, out->target[r]); max_t = std::max(max_t, out->target[r]); }
  out->min_target = n_rows ? min_t : 0.f; out->max_target = n_rows ? max_t : 0.f;
  return FMX_OK;
}

int fmx_read_libsvm(const char* path, fmx_host_rows* out, char* err, size_t err_len) {
  if (!path || !out) return io_fail(err, err_len, FMX_E_ARG, "fmx_read_libsvm: null argument");
  memset(out, 0, sizeof(*out));
  FILE* f = fopen(path, "r");
  if (!f) return io_fail(err, err_len, FMX_E_ARG, std::string("unable to open ") + path);      // Data.h:194
  std::vector<Entry> ent;
  std::vector<uint64_t> row_ptr(1, 0);
  std::vector<float> target;
  float min_t = std::numeric_limits<float>::max(), max_t = -std::numeric_limits<float>::max();
  long max_feature = -1;
  char* line = nullptr;
  size_t cap = 0;
  ssize_t len;
  int rc = FMX_OK;
  std::string msg;
  while ((len = getline(&line, &cap, f)) >= 0) {
    if (len > 0 && line[len - 1] == '\n') line[len - 1] = 0;            // std::getline drops the newline only
    const char* p = line;
    while (*p == ' ' || *p == 9) p++;                                    // Data.h:201
    if (*p == 0 || *p == '#') continue;                                  // :202
    float v;
    if (!scan_float(p, v)) { rc = FMX_E_ARG; msg = std::string("cannot parse line \"") + line + "\" at character " + p[0]; break; }
    target.push_back(v);
    min_t = std::min(min_t, v); max_t = std::max(max_t, v);
    for (;;) {                                                           // :208-213 "%d:%f"
      const char* q = p;
      long id; float x;
      if (!scan_int(q, id) || *q != ':') break;
      q++;
      if (!scan_float(q, x)) break;
      if (id < 0 || id > 0xFFFFFFFFl) { rc = FMX_E_ARG; msg = std::string("feature id out of range in line \"") + line + "\""; break; }
      Entry e; e.id = (uint32_t)id; e.value = x;
      ent.push_back(e);
      max_feature = std::max(max_feature, id);
      p = q;
    }
    if (rc) break;
    row_ptr.push_back(ent.size());
    while (*p != 0 && (*p == ' ' || *p == 9)) p++;                       // :214
    if (*p != 0 && *p != '#') { rc = FMX_E_ARG; msg = std::string("cannot parse line \"") + line + "\" at character " + p[0]; break; }
  }
  free(line);
  fclose(f);
  if (rc) return io_fail(err, err_len, rc, msg);
  const size_t n_rows = target.size();
  out->entries = malloc(std::max<size_t>(ent.size(), 1) * sizeof(Entry));
  out->row_ptr = (uint64_t*)malloc((n_rows + 1) * sizeof(uint64_t));
  out->target = (float*)malloc(std::max<size_t>(n_rows, 1) * sizeof(float));
  if (!out->entries || !out->row_ptr || !out->target) { fmx_free_host_rows(out); return io_fail(err, err_len, FMX_E_ARG, "fmx_read_libsvm: out of memory"); }
  if (!ent.empty()) memcpy(out->entries, ent.data(), ent.size() * sizeof(Entry));
  memcpy(out->row_ptr, row_ptr.data(), (n_rows + 1) * sizeof(uint64_t));
  if (n_rows) memcpy(out->target, target.data(), n_rows * sizeof(float));
  out->n_rows = (uint32_t)n_rows; out->nnz = ent.size();
  out->num_feature = (uint32_t)(max_feature + 1);                        // :226-228
  out->min_target = n_rows ? min_t : 0.f; out->max_target = n_rows ? max_t : 0.f;
  return FMX_OK;
}

}  // extern "C"

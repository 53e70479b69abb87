// fmx_io.hip -- C-ABI (include/fmx.h): host-side readers of libFM's data formats.  No device code, no handle: these
// produce exactly the buffers fmx_upload_rows takes.  Behaviour follows Data::load (src/libfm/src/Data.h:180-285):
// the same tokens are accepted (sscanf "%f" for the target, "%d:%f" pairs), the same lines are skipped (blank, '#'),
// and a line the reference would throw on is rejected with the reference's message.
#pragma GCC visibility push(default)      // the C-ABI is the only thing libfmx.so exports (-fvisibility=hidden)
#include "../../include/fmx.h"
#pragma GCC visibility pop

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
  free(r->entries); free(r->row_ptr); free(r->target);
  memset(r, 0, sizeof(*r));
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

/*
 * fm_oracle_als.c -- CPU restatement of the ALS (coordinate descent) learner.
 * TEST INFRASTRUCTURE ONLY (see fm_oracle.h).
 */
#include "fm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Data::create_data_t                          /root/reference/src/libfm/src/Data.h:292-341
 * counting sort of the entries by feature id; within a feature the row ids ascend. */
void fmo_transpose(const fmo_data *d, uint64_t n, fmo_entry **entries_t, uint64_t **col_ptr) {
  uint64_t nnz = d->row_ptr[d->n_rows];
  uint64_t *cp = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
  fmo_entry *et = (fmo_entry *)malloc(sizeof(fmo_entry) * (size_t)(nnz ? nnz : 1));
  for (uint64_t i = 0; i < nnz; i++) cp[d->entries[i].id + 1]++;
  for (uint64_t j = 0; j < n; j++) cp[j + 1] += cp[j];
  uint64_t *fill = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
  memcpy(fill, cp, sizeof(uint64_t) * (size_t)n);
  for (uint32_t r = 0; r < d->n_rows; r++)
    for (uint64_t i = d->row_ptr[r]; i < d->row_ptr[r + 1]; i++) {
      uint64_t pos = fill[d->entries[i].id]++;
      et[pos].id = r;
      et[pos].value = d->entries[i].value;
    }
  free(fill);
  *entries_t = et;
  *col_ptr = cp;
}

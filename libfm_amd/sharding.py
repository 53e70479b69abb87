"""Host-side description of the multi-GPU decomposition (no compute): which rank owns a feature, how a rank's rows
look, and the layout of the per-minibatch exchange buffer.  libfmx.so implements exactly this in fmx_upload_rows /
fmx_synth_rows (filtering) and fmx_sgd_partial / fmx_sgd_finish (buffer); bench.py and the tests use these helpers
so that the N>1 path is correct by construction and checkable on CPU (tests/test_sharded_gloo.py).

Decomposition (BASELINE.json north_star): V and w are row-sharded by feature id -- owner and local
row of a feature come from the library's ownership rule (fmx_shard_place: plain mod / div, or the hashed permutation);
every rank sees EVERY example restricted to its own features;
w0 is replicated.  Per minibatch one all-reduce (sum) of

    buffer[0 : B*KP]        S[e][f]  = sum over the rank's features of v[f][j] * x_ej       (fm_model.h:116-125)
    buffer[B*KP : B*KP+B]   c[e]     = sum over the rank's features of (w_j x_ej - 0.5 * sum_f (v_fj x_ej)^2)

after which every rank holds the global sums, computes rest_e = c_e + 0.5 * sum_f S_ef^2, the multipliers and the
w0 recurrence (identical on all ranks) and updates only its own rows.
"""
import numpy as np

from . import capi


def place(ids, n, world, shard_hash=0):
    """(owner, local_row) of feature ids: the library's own rule (fmx_shard_place, host arithmetic: plain = id mod /
    div world; hashed = the same on a fixed pseudo-random permutation of [0, n), include/fmx.h fmx_config::shard_hash)"""
    return capi.shard_place(n, world, shard_hash, ids)


def owned_ids(n, rank, world, shard_hash=0):
    """global ids of the rank's local rows 0 .. n_local-1 (fmx_shard_global)"""
    return capi.shard_global(n, world, shard_hash, rank, np.arange(n_local(n, rank, world), dtype=np.uint32))


def n_local(n, rank, world):
    return (n - rank + world - 1) // world if n > rank else 0


def filter_rows(entries, row_ptr, rank, world, n=None, shard_hash=0):
    """entries/row_ptr of the rows restricted to the rank's features, ids renumbered to local rows (what
    fmx_upload_rows does on a sharded handle)."""
    ids = entries["id"]
    if n is None:
        n = int(ids.max()) + 1 if len(ids) else 1
    own, loc = place(ids, n, world, shard_hash)
    keep = own == rank
    row_of = np.repeat(np.arange(len(row_ptr) - 1), np.diff(row_ptr.astype(np.int64)))
    out = entries[keep].copy()
    out["id"] = loc[keep]
    counts = np.bincount(row_of[keep], minlength=len(row_ptr) - 1)
    new_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    return out, new_ptr


def partial_floats(batch, k_padded):
    return batch * (k_padded + 1)


def split_partial(buf, n_rows, k_padded):
    """views (S[n_rows][KP], c[n_rows]) of an exchange buffer."""
    return buf[: n_rows * k_padded].reshape(n_rows, k_padded), buf[n_rows * k_padded: n_rows * (k_padded + 1)]

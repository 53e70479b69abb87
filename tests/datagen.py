"""Seeded data generators for the parity tests (numpy only; shared by CPU and GPU tests).

Shapes follow SURVEY.md section 8(d): an ML-100K-shaped user x item regression set, ragged real-valued
classification rows, and the field-structured one-hot synthetic workload (see oracle/fm_oracle.c)."""
import numpy as np

ENTRY_DTYPE = np.dtype([("id", np.uint32), ("value", np.float32)])


def _pack(rows_ids, rows_vals, target):
    sizes = np.array([len(r) for r in rows_ids], dtype=np.uint64)
    row_ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    ent = np.zeros(int(row_ptr[-1]), dtype=ENTRY_DTYPE)
    if len(ent):
        ent["id"] = np.concatenate([np.asarray(r, dtype=np.uint32) for r in rows_ids if len(r)])
        ent["value"] = np.concatenate([np.asarray(r, dtype=np.float32) for r in rows_vals if len(r)])
    return ent, row_ptr, np.asarray(target, dtype=np.float32)


def movielens_shaped(n_users, n_items, n_rows, seed, k_true=4, noise=0.3):
    """user one-hot + item one-hot (2 nnz/row, values 1.0), integer ratings 1..5 from a planted FM."""
    rng = np.random.default_rng(seed)
    u = rng.integers(0, n_users, n_rows)
    it = rng.integers(0, n_items, n_rows)
    bu, bi = rng.normal(0, 0.5, n_users), rng.normal(0, 0.5, n_items)
    pu, qi = rng.normal(0, 0.5, (n_users, k_true)), rng.normal(0, 0.5, (n_items, k_true))
    y = 3.5 + bu[u] + bi[it] + (pu[u] * qi[it]).sum(1) + rng.normal(0, noise, n_rows)
    y = np.clip(np.rint(y), 1, 5)
    ids = [[int(a), int(n_users + b)] for a, b in zip(u, it)]
    vals = [[1.0, 1.0]] * n_rows
    return _pack(ids, vals, y)


def ragged_real(n_features, n_rows, max_nnz, seed, classification=True, empty_every=0, duplicates=False):
    """ragged rows with real values; optionally an empty row every `empty_every` rows and repeated ids."""
    rng = np.random.default_rng(seed)
    wtrue = rng.normal(0, 1, n_features)
    ids, vals, y = [], [], []
    for r in range(n_rows):
        z = int(rng.integers(1, max_nnz + 1))
        if empty_every and r % empty_every == empty_every - 1:
            z = 0
        if duplicates and z >= 2:
            rid = rng.integers(0, n_features, z)
            rid[1] = rid[0]                       # a repeated id inside the row
        else:
            rid = rng.choice(n_features, size=z, replace=False) if z else np.zeros(0, dtype=np.int64)
        rv = np.round(rng.uniform(-1.5, 1.5, z), 3)
        rv[rv == 0] = 0.5
        s = float((wtrue[rid] * rv).sum()) + rng.normal(0, 0.2)
        ids.append([int(a) for a in rid])
        vals.append([float(a) for a in rv])
        y.append((1.0 if s > 0 else -1.0) if classification else round(s, 3))
    return _pack(ids, vals, y)


def onehot_fields(n_features, nnz, n_rows, seed, zipf=0.0, classification=True):
    """field-structured one-hot rows: field t owns ids [t*fs,(t+1)*fs); uniform or Zipf within field."""
    rng = np.random.default_rng(seed)
    fs = n_features // nnz
    if zipf > 0:
        p = 1.0 / np.arange(1, fs + 1) ** zipf
        p /= p.sum()
        off = rng.choice(fs, size=(n_rows, nnz), p=p)
    else:
        off = rng.integers(0, fs, (n_rows, nnz))
    idm = off + np.arange(nnz)[None, :] * fs
    ent = np.zeros(n_rows * nnz, dtype=ENTRY_DTYPE)
    ent["id"] = idm.reshape(-1).astype(np.uint32)
    ent["value"] = 1.0
    row_ptr = (np.arange(n_rows + 1, dtype=np.uint64) * np.uint64(nnz))
    wtrue = rng.normal(0, 1, n_features)
    s = wtrue[idm].sum(1) / np.sqrt(nnz) + rng.normal(0, 0.3, n_rows)
    y = np.where(s > 0, 1.0, -1.0) if classification else np.round(s, 3)
    return ent, row_ptr, y.astype(np.float32)

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
        z = min(int(rng.integers(1, max_nnz + 1)), n_features)      # (distinct ids: a row cannot be longer than the feature space)
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


def block_structured(n_users, n_items, n_rows, seed, n_ctx=7, k_true=3, noise=0.2, classification=False):
    """Block-structured ("relational", Rendle VLDB'13) rating data: a main table with one context attribute per row
    (one-hot of n_ctx) and two relation blocks, USERS (user one-hot + 2 real attributes) and ITEMS (item one-hot +
    3 binary genre flags, ragged).  Returns (main rows, [block rows...], [row mapping...]); block attribute ids are
    local to the block.  The flat design matrix is expand_blocks(...)."""
    rng = np.random.default_rng(seed)
    u, it, cx = rng.integers(0, n_users, n_rows), rng.integers(0, n_items, n_rows), rng.integers(0, n_ctx, n_rows)
    uattr = rng.normal(0, 1, (n_users, 2)).astype(np.float32)
    genre = rng.random((n_items, 3)) < 0.4
    users = _pack([[a, n_users, n_users + 1] for a in range(n_users)],
                  [[1.0, float(uattr[a, 0]), float(uattr[a, 1])] for a in range(n_users)], np.zeros(n_users))
    items = _pack([[b] + [n_items + g for g in range(3) if genre[b, g]] for b in range(n_items)],
                  [[1.0] + [1.0 for g in range(3) if genre[b, g]] for b in range(n_items)], np.zeros(n_items))
    bu, bi, bc = rng.normal(0, 0.5, n_users), rng.normal(0, 0.5, n_items), rng.normal(0, 0.3, n_ctx)
    pu, qi = rng.normal(0, 0.5, (n_users, k_true)), rng.normal(0, 0.5, (n_items, k_true))
    wg = rng.normal(0, 0.4, 3)
    y = 3.5 + bu[u] + bi[it] + bc[cx] + (pu[u] * qi[it]).sum(1) + genre[it] @ wg + 0.3 * uattr[u, 0] + rng.normal(0, noise, n_rows)
    y = np.where(y > 3.5, 1.0, -1.0) if classification else np.clip(np.rint(y), 1, 5)
    main = _pack([[int(c)] for c in cx], [[1.0]] * n_rows, y)
    blocks = [(users[0], users[1], n_users + 2), (items[0], items[1], n_items + 3)]      # (entries, row_ptr, num_feature)
    return main, blocks, [u.astype(np.uint32), it.astype(np.uint32)]


def expand_blocks(main_entries, main_row_ptr, blocks, maps, num_main_attr):
    """the flat rows a block-structured data set stands for: main entries, then each block's mapped row with its ids
    shifted by the block's attr_offset (libfm.cpp:213-216).  Returns (entries, row_ptr, attr_offsets)."""
    offs, o = [], num_main_attr
    for _, _, nf in blocks:
        offs.append(o)
        o += nf
    mrp = np.asarray(main_row_ptr, dtype=np.int64)
    ids, vals = [], []
    for c in range(len(mrp) - 1):
        ri, rv = list(main_entries["id"][mrp[c]:mrp[c + 1]]), list(main_entries["value"][mrp[c]:mrp[c + 1]])
        for (be, bp, _), mp, off in zip(blocks, maps, offs):
            bp = np.asarray(bp, dtype=np.int64)
            a, b = bp[mp[c]], bp[mp[c] + 1]
            ri += list(be["id"][a:b].astype(np.int64) + off)
            rv += list(be["value"][a:b])
        ids.append(ri)
        vals.append(rv)
    ent, rp, _ = _pack(ids, vals, np.zeros(len(ids)))
    return ent, rp, offs


def criteo_shaped(n_rows, seed, n_dense=13, dense_ids=100, n_cat=26, cat_ids=5000, zipf=1.05, classification=True):
    """BASELINE configs[2] / SURVEY section 8d in small: n_dense "numeric" fields binned into <= dense_ids ids with a geometric
    profile (bin = floor(12 Exp(1)): the first bin holds 8 % of the rows) + n_cat categorical fields Zipf(zipf) over cat_ids ids
    each (the head id of a field is met by ~12 % of the rows at cat_ids = 5000), values 1.0; labels from a planted FM with a
    click-through-like 1:3.5 imbalance.  Returns (entries, row_ptr, target, n)."""
    rng = np.random.default_rng(seed)
    z = n_dense + n_cat
    offs = np.concatenate([[0], np.cumsum([dense_ids] * n_dense + [cat_ids] * n_cat)])
    n = int(offs[-1])
    p = 1.0 / np.arange(1, cat_ids + 1) ** zipf
    p /= p.sum()
    cols = []
    for t in range(z):
        if t < n_dense:
            c = np.minimum(rng.exponential(12.0, n_rows).astype(np.int64), dense_ids - 1)
        else:
            c = rng.choice(cat_ids, size=n_rows, p=p)
        cols.append(c + offs[t])
    idm = np.stack(cols, 1)
    wtrue = rng.normal(0, 0.6, n)
    vtrue = rng.normal(0, 0.25, (n, 4))
    sv = vtrue[idm]
    s = wtrue[idm].sum(1) / np.sqrt(z) + 0.5 * ((sv.sum(1) ** 2).sum(1) - (sv ** 2).sum((1, 2))) / z
    s = s - np.median(s) - 0.8
    if classification:
        y = np.where(rng.random(n_rows) < 1.0 / (1.0 + np.exp(-2.0 * s)), 1.0, -1.0)
    else:
        y = np.round(s + rng.normal(0, 0.3, n_rows), 3)
    ent = np.zeros(n_rows * z, dtype=ENTRY_DTYPE)
    ent["id"] = idm.reshape(-1).astype(np.uint32)
    ent["value"] = 1.0
    row_ptr = np.arange(n_rows + 1, dtype=np.uint64) * np.uint64(z)
    return ent, row_ptr, y.astype(np.float32), n


def collision_mass(entries, n_rows, n):
    """C of fmx_sgd_opts::batch: the mean over pairs of DIFFERENT rows of sum_j |x_ej| |x_e'j| (the expected number of features two
    rows share, value-weighted) = (sum_j (sum_rows |x_j|)^2 - sum_entries x^2) / (N (N - 1))"""
    x = np.abs(entries["value"].astype(np.float64))
    cnt = np.bincount(entries["id"], weights=x, minlength=n)
    return float(max(0.0, (cnt ** 2).sum() - (x ** 2).sum()) / (n_rows * (n_rows - 1.0)))


def stable_batch(lr, task, C, default=262144, curv_scale=1.0):
    """the library's choice for fmx_sgd_opts::batch = 0 (fmx_core.hip resolve_batch): the default, cut to the largest power of two
    with lr * curvature * batch * C <= 1"""
    per_row = lr * (1.0 if task == 0 else 0.25) * curv_scale * C
    if per_row <= 0 or default * per_row <= 1.0:
        return default
    b = 1
    while (b * 2) * per_row <= 1.0 and b * 2 <= default:
        b *= 2
    return max(b, 32)

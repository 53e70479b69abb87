"""Readers / writers of libFM's data formats (host side, numpy).  They produce exactly the buffers the C-ABI takes.

  text   : libsvm lines "target id:value id:value ..."; blank lines and lines starting with '#' are skipped,
           leading blanks/tabs allowed                                (Data::load, src/libfm/src/Data.h:180-285)
  binary : <prefix>.x = file_header {u32 id = 2; u32 float_size = 4; u64 num_values; u32 num_rows; u32 num_cols}
           (24 bytes with the compiler's padding) then per row {u32 size; sparse_entry<float>[size]}
                                                                      (src/util/fmatrix.h:44-50, 124-143)
           <prefix>.y = {u32 file_version = 1; u32 data_size = 4; u32 num_rows} then float32[num_rows]
                                                                      (src/util/matrix.h:344-358)
           <prefix>.xt = the transpose in the same .x format          (tools/transpose.cpp)
  Data::load picks binary when <name>.x (and <name>.y) exist, else text (Data.h:119-125); load() does the same.
"""
import os

import numpy as np

from .capi import ENTRY_DTYPE

FMATRIX_FILE_ID = 2          # fmatrix.h:32
DVECTOR_FILE_ID = 1          # matrix.h:32
_HDR = np.dtype([("id", "<u4"), ("float_size", "<u4"), ("num_values", "<u8"), ("num_rows", "<u4"), ("num_cols", "<u4")])


def read_libsvm(path):
    """Data::load's text branch through the native reader of the C-ABI (fmx_read_libsvm: same accepted tokens, same
    rejected lines, the reference's error texts).  Returns (entries, row_ptr, target)."""
    import ctypes as C
    from . import capi
    lib = capi.load()
    rows, err = capi.HostRows(), C.create_string_buffer(512)
    rc = lib.fmx_read_libsvm(os.fsencode(path), C.byref(rows), err, len(err))
    if rc != capi.FMX_OK:
        msg = err.value.decode(errors="replace")
        raise (OSError if msg.startswith("unable to open") else ValueError)(msg)
    try:
        ent = np.ctypeslib.as_array(C.cast(rows.entries, C.POINTER(C.c_uint64)), (max(rows.nnz, 1),))[:rows.nnz].view(ENTRY_DTYPE).copy()
        row_ptr = np.ctypeslib.as_array(C.cast(rows.row_ptr, C.POINTER(C.c_uint64)), (rows.n_rows + 1,)).copy()
        y = np.ctypeslib.as_array(C.cast(rows.target, C.POINTER(C.c_float)), (max(rows.n_rows, 1),))[:rows.n_rows].copy()
    finally:
        lib.fmx_free_host_rows(C.byref(rows))
    return ent, row_ptr, y


def _host_rows_to_numpy(lib, rows):
    import ctypes as C
    try:
        ent = np.ctypeslib.as_array(C.cast(rows.entries, C.POINTER(C.c_uint64)), (max(rows.nnz, 1),))[:rows.nnz].view(ENTRY_DTYPE).copy()
        row_ptr = np.ctypeslib.as_array(C.cast(rows.row_ptr, C.POINTER(C.c_uint64)), (rows.n_rows + 1,)).copy()
        y = np.ctypeslib.as_array(C.cast(rows.target, C.POINTER(C.c_float)), (max(rows.n_rows, 1),))[:rows.n_rows].copy()
        return ent, row_ptr, y, int(rows.num_feature)
    finally:
        lib.fmx_free_host_rows(C.byref(rows))


def read_binary(prefix):
    """Data::load's binary branch through the C-ABI (fmx_read_binary): <prefix>.x + .y, or <prefix>.xt + .y (the rows are
    rebuilt from the transpose), or the older .data / .datat / .target.  Returns (entries, row_ptr, target, num_feature)."""
    import ctypes as C
    from . import capi
    lib = capi.load()
    rows, err = capi.HostRows(), C.create_string_buffer(512)
    rc = lib.fmx_read_binary(os.fsencode(prefix), C.byref(rows), err, len(err))
    if rc != capi.FMX_OK:
        raise ValueError(err.value.decode(errors="replace"))
    return _host_rows_to_numpy(lib, rows)


def read_libsvm_py(path):
    """the same format in pure Python (slow; kept as an independent cross-check of the native reader in the tests)"""
    ids, vals, sizes, ys = [], [], [], []
    with open(path) as f:
        for line in f:
            line = line.lstrip(" \t").rstrip("\r\n")
            if not line or line[0] == "#":
                continue
            toks = line.split()
            ys.append(float(toks[0]))
            n = 0
            for t in toks[1:]:
                if t[0] == "#":
                    break
                a, b = t.split(":")
                ids.append(int(a))
                vals.append(float(b))
                n += 1
            sizes.append(n)
    ent = np.zeros(len(ids), dtype=ENTRY_DTYPE)
    ent["id"] = np.asarray(ids, dtype=np.uint32)
    ent["value"] = np.asarray(vals, dtype=np.float32)
    row_ptr = np.concatenate([[0], np.cumsum(np.asarray(sizes, dtype=np.uint64))]).astype(np.uint64)
    return ent, row_ptr, np.asarray(ys, dtype=np.float32)


def read_binary_x(path):
    raw = np.fromfile(path, dtype=np.uint8)
    hdr = raw[:_HDR.itemsize].view(_HDR)[0]
    if int(hdr["id"]) != FMATRIX_FILE_ID or int(hdr["float_size"]) != 4:
        raise ValueError("%s: not a libFM binary matrix (id %d, float_size %d)" % (path, hdr["id"], hdr["float_size"]))
    n_rows, nnz = int(hdr["num_rows"]), int(hdr["num_values"])
    body = raw[_HDR.itemsize:]
    if len(body) != 4 * n_rows + 8 * nnz:
        raise ValueError("%s: size does not match its header" % path)
    words = body.view("<u4")
    ent = np.zeros(nnz, dtype=ENTRY_DTYPE)
    row_ptr = np.zeros(n_rows + 1, dtype=np.uint64)
    pos, out = 0, 0
    for r in range(n_rows):                      # rows are variable length: walk the size words
        sz = int(words[pos]); pos += 1
        ent[out:out + sz] = words[pos:pos + 2 * sz].view(ENTRY_DTYPE)
        pos += 2 * sz; out += sz
        row_ptr[r + 1] = out
    return ent, row_ptr, int(hdr["num_cols"])


def read_binary_y(path):
    raw = np.fromfile(path, dtype=np.uint8)
    ver, size, n = raw[:12].view("<u4")
    if int(ver) != DVECTOR_FILE_ID or int(size) != 4:
        raise ValueError("%s: not a libFM binary float vector" % path)
    return raw[12:12 + 4 * int(n)].view("<f4").copy()


def write_binary_matrix(path, entries, row_ptr, num_cols=None):
    """LargeSparseMatrix::saveToBinaryFile (fmatrix.h:121-140): header + per row {u32 size, entries}"""
    entries = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    n_rows = len(row_ptr) - 1
    hdr = np.zeros(1, dtype=_HDR)
    hdr["id"], hdr["float_size"], hdr["num_values"], hdr["num_rows"] = FMATRIX_FILE_ID, 4, len(entries), n_rows
    hdr["num_cols"] = num_cols if num_cols is not None else (int(entries["id"].max()) + 1 if len(entries) else 0)
    with open(path, "wb") as f:
        f.write(hdr.tobytes())
        for r in range(n_rows):
            a, b = row_ptr[r], row_ptr[r + 1]
            f.write(np.uint32(b - a).tobytes())
            f.write(entries[a:b].tobytes())


def transpose(entries, row_ptr, num_cols):
    """X -> X^T in the same CSR-with-AoS layout (tools/transpose.cpp; Data::create_data_t, Data.h:292-341): per
    column the {row id, value} pairs in ascending row order."""
    entries = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    rows = np.repeat(np.arange(len(row_ptr) - 1, dtype=np.uint32), np.diff(row_ptr))
    order = np.argsort(entries["id"], kind="stable")
    t = np.zeros(len(entries), dtype=ENTRY_DTYPE)
    t["id"], t["value"] = rows[order], entries["value"][order]
    col_ptr = np.concatenate([[0], np.cumsum(np.bincount(entries["id"], minlength=num_cols))]).astype(np.uint64)
    return t, col_ptr


class Relation:
    """RelationData (relation.h:32-51): one block of a block-structured data set -- its own design matrix
    (<prefix>.x, or <prefix>.xt transposed back: for als/mcmc the reference loads only .xt, libfm.cpp:181-185) and
    optional attribute groups (<prefix>.groups)."""

    def __init__(self, entries, row_ptr, num_feature, groups=None):
        self.entries = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        self.num_cases = len(self.row_ptr) - 1
        self.num_feature = int(num_feature)
        self.groups = groups                                     # None = one group


def read_relation(prefix):
    if os.path.exists(prefix + ".x"):
        ent, row_ptr, ncols = read_binary_x(prefix + ".x")
    elif os.path.exists(prefix + ".xt"):
        ent_t, col_ptr, n_rows = read_binary_x(prefix + ".xt")        # rows of .xt = attributes, columns = cases
        ent, row_ptr = transpose(ent_t, col_ptr, n_rows)
        ncols = len(col_ptr) - 1
    else:
        raise OSError("could not open " + prefix + ".x / .xt")
    groups = None
    if os.path.exists(prefix + ".groups"):
        with open(prefix + ".groups") as f:
            vals = [int(x) for x in f.read().split()[:ncols]]
        groups = np.zeros(ncols, dtype=np.uint32)
        groups[:len(vals)] = vals
    return Relation(ent, row_ptr, ncols, groups)


def read_row_mapping(path, expected_rows):
    """RelationJoin::load (relation.h:125-150): binary DVector<uint> or one uint per line"""
    raw = np.fromfile(path, dtype=np.uint8)
    if len(raw) >= 12:
        ver, size, n = raw[:12].view("<u4")
        if int(ver) == DVECTOR_FILE_ID and int(size) == 4:
            m = raw[12:12 + 4 * int(n)].view("<u4").copy()
            if len(m) != expected_rows:
                raise ValueError("%s: %d rows, expected %d" % (path, len(m), expected_rows))
            return m
    with open(path) as f:
        vals = [int(x) for x in f.read().split()[:expected_rows]]
    m = np.zeros(expected_rows, dtype=np.uint32)
    m[:len(vals)] = vals
    return m


def write_binary(prefix, entries, row_ptr, target, num_cols=None):
    """what tools/convert.cpp writes: <prefix>.x and <prefix>.y"""
    write_binary_matrix(prefix + ".x", entries, row_ptr, num_cols)
    n_rows = len(row_ptr) - 1
    with open(prefix + ".y", "wb") as f:
        f.write(np.array([DVECTOR_FILE_ID, 4, n_rows], dtype="<u4").tobytes())
        f.write(np.ascontiguousarray(target, dtype="<f4").tobytes())


def load(name):
    """Data::load auto-detection (Data.h:113-125): binary <name>.x/.y if present, else libsvm text <name>."""
    if os.path.exists(name + ".y") and (os.path.exists(name + ".x") or os.path.exists(name + ".xt")):
        ent, row_ptr, y, _ = read_binary(name)                       # native reader of the C-ABI
        return ent, row_ptr, y
    return read_libsvm(name)

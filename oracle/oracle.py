"""ctypes binding of oracle/libfm_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (libfm_amd) never does; it fails loudly when its HIP library is missing.

The functions mirror oracle/fm_oracle.h (which cites the reference file:line for each).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfm_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")
REF_HARNESS = os.path.join(REF_DIR, "ref_harness")
REF_LIBFM = os.path.join(REF_DIR, "libFM")

ENTRY_DTYPE = np.dtype([("id", np.uint32), ("value", np.float32)])  # sparse_entry<float>, fmatrix.h:34-37

TASK_REGRESSION = 0
TASK_CLASSIFICATION = 1


def build(force=False):
    """(Re)build the oracle .so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < max(os.path.getmtime(os.path.join(HERE, f))
                                             for f in ("fm_oracle.c", "fm_oracle_als.c", "fm_oracle.h")):
        subprocess.check_call(["make", "-s", "-C", HERE, os.path.join(HERE, "libfm_oracle.so")])
    if os.path.exists("/root/reference/src/libfm/libfm.cpp"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class _Model(C.Structure):
    _fields_ = [("n", C.c_uint64), ("k", C.c_int32), ("k0", C.c_int32), ("k1", C.c_int32),
                ("w0", C.c_double), ("w", C.c_void_p), ("v", C.c_void_p),
                ("reg0", C.c_double), ("regw", C.c_double), ("regv", C.c_double)]


class _Data(C.Structure):
    _fields_ = [("entries", C.c_void_p), ("row_ptr", C.c_void_p), ("target", C.c_void_p),
                ("n_rows", C.c_uint32)]


class _SgdaState(C.Structure):
    _fields_ = [("reg_w", C.c_void_p), ("reg_v", C.c_void_p), ("grad_w", C.c_void_p), ("grad_v", C.c_void_p),
                ("val_pos", C.c_uint32), ("num_groups", C.c_uint32), ("group", C.c_void_p)]


class _AlsReg(C.Structure):
    _fields_ = [("group", C.c_void_p), ("num_groups", C.c_uint32), ("w_lambda", C.c_void_p), ("v_lambda", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.fmo_predict_raw.argtypes = [C.POINTER(_Model), C.POINTER(_Data), C.c_void_p]
        L.fmo_predict_out.argtypes = [C.POINTER(_Model), C.POINTER(_Data), C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.fmo_evaluate.argtypes = [C.POINTER(_Model), C.POINTER(_Data), C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.fmo_evaluate.restype = C.c_double
        L.fmo_sgd_epoch_online.argtypes = [C.POINTER(_Model), C.POINTER(_Data), C.c_int, C.c_double, C.c_double, C.c_double]
        L.fmo_sgd_epoch_minibatch.argtypes = [C.POINTER(_Model), C.POINTER(_Data), C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32]
        L.fmo_sgd_epoch_minibatch_ex.argtypes = [C.POINTER(_Model), C.POINTER(_Data), C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_int]
        L.fmo_sgd_epoch_minibatch_pipelined.argtypes = L.fmo_sgd_epoch_minibatch_ex.argtypes
        L.fmo_sgd_epoch_minibatch_hot.argtypes = L.fmo_sgd_epoch_minibatch_ex.argtypes + [C.c_void_p]
        L.fmo_sgd_epoch_twolevel.argtypes = [C.POINTER(_Model), C.POINTER(_Data), C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32,
                                             C.c_uint32, C.c_int, C.c_void_p]
        L.fmo_multiplier.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
        L.fmo_multiplier.restype = C.c_double
        L.fmo_synth_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fmo_synth_id.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
        L.fmo_synth_id.restype = C.c_uint32
        L.fmo_init_value.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_double]
        L.fmo_init_value.restype = C.c_double
        L.fmo_als_learn.argtypes = [C.POINTER(_Model), C.POINTER(_Data), C.POINTER(_Data), C.c_int, C.c_int, C.c_double, C.c_double,
                                    C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.fmo_als_learn_groups.argtypes = [C.POINTER(_Model), C.POINTER(_Data), C.POINTER(_Data), C.c_int, C.c_int, C.POINTER(_AlsReg),
                                           C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.fmo_sgda_epoch.argtypes = [C.POINTER(_Model), C.POINTER(_SgdaState), C.POINTER(_Data), C.POINTER(_Data), C.c_int,
                                     C.c_double, C.c_double, C.c_double, C.c_int]
        L.fmo_sgda_epoch_minibatch.argtypes = [C.POINTER(_Model), C.POINTER(_SgdaState), C.POINTER(_Data), C.POINTER(_Data), C.c_int,
                                               C.c_double, C.c_double, C.c_double, C.c_int, C.c_uint32, C.c_uint32]
        L.fmo_fill_params.argtypes = [C.POINTER(_Model), C.c_uint64, C.c_double, C.c_int]
        L.fmo_time_sgd_synth.argtypes = [C.POINTER(_Model), C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_double]
        L.fmo_time_sgd_synth.restype = C.c_double
        _lib = L
    return _lib


class Model:
    """Reference-layout FM parameters: w0, w[n], v[k][n] factor-major fp64 (fm_model.h:46-48)."""

    def __init__(self, n, k, k0=True, k1=True, reg0=0.0, regw=0.0, regv=0.0):
        self.n, self.k, self.k0, self.k1 = int(n), int(k), bool(k0), bool(k1)
        self.reg0, self.regw, self.regv = float(reg0), float(regw), float(regv)
        self.w0 = 0.0
        self.w = np.zeros(self.n, dtype=np.float64)
        self.v = np.zeros((self.k, self.n), dtype=np.float64)

    def copy(self):
        m = Model(self.n, self.k, self.k0, self.k1, self.reg0, self.regw, self.regv)
        m.w0 = self.w0
        m.w[:] = self.w
        m.v[:] = self.v
        return m

    def _c(self):
        assert self.w.flags.c_contiguous and self.v.flags.c_contiguous
        return _Model(self.n, self.k, int(self.k0), int(self.k1), self.w0,
                      self.w.ctypes.data, self.v.ctypes.data, self.reg0, self.regw, self.regv)

    @staticmethod
    def from_dump(path, **kw):
        """Read a ref_harness parameter dump (magic FMXP, u64 n, i32 k, f64 w0, w[n], v[k][n])."""
        with open(path, "rb") as f:
            raw = f.read()
        assert raw[:4] == b"FMXP", path
        n = int(np.frombuffer(raw, dtype=np.uint64, count=1, offset=4)[0])
        k = int(np.frombuffer(raw, dtype=np.int32, count=1, offset=12)[0])
        m = Model(n, k, **kw)
        vals = np.frombuffer(raw, dtype=np.float64, offset=16)
        m.w0 = float(vals[0])
        m.w[:] = vals[1:1 + n]
        m.v[:] = vals[1 + n:1 + n + k * n].reshape(k, n)
        return m


class Data:
    """CSR with the reference's AoS entries: {u32 id; f32 value}[nnz], row_ptr u64[n_rows+1], target f32."""

    def __init__(self, entries, row_ptr, target):
        self.entries = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        self.target = np.ascontiguousarray(target, dtype=np.float32)
        self.n_rows = len(self.target)
        assert len(self.row_ptr) == self.n_rows + 1

    @property
    def row_sizes(self):
        return np.diff(self.row_ptr.astype(np.int64)).astype(np.uint32)

    @property
    def num_feature(self):
        return int(self.entries["id"].max()) + 1 if len(self.entries) else 0

    def _c(self):
        return _Data(self.entries.ctypes.data, self.row_ptr.ctypes.data, self.target.ctypes.data, self.n_rows)

    @staticmethod
    def from_rows(rows, target):
        """rows: list of lists of (id, value)."""
        sizes = np.array([len(r) for r in rows], dtype=np.uint64)
        row_ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        ent = np.zeros(int(row_ptr[-1]), dtype=ENTRY_DTYPE)
        p = 0
        for r in rows:
            for (i, v) in r:
                ent[p] = (i, v)
                p += 1
        return Data(ent, row_ptr, target)

    def write_libsvm(self, path):
        """libsvm text the reference parser accepts (Data.h:192-285)."""
        with open(path, "w") as f:
            for r in range(self.n_rows):
                a, b = int(self.row_ptr[r]), int(self.row_ptr[r + 1])
                toks = ["%.9g" % self.target[r]]
                toks += ["%d:%.9g" % (int(e["id"]), float(e["value"])) for e in self.entries[a:b]]
                f.write(" ".join(toks) + "\n")

    @staticmethod
    def read_libsvm(path):
        rows, ys = [], []
        with open(path) as f:
            for line in f:
                line = line.strip()
                if not line or line.startswith("#"):
                    continue
                toks = line.split()
                ys.append(np.float32(toks[0]))
                rows.append([(int(t.split(":")[0]), np.float32(t.split(":")[1])) for t in toks[1:]])
        return Data.from_rows(rows, np.array(ys, dtype=np.float32))


def predict_raw(m, d):
    out = np.zeros(d.n_rows, dtype=np.float64)
    cm, cd = m._c(), d._c()
    lib().fmo_predict_raw(C.byref(cm), C.byref(cd), out.ctypes.data)
    return out


def predict_out(m, d, task, min_target, max_target):
    out = np.zeros(d.n_rows, dtype=np.float64)
    cm, cd = m._c(), d._c()
    lib().fmo_predict_out(C.byref(cm), C.byref(cd), task, min_target, max_target, out.ctypes.data)
    return out


def evaluate(m, d, task, min_target, max_target):
    mae = C.c_double(0)
    cm, cd = m._c(), d._c()
    r = lib().fmo_evaluate(C.byref(cm), C.byref(cd), task, min_target, max_target, C.byref(mae))
    return r, mae.value


def sgd_epoch_online(m, d, task, lr, min_target, max_target):
    cm, cd = m._c(), d._c()
    lib().fmo_sgd_epoch_online(C.byref(cm), C.byref(cd), task, lr, min_target, max_target)
    m.w0 = cm.w0


def sgd_epoch_minibatch(m, d, task, lr, min_target, max_target, batch, w0_chunk, bias_lag=False, pipelined=False, hot=None):
    """pipelined: the multi-GPU overlap schedule -- the sums of a batch are gathered one update early (fm_oracle.h).
    hot: bool / uint8 [n] flags of the features whose linear weight advances with the bias (fmo_sgd_epoch_minibatch_hot)."""
    cm, cd = m._c(), d._c()
    if hot is not None:
        assert not pipelined
        hot = np.ascontiguousarray(hot, dtype=np.uint8)
        assert hot.shape == (m.n,)
        lib().fmo_sgd_epoch_minibatch_hot(C.byref(cm), C.byref(cd), task, lr, min_target, max_target, batch, w0_chunk, int(bias_lag),
                                          hot.ctypes.data)
    else:
        fn = lib().fmo_sgd_epoch_minibatch_pipelined if pipelined else lib().fmo_sgd_epoch_minibatch_ex
        fn(C.byref(cm), C.byref(cd), task, lr, min_target, max_target, batch, w0_chunk, int(bias_lag))
    m.w0 = cm.w0


def sgd_epoch_twolevel(m, d, task, lr, min_target, max_target, batch, window, w0_chunk, bias_lag, hot):
    """the two-level batch rule (fm_oracle.h fmo_sgd_epoch_twolevel): hot features (hot[j] != 0) frozen per `window` rows, cold ones
    per `batch` rows; bias lag counted in windows.  hot=None: no hot feature."""
    cm, cd = m._c(), d._c()
    if hot is not None:
        hot = np.ascontiguousarray(hot, dtype=np.uint8)
        assert hot.shape == (m.n,)
    lib().fmo_sgd_epoch_twolevel(C.byref(cm), C.byref(cd), task, lr, min_target, max_target, int(batch), int(window), int(w0_chunk),
                                 int(bias_lag), hot.ctypes.data if hot is not None else None)
    m.w0 = cm.w0


def hot_features(d, n, batch, hot_count):
    """a hot set for fmo_sgd_epoch_minibatch_hot (an instrument of DESIGN.md section 3a, no product mode): features that occur at least hot_count
    times per batch on average, i.e. count_j * batch >= hot_count * n_rows (batch clipped to the data set)."""
    cnt = np.bincount(d.entries["id"], minlength=n).astype(np.int64)
    b = min(int(batch), d.n_rows) if batch else d.n_rows
    return (cnt * b >= int(hot_count) * d.n_rows) & (cnt > 0)


def synth_rows(seed, row0, n_rows, nnz, n):
    ent = np.zeros(n_rows * nnz, dtype=ENTRY_DTYPE)
    rp = np.zeros(n_rows + 1, dtype=np.uint64)
    y = np.zeros(n_rows, dtype=np.float32)
    lib().fmo_synth_rows(seed, row0, n_rows, nnz, n, ent.ctypes.data, rp.ctypes.data, y.ctypes.data)
    return Data(ent, rp, y)


def init_values(seed, n, k, stdev):
    """fmo_init_value for every (j, f), vectorised in numpy (bit-identical to the C function)."""
    j = np.arange(n, dtype=np.uint64)[None, :]
    f = np.arange(k, dtype=np.uint64)[:, None]
    with np.errstate(over="ignore"):
        x = np.uint64(seed) ^ (j * np.uint64(0x9E3779B97F4A7C15) + f * np.uint64(0xD6E8FEB86659FD93) + np.uint64(0x1234567))
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return stdev * (2.0 * u - 1.0) * 1.7320508075688772


def init_values_ids(seed, ids, k, stdev):
    """fmo_init_value(seed, j, f, stdev) for the features `ids` only ([k][len(ids)]): the start values of a SUB-model of a table
    too large for the host (what fmx_init_params leaves in the rows `ids`, before their rounding to fp32)."""
    j = np.asarray(ids, dtype=np.uint64)[None, :]
    f = np.arange(k, dtype=np.uint64)[:, None]
    with np.errstate(over="ignore"):
        x = np.uint64(seed) ^ (j * np.uint64(0x9E3779B97F4A7C15) + f * np.uint64(0xD6E8FEB86659FD93) + np.uint64(0x1234567))
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return stdev * (2.0 * u - 1.0) * 1.7320508075688772


def run_ref_harness(args, cwd=None, env=None):
    """Run oracle/_ref/ref_harness (the real reference classes).  Raises if it is not built.
    env: extra environment (FMX_META, FMX_GROUP_REG -- see the header of ref_harness.cpp)."""
    if not os.path.exists(REF_HARNESS):
        raise FileNotFoundError("oracle/_ref/ref_harness not built (needs /root/reference); run make -C oracle")
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([REF_HARNESS] + [str(a) for a in args], cwd=cwd, check=True,
                          capture_output=True, text=True, env=e)


def time_sgd_synth(n, k, nnz, n_rows, seed=123, stdev=0.01, lr=0.01, regv=0.001, threads=8, row0=0):
    """cpu_baseline leg of bench.py: the restated reference loop (1 thread) on a bounded sample of the synthetic
    workload.  Parameters are np.empty + a threaded fill (untimed).  Returns (seconds, examples_per_sec)."""
    m = Model.__new__(Model)
    m.n, m.k, m.k0, m.k1 = int(n), int(k), True, True
    m.reg0, m.regw, m.regv = 0.0, 0.0, float(regv)
    m.w0 = 0.0
    m.w = np.empty(m.n, dtype=np.float64)
    m.v = np.empty((m.k, m.n), dtype=np.float64)
    cm = m._c()
    lib().fmo_fill_params(C.byref(cm), seed, stdev, threads)
    sec = lib().fmo_time_sgd_synth(C.byref(cm), seed, row0, n_rows, nnz, TASK_CLASSIFICATION, lr)
    return sec, n_rows / sec


def als_learn(m, train, test, task, num_iter, w_lambda, v_lambda, min_target, max_target):
    """fm_learn_mcmc with do_sample = 0 (ALS).  Returns (pred_this on test of the last iteration, train metric per iteration)."""
    pred = np.zeros(test.n_rows, dtype=np.float64)
    metric = np.zeros(num_iter, dtype=np.float64)
    cm, ctr, cte = m._c(), train._c(), test._c()
    lib().fmo_als_learn(C.byref(cm), C.byref(ctr), C.byref(cte), task, num_iter, w_lambda, v_lambda, min_target, max_target,
                        pred.ctypes.data, metric.ctypes.data)
    m.w0 = cm.w0
    return pred, metric


def als_learn_groups(m, train, test, task, num_iter, group, w_lambda_g, v_lambda_gf, min_target, max_target):
    """ALS with regularisation per attribute group: group[n] (uint32), w_lambda_g[G], v_lambda_gf[G][k]."""
    group = np.ascontiguousarray(group, dtype=np.uint32)
    wl = np.ascontiguousarray(w_lambda_g, dtype=np.float64)
    vl = np.ascontiguousarray(v_lambda_gf, dtype=np.float64)
    assert group.shape == (m.n,) and vl.shape == (len(wl), max(m.k, 0)) and int(group.max()) < len(wl)
    pred = np.zeros(test.n_rows, dtype=np.float64)
    metric = np.zeros(num_iter, dtype=np.float64)
    cm, ctr, cte = m._c(), train._c(), test._c()
    reg = _AlsReg(group.ctypes.data, len(wl), wl.ctypes.data, vl.ctypes.data)
    lib().fmo_als_learn_groups(C.byref(cm), C.byref(ctr), C.byref(cte), task, num_iter, C.byref(reg), min_target, max_target,
                               pred.ctypes.data, metric.ctypes.data)
    m.w0 = cm.w0
    return pred, metric


class SgdaState:
    """reg_w[G], reg_v[G][k] and the shadow gradients of fm_learn_sgd_element_adapt_reg (G attribute groups)."""

    def __init__(self, n, k, group=None):
        self.group = None if group is None else np.ascontiguousarray(group, dtype=np.uint32)
        self.num_groups = 1 if group is None else int(self.group.max()) + 1
        self.reg_w = np.zeros(self.num_groups, dtype=np.float64)
        self.reg_v = np.zeros((self.num_groups, max(k, 1)), dtype=np.float64)
        self.grad_w = np.zeros(n, dtype=np.float64)
        self.grad_v = np.zeros((max(k, 1), n), dtype=np.float64)


def sgda_learn(m, train, val, task, lr, min_target, max_target, num_iter, group=None, batch=None, w0_chunk=1):
    """fm_learn_sgd_element_adapt_reg::learn (:250-279): w := 0, regs := 0, then num_iter epochs (lambda steps from the 2nd).
    batch: None = the reference's online loop; an int = the batch restatement (fmo_sgda_epoch_minibatch)."""
    st = SgdaState(m.n, m.k, group)
    if m.k == 0:
        st.reg_v = np.zeros((st.num_groups, 0), dtype=np.float64)
    m.w[:] = 0.0
    for i in range(num_iter):
        cm, ctr, cv = m._c(), train._c(), val._c()
        cs = _SgdaState(st.reg_w.ctypes.data, st.reg_v.ctypes.data, st.grad_w.ctypes.data, st.grad_v.ctypes.data, 0,
                        st.num_groups, None if st.group is None else st.group.ctypes.data)
        if batch is None:
            lib().fmo_sgda_epoch(C.byref(cm), C.byref(cs), C.byref(ctr), C.byref(cv), task, lr, min_target, max_target, int(i > 0))
        else:
            lib().fmo_sgda_epoch_minibatch(C.byref(cm), C.byref(cs), C.byref(ctr), C.byref(cv), task, lr, min_target, max_target,
                                           int(i > 0), int(batch), int(w0_chunk))
        m.w0 = cm.w0
    return st

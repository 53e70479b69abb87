"""ctypes binding of the C-ABI in include/fmx.h (libfm_amd/libfmx.so).

This is the only way Python reaches the HIP kernels; there is no Python/CPU fallback: if the shared library is
missing or no HIP device is present, the calls raise."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMX_LIB", os.path.join(HERE, "libfmx.so"))   # FMX_LIB: experiment builds only

FMX_OK = 0
TASK_REGRESSION, TASK_CLASSIFICATION = 0, 1
SGD_SEQUENTIAL, SGD_MINIBATCH, SGD_HOGWILD = 0, 1, 2
APPLY_DEFAULT, APPLY_ATOMIC, APPLY_STORE, APPLY_SEGMENTED, APPLY_FUSED = 0, 1, 2, 3, 4
FLAG_TIME_MAIN_KERNEL = 1
FLAG_BIAS_LAG = 2
FLAG_PIPELINE = 4
FLAG_REJECT_UNSTABLE = 8
FLAG_EVENT_SYNC = 16
FLAG_KEEP_WSIDE = 32
EVAL_WSIDE = 1
STAT_BATCH_CUT, STAT_UNSTABLE = 1, 2
STAT_WARN = 3   # the two bits that say something about the RULE; the bits above are how the epoch ran (include/fmx.h, ABI 7)
STAT_SCAN_PIT, STAT_SCAN_SERIAL, STAT_SCAN_FALLBACK, STAT_EVENT_SYNC, STAT_HANDOFF_TIMEOUT, STAT_XCD_RESIDENT, STAT_SEQ_RUNS, STAT_SMALL_ONE = 4, 8, 16, 32, 64, 128, 256, 512
SYNTH_UNIFORM, SYNTH_CRITEO = 0, 1
BLOCKS_EXPAND, BLOCKS_KEEP = 0, 1
COMM_ID_BYTES = 128
MAX_SLOTS = 8

ENTRY_DTYPE = np.dtype([("id", np.uint32), ("value", np.float32)])   # sparse_entry<float>, fmatrix.h:34-37


class FmxError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libfmx error %d: %s" % (code, text))
        self.code = code
        self.text = text


class Config(C.Structure):
    _fields_ = [("num_attribute", C.c_uint64), ("num_factor", C.c_int32), ("k0", C.c_int32), ("k1", C.c_int32),
                ("task", C.c_int32), ("reg0", C.c_double), ("regw", C.c_double), ("regv", C.c_double),
                ("learn_rate", C.c_double), ("min_target", C.c_double), ("max_target", C.c_double),
                ("device", C.c_int32), ("shard_rank", C.c_int32), ("shard_world", C.c_int32), ("shard_hash", C.c_int32),
                ("place_candidates", C.c_int32), ("als_split_min", C.c_uint32), ("exchange_runs", C.c_uint32), ("exchange_algo", C.c_uint32)]


ALS_SPLIT_NEVER = 0xFFFFFFFF
ALS_SPLIT_MIN = 0           # default of Handle(als_split_min=None): 0 = the library's choice (tests set 1 / ALS_SPLIT_NEVER to force a form)


class SgdOpts(C.Structure):
    _fields_ = [("mode", C.c_int32), ("apply", C.c_int32), ("batch", C.c_uint32), ("w0_chunk", C.c_uint32),
                ("flags", C.c_uint32), ("bias_lag", C.c_uint32)]


class EpochStats(C.Structure):
    _fields_ = [("rows", C.c_uint64), ("batches", C.c_uint64), ("device_seconds", C.c_double),
                ("main_kernel_seconds", C.c_double), ("main_kernel_launches", C.c_uint64),
                ("max_feature_count", C.c_uint32), ("batch_used", C.c_uint32), ("deferred_features", C.c_uint64),
                ("collision_mass", C.c_double), ("batch_gain", C.c_double), ("status", C.c_uint32), ("w0_chunk_used", C.c_uint32),
                ("setup_seconds", C.c_double), ("phase_seconds", C.c_double * 4)]


class BatchInfo(C.Structure):
    _fields_ = [("collision_mass", C.c_double), ("batch_gain", C.c_double), ("batch", C.c_uint32), ("status", C.c_uint32)]


class PlaceInfo(C.Structure):
    _fields_ = [("method", C.c_int32), ("chunks", C.c_uint32), ("per_class", C.c_uint32 * 2), ("pool", C.c_uint32),
                ("classes_seen", C.c_uint32), ("seconds", C.c_double)]


class Eval(C.Structure):
    _fields_ = [("rmse", C.c_double), ("mae", C.c_double), ("accuracy", C.c_double),
                ("device_seconds", C.c_double), ("rows", C.c_uint64), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class Relation(C.Structure):
    _fields_ = [("entries", C.c_void_p), ("row_ptr", C.c_void_p), ("n_rows", C.c_uint32), ("reserved", C.c_uint32),
                ("nnz", C.c_uint64), ("data_row_to_relation_row", C.c_void_p), ("attr_offset", C.c_uint64)]


class HostRows(C.Structure):
    _fields_ = [("entries", C.c_void_p), ("row_ptr", C.c_void_p), ("target", C.c_void_p), ("n_rows", C.c_uint32),
                ("num_feature", C.c_uint32), ("nnz", C.c_uint64), ("min_target", C.c_float), ("max_target", C.c_float),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class AlsOpts(C.Structure):
    _fields_ = [("alpha", C.c_double), ("w_mu", C.c_double), ("w_lambda", C.c_double), ("v_mu", C.c_double),
                ("v_lambda", C.c_double), ("do_sample", C.c_int32), ("reserved", C.c_int32), ("seed", C.c_uint64),
                ("v_mu_f", C.c_void_p), ("v_lambda_f", C.c_void_p),
                ("num_groups", C.c_uint32), ("reserved2", C.c_uint32), ("w_mu_g", C.c_void_p), ("w_lambda_g", C.c_void_p),
                ("v_mu_gf", C.c_void_p), ("v_lambda_gf", C.c_void_p)]


class AlsStats(C.Structure):
    _fields_ = [("train_metric", C.c_double), ("device_seconds", C.c_double), ("levels", C.c_uint32),
                ("reserved", C.c_uint32), ("sum_e_sqr", C.c_double)]


class Info(C.Structure):
    _fields_ = [("n_local", C.c_uint64), ("k_padded", C.c_int32), ("device", C.c_int32),
                ("bytes_params", C.c_uint64), ("device_name", C.c_char * 64), ("arch", C.c_char * 32)]


# every symbol include/fmx.h declares: (name, restype, argtypes)
H = C.c_void_p
SYMBOLS = [
    ("fmx_create", C.c_int, [C.POINTER(Config), C.POINTER(H)]),
    ("fmx_destroy", C.c_int, [H]),
    ("fmx_last_error", C.c_char_p, [H]),
    ("fmx_abi_version", C.c_int, []),
    ("fmx_default_w0_chunk", C.c_uint32, [C.c_double, C.c_int]),
    ("fmx_release_cached_memory", C.c_int, []),
    ("fmx_device_count", C.c_int, []),
    ("fmx_set_params", C.c_int, [H, C.c_double, C.c_void_p, C.c_void_p]),
    ("fmx_get_params", C.c_int, [H, C.POINTER(C.c_double), C.c_void_p, C.c_void_p]),
    ("fmx_init_params", C.c_int, [H, C.c_double, C.c_double, C.c_uint64]),
    ("fmx_get_param_rows", C.c_int, [H, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("fmx_get_w0", C.c_int, [H, C.POINTER(C.c_double)]),
    ("fmx_save_model", C.c_int, [H, C.c_char_p]),
    ("fmx_load_model", C.c_int, [H, C.c_char_p]),
    ("fmx_set_groups", C.c_int, [H, C.c_void_p, C.c_uint32]),
    ("fmx_upload_rows", C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64]),
    ("fmx_upload_block_rows", C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                        C.POINTER(Relation), C.c_uint32]),
    ("fmx_upload_block_rows_ex", C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                           C.POINTER(Relation), C.c_uint32, C.c_uint32]),
    ("fmx_read_libsvm", C.c_int, [C.c_char_p, C.POINTER(HostRows), C.c_char_p, C.c_size_t]),
    ("fmx_read_binary", C.c_int, [C.c_char_p, C.POINTER(HostRows), C.c_char_p, C.c_size_t]),
    ("fmx_free_host_rows", None, [C.POINTER(HostRows)]),
    ("fmx_synth_rows", C.c_int, [H, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]),
    ("fmx_synth_rows_ex", C.c_int, [H, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]),
    ("fmx_free_rows", C.c_int, [H, C.c_int]),
    ("fmx_rows_info", C.c_int, [H, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    ("fmx_download_rows", C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("fmx_predict", C.c_int, [H, C.c_int, C.c_void_p]),
    ("fmx_evaluate", C.c_int, [H, C.c_int, C.POINTER(Eval)]),
    ("fmx_sgd_epoch", C.c_int, [H, C.c_int, C.POINTER(SgdOpts), C.POINTER(EpochStats)]),
    ("fmx_sgd_batch_info", C.c_int, [H, C.c_int, C.POINTER(SgdOpts), C.POINTER(BatchInfo)]),
    ("fmx_get_place_info", C.c_int, [H, C.POINTER(PlaceInfo)]),
    ("fmx_batch_rule", C.c_int, [C.c_int32, C.c_double, C.c_double, C.c_uint32, C.POINTER(BatchInfo)]),
    ("fmx_place_layout", C.c_int, [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    ("fmx_partial_floats", C.c_int, [H, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("fmx_sgd_partial", C.c_int, [H, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("fmx_sgd_finish", C.c_int, [H, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p, C.POINTER(SgdOpts), C.c_void_p]),
    ("fmx_predict_finish", C.c_int, [H, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("fmx_shard_place", C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    ("fmx_shard_global", C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]),
    ("fmx_comm_unique_id", C.c_int, [C.c_void_p]),
    ("fmx_comm_init_rank", C.c_int, [H, C.c_void_p, C.c_int, C.c_int]),
    ("fmx_comm_destroy", C.c_int, [H]),
    ("fmx_group_create", C.c_int, [C.POINTER(H), C.c_int, C.POINTER(H)]),
    ("fmx_group_destroy", C.c_int, [H]),
    ("fmx_group_last_error", C.c_char_p, [H]),
    ("fmx_group_set_params", C.c_int, [H, C.c_double, C.c_void_p, C.c_void_p]),
    ("fmx_group_upload_rows", C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64]),
    ("fmx_group_sgd_epoch", C.c_int, [H, C.c_int, C.POINTER(SgdOpts), C.POINTER(EpochStats)]),
    ("fmx_group_predict", C.c_int, [H, C.c_int, C.c_void_p]),
    ("fmx_group_evaluate", C.c_int, [H, C.c_int, C.POINTER(Eval)]),
    ("fmx_group_als_begin", C.c_int, [H, C.c_int]),
    ("fmx_group_als_moments", C.c_int, [H, C.c_void_p]),
    ("fmx_group_als_sweep", C.c_int, [H, C.POINTER(AlsOpts), C.POINTER(AlsStats)]),
    ("fmx_group_als_end", C.c_int, [H]),
    ("fmx_sgda_begin", C.c_int, [H]),
    ("fmx_sgda_epoch", C.c_int, [H, C.c_int, C.c_int, C.c_int, C.POINTER(EpochStats)]),
    ("fmx_sgda_epoch_minibatch", C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(EpochStats)]),
    ("fmx_sgda_get_reg", C.c_int, [H, C.c_void_p]),
    ("fmx_sgda_end", C.c_int, [H]),
    ("fmx_als_begin", C.c_int, [H, C.c_int]),
    ("fmx_als_moments", C.c_int, [H, C.c_void_p]),
    ("fmx_als_sweep", C.c_int, [H, C.POINTER(AlsOpts), C.POINTER(AlsStats)]),
    ("fmx_als_end", C.c_int, [H]),
    ("fmx_get_info", C.c_int, [H, C.POINTER(Info)]),
    ("fmx_synchronize", C.c_int, [H]),
]

_lib = None


def load():
    """dlopen libfmx.so and bind every declared symbol.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        try:                       # torch bundles its own HIP runtime: when both live in one process torch's
            import torch  # noqa: F401   # must be loaded first, or torch.cuda finds no device afterwards
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise ImportError("libfm_amd/libfmx.so is not built: run `python -m libfm_amd.build` "
                              "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)          # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def default_w0_chunk(learn_rate, task):
    """what fmx_sgd_opts::w0_chunk = 0 resolves to (host arithmetic inside libfmx.so; no device needed)"""
    return int(load().fmx_default_w0_chunk(float(learn_rate), int(task)))


def _ptr(a):
    return None if a is None else a.ctypes.data


class Handle:
    """Thin OO wrapper over an fmx_handle; every method is one C-ABI call."""

    def __init__(self, num_attribute, num_factor, k0=True, k1=True, task=TASK_REGRESSION, reg0=0.0, regw=0.0, regv=0.0,
                 learn_rate=0.0, min_target=0.0, max_target=0.0, device=-1, shard_rank=0, shard_world=1, shard_hash=0,
                 place_candidates=0, als_split_min=None, exchange_runs=0, exchange_algo=0):
        self.lib = load()
        self.cfg = Config(int(num_attribute), int(num_factor), int(bool(k0)), int(bool(k1)), int(task),
                          float(reg0), float(regw), float(regv), float(learn_rate), float(min_target),
                          float(max_target), int(device), int(shard_rank), int(shard_world), int(shard_hash),
                          int(place_candidates), int(ALS_SPLIT_MIN if als_split_min is None else als_split_min), int(exchange_runs), int(exchange_algo))
        self.h = H()
        rc = self.lib.fmx_create(C.byref(self.cfg), C.byref(self.h))
        if rc != FMX_OK:
            raise FmxError(rc, self.lib.fmx_last_error(None).decode())
        self.n, self.k = int(num_attribute), int(num_factor)
        self.G = 1                                   # attribute groups (set_groups)
        self.n_local = int(self.info().n_local)

    def _chk(self, rc):
        if rc != FMX_OK:
            raise FmxError(rc, self.lib.fmx_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.fmx_destroy(self.h)
            self.h = H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # parameters ------------------------------------------------------------------------------
    def set_params(self, w0, w, v):
        w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
        v = None if v is None else np.ascontiguousarray(v, dtype=np.float64)
        if w is not None:
            assert w.shape == (self.n,)
        if v is not None:
            assert v.shape == (self.k, self.n)
        self._chk(self.lib.fmx_set_params(self.h, float(w0), _ptr(w), _ptr(v)))

    def get_params(self, w=None, v=None):
        w0 = C.c_double(0)
        if w is None:
            w = np.zeros(self.n, dtype=np.float64)
        if v is None:
            v = np.zeros((self.k, self.n), dtype=np.float64)
        self._chk(self.lib.fmx_get_params(self.h, C.byref(w0), _ptr(w), _ptr(v) if self.k > 0 else None))
        return w0.value, w, v

    def init_params(self, mean, stdev, seed):
        self._chk(self.lib.fmx_init_params(self.h, float(mean), float(stdev), int(seed)))

    def get_param_rows(self, ids):
        """(w[ids], v[:, ids]) for spot checks when the full model does not fit the host."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        w = np.zeros(len(ids), dtype=np.float64)
        v = np.zeros((len(ids), max(self.k, 1)), dtype=np.float64)
        self._chk(self.lib.fmx_get_param_rows(self.h, _ptr(ids), len(ids), _ptr(w), _ptr(v) if self.k > 0 else None))
        return w, v[:, :self.k].T.copy()

    def save_model(self, path):
        """fm_model::saveModel (fm_model.h:132-154) straight from the device table"""
        self._chk(self.lib.fmx_save_model(self.h, os.fsencode(path)))

    def load_model(self, path):
        """fm_model::loadModel (fm_model.h:160-190); raises FmxError "malformed model file" where the reference returns 0"""
        self._chk(self.lib.fmx_load_model(self.h, os.fsencode(path)))

    def get_w0(self):
        w0 = C.c_double(0)
        self._chk(self.lib.fmx_get_w0(self.h, C.byref(w0)))
        return w0.value

    def set_groups(self, group):
        """attribute -> group ids (`-meta`, Data.h:85-97); None = one group."""
        if group is None:
            self._chk(self.lib.fmx_set_groups(self.h, None, 1))
            self.G = 1
            return
        group = np.ascontiguousarray(group, dtype=np.uint32)
        if len(group) != self.n:
            raise ValueError("set_groups: need one group id per feature (%d), got %d" % (self.n, len(group)))
        G = int(group.max()) + 1 if len(group) else 1
        self._chk(self.lib.fmx_set_groups(self.h, _ptr(group), G))
        self.G = max(G, 1)

    # rows ------------------------------------------------------------------------------------
    def upload_rows(self, slot, entries, row_ptr, target):
        entries = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        target = None if target is None else np.ascontiguousarray(target, dtype=np.float32)
        n_rows = len(row_ptr) - 1
        self._chk(self.lib.fmx_upload_rows(self.h, slot, _ptr(entries) if len(entries) else None, _ptr(row_ptr),
                                           _ptr(target), n_rows, len(entries)))
        return n_rows

    def synth_rows(self, slot, seed, row0, n_rows, nnz, shape=SYNTH_UNIFORM):
        self._chk(self.lib.fmx_synth_rows_ex(self.h, slot, int(seed), int(row0), int(n_rows), int(nnz), int(shape)))

    def download_rows(self, slot):
        n_rows, nnz = C.c_uint32(0), C.c_uint64(0)
        self._chk(self.lib.fmx_rows_info(self.h, slot, C.byref(n_rows), C.byref(nnz)))
        ent = np.zeros(nnz.value, dtype=ENTRY_DTYPE)
        rp = np.zeros(n_rows.value + 1, dtype=np.uint64)
        y = np.zeros(n_rows.value, dtype=np.float32)
        self._chk(self.lib.fmx_download_rows(self.h, slot, _ptr(ent) if nnz.value else None, _ptr(rp), _ptr(y)))
        return ent, rp, y

    def free_rows(self, slot):
        self._chk(self.lib.fmx_free_rows(self.h, slot))

    # compute ---------------------------------------------------------------------------------
    def upload_block_rows(self, slot, entries, row_ptr, target, relations, keep=False):
        """relations: list of (entries, row_ptr, data_row_to_relation_row, attr_offset) -- `-relation` blocks
        (relation.h:32-60).  keep=False: the joined rows are expanded on the device; keep=True (FMX_BLOCKS_KEEP): main rows and
        blocks stay apart, ALS / MCMC sweep the blocks through per-block-row caches like the reference."""
        entries = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        target = None if target is None else np.ascontiguousarray(target, dtype=np.float32)
        n_rows = len(row_ptr) - 1
        alive, arr = [], (Relation * max(len(relations), 1))()
        for i, (re, rp, mp, off) in enumerate(relations):
            re = np.ascontiguousarray(re, dtype=ENTRY_DTYPE)
            rp = np.ascontiguousarray(rp, dtype=np.uint64)
            mp = np.ascontiguousarray(mp, dtype=np.uint32)
            if len(mp) != n_rows:
                raise ValueError("relation %d: the row mapping has %d entries for %d data rows" % (i, len(mp), n_rows))
            alive += [re, rp, mp]
            arr[i] = Relation(_ptr(re) if len(re) else None, _ptr(rp), len(rp) - 1, 0, len(re), _ptr(mp), int(off))
        self._chk(self.lib.fmx_upload_block_rows_ex(self.h, slot, _ptr(entries) if len(entries) else None, _ptr(row_ptr),
                                                    _ptr(target), n_rows, len(entries), arr, len(relations), 1 if keep else 0))
        return n_rows

    def predict(self, slot, n_rows):
        out = np.zeros(n_rows, dtype=np.float64)
        self._chk(self.lib.fmx_predict(self.h, slot, _ptr(out)))
        return out

    def evaluate(self, slot):
        ev = Eval()
        self._chk(self.lib.fmx_evaluate(self.h, slot, C.byref(ev)))
        return ev

    def sgd_epoch(self, slot, mode, apply=APPLY_DEFAULT, batch=0, w0_chunk=0, flags=0, bias_lag=0):
        opts = SgdOpts(mode, apply, batch, w0_chunk, flags, bias_lag)
        st = EpochStats()
        self._chk(self.lib.fmx_sgd_epoch(self.h, slot, C.byref(opts), C.byref(st)))
        return st

    def sgd_batch_info(self, slot, batch=0, mode=SGD_MINIBATCH, apply=APPLY_DEFAULT):
        """the batch fmx_sgd_epoch would run with on this slot, the rows' collision mass and the gain (fmx_batch_info)"""
        opts = SgdOpts(mode, apply, batch, 0, 0, 0)
        bi = BatchInfo()
        self._chk(self.lib.fmx_sgd_batch_info(self.h, slot, C.byref(opts), C.byref(bi)))
        return bi

    def place_info(self):
        """how fmx_create placed the parameter tables (fmx_place_info)"""
        pi = PlaceInfo()
        self._chk(self.lib.fmx_get_place_info(self.h, C.byref(pi)))
        return pi

    def partial_floats(self, batch):
        n = C.c_uint64(0)
        self._chk(self.lib.fmx_partial_floats(self.h, batch, C.byref(n)))
        return n.value

    def sgd_partial(self, slot, row0, n_rows, d_partial_ptr, stream=None):
        self._chk(self.lib.fmx_sgd_partial(self.h, slot, row0, n_rows, d_partial_ptr, stream))

    def sgd_finish(self, slot, row0, n_rows, d_partial_ptr, apply=APPLY_DEFAULT, w0_chunk=0, stream=None, batch=0, flags=0, bias_lag=0):
        opts = SgdOpts(SGD_MINIBATCH, apply, batch or n_rows, w0_chunk, flags, bias_lag)
        self._chk(self.lib.fmx_sgd_finish(self.h, slot, row0, n_rows, d_partial_ptr, C.byref(opts), stream))

    def predict_finish(self, n_rows, d_partial_ptr, d_yhat_ptr, stream=None):
        self._chk(self.lib.fmx_predict_finish(self.h, n_rows, d_partial_ptr, d_yhat_ptr, stream))

    # SGDA ------------------------------------------------------------------------------------
    def sgda_begin(self):
        self._chk(self.lib.fmx_sgda_begin(self.h))

    def sgda_epoch(self, train_slot, val_slot, do_lambda):
        st = EpochStats()
        self._chk(self.lib.fmx_sgda_epoch(self.h, train_slot, val_slot, int(do_lambda), C.byref(st)))
        return st

    def sgda_epoch_minibatch(self, train_slot, val_slot, do_lambda, batch=0, w0_chunk=0):
        st = EpochStats()
        self._chk(self.lib.fmx_sgda_epoch_minibatch(self.h, train_slot, val_slot, int(do_lambda), int(batch), int(w0_chunk), C.byref(st)))
        return st

    def sgda_get_reg(self):
        """[G][1 + k]: column 0 = reg_w(g), columns 1.. = reg_v(g, f)."""
        out = np.zeros((self.G, 1 + self.k), dtype=np.float64)
        self._chk(self.lib.fmx_sgda_get_reg(self.h, _ptr(out)))
        return out

    def sgda_end(self):
        self._chk(self.lib.fmx_sgda_end(self.h))

    # ALS / MCMC ----------------------------------------------------------------------------
    def als_begin(self, train_slot):
        self._chk(self.lib.fmx_als_begin(self.h, train_slot))

    def als_moments(self):
        """(sum e^2, sum e, m) with m[1 + k][G][2]: row 0 = w, row 1+f = v_f; per group {sum theta, sum theta^2}."""
        out = np.zeros(2 + 2 * self.G * (1 + self.k), dtype=np.float64)
        self._chk(self.lib.fmx_als_moments(self.h, _ptr(out)))
        return out[0], out[1], out[2:].reshape(1 + self.k, self.G, 2)

    def als_sweep(self, w_lambda, v_lambda, alpha=1.0, w_mu=0.0, v_mu=0.0, do_sample=False, seed=0):
        """w_lambda / w_mu: scalar or [G]; v_lambda / v_mu: scalar, [k] or [G][k] (the reference's w_lambda(g),
        v_lambda(g,f); fm_learn_mcmc.h:1116-1122)."""
        opts, keep = self._als_opts(w_lambda, v_lambda, alpha, w_mu, v_mu, do_sample, seed)
        st = AlsStats()
        self._chk(self.lib.fmx_als_sweep(self.h, C.byref(opts), C.byref(st)))
        return st

    def _als_opts(self, w_lambda, v_lambda, alpha, w_mu, v_mu, do_sample, seed):
        G, k = self.G, max(self.k, 1)

        def tab_w(x):
            return np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float64), (G,)))

        def tab_v(x):
            x = np.asarray(x, dtype=np.float64)
            if x.ndim == 1 and x.shape[0] != 1:
                if G > 1 and G == k:                 # a [G] per-group vector and a [k] per-factor vector look the same
                    raise ValueError("1-D v_lambda / v_mu is ambiguous with num_groups == num_factor: pass a [G][k] table")
                if x.shape[0] == G and G > 1:
                    x = x[:, None]                   # one value per group
                elif x.shape[0] != k:
                    raise ValueError("1-D v_lambda / v_mu must have num_groups (%d) or num_factor (%d) entries" % (G, k))
            return np.ascontiguousarray(np.broadcast_to(x, (G, k)))
        wl, wm, vl, vm = tab_w(w_lambda), tab_w(w_mu), tab_v(v_lambda), tab_v(v_mu)
        opts = AlsOpts(alpha, float(wm[0]), float(wl[0]), float(vm[0, 0]), float(vl[0, 0]), int(do_sample), 0, seed,
                       _ptr(vm[0]) if G == 1 else None, _ptr(vl[0]) if G == 1 else None,
                       G if G > 1 else 0, 0, _ptr(wm), _ptr(wl), _ptr(vm), _ptr(vl))
        if G > 1 and self.k != k:                    # k == 0: tables are [G][0]
            opts.v_mu_gf = opts.v_lambda_gf = None
        return opts, (wl, wm, vl, vm)                # the arrays must outlive the call

    def als_end(self):
        self._chk(self.lib.fmx_als_end(self.h))

    # one process per GPU ---------------------------------------------------------------------
    def comm_init_rank(self, unique_id, rank, world):
        """unique_id: the COMM_ID_BYTES bytes rank 0 got from comm_unique_id() (handed over by the launcher)"""
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        self._chk(self.lib.fmx_comm_init_rank(self.h, buf, int(rank), int(world)))

    def comm_destroy(self):
        self._chk(self.lib.fmx_comm_destroy(self.h))

    def info(self):
        inf = Info()
        self._chk(self.lib.fmx_get_info(self.h, C.byref(inf)))
        return inf

    def synchronize(self):
        self._chk(self.lib.fmx_synchronize(self.h))


def release_cached_memory():
    """fmx_release_cached_memory: the placed arena fmx_destroy keeps per device for the next fmx_create goes back to the device"""
    load().fmx_release_cached_memory()


def batch_rule(task, learn_rate, collision_mass, requested=0):
    """the batch fmx_sgd_epoch would run with for rows of this collision mass (fmx_batch_rule; host arithmetic)"""
    bi = BatchInfo()
    rc = load().fmx_batch_rule(int(task), float(learn_rate), float(collision_mass), int(requested), C.byref(bi))
    if rc != 0:
        raise FmxError(rc, "fmx_batch_rule: bad argument")
    return bi


def place_layout(v_bytes, w_bytes):
    """(chunk_bytes, chunks, w_offset) of the arena fmx_create builds for tables of these sizes (fmx_place_layout; host arithmetic)"""
    ch, t, off = C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
    rc = load().fmx_place_layout(int(v_bytes), int(w_bytes), C.byref(ch), C.byref(t), C.byref(off))
    if rc != 0:
        raise FmxError(rc, "fmx_place_layout: bad argument")
    return ch.value, t.value, off.value


def shard_place(n, world, shard_hash, ids):
    """(owner, local_row) of every feature id under the library's ownership rule (host arithmetic, no device)"""
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    owner, local = np.zeros(len(ids), dtype=np.int32), np.zeros(len(ids), dtype=np.uint32)
    rc = load().fmx_shard_place(int(n), int(world), int(shard_hash), _ptr(ids), len(ids), _ptr(owner), _ptr(local))
    if rc != FMX_OK:
        raise FmxError(rc, "fmx_shard_place: bad argument")
    return owner, local


def shard_global(n, world, shard_hash, rank, local_rows):
    local_rows = np.ascontiguousarray(local_rows, dtype=np.uint32)
    ids = np.zeros(len(local_rows), dtype=np.uint32)
    rc = load().fmx_shard_global(int(n), int(world), int(shard_hash), int(rank), _ptr(local_rows), len(local_rows), _ptr(ids))
    if rc != FMX_OK:
        raise FmxError(rc, "fmx_shard_global: bad argument")
    return ids


def comm_unique_id():
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().fmx_comm_unique_id(buf)
    if rc != FMX_OK:
        raise FmxError(rc, load().fmx_last_error(None).decode())
    return buf.raw


class Group:
    """fmx_group: the feature shards of one model, driven by this one process (RCCL between devices, a local reduction
    when all shards share a device)."""

    def __init__(self, handles):
        self.lib = load()
        self.handles = list(handles)
        arr = (H * len(self.handles))(*[h.h for h in self.handles])
        self.g = H()
        rc = self.lib.fmx_group_create(arr, len(self.handles), C.byref(self.g))
        if rc != FMX_OK:
            raise FmxError(rc, (self.lib.fmx_last_error(self.handles[0].h) or self.lib.fmx_last_error(None)).decode())

    def _chk(self, rc):
        if rc != FMX_OK:
            raise FmxError(rc, self.lib.fmx_group_last_error(self.g).decode())

    def set_params(self, w0, w, v):
        """the full fm_model block for every shard, crossing PCIe once (fmx_group_set_params)"""
        h0 = self.handles[0]
        w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
        v = None if v is None else np.ascontiguousarray(v, dtype=np.float64)
        self._chk(self.lib.fmx_group_set_params(self.g, float(w0), _ptr(w), _ptr(v) if h0.k > 0 else None))

    def upload_rows(self, slot, entries, row_ptr, target):
        """rows for every shard, crossing PCIe once; each shard filters its own features on its device (fmx_group_upload_rows)"""
        entries = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        target = None if target is None else np.ascontiguousarray(target, dtype=np.float32)
        self._chk(self.lib.fmx_group_upload_rows(self.g, slot, _ptr(entries) if len(entries) else None, _ptr(row_ptr), _ptr(target),
                                                 len(row_ptr) - 1, len(entries)))

    def sgd_epoch(self, slot, mode=SGD_MINIBATCH, apply=APPLY_DEFAULT, batch=0, w0_chunk=0, flags=0, bias_lag=0):
        opts = SgdOpts(mode, apply, batch, w0_chunk, flags, bias_lag)
        st = EpochStats()
        self._chk(self.lib.fmx_group_sgd_epoch(self.g, slot, C.byref(opts), C.byref(st)))
        return st

    def predict(self, slot, n_rows):
        out = np.zeros(n_rows, dtype=np.float64)
        self._chk(self.lib.fmx_group_predict(self.g, slot, _ptr(out)))
        return out

    def evaluate(self, slot):
        ev = Eval()
        self._chk(self.lib.fmx_group_evaluate(self.g, slot, C.byref(ev)))
        return ev

    # ALS / MCMC over the shards ----------------------------------------------------------------
    def als_begin(self, train_slot):
        self._chk(self.lib.fmx_group_als_begin(self.g, train_slot))

    def als_sweep(self, w_lambda, v_lambda, alpha=1.0, w_mu=0.0, v_mu=0.0, do_sample=False, seed=0):
        opts, keep = self.handles[0]._als_opts(w_lambda, v_lambda, alpha, w_mu, v_mu, do_sample, seed)
        st = AlsStats()
        self._chk(self.lib.fmx_group_als_sweep(self.g, C.byref(opts), C.byref(st)))
        return st

    def als_moments(self):
        h0 = self.handles[0]
        out = np.zeros(2 + 2 * h0.G * (1 + h0.k), dtype=np.float64)
        self._chk(self.lib.fmx_group_als_moments(self.g, _ptr(out)))
        return out[0], out[1], out[2:].reshape(1 + h0.k, h0.G, 2)

    def als_end(self):
        self._chk(self.lib.fmx_group_als_end(self.g))

    def get_params(self):
        """the full model: every shard writes its own features into the same host arrays"""
        h0 = self.handles[0]
        w, v = np.zeros(h0.n, dtype=np.float64), np.zeros((h0.k, h0.n), dtype=np.float64)
        w0 = 0.0
        for h in self.handles:
            w0, w, v = h.get_params(w, v)
        return w0, w, v

    def close(self):
        if self.g:
            self.lib.fmx_group_destroy(self.g)
            self.g = H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""Host-side mirror of the reference's learner interface on top of the C-ABI (include/fmx.h).

The names, fields, argument meaning and error behaviour follow the reference so that the parity tests read like
the reference's own driver (there are no reference tests to imitate):

    FMModel      <-> class fm_model            /root/reference/src/fm_core/fm_model.h:36-66
    Data         <-> class Data                /root/reference/src/libfm/src/Data.h:49-73   (rows + target only)
    FMLearnSGD   <-> fm_learn_sgd_element      /root/reference/src/libfm/src/fm_learn_sgd_element.h:34-78
                     (+ fm_learn_sgd.h:34-90, fm_learn.h:31-153)

All arithmetic happens in libfmx.so on the GPU; this file only moves buffers and applies the host-side
clamp / sigmoid of fm_learn_sgd::predict (fm_learn_sgd.h:80-87).  No CPU fallback exists.
"""
import sys

import numpy as np

from . import capi

TASK_REGRESSION = capi.TASK_REGRESSION        # fm_learn.h:46
TASK_CLASSIFICATION = capi.TASK_CLASSIFICATION  # fm_learn.h:47


class Data:
    """Rows in the reference's layout: AoS entries {u32 id; f32 value} + row offsets + float targets."""

    def __init__(self, entries, row_ptr, target):
        self.entries = np.ascontiguousarray(entries, dtype=capi.ENTRY_DTYPE)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        self.target = np.ascontiguousarray(target, dtype=np.float32)
        self.num_cases = len(self.target)                       # Data.h:60
        self.num_feature = int(self.entries["id"].max()) + 1 if len(self.entries) else 0   # Data.h:59
        self.min_target = float(self.target.min()) if self.num_cases else 0.0   # Data.h:62-63
        self.max_target = float(self.target.max()) if self.num_cases else 0.0
        self.relation = []                                       # Data.h:68: DVector<RelationJoin>
        self.keep_blocks = True        # relations: keep main rows and blocks apart on the device (per-block caches, like the
        #                                reference) instead of materialising the joined rows (FMX_BLOCKS_KEEP / _EXPAND)

    def add_relation(self, rel, data_row_to_relation_row, attr_offset):
        """RelationJoin (relation.h:53-60): `rel` = a data.Relation (the block's own rows), the main-row -> block-row
        mapping of THIS data set (<prefix>.train / <prefix>.test) and the block's first global attribute id
        (RelationData::attr_offset, libfm.cpp:213-216)."""
        m = np.ascontiguousarray(data_row_to_relation_row, dtype=np.uint32)
        if len(m) != self.num_cases:
            raise ValueError("relation mapping has %d rows, the data set %d" % (len(m), self.num_cases))   # relation.h:149
        self.relation.append((rel.entries, rel.row_ptr, m, int(attr_offset)))

    def upload(self, h, slot):
        if self.relation:
            h.upload_block_rows(slot, self.entries, self.row_ptr, self.target, self.relation, keep=self.keep_blocks)
        else:
            h.upload_rows(slot, self.entries, self.row_ptr, self.target)


class FMModel:
    """fm_model: parameters in the reference layout (w0, w[n], v[k][n] fp64) and its hyper-parameters."""

    def __init__(self):
        self.num_attribute = 0
        self.num_factor = 0
        self.k0, self.k1 = True, True
        self.reg0 = self.regw = self.regv = 0.0
        self.init_stdev, self.init_mean = 0.01, 0.0             # fm_model.h:69-78
        self.w0 = 0.0
        self.w = None
        self.v = None

    def init(self, rng=None):
        """fm_model::init (fm_model.h:91-99): w0 = 0, w = 0, v ~ N(init_mean, init_stdev).
        The reference draws from libc rand(); here a numpy Generator is used (pass the reference's own
        initial parameters for trajectory parity, as the tests do)."""
        rng = rng if rng is not None else np.random.default_rng(0)
        self.w0 = 0.0
        self.w = np.zeros(self.num_attribute, dtype=np.float64)
        if self.init_stdev == 0:
            self.v = np.full((self.num_factor, self.num_attribute), self.init_mean, dtype=np.float64)
        else:
            self.v = self.init_mean + self.init_stdev * rng.standard_normal((self.num_factor, self.num_attribute))


    # fm_model::saveModel (fm_model.h:132-154): text, ostream default precision (= printf %g)
    def save_model(self, path):
        with open(path, "w") as f:
            if self.k0:
                f.write("#global bias W0\n%g\n" % self.w0)
            if self.k1:
                f.write("#unary interactions Wj\n")
                f.write("".join("%g\n" % x for x in self.w))
            f.write("#pairwise interactions Vj,f\n")
            for j in range(self.num_attribute):
                f.write(" ".join("%g" % x for x in self.v[:, j]) + "\n")

    # fm_model::loadModel (fm_model.h:160-190); returns False on a malformed file like the reference returns 0.
    # (The reference's splitString yields no token for a line without a blank, so k = 1 models fail to load there,
    #  fm_model.h:195-205; this reader accepts them.)
    def load_model(self, path):
        try:
            lines = open(path).read().splitlines()
            pos = 0
            if self.k0:
                self.w0 = float(lines[pos + 1]); pos += 2
            if self.k1:
                pos += 1
                self.w = np.array([float(x) for x in lines[pos:pos + self.num_attribute]], dtype=np.float64)
                pos += self.num_attribute
            pos += 1
            rows = [[float(x) for x in ln.split(" ") if x != ""] for ln in lines[pos:pos + self.num_attribute]]
            if len(rows) != self.num_attribute or any(len(r) != self.num_factor for r in rows):
                return False
            self.v = np.ascontiguousarray(np.array(rows, dtype=np.float64).reshape(self.num_attribute, self.num_factor).T)
            return True                                          # (-dim 1,1,0: n empty lines, as fm_model.h:160-190 reads them)
        except (OSError, ValueError, IndexError):
            return False


class FMLearnSGD:
    """fm_learn_sgd_element on the GPU.

    Public knobs are plain fields set by the driver, like libfm.cpp:271-309, 387-403 does by direct writes:
    fm, min_target, max_target, task, num_iter, learn_rate.  GPU-only knobs: mode ('sequential' | 'minibatch' |
    'hogwild'), batch, w0_chunk, apply ('default' | 'segmented' | 'atomic' | 'store'), device."""

    MODES = {"sequential": capi.SGD_SEQUENTIAL, "minibatch": capi.SGD_MINIBATCH, "hogwild": capi.SGD_HOGWILD}
    APPLY = {"default": capi.APPLY_DEFAULT, "atomic": capi.APPLY_ATOMIC, "store": capi.APPLY_STORE,
             "segmented": capi.APPLY_SEGMENTED, "fused": capi.APPLY_FUSED}

    def __init__(self):
        self.fm = None
        self.min_target = 0.0
        self.max_target = 0.0
        self.task = TASK_REGRESSION
        self.num_iter = 100                                     # libfm.cpp:274 default
        self.learn_rate = None                                  # no default in the reference (libfm.cpp:391-392)
        self.mode = "minibatch"
        self.batch = 0                                          # 0: the library's choice (262144 cut to the rows' stability bound)
        self.w0_chunk = 0
        self.apply = "fused"                                    # the batch rule in one pass (hogwild: "default")
        self.bias_lag = 2
        self.reject_unstable = True                             # an explicit batch the rule diverges at raises instead of training
        self.device = -1
        self.log = []                                           # one dict per iteration (rlog fields)
        self.out = sys.stdout
        self._h = None
        self._slots = {}

    # fm_learn::init (fm_learn.h:73-91): here it creates the device context and uploads the parameters
    def init(self):
        if self.task not in (TASK_REGRESSION, TASK_CLASSIFICATION):
            raise ValueError("unknown task")                    # fm_learn.h:81
        if self.learn_rate is None:
            raise ValueError("learn_rate must be set")          # the reference asserts (libfm.cpp:391-392)
        fm = self.fm
        self._h = capi.Handle(fm.num_attribute, fm.num_factor, fm.k0, fm.k1, self.task, fm.reg0, fm.regw, fm.regv,
                              self.learn_rate, self.min_target, self.max_target, device=self.device)
        self._h.set_params(fm.w0, fm.w, fm.v)

    def _slot(self, data):
        key = id(data)
        if key not in self._slots:
            slot = len(self._slots)
            if slot >= capi.MAX_SLOTS:
                raise RuntimeError("too many data sets")
            data.upload(self._h, slot)
            self._slots[key] = slot
        return self._slots[key]

    # fm_learn_sgd_element::learn (fm_learn_sgd_element.h:48-78)
    def learn(self, train, test):
        print("learnrate=%g" % self.learn_rate, file=self.out)   # fm_learn_sgd.h:57-59
        print("#iterations=%d" % self.num_iter, file=self.out)
        print("SGD: DON'T FORGET TO SHUFFLE THE ROWS IN TRAINING DATA TO GET THE BEST RESULTS.", file=self.out)
        st = self._slot(train)
        for i in range(self.num_iter):
            apply_ = self.APPLY[self.apply]
            if self.mode != "minibatch" and apply_ == capi.APPLY_FUSED:
                apply_ = capi.APPLY_DEFAULT
            flags = capi.FLAG_REJECT_UNSTABLE if (self.reject_unstable and self.mode == "minibatch") else 0
            flags |= capi.FLAG_KEEP_WSIDE                # the train set is evaluated after every epoch (fm_learn_sgd_element.h:69-70)
            stats = self._h.sgd_epoch(st, self.MODES[self.mode], apply_, self.batch, self.w0_chunk, flags,
                                      self.bias_lag if apply_ == capi.APPLY_FUSED else 0)
            if i == 0 and self.mode == "minibatch" and (stats.status & capi.STAT_BATCH_CUT):
                print("libfmx: batch %d (collision mass of the rows %.4g, gain %.3g)" % (stats.batch_used, stats.collision_mass,
                                                                                        stats.batch_gain), file=sys.stderr)
            rmse_train = self.evaluate(train)
            rmse_test = self.evaluate(test)
            print("#Iter=%3d\tTrain=%g\tTest=%g" % (i, rmse_train, rmse_test), file=self.out)   # :71
            self.log.append({"rmse_train": rmse_train, "time_learn": stats.device_seconds})
        self.sync_model()

    def sync_model(self):
        """after learn() the host fm_model holds the learned parameters (main reads it, libfm.cpp:431-434)."""
        self.fm.w0, self.fm.w, self.fm.v = self._h.get_params(self.fm.w, self.fm.v)

    # fm_learn::evaluate (fm_learn.h:93-153): rmse for regression, accuracy for classification
    def evaluate(self, data):
        ev = self._h.evaluate(self._slot(data))
        return ev.rmse if self.task == TASK_REGRESSION else ev.accuracy

    def predict_raw(self, data):
        """fm_learn::predict_case over the data set (fm_learn.h:63-65)."""
        return self._h.predict(self._slot(data), data.num_cases)

    # fm_learn_sgd::predict (fm_learn_sgd.h:76-90)
    def predict(self, data):
        p = self.predict_raw(data)
        if self.task == TASK_REGRESSION:
            p = np.minimum(self.max_target, p)
            p = np.maximum(self.min_target, p)
        elif self.task == TASK_CLASSIFICATION:
            p = 1.0 / (1.0 + np.exp(-p))
        else:
            raise ValueError("task not supported")              # fm_learn_sgd.h:85
        return p

    def close(self):
        if self._h is not None:
            self._h.close()
            self._h = None


def _ref_erf(x):
    """the reference's 5-term erf polynomial (random.h:45-59), vectorised."""
    x = np.asarray(x, dtype=np.float64)
    t = np.where(x >= 0, 1.0 / (1.0 + 0.3275911 * x), 1.0 / (1.0 - 0.3275911 * x))
    r = 1.0 - (t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429))))) * np.exp(-x * x)
    return np.where(x >= 0, r, -r)


def cdf_gaussian(x):
    """random.h:65-67"""
    return 0.5 + 0.5 * _ref_erf(0.707106781 * np.asarray(x, dtype=np.float64))


class FMLearnALS:
    """fm_learn_mcmc_simultaneous with do_sample = 0, do_multilevel = 0 -- what `-method als` runs
    (libfm.cpp:135-139, 283-290) -- on the GPU.  Fields follow fm_learn_mcmc (fm_learn_mcmc.h:60-88):
    fm, min_target, max_target, task, num_iter; w_lambda / v_lambda are set from -regular like libfm.cpp:326-365.
    `groups` (attribute -> group id, the `-meta` file; fm_learn.h:40, Data.h:39-46) makes them per group:
    w_lambda [G], v_lambda [G] or [G][k] (libfm.cpp:353-363)."""

    def __init__(self):
        self.fm = None
        self.min_target = self.max_target = 0.0
        self.task = TASK_REGRESSION
        self.num_iter = 100
        self.w_lambda = 0.0
        self.v_lambda = 0.0
        self.do_sample = False
        self.seed = 0
        self.device = -1
        self.groups = None             # DataMetaInfo::attr_group (None = one group)
        self.out = sys.stdout
        self.pred_this = None          # fm_learn_mcmc.h:116
        self.pred_sum_all = None       # fm_learn_mcmc.h:114
        self.log = []
        self._h = None

    def init(self):
        fm = self.fm
        self._h = capi.Handle(fm.num_attribute, fm.num_factor, fm.k0, fm.k1, self.task, fm.reg0, fm.regw, fm.regv,
                              0.0, self.min_target, self.max_target, device=self.device)
        self._h.set_params(fm.w0, fm.w, fm.v)
        self._h.set_groups(self.groups)

    def _v_table(self, x):
        """v_lambda-like value -> [G][k].  Scalar, the full [G][k] table, or a 1-D vector that ALWAYS means one value per
        attribute group (what `-regular 'r0,w_1..w_G,v_1..v_G'` supplies, libfm.cpp:353-363) -- never per factor."""
        G, k = self._h.G, max(self.fm.num_factor, 1)
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1 and x.shape[0] != 1:
            if x.shape[0] != G:
                raise ValueError("a 1-D v_lambda holds one value per attribute group (%d), got %d" % (G, x.shape[0]))
            x = x[:, None]
        return np.ascontiguousarray(np.broadcast_to(x, (G, k)))

    # fm_learn_mcmc::learn + _learn (fm_learn_mcmc.h:1160-1201, fm_learn_mcmc_simultaneous.h:56-270)
    def learn(self, train, test):
        h = self._h
        train.upload(h, 0)
        test.upload(h, 1)
        self.pred_sum_all = np.zeros(test.num_cases)
        h.als_begin(0)
        for i in range(self.num_iter):
            st = h.als_sweep(self.w_lambda, self._v_table(self.v_lambda), 1.0, 0.0, 0.0, self.do_sample, self.seed)
            p = h.predict(1, test.num_cases)
            if self.task == TASK_REGRESSION:                  # :127-138
                self.pred_this = p
                self.pred_sum_all += np.maximum(self.min_target, np.minimum(self.max_target, p))
                rmse_test = float(np.sqrt(np.mean((self.pred_sum_all / (i + 1) - test.target) ** 2)))
                print("#Iter=%3d\tTrain=%g\tTest=%g" % (i, st.train_metric, rmse_test), file=self.out)
            else:                                             # :151-161
                self.pred_this = cdf_gaussian(p)
                self.pred_sum_all += self.pred_this
                acc = float(np.mean(((self.pred_sum_all / (i + 1)) >= 0.5) == (test.target >= 0)))
                print("#Iter=%3d\tTrain=%g\tTest=%g" % (i, st.train_metric, acc), file=self.out)
            self.log.append({"train": st.train_metric, "time_learn": st.device_seconds, "levels": st.levels})
        h.als_end()
        self.fm.w0, self.fm.w, self.fm.v = h.get_params(self.fm.w, self.fm.v)

    # fm_learn_mcmc::predict (fm_learn_mcmc.h:380-404)
    def predict(self, data):
        out = self.pred_sum_all / self.num_iter if self.do_sample else self.pred_this.copy()
        if self.task == TASK_REGRESSION:
            return np.maximum(self.min_target, np.minimum(self.max_target, out))
        return np.maximum(0.0, np.minimum(1.0, out))

    def close(self):
        if self._h is not None:
            self._h.close()
            self._h = None


class FMLearnMCMC(FMLearnALS):
    """fm_learn_mcmc_simultaneous with do_sample = 1, do_multilevel = 1 -- `-method mcmc` (libfm.cpp:283-290).

    The coordinate draws run on the GPU (fmx_als_sweep with do_sample = 1); the hyper-prior draws are scalar work
    and stay on the host, in the reference's order (draw_all, fm_learn_mcmc.h:430-527): alpha (:911-939), w_lambda
    (:985-1017), w_mu (:941-983), v_lambda (:1061-1097), v_mu (:1019-1059), from statistics reduced on the device
    (fmx_als_moments).  Hyper-priors alpha_0 = gamma_0 = beta_0 = 1, mu_0 = 0 (:1106-1109).  Random numbers come
    from numpy / a counter hash, not libc rand(): parity with the reference is statistical."""

    def __init__(self):
        super().__init__()
        self.do_sample = True
        self.do_multilevel = True
        self.alpha_0 = self.gamma_0 = self.beta_0 = 1.0
        self.mu_0 = 0.0

    def learn(self, train, test):
        h = self._h
        rng = np.random.default_rng(self.seed)
        k, n, G = self.fm.num_factor, self.fm.num_attribute, h.G
        train.upload(h, 0)
        test.upload(h, 1)
        self.pred_sum_all = np.zeros(test.num_cases)
        N = train.num_cases
        # meta->num_attr_per_group (Data.h:93-95)
        n_g = np.array([float(n)]) if self.groups is None else np.bincount(np.asarray(self.groups), minlength=G).astype(np.float64)
        alpha = 1.0
        w_mu, w_lambda = np.zeros(G), np.broadcast_to(np.asarray(self.w_lambda, dtype=np.float64), (G,)).copy()
        v_mu, v_lambda = np.zeros((G, max(k, 1))), self._v_table(self.v_lambda).copy()
        a0, g0, b0, m0 = self.alpha_0, self.gamma_0, self.beta_0, self.mu_0
        h.als_begin(0)
        for i in range(self.num_iter):
            if self.do_multilevel:
                sum_e2, _, mom = h.als_moments()               # mom[1 + k][G][{sum, sum of squares}]
                alpha = rng.gamma((a0 + N) / 2.0) / ((g0 + sum_e2) / 2.0)                        # draw_alpha :911-922
                if self.fm.k1:
                    sw, sw2 = mom[0, :, 0], mom[0, :, 1]
                    gam = b0 * (w_mu - m0) ** 2 + g0 + (sw2 - 2 * w_mu * sw + n_g * w_mu * w_mu)   # draw_w_lambda :985-997
                    w_lambda = _keep_finite(rng.gamma((a0 + n_g + 1) / 2.0) / (gam / 2.0), w_lambda)
                    mean = (sw + b0 * m0) / (n_g + b0)                                            # draw_w_mu :946-959
                    w_mu = _keep_finite(mean + rng.standard_normal(G) * np.sqrt(1.0 / ((n_g + b0) * w_lambda)), w_mu)
                if k > 0:
                    sv, sv2 = mom[1:, :, 0].T, mom[1:, :, 1].T                                    # [G][k]
                    ng = n_g[:, None]
                    gam = b0 * (v_mu - m0) ** 2 + g0 + (sv2 - 2 * v_mu * sv + ng * v_mu ** 2)     # draw_v_lambda :1061-1075
                    v_lambda = _keep_finite(rng.gamma(np.broadcast_to((a0 + ng + 1) / 2.0, gam.shape)) / (gam / 2.0), v_lambda)
                    mean = (sv + b0 * m0) / (ng + b0)                                             # draw_v_mu :1019-1037
                    v_mu = _keep_finite(mean + rng.standard_normal(mean.shape) * np.sqrt(1.0 / ((ng + b0) * v_lambda)), v_mu)
            st = h.als_sweep(w_lambda, v_lambda, alpha, w_mu, v_mu, self.do_sample, self.seed * 7919 + 13)
            p = h.predict(1, test.num_cases)
            if self.task == TASK_REGRESSION:
                self.pred_this = p
                self.pred_sum_all += np.maximum(self.min_target, np.minimum(self.max_target, p))
                metric = float(np.sqrt(np.mean((self.pred_sum_all / (i + 1) - test.target) ** 2)))
            else:
                self.pred_this = cdf_gaussian(p)
                self.pred_sum_all += self.pred_this
                metric = float(np.mean(((self.pred_sum_all / (i + 1)) >= 0.5) == (test.target >= 0)))
            print("#Iter=%3d\tTrain=%g\tTest=%g" % (i, st.train_metric, metric), file=self.out)
            row = {"train": st.train_metric, "test": metric, "alpha": alpha, "time_learn": st.device_seconds}
            for g in range(G):                                                                    # rlog fields :1145-1157
                row["wmu[%d]" % g], row["wlambda[%d]" % g] = float(w_mu[g]), float(w_lambda[g])
            row["w_lambda"] = float(w_lambda[0])
            self.log.append(row)
        self.w_mu, self.w_lambda_last, self.v_mu, self.v_lambda_last = w_mu, w_lambda, v_mu, v_lambda
        h.als_end()
        self.fm.w0, self.fm.w, self.fm.v = h.get_params(self.fm.w, self.fm.v)


def _keep_finite(new, old):
    """the reference keeps the old value of a hyper-parameter whose draw is NaN / Inf (fm_learn_mcmc.h:961-975 etc.)"""
    new = np.asarray(new, dtype=np.float64)
    return np.where(np.isfinite(new), new, old)


class FMLearnSGDA(FMLearnSGD):
    """fm_learn_sgd_element_adapt_reg (`-method sgda`, fm_learn_sgd_element_adapt_reg.h:44-93) on the GPU: theta steps
    on the train rows alternate with lambda steps on `validation` (libfm.cpp:276-279).  gpu_batch = 0: the reference's
    strictly online order (one wavefront, parity); > 0: the batch form (fmx_sgda_epoch_minibatch)."""

    def __init__(self):
        super().__init__()
        self.gpu_batch, self.gpu_w0_chunk = 0, 0
        self.validation = None
        self.groups = None               # DataMetaInfo::attr_group (None = one group)
        self.reg_w = 0.0                 # one group: scalar / [k]; with groups: [G] / [G][k]  (:84-85)
        self.reg_v = None

    def learn(self, train, test):
        if self.validation is None:
            raise ValueError("sgda needs a validation set")              # the reference asserts (libfm.cpp:277)
        print("Training using self-adaptive-regularization SGD.", file=self.out)
        h = self._h
        st, sv = self._slot(train), self._slot(self.validation)
        self.fm.reg0 = self.fm.regw = self.fm.regv = 0.0                  # :257-259
        h.set_groups(self.groups)
        h.sgda_begin()
        for i in range(self.num_iter):
            stats = (h.sgda_epoch_minibatch(st, sv, i > 0, self.gpu_batch, self.gpu_w0_chunk) if self.gpu_batch > 0
                     else h.sgda_epoch(st, sv, i > 0))
            rmse_val = self.evaluate(self.validation)
            rmse_train = self.evaluate(train)
            rmse_test = self.evaluate(test)
            print("#Iter=%3d\tTrain=%g\tTest=%g" % (i, rmse_train, rmse_test), file=self.out)
            reg = h.sgda_get_reg()                                           # [G][1 + k]
            if h.G == 1:
                self.reg_w, self.reg_v = float(reg[0, 0]), reg[0, 1:].copy()
            else:
                self.reg_w, self.reg_v = reg[:, 0].copy(), reg[:, 1:].copy()
            row = {"rmse_train": rmse_train, "rmse_val": rmse_val, "time_learn": stats.device_seconds}
            for g in range(h.G):                                             # rlog fields :119-132
                row["regw[%d]" % g] = float(reg[g, 0])
                for f in range(self.fm.num_factor):
                    row["regv[%d,%d]" % (g, f)] = float(reg[g, 1 + f])
            self.log.append(row)
        h.sgda_end()
        self.sync_model()

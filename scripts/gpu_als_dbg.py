import io, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from common import Golden
from oracle import oracle as O
from libfm_amd import learner as L
g = Golden("als_reg_nolin_dup")
z = g.z
ent = z["train_entries"]; rp = z["train_row_ptr"].astype(np.int64)
def run(ent, rp, label, k=3):
    tr = O.Data(ent, rp, g.train_target); te = g.data(O, "test")
    m = O.Model(g.n, k, False, False, 0, 0, 20.0); m.v[:] = z["init_v"][:k]
    ref = m.copy()
    O.als_learn(ref, tr, te, 0, 1, 0.0, 20.0, g.min_target, g.max_target)
    fm = L.FMModel(); fm.num_attribute, fm.num_factor, fm.k0, fm.k1 = g.n, k, False, False
    fm.reg0, fm.regw, fm.regv = 0, 0, 20.0
    fm.w0, fm.w, fm.v = m.w0, m.w.copy(), m.v.copy()
    l = L.FMLearnALS(); l.fm, l.task, l.num_iter = fm, 0, 1
    l.min_target, l.max_target, l.w_lambda, l.v_lambda = g.min_target, g.max_target, 0.0, 20.0
    l.out = io.StringIO(); l.init()
    l.learn(L.Data(ent, rp, g.train_target), L.Data(z["test_entries"], z["test_row_ptr"], g.test_target))
    d = np.abs(l.fm.v - ref.v)
    print("%-28s max|v|=%.3f maxabs diff=%.3e per-factor max diff=%s" % (label, np.abs(ref.v).max(), d.max(), d.max(1)))
    l.close()
run(ent, rp, "with duplicates")
keep = np.ones(len(ent), bool)
for c in range(len(rp) - 1):
    seen = set()
    for i in range(rp[c], rp[c + 1]):
        if int(ent["id"][i]) in seen: keep[i] = False
        seen.add(int(ent["id"][i]))
rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
ent2 = ent[keep]; cnt = np.bincount(rows[keep], minlength=len(rp) - 1)
rp2 = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
run(ent2, rp2, "duplicates removed")
run(ent, rp, "with duplicates, k=1", k=1)

"""Command line with libFM's flags on top of the GPU learners (host side only; all arithmetic is in libfmx.so).

    python -m libfm_amd.cli -task r -train tr.libfm -test te.libfm -dim 1,1,8 -iter 20 -method sgd \
           -learn_rate 0.01 -regular 0,0,0.01 -init_stdev 0.1 -seed 42 -out pred.txt -save_model model.txt

Mirrors the driver of the reference (src/libfm/libfm.cpp:62-441): same flag names and defaults (:76-121), the flag
syntax of CMDLine (`-x value`, lists separated by ',' or ';', src/util/cmdline.h:80-105), binary-or-text data
auto-detection (Data.h:113-125), `-seed` reproducing the reference's initial model bit for bit (refrand.py),
targets rewritten to +-1 for `-task c` (:298-306), regularisation / learning-rate parsing (:326-404), the
`#Iter=...` progress lines, `-out` (:423-428), `-save_model` / `-load_model` (:262-268, :431-434), and the
reference's error convention: "ERROR: ..." on stderr and exit status 0 (:436-441).
`-meta` (attribute groups, :199-242 / Data.h:85-97) is honoured by als / mcmc / sgda, incl. `-regular 'r0,w_1..w_G,v_1..v_G'`
(:353-363).  `-relation a,b` (block structure, :172-196; als / mcmc only like the reference's learners) loads <a>.x or
<a>.xt, <a>.train, <a>.test and optional <a>.groups; the joined rows are expanded on the device.  `-cache_size` is
accepted and ignored (everything is resident).
GPU-only additions: -gpu_mode sequential|minibatch|hogwild (default minibatch), -batch, -w0_chunk, -device.
"""
import sys
import time

import numpy as np

from . import data as D
from . import learner as L
from . import refrand as R

FLAGS = {"task": "r=regression, c=binary classification [MANDATORY]", "meta": "filename for meta information about data set", "train": "filename for training data [MANDATORY]",
         "test": "filename for test data [MANDATORY]", "validation": "", "out": "filename for output",
         "dim": "'k0,k1,k2': k0=use bias, k1=use 1-way interactions, k2=dim of 2-way interactions; default=1,1,8",
         "regular": "'r0,r1,r2' for SGD and ALS", "init_stdev": "stdev for initialization of 2-way factors; default=0.1",
         "iter": "number of iterations; default=100", "learn_rate": "learn_rate for SGD", "method": "sgd, als, mcmc; default=mcmc",
         "verbosity": "", "rlog": "write measurements within iterations to a file", "seed": "integer value", "help": "",
         "relation": "BS: filenames for the relations, default=''", "cache_size": "", "save_model": "filename for writing the FM model",
         "load_model": "filename for reading the FM model",
         "gpu_mode": "sequential | minibatch | hogwild (default minibatch)", "batch": "", "w0_chunk": "", "device": ""}


def parse(argv):
    """CMDLine (cmdline.h:80-105): a flag is '-name' or '--name'; its value is the next token unless that starts with '-'."""
    vals, i = {}, 0
    while i < len(argv):
        a = argv[i]
        if not a.startswith("-"):
            raise ValueError("cannot parse parameter \"%s\"" % a)
        name = a.lstrip("-")
        if i + 1 < len(argv) and not argv[i + 1].startswith("-"):
            vals[name] = argv[i + 1]
            i += 2
        else:
            vals[name] = ""
            i += 1
    for k in vals:
        if k not in FLAGS:
            raise ValueError("the parameter " + k + " does not exist")      # cmdline.h:150-157
    return vals


def split_list(s):
    return [t for t in s.replace(";", ",").split(",") if t != ""]


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    try:
        return _main(argv)
    except (ValueError, OSError, RuntimeError) as e:
        print("\nERROR: %s" % e, file=sys.stderr)
        return 0                                                     # the reference exits 0 on errors (libfm.cpp:436-441)


def _main(argv):
    print("----------------------------------------------------------------------------")
    print("libfm_amd: MI355X-native FM learners behind libFM's command line")
    print("----------------------------------------------------------------------------")
    if not argv or "-help" in argv or "--help" in argv:
        for k, v in FLAGS.items():
            print("-%-14s %s" % (k, v))
        return 0
    a = parse(argv)
    seed = int(a["seed"]) if "seed" in a else int(time.time())
    method = a.get("method", "mcmc")
    init_stdev = float(a.get("init_stdev", "0.1"))
    dim = [int(x) for x in split_list(a.get("dim", "1,1,8"))]
    if len(dim) != 3:
        raise ValueError("-dim needs k0,k1,k2")
    if method == "mcmc" and ("save_model" in a or "load_model" in a):
        print("WARNING: -save_model / -load_model enabled only for SGD and ALS.")      # libfm.cpp:123-133
        return 0
    if method not in ("sgd", "sgda", "als", "mcmc"):
        raise ValueError("unknown method")
    for need in ("task", "train", "test"):
        if need not in a:
            raise ValueError("-%s is mandatory" % need)

    print("Loading train...\t")
    train = L.Data(*D.load(a["train"]))
    print("num_rows=%d\tnum_values=%d\tnum_features=%d\tmin_target=%g\tmax_target=%g" %
          (train.num_cases, len(train.entries), train.num_feature, train.min_target, train.max_target))
    print("Loading test... \t")
    test = L.Data(*D.load(a["test"]))
    print("num_rows=%d\tnum_values=%d\tnum_features=%d\tmin_target=%g\tmax_target=%g" %
          (test.num_cases, len(test.entries), test.num_feature, test.min_target, test.max_target))

    validation = None
    if method == "sgda":                                                             # libfm.cpp:173-194
        if "validation" not in a:
            raise ValueError("sgda needs -validation")
        print("Loading validation set...\t")
        validation = L.Data(*D.load(a["validation"]))

    fm = L.FMModel()
    fm.num_attribute = max(train.num_feature, test.num_feature)                      # libfm.cpp:203-206
    if validation is not None:
        fm.num_attribute = max(fm.num_attribute, validation.num_feature)
    # (1.2) relations (libfm.cpp:172-196): block attributes follow the main attributes (:213-216)
    relations = []
    if a.get("relation"):
        if method not in ("als", "mcmc"):
            raise ValueError("relations are not supported with SGD")            # fm_learn_sgd.h:61-63
        rel_names = split_list(a["relation"])
        print("#relations: %d" % len(rel_names))
        for name in rel_names:
            r = D.read_relation(name)
            print("num_cases=%d\tnum_values=%d\tnum_features=%d" % (r.num_cases, len(r.entries), r.num_feature))
            train.add_relation(r, D.read_row_mapping(name + ".train", train.num_cases), fm.num_attribute)
            test.add_relation(r, D.read_row_mapping(name + ".test", test.num_cases), fm.num_attribute)
            relations.append(r)
            fm.num_attribute += r.num_feature
    num_main_attribute = fm.num_attribute - sum(r.num_feature for r in relations)

    fm.k0, fm.k1, fm.num_factor = dim[0] != 0, dim[1] != 0, dim[2]
    fm.init_stdev = init_stdev
    R.srand(seed)                                                                     # libfm.cpp:115-116
    fm.w0 = 0.0
    fm.w = np.zeros(fm.num_attribute)
    fm.v = R.init_v(fm.num_factor, fm.num_attribute, fm.init_mean, fm.init_stdev)    # fm_model.h:96
    if "load_model" in a:
        print("Reading FM model... \t")
        if not fm.load_model(a["load_model"]):
            print("WARNING: malformed model file. Nothing will be loaded.")
            fm.w0, fm.w[:] = 0.0, 0.0

    # (1.3) meta data: attribute -> group, one id per line (DataMetaInfo::loadGroupsFromFile, Data.h:85-97;
    # DVector::load reads num_attribute values, missing ones stay 0, matrix.h:360-371)
    groups, num_groups = None, 1
    if a.get("meta") or any(r.groups is not None for r in relations):
        print("Loading meta data...\t")
        groups = np.zeros(fm.num_attribute, dtype=np.uint32)
        if a.get("meta"):
            with open(a["meta"]) as f:
                vals = f.read().split()[:num_main_attribute]
            groups[:len(vals)] = [int(x) for x in vals]
        num_groups = int(groups[:num_main_attribute].max()) + 1 if num_main_attribute else 1
        at = num_main_attribute
        for r in relations:                                                          # the joined table, :217-240
            rg = r.groups if r.groups is not None else np.zeros(r.num_feature, dtype=np.uint32)
            groups[at:at + r.num_feature] = num_groups + rg
            num_groups += int(rg.max()) + 1 if r.num_feature else 1
            at += r.num_feature
        print("#attr=%d\t#groups=%d" % (fm.num_attribute, num_groups))

    task = a["task"]
    if task not in ("r", "c"):
        raise ValueError("unknown task")
    min_t, max_t = train.min_target, train.max_target                                # libfm.cpp:295-296
    if task == "c":                                                                  # libfm.cpp:302-306
        train.target[:] = np.where(train.target <= 0.0, -1.0, 1.0)
        test.target[:] = np.where(test.target <= 0.0, -1.0, 1.0)
    reg = [float(x) for x in split_list(a.get("regular", ""))]
    if len(reg) == 0:
        reg = [0.0, 0.0, 0.0]
    elif len(reg) == 1:
        reg = [reg[0]] * 3
    group_reg = None
    if len(reg) == 1 + 2 * num_groups and len(reg) != 3 and method in ("als", "mcmc"):      # libfm.cpp:353-363
        group_reg = (np.array(reg[1:1 + num_groups]), np.array(reg[1 + num_groups:]))
        reg = [reg[0], 0.0, 0.0]
    elif len(reg) != 3:
        raise ValueError("-regular needs 0, 1, 3 or 1+2*#groups values")
    fm.reg0, fm.regw, fm.regv = reg
    num_iter = int(a.get("iter", "100"))

    if method in ("sgd", "sgda"):
        l = L.FMLearnSGD() if method == "sgd" else L.FMLearnSGDA()
        if method == "sgda":
            l.validation = validation
            l.groups = groups
            if task == "c":
                l.validation.target[:] = np.where(l.validation.target <= 0.0, -1.0, 1.0)
        lrs = [float(x) for x in split_list(a.get("learn_rate", ""))]
        if len(lrs) not in (1, 3):
            raise ValueError("-learn_rate needs 1 or 3 values")                     # the reference asserts (libfm.cpp:391-392)
        l.learn_rate = lrs[0] if len(lrs) == 1 else 0.0                              # 3 values: scalar rate 0 (libfm.cpp:396-401)
        l.mode = a.get("gpu_mode", "minibatch")
        l.batch = int(a.get("batch", "0"))
        l.w0_chunk = int(a.get("w0_chunk", "0"))
    else:
        if method == "als":
            l = L.FMLearnALS()
        else:
            l = L.FMLearnMCMC()
            l.seed = seed
        fm.w = R.init_w_normal(fm.num_attribute, fm.init_mean, fm.init_stdev)        # libfm.cpp:283
        l.w_lambda, l.v_lambda = fm.regw, fm.regv
        l.groups = groups
        if group_reg is not None:
            l.w_lambda, l.v_lambda = group_reg
    l.fm, l.num_iter, l.task = fm, num_iter, (0 if task == "r" else 1)
    l.min_target, l.max_target = min_t, max_t
    l.device = int(a.get("device", "-1"))
    l.init()
    l.learn(train, test)
    if method in ("sgd", "sgda"):
        print("Final\tTrain=%g\tTest=%g" % (l.evaluate(train), l.evaluate(test)))   # libfm.cpp:418-420
    if "rlog" in a and a["rlog"]:
        with open(a["rlog"], "w") as f:                                              # rlog.h:60-103: TSV with a header
            keys = sorted(l.log[0].keys()) if l.log else []
            f.write("\t".join(keys) + "\n")
            for row in l.log:
                f.write("\t".join("%g" % row[k] for k in keys) + "\n")
    if "out" in a and a["out"]:
        pred = l.predict(test)
        with open(a["out"], "w") as f:                                               # DVector::save, matrix.h:332-342
            f.write("".join("%g\n" % p for p in pred))
    if "save_model" in a and a["save_model"]:
        print("Writing FM model to " + a["save_model"])
        fm.save_model(a["save_model"])
    l.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""bench.py -- SGD training throughput of the FM hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (north star): n = 1e8 features, k = 64, 32 nnz/row, one-hot field rows with uniform
ids, values 1.0, +-1 labels, task = classification, lr 0.01, regular 0,0,0.001; synthetic rows are generated ON
the device (fmx_synth_rows) and the parameters are filled on the device (fmx_init_params), so everything is
resident in HBM before the timed region.  A "step" is one SGD pass over `--rows` examples.

The update rule is the SAME at every N: the restated minibatch rule of oracle/fm_oracle.h (batch 262 144, bias lag),
the rule tests/test_gpu_parity.py and tests/test_gpu_fullsize.py hold to the oracle at 1e-4.
N = 1 : FMX_APPLY_FUSED -- the rule in ONE pass over HBM (features that occur once in their batch are gathered, used and
        written back by their example's wavefront; the few that occur more than once go through the segmented kernel);
        the whole pass runs inside the library (fmx_sgd_epoch).
N > 1 : V/w are row-sharded by feature id; every rank sees every example restricted to its own features; per
        minibatch ONE all-reduce (RCCL) of the [B][k+1] partial sums, then every rank updates its shard
        (fmx_sgd_partial -> all_reduce -> fmx_sgd_finish).  Total work is fixed => "strong".
`--mode hogwild` (asynchronous, metric-level parity only) is printed as an extra key, never as `value`.

The JSON line carries `roofline` (dominant kernel: algorithmic bytes / HIP-event duration vs the 8 TB/s HBM peak, and
the V-gather read fraction next to the total) and, at N = 1, `cpu_baseline` (the REAL reference's fm_model::predict +
fm_SGD compiled from its own sources, one thread, bounded sample; the C restatement is the extra key `cpu_port`).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s copy ceiling)


def algorithmic_bytes(k, nnz, kind):
    """SURVEY section 8(d), fp32 parameters, u32 ids, f32 values."""
    read_step = nnz * (4 * k + 12) + 4          # ids+values, w, V rows, target
    write_step = nnz * (4 * k + 4)              # w, V rows
    if kind == "fused":                         # whole training step in one kernel
        return read_step + write_step
    if kind == "apply":                         # k_apply: ids+values, w, V, the example's S row and multiplier; writes w, V
        return nnz * (4 * k + 12) + 4 * k + 4 + write_step
    if kind == "rowsums":                       # k_rowsums: predict-side reads, writes S row + scalar
        return nnz * (4 * k + 12) + 4 * k + 4
    raise ValueError(kind)


def mem_available_bytes():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def cpu_baseline(n, k, nnz, rows):
    """oracle leg: the restated reference loop (fm_model::predict + fm_SGD, fp64, factor-major V) on one core."""
    from oracle import oracle as O
    n_cpu = n
    need = lambda nn: nn * k * 8 + nn * 8
    avail = mem_available_bytes()
    while avail and need(n_cpu) * 1.25 > avail and n_cpu > 10 ** 6:
        n_cpu //= 2
    cores = os.cpu_count() or 1
    t0 = time.time()
    sec, eps = O.time_sgd_synth(n_cpu, k, nnz, rows, seed=123, threads=min(cores, 64))
    setup = time.time() - t0 - sec
    sample = "%d rows of the synthetic workload at n=%d k=%d nnz=%d, fp64 reference layout, 1 epoch" % (rows, n_cpu, k, nnz)
    if n_cpu != n:
        sample += " (n reduced from %d: host RAM)" % n
    return {"value": round(eps, 1), "unit": "examples/s", "cores": 1, "kind": "port", "sample": sample,
            "seconds": round(sec, 3), "setup_seconds": round(setup, 1), "host_cores": cores}


def cpu_baseline_rows(h, n, k, rows):
    """cpu_baseline for a workload the host generators do not have (Criteo-shaped): the first `rows` rows are copied back from the
    device and the restated reference loop (fm_model::predict + fm_SGD, fp64, factor-major V; oracle/fm_oracle.c) runs over them
    once on one core."""
    import ctypes as C
    import numpy as np
    from oracle import oracle as O
    ent, rp, y = h.download_rows(0)
    rows = min(rows, len(y))
    d = O.Data(ent[:int(rp[rows])], rp[:rows + 1], y[:rows])
    avail, n_cpu = mem_available_bytes(), n
    if avail and (n * k * 8 + n * 8) * 1.25 > avail:
        return {"error": "host RAM too small for the fp64 model of this workload"}
    m = O.Model.__new__(O.Model)
    m.n, m.k, m.k0, m.k1, m.reg0, m.regw, m.regv, m.w0 = int(n), int(k), True, True, 0.0, 0.0, 0.001, 0.0
    m.w = np.empty(m.n, dtype=np.float64)
    m.v = np.empty((m.k, m.n), dtype=np.float64)
    cm = m._c()
    O.lib().fmo_fill_params(C.byref(cm), 1, 0.01, min(os.cpu_count() or 1, 64))
    t0 = time.time()
    O.sgd_epoch_online(m, d, 1, 0.01, -1.0, 1.0)
    sec = time.time() - t0
    return {"value": round(rows / sec, 1), "unit": "examples/s", "cores": 1, "kind": "port",
            "sample": "the first %d rows of the step's Criteo-shaped rows (copied back from the device), restated reference loop, "
                      "n=%d k=%d, fp64 reference layout, 1 epoch" % (rows, n, k), "seconds": round(sec, 3), "host_cores": os.cpu_count() or 1}


def v_read_fraction(value, k, nnz, world=1):
    """SURVEY section 8(d) fraction (1): examples/s x z*k*4 bytes of gathered V rows per GPU / HBM peak."""
    return round(value * nnz * k * 4 / world / 1e9 / HBM_PEAK_GBS, 4)


def cpu_reference(n, k, nnz, rows):
    """the reference's own fm_model::predict + fm_SGD (compiled from /root/reference into oracle/_ref/ref_harness),
    one thread.  The stock containers overflow at k*n >= 2^32 (matrix.h:167-169), so n is capped accordingly."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.path.exists(exe):
        return None
    n_ref = min(n, (2 ** 32 - 1) // k // 1_000_000 * 1_000_000)
    r = subprocess.run([exe, "time_sgd", str(n_ref), str(k), str(nnz), str(rows), "123"], capture_output=True, text=True)
    if r.returncode != 0:
        return {"error": r.stderr[-200:]}
    d = json.loads(r.stdout.strip().splitlines()[-1])
    return {"value": round(d["examples_per_sec"], 1), "unit": "examples/s", "cores": 1, "kind": "reference",
            "sample": "%d rows of the synthetic workload through the reference's own fm_model::predict + fm_SGD (oracle/_ref/ref_harness "
                      "time_sgd), n=%d (largest k*n the stock containers allocate, matrix.h:167-169; bench n=%d) k=%d nnz=%d, 1 epoch"
                      % (rows, n_ref, n, k, nnz), "seconds": round(d["seconds"], 3), "host_cores": os.cpu_count() or 1}


def cpu_reference_rows(h, n, k, rows, lr=0.01, regv=0.001, stdev=0.01):
    """cpu_baseline for rows only the device generator has (Criteo-shaped, BASELINE configs[2]): the first `rows` rows of slot 0 are
    copied back and the REAL reference's fm_model::predict + fm_SGD (oracle/_ref/ref_harness time_sgd_rows) runs over them once on one
    core.  Needs k * n < 2^32 (matrix.h:167-169): 3.3e7 x 64 fits."""
    import subprocess
    import tempfile
    import numpy as np
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.path.exists(exe) or k * n >= 2 ** 32:
        return None
    avail = mem_available_bytes()
    if avail and (n * k * 8 + n * 8) * 1.2 > avail:
        return {"error": "host RAM too small for the reference's fp64 model of this workload"}
    ent, rp, y = h.download_rows(0)
    rows = min(rows, len(y))
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "rows.bin")
        with open(path, "wb") as f:
            np.array([rows, 0], dtype=np.uint32).tofile(f)
            np.array([int(rp[rows])], dtype=np.uint64).tofile(f)
            rp[:rows + 1].astype(np.uint64).tofile(f)
            ent[:int(rp[rows])].tofile(f)
            y[:rows].astype(np.float32).tofile(f)
        r = subprocess.run([exe, "time_sgd_rows", path, str(n), str(k), repr(lr), repr(regv), repr(stdev)], capture_output=True, text=True)
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-200:]}
    d = json.loads(r.stdout.strip().splitlines()[-1])
    return {"value": round(d["examples_per_sec"], 1), "unit": "examples/s", "cores": 1, "kind": "reference",
            "sample": "the first %d of the step's rows (copied back from the device) through the reference's own fm_model::predict + fm_SGD "
                      "(oracle/_ref/ref_harness time_sgd_rows), n=%d k=%d, 1 epoch" % (rows, n, k),
            "seconds": round(d["seconds"], 3), "host_cores": os.cpu_count() or 1}


def run_sgd_config(capi, name, n, k, nnz, rows, criteo, steps, warmup, with_cpu, cpu_rows):
    """a BASELINE config other than the headline's on one GPU, same one-pass batch rule, the library's batch (fmx_sgd_opts::batch = 0):
    one step = one epoch over `rows` rows; returns the figure as a dict (an extra key of the line, never `value`)"""
    lr, regv = 0.01, 0.001
    h = capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, regv, lr, -1.0, 1.0, device=0)
    h.init_params(0.0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, nnz, capi.SYNTH_CRITEO if criteo else capi.SYNTH_UNIFORM)
    cpu = None
    if with_cpu:
        cpu = cpu_reference_rows(h, n, k, cpu_rows, lr, regv) if criteo else cpu_reference(n, k, nnz, cpu_rows)
    setup = 0.0
    for _ in range(warmup):
        setup += h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, 0, 2).setup_seconds
    h.synchronize()
    dev, launches, deferred, st = 0.0, 0, 0, None
    t0 = time.perf_counter()
    for _ in range(steps):
        st = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 0, 0, capi.FLAG_TIME_MAIN_KERNEL, 2)
        dev += st.main_kernel_seconds
        launches += st.main_kernel_launches
        deferred += st.deferred_features
    h.synchronize()
    elapsed = time.perf_counter() - t0
    value = steps * rows / elapsed
    per_ex = algorithmic_bytes(k, nnz, "fused")
    per_launch = min(int(st.batch_used), rows)
    avg = dev / max(launches, 1)
    achieved = per_ex * per_launch / avg / 1e9
    info = h.info()
    traffic, tsrc = None, None
    try:                                  # counter bytes of the dominant kernel from the committed PMC passes, if they are for this shape
        for e in json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).values():
            if (e.get("n"), e.get("k"), e.get("nnz"), e.get("examples_per_launch")) == (n, k, nnz, per_launch):
                traffic, tsrc = e["hbm_bytes_per_launch"] + e.get("deferred_pass_bytes_per_launch", 0), "committed profile: " + e["source"]
    except (OSError, ValueError, KeyError):
        pass
    out = {"metric": "SGD training examples/sec at k=%d, nnz=%d, %.1e feat (%s)" % (k, nnz, n, name),
           "value": round(value, 1), "unit": "examples/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": round(elapsed / steps * 1e3, 3), "dtype": "f32", "data": "synthetic",
           "config": {"workload": ("Criteo-shaped (13 fields of <= 100 ids + %d Zipf(1.05) fields)" % (nnz - 13) if criteo else "synthetic one-hot fields")
                                  + " n=%d k=%d nnz=%d, %d examples/step, task=c lr=%g regv=%g" % (n, k, nnz, rows, lr, regv),
                      "mode": "fused", "bias_lag": 2,
                      "batch_rule": {"batch": int(st.batch_used), "collision_mass": round(st.collision_mass, 6), "gain": round(st.batch_gain, 4),
                                     "cut": bool(st.status & capi.STAT_BATCH_CUT)},
                      "device": info.device_name.decode()},
           "roofline": {"bound": "hbm", "kernel": ("k_small_one<%d>: examples + deferred features + bias recurrence in ONE launch per batch" if st.status & capi.STAT_SMALL_ONE
                                                   else "k_fused<%d,EXACT> + deferred features, per batch") % info.k_padded,
                        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "v_read_frac": v_read_fraction(per_launch / avg, k, nnz), "traffic": traffic, "traffic_source": tsrc,
                        "bytes_per_example": per_ex, "examples_per_launch": per_launch, "avg_launch_ms": round(avg * 1e3, 4),
                        "launches": launches, "deferred_features_per_example": round(deferred / (steps * rows), 4),
                        "launch_time": "epoch HIP-event time / batches"},
           "one_time_setup_seconds": round(setup, 4),
           "cpu_baseline": cpu}
    h.close()
    return out


def shard_probe(capi, n, k, nnz, single_gpu_value, rows=1 << 21, batch=262144, worlds=(2, 4, 8), lag=2):
    """what ONE rank of a P-way feature-sharded step computes, measured on this GPU: rank 0 of P holds n / P features and sees every example
    restricted to them (about nnz / P entries per example); per batch its sums (fmx_sgd_partial), then -- the exchange left out -- its
    update (fmx_sgd_finish: multipliers, bias recurrence on the side stream, write-back).  The library's own kernels and schedule (bias lag
    `lag`, default micro-chunk and batch).  The WIRE is not measured (one GPU): `modelled` holds the arithmetic of DESIGN.md section 6 -- a
    direct reduce-scatter + all-gather of the [batch][k + 1] fp32 sums over the P - 1 xGMI links of a GPU at 58 GB/s per link and
    direction -- and what a step would then take; every figure under `modelled` is arithmetic, not a measurement."""
    import torch
    out = {"what": "per-rank compute of the feature-sharded step on ONE GPU (rank 0 of P, no exchange); wire and speed-up are MODELLED",
           "batch": batch, "bias_lag": lag, "rows": rows, "ranks": {}}
    for world in worlds:
        h = capi.Handle(n, k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, device=0, shard_rank=0, shard_world=world,
                        shard_hash=1)
        try:
            h.init_params(0.0, 0.01, 1)
            h.synth_rows(0, 123, 0, rows, nnz)
            kp1 = h.info().k_padded + 1
            st = torch.cuda.Stream()
            bufs = [torch.empty(batch * kp1, dtype=torch.float32, device="cuda") for _ in range(2)]

            def epoch(which):
                for i, row0 in enumerate(range(0, rows, batch)):
                    nb = min(batch, rows - row0)
                    buf = bufs[i & 1]
                    if which != "finish":
                        h.sgd_partial(0, row0, nb, buf.data_ptr(), st.cuda_stream)
                    if which != "partial":
                        h.sgd_finish(0, row0, nb, buf.data_ptr(), capi.APPLY_DEFAULT, 0, st.cuda_stream, batch, capi.FLAG_BIAS_LAG, lag)
            res = {}
            for which in ("both", "partial", "finish"):
                epoch(which); st.synchronize(); h.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    epoch(which)
                st.synchronize(); h.synchronize()
                res[which] = (time.perf_counter() - t0) / 2 / ((rows + batch - 1) // batch)       # seconds per batch
            payload = batch * kp1 * 4
            wire = (2.0 / world) * payload / 58e9
            exact = max(res["partial"], wire) + res["finish"]
            piped = max(res["partial"] + res["finish"], wire)
            out["ranks"]["P%d" % world] = {
                "per_rank_examples_per_s": round(batch / res["both"], 1), "sums_ms_per_batch": round(res["partial"] * 1e3, 4),
                "update_ms_per_batch": round(res["finish"] * 1e3, 4), "step_ms_per_batch": round(res["both"] * 1e3, 4),
                "frac_of_hbm_peak": round(batch / res["both"] * (algorithmic_bytes(k, nnz, "rowsums") + algorithmic_bytes(k, nnz, "apply")) / world / 1e9 / HBM_PEAK_GBS, 4),
                "modelled": {"wire_ms_per_batch": round(wire * 1e3, 4), "exact_rule_examples_per_s": round(batch / exact, 1),
                             "one_batch_stale_examples_per_s": round(batch / piped, 1),
                             "speedup_vs_1_gpu_exact": round(batch / exact / single_gpu_value, 3) if single_gpu_value else None,
                             "speedup_vs_1_gpu_one_batch_stale": round(batch / piped / single_gpu_value, 3) if single_gpu_value else None}}
        finally:
            h.close()
    return out


def committed_traffic(kernel, examples_per_launch, k, nnz):
    """HBM bytes per launch from the committed PMC profile (profiles/traffic.json), if it is for this kernel/shape."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        e = t.get(kernel)
        if e and e["examples_per_launch"] == examples_per_launch and e["k"] == k and e["nnz"] == nnz:
            return e["hbm_bytes_per_launch"], "committed profile: " + e.get("source", "profiles/traffic.json")
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def live_traffic(args, examples_per_launch):
    """HBM bytes per launch of the headline's kernels MEASURED NOW: two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE: one counter per pass, as
    MI355X_MICROARCH.md prescribes) over a short run of this script in a child process, corrected as profiles/r06_pmc_summary.txt does it:
    reads = 2 x FETCH_SIZE x 1024 (128-byte requests are tallied at 64 B) - the 4-byte w gathers, which really are 64-byte requests
    (profiles/r02_w_gather.txt); writes = WRITE_SIZE x 1024.  Returns (k_fused bytes, deferred-pass bytes (upper bound), text) or None."""
    import csv, glob, shutil, subprocess, tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    # not from inside a profiled process (a profiler's environment would be inherited by the child's own profiler): the committed counters then
    if any(kk.startswith(("ROCP_", "ROCPROF", "ROCTRACER", "HSA_TOOLS")) for kk in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="fmx_pmc_", dir="/tmp")
        try:
            cmd = [prof, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--no-extras", "--no-cpu-baseline", "--steps", "3", "--warmup", "1", "--features", str(args.n), "--factors", str(args.k), "--nnz", str(args.nnz),
                   "--rows", str(args.rows), "--batch", str(args.batch), "--bias-lag", str(args.bias_lag)]
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=120, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None
            agg = {}
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    name = row.get("Kernel_Name", "")
                    key = "fused" if "k_fused<" in name else ("seg" if "k_apply_seg<" in name else None)
                    if key and row.get("Counter_Name") == ctr:
                        a = agg.setdefault(key, [0, 0.0])
                        a[0] += 1
                        a[1] += float(row.get("Counter_Value", 0))
            if "fused" not in agg:
                return None
            got[ctr] = {kk: v[1] / v[0] for kk, v in agg.items()}        # KB per launch
        except (OSError, subprocess.SubprocessError, ValueError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fused = 2 * got["FETCH_SIZE"]["fused"] * 1024 - examples_per_launch * args.nnz * 64 + got["WRITE_SIZE"]["fused"] * 1024
    seg = 2 * got["FETCH_SIZE"].get("seg", 0.0) * 1024 + got["WRITE_SIZE"].get("seg", 0.0) * 1024
    return int(fused), int(seg), ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, over `bench.py --no-extras --steps 3` in a child process "
                                  "(k_fused %.0f / %.0f KB per launch; reads = 2 x FETCH_SIZE - the 64-byte w requests, profiles/r06_pmc_summary.txt)"
                                  % (got["FETCH_SIZE"]["fused"], got["WRITE_SIZE"]["fused"]))


def als_bytes_per_sweep(n_rows, nnz_total, n_seen, k):
    """algorithmic bytes of one fm_learn_mcmc sweep as the device runs it: per coordinate family
    (w and each of the k factors) the column pass touches every entry twice (8-B entry + 16-B {e,q} gather for the sums,
    the same + a 16-B store for the update) = 64 B per entry, + 8 B per feature (read + write of the coordinate);
    per factor one pass that installs q_f next to e (8 + 16 + 16 B per row); the re-prediction reads every entry with its
    parameter row (8 + 4k + 4 B) and writes k q values + e per row (8k + 16 B); the target pass 36 B per row."""
    fam = (k + 1) * (nnz_total * 64 + n_seen * 8)
    loadq = k * n_rows * 40
    eterms = nnz_total * (4 * k + 12) + n_rows * (8 * k + 16)
    return fam + loadq + eterms + n_rows * 36


def cpu_reference_mcmc(method, k, nnz, rows=150000, n=1500000, gpu_leg="n=1e7 and 100x the rows"):
    """cpu_baseline of the als / mcmc figures: the REAL reference's fm_learn_mcmc::learn (src/libfm/src/fm_learn_mcmc.h:430-641,
    1160-1201 through fm_learn_mcmc_simultaneous.h:56-270; oracle/_ref/ref_harness als | mcmc), one thread, ONE iteration on a
    bounded sample of the same synthetic rows (fewer rows and features than the GPU leg: the reference needs ~27 ns per entry and
    factor)."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.path.exists(exe):
        return None
    from oracle import oracle as O
    d = O.synth_rows(123, 0, rows, nnz, n)
    with tempfile.TemporaryDirectory() as td:
        trf, tef = os.path.join(td, "train.libfm"), os.path.join(td, "test.libfm")
        d.write_libsvm(trf)
        O.Data(d.entries[:nnz * 100], d.row_ptr[:101], d.target[:100]).write_libsvm(tef)
        # the last id must appear so that the reference sizes the model like the GPU leg's sample (num_feature = max id + 1)
        with open(trf, "a") as f:
            f.write("1 %d:1\n" % (n - 1))
        cfg = [exe, method, trf, tef, "r", "1", "1", str(k), "1"] + (["0", "1", "10"] if method == "als" else []) + ["0.01", "1", os.path.join(td, "o")]
        r = subprocess.run(cfg, capture_output=True, text=True)
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-200:]}
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{\"learn_seconds\"")]
    if not line:
        return {"error": "no timing line"}
    t = json.loads(line[-1])
    return {"value": round(t["rows"] * t["iters"] / t["learn_seconds"], 1), "unit": "examples/s", "cores": 1, "kind": "reference",
            "sample": "1 iteration of the reference's fm_learn_mcmc (%s) over %d synthetic rows, n=%d k=%d nnz=%d (oracle/_ref/ref_harness %s); "
                      "the GPU leg runs %s" % ("do_sample" if method == "mcmc" else "ALS", t["rows"], n, k, nnz, method, gpu_leg),
            "seconds": round(t["learn_seconds"], 3), "host_cores": os.cpu_count() or 1}


def run_als(capi, method, n, k, nnz, rows, steps, warmup, with_cpu, cpu_kw=None):
    """one step = one sweep (fmx_als_sweep) over `rows` examples, one GPU; returns the figure as a dict"""
    sample = method == "mcmc"
    h = capi.Handle(n, k, True, True, capi.TASK_REGRESSION, 0.0, 1.0, 10.0, 0.0, -1.0, 1.0, device=0)
    h.init_params(0.0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, nnz)
    info = h.info()
    h.als_begin(0)
    for i in range(warmup):
        h.als_sweep(1.0, 10.0, do_sample=sample, seed=i)
    h.synchronize()
    t0 = time.perf_counter()
    dev = 0.0
    for i in range(steps):
        st = h.als_sweep(1.0, 10.0, do_sample=sample, seed=100 + i)
        dev += st.device_seconds
    h.synchronize()
    elapsed = time.perf_counter() - t0
    nnz_total = rows * nnz
    n_seen = min(n, nnz_total)
    per_sweep = als_bytes_per_sweep(rows, nnz_total, n_seen, k)
    achieved = per_sweep / (dev / steps) / 1e9
    traffic, tsrc, req = None, None, None
    try:                                  # counter bytes / requests of the sweep's two kernels: the committed PMC passes, if they are for this shape
        ta = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["als_sweep"]
        if ta["shape"] == {"n": n, "k": k, "nnz": nnz, "rows": rows}:
            traffic, tsrc = ta["hbm_bytes_per_sweep"], "committed profile: " + ta["source"]
            req = {"k_als_draw_fabric_read_requests_per_s": ta["per_launch"]["k_als_draw<true,4>"]["fabric_read_requests_per_s"],
                   "random_request_ceiling_per_s": "52e9 .. 55e9 (scripts/ubench/w_gather)"}
    except (OSError, ValueError, KeyError):
        pass
    out = {"metric": "%s (fm_learn_mcmc%s) training examples/sec per sweep at k=%d, nnz=%d, %.0e feat"
                     % (method.upper(), ", do_sample" if sample else "", k, nnz, n),
           "value": round(steps * rows / elapsed, 1), "unit": "examples/s", "n_gpus": 1, "steps": steps,
           "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64 caches / f32 parameters", "data": "synthetic",
           "config": {"workload": "synthetic one-hot fields n=%d k=%d nnz=%d, %d examples/sweep, task=r, lambda_w=1 lambda_v=10"
                                  % (n, k, nnz, rows),
                      "method": method, "levels": st.levels, "device": info.device_name.decode(), "arch": info.arch.decode()},
           "roofline": {"bound": "hbm", "kernel": "k_als_draw<v> + k_als_rows<v> (80 % of the sweep) + re-prediction; whole sweep",
                        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": traffic, "traffic_source": tsrc, "requests": req, "bytes_per_sweep": per_sweep, "avg_sweep_ms": round(dev / steps * 1e3, 3),
                        "note": "bound by the fabric's random-request rate, not by bytes: the column sums gather one 16-byte {e,q} "
                                "per entry at 48 G requests/s (55 G/s is what a random 4..16-byte read gets on this part, "
                                "scripts/ubench/w_gather); the update runs as a row-ordered stream (DESIGN.md section 4b)"},
           "cpu_baseline": cpu_reference_mcmc(method, k, nnz, **(cpu_kw or {})) if with_cpu else None}
    h.als_end()
    h.close()
    return out


def bench_als(args, capi):
    """--method als | mcmc: one step = one sweep (fmx_als_sweep) over `--rows` examples, one GPU"""
    if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
        raise SystemExit("--method als/mcmc: one GPU (feature shards of the sweep go through fmx_group_*, see tests/test_gpu_group.py)")
    if capi.load().fmx_device_count() == 0:
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    out = run_als(capi, args.method, args.n, args.k, args.nnz, args.rows, args.steps, args.warmup, not args.no_cpu_baseline)
    print(json.dumps(out), flush=True)


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_group(args, capi, criteo):
    """--gpus N from ONE process (the shape libFM has, libfm.cpp:271-293,415): N feature shards, one handle each, tied together by
    fmx_group_create -- RCCL between distinct devices, the loopback reduction when they share one (--same-device: the N > 1 code
    path on a one-GPU box) -- and driven by fmx_group_sgd_epoch.  The rule is the one N = 1 runs."""
    N = args.gpus
    ndev = capi.load().fmx_device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    if not args.same_device and ndev < N:
        raise SystemExit("--gpus %d but %d HIP device(s) visible (--same-device puts every shard on device 0)" % (N, ndev))
    lr, regv = 0.01, 0.001
    hs = [capi.Handle(args.n, args.k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, regv, lr, -1.0, 1.0,
                      device=0 if args.same_device else r, shard_rank=r, shard_world=N, shard_hash=1, place_candidates=args.place,
                      exchange_algo=1 if args.exchange == "rsag" else 0)
          for r in range(N)]
    for h in hs:
        h.init_params(0.0, 0.01, 1)
        h.synth_rows(0, 123, 0, args.rows, args.nnz, capi.SYNTH_CRITEO if criteo else capi.SYNTH_UNIFORM)
    info = hs[0].info()
    g = capi.Group(hs)
    lagf = 0 if args.no_bias_lag else capi.FLAG_BIAS_LAG
    flags = lagf | (capi.FLAG_PIPELINE if args.pipeline else 0)
    lag = args.bias_lag if lagf else 0

    def sync():
        for h in hs:
            h.synchronize()
    for _ in range(args.warmup):
        g.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, args.batch, args.w0_chunk, flags, lag)
    sync()
    phases, dev_s, st = [0.0, 0.0, 0.0], 0.0, None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = g.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, args.batch, args.w0_chunk, flags | capi.FLAG_TIME_MAIN_KERNEL, lag)
        dev_s += st.device_seconds
        for i in range(3):
            phases[i] += st.phase_seconds[i]
    sync()
    elapsed = time.perf_counter() - t0
    value = args.steps * args.rows / elapsed
    n_batches = max(1, int(st.batches)) * args.steps
    per_ex = algorithmic_bytes(args.k, args.nnz, "rowsums") + algorithmic_bytes(args.k, args.nnz, "apply")
    achieved = value * per_ex / N / 1e9
    wire = 4 * (info.k_padded + 1)
    out = {
        "metric": "SGD training examples/sec at k=%d, nnz=%d, %.0e feat" % (args.k, args.nnz, args.n),
        "value": round(value, 1), "unit": "examples/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s n=%d k=%d nnz=%d, %d examples/step, task=c lr=%g regv=%g"
                               % ("Criteo-shaped" if criteo else "synthetic one-hot fields", args.n, args.k, args.nnz, args.rows, lr, regv),
                   "mode": "minibatch (split step)", "batch": int(st.batch_used), "w0_chunk": int(st.w0_chunk_used), "bias_lag": lag,
                   "pipeline": bool(args.pipeline), "sharding": "feature-id hash (permutation) over %d shards" % N,
                   "driver": "one process, fmx_group (%s)" % ("loopback: all shards on device 0" if args.same_device else "RCCL, one communicator per device"),
                   "batch_rule": {"batch": int(st.batch_used), "collision_mass": round(st.collision_mass, 6), "gain": round(st.batch_gain, 4)},
                   "placement": [{"method": pi.method, "chunks": pi.chunks, "per_class": [pi.per_class[0], pi.per_class[1]], "pool_probed": pi.pool}
                                 for pi in (h.place_info() for h in hs)],
                   "device": info.device_name.decode(), "arch": info.arch.decode()},
        "roofline": {"bound": "hbm", "kernel": "k_rowsums + update (whole step, per GPU)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "v_read_frac": v_read_fraction(value, args.k, args.nnz, N),
                     "traffic": None, "bytes_per_example": per_ex, "examples_per_launch": int(st.batch_used)},
        "exchange": {"collective": ("reduce_scatter + all_gather (sum) fp32" if args.exchange == "rsag" else "all_reduce(sum) fp32, one per batch")
                                   + (", in 4 runs of rows overlapped with the sums" if not args.same_device else " (local reduction kernel%s)" % ("s, two-phase" if args.exchange == "rsag" else "")),
                     "algo": args.exchange,
                     "bytes_per_example": wire, "payload_MB_per_batch": round(wire * min(int(st.batch_used), args.rows) / 1e6, 2),
                     "algbw_GBps": round(value * wire / 1e9, 2), "pipelined": bool(args.pipeline),
                     "backend": "loopback" if args.same_device else "rccl"},
        # where a batch's time goes on shard 0 (HIP events on its compute stream): partial sums / exposed exchange / update
        "phases_ms_per_batch": {"sums": round(phases[0] / n_batches * 1e3, 4), "exchange_exposed": round(phases[1] / n_batches * 1e3, 4),
                                "update": round(phases[2] / n_batches * 1e3, 4), "device_total": round(dev_s / n_batches * 1e3, 4)},
        "cpu_baseline": None,
    }
    g.close()
    for h in hs:
        h.close()
    return out


def bench_group(args, capi, criteo):
    print(json.dumps(run_group(args, capi, criteo)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--method", default="sgd", choices=["sgd", "als", "mcmc"],
                    help="sgd: the BASELINE metric (default).  als / mcmc: one step = one sweep of fm_learn_mcmc over the rows "
                         "(BASELINE configs[3] / [4] shapes: n=1e7, k=64, 16 nnz/row unless overridden; one GPU)")
    ap.add_argument("--workload", default="uniform", choices=["uniform", "criteo"],
                    help="uniform: the BASELINE metric's rows (uniform ids within 32 fields of 1e8 features).  criteo: BASELINE configs[2] "
                         "on one GPU -- n = 3.3e7, 39 fields (13 of <= 100 ids, 26 Zipf 1.05), fmx_synth_rows_ex(FMX_SYNTH_CRITEO); the library "
                         "cuts the batch to the rows' stability bound (fmx_sgd_opts::batch = 0)")
    ap.add_argument("--features", dest="n", type=int, default=None, help="number of features n (sgd: 1e8, als/mcmc: 1e7)")
    ap.add_argument("--factors", dest="k", type=int, default=64, help="number of factors k")
    ap.add_argument("--nnz", type=int, default=None, help="entries per row (sgd: 32, als/mcmc: 16)")
    ap.add_argument("--rows", type=int, default=1 << 22, help="examples per step")
    ap.add_argument("--mode", default="auto", choices=["auto", "fused", "minibatch", "hogwild"],
                    help="auto: the minibatch rule everywhere -- `fused` (FMX_APPLY_FUSED, one pass) on one GPU, the split step "
                         "(feature-sharded) on several; minibatch: the two-pass segmented form; hogwild: asynchronous")
    ap.add_argument("--bias-lag", type=int, default=2, help="fused: batches the multipliers' bias lags behind (1..4)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary figures (hogwild, two-pass minibatch)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo stages the all-reduce through the host (testing the N>1 path without RCCL)")
    ap.add_argument("--same-device", action="store_true", help="testing: every shard / rank on device 0 (several GPUs from one process: the loopback exchange)")
    ap.add_argument("--multi-process", action="store_true",
                    help="several GPUs: one process per GPU even with --same-device (testing the self-launch on a one-GPU box; needs "
                         "--backend gloo --driver torch there: RCCL refuses two ranks on one device)")
    ap.add_argument("--one-process", action="store_true",
                    help="several GPUs: drive all shards from this process through fmx_group_* (one host thread) instead of one process per GPU")
    ap.add_argument("--driver", default="lib", choices=["lib", "torch"],
                    help="several GPUs: lib = the library's own schedule and RCCL binding (fmx_comm_init_rank + fmx_sgd_epoch; torch only "
                         "hands the communicator id to the ranks); torch = libfm_amd/distributed.py (partial -> dist.all_reduce -> finish)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="testing: drive the multi-GPU code path (ShardedSGD + all-reduce) even with one rank")
    ap.add_argument("--apply", default="default", choices=["default", "segmented", "atomic", "store"])
    ap.add_argument("--batch", type=int, default=0, help="minibatch rows (0: the library's choice -- 262144 cut to the stability bound of the rows, fmx_sgd_opts::batch); hogwild: rows per launch (0: 262144)")
    ap.add_argument("--w0-chunk", type=int, default=0, help="micro-chunk of the bias recurrence (0: library default, fmx_default_w0_chunk)")
    ap.add_argument("--exchange", default="allreduce", choices=["allreduce", "rsag"],
                    help="several GPUs: the per-batch exchange as ONE all-reduce (ncclAllReduce) or as reduce-scatter + all-gather "
                         "(fmx_config::exchange_algo: on a fully connected xGMI node every GPU reduces its slice over all its links)")
    ap.add_argument("--pipeline", dest="pipeline", action="store_true",
                    help="sharded: overlap the all-reduce of batch b+1 with the update of batch b (the one-batch-stale pipelined rule, "
                         "oracle fmo_sgd_epoch_minibatch_pipelined).  Default OFF: every N runs the SAME rule as N = 1")
    ap.add_argument("--no-bias-lag", action="store_true", help="minibatch: keep the w0 recurrence on the critical path (exact chunk coupling)")
    ap.add_argument("--cpu-rows", type=int, default=200_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes that measure roofline.traffic in this run")
    ap.add_argument("--place", type=int, default=None,
                    help="fmx_config::place_candidates: 0 = the library's placement (big tables: chunks of two memory classes), 1 = plain "
                         "allocations.  Default 0; with --same-device 1 (shards SHARING a device are the one case measured faster out of "
                         "plain allocations, 123 vs 113 M ex/s at two shards: profiles/r03_place_two_shards_one_device.txt)")
    ap.add_argument("--traffic", type=float, default=None,
                    help="PMC HBM bytes per launch of the dominant kernel (default: profiles/traffic.json if it matches)")
    ap.add_argument("--no-cpu-reference", dest="cpu_reference", action="store_false",
                    help="skip timing the REAL reference code (oracle/_ref/ref_harness time_sgd, largest n it can allocate)")
    args = ap.parse_args()
    if args.place is None:
        args.place = 1 if (args.same_device and args.gpus > 1) else 0

    criteo = args.workload == "criteo"
    if args.n is None:
        args.n = (33_000_000 if criteo else 100_000_000) if args.method == "sgd" else 10_000_000
    if args.nnz is None:
        args.nnz = (39 if criteo else 32) if args.method == "sgd" else 16

    import torch
    import torch.distributed as dist
    from libfm_amd import capi
    if args.method != "sgd":
        return bench_als(args, capi)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: either all shards from this process (fmx_group_*), or -- the default on N distinct
        # devices -- re-exec as one process per GPU, which is how the library's RCCL schedule keeps N host threads busy
        if (args.same_device or args.one_process) and not args.multi_process:
            return bench_group(args, capi, criteo)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.force_sharded
    if args.mode == "auto":
        args.mode = "minibatch" if sharded else "fused"
    if world > 1 and args.mode != "minibatch":
        raise SystemExit("several GPUs: only --mode minibatch (the split step over feature shards) exists")
    if sharded:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    cpu, cpu_ref = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not criteo:   # (the CPU legs time the uniform workload)
        cpu = cpu_baseline(args.n, args.k, args.nnz, args.cpu_rows)
        if args.cpu_reference:
            cpu_ref = cpu_reference(args.n, args.k, args.nnz, args.cpu_rows)

    lr, regv = 0.01, 0.001
    h = capi.Handle(args.n, args.k, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, regv, lr, -1.0, 1.0,
                    device=local_rank, shard_rank=rank, shard_world=world, shard_hash=1 if world > 1 else 0,
                    place_candidates=args.place, exchange_algo=1 if args.exchange == "rsag" else 0)
    h.init_params(0.0, 0.01, 1)
    h.synth_rows(0, 123, 0, args.rows, args.nnz, capi.SYNTH_CRITEO if criteo else capi.SYNTH_UNIFORM)
    if criteo and rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_rows(h, args.n, args.k, min(args.cpu_rows, 100_000))
    info = h.info()
    mode = capi.SGD_HOGWILD if args.mode == "hogwild" else capi.SGD_MINIBATCH
    apply_ = {"default": capi.APPLY_DEFAULT, "segmented": capi.APPLY_SEGMENTED, "atomic": capi.APPLY_ATOMIC,
              "store": capi.APPLY_STORE}[args.apply]
    if args.mode == "fused":
        apply_ = capi.APPLY_FUSED

    lagf = 0 if (args.no_bias_lag or args.mode == "hogwild") else capi.FLAG_BIAS_LAG
    bias_lag = args.bias_lag if args.mode == "fused" else (1 if lagf else 0)
    deferred = 0
    batch_stats = None
    # several GPUs, the library's own schedule (default): rank 0 creates the RCCL id, torch.distributed only carries it to the others.
    # Every rank reports whether its binding came up (communicator + a first collective: the shards' shares of the rows' collision
    # mass); if ANY rank failed, all of them fall back to the torch-driven schedule (libfm_amd/distributed.py) instead of dying.
    use_lib = sharded and args.driver == "lib" and args.backend == "nccl"
    fallback_reason = None
    if use_lib:
        ok, why = 1, ""
        try:
            uid = [None]
            if rank == 0:
                try:
                    uid = [capi.comm_unique_id()]
                except Exception as exc:                      # (the others must not hang in the broadcast)
                    why = str(exc)
            dist.broadcast_object_list(uid, src=0)
            if uid[0] is None:
                raise RuntimeError("rank 0 could not create an RCCL id: " + why)
            h.comm_init_rank(uid[0], rank, world)
            batch = h.sgd_batch_info(0, args.batch).batch
        except Exception as exc:
            ok, why = 0, str(exc)
        agree = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree.item()) == 0:
            use_lib = False
            args.driver = "torch"
            fallback_reason = "the library's RCCL binding did not come up on every rank (%s): torch-driven schedule (libfm_amd/distributed.py)" % (why or "another rank failed")
            if ok:
                h.comm_destroy()
            if rank == 0 or not ok:
                print("bench.py: rank %d: the library's RCCL binding did not come up on every rank (%s): falling back to --driver torch"
                      % (rank, why or "another rank failed"), file=sys.stderr, flush=True)
    if not sharded:
        # --batch 0: the library's choice (fmx_sgd_opts::batch = 0 -> 262144 cut to the rows' stability bound; hogwild: rows per launch)
        batch = h.sgd_batch_info(0, args.batch).batch if args.mode != "hogwild" else (args.batch or 262144)
        main_time, main_launches = 0.0, 0

        setup_s = 0.0

        def step(timed):
            nonlocal main_time, main_launches, deferred, batch_stats, setup_s
            st = h.sgd_epoch(0, mode, apply_, args.batch, args.w0_chunk, (capi.FLAG_TIME_MAIN_KERNEL if timed else 0) | lagf, bias_lag)
            setup_s += st.setup_seconds
            if timed:
                main_time += st.main_kernel_seconds
                main_launches += st.main_kernel_launches
                deferred += st.deferred_features
                batch_stats = st
        rows_per_launch = min(batch, args.rows)
        kind = "fused" if args.mode in ("hogwild", "fused") else "apply"
    elif use_lib:
        lib_flags = lagf | (capi.FLAG_PIPELINE if args.pipeline else 0)
        lib_lag = args.bias_lag if lagf else 0
        phases = [0.0, 0.0, 0.0]

        class _Drv:
            def synchronize(self):
                h.synchronize()
        drv = _Drv()

        def step(timed):
            nonlocal batch_stats
            st = h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_DEFAULT, batch, args.w0_chunk,
                             lib_flags | (capi.FLAG_TIME_MAIN_KERNEL if timed else 0), lib_lag)
            if timed:
                batch_stats = st
                for i in range(3):
                    phases[i] += st.phase_seconds[i]
        bias_lag = lib_lag
        rows_per_launch = min(batch, args.rows)
        kind = "apply"
        main_time, main_launches = 0.0, 0
    else:
        from libfm_amd.distributed import ShardedSGD
        batch = args.batch or 262144                 # per-rank compute side: 691 (131 072) -> 765 M examples/s (262 144) at P = 8
        drv = ShardedSGD(h, 0, args.rows, batch, args.w0_chunk, apply_, lagf, args.backend, pipeline=args.pipeline)

        def step(timed):
            drv.epoch()
        rows_per_launch = min(batch, args.rows)
        kind = "apply"
        main_time, main_launches = 0.0, 0

    for _ in range(args.warmup):
        step(False)
    if sharded:
        drv.synchronize()
        dist.barrier()
    torch.cuda.synchronize()
    h.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    if sharded:
        drv.synchronize()
    h.synchronize()
    torch.cuda.synchronize()
    if sharded:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    extras = {}
    if rank == 0 and not sharded and not args.no_extras and not criteo:
        # secondary figures, never `value`: the asynchronous mode (metric-level parity only) and the two-pass form of the rule
        def timed_epochs(n, *a):
            h.sgd_epoch(0, *a)
            h.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                h.sgd_epoch(0, *a)
            h.synchronize()
            return round(n * args.rows / (time.perf_counter() - t1), 1)
        # the V-gather on its own: fm_model::predict over the same rows (k_rowsums + fused evaluation), the kernel north_star's
        # ">= 60 % of the HBM-read roofline on the V-gather" is about; v_read_frac = rows/s x nnz*k*4 B / 8 TB/s
        def eval_rate():
            ev_s, fl = 0.0, 0
            h.evaluate(0)
            for _ in range(5):
                ev = h.evaluate(0)
                ev_s += ev.device_seconds
                fl = ev.flags
            return 5 * args.rows / ev_s, bool(fl & capi.EVAL_WSIDE)
        rps_plain, _ = eval_rate()
        # the same pass after an epoch that kept the slot's weight side stream (FMX_FLAG_KEEP_WSIDE: what a learner that evaluates its train
        # set after every epoch runs, fm_learn_sgd_element.h:69-70): w_j of an entry's last occurrence in the slot comes out of a 4-byte stream
        # instead of a 64-byte fabric request.  Same numbers (tests/test_gpu_parity.py); the epoch itself writes 128 more bytes per example.
        t1 = time.perf_counter()
        h.sgd_epoch(0, mode, apply_, args.batch, args.w0_chunk, lagf | capi.FLAG_KEEP_WSIDE, bias_lag)
        h.sgd_epoch(0, mode, apply_, args.batch, args.w0_chunk, lagf | capi.FLAG_KEEP_WSIDE, bias_lag)
        h.synchronize()
        t2 = time.perf_counter()
        h.sgd_epoch(0, mode, apply_, args.batch, args.w0_chunk, lagf | capi.FLAG_KEEP_WSIDE, bias_lag)
        h.synchronize()
        keep_ms = (time.perf_counter() - t2) * 1e3
        rps, streamed = eval_rate() if args.mode == "fused" else (rps_plain, False)
        extras["predict"] = {"mode": "fm_model::predict + evaluate over the step's rows (k_rowsums), the full model incl. linear weights"
                                     + ("; linear weights out of the slot's side stream where the entry is the feature's last occurrence (FMX_FLAG_KEEP_WSIDE)" if streamed else ""),
                             "value": round(rps, 1), "unit": "rows/s", "v_read_frac": v_read_fraction(rps, args.k, args.nnz),
                             "frac": round(rps * (args.nnz * (4 * args.k + 12) + 4) / 1e9 / HBM_PEAK_GBS, 4),
                             "weight_side_stream": streamed,
                             "without_side_stream": {"value": round(rps_plain, 1), "v_read_frac": v_read_fraction(rps_plain, args.k, args.nnz)},
                             "epoch_keeping_the_stream_ms": round(keep_ms, 3)}
        if args.mode != "hogwild":
            extras["hogwild"] = {"mode": "hogwild (asynchronous one-pass step; parity only metric-level -- NOT the headline)",
                                 "batch": 262144, "value": timed_epochs(5, capi.SGD_HOGWILD, capi.APPLY_DEFAULT, 262144, args.w0_chunk, 0),
                                 "unit": "examples/s", "steps": 5}
        # the reference's OWN trajectory (fm_learn_sgd_element.h:56-67: one example at a time, file order) on the device: the parity mode,
        # exact to 1e-4 against the real reference's final parameters (tests/test_gpu_parity.py); eight wavefronts per example
        try:
            seq_rows = 1 << 20
            h.synth_rows(1, 321, 0, seq_rows, args.nnz)
            h.sgd_epoch(1, capi.SGD_SEQUENTIAL)                    # (incl. the one-time cut of the slot into conflict-free runs)
            h.synchronize()
            t1 = time.perf_counter()
            st = h.sgd_epoch(1, capi.SGD_SEQUENTIAL)
            h.synchronize()
            as_runs = bool(st.status & capi.STAT_SEQ_RUNS)
            extras["sequential"] = {"mode": "FMX_SGD_SEQUENTIAL: the reference's own example-by-example trajectory on the device ("
                                            + ("conflict-free runs: k_rowsums + k_run_apply per run of consecutive rows that share no feature" if as_runs
                                               else "k_sequential_wg: eight wavefronts per example") + ")",
                                    "value": round(seq_rows / (time.perf_counter() - t1), 1), "unit": "examples/s", "rows": seq_rows,
                                    "runs": int(st.batches) if as_runs else None}
        except Exception as exc:
            extras["sequential"] = {"error": str(exc)[:200]}
        if args.mode != "minibatch":
            extras["minibatch_two_pass"] = {"mode": "minibatch rule, two passes (k_rowsums + k_apply_seg), bias lag 1", "batch": 131072,
                                            "value": timed_epochs(3, capi.SGD_MINIBATCH, capi.APPLY_SEGMENTED, 131072, args.w0_chunk, capi.FLAG_BIAS_LAG),
                                            "unit": "examples/s", "steps": 3}
        # how far the headline rule ends from the reference's ONLINE loop -- MEASURED HERE, with the device on both sides: FMX_SGD_SEQUENTIAL is the
        # reference's own trajectory (held at 1e-4 against the real reference's final parameters by tests/test_gpu_parity.py, test_gpu_configs.py)
        # and, as conflict-free runs, fast enough to walk 1.18 M rows inside this run.  Same start values, same rows, one epoch each.
        # (last of the extras: it starts the handle's parameters over)
        if args.mode == "fused":
            try:
                import numpy as np
                pv_rows = 1179648                                   # 4.5 batches: the rows of the committed CPU figure (profiles/r05_parity_vs_online.json)
                h.synth_rows(1, 20240, 0, pv_rows, args.nnz)
                h.init_params(0.0, 0.01, 1)
                t1 = time.perf_counter()
                st_seq = h.sgd_epoch(1, capi.SGD_SEQUENTIAL)
                h.synchronize()
                t_seq = time.perf_counter() - t1
                p_on, w0_on = h.predict(1, pv_rows).astype(np.float64), h.get_w0()
                h.init_params(0.0, 0.01, 1)
                h.sgd_epoch(1, mode, apply_, args.batch, args.w0_chunk, lagf, bias_lag)
                h.synchronize()
                p_ru, w0_ru = h.predict(1, pv_rows).astype(np.float64), h.get_w0()
                dp = np.abs(p_ru - p_on)
                extras["parity_vs_online_live"] = {
                    "rows": pv_rows, "epochs": 1, "pred_rms": round(float(np.sqrt(np.mean(p_on * p_on))), 6),
                    "pred_mean_abs": round(float(dp.mean()), 6), "pred_max_abs": round(float(dp.max()), 6), "w0_abs": round(abs(w0_ru - w0_on), 6),
                    "online_side": "FMX_SGD_SEQUENTIAL on the device" + (" (conflict-free runs: %d)" % st_seq.batches if st_seq.status & capi.STAT_SEQ_RUNS else ""),
                    "online_side_seconds": round(t_seq, 4),
                    "source": "measured in this run: the headline rule against the device's reference-trajectory mode, same start, same rows, one epoch each"}
                # ... and the YARDSTICK, measured the same way: the reference's trajectory against ITSELF when the rows inside every batch-sized window
                # come in another order (same rows, same start): what the order of the rows is worth to online SGD
                try:
                    ent, rp, yy = h.download_rows(1)                    # fixed-length rows: entry list of row r = [r * nnz, (r + 1) * nnz)
                    rng = np.random.default_rng(7)
                    perm = np.arange(pv_rows)
                    win = batch if batch else 262144
                    for a0 in range(0, pv_rows, win):
                        rng.shuffle(perm[a0:min(a0 + win, pv_rows)])
                    h.upload_rows(2, ent.reshape(pv_rows, args.nnz)[perm].reshape(-1), rp, yy[perm])
                    del ent
                    h.init_params(0.0, 0.01, 1)
                    h.sgd_epoch(2, capi.SGD_SEQUENTIAL)
                    h.synchronize()
                    p_sh, w0_sh = h.predict(1, pv_rows).astype(np.float64), h.get_w0()   # (the SAME rows in the same order as p_on)
                    ds = np.abs(p_sh - p_on)
                    extras["parity_vs_online_live"]["reference_vs_itself_with_rows_shuffled_inside_batch_sized_windows"] = {
                        "pred_mean_abs": round(float(ds.mean()), 6), "pred_max_abs": round(float(ds.max()), 6), "w0_abs": round(abs(w0_sh - w0_on), 6),
                        "window": int(win), "source": "measured in this run: FMX_SGD_SEQUENTIAL on the rows in file order vs on the rows shuffled inside windows"}
                    # ... and under the SMALLEST change of order: every two neighbouring rows change places
                    sw = np.arange(pv_rows)
                    sw[0:pv_rows - pv_rows % 2:2] += 1
                    sw[1:pv_rows:2] -= 1
                    ent, rp, yy = h.download_rows(1)
                    h.upload_rows(2, ent.reshape(pv_rows, args.nnz)[sw].reshape(-1), rp, yy[sw])
                    del ent
                    h.init_params(0.0, 0.01, 1)
                    h.sgd_epoch(2, capi.SGD_SEQUENTIAL)
                    h.synchronize()
                    p_sw, w0_sw = h.predict(1, pv_rows).astype(np.float64), h.get_w0()
                    dw = np.abs(p_sw - p_on)
                    extras["parity_vs_online_live"]["reference_vs_itself_with_neighbouring_rows_swapped"] = {
                        "pred_mean_abs": round(float(dw.mean()), 6), "pred_max_abs": round(float(dw.max()), 6), "w0_abs": round(abs(w0_sw - w0_on), 6),
                        "source": "measured in this run: FMX_SGD_SEQUENTIAL on the rows in file order vs with rows 2i and 2i + 1 exchanged"}
                    h.free_rows(2)
                except Exception as exc:
                    extras["parity_vs_online_live"]["reference_vs_itself_with_rows_shuffled_inside_batch_sized_windows"] = {"error": str(exc)[:200]}
            except Exception as exc:
                extras["parity_vs_online_live"] = {"error": str(exc)[:200]}
        # the headline kernels' HBM bytes, MEASURED in this run (two counter passes in child processes; --no-live-traffic skips them)
        if args.mode == "fused" and not args.no_live_traffic and args.traffic is None:
            try:
                lt = live_traffic(args, min(batch if batch else 262144, args.rows))
                if lt:
                    extras["_live_traffic"] = lt
            except Exception:
                pass

    if rank == 0:
        value = args.steps * args.rows / elapsed
        roof = None
        live = None                                          # (k_fused bytes, deferred-pass bytes, text) measured in this run, if it was
        if main_launches:
            # the shard sees nnz/world entries per example; k_apply / k_fused bytes scale with them
            per_ex = algorithmic_bytes(args.k, args.nnz, kind)
            avg = main_time / main_launches
            achieved = per_ex * rows_per_launch / avg / 1e9
            kname = {"fused": "k_fused<EXACT>", "hogwild": "k_fused"}.get(args.mode) or \
                ("k_apply_seg" if args.apply in ("default", "segmented") else "k_apply")
            traffic, tsrc = (args.traffic, "--traffic") if args.traffic is not None else \
                committed_traffic(kname, rows_per_launch, args.k, args.nnz)
            live = extras.pop("_live_traffic", None) if isinstance(extras, dict) else None
            if live and args.traffic is None:
                traffic, tsrc = live[0], live[2]
            roof = {"bound": "hbm", "kernel": kname,
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "v_read_frac": v_read_fraction(rows_per_launch / avg, args.k, args.nnz),
                    "traffic": traffic, "traffic_source": tsrc,
                    "bytes_per_example": per_ex, "examples_per_launch": rows_per_launch,
                    "avg_launch_ms": round(avg * 1e3, 4), "launches": main_launches,
                    "launch_time": "epoch HIP-event time / launches: the launch of one batch incl. its collision pass and gaps"
                                   if kind == "fused" else "HIP events around every launch"}
            if args.mode == "fused":
                roof["deferred_features_per_example"] = round(deferred / (args.steps * args.rows), 4)
        exchange = None
        if sharded:
            # no per-launch timing in the multi-process driver (it would serialise the overlap): whole-step accounting.
            # HBM side, per GPU: the step's two passes over the LOCAL entries (gather + segmented update).
            per_ex = algorithmic_bytes(args.k, args.nnz, "rowsums") + algorithmic_bytes(args.k, args.nnz, "apply")
            achieved = value * per_ex / world / 1e9
            roof = {"bound": "hbm", "kernel": "k_rowsums + k_apply_seg (whole step, per GPU)", "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "v_read_frac": v_read_fraction(value, args.k, args.nnz, world), "traffic": None,
                    "bytes_per_example": per_ex, "examples_per_launch": rows_per_launch}
            # wire side: ONE all-reduce of [batch][KP + 1] fp32 per batch; algbw = payload bytes reduced per second
            wire = 4 * (info.k_padded + 1)
            exchange = {"collective": ("reduce_scatter + all_gather (sum) fp32, per run of rows" if args.exchange == "rsag" else "all_reduce(sum) fp32, one per batch"),
                        "algo": args.exchange, "bytes_per_example": wire,
                        "payload_MB_per_batch": round(wire * rows_per_launch / 1e6, 2),
                        "algbw_GBps": round(value * wire / 1e9, 2), "pipelined": bool(args.pipeline), "backend": args.backend}
        out = {
            "metric": "SGD training examples/sec at k=%d, nnz=%d, %.0e feat" % (args.k, args.nnz, args.n),
            "value": round(value, 1), "unit": "examples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("Criteo-shaped (13 fields of <= 100 ids + %d Zipf(1.05) fields) n=%d k=%d nnz=%d, %d examples/step, task=c lr=%g regv=%g"
                                    % (args.nnz - 13, args.n, args.k, args.nnz, args.rows, lr, regv)) if criteo else
                                   "synthetic one-hot fields n=%d k=%d nnz=%d, %d examples/step, task=c lr=%g regv=%g"
                                   % (args.n, args.k, args.nnz, args.rows, lr, regv),
                       "mode": args.mode, "apply": args.apply, "batch": batch,
                       "w0_chunk": int(batch_stats.w0_chunk_used) if batch_stats is not None and batch_stats.w0_chunk_used else (args.w0_chunk or capi.default_w0_chunk(lr, capi.TASK_CLASSIFICATION)),
                       "bias_lag": bias_lag, "pipeline": bool(args.pipeline) if sharded else False, "sharding": "feature-id hash (permutation) over %d shards" % world if world > 1 else "none",
                       "driver": (args.driver if sharded else "single handle"),
                       "device": info.device_name.decode(), "arch": info.arch.decode()},
            "roofline": roof,
            "cpu_baseline": cpu_ref if (cpu_ref and "value" in cpu_ref) else cpu,
        }
        if fallback_reason:
            out["driver_fallback"] = fallback_reason          # (loud: the line says which driver ran and why, not only stderr)
        if not sharded and args.mode == "fused" and not criteo:
            # how far the headline rule ends from the reference's ONLINE loop at this shape (the committed CPU measurement of the oracle's
            # two loops; asserted with the device in place of the rule by tests/test_gpu_configs.py)
            try:
                pv = json.load(open(os.path.join(ROOT, "profiles", "r05_parity_vs_online.json")))
                if (pv["n"], pv["k"], pv["nnz"], pv["batch"], pv["bias_lag"], pv["chunk"]) == (args.n, args.k, args.nnz, batch, bias_lag, out["config"]["w0_chunk"]):
                    out["parity_vs_online"] = {kk: pv[kk] for kk in ("rows", "epochs", "chunk", "pred_rms", "pred_mean_abs", "pred_max_abs", "w0_abs",
                                                                     "pred_max_rel_to_rms_without_bias", "v_max_abs", "v_max_rel_to_vmax", "w_max_abs")}
                    # the yardsticks: how far the reference's OWN online result moves (same rows, same start) when every two neighbouring rows
                    # change places, and when the rows inside every batch-sized window come in another order
                    for key, fn in (("reference_vs_itself_with_neighbouring_rows_swapped", "r05_order_noise.json"),
                                    ("reference_vs_itself_with_rows_shuffled_inside_batch_sized_windows", "r05_order_noise_window262144.json")):
                        try:
                            on = json.load(open(os.path.join(ROOT, "profiles", fn)))
                            if (on["n"], on["k"], on["nnz"], on["rows"]) == (pv["n"], pv["k"], pv["nnz"], pv["rows"]):
                                out["parity_vs_online"][key] = {kk: on[kk] for kk in ("pred_mean_abs", "pred_max_abs", "w0_abs", "v_max_rel_to_vmax", "w_max_abs")}
                        except (OSError, ValueError, KeyError):
                            pass
                    out["parity_vs_online"]["source"] = "NOT measured in this run (the key parity_vs_online_live is: the same comparison with the device on both sides): profiles/r05_parity_vs_online.json (scripts/cpu_online_vs_rule.py, ~45 CPU-minutes: oracle rule vs oracle online loop on the sub-model of the rows' features); the device equals the rule at 1e-4 (tests/test_gpu_configs.py)"
            except (OSError, ValueError, KeyError):
                pass
        if not sharded and args.mode != "hogwild" and roof is not None:
            # the same as flat scalars of `roofline`: the one-time preparation and what a 20-epoch run (BASELINE configs[0]'s count) delivers end to end
            roof["setup_seconds"] = round(setup_s, 4)
            roof["end_to_end_20_epochs_examples_per_s"] = round(20 * args.rows / (setup_s + 20 * elapsed / args.steps), 1)
        if not sharded and args.mode != "hogwild":
            # outside `value` and outside the timed steps: paid once per (data set, batch size) in the first warm-up step
            out["one_time_setup"] = {"seconds": round(setup_s, 4), "equivalent_steps": round(setup_s / (elapsed / args.steps), 2),
                                     "what": "collision mass of the rows + bucketing of the entries by (batch, feature) (device radix sort, "
                                             "segment / mask / deferred-list build; fmx_epoch_stats::setup_seconds); libFM never shuffles "
                                             "(fm_learn_sgd_element.h:56), so every later epoch reuses it"}
        pi = h.place_info()
        out["config"]["placement"] = {"method": {0: "plain allocations", 1: "best of candidate allocations",
                                                 2: "arena of 1 GiB chunks from two memory classes"}.get(pi.method, str(pi.method)),
                                      "chunks": pi.chunks, "per_class": [pi.per_class[0], pi.per_class[1]], "pool_probed": pi.pool,
                                      "classes_seen": pi.classes_seen, "seconds": round(pi.seconds, 3)}
        if batch_stats is not None and args.mode != "hogwild":
            out["config"]["batch_rule"] = {"batch": batch_stats.batch_used, "collision_mass": round(batch_stats.collision_mass, 6),
                                           "gain": round(batch_stats.batch_gain, 4),
                                           "cut": bool(batch_stats.status & capi.STAT_BATCH_CUT), "unstable": bool(batch_stats.status & capi.STAT_UNSTABLE)}
        if cpu_ref and "value" in cpu_ref and cpu is not None:
            out["cpu_port"] = cpu
        if exchange is not None:
            out["exchange"] = exchange
            if args.driver == "lib" and args.backend == "nccl" and batch_stats is not None:
                nb = max(1, int(batch_stats.batches)) * args.steps
                out["phases_ms_per_batch"] = {"sums": round(phases[0] / nb * 1e3, 4), "exchange_exposed": round(phases[1] / nb * 1e3, 4),
                                              "update": round(phases[2] / nb * 1e3, 4), "rank": 0}
        out.update(extras)
        if roof is not None and "predict" in extras:
            # what north_star scores, as flat numbers inside `roofline` (the driver's record keeps only the NAMES of the extra keys)
            pr = extras["predict"]
            # ... and once more as FLAT scalars: the driver's record keeps scalars only
            roof["predict_v_read_frac"] = pr["v_read_frac"]
            roof["predict_v_read_frac_cold"] = pr["without_side_stream"]["v_read_frac"]
            roof["predict_rows_per_s"] = pr["value"]
            roof["predict_rows_per_s_cold"] = pr["without_side_stream"]["value"]
            if isinstance(extras.get("sequential"), dict) and "value" in extras["sequential"]:
                roof["examples_per_s_sequential"] = extras["sequential"]["value"]
            pl = extras.get("parity_vs_online_live")
            if isinstance(pl, dict) and "pred_mean_abs" in pl:      # measured in this run (device on both sides)
                roof["parity_vs_online_pred_mean_abs"] = pl["pred_mean_abs"]
                roof["parity_vs_online_pred_max_abs"] = pl["pred_max_abs"]
                roof["parity_vs_online_w0_abs"] = pl["w0_abs"]
                roof["parity_vs_online_pred_rms"] = pl["pred_rms"]
                ys = pl.get("reference_vs_itself_with_rows_shuffled_inside_batch_sized_windows")
                if isinstance(ys, dict) and "pred_mean_abs" in ys:  # the yardstick: the reference's trajectory against itself, rows reordered inside windows
                    roof["order_noise_pred_mean_abs"] = ys["pred_mean_abs"]
                    roof["order_noise_pred_max_abs"] = ys["pred_max_abs"]
                    roof["order_noise_w0_abs"] = ys["w0_abs"]
                yn = pl.get("reference_vs_itself_with_neighbouring_rows_swapped")
                if isinstance(yn, dict) and "pred_mean_abs" in yn:
                    roof["pair_swap_noise_pred_mean_abs"] = yn["pred_mean_abs"]
                    roof["pair_swap_noise_w0_abs"] = yn["w0_abs"]
            roof["predict"] = {"v_read_frac": pr["v_read_frac"], "v_read_frac_cold": pr["without_side_stream"]["v_read_frac"],
                               "rows_per_s": pr["value"], "rows_per_s_cold": pr["without_side_stream"]["value"],
                               "cold": "no weight side stream (a pass that no epoch on the slot preceded)"}
        if roof is not None and roof.get("traffic") and not sharded:
            try:
                e = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(roof["kernel"], {})
                step_bytes = roof["traffic"] + (live[1] if live and args.traffic is None else e.get("deferred_pass_bytes_per_launch", 0))
                roof["step_traffic_ratio"] = round(step_bytes / (roof["bytes_per_example"] * roof["examples_per_launch"]), 4)
                roof["step_traffic_ratio_source"] = roof["traffic_source"] + " (+ the deferred-feature pass); counter bytes / algorithmic bytes of one batch"
            except (OSError, ValueError, KeyError):
                pass
    h.close()
    if sharded:
        dist.destroy_process_group()
    if rank == 0 and not sharded and not args.no_extras and not criteo and args.mode == "fused":
        # the other two learners north_star names (BASELINE configs[3] / [4] shapes on one GPU): one line each, never `value`
        keep = ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config", "roofline", "cpu_baseline")
        for method in ("als", "mcmc"):
            try:
                o = run_als(capi, method, 10_000_000, 64, 16, 1 << 22, 2, 1, not args.no_cpu_baseline)
                out[method] = {kk: o[kk] for kk in keep}
            except Exception as exc:                             # a secondary figure must not take the headline down
                out[method] = {"error": str(exc)[:200]}
        # the BASELINE configs the headline does not cover, each on one GPU with its own roofline and CPU baseline (round-3 verdict, item 2):
        #   c2       configs[1]  n = 1e7, k = 32, 16 entries/row, SGD
        #   criteo   configs[2]  n = 3.3e7, k = 64, 39 entries/row, Criteo-shaped ids: the library cuts the batch (fmx_sgd_opts::batch = 0)
        #   mcmc_c5  configs[4]  MCMC (Gibbs draws), k = 128, n = 1e8, 16 entries/row -- a 51 GB table on ONE GPU; its CPU baseline is the
        #            reference's chain at the largest n that keeps the leg inside ~20 s (it draws all n * k coordinates per iteration)
        legs = (("c2", lambda: run_sgd_config(capi, "BASELINE configs[1]", 10_000_000, 32, 16, 1 << 23, False, 5, 2, not args.no_cpu_baseline, args.cpu_rows)),
                ("criteo", lambda: run_sgd_config(capi, "BASELINE configs[2], Criteo-shaped ids", 33_000_000, 64, 39, 1 << 20, True, 3, 1, not args.no_cpu_baseline, 100_000)),
                ("mcmc_c5", lambda: run_als(capi, "mcmc", 100_000_000, 128, 16, 1 << 22, 2, 1, not args.no_cpu_baseline,
                                            {"rows": 100_000, "n": 1_000_000, "gpu_leg": "n=1e8 (the stock containers stop at k*n < 2^32, i.e. n <= 3.3e7 at k=128, and "
                                             "the chain draws all n*k coordinates per iteration: n=1e6 keeps this leg near 20 s) and 42x the rows"})))
        for key, leg in legs:
            try:
                o = leg()
                out[key] = {kk: o[kk] for kk in keep + ("one_time_setup_seconds",) if kk in o}
            except Exception as exc:
                out[key] = {"error": str(exc)[:200]}
        # BASELINE configs[2] AS IT IS WORDED ("V row-sharded across 8"): eight feature shards driven by fmx_group_sgd_epoch -- on this one GPU
        # through the loopback exchange, so the eight shards' launches of a batch run one after the other on the device (a real node runs them
        # side by side): the figure is the schedule's cost per 512-row batch, not a scaling number
        try:
            ga = argparse.Namespace(gpus=8, same_device=True, n=33_000_000, k=64, nnz=39, rows=1 << 17, steps=2, warmup=1, place=args.place,
                                    exchange="allreduce", no_bias_lag=False, pipeline=False, bias_lag=2, batch=0, w0_chunk=0)
            o = run_group(ga, capi, True)
            out["criteo_8shard"] = {kk: o[kk] for kk in ("metric", "value", "unit", "n_gpus", "ms_per_step", "steps", "dtype", "config", "roofline", "exchange",
                                                          "phases_ms_per_batch") if kk in o}
            out["criteo_8shard"]["launches_per_shard_and_batch"] = 3
            out["criteo_8shard"]["note"] = ("8 loopback shards on ONE device: 25 dependent launches per 512-row batch in one stream (1 sums + 2 update launches per shard, "
                                            "1 reduction); round 5's general schedule made ~20 host calls per shard and batch (FMX_GROUP_IN_STREAM=0)")
        except Exception as exc:
            out["criteo_8shard"] = {"error": str(exc)[:200]}
        # the per-rank compute side of the sharded step at P = 2 / 4 / 8 (the only part of the 1/2/4/8-GPU metric one GPU can measure)
        try:
            out["shard_probe"] = shard_probe(capi, args.n, args.k, args.nnz, out["value"])
        except Exception as exc:
            out["shard_probe"] = {"error": str(exc)[:200]}
        if out.get("roofline") is not None:
            out["roofline"]["per_config"] = {kk: (out[kk].get("roofline") or {}).get("frac") for kk in ("c2", "criteo", "als", "mcmc", "mcmc_c5")
                                             if isinstance(out.get(kk), dict)}
            for kk, fv in out["roofline"]["per_config"].items():   # flat scalars (the driver's record drops nested objects)
                out["roofline"]["frac_" + kk] = fv
            for kk in ("c2", "criteo", "als", "mcmc", "mcmc_c5", "criteo_8shard"):
                if isinstance(out.get(kk), dict) and "value" in out[kk]:
                    out["roofline"]["examples_per_s_" + kk] = out[kk]["value"]
            cr = out.get("criteo")
            if isinstance(cr, dict) and "value" in cr and isinstance(cr.get("config"), dict):   # what a 512-row batch of Criteo-shaped rows costs (one launch: k_small_one)
                try:
                    out["roofline"]["us_per_batch_criteo"] = round(cr["config"]["batch_rule"]["batch"] / cr["value"] * 1e6, 3)
                    out["roofline"]["batch_criteo"] = cr["config"]["batch_rule"]["batch"]
                except (KeyError, TypeError, ZeroDivisionError):
                    pass
            sq = extras.get("sequential") if isinstance(extras, dict) else None
            if isinstance(sq, dict) and sq.get("runs"):
                out["roofline"]["sequential_runs"] = sq["runs"]
                out["roofline"]["sequential_rows"] = sq["rows"]
            sp = out.get("shard_probe", {}).get("ranks", {})
            if "P8" in sp:
                out["roofline"]["shard_p8_rank_examples_per_s"] = sp["P8"]["per_rank_examples_per_s"]
                out["roofline"]["modelled_speedup_p8_exact"] = sp["P8"]["modelled"]["speedup_vs_1_gpu_exact"]
                out["roofline"]["modelled_speedup_p8_stale"] = sp["P8"]["modelled"].get("speedup_vs_1_gpu_one_batch_stale")
            out["roofline"]["shard_probe"] = {pk: {"per_rank_examples_per_s": pv["per_rank_examples_per_s"], "frac": pv["frac_of_hbm_peak"],
                                                   "modelled_speedup_exact": pv["modelled"]["speedup_vs_1_gpu_exact"]} for pk, pv in sp.items()}
    if rank == 0:
        # RCCL prints its version banner through C stdio (NCCL_DEBUG=VERSION): flush it first so that the JSON
        # line is the LAST line on stdout
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

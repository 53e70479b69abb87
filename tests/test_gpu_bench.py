"""GPU: bench.py as the driver types it.  `python bench.py --gpus 2 --same-device` must run by itself on a one-GPU box (two
feature shards on device 0 through fmx_group_*: the N > 1 code path with the loopback exchange) and print one JSON line that
carries the contract's keys plus `exchange`, `config.sharding` and the per-phase times; the N = 1 line carries `roofline` and the
library-chosen batch; the Criteo-shaped workload (BASELINE configs[2]) runs with the batch cut to its stability bound."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline"]


def run_bench(*args):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + [str(a) for a in args], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    for k in KEYS:
        assert k in out, k
    return out


def test_two_shards_on_one_device_as_typed():
    out = run_bench("--gpus", 2, "--same-device", "--features", 4_000_000, "--rows", 131072, "--batch", 32768, "--steps", 2, "--warmup", 1)
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "2 shards" in out["config"]["sharding"] and out["config"]["batch"] == 32768
    assert out["exchange"]["bytes_per_example"] == 4 * 65 and out["exchange"]["backend"] == "loopback"
    ph = out["phases_ms_per_batch"]
    assert ph["sums"] > 0 and ph["update"] > 0 and ph["exchange_exposed"] >= 0
    assert ph["sums"] + ph["exchange_exposed"] + ph["update"] <= 1.2 * ph["device_total"] + 0.05


def test_single_gpu_line_and_criteo_shape():
    out = run_bench("--features", 4_000_000, "--rows", 262144, "--steps", 2, "--warmup", 1, "--no-cpu-baseline", "--no-extras")
    assert out["n_gpus"] == 1 and out["config"]["batch_rule"]["cut"] is False and out["roofline"]["frac"] > 0
    out = run_bench("--workload", "criteo", "--features", 2_000_000, "--rows", 65536, "--steps", 1, "--warmup", 1)
    br = out["config"]["batch_rule"]
    assert br["cut"] is True and br["unstable"] is False and br["gain"] <= 1.0 and br["batch"] <= 2048
    assert "Criteo-shaped" in out["config"]["workload"]


def test_self_launch_one_process_per_rank():
    """`python bench.py --gpus 2` re-executes itself under torch.distributed.run (one process per GPU: the default on N distinct
    devices).  On a one-GPU box the two ranks share device 0 and exchange through gloo (RCCL refuses two ranks on one device); the
    launcher, the rank environment, the barriers and rank 0's JSON line are the ones the 8-GPU run uses."""
    out = run_bench("--gpus", 2, "--same-device", "--multi-process", "--backend", "gloo", "--driver", "torch", "--features", 2_000_000,
                    "--rows", 65536, "--batch", 16384, "--steps", 2, "--warmup", 1)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["driver"] == "torch"
    assert "2 shards" in out["config"]["sharding"] and out["exchange"]["backend"] == "gloo"


def test_eight_shards_and_the_pipelined_schedule_as_typed():
    """`python bench.py --gpus 8 --same-device` and `--gpus 4 --same-device --pipeline` at the bench's own shape (n = 1e8, 4 194 304 rows per
    step): what the driver's 8-GPU run types, on one device (loopback exchange).  The line carries the phase times, the exchange and the
    SAME batch rule as N = 1 (batch 262 144, nothing to cut); the pipelined schedule reports its exposed exchange from the right events
    (round-3 advisor: it used to contain the previous batch's update)."""
    one = run_bench("--steps", 1, "--warmup", 1, "--no-cpu-baseline", "--no-extras")
    rule1 = one["config"]["batch_rule"]
    assert rule1["batch"] == 262144 and rule1["cut"] is False
    for argv, n in ((("--gpus", 8, "--same-device"), 8), (("--gpus", 4, "--same-device", "--pipeline"), 4)):
        out = run_bench(*argv, "--steps", 2, "--warmup", 1)
        assert out["n_gpus"] == n and out["value"] > 0 and "%d shards" % n in out["config"]["sharding"]
        assert out["config"]["pipeline"] is ("--pipeline" in argv)
        br = out["config"]["batch_rule"]
        assert br["batch"] == rule1["batch"] and abs(br["collision_mass"] - rule1["collision_mass"]) <= 0.05 * rule1["collision_mass"] + 1e-7
        assert out["exchange"]["bytes_per_example"] == 4 * 65 and out["exchange"]["pipelined"] is ("--pipeline" in argv)
        ph = out["phases_ms_per_batch"]
        assert ph["sums"] > 0 and ph["update"] > 0 and 0 <= ph["exchange_exposed"] <= ph["device_total"]
        assert ph["sums"] + ph["exchange_exposed"] + ph["update"] <= 1.2 * ph["device_total"] + 0.05
        assert "driver_fallback" not in out

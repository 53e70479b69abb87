"""GPU (one device): the N>1 driver end to end -- two processes, each a feature shard on cuda:0, exchanging through
gloo (host-staged all-reduce), must train the same model as one unsharded handle (restated batch rule, bias-lag
variant).  This is the code path bench.py runs under torch.distributed.run with RCCL; only the transport differs."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, K, NNZ, ROWS, B = 64000, 32, 16, 20000, 4096


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from libfm_amd import capi
    from libfm_amd.distributed import ShardedSGD
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h = capi.Handle(N, K, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.02, -1.0, 1.0, device=0,
                    shard_rank=rank, shard_world=world)
    h.init_params(0.0, 0.05, 1)
    h.synth_rows(0, 123, 0, ROWS, NNZ)
    drv = ShardedSGD(h, 0, ROWS, B, 64, capi.APPLY_DEFAULT, capi.FLAG_BIAS_LAG, "gloo")
    for _ in range(3):
        drv.epoch()
    drv.synchronize()
    w = np.zeros(N)
    v = np.zeros((K, N))
    w0, w, v = h.get_params(w, v)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), w0=w0, w=w, v=v)
    h.close()
    dist.destroy_process_group()


def test_two_shards_on_one_gpu_equal_the_unsharded_run(tmp_path):
    import torch.multiprocessing as mp
    from libfm_amd import capi
    world, port = 2, 29700 + (os.getpid() % 200)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    h = capi.Handle(N, K, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.02, -1.0, 1.0)
    h.init_params(0.0, 0.05, 1)
    h.synth_rows(0, 123, 0, ROWS, NNZ)
    for _ in range(3):
        h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_SEGMENTED, B, 64, capi.FLAG_BIAS_LAG)
    w0, w, v = h.get_params()
    h.close()
    ws, vs = np.zeros_like(w), np.zeros_like(v)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert abs(float(z["w0"]) - w0) <= 1e-5 * max(1.0, abs(w0))
        own = np.arange(r, N, world)
        ws[own] = z["w"][own]
        vs[:, own] = z["v"][:, own]
    np.testing.assert_allclose(ws, w, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(vs, v, rtol=1e-4, atol=1e-6)


# ---- pipelined schedule (all-reduce of batch b+1 under the update of batch b): deterministic "one batch stale" rule ----
PN, PK, PNNZ, PROWS, PB = 6400, 16, 8, 6000, 1024


def _pipe_inputs():
    from oracle import oracle as O
    tr = O.synth_rows(77, 0, PROWS, PNNZ, PN)
    rng = np.random.default_rng(3)
    return tr, rng.normal(0, 0.05, PN), rng.normal(0, 0.05, (PK, PN))


def _pipe_worker(rank, world, port, out_dir, backend):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from libfm_amd import capi
    from libfm_amd.distributed import ShardedSGD
    dev = rank if (backend == "nccl" and world > 1) else 0      # RCCL needs one GPU per rank; gloo shards share cuda:0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev), rank=rank, world_size=world)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    tr, w, v = _pipe_inputs()
    h = capi.Handle(PN, PK, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.05, -1.0, 1.0, device=dev,
                    shard_rank=rank, shard_world=world)
    h.set_params(0.1, w, v)
    h.upload_rows(0, tr.entries, tr.row_ptr, tr.target)
    drv = ShardedSGD(h, 0, PROWS, PB, 64, capi.APPLY_DEFAULT, capi.FLAG_BIAS_LAG, backend, pipeline=True)
    for _ in range(2):
        drv.epoch()
    drv.synchronize()
    w0, wo, vo = h.get_params()
    np.savez(os.path.join(out_dir, "p%d.npz" % rank), w0=w0, w=wo, v=vo)
    h.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 1), ("nccl", 2), ("nccl", 4)])
def test_pipelined_schedule_matches_its_oracle(tmp_path, oracle, backend, world):
    import torch
    if backend == "nccl" and world > torch.cuda.device_count():
        pytest.skip("needs %d GPUs (RCCL refuses two ranks on one device)" % world)
    """gloo x 2 shards: the order of operations of the pipelined driver; nccl x 1: the same with the asynchronous RCCL
    all-reduce and its stream dependencies.  Both must equal oracle fmo_sgd_epoch_minibatch_pipelined, and must differ
    from the unpipelined rule (otherwise the test would not see the schedule)."""
    import torch.multiprocessing as mp
    O = oracle
    port = 29900 + (os.getpid() % 90) + (7 if backend == "nccl" else 0)
    mp.spawn(_pipe_worker, args=(world, port, str(tmp_path), backend), nprocs=world, join=True)
    tr, w, v = _pipe_inputs()
    ms = []
    for pipelined in (True, False):
        m = O.Model(PN, PK, True, True, 0.0, 0.0, 0.001)
        m.w0, m.w[:], m.v[:] = 0.1, w.astype(np.float32), v.astype(np.float32)
        for _ in range(2):
            O.sgd_epoch_minibatch(m, tr, 1, 0.05, -1.0, 1.0, PB, 64, True, pipelined=pipelined)
        ms.append(m)
    m, m_plain = ms
    ws, vs = np.zeros(PN), np.zeros((PK, PN))
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "p%d.npz" % r))
        assert abs(float(z["w0"]) - m.w0) <= 1e-4 * max(1.0, abs(m.w0))
        own = np.arange(r, PN, world)
        ws[own] = z["w"][own]
        vs[:, own] = z["v"][:, own]
    np.testing.assert_allclose(ws, m.w, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(vs, m.v, rtol=1e-4, atol=2e-6)
    assert np.abs(m.v - m_plain.v).max() > 1e-4 * np.abs(m.v).max()

"""probe: the per-rank compute side of the P-way feature-sharded step on ONE GPU (no exchange): rank 0 of `world`
holds n/world features and sees every example restricted to them (about nnz/world entries per example)."""
import os, sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
import torch
from libfm_amd import capi

n, k, nnz, rows, B = 100_000_000, 64, 32, 1 << 22, 131072
if len(sys.argv) > 3:
    k = int(sys.argv[3])
if len(sys.argv) > 4:
    B = int(sys.argv[4])          # a factor slice of a 2-D shard grid: k / P_f factors per rank
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 0           # 0: the library's default micro-chunk
lag = int(os.environ.get("LAG", "2"))                          # fmx_sgd_opts::bias_lag of the split step
worlds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 4, 2, 1]
for world in worlds:
    h = capi.Handle(n, k, True, True, 1, 0, 0, 0.001, 0.01, -1, 1, shard_rank=0, shard_world=world,
                    place_candidates=int(os.environ.get("PLACE", "0")))     # fmx_config::place_candidates (1 = plain allocations)
    pi = h.place_info()
    h.init_params(0, 0.01, 1)
    h.synth_rows(0, 123, 0, rows, nnz)
    kp1 = h.info().k_padded + 1
    st = torch.cuda.Stream()
    bufs = [torch.empty(B * kp1, dtype=torch.float32, device="cuda") for _ in range(2)]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def epoch(which):
        with torch.cuda.stream(st):
            for i, row0 in enumerate(range(0, rows, B)):
                buf = bufs[i & 1]
                if which in ("both", "partial"):
                    h.sgd_partial(0, row0, B, buf.data_ptr(), st.cuda_stream)
                if which in ("both", "finish"):
                    h.sgd_finish(0, row0, B, buf.data_ptr(), capi.APPLY_DEFAULT, chunk, st.cuda_stream, B, capi.FLAG_BIAS_LAG, lag)
    res = {}
    for which in ("both", "partial", "finish"):
        epoch(which); st.synchronize(); h.synchronize()
        t0 = time.perf_counter(); epoch(which); st.synchronize(); h.synchronize(); res[which] = time.perf_counter() - t0
    print("placement %d (%d chunks, %d + %d)  " % (pi.method, pi.chunks, pi.per_class[0], pi.per_class[1]), end="")
    print("k=%d B=%d chunk=%d lag=%d " % (k, B, chunk, lag) + "world=%d: gather+update %.1f Mex/s (gather alone %.1f, update alone %.1f) per rank"
          % (world, rows / res["both"] / 1e6, rows / res["partial"] / 1e6, rows / res["finish"] / 1e6), flush=True)
    h.close()

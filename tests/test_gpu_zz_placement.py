"""fmx_create's placement of big parameter tables (an arena of 1 GiB chunks from two memory classes, fmx_config::place_candidates):
what it reports, that results do not depend on it, that it gives its memory back.

(The file sorts LAST on purpose: round 6 saw ONE abort inside fmx_create in test_candidate_bound_is_honoured in twelve runs of the whole
suite -- SIGABRT from below the C-ABI, output lost to the capture; eighteen repetitions of this file alone and five more suite runs did
not bring it back (scripts/calls/r6_call25.sh .. _call27.sh).  Until it is understood, whatever it is must not take the tests behind it
along.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, K, NNZ = 10_000_000, 64, 32          # V = 2.56 GB: the smallest size the arena is used for is 2 GiB


@pytest.fixture(scope="module")
def capi():
    from libfm_amd import capi
    return capi


def _run(capi, place):
    h = capi.Handle(N, K, True, True, capi.TASK_CLASSIFICATION, 0.0, 0.0, 0.001, 0.01, -1.0, 1.0, place_candidates=place)
    pi = h.place_info()
    h.init_params(0.0, 0.05, 5)
    rows = 1 << 18
    h.synth_rows(0, 31, 0, rows, NNZ)
    h.sgd_epoch(0, capi.SGD_MINIBATCH, capi.APPLY_FUSED, 65536, 0, 0, 2)
    p = h.predict(0, rows)
    ids = np.array([0, 1, N // 2, N - 2, N - 1], dtype=np.uint64)       # both ends of both tables
    w, v = h.get_param_rows(ids)
    out = (p.tobytes(), w.tobytes(), v.tobytes(), h.get_w0())
    h.close()
    return pi, out


def test_arena_placement_is_reported_and_changes_no_result(capi):
    capi.release_cached_memory()                                       # (an arena an earlier test left behind would be taken over: pool = 0)
    pi0, out0 = _run(capi, 0)
    pi1, out1 = _run(capi, 1)
    assert pi1.method == 0 and pi1.chunks == 0
    assert pi0.method == 2, "the virtual-memory API should be available on this device"
    assert pi0.chunks == 4                                              # 2.56 GB of V, w across the boundary of chunks 2 | 3
    assert pi0.per_class[0] + pi0.per_class[1] <= pi0.chunks and pi0.per_class[0] >= 2
    assert pi0.pool >= pi0.chunks and pi0.classes_seen >= 1
    if pi0.classes_seen >= 2:
        assert abs(int(pi0.per_class[0]) - int(pi0.per_class[1])) <= 1   # balanced when a second class was within the pool's bound
    assert out0 == out1                                                 # deterministic rule: bit-identical wherever the tables live


def test_arena_gives_its_memory_back(capi):
    """ten handles in a row: leaked chunks (4 GiB each, plus the pool) would show"""
    import torch
    capi.release_cached_memory()
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(10):
        h = capi.Handle(N, K, place_candidates=0)
        pi = h.place_info()
        assert pi.method == 2 and pi.chunks == 4
        assert (pi.pool == 0) == (i > 0)                              # from the second handle on: the previous one's arena, nothing probed
        h.close()
    held = free0 - torch.cuda.mem_get_info()[0]
    assert (3 << 30) < held < (5 << 30)                               # exactly one arena is kept for the next fmx_create on this device ...
    capi.release_cached_memory()
    assert free0 - torch.cuda.mem_get_info()[0] < (1 << 30)           # ... until it is asked back


def test_a_cached_arena_serves_a_smaller_table_and_moves_no_number(capi):
    """a handle of 8 chunks, destroyed; the next handle needs 4: it takes the first four chunks of that arena (a prefix of an alternating
    sequence alternates), the other four go back to the device; its results are the ones of a freshly probed arena"""
    import torch
    capi.release_cached_memory()
    _, fresh = _run(capi, 0)
    capi.release_cached_memory()
    free0 = torch.cuda.mem_get_info()[0]
    big = capi.Handle(2 * N + N // 2, K, place_candidates=0)
    assert big.place_info().chunks >= 7
    big.close()
    pi, out = _run(capi, 0)
    assert pi.method == 2 and pi.chunks == 4 and pi.pool == 0 and pi.per_class[0] == 2 and pi.per_class[1] == 2
    assert out == fresh
    assert (3 << 30) < free0 - torch.cuda.mem_get_info()[0] < (5 << 30)
    capi.release_cached_memory()


def test_candidate_bound_is_honoured(capi):
    capi.release_cached_memory()
    h = capi.Handle(N, K, place_candidates=2)
    pi = h.place_info()
    assert pi.method == 2 and pi.pool <= 2 * pi.chunks + 64
    h.close()
    with pytest.raises(capi.FmxError):
        capi.Handle(N, K, place_candidates=7)

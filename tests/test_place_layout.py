"""CPU: the layout of the arena fmx_create places big parameter tables in (fmx_place_layout, host arithmetic): V from offset 0, w
centred on a chunk boundary behind it so that it lies in both memory classes like V does (chunks alternate between two classes)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def capi():
    from libfm_amd import build, capi
    build.build()
    return capi


def _check(capi, v_bytes, w_bytes):
    ch, t, off = capi.place_layout(v_bytes, w_bytes)
    assert ch == 1 << 30
    assert off >= v_bytes                                   # w behind V
    assert off % 256 == 0                                   # aligned like V's rows
    assert off + w_bytes <= t * ch                          # everything inside the arena
    assert t * ch - (off + w_bytes) < ch + 256              # ... and no spare chunk behind w
    w_half = -(-(w_bytes // 2) // 256) * 256
    assert (off + w_half) % ch == 0                         # a chunk boundary in the middle of w: half of it in either class
    assert off - v_bytes < ch + w_bytes                     # the gap between V and w is less than a chunk (+ w's own half)
    return ch, t, off


def test_bench_shape(capi):
    n, kp = 100_000_000, 64
    ch, t, off = _check(capi, n * kp * 4, n * 4)
    assert t == 26 and off == 25 * ch - 200_000_000 // 256 * 256    # V ends in chunk 23, w straddles the start of chunk 25 (0-based)


def test_smallest_and_largest(capi):
    _check(capi, 2 << 30, (2 << 30) // 64)                  # the smallest V the arena is used for (k = 64)
    n = 2 ** 32 - 1                                         # the reference's uint limit on ids, k = 256
    ch, t, off = _check(capi, n * 256 * 4, n * 4)
    assert t == 4112                                        # 4 TiB of V + 16 GiB of w: the arithmetic holds far beyond any device


def test_random_sizes(capi):
    rng = np.random.default_rng(7)
    for _ in range(2000):
        kp = int(rng.choice([1, 2, 8, 32, 64, 128, 256]))
        n = int(rng.integers(1 << 20, 1 << 32))
        _check(capi, n * kp * 4, n * 4 if rng.integers(0, 8) else 0)


def test_bad_arguments(capi):
    with pytest.raises(capi.FmxError):
        capi.place_layout(0, 4)

"""The reference's random initialisation, reproduced draw for draw (host side).

libFM seeds libc with srand(seed) (src/libfm/libfm.cpp:115-116) and fills V with mean + stdev * ran_gaussian()
(fm_model.h:96 -> matrix.h:398-404, f outer / j inner), ran_gaussian being Leva's ratio-of-uniforms method on
ran_uniform() = rand() / (RAND_MAX + 1) (src/util/random.h:148-174).  Calling the SAME libc rand() through ctypes
gives the same stream, so `-seed S` yields bit-identical initial parameters (verified against the reference's
dumps in tests/test_refrand.py).  Only meant for the sizes a text-format data set has; large synthetic tables use
fmx_init_params instead."""
import ctypes
import math

import numpy as np

_libc = ctypes.CDLL("libc.so.6")
_libc.rand.restype = ctypes.c_int
RAND_MAX = 2147483647


def srand(seed):
    _libc.srand(ctypes.c_uint(seed & 0xFFFFFFFF))


def ran_uniform():
    return _libc.rand() / (RAND_MAX + 1.0)                     # random.h:172-174


def ran_gaussian():
    """random.h:148-162 (Joseph L. Leva: A fast normal random number generator)"""
    while True:
        while True:
            u = ran_uniform()
            if u != 0.0:
                break
        v = 1.7156 * (ran_uniform() - 0.5)
        x = u - 0.449871
        y = abs(v) + 0.386595
        q = x * x + y * (0.19600 * y - 0.25472 * x)
        if q < 0.27597:
            break
        if not ((q > 0.27846) or ((v * v) > (-4.0 * u * u * math.log(u)))):
            break
    return v / u


def ran_gaussian_ms(mean, stdev):
    if stdev == 0.0 or math.isnan(stdev):                       # random.h:164-170
        return mean
    return mean + stdev * ran_gaussian()


def init_v(num_factor, num_attribute, mean, stdev):
    """DMatrixDouble::init(mean, stdev), matrix.h:398-404: row-major fill of the k x n block"""
    v = np.empty((num_factor, num_attribute), dtype=np.float64)
    for f in range(num_factor):
        row = v[f]
        for j in range(num_attribute):
            row[j] = ran_gaussian_ms(mean, stdev)
    return v


def init_w_normal(num_attribute, mean, stdev):
    """DVectorDouble::init_normal (matrix.h:392-396), used for mcmc/als (libfm.cpp:283)"""
    return np.array([ran_gaussian_ms(mean, stdev) for _ in range(num_attribute)], dtype=np.float64)

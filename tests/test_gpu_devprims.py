"""The two device-wide primitives of the index build on the GPU (csrc/rt_devprims.h): the stable
radix sort of (64-bit key, 32-bit value) pairs and the exclusive prefix sum of 64-bit values,
against numpy on sizes around the tile and wave boundaries, heavy ties, and a few million keys."""
import numpy as np
import pytest

import raytracing_amd as rt

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4096 + 17, 100003, 3 * 1048576 + 5]


def test_device_radix_sort_is_stable_and_complete():
    r = rt.Renderer(8, 8)
    rng = np.random.default_rng(7)
    for n in SIZES:
        for kind in ("random", "ties", "sorted", "reversed", "high_bits"):
            if kind == "random":
                k = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
            elif kind == "ties":
                k = rng.integers(0, 5, size=n, dtype=np.uint64) << np.uint64(40)
            elif kind == "sorted":
                k = np.arange(n, dtype=np.uint64) * np.uint64(0x10001)
            elif kind == "reversed":
                k = (np.uint64(n) - np.arange(n, dtype=np.uint64)) << np.uint64(33)
            else:
                k = rng.integers(0, 4, size=n, dtype=np.uint64) << np.uint64(62) | rng.integers(0, 3, size=n, dtype=np.uint64)
            v = np.arange(n, dtype=np.uint32)
            ks, vs = r.selftest_device_sort(k, v)
            order = np.argsort(k, kind="stable")
            assert np.array_equal(ks, k[order]), (n, kind)
            assert np.array_equal(vs, v[order]), (n, kind)           # equal keys keep their order
            if n > 200000 and kind != "random":
                break                                                # one large case per size is enough
    r.close()


def test_device_exclusive_sum():
    r = rt.Renderer(8, 8)
    rng = np.random.default_rng(9)
    for n in SIZES:
        # two packed 32-bit counters per value, as the clustering rounds of the build use them
        x = rng.integers(0, 2, size=n, dtype=np.uint64) | (rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(32))
        got = r.selftest_device_scan(x)
        want = np.concatenate([[np.uint64(0)], np.cumsum(x, dtype=np.uint64)[:-1]]) if n else x
        assert np.array_equal(got, want), n
        big = rng.integers(0, 1 << 40, size=n, dtype=np.uint64)
        got = r.selftest_device_scan(big)
        want = np.concatenate([[np.uint64(0)], np.cumsum(big, dtype=np.uint64)[:-1]]) if n else big
        assert np.array_equal(got, want), n
    r.close()

"""Sobol table and sequence: the library and the oracle against the reference's own data
(its v_matrix.txt, a data file identical to the V[] array of sobol.h:8-25) and against the
values recorded from the reference's CPU sobol() (sobol.h:37-50)."""
import ctypes as C
import json
import os

import numpy as np

import oracle_lib as ol
import raytracing_amd as rt


def _golden_table(golden_dir):
    txt = open(os.path.join(golden_dir, "sobol_v_matrix.txt")).read().replace(",", " ")
    v = np.array([int(x) for x in txt.split()], dtype=np.uint64)
    assert v.shape == (512,)
    return v.astype(np.uint32)


def test_default_table_is_the_reference_table(golden_dir):
    assert np.array_equal(rt.sobol_default_table(), _golden_table(golden_dir))


def test_first_dimension_is_van_der_corput():
    v = rt.sobol_default_table()
    assert [int(x) for x in v[:32]] == [1 << (31 - j) for j in range(32)]


def test_sobol_known_answers(golden_dir):
    ka = json.load(open(os.path.join(golden_dir, "sobol_known_answers.json")))
    v = _golden_table(golden_dir)
    lib = ol.load()
    vp = v.ctypes.data_as(C.c_void_p)
    assert np.float32(0xFFFFFFFF) == np.float32(ka["float_of_0xFFFFFFFF"])
    for n, g in ka["gray"].items():
        assert lib.or_gray_code(int(n)) == g
        assert rt.lib.rt_gray_code(int(n)) == g
    for n, row in ka["rows"].items():
        g = lib.or_gray_code(int(n))
        for d, want in enumerate(row):
            assert lib.or_sobol(vp, d, g) == np.float32(want)
            assert rt.lib.rt_sobol(vp, d, g) == np.float32(want)
    g = lib.or_gray_code(1024)
    for d, want in enumerate(ka["n1024_d0_15"]):
        got = lib.or_sobol(vp, d, g)
        assert abs(got - want) < 5e-10 * max(1.0, abs(want)) + 6e-10, (d, got, want)   # printed with 9 digits
        assert rt.lib.rt_sobol(vp, d, g) == got


def test_library_sobol_equals_oracle_on_many_indices(golden_dir):
    v = _golden_table(golden_dir)
    lib = ol.load()
    vp = v.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(3)
    idx = np.concatenate([np.arange(0, 300), rng.integers(0, 2**31, size=500)])
    for i in idx:
        for d in range(16):
            assert rt.lib.rt_sobol(vp, d, int(i)) == lib.or_sobol(vp, d, int(i))


def test_rand_origin_stream_is_deterministic_and_in_range():
    a = rt.rand_origins(20221003, 1000)
    b = rt.rand_origins(20221003, 1000)
    assert np.array_equal(a, b)
    assert a.min() >= 874264.0 and a.max() < 2 * 874264.0
    assert len(np.unique(a)) > 990


def test_pixel_recovered_from_its_coordinate_is_the_pixel():
    """frag:62-63 turns the interpolated coordinate back into a pixel index,
    uint(((uv*2-1)*0.5+0.5)*width) with uv = (X+0.5)/width.  The library passes X itself; that is
    the same number for every X < width <= 16384 (all widths the library accepts).  Multiplying by 2
    and by 0.5 is exact, so the fused and unfused readings agree and numpy's fp32 reproduces both."""
    half, one, two = np.float32(0.5), np.float32(1.0), np.float32(2.0)
    for w in range(1, 16385):
        x = np.arange(w, dtype=np.float32)
        wf = np.float32(w)
        uv = (x + half) / wf
        back = (((uv * two - one) * half + half) * wf).astype(np.uint32)
        assert np.array_equal(back, np.arange(w, dtype=np.uint32)), w

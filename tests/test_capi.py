"""The C-ABI library: loads, exports every symbol the header declares, and its host-side and
argument-checking paths behave (no GPU compute calls here)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

import raytracing_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = _declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(rt.lib, n), n


def test_header_compiles_as_c_and_struct_sizes_match():
    prog = r'''
#include "rt_amd.h"
#include <stdio.h>
int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(RtTriangle), sizeof(RtLinearBVHNode), sizeof(RtSphere),
                        sizeof(RtCamera), sizeof(RtStats), sizeof(RtOptions), sizeof(RtCameraRig)); return 0; }
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                               os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    assert [int(x) for x in out] == [144, 36, 40, 72, 48, 44, C.sizeof(rt.RtCameraRig)]
    assert C.sizeof(rt.RtStats) == 48 and C.sizeof(rt.RtOptions) == 44 and C.sizeof(rt.RtCameraRig) == 116


def test_no_oracle_in_the_product():
    """The shipped library and package must not reference the oracle."""
    for root, _, files in os.walk(os.path.join(ROOT, "raytracing_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and "oracle" not in txt, f
    deps = subprocess.check_output(["ldd", rt.LIB_PATH]).decode()
    assert "liboracle" not in deps and "libamdhip64" in deps


def test_host_side_argument_checks():
    assert rt.lib.rt_sobol_default_table(None) == -1
    x, y = C.c_int(), C.c_int()
    assert rt.lib.rt_texture_dims(-1, 33, C.byref(x), C.byref(y)) == -1
    assert rt.lib.rt_camera_default(None, 10, 10) == -1
    cam = np.zeros(1, dtype=rt.CAMERA_DTYPE)
    assert rt.lib.rt_camera_default(cam.ctypes.data, 0, 10) == -1
    h = C.c_void_p()
    assert rt.lib.rt_create(C.byref(h), 0, 10, 0) == -1
    assert rt.lib.rt_create(C.byref(h), 100000, 10, 0) == -1
    assert b"16384" in rt.lib.rt_last_error(None)
    assert rt.lib.rt_render_frames(None, 1, 1, None) == -1
    assert rt.lib.rt_version().startswith(b"rt_amd")
    o = rt.RtOptions()
    assert rt.lib.rt_default_options(C.byref(o)) == 0
    assert (o.max_bounces, o.pipeline, o.cull, o.count_work, o.accel, o.history8) == (4, rt.PIPELINE_PERSISTENT, 1, 0, 1, 0)


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    h = C.c_void_p()
    rc = rt.lib.rt_create(C.byref(h), 64, 64, 0)
    assert rc == -3 and not h.value
    assert b"no CPU path" in rt.lib.rt_last_error(None) or b"HIP" in rt.lib.rt_last_error(None)

"""Builds librt_amd.so (HIP kernels for gfx950 + C++ host code) in-tree with hipcc.

The library cross-compiles without a GPU.  -ffp-contract=off is part of the arithmetic
contract of csrc/rt_math.h: the sources spell out every fused multiply-add themselves.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librt_amd.so")
SOURCES = ["rt_device.hip", "rt_bvh.cpp", "rt_host.cpp", "rt_scene.cpp", "rt_accel.cpp"]
MAIN = "rt_main.cpp"
HEADERS = ["rt_internal.h", "rt_kernels.h", "rt_math.h", "rt_accel.h", "rt_traverse.h", "rt_shade.h", os.path.join("..", "..", "include", "rt_amd.h")]
# -fno-slp-vectorize: the packed fp32 instructions the SLP vectorizer forms (v_pk_mul_f32, v_pk_add_f32)
# measured 1-2 % slower than the scalar forms in the path kernel.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS + [MAIN]] + [os.path.abspath(__file__)]
    if not os.path.exists(os.path.join(HERE, "rt_render")):
        return True
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES] + ["-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    # rt_render: the reference's main loop on top of the C ABI (plain C++ host program)
    cxx = os.environ.get("CXX", "g++")
    main_cmd = [cxx, "-O2", "-std=c++17", "-Wall", os.path.join(CSRC, "rt_main.cpp"), "-o", os.path.join(HERE, "rt_render"),
                "-L" + HERE, "-lrt_amd", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(main_cmd), flush=True)
    subprocess.check_call(main_cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

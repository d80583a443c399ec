"""Builds librt_amd.so (HIP kernels for gfx950 + C++ host code) in-tree with hipcc.

The library cross-compiles without a GPU.  -ffp-contract=off is part of the arithmetic
contract of csrc/rt_math.h: the sources spell out every fused multiply-add themselves.
Every source is compiled to its own object (only the stale ones), then linked.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "librt_amd.so")
SOURCES = ["rt_device.hip", "rt_gpubuild.hip", "rt_bvh.cpp", "rt_host.cpp", "rt_scene.cpp", "rt_accel.cpp"]
MAIN = "rt_main.cpp"
HEADERS = ["rt_internal.h", "rt_kernels.h", "rt_math.h", "rt_accel.h", "rt_gpubuild.h", "rt_traverse.h", "rt_shade.h",
           os.path.join("..", "..", "include", "rt_amd.h")]
# -fno-slp-vectorize: the packed fp32 instructions the SLP vectorizer forms (v_pk_mul_f32, v_pk_add_f32)
# measured 1-2 % slower than the scalar forms in the path kernel.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _newest_dep():
    deps = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    return max(os.path.getmtime(d) for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    dep_t = _newest_dep()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), dep_t):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])
    objs = [os.path.join(OBJ, s + ".o") for s in SOURCES]

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(run, jobs))
    if jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-lpthread"])
    # rt_render: the reference's main loop on top of the C ABI (plain C++ host program)
    exe = os.path.join(HERE, "rt_render")
    main_src = os.path.join(CSRC, MAIN)
    if force or not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(main_src), os.path.getmtime(LIB)):
        cxx = os.environ.get("CXX", "g++")
        run([cxx, "-O2", "-std=c++17", "-Wall", main_src, "-o", exe, "-L" + HERE, "-lrt_amd", "-Wl,-rpath,$ORIGIN"])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

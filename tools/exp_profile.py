#!/usr/bin/env python3
"""Experiment: prints the wave-level block counts of the path kernel (iterations, node / triangle /
finish block executions and the lanes active in each).  Needs a library built with the counters:

  hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -DRT_EXP_PROFILE \
        -o /tmp/librt_profile.so raytracing_amd/csrc/rt_{device.hip,bvh.cpp,host.cpp,scene.cpp,accel.cpp} -lpthread
  RT_AMD_LIBRARY=/tmp/librt_profile.so python3 tools/exp_profile.py

(the counters reuse the work-counter fields, so that build reports no box/triangle statistics).
With -DRT_EXP_TOPK=K the "test blocks" line counts node steps at the first K nodes of the index instead,
with -DRT_EXP_EMPTY node steps in which no child was hit (measured: 1.39 of 9.45 per ray on the
1.31M-triangle scene, 1.63 of 10.75 on 10.5M: deferred nodes that lie beyond the closest hit by the time
they are taken up, and rays that pass between the children of a node; with -DRT_EXP_EMPTY=2 only the
former are counted: 0.37 resp. 0.46 per ray, so remembering a deferred node's entry distance and skipping
it when it is taken up could save 4 % of the node steps at most)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt
# arguments: [copies of the model] [bounces] [index build mode: 0 host, 1 GPU, 2 by size]
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bounces = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sc = rt.Scene(); sc.push_synthetic(0, 8, copies, 12345)
r = rt.make_renderer(sc, 1920, 1080, 0, max_bounces=bounces, pipeline=2, index_build=int(sys.argv[3]) if len(sys.argv) > 3 else 2)
print("index nodes", r.scene_info()["index_nodes"] if hasattr(r, "scene_info") else "?")
ro = rt.rand_origins(20221003, 8)
r.render(1, ro); r.synchronize()
st = r.stats()
rays = st["rays"]
def split(v): return v & 0xFFFFFFFF, v >> 32
for name, key in (("iterations", "node_tests"), ("node blocks", "tri_tests"), ("test blocks (top-node steps in an RT_EXP_TOPK build)", "tri_hits"), ("go blocks", "retraced")):
    n, lanes = split(st[key])
    print(f"{name:22.22s} count {n:10d}  lanes {lanes:12d}  lanes/block {lanes / max(n, 1):5.1f}  blocks per ray {n / rays:.4f}  lane-steps per ray {lanes / rays:.2f}")
print("rays", rays)

#!/usr/bin/env python3
"""Times the host-side stages before the first frame: reference BVH build and scene upload
(device layout + search index build + copies)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt
for kind, level, copies in ((0, 8, 1), (0, 8, 8)):
    sc = rt.Scene(); sc.push_synthetic(kind, level, copies, 12345)
    t0 = time.time(); tris, nodes = rt.bvh_build(sc.triangles()); t1 = time.time()
    r = rt.Renderer(64, 64); r.upload(tris, nodes); r.synchronize(); t2 = time.time()
    print(f"{tris.shape[0]} triangles: reference BVH build {t1 - t0:.2f} s, upload {t2 - t1:.2f} s")
    r.close()

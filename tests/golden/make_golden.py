#!/usr/bin/env python3
"""Regenerates the oracle-made golden vectors in this directory.

The reference has no tests or recorded outputs and its shader cannot be executed where this
repository is built, so these vectors are produced by the CPU oracle (oracle/rt_oracle.c,
fused arithmetic contract, fixed sin/cos sequence) from the reference's own scene data
(objects/room.obj, objects/cube.obj, main.cpp:103-110).  They pin the oracle against
accidental change and give the GPU tests committed expected values; they are not outputs of
the reference itself.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle_lib as ol      # noqa: E402
import raytracing_amd as rt  # noqa: E402

W, H, FRAMES, SEED = 96, 64, 4, 20221003


def main():
    sc = rt.Scene()
    sc.push_obj(os.path.join(HERE, "objects", "room.obj"), (5.0, 5.0, 5.0), (0.0, 5.0, 0.0))
    sc.push_obj(os.path.join(HERE, "objects", "cube.obj"), (0.5, 5.0, -1.0), (0.0, 1.0, -2.0))
    sc.push_sphere((0, 2.0, 1), 1.5, (0.9, 0.9, 0.9), 0.0, 0.0, 0)
    sc.push_sphere((-3, 2.0, 1), 1.5, (0.2, 0.2, 0.9), 0.0, 1.0, 0)
    tris, nodes = ol.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(SEED, FRAMES)
    osc = ol.scene_from_build(tris, nodes, W, H, rt.sobol_default_table(), sc.spheres())
    hist = np.zeros((H, W, 3), dtype=np.float32)
    first = None
    for k in range(FRAMES):
        hist, cur, _ = osc.render_frame(cam, k + 1, float(ro[k]), hist)
        if k == 0:
            first = cur
    rng = np.random.default_rng(11)
    o = rng.uniform(-9, 4, size=(512, 3)).astype(np.float32)
    o[:, 1] = rng.uniform(0.5, 9.5, size=512)
    d = rng.normal(size=(512, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t, ids = osc.hit_bvh(o, d)
    np.savez_compressed(os.path.join(HERE, "demo_96x64.npz"), rand_origin=ro, frame1=first, mean4=hist,
                        ray_o=o, ray_d=d, ray_t=t, ray_tri=ids,
                        tris=tris.view(np.uint8), nodes=nodes.view(np.uint8))
    print("wrote demo_96x64.npz", first.mean(), hist.mean(), (ids >= 0).mean())


if __name__ == "__main__":
    main()

"""GPU parity at the benchmark sizes: the three synthetic workloads of bench.py at their full
triangle counts and resolutions, i.e. the kernel variant, tree depth, stack overflow path and
frame batching that the benchmark actually runs.

The oracle is too slow for every pixel of these images, so each scene is checked through
  (i)   a 16-row band of two consecutive frames, bit for bit against the oracle's rows;
  (ii)  50 000 rays (random, axis-parallel, starting inside the mesh, starting far outside):
        triangle index and distance equal to the oracle's hitBVH;
  (iii) the whole image and the ray count of the default path (search index + proof) equal to
        those of the reference-order traversal of the same kernel family;
  (iv)  the share of rays that needed the exact re-trace;
  (v)   the kernel variant (waves per SIMD) and the stack split the scene was given."""
import numpy as np
import pytest

import oracle_lib as ol
import raytracing_amd as rt
import scenes

pytestmark = pytest.mark.gpu

# name: (kind, level, copies, width, height, bounces, band of rows) -- as bench.py's WORKLOADS
CASES = {
    "dragon1m": (0, 8, 1, 1920, 1080, 4, (520, 536)),
    "sponza": (1, 64, 1, 3840, 2160, 4, (700, 716)),
    "stress10m": (0, 8, 8, 1920, 1080, 8, (500, 516)),
    # three copies: 3.9M triangles, 7.9M nodes -- the largest of this family that the reference's
    # BVH texture can express: it stores node indices as floats and the shader recovers them with
    # int(f + 0.5) (frag:180-182), which is exact only below 2^23.  The 10.5M-triangle scene is
    # beyond that; the oracle then takes the same integers from an int array (oracle_lib).
    "stress3m9": (0, 8, 3, 1920, 1080, 8, (560, 576)),
}


def _rays_for(tris, seed, n=50000):
    rng = np.random.default_rng(seed)
    lo = tris["v"].reshape(-1, 3).min(axis=0)
    hi = tris["v"].reshape(-1, 3).max(axis=0)
    o, d = scenes.random_rays(n, seed, extent=1.0, axis_aligned_fraction=0.1)
    o = (lo + (o * 0.5 + 0.5) * (hi - lo)).astype(np.float32)
    # a quarter of the rays start at the centre of a mesh triangle, pushed along a random direction
    # by a small step: inside or just outside the surface
    k = n // 4
    pick = rng.integers(0, tris.shape[0], size=k)
    c = tris["v"][pick].mean(axis=1)
    o[n - k:] = (c + d[n - k:] * rng.uniform(-0.05, 0.05, size=(k, 1))).astype(np.float32)
    # some start far outside the scene (the index hands those to the exact path)
    o[n // 2:n // 2 + 500] *= np.float32(50.0)
    return o, d


@pytest.mark.parametrize("name,index", [("dragon1m", "auto"), ("dragon1m", "gpu"), ("sponza", "auto"),
                                        ("stress10m", "auto"), ("stress10m", "gpu"), ("stress3m9", "auto")])
def test_benchmark_workload_at_full_size(name, index):
    """`index`: who builds the search index.  The default (auto) builds it on the host below
    rt.INDEX_AUTO_GPU_TRIANGLES triangles and on the device from there on - what the benchmark
    runs; "host" / "gpu" force the other builder on the scenes where auto does not take it."""
    kind, level, copies, W, H, bounces, band = CASES[name]
    sc = rt.Scene()
    sc.push_synthetic(kind, level, copies, 12345)
    tris, nodes = rt.bvh_build(sc.triangles())
    sph = sc.spheres()
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(20221003, 2)
    osc = ol.scene_from_build(tris, nodes, W, H, rt.sobol_default_table(), sph, max_bounces=bounces)

    build = {"auto": rt.INDEX_BUILD_AUTO, "host": rt.INDEX_BUILD_HOST, "gpu": rt.INDEX_BUILD_GPU}[index]
    r = rt.Renderer(W, H)
    r.set_options(index_build=build)
    r.upload(tris, nodes)
    r.set_spheres(sph)
    r.set_camera(cam)
    r.set_options(max_bounces=bounces, index_build=build)   # defaults: persistent pipeline, search index
    info = r.scene_info()
    assert info["index_built_on_device"] == int(index == "gpu" or (index == "auto" and tris.shape[0] >= rt.INDEX_AUTO_GPU_TRIANGLES))
    # (v) what the benchmark runs: the 8-wave variant (7 from 4M triangles on), an index deeper than its LDS stack
    assert info["triangles"] == tris.shape[0] >= 65536
    assert info["index_enabled"] == 1 and info["paths_waves_per_simd"] == (7 if tris.shape[0] >= (4 << 20) else 8)
    assert info["index_stack_lds"] == (16 if tris.shape[0] >= (4 << 20) else 14)
    assert info["index_stack_need"] > info["index_stack_lds"] > 0
    assert info["tree_height"] >= 18

    # (ii) rays
    o, d = _rays_for(tris, 5 + len(name))
    t_want, id_want = osc.hit_bvh(o, d)
    t, ids = r.trace_rays(o, d)
    assert np.array_equal(ids, id_want)
    assert np.array_equal(t, t_want)
    assert 0.3 < (id_want >= 0).mean()

    # (i) + (iii): two frames through the default path (the counters restart: the rays above
    # included axis-parallel and far-away ones, which are re-traced by design)
    r.reset()
    r.render(1, ro)
    img = r.read_mean()
    s = r.stats()
    assert s["frames"] == 2
    hist = np.zeros((H, W, 3), dtype=np.float32)
    for k in range(2):
        hist, _, _ = osc.render_frame(cam, k + 1, float(ro[k]), hist, rows=band)
    y0, y1 = band
    assert np.array_equal(np.isnan(img[y0:y1]), np.isnan(hist[y0:y1]))
    m = ~np.isnan(hist[y0:y1])
    assert np.array_equal(img[y0:y1][m], hist[y0:y1][m])
    assert np.nan_to_num(hist[y0:y1]).max() > 0
    # (iv)
    assert s["retraced"] <= 1e-4 * s["rays"], (s["retraced"], s["rays"])

    r.set_options(max_bounces=bounces, accel=False)         # the reference's tree in the reference's order
    r.reset()
    r.render(1, ro)
    img_ref = r.read_mean()
    s_ref = r.stats()
    assert s_ref["rays"] == s["rays"] and s_ref["retraced"] == 0
    assert np.array_equal(np.isnan(img), np.isnan(img_ref))
    m = ~np.isnan(img_ref)
    assert np.array_equal(img[m], img_ref[m])
    r.close()


@pytest.mark.parametrize("case", ["one", "two", "three", "duplicates", "random", "dragon1m"])
def test_index_built_on_the_gpu(case):
    """RtOptions.index_build = GPU: the search index is a Morton-order tree built on the device.
    Whatever the index, the proof step keeps the reference's answers: rays and images must equal
    the oracle's (and the host-built index's), including scenes of one to three triangles and
    triangles with identical Morton codes."""
    W, H = 96, 64
    if case == "dragon1m":
        sc = rt.Scene()
        sc.push_synthetic(0, 8, 1, 12345)
        t0 = sc.triangles()
    elif case == "duplicates":
        t0 = scenes.random_triangles(200, 9, snap=0.5)
        t0 = np.concatenate([t0, t0[:50], t0[:50]])         # coincident triangles: equal keys, exact ties
    else:
        t0 = scenes.random_triangles({"one": 1, "two": 2, "three": 3, "random": 5000}[case], 17, extent=3.0, size=1.5)
    tris, nodes = rt.bvh_build(t0)
    osc = ol.scene_from_build(tris, nodes, W, H, rt.sobol_default_table(), None)
    r = rt.Renderer(W, H)
    r.set_options(index_build=rt.INDEX_BUILD_GPU)
    r.upload(tris, nodes)
    info = r.scene_info()
    assert info["index_enabled"] == 1
    if case == "dragon1m":
        assert info["index_nodes"] > 300000
    o, d = (_rays_for(tris, 3, n=30000) if case == "dragon1m" else scenes.random_rays(20000, 5, extent=4.0))
    t_want, id_want = osc.hit_bvh(o, d)
    t, ids = r.trace_rays(o, d)
    assert np.array_equal(ids, id_want) and np.array_equal(t, t_want)
    cam = rt.camera_default(W, H) if case == "dragon1m" else rt.camera_from_angles((0.0, 0.0, 8.0), -90.0, 0.0, 30.0, W, H)
    ro = rt.rand_origins(3, 2)
    want, st = osc.render(cam, 1, ro, want_stats=True)
    r.set_camera(cam)
    for pipeline in (rt.PIPELINE_PERSISTENT, rt.PIPELINE_WAVEFRONT):
        r.set_options(pipeline=pipeline, index_build=rt.INDEX_BUILD_GPU)
        r.reset()
        r.render(1, ro)
        got = r.read_mean()
        assert np.array_equal(np.isnan(got), np.isnan(want))
        m = ~np.isnan(want)
        assert np.array_equal(got[m], want[m])
        s = r.stats()
        assert s["rays"] == st["rays"]
        if case in ("random", "dragon1m"):
            assert s["retraced"] <= 1e-3 * s["rays"] + 5
    # uploading again with the host build replaces the index
    r.set_options(index_build=rt.INDEX_BUILD_HOST)
    r.upload(tris, nodes)
    t, ids = r.trace_rays(o, d)
    assert np.array_equal(ids, id_want) and np.array_equal(t, t_want)
    r.close()


def test_index_stack_overflow_and_small_lds_stack(monkeypatch):
    """The deferred-node stack of the index search keeps its first entries in LDS and the rest in
    a global column.  With only two LDS entries every deep traversal overflows; results must not
    change (random scene with heavy overlap, so that many nodes are deferred)."""
    monkeypatch.setenv("RT_AMD_ACCEL_LDS_ENTRIES", "2")
    tris0 = scenes.random_triangles(6000, 31, extent=3.0, size=2.5)
    tris, nodes = rt.bvh_build(tris0)
    osc = ol.scene_from_build(tris, nodes, 64, 48, rt.sobol_default_table(), None)
    r = rt.Renderer(64, 48)
    r.upload(tris, nodes)
    info = r.scene_info()
    assert info["index_stack_lds"] == 2 and info["index_stack_need"] > 2
    o, d = scenes.random_rays(20000, 3, extent=3.0)
    t_want, id_want = osc.hit_bvh(o, d)
    t, ids = r.trace_rays(o, d)
    assert np.array_equal(ids, id_want) and np.array_equal(t, t_want)
    cam = rt.camera_from_angles((0.0, 0.0, 7.0), -90.0, 0.0, 30.0, 64, 48)
    ro = rt.rand_origins(8, 2)
    want, st = osc.render(cam, 1, ro, want_stats=True)
    for pipeline in (rt.PIPELINE_PERSISTENT, rt.PIPELINE_WAVEFRONT):
        r.set_options(pipeline=pipeline)
        r.set_camera(cam)
        r.reset()
        r.render(1, ro)
        got = r.read_mean()
        assert np.array_equal(np.isnan(got), np.isnan(want))
        m = ~np.isnan(want)
        assert np.array_equal(got[m], want[m])
        assert r.stats()["rays"] == st["rays"]
    r.close()


def test_tree_with_foreign_boxes_is_rendered_in_reference_order():
    """A caller may upload a tree whose boxes are not the builder's (looser boxes here).  The
    proof step of the index relies on exact nested min/max boxes, so such a scene must fall back
    to the reference-order traversal and still match the oracle run on the same tree."""
    tris0 = scenes.random_triangles(500, 12, extent=3.0, size=1.0)
    tris, nodes = rt.bvh_build(tris0)
    loose = nodes.copy()
    loose["minBound"] -= np.float32(0.25)
    loose["maxBound"] += np.float32(0.25)
    osc = ol.scene_from_build(tris, loose, 64, 48, rt.sobol_default_table(), None)
    r = rt.Renderer(64, 48)
    r.upload(tris, loose)
    assert r.scene_info()["index_enabled"] == 0
    o, d = scenes.random_rays(10000, 4, extent=4.0)
    t_want, id_want = osc.hit_bvh(o, d)
    t, ids = r.trace_rays(o, d)
    assert np.array_equal(ids, id_want) and np.array_equal(t, t_want)
    # a box that does not contain its child: also no index
    broken = nodes.copy()
    k = int(np.nonzero(nodes["triangleNum"] > 4)[0][1])
    broken["maxBound"][k] = broken["minBound"][k]
    r.upload(tris, broken)
    assert r.scene_info()["index_enabled"] == 0
    osc2 = ol.scene_from_build(tris, broken, 64, 48, rt.sobol_default_table(), None)
    t_want, id_want = osc2.hit_bvh(o, d)
    t, ids = r.trace_rays(o, d)
    assert np.array_equal(ids, id_want) and np.array_equal(t, t_want)
    r.upload(tris, nodes)
    assert r.scene_info()["index_enabled"] == 1
    r.close()


def test_two_contexts_with_different_sobol_tables():
    """setSobol is per context: two contexts on one GPU with different direction numbers,
    rendered alternately, each match the oracle run with their own table."""
    W, H = 96, 64
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(77, 4)
    va = rt.sobol_default_table()
    vb = np.random.default_rng(5).integers(0, 1 << 32, size=va.shape[0], dtype=np.uint64).astype(np.uint32)   # any other table
    want = {}
    for key, v in (("a", va), ("b", vb)):
        want[key], _ = ol.scene_from_build(tris, nodes, W, H, v, sc.spheres()).render(cam, 1, ro)
    assert not np.array_equal(want["a"], want["b"])
    for pipeline in (rt.PIPELINE_PERSISTENT, rt.PIPELINE_WAVEFRONT, rt.PIPELINE_MEGAKERNEL):
        ra = rt.make_renderer(sc, W, H, pipeline=pipeline)
        rb = rt.make_renderer(sc, W, H, pipeline=pipeline)
        rb.set_sobol(vb)
        for k in range(len(ro)):                            # interleaved, one frame at a time
            ra.render(k + 1, ro[k:k + 1])
            rb.render(k + 1, ro[k:k + 1])
        assert np.array_equal(ra.read_mean(), want["a"]), pipeline
        assert np.array_equal(rb.read_mean(), want["b"]), pipeline
        ra.close(); rb.close()

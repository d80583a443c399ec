"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances: integer results (triangle indices, ray and test counts) must be equal; images are
compared bit for bit where the arithmetic contract makes that possible, and against the
north-star bar (>= 99.9 % of pixels within 1e-3 at 1 spp, RMSE <= 1e-3 after many frames)
everywhere."""
import numpy as np
import pytest

import oracle_lib as ol
import raytracing_amd as rt
import scenes

pytestmark = pytest.mark.gpu


def _oracle_for(tris, nodes, W, H, spheres, **kw):
    return ol.scene_from_build(tris, nodes, W, H, rt.sobol_default_table(), spheres, **kw)


def _pass_rate(a, b, tol=1e-3):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max(axis=2)
    d = np.where(np.isnan(a).any(axis=2) & np.isnan(b).any(axis=2), 0.0, d)
    return float((d <= tol).mean())


def test_device_arithmetic_matches_host():
    rng = np.random.default_rng(1)
    n = 1 << 20
    a = rng.normal(size=n).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, size=n).astype(np.float32)
    b = rng.normal(size=n).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, size=n).astype(np.float32)
    r = rt.Renderer(8, 8)
    out = r.selftest_arith(a, b)
    with np.errstate(all="ignore"):
        assert np.array_equal(out[:, 0], a / b)
        assert np.array_equal(out[:, 1], np.sqrt(np.abs(a)))
        fma = (a.astype(np.float64) * b.astype(np.float64) + a.astype(np.float64)).astype(np.float32)
    # double rounding can differ from a true fma in rare cases; allow 1 ulp on those
    same = out[:, 2] == fma
    assert same.mean() > 0.999
    # sin/cos on the range the sampler uses, [0, 2*pi], against the oracle's fixed sequence
    import ctypes as C
    lib = ol.load()
    m = 1 << 16
    th = np.concatenate([rng.uniform(0.0, 6.2831855, size=m - 5), [0.0, 1.5707964, 3.1415927, 4.712389, 6.2831855]]).astype(np.float32)
    out = r.selftest_arith(np.ones(m, dtype=np.float32), th)
    s, c = C.c_float(), C.c_float()
    for i in range(m):
        lib.or_sincos(C.c_float(float(th[i])), C.byref(s), C.byref(c))
        assert out[i, 3] == np.float32(s.value) and out[i, 4] == np.float32(c.value), (i, th[i])
        assert abs(float(out[i, 3]) - np.sin(float(th[i]))) < 3e-7 and abs(float(out[i, 4]) - np.cos(float(th[i]))) < 3e-7


def test_device_reciprocal_and_square_root_are_ieee_for_every_input():
    """rt_math.h computes 1/x, sqrt(x) and 1/sqrt(x) with one Newton step on the hardware estimate
    where that gives the IEEE result and with the compiler's full expansion elsewhere; the
    oracle uses the CPU's division and sqrtf.  All 2^32 inputs, on the device; likewise the pixel
    coordinate (X + 0.5) / W for every pixel of every width up to 16384."""
    r = rt.Renderer(8, 8)
    assert r.selftest_exact_math() == [0, 0, 0, 0]
    r.close()


MODES = [dict(accel=False, cull=False), dict(accel=False, cull=True), dict(accel=True, cull=True)]


@pytest.mark.parametrize("mode", MODES)
def test_trace_rays_demo(mode):
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    r = rt.Renderer(8, 8)
    r.upload(tris, nodes)
    r.set_options(**mode)
    osc = _oracle_for(tris, nodes, 8, 8, None)
    o, d = scenes.random_rays(20000, 7, extent=9.0)
    o[:, 1] = np.abs(o[:, 1])
    t, ids = r.trace_rays(o, d)
    to, io = osc.hit_bvh(o, d)
    assert np.array_equal(ids, io)
    assert np.array_equal(t, to)


@pytest.mark.parametrize("n,snap", [(1, None), (2, None), (3, 0.5), (7, 0.5), (100, 0.25), (3000, 0.125), (3000, None)])
def test_trace_rays_random_scenes(n, snap):
    tris0 = scenes.random_triangles(n, 100 + n, snap=snap)
    tris, nodes = rt.bvh_build(tris0)
    osc = _oracle_for(tris, nodes, 8, 8, None)
    o, d = scenes.random_rays(6000, n)
    to, io = osc.hit_bvh(o, d)
    for mode in MODES:
        r = rt.Renderer(8, 8)
        r.upload(tris, nodes)
        r.set_options(**mode)
        t, ids = r.trace_rays(o, d)
        assert np.array_equal(ids, io), (n, snap, mode)
        assert np.array_equal(t, to), (n, snap, mode)


@pytest.mark.parametrize("pipeline,cull,accel", [(rt.PIPELINE_MEGAKERNEL, False, False), (rt.PIPELINE_MEGAKERNEL, True, False),
                                                 (rt.PIPELINE_WAVEFRONT, False, False), (rt.PIPELINE_WAVEFRONT, True, False),
                                                 (rt.PIPELINE_WAVEFRONT, True, True), (rt.PIPELINE_PERSISTENT, True, True),
                                                 (rt.PIPELINE_PERSISTENT, True, False), (rt.PIPELINE_PERSISTENT, False, False)])
def test_demo_one_spp_and_running_mean(pipeline, cull, accel):
    W, H = 200, 120
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(20221003, 6)
    osc = _oracle_for(tris, nodes, W, H, sc.spheres())
    r = rt.Renderer(W, H)
    r.upload(tris, nodes)
    r.set_spheres(sc.spheres())
    r.set_camera(cam)
    r.set_options(pipeline=pipeline, cull=cull, count_work=True, accel=accel)
    # 1 spp
    r.render(1, ro[:1])
    got = r.read_mean()
    want, _, st = osc.render_frame(cam, 1, float(ro[0]), want_stats=True)
    assert _pass_rate(got, want) >= 0.999
    assert np.array_equal(got, want), f"bit-exact fraction {(got == want).all(axis=2).mean()}"
    s = r.stats()
    assert s["rays"] == st["rays"]
    assert s["tri_hits"] == st["triHits"]
    if accel:
        assert s["retraced"] < 0.01 * s["rays"]
    elif not cull:
        assert s["node_tests"] == st["nodeTests"]
        assert s["tri_tests"] == st["triTests"]
    else:
        assert s["node_tests"] <= st["nodeTests"]
    # frames 2..6 on top: the reference's running mean
    r.render(2, ro[1:])
    got = r.read_mean()
    want, _ = osc.render(cam, 1, ro)
    assert np.array_equal(got, want)
    ssum = r.read_sum() / np.float32(len(ro))
    assert np.sqrt(np.mean((ssum - want) ** 2)) < 1e-5


def test_persistent_pipeline_frame_batches():
    """150 frames in one call: the persistent pipeline renders them 64 per launch (three launches,
    the last one partial) and must reproduce the frame-by-frame running mean and sum."""
    W, H = 72, 40
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(99, 150)
    want, _ = _oracle_for(tris, nodes, W, H, sc.spheres()).render(cam, 1, ro)
    res = {}
    for pipeline in (rt.PIPELINE_PERSISTENT, rt.PIPELINE_WAVEFRONT):
        r = rt.Renderer(W, H)
        r.upload(tris, nodes)
        r.set_spheres(sc.spheres())
        r.set_camera(cam)
        r.set_options(pipeline=pipeline)
        r.render(1, ro)
        res[pipeline] = (r.read_mean(), r.read_sum(), r.stats()["rays"])
        r.close()
    assert np.array_equal(res[rt.PIPELINE_PERSISTENT][0], want)
    assert np.array_equal(res[rt.PIPELINE_PERSISTENT][0], res[rt.PIPELINE_WAVEFRONT][0])
    assert np.array_equal(res[rt.PIPELINE_PERSISTENT][1], res[rt.PIPELINE_WAVEFRONT][1])
    assert res[rt.PIPELINE_PERSISTENT][2] == res[rt.PIPELINE_WAVEFRONT][2]


@pytest.mark.parametrize("pipeline", [rt.PIPELINE_PERSISTENT, rt.PIPELINE_WAVEFRONT, rt.PIPELINE_MEGAKERNEL])
def test_eight_bit_history_emulation(pipeline):
    """history8: after every frame the running mean is what an 8-bit GL_RGB attachment keeps of
    it (RT_Screen.h:61).  The oracle is fed its own quantised history frame by frame."""
    W, H = 64, 40
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(4, 9)
    osc = _oracle_for(tris, nodes, W, H, sc.spheres())
    hist = np.zeros((H, W, 3), dtype=np.float32)
    for k in range(len(ro)):
        out, _, _ = osc.render_frame(cam, k + 1, float(ro[k]), hist)
        c = np.where(out > 0, np.minimum(out, np.float32(1.0)), np.float32(0.0)).astype(np.float32)   # NaN -> 0
        hist = (np.floor(c * np.float32(255.0) + np.float32(0.5)) / np.float32(255.0)).astype(np.float32)
    r = rt.Renderer(W, H)
    r.upload(tris, nodes)
    r.set_spheres(sc.spheres())
    r.set_camera(cam)
    r.set_options(pipeline=pipeline, history8=True)
    r.render(1, ro[:4]); r.render(5, ro[4:])
    got = r.read_mean()
    assert np.array_equal(got, hist)
    levels = np.unique(np.round(got.astype(np.float64) * 255.0))
    assert levels.min() >= 0 and levels.max() <= 255
    # the fp32 image of the same frames differs: samples reach 4*pi, and the 8-bit history clamps
    # every frame's mean to 1 (a property of the reference's storage, not an error bound)
    r.set_options(pipeline=pipeline, history8=False)
    r.reset()
    r.render(1, ro)
    ref = r.read_mean()
    assert not np.array_equal(ref, got)


def _render_all_pipelines(tris, nodes, cam, ro, W, H, spheres=None, bounces=4):
    out = {}
    for name, pipeline, accel in (("paths", rt.PIPELINE_PERSISTENT, True), ("paths_ref_order", rt.PIPELINE_PERSISTENT, False),
                                  ("wavefront", rt.PIPELINE_WAVEFRONT, True), ("mega", rt.PIPELINE_MEGAKERNEL, False)):
        r = rt.Renderer(W, H)
        r.upload(tris, nodes)
        if spheres is not None:
            r.set_spheres(spheres)
        r.set_camera(cam)
        r.set_options(max_bounces=bounces, pipeline=pipeline, accel=accel)
        r.render(1, ro)
        out[name] = (r.read_mean(), r.stats()["rays"])
        r.close()
    return out


def _same_image(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


@pytest.mark.parametrize("case", ["one_triangle", "two_triangles", "degenerate", "camera_inside_light", "all_miss", "far_scene"])
def test_small_and_awkward_scenes(case):
    """Scenes at the edges of the data structures: a root that is a leaf, zero-area and
    repeated-vertex triangles, a camera inside an emissive box, nothing in view, coordinates
    around 1e4 (the index boxes grow with the scene scale)."""
    W, H = 48, 32
    rng = np.random.default_rng(len(case))
    if case == "one_triangle":
        t = scenes.random_triangles(1, 3, extent=0.5, size=3.0)
        t["v"][0] = [[-4, -3, -6], [4, -3, -6], [0, 6, -6]]
    elif case == "two_triangles":
        t = scenes.random_triangles(2, 4, extent=0.5, size=3.0)
        t["v"][0] = [[-4, -3, -6], [4, -3, -6], [0, 6, -6]]
        t["v"][1] = [[-4, -3, -5], [4, -3, -5], [0, 6, -5]]
    elif case == "degenerate":
        t = scenes.random_triangles(200, 5, extent=3.0, size=1.5)
        t["v"][:20, 1] = t["v"][:20, 0]                     # repeated vertex: zero area
        t["v"][20:40, 2] = (t["v"][20:40, 0] + t["v"][20:40, 1]) * np.float32(0.5)     # collinear
        t["v"][40:50] = t["v"][40:50, :1]                   # a point
    elif case == "camera_inside_light":
        t = scenes.random_triangles(60, 6, extent=6.0, size=6.0)
        t["ka"][:] = (0.7, 0.8, 0.9)
    elif case == "all_miss":
        t = scenes.random_triangles(50, 7, extent=2.0, size=1.0)
        t["v"][:, :, 2] += 40.0                             # behind the camera
    else:
        t = scenes.random_triangles(500, 8, extent=4.0, size=1.5)
        t["v"] = (t["v"] * np.float32(100.0) + np.float32(10000.0)).astype(np.float32)
    tris, nodes = rt.bvh_build(t)
    if case == "far_scene":
        cam = rt.camera_from_angles((10000.0, 10000.0, 11500.0), -90.0, 0.0, 45.0, W, H)
    else:
        cam = rt.camera_default(W, H)
    ro = rt.rand_origins(31, 3)
    want, st = _oracle_for(tris, nodes, W, H, None).render(cam, 1, ro, want_stats=True)
    res = _render_all_pipelines(tris, nodes, cam, ro, W, H)
    for name, (img, rays) in res.items():
        assert _same_image(img, want), (case, name)
        assert rays == st["rays"], (case, name)
    if case == "all_miss":
        assert not want.any()
    if case in ("one_triangle", "camera_inside_light", "far_scene"):
        assert np.nan_to_num(want).max() > 0 or case == "one_triangle"


def test_resume_from_saved_accumulation():
    """rt_restore: 3 frames, save mean + sum, a new context restores them and renders 3 more:
    the same images as 6 frames in one go."""
    W, H = 80, 48
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(12, 6)

    def fresh():
        r = rt.Renderer(W, H)
        r.upload(tris, nodes)
        r.set_spheres(sc.spheres())
        r.set_camera(cam)
        return r
    r = fresh()
    r.render(1, ro)
    want_mean, want_sum = r.read_mean(), r.read_sum()
    r.close()
    r = fresh()
    r.render(1, ro[:3])
    mean3, sum3 = r.read_mean(), r.read_sum()
    assert r.stats()["frames"] == 3
    r.close()
    r = fresh()
    r.restore(mean3, sum3, 3)
    assert r.stats()["frames"] == 3
    r.render(4, ro[3:])
    assert np.array_equal(r.read_mean(), want_mean) and np.array_equal(r.read_sum(), want_sum)
    assert r.stats()["frames"] == 6
    with pytest.raises(rt.RtError):
        r.restore(None, None, 1 << 31)


def test_1024_spp_running_mean_rmse():
    """The accumulation bar: after 1024 frames the running mean equals the oracle's (RMSE 0, far
    inside the 1e-3 bound), the sum / N agrees with it to rounding, and the image has converged
    towards the 256-frame one."""
    W, H, N = 96, 54, 1024
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(20221003, N)
    r = rt.Renderer(W, H)
    r.upload(tris, nodes)
    r.set_spheres(sc.spheres())
    r.set_camera(cam)
    r.render(1, ro[:256])
    at256 = r.read_mean()
    r.render(257, ro[256:])
    got = r.read_mean()
    want, _ = _oracle_for(tris, nodes, W, H, sc.spheres()).render(cam, 1, ro)
    rmse = float(np.sqrt(np.mean((got.astype(np.float64) - want.astype(np.float64)) ** 2)))
    assert rmse <= 1e-3 and np.array_equal(got, want), rmse
    assert float(np.sqrt(np.mean((r.read_sum() / np.float32(N) - want) ** 2))) < 1e-4
    assert r.stats()["frames"] == N
    assert float(np.abs(got - at256).mean()) < 0.05


def test_golden_vectors(golden_dir):
    """Committed expected values (made by the oracle from the reference's scene files)."""
    import os
    g = np.load(os.path.join(golden_dir, "demo_96x64.npz"))
    tris = g["tris"].view(rt.TRIANGLE_DTYPE)
    nodes = g["nodes"].view(rt.NODE_DTYPE)
    sc = scenes.demo()
    cam = rt.camera_default(96, 64)
    for pipeline, accel in ((rt.PIPELINE_WAVEFRONT, True), (rt.PIPELINE_WAVEFRONT, False), (rt.PIPELINE_MEGAKERNEL, False),
                            (rt.PIPELINE_PERSISTENT, True), (rt.PIPELINE_PERSISTENT, False)):
        r = rt.Renderer(96, 64)
        r.upload(tris, nodes)
        r.set_spheres(sc.spheres())
        r.set_camera(cam)
        r.set_options(pipeline=pipeline, accel=accel)
        r.render(1, g["rand_origin"][:1])
        assert np.array_equal(r.read_mean(), g["frame1"])
        r.render(2, g["rand_origin"][1:])
        assert np.array_equal(r.read_mean(), g["mean4"])
        t, ids = r.trace_rays(g["ray_o"], g["ray_d"])
        assert np.array_equal(t, g["ray_t"]) and np.array_equal(ids, g["ray_tri"])
        r.close()


def test_upload_from_reference_textures_matches_struct_upload():
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    tt, tw, tb, bw = rt.pack_textures(tris, nodes)
    cam = rt.camera_default(64, 48)
    ro = rt.rand_origins(3, 2)
    imgs = []
    for via_tex in (False, True):
        r = rt.Renderer(64, 48)
        if via_tex:
            r.upload_textures(tt, tris.shape[0], tb, nodes.shape[0])
        else:
            r.upload(tris, nodes)
        r.set_spheres(sc.spheres())
        r.set_camera(cam)
        r.render(1, ro)
        imgs.append(r.read_mean())
    assert np.array_equal(imgs[0], imgs[1])


@pytest.mark.parametrize("W,H", [(1, 1), (7, 5), (8, 8), (9, 17), (130, 70)])
def test_image_sizes_not_multiple_of_tile(W, H):
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(11, 3)
    want, _ = _oracle_for(tris, nodes, W, H, sc.spheres()).render(cam, 1, ro)
    for pipeline in (rt.PIPELINE_WAVEFRONT, rt.PIPELINE_MEGAKERNEL, rt.PIPELINE_PERSISTENT):
        r = rt.Renderer(W, H)
        r.upload(tris, nodes)
        r.set_spheres(sc.spheres())
        r.set_camera(cam)
        r.set_options(pipeline=pipeline)
        r.render(1, ro)
        assert np.array_equal(r.read_mean(), want), (W, H, pipeline)


@pytest.mark.parametrize("nspheres", [0, 1, 10])
@pytest.mark.parametrize("bounces", [1, 4, 8])
def test_spheres_and_bounce_limits(nspheres, bounces):
    """0, 1 and the maximum of 10 spheres with all three sphere materials; 1, 4 and 8 bounces
    (8 uses all 16 Sobol dimensions)."""
    W, H = 96, 64
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    rng = np.random.default_rng(nspheres)
    sph = np.zeros(nspheres, dtype=rt.SPHERE_DTYPE)
    for i in range(nspheres):
        sph[i]["center"] = rng.uniform([-8, 0.5, -4], [4, 8, 3])
        sph[i]["radius"] = rng.uniform(0.4, 1.3)
        sph[i]["albedo"] = rng.uniform(0.2, 0.9, size=3)
        sph[i]["specularRate"] = [0.0, 1.0, 0.3][i % 3]
        sph[i]["refractionRate"] = [0.0, 1.0, 0.6][(i + 1) % 3]
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(17, 2)
    want, _ = _oracle_for(tris, nodes, W, H, sph, max_bounces=bounces).render(cam, 1, ro)
    for pipeline in (rt.PIPELINE_WAVEFRONT, rt.PIPELINE_MEGAKERNEL, rt.PIPELINE_PERSISTENT):
        r = rt.Renderer(W, H)
        r.upload(tris, nodes)
        r.set_spheres(sph)
        r.set_camera(cam)
        r.set_options(max_bounces=bounces, pipeline=pipeline)
        r.render(1, ro)
        got = r.read_mean()
        assert _pass_rate(got, want) >= 0.999
        assert np.array_equal(got, want), (nspheres, bounces, pipeline, (got == want).all(axis=2).mean())


def test_mixed_material_scene_and_arbitrary_camera():
    """Random triangles with diffuse / mirror / glass / mixed thresholds and lights, camera from
    angles (not axis aligned), NaN and miss paths included."""
    W, H = 120, 80
    tris0 = scenes.random_triangles(400, 77, extent=5.0, size=1.5)
    tris, nodes = rt.bvh_build(tris0)
    cam = rt.camera_from_angles((0.5, 0.3, 9.0), -95.0, -3.0, 40.0, W, H)
    ro = rt.rand_origins(5, 3)
    sph = np.zeros(1, dtype=rt.SPHERE_DTYPE)
    sph[0]["center"] = (0, 0, 2); sph[0]["radius"] = 1.0; sph[0]["albedo"] = (0.8, 0.8, 0.8); sph[0]["specularRate"] = 0.5; sph[0]["refractionRate"] = 0.8
    want, st = _oracle_for(tris, nodes, W, H, sph).render(cam, 1, ro, want_stats=True)
    for pipeline, accel in ((rt.PIPELINE_WAVEFRONT, True), (rt.PIPELINE_WAVEFRONT, False), (rt.PIPELINE_MEGAKERNEL, False),
                            (rt.PIPELINE_PERSISTENT, True), (rt.PIPELINE_PERSISTENT, False)):
        r = rt.Renderer(W, H)
        r.upload(tris, nodes)
        r.set_spheres(sph)
        r.set_camera(cam)
        r.set_options(pipeline=pipeline, accel=accel)
        r.render(1, ro)
        got = r.read_mean()
        assert np.array_equal(np.isnan(got), np.isnan(want))
        m = ~np.isnan(want)
        assert np.array_equal(got[m], want[m]), (pipeline, accel)
        assert r.stats()["rays"] == st["rays"]


def test_search_index_fallbacks():
    """Rays the index must hand to the exact path: axis-parallel directions (1/d infinite), NaN
    directions, origins far outside the scene, and exact ties between coincident triangles."""
    tris0 = scenes.random_triangles(300, 5, snap=0.5)
    tris0 = np.concatenate([tris0, tris0[:40]])            # 40 duplicated triangles: exact ties
    tris, nodes = rt.bvh_build(tris0)
    osc = _oracle_for(tris, nodes, 8, 8, None)
    o, d = scenes.random_rays(4000, 9, axis_aligned_fraction=0.25)
    d[1000:1010] = np.nan
    o[1100:1200] *= 100.0
    o[1200:1300] = 0.0                                      # many rays from one point
    to, io = osc.hit_bvh(o, d)
    r = rt.Renderer(8, 8)
    r.upload(tris, nodes)
    r.set_options(accel=True)
    t, ids = r.trace_rays(o, d)
    assert np.array_equal(ids, io) and np.array_equal(t, to)
    assert (io >= 0).mean() > 0.15


def test_synthetic_mesh_parity_and_work_counters():
    """5k-triangle displaced icosphere in the room (the benchmark scene at low subdivision)."""
    W, H = 160, 90
    sc = rt.Scene()
    sc.push_synthetic(0, 4, 1, 12345)
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(20221003, 2)
    want, st = _oracle_for(tris, nodes, W, H, None).render(cam, 1, ro, want_stats=True)
    for accel, pipeline in ((True, rt.PIPELINE_WAVEFRONT), (False, rt.PIPELINE_WAVEFRONT),
                            (True, rt.PIPELINE_PERSISTENT), (False, rt.PIPELINE_PERSISTENT)):
        r = rt.Renderer(W, H)
        r.upload(tris, nodes)
        r.set_camera(cam)
        r.set_options(accel=accel, cull=False, count_work=True, pipeline=pipeline)
        r.render(1, ro)
        assert np.array_equal(r.read_mean(), want)
        s = r.stats()
        assert s["rays"] == st["rays"] and s["frames"] == 2
        if accel:
            assert s["node_tests"] < 0.6 * st["nodeTests"]       # the index tests far fewer boxes
            assert s["retraced"] <= 0.001 * s["rays"] + 5
        else:
            assert s["node_tests"] == st["nodeTests"] and s["tri_tests"] == st["triTests"] and s["tri_hits"] == st["triHits"]


@pytest.mark.parametrize("occ,entries", [(5, None), (6, None), (7, None), (8, None), (8, 3), (7, 1)])
def test_every_occupancy_variant_of_the_path_kernel(occ, entries, monkeypatch):
    """The path kernel is compiled for 5, 6, 7 and 8 waves per SIMD (different register budgets,
    different spills) and picks one by scene size; RT_AMD_PATHS_OCC forces one.  Each must render
    the oracle's image, also with an LDS stack so short that deferred nodes overflow to memory."""
    W, H = 160, 90
    sc = rt.Scene()
    sc.push_synthetic(0, 4, 1, 12345)
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(20221003, 3)
    want, _ = _oracle_for(tris, nodes, W, H, sc.spheres()).render(cam, 1, ro)
    monkeypatch.setenv("RT_AMD_PATHS_OCC", str(occ))
    if entries is not None:
        monkeypatch.setenv("RT_AMD_ACCEL_LDS_ENTRIES", str(entries))
    r = rt.Renderer(W, H)
    r.upload(tris, nodes)
    r.set_spheres(sc.spheres())
    r.set_camera(cam)
    info = r.scene_info()
    assert info["paths_waves_per_simd"] == occ
    assert info["index_stack_lds"] == (entries if entries is not None else min(info["index_stack_need"], 14 if occ == 8 else 16))
    r.render(1, ro)
    assert _same_image(r.read_mean(), want)
    assert r.stats()["retraced"] <= 0.001 * r.stats()["rays"] + 5


@pytest.mark.parametrize("frames_per_pass", [3, 8])
def test_wavefront_pipeline_with_frame_batches(frames_per_pass, monkeypatch):
    """RT_AMD_WAVEFRONT_FRAMES lets the per-bounce pipeline push several frames through its queues
    at a time (an image size that is no multiple of the tile, a frame count that is no multiple of
    the batch, spheres, mixed materials): same image, same ray count as the oracle."""
    W, H = 130, 70
    sc = rt.Scene()
    sc.push_synthetic(1, 6, 1, 12345)
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(20221003, 7)
    want, st = _oracle_for(tris, nodes, W, H, sc.spheres()).render(cam, 1, ro, want_stats=True)
    monkeypatch.setenv("RT_AMD_WAVEFRONT_FRAMES", str(frames_per_pass))
    for accel in (True, False):
        r = rt.Renderer(W, H)
        r.upload(tris, nodes)
        r.set_spheres(sc.spheres())
        r.set_camera(cam)
        r.set_options(pipeline=rt.PIPELINE_WAVEFRONT, accel=accel)
        r.render(1, ro[:5])
        r.render(6, ro[5:])
        assert _same_image(r.read_mean(), want), accel
        assert r.stats()["rays"] == st["rays"] and r.stats()["frames"] == 7
        r.close()


@pytest.mark.parametrize("kind,level,copies,bounces", [(1, 6, 1, 4), (0, 3, 8, 8)])
def test_benchmark_scene_families_at_low_subdivision(kind, level, copies, bounces):
    """The other two benchmark scenes in small: the column hall (diffuse / mirror / glass columns,
    two spheres, room scaled x3) and eight copies of the displaced mesh with 8 bounces."""
    W, H = 128, 72
    sc = rt.Scene()
    sc.push_synthetic(kind, level, copies, 12345)
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(20221003, 3)
    want, st = _oracle_for(tris, nodes, W, H, sc.spheres(), max_bounces=bounces).render(cam, 1, ro, want_stats=True)
    res = _render_all_pipelines(tris, nodes, cam, ro, W, H, spheres=sc.spheres(), bounces=bounces)
    for name, (img, rays) in res.items():
        assert _same_image(img, want), (kind, name)
        assert rays == st["rays"], (kind, name)
    assert np.nan_to_num(want).max() > 0


def test_full_size_properties_1080p():
    """At the benchmark resolution the oracle is too slow for every pixel; check properties:
    both pipelines and both traversal modes give the same image; rendering frames in two calls
    equals one call; the fp32 sum equals N * mean up to rounding; a band of rows matches the oracle."""
    W, H = 1920, 1080
    sc = rt.Scene()
    sc.push_synthetic(0, 5, 1, 12345)
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(20221003, 4)
    imgs = {}
    for name, pipeline, accel in (("wf_accel", rt.PIPELINE_WAVEFRONT, True), ("wf_exact", rt.PIPELINE_WAVEFRONT, False),
                                  ("mega", rt.PIPELINE_MEGAKERNEL, False), ("paths", rt.PIPELINE_PERSISTENT, True)):
        r = rt.Renderer(W, H)
        r.upload(tris, nodes)
        r.set_camera(cam)
        r.set_options(pipeline=pipeline, accel=accel)
        if name == "wf_accel":
            r.render(1, ro[:1]); r.render(2, ro[1:])
        else:
            r.render(1, ro)
        imgs[name] = r.read_mean()
        if name == "wf_accel":
            s = r.read_sum()
            assert np.allclose(s / 4.0, imgs[name], rtol=1e-6, atol=1e-6)
        r.close()
    assert np.array_equal(imgs["wf_accel"], imgs["wf_exact"])
    assert np.array_equal(imgs["wf_accel"], imgs["mega"])
    assert np.array_equal(imgs["wf_accel"], imgs["paths"])
    osc = _oracle_for(tris, nodes, W, H, None)
    hist = np.zeros((H, W, 3), dtype=np.float32)
    for k in range(4):
        hist, _, _ = osc.render_frame(cam, k + 1, float(ro[k]), hist, rows=(500, 516))
    assert np.array_equal(imgs["wf_accel"][500:516], hist[500:516])


def test_render_from_loaded_obj_files_with_reference_camera(golden_dir, tmp_path):
    """main.cpp:103-110 end to end: the reference's own room.obj / cube.obj through the OBJ/MTL
    loader, its two spheres, and the camera exactly as Camera.h:18-46 computes it (front.x =
    cos(radians(-90)) is 4.4e-8 there, not 0: rt_camera_default's axis-aligned frame is an
    idealisation), against the oracle; plus a multi-material OBJ with a group split."""
    import os
    W, H = 150, 100                                         # the reference's 1200x800 aspect
    sc = rt.Scene()
    sc.push_obj(os.path.join(golden_dir, "objects", "room.obj"), (5.0, 5.0, 5.0), (0.0, 5.0, 0.0))
    sc.push_obj(os.path.join(golden_dir, "objects", "cube.obj"), (0.5, 5.0, -1.0), (0.0, 1.0, -2.0))
    sc.push_sphere((0.0, 2.0, 1.0), 1.5, (0.9, 0.9, 0.9), 0.0, 0.0)
    sc.push_sphere((-3.0, 2.0, 1.0), 1.5, (0.2, 0.2, 0.9), 0.0, 1.0)
    rig = rt.CameraRig(W, H)
    cam = rig.uniforms()
    assert cam[0]["front"][0] != 0.0 and abs(cam[0]["front"][0]) < 1e-6     # not the idealised frame
    ro = rt.rand_origins(20221003, 3)
    tris, nodes = rt.bvh_build(sc.triangles())
    want, st = _oracle_for(tris, nodes, W, H, sc.spheres()).render(cam, 1, ro, want_stats=True)
    res = _render_all_pipelines(tris, nodes, cam, ro, W, H, spheres=sc.spheres())
    for name, (img, rays) in res.items():
        assert _same_image(img, want), name
        assert rays == st["rays"], name
    assert np.nan_to_num(want).max() > 1.0
    # the camera after some interaction (mouse look, zoom, a step forward): LoopNum restarts at 1
    rig.update_front(600.0, 400.0); rig.update_front(640.0, 380.0)
    rig.update_fov(3.0)
    rig.process_keyboard(rt.CameraRig.FORWARD, 0.1)
    cam2 = rig.uniforms()
    assert not np.array_equal(cam2[0]["front"], cam[0]["front"])
    want2, _ = _oracle_for(tris, nodes, W, H, sc.spheres()).render(cam2, 1, ro[:2])
    res2 = _render_all_pipelines(tris, nodes, cam2, ro[:2], W, H, spheres=sc.spheres())
    for name, (img, _) in res2.items():
        assert _same_image(img, want2), name

    # a second model: several materials in one object, a group split, a light of its own
    (tmp_path / "m.mtl").write_text("newmtl lamp\nKa 1 1 1\nKd 0 0 0\nKs 0 0 0\n\nnewmtl mirror\nKd 1 1 1\nKs 0 1 0\n\n"
                                    "newmtl glass\nKd 1 1 1\nKs 0 0 0\n\nnewmtl half\nKd 0.7 0.6 0.5\nKs 0.5 0.8 0\n")
    (tmp_path / "m.obj").write_text(
        "mtllib m.mtl\nv -3 0 -4\nv 3 0 -4\nv 3 5 -4\nv -3 5 -4\nv -3 0 2\nv 3 0 2\nv 3 5 2\nv -3 5 2\n"
        "o box\nusemtl half\nf 1 2 3 4\nf 1 2 6 5\nusemtl mirror\nf 2 3 7 6\nusemtl glass\nf 1 4 8 5\n"
        "g top\nusemtl lamp\nf 4 3 7 8\n")
    sc3 = rt.Scene()
    sc3.push_obj(str(tmp_path / "m.obj"), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0))
    assert sc3.triangles().shape[0] == 10
    tris3, nodes3 = rt.bvh_build(sc3.triangles())
    cam3 = rt.camera_from_angles((0.2, 2.0, 6.0), -92.0, 2.0, 35.0, W, H)
    want3, st3 = _oracle_for(tris3, nodes3, W, H, None).render(cam3, 1, ro, want_stats=True)
    res3 = _render_all_pipelines(tris3, nodes3, cam3, ro, W, H)
    for name, (img, rays) in res3.items():
        assert _same_image(img, want3), name
        assert rays == st3["rays"], name
    assert np.nan_to_num(want3).max() > 1.0


def _globe_obj(path, nu=32, nv=16, missing_library=True):
    """A latitude/longitude unit sphere as an OBJ with v/vt/vn corners: triangles at the north
    pole, quads elsewhere; vn = the position (smooth normals)."""
    lines = ["mtllib not-there.mtl" if missing_library else "", "o globe"]
    for j in range(nv + 1):
        for i in range(nu):
            th, ph = np.pi * j / nv, 2.0 * np.pi * i / nu
            x, y, z = np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)
            lines += ["v %.6f %.6f %.6f" % (x, y, z), "vn %.4f %.4f %.4f" % (x, y, z), "vt %.4f %.4f" % (i / nu, j / nv)]
    lines += ["usemtl b0b0b0", "s 1"]
    idx = lambda i, j: j * nu + (i % nu) + 1
    ntri = 0
    for j in range(nv):
        for i in range(nu):
            a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
            if j == 0:
                lines.append("f %d/%d/%d %d/%d/%d %d/%d/%d" % (a, a, a, c, c, c, d, d, d)); ntri += 1
            else:
                lines.append("f %d/%d/%d %d/%d/%d %d/%d/%d %d/%d/%d" % (a, a, a, b, b, b, c, c, c, d, d, d)); ntri += 2
    open(path, "w").write("\n".join(lines) + "\n")
    return ntri


def test_render_model_shaped_like_the_references_globe(golden_dir, tmp_path):
    """The reference's larger models (globe-sphere.obj, low-poly-fox) are `v/vt/vn` corners, quad and
    triangle faces, one object, and - the globe - a material library that is not there.  A
    latitude/longitude sphere written the same way, placed in the reference's room: loader ->
    builder -> every pipeline against the oracle."""
    import os
    ntri = _globe_obj(str(tmp_path / "globe.obj"))
    W, H = 150, 100
    sc = rt.Scene()
    sc.push_obj(os.path.join(golden_dir, "objects", "room.obj"), (5.0, 5.0, 5.0), (0.0, 5.0, 0.0))
    sc.push_obj(str(tmp_path / "globe.obj"), (1.6, 1.6, 1.6), (-0.5, 2.0, 0.5))
    t = sc.triangles()
    assert t.shape[0] == 20 + ntri
    assert np.allclose(t["kd"][20:], 0.6) and not t["ks"][20:].any() and t["n"][20:].any() and t["tex"][20:].any()
    tris, nodes = rt.bvh_build(t)
    cam = rt.CameraRig(W, H).uniforms()
    ro = rt.rand_origins(20221003, 3)
    want, st = _oracle_for(tris, nodes, W, H, None).render(cam, 1, ro, want_stats=True)
    res = _render_all_pipelines(tris, nodes, cam, ro, W, H)
    for name, (img, rays) in res.items():
        assert _same_image(img, want), name
        assert rays == st["rays"], name
    assert np.nan_to_num(want).max() > 1.0


def test_shading_with_interpolated_vertex_normals(golden_dir, tmp_path):
    """RtOptions.shading_normals (the reference's Triangle carries n0 n1 n2, frag:131, but
    getTriangleNormal, frag:296-300, still returns the face normal): every pipeline against the
    oracle in the same mode; the image differs from the face-normal one on a smooth mirror ball;
    triangles without normals (the room's file has `vn 0 -1 0` on every face, the procedural
    scenes have zeros) keep their face normal; the option needs a scene uploaded with it set."""
    import os
    _globe_obj(str(tmp_path / "globe.obj"), missing_library=False)
    (tmp_path / "globe.obj").write_text("mtllib ball.mtl\n" + (tmp_path / "globe.obj").read_text())
    (tmp_path / "ball.mtl").write_text("newmtl b0b0b0\nKd 0.9 0.8 0.7\nKs 0.5 1.0 0\n")      # half diffuse, half mirror
    W, H = 150, 100
    sc = rt.Scene()
    sc.push_obj(os.path.join(golden_dir, "objects", "room.obj"), (5.0, 5.0, 5.0), (0.0, 5.0, 0.0))
    sc.push_obj(str(tmp_path / "globe.obj"), (1.8, 1.8, 1.8), (-0.5, 2.0, 0.5))
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.CameraRig(W, H).uniforms()
    ro = rt.rand_origins(20221003, 3)
    flat, _ = _oracle_for(tris, nodes, W, H, None).render(cam, 1, ro)
    want, st = _oracle_for(tris, nodes, W, H, None, shading_normals=True).render(cam, 1, ro, want_stats=True)
    assert not _same_image(flat, want)
    for name, pipeline, accel in (("paths", rt.PIPELINE_PERSISTENT, True), ("paths_ref_order", rt.PIPELINE_PERSISTENT, False),
                                  ("wavefront", rt.PIPELINE_WAVEFRONT, True), ("mega", rt.PIPELINE_MEGAKERNEL, False)):
        r = rt.Renderer(W, H)
        r.set_options(pipeline=pipeline, accel=accel, shading_normals=True)
        r.upload(tris, nodes)
        r.set_camera(cam)
        r.render(1, ro)
        assert _same_image(r.read_mean(), want), name
        assert r.stats()["rays"] == st["rays"], name
        # the same context with the option off again: the reference's normals
        r.set_options(pipeline=pipeline, accel=accel, shading_normals=False)
        r.reset()
        r.render(1, ro)
        assert _same_image(r.read_mean(), flat), name
        r.close()
    # zero normals: nothing to interpolate, the face normal stays
    z = tris.copy(); z["n"] = 0.0
    r = rt.Renderer(W, H)
    r.set_options(shading_normals=True)
    r.upload(z, nodes)
    r.set_camera(cam)
    r.render(1, ro)
    assert _same_image(r.read_mean(), flat)
    r.close()
    # a scene uploaded without the option has no normals on the GPU
    r = rt.Renderer(W, H)
    r.upload(tris, nodes)
    with pytest.raises(rt.RtError, match="shading_normals"):
        r.set_options(shading_normals=True)
    r.close()


def test_albedo_texture_at_interpolated_texture_coordinates(golden_dir, tmp_path):
    """RtOptions.albedo_texture (the reference's Triangle carries uv0 uv1 uv2, frag:132, and its
    loader reads diffuse maps, Model.h:167-169, but the shader never samples one): every pipeline
    against the oracle in the same mode, on a model with `vt` corners whose coordinates are
    stretched beyond [0,1] and below 0 (GL_REPEAT); together with shading_normals; off again on
    the same context; and the call-order errors."""
    import os
    _globe_obj(str(tmp_path / "globe.obj"))
    W, H = 150, 100
    sc = rt.Scene()
    sc.push_obj(os.path.join(golden_dir, "objects", "room.obj"), (5.0, 5.0, 5.0), (0.0, 5.0, 0.0))
    sc.push_obj(str(tmp_path / "globe.obj"), (1.8, 1.8, 1.8), (-0.5, 2.0, 0.5))
    t = sc.triangles()
    t["tex"] = t["tex"] * np.float32(3.7) - np.float32(1.3)
    t["tex"][:20] = t["v"][:20][:, :, [0, 2]] * np.float32(0.23) + t["v"][:20][:, :, [1, 1]] * np.float32(0.31)   # the room: from positions
    t["tex"][4:8, :, 0] = np.float32(-1e-9)     # just below a whole number: fractional part 1.0 in fp32, wraps to the first texel
    t["tex"][4:8, :, 1] = np.float32(-2.75)
    tris, nodes = rt.bvh_build(t)
    image = np.random.default_rng(11).integers(0, 256, (23, 37, 3)).astype(np.uint8)
    cam = rt.CameraRig(W, H).uniforms()
    ro = rt.rand_origins(20221003, 3)
    flat, _ = _oracle_for(tris, nodes, W, H, None).render(cam, 1, ro)
    want, st = _oracle_for(tris, nodes, W, H, None, albedo_texture=image).render(cam, 1, ro, want_stats=True)
    both, _ = _oracle_for(tris, nodes, W, H, None, albedo_texture=image, shading_normals=True).render(cam, 1, ro)
    assert not _same_image(flat, want) and not _same_image(both, want)
    for name, pipeline, accel in (("paths", rt.PIPELINE_PERSISTENT, True), ("paths_ref_order", rt.PIPELINE_PERSISTENT, False),
                                  ("wavefront", rt.PIPELINE_WAVEFRONT, True), ("mega", rt.PIPELINE_MEGAKERNEL, False)):
        r = rt.Renderer(W, H)
        r.set_options(pipeline=pipeline, accel=accel, albedo_texture=True, shading_normals=True)
        r.upload(tris, nodes)
        r.set_camera(cam)
        with pytest.raises(rt.RtError, match="rt_set_albedo_texture"):
            r.render(1, ro)
        r.set_albedo_texture(image)
        r.render(1, ro)
        assert _same_image(r.read_mean(), both), name
        r.set_options(pipeline=pipeline, accel=accel, albedo_texture=True)
        r.reset()
        r.render(1, ro)
        assert _same_image(r.read_mean(), want), name
        assert r.stats()["rays"] == st["rays"], name
        # another image on the same context, then the option off again: the reference's albedo
        r.set_albedo_texture(np.full((2, 2, 3), 255, np.uint8))
        r.reset()
        r.render(1, ro)
        assert _same_image(r.read_mean(), flat), name
        r.set_albedo_texture(image)
        r.set_options(pipeline=pipeline, accel=accel)
        r.reset()
        r.render(1, ro)
        assert _same_image(r.read_mean(), flat), name
        r.close()
    # a scene uploaded without the option has no texture coordinates on the GPU
    r = rt.Renderer(W, H)
    r.upload(tris, nodes)
    with pytest.raises(rt.RtError, match="albedo_texture"):
        r.set_options(albedo_texture=True)
    with pytest.raises(rt.RtError, match="width/height"):
        r.set_albedo_texture(np.zeros((0, 4, 3), np.uint8))
    r.close()


def test_sweep_built_tree_renders_like_the_oracle_on_the_same_tree():
    """The build mode the reference's author left commented out (BVH.h:225-256) gives another
    node order; the shader semantics on that tree are the same, so every pipeline must match the
    oracle fed with the same tree.  Deep trees of that mode are refused at upload."""
    W, H = 120, 80
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles(), sweep=True)
    m_tris, m_nodes = rt.bvh_build(sc.triangles())
    assert nodes.tobytes() != m_nodes.tobytes()
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(20221003, 3)
    want, st = _oracle_for(tris, nodes, W, H, sc.spheres()).render(cam, 1, ro, want_stats=True)
    res = _render_all_pipelines(tris, nodes, cam, ro, W, H, spheres=sc.spheres())
    for name, (img, rays) in res.items():
        assert _same_image(img, want), name
        assert rays == st["rays"], name
    # closest hits do not depend on the tree, so the image equals the median tree's wherever no
    # two triangles tie (the demo scene has none in view)
    want_m, _ = _oracle_for(m_tris, m_nodes, W, H, sc.spheres()).render(cam, 1, ro)
    assert np.array_equal(want, want_m)
    # 300 random triangles: the sweep peels them off one by one, the tree is ~300 levels deep
    t2, n2 = rt.bvh_build(scenes.random_triangles(300, 3), sweep=True)
    r = rt.Renderer(16, 16)
    with pytest.raises(rt.RtError, match="deeper than 60"):
        r.upload(t2, n2)
    r.close()


def test_overlapped_calls_equal_serial_calls():
    """RtOptions.overlap: consecutive calls alternate between two internal streams.  One frame per
    call (the reference's call pattern, main.cpp:124-172), resets, restores and reads in between:
    bit-equal to everything on the context's stream, and to the oracle."""
    W, H = 200, 120
    sc = rt.Scene()
    sc.push_synthetic(1, 6, 1, 12345)
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(31, 40)
    want, _ = _oracle_for(tris, nodes, W, H, sc.spheres()).render(cam, 1, ro[:12])
    out = {}
    for overlap in (True, False):
        r = rt.Renderer(W, H)
        r.upload(tris, nodes)
        r.set_spheres(sc.spheres())
        r.set_camera(cam)
        r.set_options(overlap=overlap)
        for k in range(40):
            r.render(k + 1, ro[k:k + 1])
        a = (r.read_mean(), r.read_sum(), r.stats()["rays"])
        r.reset()                                           # a reset between overlapped calls
        for k in range(0, 12, 3):
            r.render(k + 1, ro[k:k + 3])
        b = (r.read_mean(), r.read_sum(), r.stats()["rays"], r.stats()["frames"])
        r.restore(a[0], a[1], 40)                           # and a restore
        r.render(41, ro[:5])
        r.render(46, ro[5:6])
        c = (r.read_mean(), r.read_sum())
        out[overlap] = (a, b, c)
        r.close()
    for x, y in zip(out[True], out[False]):
        for u, v in zip(x, y):
            assert np.array_equal(u, v, equal_nan=True) if isinstance(u, np.ndarray) else u == v
    assert _same_image(out[True][1][0], want) and out[True][1][3] == 12


def test_error_paths():
    r = rt.Renderer(16, 16)
    ro = rt.rand_origins(1, 1)
    with pytest.raises(rt.RtError, match="no scene"):
        r.render(1, ro)
    tris, nodes = rt.bvh_build(scenes.demo().triangles())
    bad = nodes.copy()
    bad[3]["offsetOrRc"] = 1       # right child pointing backwards
    with pytest.raises(rt.RtError, match="malformed"):
        r.upload(tris, bad)
    with pytest.raises(rt.RtError, match="2n-1"):
        r.upload(tris, nodes[:-1])
    # more triangles than the 32-bit record offsets of the kernels can address: refused before anything is read
    too_many = rt.RT_MAX_TRIANGLES + 1
    rc = rt.lib.rt_scene_upload(r._h, tris.ctypes.data, too_many, nodes.ctypes.data, 2 * too_many - 1)
    assert rc == -1 and b"RT_MAX_TRIANGLES" in rt.lib.rt_last_error(r._h)
    r.upload(tris, nodes)
    with pytest.raises(rt.RtError, match="no camera"):
        r.render(1, ro)
    r.set_camera(rt.camera_default(16, 16))
    with pytest.raises(rt.RtError, match="first_loopnum"):
        r.render(0, ro)
    with pytest.raises(rt.RtError):
        r.set_options(max_bounces=9)
    with pytest.raises(rt.RtError, match="0..10"):
        r.set_spheres(np.zeros(11, dtype=rt.SPHERE_DTYPE))
    r.render(1, ro)
    assert r.stats()["frames"] == 1
    r.reset()
    assert r.stats()["frames"] == 0 and (r.read_sum() == 0).all()


def test_cxx_host_program_matches_python_path(tmp_path):
    """rt_render (the C++ mirror of the reference's main loop) writes the same image as the
    ctypes path for the demo scene."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(rt.LIB_PATH), "rt_render")
    out = tmp_path / "img.pfm"
    W, H, spp = 120, 80, 5
    subprocess.check_call([exe, "--demo", "--size", str(W), str(H), "--spp", str(spp), "--out", str(out), "--ppm", str(tmp_path / "img.ppm")])
    raw = open(out, "rb").read()
    header = b"PF\n%d %d\n-1.0\n" % (W, H)
    assert raw.startswith(header)
    got = np.frombuffer(raw[len(header):], dtype="<f4").reshape(H, W, 3)
    sc = scenes.demo()
    r = rt.make_renderer(sc, W, H)
    r.render(1, rt.rand_origins(20221003, spp))
    assert np.array_equal(got, r.read_mean())
    ppm = open(tmp_path / "img.ppm", "rb").read()
    assert ppm.startswith(b"P6\n%d %d\n255\n" % (W, H)) and len(ppm) == len(b"P6\n%d %d\n255\n" % (W, H)) + W * H * 3
    # the reference's own loop: one frame per call, the image fetched after every call (--show: the
    # copy of one call's image overlaps the tracing of the next); same final image
    shown = subprocess.check_output([exe, "--demo", "--size", str(W), str(H), "--spp", str(spp), "--frames-per-call", "1", "--show",
                                     "--out", str(tmp_path / "shown.pfm")]).decode()
    assert "shown: %d images" % spp in shown
    raw2 = open(tmp_path / "shown.pfm", "rb").read()
    assert np.array_equal(np.frombuffer(raw2[len(header):], dtype="<f4").reshape(H, W, 3), got)
    # --texture: a binary PPM sampled at the model's vt coordinates (RtOptions.albedo_texture)
    _globe_obj(str(tmp_path / "globe.obj"))
    image = np.random.default_rng(3).integers(0, 256, (9, 14, 3)).astype(np.uint8)
    open(tmp_path / "tex.ppm", "wb").write(b"P6\n# made by the test\n14 9\n255\n" + image.tobytes())
    subprocess.check_call([exe, "--obj", str(tmp_path / "globe.obj"), "2", "2", "2", "0", "2", "0", "--texture", str(tmp_path / "tex.ppm"),
                           "--size", str(W), str(H), "--spp", str(spp), "--out", str(out)])
    got = np.frombuffer(open(out, "rb").read()[len(header):], dtype="<f4").reshape(H, W, 3)
    sc = rt.Scene()
    sc.push_obj(str(tmp_path / "globe.obj"), (2.0, 2.0, 2.0), (0.0, 2.0, 0.0))
    tris, nodes = rt.bvh_build(sc.triangles())
    r = rt.Renderer(W, H)
    r.set_options(albedo_texture=True)
    r.upload(tris, nodes)
    r.set_albedo_texture(image)
    r.set_camera(rt.camera_default(W, H))
    r.render(1, rt.rand_origins(20221003, spp))
    assert np.array_equal(got, r.read_mean())

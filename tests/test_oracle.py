"""The shader oracle itself: committed golden vectors, arithmetic-contract sensitivity, the
fixed sin/cos sequence, and quirks of the reference that the restatement must keep."""
import ctypes as C
import os

import numpy as np

import oracle_lib as ol
import raytracing_amd as rt
import scenes


def _golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "demo_96x64.npz"))
    tris = g["tris"].view(ol.TRIANGLE_DTYPE)
    nodes = g["nodes"].view(ol.NODE_DTYPE)
    return g, tris, nodes


def test_golden_scene_is_the_demo_scene(golden_dir):
    g, tris, nodes = _golden(golden_dir)
    t2, n2 = rt.bvh_build(scenes.demo().triangles())
    assert tris.tobytes() == t2.tobytes() and nodes.tobytes() == n2.tobytes()


def test_oracle_reproduces_golden_images(golden_dir):
    g, tris, nodes = _golden(golden_dir)
    sc = scenes.demo()
    cam = rt.camera_default(96, 64)
    osc = ol.scene_from_build(tris, nodes, 96, 64, rt.sobol_default_table(), sc.spheres())
    assert np.array_equal(g["rand_origin"], rt.rand_origins(20221003, 4))
    img, cur, _ = osc.render_frame(cam, 1, float(g["rand_origin"][0]))
    assert np.array_equal(cur, g["frame1"]) and np.array_equal(img, g["frame1"])
    mean, _ = osc.render(cam, 1, g["rand_origin"])
    assert np.array_equal(mean, g["mean4"])
    t, ids = osc.hit_bvh(g["ray_o"], g["ray_d"])
    assert np.array_equal(t, g["ray_t"]) and np.array_equal(ids, g["ray_tri"])


def test_contract_sensitivity_is_inside_the_north_star_tolerance():
    """Unfused products and libm sin/cos are equally valid readings of the GLSL; at 1 spp they
    must agree with the parity target on >= 99.9 % of pixels within 1e-3."""
    W, H = 160, 100
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    V = rt.sobol_default_table()
    ro = rt.rand_origins(7, 3)
    base = ol.scene_from_build(tris, nodes, W, H, V, sc.spheres())
    plain = ol.scene_from_build(tris, nodes, W, H, V, sc.spheres(), fused=False)
    libm = ol.scene_from_build(tris, nodes, W, H, V, sc.spheres(), trig_mode=0)
    assert base.lib.or_is_fused() == 1 and plain.lib.or_is_fused() == 0
    for k in range(3):
        a, _, _ = base.render_frame(cam, k + 1, float(ro[k]))
        for other in (plain, libm):
            b, _, _ = other.render_frame(cam, k + 1, float(ro[k]))
            ok = (np.abs(a - b).max(axis=2) <= 1e-3).mean()
            assert ok >= 0.999, ok


def test_fixed_sincos_is_close_to_libm():
    lib = ol.load()
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for x in np.linspace(0.0, 6.2831855, 20001, dtype=np.float32):
        lib.or_sincos(C.c_float(float(x)), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - np.sin(np.float64(x))), abs(c.value - np.cos(np.float64(x))))
    assert worst < 2.5e-7       # about 2 ulp at 1.0


def test_first_frame_equals_sample_and_mean_recurrence():
    W, H = 48, 32
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    osc = ol.scene_from_build(tris, nodes, W, H, rt.sobol_default_table(), sc.spheres())
    ro = rt.rand_origins(5, 3)
    garbage = np.full((H, W, 3), 123.0, dtype=np.float32)
    out, cur, _ = osc.render_frame(cam, 1, float(ro[0]), garbage)       # history weight is 0 at N = 1 (frag:697)
    assert np.array_equal(out, cur)
    out2, cur2, _ = osc.render_frame(cam, 2, float(ro[1]), out)
    want = np.float32(0.5) * cur2 + np.float32(0.5) * out
    assert np.allclose(out2, want, rtol=0, atol=1e-6)
    # light seen directly is 4*pi*Ka = 12.566369 (frag:673 uses 3.1415926)
    assert out.max() == np.float32(4.0) * np.float32(3.1415926)


def test_quirks_kept():
    """A ray that misses stays black (frag:655-672); no spheres, no geometry hit -> zero image."""
    t = scenes.random_triangles(5, 3)
    t["v"] += 1000.0        # far away, behind the camera's view
    tris, nodes = rt.bvh_build(t)
    cam = rt.camera_default(16, 16)
    osc = ol.scene_from_build(tris, nodes, 16, 16, rt.sobol_default_table(), None)
    out, _, st = osc.render_frame(cam, 1, 1e6, want_stats=True)
    assert (out == 0).all()
    # the reference re-traces a missed ray up to 3 more times (Russian roulette may stop it at i = 2, 3)
    assert st["rays"] == 256 and 2 * 256 <= st["raysRef"] <= 4 * 256 and st["raysRef"] > 3 * 256 * 0.8


def test_shading_normal_mode_of_the_oracle():
    """OrScene.shadingNormals: zero vertex normals leave every pixel as in the reference's mode;
    vertex normals that tilt away from the face normal change diffuse pixels; a vertex normal
    pointing to the back of the face is turned to the front."""
    import raytracing_amd as rt
    import scenes
    W, H = 64, 40
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(20221003, 2)
    V = rt.sobol_default_table()
    z = tris.copy(); z["n"] = 0.0
    flat, _ = ol.scene_from_build(z, nodes, W, H, V, sc.spheres()).render(cam, 1, ro)
    same, _ = ol.scene_from_build(z, nodes, W, H, V, sc.spheres(), shading_normals=True).render(cam, 1, ro)
    assert np.array_equal(np.nan_to_num(flat), np.nan_to_num(same))
    t = tris.copy()
    t["n"][:, :, :] = np.array([0.3, 1.0, 0.2], dtype=np.float32)
    tilted, _ = ol.scene_from_build(t, nodes, W, H, V, sc.spheres(), shading_normals=True).render(cam, 1, ro)
    assert not np.array_equal(np.nan_to_num(flat), np.nan_to_num(tilted))
    t["n"] *= -1.0
    back, _ = ol.scene_from_build(t, nodes, W, H, V, sc.spheres(), shading_normals=True).render(cam, 1, ro)
    assert np.array_equal(np.nan_to_num(back), np.nan_to_num(tilted))


def test_albedo_texture_mode_of_the_oracle():
    """OrScene.albedoTex: a white image leaves every pixel as in the reference's mode; a one-texel
    image is the same as scaling every kd by that colour (lights keep ka); texture coordinates
    wrap as GL_REPEAT: a triangle whose three corners carry the same coordinate reads the texel
    of its fractional part whatever whole number is added."""
    import raytracing_amd as rt
    import scenes
    W, H = 64, 40
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    ro = rt.rand_origins(20221003, 2)
    V = rt.sobol_default_table()
    eq = lambda a, b: np.array_equal(np.nan_to_num(a), np.nan_to_num(b))
    render = lambda t, **kw: ol.scene_from_build(t, nodes, W, H, V, sc.spheres(), **kw).render(cam, 1, ro)[0]
    flat = render(tris)
    assert eq(render(tris, albedo_texture=np.full((3, 5, 3), 255, np.uint8)), flat)
    colour = np.array([51, 102, 255], dtype=np.uint8)
    scaled = tris.copy()
    scaled["kd"] = tris["kd"] * (colour.astype(np.float32) / np.float32(255.0))
    one = render(tris, albedo_texture=colour.reshape(1, 1, 3))
    assert eq(one, render(scaled)) and not eq(one, flat)
    # two texels side by side; every triangle sits in the middle of one of them, some whole numbers away
    rng = np.random.default_rng(5)
    two = np.array([[[200, 30, 90], [10, 250, 120]]], dtype=np.uint8)
    pick = rng.integers(0, 2, tris.shape[0])
    t = tris.copy()
    t["tex"][:, :, 0] = (0.25 + 0.5 * pick + rng.integers(-3, 4, tris.shape[0]))[:, None]
    t["tex"][:, :, 1] = rng.integers(-2, 3, tris.shape[0])[:, None] + 0.5
    scaled = tris.copy()
    scaled["kd"] = tris["kd"] * (two[0, pick].astype(np.float32) / np.float32(255.0))
    assert eq(render(t, albedo_texture=two), render(scaled))
    # a coordinate just below a whole number: its fractional part rounds to 1 in fp32, and GL_REPEAT
    # wraps that to the FIRST texel, not the last
    t["tex"][:, :, 0] = np.float32(-1e-9)
    t["tex"][:, :, 1] = np.float32(-2.75)
    assert np.float32(-1e-9) - np.floor(np.float32(-1e-9)) == np.float32(1.0)
    four = np.array([[[200, 30, 90], [10, 250, 120]], [[90, 90, 10], [20, 40, 250]]], dtype=np.uint8)
    scaled["kd"] = tris["kd"] * (four[0, 0].astype(np.float32) / np.float32(255.0))
    assert eq(render(t, albedo_texture=four), render(scaled))

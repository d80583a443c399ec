"""Host-side scene assembly: the OBJ/MTL loader on the reference's own asset files against the
procedural demo scene, the transform order of Scene::pushModel, and the synthetic generators."""
import os

import numpy as np
import pytest

import raytracing_amd as rt


def _demo_from_files(golden_dir):
    sc = rt.Scene()
    sc.push_obj(os.path.join(golden_dir, "objects", "room.obj"), (5.0, 5.0, 5.0), (0.0, 5.0, 0.0))   # main.cpp:106
    sc.push_obj(os.path.join(golden_dir, "objects", "cube.obj"), (0.5, 5.0, -1.0), (0.0, 1.0, -2.0))  # main.cpp:107
    return sc


def test_loader_reproduces_demo_scene(golden_dir):
    a = _demo_from_files(golden_dir).triangles()
    sc = rt.Scene()
    sc.push_demo()
    b = sc.triangles()
    assert a.shape[0] == b.shape[0] == 32       # the `triangle` object's face is commented out in room.obj
    assert a.tobytes() == b.tobytes()
    assert sc.spheres().shape[0] == 2


def test_demo_scene_geometry_and_materials(golden_dir):
    t = _demo_from_files(golden_dir).triangles()
    room, cube = t[:20], t[20:]
    assert room["v"].reshape(-1, 3).min(axis=0).tolist() == [-10.0, 0.0, -5.0]
    assert room["v"].reshape(-1, 3).max(axis=0).tolist() == [5.0, 10.0, 5.0]
    assert cube["v"].reshape(-1, 3).min(axis=0).tolist() == [-0.5, -4.0, -3.0]
    assert cube["v"].reshape(-1, 3).max(axis=0).tolist() == [0.5, 6.0, -1.0]
    # room.mtl: only the light has Ka; every surface is always diffuse (Ks = 1 1 x)
    assert (t["ka"].sum(axis=1) > 0).sum() == 2 and (t[:2]["ka"] == 1.0).all()
    assert (t["ks"][:, 0] == 1.0).all() and (t["ks"][:, 1] == 1.0).all()
    assert np.allclose(t[2]["kd"], [0.824222, 0.092157, 0.158824])      # left wall
    assert np.allclose(cube[0]["kd"], [0.824222, 0.692157, 0.158824])
    # light quad at y = 10 spanning +-2.5
    assert (t[:2]["v"][:, :, 1] == 10.0).all() and np.abs(t[:2]["v"][:, :, 0]).max() == 2.5
    # aiProcess_FlipUVs: "vt 0 1" becomes (0, 0)
    assert t[0]["tex"][0].tolist() == [0.0, 0.0] and t[0]["tex"][1].tolist() == [0.0, 1.0]
    assert (t["n"].reshape(-1, 3) == [0.0, -1.0, 0.0]).all()


def test_mtllib_case_fallback(tmp_path, golden_dir):
    # room.obj names "Room.mtl"; the file on disk is room.mtl
    assert "mtllib Room.mtl" in open(os.path.join(golden_dir, "objects", "room.obj"), errors="ignore").read()
    sc = rt.Scene()
    sc.push_obj(os.path.join(golden_dir, "objects", "room.obj"))
    assert (sc.triangles()["ka"].sum(axis=1) > 0).sum() == 2


def test_push_obj_errors():
    sc = rt.Scene()
    with pytest.raises(rt.RtError, match="cannot open"):
        sc.push_obj("/nonexistent/file.obj")


def test_push_model_transform_is_scale_then_translate():
    t = np.zeros(1, dtype=rt.TRIANGLE_DTYPE)
    t["v"][0] = [[0.1, 0.2, 0.3], [1, 2, 3], [-1, 0.7, 0.9]]
    s = np.array([3.0, 0.7, -1.3], dtype=np.float32)
    tr = np.array([0.1, 5.0, -2.0], dtype=np.float32)
    sc = rt.Scene()
    sc.push_triangles(t, s, tr)
    got = sc.triangles()["v"][0]
    want = (t["v"][0] * s).astype(np.float32) + tr      # two roundings, Scene.h:58-59
    assert np.array_equal(got, want)


def test_polygon_fan_and_negative_indices(tmp_path):
    p = tmp_path / "fan.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0.5 1.5 0\nv 0 1 0\nf 1 2 3 4 5\nf -5 -4 -3\n")
    sc = rt.Scene()
    sc.push_obj(str(p))
    t = sc.triangles()
    assert t.shape[0] == 4
    assert t[1]["v"].tolist() == [[0, 0, 0], [1, 1, 0], [0.5, 1.5, 0]]
    assert t[3]["v"].tolist() == [[0, 0, 0], [1, 0, 0], [1, 1, 0]]
    assert np.allclose(t[0]["kd"], 0.6)     # no material: loader default


def test_material_and_group_splitting(tmp_path):
    """One mesh per object/group statement and per change of material inside it, in file order,
    each carrying its own Ka/Kd/Ks (Model.h:158-165 reads them per mesh); an empty group and a
    repeated `usemtl` of the same name start nothing; a material that is not in the library gives
    the loader default."""
    (tmp_path / "two.mtl").write_text(
        "newmtl red\nKa 0 0 0\nKd 0.9 0.1 0.1\nKs 1 1 0\n\n"
        "newmtl glow\nKa 2 2 2\nKd 0 0 0\nKs 0 0 0\n\n"
        "newmtl glass\nKd 1 1 1\nKs 0 0 0.5\n")
    (tmp_path / "two.obj").write_text(
        "mtllib two.mtl\n"
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nv 1 0 1\nv 1 1 1\nv 0 1 1\n"
        "o body\n"
        "usemtl red\nf 1 2 3\nusemtl red\nf 1 3 4\n"        # same material again: still one mesh
        "usemtl glow\nf 5 6 7\n"                              # second material inside the object
        "g empty\n"                                            # a group without faces leaves no mesh
        "g lid\nusemtl glass\nf 5 7 8\nf 1 2 6 5\n"         # group split; a quad -> 2 triangles
        "usemtl nosuch\nf 2 3 7\n")
    sc = rt.Scene()
    sc.push_obj(str(tmp_path / "two.obj"), (2.0, 2.0, 2.0), (0.0, 0.0, -1.0))
    t = sc.triangles()
    assert t.shape[0] == 7
    kd = t["kd"].astype(np.float64).round(4).tolist()
    assert kd[0] == kd[1] == [0.9, 0.1, 0.1] and t["ks"][0].tolist() == [1.0, 1.0, 0.0]
    assert t["ka"][2].tolist() == [2.0, 2.0, 2.0] and kd[2] == [0.0, 0.0, 0.0]
    assert kd[3] == kd[4] == kd[5] == [1.0, 1.0, 1.0] and t["ks"][3].tolist() == [0.0, 0.0, 0.5]
    assert t["ka"][3].tolist() == [0.0, 0.0, 0.0]                   # Ka absent in the library: zero
    assert np.allclose(t["kd"][6], 0.6) and t["ks"][6].tolist() == [0.0, 0.0, 0.0]
    # file order of faces is kept across the splits; scale then translate
    assert t["v"][2].tolist() == [[0, 0, 1], [2, 0, 1], [2, 2, 1]]
    assert t["v"][4].tolist() == [[0, 0, -1], [2, 0, -1], [2, 0, 1]] and t["v"][5].tolist() == [[0, 0, -1], [2, 0, 1], [0, 0, 1]]
    # the same triangles go through the builder like any others
    tris, nodes = rt.bvh_build(t)
    assert nodes.shape[0] == 13 and sorted(tris["kd"].astype(np.float64).round(4).tolist()) == sorted(kd)


def test_synthetic_scenes():
    sc = rt.Scene()
    sc.push_synthetic(0, 2, 1, 12345)
    t = sc.triangles()
    assert t.shape[0] == 20 + 20 * 16
    ico = t[20:]["v"].reshape(-1, 3)
    r = np.linalg.norm(ico - np.array([-2.5, 4.0, -1.0], dtype=np.float32), axis=1)
    assert 2.5 < r.min() and r.max() < 3.5
    # watertight: every vertex position is shared by several triangles
    uniq, counts = np.unique(ico.round(5), axis=0, return_counts=True)
    assert counts.min() >= 5
    sc2 = rt.Scene()
    sc2.push_synthetic(0, 2, 8, 12345)
    assert sc2.triangles().shape[0] == 20 + 8 * 320
    sc3 = rt.Scene()
    sc3.push_synthetic(1, 8, 1, 1)
    t3 = sc3.triangles()
    assert t3.shape[0] == 20 + 32 * 2 * 8 * 8 and sc3.spheres().shape[0] == 2
    kinds = {tuple(k) for k in t3[20:]["ks"][:, :2].tolist()}
    assert kinds == {(1.0, 1.0), (0.0, 1.0), (0.0, 0.0)}
    with pytest.raises(rt.RtError):
        sc3.push_synthetic(7, 1, 1, 1)


def test_camera_default_and_from_angles():
    cam = rt.camera_default(1920, 1080)[0]
    assert cam["front"].tolist() == [0, 0, -1] and cam["right"].tolist() == [1, 0, 0] and cam["up"].tolist() == [0, 1, 0]
    assert cam["halfH"] == 1.0 and cam["halfW"] == np.float32(1920 / 1080)
    assert np.array_equal(cam["leftbottom"], np.array([-cam["halfW"], -1.0, -1.0], dtype=np.float32))
    c2 = rt.camera_from_angles((0, 2, 5), -90.0, 0.0, 45.0, 1200, 800)[0]     # Camera.h:18-46
    assert np.allclose(c2["front"], [0, 0, -1], atol=1e-6) and abs(c2["halfH"] - 1.0) < 1e-6
    assert np.allclose(c2["halfW"], 1.5, atol=1e-6)
    c3 = rt.camera_from_angles((1, 2, 3), 30.0, 20.0, 30.0, 640, 480)[0]
    for v in ("front", "right", "up"):
        assert abs(np.linalg.norm(c3[v]) - 1) < 1e-6
    assert abs(np.dot(c3["front"], c3["right"])) < 1e-6 and abs(np.dot(c3["front"], c3["up"])) < 1e-6
    assert np.allclose(c3["leftbottom"], c3["front"] - c3["halfW"] * c3["right"] - c3["halfH"] * c3["up"], atol=1e-6)


# ---- the reference's own model files (read where the reference lies; absent on a GPU box)

REF_OBJECTS = "/root/reference/光线追踪/光线追踪/objects"


def _parse_obj_independently(path):
    """A second OBJ reader, written for this test only: positions of the fan-triangulated faces in
    file order plus per-corner normals and (v-flipped) texture coordinates."""
    V, VT, VN, P, N, T = [], [], [], [], [], []
    for line in open(path, encoding="latin-1"):
        w = line.split()
        if not w:
            continue
        if w[0] == "v":
            V.append([np.float32(x) for x in w[1:4]])
        elif w[0] == "vt":
            VT.append([np.float32(x) for x in w[1:3]])
        elif w[0] == "vn":
            VN.append([np.float32(x) for x in w[1:4]])
        elif w[0] == "f":
            cs = []
            for tok in w[1:]:
                f = (tok.split("/") + ["", ""])[:3]
                vi = int(f[0]); ti = int(f[1]) if f[1] else 0; ni = int(f[2]) if f[2] else 0
                vi = vi if vi > 0 else len(V) + 1 + vi
                ti = ti if ti >= 0 else len(VT) + 1 + ti
                ni = ni if ni >= 0 else len(VN) + 1 + ni
                cs.append((vi, ti, ni))
            for i in range(1, len(cs) - 1):
                for c in (cs[0], cs[i], cs[i + 1]):
                    P.append(V[c[0] - 1])
                    N.append(VN[c[2] - 1] if c[2] else [0, 0, 0])
                    T.append([VT[c[1] - 1][0], np.float32(1.0) - VT[c[1] - 1][1]] if c[1] else [0, 0])
    shape = (len(P) // 3, 3)
    return (np.array(P, np.float32).reshape(shape + (3,)), np.array(N, np.float32).reshape(shape + (3,)),
            np.array(T, np.float32).reshape(shape + (2,)))


@pytest.mark.skipif(not os.path.isdir(REF_OBJECTS), reason="the reference's model files are not on this machine")
@pytest.mark.parametrize("name,ntri,kd,ks", [
    ("low-poly-fox-by-pixelmannen.obj", 576, [1.0, 1.0, 1.0], [0.089804] * 3),     # triangles, v/vt/vn corners
    ("globe-sphere.obj", 32040, [0.6, 0.6, 0.6], [0.0] * 3),     # 16020 quads; its mtllib names a file that does not exist
    ("first-cube.obj", 54, None, None),                                            # two objects, two materials, quads and triangles
])
def test_reference_model_files(name, ntri, kd, ks):
    """Model.h:54-182 on the models the reference ships besides room/cube: every face arrives,
    fan-triangulated and in file order, with its corner normals and flipped texture
    coordinates (Model.h:101-133) and the Ka/Kd/Ks of its material (Model.h:158-165); a missing
    material library gives the importer's default material."""
    path = os.path.join(REF_OBJECTS, name)
    sc = rt.Scene()
    sc.push_obj(path, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0))
    t = sc.triangles()
    P, N, T = _parse_obj_independently(path)
    assert t.shape[0] == ntri == P.shape[0]
    assert np.array_equal(t["v"], P) and np.array_equal(t["n"], N) and np.array_equal(t["tex"], T)
    if kd is not None:
        assert np.allclose(t["kd"], kd, atol=1e-6) and np.allclose(t["ks"], ks, atol=1e-6) and not t["ka"].any()
    else:
        assert len(np.unique(t["kd"], axis=0)) == 2      # one material per object
    # the builder takes it like any other triangle list
    tris, nodes = rt.bvh_build(t)
    assert nodes.shape[0] == 2 * ntri - 1 and tris.shape[0] == ntri

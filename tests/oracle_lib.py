"""ctypes access to the CPU oracle (oracle/liboracle*.so) for tests, smoke() and the
cpu_baseline leg of bench.py.  Never imported by the raytracing_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

TRIANGLE_DTYPE = np.dtype([("v", "<f4", (3, 3)), ("n", "<f4", (3, 3)), ("tex", "<f4", (3, 2)),
                           ("ka", "<f4", 3), ("kd", "<f4", 3), ("ks", "<f4", 3), ("center", "<f4", 3)])
NODE_DTYPE = np.dtype([("minBound", "<f4", 3), ("maxBound", "<f4", 3),
                       ("triangleNum", "<i4"), ("offsetOrRc", "<i4"), ("axis", "<i4")])


# Node indices below this survive the reference's float encoding: stored as floats
# (ObjectTexture.h:152-154) and recovered with int(f + 0.5) (frag:180-182); from 2^23 on, f + 0.5 is
# a tie that rounds to even, so odd indices come back one too large.
FLOAT_EXACT_INTS = 1 << 23


class OrCamera(C.Structure):
    _fields_ = [("camPos", C.c_float * 3), ("front", C.c_float * 3), ("right", C.c_float * 3),
                ("up", C.c_float * 3), ("halfH", C.c_float), ("halfW", C.c_float),
                ("leftbottom", C.c_float * 3), ("LoopNum", C.c_int)]


class OrSphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("albedo", C.c_float * 3),
                ("specularRate", C.c_float), ("refractionRate", C.c_float), ("roughness", C.c_float)]


class OrScene(C.Structure):
    _fields_ = [("texTriangle", C.c_void_p), ("texTriangleWidth", C.c_int),
                ("texBVH", C.c_void_p), ("texBVHWidth", C.c_int),
                ("triangleNum", C.c_int), ("bvhNum", C.c_int),
                ("width", C.c_int), ("height", C.c_int),
                ("V", C.c_void_p),
                ("sphere", OrSphere * 10), ("sphereNum", C.c_int),
                ("maxBounces", C.c_int), ("trigMode", C.c_int), ("bvhInts", C.c_void_p),
                ("shadingNormals", C.c_int),
                ("albedoTex", C.c_void_p), ("albedoTexW", C.c_int), ("albedoTexH", C.c_int)]


class OrStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("raysRef", C.c_uint64), ("nodeTests", C.c_uint64),
                ("triTests", C.c_uint64), ("triHits", C.c_uint64)]


def _make():
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


_cache = {}


def load(fused=True):
    name = "liboracle.so" if fused else "liboracle_plain.so"
    if name in _cache:
        return _cache[name]
    path = os.path.join(ORACLE_DIR, name)
    srcs = [os.path.join(ORACLE_DIR, s) for s in ("rt_oracle.c", "bvh_oracle.cpp", "Makefile")]
    if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs if os.path.exists(s)):
        _make()
    lib = C.CDLL(path)
    vp, i = C.c_void_p, C.c_int
    lib.bvh_oracle_build.restype = i
    lib.bvh_oracle_build.argtypes = [vp, i, vp]
    lib.bvh_oracle_build_sah.restype = i
    lib.bvh_oracle_build_sah.argtypes = [vp, i, vp]
    lib.bvh_oracle_sort.restype = None
    lib.bvh_oracle_sort.argtypes = [vp, vp, i]
    lib.bvh_oracle_tex_dims.restype = None
    lib.bvh_oracle_tex_dims.argtypes = [i, i, C.POINTER(i), C.POINTER(i)]
    lib.bvh_oracle_pack_triangles.restype = None
    lib.bvh_oracle_pack_triangles.argtypes = [vp, i, vp, i, i]
    lib.bvh_oracle_pack_nodes.restype = None
    lib.bvh_oracle_pack_nodes.argtypes = [vp, i, vp, i, i]
    lib.bvh_oracle_nth_element.restype = None
    lib.bvh_oracle_nth_element.argtypes = [vp, vp, i, i]
    lib.bvh_oracle_push_model.restype = None
    lib.bvh_oracle_push_model.argtypes = [vp, i, vp, vp]
    lib.or_sobol.restype = C.c_float
    lib.or_sobol.argtypes = [vp, C.c_uint32, C.c_uint32]
    lib.or_gray_code.restype = C.c_uint32
    lib.or_gray_code.argtypes = [C.c_uint32]
    lib.or_sincos.restype = None
    lib.or_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.or_shade_pixel.restype = None
    lib.or_shade_pixel.argtypes = [C.POINTER(OrScene), C.POINTER(OrCamera), C.c_float, i, i, vp, vp]
    lib.or_render_frame.restype = None
    lib.or_render_frame.argtypes = [C.POINTER(OrScene), C.POINTER(OrCamera), C.c_float, vp, vp, vp, i, i, vp]
    lib.or_render_band.restype = None
    lib.or_render_band.argtypes = [C.POINTER(OrScene), C.POINTER(OrCamera), i, i, vp, vp, vp, i, i]
    lib.or_hit_bvh.restype = i
    lib.or_hit_bvh.argtypes = [C.POINTER(OrScene), vp, vp, C.POINTER(C.c_float), C.POINTER(i)]
    lib.or_hit_bvh_many.restype = None
    lib.or_hit_bvh_many.argtypes = [C.POINTER(OrScene), vp, vp, C.c_longlong, vp, vp]
    lib.or_is_fused.restype = i
    _cache[name] = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def bvh_build(tris, lib=None, sah=False):
    """Reference-order BVH of `tris` (TRIANGLE_DTYPE; not modified): (permuted tris, nodes).
    sah: the split rule the reference's author left commented out (BVH.h:225-256)."""
    lib = lib or load()
    t = np.ascontiguousarray(tris, dtype=TRIANGLE_DTYPE).copy()
    n = t.shape[0]
    nodes = np.zeros(max(2 * n - 1, 1), dtype=NODE_DTYPE)
    cnt = (lib.bvh_oracle_build_sah if sah else lib.bvh_oracle_build)(_ptr(t), n, _ptr(nodes))
    return t, nodes[:cnt]


def pack_textures(tris, nodes, lib=None):
    lib = lib or load()
    x, y = C.c_int(), C.c_int()
    lib.bvh_oracle_tex_dims(tris.shape[0], 33, C.byref(x), C.byref(y))
    tt = np.zeros(x.value * y.value * 33, dtype=np.float32)
    lib.bvh_oracle_pack_triangles(_ptr(tris), tris.shape[0], _ptr(tt), x.value, y.value)
    tw = x.value * 33
    lib.bvh_oracle_tex_dims(nodes.shape[0], 9, C.byref(x), C.byref(y))
    tb = np.zeros(x.value * y.value * 9, dtype=np.float32)
    lib.bvh_oracle_pack_nodes(_ptr(nodes), nodes.shape[0], _ptr(tb), x.value, y.value)
    return tt, tw, tb, x.value * 9


class OracleScene:
    """Holds the reference-format textures and uniforms for the shader oracle."""

    def __init__(self, tex_tri, tex_tri_w, ntri, tex_bvh, tex_bvh_w, nnode, width, height,
                 sobol_v, spheres=None, max_bounces=4, trig_mode=1, fused=True, bvh_ints=None,
                 shading_normals=False, albedo_texture=None):
        self.lib = load(fused)
        self._tt = np.ascontiguousarray(tex_tri, dtype=np.float32)
        self._tb = np.ascontiguousarray(tex_bvh, dtype=np.float32)
        self._v = np.ascontiguousarray(sobol_v, dtype=np.uint32)
        s = OrScene()
        s.texTriangle = self._tt.ctypes.data
        s.texTriangleWidth = tex_tri_w
        s.texBVH = self._tb.ctypes.data
        s.texBVHWidth = tex_bvh_w
        s.triangleNum, s.bvhNum = ntri, nnode
        s.width, s.height = width, height
        s.V = self._v.ctypes.data
        s.sphereNum = 0
        if spheres is not None:
            s.sphereNum = len(spheres)
            for k, sp in enumerate(spheres):
                o = s.sphere[k]
                for c in range(3):
                    o.center[c] = float(sp["center"][c])
                    o.albedo[c] = float(sp["albedo"][c])
                o.radius = float(sp["radius"])
                o.specularRate = float(sp["specularRate"])
                o.refractionRate = float(sp["refractionRate"])
                o.roughness = float(sp["roughness"])
        s.maxBounces = max_bounces
        s.trigMode = trig_mode
        self._ints = None if bvh_ints is None else np.ascontiguousarray(bvh_ints, dtype=np.int32)
        s.bvhInts = None if self._ints is None else self._ints.ctypes.data
        s.shadingNormals = int(shading_normals)
        # albedo_texture: (H, W, 3) uint8, row 0 at v = 0
        self._tex = None if albedo_texture is None else np.ascontiguousarray(albedo_texture, dtype=np.uint8)
        if self._tex is not None:
            assert self._tex.ndim == 3 and self._tex.shape[2] == 3
            s.albedoTex = self._tex.ctypes.data
            s.albedoTexH, s.albedoTexW = self._tex.shape[0], self._tex.shape[1]
        self.s = s
        self.width, self.height = width, height

    @staticmethod
    def camera(cam_np, loopnum):
        c = OrCamera()
        r = cam_np.reshape(-1)[0]
        for k in range(3):
            c.camPos[k] = float(r["camPos"][k])
            c.front[k] = float(r["front"][k])
            c.right[k] = float(r["right"][k])
            c.up[k] = float(r["up"][k])
            c.leftbottom[k] = float(r["leftbottom"][k])
        c.halfH = float(r["halfH"])
        c.halfW = float(r["halfW"])
        c.LoopNum = loopnum
        return c

    def render_frame(self, cam_np, loopnum, rand_origin, hist=None, rows=None, want_stats=False):
        """One frame: returns (blended image, pre-blend sample image, stats or None)."""
        cam = self.camera(cam_np, loopnum)
        out = np.zeros((self.height, self.width, 3), dtype=np.float32)
        cur = np.zeros((self.height, self.width, 3), dtype=np.float32)
        y0, y1 = rows if rows else (0, self.height)
        st = OrStats()
        h = None if hist is None else np.ascontiguousarray(hist, dtype=np.float32)
        self.lib.or_render_frame(C.byref(self.s), C.byref(cam), C.c_float(rand_origin),
                                 _ptr(h) if h is not None else None, _ptr(out), _ptr(cur), y0, y1,
                                 C.cast(C.byref(st), C.c_void_p) if want_stats else None)
        stats = {k: getattr(st, k) for k, _ in OrStats._fields_} if want_stats else None
        return out, cur, stats

    def render(self, cam_np, first_loopnum, rand_origins, want_stats=False):
        """Frames first_loopnum.. with the reference's running mean (fp32 history)."""
        hist = np.zeros((self.height, self.width, 3), dtype=np.float32)
        total = None
        for k, ro in enumerate(rand_origins):
            hist, _, st = self.render_frame(cam_np, first_loopnum + k, float(ro), hist, want_stats=want_stats)
            if st:
                total = st if total is None else {a: total[a] + st[a] for a in st}
        return hist, total

    def render_band(self, cam_np, first_loopnum, rand_origins, rows, mean=None, total=None):
        """Frames first_loopnum.. over rows [y0, y1): (running mean, plain sum) of those rows, each
        (y1 - y0, W, 3); `mean` / `total` continue earlier bands of the same rows."""
        y0, y1 = rows
        m = np.zeros((self.height, self.width, 3), dtype=np.float32)
        s = np.zeros((self.height, self.width, 3), dtype=np.float32)
        if mean is not None:
            m[y0:y1] = mean
        if total is not None:
            s[y0:y1] = total
        ro = np.ascontiguousarray(rand_origins, dtype=np.float32)
        cam = self.camera(cam_np, first_loopnum)
        self.lib.or_render_band(C.byref(self.s), C.byref(cam), first_loopnum, ro.shape[0], _ptr(ro), _ptr(m), _ptr(s), y0, y1)
        return m[y0:y1].copy(), s[y0:y1].copy()

    def hit_bvh(self, origins, dirs):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        t = np.full(o.shape[0], -1.0, dtype=np.float32)
        ids = np.full(o.shape[0], -1, dtype=np.int32)
        self.lib.or_hit_bvh_many(C.byref(self.s), _ptr(o), _ptr(d), o.shape[0], _ptr(t), _ptr(ids))
        return t, ids


def scene_from_build(tris, nodes, width, height, sobol_v, spheres=None, **kw):
    """Oracle scene from builder output (either the oracle's or the library's arrays)."""
    tt, tw, tb, bw = pack_textures(tris, nodes)
    ints = None
    if nodes.shape[0] > FLOAT_EXACT_INTS:
        # beyond what the reference's float-encoded node indices can express: hand the oracle the
        # integers themselves (OrScene.bvhInts)
        ints = np.stack([nodes["triangleNum"], nodes["offsetOrRc"], nodes["axis"]], axis=1)
    return OracleScene(tt, tw, tris.shape[0], tb, bw, nodes.shape[0], width, height, sobol_v, spheres, bvh_ints=ints, **kw)

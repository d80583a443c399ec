"""ctypes binding of librt_amd.so plus a thin host-side mirror of the reference's main loop.

The compute path is the HIP library only: importing this package raises if the library is
missing, and nothing here falls back to a CPU implementation.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIBRARY") or os.path.join(_HERE, "librt_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python raytracing_amd/build.py` "
        "(hipcc --offload-arch=gfx950); there is no CPU fallback")



def _share_hip_runtime():
    """One HIP runtime per process, whatever the import order.

    librt_amd.so asks the loader for `libamdhip64.so.7`; torch ships a copy of the same runtime
    in its own lib directory and asks for it as `libamdhip64.so`.  The loader matches a request
    against the libraries it already holds by file and by soname (`libamdhip64.so.7` in both
    copies): with torch first, this library binds to torch's copy.  The other way round the
    loader would map torch's file as a second runtime, and the second one finds no GPU.  So when
    torch is installed but not loaded yet, its copy is loaded here first; a later `import torch`
    then finds its own file already mapped.  RT_AMD_HIP_RUNTIME names another file to share."""
    override = os.environ.get("RT_AMD_HIP_RUNTIME")
    if override:
        C.CDLL(override, mode=C.RTLD_GLOBAL)
        return override
    if "torch" in sys.modules:
        return None
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    C.CDLL(cand, mode=C.RTLD_GLOBAL)
    return cand


SHARED_HIP_RUNTIME = _share_hip_runtime()
lib = C.CDLL(LIB_PATH)

RT_MAX_SPHERES = 10
RT_SOBOL_WORDS = 512
RT_MAX_BOUNCES = 8
RT_MAX_TRIANGLES = 50000000
RT_NOALLOC_MAX_RAYS = 262144
RT_SWEEP_MAX_TRIANGLES = 16384
PIPELINE_WAVEFRONT = 0
PIPELINE_MEGAKERNEL = 1
PIPELINE_PERSISTENT = 2
INDEX_BUILD_HOST = 0
INDEX_BUILD_GPU = 1
INDEX_BUILD_AUTO = 2        # default: on the host below INDEX_AUTO_GPU_TRIANGLES triangles, on the device from there on
INDEX_AUTO_GPU_TRIANGLES = 16000000

# numpy views of the C structs of include/rt_amd.h
TRIANGLE_DTYPE = np.dtype([("v", "<f4", (3, 3)), ("n", "<f4", (3, 3)), ("tex", "<f4", (3, 2)),
                           ("ka", "<f4", 3), ("kd", "<f4", 3), ("ks", "<f4", 3), ("center", "<f4", 3)])
NODE_DTYPE = np.dtype([("minBound", "<f4", 3), ("maxBound", "<f4", 3),
                       ("triangleNum", "<i4"), ("offsetOrRc", "<i4"), ("axis", "<i4")])
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("albedo", "<f4", 3),
                         ("specularRate", "<f4"), ("refractionRate", "<f4"), ("roughness", "<f4")])
CAMERA_DTYPE = np.dtype([("camPos", "<f4", 3), ("front", "<f4", 3), ("right", "<f4", 3), ("up", "<f4", 3),
                         ("halfH", "<f4"), ("halfW", "<f4"), ("leftbottom", "<f4", 3), ("LoopNum", "<i4")])
assert TRIANGLE_DTYPE.itemsize == 144 and NODE_DTYPE.itemsize == 36
assert SPHERE_DTYPE.itemsize == 40 and CAMERA_DTYPE.itemsize == 72


class RtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("tri_hits", C.c_uint64), ("frames", C.c_uint64), ("retraced", C.c_uint64)]


class RtCameraRig(C.Structure):
    """class Camera of the reference (Camera.h:16-139), field for field."""
    _fields_ = [("cameraPos", C.c_float * 3), ("cameraFront", C.c_float * 3), ("cameraUp", C.c_float * 3),
                ("cameraRight", C.c_float * 3), ("worldUp", C.c_float * 3),
                ("fov", C.c_float), ("Pitch", C.c_float), ("Yaw", C.c_float), ("cameraSpeed", C.c_float),
                ("firstMouse", C.c_int32), ("lastX", C.c_float), ("lastY", C.c_float),
                ("ScreenRatio", C.c_float), ("halfH", C.c_float), ("halfW", C.c_float),
                ("LeftBottomCorner", C.c_float * 3), ("LoopNum", C.c_int32)]


class RtSceneInfo(C.Structure):
    _fields_ = [("triangles", C.c_int64), ("index_nodes", C.c_int64), ("tree_height", C.c_int32),
                ("index_stack_need", C.c_int32), ("index_stack_lds", C.c_int32), ("index_enabled", C.c_int32),
                ("paths_waves_per_simd", C.c_int32), ("frames_per_launch", C.c_int32),
                ("index_built_on_device", C.c_int32), ("index_build_fell_back", C.c_int32)]


class RtOptions(C.Structure):
    _fields_ = [("max_bounces", C.c_int32), ("pipeline", C.c_int32), ("cull", C.c_int32),
                ("count_work", C.c_int32), ("accel", C.c_int32), ("history8", C.c_int32), ("overlap", C.c_int32),
                ("index_build", C.c_int32), ("shading_normals", C.c_int32), ("albedo_texture", C.c_int32),
                ("coalesce", C.c_int32)]


_vp, _i64, _int = C.c_void_p, C.c_int64, C.c_int
_PROTOS = {
    "rt_bvh_build": (_int, [_vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    "rt_bvh_build_sweep": (_int, [_vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    "rt_selftest_sort": (_int, [_vp, _vp, _i64]),
    "rt_texture_dims": (_int, [_i64, _int, C.POINTER(_int), C.POINTER(_int)]),
    "rt_pack_triangle_texture": (_int, [_vp, _i64, _vp, _int, _int]),
    "rt_pack_bvh_texture": (_int, [_vp, _i64, _vp, _int, _int]),
    "rt_sobol_default_table": (_int, [_vp]),
    "rt_gray_code": (C.c_uint32, [C.c_uint32]),
    "rt_sobol": (C.c_float, [_vp, C.c_uint32, C.c_uint32]),
    "rt_camera_from_angles": (_int, [_vp, _vp, C.c_float, C.c_float, C.c_float, _int, _int]),
    "rt_camera_default": (_int, [_vp, _int, _int]),
    "rt_rand_origin": (C.c_float, [C.POINTER(C.c_uint32)]),
    "rt_camera_rig_init": (_int, [C.POINTER(RtCameraRig), _int, _int]),
    "rt_camera_rig_update_vectors": (_int, [C.POINTER(RtCameraRig)]),
    "rt_camera_rig_update_fov": (_int, [C.POINTER(RtCameraRig), C.c_float]),
    "rt_camera_rig_update_front": (_int, [C.POINTER(RtCameraRig), C.c_float, C.c_float]),
    "rt_camera_rig_update_screen_ratio": (_int, [C.POINTER(RtCameraRig), _int, _int]),
    "rt_camera_rig_process_keyboard": (_int, [C.POINTER(RtCameraRig), _int, C.c_float]),
    "rt_camera_rig_loop_increase": (_int, [C.POINTER(RtCameraRig)]),
    "rt_camera_rig_uniforms": (_int, [C.POINTER(RtCameraRig), _vp]),
    "rt_scene_create": (_int, [C.POINTER(_vp)]),
    "rt_scene_destroy": (None, [_vp]),
    "rt_scene_push_obj": (_int, [_vp, C.c_char_p, _vp, _vp]),
    "rt_scene_push_triangles": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "rt_scene_push_sphere": (_int, [_vp, _vp]),
    "rt_scene_push_demo": (_int, [_vp]),
    "rt_scene_push_synthetic": (_int, [_vp, _int, _int, _int, C.c_uint32]),
    "rt_scene_triangle_count": (_i64, [_vp]),
    "rt_scene_sphere_count": (_int, [_vp]),
    "rt_scene_get_triangles": (_int, [_vp, _vp, _i64]),
    "rt_scene_get_spheres": (_int, [_vp, _vp, _int]),
    "rt_create": (_int, [C.POINTER(_vp), _int, _int, _int]),
    "rt_destroy": (None, [_vp]),
    "rt_set_stream": (_int, [_vp, _vp]),
    "rt_default_options": (_int, [C.POINTER(RtOptions)]),
    "rt_set_options": (_int, [_vp, C.POINTER(RtOptions)]),
    "rt_scene_upload": (_int, [_vp, _vp, _i64, _vp, _i64]),
    "rt_scene_upload_textures": (_int, [_vp, _vp, _i64, _vp, _i64]),
    "rt_set_sobol": (_int, [_vp, _vp]),
    "rt_set_albedo_texture": (_int, [_vp, _vp, _int, _int]),
    "rt_set_spheres": (_int, [_vp, _vp, _int]),
    "rt_set_camera": (_int, [_vp, _vp]),
    "rt_render_frames": (_int, [_vp, _int, _int, _vp]),
    "rt_reset": (_int, [_vp]),
    "rt_synchronize": (_int, [_vp]),
    "rt_read_mean": (_int, [_vp, _vp]),
    "rt_read_sum": (_int, [_vp, _vp]),
    "rt_read_mean_async": (_int, [_vp, _vp]),
    "rt_read_wait": (_int, [_vp]),
    "rt_mean_device_ptr": (_int, [_vp, C.POINTER(_vp)]),
    "rt_host_alloc": (_vp, [C.c_size_t]),
    "rt_host_free": (None, [_vp]),
    "rt_restore": (_int, [_vp, _vp, _vp, C.c_uint64]),
    "rt_sum_device_ptr": (_int, [_vp, C.POINTER(_vp)]),
    "rt_bind_sum_buffer": (_int, [_vp, _vp]),
    "rt_get_stats": (_int, [_vp, C.POINTER(RtStats)]),
    "rt_get_kernel_time": (_int, [_vp, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "rt_trace_rays": (_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "rt_trace_rays_noalloc": (_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "rt_get_scene_info": (_int, [_vp, C.POINTER(RtSceneInfo)]),
    "rt_selftest_nth_element": (_int, [_vp, _vp, _i64, _i64]),
    "rt_selftest_arith": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "rt_selftest_index_build": (_int, [_vp, _i64, C.c_float, _vp, _i64, C.POINTER(_i64), C.POINTER(C.c_double)]),
    "rt_selftest_exact_math": (_int, [_vp, C.POINTER(C.c_uint64)]),
    "rt_selftest_device_sort": (_int, [_vp, _vp, _vp, _i64]),
    "rt_selftest_device_scan": (_int, [_vp, _vp, _i64]),
    "rt_last_error": (C.c_char_p, [_vp]),
    "rt_version": (C.c_char_p, []),
}
for _name, (_res, _args) in _PROTOS.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


class RtError(RuntimeError):
    pass


def _check(rc, ctx=None):
    if rc != 0:
        msg = lib.rt_last_error(ctx)
        raise RtError(f"rt_amd error {rc}: {msg.decode() if msg else ''}")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ host-side helpers

def sobol_default_table():
    v = np.zeros(RT_SOBOL_WORDS, dtype=np.uint32)
    _check(lib.rt_sobol_default_table(_ptr(v)))
    return v


def camera_default(width, height):
    cam = np.zeros(1, dtype=CAMERA_DTYPE)
    _check(lib.rt_camera_default(_ptr(cam), width, height))
    return cam


def camera_from_angles(pos, yaw, pitch, fov, width, height):
    cam = np.zeros(1, dtype=CAMERA_DTYPE)
    p = np.asarray(pos, dtype=np.float32)
    _check(lib.rt_camera_from_angles(_ptr(cam), _ptr(p), yaw, pitch, fov, width, height))
    return cam


class CameraRig:
    """The reference's interactive camera (Camera.h); `uniforms()` gives what rt_set_camera takes."""
    FORWARD, BACKWARD, LEFT, RIGHT = 0, 1, 2, 3

    def __init__(self, width, height):
        self.s = RtCameraRig()
        _check(lib.rt_camera_rig_init(C.byref(self.s), width, height))

    def update_fov(self, offset):
        _check(lib.rt_camera_rig_update_fov(C.byref(self.s), offset))

    def update_front(self, xpos, ypos):
        _check(lib.rt_camera_rig_update_front(C.byref(self.s), xpos, ypos))

    def update_screen_ratio(self, width, height):
        _check(lib.rt_camera_rig_update_screen_ratio(C.byref(self.s), width, height))

    def process_keyboard(self, direction, delta_time):
        _check(lib.rt_camera_rig_process_keyboard(C.byref(self.s), direction, delta_time))

    def loop_increase(self):
        _check(lib.rt_camera_rig_loop_increase(C.byref(self.s)))

    def uniforms(self):
        cam = np.zeros(1, dtype=CAMERA_DTYPE)
        _check(lib.rt_camera_rig_uniforms(C.byref(self.s), _ptr(cam)))
        return cam


def rand_origins(seed, count):
    """The explicit stand-in for the reference's time-seeded rand(): count randOrigin values."""
    state = C.c_uint32(seed)
    return np.array([lib.rt_rand_origin(C.byref(state)) for _ in range(count)], dtype=np.float32)


def bvh_build(tris, sweep=False):
    """Returns (permuted triangles, nodes); `tris` is not modified.  sweep: the split rule the
    reference's author left commented out (rt_bvh_build_sweep) instead of the median."""
    tris = np.ascontiguousarray(tris, dtype=TRIANGLE_DTYPE).copy()
    n = tris.shape[0]
    nodes = np.zeros(max(2 * n - 1, 1), dtype=NODE_DTYPE)
    cnt = C.c_int64(0)
    _check((lib.rt_bvh_build_sweep if sweep else lib.rt_bvh_build)(_ptr(tris), n, _ptr(nodes), nodes.shape[0], C.byref(cnt)))
    return tris, nodes[:cnt.value]


WIDE_NODE_DTYPE = np.dtype([("origin", "<f4", 3), ("scaleX", "<f4"), ("ref", "<u4", 4), ("qlo", "<u4", 3), ("qhi", "<u4", 3),
                            ("scaleY", "<f4"), ("scaleZ", "<f4")])
assert WIDE_NODE_DTYPE.itemsize == 64


def index_build_host(tris, want_nodes=True):
    """The host build of the search index alone (no GPU): (4-wide nodes or None, stats dict)."""
    tris = np.ascontiguousarray(tris, dtype=TRIANGLE_DTYPE)
    n = tris.shape[0]
    max_abs = float(np.abs(tris["v"]).max())
    cnt = C.c_int64(0)
    stats = (C.c_double * 4)()
    nodes = np.zeros(max(n, 1), dtype=WIDE_NODE_DTYPE) if want_nodes else None
    _check(lib.rt_selftest_index_build(_ptr(tris), n, max_abs, _ptr(nodes) if want_nodes else None, max(n, 1), C.byref(cnt), stats))
    info = {"build_ms": stats[0], "levels": int(stats[1]), "expected_visits": stats[2], "stack_bound": int(stats[3]), "nodes": cnt.value}
    return (nodes[:cnt.value] if want_nodes else None), info


def pack_textures(tris, nodes):
    """The reference's two R32F textures as flat arrays: (texTriangle, width, texBVH, width)."""
    x, y = C.c_int(), C.c_int()
    _check(lib.rt_texture_dims(tris.shape[0], 33, C.byref(x), C.byref(y)))
    tt = np.zeros(x.value * y.value * 33, dtype=np.float32)
    _check(lib.rt_pack_triangle_texture(_ptr(tris), tris.shape[0], _ptr(tt), x.value, y.value))
    tw = x.value * 33
    _check(lib.rt_texture_dims(nodes.shape[0], 9, C.byref(x), C.byref(y)))
    tb = np.zeros(x.value * y.value * 9, dtype=np.float32)
    _check(lib.rt_pack_bvh_texture(_ptr(nodes), nodes.shape[0], _ptr(tb), x.value, y.value))
    return tt, tw, tb, x.value * 9


class Scene:
    """Host-side scene assembly (the reference's Scene + Model loading)."""

    def __init__(self):
        h = C.c_void_p()
        _check(lib.rt_scene_create(C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib.rt_scene_destroy(self._h)
            self._h = None

    def push_obj(self, path, scale=(1, 1, 1), translation=(0, 0, 0)):
        s = np.asarray(scale, dtype=np.float32)
        t = np.asarray(translation, dtype=np.float32)
        _check(lib.rt_scene_push_obj(self._h, os.fsencode(path), _ptr(s), _ptr(t)))

    def push_triangles(self, tris, scale=(1, 1, 1), translation=(0, 0, 0)):
        tris = np.ascontiguousarray(tris, dtype=TRIANGLE_DTYPE)
        s = np.asarray(scale, dtype=np.float32)
        t = np.asarray(translation, dtype=np.float32)
        _check(lib.rt_scene_push_triangles(self._h, _ptr(tris), tris.shape[0], _ptr(s), _ptr(t)))

    def push_sphere(self, center, radius, albedo, specular_rate, refraction_rate, roughness=0.0):
        s = np.zeros(1, dtype=SPHERE_DTYPE)
        s["center"] = center
        s["radius"] = radius
        s["albedo"] = albedo
        s["specularRate"] = specular_rate
        s["refractionRate"] = refraction_rate
        s["roughness"] = roughness
        _check(lib.rt_scene_push_sphere(self._h, _ptr(s)))

    def push_demo(self):
        _check(lib.rt_scene_push_demo(self._h))

    def push_synthetic(self, kind, level, copies=1, seed=12345):
        _check(lib.rt_scene_push_synthetic(self._h, kind, level, copies, seed))

    def triangles(self):
        n = lib.rt_scene_triangle_count(self._h)
        out = np.zeros(n, dtype=TRIANGLE_DTYPE)
        if n:
            _check(lib.rt_scene_get_triangles(self._h, _ptr(out), n))
        return out

    def spheres(self):
        n = lib.rt_scene_sphere_count(self._h)
        out = np.zeros(n, dtype=SPHERE_DTYPE)
        if n:
            _check(lib.rt_scene_get_spheres(self._h, _ptr(out), n))
        return out


# ------------------------------------------------------------------ device side

class Renderer:
    """One GPU context: the shader pass of the reference's main loop."""

    def __init__(self, width, height, device=0):
        h = C.c_void_p()
        _check(lib.rt_create(C.byref(h), width, height, device))
        self._h = h
        self.width, self.height = width, height
        self._keep = []
        self._pinned = []

    def close(self):
        if getattr(self, "_h", None):
            lib.rt_destroy(self._h)         # waits for everything in flight, copies into host_image() arrays included
            self._h = None
            for p in self._pinned:
                lib.rt_host_free(p)
            self._pinned = []

    def __del__(self):
        self.close()

    def set_stream(self, stream_handle):
        _check(lib.rt_set_stream(self._h, C.c_void_p(stream_handle)), self._h)

    def set_options(self, max_bounces=4, pipeline=PIPELINE_PERSISTENT, cull=True, count_work=False, accel=True,
                    history8=False, overlap=True, index_build=2, shading_normals=False, albedo_texture=False, coalesce=True):
        o = RtOptions(max_bounces, pipeline, int(cull), int(count_work), int(accel), int(history8), int(overlap), int(index_build),
                      int(shading_normals), int(albedo_texture), int(coalesce))
        _check(lib.rt_set_options(self._h, C.byref(o)), self._h)

    def upload(self, tris, nodes):
        tris = np.ascontiguousarray(tris, dtype=TRIANGLE_DTYPE)
        nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
        _check(lib.rt_scene_upload(self._h, _ptr(tris), tris.shape[0], _ptr(nodes), nodes.shape[0]), self._h)

    def upload_textures(self, tex_triangle, n, tex_bvh, m):
        tt = np.ascontiguousarray(tex_triangle, dtype=np.float32)
        tb = np.ascontiguousarray(tex_bvh, dtype=np.float32)
        _check(lib.rt_scene_upload_textures(self._h, _ptr(tt), n, _ptr(tb), m), self._h)

    def set_albedo_texture(self, rgb):
        """rgb: (H, W, 3) uint8, row 0 at v = 0 (the order glTexImage2D takes rows in); None removes it.
        Used by renders with set_options(albedo_texture=True)."""
        if rgb is None:
            _check(lib.rt_set_albedo_texture(self._h, None, 0, 0), self._h)
            return
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        if rgb.ndim != 3 or rgb.shape[2] != 3:
            raise ValueError("albedo texture must be (H, W, 3) uint8")
        _check(lib.rt_set_albedo_texture(self._h, _ptr(rgb), rgb.shape[1], rgb.shape[0]), self._h)

    def set_sobol(self, v):
        v = np.ascontiguousarray(v, dtype=np.uint32)
        assert v.shape[0] == RT_SOBOL_WORDS
        _check(lib.rt_set_sobol(self._h, _ptr(v)), self._h)

    def set_spheres(self, spheres):
        s = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        _check(lib.rt_set_spheres(self._h, _ptr(s), s.shape[0]), self._h)

    def set_camera(self, cam):
        cam = np.ascontiguousarray(cam, dtype=CAMERA_DTYPE)
        _check(lib.rt_set_camera(self._h, _ptr(cam)), self._h)

    def render(self, first_loopnum, rand_origin):
        r = np.ascontiguousarray(rand_origin, dtype=np.float32)
        _check(lib.rt_render_frames(self._h, first_loopnum, r.shape[0], _ptr(r)), self._h)

    def reset(self):
        _check(lib.rt_reset(self._h), self._h)

    def synchronize(self):
        _check(lib.rt_synchronize(self._h), self._h)

    def read_mean(self):
        out = np.zeros((self.height, self.width, 3), dtype=np.float32)
        _check(lib.rt_read_mean(self._h, _ptr(out)), self._h)
        return out

    def host_image(self):
        """A (height, width, 3) float32 array in page-locked host memory for read_mean_async; freed with the renderer."""
        nbytes = self.height * self.width * 3 * 4
        p = lib.rt_host_alloc(nbytes)
        if not p:
            raise RtError("rt_host_alloc failed")
        self._pinned.append(p)
        return np.ctypeslib.as_array((C.c_float * (nbytes // 4)).from_address(p)).reshape(self.height, self.width, 3)

    def read_mean_async(self, out):
        """Enqueues the copy of the running mean into `out` (from host_image()); read_wait() blocks until it is there."""
        assert out.dtype == np.float32 and out.shape == (self.height, self.width, 3) and out.flags["C_CONTIGUOUS"]
        _check(lib.rt_read_mean_async(self._h, _ptr(out)), self._h)

    def read_wait(self):
        _check(lib.rt_read_wait(self._h), self._h)

    def mean_device_ptr(self):
        p = C.c_void_p()
        _check(lib.rt_mean_device_ptr(self._h, C.byref(p)), self._h)
        return p.value

    def read_sum(self):
        out = np.zeros((self.height, self.width, 3), dtype=np.float32)
        _check(lib.rt_read_sum(self._h, _ptr(out)), self._h)
        return out

    def restore(self, mean, sum_, frames):
        """Resume: load a running mean and a sum saved earlier; continue with render(frames + 1, ...)."""
        m = None if mean is None else np.ascontiguousarray(mean, dtype=np.float32)
        s = None if sum_ is None else np.ascontiguousarray(sum_, dtype=np.float32)
        for a in (m, s):
            if a is not None and a.shape != (self.height, self.width, 3):
                raise ValueError("restore: images must be (height, width, 3)")
        _check(lib.rt_restore(self._h, None if m is None else _ptr(m), None if s is None else _ptr(s), frames), self._h)

    def bind_sum_buffer(self, dev_ptr):
        _check(lib.rt_bind_sum_buffer(self._h, C.c_void_p(dev_ptr)), self._h)

    def sum_device_ptr(self):
        p = C.c_void_p()
        _check(lib.rt_sum_device_ptr(self._h, C.byref(p)), self._h)
        return p.value

    def stats(self):
        s = RtStats()
        _check(lib.rt_get_stats(self._h, C.byref(s)), self._h)
        return {k: getattr(s, k) for k, _ in RtStats._fields_}

    def scene_info(self):
        s = RtSceneInfo()
        _check(lib.rt_get_scene_info(self._h, C.byref(s)), self._h)
        return {k: getattr(s, k) for k, _ in RtSceneInfo._fields_}

    def trace_rays_noalloc(self, d_origins, d_dirs, n, d_t_out, d_id_out):
        """Device pointers (ints) in, device pointers out; asynchronous on the context's stream."""
        _check(lib.rt_trace_rays_noalloc(self._h, C.c_void_p(d_origins), C.c_void_p(d_dirs), n,
                                         C.c_void_p(d_t_out), C.c_void_p(d_id_out)), self._h)

    def kernel_time(self):
        ms, n = C.c_double(), C.c_int64()
        _check(lib.rt_get_kernel_time(self._h, C.byref(ms), C.byref(n)), self._h)
        return ms.value, n.value

    def trace_rays(self, origins, dirs):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        t = np.zeros(o.shape[0], dtype=np.float32)
        ids = np.zeros(o.shape[0], dtype=np.int32)
        _check(lib.rt_trace_rays(self._h, _ptr(o), _ptr(d), o.shape[0], _ptr(t), _ptr(ids)), self._h)
        return t, ids

    def selftest_exact_math(self):
        """Mismatches of the device's 1/x, sqrt, 1/sqrt against IEEE over all 2^32 inputs and of the pixel
        coordinate (X + 0.5) / W over every pixel of every width (four counts)."""
        out = (C.c_uint64 * 4)()
        _check(lib.rt_selftest_exact_math(self._h, out), self._h)
        return [int(v) for v in out]

    def selftest_device_sort(self, keys, vals):
        """The index build's radix sort on the device: (keys uint64, vals uint32) sorted by key, stable."""
        k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
        v = np.ascontiguousarray(vals, dtype=np.uint32).copy()
        assert k.shape == v.shape and k.ndim == 1
        _check(lib.rt_selftest_device_sort(self._h, _ptr(k), _ptr(v), k.shape[0]), self._h)
        return k, v

    def selftest_device_scan(self, values):
        """The index build's exclusive prefix sum on the device (uint64)."""
        x = np.ascontiguousarray(values, dtype=np.uint64).copy()
        _check(lib.rt_selftest_device_scan(self._h, _ptr(x), x.shape[0]), self._h)
        return x

    def selftest_arith(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        out = np.zeros((a.shape[0], 5), dtype=np.float32)
        _check(lib.rt_selftest_arith(self._h, _ptr(a), _ptr(b), a.shape[0], _ptr(out)), self._h)
        return out


def make_renderer(scene, width, height, device=0, camera=None, **options):
    """main.cpp:101-121 in one call: build the BVH of `scene`, upload it, set camera and spheres."""
    tris, nodes = bvh_build(scene.triangles())
    r = Renderer(width, height, device)
    if options:
        r.set_options(**options)        # index_build is read at upload
    r.upload(tris, nodes)
    r.set_spheres(scene.spheres())
    r.set_camera(camera if camera is not None else camera_default(width, height))
    if options:
        r.set_options(**options)
    return r

"""Can this machine create an OpenGL 3.3 context without a display?

The reference's shader (GLSL 330 core) could run on the host's cores under Mesa llvmpipe, as a
context baseline next to the GPU figure, if a context can be created with no window system: an
EGL surfaceless or device platform, OSMesa, or GLX against a running X server.  probe() tries
each way as far as creating the display connection and reports what it found; it never raises.

    python tools/gl_probe.py        prints the result as JSON
"""
import ctypes as C
import ctypes.util
import glob
import json
import os


def _try_load(names):
    for n in names:
        try:
            return C.CDLL(n), n
        except OSError:
            continue
    return None, None


def probe():
    found = {}
    reasons = []
    # --- EGL: surfaceless / device platforms need no window system
    egl, egl_name = _try_load(["libEGL.so.1", "libEGL.so"])
    found["libEGL"] = egl_name
    egl_ok = False
    if egl is None:
        reasons.append("libEGL is not installed")
    else:
        try:
            egl.eglGetProcAddress.restype = C.c_void_p
            egl.eglGetProcAddress.argtypes = [C.c_char_p]
            get_platform_display = egl.eglGetProcAddress(b"eglGetPlatformDisplayEXT")
            EGL_PLATFORM_SURFACELESS_MESA = 0x31DD
            dpy = None
            if get_platform_display:
                fn = C.CFUNCTYPE(C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p)(get_platform_display)
                dpy = fn(EGL_PLATFORM_SURFACELESS_MESA, None, None)
            if not dpy:
                egl.eglGetDisplay.restype = C.c_void_p
                egl.eglGetDisplay.argtypes = [C.c_void_p]
                dpy = egl.eglGetDisplay(None)
            major, minor = C.c_int(0), C.c_int(0)
            egl.eglInitialize.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
            if dpy and egl.eglInitialize(dpy, C.byref(major), C.byref(minor)):
                egl_ok = True
                found["egl_version"] = f"{major.value}.{minor.value}"
                EGL_OPENGL_API = 0x30A2
                egl.eglBindAPI.argtypes = [C.c_uint]
                if not egl.eglBindAPI(EGL_OPENGL_API):
                    egl_ok = False
                    reasons.append("EGL initialises but offers no desktop OpenGL API (GLES only; the shader is #version 330 core)")
                egl.eglTerminate.argtypes = [C.c_void_p]
                egl.eglTerminate(dpy)
            else:
                reasons.append("libEGL is present but no display could be initialised (no surfaceless/device platform)")
        except (AttributeError, OSError, ValueError) as e:      # a libEGL without these entry points
            reasons.append(f"libEGL probe failed: {e}")
    # --- OSMesa: off-screen Mesa without any window system
    osm, osm_name = _try_load(["libOSMesa.so.8", "libOSMesa.so.6", "libOSMesa.so"])
    found["libOSMesa"] = osm_name
    osmesa_ok = osm is not None and hasattr(osm, "OSMesaCreateContextAttribs")
    if osm is None:
        reasons.append("libOSMesa is not installed")
    elif not osmesa_ok:
        reasons.append("libOSMesa has no OSMesaCreateContextAttribs (no core-profile contexts)")
    # --- GLX: needs an X server
    glx, glx_name = _try_load(["libGL.so.1"])
    found["libGL"] = glx_name
    found["DISPLAY"] = os.environ.get("DISPLAY")
    found["dri_nodes"] = sorted(glob.glob("/dev/dri/*"))
    found["x_server_binaries"] = [p for p in ("/usr/bin/Xvfb", "/usr/bin/Xorg", "/usr/bin/Xvnc", "/usr/bin/xvfb-run") if os.path.exists(p)]
    glx_ok = False
    if glx is None:
        reasons.append("libGL is not installed")
    elif not found["DISPLAY"]:
        reasons.append("libGL (GLX) is installed but DISPLAY is unset and no X server binary (Xvfb, Xorg) is installed to start one"
                       if not found["x_server_binaries"] else "libGL (GLX) is installed, DISPLAY is unset")
    else:
        x11, _ = _try_load(["libX11.so.6"])
        if x11 is not None:
            x11.XOpenDisplay.restype = C.c_void_p
            x11.XOpenDisplay.argtypes = [C.c_char_p]
            d = x11.XOpenDisplay(None)
            glx_ok = bool(d)
            if d:
                x11.XCloseDisplay.argtypes = [C.c_void_p]
                x11.XCloseDisplay(d)
            else:
                reasons.append("DISPLAY is set but the X server does not answer")
    found["swrast_dri"] = sorted(glob.glob("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so"))
    found["glsl_compilers"] = [p for p in ("/usr/bin/glslangValidator", "/usr/bin/glslc") if os.path.exists(p)]
    available = egl_ok or osmesa_ok or glx_ok
    out = {"available": bool(available), "found": found}
    if available:
        out["via"] = "egl" if egl_ok else ("osmesa" if osmesa_ok else "glx")
        out["reason"] = ("a context can be created, but this build has no GLSL harness for it: the shader's "
                         "semantics are measured through the C port (cpu_baseline.value)")
    else:
        out["reason"] = "no way to create an OpenGL context on this machine: " + "; ".join(reasons)
    return out


if __name__ == "__main__":
    print(json.dumps(probe(), indent=1))

import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import raytracing_amd as rt
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (96, 54)
sc = rt.Scene(); sc.push_demo()
tris, nodes = rt.bvh_build(sc.triangles())
cam = rt.camera_default(W, H)
def run(pool, n, overlap=True):
    os.environ["RT_AMD_POOL"] = "1" if pool else "0"
    r = rt.Renderer(W, H); r.upload(tris, nodes); r.set_spheres(sc.spheres()); r.set_camera(cam); r.set_options(overlap=overlap)
    ro = rt.rand_origins(20221003, n)
    r.render(1, ro)
    m = r.read_mean(); s = r.stats(); r.close(); return m, s
for n in ((1, 2, 3) if len(sys.argv) > 2 else (1, 8, 64, 128, 256)):
    a, sa = run(False, n); b, sb = run(True, n); c, sc_ = run(True, n, overlap=False)
    bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    badc = ~((a == c) | (np.isnan(a) & np.isnan(c)))
    print(n, "rays", sa["rays"], sb["rays"], sc_["rays"], "differing values", bad.sum(), "nan", np.isnan(b).sum(), "| no overlap: differing", badc.sum(), "nan", np.isnan(c).sum())
    if len(sys.argv) > 2: print("  images", a.reshape(-1)[:6], b.reshape(-1)[:6], sa, sb)
    if bad.any():
        ys, xs, _ = np.nonzero(bad); print("   first px", list(zip(xs[:6], ys[:6])), a[ys[0], xs[0]], b[ys[0], xs[0]])

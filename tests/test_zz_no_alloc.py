"""rt_trace_rays_noalloc (device pointers in and out, working memory in a fixed array inside the
library, no allocation) against rt_trace_rays on the same rays: triangle indices and distances
must be equal, on two calls in a row of which the second is smaller (the array keeps the first
call's contents)."""
import numpy as np
import pytest

import raytracing_amd as rt
import scenes

pytestmark = pytest.mark.gpu


def _device_call(r, torch, o, d):
    n = o.shape[0]
    do = torch.zeros((max(n, 1), 3), dtype=torch.float32, device="cuda")      # never an empty (null) allocation
    dd = torch.zeros((max(n, 1), 3), dtype=torch.float32, device="cuda")
    do[:n] = torch.from_numpy(o)
    dd[:n] = torch.from_numpy(d)
    dt = torch.full((max(n, 1),), 7.0, dtype=torch.float32, device="cuda")
    di = torch.full((max(n, 1),), 7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    r.trace_rays_noalloc(do.data_ptr(), dd.data_ptr(), n, dt.data_ptr(), di.data_ptr())
    r.synchronize()
    return dt.cpu().numpy()[:n], di.cpu().numpy()[:n]


@pytest.mark.parametrize("accel", [True, False])
def test_noalloc_trace_equals_trace_rays(accel):
    import torch
    sc = rt.Scene()
    sc.push_synthetic(0, 4, 1, 12345)                       # room + 5120-triangle mesh
    tris, nodes = rt.bvh_build(sc.triangles())
    r = rt.Renderer(8, 8)
    r.upload(tris, nodes)
    r.set_options(accel=accel)
    o1, d1 = scenes.random_rays(100000, 21, extent=4.5)
    o1[:, 1] = np.abs(o1[:, 1]) + 0.5
    o2, d2 = scenes.random_rays(1237, 22, extent=4.5)       # smaller, not a multiple of 64
    o2[:, 1] = np.abs(o2[:, 1]) + 0.5
    for o, d in ((o1, d1), (o2, d2)):
        t_ref, id_ref = r.trace_rays(o, d)
        t, ids = _device_call(r, torch, o, d)
        assert np.array_equal(ids, id_ref)
        assert np.array_equal(t, t_ref)
        assert (id_ref >= 0).mean() > 0.5
    # size limit and argument checks: an error, not a fault
    big = torch.zeros(3, dtype=torch.float32, device="cuda")
    with pytest.raises(rt.RtError, match="RT_NOALLOC_MAX_RAYS"):
        r.trace_rays_noalloc(big.data_ptr(), big.data_ptr(), rt.RT_NOALLOC_MAX_RAYS + 1, big.data_ptr(), big.data_ptr())
    with pytest.raises(rt.RtError):
        r.trace_rays_noalloc(0, big.data_ptr(), 1, big.data_ptr(), big.data_ptr())
    r.close()


def test_noalloc_trace_maximum_size_and_empty():
    import torch
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    r = rt.Renderer(8, 8)
    r.upload(tris, nodes)
    n = rt.RT_NOALLOC_MAX_RAYS
    o, d = scenes.random_rays(n, 23, extent=4.0)
    o[:, 1] = np.abs(o[:, 1]) + 0.5
    t_ref, id_ref = r.trace_rays(o, d)
    t, ids = _device_call(r, torch, o, d)
    assert np.array_equal(ids, id_ref) and np.array_equal(t, t_ref)
    # n == 0: nothing is launched, nothing is written
    t0, i0 = _device_call(r, torch, o[:0], d[:0])
    assert t0.shape[0] == 0 and i0.shape[0] == 0
    # and a small call after the full-size one
    t, ids = _device_call(r, torch, o[:100], d[:100])
    assert np.array_equal(ids, id_ref[:100]) and np.array_equal(t, t_ref[:100])
    r.close()

"""Stream ordering of the overlapped launch slots (RtOptions.overlap).

The path kernels of consecutive batches run on internal streams; what they share with work on the
context's stream - the accumulators, the sum buffer a caller may own, slot 0's overflow columns
(also the stack of rt_trace_rays_noalloc's exact re-trace) - has to be ordered by events, not by
luck.  Every case here enqueues without synchronising in between and compares with the result of
the same steps done one at a time."""
import numpy as np
import pytest

import raytracing_amd as rt
import scenes

pytestmark = pytest.mark.gpu

W, H = 160, 96


def _hall():
    sc = rt.Scene()
    sc.push_synthetic(1, 6, 1, 12345)                       # column hall: diffuse / mirror / glass + two spheres
    return sc


def _eq(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_noalloc_trace_followed_by_render_without_synchronising():
    """rt_trace_rays_noalloc keeps the stack of its exact re-traces in the overflow columns that
    slot 0's path kernel uses; axis-parallel rays and rays from far away are re-traced by design."""
    import torch
    sc = _hall()
    tris, nodes = rt.bvh_build(sc.triangles())
    r = rt.Renderer(W, H)
    r.upload(tris, nodes)
    r.set_spheres(sc.spheres())
    r.set_camera(rt.camera_default(W, H))
    n = 200000
    o, d = scenes.random_rays(n, 41, extent=12.0, axis_aligned_fraction=0.5)
    o[n // 2:] *= np.float32(40.0)                          # far outside: the index hands these to the exact path
    t_want, id_want = r.trace_rays(o, d)
    ro = rt.rand_origins(9, 24)
    r.reset()
    r.render(1, ro)
    want_img = r.read_mean()
    do, dd = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    dt = torch.zeros(n, dtype=torch.float32, device="cuda")
    di = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        r.reset()
        r.trace_rays_noalloc(do.data_ptr(), dd.data_ptr(), n, dt.data_ptr(), di.data_ptr())
        for k in range(0, 24, 4):                           # slots 0, 1, 2, 0, ... start right behind the trace
            r.render(k + 1, ro[k:k + 4])
        r.trace_rays_noalloc(do.data_ptr(), dd.data_ptr(), n, dt.data_ptr(), di.data_ptr())
        r.synchronize()
        assert _eq(r.read_mean(), want_img)
        assert np.array_equal(di.cpu().numpy(), id_want) and np.array_equal(dt.cpu().numpy(), t_want)
    r.close()


def test_pipeline_and_overlap_switches_without_reset():
    """Frames 1..24 rendered in six runs of four by six different configurations, one after the
    other with no reset or synchronisation: megakernel, overlapped paths, wavefront, paths on the
    context's stream, overlapped paths one frame per call, wavefront in reference order.  Every
    configuration gives the oracle's samples bit for bit, so the running mean must equal that of
    one call."""
    sc = _hall()
    r = rt.make_renderer(sc, W, H)
    ro = rt.rand_origins(77, 24)
    r.render(1, ro)
    want = (r.read_mean(), r.read_sum(), r.stats()["rays"])
    runs = [dict(pipeline=rt.PIPELINE_MEGAKERNEL), dict(pipeline=rt.PIPELINE_PERSISTENT, overlap=True),
            dict(pipeline=rt.PIPELINE_WAVEFRONT), dict(pipeline=rt.PIPELINE_PERSISTENT, overlap=False),
            dict(pipeline=rt.PIPELINE_PERSISTENT, overlap=True), dict(pipeline=rt.PIPELINE_WAVEFRONT, accel=False)]
    for rep in range(2):
        r.reset()
        for k, opts in enumerate(runs):
            r.set_options(**opts)
            if k == 4:
                for f in range(4):
                    r.render(4 * k + f + 1, ro[4 * k + f:4 * k + f + 1])
            else:
                r.render(4 * k + 1, ro[4 * k:4 * k + 4])
        got = (r.read_mean(), r.read_sum(), r.stats()["rays"])
        assert _eq(got[0], want[0]) and _eq(got[1], want[1]) and got[2] == want[2]
    r.close()


def test_caller_work_on_the_context_stream_comes_before_the_accumulation():
    """A caller that owns the sum buffer (rt_bind_sum_buffer) clears it on the context's stream
    between two render calls, behind a kernel that keeps that stream busy for a while: the second
    call's frames must be added after the clearing, not before."""
    import torch
    if not hasattr(torch.cuda, "_sleep"):
        pytest.skip("torch.cuda._sleep is not available")
    sc = _hall()
    r = rt.make_renderer(sc, W, H)
    ro = rt.rand_origins(5, 16)
    r.render(9, ro[8:])
    want = r.read_sum()
    stream = torch.cuda.Stream()
    r.set_stream(stream.cuda_stream)
    sum_t = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    r.bind_sum_buffer(sum_t.data_ptr())
    for _ in range(3):
        r.reset()
        r.render(1, ro[:8])
        with torch.cuda.stream(stream):
            torch.cuda._sleep(200_000_000)                  # tens of milliseconds on the context's stream
            sum_t.zero_()
        r.render(9, ro[8:])                                 # a few hundred microseconds of path tracing
        r.synchronize()
        got = sum_t.cpu().numpy().reshape(H, W, 3)
        assert _eq(got, want)
    r.bind_sum_buffer(0)
    r.set_stream(0)
    r.close()


def test_explicit_gpu_index_build_falls_back_to_the_host(monkeypatch):
    """RT_INDEX_BUILD_GPU whose build gives up (here: forced through the test hook) must leave a
    working scene with a host-built index, as the automatic mode does."""
    monkeypatch.setenv("RT_AMD_TEST_FAIL_GPU_INDEX", "1")
    sc = _hall()
    tris, nodes = rt.bvh_build(sc.triangles())
    r = rt.Renderer(W, H)
    r.set_options(index_build=rt.INDEX_BUILD_GPU)
    r.upload(tris, nodes)
    info = r.scene_info()
    assert info["index_enabled"] == 1 and info["index_built_on_device"] == 0 and info["index_build_fell_back"] == 1
    monkeypatch.delenv("RT_AMD_TEST_FAIL_GPU_INDEX")
    r.upload(tris, nodes)
    info = r.scene_info()
    assert info["index_built_on_device"] == 1 and info["index_build_fell_back"] == 0
    r.close()


def test_coalesced_calls_give_the_same_images_and_launch_less_often():
    """RtOptions.coalesce: small calls issued while the GPU is busy are collected and launched
    together.  Whatever the caller does in between - reads, camera changes, resets, statistics -
    the result equals that of launching every call at once; a caller that holds the sum buffer
    (bound, or its address handed out) or a stream of its own gets one launch per call."""
    import torch
    Wc, Hc = 480, 270
    sc = _hall()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam_a = rt.camera_default(Wc, Hc)
    cam_b = rt.camera_from_angles((1.0, 3.0, 9.0), -100.0, -5.0, 40.0, Wc, Hc)
    ro = rt.rand_origins(123, 96)

    def scenario(r):
        out = []
        for k in range(40):                                 # forty calls of one frame, no waiting
            r.render(k + 1, ro[k:k + 1])
        out.append(r.read_mean())
        for k in range(40, 52, 3):                          # calls of three, then statistics
            r.render(k + 1, ro[k:k + 3])
        st = r.stats()
        out.append((st["frames"], st["rays"]))
        r.render(53, ro[52:53])
        r.set_camera(cam_b)                                 # frame 53 belongs to the old camera
        r.render(54, ro[53:54])
        r.render(56, ro[55:56])                             # not consecutive: another batch
        out.append(r.read_sum())
        r.reset()
        r.render(1, ro[:2])
        r.reset()                                           # frames not launched yet go with the reset
        r.render(1, ro[60:64])
        r.render(5, ro[64:65])
        out.append(r.read_mean())
        out.append(r.stats()["frames"])
        r.set_camera(cam_a)
        return out

    res = {}
    launches = {}
    for mode in ("each", "coalesce", "bound", "own_stream", "ptr_given"):
        r = rt.Renderer(Wc, Hc)
        r.upload(tris, nodes)
        r.set_spheres(sc.spheres())
        r.set_camera(cam_a)
        r.set_options(coalesce=(mode != "each"))
        keep = None
        if mode == "bound":
            keep = torch.zeros(Hc * Wc * 3, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            r.bind_sum_buffer(keep.data_ptr())
        elif mode == "own_stream":
            keep = torch.cuda.Stream()
            r.set_stream(keep.cuda_stream)
        elif mode == "ptr_given":
            assert r.sum_device_ptr() != 0
        res[mode] = scenario(r)
        r.reset()
        for k in range(48):
            r.render(k + 1, ro[k:k + 1])
        launches[mode] = r.kernel_time()[1]
        if mode == "bound":
            r.bind_sum_buffer(0)
        if mode == "own_stream":
            r.set_stream(0)
        r.close()
    for mode in ("coalesce", "bound", "own_stream", "ptr_given"):
        for a, b in zip(res["each"], res[mode]):
            assert _eq(a, b) if isinstance(a, np.ndarray) else a == b, mode
    assert res["each"][1][0] == 52 and res["each"][4] == 5
    assert launches["each"] == launches["bound"] == launches["own_stream"] == launches["ptr_given"] == 48
    assert launches["coalesce"] < 48


def test_async_read_shows_frame_n_while_frame_n_plus_one_is_traced():
    """rt_read_mean_async + rt_read_wait: the display loop of the reference (one frame, then show it)
    with the copy of frame N overlapping the tracing of frame N+1.  Every image read this way must be
    the running mean after exactly the frames requested before the read."""
    Wc, Hc, N = 640, 360, 24
    sc = _hall()
    r = rt.make_renderer(sc, Wc, Hc)
    ro = rt.rand_origins(321, N)
    want = []
    for k in range(N):                                      # one frame, wait, read: the plain loop
        r.render(k + 1, ro[k:k + 1])
        want.append(r.read_mean())
    for coalesce in (True, False):
        r.set_options(coalesce=coalesce)
        r.reset()
        bufs = [r.host_image(), r.host_image()]
        got = []
        for k in range(N):
            r.render(k + 1, ro[k:k + 1])
            if k > 0:
                r.read_wait()                               # the image of frame k (requested in the last round) has arrived
                got.append(bufs[(k - 1) % 2].copy())
            r.read_mean_async(bufs[k % 2])
        r.read_wait()
        got.append(bufs[(N - 1) % 2].copy())
        for k in range(N):
            assert _eq(got[k], want[k]), (coalesce, k)
        assert r.stats()["frames"] == N
    assert r.mean_device_ptr() != 0
    r.close()


def test_random_call_sequences_do_not_depend_on_overlap_or_coalescing():
    """Seeded random sequences of the calls a host makes - render 1..70 frames, read the mean or the
    sum, statistics, move the camera, change the spheres, switch pipeline or options, reset,
    restore, trace rays without allocation, read asynchronously - replayed on three contexts:
    everything on the context's stream (overlap off), overlapped launches, overlapped launches with
    call coalescing.  Every value a call returns must be the same on all three."""
    import torch
    Wc, Hc = 200, 120
    sc = _hall()
    tris, nodes = rt.bvh_build(sc.triangles())
    spheres = sc.spheres()
    cams = [rt.camera_default(Wc, Hc), rt.camera_from_angles((1.0, 3.0, 9.0), -100.0, -5.0, 40.0, Wc, Hc),
            rt.camera_from_angles((-4.0, 8.0, 2.0), -30.0, -20.0, 45.0, Wc, Hc)]
    ro_all = rt.rand_origins(99, 8192)
    o, d = scenes.random_rays(5000, 2, extent=10.0, axis_aligned_fraction=0.3)
    do, dd = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    torch.cuda.synchronize()

    def replay(seed, overlap, coalesce):
        rng = np.random.default_rng(seed)
        r = rt.Renderer(Wc, Hc)
        r.upload(tris, nodes)
        r.set_spheres(spheres)
        r.set_camera(cams[0])
        base = dict(overlap=overlap, coalesce=coalesce)
        opts = dict(base)
        r.set_options(**opts)
        dt = torch.zeros(o.shape[0], dtype=torch.float32, device="cuda")
        di = torch.zeros(o.shape[0], dtype=torch.int32, device="cuda")
        pinned = r.host_image()
        torch.cuda.synchronize()
        frame, log, saved = 0, [], None
        for step in range(60):
            op = rng.choice(["render1", "render1", "render1", "render_few", "render_many", "mean", "sum", "stats", "camera", "spheres",
                             "pipeline", "bounces", "reset", "restore", "trace", "async"])
            if op.startswith("render"):
                cnt = 1 if op == "render1" else (int(rng.integers(2, 9)) if op == "render_few" else int(rng.integers(20, 71)))
                for _ in range(int(rng.integers(1, 6)) if op == "render1" else 1):
                    r.render(frame + 1, ro_all[frame:frame + cnt])
                    frame += cnt
            elif op == "mean":
                log.append(("mean", r.read_mean()))
            elif op == "sum":
                log.append(("sum", r.read_sum()))
            elif op == "stats":
                st = r.stats()
                log.append(("stats", (st["frames"], st["rays"])))
            elif op == "camera":
                r.set_camera(cams[int(rng.integers(0, 3))])
            elif op == "spheres":
                r.set_spheres(spheres[:int(rng.integers(0, len(spheres) + 1))])
            elif op == "pipeline":
                opts = dict(base, pipeline=int(rng.choice([rt.PIPELINE_PERSISTENT, rt.PIPELINE_PERSISTENT, rt.PIPELINE_WAVEFRONT, rt.PIPELINE_MEGAKERNEL])),
                            max_bounces=opts.get("max_bounces", 4))
                r.set_options(**opts)
            elif op == "bounces":
                opts = dict(opts, max_bounces=int(rng.integers(1, 9)))
                r.set_options(**opts)
            elif op == "reset":
                r.reset()
                frame = 0
            elif op == "restore":
                if saved is None:
                    saved = (r.read_mean(), r.read_sum(), frame)
                else:
                    r.restore(saved[0], saved[1], saved[2])
                    frame = saved[2]
                    saved = None
            elif op == "trace":
                r.trace_rays_noalloc(do.data_ptr(), dd.data_ptr(), o.shape[0], dt.data_ptr(), di.data_ptr())
            elif op == "async":
                r.read_mean_async(pinned)
                if rng.integers(0, 2):
                    r.render(frame + 1, ro_all[frame:frame + 1])      # traced beside the copy, added after it
                    frame += 1
                r.read_wait()
                log.append(("async", pinned.copy()))
        log.append(("final", r.read_mean()))
        r.synchronize()
        log.append(("trace", (dt.cpu().numpy().copy(), di.cpu().numpy().copy())))
        r.close()
        return log

    for seed in range(1, 11):
        logs = [replay(seed, False, False), replay(seed, True, False), replay(seed, True, True)]
        assert len(logs[0]) == len(logs[1]) == len(logs[2])
        for a, b, c in zip(*logs):
            assert a[0] == b[0] == c[0]
            for x in (b, c):
                if isinstance(a[1], np.ndarray):
                    assert _eq(a[1], x[1]), (seed, a[0])
                elif a[0] == "trace":
                    assert np.array_equal(a[1][0], x[1][0]) and np.array_equal(a[1][1], x[1][1]), seed
                else:
                    assert a[1] == x[1], (seed, a[0])

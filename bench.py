#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one batch: `--spp-per-step` frames (samples per pixel)
of the workload at its full resolution.  For N > 1 the driver launches one process per GPU with
torch.distributed.run; every rank renders its own block of frame indices (weak scaling), and
the per-rank fp32 sums are reduced to rank 0 by one RCCL reduce inside the timed region.
With `--total-spp S` the job is fixed instead (strong scaling): S frames in all, S / (N * K) per
rank and step.

Prints ONE JSON line on rank 0.  `value` is Mrays/s over all ranks: rays actually traced
(camera ray + one per scatter; rays that already missed are not re-traced) divided by the
max-over-ranks wall time of the K timed steps.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (BASELINE.json configs index, description, width, height, bounces, scene builder args)
    "dragon1m": (2, "procedural 1.31M-triangle displaced icosphere in the reference room, 1080p, 4 bounces",
                 1920, 1080, 4, dict(kind=0, level=8, copies=1)),
    "demo": (1, "reference demo scene (32 triangles, 2 spheres), 1080p, 4 bounces",
             1920, 1080, 4, None),
    "sponza": (3, "room x3 with 32 tessellated columns (262k triangles, diffuse/mirror/glass), 4K, 4 bounces",
               3840, 2160, 4, dict(kind=1, level=64, copies=1)),
    "stress10m": (4, "8 displaced icospheres (10.5M triangles) in the reference room, 1080p, 8 bounces",
                  1920, 1080, 8, dict(kind=0, level=8, copies=8)),
}
SEED = 20221003
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak
VALU_PEAK_TFLOPS = 157.3     # MI355X fp32 vector peak


def build_scene(rt, spec):
    sc = rt.Scene()
    if spec is None:
        sc.push_demo()
    else:
        sc.push_synthetic(spec["kind"], spec["level"], spec["copies"], 12345)
    t0 = time.time()
    tris, nodes = rt.bvh_build(sc.triangles())
    return sc, tris, nodes, time.time() - t0


def cpu_baseline(rt, tris, nodes, spheres, cam, W, H, bounces, ro, budget_s=15.0):
    """The CPU oracle (a port of the reference shader, OpenMP over rows) timed on a band of rows
    of frame 1 of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    nproc = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = nproc
    # the share of the host that belongs to one GPU (16 cores on the 8-GPU nodes this runs on)
    threads = min(16, usable)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    osc = ol.scene_from_build(tris, nodes, W, H, rt.sobol_default_table(), spheres, max_bounces=bounces)
    # whole frames 1, 2, ... until about budget_s of CPU time has been spent (at least one frame)
    rays, frames, t0 = 0, 0, time.time()
    while frames < len(ro) and (frames == 0 or time.time() - t0 < budget_s):
        _, _, st = osc.render_frame(cam, frames + 1, float(ro[frames]), want_stats=True)
        rays += st["rays"]
        frames += 1
    dt = time.time() - t0
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gl_probe
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "host_nproc": nproc, "host_cpus_usable": usable,
            "sample": f"frames 1..{frames} of the same workload at full resolution ({rays} rays, {dt:.1f} s), "
                      "oracle/rt_oracle.c with OpenMP over rows; no hitMin culling, re-traces excluded from the count",
            # the context baseline the reference's shader under Mesa llvmpipe would give: only possible where a
            # GL context can be created without a display
            "llvmpipe": gl_probe.probe()}


def frames_per_launch(frames_per_call, cap):
    """rt_render_frames cuts a call into launches of equal size, none above the most a launch holds."""
    n = -(-frames_per_call // cap)
    return -(-frames_per_call // n)


def call_granularity(r, ro, opts, frames=64):
    """The drop-in call pattern: the reference's main loop (main.cpp:124-197) renders ONE frame per
    iteration, rt_render_frames(ctx, LoopNum, 1, &randOrigin), and displays the image after every
    frame.  Times `frames` frames issued as calls of 1 and of 8 frames: pipelined (no host
    synchronisation between calls) with RtOptions.coalesce on (the default: calls issued while the
    GPU is busy are launched together) and off (one launch per call), and one frame per call
    followed by rt_synchronize, by rt_read_mean (a blocking copy of the W*H*3 floats to pageable
    host memory) or by rt_read_wait + rt_read_mean_async into page-locked memory (frame N is copied
    and shown while frame N+1 is traced).  Runs on the library's own sum buffer: with a caller-owned buffer bound, as in the
    main measurement, calls are never coalesced."""
    out = []
    frames = min(frames, len(ro) // 8 * 8)
    if frames < 8:
        return out
    r.bind_sum_buffer(0)
    pinned = [r.host_image(), r.host_image()]
    for per_call, after, coalesce in ((1, None, True), (1, None, False), (8, None, True), (8, None, False),
                                      (1, "synchronize", True), (1, "read_mean", True), (1, "read_mean_async", True)):
        r.set_options(count_work=False, coalesce=coalesce, **opts)
        for timed in (False, True):                         # one untimed pass first
            r.reset()
            r.synchronize()
            t0 = time.perf_counter()
            for first in range(1, frames + 1, per_call):
                r.render(first, ro[first - 1:first - 1 + per_call])
                if after == "read_mean":
                    r.read_mean()
                elif after == "synchronize":
                    r.synchronize()
                elif after == "read_mean_async":
                    # show frame N-1 (its copy has had the tracing of frame N to arrive), ask for frame N
                    r.read_wait()
                    r.read_mean_async(pinned[(first - 1) % 2])
            r.synchronize()
            dt = time.perf_counter() - t0
        rays = r.stats()["rays"]
        _, launches = r.kernel_time()
        out.append({"frames_per_call": per_call, "after_each_call": after or "nothing (pipelined)", "coalesce": coalesce,
                    "calls": frames // per_call, "launches": launches, "value": round(rays / dt / 1e6, 2), "unit": "Mrays/s",
                    "ms_per_call": round(dt / (frames // per_call) * 1e3, 4)})
    r.set_options(count_work=False, **opts)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default 10; 8 for sponza, whose 8 x 64 frames are one GPU's "
                    "share of the 4096 samples that configuration splits over 8 GPUs)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dragon1m", choices=sorted(WORKLOADS))
    ap.add_argument("--spp-per-step", type=int, default=0,
                    help="frames per step (default 64; where a launch holds fewer, e.g. 21 at 4K, the call is cut "
                         "into launches of equal size)")
    ap.add_argument("--pipeline", default="persistent", choices=["wavefront", "megakernel", "persistent"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cull", action="store_true")
    ap.add_argument("--no-accel", action="store_true", help="traverse the reference's tree in the reference's order")
    ap.add_argument("--resolution", default=None, help="WxH override for sensitivity runs (the named workload's resolution is the default)")
    ap.add_argument("--gpu-index", action="store_true", help="build the search index on the GPU at upload whatever the scene size")
    ap.add_argument("--host-index", action="store_true", help="build the search index on the host whatever the scene size "
                    "(default: RtOptions.index_build = auto, on the GPU from 16 million triangles on)")
    ap.add_argument("--no-overlap", action="store_true", help="keep every launch on the context's stream (RtOptions.overlap = 0)")
    ap.add_argument("--total-spp", type=int, default=0,
                    help="strong scaling: this many frames in the whole job, split evenly over ranks and steps "
                         "(overrides --spp-per-step; must be divisible by gpus * steps)")
    ap.add_argument("--no-call-granularity", action="store_true",
                    help="skip the extra measurement of 1 and 8 frames per rt_render_frames call")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a one-GPU box: every rank uses GPU 0 and the reduce goes through gloo")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world

    if world > 1 and "RT_AMD_HOST_THREADS" not in os.environ:
        # every rank builds its own tree and search index on the host before the timed region:
        # share the host's cores between the ranks instead of 16 threads each
        os.environ["RT_AMD_HOST_THREADS"] = str(max(2, min(16, (os.cpu_count() or 16) // world)))
    import torch
    import raytracing_amd as rt
    from raytracing_amd import partition

    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg_idx, desc, W, H, bounces, spec = WORKLOADS[args.workload]
    if args.resolution:
        W, H = (int(v) for v in args.resolution.lower().split("x"))
        desc += f" [resolution overridden to {W}x{H}]"
    sc, tris, nodes, bvh_s = build_scene(rt, spec)
    cam = rt.camera_default(W, H)
    pipeline = {"wavefront": rt.PIPELINE_WAVEFRONT, "megakernel": rt.PIPELINE_MEGAKERNEL,
                "persistent": rt.PIPELINE_PERSISTENT}[args.pipeline]

    opts = dict(max_bounces=bounces, pipeline=pipeline, cull=not args.no_cull, accel=not args.no_accel,
                overlap=not args.no_overlap, index_build=rt.INDEX_BUILD_GPU if args.gpu_index else (rt.INDEX_BUILD_HOST if args.host_index else rt.INDEX_BUILD_AUTO))
    r = rt.Renderer(W, H, local_rank)
    r.set_options(count_work=False, **opts)
    t_up = time.time()
    r.upload(tris, nodes)
    r.synchronize()
    upload_s = time.time() - t_up
    r.set_spheres(sc.spheres())
    r.set_camera(cam)
    sum_t = torch.zeros(H * W * 3, dtype=torch.float32, device=f"cuda:{local_rank}")
    r.bind_sum_buffer(sum_t.data_ptr())
    info = r.scene_info()

    F, K, Wm = args.spp_per_step, args.steps, args.warmup
    if K <= 0:
        # sponza is stated at 4096 samples over 8 GPUs: 512 per GPU = 8 steps of 64; the others take 10 steps
        K = 8 if args.workload == "sponza" else 10
    if F <= 0:
        F = 64
    if args.total_spp:
        if args.total_spp % (world * K):
            raise SystemExit("--total-spp must be divisible by gpus * steps")
        F = args.total_spp // (world * K)
    total_frames = (K + Wm) * world * F
    ro = partition.rand_origin_stream(SEED, total_frames)

    def run_step(step):
        first, cnt = partition.frame_block(rank, world, step, F)
        r.render(first, ro[first - 1:first - 1 + cnt])

    for s in range(Wm):
        run_step(s)
    r.synchronize()
    # warmup frames are not part of the measured image or counters
    r.reset()
    sum_t.zero_()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        run_step(Wm + s)
    if dist:
        if args.rehearse_on_one_gpu:
            host = sum_t.cpu()
            partition.reduce_sum_to_root(host, dist)
            sum_t.copy_(host)
        else:
            partition.reduce_sum_to_root(sum_t, dist)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    stats = r.stats()
    kernel_ms, launches = r.kernel_time()
    rays_local = stats["rays"]
    if dist:
        t = torch.tensor([elapsed, float(rays_local)], dtype=torch.float64,
                         device="cpu" if args.rehearse_on_one_gpu else f"cuda:{local_rank}")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays_total = float(t[1])
    else:
        rays_total = float(rays_local)

    result = None
    if rank == 0:
        mean_img = (sum_t / float(K * world * F)).cpu().numpy()
        value = rays_total / elapsed / 1e6
        # instrumented re-run of this rank's timed frames: identical traversal, counts the work
        r.set_options(count_work=True, **opts)
        r.reset()
        ksteps = min(K, 2)
        for s in range(ksteps):
            run_step(Wm + s)
        r.synchronize()
        w = r.stats()
        # Algorithmic bytes per ray segment: boxes tested x bytes per box in the layout the kernel
        # reads (16 B in the 4-wide quantised index, 32 B in the reference-order binary nodes),
        # + 36 B per triangle tested + 36 B of material per final hit.
        rays_w = max(w["rays"], 1)
        node_b = 16.0 if (not args.no_accel and pipeline != rt.PIPELINE_MEGAKERNEL) else 32.0
        bytes_per_ray = (node_b * w["node_tests"] + 36.0 * w["tri_tests"] + 36.0 * w["tri_hits"]) / rays_w
        achieved = bytes_per_ray * rays_local / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        binary_bpr = (32.0 * w["node_tests"] + 36.0 * w["tri_tests"] + 36.0 * w["tri_hits"]) / rays_w
        # fp32 vector work: 21 ops per box test, 95 per triangle test, 150 per shaded ray
        flops_per_ray = (21.0 * w["node_tests"] + 95.0 * w["tri_tests"]) / rays_w + 150.0
        valu_tflops = flops_per_ray * rays_local / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        # Counters of the dominant kernel from the committed PMC passes of the same workload
        # (profiles/r0N_pmc_<workload>.json, made by tools/pmc_to_json.py; the newest round that has one):
        # traffic beyond L2, vector issue rate and lane utilisation.  They are NOT measured in this run
        # (counter collection needs rocprofv3 around the process) and are only used when they
        # describe this configuration; every such field sits in `pmc` with from_committed_profile.
        fpl = frames_per_launch(F, info["frames_per_launch"])
        traffic = None
        pmc = None
        pmc_path = None
        for rnd in ("r03", "r02"):
            cand_path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{args.workload}.json")
            try:
                cand = json.load(open(cand_path))
                if (cand["workload"] == args.workload and not args.no_accel and args.pipeline == cand["pipeline"]
                        and fpl == cand["frames_per_dispatch"] and not args.resolution):
                    pmc, pmc_path = cand, cand_path
                    break
            except (OSError, KeyError, ValueError):
                continue
        traffic_lo = traffic_hi = None
        if pmc:
            # FETCH_SIZE counts requests of 32 or 64 bytes in one unit; x2 is right for wide coalesced
            # reads (all 64-byte requests), x1 for 32-byte ones.  The divergent 16-byte gathers of a
            # traversal lie between the two, so the traffic is a range; `traffic` is its upper end.
            traffic_lo = round((pmc["fetch_size_kb_per_dispatch"] + pmc["write_size_kb_per_dispatch"]) * 1024.0)
            traffic_hi = round((pmc["fetch_size_kb_per_dispatch"] * pmc["fetch_correction"] + pmc["write_size_kb_per_dispatch"]) * 1024.0)
            traffic = traffic_hi
        launch_s = kernel_ms * 1e-3 / max(launches, 1)
        dominant = {rt.PIPELINE_WAVEFRONT: "k_extend" if args.no_accel else "k_extend_fast",
                    rt.PIPELINE_MEGAKERNEL: "k_megakernel", rt.PIPELINE_PERSISTENT: "k_paths"}[pipeline]
        result = {
            "metric": "Mrays/sec (paths x bounces) at 1080p" if H == 1080 else "Mrays/sec (paths x bounces)",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(elapsed / K * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if args.total_spp else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc} (BASELINE.json configs[{cfg_idx}])",
                       "triangles": int(tris.shape[0]), "bvh_nodes": int(nodes.shape[0]),
                       "resolution": f"{W}x{H}", "spp_per_step": F, "frames_per_call": F,
                       "frames_per_launch": fpl, "total_spp": K * world * F, "max_bounces": bounces,
                       "pipeline": args.pipeline, "cull": not args.no_cull, "search_index": not args.no_accel,
                       "parallelism": f"spp-partition x{world}, one RCCL sum-reduce" if world > 1 else "single GPU",
                       "rays_per_step_per_gpu": round(rays_local / K, 1), "bvh_build_s": round(bvh_s, 2),
                       "scene_upload_s": round(upload_s, 3), "index_build": "gpu" if info["index_built_on_device"] else "host",
                       "overlap": not args.no_overlap,
                       "image_mean": float(mean_img.mean())},
            # The scene (index + triangles) is served by L1 / L2 / Infinity Cache, so the roof that
            # binds is the fp32 vector unit, not HBM: `achieved` is the algorithmic fp32 work (21 flop
            # per box tested, 95 per triangle tested, 150 per ray shaded) over the kernel's time.
            "roofline": {"bound": "valu", "kernel": dominant, "achieved": round(valu_tflops, 3), "peak": VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(valu_tflops / VALU_PEAK_TFLOPS, 4),
                         # the same algorithmic work over the wall time of the timed region (all kernels, launch
                         # gaps and the overlap of consecutive launches included)
                         "frac_on_wall": round(flops_per_ray * rays_local / elapsed / 1e12 / VALU_PEAK_TFLOPS, 4),
                         "traffic": traffic,
                         "traffic_unit": "bytes per launch beyond L2, upper end of traffic_range (includes Infinity Cache hits)",
                         "flops_per_ray": round(flops_per_ray, 1),
                         "kernel_ms_per_launch": round(kernel_ms / max(launches, 1), 4), "launches": launches,
                         "kernel_share_of_wall": round(kernel_ms * 1e-3 / elapsed, 3),
                         "waves_per_simd": info["paths_waves_per_simd"] if pipeline == rt.PIPELINE_PERSISTENT else None,
                         "nodes_per_ray": round(w["node_tests"] / rays_w, 2),
                         "tris_per_ray": round(w["tri_tests"] / rays_w, 2),
                         # memory side, for context: algorithmic bytes (what the caches serve) and HBM-side traffic
                         "bytes_per_ray": round(bytes_per_ray, 1), "bytes_per_box": node_b,
                         "bytes_per_ray_at_32B_per_box": round(binary_bpr, 1),
                         "algorithmic_bytes_per_launch": round(bytes_per_ray * rays_local / max(launches, 1)),
                         "cache_served_tbps": round(achieved / 1e3, 3),
                         "timing_note": "kernel_ms_per_launch is the HIP-event time of each launch on its own stream; with overlap the end of "
                                        "one launch and the start of the next share the GPU, so the sum can exceed the wall time "
                                        "(kernel_share_of_wall > 1); frac_on_wall is the self-consistent figure",
                         "pmc": ({"from_committed_profile": True, "source": os.path.relpath(pmc_path, ROOT),
                                  "traffic_range": [traffic_lo, traffic_hi],
                                  "traffic_range_note": "PMC FETCH_SIZE x1 .. x2 + WRITE_SIZE per launch: x2 holds for coalesced 64-byte "
                                                        "requests, divergent 16-byte gathers lie in between",
                                  "hbm_frac_range": [round(traffic_lo / launch_s / (HBM_PEAK_GBS * 1e9), 4),
                                                     round(traffic_hi / launch_s / (HBM_PEAK_GBS * 1e9), 4)] if launch_s > 0 else None,
                                  "hbm_note": "traffic beyond L2 per launch / this run's time per launch / 8 TB/s: an upper bound on the "
                                              "HBM share, Infinity Cache hits are counted in it",
                                  "valu_issue_frac": pmc.get("valu_issue_frac_of_guide_peak"),
                                  "valu_lane_utilisation": pmc.get("valu_lane_utilisation"),
                                  "l2_hit_rate": pmc.get("l2_hit_rate")} if pmc else None)},
        }
        if world == 1 and not args.no_call_granularity and pipeline == rt.PIPELINE_PERSISTENT:
            result["call_granularity"] = call_granularity(r, ro, opts)
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(rt, tris, nodes, sc.spheres(), cam, W, H, bounces, ro)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    r.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()

"""Multi-rank path on CPU: world_size 2 over gloo.  Each rank produces the samples of its own
block of frame indices (the oracle stands in for the GPU renderer here), sums them, and one
reduce to rank 0 must give the same image as a single process rendering every frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol
import raytracing_amd as rt
import scenes
from raytracing_amd import partition

W, H, STEPS, F, SEED = 24, 16, 3, 2, 99


def _render_sum(rank, world):
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    osc = ol.scene_from_build(tris, nodes, W, H, rt.sobol_default_table(), sc.spheres())
    ro = partition.rand_origin_stream(SEED, STEPS * world * F)
    total = np.zeros((H, W, 3), dtype=np.float32)
    frames = []
    for step in range(STEPS):
        first, cnt = partition.frame_block(rank, world, step, F)
        for k in range(cnt):
            _, cur, _ = osc.render_frame(cam, first + k, float(ro[first + k - 1]))
            total += cur
            frames.append(first + k)
    return total, frames


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total, frames = _render_sum(rank, world)
    t = torch.from_numpy(total.reshape(-1).copy())
    partition.reduce_sum_to_root(t, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, frames)
    if rank == 0:
        np.save(os.path.join(out_dir, "sum.npy"), t.numpy().reshape(H, W, 3))
        np.save(os.path.join(out_dir, "frames.npy"), np.array(sorted(sum(gathered, []))))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_blocks_cover_every_frame_once():
    for world in (1, 2, 4, 8):
        seen = []
        for step in range(5):
            for rank in range(world):
                first, cnt = partition.frame_block(rank, world, step, 3)
                seen += list(range(first, first + cnt))
        assert sorted(seen) == list(range(1, 5 * world * 3 + 1))


def test_two_ranks_reduce_to_the_single_process_image(tmp_path):
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sum.npy")
    frames = np.load(tmp_path / "frames.npy")
    assert frames.tolist() == list(range(1, STEPS * world * F + 1))
    # single process: every frame 1..12 with the same randOrigin stream
    sc = scenes.demo()
    tris, nodes = rt.bvh_build(sc.triangles())
    cam = rt.camera_default(W, H)
    osc = ol.scene_from_build(tris, nodes, W, H, rt.sobol_default_table(), sc.spheres())
    ro = partition.rand_origin_stream(SEED, STEPS * world * F)
    want = np.zeros((H, W, 3), dtype=np.float64)
    for n in range(1, len(ro) + 1):
        _, cur, _ = osc.render_frame(cam, n, float(ro[n - 1]))
        want += cur
    # fp32 sums in a different order: equal up to rounding
    assert np.allclose(got, want, rtol=1e-6, atol=1e-5)
    mean = partition.mean_from_sum(got, len(ro))
    assert np.sqrt(np.mean((mean - want / len(ro)) ** 2)) < 1e-6

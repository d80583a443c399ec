"""The multi-GPU path with the real renderer: two ranks (both on GPU 0, the reduce through gloo
since one box has one GPU) each drive a Renderer whose sums go into a torch tensor bound with
rt_bind_sum_buffer, render their own blocks of frame indices, and ONE sum-reduce to rank 0 must
reproduce the image of a single process that renders every frame."""
import os
import socket

import numpy as np
import pytest
import torch  # noqa: F401

import raytracing_amd as rt
from raytracing_amd import partition

pytestmark = pytest.mark.gpu

W, H, STEPS, F, SEED = 160, 96, 3, 5, 4242


def _scene():
    sc = rt.Scene()
    sc.push_synthetic(1, 6, 1, 12345)                       # column hall: diffuse / mirror / glass + two spheres
    return sc


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    sc = _scene()
    r = rt.make_renderer(sc, W, H, 0)
    sum_t = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda:0")
    r.bind_sum_buffer(sum_t.data_ptr())
    ro = partition.rand_origin_stream(SEED, STEPS * world * F)
    frames = []
    for step in range(STEPS):
        first, cnt = partition.frame_block(rank, world, step, F)
        r.render(first, ro[first - 1:first - 1 + cnt])
        frames += list(range(first, first + cnt))
    r.synchronize()
    rays = r.stats()["rays"]
    host = sum_t.cpu()
    partition.reduce_sum_to_root(host, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (frames, rays))
    if rank == 0:
        np.save(os.path.join(out_dir, "sum.npy"), host.numpy().reshape(H, W, 3))
        np.save(os.path.join(out_dir, "frames.npy"), np.array(sorted(sum((g[0] for g in gathered), []))))
        np.save(os.path.join(out_dir, "rays.npy"), np.array([g[1] for g in gathered]))
    r.bind_sum_buffer(0)
    r.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_bound_sum_buffers_reduce_to_the_single_process_image(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.get_context("spawn")
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sum.npy")
    total = STEPS * world * F
    assert np.load(tmp_path / "frames.npy").tolist() == list(range(1, total + 1))
    # one process, every frame, same randOrigin stream
    r = rt.make_renderer(_scene(), W, H, 0)
    ro = partition.rand_origin_stream(SEED, total)
    r.render(1, ro)
    want_sum, want_mean = r.read_sum(), r.read_mean()
    assert r.stats()["rays"] == int(np.load(tmp_path / "rays.npy").sum())
    r.close()
    # fp32 sums added in a different order: equal up to rounding
    assert np.array_equal(np.isnan(got), np.isnan(want_sum))
    ok = ~np.isnan(want_sum)
    assert np.allclose(got[ok], want_sum[ok], rtol=1e-5, atol=1e-4)
    mean = partition.mean_from_sum(got, total)
    one = partition.mean_from_sum(want_sum, total)
    assert float(np.sqrt(np.mean((mean[ok].astype(np.float64) - one[ok].astype(np.float64)) ** 2))) < 1e-6
    # and against the running mean the single process kept (frag:697 recurrence, other rounding)
    assert float(np.sqrt(np.mean((mean[ok].astype(np.float64) - want_mean[ok].astype(np.float64)) ** 2))) < 1e-5
    assert np.nan_to_num(want_mean).max() > 0

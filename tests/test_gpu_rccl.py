"""The pieces of the multi-GPU run that one GPU can exercise with the real backend:

  * one HIP runtime in the process whatever the import order: raytracing_amd loaded BEFORE torch
    (the order that used to leave the process with two runtimes, the second blind to the GPU);
  * the `nccl` backend (RCCL) initialised with this library in the process, at world size 1;
  * the sum buffer the library's kernels write is a torch tensor (rt_bind_sum_buffer), and
    partition.reduce_sum_to_root runs dist.reduce on that device tensor, on torch's stream, while
    the renderer worked on its own streams.

Each case runs in a process of its own, so that the load order is the one under test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, {root!r})
ORDER = {order!r}
if ORDER == "library_first":
    import raytracing_amd as rt
    import torch
else:
    import torch
    import raytracing_amd as rt
import numpy as np
import torch.distributed as dist
from raytracing_amd import partition

# one runtime: every mapped libamdhip64 is the same file
maps = sorted({{line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line}})
assert len(maps) == 1, maps
assert torch.cuda.is_available() and torch.cuda.device_count() >= 1

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", {port!r})
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

W, H, F = 160, 96, 12
sc = rt.Scene()
sc.push_synthetic(1, 6, 1, 12345)
r = rt.make_renderer(sc, W, H, 0)
ro = partition.rand_origin_stream(4242, 2 * F)
r.render(1, ro[:F]); r.render(F + 1, ro[F:])
want = r.read_sum()

sum_t = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
r.bind_sum_buffer(sum_t.data_ptr())
r.reset()
first, cnt = partition.frame_block(0, 1, 0, F)
r.render(first, ro[first - 1:first - 1 + cnt])
first, cnt = partition.frame_block(0, 1, 1, F)
r.render(first, ro[first - 1:first - 1 + cnt])
r.synchronize()                       # the renderer's streams are not torch's: hand over explicitly, as bench.py does
partition.reduce_sum_to_root(sum_t, dist)
torch.cuda.synchronize()
got = sum_t.cpu().numpy().reshape(H, W, 3)
assert np.array_equal(got, want, equal_nan=True)
assert np.array_equal(r.read_sum(), want, equal_nan=True)       # rt_read_sum reads the bound buffer
assert np.nan_to_num(want).max() > 0

# the reduce as a stream-ordered step of the context's own stream: the library renders on the
# stream torch (and RCCL) works on, no host synchronisation in between
s = torch.cuda.Stream()
r.set_stream(s.cuda_stream)
sum_t.zero_(); torch.cuda.synchronize()
r.reset()
with torch.cuda.stream(s):
    r.render(1, ro[:F])
    dist.all_reduce(sum_t)            # world size 1: the sum of one
    r.render(F + 1, ro[F:])
    dist.all_reduce(sum_t)
s.synchronize()
assert np.array_equal(sum_t.cpu().numpy().reshape(H, W, 3), want, equal_nan=True)
r.bind_sum_buffer(0); r.set_stream(0); r.close()
dist.barrier()
dist.destroy_process_group()
print("RCCL_CHILD_OK", ORDER, maps[0])
"""


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


@pytest.mark.parametrize("order", ["library_first", "torch_first"])
def test_nccl_backend_at_world_size_one_with_a_bound_sum_buffer(order, tmp_path):
    script = tmp_path / "child.py"
    script.write_text(CHILD.format(root=ROOT, order=order, port=_free_port()))
    env = dict(os.environ)
    env.pop("RT_AMD_HIP_RUNTIME", None)
    out = subprocess.run([sys.executable, "-u", str(script)], env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-6000:]
    assert "RCCL_CHILD_OK " + order in out.stdout

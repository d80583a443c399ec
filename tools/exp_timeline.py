#!/usr/bin/env python3
"""Experiment: per-wave timeline of one path-kernel launch (needs a library built with -DRT_EXP_TIMELINE):
   RT_AMD_LIBRARY=expl/librt_tl.so python3 tools/exp_timeline.py [frames]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sc = rt.Scene(); sc.push_synthetic(0, 8, 1, 12345)
r = rt.make_renderer(sc, 1920, 1080, 0, max_bounces=4, pipeline=2, overlap=False)
ro = rt.rand_origins(20221003, 8 + frames)
r.render(1, ro[:8]); r.synchronize(); r.reset()
r.render(9, ro[8:]); r.synchronize()
ms, n = r.kernel_time()
waves = 256 * int(os.environ.get("WAVES_PER_SIMD", "8")) * 4
buf = np.zeros((waves, 10), dtype=np.uint64)
rt.lib.rt_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert rt.lib.rt_debug_timeline(r._h, buf.ctypes.data_as(C.c_void_p), waves) == 0
retr = (buf[:, 2] >> np.uint64(32)).astype(np.int64)          # rays of the wave traced again in reference order
buf[:, 2] &= np.uint64(0xFFFFFFFF)
st = r.stats()
print(f"rays traced again: {st['retraced']} (ties {st['node_tests']}, unusable rays {st['tri_tests']}, reference does not reach the triangle {st['tri_hits']})")
t0 = buf[:, 0].astype(np.float64); t1 = buf[:, 1].astype(np.float64); ex = buf[:, 4].astype(np.float64)
base = t0.min()
tick = 1e-2     # wall_clock64 ticks at 100 MHz -> microseconds
print(f"kernel {ms / n:.3f} ms; waves {waves}; rays {int(buf[:, 2].sum())}")
print(f"wave start  : min 0, median {np.median(t0 - base) * tick:.0f}, max {(t0.max() - base) * tick:.0f} us")
print(f"work ran out: min {(ex[ex > 0].min() - base) * tick:.0f}, median {np.median(ex[ex > 0] - base) * tick:.0f}, max {(ex.max() - base) * tick:.0f} us")
print(f"wave end    : min {(t1.min() - base) * tick:.0f}, median {np.median(t1 - base) * tick:.0f}, p90 {np.percentile(t1 - base, 90) * tick:.0f}, p99 {np.percentile(t1 - base, 99) * tick:.0f}, max {(t1.max() - base) * tick:.0f} us")
dur = (t1 - t0) * tick
print(f"wave life   : median {np.median(dur):.0f}, max {dur.max():.0f} us; iterations median {np.median(buf[:, 3]):.0f}, max {buf[:, 3].max()}; rays per wave median {np.median(buf[:, 2]):.0f}, max {buf[:, 2].max()}")
shades = (buf[:, 3] >> np.uint64(32)).astype(np.float64)
buf[:, 3] &= np.uint64(0xFFFFFFFF)
print(f"shade passes per wave: median {np.median(shades):.0f}, max {shades.max():.0f}; rays per pass {buf[:, 2].sum() / max(shades.sum(), 1):.1f}")
print(f"iteration time: median {np.median(dur / np.maximum(buf[:, 3], 1)):.2f} us")
hist, edges = np.histogram((t1 - base) * tick, bins=20)
for h, e in zip(hist, edges):
    print(f"  ends at {e:7.0f} us: {h}")
tail = ((t1 - ex) * tick)[ex > 0]
print(f"after work ran out a wave lives: median {np.median(tail):.0f}, p90 {np.percentile(tail, 90):.0f}, max {tail.max():.0f} us")
endt = (t1 - base) * tick
order = np.argsort(endt)[::-1][:12]
print("last waves to end (us, rays traced again in the wave):", [(int(endt[i]), int(retr[i])) for i in order])
print(f"waves with a re-trace: {int((retr > 0).sum())}; their end: median {np.median(endt[retr > 0]) if (retr > 0).any() else 0:.0f} us; others: p99 {np.percentile(endt[retr == 0], 99):.0f}, max {endt[retr == 0].max():.0f} us")
# shader-clock cycles inside the three blocks (waits for memory included); the rest is the head of the loop
life = (t1 - t0) * tick                      # us
go, node, test, nwait = (buf[:, k].astype(np.float64) for k in (5, 6, 7, 8))
tot = go + node + test
print(f"cycles in blocks per us of wave life (median): {np.median(tot / np.maximum(life, 1e-9)):.0f} (the shader clock in MHz if the head of the loop were free)")
print(f"share of the block cycles: finish {go.sum() / tot.sum():.3f}, node {node.sum() / tot.sum():.3f} (of it waiting for the records {nwait.sum() / tot.sum():.3f}), triangle {test.sum() / tot.sum():.3f}")
it = buf[:, 3].astype(np.float64)
print(f"cycles per iteration: node block {node.sum() / it.sum():.0f} (wait {nwait.sum() / it.sum():.0f}); triangle block {test.sum() / it.sum():.0f}; finish block {go.sum() / it.sum():.0f}")

"""The library's own search index (host build, 4-wide nodes with 8-bit boxes), checked without a GPU.

The search through it must be conservative - it may visit boxes it need not, never miss one - so
the decoded box of every slot has to contain every triangle below it: the triangle's own min/max
box for a leaf slot, the boxes of all triangles of the subtree for an interior slot.  Every triangle hangs in
exactly one leaf slot, and the deferred-node bound the builder reports covers the depth found."""
import numpy as np
import pytest

import raytracing_amd as rt
import scenes

LEAF = 0x80000000
NONE = 0xFFFFFFFF


def _decode(nodes):
    """lo, hi: (n, 4, 3) decoded slot boxes; ref: (n, 4)."""
    sc = np.stack([nodes["scaleX"], nodes["scaleY"], nodes["scaleZ"]], axis=1).astype(np.float64)        # (n, 3)
    org = nodes["origin"].astype(np.float64)                                                            # (n, 3)
    sh = (8 * np.arange(4, dtype=np.uint32))[None, :, None]
    qlo = (nodes["qlo"][:, None, :] >> sh) & 255
    qhi = (nodes["qhi"][:, None, :] >> sh) & 255
    lo = org[:, None, :] + qlo * sc[:, None, :]
    hi = org[:, None, :] + qhi * sc[:, None, :]
    return lo, hi, nodes["ref"]


def _check(tris, collapse_dp, monkeypatch):
    if collapse_dp:
        monkeypatch.setenv("RT_AMD_COLLAPSE_DP", "1")
    else:
        monkeypatch.delenv("RT_AMD_COLLAPSE_DP", raising=False)
    nodes, info = rt.index_build_host(tris)
    n = tris.shape[0]
    assert info["nodes"] == nodes.shape[0] >= 1
    lo, hi, ref = _decode(nodes)
    used = ref != NONE
    leaf = used & ((ref & LEAF) != 0)
    inner = used & ~leaf
    # empty slots: inverted boxes
    assert np.all((nodes["qlo"][:, None, :] >> (8 * np.arange(4, dtype=np.uint32))[None, :, None] & 255)[~used] == 255)
    # every triangle in exactly one leaf slot
    ids = (ref[leaf] & 0x7FFFFFFF).astype(np.int64)
    assert ids.shape[0] == n and np.array_equal(np.sort(ids), np.arange(n))
    # leaf slots contain their triangle's min/max box
    v = tris["v"].astype(np.float64)
    tlo, thi = v.min(axis=1), v.max(axis=1)
    assert np.all(lo[leaf] <= tlo[ids]) and np.all(hi[leaf] >= thi[ids])
    # interior slots contain every used slot box of the child node; children come after their parents once each
    pn, ps = np.nonzero(inner)
    child = ref[pn, ps].astype(np.int64)
    assert np.all(child > pn) and np.all(child < nodes.shape[0])
    assert np.array_equal(np.sort(child), np.arange(1, nodes.shape[0]))
    # exact bounds of the triangles below every node, children before parents (children have larger indices)
    blo = np.full((nodes.shape[0], 3), np.inf)
    bhi = np.full((nodes.shape[0], 3), -np.inf)
    ln, ls = np.nonzero(leaf)
    np.minimum.at(blo, ln, tlo[(ref[ln, ls] & 0x7FFFFFFF).astype(np.int64)])
    np.maximum.at(bhi, ln, thi[(ref[ln, ls] & 0x7FFFFFFF).astype(np.int64)])
    for p, c in sorted(zip(pn.tolist(), child.tolist()), reverse=True):
        blo[p] = np.minimum(blo[p], blo[c])
        bhi[p] = np.maximum(bhi[p], bhi[c])
    # an interior slot's box contains every triangle below it (the child's own slot boxes are rounded
    # outwards on the child's grid and may stick out of it: they only have to contain the triangles)
    assert np.all(lo[pn, ps] <= blo[child]) and np.all(hi[pn, ps] >= bhi[child])
    # depth: levels of wide nodes from the root
    depth = np.zeros(nodes.shape[0], dtype=np.int64)
    depth[0] = 1
    order = np.argsort(pn, kind="stable")
    for p, c in zip(pn[order], child[order]):
        depth[c] = depth[p] + 1
    assert depth.max() == info["levels"] and info["stack_bound"] >= 3 * info["levels"]
    return info


@pytest.mark.parametrize("collapse_dp", [False, True])
def test_index_boxes_contain_what_is_below_them(collapse_dp, monkeypatch):
    for tris in (scenes.random_triangles(5000, 3, extent=3.0, size=1.5),
                 scenes.random_triangles(300, 4, snap=0.5),                   # heavy ties, degenerate triangles
                 scenes.demo().triangles()):
        _check(tris, collapse_dp, monkeypatch)
    sc = rt.Scene()
    sc.push_synthetic(0, 5, 1, 12345)                                         # the room around a 20480-triangle mesh
    info = _check(sc.triangles(), collapse_dp, monkeypatch)
    assert info["expected_visits"] < 40


def test_index_of_tiny_scenes(monkeypatch):
    for n in (2, 3, 4, 5):
        _check(scenes.random_triangles(n, 11 + n), False, monkeypatch)
    nodes, info = rt.index_build_host(scenes.random_triangles(1, 1))
    assert info["nodes"] == 0           # one triangle: the root is the leaf


def test_builders_do_not_depend_on_the_number_of_host_threads(monkeypatch):
    """RT_AMD_HOST_THREADS bounds the worker threads of the host-side builders (a job with one process
    per GPU shares the host between its ranks); the reference tree and the search index must come
    out byte for byte the same with one thread and with many."""
    sc = rt.Scene()
    sc.push_synthetic(0, 7, 1, 12345)                       # 327 700 triangles: every parallel path is taken
    t = sc.triangles()
    out = {}
    for threads in ("1", "3", "16"):
        monkeypatch.setenv("RT_AMD_HOST_THREADS", threads)
        tris, nodes = rt.bvh_build(t)
        index, info = rt.index_build_host(tris)
        out[threads] = (tris.tobytes(), nodes.tobytes(), index.tobytes(), info["levels"])
    assert out["1"] == out["3"] == out["16"]

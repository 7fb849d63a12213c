"""The device-side BVH builder (raytracer_amd/csrc/rt_bvh_device.hip.h; SURVEY 8(f)2, the reference's counterpart is
KdTree::build, src/core/bounds.rs:452-610, its "pre-process scene" step): the tree it builds is the host builder's
tree (same split rule, same f64 arithmetic), it is a valid tree whatever the input, and results do not depend on which
builder ran (ids are tree-independent by construction: every leaf primitive takes the exact reference predicate)."""
import os

import numpy as np
import pytest

from conftest import load_scene

pytestmark = pytest.mark.gpu
LEAF = 0x80000000


def _build(desc, which, device):
    old = os.environ.get("MI355RT_BVH")
    os.environ["MI355RT_BVH"] = which
    try:
        return device.DeviceScene(desc, pool_bytes=2 << 30)
    finally:
        if old is None:
            del os.environ["MI355RT_BVH"]
        else:
            os.environ["MI355RT_BVH"] = old


def _leaves(nodes, ids, root):
    """-> sorted list of frozensets of primitive ids per leaf, max depth; checks every record is referenced once."""
    out, seen, depth_max = [], np.zeros(len(ids), np.int32), 0
    stack = [(root, 0)]
    while stack:
        ref, depth = stack.pop()
        depth_max = max(depth_max, depth)
        if ref & LEAF:
            first, cnt = ref & 0x0FFFFFFF, ((ref >> 28) & 7) + 1
            seen[first:first + cnt] += 1
            out.append(frozenset(ids[first:first + cnt].tolist()))
        else:
            stack.append((int(nodes["left"][ref]), depth + 1))
            stack.append((int(nodes["right"][ref]), depth + 1))
    assert (seen == 1).all()
    return sorted(out, key=min), depth_max


@pytest.mark.parametrize("name", ["scene", "mesh", "cornell_box", "mesh_test", "test"])
def test_device_tree_is_the_host_tree(name):
    from raytracer_amd import device

    s = load_scene(name, size=16)
    host_built, dev_built = _build(s.desc, "host", device), _build(s.desc, "device", device)
    hn, hi, hroot, hflag = host_built.bvh()
    dn, di, droot, dflag = dev_built.bvh()
    assert not hflag and dflag
    assert hroot == droot and len(hn) == len(dn)
    for f in ("lmin", "lmax", "rmin", "rmax", "left", "right"):  # breadth-first numbering, boxes and child refs
        assert np.array_equal(hn[f], dn[f]), f
    hl, hd = _leaves(hn, hi, hroot)
    dl, dd = _leaves(dn, di, droot)
    assert hl == dl and hd == dd  # the same primitives share leaves (their order inside a leaf is free)
    assert host_built.info()["bvh_depth"] == dev_built.info()["bvh_depth"]


def test_results_do_not_depend_on_the_builder():
    from oracle import binding as ob
    from raytracer_amd import device

    s = load_scene("mesh_test", size=96, spp=2)
    cpu = ob.OracleScene(s.desc)
    a, b = _build(s.desc, "host", device), _build(s.desc, "device", device)
    ids_c = cpu.query_primary(ob.MODE_CANONICAL)[0]
    for g in (a, b):
        assert np.array_equal(g.query_primary()[0], ids_c)
    acc_a, _, n_a, _ = a.render(want_rgba=False)
    acc_b, _, n_b, _ = b.render(want_rgba=False)
    assert n_a == n_b and np.array_equal(acc_a, acc_b)
    rng = np.random.default_rng(11)
    o = rng.uniform(-3, 3, size=(20000, 3))
    d = rng.normal(size=(20000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ids_g, t_g = b.trace_rays(o, d)
    ids_o, t_o = cpu.trace_rays(o, d, mode=ob.MODE_CANONICAL)
    assert np.array_equal(ids_g, ids_o)


def test_degenerate_inputs_build_valid_trees():
    """Duplicates (identical centroids: no SAH split exists), one primitive, two primitives, a long row (deep tree)."""
    import json

    from raytracer_amd import device, host

    def scene(objects):
        return json.dumps({"width": 8, "height": 8, "camera": {"position": [0, 0, 5]}, "lights": [{"type": "ambient", "color": [1, 1, 1]}], "objects": objects})

    ball = lambda x, r=0.1: {"type": "sphere", "radius": r, "transform": [{"translate": [x, 0, 0]}], "material": {"type": "phong", "color": [1, 0, 0]}}
    cases = {"one": [ball(0)], "two": [ball(-1), ball(1)], "dups": [ball(0.5) for _ in range(37)], "row": [ball(0.3 * i, 0.05) for i in range(300)],
             "dups_and_more": [ball(0.0) for _ in range(9)] + [ball(2.0), ball(3.0, 0.5)]}
    for name, objs in cases.items():
        s = host.LoadedScene.from_string(scene(objs))
        h, d = _build(s.desc, "host", device), _build(s.desc, "device", device)
        dn, di, droot, dflag = d.bvh()
        assert dflag, name
        leaves, depth = _leaves(dn, di, droot)
        assert sorted(i for leaf in leaves for i in leaf) == list(range(len(objs))), name
        assert depth <= 38
        assert np.array_equal(h.query_primary()[0], d.query_primary()[0]), name
        a, _, n_a, _ = h.render(want_rgba=False)
        b, _, n_b, _ = d.render(want_rgba=False)
        assert n_a == n_b and np.array_equal(a, b), name

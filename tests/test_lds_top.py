"""The traversal kernels keep the top of the scene's trees in LDS with the box coordinates already converted to f32
(rt_kernels.hip.h `stage_top_nodes`), and read the rest of the tree from memory as 16-bit coordinates converted on the
fly.  The conversion is exact, so how much of a tree is resident must not change a frame by a bit: nothing resident,
a few nodes (most descents cross from the copy into memory), and the default."""
import os

import numpy as np
import pytest

from conftest import load_scene

pytestmark = pytest.mark.gpu


def _scene(desc, top, device):
    old = os.environ.get("MI355RT_TOP_NODES")
    if top is not None:
        os.environ["MI355RT_TOP_NODES"] = str(top)
    try:
        return device.DeviceScene(desc, pool_bytes=2 << 30)  # the library reads the cap at rt_scene_create
    finally:
        if old is None:
            os.environ.pop("MI355RT_TOP_NODES", None)
        else:
            os.environ["MI355RT_TOP_NODES"] = old


@pytest.mark.parametrize("name,size,spp", [("scene", 160, 2), ("mesh_test", 96, 2), ("cornell_box", 128, 4), ("mesh", 128, 2)])
def test_frames_do_not_depend_on_how_much_of_the_tree_is_in_lds(name, size, spp):
    from raytracer_amd import device

    s = load_scene(name, size=size, spp=spp)
    ref_acc = ref_cnt = ref_ids = None
    for top in (None, 0, 7, 40):
        g = _scene(s.desc, top, device)
        ids, t, n = g.query_primary()
        acc, _, cnt, _ = g.render(want_rgba=False)
        g.close()
        if ref_acc is None:
            ref_acc, ref_cnt, ref_ids = acc, cnt, ids
            assert cnt > 0
        else:
            assert cnt == ref_cnt and np.array_equal(ids, ref_ids) and np.array_equal(acc, ref_acc), f"top={top}"

"""Two-phase GI queries (k_trace_gi, DESIGN.md "two-phase GI"): a hemisphere ray of compute_global_illumination
(raytracing_scene.rs:98-133) needs its closest hit only when that hit is a SPECIAL primitive (emissive, or
get_illumination casts child rays from it, :135-167 / :263-337); of any other closest hit it uses
`distance <= max_occlusion_distance` alone (:512-516).  So the kernel finds the closest special hit in the tree of the
special primitives and then asks whether anything precedes it / lies within the occlusion distance -- an any-hit query.
The frames must not change by a bit, ties included (cornell_box: the emissive light is coplanar with the ceiling)."""
import os

import numpy as np
import pytest

from conftest import load_scene

pytestmark = pytest.mark.gpu


def _scene(desc, mode, device):
    old = os.environ.get("MI355RT_GI_TWO_PHASE")
    os.environ["MI355RT_GI_TWO_PHASE"] = mode
    try:
        return device.DeviceScene(desc, pool_bytes=2 << 30)
    finally:
        if old is None:
            del os.environ["MI355RT_GI_TWO_PHASE"]
        else:
            os.environ["MI355RT_GI_TWO_PHASE"] = old


# physical / refraction: every primitive is special (the default there is one phase); scene / cornell_box / mesh: a few are
@pytest.mark.parametrize("name,size,spp", [("scene", 160, 2), ("cornell_box", 128, 4), ("physical", 96, 2), ("refraction", 128, 2),
                                            ("mesh", 128, 2), ("test", 128, 2), ("mesh_test", 96, 2)])
def test_two_phase_frames_are_the_one_phase_frames(name, size, spp):
    from raytracer_amd import device

    s = load_scene(name, size=size, spp=spp)
    one, two = _scene(s.desc, "0", device), _scene(s.desc, "1", device)
    acc1, _, n1, st1 = one.render(want_rgba=False, want_stats=True)
    acc2, _, n2, st2 = two.render(want_rgba=False, want_stats=True)
    assert n1 == n2 and st1.closest_rays == st2.closest_rays and st1.shadow_rays == st2.shadow_rays
    assert np.array_equal(acc1, acc2)
    assert acc1.max() > 0.0


@pytest.mark.parametrize("occlusion", [0.0, 0.05, 3.0, 1e300])
def test_two_phase_over_occlusion_distances(occlusion):
    from raytracer_amd import device

    s = load_scene("cornell_box", size=96, spp=2, max_occlusion_distance=occlusion)
    one, two = _scene(s.desc, "0", device), _scene(s.desc, "1", device)
    acc1, _, n1, _ = one.render(want_rgba=False)
    acc2, _, n2, _ = two.render(want_rgba=False)
    assert n1 == n2 and np.array_equal(acc1, acc2)

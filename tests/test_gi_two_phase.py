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


def _scene(desc, mode, device, waves=None):
    """A scene pinned to one form of GI query (and one k_trace_gi build): the library reads the pins at rt_scene_create."""
    pins = {"MI355RT_GI_TWO_PHASE": mode, "MI355RT_GI_WAVES": waves}
    old = {k: os.environ.get(k) for k in pins}
    for k, v in pins.items():
        if v is not None:
            os.environ[k] = v
    try:
        return device.DeviceScene(desc, pool_bytes=2 << 30)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


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


def test_every_variant_the_library_may_pick_gives_the_same_frame():
    """rt_api.hip GiTune tries (query form) x (k_trace_gi built for 3 or 4 waves per SIMD) on a scene's first frames and
    keeps the fastest: whatever it picks, the frame is the same."""
    from raytracer_amd import device

    s = load_scene("cornell_box", size=160, spp=4)
    frames, seen = [], set()
    for mode in ("0", "1"):
        for waves in ("3", "4"):
            g = _scene(s.desc, mode, device, waves)
            acc, _, n, st = g.render(want_rgba=False, want_stats=True)
            assert (st.gi_two_phase, st.gi_waves) == (7 * int(mode), int(waves))
            seen.add((st.gi_two_phase, st.gi_waves))
            frames.append((acc, n))
    assert len(seen) == 4
    for acc, n in frames[1:]:
        assert n == frames[0][1] and np.array_equal(acc, frames[0][0])
    # unpinned: an untimed call, two with the starting configuration (cornell_box: max_occlusion_distance 1 in a scene of
    # extent 550 -> two-phase GI queries, four waves), then two per switch (both special sets are small, so all four
    # switches are open), then one configuration stays -- and it is one of those that were tried
    g = device.DeviceScene(s.desc, pool_bytes=2 << 30)
    picks = []
    for _ in range(14):
        acc, _, n, st = g.render(want_rgba=False, want_stats=True)
        assert n == frames[0][1] and np.array_equal(acc, frames[0][0])
        picks.append((st.gi_two_phase, st.gi_waves))
    assert st.gi_rays >= 1000000, "the frame is too small to settle anything"
    assert picks[0] == picks[1] == picks[2] == (3, 4), picks
    trials = picks[1:11]
    assert all(trials[i] == trials[i + 1] for i in range(0, 10, 2)) and len(set(trials)) == 5, picks
    assert len(set(picks[11:])) == 1 and picks[11] in trials, picks


@pytest.mark.parametrize("seed", range(32))
def test_two_phase_on_random_scenes(seed):
    """Seeded random scenes (tests/random_scenes.py): special PLANES (emissive, mirrors), special and plain primitives
    overlapping at random, every material kind and side -- forced two-phase queries (both special sets, whatever their
    size, and k_trace_illum_last) against closest-hit queries, bit for bit."""
    from raytracer_amd import device, host
    from conftest import SCENES
    from random_scenes import random_scene

    s = host.LoadedScene.from_string(random_scene(seed), asset_base=SCENES)
    one, two = _scene(s.desc, "0", device), _scene(s.desc, "1", device)
    acc1, _, n1, st1 = one.render(want_rgba=False, want_stats=True)
    acc2, _, n2, st2 = two.render(want_rgba=False, want_stats=True)
    assert st2.gi_two_phase == 7 and st1.gi_two_phase == 0
    assert n1 == n2 and st1.closest_rays == st2.closest_rays and st1.shadow_rays == st2.shadow_rays
    assert np.array_equal(acc1, acc2)

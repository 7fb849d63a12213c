#!/usr/bin/env python3
"""Generate tests/golden/oracle_<scene>.npz: small fixtures produced by the CPU
oracle (oracle/rt_oracle.cpp, canonical mode, seed 0) so that the GPU tests and
later refactors of the oracle are pinned to committed numbers.

Each file holds, for a 64x64 render of the scene (spp as in the scene file):
ids / t / normals of the pixel-centre primary hits, the per-pixel sum of
clamped sample colours, the RGBA8 image and the ray count.  ray_counts.json
holds the oracle's full-size (800x800) ray counts next to the numbers the
reference publishes in renders/renders.md.

usage: python tools/make_golden.py            (about 3 minutes on 8 cores)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from raytracer_amd import host  # noqa: E402

SMALL = ["test", "scene", "mesh", "refraction", "physical", "cornell_box", "occlusion", "emissive"]
PUBLISHED = {  # renders/renders.md:9-51
    "scene": 147115715, "mesh": 38326926, "refraction": 229283763, "physical": 533716349,
    "mesh_test": 41492072, "occlusion": 54222719, "emissive": 31854333, "test": 35707210,
}


def main():
    out = os.path.join(ROOT, "tests", "golden")
    for name in SMALL:
        s = host.LoadedScene.from_file(os.path.join(ROOT, "scenes", f"{name}.json"))
        s.set_option("width", 64)
        s.set_option("height", 64)
        o = ob.OracleScene(s.desc)
        ids, t, nrm = o.query_primary(ob.MODE_CANONICAL)
        acc, rgba, count = o.render(mode=ob.MODE_CANONICAL, threads=8)
        np.savez_compressed(os.path.join(out, f"oracle_{name}.npz"), ids=ids, t=t, normals=nrm, accum=acc, rgba=rgba,
                            ray_count=np.uint64(count), spp=np.uint32(s.opts.samples_per_pixel))
        print(name, "64x64 rays", count, flush=True)
    counts = {}
    for name, pub in PUBLISHED.items():
        if name == "mesh_test" and "--all" not in sys.argv:
            continue
        s = host.LoadedScene.from_file(os.path.join(ROOT, "scenes", f"{name}.json"))
        o = ob.OracleScene(s.desc)
        _, _, kd = o.render(mode=ob.MODE_KD, threads=8, want_rgba=False)
        _, _, canon = o.render(mode=ob.MODE_CANONICAL, threads=8, want_rgba=False)
        counts[name] = {"published": pub, "oracle_kd": kd, "oracle_canonical": canon}
        print(name, counts[name], flush=True)
    with open(os.path.join(out, "ray_counts.json"), "w") as f:
        json.dump(counts, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

"""Seeded random scenes in the reference's JSON schema for differential tests: every
primitive kind, both materials, all sides, nested groups, odd options."""
import json

import numpy as np


def random_scene(seed):
    r = np.random.default_rng(seed)
    u = lambda lo, hi: float(r.uniform(lo, hi))
    vec = lambda lo, hi: [u(lo, hi), u(lo, hi), u(lo, hi)]
    pick = lambda xs: xs[int(r.integers(len(xs)))]

    def material():
        side = pick(["Front", "Front", "Both", "Back"])
        tex = pick([None, None, None, "textures/checker.png", "textures/test.jpg"])
        if r.random() < 0.5:
            m = {"type": "phong", "side": side, "color": vec(0, 1), "specular": vec(0, 1), "shininess": u(1, 60),
                 "reflectivity": pick([0.0, 0.0, u(0, 1), 1.0])}
        else:
            m = {"type": "physical", "side": side, "color": vec(0, 1), "roughness": pick([0.0, u(0, 1), 1.0]), "metalness": pick([0.0, u(0, 1), 1.0]),
                 "opacity": pick([1.0, 1.0, u(0, 1)]), "refractive_index": u(1.0, 1.8)}
        if r.random() < 0.25:
            m["emissive"] = vec(0, 0.6)
        if tex:
            m["texture"] = tex
        return m

    def transform():
        t = []
        for _ in range(int(r.integers(0, 4))):
            k = pick(["translate", "rotate", "scale"])
            if k == "translate":
                t.append({"translate": vec(-2.5, 2.5)})
            elif k == "rotate":
                t.append({"rotate": [pick([[0, 1, 0], [1, 0, 0], [1, 0, -1], [-0.25, 0.9, 0], vec(-1, 1)]), u(-180, 180)]})
            else:
                t.append({"scale": pick([[u(0.3, 2)] * 3, vec(0.3, 2)])})
        return t

    def obj(depth=0):
        kind = pick(["sphere", "cube", "triangle", "sphere", "cube", "triangle", "plane", "group", "mesh"])
        if kind == "group" and depth < 2:
            return {"type": "group", "transform": transform(), "children": [obj(depth + 1) for _ in range(int(r.integers(1, 4)))]}
        if kind == "group":
            kind = "sphere"
        o = {"type": kind, "transform": transform(), "material": material()}
        if kind == "sphere":
            o["radius"] = u(0.2, 1.5)
        elif kind == "cube":
            o["size"] = u(0.3, 2.0)
        elif kind == "plane":
            o["normal"] = pick([[0, 1, 0], [0, 0, 1], [1, 0, 0], [0.3, 1, 0.2]])
            o["transform"] = [{"translate": [0, u(-3, -1), 0]}] if o["normal"][1] else [{"translate": [u(-6, -4), 0, u(-6, -4)]}]
        elif kind == "triangle":
            o["vertices"] = [vec(-1.5, 1.5) for _ in range(3)]
        else:
            o["file"] = pick(["models/cornell_box/short_block.obj", "models/cornell_box/tall_block.obj"])
            o["transform"] = [{"scale": [0.006] * 3}] + transform()
        return o

    lights = [{"type": "ambient", "color": vec(0, 0.3)}] if r.random() < 0.7 else []
    for _ in range(int(r.integers(0, 4))):
        lights.append({"type": "point", "transform": [{"translate": vec(-5, 5)}], "color": vec(0.2, 1), "intensity": u(2, 40)})
    scene = {"width": int(r.integers(8, 40)), "height": int(r.integers(8, 40)), "max_depth": int(r.integers(1, 5)),
             "samples_per_pixel": int(r.integers(1, 5)), "max_reflected_rays": int(pick([0, 1, 4, 9, 32])),
             "max_illumination_rays": int(pick([0, 1, 5, 16])), "max_occlusion_distance": u(0.2, 12),
             "camera": {"fov": u(30, 100), "position": [u(-2, 2), u(0, 3), u(4, 9)], "target": vec(-0.5, 0.5)},
             "lights": lights, "objects": [obj() for _ in range(int(r.integers(1, 9)))]}
    return json.dumps(scene)

#!/usr/bin/env python3
"""Counterpart of the reference's `render-scenes` binary (src/render_scenes.rs: the source of every number in
renders/renders.md) on the MI355X path: the same eight scenes, three iterations each, the same report lines, the
last frame written as PNG.  usage: tools/render_scenes.py [output_dir] [--scenes NAME ...] [--iterations N]   (GPU box)
Times are the library's render call (rt_render: kernels + the copy of the frame to the host), the counterpart of
CastTimings.ray_casting_duration; a different seed per iteration, as the reference's thread_rng would give."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import _abi, device, host  # noqa: E402

SCENES = ["scene", "mesh", "refraction", "physical", "mesh_test", "occlusion", "emissive", "test"]  # render_scenes.rs:10-19
ITERATIONS = 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir", nargs="?", default="renders")
    ap.add_argument("--scenes", nargs="+", default=SCENES)
    ap.add_argument("--iterations", type=int, default=ITERATIONS)
    args = ap.parse_args()
    out_dir, iterations = args.out_dir, args.iterations
    os.makedirs(out_dir, exist_ok=True)
    for name in args.scenes:
        path = os.path.join(ROOT, "scenes", f"{name}.json")
        scene = host.LoadedScene.from_file(path)
        gpu = device.DeviceScene(scene.desc)
        print(f"Raytracing {path} ({scene.n_prims} primitives)...")
        total_s, total_rays, rgba = 0.0, 0, None
        for i in range(iterations):
            t0 = time.perf_counter()
            _, rgba, rays, _ = gpu.render(_abi.RenderParams(seed=i))
            dt = time.perf_counter() - t0
            total_s += dt
            total_rays += rays
            print(f"│ Iteration {i + 1}: rendered in {dt * 1e3:.3f}ms ({rays} rays)")
        w, h, spp = int(scene.opts.width), int(scene.opts.height), int(scene.opts.samples_per_pixel)
        print(f"│\n│ Final render: {w} x {h} pixels, {spp} spp, {scene.n_prims} primitives, {total_rays // iterations} rays, "
              f"{total_s / iterations * 1e3:.3f}ms")
        out = os.path.join(out_dir, f"{name}.png")
        host.write_png(out, rgba)
        print(f"└ Wrote rendered image to {out}\n")
        gpu.close()


if __name__ == "__main__":
    main()

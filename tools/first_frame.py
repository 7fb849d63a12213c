#!/usr/bin/env python3
"""What one frame costs a process that renders one frame (the reference's use: src/main.rs:46-74 prints load /
pre-process / render times and exits).  For each published scene:
  cli      `raytracer_amd/raytrace scenes/X.json -o x.png` in a fresh process: its own four timing lines
           (load assets | pre-process = rt_scene_create | render = first rt_render | total; PNG = the rest)
  library  in this process (HIP already initialised by an earlier scene): kernel time of the scene's first
           rt_render call, of its second, and of the settled one (after the closest-hit switches are settled),
           with the switch set each ran
usage: first_frame.py [out.json]   (on the GPU box, from the repo root)"""
import json
import re
import subprocess
import sys
import time

sys.path.insert(0, ".")
from raytracer_amd import device, host  # noqa: E402

SCENES = ["scene", "mesh", "refraction", "physical", "mesh_test", "occlusion", "emissive", "test"]  # renders/renders.md order


def form(st):
    return "".join(c if st.gi_two_phase >> i & 1 else "-" for i, c in enumerate("ILN")) + f"/{st.gi_waves}w"


def cli(name):
    t0 = time.perf_counter()
    out = subprocess.run(["raytracer_amd/raytrace", f"scenes/{name}.json", "-o", "/tmp/first_frame.png", "--no-progress"], capture_output=True, text=True, timeout=120)
    wall = time.perf_counter() - t0
    if out.returncode != 0:
        raise SystemExit(out.stdout + out.stderr)
    f = lambda pat: float(re.search(pat, out.stdout).group(1))
    load, pre, render = f(r"Took ([0-9.]+)s to load"), f(r"Took ([0-9.]+)s to pre-process"), f(r"Took ([0-9.]+)s to render")
    total, kernels, png = f(r"Output written to \S+ in ([0-9.]+)s"), f(r"kernels ([0-9.]+) ms"), f(r"\+ ([0-9.]+) ms to encode")
    return {"process_wall_ms": round(1e3 * wall, 1), "load_ms": round(1e3 * load, 2), "rt_scene_create_ms": round(1e3 * pre, 2), "first_rt_render_ms": round(1e3 * render, 2),
            "first_rt_render_kernel_ms": round(kernels, 2), "png_ms": round(png, 2), "total_ms": round(1e3 * total + png, 2)}


def library(name):
    s = host.LoadedScene.from_file(f"scenes/{name}.json")
    t0 = time.perf_counter()
    g = device.DeviceScene(s.desc)
    create = time.perf_counter() - t0
    rows = []
    for it in range(14):
        t0 = time.perf_counter()
        _, _, n, st = g.render(want_rgba=False, want_stats=True)
        rows.append({"wall_ms": round(1e3 * (time.perf_counter() - t0), 2), "kernel_ms": round(st.ms_total, 3), "switches": form(st), "batches": st.batches, "retries": st.retries})
    info = g.info()
    g.close()
    first, second, settled = rows[0], rows[1], min(rows[-2:], key=lambda r: r["kernel_ms"])
    return {"rays": int(n), "rt_scene_create_ms": round(1e3 * create, 2), "pool_node_capacity": info["node_capacity"], "first": first, "second": second, "settled": settled,
            "first_over_settled": round(first["kernel_ms"] / settled["kernel_ms"], 3), "start_is_settled": first["switches"] == settled["switches"]}


def main():
    out = {"_about": __doc__.split("usage:")[0].strip()}
    # the fresh processes first, while this one has not touched the GPU (two processes on one GPU take turns: a CLI run beside
    # this process's HIP context showed stalls of half a second); one discarded run pages the ROCm libraries in from the
    # fresh box's disk (scene.json as the first process of a box: 215 ms of rt_scene_create instead of 81)
    cli("test")
    clis = {name: cli(name) for name in SCENES}
    library("test")  # HIP start-up of this process is not a scene's cost
    for name in SCENES:
        out[name] = {"cli": clis[name], "library": library(name)}
        c, l = out[name]["cli"], out[name]["library"]
        print(f"{name:11s} cli: load {c['load_ms']:7.1f} create {c['rt_scene_create_ms']:7.1f} render {c['first_rt_render_ms']:7.1f} (kernels {c['first_rt_render_kernel_ms']:6.1f}) png {c['png_ms']:6.1f} total {c['total_ms']:7.1f} | "
              f"library: first {l['first']['kernel_ms']:7.2f} [{l['first']['switches']}, {l['first']['batches']} batches, {l['first']['retries']} retries] second {l['second']['kernel_ms']:7.2f} "
              f"settled {l['settled']['kernel_ms']:7.2f} [{l['settled']['switches']}] first/settled {l['first_over_settled']:.2f}", flush=True)
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")


if __name__ == "__main__":
    main()

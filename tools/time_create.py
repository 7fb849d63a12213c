import os, sys, tempfile, time
sys.path.insert(0, ".")
from raytracer_amd import device, host, synthetic
# one-time costs first (library load, HIP runtime start-up, code-object load), so the lines below are per-scene costs
t0 = time.perf_counter(); w = device.DeviceScene(host.LoadedScene.from_file("scenes/test.json").desc); print(f"first scene of the process (9 primitives): {time.perf_counter() - t0:.2f} s")
p = synthetic.write_scene(os.path.join(tempfile.gettempdir(), "mi355rt_synthetic"), width=1920, height=1080, spp=4)
t0 = time.perf_counter(); s = host.LoadedScene.from_file(p); t1 = time.perf_counter()
g = device.DeviceScene(s.desc); t2 = time.perf_counter()
print(f"load (JSON + OBJ parse) {t1 - t0:.2f} s, rt_scene_create (BVH build + upload) {t2 - t1:.2f} s", g.info())
for name in ("scene", "mesh_test"):
    t0 = time.perf_counter(); s = host.LoadedScene.from_file(f"scenes/{name}.json"); t1 = time.perf_counter()
    g = device.DeviceScene(s.desc); t2 = time.perf_counter()
    print(f"{name}: load {t1 - t0:.3f} s, rt_scene_create {t2 - t1:.3f} s")

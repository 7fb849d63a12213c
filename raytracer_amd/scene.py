"""Host-side mirror of the reference's public interface for the hot path.

    reference (Rust)                                   here
    serde_json::from_reader::<Scene> + load_assets     Scene.from_file / Scene.from_json
      (src/main.rs:34-43, src/render/scene.rs:65-74)
    Scene::build_raytracing_scene (scene.rs:76-78)     Scene.build_raytracing_scene()
    RaytracingScene::raytrace_to_image(use_progress)   RaytracingScene.raytrace_to_image()
      -> (RgbaImage, CastTimings, CastStats)             -> (rgba8 ndarray, CastTimings, CastStats)
      (src/render/raytracing_scene.rs:703-776)
    RaytracingScene::get_width/get_height/get_num_objects (:70-88)   same names

Errors the reference reports by panicking (bad JSON, missing assets) surface
as `HostError`; device failures as `DeviceError`.  There is no CPU renderer in
this package: `raytrace_to_image` needs the HIP library and a GPU.
"""
import time
from dataclasses import dataclass

import numpy as np

from . import _abi, device, host


@dataclass
class CastStats:  # src/render/mod.rs:96-111
    ray_count: int = 0


@dataclass
class CastTimings:  # src/render/mod.rs:59-94 (seconds)
    ray_casting_duration: float = 0.0
    post_processing_duration: float = None  # denoise pass: feature off in this build
    kernel_ms: dict = None


class Scene:
    def __init__(self, loaded):
        self._loaded = loaded

    @classmethod
    def from_file(cls, path):
        return cls(host.LoadedScene.from_file(path))

    @classmethod
    def from_json(cls, text, asset_base=""):
        return cls(host.LoadedScene.from_string(text, asset_base))

    @property
    def render_options(self):
        return self._loaded.opts

    def set_option(self, key, value):
        self._loaded.set_option(key, value)

    @property
    def desc(self):
        return self._loaded.desc

    def build_raytracing_scene(self, pool_bytes=None):
        return RaytracingScene(self._loaded, pool_bytes)


class RaytracingScene:
    def __init__(self, loaded, pool_bytes=None):
        self._loaded = loaded  # keeps the descriptor arrays alive
        self._dev = device.DeviceScene(loaded.desc, pool_bytes)

    def get_width(self):
        return self._dev.width

    def get_height(self):
        return self._dev.height

    def get_num_objects(self):
        return self._loaded.n_prims

    @property
    def device_scene(self):
        return self._dev

    def raytrace_to_image(self, use_progress=False, seed=0):
        """-> (rgba8[h,w,4], CastTimings, CastStats).  `use_progress` is accepted
        for signature parity; a frame takes a fraction of a second here."""
        params = _abi.RenderParams(seed=seed)
        t0 = time.perf_counter()
        _, rgba, count, stats = self._dev.render(params, want_rgba=True, want_stats=True)
        dt = time.perf_counter() - t0
        timings = CastTimings(ray_casting_duration=dt, kernel_ms={k: v for k, v in stats.as_dict().items() if k.startswith("ms_")})
        return rgba, timings, CastStats(ray_count=count)

    def primary_hits(self):
        """Pixel-centre closest hits (ids, t, normals): the parity AOVs."""
        return self._dev.query_primary()

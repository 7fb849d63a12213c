import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: the product has no CPU path.
    pass


@pytest.fixture(scope="session")
def libs():
    """Host front end + oracle are built on demand (CPU only); the HIP library is
    prebuilt by __graft_entry__.build() and only loaded."""
    from oracle import binding as ob
    from raytracer_amd import host

    if not os.path.exists(os.path.join(ROOT, "raytracer_amd", "libmi355rt_host.so")):
        import subprocess

        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "raytracer_amd", "csrc"), "host"])
    host.lib()
    ob.lib()
    return host, ob


def load_scene(name, size=None, spp=None, **opts):
    from raytracer_amd import host

    s = host.LoadedScene.from_file(os.path.join(SCENES, f"{name}.json"))
    if size:
        s.set_option("width", size)
        s.set_option("height", size)
    if spp:
        s.set_option("samples_per_pixel", spp)
    for k, v in opts.items():
        s.set_option(k, v)
    return s

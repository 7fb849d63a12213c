"""BASELINE config 5: the procedural height-field scene --
generator determinism on the CPU, GPU-vs-oracle parity at the full 1,000,000
triangles (reduced resolution: the oracle must finish in seconds)."""
import os
import tempfile

import numpy as np
import pytest

from raytracer_amd import synthetic

OUT = os.path.join(tempfile.gettempdir(), "mi355rt_synthetic")
NO_HIT = 0xFFFFFFFF


def test_heightfield_is_deterministic_and_well_formed():
    v, n, f = synthetic.heightfield(40, 20)
    v2, n2, f2 = synthetic.heightfield(40, 20)
    assert np.array_equal(v, v2) and np.array_equal(n, n2) and np.array_equal(f, f2)
    assert v.shape == (41 * 21, 3) and f.shape == (2 * 40 * 20, 3)
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0) and (n[:, 1] > 0).all()
    assert v[:, 0].min() == -50.0 and v[:, 0].max() == 50.0 and v[:, 2].min() == -25.0 and v[:, 2].max() == 25.0
    assert np.abs(v[:, 1]).max() <= 0.6 + 1e-12  # 0.5 sin cos + 0.1 hash
    # every quad is split into two triangles with upward geometric normals
    p = v[f]
    g = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    assert (g[:, 1] > 0).all()
    h = synthetic._hash01(np.arange(1000), np.arange(1000)[::-1])
    assert 0.0 <= h.min() and h.max() < 1.0 and 0.4 < h.mean() < 0.6


def test_scene_files_load_with_the_reference_schema(libs):
    host, _ = libs
    path = synthetic.write_scene(OUT, nx=60, nz=30, width=48, height=27, spp=2)
    s = host.LoadedScene.from_file(path)
    assert s.n_prims == 2 * 60 * 30 + 64
    assert (s.opts.width, s.opts.height, s.opts.samples_per_pixel, s.opts.max_depth) == (48, 27, 2, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("nx,nz,w,h,spp", [(200, 100, 96, 54, 2), (1000, 500, 128, 72, 1)])
def test_synthetic_scene_matches_oracle(libs, nx, nz, w, h, spp):
    """(1000, 500) is the 1,000,000-triangle mesh of config 5."""
    host, ob = libs
    from raytracer_amd import device

    s = host.LoadedScene.from_file(synthetic.write_scene(OUT, nx=nx, nz=nz, width=w, height=h, spp=spp))
    gpu, cpu = device.DeviceScene(s.desc, pool_bytes=2 << 30), ob.OracleScene(s.desc)
    info = gpu.info()
    assert info["bounded"] == 2 * nx * nz + 64 and info["bvh_depth"] <= 38
    ids_g, t_g, n_g = gpu.query_primary()
    ids_c, t_c, n_c = cpu.query_primary(ob.MODE_CANONICAL)
    assert np.array_equal(ids_g, ids_c)
    hit = ids_c != NO_HIT
    assert hit.mean() > 0.3
    assert np.all(np.abs(t_g[hit] - t_c[hit]) <= 1e-5 * np.abs(t_c[hit])) and np.all(np.abs(n_g[hit] - n_c[hit]) <= 1e-5)
    acc_g, _, cnt_g, _ = gpu.render(want_rgba=False)
    acc_c, _, cnt_c = cpu.render(mode=ob.MODE_CANONICAL, threads=16, want_rgba=False)
    assert abs(cnt_g - cnt_c) <= max(2, cnt_c // 100000)
    assert float(np.sqrt(np.mean(((acc_g - acc_c) / spp) ** 2))) <= 1e-3
    acc_2, _, cnt_2, _ = gpu.render(want_rgba=False)
    assert cnt_2 == cnt_g and np.array_equal(acc_2, acc_g)


@pytest.mark.gpu
def test_config5_at_its_full_size_composes(libs):
    """BASELINE config 5 as quoted: 1,000,000 triangles, 3840x2160, 64 spp, on ONE GPU (16 G rays, ~4 s a frame).  No
    oracle finishes that in a test, so the frame is held to the properties the domain offers: the ray count and the
    frame do not depend on the pool budget (20 batches or 100), the sample ranges [0,32) + [32,64) compose to the
    whole frame (global sample indices key the RNG; per-sample clamp before the sum, raytracing_scene.rs:605-612), and
    the eight interleaved tile shards of a multi-GPU run compose to it exactly (one contributor per pixel)."""
    import torch

    host, _ = libs
    from raytracer_amd import _abi, device

    s = host.LoadedScene.from_file(synthetic.write_scene(OUT))
    w, h, spp = int(s.opts.width), int(s.opts.height), int(s.opts.samples_per_pixel)
    assert (w, h, spp, s.n_prims) == (3840, 2160, 64, 1000064)
    gpu = device.DeviceScene(s.desc)
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    n_full, st = gpu.render_device(full.data_ptr(), _abi.RenderParams(seed=0), stream=stream, want_stats=True)
    torch.cuda.synchronize()
    print(f"config 5 full size: {n_full} rays, {st.batches} batches, {st.ms_total:.0f} ms of kernels, {n_full / st.ms_total / 1e3:.0f} Mrays/s")
    assert n_full > 1.5e10 and st.retries == 0
    assert float(full.max()) <= spp and float(full.min()) >= 0.0 and float(full.mean()) > 1.0

    part = torch.zeros_like(full)
    n_halves = 0
    for s0, s1 in ((0, 32), (32, 64)):
        n, _ = gpu.render_device(part.data_ptr(), _abi.RenderParams(seed=0, s0=s0, s1=s1), stream=stream, want_stats=False)
        n_halves += n
    torch.cuda.synchronize()
    assert n_halves == n_full
    assert float((part - full).abs().max()) <= 1e-12

    part.zero_()
    n_shards = 0
    for k in range(8):
        n, _ = gpu.render_device(part.data_ptr(), _abi.RenderParams(seed=0, tile_size=8, shard_index=k, shard_count=8), stream=stream, want_stats=False)
        n_shards += n
    torch.cuda.synchronize()
    assert n_shards == n_full
    assert torch.equal(part, full)
    gpu.close()

    small = device.DeviceScene(s.desc, pool_bytes=6 << 30)  # a quarter of the default budget: four times the batches
    part.zero_()
    n_small, st_small = small.render_device(part.data_ptr(), _abi.RenderParams(seed=0), stream=stream, want_stats=True)
    torch.cuda.synchronize()
    assert st_small.batches > st.batches
    assert n_small == n_full and torch.equal(part, full)

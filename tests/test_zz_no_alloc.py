"""rt_render_device_noalloc (every pool in the library's static device scratch) against rt_render_device (pools in the
scene's hipMalloc'ed budget), through the C ABI: bit-equal accumulation buffers and equal ray counts, on two calls in a
row, the second with a smaller input; the size limit is an error, not a fault.  Collected last (test_zz_*): the round's
end runs `pytest -x`."""
import numpy as np
import pytest

from conftest import load_scene

pytestmark = pytest.mark.gpu


def _both(gpu, torch, abi, params, h, w, check_free=False):
    ref = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    out = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    n_ref, _ = gpu.render_device(ref.data_ptr(), params, want_stats=False)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]  # hipMemGetInfo: free bytes of the device
    n_out = gpu.render_device_noalloc(out.data_ptr(), params)
    torch.cuda.synchronize()
    if check_free:  # "allocates nothing": the device's free memory is what it was (not on a process's first call, which loads code)
        assert torch.cuda.mem_get_info()[0] == free_before
    return ref.cpu().numpy(), out.cpu().numpy(), n_ref, n_out


def test_noalloc_matches_render_device_twice():
    import torch

    from raytracer_amd import _abi as abi
    from raytracer_amd import device

    # scene.json: 1,524 primitives, both material models, max_depth 5 -- ~110 nodes per camera ray, so 160x160x4 camera
    # rays need several batches of the 128 MiB scratch (overflow / re-batching runs too)
    s = load_scene("scene", size=160)
    gpu = device.DeviceScene(s.desc)
    a_ref, a_out, n_ref, n_out = _both(gpu, torch, abi, abi.RenderParams(seed=3), 160, 160)
    assert n_ref == n_out and n_ref > 160 * 160 * 4
    assert np.array_equal(a_ref, a_out)
    # second call, smaller input (a window, two of the four samples): the scratch still holds the first call's contents
    p = abi.RenderParams(seed=3, x0=40, y0=24, x1=104, y1=72, s0=1, s1=3)
    b_ref, b_out, m_ref, m_out = _both(gpu, torch, abi, p, 160, 160, check_free=True)
    assert m_ref == m_out and 0 < m_ref < n_ref
    assert np.array_equal(b_ref, b_out)
    assert not b_out[:24].any() and not b_out[:, :40].any() and b_out[24:72, 40:104].any()
    # a second scene on the same GPU shares the array (sequential calls are fine)
    s2 = load_scene("test", size=96)
    gpu2 = device.DeviceScene(s2.desc)
    c_ref, c_out, k_ref, k_out = _both(gpu2, torch, abi, abi.RenderParams(shard_index=1, shard_count=3, tile_size=8), 96, 96)
    assert k_ref == k_out and np.array_equal(c_ref, c_out)
    # and the first scene again
    d_ref, d_out, _, _ = _both(gpu, torch, abi, p, 160, 160, check_free=True)
    assert np.array_equal(d_out, b_out) and np.array_equal(d_ref, b_ref)


def test_noalloc_rejects_oversized_calls():
    import torch

    from raytracer_amd import _abi as abi
    from raytracer_amd import device

    s = load_scene("test", size=800)  # 800x800x4 = 2.56 M camera rays > RT_NOALLOC_MAX_CAMERA_RAYS (262,144)
    gpu = device.DeviceScene(s.desc)
    out = torch.zeros((800, 800, 3), dtype=torch.float64, device="cuda:0")
    with pytest.raises(device.DeviceError, match="camera rays in this call, the limit is 262144"):
        gpu.render_device_noalloc(out.data_ptr())
    torch.cuda.synchronize()
    assert not out.any().item()
    # within the limit after splitting: a 256-row window of one sample
    n = gpu.render_device_noalloc(out.data_ptr(), abi.RenderParams(x0=0, y0=0, x1=800, y1=256, s0=0, s1=1))
    torch.cuda.synchronize()
    assert n > 0 and out[:256].any().item() and not out[256:].any().item()

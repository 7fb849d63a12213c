"""The CPU oracle against the reference's own committed renders (tests/golden/renders/*.png = /root/reference/renders),
calibrated against the oracle's seed-to-seed noise -- see tests/refstats.py for what is compared and why.  The same
checks run on the HIP path at full size in tests/test_gpu_parity.py.

Bounds (measured, oracle, 6-32 seeds; `python tools/ref_bias.py` prints the table):
  quiet pixels   within one 8-bit level of the reference on >= 97 % (measured 97.9-99.97 %), |mean difference| <= 0.1
                 level (measured <= 0.06), rms <= 0.5 (measured <= 0.37);
  block means    reference inside [raw - 4 sigma - 0.15, conv + 4 sigma + 0.15] per region (0.15 level: truncating
                 re-quantisation of the denoiser's output; + 0.03 level for the sigma estimate of few seeds); rms
                 distance of the reference to that interval <= 1.15 x a single frame's own rms to the mean of the
                 others + 0.15, or half a level (banding of a denoised gradient), whichever is larger;
  post-filter    the PNG is not raw tracer output: pixel-level noise below ours in every region, non-zero pixels
                 >= 2.5 px outside the silhouettes (test.png: 346, ours: 0).
"""
import os

import numpy as np
import pytest

import refstats
from conftest import GOLDEN, load_scene

# scene -> (seeds, rows rendered and compared); windows keep the CPU suite short, the GPU suite takes whole frames
CASES = {
    "test": (6, (0, 800)),
    "mesh": (4, (320, 640)),
    "refraction": (4, (280, 600)),
    "mesh_test": (4, (240, 560)),
    "physical": (3, (376, 456)),
}


def check_against_reference(name, regions, quiet, ev, few_seeds):
    """Shared with the GPU suite.  Raises AssertionError naming the region."""
    # 0.15 level: the denoiser's f32 output is re-quantised by truncation (:765-767), which moves 4-13 % of even the
    # quiet pixels by one level and turns a smooth dark gradient into a dozen bands whose block means sit anywhere within
    # half a level of a noisy frame's (refraction.png's glass slab: -0.13 +- 0.005 over 608 blocks, rms 0.39 against
    # 0.14 of our own seeds; mesh.png's mirror plane: rms 0.09 against 0.04); + 0.03 for a sigma from 3-4 seeds
    slack = 0.15 + (0.03 if few_seeds else 0.0)
    report = refstats.format_report(name, regions, quiet, ev)
    assert quiet, report
    for q in quiet:
        assert q["within1"] >= 0.97 and abs(q["mean"]) <= 0.1 and q["rms"] <= 0.5, f"quiet pixels of region {q['region']}\n{report}"
    for s in regions:
        assert s["ref_minus_raw"] >= -4.0 * s["sigma"] - slack, f"region {s['region']}: reference darker than the raw expectation\n{report}"
        assert s["ref_minus_conv"] <= 4.0 * s["sigma"] + slack, f"region {s['region']}: reference brighter than the converged frame\n{report}"
        assert s["rmse_ref"] <= max(1.15 * s["rmse_own"] + slack, 0.5), f"region {s['region']}: block-mean rms\n{report}"
    for h in ev["highpass"]:  # the premise of the interval test: the PNG has been through a denoiser
        assert h["ref"] <= 1.02 * h["ours"], f"region {h['region']}: the reference is noisier than a raw frame\n{report}"
    return report


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_against_reference_png(name):
    from PIL import Image

    from oracle import binding as ob
    from raytracer_amd import _abi

    seeds, (y0, y1) = CASES[name]
    s = load_scene(name)
    cpu = ob.OracleScene(s.desc)
    w, spp = int(s.opts.width), int(s.opts.samples_per_pixel)
    ids = cpu.query_primary(ob.MODE_CANONICAL)[0][y0:y1]
    ref = np.array(Image.open(os.path.join(GOLDEN, "renders", f"{name}.png")).convert("RGB"))[y0:y1]

    def render(seed):
        acc, rgba, _ = cpu.render(_abi.RenderParams(seed=seed, x0=0, y0=y0, x1=w, y1=y1), mode=ob.MODE_KD, threads=os.cpu_count() or 1)
        return acc[y0:y1], rgba[y0:y1]

    frames, conv = refstats.collect(render, seeds, spp)
    regions = refstats.region_stats(ref, frames, conv, ids)
    quiet = refstats.quiet_stats(ref, frames, ids)
    ev = refstats.filter_evidence(ref, frames, ids)
    check_against_reference(name, regions, quiet, ev, few_seeds=True)
    if name == "test":  # the silhouettes of a raw frame end where the geometry ends
        assert ev["far_nonzero"] >= 100 and ev["far_nonzero_ours"] == 0, ev

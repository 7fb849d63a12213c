"""Statistics that hold a renderer against the reference's committed PNGs (tests/golden/renders/*.png), calibrated
against the renderer's own seed-to-seed noise.  Pure numpy / scipy; the renderer is a callback, so the same code
serves the CPU oracle (tests/test_reference_renders.py) and the HIP path (tests/test_gpu_parity.py).

What the PNGs are.  renders/renders.md:3 says of ALL of them "rendered, then denoised using Intel's Open Image
Denoise library" -- and that holds for test / mesh / refraction / physical / mesh_test too, although their scene
files carry `skip_denoise_pass: true` today (`filter_evidence` below measures it: non-zero pixels several pixels
outside every silhouette, where a raw tracer's output is exactly zero, and pixel-level noise below a raw 4-spp
frame's in every region).  The reference's RNG stream (`thread_rng`, shuffled pixel order) is unknown as well.  So:

  * where the noise is nil there is nothing for a denoiser to move and no stream to match: on QUIET pixels (every
    one of our seeds gives the same 8-bit value +-1, a few pixels away from any noisy pixel) the reference PNG must
    equal ours to the 8-bit level (`quiet_stats`);
  * elsewhere a denoised frame lies between the expectation of the raw 4-spp frame ("raw": mean over seeds of the
    tone-mapped frames) and the converged frame ("conv": tone-mapped mean over seeds of the linear frames) -- the two
    differ by the Jensen gap of the concave gamma curve, 3 levels on an aliased checkerboard -- so block means of
    the reference must lie inside [raw - 4 sigma, conv + 4 sigma] region by region, sigma from our own seeds
    (`region_stats`).
"""
import math

import numpy as np
from scipy import ndimage

NO_HIT = 0xFFFFFFFF
GAMMA = 2.2


def block_mean(img, b=8):
    h, w, c = img.shape
    return img.reshape(h // b, b, w // b, b, c).mean(axis=(1, 3))


def block_labels(ids, b=8, majority=0.9):
    """Per b x b block: the first-hit primitive of >= `majority` of its pixel-centre rays, -1 = miss, -2 = mixed."""
    h, w = ids.shape
    x = ids.astype(np.int64)
    x[ids == NO_HIT] = -1
    x = x.reshape(h // b, b, w // b, b).transpose(0, 2, 1, 3).reshape(h // b, w // b, b * b)
    x = np.sort(x, axis=2)
    med = x[:, :, (b * b) // 2]
    share = (x == med[:, :, None]).mean(axis=2)
    return np.where(share >= majority, med, -2)


def collect(render, n_seeds, spp):
    """render(seed) -> (rgb_sum[h,w,3] f64, rgba8[h,w,4]).  -> (frames[n,h,w,3] u8, conv[h,w,3] f64): the tone-mapped
    frames and the converged frame, tone-mapped and quantised like the reference does (raytracing_scene.rs:614-621, :757-769)."""
    frames, lin = [], None
    for seed in range(n_seeds):
        acc, rgba = render(seed)
        frames.append(np.ascontiguousarray(rgba[..., :3]))
        a = np.clip(acc / spp, 0.0, 1.0)
        lin = a if lin is None else lin + a
    conv = np.floor(255.0 * np.clip(lin / n_seeds, 0.0, 1.0) ** (1.0 / GAMMA))
    return np.array(frames), conv


def region_stats(ref, frames, conv, ids, b=8, min_blocks=200):
    """Block means of the reference against ours, region by region (region = first-hit primitive of the block).
    -> list of dicts: blocks, ref_minus_raw, ref_minus_conv, sigma (of a single frame's signed mean, from the
    seed-to-seed spread), rmse_ref (to the nearest point of [raw, conv]), rmse_own (a single frame to the mean of the
    others)."""
    n = len(frames)
    labels = block_labels(ids, b)
    fb = np.array([block_mean(f.astype(np.float64), b) for f in frames])
    raw = fb.mean(axis=0)
    cb = block_mean(conv, b)
    rb = block_mean(ref.astype(np.float64), b)
    lo, hi = np.minimum(raw, cb), np.maximum(raw, cb)
    out = []
    for lab in sorted(set(labels.ravel().tolist())):
        m = labels == lab
        nb = int(m.sum())
        if nb < min_blocks:
            continue
        own, signed = [], []
        for k in range(n):
            dk = (fb[k] - (raw * n - fb[k]) / (n - 1))[m]
            own.append(math.sqrt(float((dk ** 2).mean())))
            signed.append(float(dk.mean()))
        outside = (rb - np.clip(rb, lo, hi))[m]
        out.append({
            "region": int(lab), "blocks": nb,
            "ref_minus_raw": float((rb - raw)[m].mean()), "ref_minus_conv": float((rb - cb)[m].mean()),
            "sigma": float(np.std(signed, ddof=1)),
            "rmse_ref": math.sqrt(float((outside ** 2).mean())), "rmse_ref_raw": math.sqrt(float(((rb - raw)[m] ** 2).mean())),
            "rmse_own": float(np.mean(own)),
        })
    return out


def quiet_stats(ref, frames, ids, erosion=2, spread=1, min_pixels=2000):
    """Pixels on which all our seeds agree within `spread` 8-bit levels, `erosion` pixels away from any that do not.
    -> list of dicts per first-hit primitive: pixels, mean and rms of ref - ours, share within one level, share equal."""
    f = frames.astype(np.int16)
    quiet = (f.max(axis=0) - f.min(axis=0)).max(axis=2) <= spread
    if erosion:
        quiet = ndimage.binary_erosion(quiet, iterations=erosion)
    ours = f.mean(axis=0)
    d_all = ref.astype(np.float64) - ours
    out = []
    for v in np.unique(ids):
        m = quiet & (ids == v)
        npx = int(m.sum())
        if npx < min_pixels:
            continue
        d = d_all[m]
        out.append({"region": -1 if v == NO_HIT else int(v), "pixels": npx, "of": int((ids == v).sum()), "mean": float(d.mean()),
                    "rms": math.sqrt(float((d ** 2).mean())), "within1": float((np.abs(d) <= 1.0).all(axis=1).mean()),
                    "equal": float((np.abs(d) < 0.5).all(axis=1).mean())})
    return out


def filter_evidence(ref, frames, ids):
    """Is the PNG raw tracer output?  -> dict:
    far_nonzero  background pixels >= 2.5 pixels from the nearest pixel whose centre ray hits anything, non-zero in the
                 reference (a raw frame is exactly zero there: every sample of such a pixel is a miss = ColorData::black);
    far_nonzero_ours  the same count in our frames (max over seeds);
    highpass     per first-hit primitive of >= 8000 interior pixels: rms of (pixel - 3x3 mean) in the reference / in ours."""
    hit = ids != NO_HIT
    ev = {"far_nonzero": 0, "far_nonzero_ours": 0, "far_pixels": 0, "highpass": []}
    if not hit.all():
        far = ndimage.distance_transform_edt(~hit) >= 2.5
        ev["far_pixels"] = int(far.sum())
        ev["far_nonzero"] = int(((ref.astype(np.int32).sum(axis=2) > 0) & far).sum())
        ev["far_nonzero_ours"] = max(int(((f.astype(np.int32).sum(axis=2) > 0) & far).sum()) for f in frames)
    hp = lambda a: a - ndimage.uniform_filter(a, 3)
    ref_hp = hp(ref.astype(np.float64).mean(axis=2))
    ours_hp = [hp(f.astype(np.float64).mean(axis=2)) for f in frames[:2]]
    for v in np.unique(ids):
        if v == NO_HIT:
            continue
        m = ndimage.binary_erosion(ids == v, iterations=3)
        if m.sum() < 8000:
            continue
        r = math.sqrt(float((ref_hp[m] ** 2).mean()))
        o = float(np.mean([math.sqrt(float((x[m] ** 2).mean())) for x in ours_hp]))
        ev["highpass"].append({"region": int(v), "pixels": int(m.sum()), "ref": r, "ours": o})
    return ev


def format_report(name, regions, quiet, ev):
    lines = [f"== {name}"]
    for s in regions:
        lines.append(f"  blocks  region {s['region']:6d} n {s['blocks']:5d}  ref-raw {s['ref_minus_raw']:+7.3f}  ref-conv {s['ref_minus_conv']:+7.3f}  sigma {s['sigma']:.4f}  "
                     f"rmse: ref to [raw,conv] {s['rmse_ref']:6.3f}, ref to raw {s['rmse_ref_raw']:6.3f}, own {s['rmse_own']:6.3f}")
    for q in quiet:
        lines.append(f"  quiet   region {q['region']:6d} pixels {q['pixels']:7d} of {q['of']:7d}  ref-ours mean {q['mean']:+.3f} rms {q['rms']:.3f}  within one level {q['within1']:.4f}  equal {q['equal']:.4f}")
    lines.append(f"  filter  background >= 2.5 px from any hit: {ev['far_pixels']} pixels, non-zero in the reference {ev['far_nonzero']}, in ours {ev['far_nonzero_ours']}")
    for h in ev["highpass"]:
        lines.append(f"  filter  region {h['region']:6d} pixel-level noise (rms of pixel - 3x3 mean): reference {h['ref']:.3f}, ours {h['ours']:.3f}, ratio {h['ref'] / max(h['ours'], 1e-12):.2f}")
    return "\n".join(lines)

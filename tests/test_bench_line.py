"""bench.py's one JSON line at N = 1 (the driver's BENCH_rNN.json): the fields the contract names are there, every
fraction is a fraction, and the CPU baseline says what it ran on.  A reduced frame (--size) keeps it to seconds; the
PMC replay (`traffic`, `valu`) belongs to the default workload only and is checked for presence of its keys."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_single_gpu_line():
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--size", "256", "--pool-gib", "4"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["dtype"] == "f64" and line["vs_baseline"] is None
    assert line["value"] > 0 and abs(line["value"] - line["config"]["rays_per_step"] / line["ms_per_step"] / 1e3) <= 0.01 * line["value"]
    r = line["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_hbm_stream", "frac_hbm_measured", "launches_per_step", "avg_launch_ms",
              "kernel_ms_per_step", "frame", "gi_query_form"):
        assert k in r, k
    assert r["kernel"] in ("k_trace_closest", "k_trace_gi", "k_trace_illum_last", "k_trace_shadow") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    fracs = [r["frac"], r["frac_hbm_stream"], r["frame"]["frac_hbm_stream"]] + [x for x in (r["frac_hbm_measured"], r["frame"]["frac_hbm_measured"]) if x is not None]
    assert all(0.0 < f <= 1.0 for f in fracs), fracs
    # the class the line names is the one with the largest time, and its launches and time are those of that class alone
    ms = r["kernel_ms_per_step"]
    assert set(ms) == {"raygen", "closest", "gi", "illum", "shadow", "expand", "resolve", "accum"}
    assert r["dominant_class"] == max(ms, key=ms.get)
    if r["dominant_class"] in ("closest", "gi", "shadow"):
        assert r["kernel"] == "k_trace_" + r["dominant_class"]
        assert abs(r["avg_launch_ms"] * r["launches_per_step"] - ms[r["dominant_class"]]) <= 0.02 * ms[r["dominant_class"]] + 1e-3
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["one_thread"]["cores"] == 1 and c["one_thread"]["value"] > 0
    assert c["host"]["nproc"] >= c["cores"] and c["host"]["cpu_model"]
    assert c["value"] < line["value"]  # a reported baseline, not a target -- but a GPU path slower than the port would be news

"""The state machine that settles a scene's closest-hit switches on its first frames (raytracer_amd/csrc/rt_tune.h; no
counterpart in the reference, whose one code path it chooses between equivalent forms of) driven on the CPU with
synthetic timings.  The GPU side -- every setting gives the same frame -- is tests/test_gi_two_phase.py."""
import os
import subprocess

from conftest import ROOT


def test_switches_are_settled_by_a_coordinate_walk(tmp_path):
    exe = str(tmp_path / "tune_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "native", "tune_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr

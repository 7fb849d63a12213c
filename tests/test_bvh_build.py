"""Host BVH builder (raytracer_amd/csrc/rt_bvh.cpp; replaces KdTree::build, src/core/bounds.rs:452-610, for the GPU
path) on the CPU: the tree is the same for any number of build workers, every primitive sits in exactly one leaf,
quantised node boxes contain what is below them, only triangles of one mesh share a leaf, depth fits the stack."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.parametrize("n", [1, 2, 1000, 200000])
def test_bvh_is_valid_and_independent_of_the_thread_count(tmp_path, n):
    exe = str(tmp_path / "bvh_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "bvh_check.cpp"),
                           os.path.join(ROOT, "raytracer_amd", "csrc", "rt_bvh.cpp"), "-o", exe])
    out = subprocess.run([exe, str(n)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr

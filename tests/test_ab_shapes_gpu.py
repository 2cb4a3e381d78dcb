"""The launch shapes of finished A/Bs — resident waves that refill finished lanes (RT_OPT_VARIANT 2) and the streamed shape
(4: primary pass -> path queue -> resident waves refilled from the queue) — are compiled only into the A/B build
(`make -C rsraytracingv3_amd/csrc ab` -> librt355_ab.so, -DRT_AB_SHAPES); the shipped librt355.so maps both values to the
re-packing shape.  The parity tests that are parametrized over the launch shapes run in this process against the shipped
library; here the shape-specific ones run once more, in ONE child process, against the A/B build — where variants 2 and 4 are the
real kernels — so that the A/B tools keep a library whose shapes are still bit-identical to the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "rsraytracingv3_amd", "librt355_ab.so")
SELECT = ("test_config1_plumbing_256 or test_headline_scene_small or test_batch_launch_equals_sequential_frames or test_streamed_shape or "
          "test_persistent or test_direction_mask_table_never_changes_a_frame or test_empty_scene_is_all_sky or "
          "test_planes_only_and_unknown_geometry_option")


@pytest.mark.gpu
def test_ab_build_runs_the_resident_wave_and_streamed_shapes_bit_identically():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "rsraytracingv3_amd", "csrc"), "ab"], check=True, capture_output=True)
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from rsraytracingv3_amd import _native as N; "
                            "print(N.load().rt_has_ab_shapes())" % ROOT], env=dict(os.environ, RT355_LIB=LIB), capture_output=True, text=True)
    assert probe.stdout.strip() == "1", probe.stdout + probe.stderr
    run = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "tests/test_parity_gpu.py",
                          "tests/test_edge_cases_gpu.py", "-k", SELECT], cwd=ROOT, env=dict(os.environ, RT355_LIB=LIB),
                         capture_output=True, text=True, timeout=1500)
    tail = (run.stdout + run.stderr)[-3000:]
    assert run.returncode == 0 and " passed" in run.stdout, tail


def test_shipped_library_leaves_the_ab_shapes_out():
    from rsraytracingv3_amd import _native as N
    if os.environ.get("RT355_LIB"):
        pytest.skip("an explicit library is under test")
    assert N.load().rt_has_ab_shapes() == 0
    syms = subprocess.run(["nm", "-C", "--defined-only", N.LIB_PATH], capture_output=True, text=True).stdout
    assert "render_blocks" in syms and "render_persistent" not in syms and "bounce_stream" not in syms and "primary_stream" not in syms

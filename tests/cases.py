"""Shared synthetic cases for the parity tests (scaled to sizes the
CPU oracle finishes in seconds)."""
from __future__ import annotations

import numpy as np

from rsraytracingv3_amd import synth
from rsraytracingv3_amd.scene import Camera, Geometry, Material, Object, Scene


def spheres_scene(n_spheres: int, reflection_limit: int, seed: int = 42, min_r=3.0, max_r=8.0,
                  placement=50.0, request=None) -> Scene:
    """First ``n_spheres`` placed spheres of the reference-style generator + the ground plane last."""
    req = request if request is not None else max(int(n_spheres * 1.6), n_spheres + 4)
    s = Scene.random_spheres(min_r, max_r, placement, req, seed, reflection_limit)
    spheres = [o for o in s.objects if o.geometry.kind == 0]
    assert len(spheres) >= n_spheres, "only %d spheres placed" % len(spheres)
    s.objects = spheres[:n_spheres] + [s.objects[-1]]
    return s


def config1_scene() -> Scene:
    """BASELINE config 1: 4 spheres (radius U(3,8), disc radius 20) + plane, 1 bounce."""
    s = spheres_scene(4, 1, seed=42, placement=20.0, request=8)
    s.camera = Camera((28.0, 30.0, 28.0), (0.7, -0.8, 0.0), 70.0)
    return s


def headline_scene(reflection_limit: int = 8) -> Scene:
    """BASELINE config 2: 64 spheres + plane, camera (55,65,55) rot (0.8,-0.8,0) fov 70."""
    return spheres_scene(64, reflection_limit, seed=42, placement=50.0, request=100)


def stress_scene(n: int = 1024, reflection_limit: int = 8) -> Scene:
    """BASELINE config 3: 1024 spheres (radius U(1,3), disc radius 150) + plane."""
    return spheres_scene(n, reflection_limit, seed=42, min_r=1.0, max_r=3.0, placement=150.0, request=n + 64)


def horizon_scene(reflection_limit: int = 3) -> Scene:
    """The screenshot's camera (34.4,37.3,105.5) rot (0.5,-0.3,0): sky, horizon and spheres in view."""
    s = headline_scene(reflection_limit)
    s.camera = Camera((34.4, 37.3, 105.5), (0.5, -0.3, 0.0), 70.0)
    return s


def inputs(sky_w=256, sky_h=128, rnd_seed=1):
    return synth.procedural_skybox(sky_w, sky_h), synth.random_texture(rnd_seed)


def reference_default_scene(reflection_limit: int = 3) -> Scene:
    """The reference's start-up scene (scene.rs:81-89, seed 42) regenerated on its own random stream,
    seen from the camera of its screenshot: position as its panel shows, rotation (shown as 0.5 / -0.3, one
    decimal) as fitted inside that rounding interval by tests/golden/make_screenshot_fixture.py."""
    import json
    import os
    s = Scene.random_spheres_default_config()
    s.reflection_limit = reflection_limit
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_screenshot_panel.json")) as f:
        cam = json.load(f)["camera"]
    s.camera = Camera(tuple(cam["position"]), tuple(cam["rotation"]), cam["fov"])
    return s

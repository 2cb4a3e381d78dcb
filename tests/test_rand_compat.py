"""Pins the re-implementation of the reference's random streams (rand 0.8.5 StdRng = ChaCha12,
rand_distr UnitDisc, gen_range) and the reference-exact scene generator built on it.

Golden values come from the reference's own screenshot (screenshots/screenshot.png): its object panel
shows Sphere 0 of the default scene (`Scene::random_spheres_default_config`, seed 42, scene.rs:81-89):
position 2.7 / 3.7 / -25.1, radius 3.7, emission yellow, strength 5.34, metallic 0.617, roughness
0.348 (egui shows 1, 2 or 3 decimals).  The ChaCha core is also checked against RFC 7539 §2.3.2."""
import numpy as np

from rsraytracingv3_amd.rand_compat import StdRng, _chacha_block
from rsraytracingv3_amd.scene import Scene


def test_chacha_block_rfc7539_vector():
    key = [int.from_bytes(bytes(range(4 * i, 4 * i + 4)), "little") for i in range(8)]
    # RFC layout: word 12 = block counter 1, words 13..15 = nonce 00000009 0000004a 00000000
    out = _chacha_block(key, 1 | (0x09000000 << 32), 0x4A000000, 20)
    assert out[:4] == [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3]
    assert out[12:] == [0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]


def test_sphere0_of_the_default_scene_matches_the_reference_screenshot():
    scene = Scene.random_spheres_default_config()
    s0 = scene.objects[0]
    assert s0.name == "Sphere 0" and s0.geometry.kind == 0
    cx, cy, cz = s0.geometry.center
    assert ("%.1f" % cx, "%.1f" % cy, "%.1f" % cz) == ("2.7", "3.7", "-25.1")
    assert "%.1f" % s0.geometry.radius == "3.7" and cy == s0.geometry.radius       # y = radius (scene.rs:111)
    assert s0.material.emission == (1.0, 1.0, 0.0)                                  # palette[2], yellow
    assert "%.2f" % s0.material.emission_strength == "5.34"
    assert "%.3f" % s0.material.metallic == "0.617"
    assert "%.3f" % s0.material.roughness == "0.348"


def test_default_scene_structure():
    scene = Scene.random_spheres_default_config()
    spheres = [o for o in scene.objects if o.geometry.kind == 0]
    assert scene.objects[-1].name == "Plane" and scene.objects[-1].geometry.size == 100_000.0
    assert 60 <= len(spheres) <= 79          # the plane still fits the reference's 80 slots (bytes.rs:75)
    assert scene.reflection_limit == 3 and scene.camera.position == (55.0, 65.0, 55.0)
    for i, a in enumerate(spheres):
        assert 3.0 <= a.geometry.radius < 8.0
        for b in spheres[:i]:
            d = np.linalg.norm(np.subtract(a.geometry.center, b.geometry.center))
            assert d >= a.geometry.radius + b.geometry.radius - 1e-4
    o, l, c = scene.as_bytes(1000, 900)      # the reference's initial render size (app.rs:365)
    assert len(o) == 8960 and o[112 * len(scene.objects):] == bytes(112 * (80 - len(scene.objects)))


def test_stream_primitives():
    r = StdRng.seed_from_u64(42)
    a = [r.next_u32() for _ in range(130)]            # crosses a 64-word refill twice
    r2 = StdRng.seed_from_u64(42)
    b = []
    for _ in range(65):
        v = r2.next_u64()
        b += [v & 0xFFFFFFFF, v >> 32]
    assert a == b                                     # next_u64 = two consecutive words, low first
    r3 = StdRng.seed_from_u64(42)
    [r3.next_u32() for _ in range(63)]
    v = r3.next_u64()                                 # straddles the buffer edge
    assert v == (a[64] << 32 | a[63])
    r4 = StdRng.seed_from_u64(7)
    draws = [r4.gen_range_usize(0, 7) for _ in range(2000)]
    assert set(draws) == set(range(7))
    x = [float(r4.uniform_f32(-1.0, 1.0)) for _ in range(2000)]
    assert -1.0 <= min(x) and max(x) < 1.0 and abs(np.mean(x)) < 0.08
    y = [float(r4.gen_range_f32_inclusive(5.0, 15.0)) for _ in range(2000)]
    assert 5.0 <= min(y) and max(y) <= 15.0

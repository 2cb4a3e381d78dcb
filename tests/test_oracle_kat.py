"""Known-answer tests of the CPU oracle, each derivable by hand from the shader text
(SURVEY.md §8c list (i)-(x)).  The reference ships no tests or golden vectors, so these are
what pins the oracle ("parity unpinned" otherwise)."""
import struct

import numpy as np
import pytest

import oracle
from rsraytracingv3_amd.scene import Geometry, Material, Object, Scene


def sphere(center, radius, colour=(1, 0, 0), emission=(0, 0, 0), strength=0.0, roughness=0.0):
    return Object("s", Material(colour, emission, strength, 0.5, roughness), Geometry.sphere(center, radius))


def plane(center=(0, 0, 0), normal=(0, 1, 0), size=1.0, roughness=0.5):
    return Object("p", Material((0.5, 0.5, 0.5), (0, 0, 0), 0.0, 0.2, roughness), Geometry.plane(center, normal, size))


def blob(objs, slots=None):
    data = b"".join(o.as_bytes() for o in objs)
    n = len(objs)
    if slots:
        data += bytes(112 * (slots - n))
        n = slots
    return data, n


def config(w, h, r, position=(0, 0, 0), forward=(0, 0, -1), right=(1, 0, 0), up=(0, 1, 0), fov=90.0):
    b = bytearray(112)
    struct.pack_into("<3f", b, 0, *position)
    struct.pack_into("<3f", b, 16, *forward)
    struct.pack_into("<3f", b, 32, *right)
    struct.pack_into("<3f", b, 48, *up)
    struct.pack_into("<f", b, 92, fov)
    struct.pack_into("<3I", b, 96, r, w, h)
    return bytes(b)


def sky_ramp(w=16, h=8):
    """Every texel distinct and exactly representable."""
    s = np.zeros((h, w, 4), np.float32)
    s[..., 0] = np.arange(w)[None, :] + 1
    s[..., 1] = np.arange(h)[:, None] + 1
    s[..., 2] = 0.25
    s[..., 3] = 1
    return s


# (i) (ii) sphere ------------------------------------------------------------------------
def test_sphere_head_on():
    b, n = blob([sphere((0, 0, 5), 1.0)])
    idx, t, pos, nrm = oracle.probe_intersect(b, n, (0, 0, 0), (0, 0, 1))
    assert idx == 0 and t == 4.0  # b=-10, c=24, disc=4, t=(10-2)/2
    assert np.array_equal(pos, [0, 0, 4]) and np.array_equal(nrm, [0, 0, -1])


def test_ray_from_inside_sphere_misses():
    b, n = blob([sphere((0, 0, 5), 1.0)])
    idx, t, _, _ = oracle.probe_intersect(b, n, (0, 0, 5), (0, 0, 1))
    assert idx == -1 and t == 1000000.0  # near root negative -> rejected; far root never used (utils.hlsl:12)


def test_sphere_behind_ray_misses():
    b, n = blob([sphere((0, 0, -5), 1.0)])
    assert oracle.probe_intersect(b, n, (0, 0, 0), (0, 0, 1))[0] == -1


def test_nearest_of_two_and_lowest_index_on_tie():
    b, n = blob([sphere((0, 0, 9), 1.0), sphere((0, 0, 5), 1.0)])
    assert oracle.probe_intersect(b, n, (0, 0, 0), (0, 0, 1))[0] == 1
    b, n = blob([sphere((0, 0, 5), 1.0), sphere((0, 0, 5), 1.0)])
    assert oracle.probe_intersect(b, n, (0, 0, 0), (0, 0, 1))[0] == 0  # strict '<' (ray.hlsl:62, D10)


# (iii) plane -----------------------------------------------------------------------------
def test_bounded_plane():
    b, n = blob([plane()])
    idx, t, pos, nrm = oracle.probe_intersect(b, n, (0, 1, 0), (0, -1, 0))
    assert idx == 0 and t == 1.0 and np.array_equal(pos, [0, 0, 0]) and np.array_equal(nrm, [0, 1, 0])
    assert oracle.probe_intersect(b, n, (2, 1, 0), (0, -1, 0))[0] == -1      # outside the box (ray.hlsl:79-85)
    assert oracle.probe_intersect(b, n, (0, 1, 0), (1, 0, 0))[0] == -1       # parallel: |den| < 1e-6
    assert oracle.probe_intersect(b, n, (0, -1, 0), (0, -1, 0))[0] == -1     # behind: t < EPSILON


def test_plane_normal_is_not_renormalised():
    b, n = blob([plane(normal=(0, 2, 0))])
    idx, t, _, nrm = oracle.probe_intersect(b, n, (0, 1, 0), (0, -1, 0))
    assert idx == 0 and t == 1.0 and np.array_equal(nrm, [0, 2, 0])  # ray.hlsl:43-45


# (iv) skybox addressing -------------------------------------------------------------------
def test_hdri_texel_addressing():
    assert oracle.probe_hdri_texel((0, 0, 1), 16, 8) == (8, 4)    # u = v = 0.5
    assert oracle.probe_hdri_texel((0, 1, 0), 16, 8) == (8, 0)    # straight up -> top row (row 0 = top)
    assert oracle.probe_hdri_texel((0, -1, 0), 16, 8) == (8, 7)   # straight down -> clamped last row
    assert oracle.probe_hdri_texel((0, 0, -1), 16, 8) == (15, 4)  # u = 1.0 -> clamp to W-1
    assert oracle.probe_hdri_texel((1, 0, 0), 16, 8) == (12, 4)   # u = 0.75
    assert oracle.probe_hdri_texel((-1, 0, 0), 16, 8) == (4, 4)   # u = 0.25


def test_miss_returns_sky_texel_and_ends_after_one_segment():
    sky = sky_ramp()
    f = oracle.render(b"", 0, config(1, 1, 5), (0, 0), 0, sky, np.zeros((4, 4), np.float32))
    assert f.segments[0, 0] == 1 and f.hits[0, 0, 0] == -1 and (f.hits[0, 0, 1:] == -2).all()
    assert np.array_equal(f.rgbf32[0, 0], sky[4, 8, :3])  # centre ray = (0,0,1)


# (v) (vi) shading ------------------------------------------------------------------------
def test_emissive_sphere_head_on():
    b, n = blob([sphere((0, 0, 5), 1.0, emission=(1, 1, 0), strength=5.0)])
    f = oracle.render(b, n, config(1, 1, 1), (0, 0), 0, sky_ramp(), np.zeros((4, 4), np.float32))
    assert np.array_equal(f.rgbf32[0, 0], [5, 5, 0]) and f.segments[0, 0] == 1
    assert tuple(f.rgba8[0, 0]) == (255, 255, 0, 255)  # clamp + alpha 1 (frag.hlsl:65)


def test_mirror_sphere_two_segments():
    sky = sky_ramp()
    b, n = blob([sphere((0, 0, 5), 1.0, colour=(0.25, 0.5, 1.0))])
    f = oracle.render(b, n, config(1, 1, 2), (0, 0), 0, sky, np.zeros((4, 4), np.float32))
    # reflect((0,0,1), (0,0,-1)) = (0,0,-1); cos = 1; energy = 2*albedo; the bounce sees the texel
    # along -z: u = 1.0 -> column W-1, v = 0.5
    assert f.segments[0, 0] == 2 and list(f.hits[0, 0]) == [0, -1]
    assert np.array_equal(f.rgbf32[0, 0], np.array([0.5, 1.0, 2.0], np.float32) * sky[4, 15, :3])


def test_energy_is_read_before_shade_updates_it():
    """App. B D1: with the order reversed a miss would contribute energy(=0)*sky = 0."""
    f = oracle.render(b"", 0, config(1, 1, 3), (0, 0), 0, sky_ramp(), np.zeros((4, 4), np.float32))
    assert f.rgbf32[0, 0, 0] > 0


# (vii) create_ray ------------------------------------------------------------------------
def test_create_ray_fov90():
    d = oracle.probe_create_ray(config(2, 2, 1), 0.25, 0.25)  # pixel (0,0) of a 2x2 target, jitter 0
    assert np.allclose(d, np.array([-0.5, 0.5, 1.0]) / np.sqrt(1.5), atol=2e-7)
    d = oracle.probe_create_ray(config(2, 2, 1), 0.5, 0.5)
    assert np.array_equal(d, [0, 0, 1])  # rays leave along -forward (D9)
    d = oracle.probe_create_ray(config(4, 2, 1), 0.5, 0.0)  # aspect: half_height = half_width*H/W
    assert np.allclose(d, np.array([0, 0.5, 1.0]) / np.sqrt(1.25), atol=2e-7)


# (viii) random() --------------------------------------------------------------------------
def test_random_on_constant_texture():
    tex = np.full((600, 600), 0.5, np.float32)
    vals, seeds = oracle.probe_random(tex, 0.3, 0.7, 4)
    assert (vals == 0.5).all()
    seed, exp = np.float32(42.0), []
    for _ in range(4):  # seed = fmod(seed + x*532.3412, 600); 0.5*c is exact so fma == mul+add
        seed = np.float32(np.fmod(np.float32(np.float32(0.5) * np.float32(532.3412) + seed), np.float32(600)))
        exp.append(seed)
    assert np.array_equal(seeds, np.array(exp, np.float32))
    assert np.allclose(seeds, [308.1706, 574.3412, 240.5118, 506.6824], atol=1e-3)


def test_random_addresses_the_expected_texel():
    tex = np.arange(600 * 600, dtype=np.float32).reshape(600, 600) / np.float32(600 * 600)
    vals, _ = oracle.probe_random(tex, 0.25, 0.5, 1)
    # coord = frac((0.25, 0.5) + 42) -> texel (150, 300): both components add the same seed
    assert vals[0] == tex[300, 150]


def test_roughness_zero_returns_mirror_direction_exactly():
    tex = np.full((8, 8), 0.3, np.float32)
    n = np.array([0.6, 0.0, -0.8], np.float32)
    assert np.array_equal(oracle.probe_hemisphere(tex, 0.1, 0.2, n, 0.0), n)  # random.hlsl:26-28


def test_hemisphere_lobe_is_around_the_mirror_direction():
    rs = np.random.RandomState(5)
    tex = rs.uniform(0, 1, (64, 64)).astype(np.float32)
    n = np.array([0.0, 1.0, 0.0], np.float32)
    dots = []
    for k in range(200):
        d = oracle.probe_hemisphere(tex, 0.013 * k, 0.007 * k, n, 0.3)
        assert abs(np.linalg.norm(d) - 1) < 1e-5
        dots.append(float(d @ n))
    assert min(dots) > 0 and np.mean(dots) > 0.8  # D2: tangent*x + binormal*y + normal*z


# (ix) zero-padded slots ---------------------------------------------------------------------
def test_zero_padded_slot_is_a_radius0_sphere_at_the_origin():
    b, n = blob([sphere((0, 0, 5), 1.0)], slots=80)
    assert oracle.probe_intersect(b, n, (0, 0, 0.5), (0, 0, 1))[0] == 0   # real sphere wins
    assert oracle.probe_intersect(b, n, (3, 0, -5), (0, 0, 1))[0] == -1   # padding is not hit
    # D5: only a ray through the world origin itself "hits" a padded slot (disc == 0), lowest index first
    idx, t, _, _ = oracle.probe_intersect(b, n, (0, 0, -5), (0, 0, 1))
    assert idx == 1 and t == 5.0


# (x) sRGB8 + accumulation -------------------------------------------------------------------
def test_srgb8_table():
    assert oracle.srgb8_encode(0.0) == 0 and oracle.srgb8_encode(1.0) == 255
    assert oracle.srgb8_encode(2.0) == 255 and oracle.srgb8_encode(-1.0) == 0
    assert oracle.srgb8_encode(float("nan")) == 0
    assert oracle.srgb8_encode(0.0031308) == 10 and oracle.srgb8_encode(0.5) == 188
    assert oracle.srgb8_decode(0) == 0.0 and oracle.srgb8_decode(255) == 1.0
    for b in range(256):
        assert oracle.srgb8_encode(oracle.srgb8_decode(b)) == b
        s = b / 255.0
        ref = s / 12.92 if s <= 0.04045 else ((s + 0.055) / 1.055) ** 2.4
        assert abs(oracle.srgb8_decode(b) - ref) <= 1e-6 * max(ref, 1e-3) + 1e-9  # pow = exp2(y*log2 x): a few ulp


@pytest.mark.parametrize("pc,w_new", [(0, 1.0), (1, 0.5), (2, 1.0 / 3.0)])
def test_progressive_mix(pc, w_new):
    b, n = blob([sphere((0, 0, 5), 1.0, emission=(1, 1, 1), strength=0.5)])
    hist = np.zeros((1, 1, 4), np.uint8)
    hist[0, 0] = (188, 0, 255, 255)  # linear ~ (0.5, 0, 1)
    f = oracle.render(b, n, config(1, 1, 1), (0, 0), pc, sky_ramp(), np.zeros((4, 4), np.float32),
                      history_rgba8=hist)
    prev = np.array([oracle.srgb8_decode(188), 0.0, 1.0])
    assert np.allclose(f.rgbf32[0, 0], 0.5 * w_new + prev * (1 - w_new), atol=1e-6)  # frag.hlsl:61-62


def test_history_is_the_same_pixel_whatever_the_jitter():
    """D13: the history is sampled at the un-jittered coord -> always this pixel's own texel."""
    scene = Scene.random_spheres(3, 8, 20, 6, 1, 2)
    sky, rnd = sky_ramp(32, 16), np.random.RandomState(1).uniform(0, 1, (600, 600)).astype(np.float32)
    w, h = 24, 16
    o, _, c = scene.as_bytes(w, h)
    hist = np.random.RandomState(2).randint(0, 256, (h, w, 4)).astype(np.uint8)
    a = oracle.render(o, 80, c, (0.49, -0.49), 1, sky, rnd, history_rgba8=hist)
    z = oracle.render(o, 80, c, (0.49, -0.49), 1, sky, rnd, history_rgba8=np.zeros_like(hist))
    lut = np.array([oracle.srgb8_decode(b) for b in range(256)], np.float32)
    assert np.allclose(a.rgbf32 - z.rgbf32, 0.5 * lut[hist[..., :3]], atol=1e-5)

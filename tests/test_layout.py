"""Byte-layout known-answer tests: the wire formats of SURVEY.md App. A, derived line by
line from the Rust writers (objects.rs:41-57,125-157,202-214,244-277; scene.rs:252-271;
connection.rs:30-40), checked on the host serialisers, the C header and the oracle's parser."""
import os
import struct
import subprocess
import tempfile

import numpy as np

import oracle
from rsraytracingv3_amd.bytes import bytes_concat, bytes_concat_owned, show_bytes
from rsraytracingv3_amd.connection import FrameData
from rsraytracingv3_amd.scene import Camera, Geometry, Light, Material, Object, Scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def f32(b, off, n=1):
    return struct.unpack_from("<%df" % n, b, off)


def test_material_48_bytes():
    b = Material((0.1, 0.2, 0.3), (0.4, 0.5, 0.6), 7.0, 0.8, 0.9).as_bytes()
    assert len(b) == 48
    assert np.allclose(f32(b, 0, 3), (0.1, 0.2, 0.3)) and b[12:16] == bytes(4)
    assert np.allclose(f32(b, 16, 3), (0.4, 0.5, 0.6))
    assert f32(b, 28) == (7.0,) and np.allclose(f32(b, 32, 2), (0.8, 0.9)) and b[40:48] == bytes(8)


def test_sphere_and_plane_geometry_64_bytes():
    s = Geometry.sphere((1.0, 2.0, 3.0), 4.0).as_bytes()
    assert len(s) == 64 and struct.unpack_from("<I", s, 0) == (0,) and s[4:16] == bytes(12)
    assert f32(s, 16, 3) == (1.0, 2.0, 3.0) and s[28:44] == bytes(16) and f32(s, 44) == (4.0,)
    assert s[48:] == bytes(16)
    p = Geometry.plane((1.0, 2.0, 3.0), (0.0, 1.0, 0.0), 9.0).as_bytes()
    assert struct.unpack_from("<I", p, 0) == (1,) and f32(p, 16, 3) == (1.0, 2.0, 3.0) and p[28:32] == bytes(4)
    assert f32(p, 32, 3) == (0.0, 1.0, 0.0) and f32(p, 44) == (9.0,)


def test_object_112_bytes_offsets():
    o = Object("o", Material((0.1, 0.2, 0.3), (1, 0, 1), 5.0, 0.25, 0.75), Geometry.sphere((7, 8, 9), 2.5)).as_bytes()
    assert len(o) == 112 == Object.BUFFER_SIZE
    assert f32(o, 28) == (5.0,) and f32(o, 36) == (0.75,)          # emission_strength, roughness
    assert struct.unpack_from("<I", o, 48) == (0,)                 # geometry.option
    assert f32(o, 64, 3) == (7.0, 8.0, 9.0) and f32(o, 92) == (2.5,)  # center, f32_data
    pl = Object("p", Material(), Geometry.plane((0, 0, 0), (0, 1, 0), 1e5)).as_bytes()
    assert struct.unpack_from("<I", pl, 48) == (1,) and f32(pl, 80, 3) == (0.0, 1.0, 0.0) and f32(pl, 92) == (1e5,)


def test_light_48_bytes():
    b = Light(1, (0.4, 0.4, 0.4), (0.0, 2.0, 0.0)).as_bytes()
    assert len(b) == 48 and struct.unpack_from("<I", b, 0) == (1,)
    assert np.allclose(f32(b, 16, 3), (0.4, 0.4, 0.4)) and f32(b, 32, 3) == (0.0, 2.0, 0.0)


def test_config_112_bytes_offsets():
    s = Scene(camera=Camera((1.0, 2.0, 3.0), (0.0, 0.0, 0.0), 55.0), reflection_limit=6)
    c = s.config_bytes(640, 360)
    assert len(c) == 112
    assert f32(c, 0, 3) == (1.0, 2.0, 3.0)
    assert np.allclose(f32(c, 16, 3), (0, 0, 1)) and np.allclose(f32(c, 32, 3), (1, 0, 0)) and np.allclose(f32(c, 48, 3), (0, 1, 0))
    assert np.allclose(f32(c, 64, 3), (0.5, 0.8, 1.0)) and np.allclose(f32(c, 80, 3), (0.2, 0.2, 0.2))
    assert f32(c, 92) == (55.0,) and struct.unpack_from("<3I", c, 96) == (6, 640, 360) and c[108:] == bytes(4)


def test_framedata_16_bytes():
    fd = FrameData((0.25, -0.125))
    fd.progressive_count = 7
    b = fd.as_bytes()
    assert len(b) == 16 and f32(b, 0, 2) == (0.25, -0.125) and struct.unpack_from("<I", b, 8) == (7,) and b[12:] == bytes(4)


def test_scene_buffers_pad_to_80_slots_and_truncate():
    s = Scene.random_spheres(3, 8, 50, 10, 1)
    o, l, c = s.as_bytes(100, 50)
    assert (len(o), len(l), len(c)) == Scene.BUFFER_SIZE == (8960, 96, 112)
    n = len(s.objects)
    assert o[: 112 * n] == s.live_object_bytes() and o[112 * n:] == bytes(112 * (80 - n))
    big = Scene.random_spheres(1, 2, 200, 90, 1)   # > 79 spheres: the plane (pushed last) is dropped (bytes.rs:75)
    assert len(big.objects) == 91
    ob = big.as_bytes(10, 10)[0]
    assert len(ob) == 8960 and ob == b"".join(x.as_bytes() for x in big.objects[:80])
    assert len(big.as_bytes(10, 10, max_objects=128)[0]) == 128 * 112


def test_bytes_concat_semantics():
    assert bytes_concat(6, [b"ab", b"cde"]) == b"abcde\0"
    assert bytes_concat(4, [b"ab", b"cde", b"zz"]) == b"abcd"      # truncation (bytes.rs:38-41)
    assert bytes_concat_owned(2, 6, [b"ab", b"cd"]) == b"abcd\0\0"
    assert bytes_concat_owned(2, 4, [b"ab", b"cd", b"ef"]) == b"abcd"  # take(N/M) (bytes.rs:75)
    try:
        bytes_concat_owned(3, 4, [])
        assert False
    except AssertionError:
        pass  # assert_eq!(N % M, 0) (bytes.rs:71)
    assert show_bytes(b"\x01\xff").strip().split() == ["01", "ff"]


def test_camera_basis_matches_nalgebra_convention():
    f, r, u = Camera((0, 0, 0), (0.0, 0.0, 0.0), 70).get_vectors_fru()
    assert np.allclose(f, (0, 0, 1)) and np.allclose(r, (1, 0, 0)) and np.allclose(u, (0, 1, 0))
    f, r, u = Camera((0, 0, 0), (0.8, -0.8, 0.0), 70).get_vectors_fru()
    # forward = Ry(0.8)·Rx(-0.8)·z ; orthonormal right-handed-by-construction basis
    assert np.allclose(f, (np.sin(0.8) * np.cos(0.8), np.sin(0.8), np.cos(0.8) ** 2), atol=1e-6)
    assert abs(np.dot(f, r)) < 1e-6 and abs(np.dot(f, u)) < 1e-6 and abs(np.dot(r, u)) < 1e-6
    assert np.allclose(np.cross(f, r), u, atol=1e-6)


def test_c_header_compiles_as_c_and_asserts_hold():
    """rt355_layout.h carries _Static_asserts on every offset; compile it as C11 and C++17."""
    src = '#include "rt355.h"\nint main(void){return (int)sizeof(rt_wire_object)-112;}\n'
    with tempfile.TemporaryDirectory() as d:
        for comp, name, std in (("gcc", "t.c", "-std=c11"), ("g++", "t.cpp", "-std=c++17")):
            path = os.path.join(d, name)
            with open(path, "w") as f:
                f.write(src)
            exe = os.path.join(d, name + ".out")
            subprocess.run([comp, std, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), path, "-o", exe], check=True)
            assert subprocess.run([exe]).returncode == 0


def test_oracle_parses_the_same_offsets():
    """A sphere placed through the host serialiser is hit where the bytes say it is."""
    o = Object("o", Material(), Geometry.sphere((0.0, 0.0, 10.0), 2.0)).as_bytes()
    idx, t, _, _ = oracle.probe_intersect(o, 1, (0, 0, 0), (0, 0, 1))
    assert idx == 0 and t == 8.0

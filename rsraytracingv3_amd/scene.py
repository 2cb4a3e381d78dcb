"""Scene model and HLSL-layout serialisers — host-side mirror of the reference's
``src/ray_tracer/{objects,scene,camera}.rs``.

Only what feeds the hot path is mirrored: the byte layouts (``as_bytes``), the
size constants, the camera basis and two scene generators with the distributions of
``Scene::random_spheres`` (scene.rs:95-231): ``Scene.random_spheres`` on a documented
SplitMix64 stream (``synth.SplitMix64``; the benchmark scenes), and
``Scene.random_spheres_reference`` on the reference's own stream (``rand``'s ChaCha12
``StdRng``, ``rand_compat``), which regenerates the reference's seed-42 start-up scene
bit for bit (pinned by the values its screenshot shows).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from .bytes import bytes_concat, bytes_concat_owned, f32_bytes, u32_bytes
from .synth import SplitMix64

Vec3 = Tuple[float, float, float]


@dataclass
class Material:
    """objects.rs:6-22; ``BUFFER_SIZE`` objects.rs:38."""
    colour: Vec3 = (1.0, 0.0, 0.0)
    emission: Vec3 = (0.0, 0.0, 0.0)
    emission_strength: float = 0.0
    metallic: float = 0.5
    roughness: float = 0.5

    BUFFER_SIZE = 48

    def as_bytes(self) -> bytes:
        """objects.rs:41-57."""
        return bytes_concat(self.BUFFER_SIZE, [
            f32_bytes(*self.colour),
            bytes(4),
            f32_bytes(*self.emission),
            f32_bytes(self.emission_strength),
            f32_bytes(self.metallic),
            f32_bytes(self.roughness),
        ])


@dataclass
class Geometry:
    """objects.rs:59-98 — ``kind`` 0 = Sphere{center, radius}, 1 = Plane{center, normal, size}."""
    kind: int
    center: Vec3
    radius: float = 1.0                 # sphere
    normal: Vec3 = (0.0, 1.0, 0.0)      # plane
    size: float = 1.0                   # plane half extent

    BUFFER_SIZE = 64

    @staticmethod
    def sphere(center: Vec3, radius: float) -> "Geometry":
        return Geometry(0, center, radius=radius)

    @staticmethod
    def plane(center: Vec3, normal: Vec3, size: float) -> "Geometry":
        return Geometry(1, center, normal=normal, size=size)

    def as_bytes(self) -> bytes:
        """objects.rs:125-157."""
        if self.kind == 0:
            return bytes_concat(self.BUFFER_SIZE, [
                u32_bytes(0), bytes(12), f32_bytes(*self.center), bytes(16), f32_bytes(self.radius)])
        return bytes_concat(self.BUFFER_SIZE, [
            u32_bytes(1), bytes(12), f32_bytes(*self.center), bytes(4), f32_bytes(*self.normal),
            f32_bytes(self.size)])


@dataclass
class Object:
    """objects.rs:160-214 (id/uuid omitted: never serialised)."""
    name: str
    material: Material
    geometry: Geometry

    BUFFER_SIZE = Material.BUFFER_SIZE + Geometry.BUFFER_SIZE  # 112

    def as_bytes(self) -> bytes:
        """objects.rs:202-214."""
        return bytes_concat(self.BUFFER_SIZE, [self.material.as_bytes(), self.geometry.as_bytes()])


@dataclass
class Light:
    """objects.rs:221-277 — ``kind`` 0 = Direction, 1 = Point.  Uploaded, never read by the shader."""
    kind: int
    intensity: Vec3
    vector: Vec3

    BUFFER_SIZE = 48

    def as_bytes(self) -> bytes:
        """objects.rs:244-277."""
        return bytes_concat(self.BUFFER_SIZE, [
            u32_bytes(self.kind), bytes(12), f32_bytes(*self.intensity), bytes(4), f32_bytes(*self.vector)])


def _f32(x) -> np.float32:
    return np.float32(x)


@dataclass
class Camera:
    """camera.rs:6-16."""
    position: Vec3 = (0.0, 0.0, -5.0)
    rotation: Vec3 = (0.0, 0.0, 0.0)
    fov: float = 70.0

    def get_vectors_fru(self) -> Tuple[Vec3, Vec3, Vec3]:
        """camera.rs:55-66: forward = R(-rx, -ry, 0)·(0,0,1) with nalgebra's
        ``from_euler_angles(roll, pitch, yaw)`` = Rz(yaw)·Ry(pitch)·Rx(roll);
        right = normalize(temp × forward), temp = Rz(-rz)·(0,1,0);
        up = normalize(forward × right).  Evaluated in float32."""
        rx, ry, rz = (_f32(v) for v in self.rotation)
        a, b = -rx, -ry
        sa, ca = _f32(math.sin(a)), _f32(math.cos(a))
        sb, cb = _f32(math.sin(b)), _f32(math.cos(b))
        # Rx(a)·(0,0,1) = (0, -sin a, cos a); Ry(b)·(x,y,z) = (cb x + sb z, y, -sb x + cb z)
        fwd = np.array([sb * ca, -sa, cb * ca], dtype=np.float32)
        temp = np.array([_f32(math.sin(rz)), _f32(math.cos(rz)), 0.0], dtype=np.float32)
        right = np.cross(temp, fwd).astype(np.float32)
        right = (right / np.float32(np.linalg.norm(right))).astype(np.float32)
        up = np.cross(fwd, right).astype(np.float32)
        up = (up / np.float32(np.linalg.norm(up))).astype(np.float32)
        return tuple(map(float, fwd)), tuple(map(float, right)), tuple(map(float, up))


_EMISSION_PALETTE: List[Vec3] = [  # scene.rs:121-129
    (1., 1., 1.), (1., 0., 0.), (1., 1., 0.), (0., 1., 0.), (0., 1., 1.), (0., 0., 1.), (1., 0., 1.)]


@dataclass
class Scene:
    """scene.rs:8-23."""
    objects: List[Object] = field(default_factory=list)
    lights: List[Light] = field(default_factory=list)
    camera: Camera = field(default_factory=Camera)
    background_colour: Vec3 = (0.5, 0.8, 1.0)
    ambient_light: Vec3 = (0.2, 0.2, 0.2)
    reflection_limit: int = 3
    do_objects_spin: bool = False

    MAX_OBJECTS = 80     # scene.rs:28
    MAX_LIGHTS = 2       # scene.rs:31
    CONFIG_SIZE = 112    # scene.rs:33
    BUFFER_SIZE = (Object.BUFFER_SIZE * MAX_OBJECTS, Light.BUFFER_SIZE * MAX_LIGHTS, CONFIG_SIZE)

    # -- serialisation ------------------------------------------------------
    def config_bytes(self, width: int, height: int) -> bytes:
        """The Config blob of scene.rs:252-271."""
        fwd, right, up = self.camera.get_vectors_fru()
        return bytes_concat(self.CONFIG_SIZE, [
            f32_bytes(*self.camera.position), bytes(4),
            f32_bytes(*fwd), bytes(4),
            f32_bytes(*right), bytes(4),
            f32_bytes(*up), bytes(4),
            f32_bytes(*self.background_colour), bytes(4),
            f32_bytes(*self.ambient_light),
            f32_bytes(self.camera.fov),
            u32_bytes(self.reflection_limit),
            u32_bytes(width),
            u32_bytes(height),
        ])

    def as_bytes(self, width: int, height: int, max_objects: Optional[int] = None):
        """scene.rs:236-273 → (objects, lights, config).  ``max_objects`` None keeps
        the reference's 80-slot zero-padded buffer (objects past 80 dropped,
        bytes.rs:75); an integer lifts the cap to that many slots (BASELINE
        config 3 needs 1025)."""
        slots = self.MAX_OBJECTS if max_objects is None else max_objects
        return (
            bytes_concat_owned(Object.BUFFER_SIZE, Object.BUFFER_SIZE * slots,
                               (o.as_bytes() for o in self.objects)),
            bytes_concat_owned(Light.BUFFER_SIZE, self.BUFFER_SIZE[1], (l.as_bytes() for l in self.lights)),
            self.config_bytes(width, height),
        )

    def live_object_bytes(self) -> bytes:
        """Exactly len(objects) records, no padding (explicit-count mode of the C-ABI)."""
        return b"".join(o.as_bytes() for o in self.objects)

    # -- generators ---------------------------------------------------------
    @staticmethod
    def simple() -> "Scene":
        """scene.rs:45-75."""
        return Scene(
            objects=[Object("Sphere", Material((1., 0., 0.), (0., 0., 0.), 0., 1., 0.5),
                            Geometry.sphere((0., 0., 0.), 1.))],
            lights=[Light(1, (1., 1., 1.), (0., 2., 0.))],
            camera=Camera((0., 0., -5.), (0., 0., 0.), 70.),
            reflection_limit=4)

    @staticmethod
    def random_sphere(rng: SplitMix64, name: str, min_radius: float, max_radius: float,
                      placement_radius: float, existing: List[Object]) -> Optional[Object]:
        """scene.rs:95-148 — up to 100 rejection tries; radius U(min,max); (x,z) uniform
        on the unit disc × placement radius; y = radius; reject on overlap with any
        existing sphere (scene.rs:171-187)."""
        for _ in range(100):
            radius = float(_f32(rng.f32()) * _f32(max_radius - min_radius) + _f32(min_radius))
            while True:  # rand_distr::UnitDisc: rejection from the square
                x = 2.0 * rng.f32() - 1.0
                y = 2.0 * rng.f32() - 1.0
                if x * x + y * y <= 1.0:
                    break
            center = (float(_f32(x * placement_radius)), radius, float(_f32(y * placement_radius)))
            ok = True
            for other in existing:
                if other.geometry.kind != 0:
                    continue
                oc = other.geometry.center
                d = math.sqrt(sum((oc[k] - center[k]) ** 2 for k in range(3)))
                if d < radius + other.geometry.radius:
                    ok = False
                    break
            if not ok:
                continue
            colour = (rng.f32(), rng.f32(), rng.f32())
            emission = _EMISSION_PALETTE[rng.below(7)]
            strength = (5.0 + 10.0 * rng.f32()) if rng.f32() > 0.85 else 0.0
            metallic = rng.f32()
            roughness = 0.0 if rng.f32() < 0.2 else rng.f32()
            return Object(name, Material(colour, emission, strength, metallic, roughness),
                          Geometry.sphere(center, radius))
        return None

    @staticmethod
    def random_spheres(min_radius: float, max_radius: float, placement_radius: float,
                       sphere_count: int, seed: int = 42, reflection_limit: int = 3) -> "Scene":
        """scene.rs:152-231 — spheres, then the ground plane pushed LAST (scene.rs:193-207),
        camera (55,65,55) rot (0.8,-0.8,0) fov 70 (scene.rs:210-214)."""
        rng = SplitMix64(seed)
        objects: List[Object] = []
        for i in range(sphere_count):
            o = Scene.random_sphere(rng, "Sphere %d" % i, min_radius, max_radius, placement_radius, objects)
            if o is not None:
                objects.append(o)
        objects.append(Object("Plane", Material((0.5, 0.5, 0.5), (0., 0., 0.), 0., 0.2, 0.5),
                              Geometry.plane((0., 0., 0.), (0., 1., 0.), 100_000.)))
        return Scene(
            objects=objects,
            lights=[Light(0, (0.4, 0.4, 0.4), tuple(v / math.sqrt(3.5) for v in (-1., -1.5, -0.5))),
                    Light(1, (0.4, 0.4, 0.4), (0., 2., 0.))],
            camera=Camera((55., 65., 55.), (0.8, -0.8, 0.), 70.),
            reflection_limit=reflection_limit)

    @staticmethod
    def random_spheres_reference(min_radius: float = 3.0, max_radius: float = 8.0, placement_radius: float = 50.0,
                                 sphere_count: int = 100, seed: int = 42) -> "Scene":
        """scene.rs:152-231 on the reference's own random stream (``StdRng::seed_from_u64`` = ChaCha12,
        ``rand_compat.StdRng``): reproduces the reference's default scene bit for bit — Sphere 0 matches
        the values its screenshot shows.  float32 arithmetic follows the Rust expressions."""
        from .rand_compat import StdRng, fma_f32
        f = np.float32
        rng = StdRng.seed_from_u64(seed)
        objects: List[Object] = []
        for i in range(sphere_count):
            placed = None
            for _ in range(100):                                   # scene.rs:104
                radius = fma_f32(rng.gen_f32(), f(max_radius) - f(min_radius), f(min_radius))  # mul_add: ONE rounding (scene.rs:105-107)
                x, y = rng.unit_disc()                             # scene.rs:108
                cx, cz = f(x * f(placement_radius)), f(y * f(placement_radius))
                ok = True
                for other in objects:                              # scene.rs:171-187
                    oc = other.geometry.center
                    dx, dy, dz = f(f(oc[0]) - cx), f(f(oc[1]) - radius), f(f(oc[2]) - cz)
                    dist = f(np.sqrt(f(f(f(dx * dx) + f(dy * dy)) + f(dz * dz))))
                    if dist < f(radius + f(other.geometry.radius)):
                        ok = False
                        break
                if not ok:
                    continue
                colour = (float(rng.gen_f32()), float(rng.gen_f32()), float(rng.gen_f32()))
                emission = _EMISSION_PALETTE[rng.gen_range_usize(0, 7)]
                strength = float(rng.gen_range_f32_inclusive(5.0, 15.0)) if rng.gen_f32() > f(0.85) else 0.0
                metallic = float(rng.gen_f32())
                roughness = 0.0 if rng.gen_f32() < f(0.2) else float(rng.gen_f32())
                placed = Object("Sphere %d" % i, Material(colour, emission, strength, metallic, roughness),
                                Geometry.sphere((float(cx), float(radius), float(cz)), float(radius)))
                break
            if placed is not None:
                objects.append(placed)
        objects.append(Object("Plane", Material((0.5, 0.5, 0.5), (0., 0., 0.), 0., 0.2, 0.5),
                              Geometry.plane((0., 0., 0.), (0., 1., 0.), 100_000.)))
        return Scene(
            objects=objects,
            lights=[Light(0, (0.4, 0.4, 0.4), tuple(v / math.sqrt(3.5) for v in (-1., -1.5, -0.5))),
                    Light(1, (0.4, 0.4, 0.4), (0., 2., 0.))],
            camera=Camera((55., 65., 55.), (0.8, -0.8, 0.), 70.),
            reflection_limit=3)

    @staticmethod
    def random_spheres_default_config() -> "Scene":
        """scene.rs:81-89 on a machine with more than two cores: the reference's start-up scene."""
        return Scene.random_spheres_reference(3., 8., 50., 100, 42)

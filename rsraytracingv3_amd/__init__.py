"""rt355 — MI355X-native per-pixel ray-tracing path (drop-in for the fragment-shader
path of arlohb/RSRayTracingV3).  The render path is ``librt355.so`` (HIP, gfx950) behind
the C-ABI of ``include/rt355.h``; this package is its host-side mirror of the reference's
``src/gpu`` + ``src/ray_tracer`` serialisers.  Importing the package does not load the
native library; using ``Device`` does, and fails loudly if it is not built."""
from .bytes import bytes_concat, bytes_concat_owned  # noqa: F401
from .scene import Camera, Geometry, Light, Material, Object, Scene  # noqa: F401
from .connection import Connection, Device, FrameData, MultiDevice, RandomTexture, RenderTarget, render_scene  # noqa: F401

__all__ = ["Camera", "Geometry", "Light", "Material", "Object", "Scene", "Connection", "Device", "MultiDevice", "FrameData",
           "RandomTexture", "RenderTarget", "render_scene", "bytes_concat", "bytes_concat_owned"]

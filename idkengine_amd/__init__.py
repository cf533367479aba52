"""idkengine_amd — MI355X-native (gfx950) wavefront path tracer behind IDKEngine's PathTracer API.

    from idkengine_amd import PathTracer, NativeBuilder, scenes
"""
from .pathtracer import PathTracer, IdkPtError  # noqa: F401
from .bvh import NativeBuilder  # noqa: F401
from . import scenes, gputypes  # noqa: F401

__all__ = ["PathTracer", "IdkPtError", "NativeBuilder", "scenes", "gputypes"]

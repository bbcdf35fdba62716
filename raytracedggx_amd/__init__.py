"""raytracedggx_amd -- MI355X (gfx950) implementation of the RayTracedGGX hot path.

  capi  ctypes binding of librtggx.so (hand-written HIP behind the C ABI of include/rtggx.h)
  app   the C++ host layer (RayTracedGGX / RayTracer / Denoiser classes, reference CLI) for Python drivers

Nothing here computes on the CPU: without librtggx.so and a HIP device the package raises.
"""
from . import capi  # noqa: F401

__all__ = ["capi", "app"]

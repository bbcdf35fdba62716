"""raytracedggx_amd -- MI355X (gfx950) implementation of the RayTracedGGX hot path.

  capi  ctypes binding of librtggx.so (hand-written HIP behind the C ABI of include/rtggx.h)
  app   the C++ host layer (RayTracedGGX / RayTracer / Denoiser classes, reference CLI) for Python drivers

Nothing here computes on the CPU: without librtggx.so and a HIP device the package raises.
"""
import os as _os

# The frame runs on four HIP streams, a fifth with the strip exchange of several GPUs, and RCCL brings its own: HIP's default of four
# hardware queues makes the fifth stream share a queue with another one, which serialises what was meant to overlap (a 1920 x 135 strip:
# 0.077 ms per frame with eight queues, 0.10 - 0.14 ms with four; profiles/r03_h_strip_projection.txt).  The runtime reads the variable when
# it initialises, which importing torch does not do; a value the caller has set wins.  host/Main.cpp does the same for the executable.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import capi  # noqa: E402,F401

__all__ = ["capi", "app"]

"""raytracer_amd -- MI355X (gfx950) ray-cast / BVH-traversal path behind a C ABI.

The package is a thin ctypes binding of two in-tree shared libraries:

* ``librtg.so``      -- include/rtg.h: the HIP kernels and their C ABI
* ``librt_host.so``  -- host-side scene / BVH build code above that ABI

There is no CPU fallback: importing :mod:`raytracer_amd.rtg` fails loudly when a
library is missing, and every compute call fails when no HIP device is present.
"""
from . import rtg  # noqa: F401

__all__ = ["rtg"]

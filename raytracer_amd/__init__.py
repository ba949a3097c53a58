"""raytracer_amd -- MI355X (gfx950) backend for the Luma per-pixel render loop.

The product is the C-ABI shared library libluma_hip.so (include/luma_hip.h, sources in
raytracer_amd/csrc/).  `abi` is its ctypes binding; `bands` holds the scanline-band
partition used when a frame is split over several GPUs.
"""
from . import abi  # noqa: F401

"""rtabmap_amd -- MI355X-native loop-closure detection engine behind RTAB-Map's VWDictionary / computeLikelihood.

The product is the C-ABI shared library rtabmap_amd/liblcd_hip.so (include/lcd.h), hand-written HIP for gfx950.
This package holds its sources (csrc/), the build script, a ctypes binding and the host-side mirror of the
reference's VWDictionary interface used by the parity tests.  Nothing here computes on the CPU.
"""
from . import synth  # noqa: F401
from .capi import Engine, LcdError, load, library_path  # noqa: F401

"""aircraftoptimalcontrol_amd — batched Newton/LQR trajectory optimiser for the planar aircraft of
MohamedAtwan/AirCraftOptimalControl, hand-written HIP for MI355X (gfx950) behind a C-ABI.

Layout of the package (only what the hot path needs):
  csrc/      HIP kernels + the C-ABI (include/aoc.h)          -> lib/libaoc_hip.so
  _lib.py    ctypes binding of the C-ABI (fails loudly when the library or the GPU is missing)
  batch.py   batched host API (BatchProblem, NewtonBatchSolver, ...)
  problems.py  the reference drivers' problem set-ups (weights, reference curves) as data
  dropin/    modules named like the reference's (optcon, aircraft_simplified, lqr_tracking) that keep
             its call surface and run on the HIP library
"""
from ._lib import AocError, build_library, library_path  # noqa: F401

__all__ = ["AocError", "build_library", "library_path"]

"""MI355X-native (gfx950, hand-written HIP) CFEAR radar-odometry hot path.

The product is libcfear_hip.so (C ABI: include/cfear_hip.h). This Python package is glue for
tests and benchmarks only: it builds the library, binds the C ABI with ctypes and mirrors the
reference's radarDriver / MapPointNormal / n_scan_normal_reg call structure (host.py).
There is NO CPU fallback: everything raises if the HIP library is missing or a call fails.
"""
from . import build as _build  # noqa: F401
from .capi import (CfearError, Context, Params, default_params, lib, lib_path)  # noqa: F401

__all__ = ["CfearError", "Context", "Params", "default_params", "lib", "lib_path"]

"""MI355X-native (gfx950, hand-written HIP) CFEAR radar-odometry hot path.

The product is libcfear_hip.so (C ABI: include/cfear_hip.h). This Python package is glue for
tests and benchmarks only: it builds the library (build.py) and binds the C ABI with ctypes (capi.py); readers.py /
replay.py / kitti.py are the dataset readers, the recorded-sequence replay and the KITTI drift metric, synth.py the synthetic
sweeps. The C++ mirror of the reference's radarDriver / MapPointNormal / n_scan_normal_reg classes is include/cfear_hip/cfear_host.hpp.
There is NO CPU fallback: everything raises if the HIP library is missing or a call fails.
"""
from . import build as _build  # noqa: F401
from .capi import (CfearError, Context, Params, default_params, lib, lib_path)  # noqa: F401

__all__ = ["CfearError", "Context", "Params", "default_params", "lib", "lib_path"]

"""featuredetection_amd -- MI355X-native (gfx950) implementation of the FeatureDetection hot path:
sliding-window extraction + HistEq64/HOG features + WVM/SVM scoring, and the SDM landmark step.

The product is the C-ABI shared library (include/fd_hip.h, featuredetection_amd/libfd_hip.so, built
from featuredetection_amd/csrc/*.hip) plus the C++ host mirror of the reference interfaces
(featuredetection_amd/host/).  This Python package is only the ctypes binding used by tests/bench.
"""
from . import capi, synth  # noqa: F401

__all__ = ["capi", "synth"]

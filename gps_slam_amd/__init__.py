"""gps_slam_amd -- MI355X-native hot path of GPS-SLAM (splat `ges` rasterizer + TSDF fusion).

The product is the C-ABI library built from csrc/*.hip (include/gps_slam_hip.h); the Python
modules here are the host-side mirror of the reference's operator surface
(gsplat/gsplat_wapper.hpp, raw_gs_model, ITMBasicEngine) and only move pointers.
"""
from ._lib import lib, load_library  # noqa: F401

__all__ = ["lib", "load_library"]

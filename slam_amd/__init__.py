"""slam_amd — MI355X-native FastSLAM inner loop (hand-written HIP for gfx950 behind the C ABI in
include/slamgpu.h).  This Python package is only the test/bench binding over that C ABI: the product
is slam_amd/libslamgpu.so (+ the C++ host driver slam_amd/bin/slam-backend).  There is no CPU or
PyTorch fallback: importing works anywhere, but every compute call needs the built library and a GPU.
"""
from .capi import (Config, ShardPlan, SlamGpu, SlamGpuError, FASTSLAM1, FASTSLAM2, RNG_TAPE, RNG_PHILOX, MATH_STRICT, MATH_FAST,
                   lib_path, load_library, jacobians, kat, device_count, DECLARED_SYMBOLS, DistGroup, dist_comm_id)

__all__ = ["Config", "ShardPlan", "SlamGpu", "SlamGpuError", "FASTSLAM1", "FASTSLAM2", "RNG_TAPE", "RNG_PHILOX", "MATH_STRICT",
           "MATH_FAST", "lib_path", "load_library", "jacobians", "kat", "device_count", "DECLARED_SYMBOLS", "DistGroup", "dist_comm_id"]

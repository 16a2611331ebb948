// Shared by the two exact distCUDA2 kernels (tiled brute force in splat_init.hip, uniform grid in splat_knn.hip): one squared
// distance, one "keep the three smallest" step -- the same float operations, so both kernels return the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace gps {

__device__ __forceinline__ float knn_dist2(float dx, float dy, float dz) {
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// simple_knn.cu:137-150: b0 <= b1 <= b2 stay sorted
__device__ __forceinline__ void keep3(float d, float& b0, float& b1, float& b2) {
    if (b0 > d) { float t = b0; b0 = d; d = t; }
    if (b1 > d) { float t = b1; b1 = d; d = t; }
    if (b2 > d) { b2 = d; }
}

}  // namespace gps

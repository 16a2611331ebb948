// Shared device helpers of the TSDF kernels (hash, 4x4 products in the reference's operation
// order, frustum tests).  Files including this header are compiled with -ffp-contract=off.
#pragma once
#include "common.hpp"

namespace gpst {

typedef gps_tsdf_state TsdfState;

constexpr int BLK = 8;     // SDF_BLOCK_SIZE
constexpr int BLK3 = 512;  // SDF_BLOCK_SIZE3
constexpr int BAND_STEP_BITS = 6;
constexpr int MAX_BAND_STEPS = 1 << BAND_STEP_BITS;
constexpr float FAR_AWAY = 999999.9f;
constexpr float VERY_CLOSE = 0.05f;
constexpr int MINMAX_SUB = 8;
constexpr int MAX_RENDERING_BLOCKS = 65536 * 4;

struct Mat4 { float m[16]; };  // ORUtils layout m[col*4 + row]

static inline Mat4 load_mat(const float* p) { Mat4 r; for (int i = 0; i < 16; i++) r.m[i] = p[i]; return r; }

static inline bool state_valid(const gps_tsdf_state& s) {
    if (s.width <= 0 || s.height <= 0 || s.n_blocks <= 0 || s.n_buckets <= 0 || s.n_excess <= 0) return false;
    if ((s.n_buckets & (s.n_buckets - 1)) != 0) return false;  // hash mask = n_buckets - 1
    if (!(s.voxel_size > 0.f) || !(s.mu > 0.f) || s.max_w <= 0 || s.max_w > 255) return false;
    return s.vba && s.vba_alloc_list && s.hash && s.excess_list && s.counters && s.alloc_prio && s.scan_scratch &&
           s.visible_type && s.visible_ids && s.depth && s.rgb && s.minmax && s.raycast && s.icp_points &&
           s.icp_normals && s.fv_visible_ids && s.fv_minmax && s.fv_raycast && s.fv_colour;
}

// r = M * (x,y,z,w), xyz only; ((m0*x + m4*y) + m8*z) + m12*w  (ORUtils/Matrix.h:130-137)
__device__ __forceinline__ void mul_point(const Mat4& M, float x, float y, float z, float w, float& rx, float& ry,
                                          float& rz) {
    rx = M.m[0] * x + M.m[4] * y + M.m[8] * z + M.m[12] * w;
    ry = M.m[1] * x + M.m[5] * y + M.m[9] * z + M.m[13] * w;
    rz = M.m[2] * x + M.m[6] * y + M.m[10] * z + M.m[14] * w;
}

// ITMRepresentationAccess.h:8-11
__device__ __forceinline__ int hash_index(int x, int y, int z, int mask) {
    return (int)((((uint32_t)x * 73856093u) ^ ((uint32_t)y * 19349669u) ^ ((uint32_t)z * 83492791u)) & (uint32_t)mask);
}

// checkPointVisibility<false> (Reconstruction Shared.h:326-353)
__device__ __forceinline__ bool point_visible(const TsdfState& s, const Mat4& M, float px, float py, float pz) {
    float bx, by, bz;
    mul_point(M, px, py, pz, 1.0f, bx, by, bz);
    if (bz < 1e-10f) return false;
    bx = s.fx * bx / bz + s.cx;
    by = s.fy * by / bz + s.cy;
    return bx >= 0 && bx < s.width && by >= 0 && by < s.height;
}

// checkBlockVisibility<false> (Shared.h:358-422): corners visited in the reference's order with the
// reference's running += / -= updates (the float value of a corner depends on that path).
__device__ __forceinline__ bool block_visible(const TsdfState& s, const Mat4& M, int bx, int by, int bz) {
    const float factor = (float)BLK * s.voxel_size;
    float x = (float)bx * factor, y = (float)by * factor, z = (float)bz * factor;
    if (point_visible(s, M, x, y, z)) return true;
    z += factor; if (point_visible(s, M, x, y, z)) return true;
    y += factor; if (point_visible(s, M, x, y, z)) return true;
    x += factor; if (point_visible(s, M, x, y, z)) return true;
    z -= factor; if (point_visible(s, M, x, y, z)) return true;
    y -= factor; if (point_visible(s, M, x, y, z)) return true;
    x -= factor; y += factor; if (point_visible(s, M, x, y, z)) return true;
    x += factor; y -= factor; z += factor; if (point_visible(s, M, x, y, z)) return true;
    return false;
}

}  // namespace gpst

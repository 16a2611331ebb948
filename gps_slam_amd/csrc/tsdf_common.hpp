// Shared device helpers of the TSDF kernels (hash, 4x4 products in the reference's operation
// order, frustum tests).  Files including this header are compiled with -ffp-contract=off.
#pragma once
#include <string.h>

#include "common.hpp"

// Wave priority of the frame chain's kernels (tracking, fusion, raycast).  In the overlap schedule they share the SIMDs with the
// map stream's rasterizer waves and the frame chain is the critical path: s_setprio raises their share of the issue slots
// (user priority 0..3; it only orders instruction arbitration between resident waves, no effect when they run alone).
#ifndef GPS_FRAME_PRIO_LEVEL
#define GPS_FRAME_PRIO_LEVEL 3
#endif
#define GPS_FRAME_PRIO() __builtin_amdgcn_s_setprio(GPS_FRAME_PRIO_LEVEL)
GPS_TUNABLE_REPORT(GPS_FRAME_PRIO_LEVEL, 3);   // (once per including translation unit; the registry keeps one copy)

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
constexpr int ED_GROUPS = 128;  // per-workgroup partial min/max images of CreateExpectedDepths (x 512 threads: one visible block per thread up to 65,536)

// Layout of gps_tsdf_state.scan_scratch (int32 words; sized by gps_tsdf_scratch_bytes):
//   [3 * nblk + 16]             per-1024-slot counts of the ordered sweeps + totals
//   [(n_total + 3) / 4 + 2]     one flag byte per hash slot (visible-list compaction)
//   [ED_GROUPS * sw * sh * 2]   partial min/max images
//   [16]                        pad
//   [n_buckets / 32]            PERSISTENT: one bit per bucket, set iff its head entry is non-empty (ptr != -2).  Blocks are
//                               never freed, so alloc_apply only ever sets bits; gps_tsdf_reset clears them,
//                               gps_tsdf_rebuild_index recomputes them from the table (after a scene was loaded into it).
//                               The raycaster's free-space march asks this 128 KB bitmap instead of the 16 MB table.
//   [16]                        pad
//   [4 + 4 * waves]             ray statistics of the last raycast launch (ray_stats_rows below), waves = 4 per 16 x 16 pixel patch
static inline int64_t scratch_words_before_bits(const gps_tsdf_state& s) {
    const int64_t n_total = (int64_t)s.n_buckets + s.n_excess;
    const int64_t nblk = (n_total + 1023) / 1024;
    const int64_t sw = s.width / MINMAX_SUB + 2, sh = s.height / MINMAX_SUB + 2;
    return 3 * nblk + 16 + (n_total + 3) / 4 + 2 + (int64_t)ED_GROUPS * sw * sh * 2 + 16;
}
static inline uint32_t* bucket_bits(const gps_tsdf_state& s) {
    return reinterpret_cast<uint32_t*>(s.scan_scratch) + scratch_words_before_bits(s);
}
// behind the bitmap (+ 16 words of pad): the ray statistics of the LAST raycast launch on this scratch, one {castRay steps,
// voxel reads, rays, 0} quadruple of floats per wave of that launch, written with plain stores (no atomics, no zero-fill;
// gps_tsdf_ray_stats sums them into counters[GPS_TSDF_RAY_STEPS..] on demand)
__host__ __device__ inline int ray_stat_waves(const gps_tsdf_state& s) { return ((s.width + 7) / 8) * ((s.height + 7) / 8) * 4; }  // (room for 4 x 4 pixel waves)
__host__ __device__ inline int64_t ray_stats_offset_words(const gps_tsdf_state& s) {
    const int64_t n_total = (int64_t)s.n_buckets + s.n_excess;
    const int64_t nblk = (n_total + 1023) / 1024;
    const int64_t sw = s.width / 8 + 2, sh = s.height / 8 + 2;
    return 3 * nblk + 16 + (n_total + 3) / 4 + 2 + (int64_t)ED_GROUPS * sw * sh * 2 + 16 + ((int64_t)s.n_buckets + 31) / 32 + 16;
}
__host__ __device__ inline float4* ray_stats_rows(const gps_tsdf_state& s) {
    return reinterpret_cast<float4*>(s.scan_scratch + ((ray_stats_offset_words(s) + 3) & ~(int64_t)3));   // (16-byte aligned rows)
}

struct Mat4 { float m[16]; };  // ORUtils layout m[col*4 + row]

static inline Mat4 load_mat(const float* p) { Mat4 r; for (int i = 0; i < 16; i++) r.m[i] = p[i]; return r; }

// One free view of a batch (gps_tsdf_free_raycast_batch): the table the batch kernels read with blockIdx.z.  Every kernel of
// the free-view chain takes the scene by value as before plus an optional table; with a table, apply_view() swaps in the view's
// own render-state buffers, scratch, counters, intrinsics and pose -- the kernel body is the single-view code.
struct ViewRec {
    Mat4 M, invM;
    float fx, fy, cx, cy;
    int32_t* visible_ids;
    float* minmax;
    float* raycast;
    uint8_t* colour;
    int32_t* scratch;
    int32_t* counters;
    // optional runRaycastByCam glue of the view (gps_raycast_to_maps), written by the batch's last kernel; color_map == NULL: none
    Mat4 w2c_rm;
    float *color_map, *vertex_map, *conf_map, *depth_map, *depth_clamped;
};
constexpr int MAX_BATCH_VIEWS = 12;
struct ViewTable { ViewRec v[MAX_BATCH_VIEWS]; };

__device__ __forceinline__ void apply_view(TsdfState& s, const ViewRec& v) {
    s.fv_visible_ids = v.visible_ids; s.fv_minmax = v.minmax; s.fv_raycast = v.raycast; s.fv_colour = v.colour;
    s.scan_scratch = v.scratch; s.counters = v.counters;
    s.fx = v.fx; s.fy = v.fy; s.cx = v.cx; s.cy = v.cy;
}
// regions of scan_scratch the ordered visible-list sweep and the expected-depth pass use (see the layout above)
__host__ __device__ __forceinline__ int sweep_blocks(const TsdfState& s) { return (s.n_buckets + s.n_excess + 1023) / 1024; }
__host__ __device__ __forceinline__ int32_t* sweep_counts(const TsdfState& s) { return s.scan_scratch + 2 * sweep_blocks(s); }
__host__ __device__ __forceinline__ uint8_t* sweep_flags(const TsdfState& s) {
    return reinterpret_cast<uint8_t*>(s.scan_scratch + 3 * sweep_blocks(s) + 16);
}
__host__ __device__ __forceinline__ uint2* minmax_partials(const TsdfState& s) {
    return reinterpret_cast<uint2*>(s.scan_scratch + 3 * sweep_blocks(s) + 16 + (s.n_buckets + s.n_excess + 3) / 4 + 2);
}

static inline bool state_valid(const gps_tsdf_state& s) {
    if (s.width <= 0 || s.height <= 0 || s.n_blocks <= 0 || s.n_buckets <= 0 || s.n_excess <= 0) return false;
    if ((s.n_buckets & (s.n_buckets - 1)) != 0 || s.n_buckets < 32) return false;  // hash mask = n_buckets - 1
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

// A hash entry is 16 bytes {short pos[3]; pad; int offset; int ptr}.  Reading it field by field -- or as a uint4
// the optimiser is free to narrow -- makes the compiler emit one small load per field *at its first use*, i.e. a chain
// of dependent round trips behind the compares.  load_raw fetches it as one aligned dwordx4; pin() is an empty asm
// that names all four dwords, so the load stays whole and stays where it was issued.  Batches: issue all load_raw
// calls first, pin them afterwards (a pin waits for its operands), then decode.
struct HashEntry { int x, y, z, offset, ptr; };
__device__ __forceinline__ uint4 load_raw(const gps_hash_entry* __restrict__ h, int idx) {
    return reinterpret_cast<const uint4*>(h)[idx];
}
__device__ __forceinline__ void pin(uint4& a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }
__device__ __forceinline__ void pin(uint4& a, uint4& b) {
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
}
__device__ __forceinline__ HashEntry decode_entry(const uint4 r) {
    HashEntry e;
    e.x = (int)(int16_t)(r.x & 0xFFFFu); e.y = (int)(int16_t)(r.x >> 16); e.z = (int)(int16_t)(r.y & 0xFFFFu);
    e.offset = (int)r.z; e.ptr = (int)r.w;
    return e;
}
__device__ __forceinline__ HashEntry load_entry(const gps_hash_entry* __restrict__ h, int idx) {
    uint4 r = load_raw(h, idx);
    pin(r);
    return decode_entry(r);
}
__device__ __forceinline__ bool entry_is(const HashEntry& e, int bx, int by, int bz) {
    return (e.x == bx) & (e.y == by) & (e.z == bz);
}

// ---- correctly rounded divisions without the generic 11-instruction expansion ----------------------------------------
// `a / b` in IEEE float costs div_scale x2 + rcp + 5 fma + div_fmas + div_fixup; the integrate kernel does 14 of them
// per voxel (44 % of its VALU work).  Two exact shortcuts, both checked exhaustively / by 2e9 random cases on the CPU
// against `a / b` (tests/test_division_identities.py):
//  * divisor known in advance (255, 32767, mu): y = RN(1/b); q = a*y; r = fma(-b, q, a); q' = fma(r, y, q) is the
//    correctly rounded quotient for every a when b's significand is not all ones (Markstein's theorem; 0 mismatches
//    over all 6.7e8 floats in [2^-60, 2^20] for the three divisors used here);
//  * several numerators over one divisor: the hardware algorithm itself (reciprocal estimate, one Newton step, quotient
//    with two residual corrections) with the refined reciprocal shared; valid without the range scaling of
//    v_div_scale / v_div_fixup as long as b and a/b are well inside the normal range (callers guarantee it).
__device__ __forceinline__ float div_known(float a, float b, float y) {
    const float q = a * y;
    const float r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ float refined_rcp(float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(y0, e, y0);
}
__device__ __forceinline__ float div_shared(float a, float b, float y1) {
    const float q0 = a * y1;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y1, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y1, q1);
}
// host: is div_known(a, b, 1/b) == a / b for every a?  (significand of b not all ones)
static inline bool div_known_safe(float b) {
    uint32_t u;
    memcpy(&u, &b, 4);
    return (u & 0x7FFFFFu) != 0x7FFFFFu && b > 1e-30f && b < 1e30f;
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

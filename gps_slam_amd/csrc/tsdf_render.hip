// TSDF rendering for gfx950: expected-depth (min/max) image, ray casting through the voxel-block
// hash, ICP point/normal maps, colour rendering; plus the frame-level entry points.
// Compiled with -ffp-contract=off (ray positions are compared bit-for-bit with the CPU engine).
//
// Per-element maths: InfiniTAM/ITMLib/Engines/Visualisation/Shared/ITMVisualisationEngine_Shared.h and
// Objects/Scene/ITMRepresentationAccess.h (restated line by line in oracle/tsdf_oracle.c).
//
// Differences in organisation from the reference CUDA back-end:
//  * CreateExpectedDepths does not materialise the <= 262,144 "rendering block" list nor copy its
//    length to the host (…_CUDA.tcu:137-184): each visible block projects its 8 corners and
//    min/max-es its bounding box straight into the 1/8-resolution image with integer atomics
//    (positive floats order like their bit patterns).  Identical result unless the reference's
//    MAX_RENDERING_BLOCKS cap would have been hit, which is flagged in counters[GPS_TSDF_OVERFLOW].
//  * All kernels read their sizes from the device counters: no blocking readback anywhere.
#include <math.h>

#include "launch_timing.hpp"
#include "tsdf_common.hpp"
#include "tsdf_pose.hpp"
#include "wave_reduce.hpp"

using namespace gpst;

namespace {

// CreateExpectedDepths in two passes without global atomics:
//  A) each workgroup keeps a private copy of the 1/8-resolution min/max image in LDS (<= 160 KB: 640x480 -> 82x62x8 B
//     = 40 KB, a plane of mins and a plane of maxes), takes its share of the visible list, projects the 8 block corners
//     (ProjectSingleBlock, Shared.h:36-91) and min/max-es the bounding box with LDS integer atomics (positive floats order like
//     their bit patterns), then writes its image to a partial buffer with plain coalesced stores;
//  B) one thread per cell of that window reduces the partials into the full-resolution-stride image the raycaster indexes
//     (x/8 + (y/8)*W).  Every other pixel of that image holds (FAR_AWAY, VERY_CLOSE) -- the reference's memset kernel
//     rewrites them every call; here gps_tsdf_reset writes them once and nothing ever touches them again.
// The rendering-block count accumulates in a scratch counter that pass B publishes and clears, so no memset launch
// precedes pass A.
// (Round 5, measured and dropped: device-scope atomics straight into ONE global image -- no LDS image, no partials, nothing to
// reduce: 52 us against 10.8 for pass A; ~500 k read-modify-writes on 4,800 words execute at the memory side, a few ns apart
// per word.)
#ifndef GPS_ED_THREADS
#define GPS_ED_THREADS 512
#endif
GPS_TUNABLE_REPORT(GPS_ED_THREADS, 512);
constexpr int ED_THREADS = GPS_ED_THREADS;

// One visible block: ProjectSingleBlock (Shared.h:36-91) + the min / max of its bounding box into the workgroup's LDS image
// (two planes -- every bank serves the mins and every bank serves the maxes; interleaved, each plane sat on half the banks).
// -> rendering blocks the reference would have created for it (CreateRenderingBlocks counts 16 x 16 pixel pieces)
__device__ __forceinline__ int ed_one_block(const TsdfState& s, const Mat4& M, const HashEntry& he, int sw, int sh,
                                            uint32_t* __restrict__ lo, uint32_t* __restrict__ hi) {
    const int W = s.width, H = s.height;
    if (he.ptr < 0) return 0;
    int ulx = W / MINMAX_SUB, uly = H / MINMAX_SUB, lrx = -1, lry = -1;
    float zmin = FAR_AWAY, zmax = VERY_CLOSE;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        short tx = (short)he.x, ty = (short)he.y, tz = (short)he.z;
        tx += (corner & 1) ? 1 : 0; ty += (corner & 2) ? 1 : 0; tz += (corner & 4) ? 1 : 0;
        float px, py, pz;
        mul_point(M, (float)tx * (float)BLK * s.voxel_size, (float)ty * (float)BLK * s.voxel_size,
                  (float)tz * (float)BLK * s.voxel_size, 1.0f, px, py, pz);
        if (pz < 1e-6) continue;
        const float u = (s.fx * px / pz + s.cx) / MINMAX_SUB;
        const float v = (s.fy * py / pz + s.cy) / MINMAX_SUB;
        if (ulx > floorf(u)) ulx = (int)floorf(u);
        if (lrx < ceilf(u)) lrx = (int)ceilf(u);
        if (uly > floorf(v)) uly = (int)floorf(v);
        if (lry < ceilf(v)) lry = (int)ceilf(v);
        if (zmin > pz) zmin = pz;
        if (zmax < pz) zmax = pz;
    }
    if (ulx < 0) ulx = 0;
    if (uly < 0) uly = 0;
    if (lrx >= W) lrx = W - 1;
    if (lry >= H) lry = H - 1;
    if (ulx > lrx) return 0;
    if (uly > lry) return 0;
    if (zmin < VERY_CLOSE) zmin = VERY_CLOSE;
    if (zmax < VERY_CLOSE) return 0;
    const int blocks = (int)ceilf((float)(lrx - ulx + 1) / 16.0f) * (int)ceilf((float)(lry - uly + 1) / 16.0f);
    // the bounding box can only leave the sw x sh window in degenerate projections (the reference clamps to the
    // FULL image size, Shared.h:77-80); those pixels are never read by the raycaster -> clip to the window
    lrx = min(lrx, sw - 1); lry = min(lry, sh - 1);
    for (int y = uly; y <= lry; ++y)
        for (int x = ulx; x <= lrx; ++x) {
            atomicMin(&lo[x + y * sw], __float_as_uint(zmin));
            atomicMax(&hi[x + y * sw], __float_as_uint(zmax));
        }
    return blocks;
}

// ED_GROUPS x ED_THREADS threads = one visible block per thread up to 65,536 blocks (a 640x480 frame of 5 mm voxels sees 40-60 k):
// ONE dependent chain per thread -- list entry, hash entry, 8 projections, a handful of LDS atomics -- instead of two back to
// back (round 4: 64 x 512 threads, two trips; 19 us).  The chain's head does not wait for the list LENGTH either: the first
// trip's list entry and hash entry are requested before the counter has arrived (the list buffer is n_blocks long and holds
// slot numbers of earlier frames behind its end: the speculative reads stay inside the table, their result is dropped).
__global__ __launch_bounds__(ED_THREADS) void expected_depths_partial_kernel(TsdfState s, Mat4 M,
                                                                     const int32_t* __restrict__ vis_ids, int count_slot,
                                                                     int sw, int sh, uint2* __restrict__ partial,
                                                                     const ViewRec* __restrict__ views) {
    GPS_FRAME_PRIO();
    extern __shared__ uint32_t ed_img[];  // [sw*sh] min bits | [sw*sh] max bits
    int32_t* const overflow_word = s.counters + GPS_TSDF_OVERFLOW;  // (of the scene, also for a view of a batch)
    if (views) { apply_view(s, views[blockIdx.z]); M = views[blockIdx.z].M; vis_ids = s.fv_visible_ids; partial = minmax_partials(s); }
    const int cells = sw * sh;
    uint32_t* const lo = ed_img;
    uint32_t* const hi = ed_img + cells;
    const int stride = gridDim.x * blockDim.x, k0 = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_total = s.n_buckets + s.n_excess;
    // first trip, speculatively (k0 < n_blocks: the grid is smaller than the list buffer)
    const bool spec = k0 < s.n_blocks;
    const int id0 = spec ? vis_ids[k0] : -1;
    const bool id_ok = id0 >= 0 && id0 < n_total;
    uint4 raw0 = make_uint4(0, 0, 0, 0);
    if (id_ok) raw0 = load_raw(s.hash, id0);
    const int n = s.counters[count_slot];
    for (int i = threadIdx.x; i < cells; i += blockDim.x) { lo[i] = __float_as_uint(FAR_AWAY); hi[i] = __float_as_uint(VERY_CLOSE); }
    __syncthreads();
    int my_blocks = 0;
    if (k0 < n && id_ok) {
        pin(raw0);
        my_blocks += ed_one_block(s, M, decode_entry(raw0), sw, sh, lo, hi);
    }
    for (int k = k0 + stride; k < n; k += stride) my_blocks += ed_one_block(s, M, load_entry(s.hash, vis_ids[k]), sw, sh, lo, hi);
    // one atomic per WORKGROUP on the rendering-block counter (same-address atomics serialise at the memory side: the tracker's
    // prepare kernel spent 10 of its 16 us on 1,200 of them)
    __shared__ int wave_tot[ED_THREADS / 64];
    const int tot_w = wave_sum_i(my_blocks);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = tot_w;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int k = 0; k < ED_THREADS / 64; k++) tot += wave_tot[k];
        if (tot) {
            const int before = atomicAdd(&s.counters[GPS_TSDF_SCRATCH2], tot);
            if (before + tot >= MAX_RENDERING_BLOCKS) *overflow_word = 1;
        }
    }
    uint2* out = partial + (size_t)blockIdx.x * cells;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) out[i] = make_uint2(lo[i], hi[i]);
}

__global__ __launch_bounds__(256) void expected_depths_reduce_kernel(TsdfState s, int sw, int sh, int groups,
                                                                    const uint2* __restrict__ partial,
                                                                    float2* __restrict__ mm, const ViewRec* __restrict__ views) {
    GPS_FRAME_PRIO();
    if (views) { apply_view(s, views[blockIdx.z]); partial = minmax_partials(s); mm = reinterpret_cast<float2*>(s.fv_minmax); }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {  // pass A of this call is complete (stream order): publish its rendering-block count, clear the scratch
        s.counters[GPS_TSDF_RENDER_BLOCKS] = s.counters[GPS_TSDF_SCRATCH2];
        s.counters[GPS_TSDF_SCRATCH2] = 0;
    }
    if (i >= sw * sh) return;
    const int y = i / sw, x = i - y * sw;
    if (x >= s.width || y >= s.height) return;
    uint32_t lo = __float_as_uint(FAR_AWAY), hi = __float_as_uint(VERY_CLOSE);
#pragma unroll 16
    for (int g = 0; g < groups; g++) {
        const uint2 p = partial[(size_t)g * sw * sh + i];
        lo = min(lo, p.x); hi = max(hi, p.y);
    }
    mm[x + y * s.width] = make_float2(__uint_as_float(lo), __uint_as_float(hi));
}

// ---------------------------------------------------------------- voxel access (ITMRepresentationAccess.h)
struct BlockCache { int bx, by, bz, ptr; };
#ifndef GPS_RAYCAST_SKIP
#define GPS_RAYCAST_SKIP 5
#endif
GPS_TUNABLE_REPORT(GPS_RAYCAST_SKIP, 5);
#ifdef GPS_ED_REDUCE_LAUNCH   // (probe builds: pass B of the expected depths as its own launch)
GPS_SWITCH_REPORT(GPS_ED_REDUCE_LAUNCH);
#endif
constexpr int SKIP = GPS_RAYCAST_SKIP;
#ifndef GPS_RAYCAST_RAYS_PER_WAVE
#define GPS_RAYCAST_RAYS_PER_WAVE 64
#endif
GPS_TUNABLE_REPORT(GPS_RAYCAST_RAYS_PER_WAVE, 64);
// pixels a wave's rays cover: 8 x 8 (one cell of the min/max image), 8 x 4 or 4 x 4 (fewer rays per wave: less divergence between
// the lanes of a wave, more waves; the patch stays inside one min/max cell)
constexpr int RC_PW = GPS_RAYCAST_RAYS_PER_WAVE >= 32 ? 8 : 4, RC_PH = GPS_RAYCAST_RAYS_PER_WAVE >= 64 ? 8 : 4;
static_assert(GPS_RAYCAST_RAYS_PER_WAVE == 64 || GPS_RAYCAST_RAYS_PER_WAVE == 32 || GPS_RAYCAST_RAYS_PER_WAVE == 16, "rays per wave");  // free-space look-ahead of the raycaster, 1..31

__device__ __forceinline__ int floor_div_blk(int v) { return v >> 3; }  // floor(v / 8): arithmetic shift (BLK == 8)

// findVoxel (ITMRepresentationAccess.h:81-113) split into "which block" and "which voxel".  resolve_with_head returns
// the first voxel index of block (bx,by,bz) or -1 with the reference's cache and vmIndex side effects (cache hit ->
// vmIndex = 1; hash hit -> vmIndex = entry + 1 and the cache moves; miss -> vmIndex = 0).  The bucket-head entry is
// passed in: callers fetch the heads of everything they are about to look up in ONE batch of independent loads, then
// replay the lookups in the reference's order from registers (only excess-list chains, which are rare, cost further
// round trips), then issue the voxel loads together.  A wave therefore pays ~2 memory round trips per lookup group
// instead of one per lane-divergent branch.
__device__ __forceinline__ int resolve_with_head(const TsdfState& s, int bx, int by, int bz, HashEntry he, int hashIdx,
                                                 int& vmIndex, BlockCache& c) {
    if (bx == c.bx && by == c.by && bz == c.bz) { vmIndex = 1; return c.ptr; }  // :89-93 (vmIndex = true)
    while (true) {
        if (entry_is(he, bx, by, bz) & (he.ptr >= 0)) {
            c.bx = bx; c.by = by; c.bz = bz; c.ptr = he.ptr * BLK3;
            vmIndex = hashIdx + 1;
            return c.ptr;
        }
        if (he.offset < 1) break;
        hashIdx = s.n_buckets + he.offset - 1;
        he = load_entry(s.hash, hashIdx);
    }
    vmIndex = 0;
    return -1;
}

__device__ __forceinline__ int resolve_block(const TsdfState& s, int px, int py, int pz, int& lin, int& vmIndex,
                                             BlockCache& c) {
    const int bx = floor_div_blk(px), by = floor_div_blk(py), bz = floor_div_blk(pz);
    lin = px + (py - bx) * BLK + (pz - by) * BLK * BLK - bz * BLK3;
    if (bx == c.bx && by == c.by && bz == c.bz) { vmIndex = 1; return c.ptr; }
    const int hashIdx = hash_index(bx, by, bz, s.n_buckets - 1);
    return resolve_with_head(s, bx, by, bz, load_entry(s.hash, hashIdx), hashIdx, vmIndex, c);
}

constexpr uint64_t EMPTY_VOXEL = 0x7FFFull;  // TVoxel(): sdf 32767, weights / colour 0

__device__ __forceinline__ uint64_t read_voxel_raw(const TsdfState& s, int px, int py, int pz, int& vmIndex,
                                                   BlockCache& c) {
    int lin;
    const int base = resolve_block(s, px, py, pz, lin, vmIndex, c);
    return base >= 0 ? reinterpret_cast<const uint64_t*>(s.vba)[base + lin] : EMPTY_VOXEL;
}
__device__ __forceinline__ float vox_sdf(uint64_t raw) { return (float)(int16_t)(raw & 0xFFFF); }
__device__ __forceinline__ float vox_wdepth(uint64_t raw) { return (float)((raw >> 16) & 0xFF); }
__device__ __forceinline__ float roundf_ref(float x) { return (x < 0) ? (x - 0.5f) : (x + 0.5f); }

// the trilinear blend of readFromSDF_float_interpolated, in the reference's operation order; r[dx + 2*dy + 4*dz]
template <bool WITH_CONF>
__device__ __forceinline__ float blend_corners(const uint64_t (&r)[8], float cx, float cy, float cz, float& conf) {
    float res1 = (1.0f - cx) * vox_sdf(r[0]) + cx * vox_sdf(r[1]);
    res1 = (1.0f - cy) * res1 + cy * ((1.0f - cx) * vox_sdf(r[2]) + cx * vox_sdf(r[3]));
    float res2 = (1.0f - cx) * vox_sdf(r[4]) + cx * vox_sdf(r[5]);
    res2 = (1.0f - cy) * res2 + cy * ((1.0f - cx) * vox_sdf(r[6]) + cx * vox_sdf(r[7]));
    if (WITH_CONF) {
        float c1 = (1.0f - cx) * vox_wdepth(r[0]) + cx * vox_wdepth(r[1]);
        c1 = (1.0f - cy) * c1 + cy * ((1.0f - cx) * vox_wdepth(r[2]) + cx * vox_wdepth(r[3]));
        float c2 = (1.0f - cx) * vox_wdepth(r[4]) + cx * vox_wdepth(r[5]);
        c2 = (1.0f - cy) * c2 + cy * ((1.0f - cx) * vox_wdepth(r[6]) + cx * vox_wdepth(r[7]));
        conf = (1.0f - cz) * c1 + cz * c2;
    }
    return ((1.0f - cz) * res1 + cz * res2) / 32767.0f;
}

// readFromSDF_float_interpolated / readWithConfidenceFromSDF_float_interpolated (ITMRepresentationAccess.h:117-187).
// The reference reads the eight corners one after the other through the block cache.  Here:
//  * cell inside one block (local coordinates <= 6, 67% of cells): after the first lookup either the cache holds that
//    block and all eight voxels are loaded together, or the block is unallocated and the other seven lookups would fail
//    exactly like the first (no side effects, cache untouched);
//  * cell straddling blocks: the eight bucket heads are fetched together, the eight lookups are replayed in the
//    reference's order from registers (identical cache evolution), then the eight voxels are loaded together.
// the eight voxels of the interpolation cell at (ix, iy, iz), x fastest: r[dx + 2*dy + 4*dz]; unallocated -> EMPTY_VOXEL
__device__ __forceinline__ void fetch_cell(const TsdfState& s, int ix, int iy, int iz, int& vmIndex, BlockCache& c, uint64_t (&r)[8]) {
    const uint64_t* vox = reinterpret_cast<const uint64_t*>(s.vba);
    if ((ix & 7) < 7 && (iy & 7) < 7 && (iz & 7) < 7) {
        int lin;
        const int base = resolve_block(s, ix, iy, iz, lin, vmIndex, c);
        if (base >= 0) {
            const uint64_t* v = vox + base + lin;
            r[0] = v[0]; r[1] = v[1]; r[2] = v[BLK]; r[3] = v[BLK + 1];
            r[4] = v[BLK * BLK]; r[5] = v[BLK * BLK + 1]; r[6] = v[BLK * BLK + BLK]; r[7] = v[BLK * BLK + BLK + 1];
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) r[k] = EMPTY_VOXEL;
        }
    } else {
        int kx[8], ky[8], kz[8], hidx[8], idx[8];
        uint4 hraw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            kx[k] = floor_div_blk(ix + (k & 1)); ky[k] = floor_div_blk(iy + ((k >> 1) & 1));
            kz[k] = floor_div_blk(iz + (k >> 2));
            hidx[k] = hash_index(kx[k], ky[k], kz[k], s.n_buckets - 1);
            hraw[k] = load_raw(s.hash, hidx[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; k += 2) pin(hraw[k], hraw[k + 1]);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int vx = ix + (k & 1), vy = iy + ((k >> 1) & 1), vz = iz + (k >> 2);
            const int lin = vx + (vy - kx[k]) * BLK + (vz - ky[k]) * BLK * BLK - kz[k] * BLK3;
            const int base = resolve_with_head(s, kx[k], ky[k], kz[k], decode_entry(hraw[k]), hidx[k], vmIndex, c);
            idx[k] = base >= 0 ? base + lin : -1;
        }
        uint64_t t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = vox[idx[k] >= 0 ? idx[k] : 0];  // unconditional: one batch, no branches
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = idx[k] >= 0 ? t[k] : EMPTY_VOXEL;
    }
}

template <bool WITH_CONF>
__device__ __forceinline__ float read_sdf_interp(const TsdfState& s, float px, float py, float pz, int& vmIndex,
                                                 BlockCache& c, float& conf) {
    const float fx_ = floorf(px), fy_ = floorf(py), fz_ = floorf(pz);
    const float cx = px - fx_, cy = py - fy_, cz = pz - fz_;
    uint64_t r[8];  // x fastest: r[dx + 2*dy + 4*dz]
    fetch_cell(s, (int)fx_, (int)fy_, (int)fz_, vmIndex, c, r);
    vmIndex = 1;
    return blend_corners<WITH_CONF>(r, cx, cy, cz, conf);
}

// castRay (Shared.h:122-221); 16x16 pixel workgroups of four 8x8 wave patches
// `partial` != NULL: pass B of the expected depths (expected_depths_reduce_kernel) happens HERE -- a wave's 64 rays share ONE
// cell of the min/max image, so its 64 lanes fetch the cell's ED_GROUPS partial values (one each), reduce them with six
// shuffles, lane 0 writes the cell where pass B would have (the image other readers and the tests see), and the first wave
// publishes + clears the rendering-block counter: one launch and its ~8-10 us leave the frame chain.
// Waves per SIMD the register allocation must leave room for.  Left alone the kernel takes 98 VGPRs = 4 waves per SIMD = 4,096 waves
// on the chip, fewer than the 4,800 waves of a 640x480 launch: a second round for the last rows of the image.  5 -> 95 VGPRs, no
// spill, the whole launch resident at once (6 spills 18 registers).  Measured (round 6): a batch of 7 free views 516 -> 489 us,
// bench windows 1,042 / 1,369 -> 1,052 / 1,381 frames/s (2 + 2 runs: inside the noise, never below).  0 = no constraint.
#ifndef GPS_RAYCAST_WAVES_PER_EU
#define GPS_RAYCAST_WAVES_PER_EU 5
#endif
GPS_TUNABLE_REPORT(GPS_RAYCAST_WAVES_PER_EU, 5);
#if GPS_RAYCAST_WAVES_PER_EU > 0
#define GPS_RAYCAST_BOUNDS __launch_bounds__(256, GPS_RAYCAST_WAVES_PER_EU)
#else
#define GPS_RAYCAST_BOUNDS __launch_bounds__(256)
#endif
template <bool MODIFY_VISIBLE>
__global__ GPS_RAYCAST_BOUNDS void raycast_kernel(TsdfState s, Mat4 invM, const float2* __restrict__ minmax,
                                                     float4* __restrict__ rays, const uint32_t* __restrict__ bits,
                                                     const ViewRec* __restrict__ views, const uint2* __restrict__ partial, int sw,
                                                     int sh, float2* __restrict__ mm_out, gps::LaunchStamp stamp) {
    gps::StampScope timed(stamp);
    GPS_FRAME_PRIO();
    if (views) {
        apply_view(s, views[blockIdx.z]); invM = views[blockIdx.z].invM;
        minmax = reinterpret_cast<const float2*>(s.fv_minmax); rays = reinterpret_cast<float4*>(s.fv_raycast);
        if (sw) { partial = minmax_partials(s); mm_out = reinterpret_cast<float2*>(s.fv_minmax); }
    }
    // one wave64 = one 8x8 pixel patch = exactly one cell of the 1/8-resolution min/max image: all 64 rays share their
    // [min, max] range, so their free-space runs and step counts stay close (a 16x4 strip straddles two cells)
    const int wave_in_wg = threadIdx.x >> 6, lane_ = threadIdx.x & 63;
    // RC_PW x RC_PH pixels per wave (GPS_RAYCAST_RAYS_PER_WAVE rays; lanes beyond them idle), 2 x 2 waves per workgroup
    const int wave_x0 = blockIdx.x * (2 * RC_PW) + (wave_in_wg & 1) * RC_PW, wave_y0 = blockIdx.y * (2 * RC_PH) + (wave_in_wg >> 1) * RC_PH;
    const int x = wave_x0 + (lane_ % RC_PW), y = wave_y0 + (lane_ / RC_PW);
    const bool ray_lane = lane_ < RC_PW * RC_PH;
    float2 mm_cell = make_float2(0.f, 0.f);
    if (partial) {   // (before the early return: every lane of the wave takes part)
        const int cell_x = wave_x0 / MINMAX_SUB, cell_y = wave_y0 / MINMAX_SUB;
        uint32_t lo = __float_as_uint(FAR_AWAY), hi = __float_as_uint(VERY_CLOSE);
        if (cell_x < sw && cell_y < sh) {
            for (int g = lane_; g < ED_GROUPS; g += 64) {
                const uint2 pv = partial[(size_t)g * sw * sh + cell_x + cell_y * sw];
                lo = min(lo, pv.x); hi = max(hi, pv.y);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64)); }
        mm_cell = make_float2(__uint_as_float(lo), __uint_as_float(hi));
        const bool first_pixel_inside = wave_x0 < s.width && wave_y0 < s.height;
        if (lane_ == 0 && first_pixel_inside) mm_out[cell_x + cell_y * s.width] = mm_cell;
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // pass A of this call is complete (stream order)
            s.counters[GPS_TSDF_RENDER_BLOCKS] = s.counters[GPS_TSDF_SCRATCH2];
            s.counters[GPS_TSDF_SCRATCH2] = 0;
        }
    }
    // Lanes outside the image stay in the wave with an empty range (no loop trip, no store): the step statistics below are summed
    // with cross-lane operations that every lane takes part in.
    const bool inside = ray_lane && x < s.width && y < s.height;
    const int W = s.width;
    const int loc2 = (int)floorf((float)x / MINMAX_SUB) + (int)floorf((float)y / MINMAX_SUB) * W;
    const float2 mm = !inside ? make_float2(0.f, 0.f) : partial ? mm_cell : minmax[loc2];
    const float oneOverVoxelSize = 1.0f / s.voxel_size;
    const float ipx = 1.0f / s.fx, ipy = 1.0f / s.fy, ipz = -s.cx, ipw = -s.cy;
    const float stepScale = s.mu * oneOverVoxelSize;
    float cz = mm.x;
    float cx = cz * (((float)x + ipz) * ipx);
    float cy = cz * (((float)y + ipw) * ipy);
    float l2 = 0; l2 += cx * cx; l2 += cy * cy; l2 += cz * cz;
    float totalLength = sqrtf(l2) * oneOverVoxelSize;
    float qx, qy, qz;
    mul_point(invM, cx, cy, cz, 1.0f, qx, qy, qz);
    const float sx = qx * oneOverVoxelSize, sy = qy * oneOverVoxelSize, sz = qz * oneOverVoxelSize;
    cz = mm.y;
    cx = cz * (((float)x + ipz) * ipx);
    cy = cz * (((float)y + ipw) * ipy);
    l2 = 0; l2 += cx * cx; l2 += cy * cy; l2 += cz * cz;
    const float totalLengthMax = sqrtf(l2) * oneOverVoxelSize;
    mul_point(invM, cx, cy, cz, 1.0f, qx, qy, qz);
    float rx = qx * oneOverVoxelSize - sx, ry = qy * oneOverVoxelSize - sy, rz = qz * oneOverVoxelSize - sz;
    const float dn = 1.0f / sqrtf(rx * rx + ry * ry + rz * rz);
    rx *= dn; ry *= dn; rz *= dn;
    float px = sx, py = sy, pz = sz;
    BlockCache cache = {0x7fffffff, 0x7fffffff, 0x7fffffff, -1};
    float sdfValue = 1.0f, confidence = 0.f, stepLength;
    int vmIndex = 0;
    bool look = true;    // fetch the free-space candidates with the next uncached lookup (the last lookup failed / none yet)
    uint32_t n_log = 0;  // S-bar log, packed: trips of this loop (= voxel reads) in the low half, skipped candidates in the high half
    const uint64_t* vox = reinterpret_cast<const uint64_t*>(s.vba);
    while (totalLength < totalLengthMax) {
        // Candidate positions: [0] is this step's sample; [1..SKIP] are where the next steps land IF this and the
        // following lookups fail (a failed lookup always advances by one block edge, stepLength = 8 voxels) -- the same
        // float additions in the same order as the one-lookup-per-step loop.  A lane whose sample is in its cached
        // block needs no bucket at all; the others fetch all 1+SKIP bucket heads in one batch.
        const int vx = (int)roundf_ref(px), vy = (int)roundf_ref(py), vz = (int)roundf_ref(pz);
        const int kx0 = floor_div_blk(vx), ky0 = floor_div_blk(vy), kz0 = floor_div_blk(vz);
        const int lin = vx + (vy - kx0) * BLK + (vz - ky0) * BLK * BLK - kz0 * BLK3;
        const bool cached = kx0 == cache.bx && ky0 == cache.by && kz0 == cache.bz;
        int hidx0 = 0;
        uint4 hraw0 = {};
        uint32_t occupied = ~0u;  // bit j: the bucket of candidate j has a non-empty head (or was not looked at)
        if (!cached) {
            hidx0 = hash_index(kx0, ky0, kz0, s.n_buckets - 1);
            hraw0 = load_raw(s.hash, hidx0);
            // candidates 1..SKIP: where the next steps land IF this and the following lookups fail (a failed lookup always
            // advances by one block edge) -- the same float additions in the same order as the one-lookup-per-step loop.
            // Only the occupancy BIT of their buckets is fetched (one batch of dword loads from a 128 KB bitmap, see
            // tsdf_common.hpp); an empty head is a certain miss (ITMRepresentationAccess.h:95-110: no entry, no chain).
            // Round 1 fetched the 16-byte heads of 4 candidates; measured on the bench scene (tools/probe/raycast_time.py,
            // live raycast): heads x 4: 137.5 us; bits x 1 / 2 / 3 / 5 / 7 / 9 / 15: 123.6 / 115.4 / 116.5 / 112.5 / 116.1 /
            // 122.2 / 141.1 us -- past ~7 the per-candidate index arithmetic (round, shift, hash: ~25 VALU ops) outweighs
            // the saved round trips (on a scene of 10 m free-space runs: 926 -> 673 us with ONE candidate, slower again
            // with more).
            // (round 6) ... and only for a lane whose LAST lookup failed (or that has not looked anything up yet): a ray marching
            // through the truncation band enters its next block every few trips, finds it allocated nearly always, and has no use
            // for the candidates -- but one such lane per trip made the whole wave issue their ~125 instructions.  The longest
            // rays of a launch are such rays (tools/raycast_wave_hist.py), and a lone wave issues an instruction every ~5 cycles.
            // Without the candidates a failed lookup advances by its own block edge only and the next trip looks ahead: the same
            // positions, the same additions, one more trip at a band -> free-space transition.
            if (look) {
                uint32_t word[SKIP], shift[SKIP];
                float cx_ = px, cy_ = py, cz_ = pz;
#pragma unroll
                for (int j = 0; j < SKIP; j++) {
                    cx_ += (float)BLK * rx; cy_ += (float)BLK * ry; cz_ += (float)BLK * rz;
                    const int h = hash_index(floor_div_blk((int)roundf_ref(cx_)), floor_div_blk((int)roundf_ref(cy_)),
                                             floor_div_blk((int)roundf_ref(cz_)), s.n_buckets - 1);
                    word[j] = bits[h >> 5];
                    shift[j] = (uint32_t)h & 31u;
                }
                pin(hraw0);
                occupied = 1u;
#pragma unroll
                for (int j = 0; j < SKIP; j++) occupied |= ((word[j] >> shift[j]) & 1u) << (j + 1);
            }
        }
        const HashEntry head0 = decode_entry(hraw0);
        int base;
        if (cached) { vmIndex = 1; base = cache.ptr; }
        else base = resolve_with_head(s, kx0, ky0, kz0, head0, hidx0, vmIndex, cache);
        // The sample voxel round(p) is a corner of the interpolation cell [floor(p), floor(p)+1]^3.  If that cell lies
        // inside one block it is this block, so its eight voxels ride along with the sample in the same round trip and
        // the interpolated read below -- whose corner lookups would all hit the cache that now holds this block --
        // needs no memory access at all.
        const float ffx = floorf(px), ffy = floorf(py), ffz = floorf(pz);
        const int cix = (int)ffx, ciy = (int)ffy, ciz = (int)ffz;
        const bool cell_in_block = (cix & 7) < 7 && (ciy & 7) < 7 && (ciz & 7) < 7;
        uint64_t raw = EMPTY_VOXEL;
        uint64_t corner[8];
        if (base >= 0) {  // (skipped entirely by a wave whose live lanes are all in free space)
            raw = vox[base + lin];
            if (cell_in_block) {
                const uint64_t* v = vox + base + (cix & 7) + (ciy & 7) * BLK + (ciz & 7) * BLK * BLK;
                corner[0] = v[0]; corner[1] = v[1]; corner[2] = v[BLK]; corner[3] = v[BLK + 1];
                corner[4] = v[BLK * BLK]; corner[5] = v[BLK * BLK + 1]; corner[6] = v[BLK * BLK + BLK];
                corner[7] = v[BLK * BLK + BLK + 1];
            }
        }
        sdfValue = vox_sdf(raw) / 32767.0f;
        if (MODIFY_VISIBLE) { if (vmIndex) s.visible_type[vmIndex - 1] = 1; }  // incl. the vmIndex==1 cache-hit quirk
        n_log++;
        look = !vmIndex;
        if (!vmIndex) {
            stepLength = BLK;
            // advance over the candidates that are certainly unallocated steps inside the range (empty bucket head); the
            // first one that is anything else is sampled by the next iteration.  (One batch per iteration: an inner
            // run-to-completion loop would serialise the lanes.)
            int adv = 0;
            float cx_ = px, cy_ = py, cz_ = pz, cl_ = totalLength;
#pragma unroll
            for (int j = 1; j <= SKIP; j++) {
                cx_ += (float)BLK * rx; cy_ += (float)BLK * ry; cz_ += (float)BLK * rz; cl_ += (float)BLK;
                const bool plain = cl_ < totalLengthMax && !((occupied >> j) & 1u);
                if (adv == j - 1 && plain) { adv = j; px = cx_; py = cy_; pz = cz_; totalLength = cl_; }
            }
            n_log += (uint32_t)adv << 16;  // (each skipped candidate is one failed lookup = one step of the reference loop)
        } else {
            if ((sdfValue <= 0.1f) && (sdfValue >= -0.5f)) {
                float dummy;
                if (cell_in_block) {
                    sdfValue = blend_corners<false>(corner, px - ffx, py - ffy, pz - ffz, dummy);
                    vmIndex = 1;
                } else {
                    sdfValue = read_sdf_interp<false>(s, px, py, pz, vmIndex, cache, dummy);
                }
            }
            if (sdfValue <= 0.0f) break;
            const float a = sdfValue * stepScale;
            stepLength = (a < 1.0f) ? 1.0f : a;
        }
        px += stepLength * rx; py += stepLength * ry; pz += stepLength * rz;
        totalLength += stepLength;
    }
    bool found;
    if (sdfValue <= 0.0f) {
        stepLength = sdfValue * stepScale;
        px += stepLength * rx; py += stepLength * ry; pz += stepLength * rz;
        sdfValue = read_sdf_interp<true>(s, px, py, pz, vmIndex, cache, confidence);
        stepLength = sdfValue * stepScale;
        px += stepLength * rx; py += stepLength * ry; pz += stepLength * rz;
        found = true;
    } else {
        found = false;
    }
    if (inside) rays[x + y * W] = make_float4(px, py, pz, found ? confidence + 1.0f : 0.0f);
    // S-bar ("mean steps/ray logged by the kernel"): castRay steps as the reference's loop counts them (Shared.h:158-190:
    // one per lookup; the free-space look-ahead above folds up to 1 + SKIP of them into one trip), the trips of THIS loop (= voxel
    // reads) and the rays cast, summed over the wave (swap / DPP adds of small exact floats, no LDS) and stored as this wave's
    // row: plain stores, no atomics, no zero-fill (an atomic per workgroup on shared counters cost 8 us of the kernel's 115).
    // Valid for the LAST live or single-free-view launch on the SCENE's scratch (gps_tsdf_ray_stats reads those rows; a launch on
    // another stream that shares the scratch overwrites them).  The views of a batch are not logged (kernel-argument-uniform branch):
    // nobody reads their rows.
    if (views == nullptr) {
        const float f_reads = (float)(n_log & 0xFFFFu), f_steps = f_reads + (float)(n_log >> 16);
        const float tot = gps::reduce4(f_steps, f_reads, inside ? 1.0f : 0.0f, 0.0f);   // rows 0..3, lane 15: sums of (steps, rays, reads, 0)
        // (round 6) the wave's LONGEST ray in trips of this loop: the wave is resident until that lane leaves the loop -- is the launch as
        // long as its longest rays?  (tools/raycast_wave_hist.py; six shuffles per wave)
        int wmax = (int)(n_log & 0xFFFFu);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o, 64));
        float* row = reinterpret_cast<float*>(ray_stats_rows(s) + ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave_in_wg));
        if ((lane_ & 15) == 15 && lane_ < 48) row[lane_ == 15 ? 0 : lane_ == 31 ? 2 : 1] = tot;   // {steps, reads, rays}
        if (lane_ == 63) row[3] = (float)wmax;
    }
}

// the rows of the last raycast launch summed into counters[GPS_TSDF_RAY_STEPS / _READS / _RAYS] (three unsigned 64-bit sums)
__global__ __launch_bounds__(256) void ray_stats_sum_kernel(TsdfState s) {
    __shared__ double part[3][4];
    const float4* rows = ray_stats_rows(s);
    double a[3] = {0.0, 0.0, 0.0};
    const int n_rows = ((s.width + 2 * RC_PW - 1) / (2 * RC_PW)) * ((s.height + 2 * RC_PH - 1) / (2 * RC_PH)) * 4;   // waves of one raycast launch
    for (int i = threadIdx.x; i < n_rows; i += blockDim.x) { const float4 r = rows[i]; a[0] += r.x; a[1] += r.y; a[2] += r.z; }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[k] += __shfl_xor(a[k], o, 64);
        if ((threadIdx.x & 63) == 0) part[k][threadIdx.x >> 6] = a[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(s.counters + GPS_TSDF_RAY_STEPS);
        st[threadIdx.x] = (unsigned long long)(part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3]);
    }
}

// processPixelICP<useSmoothing = true, flipNormals = false> (Shared.h:252-330, 438-480)
__global__ __launch_bounds__(256) void icp_kernel(TsdfState s, Mat4 invM, const float4* __restrict__ pr,
                                                 float4* __restrict__ points, float4* __restrict__ normals) {
    GPS_FRAME_PRIO();
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    const int W = s.width, H = s.height;
    if (x >= W || y >= H) return;
    const int loc = x + y * W;
    const float4 point = pr[loc];
    bool found = point.w > 0.0f;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (found && (y <= 2 || y >= H - 3 || x <= 2 || x >= W - 3)) found = false;
    if (found) {
        float4 xp = pr[(x + 2) + y * W], yp = pr[x + (y + 2) * W], xm = pr[(x - 2) + y * W], ym = pr[x + (y - 2) * W];
        float dxx = 0.f, dxy = 0.f, dxz = 0.f, dyx = 0.f, dyy = 0.f, dyz = 0.f;
        bool doPlus1 = false;
        if (xp.w <= 0 || yp.w <= 0 || xm.w <= 0 || ym.w <= 0) doPlus1 = true;
        else {
            dxx = xp.x - xm.x; dxy = xp.y - xm.y; dxz = xp.z - xm.z;
            dyx = yp.x - ym.x; dyy = yp.y - ym.y; dyz = yp.z - ym.z;
            const float la = dxx * dxx + dxy * dxy + dxz * dxz, lb = dyx * dyx + dyy * dyy + dyz * dyz;
            const float ld = (la < lb) ? lb : la;
            if (ld * s.voxel_size * s.voxel_size > (0.15f * 0.15f)) doPlus1 = true;
        }
        if (doPlus1) {
            xp = pr[(x + 1) + y * W]; yp = pr[x + (y + 1) * W]; xm = pr[(x - 1) + y * W]; ym = pr[x + (y - 1) * W];
            dxx = xp.x - xm.x; dxy = xp.y - xm.y; dxz = xp.z - xm.z;
            dyx = yp.x - ym.x; dyy = yp.y - ym.y; dyz = yp.z - ym.z;
            if (xp.w <= 0 || yp.w <= 0 || xm.w <= 0 || ym.w <= 0) found = false;
        }
        if (found) {
            nx = -(dxy * dyz - dxz * dyy);
            ny = -(dxz * dyx - dxx * dyz);
            nz = -(dxx * dyy - dxy * dyx);
            const float ns = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
            nx *= ns; ny *= ns; nz *= ns;
            const float angle = nx * (-invM.m[8]) + ny * (-invM.m[9]) + nz * (-invM.m[10]);
            if (!(angle > 0.0)) found = false;
        }
    }
    if (found) {
        points[loc] = make_float4(point.x * s.voxel_size, point.y * s.voxel_size, point.z * s.voxel_size, point.w);
        normals[loc] = make_float4(nx, ny, nz, 0.0f);
    } else {
        const float4 o = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
        points[loc] = o; normals[loc] = o;
    }
}

// runRaycastByCam glue (slam/slam_pipeline.cpp:386-403 + cv_utils.cpp:322-341) in one pass:
// colour uchar4 -> float3 / 255; vertex = xyz * [w > 0] * voxel_size; confidence = w;
// depth = z of (w2c * [vertex, 1]) / w-row, forced to 0 where vertex.sum() == 0.
__device__ __forceinline__ void write_view_maps(int i, const float4 r, const uchar4 c, float voxel_size, const Mat4& w2c_rm,
                                                float* __restrict__ color_map, float* __restrict__ vertex_map,
                                                float* __restrict__ conf_map, float* __restrict__ depth_map,
                                                float* __restrict__ depth_clamped) {
    // .to(kFloat).div(255.0) (cv_utils.cpp:327): ATen divides a tensor by a host scalar as a multiplication with the float
    // reciprocal (BinaryDivTrueKernel: inv_b = 1 / b), which is NOT the IEEE quotient for every byte value -- reproduced
    const float inv255 = 1.0f / 255.0f;
    color_map[3 * i] = (float)c.x * inv255; color_map[3 * i + 1] = (float)c.y * inv255; color_map[3 * i + 2] = (float)c.z * inv255;
    const float keep = r.w > 0 ? 1.0f : 0.0f;
    const float vx = (r.x * keep) * voxel_size, vy = (r.y * keep) * voxel_size, vz = (r.z * keep) * voxel_size;
    vertex_map[3 * i] = vx; vertex_map[3 * i + 1] = vy; vertex_map[3 * i + 2] = vz;
    conf_map[i] = r.w;
    const float* m = w2c_rm.m;  // row-major here
    const float tz = m[8] * vx + m[9] * vy + m[10] * vz + m[11] * 1.0f;
    const float tw = m[12] * vx + m[13] * vy + m[14] * vz + m[15] * 1.0f;
    const float dz = ((vx + vy) + vz == 0.0f) ? 0.0f : tz / tw;
    depth_map[i] = dz;
    if (depth_clamped) depth_clamped[i] = dz < 0.01f ? 1000.0f : dz;
}

__global__ __launch_bounds__(256) void raycast_maps_kernel(int P, const float4* __restrict__ rays,
                                                          const uchar4* __restrict__ colour, float voxel_size, Mat4 w2c_rm,
                                                          float* __restrict__ color_map, float* __restrict__ vertex_map,
                                                          float* __restrict__ conf_map, float* __restrict__ depth_map,
                                                          float* __restrict__ depth_clamped) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    write_view_maps(i, rays[i], colour[i], voxel_size, w2c_rm, color_map, vertex_map, conf_map, depth_map, depth_clamped);
}

// readFromSDF_color4u_interpolated, GPS-SLAM variant renormalised over w_color >= 1
// (ITMRepresentationAccess.h:344-423) + drawPixelColour (Shared.h:384-394)
__global__ __launch_bounds__(256) void colour_kernel(TsdfState s, const float4* __restrict__ rays, uchar4* __restrict__ out,
                                                    const ViewRec* __restrict__ views) {
    GPS_FRAME_PRIO();
    const ViewRec* maps = nullptr;  // the view's tensor glue rides on this kernel (it holds the ray and the colour of the pixel)
    if (views) {
        apply_view(s, views[blockIdx.z]);
        rays = reinterpret_cast<const float4*>(s.fv_raycast); out = reinterpret_cast<uchar4*>(s.fv_colour);
        if (views[blockIdx.z].color_map) maps = views + blockIdx.z;
    }
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= s.width || y >= s.height) return;
    const int loc = x + y * s.width;
    const float4 r = rays[loc];
    if (!(r.w > 0)) {
        const uchar4 c = make_uchar4(0, 0, 0, 0);
        out[loc] = c;
        if (maps) write_view_maps(loc, r, c, s.voxel_size, maps->w2c_rm, maps->color_map, maps->vertex_map, maps->conf_map,
                                  maps->depth_map, maps->depth_clamped);
        return;
    }
    const float fx_ = floorf(r.x), fy_ = floorf(r.y), fz_ = floorf(r.z);
    const float cx = r.x - fx_, cy = r.y - fy_, cz = r.z - fz_;
    const int ix = (int)fx_, iy = (int)fy_, iz = (int)fz_;
    BlockCache c = {0x7fffffff, 0x7fffffff, 0x7fffffff, -1};
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, wsum = 0.f;
    int vm = 0;
    // the reference reads the eight corners one after the other through the block cache (up to 16 dependent round trips for a
    // cell that straddles blocks); the lookups have no side effect the colour depends on, so they go out as the raycaster's
    // interpolated read does: one lookup + eight voxels for a cell inside one block, eight heads then eight voxels otherwise
    uint64_t corner[8];
    fetch_cell(s, ix, iy, iz, vm, c, corner);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int dx = k & 1, dy = (k >> 1) & 1, dz = (k >> 2) & 1;
        const uint64_t raw = corner[k];
        if (((raw >> 48) & 0xFF) >= 1) {
            const float wx = dx ? cx : (1.0f - cx), wy = dy ? cy : (1.0f - cy), wz = dz ? cz : (1.0f - cz);
            const float w = wx * wy * wz;
            r0 += w * (float)((raw >> 24) & 0xFF);
            r1 += w * (float)((raw >> 32) & 0xFF);
            r2 += w * (float)((raw >> 40) & 0xFF);
            wsum += w;
        }
    }
    r0 /= wsum; r1 /= wsum; r2 /= wsum;
    const float c0 = r0 / 255.0f, c1 = r1 / 255.0f, c2 = r2 / 255.0f;
    const uchar4 px = make_uchar4((unsigned char)(c0 * 255.0f), (unsigned char)(c1 * 255.0f), (unsigned char)(c2 * 255.0f), 255);
    out[loc] = px;
    if (maps) write_view_maps(loc, r, px, s.voxel_size, maps->w2c_rm, maps->color_map, maps->vertex_map, maps->conf_map,
                              maps->depth_map, maps->depth_clamped);
}


// the table of a batch travels as a kernel argument (no staging buffer, no copy engine) into device memory
__global__ void upload_views_kernel(ViewTable t, int n, ViewRec* __restrict__ dst) {
    const int words = n * (int)(sizeof(ViewRec) / 4);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&t);
    for (int i = threadIdx.x; i < words; i += blockDim.x) reinterpret_cast<uint32_t*>(dst)[i] = src[i];
}

__global__ __launch_bounds__(256) void view_init_kernel(int P, float2* __restrict__ mm, int32_t* __restrict__ counters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) mm[i] = make_float2(FAR_AWAY, VERY_CLOSE);
    if (i < GPS_TSDF_N_COUNTERS) counters[i] = 0;
}

}  // namespace

namespace gpst { int find_visible_batch(const TsdfState& s, int n, const ViewRec* table, hipStream_t st); }

extern "C" {

int64_t gps_tsdf_view_table_bytes(int n_views) {
    return n_views > 0 && n_views <= MAX_BATCH_VIEWS ? (int64_t)n_views * (int64_t)sizeof(ViewRec) : GPS_ERR_ARG;
}

int gps_tsdf_view_init(const gps_tsdf_state* sp, const gps_tsdf_view* v, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp != nullptr && v != nullptr && v->minmax && v->counters);
    const int P = sp->width * sp->height;
    view_init_kernel<<<gps_div_up(max(P, (int)GPS_TSDF_N_COUNTERS), 256), 256, 0, (hipStream_t)stream>>>(
        P, reinterpret_cast<float2*>(v->minmax), v->counters);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_tsdf_free_raycast_batch(const gps_tsdf_state* sp, int n_views, const gps_tsdf_view* views, void* table,
                                gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp != nullptr && views != nullptr && table != nullptr && n_views > 0 && n_views <= MAX_BATCH_VIEWS);
    GPS_REQUIRE(state_valid(*sp));
    TsdfState s = *sp;
    hipStream_t st = (hipStream_t)stream;
    static_assert(sizeof(ViewTable) <= 3584, "the table is a kernel argument");
    ViewTable t = {};
    for (int k = 0; k < n_views; k++) {
        const gps_tsdf_view& v = views[k];
        GPS_REQUIRE(v.visible_ids && v.minmax && v.raycast && v.colour && v.scratch && v.counters);
        ViewRec& r = t.v[k];
        r.M = load_mat(v.M); r.invM = load_mat(v.invM);
        r.fx = v.fx; r.fy = v.fy; r.cx = v.cx; r.cy = v.cy;
        r.visible_ids = v.visible_ids; r.minmax = v.minmax; r.raycast = v.raycast; r.colour = v.colour;
        r.scratch = v.scratch; r.counters = v.counters;
        if (v.color_map) {
            GPS_REQUIRE(v.vertex_map && v.confidence_map && v.depth_map);
            r.w2c_rm = load_mat(v.w2c);
            r.color_map = v.color_map; r.vertex_map = v.vertex_map; r.conf_map = v.confidence_map; r.depth_map = v.depth_map;
            r.depth_clamped = v.depth_map_clamped;
        }
    }
    ViewRec* tab = reinterpret_cast<ViewRec*>(table);
    upload_views_kernel<<<1, 256, 0, st>>>(t, n_views, tab);
    int r;
    if ((r = gpst::find_visible_batch(s, n_views, tab, st)) != GPS_OK) return r;
    const int sw = s.width / MINMAX_SUB + 2, sh = s.height / MINMAX_SUB + 2;
    const size_t lds = (size_t)sw * sh * sizeof(uint2);
    GPS_REQUIRE(lds <= 160 * 1024);
    const Mat4 none = {};
    expected_depths_partial_kernel<<<dim3(ED_GROUPS, 1, n_views), ED_THREADS, lds, st>>>(s, none, nullptr, GPS_TSDF_N_VISIBLE_FREE, sw,
                                                                                     sh, nullptr, tab);
    const dim3 grid(gps_div_up(s.width, 2 * RC_PW), gps_div_up(s.height, 2 * RC_PH), n_views);   // (the raycaster's)
#ifdef GPS_ED_REDUCE_LAUNCH
    expected_depths_reduce_kernel<<<dim3(gps_div_up(sw * sh, 256), 1, n_views), 256, 0, st>>>(s, sw, sh, ED_GROUPS, nullptr, nullptr, tab);
    raycast_kernel<false><<<grid, 256, 0, st>>>(s, none, nullptr, nullptr, bucket_bits(s), tab, nullptr, 0, 0, nullptr, gps::LaunchStamp{nullptr});
#else
    // (pass B of the expected depths rides in the raycaster)
    raycast_kernel<false><<<grid, 256, 0, st>>>(s, none, nullptr, nullptr, bucket_bits(s), tab, nullptr, sw, sh, nullptr, gps::LaunchStamp{nullptr});
#endif
    colour_kernel<<<dim3(gps_div_up(s.width, 16), gps_div_up(s.height, 16), n_views), 256, 0, st>>>(s, nullptr, nullptr, tab);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

// pass A (+ pass B unless the raycaster that follows does it: gps_tsdf_expected_depths_and_raycast)
static int expected_depths_impl(const gps_tsdf_state* sp, const float* M, int free_view, bool with_reduce, gps_stream stream) {
    GPS_REQUIRE(sp != nullptr && M != nullptr);
    GPS_REQUIRE(state_valid(*sp));
    TsdfState s = *sp;
    hipStream_t st = (hipStream_t)stream;
    const int P = s.width * s.height;
    // window of the 1/8-resolution image that bounding boxes can touch: [0, W/8] x [0, H/8] (ceil of a coordinate
    // just inside the image) -- anything further is clipped (never read by the raycaster)
    const int sw = s.width / MINMAX_SUB + 2, sh = s.height / MINMAX_SUB + 2;
    const size_t lds = (size_t)sw * sh * sizeof(uint2);
    GPS_REQUIRE(lds <= 160 * 1024);
    const int n_total = s.n_buckets + s.n_excess;
    const int nblk = gps_div_up(n_total, 1024);
    // partial images live behind the sweep scratch (3*nblk + 16 ints + n_total flag bytes), see gps_tsdf_scratch_bytes
    uint2* partial = reinterpret_cast<uint2*>(s.scan_scratch + 3 * nblk + 16 + (n_total + 3) / 4 + 2);
    float2* mm = reinterpret_cast<float2*>(free_view ? s.fv_minmax : s.minmax);
    expected_depths_partial_kernel<<<ED_GROUPS, ED_THREADS, lds, st>>>(s, load_mat(M), free_view ? s.fv_visible_ids : s.visible_ids,
                                                                free_view ? GPS_TSDF_N_VISIBLE_FREE : GPS_TSDF_N_VISIBLE,
                                                                sw, sh, partial, nullptr);
    if (with_reduce) expected_depths_reduce_kernel<<<gps_div_up(sw * sh, 256), 256, 0, st>>>(s, sw, sh, ED_GROUPS, partial, mm, nullptr);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_tsdf_expected_depths(const gps_tsdf_state* sp, const float* M, int free_view, gps_stream stream) {
    GPS_ENTER();
    return expected_depths_impl(sp, M, free_view, true, stream);
}

int gps_tsdf_expected_depths_partial(const gps_tsdf_state* sp, const float* M, int free_view, gps_stream stream) {
    GPS_ENTER();
    return expected_depths_impl(sp, M, free_view, false, stream);
}

int64_t gps_tsdf_scratch_bytes(int width, int height, int n_buckets, int n_excess) {
    if (width <= 0 || height <= 0 || n_buckets <= 0 || n_excess <= 0) return GPS_ERR_ARG;
    const int64_t n_total = (int64_t)n_buckets + n_excess;
    const int64_t nblk = (n_total + 1023) / 1024;
    const int64_t sw = width / MINMAX_SUB + 2, sh = height / MINMAX_SUB + 2;
    (void)nblk; (void)sw; (void)sh;
    gps_tsdf_state t = {};
    t.width = width; t.height = height; t.n_buckets = n_buckets; t.n_excess = n_excess;
    return 4 * (ray_stats_offset_words(t) + 4 + 4 * (int64_t)ray_stat_waves(t));
}

static int raycast_impl(const gps_tsdf_state* sp, const float* invM, int free_view, int update_visible, bool reduce_here,
                        gps_stream stream) {
    GPS_REQUIRE(sp != nullptr && invM != nullptr);
    GPS_REQUIRE(state_valid(*sp));
    TsdfState s = *sp;
    dim3 grid(gps_div_up(s.width, 2 * RC_PW), gps_div_up(s.height, 2 * RC_PH));
    float2* mm = reinterpret_cast<float2*>(free_view ? s.fv_minmax : s.minmax);
    float4* rays = reinterpret_cast<float4*>(free_view ? s.fv_raycast : s.raycast);
    const int sw = s.width / MINMAX_SUB + 2, sh = s.height / MINMAX_SUB + 2;
    const int n_total = s.n_buckets + s.n_excess, nblk = gps_div_up(n_total, 1024);
    // (where expected_depths_impl put the partial images)
    const uint2* partial = reduce_here ? reinterpret_cast<const uint2*>(s.scan_scratch + 3 * nblk + 16 + (n_total + 3) / 4 + 2) : nullptr;
    const ViewRec* no_table = nullptr;
    if (update_visible)
        gps::launch_kernel(gps::TK_RAYCAST, 0, raycast_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, s, load_mat(invM), (const float2*)mm, rays,
                           bucket_bits(s), no_table, partial, sw, sh, mm);
    else
        gps::launch_kernel(gps::TK_RAYCAST, 1, raycast_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, s, load_mat(invM), (const float2*)mm, rays,
                           bucket_bits(s), no_table, partial, sw, sh, mm);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_tsdf_raycast(const gps_tsdf_state* sp, const float* invM, int free_view, int update_visible, gps_stream stream) {
    GPS_ENTER();
    return raycast_impl(sp, invM, free_view, update_visible, false, stream);
}

// The two calls of the frame chain as one: pass B of the expected depths is done by the raycaster's waves (same images, same
// counters afterwards, one launch less).
int gps_tsdf_expected_depths_and_raycast(const gps_tsdf_state* sp, const float* M, const float* invM, int free_view,
                                         int update_visible, gps_stream stream) {
    GPS_ENTER();
#ifdef GPS_ED_REDUCE_LAUNCH   // (probe builds: the three-launch form, for an A/B)
    const int r = expected_depths_impl(sp, M, free_view, true, stream);
    return r != GPS_OK ? r : raycast_impl(sp, invM, free_view, update_visible, false, stream);
#else
    const int r = expected_depths_impl(sp, M, free_view, false, stream);
    return r != GPS_OK ? r : raycast_impl(sp, invM, free_view, update_visible, true, stream);
#endif
}

int gps_tsdf_ray_stats(const gps_tsdf_state* sp, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp != nullptr);
    GPS_REQUIRE(state_valid(*sp));
    ray_stats_sum_kernel<<<1, 256, 0, (hipStream_t)stream>>>(*sp);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_tsdf_ray_wave_rows(const gps_tsdf_state* sp, float* rows_out, int capacity_rows, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp != nullptr && rows_out != nullptr && capacity_rows >= 0);
    GPS_REQUIRE(state_valid(*sp));
    const TsdfState& s = *sp;
    const int n_rows = ((s.width + 2 * RC_PW - 1) / (2 * RC_PW)) * ((s.height + 2 * RC_PH - 1) / (2 * RC_PH)) * 4;   // waves of one raycast launch
    const int n = n_rows < capacity_rows ? n_rows : capacity_rows;
    if (n > 0 && hipMemcpyAsync(rows_out, ray_stats_rows(s), (size_t)n * 16, hipMemcpyDefault, (hipStream_t)stream) != hipSuccess) return GPS_ERR_LAUNCH;
    return n;
}

int gps_tsdf_icp_maps(const gps_tsdf_state* sp, const float* invM, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp != nullptr && invM != nullptr);
    GPS_REQUIRE(state_valid(*sp));
    TsdfState s = *sp;
    dim3 grid(gps_div_up(s.width, 16), gps_div_up(s.height, 16));
    icp_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(s, load_mat(invM), reinterpret_cast<const float4*>(s.raycast),
                                                     reinterpret_cast<float4*>(s.icp_points),
                                                     reinterpret_cast<float4*>(s.icp_normals));
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_tsdf_render_colour(const gps_tsdf_state* sp, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp != nullptr);
    GPS_REQUIRE(state_valid(*sp));
    TsdfState s = *sp;
    dim3 grid(gps_div_up(s.width, 16), gps_div_up(s.height, 16));
    colour_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(s, reinterpret_cast<const float4*>(s.fv_raycast),
                                                        reinterpret_cast<uchar4*>(s.fv_colour), nullptr);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_tsdf_process_frame(const gps_tsdf_state* s, const int16_t* depth_mm, const float* M, const float* invM,
                           gps_stream stream) {
    int r;
    if ((r = gps_tsdf_convert_depth(s, depth_mm, stream)) != GPS_OK) return r;
    if ((r = gps_tsdf_allocate(s, M, invM, stream)) != GPS_OK) return r;
    if ((r = gps_tsdf_integrate(s, M, stream)) != GPS_OK) return r;
    if ((r = gps_tsdf_expected_depths_and_raycast(s, M, invM, 0, 1, stream)) != GPS_OK) return r;
    return gps_tsdf_icp_maps(s, invM, stream);
}

int gps_tsdf_free_raycast(const gps_tsdf_state* s, const float* M, const float* invM, gps_stream stream) {
    int r;
    if ((r = gps_tsdf_find_visible(s, M, stream)) != GPS_OK) return r;
    if ((r = gps_tsdf_expected_depths_and_raycast(s, M, invM, 1, 0, stream)) != GPS_OK) return r;
    return gps_tsdf_render_colour(s, stream);
}

int gps_raycast_to_maps(int width, int height, const float* rays, const uint8_t* colour, float voxel_size,
                        const float* w2c_row_major, float* color_map, float* vertex_map, float* confidence_map,
                        float* depth_map, float* depth_map_clamped, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(width > 0 && height > 0 && rays && colour && w2c_row_major && color_map && vertex_map && confidence_map && depth_map);
    const int P = width * height;
    raycast_maps_kernel<<<gps_div_up(P, 256), 256, 0, (hipStream_t)stream>>>(
        P, reinterpret_cast<const float4*>(rays), reinterpret_cast<const uchar4*>(colour), voxel_size,
        load_mat(w2c_row_major), color_map, vertex_map, confidence_map, depth_map, depth_map_clamped);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_pose_from_c2w(const float* c2w_row_major, float* M, float* invM) {
    if (!c2w_row_major || !M || !invM) return GPS_ERR_ARG;
    float c2w[16], M0[16], prm[6];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) c2w[c * 4 + r] = c2w_row_major[r * 4 + c];
    if (!mat4_inverse(c2w, M0)) return GPS_ERR_ARG;  // SetInvM
    pose_log(M0, prm);                               // SetParamsFromModelView (twice: SetInvM, then Coerce)
    pose_log(M0, prm);
    pose_exp(prm, M);                                // SetModelViewFromParams
    if (!mat4_inverse(M, invM)) return GPS_ERR_ARG;  // GetInvM
    return GPS_OK;
}

}  // extern "C"

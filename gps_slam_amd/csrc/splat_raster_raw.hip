// `raw` render method: classic front-to-back alpha compositing over depth-sorted tile lists.
//   rasterize_to_pixels_fwd.cu:18-203  -> raster_raw_fwd_kernel
//   rasterize_to_pixels_bwd.cu:20-297  -> raster_raw_bwd_kernel
// COLOR_DIM = 4 (rgb + depth, raw_gs_model.cpp:117), one camera, no tile masks; backgrounds optional.
//
// Both kernels: one workgroup (4 wave64) per 16x16 tile, one pixel per lane, the tile's depth-sorted list staged through
// LDS in batches of 256 records.  Unlike the order-independent `ges` sum, compositing is a sequential recurrence in T per
// pixel, so the inner loop keeps the reference's structure; what changes for gfx950 is the gradient reduction of the
// backward: per Gaussian the 10 partial gradients of a wave's 64 pixels are summed with DPP adds (no LDS crossbar), the 4
// waves of the tile accumulate into an LDS gradient tile with ds_add_f32, and ONE pass per batch adds the tile's 256 x 10
// totals to global memory -- 10 global atomics per (tile, Gaussian) pair instead of the reference's 10 per (warp, Gaussian)
// (8 warps per tile).
#include "common.hpp"
#include "wave_reduce.hpp"

namespace {

using gps::reduce4;
using gps::sum_halves;
using gps::row_sum_to_lane15;
using gps::dpp_add_masked;

constexpr int RAW_BATCH = 256;

struct RawRec {  // 48 bytes
    float x, y, opac, ca;
    float cb, cc, c0, c1;
    float c2, c3;
    int id, pad;
};

__device__ __forceinline__ RawRec load_rec(int g, const float2* __restrict__ means2d, const float* __restrict__ conics,
                                           const float4* __restrict__ colors, const float* __restrict__ opacities) {
    RawRec r;
    const float2 xy = means2d[g];
    const float4 c = colors[g];
    r.x = xy.x; r.y = xy.y; r.opac = opacities[g];
    r.ca = conics[3 * g]; r.cb = conics[3 * g + 1]; r.cc = conics[3 * g + 2];
    r.c0 = c.x; r.c1 = c.y; r.c2 = c.z; r.c3 = c.w;
    r.id = g; r.pad = 0;
    return r;
}

__global__ __launch_bounds__(256) void raster_raw_fwd_kernel(
    const float2* __restrict__ means2d, const float* __restrict__ conics, const float4* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ backgrounds, int W, int H, int tw, int th,
    const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids, const int64_t* __restrict__ counts,
    float4* __restrict__ render_colors, float* __restrict__ render_alphas, int32_t* __restrict__ last_ids) {
    __shared__ RawRec recs[RAW_BATCH];
    const int tile_id = blockIdx.x;
    const int ty = tile_id / tw, tx = tile_id - ty * tw;
    const int tid = threadIdx.x;
    // one 8x8 pixel patch per wave: compact footprints make the wave-uniform skips (all lanes done / no lane touched) fire
    const int i = ty * 16 + ((tid >> 7) << 3) + ((tid >> 3) & 7), j = tx * 16 + (((tid >> 6) & 1) << 3) + (tid & 7);
    const bool inside = (i < H) && (j < W);
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    const int n_isects = (int)counts[0];
    const int range_start = tile_offsets[tile_id];
    const int range_end = (tile_id == tw * th - 1) ? n_isects : tile_offsets[tile_id + 1];
    bool done = !inside;
    float T = 1.0f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
    uint32_t cur_idx = 0;
    for (int batch_start = range_start; batch_start < range_end; batch_start += RAW_BATCH) {
        if (__syncthreads_count(done) >= (int)blockDim.x) break;  // also the barrier before the LDS tile is overwritten
        const int idx = batch_start + tid;
        if (idx < range_end) recs[tid] = load_rec(flatten_ids[idx], means2d, conics, colors, opacities);
        else recs[tid] = RawRec{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, -1, 0};  // opacity 0: never contributes
        __syncthreads();
        const int n = min(RAW_BATCH, range_end - batch_start);
        if (__all(done)) continue;  // this wave's patch is finished; keep taking part in the barriers
        // Branch-free body (a finished or untouched pixel adds weight 0), wave-uniform exit test every 4 Gaussians.  The
        // reference compares the float next_T with the double literal 1e-4; the largest float that passes is 1e-4f.
        for (int t0 = 0; t0 < n; t0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u;  // may run up to 3 past n: those slots hold null records
                const RawRec r = recs[t];
                const float dx = r.x - px, dy = r.y - py;
                const float sigma = 0.5f * (r.ca * dx * dx + r.cc * dy * dy) + r.cb * dx * dy;
                const float alpha = fminf(0.999f, r.opac * __expf(-sigma));
                const bool ok = !done && sigma >= 0.f && alpha >= 1.f / 255.f;
                const float next_T = T * (1.0f - alpha);
                const bool stop = ok && next_T <= 1e-4f;
                const bool use = ok && !stop;
                done = done || stop;
                const float vis = use ? alpha * T : 0.f;
                o0 += r.c0 * vis; o1 += r.c1 * vis; o2 += r.c2 * vis; o3 += r.c3 * vis;
                cur_idx = use ? (uint32_t)(batch_start + t) : cur_idx;
                T = use ? next_T : T;
            }
            if (__all(done)) break;
        }
    }
    if (inside) {
        const int pix = i * W + j;
        render_alphas[pix] = 1.0f - T;
        if (backgrounds) { o0 += T * backgrounds[0]; o1 += T * backgrounds[1]; o2 += T * backgrounds[2]; o3 += T * backgrounds[3]; }
        render_colors[pix] = make_float4(o0, o1, o2, o3);
        last_ids[pix] = (int32_t)cur_idx;
    }
}

// ---- wave64 reduction of the per-pixel gradient vector, 12 values at a time
// gfx950's v_permlane32_swap / v_permlane16_swap exchange half-waves / 16-lane rows between TWO registers in one
// instruction, so "swap, add" halves the lane span of two values at once: 12 values -> 6 registers (two values each, one
// per half-wave) -> 3 registers (four values each, one per row) in 9 swaps + 9 adds; the remaining sum inside each 16-lane
// row is 4 DPP row_shr adds per register.  30 VALU ops per Gaussian instead of 72 for twelve full DPP reductions, and the
// totals come out in lane 15 of each row: 3 ds_add_f32 (4 lanes each) instead of 12.
// (helpers: wave_reduce.hpp)

__global__ __launch_bounds__(256) void zero_raw_grads_kernel(int N, float* __restrict__ v_means2d, float* __restrict__ v_conics,
                                                            float* __restrict__ v_colors, float* __restrict__ v_opacities,
                                                            float* __restrict__ v_abs) {
    const int stride = gridDim.x * blockDim.x;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < 4 * N; k += stride) {
        v_colors[k] = 0.f;
        if (k < 3 * N) v_conics[k] = 0.f;
        if (k < 2 * N) { v_means2d[k] = 0.f; if (v_abs) v_abs[k] = 0.f; }
        if (k < N) v_opacities[k] = 0.f;
    }
}

template <bool ABS>
__global__ __launch_bounds__(256) void raster_raw_bwd_kernel(
    const float2* __restrict__ means2d, const float* __restrict__ conics, const float4* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ backgrounds, int W, int H, int tw, int th,
    const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids, const int64_t* __restrict__ counts,
    const float* __restrict__ render_alphas, const int32_t* __restrict__ last_ids, const float4* __restrict__ v_render_colors,
    const float* __restrict__ v_render_alphas, float* __restrict__ v_means2d_abs, float* __restrict__ v_means2d,
    float* __restrict__ v_conics, float* __restrict__ v_colors, float* __restrict__ v_opacities) {
    constexpr int NG = 12;  // rgb d | conic abc | xy | opac | (|x| |y|)
    __shared__ RawRec recs[RAW_BATCH];
    __shared__ float grad[RAW_BATCH][NG];
    __shared__ int tile_last;
    const int tile_id = blockIdx.x;
    const int ty = tile_id / tw, tx = tile_id - ty * tw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int row_slot = ((lane >> 4) & 1) * 2 + (lane >> 5);  // which value of a reduce4 group this lane's row ends up with
    // one 8x8 pixel patch per wave: compact footprints make the wave-uniform skips (all lanes done / no lane touched) fire
    const int i = ty * 16 + ((tid >> 7) << 3) + ((tid >> 3) & 7), j = tx * 16 + (((tid >> 6) & 1) << 3) + (tid & 7);
    const bool inside = (i < H) && (j < W);
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    const int pix = inside ? i * W + j : 0;
    const int n_isects = (int)counts[0];
    const int range_start = tile_offsets[tile_id];
    const int range_end = (tile_id == tw * th - 1) ? n_isects : tile_offsets[tile_id + 1];
    if (range_end <= range_start) return;
    const float T_final = inside ? 1.0f - render_alphas[pix] : 1.0f;
    float T = T_final;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;  // colour accumulated BEHIND the current Gaussian
    const int bin_final = inside ? last_ids[pix] : -1;
    const float4 vc = inside ? v_render_colors[pix] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float va = inside ? v_render_alphas[pix] : 0.f;
    float bg_dot = 0.f;
    if (backgrounds) bg_dot = backgrounds[0] * vc.x + backgrounds[1] * vc.y + backgrounds[2] * vc.z + backgrounds[3] * vc.w;
    // nothing behind the tile's furthest contributing Gaussian matters: start the walk there
    if (tid == 0) tile_last = -1;
    __syncthreads();
    {
        int m = bin_final;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
        if (lane == 0) atomicMax(&tile_last, m);
    }
    __syncthreads();
    const int walk_end = min(range_end - 1, tile_last);  // inclusive
    for (int batch_end = walk_end; batch_end >= range_start; batch_end -= RAW_BATCH) {
        __syncthreads();
        const int n = min(RAW_BATCH, batch_end + 1 - range_start);
        if (tid < n) recs[tid] = load_rec(flatten_ids[batch_end - tid], means2d, conics, colors, opacities);  // [0] = furthest back
        for (int k = tid; k < RAW_BATCH * NG; k += 256) (&grad[0][0])[k] = 0.f;
        __syncthreads();
        for (int t = 0; t < n; ++t) {
            const RawRec r = recs[t];
            const float dx = r.x - px, dy = r.y - py;
            const float sigma = 0.5f * (r.ca * dx * dx + r.cc * dy * dy) + r.cb * dx * dy;
            const float vis_raw = __expf(-sigma);
            const float alpha_raw = fminf(0.999f, r.opac * vis_raw);
            const bool valid = inside && (batch_end - t <= bin_final) && sigma >= 0.f && alpha_raw >= 1.f / 255.f;
            if (!__any(valid)) continue;  // wave uniform: no pixel of this 8x8 patch is touched
            // Branch-free from here: a pixel the Gaussian does not contribute to runs with alpha = vis = 0, which leaves
            // T and the colour behind untouched and makes all of its gradient terms exact zeros.
            const float alpha = valid ? alpha_raw : 0.f;
            const float vis = valid ? vis_raw : 0.f;
            const float d = 1.0f - alpha;
            float ra = __builtin_amdgcn_rcpf(d);
            ra = fmaf(fmaf(-d, ra, 1.0f), ra, ra);  // one Newton step: 1/(1-alpha) to rounding, 3 ops instead of IEEE div's 11
            T *= ra;
            const float fac = alpha * T;
            const float g0 = fac * vc.x, g1 = fac * vc.y, g2 = fac * vc.z, g3 = fac * vc.w;
            float v_alpha = (r.c0 * T - b0 * ra) * vc.x + (r.c1 * T - b1 * ra) * vc.y + (r.c2 * T - b2 * ra) * vc.z +
                            (r.c3 * T - b3 * ra) * vc.w;
            v_alpha += T_final * ra * (va - bg_dot);  // bg_dot = 0 without backgrounds
            const float ov = r.opac * vis;
            const bool unclamped = ov <= 0.999f;
            const float v_sigma = unclamped ? -ov * v_alpha : 0.f;
            const float g4 = 0.5f * v_sigma * dx * dx;
            const float g5 = v_sigma * dx * dy;
            const float g6 = 0.5f * v_sigma * dy * dy;
            const float g7 = v_sigma * (r.ca * dx + r.cb * dy);
            const float g8 = v_sigma * (r.cb * dx + r.cc * dy);
            const float g9 = unclamped ? vis * v_alpha : 0.f;
            b0 += r.c0 * fac; b1 += r.c1 * fac; b2 += r.c2 * fac; b3 += r.c3 * fac;
            // rows of z0 / z1 hold the totals of (g0,g2,g1,g3) / (g4,g6,g5,g7) in lane 15; ds_add_f32: the 4 waves of the
            // tile meet in the LDS gradient tile
            const float z0 = reduce4(g0, g1, g2, g3);
            const float z1 = reduce4(g4, g5, g6, g7);
            if ((lane & 15) == 15) {
                float* gt = &grad[t][row_slot];
                atomicAdd(gt, z0);
                atomicAdd(gt + 4, z1);
            }
            if (ABS) {
                const float z2 = reduce4(g8, g9, fabsf(g7), fabsf(g8));  // rows: g8 |g7| g9 |g8| -> slots 8 10 9 11
                if ((lane & 15) == 15) atomicAdd(&grad[t][8 + row_slot], z2);
            } else {
                // two values left: one per half-wave, row sums, then row 0 -> 1 and row 2 -> 3 (row_bcast15)
                const float z2 = dpp_add_masked<0x142, 0xA>(row_sum_to_lane15(sum_halves(g8, g9)));
                if ((lane & 31) == 31) atomicAdd(&grad[t][8 + (lane >> 5)], z2);
            }
        }
        __syncthreads();
        // one pass per batch: the tile's totals go to global memory, 10 (12) atomics per Gaussian that received anything
        if (tid < n) {
            const int gid = recs[tid].id;
            const float* gt = grad[tid];
            float* vcol = v_colors + 4 * (size_t)gid;
            float* vk = v_conics + 3 * (size_t)gid;
            float* vm = v_means2d + 2 * (size_t)gid;
            if (gt[0] != 0.f) atomicAdd(vcol + 0, gt[0]);
            if (gt[1] != 0.f) atomicAdd(vcol + 1, gt[1]);
            if (gt[2] != 0.f) atomicAdd(vcol + 2, gt[2]);
            if (gt[3] != 0.f) atomicAdd(vcol + 3, gt[3]);
            if (gt[4] != 0.f) atomicAdd(vk + 0, gt[4]);
            if (gt[5] != 0.f) atomicAdd(vk + 1, gt[5]);
            if (gt[6] != 0.f) atomicAdd(vk + 2, gt[6]);
            if (gt[7] != 0.f) atomicAdd(vm + 0, gt[7]);
            if (gt[8] != 0.f) atomicAdd(vm + 1, gt[8]);
            if (gt[9] != 0.f) atomicAdd(v_opacities + gid, gt[9]);
            if (ABS) {
                if (gt[10] != 0.f) atomicAdd(v_means2d_abs + 2 * (size_t)gid + 0, gt[10]);
                if (gt[11] != 0.f) atomicAdd(v_means2d_abs + 2 * (size_t)gid + 1, gt[11]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Exact tile-parallel adjoint of the `ges` forward (rasterize_to_pixels_bwd_ges.cu:18-291): what the reference's
// RasterizeToPixelsGes autograd Function runs (gsplat_wapper.hpp:355-487; the shipped models use the Gaussian-parallel
// box backward of splat_raster.hip instead).  Same tile / patch geometry and the same swap-based wave reduction + LDS
// gradient tile as raster_raw_bwd_kernel above; the order-independent sum has no transmittance recurrence, so the tile's
// list is walked front to back and a pixel's terms depend on that Gaussian alone.
__global__ __launch_bounds__(256) void raster_ges_bwd_exact_kernel(
    const float2* __restrict__ means2d, const float* __restrict__ conics, const float4* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ ref_depth, float delta_depth, int W, int H, int tw, int th,
    const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids, const int64_t* __restrict__ counts,
    const float4* __restrict__ v_render_colors, const float* __restrict__ v_render_alphas, float* __restrict__ v_means2d,
    float* __restrict__ v_conics, float* __restrict__ v_colors, float* __restrict__ v_opacities) {
    constexpr int NG = 12;
    __shared__ RawRec recs[RAW_BATCH];
    __shared__ float grad[RAW_BATCH][NG];
    const int tile_id = blockIdx.x;
    const int ty = tile_id / tw, tx = tile_id - ty * tw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int row_slot = ((lane >> 4) & 1) * 2 + (lane >> 5);
    const int i = ty * 16 + ((tid >> 7) << 3) + ((tid >> 3) & 7), j = tx * 16 + (((tid >> 6) & 1) << 3) + (tid & 7);
    const bool inside = (i < H) && (j < W);
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    const int pix = inside ? i * W + j : 0;
    const int n_isects = (int)counts[0];
    const int range_start = tile_offsets[tile_id];
    const int range_end = (tile_id == tw * th - 1) ? n_isects : tile_offsets[tile_id + 1];
    if (range_end <= range_start) return;
    const float cut = inside ? ref_depth[pix] + delta_depth : -3.0e38f;
    const float4 vc = inside ? v_render_colors[pix] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float va = inside ? v_render_alphas[pix] : 0.f;
    for (int batch_start = range_start; batch_start < range_end; batch_start += RAW_BATCH) {
        __syncthreads();
        const int n = min(RAW_BATCH, range_end - batch_start);
        if (tid < n) recs[tid] = load_rec(flatten_ids[batch_start + tid], means2d, conics, colors, opacities);
        for (int k = tid; k < RAW_BATCH * NG; k += 256) (&grad[0][0])[k] = 0.f;
        __syncthreads();
        for (int t = 0; t < n; ++t) {
            const RawRec r = recs[t];
            const float dx = r.x - px, dy = r.y - py;
            const float sigma = 0.5f * (r.ca * dx * dx + r.cc * dy * dy) + r.cb * dx * dy;
            const float vis_raw = __expf(-sigma);
            const float alpha_raw = fminf(0.999f, r.opac * vis_raw);
            const bool valid = inside && !(r.c3 > cut) && sigma >= 0.f && alpha_raw >= 1.f / 255.f;
            if (!__any(valid)) continue;  // wave uniform
            const float alpha = valid ? alpha_raw : 0.f;
            const float vis = valid ? vis_raw : 0.f;
            const float g0 = alpha * vc.x, g1 = alpha * vc.y, g2 = alpha * vc.z, g3 = alpha * vc.w;
            const float v_alpha = r.c0 * vc.x + r.c1 * vc.y + r.c2 * vc.z + r.c3 * vc.w + va;
            const float ov = r.opac * vis;
            const bool unclamped = ov <= 0.999f;
            const float v_sigma = unclamped ? -ov * v_alpha : 0.f;
            const float g4 = 0.5f * v_sigma * dx * dx;
            const float g5 = v_sigma * dx * dy;
            const float g6 = 0.5f * v_sigma * dy * dy;
            const float g7 = v_sigma * (r.ca * dx + r.cb * dy);
            const float g8 = v_sigma * (r.cb * dx + r.cc * dy);
            const float g9 = unclamped ? vis * v_alpha : 0.f;
            const float z0 = reduce4(g0, g1, g2, g3);
            const float z1 = reduce4(g4, g5, g6, g7);
            if ((lane & 15) == 15) {
                float* gt = &grad[t][row_slot];
                atomicAdd(gt, z0);
                atomicAdd(gt + 4, z1);
            }
            const float z2 = dpp_add_masked<0x142, 0xA>(row_sum_to_lane15(sum_halves(g8, g9)));
            if ((lane & 31) == 31) atomicAdd(&grad[t][8 + (lane >> 5)], z2);
        }
        __syncthreads();
        if (tid < n) {
            const int gid = recs[tid].id;
            const float* gt = grad[tid];
            float* vcol = v_colors + 4 * (size_t)gid;
            float* vk = v_conics + 3 * (size_t)gid;
            float* vm = v_means2d + 2 * (size_t)gid;
            if (gt[0] != 0.f) atomicAdd(vcol + 0, gt[0]);
            if (gt[1] != 0.f) atomicAdd(vcol + 1, gt[1]);
            if (gt[2] != 0.f) atomicAdd(vcol + 2, gt[2]);
            if (gt[3] != 0.f) atomicAdd(vcol + 3, gt[3]);
            if (gt[4] != 0.f) atomicAdd(vk + 0, gt[4]);
            if (gt[5] != 0.f) atomicAdd(vk + 1, gt[5]);
            if (gt[6] != 0.f) atomicAdd(vk + 2, gt[6]);
            if (gt[7] != 0.f) atomicAdd(vm + 0, gt[7]);
            if (gt[8] != 0.f) atomicAdd(vm + 1, gt[8]);
            if (gt[9] != 0.f) atomicAdd(v_opacities + gid, gt[9]);
        }
    }
}

}  // namespace

extern "C" {

int gps_raster_ges_bwd_exact(int N, const float* means2d, const float* conics, const float* colors, const float* opacities,
                             const float* ref_depth_map, int width, int height, int tile_size, const int32_t* tile_offsets,
                             const int32_t* flatten_ids, const int64_t* counts, float delta_depth, const float* v_render_colors,
                             const float* v_render_alphas, float* v_means2d, float* v_conics, float* v_colors,
                             float* v_opacities, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(N >= 0 && width > 0 && height > 0);
    GPS_REQUIRE(tile_size == 16);
    if (N == 0) return GPS_OK;
    GPS_REQUIRE(means2d && conics && colors && opacities && ref_depth_map && tile_offsets && flatten_ids && counts &&
                v_render_colors && v_render_alphas && v_means2d && v_conics && v_colors && v_opacities);
    const int tw = gps_div_up(width, 16), th = gps_div_up(height, 16);
    hipStream_t s = (hipStream_t)stream;
    zero_raw_grads_kernel<<<min(2048, gps_div_up(4 * (int64_t)N, 256)), 256, 0, s>>>(N, v_means2d, v_conics, v_colors, v_opacities,
                                                                                    nullptr);
    raster_ges_bwd_exact_kernel<<<tw * th, 256, 0, s>>>((const float2*)means2d, conics, (const float4*)colors, opacities,
                                                       ref_depth_map, delta_depth, width, height, tw, th, tile_offsets,
                                                       flatten_ids, counts, (const float4*)v_render_colors, v_render_alphas,
                                                       v_means2d, v_conics, v_colors, v_opacities);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_raster_raw_fwd(int N, const float* means2d, const float* conics, const float* colors, const float* opacities,
                       const float* backgrounds, int width, int height, int tile_size, const int32_t* tile_offsets,
                       const int32_t* flatten_ids, const int64_t* counts, float* render_colors, float* render_alphas,
                       int32_t* last_ids, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(N >= 0 && width > 0 && height > 0);
    GPS_REQUIRE(tile_size == 16);
    GPS_REQUIRE(tile_offsets && flatten_ids && counts && render_colors && render_alphas && last_ids);
    GPS_REQUIRE(N == 0 || (means2d && conics && colors && opacities));
    const int tw = gps_div_up(width, 16), th = gps_div_up(height, 16);
    raster_raw_fwd_kernel<<<tw * th, 256, 0, (hipStream_t)stream>>>(
        (const float2*)means2d, conics, (const float4*)colors, opacities, backgrounds, width, height, tw, th, tile_offsets,
        flatten_ids, counts, (float4*)render_colors, render_alphas, last_ids);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_raster_raw_bwd(int N, const float* means2d, const float* conics, const float* colors, const float* opacities,
                       const float* backgrounds, int width, int height, int tile_size, const int32_t* tile_offsets,
                       const int32_t* flatten_ids, const int64_t* counts, const float* render_alphas, const int32_t* last_ids,
                       const float* v_render_colors, const float* v_render_alphas, float* v_means2d_abs, float* v_means2d,
                       float* v_conics, float* v_colors, float* v_opacities, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(N >= 0 && width > 0 && height > 0);
    GPS_REQUIRE(tile_size == 16);
    if (N == 0) return GPS_OK;
    GPS_REQUIRE(means2d && conics && colors && opacities && tile_offsets && flatten_ids && counts && render_alphas && last_ids &&
                v_render_colors && v_render_alphas && v_means2d && v_conics && v_colors && v_opacities);
    const int tw = gps_div_up(width, 16), th = gps_div_up(height, 16);
    hipStream_t s = (hipStream_t)stream;
    zero_raw_grads_kernel<<<min(2048, gps_div_up(4 * (int64_t)N, 256)), 256, 0, s>>>(N, v_means2d, v_conics, v_colors, v_opacities,
                                                                                    v_means2d_abs);
#define GPS_RAW_BWD_ARGS (const float2*)means2d, conics, (const float4*)colors, opacities, backgrounds, width, height, tw, th, \
                         tile_offsets, flatten_ids, counts, render_alphas, last_ids, (const float4*)v_render_colors,          \
                         v_render_alphas, v_means2d_abs, v_means2d, v_conics, v_colors, v_opacities
    if (v_means2d_abs) raster_raw_bwd_kernel<true><<<tw * th, 256, 0, s>>>(GPS_RAW_BWD_ARGS);
    else raster_raw_bwd_kernel<false><<<tw * th, 256, 0, s>>>(GPS_RAW_BWD_ARGS);
#undef GPS_RAW_BWD_ARGS
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

}  // extern "C"

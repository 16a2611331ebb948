// Gaussian-creation helpers on the every-10-frames path (SLAMGaussianModel::addGaussians ->
// RawGaussianParams::init, src/raw_gs_param.cpp:11-74; SLAMPipeline::initNewGaussians, slam_pipeline.cpp:503):
//
// gps_knn_mean_dist2 <- distCUDA2 (gsplat/rasterizer/simple_knn.cu:191-240): mean squared distance to the 3 nearest
//                       neighbours.  The reference builds a Morton order + 1024-point boxes with cub/thrust,
//                       cudaMalloc/cudaFree and two blocking memcpys per call; P is 1e2..1e4 on a steady keyframe, where an
//                       exact LDS-tiled brute force (P^2 / 2^24 lane-steps) is both simpler and faster and needs no
//                       scratch memory.  Same result: the 3 smallest squared distances are a set property.  Larger sets
//                       (a first keyframe, a newly revealed room): the exact uniform-grid search of splat_knn.hip.
// gps_normal_map     <- computeNormalMap (src/tensor_math.cpp:217-248, 278-300): Sobel gradients of the vertex map
//                       with replicate padding, cross(dy, dx), normalise, zero where vertex z <= 0.
#include <float.h>

#include "common.hpp"
#include "splat_knn.hpp"

namespace {

constexpr int KNN_TILE = 256;

using gps::keep3;

// 16 queries x 16 candidate slices per workgroup: thread (q, slice) scans the candidates t == slice (mod 16) of every
// 256-point LDS tile, so P points give P/16 workgroups (P is only 1e3..1e4 on the per-keyframe path -- one thread per
// query would leave most of the chip idle); the 16 partial triples of a query are merged through LDS.
constexpr int KNN_Q = 16, KNN_S = 16;

__global__ __launch_bounds__(KNN_TILE) void knn_kernel(int P, const float* __restrict__ pts, float* __restrict__ out) {
    __shared__ float sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE];
    __shared__ float best[KNN_S][KNN_Q][3];
    const int ql = threadIdx.x & (KNN_Q - 1), slice = threadIdx.x / KNN_Q;
    const int q = blockIdx.x * KNN_Q + ql;
    const bool live = q < P;
    const float qx = live ? pts[3 * q] : 0.f, qy = live ? pts[3 * q + 1] : 0.f, qz = live ? pts[3 * q + 2] : 0.f;
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    for (int base = 0; base < P; base += KNN_TILE) {
        const int j = base + threadIdx.x;
        __syncthreads();
        if (j < P) { sx[threadIdx.x] = pts[3 * j]; sy[threadIdx.x] = pts[3 * j + 1]; sz[threadIdx.x] = pts[3 * j + 2]; }
        __syncthreads();
        const int n = min(KNN_TILE, P - base);
#pragma unroll 4
        for (int t = slice; t < n; t += KNN_S) {
            if (base + t == q) continue;
            keep3(gps::knn_dist2(sx[t] - qx, sy[t] - qy, sz[t] - qz), b0, b1, b2);
        }
    }
    best[slice][ql][0] = b0; best[slice][ql][1] = b1; best[slice][ql][2] = b2;
    __syncthreads();
    if (slice == 0 && live) {
        for (int s2 = 1; s2 < KNN_S; s2++) {
            keep3(best[s2][ql][0], b0, b1, b2); keep3(best[s2][ql][1], b0, b1, b2); keep3(best[s2][ql][2], b0, b1, b2);
        }
        out[q] = (b0 + b1 + b2) / 3.0f;
    }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void normal_map_kernel(int W, int H, const float* __restrict__ v, float* __restrict__ n) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    float gx[3], gy[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float p[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) p[i][j] = v[3 * (clampi(y + i - 1, 0, H - 1) * W + clampi(x + j - 1, 0, W - 1)) + c];
        gx[c] = (p[0][2] - p[0][0]) + 2.f * (p[1][2] - p[1][0]) + (p[2][2] - p[2][0]);
        gy[c] = (p[2][0] - p[0][0]) + 2.f * (p[2][1] - p[0][1]) + (p[2][2] - p[0][2]);
    }
    // cross(img_dy, img_dx)
    float nx = gy[1] * gx[2] - gy[2] * gx[1], ny = gy[2] * gx[0] - gy[0] * gx[2], nz = gy[0] * gx[1] - gy[1] * gx[0];
    const float mag = sqrtf(nx * nx + ny * ny + nz * nz) + 1e-8f;
    const int o = 3 * (y * W + x);
    const bool invalid = v[o + 2] <= 0.f;  // the reference tests the WORLD z of the vertex (tensor_math.cpp:294-295)
    n[o] = invalid ? 0.f : nx / mag; n[o + 1] = invalid ? 0.f : ny / mag; n[o + 2] = invalid ? 0.f : nz / mag;
}

}  // namespace

namespace {
struct SmallFloats { float v[64]; };
// ---------------------------------------------------------------- new-Gaussian sampling (initNewGaussians / addGaussians)
// slam_pipeline.cpp:450-526 builds the sample mask with ~12 elementwise / reduce tensor ops; one thread per pixel here, with
// ATen's float sequence on this stack (tools/probe/aten_mean_probe.py: a 3-element reduction runs two accumulators, even and
// odd elements): sum = (x + z) + y, mean = ((a + c) + b) * RN(1/3).
__global__ __launch_bounds__(256) void new_gaussian_mask_kernel(int P, const float* __restrict__ depth, const float* __restrict__ src,
                                                               const float* __restrict__ image, const float* __restrict__ vertex,
                                                               const float* __restrict__ alpha, float dmin, float dmax,
                                                               float err_thres, float alpha_max, uint8_t* __restrict__ mask) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float d = depth[p];
    bool valid = (d > dmin) && (d < dmax);
    const float vs = (vertex[3 * p] + vertex[3 * p + 2]) + vertex[3 * p + 1];
    valid = valid && !(vs == 0.0f);
    const float e0 = fabsf(src[3 * p] - image[3 * p]), e1 = fabsf(src[3 * p + 1] - image[3 * p + 1]),
                e2 = fabsf(src[3 * p + 2] - image[3 * p + 2]);
    const float err = ((e0 + e2) + e1) * (1.0f / 3.0f);
    bool m = (err > err_thres) && valid;
    if (alpha) m = m && (alpha[p] < alpha_max);
    mask[p] = m ? 1 : 0;
}

// masked_select's ORDER (row-major) without its host round trips, in two launches over 4096-byte blocks (16 consecutive
// bytes per thread, one coalesced dwordx4 each): (1) set bytes per block; (2) every block sums the counts in front of it (a
// few hundred adds by one wave), scans its own threads and writes the ids; block 0 also publishes the total to a device
// word and (optionally) a pinned host word.
constexpr int CM_THREADS = 256, CM_PER = 16, CM_BLOCK = CM_THREADS * CM_PER;

__device__ __forceinline__ int load_mask16(int n, const uint8_t* __restrict__ mask, int first, uint32_t (&w)[4]) {
    // bytes [first, first + 16) as four words (zero past n); returns the number of non-zero bytes
    if (first + CM_PER <= n && (reinterpret_cast<uintptr_t>(mask + first) & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4*>(mask + first);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            w[q] = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int p = first + 4 * q + k;
                if (p < n) w[q] |= (uint32_t)mask[p] << (8 * k);
            }
        }
    }
    int c = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int k = 0; k < 4; k++) c += ((w[q] >> (8 * k)) & 0xFFu) ? 1 : 0;
    return c;
}

__global__ __launch_bounds__(CM_THREADS) void mask_count_kernel(int n, const uint8_t* __restrict__ mask, int32_t* __restrict__ blk) {
    __shared__ int red[CM_THREADS / 64];
    uint32_t w[4];
    int c = load_mask16(n, mask, blockIdx.x * CM_BLOCK + threadIdx.x * CM_PER, w);
    c = wave_sum_i(c);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(CM_THREADS) void mask_write_kernel(int n, const uint8_t* __restrict__ mask, const int32_t* __restrict__ blk,
                                                               int nblk, int32_t* __restrict__ ids, int32_t* __restrict__ count,
                                                               volatile int32_t* host_count) {
    __shared__ int ws[CM_THREADS / 64 + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {  // ids in front of this block, and (block 0) the grand total
        int before = 0, total = 0;
        for (int b = lane; b < nblk; b += 64) { const int v = blk[b]; total += v; if (b < (int)blockIdx.x) before += v; }
        before = wave_sum_i(before); total = wave_sum_i(total);
        if (lane == 0) {
            ws[CM_THREADS / 64] = before;
            if (blockIdx.x == 0) { count[0] = total; if (host_count) host_count[0] = total; }
        }
    }
    uint32_t w[4];
    const int first = blockIdx.x * CM_BLOCK + threadIdx.x * CM_PER;
    const int c = load_mask16(n, mask, first, w);
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    int run = ws[CM_THREADS / 64] + incl - c;
    for (int k = 0; k < wave; k++) run += ws[k];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((w[q] >> (8 * k)) & 0xFFu) ids[run++] = first + 4 * q + k;
}

// rows subset[j] of the masked pixels: vertex, image colour, normal -> three [k,3] arrays
__global__ __launch_bounds__(256) void gather_pixels_kernel(int k, const int32_t* __restrict__ ids, const int32_t* __restrict__ subset,
                                                           const float* __restrict__ vertex, const float* __restrict__ image,
                                                           const float* __restrict__ normal, float* __restrict__ verts,
                                                           float* __restrict__ cols, float* __restrict__ norms) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    const int p = ids[subset[j]];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        verts[3 * j + c] = vertex[3 * p + c];
        cols[3 * j + c] = image[3 * p + c];
        norms[3 * j + c] = normal[3 * p + c];
    }
}

// RawGaussianParams::init (src/raw_gs_param.cpp:11-74) for k new Gaussians in one launch: KNN scale (clamped, z x 0.1 when a
// normal is given), normal -> quaternion (computeQuat / quaternionFromAxisAngle, src/tensor_math.cpp:184-201), colour -> SH
// DC, zero higher bands, logit(opacity).  Written straight into the rows the caller points at (the tail of the model's
// capacity buffers), in the reference's float sequence (3-element reductions as ATen runs them: (x0 + x2) + x1).
__global__ __launch_bounds__(256) void init_gaussians_kernel(int k, const float* __restrict__ xyz, const float* __restrict__ rgb,
                                                            const float* __restrict__ normals, const float* __restrict__ knn,
                                                            int K, float init_opac, float max_scale, float min_scale,
                                                            float* __restrict__ means, float* __restrict__ log_scales,
                                                            float* __restrict__ quats, float* __restrict__ sh_dc,
                                                            float* __restrict__ sh_rest, float* __restrict__ opac) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    means[3 * i] = xyz[3 * i]; means[3 * i + 1] = xyz[3 * i + 1]; means[3 * i + 2] = xyz[3 * i + 2];
    const float raw = fminf(fmaxf(sqrtf(knn[i]), min_scale), max_scale);  // torch::clamp(sqrt(distCUDA2), min, max)
    float q0 = 1.0f, q1 = 1.0f, q2 = 1.0f, q3 = 1.0f;                       // torch::ones without normals
    float sz = raw;
    if (normals) {
        sz = raw * 0.1f;
        const float nx = normals[3 * i], ny = normals[3 * i + 1], nz = normals[3 * i + 2];
        // axis = cross((0,0,1), n); axis / (|axis| + 1e-8), twice; angle = acos(dot((0,0,1), n))
        float ax = 0.0f * nz - 1.0f * ny, ay = 1.0f * nx - 0.0f * nz, az = 0.0f * ny - 0.0f * nx;
        float nrm = sqrtf((ax * ax + az * az) + ay * ay) + 1e-8f;
        ax = ax / nrm; ay = ay / nrm; az = az / nrm;
        const float angle = acosf((0.0f * nx + 1.0f * nz) + 0.0f * ny);
        nrm = sqrtf((ax * ax + az * az) + ay * ay) + 1e-8f;
        ax = ax / nrm; ay = ay / nrm; az = az / nrm;
        const float half = angle * 0.5f;
        const float sn = sinf(half);
        q0 = cosf(half); q1 = ax * sn; q2 = ay * sn; q3 = az * sn;
    }
    log_scales[3 * i] = logf(raw); log_scales[3 * i + 1] = logf(raw); log_scales[3 * i + 2] = logf(sz);
    *reinterpret_cast<float4*>(quats + 4 * (size_t)i) = make_float4(q0, q1, q2, q3);
    // (rgb - 0.5) / SH_C0 with a host scalar: ATen multiplies by the reciprocal, taken in DOUBLE and then rounded
    // (tools/probe/aten_div_probe.py: f32(1 / C0) = 3.5449078, not 1 / f32(C0) = 3.5449076)
    const float inv_c0 = (float)(1.0 / 0.28209479177387814);
    sh_dc[3 * i] = (rgb[3 * i] - 0.5f) * inv_c0; sh_dc[3 * i + 1] = (rgb[3 * i + 1] - 0.5f) * inv_c0;
    sh_dc[3 * i + 2] = (rgb[3 * i + 2] - 0.5f) * inv_c0;
    float* r = sh_rest + (size_t)i * (K - 1) * 3;
    for (int c = 0; c < (K - 1) * 3; c++) r[c] = 0.0f;
    opac[i] = logf(init_opac / (1.0f - init_opac));  // torch::logit
}

// removeRedundantGs' delete mask (slam/slam_pipeline.cpp:564-586): max real scale < small or > large, or real opacity < low;
// keep = !delete.  exp / sigmoid as the tensor ops evaluate them (expf; 1 / (1 + expf(-x))).
__global__ __launch_bounds__(256) void prune_mask_kernel(int N, const float* __restrict__ log_scales, const float* __restrict__ opac_logit,
                                                        float small_thres, float large_thres, float low_opac,
                                                        uint8_t* __restrict__ del, uint8_t* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float s0 = expf(log_scales[3 * i]), s1 = expf(log_scales[3 * i + 1]), s2 = expf(log_scales[3 * i + 2]);
    const float smax = fmaxf(fmaxf(s0, s1), s2);
    const float op = 1.0f / (1.0f + expf(-opac_logit[i]));
    const bool d = (smax < small_thres) || (smax > large_thres) || (op < low_opac);
    del[i] = d ? 1 : 0;
    keep[i] = d ? 0 : 1;
}

// index_select of rows ids[0..m) for up to 8 row-major float tensors in one launch (prunePoints: the six parameter tensors)
struct GatherArgs { const float* src[8]; float* dst[8]; int row[8]; int64_t end[8]; int n; };
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherArgs a, const int32_t* __restrict__ ids) {
    const int64_t total = a.end[a.n - 1];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) s += (k < a.n - 1 && e >= a.end[k]) ? 1 : 0;
        const float* src = a.src[0]; float* dst = a.dst[0]; int row = a.row[0]; int64_t first = 0;
#pragma unroll
        for (int k = 1; k < 8; k++) if (s == k) { src = a.src[k]; dst = a.dst[k]; row = a.row[k]; first = a.end[k - 1]; }
        const int64_t le = e - first;
        const int r = (int)(le / row), c = (int)(le - (int64_t)r * row);
        dst[le] = src[(int64_t)ids[r] * row + c];
    }
}

// uchar4 frame -> float3 image in [0, 1]: tensor.slice(2, 0, 3).to(float).div_(255) (x * RN(1/255)) in one launch
// (+ optionally up to 64 host floats through the kernel-argument buffer, as upload_floats_kernel: Camera::toGPU's pose / intrinsics
// pack in the same launch as the frame's image)
__global__ __launch_bounds__(256) void rgba8_to_rgbf_kernel(int P, const uchar4* __restrict__ src, float* __restrict__ dst,
                                                           SmallFloats v, int n_floats, float* __restrict__ floats_dst) {
    if (blockIdx.x == 0 && (int)threadIdx.x < n_floats) floats_dst[threadIdx.x] = v.v[threadIdx.x];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const uchar4 c = src[p];
    const float k = (float)(1.0 / 255.0);
    dst[3 * p] = (float)c.x * k; dst[3 * p + 1] = (float)c.y * k; dst[3 * p + 2] = (float)c.z * k;
}

__global__ __launch_bounds__(64) void upload_floats_kernel(SmallFloats v, int n, float* __restrict__ dst) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = v.v[threadIdx.x];
}
}  // namespace

extern "C" {

int gps_knn_mean_dist2(int P, const float* points, float* mean_dist2, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(P >= 0);
    if (P == 0) return GPS_OK;
    GPS_REQUIRE(points && mean_dist2);
    knn_kernel<<<gps_div_up(P, KNN_Q), KNN_TILE, 0, (hipStream_t)stream>>>(P, points, mean_dist2);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_normal_map(int width, int height, const float* vertex_map, float* normal_map, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(width > 0 && height > 0 && vertex_map && normal_map);
    dim3 grid(gps_div_up(width, 16), gps_div_up(height, 16));
    normal_map_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(width, height, vertex_map, normal_map);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_new_gaussian_mask(int width, int height, const float* depth_map, const float* src_rgb, const float* image,
                          const float* vertex_map, const float* alpha, float depth_vis_min, float depth_vis_max,
                          float color_error_thres, float alpha_vis_max, uint8_t* mask, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(width > 0 && height > 0 && depth_map && src_rgb && image && vertex_map && mask);
    const int P = width * height;
    new_gaussian_mask_kernel<<<gps_div_up(P, 256), 256, 0, (hipStream_t)stream>>>(P, depth_map, src_rgb, image, vertex_map, alpha,
                                                                                 depth_vis_min, depth_vis_max, color_error_thres,
                                                                                 alpha_vis_max, mask);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int64_t gps_compact_mask_workspace_bytes(int n) { return n < 0 ? GPS_ERR_ARG : (int64_t)sizeof(int32_t) * (gps_div_up(n, CM_BLOCK) + 1); }

int gps_compact_mask(int n, const uint8_t* mask, int32_t* ids, int32_t* count, int32_t* host_count, void* workspace,
                     int64_t workspace_bytes, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(n >= 0 && count && workspace && (n == 0 || (mask && ids)));
    if (workspace_bytes < gps_compact_mask_workspace_bytes(n)) return GPS_ERR_CAPACITY;
    const int nblk = gps_div_up(n, CM_BLOCK);
    int32_t* blk = (int32_t*)workspace;
    hipStream_t s = (hipStream_t)stream;
    if (nblk > 0) mask_count_kernel<<<nblk, CM_THREADS, 0, s>>>(n, mask, blk);
    mask_write_kernel<<<nblk > 0 ? nblk : 1, CM_THREADS, 0, s>>>(n, mask, blk, nblk, ids, count, host_count);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_gather_pixels(int k, const int32_t* ids, const int32_t* subset, const float* vertex_map, const float* image,
                      const float* normal_map, float* verts, float* cols, float* norms, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(k >= 0);
    if (k == 0) return GPS_OK;
    GPS_REQUIRE(ids && subset && vertex_map && image && normal_map && verts && cols && norms);
    gather_pixels_kernel<<<gps_div_up(k, 256), 256, 0, (hipStream_t)stream>>>(k, ids, subset, vertex_map, image, normal_map, verts,
                                                                             cols, norms);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_init_gaussians(int k, const float* xyz, const float* rgb, const float* normals, const float* knn_mean_dist2, int K,
                       float init_opacity, float max_scale, float min_scale, float* means, float* log_scales, float* quats,
                       float* sh_dc, float* sh_rest, float* opac_logit, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(k >= 0 && K >= 1);
    if (k == 0) return GPS_OK;
    GPS_REQUIRE(xyz && rgb && knn_mean_dist2 && means && log_scales && quats && sh_dc && opac_logit && (K == 1 || sh_rest));
    GPS_REQUIRE((((uintptr_t)quats) & 15) == 0);
    init_gaussians_kernel<<<gps_div_up(k, 256), 256, 0, (hipStream_t)stream>>>(k, xyz, rgb, normals, knn_mean_dist2, K, init_opacity,
                                                                              max_scale, min_scale, means, log_scales, quats,
                                                                              sh_dc, sh_rest, opac_logit);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_prune_mask(int N, const float* log_scales, const float* opac_logit, float small_scale_thres, float large_scale_thres,
                   float low_opac_thres, uint8_t* delete_mask, uint8_t* keep_mask, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(N >= 0);
    if (N == 0) return GPS_OK;
    GPS_REQUIRE(log_scales && opac_logit && delete_mask && keep_mask);
    prune_mask_kernel<<<gps_div_up(N, 256), 256, 0, (hipStream_t)stream>>>(N, log_scales, opac_logit, small_scale_thres,
                                                                          large_scale_thres, low_opac_thres, delete_mask, keep_mask);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_gather_rows(int m, const int32_t* ids, int n_tensors, const float* const* srcs, float* const* dsts, const int32_t* row_floats,
                    gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(m >= 0 && n_tensors >= 1 && n_tensors <= 8 && srcs && dsts && row_floats);
    if (m == 0) return GPS_OK;
    GPS_REQUIRE(ids != nullptr);
    GatherArgs a = {};
    int64_t run = 0;
    for (int k = 0; k < n_tensors; k++) {
        GPS_REQUIRE(srcs[k] && dsts[k] && row_floats[k] >= 1);
        a.src[k] = srcs[k]; a.dst[k] = dsts[k]; a.row[k] = row_floats[k];
        run += (int64_t)m * row_floats[k];
        a.end[k] = run;
    }
    for (int k = n_tensors; k < 8; k++) { a.src[k] = srcs[0]; a.dst[k] = dsts[0]; a.row[k] = 1; a.end[k] = run; }
    a.n = n_tensors;
    gather_rows_kernel<<<(int)min((int64_t)8192, (int64_t)gps_div_up(run, 256)), 256, 0, (hipStream_t)stream>>>(a, ids);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_rgba8_to_rgbf(int n_pixels, const uint8_t* rgba, float* rgb, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(n_pixels >= 0);
    if (n_pixels == 0) return GPS_OK;
    GPS_REQUIRE(rgba && rgb && (((uintptr_t)rgba) & 3) == 0);
    SmallFloats none = {};
    rgba8_to_rgbf_kernel<<<gps_div_up(n_pixels, 256), 256, 0, (hipStream_t)stream>>>(n_pixels, reinterpret_cast<const uchar4*>(rgba), rgb,
                                                                                     none, 0, nullptr);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_rgba8_to_rgbf_and_floats(int n_pixels, const uint8_t* rgba, float* rgb, float* floats_dst, const float* host_values,
                                 int n_floats, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(n_pixels > 0 && rgba && rgb && (((uintptr_t)rgba) & 3) == 0);
    GPS_REQUIRE(floats_dst && host_values && n_floats > 0 && n_floats <= 64);
    SmallFloats v;
    for (int k = 0; k < 64; k++) v.v[k] = k < n_floats ? host_values[k] : 0.0f;
    rgba8_to_rgbf_kernel<<<gps_div_up(n_pixels, 256), 256, 0, (hipStream_t)stream>>>(n_pixels, reinterpret_cast<const uchar4*>(rgba), rgb,
                                                                                     v, n_floats, floats_dst);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_upload_floats(float* dst, const float* host_values, int n, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(dst && host_values && n > 0 && n <= 64);
    SmallFloats v;
    for (int k = 0; k < 64; k++) v.v[k] = k < n ? host_values[k] : 0.0f;
    upload_floats_kernel<<<1, 64, 0, (hipStream_t)stream>>>(v, n, dst);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

}  // extern "C"

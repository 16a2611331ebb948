// Gaussian-creation helpers on the every-10-frames path (SLAMGaussianModel::addGaussians ->
// RawGaussianParams::init, src/raw_gs_param.cpp:11-74; SLAMPipeline::initNewGaussians, slam_pipeline.cpp:503):
//
// gps_knn_mean_dist2 <- distCUDA2 (gsplat/rasterizer/simple_knn.cu:191-240): mean squared distance to the 3 nearest
//                       neighbours.  The reference builds a Morton order + 1024-point boxes with cub/thrust,
//                       cudaMalloc/cudaFree and two blocking memcpys per call; P is 1e2..1e4 here, so an exact
//                       LDS-tiled brute force (P^2 / 2^24 lane-steps) is both simpler and faster and needs no
//                       scratch memory.  Same result: the 3 smallest squared distances are a set property.
// gps_normal_map     <- computeNormalMap (src/tensor_math.cpp:217-248, 278-300): Sobel gradients of the vertex map
//                       with replicate padding, cross(dy, dx), normalise, zero where vertex z <= 0.
#include <float.h>

#include "common.hpp"

namespace {

constexpr int KNN_TILE = 256;

__device__ __forceinline__ void keep3(float d, float& b0, float& b1, float& b2) {  // simple_knn.cu:137-150
    if (b0 > d) { float t = b0; b0 = d; d = t; }
    if (b1 > d) { float t = b1; b1 = d; d = t; }
    if (b2 > d) { b2 = d; }
}

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
            const float dx = sx[t] - qx, dy = sy[t] - qy, dz = sz[t] - qz;
            keep3(dx * dx + dy * dy + dz * dz, b0, b1, b2);
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

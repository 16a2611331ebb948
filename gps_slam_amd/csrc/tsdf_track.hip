// Depth-only ExtendedTracker: the tracker ITMLibSettings.cpp:54-57 configures and ITMBasicEngine
// runs when use_gt_pose is false.
//
//   ITMLib/Trackers/Interface/ITMExtendedTracker.cpp:143-177 (SetupLevels), :216-268 (PrepareForEvaluation: depth pyramid),
//   :293-375 (ComputeDelta / HasConverged / ApplyDelta), :470-665 (TrackCamera);
//   Trackers/Shared/ITMExtendedTracker_Shared.h:51-143, 298-328 (per point); Trackers/CUDA/ITMExtendedTracker_CUDA.cu
//   (per-iteration reduction + 32-float read-back); Engines/LowLevel/Shared/ITMLowLevelEngine_Shared.h:48-69.
//
// Split exactly like the reference's GPU tracker: the per-pixel residual / Jacobian evaluation and its reduction are
// kernels, the 6x6 Levenberg-Marquardt bookkeeping (Cholesky, step, SE3 coercion, accept / reject) is host code fed by 32
// floats per iteration (written by the kernel straight into a pinned host mailbox the host spins on).  The reduction is a fixed two-stage tree (per-thread register sums -> wave -> workgroup
// -> 32 row-group sums over the <= 128 workgroup partials, added in fixed order), so results are reproducible run to run; they differ from the CPU engine's
// scan-order sums only by float re-association (poses agree to ~1e-6, tests/test_tsdf_gpu.py).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "tsdf_common.hpp"

using namespace gpst;

namespace {

constexpr int TRK_ROTATION = 0, TRK_TRANSLATION = 1, TRK_BOTH = 2, TRK_NONE = 3;
constexpr int GH_MAX_WGS = 2048;
constexpr int GH_WGS = 512, GH_SLOTS = 32;  // partial[GH_WGS][GH_SLOTS]: 0 = count, 1 = f, 2.. = nabla, then lower-tri hessian

__global__ __launch_bounds__(256) void subsample_with_holes_kernel(const float* __restrict__ in, int w_in, int w, int h,
                                                                   float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    float acc = 0.0f, good = 0.0f, v;
    v = in[(2 * x + 0) + (2 * y + 0) * w_in]; if (v > 0.0f) { acc += v; good++; }
    v = in[(2 * x + 1) + (2 * y + 0) * w_in]; if (v > 0.0f) { acc += v; good++; }
    v = in[(2 * x + 0) + (2 * y + 1) * w_in]; if (v > 0.0f) { acc += v; good++; }
    v = in[(2 * x + 1) + (2 * y + 1) * w_in]; if (v > 0.0f) { acc += v; good++; }
    if (good > 0) acc /= good;
    out[i] = acc;
}

__global__ __launch_bounds__(256) void count_valid_kernel(const float* __restrict__ depth, int n, int* __restrict__ out) {
    __shared__ int red[4];
    int c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c += depth[i] > 0.0f ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// interpolateBilinear_withHoles (Utils/ITMPixelUtils.h:78-106)
__device__ __forceinline__ float4 bilinear_with_holes(const float4* __restrict__ src, float px, float py, int W) {
    const short ix = (short)floorf(px), iy = (short)floorf(py);
    const float dx = px - (float)ix, dy = py - (float)iy;
    const float4 a = src[ix + iy * W], b = src[(ix + 1) + iy * W], c = src[ix + (iy + 1) * W], d = src[(ix + 1) + (iy + 1) * W];
    if (a.w < 0 || b.w < 0 || c.w < 0 || d.w < 0) return make_float4(0.f, 0.f, 0.f, -1.0f);
    float4 r;
    r.x = (a.x * (1.0f - dx) * (1.0f - dy) + b.x * dx * (1.0f - dy) + c.x * (1.0f - dx) * dy + d.x * dx * dy);
    r.y = (a.y * (1.0f - dx) * (1.0f - dy) + b.y * dx * (1.0f - dy) + c.y * (1.0f - dx) * dy + d.y * dx * dy);
    r.z = (a.z * (1.0f - dx) * (1.0f - dy) + b.z * dx * (1.0f - dy) + c.z * (1.0f - dx) * dy + d.z * dx * dy);
    r.w = (a.w * (1.0f - dx) * (1.0f - dy) + b.w * dx * (1.0f - dy) + c.w * (1.0f - dx) * dy + d.w * dx * dy);
    return r;
}

// The scene side as the tracker reads it: point and normal of a pixel in ONE 32-byte record (pn[2i], pn[2i+1]).  The two maps
// are separate arrays in gps_tsdf_state (ICP maps, API-visible); gathering a bilinear footprint from both touches 4 + 4 cache
// lines in two dependent phases (normals only after the distance test).  Interleaved, the footprint is 2 x 64 contiguous bytes
// fetched in one phase -- half the lines to keep in (or re-fetch into) an L2 the map stream's rasterizer kernels keep sweeping.
__global__ __launch_bounds__(256) void interleave_maps_kernel(const float4* __restrict__ points, const float4* __restrict__ normals,
                                                             int n, float4* __restrict__ pn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pn[2 * i] = points[i];
    pn[2 * i + 1] = normals[i];
}

__device__ __forceinline__ float4 bilerp4(const float4 a, const float4 b, const float4 c, const float4 d, float dx, float dy) {
    if (a.w < 0 || b.w < 0 || c.w < 0 || d.w < 0) return make_float4(0.f, 0.f, 0.f, -1.0f);
    float4 r;
    r.x = (a.x * (1.0f - dx) * (1.0f - dy) + b.x * dx * (1.0f - dy) + c.x * (1.0f - dx) * dy + d.x * dx * dy);
    r.y = (a.y * (1.0f - dx) * (1.0f - dy) + b.y * dx * (1.0f - dy) + c.y * (1.0f - dx) * dy + d.y * dx * dy);
    r.z = (a.z * (1.0f - dx) * (1.0f - dy) + b.z * dx * (1.0f - dy) + c.z * (1.0f - dx) * dy + d.z * dx * dy);
    r.w = (a.w * (1.0f - dx) * (1.0f - dy) + b.w * dx * (1.0f - dy) + c.w * (1.0f - dx) * dy + d.w * dx * dy);
    return r;
}

// interpolateBilinear_withHoles of the point map and of the normal map at the same position (each with its own hole rule)
__device__ __forceinline__ void bilinear_pair_with_holes(const float4* __restrict__ pn, float px, float py, int W, float4& cp, float4& n) {
    const short ix = (short)floorf(px), iy = (short)floorf(py);
    const float dx = px - (float)ix, dy = py - (float)iy;
    const float4* r0 = pn + 2 * (ix + iy * W);
    const float4* r1 = pn + 2 * (ix + (iy + 1) * W);
    const float4 pa = r0[0], na = r0[1], pb = r0[2], nb = r0[3], pc = r1[0], nc = r1[1], pd = r1[2], nd = r1[3];
    cp = bilerp4(pa, pb, pc, pd, dx, dy);
    n = bilerp4(na, nb, nc, nd, dx, dy);
}

struct GhArgs {
    const float* depth;
    int vw, vh;
    float4 view_intr;
    const float4* pn;  // interleaved point | normal records (interleave_maps_kernel)
    int sw, sh;
    float4 scene_intr;
    Mat4 approxInvPose, scenePose;
    float space_thresh, tukey_cutoff, vf_min, vf_max;
    int use_weights, frames_to_skip, frames_to_weight;
};

// computePerPointGH_exDepth for one pixel, accumulated into the caller's registers.  ITER: 0 rotation, 1 translation, 2 both.
template <int ITER>
__device__ __forceinline__ void gh_point(const GhArgs& a, int x, int y, float& cnt, float& f, float* nabla, float* hess) {
    constexpr int NP = ITER == TRK_BOTH ? 6 : 3;
    const float depth = a.depth[x + y * a.vw];
    if (depth <= 1e-8f) return;
    float px = depth * (((float)x - a.view_intr.z) / a.view_intr.x);
    float py = depth * (((float)y - a.view_intr.w) / a.view_intr.y);
    float tx, ty, tz;
    mul_point(a.approxInvPose, px, py, depth, 1.0f, tx, ty, tz);
    float qx, qy, qz;
    mul_point(a.scenePose, tx, ty, tz, 1.0f, qx, qy, qz);
    if (qz <= 0.0f) return;
    const float u = a.scene_intr.x * qx / qz + a.scene_intr.z;
    const float v = a.scene_intr.y * qy / qz + a.scene_intr.w;
    if (!((u >= 0.0f) && (u <= a.sw - 2) && (v >= 0.0f) && (v <= a.sh - 2))) return;
    float4 cp, n;
    bilinear_pair_with_holes(a.pn, u, v, a.sw, cp, n);
    if (cp.w < 0.0f) return;
    const float dx = cp.x - tx, dy = cp.y - ty, dz = cp.z - tz;
    const float dist = dx * dx + dy * dy + dz * dz;
    if (dist > a.tukey_cutoff * a.space_thresh) return;
    float w = fmaxf(0.0f, 1.0f - (depth - a.vf_min) / (a.vf_max - a.vf_min));
    w *= w;
    if (a.use_weights) {
        if (cp.w < a.frames_to_skip) return;
        w *= (cp.w - a.frames_to_skip) / a.frames_to_weight;
    }
    const float b = n.x * dx + n.y * dy + n.z * dz;
    float A[6];
    if (ITER == TRK_TRANSLATION) {
        A[0] = n.x; A[1] = n.y; A[2] = n.z;
    } else {
        A[0] = +tz * n.y - ty * n.z;
        A[1] = -tz * n.x + tx * n.z;
        A[2] = +ty * n.x - tx * n.y;
        if (ITER == TRK_BOTH) { A[3] = n.x; A[4] = n.y; A[5] = n.z; }
    }
    const float h = a.space_thresh;
    // rho / rho' / rho'' (Shared.h:51-69): Huber-like norm
    float t = fabsf(b) - h; t = fmaxf(t, 0.0f);
    const float rho = b * b - t * t;
    const float rho1 = 2.0f * fminf(fmaxf(b, -h), h);
    const float rho2 = fabsf(b) < h ? 2.0f : 0.0f;
    cnt += 1.0f;
    f += rho * w;
#pragma unroll
    for (int r = 0, counter = 0; r < NP; r++) {
        nabla[r] += rho1 * w * A[r];
#pragma unroll
        for (int c = 0; c <= r; c++, counter++) hess[counter] += rho2 * w * A[r] * A[c];
    }
}

// 256 threads: thread (q, r) adds the q-th float4 of rows r, r + 32, r + 64, ... (independent 16-byte loads, issued eight
// at a time), then the 32 row-group sums of every slot are added in fixed order; totals -> result (+ host mailbox)
__device__ __forceinline__ void final_sum(const float* __restrict__ partial, int rows, float* __restrict__ result,
                                          volatile float* mailbox) {
    __shared__ float4 group[32][8];
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int q = threadIdx.x & 7, r = threadIdx.x >> 3;
    const f4v* p4 = reinterpret_cast<const f4v*>(partial);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = r; b < rows; b += 32 * 8) {
        f4v v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int row = b + 32 * k;
            const f4v zero = {0.f, 0.f, 0.f, 0.f};
            v[k] = row < rows ? __builtin_nontemporal_load(&p4[row * 8 + q]) : zero;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
    }
    group[r][q] = s;
    __syncthreads();
    if (threadIdx.x < GH_SLOTS) {
        const float* g = reinterpret_cast<const float*>(group);
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 32; k++) t += g[k * GH_SLOTS + threadIdx.x];
        result[threadIdx.x] = t;
        if (mailbox) mailbox[threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void track_sum_kernel(const float* __restrict__ partial, int rows, float* __restrict__ result,
                                                       volatile float* mailbox, int seq, const int* __restrict__ valid_count) {
    final_sum(partial, rows, result, mailbox);
    if (mailbox) {
        // the frame's valid-pixel count (count_valid_kernel finished long ago on this stream) rides along, so that the pose
        // quality score at the end of the frame needs no read-back of its own
        if (threadIdx.x == 0) mailbox[GH_SLOTS + 1] = __int_as_float(*valid_count);
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) { mailbox[GH_SLOTS] = __int_as_float(seq); __threadfence_system(); }
    }
}

// Per LM iteration: this kernel leaves one row of partial sums per workgroup, track_sum_kernel adds the rows in a fixed
// order (a last-block-done ticket inside this kernel was measured slower than the second launch: one device-scope atomic
// per workgroup on a single address plus the fences cost more than ~5 us of launch).
template <int ITER>
__global__ __launch_bounds__(256) void track_gh_kernel(GhArgs a, float* __restrict__ partial) {
    constexpr int NP = ITER == TRK_BOTH ? 6 : 3, NSQ = ITER == TRK_BOTH ? 21 : 6, NV = 2 + NP + NSQ;
    __shared__ float red[4][GH_SLOTS];
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = 0.0f;
    const int n = a.vw * a.vh;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = i / a.vw, x = i - y * a.vw;
        gh_point<ITER>(a, x, y, acc[0], acc[1], acc + 2, acc + 2 + NP);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        float v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < GH_SLOTS)
        partial[blockIdx.x * GH_SLOTS + threadIdx.x] =
            threadIdx.x < NV ? ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) : 0.0f;
}

// ---------------------------------------------------------------- host side: ORUtils::Cholesky, TrackCamera bookkeeping
struct Chol {
    float ch[36];
    int size;
    Chol(const float* mat, int n) : size(n) {
        for (int i = 0; i < n * n; i++) ch[i] = mat[i];
        for (int c = 0; c < n; c++) {
            float inv_diag = 1;
            for (int r = c; r < n; r++) {
                float val = ch[c + r * n];
                for (int c2 = 0; c2 < c; c2++) val -= ch[c + c2 * n] * ch[c2 + r * n];
                if (r == c) { ch[c + r * n] = val; inv_diag = 1.0f / val; }
                else { ch[r + c * n] = val; ch[c + r * n] = val * inv_diag; }
            }
        }
    }
    void backsub(float* result, const float* v) const {
        float y[6];
        for (int i = 0; i < size; i++) {
            float val = v[i];
            for (int j = 0; j < i; j++) val -= ch[j + i * size] * y[j];
            y[i] = val;
        }
        for (int i = 0; i < size; i++) y[i] /= ch[i + i * size];
        for (int i = size - 1; i >= 0; i--) {
            float val = y[i];
            for (int j = i + 1; j < size; j++) val -= ch[i + j * size] * result[j];
            result[i] = val;
        }
    }
    float determinant() const {
        float ret = 1.0f;
        for (int i = 0; i < size; ++i) ret *= ch[i + i * size];
        return ret * ret;
    }
};

void m4_mul(const float* a, const float* b, float* out) {  // ORUtils Matrix4 operator*
    float r[16];
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++) {
            float acc = 0;
            for (int k = 0; k < 4; k++) acc += a[k * 4 + row] * b[col * 4 + k];
            r[col * 4 + row] = acc;
        }
    memcpy(out, r, sizeof(r));
}

bool set_invM_coerce(const float* invM_in, float* M, float* invM) {  // pose_d->SetInvM(m); Coerce()
    float rm[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) rm[r * 4 + c] = invM_in[c * 4 + r];
    return gps_pose_from_c2w(rm, M, invM) == GPS_OK;
}

inline int float_bits(float f) { int i; memcpy(&i, &f, 4); return i; }

struct Scratch {
    float* level[GPS_TRACK_MAX_LEVELS];  // [0] unused (= s.depth)
    float *partial, *result;
    int* count;
    float4* pn;
};

size_t carve(Scratch* w, char* base, int W, int H) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return base ? base + o : nullptr; };
    int lw = W, lh = H;
    for (int l = 1; l < GPS_TRACK_MAX_LEVELS; l++) {
        lw /= 2; lh /= 2;
        char* p = take((size_t)(lw > 0 && lh > 0 ? lw * lh : 1) * sizeof(float));
        if (w) w->level[l] = (float*)p;
    }
    char* p = take((size_t)GH_MAX_WGS * GH_SLOTS * sizeof(float)); if (w) w->partial = (float*)p;
    p = take(GH_SLOTS * sizeof(float)); if (w) w->result = (float*)p;
    p = take(sizeof(int)); if (w) w->count = (int*)p;
    p = take((size_t)W * H * 2 * sizeof(float4)); if (w) w->pn = (float4*)p;
    return off;
}

}  // namespace

extern "C" {

int gps_track_config_init(gps_track_config* c, const char* levels, int num_iter_coarse, int num_iter_fine, float thresh_coarse,
                          float thresh_fine, float term_thresh, float tukey_cutoff, int frames_to_skip, int frames_to_weight) {
    if (!c || !levels) return GPS_ERR_ARG;
    const int n = (int)strlen(levels);
    if (n < 2 || n > GPS_TRACK_MAX_LEVELS) return GPS_ERR_ARG;
    memset(c, 0, sizeof(*c));
    c->n_levels = n;
    for (int i = n - 1, k = 0; i >= 0; --i, ++k)  // parsed from the END (ITMTrackerFactory.h:145-167): level 0 = last character
        c->iter_type[k] = levels[i] == 'r' ? TRK_ROTATION : levels[i] == 't' ? TRK_TRANSLATION : levels[i] == 'b' ? TRK_BOTH : TRK_NONE;
    {
        float step = (float)(num_iter_coarse - num_iter_fine) / (float)(n - 1);
        float val = (float)num_iter_coarse;
        for (int l = n - 1; l >= 0; l--) { c->n_iter[l] = (int)round(val); val -= step; }
    }
    {
        float step = (float)(thresh_coarse - thresh_fine) / (float)(n - 1);
        float val = thresh_coarse;
        for (int l = n - 1; l >= 0; l--) { c->space_thresh[l] = val; val -= step; }
    }
    c->term_thresh = term_thresh; c->tukey_cutoff = tukey_cutoff;
    c->frames_to_skip = frames_to_skip; c->frames_to_weight = frames_to_weight;
    return GPS_OK;
}

int gps_track_state_reset(gps_track_state* ts) {
    if (!ts) return GPS_ERR_ARG;
    memset(ts, 0, sizeof(*ts));
    for (int i = 0; i < 16; i += 5) ts->pose_M[i] = ts->pose_invM[i] = ts->pose_pc_M[i] = 1.0f;
    ts->age_point_cloud = -1;
    return GPS_OK;
}

int64_t gps_track_scratch_bytes(int width, int height) {
    if (width <= 0 || height <= 0) return GPS_ERR_ARG;
    return (int64_t)carve(nullptr, nullptr, width, height);
}

int gps_tsdf_track_camera(const gps_tsdf_state* sp, const gps_track_config* c, gps_track_state* ts, void* scratch,
                          int64_t scratch_bytes, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp && c && ts && scratch);
    GPS_REQUIRE(state_valid(*sp));
    GPS_REQUIRE(c->n_levels >= 2 && c->n_levels <= GPS_TRACK_MAX_LEVELS);
    const TsdfState s = *sp;
    const int W = s.width, H = s.height;
    GPS_REQUIRE((W >> (c->n_levels - 1)) > 0 && (H >> (c->n_levels - 1)) > 0);
    if (scratch_bytes < gps_track_scratch_bytes(W, H)) return GPS_ERR_CAPACITY;
    Scratch w;
    carve(&w, (char*)scratch, W, H);
    hipStream_t st = (hipStream_t)stream;

    // PrepareForEvaluation: depth pyramid; the scene side (ICP maps) always stays at full resolution
    const float* dl[GPS_TRACK_MAX_LEVELS];
    int lw[GPS_TRACK_MAX_LEVELS], lh[GPS_TRACK_MAX_LEVELS];
    float lintr[GPS_TRACK_MAX_LEVELS][4] = {{s.fx, s.fy, s.cx, s.cy}};
    dl[0] = s.depth; lw[0] = W; lh[0] = H;
    for (int l = 1; l < c->n_levels; l++) {
        lw[l] = lw[l - 1] / 2; lh[l] = lh[l - 1] / 2;
        subsample_with_holes_kernel<<<gps_div_up(lw[l] * lh[l], 256), 256, 0, st>>>(dl[l - 1], lw[l - 1], lw[l], lh[l], w.level[l]);
        dl[l] = w.level[l];
        for (int k = 0; k < 4; k++) lintr[l][k] = lintr[l - 1][k] * 0.5f;
    }
    if (hipMemsetAsync(w.count, 0, sizeof(int), st) != hipSuccess) return GPS_ERR_LAUNCH;
    count_valid_kernel<<<128, 256, 0, st>>>(s.depth, W * H, w.count);
    interleave_maps_kernel<<<gps_div_up(W * H, 256), 256, 0, st>>>(reinterpret_cast<const float4*>(s.icp_points),
                                                                    reinterpret_cast<const float4*>(s.icp_normals), W * H, w.pn);
    GPS_LAUNCH_CHECK();

    float hessian_good[36] = {0}, nabla_good[6] = {0}, hessian_depth_good[36] = {0}, f_depth_good = 0;
    int nvalid_depth_good = 0;
    float M[16], invM[16];
    memcpy(M, ts->pose_M, 64);
    memcpy(invM, ts->pose_invM, 64);  // kept consistent with pose_M by every writer of the state
    int last_type = TRK_NONE;
    for (int k = 0; k < 16; k++) ts->diag[k] = 0;
    const int use_weights = ts->frames_processed >= 100;
    static int mail_seq = 0;  // sequence numbers are process-wide so a stale mailbox value can never match
    int mailbox_iterations = 0;

    for (int level = c->n_levels - 1; level >= 0; level--) {
        const int it = c->iter_type[level];
        if (it == TRK_NONE) continue;
        last_type = it;
        float approxInvPose[16], lastGoodM[16], lastGoodInvM[16];
        memcpy(approxInvPose, invM, 64);
        memcpy(lastGoodM, M, 64); memcpy(lastGoodInvM, invM, 64);
        float f_old = 3.402823466e+38f, lambda = 1.0f;
        const int noPara = it == TRK_BOTH ? 6 : 3;
        for (int iter = 0; iter < c->n_iter[level]; iter++) {
            GhArgs a;
            a.depth = dl[level]; a.vw = lw[level]; a.vh = lh[level];
            a.view_intr = make_float4(lintr[level][0], lintr[level][1], lintr[level][2], lintr[level][3]);
            a.pn = w.pn;
            a.sw = W; a.sh = H;
            a.scene_intr = make_float4(lintr[0][0], lintr[0][1], lintr[0][2], lintr[0][3]);
            a.approxInvPose = load_mat(approxInvPose); a.scenePose = load_mat(ts->pose_pc_M);
            a.space_thresh = c->space_thresh[level]; a.tukey_cutoff = c->tukey_cutoff; a.vf_min = s.view_frustum_min;
            a.vf_max = s.view_frustum_max; a.use_weights = use_weights; a.frames_to_skip = c->frames_to_skip;
            a.frames_to_weight = c->frames_to_weight;
            const int n_wgs = min(GH_WGS, gps_div_up(a.vw * a.vh, 256));
            volatile float* mailbox = reinterpret_cast<volatile float*>(ts->host_mailbox);
            const int seq = ++mail_seq;
            if (it == TRK_ROTATION) track_gh_kernel<TRK_ROTATION><<<n_wgs, 256, 0, st>>>(a, w.partial);
            else if (it == TRK_TRANSLATION) track_gh_kernel<TRK_TRANSLATION><<<n_wgs, 256, 0, st>>>(a, w.partial);
            else track_gh_kernel<TRK_BOTH><<<n_wgs, 256, 0, st>>>(a, w.partial);
            track_sum_kernel<<<1, 256, 0, st>>>(w.partial, n_wgs, w.result, mailbox, seq, w.count);
            GPS_LAUNCH_CHECK();
            float host[GH_SLOTS];
            if (mailbox) {
                // spin on the sequence number the kernel writes last (bounded: fall back to a stream synchronise)
                bool got = false;
                for (long spin = 0; spin < 200000000L; spin++) {
                    if (float_bits(mailbox[GH_SLOTS]) == seq) { got = true; break; }
                }
                if (!got && hipStreamSynchronize(st) != hipSuccess) return GPS_ERR_LAUNCH;
                for (int k = 0; k < GH_SLOTS; k++) host[k] = mailbox[k];
                mailbox_iterations++;
            } else {
                // the reference's GPU tracker reads its 32 accumulators back every iteration as well
                if (hipMemcpyAsync(host, w.result, sizeof(host), hipMemcpyDeviceToHost, st) != hipSuccess) return GPS_ERR_LAUNCH;
                if (hipStreamSynchronize(st) != hipSuccess) return GPS_ERR_LAUNCH;
            }

            float hessian_depth[36] = {0}, nabla_depth[6] = {0};
            const int nvalid = (int)host[0];
            float f_depth = host[1];
            for (int r = 0; r < noPara; r++) nabla_depth[r] = host[2 + r];
            for (int r = 0, counter = 0; r < noPara; r++)
                for (int cc = 0; cc <= r; cc++, counter++) hessian_depth[r + cc * 6] = host[2 + noPara + counter];
            for (int r = 0; r < noPara; ++r)
                for (int cc = r + 1; cc < noPara; cc++) hessian_depth[r + cc * 6] = hessian_depth[cc + r * 6];
            if (nvalid > 100) {
                for (int i = 0; i < 36; ++i) hessian_depth[i] /= nvalid;
                for (int i = 0; i < 6; ++i) nabla_depth[i] /= nvalid;
                f_depth /= nvalid;
            } else {
                f_depth = 3.402823466e+38f;
            }
            ts->diag[level] += 1;
            if ((nvalid <= 0) || (f_depth >= f_old)) {
                memcpy(M, lastGoodM, 64); memcpy(invM, lastGoodInvM, 64);
                memcpy(approxInvPose, invM, 64);
                lambda *= 10.0f;
            } else {
                memcpy(lastGoodM, M, 64); memcpy(lastGoodInvM, invM, 64);
                f_old = f_depth;
                memcpy(hessian_good, hessian_depth, sizeof(hessian_good));
                memcpy(nabla_good, nabla_depth, sizeof(nabla_good));
                lambda /= 10.0f;
                nvalid_depth_good = nvalid; f_depth_good = f_depth;
                memcpy(hessian_depth_good, hessian_depth, sizeof(hessian_depth));
            }
            float A[36];
            for (int i = 0; i < 36; ++i) A[i] = hessian_good[i];
            for (int i = 0; i < 6; ++i) A[i + i * 6] *= 1.0f + lambda;
            float step[6] = {0, 0, 0, 0, 0, 0};
            if (it != TRK_BOTH) {
                float small[9];
                for (int r = 0; r < 3; r++)
                    for (int cc = 0; cc < 3; cc++) small[r + cc * 3] = A[r + cc * 6];
                Chol(small, 3).backsub(step, nabla_good);
            } else {
                Chol(A, 6).backsub(step, nabla_good);
            }
            float s6[6] = {0, 0, 0, 0, 0, 0};
            if (it == TRK_ROTATION) { s6[0] = step[0]; s6[1] = step[1]; s6[2] = step[2]; }
            else if (it == TRK_TRANSLATION) { s6[3] = step[0]; s6[4] = step[1]; s6[5] = step[2]; }
            else { for (int i = 0; i < 6; i++) s6[i] = step[i]; }
            float Tinc[16];
            Tinc[0 * 4 + 0] = 1.0f;   Tinc[1 * 4 + 0] = s6[2];  Tinc[2 * 4 + 0] = -s6[1]; Tinc[3 * 4 + 0] = s6[3];
            Tinc[0 * 4 + 1] = -s6[2]; Tinc[1 * 4 + 1] = 1.0f;   Tinc[2 * 4 + 1] = s6[0];  Tinc[3 * 4 + 1] = s6[4];
            Tinc[0 * 4 + 2] = s6[1];  Tinc[1 * 4 + 2] = -s6[0]; Tinc[2 * 4 + 2] = 1.0f;   Tinc[3 * 4 + 2] = s6[5];
            Tinc[0 * 4 + 3] = 0.0f;   Tinc[1 * 4 + 3] = 0.0f;   Tinc[2 * 4 + 3] = 0.0f;   Tinc[3 * 4 + 3] = 1.0f;
            m4_mul(Tinc, approxInvPose, approxInvPose);
            if (!set_invM_coerce(approxInvPose, M, invM)) return GPS_ERR_ARG;
            memcpy(approxInvPose, invM, 64);
            bool converged = true;
            for (int i = 0; i < 6; i++)
                if (fabs(step[i]) > c->term_thresh) { converged = false; break; }
            if (converged) break;
        }
    }
    memcpy(ts->pose_M, M, 64); memcpy(ts->pose_invM, invM, 64);
    // UpdatePoseQuality: the residual score (the SVM verdict only feeds failure modes that are off by default,
    // ITMLibSettings.cpp:42 behaviourOnFailure = FAILUREMODE_IGNORE)
    int n_max = 0;
    if (ts->host_mailbox && mailbox_iterations > 0) {
        n_max = float_bits(reinterpret_cast<volatile float*>(ts->host_mailbox)[GH_SLOTS + 1]);  // delivered with the last iteration
    } else {
        if (hipMemcpyAsync(&n_max, w.count, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return GPS_ERR_LAUNCH;
        if (hipStreamSynchronize(st) != hipSuccess) return GPS_ERR_LAUNCH;
    }
    ts->diag[8] = (float)nvalid_depth_good; ts->diag[9] = f_depth_good;
    ts->diag[10] = n_max > 0 ? sqrtf(((float)nvalid_depth_good * f_depth_good + (float)(n_max - nvalid_depth_good) * c->space_thresh[0]) /
                                    (float)n_max) : 0.0f;
    float det = 0.0f;
    if (last_type == TRK_BOTH) { det = Chol(hessian_depth_good, 6).determinant(); if (isnan(det)) det = 0.0f; }
    ts->diag[11] = det;
    return GPS_OK;
}

int gps_tsdf_process_frame_tracked(const gps_tsdf_state* s, const int16_t* depth_mm, const gps_track_config* cfg,
                                   gps_track_state* ts, void* scratch, int64_t scratch_bytes, gps_stream stream) {
    return gps_tsdf_process_frame_tracked_gated(s, depth_mm, cfg, ts, scratch, scratch_bytes, stream, nullptr, nullptr);
}

int gps_tsdf_process_frame_tracked_gated(const gps_tsdf_state* s, const int16_t* depth_mm, const gps_track_config* cfg,
                                         gps_track_state* ts, void* scratch, int64_t scratch_bytes, gps_stream stream,
                                         void (*before_fusion)(void*), void* user) {
    GPS_REQUIRE(s && depth_mm && cfg && ts);
    int r;
    if ((r = gps_tsdf_convert_depth(s, depth_mm, stream)) != GPS_OK) return r;
    if (ts->age_point_cloud != -1) {  // ITMTrackingState::HasValidPointCloud
        if (ts->age_point_cloud >= 0) ts->frames_processed++; else ts->frames_processed = 0;
        if ((r = gps_tsdf_track_camera(s, cfg, ts, scratch, scratch_bytes, stream)) != GPS_OK) return r;
    }
    if (before_fusion) before_fusion(user);  // everything above only READ the volume; what follows modifies it
    if ((r = gps_tsdf_allocate(s, ts->pose_M, ts->pose_invM, stream)) != GPS_OK) return r;
    if ((r = gps_tsdf_integrate(s, ts->pose_M, stream)) != GPS_OK) return r;
    if ((r = gps_tsdf_expected_depths(s, ts->pose_M, 0, stream)) != GPS_OK) return r;
    if ((r = gps_tsdf_raycast(s, ts->pose_invM, 0, 1, stream)) != GPS_OK) return r;
    if ((r = gps_tsdf_icp_maps(s, ts->pose_invM, stream)) != GPS_OK) return r;
    memcpy(ts->pose_pc_M, ts->pose_M, 64);  // pose_pointCloud := pose_d (ITMTrackingController.h:87-93)
    ts->age_point_cloud = (ts->age_point_cloud == -1) ? -2 : 0;
    return GPS_OK;
}

}  // extern "C"

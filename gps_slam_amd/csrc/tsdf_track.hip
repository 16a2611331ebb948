// Depth-only ExtendedTracker: the tracker ITMLibSettings.cpp:54-57 configures and ITMBasicEngine
// runs when use_gt_pose is false.
//
//   ITMLib/Trackers/Interface/ITMExtendedTracker.cpp:143-177 (SetupLevels), :216-268 (PrepareForEvaluation: depth pyramid),
//   :293-375 (ComputeDelta / HasConverged / ApplyDelta), :470-665 (TrackCamera);
//   Trackers/Shared/ITMExtendedTracker_Shared.h:51-143, 298-328 (per point); Trackers/CUDA/ITMExtendedTracker_CUDA.cu
//   (per-iteration reduction + 32-float read-back); Engines/LowLevel/Shared/ITMLowLevelEngine_Shared.h:48-69;
//   ORUtils/Cholesky.h, ORUtils/SE3Pose.cpp (tsdf_pose.hpp).
//
// The reference's GPU tracker is a host loop: per Levenberg-Marquardt iteration one evaluation kernel, a 32-float read-back,
// the 6x6 solve + SE3 update on the host, the next launch -- ~18 dependent host round trips per frame, which on this path
// was the frame stream's critical chain (round 1: 0.37 of 0.70 ms per frame).  Here the WHOLE of TrackCamera is one
// persistent launch (track_lm_kernel): pyramid, then per level a device-side loop
//     evaluate all pixels -> one row of partial sums per workgroup -> grid rendezvous -> every workgroup adds the rows in a
//     fixed order and runs the LM bookkeeping (Cholesky, damping, ApplyDelta, SetInvM + Coerce, convergence test) on its own
//     LDS copy of the state
// and ONE read-back per frame (pose + diagnostics, written straight into a pinned host mailbox).  The reduction is a fixed
// tree (per-thread register sums -> wave -> workgroup -> 16 row groups added in order), so results are reproducible run to
// run; they differ from the CPU engine's scan-order sums only by float re-association (poses agree to ~1e-5,
// tests/test_tsdf_gpu.py).
#include <math.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>

#include "tsdf_common.hpp"
#include "tsdf_pose.hpp"

using namespace gpst;

namespace {

constexpr int TRK_ROTATION = 0, TRK_TRANSLATION = 1, TRK_BOTH = 2, TRK_NONE = 3;
constexpr int GH_SLOTS = 32;  // one row of sums: 0 = count, 1 = f, 2.. = nabla, then lower-triangular hessian

// The scene side as the tracker reads it: point and normal of a pixel in ONE 32-byte record (pn[2i], pn[2i+1]).  The two maps
// are separate arrays in gps_tsdf_state (ICP maps, API-visible); gathering a bilinear footprint from both touches 4 + 4 cache
// lines in two dependent phases (normals only after the distance test).  Interleaved, the footprint is 2 x 64 contiguous bytes
// fetched in one phase -- half the lines to keep in (or re-fetch into) an L2 the map stream's rasterizer kernels keep sweeping.
__device__ __forceinline__ float4 bilerp4(const float4 a, const float4 b, const float4 c, const float4 d, float dx, float dy) {
    if (a.w < 0 || b.w < 0 || c.w < 0 || d.w < 0) return make_float4(0.f, 0.f, 0.f, -1.0f);
    float4 r;
    r.x = (a.x * (1.0f - dx) * (1.0f - dy) + b.x * dx * (1.0f - dy) + c.x * (1.0f - dx) * dy + d.x * dx * dy);
    r.y = (a.y * (1.0f - dx) * (1.0f - dy) + b.y * dx * (1.0f - dy) + c.y * (1.0f - dx) * dy + d.y * dx * dy);
    r.z = (a.z * (1.0f - dx) * (1.0f - dy) + b.z * dx * (1.0f - dy) + c.z * (1.0f - dx) * dy + d.z * dx * dy);
    r.w = (a.w * (1.0f - dx) * (1.0f - dy) + b.w * dx * (1.0f - dy) + c.w * (1.0f - dx) * dy + d.w * dx * dy);
    return r;
}

// interpolateBilinear_withHoles (Utils/ITMPixelUtils.h:78-106) of the point map and of the normal map at the same position
// (each with its own hole rule)
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
    const float4* pn;  // interleaved point | normal records
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

// ---------------------------------------------------------------- the Levenberg-Marquardt loop, on the device
// ORUtils::Cholesky (ORUtils/Cholesky.h) for the 3x3 / 6x6 normal equations.  The size is a template parameter so that every
// loop unrolls and the factor lives in registers (with a run-time size the arrays go to scratch memory -- allocated for every
// lane of every resident wave although one thread per workgroup solves).
template <int N>
struct Chol {
    float ch[N * N];
    __device__ __forceinline__ explicit Chol(const float* mat) {
#pragma unroll
        for (int i = 0; i < N * N; i++) ch[i] = mat[i];
#pragma unroll
        for (int c = 0; c < N; c++) {
            float inv_diag = 1;
#pragma unroll
            for (int r = c; r < N; r++) {
                float val = ch[c + r * N];
#pragma unroll
                for (int c2 = 0; c2 < c; c2++) val -= ch[c + c2 * N] * ch[c2 + r * N];
                if (r == c) { ch[c + r * N] = val; inv_diag = 1.0f / val; }
                else { ch[r + c * N] = val; ch[c + r * N] = val * inv_diag; }
            }
        }
    }
    __device__ __forceinline__ void backsub(float* result, const float* v) const {
        float y[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            float val = v[i];
#pragma unroll
            for (int j = 0; j < i; j++) val -= ch[j + i * N] * y[j];
            y[i] = val;
        }
#pragma unroll
        for (int i = 0; i < N; i++) y[i] /= ch[i + i * N];
#pragma unroll
        for (int i = N - 1; i >= 0; i--) {
            float val = y[i];
#pragma unroll
            for (int j = i + 1; j < N; j++) val -= ch[i + j * N] * result[j];
            result[i] = val;
        }
    }
    __device__ __forceinline__ float determinant() const {
        float ret = 1.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) ret *= ch[i + i * N];
        return ret * ret;
    }
};

__device__ __forceinline__ void m4_mul(const float* a, const float* b, float* out) {  // ORUtils Matrix4 operator*
    float r[16];
#pragma unroll
    for (int col = 0; col < 4; col++)
#pragma unroll
        for (int row = 0; row < 4; row++) {
            float acc = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) acc += a[k * 4 + row] * b[col * 4 + k];
            r[col * 4 + row] = acc;
        }
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = r[i];
}

constexpr int LM_THREADS = 512;         // 8 waves with a 256-VGPR budget each: the 29 accumulators + 8 gathered float4 of the
                                        // 6-parameter evaluation fit without spilling (a 1024-thread group spilled 50+ registers)
constexpr int LM_MAX_WGS = 256;         // rows of the partial table; <= one workgroup per CU: co-resident by construction
constexpr int LM_ROW_GROUPS = LM_THREADS / 32;
constexpr int RES_FLOATS = 64;          // result block: M[16] | invM[16] | diag[16] | status, valid-pixel count | seq
constexpr uint32_t LM_SPIN_LIMIT = 1u << 22;

struct LmArgs {
    gps_track_config cfg;
    const float* depth0;                // full-resolution depth (s.depth)
    float* level[GPS_TRACK_MAX_LEVELS]; // [0] unused
    const float4* points;               // ICP maps of the last raycast
    const float4* normals;
    float4* pn;
    int W, H;
    float4 intr;                        // fx fy cx cy of level 0
    float vf_min, vf_max;
    int use_weights;
    Mat4 pose_M, pose_invM, pose_pc_M;
    uint32_t* partial;                  // [2][LM_MAX_WGS][GH_SLOTS] float bits
    uint32_t* sync;                     // [0] rendezvous arrivals, [1] valid-pixel count (both zeroed before the launch)
    float* result;                      // device [RES_FLOATS]
    volatile float* mailbox;            // pinned host copy of `result` with the sequence number last, or nullptr
    int seq;
};

// The state the host loop of ITMExtendedTracker::TrackCamera keeps between iterations (ITMExtendedTracker.cpp:470-665).
// Every workgroup holds an identical copy in LDS and updates it with identical arithmetic from identical totals, so the new
// pose needs no broadcast: ONE grid-wide rendezvous per iteration.
struct LmState {
    float M[16], invM[16], approxInvPose[16], lastGoodM[16], lastGoodInvM[16];
    float hessian_good[36], nabla_good[6], hessian_depth_good[36];
    float f_old, lambda, f_depth_good;
    int nvalid_depth_good, last_type, iters[GPS_TRACK_MAX_LEVELS];
    int converged, failed;
};

// Grid-wide rendezvous without L2 flushes.  The XCDs' L2s are not coherent with each other: an agent-scope release / acquire
// pair means buffer_wbl2 + buffer_inv of the WHOLE L2 -- per workgroup, per iteration, with the map stream's rasterizer
// sharing that L2 (measured in round 1 on a last-block-done ticket: 758 -> 600 frames/s).  Everything exchanged between
// workgroups inside the iteration loop therefore goes through agent-scope RELAXED atomics (sc1 stores / loads: write-through
// to, and read from, the memory-side coherence point), ordered by program order + s_waitcnt: the partial row is stored,
// vmcnt(0) waits for the stores' acknowledgements, then the arrival counter is incremented; a workgroup that has seen the
// count reads the rows with sc1 loads.  No cache line of anything else is written back or invalidated.  The spin is bounded:
// a workgroup that never sees the count reports failure instead of hanging the queue.
__device__ __forceinline__ bool grid_rendezvous(uint32_t* counter, uint32_t target) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt vmcnt(0): this wave's sc1 stores are acknowledged
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < LM_SPIN_LIMIT)
            __builtin_amdgcn_s_sleep(2);
        ok = spins < LM_SPIN_LIMIT;
    }
    __syncthreads();
    return ok != 0;
}

// PrepareForEvaluation (ITMExtendedTracker.cpp:216-268) in one pass: a work item is one 2^(L-1) x 2^(L-1) block of
// full-resolution pixels = one pixel of the coarsest level; it produces every pyramid pixel above that block locally
// (filterSubsampleWithHoles is hierarchical: a level-l pixel is the mean of the VALID pixels among its four level-(l-1)
// children, ITMLowLevelEngine_Shared.h:48-69), counts the valid full-resolution pixels and interleaves the ICP maps.
__device__ __forceinline__ void build_pyramid(const LmArgs& a, int n_wgs) {
    const int L = a.cfg.n_levels;
    const int B = 1 << (L - 1);  // block edge in level-0 pixels
    const int bw = (a.W + B - 1) / B, bh = (a.H + B - 1) / B;
    // level sizes: repeated floor halving == a right shift
    for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < bw * bh; item += n_wgs * blockDim.x) {
        const int by = item / bw, bx = item - by * bw;
        // bottom-up: level l reads the level below from memory THIS thread wrote (level 0: the input); no other thread
        // touches the block
        for (int l = 1; l < L; l++) {
            const int e = B >> l;  // edge of the block at level l
            const float* src = l == 1 ? a.depth0 : a.level[l - 1];
            const int w_in = a.W >> (l - 1), lwl = a.W >> l, lhl = a.H >> l;
            for (int yy = 0; yy < e; yy++)
                for (int xx = 0; xx < e; xx++) {
                    const int x = bx * e + xx, y = by * e + yy;
                    if (x >= lwl || y >= lhl) continue;
                    float acc = 0.0f, good = 0.0f, v;
                    v = src[(2 * x + 0) + (2 * y + 0) * w_in]; if (v > 0.0f) { acc += v; good++; }
                    v = src[(2 * x + 1) + (2 * y + 0) * w_in]; if (v > 0.0f) { acc += v; good++; }
                    v = src[(2 * x + 0) + (2 * y + 1) * w_in]; if (v > 0.0f) { acc += v; good++; }
                    v = src[(2 * x + 1) + (2 * y + 1) * w_in]; if (v > 0.0f) { acc += v; good++; }
                    if (good > 0) acc /= good;
                    a.level[l][x + y * lwl] = acc;
                }
        }
    }
    int valid = 0;
    const int n = a.W * a.H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += n_wgs * blockDim.x) {
        valid += a.depth0[i] > 0.0f ? 1 : 0;
        a.pn[2 * i] = a.points[i];
        a.pn[2 * i + 1] = a.normals[i];
    }
    for (int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o, 64);
    if ((threadIdx.x & 63) == 0 && valid) __hip_atomic_fetch_add(&a.sync[1], (uint32_t)valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int ITER>
__device__ __forceinline__ void eval_level(const GhArgs& g, int n_wgs, float* row /* LDS [GH_SLOTS] */) {
    constexpr int NP = ITER == TRK_BOTH ? 6 : 3, NSQ = ITER == TRK_BOTH ? 21 : 6, NV = 2 + NP + NSQ;
    __shared__ float red[LM_THREADS / 64][GH_SLOTS];
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = 0.0f;
    const int n = g.vw * g.vh;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += n_wgs * blockDim.x) {
        const int y = i / g.vw, x = i - y * g.vw;
        gh_point<ITER>(g, x, y, acc[0], acc[1], acc + 2, acc + 2 + NP);
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
    if (threadIdx.x < GH_SLOTS) {
        float t = 0.0f;
        if (threadIdx.x < NV)
            for (int w = 0; w < LM_THREADS / 64; w++) t += red[w][threadIdx.x];  // fixed order
        row[threadIdx.x] = t;
    }
    __syncthreads();
}

// One LM iteration's bookkeeping from the 32 totals (ITMExtendedTracker.cpp:560-640: normalise, accept / reject, damped
// solve, ApplyDelta, SetInvM + Coerce, HasConverged) -- what thread 0 of every workgroup runs on its own LDS copy.
// IT (the level's iteration type) is a template parameter: with compile-time sizes every small array below is registers.
template <int IT>
__device__ __noinline__ void lm_update(LmState& S, float term_thresh, int level, const float* tot) {
    constexpr int noPara = IT == TRK_BOTH ? 6 : 3;
    float hessian_depth[36], nabla_depth[6];
#pragma unroll
    for (int i = 0; i < 36; i++) hessian_depth[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; i++) nabla_depth[i] = 0.0f;
    const int nvalid = (int)tot[0];
    float f_depth = tot[1];
#pragma unroll
    for (int r = 0; r < noPara; r++) nabla_depth[r] = tot[2 + r];
    {
        int counter = 0;
#pragma unroll
        for (int r = 0; r < noPara; r++)
#pragma unroll
            for (int cc = 0; cc <= r; cc++, counter++) hessian_depth[r + cc * 6] = tot[2 + noPara + counter];
    }
#pragma unroll
    for (int r = 0; r < noPara; ++r)
#pragma unroll
        for (int cc = r + 1; cc < noPara; cc++) hessian_depth[r + cc * 6] = hessian_depth[cc + r * 6];
    if (nvalid > 100) {
#pragma unroll
        for (int i = 0; i < 36; ++i) hessian_depth[i] /= nvalid;
#pragma unroll
        for (int i = 0; i < 6; ++i) nabla_depth[i] /= nvalid;
        f_depth /= nvalid;
    } else {
        f_depth = 3.402823466e+38f;
    }
    S.iters[level] += 1;
    if ((nvalid <= 0) || (f_depth >= S.f_old)) {
#pragma unroll
        for (int i = 0; i < 16; i++) { S.M[i] = S.lastGoodM[i]; S.invM[i] = S.lastGoodInvM[i]; S.approxInvPose[i] = S.invM[i]; }
        S.lambda *= 10.0f;
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) { S.lastGoodM[i] = S.M[i]; S.lastGoodInvM[i] = S.invM[i]; }
        S.f_old = f_depth;
#pragma unroll
        for (int i = 0; i < 36; i++) { S.hessian_good[i] = hessian_depth[i]; S.hessian_depth_good[i] = hessian_depth[i]; }
#pragma unroll
        for (int i = 0; i < 6; i++) S.nabla_good[i] = nabla_depth[i];
        S.lambda /= 10.0f;
        S.nvalid_depth_good = nvalid; S.f_depth_good = f_depth;
    }
    float A[36], nabla[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) A[i] = S.hessian_good[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) { A[i + i * 6] *= 1.0f + S.lambda; nabla[i] = S.nabla_good[i]; }
    float step[6] = {0, 0, 0, 0, 0, 0};
    if (IT != TRK_BOTH) {
        float small[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) small[r + cc * 3] = A[r + cc * 6];
        Chol<3>(small).backsub(step, nabla);
    } else {
        Chol<6>(A).backsub(step, nabla);
    }
    float s6[6] = {0, 0, 0, 0, 0, 0};
    if (IT == TRK_ROTATION) { s6[0] = step[0]; s6[1] = step[1]; s6[2] = step[2]; }
    else if (IT == TRK_TRANSLATION) { s6[3] = step[0]; s6[4] = step[1]; s6[5] = step[2]; }
    else {
#pragma unroll
        for (int i = 0; i < 6; i++) s6[i] = step[i];
    }
    float Tinc[16];
    Tinc[0 * 4 + 0] = 1.0f;   Tinc[1 * 4 + 0] = s6[2];  Tinc[2 * 4 + 0] = -s6[1]; Tinc[3 * 4 + 0] = s6[3];
    Tinc[0 * 4 + 1] = -s6[2]; Tinc[1 * 4 + 1] = 1.0f;   Tinc[2 * 4 + 1] = s6[0];  Tinc[3 * 4 + 1] = s6[4];
    Tinc[0 * 4 + 2] = s6[1];  Tinc[1 * 4 + 2] = -s6[0]; Tinc[2 * 4 + 2] = 1.0f;   Tinc[3 * 4 + 2] = s6[5];
    Tinc[0 * 4 + 3] = 0.0f;   Tinc[1 * 4 + 3] = 0.0f;   Tinc[2 * 4 + 3] = 0.0f;   Tinc[3 * 4 + 3] = 1.0f;
    float cur[16], next[16], M[16], invM[16];
#pragma unroll
    for (int i = 0; i < 16; i++) cur[i] = S.approxInvPose[i];
    m4_mul(Tinc, cur, next);
    // pose_d->SetInvM(approxInvPose); pose_d->Coerce(); approxInvPose = pose_d->GetInvM()
    if (!pose_set_invM_coerce(next, M, invM)) { S.failed = 1; return; }
#pragma unroll
    for (int i = 0; i < 16; i++) { S.M[i] = M[i]; S.invM[i] = invM[i]; S.approxInvPose[i] = invM[i]; }
    bool converged = true;
#pragma unroll
    for (int i = 0; i < 6; i++)
        if (fabsf(step[i]) > term_thresh) converged = false;
    S.converged = converged ? 1 : 0;
}

// ITMExtendedTracker::TrackCamera as ONE launch.
__global__ __launch_bounds__(LM_THREADS) void track_lm_kernel(LmArgs a) {
    __shared__ LmState S;
    __shared__ float row[GH_SLOTS];
    __shared__ float group[LM_ROW_GROUPS][GH_SLOTS];
    __shared__ float tot[GH_SLOTS];
    const int n_wgs = gridDim.x;
    const int tid = threadIdx.x;
    uint32_t target = 0;
    bool alive = true;

    build_pyramid(a, n_wgs);
    if (tid == 0) {
        for (int i = 0; i < 16; i++) { S.M[i] = a.pose_M.m[i]; S.invM[i] = a.pose_invM.m[i]; }
        for (int i = 0; i < 36; i++) { S.hessian_good[i] = 0.0f; S.hessian_depth_good[i] = 0.0f; }
        for (int i = 0; i < 6; i++) S.nabla_good[i] = 0.0f;
        for (int i = 0; i < GPS_TRACK_MAX_LEVELS; i++) S.iters[i] = 0;
        S.f_depth_good = 0.0f; S.nvalid_depth_good = 0; S.last_type = TRK_NONE; S.failed = 0; S.converged = 0;
    }
    // The pyramid and the interleaved maps are plain stores that OTHER workgroups (other XCDs) read afterwards: write them
    // back and drop stale lines ONCE per frame (agent-scope release before / acquire after the first rendezvous).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    target += n_wgs;
    alive = grid_rendezvous(&a.sync[0], target);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

    int parity = 0;
    for (int level = a.cfg.n_levels - 1; level >= 0 && alive; level--) {
        const int it = a.cfg.iter_type[level];
        if (it == TRK_NONE) continue;
        if (tid == 0) {
            S.last_type = it;
            for (int i = 0; i < 16; i++) { S.approxInvPose[i] = S.invM[i]; S.lastGoodM[i] = S.M[i]; S.lastGoodInvM[i] = S.invM[i]; }
            S.f_old = 3.402823466e+38f; S.lambda = 1.0f; S.converged = 0;
        }
        __syncthreads();
        GhArgs g;
        g.depth = level == 0 ? a.depth0 : a.level[level];
        g.vw = a.W >> level; g.vh = a.H >> level;
        // the reference halves the intrinsics level by level (ITMExtendedTracker.cpp:230-233): repeated * 0.5f is exact
        float4 vi = a.intr;
        for (int l = 0; l < level; l++) { vi.x *= 0.5f; vi.y *= 0.5f; vi.z *= 0.5f; vi.w *= 0.5f; }
        g.view_intr = vi;
        g.pn = a.pn; g.sw = a.W; g.sh = a.H;
        g.scene_intr = a.intr;
        g.scenePose = a.pose_pc_M;
        g.space_thresh = a.cfg.space_thresh[level]; g.tukey_cutoff = a.cfg.tukey_cutoff;
        g.vf_min = a.vf_min; g.vf_max = a.vf_max; g.use_weights = a.use_weights;
        g.frames_to_skip = a.cfg.frames_to_skip; g.frames_to_weight = a.cfg.frames_to_weight;
        for (int iter = 0; iter < a.cfg.n_iter[level] && alive; iter++) {
            // wave-uniform values read from LDS: pin them to SGPRs (16 VGPRs less in a loop that needs every one)
#pragma unroll
            for (int i = 0; i < 16; i++)
                g.approxInvPose.m[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(S.approxInvPose[i])));
            if (it == TRK_ROTATION) eval_level<TRK_ROTATION>(g, n_wgs, row);
            else if (it == TRK_TRANSLATION) eval_level<TRK_TRANSLATION>(g, n_wgs, row);
            else eval_level<TRK_BOTH>(g, n_wgs, row);
            uint32_t* mine = a.partial + ((size_t)parity * LM_MAX_WGS + blockIdx.x) * GH_SLOTS;
            if (tid < GH_SLOTS) __hip_atomic_store(mine + tid, __float_as_uint(row[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            target += n_wgs;
            alive = grid_rendezvous(&a.sync[0], target);
            // every workgroup adds all rows: slot k of rows r, r + 16, ... by thread (r, k), then the row groups in order
            {
                const int k = tid & (GH_SLOTS - 1), r = tid >> 5;  // LM_ROW_GROUPS row groups x 32 slots
                const uint32_t* tab = a.partial + (size_t)parity * LM_MAX_WGS * GH_SLOTS;
                float s = 0.0f;
                for (int rr = r; rr < n_wgs; rr += LM_ROW_GROUPS)
                    s += __uint_as_float(__hip_atomic_load(tab + rr * GH_SLOTS + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                group[r][k] = s;
            }
            __syncthreads();
            if (tid < GH_SLOTS) {
                float t = 0.0f;
#pragma unroll
                for (int r = 0; r < LM_ROW_GROUPS; r++) t += group[r][tid];
                tot[tid] = t;
            }
            __syncthreads();
            if (tid == 0) {
                if (it == TRK_ROTATION) lm_update<TRK_ROTATION>(S, a.cfg.term_thresh, level, tot);
                else if (it == TRK_TRANSLATION) lm_update<TRK_TRANSLATION>(S, a.cfg.term_thresh, level, tot);
                else lm_update<TRK_BOTH>(S, a.cfg.term_thresh, level, tot);
            }
            __syncthreads();
            parity ^= 1;
            if (S.failed) alive = false;
            if (S.converged) break;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        float* res = a.result;
        for (int i = 0; i < 16; i++) { res[i] = S.M[i]; res[16 + i] = S.invM[i]; }
        for (int i = 0; i < 8; i++) res[32 + i] = (float)S.iters[i];
        const int n_max = (int)__hip_atomic_load(&a.sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        res[40] = (float)S.nvalid_depth_good;
        res[41] = S.f_depth_good;
        // UpdatePoseQuality: the residual score (the SVM verdict only feeds failure modes that are off by default,
        // ITMLibSettings.cpp:42 behaviourOnFailure = FAILUREMODE_IGNORE)
        res[42] = n_max > 0 ? sqrtf(((float)S.nvalid_depth_good * S.f_depth_good +
                                     (float)(n_max - S.nvalid_depth_good) * a.cfg.space_thresh[0]) / (float)n_max) : 0.0f;
        float det = 0.0f;
        if (S.last_type == TRK_BOTH) {
            float hg[36];
#pragma unroll
            for (int i = 0; i < 36; i++) hg[i] = S.hessian_depth_good[i];
            det = Chol<6>(hg).determinant();
            if (isnan(det)) det = 0.0f;
        }
        res[43] = det;
        for (int i = 44; i < 48; i++) res[i] = 0.0f;
        res[48] = alive ? 1.0f : -1.0f;  // status: -1 = a rendezvous timed out or a singular pose
        res[49] = __int_as_float(n_max);
        if (a.mailbox) {
            for (int i = 0; i < 50; i++) a.mailbox[i] = res[i];
            __threadfence_system();
            a.mailbox[50] = __int_as_float(a.seq);
            __threadfence_system();
        }
    }
}

inline int float_bits(float f) { int i; memcpy(&i, &f, 4); return i; }

struct Scratch {
    float* level[GPS_TRACK_MAX_LEVELS];  // [0] unused (= s.depth)
    uint32_t *partial, *sync;
    float* result;
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
    char* p = take((size_t)2 * LM_MAX_WGS * GH_SLOTS * sizeof(uint32_t)); if (w) w->partial = (uint32_t*)p;
    p = take(RES_FLOATS * sizeof(float)); if (w) w->result = (float*)p;
    p = take(64); if (w) w->sync = (uint32_t*)p;
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
    void* mailbox = ts->host_mailbox;
    const int32_t seq = ts->mail_seq;
    memset(ts, 0, sizeof(*ts));
    ts->host_mailbox = mailbox;
    ts->mail_seq = seq;
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

    LmArgs a;
    a.cfg = *c;
    a.depth0 = s.depth;
    for (int l = 0; l < GPS_TRACK_MAX_LEVELS; l++) a.level[l] = l == 0 ? nullptr : w.level[l];
    a.points = reinterpret_cast<const float4*>(s.icp_points);
    a.normals = reinterpret_cast<const float4*>(s.icp_normals);
    a.pn = w.pn;
    a.W = W; a.H = H;
    a.intr = make_float4(s.fx, s.fy, s.cx, s.cy);
    a.vf_min = s.view_frustum_min; a.vf_max = s.view_frustum_max;
    a.use_weights = ts->frames_processed >= 100;
    a.pose_M = load_mat(ts->pose_M);
    a.pose_invM = load_mat(ts->pose_invM);  // kept consistent with pose_M by every writer of the state
    a.pose_pc_M = load_mat(ts->pose_pc_M);
    a.partial = w.partial; a.sync = w.sync; a.result = w.result;
    volatile float* mailbox = reinterpret_cast<volatile float*>(ts->host_mailbox);
    a.mailbox = mailbox;
    // the sequence number the kernel writes last: unique per call on this state (a stale mailbox value can never match)
    const int seq = (int)(ts->mail_seq = ts->mail_seq >= 0x3FFFFFFF ? 1 : ts->mail_seq + 1);  // >= 1
    a.seq = seq;
    if (mailbox) mailbox[50] = 0.0f;  // 0 is never a sequence number: nothing stale can match
    if (hipMemsetAsync(w.sync, 0, 64, st) != hipSuccess) return GPS_ERR_LAUNCH;
    // enough workgroups to give every thread ~2 pixels of the finest level, never more rows than the partial table holds
    const int n_wgs = max(1, min(LM_MAX_WGS, gps_div_up((int64_t)W * H, 2 * LM_THREADS)));
    track_lm_kernel<<<n_wgs, LM_THREADS, 0, st>>>(a);
    GPS_LAUNCH_CHECK();

    float host[RES_FLOATS];
    if (mailbox) {
        // ONE wait per frame: spin on the sequence number, bounded (fall back to a stream synchronise); yield now and then so
        // that an oversubscribed host (8 ranks x 2 threads) does not burn a core per rank for nothing
        bool got = false;
        for (long spin = 0; spin < 400000000L; spin++) {
            if (float_bits(mailbox[50]) == seq) { got = true; break; }
            if ((spin & 0x3FFF) == 0x3FFF) sched_yield();
        }
        if (!got && hipStreamSynchronize(st) != hipSuccess) return GPS_ERR_LAUNCH;
        for (int k = 0; k < 50; k++) host[k] = mailbox[k];
    } else {
        if (hipMemcpyAsync(host, w.result, 50 * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess) return GPS_ERR_LAUNCH;
        if (hipStreamSynchronize(st) != hipSuccess) return GPS_ERR_LAUNCH;
    }
    if (!(host[48] > 0.0f)) return GPS_ERR_LAUNCH;  // rendezvous time-out / singular pose: reported, never silent
    memcpy(ts->pose_M, host, 64);
    memcpy(ts->pose_invM, host + 16, 64);
    for (int k = 0; k < 16; k++) ts->diag[k] = 0;
    for (int k = 0; k < 8; k++) ts->diag[k] = host[32 + k];
    ts->diag[8] = host[40]; ts->diag[9] = host[41]; ts->diag[10] = host[42]; ts->diag[11] = host[43];
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

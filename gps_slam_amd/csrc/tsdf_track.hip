// Depth-only ExtendedTracker: the tracker ITMLibSettings.cpp:54-57 configures and ITMBasicEngine
// runs when use_gt_pose is false.
//
//   ITMLib/Trackers/Interface/ITMExtendedTracker.cpp:143-177 (SetupLevels), :216-268 (PrepareForEvaluation: depth pyramid),
//   :293-375 (ComputeDelta / HasConverged / ApplyDelta), :470-665 (TrackCamera);
//   Trackers/Shared/ITMExtendedTracker_Shared.h:51-143, 298-328 (per point); Trackers/CUDA/ITMExtendedTracker_CUDA.cu
//   (per-iteration reduction + 32-float read-back); Engines/LowLevel/Shared/ITMLowLevelEngine_Shared.h:48-69;
//   ORUtils/Cholesky.h, ORUtils/SE3Pose.cpp (tsdf_pose.hpp).
//
// Split like the reference's GPU tracker -- the per-pixel residual / Jacobian evaluation and its reduction are kernels, the
// 6x6 Levenberg-Marquardt bookkeeping (Cholesky, step, SE3 coercion, accept / reject) is host code fed by 32 floats per
// iteration -- with the launch train around it cut down:
//
//   track_prepare_tile_kernel depth pyramid (all levels) + valid-pixel count + interleaved ICP maps: ONE launch per frame
//                             (the reference: one subsample kernel per level; round 1 here: 5 launches)
//   track_eval_poll_kernel    one LM iteration, PRE-LAUNCHED: its workgroups wait for a 64-byte argument line (level, kind,
//                             pose) the host writes once it has decided; every workgroup evaluates its pixels and stores one
//                             sequence-tagged row of partial sums; workgroup 0 re-reads the rows until all are this launch's,
//                             adds them in a fixed order and writes the totals into a pinned host mailbox the host spins on
//                             (no summing kernel, no memcpy, no stream synchronise, no ticket; the only acknowledgement is
//                             the one a RETIRED launch gives before the host reuses the argument line, once per frame)
//   track_eval_kernel<ITER>   the same body as a plain launch with kernel arguments (host_mailbox == NULL, or the fall-back)
//
// What was tried in round 2 and measured slower (kept out of the tree): (1) the whole LM loop in ONE persistent launch with a
// grid-wide rendezvous per iteration -- correct, but all its workgroups must be resident at once, and next to the map stream's
// rasterizer kernels the big spinning workgroups starved behind a steady supply of small ones (800 -> 120 frames/s with the
// mapping overlap on); (2) a device-side state machine, one ordinary launch per iteration, the host only topping the queue up
// ahead of a progress counter -- no host round trip at all, yet 0.90 instead of 0.71 ms per tracked frame: the 6x6 Cholesky +
// two 4x4 inverses + SE3 log / exp run as ONE dependent instruction chain in one lane (~10 us; the host does it in < 1 us,
// i.e. less than the PCIe round trip it saves), and carrying that code raised the evaluation's register allocation to 164
// VGPRs (3 waves per SIMD), tripling the gather-latency-bound evaluation time.
// The reduction is a fixed tree (per-thread register sums -> wave -> workgroup -> 8 row groups added in order), so results
// are reproducible run to run; they differ from the CPU engine's scan-order sums only by float re-association (poses agree
// to ~1e-5, tests/test_tsdf_gpu.py).
#if defined(__x86_64__)
#include <immintrin.h>
#include <x86intrin.h>
#endif
#include <math.h>
#include <sched.h>
#include <setjmp.h>
#include <mutex>
#include <pthread.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>

#include "tsdf_common.hpp"
#include "wave_reduce.hpp"
#include "tsdf_pose.hpp"

// a failure of the tracker's host loop names its line on stderr (as GPS_LAUNCH_CHECK does for launches): -2 alone says nothing
#define GPS_FAIL_LAUNCH()                                                                                   \
    do {                                                                                                    \
        fprintf(stderr, "[gps_slam_hip] %s:%d tracker: no result / hip error (%s)\n", __FILE__, __LINE__,  \
                hipGetErrorString(hipPeekAtLastError()));                                                   \
        return GPS_ERR_LAUNCH;                                                                              \
    } while (0)

using namespace gpst;

// Host-side store fence and cycle counter of the argument-line protocol.  x86-64: sfence drains the write-combining buffer a
// BAR line gathers in (see publish()), rdtsc is the free diagnostic clock of the poll loop; elsewhere a full fence and the
// steady clock (nanoseconds) stand in.
#if defined(__x86_64__)
static inline void host_store_fence() { _mm_sfence(); }
static inline unsigned long long host_cycles() { return __rdtsc(); }
static inline bool host_flushes_denormals() { return (_mm_getcsr() & 0x8040u) != 0; }   // MXCSR: FTZ (bit 15) | DAZ (bit 6)
#else
#include <time.h>
static inline void host_store_fence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline bool host_flushes_denormals() { return false; }
static inline unsigned long long host_cycles() {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (unsigned long long)t.tv_sec * 1000000000ull + (unsigned long long)t.tv_nsec;
}
#endif

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
// Split in two around the bilinear footprint's eight 16-byte loads so that a thread can have the footprints of ALL its pixels (and of
// all the poses of a fused evaluation) in flight at once: gh_project decides where the footprint lies (no memory access, no
// branch), the caller issues the loads, gh_accumulate does everything behind them.  The arithmetic and its order per pixel and
// pose are those of the one-piece version (-ffp-contract=off: no re-association across the split).
struct GhProj {
    float tx, ty, tz;   // the view point in the scene's frame of the last raycast
    float dx, dy;       // bilinear weights
    int idx;            // pixel index of the footprint's upper-left corner in the scene maps (0 when !ok)
    bool ok;            // the point projects inside the scene image
};

__device__ __forceinline__ GhProj gh_project(const GhArgs& a, const Mat4& approxInvPose, float px, float py, float depth, bool live) {
    GhProj r;
    mul_point(approxInvPose, px, py, depth, 1.0f, r.tx, r.ty, r.tz);
    float qx, qy, qz;
    mul_point(a.scenePose, r.tx, r.ty, r.tz, 1.0f, qx, qy, qz);
    const float u = a.scene_intr.x * qx / qz + a.scene_intr.z;
    const float v = a.scene_intr.y * qy / qz + a.scene_intr.w;
    r.ok = live && !(qz <= 0.0f) && ((u >= 0.0f) && (u <= a.sw - 2) && (v >= 0.0f) && (v <= a.sh - 2));
    // interpolateBilinear_withHoles (Utils/ITMPixelUtils.h:78-106): corner and weights
    const short ix = (short)floorf(r.ok ? u : 0.0f), iy = (short)floorf(r.ok ? v : 0.0f);
    r.dx = u - (float)ix; r.dy = v - (float)iy;
    r.idx = r.ok ? ix + iy * a.sw : 0;
    return r;
}

// the footprint: point | normal records of the four corners (pn[2i], pn[2i+1]), two rows of 64 contiguous bytes
struct GhFoot { float4 pa, na, pb, nb, pc, nc, pd, nd; };
__device__ __forceinline__ GhFoot gh_fetch(const GhArgs& a, const GhProj& r) {
    const float4* r0 = a.pn + 2 * r.idx;
    const float4* r1 = a.pn + 2 * (r.idx + a.sw);
    GhFoot f;
    f.pa = r0[0]; f.na = r0[1]; f.pb = r0[2]; f.nb = r0[3]; f.pc = r1[0]; f.nc = r1[1]; f.pd = r1[2]; f.nd = r1[3];
    return f;
}

template <int ITER>
__device__ __forceinline__ void gh_accumulate(const GhArgs& a, const GhProj& r, const GhFoot& ft, float depth, float& cnt, float& f,
                                              float* nabla, float* hess) {
    constexpr int NP = ITER == TRK_BOTH ? 6 : 3;
    if (!r.ok) return;
    const float4 cp = bilerp4(ft.pa, ft.pb, ft.pc, ft.pd, r.dx, r.dy);
    const float4 n = bilerp4(ft.na, ft.nb, ft.nc, ft.nd, r.dx, r.dy);
    if (cp.w < 0.0f) return;
    const float tx = r.tx, ty = r.ty, tz = r.tz;
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
    for (int rr = 0, counter = 0; rr < NP; rr++) {
        nabla[rr] += rho1 * w * A[rr];
#pragma unroll
        for (int c = 0; c <= rr; c++, counter++) hess[counter] += rho2 * w * A[rr] * A[c];
    }
}

#ifndef GPS_TRACK_EV_THREADS
#define GPS_TRACK_EV_THREADS 256
#endif
GPS_TUNABLE_REPORT(GPS_TRACK_EV_THREADS, 256);
constexpr int EV_THREADS = GPS_TRACK_EV_THREADS;
#ifndef GPS_TRACK_EV_MAX_WGS
#define GPS_TRACK_EV_MAX_WGS 256
#endif
GPS_TUNABLE_REPORT(GPS_TRACK_EV_MAX_WGS, 256);
constexpr int EV_MAX_WGS = GPS_TRACK_EV_MAX_WGS;           // rows of the partial table.  Measured on the 640x480 loop (ms per tracked frame): 128 rows
                                          // 0.750, 256 rows 0.725, 512 rows 0.735, 1280 rows (one pixel per thread) 0.818 -- the last
                                          // workgroup's fixed-order sum costs what the evaluation's extra parallelism buys.  (Two
                                          // pixels in flight per thread -- both depth loads, then both bilinear footprints -- changed
                                          // nothing either, 0.721: launch + tail + the host round trip dominate an iteration.)
constexpr int EV_ROW_GROUPS = EV_THREADS / 32;
// Poses one pre-launched evaluation can take: the LM loop's next pose + the poses the loop would evaluate AFTER it if it is
// rejected (they depend on the last good state only, so the host knows them before the evaluation returns -- see
// track_camera_impl).  Each pose is evaluated by its own group of EV_MAX_WGS workgroups with its own argument line, row table
// and result block.
#ifndef GPS_TRACK_EV_GROUPS
#define GPS_TRACK_EV_GROUPS 4
#endif
GPS_TUNABLE_REPORT(GPS_TRACK_EV_GROUPS, 4);
constexpr int EV_GROUPS = GPS_TRACK_EV_GROUPS;
// pixels of a thread whose memory round trips are taken together (eval_fused): 5 = a whole finest-level share per batch -- own pixel
// loop 6.1 vs 6.3 us there, but 5.7 vs 3.5 us at the coarse levels (one pixel per thread, four empty slots of unrolled code)
#ifndef GPS_TRACK_EV_BATCH
#define GPS_TRACK_EV_BATCH 1
#endif
GPS_TUNABLE_REPORT(GPS_TRACK_EV_BATCH, 1);
constexpr int MAILBOX_GROUP_WORDS = 64;   // a group's block of the host mailbox (words 0..31: its result row)

// The frame's valid-pixel count lives behind the 16-word block, spread over VC_SLOTS words per frame parity: the prepare
// kernel's workgroups add into slot (workgroup % VC_SLOTS) -- 1,200 atomics on ONE word cost the kernel 10 of its 16 us
// (tools/probe/prep_probe.sh: 16.3 us, 6.1 without the atomic, 15.3 without the point / normal interleave) -- and the summer's
// first 32 lanes add the slots up (and clear the other parity's) where one lane used to read the word.
constexpr int VC_BASE = 16, VC_SLOTS = 32;
static_assert(VC_SLOTS == GH_SLOTS, "one slot per lane of the result's half-wave");

// per-level constants of the evaluation (kernel arguments of track_eval_poll_kernel)
struct LevelTab { const float* depth; int vw, vh; float ix, iy, iz, iw; float space_thresh; int n_wgs; };

struct PrepArgs {
    gps_track_config cfg;
    LevelTab tab_vals[GPS_TRACK_MAX_LEVELS];  // (host-side staging of the per-level constants; the kernels do not read it)
    const float* depth0;                // full-resolution depth (s.depth)
    const int16_t* depth_mm;            // != NULL (tile kernel only): the frame's raw depth; the kernel converts it and
    float* depth0_out;                  //   writes s.depth itself (gps_tsdf_convert_depth folded into this launch)
    float* level[GPS_TRACK_MAX_LEVELS]; // [0] unused
    const float4* points;               // ICP maps of the last raycast
    const float4* normals;
    float4* pn;
    int W, H;
    uint32_t* sync;                     // [0] the evaluation's ticket, [VC_BASE + parity * VC_SLOTS + k] valid-pixel count slots of this frame (zero on entry)
    int parity;
};

// PrepareForEvaluation (ITMExtendedTracker.cpp:216-268) in one pass.  A work item is one
// 2^(L-1) x 2^(L-1) block of full-resolution pixels = one pixel of the coarsest level; it produces every pyramid pixel above
// that block locally (filterSubsampleWithHoles is hierarchical: a level-l pixel is the mean of the VALID pixels among its
// four level-(l-1) children, ITMLowLevelEngine_Shared.h:48-69), counts the valid full-resolution pixels and interleaves the
// ICP maps.
__global__ __launch_bounds__(256) void track_prepare_kernel(PrepArgs a) {
    GPS_FRAME_PRIO();
    const int L = a.cfg.n_levels;
    const int B = 1 << (L - 1);  // block edge in level-0 pixels
    const int bw = (a.W + B - 1) / B, bh = (a.H + B - 1) / B;
    const int stride = gridDim.x * blockDim.x;
    for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < bw * bh; item += stride) {
        const int by = item / bw, bx = item - by * bw;
        // bottom-up: level l reads the level below from memory THIS thread wrote (level 0: the input)
        for (int l = 1; l < L; l++) {
            const int e = B >> l;  // edge of the block at level l
            const float* src = l == 1 ? a.depth0 : a.level[l - 1];
            const int w_in = a.W >> (l - 1), lwl = a.W >> l, lhl = a.H >> l;  // repeated floor halving == a right shift
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
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        valid += a.depth0[i] > 0.0f ? 1 : 0;
        a.pn[2 * i] = a.points[i];
        a.pn[2 * i + 1] = a.normals[i];
    }
    for (int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o, 64);
    // zero on entry: the evaluation launches of the PREVIOUS frame cleared this slot (they read the other one)
    if ((threadIdx.x & 63) == 0 && valid) atomicAdd(&a.sync[VC_BASE + a.parity * VC_SLOTS + (blockIdx.x & (VC_SLOTS - 1))], (uint32_t)valid);
}

// The same for pyramids of up to 5 levels (coarsest pixel <= 16 x 16 full-resolution pixels; the default "rrbb" has 4): one
// workgroup per 16 x 16 tile, thread = pixel.  Level 0 goes through LDS once (row-coalesced loads), every level above is
// computed from the LDS copy of the level below by the first (16 >> l)^2 threads -- the first version gave each THREAD a whole
// coarsest-level block and walked it with strided loads (25 us; this one: see LABBOOK.md section 4).
__global__ __launch_bounds__(256) void track_prepare_tile_kernel(PrepArgs a) {
    GPS_FRAME_PRIO();
    __shared__ float lv[2][16 * 16];
    __shared__ int wv[4];
    const int L = a.cfg.n_levels;
    const int tiles_x = (a.W + 15) >> 4;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x = tx * 16 + lx, y = ty * 16 + ly;
    float d = 0.0f;
    const bool in = x < a.W && y < a.H;
    if (in) {
        const int i = x + y * a.W;
        if (a.depth_mm) {  // convertDepthAffineToFloat with the reference's 1/1000, 0 calibration (tsdf_fusion.hip: convert_depth_kernel)
            const int16_t mm = a.depth_mm[i];
            d = mm <= 0 ? -1.0f : (float)mm * (1.0f / 1000.0f) + 0.0f;
            a.depth0_out[i] = d;
        } else {
            d = a.depth0[i];
        }
        a.pn[2 * i] = a.points[i];
        a.pn[2 * i + 1] = a.normals[i];
    }
    lv[0][threadIdx.x] = d;
    int valid = (in && d > 0.0f) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o, 64);
    if ((threadIdx.x & 63) == 0) wv[threadIdx.x >> 6] = valid;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int v = (wv[0] + wv[1]) + (wv[2] + wv[3]);
        // zero on entry: the evaluation launches of the PREVIOUS frame cleared this slot (they read the other one)
        if (v) atomicAdd(&a.sync[VC_BASE + a.parity * VC_SLOTS + (blockIdx.x & (VC_SLOTS - 1))], (uint32_t)v);
    }
    // level l from level l - 1: filterSubsampleWithHoles (mean of the valid children, ITMLowLevelEngine_Shared.h:48-69)
    for (int l = 1; l < L; l++) {
        const int e = 16 >> l, e_in = 16 >> (l - 1);  // tile edge at level l / l - 1
        const float* src = lv[(l - 1) & 1];
        float* dst = lv[l & 1];
        if ((int)threadIdx.x < e * e) {
            const int px = threadIdx.x % e, py = threadIdx.x / e;
            float acc = 0.0f, good = 0.0f, v;
            v = src[(2 * px + 0) + (2 * py + 0) * e_in]; if (v > 0.0f) { acc += v; good++; }
            v = src[(2 * px + 1) + (2 * py + 0) * e_in]; if (v > 0.0f) { acc += v; good++; }
            v = src[(2 * px + 0) + (2 * py + 1) * e_in]; if (v > 0.0f) { acc += v; good++; }
            v = src[(2 * px + 1) + (2 * py + 1) * e_in]; if (v > 0.0f) { acc += v; good++; }
            if (good > 0) acc /= good;
            dst[px + py * e] = acc;
            const int gx = tx * e + px, gy = ty * e + py, lwl = a.W >> l, lhl = a.H >> l;
            if (gx < lwl && gy < lhl) a.level[l][gx + gy * lwl] = acc;
        }
        __syncthreads();
    }
}

// One LM iteration's evaluation + reduction.  Cross-workgroup traffic inside the launch avoids L2 flushes: the XCDs' L2s are
// not coherent with each other, and an agent-scope release / acquire pair means buffer_wbl2 + buffer_inv of the WHOLE L2 per
// workgroup (round 1 measured a fenced last-block-done ticket 20 % slower than a second launch, with the map stream's
// rasterizer sharing that L2).  The rows therefore travel as agent-scope RELAXED atomic stores / loads (sc1: write-through to /
// read from the memory-side coherence point), ordered by program order + s_waitcnt: rows stored, vmcnt(0) waits for their
// acknowledgement, THEN the ticket is taken; the workgroup that draws the last ticket reads the rows with sc1 loads and resets
// the ticket.  Nothing else is written back or invalidated.  Totals + the frame's valid-pixel count go to the pinned host
// mailbox, sequence number last.
// One evaluation by the workgroups [0, n_rows) of a launch (shared by the two kernels below).
//
// Row hand-over WITHOUT acknowledgements or a ticket (round 2, second step).  A row of the table is two 64-byte chunks, each
// 15 sums + the launch's sequence number in its last word (sum d lives in word d + d / 15).  A workgroup stores its row with
// one sc1 store instruction (two 64-byte write transactions) and is DONE: no wait for the acknowledgement, no ticket.
// Workgroup 0 is the summer: after its own row it reads all rows (sc1 loads) until both tags of every row carry this
// launch's number -- stale rows carry older ones, so nothing is ever reset -- and adds them in fixed order.  The totals go to
// the pinned mailbox in the same chunked format (the host waits for both tags), again without an acknowledgement wait.
// Against the ticket version this takes three memory-side round trips out of every iteration's dependent chain (row
// acknowledgement, ticket, mailbox acknowledgement); under the map stream's memory traffic each of them was ~2-3 us.
__device__ __forceinline__ int chunk_word(int d) { return d + d / 15; }
// words of the 16-word `sync` block besides [0] ticket ([1], [2]: unused since the valid-pixel counts moved behind the block): the pre-launched evaluation's profile
constexpr int SYNC_SPIN_TICKS = 8, SYNC_EVAL_TICKS = 9, SYNC_EVALS = 10, SYNC_SKIPPED = 11;
// where an evaluation's time goes, as workgroup 0 (the summer) sees it, for the finest level and for the coarser ones: number of
// evaluations, ticks from the argument line's arrival to the end of its own pixel loop, from there until every row of the table
// carries this launch's tag (the slowest workgroup + the hand-over through memory), from there to the mailbox store
constexpr int SYNC_PHASE_L0 = 1, SYNC_PHASE_COARSE = 5;   // words [1..4] and [5..7] + [12]
__host__ __device__ __forceinline__ int phase_word(int base, int k) { return base == SYNC_PHASE_COARSE && k == 3 ? 12 : base + k; }
// (2 s, not the 50 ms of rounds 3-4: a host thread that is frozen for one CPU-accounting period -- 100 ms, a container over its
// quota -- must find its launch still waiting when it comes back; the timeouts are for a host that never comes back)
constexpr long long ROW_TIMEOUT = 2000LL * 1000 * 100;  // wall_clock64 ticks (100 MHz): 2 s

// LDS of an evaluation workgroup (declared once per kernel: the fused body below is instantiated per kind and pose count)
struct EvLds {
    float red[EV_GROUPS][EV_THREADS / 64][GH_SLOTS];   // per pose: the waves' totals
    float group[EV_ROW_GROUPS][GH_SLOTS];              // the summer's row groups
    int all_ok;
};

// The summer of ONE pose's row table (every thread of its workgroup): re-reads the rows until all n_rows carry this launch's tag,
// adds them in fixed order, writes the totals + the frame's valid-pixel count to `result` and the host mailbox block `mailbox`.
// -> wall-clock stamp of "all rows seen" (0 when it gave up)
__device__ __forceinline__ long long sum_rows_and_answer(EvLds& lds, int n_rows, const uint32_t* __restrict__ partial,
                                                         uint32_t* __restrict__ sync, float* __restrict__ result,
                                                         volatile float* mailbox, int seq, int parity) {
    const int tid = threadIdx.x;
    // word k of rows r, r + 8, ... by thread (r, k); a round is ONE batch of loads
    const int k = tid & (GH_SLOTS - 1), r = tid >> 5;
    const int rows = n_rows;
    float s = 0.0f;
    const long long t0 = wall_clock64();
    for (;;) {
        bool ok = true;
        s = 0.0f;
        constexpr int INFLIGHT = 32;  // a 256-row table is 32 loads per thread: one memory-side round trip per round
        for (int rr = r; rr < rows; rr += INFLIGHT * EV_ROW_GROUPS) {
            uint32_t v[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) {
                const int row = rr + u * EV_ROW_GROUPS;
                v[u] = row < rows ? __hip_atomic_load(partial + (size_t)row * GH_SLOTS + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : (uint32_t)seq;
            }
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) {
                if ((k & 15) == 15) ok = ok && v[u] == (uint32_t)seq;  // this lane holds a tag word of the row
                else s += __uint_as_float(v[u]);                       // rows added in row order
            }
        }
        if (tid == 0) lds.all_ok = 1;
        __syncthreads();
        if (!ok) lds.all_ok = 0;
        __syncthreads();
        const int done = lds.all_ok;
        __syncthreads();
        if (done) break;
        if (wall_clock64() - t0 > ROW_TIMEOUT) return 0;  // a row never arrived: no result (the host's bounded wait reports it)
        __builtin_amdgcn_s_sleep(1);
    }
    const long long t_rows = wall_clock64();
    lds.group[r][k] = s;
    __syncthreads();
    if (tid < GH_SLOTS) {  // word tid of the result, chunked like a row: sums, then the valid-pixel count as payload 29
        const int d = tid - (tid >> 4);
        uint32_t wv = (uint32_t)seq;
        float t = 0.0f;
        // the frame's valid-pixel count: lane k fetches slot k, clears the next frame's slot k; five shuffles add them up
        uint32_t vc = __hip_atomic_load(&sync[VC_BASE + parity * VC_SLOTS + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sync[VC_BASE + (1 - parity) * VC_SLOTS + tid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) vc += (uint32_t)__shfl_xor((int)vc, o, 32);
        if ((tid & 15) != 15) {
            if (d < 29) {
#pragma unroll
                for (int g2 = 0; g2 < EV_ROW_GROUPS; g2++) t += lds.group[g2][tid];
                wv = __float_as_uint(t);
            } else {  // d == 29: the frame's valid-pixel count
                wv = vc;
            }
        }
        result[tid] = __uint_as_float(wv);  // (device copy, same layout: the host_mailbox == NULL path reads it back)
        if (mailbox) reinterpret_cast<volatile uint32_t*>(mailbox)[tid] = wv;  // one store instruction: two 64-byte chunks
    }
    __syncthreads();   // (lds.group is reused by a later pass of the same workgroup)
    return t_rows;
}

// K poses of ONE level and kind evaluated by the workgroups [0, n_rows) of a launch: a pixel's depth and back-projection are shared,
// every pose keeps its own sums (accumulated in the order a single-pose evaluation accumulates them: bit-equal), its own row
// table (`partial` + k tables), result block and mailbox block; pose k's table is summed by workgroup k.
template <int ITER, int K>
__device__ __forceinline__ void eval_fused(EvLds& lds, const GhArgs& a, const Mat4* poses, int bid, int n_rows,
                                           uint32_t* __restrict__ partial, uint32_t* __restrict__ sync,
                                           float* __restrict__ result, volatile float* mailbox, int seq, int parity,
                                           long long t_arrived = -1, int phase_base = 0, volatile uint32_t* host_rows = nullptr) {
    constexpr int NP = ITER == TRK_BOTH ? 6 : 3, NSQ = ITER == TRK_BOTH ? 21 : 6, NV = 2 + NP + NSQ, NQ = (NV + 3) / 4;
    // HOST-SUMMED ROWS (host_rows != NULL, round 5): every workgroup stores its row straight into the pinned host table and is done
    // -- no summer, no second hop through the memory-side coherence point; the host, which spins for the answer anyway, adds the
    // rows in the summer's order.  Workgroup 0's row carries the frame's valid-pixel count (payload 29): requested HERE, before
    // the pixel loop, so that its round trip is not in front of the row store.
    uint32_t vc = 0;
    if (host_rows && bid == 0 && threadIdx.x < GH_SLOTS) {
        vc = __hip_atomic_load(&sync[VC_BASE + parity * VC_SLOTS + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sync[VC_BASE + (1 - parity) * VC_SLOTS + threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static_assert(NV <= 29, "29 sums + the valid-pixel count fill the two 15-word chunks");
    static_assert(K >= 1 && K <= EV_GROUPS && K * GH_SLOTS <= EV_THREADS, "one thread per row word of every pose");
    float acc[K][4 * NQ];
#pragma unroll
    for (int p = 0; p < K; p++)
#pragma unroll
        for (int k = 0; k < 4 * NQ; k++) acc[p][k] = 0.0f;
    const int n = a.vw * a.vh;
    // A thread's pixels in batches of B with every memory round trip of a batch taken ONCE: B depth loads, then the B x K bilinear
    // footprints (8 loads of 16 bytes each; a footprint that does not exist reads pixel 0 and is dropped), then the sums in pixel
    // order.  The finest level of a 640x480 frame is 4.7 pixels per thread: one batch, two round trips -- the pixel-at-a-time
    // loop took two per pixel (10 us of the evaluation's 18).  One wave per SIMD runs here (256 workgroups of 4 waves on 256
    // compute units), so the ~200 registers of a batch cost no occupancy.
    constexpr int B = K == 1 ? GPS_TRACK_EV_BATCH : K == 2 ? 3 : 2;
    const int stride = n_rows * blockDim.x;
    for (int i0 = bid * blockDim.x + threadIdx.x; i0 < n; i0 += B * stride) {
        float depth[B];
#pragma unroll
        for (int b = 0; b < B; b++) { const int i = i0 + b * stride; depth[b] = i < n ? a.depth[i] : 0.0f; }
        GhProj pr[B][K];
#pragma unroll
        for (int b = 0; b < B; b++) {
            const int i = i0 + b * stride;
            const int y = i / a.vw, x = i - y * a.vw;
            const bool live = !(depth[b] <= 1e-8f);
            const float px = depth[b] * (((float)x - a.view_intr.z) / a.view_intr.x);
            const float py = depth[b] * (((float)y - a.view_intr.w) / a.view_intr.y);
#pragma unroll
            for (int p = 0; p < K; p++) pr[b][p] = gh_project(a, poses[p], px, py, depth[b], live);
        }
        GhFoot ft[B][K];
#pragma unroll
        for (int b = 0; b < B; b++)
#pragma unroll
            for (int p = 0; p < K; p++) ft[b][p] = gh_fetch(a, pr[b][p]);
#pragma unroll
        for (int b = 0; b < B; b++)
#pragma unroll
            for (int p = 0; p < K; p++) gh_accumulate<ITER>(a, pr[b][p], ft[b][p], depth[b], acc[p][0], acc[p][1], acc[p] + 2, acc[p] + 2 + NP);
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool prof = bid == 0 && t_arrived >= 0 && phase_base != 0;
    const long long t_loop = prof ? wall_clock64() : 0;
    // wave totals, four values per register (wave_reduce.hpp): 10 VALU ops per quad instead of 24 ds_bpermute round trips
    const int row_slot = ((lane >> 4) & 1) * 2 + (lane >> 5);  // reduce4 leaves (a, c, b, d) in rows 0..3
#pragma unroll
    for (int p = 0; p < K; p++)
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const float z = gps::reduce4(acc[p][4 * q], acc[p][4 * q + 1], acc[p][4 * q + 2], acc[p][4 * q + 3]);
            if ((lane & 15) == 15) lds.red[p][wave][4 * q + row_slot] = z;
        }
    __syncthreads();
    if (tid < K * GH_SLOTS) {  // word w of this workgroup's row of pose p: sums in words d + d / 15, the sequence number in words 15 and 31
        const int p = tid >> 5, w = tid & (GH_SLOTS - 1);
        const int d = w - (w >> 4);  // inverse of chunk_word for non-tag words
        uint32_t wv = (uint32_t)seq;
        if ((w & 15) != 15) {
            float t = 0.0f;
            if (d < NV) {
#pragma unroll
                for (int w2 = 0; w2 < EV_THREADS / 64; w2 += 2) t += lds.red[p][w2][d] + lds.red[p][w2 + 1][d];  // fixed order
            }
            wv = __float_as_uint(t);
        }
        if (host_rows) {
            if (bid == 0) {   // payload 29 (word 30) of row 0: the frame's valid-pixel count
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) vc += (uint32_t)__shfl_xor((int)vc, o, 32);
                if (w == 30) wv = vc;
            }
            host_rows[((size_t)p * EV_MAX_WGS + bid) * GH_SLOTS + w] = wv;   // one store instruction: two 64-byte chunks across PCIe
        } else {
            __hip_atomic_store(partial + ((size_t)p * EV_MAX_WGS + bid) * GH_SLOTS + w, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (host_rows) {
        if (tid == 0 && bid == 0 && t_arrived >= 0) {   // (profile: this workgroup's share; "until all rows" is the host's now)
            const long long t_end = wall_clock64();
            sync[SYNC_EVAL_TICKS] += (uint32_t)(t_end - t_arrived); sync[SYNC_EVALS] += 1u;
            if (prof) { sync[phase_word(phase_base, 0)] += 1u; sync[phase_word(phase_base, 1)] += (uint32_t)(t_loop - t_arrived);
                        sync[phase_word(phase_base, 3)] += (uint32_t)(t_end - t_loop); }
        }
        return;
    }
    if (bid >= K) return;
    // ---- workgroup p < K is the summer of pose p
    const long long t_rows = sum_rows_and_answer(lds, n_rows, partial + (size_t)bid * EV_MAX_WGS * GH_SLOTS, sync, result + bid * MAILBOX_GROUP_WORDS,
                                                 mailbox ? mailbox + bid * MAILBOX_GROUP_WORDS : nullptr, seq, parity);
    // profile words (gps_track_poll_profile / _phases): wall-clock ticks between the arrival of this launch's argument line and its
    // result leaving, and the number of evaluations.  Launches of a stream run one after the other: plain read-modify-write
    if (tid == 0 && bid == 0 && t_arrived >= 0 && t_rows) {
        const long long t_end = wall_clock64();
        sync[SYNC_EVAL_TICKS] += (uint32_t)(t_end - t_arrived); sync[SYNC_EVALS] += 1u;
        if (prof) {
            sync[phase_word(phase_base, 0)] += 1u;
            sync[phase_word(phase_base, 1)] += (uint32_t)(t_loop - t_arrived);
            sync[phase_word(phase_base, 2)] += (uint32_t)(t_rows - t_loop);
            sync[phase_word(phase_base, 3)] += (uint32_t)(t_end - t_rows);
        }
    }
}

// (1) one launch per LM iteration with its arguments as kernel arguments: the host_mailbox == NULL path
template <int ITER>
__global__ __launch_bounds__(EV_THREADS) void track_eval_kernel(GhArgs a, uint32_t* __restrict__ partial, uint32_t* __restrict__ sync,
                                                              float* __restrict__ result, volatile float* mailbox, int seq, int parity) {
    GPS_FRAME_PRIO();
    __shared__ EvLds lds;
    eval_fused<ITER, 1>(lds, a, &a.approxInvPose, (int)blockIdx.x, (int)gridDim.x, partial, sync, result, mailbox, seq, parity);
}

// (2) PRE-LAUNCHED evaluation (mailbox path).  What an LM iteration evaluates -- level, kind, pose -- is decided by the host
// from the previous evaluation's sums, so with (1) every iteration pays hipLaunchKernel + the dispatch latency (~8 us) AFTER
// the host has decided.  This kernel is enqueued BEFORE that: its workgroups start as soon as the previous evaluation has
// drained, read one 64-byte argument line from pinned host memory until it carries this launch's sequence number, and then
// run the same evaluation.  The host's decision travels as one cache-line write instead of a launch:
//     word 0 sequence number | 1 command, kind, level | 2..13 approxInvPose (3 rows x 4 columns) | 14 0 | 15 xor of 0..14
// The line is written payload first, sequence number last, and read by ONE load instruction of 16 lanes (one 64-byte
// request); the xor word rejects a torn line anyway.  ARG_SKIP retires a launch the LM loop did not need (the iteration
// count is data dependent); a launch whose line never arrives gives up after ARG_TIMEOUT (host error paths), so a stream
// can never be wedged by it.  Per-level constants are kernel arguments.
// Measured after this: keeping ONE launch resident for all evaluations of a frame (workgroups wait for the next line instead
// of exiting; late workgroups join at the current evaluation) -- no faster (0.571 vs 0.560 ms per tracked frame, overlap
// schedule 934 vs 960 frames/s): with the next launch already queued, launch-to-launch overhead is off the critical path;
// an iteration is the host <-> device loop itself (sums -> PCIe -> solve -> line -> PCIe -> relay -> rows -> sums).
//
// Round 5, measured and dropped: THE REJECTION CHAIN IN THE SAME WORKGROUPS -- consecutive argument lines of one level and kind
// evaluated by the same 256 workgroups in one pass over the pixels (depth load and back-projection shared, every pose its own sums,
// row table and mailbox block; bit-equal poses with 0..3 poses riding along).  The evaluation is latency-bound, not bandwidth-
// bound: a second pose in the same thread adds its own dependent footprint round trip (own pixel loop of the finest level 10.0 ->
// 14.1 us, of the coarser levels 4.7 -> 7.5 us; with all footprints of a batch of pixels and poses in flight at once 13.3 / 9.3 us
// at 256 registers), whereas a second GROUP of workgroups runs beside the first on otherwise idle compute units for free.  The
// groups stay (tools/probe/track_phases.py prints the phases these numbers are).
struct PollArgs {
    const float4* pn; int sw, sh;
    float4 scene_intr;
    Mat4 scenePose;
    float tukey_cutoff, vf_min, vf_max;
    int use_weights, frames_to_skip, frames_to_weight;
    LevelTab tab[GPS_TRACK_MAX_LEVELS];  // per-level constants (kernel arguments: selected with static indices, no memory round trip)
    const uint32_t* arg_line;  // pinned host memory, 64-byte aligned
    uint32_t* dev_line;        // device copy of the line (relayed by workgroup 0)
    const uint32_t* bar_line;  // != NULL: the host writes the lines into THIS block of fine-grained device memory through the
                               // BAR (gps_track_state.dev_arg_line) and every workgroup polls them here: no PCIe read, no relay
    volatile uint32_t* host_rows;  // != NULL: the row tables live in the pinned host mailbox and the HOST adds the rows (eval_fused)
};
// Round 5, measured and dropped: WARMING UP WHILE THE LAUNCH WAITS FOR ITS LINE.  The host knows, when it enqueues launch n + 1, the
// level and pose of evaluation n; while the workgroups waited (~3 us) every thread fetched its first pixel's depth at that level and
// waves 1..3 gathered the first pixels' footprints under that pose and dropped them.  Own pixel loop 7.3 -> 6.2 us (finest level),
// 4.0 -> 3.6 us (coarser levels) -- and nothing end to end (tracked frame 0.454 vs 0.457 ms, sequential 1,038 vs 1,032 frames/s in
// a 3 + 3 A/B): the loop is not two bare memory latencies, and the host's share of an iteration (rows, solve, line) hides the rest.
constexpr uint32_t ARG_RUN = 1, ARG_SKIP = 2;
constexpr long long ARG_TIMEOUT = 2000LL * 1000 * 100;  // wall_clock64 ticks (100 MHz): 2 s (see ROW_TIMEOUT)

__global__ __launch_bounds__(EV_THREADS) void track_eval_poll_kernel(PollArgs pa, uint32_t* __restrict__ partial,
                                                                   uint32_t* __restrict__ sync, float* __restrict__ result,
                                                                   volatile float* mailbox, int seq, int parity) {
    __shared__ uint32_t line[16];
    __shared__ long long t_arrived;
    __shared__ EvLds lds;
    // group 0 = the LM loop's evaluation; groups 1 .. EV_GROUPS - 1 (BAR line only: the grid has them only then) = the poses the
    // loop would evaluate next if that one is rejected.  A group has its own argument line (64 bytes apart), rows and result.
    const int grp = (int)blockIdx.x / EV_MAX_WGS, bid = (int)blockIdx.x - grp * EV_MAX_WGS;
    if (threadIdx.x < 16) {
        // Workgroup 0 polls the host line and relays it through a device-memory copy the other workgroups poll: with all 256
        // workgroups reading the host line across PCIe, the CPU's store waited ~20 us for ownership of its own cache line
        // (measured: 33 us per iteration instead of ~20; one poller: 1.7 us host -> GPU -> host round trip).
        const bool direct = pa.bar_line != nullptr;   // (wave-uniform: a kernel argument)
        const bool relay = !direct && blockIdx.x == 0;
        const uint32_t* src = direct ? pa.bar_line + 16 * grp : relay ? pa.arg_line : pa.dev_line;
        const long long t0 = wall_clock64();
        uint32_t v;
        for (;;) {
            v = (relay || direct) ? __hip_atomic_load(src + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                  : __hip_atomic_load(src + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t x = threadIdx.x < 15 ? v : 0u;  // xor of words 0..14 over the 16 active lanes (row 0 of the wave)
            x ^= __shfl_xor(x, 1, 16); x ^= __shfl_xor(x, 2, 16); x ^= __shfl_xor(x, 4, 16); x ^= __shfl_xor(x, 8, 16);
            const uint32_t w0 = __shfl(v, 0, 16), w15 = __shfl(v, 15, 16);
            if (w0 == (uint32_t)seq && w15 == x) break;
            // A valid line of a LATER launch: this launch's line has come and gone.  The host publishes launch seq + 1 only after
            // every group of launch seq that was told to RUN has answered (every workgroup with pixels has delivered its row, so
            // this one has none) and group 0 has acknowledged a retirement (SKIP, workgroup 0) -- either way there is nothing
            // left for this workgroup to do.  Only the direct (BAR) line can show this: a workgroup that became resident late,
            // beside another stream's kernels, after its group had answered (the relayed copy is rewritten by the NEXT launch,
            // which cannot start before this one has drained).  Without it such a workgroup sat out ARG_TIMEOUT with the frame
            // stream behind it.
            const bool superseded = w15 == x && (int32_t)(w0 - (uint32_t)seq) > 0 && w0 < 0x40000000u;
            if (superseded || wall_clock64() - t0 > ARG_TIMEOUT) {  // give up: behave like ARG_SKIP (and relay that)
                v = threadIdx.x == 0 ? (uint32_t)seq : threadIdx.x == 1 ? ARG_SKIP : 0u;
                uint32_t y = threadIdx.x < 15 ? v : 0u;
                y ^= __shfl_xor(y, 1, 16); y ^= __shfl_xor(y, 2, 16); y ^= __shfl_xor(y, 4, 16); y ^= __shfl_xor(y, 8, 16);
                if (threadIdx.x == 15) v = y;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        line[threadIdx.x] = v;
        if (threadIdx.x == 0) {
            // how long this launch sat on the GPU waiting for the host's decision (workgroup 0: the one that polls the host line)
            const long long now = wall_clock64();
            t_arrived = now;
            if (blockIdx.x == 0) sync[SYNC_SPIN_TICKS] += (uint32_t)(now - t0);
        }
        if (relay) {  // payload, wait for the acknowledgement (sc1 stores), then the sequence number
            // (round 3: the whole line as ONE 16-lane store instruction, torn lines left to the xor word -- no faster in an A/B:
            // 1,175 vs 1,187 frames/s, within noise; the ordered form stays)
            if (threadIdx.x != 0) __hip_atomic_store(pa.dev_line + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (threadIdx.x == 0) __hip_atomic_store(pa.dev_line, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    GPS_FRAME_PRIO();   // (only now: a launch that is still waiting for its line should not crowd out anybody)
    const uint32_t ctl = __builtin_amdgcn_readfirstlane(line[1]);
    if ((ctl & 0xFF) != ARG_RUN) {
        // retired: tell the host its line has been consumed -- the host must not write the next launch's arguments into a line
        // while this launch, queued but not yet running, still has to find its own (it would then wait out ARG_TIMEOUT with the
        // whole frame stream behind it).  Group 0 answers for the launch; the other groups' late workgroups are covered by
        // `superseded` above (their lines are rewritten together with group 0's).
        if (blockIdx.x == 0 && threadIdx.x == 0 && mailbox)
            __hip_atomic_store(reinterpret_cast<uint32_t*>(const_cast<float*>(mailbox)) + 32, (uint32_t)seq, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        if (blockIdx.x == 0 && threadIdx.x == 0) sync[SYNC_SKIPPED] += 1u;
        return;
    }
    const int kind = (int)((ctl >> 8) & 0xFF), level = (int)((ctl >> 16) & 0xFF);
    LevelTab lt = pa.tab[0];
#pragma unroll
    for (int l = 1; l < GPS_TRACK_MAX_LEVELS; l++)  // (a dynamically indexed kernel-argument array would be copied to scratch)
        if (level == l) lt = pa.tab[l];
    if (bid >= lt.n_wgs) return;
    GhArgs a;
    a.depth = lt.depth; a.vw = lt.vw; a.vh = lt.vh;
    a.view_intr = make_float4(lt.ix, lt.iy, lt.iz, lt.iw);
    a.pn = pa.pn; a.sw = pa.sw; a.sh = pa.sh; a.scene_intr = pa.scene_intr;
    // approxInvPose, ORUtils layout m[col * 4 + row]: rows 0..2 travel, row 3 is (0, 0, 0, 1)
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int r = 0; r < 3; r++)
            a.approxInvPose.m[c * 4 + r] = __uint_as_float(__builtin_amdgcn_readfirstlane(line[2 + c * 3 + r]));
        a.approxInvPose.m[c * 4 + 3] = c == 3 ? 1.0f : 0.0f;
    }
    a.scenePose = pa.scenePose;
    a.space_thresh = lt.space_thresh; a.tukey_cutoff = pa.tukey_cutoff; a.vf_min = pa.vf_min; a.vf_max = pa.vf_max;
    a.use_weights = pa.use_weights; a.frames_to_skip = pa.frames_to_skip; a.frames_to_weight = pa.frames_to_weight;
    const long long ta = blockIdx.x == 0 ? t_arrived : -1;
    uint32_t* const rows = partial + (size_t)grp * EV_MAX_WGS * GH_SLOTS;
    float* const res = result + grp * MAILBOX_GROUP_WORDS;
    volatile float* const mb = mailbox ? mailbox + grp * MAILBOX_GROUP_WORDS : nullptr;
    const int pb = level == 0 ? SYNC_PHASE_L0 : SYNC_PHASE_COARSE;   // (group 0's workgroup 0 keeps the profile: ta >= 0 only there)
    volatile uint32_t* const hr = pa.host_rows ? pa.host_rows + (size_t)grp * EV_MAX_WGS * GH_SLOTS : nullptr;
    if (kind == TRK_ROTATION) eval_fused<TRK_ROTATION, 1>(lds, a, &a.approxInvPose, bid, lt.n_wgs, rows, sync, res, mb, seq, parity, ta, pb, hr);
    else if (kind == TRK_TRANSLATION) eval_fused<TRK_TRANSLATION, 1>(lds, a, &a.approxInvPose, bid, lt.n_wgs, rows, sync, res, mb, seq, parity, ta, pb, hr);
    else eval_fused<TRK_BOTH, 1>(lds, a, &a.approxInvPose, bid, lt.n_wgs, rows, sync, res, mb, seq, parity, ta, pb, hr);
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

// The host half of ITMExtendedTracker::TrackCamera (ITMExtendedTracker.cpp:470-665; oracle/tsdf_oracle.c: orc_track_camera) as
// a state machine: request() = what the loop evaluates next, apply() = what it does with an evaluation's sums (accept / reject,
// damping, Cholesky solve, SE3 update, convergence test, level change).  apply(nullptr) is the REJECT branch, which reads
// nothing of the evaluation it rejects: run on a copy of the state it tells, before an evaluation has returned, which pose the
// loop evaluates after it should it be rejected (and after that one, and ...) -- those poses ride along with the evaluation.
struct LmRequest {
    bool valid = false;
    int level = 0, kind = 0;
    float pose[16] = {0};   // approxInvPose
    bool same(const LmRequest& o) const { return valid == o.valid && level == o.level && kind == o.kind && memcmp(pose, o.pose, 64) == 0; }
};

struct LmLoop {
    const gps_track_config* c = nullptr;
    bool active = false, bad_pose = false;
    int level = 0, iter = 0;
    float M[16], invM[16], approxInvPose[16], lastGoodM[16], lastGoodInvM[16];
    float f_old = 0, lambda = 0;
    float hessian_good[36] = {0}, nabla_good[6] = {0}, hessian_depth_good[36] = {0}, f_depth_good = 0;
    int nvalid_depth_good = 0, last_type = TRK_NONE, n_valid_bits = 0;
    int evals[GPS_TRACK_MAX_LEVELS] = {0};   // evaluations consumed per level

    void start(const gps_track_config* cfg, const float* M0, const float* invM0) {
        c = cfg;
        memcpy(M, M0, 64); memcpy(invM, invM0, 64);
        level = c->n_levels;
        next_level();
    }
    void next_level() {
        active = false;
        for (level--; level >= 0; level--) {
            const int it = c->iter_type[level];
            if (it == TRK_NONE) continue;
            last_type = it;
            memcpy(approxInvPose, invM, 64);
            memcpy(lastGoodM, M, 64); memcpy(lastGoodInvM, invM, 64);
            f_old = 3.402823466e+38f; lambda = 1.0f;
            iter = 0;
            if (c->n_iter[level] > 0) { active = true; return; }
        }
    }
    LmRequest request() const {
        LmRequest r;
        if (!active) return r;
        r.valid = true; r.level = level; r.kind = c->iter_type[level];
        memcpy(r.pose, approxInvPose, 64);
        return r;
    }
    // host = the 30 payload words of an evaluation's result (eval_body), or nullptr = the evaluation is taken as rejected.
    // Returns whether the evaluation was rejected.
    bool apply(const float* host) {
        const int it = c->iter_type[level], noPara = it == TRK_BOTH ? 6 : 3;
        bool reject = true;
        float hessian_depth[36] = {0}, nabla_depth[6] = {0}, f_depth = 0.0f;
        int nvalid = 0;
        if (host) {
            n_valid_bits = float_bits(host[29]);
            nvalid = (int)host[0];
            f_depth = host[1];
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
            evals[level] += 1;
            reject = (nvalid <= 0) || (f_depth >= f_old);
        }
        if (reject) {
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
        if (!set_invM_coerce(approxInvPose, M, invM)) { bad_pose = true; active = false; return reject; }
        memcpy(approxInvPose, invM, 64);
        bool converged = true;
        for (int i = 0; i < 6; i++)
            if (fabs(step[i]) > c->term_thresh) { converged = false; break; }
        iter++;
        if (converged || iter >= c->n_iter[level]) next_level();
        return reject;
    }
};

struct Scratch {
    float* level[GPS_TRACK_MAX_LEVELS];  // [0] unused (= s.depth)
    uint32_t *partial, *sync;
    float* result;
    float4* pn;
    uint32_t* dev_line;
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
    char* p = take((size_t)EV_GROUPS * EV_MAX_WGS * GH_SLOTS * sizeof(uint32_t)); if (w) w->partial = (uint32_t*)p;   // a row table per group
    p = take((size_t)EV_GROUPS * MAILBOX_GROUP_WORDS * sizeof(float)); if (w) w->result = (float*)p;
    p = take((VC_BASE + 2 * VC_SLOTS) * sizeof(uint32_t)); if (w) w->sync = (uint32_t*)p;   // 16 control words + the valid-count slots of both parities
    p = take(64); if (w) w->dev_line = (uint32_t*)p;
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
    void* mailbox = ts->host_mailbox;  // the mailbox and the sequence counter belong to the state's owner / the library
    void* line = ts->dev_arg_line;
    const int32_t seq = ts->mail_seq, mailbox_bytes = ts->mailbox_bytes;
    memset(ts, 0, sizeof(*ts));
    ts->host_mailbox = mailbox;
    ts->mailbox_bytes = mailbox_bytes;
    ts->dev_arg_line = line;
    ts->mail_seq = seq;
    for (int i = 0; i < 16; i += 5) ts->pose_M[i] = ts->pose_invM[i] = ts->pose_pc_M[i] = 1.0f;
    ts->age_point_cloud = -1;
    return GPS_OK;
}

// Probe of one host store + load through the BAR mapping under a fault handler.  Process-wide signal dispositions are involved:
// the probe is serialised (one at a time), the handler only takes a fault of the PROBING thread at an address inside the
// probed 4 KB block and otherwise hands the signal to whoever was installed before (Python's faulthandler, torch's crash
// handlers, the default action).
static std::mutex g_probe_mu;
static sigjmp_buf g_probe_jmp;
static volatile uintptr_t g_probe_lo = 0;
static pthread_t g_probe_thread;
static struct sigaction g_probe_old_segv, g_probe_old_bus;
static void probe_fault(int sig, siginfo_t* info, void* ctx) {
    const uintptr_t a = info ? reinterpret_cast<uintptr_t>(info->si_addr) : 0;
    if (g_probe_lo && pthread_equal(pthread_self(), g_probe_thread) && a >= g_probe_lo && a < g_probe_lo + 4096) siglongjmp(g_probe_jmp, 1);
    const struct sigaction& old = sig == SIGBUS ? g_probe_old_bus : g_probe_old_segv;
    if (old.sa_flags & SA_SIGINFO) {
        if (old.sa_sigaction) { old.sa_sigaction(sig, info, ctx); return; }
    } else if (old.sa_handler != SIG_DFL && old.sa_handler != SIG_IGN) {
        old.sa_handler(sig);
        return;
    }
    signal(sig, SIG_DFL);   // not ours and nobody else's: the default action on return (the faulting instruction re-executes)
}
static bool host_can_write(void* p) {
    std::lock_guard<std::mutex> lock(g_probe_mu);
    struct sigaction sa = {};
    sa.sa_sigaction = probe_fault;
    sa.sa_flags = SA_SIGINFO;
    sigemptyset(&sa.sa_mask);
    g_probe_thread = pthread_self();
    if (sigaction(SIGSEGV, &sa, &g_probe_old_segv) != 0) return false;
    if (sigaction(SIGBUS, &sa, &g_probe_old_bus) != 0) { sigaction(SIGSEGV, &g_probe_old_segv, nullptr); return false; }
    bool ok = false;
    g_probe_lo = reinterpret_cast<uintptr_t>(p);
    if (sigsetjmp(g_probe_jmp, 1) == 0) {
        volatile uint64_t* q = static_cast<volatile uint64_t*>(p);
        q[7] = 0x5A5A5A5A5A5A5A5Aull;
        host_store_fence();
        ok = q[7] == 0x5A5A5A5A5A5A5A5Aull;
        q[7] = 0;
        host_store_fence();
    }
    g_probe_lo = 0;
    sigaction(SIGSEGV, &g_probe_old_segv, nullptr);
    sigaction(SIGBUS, &g_probe_old_bus, nullptr);
    return ok;
}

int gps_track_arg_line_alloc(void** line) {
    if (!line) return GPS_ERR_ARG;
    *line = nullptr;
    int dev = 0, large_bar = 0;
    if (hipGetDevice(&dev) != hipSuccess) return GPS_ERR_LAUNCH;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess || !large_bar) {
        (void)hipGetLastError();
        return GPS_OK;  // device memory is not host-visible: the pinned line stays
    }
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return GPS_OK; }
    if (hipMemset(p, 0, 4096) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return GPS_ERR_LAUNCH; }
    // The attribute says the aperture exists, not that THIS process may store through it (a restricted container can map the
    // device without it): probe one store + load under a fault handler, once, here -- never in the tracking loop.
    if (!host_can_write(p)) { (void)hipFree(p); return GPS_OK; }
    *line = p;
    return GPS_OK;
}

int gps_track_arg_line_free(void* line) {
    if (!line) return GPS_OK;
    return hipFree(line) == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_track_poll_profile(const void* scratch, int width, int height, uint32_t out[4], gps_stream stream) {
    if (!scratch || !out || width <= 0 || height <= 0) return GPS_ERR_ARG;
    Scratch w;
    carve(&w, (char*)const_cast<void*>(scratch), width, height);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(out, w.sync + SYNC_SPIN_TICKS, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess) GPS_FAIL_LAUNCH();
    return hipStreamSynchronize(st) == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

int gps_track_poll_phases(const void* scratch, int width, int height, uint32_t out[8], gps_stream stream) {
    if (!scratch || !out || width <= 0 || height <= 0) return GPS_ERR_ARG;
    Scratch w;
    carve(&w, (char*)const_cast<void*>(scratch), width, height);
    hipStream_t st = (hipStream_t)stream;
    uint32_t words[16];
    if (hipMemcpyAsync(words, w.sync, sizeof(words), hipMemcpyDeviceToHost, st) != hipSuccess) GPS_FAIL_LAUNCH();
    if (hipStreamSynchronize(st) != hipSuccess) return GPS_ERR_LAUNCH;
    for (int k = 0; k < 4; k++) { out[k] = words[phase_word(SYNC_PHASE_L0, k)]; out[4 + k] = words[phase_word(SYNC_PHASE_COARSE, k)]; }
    return GPS_OK;
}

int64_t gps_track_scratch_bytes(int width, int height) {
    if (width <= 0 || height <= 0) return GPS_ERR_ARG;
    return (int64_t)carve(nullptr, nullptr, width, height);
}

static int track_camera_impl(const gps_tsdf_state* sp, const gps_track_config* c, gps_track_state* ts, void* scratch,
                             int64_t scratch_bytes, gps_stream stream, const int16_t* depth_mm);
static inline double wall_ms() {
    timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

int gps_tsdf_track_camera(const gps_tsdf_state* sp, const gps_track_config* c, gps_track_state* ts, void* scratch,
                          int64_t scratch_bytes, gps_stream stream) {
    return track_camera_impl(sp, c, ts, scratch, scratch_bytes, stream, nullptr);
}

// depth_mm != NULL: s.depth has NOT been converted yet; the prepare launch does it (tile kernel) or it is converted first
static int track_camera_impl(const gps_tsdf_state* sp, const gps_track_config* c, gps_track_state* ts, void* scratch,
                             int64_t scratch_bytes, gps_stream stream, const int16_t* depth_mm) {
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
    PrepArgs pa;
    pa.cfg = *c;
    pa.depth0 = s.depth;
    pa.depth_mm = nullptr; pa.depth0_out = s.depth;
    if (depth_mm) {
        if (c->n_levels <= 5) pa.depth_mm = depth_mm;
        else { int r0 = gps_tsdf_convert_depth(sp, depth_mm, stream); if (r0 != GPS_OK) return r0; }
    }
    pa.level[0] = nullptr;
    for (int l = 1; l < GPS_TRACK_MAX_LEVELS; l++) pa.level[l] = w.level[l];
    for (int l = 1; l < c->n_levels; l++) {
        lw[l] = lw[l - 1] / 2; lh[l] = lh[l - 1] / 2;
        dl[l] = w.level[l];
        for (int k = 0; k < 4; k++) lintr[l][k] = lintr[l - 1][k] * 0.5f;
    }
    pa.points = reinterpret_cast<const float4*>(s.icp_points);
    pa.normals = reinterpret_cast<const float4*>(s.icp_normals);
    pa.pn = w.pn; pa.W = W; pa.H = H; pa.sync = w.sync;
    // ts->scratch_epoch: 0 = the scratch words have to be zeroed (first call on this state / the last call failed), else
    // 1 + parity of the valid-count slot this frame uses.  The ticket resets itself and every evaluation launch clears the
    // other frame's count slot, so the steady state needs no memset launch in front of the frame.
    int parity = 0;
    if (ts->scratch_epoch == 0) {
        if (hipMemsetAsync(w.sync, 0, (VC_BASE + 2 * VC_SLOTS) * sizeof(uint32_t), st) != hipSuccess) GPS_FAIL_LAUNCH();
        // row tags: sequence numbers are >= 1, so a zeroed table can never look like a delivered row
        if (hipMemsetAsync(w.partial, 0, (size_t)EV_GROUPS * EV_MAX_WGS * GH_SLOTS * sizeof(uint32_t), st) != hipSuccess) GPS_FAIL_LAUNCH();
    } else {
        parity = ts->scratch_epoch - 1;
    }
    ts->scratch_epoch = 0;  // restored on success
    pa.parity = parity;
    for (int l = 0; l < GPS_TRACK_MAX_LEVELS; l++) pa.tab_vals[l] = LevelTab{};
    for (int l = 0; l < c->n_levels; l++)
        pa.tab_vals[l] = LevelTab{dl[l], lw[l], lh[l], lintr[l][0], lintr[l][1], lintr[l][2], lintr[l][3], c->space_thresh[l],
                                  min(EV_MAX_WGS, gps_div_up(lw[l] * lh[l], EV_THREADS))};
    if (c->n_levels <= 5) track_prepare_tile_kernel<<<((W + 15) >> 4) * ((H + 15) >> 4), 256, 0, st>>>(pa);
    else track_prepare_kernel<<<gps_div_up((int64_t)W * H, 4 * 256), 256, 0, st>>>(pa);
    GPS_LAUNCH_CHECK();

    // Mailbox path: evaluations are PRE-LAUNCHED (track_eval_poll_kernel); the host's per-iteration decision is one
    // cache-line write into the argument line (words 48..63 of the mailbox block).
    volatile float* const mailbox = reinterpret_cast<volatile float*>(ts->host_mailbox);
    volatile uint32_t* const arg_line = mailbox ? reinterpret_cast<volatile uint32_t*>(ts->host_mailbox) + 48 : nullptr;
    GPS_REQUIRE(!mailbox || (reinterpret_cast<uintptr_t>(ts->host_mailbox) & 63) == 0);
    PollArgs pl = {};
    if (mailbox) {
        pl.pn = w.pn; pl.sw = W; pl.sh = H;
        pl.scene_intr = make_float4(lintr[0][0], lintr[0][1], lintr[0][2], lintr[0][3]);
        pl.scenePose = load_mat(ts->pose_pc_M);
        pl.tukey_cutoff = c->tukey_cutoff; pl.vf_min = s.view_frustum_min; pl.vf_max = s.view_frustum_max;
        pl.use_weights = ts->frames_processed >= 100; pl.frames_to_skip = c->frames_to_skip; pl.frames_to_weight = c->frames_to_weight;
        for (int l = 0; l < GPS_TRACK_MAX_LEVELS; l++) pl.tab[l] = pa.tab_vals[l];
        pl.arg_line = const_cast<const uint32_t*>(arg_line);
        pl.dev_line = w.dev_line;
        pl.bar_line = reinterpret_cast<const uint32_t*>(ts->dev_arg_line);
    }
    GPS_REQUIRE((reinterpret_cast<uintptr_t>(ts->dev_arg_line) & 63) == 0);
    volatile uint32_t* const bar_line = mailbox ? reinterpret_cast<volatile uint32_t*>(ts->dev_arg_line) : nullptr;
    // (when the sequence wraps, the line of the last launch before the wrap -- a LARGER number than everything that follows -- is
    // wiped first: a late workgroup must not mistake it for a later launch's line, see `superseded` in the kernel)
    auto next_seq = [&]() {
        if (ts->mail_seq >= 0x3FFFFFFF) {
            ts->mail_seq = 0;
            if (mailbox) mailbox[32] = 0.0f;   // (the retirement word compares like the line: `retired` below)
            if (bar_line) {
                for (int k = 0; k < 8 * EV_GROUPS; k++) reinterpret_cast<volatile uint64_t*>(bar_line)[k] = 0;
                host_store_fence();
            }
        }
        return (int)(++ts->mail_seq);
    };
    // groups of an evaluation launch: the loop's pose + the poses that ride along (BAR lines and a mailbox block per group)
    // mailbox layout (gps_track_state.mailbox_bytes): G answer blocks of 256 bytes -- and, when there is room for them, G row tables of
    // EV_MAX_WGS x 128 bytes behind the blocks: the workgroups then store their rows THERE and this thread adds them (host_rows)
    constexpr int BLOCK_BYTES = MAILBOX_GROUP_WORDS * 4, ROWS_BYTES = EV_MAX_WGS * GH_SLOTS * 4;
    static_assert(BLOCK_BYTES == GPS_TRACK_MAILBOX_BLOCK_BYTES && ROWS_BYTES == GPS_TRACK_MAILBOX_ROWS_BYTES, "include/gps_slam_hip.h");
    const int groups_with_rows = ts->mailbox_bytes / (BLOCK_BYTES + ROWS_BYTES);
    // (host-summed rows reproduce the device summer's bits only with IEEE float adds on this thread: with flush-to-zero or
    // denormals-are-zero set in MXCSR -- e.g. a host built with -ffast-math that set them process-wide -- the device summer is used;
    // round-5 advisor finding.  The 64-byte row chunks leave the device as single store instructions and are tagged in their last
    // word; tests/test_tsdf_gpu.py keeps the host-summed == device-summed equality test in the default set)
    const bool host_sums = mailbox && groups_with_rows >= 1 && !host_flushes_denormals();
    const int mailbox_groups = host_sums ? groups_with_rows : ts->mailbox_bytes >= 2 * BLOCK_BYTES ? ts->mailbox_bytes / BLOCK_BYTES : 1;
    const int n_groups = (mailbox && bar_line) ? max(1, min(EV_GROUPS, mailbox_groups)) : 1;
    volatile uint32_t* const host_rows = host_sums ? reinterpret_cast<volatile uint32_t*>(ts->host_mailbox) + (size_t)min(EV_GROUPS, mailbox_groups) * MAILBOX_GROUP_WORDS : nullptr;
    pl.host_rows = host_rows;
    // The answer of group g to launch `seq`: the summer's block, or -- host_rows -- the rows of its table added in the summer's order
    // (thread (r, k) of the summer adds word k of rows r, r + 8, ... in increasing order, then the eight partial sums in order:
    // eval_fused / sum_rows_and_answer; IEEE additions only, so the bits are the device's).  `next` remembers the first row not yet
    // seen: rows are stable once tagged (nothing rewrites them before the next launch's lines are published).
    // Rows are added AS THEY ARRIVE (in row order: the scan below only ever moves forward), so that the last row to land costs one
    // row's additions, not a pass over the table.
    int rows_next[EV_GROUPS] = {0};
    float rows_part[EV_GROUPS][EV_ROW_GROUPS][GH_SLOTS];
    auto rows_reset = [&](int g) {
        rows_next[g] = 0;
        for (int q = 0; q < EV_ROW_GROUPS; q++)
            for (int k = 0; k < GH_SLOTS; k++) rows_part[g][q][k] = 0.0f;
    };
    auto answered = [&](int g, int seq, int n_rows) -> bool {
        if (!host_rows) {
            volatile float* const mb = mailbox + g * MAILBOX_GROUP_WORDS;
            return float_bits(mb[15]) == seq && float_bits(mb[31]) == seq;
        }
        volatile uint32_t* const T = host_rows + (size_t)g * EV_MAX_WGS * GH_SLOTS;
        int& r = rows_next[g];
        while (r < n_rows && (int)T[r * GH_SLOTS + 15] == seq && (int)T[r * GH_SLOTS + 31] == seq) {
            float* dst = rows_part[g][r % EV_ROW_GROUPS];
            for (int k = 0; k < GH_SLOTS; k++) {
                if ((k & 15) == 15) continue;   // the row's tags
                const uint32_t v = T[r * GH_SLOTS + k]; float f; memcpy(&f, &v, 4); dst[k] += f;
            }
            r++;
        }
        return r >= n_rows;
    };
    auto collect = [&](int g, int n_rows, float* raw /* GH_SLOTS words, chunked like a row */) {
        if (!host_rows) {
            volatile float* const mb = mailbox + g * MAILBOX_GROUP_WORDS;
            for (int k = 0; k < GH_SLOTS; k++) raw[k] = mb[k];
            return;
        }
        (void)n_rows;   // (answered(g) has added every row)
        volatile uint32_t* const T = host_rows + (size_t)g * EV_MAX_WGS * GH_SLOTS;
        for (int k = 0; k < GH_SLOTS; k++) {
            float t = 0.0f;
            for (int q = 0; q < EV_ROW_GROUPS; q++) t += rows_part[g][q][k];
            raw[k] = t;
        }
        const uint32_t vcount = T[30];   // payload 29 of row 0: the frame's valid-pixel count (bits)
        memcpy(&raw[30], &vcount, 4);
    };
    // One argument line: payload first, sequence number last (x86 stores are not reordered with each other; the compiler barrier
    // keeps the order).  Group g's line sits 64 bytes behind group g - 1's in the BAR block; the pinned line has group 0 only.
    auto write_line = [&](int grp, int seq, uint32_t cmd, int kind, int level, const float* pose) {
        uint32_t wds[16] = {0};
        wds[1] = cmd | ((uint32_t)kind << 8) | ((uint32_t)level << 16);
        if (pose)
            for (int col = 0; col < 4; col++)
                for (int r = 0; r < 3; r++) memcpy(&wds[2 + col * 3 + r], &pose[col * 4 + r], 4);
        wds[0] = (uint32_t)seq;
        uint32_t x = 0;
        for (int k = 0; k < 15; k++) x ^= wds[k];
        wds[15] = x;
        if (bar_line) {
            // through the BAR: the mapping is write-combining, so the 64 bytes gather in one of the core's WC buffers and leave
            // as one posted write when the sfence drains it (without the fence the line sits there until something evicts
            // it: tools/probe/pingpong.hip reads 560 us per round trip instead of 1.8).  A torn line fails the xor word.
            volatile uint64_t* dst = reinterpret_cast<volatile uint64_t*>(bar_line + 16 * grp);
            for (int k = 0; k < 16; k += 2) dst[k >> 1] = (uint64_t)wds[k] | ((uint64_t)wds[k + 1] << 32);
            return;
        }
        for (int k = 1; k < 16; k++) arg_line[k] = wds[k];
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        arg_line[0] = wds[0];
    };
    // group 0's line, then the fence that sends every line written since the last one on its way
    auto publish = [&](int seq, uint32_t cmd, int kind, int level, const float* pose) {
        write_line(0, seq, cmd, kind, level, pose);
        if (bar_line) host_store_fence();
    };
    // Retire a pre-launched evaluation the loop did not need: ARG_SKIP, then wait until the launch has SEEN it (mailbox word
    // 32 := its sequence number).  Usually that launch is already polling and answers within a PCIe round trip; when it is
    // still queued behind other streams' kernels the wait is what keeps the next frame's first arguments from overwriting the
    // line it has yet to read (measured without it: 1 overlap run in ~10 lost 50 ms -- one ARG_TIMEOUT -- in a single frame).
    // launch `seq` -- or a later one, which cannot have started before `seq` had drained -- has taken itself out (ARG_SKIP or its
    // own ARG_TIMEOUT) and said so in mailbox word 32
    auto retired = [&](int seq) -> bool { return (int32_t)((uint32_t)float_bits(mailbox[32]) - (uint32_t)seq) >= 0; };
    auto retire = [&](int seq) {
        for (int g = 1; g < n_groups; g++) write_line(g, seq, ARG_SKIP, 0, 0, nullptr);
        publish(seq, ARG_SKIP, 0, 0, nullptr);
        for (long spin = 0; spin < 400000000L; spin++) {  // (bounded; the launch gives up by itself after ARG_TIMEOUT)
            if (retired(seq)) break;
            if ((spin & 0xFFFF) == 0xFFFF) sched_yield();
        }
    };
    // a pre-launched evaluation that has not been given its arguments yet; retired on every way out
    struct Pending {
        int seq = 0;
        decltype(retire)* ret;
        ~Pending() { if (seq) (*ret)(seq); }
    } pending;
    pending.ret = &retire;
    auto prelaunch = [&]() -> int {
        pending.seq = next_seq();
        track_eval_poll_kernel<<<n_groups * EV_MAX_WGS, EV_THREADS, 0, st>>>(pl, w.partial, w.sync, w.result, mailbox, pending.seq, parity);
        return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
    };

    // The LM loop.  An evaluation is handed the loop's next pose AND (BAR line + a mailbox with room for the answers,
    // gps_track_state.mailbox_bytes) the poses the loop would evaluate after it if it is rejected: a rejection reads nothing of
    // the evaluation it rejects (the pose goes back to the last good one, the damping goes up, the step is solved from the
    // last good Hessian), so those poses are known now.  28 % of the evaluations of a frame are rejections, mostly in runs
    // (tools/lm_trace.py); with the follow-up already evaluated the loop goes on without another host <-> device round trip.
    // The sums it consumes are those of the same launches of the same kernel body: the poses are bit-equal with and without.
    LmLoop lm;
    lm.start(c, ts->pose_M, ts->pose_invM);
    int eval_launches = 0, spec_issued = 0, spec_used = 0;
    for (int k = 0; k < 16; k++) ts->diag[k] = 0;
    const int use_weights = ts->frames_processed >= 100;

    while (lm.active) {
        const LmRequest req = lm.request();
        const int level = req.level, it = req.kind;
        LmRequest cand[EV_GROUPS];
        int nc = 0;
        // (the first evaluation of a level is only rejected when it has no valid pixel at all: nothing rides along with it)
        if (n_groups > 1 && lm.iter > 0) {
            LmLoop sim = lm;
            while (nc < n_groups - 1) {
                sim.apply(nullptr);
                if (!sim.active) break;
                cand[nc++] = sim.request();
            }
        }
        GhArgs a;
        a.depth = dl[level]; a.vw = lw[level]; a.vh = lh[level];
        a.view_intr = make_float4(lintr[level][0], lintr[level][1], lintr[level][2], lintr[level][3]);
        a.pn = w.pn;
        a.sw = W; a.sh = H;
        a.scene_intr = make_float4(lintr[0][0], lintr[0][1], lintr[0][2], lintr[0][3]);
        a.approxInvPose = load_mat(req.pose); a.scenePose = load_mat(ts->pose_pc_M);
        a.space_thresh = c->space_thresh[level]; a.tukey_cutoff = c->tukey_cutoff; a.vf_min = s.view_frustum_min;
        a.vf_max = s.view_frustum_max; a.use_weights = use_weights; a.frames_to_skip = c->frames_to_skip;
        a.frames_to_weight = c->frames_to_weight;
        const int n_wgs = min(EV_MAX_WGS, gps_div_up(a.vw * a.vh, EV_THREADS));
        float raw[EV_GROUPS][GH_SLOTS];  // per group two 64-byte chunks: 15 payload words + the sequence number each (eval_body)
        int answers = 1;                 // raw[0 .. answers) hold results: the evaluation itself, then the poses that rode along
        int ride_seq = 0;                // host-summed rows: the launch whose riding-along groups may still be collected (0: none)
        if (mailbox) {
            // this evaluation is the pre-launched kernel (or the frame's first launch): hand it its arguments, then put
            // the NEXT evaluation on the stream before waiting -- its launch cost overlaps this evaluation
            if (!pending.seq && prelaunch() != GPS_OK) GPS_FAIL_LAUNCH();
            const int seq = pending.seq;
            pending.seq = 0;
            for (int g = 0; g < n_groups; g++) {   // per-state sequence numbers (>= 1): nothing stale can match
                mailbox[g * MAILBOX_GROUP_WORDS + 15] = 0.0f; mailbox[g * MAILBOX_GROUP_WORDS + 31] = 0.0f;
                rows_reset(g);
            }
            for (int g = 1; g < n_groups; g++) {
                if (g <= nc) write_line(g, seq, ARG_RUN, cand[g - 1].kind, cand[g - 1].level, cand[g - 1].pose);
                else write_line(g, seq, ARG_SKIP, 0, 0, nullptr);
            }
            publish(seq, ARG_RUN, it, level, req.pose);
            eval_launches++;
            spec_issued += nc;
            if (prelaunch() != GPS_OK) GPS_FAIL_LAUNCH();
            // spin on the sequence number the kernel writes last (bounded)
            bool got = false, by_mailbox = true;
            // (diagnostic, free: the time-stamp counter around every poll tells a GPU that answered late from a host
            // thread that was not running -- the longest gap between two consecutive polls is ~20 ns unless the thread
            // was descheduled in between)
            const unsigned long long tsc0 = host_cycles();
            unsigned long long tsc_prev = tsc0, tsc_gap = 0;
            for (long spin = 0; spin < 200000000L; spin++) {
                if (answered(0, seq, n_wgs)) { got = true; break; }
                // the launch gave up before its line arrived (this thread did not run for longer than ARG_TIMEOUT): no answer will
                // come -- straight to the plain launch below instead of spinning out the budget (rounds 3-4 did: 2 s per event)
                if ((spin & 0xFF) == 0xFF && retired(seq)) { got = answered(0, seq, n_wgs); break; }
                const unsigned long long now = host_cycles();
                if (now - tsc_prev > tsc_gap) tsc_gap = now - tsc_prev;
                tsc_prev = now;
                // a result normally lands within ~20 us (a few thousand polls); a host that is still spinning far beyond
                // that is oversubscribed or the GPU is busy elsewhere: stop burning the core between polls
                if ((spin & 0xFFFF) == 0xFFFF) sched_yield();
            }
            if (tsc_prev - tsc0 > 6000000ull) {  // > ~2-3 ms at 2-3 GHz: rare; say which side lost the time
                static int said = 0;
                if (said < 16 && ++said)
                    fprintf(stderr, "[gps_slam_hip] tracker: evaluation %d answered after %.2f Mcycles (TSC); longest gap between two "
                                    "polls of this thread %.2f Mcycles (level %d, iteration %d)\n", seq, (tsc_prev - tsc0) * 1e-6,
                            tsc_gap * 1e-6, level, lm.iter);
            }
            if (!got) {
                // No answer within the spin budget.  Either launch `seq` gave up before its line arrived (this thread was
                // descheduled for longer than ARG_TIMEOUT between the launch and the publish) or it has not STARTED yet (the
                // stream is held behind another stream's gate).  Launch seq + 1 may only be retired through the argument line
                // once launch seq is known to have read it -- otherwise seq, starting late, would never find its line and sit
                // out ARG_TIMEOUT with the frame stream behind it.  So: leave RUN(seq) in place and wait until seq has answered
                // after all (tags) or has retired itself (acknowledgement word).
                bool late = false, gone = false;
                for (long spin = 0; spin < 400000000L && !late && !gone; spin++) {
                    late = answered(0, seq, n_wgs);
                    gone = retired(seq);
                    if ((spin & 0xFFFF) == 0xFFFF) sched_yield();
                }
                if (!late && !gone) {
                    // Neither an answer nor a retirement (e.g. the line arrived while part of the launch's workgroups had
                    // already sat out ARG_TIMEOUT: the summer then waits for rows that never come and gives up silently).
                    // Drain the stream instead of failing the frame: launch seq ends by one of its timeouts, the queued
                    // launch seq + 1 never finds its line and retires itself; then nobody polls the line any more and the
                    // plain launch below redoes the evaluation (same inputs, same fixed-order sums).
                    fprintf(stderr, "[gps_slam_hip] tracker: evaluation %d neither answered nor retired; draining the stream\n", seq);
                    if (hipStreamSynchronize(st) != hipSuccess) GPS_FAIL_LAUNCH();
                    pending.seq = 0;
                }
                got = late;
            }
            if (!got) {
                static int warned = 0;   // (rare by construction; a steady stream of these is a bug worth seeing)
                if (warned < 8 && ++warned)
                    fprintf(stderr, "[gps_slam_hip] tracker: evaluation %d gave up waiting for its argument line; redone by a plain launch\n", seq);
                if (pending.seq) retire(pending.seq);
                pending.seq = 0;
                // a launch that gave up may have left the evaluation ticket partially counted and rows half delivered:
                // both start from zero for the plain launch (the valid-pixel counts next to the ticket stay)
                if (hipMemsetAsync(w.sync, 0, sizeof(uint32_t), st) != hipSuccess) GPS_FAIL_LAUNCH();
                if (hipMemsetAsync(w.partial, 0, (size_t)EV_GROUPS * EV_MAX_WGS * GH_SLOTS * sizeof(uint32_t), st) != hipSuccess) GPS_FAIL_LAUNCH();
                const int seq2 = next_seq();
                if (it == TRK_ROTATION) track_eval_kernel<TRK_ROTATION><<<n_wgs, EV_THREADS, 0, st>>>(a, w.partial, w.sync, w.result, mailbox, seq2, parity);
                else if (it == TRK_TRANSLATION) track_eval_kernel<TRK_TRANSLATION><<<n_wgs, EV_THREADS, 0, st>>>(a, w.partial, w.sync, w.result, mailbox, seq2, parity);
                else track_eval_kernel<TRK_BOTH><<<n_wgs, EV_THREADS, 0, st>>>(a, w.partial, w.sync, w.result, mailbox, seq2, parity);
                GPS_LAUNCH_CHECK();
                if (hipStreamSynchronize(st) != hipSuccess || float_bits(mailbox[15]) != seq2 || float_bits(mailbox[31]) != seq2)
                    GPS_FAIL_LAUNCH();
                by_mailbox = false;   // (whatever rode along with launch seq is not waited for)
            }
            if (by_mailbox) collect(0, n_wgs, raw[0]);
            else for (int k = 0; k < GH_SLOTS; k++) raw[0][k] = mailbox[k];   // (the plain launch's summer wrote the block)
            // The poses that rode along: their groups started with group 0's and end within a few microseconds of it.
            // Device summer: ALWAYS waited for, needed or not -- the next launch's lines may only be written once every workgroup of
            // this one that has pixels has delivered its row (a workgroup that finds a later line takes itself out; its group's
            // summer would then wait for that row until ROW_TIMEOUT, with the frame stream behind it).
            // Host-summed rows: nobody on the device waits for a row, so a pose that rode along costs the host nothing unless the
            // loop gets to it -- its rows are waited for and added only then (ride_seq below).
            for (int g = 1; by_mailbox && !host_rows && g <= nc; g++) {
                const int lg = cand[g - 1].level;
                const int n_wgs_g = min(EV_MAX_WGS, gps_div_up(lw[lg] * lh[lg], EV_THREADS));
                bool have = false;
                for (long spin = 0; spin < 200000000L; spin++) {
                    if (answered(g, seq, n_wgs_g)) { have = true; break; }
                    if ((spin & 0xFFFF) == 0xFFFF) sched_yield();
                }
                if (!have) {
                    static int warned2 = 0;
                    if (warned2 < 8 && ++warned2)
                        fprintf(stderr, "[gps_slam_hip] tracker: the pose riding along with evaluation %d (group %d) never answered\n", seq, g);
                    break;
                }
                collect(g, n_wgs_g, raw[g]);
                answers = g + 1;
            }
            if (by_mailbox && host_rows) ride_seq = seq;
        } else {
            const int seq = next_seq();
            if (it == TRK_ROTATION) track_eval_kernel<TRK_ROTATION><<<n_wgs, EV_THREADS, 0, st>>>(a, w.partial, w.sync, w.result, nullptr, seq, parity);
            else if (it == TRK_TRANSLATION) track_eval_kernel<TRK_TRANSLATION><<<n_wgs, EV_THREADS, 0, st>>>(a, w.partial, w.sync, w.result, nullptr, seq, parity);
            else track_eval_kernel<TRK_BOTH><<<n_wgs, EV_THREADS, 0, st>>>(a, w.partial, w.sync, w.result, nullptr, seq, parity);
            GPS_LAUNCH_CHECK();
            eval_launches++;
            // the reference's GPU tracker reads its 32 accumulators back every iteration as well
            if (hipMemcpyAsync(raw[0], w.result, sizeof(raw[0]), hipMemcpyDeviceToHost, st) != hipSuccess) GPS_FAIL_LAUNCH();
            if (hipStreamSynchronize(st) != hipSuccess) GPS_FAIL_LAUNCH();
            if (float_bits(raw[0][15]) != seq || float_bits(raw[0][31]) != seq) GPS_FAIL_LAUNCH();  // the summer gave up
        }
        // the loop's decision on the evaluation -- and, while it keeps rejecting, on the poses that rode along with it
        bool rejected = true;
        for (int g = 0; g < (ride_seq ? nc + 1 : answers) && rejected && lm.active; g++) {
            if (g > 0) {
                if (!lm.request().same(cand[g - 1])) break;   // (cannot happen: the same arithmetic on the same state)
                if (ride_seq) {   // the rows of this pose's group: waited for and added only now that the loop has got to it
                    const int lg = cand[g - 1].level;
                    const int n_wgs_g = min(EV_MAX_WGS, gps_div_up(lw[lg] * lh[lg], EV_THREADS));
                    bool have = false;
                    for (long spin = 0; spin < 200000000L; spin++) {
                        if (answered(g, ride_seq, n_wgs_g)) { have = true; break; }
                        if ((spin & 0xFFFF) == 0xFFFF) sched_yield();
                    }
                    if (!have) break;   // (its group never delivered: the loop evaluates the pose itself, as if nothing had ridden along)
                    collect(g, n_wgs_g, raw[g]);
                }
                spec_used++;
            }
            float host[GH_SLOTS];
            for (int d = 0; d < 30; d++) host[d] = raw[g][d + d / 15];  // payload d lives in word d + d / 15
            rejected = lm.apply(host);
            if (lm.bad_pose) return GPS_ERR_ARG;
        }
    }
    const int n_valid_bits = lm.n_valid_bits, last_type = lm.last_type, nvalid_depth_good = lm.nvalid_depth_good;
    const float f_depth_good = lm.f_depth_good;
    const float *M = lm.M, *invM = lm.invM, *hessian_depth_good = lm.hessian_depth_good;
    for (int l = 0; l < GPS_TRACK_MAX_LEVELS && l < 8; l++) ts->diag[l] = (float)lm.evals[l];
    ts->diag[12] = (float)spec_issued; ts->diag[13] = (float)spec_used;   // poses that rode along / that the loop then consumed
    memcpy(ts->pose_M, M, 64); memcpy(ts->pose_invM, invM, 64);
    // UpdatePoseQuality: the residual score (the SVM verdict only feeds failure modes that are off by default,
    // ITMLibSettings.cpp:42 behaviourOnFailure = FAILUREMODE_IGNORE)
    int n_max = 0;
    if (eval_launches > 0) {
        n_max = n_valid_bits;  // delivered with every evaluation's sums
    } else {
        int slots[VC_SLOTS];
        if (hipMemcpyAsync(slots, w.sync + VC_BASE + parity * VC_SLOTS, sizeof(slots), hipMemcpyDeviceToHost, st) != hipSuccess) GPS_FAIL_LAUNCH();
        if (hipStreamSynchronize(st) != hipSuccess) GPS_FAIL_LAUNCH();
        for (int k = 0; k < VC_SLOTS; k++) n_max += slots[k];
    }
    ts->diag[8] = (float)nvalid_depth_good; ts->diag[9] = f_depth_good;
    ts->diag[10] = n_max > 0 ? sqrtf(((float)nvalid_depth_good * f_depth_good + (float)(n_max - nvalid_depth_good) * c->space_thresh[0]) /
                                    (float)n_max) : 0.0f;
    float det = 0.0f;
    if (last_type == TRK_BOTH) { det = Chol(hessian_depth_good, 6).determinant(); if (isnan(det)) det = 0.0f; }
    ts->diag[11] = det;
    if (eval_launches > 0) ts->scratch_epoch = 2 - parity;  // 1 + the other parity (its slot was cleared by those launches)
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
    if (ts->age_point_cloud != -1) {  // ITMTrackingState::HasValidPointCloud
        if (ts->age_point_cloud >= 0) ts->frames_processed++; else ts->frames_processed = 0;
        // (the depth conversion rides in the tracker's prepare launch)
        const double t0 = wall_ms();
        if ((r = track_camera_impl(s, cfg, ts, scratch, scratch_bytes, stream, depth_mm)) != GPS_OK) return r;
        ts->diag[14] = (float)(wall_ms() - t0);   // host milliseconds in the tracker (launches + LM loop)
    } else {
        if ((r = gps_tsdf_convert_depth(s, depth_mm, stream)) != GPS_OK) return r;
    }
    if (before_fusion) before_fusion(user);  // everything above only READ the volume; what follows modifies it
    const double t1 = wall_ms();
    if ((r = gps_tsdf_allocate(s, ts->pose_M, ts->pose_invM, stream)) != GPS_OK) return r;
    if ((r = gps_tsdf_integrate(s, ts->pose_M, stream)) != GPS_OK) return r;
    if ((r = gps_tsdf_expected_depths_and_raycast(s, ts->pose_M, ts->pose_invM, 0, 1, stream)) != GPS_OK) return r;
    if ((r = gps_tsdf_icp_maps(s, ts->pose_invM, stream)) != GPS_OK) return r;
    memcpy(ts->pose_pc_M, ts->pose_M, 64);  // pose_pointCloud := pose_d (ITMTrackingController.h:87-93)
    ts->age_point_cloud = (ts->age_point_cloud == -1) ? -2 : 0;
    ts->diag[15] = (float)(wall_ms() - t1);   // host milliseconds spent ENQUEUEING the fusion + raycast + ICP-map kernels
    return GPS_OK;
}

}  // extern "C"

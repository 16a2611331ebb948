// Backward of the `ges` rasterizer, column-strip decomposition.
//
//   <- gsplat::rasterize_to_pixels_bwd_ges_gs_parallel_tensor
//      (gsplat/rasterizer/rasterize_to_pixels_bwd_ges_new_parallel.cu:18-201)
//
// The reference enumerates a Gaussian's 2r x 2r pixel box (x_min + 1 .. x_min + 2r, x_min = int(x) - r; rows likewise;
// :83-96) as ceil(4 r^2 / 32) groups of 32 pixel slots, one lane per slot, and reduces ten partial sums per group across the
// lanes (:175-199).  The same pixels with the same per-pixel arithmetic are visited here in another order:
//
//   * a Gaussian owns GW = 4 / 8 / 16 / 32 / 64 adjacent lanes (the smallest power of two >= r); lane l of the group walks DOWN
//     the box in two columns, l and l + r, one row per trip.  dx is a per-lane constant, so the quadratic form per row is two
//     packed FMAs for the lane's two pixels, and the gradient sums factor:  with S0 = sum v_sigma, S1 = sum v_sigma dy,
//     S2 = sum v_sigma dy^2 per column
//         v_conic = (dx^2 S0 / 2, dx S1, S2 / 2)   v_xy = (ca dx S0 + cb S1, cb dx S0 + cc S1)   v_opacity = -S0 / opacity
//     -- three packed accumulations per trip instead of six scalar ones, and NO cross-lane reduction inside the loop;
//   * the ten totals are reduced once per Gaussian over its GW lanes (2..4 DPP adds per value for the common classes) and
//     written by the group's first lane as ONE 48-byte row {v_colors[4], v_conics[3], v_means2d[2], v_opacity, 0, 0}: a
//     Gaussian is owned by exactly one group, so the row is a plain store -- no atomics, no zero-fill, one dirty sector;
//   * the box is trimmed to the rows / columns the {alpha >= 1/255} ellipse can reach (the record's bounds, pack_record):
//     pixels outside contribute exact zeros in the reference too;
//   * per-pixel inputs: d loss / d colours as float4 and {d loss / d weight sum, depth cut} as one float2 per pixel (written
//     by the forward's compose epilogue): lanes of a group read r consecutive pixels -> 16 r and 8 r contiguous bytes.
//
// Work list: Gaussian ids by class, inside a class by (image band, id) so that neighbouring groups -- and an XCD's contiguous
// eighth of the list -- gather from neighbouring pixels; built by the binning (splat_bin.hpp: bwd_key); 64 / GW Gaussians per
// wave task.  Any order gives the same rows.
#include <cstdlib>

#include "common.hpp"
#include "launch_timing.hpp"
#include "splat_bin.hpp"
#include "splat_math.hpp"

// Residency reserve for the frame chain (gps_set_frame_chain_reserve).
// The strip kernel runs 6 workgroups = 6 waves x 80 VGPRs per SIMD: 480 of the 512 registers, and a wave of it that retires
// frees 80 -- never the 112 one wave of the tracker's pre-launched evaluation needs, so beside this kernel an evaluation's
// workgroups waited until two strip waves of one SIMD retired at the same moment, while every freed slot went to the next strip
// workgroup (kernel timeline around a keyframe: 17 evaluations in 626 us instead of ~190).  With the reserve on, 28 KB of unused
// dynamic LDS per workgroup lets 5 of them share a compute unit (5 x 28 <= 160 KB < 6 x 28): 400 registers per SIMD, 112 free.
// Measured (tools/probe/ab_envval.sh, 3 + 3 runs): overlap schedule 1,290 -> 1,315 frames/s; the strip kernel alone is slower
// with 5 waves (sequential schedule 963 -> 951), so the pipeline switches the reserve on only while tracking and mapping overlap.
// The forward rasterizer (a retiring workgroup frees 2 x 64), the batched free-view raycaster and colour kernels: no gain, left alone.
// (bit 0: the strip kernel's reserve + the forward's row-major launch order; bit 1, an experiment switch: the forward rasterizer
// held to 3 workgroups per compute unit -- splat_raster.hip)
static int g_frame_chain_reserve = 0;
static inline int frame_chain_reserve_lds() {
    return (__atomic_load_n(&g_frame_chain_reserve, __ATOMIC_RELAXED) & 1) ? 28 * 1024 : 0;
}


namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);  // raw dword buffer, stride 0
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
    return v + __int_as_float(moved);
}

// the ten totals summed over the gw = 4 << k lanes of every group (all lanes of a group end up with the totals).  One body
// for every class: the class only decides, wave-uniformly, how many of the steps run.
struct Totals { float v[10]; };
template <int CTRL>
__device__ __forceinline__ void dpp_step(Totals& t) {
#pragma unroll
    for (int k = 0; k < 10; k++) t.v[k] = dpp_add<CTRL>(t.v[k]);
}
__device__ __forceinline__ void group_sums(Totals& t, int cls) {
    dpp_step<0xB1>(t);                  // quad_perm [1,0,3,2]
    dpp_step<0x4E>(t);                  // quad_perm [2,3,0,1]
    if (cls >= 1) dpp_step<0x141>(t);   // row_half_mirror
    if (cls >= 2) dpp_step<0x140>(t);   // row_mirror
    if (cls >= 3) {
#pragma unroll
        for (int k = 0; k < 10; k++) t.v[k] += __shfl_xor(t.v[k], 16, 64);
    }
    if (cls >= 4) {
#pragma unroll
        for (int k = 0; k < 10; k++) t.v[k] += __shfl_xor(t.v[k], 32, 64);
    }
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

struct StripPix {
    __amdgpu_buffer_rsrc_t rc;    // float4 per pixel: d loss / d render colours
    __amdgpu_buffer_rsrc_t p2;    // float2 per pixel: {d loss / d weight sum, ref_depth + delta_depth}
    int W, H;
};

#ifndef GPS_STRIP_PIPE
#define GPS_STRIP_PIPE 0
#endif
GPS_TUNABLE_REPORT(GPS_STRIP_PIPE, 0);
#ifndef GPS_STRIP_COOP4
#define GPS_STRIP_COOP4 1
#endif
GPS_TUNABLE_REPORT(GPS_STRIP_COOP4, 1);
#ifndef GPS_STRIP_WAVES
#define GPS_STRIP_WAVES 6
#endif
GPS_TUNABLE_REPORT(GPS_STRIP_WAVES, 6);
constexpr float STRIP_LOG2E = 1.4426950408889634f;
constexpr uint32_t STRIP_OOB = 0x7FFFFFF0u;  // a byte offset past every buffer: the load returns 0 without touching memory

// One wave task of class `cls` (wave-uniform): 64 / gw Gaussians, gw = 4 << cls lanes each.
// The 64-lane class covers every radius above 32: a half box wider than 64 columns takes several passes over the same task
// (pass p: columns 64 p + l of each half); pass 0 stores the Gaussian's row, a later pass adds to it.  Returns whether the
// task needs another pass.
// `share` != nullptr (the 64-lane class only): the FOUR waves of the workgroup run the same task, wave w on rows Q_lo + w, + 4,
// ...; each leaves its ten totals in share[w][] and wave 0 adds them in wave order and stores the row (a workgroup barrier on
// both sides: every wave of the workgroup makes the same calls).  A radius-100 Gaussian is 2 passes x 200 rows = 400 dependent
// row trips -- as long as everything else a wave does in a launch at 1200x680 (tools/probe/radius_hist.py): dealt to ONE wave it
// set the kernel's length; shared by four it is 100.
__device__ __forceinline__ bool strip_task(const float4* __restrict__ recs, const int32_t* __restrict__ radii,
                                           const int32_t* __restrict__ ids, int n_ids, int task, int cls, int pass,
                                           const StripPix& px, float* __restrict__ v_rows, int lane,
                                           float (*share)[12] = nullptr, int wave_in_wg = 0) {
    const int row_step = share ? 4 : 1, row_phase = share ? wave_in_wg : 0;
    const int gw_log2 = cls + 2;
    const int l = lane & ((1 << gw_log2) - 1);
    const int slot = (task << (6 - gw_log2)) + (lane >> gw_log2);
    const bool have = slot < n_ids;
    const int id = have ? ids[slot] : 0;
    const float4 ra = recs[3 * (size_t)id], rb = recs[3 * (size_t)id + 1], rcd = recs[3 * (size_t)id + 2];
    const int r = have ? radii[id] : 0;
    const float mx = ra.x, my = ra.y, ca = ra.z, cb = ra.w, cc = rb.x, opac = rb.y, depth = rb.z, col_r = rb.w, col_g = rcd.x,
                col_b = rcd.y;
    const int xb = __float_as_int(rcd.z), yb = __float_as_int(rcd.w);
    const int ex_lo = (int)(short)(xb & 0xffff), ex_hi = xb >> 16, ey_lo = (int)(short)(yb & 0xffff), ey_hi = yb >> 16;
    // the reference's box: columns x0 .. x0 + 2r - 1, rows y0 .. y0 + 2r - 1 (int() truncates toward zero, :83-90)
    const int x0 = (int)mx - r + 1, y0 = (int)my - r + 1;
    // rows this Gaussian can contribute on: box, image, ellipse
    const int q_lo = max(max(0, -y0), ey_lo - y0);
    const int q_hi = have ? min(min(2 * r, px.H - y0), ey_hi - y0 + 1) : 0;   // exclusive
    const int span = max(q_hi - q_lo, 0);
    // wave-uniform trip range (scalar registers); an all-empty task runs no trip
    const int Q_hi = __builtin_amdgcn_readfirstlane(wave_max_i(span > 0 ? q_hi : 0));
    const int Q_lo = min(Q_hi, __builtin_amdgcn_readfirstlane(wave_min_i(span > 0 ? q_lo : 0x7fffffff)));
    // conic in log2 units: alpha = min(0.999, opac * exp2(-s)), s = hA dx^2 + bL dx dy + hC dy^2
    const float hA = 0.5f * STRIP_LOG2E * ca, bL = STRIP_LOG2E * cb, hC = 0.5f * STRIP_LOG2E * cc;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, ka, kb, kc, gx, gy, gop;
    {
        const int colA = l + 64 * pass;                 // column inside the left half of the box; the right one is colA + r
        const int jA = x0 + colA, jB = jA + r;
        const bool okA = have && colA < r && jA >= 0 && jA < px.W && jA >= ex_lo && jA <= ex_hi;
        const bool okB = have && colA < r && jB >= 0 && jB < px.W && jB >= ex_lo && jB <= ex_hi;
        const v2f o2 = {okA ? opac : 0.f, okB ? opac : 0.f};   // a column outside the box / image / ellipse: alpha = 0 fails the test
        const v2f dx = {mx - ((float)jA + 0.5f), mx - ((float)jB + 0.5f)};
        const v2f A2 = hA * dx * dx, Bd = bL * dx;
        v2f S0 = {0.f, 0.f}, S1 = S0, S2 = S0;
        float pyf = (float)(y0 + Q_lo + row_phase) + 0.5f;
        uint32_t offA = (uint32_t)((y0 + Q_lo + row_phase) * px.W + jA) * 16u, offB = offA + (uint32_t)r * 16u;
        const uint32_t row_bytes = (uint32_t)px.W * 16u * (uint32_t)row_step;
        const float row_stepf = (float)row_step;
        // One row of the lane's two columns in two halves: eval() needs the record only and ends by ISSUING the row's four
        // gathers; accum() consumes them.  GPS_STRIP_PIPE = 1: the next row is evaluated (and its gathers issued) before the
        // current one is accumulated, so a wave always has two rows' gathers in flight (two Row sets, trip unrolled by two:
        // no register moves); an odd row count is padded with a row no lane is on (it reads nothing).
        struct Row { float dy, alA, alB; v2f ov; bool onA, onB; float4 vcA, vcB; float2 pA, pB; };
        int q = Q_lo + row_phase;
        auto eval = [&](Row& R) {
            R.dy = my - pyf;
            const bool rowok = (uint32_t)(q - q_lo) < (uint32_t)span;
            const float t = hC * R.dy;
            const v2f s = (Bd + t) * R.dy + A2;
            const v2f vis = {__builtin_amdgcn_exp2f(-s.x), __builtin_amdgcn_exp2f(-s.y)};
            R.ov = o2 * vis;
            R.alA = fminf(0.999f, R.ov.x); R.alB = fminf(0.999f, R.ov.y);
            R.onA = rowok && !(s.x < 0.f) && !(R.alA < 1.f / 255.f);
            R.onB = rowok && !(s.y < 0.f) && !(R.alB < 1.f / 255.f);
            const uint32_t fA = R.onA ? offA : STRIP_OOB, fB = R.onB ? offB : STRIP_OOB;
            R.vcA = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(px.rc, fA, 0, 0));
            R.vcB = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(px.rc, fB, 0, 0));
            R.pA = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(px.p2, fA >> 1, 0, 0));
            R.pB = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(px.p2, fB >> 1, 0, 0));
            pyf += row_stepf;
            offA += row_bytes; offB += row_bytes;
            q += row_step;
        };
        auto accum = [&](const Row& R) {
            // (a lane that failed the test read zeros: cut = 0 < depth, so it fails here as well)
            const bool hitA = R.onA && !(depth > R.pA.y), hitB = R.onB && !(depth > R.pB.y);
            const float wA = hitA ? R.alA : 0.f, wB = hitB ? R.alB : 0.f;
            c0 += wA * R.vcA.x; c1 += wA * R.vcA.y; c2 += wA * R.vcA.z; c3 += wA * R.vcA.w;
            c0 += wB * R.vcB.x; c1 += wB * R.vcB.y; c2 += wB * R.vcB.z; c3 += wB * R.vcB.w;
            const float vaA = col_r * R.vcA.x + col_g * R.vcA.y + col_b * R.vcA.z + depth * R.vcA.w + R.pA.x;
            const float vaB = col_r * R.vcB.x + col_g * R.vcB.y + col_b * R.vcB.z + depth * R.vcB.w + R.pB.x;
            // v_sigma = -opac vis v_alpha where the clamp is inactive (:159-171)
            const v2f vs = {(hitA && R.ov.x <= 0.999f) ? -R.ov.x * vaA : 0.f, (hitB && R.ov.y <= 0.999f) ? -R.ov.y * vaB : 0.f};
            S0 += vs;
            const v2f vsy = vs * R.dy;
            S1 += vsy;
            S2 += vsy * R.dy;
        };
#if GPS_STRIP_PIPE
        if (Q_lo + row_phase < Q_hi) {
            const int n2 = ((Q_hi - Q_lo - row_phase + row_step - 1) / row_step + 1) & ~1;
            Row R0, R1;
            eval(R0);
#pragma unroll 1
            for (int k = 1; k < n2 - 1; k += 2) {
                eval(R1); accum(R0);
                eval(R0); accum(R1);
            }
            eval(R1); accum(R0); accum(R1);
        }
#else
#pragma unroll 1
        while (q < Q_hi) {
            Row R;
            eval(R);
            accum(R);
        }
#endif
        // per column -> this lane's share of the six geometric gradients
        const v2f dS0 = dx * S0;
        ka = 0.5f * (dx.x * dS0.x + dx.y * dS0.y);
        kb = dx.x * S1.x + dx.y * S1.y;
        kc = 0.5f * (S2.x + S2.y);
        const float s0d = dS0.x + dS0.y, s1 = S1.x + S1.y;
        gx = ca * s0d + cb * s1;
        gy = cb * s0d + cc * s1;
        gop = -(S0.x + S0.y);
    }
    Totals t = {{c0, c1, c2, c3, ka, kb, kc, gx, gy, gop}};
    group_sums(t, cls);
    if (share) {   // (wave-uniform)
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 10; k++) share[wave_in_wg][k] = t.v[k];
        }
        __syncthreads();
        if (wave_in_wg == 0) {
#pragma unroll
            for (int k = 0; k < 10; k++) t.v[k] = ((share[0][k] + share[1][k]) + share[2][k]) + share[3][k];
        }
        __syncthreads();   // (the next pass / task overwrites the slots)
        if (wave_in_wg != 0) return cls == 4 && __builtin_amdgcn_readfirstlane(wave_max_i(r)) > 64 * (pass + 1);
    }
    if (have && l == 0) {
        float4* row = reinterpret_cast<float4*>(v_rows + 12 * (size_t)id);
        float4 o0 = make_float4(t.v[0], t.v[1], t.v[2], t.v[3]), o1 = make_float4(t.v[4], t.v[5], t.v[6], t.v[7]),
               o2r = make_float4(t.v[8], opac > 0.f ? t.v[9] / opac : 0.f, 0.f, 0.f);
        if (pass > 0) {  // (same lane, same addresses as its own pass-0 store: program order)
            const float4 p0 = row[0], p1 = row[1], p2 = row[2];
            o0 = make_float4(o0.x + p0.x, o0.y + p0.y, o0.z + p0.z, o0.w + p0.w);
            o1 = make_float4(o1.x + p1.x, o1.y + p1.y, o1.z + p1.z, o1.w + p1.w);
            o2r = make_float4(o2r.x + p2.x, o2r.y + p2.y, 0.f, 0.f);
        }
        row[0] = o0; row[1] = o1; row[2] = o2r;
    }
    return cls == 4 && __builtin_amdgcn_readfirstlane(wave_max_i(r)) > 64 * (pass + 1);
}

struct StripArgs {
    const float4* recs;
    const int32_t* radii;
    const int32_t* cls_ids;     // [GPS_BWD_CLASSES][cls_stride]
    const int32_t* cls_counts;  // [GPS_BWD_CLASSES], device
    int cls_stride;
    const float* v_render_colors;
    const float* pix2;
    int W, H;
    float* v_rows;
};

constexpr int NCLS = GPS_BWD_CLASSES;

__global__ __launch_bounds__(256, GPS_STRIP_WAVES) void raster_ges_bwd_strip_kernel(StripArgs a, gps::LaunchStamp stamp) {
    gps::StampScope timed(stamp);
    const uint32_t n_px = (uint32_t)(a.W * a.H);
    const StripPix px = {buf_rsrc(a.v_render_colors, n_px * 16u), buf_rsrc(a.pix2, n_px * 8u), a.W, a.H};
    const int lane = threadIdx.x & 63, wave_in_wg = threadIdx.x >> 6;
    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8): every XCD takes ONE contiguous eighth of every class's
    // list (neighbouring ids cover neighbouring pixels: an XCD then gathers from one band of the gradient images, which its
    // L2 holds); inside an XCD the wide classes come first, the 4-lane tasks fill the tail.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    const int wave_in_xcd = slot * 4 + wave_in_wg, waves_per_xcd = wgs_per_xcd * 4;
    int n_ids[NCLS], t_lo[NCLS], t_cnt[NCLS], total = 0;
#pragma unroll
    for (int k = 0; k < NCLS; k++) {
        n_ids[k] = a.cls_counts[k];
        const int per_task = 64 / (4 << k);
        const int tasks = (n_ids[k] + per_task - 1) / per_task;
        const int per_xcd = (tasks + 7) >> 3;
        t_lo[k] = min(tasks, xcd * per_xcd);
        t_cnt[k] = min(tasks, t_lo[k] + per_xcd) - t_lo[k];
        total += t_cnt[k];
    }
#if GPS_STRIP_COOP4
    // the 64-lane class: one task per WORKGROUP at a time, its rows shared by the four waves (strip_task, `share`)
    __shared__ float share[4][12];
    for (int t4 = slot; t4 < t_cnt[NCLS - 1]; t4 += wgs_per_xcd) {
        int pass = 0;
        while (strip_task(a.recs, a.radii, a.cls_ids + (NCLS - 1) * (size_t)a.cls_stride, n_ids[NCLS - 1], t_lo[NCLS - 1] + t4, NCLS - 1, pass,
                          px, a.v_rows, lane, share, wave_in_wg))
            ++pass;
    }
    total -= t_cnt[NCLS - 1];
    constexpr int TOP = NCLS - 2;
#else
    constexpr int TOP = NCLS - 1;
#endif
    for (int f = wave_in_xcd; f < total; f += waves_per_xcd) {
        int rest = f, cls = TOP;   // widest class first
        while (cls > 0 && rest >= t_cnt[cls]) { rest -= t_cnt[cls]; --cls; }
        cls = __builtin_amdgcn_readfirstlane(cls);
        int pass = 0;
        while (strip_task(a.recs, a.radii, a.cls_ids + cls * (size_t)a.cls_stride, n_ids[cls], t_lo[cls] + rest, cls, pass, px, a.v_rows,
                          lane))
            ++pass;
    }
}

// {d loss / d weight sum, ref_depth + delta_depth} per pixel from the two separate images (the fused train step's forward
// epilogue writes the pair directly)
__global__ __launch_bounds__(256) void pair_image_kernel(int P, const float* __restrict__ v_render_alphas,
                                                        const float* __restrict__ ref_depth, float delta_depth,
                                                        float2* __restrict__ pix2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) pix2[i] = make_float2(v_render_alphas[i], ref_depth[i] + delta_depth);
}

// the 48-byte records of gps_gauss_preprocess_fwd from the operator-level arrays (for callers that hold those, not records)
__global__ __launch_bounds__(256) void pack_records_kernel(int N, const float2* __restrict__ means2d, const float* __restrict__ conics,
                                                          const float4* __restrict__ colors, const float* __restrict__ opacities,
                                                          const int32_t* __restrict__ radii, float4* __restrict__ recs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    gps::Proj o;
    const float2 m = means2d[i];
    const float4 c = colors[i];
    o.mx = m.x; o.my = m.y; o.z = c.w;
    o.ca = conics[3 * i]; o.cb = conics[3 * i + 1]; o.cc = conics[3 * i + 2];
    o.radius = radii[i];
    gps::pack_record(o, c.x, c.y, c.z, opacities[i], recs + 3 * (size_t)i);
}

}  // namespace

namespace gps {

int raster_ges_bwd_strips_launch(int N, const float* records, const int32_t* radii, const int32_t* cls_ids,
                                 const int32_t* cls_counts, int cls_stride, const float* v_render_colors, const float* pix2,
                                 int width, int height, float* v_rows, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(N >= 0 && width > 0 && height > 0 && cls_stride >= 0);
    if (N == 0) return GPS_OK;
    GPS_REQUIRE(records && radii && cls_ids && cls_counts && v_render_colors && pix2 && v_rows);
    GPS_REQUIRE((int64_t)width * height * 16 < (int64_t)STRIP_OOB);
    StripArgs a = {(const float4*)records, radii, cls_ids, cls_counts, cls_stride, v_render_colors, pix2, width, height, v_rows};
#ifndef GPS_BWD_STRIP_BLOCKS
#define GPS_BWD_STRIP_BLOCKS (256 * GPS_STRIP_WAVES)   // every wave resident at once: 256 CUs x GPS_STRIP_WAVES workgroups of 4 waves
#endif
    // (with the reserve on, 5 workgroups share a compute unit: launch what is resident at once -- the work is dealt to the
    // launched waves in equal shares, so a sixth workgroup per unit would run its share after the others, in a second round.
    // 4 + 4 and 6 + 6 bench runs on two boxes: overlap 1,328 -> 1,337 and 1,306 -> 1,318 frames/s)
    const int lds_reserve = frame_chain_reserve_lds();
    const int blocks = lds_reserve ? GPS_BWD_STRIP_BLOCKS / GPS_STRIP_WAVES * (GPS_STRIP_WAVES - 1) : GPS_BWD_STRIP_BLOCKS;
    launch_kernel(TK_RASTER_BWD_STRIPS, 0, raster_ges_bwd_strip_kernel, dim3(blocks), dim3(256), (size_t)lds_reserve, (hipStream_t)stream, a);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

// (splat_step.hip: the forward's launch order by list length only pays when the map kernels have the chip to themselves)
bool map_runs_beside_frame_chain() { return (__atomic_load_n(&g_frame_chain_reserve, __ATOMIC_RELAXED) & 1) != 0; }
int frame_chain_reserve_bits() { return __atomic_load_n(&g_frame_chain_reserve, __ATOMIC_RELAXED); }

}  // namespace gps

extern "C" {

void gps_set_frame_chain_reserve(int on) { __atomic_store_n(&g_frame_chain_reserve, on < 0 ? 0 : on, __ATOMIC_RELAXED); }


int gps_raster_ges_bwd_strips(int N, const float* records, const int32_t* radii, const int32_t* cls_ids,
                              const int32_t* cls_counts, int cls_stride, const float* v_render_colors, const float* pix2,
                              int width, int height, float* v_rows, gps_stream stream) {
    return gps::raster_ges_bwd_strips_launch(N, records, radii, cls_ids, cls_counts, cls_stride, v_render_colors, pix2, width,
                                             height, v_rows, stream);
}

int gps_raster_pack_records(int N, const float* means2d, const float* conics, const float* colors, const float* opacities,
                            const int32_t* radii, float* records, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(N >= 0);
    if (N == 0) return GPS_OK;
    GPS_REQUIRE(means2d && conics && colors && opacities && radii && records);
    pack_records_kernel<<<gps_div_up(N, 256), 256, 0, (hipStream_t)stream>>>(N, (const float2*)means2d, conics, (const float4*)colors,
                                                                            opacities, radii, (float4*)records);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_raster_pair_image(int width, int height, const float* v_render_alphas, const float* ref_depth_map, float delta_depth,
                          float* pix2, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(width > 0 && height > 0 && v_render_alphas && ref_depth_map && pix2);
    const int P = width * height;
    pair_image_kernel<<<gps_div_up(P, 256), 256, 0, (hipStream_t)stream>>>(P, v_render_alphas, ref_depth_map, delta_depth,
                                                                          (float2*)pix2);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

}  // extern "C"

// build-time tunables defined inside functions above (gps_build_flags)
GPS_TUNABLE_REPORT(GPS_BWD_STRIP_BLOCKS, 256 * 6);

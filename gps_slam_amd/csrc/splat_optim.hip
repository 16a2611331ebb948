// Fused replacements for the libtorch glue around the rasterizer:
//
// gps_compose_l1 <- raw_gs_model.cpp:318-326 (compose with the TSDF layer) +
//                   :369-417 computeLoss (L1 only: ssim_weight = depth_weight = 0 in every
//                   shipped config) + the autograd backward of both  (~35 libtorch launches)
// gps_adam_step  <- 7 x torch::optim::Adam::step (raw_gs_model.cpp:654-705), one launch
//
// Both are pure HBM streams: one pass, float4 where the layout allows it.
#include <math.h>

#include "common.hpp"
#include "splat_adam.hpp"

namespace {

__global__ __launch_bounds__(256) void compose_l1_kernel(int P, const float4* __restrict__ render_colors,
                                                        const float* __restrict__ weight_sum,
                                                        const float* __restrict__ base_color,
                                                        const float* __restrict__ ref_depth_raw,
                                                        const float* __restrict__ gt_rgb, float* __restrict__ rgb,
                                                        float* __restrict__ depth, float* __restrict__ loss,
                                                        float4* __restrict__ v_render_colors,
                                                        float* __restrict__ v_render_alphas, float inv_count) {
    __shared__ float red[4];
    float part = 0.f;
    const int stride = gridDim.x * blockDim.x;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        const float4 rc = render_colors[p];
        const float w = weight_sum[p];
        const float den = w + 1.0f;  // base colour weight is always 1 (raw_gs_model.cpp:321-323)
        const float n0 = rc.x + base_color[3 * p], n1 = rc.y + base_color[3 * p + 1], n2 = rc.z + base_color[3 * p + 2];
        const float c0 = n0 / den, c1 = n1 / den, c2 = n2 / den;
        rgb[3 * p] = c0; rgb[3 * p + 1] = c1; rgb[3 * p + 2] = c2;
        if (depth) {
            const float ref = ref_depth_raw[p];
            const float bw = ref > 0.f ? 1.f : 0.f;  // depth weight only where the raycast hit (:324-326)
            depth[p] = (rc.w + ref * bw) / (w + bw);
        }
        if (gt_rgb == nullptr) continue;  // render-only call (NoGradGuard paths of the reference)
        const float d0 = gt_rgb[3 * p] - c0, d1 = gt_rgb[3 * p + 1] - c1, d2 = gt_rgb[3 * p + 2] - c2;
        part += fabsf(d0) + fabsf(d1) + fabsf(d2);
        if (v_render_colors) {
            // d mean|gt - rgb| / d rgb = -sgn(gt - rgb) / (3P); sgn(0) = 0 as in torch
            const float g0 = d0 > 0.f ? -inv_count : (d0 < 0.f ? inv_count : 0.f);
            const float g1 = d1 > 0.f ? -inv_count : (d1 < 0.f ? inv_count : 0.f);
            const float g2 = d2 > 0.f ? -inv_count : (d2 < 0.f ? inv_count : 0.f);
            v_render_colors[p] = make_float4(g0 / den, g1 / den, g2 / den, 0.f);
            const float dd = den * den;
            v_render_alphas[p] = -(g0 * n0) / dd - (g1 * n1) / dd - (g2 * n2) / dd;
        }
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0 && loss != nullptr) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_count);
}

// torch::optim::Adam's fresh state (zeros_like per parameter) for up to 16 tensors in one launch
struct ZeroArgs { float* p[16]; int64_t end4[16]; int n; };
__global__ __launch_bounds__(256) void zero_many_kernel(ZeroArgs a) {
    const int64_t total4 = a.end4[a.n - 1];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += stride) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < 15; k++) s += (k < a.n - 1 && e >= a.end4[k]) ? 1 : 0;
        float* base = a.p[0];
        int64_t first = 0;
#pragma unroll
        for (int k = 1; k < 16; k++) if (s == k) { base = a.p[k]; first = a.end4[k - 1]; }   // (static indices: no scratch copy)
        reinterpret_cast<float4*>(base)[e - first] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

struct AdamArgs {
    gps_adam_segment seg[GPS_ADAM_MAX_SEGMENTS];
    float step_size[GPS_ADAM_MAX_SEGMENTS];  // lr / (1 - b1^t), computed in double on the host like libtorch
    int64_t seg_end[GPS_ADAM_MAX_SEGMENTS];  // exclusive prefix end in the flattened index space
    int n_segments;
    float beta1, beta2, one_minus_b1, one_minus_b2, inv_bc2_sqrt, eps;
    int fresh;   // step 1: the moments are zero by definition and not read (AdamScalars::fresh)
};

__device__ __forceinline__ void adam_update(const AdamArgs& a, int s, float g, float& m, float& v, float& p) {
    gps::AdamScalars sc = {a.beta1, a.beta2, a.one_minus_b1, a.one_minus_b2, a.inv_bc2_sqrt, a.eps, a.step_size[s], a.fresh};
    gps::adam_update(sc, g, m, v, p);
}

// One float4 per thread-iteration when the segment allows it (all parameter tensors are 16-byte aligned torch
// allocations; a segment's tail of < 4 elements falls back to scalars): 7 x 16-byte streams per lane.
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    const int64_t total4 = a.seg_end[a.n_segments - 1];  // in units of 4 elements, per-segment padded (see host)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += stride) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < GPS_ADAM_MAX_SEGMENTS - 1; k++) s += (k < a.n_segments - 1 && e >= a.seg_end[k]) ? 1 : 0;
        const int64_t i4 = e - (s == 0 ? 0 : a.seg_end[s - 1]);
        const gps_adam_segment& sg = a.seg[s];
        const int64_t i = i4 * 4;
        if (i + 4 <= sg.numel) {
            const float4 g = *reinterpret_cast<const float4*>(sg.grad + i);
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f), v = m;
            if (!a.fresh) { m = *reinterpret_cast<float4*>(sg.exp_avg + i); v = *reinterpret_cast<float4*>(sg.exp_avg_sq + i); }
            float4 p = *reinterpret_cast<float4*>(sg.param + i);
            adam_update(a, s, g.x, m.x, v.x, p.x); adam_update(a, s, g.y, m.y, v.y, p.y);
            adam_update(a, s, g.z, m.z, v.z, p.z); adam_update(a, s, g.w, m.w, v.w, p.w);
            *reinterpret_cast<float4*>(sg.exp_avg + i) = m;
            *reinterpret_cast<float4*>(sg.exp_avg_sq + i) = v;
            *reinterpret_cast<float4*>(sg.param + i) = p;
        } else {
            for (int64_t j = i; j < sg.numel; j++) {
                float m = 0.f, v = 0.f, p = sg.param[j];
                if (!a.fresh) { m = sg.exp_avg[j]; v = sg.exp_avg_sq[j]; }
                adam_update(a, s, sg.grad[j], m, v, p);
                sg.exp_avg[j] = m; sg.exp_avg_sq[j] = v; sg.param[j] = p;
            }
        }
    }
}

}  // namespace

extern "C" {

int gps_compose_l1(int width, int height, const float* render_colors, const float* weight_sum,
                   const float* base_color, const float* ref_depth_raw, const float* gt_rgb, float* rgb, float* depth,
                   float* loss, float* v_render_colors, float* v_render_alphas, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(width > 0 && height > 0);
    GPS_REQUIRE(render_colors && weight_sum && base_color && rgb);
    GPS_REQUIRE(gt_rgb == nullptr || loss != nullptr);
    GPS_REQUIRE(gt_rgb != nullptr || v_render_colors == nullptr);
    GPS_REQUIRE(depth == nullptr || ref_depth_raw != nullptr);
    GPS_REQUIRE((v_render_colors == nullptr) == (v_render_alphas == nullptr));
    const int P = width * height;
    hipStream_t s = (hipStream_t)stream;
    compose_l1_kernel<<<min(1024, gps_div_up(P, 256)), 256, 0, s>>>(P, (const float4*)render_colors, weight_sum,
                                                                    base_color, ref_depth_raw, gt_rgb, rgb, depth,
                                                                    loss, (float4*)v_render_colors, v_render_alphas,
                                                                    1.0f / (3.0f * (float)P));
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_zero_floats(int n, float* const* ptrs, const int64_t* numels, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(n >= 1 && n <= 16 && ptrs && numels);
    ZeroArgs a = {};
    int64_t run = 0;
    for (int k = 0; k < n; k++) {
        // float4 stores: 16-byte aligned buffers whose length is a multiple of 4 floats (or padded to one by the owner)
        GPS_REQUIRE(numels[k] >= 0 && (numels[k] == 0 || ptrs[k]) && (((uintptr_t)ptrs[k]) & 15) == 0 && (numels[k] & 3) == 0);
        a.p[k] = ptrs[k];
        run += numels[k] / 4;
        a.end4[k] = run;
    }
    for (int k = n; k < 16; k++) { a.p[k] = ptrs[0]; a.end4[k] = run; }
    a.n = n;
    if (run == 0) return GPS_OK;
    zero_many_kernel<<<(int)min((int64_t)4096, (int64_t)gps_div_up(run, 256)), 256, 0, (hipStream_t)stream>>>(a);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_adam_step(const gps_adam_segment* segments, int n_segments, double beta1, double beta2, double eps, int step,
                  gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(segments && n_segments >= 1 && n_segments <= GPS_ADAM_MAX_SEGMENTS && step >= 1);
    AdamArgs a;
    int64_t run = 0;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    for (int k = 0; k < n_segments; k++) {
        GPS_REQUIRE(segments[k].numel >= 0);
        GPS_REQUIRE(segments[k].numel == 0 ||
                    (segments[k].param && segments[k].grad && segments[k].exp_avg && segments[k].exp_avg_sq));
        a.seg[k] = segments[k];
        a.step_size[k] = (float)(segments[k].lr / bc1);
        GPS_REQUIRE((((uintptr_t)segments[k].param | (uintptr_t)segments[k].grad | (uintptr_t)segments[k].exp_avg |
                      (uintptr_t)segments[k].exp_avg_sq) & 15) == 0);  // float4 path
        run += (segments[k].numel + 3) / 4;
        a.seg_end[k] = run;
    }
    for (int k = n_segments; k < GPS_ADAM_MAX_SEGMENTS; k++) { a.seg[k] = segments[0]; a.step_size[k] = 0.f; a.seg_end[k] = run; }
    if (run == 0) return GPS_OK;
    a.n_segments = n_segments;
    a.beta1 = (float)beta1; a.beta2 = (float)beta2;
    a.one_minus_b1 = (float)(1.0 - beta1);
    a.one_minus_b2 = (float)(1.0 - beta2);
    a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    a.eps = (float)eps;
    a.fresh = step == 1 ? 1 : 0;
    adam_kernel<<<min((int64_t)4096, (int64_t)gps_div_up(run, 256)), 256, 0, (hipStream_t)stream>>>(a);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

}  // extern "C"

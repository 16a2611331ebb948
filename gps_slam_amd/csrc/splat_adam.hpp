// The Adam update shared by adam_kernel (splat_optim.hip) and the fused preprocess-backward (splat_fused.hip).
#pragma once
#include <math.h>

#include "common.hpp"

namespace gps {

// scalars of one Adam step for one parameter tensor (host side computes them in double like libtorch)
struct AdamScalars {
    float beta1, beta2, one_minus_b1, one_minus_b2, inv_bc2_sqrt, eps, step_size;  // step_size = lr / (1 - beta1^t)
    // step 1: torch::optim::Adam creates exp_avg / exp_avg_sq as zeros at a parameter's first step, so the kernels take m = v = 0
    // WITHOUT reading the buffers (same operations on the same values as with zeroed buffers) -- a host that re-creates its
    // optimizers (raw_gs_model.cpp:654-659, every localOptimize) need not zero 2 x 59 floats per Gaussian first, and the first
    // step reads a third less
    int fresh;
};

static inline AdamScalars adam_scalars(double lr, double beta1, double beta2, double eps, int step) {
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars a;
    a.beta1 = (float)beta1; a.beta2 = (float)beta2;
    a.one_minus_b1 = (float)(1.0 - beta1); a.one_minus_b2 = (float)(1.0 - beta2);
    a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    a.eps = (float)eps;
    a.step_size = (float)(lr / bc1);
    a.fresh = step == 1 ? 1 : 0;
    return a;
}

// Rounding sequence measured against ATen on gfx950 (tools/adam_probe.py, 2^20 samples, 100% bitwise):
//   add_(g, alpha)        -> fma(alpha, g, m*b1)
//   addcmul_(g, g, value) -> fma(value, g*g, v*b2)
//   sqrt()/c              -> sqrt * float(1/c)   ; add_(eps) unfused
//   addcdiv_(m, d, value) -> fma(value, m/d, p)   (m/d IEEE-correct here, as nvcc's --prec-div default)
__device__ __forceinline__ void adam_update(const AdamScalars& a, float g, float& m, float& v, float& p) {
    m = fmaf(a.one_minus_b1, g, __fmul_rn(m, a.beta1));
    v = fmaf(a.one_minus_b2, __fmul_rn(g, g), __fmul_rn(v, a.beta2));
    const float denom = __fadd_rn(__fmul_rn(sqrtf(v), a.inv_bc2_sqrt), a.eps);
    p = fmaf(-a.step_size, __fdiv_rn(m, denom), p);
}

// splat_fused.hip: backward of the per-Gaussian preprocessing, optionally with the Adam step of sh_rest fused in
// The NEXT optimise iteration's preprocessing forward, run in the tail of this iteration's backward + Adam kernel
// (gps_splat_step::next_viewmat; splat_fused.hip NextFwd): the next camera, the forward's thresholds and outputs, the superblock
// binning's count targets.
struct BinCountOut;
struct NextForward {
    const float *viewmat, *Kmat, *cam_pos;
    int max_gs_radii;
    float near_plane, far_plane, radius_clip;
    int32_t* radii;
    float *means2d, *depths, *conics, *colors, *opacities, *records;
    const BinCountOut* count;
};

int preprocess_bwd_launch(int N, int K, int sh_degree, const float* means, const float* log_scales, const float* quats,
                          const float* opac_logit, const float* sh_dc, const float* sh_rest, const float* viewmat,
                          const float* Kmat, const float* cam_pos, int width, int height, float eps2d,
                          const int32_t* radii, const float* conics, const float* v_means2d, const float* v_conics,
                          const float* v_colors, const float* v_opacities, float* v_means, float* v_log_scales,
                          float* v_quats, float* v_opac_logit, float* v_sh_dc, float* v_sh_rest, float* adam_param,
                          float* adam_m, float* adam_v, AdamScalars sc, const gps_adam_segment* small5,
                          const float* small_step, gps_stream stream, const float* v_rows = nullptr,
                          const NextForward* next = nullptr);

}  // namespace gps

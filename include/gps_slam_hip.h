/*
 * gps_slam_hip.h -- C-ABI of the MI355X (gfx950) hot path of GPS-SLAM.
 *
 * One entry point per kernel-level operator of the reference's splat path
 * (gsplat/rasterizer/bindings.h launchers, called through
 * gsplat/gsplat_wapper.hpp) and of its TSDF path (ITMLib engines called from
 * ITMBasicEngine / ITMDenseMapper / ITMVisualisationEngine).
 *
 * Conventions
 *  - extern "C", plain device pointers + sizes, no torch / C++ types.
 *  - every function takes the hipStream_t to launch on (as void*), never
 *    allocates, never synchronises, and returns GPS_OK (0) or a negative
 *    gps_status.  Capacity overflows are reported through a device-side
 *    status word (see gps_bin_*), because sizes such as n_isects live on the
 *    device: the reference's two host syncs per forward
 *    (isect_tiles_no_depth.cu:238-239) do not exist here.
 *  - all splat tensors are fp32, C = 1 camera (raw_gs_model.cpp:225 always
 *    unsqueezes a single camera), row-major contiguous.
 */
#ifndef GPS_SLAM_HIP_H
#define GPS_SLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GPS_OK = 0,
    GPS_ERR_ARG = -1,      /* invalid argument (null pointer, bad size, unsupported degree ...) */
    GPS_ERR_LAUNCH = -2,   /* hip launch error */
    GPS_ERR_CAPACITY = -3  /* a caller-provided buffer is too small */
} gps_status;

typedef void *gps_stream; /* hipStream_t */

#define GPS_API __attribute__((visibility("default")))

GPS_API const char *gps_version(void);

/* ------------------------------------------------------------------ */
/* Splat: projection                                                   */
/* ------------------------------------------------------------------ */

/* replaces gsplat::fully_fused_projection_fwd_tensor
 * (gsplat/rasterizer/fully_fused_projection_fwd.cu:196-273; PINHOLE, quats+scales path,
 * compensations off as in raw_gs_model.h:283-288).
 * means[N,3] quats[N,4] (wxyz, unnormalised) scales[N,3] (already exp'ed)
 * viewmat[16] K[9] live in device memory.
 * Outputs radii[N] means2d[N,2] depths[N] conics[N,3]; rows with radii==0 are zero-filled
 * (the reference leaves them uninitialised). */
GPS_API int gps_proj_fwd(int N, const float *means, const float *quats, const float *scales, const float *viewmat,
                 const float *K, int width, int height, float eps2d, float near_plane, float far_plane,
                 float radius_clip, int32_t *radii, float *means2d, float *depths, float *conics,
                 gps_stream stream);

/* replaces gsplat::fully_fused_projection_bwd_tensor (fully_fused_projection_bwd.cu:288-403).
 * Writes (does not accumulate) v_means[N,3] v_quats[N,4] v_scales[N,3]; rows with radii<=0 get 0. */
GPS_API int gps_proj_bwd(int N, const float *means, const float *quats, const float *scales, const float *viewmat,
                 const float *K, int width, int height, float eps2d, const int32_t *radii, const float *conics,
                 const float *v_means2d, const float *v_depths, const float *v_conics, float *v_means,
                 float *v_quats, float *v_scales, gps_stream stream);

/* ------------------------------------------------------------------ */
/* Splat: spherical harmonics                                          */
/* ------------------------------------------------------------------ */

/* replaces gsplat::compute_sh_fwd_tensor (compute_sh_fwd.cu:40-72).
 * dirs[N,3] coeffs[N,K,3] masks[N] (uint8, may be NULL) -> colors[N,3]; masked rows are written as 0. */
GPS_API int gps_sh_fwd(int N, int K, int degrees_to_use, const float *dirs, const float *coeffs, const uint8_t *masks,
               float *colors, gps_stream stream);

/* replaces gsplat::compute_sh_bwd_tensor (compute_sh_bwd.cu:56-123).
 * Writes v_coeffs[N,K,3] completely (bands above degrees_to_use and masked rows = 0) and,
 * if v_dirs != NULL, v_dirs[N,3]. */
GPS_API int gps_sh_bwd(int N, int K, int degrees_to_use, const float *dirs, const float *coeffs, const uint8_t *masks,
               const float *v_colors, float *v_coeffs, float *v_dirs, gps_stream stream);

/* ------------------------------------------------------------------ */
/* Splat: tile binning without depth key                               */
/* ------------------------------------------------------------------ */

/* Size in bytes of the scratch buffer gps_isect_tiles_no_depth needs for N Gaussians and
 * room for isect_capacity intersections. */
GPS_API int64_t gps_isect_workspace_bytes(int N, int64_t isect_capacity);

/* replaces gsplat::isect_tiles_tensor_no_depth + isect_offset_encode_tensor_no_depth
 * (isect_tiles_no_depth.cu:132-461) in ONE call without host synchronisation.
 * In : means2d[N,2], radii[N] (already clamped to max_gs_radii by the caller).
 * Out: tiles_per_gauss[N];
 *      isect_ids[isect_capacity] (int64 tile ids, sorted; may be NULL);
 *      flatten_ids[isect_capacity] (Gaussian ids, stably sorted by tile);
 *      group_gs_ids / group_starts [group_capacity] (32-pixel backward groups);
 *      tile_offsets[tile_h*tile_w];
 *      counts[4] (device int64): {n_isects, n_groups, overflow_flag, n_visible}.
 * If n_isects > isect_capacity or n_groups > group_capacity the excess is dropped, counts[2] is
 * set non-zero and counts[0]/[1] hold the clamped values. */
GPS_API int gps_isect_tiles_no_depth(int N, const float *means2d, const int32_t *radii, int tile_size, int tile_width,
                             int tile_height, int64_t isect_capacity, int64_t group_capacity,
                             int32_t *tiles_per_gauss, int64_t *isect_ids, int32_t *flatten_ids,
                             int32_t *group_gs_ids, int32_t *group_starts, int32_t *tile_offsets, int64_t *counts,
                             void *workspace, int64_t workspace_bytes, gps_stream stream);

/* ------------------------------------------------------------------ */
/* Splat: ges rasterizer                                               */
/* ------------------------------------------------------------------ */

/* replaces gsplat::rasterize_to_pixels_fwd_ges_tensor (rasterize_to_pixels_fwd_ges.cu:223-407),
 * COLOR_DIM = 4 (rgb + depth; channel 3 is the depth the cut is tested on), no backgrounds/masks.
 * n_isects is read from counts[0] on the device.
 * Out: render_colors[H,W,4], render_alphas[H,W] (= weight sum), last_ids[H,W] (may be NULL). */
GPS_API int gps_raster_ges_fwd(int N, const float *means2d, const float *conics, const float *colors,
                       const float *opacities, const float *ref_depth_map, int width, int height, int tile_size,
                       const int32_t *tile_offsets, const int32_t *flatten_ids, const int64_t *counts,
                       float delta_depth, float *render_colors, float *render_alphas, int32_t *last_ids,
                       gps_stream stream);

/* replaces gsplat::rasterize_to_pixels_bwd_ges_gs_parallel_tensor
 * (rasterize_to_pixels_bwd_ges_new_parallel.cu:203-385): Gaussian-parallel backward over the
 * 2r x 2r integer pixel box of every Gaussian.  n_groups is read from counts[1].
 * v_means2d[N,2] v_conics[N,3] v_colors[N,4] v_opacities[N] are zero-filled by this call and then
 * accumulated. */
GPS_API int gps_raster_ges_bwd_gs(int N, const float *means2d, const float *conics, const float *colors,
                          const float *opacities, const int32_t *radii, const float *ref_depth_map, int width,
                          int height, const int32_t *group_gs_ids, const int32_t *group_starts,
                          const int64_t *counts, float delta_depth, const float *v_render_colors,
                          const float *v_render_alphas, float *v_means2d, float *v_conics, float *v_colors,
                          float *v_opacities, gps_stream stream);

/* ------------------------------------------------------------------ */
/* Splat: compose + L1 loss (fused replacement of libtorch glue)       */
/* ------------------------------------------------------------------ */

/* Fuses raw_gs_model.cpp:318-326 (compose with the TSDF layer), :369-417 (L1 loss, ssim/depth
 * weights 0) and their autograd backward.
 *   rgb   = (raw_rgb + base_color) / (W + 1)
 *   depth = (raw_d + ref*[ref>0]) / (W + [ref>0])
 *   loss  = mean |gt - rgb|  over H*W*3
 * In : render_colors[H,W,4], weight_sum[H,W], base_color[H,W,3], ref_depth_raw[H,W], gt_rgb[H,W,3]
 * Out: rgb[H,W,3], depth[H,W] (may be NULL), loss[1] (device float, accumulated: caller zeroes it),
 *      v_render_colors[H,W,4], v_render_alphas[H,W] (NULL to skip the backward half). */
GPS_API int gps_compose_l1(int width, int height, const float *render_colors, const float *weight_sum,
                   const float *base_color, const float *ref_depth_raw, const float *gt_rgb, float *rgb,
                   float *depth, float *loss, float *v_render_colors, float *v_render_alphas, gps_stream stream);

/* ------------------------------------------------------------------ */
/* Splat: fused multi-tensor Adam                                      */
/* ------------------------------------------------------------------ */

#define GPS_ADAM_MAX_SEGMENTS 8
typedef struct {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int64_t numel;
    double lr; /* libtorch keeps lr/betas/eps as double and rounds once per use */
} gps_adam_segment;

/* One optimiser step over up to 8 parameter tensors, bit-compatible update order with libtorch's
 * torch::optim::Adam (raw_gs_model.cpp:654-705; eps 1e-15, betas (0.9,0.999), no weight decay):
 *   m = m*b1 + g*(1-b1); v = v*b2 + g*g*(1-b2);
 *   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * step is the 1-based step count t. */
GPS_API int gps_adam_step(const gps_adam_segment *segments, int n_segments, double beta1, double beta2, double eps, int step,
                  gps_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* GPS_SLAM_HIP_H */

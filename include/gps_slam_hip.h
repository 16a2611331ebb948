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
/* Compile-time probe switches / tunables (-DGPS_...) this library was built with that differ from the shipped defaults, as
 * "NAME=value NAME=value"; "" for the shipped build (asserted by tests/test_abi_cpu.py and __graft_entry__.smoke()): A/B
 * builds made with tools/probe/variant.py identify themselves. */
GPS_API const char *gps_build_flags(void);

/* In-loop launch timing -- measurement support for bench.py's `roofline` ("achieved = algorithmic bytes / that
 * kernel's average launch duration, measured live with HIP events on the stream the kernel is launched on").  Between _start and
 * _stop every launch of the kernels below, from whichever entry point, host thread and stream, stamps the device's 100 MHz
 * wall clock at the start and at the end of each of its waves into its workgroup's {first start, last end} slot
 * (csrc/launch_timing.hpp; a launch's interval = last end - first start over its workgroups):
 * the kernel's own execution interval, as a rocprofv3 kernel trace reports it minus the dispatch's ramp-up and the end-of-kernel
 * release; _stop synchronises the device, reads the slots and sums per kind.  `flagged` = the subset with the kind's flag set
 * (PREPROCESS_BWD: the next iteration's preprocessing forward rode in the launch; RASTER_FWD: with the compose + L1 epilogue;
 * RAYCAST: a free view).  Results never depend on it; off by default and after _stop.  capacity = workgroup slots (16 bytes
 * each) a window can hold; launches that no longer fit run unstamped and are counted in *dropped.  _start is the one entry point
 * of this library that allocates (the slot buffer, hipMalloc, kept for the next window) and, with _stop, that synchronises the
 * device: measurement calls, outside any timed region.  No reference counterpart (the reference times stages on the host,
 * slam_pipeline.cpp:135-167). */
#define GPS_TIMED_PREPROCESS_BWD 0
#define GPS_TIMED_PREPROCESS_FWD 1
#define GPS_TIMED_RASTER_FWD 2
#define GPS_TIMED_RASTER_BWD_STRIPS 3
#define GPS_TIMED_SB_SCAN 4
#define GPS_TIMED_SB_SCATTER 5
#define GPS_TIMED_INTEGRATE 6
#define GPS_TIMED_RAYCAST 7
#define GPS_TIMED_KINDS 8
GPS_API int gps_launch_timing_start(int capacity);
GPS_API int gps_launch_timing_stop(void);
GPS_API int gps_launch_timing_read(int kind, double *total_us, int64_t *launches, double *total_us_flagged,
                                   int64_t *launches_flagged, double *max_us, int64_t *dropped);

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
/* Zero-fills a binning workspace.  Call once after allocating (or re-allocating) the blob handed to gps_splat_render /
 * gps_splat_train_step: the fused path's count tables are kept zero BETWEEN launches by the kernels themselves. */
GPS_API int gps_isect_workspace_init(void *workspace, int64_t workspace_bytes, gps_stream stream);

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
 * set non-zero and counts[0]/[1] hold the clamped values.  counts[2] is sticky: the library only ever raises it (the
 * caller zeroes counts before the first call and after handling an overflow), so a host that reads it once per
 * keyframe still learns of an overflow in any launch in between. */
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

/* Same operator (rasterize_to_pixels_fwd_ges.cu:223-407) fed by the packed 48-byte records gps_gauss_preprocess_fwd writes
 * instead of the four per-Gaussian arrays: the forward the fused model path (gps_splat_render / gps_splat_train_step) runs.
 * 256-record LDS batches, two pixels per lane with packed fp32 math, alpha = exp2 of a pre-scaled exponent; agrees with
 * gps_raster_ges_fwd to float rounding (tests/test_splat_gpu.py). */
GPS_API int gps_raster_ges_fwd_rec(int N, const float *records, const float *ref_depth_map, int width, int height,
                                   const int32_t *tile_offsets, const int32_t *flatten_ids, const int64_t *counts,
                                   float delta_depth, float *render_colors, float *render_alphas, gps_stream stream);
/* The same launch with its tile workgroups dealt in a given order (tile_order[tile_width * tile_height]: a permutation of the
 * tile ids; NULL = row-major).  The fused path passes the tiles by descending list length, which the superblock binning leaves in
 * its workspace (gps_isect_workspace_tile_order: a device pointer into `workspace`, valid after gps_splat_render /
 * gps_splat_train_step with the strip buffers present): the chip holds 1,024 tile workgroups, 640x480 has 1,200 -- the part-filled
 * second round should be the short lists.  Same output. */
GPS_API int gps_raster_ges_fwd_rec_ordered(int N, const float *records, const float *ref_depth_map, int width, int height,
                                           const int32_t *tile_offsets, const int32_t *flatten_ids, const int64_t *counts,
                                           float delta_depth, float *render_colors, float *render_alphas,
                                           const int32_t *tile_order, gps_stream stream);
GPS_API const int32_t *gps_isect_workspace_tile_order(void *workspace, int N, int64_t isect_capacity);

/* replaces gsplat::rasterize_to_pixels_bwd_ges_gs_parallel_tensor
 * (rasterize_to_pixels_bwd_ges_new_parallel.cu:203-385): Gaussian-parallel backward over the
 * 2r x 2r integer pixel box of every Gaussian.  n_groups is read from counts[1].
 * v_means2d[N,2] v_conics[N,3] v_colors[N,4] v_opacities[N]: accumulate == 0 -> zero-filled by this call and then
 * accumulated (the reference's zeros_like + atomicAdd); accumulate != 0 -> added to what the buffers hold. */
GPS_API int gps_raster_ges_bwd_gs(int N, const float *means2d, const float *conics, const float *colors,
                          const float *opacities, const int32_t *radii, const float *ref_depth_map, int width,
                          int height, const int32_t *group_gs_ids, const int32_t *group_starts,
                          const int64_t *counts, float delta_depth, const float *v_render_colors,
                          const float *v_render_alphas, float *v_means2d, float *v_conics, float *v_colors,
                          float *v_opacities, int accumulate, gps_stream stream);

/* The same operator (rasterize_to_pixels_bwd_ges_new_parallel.cu:18-201: every pixel of the 2r x 2r box, the same per-pixel
 * arithmetic) in the decomposition the fused model path runs: a Gaussian owns 4 / 8 / 16 / 32 / 64 adjacent lanes (class k:
 * the smallest 4 << k >= r; class 4 covers every larger radius), each lane walks two columns of the box row by row, the ten
 * gradient totals are reduced once per Gaussian and written as ONE 48-byte row
 *     v_rows[id] = { v_colors[4], v_conics[3], v_means2d[2], v_opacity, 0, 0 }
 * -- a plain store per Gaussian of the class lists: no atomics, no zero-fill.  Rows of Gaussians that are in no list are not
 * touched.  records: gps_gauss_preprocess_fwd's 48-byte records; radii: the clamped radii binning used;
 * cls_ids[GPS_BWD_CLASSES][cls_stride]: ascending Gaussian ids per class, cls_counts[GPS_BWD_CLASSES] on the device
 * (written by the binning of gps_splat_train_step, or by gps_raster_bwd_classes for arrays that come from elsewhere);
 * v_render_colors[P,4]; pix2[P,2] = {d loss / d weight_sum, ref_depth + delta_depth} per pixel (gps_raster_pair_image, or the
 * fused forward's epilogue). */
#define GPS_BWD_CLASSES 5
GPS_API int gps_raster_ges_bwd_strips(int N, const float *records, const int32_t *radii, const int32_t *cls_ids,
                                      const int32_t *cls_counts, int cls_stride, const float *v_render_colors,
                                      const float *pix2, int width, int height, float *v_rows, gps_stream stream);
/* Scheduling hint, process-wide (no reference counterpart: the reference runs tracking and mapping one after the other).  on != 0:
 * gps_raster_ges_bwd_strips / gps_splat_train_step launch the strip backward with 28 KB of unused dynamic LDS per workgroup, so
 * that 5 instead of 6 of its workgroups share a compute unit and one workgroup of the tracker's pre-launched evaluation (112
 * VGPRs per wave) always finds room beside it.  For hosts that run a frame chain (tracking / fusion) on one stream WHILE the map
 * update runs on another (host/slam_pipeline.cpp: overlap_mapping); a host that runs them in turn leaves it off -- the strip
 * kernel alone is faster with 6.  The same switch picks the forward rasterizer's launch order inside gps_splat_render /
 * gps_splat_train_step: off -> tiles by descending list length (the chip holds 1,024 of the 1,200 tile workgroups of 640x480: the
 * part-filled second round is then the short lists; iteration 257 -> 248 us), on -> row-major (beside a frame chain the ordered
 * launch is 0.7 % slower).  Results do not depend on it. */
GPS_API void gps_set_frame_chain_reserve(int on);
/* Experiment switch (round 6): the record forward (gps_raster_ges_fwd_rec*, gps_splat_render, gps_splat_train_step) as a persistent
 * launch -- 3 workgroups per compute unit walk a static, cost-balanced share of the tiles and stage the next batch's records
 * while the current one is evaluated (csrc/splat_raster.hip: raster_ges_fwd_pp_kernel).  OFF by default: bit-identical images
 * (tests/test_splat_gpu.py) but 51 against 43 us at 640x480 -- the forward is bound by its VALU work and one synchronised
 * start-up, not by the later tiles' load chain (LABBOOK.md).  1 = on, for that test and for A/B timing. */
GPS_API void gps_set_raster_fwd_persistent(int on);
/* records[N,12] (the 48-byte records gps_gauss_preprocess_fwd writes, incl. the ellipse bounds) from the operator-level arrays
 * means2d[N,2] conics[N,3] colors[N,4] (rgb + depth) opacities[N] radii[N]: lets a caller that holds those run the strip backward */
GPS_API int gps_raster_pack_records(int N, const float *means2d, const float *conics, const float *colors,
                                    const float *opacities, const int32_t *radii, float *records, gps_stream stream);
GPS_API int gps_raster_pair_image(int width, int height, const float *v_render_alphas, const float *ref_depth_map,
                                  float delta_depth, float *pix2, gps_stream stream);

/* replaces gsplat::rasterize_to_pixels_bwd_ges_tensor (rasterize_to_pixels_bwd_ges.cu:18-291): the exact tile-parallel
 * adjoint of gps_raster_ges_fwd, what the reference's RasterizeToPixelsGes autograd Function runs (gsplat_wapper.hpp:355-487;
 * the shipped models use the Gaussian-parallel box backward above).  n_isects is read from counts[0].
 * Out (zero-filled by this call, then accumulated): v_means2d[N,2] v_conics[N,3] v_colors[N,4] v_opacities[N]. */
GPS_API int gps_raster_ges_bwd_exact(int N, const float *means2d, const float *conics, const float *colors,
                                     const float *opacities, const float *ref_depth_map, int width, int height,
                                     int tile_size, const int32_t *tile_offsets, const int32_t *flatten_ids,
                                     const int64_t *counts, float delta_depth, const float *v_render_colors,
                                     const float *v_render_alphas, float *v_means2d, float *v_conics, float *v_colors,
                                     float *v_opacities, gps_stream stream);

/* ------------------------------------------------------------------ */
/* Splat: fused SSIM map (the `ssim_weight > 0` loss option)            */
/* ------------------------------------------------------------------ */

/* replaces fusedssim (gsplat/rasterizer/ssim.cu:385-421 launcher, :209-303 kernel): per-channel SSIM map of img1 vs img2
 * with the 11-tap Gaussian window (sigma 1.5), zero padding.  C1 = 0.01^2, C2 = 0.03^2 in raw_gs_model.cpp:388-389.
 * channels_last = 0: [B,CH,H,W] planar, the reference's layout; 1: [B,H,W,CH] interleaved (how renders and camera images are
 * held -- no permute copy).  All maps use the same layout as the images.  dm_* (all three or none, NULL when not
 * training) receive the partial derivatives the backward needs. */
GPS_API int gps_ssim_fwd(int B, int CH, int H, int W, int channels_last, float C1, float C2, const float *img1,
                         const float *img2, float *ssim_map, float *dm_dmu1, float *dm_dsigma1_sq, float *dm_dsigma12,
                         gps_stream stream);

/* replaces fusedssim_backward (ssim.cu:423-460, :305-383): dL_dimg1 from dL_dmap and the saved maps (every element is
 * written; the reference's `padding == "valid"` crop is the caller's zero border in dL_dmap, gsplat_wapper.hpp:664-669). */
GPS_API int gps_ssim_bwd(int B, int CH, int H, int W, int channels_last, const float *img1, const float *img2,
                         const float *dL_dmap, const float *dm_dmu1, const float *dm_dsigma1_sq, const float *dm_dsigma12,
                         float *dL_dimg1, gps_stream stream);

/* ------------------------------------------------------------------ */
/* Splat: `raw` render method (front-to-back alpha compositing)         */
/* ------------------------------------------------------------------ */

/* replaces isectTiles + isectOffsetEncode (gsplat_wapper.cpp:3-48, isect_tiles.cu:30-430): depth-keyed binning.  Every
 * tile's list comes out ordered by (depth, Gaussian index) -- the order a stable sort of the reference's
 * (tile_id << 32 | bits(depth)) keys gives.  isect_ids (optional) receives exactly those 64-bit keys, sorted.
 * counts / workspace / capacities as gps_isect_tiles_no_depth (gps_isect_workspace_bytes sizes the workspace for both). */
GPS_API int gps_isect_tiles(int N, const float *means2d, const int32_t *radii, const float *depths, int tile_size,
                            int tile_width, int tile_height, int64_t isect_capacity, int32_t *tiles_per_gauss,
                            int64_t *isect_ids, int32_t *flatten_ids, int32_t *tile_offsets, int64_t *counts,
                            void *workspace, int64_t workspace_bytes, gps_stream stream);

/* replaces rasterize_to_pixels_fwd_tensor (rasterize_to_pixels_fwd.cu:18-376), COLOR_DIM = 4 (rgb + depth), one camera,
 * no tile masks: front-to-back compositing with transmittance T, a pixel stops when T * (1 - alpha) <= 1e-4.
 * backgrounds: device float[4] or NULL.  Out: render_colors[H,W,4] (+ T * background), render_alphas[H,W] = 1 - T,
 * last_ids[H,W] = position (in flatten_ids) of the last Gaussian that contributed. */
GPS_API int gps_raster_raw_fwd(int N, const float *means2d, const float *conics, const float *colors,
                               const float *opacities, const float *backgrounds, int width, int height, int tile_size,
                               const int32_t *tile_offsets, const int32_t *flatten_ids, const int64_t *counts,
                               float *render_colors, float *render_alphas, int32_t *last_ids, gps_stream stream);

/* replaces rasterize_to_pixels_bwd_tensor (rasterize_to_pixels_bwd.cu:20-511): per pixel back to front from last_ids.
 * Out (zero-filled by this call): v_means2d[N,2], v_conics[N,3], v_colors[N,4], v_opacities[N]; v_means2d_abs[N,2] if
 * not NULL (absgrad). */
GPS_API int gps_raster_raw_bwd(int N, const float *means2d, const float *conics, const float *colors,
                               const float *opacities, const float *backgrounds, int width, int height, int tile_size,
                               const int32_t *tile_offsets, const int32_t *flatten_ids, const int64_t *counts,
                               const float *render_alphas, const int32_t *last_ids, const float *v_render_colors,
                               const float *v_render_alphas, float *v_means2d_abs, float *v_means2d, float *v_conics,
                               float *v_colors, float *v_opacities, gps_stream stream);

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
 *      v_render_colors[H,W,4], v_render_alphas[H,W] (NULL to skip the backward half).
 * gt_rgb == NULL renders only (rgb/depth), as the reference's NoGradGuard call sites do. */
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

/* Zero-fills n (<= 16) float buffers in one launch -- a fresh torch::optim::Adam's exp_avg / exp_avg_sq for every parameter
 * (raw_gs_model.cpp:654-675 builds new optimizers at every localOptimize).  ptrs / numels are HOST arrays; each buffer is
 * 16-byte aligned and numels[k] is a multiple of 4 (round the length up inside a capacity-sized buffer). */
GPS_API int gps_zero_floats(int n, float *const *ptrs, const int64_t *numels, gps_stream stream);

/* One optimiser step over up to 8 parameter tensors, bit-compatible update order with libtorch's
 * torch::optim::Adam (raw_gs_model.cpp:654-705; eps 1e-15, betas (0.9,0.999), no weight decay):
 *   m = m*b1 + g*(1-b1); v = v*b2 + g*g*(1-b2);
 *   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * step is the 1-based step count t.  step == 1 is the first step of a freshly created optimizer: libtorch creates exp_avg /
 * exp_avg_sq as zeros then, and so do these kernels -- the two buffers are WRITTEN but NOT READ at step 1 (here, in
 * gps_gauss_preprocess_bwd_adam and in gps_splat_train_step), whatever they hold: a host that re-creates its optimizers need not
 * zero them (gps_zero_floats) first. */
GPS_API int gps_adam_step(const gps_adam_segment *segments, int n_segments, double beta1, double beta2, double eps, int step,
                  gps_stream stream);

/* ------------------------------------------------------------------ */
/* Splat: fused per-Gaussian pre-/post-processing (model level)        */
/* ------------------------------------------------------------------ */

/* Everything RawGaussianModel::gesForward does per Gaussian before binning (src/raw_gs_model.cpp:207-286) in one
 * pass: exp(log_scales) -> projection -> radii clamp (max_gs_radii, 0 = off) -> dirs = means - cam_pos ->
 * SH (coefficients given as the model stores them: sh_dc[N,3] + sh_rest[N,K-1,3], no torch::cat) ->
 * colors[N,4] = {clamp_min(sh + 0.5, 0), depth} -> opacities[N] = sigmoid(opac_logit).
 * viewmat[16], Kmat[9], cam_pos[3] are device arrays.  records (optional, [N,12] floats): the packed per-Gaussian
 * record {xy, conic, opac, depth, rgb, int16 pixel bounds of the alpha >= 1/255 box} gps_raster_ges_fwd_rec streams. */
GPS_API int gps_gauss_preprocess_fwd(int N, int K, int sh_degree, const float *means, const float *log_scales,
                                     const float *quats, const float *opac_logit, const float *sh_dc,
                                     const float *sh_rest, const float *viewmat, const float *Kmat,
                                     const float *cam_pos, int width, int height, float eps2d, float near_plane,
                                     float far_plane, float radius_clip, int max_gs_radii, int32_t *radii,
                                     float *means2d, float *depths, float *conics, float *colors, float *opacities,
                                     float *records, gps_stream stream);

/* Adjoint of gps_gauss_preprocess_fwd == the libtorch autograd chain cat/clamp_min/SH/projection/exp/sigmoid
 * backward (gsplat_wapper.hpp:56-95,156-240 + ATen), writing all six parameter gradients in one pass. */
GPS_API int gps_gauss_preprocess_bwd(int N, int K, int sh_degree, const float *means, const float *log_scales,
                                     const float *quats, const float *opac_logit, const float *sh_dc,
                                     const float *sh_rest, const float *viewmat, const float *Kmat,
                                     const float *cam_pos, int width, int height, float eps2d, const int32_t *radii,
                                     const float *conics, const float *v_means2d, const float *v_conics,
                                     const float *v_colors, const float *v_opacities, float *v_means,
                                     float *v_log_scales, float *v_quats, float *v_opac_logit, float *v_sh_dc,
                                     float *v_sh_rest, gps_stream stream);

/* gps_gauss_preprocess_bwd with the Adam step of sh_rest applied in the same kernel (what gps_splat_train_step runs when
 * gps_splat_step.fuse_sh_rest_adam is set): sh_rest, exp_avg, exp_avg_sq [N,K-1,3] are updated in place exactly as
 * gps_adam_step would with the gradient gps_gauss_preprocess_bwd computes (bit-identical); v_sh_rest may be NULL (the
 * gradient then never leaves the chip) or a buffer to also receive it.  Other gradients as gps_gauss_preprocess_bwd.
 * small5 (optional): five segments {means, log_scales, quats, sh_dc, opac_logit} (param = the arrays passed above, with
 * exp_avg / exp_avg_sq / lr; .grad unused) to be stepped in the same kernel as well -- no separate Adam launch is needed
 * then, and the five v_* outputs may be NULL. */
GPS_API int gps_gauss_preprocess_bwd_adam(int N, int K, int sh_degree, const float *means, const float *log_scales,
                                          const float *quats, const float *opac_logit, const float *sh_dc,
                                          float *sh_rest, const float *viewmat, const float *Kmat, const float *cam_pos,
                                          int width, int height, float eps2d, const int32_t *radii, const float *conics,
                                          const float *v_means2d, const float *v_conics, const float *v_colors,
                                          const float *v_opacities, float *v_means, float *v_log_scales, float *v_quats,
                                          float *v_opac_logit, float *v_sh_dc, float *v_sh_rest, float *exp_avg,
                                          float *exp_avg_sq, double lr, const gps_adam_segment *small5, double beta1,
                                          double beta2, double eps, int step, gps_stream stream);

/* ------------------------------------------------------------------ */
/* Splat: Gaussian creation helpers (every local_opt_interval frames)  */
/* ------------------------------------------------------------------ */

/* replaces distCUDA2 (gsplat/rasterizer/simple_knn.cu:191-240; called from raw_gs_param.cpp:28):
 * mean_dist2[i] = mean of the squared distances from points[i] to its 3 nearest other points (FLT_MAX terms if
 * P < 4, as in the reference).  Exact, no scratch memory, no host sync. */
GPS_API int gps_knn_mean_dist2(int P, const float *points, float *mean_dist2, gps_stream stream);

/* The same numbers, bit for bit, in sub-quadratic time for large P (the reference prunes with Morton-ordered boxes,
 * simple_knn.cu:67-227; here: counting sort into a uniform grid over the bounding box, then an exact ring-by-ring search per
 * point).  workspace: gps_knn_grid_workspace_bytes(P) bytes of device memory, 16-byte aligned, no initialisation needed, free
 * for other use between calls.  Eight short launches, no host sync.  MI355X: 90 us at 76,800 points (brute force 2.4 ms), 217 us at 230,400 (17.6 ms). */
#define GPS_KNN_GRID_MIN_POINTS 8192 /* below: the brute force (one launch) is faster than the grid's eight (MI355X: 73 vs 65 us at 8,000 points) */
GPS_API int64_t gps_knn_grid_workspace_bytes(int P);
GPS_API int gps_knn_mean_dist2_grid(int P, const float *points, float *mean_dist2, void *workspace, int64_t workspace_bytes,
                                    gps_stream stream);

/* The sample mask of SLAMPipeline::initNewGaussians (slam/slam_pipeline.cpp:450-526) in one launch instead of ~12 tensor ops:
 *   valid = depth in (depth_vis_min, depth_vis_max) and vertex.sum(-1) != 0
 *   mask  = mean(|src_rgb - image|, -1) > color_error_thres  and  valid  [and alpha < alpha_vis_max, if alpha != NULL]
 * src_rgb: what the colour error is measured on -- the raycast colour for an empty model, the render otherwise.  All maps
 * [H,W,c] float32 device; mask [H,W] bytes (a torch bool tensor).  Float sequence as ATen's 3-element reductions
 * (sum = (x+z)+y, mean = ((a+c)+b) * RN(1/3)), so that pixels whose error lands on the threshold decide as the tensor ops do. */
GPS_API int gps_new_gaussian_mask(int width, int height, const float *depth_map, const float *src_rgb, const float *image,
                                  const float *vertex_map, const float *alpha, float depth_vis_min, float depth_vis_max,
                                  float color_error_thres, float alpha_vis_max, uint8_t *mask, gps_stream stream);

/* torch::masked_select's selection order (slam/slam_gs_model.cpp:14-19) as indices: ids[0..count) = positions of the
 * non-zero bytes of mask[0..n) in ascending order; count[0] (device) and, if host_count != NULL (pinned, device-visible
 * host memory), host_count[0] receive the number.  Two launches over 4096-byte blocks (workspace: one int per block), no host
 * synchronisation inside. */
GPS_API int64_t gps_compact_mask_workspace_bytes(int n);
GPS_API int gps_compact_mask(int n, const uint8_t *mask, int32_t *ids, int32_t *count, int32_t *host_count, void *workspace,
                             int64_t workspace_bytes, gps_stream stream);

/* Rows ids[subset[j]], j < k, of three [P,3] maps -> three [k,3] arrays (the index_select of the sampled subset,
 * slam_gs_model.cpp:27-33, without materialising the three masked_select results first). */
GPS_API int gps_gather_pixels(int k, const int32_t *ids, const int32_t *subset, const float *vertex_map, const float *image,
                              const float *normal_map, float *verts, float *cols, float *norms, gps_stream stream);

/* RawGaussianParams::init (src/raw_gs_param.cpp:11-74) for k points in one launch instead of ~35 tensor ops: from points,
 * colours, optional normals (NULL: identity-like quaternion of ones, isotropic scale) and the KNN mean squared distances
 * (gps_knn_mean_dist2) to the six parameter tensors, written to the rows the output pointers name (e.g. the tail of
 * capacity-sized buffers): means[k,3], log_scales[k,3] = log(clamp(sqrt(knn), min_scale, max_scale)) (z x 0.1 with normals),
 * quats[k,4] = rotation of (0,0,1) onto the normal (tensor_math.cpp:184-201), sh_dc[k,3] = (rgb - 0.5) / C0,
 * sh_rest[k,K-1,3] = 0, opac_logit[k] = logit(init_opacity). */
GPS_API int gps_init_gaussians(int k, const float *xyz, const float *rgb, const float *normals, const float *knn_mean_dist2,
                               int K, float init_opacity, float max_scale, float min_scale, float *means, float *log_scales,
                               float *quats, float *sh_dc, float *sh_rest, float *opac_logit, gps_stream stream);

/* The delete mask of SLAMPipeline::removeRedundantGs (slam/slam_pipeline.cpp:564-586) in one launch:
 * delete = max(exp(log_scales)) < small or > large, or sigmoid(opac_logit) < low_opac; keep = !delete (both [N] bytes). */
GPS_API int gps_prune_mask(int N, const float *log_scales, const float *opac_logit, float small_scale_thres,
                           float large_scale_thres, float low_opac_thres, uint8_t *delete_mask, uint8_t *keep_mask,
                           gps_stream stream);

/* prunePoints' index_select (src/raw_gs_model.cpp:635-644) for up to 8 row-major float tensors in ONE launch:
 * dsts[t][j, :] = srcs[t][ids[j], :], j < m, rows of row_floats[t] floats.  srcs / dsts / row_floats are HOST arrays;
 * dsts must not alias srcs. */
GPS_API int gps_gather_rows(int m, const int32_t *ids, int n_tensors, const float *const *srcs, float *const *dsts,
                            const int32_t *row_floats, gps_stream stream);

/* replaces computeNormalMap (src/tensor_math.cpp:278-300 + featureGradient :217-248): vertex_map[H,W,3] ->
 * normal_map[H,W,3] (Sobel, replicate padding, cross(dy,dx) normalised, 0 where vertex z <= 0). */
GPS_API int gps_normal_map(int width, int height, const float *vertex_map, float *normal_map, gps_stream stream);

/* uchar4 frame -> float image: rgb[p, c] = rgba[p, c] * (1/255), c < 3 (what Camera::toGPU's image / 255 amounts to when
 * the frame is already in HBM as the uchar4 image UpdateView uploaded: 3 of its 4 bytes per pixel instead of a second
 * 12-byte-per-pixel upload). */
GPS_API int gps_rgba8_to_rgbf(int n_pixels, const uint8_t *rgba, float *rgb, gps_stream stream);
/* ... and gps_upload_floats(floats_dst, host_values, n_floats) in the SAME launch: the two things curr_cam.toGPU()
 * (slam/slam_pipeline.cpp:84) puts into HBM for a frame -- its image and its pose / intrinsics -- as one kernel. */
GPS_API int gps_rgba8_to_rgbf_and_floats(int n_pixels, const uint8_t *rgba, float *rgb, float *floats_dst,
                                         const float *host_values, int n_floats, gps_stream stream);

/* Uploads up to 64 floats from host memory into device memory THROUGH THE KERNEL ARGUMENT BUFFER (values are read on the host at
 * call time; no pinned staging, no copy-engine transfer, ordered on `stream` like any kernel).  Camera::toGPU() uses it for the
 * 28-float viewmat | K | camera position pack of every frame: a hipMemcpyAsync of that size costs ~15 us of copy-engine latency
 * on the frame stream's dependent chain. */
GPS_API int gps_upload_floats(float *dst, const float *host_values, int n, gps_stream stream);

/* ------------------------------------------------------------------ */
/* Splat: one optimise iteration / one render as a single call         */
/* ------------------------------------------------------------------ */

/* Everything one iteration of SLAMPipeline::localOptimize touches (slam/slam_pipeline.cpp:247-254).  All pointers
 * are device memory owned by the caller; sizes follow the op-level entry points above. */
typedef struct {
    int32_t N, K, sh_degree, width, height, max_gs_radii;
    float eps2d, near_plane, far_plane, radius_clip, delta_depth;
    /* parameters (updated in place by the Adam step) */
    float *means, *log_scales, *quats, *opac_logit, *sh_dc, *sh_rest;
    /* camera (device): viewmat[16] row-major world->camera, Kmat[9], cam_pos[3] */
    const float *viewmat, *Kmat, *cam_pos;
    /* per-camera images: TSDF raycast depth clamped (ref < 0.01 -> 1000, raw_gs_model.cpp:207), raycast colour,
     * ground-truth image */
    const float *ref_depth_clamped, *base_color, *gt_rgb;
    /* per-Gaussian intermediates */
    int32_t *radii;
    float *means2d, *depths, *conics, *colors, *opacities, *records;
    /* binning */
    int64_t isect_capacity, group_capacity, workspace_bytes;
    int32_t *tiles_per_gauss, *flatten_ids, *group_gs_ids, *group_starts, *tile_offsets;
    int64_t *counts;
    void *workspace;
    /* images */
    float *render_colors, *weight_sum, *rgb, *loss, *v_render_colors, *v_render_alphas;
    /* rasterizer gradients */
    float *v_means2d, *v_conics, *v_colors, *v_opacities;
    /* parameter gradients and Adam state, same order as the parameters */
    float *g_means, *g_log_scales, *g_quats, *g_opac_logit, *g_sh_dc, *g_sh_rest;
    float *m_means, *m_log_scales, *m_quats, *m_opac_logit, *m_sh_dc, *m_sh_rest;
    float *v_means, *v_log_scales, *v_quats, *v_opac_logit, *v_sh_dc, *v_sh_rest;
    double lr[6]; /* means, log_scales, quats, sh_dc, sh_rest, opac_logit */
    double beta1, beta2, adam_eps;
    /* 0: every gradient is written and all six tensors are stepped by the multi-tensor Adam kernel.
     * 1: the Adam step of sh_rest (45 of the 59 parameters) happens inside the preprocessing backward kernel -- its gradient
     *    never goes to HBM, g_sh_rest is NOT written; the other five tensors go through the Adam kernel.
     * 2: all six tensors are stepped inside the backward kernel (no Adam launch; no g_* is written).
     * The parameter update is bit-identical in all three modes. */
    int32_t fuse_sh_rest_adam;
    /* Strip backward (gps_raster_ges_bwd_strips).  All five set: the train step bins with the superblock counting sort (which
     * also writes the class lists), its forward epilogue writes pix2, its backward rasterizer writes v_rows and the
     * preprocessing backward reads them; group_gs_ids / group_starts / v_means2d.. are then not touched.  Any NULL (or more
     * than 4096 tiles): the sorted-key binning + the group kernel.
     * v_rows[capacity,12]  pix2[H*W,2]  cls_ids[GPS_BWD_CLASSES, cls_stride]  cls_counts[8] (int32, device) */
    float *v_rows, *pix2;
    int32_t *cls_ids, *cls_counts;
    int64_t cls_stride;
    /* Cross-iteration fusion inside an optimise loop (optional; gps_splat_can_prefetch() says whether this step supports it).
     * next_viewmat / next_Kmat / next_cam_pos != NULL in gps_splat_train_step: the camera of the NEXT iteration -- its
     * preprocessing forward (projection, SH, records, the binning's count pass) runs in the tail of this iteration's backward +
     * Adam kernel, on the parameters that kernel has just stepped and still holds, instead of re-reading them in a launch of its
     * own.  The caller then sets `preprocessed` != 0 on that next call (same N, same camera arrays, nothing else run on these
     * buffers in between): its preprocessing launch is skipped.  Same arithmetic on the same values as the separate launch. */
    const float *next_viewmat, *next_Kmat, *next_cam_pos;
    int32_t preprocessed;
} gps_splat_step;

/* != 0: gps_splat_train_step(a) can run the next iteration's preprocessing in its tail (strip backward + superblock binning in
 * use, all six tensors stepped inside the backward kernel, K > 1, a workgroup / LDS tile that holds the binning's histogram). */
GPS_API int gps_splat_can_prefetch(const gps_splat_step *a);
/* A prefetched forward that will NOT be consumed (the caller runs something else on these buffers before the train step it was
 * meant for) has left its counts in the binning's persistent tables: this clears them (two memsets).  Hosts call it when they
 * disarm a pending prefetch; a prefetch consumed by the next gps_splat_train_step(preprocessed != 0) needs nothing. */
GPS_API int gps_splat_discard_prefetch(const gps_splat_step *a, gps_stream stream);

/* gesForward up to the rasterizer (preprocess -> binning -> ges forward): fills render_colors / weight_sum. */
GPS_API int gps_splat_render(const gps_splat_step *a, gps_stream stream);

/* forward + L1 loss + backward + fused Adam step number `adam_step` (1-based; 1 = fresh optimizers: the moment buffers are
 * written, not read -- see gps_adam_step); `loss` must be zeroed by the caller. */
GPS_API int gps_splat_train_step(const gps_splat_step *a, int adam_step, gps_stream stream);

/* ------------------------------------------------------------------ */
/* TSDF: voxel-block-hash fusion and raycast (InfiniTAM ITMLib path)   */
/* ------------------------------------------------------------------ */

/* Storage layouts are the reference's (sizes checked by compiling it: 8 and 16 bytes). */
typedef struct {            /* ITMLib/Objects/Scene/ITMVoxelTypes.h:41-69 (ITMVoxel_s_rgb) */
    int16_t sdf;            /* (short)(f * 32767), initial 32767 */
    uint8_t w_depth;
    uint8_t clr[3];
    uint8_t w_color;
    uint8_t pad_;
} gps_voxel;

typedef struct {            /* ITMLib/Objects/Scene/ITMVoxelBlockHash.h:36-48 (ITMHashEntry) */
    int16_t pos[3];
    int16_t pad_;
    int32_t offset;         /* 1-based link into the excess list, 0 = none */
    int32_t ptr;            /* >= 0 voxel block, -1 swapped out, -2 empty */
} gps_hash_entry;

/* indices into gps_tsdf_state.counters (device int32[16]) */
enum {
    GPS_TSDF_LAST_FREE_BLOCK = 0,  /* ITMLocalVBA::lastFreeBlockId */
    GPS_TSDF_LAST_FREE_EXCESS = 1, /* ITMVoxelBlockHash::lastFreeExcessListId */
    GPS_TSDF_N_VISIBLE = 2,        /* renderState_live->noVisibleEntries */
    GPS_TSDF_N_VISIBLE_FREE = 3,   /* renderState_freeview->noVisibleEntries */
    GPS_TSDF_RENDER_BLOCKS = 4,    /* rendering blocks requested by the last CreateExpectedDepths */
    GPS_TSDF_OVERFLOW = 5,         /* non-zero: MAX_RENDERING_BLOCKS (262144) exceeded -> min/max image not reference-exact */
    GPS_TSDF_SCRATCH0 = 6,
    GPS_TSDF_SCRATCH1 = 7,
    GPS_TSDF_SCRATCH2 = 8,         /* rendering blocks of the CreateExpectedDepths in flight (published to [4], then cleared) */
    /* ray statistics (S-bar, the mean steps per ray): three unsigned 64-bit sums in words [10..15] -- castRay steps
     * as the reference's loop counts them (ITMVisualisationEngine_Shared.h:158-190), voxel reads of the kernel's own loop (it folds
     * runs of unallocated steps into one trip), rays cast -- of the LAST raycast launch on the scene's scratch (live or single free
     * view).  The raycaster writes one row per wave into the scratch area (plain stores); gps_tsdf_ray_stats() sums the rows
     * into these words on demand (measurement only). */
    GPS_TSDF_RAY_STEPS = 10,
    GPS_TSDF_RAY_READS = 12,
    GPS_TSDF_RAYS = 14,
    GPS_TSDF_N_COUNTERS = 16
};

/* All pointers are device memory owned by the caller (the reference owns the same buffers through
 * ORUtils::MemoryBlock: ITMLocalVBA, ITMVoxelBlockHash, ITMRenderState_VH, ITMView, ITMTrackingState). */
typedef struct {
    /* geometry / parameters (ITMSceneParams, ITMIntrinsics::projectionParamsSimple) */
    int32_t width, height;
    float fx, fy, cx, cy;
    float voxel_size, mu, view_frustum_min, view_frustum_max;
    int32_t max_w;                   /* 100, ITMLibSettings.cpp:10 */
    /* capacities (reference: 0x40000 blocks, 0x100000 buckets (power of two), 0x20000 excess) */
    int32_t n_blocks, n_buckets, n_excess;
    /* scene */
    gps_voxel *vba;                  /* [n_blocks * 512] */
    int32_t *vba_alloc_list;         /* [n_blocks] */
    gps_hash_entry *hash;            /* [n_buckets + n_excess] */
    int32_t *excess_list;            /* [n_excess] */
    int32_t *counters;               /* [16] */
    /* allocation scratch */
    uint32_t *alloc_prio;            /* [n_buckets + n_excess], zero between calls */
    int32_t *scan_scratch;           /* gps_tsdf_scratch_bytes(width, height, n_buckets, n_excess) bytes */
    /* live render state + view */
    uint8_t *visible_type;           /* [n_buckets + n_excess] */
    int32_t *visible_ids;            /* [n_blocks] */
    float *depth;                    /* [H*W] metres, -1 invalid */
    const uint8_t *rgb;              /* [H*W*4] uchar4 */
    float *minmax;                   /* [H*W*2] */
    float *raycast;                  /* [H*W*4] voxel-unit xyz + confidence+1 */
    float *icp_points, *icp_normals; /* [H*W*4] */
    /* free-view render state */
    int32_t *fv_visible_ids;         /* [n_blocks] */
    float *fv_minmax;                /* [H*W*2] */
    float *fv_raycast;               /* [H*W*4] */
    uint8_t *fv_colour;              /* [H*W*4] uchar4 */
} gps_tsdf_state;

/* Size in bytes of gps_tsdf_state.scan_scratch (sweep counts + flags + per-workgroup min/max partial images + the persistent
 * bucket-occupancy bitmap: the buffer belongs to the state for its whole life, not to a call). */
GPS_API int64_t gps_tsdf_scratch_bytes(int width, int height, int n_buckets, int n_excess);

/* ITMSceneReconstructionEngine::ResetScene (Reconstruction/CUDA/ITMSceneReconstructionEngine_CUDA.tcu:52-80).  Also fills
 * minmax / fv_minmax with (FAR_AWAY, VERY_CLOSE): CreateExpectedDepths only ever rewrites the 1/8-resolution window of those
 * images, so the reference's per-call memset of the rest happens once, here.  Must precede every other call on a state. */
GPS_API int gps_tsdf_reset(const gps_tsdf_state *s, gps_stream stream);

/* Recomputes the library's private index over the hash table (a bucket-occupancy bitmap kept in scan_scratch, which the
 * raycaster's free-space march reads instead of the table).  The fusion keeps it current by itself; call this after writing
 * the table from OUTSIDE the library (ITMScene::LoadFromDirectory: hash.dat / excess.dat copied into s->hash). */
GPS_API int gps_tsdf_rebuild_index(const gps_tsdf_state *s, gps_stream stream);

/* ITMViewBuilder::UpdateView depth conversion (ViewBuilding/Shared/ITMViewBuilder_Shared.h:27-36):
 * depth_mm int16[H*W] -> s->depth (d <= 0 ? -1 : d * 0.001f). */
GPS_API int gps_tsdf_convert_depth(const gps_tsdf_state *s, const int16_t *depth_mm, gps_stream stream);

/* ITMSceneReconstructionEngine::AllocateSceneFromDepth (…_CUDA.tcu:95-201 / CPU.tpp:129-341): hash-block
 * allocation from s->depth + visible list.  Deterministic: identical to the single-threaded CPU engine
 * (bucket collisions resolved by last pixel in scan order; blocks handed out in ascending slot order).
 * M / invM: host float[16], ORUtils layout m[col*4+row] (pose_d->GetM() and its inverse). */
GPS_API int gps_tsdf_allocate(const gps_tsdf_state *s, const float *M, const float *invM, gps_stream stream);

/* ITMSceneReconstructionEngine::IntegrateIntoScene (…_CUDA.tcu:203-250,348-383; Shared:8-54,105-174) */
GPS_API int gps_tsdf_integrate(const gps_tsdf_state *s, const float *M, gps_stream stream);

/* ITMVisualisationEngine::CreateExpectedDepths (Visualisation/CUDA/…_CUDA.tcu:137-184); free_view selects the
 * render state (0 = live, 1 = free view). */
GPS_API int gps_tsdf_expected_depths(const gps_tsdf_state *s, const float *M, int free_view, gps_stream stream);
/* Pass A of the above alone (the per-workgroup partial min/max images; the rendering-block count stays in its scratch counter): what
 * gps_tsdf_expected_depths_and_raycast launches in front of its raycaster, which does pass B.  A measurement hook -- a caller that
 * wants the min/max image calls gps_tsdf_expected_depths or gps_tsdf_expected_depths_and_raycast NEXT on the same render state. */
GPS_API int gps_tsdf_expected_depths_partial(const gps_tsdf_state *s, const float *M, int free_view, gps_stream stream);

/* GenericRaycast / castRay (…_CUDA.tcu:186-225, Shared:122-221).  update_visible = 1 reproduces
 * CreateICPMaps' modifyVisibleEntries = true (only meaningful for the live state). */
GPS_API int gps_tsdf_raycast(const gps_tsdf_state *s, const float *invM, int free_view, int update_visible,
                             gps_stream stream);

/* gps_tsdf_expected_depths followed by gps_tsdf_raycast as the frame chain issues them, in two launches instead of three: the
 * second pass of the expected depths (reduction of the per-workgroup min / max images) is done by the raycaster's waves.  Same
 * min / max image (every cell the raycaster reads), rays, visibility updates and counters as the two calls. */
GPS_API int gps_tsdf_expected_depths_and_raycast(const gps_tsdf_state *s, const float *M, const float *invM, int free_view,
                                                 int update_visible, gps_stream stream);

/* renderICP_device<false> with smoothing (ITMVisualisationHelpers_CUDA.h:71-81, Shared:438-480) on the live raycast */
GPS_API int gps_tsdf_icp_maps(const gps_tsdf_state *s, const float *invM, gps_stream stream);
/* counters[GPS_TSDF_RAY_STEPS / _READS / _RAYS] := the ray statistics of the last raycast launch on this scene's scratch
 * (one short launch; measurement only -- bench.py prices the raycaster's ray term with them) */
GPS_API int gps_tsdf_ray_stats(const gps_tsdf_state *s, gps_stream stream);
/* The per-wave rows behind those sums, copied to rows_out[capacity_rows][4] (host or device memory, stream-ordered): {castRay steps,
 * loop trips (voxel reads), rays, the LONGEST ray of the wave in loop trips}, one row per wave (8 x 8 pixels) of the last raycast
 * launch on the scene's scratch, workgroup-major (16 x 16 pixel patches row by row, 4 waves each).  Returns the rows copied (>= 0) or
 * a negative error.  Measurement only (tools/raycast_wave_hist.py: is the launch as long as its longest rays?). */
GPS_API int gps_tsdf_ray_wave_rows(const gps_tsdf_state *s, float *rows_out, int capacity_rows, gps_stream stream);

/* ITMVisualisationEngine::FindVisibleBlocks for a free view (…_CUDA.tcu:77-92, buildCompleteVisibleList_device) */
GPS_API int gps_tsdf_find_visible(const gps_tsdf_state *s, const float *M, gps_stream stream);

/* renderColour_device (ITMVisualisationHelpers_CUDA.h:227-245) on the free-view raycast -> s->fv_colour */
GPS_API int gps_tsdf_render_colour(const gps_tsdf_state *s, gps_stream stream);

/* ITMBasicEngine::ProcessFrame with tracking off (Core/ITMBasicEngine.tpp:260-385) = convert_depth + allocate +
 * integrate + expected_depths(live) + raycast(live, update_visible) + icp_maps, on one stream, no host sync. */
GPS_API int gps_tsdf_process_frame(const gps_tsdf_state *s, const int16_t *depth_mm, const float *M, const float *invM,
                                   gps_stream stream);

/* ITMBasicEngine::runRaycast(pose, intrinsics) (Core/ITMBasicEngine.tpp:519-525) = find_visible +
 * expected_depths(free) + raycast(free) + render_colour. */
GPS_API int gps_tsdf_free_raycast(const gps_tsdf_state *s, const float *M, const float *invM, gps_stream stream);

/* The same for SEVERAL views at once.  A keyframe update of SLAMPipeline renders up to 9 free views of one volume state
 * (slam/slam_pipeline.cpp:416-447: localFrameRaycast + keyFrameRaycast call runRaycastByCam once per camera); one view is a
 * chain of 6 short, latency-bound launches (~190 us), and the next frame's fusion has to wait for the last of them.  Here
 * every launch of the chain covers all views (grid.z = view): the same kernels, the same per-view results (bit-identical to
 * gps_tsdf_free_raycast, tests/test_tsdf_gpu.py), the views' dependent memory round trips overlap instead of queueing.
 *
 * Each view brings the render state one ITMRenderState_VH holds (its own visible list, min/max image, ray image, colour
 * image), a scratch area of gps_tsdf_scratch_bytes() and a private counter block (int32[GPS_TSDF_N_COUNTERS]; the list length
 * lands in counters[GPS_TSDF_N_VISIBLE_FREE]).  gps_tsdf_view_init prepares minmax / counters ONCE after allocation (what
 * gps_tsdf_reset does for the state's own free view).  `table` is device memory of gps_tsdf_view_table_bytes(n_views) bytes the
 * call overwrites; n_views <= 12.  The scene's own free-view buffers and scratch are not touched; MAX_RENDERING_BLOCKS
 * overflow of any view is raised in the SCENE's counters[GPS_TSDF_OVERFLOW]. */
typedef struct gps_tsdf_view {
    float M[16], invM[16];  /* pose of the view (ORUtils layout, as gps_tsdf_free_raycast) */
    float fx, fy, cx, cy;   /* intrinsics of the view (image size = the state's) */
    int32_t *visible_ids;   /* [n_blocks] */
    float *minmax;          /* [H*W*2] */
    float *raycast;         /* [H*W*4] out: what fv_raycast receives */
    uint8_t *colour;        /* [H*W*4] out: what fv_colour receives */
    int32_t *scratch;       /* gps_tsdf_scratch_bytes(width, height, n_buckets, n_excess) bytes */
    int32_t *counters;      /* [GPS_TSDF_N_COUNTERS] */
    /* optional: the view's runRaycastByCam tensor glue (exactly gps_raycast_to_maps on this view's rays / colour, written by
     * the batch's last kernel instead of one more launch per view).  color_map == NULL: none. */
    float w2c[16];          /* ROW-major, as gps_raycast_to_maps */
    float *color_map, *vertex_map, *confidence_map, *depth_map, *depth_map_clamped;  /* depth_map_clamped may be NULL */
} gps_tsdf_view;
GPS_API int64_t gps_tsdf_view_table_bytes(int n_views);
GPS_API int gps_tsdf_view_init(const gps_tsdf_state *s, const gps_tsdf_view *view, gps_stream stream);
GPS_API int gps_tsdf_free_raycast_batch(const gps_tsdf_state *s, int n_views, const gps_tsdf_view *views, void *table,
                                        gps_stream stream);

/* SLAMPipeline::runRaycastByCam tensor glue (slam/slam_pipeline.cpp:386-403, src/cv_utils.cpp:322-341) fused:
 * rays float4[H*W] + colour uchar4[H*W] (device) -> color_map[H,W,3] (/255), vertex_map[H,W,3] (metres, 0 where no hit),
 * confidence_map[H,W,1], depth_map[H,W,1] = camera-space z under w2c (host float[16], ROW-major), 0 where no hit;
 * depth_map_clamped (optional) = where(depth < 0.01, 1000, depth), the form the rasterizer consumes
 * (raw_gs_model.cpp:205-207). */
GPS_API int gps_raycast_to_maps(int width, int height, const float *rays, const uint8_t *colour, float voxel_size,
                                const float *w2c_row_major, float *color_map, float *vertex_map,
                                float *confidence_map, float *depth_map, float *depth_map_clamped,
                                gps_stream stream);

/* Host-side pose algebra of ORUtils::SE3Pose as used by ITMBasicEngine.tpp:278-279 and slam_pipeline.cpp:367-371:
 * pose.SetInvM(c2w); pose.Coerce(); -> M = pose.GetM(), invM = pose.GetInvM().
 * c2w is row-major (tensor layout); M/invM use the ORUtils layout.  Pure host code, no GPU. */
GPS_API int gps_pose_from_c2w(const float *c2w_row_major, float *M, float *invM);

/* ------------------------------------------------------------------ */
/* TSDF: meshing (marching cubes over the allocated voxel blocks)       */
/* ------------------------------------------------------------------ */

/* One triangle as ITMMesh::Triangle stores it (Objects/Meshing/ITMMesh.h:18-21): p0 p1 p2 (metres), c0 c1 c2 (vertex
 * colours in [0,1]), clr (colour of the cube's origin voxel) -- 21 floats. */
#define GPS_MESH_TRIANGLE_FLOATS 21

/* Scratch bytes gps_tsdf_mesh_scene needs for this state's capacities (-1 on a bad state). */
GPS_API int64_t gps_tsdf_mesh_workspace_bytes(const gps_tsdf_state *s);

/* replaces ITMMeshingEngine_CUDA<TVoxel, ITMVoxelBlockHash>::MeshScene (Engines/Meshing/CUDA/ITMMeshingEngine_CUDA.tcu:43-80,
 * 101-134; Shared/ITMMeshingEngine_Shared.h:279-471).  triangles: device float[max_triangles][21].
 * counts: device int64[2] = {noTotalTriangles = min(generated, max_triangles - 1) as the reference clamps it, generated}.
 * Triangle ORDER is the reference CPU engine's (hash entry, z, y, x, case table), identical every run -- the reference's
 * CUDA engine appends with atomicAdd in arrival order.  No host synchronisation. */
GPS_API int gps_tsdf_mesh_scene(const gps_tsdf_state *s, int64_t max_triangles, float *triangles, int64_t *counts,
                                void *workspace, int64_t workspace_bytes, gps_stream stream);

/* ------------------------------------------------------------------ */
/* TSDF: camera tracking (depth-only ExtendedTracker, use_gt_pose = false) */
/* ------------------------------------------------------------------ */

#define GPS_TRACK_MAX_LEVELS 8

/* ITMExtendedTracker configuration after SetupLevels (Trackers/Interface/ITMExtendedTracker.cpp:143-177).  Level 0 is the
 * finest.  iter_type: 0 rotation only, 1 translation only, 2 both, 3 none (TrackerIterationType). */
typedef struct {
    int32_t n_levels;
    int32_t iter_type[GPS_TRACK_MAX_LEVELS];
    int32_t n_iter[GPS_TRACK_MAX_LEVELS];
    float space_thresh[GPS_TRACK_MAX_LEVELS];
    float term_thresh, tukey_cutoff;
    int32_t frames_to_skip, frames_to_weight;
} gps_track_config;

/* ITMTrackingState as the tracker uses it (host memory): pose_d (M / invM), pose_pointCloud (M), age_pointCloud,
 * framesProcessed; diag = {iterations run on level 0..7, noValidPoints, f, trackerScore, det(H), [12] poses that rode along with
 * evaluations, [13] those the loop consumed, [14] host ms in the tracker, [15] host ms enqueueing the fusion kernels
 * (gps_tsdf_process_frame_tracked)} of the last call.
 * Matrices in ORUtils layout m[col*4 + row]. */
#define GPS_TRACK_MAILBOX_BLOCK_BYTES 256    /* one answer block of gps_track_state.host_mailbox */
#define GPS_TRACK_MAILBOX_ROWS_BYTES 32768   /* one row table (256 workgroups x 128 bytes) behind the blocks: host-summed rows */
typedef struct {
    float pose_M[16], pose_invM[16], pose_pc_M[16];
    int32_t age_point_cloud, frames_processed;
    float diag[16];
    /* optional: >= 256 bytes of PINNED, device-visible host memory (hipHostMalloc / torch pinned tensor).  If set, every
     * evaluation kernel writes its reduced sums (sequence number last) straight into it and the host spins on the sequence
     * number instead of paying hipMemcpy + stream synchronise per iteration; NULL = the memcpy path.  Layout (32-bit words,
     * owned by the library): 0..31 the result row (two sequence-tagged 64-byte chunks), 32 the acknowledgement of a retired
     * pre-launched evaluation, 48..63 the argument line the pre-launched evaluation reads. */
    void *host_mailbox;
    /* sequence number of the last tracking call on this state (owned by the library; kept by gps_track_state_reset) */
    int32_t mail_seq;
    /* 0: the next tracking call zeroes its scratch words first (set by gps_track_state_reset and after a failed call; set it
     * to 0 yourself when you hand the state a DIFFERENT scratch buffer); otherwise owned by the library */
    int32_t scratch_epoch;
    /* optional (with host_mailbox): 64 bytes of HOST-WRITABLE DEVICE memory from gps_track_arg_line_alloc.  If set, the host
     * writes a pre-launched evaluation's argument line straight into HBM through the PCIe BAR (write-combining stores + sfence)
     * and every workgroup polls it there; NULL = the line lives in the pinned mailbox, one workgroup polls it across PCIe and
     * relays it.  Owned by the caller; kept by gps_track_state_reset. */
    void *dev_arg_line;
    /* size of host_mailbox in bytes; 0 = the 256 bytes above.  With dev_arg_line set and 256 * G bytes (G <= 3) an evaluation
     * also takes the G - 1 poses the LM loop would evaluate next IF the evaluation is rejected (a rejection reads nothing of
     * the evaluation it rejects -- ITMExtendedTracker.cpp:601-612 -- so the host knows them in advance): group g's answer lands
     * in words 64 g .. 64 g + 31 of the mailbox, its argument line 64 g bytes into dev_arg_line's block (the block
     * gps_track_arg_line_alloc returns has room).  Same poses, bit for bit; fewer host <-> device round trips per frame
     * (diag[12] / diag[13] = poses that rode along / that the loop consumed).  Kept by gps_track_state_reset. */
    int32_t mailbox_bytes;
} gps_track_state;

/* Builds the configuration from the reference's tracker string parameters (ITMLibSettings.cpp:54-57 default:
 * levels "rrbb", numiterC 20, numiterF 50, outlierSpaceC 0.1, outlierSpaceF 0.004, minstep 1e-4, tukeyCutOff 8,
 * framesToSkip 20, framesToWeight 50). */
GPS_API int gps_track_config_init(gps_track_config *c, const char *levels, int num_iter_coarse, int num_iter_fine,
                                  float thresh_coarse, float thresh_fine, float term_thresh, float tukey_cutoff,
                                  int frames_to_skip, int frames_to_weight);

/* ITMTrackingState::Reset: identity poses, no point cloud yet. */
GPS_API int gps_track_state_reset(gps_track_state *ts);

/* A 4 KB block (64-byte aligned; the tracker uses its first 64 * G bytes, one line per group) of fine-grained device memory the HOST can write through the BAR, for
 * gps_track_state.dev_arg_line.  *line = NULL (and GPS_OK) when the device's memory is not host-visible (no large BAR): the
 * tracker then keeps its pinned argument line.  The one place the library allocates: ordinary device memory (hipMalloc, a torch
 * tensor) is not host-writable, so the caller cannot provide this block itself.  Free with gps_track_arg_line_free. */
GPS_API int gps_track_arg_line_alloc(void **line);
GPS_API int gps_track_arg_line_free(void *line);

/* Device scratch (depth pyramid levels >= 1, reduction partials) for images of this size. */
GPS_API int64_t gps_track_scratch_bytes(int width, int height);
/* Test / measurement hook: cumulative profile of the pre-launched evaluation kernels since the scratch block was last zeroed
 * (blocking read-back): out = { wall-clock ticks (100 MHz) launches spent on the GPU WAITING for the host's argument line,
 * ticks between the line's arrival and the result leaving (the evaluation proper), evaluations run, launches retired unused }. */
GPS_API int gps_track_poll_profile(const void *scratch, int width, int height, uint32_t out[4], gps_stream stream);
/* The same clock split into phases, as workgroup 0 (the summer) of an evaluation sees them, for the finest level (out[0..3]) and
 * for the coarser levels (out[4..7]): { evaluations, ticks from the argument line's arrival to the end of its own pixel loop,
 * ticks from there until every row of the table carries the launch's tag, ticks from there to the mailbox store }. */
GPS_API int gps_track_poll_phases(const void *scratch, int width, int height, uint32_t out[8], gps_stream stream);

/* ITMExtendedTracker::TrackCamera (useDepth, !useColour): refines ts->pose_M / pose_invM against the ICP maps of the last
 * raycast (s->icp_points / s->icp_normals, rendered from ts->pose_pc_M) using s->depth.
 * One launch builds the depth pyramid levels and the valid-pixel count; every Levenberg-Marquardt iteration is then ONE
 * launch (ITMExtendedTracker_CUDA.cu's depthTrackerOneLevel_g_rt_device evaluation + a fixed-order reduction by the last
 * workgroup to finish, written straight into ts->host_mailbox) followed by the 6x6 Cholesky solve, damping, SE3 update and
 * convergence test on the host (ITMExtendedTracker.cpp:470-665) -- the next iteration's kernel arguments depend on that
 * decision.  HOST-SYNCHRONOUS: the call returns when the refined pose is in ts (the caller needs it to enqueue the fusion).
 * GPS_ERR_LAUNCH also reports a kernel that never delivered its result (bounded wait on the mailbox). */
GPS_API int gps_tsdf_track_camera(const gps_tsdf_state *s, const gps_track_config *cfg, gps_track_state *ts, void *scratch,
                                  int64_t scratch_bytes, gps_stream stream);

/* ITMBasicEngine::ProcessFrame with tracking active and the default failure mode (Core/ITMBasicEngine.tpp:260-385):
 * convert_depth -> track (once a point cloud exists) -> allocate + integrate -> expected depths + raycast + ICP maps at
 * the tracked pose; updates ts (pose_pointCloud := pose_d, age_pointCloud, framesProcessed). */
GPS_API int gps_tsdf_process_frame_tracked(const gps_tsdf_state *s, const int16_t *depth_mm, const gps_track_config *cfg,
                                           gps_track_state *ts, void *scratch, int64_t scratch_bytes, gps_stream stream);

/* Same, with a hook between the part of the frame that only READS the voxel volume (depth conversion, tracking) and the part
 * that modifies it (allocate, integrate, ...): before_fusion(user), if not NULL, is called on the calling thread at that point.
 * A pipeline that renders free views of the volume on another stream passes a callback that makes `stream` wait for them
 * (hipStreamWaitEvent), so that the next frame's tracking overlaps those raycasts instead of queueing behind them. */
GPS_API int gps_tsdf_process_frame_tracked_gated(const gps_tsdf_state *s, const int16_t *depth_mm, const gps_track_config *cfg,
                                                 gps_track_state *ts, void *scratch, int64_t scratch_bytes, gps_stream stream,
                                                 void (*before_fusion)(void *user), void *user);

#ifdef __cplusplus
}
#endif
#endif /* GPS_SLAM_HIP_H */

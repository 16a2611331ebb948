// Launcher level of the splat drop-in boundary: the `gsplat::*_tensor` functions (gsplat/rasterizer/bindings.h:44-426)
// and the three free launchers (rasterizer/ssim.h, rasterizer/simple_knn.h:21) that gsplat/gsplat_wapper.{hpp,cpp} call,
// with THE REFERENCE'S SIGNATURES -- same namespace, names, parameter types and order, return tuples -- implemented on the
// C-ABI of include/gps_slam_hip.h.  A maintainer who keeps the reference's own gsplat_wapper.{hpp,cpp} (the autograd
// Functions raw_gs_model.cpp programs against) links it against these definitions instead of the .cu files; the symbol
// match is proven by tests/test_reference_binding_cpu.py, which builds the reference's wrapper (host code run through
// torch's own hipify, as every PyTorch extension build on ROCm does) and links it against this object with no
// unresolved symbol.  host/gsplat_wapper.cpp -- this repository's mirror of that wrapper -- goes through the same
// launchers, so they are the product path of the operator surface, not a side demo.
//
// Scope: the launchers the wrapper calls (13 + 3).  Arguments GPS-SLAM never exercises (covars, compensations, non-pinhole
// cameras, tile masks, unsorted binning, absgrad on the ges variants, packed mode) raise c10::Error like the reference's
// AT_ERROR paths; C == 1 camera (raw_gs_model.cpp:225 always unsqueezes one).
#pragma once
#include <torch/torch.h>

#include <tuple>

namespace gsplat {

enum CameraModelType { PINHOLE = 0, ORTHO = 1, FISHEYE = 2 };  // bindings.h:37-42

// bindings.h:96-115 <- fully_fused_projection_fwd.cu:196-273.  -> {radii[C,N] i32, means2d[C,N,2], depths[C,N],
// conics[C,N,3], compensations (undefined)}
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> fully_fused_projection_fwd_tensor(
    const torch::Tensor& means, const at::optional<torch::Tensor>& covars, const at::optional<torch::Tensor>& quats,
    const at::optional<torch::Tensor>& scales, const torch::Tensor& viewmats, const torch::Tensor& Ks,
    const uint32_t image_width, const uint32_t image_height, const float eps2d, const float near_plane,
    const float far_plane, const float radius_clip, const bool calc_compensations, const CameraModelType camera_model);

// bindings.h:117-145 <- fully_fused_projection_bwd.cu:288-403.  -> {v_means, v_covars (undefined), v_quats, v_scales,
// v_viewmats (undefined)}
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> fully_fused_projection_bwd_tensor(
    const torch::Tensor& means, const at::optional<torch::Tensor>& covars, const at::optional<torch::Tensor>& quats,
    const at::optional<torch::Tensor>& scales, const torch::Tensor& viewmats, const torch::Tensor& Ks,
    const uint32_t image_width, const uint32_t image_height, const float eps2d, const CameraModelType camera_model,
    const torch::Tensor& radii, const torch::Tensor& conics, const at::optional<torch::Tensor>& compensations,
    const torch::Tensor& v_means2d, const torch::Tensor& v_depths, const torch::Tensor& v_conics,
    const at::optional<torch::Tensor>& v_compensations, const bool viewmats_requires_grad);

// bindings.h:147-158, 160-164 <- isect_tiles.cu.  -> {tiles_per_gauss[C,N], isect_ids i64[I], flatten_ids i32[I]}
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> isect_tiles_tensor(
    const torch::Tensor& means2d, const torch::Tensor& radii, const torch::Tensor& depths,
    const at::optional<torch::Tensor>& camera_ids, const at::optional<torch::Tensor>& gaussian_ids, const uint32_t C,
    const uint32_t tile_size, const uint32_t tile_width, const uint32_t tile_height, const bool sort,
    const bool double_buffer);
torch::Tensor isect_offset_encode_tensor(const torch::Tensor& isect_ids, const uint32_t C, const uint32_t tile_width,
                                         const uint32_t tile_height);

// bindings.h:166-183 <- isect_tiles_no_depth.cu:132-461.  -> {tiles_per_gauss, isect_ids, flatten_ids, group_gs_ids,
// group_starts}, exact sizes
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> isect_tiles_tensor_no_depth(
    const torch::Tensor& means2d, const torch::Tensor& radii, const torch::Tensor& depths,
    const at::optional<torch::Tensor>& camera_ids, const at::optional<torch::Tensor>& gaussian_ids, const uint32_t C,
    const uint32_t tile_size, const uint32_t tile_width, const uint32_t tile_height, const bool sort,
    const bool double_buffer);
torch::Tensor isect_offset_encode_tensor_no_depth(const torch::Tensor& isect_ids, const uint32_t C,
                                                  const uint32_t tile_width, const uint32_t tile_height);

// bindings.h:185-202 <- rasterize_to_pixels_fwd.cu.  -> {render_colors[C,H,W,D], render_alphas[C,H,W,1], last_ids[C,H,W]}
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> rasterize_to_pixels_fwd_tensor(
    const torch::Tensor& means2d, const torch::Tensor& conics, const torch::Tensor& colors, const torch::Tensor& opacities,
    const at::optional<torch::Tensor>& backgrounds, const at::optional<torch::Tensor>& mask, const uint32_t image_width,
    const uint32_t image_height, const uint32_t tile_size, const torch::Tensor& tile_offsets,
    const torch::Tensor& flatten_ids);

// bindings.h:204-224 <- rasterize_to_pixels_fwd_ges.cu:223-407
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> rasterize_to_pixels_fwd_ges_tensor(
    const torch::Tensor& means2d, const torch::Tensor& conics, const torch::Tensor& colors, const torch::Tensor& opacities,
    const torch::Tensor& ref_depth_map, const torch::Tensor& base_color_map, const at::optional<torch::Tensor>& backgrounds,
    const at::optional<torch::Tensor>& mask, const uint32_t image_width, const uint32_t image_height,
    const uint32_t tile_size, const torch::Tensor& tile_offsets, const torch::Tensor& flatten_ids, const float delta_depth);

// bindings.h:226-256 <- rasterize_to_pixels_bwd.cu.  -> {v_means2d_abs, v_means2d, v_conics, v_colors, v_opacities}
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> rasterize_to_pixels_bwd_tensor(
    const torch::Tensor& means2d, const torch::Tensor& conics, const torch::Tensor& colors, const torch::Tensor& opacities,
    const at::optional<torch::Tensor>& backgrounds, const at::optional<torch::Tensor>& mask, const uint32_t image_width,
    const uint32_t image_height, const uint32_t tile_size, const torch::Tensor& tile_offsets,
    const torch::Tensor& flatten_ids, const torch::Tensor& render_alphas, const torch::Tensor& last_ids,
    const torch::Tensor& v_render_colors, const torch::Tensor& v_render_alphas, bool absgrad);

// bindings.h:258-292 <- rasterize_to_pixels_bwd_ges.cu:164-291 (exact tile-parallel adjoint)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> rasterize_to_pixels_bwd_ges_tensor(
    const torch::Tensor& means2d, const torch::Tensor& conics, const torch::Tensor& colors, const torch::Tensor& opacities,
    const torch::Tensor& ref_depth_map, const torch::Tensor& base_color_map, const at::optional<torch::Tensor>& backgrounds,
    const at::optional<torch::Tensor>& mask, const uint32_t image_width, const uint32_t image_height,
    const uint32_t tile_size, const torch::Tensor& tile_offsets, const torch::Tensor& flatten_ids, const float delta_depth,
    const torch::Tensor& render_alphas, const torch::Tensor& last_ids, const torch::Tensor& v_render_colors,
    const torch::Tensor& v_render_alphas, bool absgrad);

// bindings.h:294-326 <- rasterize_to_pixels_bwd_ges_new_parallel.cu:203-385 (Gaussian-parallel 2r x 2r box backward)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterize_to_pixels_bwd_ges_gs_parallel_tensor(
    const torch::Tensor& means2d, const torch::Tensor& conics, const torch::Tensor& colors, const torch::Tensor& opacities,
    const torch::Tensor& radiis, const torch::Tensor& ref_depth_map, const torch::Tensor& base_color_map,
    const at::optional<torch::Tensor>& backgrounds, const uint32_t image_width, const uint32_t image_height,
    const uint32_t n_isects, const torch::Tensor& group_gs_ids, const torch::Tensor& group_starts, const float delta_depth,
    const torch::Tensor& render_alphas, const torch::Tensor& v_render_colors, const torch::Tensor& v_render_alphas,
    bool absgrad);

// bindings.h:344-349, 351-359 <- compute_sh_fwd.cu:40-72, compute_sh_bwd.cu:56-123.  bwd -> {v_coeffs, v_dirs}
torch::Tensor compute_sh_fwd_tensor(const uint32_t degrees_to_use, const torch::Tensor& dirs, const torch::Tensor& coeffs,
                                    const at::optional<torch::Tensor> masks);
std::tuple<torch::Tensor, torch::Tensor> compute_sh_bwd_tensor(const uint32_t K, const uint32_t degrees_to_use,
                                                               const torch::Tensor& dirs, const torch::Tensor& coeffs,
                                                               const at::optional<torch::Tensor> masks,
                                                               const torch::Tensor& v_colors, bool compute_v_dirs);

}  // namespace gsplat

// rasterizer/ssim.h:7-25 <- ssim.cu:385-460.  [B,CH,H,W]; permuted views of contiguous [B,H,W,CH] memory are read in
// place (the maps then come back as the same kind of view).
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> fusedssim(float C1, float C2, torch::Tensor& img1,
                                                                                 torch::Tensor& img2, bool train);
torch::Tensor fusedssim_backward(float C1, float C2, torch::Tensor& img1, torch::Tensor& img2, torch::Tensor& dL_dmap,
                                 torch::Tensor& dm_dmu1, torch::Tensor& dm_dsigma1_sq, torch::Tensor& dm_dsigma12);

// rasterizer/simple_knn.h:21 <- simple_knn.cu:191-240
torch::Tensor distCUDA2(const torch::Tensor& points);

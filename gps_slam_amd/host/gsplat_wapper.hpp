// The splat operator surface raw_gs_model.cpp programs against, with the reference's names, argument order, tensor
// shapes and error behaviour (gsplat/gsplat_wapper.hpp:16-95, 97-241, 489-620, 679-709; gsplat_wapper.cpp:3-139),
// implemented on the C-ABI of include/gps_slam_hip.h.  Configurations GPS-SLAM never uses (covars, compensations,
// non-pinhole cameras, backgrounds, masks, absgrad) throw, like the reference's AT_ERROR paths.
#pragma once
#include "gps_host_common.hpp"
#include "hip_bindings.hpp"  // the launcher level (rasterizer/bindings.h, ssim.h, simple_knn.h signatures) + distCUDA2

double getDuration(struct timespec start, struct timespec end);  // gsplat_wapper.hpp:12, gsplat_wapper.cpp:3-13 (ms)

// gsplat_wapper.hpp:16-95: apply(sh_degree, dirs[...,3], coeffs[...,K,3], masks[...]) -> colors[...,3]
struct SphericalHarmonicsNew : public torch::autograd::Function<SphericalHarmonicsNew> {
    static torch::Tensor forward(torch::autograd::AutogradContext* ctx, int sh_degree, torch::Tensor dirs,
                                 torch::Tensor coeffs, torch::Tensor masks);
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::tensor_list grad_outputs);
};

// gsplat_wapper.hpp:97-241: apply(means, covars, quats, scales, viewmats[C,4,4], Ks[C,3,3], width, height, eps2d,
// near_plane, far_plane, radius_clip, calc_compensations, camera_model) -> {radii, means2d, depths, conics,
// compensations (undefined)}
struct FullyFusedProjection : public torch::autograd::Function<FullyFusedProjection> {
    static torch::autograd::tensor_list forward(torch::autograd::AutogradContext* ctx, torch::Tensor means,
                                                c10::optional<torch::Tensor> covars, torch::Tensor quats,
                                                torch::Tensor scales, torch::Tensor viewmats, torch::Tensor Ks,
                                                int width, int height, float eps2d, float near_plane, float far_plane,
                                                float radius_clip, bool calc_compensations, std::string camera_model);
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::tensor_list grad_outputs);
};

// gsplat_wapper.hpp:489-620: apply(means2d, conics, colors, opacities, radiis, ref_depth_map, base_color_map,
// backgrounds, masks, width, height, tile_size, isect_offsets, flatten_ids, group_gs_ids, group_starts, absgrad,
// delta_depth) -> {render_colors[1,H,W,4], weight_sum[1,H,W,1]}
struct RasterizeToPixelsGes_NewParallel : public torch::autograd::Function<RasterizeToPixelsGes_NewParallel> {
    static torch::autograd::tensor_list forward(torch::autograd::AutogradContext* ctx, torch::Tensor means2d,
                                                torch::Tensor conics, torch::Tensor colors, torch::Tensor opacities,
                                                torch::Tensor radiis, torch::Tensor ref_depth_map,
                                                torch::Tensor base_color_map, c10::optional<torch::Tensor> backgrounds,
                                                c10::optional<torch::Tensor> masks, int width, int height,
                                                int tile_size, torch::Tensor isect_offsets, torch::Tensor flatten_ids,
                                                torch::Tensor group_gs_ids, torch::Tensor group_starts, bool absgrad,
                                                float delta_depth);
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::tensor_list grad_outputs);
};

// gsplat_wapper.hpp:355-487: apply(means2d, conics, colors, opacities, ref_depth_map, base_color_map, backgrounds, masks,
// width, height, tile_size, isect_offsets, flatten_ids, absgrad, delta_depth) -> {render_colors, weight_sum}; the backward is
// the EXACT tile-parallel adjoint (rasterize_to_pixels_bwd_ges.cu).  Unused by raw_gs_model.cpp (:291 picks _NewParallel).
struct RasterizeToPixelsGes : public torch::autograd::Function<RasterizeToPixelsGes> {
    static torch::autograd::tensor_list forward(torch::autograd::AutogradContext* ctx, torch::Tensor means2d,
                                                torch::Tensor conics, torch::Tensor colors, torch::Tensor opacities,
                                                torch::Tensor ref_depth_map, torch::Tensor base_color_map,
                                                c10::optional<torch::Tensor> backgrounds,
                                                c10::optional<torch::Tensor> masks, int width, int height, int tile_size,
                                                torch::Tensor isect_offsets, torch::Tensor flatten_ids, bool absgrad,
                                                float delta_depth);
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::tensor_list grad_outputs);
};

// gsplat_wapper.hpp:243-353: apply(means2d, conics, colors, opacities, backgrounds, masks, width, height, tile_size,
// isect_offsets, flatten_ids, absgrad) -> {render_colors[1,H,W,4], render_alphas[1,H,W,1]}.  As in the reference the
// backward runs WITHOUT the backgrounds (gsplat_wapper.hpp:307-315 passes an empty optional) and v_backgrounds is the
// sum of v_render_colors * (1 - render_alphas).  masks throw (never used); absgrad computes v_means2d_abs, which the
// reference discards (:323-326 overwrites means2d with |means2d| instead -- not reproduced, see LABBOOK.md).
struct RasterizeToPixels : public torch::autograd::Function<RasterizeToPixels> {
    static torch::autograd::tensor_list forward(torch::autograd::AutogradContext* ctx, torch::Tensor means2d,
                                                torch::Tensor conics, torch::Tensor colors, torch::Tensor opacities,
                                                c10::optional<torch::Tensor> backgrounds,
                                                c10::optional<torch::Tensor> masks, int width, int height,
                                                int tile_size, torch::Tensor isect_offsets, torch::Tensor flatten_ids,
                                                bool absgrad);
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::tensor_list grad_outputs);
};

// gsplat_wapper.hpp:622-677: apply(C1, C2, img1[B,CH,H,W], img2, padding, train) -> ssim_map ("valid" crops 5 px per side);
// gradient w.r.t. img1 only.  Permuted views of [H,W,CH] images (raw_gs_model.cpp:390-395) are read in place through the
// kernels' channels-last layout instead of being copied to planar form.
struct FusedSSIMMap : public torch::autograd::Function<FusedSSIMMap> {
    static torch::Tensor forward(torch::autograd::AutogradContext* ctx, double C1, double C2, torch::Tensor img1,
                                 torch::Tensor img2, std::string padding, bool train);
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::tensor_list grad_outputs);
};

using torch::autograd::variable_list;

// gsplat_wapper.cpp:55-85: -> {tiles_per_gauss[1,N], isect_ids i64[I], flatten_ids i32[I], group_gs_ids i32[G],
// group_starts i32[G]} with exact sizes (one host read of the two counts, where the reference syncs twice).
variable_list isectTilesNoDepth(torch::Tensor means2d, torch::Tensor radii, torch::Tensor depths, int tile_size,
                                int tile_width, int tile_height, bool sort = true);

// gsplat_wapper.cpp:88-91: sorted isect_ids (tile id in the low 32 bits) -> offsets[C, tile_height, tile_width]
torch::Tensor isectOffsetEncodeNoDepth(torch::Tensor isect_ids, int n_cameras, int tile_width, int tile_height);

// gsplat_wapper.cpp:15-43: depth-keyed binning -> {tiles_per_gauss[1,N], isect_ids i64[I] (tile << 32 | depth bits,
// sorted), flatten_ids i32[I]}; gsplat_wapper.cpp:45-48: offsets[C, tile_height, tile_width] from the sorted keys
variable_list isectTiles(torch::Tensor means2d, torch::Tensor radii, torch::Tensor depths, int tile_size, int tile_width,
                         int tile_height, bool sort = true);
torch::Tensor isectOffsetEncode(torch::Tensor isect_ids, int n_cameras, int tile_width, int tile_height);

// gsplat_wapper.cpp:50-53 -> distCUDA2 (simple_knn.h:21, hip_bindings.hpp): mean squared distance to the 3 nearest neighbours
torch::Tensor simpleKNN(torch::Tensor points);

// gsplat_wapper.cpp:105-139
int degFromSh(int numBases);
int numShBases(int degree);
torch::Tensor rgb2sh(const torch::Tensor& rgb);
torch::Tensor sh2rgb(const torch::Tensor& sh);

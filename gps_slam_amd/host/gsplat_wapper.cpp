// See gsplat_wapper.hpp.  Two levels, like the reference: these autograd Functions / free functions own the autograd
// bookkeeping and the argument conventions of gsplat/gsplat_wapper.{hpp,cpp}; the launches go through the `gsplat::*_tensor`
// launchers of hip_bindings.cpp (the reference's rasterizer/bindings.h surface on the C-ABI).  Nothing is computed on the host.
#include "gsplat_wapper.hpp"

#include <ctime>

#include "hip_bindings.hpp"

using namespace gpsh;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

namespace {

torch::Tensor contig_f32(const torch::Tensor& t, const char* name) {
    TORCH_CHECK(t.defined() && t.is_cuda(), name, " must be a device tensor");
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32");
    return t.contiguous();
}

at::optional<torch::Tensor> opt(const torch::Tensor& t) { return t.defined() ? at::optional<torch::Tensor>(t) : at::nullopt; }

}  // namespace

// gsplat_wapper.cpp:3-13
double getDuration(struct timespec start, struct timespec end) {
    return (double)(end.tv_sec - start.tv_sec) * 1000.0 + (double)(end.tv_nsec - start.tv_nsec) / 1.0e6;
}

// ------------------------------------------------------------------------------------------------ SH
torch::Tensor SphericalHarmonicsNew::forward(AutogradContext* ctx, int sh_degree, torch::Tensor dirs,
                                             torch::Tensor coeffs, torch::Tensor masks) {
    dirs = contig_f32(dirs, "dirs");
    coeffs = contig_f32(coeffs, "coeffs");
    torch::Tensor m;
    if (masks.defined()) m = masks.contiguous();
    ctx->save_for_backward({dirs, coeffs, m});
    ctx->saved_data["sh_degree"] = (int64_t)sh_degree;
    ctx->saved_data["num_bases"] = (int64_t)coeffs.size(-2);
    ctx->saved_data["need_dirs"] = dirs.requires_grad();
    return gsplat::compute_sh_fwd_tensor((uint32_t)sh_degree, dirs, coeffs, opt(m));
}

tensor_list SphericalHarmonicsNew::backward(AutogradContext* ctx, tensor_list grad_outputs) {
    auto saved = ctx->get_saved_variables();
    const int sh_degree = (int)ctx->saved_data["sh_degree"].toInt(), K = (int)ctx->saved_data["num_bases"].toInt();
    auto r = gsplat::compute_sh_bwd_tensor((uint32_t)K, (uint32_t)sh_degree, saved[0], saved[1], opt(saved[2]),
                                           contig_f32(grad_outputs[0], "v_colors"), ctx->saved_data["need_dirs"].toBool());
    return {torch::Tensor(), std::get<1>(r), std::get<0>(r), torch::Tensor()};
}

// ------------------------------------------------------------------------------------------------ projection
tensor_list FullyFusedProjection::forward(AutogradContext* ctx, torch::Tensor means, c10::optional<torch::Tensor> covars,
                                          torch::Tensor quats, torch::Tensor scales, torch::Tensor viewmats,
                                          torch::Tensor Ks, int width, int height, float eps2d, float near_plane,
                                          float far_plane, float radius_clip, bool calc_compensations,
                                          std::string camera_model) {
    TORCH_CHECK(camera_model == "pinhole" || camera_model == "ortho" || camera_model == "fisheye",
                "camera_model must be pinhole / ortho / fisheye");  // gsplat_wapper.hpp:118-131
    const gsplat::CameraModelType cm = camera_model == "pinhole" ? gsplat::PINHOLE
                                       : camera_model == "ortho" ? gsplat::ORTHO : gsplat::FISHEYE;
    means = contig_f32(means, "means"); quats = contig_f32(quats, "quats"); scales = contig_f32(scales, "scales");
    viewmats = contig_f32(viewmats, "viewmats"); Ks = contig_f32(Ks, "Ks");
    auto r = gsplat::fully_fused_projection_fwd_tensor(means, covars, quats, scales, viewmats, Ks, (uint32_t)width,
                                                       (uint32_t)height, eps2d, near_plane, far_plane, radius_clip,
                                                       calc_compensations, cm);
    auto radii = std::get<0>(r), means2d = std::get<1>(r), depths = std::get<2>(r), conics = std::get<3>(r);
    ctx->save_for_backward({means, quats, scales, viewmats, Ks, radii, conics});
    ctx->saved_data["width"] = (int64_t)width;
    ctx->saved_data["height"] = (int64_t)height;
    ctx->saved_data["eps2d"] = (double)eps2d;
    // compensations: never computed here (calc_compensations is rejected by the launcher); autograd wants defined outputs
    auto compensations = torch::empty({0}, f32(means.device()));
    ctx->mark_non_differentiable({radii, compensations});
    return {radii, means2d, depths, conics, compensations};
}

tensor_list FullyFusedProjection::backward(AutogradContext* ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const torch::Tensor &means = s[0], &quats = s[1], &scales = s[2], &viewmats = s[3], &Ks = s[4], &radii = s[5],
                        &conics = s[6];
    const int width = (int)ctx->saved_data["width"].toInt(), height = (int)ctx->saved_data["height"].toInt();
    const float eps2d = (float)ctx->saved_data["eps2d"].toDouble();
    const int N = (int)means.size(0);
    // undefined incoming gradients mean "zero" (autograd materialises them only on request)
    auto v_means2d = g[1].defined() ? contig_f32(g[1], "v_means2d") : torch::zeros({1, N, 2}, means.options());
    auto v_depths = g[2].defined() ? contig_f32(g[2], "v_depths") : torch::zeros({1, N}, means.options());
    auto v_conics = g[3].defined() ? contig_f32(g[3], "v_conics") : torch::zeros({1, N, 3}, means.options());
    auto r = gsplat::fully_fused_projection_bwd_tensor(means, at::nullopt, quats, scales, viewmats, Ks, (uint32_t)width,
                                                       (uint32_t)height, eps2d, gsplat::PINHOLE, radii, conics, at::nullopt,
                                                       v_means2d, v_depths, v_conics, at::nullopt, false);
    tensor_list out(14);
    out[0] = std::get<0>(r); out[2] = std::get<2>(r); out[3] = std::get<3>(r);
    return out;
}

// ------------------------------------------------------------------------------------------------ ges rasterizer
namespace {
// forward half shared by the two ges Functions (gsplat_wapper.hpp:376-395, 510-529)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> ges_forward(torch::Tensor& means2d, torch::Tensor& conics,
                                                                    torch::Tensor& colors, torch::Tensor& opacities,
                                                                    torch::Tensor& ref_depth_map, torch::Tensor& base_color_map,
                                                                    const c10::optional<torch::Tensor>& backgrounds,
                                                                    const c10::optional<torch::Tensor>& masks, int width,
                                                                    int height, int tile_size, torch::Tensor& isect_offsets,
                                                                    torch::Tensor& flatten_ids, float delta_depth) {
    means2d = contig_f32(means2d, "means2d"); conics = contig_f32(conics, "conics");
    colors = contig_f32(colors, "colors"); opacities = contig_f32(opacities, "opacities");
    ref_depth_map = contig_f32(ref_depth_map, "ref_depth_map");
    base_color_map = base_color_map.defined() ? base_color_map.contiguous() : torch::empty({0}, means2d.options());
    isect_offsets = isect_offsets.contiguous(); flatten_ids = flatten_ids.contiguous();
    return gsplat::rasterize_to_pixels_fwd_ges_tensor(means2d, conics, colors, opacities, ref_depth_map, base_color_map,
                                                      backgrounds, masks, (uint32_t)width, (uint32_t)height,
                                                      (uint32_t)tile_size, isect_offsets, flatten_ids, delta_depth);
}
torch::Tensor grad_or_zeros(const torch::Tensor& g, at::IntArrayRef shape, const torch::Tensor& like, const char* name) {
    return g.defined() ? contig_f32(g, name) : torch::zeros(shape, like.options());
}
}  // namespace

tensor_list RasterizeToPixelsGes_NewParallel::forward(
    AutogradContext* ctx, torch::Tensor means2d, torch::Tensor conics, torch::Tensor colors, torch::Tensor opacities,
    torch::Tensor radiis, torch::Tensor ref_depth_map, torch::Tensor base_color_map,
    c10::optional<torch::Tensor> backgrounds, c10::optional<torch::Tensor> masks, int width, int height, int tile_size,
    torch::Tensor isect_offsets, torch::Tensor flatten_ids, torch::Tensor group_gs_ids, torch::Tensor group_starts,
    bool absgrad, float delta_depth) {
    TORCH_CHECK(!absgrad, "absgrad is never used by the ges path and is not implemented");
    auto r = ges_forward(means2d, conics, colors, opacities, ref_depth_map, base_color_map, backgrounds, masks, width, height,
                         tile_size, isect_offsets, flatten_ids, delta_depth);
    radiis = radiis.contiguous(); group_gs_ids = group_gs_ids.contiguous(); group_starts = group_starts.contiguous();
    auto render_alphas = std::get<1>(r);
    ctx->save_for_backward({means2d, conics, colors, opacities, radiis, ref_depth_map, base_color_map, render_alphas,
                            group_gs_ids, group_starts});
    ctx->saved_data["width"] = (int64_t)width;
    ctx->saved_data["height"] = (int64_t)height;
    ctx->saved_data["delta_depth"] = (double)delta_depth;
    ctx->saved_data["n_isects"] = (int64_t)flatten_ids.size(0);
    return {std::get<0>(r), render_alphas};
}

tensor_list RasterizeToPixelsGes_NewParallel::backward(AutogradContext* ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const int width = (int)ctx->saved_data["width"].toInt(), height = (int)ctx->saved_data["height"].toInt();
    const float delta_depth = (float)ctx->saved_data["delta_depth"].toDouble();
    auto v_rc = grad_or_zeros(g[0], {1, height, width, 4}, s[0], "v_render_colors");
    auto v_ra = grad_or_zeros(g[1], {1, height, width, 1}, s[0], "v_render_alphas");
    auto r = gsplat::rasterize_to_pixels_bwd_ges_gs_parallel_tensor(
        s[0], s[1], s[2], s[3], s[4], s[5], s[6], at::nullopt, (uint32_t)width, (uint32_t)height,
        (uint32_t)ctx->saved_data["n_isects"].toInt(), s[8], s[9], delta_depth, s[7], v_rc, v_ra, false);
    tensor_list out(18);
    out[0] = std::get<1>(r); out[1] = std::get<2>(r); out[2] = std::get<3>(r); out[3] = std::get<4>(r);
    return out;
}

tensor_list RasterizeToPixelsGes::forward(AutogradContext* ctx, torch::Tensor means2d, torch::Tensor conics,
                                          torch::Tensor colors, torch::Tensor opacities, torch::Tensor ref_depth_map,
                                          torch::Tensor base_color_map, c10::optional<torch::Tensor> backgrounds,
                                          c10::optional<torch::Tensor> masks, int width, int height, int tile_size,
                                          torch::Tensor isect_offsets, torch::Tensor flatten_ids, bool absgrad,
                                          float delta_depth) {
    TORCH_CHECK(!absgrad, "absgrad is never used by the ges path and is not implemented");
    auto r = ges_forward(means2d, conics, colors, opacities, ref_depth_map, base_color_map, backgrounds, masks, width, height,
                         tile_size, isect_offsets, flatten_ids, delta_depth);
    auto render_alphas = std::get<1>(r);
    ctx->save_for_backward({means2d, conics, colors, opacities, ref_depth_map, base_color_map, isect_offsets, flatten_ids,
                            render_alphas, std::get<2>(r)});
    ctx->saved_data["width"] = (int64_t)width;
    ctx->saved_data["height"] = (int64_t)height;
    ctx->saved_data["tile_size"] = (int64_t)tile_size;
    ctx->saved_data["delta_depth"] = (double)delta_depth;
    return {std::get<0>(r), render_alphas};
}

tensor_list RasterizeToPixelsGes::backward(AutogradContext* ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const int width = (int)ctx->saved_data["width"].toInt(), height = (int)ctx->saved_data["height"].toInt();
    const int tile_size = (int)ctx->saved_data["tile_size"].toInt();
    const float delta_depth = (float)ctx->saved_data["delta_depth"].toDouble();
    auto v_rc = grad_or_zeros(g[0], {1, height, width, 4}, s[0], "v_render_colors");
    auto v_ra = grad_or_zeros(g[1], {1, height, width, 1}, s[0], "v_render_alphas");
    auto r = gsplat::rasterize_to_pixels_bwd_ges_tensor(s[0], s[1], s[2], s[3], s[4], s[5], at::nullopt, at::nullopt,
                                                        (uint32_t)width, (uint32_t)height, (uint32_t)tile_size, s[6], s[7],
                                                        delta_depth, s[8], s[9], v_rc, v_ra, false);
    tensor_list out(15);
    out[0] = std::get<1>(r); out[1] = std::get<2>(r); out[2] = std::get<3>(r); out[3] = std::get<4>(r);
    return out;
}

// ------------------------------------------------------------------------------------------------ raw rasterizer
tensor_list RasterizeToPixels::forward(AutogradContext* ctx, torch::Tensor means2d, torch::Tensor conics, torch::Tensor colors,
                                       torch::Tensor opacities, c10::optional<torch::Tensor> backgrounds,
                                       c10::optional<torch::Tensor> masks, int width, int height, int tile_size,
                                       torch::Tensor isect_offsets, torch::Tensor flatten_ids, bool absgrad) {
    means2d = contig_f32(means2d, "means2d"); conics = contig_f32(conics, "conics");
    colors = contig_f32(colors, "colors"); opacities = contig_f32(opacities, "opacities");
    TORCH_CHECK(means2d.size(0) == 1, "single camera (C == 1)");
    isect_offsets = isect_offsets.contiguous(); flatten_ids = flatten_ids.contiguous();
    if (backgrounds.has_value() && backgrounds->defined()) backgrounds = contig_f32(*backgrounds, "backgrounds");
    auto r = gsplat::rasterize_to_pixels_fwd_tensor(means2d, conics, colors, opacities, backgrounds, masks, (uint32_t)width,
                                                    (uint32_t)height, (uint32_t)tile_size, isect_offsets, flatten_ids);
    auto ra = std::get<1>(r);
    ctx->save_for_backward({means2d, conics, colors, opacities, isect_offsets, flatten_ids, ra, std::get<2>(r)});
    ctx->saved_data["width"] = (int64_t)width;
    ctx->saved_data["height"] = (int64_t)height;
    ctx->saved_data["tile_size"] = (int64_t)tile_size;
    ctx->saved_data["absgrad"] = absgrad;
    return {std::get<0>(r), ra};
}

tensor_list RasterizeToPixels::backward(AutogradContext* ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const torch::Tensor& render_alphas = s[6];
    const int width = (int)ctx->saved_data["width"].toInt(), height = (int)ctx->saved_data["height"].toInt();
    const int tile_size = (int)ctx->saved_data["tile_size"].toInt();
    const bool absgrad = ctx->saved_data["absgrad"].toBool();
    auto v_rc = grad_or_zeros(g[0], {1, height, width, 4}, s[0], "v_render_colors");
    auto v_ra = grad_or_zeros(g[1], {1, height, width, 1}, s[0], "v_render_alphas");
    // backgrounds = nullopt: the reference's backward never receives them (gsplat_wapper.hpp:307)
    auto r = gsplat::rasterize_to_pixels_bwd_tensor(s[0], s[1], s[2], s[3], at::nullopt, at::nullopt, (uint32_t)width,
                                                    (uint32_t)height, (uint32_t)tile_size, s[4], s[5], render_alphas, s[7],
                                                    v_rc, v_ra, absgrad);
    tensor_list out(12);
    out[0] = std::get<1>(r); out[1] = std::get<2>(r); out[2] = std::get<3>(r); out[3] = std::get<4>(r);
    if (ctx->needs_input_grad(4)) out[4] = (v_rc * (1.0 - render_alphas)).sum({1, 2});
    return out;
}

// ------------------------------------------------------------------------------------------------ fused SSIM
torch::Tensor FusedSSIMMap::forward(AutogradContext* ctx, double C1, double C2, torch::Tensor img1, torch::Tensor img2,
                                    std::string padding, bool train) {
    auto r = fusedssim((float)C1, (float)C2, img1, img2, train);
    auto m = std::get<0>(r);
    if (train) ctx->save_for_backward({img1, img2, std::get<1>(r), std::get<2>(r), std::get<3>(r)});
    ctx->saved_data["C1"] = C1;
    ctx->saved_data["C2"] = C2;
    ctx->saved_data["padding"] = padding;
    if (padding == "valid") m = m.slice(2, 5, -5).slice(3, 5, -5);
    return m;
}

tensor_list FusedSSIMMap::backward(AutogradContext* ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    TORCH_CHECK(s.size() == 5, "FusedSSIMMap: forward ran with train = false");
    const std::string padding = ctx->saved_data["padding"].toStringRef();
    auto dL = g[0];
    if (padding == "valid") {
        auto full = torch::zeros_like(s[0]);  // keeps img1's memory layout (planar, or the permuted interleaved view)
        full.slice(2, 5, -5).slice(3, 5, -5).copy_(dL);
        dL = full;
    }
    auto grad = fusedssim_backward((float)ctx->saved_data["C1"].toDouble(), (float)ctx->saved_data["C2"].toDouble(), s[0], s[1],
                                   dL, s[2], s[3], s[4]);
    return {torch::Tensor(), torch::Tensor(), grad, torch::Tensor(), torch::Tensor(), torch::Tensor()};
}

// ------------------------------------------------------------------------------------------------ binning
variable_list isectTilesNoDepth(torch::Tensor means2d, torch::Tensor radii, torch::Tensor depths, int tile_size,
                                int tile_width, int tile_height, bool sort) {
    auto t = gsplat::isect_tiles_tensor_no_depth(contig_f32(means2d, "means2d"), radii.contiguous(), depths.contiguous(),
                                                 at::nullopt, at::nullopt, (uint32_t)means2d.size(0), (uint32_t)tile_size,
                                                 (uint32_t)tile_width, (uint32_t)tile_height, sort, true);
    return {std::get<0>(t), std::get<1>(t), std::get<2>(t), std::get<3>(t), std::get<4>(t)};
}

torch::Tensor isectOffsetEncodeNoDepth(torch::Tensor isect_ids, int n_cameras, int tile_width, int tile_height) {
    return gsplat::isect_offset_encode_tensor_no_depth(isect_ids.contiguous(), (uint32_t)n_cameras, (uint32_t)tile_width,
                                                       (uint32_t)tile_height);
}

variable_list isectTiles(torch::Tensor means2d, torch::Tensor radii, torch::Tensor depths, int tile_size, int tile_width,
                         int tile_height, bool sort) {
    auto t = gsplat::isect_tiles_tensor(contig_f32(means2d, "means2d"), radii.contiguous(), contig_f32(depths, "depths"),
                                        at::nullopt, at::nullopt, (uint32_t)means2d.size(0), (uint32_t)tile_size,
                                        (uint32_t)tile_width, (uint32_t)tile_height, sort, true);
    return {std::get<0>(t), std::get<1>(t), std::get<2>(t)};
}

torch::Tensor isectOffsetEncode(torch::Tensor isect_ids, int n_cameras, int tile_width, int tile_height) {
    return gsplat::isect_offset_encode_tensor(isect_ids.contiguous(), (uint32_t)n_cameras, (uint32_t)tile_width,
                                              (uint32_t)tile_height);
}

// ------------------------------------------------------------------------------------------------ KNN, SH helpers
torch::Tensor simpleKNN(torch::Tensor points) { return distCUDA2(points); }

int degFromSh(int numBases) {
    switch (numBases) {
        case 1: return 0;
        case 4: return 1;
        case 9: return 2;
        case 16: return 3;
        default: return 4;
    }
}

int numShBases(int degree) {
    switch (degree) {
        case 0: return 1;
        case 1: return 4;
        case 2: return 9;
        case 3: return 16;
        default: return 25;
    }
}

static const double SH_C0 = 0.28209479177387814;

torch::Tensor rgb2sh(const torch::Tensor& rgb) { return (rgb - 0.5) / SH_C0; }

torch::Tensor sh2rgb(const torch::Tensor& sh) { return torch::clamp(sh * SH_C0 + 0.5, 0.0, 1.0); }

// See gsplat_wapper.hpp.  Every function allocates its outputs through libtorch (as the reference's launchers do with
// torch::empty / torch::zeros, gsplat/rasterizer/bindings.h:24-32) and hands raw pointers plus the current stream to
// the C-ABI; nothing is computed on the host.
#include "gsplat_wapper.hpp"

using namespace gpsh;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

namespace {

torch::Tensor contig_f32(const torch::Tensor& t, const char* name) {
    TORCH_CHECK(t.defined() && t.is_cuda(), name, " must be a device tensor");
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32");
    return t.contiguous();
}

// device int64[4] {n_isects, n_groups, overflow, n_visible} for exact-size operator-level tensors
torch::Tensor counts_for(int64_t n_isects, int64_t n_groups, const torch::Device& dev) {
    auto host = torch::empty({4}, torch::TensorOptions().dtype(torch::kInt64).pinned_memory(true));
    int64_t* h = host.data_ptr<int64_t>();
    h[0] = n_isects; h[1] = n_groups; h[2] = 0; h[3] = 0;
    return host.to(dev, /*non_blocking=*/true);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ SH
torch::Tensor SphericalHarmonicsNew::forward(AutogradContext* ctx, int sh_degree, torch::Tensor dirs,
                                             torch::Tensor coeffs, torch::Tensor masks) {
    dirs = contig_f32(dirs, "dirs");
    coeffs = contig_f32(coeffs, "coeffs");
    TORCH_CHECK(dirs.size(-1) == 3 && coeffs.size(-1) == 3, "dirs[...,3], coeffs[...,K,3]");
    const int K = (int)coeffs.size(-2);
    const int N = (int)(coeffs.numel() / (K * 3));
    torch::Tensor m;
    if (masks.defined()) m = masks.contiguous().to(torch::kUInt8);
    auto colors = torch::empty_like(dirs);
    check(gps_sh_fwd(N, K, sh_degree, fptr(dirs), fptr(coeffs), ptr<uint8_t>(m), fptr(colors), current_stream()),
          "gps_sh_fwd");
    ctx->save_for_backward({dirs, coeffs, m});
    ctx->saved_data["sh_degree"] = (int64_t)sh_degree;
    ctx->saved_data["K"] = (int64_t)K;
    ctx->saved_data["need_dirs"] = dirs.requires_grad();
    return colors;
}

tensor_list SphericalHarmonicsNew::backward(AutogradContext* ctx, tensor_list grad_outputs) {
    auto saved = ctx->get_saved_variables();
    const torch::Tensor &dirs = saved[0], &coeffs = saved[1], &m = saved[2];
    const int sh_degree = (int)ctx->saved_data["sh_degree"].toInt(), K = (int)ctx->saved_data["K"].toInt();
    const bool need_dirs = ctx->saved_data["need_dirs"].toBool();
    auto v_colors = contig_f32(grad_outputs[0], "v_colors");
    const int N = (int)(coeffs.numel() / (K * 3));
    auto v_coeffs = torch::empty_like(coeffs);
    torch::Tensor v_dirs;
    if (need_dirs) v_dirs = torch::empty_like(dirs);
    check(gps_sh_bwd(N, K, sh_degree, fptr(dirs), fptr(coeffs), ptr<uint8_t>(m), fptr(v_colors), fptr(v_coeffs),
                     fptr(v_dirs), current_stream()), "gps_sh_bwd");
    return {torch::Tensor(), v_dirs, v_coeffs, torch::Tensor()};
}

// ------------------------------------------------------------------------------------------------ projection
tensor_list FullyFusedProjection::forward(AutogradContext* ctx, torch::Tensor means, c10::optional<torch::Tensor> covars,
                                          torch::Tensor quats, torch::Tensor scales, torch::Tensor viewmats,
                                          torch::Tensor Ks, int width, int height, float eps2d, float near_plane,
                                          float far_plane, float radius_clip, bool calc_compensations,
                                          std::string camera_model) {
    TORCH_CHECK(!(covars.has_value() && covars->defined()) && !calc_compensations && camera_model == "pinhole",
                "gfx950 path implements the configuration GPS-SLAM ships: quats+scales, no compensations, pinhole "
                "(raw_gs_model.cpp:225-245)");
    means = contig_f32(means, "means"); quats = contig_f32(quats, "quats"); scales = contig_f32(scales, "scales");
    viewmats = contig_f32(viewmats, "viewmats"); Ks = contig_f32(Ks, "Ks");
    TORCH_CHECK(viewmats.size(0) == 1 && Ks.size(0) == 1, "single camera (C == 1), as raw_gs_model.cpp always passes");
    const int N = (int)means.size(0);
    const auto dev = means.device();
    auto radii = torch::empty({1, N}, i32(dev));
    auto means2d = torch::empty({1, N, 2}, f32(dev));
    auto depths = torch::empty({1, N}, f32(dev));
    auto conics = torch::empty({1, N, 3}, f32(dev));
    check(gps_proj_fwd(N, fptr(means), fptr(quats), fptr(scales), fptr(viewmats), fptr(Ks), width, height, eps2d,
                       near_plane, far_plane, radius_clip, iptr(radii), fptr(means2d), fptr(depths), fptr(conics),
                       current_stream()), "gps_proj_fwd");
    ctx->save_for_backward({means, quats, scales, viewmats, Ks, radii, conics});
    ctx->saved_data["width"] = (int64_t)width;
    ctx->saved_data["height"] = (int64_t)height;
    ctx->saved_data["eps2d"] = (double)eps2d;
    // compensations: never computed here (calc_compensations is rejected above); autograd wants defined outputs
    auto compensations = torch::empty({0}, f32(dev));
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
    auto v_means = torch::empty_like(means), v_quats = torch::empty_like(quats), v_scales = torch::empty_like(scales);
    check(gps_proj_bwd(N, fptr(means), fptr(quats), fptr(scales), fptr(viewmats), fptr(Ks), width, height, eps2d,
                       iptr(radii), fptr(conics), fptr(v_means2d), fptr(v_depths), fptr(v_conics), fptr(v_means),
                       fptr(v_quats), fptr(v_scales), current_stream()), "gps_proj_bwd");
    tensor_list out(14);
    out[0] = v_means; out[2] = v_quats; out[3] = v_scales;
    return out;
}

// ------------------------------------------------------------------------------------------------ rasterizer
tensor_list RasterizeToPixelsGes_NewParallel::forward(
    AutogradContext* ctx, torch::Tensor means2d, torch::Tensor conics, torch::Tensor colors, torch::Tensor opacities,
    torch::Tensor radiis, torch::Tensor ref_depth_map, torch::Tensor base_color_map,
    c10::optional<torch::Tensor> backgrounds, c10::optional<torch::Tensor> masks, int width, int height, int tile_size,
    torch::Tensor isect_offsets, torch::Tensor flatten_ids, torch::Tensor group_gs_ids, torch::Tensor group_starts,
    bool absgrad, float delta_depth) {
    TORCH_CHECK(!(backgrounds.has_value() && backgrounds->defined()) && !(masks.has_value() && masks->defined()) &&
                    !absgrad, "backgrounds / masks / absgrad are never used by GPS-SLAM and are not implemented");
    means2d = contig_f32(means2d, "means2d"); conics = contig_f32(conics, "conics");
    colors = contig_f32(colors, "colors"); opacities = contig_f32(opacities, "opacities");
    ref_depth_map = contig_f32(ref_depth_map, "ref_depth_map");
    TORCH_CHECK(colors.size(-1) == 4, "the ges path renders rgb + depth (raw_gs_model.cpp:286)");
    radiis = radiis.contiguous(); isect_offsets = isect_offsets.contiguous(); flatten_ids = flatten_ids.contiguous();
    group_gs_ids = group_gs_ids.contiguous(); group_starts = group_starts.contiguous();
    const int N = (int)opacities.numel();
    const auto dev = means2d.device();
    auto counts = counts_for(flatten_ids.numel(), group_gs_ids.numel(), dev);
    auto rc = torch::empty({1, height, width, 4}, f32(dev));
    auto ra = torch::empty({1, height, width, 1}, f32(dev));
    check(gps_raster_ges_fwd(N, fptr(means2d), fptr(conics), fptr(colors), fptr(opacities), fptr(ref_depth_map), width,
                             height, tile_size, iptr(isect_offsets), iptr(flatten_ids), ptr<int64_t>(counts),
                             delta_depth, fptr(rc), fptr(ra), nullptr, current_stream()), "gps_raster_ges_fwd");
    ctx->save_for_backward({means2d, conics, colors, opacities, radiis, ref_depth_map, group_gs_ids, group_starts, counts});
    ctx->saved_data["width"] = (int64_t)width;
    ctx->saved_data["height"] = (int64_t)height;
    ctx->saved_data["delta_depth"] = (double)delta_depth;
    return {rc, ra};
}

tensor_list RasterizeToPixelsGes_NewParallel::backward(AutogradContext* ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const torch::Tensor &means2d = s[0], &conics = s[1], &colors = s[2], &opacities = s[3], &radiis = s[4],
                        &ref_depth_map = s[5], &group_gs_ids = s[6], &group_starts = s[7], &counts = s[8];
    const int width = (int)ctx->saved_data["width"].toInt(), height = (int)ctx->saved_data["height"].toInt();
    const float delta_depth = (float)ctx->saved_data["delta_depth"].toDouble();
    const int N = (int)opacities.numel();
    auto v_rc = g[0].defined() ? contig_f32(g[0], "v_render_colors") : torch::zeros({1, height, width, 4}, means2d.options());
    auto v_ra = g[1].defined() ? contig_f32(g[1], "v_render_alphas") : torch::zeros({1, height, width, 1}, means2d.options());
    auto v_means2d = torch::empty_like(means2d), v_conics = torch::empty_like(conics);
    auto v_colors = torch::empty_like(colors), v_opacities = torch::empty_like(opacities);
    check(gps_raster_ges_bwd_gs(N, fptr(means2d), fptr(conics), fptr(colors), fptr(opacities), iptr(radiis),
                                fptr(ref_depth_map), width, height, iptr(group_gs_ids), iptr(group_starts),
                                ptr<int64_t>(counts), delta_depth, fptr(v_rc), fptr(v_ra), fptr(v_means2d),
                                fptr(v_conics), fptr(v_colors), fptr(v_opacities), 0, current_stream()),
          "gps_raster_ges_bwd_gs");
    tensor_list out(18);
    out[0] = v_means2d; out[1] = v_conics; out[2] = v_colors; out[3] = v_opacities;
    return out;
}

// ------------------------------------------------------------------------------------------------ raw rasterizer
tensor_list RasterizeToPixels::forward(AutogradContext* ctx, torch::Tensor means2d, torch::Tensor conics, torch::Tensor colors,
                                       torch::Tensor opacities, c10::optional<torch::Tensor> backgrounds,
                                       c10::optional<torch::Tensor> masks, int width, int height, int tile_size,
                                       torch::Tensor isect_offsets, torch::Tensor flatten_ids, bool absgrad) {
    TORCH_CHECK(!(masks.has_value() && masks->defined()), "tile masks are never used by GPS-SLAM and are not implemented");
    means2d = contig_f32(means2d, "means2d"); conics = contig_f32(conics, "conics");
    colors = contig_f32(colors, "colors"); opacities = contig_f32(opacities, "opacities");
    TORCH_CHECK(colors.size(-1) == 4, "the raw path renders rgb + depth (raw_gs_model.cpp:117)");
    TORCH_CHECK(means2d.size(0) == 1, "single camera (C == 1)");
    isect_offsets = isect_offsets.contiguous(); flatten_ids = flatten_ids.contiguous();
    torch::Tensor bg;
    if (backgrounds.has_value() && backgrounds->defined()) {
        bg = contig_f32(*backgrounds, "backgrounds");
        TORCH_CHECK(bg.numel() == 4, "backgrounds[1,4]");
    }
    const int N = (int)opacities.numel();
    const auto dev = means2d.device();
    auto counts = counts_for(flatten_ids.numel(), 0, dev);
    auto rc = torch::empty({1, height, width, 4}, f32(dev));
    auto ra = torch::empty({1, height, width, 1}, f32(dev));
    auto last = torch::empty({1, height, width}, i32(dev));
    check(gps_raster_raw_fwd(N, fptr(means2d), fptr(conics), fptr(colors), fptr(opacities), bg.defined() ? fptr(bg) : nullptr,
                             width, height, tile_size, iptr(isect_offsets), iptr(flatten_ids), ptr<int64_t>(counts), fptr(rc),
                             fptr(ra), iptr(last), current_stream()), "gps_raster_raw_fwd");
    ctx->save_for_backward({means2d, conics, colors, opacities, isect_offsets, flatten_ids, ra, last, counts});
    ctx->saved_data["width"] = (int64_t)width;
    ctx->saved_data["height"] = (int64_t)height;
    ctx->saved_data["tile_size"] = (int64_t)tile_size;
    ctx->saved_data["absgrad"] = absgrad;
    return {rc, ra};
}

tensor_list RasterizeToPixels::backward(AutogradContext* ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const torch::Tensor &means2d = s[0], &conics = s[1], &colors = s[2], &opacities = s[3], &isect_offsets = s[4],
                        &flatten_ids = s[5], &render_alphas = s[6], &last_ids = s[7], &counts = s[8];
    const int width = (int)ctx->saved_data["width"].toInt(), height = (int)ctx->saved_data["height"].toInt();
    const int tile_size = (int)ctx->saved_data["tile_size"].toInt();
    const bool absgrad = ctx->saved_data["absgrad"].toBool();
    const int N = (int)opacities.numel();
    auto v_rc = g[0].defined() ? contig_f32(g[0], "v_render_colors") : torch::zeros({1, height, width, 4}, means2d.options());
    auto v_ra = g[1].defined() ? contig_f32(g[1], "v_render_alphas") : torch::zeros({1, height, width, 1}, means2d.options());
    auto v_means2d = torch::empty_like(means2d), v_conics = torch::empty_like(conics);
    auto v_colors = torch::empty_like(colors), v_opacities = torch::empty_like(opacities);
    torch::Tensor v_abs;
    if (absgrad) v_abs = torch::empty_like(means2d);
    // backgrounds = NULL: the reference's backward never receives them (gsplat_wapper.hpp:307)
    check(gps_raster_raw_bwd(N, fptr(means2d), fptr(conics), fptr(colors), fptr(opacities), nullptr, width, height, tile_size,
                             iptr(isect_offsets), iptr(flatten_ids), ptr<int64_t>(counts), fptr(render_alphas),
                             iptr(last_ids), fptr(v_rc), fptr(v_ra), absgrad ? fptr(v_abs) : nullptr, fptr(v_means2d),
                             fptr(v_conics), fptr(v_colors), fptr(v_opacities), current_stream()), "gps_raster_raw_bwd");
    tensor_list out(12);
    out[0] = v_means2d; out[1] = v_conics; out[2] = v_colors; out[3] = v_opacities;
    if (ctx->needs_input_grad(4)) out[4] = (v_rc * (1.0 - render_alphas)).sum({1, 2});
    return out;
}

// ------------------------------------------------------------------------------------------------ fused SSIM
namespace {
bool channels_last_view(const torch::Tensor& t) {  // [B,CH,H,W] whose memory is a contiguous [B,H,W,CH] array
    return t.dim() == 4 && !t.is_contiguous() && t.permute({0, 2, 3, 1}).is_contiguous();
}
}  // namespace

torch::Tensor FusedSSIMMap::forward(AutogradContext* ctx, double C1, double C2, torch::Tensor img1, torch::Tensor img2,
                                    std::string padding, bool train) {
    TORCH_CHECK(img1.dim() == 4 && img1.sizes() == img2.sizes(), "img1 / img2: [B,CH,H,W]");
    const bool cl = channels_last_view(img1) && channels_last_view(img2);
    auto a = cl ? img1.permute({0, 2, 3, 1}) : contig_f32(img1, "img1");
    auto b = cl ? img2.permute({0, 2, 3, 1}) : contig_f32(img2, "img2");
    check_f32_dev(a, "img1"); check_f32_dev(b, "img2");
    const int B = (int)img1.size(0), CH = (int)img1.size(1), H = (int)img1.size(2), W = (int)img1.size(3);
    auto m = torch::empty_like(a);
    torch::Tensor d1, d2, d3;
    if (train) { d1 = torch::empty_like(a); d2 = torch::empty_like(a); d3 = torch::empty_like(a); }
    check(gps_ssim_fwd(B, CH, H, W, cl ? 1 : 0, (float)C1, (float)C2, fptr(a), fptr(b), fptr(m), train ? fptr(d1) : nullptr,
                       train ? fptr(d2) : nullptr, train ? fptr(d3) : nullptr, current_stream()), "gps_ssim_fwd");
    if (train) ctx->save_for_backward({a, b, d1, d2, d3});
    ctx->saved_data["padding"] = padding;
    ctx->saved_data["cl"] = cl;
    if (cl) m = m.permute({0, 3, 1, 2});
    if (padding == "valid") m = m.slice(2, 5, -5).slice(3, 5, -5);
    return m;
}

tensor_list FusedSSIMMap::backward(AutogradContext* ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    TORCH_CHECK(s.size() == 5, "FusedSSIMMap: forward ran with train = false");
    const torch::Tensor &a = s[0], &b = s[1];
    const bool cl = ctx->saved_data["cl"].toBool();
    const std::string padding = ctx->saved_data["padding"].toStringRef();
    const int B = (int)a.size(0), CH = (int)(cl ? a.size(3) : a.size(1)), H = (int)(cl ? a.size(1) : a.size(2)),
              W = (int)(cl ? a.size(2) : a.size(3));
    auto dL = g[0];
    if (padding == "valid") {
        auto full = torch::zeros({B, CH, H, W}, a.options());
        full.slice(2, 5, -5).slice(3, 5, -5).copy_(dL);
        dL = full;
    }
    if (cl) dL = dL.permute({0, 2, 3, 1});
    dL = dL.contiguous();
    auto grad = torch::empty_like(a);
    check(gps_ssim_bwd(B, CH, H, W, cl ? 1 : 0, fptr(a), fptr(b), fptr(dL), fptr(s[2]), fptr(s[3]), fptr(s[4]), fptr(grad),
                       current_stream()), "gps_ssim_bwd");
    if (cl) grad = grad.permute({0, 3, 1, 2});
    return {torch::Tensor(), torch::Tensor(), grad, torch::Tensor(), torch::Tensor(), torch::Tensor()};
}

// ------------------------------------------------------------------------------------------------ binning
variable_list isectTilesNoDepth(torch::Tensor means2d, torch::Tensor radii, torch::Tensor depths, int tile_size,
                                int tile_width, int tile_height, bool sort) {
    (void)depths;
    TORCH_CHECK(sort, "isectTilesNoDepth: the unsorted variant is never used by GPS-SLAM");
    means2d = contig_f32(means2d, "means2d");
    radii = radii.contiguous();
    TORCH_CHECK(radii.scalar_type() == torch::kInt32, "radii must be int32");
    const int N = (int)radii.numel();
    const auto dev = means2d.device();
    // capacity-sized scratch outputs, trimmed to the exact sizes the reference returns after ONE host read
    const int64_t icap = std::max<int64_t>(1 << 20, 16 * (int64_t)N), gcap = std::max<int64_t>(1 << 20, 32 * (int64_t)N);
    auto tiles_per_gauss = torch::empty({1, N}, i32(dev));
    auto isect_ids = torch::empty({icap}, i64(dev));
    auto flatten_ids = torch::empty({icap}, i32(dev));
    auto group_gs_ids = torch::empty({gcap}, i32(dev));
    auto group_starts = torch::empty({gcap}, i32(dev));
    auto offsets = torch::empty({1, tile_height, tile_width}, i32(dev));
    auto counts = torch::zeros({4}, i64(dev));
    const int64_t ws_bytes = gps_isect_workspace_bytes(N, icap);
    auto ws = torch::empty({ws_bytes}, u8(dev));
    check(gps_isect_tiles_no_depth(N, fptr(means2d), iptr(radii), tile_size, tile_width, tile_height, icap, gcap,
                                   iptr(tiles_per_gauss), ptr<int64_t>(isect_ids), iptr(flatten_ids),
                                   iptr(group_gs_ids), iptr(group_starts), iptr(offsets), ptr<int64_t>(counts),
                                   ws.data_ptr(), ws_bytes, current_stream()), "gps_isect_tiles_no_depth");
    auto c = counts.cpu();
    const int64_t* h = c.data_ptr<int64_t>();
    TORCH_CHECK(h[2] == 0, "isectTilesNoDepth: intersection capacity exceeded");
    using torch::indexing::Slice;
    return {tiles_per_gauss, isect_ids.index({Slice(0, h[0])}), flatten_ids.index({Slice(0, h[0])}),
            group_gs_ids.index({Slice(0, h[1])}), group_starts.index({Slice(0, h[1])})};
}

torch::Tensor isectOffsetEncodeNoDepth(torch::Tensor isect_ids, int n_cameras, int tile_width, int tile_height) {
    TORCH_CHECK(n_cameras == 1, "single camera (C == 1)");
    // offsets[t] = first position whose tile id is >= t (isect_tiles_no_depth.cu:373-425): a lower bound per tile
    auto tiles = torch::arange((int64_t)tile_width * tile_height, isect_ids.options());
    auto off = torch::searchsorted(isect_ids.contiguous(), tiles, /*out_int32=*/true, /*right=*/false);
    return off.view({1, tile_height, tile_width});
}

variable_list isectTiles(torch::Tensor means2d, torch::Tensor radii, torch::Tensor depths, int tile_size, int tile_width,
                         int tile_height, bool sort) {
    TORCH_CHECK(sort, "isectTiles: the unsorted variant is never used by GPS-SLAM");
    means2d = contig_f32(means2d, "means2d");
    depths = contig_f32(depths, "depths");
    radii = radii.contiguous();
    TORCH_CHECK(radii.scalar_type() == torch::kInt32, "radii must be int32");
    TORCH_CHECK(means2d.size(0) == 1, "single camera (C == 1)");
    const int N = (int)radii.numel();
    const auto dev = means2d.device();
    const int64_t icap = std::max<int64_t>(1 << 20, 16 * (int64_t)N);
    auto tiles_per_gauss = torch::empty({1, N}, i32(dev));
    auto isect_ids = torch::empty({icap}, i64(dev));
    auto flatten_ids = torch::empty({icap}, i32(dev));
    auto offsets = torch::empty({1, tile_height, tile_width}, i32(dev));
    auto counts = torch::zeros({4}, i64(dev));
    const int64_t ws_bytes = gps_isect_workspace_bytes(N, icap);
    auto ws = torch::empty({ws_bytes}, u8(dev));
    check(gps_isect_tiles(N, fptr(means2d), iptr(radii), fptr(depths), tile_size, tile_width, tile_height, icap,
                          iptr(tiles_per_gauss), ptr<int64_t>(isect_ids), iptr(flatten_ids), iptr(offsets),
                          ptr<int64_t>(counts), ws.data_ptr(), ws_bytes, current_stream()), "gps_isect_tiles");
    auto c = counts.cpu();
    const int64_t* h = c.data_ptr<int64_t>();
    TORCH_CHECK(h[2] == 0, "isectTiles: intersection capacity exceeded");
    using torch::indexing::Slice;
    return {tiles_per_gauss, isect_ids.index({Slice(0, h[0])}), flatten_ids.index({Slice(0, h[0])})};
}

torch::Tensor isectOffsetEncode(torch::Tensor isect_ids, int n_cameras, int tile_width, int tile_height) {
    TORCH_CHECK(n_cameras == 1, "single camera (C == 1)");
    // offsets[t] = first position whose key is >= (t << 32) (isect_tiles.cu:359-430)
    auto firsts = torch::arange((int64_t)tile_width * tile_height, isect_ids.options()) * ((int64_t)1 << 32);
    auto off = torch::searchsorted(isect_ids.contiguous(), firsts, /*out_int32=*/true, /*right=*/false);
    return off.view({1, tile_height, tile_width});
}

// ------------------------------------------------------------------------------------------------ KNN, SH helpers
torch::Tensor distCUDA2(const torch::Tensor& points_in) {
    auto points = contig_f32(points_in, "points");
    TORCH_CHECK(points.dim() == 2 && points.size(1) == 3, "points[P,3]");
    auto out = torch::empty({points.size(0)}, points.options());
    check(gps_knn_mean_dist2((int)points.size(0), fptr(points), fptr(out), current_stream()), "gps_knn_mean_dist2");
    return out;
}

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

// See hip_bindings.hpp.  Every launcher checks its inputs like the reference's GSPLAT_CHECK_INPUT (device, contiguous),
// allocates its outputs through libtorch (bindings.h:24-32: caching allocator) and hands raw pointers + the current HIP
// stream to the C-ABI.  Nothing is computed on the host.
#include "hip_bindings.hpp"

#include "gps_host_common.hpp"

using namespace gpsh;

namespace {

#define GPS_CHECK_INPUT(x)                                               \
    TORCH_CHECK((x).is_cuda(), #x " must be a CUDA tensor");             \
    TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")

inline bool has(const at::optional<torch::Tensor>& t) { return t.has_value() && t->defined(); }

inline void f32_input(const torch::Tensor& t, const char* name) {
    TORCH_CHECK(t.defined() && t.is_cuda(), name, " must be a CUDA tensor");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32");
}

// The kernels read n_isects / n_groups from a device int64[4] {n_isects, n_groups, overflow, n_visible}: at this level the
// sizes are the tensors' sizes, so the words are built on the host and travel with one small asynchronous upload.
torch::Tensor counts_for(int64_t n_isects, int64_t n_groups, const torch::Device& dev) {
    auto host = torch::empty({4}, torch::TensorOptions().dtype(torch::kInt64).pinned_memory(true));
    int64_t* h = host.data_ptr<int64_t>();
    h[0] = n_isects; h[1] = n_groups; h[2] = 0; h[3] = 0;
    return host.to(dev, /*non_blocking=*/true);
}

bool channels_last_view(const torch::Tensor& t) {  // [B,CH,H,W] whose memory is a contiguous [B,H,W,CH] array
    return t.dim() == 4 && !t.is_contiguous() && t.permute({0, 2, 3, 1}).is_contiguous();
}

struct Binned { torch::Tensor tiles_per_gauss, isect_ids, flatten_ids, group_gs_ids, group_starts; };

// capacity-sized scratch outputs, trimmed to the exact sizes the reference returns after ONE host read of the counts
// (the reference syncs twice, isect_tiles_no_depth.cu:238-239)
Binned bin_tiles(const torch::Tensor& means2d, const torch::Tensor& radii, const torch::Tensor* depths, uint32_t C,
                 uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, bool sort, const char* who) {
    TORCH_CHECK(C == 1, who, ": single camera (C == 1)");
    TORCH_CHECK(sort, who, ": the unsorted variant is never used by GPS-SLAM");
    f32_input(means2d, "means2d");
    GPS_CHECK_INPUT(radii);
    TORCH_CHECK(radii.scalar_type() == torch::kInt32, "radii must be int32");
    const int N = (int)radii.numel();
    const auto dev = means2d.device();
    // capacity-sized buffers, the counts come back with ONE host read; if they did not fit (the sticky overflow word), the
    // call is repeated with doubled capacities -- the reference sizes its buffers exactly and never fails here
    int64_t icap = std::max<int64_t>(1 << 20, 16 * (int64_t)N), gcap = std::max<int64_t>(1 << 20, 32 * (int64_t)N);
    Binned b;
    b.tiles_per_gauss = torch::empty({1, N}, i32(dev));
    auto offsets = torch::empty({1, (int64_t)tile_height, (int64_t)tile_width}, i32(dev));
    torch::Tensor isect_ids, flatten_ids, ggs, gst, c;
    const int64_t* h = nullptr;
    if (depths) f32_input(*depths, "depths");
    for (int attempt = 0;; attempt++) {
        isect_ids = torch::empty({icap}, i64(dev));
        flatten_ids = torch::empty({icap}, i32(dev));
        auto counts = torch::zeros({4}, i64(dev));
        const int64_t ws_bytes = gps_isect_workspace_bytes(N, icap);
        auto ws = torch::empty({ws_bytes}, u8(dev));
        if (depths) {
            check(gps_isect_tiles(N, fptr(means2d), iptr(radii), fptr(*depths), (int)tile_size, (int)tile_width, (int)tile_height,
                                  icap, iptr(b.tiles_per_gauss), ptr<int64_t>(isect_ids), iptr(flatten_ids), iptr(offsets),
                                  ptr<int64_t>(counts), ws.data_ptr(), ws_bytes, current_stream()), "gps_isect_tiles");
        } else {
            ggs = torch::empty({gcap}, i32(dev));
            gst = torch::empty({gcap}, i32(dev));
            check(gps_isect_tiles_no_depth(N, fptr(means2d), iptr(radii), (int)tile_size, (int)tile_width, (int)tile_height, icap,
                                           gcap, iptr(b.tiles_per_gauss), ptr<int64_t>(isect_ids), iptr(flatten_ids), iptr(ggs),
                                           iptr(gst), iptr(offsets), ptr<int64_t>(counts), ws.data_ptr(), ws_bytes,
                                           current_stream()), "gps_isect_tiles_no_depth");
        }
        c = counts.cpu();
        h = c.data_ptr<int64_t>();
        if (h[2] == 0) break;
        TORCH_CHECK(attempt < 10 && icap < ((int64_t)1 << 29), who, ": intersection capacity exceeded");
        icap *= 2; gcap *= 2;
    }
    using torch::indexing::Slice;
    b.isect_ids = isect_ids.index({Slice(0, h[0])});
    b.flatten_ids = flatten_ids.index({Slice(0, h[0])});
    if (!depths) {
        b.group_gs_ids = ggs.index({Slice(0, h[1])});
        b.group_starts = gst.index({Slice(0, h[1])});
    }
    return b;
}

}  // namespace

namespace gsplat {

// ------------------------------------------------------------------------------------------------ projection
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> fully_fused_projection_fwd_tensor(
    const torch::Tensor& means, const at::optional<torch::Tensor>& covars, const at::optional<torch::Tensor>& quats,
    const at::optional<torch::Tensor>& scales, const torch::Tensor& viewmats, const torch::Tensor& Ks,
    const uint32_t image_width, const uint32_t image_height, const float eps2d, const float near_plane,
    const float far_plane, const float radius_clip, const bool calc_compensations, const CameraModelType camera_model) {
    TORCH_CHECK(!has(covars) && has(quats) && has(scales) && !calc_compensations && camera_model == PINHOLE,
                "gfx950 path implements the configuration GPS-SLAM ships: quats+scales, no compensations, pinhole "
                "(raw_gs_model.cpp:225-245)");
    f32_input(means, "means"); f32_input(*quats, "quats"); f32_input(*scales, "scales");
    f32_input(viewmats, "viewmats"); f32_input(Ks, "Ks");
    TORCH_CHECK(viewmats.size(0) == 1 && Ks.size(0) == 1, "single camera (C == 1), as raw_gs_model.cpp always passes");
    const int N = (int)means.size(0);
    const auto dev = means.device();
    auto radii = torch::empty({1, N}, i32(dev));
    auto means2d = torch::empty({1, N, 2}, f32(dev));
    auto depths = torch::empty({1, N}, f32(dev));
    auto conics = torch::empty({1, N, 3}, f32(dev));
    check(gps_proj_fwd(N, fptr(means), fptr(*quats), fptr(*scales), fptr(viewmats), fptr(Ks), (int)image_width,
                       (int)image_height, eps2d, near_plane, far_plane, radius_clip, iptr(radii), fptr(means2d), fptr(depths),
                       fptr(conics), current_stream()), "gps_proj_fwd");
    return std::make_tuple(radii, means2d, depths, conics, torch::Tensor());
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> fully_fused_projection_bwd_tensor(
    const torch::Tensor& means, const at::optional<torch::Tensor>& covars, const at::optional<torch::Tensor>& quats,
    const at::optional<torch::Tensor>& scales, const torch::Tensor& viewmats, const torch::Tensor& Ks,
    const uint32_t image_width, const uint32_t image_height, const float eps2d, const CameraModelType camera_model,
    const torch::Tensor& radii, const torch::Tensor& conics, const at::optional<torch::Tensor>& compensations,
    const torch::Tensor& v_means2d, const torch::Tensor& v_depths, const torch::Tensor& v_conics,
    const at::optional<torch::Tensor>& v_compensations, const bool viewmats_requires_grad) {
    TORCH_CHECK(!has(covars) && has(quats) && has(scales) && !has(compensations) && !has(v_compensations) &&
                    camera_model == PINHOLE && !viewmats_requires_grad,
                "gfx950 path: quats+scales, no compensations, pinhole, fixed camera (raw_gs_model.cpp:225-245)");
    f32_input(means, "means"); f32_input(*quats, "quats"); f32_input(*scales, "scales");
    f32_input(viewmats, "viewmats"); f32_input(Ks, "Ks"); f32_input(conics, "conics");
    f32_input(v_means2d, "v_means2d"); f32_input(v_depths, "v_depths"); f32_input(v_conics, "v_conics");
    GPS_CHECK_INPUT(radii);
    const int N = (int)means.size(0);
    auto v_means = torch::empty_like(means), v_quats = torch::empty_like(*quats), v_scales = torch::empty_like(*scales);
    check(gps_proj_bwd(N, fptr(means), fptr(*quats), fptr(*scales), fptr(viewmats), fptr(Ks), (int)image_width,
                       (int)image_height, eps2d, iptr(radii), fptr(conics), fptr(v_means2d), fptr(v_depths), fptr(v_conics),
                       fptr(v_means), fptr(v_quats), fptr(v_scales), current_stream()), "gps_proj_bwd");
    return std::make_tuple(v_means, torch::Tensor(), v_quats, v_scales, torch::Tensor());
}

// ------------------------------------------------------------------------------------------------ binning
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> isect_tiles_tensor(
    const torch::Tensor& means2d, const torch::Tensor& radii, const torch::Tensor& depths,
    const at::optional<torch::Tensor>& camera_ids, const at::optional<torch::Tensor>& gaussian_ids, const uint32_t C,
    const uint32_t tile_size, const uint32_t tile_width, const uint32_t tile_height, const bool sort,
    const bool double_buffer) {
    (void)double_buffer;
    TORCH_CHECK(!has(camera_ids) && !has(gaussian_ids), "isect_tiles: packed mode is never used by GPS-SLAM");
    Binned b = bin_tiles(means2d, radii, &depths, C, tile_size, tile_width, tile_height, sort, "isect_tiles");
    return std::make_tuple(b.tiles_per_gauss, b.isect_ids, b.flatten_ids);
}

torch::Tensor isect_offset_encode_tensor(const torch::Tensor& isect_ids, const uint32_t C, const uint32_t tile_width,
                                         const uint32_t tile_height) {
    TORCH_CHECK(C == 1, "single camera (C == 1)");
    GPS_CHECK_INPUT(isect_ids);
    // offsets[t] = first position whose key is >= (t << 32) (isect_tiles.cu:359-430)
    auto firsts = torch::arange((int64_t)tile_width * tile_height, isect_ids.options()) * ((int64_t)1 << 32);
    auto off = torch::searchsorted(isect_ids, firsts, /*out_int32=*/true, /*right=*/false);
    return off.view({1, (int64_t)tile_height, (int64_t)tile_width});
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> isect_tiles_tensor_no_depth(
    const torch::Tensor& means2d, const torch::Tensor& radii, const torch::Tensor& depths,
    const at::optional<torch::Tensor>& camera_ids, const at::optional<torch::Tensor>& gaussian_ids, const uint32_t C,
    const uint32_t tile_size, const uint32_t tile_width, const uint32_t tile_height, const bool sort,
    const bool double_buffer) {
    (void)depths; (void)double_buffer;  // the no-depth keys are tile ids only (isect_tiles_no_depth.cu:57-129)
    TORCH_CHECK(!has(camera_ids) && !has(gaussian_ids), "isect_tiles_no_depth: packed mode is never used by GPS-SLAM");
    Binned b = bin_tiles(means2d, radii, nullptr, C, tile_size, tile_width, tile_height, sort, "isect_tiles_no_depth");
    return std::make_tuple(b.tiles_per_gauss, b.isect_ids, b.flatten_ids, b.group_gs_ids, b.group_starts);
}

torch::Tensor isect_offset_encode_tensor_no_depth(const torch::Tensor& isect_ids, const uint32_t C,
                                                  const uint32_t tile_width, const uint32_t tile_height) {
    TORCH_CHECK(C == 1, "single camera (C == 1)");
    GPS_CHECK_INPUT(isect_ids);
    // offsets[t] = first position whose tile id is >= t (isect_tiles_no_depth.cu:373-425): a lower bound per tile
    auto tiles = torch::arange((int64_t)tile_width * tile_height, isect_ids.options());
    auto off = torch::searchsorted(isect_ids, tiles, /*out_int32=*/true, /*right=*/false);
    return off.view({1, (int64_t)tile_height, (int64_t)tile_width});
}

// ------------------------------------------------------------------------------------------------ raw rasterizer
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> rasterize_to_pixels_fwd_tensor(
    const torch::Tensor& means2d, const torch::Tensor& conics, const torch::Tensor& colors, const torch::Tensor& opacities,
    const at::optional<torch::Tensor>& backgrounds, const at::optional<torch::Tensor>& mask, const uint32_t image_width,
    const uint32_t image_height, const uint32_t tile_size, const torch::Tensor& tile_offsets,
    const torch::Tensor& flatten_ids) {
    TORCH_CHECK(!has(mask), "tile masks are never used by GPS-SLAM and are not implemented");
    f32_input(means2d, "means2d"); f32_input(conics, "conics"); f32_input(colors, "colors"); f32_input(opacities, "opacities");
    GPS_CHECK_INPUT(tile_offsets); GPS_CHECK_INPUT(flatten_ids);
    TORCH_CHECK(colors.size(-1) == 4, "the raw path renders rgb + depth (raw_gs_model.cpp:117)");
    TORCH_CHECK(tile_offsets.size(0) == 1, "single camera (C == 1)");
    torch::Tensor bg;
    if (has(backgrounds)) {
        bg = *backgrounds;
        f32_input(bg, "backgrounds");
        TORCH_CHECK(bg.numel() == 4, "backgrounds[1,4]");
    }
    const int N = (int)opacities.numel(), W = (int)image_width, H = (int)image_height;
    const auto dev = means2d.device();
    auto counts = counts_for(flatten_ids.numel(), 0, dev);
    auto rc = torch::empty({1, H, W, 4}, f32(dev));
    auto ra = torch::empty({1, H, W, 1}, f32(dev));
    auto last = torch::empty({1, H, W}, i32(dev));
    check(gps_raster_raw_fwd(N, fptr(means2d), fptr(conics), fptr(colors), fptr(opacities), bg.defined() ? fptr(bg) : nullptr,
                             W, H, (int)tile_size, iptr(tile_offsets), iptr(flatten_ids), ptr<int64_t>(counts), fptr(rc),
                             fptr(ra), iptr(last), current_stream()), "gps_raster_raw_fwd");
    return std::make_tuple(rc, ra, last);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> rasterize_to_pixels_bwd_tensor(
    const torch::Tensor& means2d, const torch::Tensor& conics, const torch::Tensor& colors, const torch::Tensor& opacities,
    const at::optional<torch::Tensor>& backgrounds, const at::optional<torch::Tensor>& mask, const uint32_t image_width,
    const uint32_t image_height, const uint32_t tile_size, const torch::Tensor& tile_offsets,
    const torch::Tensor& flatten_ids, const torch::Tensor& render_alphas, const torch::Tensor& last_ids,
    const torch::Tensor& v_render_colors, const torch::Tensor& v_render_alphas, bool absgrad) {
    TORCH_CHECK(!has(mask), "tile masks are never used by GPS-SLAM and are not implemented");
    f32_input(means2d, "means2d"); f32_input(conics, "conics"); f32_input(colors, "colors"); f32_input(opacities, "opacities");
    f32_input(render_alphas, "render_alphas"); f32_input(v_render_colors, "v_render_colors");
    f32_input(v_render_alphas, "v_render_alphas");
    GPS_CHECK_INPUT(tile_offsets); GPS_CHECK_INPUT(flatten_ids); GPS_CHECK_INPUT(last_ids);
    TORCH_CHECK(colors.size(-1) == 4, "the raw path renders rgb + depth (raw_gs_model.cpp:117)");
    torch::Tensor bg;
    if (has(backgrounds)) { bg = *backgrounds; f32_input(bg, "backgrounds"); }
    const int N = (int)opacities.numel();
    auto counts = counts_for(flatten_ids.numel(), 0, means2d.device());
    auto v_means2d = torch::empty_like(means2d), v_conics = torch::empty_like(conics);
    auto v_colors = torch::empty_like(colors), v_opacities = torch::empty_like(opacities);
    torch::Tensor v_abs;
    if (absgrad) v_abs = torch::empty_like(means2d);
    check(gps_raster_raw_bwd(N, fptr(means2d), fptr(conics), fptr(colors), fptr(opacities), bg.defined() ? fptr(bg) : nullptr,
                             (int)image_width, (int)image_height, (int)tile_size, iptr(tile_offsets), iptr(flatten_ids),
                             ptr<int64_t>(counts), fptr(render_alphas), iptr(last_ids), fptr(v_render_colors),
                             fptr(v_render_alphas), absgrad ? fptr(v_abs) : nullptr, fptr(v_means2d), fptr(v_conics),
                             fptr(v_colors), fptr(v_opacities), current_stream()), "gps_raster_raw_bwd");
    return std::make_tuple(v_abs, v_means2d, v_conics, v_colors, v_opacities);
}

// ------------------------------------------------------------------------------------------------ ges rasterizer
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> rasterize_to_pixels_fwd_ges_tensor(
    const torch::Tensor& means2d, const torch::Tensor& conics, const torch::Tensor& colors, const torch::Tensor& opacities,
    const torch::Tensor& ref_depth_map, const torch::Tensor& base_color_map, const at::optional<torch::Tensor>& backgrounds,
    const at::optional<torch::Tensor>& mask, const uint32_t image_width, const uint32_t image_height,
    const uint32_t tile_size, const torch::Tensor& tile_offsets, const torch::Tensor& flatten_ids, const float delta_depth) {
    TORCH_CHECK(!has(backgrounds) && !has(mask), "backgrounds / masks are never used by the ges path and are not implemented");
    f32_input(means2d, "means2d"); f32_input(conics, "conics"); f32_input(colors, "colors"); f32_input(opacities, "opacities");
    f32_input(ref_depth_map, "ref_depth_map");
    GPS_CHECK_INPUT(base_color_map);  // unused by the kernel, exactly as in the reference (rasterize_to_pixels_fwd_ges.cu:67-215)
    GPS_CHECK_INPUT(tile_offsets); GPS_CHECK_INPUT(flatten_ids);
    TORCH_CHECK(colors.size(-1) == 4, "the ges path renders rgb + depth (raw_gs_model.cpp:286)");
    TORCH_CHECK(tile_offsets.size(0) == 1, "single camera (C == 1)");
    const int N = (int)opacities.numel(), W = (int)image_width, H = (int)image_height;
    const auto dev = means2d.device();
    auto counts = counts_for(flatten_ids.numel(), 0, dev);
    auto rc = torch::empty({1, H, W, 4}, f32(dev));
    auto ra = torch::empty({1, H, W, 1}, f32(dev));
    auto last = torch::empty({1, H, W}, i32(dev));
    check(gps_raster_ges_fwd(N, fptr(means2d), fptr(conics), fptr(colors), fptr(opacities), fptr(ref_depth_map), W, H,
                             (int)tile_size, iptr(tile_offsets), iptr(flatten_ids), ptr<int64_t>(counts), delta_depth,
                             fptr(rc), fptr(ra), iptr(last), current_stream()), "gps_raster_ges_fwd");
    return std::make_tuple(rc, ra, last);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> rasterize_to_pixels_bwd_ges_tensor(
    const torch::Tensor& means2d, const torch::Tensor& conics, const torch::Tensor& colors, const torch::Tensor& opacities,
    const torch::Tensor& ref_depth_map, const torch::Tensor& base_color_map, const at::optional<torch::Tensor>& backgrounds,
    const at::optional<torch::Tensor>& mask, const uint32_t image_width, const uint32_t image_height,
    const uint32_t tile_size, const torch::Tensor& tile_offsets, const torch::Tensor& flatten_ids, const float delta_depth,
    const torch::Tensor& render_alphas, const torch::Tensor& last_ids, const torch::Tensor& v_render_colors,
    const torch::Tensor& v_render_alphas, bool absgrad) {
    (void)base_color_map; (void)render_alphas; (void)last_ids;  // not read by the reference kernel either (:83-291)
    TORCH_CHECK(!has(backgrounds) && !has(mask) && !absgrad,
                "backgrounds / masks / absgrad are never used by the ges path and are not implemented");
    f32_input(means2d, "means2d"); f32_input(conics, "conics"); f32_input(colors, "colors"); f32_input(opacities, "opacities");
    f32_input(ref_depth_map, "ref_depth_map"); f32_input(v_render_colors, "v_render_colors");
    f32_input(v_render_alphas, "v_render_alphas");
    GPS_CHECK_INPUT(tile_offsets); GPS_CHECK_INPUT(flatten_ids);
    TORCH_CHECK(colors.size(-1) == 4, "the ges path renders rgb + depth (raw_gs_model.cpp:286)");
    const int N = (int)opacities.numel();
    auto counts = counts_for(flatten_ids.numel(), 0, means2d.device());
    auto v_means2d = torch::empty_like(means2d), v_conics = torch::empty_like(conics);
    auto v_colors = torch::empty_like(colors), v_opacities = torch::empty_like(opacities);
    check(gps_raster_ges_bwd_exact(N, fptr(means2d), fptr(conics), fptr(colors), fptr(opacities), fptr(ref_depth_map),
                                   (int)image_width, (int)image_height, (int)tile_size, iptr(tile_offsets), iptr(flatten_ids),
                                   ptr<int64_t>(counts), delta_depth, fptr(v_render_colors), fptr(v_render_alphas),
                                   fptr(v_means2d), fptr(v_conics), fptr(v_colors), fptr(v_opacities), current_stream()),
          "gps_raster_ges_bwd_exact");
    return std::make_tuple(torch::Tensor(), v_means2d, v_conics, v_colors, v_opacities);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterize_to_pixels_bwd_ges_gs_parallel_tensor(
    const torch::Tensor& means2d, const torch::Tensor& conics, const torch::Tensor& colors, const torch::Tensor& opacities,
    const torch::Tensor& radiis, const torch::Tensor& ref_depth_map, const torch::Tensor& base_color_map,
    const at::optional<torch::Tensor>& backgrounds, const uint32_t image_width, const uint32_t image_height,
    const uint32_t n_isects, const torch::Tensor& group_gs_ids, const torch::Tensor& group_starts, const float delta_depth,
    const torch::Tensor& render_alphas, const torch::Tensor& v_render_colors, const torch::Tensor& v_render_alphas,
    bool absgrad) {
    (void)base_color_map; (void)render_alphas; (void)n_isects;  // carried through the API, not read by the kernel (:47-200)
    TORCH_CHECK(!has(backgrounds) && !absgrad, "backgrounds / absgrad are never used by the ges path and are not implemented");
    f32_input(means2d, "means2d"); f32_input(conics, "conics"); f32_input(colors, "colors"); f32_input(opacities, "opacities");
    f32_input(ref_depth_map, "ref_depth_map"); f32_input(v_render_colors, "v_render_colors");
    f32_input(v_render_alphas, "v_render_alphas");
    GPS_CHECK_INPUT(radiis); GPS_CHECK_INPUT(group_gs_ids); GPS_CHECK_INPUT(group_starts);
    TORCH_CHECK(radiis.scalar_type() == torch::kInt32, "radiis must be int32");
    TORCH_CHECK(colors.size(-1) == 4, "the ges path renders rgb + depth (raw_gs_model.cpp:286)");
    const int N = (int)opacities.numel();
    auto counts = counts_for(0, group_gs_ids.numel(), means2d.device());
    auto v_means2d = torch::empty_like(means2d), v_conics = torch::empty_like(conics);
    auto v_colors = torch::empty_like(colors), v_opacities = torch::empty_like(opacities);
    check(gps_raster_ges_bwd_gs(N, fptr(means2d), fptr(conics), fptr(colors), fptr(opacities), iptr(radiis),
                                fptr(ref_depth_map), (int)image_width, (int)image_height, iptr(group_gs_ids),
                                iptr(group_starts), ptr<int64_t>(counts), delta_depth, fptr(v_render_colors),
                                fptr(v_render_alphas), fptr(v_means2d), fptr(v_conics), fptr(v_colors), fptr(v_opacities), 0,
                                current_stream()), "gps_raster_ges_bwd_gs");
    return std::make_tuple(torch::Tensor(), v_means2d, v_conics, v_colors, v_opacities);
}

// ------------------------------------------------------------------------------------------------ SH
torch::Tensor compute_sh_fwd_tensor(const uint32_t degrees_to_use, const torch::Tensor& dirs, const torch::Tensor& coeffs,
                                    const at::optional<torch::Tensor> masks) {
    f32_input(dirs, "dirs"); f32_input(coeffs, "coeffs");
    TORCH_CHECK(dirs.size(-1) == 3 && coeffs.size(-1) == 3, "dirs[...,3], coeffs[...,K,3]");
    const int K = (int)coeffs.size(-2);
    const int N = (int)(coeffs.numel() / (K * 3));
    torch::Tensor m;
    if (has(masks)) m = masks->contiguous().to(torch::kUInt8);
    auto colors = torch::empty_like(dirs);
    check(gps_sh_fwd(N, K, (int)degrees_to_use, fptr(dirs), fptr(coeffs), ptr<uint8_t>(m), fptr(colors), current_stream()),
          "gps_sh_fwd");
    return colors;
}

std::tuple<torch::Tensor, torch::Tensor> compute_sh_bwd_tensor(const uint32_t K, const uint32_t degrees_to_use,
                                                               const torch::Tensor& dirs, const torch::Tensor& coeffs,
                                                               const at::optional<torch::Tensor> masks,
                                                               const torch::Tensor& v_colors, bool compute_v_dirs) {
    f32_input(dirs, "dirs"); f32_input(coeffs, "coeffs"); f32_input(v_colors, "v_colors");
    TORCH_CHECK((int64_t)K == coeffs.size(-2), "K must be coeffs.size(-2)");
    const int N = (int)(coeffs.numel() / ((int64_t)K * 3));
    torch::Tensor m;
    if (has(masks)) m = masks->contiguous().to(torch::kUInt8);
    auto v_coeffs = torch::empty_like(coeffs);
    torch::Tensor v_dirs;
    if (compute_v_dirs) v_dirs = torch::empty_like(dirs);
    check(gps_sh_bwd(N, (int)K, (int)degrees_to_use, fptr(dirs), fptr(coeffs), ptr<uint8_t>(m), fptr(v_colors), fptr(v_coeffs),
                     fptr(v_dirs), current_stream()), "gps_sh_bwd");
    return std::make_tuple(v_coeffs, v_dirs);
}

}  // namespace gsplat

// ------------------------------------------------------------------------------------------------ fused SSIM
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> fusedssim(float C1, float C2, torch::Tensor& img1,
                                                                                 torch::Tensor& img2, bool train) {
    TORCH_CHECK(img1.dim() == 4 && img1.sizes() == img2.sizes(), "img1 / img2: [B,CH,H,W]");
    const bool cl = channels_last_view(img1) && channels_last_view(img2);
    auto a = cl ? img1.permute({0, 2, 3, 1}) : img1.contiguous();
    auto b = cl ? img2.permute({0, 2, 3, 1}) : img2.contiguous();
    f32_input(a, "img1"); f32_input(b, "img2");
    const int B = (int)img1.size(0), CH = (int)img1.size(1), H = (int)img1.size(2), W = (int)img1.size(3);
    auto m = torch::empty_like(a);
    torch::Tensor d1, d2, d3;
    if (train) { d1 = torch::empty_like(a); d2 = torch::empty_like(a); d3 = torch::empty_like(a); }
    check(gps_ssim_fwd(B, CH, H, W, cl ? 1 : 0, C1, C2, fptr(a), fptr(b), fptr(m), train ? fptr(d1) : nullptr,
                       train ? fptr(d2) : nullptr, train ? fptr(d3) : nullptr, current_stream()), "gps_ssim_fwd");
    if (cl) {  // hand the maps back in the callers' [B,CH,H,W] indexing (views of the interleaved memory)
        m = m.permute({0, 3, 1, 2});
        if (train) { d1 = d1.permute({0, 3, 1, 2}); d2 = d2.permute({0, 3, 1, 2}); d3 = d3.permute({0, 3, 1, 2}); }
    }
    return std::make_tuple(m, d1, d2, d3);
}

torch::Tensor fusedssim_backward(float C1, float C2, torch::Tensor& img1, torch::Tensor& img2, torch::Tensor& dL_dmap,
                                 torch::Tensor& dm_dmu1, torch::Tensor& dm_dsigma1_sq, torch::Tensor& dm_dsigma12) {
    (void)C1; (void)C2;  // already folded into the saved derivative maps (ssim.cu:305-383 ignores them as well)
    TORCH_CHECK(img1.dim() == 4 && img1.sizes() == img2.sizes(), "img1 / img2: [B,CH,H,W]");
    const bool cl = channels_last_view(img1) && channels_last_view(img2) && channels_last_view(dm_dmu1);
    auto lay = [&](const torch::Tensor& t) { return cl ? t.permute({0, 2, 3, 1}).contiguous() : t.contiguous(); };
    auto a = lay(img1), b = lay(img2), dL = lay(dL_dmap), d1 = lay(dm_dmu1), d2 = lay(dm_dsigma1_sq), d3 = lay(dm_dsigma12);
    f32_input(a, "img1"); f32_input(dL, "dL_dmap"); f32_input(d1, "dm_dmu1");
    const int B = (int)img1.size(0), CH = (int)img1.size(1), H = (int)img1.size(2), W = (int)img1.size(3);
    auto grad = torch::empty_like(a);
    check(gps_ssim_bwd(B, CH, H, W, cl ? 1 : 0, fptr(a), fptr(b), fptr(dL), fptr(d1), fptr(d2), fptr(d3), fptr(grad),
                       current_stream()), "gps_ssim_bwd");
    return cl ? grad.permute({0, 3, 1, 2}) : grad;
}

// ------------------------------------------------------------------------------------------------ KNN
torch::Tensor distCUDA2(const torch::Tensor& points_in) {
    TORCH_CHECK(points_in.defined() && points_in.is_cuda() && points_in.scalar_type() == torch::kFloat32,
                "points must be a float32 device tensor");
    auto points = points_in.contiguous();
    TORCH_CHECK(points.dim() == 2 && points.size(1) == 3, "points[P,3]");
    auto out = torch::empty({points.size(0)}, points.options());
    const int P = (int)points.size(0);
    if (P > GPS_KNN_GRID_MIN_POINTS) {   // exact uniform-grid search (same bits as the brute force, sub-quadratic)
        const int64_t nbytes = gps_knn_grid_workspace_bytes(P);
        auto ws = torch::empty({nbytes}, points.options().dtype(torch::kUInt8));
        check(gps_knn_mean_dist2_grid(P, fptr(points), fptr(out), ws.data_ptr(), nbytes, current_stream()), "gps_knn_mean_dist2_grid");
    } else {
        check(gps_knn_mean_dist2(P, fptr(points), fptr(out), current_stream()), "gps_knn_mean_dist2");
    }
    return out;
}

#include "raw_gs_model.hpp"

using namespace gpsh;

// RawGaussianModel::initOptimizers computes eps and the betas in FLOAT variables (raw_gs_model.cpp:661-664: float eps = 1e-15 /
// sqrt(BS), float B1 = 1 - BS * (1 - 0.9), float B2 = 1 - BS * (1 - 0.999), BS = 1) and AdamOptions widens them: the optimiser
// runs with beta1 = 0.8999999761581421, beta2 = 0.9990000128746033 (1 - beta2 is 1.3e-5 smaller than 0.001), eps =
// 1.0000000036274937e-15.  Pinned by tests/test_adam_libtorch_gpu.py against torch::optim::Adam built the reference's way.
static const double kAdamBeta1 = (double)0.9f, kAdamBeta2 = (double)0.999f, kAdamEps = (double)1e-15f;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;
using torch::indexing::Slice;

// Parameter order everywhere in this file = RawGaussianParams order:
//   0 means, 1 scales(log), 2 quats, 3 featuresDc, 4 featuresRest, 5 opacities(logit)
// gps_splat_step::lr order = means, log_scales, quats, sh_dc, sh_rest, opac_logit (the same).

void RawGaussianModel::loadConfig(const Config& c) {
    maxSH = (int)c.get("sh_degree", maxSH);
    degreesToUse = maxSH;
    shDegreeInterval = (int)c.get("sh_degree_interval", shDegreeInterval);
    max_gs_radii = (int)c.get("max_gs_radii", max_gs_radii);
    delta_depth = (float)c.get("delta_depth", delta_depth);
    maxInitScale = (float)c.get("max_init_scale", maxInitScale);
    minInitScale = (float)c.get("min_init_scale", minInitScale);
    defaultOpacities = (float)c.get("default_opacities", defaultOpacities);
    means_lr = c.get("means_lr", means_lr); scales_lr = c.get("scales_lr", scales_lr);
    quats_lr = c.get("quats_lr", quats_lr); featuresDc_lr = c.get("featuresDc_lr", featuresDc_lr);
    featuresRest_lr = c.get("featuresRest_lr", featuresRest_lr); opacities_lr = c.get("opacities_lr", opacities_lr);
    isect_capacity = (int64_t)c.get("isect_capacity", (double)isect_capacity);
    fuse_sh_rest_adam = c.get("fuse_sh_rest_adam", fuse_sh_rest_adam ? 1.0 : 0.0) != 0.0;
    strip_backward = c.get("strip_backward", strip_backward ? 1.0 : 0.0) != 0.0;
    render_method = c.gets("render_method", render_method);
    const int64_t cap = (int64_t)c.get("capacity", 1 << 19);
    opt_gs_params.reserve(cap, numShBases(maxSH), device);
}

void RawGaussianModel::updateSH(int curr_iter) {
    if (curr_iter >= 0 && shDegreeInterval > 0) degreesToUse = std::min<int>(maxSH, curr_iter / shDegreeInterval);
    else degreesToUse = maxSH;
}

torch::Tensor RawGaussianModel::clampRefDepth(const torch::Tensor& ref_depth) {
    return torch::where(ref_depth < 0.01, torch::full_like(ref_depth, 1000.0), ref_depth);
}

// ------------------------------------------------------------------------------------------------ launch descriptor
gps_splat_step& RawGaussianModel::stepStruct(int W, int H) {
    RawGaussianParams& p = opt_gs_params;
    if (!p.buffer(0).defined()) p.reserve(1 << 19, numShBases(maxSH), device);
    const int64_t cap = p.capacity();
    if (step_cap_ != cap || step_w_ != W || step_h_ != H) {
        const int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
        const int64_t icap = isect_capacity > 0 ? isect_capacity : std::max<int64_t>(1 << 20, 16 * cap);
        const int64_t gcap = 2 * icap;
        const auto F = f32(device);
        const auto I = i32(device);
        B_.radii = torch::empty({cap}, I);
        B_.means2d = torch::empty({cap, 2}, F); B_.depths = torch::empty({cap}, F);
        B_.conics = torch::empty({cap, 3}, F); B_.colors = torch::empty({cap, 4}, F);
        B_.opacities = torch::empty({cap}, F); B_.records = torch::empty({cap, 12}, F);
        B_.tiles_per_gauss = torch::empty({cap}, I);
        B_.flatten_ids = torch::empty({icap}, I);
        B_.group_gs_ids = torch::empty({gcap}, I); B_.group_starts = torch::empty({gcap}, I);
        B_.tile_offsets = torch::empty({th * tw}, I);
        B_.counts = torch::zeros({4}, i64(device));
        const int64_t ws = gps_isect_workspace_bytes((int)cap, icap);
        B_.workspace = torch::zeros({ws}, u8(device));  // (the superblock binning's count tables are zero between launches)
        B_.render_colors = torch::empty({1, H, W, 4}, F); B_.weight_sum = torch::empty({1, H, W, 1}, F);
        B_.rgb = torch::empty({H, W, 3}, F); B_.depth = torch::empty({H, W, 1}, F);
        B_.loss = torch::zeros({1}, F);
        B_.v_render_colors = torch::empty({1, H, W, 4}, F); B_.v_render_alphas = torch::empty({1, H, W, 1}, F);
        B_.v_means2d = torch::empty({cap, 2}, F); B_.v_conics = torch::empty({cap, 3}, F);
        B_.v_colors = torch::empty({cap, 4}, F); B_.v_opacities = torch::empty({cap}, F);
        gps_splat_step& s = step_;
        s = gps_splat_step{};
        s.width = W; s.height = H;
        s.radii = iptr(B_.radii); s.means2d = fptr(B_.means2d); s.depths = fptr(B_.depths);
        s.conics = fptr(B_.conics); s.colors = fptr(B_.colors); s.opacities = fptr(B_.opacities);
        s.records = fptr(B_.records);
        s.isect_capacity = icap; s.group_capacity = gcap; s.workspace_bytes = ws;
        s.tiles_per_gauss = iptr(B_.tiles_per_gauss); s.flatten_ids = iptr(B_.flatten_ids);
        s.group_gs_ids = iptr(B_.group_gs_ids); s.group_starts = iptr(B_.group_starts);
        s.tile_offsets = iptr(B_.tile_offsets); s.counts = ptr<int64_t>(B_.counts); s.workspace = B_.workspace.data_ptr();
        s.render_colors = fptr(B_.render_colors); s.weight_sum = fptr(B_.weight_sum); s.rgb = fptr(B_.rgb);
        s.loss = fptr(B_.loss); s.v_render_colors = fptr(B_.v_render_colors); s.v_render_alphas = fptr(B_.v_render_alphas);
        s.v_means2d = fptr(B_.v_means2d); s.v_conics = fptr(B_.v_conics); s.v_colors = fptr(B_.v_colors);
        s.v_opacities = fptr(B_.v_opacities);
        if (strip_backward) {
            B_.v_rows = torch::empty({cap, 12}, F); B_.pix2 = torch::empty({(int64_t)H * W, 2}, F);
            B_.cls_ids = torch::empty({GPS_BWD_CLASSES, cap}, I); B_.cls_counts = torch::zeros({8}, I);
            s.v_rows = fptr(B_.v_rows); s.pix2 = fptr(B_.pix2);
            s.cls_ids = iptr(B_.cls_ids); s.cls_counts = iptr(B_.cls_counts); s.cls_stride = cap;
        }
        s.beta1 = kAdamBeta1; s.beta2 = kAdamBeta2; s.adam_eps = kAdamEps;
        step_cap_ = cap; step_w_ = W; step_h_ = H;
        // fresh intermediates: a forward the last trainStep() ran ahead lived in the buffers just replaced (checkBinningCapacity()
        // growing the tables, a parameter-capacity change, another image size) -- the next step preprocesses itself
        prefetched_ = PrefetchKey{};
    }
    gps_splat_step& s = step_;
    s.N = p.getGaussianNum(); s.K = p.shK(); s.sh_degree = degreesToUse; s.max_gs_radii = max_gs_radii;
    s.eps2d = eps2d; s.near_plane = near_plane; s.far_plane = far_plane; s.radius_clip = radius_clip;
    s.delta_depth = delta_depth;
    s.means = fptr(p.buffer(0)); s.log_scales = fptr(p.buffer(1)); s.quats = fptr(p.buffer(2));
    s.sh_dc = fptr(p.buffer(3)); s.sh_rest = fptr(p.buffer(4)); s.opac_logit = fptr(p.buffer(5));
    if (have_opt_ && adam_cap_ == cap) {
        float** g[6] = {&s.g_means, &s.g_log_scales, &s.g_quats, &s.g_sh_dc, &s.g_sh_rest, &s.g_opac_logit};
        float** m[6] = {&s.m_means, &s.m_log_scales, &s.m_quats, &s.m_sh_dc, &s.m_sh_rest, &s.m_opac_logit};
        float** v[6] = {&s.v_means, &s.v_log_scales, &s.v_quats, &s.v_sh_dc, &s.v_sh_rest, &s.v_opac_logit};
        for (int k = 0; k < 6; k++) {
            *g[k] = fptr(adam_g_[k]); *m[k] = fptr(adam_m_[k]); *v[k] = fptr(adam_v_[k]);
            s.lr[k] = lrs_[k];
        }
    }
    return s;
}

std::pair<int64_t, int64_t> RawGaussianModel::checkBinningCapacity() {
    if (!B_.counts.defined()) return {0, 0};
    auto c = B_.counts.cpu();  // blocking 32-byte read-back
    const int64_t* h = c.data_ptr<int64_t>();
    const int64_t ni = h[0], ng = h[1];
    const bool overflow = h[2] != 0;
    const int64_t icap = step_.isect_capacity, gcap = step_.group_capacity;
    // grow ahead of need (group capacity is 2 x the intersection capacity): the next stepStruct() re-creates the
    // capacity-sized buffers, and the sticky flag with them
    int64_t want = icap;
    const int64_t need = std::max(ni, (ng + 1) / 2);
    while (want < 2 * need) want *= 2;
    if (overflow) want = std::max(want, 2 * icap);
    if (want != icap) { isect_capacity = want; step_cap_ = -1; }
    (void)gcap;
    // The reference sizes these buffers exactly after a host read-back per forward and never fails here; with device-resident
    // counts an overflow is only seen now: Gaussians were dropped from the renders / backward passes since the last check.  Grow
    // and carry on (the next iteration runs with the larger buffers) -- loudly, once per occurrence.
    if (overflow) {
        binning_overflows++;
        TORCH_WARN("tile-intersection buffers overflowed (capacity ", icap, " intersections / ", gcap, " groups): Gaussians were "
                   "dropped from a render or a backward pass since the last check.  The capacity has been raised to ",
                   isect_capacity, " for the following iterations; set isect_capacity in the model configuration to start there.");
    }
    return {ni, ng};
}

void RawGaussianModel::bindCamera(gps_splat_step& st, const Camera& cam, const torch::Tensor& ref_depth_clamped,
                                  const torch::Tensor& base_color, const torch::Tensor& gt_rgb, bool consumes_prefetch) {
    TORCH_CHECK(cam.on_device(), "Camera::toGPU() must run before the camera is rendered (slam_pipeline.cpp:84)");
    if (prefetched_.camera != 0 && !consumes_prefetch)   // a forward run ahead for a train step that is not coming
        check(gps_splat_discard_prefetch(&st, current_stream()), "gps_splat_discard_prefetch");
    check_f32_dev(ref_depth_clamped, "ref_depth");
    check_f32_dev(base_color, "base_color");
    st.viewmat = cam.viewmat(); st.Kmat = cam.Kmat(); st.cam_pos = cam.cam_pos();
    st.ref_depth_clamped = fptr(ref_depth_clamped);
    st.base_color = fptr(base_color);
    st.gt_rgb = gt_rgb.defined() ? fptr(gt_rgb) : nullptr;
    // whatever runs on the step buffers invalidates a forward the last trainStep() ran ahead (trainStep re-arms it itself)
    st.next_viewmat = st.next_Kmat = st.next_cam_pos = nullptr;
    st.preprocessed = 0;
    prefetched_ = PrefetchKey{};
    keep_ = {ref_depth_clamped, base_color, gt_rgb};
}

// ------------------------------------------------------------------------------------------------ forward
namespace {

// grad-mode gesForward: inputs = the six parameter leaves; outputs = rgb, depth, alpha (views of model buffers).
// backward: d(rgb, depth, alpha) -> d(render_colors, weight_sum) with a few elementwise ops, then the fused chain
// raster bwd -> preprocess bwd through the C-ABI.
struct GesRenderFunction : public torch::autograd::Function<GesRenderFunction> {
    static tensor_list forward(AutogradContext* ctx, torch::Tensor means, torch::Tensor scales, torch::Tensor quats,
                               torch::Tensor dc, torch::Tensor rest, torch::Tensor opac, int64_t model_ptr,
                               int64_t cam_ptr, torch::Tensor ref_depth, torch::Tensor ref_clamped,
                               torch::Tensor base_color) {
        auto* model = reinterpret_cast<RawGaussianModel*>(model_ptr);
        const auto* cam = reinterpret_cast<const Camera*>(cam_ptr);
        gps_splat_step& st = model->stepStruct(cam->width, cam->height);
        model->bindCamera(st, *cam, ref_clamped, base_color, torch::Tensor());
        // this node's backward runs the operator-level group kernel on the render's 32-pixel group table: the sorted-key binning
        // writes it (the superblock binning of the strip path has no use for one)
        gps_splat_step with_groups = st;
        with_groups.v_rows = nullptr;
        check(gps_splat_render(&with_groups, current_stream()), "gps_splat_render");
        ctx->saved_data["launch_id"] = model->nextLaunchId();
        auto& B = model->buffers();
        check(gps_compose_l1(cam->width, cam->height, fptr(B.render_colors), fptr(B.weight_sum), fptr(base_color),
                             fptr(ref_depth), nullptr, fptr(B.rgb), fptr(B.depth), nullptr, nullptr, nullptr,
                             current_stream()), "gps_compose_l1");
        ctx->saved_data["model"] = model_ptr;
        ctx->saved_data["width"] = (int64_t)cam->width;
        ctx->saved_data["height"] = (int64_t)cam->height;
        // the camera's device pack travels with the node (the Camera object may be gone by the time backward runs)
        ctx->save_for_backward({ref_depth, ref_clamped, base_color, cam->pack_tensor()});
        (void)means; (void)scales; (void)quats; (void)dc; (void)rest; (void)opac;
        // fresh tensors: autograd owns its outputs, the model buffers are reused by the next launch
        return {B.rgb.clone(), B.depth.clone(), B.weight_sum.index({0}).clone()};
    }

    static tensor_list backward(AutogradContext* ctx, tensor_list g) {
        auto* model = reinterpret_cast<RawGaussianModel*>(ctx->saved_data["model"].toInt());
        auto saved = ctx->get_saved_variables();
        const torch::Tensor &ref_depth = saved[0], &ref_clamped = saved[1], &base_color = saved[2], &cam_pack = saved[3];
        const int H = (int)ctx->saved_data["height"].toInt(), W = (int)ctx->saved_data["width"].toInt();
        // The per-launch intermediates live in the model's buffers, not in this node (they are capacity sized: cloning them per
        // forward would cost more than the forward).  They are only valid until the next launch: refuse, loudly, to
        // differentiate a stale graph (two grad-mode forwards before one backward, an eval forward in between, ...).
        TORCH_CHECK(model->launchId() == ctx->saved_data["launch_id"].toInt(),
                    "gesForward: backward() of a render whose intermediates were overwritten by a later forward() / trainStep() "
                    "of the same model; call backward() before the next launch (one camera per backward, as "
                    "slam_pipeline.cpp:247-254 does)");
        auto& B = model->buffers();
        gps_splat_step& st = model->stepStruct(W, H);
        // compose: rgb = (raw_rgb + base)/(Ws + 1); depth = (raw_d + ref*b)/(Ws + b), b = [ref > 0]; alpha = Ws
        auto Ws = B.weight_sum.index({0});                                   // [H,W,1]
        auto v_rc = torch::zeros({1, H, W, 4}, Ws.options());
        auto v_ra = torch::zeros({H, W, 1}, Ws.options());
        if (g[0].defined()) {
            auto inv = 1.0 / (Ws + 1.0);
            v_rc.index_put_({0, Slice(), Slice(), Slice(0, 3)}, g[0] * inv);
            v_ra -= (g[0] * B.rgb).sum(-1, true) * inv;
        }
        if (g[1].defined()) {
            auto b = (ref_depth > 0).to(torch::kFloat32);
            auto inv = 1.0 / (Ws + b);
            inv = torch::where(torch::isfinite(inv), inv, torch::zeros_like(inv));  // Ws = 0, no reference depth
            v_rc.index_put_({0, Slice(), Slice(), Slice(3, 4)}, g[1] * inv);
            v_ra -= g[1] * B.depth * inv;
        }
        if (g[2].defined()) v_ra += g[2];
        v_ra = v_ra.unsqueeze(0).contiguous();
        st.viewmat = fptr(cam_pack); st.Kmat = fptr(cam_pack) + 16; st.cam_pos = fptr(cam_pack) + 25;
        st.ref_depth_clamped = fptr(ref_clamped);
        check(gps_raster_ges_bwd_gs(st.N, st.means2d, st.conics, st.colors, st.opacities, st.radii, st.ref_depth_clamped,
                                    W, H, st.group_gs_ids, st.group_starts, st.counts, st.delta_depth, fptr(v_rc),
                                    fptr(v_ra), st.v_means2d, st.v_conics, st.v_colors, st.v_opacities, 0,
                                    current_stream()), "gps_raster_ges_bwd_gs");
        RawGaussianParams& p = model->getGaussianParms();
        tensor_list grads(11);
        for (int k = 0; k < 6; k++) grads[k] = torch::empty_like(p.buffer(k).slice(0, 0, st.N));
        check(gps_gauss_preprocess_bwd(st.N, st.K, st.sh_degree, st.means, st.log_scales, st.quats, st.opac_logit,
                                       st.sh_dc, st.sh_rest, st.viewmat, st.Kmat, st.cam_pos, W, H, st.eps2d, st.radii,
                                       st.conics, st.v_means2d, st.v_conics, st.v_colors, st.v_opacities,
                                       fptr(grads[0]), fptr(grads[1]), fptr(grads[2]), fptr(grads[5]), fptr(grads[3]),
                                       fptr(grads[4]), current_stream()), "gps_gauss_preprocess_bwd");
        return grads;
    }
};

}  // namespace

TensorDict RawGaussianModel::forward(const Camera& cam, const torch::Tensor& ref_depth, const torch::Tensor& base_color) {
    if (render_method == "raw") return rawForward(cam);  // raw_gs_model.h:35-47
    TORCH_CHECK(render_method == "ges", "UNSUPPORTED RENDER METHOD: ", render_method);
    return gesForward(cam, ref_depth, base_color);
}

// raw_gs_model.cpp:43-185: projection -> SH colours -> depth-keyed binning -> front-to-back compositing, composed from
// the operator-level autograd Functions exactly as the reference composes them (the `raw` method is not on the SLAM loop's
// path, so it does not get the fused step struct; every stage is still one C-ABI call).
TensorDict RawGaussianModel::rawForward(const Camera& cam) {
    const int N = getGaussianNum();
    const bool grad = torch::GradMode::is_enabled() && !leaf_.empty();
    auto P = [&](int k) { return grad ? leaf_[k] : opt_gs_params.view(k); };
    auto c2w = cam.c2w.to(torch::kCPU, torch::kFloat32).contiguous();  // the dataset pose, not c2w_slam (:56)
    auto viewMat = poseInv(c2w).to(device);
    auto cam_T = c2w.index({Slice(0, 3), Slice(3, 4)}).to(device);
    auto Ks = cam.K.to(device, torch::kFloat32);
    auto world_means = P(0).contiguous();
    auto world_scales = torch::exp(P(1)).contiguous();
    auto proj = FullyFusedProjection::apply(world_means, c10::nullopt, P(2), world_scales, viewMat.unsqueeze(0), Ks.unsqueeze(0),
                                            cam.width, cam.height, eps2d, near_plane, far_plane, radius_clip, false,
                                            std::string("pinhole"));
    auto radiis = proj[0], means2d = proj[1], depths = proj[2], conics = proj[3];
    auto shs = torch::cat({P(3).view({N, 1, 3}), P(4)}, 1);
    auto viewDirs = world_means - cam_T.transpose(0, 1);
    auto colors = SphericalHarmonicsNew::apply(degreesToUse, viewDirs.unsqueeze(0), shs.unsqueeze(0), radiis > 0);
    colors = torch::clamp_min(colors + 0.5f, 0.0f);
    const int tile_width = (int)std::ceil(float(cam.width) / float(tile_size));
    const int tile_height = (int)std::ceil(float(cam.height) / float(tile_size));
    auto isec = isectTiles(means2d, radiis, depths, tile_size, tile_width, tile_height);
    auto isect_offsets = isectOffsetEncode(isec[1], 1, tile_width, tile_height);
    colors = torch::cat({colors, depths.unsqueeze(-1)}, 2);
    c10::optional<torch::Tensor> bg;
    if (backgrounds.defined()) bg = backgrounds;
    auto rast = RasterizeToPixels::apply(means2d, conics, colors, torch::sigmoid(P(5)), bg, c10::nullopt, cam.width, cam.height,
                                         tile_size, isect_offsets, isec[2], abs_grad);
    auto render_colors = rast[0], render_alphas = rast[1];
    const int64_t last_dim = render_colors.size(-1);
    auto rgb = render_colors.slice(-1, 0, last_dim - 1);
    auto raw_depth = render_colors.slice(-1, last_dim - 1);
    auto expected_depth = raw_depth / render_alphas.clamp(1e-10);
    TensorDict res;
    res["rgb"] = rgb[0];
    res["depth"] = expected_depth[0];
    res["alpha"] = render_alphas[0];
    res["radiis"] = radiis[0];
    res["means2d"] = means2d;
    return res;
}

TensorDict RawGaussianModel::gesForward(const Camera& cam, const torch::Tensor& ref_depth, const torch::Tensor& base_color) {
    check_f32_dev(ref_depth, "ref_depth");
    check_f32_dev(base_color, "base_color");
    auto ref_clamped = clampRefDepth(ref_depth);
    const int N = getGaussianNum();
    TensorDict res;
    if (torch::GradMode::is_enabled() && !leaf_.empty()) {
        auto out = GesRenderFunction::apply(leaf_[0], leaf_[1], leaf_[2], leaf_[3], leaf_[4], leaf_[5],
                                            (int64_t)reinterpret_cast<intptr_t>(this),
                                            (int64_t)reinterpret_cast<intptr_t>(&cam), ref_depth, ref_clamped, base_color);
        res["rgb"] = out[0]; res["depth"] = out[1]; res["alpha"] = out[2];
    } else {
        gps_splat_step& st = stepStruct(cam.width, cam.height);
        bindCamera(st, cam, ref_clamped, base_color, torch::Tensor());
        check(gps_splat_render(&st, current_stream()), "gps_splat_render");
        nextLaunchId();
        check(gps_compose_l1(cam.width, cam.height, fptr(B_.render_colors), fptr(B_.weight_sum), fptr(base_color),
                             fptr(ref_depth), nullptr, fptr(B_.rgb), fptr(B_.depth), nullptr, nullptr, nullptr,
                             current_stream()), "gps_compose_l1");
        res["rgb"] = B_.rgb; res["depth"] = B_.depth; res["alpha"] = B_.weight_sum.index({0});
    }
    res["radiis"] = B_.radii.slice(0, 0, N);
    res["means2d"] = B_.means2d.slice(0, 0, N);
    return res;
}

TensorDict RawGaussianModel::computeLoss(TensorDict& render_res, const Camera& cam, const Config& w, const torch::Tensor& mask) {
    // raw_gs_model.cpp:369-417: L1 (+ SSIM with the fused kernel, + masked depth L1) -> {"total", "rgb"[, "depth"]};
    // "loss" / "l1_loss" are kept as aliases for callers of the first version of this host layer
    auto gt_rgb = cam.image.to(device);
    const auto& rendered_rgb = render_res.at("rgb");
    auto l1 = mask.defined() ? torch::mean(torch::abs(gt_rgb.masked_select(mask) - rendered_rgb.masked_select(mask)))
                             : torch::mean(torch::abs(gt_rgb - rendered_rgb));  // tensor_math.cpp:41-44
    torch::Tensor rgb_loss;
    const float ssimWeight = (float)w.get("ssim_weight", 0.0);
    if (ssimWeight > 0) {
        const float C1 = 0.01 * 0.01, C2 = 0.03 * 0.03;
        auto ssimLoss = 1.0f - FusedSSIMMap::apply((double)C1, (double)C2, rendered_rgb.permute({2, 0, 1}).unsqueeze(0),
                                                   gt_rgb.permute({2, 0, 1}).unsqueeze(0), std::string("valid"), true).mean();
        rgb_loss = (1.0f - ssimWeight) * l1 + ssimWeight * ssimLoss;
    } else {
        rgb_loss = l1 * w.get("l1_weight", 1.0);
    }
    TensorDict out;
    out["total"] = rgb_loss;
    out["rgb"] = rgb_loss;
    const float depth_weight = (float)w.get("depth_weight", 0.0);
    if (depth_weight > 0 && cam.has_depth) {
        auto gt_depth = cam.depth.to(device);
        const auto& rendered_depth = render_res.at("depth");
        auto valid = (gt_depth > 0) & (rendered_depth > 0);
        auto depthLoss = torch::mean(torch::abs(gt_depth.masked_select(valid) - rendered_depth.masked_select(valid)));
        out["depth"] = depthLoss;
        out["total"] = out["total"] + depth_weight * depthLoss;
    }
    out["l1_loss"] = l1;
    out["loss"] = out["total"];
    return out;
}

// ------------------------------------------------------------------------------------------------ optimisation
void RawGaussianModel::setParamsRequireGrad() {
    leaf_.clear();
    for (int k = 0; k < RawGaussianParams::NUM; k++) {
        auto v = opt_gs_params.buffer(k).slice(0, 0, getGaussianNum()).detach();
        v.set_requires_grad(true);
        leaf_.push_back(v);
    }
}

void RawGaussianModel::initOptimizers(int max_iterations, float scene_scale) {
    (void)max_iterations;
    RawGaussianParams& p = opt_gs_params;
    const int64_t cap = p.capacity();
    if (!have_opt_ || adam_cap_ != cap) {
        adam_m_.clear(); adam_v_.clear(); adam_g_.clear();
        for (int k = 0; k < 6; k++) {
            adam_m_.push_back(torch::zeros_like(p.buffer(k)));
            adam_v_.push_back(torch::zeros_like(p.buffer(k)));
            adam_g_.push_back(torch::zeros_like(p.buffer(k)));
        }
        adam_cap_ = cap;
    }
    // (fresh state for the live rows: nothing to do -- step 1 of every route (gps_splat_train_step, gps_adam_step) takes the
    // moments as zero without reading them and writes them for every live row; rounds 1-4 zeroed 2 x 59 floats per Gaussian here,
    // 50 us per keyframe at 245 k)
    pending_prunes_.clear();  // the state those would have compacted has just been replaced
    // the reference holds its rates in float members (config[...].as<float>(), raw_gs_model.cpp:26-32), multiplies meansLr by the
    // float scene_scale and widens the float result for AdamOptions (:666-671): the doubles the optimiser divides by 1 - beta1^t
    // are those floats' values
    lrs_[0] = (double)((float)means_lr * scene_scale); lrs_[1] = (double)(float)scales_lr; lrs_[2] = (double)(float)quats_lr;
    lrs_[3] = (double)(float)featuresDc_lr; lrs_[4] = (double)(float)featuresRest_lr; lrs_[5] = (double)(float)opacities_lr;
    adam_step_ = 0;
    have_opt_ = true;
    setParamsRequireGrad();
}

void RawGaussianModel::reserveWorkspace(int width, int height) {
    stepStruct(width, height);
    const bool had = have_opt_;
    const int step = adam_step_;
    initOptimizers(-1, 1.0f);  // allocates m / v / g at capacity
    have_opt_ = had;           // ... but leaves the optimiser logically un-initialised if it was
    adam_step_ = step;
}

void RawGaussianModel::optimizersZeroGrad() {
    for (auto& t : leaf_) t.mutable_grad() = torch::Tensor();
}

void RawGaussianModel::optimizersStep() {
    // the autograd route: gradients sit in the leaves' .grad(); one fused multi-tensor Adam launch
    TORCH_CHECK(have_opt_, "initOptimizers() first");
    applyPendingPrunes();
    gps_adam_segment seg[6];
    const int64_t N = getGaussianNum();
    std::vector<torch::Tensor> hold;
    for (int k = 0; k < 6; k++) {
        TORCH_CHECK(leaf_[k].grad().defined(), "optimizersStep: parameter ", k, " has no gradient");
        auto g = leaf_[k].grad().contiguous();
        hold.push_back(g);
        seg[k].param = fptr(opt_gs_params.buffer(k)); seg[k].grad = fptr(g);
        seg[k].exp_avg = fptr(adam_m_[k]); seg[k].exp_avg_sq = fptr(adam_v_[k]);
        seg[k].numel = N * (opt_gs_params.buffer(k).numel() / opt_gs_params.capacity());
        seg[k].lr = lrs_[k];
    }
    adam_step_ += 1;
    check(gps_adam_step(seg, 6, kAdamBeta1, kAdamBeta2, kAdamEps, adam_step_, current_stream()), "gps_adam_step");
}

void RawGaussianModel::trainStep(const Camera& cam, const torch::Tensor& ref_depth, const torch::Tensor& base_color,
                                 const torch::Tensor& ref_depth_clamped, const Camera* next_cam) {
    TORCH_CHECK(have_opt_ && adam_cap_ == opt_gs_params.capacity(), "initOptimizers() first");
    TORCH_CHECK(cam.image.defined() && cam.image.is_cuda(), "camera image must be on the device");
    applyPendingPrunes();
    auto clamped = ref_depth_clamped.defined() ? ref_depth_clamped : clampRefDepth(ref_depth);
    gps_splat_step& st = stepStruct(cam.width, cam.height);
    // did the previous trainStep() run this camera's preprocessing forward in its backward kernel's tail (next_cam)?
    const PrefetchKey mine{cam.pack_serial(), (int64_t)st.N, cam.width, cam.height, opt_gs_params.version()};
    const bool skip = prefetched_.camera != 0 && prefetched_ == mine;
    bindCamera(st, cam, clamped, base_color, cam.image, skip);   // (clears prefetched_)
    st.fuse_sh_rest_adam = fuse_sh_rest_adam ? 2 : 0;  // all six tensors stepped inside the backward kernel
    st.preprocessed = skip ? 1 : 0;
    const bool ahead = next_cam && next_cam->on_device() && next_cam->width == cam.width && next_cam->height == cam.height &&
                       gps_splat_can_prefetch(&st);
    if (ahead) { st.next_viewmat = next_cam->viewmat(); st.next_Kmat = next_cam->Kmat(); st.next_cam_pos = next_cam->cam_pos(); }
    adam_step_ += 1;
    check(gps_splat_train_step(&st, adam_step_, current_stream()), "gps_splat_train_step");   // throws on error: nothing armed then
    if (ahead) prefetched_ = PrefetchKey{next_cam->pack_serial(), (int64_t)st.N, cam.width, cam.height, opt_gs_params.version()};
    nextLaunchId();
}

std::vector<torch::Tensor> RawGaussianModel::grads() {
    std::vector<torch::Tensor> out;
    for (int k = 0; k < 6; k++) out.push_back(adam_g_[k].slice(0, 0, getGaussianNum()));
    return out;
}

void RawGaussianModel::prunePoints(const torch::Tensor& deleteMask) { pruneKeep(~deleteMask); }

int64_t RawGaussianModel::pruneKeep(const torch::Tensor& keepMask) {
    RawGaussianParams& p = opt_gs_params;
    const int64_t N = p.getGaussianNum();
    const int64_t m = p.removeKeep(keepMask);
    // removeFromOptimizer (raw_gs_model.cpp:640-644): the Adam state follows the parameters -- LAZILY.  The pipeline builds
    // fresh optimizers before the next step (initOptimizers at every localOptimize), which makes compacting 2 x 59 floats per
    // Gaussian here dead work; the compaction is recorded and carried out only if a step / state read comes first.
    if (m != N && have_opt_ && adam_cap_ == p.capacity()) pending_prunes_.push_back({p.keep_index(), N});
    if (!leaf_.empty()) setParamsRequireGrad();
    return m;
}

void RawGaussianModel::applyPendingPrunes() {
    if (pending_prunes_.empty()) return;
    RawGaussianParams& p = opt_gs_params;
    for (auto& pr : pending_prunes_) {
        const int64_t m = pr.keep.size(0);
        for (int k = 0; k < 6; k++) {
            for (auto* vec : {&adam_m_, &adam_v_}) {
                auto& t = (*vec)[k];
                auto tmp = p.alt_[k].slice(0, 0, m);  // the alternate parameter buffer is free scratch between remove() calls
                torch::index_select_out(tmp, t.slice(0, 0, pr.n_before), 0, pr.keep);
                t.slice(0, 0, m).copy_(tmp);
            }
        }
    }
    pending_prunes_.clear();
}

// ------------------------------------------------------------------------------------------------ addGaussians
int SLAMGaussianModel::addGaussians(const Camera& cam, const TensorDict& frame_maps, const torch::Tensor& sample_mask,
                                    float new_gs_sample_ratio, int frame_num, c10::optional<at::Generator> gen) {
    (void)frame_num;
    const int64_t H = cam.image.size(0), W = cam.image.size(1), P = H * W;
    const auto dev = cam.image.device();
    // The reference materialises three masked_select results (each a nonzero + gather with its own host round trip) and then
    // keeps a random tenth of the rows.  Here: ONE ordered compaction of the mask into pixel ids (same selection order),
    // one read of their number, and a gather of only the sampled rows.
    TORCH_CHECK(sample_mask.scalar_type() == torch::kBool && sample_mask.numel() == P, "sample_mask: bool [H,W,1]");
    auto mask = sample_mask.contiguous();
    auto image = cam.image.contiguous();
    auto vertex = frame_maps.at("vertex_map").contiguous(), normal = frame_maps.at("normal_map").contiguous();
    gpsh::check_f32_dev(image, "cam.image"); gpsh::check_f32_dev(vertex, "vertex_map"); gpsh::check_f32_dev(normal, "normal_map");
    auto ids = torch::empty({P}, gpsh::i32(dev));
    auto count = torch::empty({1}, gpsh::i32(dev));
    if (!host_count_.defined()) host_count_ = torch::zeros({16}, torch::TensorOptions().dtype(torch::kInt32).pinned_memory(true));
    auto stream = c10::hip::getCurrentHIPStream();
    auto cws = torch::empty({gps_compact_mask_workspace_bytes((int)P)}, gpsh::u8(dev));
    gpsh::check(gps_compact_mask((int)P, reinterpret_cast<const uint8_t*>(mask.data_ptr<bool>()), gpsh::iptr(ids), gpsh::iptr(count),
                                 host_count_.data_ptr<int32_t>(), cws.data_ptr(), cws.numel(), (gps_stream)stream.stream()),
                "gps_compact_mask");
    stream.synchronize();
    const int64_t n = host_count_.data_ptr<int32_t>()[0];
    const int64_t num_select = (int64_t)(n * new_gs_sample_ratio);
    if (num_select <= 0) return 0;
    // uniformly random subset of num_select of the n masked pixels (the reference: torch::randperm(n)[:num_select]), drawn on
    // the host, n is known here.
    // The subset is the reference's; the ORDER is pixel order, not permutation order: Gaussians with neighbouring ids then
    // splat onto neighbouring pixels, which keeps the gradient-image gathers of the Gaussian-parallel backward inside each
    // XCD's L2 (with random ids every XCD sweeps the whole 7 MB image: rocprofv3 FETCH_SIZE 190 MB per launch).
    // Drawn with Floyd's algorithm (k draws into a bitmap of n bits, then one sweep of the bitmap: the ascending order falls out
    // of it) instead of randperm(n) + sort: the map stream sits idle while the host draws, and a permutation of ALL n masked
    // pixels plus an O(k log k) sort cost ~200 us per keyframe against ~30 us for this -- same distribution (every k-subset
    // equally likely; the reference's own generator is std::random_device-seeded, dataset_reader.h:38-39).
    // pinned staging buffer, allocated once (hipHostMalloc costs milliseconds and synchronises the device); the stream
    // synchronise above guarantees the previous call's copy out of it has completed
    if (!host_subset_.defined() || host_subset_.numel() < num_select)
        host_subset_ = torch::empty({std::max<int64_t>(P, num_select)}, torch::TensorOptions().dtype(torch::kInt32).pinned_memory(true));
    auto perm32 = host_subset_.slice(0, 0, num_select);
    {
        at::Generator g = gen.has_value() ? *gen : at::detail::getDefaultCPUGenerator();
        auto* impl = at::check_generator<at::CPUGeneratorImpl>(g);
        std::lock_guard<std::mutex> lock(impl->mutex_);
        std::vector<uint64_t> bits((size_t)(n + 63) / 64, 0);
        auto test_set = [&](int64_t v) { const bool was = (bits[v >> 6] >> (v & 63)) & 1; bits[v >> 6] |= 1ull << (v & 63); return was; };
        for (int64_t j = n - num_select; j < n; j++) {
            const int64_t t = (int64_t)(impl->random64() % (uint64_t)(j + 1));   // uniform on [0, j] (bias < 2^-40 for n < 2^24)
            if (test_set(t)) test_set(j);
        }
        int32_t* out = perm32.data_ptr<int32_t>();
        int64_t w = 0;
        for (size_t q = 0; q < bits.size(); q++)
            for (uint64_t m = bits[q]; m; m &= m - 1) out[w++] = (int32_t)(q * 64 + __builtin_ctzll(m));
        TORCH_CHECK(w == num_select, "addGaussians: subset size");
    }
    auto subset = torch::empty({num_select}, gpsh::i32(dev));
    TORCH_CHECK(hipMemcpyAsync(subset.data_ptr(), perm32.data_ptr(), (size_t)num_select * 4, hipMemcpyHostToDevice, stream.stream()) ==
                    hipSuccess, "hipMemcpyAsync(subset)");
    auto verts = torch::empty({num_select, 3}, gpsh::f32(dev)), cols = torch::empty({num_select, 3}, gpsh::f32(dev)),
         norms = torch::empty({num_select, 3}, gpsh::f32(dev));
    gpsh::check(gps_gather_pixels((int)num_select, gpsh::iptr(ids), gpsh::iptr(subset), gpsh::fptr(vertex), gpsh::fptr(image),
                                  gpsh::fptr(normal), gpsh::fptr(verts), gpsh::fptr(cols), gpsh::fptr(norms),
                                  (gps_stream)stream.stream()), "gps_gather_pixels");
    if (!opt_gs_params.buffer(0).defined()) opt_gs_params.reserve(1 << 19, numShBases(maxSH), verts.device());
    opt_gs_params.appendInit(verts, cols, norms, maxSH, defaultOpacities, maxInitScale, minInitScale);
    if (!leaf_.empty()) setParamsRequireGrad();
    return (int)num_select;
}

// One optimise iteration of SLAMPipeline::localOptimize (slam/slam_pipeline.cpp:247-254):
//   model.forward -> computeLoss -> loss.backward() -> optimizersStep() -> optimizersZeroGrad()
// as ONE C-ABI call that enqueues the whole kernel chain on a stream (and the render-only half for the
// NoGradGuard call sites).  It only sequences the entry points the parity tests pin individually; it exists so
// that a host in any language pays one FFI crossing per iteration instead of ~10, and so that the chain can be
// captured into a hipGraph by the caller (nothing here allocates or synchronises).
#include "common.hpp"
#include "splat_adam.hpp"
#include "splat_bin.hpp"

extern "C" {

// projection + binning + forward rasterizer; `compose` / `zero`: the train step's compose + L1 and gradient zero-fill riding
// along in those kernels (nullptr: the plain render)
static int render_chain(const gps_splat_step* a, const gps::FwdCompose* compose, const gps::ZeroGrads* zero, gps_stream stream) {
    GPS_REQUIRE(a != nullptr);
    const int tw = gps_div_up(a->width, 16), th = gps_div_up(a->height, 16);
    int r;
    // the preprocessing kernel also writes the first pass of the tile binning (tiles / groups per Gaussian + block sums)
    gps::BinCountOut cnt;
    r = gps::isect_count_targets(a->N, a->isect_capacity, a->tiles_per_gauss, 16, tw, th, a->workspace, a->workspace_bytes, &cnt);
    if (r != GPS_OK) return r;
    r = gps::preprocess_fwd_launch(a->N, a->K, a->sh_degree, a->means, a->log_scales, a->quats, a->opac_logit, a->sh_dc,
                                   a->sh_rest, a->viewmat, a->Kmat, a->cam_pos, a->width, a->height, a->eps2d,
                                   a->near_plane, a->far_plane, a->radius_clip, a->max_gs_radii, a->radii, a->means2d,
                                   a->depths, a->conics, a->colors, a->opacities, a->records, &cnt, zero, stream);
    if (r != GPS_OK) return r;
    r = gps::isect_tiles_no_depth_counted(a->N, a->means2d, a->radii, 16, tw, th, a->isect_capacity, a->group_capacity,
                                          a->tiles_per_gauss, a->flatten_ids, a->group_gs_ids, a->group_starts,
                                          a->tile_offsets, a->counts, a->workspace, a->workspace_bytes, stream);
    if (r != GPS_OK) return r;
    if (a->records)
        r = gps::raster_ges_fwd_rec_launch(a->N, a->records, a->ref_depth_clamped, a->width, a->height, a->tile_offsets,
                                           a->flatten_ids, a->counts, a->delta_depth, a->render_colors, a->weight_sum,
                                           compose, stream);
    else
        r = gps_raster_ges_fwd(a->N, a->means2d, a->conics, a->colors, a->opacities, a->ref_depth_clamped, a->width,
                               a->height, 16, a->tile_offsets, a->flatten_ids, a->counts, a->delta_depth,
                               a->render_colors, a->weight_sum, nullptr, stream);
    return r;
}

int gps_splat_render(const gps_splat_step* a, gps_stream stream) { return render_chain(a, nullptr, nullptr, stream); }

int gps_splat_train_step(const gps_splat_step* a, int adam_step, gps_stream stream) {
    GPS_REQUIRE(a != nullptr && adam_step >= 1);
    GPS_REQUIRE(a->gt_rgb && a->loss && a->v_render_colors && a->v_render_alphas);
    // Launch sites of one iteration (8): preprocess (+ binning count pass + zero-fill of the rasterizer gradients), expand
    // (+ block prefix and totals), count table, row scan, scatter, forward rasterizer (+ compose + L1 + image gradients in its epilogue), backward
    // rasterizer, preprocess backward (+ Adam).
    GPS_REQUIRE(a->base_color != nullptr);
    const bool fused_fwd = a->records != nullptr;  // the record rasterizer carries the compose epilogue
    gps::FwdCompose fc = {a->base_color, a->gt_rgb, a->rgb, a->loss, a->v_render_colors, a->v_render_alphas,
                          1.0f / (3.0f * (float)(a->width * a->height))};
    gps::ZeroGrads zg = {a->v_means2d, a->v_conics, a->v_colors, a->v_opacities};
    int r = render_chain(a, fused_fwd ? &fc : nullptr, &zg, stream);
    if (r != GPS_OK) return r;
    if (!fused_fwd) {
        r = gps_compose_l1(a->width, a->height, a->render_colors, a->weight_sum, a->base_color, nullptr, a->gt_rgb, a->rgb,
                           nullptr, a->loss, a->v_render_colors, a->v_render_alphas, stream);
        if (r != GPS_OK) return r;
    }
    r = gps::raster_ges_bwd_gs_launch(a->N, a->means2d, a->conics, a->colors, a->opacities, a->radii, a->ref_depth_clamped,
                                      a->width, a->height, a->group_gs_ids, a->group_starts, a->counts, a->delta_depth,
                                      a->v_render_colors, a->v_render_alphas, a->v_means2d, a->v_conics, a->v_colors,
                                      a->v_opacities, a->N > 0 ? 2 : 0, stream);
    if (r != GPS_OK) return r;
    const int mode = a->K > 1 ? a->fuse_sh_rest_adam : 0;
    const bool fuse = mode >= 1, all = mode >= 2;
    gps_adam_segment seg[6] = {
        {a->means, a->g_means, a->m_means, a->v_means, (int64_t)a->N * 3, a->lr[0]},
        {a->log_scales, a->g_log_scales, a->m_log_scales, a->v_log_scales, (int64_t)a->N * 3, a->lr[1]},
        {a->quats, a->g_quats, a->m_quats, a->v_quats, (int64_t)a->N * 4, a->lr[2]},
        {a->sh_dc, a->g_sh_dc, a->m_sh_dc, a->v_sh_dc, (int64_t)a->N * 3, a->lr[3]},
        {a->opac_logit, a->g_opac_logit, a->m_opac_logit, a->v_opac_logit, (int64_t)a->N, a->lr[5]},
        {a->sh_rest, a->g_sh_rest, a->m_sh_rest, a->v_sh_rest, (int64_t)a->N * (a->K - 1) * 3, a->lr[4]},
    };
    float sstep[5];
    for (int k = 0; k < 5; k++) sstep[k] = gps::adam_scalars(seg[k].lr, a->beta1, a->beta2, a->adam_eps, adam_step).step_size;
    r = gps::preprocess_bwd_launch(a->N, a->K, a->sh_degree, a->means, a->log_scales, a->quats, a->opac_logit, a->sh_dc,
                                   a->sh_rest, a->viewmat, a->Kmat, a->cam_pos, a->width, a->height, a->eps2d, a->radii,
                                   a->conics, a->v_means2d, a->v_conics, a->v_colors, a->v_opacities,
                                   all ? nullptr : a->g_means, a->g_log_scales, a->g_quats, a->g_opac_logit, a->g_sh_dc,
                                   fuse ? nullptr : a->g_sh_rest, fuse ? a->sh_rest : nullptr,
                                   fuse ? a->m_sh_rest : nullptr, fuse ? a->v_sh_rest : nullptr,
                                   gps::adam_scalars(a->lr[4], a->beta1, a->beta2, a->adam_eps, adam_step),
                                   all ? seg : nullptr, all ? sstep : nullptr, stream);
    if (r != GPS_OK || all) return r;
    return gps_adam_step(seg, fuse ? 5 : 6, a->beta1, a->beta2, a->adam_eps, adam_step, stream);
}

}  // extern "C"

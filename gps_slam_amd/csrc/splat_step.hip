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

// the strip backward's buffers are all there and the tile ids fit the superblock binning's LDS histograms
static bool strips_on(const gps_splat_step* a) {
    const int tw = gps_div_up(a->width, 16), th = gps_div_up(a->height, 16);
    return a->v_rows && a->pix2 && a->cls_ids && a->cls_counts && a->cls_stride >= a->N && a->records &&
           (a->N <= 0 || gps::sb_supported(a->N, tw, th));   // (tile count, packed box fields, the scatter's LDS on this device)
}

// projection + binning + forward rasterizer; `compose` / `zero`: the train step's compose + L1 and gradient zero-fill riding
// along in those kernels (nullptr: the plain render; zero != nullptr also marks a train step: the binning then writes the
// backward's class lists)
static int render_chain(const gps_splat_step* a, const gps::FwdCompose* compose, const gps::ZeroGrads* zero, gps_stream stream,
                        bool preprocessed = false) {
    GPS_REQUIRE(a != nullptr);
    const int tw = gps_div_up(a->width, 16), th = gps_div_up(a->height, 16);
    int r;
    // the preprocessing kernel also writes the first pass of the tile binning: tiles (/ groups) per Gaussian + block sums, and
    // with the strip backward's buffers present the superblock binning's histogram pass
    const bool sb = strips_on(a) && a->N > 0;
    gps::BinCountOut cnt;
    r = gps::isect_count_targets(a->N, a->isect_capacity, a->tiles_per_gauss, 16, tw, th, a->workspace, a->workspace_bytes, sb, &cnt);
    if (r != GPS_OK) return r;
    // (preprocessed: the previous train step's backward kernel has already run this forward -- gps_splat_step::next_viewmat)
    GPS_REQUIRE(!preprocessed || sb);
    r = preprocessed ? GPS_OK
                     : gps::preprocess_fwd_launch(a->N, a->K, a->sh_degree, a->means, a->log_scales, a->quats, a->opac_logit, a->sh_dc,
                                                  a->sh_rest, a->viewmat, a->Kmat, a->cam_pos, a->width, a->height, a->eps2d,
                                                  a->near_plane, a->far_plane, a->radius_clip, a->max_gs_radii, a->radii, a->means2d,
                                                  a->depths, a->conics, a->colors, a->opacities, a->records, &cnt, zero, stream);
    if (r != GPS_OK) {
        if (sb) (void)gps::sb_tables_clear(cnt.sb, stream);   // (the kernel may have added counts that no scan will clear)
        return r;
    }
    if (sb)   // scan + scatter (+ the backward's class lists when this is a train step)
        r = gps::isect_tiles_superblock(a->N, a->means2d, a->radii, cnt, a->isect_capacity, a->tiles_per_gauss, a->flatten_ids,
                                        a->tile_offsets, a->counts, zero ? a->cls_ids : nullptr, zero ? a->cls_counts : nullptr,
                                        a->cls_stride, stream);
    else
        r = gps::isect_tiles_no_depth_counted(a->N, a->means2d, a->radii, 16, tw, th, a->isect_capacity, a->group_capacity,
                                              a->tiles_per_gauss, a->flatten_ids, a->group_gs_ids, a->group_starts,
                                              a->tile_offsets, a->counts, a->workspace, a->workspace_bytes, stream);
    if (r != GPS_OK) return r;
    if (a->records)
        r = gps::raster_ges_fwd_rec_launch(a->N, a->records, a->ref_depth_clamped, a->width, a->height, a->tile_offsets,
                                           a->flatten_ids, a->counts, a->delta_depth, a->render_colors, a->weight_sum,
                                           compose, stream,
                                           // longest lists first when the map kernels run alone (iteration 257 -> 248 us); beside a
                                           // frame chain row-major order is the better one (overlap 1,285 vs 1,276 frames/s, 6 + 6 runs)
                                           sb && !gps::map_runs_beside_frame_chain() ? cnt.sb.tile_order : nullptr);
    else
        r = gps_raster_ges_fwd(a->N, a->means2d, a->conics, a->colors, a->opacities, a->ref_depth_clamped, a->width,
                               a->height, 16, a->tile_offsets, a->flatten_ids, a->counts, a->delta_depth,
                               a->render_colors, a->weight_sum, nullptr, stream);
    return r;
}

int gps_splat_render(const gps_splat_step* a, gps_stream stream) { return render_chain(a, nullptr, nullptr, stream); }

int gps_splat_discard_prefetch(const gps_splat_step* a, gps_stream stream) {
    GPS_REQUIRE(a != nullptr && a->workspace != nullptr);
    // the prefetched forward has added its counts to the superblock binning's persistent tables, which only the scan of the step
    // that consumes it would clear: back to "zero between launches"
    // (whatever N and image size the struct holds by now: the model may have been pruned or grown since)
    gps::SbTables t;
    const int r = gps::isect_workspace_tables(a->workspace, a->workspace_bytes, &t);
    if (r != GPS_OK) return r;
    return gps::sb_tables_clear(t, stream);
}

int gps_splat_can_prefetch(const gps_splat_step* a) {
    if (!a || a->N <= 0 || a->K <= 1 || a->fuse_sh_rest_adam < 2 || !strips_on(a)) return 0;
    // the backward kernel's workgroup (splat_fused.hip: GPS_FUSED_ADAM_THREADS, halved until two row tiles fit 64 KB) must divide
    // the binning's 256-Gaussian blocks and its gradient tile must hold the binning's histogram
    int threads = 128;
    while (threads > 64 && (size_t)threads * (a->K - 1) * 3 * 4 * 2 > 65536) threads >>= 1;
    return (size_t)threads * (a->K - 1) * 3 * 4 >= (size_t)(gps::SB_MAX_TILES + gps::BWD_KEYS) * 4 ? 1 : 0;
}

int gps_splat_train_step(const gps_splat_step* a, int adam_step, gps_stream stream) {
    GPS_REQUIRE(a != nullptr && adam_step >= 1);
    GPS_REQUIRE(a->gt_rgb && a->loss && a->v_render_colors && a->v_render_alphas);
    // Launch sites of one iteration with the strip backward's buffers (6): preprocess (+ the binning's histogram pass), row scan,
    // scatter (+ the backward's class lists), forward rasterizer (+ compose + L1 + image gradients in its epilogue), backward
    // rasterizer, preprocess backward (+ Adam).  Without them (8): preprocess (+ count pass + zero-fill of the rasterizer
    // gradients), expand, count table, row scan, scatter, forward, group backward, preprocess backward.
    GPS_REQUIRE(a->base_color != nullptr);
    const bool fused_fwd = a->records != nullptr;  // the record rasterizer carries the compose epilogue
    const bool strips = strips_on(a);              // (implies fused_fwd)
    gps::FwdCompose fc = {a->base_color, a->gt_rgb, a->rgb, a->loss, a->v_render_colors, a->v_render_alphas,
                          1.0f / (3.0f * (float)(a->width * a->height)), strips ? a->pix2 : nullptr};
    // group kernel: its four accumulator arrays are zero-filled by the preprocessing kernel; strips: every row is a plain store
    gps::ZeroGrads zg = {a->v_means2d, a->v_conics, a->v_colors, a->v_opacities};
    gps::ZeroGrads no_zero = {};
    GPS_REQUIRE(!a->preprocessed || gps_splat_can_prefetch(a));
    GPS_REQUIRE(!a->next_viewmat || (a->next_Kmat && a->next_cam_pos && gps_splat_can_prefetch(a)));
    int r = render_chain(a, fused_fwd ? &fc : nullptr, strips ? &no_zero : &zg, stream, a->preprocessed != 0);
    if (r != GPS_OK) return r;
    if (!fused_fwd) {
        r = gps_compose_l1(a->width, a->height, a->render_colors, a->weight_sum, a->base_color, nullptr, a->gt_rgb, a->rgb,
                           nullptr, a->loss, a->v_render_colors, a->v_render_alphas, stream);
        if (r != GPS_OK) return r;
    }
    if (strips)
        r = gps::raster_ges_bwd_strips_launch(a->N, a->records, a->radii, a->cls_ids, a->cls_counts, (int)a->cls_stride,
                                              a->v_render_colors, a->pix2, a->width, a->height, a->v_rows, stream);
    else
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
    // the next iteration's preprocessing in this kernel's tail: its binning count targets are the same carve as this iteration's
    gps::BinCountOut next_cnt;
    gps::NextForward next = {};
    if (a->next_viewmat) {
        const int tw = gps_div_up(a->width, 16), th = gps_div_up(a->height, 16);
        r = gps::isect_count_targets(a->N, a->isect_capacity, a->tiles_per_gauss, 16, tw, th, a->workspace, a->workspace_bytes, true, &next_cnt);
        if (r != GPS_OK) return r;
        next = {a->next_viewmat, a->next_Kmat, a->next_cam_pos, a->max_gs_radii, a->near_plane, a->far_plane, a->radius_clip,
                a->radii, a->means2d, a->depths, a->conics, a->colors, a->opacities, a->records, &next_cnt};
    }
    float sstep[5];
    for (int k = 0; k < 5; k++) sstep[k] = gps::adam_scalars(seg[k].lr, a->beta1, a->beta2, a->adam_eps, adam_step).step_size;
    r = gps::preprocess_bwd_launch(a->N, a->K, a->sh_degree, a->means, a->log_scales, a->quats, a->opac_logit, a->sh_dc,
                                   a->sh_rest, a->viewmat, a->Kmat, a->cam_pos, a->width, a->height, a->eps2d, a->radii,
                                   a->conics, a->v_means2d, a->v_conics, a->v_colors, a->v_opacities,
                                   all ? nullptr : a->g_means, a->g_log_scales, a->g_quats, a->g_opac_logit, a->g_sh_dc,
                                   fuse ? nullptr : a->g_sh_rest, fuse ? a->sh_rest : nullptr,
                                   fuse ? a->m_sh_rest : nullptr, fuse ? a->v_sh_rest : nullptr,
                                   gps::adam_scalars(a->lr[4], a->beta1, a->beta2, a->adam_eps, adam_step),
                                   all ? seg : nullptr, all ? sstep : nullptr, stream, strips ? a->v_rows : nullptr,
                                   a->next_viewmat ? &next : nullptr);
    if (r != GPS_OK || all) return r;
    return gps_adam_step(seg, fuse ? 5 : 6, a->beta1, a->beta2, a->adam_eps, adam_step, stream);
}

}  // extern "C"

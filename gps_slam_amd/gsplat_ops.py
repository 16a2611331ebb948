"""Host-side mirror of the reference's splat operator surface (gsplat/gsplat_wapper.hpp,
gsplat/rasterizer/bindings.h) on top of the C-ABI.

Each function takes/returns torch tensors resident on the GPU, allocates the outputs through
torch's caching allocator (as the reference's launchers do with torch::empty/zeros) and passes
raw pointers + the current HIP stream to libgpsslam_hip.so.  Nothing here computes: if the HIP
library is missing these functions raise.

Shapes follow the reference with the camera dimension C == 1 kept where the reference has it
(raw_gs_model.cpp:225-226 always unsqueezes one camera).
"""
import ctypes as C

import torch

from ._lib import AdamSegment, check, lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t):
    assert t.is_cuda and t.dtype == torch.float32, "expects float32 GPU tensors"
    return t.contiguous()


# ----------------------------------------------------------------------------- projection
def fully_fused_projection_fwd(means, quats, scales, viewmats, Ks, width, height, eps2d=0.3, near_plane=0.01,
                               far_plane=1e10, radius_clip=0.0):
    """gsplat::fully_fused_projection_fwd_tensor (fully_fused_projection_fwd.cu:196-273).
    means[N,3] quats[N,4] scales[N,3] viewmats[1,4,4] Ks[1,3,3] -> radii[1,N] i32, means2d[1,N,2],
    depths[1,N], conics[1,N,3]."""
    means, quats, scales = _f32c(means), _f32c(quats), _f32c(scales)
    viewmats, Ks = _f32c(viewmats), _f32c(Ks)
    assert viewmats.shape[0] == 1 and Ks.shape[0] == 1, "single camera (C == 1) as in raw_gs_model.cpp"
    N = means.shape[0]
    dev = means.device
    radii = torch.empty((1, N), dtype=torch.int32, device=dev)
    means2d = torch.empty((1, N, 2), dtype=torch.float32, device=dev)
    depths = torch.empty((1, N), dtype=torch.float32, device=dev)
    conics = torch.empty((1, N, 3), dtype=torch.float32, device=dev)
    check(lib.gps_proj_fwd(N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(viewmats), _ptr(Ks), width, height, eps2d,
                           near_plane, far_plane, radius_clip, _ptr(radii), _ptr(means2d), _ptr(depths), _ptr(conics),
                           _stream()), "gps_proj_fwd")
    return radii, means2d, depths, conics


def fully_fused_projection_bwd(means, quats, scales, viewmats, Ks, width, height, eps2d, radii, conics, v_means2d,
                               v_depths, v_conics):
    """gsplat::fully_fused_projection_bwd_tensor (fully_fused_projection_bwd.cu:288-403) -> v_means, v_quats, v_scales"""
    means, quats, scales = _f32c(means), _f32c(quats), _f32c(scales)
    viewmats, Ks, conics = _f32c(viewmats), _f32c(Ks), _f32c(conics)
    v_means2d, v_depths, v_conics = _f32c(v_means2d), _f32c(v_depths), _f32c(v_conics)
    radii = radii.contiguous()
    N = means.shape[0]
    v_means, v_quats, v_scales = torch.empty_like(means), torch.empty_like(quats), torch.empty_like(scales)
    check(lib.gps_proj_bwd(N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(viewmats), _ptr(Ks), width, height, eps2d,
                           _ptr(radii), _ptr(conics), _ptr(v_means2d), _ptr(v_depths), _ptr(v_conics), _ptr(v_means),
                           _ptr(v_quats), _ptr(v_scales), _stream()), "gps_proj_bwd")
    return v_means, v_quats, v_scales


# ----------------------------------------------------------------------------- spherical harmonics
def compute_sh_fwd(degrees_to_use, dirs, coeffs, masks=None):
    """gsplat::compute_sh_fwd_tensor (compute_sh_fwd.cu:40-72). dirs[...,3] coeffs[...,K,3] masks[...] -> [...,3]"""
    dirs, coeffs = _f32c(dirs), _f32c(coeffs)
    K = coeffs.shape[-2]
    N = coeffs.numel() // (K * 3)
    m = None if masks is None else masks.contiguous().to(torch.uint8)
    colors = torch.empty(dirs.shape, dtype=torch.float32, device=dirs.device)
    check(lib.gps_sh_fwd(N, K, degrees_to_use, _ptr(dirs), _ptr(coeffs), _ptr(m), _ptr(colors), _stream()),
          "gps_sh_fwd")
    return colors


def compute_sh_bwd(K, degrees_to_use, dirs, coeffs, masks, v_colors, compute_v_dirs=True):
    """gsplat::compute_sh_bwd_tensor (compute_sh_bwd.cu:56-123) -> v_coeffs, v_dirs"""
    dirs, coeffs, v_colors = _f32c(dirs), _f32c(coeffs), _f32c(v_colors)
    N = coeffs.numel() // (K * 3)
    m = None if masks is None else masks.contiguous().to(torch.uint8)
    v_coeffs = torch.empty_like(coeffs)
    v_dirs = torch.empty_like(dirs) if compute_v_dirs else None
    check(lib.gps_sh_bwd(N, K, degrees_to_use, _ptr(dirs), _ptr(coeffs), _ptr(m), _ptr(v_colors), _ptr(v_coeffs),
                         _ptr(v_dirs), _stream()), "gps_sh_bwd")
    return v_coeffs, v_dirs


# ----------------------------------------------------------------------------- binning
class IsectResult:
    """Device-resident result of isect_tiles_no_depth.  `counts` = int64[4] on the device:
    {n_isects, n_groups, overflow, n_visible}; buffers are capacity sized, valid prefix given by counts."""

    __slots__ = ("tiles_per_gauss", "isect_ids", "flatten_ids", "group_gs_ids", "group_starts", "isect_offsets",
                 "counts", "tile_width", "tile_height")

    def sizes(self):
        """Host copy of counts (this is the only place a sync happens; the hot loop never calls it)."""
        c = self.counts.cpu().tolist()
        if c[2]:
            raise RuntimeError("isect capacity overflow: raise isect_capacity/group_capacity")
        return int(c[0]), int(c[1])

    def trimmed(self):
        """The 5+1 tensors exactly as the reference returns them (exact sizes; syncs once)."""
        ni, ng = self.sizes()
        return (self.tiles_per_gauss, None if self.isect_ids is None else self.isect_ids[:ni], self.flatten_ids[:ni],
                self.group_gs_ids[:ng], self.group_starts[:ng], self.isect_offsets)


_WS = {}


def _workspace(dev, nbytes):
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _WS[key] = buf
    return buf


def isect_tiles_no_depth(means2d, radii, tile_size, tile_width, tile_height, isect_capacity=None, group_capacity=None,
                         want_isect_ids=False, out=None):
    """isectTilesNoDepth + isectOffsetEncodeNoDepth (gsplat_wapper.cpp:55-91, isect_tiles_no_depth.cu:132-461)
    in one call.  means2d[1,N,2], radii[1,N] (clamped).
    With explicit capacities (or `out` buffers) the call is sync-free and an overflow is REPORTED in counts[2] (the lists are then
    truncated).  With the default capacities the result must be complete, as the reference's exactly-sized tensors are: the
    overflow word is read back (the reference blocks on two .item() calls here, isect_tiles_no_depth.cu:238-239) and the call is
    repeated with doubled capacities until everything fits."""
    if isect_capacity is None and group_capacity is None and out is None:
        N_ = radii.numel()
        icap, gcap = max(1 << 20, 16 * N_), max(1 << 20, 32 * N_)
        while True:
            r = isect_tiles_no_depth(means2d, radii, tile_size, tile_width, tile_height, icap, gcap, want_isect_ids)
            if int(r.counts[2]) == 0:
                return r
            icap, gcap = 2 * icap, 2 * gcap
    means2d = _f32c(means2d)
    radii = radii.contiguous()
    assert radii.dtype == torch.int32
    N = radii.numel()
    dev = means2d.device
    icap = int(isect_capacity or max(1 << 20, 16 * N))
    gcap = int(group_capacity or max(1 << 20, 32 * N))
    r = out if out is not None else IsectResult()
    if out is None:
        r.tiles_per_gauss = torch.empty((1, N), dtype=torch.int32, device=dev)
        r.isect_ids = torch.empty(icap, dtype=torch.int64, device=dev) if want_isect_ids else None
        r.flatten_ids = torch.empty(icap, dtype=torch.int32, device=dev)
        r.group_gs_ids = torch.empty(gcap, dtype=torch.int32, device=dev)
        r.group_starts = torch.empty(gcap, dtype=torch.int32, device=dev)
        r.isect_offsets = torch.empty((1, tile_height, tile_width), dtype=torch.int32, device=dev)
        r.counts = torch.zeros(4, dtype=torch.int64, device=dev)
        r.tile_width, r.tile_height = tile_width, tile_height
    nbytes = lib.gps_isect_workspace_bytes(N, icap)
    ws = _workspace(dev, nbytes)
    check(lib.gps_isect_tiles_no_depth(N, _ptr(means2d), _ptr(radii), tile_size, tile_width, tile_height, icap, gcap,
                                       _ptr(r.tiles_per_gauss), _ptr(r.isect_ids), _ptr(r.flatten_ids),
                                       _ptr(r.group_gs_ids), _ptr(r.group_starts), _ptr(r.isect_offsets),
                                       _ptr(r.counts), _ptr(ws), ws.numel(), _stream()), "gps_isect_tiles_no_depth")
    return r


def isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, isect_capacity=None, want_isect_ids=True):
    """isectTiles + isectOffsetEncode (gsplat_wapper.cpp:3-48, isect_tiles.cu:30-430): depth-keyed binning for the `raw`
    render method, one call.  means2d[1,N,2], radii[1,N] (clamped), depths[1,N].  Explicit capacity: sync-free, overflow reported in
    counts[2]; default capacity: repeated with a doubled capacity until complete (see isect_tiles_no_depth)."""
    if isect_capacity is None:
        icap = max(1 << 20, 16 * radii.numel())
        while True:
            r = isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, icap, want_isect_ids)
            if int(r.counts[2]) == 0:
                return r
            icap *= 2
    means2d, depths = _f32c(means2d), _f32c(depths)
    radii = radii.contiguous()
    assert radii.dtype == torch.int32
    N = radii.numel()
    dev = means2d.device
    icap = int(isect_capacity or max(1 << 20, 16 * N))
    r = IsectResult()
    r.tiles_per_gauss = torch.empty((1, N), dtype=torch.int32, device=dev)
    r.isect_ids = torch.empty(icap, dtype=torch.int64, device=dev) if want_isect_ids else None
    r.flatten_ids = torch.empty(icap, dtype=torch.int32, device=dev)
    r.group_gs_ids = r.group_starts = None
    r.isect_offsets = torch.empty((1, tile_height, tile_width), dtype=torch.int32, device=dev)
    r.counts = torch.zeros(4, dtype=torch.int64, device=dev)
    r.tile_width, r.tile_height = tile_width, tile_height
    ws = _workspace(dev, lib.gps_isect_workspace_bytes(N, icap))
    check(lib.gps_isect_tiles(N, _ptr(means2d), _ptr(radii), _ptr(depths), tile_size, tile_width, tile_height, icap,
                              _ptr(r.tiles_per_gauss), _ptr(r.isect_ids), _ptr(r.flatten_ids), _ptr(r.isect_offsets),
                              _ptr(r.counts), _ptr(ws), ws.numel(), _stream()), "gps_isect_tiles")
    return r


# ----------------------------------------------------------------------------- rasterizer
def rasterize_to_pixels_fwd_ges(means2d, conics, colors, opacities, ref_depth_map, width, height, tile_size,
                                isect, delta_depth, want_last_ids=False):
    """gsplat::rasterize_to_pixels_fwd_ges_tensor (rasterize_to_pixels_fwd_ges.cu:223-407), COLOR_DIM = 4.
    -> render_colors[1,H,W,4], render_alphas[1,H,W,1] (weight sum), last_ids[1,H,W] | None"""
    means2d, conics, colors, opacities = _f32c(means2d), _f32c(conics), _f32c(colors), _f32c(opacities)
    ref_depth_map = _f32c(ref_depth_map)
    assert colors.shape[-1] == 4, "ges path renders rgb + depth (raw_gs_model.cpp:286)"
    N = opacities.numel()
    dev = means2d.device
    rc = torch.empty((1, height, width, 4), dtype=torch.float32, device=dev)
    ra = torch.empty((1, height, width, 1), dtype=torch.float32, device=dev)
    last = torch.empty((1, height, width), dtype=torch.int32, device=dev) if want_last_ids else None
    check(lib.gps_raster_ges_fwd(N, _ptr(means2d), _ptr(conics), _ptr(colors), _ptr(opacities), _ptr(ref_depth_map),
                                 width, height, tile_size, _ptr(isect.isect_offsets), _ptr(isect.flatten_ids),
                                 _ptr(isect.counts), delta_depth, _ptr(rc), _ptr(ra), _ptr(last), _stream()),
          "gps_raster_ges_fwd")
    return rc, ra, last


def rasterize_to_pixels_bwd_ges_gs_parallel(means2d, conics, colors, opacities, radii, ref_depth_map, width, height,
                                            isect, delta_depth, v_render_colors, v_render_alphas, out=None,
                                            accumulate=False):
    """gsplat::rasterize_to_pixels_bwd_ges_gs_parallel_tensor (rasterize_to_pixels_bwd_ges_new_parallel.cu:203-385)
    -> v_means2d[1,N,2], v_conics[1,N,3], v_colors[1,N,4], v_opacities (shape of opacities)"""
    means2d, conics, colors, opacities = _f32c(means2d), _f32c(conics), _f32c(colors), _f32c(opacities)
    ref_depth_map, v_render_colors, v_render_alphas = _f32c(ref_depth_map), _f32c(v_render_colors), _f32c(v_render_alphas)
    radii = radii.contiguous()
    N = opacities.numel()
    if out is None:
        assert not accumulate
        out = (torch.empty_like(means2d), torch.empty_like(conics), torch.empty_like(colors),
               torch.empty_like(opacities))
    v_m, v_c, v_col, v_o = out
    check(lib.gps_raster_ges_bwd_gs(N, _ptr(means2d), _ptr(conics), _ptr(colors), _ptr(opacities), _ptr(radii),
                                    _ptr(ref_depth_map), width, height, _ptr(isect.group_gs_ids),
                                    _ptr(isect.group_starts), _ptr(isect.counts), delta_depth, _ptr(v_render_colors),
                                    _ptr(v_render_alphas), _ptr(v_m), _ptr(v_c), _ptr(v_col), _ptr(v_o),
                                    1 if accumulate else 0, _stream()),
          "gps_raster_ges_bwd_gs")
    return v_m, v_c, v_col, v_o


def rasterize_to_pixels_bwd_ges_strips(means2d, conics, colors, opacities, radii, ref_depth_map, width, height, delta_depth,
                                       v_render_colors, v_render_alphas):
    """The same operator through the kernel the fused train step runs (gps_raster_ges_bwd_strips: column strips, one 48-byte
    gradient row per Gaussian, no group table): records, the {v_alpha, depth cut} pair image and the class lists are built here
    from the operator-level arrays (the train step's binning writes them on the device; this wrapper, used by the parity tests,
    orders the lists with torch).  Same outputs as rasterize_to_pixels_bwd_ges_gs_parallel; untouched rows are zero."""
    means2d, conics, colors, opacities = _f32c(means2d), _f32c(conics), _f32c(colors), _f32c(opacities)
    ref_depth_map, v_render_colors, v_render_alphas = _f32c(ref_depth_map), _f32c(v_render_colors), _f32c(v_render_alphas)
    radii = radii.contiguous().view(-1)
    N, dev = opacities.numel(), means2d.device
    recs = torch.empty((max(N, 1), 12), dtype=torch.float32, device=dev)
    check(lib.gps_raster_pack_records(N, _ptr(means2d), _ptr(conics), _ptr(colors), _ptr(opacities), _ptr(radii), _ptr(recs), _stream()),
          "gps_raster_pack_records")
    pix2 = torch.empty((height * width, 2), dtype=torch.float32, device=dev)
    check(lib.gps_raster_pair_image(width, height, _ptr(v_render_alphas), _ptr(ref_depth_map), delta_depth, _ptr(pix2), _stream()),
          "gps_raster_pair_image")
    cls = torch.full((N,), -1, dtype=torch.long, device=dev)
    vis = radii > 0
    cls[vis] = torch.bucketize(radii[vis].long(), torch.tensor([4, 8, 16, 32], device=dev), right=False)
    ids = torch.zeros((5, max(N, 1)), dtype=torch.int32, device=dev)
    counts = torch.zeros(8, dtype=torch.int32, device=dev)
    for k in range(5):
        sel = torch.nonzero(cls == k)[:, 0].int()
        ids[k, :sel.numel()] = sel
        counts[k] = sel.numel()
    rows = torch.zeros((max(N, 1), 12), dtype=torch.float32, device=dev)
    check(lib.gps_raster_ges_bwd_strips(N, _ptr(recs), _ptr(radii), _ptr(ids), _ptr(counts), max(N, 1), _ptr(v_render_colors), _ptr(pix2),
                                        width, height, _ptr(rows), _stream()), "gps_raster_ges_bwd_strips")
    rows = rows[:N]
    return (rows[:, 7:9].reshape(means2d.shape).contiguous(), rows[:, 4:7].reshape(conics.shape).contiguous(),
            rows[:, 0:4].reshape(colors.shape).contiguous(), rows[:, 9].reshape(opacities.shape).contiguous())


def rasterize_to_pixels_bwd_ges_exact(means2d, conics, colors, opacities, ref_depth_map, width, height, tile_size, isect, delta_depth,
                                      v_render_colors, v_render_alphas):
    """gsplat::rasterize_to_pixels_bwd_ges_tensor (rasterize_to_pixels_bwd_ges.cu:164-291): the exact tile-parallel adjoint of the
    ges forward -> v_means2d, v_conics, v_colors, v_opacities"""
    means2d, conics, colors, opacities = _f32c(means2d), _f32c(conics), _f32c(colors), _f32c(opacities)
    ref_depth_map, v_render_colors, v_render_alphas = _f32c(ref_depth_map), _f32c(v_render_colors), _f32c(v_render_alphas)
    N = opacities.numel()
    out = (torch.empty_like(means2d), torch.empty_like(conics), torch.empty_like(colors), torch.empty_like(opacities))
    check(lib.gps_raster_ges_bwd_exact(N, _ptr(means2d), _ptr(conics), _ptr(colors), _ptr(opacities), _ptr(ref_depth_map), width, height,
                                       tile_size, _ptr(isect.isect_offsets), _ptr(isect.flatten_ids), _ptr(isect.counts), delta_depth,
                                       _ptr(v_render_colors), _ptr(v_render_alphas), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]),
                                       _stream()), "gps_raster_ges_bwd_exact")
    return out


# ----------------------------------------------------------------------------- compose + loss, Adam
def compose_l1(render_colors, weight_sum, base_color, ref_depth_raw, gt_rgb, need_grad=True, need_depth=True):
    """Fused raw_gs_model.cpp:318-326 + computeLoss (:369-417, L1 only) + backward.
    -> rgb[H,W,3], depth[H,W,1]|None, loss[1], v_render_colors[1,H,W,4]|None, v_render_alphas[1,H,W,1]|None"""
    render_colors, weight_sum = _f32c(render_colors), _f32c(weight_sum)
    base_color = _f32c(base_color)
    gt_rgb = None if gt_rgb is None else _f32c(gt_rgb)
    need_grad = need_grad and gt_rgb is not None
    H, W = render_colors.shape[-3], render_colors.shape[-2]
    dev = render_colors.device
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    depth = torch.empty((H, W, 1), dtype=torch.float32, device=dev) if need_depth else None
    ref = _f32c(ref_depth_raw) if need_depth else None
    loss = torch.zeros(1, dtype=torch.float32, device=dev) if gt_rgb is not None else None
    v_rc = torch.empty_like(render_colors) if need_grad else None
    v_ra = torch.empty_like(weight_sum) if need_grad else None
    check(lib.gps_compose_l1(W, H, _ptr(render_colors), _ptr(weight_sum), _ptr(base_color), _ptr(ref), _ptr(gt_rgb),
                             _ptr(rgb), _ptr(depth), _ptr(loss), _ptr(v_rc), _ptr(v_ra), _stream()), "gps_compose_l1")
    return rgb, depth, loss, v_rc, v_ra


def adam_step(params, grads, exp_avgs, exp_avg_sqs, lrs, step, betas=(0.9, 0.999), eps=1e-15):
    """One fused step over all parameter tensors (7 x torch::optim::Adam::step, raw_gs_model.cpp:654-705)."""
    n = len(params)
    segs = (AdamSegment * n)()
    for k in range(n):
        p, g, m, v = params[k], grads[k], exp_avgs[k], exp_avg_sqs[k]
        assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
        segs[k].param, segs[k].grad = p.data_ptr(), g.data_ptr()
        segs[k].exp_avg, segs[k].exp_avg_sq = m.data_ptr(), v.data_ptr()
        segs[k].numel, segs[k].lr = p.numel(), float(lrs[k])
    check(lib.gps_adam_step(segs, n, float(betas[0]), float(betas[1]), float(eps), int(step), _stream()),
          "gps_adam_step")


# ----------------------------------------------------------------------------- fused model-level kernels
def gauss_preprocess_fwd(means, log_scales, quats, opac_logit, sh_dc, sh_rest, sh_degree, viewmat, Kmat, cam_pos,
                         width, height, eps2d=0.3, near_plane=0.01, far_plane=1e10, radius_clip=0.0, max_gs_radii=100,
                         out=None, records=None):
    """One pass of raw_gs_model.cpp:207-286 (see include/gps_slam_hip.h: gps_gauss_preprocess_fwd).
    -> radii[N] i32 (clamped), means2d[N,2], depths[N], conics[N,3], colors[N,4], opacities[N]"""
    N = means.shape[0]
    K = 1 + (sh_rest.shape[1] if sh_rest is not None and sh_rest.numel() else 0)
    dev = means.device
    if out is None:
        out = (torch.empty(N, dtype=torch.int32, device=dev), torch.empty((N, 2), dtype=torch.float32, device=dev),
               torch.empty(N, dtype=torch.float32, device=dev), torch.empty((N, 3), dtype=torch.float32, device=dev),
               torch.empty((N, 4), dtype=torch.float32, device=dev), torch.empty(N, dtype=torch.float32, device=dev))
    radii, means2d, depths, conics, colors, opac = out
    check(lib.gps_gauss_preprocess_fwd(N, K, sh_degree, _ptr(means), _ptr(log_scales), _ptr(quats), _ptr(opac_logit),
                                       _ptr(sh_dc), _ptr(sh_rest), _ptr(viewmat), _ptr(Kmat), _ptr(cam_pos), width,
                                       height, eps2d, near_plane, far_plane, radius_clip, int(max_gs_radii), _ptr(radii),
                                       _ptr(means2d), _ptr(depths), _ptr(conics), _ptr(colors), _ptr(opac),
                                       _ptr(records), _stream()),
          "gps_gauss_preprocess_fwd")
    return out


def gauss_preprocess_bwd(means, log_scales, quats, opac_logit, sh_dc, sh_rest, sh_degree, viewmat, Kmat, cam_pos,
                         width, height, eps2d, radii, conics, v_means2d, v_conics, v_colors, v_opacities, out=None):
    """Adjoint of gauss_preprocess_fwd -> v_means, v_log_scales, v_quats, v_opac_logit, v_sh_dc, v_sh_rest"""
    N = means.shape[0]
    K = 1 + (sh_rest.shape[1] if sh_rest is not None and sh_rest.numel() else 0)
    if out is None:
        out = (torch.empty_like(means), torch.empty_like(log_scales), torch.empty_like(quats),
               torch.empty_like(opac_logit), torch.empty_like(sh_dc), torch.empty_like(sh_rest))
    v_means, v_ls, v_q, v_ol, v_dc, v_rest = out
    check(lib.gps_gauss_preprocess_bwd(N, K, sh_degree, _ptr(means), _ptr(log_scales), _ptr(quats), _ptr(opac_logit),
                                       _ptr(sh_dc), _ptr(sh_rest), _ptr(viewmat), _ptr(Kmat), _ptr(cam_pos), width,
                                       height, eps2d, _ptr(radii), _ptr(conics), _ptr(v_means2d), _ptr(v_conics),
                                       _ptr(v_colors), _ptr(v_opacities), _ptr(v_means), _ptr(v_ls), _ptr(v_q),
                                       _ptr(v_ol), _ptr(v_dc), _ptr(v_rest), _stream()), "gps_gauss_preprocess_bwd")
    return out


def rasterize_to_pixels_fwd_ges_rec(records, ref_depth_map, width, height, isect, delta_depth):
    """gps_raster_ges_fwd_rec: the record-streaming forward (same result as rasterize_to_pixels_fwd_ges)."""
    N = records.shape[0]
    dev = records.device
    rc = torch.empty((1, height, width, 4), dtype=torch.float32, device=dev)
    ra = torch.empty((1, height, width, 1), dtype=torch.float32, device=dev)
    check(lib.gps_raster_ges_fwd_rec(N, _ptr(records), _ptr(_f32c(ref_depth_map)), width, height,
                                     _ptr(isect.isect_offsets), _ptr(isect.flatten_ids), _ptr(isect.counts), delta_depth,
                                     _ptr(rc), _ptr(ra), _stream()), "gps_raster_ges_fwd_rec")
    return rc, ra


# ----------------------------------------------------------------------------- `raw` rasterizer
def rasterize_to_pixels_fwd(means2d, conics, colors, opacities, backgrounds, width, height, tile_size, isect):
    """gsplat::rasterize_to_pixels_fwd_tensor (rasterize_to_pixels_fwd.cu:206-376), COLOR_DIM = 4, one camera.
    -> render_colors[1,H,W,4], render_alphas[1,H,W,1] (= 1 - T), last_ids[1,H,W]"""
    means2d, conics, colors, opacities = _f32c(means2d), _f32c(conics), _f32c(colors), _f32c(opacities)
    backgrounds = _f32c(backgrounds) if backgrounds is not None else None
    assert colors.shape[-1] == 4, "the raw path renders rgb + depth (raw_gs_model.cpp:117)"
    N = opacities.numel()
    dev = means2d.device
    rc = torch.empty((1, height, width, 4), dtype=torch.float32, device=dev)
    ra = torch.empty((1, height, width, 1), dtype=torch.float32, device=dev)
    last = torch.empty((1, height, width), dtype=torch.int32, device=dev)
    check(lib.gps_raster_raw_fwd(N, _ptr(means2d), _ptr(conics), _ptr(colors), _ptr(opacities), _ptr(backgrounds), width,
                                 height, tile_size, _ptr(isect.isect_offsets), _ptr(isect.flatten_ids), _ptr(isect.counts),
                                 _ptr(rc), _ptr(ra), _ptr(last), _stream()), "gps_raster_raw_fwd")
    return rc, ra, last


def rasterize_to_pixels_bwd(means2d, conics, colors, opacities, backgrounds, width, height, tile_size, isect, render_alphas,
                            last_ids, v_render_colors, v_render_alphas, absgrad=False):
    """gsplat::rasterize_to_pixels_bwd_tensor (rasterize_to_pixels_bwd.cu:299-470)
    -> v_means2d_abs[1,N,2] | None, v_means2d[1,N,2], v_conics[1,N,3], v_colors[1,N,4], v_opacities (shape of opacities)"""
    means2d, conics, colors, opacities = _f32c(means2d), _f32c(conics), _f32c(colors), _f32c(opacities)
    backgrounds = _f32c(backgrounds) if backgrounds is not None else None
    render_alphas, v_render_colors, v_render_alphas = _f32c(render_alphas), _f32c(v_render_colors), _f32c(v_render_alphas)
    last_ids = last_ids.contiguous()
    N = opacities.numel()
    v_means2d = torch.empty_like(means2d)
    v_abs = torch.empty_like(means2d) if absgrad else None
    v_conics = torch.empty_like(conics)
    v_colors = torch.empty_like(colors)
    v_opac = torch.empty_like(opacities)
    check(lib.gps_raster_raw_bwd(N, _ptr(means2d), _ptr(conics), _ptr(colors), _ptr(opacities), _ptr(backgrounds), width,
                                 height, tile_size, _ptr(isect.isect_offsets), _ptr(isect.flatten_ids), _ptr(isect.counts),
                                 _ptr(render_alphas), _ptr(last_ids), _ptr(v_render_colors), _ptr(v_render_alphas),
                                 _ptr(v_abs), _ptr(v_means2d), _ptr(v_conics), _ptr(v_colors), _ptr(v_opac), _stream()),
          "gps_raster_raw_bwd")
    return v_abs, v_means2d, v_conics, v_colors, v_opac


# ----------------------------------------------------------------------------- fused SSIM
def fusedssim(C1, C2, img1, img2, train=True, channels_last=False):
    """gsplat fusedssim (ssim.cu:385-421).  img [B,CH,H,W] (or [B,H,W,CH] with channels_last) -> ssim_map, dm_dmu1,
    dm_dsigma1_sq, dm_dsigma12 (the three are None when not training), all in the images' layout."""
    img1, img2 = _f32c(img1), _f32c(img2)
    assert img1.dim() == 4 and img1.shape == img2.shape
    if channels_last:
        B, H, W, CH = img1.shape
    else:
        B, CH, H, W = img1.shape
    m = torch.empty_like(img1)
    d1, d2, d3 = (torch.empty_like(img1) for _ in range(3)) if train else (None, None, None)
    check(lib.gps_ssim_fwd(B, CH, H, W, int(channels_last), C1, C2, _ptr(img1), _ptr(img2), _ptr(m), _ptr(d1), _ptr(d2), _ptr(d3),
                           _stream()), "gps_ssim_fwd")
    return m, d1, d2, d3


def fusedssim_backward(C1, C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, channels_last=False):
    """gsplat fusedssim_backward (ssim.cu:423-460) -> dL_dimg1"""
    img1, img2, dL_dmap = _f32c(img1), _f32c(img2), _f32c(dL_dmap)
    if channels_last:
        B, H, W, CH = img1.shape
    else:
        B, CH, H, W = img1.shape
    g = torch.empty_like(img1)
    check(lib.gps_ssim_bwd(B, CH, H, W, int(channels_last), _ptr(img1), _ptr(img2), _ptr(dL_dmap), _ptr(dm_dmu1), _ptr(dm_dsigma1_sq),
                           _ptr(dm_dsigma12), _ptr(g), _stream()), "gps_ssim_bwd")
    return g


"""Mirror of gsplat/gsplat_wapper.{hpp,cpp}: the autograd operator surface raw_gs_model.cpp programs against.

Same class / function names, argument order and tensor shapes as the reference's torch::autograd::Function
classes; forward/backward call the C-ABI launchers in gsplat_ops.py.  (The optimisation loop itself uses the fused
gps_splat_train_step; these classes are the drop-in surface and what the end-to-end gradient test differentiates.)
"""
import numpy as np
import torch

from . import gsplat_ops as ops


class SphericalHarmonicsNew(torch.autograd.Function):
    """gsplat_wapper.hpp:16-95: apply(sh_degree, dirs[...,3], coeffs[...,K,3], masks[...]) -> colors[...,3]"""

    @staticmethod
    def forward(ctx, sh_degree, dirs, coeffs, masks):
        colors = ops.compute_sh_fwd(sh_degree, dirs, coeffs, masks)
        ctx.save_for_backward(dirs, coeffs, masks)
        ctx.sh_degree, ctx.K = sh_degree, coeffs.shape[-2]
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        dirs, coeffs, masks = ctx.saved_tensors
        compute_v_dirs = ctx.needs_input_grad[1]
        v_coeffs, v_dirs = ops.compute_sh_bwd(ctx.K, ctx.sh_degree, dirs, coeffs, masks, v_colors.contiguous(),
                                              compute_v_dirs)
        return None, v_dirs, v_coeffs, None


class FullyFusedProjection(torch.autograd.Function):
    """gsplat_wapper.hpp:97-241: apply(means, covars(None), quats, scales, viewmats[C,4,4], Ks[C,3,3], width, height,
    eps2d, near_plane, far_plane, radius_clip, calc_compensations, camera_model)
    -> radii, means2d, depths, conics, compensations(None)"""

    @staticmethod
    def forward(ctx, means, covars, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                radius_clip, calc_compensations, camera_model):
        if covars is not None or calc_compensations or camera_model != "pinhole":
            raise RuntimeError("gfx950 path implements the configuration GPS-SLAM ships (raw_gs_model.h:283-288)")
        radii, means2d, depths, conics = ops.fully_fused_projection_fwd(means, quats, scales, viewmats, Ks, width,
                                                                        height, eps2d, near_plane, far_plane, radius_clip)
        ctx.save_for_backward(means, quats, scales, viewmats, Ks, radii, conics)
        ctx.cfg = (width, height, eps2d)
        ctx.mark_non_differentiable(radii)
        return radii, means2d, depths, conics

    @staticmethod
    def backward(ctx, v_radii, v_means2d, v_depths, v_conics):
        means, quats, scales, viewmats, Ks, radii, conics = ctx.saved_tensors
        width, height, eps2d = ctx.cfg
        v_means, v_quats, v_scales = ops.fully_fused_projection_bwd(means, quats, scales, viewmats, Ks, width, height,
                                                                    eps2d, radii, conics, v_means2d.contiguous(),
                                                                    v_depths.contiguous(), v_conics.contiguous())
        return (v_means, None, v_quats, v_scales) + (None,) * 10


class RasterizeToPixelsGes_NewParallel(torch.autograd.Function):
    """gsplat_wapper.hpp:489-620: apply(means2d, conics, colors, opacities, radiis, ref_depth_map, base_color_map,
    backgrounds, masks, width, height, tile_size, isect_offsets, flatten_ids, group_gs_ids, group_starts, absgrad,
    delta_depth) -> render_colors[1,H,W,4], weight_sum[1,H,W,1].
    isect_offsets/flatten_ids/group_* are the fields of the IsectResult returned by isectTilesNoDepth (they stay on
    the device together with their counts)."""

    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, radiis, ref_depth_map, base_color_map, backgrounds, masks,
                width, height, tile_size, isect, absgrad, delta_depth):
        if backgrounds is not None or masks is not None or absgrad:
            raise RuntimeError("backgrounds / masks / absgrad are never used by GPS-SLAM and are not implemented")
        rc, ra, _ = ops.rasterize_to_pixels_fwd_ges(means2d, conics, colors, opacities, ref_depth_map, width, height,
                                                    tile_size, isect, delta_depth)
        ctx.save_for_backward(means2d, conics, colors, opacities, radiis, ref_depth_map)
        ctx.cfg = (width, height, isect, delta_depth)
        return rc, ra

    @staticmethod
    def backward(ctx, v_render_colors, v_render_alphas):
        means2d, conics, colors, opacities, radiis, ref_depth_map = ctx.saved_tensors
        width, height, isect, delta_depth = ctx.cfg
        v_m, v_c, v_col, v_o = ops.rasterize_to_pixels_bwd_ges_gs_parallel(
            means2d, conics, colors, opacities, radiis, ref_depth_map, width, height, isect, delta_depth,
            v_render_colors.contiguous(), v_render_alphas.contiguous())
        return (v_m, v_c, v_col, v_o) + (None,) * 11


def isectTilesNoDepth(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True, **kw):
    """gsplat_wapper.cpp:55-85 (+ the offsets of isectOffsetEncodeNoDepth, computed in the same sync-free call)."""
    return ops.isect_tiles_no_depth(means2d, radii, tile_size, tile_width, tile_height, **kw)


def isectOffsetEncodeNoDepth(isect, n_cameras, tile_width, tile_height):
    """gsplat_wapper.cpp:88-91"""
    assert n_cameras == 1
    return isect.isect_offsets


def ges_forward(params, cam_dev, width, height, ref_depth, base_color, sh_degree=3, max_gs_radii=100, delta_depth=0.1,
                tile_size=16, eps2d=0.3, near_plane=0.01, far_plane=1e10, radius_clip=0.0):
    """RawGaussianModel::gesForward (src/raw_gs_model.cpp:188-367) written against the operator surface above, with
    the reference's libtorch glue as torch ops -- differentiable through autograd."""
    import math
    means, log_scales, quats, dc, rest, opac_logit = params
    tw, th = math.ceil(width / tile_size), math.ceil(height / tile_size)
    ref_clamped = torch.where(ref_depth < 0.01, torch.full_like(ref_depth, 1000.0), ref_depth)
    scales = torch.exp(log_scales)
    radii, means2d, depths, conics = FullyFusedProjection.apply(
        means, None, quats, scales, cam_dev["viewmat"].unsqueeze(0), cam_dev["K"].unsqueeze(0), width, height, eps2d,
        near_plane, far_plane, radius_clip, False, "pinhole")
    if max_gs_radii > 0:
        radii = torch.clamp_max(radii, max_gs_radii)
    shs = torch.cat([dc[:, None, :], rest], 1)
    dirs = means - cam_dev["cam_pos"][None, :]
    colors = SphericalHarmonicsNew.apply(sh_degree, dirs.unsqueeze(0), shs.unsqueeze(0), radii > 0)
    colors = torch.clamp_min(colors + 0.5, 0.0)
    isect = isectTilesNoDepth(means2d, radii, depths, tile_size, tw, th)
    colors = torch.cat([colors, depths.unsqueeze(-1)], 2)
    rc, ws = RasterizeToPixelsGes_NewParallel.apply(means2d, conics, colors, torch.sigmoid(opac_logit), radii,
                                                    ref_clamped, base_color, None, None, width, height, tile_size,
                                                    isect, False, delta_depth)
    raw_rgb, raw_depth = rc[..., :3], rc[..., 3:]
    bcw = torch.ones_like(ws)
    rgb = (raw_rgb + base_color * bcw) / (ws + bcw)
    bdw = torch.zeros_like(ws).masked_fill(ref_depth > 0, 1)
    depth = (raw_depth + ref_depth * bdw) / (ws + bdw)
    return dict(rgb=rgb[0], depth=depth[0], alpha=ws[0], radiis=radii[0], means2d=means2d)


# ----------------------------------------------------------------------------- `raw` render method
class RasterizeToPixels(torch.autograd.Function):
    """gsplat_wapper.hpp:243-353: apply(means2d, conics, colors, opacities, backgrounds, masks, width, height, tile_size,
    isect, absgrad) -> render_colors[1,H,W,4], render_alphas[1,H,W,1].  `isect` is the IsectResult of isectTiles.
    As in the reference the backward runs without the backgrounds (gsplat_wapper.hpp:307-315) and v_backgrounds is
    sum(v_render_colors * (1 - render_alphas))."""

    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, masks, width, height, tile_size, isect, absgrad):
        if masks is not None:
            raise RuntimeError("tile masks are never used by GPS-SLAM and are not implemented")
        rc, ra, last = ops.rasterize_to_pixels_fwd(means2d, conics, colors, opacities, backgrounds, width, height,
                                                   tile_size, isect)
        ctx.save_for_backward(means2d, conics, colors, opacities, ra, last)
        ctx.cfg = (width, height, tile_size, isect, absgrad)
        return rc, ra

    @staticmethod
    def backward(ctx, v_render_colors, v_render_alphas):
        means2d, conics, colors, opacities, ra, last = ctx.saved_tensors
        width, height, tile_size, isect, absgrad = ctx.cfg
        v_abs, v_m, v_c, v_col, v_o = ops.rasterize_to_pixels_bwd(
            means2d, conics, colors, opacities, None, width, height, tile_size, isect, ra, last,
            v_render_colors.contiguous(), v_render_alphas.contiguous(), absgrad=absgrad)
        v_bg = None
        if ctx.needs_input_grad[4]:
            v_bg = (v_render_colors * (1.0 - ra)).sum((1, 2))
        return (v_m, v_c, v_col, v_o, v_bg) + (None,) * 6


def isectTiles(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True, **kw):
    """gsplat_wapper.cpp:15-43 (+ the offsets of isectOffsetEncode, computed in the same sync-free call)."""
    assert sort
    return ops.isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, **kw)


def isectOffsetEncode(isect, n_cameras, tile_width, tile_height):
    """gsplat_wapper.cpp:45-48"""
    assert n_cameras == 1
    return isect.isect_offsets


def raw_forward(params, cam_dev, width, height, sh_degree=3, tile_size=16, eps2d=0.3, near_plane=0.01, far_plane=1e10,
                radius_clip=0.0, backgrounds=None, abs_grad=False):
    """RawGaussianModel::rawForward (src/raw_gs_model.cpp:43-185) on the operator surface above."""
    import math
    means, log_scales, quats, dc, rest, opac_logit = params
    tw, th = math.ceil(width / tile_size), math.ceil(height / tile_size)
    radii, means2d, depths, conics = FullyFusedProjection.apply(
        means, None, quats, torch.exp(log_scales), cam_dev["viewmat"].unsqueeze(0), cam_dev["K"].unsqueeze(0), width, height,
        eps2d, near_plane, far_plane, radius_clip, False, "pinhole")
    shs = torch.cat([dc[:, None, :], rest], 1)
    dirs = means - cam_dev["cam_pos"][None, :]
    colors = SphericalHarmonicsNew.apply(sh_degree, dirs.unsqueeze(0), shs.unsqueeze(0), radii > 0)
    colors = torch.clamp_min(colors + 0.5, 0.0)
    isect = isectTiles(means2d, radii, depths, tile_size, tw, th)
    colors = torch.cat([colors, depths.unsqueeze(-1)], 2)
    rc, ra = RasterizeToPixels.apply(means2d, conics, colors, torch.sigmoid(opac_logit), backgrounds, None, width, height,
                                     tile_size, isect, abs_grad)
    rgb, raw_depth = rc[..., :3], rc[..., 3:]
    depth = raw_depth / ra.clamp(1e-10)
    return dict(rgb=rgb[0], depth=depth[0], alpha=ra[0], radiis=radii[0], means2d=means2d)


# ----------------------------------------------------------------------------- fused SSIM
class FusedSSIMMap(torch.autograd.Function):
    """gsplat_wapper.hpp:622-677: apply(C1, C2, img1[B,CH,H,W], img2, padding, train) -> ssim_map; padding == "valid" crops 5
    pixels per side.  Gradient w.r.t. img1 only, like the reference.  Permuted views of [H,W,CH] images (what
    raw_gs_model.cpp:390-395 passes) are read in place through the channels-last layout of the kernels -- no copy."""

    @staticmethod
    def forward(ctx, C1, C2, img1, img2, padding="same", train=True):
        cl = _is_channels_last_view(img1) and _is_channels_last_view(img2)
        a, b = (img1.permute(0, 2, 3, 1), img2.permute(0, 2, 3, 1)) if cl else (img1.contiguous(), img2.contiguous())
        m, d1, d2, d3 = ops.fusedssim(C1, C2, a, b, train=train, channels_last=cl)
        ctx.save_for_backward(a, b, d1, d2, d3)
        ctx.cfg = (C1, C2, padding, cl)
        if cl:
            m = m.permute(0, 3, 1, 2)
        return m[:, :, 5:-5, 5:-5] if padding == "valid" else m

    @staticmethod
    def backward(ctx, dL_dmap):
        a, b, d1, d2, d3 = ctx.saved_tensors
        C1, C2, padding, cl = ctx.cfg
        if padding == "valid":
            B, CH = dL_dmap.shape[:2]
            full = torch.zeros((B, CH, dL_dmap.shape[2] + 10, dL_dmap.shape[3] + 10), dtype=dL_dmap.dtype, device=dL_dmap.device)
            full[:, :, 5:-5, 5:-5] = dL_dmap
            dL_dmap = full
        if cl:
            dL_dmap = dL_dmap.permute(0, 2, 3, 1)
        g = ops.fusedssim_backward(C1, C2, a, b, dL_dmap.contiguous(), d1, d2, d3, channels_last=cl)
        if cl:
            g = g.permute(0, 3, 1, 2)
        return None, None, g, None, None, None


def _is_channels_last_view(t):
    """[B,CH,H,W] tensor whose memory is a contiguous [B,H,W,CH] array (e.g. rgb.permute(2,0,1).unsqueeze(0))"""
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous() and not t.is_contiguous()


def compute_loss(render_res, gt_rgb, gt_depth=None, has_depth=False, ssim_weight=0.0, depth_weight=0.0, mask=None):
    """RawGaussianModel::computeLoss (raw_gs_model.cpp:369-417) -> {"total", "rgb"[, "depth"]}"""
    rgb = render_res["rgb"]
    l1 = (gt_rgb[mask] - rgb[mask]).abs().mean() if mask is not None else (gt_rgb - rgb).abs().mean()
    if ssim_weight > 0:
        C1, C2 = float(np.float32(0.01 * 0.01)), float(np.float32(0.03 * 0.03))
        ssim = FusedSSIMMap.apply(C1, C2, rgb.permute(2, 0, 1).unsqueeze(0), gt_rgb.permute(2, 0, 1).unsqueeze(0), "valid", True)
        rgb_loss = (1.0 - ssim_weight) * l1 + ssim_weight * (1.0 - ssim.mean())
    else:
        rgb_loss = l1
    loss = dict(total=rgb_loss, rgb=rgb_loss)
    if depth_weight > 0 and has_depth:
        valid = (gt_depth > 0) & (render_res["depth"] > 0)
        loss["depth"] = (gt_depth[valid] - render_res["depth"][valid]).abs().mean()
        loss["total"] = loss["total"] + depth_weight * loss["depth"]
    return loss

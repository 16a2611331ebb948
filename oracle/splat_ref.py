"""ctypes front-end for oracle/liboracle_splat.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  numpy in, numpy out; every function mirrors one reference
launcher (see splat_oracle.c for the file:line citations).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle_splat.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", _HERE, so])
        _LIB = C.CDLL(so)
    return _LIB


def _p(a, t=None):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle needs contiguous arrays"
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def proj_fwd(means, quats, scales, viewmat, K, W, H, eps2d=0.3, near=0.01, far=1e10, radius_clip=0.0):
    means, quats, scales, viewmat, K = map(_f32, (means, quats, scales, viewmat, K))
    N = means.shape[0]
    radii = np.zeros(N, np.int32)
    means2d = np.zeros((N, 2), np.float32)
    depths = np.zeros(N, np.float32)
    conics = np.zeros((N, 3), np.float32)
    _lib().orc_proj_fwd(C.c_int(N), _p(means), _p(quats), _p(scales), _p(viewmat), _p(K), C.c_int(W), C.c_int(H),
                        C.c_float(eps2d), C.c_float(near), C.c_float(far), C.c_float(radius_clip), _p(radii),
                        _p(means2d), _p(depths), _p(conics))
    return radii, means2d, depths, conics


def quat_to_rotmat(quats):
    """orc_quat_to_rotmat (utils.cuh:14-36): [N,4] (w, x, y, z; normalised inside) -> [N,3,3] row-major"""
    quats = _f32(quats)
    R = np.zeros((quats.shape[0], 3, 3), np.float32)
    _lib().orc_quat_to_rotmat(C.c_int(quats.shape[0]), _p(quats), _p(R))
    return R


def quat_scale_to_covar(quats, scales):
    """orc_quat_scale_to_covar (utils.cuh:64-96): (R S)(R S)^T -> [N,3,3]"""
    quats, scales = _f32(quats), _f32(scales)
    cov = np.zeros((quats.shape[0], 3, 3), np.float32)
    _lib().orc_quat_scale_to_covar(C.c_int(quats.shape[0]), _p(quats), _p(scales), _p(cov))
    return cov


def quat_to_rotmat_vjp(quats, v_R):
    """orc_quat_to_rotmat_vjp (utils.cuh:38-62): dL/dR [N,3,3] row-major -> dL/dq [N,4]"""
    quats, v_R = _f32(quats), _f32(v_R)
    vq = np.zeros((quats.shape[0], 4), np.float32)
    _lib().orc_quat_to_rotmat_vjp(C.c_int(quats.shape[0]), _p(quats), _p(v_R), _p(vq))
    return vq


def proj_bwd(means, quats, scales, viewmat, K, W, H, radii, conics, v_means2d, v_depths, v_conics):
    means, quats, scales, viewmat, K, conics = map(_f32, (means, quats, scales, viewmat, K, conics))
    v_means2d, v_depths, v_conics = map(_f32, (v_means2d, v_depths, v_conics))
    radii = _i32(radii)
    N = means.shape[0]
    v_means = np.zeros((N, 3), np.float32)
    v_quats = np.zeros((N, 4), np.float32)
    v_scales = np.zeros((N, 3), np.float32)
    _lib().orc_proj_bwd(C.c_int(N), _p(means), _p(quats), _p(scales), _p(viewmat), _p(K), C.c_int(W), C.c_int(H),
                        _p(radii), _p(conics), _p(v_means2d), _p(v_depths), _p(v_conics), _p(v_means), _p(v_quats),
                        _p(v_scales))
    return v_means, v_quats, v_scales


def sh_fwd(degree, dirs, coeffs, masks=None):
    dirs, coeffs = _f32(dirs), _f32(coeffs)
    N, K = coeffs.shape[0], coeffs.shape[1]
    m = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint8)
    colors = np.zeros((N, 3), np.float32)
    _lib().orc_sh_fwd(C.c_int(N), C.c_int(K), C.c_int(degree), _p(dirs), _p(coeffs), _p(m), _p(colors))
    return colors


def sh_bwd(degree, dirs, coeffs, masks, v_colors, want_v_dirs=True):
    dirs, coeffs, v_colors = _f32(dirs), _f32(coeffs), _f32(v_colors)
    N, K = coeffs.shape[0], coeffs.shape[1]
    m = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint8)
    v_coeffs = np.zeros((N, K, 3), np.float32)
    v_dirs = np.zeros((N, 3), np.float32) if want_v_dirs else None
    _lib().orc_sh_bwd(C.c_int(N), C.c_int(K), C.c_int(degree), _p(dirs), _p(coeffs), _p(m), _p(v_colors),
                      _p(v_coeffs), _p(v_dirs))
    return v_coeffs, v_dirs


def isect_tiles(means2d, radii, tile_size, tw, th):
    """-> tiles_per_gauss, isect_ids(sorted), flatten_ids(sorted), group_gs_ids, group_starts, offsets[th,tw]"""
    means2d, radii = _f32(means2d), _i32(radii)
    N = radii.shape[0]
    tpg = np.zeros(N, np.int32)
    gpg = np.zeros(N, np.int32)
    ni, ng = C.c_int64(0), C.c_int64(0)
    _lib().orc_isect_count(C.c_int(N), _p(means2d), _p(radii), C.c_int(tile_size), C.c_int(tw), C.c_int(th),
                           _p(tpg), _p(gpg), C.byref(ni), C.byref(ng))
    ni, ng = ni.value, ng.value
    isect_ids = np.zeros(max(ni, 1), np.int64)
    flatten_ids = np.zeros(max(ni, 1), np.int32)
    ggs = np.zeros(max(ng, 1), np.int32)
    gst = np.zeros(max(ng, 1), np.int32)
    offsets = np.zeros((th, tw), np.int32)
    _lib().orc_isect_fill_sort(C.c_int(N), _p(means2d), _p(radii), C.c_int(tile_size), C.c_int(tw), C.c_int(th),
                               _p(gpg), C.c_int64(ni), C.c_int64(ng), _p(isect_ids), _p(flatten_ids), _p(ggs),
                               _p(gst), _p(offsets))
    return tpg, isect_ids[:ni], flatten_ids[:ni], ggs[:ng], gst[:ng], offsets


def raster_ges_fwd(means2d, conics, colors, opacities, ref_depth, W, H, tile_size, offsets, flatten_ids,
                   delta_depth):
    means2d, conics, colors, opacities, ref_depth = map(_f32, (means2d, conics, colors, opacities, ref_depth))
    offsets, flatten_ids = _i32(offsets), _i32(flatten_ids)
    th, tw = offsets.shape
    rc = np.zeros((H, W, 4), np.float32)
    ra = np.zeros((H, W), np.float32)
    last = np.zeros((H, W), np.int32)
    _lib().orc_raster_ges_fwd(C.c_int(W), C.c_int(H), C.c_int(tile_size), C.c_int(tw), C.c_int(th),
                              C.c_int64(flatten_ids.shape[0]), C.c_float(delta_depth), _p(means2d), _p(conics),
                              _p(colors), _p(opacities), _p(ref_depth), _p(offsets), _p(flatten_ids), _p(rc), _p(ra),
                              _p(last))
    return rc, ra, last


def raster_ges_bwd_gs(means2d, conics, colors, opacities, radiis, ref_depth, W, H, group_gs_ids, group_starts,
                      delta_depth, v_render_colors, v_render_alphas):
    means2d, conics, colors, opacities, ref_depth = map(_f32, (means2d, conics, colors, opacities, ref_depth))
    v_render_colors, v_render_alphas = _f32(v_render_colors), _f32(v_render_alphas)
    radiis, group_gs_ids, group_starts = map(_i32, (radiis, group_gs_ids, group_starts))
    N = radiis.shape[0]
    v_m = np.zeros((N, 2), np.float32)
    v_c = np.zeros((N, 3), np.float32)
    v_col = np.zeros((N, 4), np.float32)
    v_o = np.zeros(N, np.float32)
    _lib().orc_raster_ges_bwd_gs(C.c_int(W), C.c_int(H), C.c_int64(group_gs_ids.shape[0]), C.c_float(delta_depth),
                                 _p(group_gs_ids), _p(group_starts), _p(means2d), _p(conics), _p(colors),
                                 _p(opacities), _p(radiis), _p(ref_depth), _p(v_render_colors), _p(v_render_alphas),
                                 _p(v_m), _p(v_c), _p(v_col), _p(v_o))
    return v_m, v_c, v_col, v_o


def raster_ges_fwd_flip_budget(means2d, conics, colors, opacities, ref_depth, W, H, tile_size, offsets, flatten_ids,
                               delta_depth, rel_band=1e-5):
    """Per pixel: what a flip of every borderline accept/reject decision could change -> (budget[H,W,5], n_pairs,
    n_pixels).  See orc_raster_ges_fwd_flip_budget."""
    means2d, conics, colors, opacities, ref_depth = map(_f32, (means2d, conics, colors, opacities, ref_depth))
    offsets, flatten_ids = _i32(offsets), _i32(flatten_ids)
    th, tw = offsets.shape
    budget = np.zeros((H, W, 5), np.float32)
    n = np.zeros(2, np.int64)
    _lib().orc_raster_ges_fwd_flip_budget(C.c_int(W), C.c_int(H), C.c_int(tile_size), C.c_int(tw), C.c_int(th),
                                          C.c_int64(flatten_ids.shape[0]), C.c_float(delta_depth), _p(means2d),
                                          _p(conics), _p(colors), _p(opacities), _p(ref_depth), _p(offsets),
                                          _p(flatten_ids), C.c_float(rel_band), _p(budget), _p(n))
    return budget, int(n[0]), int(n[1])


def raster_ges_bwd_gs_flip_budget(means2d, conics, colors, opacities, radiis, ref_depth, W, H, group_gs_ids,
                                  group_starts, delta_depth, v_render_colors, v_render_alphas, rel_band=1e-5):
    """Per Gaussian: |contribution| of every borderline pixel slot to the 10 gradient entries, ordered
    {v_colors[4], v_conics[3], v_means2d[2], v_opacities} -> (budget[N,10], n_pairs, n_gaussians)."""
    means2d, conics, colors, opacities, ref_depth = map(_f32, (means2d, conics, colors, opacities, ref_depth))
    v_render_colors, v_render_alphas = _f32(v_render_colors), _f32(v_render_alphas)
    radiis, group_gs_ids, group_starts = map(_i32, (radiis, group_gs_ids, group_starts))
    N = radiis.shape[0]
    budget = np.zeros((N, 10), np.float32)
    n = np.zeros(2, np.int64)
    _lib().orc_raster_ges_bwd_gs_flip_budget(C.c_int(W), C.c_int(H), C.c_int(N), C.c_int64(group_gs_ids.shape[0]),
                                             C.c_float(delta_depth), _p(group_gs_ids), _p(group_starts), _p(means2d),
                                             _p(conics), _p(colors), _p(opacities), _p(radiis), _p(ref_depth),
                                             _p(v_render_colors), _p(v_render_alphas), C.c_float(rel_band), _p(budget),
                                             _p(n))
    return budget, int(n[0]), int(n[1])


def raster_ges_bwd_exact(means2d, conics, colors, opacities, ref_depth, W, H, tile_size, offsets, flatten_ids,
                         delta_depth, v_render_colors, v_render_alphas):
    means2d, conics, colors, opacities, ref_depth = map(_f32, (means2d, conics, colors, opacities, ref_depth))
    v_render_colors, v_render_alphas = _f32(v_render_colors), _f32(v_render_alphas)
    offsets, flatten_ids = _i32(offsets), _i32(flatten_ids)
    th, tw = offsets.shape
    N = means2d.shape[0]
    v_m = np.zeros((N, 2), np.float32)
    v_c = np.zeros((N, 3), np.float32)
    v_col = np.zeros((N, 4), np.float32)
    v_o = np.zeros(N, np.float32)
    _lib().orc_raster_ges_bwd_exact(C.c_int(W), C.c_int(H), C.c_int(tile_size), C.c_int(tw), C.c_int(th),
                                    C.c_int64(flatten_ids.shape[0]), C.c_float(delta_depth), _p(means2d), _p(conics),
                                    _p(colors), _p(opacities), _p(ref_depth), _p(offsets), _p(flatten_ids),
                                    _p(v_render_colors), _p(v_render_alphas), _p(v_m), _p(v_c), _p(v_col), _p(v_o))
    return v_m, v_c, v_col, v_o


# ----------------------------------------------------------------------------- `raw` render method
def isect_tiles_depth(means2d, radii, depths, tile_size, tw, th):
    """isectTiles + isectOffsetEncode (depth-keyed): -> tiles_per_gauss, isect_ids(sorted), flatten_ids(sorted), offsets"""
    means2d, radii, depths = _f32(means2d), _i32(radii), _f32(depths)
    N = radii.shape[0]
    tpg = np.zeros(N, np.int32)
    gpg = np.zeros(N, np.int32)
    ni, ng = C.c_int64(0), C.c_int64(0)
    _lib().orc_isect_count(C.c_int(N), _p(means2d), _p(radii), C.c_int(tile_size), C.c_int(tw), C.c_int(th), _p(tpg),
                           _p(gpg), C.byref(ni), C.byref(ng))
    ni = ni.value
    isect_ids = np.zeros(max(ni, 1), np.int64)
    flatten_ids = np.zeros(max(ni, 1), np.int32)
    offsets = np.zeros((th, tw), np.int32)
    _lib().orc_isect_tiles_depth(C.c_int(N), _p(means2d), _p(radii), _p(depths), C.c_int(tile_size), C.c_int(tw),
                                 C.c_int(th), C.c_int64(ni), _p(isect_ids), _p(flatten_ids), _p(offsets))
    return tpg, isect_ids[:ni], flatten_ids[:ni], offsets


def raster_raw_fwd(means2d, conics, colors, opacities, W, H, tile_size, offsets, flatten_ids, backgrounds=None):
    """-> render_colors[H,W,4], render_alphas[H,W], last_ids[H,W]"""
    means2d, conics, colors, opacities = _f32(means2d), _f32(conics), _f32(colors), _f32(opacities)
    offsets, flatten_ids = _i32(offsets), _i32(flatten_ids)
    th, tw = offsets.shape
    bg = None if backgrounds is None else _f32(backgrounds)
    rc = np.zeros((H, W, 4), np.float32)
    ra = np.zeros((H, W), np.float32)
    last = np.zeros((H, W), np.int32)
    _lib().orc_raster_raw_fwd(C.c_int(W), C.c_int(H), C.c_int(tile_size), C.c_int(tw), C.c_int(th),
                              C.c_int64(flatten_ids.shape[0]), _p(means2d), _p(conics), _p(colors), _p(opacities),
                              _p(bg) if bg is not None else None, _p(offsets), _p(flatten_ids), _p(rc), _p(ra), _p(last))
    return rc, ra, last


def raster_raw_bwd(means2d, conics, colors, opacities, W, H, tile_size, offsets, flatten_ids, render_alphas, last_ids,
                   v_render_colors, v_render_alphas, backgrounds=None, absgrad=False):
    """-> v_means2d[N,2], v_conics[N,3], v_colors[N,4], v_opacities[N] (+ v_means2d_abs if absgrad)"""
    means2d, conics, colors, opacities = _f32(means2d), _f32(conics), _f32(colors), _f32(opacities)
    offsets, flatten_ids = _i32(offsets), _i32(flatten_ids)
    th, tw = offsets.shape
    N = opacities.shape[0]
    bg = None if backgrounds is None else _f32(backgrounds)
    ra, last = _f32(render_alphas), _i32(last_ids)
    v_rc, v_ra = _f32(v_render_colors), _f32(v_render_alphas)
    v_m, v_c, v_col, v_o = (np.zeros((N, 2), np.float32), np.zeros((N, 3), np.float32), np.zeros((N, 4), np.float32),
                            np.zeros(N, np.float32))
    v_abs = np.zeros((N, 2), np.float32) if absgrad else None
    _lib().orc_raster_raw_bwd(C.c_int(W), C.c_int(H), C.c_int(tile_size), C.c_int(tw), C.c_int(th),
                              C.c_int64(flatten_ids.shape[0]), _p(means2d), _p(conics), _p(colors), _p(opacities),
                              _p(bg) if bg is not None else None, _p(offsets), _p(flatten_ids), _p(ra), _p(last), _p(v_rc),
                              _p(v_ra), _p(v_abs) if absgrad else None, _p(v_m), _p(v_c), _p(v_col), _p(v_o))
    return (v_m, v_c, v_col, v_o, v_abs) if absgrad else (v_m, v_c, v_col, v_o)


# ----------------------------------------------------------------------------- fused SSIM (ssim.cu)
def ssim_fwd(img1, img2, C1=0.01 ** 2, C2=0.03 ** 2, train=True):
    """img [B,CH,H,W] -> ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 (None x3 if not train)"""
    img1, img2 = _f32(img1), _f32(img2)
    B, CH, H, W = img1.shape
    m = np.zeros_like(img1)
    d = [np.zeros_like(img1) for _ in range(3)] if train else [None] * 3
    _lib().orc_ssim_fwd(C.c_int(B), C.c_int(CH), C.c_int(H), C.c_int(W), C.c_float(C1), C.c_float(C2), _p(img1), _p(img2), _p(m),
                        *[_p(x) if x is not None else None for x in d])
    return (m, *d)


def ssim_bwd(img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
    img1, img2, dL_dmap = _f32(img1), _f32(img2), _f32(dL_dmap)
    B, CH, H, W = img1.shape
    g = np.zeros_like(img1)
    _lib().orc_ssim_bwd(C.c_int(B), C.c_int(CH), C.c_int(H), C.c_int(W), _p(img1), _p(img2), _p(dL_dmap), _p(_f32(dm_dmu1)),
                        _p(_f32(dm_dsigma1_sq)), _p(_f32(dm_dsigma12)), _p(g))
    return g


# ----------------------------------------------------------------------------- gesForward as a whole
def ges_render(means, log_scales, quats, sh_dc, sh_rest, opac_logit, c2w, K, W, H, ref_depth, base_color, delta_depth=0.1,
               sh_degree=3, max_gs_radii=100, tile_size=16):
    """RawGaussianModel::gesForward under NoGradGuard (src/raw_gs_model.cpp:188-367) with the oracle operators:
    exp / projection / radii clamp / SH / clamp_min / sigmoid -> binning -> ges rasterizer -> compose with the TSDF layer.
    ref_depth [H,W] is the RAW raycast depth (0 = miss; clamped to 1000 below 0.01 as raw_gs_model.cpp:205-207),
    base_color [H,W,3].  -> rgb [H,W,3] float32, weight_sum [H,W]."""
    f = lambda a: np.ascontiguousarray(a, np.float32)
    means, log_scales, quats, sh_dc, sh_rest, opac_logit = map(f, (means, log_scales, quats, sh_dc, sh_rest, opac_logit))
    c2w = np.asarray(c2w, np.float32)
    R, t = c2w[:3, :3], c2w[:3, 3]
    vm = np.eye(4, dtype=np.float32)
    vm[:3, :3] = R.T
    vm[:3, 3] = -R.T @ t
    radii, m2, depths, conics = proj_fwd(means, quats, np.exp(log_scales), vm, np.asarray(K, np.float32), W, H)
    radii = np.minimum(radii, max_gs_radii)
    sh = np.concatenate([sh_dc[:, None], sh_rest], 1)
    rgb = np.maximum(sh_fwd(sh_degree, means - t[None], sh, radii > 0) + np.float32(0.5), np.float32(0.0))
    colors = np.concatenate([rgb, depths[:, None]], 1).astype(np.float32)
    opac = (np.float32(1.0) / (np.float32(1.0) + np.exp(-opac_logit.reshape(-1)))).astype(np.float32)
    tw, th = (W + tile_size - 1) // tile_size, (H + tile_size - 1) // tile_size
    _, _, flat, _, _, offs = isect_tiles(m2, radii, tile_size, tw, th)
    ref = f(ref_depth).reshape(H, W)
    ref_c = np.where(ref < 0.01, np.float32(1000.0), ref).astype(np.float32)
    rc, ws, _ = raster_ges_fwd(m2, conics, colors, opac, ref_c, W, H, tile_size, offs, flat, delta_depth)
    out = (rc[..., :3] + f(base_color)) / (ws[..., None] + np.float32(1.0))
    return out.astype(np.float32), ws

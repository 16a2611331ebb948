"""GPU parity: every splat C-ABI entry point against the CPU oracle on identical seeded inputs.

Tolerances (fp32 work; the reference itself uses fast-math intrinsics and float atomics):
  projection / SH values        rtol 1e-4
  rasterizer forward            rtol 2e-4, atol 2e-4   (__expf vs expf, summation order)
  gradients                     rtol 2e-3, atol 2e-4 * max|ref|
Integer outputs of the binning (tile counts, sorted ids, group table, offsets) are bit-exact.
"""
import math

import numpy as np
import pytest
import torch

from tests import scenes

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def T(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a)).to(_dev())
    return t if dtype is None else t.to(dtype)


def N_(t):
    return t.detach().cpu().numpy()


def _setup(N, W, H, seed, scale_range=(0.003, 0.05)):
    g = scenes.random_gaussians(N, seed=seed, scale_range=scale_range)
    c2w, K = scenes.default_camera(W, H, seed=seed)
    return g, scenes.pose_inv(c2w), K, c2w


def _grad_close(got, ref, name, rtol=2e-3, atol_rel=2e-4):
    scale = max(1e-12, float(np.abs(ref).max()))
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol_rel * scale, err_msg=name)


@pytest.mark.parametrize("N,W,H", [(5000, 96, 64), (100000, 640, 480)])
def test_projection_fwd_bwd(N, W, H):
    from gps_slam_amd import gsplat_ops as ops
    from oracle import splat_ref as orc
    g, vm, K, _ = _setup(N, W, H, seed=N)
    scales = np.exp(g["log_scales"])
    r0, m0, d0, c0 = orc.proj_fwd(g["means"], g["quats"], scales, vm, K, W, H)
    r1, m1, d1, c1 = ops.fully_fused_projection_fwd(T(g["means"]), T(g["quats"]), T(scales), T(vm)[None], T(K)[None],
                                                    W, H)
    r1, m1, d1, c1 = N_(r1)[0], N_(m1)[0], N_(d1)[0], N_(c1)[0]
    both = (r0 > 0) & (r1 > 0)
    assert both.sum() > 0.3 * N
    # a radius may differ (by one) only where ceil(3 sqrt(lambda)) is decided within rounding of an integer: listed from the
    # conics, every difference must be on the list; cull decisions (on-screen tests use the radius) likewise
    edge = scenes.radius_is_borderline(c0) | scenes.radius_is_borderline(c1)
    assert (np.abs(r0 - r1)[both] <= 1).all() and not ((r0 != r1) & both & ~edge).any(), int(((r0 != r1) & both & ~edge).sum())
    assert ((r0 > 0) != (r1 > 0)).sum() <= 1e-4 * N + 2
    np.testing.assert_allclose(m1[both], m0[both], rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(d1[both], d0[both], rtol=1e-5)
    # conics: every row within its own condition budget (the 2x2 inverse amplifies rounding by the covariance's condition number)
    cb_ = scenes.condition_budget(lambda mm, qq, ss: (orc.proj_fwd(mm, qq, ss, vm, K, W, H)[3],), (g["means"], g["quats"], scales), (c0,))[0]
    cerr = np.abs(c1.astype(np.float64) - c0).max(1)
    print("conics: max error / budget %.3f (budget / |row| max %.1e)" % ((cerr[both] / cb_[both]).max(), (cb_[both] / np.abs(c0[both]).max(1)).max()))
    assert (cerr[both] <= cb_[both]).all()
    # backward on the oracle's forward state
    rng = np.random.default_rng(1)
    v_m2 = rng.normal(size=(N, 2)).astype(np.float32)
    v_d = rng.normal(size=N).astype(np.float32)
    v_c = (rng.normal(size=(N, 3)) * 0.1).astype(np.float32)
    e = orc.proj_bwd(g["means"], g["quats"], scales, vm, K, W, H, r0, c0, v_m2, v_d, v_c)
    o = ops.fully_fused_projection_bwd(T(g["means"]), T(g["quats"]), T(scales), T(vm)[None], T(K)[None], W, H, 0.3,
                                       T(r0)[None], T(c0)[None], T(v_m2)[None], T(v_d)[None], T(v_c)[None])
    vis = r0 > 0
    # EVERY row within its own condition budget (tests/scenes.py: the oracle's sensitivity to 1-ulp jitter of its float inputs --
    # the adjoint inverts the 2x2 conic and its terms cancel, so rows differ by orders of magnitude in what rounding can do)
    fn = lambda mm, qq, ss, cc, a, b, c: orc.proj_bwd(mm, qq, ss, vm, K, W, H, r0, cc, a, b, c)
    budget = scenes.condition_budget(fn, (g["means"], g["quats"], scales, c0, v_m2, v_d, v_c), e)
    for got, ref, bud, name in zip(o, e, budget, ("v_means", "v_quats", "v_scales")):
        got = N_(got)
        assert (got[~vis] == 0).all()
        err = np.abs(got.astype(np.float64) - ref).reshape(N, -1).max(1)
        ratio = err[vis] / (bud[vis] + 1e-300)
        print("%s: max error / budget %.3f (budget / |row| median %.1e, max %.1e)" % (
            name, ratio.max(), np.median(bud[vis] / (np.abs(ref[vis]).max(1) + 1e-300)), (bud[vis] / (np.abs(ref[vis]).max(1) + 1e-300)).max()))
        assert (err[vis] <= bud[vis]).all(), (name, int((err[vis] > bud[vis]).sum()), float(ratio.max()))


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_sh_fwd_bwd(deg):
    from gps_slam_amd import gsplat_ops as ops
    from oracle import splat_ref as orc
    rng = np.random.default_rng(deg)
    N, K = 20000, 25 if deg == 4 else 16
    dirs = (rng.normal(size=(N, 3)) * 2).astype(np.float32)
    coeffs = rng.normal(size=(N, K, 3)).astype(np.float32)
    masks = rng.uniform(size=N) > 0.2
    e = orc.sh_fwd(deg, dirs, coeffs, masks)
    o = N_(ops.compute_sh_fwd(deg, T(dirs)[None], T(coeffs)[None], T(masks)[None]))[0]
    np.testing.assert_allclose(o, e, rtol=1e-4, atol=1e-5)
    v_col = rng.normal(size=(N, 3)).astype(np.float32)
    ec, ed = orc.sh_bwd(deg, dirs, coeffs, masks, v_col)
    oc, od = ops.compute_sh_bwd(K, deg, T(dirs)[None], T(coeffs)[None], T(masks)[None], T(v_col)[None])
    np.testing.assert_allclose(N_(oc)[0], ec, rtol=1e-4, atol=1e-5)
    _grad_close(N_(od)[0], ed, "v_dirs", rtol=2e-3, atol_rel=1e-5)


def _bin_inputs(N, W, H, seed, rmax=40):
    rng = np.random.default_rng(seed)
    m2 = np.stack([rng.uniform(-20, W + 20, N), rng.uniform(-20, H + 20, N)], 1).astype(np.float32)
    radii = np.minimum(rng.geometric(0.15, N), rmax).astype(np.int32)
    radii[rng.uniform(size=N) < 0.3] = 0
    radii[:3] = [100, 0, 1]
    return m2, radii


# 96x64: 24 tiles; 640x480: 1,200; 1280x720: 3,600 (the one-pass counting sort on the whole tile id, <= 4,096 tiles);
# 2048x1200: 9,600 tiles (two LSD passes + offsets pass)
# 1200x680: Replica's image size (75 x 43 = 3,225 tiles, ragged last tile row)
@pytest.mark.parametrize("N,W,H", [(0, 96, 64), (1, 96, 64), (3000, 96, 64), (100000, 640, 480), (60000, 1280, 720),
                                   (777, 50, 37), (50000, 2048, 1200), (80000, 1200, 680)])
def test_binning_bit_exact(N, W, H):
    from gps_slam_amd import gsplat_ops as ops
    from oracle import splat_ref as orc
    TS = 16
    tw, th = (W + TS - 1) // TS, (H + TS - 1) // TS
    m2, radii = _bin_inputs(max(N, 3), W, H, seed=N + W)
    m2, radii = m2[:N], radii[:N]
    tpg, ids, flat, ggs, gst, offs = orc.isect_tiles(m2, radii, TS, tw, th)
    r = ops.isect_tiles_no_depth(T(m2).reshape(1, N, 2), T(radii).reshape(1, N), TS, tw, th, want_isect_ids=True)
    ni, ng = r.sizes()
    assert ni == ids.shape[0] and ng == ggs.shape[0]
    g_tpg, g_ids, g_flat, g_ggs, g_gst, g_offs = r.trimmed()
    np.testing.assert_array_equal(N_(g_tpg)[0], tpg)
    np.testing.assert_array_equal(N_(g_ids), ids)
    np.testing.assert_array_equal(N_(g_flat), flat)
    np.testing.assert_array_equal(N_(g_ggs), ggs)
    np.testing.assert_array_equal(N_(g_gst), gst)
    np.testing.assert_array_equal(N_(g_offs)[0], offs)
    assert int(N_(r.counts)[3]) == int((radii > 0).sum())


@pytest.mark.parametrize("N,W,H", [(100000, 640, 480), (0, 96, 64), (50000, 2048, 1200)])
def test_binning_without_isect_ids_writes_the_same_lists_and_offsets(N, W, H):
    """The model path asks for no int64 isect_ids: the tile offsets then come out of the scatter (exclusive scan of the tile
    totals) instead of a pass over the sorted keys."""
    from gps_slam_amd import gsplat_ops as ops
    from oracle import splat_ref as orc
    TS = 16
    tw, th = (W + TS - 1) // TS, (H + TS - 1) // TS
    m2, radii = _bin_inputs(max(N, 3), W, H, seed=N + W + 1)
    m2, radii = m2[:N], radii[:N]
    tpg, ids, flat, ggs, gst, offs = orc.isect_tiles(m2, radii, TS, tw, th)
    r = ops.isect_tiles_no_depth(T(m2).reshape(1, N, 2), T(radii).reshape(1, N), TS, tw, th, want_isect_ids=False)
    ni, ng = r.sizes()
    assert ni == flat.shape[0] and ng == ggs.shape[0]
    np.testing.assert_array_equal(N_(r.flatten_ids)[:ni], flat)
    np.testing.assert_array_equal(N_(r.isect_offsets).reshape(-1), offs.reshape(-1))
    np.testing.assert_array_equal(N_(r.group_gs_ids)[:ng], ggs)


def test_binning_capacity_overflow_is_reported():
    from gps_slam_amd import gsplat_ops as ops
    m2, radii = _bin_inputs(5000, 640, 480, seed=5)
    r = ops.isect_tiles_no_depth(T(m2).reshape(1, -1, 2), T(radii).reshape(1, -1), 16, 40, 30, isect_capacity=1000,
                                 group_capacity=1000)
    with pytest.raises(RuntimeError):
        r.sizes()


def _raster_state(N, W, H, seed):
    """Oracle-projected scene -> everything the rasterizer consumes."""
    from oracle import splat_ref as orc
    g, vm, K, c2w = _setup(N, W, H, seed)
    scales = np.exp(g["log_scales"])
    radii, m2, depths, conics = orc.proj_fwd(g["means"], g["quats"], scales, vm, K, W, H)
    radii = np.minimum(radii, 100)
    dirs = g["means"] - c2w[:3, 3][None]
    rgb = np.maximum(orc.sh_fwd(3, dirs, g["sh"], radii > 0) + 0.5, 0.0)
    colors = np.concatenate([rgb, depths[:, None]], 1).astype(np.float32)
    opac = (1.0 / (1.0 + np.exp(-g["opac_logit"][:, 0]))).astype(np.float32)
    rng = np.random.default_rng(seed)
    ref_depth = rng.uniform(1.5, 4.5, (H, W)).astype(np.float32)
    ref_depth[rng.uniform(size=(H, W)) < 0.1] = 1000.0
    return radii, m2, depths, conics, colors, opac, ref_depth


@pytest.mark.parametrize("N,W,H", [(3000, 96, 64), (100000, 640, 480), (4000, 50, 37)])
def test_raster_ges_fwd_bwd(N, W, H):
    from gps_slam_amd import gsplat_ops as ops
    from oracle import splat_ref as orc
    TS, delta = 16, 0.1
    tw, th = (W + TS - 1) // TS, (H + TS - 1) // TS
    radii, m2, depths, conics, colors, opac, ref_depth = _raster_state(N, W, H, seed=N)
    tpg, ids, flat, ggs, gst, offs = orc.isect_tiles(m2, radii, TS, tw, th)
    e_rc, e_ra, e_last = orc.raster_ges_fwd(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, delta)
    isect = ops.isect_tiles_no_depth(T(m2)[None], T(radii)[None], TS, tw, th)
    tm2, tcon, tcol, top, tref = T(m2)[None], T(conics)[None], T(colors)[None], T(opac)[:, None], T(ref_depth)[None, ..., None]
    rc, ra, last = ops.rasterize_to_pixels_fwd_ges(tm2, tcon, tcol, top, tref, W, H, TS, isect, delta,
                                                   want_last_ids=True)
    # A (pixel, Gaussian) pair whose o * exp(-sigma) sits within rounding of 1/255 (or whose depth sits on the cut) can be
    # decided differently by __expf and expf.  No outlier budget: the oracle lists those borderline pairs and what each could
    # contribute (orc_raster_ges_*_flip_budget); every element must satisfy
    #     |hip - oracle| <= 2e-5 * sum|terms| + 2 ulp * sum|terms| weighted by sum|terms of sigma|
    #                       + contribution of its borderline pairs,
    # and the elements that needed the last term are counted against the number of borderline pairs.  The middle term matters
    # for this synthetic state only (radii are not tied to the conics: narrow Gaussians are hit 10+ sigma-terms from their
    # centre, where sigma's own float rounding is a relative error of exp(-sigma); measured: < 1 ulp of sigma's largest term).
    REL, BAND, SIG = 2e-5, 1e-5, 2 * 2.0 ** -23
    scale_f, _, _ = orc.raster_ges_fwd_flip_budget(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, delta, rel_band=-1.0)
    sig_f, _, _ = orc.raster_ges_fwd_flip_budget(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, delta, rel_band=-2.0)
    scale_f = scale_f + (SIG / REL) * sig_f
    flip_f, n_pairs, n_pix = orc.raster_ges_fwd_flip_budget(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, delta, rel_band=BAND)
    got = np.concatenate([N_(rc)[0], N_(ra)[0]], -1)
    exp = np.concatenate([e_rc, e_ra[..., None]], -1)
    d = np.abs(got - exp)
    assert (d <= REL * scale_f + 1e-7 + 1.001 * flip_f).all()
    flipped_px = int((d > REL * scale_f + 1e-7).any(-1).sum())
    assert flipped_px <= n_pix
    print("fwd: %d borderline pairs on %d pixels, %d pixels flipped" % (n_pairs, n_pix, flipped_px))
    assert (N_(last)[0] == e_last).mean() > 0.999
    assert e_ra.max() > 1.0  # the scene actually covers pixels
    rng = np.random.default_rng(3)
    v_rc = rng.normal(size=(H, W, 4)).astype(np.float32)
    v_ra = rng.normal(size=(H, W)).astype(np.float32)
    e = orc.raster_ges_bwd_gs(m2, conics, colors, opac, radii, ref_depth, W, H, ggs, gst, delta, v_rc, v_ra)
    o = ops.rasterize_to_pixels_bwd_ges_gs_parallel(tm2, tcon, tcol, top, T(radii)[None], tref, W, H, isect, delta,
                                                    T(v_rc)[None], T(v_ra)[None, ..., None])
    scale_b, _, _ = orc.raster_ges_bwd_gs_flip_budget(m2, conics, colors, opac, radii, ref_depth, W, H, ggs, gst, delta, v_rc, v_ra,
                                                      rel_band=-1.0)
    sig_b, _, _ = orc.raster_ges_bwd_gs_flip_budget(m2, conics, colors, opac, radii, ref_depth, W, H, ggs, gst, delta, v_rc, v_ra,
                                                    rel_band=-2.0)
    scale_b = scale_b + (SIG / REL) * sig_b
    flip_b, nb_pairs, nb_g = orc.raster_ges_bwd_gs_flip_budget(m2, conics, colors, opac, radii, ref_depth, W, H, ggs, gst, delta, v_rc,
                                                               v_ra, rel_band=BAND)
    Ng = m2.shape[0]
    got_b = np.concatenate([N_(o[2]).reshape(Ng, 4), N_(o[1]).reshape(Ng, 3), N_(o[0]).reshape(Ng, 2), N_(o[3]).reshape(Ng, 1)], 1)
    exp_b = np.concatenate([e[2], e[1], e[0], e[3][:, None]], 1)
    db = np.abs(got_b - exp_b)
    assert (db <= REL * scale_b + 1e-30 + 1.001 * flip_b).all()
    flipped_g = int((db > REL * scale_b + 1e-30).any(-1).sum())
    assert flipped_g <= nb_g
    print("bwd: %d borderline slots on %d Gaussians, %d Gaussians flipped" % (nb_pairs, nb_g, flipped_g))
    # the same operator through the kernel the fused train step runs (column strips; no group table): same oracle, same budgets
    s_ = ops.rasterize_to_pixels_bwd_ges_strips(tm2, tcon, tcol, top, T(radii)[None], tref, W, H, delta, T(v_rc)[None], T(v_ra)[None, ..., None])
    got_s = np.concatenate([N_(s_[2]).reshape(Ng, 4), N_(s_[1]).reshape(Ng, 3), N_(s_[0]).reshape(Ng, 2), N_(s_[3]).reshape(Ng, 1)], 1)
    ds = np.abs(got_s - exp_b)
    assert (ds <= REL * scale_b + 1e-30 + 1.001 * flip_b).all(), (int((ds > REL * scale_b + 1e-30 + 1.001 * flip_b).sum()), float(ds.max()))
    flipped_s = int((ds > REL * scale_b + 1e-30).any(-1).sum())
    assert flipped_s <= nb_g
    s2 = ops.rasterize_to_pixels_bwd_ges_strips(tm2, tcon, tcol, top, T(radii)[None], tref, W, H, delta, T(v_rc)[None], T(v_ra)[None, ..., None])
    assert all(torch.equal(a, b) for a, b in zip(s_, s2)), "the strip backward has no atomics: bit-reproducible"
    # the residency reserve (gps_set_frame_chain_reserve: unused dynamic LDS, 5 instead of 6 workgroups per compute unit) is a
    # scheduling hint: the same bits with it on
    from gps_slam_amd import _lib
    _lib.load_library().gps_set_frame_chain_reserve(1)
    try:
        s3 = ops.rasterize_to_pixels_bwd_ges_strips(tm2, tcon, tcol, top, T(radii)[None], tref, W, H, delta, T(v_rc)[None], T(v_ra)[None, ..., None])
    finally:
        _lib.load_library().gps_set_frame_chain_reserve(0)
    assert all(torch.equal(a, b) for a, b in zip(s_, s3)), "results must not depend on the residency reserve"
    print("bwd (strips): %d Gaussians flipped" % flipped_s)
    # determinism of the sorted order: two runs give bit-identical forward output
    rc2, ra2, _ = ops.rasterize_to_pixels_fwd_ges(tm2, tcon, tcol, top, tref, W, H, TS, isect, delta)
    assert torch.equal(rc, rc2) and torch.equal(ra, ra2)


def test_box_backward_equals_exact_adjoint_where_the_box_holds_the_footprint():
    """The shipped backward visits the 2r x 2r box of a Gaussian (rasterize_to_pixels_bwd_ges_new_parallel.cu:83-96); the
    reference's unused exact adjoint (rasterize_to_pixels_bwd_ges.cu:164-291) visits every pixel of every tile the Gaussian is
    binned to.  They must agree for every Gaussian whose {alpha >= 1/255} footprint -- the ellipse
    0.5 (a dx^2 + c dy^2) + b dx dy <= ln(255 o) -- lies inside its box: checked at 640x480 on the HIP outputs of both box
    kernels (32-pixel groups; column strips) against the HIP exact adjoint, with a realistic radius (ceil of 3 sigma of the major
    axis, so that most footprints DO fit) and on the Gaussians where it does not fit the box version must be the smaller sum."""
    from gps_slam_amd import gsplat_ops as ops
    from oracle import splat_ref as orc
    N, W, H, TS, delta = 100000, 640, 480, 16, 0.1
    tw, th = W // TS, H // TS
    radii, m2, depths, conics, colors, opac, ref_depth = _raster_state(N, W, H, seed=11)
    tm2, tcon, tcol, top, tref = T(m2)[None], T(conics)[None], T(colors)[None], T(opac)[:, None], T(ref_depth)[None, ..., None]
    isect = ops.isect_tiles_no_depth(tm2, T(radii)[None], TS, tw, th)
    rng = np.random.default_rng(5)
    v_rc, v_ra = rng.normal(size=(H, W, 4)).astype(np.float32), rng.normal(size=(H, W)).astype(np.float32)
    tv_rc, tv_ra = T(v_rc)[None], T(v_ra)[None, ..., None]
    ex = ops.rasterize_to_pixels_bwd_ges_exact(tm2, tcon, tcol, top, tref, W, H, TS, isect, delta, tv_rc, tv_ra)
    gr = ops.rasterize_to_pixels_bwd_ges_gs_parallel(tm2, tcon, tcol, top, T(radii)[None], tref, W, H, isect, delta, tv_rc, tv_ra)
    st = ops.rasterize_to_pixels_bwd_ges_strips(tm2, tcon, tcol, top, T(radii)[None], tref, W, H, delta, tv_rc, tv_ra)
    cat = lambda o: np.concatenate([N_(o[2]).reshape(N, 4), N_(o[1]).reshape(N, 3), N_(o[0]).reshape(N, 2), N_(o[3]).reshape(N, 1)], 1)
    g_ex, g_gr, g_st = cat(ex), cat(gr), cat(st)
    # footprint inside the box?  half extents of the ellipse (+ 1 % and a pixel of slack), box = columns int(x) - r + 1 .. int(x) + r
    a, b, c = conics[:, 0].astype(np.float64), conics[:, 1].astype(np.float64), conics[:, 2].astype(np.float64)
    tau = np.log(np.maximum(255.0 * opac.astype(np.float64), 1e-30))
    det = a * c - b * b
    vis = (radii > 0) & (tau > 0) & (det > 0)
    ex_ = np.sqrt(np.where(vis, 2 * tau * c / det, 0.0)) * 1.01 + 1.0
    ey_ = np.sqrt(np.where(vis, 2 * tau * a / det, 0.0)) * 1.01 + 1.0
    x0 = np.trunc(m2[:, 0]).astype(np.int64) - radii + 1
    y0 = np.trunc(m2[:, 1]).astype(np.int64) - radii + 1
    # pixel centres j + 0.5 in [x - ex, x + ex] must all be box columns (or outside the image, where neither version looks)
    lo_ok = (np.maximum(m2[:, 0] - ex_ - 0.5, 0) >= x0) & (np.maximum(m2[:, 1] - ey_ - 0.5, 0) >= y0)
    hi_ok = (np.minimum(m2[:, 0] + ex_ - 0.5, W - 1) <= x0 + 2 * radii - 1) & (np.minimum(m2[:, 1] + ey_ - 0.5, H - 1) <= y0 + 2 * radii - 1)
    inside = vis & lo_ok & hi_ok
    assert inside.sum() > 10000, (int(inside.sum()), int((radii > 0).sum()))  # (opacity 0.5: the 1/255 contour lies at 3.1 sigma, the box ends at ceil(3 sigma))
    # budgets as in test_raster_ges_fwd_bwd: rounding relative to sum |terms| of the element + the oracle's borderline pairs
    tpg, ids, flat, ggs, gst, offs = orc.isect_tiles(m2, radii, TS, tw, th)
    REL, BAND, SIG = 2e-5, 1e-5, 2 * 2.0 ** -23
    scale_b, _, _ = orc.raster_ges_bwd_gs_flip_budget(m2, conics, colors, opac, radii, ref_depth, W, H, ggs, gst, delta, v_rc, v_ra, rel_band=-1.0)
    sig_b, _, _ = orc.raster_ges_bwd_gs_flip_budget(m2, conics, colors, opac, radii, ref_depth, W, H, ggs, gst, delta, v_rc, v_ra, rel_band=-2.0)
    flip_b, _, nb_g = orc.raster_ges_bwd_gs_flip_budget(m2, conics, colors, opac, radii, ref_depth, W, H, ggs, gst, delta, v_rc, v_ra, rel_band=BAND)
    tol = REL * (scale_b + (SIG / REL) * sig_b) + 1e-30 + 1.001 * flip_b
    for name, g_box in (("groups", g_gr), ("strips", g_st)):
        d = np.abs(g_box - g_ex)
        bad = (d > 2 * tol)[inside]
        assert not bad.any(), (name, int(bad.sum()), float(d[inside].max()))
    # and where the footprint sticks out of the box the two versions really differ (the test is not vacuous)
    outside = vis & ~inside
    rel = np.abs(g_gr - g_ex)[outside].max(1) / (np.abs(g_ex)[outside].max(1) + 1e-30)
    assert outside.sum() > 100 and (rel > 1e-3).mean() > 0.05, (int(outside.sum()), float((rel > 1e-3).mean()))
    print("%d of %d visible Gaussians have their footprint inside the box; box == exact adjoint on all of them" % (int(inside.sum()), int((radii > 0).sum())))


@pytest.mark.parametrize("N,W,H,with_bg", [(3000, 96, 64, True), (100000, 640, 480, False), (4000, 50, 37, True)])
def test_raw_binning_and_raster_fwd_bwd(N, W, H, with_bg):
    """`raw` render method: depth-keyed binning bit-exact, front-to-back compositing forward/backward vs the oracle."""
    from gps_slam_amd import gsplat_ops as ops
    from oracle import splat_ref as orc
    TS = 16
    tw, th = (W + TS - 1) // TS, (H + TS - 1) // TS
    radii, m2, depths, conics, colors, opac, _ = _raster_state(N, W, H, seed=N + 1)
    if N == 3000:  # equal depths inside a tile: ties must come out in Gaussian-index order (stable sort)
        depths[::7] = depths[0]
        colors[:, 3] = depths
    bg = np.array([0.2, 0.4, 0.1, 0.0], np.float32) if with_bg else None
    e_tpg, e_ids, e_flat, e_offs = orc.isect_tiles_depth(m2, radii, depths, TS, tw, th)
    isect = ops.isect_tiles(T(m2)[None], T(radii)[None], T(depths)[None], TS, tw, th)
    ni, _ = isect.sizes()
    assert ni == e_flat.shape[0]
    assert np.array_equal(N_(isect.tiles_per_gauss)[0], e_tpg)
    assert np.array_equal(N_(isect.isect_ids)[:ni], e_ids)
    assert np.array_equal(N_(isect.flatten_ids)[:ni], e_flat)
    assert np.array_equal(N_(isect.isect_offsets)[0], e_offs)
    e_rc, e_ra, e_last = orc.raster_raw_fwd(m2, conics, colors, opac, W, H, TS, e_offs, e_flat, backgrounds=bg)
    tm2, tcon, tcol, top = T(m2)[None], T(conics)[None], T(colors)[None], T(opac)[None]
    tbg = None if bg is None else T(bg)[None]
    rc, ra, last = ops.rasterize_to_pixels_fwd(tm2, tcon, tcol, top, tbg, W, H, TS, isect)
    # threshold decisions (alpha >= 1/255, T(1-alpha) <= 1e-4) can flip between expf and __expf for isolated pairs
    for got, ref in ((N_(rc)[0], e_rc), (N_(ra)[0, ..., 0], e_ra)):
        bad = np.abs(got - ref) > (2e-4 * np.abs(ref) + 2e-4)
        assert bad.mean() <= 2e-5, bad.mean()
        assert np.abs(got - ref).max() < 0.05
    assert (N_(last)[0] == e_last).mean() > 0.999
    assert e_ra.max() > 0.9
    # backward on the oracle's forward state (so both walk exactly the same lists)
    rng = np.random.default_rng(5)
    v_rc = rng.normal(size=(H, W, 4)).astype(np.float32)
    v_ra = rng.normal(size=(H, W)).astype(np.float32)
    e = orc.raster_raw_bwd(m2, conics, colors, opac, W, H, TS, e_offs, e_flat, e_ra, e_last, v_rc, v_ra, backgrounds=bg,
                           absgrad=True)
    o = ops.rasterize_to_pixels_bwd(tm2, tcon, tcol, top, tbg, W, H, TS, isect, T(e_ra)[None, ..., None], T(e_last)[None],
                                    T(v_rc)[None], T(v_ra)[None, ..., None], absgrad=True)
    got = dict(zip(("v_abs", "v_means2d", "v_conics", "v_colors", "v_opacities"), o))
    ref = dict(zip(("v_means2d", "v_conics", "v_colors", "v_opacities", "v_abs"), e))
    for name in ref:
        g_, r_ = N_(got[name]).reshape(ref[name].shape), ref[name]
        scale = np.abs(r_).max()
        assert scale > 0
        bad = np.abs(g_ - r_) > (2e-3 * np.abs(r_) + 5e-4 * scale)
        assert bad.mean() < 1e-4, (name, bad.mean())
    # without absgrad the other four gradients are the same numbers up to atomic order
    o2 = ops.rasterize_to_pixels_bwd(tm2, tcon, tcol, top, tbg, W, H, TS, isect, T(e_ra)[None, ..., None], T(e_last)[None],
                                     T(v_rc)[None], T(v_ra)[None, ..., None])
    assert o2[0] is None
    for a, b in zip(o[1:], o2[1:]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()))
    # the forward is deterministic
    rc2, ra2, last2 = ops.rasterize_to_pixels_fwd(tm2, tcon, tcol, top, tbg, W, H, TS, isect)
    assert torch.equal(rc, rc2) and torch.equal(ra, ra2) and torch.equal(last, last2)


def test_raw_raster_empty_scene():
    from gps_slam_amd import gsplat_ops as ops
    W, H, TS = 64, 48, 16
    m2 = torch.zeros((1, 4, 2), device=_dev())
    radii = torch.zeros((1, 4), dtype=torch.int32, device=_dev())
    isect = ops.isect_tiles(m2, radii, torch.ones((1, 4), device=_dev()), TS, 4, 3)
    assert isect.sizes()[0] == 0
    bg = torch.tensor([[0.5, 0.25, 0.125, 0.0]], device=_dev())
    rc, ra, last = ops.rasterize_to_pixels_fwd(m2, torch.ones((1, 4, 3), device=_dev()), torch.ones((1, 4, 4), device=_dev()),
                                               torch.ones((1, 4), device=_dev()), bg, W, H, TS, isect)
    assert float(ra.abs().max()) == 0.0 and int(last.abs().max()) == 0
    assert torch.equal(rc, bg.expand(1, H, W, 4).reshape(1, H, W, 4))
    o = ops.rasterize_to_pixels_bwd(m2, torch.ones((1, 4, 3), device=_dev()), torch.ones((1, 4, 4), device=_dev()),
                                    torch.ones((1, 4), device=_dev()), bg, W, H, TS, isect, ra, last, torch.ones_like(rc),
                                    torch.ones_like(ra), absgrad=True)
    assert all(float(t.abs().max()) == 0.0 for t in o)


def test_raster_linearity_in_colour():
    """Size-independent property: the ges forward is linear in the colour channels."""
    from gps_slam_amd import gsplat_ops as ops
    N, W, H, TS, delta = 200000, 640, 480, 16, 0.1
    radii, m2, depths, conics, colors, opac, ref_depth = _raster_state(N, W, H, seed=9)
    isect = ops.isect_tiles_no_depth(T(m2)[None], T(radii)[None], TS, 40, 30)
    tm2, tcon, top, tref = T(m2)[None], T(conics)[None], T(opac)[:, None], T(ref_depth)[None, ..., None]
    c1 = T(colors)[None]
    c2 = c1.clone()
    c2[..., :3] = c2[..., :3] * 2.0 + 0.0
    r1, a1, _ = ops.rasterize_to_pixels_fwd_ges(tm2, tcon, c1, top, tref, W, H, TS, isect, delta)
    r2, a2, _ = ops.rasterize_to_pixels_fwd_ges(tm2, tcon, c2, top, tref, W, H, TS, isect, delta)
    assert torch.equal(a1, a2)
    torch.testing.assert_close(r2[..., :3], 2.0 * r1[..., :3], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(r2[..., 3], r1[..., 3], rtol=0, atol=0)


def test_compose_l1_matches_autograd():
    from gps_slam_amd import gsplat_ops as ops
    H, W = 480, 640
    gen = torch.Generator(device="cpu").manual_seed(0)
    rc = (torch.rand((1, H, W, 4), generator=gen) * 3).to(_dev()).requires_grad_(True)
    ws = (torch.rand((1, H, W, 1), generator=gen) * 4).to(_dev()).requires_grad_(True)
    base = torch.rand((H, W, 3), generator=gen).to(_dev())
    ref = (torch.rand((H, W, 1), generator=gen) * 4 - 0.5).clamp_min(0).to(_dev())
    gt = torch.rand((H, W, 3), generator=gen).to(_dev())
    # the reference's libtorch sequence (raw_gs_model.cpp:318-326, :369-417)
    raw_rgb, raw_d = rc[..., :3], rc[..., 3:]
    bcw = torch.ones_like(ws)
    e_rgb = ((raw_rgb + base * bcw) / (ws + bcw))[0]
    bdw = torch.zeros_like(ws).masked_fill(ref[None] > 0, 1)
    e_depth = ((raw_d + ref * bdw) / (ws + bdw))[0]
    e_loss = (gt - e_rgb).abs().mean()
    e_loss.backward()
    rgb, depth, loss, v_rc, v_ra = ops.compose_l1(rc.detach(), ws.detach(), base, ref, gt)
    torch.testing.assert_close(rgb, e_rgb.detach(), rtol=1e-6, atol=1e-7)
    m = torch.isfinite(e_depth.detach())
    torch.testing.assert_close(depth[m], e_depth.detach()[m], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(loss[0], e_loss.detach(), rtol=1e-4, atol=0)
    torch.testing.assert_close(v_rc, rc.grad, rtol=1e-5, atol=1e-12)
    torch.testing.assert_close(v_ra, ws.grad, rtol=1e-4, atol=1e-11)


def test_adam_matches_libtorch_sequence():
    """Pins the fused Adam against the op sequence of torch::optim::Adam::step run with ATen on the GPU
    (mul_/add_, mul_/addcmul_, sqrt/div/add_, addcdiv_) for several steps, including zero gradients."""
    from gps_slam_amd import gsplat_ops as ops
    gen = torch.Generator(device="cpu").manual_seed(1)
    shapes = [(50000, 3), (50000, 3), (50000, 4), (50000, 3), (50000, 15, 3), (50000, 1), (10, 3, 4)]
    lrs = [1.6e-4 * 1.1 * 3.3, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, 1e-3]
    b1, b2, eps = 0.9, 0.999, 1e-15
    P = [torch.randn(s, generator=gen).to(_dev()) for s in shapes]
    Pe = [p.clone() for p in P]
    M = [torch.zeros_like(p) for p in P]
    V = [torch.zeros_like(p) for p in P]
    Me = [torch.zeros_like(p) for p in P]
    Ve = [torch.zeros_like(p) for p in P]
    for step in range(1, 6):
        G = [torch.randn(s, generator=gen).to(_dev()) * (0.0 if (step == 3 and k % 2) else 1e-3) for k, s in enumerate(shapes)]
        for p, g, m, v, lr in zip(Pe, G, Me, Ve, lrs):
            bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)  # libtorch: std::sqrt(bias_correction2)
            p.addcdiv_(m, denom, value=-(lr / bc1))
        ops.adam_step(P, G, M, V, lrs, step, (b1, b2), eps)
        for a, b in zip(M + V, Me + Ve):
            assert torch.equal(a, b), "exp_avg / exp_avg_sq must be bit-identical to the ATen op sequence"
        # param update: fma(-lr/bc1, m/denom, p) with an IEEE-correct m/denom (what nvcc's default --prec-div gives the
        # CUDA reference).  ATen-on-ROCm's tensor/tensor division is not always correctly rounded, so a handful of
        # elements may differ from it by one ulp; nothing else may.
        for a, b in zip(P, Pe):
            ne = a != b
            assert ne.float().mean().item() < 1e-4
            torch.testing.assert_close(a, b, rtol=1e-6, atol=3e-8)  # 1 ulp of the update term (cancellation p ~ -update)


@pytest.mark.parametrize("N,W,H", [(3000, 96, 64), (150000, 640, 480), (4000, 50, 37), (1, 33, 17), (30000, 1600, 720)])
def test_record_streaming_forward_equals_lds_forward(N, W, H):
    """gps_raster_ges_fwd_rec (packed records, 2 px/lane packed math, exp2 with the opacity folded into the exponent -- the
    forward the fused path times) must reproduce the operator-level gps_raster_ges_fwd."""
    from gps_slam_amd import gsplat_ops as ops
    TS, delta = 16, 0.1
    tw, th = (W + TS - 1) // TS, (H + TS - 1) // TS
    g, vm, K, c2w = _setup(N, W, H, seed=N + 1)
    rng = np.random.default_rng(N)
    ref_depth = rng.uniform(1.5, 4.5, (H, W)).astype(np.float32)
    ref_depth[rng.uniform(size=(H, W)) < 0.1] = 1000.0
    rec = torch.empty((N, 12), device=_dev())
    sh = T(g["sh"])
    radii, m2, depths, conics, colors, opac = ops.gauss_preprocess_fwd(
        T(g["means"]), T(g["log_scales"]), T(g["quats"]), T(g["opac_logit"]).view(-1), sh[:, 0].contiguous(),
        sh[:, 1:].contiguous(), 3, T(vm), T(K), T(c2w[:3, 3].copy()), W, H, records=rec)
    isect = ops.isect_tiles_no_depth(m2.view(1, N, 2), radii.view(1, N), TS, tw, th)
    tref = T(ref_depth)[None, ..., None]
    rc1, ra1, _ = ops.rasterize_to_pixels_fwd_ges(m2, conics, colors, opac, tref, W, H, TS, isect, delta)
    rc2, ra2 = ops.rasterize_to_pixels_fwd_ges_rec(rec, tref, W, H, isect, delta)
    assert N < 1000 or ra1.max().item() > 1.0
    # The streaming kernel folds the opacity into the exponent (alpha = exp2(-(sigma*log2e - log2 o))) and adds the
    # tile's list in two halves, so it agrees with the operator-level kernel to rounding, except for (pixel, Gaussian)
    # pairs whose alpha sits within rounding of the 1/255 cut-off (a flip moves the pixel by ~4e-3*|c|).  Same bar as
    # the oracle comparison of the forward pass: rtol 2e-4 / atol 2e-4, <= 2e-5 of the pixels off by a flip.
    for got, ref in ((rc2, rc1), (ra2, ra1)):
        d = (got - ref).abs()
        bad = d > (2e-4 * ref.abs() + 2e-4)
        print("streaming-vs-lds: max %.3g, frac over tol %.3g" % (d.max().item(), bad.float().mean().item()))
        assert bad.float().mean().item() <= 2e-5
        assert d.max().item() < 0.05


@pytest.mark.parametrize("N,W,H", [(150000, 640, 480), (300000, 1200, 680), (4000, 50, 37), (1, 33, 17), (30000, 1600, 720), (600, 2560, 1440)])
def test_persistent_forward_is_bit_identical_to_the_per_tile_forward(N, W, H):
    """raster_ges_fwd_pp_kernel (persistent workgroups, snake-dealt tiles, records staged one item ahead) against
    raster_ges_fwd_pk_kernel (one workgroup per tile): the same staging arithmetic, culling, survivor partition and summation order,
    so the render and the weight sum must be IDENTICAL bit for bit -- row-major tile order and a permuted tile_order (the dealing
    changes which workgroup renders a tile, never what it computes).  Sizes: BASELINE's 640x480, Replica's 1200x680 (ragged last
    tile row), tiny images (fewer tiles than workgroups: the launcher keeps the per-tile kernel), 1600x720 and a 1440p image whose
    14,400 tiles are 19 passes per workgroup, many of them with an empty list."""
    from gps_slam_amd import gsplat_ops as ops
    from gps_slam_amd._lib import check, lib
    import ctypes as C
    TS, delta = 16, 0.1
    tw, th = (W + TS - 1) // TS, (H + TS - 1) // TS
    g, vm, K, c2w = _setup(N, W, H, seed=N + 3)
    rng = np.random.default_rng(N)
    ref_depth = rng.uniform(1.5, 4.5, (H, W)).astype(np.float32)
    ref_depth[rng.uniform(size=(H, W)) < 0.1] = 1000.0
    rec = torch.empty((N, 12), device=_dev())
    sh = T(g["sh"])
    radii, m2, depths, conics, colors, opac = ops.gauss_preprocess_fwd(
        T(g["means"]), T(g["log_scales"]), T(g["quats"]), T(g["opac_logit"]).view(-1), sh[:, 0].contiguous(),
        sh[:, 1:].contiguous(), 3, T(vm), T(K), T(c2w[:3, 3].copy()), W, H, records=rec)
    isect = ops.isect_tiles_no_depth(m2.view(1, N, 2), radii.view(1, N), TS, tw, th)
    tref = T(ref_depth)[None, ..., None].contiguous()
    order = torch.as_tensor(rng.permutation(tw * th).astype(np.int32)).to(_dev())
    ptr = lambda t: C.c_void_p(t.data_ptr())
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def render(persistent, tile_order):
        lib.gps_set_raster_fwd_persistent(1 if persistent else 0)
        rc = torch.full((1, H, W, 4), float("nan"), device=_dev())
        ra = torch.full((1, H, W, 1), float("nan"), device=_dev())
        check(lib.gps_raster_ges_fwd_rec_ordered(N, ptr(rec), ptr(tref), W, H, ptr(isect.isect_offsets), ptr(isect.flatten_ids), ptr(isect.counts),
                                                 delta, ptr(rc), ptr(ra), ptr(tile_order) if tile_order is not None else C.c_void_p(0), sp),
              "gps_raster_ges_fwd_rec_ordered")
        torch.cuda.synchronize()
        return rc, ra
    try:
        rc0, ra0 = render(False, None)
        assert torch.isfinite(rc0).all() and torch.isfinite(ra0).all()
        for tile_order in (None, order):
            rc1, ra1 = render(True, tile_order)
            assert torch.equal(rc1, rc0) and torch.equal(ra1, ra0), "persistent forward differs (tile_order %s)" % ("given" if tile_order is not None else "row-major")
    finally:
        lib.gps_set_raster_fwd_persistent(0)   # (the shipped default)


def test_adam_step_one_writes_the_moments_without_reading_them():
    """Step 1 is the first step of a fresh torch::optim::Adam (state created as zeros): gps_adam_step takes m = v = 0 without
    reading exp_avg / exp_avg_sq, so buffers full of NaN give the result of zeroed ones, and step 2 carries on from what step 1
    wrote (sizes with a scalar tail included)."""
    from gps_slam_amd import gsplat_ops as ops
    gen = torch.Generator(device="cpu").manual_seed(2)
    shapes = [(40001, 3), (40001, 15, 3), (7,), (40001, 1)]
    lrs = [1e-3, 1.25e-4, 5e-2, 5e-3]
    P = [torch.randn(s, generator=gen).to(_dev()) for s in shapes]
    Pz = [p.clone() for p in P]
    M, V = [torch.full_like(p, float("nan")) for p in P], [torch.full_like(p, float("nan")) for p in P]
    Mz, Vz = [torch.zeros_like(p) for p in P], [torch.zeros_like(p) for p in P]
    for step in (1, 2):
        G = [torch.randn(s, generator=gen).to(_dev()) * 1e-3 for s in shapes]
        ops.adam_step(P, G, M, V, lrs, step)
        ops.adam_step(Pz, G, Mz, Vz, lrs, step)
        for x, y in zip(P + M + V, Pz + Mz + Vz):
            assert torch.isfinite(x).all() and torch.equal(x, y)


def test_fused_adam_is_bit_identical_to_separate_step():
    """gps_gauss_preprocess_bwd_adam (Adam step inside the backward kernel: sh_rest on the LDS tiles, optionally the five
    small tensors per thread) must leave exactly the parameters / exp_avg / exp_avg_sq that gps_gauss_preprocess_bwd +
    gps_adam_step produce, for several steps, and the same gradients where they are requested.  N is not a multiple of the
    workgroup rows (tail tile)."""
    import ctypes as C
    from gps_slam_amd import gsplat_ops as ops
    from gps_slam_amd._lib import AdamSegment, check, lib
    N, W, H = 10007, 160, 120
    g, vm, K, c2w = _setup(N, W, H, seed=11)
    sh = T(g["sh"])
    names = ("means", "ls", "q", "ol", "dc", "rest")
    P0 = dict(means=T(g["means"]), ls=T(g["log_scales"]), q=T(g["quats"]), ol=T(g["opac_logit"]).view(-1).contiguous(),
              dc=sh[:, 0].contiguous(), rest=sh[:, 1:].contiguous())
    lrs = dict(means=1.6e-4, ls=5e-3, q=1e-3, ol=5e-2, dc=2.5e-3, rest=5e-4)
    vmT, KT, cp = T(vm), T(K), T(c2w[:3, 3].copy())
    gen = torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=gen).to(_dev())
    v_m2, v_con, v_col, v_op = rnd(N, 2), rnd(N, 3), rnd(N, 4), rnd(N)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    for fuse_small in (False, True):
        A = {k: v.clone() for k, v in P0.items()}   # separate kernels
        B = {k: v.clone() for k, v in P0.items()}   # fused
        mA, vA = {k: torch.zeros_like(v) for k, v in A.items()}, {k: torch.zeros_like(v) for k, v in A.items()}
        mB, vB = {k: torch.zeros_like(v) for k, v in B.items()}, {k: torch.zeros_like(v) for k, v in B.items()}
        stepped = names if fuse_small else ("rest",)
        for step in (1, 2, 3):
            # the projected state both paths consume (parameters move every step when the small tensors are stepped)
            radii, m2, depths, conics, colors, opac = ops.gauss_preprocess_fwd(A["means"], A["ls"], A["q"], A["ol"], A["dc"],
                                                                              A["rest"], 3, vmT, KT, cp, W, H)
            gA = ops.gauss_preprocess_bwd(A["means"], A["ls"], A["q"], A["ol"], A["dc"], A["rest"], 3, vmT, KT, cp, W, H, 0.3,
                                          radii, conics, v_m2, v_con, v_col, v_op)
            gmap = dict(means=gA[0], ls=gA[1], q=gA[2], ol=gA[3], dc=gA[4], rest=gA[5])
            ops.adam_step([A[k] for k in stepped], [gmap[k] for k in stepped], [mA[k] for k in stepped],
                          [vA[k] for k in stepped], [lrs[k] for k in stepped], step)
            want_g = step == 2 or not fuse_small
            outB = [torch.empty_like(x) for x in gA[:5]]
            g_restB = torch.empty_like(B["rest"])
            small = None
            if fuse_small:
                small = (AdamSegment * 5)()
                for j, k in enumerate(("means", "ls", "q", "dc", "ol")):
                    small[j].param, small[j].exp_avg, small[j].exp_avg_sq = B[k].data_ptr(), mB[k].data_ptr(), vB[k].data_ptr()
                    small[j].numel, small[j].lr = B[k].numel(), lrs[k]
            og = [p(x) if want_g else None for x in outB]
            check(lib.gps_gauss_preprocess_bwd_adam(N, 16, 3, p(B["means"]), p(B["ls"]), p(B["q"]), p(B["ol"]), p(B["dc"]),
                                                    p(B["rest"]), p(vmT), p(KT), p(cp), W, H, 0.3, p(radii), p(conics), p(v_m2),
                                                    p(v_con), p(v_col), p(v_op), og[0], og[1], og[2], og[3], og[4],
                                                    p(g_restB) if step == 2 else None, p(mB["rest"]), p(vB["rest"]), lrs["rest"],
                                                    small, 0.9, 0.999, 1e-15, step, st), "gps_gauss_preprocess_bwd_adam")
            torch.cuda.synchronize()
            for k in names:
                assert torch.equal(A[k], B[k]) and torch.equal(mA[k], mB[k]) and torch.equal(vA[k], vB[k]), (fuse_small, step, k)
            if want_g:
                for a, b in zip(gA[:5], outB):
                    assert torch.equal(a, b)
            if step == 2:
                assert torch.equal(g_restB, gA[5])
        for k in stepped:
            assert (A[k] != P0[k]).any(), k


def test_hip_chain_matches_committed_splat_golden():
    """The HIP operators on the inputs of tests/golden/splat_96x64_n300.npz against the frozen oracle outputs (no oracle
    call here: fixture only).  Binning exactly; floats at the documented tolerances."""
    import os
    from gps_slam_amd import gsplat_ops as ops
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "splat_96x64_n300.npz"))
    W, H, TS = int(G["W"]), int(G["H"]), int(G["TS"])
    tw, th = (W + TS - 1) // TS, (H + TS - 1) // TS
    scales = np.exp(G["log_scales"]).astype(np.float32)
    radii, m2, depths, conics = ops.fully_fused_projection_fwd(T(G["means"]), T(G["quats"]), T(scales), T(G["viewmat"])[None],
                                                               T(G["K"])[None], W, H)
    radii = radii.clamp_max(100)
    assert np.array_equal(N_(radii)[0], G["radii"])
    np.testing.assert_allclose(N_(m2)[0], G["means2d"], rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(N_(conics)[0], G["conics"], rtol=2e-3, atol=1e-6)
    # downstream operators on the FIXTURE's projected state, so that every stage is compared on identical inputs
    isect = ops.isect_tiles_no_depth(T(G["means2d"])[None], T(G["radii"])[None], TS, tw, th, want_isect_ids=True)
    tpg, ids, flat, ggs, gst, offs = isect.trimmed()
    assert np.array_equal(N_(tpg)[0], G["tiles_per_gauss"]) and np.array_equal(N_(ids), G["isect_ids"])
    assert np.array_equal(N_(flat), G["flatten_ids"]) and np.array_equal(N_(ggs), G["group_gs_ids"])
    assert np.array_equal(N_(gst), G["group_starts"]) and np.array_equal(N_(offs).reshape(-1), G["offsets"].reshape(-1))
    tref = T(G["ref_depth"])[None, ..., None]
    rc, ra, _ = ops.rasterize_to_pixels_fwd_ges(T(G["means2d"])[None], T(G["conics"])[None], T(G["colors"])[None],
                                                T(G["opac"])[:, None], tref, W, H, TS, isect, 0.1)
    # per element: rounding (2e-5 of the sum of |terms|) + what the fixture's borderline accept/reject pairs could contribute
    got = np.concatenate([N_(rc)[0], N_(ra)[0]], -1)
    ref = np.concatenate([G["render_colors"], G["weight_sum"][..., None]], -1)
    SIG = 2 * 2.0 ** -23  # 2 ulp of sigma's largest term (see test_raster_ges_fwd_bwd)
    assert (np.abs(got - ref) <= 2e-5 * G["fwd_scale"] + SIG * G["fwd_sig"] + 1e-7 + 1.001 * G["fwd_flip"]).all()
    o = ops.rasterize_to_pixels_bwd_ges_gs_parallel(T(G["means2d"])[None], T(G["conics"])[None], T(G["colors"])[None],
                                                    T(G["opac"])[:, None], T(G["radii"])[None], tref, W, H, isect, 0.1,
                                                    T(G["v_rc"])[None], T(G["v_ra"])[None, ..., None])
    Ng = G["means2d"].shape[0]
    got = np.concatenate([N_(o[2]).reshape(Ng, 4), N_(o[1]).reshape(Ng, 3), N_(o[0]).reshape(Ng, 2), N_(o[3]).reshape(Ng, 1)], 1)
    ref = np.concatenate([G["v_colors"], G["v_conics"], G["v_means2d"], G["v_opacities"].reshape(Ng, 1)], 1)
    assert (np.abs(got - ref) <= 2e-5 * G["bwd_scale"] + SIG * G["bwd_sig"] + 1e-30 + 1.001 * G["bwd_flip"]).all()


@pytest.mark.parametrize("B,CH,H,W", [(1, 3, 37, 50), (2, 3, 64, 96), (1, 1, 33, 31)])
def test_fused_ssim_fwd_bwd_vs_oracle_in_both_layouts(B, CH, H, W):
    """gps_ssim_fwd / gps_ssim_bwd vs the restatement of ssim.cu (itself pinned against float64 conv2d + autograd): planar
    (the reference's NCHW) and channels-last (HWC, read in place) give the same numbers."""
    from gps_slam_amd import gsplat_ops as ops
    from oracle import splat_ref as orc
    rng = np.random.default_rng(B * 100 + H)
    img2 = rng.uniform(0, 1, (B, CH, H, W)).astype(np.float32)
    img1 = np.clip(img2 + rng.normal(0, 0.15, img2.shape), 0, 1).astype(np.float32)
    C1, C2 = float(np.float32(0.01 * 0.01)), float(np.float32(0.03 * 0.03))
    e_m, e1, e2, e3 = orc.ssim_fwd(img1, img2, C1, C2)
    m, d1, d2, d3 = ops.fusedssim(C1, C2, T(img1), T(img2), train=True)
    for got, ref, name in ((m, e_m, "map"), (d1, e1, "dm_dmu1"), (d2, e2, "dm_dsigma1_sq"), (d3, e3, "dm_dsigma12")):
        np.testing.assert_allclose(N_(got), ref, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=name)
    dL = rng.normal(size=img1.shape).astype(np.float32)
    e_g = orc.ssim_bwd(img1, img2, dL, e1, e2, e3)
    g = ops.fusedssim_backward(C1, C2, T(img1), T(img2), T(dL), T(e1), T(e2), T(e3))
    _grad_close(N_(g), e_g, "dL_dimg1", rtol=1e-3, atol_rel=1e-4)
    # channels-last: bit-identical values at transposed positions
    cl = lambda a: T(np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1))))
    m_cl, c1, c2, c3 = ops.fusedssim(C1, C2, cl(img1), cl(img2), train=True, channels_last=True)
    assert torch.equal(m_cl.permute(0, 3, 1, 2), m) and torch.equal(c2.permute(0, 3, 1, 2), d2)
    g_cl = ops.fusedssim_backward(C1, C2, cl(img1), cl(img2), cl(dL), cl(e1), cl(e2), cl(e3), channels_last=True)
    assert torch.equal(g_cl.permute(0, 3, 1, 2), g)
    # not training: the map only
    m2, n1, _, _ = ops.fusedssim(C1, C2, T(img1), T(img2), train=False)
    assert n1 is None and torch.equal(m2, m)


def test_fused_ssim_loss_autograd_matches_conv2d_autograd_at_full_size():
    """FusedSSIMMap through compute_loss (raw_gs_model.cpp:369-417, ssim_weight 0.2, padding "valid") at 640x480 vs the same loss
    written with torch conv2d in float64 and differentiated by autograd."""
    import torch.nn.functional as F
    from gps_slam_amd import gsplat_wapper as gw
    H, W = 480, 640
    gen = torch.Generator().manual_seed(4)
    gt = torch.rand((H, W, 3), generator=gen).to(_dev())
    rgb = (gt + 0.1 * torch.randn((H, W, 3), generator=gen).to(_dev())).clamp(0, 1).requires_grad_(True)
    loss = gw.compute_loss(dict(rgb=rgb), gt, ssim_weight=0.2)
    loss["total"].backward()
    g = torch.tensor([0.001028380123898387, 0.0075987582094967365, 0.036000773310661316, 0.10936068743467331, 0.21300552785396576,
                      0.26601171493530273, 0.21300552785396576, 0.10936068743467331, 0.036000773310661316, 0.0075987582094967365,
                      0.001028380123898387], dtype=torch.float64, device=_dev())
    win = (g[:, None] * g[None, :]).expand(3, 1, 11, 11).contiguous()
    conv = lambda t: F.conv2d(t, win, padding=5, groups=3)
    r64 = rgb.detach().double().requires_grad_(True)
    a, b = r64.permute(2, 0, 1)[None], gt.double().permute(2, 0, 1)[None]
    mu1, mu2 = conv(a), conv(b)
    s1, s2, s12 = conv(a * a) - mu1 * mu1, conv(b * b) - mu2 * mu2, conv(a * b) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    want = 0.8 * (b - a).abs().mean() + 0.2 * (1.0 - ssim[:, :, 5:-5, 5:-5].mean())
    want.backward()
    torch.testing.assert_close(loss["total"].double(), want, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rgb.grad.double(), r64.grad, rtol=2e-3, atol=2e-4 * float(r64.grad.abs().max()))

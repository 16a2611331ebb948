"""Pins the splat CPU oracle (oracle/splat_oracle.c).

The reference ships no tests/fixtures for the splat path and is CUDA-only, so
the oracle is cross-checked here against an independent dense float64 PyTorch
formulation of the same maths (values + autograd gradients) and against the
exact tile adjoint.  CPU only.
"""
import os

import numpy as np
import pytest
import torch

from oracle import splat_ref as orc
from tests import scenes

torch.manual_seed(0)
W, H, TS = 96, 64, 16
TW, TH = (W + TS - 1) // TS, (H + TS - 1) // TS


def _torch_project(means, quats, scales, viewmat, K, W, H, eps2d=0.3):
    """Dense float64 restatement of the pinhole projection maths (for autograd)."""
    R, t = viewmat[:3, :3], viewmat[:3, 3]
    mc = means @ R.T + t
    q = quats / quats.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    Rq = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    M = Rq * scales[:, None, :]
    cov = M @ M.transpose(1, 2)
    cov_c = R @ cov @ R.T
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    X, Y, Z = mc.unbind(1)
    tfx, tfy = 0.5 * W / fx, 0.5 * H / fy
    lxp, lxn = (W - cx) / fx + 0.3 * tfx, cx / fx + 0.3 * tfx
    lyp, lyn = (H - cy) / fy + 0.3 * tfy, cy / fy + 0.3 * tfy
    rz = 1 / Z
    tx = Z * torch.minimum(lxp, torch.maximum(-lxn, X * rz))
    ty = Z * torch.minimum(lyp, torch.maximum(-lyn, Y * rz))
    zero = torch.zeros_like(Z)
    J = torch.stack([fx * rz, zero, -fx * tx * rz * rz, zero, fy * rz, -fy * ty * rz * rz], 1).reshape(-1, 2, 3)
    c2 = J @ cov_c @ J.transpose(1, 2)
    c2 = c2 + eps2d * torch.eye(2, dtype=c2.dtype)
    det = c2[:, 0, 0] * c2[:, 1, 1] - c2[:, 0, 1] * c2[:, 1, 0]
    conic = torch.stack([c2[:, 1, 1] / det, -c2[:, 0, 1] / det, c2[:, 0, 0] / det], 1)
    m2 = torch.stack([fx * X * rz + cx, fy * Y * rz + cy], 1)
    b = 0.5 * (c2[:, 0, 0] + c2[:, 1, 1])
    radius = torch.ceil(3 * torch.sqrt(b + torch.sqrt(torch.clamp(b * b - det, min=0.01))))
    return m2, Z, conic, radius, det


def _scene(N=400, seed=3):
    g = scenes.random_gaussians(N, seed=seed, scale_range=(0.01, 0.12))
    c2w, K = scenes.default_camera(W, H, seed=seed)
    return g, scenes.pose_inv(c2w), K, c2w


def test_projection_forward_matches_dense_torch():
    g, vm, K, _ = _scene()
    scales = np.exp(g["log_scales"])
    radii, m2, depths, conics = orc.proj_fwd(g["means"], g["quats"], scales, vm, K, W, H)
    tm2, tz, tconic, tr, tdet = _torch_project(*[torch.tensor(a, dtype=torch.float64) for a in
                                                 (g["means"], g["quats"], scales, vm, K)], W, H)
    vis = radii > 0
    assert vis.sum() > 50
    np.testing.assert_allclose(m2[vis], tm2.numpy()[vis], rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(depths[vis], tz.numpy()[vis], rtol=1e-5)
    np.testing.assert_allclose(conics[vis], tconic.numpy()[vis], rtol=2e-3, atol=1e-5)
    # radius: identical except where 3*sqrt(v1) sits within rounding of an integer
    assert (np.abs(radii[vis] - tr.numpy()[vis]) <= 1).all()
    assert (radii[vis] == tr.numpy()[vis]).mean() > 0.99
    # culling: behind-camera, and fully off-screen Gaussians have radius 0
    off = (tm2[:, 0] + tr <= 0) | (tm2[:, 0] - tr >= W) | (tm2[:, 1] + tr <= 0) | (tm2[:, 1] - tr >= H) | (tz < 0.01)
    assert (radii[off.numpy()] == 0).all()


def test_projection_backward_matches_autograd():
    g, vm, K, _ = _scene(N=300, seed=5)
    scales = np.exp(g["log_scales"])
    radii, m2, depths, conics = orc.proj_fwd(g["means"], g["quats"], scales, vm, K, W, H)
    rng = np.random.default_rng(0)
    N = radii.shape[0]
    v_m2 = rng.normal(size=(N, 2)).astype(np.float32)
    v_d = rng.normal(size=N).astype(np.float32)
    v_c = rng.normal(size=(N, 3)).astype(np.float32) * 0.1
    vm_, vq_, vs_ = orc.proj_bwd(g["means"], g["quats"], scales, vm, K, W, H, radii, conics, v_m2, v_d, v_c)
    tm = torch.tensor(g["means"], dtype=torch.float64, requires_grad=True)
    tq = torch.tensor(g["quats"], dtype=torch.float64, requires_grad=True)
    ts = torch.tensor(scales, dtype=torch.float64, requires_grad=True)
    tm2, tz, tconic, _, _ = _torch_project(tm, tq, ts, torch.tensor(vm, dtype=torch.float64),
                                           torch.tensor(K, dtype=torch.float64), W, H)
    mask = torch.tensor(radii > 0)
    loss = ((tm2 * torch.tensor(v_m2))[mask].sum() + (tz * torch.tensor(v_d))[mask].sum()
            + (tconic * torch.tensor(v_c))[mask].sum())
    loss.backward()
    vis = radii > 0
    for got, ref in ((vm_, tm.grad), (vq_, tq.grad), (vs_, ts.grad)):
        ref = ref.numpy()
        scale = np.abs(ref[vis]).max()
        np.testing.assert_allclose(got[vis], ref[vis], rtol=5e-3, atol=2e-4 * scale)
        assert (got[~vis] == 0).all()


def _torch_sh(deg, dirs, coeffs):
    """Real SH (Sloan) written from the closed-form polynomials, float64."""
    d = dirs / dirs.norm(dim=1, keepdim=True)
    x, y, z = d.unbind(1)
    Y = [0.2820947917738781 * torch.ones_like(x)]
    if deg >= 1:
        Y += [-0.48860251190292 * y, 0.48860251190292 * z, -0.48860251190292 * x]
    if deg >= 2:
        Y += [1.092548430592079 * x * y, -1.092548430592079 * y * z, 0.9461746957575601 * z * z - 0.3153915652525201,
              -1.092548430592079 * x * z, 0.5462742152960395 * (x * x - y * y)]
    if deg >= 3:
        Y += [-0.5900435899266435 * y * (3 * x * x - y * y), 2.890611442640554 * x * y * z,
              -0.4570457994644658 * y * (5 * z * z - 1), 0.3731763325901154 * z * (5 * z * z - 3),
              -0.4570457994644658 * x * (5 * z * z - 1), 1.445305721320277 * z * (x * x - y * y),
              -0.5900435899266435 * x * (x * x - 3 * y * y)]
    Y = torch.stack(Y, 1)
    return (Y[:, :, None] * coeffs[:, :Y.shape[1]]).sum(1)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_forward_backward(deg):
    rng = np.random.default_rng(deg)
    N, K = 200, 16
    dirs = rng.normal(size=(N, 3)).astype(np.float32) * 2
    coeffs = rng.normal(size=(N, K, 3)).astype(np.float32)
    masks = rng.uniform(size=N) > 0.2
    col = orc.sh_fwd(deg, dirs, coeffs, masks)
    td = torch.tensor(dirs, dtype=torch.float64, requires_grad=True)
    tc = torch.tensor(coeffs, dtype=torch.float64, requires_grad=True)
    tcol = _torch_sh(deg, td, tc)
    np.testing.assert_allclose(col[masks], tcol.detach().numpy()[masks], rtol=1e-4, atol=1e-5)
    assert (col[~masks] == 0).all()  # untouched
    v_col = rng.normal(size=(N, 3)).astype(np.float32)
    (tcol * torch.tensor(v_col))[torch.tensor(masks)].sum().backward()
    v_coeffs, v_dirs = orc.sh_bwd(deg, dirs, coeffs, masks, v_col)
    np.testing.assert_allclose(v_coeffs, tc.grad.numpy(), rtol=1e-4, atol=1e-5)
    tdg = np.zeros_like(dirs) if td.grad is None else td.grad.numpy()  # degree 0 has no view dependence
    np.testing.assert_allclose(v_dirs, tdg, rtol=2e-3, atol=1e-4)


def _raster_inputs(N=300, seed=7, big=False):
    rng = np.random.default_rng(seed)
    m2 = np.stack([rng.uniform(-6, W + 6, N), rng.uniform(-6, H + 6, N)], 1).astype(np.float32)
    radii = rng.integers(1, 30 if big else 9, N).astype(np.int32)
    radii[rng.uniform(size=N) < 0.1] = 0
    # conic consistent with the radius: sigma ~ radius/3
    s = np.maximum(radii, 1) / 3.0
    a = 1.0 / (s * rng.uniform(0.6, 1.0, N)) ** 2
    c = 1.0 / (s * rng.uniform(0.6, 1.0, N)) ** 2
    b = rng.uniform(-0.5, 0.5, N) * np.sqrt(a * c)
    conics = np.stack([a, b, c], 1).astype(np.float32)
    colors = np.concatenate([rng.uniform(0, 1, (N, 3)), rng.uniform(0.5, 4.0, (N, 1))], 1).astype(np.float32)
    opac = rng.uniform(0.02, 1.0, N).astype(np.float32)
    ref_depth = rng.uniform(1.0, 4.5, (H, W)).astype(np.float32)
    ref_depth[rng.uniform(size=(H, W)) < 0.1] = 1000.0
    return m2, radii, conics, colors, opac, ref_depth


def _dense_raster(m2, conics, colors, opac, ref_depth, cover, delta):
    """Dense float64 formulation: cover[N,H,W] says which (gaussian,pixel) pairs the tiles visit."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64) + 0.5, torch.arange(W, dtype=torch.float64) + 0.5,
                            indexing="ij")
    dx = m2[:, 0, None, None] - xs
    dy = m2[:, 1, None, None] - ys
    sigma = 0.5 * (conics[:, 0, None, None] * dx * dx + conics[:, 2, None, None] * dy * dy) + conics[:, 1, None, None] * dx * dy
    vis = torch.exp(-sigma)
    alpha = torch.clamp(opac[:, None, None] * vis, max=0.999)
    ok = cover & (sigma >= 0) & (alpha >= 1.0 / 255.0) & ~(colors[:, 3, None, None] > ref_depth + delta)
    alpha = torch.where(ok, alpha, torch.zeros_like(alpha))
    out = (alpha[:, :, :, None] * colors[:, None, None, :]).sum(0)
    return out, alpha.sum(0)


def _tile_cover(m2, radii):
    cover = np.zeros((m2.shape[0], H, W), bool)
    for g in range(m2.shape[0]):
        if radii[g] <= 0:
            continue
        r = float(radii[g])
        tx0 = min(max(0, int(np.floor(np.float32(m2[g, 0] / TS) - np.float32(r / TS)))), TW)
        ty0 = min(max(0, int(np.floor(np.float32(m2[g, 1] / TS) - np.float32(r / TS)))), TH)
        tx1 = min(max(0, int(np.ceil(np.float32(m2[g, 0] / TS) + np.float32(r / TS)))), TW)
        ty1 = min(max(0, int(np.ceil(np.float32(m2[g, 1] / TS) + np.float32(r / TS)))), TH)
        cover[g, ty0 * TS:ty1 * TS, tx0 * TS:tx1 * TS] = True
    return cover


def test_binning_invariants():
    m2, radii, *_ = _raster_inputs(N=500, seed=11, big=True)
    tpg, ids, flat, ggs, gst, offs = orc.isect_tiles(m2, radii, TS, TW, TH)
    cover = _tile_cover(m2, radii)
    assert tpg.sum() == ids.shape[0] == flat.shape[0]
    # every (gaussian, tile) pair appears exactly once, sorted by tile then by gaussian index
    assert (np.diff(ids) >= 0).all()
    for t in range(TW * TH):
        s, e = offs.reshape(-1)[t], (offs.reshape(-1)[t + 1] if t + 1 < TW * TH else ids.shape[0])
        assert (ids[s:e] == t).all()
        assert (np.diff(flat[s:e]) > 0).all()
        ty, tx = divmod(t, TW)
        expect = np.nonzero(cover[:, ty * TS, tx * TS])[0]
        np.testing.assert_array_equal(flat[s:e], expect)
    # group table: ceil(4 r^2 / 32) groups per visible gaussian, starts = exclusive prefix
    gpg = np.where(radii > 0, (4 * radii.astype(np.int64) ** 2 + 31) // 32, 0)
    assert ggs.shape[0] == gpg.sum()
    np.testing.assert_array_equal(ggs, np.repeat(np.arange(radii.shape[0]), gpg))
    starts = np.concatenate([[0], np.cumsum(gpg)[:-1]])
    np.testing.assert_array_equal(gst, np.repeat(starts, gpg))


def test_raster_forward_and_exact_adjoint_match_dense_autograd():
    m2, radii, conics, colors, opac, ref_depth = _raster_inputs()
    delta = 0.1
    tpg, ids, flat, ggs, gst, offs = orc.isect_tiles(m2, radii, TS, TW, TH)
    rc, ra, last = orc.raster_ges_fwd(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, delta)
    T = lambda a: torch.tensor(a, dtype=torch.float64)
    tm2, tcon, tcol, top = [T(a).requires_grad_(True) for a in (m2, conics, colors, opac)]
    cover = torch.tensor(_tile_cover(m2, radii))
    out, wsum = _dense_raster(tm2, tcon, tcol, top, T(ref_depth), cover, delta)
    np.testing.assert_allclose(rc, out.detach().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(ra, wsum.detach().numpy(), rtol=2e-4, atol=2e-4)
    rng = np.random.default_rng(1)
    v_rc = rng.normal(size=(H, W, 4)).astype(np.float32)
    v_ra = rng.normal(size=(H, W)).astype(np.float32)
    ((out * T(v_rc)).sum() + (wsum * T(v_ra)).sum()).backward()
    v_m, v_c, v_col, v_o = orc.raster_ges_bwd_exact(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, delta,
                                                    v_rc, v_ra)
    # the reference's hand-written adjoint treats min(0.999, .) as pass-through when opac*vis<=0.999
    # and blocks it otherwise, exactly like autograd's clamp -> same gradient.
    for got, ref in ((v_m, tm2.grad), (v_c, tcon.grad), (v_col, tcol.grad), (v_o, top.grad)):
        ref = ref.numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-4 * max(1.0, np.abs(ref).max()))


def test_gaussian_parallel_backward_box_semantics():
    """The shipped backward only visits the 2r x 2r integer box (bwd_ges_new_parallel.cu:83-96)."""
    m2, radii, conics, colors, opac, ref_depth = _raster_inputs(N=200, seed=21)
    delta = 0.1
    tpg, ids, flat, ggs, gst, offs = orc.isect_tiles(m2, radii, TS, TW, TH)
    rng = np.random.default_rng(2)
    v_rc = rng.normal(size=(H, W, 4)).astype(np.float32)
    v_ra = rng.normal(size=(H, W)).astype(np.float32)
    got = orc.raster_ges_bwd_gs(m2, conics, colors, opac, radii, ref_depth, W, H, ggs, gst, delta, v_rc, v_ra)
    # independent dense formulation restricted to the box
    T = lambda a: torch.tensor(a, dtype=torch.float64)
    N = m2.shape[0]
    box = np.zeros((N, H, W), bool)
    for g in range(N):
        r = int(radii[g])
        if r <= 0:
            continue
        x0, y0 = int(m2[g, 0]) - r, int(m2[g, 1]) - r  # int() truncates toward zero like the C cast
        js = np.arange(x0 + 1, x0 + 2 * r + 1)
        is_ = np.arange(y0 + 1, y0 + 2 * r + 1)
        js, is_ = js[(js >= 0) & (js < W)], is_[(is_ >= 0) & (is_ < H)]
        box[g][np.ix_(is_, js)] = True
    tm2, tcon, tcol, top = [T(a).requires_grad_(True) for a in (m2, conics, colors, opac)]
    out, wsum = _dense_raster(tm2, tcon, tcol, top, T(ref_depth), torch.tensor(box), delta)
    ((out * T(v_rc)).sum() + (wsum * T(v_ra)).sum()).backward()
    for a, ref in zip(got, (tm2.grad, tcon.grad, tcol.grad, top.grad)):
        ref = ref.numpy()
        np.testing.assert_allclose(a, ref, rtol=2e-3, atol=2e-4 * max(1.0, np.abs(ref).max()))
    # and where the box covers every pixel the forward touched, it equals the exact adjoint
    exact = orc.raster_ges_bwd_exact(m2, conics, colors, opac, ref_depth, W, H, TS, offs, flat, delta, v_rc, v_ra)
    cover = _tile_cover(m2, radii)
    dxs = (np.arange(W) + 0.5)[None, None, :] - m2[:, 0, None, None]
    dys = (np.arange(H) + 0.5)[None, :, None] - m2[:, 1, None, None]
    sig = 0.5 * (conics[:, 0, None, None] * dxs ** 2 + conics[:, 2, None, None] * dys ** 2) + conics[:, 1, None, None] * dxs * dys
    active = cover & (opac[:, None, None] * np.exp(-sig) >= 1 / 255.) & (sig >= 0)
    inside = ~(active & ~box).any(axis=(1, 2))
    assert inside.sum() > 20
    np.testing.assert_allclose(got[2][inside], exact[2][inside], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(got[3][inside], exact[3][inside], rtol=1e-3, atol=1e-4)


def test_oracle_reproduces_committed_splat_golden():
    """tests/golden/splat_96x64_n300.npz (made by tests/golden/make_splat_golden.py from this oracle after the cross-checks
    above) freezes the restatement: integer outputs exactly, float outputs to a few ulp (libm differences between hosts)."""
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "splat_96x64_n300.npz"))
    Wg, Hg, TSg = int(G["W"]), int(G["H"]), int(G["TS"])
    twg, thg = (Wg + TSg - 1) // TSg, (Hg + TSg - 1) // TSg
    scales = np.exp(G["log_scales"]).astype(np.float32)
    radii, m2, depths, conics = orc.proj_fwd(G["means"], G["quats"], scales, G["viewmat"], G["K"], Wg, Hg)
    radii = np.minimum(radii, 100).astype(np.int32)
    assert np.array_equal(radii, G["radii"])
    np.testing.assert_allclose(m2, G["means2d"], rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(conics, G["conics"], rtol=2e-5, atol=1e-7)
    tpg, ids, flat, ggs, gst, offs = orc.isect_tiles(G["means2d"], G["radii"], TSg, twg, thg)
    for a, b in ((tpg, "tiles_per_gauss"), (ids, "isect_ids"), (flat, "flatten_ids"), (ggs, "group_gs_ids"),
                 (gst, "group_starts"), (offs, "offsets")):
        assert np.array_equal(a, G[b]), b
    rc, ra, _ = orc.raster_ges_fwd(G["means2d"], G["conics"], G["colors"], G["opac"], G["ref_depth"], Wg, Hg, TSg,
                                   G["offsets"], G["flatten_ids"], 0.1)
    np.testing.assert_allclose(rc, G["render_colors"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ra, G["weight_sum"], rtol=1e-5, atol=1e-6)
    got = orc.raster_ges_bwd_gs(G["means2d"], G["conics"], G["colors"], G["opac"], G["radii"], G["ref_depth"], Wg, Hg,
                                G["group_gs_ids"], G["group_starts"], 0.1, G["v_rc"], G["v_ra"])
    for a, name in zip(got, ("v_means2d", "v_conics", "v_colors", "v_opacities")):
        np.testing.assert_allclose(a, G[name], rtol=1e-4, atol=1e-5 * np.abs(G[name]).max(), err_msg=name)


# ----------------------------------------------------------------------------- `raw` render method
def _dense_raw(means2d, conics, colors, opac, depths, radii, bg, Wd, Hd):
    """Dense float64 front-to-back compositing: every pixel sees every visible Gaussian in (depth, index) order, with the
    reference's per-pair rejection rules and its termination rule T * (1 - alpha) <= 1e-4 (exclusive)."""
    vis = torch.nonzero(radii > 0).squeeze(1)
    order = vis[torch.argsort(depths[vis], stable=True)]
    ys, xs = torch.meshgrid(torch.arange(Hd, dtype=torch.float64) + 0.5, torch.arange(Wd, dtype=torch.float64) + 0.5, indexing="ij")
    T = torch.ones(Hd, Wd, dtype=torch.float64)
    done = torch.zeros(Hd, Wd, dtype=torch.bool)
    out = torch.zeros(Hd, Wd, 4, dtype=torch.float64)
    for g in order.tolist():
        dx, dy = means2d[g, 0] - xs, means2d[g, 1] - ys
        sigma = 0.5 * (conics[g, 0] * dx * dx + conics[g, 2] * dy * dy) + conics[g, 1] * dx * dy
        alpha = torch.clamp_max(opac[g] * torch.exp(-sigma), 0.999)
        ok = (sigma >= 0) & (alpha >= 1.0 / 255.0) & ~done
        # pixels outside the Gaussian's tile footprint never see it in the tiled version: reject by the 3-sigma radius box
        ok = ok & (dx.abs() <= radii[g] + 16) & (dy.abs() <= radii[g] + 16)
        next_T = T * (1 - alpha)
        stop = ok & (next_T <= 1e-4)
        done = done | stop
        use = ok & ~stop
        out = out + torch.where(use[..., None], colors[g][None, None, :] * (alpha * T)[..., None], torch.zeros(1, dtype=torch.float64))
        T = torch.where(use, next_T, T)
    if bg is not None:
        out = out + T[..., None] * bg
    return out, 1 - T


def _raw_scene(N=250, seed=9):
    g = scenes.random_gaussians(N, seed=seed, scale_range=(0.02, 0.12))
    c2w, K = scenes.default_camera(W, H, seed=seed)
    vm = scenes.pose_inv(c2w)
    scales = np.exp(g["log_scales"]).astype(np.float32)
    radii, m2, depths, conics = orc.proj_fwd(g["means"], g["quats"], scales, vm, K, W, H)
    rng = np.random.default_rng(seed)
    colors = np.concatenate([rng.uniform(0, 1, (N, 3)), depths[:, None]], 1).astype(np.float32)
    opac = rng.uniform(0.05, 0.95, N).astype(np.float32)
    return radii, m2, depths, conics, colors, opac


def test_raw_binning_is_depth_sorted_and_stable():
    radii, m2, depths, conics, colors, opac = _raw_scene()
    depths = depths.copy()
    depths[10] = depths[11] = depths[12]  # ties keep index order
    tpg, ids, flat, offs = orc.isect_tiles_depth(m2, radii, depths, TS, TW, TH)
    assert len(ids) == tpg.sum() and np.all(np.diff(ids) >= 0)
    tiles = (ids >> 32).astype(np.int64)
    for t in range(TW * TH):
        lo, hi = offs.reshape(-1)[t], (offs.reshape(-1)[t + 1] if t + 1 < TW * TH else len(ids))
        assert np.all(tiles[lo:hi] == t)
        d = depths[flat[lo:hi]]
        assert np.all(np.diff(d) >= 0)
        same = np.nonzero(np.diff(d) == 0)[0]
        assert np.all(flat[lo:hi][same] < flat[lo:hi][same + 1])
    assert np.array_equal((ids & 0xFFFFFFFF).astype(np.uint32), depths[flat].view(np.uint32))


@pytest.mark.parametrize("with_bg", [False, True])
def test_raw_forward_and_backward_match_dense_compositing(with_bg):
    radii, m2, depths, conics, colors, opac = _raw_scene()
    bg = np.array([0.2, 0.4, 0.6, 0.0], np.float32) if with_bg else None
    tpg, ids, flat, offs = orc.isect_tiles_depth(m2, radii, depths, TS, TW, TH)
    rc, ra, last = orc.raster_raw_fwd(m2, conics, colors, opac, W, H, TS, offs, flat, backgrounds=bg)
    t = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    tm2, tcon, tcol, top = t(m2), t(conics), t(colors), t(opac)
    d_rc, d_ra = _dense_raw(tm2, tcon, tcol, top, torch.tensor(depths), torch.tensor(radii), None if bg is None else torch.tensor(bg, dtype=torch.float64), W, H)
    # a pair within float rounding of a threshold (alpha = 1/255, T = 1e-4) may flip: allow a handful of pixels
    for got, ref in ((rc, d_rc.detach().numpy()), (ra, d_ra.detach().numpy())):
        bad = np.abs(got - ref) > (1e-4 * np.abs(ref) + 1e-4)
        assert bad.mean() < 2e-3, bad.mean()
    assert ra.max() > 0.9 and (last > 0).any()
    rng = np.random.default_rng(1)
    v_rc = rng.normal(size=(H, W, 4)).astype(np.float32)
    v_ra = rng.normal(size=(H, W)).astype(np.float32)
    (d_rc * torch.tensor(v_rc, dtype=torch.float64)).sum().backward(retain_graph=True)
    (d_ra * torch.tensor(v_ra, dtype=torch.float64)).sum().backward()
    out = orc.raster_raw_bwd(m2, conics, colors, opac, W, H, TS, offs, flat, ra, last, v_rc, v_ra, backgrounds=bg, absgrad=True)
    for got, ref, name in zip(out[:4], (tm2.grad, tcon.grad, tcol.grad, top.grad), ("v_means2d", "v_conics", "v_colors", "v_opacities")):
        ref = ref.numpy().reshape(got.shape)
        scale = np.abs(ref).max()
        bad = np.abs(got - ref) > (5e-3 * np.abs(ref) + 2e-3 * scale)
        assert bad.mean() < 0.02, (name, bad.mean(), scale)
    assert np.all(out[4] >= np.abs(out[0]) - 1e-6)  # absgrad dominates |grad|


# ----------------------------------------------------------------------------- fused SSIM (ssim.cu) vs float64 conv2d + autograd
def _ssim_torch64(img1, img2, C1, C2):
    import torch.nn.functional as F
    g = torch.tensor([0.001028380123898387, 0.0075987582094967365, 0.036000773310661316, 0.10936068743467331, 0.21300552785396576,
                      0.26601171493530273, 0.21300552785396576, 0.10936068743467331, 0.036000773310661316, 0.0075987582094967365,
                      0.001028380123898387], dtype=torch.float64)
    CH = img1.shape[1]
    win = (g[:, None] * g[None, :]).expand(CH, 1, 11, 11).contiguous()
    conv = lambda t: F.conv2d(t, win, padding=5, groups=CH)
    mu1, mu2 = conv(img1), conv(img2)
    s1, s2, s12 = conv(img1 * img1) - mu1 * mu1, conv(img2 * img2) - mu2 * mu2, conv(img1 * img2) - mu1 * mu2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))


def test_ssim_restatement_matches_float64_conv_and_autograd():
    """orc_ssim_fwd / orc_ssim_bwd vs the textbook definition (zero-padded 11x11 Gaussian window) in float64 with autograd,
    for the full map and for the reference's `padding = "valid"` loss 1 - mean(map[5:-5, 5:-5]) (raw_gs_model.cpp:386-397)."""
    rng = np.random.default_rng(2)
    B, CH, Hs, Ws = 1, 3, 37, 50  # not multiples of any tile
    img2 = rng.uniform(0, 1, (B, CH, Hs, Ws)).astype(np.float32)
    img1 = np.clip(img2 + rng.normal(0, 0.15, img2.shape), 0, 1).astype(np.float32)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m, d1, d2, d3 = orc.ssim_fwd(img1, img2, C1, C2)
    t1 = torch.tensor(img1, dtype=torch.float64, requires_grad=True)
    ref = _ssim_torch64(t1, torch.tensor(img2, dtype=torch.float64), C1, C2)
    np.testing.assert_allclose(m, ref.detach().numpy(), rtol=2e-4, atol=2e-5)
    assert 0.05 < m.mean() < 0.95
    loss = 1.0 - ref[:, :, 5:-5, 5:-5].mean()
    loss.backward()
    dL = np.zeros_like(img1)
    dL[:, :, 5:-5, 5:-5] = -1.0 / (B * CH * (Hs - 10) * (Ws - 10))
    g = orc.ssim_bwd(img1, img2, dL, d1, d2, d3)
    want = t1.grad.numpy()
    np.testing.assert_allclose(g, want, rtol=2e-3, atol=2e-4 * np.abs(want).max())
    # not training: same map, no derivative maps
    m2, n1, n2, n3 = orc.ssim_fwd(img1, img2, C1, C2, train=False)
    assert n1 is None and np.array_equal(m, m2)
    # identical images -> SSIM = 1 everywhere the window is inside; gradient of the map w.r.t. img1 vanishes there
    m3, e1, e2, e3 = orc.ssim_fwd(img2, img2, C1, C2)
    np.testing.assert_allclose(m3, 1.0, atol=1e-5)


def test_ssim_restatement_matches_the_reference_kernels_outputs():
    """tests/golden/ssim_ref_gfx950.npz holds what the REFERENCE's own fusedssimCUDA / fusedssim_backwardCUDA (gsplat/rasterizer/
    ssim.cu compiled for gfx950 by oracle/ref_ssim_build.py, run on an MI355X by tests/golden/make_ssim_ref_golden.py) gave on two
    seeded image pairs: the one part of the splat oracle that is pinned by reference OUTPUT (tolerance = float32 rounding of the
    121-term window sums; the GPU test regenerates the fixture bit for bit)."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ssim_ref_gfx950.npz"))
    C1, C2 = float(np.float32(0.01 * 0.01)), float(np.float32(0.03 * 0.03))
    for tag in ("a", "b"):
        img1, img2, dL = z[tag + "_img1"], z[tag + "_img2"], z[tag + "_dL"]
        outs = orc.ssim_fwd(img1, img2, C1, C2)
        for got, name in zip(outs, ("map", "dm_dmu1", "dm_dsigma1_sq", "dm_dsigma12")):
            want = z[tag + "_" + name]
            np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(want).max()), err_msg=tag + name)
        g = orc.ssim_bwd(img1, img2, dL, z[tag + "_dm_dmu1"], z[tag + "_dm_dsigma1_sq"], z[tag + "_dm_dsigma12"])
        want = z[tag + "_grad"]
        np.testing.assert_allclose(g, want, rtol=1e-3, atol=1e-4 * np.abs(want).max(), err_msg=tag + "grad")

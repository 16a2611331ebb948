"""The kernels bench.py TIMES, pinned to the oracle at BASELINE sizes.

bench.py's optimise iteration is gps_splat_train_step with every Adam step fused into the backward kernel
(fuse_sh_rest_adam = 2): preprocess_fwd_kernel<3> -> binning -> raster_ges_fwd_pk_kernel -> compose_l1_kernel ->
raster_ges_bwd_strip_kernel -> preprocess_bwd_kernel<3> (+ Adam).  This test runs that call for three Adam steps at
640x480 / 200k Gaussians (BASELINE configs[2]) and 1280x720 / 400k (configs[3]) and checks EVERY stage of every step
against the CPU oracle (oracle/splat_oracle.c restating src/raw_gs_model.cpp:188-417 and the gsplat kernels it calls)
on the inputs that stage actually received, plus the Adam update against ATen's op sequence:

  parameters_k --HIP--> per-Gaussian state --HIP--> tile lists --HIP--> render --HIP--> image grads --HIP--> raster
       |  oracle(params_k)      |  oracle(HIP state)     |  oracle(HIP state, lists)     | numpy          grads ...
       +-- compare              +-- bit-exact            +-- compare                     +-- compare

so that one stage's rounding cannot hide in (or be blamed on) another's.  The gradients come from a twin model running
fuse mode 0 (same kernels, gradients written out); the fused-mode model must leave bit-identical parameters.

Accept/reject flips.  A (pixel, Gaussian) pair counts iff alpha = min(0.999, o*exp(-sigma)) >= 1/255 and the Gaussian is
not behind the depth cut.  __expf / exp2-with-folded-opacity round differently from expf, so a pair whose o*exp(-sigma)
lies within 1e-5 (relative) of 1/255 may be decided differently.  Instead of a generic outlier budget, the oracle lists
those borderline pairs (orc_raster_ges_*_flip_budget) and every output element must satisfy
    |hip - oracle| <= 2e-5 * sum|terms of that element| + (what its borderline pairs could contribute);
the number of elements that needed the second term is printed and asserted <= the number of borderline pairs.
"""
import math

import numpy as np
import pytest
import torch

from tests import scenes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REL = 2e-5      # rounding tolerance relative to sum |terms| (exp: ~2e-6 relative, sums of <= a few thousand terms)
BAND = 1e-5     # relative width of the borderline band around 1/255 and around the depth cut


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def N_(t):
    return t.detach().cpu().numpy()


def _build(N, W, H, scale_range, seed, strips=True, intr=None):
    from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
    g = scenes.random_gaussians(N, seed=seed, scale_range=scale_range)
    c2w, K = scenes.default_camera(W, H, seed=seed)
    if intr is not None:   # (fx, fy, cx, cy) instead of the 90-degree pinhole
        K = np.array([[intr[0], 0, intr[2]], [0, intr[1], intr[3]], [0, 0, 1]], np.float32)
    models = []
    for mode in (0, 2):
        m = SLAMGaussianModel(dict(capacity=1 << 19, fuse_sh_rest_adam=mode, strip_backward=strips), device=DEV)
        m.add_params(dict(means=T(g["means"]), scales=T(g["log_scales"]), quats=T(g["quats"]),
                          featuresDc=T(g["sh"][:, 0].copy()), featuresRest=T(g["sh"][:, 1:].copy()),
                          opacities=T(g["opac_logit"])))
        m.initOptimizers(-1, 3.3)
        models.append(m)
    gen = torch.Generator().manual_seed(seed)
    gt = torch.rand((H, W, 3), generator=gen).to(DEV)
    base = torch.rand((H, W, 3), generator=gen).to(DEV)
    ref = (torch.rand((H, W, 1), generator=gen) * 3.5 + 0.5).to(DEV)
    ref[torch.rand((H, W, 1), generator=gen).to(DEV) < 0.1] = 0.0  # raycast misses -> clamped to 1000
    cam = Camera(0, W, H, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), c2w, image=gt, device=DEV)
    return models, cam, ref, base, gt, c2w, K


def _oracle_preprocess(P, vm, K, cam_pos, W, H, max_radii=100, deg=3):
    """RawGaussianModel::gesForward up to the binning (raw_gs_model.cpp:207-286) with the oracle operators."""
    from oracle import splat_ref as orc
    means, ls, quats, dc, rest, ol = P
    scales = np.exp(ls)
    radii, m2, depths, conics = orc.proj_fwd(means, quats, scales, vm, K, W, H)
    radii = np.minimum(radii, max_radii)
    dirs = means - cam_pos[None]
    sh = np.concatenate([dc[:, None], rest], 1)
    rgb = np.maximum(orc.sh_fwd(deg, dirs, sh, radii > 0) + np.float32(0.5), np.float32(0.0))
    colors = np.concatenate([rgb, depths[:, None]], 1).astype(np.float32)
    opac = (np.float32(1.0) / (np.float32(1.0) + np.exp(-ol[:, 0]))).astype(np.float32)
    return radii, m2, depths, conics, colors, opac, scales, dirs, sh


def _oracle_preprocess_bwd(P, vm, K, cam_pos, W, H, radii, conics, v_m2, v_con, v_col, v_op, deg=3):
    """adjoint of the chain above: clamp_min mask -> SH bwd (v_dirs -> v_means) -> projection bwd -> exp / sigmoid."""
    from oracle import splat_ref as orc
    means, ls, quats, dc, rest, ol = P
    scales = np.exp(ls)
    dirs = means - cam_pos[None]
    sh = np.concatenate([dc[:, None], rest], 1)
    vis = radii > 0
    raw = orc.sh_fwd(deg, dirs, sh, vis) + np.float32(0.5)
    v_rgb = np.where(raw >= 0, v_col[:, :3], 0).astype(np.float32)
    v_sh, v_dirs = orc.sh_bwd(deg, dirs, sh, vis, v_rgb)
    v_means, v_quats, v_scales = orc.proj_bwd(means, quats, scales, vm, K, W, H, radii, conics, v_m2,
                                              np.ascontiguousarray(v_col[:, 3]), v_con)
    v_means = v_means + v_dirs
    o = 1.0 / (1.0 + np.exp(-ol[:, 0].astype(np.float64)))
    v_ol = (v_op.astype(np.float64) * o * (1 - o)).astype(np.float32)[:, None]
    return v_means, (v_scales * scales).astype(np.float32), v_quats, v_sh[:, 0], v_sh[:, 1:], v_ol


def _row_rel(got, ref, vis):
    """per-Gaussian error relative to that Gaussian's gradient magnitude (the adjoints cancel internally)"""
    got, ref = got.reshape(got.shape[0], -1)[vis], ref.reshape(ref.shape[0], -1)[vis]
    num = np.abs(got - ref).max(axis=1)
    den = np.abs(ref).max(axis=1) + 1e-6 * np.abs(ref).max()
    return num / den


def _rasterizer_grads(B, N, strips):
    """[N, 10] = v_colors[4] | v_conics[3] | v_means2d[2] | v_opacity of the backward rasterizer: the strip kernel's 48-byte
    rows, or the group kernel's four arrays"""
    if strips:
        return N_(B["v_rows"][:N, :10])
    return np.concatenate([N_(B["v_colors"][:N]), N_(B["v_conics"][:N]), N_(B["v_means2d"][:N]), N_(B["v_opacities"][:N])[:, None]], 1)


# Replica's camera (every configs/release/replica/*.yaml, e.g. office0.yaml:18-20): 1200 x 680, fx = fy = 600, c = (599.5, 339.5) --
# 75 x 43 tiles with a ragged last tile row (680 = 42 x 16 + 8), 3,225 tiles (the superblock scatter's LDS grows with the tile count)
REPLICA = (600.0, 600.0, 599.5, 339.5)


@pytest.mark.parametrize("N,W,H,scale_range,strips,intr", [(200000, 640, 480, (0.003, 0.02), True, None), (400000, 1280, 720, (0.002, 0.011), True, None),
                                                           (200000, 640, 480, (0.003, 0.02), False, None),
                                                           (300000, 1200, 680, (0.002, 0.012), True, REPLICA)],
                         ids=["640x480-200k", "1280x720-400k", "640x480-200k-group-kernel", "replica-1200x680-300k"])
def test_timed_train_step_chain_matches_oracle_at_baseline_sizes(N, W, H, scale_range, strips, intr):
    """strips = True: what bench.py times -- superblock binning (histogram in the preprocessing kernel, scan, scatter + class
    lists) and the column-strip backward; False: the sorted-key binning + 32-pixel-group backward (the operator-level kernels,
    and the train step's path when a host does not provide the strip buffers)."""
    from oracle import splat_ref as orc
    (mA, mB), cam, ref, base, gt, c2w, K = _build(N, W, H, scale_range, seed=N // 1000, strips=strips, intr=intr)
    TS, delta = 16, mA.delta_depth
    tw, th = math.ceil(W / TS), math.ceil(H / TS)
    vm = scenes.pose_inv(c2w)
    cam_pos = c2w[:3, 3].astype(np.float32)
    ref_c = mA.clamp_ref_depth(ref)
    ref_np = N_(ref_c)[..., 0]
    lrs = list(mA._opt["lrs"])  # NAMES order; initOptimizers(-1, 3.3) scales the means lr (float product, as the reference's)
    assert lrs == [float(np.float32(x)) for x in (np.float32(1.6e-4) * np.float32(3.3), 5e-3, 1e-3, 2.5e-3, 5e-4, 5e-2)]
    # ATen reference Adam state (GPU tensors, stepped with the op sequence of torch::optim::Adam::step)
    Pe = [t.clone() for t in mA.opt_gs_params.tensors()]
    Me = [torch.zeros_like(t) for t in Pe]
    Ve = [torch.zeros_like(t) for t in Pe]
    from gps_slam_amd.gs_model import ADAM_BETA1 as b1, ADAM_BETA2 as b2, ADAM_EPS as eps   # the reference's float-derived scalars

    for step in range(1, 4):
        P = [N_(t).copy() for t in mA.opt_gs_params.tensors()]          # parameters_k (NAMES order)
        P_o = (P[0], P[1], P[2], P[3], P[4], P[5])
        for m in (mA, mB):
            m._step_struct(W, H)  # (allocates the persistent buffers on first use)
            m.loss_sum().zero_()
            m.train_step(cam, ref, base, gt, ref_depth_clamped=ref_c)
        torch.cuda.synchronize()
        # ---- the timed (fully fused, mode 2) call against its gradient-writing twin (mode 0).  Same machine code for the
        # gradients and the update (bit-identity on identical inputs: test_fused_adam_is_bit_identical_to_separate_step), but two
        # RUNS of the rasterizer backward differ at the ulp level where a Gaussian's sums are completed with float atomics
        # by >= 3 half-waves, so parameters agree to rounding; an Adam step is lr * m / sqrt(v): a gradient that cancels to
        # ~0 can flip sign between runs and move that element by up to 2 lr -- counted, printed, bounded.
        n_sign = 0
        for a, b, lr in zip(mA.opt_gs_params.tensors(), mB.opt_gs_params.tensors(), lrs):
            d = (a - b).abs()
            off = d > (1e-6 * a.abs() + 1e-3 * lr)
            n_sign += int(off.sum())
            assert float(d.max()) <= 2.001 * lr * step
        assert n_sign <= 1e-5 * 59 * N, n_sign
        if strips:  # no float atomics anywhere in the strip path: the two runs are the same computation
            assert n_sign == 0
            assert all(torch.equal(a, b) for a, b in zip(mA.opt_gs_params.tensors(), mB.opt_gs_params.tensors()))
        assert abs(float(mA.loss_sum()[0]) - float(mB.loss_sum()[0])) <= 1e-5 * float(mA.loss_sum()[0])  # float-atomic sum order
        B = mA._B
        counts = N_(B["counts"])
        ni, ng = int(counts[0]), int(counts[1])
        assert counts[2] == 0, "binning capacity overflow"

        # ---- (a) per-Gaussian preprocessing vs oracle(parameters_k)
        r0, m0, d0, c0, col0, op0, _, _, _ = _oracle_preprocess(P_o, vm, K, cam_pos, W, H)
        r1, m1, c1, col1, op1 = (N_(B[k][:N]) for k in ("radii", "means2d", "conics", "colors", "opacities"))
        both = (r0 > 0) & (r1 > 0)
        n_cull_flip = int(((r0 > 0) != (r1 > 0)).sum())
        n_rad_flip = int((r0 != r1)[both].sum())
        assert both.sum() > 0.5 * N
        edge = scenes.radius_is_borderline(c0) | scenes.radius_is_borderline(c1)   # ceil(3 sqrt(lambda)) decided within rounding
        assert n_cull_flip <= 1e-4 * N and (np.abs(r0 - r1)[both] <= 1).all() and not ((r0 != r1) & both & ~edge).any(), n_rad_flip
        np.testing.assert_allclose(m1[both], m0[both], rtol=1e-4, atol=1e-3)
        cbud = scenes.condition_budget(lambda pm, pls, pq: (_oracle_preprocess((pm, pls, pq, P_o[3], P_o[4], P_o[5]), vm, K, cam_pos, W, H)[3],),
                                       (P_o[0], P_o[1], P_o[2]), (c0,), trials=2)[0]
        assert (np.abs(c1.astype(np.float64) - c0).max(1)[both] <= cbud[both]).all()
        np.testing.assert_allclose(col1[both], col0[both], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(op1, op0, rtol=2e-6)

        # ---- (b) binning of the HIP state: bit-exact at full size
        tpg, ids, flat, ggs, gst, offs = orc.isect_tiles(m1, r1, TS, tw, th)
        assert ni == flat.shape[0]
        assert np.array_equal(N_(B["flatten_ids"][:ni]), flat) and np.array_equal(N_(B["tile_offsets"]), offs.reshape(-1))
        assert np.array_equal(N_(B["tiles_per_gauss"][:N]), tpg)
        if strips:
            # the backward's work lists: per class (smallest 4 << k >= radius; the last class takes the rest) by (image band, id)
            cc = N_(B["cls_counts"])
            lists = scenes.bwd_class_lists(m1, r1, TS, tw, th)
            for k in range(5):
                want = lists[k]
                assert int(cc[k]) == want.shape[0] and np.array_equal(N_(B["cls_ids"][k, :want.shape[0]]), want), k
            assert int(counts[3]) == int((r1 > 0).sum()) and ng == 0
        else:
            assert ng == ggs.shape[0]
            assert np.array_equal(N_(B["group_gs_ids"][:ng]), ggs) and np.array_equal(N_(B["group_starts"][:ng]), gst)

        # ---- (c) forward rasterizer (raster_ges_fwd_pk_kernel) on the HIP state
        e_rc, e_ra, _ = orc.raster_ges_fwd(m1, c1, col1, op1, ref_np, W, H, TS, offs, flat, delta)
        scale_f, _, _ = orc.raster_ges_fwd_flip_budget(m1, c1, col1, op1, ref_np, W, H, TS, offs, flat, delta, rel_band=-1.0)
        flip_f, n_bpairs, n_bpix = orc.raster_ges_fwd_flip_budget(m1, c1, col1, op1, ref_np, W, H, TS, offs, flat, delta,
                                                                  rel_band=BAND)
        got = np.concatenate([N_(B["render_colors"])[0], N_(B["weight_sum"])[0]], -1)
        exp = np.concatenate([e_rc, e_ra[..., None]], -1)
        d = np.abs(got - exp)
        rounding = REL * scale_f + 1e-7
        need_flip = (d > rounding).any(-1)
        excess = d - (rounding + 1.001 * flip_f)
        assert (excess <= 0).all(), ("forward: difference not explained by rounding + borderline pairs", int((excess > 0).sum()),
                                     float(excess.max()), float((d / (scale_f + 1e-30)).max()))
        assert need_flip.sum() <= n_bpix
        print("step %d fwd: I=%d, %d borderline pairs on %d pixels, %d pixels actually flipped, max|d|=%.3g"
              % (step, ni, n_bpairs, n_bpix, int(need_flip.sum()), d.max()))
        assert e_ra.max() > 1.0

        # ---- (d) compose + L1 + image gradients (compose_l1_kernel) on the HIP render
        rc_h, ws_h = N_(B["render_colors"])[0], N_(B["weight_sum"])[0]
        base_n, gt_n = N_(base), N_(gt)
        rgb_e = (rc_h[..., :3] + base_n) / (ws_h + np.float32(1.0))
        loss_e = np.abs(gt_n.astype(np.float64) - rgb_e).mean()
        sgn = -np.sign(gt_n - rgb_e) / np.float32(3 * W * H)           # d loss / d rgb
        v_rc_e = np.concatenate([sgn / (ws_h + 1.0), np.zeros((H, W, 1), np.float32)], -1)
        v_ra_e = -(sgn * rgb_e).sum(-1, keepdims=True) / (ws_h + 1.0)
        np.testing.assert_allclose(N_(B["rgb"]), rgb_e, rtol=2e-6, atol=1e-7)
        assert abs(float(mA.loss_sum()[0]) - loss_e) <= 2e-5 * loss_e
        v_rc_h, v_ra_h = N_(B["v_render_colors"])[0], N_(B["v_render_alphas"])[0]
        np.testing.assert_allclose(v_rc_h, v_rc_e, rtol=1e-5, atol=1e-12)
        np.testing.assert_allclose(v_ra_h, v_ra_e, rtol=1e-4, atol=1e-11)

        # ---- (e) backward rasterizer (raster_ges_bwd_gs_kernel) on the HIP state + HIP image gradients
        e_bwd = orc.raster_ges_bwd_gs(m1, c1, col1, op1, r1, ref_np, W, H, ggs, gst, delta, v_rc_h, v_ra_h[..., 0])
        scale_b, _, _ = orc.raster_ges_bwd_gs_flip_budget(m1, c1, col1, op1, r1, ref_np, W, H, ggs, gst, delta, v_rc_h,
                                                          v_ra_h[..., 0], rel_band=-1.0)
        flip_b, nb_pairs, nb_g = orc.raster_ges_bwd_gs_flip_budget(m1, c1, col1, op1, r1, ref_np, W, H, ggs, gst, delta,
                                                                   v_rc_h, v_ra_h[..., 0], rel_band=BAND)
        got_b = _rasterizer_grads(B, N, strips)
        if strips:  # rows of Gaussians the rasterizer never sees are not written (and not read by the preprocessing backward)
            got_b[r1 <= 0] = 0.0
            assert np.isfinite(got_b).all()
            assert np.array_equal(N_(B["pix2"]).reshape(H, W, 2)[..., 0], v_ra_h[..., 0])
            assert np.array_equal(N_(B["pix2"]).reshape(H, W, 2)[..., 1], ref_np + np.float32(delta))
        exp_b = np.concatenate([e_bwd[2], e_bwd[1], e_bwd[0], e_bwd[3][:, None]], 1)
        db = np.abs(got_b - exp_b)
        rounding_b = REL * scale_b + 1e-30
        need_b = (db > rounding_b).any(-1)
        excess = db - (rounding_b + 1.001 * flip_b)
        bad = np.argwhere(excess > 0)
        assert bad.shape[0] == 0, ("backward: difference not explained by rounding + borderline pairs", bad.shape[0], bad[:5].tolist(),
                                   [(float(db[i, k]), float(scale_b[i, k]), float(flip_b[i, k]), int(r1[i])) for i, k in bad[:5]])
        assert need_b.sum() <= nb_g
        print("step %d bwd: G=%d, %d borderline slots on %d Gaussians, %d Gaussians actually flipped"
              % (step, ggs.shape[0], nb_pairs, nb_g, int(need_b.sum())))

        # ---- (f) preprocessing backward (preprocess_bwd_kernel<3>) on the HIP rasterizer gradients
        e_g = _oracle_preprocess_bwd(P_o, vm, K, cam_pos, W, H, r1, c1, np.ascontiguousarray(got_b[:, 7:9]),
                                     np.ascontiguousarray(got_b[:, 4:7]), np.ascontiguousarray(got_b[:, 0:4]),
                                     np.ascontiguousarray(got_b[:, 9]))
        g_hip = [N_(t) for t in mA.grads()]                               # NAMES: means scales quats dc rest opac
        vis = r1 > 0
        # EVERY Gaussian within its own condition budget: the oracle adjoint's sensitivity to 1-ulp jitter of its float inputs
        # (tests/scenes.py: condition_budget) -- no quantiles, no global slack for the ill-conditioned rows
        fn = lambda pm, pls, pq, pdc, prest, pol, cc, a, b, c, d: _oracle_preprocess_bwd(
            (pm, pls, pq, pdc, prest, pol), vm, K, cam_pos, W, H, r1, cc, a, b, c, d)
        ins = (P_o[0], P_o[1], P_o[2], P_o[3], P_o[4], P_o[5], c1, np.ascontiguousarray(got_b[:, 7:9]), np.ascontiguousarray(got_b[:, 4:7]),
               np.ascontiguousarray(got_b[:, 0:4]), np.ascontiguousarray(got_b[:, 9]))
        budget = scenes.condition_budget(fn, ins, e_g, trials=2)
        # sigmoid': v o (1 - o) in float32 (ATen's sigmoid_backward does the same from the float32 y) loses (1 - o) to cancellation
        # as o -> 1 -- an absolute error of a few ulp(1) on (1 - o) that the oracle's float64 evaluation does not have
        o64 = 1.0 / (1.0 + np.exp(-P_o[5][:, 0].astype(np.float64)))
        budget[5] = budget[5] + 4 * 2.0 ** -23 * np.abs(got_b[:, 9].astype(np.float64) * o64)
        for name, got_g, ref_g, bud in zip(("means", "scales", "quats", "featuresDc", "featuresRest", "opacities"), g_hip,
                                           (e_g[0], e_g[1], e_g[2], e_g[3], e_g[4], e_g[5]), budget):
            assert (got_g.reshape(N, -1)[~vis] == 0).all() or name == "opacities", name
            err = np.abs(got_g.reshape(N, -1).astype(np.float64) - ref_g.reshape(N, -1)).max(1)
            ratio = err[vis] / (bud[vis] + 1e-300)
            if step == 1:
                print("  preprocess bwd %s: max error / budget %.3f" % (name, ratio.max()))
            assert (err[vis] <= bud[vis]).all(), (name, int((err[vis] > bud[vis]).sum()), float(ratio.max()))

        # ---- (g) Adam (fused into the backward kernel in the timed path) vs ATen's op sequence on the HIP gradients
        G = list(mA.grads())
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        for p, g, m, v, lr in zip(Pe, G, Me, Ve, lrs):
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(m, denom, value=-(lr / bc1))
        for a, b in zip(mA._opt["m"], Me):
            assert torch.equal(a[:N], b), "exp_avg must be bit-identical to the ATen op sequence"
        for a, b in zip(mA._opt["v"], Ve):
            assert torch.equal(a[:N], b), "exp_avg_sq must be bit-identical to the ATen op sequence"
        for a, b in zip(mA.opt_gs_params.tensors(), Pe):
            # (a few 1e-3 of the elements differ in the last bit: ATen-on-ROCm's division is not always the IEEE quotient)
            torch.testing.assert_close(a, b, rtol=1e-6, atol=3e-8)  # <= 1 ulp of the update (ATen's non-IEEE division)
            b.copy_(a)  # keep the ATen twin on the HIP trajectory (1-ulp division differences must not accumulate)
        # the fused twin continues from the same state (so that step k+1 compares the kernels, not accumulated ulps)
        for k in ("m", "v"):
            for a, b in zip(mA._opt[k], mB._opt[k]):
                b[:N].copy_(a[:N])
        for a, b in zip(mA.opt_gs_params.tensors(), mB.opt_gs_params.tensors()):
            b.copy_(a)
        print("step %d: %d of %d parameter elements moved by a sign-flipped ~0 gradient between the two runs" % (step, n_sign, 59 * N))

    # ---- end to end, no re-synchronisation: the oracle's own chain from the last parameters reproduces the loss
    P = [N_(t).copy() for t in mB.opt_gs_params.tensors()]
    r0, m0, d0, c0, col0, op0, _, _, _ = _oracle_preprocess((P[0], P[1], P[2], P[3], P[4], P[5]), vm, K, cam_pos, W, H)
    tpg, ids, flat, ggs, gst, offs = orc.isect_tiles(m0, r0, TS, tw, th)
    e_rc, e_ra, _ = orc.raster_ges_fwd(m0, c0, col0, op0, ref_np, W, H, TS, offs, flat, delta)
    rgb_e = (e_rc[..., :3] + N_(base)) / (e_ra[..., None] + np.float32(1.0))
    loss_e = np.abs(N_(gt).astype(np.float64) - rgb_e).mean()
    res = mB.forward(cam, ref, base, ref_depth_clamped=ref_c)
    loss_h = float((gt - res["rgb"]).abs().double().mean())
    assert abs(loss_h - loss_e) <= 1e-4 * loss_e, (loss_h, loss_e)
    psnr = -10.0 * math.log10(float(((res["rgb"].double() - T(rgb_e).double()) ** 2).mean()) + 1e-30)
    print("render PSNR HIP vs oracle chain after 3 steps: %.1f dB" % psnr)
    assert psnr > 60.0


@pytest.mark.parametrize("fuse", [0, 2])
def test_sh_degree_4_preprocess_and_train_step(fuse):
    """SH degree 4 (K = 25): the backward's LDS tiles (72 floats per row) only fit with a smaller workgroup -- the
    preprocessing forward/backward against the oracle and two fused train steps in both Adam modes."""
    from gps_slam_amd import gsplat_ops as ops
    from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
    N, W, H, deg = 20011, 320, 240, 4
    g = scenes.random_gaussians(N, seed=4, scale_range=(0.004, 0.03), sh_k=25)
    c2w, K = scenes.default_camera(W, H, seed=4)
    vm, cam_pos = scenes.pose_inv(c2w), c2w[:3, 3].astype(np.float32)
    P = (g["means"], g["log_scales"], g["quats"], g["sh"][:, 0].copy(), g["sh"][:, 1:].copy(), g["opac_logit"])
    Pt = [T(a) for a in P]
    radii, m2, depths, conics, colors, opac = ops.gauss_preprocess_fwd(Pt[0], Pt[1], Pt[2], Pt[5].view(-1), Pt[3], Pt[4], deg,
                                                                       T(vm), T(K), T(cam_pos), W, H)
    r0, m0, d0, c0, col0, op0, _, _, _ = _oracle_preprocess(P, vm, K, cam_pos, W, H, deg=deg)
    both = (r0 > 0) & (N_(radii) > 0)
    assert both.sum() > 0.5 * N and ((r0 > 0) != (N_(radii) > 0)).sum() <= 1e-3 * N
    np.testing.assert_allclose(N_(colors)[both], col0[both], rtol=1e-4, atol=2e-5)
    rng = np.random.default_rng(0)
    v_m2 = rng.normal(size=(N, 2)).astype(np.float32)
    v_con = (rng.normal(size=(N, 3)) * 0.1).astype(np.float32)
    v_col = rng.normal(size=(N, 4)).astype(np.float32)
    v_op = rng.normal(size=N).astype(np.float32)
    out = ops.gauss_preprocess_bwd(Pt[0], Pt[1], Pt[2], Pt[5].view(-1), Pt[3], Pt[4], deg, T(vm), T(K), T(cam_pos), W, H, 0.3,
                                   radii, conics, T(v_m2), T(v_con), T(v_col), T(v_op))
    e = _oracle_preprocess_bwd(P, vm, K, cam_pos, W, H, N_(radii), N_(conics), v_m2, v_con, v_col, v_op, deg=deg)
    vis = N_(radii) > 0
    # ops order: v_means, v_log_scales, v_quats, v_opac_logit, v_sh_dc, v_sh_rest
    for name, got, ref in zip(("means", "scales", "quats", "opac", "dc", "rest"), out, (e[0], e[1], e[2], e[5][:, 0], e[3], e[4])):
        rel = _row_rel(N_(got).reshape(N, -1), ref.reshape(N, -1), vis)
        assert np.quantile(rel, 0.999) < 5e-3, (name, np.quantile(rel, 0.999))
    model = SLAMGaussianModel(dict(capacity=1 << 15, sh_degree=4, fuse_sh_rest_adam=fuse), device=DEV)
    model.add_params(dict(means=Pt[0], scales=Pt[1], quats=Pt[2], featuresDc=Pt[3], featuresRest=Pt[4], opacities=Pt[5]))
    model.initOptimizers(-1, 1.0)
    gen = torch.Generator().manual_seed(1)
    gt, base = torch.rand((H, W, 3), generator=gen).to(DEV), torch.rand((H, W, 3), generator=gen).to(DEV)
    ref = (torch.rand((H, W, 1), generator=gen) * 4).to(DEV)
    cam = Camera(0, W, H, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), c2w, image=gt, device=DEV)
    before = model.opt_gs_params.featuresRest.clone()
    for _ in range(2):
        model.train_step(cam, ref, base, gt)
    torch.cuda.synchronize()
    after = model.opt_gs_params.featuresRest
    assert torch.isfinite(after).all() and (after != before).any()


@pytest.mark.parametrize("fuse", [0, 1, 2])
def test_first_step_after_init_optimizers_does_not_read_the_moments(fuse):
    """initOptimizers re-creates the Adam state (raw_gs_model.cpp:654-659, every localOptimize); step 1 of every route takes
    exp_avg / exp_avg_sq as zero WITHOUT reading the buffers (include/gps_slam_hip.h, gps_adam_step), so the hosts do not zero
    them: a model whose moment buffers hold NaN after initOptimizers must end three train steps with exactly the parameters and
    moments of one whose buffers hold zeros -- in all three Adam modes, and again after a second initOptimizers."""
    from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
    N, W, H = 30011, 320, 240
    g = scenes.random_gaussians(N, seed=7, scale_range=(0.004, 0.03))
    c2w, K = scenes.default_camera(W, H, seed=7)
    gen = torch.Generator().manual_seed(3)
    gt, base = torch.rand((H, W, 3), generator=gen).to(DEV), torch.rand((H, W, 3), generator=gen).to(DEV)
    ref = (torch.rand((H, W, 1), generator=gen) * 4).to(DEV)
    cam = Camera(0, W, H, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), c2w, image=gt, device=DEV)
    models = []
    for poison in (False, True):
        m = SLAMGaussianModel(dict(capacity=1 << 15, sh_degree=3, fuse_sh_rest_adam=fuse), device=DEV)
        m.add_params(dict(means=T(g["means"]), scales=T(g["log_scales"]), quats=T(g["quats"]), featuresDc=T(g["sh"][:, 0].copy()),
                          featuresRest=T(g["sh"][:, 1:].copy()), opacities=T(g["opac_logit"])))
        for rnd in range(2):
            m.initOptimizers(-1, 1.0)
            if poison:
                for t in m._opt["m"] + m._opt["v"]:
                    t.fill_(float("nan"))
            for _ in range(3):
                m.train_step(cam, ref, base, gt)
        torch.cuda.synchronize()
        models.append(m)
    a, b = models
    for x, y in zip(a.opt_gs_params._buf.values() if isinstance(a.opt_gs_params._buf, dict) else a.opt_gs_params._buf,
                    b.opt_gs_params._buf.values() if isinstance(b.opt_gs_params._buf, dict) else b.opt_gs_params._buf):
        assert torch.equal(x[:N], y[:N])
    for k in ("m", "v"):
        for x, y in zip(a._opt[k], b._opt[k]):
            assert torch.isfinite(y[:N]).all() and torch.equal(x[:N], y[:N])

"""The C++ host layer (gps_slam_amd/host/, module gps_slam_amd._host) against the Python mirror: both sit on the same
C-ABI, so operator outputs, gradients and whole optimise iterations must agree; the reference's call sequence
forward -> computeLoss -> loss.backward() -> optimizersStep() must equal the fused trainStep()."""
import numpy as np
import pytest
import torch

from tests import scenes, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _host():
    import gps_slam_amd._lib as L
    L.load_library()
    import gps_slam_amd._host as h
    return h


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def _scene(N=20000, W=320, H=240, seed=3):
    g = scenes.random_gaussians(N, seed=seed, scale_range=(0.004, 0.03))
    c2w, K = scenes.default_camera(W, H, seed=seed)
    gen = torch.Generator().manual_seed(seed)
    gt = torch.rand((H, W, 3), generator=gen).to(DEV)
    base = torch.rand((H, W, 3), generator=gen).to(DEV)
    ref = (torch.rand((H, W, 1), generator=gen) * 4).to(DEV)
    ref[ref < 0.4] = 0.0
    tensors = [T(g["means"]), T(g["log_scales"]), T(g["quats"]), T(g["sh"][:, 0].copy()), T(g["sh"][:, 1:].copy()),
               T(g["opac_logit"])]
    return tensors, c2w, K, gt, base, ref


def _cpp_model(h, tensors, lr0=False):
    m = h.SLAMGaussianModel()
    cfg = dict(capacity=1 << 16)
    if lr0:
        cfg.update({k: 0.0 for k in ("means_lr", "scales_lr", "quats_lr", "featuresDc_lr", "featuresRest_lr", "opacities_lr")})
    m.loadConfig(cfg)
    m.getGaussianParms().add([t.clone() for t in tensors])
    return m


def _cpp_cam(h, W, H, K, c2w, image):
    cam = h.Camera(W, H, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), True,
                   torch.as_tensor(np.asarray(c2w, np.float32)))
    cam.id = 0
    cam.image = image
    cam.toGPU()
    return cam


def test_operator_surface_matches_python_mirror():
    h = _host()
    from gps_slam_amd import gsplat_wapper as gw
    tensors, c2w, K, gt, base, ref = _scene(N=8000, W=160, H=112)
    W, H = 160, 112
    means, ls, quats, dc, rest, ol = tensors
    from gps_slam_amd.gs_model import pose_inv
    vm = pose_inv(torch.as_tensor(np.asarray(c2w, np.float32))).to(DEV)[None]
    Kt = T(np.asarray(K, np.float32))[None]

    def chain(ffp, shn, isect_fn, rast, leaves):
        m, s, q, d, r, o = leaves
        radii, m2, depths, conics = ffp(m, q, torch.exp(s), vm, Kt)[:4]
        radii = torch.clamp_max(radii, 100)
        shs = torch.cat([d[:, None, :], r], 1)
        dirs = m - T(np.asarray(c2w, np.float32)[:3, 3])[None]
        cols = torch.clamp_min(shn(3, dirs[None], shs[None], radii > 0) + 0.5, 0.0)
        cols = torch.cat([cols, depths.unsqueeze(-1)], 2)
        refc = torch.where(ref < 0.01, torch.full_like(ref, 1000.0), ref)
        return rast(m2, conics, cols, torch.sigmoid(o), radii, refc[None], isect_fn(m2, radii, depths))

    # Python mirror
    lp = [t.clone().requires_grad_(True) for t in tensors]
    tw, th = (W + 15) // 16, (H + 15) // 16
    py_ffp = lambda m, q, s, v, k: gw.FullyFusedProjection.apply(m, None, q, s, v, k, W, H, 0.3, 0.01, 1e10, 0.0, False, "pinhole")
    pstate = {}

    def py_isect(m2, r, d):
        pstate["isect"] = gw.isectTilesNoDepth(m2, r, d, 16, tw, th)
        return pstate["isect"]

    py_rast = lambda m2, c, col, o, r, refc, isect: gw.RasterizeToPixelsGes_NewParallel.apply(
        m2, c, col, o, r, refc, base, None, None, W, H, 16, isect, False, 0.1)
    rc_p, ws_p = chain(py_ffp, gw.SphericalHarmonicsNew.apply, py_isect, py_rast, lp)
    (rc_p.sum() + 0.5 * ws_p.sum()).backward()

    # C++ host
    lc = [t.clone().requires_grad_(True) for t in tensors]
    c_ffp = lambda m, q, s, v, k: h.FullyFusedProjection(m, q, s, v, k, W, H, 0.3, 0.01, 1e10, 0.0)
    state = {}

    def c_isect(m2, r, d):
        tpg, ids, flat, ggs, gst = h.isectTilesNoDepth(m2, r, d, 16, tw, th)
        state.update(ids=ids, flat=flat, ggs=ggs, gst=gst, off=h.isectOffsetEncodeNoDepth(ids, 1, tw, th))
        return state

    c_rast = lambda m2, c, col, o, r, refc, st: h.RasterizeToPixelsGes_NewParallel(
        m2, c, col, o, r, refc, base, W, H, 16, st["off"], st["flat"], st["ggs"], st["gst"], 0.1)
    rc_c, ws_c = chain(c_ffp, h.SphericalHarmonicsNew, c_isect, c_rast, lc)
    (rc_c.sum() + 0.5 * ws_c.sum()).backward()

    assert torch.equal(rc_c, rc_p) and torch.equal(ws_c, ws_p)
    _, ids_p, flat_p, ggs_p, gst_p, off_p = pstate["isect"].trimmed()
    assert torch.equal(state["flat"], flat_p) and torch.equal(state["ggs"], ggs_p) and torch.equal(state["gst"], gst_p)
    assert torch.equal(state["off"].view(-1), off_p.view(-1))
    for a, b in zip(lc, lp):
        # same kernels; the rasterizer backward accumulates with float atomics, so order-level differences only
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-5 * b.grad.abs().max().item())


def test_raw_render_method_cpp_host_matches_python_mirror_and_trains():
    """render_method "raw" (rawForward, raw_gs_model.cpp:43-185): the C++ operator chain isectTiles -> isectOffsetEncode ->
    RasterizeToPixels equals the Python mirror, and the reference's optimise sequence runs through it."""
    h = _host()
    from gps_slam_amd import gsplat_wapper as gw
    from gps_slam_amd.gs_model import pose_inv
    W, H = 160, 112
    tensors, c2w, K, gt, base, ref = _scene(N=8000, W=W, H=H)
    c2w_t = torch.as_tensor(np.asarray(c2w, np.float32))
    cam_dev = dict(viewmat=pose_inv(c2w_t).to(DEV), K=T(np.asarray(K, np.float32)), cam_pos=c2w_t[:3, 3].to(DEV))
    bg = torch.tensor([[0.3, 0.1, 0.2, 0.0]], device=DEV)
    # Python mirror with autograd
    lp = [t.clone().requires_grad_(True) for t in tensors]
    r_p = gw.raw_forward(lp, cam_dev, W, H, backgrounds=bg)
    ((r_p["rgb"] - gt[:H, :W]).abs().mean() + 0.1 * r_p["depth"].mean()).backward()
    # C++ host: rawForward on a model whose leaves are the parameters
    m = _cpp_model(h, tensors, lr0=True)
    m.render_method = "raw"
    m.setBackgrounds(bg)
    cam = _cpp_cam(h, W, H, K, c2w, gt[:H, :W].contiguous())
    m.initOptimizers(-1, 1.0)
    r_c = m.forward(cam, ref[:H, :W].contiguous(), base[:H, :W].contiguous())
    assert torch.equal(r_c["radiis"], r_p["radiis"])
    assert torch.equal(r_c["rgb"], r_p["rgb"]) and torch.equal(r_c["alpha"], r_p["alpha"])
    torch.testing.assert_close(r_c["depth"], r_p["depth"], rtol=1e-6, atol=1e-7)
    assert 0.5 < float(r_c["alpha"].detach().max()) <= 1.0
    ((r_c["rgb"] - gt[:H, :W]).abs().mean() + 0.1 * r_c["depth"].mean()).backward()
    m.optimizersStep()  # lr = 0: parameters unchanged, gradients consumed
    # the C++ leaves are internal; compare through the operator surface instead
    lc = [t.clone().requires_grad_(True) for t in tensors]
    means, ls, quats, dc, rest, ol = lc
    radii, m2, depths, conics = h.FullyFusedProjection(means, quats, torch.exp(ls), cam_dev["viewmat"][None], cam_dev["K"][None],
                                                       W, H, 0.3, 0.01, 1e10, 0.0)[:4]
    shs = torch.cat([dc[:, None, :], rest], 1)
    cols = torch.clamp_min(h.SphericalHarmonicsNew(3, (means - cam_dev["cam_pos"][None])[None], shs[None], radii > 0) + 0.5, 0.0)
    tw, th = (W + 15) // 16, (H + 15) // 16
    tpg, ids, flat = h.isectTiles(m2, radii, depths, 16, tw, th)
    off = h.isectOffsetEncode(ids, 1, tw, th)
    isect_p = gw.isectTiles(m2.detach(), radii, depths.detach(), 16, tw, th)
    n = isect_p.sizes()[0]
    assert torch.equal(ids, isect_p.isect_ids[:n]) and torch.equal(flat, isect_p.flatten_ids[:n])
    assert torch.equal(off, isect_p.isect_offsets) and torch.equal(tpg, isect_p.tiles_per_gauss)
    rc, ra = h.RasterizeToPixels(m2, conics, torch.cat([cols, depths.unsqueeze(-1)], 2), torch.sigmoid(ol), bg, W, H, 16, off,
                                 flat, False)
    assert torch.equal(rc[0, ..., :3], r_p["rgb"])
    depth_c = rc[0, ..., 3:] / ra[0].clamp(1e-10)
    ((rc[0, ..., :3] - gt[:H, :W]).abs().mean() + 0.1 * depth_c.mean()).backward()
    for a, b in zip(lc, lp):
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-5 * b.grad.abs().max().item())

    # the reference's optimise sequence through rawForward lowers the L1 loss
    t_model = _cpp_model(h, tensors)
    t_model.render_method = "raw"
    t_model.initOptimizers(-1, 1.0)
    target = r_p["rgb"].detach() * 0.5 + 0.25  # reachable colours
    cam_t = _cpp_cam(h, W, H, K, c2w, target.contiguous())
    losses = []
    for _ in range(12):
        res = t_model.forward(cam_t, ref[:H, :W].contiguous(), base[:H, :W].contiguous())
        loss = t_model.computeLoss(res, cam_t, dict(l1_weight=1.0))
        loss["loss"].backward()
        t_model.optimizersStep()
        t_model.optimizersZeroGrad()
        losses.append(float(loss["loss"].detach()))
    assert losses[-1] < 0.95 * losses[0] and all(b < a for a, b in zip(losses, losses[1:])), losses


def test_cpp_model_matches_python_model_and_autograd_route_matches_fused():
    h = _host()
    from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
    W, H = 320, 240
    tensors, c2w, K, gt, base, ref = _scene()
    # python model
    pm = SLAMGaussianModel(device=DEV)
    pm.add_params(dict(zip(("means", "scales", "quats", "featuresDc", "featuresRest", "opacities"), [t.clone() for t in tensors])))
    pcam = Camera(0, W, H, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), c2w, image=gt, device=DEV)
    # C++ model
    cm = _cpp_model(h, tensors)
    ccam = _cpp_cam(h, W, H, K, c2w, gt)
    with torch.no_grad():
        r_p = pm.forward(pcam, ref, base)
        r_c = cm.forward(ccam, ref, base)
    torch.testing.assert_close(r_c["rgb"], r_p["rgb"], rtol=1e-5, atol=1e-6)
    assert torch.equal(r_c["radiis"], r_p["radiis"])
    # three fused iterations on each host
    pm.initOptimizers(-1, 3.3)
    cm.initOptimizers(-1, 3.3)
    for _ in range(3):
        pm.train_step(pcam, ref, base, gt)
        cm.trainStep(ccam, ref, base)
    torch.cuda.synchronize()
    torch.testing.assert_close(cm.lossSum(), pm.loss_sum(), rtol=1e-5, atol=0)
    cp = cm.getGaussianParms()
    got = [cp.getMeans(), cp.getScales(), cp.getQuats(), cp.getFeaturesDc(), cp.getFeaturesRest(), cp.getOpacities()]
    for a, b in zip(got, pm.opt_gs_params.tensors()):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)

    # reference call sequence on a fresh C++ model == fused trainStep on another
    a_model, f_model = _cpp_model(h, tensors), _cpp_model(h, tensors)
    a_model.initOptimizers(-1, 3.3)
    f_model.initOptimizers(-1, 3.3)
    for _ in range(2):
        res = a_model.forward(ccam, ref, base)
        loss = a_model.computeLoss(res, ccam, dict(l1_weight=1.0))
        loss["loss"].backward()
        a_model.optimizersStep()
        a_model.optimizersZeroGrad()
        f_model.trainStep(ccam, ref, base)
    torch.cuda.synchronize()
    ap, fp = a_model.getGaussianParms(), f_model.getGaussianParms()
    for name in ("getMeans", "getScales", "getQuats", "getFeaturesDc", "getFeaturesRest", "getOpacities"):
        a, b = getattr(ap, name)(), getattr(fp, name)()
        # Adam normalises the step, so gradient rounding differences (atomics order, compose in torch vs fused) show up
        # at the 1e-3 * lr level at most
        torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-4)


def test_cpp_train_step_run_ahead_equals_plain_steps_and_is_void_after_prune_or_new_pose():
    """RawGaussianModel::trainStep(next_cam): twin C++ models, one running the next camera's preprocessing ahead -- equal
    parameters after every step; a prune between two steps, or a camera whose pose was uploaded again, voids the forward that
    was run ahead (PrefetchKey: Camera::pack_serial + RawGaussianParams::version) and the results still agree."""
    h = _host()
    W, H = 320, 240
    tensors, c2w, K, gt, base, ref = _scene(N=24000, seed=5)
    c2w_b, _ = scenes.default_camera(W, H, seed=6)
    a, b = _cpp_model(h, tensors), _cpp_model(h, tensors)
    a.initOptimizers(-1, 1.0)
    b.initOptimizers(-1, 1.0)
    cams = [_cpp_cam(h, W, H, K, c2w, gt), _cpp_cam(h, W, H, K, c2w_b, gt)]

    def same():
        torch.cuda.synchronize()
        ap, bp = a.getGaussianParms(), b.getGaussianParms()
        assert ap.getGaussianNum() == bp.getGaussianNum()
        for name in ("getMeans", "getScales", "getQuats", "getFeaturesDc", "getFeaturesRest", "getOpacities"):
            assert torch.equal(getattr(ap, name)(), getattr(bp, name)()), name

    seq = [0, 1, 1, 0]
    for it, k in enumerate(seq):
        a.trainStep(cams[k], ref, base, next_cam=cams[seq[it + 1]] if it + 1 < len(seq) else None)
        b.trainStep(cams[k], ref, base)
        same()
    # prune with a forward run ahead outstanding
    a.trainStep(cams[0], ref, base, next_cam=cams[1])
    b.trainStep(cams[0], ref, base)
    delete = torch.zeros(a.getGaussianNum(), dtype=torch.bool, device=DEV)
    delete[::9] = True
    a.prunePoints(delete)
    b.prunePoints(delete)
    a.trainStep(cams[1], ref, base, next_cam=cams[0])
    b.trainStep(cams[1], ref, base)
    same()
    # the camera the forward was run ahead for gets another pose
    moved = torch.as_tensor(np.asarray(c2w, np.float32)).clone()
    moved[:3, 3] += torch.tensor([0.01, -0.02, 0.005])
    cams[0].c2w_slam = moved
    cams[0].invalidate()
    cams[0].toGPU()
    a.trainStep(cams[0], ref, base)
    b.trainStep(cams[0], ref, base)
    same()
    assert a.checkBinningCapacity() == b.checkBinningCapacity()
    # the step buffers are re-created between the two steps (checkBinningCapacity() growing the intersection tables): the forward
    # that was run ahead lived in the old buffers and must not be consumed
    ni, ng = b.checkBinningCapacity()
    tight = dict(capacity=1 << 16, isect_capacity=int(1.5 * max(ni, (ng + 1) // 2)))
    a, b = h.SLAMGaussianModel(), h.SLAMGaussianModel()
    for m in (a, b):
        m.loadConfig(tight)
        m.getGaussianParms().add([t.clone() for t in tensors])
        m.initOptimizers(-1, 1.0)
    a.trainStep(cams[0], ref, base, next_cam=cams[1])
    b.trainStep(cams[0], ref, base)
    assert a.checkBinningCapacity() == b.checkBinningCapacity()   # more than half in use: both grow, the next step re-creates B_
    a.trainStep(cams[1], ref, base, next_cam=cams[0])
    b.trainStep(cams[1], ref, base)
    same()
    a.trainStep(cams[0], ref, base)
    b.trainStep(cams[0], ref, base)
    same()


def test_cpp_pipeline_runs_and_tsdf_state_equals_python_pipeline():
    h = _host()
    from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
    from gps_slam_amd.slam_pipeline import SLAMPipeline
    from gps_slam_amd.tsdf_engine import TsdfEngine
    W, Hh, n = 160, 120, 31
    seq = synth.make_sequence(W, Hh, n, step_deg=0.5)
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)
    rgb = torch.as_tensor(rgba).to(DEV)
    dep = torch.as_tensor(seq["depth"].astype(np.int16)).to(DEV)
    # python host
    eng_p = TsdfEngine(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel_size=0.01, mu=0.04, device=DEV)
    pipe_p = SLAMPipeline(eng_p, SLAMGaussianModel(device=DEV), seed=7)
    # C++ host
    eng_c = h.ITMBasicEngine(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.01, 0.04, 0.2, 10.0)
    model_c = h.SLAMGaussianModel()
    model_c.loadConfig(dict(capacity=1 << 17))
    pipe_c = h.SLAMPipeline(eng_c, model_c, 7)
    for i in range(n):
        img = rgb[i][..., :3].float() / 255.0
        d = (dep[i].float() / 1000.0).unsqueeze(-1)
        cam_p = Camera(i, W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], seq["c2w"][i], image=img, depth=d, device=DEV)
        pipe_p.process_frame(i, cam_p, rgb[i], dep[i])
        cam_c = h.Camera(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][i].astype(np.float32)))
        cam_c.id = i
        cam_c.image, cam_c.depth = img, d
        pipe_c.processFrame(i, cam_c, rgb[i], dep[i])
    torch.cuda.synchronize()
    st = pipe_c.stats()
    assert st["frames"] == n and st["opt_iters"] == 60 and st["raycasts"] == pipe_p.stats["raycasts"]
    # TSDF is independent of the random camera choices: identical state on both hosts
    # (counters[3], the scene's own free-view list length, is only written by the one-view-at-a-time path the Python mirror
    # keeps; the C++ host renders a keyframe's free views as a batch into per-view render states)
    assert torch.equal(eng_c.counters().cpu()[:3], eng_p.counters.cpu()[:3])
    assert torch.equal(eng_c.GetLiveVertex().view(-1), eng_p.raycast.view(-1))
    # same number of Gaussians were sampled (same masks up to the first optimisation, same seeded randperm) and the model
    # is healthy
    assert st["added"] > 100 and model_c.getGaussianNum() > 100
    cams, rcs = pipe_c.optCams(), pipe_c.optRaycasts()
    # the window cameras' raycasts (deterministic head of the list): the C++ host's batched free views against the Python
    # mirror's one-at-a-time free views, bit for bit
    n_win = len(pipe_p.localframe_raycast_window)
    assert n_win >= 1 and [c.id for c in cams[:n_win]] == [c.id for c in pipe_p.opt_cam_list[:n_win]]
    for k in range(n_win):
        for name in ("color_map", "vertex_map", "confidence_map", "depth_map", "depth_map_clamped"):
            assert torch.equal(rcs[k][name], pipe_p.opt_raycast_list[k][name]), (k, name)
    with torch.no_grad():
        res = model_c.forward(cams[0], rcs[0]["depth_map"], rcs[0]["color_map"])
    err_render = (res["rgb"] - cams[0].image).abs().mean().item()
    err_tsdf = (rcs[0]["color_map"] - cams[0].image).abs().mean().item()
    assert err_render <= err_tsdf * 1.02, (err_render, err_tsdf)
    # renderEvalImgs (slam_pipeline.cpp:588-695, tensors instead of image files): raycast + no-grad forward per camera
    res = {k: v.clone() for k, v in res.items()}  # forward() returns views of buffers the next render overwrites
    ev = pipe_c.renderEvalImgs(cams[:2], ["rgb", "alpha", "depth"])
    assert len(ev) == 2 and set(ev[0]) >= {"raycast_color", "raycast_depth", "rgb", "alpha", "depth", "psnr"}
    assert float(ev[0]["rgb"].min()) >= 0.0 and float(ev[0]["rgb"].max()) <= 1.0
    torch.testing.assert_close(ev[0]["rgb"], res["rgb"].clamp(0, 1), rtol=1e-5, atol=1e-6)
    assert 10.0 < float(ev[0]["psnr"]) < 60.0 and not torch.equal(ev[0]["rgb"], ev[1]["rgb"])
    # ---- the eval consumer as the reference measures it (slam_pipeline.cpp:588-695 -> scripts/metric.py): 8-bit images,
    # PSNR of the quantised render against the quantised ground truth -- against the ORACLE's render of the same state
    # (oracle/splat_ref.py: ges_render) quantised the same way.  A pixel channel may differ by one level only where the oracle's
    # 255 * value lies within the float tolerance of the render (5e-4) of an integer: listed, counted, bounded.
    from oracle import splat_ref as orc
    n_ = lambda t: t.detach().cpu().numpy()
    cp = model_c.getGaussianParms()
    Kmat = np.array([[seq["fx"], 0, seq["cx"]], [0, seq["fy"], seq["cy"]], [0, 0, 1]], np.float32)
    for k in range(2):
        cam, e = cams[k], ev[k]
        e_rgb, _ = orc.ges_render(n_(cp.getMeans()), n_(cp.getScales()), n_(cp.getQuats()), n_(cp.getFeaturesDc()), n_(cp.getFeaturesRest()),
                                  n_(cp.getOpacities()), n_(cam.c2w_slam), Kmat, W, Hh, n_(e["raycast_depth"])[..., 0], n_(e["raycast_color"]),
                                  delta_depth=0.1)
        x = np.clip(e_rgb, 0.0, 1.0).astype(np.float32) * np.float32(255.0)
        want = x.astype(np.uint8)                                   # truncation, as toType(kU8)
        got = n_(e["rgb_u8"])
        assert got.dtype == np.uint8 and got.shape == (Hh, W, 3)
        diff = got.astype(np.int32) - want.astype(np.int32)
        borderline = np.abs(x - np.round(x)) < 255.0 * 5e-4          # the oracle's value sits on a quantisation step
        assert (np.abs(diff) <= 1).all() and not (diff != 0)[~borderline].any(), (int((diff != 0).sum()), int(borderline.sum()))
        print("eval camera %d: %d of %d channels on a quantisation step, %d differ by one level" % (k, int(borderline.sum()), diff.size, int((diff != 0).sum())))
        # (the synthetic frames are 8-bit, so wherever the TSDF colour dominates the value sits ON a step: what is bounded is the
        # number of channels that actually land on the other side)
        assert (diff != 0).sum() <= 1e-4 * diff.size
        gt8 = (n_(cam.image) * np.float32(255.0)).astype(np.uint8)
        assert np.array_equal(n_(e["gt_u8"]), gt8)
        # the reference's own PSNR (scripts/utils/image_utils.py:19-21; the helper is pinned by its output in the CPU suite)
        from tests.test_reference_python_pin import psnr_like_the_reference
        psnr_o = psnr_like_the_reference(want / 255.0, gt8 / 255.0)
        assert abs(float(e["psnr_u8"]) - psnr_o) < 0.01, (float(e["psnr_u8"]), psnr_o)       # BASELINE's bar is 0.1 dB
        assert np.array_equal(n_(e["raycast_color_u8"]), (n_(e["raycast_color"]) * np.float32(255.0)).astype(np.uint8))
        d_mm = np.clip(np.rint(n_(e["raycast_depth"]) * np.float32(1000.0)), 0, 65535).astype(np.int32)
        assert np.array_equal(n_(e["raycast_depth_u16"]), d_mm)


def test_sample_method_ours_keeps_a_loss_record_per_keyframe_and_adds_no_history_views():
    """keyframe_sample_configs.sample_method == "ours" as the reference ships it (slam_pipeline.cpp:130-131, 293-317, 538): the
    optimise list holds the local window only (keyFrameRaycast has a "random" branch and nothing else), every keyframe gets
    the record {0.1, frame, 0, 0, 0} when it is added (:355), and checkKeyFrameError() rewrites the records of the history views in
    the list as {loss.total with the raycast's depth > 0 mask, current frame, mean confidence, count of losses above loss_thres}."""
    h = _host()
    W, Hh, n = 160, 120, 21
    seq = synth.make_sequence(W, Hh, n, step_deg=0.5)
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)
    rgb = torch.as_tensor(rgba).to(DEV)
    dep = torch.as_tensor(seq["depth"].astype(np.int16)).to(DEV)
    eng = h.ITMBasicEngine(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.01, 0.04, 0.2, 10.0)
    model = h.SLAMGaussianModel()
    model.loadConfig(dict(capacity=1 << 17))
    pipe = h.SLAMPipeline(eng, model, 7)
    pipe.loadConfig(dict(sample_method="ours", loss_thres=0.01, keyframe_theta_thres=1.0, keyframe_trans_thres=0.02))
    with pytest.raises(Exception):
        pipe.loadConfig(dict(sample_method="nearest"))
    cams = []
    for i in range(n):
        cam = h.Camera(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][i].astype(np.float32)))
        cam.id = i
        cam.image, cam.depth = rgb[i][..., :3].float() / 255.0, (dep[i].float() / 1000.0).unsqueeze(-1)
        pipe.processFrame(i, cam, rgb[i], dep[i])
        cams.append(cam)
    torch.cuda.synchronize()
    st = pipe.stats()
    opt = pipe.optCams()
    assert st["opt_iters"] == 40 and 1 <= len(opt) <= 2             # window views only: no history keyframes were raycast
    rec = pipe.keyframeLossDict()
    assert len(rec) >= 3 and all(v[0] == pytest.approx(0.1) and v[2:] == [0.0, 0.0, 0.0] for v in rec.values()), rec
    assert all(v[1] == float(k) for k, v in rec.items())            # recorded at the frame that became the keyframe
    # a history view by hand: the record becomes the loss of that view
    key_id = sorted(rec)[1]
    kc = cams[key_id]
    kc.toGPU()   # (processFrame sends a COPY of the caller's camera to the device, slam_pipeline.cpp:83-84)
    rc = pipe.runRaycastByCam(kc, False)
    pipe.appendOptView(kc, rc)
    pipe.checkKeyFrameError()
    with torch.no_grad():
        res = model.forward(kc, rc["depth_map"], rc["color_map"])
        m = (rc["depth_map"] > 0).expand_as(res["rgb"])
        want = (kc.image.to(DEV)[m] - res["rgb"][m]).abs().mean().item()
    got = pipe.keyframeLossDict()[key_id]
    assert got[0] == pytest.approx(want, rel=1e-5) and got[1] == float(n - 1)
    assert got[2] == pytest.approx(rc["confidence_map"].mean().item(), rel=1e-6) and got[3] == (1.0 if want > 0.01 else 0.0)
    pipe.checkKeyFrameError()                                      # the count accumulates
    assert pipe.keyframeLossDict()[key_id][3] == (2.0 if want > 0.01 else 0.0)
    untouched = [k for k in rec if k != key_id]
    assert all(pipe.keyframeLossDict()[k] == rec[k] for k in untouched)


def test_full_loop_at_1280x720_cpp_host():
    """BASELINE configs[3] geometry (1280x720, 80x45 = 3600 tiles -> 12 sort bits, 921,600 rays): one keyframe block of the
    whole loop through the C++ host; TSDF state equals the Python host's, the optimised render beats the TSDF colour."""
    h = _host()
    from gps_slam_amd.tsdf_engine import TsdfEngine, pose_from_c2w
    W, Hh, n = 1280, 720, 11
    seq = synth.make_sequence(W, Hh, n, step_deg=0.5)
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)
    rgb = torch.as_tensor(rgba).to(DEV)
    dep = torch.as_tensor(seq["depth"].astype(np.int16)).to(DEV)
    eng_c = h.ITMBasicEngine(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.01, 0.04, 0.2, 10.0)
    model_c = h.SLAMGaussianModel()
    model_c.loadConfig(dict(capacity=1 << 19))
    pipe_c = h.SLAMPipeline(eng_c, model_c, 3)
    eng_p = TsdfEngine(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel_size=0.01, mu=0.04, device=DEV)
    for i in range(n):
        cam = h.Camera(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][i].astype(np.float32)))
        cam.id = i
        cam.image = rgb[i][..., :3].float() / 255.0
        cam.depth = (dep[i].float() / 1000.0).unsqueeze(-1)
        pipe_c.processFrame(i, cam, rgb[i], dep[i])
        eng_p.ProcessFrame(rgb[i], dep[i], seq["c2w"][i])
    torch.cuda.synchronize()
    st = pipe_c.stats()
    assert st["frames"] == n and st["opt_iters"] == 20 and st["added"] > 1000
    assert torch.equal(eng_c.counters().cpu()[:3], eng_p.counters.cpu()[:3])  # [3] = free-view list, pipeline only
    assert torch.equal(eng_c.GetLiveVertex().view(-1), eng_p.raycast.view(-1))
    cams, rcs = pipe_c.optCams(), pipe_c.optRaycasts()
    with torch.no_grad():
        res = model_c.forward(cams[-1], rcs[-1]["depth_map"], rcs[-1]["color_map"])
    assert torch.isfinite(res["rgb"]).all()
    err_render = (res["rgb"] - cams[-1].image).abs().mean().item()
    err_tsdf = (rcs[-1]["color_map"] - cams[-1].image).abs().mean().item()
    assert err_render <= err_tsdf * 1.02, (err_render, err_tsdf)


def test_cpp_engine_tracks_like_the_reference():
    """C++ ITMBasicEngine with tracking left ON (the reference's default; createTsdfEngine switches it off only for
    use_gt_pose): poses against the reference CPU engine's (tests/golden/track_320x240.npz)."""
    import os
    h = _host()
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "track_320x240.npz"))
    W, Hh, n = int(G["W"]), int(G["H"]), int(G["n_frames"])
    seq = synth.make_sequence(W, Hh, n, step_deg=float(G["step_deg"]))
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)
    runs = []
    for riders in (1, 0, 2, 3):   # poses of the LM loop's reject branch riding along with every evaluation (setPosesRidingAlong)
        eng = h.ITMBasicEngine(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], float(G["voxel"]), float(G["mu"]),
                               float(G["vf_min"]), float(G["vf_max"]))
        assert eng.posesRidingAlong() == 1   # the default
        eng.setPosesRidingAlong(riders)
        poses, consumed = [], 0
        for f in range(n):
            eng.ProcessFrame(T(rgba[f]), T(seq["depth"][f].astype(np.int16)))
            pose = eng.lastPose().numpy()
            assert np.abs(pose[0] - G["M"][f]).max() < 2e-5 and np.abs(pose[1] - G["invM"][f]).max() < 2e-5, f
            poses.append(pose.copy())
            rode, used = eng.ridingAlongStats()
            assert 0 <= used <= rode and (riders > 0 or rode == 0)
            consumed += used
        assert eng.trackDiag()[8] > 10000  # inliers of the last accepted evaluation
        assert riders == 0 or not eng.usesBarArgLine() or consumed > 0
        runs.append(np.stack(poses))
    assert all(np.array_equal(runs[0], r) for r in runs[1:])   # the same poses, bit for bit


def test_cpp_engine_mesh_and_state_files_equal_python_host(tmp_path):
    """ITMBasicEngine::SaveSceneToMesh / SaveToFile / LoadFromFile and SLAMPipeline::saveMesh / saveEngine / loadEngine in the
    C++ host write the same bytes as the Python mirror (which tests/test_tsdf_gpu.py compares with the reference's own files)."""
    h = _host()
    from gps_slam_amd.tsdf_engine import TsdfEngine
    W, Hh, n = 96, 72, 3
    seq = synth.make_sequence(W, Hh, n, step_deg=1.0)
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)
    rgb = torch.as_tensor(rgba).to(DEV)
    dep = torch.as_tensor(seq["depth"].astype(np.int16)).to(DEV)
    eng_c = h.ITMBasicEngine(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.01, 0.04, 0.2, 10.0)
    eng_c.turnOffTracking()
    eng_p = TsdfEngine(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel_size=0.01, mu=0.04, device=DEV)
    for i in range(n):
        eng_c.pushGtPose(torch.as_tensor(seq["c2w"][i].astype(np.float32)))
        eng_c.ProcessFrame(rgb[i], dep[i])
        eng_p.ProcessFrame(rgb[i], dep[i], seq["c2w"][i])
    pipe = h.SLAMPipeline(eng_c, h.SLAMGaussianModel(), 1)
    pipe.workspace_dir, pipe.saved_mesh, pipe.saved_engine = str(tmp_path), "c_mesh.ply", "c_state/"
    pipe.saveMesh()
    pipe.saveEngine()
    n_tri = eng_p.SaveSceneToMesh(str(tmp_path / "p_mesh.ply"))
    eng_p.SaveToFile(str(tmp_path / "p_state"))
    assert n_tri > 100000
    assert open(tmp_path / "c_mesh.ply", "rb").read() == open(tmp_path / "p_mesh.ply", "rb").read()
    for name in ("voxel.dat", "alloc.dat", "vba.txt", "hash.dat", "excess.dat", "last.txt"):
        a = np.fromfile(tmp_path / "c_state" / "Scene" / name, np.uint8)
        b = np.fromfile(tmp_path / "p_state" / "Scene" / name, np.uint8)
        assert a.shape == b.shape and bool((a == b).all()), name
    tri_c, counts_c = eng_c.MeshScene(1 << 21)
    # load into a fresh C++ engine through the pipeline: same mesh afterwards
    eng_2 = h.ITMBasicEngine(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.01, 0.04, 0.2, 10.0)
    pipe_2 = h.SLAMPipeline(eng_2, h.SLAMGaussianModel(), 1)
    pipe_2.workspace_dir, pipe_2.saved_engine = str(tmp_path), "c_state/"
    pipe_2.loadEngine()
    tri_2, counts_2 = eng_2.MeshScene(1 << 21)
    assert torch.equal(counts_c, counts_2) and torch.equal(tri_c[:n_tri], tri_2[:n_tri])
    assert torch.equal(eng_2.counters()[:2], eng_c.counters()[:2])


def test_gaussian_model_files_cpp_and_python_hosts(tmp_path):
    """RawGaussianParams::savePly / saveTensor / loadTensor (raw_gs_param.cpp:159-254): the C++ host and the Python mirror write
    the same PLY bytes, the archive round-trips and is readable by torch.jit.load under the reference's keys."""
    h = _host()
    from gps_slam_amd.gs_model import RawGaussianParams, read_gaussian_ply
    tensors, *_ = _scene(N=3000)
    m = _cpp_model(h, tensors)
    p = m.getGaussianParms()
    p.savePly(str(tmp_path / "c.ply"))
    pp = RawGaussianParams(device=DEV, capacity=4096)
    pp.add(dict(zip(RawGaussianParams.NAMES, tensors)))
    pp.savePly(str(tmp_path / "p.ply"))
    assert open(tmp_path / "c.ply", "rb").read() == open(tmp_path / "p.ply", "rb").read()
    back = read_gaussian_ply(str(tmp_path / "c.ply"))
    for name, t in zip(RawGaussianParams.NAMES, tensors):
        assert np.array_equal(back[name].reshape(t.shape), t.cpu().numpy()), name
    p.saveTensor(str(tmp_path / "model.pt"))
    arch = torch.jit.load(str(tmp_path / "model.pt"), map_location="cpu")
    keys = {k for k, _ in arch.named_parameters()} | {k for k, _ in arch.named_buffers()} | set(dir(arch))
    for k in ("means", "scales", "quats", "featuresDc", "featuresRest", "opacities", "exposure"):
        assert k in keys, k
    m2 = h.SLAMGaussianModel()
    m2.loadConfig(dict(capacity=1 << 16))
    m2.getGaussianParms().loadTensor(str(tmp_path / "model.pt"))
    p2 = m2.getGaussianParms()
    assert p2.getGaussianNum() == 3000
    for a, b in zip((p2.getMeans(), p2.getScales(), p2.getQuats(), p2.getFeaturesDc(), p2.getFeaturesRest(), p2.getOpacities()), tensors):
        assert torch.equal(a, b)


def test_compute_loss_with_ssim_and_depth_terms_cpp_equals_python_and_pipeline_trains_with_it():
    """computeLoss with ssim_weight / depth_weight (raw_gs_model.cpp:369-417): FusedSSIMMap in the C++ host equals the Python
    mirror (values and gradients through the ges renderer), and SLAMPipeline runs the reference's optimise sequence when the
    loss has more than the L1 term."""
    h = _host()
    from gps_slam_amd import gsplat_wapper as gw
    W, H = 320, 240
    tensors, c2w, K, gt, base, ref = _scene()
    m = _cpp_model(h, tensors, lr0=True)
    cam = _cpp_cam(h, W, H, K, c2w, gt)
    cam.depth = ref.clone()
    m.initOptimizers(-1, 1.0)
    res = m.forward(cam, ref, base)
    loss = m.computeLoss(res, cam, dict(ssim_weight=0.2, depth_weight=0.1))
    assert set(loss) >= {"total", "rgb", "depth"}
    r2 = dict(rgb=res["rgb"].detach().clone().requires_grad_(True), depth=res["depth"].detach().clone().requires_grad_(True))
    want = gw.compute_loss(r2, gt, gt_depth=ref, has_depth=True, ssim_weight=0.2, depth_weight=0.1)
    for k in ("total", "rgb", "depth"):
        torch.testing.assert_close(loss[k].detach(), want[k].detach(), rtol=1e-5, atol=1e-7)
    # gradient of the C++ loss w.r.t. the rendered image == the mirror's
    g_c = torch.autograd.grad(loss["total"], res["rgb"], retain_graph=True)[0]
    g_p = torch.autograd.grad(want["total"], r2["rgb"])[0]
    torch.testing.assert_close(g_c, g_p, rtol=1e-4, atol=1e-6 * float(g_p.abs().max()) + 1e-12)
    loss["total"].backward()
    m.optimizersStep()  # lr 0: only checks that every parameter received a gradient
    # operator level: planar input goes through the planar layout, same numbers
    a = res["rgb"].detach().permute(2, 0, 1).unsqueeze(0)
    b = gt.permute(2, 0, 1).unsqueeze(0)
    m_cl = h.FusedSSIMMap(1e-4, 9e-4, a, b, "same", False)
    m_pl = h.FusedSSIMMap(1e-4, 9e-4, a.contiguous(), b.contiguous(), "same", False)
    assert torch.equal(m_cl, m_pl)

    # pipeline with an SSIM term: runs the reference sequence and still improves the render
    from gps_slam_amd.tsdf_engine import TsdfEngine  # noqa: F401
    Wp, Hp, n = 160, 120, 11
    seq = synth.make_sequence(Wp, Hp, n, step_deg=0.5)
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)
    rgb = torch.as_tensor(rgba).to(DEV)
    dep = torch.as_tensor(seq["depth"].astype(np.int16)).to(DEV)
    eng = h.ITMBasicEngine(Wp, Hp, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.01, 0.04, 0.2, 10.0)
    model = h.SLAMGaussianModel()
    model.loadConfig(dict(capacity=1 << 16))
    pipe = h.SLAMPipeline(eng, model, 5)
    pipe.loadConfig(dict(ssim_weight=0.2))
    for i in range(n):
        c = h.Camera(Wp, Hp, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][i].astype(np.float32)))
        c.id = i
        c.image = rgb[i][..., :3].float() / 255.0
        c.depth = (dep[i].float() / 1000.0).unsqueeze(-1)
        pipe.processFrame(i, c, rgb[i], dep[i])
    torch.cuda.synchronize()
    st = pipe.stats()
    assert st["opt_iters"] == 20 and model.getGaussianNum() > 50
    cams, rcs = pipe.optCams(), pipe.optRaycasts()
    with torch.no_grad():
        out = model.forward(cams[0], rcs[0]["depth_map"], rcs[0]["color_map"])
    assert (out["rgb"] - cams[0].image).abs().mean().item() <= (rcs[0]["color_map"] - cams[0].image).abs().mean().item() * 1.02


def test_overlapped_mapping_equals_sequential_schedule():
    """SLAMPipeline.overlap_mapping (map update on a second stream while the next frames are tracked and fused) must give the
    results of the sequential schedule: same TSDF state, same Gaussian model (same seeds -> same random choices), same stats."""
    h = _host()
    W, Hh, n = 160, 120, 31
    seq = synth.make_sequence(W, Hh, n, step_deg=0.5)
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)
    rgb = torch.as_tensor(rgba).to(DEV)
    dep = torch.as_tensor(seq["depth"].astype(np.int16)).to(DEV)

    def run(overlap, thread=False, merge=False):
        eng = h.ITMBasicEngine(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.01, 0.04, 0.2, 10.0)
        model = h.SLAMGaussianModel()
        model.loadConfig(dict(capacity=1 << 16))
        pipe = h.SLAMPipeline(eng, model, 11, False)  # tracking on: the latency-bound part that overlaps
        pipe.overlap_mapping = overlap
        pipe.mapping_thread = thread
        pipe.merge_keyframe_raycasts = merge  # an update's free views as one batch instead of window + keyframes
        for i in range(n):
            c = h.Camera(W, Hh, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][i].astype(np.float32)))
            c.id = i
            c.image = rgb[i][..., :3].float() / 255.0
            c.depth = (dep[i].float() / 1000.0).unsqueeze(-1)
            pipe.processFrame(i, c, rgb[i], dep[i])
        pipe.flush()
        torch.cuda.synchronize()
        p = model.getGaussianParms()
        return (pipe.stats(), eng.counters().cpu().clone(), eng.GetLiveVertex().clone(),
                [t.clone() for t in (p.getMeans(), p.getScales(), p.getQuats(), p.getFeaturesDc(), p.getFeaturesRest(), p.getOpacities())])

    st_s, cnt_s, live_s, par_s = run(False)
    # run to run the loop is bit-reproducible (ordered allocation sweeps, stable counting sort, strip backward without atomics, fixed-
    # order tracker sums): the same schedule twice gives the same bits in every parameter
    st_r, cnt_r, live_r, par_r = run(False)
    assert st_r == st_s and torch.equal(cnt_r[:4], cnt_s[:4]) and torch.equal(live_r, live_s)
    assert all(torch.equal(a, b) for a, b in zip(par_s, par_r)), "sequential schedule: two runs differ"
    st_t, _, _, par_t = run(True, True)
    st_t2, _, _, par_t2 = run(True, True)
    assert st_t == st_t2 and all(torch.equal(a, b) for a, b in zip(par_t, par_t2)), "overlap schedule (mapping thread): two runs differ"
    assert all(torch.equal(a, b) for a, b in zip(par_s, par_t)), "overlap schedule != sequential schedule"
    # streams on one host thread; tracking thread + mapping thread; the latter with one free-view batch per update
    for mode in ((True, False), (True, True), (True, True, True)):
        st_o, cnt_o, live_o, par_o = run(*mode)
        assert st_s == st_o and st_s["opt_iters"] == 60 and st_s["added"] > 100
        assert torch.equal(cnt_s[:4], cnt_o[:4]) and torch.equal(live_s, live_o)
        for a, b in zip(par_s, par_o):
            assert a.shape == b.shape
            # same kernels on the same inputs and no float atomics anywhere on the path (round 3's strip backward): the same bits
            assert torch.equal(a, b)

"""End-to-end GPU checks of the host mirror: the fused iteration (gps_splat_train_step) against the reference's
operator chain differentiated by autograd (gsplat_wapper mirror + the libtorch glue of raw_gs_model.cpp), the render-only
forward, and a short run of the whole SLAM loop."""
import numpy as np
import pytest
import torch

from tests import scenes, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model_and_maps(N=20000, W=320, H=240, seed=3):
    from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
    g = scenes.random_gaussians(N, seed=seed, scale_range=(0.004, 0.03))
    c2w, K = scenes.default_camera(W, H, seed=seed)
    model = SLAMGaussianModel(device=DEV)
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    model.add_params(dict(means=T(g["means"]), scales=T(g["log_scales"]), quats=T(g["quats"]),
                          featuresDc=T(g["sh"][:, 0].copy()), featuresRest=T(g["sh"][:, 1:].copy()),
                          opacities=T(g["opac_logit"])))
    gen = torch.Generator().manual_seed(seed)
    gt = torch.rand((H, W, 3), generator=gen).to(DEV)
    base = torch.rand((H, W, 3), generator=gen).to(DEV)
    ref = (torch.rand((H, W, 1), generator=gen) * 4).to(DEV)
    ref[ref < 0.4] = 0.0  # raycast misses
    cam = Camera(0, W, H, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), c2w, image=gt, device=DEV)
    return model, cam, ref, base, gt


@pytest.mark.parametrize("W,H", [(320, 240), (333, 250)])
def test_fused_iteration_matches_autograd_of_operator_chain(W, H):
    from gps_slam_amd import gsplat_wapper as gw
    model, cam, ref, base, gt = _model_and_maps(W=W, H=H)
    p = model.opt_gs_params
    # reference-style: autograd through the operator surface + libtorch glue + L1 (raw_gs_model.cpp:188-417)
    leaves = [t.clone().requires_grad_(True) for t in (p.means, p.scales, p.quats, p.featuresDc, p.featuresRest, p.opacities)]
    out = gw.ges_forward(leaves, cam.toGPU(), cam.width, cam.height, ref, base)
    loss = (gt - out["rgb"]).abs().mean()
    loss.backward()
    # fused path with lr = 0 so that parameters stay put and o["g"] holds the gradients
    model.lrs = {k: 0.0 for k in model.lrs}
    model.initOptimizers(-1, 1.0)
    before = [t.clone() for t in p.tensors()]
    model.train_step(cam, ref, base, gt)
    torch.cuda.synchronize()
    for a, b in zip(before, p.tensors()):
        assert torch.equal(a, b)
    torch.testing.assert_close(model.loss_sum()[0], loss.detach(), rtol=1e-4, atol=0)
    names = ("means", "scales", "quats", "featuresDc", "featuresRest", "opacities")
    for name, g_fused, leaf in zip(names, model.grads(), leaves):
        ref_g = leaf.grad
        scale = ref_g.abs().max().item()
        assert scale > 0, name
        bad = (g_fused - ref_g).abs() > (2e-3 * ref_g.abs() + 1e-3 * scale)
        assert bad.float().mean().item() < 1e-4, (name, bad.float().mean().item())
    # render-only forward == the operator chain's outputs
    res = model.forward(cam, ref, base)
    torch.testing.assert_close(res["rgb"], out["rgb"].detach(), rtol=1e-5, atol=1e-6)
    m = torch.isfinite(out["depth"].detach())
    torch.testing.assert_close(res["depth"][m], out["depth"].detach()[m], rtol=1e-5, atol=1e-6)
    assert torch.equal(res["radiis"], out["radiis"])


@pytest.mark.parametrize("W,H", [(1600, 720), (4112, 48)], ids=["4500-tiles", "257-tiles-wide"])
def test_fused_iteration_outside_the_superblock_binnings_limits_takes_the_sorted_key_route_and_matches_autograd(W, H):
    """What gps::sb_supported() rejects must run, not fail: more than SB_MAX_TILES (4,096) tiles (100 x 45) -- the LDS histograms
    do not cover the tile ids -- or a grid more than 255 tiles wide (257 x 3: the scatter packs a box's width into 8 bits).  The
    fused step falls back to the sorted-key binning + group backward (splat_step.hip: strips_on) with the strip buffers still
    allocated -- same gradients as autograd through the operator chain, and the same step as a model built without the strip buffers."""
    from gps_slam_amd import gsplat_wapper as gw
    from gps_slam_amd.gs_model import SLAMGaussianModel
    model, cam, ref, base, gt = _model_and_maps(N=20000, W=W, H=H, seed=11)
    p = model.opt_gs_params
    leaves = [t.clone().requires_grad_(True) for t in (p.means, p.scales, p.quats, p.featuresDc, p.featuresRest, p.opacities)]
    out = gw.ges_forward(leaves, cam.toGPU(), cam.width, cam.height, ref, base)
    loss = (gt - out["rgb"]).abs().mean()
    loss.backward()
    twin = SLAMGaussianModel(dict(strip_backward=False), device=DEV)
    twin.add_params({k: getattr(p, k).clone() for k in ("means", "scales", "quats", "featuresDc", "featuresRest", "opacities")})
    for m in (model, twin):
        m.lrs = {k: 0.0 for k in m.lrs}
        m.initOptimizers(-1, 1.0)
        m.train_step(cam, ref, base, gt)
    torch.cuda.synchronize()
    assert model.strip_backward and not twin.strip_backward
    torch.testing.assert_close(model.loss_sum()[0], loss.detach(), rtol=1e-4, atol=0)
    for name, g_fused, g_twin, leaf in zip(("means", "scales", "quats", "featuresDc", "featuresRest", "opacities"), model.grads(),
                                           twin.grads(), leaves):
        scale = leaf.grad.abs().max().item()
        bad = (g_fused - leaf.grad).abs() > (2e-3 * leaf.grad.abs() + 1e-3 * scale)
        assert scale > 0 and bad.float().mean().item() < 1e-4, (name, bad.float().mean().item())
        # the group backward accumulates shared Gaussians with float atomics: equal up to summation order
        torch.testing.assert_close(g_fused, g_twin, rtol=1e-4, atol=1e-5 * scale)
    assert int(model._B["counts"][2]) == 0 and int(model._B["counts"][1]) > 0   # no overflow; the GROUP table was built


def test_forward_launch_order_is_a_permutation_by_list_length_and_changes_nothing():
    """The superblock binning leaves the forward rasterizer's launch order in the workspace (tiles by descending list length,
    64 classes of 32 entries): a permutation of the tile ids, non-increasing in the class of the tile's list length; the ordered
    launch writes the same image, bit for bit, as the row-major one."""
    import ctypes as C
    from gps_slam_amd._lib import lib
    W, H = 640, 480
    model, cam, ref, base, gt = _model_and_maps(N=60000, W=W, H=H, seed=5)
    model.initOptimizers(-1, 1.0)
    model.train_step(cam, ref, base, gt)
    torch.cuda.synchronize()
    B, st = model._B, model._step
    T = ((W + 15) // 16) * ((H + 15) // 16)
    p = lib.gps_isect_workspace_tile_order(C.c_void_p(B["workspace"].data_ptr()), st.N, st.isect_capacity)
    off = p - B["workspace"].data_ptr()
    assert 0 < off < B["workspace"].numel()
    order = B["workspace"][off:off + 4 * T].view(torch.int32).cpu().numpy()
    assert sorted(order.tolist()) == list(range(T))
    offs = B["tile_offsets"][:T].cpu().numpy().astype(np.int64)
    n_isects = int(B["counts"][0])
    length = np.diff(np.concatenate([offs, [n_isects]]))
    cls = np.minimum(63, length[order] >> 5)
    assert (np.diff(cls) <= 0).all() and cls[0] > cls[-1]
    ptr = lambda t: C.c_void_p(t.data_ptr())
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    refc = model.clamp_ref_depth(ref)
    outs = []
    for o in (C.c_void_p(0), C.c_void_p(p)):
        rc, ws = torch.zeros_like(B["render_colors"]), torch.zeros_like(B["weight_sum"])
        assert lib.gps_raster_ges_fwd_rec_ordered(st.N, ptr(B["records"]), ptr(refc), W, H, ptr(B["tile_offsets"]), ptr(B["flatten_ids"]),
                                                  ptr(B["counts"]), model.delta_depth, ptr(rc), ptr(ws), o, sp) == 0
        outs.append((rc, ws))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and outs[0][1].abs().sum() > 0


def _two_cameras(W, H, seed):
    from gps_slam_amd.gs_model import Camera
    cams = []
    for k in range(2):
        c2w, K = scenes.default_camera(W, H, seed=seed + k)
        cams.append(Camera(k, W, H, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), c2w, device=DEV))
    return cams


@pytest.mark.parametrize("N,W,H", [(20000, 320, 240), (70001, 640, 480)])
def test_next_iterations_preprocess_in_the_backward_kernel_changes_nothing(N, W, H):
    """gps_splat_step::next_viewmat: the NEXT iteration's preprocessing forward runs in the tail of this iteration's backward +
    Adam kernel (on the parameters that kernel has just stepped) and the next call skips its preprocessing launch.  Six
    iterations alternating between two cameras, with and without it, on twin models: every per-Gaussian intermediate of the last
    forward, the binning's lists, the parameters and both Adam moments after every step are EQUAL (same arithmetic on the same
    values; N = 70,001: a last workgroup with one Gaussian, two workgroups of the backward kernel per 256-Gaussian binning block)."""
    from gps_slam_amd.gs_model import SLAMGaussianModel
    from gps_slam_amd._lib import lib
    import ctypes as C
    g = scenes.random_gaussians(N, seed=21, scale_range=(0.004, 0.03))
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    models = []
    for _ in range(2):
        m = SLAMGaussianModel(dict(fuse_sh_rest_adam=2), device=DEV)
        m.add_params(dict(means=T(g["means"]), scales=T(g["log_scales"]), quats=T(g["quats"]), featuresDc=T(g["sh"][:, 0].copy()),
                          featuresRest=T(g["sh"][:, 1:].copy()), opacities=T(g["opac_logit"])))
        m.initOptimizers(-1, 1.0)
        models.append(m)
    cams = _two_cameras(W, H, seed=4)
    gen = torch.Generator().manual_seed(9)
    gts = [torch.rand((H, W, 3), generator=gen).to(DEV) for _ in range(2)]
    base = torch.rand((H, W, 3), generator=gen).to(DEV)
    ref = (torch.rand((H, W, 1), generator=gen) * 4).to(DEV)
    seq = [0, 1, 0, 0, 1, 0]
    a, b = models
    held = None
    for it, k in enumerate(seq):
        nxt = cams[seq[it + 1]] if it + 1 < len(seq) else None
        a.train_step(cams[k], ref, base, gts[k], next_cam=nxt)
        if nxt is not None:
            assert a._prefetched is not None and lib.gps_splat_can_prefetch(C.byref(a._step)) == 1
        b.train_step(cams[k], ref, base, gts[k])
        torch.cuda.synchronize()
        if it > 0:
            assert int(a._step.preprocessed) == 1 and int(b._step.preprocessed) == 0
        for name in ("means", "scales", "quats", "featuresDc", "featuresRest", "opacities"):
            assert torch.equal(getattr(a.opt_gs_params, name), getattr(b.opt_gs_params, name)), (it, name)
        for k2 in range(6):
            assert torch.equal(a._opt["m"][k2][:N], b._opt["m"][k2][:N]) and torch.equal(a._opt["v"][k2][:N], b._opt["v"][k2][:N]), (it, k2)
        # the per-Gaussian intermediates: `a` holds the NEXT iteration's already (written by this step's backward kernel), `b` gets
        # them at the start of its next step -- compare a's of step it - 1 with b's of step it
        inter = ("radii", "means2d", "depths", "conics", "colors", "opacities", "records", "tiles_per_gauss")
        if it > 0:
            for name in inter:
                assert torch.equal(held[name], b._B[name][:N]), (it, name)
        held = {name: a._B[name][:N].clone() for name in inter}
        ni = int(a._B["counts"][0])
        assert ni == int(b._B["counts"][0]) and torch.equal(a._B["flatten_ids"][:ni], b._B["flatten_ids"][:ni]), it
        assert torch.equal(a._B["render_colors"], b._B["render_colors"]), it
    torch.testing.assert_close(a.loss_sum(), b.loss_sum(), rtol=1e-5, atol=0)   # (a sum of per-tile float atomics: order-dependent)
    # anything else on the step buffers disarms the prefetch: a render in between, then a step -> the step preprocesses itself
    # (and returns the binning's tables to zero -- the forward that was run ahead had added its counts to them)
    a.train_step(cams[0], ref, base, gts[0], next_cam=cams[1])
    ra = a.forward(cams[1], ref, base)["rgb"].clone()
    a.train_step(cams[1], ref, base, gts[1])
    assert int(a._step.preprocessed) == 0
    b.train_step(cams[0], ref, base, gts[0])
    rb = b.forward(cams[1], ref, base)["rgb"].clone()
    b.train_step(cams[1], ref, base, gts[1])
    torch.cuda.synchronize()
    assert torch.equal(ra, rb)
    for name in ("means", "scales", "quats", "featuresDc", "featuresRest", "opacities"):
        assert torch.equal(getattr(a.opt_gs_params, name), getattr(b.opt_gs_params, name)), name
    assert int(a._B["counts"][2]) == 0 and int(b._B["counts"][2]) == 0   # no binning overflow on either side


def test_a_forward_run_ahead_is_void_after_a_structure_edit_or_a_new_pose():
    """The key of a forward run ahead (gs_model.train_step / RawGaussianModel::trainStep) holds the camera's upload serial and the
    parameter container's version: a prune + add that leaves N unchanged, a prune that changes it, and a camera whose pose was
    re-uploaded (possibly to the same address) all void it -- the next step preprocesses itself, the binning's tables are back to
    zero, and the twin model that never prefetches ends with the same bits."""
    from gps_slam_amd.gs_model import SLAMGaussianModel
    N, W, H = 30000, 320, 240
    g = scenes.random_gaussians(N, seed=33, scale_range=(0.004, 0.03))
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    rows = dict(means=T(g["means"]), scales=T(g["log_scales"]), quats=T(g["quats"]), featuresDc=T(g["sh"][:, 0].copy()),
                featuresRest=T(g["sh"][:, 1:].copy()), opacities=T(g["opac_logit"]))
    models = []
    for _ in range(2):
        m = SLAMGaussianModel(dict(fuse_sh_rest_adam=2), device=DEV)
        m.add_params(rows)
        m.initOptimizers(-1, 1.0)
        models.append(m)
    a, b = models
    cams = _two_cameras(W, H, seed=6)
    gen = torch.Generator().manual_seed(10)
    gts = [torch.rand((H, W, 3), generator=gen).to(DEV) for _ in range(2)]
    base = torch.rand((H, W, 3), generator=gen).to(DEV)
    ref = (torch.rand((H, W, 1), generator=gen) * 4).to(DEV)

    def same():
        torch.cuda.synchronize()
        assert a.getGaussianNum() == b.getGaussianNum()
        n = a.getGaussianNum()
        for name in ("means", "scales", "quats", "featuresDc", "featuresRest", "opacities"):
            assert torch.equal(getattr(a.opt_gs_params, name), getattr(b.opt_gs_params, name)), name
        for k in range(6):
            assert torch.equal(a._opt["m"][k][:n], b._opt["m"][k][:n]) and torch.equal(a._opt["v"][k][:n], b._opt["v"][k][:n]), k
        assert torch.equal(a._B["render_colors"], b._B["render_colors"])
        assert int(a._B["counts"][2]) == 0 and int(b._B["counts"][2]) == 0

    def edit(m, delete, extra):
        m.prunePoints(delete)
        if extra is not None:
            m.add_params(extra)
            n = m.getGaussianNum()
            for k in ("m", "v"):   # the appended rows start with zero moments on both twins
                for t in m._opt[k]:
                    t[n - extra["means"].shape[0]:n].zero_()

    # 1. prune 500 + add 500: N unchanged, different rows
    delete = torch.zeros(N, dtype=torch.bool, device=DEV)
    delete[torch.arange(0, 500 * 7, 7, device=DEV)] = True
    extra = {k: v[1000:1500].clone() for k, v in rows.items()}
    a.train_step(cams[0], ref, base, gts[0], next_cam=cams[1])
    b.train_step(cams[0], ref, base, gts[0])
    edit(a, delete, extra); edit(b, delete, extra)
    assert a.getGaussianNum() == N and a._prefetched is not None
    a.train_step(cams[1], ref, base, gts[1], next_cam=cams[0])
    assert int(a._step.preprocessed) == 0
    b.train_step(cams[1], ref, base, gts[1])
    same()
    # 2. prune only (N changes) with a forward run ahead outstanding
    delete = torch.zeros(N, dtype=torch.bool, device=DEV)
    delete[-700:] = True
    edit(a, delete, None); edit(b, delete, None)
    a.train_step(cams[0], ref, base, gts[0], next_cam=cams[1])
    assert int(a._step.preprocessed) == 0
    b.train_step(cams[0], ref, base, gts[0])
    same()
    # 3. the next camera's pose is re-uploaded between the two steps
    cams[1].c2w_slam = cams[1].c2w_slam.clone()
    cams[1].c2w_slam[:3, 3] += torch.tensor([0.01, -0.02, 0.005])
    cams[1].invalidate()
    a.train_step(cams[1], ref, base, gts[1])
    assert int(a._step.preprocessed) == 0
    b.train_step(cams[1], ref, base, gts[1])
    same()
    # ... and an undisturbed pair of steps still takes the forward that was run ahead
    a.train_step(cams[0], ref, base, gts[0], next_cam=cams[1])
    b.train_step(cams[0], ref, base, gts[0])
    a.train_step(cams[1], ref, base, gts[1])
    assert int(a._step.preprocessed) == 1
    b.train_step(cams[1], ref, base, gts[1])
    same()


def test_optimisation_reduces_the_loss():
    model, cam, ref, base, gt = _model_and_maps(N=30000, seed=5)
    model.initOptimizers(-1, 3.3)
    losses = []
    for it in range(30):
        model.train_step(cam, ref, base, gt)
        if it in (0, 29):
            torch.cuda.synchronize()
            losses.append(float(model.loss_sum()[0]))
        model.loss_sum().zero_()
    assert losses[1] < 0.97 * losses[0], losses


def test_slam_loop_runs_end_to_end():
    """SLAMTrainCams on a small synthetic sequence: TSDF every frame, Gaussian block every 10 frames."""
    from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
    from gps_slam_amd.slam_pipeline import SLAMPipeline
    from gps_slam_amd.tsdf_engine import TsdfEngine
    W, H, n = 160, 120, 31
    seq = synth.make_sequence(W, H, n, step_deg=0.5)
    eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel_size=0.01, mu=0.04, device=DEV)
    model = SLAMGaussianModel(device=DEV)
    pipe = SLAMPipeline(eng, model, seed=7)
    rgb = torch.as_tensor(seq["rgb"]).to(DEV)
    dep = torch.as_tensor(seq["depth"].astype(np.int16)).to(DEV)
    for i in range(n):
        cam = Camera(i, W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], seq["c2w"][i], image=rgb[i].float() / 255.0,
                     depth=(dep[i].float() / 1000.0).unsqueeze(-1), device=DEV)
        pipe.process_frame(i, cam, rgb[i], dep[i])
    torch.cuda.synchronize()
    assert pipe.stats["frames"] == n and pipe.stats["opt_iters"] == 60 and pipe.stats["raycasts"] >= 6
    assert model.getGaussianNum() > 100
    for t in model.opt_gs_params.tensors():
        assert torch.isfinite(t).all()
    # the composed render of the last optimisation camera is at least as close to the image as the raycast colour
    cam, rc = pipe.opt_cam_list[0], pipe.opt_raycast_list[0]
    res = model.forward(cam, rc["depth_map"], rc["color_map"])
    err_render = (res["rgb"] - cam.image).abs().mean().item()
    err_tsdf = (rc["color_map"] - cam.image).abs().mean().item()
    assert err_render <= err_tsdf * 1.02, (err_render, err_tsdf)


def test_knn_and_normal_map_match_torch_formulations():
    from gps_slam_amd.gs_model import knn_mean_dist2
    from gps_slam_amd.slam_pipeline import compute_normal_map
    gen = torch.Generator().manual_seed(0)
    for P in (1, 3, 4, 257, 5000):
        pts = torch.rand((P, 3), generator=gen).to(DEV)
        got = knn_mean_dist2(pts)
        d = torch.cdist(pts.double(), pts.double()).pow(2)
        d.fill_diagonal_(float("inf"))
        k = min(3, P - 1)
        ref = torch.topk(d, k, dim=1, largest=False).values.sum(1) / 3.0 if k > 0 else torch.zeros(P, device=DEV)
        if P >= 4:
            torch.testing.assert_close(got.double(), ref, rtol=1e-4, atol=1e-9)
        else:
            assert (got > 1e30).all()  # FLT_MAX terms, as in simple_knn.cu:156-187 -> clamped by max_init_scale
    H, W = 120, 160
    v = (torch.rand((H, W, 3), generator=gen) * 2 - 0.5).to(DEV)
    got = compute_normal_map(v)
    wx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=torch.float32, device=DEV).view(1, 1, 3, 3)
    wy = torch.tensor([[-1, -2, -1], [0, 0, 0], [1, 2, 1]], dtype=torch.float32, device=DEV).view(1, 1, 3, 3)
    x = torch.nn.functional.pad(v.permute(2, 0, 1).reshape(-1, 1, H, W), (1, 1, 1, 1), mode="replicate")
    dx = torch.nn.functional.conv2d(x, wx).squeeze(1).permute(1, 2, 0)
    dy = torch.nn.functional.conv2d(x, wy).squeeze(1).permute(1, 2, 0)
    n = torch.cross(dy.reshape(-1, 3), dx.reshape(-1, 3), dim=-1).view(H, W, 3)
    n = n / (torch.norm(n, 2, -1, True) + 1e-8)
    n = torch.where((v[:, :, 2] <= 0).unsqueeze(-1), torch.zeros_like(n), n)
    torch.testing.assert_close(got, n, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("W,H", [(320, 240), (333, 250)])
def test_fused_binning_and_class_lists_survive_add_and_prune(W, H):
    """The fused iteration's superblock binning keeps tables in the caller's workspace ACROSS launches (zero between them)
    while the Gaussian count changes with every prune / add: after each change the tile lists, tile offsets and the backward's
    class lists must still be the oracle's for the state of that launch.  (333 x 250: ragged last tile column and row.)"""
    from oracle import splat_ref as orc
    model, cam, ref, base, gt = _model_and_maps(W=W, H=H)
    model.initOptimizers(-1, 1.0)
    W, H = cam.width, cam.height
    tw, th = (W + 15) // 16, (H + 15) // 16
    g = torch.Generator().manual_seed(3)
    removed = None
    for round_ in range(4):
        model.train_step(cam, ref, base, gt)
        torch.cuda.synchronize()
        B, N = model._B, model.getGaussianNum()
        m2, r = B["means2d"][:N].cpu().numpy(), B["radii"][:N].cpu().numpy()
        tpg, ids, flat, ggs, gst, offs = orc.isect_tiles(m2, r, 16, tw, th)
        counts = B["counts"].cpu().numpy()
        assert int(counts[0]) == flat.shape[0] and int(counts[2]) == 0 and int(counts[3]) == int((r > 0).sum()), (round_, counts)
        assert np.array_equal(B["flatten_ids"][:flat.shape[0]].cpu().numpy(), flat), round_
        assert np.array_equal(B["tile_offsets"].cpu().numpy(), offs.reshape(-1)), round_
        cc = B["cls_counts"].cpu().numpy()
        lists = scenes.bwd_class_lists(m2, r, 16, tw, th)
        for k in range(5):
            want = lists[k]
            assert int(cc[k]) == want.shape[0] and np.array_equal(B["cls_ids"][k, :want.shape[0]].cpu().numpy(), want), (round_, k)
        p = model.opt_gs_params
        if round_ % 2 == 0:   # prune a random third ...
            mask = (torch.rand(N, generator=g) < 0.33).to(DEV)
            removed = {n: getattr(p, n)[mask].clone() for n in p.NAMES}
            model.prunePoints(mask)
            assert model.getGaussianNum() < N
        else:                 # ... and put them back behind the rest
            model.add_params(removed)
            assert model.getGaussianNum() == N + removed["means"].shape[0]


@pytest.mark.parametrize("case", ["no gaussians", "all behind the camera", "all outside the image"])
def test_a_model_nothing_of_which_reaches_the_image_renders_the_base_colour(case):
    """Edge cases of gesForward (raw_gs_model.cpp:188-367): with no Gaussian at all (the pipeline's state before the first
    keyframe), with every Gaussian behind the camera (culled by the near plane: radius 0, no tile) or projected outside the
    image (no tile), the tile lists are empty, the render is (0 + base) / (0 + 1) = the base colour EXACTLY, the weight sum 0,
    the loss mean |gt - base|; a train step runs to the end, leaves every parameter finite and -- no gradient reaches anything --
    only moves parameters by Adam's zero-gradient step (exactly nothing: m = v = 0)."""
    from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
    W, H = 130, 70   # (ragged: 8.1 x 4.4 tiles)
    c2w, K = scenes.default_camera(W, H, seed=9)
    model = SLAMGaussianModel(dict(capacity=1 << 12), device=DEV)
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    if case != "no gaussians":
        g = scenes.random_gaussians(500, seed=9, scale_range=(0.004, 0.03))
        means = g["means"].copy()
        if case == "all behind the camera":
            means[:, 2] = -np.abs(means[:, 2]) - 1.0      # the camera sits near the origin and looks down +z
        else:
            means[:, 0] += 1.0e3                          # far off to the side: projected centre thousands of pixels outside
        model.add_params(dict(means=T(means), scales=T(g["log_scales"]), quats=T(g["quats"]), featuresDc=T(g["sh"][:, 0].copy()),
                              featuresRest=T(g["sh"][:, 1:].copy()), opacities=T(g["opac_logit"])))
    gen = torch.Generator().manual_seed(9)
    gt = torch.rand((H, W, 3), generator=gen).to(DEV)
    base = torch.rand((H, W, 3), generator=gen).to(DEV)
    ref = (torch.rand((H, W, 1), generator=gen) * 4).to(DEV)
    cam = Camera(0, W, H, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), c2w, image=gt, device=DEV)
    res = model.forward(cam, ref, base)
    torch.cuda.synchronize()
    assert torch.equal(res["rgb"], base), case
    assert float(res["alpha"].abs().max()) == 0.0
    if model.getGaussianNum() > 0:
        assert int((res["radiis"] > 0).sum()) == 0
        before = [t.clone() for t in model.opt_gs_params.tensors()]
        model.initOptimizers(-1, 1.0)
        model._step_struct(W, H)
        model.loss_sum().zero_()
        model.train_step(cam, ref, base, gt)
        torch.cuda.synchronize()
        want = float((gt - base).abs().double().mean())
        assert abs(float(model.loss_sum()[0]) - want) <= 1e-5 * want
        for a, b in zip(model.opt_gs_params.tensors(), before):
            assert torch.isfinite(a).all() and torch.equal(a, b), case

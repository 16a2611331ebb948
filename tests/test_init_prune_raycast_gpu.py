"""Direct tests of the three host-side rows SURVEY 8(a) lists next to the kernels, each against a torch restatement of the
reference's own libtorch sequence:

  a17  runRaycastByCam + ITMU*ImageToTensor  (slam/slam_pipeline.cpp:362-415, src/cv_utils.cpp:322-341)  gps_raycast_to_maps
  a10  RawGaussianParams::init / add + SLAMGaussianModel::addGaussians  (src/raw_gs_param.cpp:11-74, 123-145,
       src/tensor_math.cpp:184-201, slam/slam_gs_model.cpp:5-56)
  a11  removeRedundantGs / prunePoints  (slam/slam_pipeline.cpp:564-586, src/raw_gs_model.cpp:635-644)
Both hosts (C++ gps_slam_amd._host and the Python mirror) are checked.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _host():
    import gps_slam_amd._lib as L
    L.load_library()
    import gps_slam_amd._host as h
    return h


# ------------------------------------------------------------------------------------------------ a17
def _ref_raycast_maps(rays, colour, voxel_size, c2w):
    """slam_pipeline.cpp:386-403 + cv_utils.cpp:322-341 + tensor_math.cpp:56-81, op for op in torch"""
    color = colour.clone().to(torch.float32).div(255.0)[..., :3].contiguous()               # ITMUChar4ImageToTensor
    value, confidence = rays[..., :3], rays[..., 3:4]
    vc = torch.cat([value * confidence.gt(0), confidence], 2).contiguous()                  # ITMUFloat4ImageToTensor
    vertex = vc[..., :3].contiguous() * voxel_size
    conf = vc[..., 3:4].contiguous()
    R, T = c2w[:3, :3], c2w[:3, 3:4]                                                        # poseInv
    w2c = torch.eye(4, dtype=c2w.dtype)
    w2c[:3, :3] = R.t()
    w2c[:3, 3:4] = torch.matmul(-R.t(), T)
    w2c = w2c.to(rays.device)
    H, W = vertex.shape[:2]
    hom = torch.ones((H * W, 4), device=rays.device)                                        # verticesTransform
    hom[:, :3] = vertex.reshape(-1, 3)
    t = w2c.matmul(hom.t()).t()
    t = (t[:, :3] / t[:, 3:4]).reshape(H, W, 3)
    depth = t[..., 2].unsqueeze(-1).contiguous()
    depth = depth.masked_fill((vertex.sum(2) == 0).unsqueeze(-1), 0)
    return color, vertex, conf, depth


def test_raycast_to_maps_matches_the_reference_tensor_sequence():
    from gps_slam_amd._lib import check, lib
    H, W, voxel = 120, 160, 0.005
    gen = torch.Generator().manual_seed(0)
    rays = (torch.rand((H, W, 4), generator=gen) * 800 - 100).to(DEV)
    rays[..., 3] = torch.where(torch.rand((H, W), generator=gen).to(DEV) < 0.2, torch.zeros(1, device=DEV),
                               torch.rand((H, W), generator=gen).to(DEV) * 50 + 1)          # w = confidence + 1, 0 / <0 = miss
    rays[5, 7, 3] = -1.0
    rays[9, 9, :3] = 0.0                                                                    # a hit whose xyz sums to 0 -> depth 0
    colour = torch.randint(0, 256, (H, W, 4), generator=gen, dtype=torch.uint8).to(DEV)
    c2w = torch.as_tensor(synth.make_sequence(16, 12, 3, step_deg=2.0)["c2w"][2].astype(np.float32))
    e_color, e_vertex, e_conf, e_depth = _ref_raycast_maps(rays, colour, voxel, c2w)
    out = [torch.empty((H, W, k), device=DEV) for k in (3, 3, 1, 1, 1)]
    R, T = c2w[:3, :3], c2w[:3, 3:4]
    w2c = torch.eye(4)
    w2c[:3, :3], w2c[:3, 3:4] = R.t(), torch.matmul(-R.t(), T)
    w2c = np.ascontiguousarray(w2c.numpy().astype(np.float32))
    p = lambda t: C.c_void_p(t.data_ptr())
    check(lib.gps_raycast_to_maps(W, H, p(rays), p(colour), voxel, w2c.ctypes.data, p(out[0]), p(out[1]), p(out[2]), p(out[3]),
                                  p(out[4]), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gps_raycast_to_maps")
    assert torch.equal(out[0], e_color)                     # uchar * (1/255): ATen's division by a host scalar
    assert torch.equal(out[1], e_vertex)                    # xyz zeroed where w <= 0, then * voxel_size
    assert torch.equal(out[2], e_conf)
    assert (out[1][5, 7] == 0).all() and (out[1][rays[..., 3] <= 0] == 0).all()
    torch.testing.assert_close(out[3], e_depth, rtol=2e-6, atol=2e-6)   # fma chain vs the [4 x P] matmul
    assert (out[3][(e_vertex.sum(2) == 0)] == 0).all() and out[3][9, 9, 0] == 0
    assert torch.equal(out[4], torch.where(out[3] < 0.01, torch.full_like(out[3], 1000.0), out[3]))  # raw_gs_model.cpp:205-207


@pytest.mark.parametrize("which", ["cpp", "python"])
def test_run_raycast_by_cam_uses_stored_pose_for_rays_and_dataset_pose_for_depth(which):
    """Trap: the raycast uses the engine's stored pose of cam.id (slam_pipeline.cpp:367-371) while depth_map is the z under
    poseInv(cam.c2w) -- the DATASET pose, not c2w_slam (:398)."""
    h = _host()
    W, H, n = 160, 120, 4
    seq = synth.make_sequence(W, H, n, step_deg=1.0)
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    if which == "cpp":
        eng = h.ITMBasicEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.01, 0.04, 0.2, 10.0)
        model = h.SLAMGaussianModel()
        model.loadConfig(dict(capacity=1 << 12))
        pipe = h.SLAMPipeline(eng, model, 1)
        pipe.work_mode = "recon"
        for i in range(n):
            c = h.Camera(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][i].astype(np.float32)))
            c.id = i
            pipe.processFrame(i, c, T(rgba[i]), T(seq["depth"][i].astype(np.int16)))
        other = torch.as_tensor(seq["c2w"][3].astype(np.float32))
        cam = h.Camera(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, other)   # dataset pose of frame 3 ...
        cam.id = 1                                                                       # ... but the id of frame 1
        m = pipe.runRaycastByCam(cam, False)
        eng.runRaycastC2w(torch.as_tensor(seq["c2w"][1].astype(np.float32)))            # rays of the STORED pose of id 1
        rays, colour = eng.GetFreeVertex().view(H, W, 4).clone(), eng.GetFreeImage().view(H, W, 4).clone()
        voxel = eng.getVoxelSize()
    else:
        from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
        from gps_slam_amd.slam_pipeline import SLAMPipeline
        from gps_slam_amd.tsdf_engine import TsdfEngine
        eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel_size=0.01, mu=0.04, device=DEV)
        pipe = SLAMPipeline(eng, SLAMGaussianModel(dict(capacity=1 << 12), device=DEV), work_mode="recon")
        for i in range(n):
            c = Camera(i, W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], seq["c2w"][i], device=DEV)
            pipe.process_frame(i, c, T(rgba[i]), T(seq["depth"][i].astype(np.int16)))
        other = torch.as_tensor(seq["c2w"][3].astype(np.float32))
        cam = Camera(1, W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], seq["c2w"][3], device=DEV)
        m = pipe.runRaycastByCam(cam)
        eng.runRaycast(c2w=seq["c2w"][1])
        rays, colour = eng.fv_raycast.view(H, W, 4).clone(), eng.fv_colour.view(H, W, 4).clone()
        voxel = eng.getVoxelSize()
    e_color, e_vertex, e_conf, e_depth = _ref_raycast_maps(rays, colour, voxel, other)
    assert (e_conf > 0).float().mean() > 0.5
    assert torch.equal(m["color_map"], e_color) and torch.equal(m["vertex_map"], e_vertex)
    assert torch.equal(m["confidence_map"], e_conf)
    torch.testing.assert_close(m["depth_map"], e_depth, rtol=2e-6, atol=2e-6)


# ------------------------------------------------------------------------------------------------ a10
def _ref_init(xyz, rgb, normals, max_sh_degree=3, init_opacs=0.5, max_scale=0.01, min_scale=-1.0):
    """RawGaussianParams::init (raw_gs_param.cpp:11-74) + computeQuat / quaternionFromAxisAngle (tensor_math.cpp:184-201),
    op for op; distCUDA2 restated as the exact 3-NN mean of squared distances (simple_knn.cu:191-240)."""
    P = xyz.shape[0]
    d = torch.cdist(xyz.double(), xyz.double()).pow(2)
    d.fill_diagonal_(float("inf"))
    knn = (torch.topk(d, 3, dim=1, largest=False).values.sum(1) / 3.0).float()
    raw_scales = torch.sqrt(knn).clamp(min_scale, max_scale).unsqueeze(1).repeat(1, 3)
    raw_scales[:, 2] = raw_scales[:, 2] * 0.1
    z_axis = torch.zeros_like(raw_scales)
    z_axis[:, 2] = 1
    axis = torch.cross(z_axis, normals, dim=1)
    axis = axis / (torch.norm(axis, 2, -1, True) + 1e-8)
    angle = torch.acos(torch.sum(z_axis * normals, 1)).unsqueeze(-1)
    naxis = axis / (torch.norm(axis, 2, -1, True) + 1e-8)
    quats = torch.cat([torch.cos(angle / 2), naxis * torch.sin(angle / 2)], 1)
    K = (1, 4, 9, 16, 25)[max_sh_degree]
    shs = torch.zeros((P, K, 3), device=xyz.device)
    shs[:, 0, :3] = (rgb - 0.5) / 0.28209479177387814
    opac = torch.logit(init_opacs * torch.ones((P, 1), device=xyz.device))
    return [xyz, raw_scales.log(), quats, shs[:, 0, :], shs[:, 1:, :], opac]


def _points(P, seed):
    gen = torch.Generator().manual_seed(seed)
    xyz = (torch.rand((P, 3), generator=gen) * 0.2).to(DEV)          # dense enough that some KNN scales hit the 0.01 clamp
    xyz[: P // 2] *= 0.05                                             # ... and some do not
    rgb = torch.rand((P, 3), generator=gen).to(DEV)
    nrm = torch.randn((P, 3), generator=gen).to(DEV)
    nrm = nrm / nrm.norm(dim=1, keepdim=True)
    nrm[0] = torch.tensor([0.0, 0.0, 1.0])                            # parallel to z: zero axis, angle 0 -> identity rotation
    nrm[1] = torch.tensor([0.0, 0.0, -1.0])                           # anti-parallel: zero axis, angle pi
    return xyz, rgb, nrm


@pytest.mark.parametrize("which", ["cpp", "python"])
def test_init_params_match_reference_restatement(which):
    h = _host()
    xyz, rgb, nrm = _points(3000, seed=1)
    exp = _ref_init(xyz, rgb, nrm)
    if which == "cpp":
        got = h.RawGaussianParamsMake(xyz, rgb, nrm, 3, 0.5, 0.01, -1.0)
    else:
        from gps_slam_amd.gs_model import SLAMGaussianModel
        d = SLAMGaussianModel(dict(capacity=1 << 12), device=DEV).init_params(xyz, rgb, nrm)
        got = [d[k] for k in ("means", "scales", "quats", "featuresDc", "featuresRest", "opacities")]
    names = ("means", "scales", "quats", "featuresDc", "featuresRest", "opacities")
    for name, a, b in zip(names, got, exp):
        assert a.shape == b.shape, name
    assert torch.equal(got[0], exp[0])
    torch.testing.assert_close(got[1], exp[1], rtol=1e-5, atol=1e-6)       # log(sqrt(knn)): knn kernel vs float64 cdist
    assert (exp[1][:, 0].exp() < 0.0099).any() and (exp[1][:, 0].exp() > 0.00999).any()  # both sides of the max_scale clamp
    torch.testing.assert_close(got[1][:, 2], got[1][:, 0] + float(np.log(np.float32(0.1))), rtol=0, atol=2e-6)  # z axis x 0.1
    torch.testing.assert_close(got[2], exp[2], rtol=1e-5, atol=1e-6)
    assert torch.allclose(got[2][0], torch.tensor([1.0, 0.0, 0.0, 0.0], device=DEV), atol=1e-6)
    assert torch.equal(got[3], exp[3]) and torch.equal(got[4], exp[4]) and (got[4] == 0).all()
    assert torch.equal(got[5], exp[5]) and (got[5] == 0).all()             # logit(0.5)


def _surface_points(W, H, seed):
    """W x H / 4-ish samples of the synthetic room's surfaces as a first keyframe adds them: back-projected depth of one view
    (tests/synth.py), every pixel -> W * H points in pixel order (the caller subsamples)"""
    from tests import synth
    seq = synth.make_sequence(W, H, 1, step_deg=0.5)
    d = torch.as_tensor(seq["depth"][0].astype(np.float32) / 1000.0)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    pc = torch.stack([(xs - seq["cx"]) / seq["fx"] * d, (ys - seq["cy"]) / seq["fy"] * d, d], -1).reshape(-1, 3)
    c2w = torch.as_tensor(seq["c2w"][0])
    pw = pc @ c2w[:3, :3].T + c2w[:3, 3]
    gen = torch.Generator().manual_seed(seed)
    return pw[d.reshape(-1) > 0], gen


def _knn3_float64(x, chunk=4096):
    """mean of the three smallest squared distances to OTHER points, float64, chunked on the device"""
    xd = x.double()
    out = torch.empty(x.shape[0], dtype=torch.float64, device=x.device)
    for lo in range(0, x.shape[0], chunk):
        d2 = torch.cdist(xd[lo:lo + chunk], xd) ** 2
        d2[torch.arange(d2.shape[0], device=x.device), torch.arange(lo, lo + d2.shape[0], device=x.device)] = float("inf")
        out[lo:lo + chunk] = torch.topk(d2, 3, dim=1, largest=False).values.sum(1) / 3.0
    return out


@pytest.mark.parametrize("P,W,H", [(76800, 640, 480), (230400, 1280, 720)])
def test_grid_knn_is_exact_on_a_first_keyframe_of_surface_points(P, W, H):
    """distCUDA2 at the size a first keyframe / a newly revealed room adds (0.25 x W x H surface samples): the uniform-grid
    search (gps_knn_mean_dist2_grid) returns the tiled brute force's numbers BIT FOR BIT, and both agree with a float64 cdist;
    timed: the grid search is the sub-quadratic one (reported, asserted < 2 ms at 76,800 points)."""
    from gps_slam_amd.gs_model import knn_mean_dist2
    pts, gen = _surface_points(W, H, seed=P)
    sel = torch.randperm(pts.shape[0], generator=gen)[:P].sort().values       # a random quarter of the pixels, pixel order
    x = pts[sel].contiguous().to(DEV)
    assert x.shape[0] == P
    grid = knn_mean_dist2(x, method="grid")
    brute = knn_mean_dist2(x, method="brute")
    assert torch.equal(grid, brute)
    ref = _knn3_float64(x)
    torch.testing.assert_close(grid.double(), ref, rtol=2e-5, atol=1e-12)
    assert torch.equal(knn_mean_dist2(x), grid)                                # the default picks the grid above 8,192 points
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(5):
        knn_mean_dist2(x, method="grid")
    ev[1].record()
    knn_mean_dist2(x, method="brute")
    ev[2].record()
    torch.cuda.synchronize()
    t_grid, t_brute = ev[0].elapsed_time(ev[1]) / 5, ev[1].elapsed_time(ev[2])
    print("P = %d: grid %.3f ms, brute force %.1f ms" % (P, t_grid, t_brute))
    if P == 76800:
        assert t_grid < 2.0, t_grid


@pytest.mark.parametrize("case", ["cube", "plane", "duplicates", "outlier", "line", "tiny", "two_clusters", "nan_inf"])
def test_grid_knn_equals_the_brute_force_on_awkward_sets(case):
    """shapes that stress the grid: uniform volume, an exactly planar set (one axis of the bounding box is zero), many exact
    duplicates (zero distances, one crowded cell), a dense cluster with one far outlier (the outlier's rings cross the whole
    grid), a line, fewer than four points (a FLT_MAX term, as the brute force and the reference), two far clusters (empty middle), NaN / inf coordinates."""
    from gps_slam_amd.gs_model import knn_mean_dist2
    gen = torch.Generator().manual_seed(11)
    P = 6000
    x = torch.rand((P, 3), generator=gen)
    if case == "plane":
        x[:, 2] = 0.25
    elif case == "duplicates":
        x[: P // 2] = x[P // 2:P // 2 * 2][torch.randint(0, 20, (P // 2,), generator=gen)]
    elif case == "outlier":
        x *= 0.01
        x[17] = torch.tensor([5.0, -3.0, 2.0])
    elif case == "line":
        x[:, 1] = x[:, 0] * 0.5
        x[:, 2] = -x[:, 0]
    elif case == "tiny":
        x = x[:3]
    elif case == "two_clusters":
        x *= 0.02
        x[P // 2:] += torch.tensor([4.0, 4.0, -4.0])
    elif case == "nan_inf":   # non-finite points: no neighbour of anything, their own result inf -- and they must not decide the grid
        x[5] = float("nan"); x[77, 1] = float("inf"); x[4000] = torch.tensor([float("-inf"), 0.5, float("nan")])
    x = x.contiguous().to(DEV)
    grid, brute = knn_mean_dist2(x, method="grid"), knn_mean_dist2(x, method="brute")
    assert torch.equal(grid, brute), (case, (grid != brute).sum())
    if case == "nan_inf":
        assert torch.isinf(grid[[5, 77, 4000]]).all() and torch.isfinite(grid).sum() == P - 3
        return
    if case == "tiny":
        assert (grid > 1e37).all()     # two real distances + one FLT_MAX term, / 3 (simple_knn.cu:186: best[] starts at FLT_MAX)
    else:
        torch.testing.assert_close(grid.double(), _knn3_float64(x), rtol=2e-5, atol=1e-12)


@pytest.mark.parametrize("which", ["cpp", "python"])
def test_add_gaussians_samples_masked_pixels_and_appends(which):
    """slam_gs_model.cpp:5-56: masked_select of vertex / colour / normal maps, a random subset of floor(n * ratio) of them,
    init(), append after the existing Gaussians.  With ratio 1 the subset is everything -> fully deterministic comparison
    (this repository appends the subset in pixel order, LABBOOK.md section 4 "Order of new Gaussians"); with ratio 0.25 the count and membership are checked."""
    h = _host()
    H, W = 48, 64
    gen = torch.Generator().manual_seed(3)
    vertex = (torch.rand((H, W, 3), generator=gen) * 0.3).to(DEV)
    normal = torch.randn((H, W, 3), generator=gen).to(DEV)
    normal = normal / normal.norm(dim=2, keepdim=True)
    image = torch.rand((H, W, 3), generator=gen).to(DEV)
    mask = (torch.rand((H, W, 1), generator=gen) < 0.3).to(DEV)
    n = int(mask.sum())
    sel = mask.expand(H, W, 3)
    verts, cols, norms = (torch.masked_select(t, sel).reshape(-1, 3) for t in (vertex, image, normal))
    xyz0, rgb0, nrm0 = _points(500, seed=2)
    first = _ref_init(xyz0, rgb0, nrm0)
    if which == "cpp":
        model = h.SLAMGaussianModel()
        model.loadConfig(dict(capacity=1 << 13))
        model.getGaussianParms().add(first)
        cam = h.Camera(W, H, 50.0, 50.0, 32.0, 24.0, True, torch.eye(4))
        cam.image = image
        add = lambda ratio: model.addGaussians(cam, dict(vertex_map=vertex, normal_map=normal), mask, ratio, 10)
        params = lambda: [getattr(model.getGaussianParms(), k)() for k in ("getMeans", "getScales", "getQuats", "getFeaturesDc",
                                                                          "getFeaturesRest", "getOpacities")]
    else:
        from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
        model = SLAMGaussianModel(dict(capacity=1 << 13), device=DEV)
        model.add_params(dict(zip(("means", "scales", "quats", "featuresDc", "featuresRest", "opacities"), first)))
        cam = Camera(0, W, H, 50.0, 50.0, 32.0, 24.0, np.eye(4, dtype=np.float32), image=image, device=DEV)
        g = torch.Generator().manual_seed(5)
        add = lambda ratio: model.addGaussians(cam, dict(vertex_map=vertex, normal_map=normal), mask, ratio, 10, generator=g)
        params = lambda: model.opt_gs_params.tensors()
    assert add(1.0) == n and model.getGaussianNum() == 500 + n
    exp_new = _ref_init(verts, cols, norms)
    for k, (a, f, e) in enumerate(zip(params(), first, exp_new)):
        assert torch.equal(a[:500], f), k                                  # RawGaussianParams::add: cat after the old rows
        if k in (1, 2):
            torch.testing.assert_close(a[500:], e, rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(a[500:], e), k
    k25 = add(0.25)
    assert k25 == int(n * 0.25) and model.getGaussianNum() == 500 + n + k25
    new_means = params()[0][500 + n:]
    # every sampled mean is one of the masked vertices, no vertex twice, pixel (= ascending index) order
    idx = (new_means[:, None, :] == verts[None, :, :]).all(-1).float().argmax(1)
    assert torch.equal(verts[idx], new_means) and idx.unique().numel() == k25 and (idx[1:] > idx[:-1]).all()


# ------------------------------------------------------------------------------------------------ a11
@pytest.mark.parametrize("which", ["cpp", "python"])
def test_remove_redundant_gs_thresholds_order_and_adam_state(which):
    """slam_pipeline.cpp:564-586 (max real scale < small or > large, real opacity < low) + raw_gs_model.cpp:635-644 /
    removeFromOptimizer: the survivors keep their order, and every survivor's Adam state rows stay with it."""
    h = _host()
    N = 4000
    gen = torch.Generator().manual_seed(9)
    xyz, rgb, nrm = _points(N, seed=4)
    P = _ref_init(xyz, rgb, nrm)
    P[4] = torch.randn(P[4].shape, generator=gen).to(DEV) * 0.1
    small, large, low = 0.003, 0.1, 0.005
    ls = torch.log(torch.rand((N, 3), generator=gen) * 0.02 + 0.004).to(DEV)      # all inside (small, large) ...
    ls[10] = torch.log(torch.tensor([0.001, 0.002, 0.0029]))                      # max < small            -> removed
    ls[11] = torch.log(torch.tensor([0.001, 0.002, 0.0031]))                      # max just above small   -> kept
    ls[12] = torch.log(torch.tensor([0.001, 0.2, 0.002]))                         # max > large            -> removed
    ls[13] = torch.log(torch.tensor([0.001, 0.0999, 0.002]))                      # max just below large   -> kept
    P[1] = ls
    ol = torch.full((N, 1), 0.3, device=DEV)
    ol[20] = torch.logit(torch.tensor(0.004))                                     # opacity < low          -> removed
    ol[21] = torch.logit(torch.tensor(0.006))                                     # kept
    ol[12] = torch.logit(torch.tensor(0.001))                                     # removed for two reasons at once
    rnd = torch.rand(N, generator=gen).to(DEV) < 0.1
    rnd[[10, 11, 12, 13, 20, 21]] = False
    ol[rnd] = torch.logit(torch.tensor(0.002))                                    # + ~10 % low-opacity rows
    P[5] = ol
    smax = P[1].exp().max(-1).values
    exp_mask = (smax < small) | (smax > large) | (torch.sigmoid(P[5]).squeeze(-1) < low)
    assert exp_mask[[10, 12, 20]].all() and not exp_mask[[11, 13, 21]].any()
    keep = ~exp_mask
    tag = torch.arange(N, device=DEV, dtype=torch.float32)
    if which == "cpp":
        model = h.SLAMGaussianModel()
        model.loadConfig(dict(capacity=1 << 13))
        model.getGaussianParms().add(P)
        model.initOptimizers(-1, 1.0)
        state = model.adamState()
        for t in state:                                                           # recognisable per-row Adam state
            t.copy_((tag.view(-1, *([1] * (t.dim() - 1))) + 0.25).expand_as(t))
        eng = h.ITMBasicEngine(32, 24, 20.0, 20.0, 16.0, 12.0, 0.02, 0.08, 0.2, 10.0)
        pipe = h.SLAMPipeline(eng, model, 1)
        pipe.small_scale_thres, pipe.large_scale_thres, pipe.low_opac_thres = small, large, low
        pipe.removeRedundantGs()
        got = [getattr(model.getGaussianParms(), k)() for k in ("getMeans", "getScales", "getQuats", "getFeaturesDc",
                                                               "getFeaturesRest", "getOpacities")]
        got_state = model.adamState()
        pruned = pipe.stats()["pruned"]
    else:
        from gps_slam_amd.gs_model import SLAMGaussianModel
        from gps_slam_amd.slam_pipeline import SLAMPipeline
        from gps_slam_amd.tsdf_engine import TsdfEngine
        model = SLAMGaussianModel(dict(capacity=1 << 13), device=DEV)
        model.add_params(dict(zip(("means", "scales", "quats", "featuresDc", "featuresRest", "opacities"), P)))
        model.initOptimizers(-1, 1.0)
        for k in ("m", "v"):
            for t in model._opt[k]:
                t[:N].copy_((tag.view(-1, *([1] * (t.dim() - 1))) + 0.25).expand_as(t[:N]))
        eng = TsdfEngine(32, 24, 20.0, 20.0, 16.0, 12.0, voxel_size=0.02, mu=0.08, device=DEV)
        pipe = SLAMPipeline(eng, model, pipe_cfg=dict(small_scale_thres=small, large_scale_thres=large, low_opac_thres=low))
        pipe.removeRedundantGs()
        got = model.opt_gs_params.tensors()
        M = model.getGaussianNum()
        got_state = [t[:M] for t in model._opt["m"]] + [t[:M] for t in model._opt["v"]]
        pruned = pipe.stats["pruned"]
    M = int(keep.sum())
    assert model.getGaussianNum() == M and pruned == N - M and 0.05 * N < N - M < 0.2 * N
    for a, b in zip(got, P):
        assert torch.equal(a, b[keep])                                            # stable order, every tensor
    for t in got_state:
        assert torch.equal(t.reshape(M, -1)[:, 0], tag[keep] + 0.25)              # Adam rows followed their Gaussians


# ------------------------------------------------------------------------------------------------ sampling kernels
@pytest.mark.parametrize("W,H,with_alpha", [(640, 480, True), (50, 37, False), (1280, 720, True)])
def test_new_gaussian_mask_compaction_and_gather_match_the_tensor_sequence(W, H, with_alpha):
    """gps_new_gaussian_mask == the reference's mask expression (slam_pipeline.cpp:455-480) evaluated with torch ops on the
    GPU (bit for bit: same float sequence); gps_compact_mask == masked_select's row-major order; gps_gather_pixels == the
    index_select of a subset of those rows (slam_gs_model.cpp:14-33)."""
    from gps_slam_amd._lib import lib
    g = torch.Generator(device=DEV).manual_seed(W + H)
    P = W * H
    R = lambda *s: torch.rand(*s, device=DEV, generator=g)
    depth = R(H, W, 1) * 8.0
    src, image = R(H, W, 3), R(H, W, 3)
    # values that make the error land ON the threshold for some pixels, and exact zeros in the vertex sum
    image[::7, ::5] = src[::7, ::5] + 0.1
    vertex = R(H, W, 3) - 0.5
    vertex[::3, ::4] = 0.0
    vertex[1::9, 2::6, 2] = -(vertex[1::9, 2::6, 0] + vertex[1::9, 2::6, 1])
    alpha = R(H, W, 1) if with_alpha else None
    dmin, dmax, thr, amax = 0.5, 6.0, 0.1, 0.7
    valid = (depth > dmin) & (depth < dmax)
    valid = valid & ~((vertex.sum(2) == 0).unsqueeze(-1))
    err = torch.mean(torch.abs(src - image), -1, True)
    ref_mask = (err > thr) & valid
    if with_alpha:
        ref_mask = ref_mask & (alpha < amax)
    mask = torch.empty(H, W, 1, dtype=torch.bool, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.gps_new_gaussian_mask(W, H, depth.data_ptr(), src.data_ptr(), image.data_ptr(), vertex.data_ptr(),
                                     alpha.data_ptr() if with_alpha else None, dmin, dmax, thr, amax, mask.data_ptr(), st) == 0
    assert torch.equal(mask, ref_mask)
    assert 0.05 < ref_mask.float().mean() < 0.95

    ids = torch.full((P,), -1, dtype=torch.int32, device=DEV)
    count = torch.zeros(1, dtype=torch.int32, device=DEV)
    host_count = torch.zeros(16, dtype=torch.int32).pin_memory()
    ws = torch.empty(int(lib.gps_compact_mask_workspace_bytes(P)), dtype=torch.uint8, device=DEV)
    assert lib.gps_compact_mask(P, mask.data_ptr(), ids.data_ptr(), count.data_ptr(), host_count.data_ptr(), ws.data_ptr(),
                                ws.numel(), st) == 0
    torch.cuda.synchronize()
    ref_ids = torch.nonzero(ref_mask.reshape(-1)).squeeze(1).to(torch.int32)
    n = int(ref_ids.numel())
    assert int(count) == n and int(host_count[0]) == n
    assert torch.equal(ids[:n], ref_ids) and bool((ids[n:] == -1).all())
    # ragged length: not a multiple of the 4096-byte blocks, unaligned start
    for off, m in ((3, P - 5), (0, 1), (1, 0)):
        sub = mask.reshape(-1)[off:off + m]
        cnt2 = torch.zeros(1, dtype=torch.int32, device=DEV)
        ids2 = torch.full((max(m, 1),), -1, dtype=torch.int32, device=DEV)
        assert lib.gps_compact_mask(m, sub.data_ptr(), ids2.data_ptr(), cnt2.data_ptr(), None, ws.data_ptr(), ws.numel(), st) == 0
        r2 = torch.nonzero(sub).squeeze(1).to(torch.int32)
        assert int(cnt2) == r2.numel() and torch.equal(ids2[:r2.numel()], r2)

    k = max(1, n // 10)
    subset = torch.randperm(n, generator=torch.Generator().manual_seed(3))[:k].sort().values.to(torch.int32).to(DEV)
    normal = R(H, W, 3)
    outs = [torch.empty(k, 3, device=DEV) for _ in range(3)]
    assert lib.gps_gather_pixels(k, ids.data_ptr(), subset.data_ptr(), vertex.data_ptr(), image.data_ptr(), normal.data_ptr(),
                                 outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), st) == 0
    m3 = ref_mask.expand(H, W, 3)
    for o, src_map in zip(outs, (vertex, image, normal)):
        assert torch.equal(o, torch.masked_select(src_map, m3).reshape(-1, 3)[subset.long()])


def test_prune_mask_and_row_gather_match_the_tensor_sequence():
    """gps_prune_mask == slam_pipeline.cpp:566-575 evaluated with torch ops (bit for bit, thresholds ON the values for some rows);
    gps_gather_rows == index_select of the six parameter tensors (raw_gs_param.cpp:148-157) in one launch."""
    from gps_slam_amd._lib import lib
    g = torch.Generator(device=DEV).manual_seed(5)
    N = 100003
    ls = torch.randn(N, 3, device=DEV, generator=g) * 1.5 - 4.0
    ol = torch.randn(N, 1, device=DEV, generator=g) * 3.0
    small, large, low = 0.004, 0.08, 0.05
    ls[::11, 0] = float(np.log(np.float32(large)))  # max scale lands on / next to the threshold
    ls[::11, 1:] = -9.0
    smax = torch.exp(ls).max(-1).values
    ref = (smax < small) | (smax > large) | (torch.sigmoid(ol).squeeze(-1) < low)
    dele = torch.empty(N, dtype=torch.bool, device=DEV)
    keep = torch.empty(N, dtype=torch.bool, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.gps_prune_mask(N, ls.data_ptr(), ol.data_ptr(), small, large, low, dele.data_ptr(), keep.data_ptr(), st) == 0
    assert torch.equal(dele, ref) and torch.equal(keep, ~ref)
    assert 0.05 < ref.float().mean() < 0.95
    ids = torch.nonzero(keep).squeeze(1).to(torch.int32)
    m = int(ids.numel())
    rows = [3, 3, 4, 3, 45, 1]
    srcs = [torch.randn(N, r, device=DEV, generator=g) for r in rows]
    dsts = [torch.full((N, r), float("nan"), device=DEV) for r in rows]
    P6 = C.c_void_p * 6
    assert lib.gps_gather_rows(m, ids.data_ptr(), 6, P6(*[t.data_ptr() for t in srcs]), P6(*[t.data_ptr() for t in dsts]),
                               (C.c_int32 * 6)(*rows), st) == 0
    for s_, d_ in zip(srcs, dsts):
        assert torch.equal(d_[:m], s_[ids.long()]) and bool(torch.isnan(d_[m:]).all())


def test_rgba8_to_rgbf_equals_the_tensor_sequence():
    """gps_rgba8_to_rgbf == frame[..., :3].to(float32).div_(255) bit for bit (all 256 byte values)."""
    from gps_slam_amd._lib import lib
    H, W = 37, 50
    g = torch.Generator().manual_seed(2)
    rgba = torch.randint(0, 256, (H, W, 4), generator=g, dtype=torch.uint8)
    rgba.view(-1)[:256] = torch.arange(256, dtype=torch.uint8)
    rgba = rgba.to(DEV)
    out = torch.empty(H, W, 3, device=DEV)
    assert lib.gps_rgba8_to_rgbf(H * W, rgba.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    assert torch.equal(out, rgba[..., :3].to(torch.float32).div_(255.0))
    # the same launch also carrying Camera::toGPU's 28-float pack (gps_rgba8_to_rgbf_and_floats)
    import numpy as np
    vals = np.random.default_rng(3).standard_normal(28).astype(np.float32)
    out2, pack = torch.zeros(H, W, 3, device=DEV), torch.zeros(32, device=DEV)
    assert lib.gps_rgba8_to_rgbf_and_floats(H * W, rgba.data_ptr(), out2.data_ptr(), pack.data_ptr(), vals.ctypes.data, 28,
                                            torch.cuda.current_stream().cuda_stream) == 0
    assert torch.equal(out2, out) and np.array_equal(pack[:28].cpu().numpy(), vals) and float(pack[28:].abs().sum()) == 0.0

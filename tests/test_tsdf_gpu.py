"""GPU parity of the TSDF path, through the C-ABI: bit-exact against (i) the golden dumps of the reference's own
ITMLib CPU engine and (ii) the CPU oracle on larger seeded sequences (640x480 / 5 mm voxels).
Integer results (hash slots, block coordinates, ptr/offset, visible lists, voxel payloads, colours) AND float
images (depth, min/max, raycast points, ICP maps) must match bit for bit."""
import glob
import os

import numpy as np
import pytest
import torch

from tests import synth
from tests.test_oracle_tsdf import bits_equal, check_frame, check_free_view, crc_of_blocks, minmax_equal  # noqa: F401

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "tsdf_*.npz")))


class EngineView:
    """Adapts gps_slam_amd.tsdf_engine.TsdfEngine to the accessor interface the oracle checks use."""

    def __init__(self, eng):
        self.e = eng

    def _c(self):
        return self.e.counters_host()

    n_visible = property(lambda s: int(s._c()[2]))
    last_free_block = property(lambda s: int(s._c()[0]))
    last_free_excess = property(lambda s: int(s._c()[1]))
    fv_n_visible = property(lambda s: int(s._c()[3]))

    def visible_ids(self):
        return self.e.visible_ids.cpu().numpy()[:self.n_visible]

    def fv_visible_ids(self):
        return self.e.fv_visible_ids.cpu().numpy()[:self.fv_n_visible]

    def visible_type(self):
        return self.e.visible_type.cpu().numpy()

    def hash_rows(self):
        h = self.e.hash_host()
        keep = ~((h["ptr"] == -2) & (h["offset"] == 0) & (h["pos"] == 0).all(1))
        idx = np.nonzero(keep)[0]
        e = h[idx]
        return np.concatenate([idx[:, None], e["pos"].astype(np.int64), e["offset"][:, None], e["ptr"][:, None]],
                              1).astype(np.int32)

    def allocated_blocks(self):
        h = self.e.hash_host()
        ptr = h["ptr"][h["ptr"] >= 0]
        v = self.e.vba_host(ptr)
        return v.view(np.uint8).reshape(v.shape + (8,))[..., :7]

    def image(self, name):
        H, W = self.e.H, self.e.W
        t = {"minmax": (self.e.minmax, (H, W, 2)), "raycast": (self.e.raycast, (H, W, 4)),
             "icp_points": (self.e.icp_points, (H, W, 4)), "icp_normals": (self.e.icp_normals, (H, W, 4)),
             "depth": (self.e.depth, (H, W)), "fv_minmax": (self.e.fv_minmax, (H, W, 2)),
             "fv_raycast": (self.e.fv_raycast, (H, W, 4)), "fv_colour": (self.e.fv_colour, (H, W, 4))}[name]
        return t[0].cpu().numpy().reshape(t[1])


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_engine_reproduces_reference_golden(path):
    from gps_slam_amd.tsdf_engine import TsdfEngine
    g = np.load(path)
    get = lambda k, f: g["%s@%d" % (k, f)]
    W, H = int(g["W"]), int(g["H"])
    eng = TsdfEngine(W, H, float(g["fx"]), float(g["fy"]), float(g["cx"]), float(g["cy"]), float(g["voxel"]),
                     float(g["mu"]), float(g["vf_min"]), float(g["vf_max"]))
    v = EngineView(eng)
    n = g["rgb"].shape[0]
    for f in range(n):
        M, invM = eng.ProcessFrame(_dev(g["rgb"][f]), _dev(g["depth"][f].astype(np.int16)), g["c2w"][f])
        assert bits_equal(M, get("M", f)) and bits_equal(invM, get("invM", f))  # host pose algebra (SE3Pose)
        check_frame(v, get, f, window_only=True)
        for k, fr in enumerate(g["free_frames"]):
            if fr == f:
                tag = f * 1000 + k
                fM, fInv = eng.runRaycast(g["free_c2w"][k])
                assert bits_equal(fM, get("fv_M", tag)) and bits_equal(fInv, get("fv_invM", tag))
                check_free_view(v, get, tag, window_only=True)
    assert bits_equal(v.allocated_blocks()[::8], get("vba", n - 1))
    assert int(v._c()[5]) == 0  # no rendering-block overflow


@pytest.mark.parametrize("W,H,voxel,mu,frames", [(640, 480, 0.005, 0.02, 3), (320, 240, 0.01, 0.04, 6)])
def test_engine_matches_oracle_full_size(W, H, voxel, mu, frames):
    from gps_slam_amd.tsdf_engine import TsdfEngine
    from oracle import tsdf_ref as R
    seq = synth.make_sequence(W, H, frames, step_deg=1.0)
    o = R.TsdfOracle(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel, mu, 0.2, 10.0)
    eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel, mu, 0.2, 10.0)
    v = EngineView(eng)

    def same_frame():
        assert [v.n_visible, v.last_free_block, v.last_free_excess] == [o.n_visible, o.last_free_block, o.last_free_excess]
        assert bits_equal(v.visible_ids(), o.visible_ids())
        assert bits_equal(v.hash_rows(), o.hash_rows())
        assert bits_equal(v.visible_type(), o.visible_type())
        for name in ("depth", "raycast", "icp_points", "icp_normals"):
            assert bits_equal(v.image(name), o.image(name)), name
        assert minmax_equal(v.image("minmax"), o.image("minmax"), True)
        assert crc_of_blocks(v.allocated_blocks()) == crc_of_blocks(o.allocated_blocks())

    # 640x480 / 5 mm is also the sequence the REFERENCE's own CPU engine was run on at BASELINE size
    # (tests/golden/refdigest_tsdf_640x480_v5mm.npz): the HIP state is compared with the reference engine's digests directly,
    # not only with the restatement
    from tests.test_oracle_tsdf import FULL, check_against_fullsize_digest, check_free_view_against_fullsize_digest
    G = np.load(FULL) if (W, H, voxel) == (640, 480, 0.005) else None
    if G is not None:
        assert (int(G["n_frames"]), float(G["step_deg"]), float(G["mu"])) == (frames, 1.0, mu)
    for f in range(frames):
        M, invM = eng.ProcessFrame(_dev(seq["rgb"][f]), _dev(seq["depth"][f].astype(np.int16)), seq["c2w"][f])
        oM, oInv = R.pose_from_c2w(seq["c2w"][f])
        assert bits_equal(M, oM) and bits_equal(invM, oInv)
        o.process_frame(seq["rgb"][f], seq["depth"][f], oM, oInv)
        same_frame()
        if G is not None:
            check_against_fullsize_digest(v, G, f)
    assert v.n_visible > 1000 and (v.image("raycast")[..., 3] > 0).mean() > 0.8
    fM, fInv = eng.runRaycast(seq["c2w"][0])
    o.free_raycast(fM, fInv)
    if G is not None:
        check_free_view_against_fullsize_digest(v, G, (frames - 1) * 1000)
    assert v.fv_n_visible == o.fv_n_visible
    assert bits_equal(v.fv_visible_ids(), o.fv_visible_ids())
    for name in ("fv_raycast", "fv_colour"):
        assert bits_equal(v.image(name), o.image(name)), name
    assert minmax_equal(v.image("fv_minmax"), o.image("fv_minmax"), True)
    o.close()


def test_engine_matches_the_reference_engine_when_the_block_array_is_exhausted():
    """Maximum size: 0x40000 voxel blocks exhausted in frame 5, two more frames fused with nothing left to allocate -- the HIP
    engine against the REFERENCE engine's own digests (tests/golden/refdigest_tsdf_exhaustion_320x240_v2mm.npz), every frame."""
    from gps_slam_amd.tsdf_engine import TsdfEngine
    from tests.test_oracle_tsdf import check_against_fullsize_digest, exhaustion_scene
    G, seq, args, n = exhaustion_scene()
    eng = TsdfEngine(*args)
    v = EngineView(eng)
    for f in range(n):
        eng.ProcessFrame(_dev(seq["rgb"][f]), _dev(seq["depth"][f].astype(np.int16)), seq["c2w"][f])
        check_against_fullsize_digest(v, G, f)
    assert v.last_free_block == -1
    assert int(v._c()[5]) == 0  # the rendering-block cap was not hit (its order dependence is the one documented difference)


@pytest.mark.parametrize("W,H,n_blocks,n_buckets,n_excess", [(100, 75, 3000, 1 << 12, 64), (160, 120, 1 << 14, 1 << 10, 1 << 9),
                                                            (77, 50, 1 << 15, 1 << 16, 1 << 12)])
def test_small_tables_ragged_images_and_empty_frames_match_the_oracle(W, H, n_blocks, n_buckets, n_excess):
    """Exhaustion of the voxel-block array (n_blocks) AND of the excess list (few buckets: long chains, then no excess entry left),
    image sizes that are no multiple of the 8-pixel min/max cell or the 16-pixel workgroup patch, a frame without a single valid
    depth (all zero) in the middle and one with a hole: HIP == the CPU restatement bit for bit, frame by frame."""
    from gps_slam_amd.tsdf_engine import TsdfEngine
    from oracle import tsdf_ref as R
    frames = 6
    seq = synth.make_sequence(W, H, frames, step_deg=12.0)
    depth = seq["depth"].copy()
    depth[2] = 0                                   # an empty frame
    depth[3, H // 4:H // 2, W // 3:2 * W // 3] = 0  # a hole
    geo = (W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.01, 0.04, 0.2, 10.0)
    o = R.TsdfOracle(*geo, n_blocks=n_blocks, n_buckets=n_buckets, n_excess=n_excess)
    eng = TsdfEngine(*geo, n_blocks=n_blocks, n_buckets=n_buckets, n_excess=n_excess)
    v = EngineView(eng)
    o_prev = {"steps": 0, "rays": 0}
    for f in range(frames):
        M, invM = eng.ProcessFrame(_dev(seq["rgb"][f]), _dev(depth[f].astype(np.int16)), seq["c2w"][f])
        oM, oInv = R.pose_from_c2w(seq["c2w"][f])
        o.process_frame(seq["rgb"][f], depth[f], oM, oInv)
        assert [v.n_visible, v.last_free_block, v.last_free_excess] == [o.n_visible, o.last_free_block, o.last_free_excess], f
        assert bits_equal(v.visible_ids(), o.visible_ids()), f
        assert bits_equal(v.hash_rows(), o.hash_rows()), f
        assert bits_equal(v.visible_type(), o.visible_type()), f
        for name in ("depth", "raycast", "icp_points", "icp_normals"):
            assert bits_equal(v.image(name), o.image(name)), (name, f)
        assert minmax_equal(v.image("minmax"), o.image("minmax"), True), f
        assert crc_of_blocks(v.allocated_blocks()) == crc_of_blocks(o.allocated_blocks()), f
        # S-bar: the kernel's step log counts the steps of the reference's castRay loop -- an integer, equal to the
        # oracle's trip count although the kernel folds runs of unallocated steps into one trip (reads <= steps)
        hs, os_ = eng.ray_stats(), o.ray_stats()   # HIP: of this frame's launch; oracle: cumulative
        assert (hs["steps"], hs["rays"]) == (os_["steps"] - o_prev["steps"], os_["rays"] - o_prev["rays"]) and hs["reads"] <= hs["steps"], (f, hs, os_)
        o_prev = os_
    print("counters at the end:", [v.n_visible, v.last_free_block, v.last_free_excess], "ray stats of the last frame:", eng.ray_stats())
    fM, fInv = eng.runRaycast(seq["c2w"][1])
    o.free_raycast(fM, fInv)
    hs, os_ = eng.ray_stats(), o.ray_stats()
    assert (hs["steps"], hs["rays"]) == (os_["steps"] - o_prev["steps"], os_["rays"] - o_prev["rays"]), (hs, os_)
    assert bits_equal(v.fv_visible_ids(), o.fv_visible_ids())
    for name in ("fv_raycast", "fv_colour"):
        assert bits_equal(v.image(name), o.image(name)), name
    o.close()


@pytest.mark.parametrize("W,H,voxel,mu,frames,n_views", [(160, 120, 0.01, 0.04, 6, 5), (640, 480, 0.005, 0.02, 12, 9)])
def test_batched_free_views_equal_one_view_at_a_time(W, H, voxel, mu, frames, n_views):
    """gps_tsdf_free_raycast_batch (every launch of the free-view chain covers all views, per-view render state) against
    gps_tsdf_free_raycast called once per pose on the same volume: rays, colours, visible lists and min/max windows bit for
    bit; the scene's own free-view buffers untouched by the batch; a second batch (other poses, reused views) still exact."""
    from gps_slam_amd.tsdf_engine import TsdfEngine, pose_from_c2w
    seq = synth.make_sequence(W, H, frames, step_deg=2.0)
    eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel, mu, 0.2, 10.0)
    for f in range(frames):
        eng.ProcessFrame(_dev(seq["rgb"][f]), _dev(seq["depth"][f].astype(np.int16)), seq["c2w"][f])
    for order in (lambda k: k % frames, lambda k: (frames - 1 - 2 * k) % frames):
        poses = [pose_from_c2w(seq["c2w"][order(k)]) for k in range(n_views)]
        want = []
        for M, invM in poses:
            eng.runRaycast(pose=(M, invM))
            n = int(eng.counters.cpu()[3])
            want.append((eng.GetFreeVertex().clone(), eng.GetFreeImage().clone(), eng.fv_visible_ids[:n].clone(),
                         eng.fv_minmax.view(H, W, 2)[:H // 8 + 1, :W // 8 + 1].clone()))
        keep = (eng.fv_raycast.clone(), eng.fv_colour.clone(), eng.counters.clone())
        eng.runRaycastBatch(poses)
        torch.cuda.synchronize()
        for k in range(n_views):
            rays, col, ids, mm = want[k]
            v = eng._views[k]
            assert torch.equal(eng.GetFreeVertex(k), rays), k
            assert torch.equal(eng.GetFreeImage(k), col), k
            n = int(v["counters"].cpu()[3])
            assert n == ids.numel() and torch.equal(v["visible_ids"][:n], ids), k
            assert torch.equal(v["minmax"].view(H, W, 2)[:H // 8 + 1, :W // 8 + 1], mm), k
        assert (want[0][0][..., 3] > 0).float().mean() > 0.5
        assert torch.equal(eng.fv_raycast, keep[0]) and torch.equal(eng.fv_colour, keep[1]) and torch.equal(eng.counters, keep[2])


def test_integration_is_idempotent_in_weight_and_converges():
    """Size-independent property at full size: fusing the SAME frame twice keeps every allocated block, raises
    w_depth by exactly one where the voxel was updated, and leaves sdf unchanged up to the truncating store."""
    from gps_slam_amd.tsdf_engine import TsdfEngine
    W, H = 640, 480
    seq = synth.make_sequence(W, H, 1)
    eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, 0.2, 10.0)
    v = EngineView(eng)
    rgb, d = _dev(seq["rgb"][0]), _dev(seq["depth"][0].astype(np.int16))
    eng.ProcessFrame(rgb, d, seq["c2w"][0])
    h1, b1 = v.hash_rows(), v.allocated_blocks().copy()
    eng.ProcessFrame(rgb, d, seq["c2w"][0])
    h2, b2 = v.hash_rows(), v.allocated_blocks()
    # second pass may only add blocks that lost a same-frame bucket collision
    assert h2.shape[0] >= h1.shape[0] and (h2.shape[0] - h1.shape[0]) < 0.10 * h1.shape[0]
    keep = np.isin(h2[:, 0], h1[:, 0]) & (h2[:, 5] >= 0)
    b2k = b2[keep[h2[:, 5] >= 0]]
    w1, w2 = b1[..., 2].astype(int), b2k[..., 2].astype(int)
    assert set(np.unique(w2 - w1)) <= {0, 1}
    s1 = b1[..., :2].copy().view(np.int16)[..., 0].astype(int)
    s2 = b2k[..., :2].copy().view(np.int16)[..., 0].astype(int)
    upd = (w2 - w1) == 1
    assert np.abs(s2 - s1)[upd & (w1 > 0)].max() <= 1


def test_tracked_process_frame_matches_reference_poses():
    """ProcessFrame with the depth-only ExtendedTracker ON: the HIP path (per-pixel residuals + tree reduction on the GPU,
    6x6 LM on the host) against the poses the REFERENCE's CPU engine estimated on the same sequence
    (tests/golden/track_320x240.npz).  The reduction order differs (tree vs scan order), so poses agree to float
    re-association, not bit for bit: 2e-5 on every matrix entry; iteration counts and the inlier count follow."""
    import os
    from gps_slam_amd.tsdf_engine import TsdfEngine
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "track_320x240.npz"))
    W, H, n = int(G["W"]), int(G["H"]), int(G["n_frames"])
    seq = synth.make_sequence(W, H, n, step_deg=float(G["step_deg"]))
    eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel_size=float(G["voxel"]), mu=float(G["mu"]),
                     view_frustum_min=float(G["vf_min"]), view_frustum_max=float(G["vf_max"]), device="cuda:0")
    eng.turnOnTracking()
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)
    for f in range(n):
        M, invM = eng.ProcessFrameTracked(_dev(rgba[f]), _dev(seq["depth"][f].astype(np.int16)))
        assert np.abs(M - G["M"][f]).max() < 2e-5 and np.abs(invM - G["invM"][f]).max() < 2e-5, (f, np.abs(invM - G["invM"][f]).max())
        d = eng.track_diag()
        if f > 0:
            assert abs(d[10] - G["score"][f][0]) < 2e-3 * G["score"][f][0]  # (a few inliers more or less)
        # the map built along the tracked trajectory is the reference's up to a handful of blocks at the band's edge
        cnt = eng.counters_host()
        assert abs(int(cnt[2]) - int(G["n_visible"][f])) <= max(3, int(0.002 * G["n_visible"][f])), (f, cnt[2], G["n_visible"][f])


def test_tracked_frames_match_the_reference_engine_at_baseline_size():
    """640x480 / 5 mm, tracking ON: poses against the ones the REFERENCE's CPU engine estimated on the same frames
    (tests/golden/refdigest_tsdf_640x480_v5mm.npz, `trk_*`).  Tree vs scan-order float sums: 2e-5 per matrix entry; the
    allocation counters along the HIP-tracked trajectory may differ by the handful of band-edge blocks such a pose difference
    moves (bit-equality along the HIP poses is tests/test_pipeline_full_gpu.py's job)."""
    from gps_slam_amd.tsdf_engine import TsdfEngine
    from tests.test_oracle_tsdf import FULL
    G = np.load(FULL)
    W, H, n = int(G["W"]), int(G["H"]), int(G["n_frames"])
    seq = synth.make_sequence(W, H, n, step_deg=float(G["step_deg"]))
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)
    eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel_size=float(G["voxel"]), mu=float(G["mu"]), device="cuda:0")
    eng.turnOnTracking()
    v = EngineView(eng)
    for f in range(n):
        M, invM = eng.ProcessFrameTracked(_dev(rgba[f]), _dev(seq["depth"][f].astype(np.int16)))
        assert np.abs(invM - G["trk_invM"][f]).max() < 2e-5 and np.abs(M - G["trk_M"][f]).max() < 2e-5, (f, np.abs(invM - G["trk_invM"][f]).max())
        ref_vis, ref_free = int(G["trk_counts"][f][0]), int(G["trk_counts"][f][1])
        assert abs(v.n_visible - ref_vis) <= 0.002 * ref_vis + 2 and abs(v.last_free_block - ref_free) <= 0.002 * ref_vis + 2, f


def test_tracker_follows_ground_truth_at_full_size():
    """640x480, 5 mm voxels, 12 frames: tracked relative poses stay within 2 mm / 2e-3 of the ground-truth motion and two
    runs give bit-identical poses (fixed reduction tree)."""
    from gps_slam_amd.tsdf_engine import TsdfEngine
    W, H, n = 640, 480, 12
    seq = synth.make_sequence(W, H, n, step_deg=0.3)
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)
    c0inv = np.linalg.inv(seq["c2w"][0])
    runs = []
    for _ in range(2):
        eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel_size=0.005, mu=0.02, device="cuda:0")
        eng.turnOnTracking()
        poses = []
        for f in range(n):
            M, invM = eng.ProcessFrameTracked(_dev(rgba[f]), _dev(seq["depth"][f].astype(np.int16)))
            gt = (c0inv @ seq["c2w"][f]).T.reshape(-1)
            assert np.abs(invM - gt).max() < 2e-3, (f, np.abs(invM - gt).max())
            poses.append(invM.copy())
        runs.append(np.stack(poses))
    assert np.array_equal(runs[0], runs[1])


def test_tracker_without_mailbox_gives_the_same_poses():
    """gps_track_state.host_mailbox == NULL: every evaluation is a plain launch and the sums come back by memcpy -- the same
    kernels' body, the same fixed-order sums, so the poses are bit-identical to the pre-launched / mailbox path; a state that is
    handed a fresh scratch buffer (scratch_epoch = 0) keeps working.  "alone" / "two-along": the default hand-over evaluates,
    with every evaluation, the pose the LM loop would take next after a rejection (gps_track_state.mailbox_bytes) -- with none
    or two of them riding along the poses and the per-level evaluation counts are the same, bit for bit."""
    from gps_slam_amd.tsdf_engine import TsdfEngine
    W, H, n = 320, 240, 8
    seq = synth.make_sequence(W, H, n, step_deg=0.4)
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)
    runs, counts, along = [], [], {}
    for mode in ("mailbox", "memcpy", "new-scratch", "pinned-line", "alone", "two-along", "three-along", "device-summer", "device-summer-pinned-line"):
        eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel_size=0.01, mu=0.04, device="cuda:0")
        # "mailbox": the default hand-over -- pre-launched evaluations, argument line written through the BAR into device memory
        # (gps_track_state.dev_arg_line) when the device has a large BAR; "pinned-line": the line in the pinned mailbox, relayed
        eng.turnOnTracking(bar_arg_line="pinned-line" not in mode, poses_riding_along={"alone": 0, "two-along": 2, "three-along": 3}.get(mode, 1),
                           host_summed_rows="device-summer" not in mode)   # (host-summed rows are the default hand-over)
        assert (eng.track_state.dev_arg_line is None) == ("pinned-line" in mode or eng._arg_line is None)
        if mode == "memcpy":
            eng.track_state.host_mailbox = None
        poses, levels, rode, used = [], [], 0, 0
        for f in range(n):
            if mode == "new-scratch" and f == 4:
                eng.track_scratch = torch.full_like(eng.track_scratch, 0x5A)  # garbage, not zeros
                eng.track_state.scratch_epoch = 0
            M, invM = eng.ProcessFrameTracked(_dev(rgba[f]), _dev(seq["depth"][f].astype(np.int16)))
            poses.append(invM.copy())
            d = np.array(eng.track_state.diag[:], np.float32)
            levels.append(d[:12].copy())
            rode += int(d[12]); used += int(d[13])
        runs.append(np.stack(poses))
        counts.append(np.stack(levels))
        along[mode] = (rode, used)
    assert all(np.array_equal(runs[0], r) for r in runs[1:])
    assert all(np.array_equal(counts[0], c) for c in counts[1:])   # evaluations per level, valid points, f, score, det(H)
    assert np.abs(runs[0][-1] - runs[0][0]).max() > 1e-3  # the camera actually moved
    assert along["memcpy"] == (0, 0) and along["pinned-line"] == (0, 0) and along["alone"] == (0, 0) and along["device-summer-pinned-line"] == (0, 0)
    if eng._arg_line is not None:   # (a device whose memory the host can write: the BAR line)
        rode, used = along["mailbox"]
        assert rode > 0 and 0 < used <= rode, along
        assert along["two-along"][0] > rode and along["two-along"][1] >= used, along
        assert along["three-along"][0] > along["two-along"][0] and along["three-along"][1] >= along["two-along"][1], along


# ----------------------------------------------------------------------------- meshing + persistence
def _fused_pair(W, H, voxel, mu, frames, **kw):
    from gps_slam_amd.tsdf_engine import TsdfEngine
    from oracle import tsdf_ref as R
    seq = synth.make_sequence(W, H, frames, step_deg=1.0)
    o = R.TsdfOracle(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel, mu, 0.2, 10.0)
    eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel, mu, 0.2, 10.0, **kw)
    for f in range(frames):
        M, invM = eng.ProcessFrame(_dev(seq["rgb"][f]), _dev(seq["depth"][f].astype(np.int16)), seq["c2w"][f])
        o.process_frame(seq["rgb"][f], seq["depth"][f], M, invM)
    return seq, eng, o


@pytest.mark.parametrize("W,H,voxel,mu,frames", [(96, 72, 0.01, 0.04, 4), (256, 192, 0.008, 0.032, 3)])
def test_mesh_scene_is_bit_equal_to_the_cpu_engine_order(W, H, voxel, mu, frames):
    """gps_tsdf_mesh_scene vs the restatement of the reference CPU meshing engine (itself bit-equal to the reference on the
    committed golden): same triangles, same ORDER, same bits (positions, vertex colours, clr)."""
    seq, eng, o = _fused_pair(W, H, voxel, mu, frames)
    want = o.mesh()
    tri, counts = eng.MeshScene(max_triangles=want.shape[0] + 1000)
    n, gen = (int(v) for v in counts.cpu())
    assert n == gen == want.shape[0] and n > 10000
    got = tri[:n].cpu().numpy()
    assert bits_equal(got, want)
    # deterministic order: a second run writes the same bytes
    tri2, _ = eng.MeshScene(max_triangles=want.shape[0] + 1000)
    assert torch.equal(tri[:n], tri2[:n])
    # the reference's clamp: noTotalTriangles stops at max - 1 and the kept prefix is unchanged
    cap = 5000
    tri3, counts3 = eng.MeshScene(max_triangles=cap)
    assert counts3.cpu().tolist() == [cap - 1, want.shape[0]]
    assert bits_equal(tri3[:cap - 1].cpu().numpy(), want[:cap - 1])
    o.close()


def test_mesh_properties_at_full_size():
    """640x480, 5 mm voxels (BASELINE's size; the oracle would take minutes): size-independent properties -- every vertex
    lies on a voxel-grid edge of its cube, inside the scene's bounding box, triangle count reproducible."""
    from gps_slam_amd.tsdf_engine import TsdfEngine
    W, H = 640, 480
    seq = synth.make_sequence(W, H, 4, step_deg=1.0)
    eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, 0.2, 10.0)
    for f in range(4):
        eng.ProcessFrame(_dev(seq["rgb"][f]), _dev(seq["depth"][f].astype(np.int16)), seq["c2w"][f])
    tri, counts = eng.MeshScene(max_triangles=1 << 23)
    n = int(counts[0])
    assert 500000 < n < (1 << 23) - 1
    p = tri[:n, :3].reshape(-1, 3) / 0.005  # voxel units
    frac = (p - torch.round(p)).abs()
    on_grid = (frac < 1e-3).sum(1)
    assert int((on_grid < 2).sum()) == 0  # at least two coordinates integral: the vertex sits on a cube edge
    c = tri[:n, 3:7]
    assert float(c.min()) >= 0.0 and float(c.max()) <= 1.0
    # the surface lies where the depth maps put it: compare with the live raycast points
    pts = eng.GetLiveVertex()
    hit = pts[..., 3] > 0
    lo = pts[..., :3][hit].min(0).values * 0.005 - 0.1
    hi = pts[..., :3][hit].max(0).values * 0.005 + 0.1
    verts = tri[:n, :3].reshape(-1, 3)
    inside = ((verts >= lo) & (verts <= hi)).all(1).float().mean()
    assert float(inside) > 0.5


def test_save_mesh_ply_and_scene_files_match_the_reference_byte_for_byte(tmp_path):
    """SaveSceneToMesh (WritePLY) and SaveToFile / LoadFromFile against files written by the reference's own CPU engine
    (oracle/_ref/itm_ref, mesh mode) on the same sequence: vertex positions and faces of the PLY are the same text (the CPU
    engine leaves colours at 0, the CUDA engine and this one fill them); voxel.dat / alloc.dat / vba.txt / hash.dat /
    excess.dat / last.txt are the same bytes (voxel pad byte excluded)."""
    from gps_slam_amd.tsdf_engine import TsdfEngine
    from oracle import tsdf_ref as R
    if not R.available():
        pytest.skip("oracle/_ref/itm_ref not built")
    W, H = 96, 72
    seq = synth.make_sequence(W, H, 3, step_deg=1.0)
    ref_dir = tmp_path / "ref"
    ref_dir.mkdir()
    ref = R.run(seq, 0.01, 0.04, 0.2, 10.0, mesh=True, save_dir=str(ref_dir))
    # reference capacities so that the raw dumps have the same size
    eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.01, 0.04, 0.2, 10.0)
    for f in range(3):
        eng.ProcessFrame(_dev(seq["rgb"][f]), _dev(seq["depth"][f].astype(np.int16)), seq["c2w"][f])
    n = eng.SaveSceneToMesh(str(tmp_path / "mesh.ply"))
    assert n == ref[("mesh", 2)].shape[0]
    mine = open(tmp_path / "mesh.ply").read().split("\n")
    theirs = open(ref_dir / "mesh.ply").read().split("\n")
    assert len(mine) == len(theirs) and mine[:12] == theirs[:12]
    assert [" ".join(l.split()[:3]) for l in mine[12:12 + 3 * n]] == [" ".join(l.split()[:3]) for l in theirs[12:12 + 3 * n]]
    assert mine[12 + 3 * n:] == theirs[12 + 3 * n:]
    assert any(l.split()[3:] != ["0", "0", "0"] for l in mine[12:12 + 3 * n])
    eng.SaveToFile(str(tmp_path / "state"))
    for name in ("alloc.dat", "vba.txt", "hash.dat", "excess.dat", "last.txt"):
        assert open(tmp_path / "state" / "Scene" / name, "rb").read() == open(ref_dir / name, "rb").read(), name
    a = np.fromfile(tmp_path / "state" / "Scene" / "voxel.dat", np.uint8)
    b = np.fromfile(ref_dir / "voxel.dat", np.uint8)
    assert a.shape == b.shape and bits_equal(a[:8], b[:8])
    av, bv = a[8:].reshape(-1, 8), b[8:].reshape(-1, 8)
    assert bits_equal(av[:, :7], bv[:, :7])
    # round trip into a fresh engine: same mesh, and fusion continues identically
    e2 = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.01, 0.04, 0.2, 10.0)
    e2.LoadFromFile(str(tmp_path / "state"))
    t1, c1 = eng.MeshScene(1 << 21)
    t2, c2 = e2.MeshScene(1 << 21)
    assert torch.equal(c1, c2) and torch.equal(t1[:n], t2[:n])
    assert torch.equal(eng.counters[:2], e2.counters[:2])


def test_mesh_scene_reproduces_reference_cpu_engine_golden():
    """HIP meshing vs the triangles the reference's ITMMeshingEngine_CPU produced (tests/golden/mesh_64x48.npz): bit-equal."""
    from gps_slam_amd.tsdf_engine import TsdfEngine
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mesh_64x48.npz"))
    W, H, n = int(g["W"]), int(g["H"]), int(g["n_frames"])
    seq = synth.make_sequence(W, H, n, step_deg=float(g["step_deg"]))
    eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], float(g["voxel"]), float(g["mu"]), float(g["vf_min"]),
                     float(g["vf_max"]))
    for f in range(n):
        eng.ProcessFrame(_dev(seq["rgb"][f]), _dev(seq["depth"][f].astype(np.int16)), seq["c2w"][f])
    tri, counts = eng.MeshScene(max_triangles=1 << 18)
    assert int(counts[0]) == g["triangles"].shape[0]
    assert bits_equal(tri[:int(counts[0]), :3].cpu().numpy(), g["triangles"])

#!/usr/bin/env python
"""bench.py -- SLAM frames/s of the GPS-SLAM hot path on MI355X.

One "step" = one SLAM frame of SLAMPipeline::SLAMTrainCams (slam/slam_pipeline.cpp:69-132) on synthetic 640x480 RGB-D with
~200k Gaussians: per frame the upload of the frame (rgb uchar4 + depth int16, pinned host -> HBM, ITMViewBuilder::UpdateView),
depth-ICP tracking, TSDF fuse + live raycast; every 10th frame the <= 9 free-view raycasts, new-Gaussian sampling, 20 optimise
iterations (forward, L1, backward, Adam) and the prune.  The loop is driven through the reference's construction path on the
C++ host layer: createTsdfEngine(DatasetReader, config) -> CLIEngine -> SLAMPipeline::setTsdfEngine -> one processFrame per
step (gps_slam_amd/host/infinitam_tools.hpp, slam_pipeline.hpp).

    python bench.py --gpus N --steps K --warmup W
N > 1: launched by torch.distributed.run, one rank per GPU, each rank runs an INDEPENDENT scene (different seed) -> weak
scaling, no data-path collective; barrier + max-over-ranks timing only.

Timed window.  The map update of a keyframe (every 10th frame) is ~10 frames' worth of GPU work; a window that cuts a
keyframe period in the middle measures a different mix for every (K, W), and the first update of a freshly seeded scene is
not the steady state (it prunes / adds ten times what later ones do; the keyframe list is still filling).  An untimed
prologue is therefore run before the W warm-up frames so that timed step 0 is frame max(30, ceil(W / 10) * 10), a keyframe:
a window of K = 10 m steps then holds exactly m whole periods (m keyframe updates, each with its 9 following frames) of
the settled loop, and --steps 20 and --steps 100 measure the same thing.
Both schedules are timed on identical scenes: `sequential` (the reference's: the update completes before the next frame is
looked at) and `overlap` (tracking/fusion of the following frames on a second stream while the update runs); `value` is the
overlap schedule, `config.schedules` carries both.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
PERIOD = 10            # local_opt_interval of every shipped config


def synthetic_sequence(W, H, n_frames, seed):
    """Procedural room sequence (tests/synth.py) with the first camera as the world frame (what a tracked run uses)."""
    from tests import synth
    seq = synth.make_sequence(W, H, n_frames, step_deg=0.25 + 0.01 * (seed % 7))
    c0inv = np.linalg.inv(seq["c2w"][0].astype(np.float64))
    seq["c2w"] = np.stack([(c0inv @ c.astype(np.float64)) for c in seq["c2w"]]).astype(np.float32)
    return seq


def synthetic_sequence_device(W, H, n_frames, seed, device, intrinsics=None, step_deg=None, texture="room", world_scale=1.0):
    """tests/synth.py's room rendered with torch on the GPU (float64, the same formulas): the numpy renderer needs 0.2-0.6 s per
    frame, too slow for the extra configurations measured after the main windows (config.other_configs).  Same dictionary as
    synthetic_sequence (arrays on the host).  intrinsics = (fx, fy, cx, cy) or None for the 90-degree pinhole.
    step_deg: the orbit's angle per frame (default 0.25-0.31 by seed).  world_scale: the whole scene -- room, spheres, orbit --
    scaled about the origin (0.4: the camera is 0.4-1.5 m from the surfaces, a pixel's footprint 1.3-4.7 mm, below the 5 mm voxel).
    texture "fine": no checker; the low-frequency colour at 0.55 contrast + three sine gratings of 12 / 15 / 19 mm period (amplitude 0.15
    each) in different directions: detail of 2-4 voxels per period that the image resolves (4-12 pixels per period at 0.5-1.5 m)
    and a 5 mm colour volume attenuates -- a workload where the Gaussians have something to add (round-5 review, item 8a)."""
    from tests import synth
    fx, fy, cx, cy = intrinsics if intrinsics else (0.5 * W, 0.5 * W, (W - 1) / 2.0, (H - 1) / 2.0)
    poses = synth.orbit_poses(n_frames, step_deg=step_deg if step_deg is not None else 0.25 + 0.01 * (seed % 7))
    ws = float(world_scale)
    if ws != 1.0:
        poses = [np.concatenate([np.concatenate([p[:3, :3], p[:3, 3:4] * ws], 1), p[3:4]], 0) for p in poses]
    dd = dict(dtype=torch.float64, device=device)
    ys, xs = torch.meshgrid(torch.arange(H, **dd), torch.arange(W, **dd), indexing="ij")
    d_cam = torch.stack([(xs - cx) / fx, (ys - cy) / fy, torch.ones_like(xs)], -1)
    half = torch.tensor((3.0 * ws, 1.5 * ws, 2.5 * ws), **dd)
    spheres = tuple(tuple(v * ws for v in sp) for sp in ((0.4, 0.2, 0.3, 0.45), (-0.8, 0.5, -0.4, 0.35)))
    rgbs, depths = [], []
    for c2w in poses:
        R, o = torch.as_tensor(c2w[:3, :3], **dd), torch.as_tensor(c2w[:3, 3], **dd)
        d = d_cam @ R.T
        t_best = torch.full((H, W), float("inf"), **dd)
        for ax in range(3):
            for sgn in (-1.0, 1.0):
                t = (sgn * half[ax] - o[ax]) / d[..., ax]
                p = o + t[..., None] * d
                ok = t > 1e-6
                for a2 in range(3):
                    if a2 != ax:
                        ok &= p[..., a2].abs() <= half[a2] + 1e-9
                t_best = torch.where(ok & (t < t_best), t, t_best)
        for sx, sy, sz, sr in spheres:
            oc = o - torch.tensor((sx, sy, sz), **dd)
            a, b, cc = (d * d).sum(-1), 2 * (d * oc).sum(-1), (oc * oc).sum() - sr * sr
            disc = b * b - 4 * a * cc
            t = (-b - torch.sqrt(disc)) / (2 * a)
            ok = (disc > 0) & (t > 1e-6)
            t_best = torch.where(ok & (t < t_best), t, t_best)
        hit = torch.isfinite(t_best)
        tb = torch.where(hit, t_best, torch.zeros_like(t_best))
        p = (o + tb[..., None] * d) / ws   # (texture coordinates of the unscaled room)
        r = 0.5 + 0.5 * torch.sin(3.1 * p[..., 0] + 1.7 * p[..., 1])
        g = 0.5 + 0.5 * torch.sin(2.3 * p[..., 1] - 2.9 * p[..., 2] + 1.0)
        b_ = 0.5 + 0.5 * torch.sin(4.1 * p[..., 2] + 0.7 * p[..., 0] - 0.5)
        if texture == "fine":
            q = p * ws   # (metres of the scene as rendered: the gratings' periods are physical)
            tau = 2.0 * np.pi
            g1 = torch.sin(tau * (q[..., 0] + 0.5 * q[..., 1] + 0.3 * q[..., 2]) / 0.012)
            g2 = torch.sin(tau * (0.4 * q[..., 0] - q[..., 1] + 0.6 * q[..., 2]) / 0.015)
            g3 = torch.sin(tau * (0.3 * q[..., 0] + 0.5 * q[..., 1] - q[..., 2]) / 0.019)
            base = 0.225 + 0.55 * torch.stack([r, g, b_], -1)
            tex = (base + 0.15 * torch.stack([g1, g2, g3], -1)).clamp(0, 1)
        else:
            checker = ((torch.floor(p[..., 0] * 2) + torch.floor(p[..., 1] * 2) + torch.floor(p[..., 2] * 2)) % 2) * 0.25
            tex = (torch.stack([r, g, b_], -1) * 0.75 + checker[..., None]).clamp(0, 1)
        rgbs.append((torch.where(hit[..., None], tex, torch.zeros_like(tex)) * 255.0 + 0.5).to(torch.uint8).cpu())
        depths.append(torch.where(hit, torch.round(tb * 1000.0).clamp(0, 65535), torch.zeros_like(tb)).to(torch.int32).cpu())
    c2w = np.stack(poses).astype(np.float64)
    c0inv = np.linalg.inv(c2w[0])
    return dict(W=W, H=H, fx=float(fx), fy=float(fy), cx=float(cx), cy=float(cy), rgb=torch.stack(rgbs).numpy(),
                depth=torch.stack(depths).numpy().astype(np.uint16), c2w=np.stack([c0inv @ c for c in c2w]).astype(np.float32))


def seed_gaussians(seq, n_gauss, seed, device):
    """~n_gauss Gaussians on the scene surfaces: a few views' depth back-projected, RawGaussianParams::init on the samples
    (KNN scale, normal -> quaternion, colour -> SH DC), small random higher-order SH so all 16 bands carry signal."""
    from gps_slam_amd.gs_model import SLAMGaussianModel
    from gps_slam_amd.slam_pipeline import compute_normal_map
    W, H, n_frames = seq["W"], seq["H"], seq["rgb"].shape[0]
    fx, fy, cx, cy = seq["fx"], seq["fy"], seq["cx"], seq["cy"]
    g = torch.Generator(device=device).manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device), indexing="ij")
    pts, cols, nrm = [], [], []
    for k in range(0, n_frames, max(1, n_frames // 6)):
        d = torch.as_tensor(seq["depth"][k].astype(np.float32) / 1000.0).to(device)
        img = torch.as_tensor(seq["rgb"][k]).to(device).float() / 255.0
        c2w = torch.as_tensor(seq["c2w"][k]).to(device)
        pc = torch.stack([(xs - cx) / fx * d, (ys - cy) / fy * d, d], -1)
        pw = pc @ c2w[:3, :3].T + c2w[:3, 3]
        n = compute_normal_map(pw)
        ok = (d > 0.3).reshape(-1)
        pts.append(pw.reshape(-1, 3)[ok]); cols.append(img.reshape(-1, 3)[ok]); nrm.append(n.reshape(-1, 3)[ok])
    pts, cols, nrm = torch.cat(pts), torch.cat(cols), torch.cat(nrm)
    sel = torch.randperm(pts.shape[0], device=device, generator=g)[:n_gauss].sort().values  # (view, pixel) order, as addGaussians appends
    helper = SLAMGaussianModel(dict(capacity=1 << 12), device=device)
    new = helper.init_params(pts[sel].contiguous(), cols[sel].contiguous(), nrm[sel].contiguous())
    new["featuresRest"] = (torch.randn(new["featuresRest"].shape, device=device, generator=g) * 0.02).contiguous()
    return [new[k].contiguous() for k in ("means", "scales", "quats", "featuresDc", "featuresRest", "opacities")]


ENV_SWITCHES = ("GPS_BENCH_PINNED_LINE", "GPS_BENCH_RIDING_ALONG", "GPS_BENCH_DEVICE_SUMMER", "GPS_BENCH_OPT_ITERS", "GPS_BENCH_PREFETCH", "GPS_BENCH_ASYNC_RAYCASTS",
                "GPS_BENCH_STREAMS", "GPS_BENCH_MERGE", "GPS_BENCH_RESERVE", "GPS_BENCH_PIPELINE_RAYCASTS", "GPS_BENCH_FRAME_TIMES", "GPS_BENCH_PIPE_TIMES", "GPS_BENCH_SHARE_GPU")


def env_overrides():
    """the A/B switches of this file that are set in the environment, as one string ("" in a normal run): every one of them
    changes what the line's numbers mean, so the line says which were on"""
    return ",".join("%s=%s" % (k, os.environ[k]) for k in ENV_SWITCHES if os.environ.get(k))


class Scene:
    """One scene on the C++ host layer, built the way slam_trainer.cpp builds it.  `seeds` = None: the model starts EMPTY, as the
    reference's does (the first keyframe update fills it from the whole first view)."""

    def __init__(self, seq, seeds, seed, use_gt_pose, overlap, n_frames, keyframe_theta, keyframe_trans, capacity=1 << 19, tsdf=None):
        import gps_slam_amd._host as H_
        self.H_ = H_
        W, H = seq["W"], seq["H"]
        reader = H_.DatasetReader(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"])
        self.cams = []
        for k in range(n_frames):
            c = H_.Camera(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][k]))
            c.id = k
            # what the dataset's files hold (uint8 colour, uint16 millimetres): createTsdfEngine turns them into the uchar4 / short
            # images the engine consumes -- the same bytes as from the reader's float image / depth (tests/test_tsdf_facade_gpu.py),
            # without 16 bytes per pixel and frame of host memory for the floats
            c.image = torch.as_tensor(seq["rgb"][k])
            c.depth = torch.as_tensor(seq["depth"][k].view(np.int16))
            reader.addTrainCamera(c)
            # the pipeline's camera of this frame carries no float image: it is derived on the device from the frame the
            # engine uploads (3 of its 4 bytes per pixel) instead of a second, 12-byte-per-pixel upload
            pc = H_.Camera(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][k]))
            pc.id = k
            self.cams.append(pc)
        self.cli = H_.createTsdfEngine(reader, dict(dict(voxel_size=0.005, trunc_dist=0.02, viewFrustum_min=0.2, viewFrustum_max=10.0,
                                                         use_gt_pose=1 if use_gt_pose else 0), **(tsdf or {})))
        self.engine = self.cli.getMainEngine()
        if os.environ.get("GPS_BENCH_PINNED_LINE"):  # A/B aid: the tracker's argument line in the pinned mailbox (relay path)
            self.engine.setBarArgLine(False)
        if os.environ.get("GPS_BENCH_DEVICE_SUMMER"):  # A/B aid: the evaluation's rows added by a workgroup on the device (round 4) instead of by the tracking thread
            self.engine.setHostSummedRows(False)
        if os.environ.get("GPS_BENCH_RIDING_ALONG"):  # A/B aid: poses of the LM loop's reject branch evaluated with every evaluation (0..2, default 1)
            self.engine.setPosesRidingAlong(int(os.environ["GPS_BENCH_RIDING_ALONG"]))
        self.model = H_.SLAMGaussianModel()
        self.model.loadConfig(dict(capacity=capacity, isect_capacity=8 << 20))
        if seeds is not None:
            self.model.getGaussianParms().add([t.clone() for t in seeds])
        self.pipe = H_.SLAMPipeline(seed)
        self.pipe.setTsdfEngine(self.cli)
        self.pipe.setModel(self.model)
        self.pipe.loadConfig(dict(keyframe_theta_thres=keyframe_theta, keyframe_trans_thres=keyframe_trans))
        if os.environ.get("GPS_BENCH_OPT_ITERS"):  # probe aid (tools/probe/critical_path.sh): NOT the metric's workload -- the line says so
            self.pipe.loadConfig(dict(local_opt_iters=int(os.environ["GPS_BENCH_OPT_ITERS"])))
        self.pipe.overlap_mapping = bool(overlap)
        self.pipe.mapping_thread = bool(overlap)
        if os.environ.get("GPS_BENCH_PREFETCH"):  # A/B aid: the next iteration's preprocessing in the backward kernel's tail (1, default) or its own launch (0)
            self.pipe.prefetch_next_preprocess = os.environ["GPS_BENCH_PREFETCH"] != "0"
        if os.environ.get("GPS_BENCH_ASYNC_RAYCASTS"):  # A/B aid (tools/probe/outliers.sh): the keyframe views' raycasts beside the first iterations (1, default) or before them (0)
            self.pipe.async_raycasts = os.environ["GPS_BENCH_ASYNC_RAYCASTS"] != "0"
        if os.environ.get("GPS_BENCH_STREAMS"):  # A/B aid: stream kinds "frame,map,raycast" (SLAMPipeline::frame_stream_kind ...; default 3,4,2 = own streams at the highest / default / lowest priority; 0 / 1 = torch's high- / normal-priority pool)
            f_, m_, r_ = (int(x) for x in os.environ["GPS_BENCH_STREAMS"].split(","))
            self.pipe.frame_stream_kind, self.pipe.map_stream_kind, self.pipe.raycast_stream_kind = f_, m_, r_
        if os.environ.get("GPS_BENCH_PIPELINE_RAYCASTS"):  # A/B aid: an update's free views enqueued by the frame thread at the keyframe (1, default) or by the map worker at the start of its job (0)
            self.pipe.pipeline_raycasts = os.environ["GPS_BENCH_PIPELINE_RAYCASTS"] != "0"
        if os.environ.get("GPS_BENCH_RESERVE"):  # A/B aid: what gps_set_frame_chain_reserve gets in the overlap schedule (default 1; 0 = off; 3 = + forward rasterizer at 3 workgroups per unit)
            self.pipe.frame_chain_reserve = int(os.environ["GPS_BENCH_RESERVE"])
        if os.environ.get("GPS_BENCH_MERGE"):  # A/B aid: window and keyframe views raycast as one batch (default 0)
            self.pipe.merge_keyframe_raycasts = os.environ["GPS_BENCH_MERGE"] != "0"
        if os.environ.get("GPS_BENCH_PIPE_TIMES"):  # debug aid: processFrame calls longer than this many ms print where the host time went
            self.pipe.frame_report_ms = float(os.environ["GPS_BENCH_PIPE_TIMES"])
        self.model.reserveWorkspace(W, H)

    def run(self, lo, hi):
        trace = os.environ.get("GPS_BENCH_FRAME_TIMES")  # debug aid: host-side duration of every processFrame call
        for i in range(lo, hi):
            t = time.perf_counter()
            self.pipe.processFrameCLI(i, self.cams[i])
            if trace:
                sys.stderr.write("frame %d: %.3f ms\n" % (i, 1e3 * (time.perf_counter() - t)))
        t = time.perf_counter()
        self.pipe.flush()
        if trace:
            sys.stderr.write("flush: %.3f ms\n" % (1e3 * (time.perf_counter() - t)))

    def close(self):
        self.cli.Shutdown()


def prime(device):
    """Run the whole per-frame loop once on a tiny throwaway scene (loads every kernel, warms the allocator).  Also bounds
    libtorch's intra-op pool by the container's CPU quota for callers that did not go through main()'s pin_to_gpu_numa (the
    tools, the soak test; GPS_BENCH_NO_THREAD_CAP=1 leaves it alone: tools/probe/early_stall.py reproduces the frozen-process
    stalls with it)."""
    if not os.environ.get("GPS_BENCH_NO_THREAD_CAP"):
        from gps_slam_amd.dist_util import cap_host_threads
        cap_host_threads()
    seq = synthetic_sequence(64, 48, 21, 1)
    seeds = seed_gaussians(seq, 500, 1, device)
    for use_gt in (True, False):
        s = Scene(seq, seeds, 1, use_gt, overlap=not use_gt, n_frames=21, keyframe_theta=1.0, keyframe_trans=0.02)
        s.run(0, 21)
        torch.cuda.synchronize()
        s.close()


def timed_window(warmup):
    """-> (index of timed step 0, untimed prologue frames run before the warm-up frames).
    Untimed prologue: (a) timed step 0 is a keyframe (see the module docstring), (b) the loop has reached its own steady state
    before anything is timed -- the first map update of a freshly seeded scene prunes ~8 % of the seeds and adds 20k Gaussians
    at once, and the keyframe list needs ~28 frames to fill to the 7 every later update draws from."""
    settle = 3 * PERIOD
    first = max(settle, -(-warmup // PERIOD) * PERIOD)  # a multiple of the keyframe period, >= settle
    return first, first - warmup


def setup_ranks(backend="nccl", need_gpu=True):
    """One process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*): pick the device, pin the
    process to its GPU's NUMA node, join the process group (RCCL; used for barrier + max-over-ranks only -- ranks run
    independent scenes).  -> rank, local_rank, world, Group, placement description, device string."""
    from gps_slam_amd.dist_util import Group, env_ranks, pin_to_gpu_numa
    rank, local_rank, world = env_ranks()
    device = None
    grp_device = None
    if need_gpu:
        assert torch.cuda.is_available(), "bench.py needs the MI355X"
        dev_index = local_rank
        ndev = torch.cuda.device_count()
        if ndev < world and os.environ.get("GPS_BENCH_SHARE_GPU"):
            # rehearsal of the N > 1 launch on a box with fewer GPUs than ranks (tools/probe/share_gpu.sh): the ranks share the
            # devices there are and meet over gloo (RCCL refuses two ranks on one device).  Exercises everything per-rank --
            # seeds, threads, pinned buffers, BAR lines, barriers, the aggregate line -- NOT a scaling measurement.
            dev_index, backend = local_rank % ndev, "gloo"
        torch.cuda.set_device(dev_index)
        device = "cuda:%d" % dev_index
        grp_device = device if backend == "nccl" else None
    ndev_ = torch.cuda.device_count() if need_gpu else 0
    placement = pin_to_gpu_numa(local_rank, world, device_of_rank=(lambda r: r % ndev_) if 0 < ndev_ < world else None)
    grp = Group(backend=backend, device=grp_device)
    return rank, local_rank, world, grp, placement, device


def _detrended_spread(ms):
    """(max - min) / median of the windows' ms_per_step after removing their least-squares LINE: consecutive windows are not
    repeats of one workload -- the orbit keeps revealing new surface, every keyframe adds Gaussians (windows_gaussians) and
    later windows are slower for that reason; what is left after the trend is the run-to-run noise."""
    n = len(ms)
    if n < 3:
        return 0.0
    xs = np.arange(n, dtype=np.float64)
    a, b = np.polyfit(xs, np.asarray(ms, np.float64), 1)
    res = np.asarray(ms) - (a * xs + b)
    return float((res.max() - res.min()) / np.median(ms))


def _no_gpu_sync():
    pass


def _device_mallocs(need_gpu):
    """hipMalloc calls of the caching allocator so far (a segment allocated inside a timed window stalls every stream)"""
    return int(torch.cuda.memory_stats().get("num_device_alloc", 0)) if need_gpu else 0


def main(argv=None, scene_factory=None, backend="nccl", need_gpu=True, extras=True):
    """argv / scene_factory / backend / need_gpu / extras exist for tests/test_multi_rank_cpu.py, which runs this very function
    with world_size 2 on gloo and a stub scene (no GPU in the build container): the N > 1 control flow -- ranks, seeds, barriers,
    max-over-ranks per window, whole-job aggregation, rank-0-only JSON line, final barrier -- is then the code the driver runs."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=5,
                    help="consecutive K-step windows timed per schedule (each bracketed by barrier + synchronize, max over ranks); "
                         "`value` is the MEDIAN window, config.windows_ms_per_step lists all of them")
    ap.add_argument("--schedule", choices=("both", "overlap", "sequential"), default="both",
                    help="which keyframe schedule(s) to time; `value` is the overlap schedule unless only sequential is run")
    ap.add_argument("--gaussians", type=int, default=200000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--keyframe-theta", type=float, default=1.0,
                    help="keyframe rotation threshold in degrees.  The reference's 30 deg / 0.3 m never trigger on a few hundred "
                         "frames of a 0.25 deg/frame orbit; scaled to the synthetic motion so that the keyframe list fills (>= 7) "
                         "as it does on a real sequence and every map update renders its 2 + 7 free views")
    ap.add_argument("--keyframe-trans", type=float, default=0.02)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle-psnr", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip config.other_configs (BASELINE configs[0], [1], [3] measured after the main windows at N = 1)")
    ap.add_argument("--whole-run-frames", type=int, default=1000,
                    help="frames of the whole-sequence run from frame 0 (the reference's own FPS definition; N = 1 only, after the "
                         "headline windows; 0 = skip): config.whole_run_fps / whole_run_fps_sequential")
    ap.add_argument("--no-workloads", action="store_true",
                    help="skip the two extra whole-sequence workloads (fine-texture detail run -> whole_run_gain_db; the reference's 30 deg / "
                         "0.3 m keyframe thresholds on a 3 deg/frame orbit -> whole_run_fps_ref_thresholds)")
    ap.add_argument("--full-line", action="store_true",
                    help="print the FULL record (tens of kilobytes) as the stdout line, as rounds 1-5 did, instead of the compact one; "
                         "the full record is written to bench_full.json either way")
    ap.add_argument("--gt-pose", action="store_true",
                    help="use_gt_pose: true (what every shipped config sets: the tracker is off, poses are given).  Default: the "
                         "depth-only ExtendedTracker estimates the pose of every frame, as BASELINE configs[2] "
                         "(\"full track + TSDF + Gaussian optimize\") describes")
    args = ap.parse_args(argv)

    from gps_slam_amd.dist_util import scene_seed
    rank, local_rank, world, grp, placement, device = setup_ranks(backend=backend, need_gpu=need_gpu)
    assert args.gpus == world, "--gpus %d but WORLD_SIZE is %d (launch with torch.distributed.run --nproc-per-node N)" % (args.gpus, world)
    dev_sync = torch.cuda.synchronize if need_gpu else _no_gpu_sync
    marker = (lambda: torch.cuda._sleep(1)) if need_gpu else _no_gpu_sync  # spin_kernel: phase marker for tools/prof_summary.py
    W, H, K, Wm, NW = args.width, args.height, args.steps, args.warmup, max(1, args.windows)
    first, prologue = timed_window(Wm)
    in_loop = extras and need_gpu   # one more K-frame window per schedule with every launch of the frame's kernels event-timed (roofline)
    n_frames = first + (NW + (1 if in_loop else 0)) * K
    seed = scene_seed(rank)
    seq = None
    if scene_factory is None:
        # (rendered on the device: the numpy ray-tracer of tests/synth.py needs 0.2-0.6 s per frame -- with N ranks on one host that
        # was 30-60 s per rank before the first barrier; same formulas, float64)
        seq = synthetic_sequence_device(W, H, n_frames, seed, device)
        seeds = seed_gaussians(seq, args.gaussians, seed, device)
        # One-off costs that are not part of any SLAM frame -- loading the code objects of every kernel, first-touch of the
        # allocator -- are paid here, before the warm-up frames.
        prime(device)
        scene_factory = lambda overlap: Scene(seq, seeds, seed, args.gt_pose, overlap=overlap, n_frames=n_frames,
                                              keyframe_theta=args.keyframe_theta, keyframe_trans=args.keyframe_trans)

    schedules = ("sequential", "overlap") if args.schedule == "both" else (args.schedule,)
    results, scene = {}, None
    for sched in schedules:
        if scene is not None:
            scene.close()
            del scene
            if need_gpu:
                torch.cuda.empty_cache()
        scene = scene_factory(sched == "overlap")
        dev_sync()
        scene.run(0, first)  # untimed: prologue + warm-up frames
        windows = []
        for w in range(NW):
            lo = first + w * K
            up0, st0 = scene.cli.uploadedBytes, dict(scene.pipe.stats())
            mallocs0 = _device_mallocs(need_gpu)
            dev_sync()
            grp.barrier()
            dev_sync()
            marker()  # start of a timed region
            t0 = time.perf_counter()
            scene.run(lo, lo + K)
            dev_sync()
            grp.barrier()
            dev_sync()
            dt = grp.max_over_ranks(time.perf_counter() - t0)
            marker()  # end of the timed region
            dev_sync()
            windows.append(dict(seconds=dt, frames_per_s=world * K / dt, ms_per_step=1000.0 * dt / K,
                                uploaded_bytes_per_frame=(scene.cli.uploadedBytes - up0) / K,
                                stats={k: int(v) - int(st0[k]) for k, v in dict(scene.pipe.stats()).items()},  # this window only
                                gaussians=int(scene.model.getGaussianNum()) if hasattr(scene, "model") else 0,
                                device_mallocs=_device_mallocs(need_gpu) - mallocs0))
        order = sorted(range(NW), key=lambda i: windows[i]["seconds"])
        med = windows[order[NW // 2]]  # the median window (upper median for an even count): every reported number is of ONE window
        ms = [w["ms_per_step"] for w in windows]
        if in_loop and rank == 0:
            from bench_kernels import in_loop_kernel_times
            lo = first + NW * K
            marker()   # (the in-loop window between its own pair of phase markers: tools/prof_summary.py tabulates it separately)
            med = dict(med, schedule=sched, in_loop=in_loop_kernel_times(lambda: scene.run(lo, lo + K), K))
            marker()
            dev_sync()
        results[sched] = dict(med, windows_ms_per_step=ms, window_spread=(max(ms) - min(ms)) / med["ms_per_step"],
                              window_spread_detrended=_detrended_spread(ms), windows_gaussians=[w["gaussians"] for w in windows],
                              windows_device_mallocs=[w["device_mallocs"] for w in windows])
    main_sched = "overlap" if "overlap" in results else schedules[0]
    dt = results[main_sched]["seconds"]

    if rank == 0:
        out = {
            "metric": "SLAM frames/sec @640x480, ~200k Gaussians; render PSNR vs ref",
            "value": world * K / dt, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": 1000.0 * dt / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"keyframe_thresholds": {"theta_deg": args.keyframe_theta, "trans_m": args.keyframe_trans,
                                               "reference": {"theta_deg": 30.0, "trans_m": 0.3}},
                       "schedule": main_sched, "windows": NW, "marker_windows_per_schedule": NW + (1 if in_loop else 0), "windows_ms_per_step": results[main_sched]["windows_ms_per_step"],
                       "window_spread": results[main_sched]["window_spread"],
                       "window_spread_detrended": results[main_sched]["window_spread_detrended"],
                       "windows_gaussians": results[main_sched]["windows_gaussians"],
                       "window_note": "%d consecutive %d-step windows per schedule, each bracketed by barrier + synchronize and "
                                      "maxed over ranks; value / ms_per_step = the MEDIAN window; the scene grows from window to window "
                                      "(windows_gaussians), window_spread_detrended is the spread around that trend" % (NW, K),
                       "local_opt_interval": PERIOD, "local_opt_iters": int(os.environ.get("GPS_BENCH_OPT_ITERS", 20)), "frames_per_step": 1,
                       "use_gt_pose": bool(args.gt_pose), "prologue_frames": prologue,
                       "schedules": {k: {kk: vv for kk, vv in v.items() if kk != "seconds"} for k, v in results.items()},
                       "stats": results[main_sched]["stats"], "placement": placement},
        }
        flat = {"keyframe_theta_deg": args.keyframe_theta, "keyframe_trans_m": args.keyframe_trans, "env_overrides": env_overrides()}
        if "sequential" in results:
            flat["sequential_fps"] = world * K / results["sequential"]["seconds"]
        if "overlap" in results:
            flat["overlap_fps"] = world * K / results["overlap"]["seconds"]
        def leg(name, fn):
            """one measurement leg behind the timed windows: a failure there is reported in the line (`<name>_error`, traceback on
            stderr) instead of costing the line -- and the other ranks their barrier"""
            try:
                fn()
            except Exception as e:   # noqa: BLE001
                import traceback
                traceback.print_exc(file=sys.stderr)
                flat["%s_error" % name] = ("%s: %s" % (type(e).__name__, e))[:200]

        def leg_roofline():
            out["config"].update(_describe_and_measure(args, scene, seq, dict(results[main_sched], in_loop_sequential=(results.get("sequential") or {}).get("in_loop")),
                                                        first, K, dt, marker))
            out["roofline"] = out["config"].pop("roofline")
            flat["frame_frac"] = out["roofline"]["frame"]["frac"]
            flat["iteration_frac"] = out["roofline"]["iteration"]["frac"]
            flat["iteration_us"] = out["roofline"]["iteration"]["avg_us"]

        if extras:
            leg("roofline", leg_roofline)
            if world == 1 and (not args.no_other_configs or args.whole_run_frames > 0):
                scene.close()
                scene = None
                torch.cuda.empty_cache()
            # (the whole-sequence run first: it is the metric as the reference defines it)
            def leg_whole_run():
                t_w = time.perf_counter()
                wr = whole_run(args, seed, device, args.whole_run_frames)
                wr["seconds_total"] = time.perf_counter() - t_w
                out["config"]["whole_run"] = wr
                flat.update(whole_run_fps=wr["overlap"]["fps"], whole_run_fps_sequential=wr["sequential"]["fps"],
                            whole_run_frames=wr["frames"], whole_run_gaussians_end=wr["overlap"]["gaussians_end"],
                            whole_run_fusion_fps=wr["overlap"]["fusion_fps"], whole_run_gaussian_fps=wr["overlap"]["gaussian_fps"],
                            whole_run_fusion_fps_sequential=wr["sequential"]["fusion_fps"],
                            whole_run_gaussian_fps_sequential=wr["sequential"]["gaussian_fps"],
                            whole_run_slowest_frame_ms=wr["overlap"]["slowest_frame_ms_after_30"], whole_run_gpu_memory_mb=wr["overlap"]["gpu_memory_mb"],
                            whole_run_seconds_total=wr["seconds_total"])

            def leg_workloads():
                t_w = time.perf_counter()
                dr = detail_run(args, seed, device)
                # (every shipped config sets use_gt_pose: true; the headline run tracks -- both pose sources on the same sequence)
                dg = detail_run(args, seed, device, gt_pose=not args.gt_pose)
                rr = ref_threshold_run(args, seed, device)
                out["config"]["detail_run"], out["config"]["detail_run_other_pose_source"], out["config"]["ref_threshold_run"] = dr, dg, rr
                flat.update(whole_run_gain_db=dr["gain_db"], whole_run_detail_render_psnr_db=dr["render_psnr_db"],
                            whole_run_detail_tsdf_psnr_db=dr["tsdf_colour_psnr_db"], whole_run_detail_gaussians_end=dr["gaussians_end"],
                            **{"whole_run_gain_db_%s" % ("gt_pose" if dg["use_gt_pose"] else "tracked"): dg["gain_db"]},
                            whole_run_fps_ref_thresholds=rr["overlap"]["fps"], whole_run_fps_ref_thresholds_sequential=rr["sequential"]["fps"],
                            workloads_seconds=time.perf_counter() - t_w)

            def leg_other_configs():
                oc = other_configs(args, seq, seed, device, first)
                out["config"]["other_configs"] = oc
                flat.update(cfg0_fps=oc["configs0_tsdf_only_gt_pose"]["frames_per_s"],
                            cfg1_fps=oc["configs1_gt_pose_100k"]["frames_per_s"],
                            cfg1_fps_sequential=oc["configs1_gt_pose_100k"]["schedules"]["sequential"]["frames_per_s"],
                            cfg1_iters_per_s=oc["configs1_gt_pose_100k"]["iterations_per_s"],
                            cfg3_fps=oc["configs3_720p_400k"]["frames_per_s"],
                            cfg3_fps_sequential=oc["configs3_720p_400k"]["schedules"]["sequential"]["frames_per_s"],
                            cfg3_gaussians=oc["configs3_720p_400k"]["schedules"]["overlap"]["gaussians"],
                            cfgR_fps=oc["configsR_replica_1200x680_300k"]["frames_per_s"],
                            cfgR_fps_sequential=oc["configsR_replica_1200x680_300k"]["schedules"]["sequential"]["frames_per_s"],
                            cfgR_gaussians=oc["configsR_replica_1200x680_300k"]["schedules"]["overlap"]["gaussians"],
                            other_configs_seconds=oc["seconds"])

            def leg_cpu_baseline():
                from bench_kernels import cpu_baseline
                out["cpu_baseline"] = cpu_baseline(seq, W, H)

            if world == 1 and args.whole_run_frames > 0:
                leg("whole_run", leg_whole_run)
            if world == 1 and args.whole_run_frames > 0 and not args.no_workloads:
                leg("workloads", leg_workloads)
            if world == 1 and not args.no_other_configs:
                leg("other_configs", leg_other_configs)
            if not args.no_cpu_baseline and world == 1:
                leg("cpu_baseline", leg_cpu_baseline)
        # flat scalars FIRST in `config`: the driver's record keeps scalar fields only -- every number that matters beside `value`
        # (the reference's own whole-run FPS, the sequential schedule, the other single-GPU configurations, the keyframe
        # thresholds, the frame / iteration HBM fractions, which A/B switches were set) is one of them
        out["config"] = dict(flat, **out["config"])
        emit(out, full_line=args.full_line, side_dir=os.environ.get("GPS_BENCH_SIDE_DIR"))
    # ranks != 0 wait here while rank 0 runs its post-window measurements and prints: no rank tears the process group down
    # (or exits, which torch.distributed.run treats as the job ending) under another rank's feet
    grp.barrier()
    if scene is not None:
        scene.close()
    grp.close()


LINE_LIMIT = 4096   # bytes of the stdout line: the driver's record of round 5 (23 KB line) came back unparsed, round 4's 16 KB parsed


def _finite(o, path="line"):
    """every float of the record is finite: json.dumps would print NaN / Infinity, which are not JSON and which a strict parser
    (the driver's) rejects together with the whole line"""
    if isinstance(o, float):
        assert o == o and abs(o) != float("inf"), "%s is %r: not representable in JSON" % (path, o)
    elif isinstance(o, dict):
        for k, v in o.items():
            _finite(v, "%s.%s" % (path, k))
    elif isinstance(o, (list, tuple)):
        for i, v in enumerate(o):
            _finite(v, "%s[%d]" % (path, i))


def _sig(v, digits=6):
    """floats to 6 significant digits (the line is for reading and for the driver's record; bench_full.json keeps every bit)"""
    if isinstance(v, float):
        return float("%.*g" % (digits, v))
    return v


def compact_line(out):
    """The stdout line of the contract, <= LINE_LIMIT bytes: the headline fields, `config` = workload + the flat scalars (the
    reference's own whole-run FPS, both schedules, the other single-GPU configurations, thresholds, A/B switches), `roofline` =
    the dominant kernel's in-loop figures + the frame / iteration fractions, `cpu_baseline` = the reference CPU engine's rate.
    Everything nested (per-kernel tables, per-window lists, notes) stays in bench_full.json."""
    line = {k: _sig(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                      "vs_baseline", "dtype", "data")}
    cfg = out.get("config", {})
    c = {"workload": cfg.get("workload", "synthetic RGB-D SLAM frames, independent scene per GPU")}
    for k, v in cfg.items():
        if k != "workload" and isinstance(v, (int, float, bool)) or (isinstance(v, str) and (k in ("schedule", "host", "env_overrides", "placement") or k.endswith("_error"))):
            c[k] = _sig(v)
    q = cfg.get("quality") or {}
    for k in ("render_psnr_db_vs_input", "tsdf_colour_psnr_db_vs_input", "render_psnr_db_vs_oracle"):
        if isinstance(q.get(k), (int, float)):
            c[k] = _sig(float(q[k]), 5)
    c["full_record"] = "bench_full.json"
    line["config"] = c
    r = out.get("roofline")
    if r:
        keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "avg_launch_us", "timed_in",
                "launches_timed", "frac_alone", "avg_launch_us_alone", "frac_sequential", "avg_launch_us_sequential", "measured_copy_GBs", "traffic_calibrated", "traffic_library_commit")
        rr = {k: _sig(r[k]) for k in keep if k in r}
        rr["frame_frac"], rr["iteration_frac"] = _sig(r["frame"]["frac"]), _sig(r["iteration"]["frac"])
        rr["frame_ms"], rr["iteration_us"] = _sig(r["frame"]["ms"]), _sig(r["iteration"]["avg_us"])
        line["roofline"] = rr
    b = out.get("cpu_baseline")
    if b:
        bb = {k: _sig(b[k]) for k in ("value", "unit", "cores", "threads", "kind", "tracked_value", "cpu_quota") if k in b}
        bb["sample"] = str(b.get("sample", ""))[:300]
        line["cpu_baseline"] = bb
    return line


def emit(out, full_line=False, side_dir=None):
    """rank 0: the full record to bench_full.json (beside this file; also under gpurun_out/ when that exists, so that a gpurun
    call brings it back), ONE compact JSON line to stdout as the last thing printed there"""
    _finite(out)
    full = json.dumps(out, allow_nan=False)
    for d in ([side_dir] if side_dir else [ROOT, os.path.join(ROOT, "gpurun_out")]):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_full.json"), "w") as f:
                    f.write(full + "\n")
            except OSError as e:   # (a read-only checkout must not cost the line)
                print("bench_full.json not written in %s: %s" % (d, e), file=sys.stderr)
    # (the full record is NOT echoed to stderr: a driver that captures stdout and stderr into one buffer would then see a 20+ KB
    # line next to -- possibly after -- the compact one)
    print("bench: full record (%d bytes) in bench_full.json" % len(full), file=sys.stderr, flush=True)
    text = full if full_line else json.dumps(compact_line(out), allow_nan=False, separators=(",", ":"))
    assert full_line or len(text.encode()) <= LINE_LIMIT, "compact line is %d bytes (> %d)" % (len(text.encode()), LINE_LIMIT)
    sys.stdout.flush()
    print(text, flush=True)


def _time_scene(factory, first, K, NW):
    """median-of-NW-windows frames/s of one scene per schedule (world size 1: no barriers), as main() times the headline scene"""
    out = {}
    for sched in ("sequential", "overlap"):
        scene = factory(sched == "overlap")
        torch.cuda.synchronize()
        scene.run(0, first)
        ms = []
        for w in range(NW):
            lo = first + w * K
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            scene.run(lo, lo + K)
            torch.cuda.synchronize()
            ms.append(1000.0 * (time.perf_counter() - t0) / K)
        out[sched] = {"frames_per_s": 1000.0 / sorted(ms)[len(ms) // 2], "windows_ms_per_step": ms,
                      "gaussians": int(scene.model.getGaussianNum())}
        last = scene
        if sched == "sequential":
            scene.close()
            del scene
            torch.cuda.empty_cache()
    return out, last


def whole_run(args, seed, device, n_frames):
    """The metric as the reference defines it (slam/slam_pipeline.cpp:52-173 with LOG_PIPELINE_TIME): SLAMTrainCams over the WHOLE
    sequence from frame 0 with an EMPTY model -- the first keyframe update (the whole first view sampled into Gaussians + its KNN),
    the keyframe list filling up, the hash table filling up, the growth of N with every newly seen surface are all inside --
    FPS = frames / wall (:162-167; here the clock stops after flush() and a device synchronise), and the Fusion / Gaussian split
    run/read_results.py:38-39 derives from "per frame fusion time".  Both schedules on the same sequence (device-rendered, held
    as uint8 / uint16 like the dataset's files); N = 1 only, after the headline windows."""
    W, H = args.width, args.height
    t0 = time.perf_counter()
    seq = synthetic_sequence_device(W, H, n_frames, seed, device)
    t_gen = time.perf_counter() - t0
    res = {"frames": n_frames, "size": "%dx%d" % (W, H), "input_render_seconds": t_gen,
           "what": "SLAMPipeline::SLAMTrainCams from frame 0, empty model, %s, keyframe thresholds %.3g deg / %.3g m; "
                   "frames / wall with the clock stopped after flush() + device synchronise"
                   % ("given poses" if args.gt_pose else "depth ICP tracking", args.keyframe_theta, args.keyframe_trans)}
    for sched in ("sequential", "overlap"):
        t0 = time.perf_counter()
        sc = Scene(seq, None, seed, args.gt_pose, overlap=sched == "overlap", n_frames=n_frames,
                   keyframe_theta=args.keyframe_theta, keyframe_trans=args.keyframe_trans)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        mallocs0 = _device_mallocs(True)
        torch.cuda._sleep(1)   # (phase marker for kernel traces: tools/probe/queues.sh)
        tm = sc.pipe.SLAMTrainCamsTimed(sc.model, sc.cams)
        torch.cuda._sleep(1)
        st = {k: int(v) for k, v in dict(sc.pipe.stats()).items()}
        res[sched] = {"fps": tm.fps(), "seconds": tm.slam_total * 1e-3, "fusion_fps": tm.fusion_fps(), "gaussian_fps": tm.gaussian_fps(),
                      "per_frame_fusion_ms": tm.per_frame / max(1, tm.frames), "keyframe_step_host_ms_per_frame": tm.keyframe_step / max(1, tm.frames),
                      "stage_ms_per_frame": {k: getattr(tm, k) / max(1, tm.frames) for k in
                                             ("localFrameRaycast", "keyFrameRaycast", "initNewGaussians", "localOptimize", "removeGaussian")},
                      "slowest_frame_ms_after_30": tm.max_frame_after_30, "slowest_frame_id": tm.max_frame_id, "gpu_memory_mb": int(tm.gpu_memory_mb),
                      "gaussians_end": int(sc.model.getGaussianNum()), "pipeline_stats": st, "device_mallocs": _device_mallocs(True) - mallocs0,
                      "visible_blocks_end": int(sc.engine.counters().cpu()[2]), "allocated_blocks_end": int((1 << 18) - 1 - int(sc.engine.counters().cpu()[0])),
                      "scene_build_seconds": t_build}
        if not args.gt_pose:
            fr, ev, rode, used = sc.engine.trackerTotals()
            res[sched]["tracker_evaluations_per_frame"] = ev / max(1, fr)
            # tracked against ground truth: the final estimated pose of the orbit (camera 0 is the world frame)
            est = sc.engine.lastPose()[1].reshape(4, 4).T.numpy().astype(np.float64)   # invM, column-major -> c2w
            gt = seq["c2w"][n_frames - 1].astype(np.float64)
            res[sched]["final_pose_trans_err_mm"] = float(np.linalg.norm(est[:3, 3] - gt[:3, 3]) * 1e3)
        sc.close()
        del sc
        torch.cuda.empty_cache()
    return res


def _held_out_psnr(sc, seq, device, frames):
    """render and TSDF-colour PSNR against the input image on `frames` (never optimise cameras), from their tracked poses, on the
    scene's final state (tools/convergence.py does the same per keyframe update): -> (render dB, TSDF colour dB), means"""
    def _psnr(a, b):
        return float(-10.0 * torch.log10(((a.clamp(0, 1) - b) ** 2).mean()))
    r, t = [], []
    with torch.no_grad():
        for j in frames:
            cam = sc.cams[j]
            rc = sc.pipe.runRaycastByCam(cam, False)
            img = torch.as_tensor(seq["rgb"][j]).to(device).float() / 255.0
            cam.image = img
            cam.toGPU()
            r.append(_psnr(sc.model.forward(cam, rc["depth_map"], rc["color_map"])["rgb"], img))
            t.append(_psnr(rc["color_map"], img))
    return sum(r) / len(r), sum(t) / len(t)


def detail_run(args, seed, device, n_frames=300, gt_pose=None):
    """A workload where the Gaussians have something to add (round-5 review, item 8a): the room scaled to 0.4 (the camera 0.4-1.5 m
    from the surfaces: a pixel's footprint is below the 5 mm voxel) with the `fine` texture (12 / 15 / 19 mm gratings, no hard
    steps), SLAMTrainCams from an empty model with the bench's keyframe thresholds, sequential schedule.  Afterwards the held-out
    frames of the last three keyframe periods (two per period; the local window takes every 5th frame and keyframes are picked
    from those) are rendered from their tracked poses: gain = render PSNR - TSDF-colour PSNR against the input images.  (The
    most recent periods, as tools/convergence.py evaluates after every update -- profiles/r06_convergence_detail.md has the whole
    curve: the gain grows with the number of updates a region has been through, + 0.1 dB at frame 59, + 2.4 at 199, + 3.7 at 299;
    frames of a hundred frames ago are rendered from poses and Gaussians that have since moved on: + 1.9 dB over the last ten periods.)"""
    W, H = args.width, args.height
    gt_pose = args.gt_pose if gt_pose is None else gt_pose
    seq = synthetic_sequence_device(W, H, n_frames, seed, device, texture="fine", world_scale=0.4)
    sc = Scene(seq, None, seed, gt_pose, overlap=False, n_frames=n_frames, keyframe_theta=args.keyframe_theta, keyframe_trans=args.keyframe_trans)
    torch.cuda.synchronize()
    tm = sc.pipe.SLAMTrainCamsTimed(sc.model, sc.cams)
    last_kf = (n_frames - 1) // PERIOD * PERIOD
    held = [k for p0 in range(max(PERIOD, last_kf - 2 * PERIOD), last_kf + 1, PERIOD) for k in (p0 + 3, p0 + 7) if k < n_frames]
    r_db, t_db = _held_out_psnr(sc, seq, device, held)
    out = {"frames": n_frames, "size": "%dx%d" % (W, H), "use_gt_pose": bool(gt_pose), "fps_sequential": tm.fps(), "gaussians_end": int(sc.model.getGaussianNum()),
           "held_out_frames": len(held), "render_psnr_db": r_db, "tsdf_colour_psnr_db": t_db, "gain_db": r_db - t_db,
           "what": "room x 0.4, `fine` texture (12 / 15 / 19 mm sine gratings), empty model, sequential schedule; PSNR of the render and of the "
                   "TSDF colour against the input on %d held-out frames of the last three keyframe periods, %s poses"
                   % (len(held), "given" if gt_pose else "tracked")}
    sc.close()
    del sc
    torch.cuda.empty_cache()
    return out


def ref_threshold_run(args, seed, device, n_frames=300, step_deg=3.0):
    """The reference's OWN keyframe thresholds (keyframe_theta_thres 30 deg, keyframe_trans_thres 0.3 m: configs/release/replica/
    office0.yaml:54-55) on an orbit fast enough for them to trigger (3 deg per frame: a keyframe every ~10 frames), SLAMTrainCams
    from an empty model, both schedules (round-5 review, item 8b)."""
    W, H = args.width, args.height
    seq = synthetic_sequence_device(W, H, n_frames, seed, device, step_deg=step_deg)
    out = {"frames": n_frames, "size": "%dx%d" % (W, H), "orbit_deg_per_frame": step_deg, "keyframe_theta_deg": 30.0, "keyframe_trans_m": 0.3}
    for sched in ("sequential", "overlap"):
        sc = Scene(seq, None, seed, args.gt_pose, overlap=sched == "overlap", n_frames=n_frames, keyframe_theta=30.0, keyframe_trans=0.3)
        torch.cuda.synchronize()
        tm = sc.pipe.SLAMTrainCamsTimed(sc.model, sc.cams)
        st = {k: int(v) for k, v in dict(sc.pipe.stats()).items()}
        out[sched] = {"fps": tm.fps(), "gaussians_end": int(sc.model.getGaussianNum()), "keyframes": int(sc.pipe.keyframeCount()) if hasattr(sc.pipe, "keyframeCount") else None,
                      "pipeline_stats": st}
        if sched == "overlap" and not args.gt_pose:
            # the tracked trajectory against the given one at the last frame (the fast orbit is a harder tracking problem)
            fr, ev, rode, used = sc.engine.trackerTotals()
            out["tracker_evaluations_per_frame"] = ev / max(1, fr)
        sc.close()
        del sc
        torch.cuda.empty_cache()
    return out


def other_configs(args, seq, seed, device, first):
    """The other single-GPU configurations of BASELINE.json, measured after the headline windows on the same box (N = 1 only;
    bounded: three 20-step windows per schedule, device-rendered input):
      configs[0]  TSDF-only `recon` loop with given poses, HIP engine (the cpu_baseline's counterpart)
      configs[1]  640x480, use_gt_pose=true, ~100 k Gaussians: frames/s of the loop and optimise iterations/s (fwd + L1 + bwd + Adam)
      configs[3]  1280x720 (Azure Kinect intrinsics), tracking on, ~400 k Gaussians: both schedules, its own roofline units"""
    from bench_kernels import config_units, fusion_split
    t_all = time.perf_counter()
    K, NW = 20, 3
    res = {"note": "N = 1, after the headline windows, same box; %d windows x %d steps per schedule, median; timed step 0 is frame %d "
                   "(a keyframe) as in the headline run" % (NW, K, first)}
    kf = dict(keyframe_theta=args.keyframe_theta, keyframe_trans=args.keyframe_trans)
    # configs[0]
    t0 = time.perf_counter()
    r0 = fusion_split(seq, first, 2 * K, True, 1.0)
    res["configs0_tsdf_only_gt_pose"] = {"frames_per_s": r0["fusion_fps"], "ms_per_frame": r0["fusion_ms_per_frame"], "size": "%dx%d" % (seq["W"], seq["H"]),
                                         "what": "upload + TSDF fuse + live raycast + ICP maps, HIP engine, given poses, no Gaussians (work_mode recon)",
                                         "seconds": time.perf_counter() - t0}
    # configs[1]
    t0 = time.perf_counter()
    n1 = first + NW * K
    seeds1 = seed_gaussians(seq, 100000, seed, device)
    r1, sc = _time_scene(lambda ov: Scene(seq, seeds1, seed, True, overlap=ov, n_frames=n1, **kf), first, K, NW)
    cam, rc = sc.pipe.optCams()[-1], sc.pipe.optRaycasts()[-1]
    sc.pipe.flush()
    sc.model.initOptimizers(-1, 1.0)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sc.model.trainStep(cam, rc["depth_map"], rc["color_map"])
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(50):   # (as the pipeline's loop runs them: the next iteration's preprocessing in this one's backward kernel)
        sc.model.trainStep(cam, rc["depth_map"], rc["color_map"], None, cam)
    ev1.record()
    torch.cuda.synchronize()
    it_s = ev0.elapsed_time(ev1) * 1e-3 / 50
    roof1 = config_units(sc, seq, 1000.0 / r1["overlap"]["frames_per_s"], HBM_PEAK_GBS)
    sc.close()
    del sc
    torch.cuda.empty_cache()
    res["configs1_gt_pose_100k"] = {"roofline": roof1, "size": "%dx%d" % (seq["W"], seq["H"]), "schedules": r1, "frames_per_s": r1["overlap"]["frames_per_s"],
                                    "iterations_per_s": 1.0 / it_s, "iteration_us": it_s * 1e6,
                                    "what": "use_gt_pose=true (tracker off), ~100 k Gaussians; iterations = forward + L1 + backward + fused Adam "
                                            "of one optimise camera, 50 back-to-back (HIP events)",
                                    "seconds": time.perf_counter() - t0}
    # configs[3]
    t0 = time.perf_counter()
    W3, H3 = 1280, 720
    n3 = first + NW * K
    seq3 = synthetic_sequence_device(W3, H3, n3, seed, device, intrinsics=(605.0, 605.0, 635.3, 366.5))
    t_gen = time.perf_counter() - t0
    seeds3 = seed_gaussians(seq3, 400000, seed, device)
    r3, sc = _time_scene(lambda ov: Scene(seq3, seeds3, seed, False, overlap=ov, n_frames=n3, **kf), first, K, NW)
    st = dict(sc.pipe.stats())
    roof3 = config_units(sc, seq3, 1000.0 / r3["overlap"]["frames_per_s"], HBM_PEAK_GBS)
    sc.close()
    del sc
    torch.cuda.empty_cache()
    res["configs3_720p_400k"] = {"roofline": roof3,"size": "%dx%d" % (W3, H3), "schedules": r3, "frames_per_s": r3["overlap"]["frames_per_s"],
                                 "what": "Azure-Kinect-like 720p intrinsics (fx = fy = 605), depth ICP tracking + TSDF fuse + ges splat optimise, "
                                         "~400 k Gaussians", "pipeline_stats": {k: int(v) for k, v in st.items()},
                                 "input_render_seconds": t_gen, "seconds": time.perf_counter() - t0}
    # Replica-native geometry: every Replica config is 1200x680, f = 600, c = (599.5, 339.5) (configs/release/replica/office0.yaml:18-20):
    # 75 x 43 tiles with a ragged last tile row (680 = 42 x 16 + 8), 13 sort bits
    t0 = time.perf_counter()
    WR, HR = 1200, 680
    nR = first + NW * K
    seqR = synthetic_sequence_device(WR, HR, nR, seed, device, intrinsics=(600.0, 600.0, 599.5, 339.5))
    t_gen = time.perf_counter() - t0
    seedsR = seed_gaussians(seqR, 300000, seed, device)
    rR, sc = _time_scene(lambda ov: Scene(seqR, seedsR, seed, False, overlap=ov, n_frames=nR, **kf), first, K, NW)
    st = dict(sc.pipe.stats())
    roofR = config_units(sc, seqR, 1000.0 / rR["overlap"]["frames_per_s"], HBM_PEAK_GBS)
    sc.close()
    del sc
    torch.cuda.empty_cache()
    res["configsR_replica_1200x680_300k"] = {"roofline": roofR, "size": "%dx%d" % (WR, HR), "schedules": rR, "frames_per_s": rR["overlap"]["frames_per_s"],
                                             "what": "Replica's camera (1200x680, fx = fy = 600, c = (599.5, 339.5)), depth ICP tracking + TSDF fuse + "
                                                     "ges splat optimise, ~300 k Gaussians", "pipeline_stats": {k: int(v) for k, v in st.items()},
                                             "input_render_seconds": t_gen, "seconds": time.perf_counter() - t0}
    res["seconds"] = time.perf_counter() - t_all
    return res


def _describe_and_measure(args, scene, seq, result, first, K, dt, marker):
    """rank 0, after the timed windows: workload description, render quality against the oracle, the Fusion / Gaussian split
    and the roofline section (all measured on the state the timed run ended in)."""
    from bench_kernels import roofline_section, fusion_split, render_psnr_vs_oracle
    W, H = args.width, args.height
    N = scene.model.getGaussianNum()
    views = list(zip(scene.pipe.optCams(), scene.pipe.optRaycasts()))[-5:]

    def _psnr(a, b):
        return float(-10.0 * torch.log10(((a.clamp(0, 1) - b) ** 2).mean()))
    with torch.no_grad():
        psnr_render = [_psnr(scene.model.forward(c, rc["depth_map"], rc["color_map"])["rgb"], c.image) for c, rc in views]
        psnr_tsdf = [_psnr(rc["color_map"], c.image) for c, rc in views]
    quality = {"views": len(views), "render_psnr_db_vs_input": sum(psnr_render) / max(1, len(views)),
               "tsdf_colour_psnr_db_vs_input": sum(psnr_tsdf) / max(1, len(views))}
    if not args.no_oracle_psnr and views:
        quality.update(render_psnr_vs_oracle(scene.model, views[-1][0], views[-1][1], seq))
    tracker = None
    if not args.gt_pose:   # the LM loop of the whole run of this scene (prologue + warm-up + timed windows)
        fr, ev, rode, used = scene.engine.trackerTotals()
        tracker = {"poses_riding_along": scene.engine.posesRidingAlong(), "evaluations_per_frame": ev / max(1, fr),
                   "rode_along_per_frame": rode / max(1, fr), "consumed_per_frame": used / max(1, fr),
                   "note": "with every evaluation the pose the LM loop would take next after a REJECTION is evaluated too (known "
                           "beforehand: a rejection reads nothing of the evaluation it rejects); consumed = evaluations answered "
                           "without another host <-> device round trip"}
    marker()  # phase marker: everything below is measurement scaffolding, not SLAM frames
    split = fusion_split(seq, first, K, args.gt_pose, dt)
    roof = roofline_section(scene, seq, result, HBM_PEAK_GBS, K, gt_pose=args.gt_pose)
    return {"workload": "synthetic room0-like RGB-D %dx%d: per-frame upload (6 B/px) + %s + TSDF fuse (5mm voxels) + ges "
                        "splat optimise, ~%dk Gaussians; independent scene per GPU"
                        % (W, H, "given poses (use_gt_pose=true, as every shipped config)" if args.gt_pose else
                           "depth ICP tracking (ExtendedTracker, use_gt_pose=false)", N // 1000),
            "gaussians": N, "host": "cpp (createTsdfEngine -> CLIEngine -> SLAMPipeline)", "quality": quality, "split": split,
            "tracker": tracker,
            "roofline": roof}


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py -- SLAM frames/s of the GPS-SLAM hot path on MI355X.

One "step" = one SLAM frame of SLAMPipeline::SLAMTrainCams (slam/slam_pipeline.cpp:69-132) on synthetic
640x480 RGB-D with ~200k Gaussians: TSDF fuse + live raycast every frame; every 10th frame the <= 9 free-view
raycasts, new-Gaussian sampling, 20 optimise iterations (forward, L1, backward, Adam) and the prune.
Inputs (rgb uint8, depth int16 mm, GT poses) are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W
N > 1: launched by torch.distributed.run, one rank per GPU, each rank runs an INDEPENDENT scene (different seed)
-> weak scaling, no data-path collective; barrier + max-over-ranks timing only.
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


def build_scene(W, H, n_frames, n_gauss, seed, device):
    """Synthetic room sequence + a Gaussian model pre-populated on the scene surfaces."""
    from tests import synth
    from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
    from gps_slam_amd.slam_pipeline import SLAMPipeline, compute_normal_map
    from gps_slam_amd.tsdf_engine import TsdfEngine
    seq = synth.make_sequence(W, H, n_frames, step_deg=0.25 + 0.01 * (seed % 7))
    # world := first camera (what a tracked run uses as its world, ITMTrackingState::Reset starts at the identity), so
    # that the given poses and the tracked poses live in the same frame
    c0inv = np.linalg.inv(seq["c2w"][0].astype(np.float64))
    seq["c2w"] = np.stack([(c0inv @ c.astype(np.float64)) for c in seq["c2w"]]).astype(np.float32)
    fx, fy, cx, cy = seq["fx"], seq["fy"], seq["cx"], seq["cy"]
    eng = TsdfEngine(W, H, fx, fy, cx, cy, voxel_size=0.005, mu=0.02, view_frustum_min=0.2, view_frustum_max=10.0,
                     device=device)
    model = SLAMGaussianModel(dict(isect_capacity=8 << 20), device=device)
    pipe = SLAMPipeline(eng, model, seed=seed)
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)  # uchar4 frames
    rgb_dev = torch.as_tensor(rgba).to(device)
    depth_dev = torch.as_tensor(seq["depth"].astype(np.int16)).to(device)
    cams = []
    for k in range(n_frames):
        img = rgb_dev[k][..., :3].float() / 255.0
        dep = (depth_dev[k].float() / 1000.0).unsqueeze(-1)
        cams.append(Camera(k, W, H, fx, fy, cx, cy, seq["c2w"][k], image=img, depth=dep, device=device))
    # pre-populate: back-project a few views' depth to world points (colour from the image, normals from Sobel)
    g = torch.Generator(device=device).manual_seed(seed)
    pts, cols, nrm = [], [], []
    ys, xs = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device), indexing="ij")
    for k in range(0, n_frames, max(1, n_frames // 6)):
        d = cams[k].depth[..., 0]
        c2w = cams[k].c2w.to(device)
        pc = torch.stack([(xs - cx) / fx * d, (ys - cy) / fy * d, d], -1)
        pw = pc @ c2w[:3, :3].T + c2w[:3, 3]
        n = compute_normal_map(pw)
        ok = (d > 0.3).reshape(-1)
        pts.append(pw.reshape(-1, 3)[ok]); cols.append(cams[k].image.reshape(-1, 3)[ok]); nrm.append(n.reshape(-1, 3)[ok])
    pts, cols, nrm = torch.cat(pts), torch.cat(cols), torch.cat(nrm)
    sel = torch.randperm(pts.shape[0], device=device, generator=g)[:n_gauss].sort().values  # (view, pixel) order, as addGaussians appends
    new = model.init_params(pts[sel].contiguous(), cols[sel].contiguous(), nrm[sel].contiguous())
    # view-dependent detail so all 16 SH bands carry signal
    new["featuresRest"] = (torch.randn(new["featuresRest"].shape, device=device, generator=g) * 0.02).contiguous()
    model.add_params(new)
    return seq, eng, model, pipe, cams, rgb_dev, depth_dev


def prime(host, device):
    """Run the whole per-frame loop once on a tiny throwaway scene (loads every kernel, warms the allocator)."""
    from tests import synth
    from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
    from gps_slam_amd.slam_pipeline import SLAMPipeline
    from gps_slam_amd.tsdf_engine import TsdfEngine
    W, H, n = 64, 48, 21
    seq = synth.make_sequence(W, H, n, step_deg=0.5)
    rgba = np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)
    rgb = torch.as_tensor(rgba).to(device)
    dep = torch.as_tensor(seq["depth"].astype(np.int16)).to(device)
    if host == "cpp":
        import gps_slam_amd._host as H_
        eng = H_.ITMBasicEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.02, 0.08, 0.2, 10.0)
        model = H_.SLAMGaussianModel()
        model.loadConfig(dict(capacity=1 << 14))
        pipe = H_.SLAMPipeline(eng, model, 1)
    else:
        eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel_size=0.02, mu=0.08, device=device)
        pipe = SLAMPipeline(eng, SLAMGaussianModel(dict(capacity=1 << 14), device=device), seed=1)
    for i in range(n):
        img = rgb[i][..., :3].float() / 255.0
        d = (dep[i].float() / 1000.0).unsqueeze(-1)
        if host == "cpp":
            import gps_slam_amd._host as H_
            c = H_.Camera(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][i].astype(np.float32)))
            c.id, c.image, c.depth = i, img, d
            pipe.processFrame(i, c, rgb[i], dep[i])
        else:
            c = Camera(i, W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], seq["c2w"][i], image=img, depth=d, device=device)
            pipe.process_frame(i, c, rgb[i], dep[i])
    torch.cuda.synchronize()


def iteration_bytes(N, Nv, I, G, P, T):
    """Algorithmic (compulsory) HBM bytes of one optimise iteration, SURVEY.md 8(d)."""
    return (68 * N + 217 * Nv + 24 * N + 44 * I + 8 * G + 4 * T + 44 * I + 28 * P + 40 * P + 52 * G + 24 * P + 40 * G +
            408 * Nv + 116 * Nv + 40 * N + 28 * 59 * N)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-map-thread", action="store_true", help="C++ host, with overlap: interleave the map update's host work "
                    "with the frames on one thread instead of giving it a worker thread")
    ap.add_argument("--no-overlap", action="store_true", help="C++ host: run the keyframe map update to completion before the "
                    "next frame (the reference's schedule) instead of overlapping it with tracking/fusion on a second stream")
    ap.add_argument("--gaussians", type=int, default=200000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gt-pose", action="store_true",
                    help="use_gt_pose: true (what every shipped config sets: the tracker is off, poses are given).  Default: the "
                         "depth-only ExtendedTracker estimates the pose of every frame, as BASELINE configs[2] "
                         "(\"full track + TSDF + Gaussian optimize\") describes")
    ap.add_argument("--host", choices=("cpp", "python"), default="cpp",
                    help="which host layer drives the C-ABI: the C++/libtorch one (gps_slam_amd/host) or its Python mirror")
    args = ap.parse_args()

    from gps_slam_amd.dist_util import Group, env_ranks, scene_seed
    rank, local_rank, world = env_ranks()
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    grp = Group(backend="nccl", device=device)  # RCCL; used for barrier + max-over-ranks only

    W, H, K, Wm = args.width, args.height, args.steps, args.warmup
    n_frames = K + Wm + 1
    seq, eng, model, pipe, cams, rgb_dev, depth_dev = build_scene(W, H, n_frames, args.gaussians, scene_seed(rank), device)
    pipe.use_gt_pose = args.gt_pose
    if not args.gt_pose:
        eng.turnOnTracking()

    if args.host == "cpp":
        # same scene, driven by the C++ host layer (what a C++ slam_trainer links against); the Python objects built
        # above only supply the synthetic inputs and the initial Gaussians
        import gps_slam_amd._host as H_
        ceng = H_.ITMBasicEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, 0.2, 10.0)
        cmodel = H_.SLAMGaussianModel()
        cmodel.loadConfig(dict(capacity=1 << 19, isect_capacity=8 << 20))
        cmodel.getGaussianParms().add([t.clone() for t in model.opt_gs_params.tensors()])
        cpipe = H_.SLAMPipeline(ceng, cmodel, scene_seed(rank), args.gt_pose)
        # tracking / mapping overlap (host/slam_pipeline.hpp): the keyframe's map update runs on a second stream while the next
        # frames are tracked and fused; same results as the sequential schedule.  flush() below closes the timed region, so every
        # frame's work (incl. the deferred prune) is inside it.
        cpipe.overlap_mapping = not args.no_overlap
        cpipe.mapping_thread = not args.no_map_thread
        ccams = []
        for k in range(n_frames):
            c = H_.Camera(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][k].astype(np.float32)))
            c.id, c.image, c.depth = k, cams[k].image, cams[k].depth
            ccams.append(c)

        frame_times = [] if os.environ.get("GPS_BENCH_FRAME_TIMES") else None

        def run(lo, hi):
            for i in range(lo, hi):
                if frame_times is not None:
                    t = time.perf_counter()
                cpipe.processFrame(i, ccams[i], rgb_dev[i], depth_dev[i])
                if frame_times is not None:
                    frame_times.append((i, time.perf_counter() - t))
            if frame_times is not None:
                t = time.perf_counter()
            cpipe.flush()
            if frame_times is not None:
                frame_times.append((-1, time.perf_counter() - t))
    else:
        def run(lo, hi):
            for i in range(lo, hi):
                pipe.process_frame(i, cams[i], rgb_dev[i], depth_dev[i])

    # One-off costs that are not part of any SLAM frame -- loading the code objects of every kernel, allocating the
    # capacity-sized intermediates / Adam state -- are paid here, before the warm-up frames, so that the timed region is
    # the same steady-state loop for any --warmup (a throwaway 64x48 sequence drives the full loop once).
    prime(args.host, device)
    if args.host == "cpp":
        cmodel.reserveWorkspace(W, H)
    else:
        model._step_struct(W, H)
        model.initOptimizers(-1, 1.0)
        model._opt["step"] = 0
    torch.cuda.synchronize()
    run(0, Wm)  # untimed warm-up frames
    torch.cuda.synchronize()
    grp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(Wm, Wm + K)
    torch.cuda.synchronize()
    grp.barrier()
    torch.cuda.synchronize()
    dt = grp.max_over_ranks(time.perf_counter() - t0)
    if args.host == "cpp" and frame_times is not None and rank == 0:  # host-side duration of every processFrame call (debug aid)
        timed = [(i, d) for i, d in frame_times if i >= Wm or i == -1][-(K + 1):]
        key = [d for i, d in timed if i > 0 and i % 10 == 0]
        rest = [d for i, d in timed if i > 0 and i % 10 != 0]
        sys.stderr.write("processFrame host time: keyframes mean %.3f ms, other frames mean %.3f ms (min %.3f max %.3f), final flush %.3f ms\n"
                         % (1e3 * sum(key) / max(1, len(key)), 1e3 * sum(rest) / max(1, len(rest)), 1e3 * min(rest), 1e3 * max(rest),
                            1e3 * timed[-1][1]))

    out = None
    if rank == 0:
        from bench_kernels import dominant_kernel_roofline, cpu_baseline
        if args.host == "cpp":
            # hand the C++ model's state to the Python mirror for the per-kernel measurement below (same C-ABI, same buffers
            # layout); the timed region above never touched the Python model
            N = cmodel.getGaussianNum()
            cp = cmodel.getGaussianParms()
            model.opt_gs_params.N = 0
            model.add_params(dict(means=cp.getMeans(), scales=cp.getScales(), quats=cp.getQuats(),
                                  featuresDc=cp.getFeaturesDc(), featuresRest=cp.getFeaturesRest(),
                                  opacities=cp.getOpacities()))
            oc, orc = cpipe.optCams(), cpipe.optRaycasts()
            pcam = cams[oc[-1].id]
            pcam.c2w_slam = oc[-1].c2w_slam.cpu()
            pcam.invalidate()
            pipe.opt_cam_list, pipe.opt_raycast_list = [pcam], [orc[-1]]
            stats = dict(cpipe.stats())
        else:
            N = model.getGaussianNum()
            stats = pipe.stats
        # render quality of the state the timed run ended in (after the timed region): PSNR of the composed render and of
        # the TSDF raycast colour alone against the input images of the last optimisation cameras
        # (formula: scripts/utils/image_utils.py:19-21).  Synthetic scene -> an absolute figure, not the Replica number.
        def _psnr(a, b):
            return float(-10.0 * torch.log10(((a.clamp(0, 1) - b) ** 2).mean()))
        with torch.no_grad():
            if args.host == "cpp":
                views = list(zip(cpipe.optCams(), cpipe.optRaycasts()))[-5:]
                fwd = lambda c, rc: cmodel.forward(c, rc["depth_map"], rc["color_map"])["rgb"]
            else:
                views = list(zip(pipe.opt_cam_list, pipe.opt_raycast_list))[-5:]
                fwd = lambda c, rc: model.forward(c, rc["depth_map"], rc["color_map"])["rgb"]
            psnr_render = [_psnr(fwd(c, rc), c.image) for c, rc in views]
            psnr_tsdf = [_psnr(rc["color_map"], c.image) for c, rc in views]
        quality = {"views": len(views), "render_psnr_db_vs_input": sum(psnr_render) / max(1, len(views)),
                   "tsdf_colour_psnr_db_vs_input": sum(psnr_tsdf) / max(1, len(views))}
        # Fusion-FPS / Gaussian-FPS split as the reference reports it (run/read_results.py:38-39): the TSDF-only `recon`
        # loop over the same timed frames on a fresh engine gives the fusion share, the rest is the Gaussian share
        if args.host == "cpp":
            e2 = H_.ITMBasicEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, 0.2, 10.0)
            p2 = H_.SLAMPipeline(e2, H_.SLAMGaussianModel(), 1, args.gt_pose)
            p2.work_mode = "recon"
            step2 = lambda i: p2.processFrame(i, ccams[i], rgb_dev[i], depth_dev[i])
        else:
            from gps_slam_amd.gs_model import SLAMGaussianModel as _M
            from gps_slam_amd.slam_pipeline import SLAMPipeline as _P
            from gps_slam_amd.tsdf_engine import TsdfEngine as _E
            p2 = _P(_E(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel_size=0.005, mu=0.02, device=device),
                    _M(device=device), work_mode="recon", use_gt_pose=args.gt_pose)
            step2 = lambda i: p2.process_frame(i, cams[i], rgb_dev[i], depth_dev[i])
        for i in range(Wm):
            step2(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(Wm, Wm + K):
            step2(i)
        torch.cuda.synchronize()
        fusion_ms = 1000.0 * (time.perf_counter() - t1) / K
        total_ms = 1000.0 * dt / K
        split = {"fusion_ms_per_frame": fusion_ms, "gaussian_ms_per_frame": max(0.0, total_ms - fusion_ms),
                 "fusion_fps": 1000.0 / fusion_ms, "gaussian_fps": 1000.0 / max(1e-9, total_ms - fusion_ms)}
        roof = dominant_kernel_roofline(model, pipe, eng, cams, device, HBM_PEAK_GBS)
        out = {
            "metric": "SLAM frames/sec @640x480, ~200k Gaussians; render PSNR vs ref",
            "value": world * K / dt, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": 1000.0 * dt / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic room0-like RGB-D %dx%d: %s + TSDF fuse (5mm voxels) + ges splat optimise, ~%dk "
                                   "Gaussians; independent scene per GPU"
                                   % (W, H, "given poses (use_gt_pose=true, as every shipped config)" if args.gt_pose else
                                      "depth ICP tracking (ExtendedTracker, use_gt_pose=false)", N // 1000),
                       "gaussians": N, "local_opt_interval": 10, "local_opt_iters": 20,
                       "frames_per_step": 1, "stats": stats, "host": args.host, "overlap_mapping": bool(args.host == "cpp" and not args.no_overlap), "mapping_thread": bool(args.host == "cpp" and not args.no_overlap and not args.no_map_thread), "use_gt_pose": bool(args.gt_pose), "quality": quality, "split": split},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(seq, W, H)
        print(json.dumps(out), flush=True)
    grp.close()


if __name__ == "__main__":
    main()

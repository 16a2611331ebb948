"""Measurement helpers for bench.py: live roofline of the dominant kernel (HIP events on the launch stream) and
the CPU baseline (the oracle restatement of the reference's ITMLib CPU path, timed on this box's host cores)."""
import ctypes as C
import os
import time

import numpy as np
import torch


def _time_launches(fn, n, stream):
    """average duration (s) of n back-to-back launches of fn() measured with events on `stream`"""
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record(stream)
    for _ in range(n):
        fn()
    end.record(stream)
    torch.cuda.synchronize()
    return start.elapsed_time(end) * 1e-3 / n


def dominant_kernel_roofline(model, pipe, eng, cams, device, hbm_peak_gbs):
    """Dominant kernel of the step = the Gaussian-parallel ges backward (raster_ges_bwd_gs_kernel): it has the
    largest share of GPU time per SLAM frame (2 launches/frame; see profiles/).  Measured live with events on the
    stream the C-ABI launches on (torch's current stream), on the last optimisation camera of the run.
    Also reports the forward rasterizer, the live raycast and the fused Adam for context (`others`)."""
    import json
    from gps_slam_amd import gsplat_ops as ops
    stream = torch.cuda.current_stream()
    cam = pipe.opt_cam_list[-1] if pipe.opt_cam_list else cams[-1]
    rc = pipe.opt_raycast_list[-1] if pipe.opt_raycast_list else pipe.runRaycastByCam(cam)
    st = model._render(cam, rc["depth_map"], rc["color_map"])
    ni, ng = model._isect.sizes()
    W, H = st["W"], st["H"]
    P = W * H
    N = model.getGaussianNum()
    rgb, _, loss, v_rc, v_ra = ops.compose_l1(st["render_colors"], st["weight_sum"], rc["color_map"], rc["depth_map"],
                                             cam.image, need_depth=False)

    def fwd():
        ops.rasterize_to_pixels_fwd_ges(st["means2d"], st["conics"], st["colors"], st["opac"], st["ref_clamped"], W, H,
                                        model.tile_size, model._isect, model.delta_depth)

    gbuf = (torch.zeros_like(st["means2d"]), torch.zeros_like(st["conics"]), torch.zeros_like(st["colors"]),
            torch.zeros_like(st["opac"]))

    def bwd():  # accumulate=True: exactly one launch of raster_ges_bwd_gs_kernel, no zero-fill kernel
        ops.rasterize_to_pixels_bwd_ges_gs_parallel(st["means2d"], st["conics"], st["colors"], st["opac"], st["radii"],
                                                    st["ref_clamped"], W, H, model._isect, model.delta_depth, v_rc, v_ra,
                                                    out=gbuf, accumulate=True)

    t_bwd = _time_launches(bwd, 50, stream)
    t_fwd = _time_launches(fwd, 50, stream)
    # algorithmic bytes, SURVEY 8(d) raster-bwd row: 52 B Gaussian record per 32-px group + gradient image once
    # (24 B/px) + 40 B of accumulations per group
    alg_bwd = 52.0 * ng + 24.0 * P + 40.0 * ng
    alg_fwd = 44.0 * ni + 4.0 * P + 20.0 * P
    ach = alg_bwd / t_bwd / 1e9
    traffic = None
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_raster_bwd.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            traffic = None
    return {"bound": "hbm", "kernel": "raster_ges_bwd_gs_kernel", "achieved": ach, "peak": hbm_peak_gbs, "unit": "GB/s",
            "frac": ach / hbm_peak_gbs, "traffic": traffic, "avg_launch_us": t_bwd * 1e6,
            "algorithmic_bytes": alg_bwd, "units": {"n_groups": ng, "pixels": P, "n_isects": ni, "gaussians": N,
                                                    "n_visible": int(model._isect.counts[3])},
            "note": "rasterization is ALU/LDS-issue bound (exp + ~40 flop per pixel-Gaussian pair), not a stream; "
                    "the HBM fraction is reported because it is the contract's yardstick",
            "others": {"raster_ges_fwd_kernel": {"avg_launch_us": t_fwd * 1e6, "algorithmic_bytes": alg_fwd,
                                                 "achieved_GBs": alg_fwd / t_fwd / 1e9}}}


def cpu_baseline(seq, W, H, max_seconds=20.0):
    """ProcessFrame of the reference's ITMLib CPU path (config[0]: TSDF-only `recon` loop), timed through the
    bit-exact C restatement (oracle/tsdf_oracle.c; the reference sources do not exist on the GPU box).
    Scalar single-thread port -> cores = 1.  Bounded sample: as many 640x480 frames as fit in ~max_seconds."""
    from oracle import tsdf_ref as R
    o = R.TsdfOracle(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, 0.2, 10.0)
    n = 0
    t_used = 0.0
    frames = seq["rgb"].shape[0]
    while t_used < max_seconds and n < frames:
        M, invM = R.pose_from_c2w(seq["c2w"][n])
        t0 = time.perf_counter()
        o.process_frame(seq["rgb"][n], seq["depth"][n], M, invM)
        dt = time.perf_counter() - t0
        if n > 0:  # first frame (bulk allocation) excluded, as BASELINE.md prescribes
            t_used += dt
        n += 1
    o.close()
    timed = max(1, n - 1)
    return {"value": timed / t_used if t_used > 0 else 0.0, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d ProcessFrame calls (TSDF fuse + live raycast + ICP maps, no Gaussians) of the same %dx%d "
                      "synthetic sequence, first frame excluded; host CPU: %s" % (timed, W, H, _cpu_name())}


def _cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip() + " (%d logical cores visible)" % os.cpu_count()
    except OSError:
        pass
    return "unknown"

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
    stream the C-ABI launches on (torch's current stream), on the last optimisation camera of the run, through the
    same C-ABI entry point with accumulate=1 (exactly one kernel per call).  `others` lists the forward rasterizer."""
    import json
    from gps_slam_amd._lib import lib
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    cam = pipe.opt_cam_list[-1] if pipe.opt_cam_list else cams[-1]
    rc = pipe.opt_raycast_list[-1] if pipe.opt_raycast_list else pipe.runRaycastByCam(cam)
    if model._opt is None:
        model.initOptimizers(-1, 1.0)
    model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
    torch.cuda.synchronize()
    B, st = model._B, model._step
    counts = B["counts"].cpu().tolist()
    ni, ng, nvis = int(counts[0]), int(counts[1]), int(counts[3])
    W, H, N = st.width, st.height, st.N
    P = W * H
    ref = rc["depth_map_clamped"]
    ptr = lambda t: C.c_void_p(t.data_ptr())
    pp = model.opt_gs_params

    def fwd():  # the forward the fused step launches (packed-math kernel over the preprocess records)
        lib.gps_raster_ges_fwd_rec(N, ptr(B["records"]), ptr(ref), W, H, ptr(B["tile_offsets"]), ptr(B["flatten_ids"]),
                                   ptr(B["counts"]), model.delta_depth, ptr(B["render_colors"]), ptr(B["weight_sum"]), sp)

    def bwd():
        lib.gps_raster_ges_bwd_gs(N, ptr(B["means2d"]), ptr(B["conics"]), ptr(B["colors"]), ptr(B["opacities"]),
                                  ptr(B["radii"]), ptr(ref), W, H, ptr(B["group_gs_ids"]), ptr(B["group_starts"]),
                                  ptr(B["counts"]), model.delta_depth, ptr(B["v_render_colors"]),
                                  ptr(B["v_render_alphas"]), ptr(B["v_means2d"]), ptr(B["v_conics"]), ptr(B["v_colors"]),
                                  ptr(B["v_opacities"]), 1, sp)

    t_bwd = _time_launches(bwd, 50, stream)
    t_fwd = _time_launches(fwd, 50, stream)
    # algorithmic bytes, SURVEY 8(d) raster-bwd row: 52 B Gaussian record per 32-px group + gradient image once
    # (24 B/px) + 40 B of accumulations per group
    alg_bwd = 52.0 * ng + 24.0 * P + 40.0 * ng
    alg_fwd = 44.0 * ni + 4.0 * P + 20.0 * P
    ach = alg_bwd / t_bwd / 1e9
    traffic, valu = None, None
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_raster_bwd.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            traffic = rec.get("hbm_bytes_per_launch")
            insts = (rec.get("wave_instructions_per_launch") or {}).get("SQ_INSTS_VALU")
            if insts:
                # issue-side yardstick for a VALU-bound kernel: a wave64 VALU instruction occupies its SIMD for 4 cycles;
                # 256 CUs x 4 SIMDs at 2.4 GHz (MI355X_MICROARCH.md) -> fraction of the chip's VALU issue slots this launch used
                valu = {"wave_instructions": insts, "issue_frac": insts * 4.0 / (1024 * 2.4e9 * t_bwd),
                        "note": "SQ_INSTS_VALU (own --pmc pass, profiles/pmc_raster_bwd.json) x 4 cycles / (1024 SIMDs x 2.4 GHz x launch time)"}
        except (OSError, ValueError):
            traffic = None
    return {"bound": "hbm", "kernel": "raster_ges_bwd_gs_kernel", "achieved": ach, "peak": hbm_peak_gbs, "unit": "GB/s",
            "frac": ach / hbm_peak_gbs, "traffic": traffic, "valu": valu, "avg_launch_us": t_bwd * 1e6,
            "algorithmic_bytes": alg_bwd, "units": {"n_groups": ng, "pixels": P, "n_isects": ni, "gaussians": N,
                                                    "n_visible": nvis},
            "note": "rasterization is ALU/LDS-issue bound (exp + ~40 flop per pixel-Gaussian pair), not a stream; "
                    "the HBM fraction is reported because it is the contract's yardstick",
            "others": {"raster_ges_fwd_pk_kernel": {"avg_launch_us": t_fwd * 1e6, "algorithmic_bytes": alg_fwd,
                                                 "achieved_GBs": alg_fwd / t_fwd / 1e9}}}


def cpu_baseline(seq, W, H, max_seconds=20.0):
    """CPU baseline of the TSDF-only `recon` loop (BASELINE config[0]).  Preferred: the REFERENCE's own ITMLib CPU engine,
    oracle/_ref/itm_ref_omp (compiled from the reference sources like upstream, -O3 + OpenMP; the binary travels with the
    snapshot), on a bounded sample of the same sequence with one OpenMP thread per physical core -> kind "reference".
    Fallback when the binary is absent: the bit-exact single-thread C restatement -> kind "port"."""
    from oracle import tsdf_ref as R
    if os.path.exists(R.BIN_OMP):
        cores = _physical_cores()
        probe = R.time_reference(seq, 4, 0.005, 0.02, 0.2, 10.0, threads=cores)
        per = max(1e-3, probe["seconds"] / probe["frames"])
        n = int(max(6, min(seq["rgb"].shape[0], 1 + max_seconds / per)))
        res = R.time_reference(seq, n, 0.005, 0.02, 0.2, 10.0, threads=cores)
        one = R.time_reference(seq, min(n, 8), 0.005, 0.02, 0.2, 10.0, threads=1)
        return {"value": res["frames"] / res["seconds"], "unit": "frames/s", "cores": cores, "kind": "reference",
                "sample": "%d ProcessFrame calls (TSDF fuse + live raycast + ICP maps, tracking off, no Gaussians) of the same "
                          "%dx%d synthetic sequence by the reference's ITMLib CPU engine (oracle/_ref/itm_ref_omp: g++ -O3 "
                          "-fopenmp as upstream), OMP_NUM_THREADS=%d, first frame excluded; single thread: %.2f frames/s; "
                          "host CPU: %s" % (res["frames"], W, H, cores, one["frames"] / one["seconds"], _cpu_name())}
    return cpu_baseline_port(seq, W, H, max_seconds)


def _physical_cores():
    try:
        pairs = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip() and phys is not None and core is not None:
                pairs.add((phys, core)); phys = core = None
        if pairs:
            return min(len(pairs), len(os.sched_getaffinity(0)))
    except OSError:
        pass
    return max(1, len(os.sched_getaffinity(0)) // 2)


def cpu_baseline_port(seq, W, H, max_seconds=20.0):
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

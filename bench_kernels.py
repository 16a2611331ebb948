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


def iteration_bytes(N, Nv, I, G, P, T):
    """Algorithmic (compulsory) HBM bytes of one optimise iteration, SURVEY.md 8(d), term by term."""
    terms = {
        "proj_fwd": 68 * N, "sh_fwd": 217 * Nv, "binning": 24 * N + 44 * I + 8 * G + 4 * T, "raster_fwd": 44 * I + 28 * P,
        "compose_l1": 40 * P, "raster_bwd": 52 * G + 24 * P + 40 * G, "sh_bwd": 408 * Nv, "proj_bwd": 116 * Nv + 40 * N,
        "adam": 28 * 59 * N,
    }
    return float(sum(terms.values())), {k: float(v) for k, v in terms.items()}


def fusion_bytes(P, V, S):
    """Algorithmic HBM bytes of one TSDF frame, SURVEY.md 8(d), WITHOUT the ray term (hash / voxel gathers of neighbouring
    rays are largely L2 resident; S-bar, the mean steps per ray, is not logged by the kernel): upload + convert + allocate (depth
    + 2 probes x 16 B) + two table sweeps + integrate R/W + visible list + min/max image + raycast outputs."""
    return float(6 * P + 6 * P + 4 * P + 32 * P + 2 * S * 17 + V * 8192 + V * 16 + 8 * P / 64.0 + 20 * P)


def _python_twin(scene, device):
    """The C++ model's state in the Python mirror (same C-ABI, same buffer layout) for the per-kernel measurements below;
    the timed region never touched it."""
    from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
    cp = scene.model.getGaussianParms()
    model = SLAMGaussianModel(dict(capacity=1 << 19, isect_capacity=8 << 20), device=device)
    model.add_params(dict(means=cp.getMeans().clone(), scales=cp.getScales().clone(), quats=cp.getQuats().clone(),
                          featuresDc=cp.getFeaturesDc().clone(), featuresRest=cp.getFeaturesRest().clone(),
                          opacities=cp.getOpacities().clone()))
    oc, orc = scene.pipe.optCams(), scene.pipe.optRaycasts()
    c, rc = oc[-1], orc[-1]
    cam = Camera(c.id, c.width, c.height, c.fx, c.fy, c.cx, c.cy, c.c2w.cpu().numpy(), image=c.image, device=device)
    cam.c2w_slam = c.c2w_slam.cpu()
    cam.invalidate()
    return model, cam, rc


def iteration_roofline(scene, seq, result, hbm_peak_gbs, K):
    """`roofline` of the bench line, all of it measured on the state the timed run ended in (no stored constants are divided
    by live times):

    * dominant kernel = the Gaussian-parallel ges backward (largest share of GPU time per SLAM frame): `achieved` =
      algorithmic bytes (SURVEY 8(d) raster-bwd row x the G, P of this launch) / average launch duration, HIP events on the
      launch stream, accumulate = 1 so exactly one kernel per call;
    * `iteration`: B_iter of SURVEY 8(d) evaluated with the logged N, Nv, I, G / the measured duration of one whole optimise
      iteration (gps_splat_train_step, 20 back-to-back) -> fraction of HBM peak;
    * `frame`: (2 B_iter + B_fuse) / the measured ms_per_step (20 iterations per 10 frames; B_fuse without the ray term);
    * `traffic` (HBM bytes per launch from --pmc passes) only if profiles/pmc_raster_bwd.json was collected on a scene whose
      (N, G) match this one within 1 %; otherwise null.  tools/profile.sh regenerates the file.
    """
    import json
    from gps_slam_amd._lib import lib
    device = "cuda:%d" % torch.cuda.current_device()
    model, cam, rc = _python_twin(scene, device)
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    model.initOptimizers(-1, 1.0)
    model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
    torch.cuda.synchronize()
    B, st = model._B, model._step
    counts = B["counts"].cpu().tolist()
    ni, ng, nvis = int(counts[0]), int(counts[1]), int(counts[3])
    W, H, N = st.width, st.height, st.N
    P, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    ref = rc["depth_map_clamped"]
    ptr = lambda t: C.c_void_p(t.data_ptr())

    def fwd():  # the forward the fused step launches (packed-math kernel over the preprocess records)
        lib.gps_raster_ges_fwd_rec(N, ptr(B["records"]), ptr(ref), W, H, ptr(B["tile_offsets"]), ptr(B["flatten_ids"]),
                                   ptr(B["counts"]), model.delta_depth, ptr(B["render_colors"]), ptr(B["weight_sum"]), sp)

    def bwd():
        lib.gps_raster_ges_bwd_gs(N, ptr(B["means2d"]), ptr(B["conics"]), ptr(B["colors"]), ptr(B["opacities"]),
                                  ptr(B["radii"]), ptr(ref), W, H, ptr(B["group_gs_ids"]), ptr(B["group_starts"]),
                                  ptr(B["counts"]), model.delta_depth, ptr(B["v_render_colors"]),
                                  ptr(B["v_render_alphas"]), ptr(B["v_means2d"]), ptr(B["v_conics"]), ptr(B["v_colors"]),
                                  ptr(B["v_opacities"]), 1, sp)

    def step():
        model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])

    t_bwd = _time_launches(bwd, 50, stream)
    t_fwd = _time_launches(fwd, 50, stream)
    t_iter = _time_launches(step, 20, stream)
    alg_bwd = 52.0 * ng + 24.0 * P + 40.0 * ng
    alg_fwd = 44.0 * ni + 4.0 * P + 20.0 * P
    ach = alg_bwd / t_bwd / 1e9
    b_iter, terms = iteration_bytes(N, nvis, ni, ng, P, T)
    V = int(scene.engine.counters().cpu()[2])  # GPS_TSDF_N_VISIBLE of the last fused frame
    S = 0x100000 + 0x20000
    b_fuse = fusion_bytes(P, V, S)
    t_frame = result["ms_per_step"] * 1e-3
    b_frame = 2.0 * b_iter + b_fuse
    traffic, traffic_note, valu = None, "no PMC file for this scene", None
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_raster_bwd.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            u = rec.get("units") or {}
            if u and abs(u.get("n_groups", 0) - ng) <= 0.01 * ng and abs(u.get("gaussians", 0) - N) <= 0.01 * N:
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_note = "profiles/pmc_raster_bwd.json (same scene: N and G within 1 %): FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, separate --pmc passes"
                valu = rec.get("valu")
                if valu and valu.get("wave_instructions_per_launch", {}).get("SQ_INSTS_VALU"):
                    # counter collection serialises and slows the launches (launch_us_same_run); the instruction COUNT is
                    # what carries over -- priced here against this run's live launch time
                    valu = dict(valu, issue_frac_live=valu["wave_instructions_per_launch"]["SQ_INSTS_VALU"] * 4.0 /
                                (1024 * 2.4e9 * t_bwd), launch_us_live=t_bwd * 1e6)
            else:
                traffic_note = "profiles/pmc_raster_bwd.json was collected on a different scene (its N, G: %s, %s) -- not reported" % (
                    u.get("gaussians"), u.get("n_groups"))
        except (OSError, ValueError):
            pass
    return {"bound": "hbm", "kernel": "raster_ges_bwd_gs_kernel", "achieved": ach, "peak": hbm_peak_gbs, "unit": "GB/s",
            "frac": ach / hbm_peak_gbs, "traffic": traffic, "traffic_note": traffic_note, "valu": valu,
            "avg_launch_us": t_bwd * 1e6, "algorithmic_bytes": alg_bwd,
            "units": {"n_groups": ng, "pixels": P, "n_isects": ni, "gaussians": N, "n_visible": nvis, "tiles": T,
                      "visible_blocks": V},
            "iteration": {"algorithmic_bytes": b_iter, "terms": terms, "avg_us": t_iter * 1e6,
                          "achieved_GBs": b_iter / t_iter / 1e9, "frac": b_iter / t_iter / 1e9 / hbm_peak_gbs},
            "frame": {"algorithmic_bytes": b_frame, "fusion_bytes_without_ray_term": b_fuse, "ms": t_frame * 1e3,
                      "achieved_GBs": b_frame / t_frame / 1e9, "frac": b_frame / t_frame / 1e9 / hbm_peak_gbs},
            "note": "rasterization is ALU/LDS-issue bound (exp + ~40 flop per pixel-Gaussian pair), not a stream; the HBM "
                    "fraction is reported because it is the contract's yardstick",
            "others": {"raster_ges_fwd_pk_kernel": {"avg_launch_us": t_fwd * 1e6, "algorithmic_bytes": alg_fwd,
                                                 "achieved_GBs": alg_fwd / t_fwd / 1e9}}}


def roofline_section(scene, seq, result, hbm_peak_gbs, K, gt_pose=False):
    return iteration_roofline(scene, seq, result, hbm_peak_gbs, K)


def render_psnr_vs_oracle(model, cam, rc, seq):
    """PSNR (scripts/utils/image_utils.py:19-21) of the HIP render of the final state against the CPU
    restatement's render of the same state (same parameters, pose, raycast maps), both clamped to [0,1], full image."""
    from oracle import splat_ref as orc
    cp = model.getGaussianParms()
    n = lambda t: t.detach().cpu().numpy()
    K = np.array([[seq["fx"], 0, seq["cx"]], [0, seq["fy"], seq["cy"]], [0, 0, 1]], np.float32)
    W, H = seq["W"], seq["H"]
    t0 = time.perf_counter()
    e_rgb, _ = orc.ges_render(n(cp.getMeans()), n(cp.getScales()), n(cp.getQuats()), n(cp.getFeaturesDc()), n(cp.getFeaturesRest()),
                              n(cp.getOpacities()), n(cam.c2w_slam), K, W, H, n(rc["depth_map"])[..., 0], n(rc["color_map"]),
                              delta_depth=0.1)
    oracle_s = time.perf_counter() - t0
    with torch.no_grad():
        got = model.forward(cam, rc["depth_map"], rc["color_map"])["rgb"].clamp(0, 1).double()
    exp = torch.as_tensor(e_rgb).to(got.device).clamp(0, 1).double()
    mse = float(((got - exp) ** 2).mean())
    return {"render_psnr_db_vs_oracle": (-10.0 * float(np.log10(mse))) if mse > 0 else float("inf"),
            "render_max_abs_diff_vs_oracle": float((got - exp).abs().max()), "oracle_render_seconds": oracle_s}


def fusion_split(seq, first, K, gt_pose, dt_total):
    """Fusion-FPS / Gaussian-FPS split as the reference reports it (run/read_results.py:38-39): the TSDF-only `recon` loop over
    the same timed frames (upload + tracking + fuse + raycast, no Gaussians) on a fresh engine gives the fusion share, the
    rest of the step time is the Gaussian share."""
    import gps_slam_amd._host as H_
    W, H = seq["W"], seq["H"]
    reader = H_.DatasetReader(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"])
    cams = []
    for k in range(first + K):
        c = H_.Camera(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][k]))
        c.id = k
        c.image = torch.as_tensor(seq["rgb"][k].astype(np.float32) / 255.0)
        c.depth = torch.as_tensor(seq["depth"][k].astype(np.float32) / 1000.0)[..., None]
        reader.addTrainCamera(c)
        pc = H_.Camera(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], True, torch.as_tensor(seq["c2w"][k]))
        pc.id = k
        cams.append(pc)
    cli = H_.createTsdfEngine(reader, dict(voxel_size=0.005, trunc_dist=0.02, viewFrustum_min=0.2, viewFrustum_max=10.0,
                                           use_gt_pose=1 if gt_pose else 0))
    model = H_.SLAMGaussianModel()
    pipe = H_.SLAMPipeline(1)
    pipe.setTsdfEngine(cli)
    pipe.setModel(model)
    pipe.work_mode = "recon"
    for i in range(first):
        pipe.processFrameCLI(i, cams[i])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(first, first + K):
        pipe.processFrameCLI(i, cams[i])
    torch.cuda.synchronize()
    fusion_ms = 1000.0 * (time.perf_counter() - t1) / K
    total_ms = 1000.0 * dt_total / K
    return {"fusion_ms_per_frame": fusion_ms, "gaussian_ms_per_frame": max(0.0, total_ms - fusion_ms),
            "fusion_fps": 1000.0 / fusion_ms, "gaussian_fps": 1000.0 / max(1e-9, total_ms - fusion_ms)}


def cpu_baseline(seq, W, H, max_seconds=15.0):
    """CPU baseline of the TSDF-only `recon` loop (BASELINE config[0]).  Preferred: the REFERENCE's own ITMLib CPU engine,
    oracle/_ref/itm_ref_omp (compiled from the reference sources like upstream, -O3 + OpenMP; the binary travels with the
    snapshot), on a bounded sample of the same sequence with one OpenMP thread per physical core -> kind "reference".
    Fallback when the binary is absent: the bit-exact single-thread C restatement -> kind "port"."""
    from oracle import tsdf_ref as R
    if os.path.exists(R.BIN_OMP):
        cores = _physical_cores()
        # the reference at its best: a short sweep over OpenMP thread counts (most of ITMLib's CPU path is serial sweeps over
        # the hash table, so one thread per core mostly oversubscribes), then the bounded sample with the fastest count
        allowed = len(os.sched_getaffinity(0))
        sweep = {}
        for t in sorted({t for t in (1, 8, 16, 32, 64, cores) if t <= max(1, allowed)}):
            r = R.time_reference(seq, 6, 0.005, 0.02, 0.2, 10.0, threads=t)
            sweep[t] = r["frames"] / r["seconds"]
        best = max(sweep, key=sweep.get)
        per = 1.0 / max(1e-3, sweep[best])
        n = int(max(6, min(seq["rgb"].shape[0], 1 + max_seconds / per)))
        res = R.time_reference(seq, n, 0.005, 0.02, 0.2, 10.0, threads=best)
        return {"value": res["frames"] / res["seconds"], "unit": "frames/s", "cores": best, "kind": "reference",
                "thread_sweep_frames_per_s": {str(t): round(v, 3) for t, v in sweep.items()},
                "sample": "%d ProcessFrame calls (TSDF fuse + live raycast + ICP maps, tracking off, no Gaussians) of the same "
                          "%dx%d synthetic sequence by the reference's ITMLib CPU engine (oracle/_ref/itm_ref_omp: g++ -O3 "
                          "-fopenmp as upstream), OMP_NUM_THREADS=%d = the fastest of a sweep over %s threads (6 frames each; %d "
                          "physical cores), first frame excluded; host CPU: %s"
                          % (res["frames"], W, H, best, "/".join(str(t) for t in sweep), cores, _cpu_name())}
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

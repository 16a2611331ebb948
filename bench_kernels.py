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


TIMED_KINDS = ("preprocess_bwd_kernel", "preprocess_fwd_kernel", "raster_ges_fwd_pk_kernel", "raster_ges_bwd_strip_kernel",
               "sb_scan_kernel", "sb_scatter_kernel", "integrate_kernel", "raycast_kernel")   # == GPS_TIMED_* (include/gps_slam_hip.h)


def in_loop_kernel_times(run, frames):
    """Per-kernel launch durations INSIDE the running schedule: gps_launch_timing_start, `run()` (K SLAM frames of the scene the
    timed windows just measured, same schedule, same streams and host threads), gps_launch_timing_stop.  Every launch of the
    instrumented kernels stamps the device's 100 MHz wall clock at the start and end of each of its waves into its workgroup's
    {first start, last end} slot (csrc/launch_timing.hpp: the kernel's own execution interval -- a rocprofv3 kernel trace reports the
    same launches 1-3 us longer, the dispatch's ramp-up and the end-of-kernel release); this is what the `roofline` of the
    line is priced with -- the kernels-alone micro-loops of roofline_section run with nothing beside them and warm
    inputs.  The window is an extra one after the timed windows."""
    from gps_slam_amd._lib import lib
    assert lib.gps_launch_timing_start(1 << 21) == 0   # workgroup slots (32 MB): ~0.4 M workgroups in 20 frames
    t0 = time.perf_counter()
    try:
        run()
        torch.cuda.synchronize()
    finally:
        rc = lib.gps_launch_timing_stop()
    wall = time.perf_counter() - t0
    assert rc == 0, "gps_launch_timing_stop: %d" % rc
    out = {"frames": frames, "window_ms_per_step": 1e3 * wall / frames, "kernels": {}}
    for kind, name in enumerate(TIMED_KINDS):
        tot, totf, mx = C.c_double(), C.c_double(), C.c_double()
        n, nf, dropped = C.c_int64(), C.c_int64(), C.c_int64()
        assert lib.gps_launch_timing_read(kind, C.byref(tot), C.byref(n), C.byref(totf), C.byref(nf), C.byref(mx), C.byref(dropped)) == 0
        assert dropped.value == 0, "launch timing ring overflowed"
        if n.value:
            out["kernels"][name] = {"launches": n.value, "avg_us": tot.value / n.value, "us_per_frame": tot.value / frames, "max_us": mx.value,
                                    "launches_flagged": nf.value, "avg_us_flagged": totf.value / nf.value if nf.value else None,
                                    "avg_us_unflagged": (tot.value - totf.value) / (n.value - nf.value) if n.value > nf.value else None}
    return out


def iteration_bytes(N, Nv, I, G, P, T):
    """Algorithmic (compulsory) HBM bytes of one optimise iteration, SURVEY.md 8(d), term by term."""
    terms = {
        "proj_fwd": 68 * N, "sh_fwd": 217 * Nv, "binning": 24 * N + 44 * I + 8 * G + 4 * T, "raster_fwd": 44 * I + 28 * P,
        "compose_l1": 40 * P, "raster_bwd": 52 * G + 24 * P + 40 * G, "sh_bwd": 408 * Nv, "proj_bwd": 116 * Nv + 40 * N,
        "adam": 28 * 59 * N,
    }
    return float(sum(terms.values())), {k: float(v) for k, v in terms.items()}


def fusion_bytes(P, V, S, s_bar=0.0):
    """Algorithmic HBM bytes of one TSDF frame, SURVEY.md 8(d): upload + convert + allocate (depth + 2 probes x 16 B) + two table
    sweeps + integrate R/W + visible list + min/max image + raycast outputs, plus the ray term P * S-bar * 24 (one 16-byte hash
    entry + one 8-byte voxel per castRay step; S-bar = mean steps per ray as the kernel logs them, GPS_TSDF_RAY_STEPS).  The ray
    term is an upper bound -- neighbouring rays share entries and voxels in the caches -- so callers report both figures."""
    return float(6 * P + 6 * P + 4 * P + 32 * P + 2 * S * 17 + V * 8192 + V * 16 + 8 * P / 64.0 + 20 * P + 24.0 * P * s_bar)


def _python_twin(scene, device, strip_backward=True):
    """The C++ model's state in the Python mirror (same C-ABI, same buffer layout) for the per-kernel measurements below;
    the timed region never touched it.  strip_backward=False: the round-2 route (sorted keys, group tables, group backward) --
    what the probes that time the operator-level group kernel need."""
    from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
    cp = scene.model.getGaussianParms()
    # (fuse_sh_rest_adam = 2: all six tensors stepped inside the backward kernel, as the C++ host runs the step)
    model = SLAMGaussianModel(dict(capacity=1 << 19, isect_capacity=8 << 20, strip_backward=strip_backward, fuse_sh_rest_adam=2), device=device)
    model.add_params(dict(means=cp.getMeans().clone(), scales=cp.getScales().clone(), quats=cp.getQuats().clone(),
                          featuresDc=cp.getFeaturesDc().clone(), featuresRest=cp.getFeaturesRest().clone(),
                          opacities=cp.getOpacities().clone()))
    oc, orc = scene.pipe.optCams(), scene.pipe.optRaycasts()
    c, rc = oc[-1], orc[-1]
    cam = Camera(c.id, c.width, c.height, c.fx, c.fy, c.cx, c.cy, c.c2w.cpu().numpy(), image=c.image, device=device)
    cam.c2w_slam = c.c2w_slam.cpu()
    cam.invalidate()
    return model, cam, rc


VALU_CYCLES = 2.0   # a wave64 VALU instruction on CDNA4's SIMD-32 (MI355X_MICROARCH.md; tools/probe/valu_rate.hip: 1.8 s_memtime
#                     ticks per instruction per SIMD with 8 resident waves in steady state -- v_fma, packed and VOP3 / SGPR / literal
#                     forms alike --, 3.4 for v_exp / v_rcp, 3.7 for v_readlane; ONE wave issues at most one per 5.3 ticks)
SHADER_GHZ = 2.4


def _pmc(kernel, N, tol=0.02):
    """profiles/pmc_<kernel>.json (tools/profile.sh: separate --pmc passes over THIS program's micro-benchmark loops) if it was
    collected on a scene of the same size (its recorded Gaussian count within 2 %), else None"""
    import json
    import re
    fn = "pmc_binning.json" if kernel.startswith("binning") else \
        "pmc_%s.json" % re.sub(r"[^A-Za-z0-9_]+", "_", kernel.split(" (")[0]).strip("_")   # (tools/pmc_extract.pmc_file_name)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", fn)
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    n = (rec.get("units") or {}).get("gaussians")
    return rec if n and abs(n - N) <= tol * N else None


def _kernel_row(name, calls_per_frame, t, alg_bytes, hbm_peak_gbs, N, bound, note=""):
    row = {"kernel": name, "calls_per_frame": calls_per_frame, "avg_us": t * 1e6, "us_per_frame": calls_per_frame * t * 1e6,
           "algorithmic_bytes": alg_bytes, "achieved_GBs": alg_bytes / t / 1e9, "frac": alg_bytes / t / 1e9 / hbm_peak_gbs,
           "bound": bound}
    if note:
        row["note"] = note
    rec = _pmc(name, N)
    if rec:
        if rec.get("hbm_bytes_per_launch"):
            row["traffic"] = rec["hbm_bytes_per_launch"]
            row["traffic_library_commit"] = rec.get("library_commit", "unknown (collected before round 6)")
            row["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / max(1.0, alg_bytes)
            # 2 x FETCH_SIZE + WRITE_SIZE.  The factor 2 is measured for streams AND for narrow gathers on this gfx950: every L2 miss
            # -- a 4-byte gather of an otherwise untouched line included -- is ONE 128-byte memory-side request that FETCH_SIZE
            # tallies as 64 bytes (tools/probe/fetch_calib.hip, profiles/r05_fetch_calibration.md: TCC_EA0_RDREQ_128B == lines touched)
            row["traffic_calibrated"] = True
        v = (rec.get("sq") or {}).get("SQ_INSTS_VALU")
        if v:
            # counter collection serialises and slows the launches; the instruction COUNT carries over and is priced against
            # this run's live launch time
            row["valu_wave_instructions"] = v
            row["valu_issue_frac"] = v * VALU_CYCLES / (1024 * SHADER_GHZ * 1e9 * t)
        for k in ("wait_any_frac", "lds_conflict_per_active_lds"):
            if k in (rec.get("sq") or {}):
                row[k] = rec["sq"][k]
    return row


def fused_pbwd_bytes(N, Nv):
    """what the fused preprocessing backward + Adam kernel has to move: parameters, exp_avg and exp_avg_sq of all 59 floats per
    Gaussian read and written (24 x 59 N), the radius of every Gaussian (4 N), the rasterizer's 10-float gradient row and the
    conic of every visible one (52 Nv).  The 59-float gradient itself lives in registers."""
    return 24.0 * 59 * N + 4.0 * N + 52.0 * Nv


def _with_own_bytes(row, own_bytes, t, hbm_peak_gbs, note):
    """the bytes THIS kernel has to move (its own inputs once + its own outputs), beside SURVEY 8(d)'s figure for the reference's
    algorithm in `algorithmic_bytes` (the contract's yardstick): where the two differ the fraction of peak on the kernel's own
    bytes is the one that says how well the kernel runs"""
    row["own_bytes"] = own_bytes
    row["own_frac"] = own_bytes / t / 1e9 / hbm_peak_gbs
    row["own_bytes_note"] = note
    if row.get("traffic"):
        row["traffic_over_own"] = row["traffic"] / max(1.0, own_bytes)
    return row


def _with_survey_formula(row, survey_bytes, t, hbm_peak_gbs, note):
    row["survey_formula_bytes"] = survey_bytes
    row["survey_formula_frac"] = survey_bytes / t / 1e9 / hbm_peak_gbs
    row["survey_formula_note"] = note
    return row


def _sequential_in_loop(result, top, pre_bytes, hbm_peak_gbs):
    k = (((result.get("in_loop_sequential") or {}).get("kernels")) or {}).get(top["kernel"])
    if not k:
        return {}
    b = top["algorithmic_bytes"] + (k["launches_flagged"] / k["launches"] * pre_bytes if top["kernel"] == "preprocess_bwd_kernel" else 0.0)
    return {"avg_launch_us_sequential": k["avg_us"], "frac_sequential": b / k["avg_us"] / 1e3 / hbm_peak_gbs}


def measured_copy_bandwidth(device, nbytes=1 << 30, reps=10):
    """"record the measured copy bandwidth of the box" -- a device-to-device copy of 1 GiB (read + write counted),
    best of `reps`, HIP events on the current stream.  `roofline.peak` stays the nominal 8 TB/s; this is the box's own ceiling."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=device)
    b = torch.empty_like(a)
    b.copy_(a)
    best = float("inf")
    for _ in range(reps):
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record(); b.copy_(a); s1.record()
        torch.cuda.synchronize()
        best = min(best, s0.elapsed_time(s1) * 1e-3)
    del a, b
    return 2.0 * nbytes / best / 1e9


def roofline_section(scene, seq, result, hbm_peak_gbs, K, gt_pose=False):
    """`roofline` of the bench line.  Every duration is measured live, here, with HIP events on the launch stream over 50
    back-to-back launches of the kernel on the state the timed run ended in (Python twins of the model and of the TSDF
    engine: same C-ABI, same buffers; the timed region never touched them); nothing stored is divided by anything live.

    `kernels`: the kernels that carry a SLAM frame -- per kernel the live average launch time, launches per frame in the settled
    loop, algorithmic (compulsory) HBM bytes per launch (evaluated with the scene's own N, Nv, I, G, P, T, V),
    the fraction of the 8 TB/s HBM peak, and, where a PMC file of the same scene exists (tools/profile.sh), measured HBM traffic
    and the VALU issue fraction.  The top-level `kernel` / `achieved` / `frac` are those of the kernel with the LARGEST share of
    a frame's GPU time (calls x average), so the headline fraction is the dominant kernel's whatever it is.
    `iteration` / `frame`: all algorithmic bytes of one optimise iteration / one frame over their measured times."""
    from gps_slam_amd._lib import lib
    device = "cuda:%d" % torch.cuda.current_device()
    model, cam, rc = _python_twin(scene, device)
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    model.initOptimizers(-1, 1.0)
    lib.gps_set_frame_chain_reserve(0)   # the micro-loops below time kernels that have the chip to themselves (the last scene ran overlapped)

    def step():   # the product's iteration: the next iteration's preprocessing rides in this one's backward kernel (next_cam)
        model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"], next_cam=cam)

    def plain_step():
        model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
    plain_step()
    torch.cuda.synchronize()
    B, st = model._B, model._step
    counts = B["counts"].cpu().tolist()
    ni, nvis = int(counts[0]), int(counts[3])
    W, H, N = st.width, st.height, st.N
    P, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    r = B["radii"][:N].long()
    ng = int(((4 * r * r + 31) // 32)[r > 0].sum())   # the reference's 32-pixel groups (isect_tiles_no_depth.cu:87)
    ref = rc["depth_map_clamped"]
    ptr = lambda t: C.c_void_p(t.data_ptr())
    strips = "v_rows" in B

    # (the fused step deals the tile workgroups by descending list length: the order the superblock binning left in the workspace)
    order = C.c_void_p(lib.gps_isect_workspace_tile_order(ptr(B["workspace"]), N, st.isect_capacity)) if strips else C.c_void_p(0)

    def fwd():  # the forward the fused step launches (packed-math kernel over the preprocess records), without the compose epilogue
        lib.gps_raster_ges_fwd_rec_ordered(N, ptr(B["records"]), ptr(ref), W, H, ptr(B["tile_offsets"]), ptr(B["flatten_ids"]),
                                           ptr(B["counts"]), model.delta_depth, ptr(B["render_colors"]), ptr(B["weight_sum"]), order, sp)

    def bwd():
        if strips:
            lib.gps_raster_ges_bwd_strips(N, ptr(B["records"]), ptr(B["radii"]), ptr(B["cls_ids"]), ptr(B["cls_counts"]), st.cls_stride,
                                          ptr(B["v_render_colors"]), ptr(B["pix2"]), W, H, ptr(B["v_rows"]), sp)
        else:
            lib.gps_raster_ges_bwd_gs(N, ptr(B["means2d"]), ptr(B["conics"]), ptr(B["colors"]), ptr(B["opacities"]),
                                      ptr(B["radii"]), ptr(ref), W, H, ptr(B["group_gs_ids"]), ptr(B["group_starts"]),
                                      ptr(B["counts"]), model.delta_depth, ptr(B["v_render_colors"]),
                                      ptr(B["v_render_alphas"]), ptr(B["v_means2d"]), ptr(B["v_conics"]), ptr(B["v_colors"]),
                                      ptr(B["v_opacities"]), 1, sp)

    pr = model.opt_gs_params
    cg = cam.toGPU()

    def pre():
        lib.gps_gauss_preprocess_fwd(N, pr.K, model.degreesToUse, ptr(pr._buf["means"]), ptr(pr._buf["scales"]), ptr(pr._buf["quats"]),
                                     ptr(pr._buf["opacities"]), ptr(pr._buf["featuresDc"]), ptr(pr._buf["featuresRest"]),
                                     ptr(cg["viewmat"]), ptr(cg["K"]), ptr(cg["cam_pos"]), W, H, model.eps2d, model.near_plane,
                                     model.far_plane, model.radius_clip, int(model.max_gs_radii), ptr(B["radii"]), ptr(B["means2d"]),
                                     ptr(B["depths"]), ptr(B["conics"]), ptr(B["colors"]), ptr(B["opacities"]), ptr(B["records"]), sp)

    def render():  # preprocess (+ histogram) + scan + scatter + forward
        model._bind_camera(st, cam, rc["depth_map_clamped"], rc["color_map"], None)
        lib.gps_splat_render(C.byref(st), sp)

    # preprocessing backward + fused Adam of all six tensors, as the train step launches it (fuse mode 2); the operator-level
    # entry point reads the rasterizer's gradients from four arrays instead of the strip kernel's rows: same kernel, same bytes
    from gps_slam_amd._lib import AdamSegment
    o = model._opt
    if strips:
        rows = B["v_rows"][:N]
        for name, sl in (("v_colors", slice(0, 4)), ("v_conics", slice(4, 7)), ("v_means2d", slice(7, 9))):
            B[name][:N] = rows[:, sl]
        B["v_opacities"][:N] = rows[:, 9]
    seg = (AdamSegment * 5)()
    for j, (k, pname) in enumerate(((0, "means"), (1, "scales"), (2, "quats"), (3, "featuresDc"), (5, "opacities"))):
        seg[j].param, seg[j].grad = pr._buf[pname].data_ptr(), o["g"][k].data_ptr()
        seg[j].exp_avg, seg[j].exp_avg_sq = o["m"][k].data_ptr(), o["v"][k].data_ptr()
        seg[j].numel, seg[j].lr = pr._buf[pname][:N].numel(), 0.0   # (lr 0: the micro-benchmark leaves the parameters alone)
    null = C.c_void_p(0)

    def pbwd():
        lib.gps_gauss_preprocess_bwd_adam(N, pr.K, model.degreesToUse, ptr(pr._buf["means"]), ptr(pr._buf["scales"]), ptr(pr._buf["quats"]),
                                          ptr(pr._buf["opacities"]), ptr(pr._buf["featuresDc"]), ptr(pr._buf["featuresRest"]),
                                          ptr(cg["viewmat"]), ptr(cg["K"]), ptr(cg["cam_pos"]), W, H, model.eps2d, ptr(B["radii"]),
                                          ptr(B["conics"]), ptr(B["v_means2d"]), ptr(B["v_conics"]), ptr(B["v_colors"]),
                                          ptr(B["v_opacities"]), null, null, null, null, null, null, ptr(o["m"][4]), ptr(o["v"][4]), 0.0,
                                          seg, 0.9, 0.999, 1e-15, 2, sp)   # (step 2: the steady iteration -- step 1 does not read the moments)

    marker = lambda: torch.cuda._sleep(1)   # (tools/profile.sh finds each loop's launches behind its marker)
    t = {}
    for name, fn, n in (("bwd", bwd, 50), ("fwd", fwd, 50), ("pre", pre, 50), ("render", render, 50), ("step", step, 20),
                        ("pbwd", pbwd, 20)):
        marker()
        t[name] = _time_launches(fn, n, stream)
        if name == "step":
            plain_step()   # (consumes the last run-ahead forward: the loops below use the step buffers directly)
    t_bin = max(1e-7, t["render"] - t["pre"] - t["fwd"])          # derived: scan + scatter (+ the histogram's share of preprocess)
    t_pbwd = t["pbwd"]
    fus = _fusion_timings(seq, gt_pose, device)
    V = fus["visible_blocks"]
    S = 0x100000 + 0x20000
    b_iter, terms = iteration_bytes(N, nvis, ni, ng, P, T)
    s_bar = fus["s_bar"]
    b_fuse = fusion_bytes(P, V, S)
    b_fuse_rays = fusion_bytes(P, V, S, s_bar)
    bwd_row = _kernel_row("raster_ges_bwd_strip_kernel" if strips else "raster_ges_bwd_gs_kernel", 2.0, t["bwd"], 92.0 * ng + 24.0 * P,
                          hbm_peak_gbs, N, "latency (pixel gathers) / valu")
    if strips:
        bwd_row = _with_own_bytes(bwd_row, 104.0 * nvis + 24.0 * P, t["bwd"], hbm_peak_gbs,
                                  "one task per visible Gaussian: class-list id 4 + radius 4 + 48-byte record in, ONE 48-byte gradient row out "
                                  "(104 Nv); the gradient image and the {v_alpha, depth cut} pair image once (24 P); the box's pixel "
                                  "gathers are cache-served re-reads.  algorithmic_bytes is SURVEY's 92 G + 24 P for the reference's "
                                  "32-pixel groups, which this kernel does not have")
    rows = [
        bwd_row,
        _kernel_row("raster_ges_fwd_pk_kernel", 2.1, t["fwd"], 44.0 * ni + 28.0 * P, hbm_peak_gbs, N, "valu issue + per-tile tail"),
        _with_survey_formula(
            _kernel_row("preprocess_bwd_kernel", 2.0, t_pbwd, fused_pbwd_bytes(N, nvis), hbm_peak_gbs, N, "hbm",
                        "projection + SH backward with the Adam step of all 59 parameters fused (fuse mode 2): the gradient is "
                        "consumed in registers, never written or re-read -- algorithmic_bytes = 24 x 59 N (parameters and both "
                        "moments, read and written) + 4 N (radii) + 52 Nv (the rasterizer's gradient row + conic)"),
            524.0 * nvis + 40.0 * N + 28.0 * 59 * N, t_pbwd, hbm_peak_gbs,
            "SURVEY 8(d)'s figure for the UNFUSED reference chain (SH bwd 408 Nv + projection bwd 116 Nv + 40 N + Adam 28 x 59 N: "
            "gradients written, zeroed and re-read); above 1 because those bytes do not exist in the fused kernel"),
        _kernel_row("preprocess_fwd_kernel", 2.1, t["pre"], 68.0 * N + 217.0 * nvis, hbm_peak_gbs, N, "hbm"),
        _with_own_bytes(
            _kernel_row("binning (sb_scan_kernel + sb_scatter_kernel)", 2.1, t_bin, 24.0 * N + 44.0 * ni + 8.0 * ng + 4.0 * T, hbm_peak_gbs,
                        N, "launch latency", "derived: render - preprocess - forward rasterizer; bytes = SURVEY's figure for the reference's "
                        "count + sort + offsets, this implementation writes no key / value arrays"),
            16.0 * N + 4.0 * ni + 4.0 * nvis + 6152.0 * T, t_bin, hbm_peak_gbs,
            "scan: a tile's row of the count table read, zeroed and its prefix written (3 x 512 x 4 B per tile) + tile totals; scatter: "
            "tiles-per-Gaussian 4 + radius 4 + mean 8 per Gaussian (16 N), one id per intersection out (4 I), the backward's class "
            "lists (4 Nv), tile offsets / order (8 T)"),
        dict(_kernel_row("raycast_kernel", 1.0, fus["raycast_s"], 24.0 * P * s_bar + 20.0 * P, hbm_peak_gbs, N,
                         "latency (dependent gathers along the ray)",
                         "algorithmic_bytes = P x S-bar x 24 (one hash entry + one voxel per castRay step, an upper bound: neighbouring "
                         "rays share them in the caches) + 20 P of outputs; S-bar logged by the kernel itself"),
             s_bar=s_bar, reads_per_ray=fus["reads_per_ray"], bytes_without_ray_term=20.0 * P,
             frac_without_ray_term=20.0 * P / fus["raycast_s"] / 1e9 / hbm_peak_gbs),
        _kernel_row("integrate_kernel", 1.0, fus["integrate_s"], 8192.0 * V, hbm_peak_gbs, N, "hbm / valu"),
    ]
    if fus.get("evals_per_frame"):
        ev = fus["evals_per_frame"]
        trk_row = _kernel_row("track_eval_poll_kernel", ev, fus["poll_eval_s"] + fus["poll_spin_s"],
                                     36.0 * P * 0.332, hbm_peak_gbs, N, "host <-> device loop",
                                     "avg_us = GPU residency of one pre-launched evaluation = waiting for the host's argument line + "
                                     "evaluating; bytes assume the evaluations spread evenly over the 4 pyramid levels and count ONE evaluation per "
                                     "launch (the pose that rides along, config.tracker, reads as much again when it runs: not counted, the "
                                     "counters see it)")
        # What an evaluation HAS to move, in the 128-byte lines the memory side deals in (profiles/r05_fetch_calibration.md): the
        # level's depth image (4 P_l) + the lines of the FULL-resolution point | normal map (32 B per pixel; the reference keeps the
        # scene side at full resolution at every level, ITMExtendedTracker.cpp:294-298) its bilinear footprints touch -- two 64-byte
        # runs per view pixel, a run starting at byte 96 of a line straddles two: levels 0 and 1 touch every line of the map
        # (32 P), level 2 every line of every second row pair (16 P), level 3 a quarter of the rows and 5 of 8 lines there (5 P)
        lv = fus.get("evals_per_level_per_frame") or [0, 0, 0, 0]
        per_level_bytes = [4.0 * P + 32.0 * P, 4.0 * P / 4 + 32.0 * P, 4.0 * P / 16 + 16.0 * P, 4.0 * P / 64 + 5.0 * P]
        own = sum(n_ * b_ for n_, b_ in zip(lv, per_level_bytes)) / max(1e-9, sum(lv))
        trk_row = _with_own_bytes(trk_row, own, fus["poll_eval_s"] + fus["poll_spin_s"], hbm_peak_gbs,
                                  "per evaluation, averaged over the levels with this run's evaluations per level %s: the level's depth "
                                  "image + the 128-byte lines of the full-resolution point | normal map its bilinear footprints touch (a "
                                  "4-byte gather moves a 128-byte line on this device: profiles/r05_fetch_calibration.md).  SURVEY's 36 P_L "
                                  "counts 36 bytes per VIEW pixel, which is right at the finest level only; the counters' traffic is per "
                                  "LAUNCH = this evaluation + the pose riding along, each line fetched by ~2 of the 8 XCDs' L2s" % [round(v, 2) for v in lv])
        rows.append(dict(trk_row,
                         spin_us=fus["poll_spin_s"] * 1e6, eval_us=fus["poll_eval_s"] * 1e6, phases=fus.get("poll_phases"),
                         evals_per_level_per_frame=lv,
                         gpu_held_idle_us_per_frame=ev * fus["poll_spin_s"] * 1e6,
                         tracking_ms_per_frame=fus["tracking_ms_per_frame"]))
    if fus.get("freeview"):
        fv = fus["freeview"]
        rows.append(dict(_kernel_row("raycast_kernel<false> (free views)", 0.9, fv["raycast_s"], 24.0 * P * fv["s_bar"] + 20.0 * P,
                                     hbm_peak_gbs, N, "latency (dependent gathers along the ray)",
                                     "the 2 + 7 free views of a map update, 9 per 10 frames; timed one view per launch"),
                         s_bar=fv["s_bar"]))
        rows.append(_kernel_row("colour_kernel", 0.9, fv["colour_s"], 56.0 * P, hbm_peak_gbs, N, "latency (corner gathers)",
                                "16 P of rays in, 4 P of colour + 36 P of view maps out (the batched launch writes the maps; timed "
                                "here without them); the eight corner voxels are cache-served gathers, not counted"))
        rows.append(_kernel_row("expected_depths_partial_kernel", 1.9, fv["ed_s"], 16.0 * fv["visible_blocks"] + 128.0 * fv["cells"] * 8,
                                hbm_peak_gbs, N, "latency (one entry per thread, LDS atomics)",
                                "16 B per visible entry in, 128 partial min/max images out (one visible block per thread); pass B rides in the raycaster"))
    # In-loop figures of the schedule `value` reports (result["in_loop"], in_loop_kernel_times): per instrumented kernel the
    # average of ALL its launches in K frames of the running loop.  The fused backward's launches that carry the next
    # iteration's preprocessing forward (flagged) move that kernel's bytes as well: counted by their share.
    il = dict((result.get("in_loop") or {}).get("kernels") or {})
    if "sb_scan_kernel" in il and "sb_scatter_kernel" in il:   # the binning row = one scan + one scatter launch
        a_, b_ = il["sb_scan_kernel"], il["sb_scatter_kernel"]
        il["binning (sb_scan_kernel + sb_scatter_kernel)"] = {
            "launches": b_["launches"], "avg_us": a_["avg_us"] + b_["avg_us"], "us_per_frame": a_["us_per_frame"] + b_["us_per_frame"],
            "max_us": a_["max_us"] + b_["max_us"], "launches_flagged": 0}
    pre_bytes = 68.0 * N + 217.0 * nvis
    for row in rows:
        k = il.get(row["kernel"])
        if not k:
            continue
        b = row["algorithmic_bytes"]
        own = row.get("own_bytes")
        if row["kernel"] == "preprocess_bwd_kernel":
            share = k["launches_flagged"] / k["launches"]
            b, row["in_loop_tail_forward_share"], row["in_loop_tail_forward_bytes"] = b + share * pre_bytes, share, pre_bytes
        row.update(in_loop_avg_us=k["avg_us"], in_loop_launches=k["launches"], in_loop_us_per_frame=k["us_per_frame"], in_loop_max_us=k["max_us"],
                   in_loop_bytes=b, in_loop_achieved_GBs=b / k["avg_us"] / 1e3, in_loop_frac=b / k["avg_us"] / 1e3 / hbm_peak_gbs)
        if own:
            row["in_loop_own_frac"] = own / k["avg_us"] / 1e3 / hbm_peak_gbs
    rows.sort(key=lambda x: -x.get("in_loop_us_per_frame", x["us_per_frame"]))
    timed = [r_ for r_ in rows if "in_loop_avg_us" in r_]
    top = timed[0] if timed else rows[0]
    in_loop = bool(timed)
    t_frame = result["ms_per_step"] * 1e-3
    b_frame = 2.0 * b_iter + b_fuse
    copy_gbs = measured_copy_bandwidth(device)
    return {"bound": "hbm", "kernel": top["kernel"],
            "achieved": top["in_loop_achieved_GBs"] if in_loop else top["achieved_GBs"], "peak": hbm_peak_gbs, "unit": "GB/s",
            "timed_in": ("the running %s schedule: every launch of the kernel stamps the device clock at its first wave's start and last wave's "
                         "end (gps_launch_timing_*), averaged over all its launches in %d extra frames after the timed windows" % (result.get("schedule", "?"), (result.get("in_loop") or {}).get("frames", 0)))
                        if in_loop else "kernels-alone micro-loop (no in-loop window in this run)",
            "launches_timed": top.get("in_loop_launches"),
            "frac_alone": top["frac"], "avg_launch_us_alone": top["avg_us"],
            "in_loop_window_ms_per_step": (result.get("in_loop") or {}).get("window_ms_per_step"),
            # the same kernel inside the SEQUENTIAL schedule (nothing runs beside it: the figure a rocprofv3 table of either schedule's
            # in-loop window reproduces to a few percent; the overlap figure moves with how much of the other chain it meets)
            **_sequential_in_loop(result, top, pre_bytes, hbm_peak_gbs),
            "measured_copy_GBs": copy_gbs,
            "measured_copy_note": "device-to-device copy of 1 GiB on this box, read + write bytes / best of 10 (SURVEY 8(d)); fractions use the nominal peak",
            "frac": top["in_loop_frac"] if in_loop else top["frac"], "traffic": top.get("traffic"),
            "avg_launch_us": top["in_loop_avg_us"] if in_loop else top["avg_us"],
            "traffic_source": "profiles/pmc_*.json: separate rocprofv3 --pmc passes over this program's micro-loops on a scene of the same "
                              "size (N within 2 %), committed -- not measured in this run; 2 x FETCH_SIZE + WRITE_SIZE, factor calibrated "
                              "for streams and gathers (profiles/r05_fetch_calibration.md)",
            "traffic_calibrated": bool(top.get("traffic_calibrated", False)),
            "traffic_library_commit": top.get("traffic_library_commit"),
            "algorithmic_bytes": top["in_loop_bytes"] if in_loop else top["algorithmic_bytes"],
            "dominant_by": "largest in-loop time per frame (launches x average duration inside the running schedule) among the kernels "
                           "that move data; kernels[] is sorted by it.  track_eval_poll_kernel's residency is mostly its wait for the "
                           "host's argument line and is not instrumented: its row carries the micro-loop figures",
            "kernels": rows,
            "units": {"gaussians": N, "n_visible": nvis, "n_isects": ni, "n_groups": ng, "pixels": P, "tiles": T, "visible_blocks": V},
            "iteration": {"algorithmic_bytes": b_iter, "terms": terms, "avg_us": t["step"] * 1e6,
                          "achieved_GBs": b_iter / t["step"] / 1e9, "frac": b_iter / t["step"] / 1e9 / hbm_peak_gbs},
            "frame": {"algorithmic_bytes": b_frame, "fusion_bytes_without_ray_term": b_fuse, "fusion_bytes_with_ray_term": b_fuse_rays,
                      "s_bar": s_bar, "ms": t_frame * 1e3, "achieved_GBs": b_frame / t_frame / 1e9,
                      "frac": b_frame / t_frame / 1e9 / hbm_peak_gbs,
                      "frac_with_ray_term": (2.0 * b_iter + b_fuse_rays) / t_frame / 1e9 / hbm_peak_gbs},
            "fusion": {k: v for k, v in fus.items() if k.endswith("_ms_per_frame") or k in ("visible_blocks", "evals_per_frame")},
            "micro_order": ["bwd", "fwd", "pre", "render", "step", "pbwd", "integrate", "raycast", "freeview"] +
                           (["tracked"] if fus.get("evals_per_frame") else []),
            "note": "no kernel on this path is a dense contraction (no MFMA); the rasterizers are latency / VALU-issue bound, their HBM "
                    "fraction is small by construction and is reported because it is the contract's yardstick"}


def config_units(scene, seq, ms_per_step, hbm_peak_gbs, with_tracking_bytes=True):
    """Roofline units of ONE extra configuration (bench.other_configs): the scene's own N, Nv, I, G, P, T, V after its timed
    windows (Python twin of the model: one train step on the last optimise camera; the engine's counters), the algorithmic
    bytes of an optimise iteration and of a frame (the frame without the ray term -- S-bar is logged on the
    headline configuration), a live iteration time (20 back-to-back steps, HIP events) and the two HBM fractions."""
    from gps_slam_amd._lib import lib
    device = "cuda:%d" % torch.cuda.current_device()
    model, cam, rc = _python_twin(scene, device)
    model.initOptimizers(-1, 1.0)
    lib.gps_set_frame_chain_reserve(0)   # (kernels alone, as in roofline_section)
    stream = torch.cuda.current_stream()

    def step():
        model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"], next_cam=cam)
    model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
    torch.cuda.synchronize()
    B, st = model._B, model._step
    counts = B["counts"].cpu().tolist()
    ni, nvis = int(counts[0]), int(counts[3])
    W, H, N = st.width, st.height, st.N
    P, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    r = B["radii"][:N].long()
    ng = int(((4 * r * r + 31) // 32)[r > 0].sum())
    V = int(scene.engine.counters().cpu()[2])
    t_iter = _time_launches(step, 20, stream)
    b_iter, _ = iteration_bytes(N, nvis, ni, ng, P, T)
    b_frame = 2.0 * b_iter + fusion_bytes(P, V, 0x100000 + 0x20000)
    t_frame = ms_per_step * 1e-3
    return {"units": {"gaussians": N, "n_visible": nvis, "n_isects": ni, "n_groups": ng, "pixels": P, "tiles": T, "visible_blocks": V},
            "iteration": {"algorithmic_bytes": b_iter, "avg_us": t_iter * 1e6, "frac": b_iter / t_iter / 1e9 / hbm_peak_gbs},
            "frame": {"algorithmic_bytes": b_frame, "ms": ms_per_step, "frac": b_frame / t_frame / 1e9 / hbm_peak_gbs}}


def _fusion_timings(seq, gt_pose, device, n_sub=12):
    """TSDF side of the roofline section on a Python twin engine fed the LAST n_sub frames of the sequence (re-based to their
    first camera): live durations of integrate_kernel and raycast_kernel (HIP events, 20 launches), the untracked and tracked
    whole-frame times, and the pre-launched tracker evaluations' own profile (gps_track_poll_profile)."""
    from gps_slam_amd._lib import lib
    from gps_slam_amd.tsdf_engine import TsdfEngine, pose_from_c2w
    W, H = seq["W"], seq["H"]
    n = seq["rgb"].shape[0]
    lo = max(0, n - n_sub)
    c0inv = np.linalg.inv(seq["c2w"][lo].astype(np.float64))
    c2w = [(c0inv @ seq["c2w"][k].astype(np.float64)).astype(np.float32) for k in range(lo, n)]
    rgba = [torch.as_tensor(np.concatenate([seq["rgb"][k], np.full((H, W, 1), 255, np.uint8)], -1)).to(device) for k in range(lo, n)]
    dmm = [torch.as_tensor(seq["depth"][k].astype(np.int16)).to(device) for k in range(lo, n)]
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, 0.2, 10.0, device=device)
    for k in range(len(c2w)):
        M, invM = eng.ProcessFrame(rgba[k], dmm[k], c2w[k])
    torch.cuda.synchronize()
    V = int(eng.counters_host()[2])
    torch.cuda._sleep(1)
    t_int = _time_launches(lambda: lib.gps_tsdf_integrate(C.byref(eng.state), M.ctypes.data, sp), 20, stream)
    torch.cuda._sleep(1)
    t_ray = _time_launches(lambda: lib.gps_tsdf_raycast(C.byref(eng.state), invM.ctypes.data, 0, 1, sp), 20, stream)
    st1 = eng.ray_stats()   # of the last launch of the loop above (the launches are identical)
    k_last = len(c2w) - 1
    t_frame = _time_launches(lambda: eng.ProcessFrame(rgba[k_last], dmm[k_last], c2w[k_last]), 10, stream)
    d_rays = max(1, st1["rays"])
    out = {"visible_blocks": V, "integrate_s": t_int, "raycast_s": t_ray, "untracked_ms_per_frame": t_frame * 1e3,
           "s_bar": st1["steps"] / d_rays, "reads_per_ray": st1["reads"] / d_rays}
    # the kernels of a map update's free views, one view per launch (the product batches 2 + 7 views per launch: same kernels with
    # blockIdx.z = view), on a pose a few frames back
    fM, fInv = pose_from_c2w(c2w[max(0, k_last - 5)])
    torch.cuda._sleep(1)   # marker: "freeview"
    state = C.byref(eng.state)
    lib.gps_tsdf_find_visible(state, fM.ctypes.data, sp)
    # (pass A alone, as the product launches it in front of the raycaster -- which does pass B and publishes the block count: the
    # full two-launch form runs once afterwards so that the scratch counter these launches add to is cleared)
    t_ed = _time_launches(lambda: lib.gps_tsdf_expected_depths_partial(state, fM.ctypes.data, 1, sp), 20, stream)
    lib.gps_tsdf_expected_depths(state, fM.ctypes.data, 1, sp)
    t_fray = _time_launches(lambda: lib.gps_tsdf_raycast(state, fInv.ctypes.data, 1, 0, sp), 20, stream)
    fs1 = eng.ray_stats()
    t_col = _time_launches(lambda: lib.gps_tsdf_render_colour(state, sp), 20, stream)
    out["freeview"] = {"ed_s": t_ed, "raycast_s": t_fray, "colour_s": t_col, "visible_blocks": int(eng.counters_host()[3]),
                       "cells": (W // 8 + 2) * (H // 8 + 2),
                       "s_bar": fs1["steps"] / max(1, fs1["rays"])}
    if not gt_pose:
        trk = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, 0.2, 10.0, device=device)
        trk.turnOnTracking(host_summed_rows=not os.environ.get("GPS_BENCH_DEVICE_SUMMER"),
                           poses_riding_along=int(os.environ.get("GPS_BENCH_RIDING_ALONG", 1)))
        trk.ProcessFrameTracked(rgba[0], dmm[0])
        trk.ProcessFrameTracked(rgba[1], dmm[1])
        torch.cuda.synchronize()
        prof0 = trk.track_poll_profile()
        ph0 = trk.track_poll_phases()
        torch.cuda._sleep(1)   # marker: "tracked"
        t0 = time.perf_counter()
        per_level = np.zeros(8)
        for k in range(2, len(c2w)):
            trk.ProcessFrameTracked(rgba[k], dmm[k])
            per_level += np.asarray(trk.track_state.diag[:8], np.float64)   # evaluations the LM loop consumed per pyramid level
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / max(1, len(c2w) - 2)
        prof1 = trk.track_poll_profile()
        d = [(b - a) & 0xFFFFFFFF for a, b in zip(prof0, prof1)]
        ph1 = trk.track_poll_phases()
        phases = {}
        for key in ("level0", "coarse"):   # per evaluation, microseconds (100 MHz ticks)
            dd = [(b - a) & 0xFFFFFFFF for a, b in zip(ph0[key], ph1[key])]
            cnt = max(1, dd[0])
            phases[key] = {"evaluations_per_frame": dd[0] / max(1, len(c2w) - 2), "own_pixel_loop_us": dd[1] * 0.01 / cnt,
                           "until_all_rows_us": dd[2] * 0.01 / cnt, "sum_and_mailbox_us": dd[3] * 0.01 / cnt}
        out["poll_phases"] = phases
        out["evals_per_level_per_frame"] = [float(v) / max(1, len(c2w) - 2) for v in per_level[:4]]
        evals = max(1, d[2])
        out.update(tracked_ms_per_frame=dt * 1e3, tracking_ms_per_frame=max(0.0, dt - t_frame) * 1e3,
                   evals_per_frame=d[2] / max(1, len(c2w) - 2), poll_spin_s=d[0] * 1e-8 / (evals + d[3]),
                   poll_eval_s=d[1] * 1e-8 / evals)
    return out


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
        c.image = torch.as_tensor(seq["rgb"][k])                       # uint8 / uint16 millimetres, as bench.Scene hands them over
        c.depth = torch.as_tensor(seq["depth"][k].view(np.int16))
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
        # the same loop with the reference's depth tracker estimating every pose (what the headline's frames do): a shorter sample
        nt = int(max(6, min(n, 1 + 0.4 * max_seconds / per)))
        trk = R.time_reference(seq, nt, 0.005, 0.02, 0.2, 10.0, threads=best, track=True)
        from gps_slam_amd.dist_util import cpu_quota
        quota = cpu_quota()
        return {"value": res["frames"] / res["seconds"], "unit": "frames/s", "cores": cores, "threads": best, "kind": "reference",
                "cpu_quota": quota,
                "tracked_value": trk["frames"] / trk["seconds"] if trk else None, "tracked_sample_frames": trk["frames"] if trk else 0,
                "thread_sweep_frames_per_s": {str(t): round(v, 3) for t, v in sweep.items()},
                "sample": "%d ProcessFrame calls (TSDF fuse + live raycast + ICP maps, tracking off, no Gaussians) of the same "
                          "%dx%d synthetic sequence by the reference's ITMLib CPU engine (oracle/_ref/itm_ref_omp: g++ -O3 "
                          "-fopenmp as upstream), OMP_NUM_THREADS=%d = the fastest of a sweep over %s threads (6 frames each) on %d "
                          "physical cores%s, first frame excluded; tracked_value = the same loop with the engine's depth tracker on; "
                          "host CPU: %s" % (res["frames"], W, H, best, "/".join(str(t) for t in sweep), cores,
                                            "" if quota is None else " of which this container may use %.3g at a time (cgroup cpu.max: "
                                            "thread counts above that are throttled, as the sweep shows)" % quota, _cpu_name())}
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
    return {"value": timed / t_used if t_used > 0 else 0.0, "unit": "frames/s", "cores": 1, "threads": 1, "kind": "port",
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

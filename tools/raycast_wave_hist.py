"""Is the raycaster as long as its longest rays?  Per wave (8 x 8 rays = one cell of the min/max image) the live raycast logs the
sum of its rays' loop trips and -- round 6 -- the trips of its LONGEST ray (gps_tsdf_ray_wave_rows).  A wave stays resident until
its longest ray has left the march loop and a trip is one dependent memory round trip, so a wave's time is ~ its longest ray's
trips x the trip latency; the launch's time is bounded below by (a) the longest wave and (b) total wave-trips / waves in flight.
This tool prints the distribution on the bench scene after `n` frames, the two bounds with the measured launch time, and what
pairing long with short waves could buy; the histogram goes to <out>.json / .md (copy into profiles/).

usage (GPU box): python tools/raycast_wave_hist.py [--frames 60] [--out gpurun_out/r06_raycast_waves]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gps_slam_amd._lib import lib  # noqa: E402
from gps_slam_amd.tsdf_engine import TsdfEngine, pose_from_c2w  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_raycast_waves"))
    a = ap.parse_args()
    W, H, n = a.width, a.height, a.frames
    dev = "cuda:0"
    seq = bench.synthetic_sequence_device(W, H, n, 1234, dev)
    eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, device=dev)
    for i in range(n):
        rgb = torch.from_numpy(seq["rgb"][i]).cuda().contiguous()
        dmm = torch.from_numpy(seq["depth"][i].astype(np.int16)).cuda().contiguous()
        eng.ProcessFrame(rgb, dmm, seq["c2w"][i])
    torch.cuda.synchronize()
    M, invM = pose_from_c2w(seq["c2w"][n - 1])
    live = lambda: lib.gps_tsdf_raycast(C.byref(eng.state), invM.ctypes.data, 0, 0, None)
    live(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        live()
    e1.record(); torch.cuda.synchronize()
    t_us = e0.elapsed_time(e1) / 30 * 1e3
    n_rows = ((W + 15) // 16) * ((H + 15) // 16) * 4
    rows = torch.zeros((n_rows, 4), device=dev)
    got = lib.gps_tsdf_ray_wave_rows(C.byref(eng.state), C.c_void_p(rows.data_ptr()), n_rows, None)
    torch.cuda.synchronize()
    assert got == n_rows, got
    r = rows.cpu().numpy().astype(np.float64)
    steps, reads, rays, wmax = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    live_w = rays > 0
    mean_trips = reads[live_w] / rays[live_w]
    wmax = wmax[live_w]
    waves = int(live_w.sum())
    # (a) the longest wave, (b) all wave-trips shared by the waves the chip holds at once (256 CUs x 4 SIMDs x 8 waves; the kernel's
    # 256-thread workgroups at its register count: see the resource usage in DESIGN.md) -- in units of one trip's latency
    per_trip_us = t_us / max(1.0, wmax.max())   # if the launch were exactly as long as its longest wave
    total_wave_trips = float(wmax.sum())
    resident = 256 * 4 * 8
    out = {
        "size": "%dx%d" % (W, H), "frames_fused": n, "launch_us": t_us, "waves": waves, "rays": int(rays.sum()),
        "mean_trips_per_ray": float(reads.sum() / rays.sum()), "mean_steps_per_ray": float(steps.sum() / rays.sum()),
        "wave_longest_ray_trips": {"mean": float(wmax.mean()), "median": float(np.median(wmax)), "p90": float(np.percentile(wmax, 90)),
                                   "p99": float(np.percentile(wmax, 99)), "max": float(wmax.max())},
        "wave_mean_ray_trips": {"mean": float(mean_trips.mean()), "median": float(np.median(mean_trips)), "max": float(mean_trips.max())},
        "lane_utilisation": float(reads[live_w].sum() / (wmax * rays[live_w]).sum()),   # trips done / trips the resident lanes sit through
        "histogram_wave_longest_ray_trips": {str(int(k)): int(v) for k, v in zip(*np.unique(np.minimum(wmax, 64).astype(int), return_counts=True))},
        "bounds": {"waves_resident_at_once": resident, "rounds_of_waves": waves / resident,
                   "trip_latency_us_if_launch_is_its_longest_wave": per_trip_us,
                   "sum_of_wave_trips": total_wave_trips,
                   "trip_latency_us_if_every_slot_is_always_busy": t_us * min(waves, resident) / total_wave_trips},
    }
    # pairing: today's launch order puts waves on SIMD slots in row-major patch order; what if each resident slot got an equal share
    # of the summed longest-ray trips (perfect balance) -- the best any re-ordering of whole waves can do
    if waves > resident:
        out["bounds"]["balanced_share_trips_per_slot"] = total_wave_trips / resident
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out + ".json", "w"), indent=1)
    with open(a.out + ".md", "w") as f:
        f.write("# Raycast: per-wave longest ray (tools/raycast_wave_hist.py, MI355X)\n\n")
        f.write("Live raycast of the bench scene after %d fused frames, %dx%d: %.1f us per launch (30 back-to-back), %d waves of 64 rays.\n\n" % (n, W, H, t_us, waves))
        f.write("| | trips |\n|---|---|\n")
        f.write("| mean trips per ray | %.2f |\n| mean castRay steps per ray (the reference's loop) | %.2f |\n" % (out["mean_trips_per_ray"], out["mean_steps_per_ray"]))
        for k, v in out["wave_longest_ray_trips"].items():
            f.write("| longest ray of a wave, %s | %.1f |\n" % (k, v))
        f.write("| lane utilisation (trips done / trips the wave's 64 lanes sit through) | %.3f |\n\n" % out["lane_utilisation"])
        f.write("Histogram of the waves' longest ray (trips: waves):\n\n```\n")
        mx = max(out["histogram_wave_longest_ray_trips"].values())
        for k, v in sorted(out["histogram_wave_longest_ray_trips"].items(), key=lambda kv: int(kv[0])):
            f.write("%3s%s %6d %s\n" % (k, "+" if int(k) == 64 else " ", v, "#" * int(60 * v / mx)))
        f.write("```\n\n")
        b = out["bounds"]
        f.write("The chip holds %d waves at once (256 CUs x 4 SIMDs x 8); the launch has %d = %.2f rounds.\n" % (resident, waves, b["rounds_of_waves"]))
        f.write("If the launch were exactly as long as its longest wave (%d trips) a trip would take %.2f us; if every wave slot were busy from start to end "
                "(sum of the waves' longest-ray trips %.0f shared by %d slots) a trip would take %.2f us.\n"
                % (int(out["wave_longest_ray_trips"]["max"]), b["trip_latency_us_if_launch_is_its_longest_wave"], b["sum_of_wave_trips"],
                   min(waves, resident), b["trip_latency_us_if_every_slot_is_always_busy"]))
    print(open(a.out + ".md").read())


if __name__ == "__main__":
    main()

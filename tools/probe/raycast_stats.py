"""Raycaster step / time statistics on the bench scene.  The instrumented kernel (per-ray step counts and 100 MHz ticks written OVER the
ray image) left the product source in round 4 -- the product kernel now logs S-bar itself (counters[GPS_TSDF_RAY_STEPS..], TsdfEngine.ray_stats());
for the per-wave distributions build the round-3 kernel as a variant:
    mkdir -p /tmp/r3 && git show 0ef4a24:gps_slam_amd/csrc/tsdf_render.hip > /tmp/r3/tsdf_render.hip
    python tools/probe/variant.py stats /tmp/r3/tsdf_render.hip -DGPS_RAYCAST_STATS
usage (GPU box):  GPS_SLAM_HIP_LIB=tools/probe/libgps_stats.so python tools/probe/raycast_stats.py
                  GPS_SLAM_HIP_LIB=tools/probe/libgps_sections.so python tools/probe/raycast_stats.py sections"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from gps_slam_amd.tsdf_engine import TsdfEngine

mode = sys.argv[1] if len(sys.argv) > 1 else "stats"
W, H, n = 640, 480, 40
seq = bench.synthetic_sequence(W, H, n, 1234)
eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, device="cuda:0")
for i in range(n):
    rgb = torch.from_numpy(seq["rgb"][i]).cuda().contiguous()
    dmm = torch.from_numpy(seq["depth"][i].astype(np.int16)).cuda().contiguous()
    eng.ProcessFrame(rgb, dmm, seq["c2w"][i])
torch.cuda.synchronize()
v = eng.GetLiveVertex().cpu().numpy()   # the instrumented kernel wrote its statistics here (last live raycast)
if mode == "stats":
    steps, n_un = v[..., 0], v[..., 1]
    t0 = v[..., 2].view(np.uint32).astype(np.int64); t1 = v[..., 3].view(np.uint32).astype(np.int64)
    dur = ((t1 - t0) & 0xFFFFFFFF) * 0.01  # us (100 MHz)
    print("steps per ray: mean %.1f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f; unallocated-step share %.2f" % (
        steps.mean(), np.percentile(steps, 50), np.percentile(steps, 90), np.percentile(steps, 99), steps.max(), n_un.sum() / steps.sum()))
    ps = steps.reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))          # per wave (8x8 patch): lock-step iterations
    pd = dur.reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
    print("per wave: iterations mean %.1f p90 %.0f max %.0f; duration mean %.1f us p90 %.1f max %.1f; us per iteration %.2f" % (
        ps.mean(), np.percentile(ps, 90), ps.max(), pd.mean(), np.percentile(pd, 90), pd.max(), (pd / np.maximum(ps, 1)).mean()))
    span = (t1.max() - t0.min()) * 0.01
    print("kernel span by the tick counters: %.1f us; waves %d" % (span, ps.size))
    st = ((t0 - t0.min()) * 0.01).reshape(H // 8, 8, W // 8, 8).min(axis=(1, 3)).reshape(-1)
    en = ((t1 - t0.min()) * 0.01).reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3)).reshape(-1)
    print("wave start times: p50 %.1f p90 %.1f max %.1f us; end times p50 %.1f p90 %.1f p99 %.1f max %.1f" % (
        np.percentile(st, 50), np.percentile(st, 90), st.max(), np.percentile(en, 50), np.percentile(en, 90), np.percentile(en, 99), en.max()))
    for lo, hi in ((0, 5), (5, 10), (10, 15), (15, 20), (20, 40)):
        m = (ps.reshape(-1) >= lo) & (ps.reshape(-1) < hi)
        if m.any():
            print("   waves with %2d..%2d iterations: %5d, duration mean %.1f us (%.2f us/iteration)" % (lo, hi, m.sum(), pd.reshape(-1)[m].mean(), (pd.reshape(-1)[m] / np.maximum(ps.reshape(-1)[m], 1)).mean()))
    running = [(int(((st <= t) & (en > t)).sum())) for t in np.arange(0, en.max(), en.max() / 12)]
    print("   waves in flight over time:", running)
    order = np.argsort(pd.reshape(-1))[::-1][:5]
    for o in order:
        print("  slow wave %d: %d iterations, %.1f us" % (o, ps.reshape(-1)[o], pd.reshape(-1)[o]))
else:
    n_un, adv, n_al, rng = (v[..., k].astype(np.float64) for k in range(4))
    live = rng > 0
    print("rays with a range: %.1f%%; range (voxels) mean %.0f p90 %.0f max %.0f" % (100 * live.mean(), rng[live].mean(), np.percentile(rng[live], 90), rng[live].max()))
    print("per live ray: unallocated iterations %.1f (+ %.1f batched skips), in-block iterations %.1f" % (n_un[live].mean(), adv[live].mean(), n_al[live].mean()))
    it = (n_un + n_al).reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
    un = n_un.reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3)); al = n_al.reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
    print("per wave (max over lanes): iterations %.1f, unallocated %.1f, in-block %.1f; worst wave %d iterations" % (it.mean(), un.mean(), al.mean(), it.max()))
    hist = np.histogram(n_al[live], bins=[0, 5, 10, 20, 40, 80, 200])[0]
    print("in-block iterations histogram [0,5,10,20,40,80,200):", hist.tolist())

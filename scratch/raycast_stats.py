"""Per-ray step statistics + per-wave timeline of the raycaster (scratch/statlib/libstats*.so, -DGPS_RAYCAST_STATS)."""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gps_slam_amd._lib as L
L._LIBPATH = os.environ.get("GPS_ALT_LIB", os.path.join(ROOT, "scratch", "statlib", "libstats.so"))
import torch, numpy as np
import bench
W, H, K = 640, 480, 40
seq, eng, model, pipe, cams, rgb_dev, depth_dev = bench.build_scene(W, H, K + 1, 1000, 0, 'cuda:0')
for i in range(K):
    eng.ProcessFrame(rgb_dev[i], depth_dev[i], cams[i].c2w.numpy())
torch.cuda.synchronize()
r = eng.raycast.view(H, W, 4).cpu().numpy()
tot, un = r[..., 0], r[..., 1]
t0 = r[..., 2].copy().view(np.uint32).astype(np.int64); t1 = r[..., 3].copy().view(np.uint32).astype(np.int64)
base = t0.min()
# per wave = 16x4 patch
def wave(a, f): return f(a.reshape(H // 4, 4, W // 16, 16), axis=(1, 3))
ws, we = wave(t0, np.min) - base, wave(t1, np.max) - base
dur = (we - ws) / 100.0  # us (100 MHz)
steps = wave(tot, np.max); uns = wave(un, np.max)
print("kernel span %.1f us; waves %d" % (we.max() / 100.0, dur.size))
print("wave start: p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(ws / 100.0, [50, 90, 100])))
print("wave dur:   mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us" % (dur.mean(), *np.percentile(dur, [50, 90, 99, 100])))
print("wave end:   p50 %.1f p90 %.1f p99 %.1f us" % tuple(np.percentile(we / 100.0, [50, 90, 99])))
print("steps/wave(max lane): mean %.1f p99 %.0f max %.0f" % (steps.mean(), np.percentile(steps, 99), steps.max()))
print("us per step (dur/steps): mean %.2f" % (dur / np.maximum(steps, 1)).mean())
idx = np.argsort(dur.ravel())[-5:]
for i in idx: print("  slow wave: dur %.1f us steps %d unalloc %d start %.1f" % (dur.ravel()[i], steps.ravel()[i], uns.ravel()[i], ws.ravel()[i] / 100.0))
print("corr(dur, steps) %.3f" % np.corrcoef(dur.ravel(), steps.ravel())[0, 1])

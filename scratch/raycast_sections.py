"""Where a raycast wave's time goes (libstats built with -DGPS_RAYCAST_STATS -DGPS_RAYCAST_STATS_SECTIONS)."""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gps_slam_amd._lib as L
L._LIBPATH = os.environ["GPS_ALT_LIB"]
import torch, numpy as np
import bench
W, H, K = 640, 480, 40
seq, eng, model, pipe, cams, rgb_dev, depth_dev = bench.build_scene(W, H, K + 1, 1000, 0, 'cuda:0')
for i in range(K):
    eng.ProcessFrame(rgb_dev[i], depth_dev[i], cams[i].c2w.numpy())
torch.cuda.synchronize()
r = eng.raycast.view(H, W, 4).cpu().numpy() / 100.0
tot = r[..., 3]
# the lane that lives longest in each wave
rr = r.reshape(H // 4, 4, W // 16, 16, 4).transpose(0, 2, 1, 3, 4).reshape(-1, 64, 4)
k = (rr[..., 0] + rr[..., 1] + rr[..., 2]).argmax(1)
lane = rr[np.arange(rr.shape[0]), k]
tail_incl_interp, heads, vox, total = lane.T
print("longest lane per wave, mean us: heads %.1f voxel %.1f tail(after voxel, incl interp) %.1f total %.1f" % (heads.mean(), vox.mean(), tail_incl_interp.mean(), total.mean()))
sel = total >= np.percentile(total, 99)
print("slowest 1%% waves:              heads %.1f voxel %.1f tail %.1f total %.1f" % (heads[sel].mean(), vox[sel].mean(), tail_incl_interp[sel].mean(), total[sel].mean()))

import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import synth
from gps_slam_amd.tsdf_engine import TsdfEngine
W, H, n = 640, 480, 40
seq = synth.make_sequence(W, H, n, step_deg=0.25)
rgba = torch.as_tensor(np.concatenate([seq["rgb"], np.full(seq["rgb"].shape[:-1] + (1,), 255, np.uint8)], -1)).cuda()
dep = torch.as_tensor(seq["depth"].astype(np.int16)).cuda()
eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], voxel_size=0.005, mu=0.02, device="cuda:0")
eng.turnOnTracking()
c0inv = np.linalg.inv(seq["c2w"][0])
its = []
for f in range(n):
    if f == 10:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    M, invM = eng.ProcessFrameTracked(rgba[f], dep[f])
    its.append(eng.track_diag()[:4].sum())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (n - 10)
gt = (c0inv @ seq["c2w"][n - 1]).T.reshape(-1)
print("tracked ProcessFrame: %.2f ms/frame (%.0f fps), mean LM iterations/frame %.1f, final pose error %.2e" % (1e3 * dt, 1 / dt, np.mean(its[10:]), np.abs(invM - gt).max()))

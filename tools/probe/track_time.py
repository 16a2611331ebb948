"""ms per tracked TSDF frame (ProcessFrameTracked of the Python mirror) on the bench sequence for GPS_SLAM_HIP_LIB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from gps_slam_amd.tsdf_engine import TsdfEngine
W, H, n = 640, 480, 60
seq = bench.synthetic_sequence(W, H, n, 1234)
eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, device="cuda:0")
eng.turnOnTracking()
fr = [(torch.from_numpy(seq["rgb"][i]).cuda().contiguous(), torch.from_numpy(seq["depth"][i].astype(np.int16)).cuda().contiguous()) for i in range(n)]
for i in range(20):
    eng.ProcessFrameTracked(*fr[i])
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(20, n):
    eng.ProcessFrameTracked(*fr[i])
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / (n - 20)
d = np.array(eng.track_state.diag[:8])
print("%s: %.3f ms per tracked frame; LM iterations of the last frame per level %s" % (os.environ.get("GPS_SLAM_HIP_LIB", "default"), dt * 1e3, d[:4].tolist()))

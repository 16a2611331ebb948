"""Live + free-view raycast time on the bench scene for the library named by GPS_SLAM_HIP_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
import bench
from gps_slam_amd.tsdf_engine import TsdfEngine, pose_from_c2w
from gps_slam_amd._lib import lib

W, H, n = 640, 480, int(os.environ.get("NFRAMES", 40))
seq = bench.synthetic_sequence(W, H, int(os.environ.get("SEQLEN", n)), 1234)
eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, device="cuda:0")
for i in range(n):
    rgb = torch.from_numpy(seq["rgb"][i]).cuda().contiguous()
    dmm = torch.from_numpy(seq["depth"][i].astype(np.int16)).cuda().contiguous()
    eng.ProcessFrame(rgb, dmm, seq["c2w"][i])
torch.cuda.synchronize()
M, invM = pose_from_c2w(seq["c2w"][n - 1])
ref = eng.GetLiveVertex().clone()
def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t_live = timed(lambda: lib.gps_tsdf_raycast(C.byref(eng.state), invM.ctypes.data, 0, 0, None))
same = torch.equal(eng.GetLiveVertex(), ref)
import hashlib
digest = hashlib.sha1(eng.GetLiveVertex().cpu().numpy().tobytes()).hexdigest()[:12]
M2, invM2 = pose_from_c2w(seq["c2w"][n - 15])
t_free = timed(lambda: lib.gps_tsdf_free_raycast(C.byref(eng.state), M2.ctypes.data, invM2.ctypes.data, None))
mm = eng.minmax.view(H, W, 2)[:H // 8, :W // 8].cpu().numpy()
print("   minmax window: zmin mean %.2f zmax mean %.2f; visible blocks %d" % (mm[..., 0].mean(), mm[..., 1].mean(), int(eng.counters.cpu()[2])))
print("   live vertex map sha1 %s" % digest)
print("%s: live raycast %.1f us (output unchanged: %s), free-view raycast call %.1f us" % (os.environ.get("GPS_SLAM_HIP_LIB", "default"), t_live, same, t_free))

# nine free views: one at a time vs gps_tsdf_free_raycast_batch
poses = [pose_from_c2w(seq["c2w"][max(0, n - 1 - 4 * k)]) for k in range(9)]
def one_by_one():
    for M_, i_ in poses:
        lib.gps_tsdf_free_raycast(C.byref(eng.state), M_.ctypes.data, i_.ctypes.data, None)
eng.runRaycastBatch(poses)
print("9 free views: one at a time %.1f us, batched %.1f us" % (timed(one_by_one, 10), timed(lambda: eng.runRaycastBatch(poses), 10)))
for nv in (2, 4, 7):
    print("   batch of %d: %.1f us" % (nv, timed(lambda: eng.runRaycastBatch(poses[:nv]), 10)))

"""expected_depths_partial_kernel alone on the bench scene (run under rocprofv3 --kernel-trace --stats: tools/probe/ed_time.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
import bench
from gps_slam_amd.tsdf_engine import TsdfEngine, pose_from_c2w
from gps_slam_amd._lib import lib

W, H, n = 640, 480, int(os.environ.get("NFRAMES", 130))
seq = bench.synthetic_sequence_device(W, H, n, 1234, "cuda:0")
eng = TsdfEngine(W, H, seq["fx"], seq["fy"], seq["cx"], seq["cy"], 0.005, 0.02, device="cuda:0")
for i in range(n):
    rgba = torch.as_tensor(np.concatenate([seq["rgb"][i], np.full((H, W, 1), 255, np.uint8)], -1)).cuda()
    dmm = torch.from_numpy(seq["depth"][i].astype(np.int16)).cuda().contiguous()
    eng.ProcessFrame(rgba, dmm, seq["c2w"][i])
torch.cuda.synchronize()
M, invM = pose_from_c2w(seq["c2w"][n - 1])
print("visible blocks", int(eng.counters.cpu()[2]))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for fv in (0,):
    lib.gps_tsdf_expected_depths(C.byref(eng.state), M.ctypes.data, fv, None)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(100):
        lib.gps_tsdf_expected_depths(C.byref(eng.state), M.ctypes.data, fv, None)
    e1.record(); torch.cuda.synchronize()
    print("expected depths (pass A + reduce launch), free_view=%d: %.2f us per call" % (fv, e0.elapsed_time(e1) * 10.0))

"""TSDF stage micro-benchmark: replays the bench scene through the individual C-ABI entry points with HIP-event timing."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch, numpy as np
import bench
if os.environ.get("GPS_ALT_LIB"):
    import gps_slam_amd._lib as L; L._LIBPATH = os.environ["GPS_ALT_LIB"]
from gps_slam_amd._lib import lib, check
from gps_slam_amd.tsdf_engine import pose_from_c2w
W, H, K = 640, 480, 40
seq, eng, model, pipe, cams, rgb_dev, depth_dev = bench.build_scene(W, H, K + 1, 1000, 0, 'cuda:0')
st = eng._stream()
for i in range(K - 1):
    eng.ProcessFrame(rgb_dev[i], depth_dev[i], cams[i].c2w.numpy())
torch.cuda.synchronize()
print("counters", eng.counters_host()[:8])
i = K - 1
M, invM = pose_from_c2w(cams[i].c2w.numpy())
eng.state.rgb = rgb_dev[i].data_ptr()
s = C.byref(eng.state)
stages = [
    ("convert", lambda: lib.gps_tsdf_convert_depth(s, depth_dev[i].data_ptr(), st)),
    ("allocate", lambda: lib.gps_tsdf_allocate(s, M.ctypes.data, invM.ctypes.data, st)),
    ("integrate", lambda: lib.gps_tsdf_integrate(s, M.ctypes.data, st)),
    ("expected", lambda: lib.gps_tsdf_expected_depths(s, M.ctypes.data, 0, st)),
    ("raycast", lambda: lib.gps_tsdf_raycast(s, invM.ctypes.data, 0, 1, st)),
    ("icp_maps", lambda: lib.gps_tsdf_icp_maps(s, invM.ctypes.data, st)),
    ("free_raycast", lambda: lib.gps_tsdf_free_raycast(s, M.ctypes.data, invM.ctypes.data, st)),
]
for name, fn in stages:
    for _ in range(3): check(fn(), name)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): check(fn(), name)
    e1.record(); torch.cuda.synchronize()
    print("%-14s %8.1f us" % (name, e0.elapsed_time(e1) * 50))
print("counters", eng.counters_host()[:8])

"""Where a tile's workgroup spends its time in the forward rasterizer: a probe build (tools/probe/variant.py stamps splat_raster.hip
-DGPS_FWD_STAMPS) writes 100 MHz timestamps at workgroup phases; printed relative to the first workgroup's start.
usage: python tools/probe/fwd_stamps.py tools/probe/libgps_stamps.so"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from bench_kernels import _python_twin
from gps_slam_amd._lib import load_library

W, H, NG = 640, 480, 200000
seq = bench.synthetic_sequence(W, H, 31, 1234)
seeds = bench.seed_gaussians(seq, NG, 1234, "cuda:0")
scene = bench.Scene(seq, seeds, 1234, True, False, 31, 1.0, 0.02)
scene.run(0, 31)
model, cam, rc = _python_twin(scene, "cuda:0")
model.initOptimizers(-1, 1.0)
model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
torch.cuda.synchronize()
B, st = model._B, model._step
N = st.N
L = load_library(sys.argv[1])
STAMPS = "--no-stamps" not in sys.argv   # (--no-stamps: just launch the kernel three times, e.g. under a counter collection)
if STAMPS:
    L.gps_fwd_stamps.restype = C.c_void_p
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
o = (torch.zeros_like(B["render_colors"]), torch.zeros_like(B["weight_sum"]))
hip = C.CDLL("libamdhip64.so")
host = np.zeros(4096 * 8, np.uint64)
for rep in range(3):
    assert L.gps_raster_ges_fwd_rec(N, p(B["records"]), p(rc["depth_map_clamped"]), W, H, p(B["tile_offsets"]), p(B["flatten_ids"]),
                                    p(B["counts"]), model.delta_depth, p(o[0]), p(o[1]), sp) == 0
    torch.cuda.synchronize()
if not STAMPS:
    scene.close()
    sys.exit(0)
hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(L.gps_fwd_stamps()), C.c_size_t(host.nbytes), 2)
t = host.reshape(4096, 8)[:1200, :6].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
names = ["start", "staged (thread 0)", "after staging barrier", "list done (thread 0)", "after sum barrier", "end"]
for k, nm in enumerate(names):
    q = np.percentile(us[:, k], [0, 25, 50, 75, 100])
    print("%-24s min %.1f  q25 %.1f  median %.1f  q75 %.1f  max %.1f us" % (nm, *q))
d = np.diff(us, axis=1)
for k in range(5):
    print("phase %-22s -> %-22s median %.2f  mean %.2f  max %.2f us" % (names[k], names[k + 1], np.median(d[:, k]), d[:, k].mean(), d[:, k].max()))
first = np.argsort(us[:, 0])
print("workgroups started within the first 2 us: %d; started after 10 us: %d" % ((us[:, 0] < 2).sum(), (us[:, 0] > 10).sum()))
scene.close()

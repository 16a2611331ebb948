"""Phases of the persistent forward rasterizer's workgroups (probe build: tools/probe/variant.py stamps splat_raster.hip -DGPS_FWD_STAMPS).
usage: python tools/probe/fwd_pp_stamps.py tools/probe/libgps_stamps.so [width height gaussians fx fy cx cy]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from bench_kernels import _python_twin
from gps_slam_amd._lib import lib, load_library

a = sys.argv[2:]
W, H, NG = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (640, 480, 200000)
intr = tuple(float(x) for x in a[3:7]) if len(a) >= 7 else None
dev = "cuda:0"
bench.prime(dev)
seq = bench.synthetic_sequence_device(W, H, 31, 1234, dev, intrinsics=intr)
seeds = bench.seed_gaussians(seq, NG, 1234, dev)
scene = bench.Scene(seq, seeds, 1234, True, False, 31, 1.0, 0.02)
scene.run(0, 31)
model, cam, rc = _python_twin(scene, dev)
model.initOptimizers(-1, 1.0)
lib.gps_set_frame_chain_reserve(0)
model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
torch.cuda.synchronize()
B, st = model._B, model._step
N = st.N
L = load_library(sys.argv[1])
L.gps_fwd_stamps.restype = C.c_void_p
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
o = (torch.zeros_like(B["render_colors"]), torch.zeros_like(B["weight_sum"]))
order = C.c_void_p(lib.gps_isect_workspace_tile_order(p(B["workspace"]), N, st.isect_capacity))
hip = C.CDLL("libamdhip64.so")
host = np.zeros(4096 * 8, np.uint64)
for rep in range(3):
    assert L.gps_raster_ges_fwd_rec_ordered(N, p(B["records"]), p(rc["depth_map_clamped"]), W, H, p(B["tile_offsets"]), p(B["flatten_ids"]),
                                            p(B["counts"]), model.delta_depth, p(o[0]), p(o[1]), order, sp) == 0
    torch.cuda.synchronize()
hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(L.gps_fwd_stamps()), C.c_size_t(host.nbytes), 2)
G = 768
raw = host.reshape(4096, 8)[:G]
n_mine = raw[:, 7].astype(np.int64)
t = raw[:, :7].astype(np.int64)
us = (t - t[:, 0].min()) / 100.0
names = ["start", "tile table read", "item 0 staged + compacted", "item 0 evaluated (wave 0)", "tile 0 all waves done", "tile 0 stored", "end"]
for k, nm in enumerate(names):
    q = np.percentile(us[:, k], [0, 25, 50, 75, 100])
    print("%-28s min %.1f  q25 %.1f  median %.1f  q75 %.1f  max %.1f us" % (nm, *q))
d = np.diff(us, axis=1)
for k in range(6):
    print("phase %-28s -> %-28s median %.2f  mean %.2f  max %.2f us" % (names[k], names[k + 1], np.median(d[:, k]), d[:, k].mean(), d[:, k].max()))
for n in sorted(set(n_mine.tolist())):
    m = n_mine == n
    rest = (us[m, 6] - us[m, 5])
    print("workgroups with %d tiles: %d; after tile 0: median %.2f us, max %.2f (%.2f per further tile)" % (n, m.sum(), np.median(rest), rest.max(), np.median(rest) / max(1, n - 1)))
scene.close()

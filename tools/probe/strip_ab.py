"""A/B of the strip backward between the shipped library and a variant build (tools/probe/variant.py), on the bench scene's state:
outputs compared bit for bit, timings alternating, best of 3 x 50 launches.
usage: python tools/probe/strip_ab.py tools/probe/libgps_<variant>.so [width height gaussians [fx fy cx cy]] [--frames n]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from bench_kernels import _python_twin, _time_launches
from gps_slam_amd._lib import lib, load_library

alt = load_library(sys.argv[1])
a = [x for x in sys.argv[2:] if not x.startswith("--")]
W, H, NG = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (640, 480, 200000)
intr = tuple(float(x) for x in a[3:7]) if len(a) >= 7 else None
n_frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 31
dev = "cuda:0"
bench.prime(dev)
seq = bench.synthetic_sequence_device(W, H, n_frames, 1234, dev, intrinsics=intr)
seeds = bench.seed_gaussians(seq, NG, 1234, dev) if NG > 0 else None
scene = bench.Scene(seq, seeds, 1234, True, False, n_frames, 1.0, 0.02)
scene.run(0, n_frames)
model, cam, rc = _python_twin(scene, dev)
model.initOptimizers(-1, 1.0)
lib.gps_set_frame_chain_reserve(0)
alt.gps_set_frame_chain_reserve(0)
model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
torch.cuda.synchronize()
B, st = model._B, model._step
N = st.N
counts = B["counts"].cpu().tolist()
r = B["radii"][:N]
print("%dx%d  N %d  visible %d  class counts %s  max radius %d" % (W, H, N, counts[3], B["cls_counts"].cpu().tolist()[:5], int(r.max())))
stream = torch.cuda.current_stream()
sp = C.c_void_p(stream.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())


def bwd(L, rows):
    return lambda: L.gps_raster_ges_bwd_strips(N, p(B["records"]), p(B["radii"]), p(B["cls_ids"]), p(B["cls_counts"]), st.cls_stride,
                                               p(B["v_render_colors"]), p(B["pix2"]), W, H, p(rows), sp)


ra, rb = torch.full_like(B["v_rows"], float("nan")), torch.full_like(B["v_rows"], float("nan"))
bwd(lib, ra)(); bwd(alt, rb)()
torch.cuda.synchronize()
vis = (r > 0)
same = torch.equal(ra[:N][vis][:, :10], rb[:N][vis][:, :10])
print("rows of the visible Gaussians equal bit for bit: %s" % same)
best = {}
for _ in range(3):
    for name, L, rows in (("shipped", lib, ra), (os.path.basename(sys.argv[1]), alt, rb)):
        t = 1e6 * _time_launches(bwd(L, rows), 50, stream)
        best[name] = min(best.get(name, 1e9), t)
for k, v in best.items():
    print("strip backward %-28s %.1f us" % (k, v))
scene.close()

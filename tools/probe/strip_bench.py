"""Backward rasterizer: the column-strip kernel (gps_raster_ges_bwd_strips) against the group kernel (gps_raster_ges_bwd_gs) on
the bench scene's optimise iteration: outputs compared, both timed (HIP events, 50 launches, alternating, best of 3).
usage: python tools/probe/strip_bench.py [variant .so ...]   (each variant's strip kernel is timed as well)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from bench_kernels import _python_twin, _time_launches
from gps_slam_amd._lib import lib, load_library

W, H, NG = 640, 480, 200000
seq = bench.synthetic_sequence(W, H, 31, 1234)
seeds = bench.seed_gaussians(seq, NG, 1234, "cuda:0")
scene = bench.Scene(seq, seeds, 1234, True, False, 31, 1.0, 0.02)
scene.run(0, 31)
model, cam, rc = _python_twin(scene, "cuda:0", strip_backward=False)   # (group tables for the group kernel)
model.initOptimizers(-1, 1.0)
model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
torch.cuda.synchronize()
B, st = model._B, model._step
N = st.N
stream = torch.cuda.current_stream()
sp = C.c_void_p(stream.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
ref = rc["depth_map_clamped"]
radii = B["radii"][:N]
# class lists: class k = smallest 4 << k >= r (k = 4: everything wider), ascending ids
cls = torch.full((N,), -1, dtype=torch.long, device="cuda:0")
vis = radii > 0
cls[vis] = torch.bucketize(radii[vis].long(), torch.tensor([4, 8, 16, 32], device="cuda:0"), right=False)
stride = N
ids = torch.zeros((5, stride), dtype=torch.int32, device="cuda:0")
counts = torch.zeros(5, dtype=torch.int32, device="cuda:0")
for k in range(5):
    sel = torch.nonzero(cls == k)[:, 0].int()
    ids[k, :sel.numel()] = sel
    counts[k] = sel.numel()
print("N %d  class counts %s" % (N, counts.tolist()))
pix2 = torch.empty((H, W, 2), device="cuda:0")
assert lib.gps_raster_pair_image(W, H, p(B["v_render_alphas"]), p(ref), model.delta_depth, p(pix2), sp) == 0
rows = torch.full((N, 12), float("nan"), device="cuda:0")


def strips(L):
    return lambda: L.gps_raster_ges_bwd_strips(N, p(B["records"]), p(B["radii"]), p(ids), p(counts), stride, p(B["v_render_colors"]),
                                               p(pix2), W, H, p(rows), sp)


def groups():
    return lib.gps_raster_ges_bwd_gs(N, p(B["means2d"]), p(B["conics"]), p(B["colors"]), p(B["opacities"]), p(B["radii"]), p(ref), W, H,
                                     p(B["group_gs_ids"]), p(B["group_starts"]), p(B["counts"]), model.delta_depth,
                                     p(B["v_render_colors"]), p(B["v_render_alphas"]), p(B["v_means2d"]), p(B["v_conics"]),
                                     p(B["v_colors"]), p(B["v_opacities"]), 0, sp)


assert groups() == 0
assert strips(lib)() == 0
torch.cuda.synchronize()
want = torch.cat([B["v_colors"][:N], B["v_conics"][:N], B["v_means2d"][:N], B["v_opacities"][:N].reshape(N, 1)], 1)
got = rows[:, :10]
v = vis
print("rows written for every visible Gaussian: %s; untouched elsewhere: %s" % (bool(torch.isfinite(got[v]).all()), bool(torch.isnan(got[~v]).all())))
for name, sl in (("v_colors", slice(0, 4)), ("v_conics", slice(4, 7)), ("v_means2d", slice(7, 9)), ("v_opacities", slice(9, 10))):
    a, b = got[v][:, sl], want[v][:, sl]
    scale = b.abs().max()
    d = (a - b).abs()
    print("%-12s max |diff| %.3e  (max |value| %.3e)  rel-to-max %.2e   rows off by > 1e-4 of max: %d" %
          (name, float(d.max()), float(scale), float(d.max() / scale), int((d.amax(1) > 1e-4 * scale).sum())))
libs = [("shipped", lib)] + [(os.path.basename(a), load_library(a)) for a in sys.argv[1:]]
for _ in range(2):
    t_g = min(1e6 * _time_launches(groups, 50, stream) for _ in range(3))
    line = "group kernel %.1f us |" % t_g
    for name, L in libs:
        line += " strips[%s] %.1f us |" % (name, min(1e6 * _time_launches(strips(L), 50, stream) for _ in range(3)))
    print(line)
scene.close()

"""Strip backward (gps_raster_ges_bwd_strips) on the bench scene's optimise iteration with two orders of the class lists: ascending
Gaussian id, and the binning's (image band, id) order.  Same rows either way (a Gaussian's row does not depend on its place in
the list); what changes is which part of the gradient images an XCD's eighth of a list gathers from.
usage: python tools/probe/strip_order.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from bench_kernels import _python_twin, _time_launches
from gps_slam_amd._lib import lib

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
stream = torch.cuda.current_stream()
sp = C.c_void_p(stream.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
stride = B["cls_ids"].shape[1]
by_band, counts = B["cls_ids"], B["cls_counts"]
by_id = torch.zeros_like(by_band)
for k in range(5):
    n = int(counts[k])
    by_id[k, :n] = by_band[k, :n].sort().values
print("N %d  class counts %s" % (N, counts[:5].tolist()))
rows = {}
for name, ids in (("id order", by_id), ("band order", by_band)):
    out = torch.full((N, 12), float("nan"), device="cuda:0")
    f = lambda: lib.gps_raster_ges_bwd_strips(N, p(B["records"]), p(B["radii"]), p(ids), p(counts), stride, p(B["v_render_colors"]),
                                              p(B["pix2"]), W, H, p(out), sp)
    assert f() == 0
    torch.cuda.synchronize()
    rows[name] = out.clone()
    print("%-10s %s us" % (name, " ".join("%.1f" % min(1e6 * _time_launches(f, 50, stream) for _ in range(3)) for _ in range(2))))
vis = B["radii"][:N] > 0
print("rows bit-equal on visible Gaussians:", torch.equal(rows["id order"][vis], rows["band order"][vis]))
scene.close()
